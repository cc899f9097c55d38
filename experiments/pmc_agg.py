#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes for kernels whose name contains a substring.
usage: pmc_agg.py SUBSTR csv..."""
import csv, collections, sys
sub = sys.argv[1]
per = collections.defaultdict(float)
for path in sys.argv[2:]:
    for row in csv.DictReader(open(path)):
        if sub not in row["Kernel_Name"]: continue
        per[(path, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
acc = collections.defaultdict(lambda: [0.0, 0])
for (p, d, c), v in per.items():
    acc[c][0] += v; acc[c][1] += 1
for c, (v, n) in sorted(acc.items()):
    print("%-24s %15.0f  (%d dispatches)" % (c, v / n, n))
