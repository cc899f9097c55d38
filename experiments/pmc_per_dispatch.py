#!/usr/bin/env python3
"""Per-dispatch counter values of kernels whose name contains SUBSTR, in dispatch order.  usage: pmc_per_dispatch.py SUBSTR csv"""
import csv, collections, sys
per = collections.OrderedDict()
for row in csv.DictReader(open(sys.argv[2])):
    if sys.argv[1] not in row["Kernel_Name"]: continue
    k = (int(row["Dispatch_Id"]), row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"])
    per[k] = per.get(k, 0.0) + float(row["Counter_Value"])
for (d, n, c), v in sorted(per.items()): print(d, n, c, "%.0f" % v)
