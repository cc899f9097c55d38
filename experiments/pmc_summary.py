#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files (one pass per counter) into per-kernel
average counter values.  usage: pmc_summary.py FETCH_csv WRITE_csv > summary.json"""
import csv, json, sys, collections

def load(path):
    acc = collections.defaultdict(lambda: [0.0, 0])
    # one row per (dispatch, counter); several rows per dispatch when the counter is split per XCD/SE
    per_dispatch = collections.defaultdict(float)
    names = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[k] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = row["Kernel_Name"]
    for (d, c), v in per_dispatch.items():
        n = names[d]
        if "pandrs::" not in n:
            continue
        short = n.split("pandrs::")[1].split("(")[0]
        acc[(short, c)][0] += v
        acc[(short, c)][1] += 1
    return {"%s|%s" % k: {"avg": v[0] / v[1], "calls": v[1]} for k, v in acc.items()}

out = {}
for p in sys.argv[1:]:
    out.update(load(p))
print(json.dumps(out, indent=1, sort_keys=True))
