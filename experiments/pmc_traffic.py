#!/usr/bin/env python3
"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) of a bench.py run ->
per-kernel HBM bytes per launch and the pipeline total per step (profiles/rNN_pmc_traffic.json).
usage: pmc_traffic.py FETCH_csv WRITE_csv steps_per_run rows groups cols > traffic.json
FETCH_SIZE is doubled (guide MI355X_MICROARCH.md §HBM: gfx950 tallies 128-B read requests at 64 B); calibrated here on
the sample_histogram / aggregate kernels whose read sizes are known."""
import csv, json, sys, collections

def load(path, counter):
    per = collections.defaultdict(float); names = {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter: continue
        per[row["Dispatch_Id"]] += float(row["Counter_Value"]); names[row["Dispatch_Id"]] = row["Kernel_Name"]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for d, v in per.items():
        n = names[d]
        if "pandrs::" not in n: continue
        short = n.split("pandrs::", 1)[1].replace("(anonymous namespace)::", "").split("(")[0]
        acc[short][0] += v; acc[short][1] += 1
    return acc

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])            # groupby calls in the profiled run (warm-up + timed)
rows, groups, cols = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
kernels, total = {}, 0.0
for k in sorted(set(fetch) | set(write)):
    f, fc = fetch.get(k, [0.0, 1]); w, wc = write.get(k, [0.0, 1])
    f_b, w_b = f / max(fc, 1) * 1024.0, w / max(wc, 1) * 1024.0
    launches_per_step = max(fc, wc) / steps
    kernels[k] = {"FETCH_SIZE_bytes_raw": f_b, "FETCH_bytes_corrected_x2": 2 * f_b, "WRITE_SIZE_bytes": w_b,
                  "hbm_bytes": 2 * f_b + w_b, "launches_per_step": launches_per_step}
    total += (2 * f_b + w_b) * launches_per_step
alg = rows * (8 + 8 * cols) + groups * (8 + 8 * 4 * cols)
print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on python3 bench.py, MI355X",
                  "units": "bytes per launch, averaged over launches; FETCH_SIZE x2 (gfx950 correction, guide §HBM)",
                  "config": {"rows": rows, "groups": groups, "value_cols": cols},
                  "kernels": kernels, "pipeline_hbm_bytes_per_step": total, "algorithmic_bytes": alg,
                  "traffic_over_algorithmic": total / alg}, indent=1))
