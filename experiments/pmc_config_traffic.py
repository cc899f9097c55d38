#!/usr/bin/env python3
"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of experiments/one_config.py <cfg> -> per-kernel HBM bytes per launch and the
pipeline total per call.  usage: pmc_config_traffic.py cfg FETCH_csv WRITE_csv calls > traffic.json
FETCH_SIZE is doubled (guide MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-byte read requests at 64 B; calibrated in round 2
on a pure 8-byte-per-lane read of 4 GB: 3.98 GB counted)."""
import csv, json, sys, collections

ALG = {  # every input byte read once, every output byte written once
    "north_star": 100_000_000 * 16 + 1_000_000 * 16,
    "c3": 100_000_000 * (4 + 16) + 10_000 * (4 + 8 * 9),
    "c4_shard": 125_000_000 * 16 + 10_000_000 * 24,
    "c5_shard": 62_500_000 * 16 + 50_000_000 * 16 + 100_000 * 16,
    "c5_one_gpu": 500_000_000 * 16 + 50_000_000 * 16 + 100_000 * 16,
    "join_indices": 50_000_000 * 8 + 5_000_000 * 8 + 50_000_000 * 16,
}

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

cfg, calls = sys.argv[1], int(sys.argv[4])
fetch, write = load(sys.argv[2], "FETCH_SIZE"), load(sys.argv[3], "WRITE_SIZE")
kernels, total = {}, 0.0
for k in sorted(set(fetch) | set(write)):
    f, fc = fetch.get(k, [0.0, 1]); w, wc = write.get(k, [0.0, 1])
    f_b, w_b = f / max(fc, 1) * 1024.0, w / max(wc, 1) * 1024.0
    per_call = max(fc, wc) / calls
    kernels[k] = {"FETCH_bytes_corrected_x2": 2 * f_b, "WRITE_SIZE_bytes": w_b, "hbm_bytes": 2 * f_b + w_b, "launches_per_call": per_call}
    total += (2 * f_b + w_b) * per_call
print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on experiments/one_config.py %s, MI355X" % cfg,
                  "units": "bytes per launch, averaged over launches; FETCH_SIZE x2 (gfx950 correction)", "config": cfg, "calls_in_run": calls,
                  "kernels": kernels, "pipeline_hbm_bytes_per_call": total, "algorithmic_bytes": ALG[cfg],
                  "traffic_over_algorithmic": total / ALG[cfg]}, indent=1))
