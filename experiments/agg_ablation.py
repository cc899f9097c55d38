#!/usr/bin/env python3
"""Aggregate-kernel ablation on the C2 input: which per-row LDS work costs what.  GPU box only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
from bench import make_shard
n, g, ncol = 100_000_000, 1_000_000, 4
keys, vals = make_shard(torch, n, g, ncol, 43, "cuda:0")
ctx = pa.Context(0)
K = [(keys, None, pa.I64)]; V = [(v, None, pa.F64) for v in vals]
sets = {
  "sum+mean+min+max x4": [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)],
  "sum x4": [(c, pa.SUM) for c in range(4)],
  "min+max x4": [(c, op) for c in range(4) for op in (pa.MIN, pa.MAX)],
  "min x4": [(c, pa.MIN) for c in range(4)],
  "sum x2": [(c, pa.SUM) for c in range(2)],
  "sum x1": [(0, pa.SUM)],
  "count only (1 col moved)": [(0, pa.COUNT)],
}
for name, aggs in sets.items():
    for P in (0, 1152):
        ctx.set_option("partitions", P)
        best = None
        for it in range(3):
            ctx.groupby_compute(K, n, V, aggs); t = ctx.timings()
            if best is None or t["phase_ms"]["aggregate"] < best["phase_ms"]["aggregate"]: best = t
        print(json.dumps({"aggs": name, "P": best["n_partitions"], "T": best["table_slots"], "aggregate": round(best["phase_ms"]["aggregate"], 3), "scatter": round(best["phase_ms"]["scatter"], 3), "total": round(best["total_ms"], 3)}), flush=True)
