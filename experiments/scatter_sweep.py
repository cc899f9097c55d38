#!/usr/bin/env python3
"""Sweep the radix fan-out P (and staged vs direct scatter) on a C2-shaped input whose group count
is small enough that every P is valid; prints per-phase hipEvent times.  GPU box only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
from bench import make_shard

n = int(os.environ.get("ROWS", 100_000_000)); g = int(os.environ.get("GROUPS", 100_000)); ncol = int(os.environ.get("COLS", 4))
keys, vals = make_shard(torch, n, g, ncol, 7, "cuda:0")
aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
ctx = pa.Context(0)
print(torch.cuda.get_device_properties(0))
for staged in (1, 0):
    for P in [int(x) for x in os.environ.get("PS", "160,256,384,512,768,1024,1536,2048,4096").split(",")]:
        ctx.set_option("scatter_staged", staged); ctx.set_option("partitions", P); ctx.set_option("groups_hint", g)
        best = None
        for it in range(3):
            ctx.groupby_compute([(keys, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs)
            t = ctx.timings()
            if best is None or t["total_ms"] < best["total_ms"]: best = t
        print(json.dumps({"staged": staged, "P": P, "total_ms": round(best["total_ms"], 3),
                          **{k: round(v, 3) for k, v in best["phase_ms"].items()}, "retries": best["retries"]}), flush=True)
