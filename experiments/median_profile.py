#!/usr/bin/env python3
"""Median at C2 scale (100 M rows, 1 M groups) for a rocprofv3 --kernel-trace run.  GPU box only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
d = "cuda:0"
ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
n, g = int(os.environ.get("ROWS", 100_000_000)), int(os.environ.get("GROUPS", 1_000_000))
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
op = pa.NUNIQUE if os.environ.get("OP") == "nunique" else pa.MEDIAN
ctx.set_option("median_generic", int(os.environ.get("GENERIC", "0")))
for i in range(3):
    ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, op)])
    t = ctx.timings()
    print(json.dumps({"ms": round(t["total_ms"], 3), "phases": {a: round(b, 3) for a, b in t["phase_ms"].items()}}), flush=True)
