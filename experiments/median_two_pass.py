#!/usr/bin/env python3
"""Median / group_by row lists at 100 M rows with and without the two-pass pair partition.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
n, g = 100_000_000, 1_000_000
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
for tp in (0, -1, 0, -1):
    ctx.set_option("two_pass", tp)
    for op, name in ((pa.MEDIAN, "median"), (pa.NUNIQUE, "nunique")):
        best = None
        for _ in range(3):
            ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, op)])
            t = ctx.timings()
            if best is None or t["total_ms"] < best["total_ms"]: best = t
        print("two_pass=%2d %-8s %.3f ms  %s" % (tp, name, best["total_ms"], {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
ctx.set_option("two_pass", 0)
