#!/usr/bin/env python3
"""Median / Nunique / Std / First+Last / row lists by GROUP COUNT (uniform keys): looking for cliffs.  50 M rows.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
n = 50_000_000
v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
for g in (1, 10, 1_000, 30_000, 1_000_000, 10_000_000, 40_000_000):
    k = torch.randint(0, g, (n,), device=d, generator=gen) * -7046029254386353131
    row = []
    for name, fn in (("median", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.MEDIAN)])),
                     ("nunique", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.NUNIQUE)])),
                     ("std", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.STD)])),
                     ("first+last", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.FIRST), (0, pa.LAST)])),
                     ("sum", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)])),
                     ("row lists", lambda: ctx.groupby_indices([(k, None, pa.I64)], n))):
        best = None
        try:
            for _ in range(2):
                out = fn(); del out
                t = ctx.timings()["total_ms"]
                best = t if best is None else min(best, t)
            row.append("%s %.2f" % (name, best))
        except Exception as e:
            row.append("%s FAILED (%s)" % (name, str(e)[:40]))
    print("%9d groups: %s  (ms)" % (g, "  ".join(row)), flush=True)
    del k
