#!/usr/bin/env python3
"""Non-mergeable aggregates (Median, Nunique, Std, First/Last) and group_by row lists under skewed keys: looking for cliffs.  50 M rows.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
n, g = 50_000_000, 1_000_000
v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
for share, hot in ((0.0, 1), (0.8, 2000), (0.5, 1), (0.8, g // 5)):
    sel = torch.rand(n, device=d, generator=gen) < share
    k = torch.where(sel, torch.randint(0, hot, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)) * -7046029254386353131
    del sel
    row = []
    for name, fn in (("median", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.MEDIAN)])),
                     ("nunique", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.NUNIQUE)])),
                     ("std", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.STD)])),
                     ("first+last", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.FIRST), (0, pa.LAST)])),
                     ("row lists", lambda: ctx.groupby_indices([(k, None, pa.I64)], n))):
        best = None
        for _ in range(2):
            out = fn(); del out
            t = ctx.timings()["total_ms"]
            best = t if best is None else min(best, t)
        row.append("%s %.2f" % (name, best))
    print("%3.0f %% of the rows on %7d keys: %s  (ms)" % (share * 100, hot, "  ".join(row)), flush=True)
    del k
