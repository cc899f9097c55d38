#!/usr/bin/env python3
"""Where Median / Nunique / row lists lose their time under skew: phases of the uniform and the 80 %-on-2000-keys shape.  50 M rows.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
n, g = 50_000_000, 1_000_000
v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
for share, hot in ((0.0, 1), (0.8, 2000), (0.5, 1)):
    sel = torch.rand(n, device=d, generator=gen) < share
    k = torch.where(sel, torch.randint(0, hot, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)) * -7046029254386353131
    del sel
    for name, fn in (("median", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.MEDIAN)])),
                     ("row lists", lambda: ctx.groupby_indices([(k, None, pa.I64)], n))):
        for _ in range(2):
            out = fn(); del out
        t = ctx.timings()
        print("%3.0f %% on %5d keys  %-9s total %.2f  P=%d  " % (share * 100, hot, name, t["total_ms"], t["n_partitions"]) +
              "  ".join("%s %.2f" % (p, ms) for p, ms in t["phase_ms"].items() if ms > 0.005), flush=True)
    del k
