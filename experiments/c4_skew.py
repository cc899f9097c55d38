#!/usr/bin/env python3
"""C4 shard's shape (125 M rows, 10 M groups, sum + count) with skewed keys: looking for cliffs.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(1)
n, g = 125_000_000, 10_000_000
v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
for share, hot in ((0.0, 1), (0.8, g // 5), (0.8, 5000), (0.5, 1), (0.3, 10)):
    sel = torch.rand(n, device=d, generator=gen) < share
    k = torch.where(sel, torch.randint(0, hot, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)) * -7046029254386353131
    del sel
    best = None
    for _ in range(3):
        ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM), (0, pa.COUNT)])
        t = ctx.timings()
        if best is None or t["total_ms"] < best["total_ms"]: best = t
    print("%3.0f %% of the rows on %8d keys: %.3f ms  P %d  absorbed %.1f %%  retries %d  est %d  %s" % (share * 100, hot, best["total_ms"], best["n_partitions"],
          100.0 * best["absorbed_rows"] / n, best["retries"], best["estimated_groups"], {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
    del k
