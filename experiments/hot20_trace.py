#!/usr/bin/env python3
"""50 M rows x 2 columns, half the rows on 20 keys: what runs besides the phases?  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
g, n, ncol = 1_000_000, 50_000_000, 2
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
ids = torch.randint(0, g, (n,), device=d, generator=gen)
hot = torch.where(torch.rand(n, device=d, generator=gen) < 0.5, torch.randint(0, 20, (n,), device=d, generator=gen), ids) * -7046029254386353131
for i in range(3):
    if i == 2: print("---- third call", flush=True); os.environ["X"] = "1"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.groupby_compute([(hot, None, pa.I64)], n, v, aggs)
    torch.cuda.synchronize(); print("wall %.2f ms" % ((time.perf_counter() - t0) * 1e3), ctx.timings()["total_ms"], flush=True)
