#!/usr/bin/env python3
"""rocprofv3 --kernel-trace target: the 80/20 variant of C2, a few calls (census on).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
n, g, ncol = 100_000_000, 1_000_000, 4
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
k = torch.where(torch.rand(n, device=d, generator=gen) < 0.8, torch.randint(0, g // 5, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)) * -7046029254386353131
for i in range(6):
    ng = ctx.groupby_compute([(k, None, pa.I64)], n, v, aggs)
    t = ctx.timings()
    print("call %d: %d groups est %d total %.3f ms P=%d retries=%d %s" % (i, ng, t["estimated_groups"], t["total_ms"], t["n_partitions"], t["retries"], {a: round(b, 3) for a, b in t["phase_ms"].items() if b > 0.005}), flush=True)
