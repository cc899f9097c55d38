#!/usr/bin/env python3
"""C2 with 80 % of the rows on 200 K of the 1 M keys: what the estimate sees and what the engine does with it.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
g, n, ncol = 1_000_000, 100_000_000, 4
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
k = torch.where(torch.rand(n, device=d, generator=gen) < 0.8, torch.randint(0, g // 5, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)) * -7046029254386353131
for hint in (0, 1_000_000):
    ctx.set_option("groups_hint", hint)
    for i in range(3):
        if i == 2: os.environ["PANDRS_HIP_ENGINE_TRACE"] = "1"
        ng = ctx.groupby_compute([(k, None, pa.I64)], n, v, aggs)
        os.environ.pop("PANDRS_HIP_ENGINE_TRACE", None)
    t = ctx.timings()
    print("groups_hint %d: %d groups, total %.2f P=%d retries=%d  %s" % (hint, ng, t["total_ms"], t["n_partitions"], t["retries"], {a: round(b, 2) for a, b in t["phase_ms"].items() if b > 0.005}), flush=True)
