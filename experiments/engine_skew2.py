#!/usr/bin/env python3
"""groupby-sum of 62.5 M rows whose keys are: 60 % in 16 hot groups, the rest one group each (25 M groups) — the pair groupby of a fused join with such a group column.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
for kv in os.environ.get("PANDRS_OPTS", "").split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
gen = torch.Generator(device=d); gen.manual_seed(5)
n = 62_500_000
for name, k in (("60 % in 16 groups + one group per remaining row", torch.where(torch.rand(n, device=d, generator=gen) < 0.6, torch.randint(0, 16, (n,), device=d, generator=gen), 1000 + torch.arange(n, device=d))),
                ("uniform 25 M groups", torch.randint(0, 25_000_000, (n,), device=d, generator=gen, dtype=torch.int64))):
    k = k * -7046029254386353131
    v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    best = None
    for _ in range(3):
        ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)])
        t = ctx.timings()
        if best is None or t["total_ms"] < best["total_ms"]: best = t
    print("%s: %.2f ms  P %d  retries %d  est %d  %s" % (name, best["total_ms"], best["n_partitions"], best["retries"], best["estimated_groups"], {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
