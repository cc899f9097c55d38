#!/usr/bin/env python3
"""Median with one key on a large share of the rows (its hash partition also holds ~100 ordinary keys)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
d = "cuda:0"
ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
n, g = 100_000_000, 1_000_000
for share in [float(x) for x in os.environ.get("SHARES", "0.0,0.1,0.5,0.9").split(",")]:
    ids = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64)
    ids[torch.rand(n, device=d, generator=gen) < share] = 4242
    k = ids * -7046029254386353131
    v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    best = 1e9
    for _ in range(2):
        ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.MEDIAN)])
        best = min(best, ctx.timings()["total_ms"])
    print(json.dumps({"hot_share": share, "ms": round(best, 3)}), flush=True)
    del ids, k, v
