#!/usr/bin/env python3
"""Full-size C5 probe kernel with parts switched off (agg_ablate 1: no pair output, 2: no walk either).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
MIX = -7046029254386353131
nl, nr, g = 500_000_000, 50_000_000, 100_000
rkey = torch.randperm(nr, device=d, generator=gen) * MIX
rgrp = torch.randint(0, g, (nr,), device=d, generator=gen, dtype=torch.int64)
lkey = torch.randint(0, nr, (nl,), device=d, generator=gen, dtype=torch.int64) * MIX
lval = torch.randn(nl, device=d, generator=gen, dtype=torch.float64)
for ab in (0, 1, 2):
    ctx.set_option("agg_ablate", ab)
    for _ in range(2):
        try:
            ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr)
        except Exception as e:
            print("err", e)
        t = ctx.timings()
    print("ablate", ab, "total %.2f" % t["total_ms"], {a: round(b, 2) for a, b in t["phase_ms"].items()}, flush=True)
