#!/usr/bin/env python3
"""Fused join -> groupby-sum with skewed probe keys / skewed groups: looking for cliffs.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
MIX = -7046029254386353131
for kv in os.environ.get("PANDRS_OPTS", "").split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for nl, nr in ((100_000_000, 10_000_000), (62_500_000, 50_000_000)):
    rkey = torch.randperm(nr, device=d, generator=gen) * MIX
    for gname, rgrp in (("100K groups", torch.randint(0, 100_000, (nr,), device=d, generator=gen, dtype=torch.int64)),
                        ("groups: 60 % of the build rows in 16, the rest one each", torch.where(torch.rand(nr, device=d, generator=gen) < 0.6, torch.randint(0, 16, (nr,), device=d, generator=gen), 1000 + torch.arange(nr, device=d)))):
        for pname, share, hot in (("uniform", 0.0, 1), ("half the probe rows on ONE build key", 0.5, 1), ("80 % on 1000 build keys", 0.8, 1000)):
            sel = torch.rand(nl, device=d, generator=gen) < share
            lkey = torch.where(sel, torch.randint(0, hot, (nl,), device=d, generator=gen), torch.randint(0, nr, (nl,), device=d, generator=gen))
            lkey = rkey[lkey]
            del sel
            lval = torch.randn(nl, device=d, generator=gen, dtype=torch.float64)
            best = None
            for _ in range(3):
                ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr)
                t = ctx.timings()
                if best is None or t["total_ms"] < best["total_ms"]: best = t
            print("%dM x %dM | %s | probe: %s: %.2f ms  retries %d  %s" % (nl // 10**6, nr // 10**6, gname, pname, best["total_ms"], best["retries"],
                  {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
            del lkey, lval
    del rkey
