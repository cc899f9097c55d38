#!/usr/bin/env python3
"""C5 whole (500 M x 50 M): the L2-region probe's pair fan-out.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
MIX = -7046029254386353131
nl, nr, g = 500_000_000, 50_000_000, int(os.environ.get("GROUPS", 100_000))
rkey = torch.randperm(nr, device=d, generator=gen) * MIX
rgrp = torch.randint(0, g, (nr,), device=d, generator=gen, dtype=torch.int64)
lkey = torch.randint(0, nr, (nl,), device=d, generator=gen, dtype=torch.int64) * MIX
lval = torch.randn(nl, device=d, generator=gen, dtype=torch.float64)
for optset in sys.argv[1:] or [""]:
    opts = [kv.split("=") for kv in optset.split(",") if kv]
    for name, val in opts: ctx.set_option(name, int(val))
    best = None
    for _ in range(3):
        ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr)
        t = ctx.timings()
        if best is None or t["total_ms"] < best["total_ms"]: best = t
    print("[%-20s] %.2f ms  retries %d  %s" % (optset, best["total_ms"], best["retries"], {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
    for name, val in opts: ctx.set_option(name, 0)
