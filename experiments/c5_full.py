#!/usr/bin/env python3
"""C5 at full size on one GPU (500 M x 50 M -> 100 K groups), 4 calls: for kernel traces.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
MIX = -7046029254386353131
nl, nr, g = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (500_000_000, 50_000_000, 100_000)))
rkey = torch.randperm(nr, device=d, generator=gen) * MIX
rgrp = torch.randint(0, g, (nr,), device=d, generator=gen, dtype=torch.int64)
lkey = torch.randint(0, nr, (nl,), device=d, generator=gen, dtype=torch.int64) * MIX
lval = torch.randn(nl, device=d, generator=gen, dtype=torch.float64)
for _ in range(4):
    ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr)
    print("%.2f ms" % ctx.timings()["total_ms"], flush=True)
