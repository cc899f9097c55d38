#!/usr/bin/env python3
"""Two-pass radix partition (partition.hip): bucket count and threshold on the fused join's shard shapes.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
MIX = -7046029254386353131
shapes = ((62_500_000, 50_000_000), (60_000_000, 20_000_000), (40_000_000, 12_000_000))
data = []
for nl, nr in shapes:
    rkey = torch.randperm(nr, device=d, generator=gen) * MIX
    rgrp = torch.randint(0, 100_000, (nr,), device=d, generator=gen, dtype=torch.int64)
    lkey = torch.randint(0, nr, (nl,), device=d, generator=gen, dtype=torch.int64) * MIX
    lval = torch.randn(nl, device=d, generator=gen, dtype=torch.float64)
    data.append((nl, nr, rkey, rgrp, lkey, lval))
for optset in sys.argv[1:] or [""]:
    opts = [kv.split("=") for kv in optset.split(",") if kv]
    for name, val in opts: ctx.set_option(name, int(val))
    for nl, nr, rkey, rgrp, lkey, lval in data:
        best = None
        for _ in range(4):
            ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr)
            t = ctx.timings()
            if best is None or t["total_ms"] < best["total_ms"]: best = t
        print("[%-32s] %dM x %dM: %.2f ms  P %d  %s" % (optset, nl // 10**6, nr // 10**6, best["total_ms"], best["n_partitions"],
              {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
    for name, val in opts: ctx.set_option(name, 0)
