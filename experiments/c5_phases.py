#!/usr/bin/env python3
"""C5 fused join -> groupby-sum: phase times at the per-GPU shard size and at full size on one GPU.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
MIX = -7046029254386353131
for optset in sys.argv[1:] or [""]:
    opts = [kv.split("=") for kv in optset.split(",") if kv]
    for name, val in opts: ctx.set_option(name, int(val))
    for nl, nr, g in ((62_500_000, 50_000_000, 100_000), (50_000_000, 5_000_000, 100_000), (500_000_000, 50_000_000, 100_000)):
        rkey = torch.randperm(nr, device=d, generator=gen) * MIX
        rgrp = torch.randint(0, g, (nr,), device=d, generator=gen, dtype=torch.int64)
        lkey = torch.randint(0, nr, (nl,), device=d, generator=gen, dtype=torch.int64) * MIX
        lval = torch.randn(nl, device=d, generator=gen, dtype=torch.float64)
        best = None
        for _ in range(3):
            ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr)
            t = ctx.timings()
            if best is None or t["total_ms"] < best["total_ms"]: best = t
        print("[%s] %dM x %dM: %.2f ms  P %d  %s" % (optset, nl // 10**6, nr // 10**6, best["total_ms"], best["n_partitions"],
              {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
        del rkey, rgrp, lkey, lval
    for name, val in opts: ctx.set_option(name, 0)
