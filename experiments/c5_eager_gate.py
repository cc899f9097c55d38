#!/usr/bin/env python3
"""The gate for eager aggregation in C5: accumulating v per BUILD ENTRY needs the accumulators in LDS (device-scope
atomics retire at 23.7 G/s, experiments/ubench/l2_atomic.hip: 500 M of them = 21 ms), i.e. the LDS-multimap path, whose probe side
must be partitioned as finely as the build side — P = 8192 for 50 M build rows, in two passes.  This measures exactly that partition
on C5's probe side (500 M rows x 16 bytes) by forcing the LDS-multimap path (join_no_l2 = 1) and reading its phases; go only if the
two passes take <= 7.5 ms.  GPU box only."""
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
for no_l2 in (0, 1):
    ctx.set_option("join_no_l2", no_l2)
    for i in range(3):
        ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr)
    t = ctx.timings()
    print("join_no_l2=%d: total %.2f ms  P=%d  %s" % (no_l2, t["total_ms"], t["n_partitions"], {a: round(b, 2) for a, b in t["phase_ms"].items() if b > 0.01}), flush=True)
