#!/usr/bin/env python3
"""An estimate that is too low where the plan is ROUNDS (no overflow run behind a full table): what does the retry cost against the single
lean round with its overflow run?  C2's shape, 6 M uniform groups, the estimate forced with `groups_hint`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
n = 100_000_000
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(4)]
aggs = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
k = torch.randint(0, 6_000_000, (n,), device=d, generator=gen) * -7046029254386353131
for hint in (0, 6_000_000, 5_000_000, 4_500_000, 3_500_000):
    for nlr in (0, 1):
        ctx.set_option("groups_hint", hint); ctx.set_option("no_lean_rounds", nlr)
        for i in range(3): ng = ctx.groupby_compute([(k, None, pa.I64)], n, v, aggs)
        t = ctx.timings()
        print("groups_hint %8d no_lean_rounds %d: %6.2f ms  groups %d P=%d T=%d retries=%d" % (hint, nlr, t["total_ms"], ng, t["n_partitions"], t["table_slots"], t["retries"]), flush=True)
