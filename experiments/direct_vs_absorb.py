#!/usr/bin/env python3
"""Few groups, many rows: the direct path (round-1 aggregate kernel in direct mode) against the absorb kernel forced on.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
n = 100_000_000
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(2)]
def best(fn, reps=4):
    b = None
    for _ in range(reps):
        fn(); t = ctx.timings()
        if b is None or t["total_ms"] < b["total_ms"]: b = t
    return b
for g in (3, 100, 1000, 2500):
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
    for name, aggs, nv in (("sum", [(0, pa.SUM)], 1), ("sum/min/max x2", [(c, op) for c in range(2) for op in (pa.SUM, pa.MIN, pa.MAX)], 2)):
        for opts in ({}, {"no_direct": 1, "no_absorb": -1}):
            for o, v in opts.items(): ctx.set_option(o, v)
            t = best(lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(x, None, pa.F64) for x in vals[:nv]], aggs))
            for o in opts: ctx.set_option(o, 0)
            print("g=%5d %-15s %-32s total %.3f ms  absorbed %d  %s" % (g, name, opts or "default (direct path)", t["total_ms"], t["absorbed_rows"],
                  " ".join("%s %.3f" % kv for kv in t["phase_ms"].items())), flush=True)
