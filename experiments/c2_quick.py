#!/usr/bin/env python3
"""C2 + north-star + C4-shard phase times with option sets from argv: name=value,...  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(1)
n, g = 100_000_000, 1_000_000
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(4)]
aggs4 = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
def best(fn, reps=5):
    b = None
    for _ in range(reps):
        fn(); t = ctx.timings()
        if b is None or t["total_ms"] < b["total_ms"]: b = t
    return b
def show(tag, t):
    print("%-44s total %.3f  P %5d T %5d  " % (tag, t["total_ms"], t["n_partitions"], t["table_slots"]) +
          "  ".join("%s %.3f" % (p, v) for p, v in t["phase_ms"].items()), flush=True)
c2 = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs4)
ns = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(vals[0], None, pa.F64)], [(0, pa.SUM)])
for optset in sys.argv[1:] or [""]:
    opts = [kv.split("=") for kv in optset.split(",") if kv]
    for name, val in opts: ctx.set_option(name, int(val))
    show("C2 [%s]" % optset, best(c2))
    show("north-star sum [%s]" % optset, best(ns))
    for name, val in opts: ctx.set_option(name, 0)
