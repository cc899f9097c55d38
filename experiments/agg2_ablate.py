#!/usr/bin/env python3
"""C2 / north-star / C4-shard aggregate phase: old kernel (agg_v1) vs aggregate2, and aggregate2's ablations
(1 no min/max, 2 lookup + group size only, 3 HBM stream only).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(1)
n, g = 100_000_000, 1_000_000
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(4)]
aggs4 = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]

def best(fn, reps=4):
    b = None
    for _ in range(reps):
        fn(); t = ctx.timings()
        if b is None or t["total_ms"] < b["total_ms"]: b = t
    return b

def show(tag, t):
    print("%-34s total %.3f  P %5d T %5d  " % (tag, t["total_ms"], t["n_partitions"], t["table_slots"]) +
          "  ".join("%s %.3f" % (p, v) for p, v in t["phase_ms"].items()), flush=True)

c2 = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs4)
ns = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(vals[0], None, pa.F64)], [(0, pa.SUM)])
for v1 in (1, 0):
    ctx.set_option("agg_v1", v1)
    show("C2 agg_v1=%d" % v1, best(c2))
    show("north-star sum agg_v1=%d" % v1, best(ns))
ctx.set_option("agg_v1", 0)
for ab in (1, 2, 3):
    ctx.set_option("agg_ablate", ab)
    show("C2 aggregate2 ablate=%d" % ab, best(c2))
ctx.set_option("agg_ablate", 0)
for P in (768, 1024, 1280, 1536):
    ctx.set_option("partitions", P)
    try:
        show("C2 aggregate2 P=%d" % P, best(c2))
    except pa.PandrsHipError as e:
        print("P=%d: %s" % (P, e))
ctx.set_option("partitions", 0)
