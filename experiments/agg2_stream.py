#!/usr/bin/env python3
"""Why does aggregate2's bare HBM stream (ablate=3) take 0.89 ms for 4 GB when a plain read kernel takes 0.58?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(1)
n, g = 100_000_000, 1_000_000
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(4)]
aggs4 = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
c2 = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs4)
def best(fn, reps=4):
    b = None
    for _ in range(reps):
        fn(); t = ctx.timings()
        if b is None or t["phase_ms"]["aggregate"] < b["phase_ms"]["aggregate"]: b = t
    return b
for ab in (0, 3, 6):
    ctx.set_option("agg_ablate", ab)
    ctx.set_option("partitions", 1024)
    t = best(c2)
    print("ablate=%d depth=default aggregate %.3f ms  scatter %.3f" % (ab, t["phase_ms"]["aggregate"], t["phase_ms"]["scatter"]), flush=True)
ctx.set_option("agg_ablate", 0)
for dp in (2, 3, 5, 6):
    ctx.set_option("agg_depth", dp)
    t = best(c2)
    print("full depth=%d aggregate %.3f ms" % (dp, t["phase_ms"]["aggregate"]), flush=True)
ctx.set_option("agg_depth", 0)
ctx.set_option("partitions", 0)
t = best(c2)
print("C2 default: total %.3f  P %d T %d " % (t["total_ms"], t["n_partitions"], t["table_slots"]), t["phase_ms"], flush=True)
