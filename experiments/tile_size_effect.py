#!/usr/bin/env python3
"""Does the scatter's run length matter at P = 1024 with 5 columns?  Exact partition with 4096-row tiles (512 threads) against
8192-row tiles (1024 threads): the same fan-out and write frontier, half the run length.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(1)
n, g = 100_000_000, 1_000_000
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(4)]
def best(fn, reps=5):
    b = None
    for _ in range(reps):
        fn(); t = ctx.timings()
        if b is None or t["total_ms"] < b["total_ms"]: b = t
    return b
ctx.set_option("exact_partition", 1)
for P in (512, 1024):
    ctx.set_option("partitions", P)
    for th in (1024, 512):
        ctx.set_option("scatter_threads", th)
        aggs = [(c, pa.SUM) for c in range(4)]
        t = best(lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs))
        print("P=%4d tile %5d rows: total %.3f  %s" % (t["n_partitions"], th * 8, t["total_ms"], " ".join("%s %.3f" % kv for kv in t["phase_ms"].items())), flush=True)
