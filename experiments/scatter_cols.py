#!/usr/bin/env python3
"""Scatter time against the number of 8-byte columns moved with the key (100 M rows, 1 M groups): COUNT only (key alone), 1..4 sum columns.  GPU box only."""
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
for P in ([int(x) for x in sys.argv[1:]] or (0, 512, 1024)):
    ctx.set_option("partitions", P)
    for nv in range(0, 5):
        aggs = [(c, pa.SUM) for c in range(nv)] or [(0, pa.COUNT)]
        t = best(lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals[:max(nv, 1)]], aggs))
        print("P=%4d (%4d) value columns %d: total %.3f  %s" % (P, t["n_partitions"], nv, t["total_ms"], " ".join("%s %.3f" % kv for kv in t["phase_ms"].items())), flush=True)
