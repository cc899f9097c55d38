#!/usr/bin/env python3
"""C2: the local phase of the distributed step (partial states out) against the ordinary call (finished aggregates out).  GPU box only."""
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
for name, fn in (("final", lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs4)),
                 ("partials", lambda: ctx.groupby_partials([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs4))) * 2:
    t = best(fn)
    print("%-9s total %.3f  P %d  %s" % (name, t["total_ms"], t["n_partitions"], "  ".join("%s %.3f" % kv for kv in t["phase_ms"].items())), flush=True)
