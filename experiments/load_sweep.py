"""LDS table load factor (option load_pct) against total time on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_shard
d = "cuda:0"; ctx = pa.Context(0)
n, g = 100_000_000, 1_000_000
keys, vals = make_shard(torch, n, g, 4, 43, d)
aggs = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
torch.cuda.synchronize()
for pct in (0, 50, 55, 60, 65, 70, 75):
    ctx.set_option("load_pct", pct)
    best = None
    for _ in range(4):
        ctx.groupby_compute([(keys, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs)
        t = ctx.timings()
        if best is None or t["total_ms"] < best["total_ms"]: best = t
    print("load_pct %2d  P %5d  total %.3f  %s" % (pct, best["n_partitions"], best["total_ms"], {a: round(b, 3) for a, b in best["phase_ms"].items()}), flush=True)
