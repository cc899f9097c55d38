import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
from bench import make_shard
n, g, ncol = 100_000_000, 1_000_000, 4
keys, vals = make_shard(torch, n, g, ncol, 43, "cuda:0")
aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
ctx = pa.Context(0)
K = [(keys, None, pa.I64)]; V = [(v, None, pa.F64) for v in vals]
for it in range(3):
    ctx.groupby_compute(K, n, V, aggs); print("final   ", json.dumps(ctx.timings()["phase_ms"]))
    ctx.groupby_partials(K, n, V, aggs); print("partials", json.dumps(ctx.timings()["phase_ms"]))
