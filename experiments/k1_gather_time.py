#!/usr/bin/env python3
"""Whole-column reductions (K1) and join gathers at 100 M rows: both are one HBM pass.  GPU box only."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
d = "cuda:0"
ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(1)
n = 100_000_000
v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
w = torch.randint(-10**9, 10**9, (n,), device=d, generator=gen, dtype=torch.int64)
idx = torch.randint(0, n, (n,), device=d, generator=gen, dtype=torch.int64)
seq = torch.arange(n, device=d, dtype=torch.int64)
def timed(name, fn, bytes_):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    print(json.dumps({"op": name, "wall_ms": round(best, 3), "GB/s": round(bytes_ / best / 1e6, 1)}), flush=True)
timed("reduce_column f64 (sum/mean/min/max)", lambda: ctx.reduce_column((v, None, pa.F64), n), n * 8)
timed("reduce_column i64", lambda: ctx.reduce_column((w, None, pa.I64), n), n * 8)
timed("column_std f64 (population)", lambda: ctx.column_std((v, None, pa.F64), n), n * 8)
timed("gather f64, random indices", lambda: ctx.gather(v, None, idx, 0.0, pa.F64), n * 24)
timed("gather f64, ascending indices", lambda: ctx.gather(v, None, seq, 0.0, pa.F64), n * 24)
