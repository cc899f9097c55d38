#!/usr/bin/env python3
"""C1 alone (1 M rows, 1 K groups, one f64 sum), 200 calls: for kernel traces of the small path.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
n, g = 1_000_000, 1_000
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64)
v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
f = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)])
for _ in range(5): f()
t0 = time.perf_counter()
for _ in range(200): f()
print("wall %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
