#!/usr/bin/env python3
"""C1 wall time per call for several small-path chunk sizes.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
for n, g in ((1_000_000, 1_000), (100_000, 100), (2_000_000, 5_000)):
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64)
    v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    f = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)])
    for chunk in (0, 4096, 8192, 16384, 32768):
        ctx.set_option("small_chunk", chunk)
        for _ in range(5): f()
        t0 = time.perf_counter()
        for _ in range(300): f()
        wall = (time.perf_counter() - t0) / 300 * 1e6
        print("n %8d g %5d chunk %6d  wall %.1f us  device %.1f us" % (n, g, chunk, wall, ctx.timings()["total_ms"] * 1e3), flush=True)
