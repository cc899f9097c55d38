#!/usr/bin/env python3
"""Call sizes between the reference's 1 M-row case and the headline's 100 M: where does the fixed cost of a call show?  One f64 sum and
C2's 16 aggregates over random sparse i64 keys.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(4)
MIX = -7046029254386353131
for n in (500_000, 1_000_000, 2_000_000, 2_500_000, 5_000_000, 10_000_000, 20_000_000, 50_000_000):
    v = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(4)]
    for g in (1_000, 100_000, 1_000_000):
        if g * 2 > n: continue
        k = torch.randint(0, g, (n,), device=d, generator=gen) * MIX
        row = []
        for name, nv, aggs in (("sum", 1, [(0, pa.SUM)]), ("4x4", 4, [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)])):
            vals = [(v[i], None, pa.F64) for i in range(nv)]
            for i in range(3): ctx.groupby_compute([(k, None, pa.I64)], n, vals, aggs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(10): ctx.groupby_compute([(k, None, pa.I64)], n, vals, aggs)
            wall = (time.perf_counter() - t0) / 10 * 1e3
            t = ctx.timings()
            row.append("%s %.3f ms wall (%.1f Grows/s, P=%d)" % (name, wall, n / wall / 1e6, t["n_partitions"]))
        print("rows %9d groups %8d   %s" % (n, g, "   ".join(row)), flush=True)
        del k
    del v
