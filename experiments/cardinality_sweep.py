"""100 M rows, one f64 column (sum + count), group count from 10 to 100 M: where are the cliffs?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(9)
n = 100_000_000
v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
for g in (10, 1_000, 4_000, 10_000, 50_000, 100_000, 1_000_000, 5_000_000, 10_000_000, 20_000_000, 50_000_000, 100_000_000):
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
    best = None
    for _ in range(3):
        ng = ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM), (0, pa.COUNT)])
        t = ctx.timings()
        if best is None or t["total_ms"] < best["total_ms"]: best = t
    print("G %10d  groups %10d  est %10d  P %5d  retries %d  %.2f ms  %5.1f Grows/s  %s" % (g, ng, best["estimated_groups"], best["n_partitions"], best["retries"],
          best["total_ms"], n / best["total_ms"] / 1e6, {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
    del k
