#!/usr/bin/env python3
"""C2's 80/20-skew variant (enhanced_comprehensive_benchmark.rs:53-59) and heavier skews, 100 M rows / 1 M groups,
sum/mean/min/max x 4 f64 columns.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(1)
n, g = 100_000_000, 1_000_000
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(4)]
aggs4 = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
CASES = ((0.0, 1), (0.8, g // 5), (0.8, 2000), (0.95, 2000), (0.5, 1), (0.9, 1), (0.8, 1000), (0.9, 500), (0.7, 100))
for share, hot in [CASES[int(i)] for i in sys.argv[1:]] or CASES:
    sel = torch.rand(n, device=d, generator=gen) < share
    k = torch.where(sel, torch.randint(0, hot, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)) * -7046029254386353131
    del sel
    best = None
    for _ in range(4):
        ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs4)
        t = ctx.timings()
        if best is None or t["total_ms"] < best["total_ms"]: best = t
    print("%3.0f %% of the rows on %7d keys: %.3f ms  P %d  absorbed %.1f %%  retries %d  %s" % (share * 100, hot, best["total_ms"], best["n_partitions"],
          100.0 * best["absorbed_rows"] / n, best["retries"], {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
    del k
