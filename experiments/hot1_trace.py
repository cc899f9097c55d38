#!/usr/bin/env python3
"""C2's shape with 90 % / 50 % of the rows on ONE key: which path, and what does a forced absorb pass cost?  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
g, n, ncol = 1_000_000, 100_000_000, 4
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
ids = torch.randint(0, g, (n,), device=d, generator=gen)
for share in (0.9, 0.75):
    k = torch.where(torch.rand(n, device=d, generator=gen) < share, torch.zeros_like(ids), ids) * -7046029254386353131
    for force in (0, 1):
        ctx.set_option("no_runs", force)
        for i in range(3):
            if i == 2: os.environ["PANDRS_HIP_ENGINE_TRACE"] = "1"
            ctx.groupby_compute([(k, None, pa.I64)], n, v, aggs)
            os.environ.pop("PANDRS_HIP_ENGINE_TRACE", None)
        t = ctx.timings()
        print("%.0f %% on one key, no_runs=%d: total %.2f absorbed %d  %s" % (share * 100, force, t["total_ms"], t["absorbed_rows"], {k2: round(x, 2) for k2, x in t["phase_ms"].items() if x > 0.005}), flush=True)
    ctx.set_option("no_runs", 0)
    del k
