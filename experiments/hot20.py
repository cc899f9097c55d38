#!/usr/bin/env python3
"""Half the rows on 20 keys among 1 M groups: phases.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
if os.environ.get("FORCE_ABSORB"): ctx.set_option("no_absorb", -1)
gen = torch.Generator(device=d); gen.manual_seed(5)
g = 1_000_000
for n, ncol in ((50_000_000, 2), (100_000_000, 4)):
    v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
    aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
    ids = torch.randint(0, g, (n,), device=d, generator=gen)
    for hotn, share in ((20, 0.5), (20, 0.7), (200, 0.5), (3, 0.5)):
        hot = torch.where(torch.rand(n, device=d, generator=gen) < share, torch.randint(0, hotn, (n,), device=d, generator=gen), ids) * -7046029254386353131
        for _ in range(3): ctx.groupby_compute([(hot, None, pa.I64)], n, v, aggs)
        t = ctx.timings()
        print("%dM x %d cols, %.0f %% on %d keys: total %.2f P=%d retries=%d absorbed=%d  " % (n // 10**6, ncol, share * 100, hotn, t["total_ms"], t["n_partitions"], t["retries"], t["absorbed_rows"]) +
              "  ".join("%s %.2f" % (k, ms) for k, ms in t["phase_ms"].items() if ms > 0.005), flush=True)
        del hot
    del v, ids
