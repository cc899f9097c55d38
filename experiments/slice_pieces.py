#!/usr/bin/env python3
"""Pieces of an oversized partition: as long as the cutting threshold (wide_slices = 1, the round-2 behaviour) or average-sized.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
def bits(n, p):
    m = (torch.rand(n, device=d, generator=gen) < p).view(-1, 8).to(torch.uint8)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], device=d, dtype=torch.uint8)
    return (m * w).sum(1).to(torch.uint8)
def run(name, keys, n, v, aggs):
    out = []
    for wide in (1, 0):
        ctx.set_option("wide_slices", wide)
        best = None
        for _ in range(4):
            ctx.groupby_compute(keys, n, v, aggs); t = ctx.timings()
            if best is None or t["total_ms"] < best["total_ms"]: best = t
        out.append("%s total %.2f aggregate %.2f" % ("wide pieces:" if wide else "average pieces:", best["total_ms"], best["phase_ms"]["aggregate"]))
    ctx.set_option("wide_slices", 0)
    print("%-52s %s" % (name, "   ".join(out)), flush=True)
g = 1_000_000
for n, ncol in ((50_000_000, 2), (100_000_000, 4)):
    v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
    aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
    ids = torch.randint(0, g, (n,), device=d, generator=gen)
    run("%dM rows x %d cols, uniform" % (n // 10**6, ncol), [(ids * -7046029254386353131, None, pa.I64)], n, v, aggs)
    run("%dM rows x %d cols, 5 %% null keys" % (n // 10**6, ncol), [(ids * -7046029254386353131, bits(n, 0.05), pa.I64)], n, v, aggs)
    hot = torch.where(torch.rand(n, device=d, generator=gen) < 0.5, torch.zeros_like(ids), ids)
    run("%dM rows x %d cols, half the rows on one key" % (n // 10**6, ncol), [(hot * -7046029254386353131, None, pa.I64)], n, v, aggs)
    hot = torch.where(torch.rand(n, device=d, generator=gen) < 0.5, torch.randint(0, 20, (n,), device=d, generator=gen), ids)
    run("%dM rows x %d cols, half the rows on 20 keys" % (n // 10**6, ncol), [(hot * -7046029254386353131, None, pa.I64)], n, v, aggs)
    del v, ids, hot
