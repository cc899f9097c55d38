#!/usr/bin/env python3
"""5 % null keys at 1 M groups: where does the extra millisecond go?  50 M rows.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
n, g = 50_000_000, 1_000_000
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(2)]
aggs = [(c, op) for c in range(2) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
def bits(p):
    m = (torch.rand(n, device=d, generator=gen) < p).view(-1, 8).to(torch.uint8)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], device=d, dtype=torch.uint8)
    return (m * w).sum(1).to(torch.uint8)
ids = torch.randint(0, g, (n,), device=d, generator=gen) * -7046029254386353131
for p in (0.0, 0.0001, 0.05, 0.5):
    keys = [(ids, bits(p) if p > 0 else None, pa.I64)]
    for _ in range(3): ctx.groupby_compute(keys, n, v, aggs)
    t = ctx.timings()
    print("%.4f null keys: total %.2f  P=%d retries=%d absorbed=%d  " % (p, t["total_ms"], t["n_partitions"], t["retries"], t["absorbed_rows"]) +
          "  ".join("%s %.2f" % (k, ms) for k, ms in t["phase_ms"].items() if ms > 0.005), flush=True)
