#!/usr/bin/env python3
"""Rows SORTED by key (long and short runs), and a dominant key in random order: what the clustered-rows signal decides.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
n, ncol = 100_000_000, 4
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
profiles = {"sum x1": ([(0, pa.SUM)], v[:1]), "C2 aggs": ([(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)], v)}
M = -7046029254386353131
def shapes():
    for g in (100, 10_000, 1_000_000, 20_000_000):
        yield "sorted, %d groups" % g, torch.sort(torch.randint(0, g, (n,), device=d, generator=gen))[0] * M
    yield "runs of 20, 5M groups", torch.repeat_interleave(torch.randint(0, 5_000_000, (n // 20,), device=d, generator=gen), 20) * M
    ids = torch.randint(0, 1_000_000, (n,), device=d, generator=gen)
    yield "90 % on one key + 1M others", torch.where(torch.rand(n, device=d, generator=gen) < 0.9, torch.zeros_like(ids), ids) * M
    yield "75 % on one key + 1M others", torch.where(torch.rand(n, device=d, generator=gen) < 0.75, torch.zeros_like(ids), ids) * M
    yield "90 % on one key + 1K others", torch.where(torch.rand(n, device=d, generator=gen) < 0.9, torch.zeros_like(ids), ids % 1000 + 1) * M
for name, k in shapes():
    row = []
    for pname, (aggs, vals) in profiles.items():
        best = None
        for _ in range(3):
            ctx.groupby_compute([(k, None, pa.I64)], n, vals, aggs); t = ctx.timings()["total_ms"]
            best = t if best is None else min(best, t)
        row.append("%s %.2f" % (pname, best))
    print("%-32s %s  (ms)" % (name, "   ".join(row)), flush=True)
    del k
