#!/usr/bin/env python3
"""Aggregate-phase timing of a few shapes, for experiments/ab.sh (two library builds on one box).  GPU box only."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(11)
MIX = -7046029254386353131
def run(name, n, g, ncol, aggs, skew=False):
    if skew:
        k = torch.where(torch.rand(n, device=d, generator=gen) < 0.8, torch.randint(0, g // 5, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)).to(torch.int32)
        key = (k, None, pa.U32CODE)
    else:
        key = (torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * MIX, None, pa.I64)
    v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
    a, tt = [], []
    for i in range(12):
        ctx.groupby_compute([key], n, v, aggs)
        t = ctx.timings()
        if i >= 3: a.append(t["phase_ms"]["aggregate"]); tt.append(t["total_ms"])
    print("%-12s aggregate best %.3f median %.3f ms   total best %.3f median %.3f ms" % (name, min(a), statistics.median(a), min(tt), statistics.median(tt)), flush=True)
ALL4 = lambda nc: [(c, op) for c in range(nc) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
run("C2", 100_000_000, 1_000_000, 4, ALL4(4))
run("2col-minmax", 100_000_000, 300_000, 2, [(0, pa.MIN), (0, pa.MAX), (1, pa.MIN), (1, pa.MAX)])
run("C3", 100_000_000, 10_000, 2, ALL4(2) + [(0, pa.COUNT)], skew=True)
run("north-star", 100_000_000, 1_000_000, 1, [(0, pa.SUM)])
