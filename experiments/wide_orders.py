#!/usr/bin/env python3
"""Wide aggregations (8 columns x sum / mean / min / max) under row orders: random, sorted, short runs, nearly sorted.  50 M rows, 1 M groups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
n = 50_000_000
MIX = -7046029254386353131
V = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(8)]
for a in sys.argv[1:]:
    if "=" in a: ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
def run(name, k):
    for nc in (4, 8):
        aggs = [(c, op) for c in range(nc) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
        for i in range(3): ng = ctx.groupby_compute([(k, None, pa.I64)], n, [(V[j], None, pa.F64) for j in range(nc)], aggs)
        t = ctx.timings()
        print("%-44s %d cols %7.2f ms  groups %8d P=%5d T=%5d retries=%3d  %s" % (name, nc, t["total_ms"], ng, t["n_partitions"], t["table_slots"], t["retries"], {a: round(b, 2) for a, b in t["phase_ms"].items() if b > 0.1}), flush=True)
ids = torch.randint(0, 1_000_000, (n,), device=d, generator=gen)
run("random", ids * MIX)
run("sorted", torch.sort(ids)[0] * MIX)
run("runs of 4", torch.repeat_interleave(torch.randint(0, 1_000_000, (n // 4 + 1,), device=d, generator=gen), 4)[:n] * MIX)
i = torch.arange(n, device=d)
run("50 rows per key within +-50 rows", ((i + torch.randint(-50, 51, (n,), device=d, generator=gen)).clamp_(0, n - 1) // 50) * MIX)
