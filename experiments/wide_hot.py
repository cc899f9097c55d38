#!/usr/bin/env python3
"""Wide aggregations (8 columns) with hot keys: rounds switch the slicing of oversized partitions off — does a hot key's partition
become one workgroup's job?  50 M rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
n = 50_000_000
MIX = -7046029254386353131
F = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(8)]
I = [torch.randint(-10**9, 10**9, (n,), device=d, generator=gen) for _ in range(4)]
for a in sys.argv[1:]:
    if "=" in a: ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
def run(name, k):
    for vn, vals, aggs in (("8 f64 x sum/min/max", [(F[i], None, pa.F64) for i in range(8)], [(c, op) for c in range(8) for op in (pa.SUM, pa.MIN, pa.MAX)]),
                           ("4 f64 + 4 i64 x sum", [(F[i], None, pa.F64) for i in range(4)] + [(I[i], None, pa.I64) for i in range(4)], [(c, pa.SUM) for c in range(8)])):
        for i in range(3): ng = ctx.groupby_compute([(k, None, pa.I64)], n, vals, aggs)
        t = ctx.timings()
        print("%-40s %-22s %8.2f ms  groups %8d P=%5d T=%5d retries=%3d  %s" % (name, vn, t["total_ms"], ng, t["n_partitions"], t["table_slots"], t["retries"], {a: round(b, 2) for a, b in t["phase_ms"].items() if b > 0.1}), flush=True)
ids = torch.randint(0, 1_000_000, (n,), device=d, generator=gen)
run("uniform 1M", ids * MIX)
run("half the rows on one key + 1M", torch.where(torch.rand(n, device=d, generator=gen) < 0.5, torch.zeros_like(ids), ids) * MIX)
run("10 % on one key + 1M", torch.where(torch.rand(n, device=d, generator=gen) < 0.1, torch.zeros_like(ids), ids) * MIX)
run("80 % on 200 keys + 1M", torch.where(torch.rand(n, device=d, generator=gen) < 0.8, torch.randint(0, 200, (n,), device=d, generator=gen), ids) * MIX)
