#!/usr/bin/env python3
"""Wide aggregations over columns of MIXED kinds / ops (f64 and i64 columns, some with null masks, sums on some and min / max on others):
no uniform profile.  50 M rows, 1 M groups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
n = 50_000_000
MIX = -7046029254386353131
F = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(4)]
I = [torch.randint(-10**9, 10**9, (n,), device=d, generator=gen) for _ in range(4)]
for a in sys.argv[1:]:
    if "=" in a: ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
k = torch.randint(0, 1_000_000, (n,), device=d, generator=gen) * MIX
def run(name, vals, aggs):
    for i in range(3): ng = ctx.groupby_compute([(k, None, pa.I64)], n, vals, aggs)
    t = ctx.timings()
    print("%-64s %7.2f ms  P=%5d T=%5d retries=%3d  %s" % (name, t["total_ms"], t["n_partitions"], t["table_slots"], t["retries"], {a: round(b, 2) for a, b in t["phase_ms"].items() if b > 0.1}), flush=True)
S3 = (pa.SUM, pa.MIN, pa.MAX)
run("4 f64 x sum/min/max (uniform)", [(F[i], None, pa.F64) for i in range(4)], [(c, op) for c in range(4) for op in S3])
run("2 f64 + 2 i64 x sum", [(F[0], None, pa.F64), (F[1], None, pa.F64), (I[0], None, pa.I64), (I[1], None, pa.I64)], [(c, pa.SUM) for c in range(4)])
run("2 f64 + 2 i64 x sum/min/max", [(F[0], None, pa.F64), (F[1], None, pa.F64), (I[0], None, pa.I64), (I[1], None, pa.I64)], [(c, op) for c in range(4) for op in S3])
run("4 f64 + 4 i64 x sum", [(F[i], None, pa.F64) for i in range(4)] + [(I[i], None, pa.I64) for i in range(4)], [(c, pa.SUM) for c in range(8)])
run("4 f64 + 4 i64 x sum/min/max", [(F[i], None, pa.F64) for i in range(4)] + [(I[i], None, pa.I64) for i in range(4)], [(c, op) for c in range(8) for op in S3])
run("4 f64: sum of two, min/max of the other two", [(F[i], None, pa.F64) for i in range(4)], [(0, pa.SUM), (1, pa.SUM), (2, pa.MIN), (2, pa.MAX), (3, pa.MIN), (3, pa.MAX)])
run("6 f64: sum/mean of three, sum/min/max of three", [(F[i % 4], None, pa.F64) for i in range(6)], [(0, pa.SUM), (1, pa.MEAN), (2, pa.SUM)] + [(c, op) for c in (3, 4, 5) for op in S3])
