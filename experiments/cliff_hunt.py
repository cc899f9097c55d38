#!/usr/bin/env python3
"""Looking for cliffs: one groupby shape per line, far from the headline's uniform random i64 keys — key layouts, key bit patterns,
null shares, aggregate sets the lean kernel has no instantiation for, column counts.  50 M rows each; a line that costs several times
the first one is the thing to look at.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
n = 50_000_000
MIX = -7046029254386353131
V = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(8)]
VI = [torch.randint(-10**9, 10**9, (n,), device=d, generator=gen) for _ in range(2)]
def bits(p):
    n8 = (n + 7) // 8 * 8
    m = (torch.rand(n8, device=d, generator=gen) < p).view(-1, 8).to(torch.uint8)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], device=d, dtype=torch.uint8)
    return (m * w).sum(1).to(torch.uint8)
def ids(g): return torch.randint(0, g, (n,), device=d, generator=gen)
A4 = lambda nc: [(c, op) for c in range(nc) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
only = [a for a in sys.argv[1:] if "=" not in a]
for a in sys.argv[1:]:
    if "=" in a: ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
def run(name, key, vals, aggs):
    if only and not any(o in name for o in only): return
    try:
        for i in range(3): ng = ctx.groupby_compute([key], n, vals, aggs)
        t = ctx.timings()
        print("%-58s %7.2f ms  groups %9d est %9d P=%5d T=%5d retries=%3d  %s" % (name, t["total_ms"], ng, t["estimated_groups"], t["n_partitions"], t["table_slots"], t["retries"],
              {a: round(b, 2) for a, b in t["phase_ms"].items() if b > 0.1}), flush=True)
    except Exception as e:
        print("%-58s FAILED: %s" % (name, e), flush=True)
f = lambda k: [(V[i], None, pa.F64) for i in range(k)]
k1m = ids(1_000_000)
run("baseline: random i64 keys (mixed bits), 1M groups, 4x4 aggs", (k1m * MIX, None, pa.I64), f(4), A4(4))
run("dense keys 0..1M (no mixing)", (k1m, None, pa.I64), f(4), A4(4))
run("keys = multiples of 4096", (k1m * 4096, None, pa.I64), f(4), A4(4))
run("keys = multiples of 2^32", (k1m << 32, None, pa.I64), f(4), A4(4))
run("keys = ns timestamps rounded to seconds (x 10^9)", (k1m * 1_000_000_000 + 1_700_000_000_000_000_000, None, pa.I64), f(4), A4(4))
run("negative keys", (-k1m - 1, None, pa.I64), f(4), A4(4))
run("f64 keys (1M distinct reals)", ((k1m.to(torch.float64) * 0.37 - 1e5), None, pa.F64), f(4), A4(4))
run("u32 code keys, 1M", (k1m.to(torch.int32), None, pa.U32CODE), f(4), A4(4))
run("round-robin keys i % 1M", (torch.arange(n, device=d) % 1_000_000 * MIX, None, pa.I64), f(4), A4(4))
run("round-robin keys i % 1000", (torch.arange(n, device=d) % 1000 * MIX, None, pa.I64), f(4), A4(4))
run("sorted descending, 1M", (torch.sort(k1m, descending=True)[0] * MIX, None, pa.I64), f(4), A4(4))
two = torch.stack([torch.sort(ids(1_000_000)[: n // 2])[0], torch.sort(ids(1_000_000)[: n // 2])[0]], 1).reshape(-1)
run("two sorted streams interleaved row by row", (two * MIX, None, pa.I64), f(4), A4(4))
del two
run("one group", (torch.zeros(n, dtype=torch.int64, device=d), None, pa.I64), f(4), A4(4))
run("two groups", (ids(2) * MIX, None, pa.I64), f(4), A4(4))
run("all keys distinct (50M groups), one sum", (torch.randperm(n, device=d, generator=gen) * MIX, None, pa.I64), f(1), [(0, pa.SUM)])
run("30M groups, 4x4 aggs", (ids(30_000_000) * MIX, None, pa.I64), f(4), A4(4))
run("half the keys NULL", (k1m * MIX, bits(0.5), pa.I64), f(4), A4(4))
run("half of every value column NULL", (k1m * MIX, None, pa.I64), [(V[i], bits(0.5), pa.F64) for i in range(4)], A4(4))
run("8 value columns x 4 aggs", (k1m * MIX, None, pa.I64), f(8), A4(8))
run("count only", (k1m * MIX, None, pa.I64), f(1), [(0, pa.COUNT)])
run("mean only", (k1m * MIX, None, pa.I64), f(1), [(0, pa.MEAN)])
run("sum + max (profile without min)", (k1m * MIX, None, pa.I64), f(2), [(0, pa.SUM), (0, pa.MAX), (1, pa.SUM), (1, pa.MAX)])
run("f64 and i64 columns together (sum of each)", (k1m * MIX, None, pa.I64), [(V[0], None, pa.F64), (VI[0], None, pa.I64)], [(0, pa.SUM), (1, pa.SUM)])
run("i64 sum/min/max x 2", (k1m * MIX, None, pa.I64), [(VI[i], None, pa.I64) for i in range(2)], [(c, op) for c in range(2) for op in (pa.SUM, pa.MIN, pa.MAX)])
run("sum of col 0, min of col 1 (different ops per column)", (k1m * MIX, None, pa.I64), f(2), [(0, pa.SUM), (1, pa.MIN)])
run("std of one column", (k1m * MIX, None, pa.I64), f(1), [(0, pa.STD)])
run("first + last of one column", (k1m * MIX, None, pa.I64), f(1), [(0, pa.FIRST), (0, pa.LAST)])
run("sorted, std of one column", (torch.sort(k1m)[0] * MIX, None, pa.I64), f(1), [(0, pa.STD)])
run("sorted, sum + max", (torch.sort(k1m)[0] * MIX, None, pa.I64), f(2), [(0, pa.SUM), (0, pa.MAX), (1, pa.SUM), (1, pa.MAX)])
run("sorted, count only", (torch.sort(k1m)[0] * MIX, None, pa.I64), f(1), [(0, pa.COUNT)])
run("10K groups, 8 value columns x 4 aggs", (ids(10_000) * MIX, None, pa.I64), f(8), A4(8))
run("100 groups, std + median-free mix (sum, std)", (ids(100) * MIX, None, pa.I64), f(2), [(0, pa.SUM), (1, pa.STD)])
