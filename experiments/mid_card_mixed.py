#!/usr/bin/env python3
"""C3's shape with columns of mixed kinds (f64 + masked i64, sum/mean/min/max + count; 6 K and 10 K groups, 80/20 skew): the mid-cardinality
plan of the older kernel (16-64 sliced partitions) against the lean kernel's rounds grouped by profile.  100 M rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(2)
n = 100_000_000
f = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
i64 = torch.randint(-10**6, 10**6, (n,), device=d, generator=gen)
n8 = (n + 7) // 8
mask = torch.randint(0, 256, (n8,), device=d, generator=gen, dtype=torch.int32).to(torch.uint8) & torch.randint(0, 256, (n8,), device=d, generator=gen, dtype=torch.int32).to(torch.uint8) & 0x11
aggs = [(c, op) for c in range(2) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)] + [(0, pa.COUNT)]
for g in (6_000, 10_000, 100_000):
    hot = torch.rand(n, device=d, generator=gen) < 0.8
    ids = torch.where(hot, torch.randint(0, g // 5, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)).to(torch.int32)
    for npr in (1, 0):
        ctx.set_option("no_profile_rounds", npr)
        for i in range(3): ng = ctx.groupby_compute([(ids, None, pa.U32CODE)], n, [(f, None, pa.F64), (i64, mask, pa.I64)], aggs)
        t = ctx.timings()
        print("groups %7d no_profile_rounds %d: %6.2f ms P=%d T=%d  %s" % (g, npr, t["total_ms"], t["n_partitions"], t["table_slots"], {a: round(b, 2) for a, b in t["phase_ms"].items() if b > 0.1}), flush=True)
