#!/usr/bin/env python3
"""The estimate's second stage (hash-slice census, partition.hip): which key distributions trigger it, what it says, what the call costs
with and without it.  C2's shape (100 M rows x 4 f64 columns, sum / mean / min / max).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
n, ncol = 100_000_000, 4
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
aggs = [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
MIX = -7046029254386353131
def uniform(g): return torch.randint(0, g, (n,), device=d, generator=gen) * MIX
def two_class(hot_share, hot_keys, g):
    return torch.where(torch.rand(n, device=d, generator=gen) < hot_share, torch.randint(0, hot_keys, (n,), device=d, generator=gen),
                       torch.randint(0, g, (n,), device=d, generator=gen)) * MIX
def zipf(g, a):
    u = torch.rand(n, device=d, generator=gen, dtype=torch.float64)
    return (((g ** (1 - a) - 1) * u + 1) ** (1 / (1 - a))).to(torch.int64).clamp_(1, g) * MIX
cases = [("uniform 1M", lambda: uniform(1_000_000)), ("uniform 300K", lambda: uniform(300_000)), ("uniform 10M", lambda: uniform(10_000_000)),
         ("uniform 50K", lambda: uniform(50_000)),
         ("80/20 of 1M (SURVEY 8d)", lambda: two_class(0.8, 200_000, 1_000_000)), ("80% on 2K + 1M", lambda: two_class(0.8, 2_000, 1_000_000)),
         ("50% on 500K + 5M", lambda: two_class(0.5, 500_000, 5_000_000)), ("90% on 100K + 3M", lambda: two_class(0.9, 100_000, 3_000_000)),
         ("zipf 0.8 over 5M", lambda: zipf(5_000_000, 0.8)), ("zipf 1.2 over 5M", lambda: zipf(5_000_000, 1.2))]
only = sys.argv[1:]
for name, make in cases:
    if only and not any(o in name for o in only): continue
    k = make()
    true_g = torch.unique(k).numel()
    for census in (1, 0):
        ctx.set_option("no_census", 1 - census)
        for i in range(4):
            if i == 3 and census: os.environ["PANDRS_HIP_ENGINE_TRACE"] = "1"
            ng = ctx.groupby_compute([(k, None, pa.I64)], n, v, aggs)
            os.environ.pop("PANDRS_HIP_ENGINE_TRACE", None)
        t = ctx.timings()
        print("%-26s census %d: true %8d est %8d  total %.2f ms P=%d retries=%d  %s" % (name, census, true_g, t["estimated_groups"], t["total_ms"], t["n_partitions"], t["retries"],
              {a: round(b, 3) for a, b in t["phase_ms"].items() if b > 0.005}), flush=True)
    del k
