#!/usr/bin/env python3
"""aggregate2's wave fold (lanes of one wave in ONE slot are reduced on the VALU before one lane updates the table): from how many
lanes does it pay?  Sweeps the `fold_min` option over skewed key distributions at C2's shape.  GPU box only."""
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
def interleaved(g):
    a = torch.sort(torch.randint(0, g, (n // 2,), device=d, generator=gen))[0]; b = torch.sort(torch.randint(0, g, (n // 2,), device=d, generator=gen))[0]
    return torch.stack([a, b], 1).reshape(-1) * MIX
cases = [("uniform 1M", lambda: uniform(1_000_000)), ("two sorted streams interleaved", lambda: interleaved(1_000_000)), ("80% on 2K + 1M", lambda: two_class(0.8, 2_000, 1_000_000)),
         ("50% on 20K + 1M", lambda: two_class(0.5, 20_000, 1_000_000)),
         ("zipf 0.8 over 5M", lambda: zipf(5_000_000, 0.8)), ("zipf 0.6 over 1M", lambda: zipf(1_000_000, 0.6)), ("zipf 1.0 over 1M", lambda: zipf(1_000_000, 0.999))]
folds = [int(x) for x in os.environ.get("FOLDS", "65,40,24,16,10,6").split(",")]
only = sys.argv[1:]
for name, make in cases:
    if only and not any(o in name for o in only): continue
    k = make()
    for fm in folds:
        ctx.set_option("fold_min", fm)
        for i in range(4): ng = ctx.groupby_compute([(k, None, pa.I64)], n, v, aggs)
        t = ctx.timings()
        print("%-20s fold_min %2d: groups %8d total %6.2f ms P=%d retries=%d  %s" % (name, fm, ng, t["total_ms"], t["n_partitions"], t["retries"],
              {a: round(b, 3) for a, b in t["phase_ms"].items() if b > 0.05}), flush=True)
    del k
