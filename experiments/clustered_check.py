#!/usr/bin/env python3
"""Rows clustered by key (sorted input, input grouped by key, time-ordered keys) at C2's shape: what does the call cost against
random row order?  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
n, ncol = 100_000_000, 4
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(ncol)]
MIX = -7046029254386353131
def ids(g): return torch.randint(0, g, (n,), device=d, generator=gen)
def runs(g, run):        # rows grouped in runs of `run` equal keys, the runs' keys random (a key comes back in ~n / (g * run) runs)
    return torch.repeat_interleave(torch.randint(0, g, ((n + run - 1) // run,), device=d, generator=gen), run)[:n]
cases = [("random 1M", lambda: ids(1_000_000) * MIX),
         ("sorted 1M", lambda: torch.sort(ids(1_000_000))[0] * MIX),
         ("sorted 10K", lambda: torch.sort(ids(10_000))[0] * MIX),
         ("sorted 10M", lambda: torch.sort(ids(10_000_000))[0] * MIX),
         ("sorted mixed key bits 1M", lambda: torch.sort(ids(1_000_000) * MIX)[0]),
         ("runs of 64, 1M keys", lambda: runs(1_000_000, 64) * MIX),
         ("runs of 8, 1M keys", lambda: runs(1_000_000, 8) * MIX),
         ("runs of 4, 1M keys", lambda: runs(1_000_000, 4) * MIX),
         ("runs of 3, 1M keys", lambda: runs(1_000_000, 3) * MIX),
         ("runs of 1000, 100K keys", lambda: runs(100_000, 1000) * MIX)]
agg_sets = {"4x(sum,mean,min,max)": [(c, op) for c in range(ncol) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)], "sum of one": [(0, pa.SUM)]}
only = [a for a in sys.argv[1:] if "=" not in a]
for a in sys.argv[1:]:
    if "=" in a: ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
for name, make in cases:
    if only and not any(o in name for o in only): continue
    k = make()
    for an, aggs in agg_sets.items():
        nv = 4 if len(aggs) > 1 else 1
        for i in range(4): ng = ctx.groupby_compute([(k, None, pa.I64)], n, v[:nv], aggs)
        t = ctx.timings()
        print("%-26s %-22s groups %8d est %8d total %6.2f ms P=%d retries=%d  %s" % (name, an, ng, t["estimated_groups"], t["total_ms"], t["n_partitions"], t["retries"],
              {a: round(b, 3) for a, b in t["phase_ms"].items() if b > 0.05}), flush=True)
    del k
