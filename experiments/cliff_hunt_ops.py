#!/usr/bin/env python3
"""Cliff hunt, second part: the operators beyond sum / min / max (Median, Nunique, Std, First / Last) and group_by's own result, under
row orders and key distributions the headline never sees.  50 M rows.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(6)
n = 50_000_000
MIX = -7046029254386353131
v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
vfew = torch.randint(0, 7, (n,), device=d, generator=gen).to(torch.float64)
def ids(g): return torch.randint(0, g, (n,), device=d, generator=gen)
only = sys.argv[1:]
def run(name, key, val, aggs):
    if only and not any(o in name for o in only): return
    try:
        for i in range(3): ng = ctx.groupby_compute([key], n, [(val, None, pa.F64)], aggs)
        t = ctx.timings()
        print("%-62s %7.2f ms  groups %9d P=%5d retries=%3d  %s" % (name, t["total_ms"], ng, t["n_partitions"], t["retries"], {a: round(b, 2) for a, b in t["phase_ms"].items() if b > 0.1}), flush=True)
    except Exception as e:
        print("%-62s FAILED: %s" % (name, e), flush=True)
k1m = ids(1_000_000)
layouts = [("random 1M groups", k1m * MIX), ("sorted 1M groups", torch.sort(k1m)[0] * MIX), ("1K groups", ids(1000) * MIX), ("sorted 1K groups", torch.sort(ids(1000))[0] * MIX),
           ("half the rows on one key + 1M", torch.where(torch.rand(n, device=d, generator=gen) < 0.5, torch.zeros(n, dtype=torch.int64, device=d), k1m) * MIX),
           ("20M groups", ids(20_000_000) * MIX)]
for lname, k in layouts:
    for oname, val, aggs in (("median", v, [(0, pa.MEDIAN)]), ("nunique (7 distinct values)", vfew, [(0, pa.NUNIQUE)]), ("nunique (all distinct)", v, [(0, pa.NUNIQUE)]),
                             ("std", v, [(0, pa.STD)]), ("first+last", v, [(0, pa.FIRST), (0, pa.LAST)]), ("sum+median", v, [(0, pa.SUM), (0, pa.MEDIAN)])):
        run("%s: %s" % (lname, oname), (k, None, pa.I64), val, aggs)
for lname, k in layouts:
    if only and not any(o in "group_by indices" for o in only): continue
    try:
        for i in range(2): out = ctx.groupby_indices([(k, None, pa.I64)], n)
        print("%-62s %7.2f ms" % (lname + ": group_by indices (CSR)", ctx.timings()["total_ms"]), flush=True)
        del out
    except Exception as e:
        print("%-62s FAILED: %s" % (lname + ": group_by indices", e), flush=True)
