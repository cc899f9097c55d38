#!/usr/bin/env python3
"""Where does one round of the lean kernel at a large fan-out lose to several rounds of the older kernel at a smaller one?
C2's shape (4 f64 columns x sum / mean / min / max: 12 states, lean T = 1488) over group counts around the `p_target` rule (3072).
GPU box only."""
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
def zipf(g, a):
    u = torch.rand(n, device=d, generator=gen, dtype=torch.float64)
    return (((g ** (1 - a) - 1) * u + 1) ** (1 / (1 - a))).to(torch.int64).clamp_(1, g) * MIX
cases = [("uniform 3M", lambda: uniform(3_000_000)), ("uniform 4M", lambda: uniform(4_000_000)), ("uniform 5M", lambda: uniform(5_000_000)),
         ("uniform 7M", lambda: uniform(7_000_000)), ("uniform 10M", lambda: uniform(10_000_000)), ("zipf 0.8 over 5M", lambda: zipf(5_000_000, 0.8))]
only = sys.argv[1:]
for name, make in cases:
    if only and not any(o in name for o in only): continue
    k = make()
    for pt in [int(x) for x in os.environ.get("PTS", "3072,4608,6144,8192").split(",")]:
        ctx.set_option("p_target", pt)
        for i in range(4): ng = ctx.groupby_compute([(k, None, pa.I64)], n, v, aggs)
        t = ctx.timings()
        print("%-18s p_target %4d: groups %8d total %6.2f ms P=%d T=%d retries=%d  %s" % (name, pt, ng, t["total_ms"], t["n_partitions"], t["table_slots"], t["retries"],
              {a: round(b, 3) for a, b in t["phase_ms"].items() if b > 0.05}), flush=True)
    del k
