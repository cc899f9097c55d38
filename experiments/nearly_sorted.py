#!/usr/bin/env python3
"""Nearly sorted keys — event times arriving slightly out of order: key = bucket of (i + noise) — neither random (every tile of the
scatter holds a narrow key range) nor clustered by the estimate's adjacent-pair test (neighbours often differ).  C2's shape.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(12)
n = 100_000_000
MIX = -7046029254386353131
V = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(4)]
A4 = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
only = [a for a in sys.argv[1:] if "=" not in a]
for a in sys.argv[1:]:
    if "=" in a: ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
def run(name, k):
    if only and not any(o in name for o in only): return
    for nv, aggs, an in ((4, A4, "4x4"), (1, [(0, pa.SUM)], "sum")):
        for i in range(3): ng = ctx.groupby_compute([(k, None, pa.I64)], n, [(V[j], None, pa.F64) for j in range(nv)], aggs)
        t = ctx.timings()
        print("%-52s %-4s %7.2f ms  groups %9d est %9d P=%5d retries=%3d  %s" % (name, an, t["total_ms"], ng, t["estimated_groups"], t["n_partitions"], t["retries"],
              {a: round(b, 2) for a, b in t["phase_ms"].items() if b > 0.1}), flush=True)
i = torch.arange(n, device=d)
for rows_per_key, jitter in ((10, 50), (10, 1000), (100, 50), (100, 5000), (3, 20), (1000, 100_000)):
    noise = torch.randint(-jitter, jitter + 1, (n,), device=d, generator=gen)
    k = ((i + noise).clamp_(0, n - 1) // rows_per_key) * MIX
    run("%d rows per key, shuffled within +-%d rows" % (rows_per_key, jitter), k)
    del noise, k
