#!/usr/bin/env python3
"""join_indices 50 M x 5 M (inner), 6 calls: for kernel traces.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
ctx = pa.Context(0); d = "cuda:0"
for optset in sys.argv[1:] or [""]:
    for kv in optset.split(","):
        if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
nb, npb = 5_000_000, 50_000_000
rk = torch.randperm(nb * 2, device=d)[:nb].to(torch.int64) * -7046029254386353131
pick = torch.randint(0, nb, (npb,), device=d)
lk = rk[pick]
for it in range(6):
    n = ctx.join_compute((lk, None, pa.I64), npb, (rk, None, pa.I64), nb, pa.INNER) if hasattr(ctx, "join_compute") else ctx.join_indices((lk, None, pa.I64), npb, (rk, None, pa.I64), nb, pa.INNER)
    t = ctx.timings()
    print("%.3f ms" % t["total_ms"], {a: round(b, 3) for a, b in t["phase_ms"].items()}, flush=True)
