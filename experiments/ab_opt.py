#!/usr/bin/env python3
"""A/B of an option inside one process: ab_opt.py SCRIPT_KIND name=value ...  (kinds: indices).  GPU box only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
n, g = 100_000_000, 1_000_000
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
for optset in sys.argv[1:] or [""]:
    opts = [kv.split("=") for kv in optset.split(",") if kv]
    for name, val in opts: ctx.set_option(name, int(val))
    best = None
    for i in range(3):
        out = ctx.groupby_indices([(k, None, pa.I64)], n)
        t = ctx.timings(); del out
        if best is None or t["total_ms"] < best["total_ms"]: best = t
    print("indices 100M/1M [%s] %.3f ms %s" % (optset, best["total_ms"], {a: round(b, 2) for a, b in best["phase_ms"].items()}), flush=True)
    for name, val in opts: ctx.set_option(name, 0)
