#!/usr/bin/env python3
"""C2 shape: one aggregate round (P ~ 1152 partitions) vs several rounds over fewer, larger partitions
(src_per_round: value columns whose states share one LDS table pass).  GPU box only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
d = "cuda:0"
ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
n, g = 100_000_000, 1_000_000
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
v = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(4)]
aggs = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
for spr in (0, 2, 1):
    ctx.set_option("src_per_round", spr)
    best = None
    for _ in range(4):
        ctx.groupby_compute([(k, None, pa.I64)], n, [(x, None, pa.F64) for x in v], aggs)
        t = ctx.timings()
        if best is None or t["total_ms"] < best["total_ms"]: best = t
    print(json.dumps({"src_per_round": spr, "ms": round(best["total_ms"], 3), "P": best["n_partitions"], "T": best["table_slots"],
                      **{a: round(b, 3) for a, b in best["phase_ms"].items()}}), flush=True)
