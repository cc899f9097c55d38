#!/usr/bin/env python3
"""Where the nested merge of partial records (sliced partitions, direct path) spends its time.
Run under `rocprofv3 --kernel-trace --stats`; prints the engine's own totals.  GPU box only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
d = "cuda:0"
ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
n, g = 100_000_000, 10_000
hot = torch.rand(n, device=d, generator=gen) < 0.8
k = torch.where(hot, torch.randint(0, g // 5, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)).to(torch.int32)
v = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(2)]
aggs = [(c, op) for c in range(2) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)] + [(0, pa.COUNT)]
mode = os.environ.get("MODE", "sliced")
if mode == "sliced":
    ctx.set_option("partitions", 32); ctx.set_option("slice_rows", n // 512)
for i in range(6):
    ctx.groupby_compute([(k, None, pa.U32CODE)], n, [(x, None, pa.F64) for x in v], aggs)
    t = ctx.timings()
    print(json.dumps({"mode": mode, "ms": round(t["total_ms"], 3), "phases": {a: round(b, 3) for a, b in t["phase_ms"].items()}}), flush=True)
