#!/usr/bin/env python3
"""A few C2 steps for rocprofv3 (kernel trace or one --pmc pass).  GPU box only.
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 experiments/prof_c2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(1)
n, g = int(os.environ.get("ROWS", 100_000_000)), int(os.environ.get("GROUPS", 1_000_000))
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(4)]
aggs4 = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
for name, val in [kv.split("=") for kv in os.environ.get("OPTS", "").split(",") if kv]:
    ctx.set_option(name, int(val))
for _ in range(int(os.environ.get("STEPS", 4))):
    ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs4)
print(ctx.timings())
