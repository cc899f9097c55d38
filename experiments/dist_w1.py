#!/usr/bin/env python3
"""World-1 rehearsal of the distributed C2 step through the library's RCCL path, for kernel traces.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
ctx.comm_init(pa.Context.comm_unique_id(), 0, 1)
for kv in os.environ.get("PANDRS_OPTS", "").split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
gen = torch.Generator(device=d); gen.manual_seed(1)
n, g = 100_000_000, 1_000_000
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(4)]
aggs4 = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
import time
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.dist_groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs4)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    t = ctx.timings()
    print("step %d wall %.3f ms  merge %s" % (i, dt, {a: round(b, 3) for a, b in t["phase_ms"].items()}), flush=True)
