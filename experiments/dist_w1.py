#!/usr/bin/env python3
"""World-1 rehearsal of the distributed step through the library's RCCL path (pandrs_hip_dist_groupby_agg): at world 1 a rank
keeps every record it produces, which is also what a rank of an 8-GPU job RECEIVES (8 x 1/8 of every peer's records) — the split,
exchange and merge phases run at world-8-shaped record counts; only the wire is missing (a device copy stands in for it).
  dist_w1.py [c2|c4]       PANDRS_HIP_DIST_TRACE=1 prints every stage.   GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
d = "cuda:0"; ctx = pa.Context(0)
ctx.comm_init(pa.Context.comm_unique_id(), 0, 1)
for kv in os.environ.get("PANDRS_OPTS", "").split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
gen = torch.Generator(device=d); gen.manual_seed(1)
if which == "c4":
    n, g, ncol = 125_000_000, 10_000_000, 1
    aggs = [(0, pa.SUM), (0, pa.COUNT)]
else:
    n, g, ncol = 100_000_000, 1_000_000, 4
    aggs = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(ncol)]
cols = [(v, None, pa.F64) for v in vals]
loc = []
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.groupby_compute([(k, None, pa.I64)], n, cols, aggs)
    torch.cuda.synchronize(); loc.append((time.perf_counter() - t0) * 1e3)
print("%s local (final aggregates, no exchange): wall %.3f ms (best of 6)" % (which, min(loc)), flush=True)
walls = []
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.dist_groupby_compute([(k, None, pa.I64)], n, cols, aggs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    walls.append(dt)
    t = ctx.timings()
    print("step %d wall %.3f ms  merge %s" % (i, dt, {a: round(b, 3) for a, b in t["phase_ms"].items()}), flush=True)
print("%s dist step: wall %.3f ms (best of 8) = local + %.3f ms" % (which, min(walls[2:]), min(walls[2:]) - min(loc)), flush=True)
