#!/usr/bin/env python3
"""pandrs_hip_groupby_indices (group_by's own result: CSR of ascending rows per group) at C2 scale, device only."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
d = "cuda:0"
ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
for n, g in ((100_000_000, 1_000_000), (100_000_000, 1_000), (20_000_000, 10_000_000)):
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * -7046029254386353131
    best = None
    for i in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cells, nulls, off, rows = ctx.groupby_indices([(k, None, pa.I64)], n)      # device in, device out (compute + fetch copies)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        t = ctx.timings()
        del cells, nulls, off, rows
        best = min(best, t["total_ms"]) if best else t["total_ms"]
    print(json.dumps({"rows": n, "groups": g, "ms": round(best, 3), "wall_ms_last": round(dt, 3), "Grows/s": round(n / best / 1e6, 2)}), flush=True)
    del k
