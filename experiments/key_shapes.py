#!/usr/bin/env python3
"""Key dtypes / composite keys / null masks at scale: looking for cliffs.  50 M rows, sum+mean+min+max over 2 f64 columns.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
n = 50_000_000
v = [(torch.randn(n, device=d, generator=gen, dtype=torch.float64), None, pa.F64) for _ in range(2)]
aggs = [(c, op) for c in range(2) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
def bits(p):
    m = (torch.rand(n, device=d, generator=gen) < p).view(-1, 8).to(torch.uint8)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], device=d, dtype=torch.uint8)
    return (m * w).sum(1).to(torch.uint8)
for g in (1_000, 1_000_000):
    ids = torch.randint(0, g, (n,), device=d, generator=gen)
    shapes = {
        "i64": [(ids * -7046029254386353131, None, pa.I64)],
        "i64 + 5 % null keys": [(ids * -7046029254386353131, bits(0.05), pa.I64)],
        "f64": [((ids.to(torch.float64) * 0.37), None, pa.F64)],
        "u32 codes": [(ids.to(torch.int32).view(torch.uint32) if hasattr(torch, "uint32") else ids.to(torch.int32), None, pa.U32CODE)],
        "i64 x u32 (2 keys)": [((ids // 7) * 11, None, pa.I64), ((ids % 7).to(torch.int32), None, pa.U32CODE)],
        "i64 x u32 x f64 (3 keys)": [((ids // 21) * 11, None, pa.I64), ((ids % 7).to(torch.int32), None, pa.U32CODE), (((ids // 7) % 3).to(torch.float64), None, pa.F64)],
    }
    for name, keys in shapes.items():
        best = None
        try:
            for _ in range(3):
                ng = ctx.groupby_compute(keys, n, v, aggs)
                t = ctx.timings(); best = t["total_ms"] if best is None else min(best, t["total_ms"])
            print("%8d groups  %-26s %.2f ms  (%d groups out)" % (g, name, best, ng), flush=True)
        except Exception as e:
            print("%8d groups  %-26s FAILED %s" % (g, name, e), flush=True)
    del shapes, ids
