"""How much of the aggregate kernel's time is per-workgroup fixed cost (table init, compaction,
finalisation)?  Same 1 M-group key space and aggregates as BASELINE config 2, but few rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa

dev = "cuda:0"
ctx = pa.Context(0)
g = 1_000_000
for n in (2_000_000, 10_000_000, 100_000_000):
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    ids = torch.randint(0, g, (n,), device=dev, generator=gen, dtype=torch.int64)
    keys = ids * (-7046029254386353131)
    vals = [torch.randn(n, device=dev, generator=gen, dtype=torch.float64) for _ in range(4)]
    aggs = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
    for rep in range(4):
        ctx.groupby_compute([(keys, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs)
    t = ctx.timings()
    print(n, "P", t["n_partitions"], "T", t["table_slots"], {k: round(v, 3) for k, v in t["phase_ms"].items()}, flush=True)
