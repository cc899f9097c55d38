"""Rows sorted (or clustered) by key: a strided sample sees no repeated key, so the sample-only
estimate said 'all distinct' and sent C2-shaped inputs through the two-level path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(3)
n, g = 100_000_000, 1_000_000
ids = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64)
vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(4)]
aggs = [(c, op) for c in range(4) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)]
for name, k in (("shuffled", ids * -7046029254386353131), ("sorted", torch.sort(ids)[0] * -7046029254386353131),
                ("clustered runs of 16", (ids // 16 * 16 + 0).repeat_interleave(1) * -7046029254386353131)):
    for _ in range(3):
        ng = ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64) for v in vals], aggs)
    t = ctx.timings()
    print("%-22s groups %8d  est %10d  P %5d  %.2f ms  %s" % (name, ng, t["estimated_groups"], t["n_partitions"], t["total_ms"],
          {a: round(b, 2) for a, b in t["phase_ms"].items()}), flush=True)
