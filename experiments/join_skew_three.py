import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(5)
MIX = -7046029254386353131
nl, nr = 100_000_000, 10_000_000
rkey = torch.randperm(nr, device=d, generator=gen) * MIX
rgrp = torch.where(torch.rand(nr, device=d, generator=gen) < 0.6, torch.randint(0, 16, (nr,), device=d, generator=gen), 1000 + torch.arange(nr, device=d))
sel = torch.rand(nl, device=d, generator=gen) < 0.5
lkey = rkey[torch.where(sel, torch.zeros(nl, device=d, dtype=torch.int64), torch.randint(0, nr, (nl,), device=d, generator=gen))]
lval = torch.randn(nl, device=d, generator=gen, dtype=torch.float64)
for _ in range(3):
    ctx.join_groupby_sum((lkey, None, pa.I64), (lval, None, pa.F64), nl, (rkey, None, pa.I64), (rgrp, None, pa.I64), nr)
    t = ctx.timings()
    print("%.2f ms retries %d %s" % (t["total_ms"], t["retries"], {a: round(b, 2) for a, b in t["phase_ms"].items()}), flush=True)
