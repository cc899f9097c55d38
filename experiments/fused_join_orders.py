#!/usr/bin/env python3
"""The fused join -> groupby(sum) at C5's per-GPU shard shape (62.5 M probe rows, 50 M unique build keys, 100 K groups) and at
50 M x 5 M, under row orders the headline case (everything random) never sees: probe side sorted by key, build side sorted by key,
group ids clustered along the build side, both.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(9)
M = -7046029254386353131
def run(name, nl, nr, lk, rk, rg):
    lv = torch.randn(nl, device=d, generator=gen, dtype=torch.float64)
    best = None
    for _ in range(4):
        out = ctx.join_groupby_sum((lk, None, pa.I64), (lv, None, pa.F64), nl, (rk, None, pa.I64), (rg, None, pa.I64), nr)
        t = ctx.timings(); best = t if best is None or t["total_ms"] < best["total_ms"] else best
    print("%-64s %6.2f ms  groups %d  %s" % (name, best["total_ms"], out[0].shape[1], {a: round(b, 2) for a, b in best["phase_ms"].items() if b > 0.05}), flush=True)
    del lv
for nl, nr, G in ((62_500_000, 50_000_000, 100_000), (50_000_000, 5_000_000, 100_000)):
    print("--- %d probe rows x %d build rows, %d groups" % (nl, nr, G), flush=True)
    perm = torch.randperm(nr, device=d, generator=gen)
    probe = torch.randint(0, nr, (nl,), device=d, generator=gen)
    grp = torch.randint(0, G, (nr,), device=d, generator=gen)
    run("random order (headline)", nl, nr, probe * M, perm * M, grp)
    run("probe rows sorted by key", nl, nr, torch.sort(probe)[0] * M, perm * M, grp)
    run("probe rows sorted by the key's mixed bits", nl, nr, torch.sort(probe * M)[0], perm * M, grp)
    run("build rows sorted by key (0, 1, 2, ...)", nl, nr, probe * M, torch.arange(nr, device=d) * M, grp)
    run("build rows sorted, group ids clustered along them", nl, nr, probe * M, torch.arange(nr, device=d) * M, torch.arange(nr, device=d) // (nr // G + 1))
    run("both sides sorted, group ids clustered", nl, nr, torch.sort(probe)[0] * M, torch.arange(nr, device=d) * M, torch.arange(nr, device=d) // (nr // G + 1))
    del perm, probe, grp
