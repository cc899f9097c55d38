#!/usr/bin/env python3
"""join_indices over shapes the headline case (unique build keys, every probe row matches) never sees: looking for cliffs.  50 M probe rows.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(9)
nl, nr = 50_000_000, 5_000_000
M = -7046029254386353131
def bits(n, p):
    n8 = (n + 7) // 8 * 8
    m = (torch.rand(n8, device=d, generator=gen) < p).view(-1, 8).to(torch.uint8)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], device=d, dtype=torch.uint8)
    return (m * w).sum(1).to(torch.uint8)
def run(name, lk, rk, hows=(pa.INNER, pa.LEFT, pa.OUTER)):
    row = []
    for how in hows:
        best = None
        for _ in range(3):
            n, _sp = ctx.join_indices_compute(lk, nl, rk, nr, how); t = ctx.timings()["total_ms"]
            best = t if best is None else min(best, t)
        row.append("%s %.2f ms (%d pairs)" % (["inner", "left", "right", "outer"][how], best, n))
    print("%-46s %s" % (name, "   ".join(row)), flush=True)
perm = torch.randperm(nr, device=d, generator=gen)
run("unique build keys, every probe row matches", (torch.randint(0, nr, (nl,), device=d, generator=gen) * M, None, pa.I64), (perm * M, None, pa.I64))
run("half the probe rows match", (torch.randint(0, 2 * nr, (nl,), device=d, generator=gen) * M, None, pa.I64), (perm * M, None, pa.I64))
run("no probe row matches", ((torch.randint(0, nr, (nl,), device=d, generator=gen) + nr) * M, None, pa.I64), (perm * M, None, pa.I64))
run("build keys x 4 (200 M pairs)", (torch.randint(0, nr // 4, (nl,), device=d, generator=gen) * M, None, pa.I64), ((perm % (nr // 4)) * M, None, pa.I64), hows=(pa.INNER,))
run("5 % null probe keys, 5 % null build keys", (torch.randint(0, nr, (nl,), device=d, generator=gen) * M, bits(nl, 0.05), pa.I64), (perm * M, bits(nr, 0.05), pa.I64))
hotp = torch.where(torch.rand(nl, device=d, generator=gen) < 0.3, torch.zeros(nl, dtype=torch.int64, device=d), torch.randint(0, nr, (nl,), device=d, generator=gen))
run("30 % of the probe rows on one key", (hotp * M, None, pa.I64), (perm * M, None, pa.I64))
run("u32 code keys", (torch.randint(0, nr, (nl,), device=d, generator=gen).to(torch.int32), None, pa.U32CODE), (perm.to(torch.int32), None, pa.U32CODE))
run("sorted probe keys", (torch.sort(torch.randint(0, nr, (nl,), device=d, generator=gen))[0] * M, None, pa.I64), (perm * M, None, pa.I64))
