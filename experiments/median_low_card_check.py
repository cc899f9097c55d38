#!/usr/bin/env python3
"""Median of a few huge groups at 100 M rows: result against a torch sort of each group, and the time.  GPU box only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
d = "cuda:0"
ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(7)
n = int(os.environ.get("ROWS", 100_000_000))
OP = pa.NUNIQUE if os.environ.get("OP") == "nunique" else pa.MEDIAN
DISTINCT = int(os.environ.get("DISTINCT", "0"))          # > 0: values drawn from that many distinct numbers
for g in (1, 3, 50, 400, 1000, 3000):
    ids = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64)
    k = ids * -7046029254386353131
    v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    if DISTINCT: v = torch.randint(0, DISTINCT, (n,), device=d, generator=gen).to(torch.float64) / 4
    for generic in (0, 1):
        ctx.set_option("median_generic", generic)
        best = 1e9
        for _ in range(2):
            kc, kn, oa = ctx.groupby_agg([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, OP), (0, pa.COUNT)])
            best = min(best, ctx.timings()["total_ms"])
        got = {int(a): (float(m), int(c)) for a, m, c in zip(kc[0].cpu().view(torch.int64).tolist(), oa[0].cpu().tolist(), oa[1].cpu().tolist())}
        bad = 0
        for gid in range(min(g, 5)):
            x = torch.sort(v[ids == gid]).values
            m = x.numel()
            want = float(x[m // 2]) if m % 2 else float((x[m // 2 - 1] + x[m // 2]) / 2)
            if OP == pa.NUNIQUE: want = float(torch.unique(x).numel())
            key = int(torch.tensor(gid, dtype=torch.int64) * -7046029254386353131)
            bad += got[key] != (want, m)
        print(json.dumps({"groups": g, "generic": generic, "ms": round(best, 3), "mismatches": bad}), flush=True)
    ctx.set_option("median_generic", 0)
    del ids, k, v
