#!/usr/bin/env python3
"""What the absorb DECISION costs an input it does not help: C3's shape with UNIFORM keys (nothing to absorb), default vs no_absorb = 1.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
n = 100_000_000
v = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(2)]
aggs = [(c, op) for c in range(2) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)] + [(0, pa.COUNT)]
def best(fn, reps=6):
    b = None
    for _ in range(reps):
        fn(); t = ctx.timings()
        if b is None or t["total_ms"] < b["total_ms"]: b = t
    return b
for g in (10_000, 40_000, 150_000):
    k = torch.randint(0, g, (n,), device=d, generator=gen).to(torch.int32)
    for na in (0, 1):
        ctx.set_option("no_absorb", na)
        t = best(lambda: ctx.groupby_compute([(k, None, pa.U32CODE)], n, [(x, None, pa.F64) for x in v], aggs))
        print("uniform g=%6d no_absorb=%d: total %.3f ms absorbed %d  %s" % (g, na, t["total_ms"], t["absorbed_rows"], " ".join("%s %.3f" % kv for kv in t["phase_ms"].items())), flush=True)
    ctx.set_option("no_absorb", 0)
