#!/usr/bin/env python3
"""C3 (100 M rows, u32 codes, 10 K groups 80/20, 9 aggregates) and C4 shard with option sets from argv.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
n, g = 100_000_000, 10_000
hot = torch.rand(n, device=d, generator=gen) < 0.8
k = torch.where(hot, torch.randint(0, g // 5, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)).to(torch.int32)
del hot
v = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(2)]
aggs = [(c, op) for c in range(2) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)] + [(0, pa.COUNT)]
def best(fn, reps=4):
    b = None
    for _ in range(reps):
        fn(); t = ctx.timings()
        if b is None or t["total_ms"] < b["total_ms"]: b = t
    return b
def show(tag, t):
    print("%-44s total %.3f  P %5d T %5d absorbed %5.1f%%  " % (tag, t["total_ms"], t["n_partitions"], t["table_slots"], 100.0 * t.get("absorbed_rows", 0) / n) +
          "  ".join("%s %.3f" % (p, v) for p, v in t["phase_ms"].items()), flush=True)
c3 = lambda: ctx.groupby_compute([(k, None, pa.U32CODE)], n, [(x, None, pa.F64) for x in v], aggs)
for optset in sys.argv[1:] or [""]:
    opts = [kv.split("=") for kv in optset.split(",") if kv]
    for name, val in opts: ctx.set_option(name, int(val))
    show("C3 [%s]" % optset, best(c3))
    for name, val in opts: ctx.set_option(name, 0)
