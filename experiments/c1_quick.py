#!/usr/bin/env python3
"""C1 (1 M rows, 1 K groups, one f64 sum) and a few other small calls: wall time per call and phases.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pandrs_amd as pa
d = "cuda:0"; ctx = pa.Context(0)
gen = torch.Generator(device=d); gen.manual_seed(42)
def run(tag, n, g, opts):
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64)
    v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    for name, val in opts: ctx.set_option(name, int(val))
    f = lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)])
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): f()
    wall = (time.perf_counter() - t0) / 50 * 1e3
    t = ctx.timings()
    print("%-40s wall %.3f ms  device %.3f ms  P %d  " % (tag, wall, t["total_ms"], t["n_partitions"]) + " ".join("%s %.3f" % kv for kv in t["phase_ms"].items()), flush=True)
    for name, val in opts: ctx.set_option(name, 0)
for optset in sys.argv[1:] or [""]:
    opts = [kv.split("=") for kv in optset.split(",") if kv]
    run("C1 1M/1K [%s]" % optset, 1_000_000, 1_000, opts)
    run("100K rows/100 groups [%s]" % optset, 100_000, 100, opts)
    run("2M rows/5K groups [%s]" % optset, 2_000_000, 5_000, opts)
    run("2M rows/12K groups [%s]" % optset, 2_000_000, 12_000, opts)
    run("4M rows/50K groups [%s]" % optset, 4_000_000, 50_000, opts)
