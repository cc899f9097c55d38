#!/usr/bin/env python3
"""All BASELINE configs that fit one GPU (shapes of SURVEY.md §8d), per-phase hipEvent times.  GPU box only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pandrs_amd as pa
d = "cuda:0"
ctx = pa.Context(0)
MIX = -7046029254386353131

def report(name, rows, t):
    ph = {k: round(v, 3) for k, v in t["phase_ms"].items()}
    print(json.dumps({"cfg": name, "ms": round(t["total_ms"], 3), "Mrows/s": round(rows / t["total_ms"] / 1e3, 1),
                      "alg_GB/s": round(t["algorithmic_bytes"] / t["total_ms"] / 1e6, 1), "P": t["n_partitions"], "retries": t["retries"], **ph}), flush=True)

def run(name, rows, fn, reps=3):
    best = None
    for _ in range(reps):
        fn(); t = ctx.timings()
        if best is None or t["total_ms"] < best["total_ms"]: best = t
    report(name, rows, best)

gen = torch.Generator(device=d); gen.manual_seed(42)
which = os.environ.get("CFG", "c1,ns,c3,c4,join,fused").split(",")
if "c1" in which:
    n, g = 1_000_000, 1_000
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64); v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    run("C1 1M rows/1K groups/sum", n, lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)]))
if "ns" in which:
    n, g = 100_000_000, 1_000_000
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * MIX; v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    run("north-star 100M/1M groups/sum", n, lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM)]))
    del k, v
if "c3" in which:
    n, g = 100_000_000, 10_000
    hot = torch.rand(n, device=d, generator=gen) < 0.8
    k = torch.where(hot, torch.randint(0, g // 5, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen)).to(torch.int32)
    v = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(2)]
    aggs = [(c, op) for c in range(2) for op in (pa.SUM, pa.MEAN, pa.MIN, pa.MAX)] + [(0, pa.COUNT)]
    run("C3 100M/u32 codes 10K groups 80-20/9 aggs", n, lambda: ctx.groupby_compute([(k, None, pa.U32CODE)], n, [(x, None, pa.F64) for x in v], aggs))
    if "c3u" in which:      # same shape, uniform keys: how much of C3's aggregate time is hot-key LDS contention?
        k2 = torch.randint(0, g, (n,), device=d, generator=gen).to(torch.int32)
        run("C3-uniform 100M/u32 codes 10K groups uniform/9 aggs", n, lambda: ctx.groupby_compute([(k2, None, pa.U32CODE)], n, [(x, None, pa.F64) for x in v], aggs))
        for P in (16, 32, 64, 128, 256, 1024):
            ctx.set_option("partitions", P)
            run("C3 80-20 P=%d" % P, n, lambda: ctx.groupby_compute([(k, None, pa.U32CODE)], n, [(x, None, pa.F64) for x in v], aggs))
        ctx.set_option("partitions", 0)
        # fewer, larger partitions cut into row slices (tasks ~ 512): less scatter fan-out, more distinct
        # groups per LDS table (fewer same-address atomics), at the price of merging the slices' partials
        for P in (8, 16, 32, 64, 128):
            ctx.set_option("partitions", P); ctx.set_option("slice_rows", n // 512)
            run("C3 80-20 P=%d sliced/512" % P, n, lambda: ctx.groupby_compute([(k, None, pa.U32CODE)], n, [(x, None, pa.F64) for x in v], aggs))
        for P in (16, 32, 64):
            ctx.set_option("partitions", P); ctx.set_option("slice_rows", n // 512)
            run("C3-uniform P=%d sliced/512" % P, n, lambda: ctx.groupby_compute([(k2, None, pa.U32CODE)], n, [(x, None, pa.F64) for x in v], aggs))
        ctx.set_option("partitions", 0); ctx.set_option("slice_rows", 0)
        del k2
    del k, v, hot
if "sortops" in which:       # the sort-based aggregates: one (key, value) segmented sort per column
    n, g = 100_000_000, 1_000_000
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * MIX
    v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    w = torch.randint(0, 50, (n,), device=d, generator=gen, dtype=torch.int64)
    run("100M/1M groups/median(f64)", n, lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.MEDIAN)]))
    run("100M/1M groups/nunique(i64, 50 values)", n, lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(w, None, pa.I64)], [(0, pa.NUNIQUE)]))
    run("100M/1M groups/std(f64)", n, lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.STD)]))
    run("100M/1M groups/first+last(f64)", n, lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.FIRST), (0, pa.LAST)]))
    del k, v, w
if "c4" in which:
    n, g = 125_000_000, 10_000_000
    k = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64) * MIX; v = torch.randn(n, device=d, generator=gen, dtype=torch.float64)
    run("C4 shard 125M/10M groups/sum+count", n, lambda: ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM), (0, pa.COUNT)]))
    del k, v
if "join" in which or "fused" in which:
    nb, npb = 5_000_000, 50_000_000
    rk = torch.randperm(nb * 2, device=d, generator=gen)[:nb].to(torch.int64) * MIX
    rg = torch.randint(0, 100_000, (nb,), device=d, generator=gen, dtype=torch.int64)
    lk = rk[torch.randint(0, nb, (npb,), device=d, generator=gen)]
    lv = torch.randn(npb, device=d, generator=gen, dtype=torch.float64)
    if "join" in which:
        run("inner join 50M x 5M (index pairs)", npb, lambda: ctx.join_indices((lk, None, pa.I64), npb, (rk, None, pa.I64), nb, pa.INNER))
    if "fused" in which:
        run("fused join 50M x 5M -> groupby 100K sum", npb, lambda: ctx.join_groupby_sum((lk, None, pa.I64), (lv, None, pa.F64), npb, (rk, None, pa.I64), (rg, None, pa.I64), nb))
