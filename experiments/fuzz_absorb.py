#!/usr/bin/env python3
"""Randomised parity of the hot-key absorb-and-spill pass (absorb.hip) against the oracle: forced on (no_absorb = -1) over
random sizes, key dtypes, skews, null patterns and uniform aggregate profiles.  GPU box only.
usage: fuzz_absorb.py [n_cases] [seed]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandrs_amd as pa
from oracle import oracle as O
from tests.helpers import assert_groupby_equal

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = pa.Context(0)
fails = taken = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 * 7919 + case)
    try:
        n = int(rng.choice([70_000, 300_000, 1_200_000, 3_000_000, 6_000_000]))
        g = int(rng.choice([3_000, 9_000, 40_000, 150_000]))
        hot_keys = int(rng.choice([1, 50, 800, 2_000]))
        share = float(rng.choice([0.0, 0.5, 0.8, 0.97]))
        ids = np.where(rng.random(n) < share, rng.integers(0, hot_keys, n), rng.integers(0, g, n))
        kd = int(rng.choice([O.I64, O.I64, O.F64, O.U32CODE]))
        if kd == O.I64:
            k = (ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
            if rng.random() < 0.4: k[rng.random(n) < 0.01] = -1
        elif kd == O.F64:
            pool = np.concatenate([rng.normal(size=g), [0.0, -0.0, np.nan, np.inf]])
            k = pool[ids % len(pool)]
        else:
            k = ids.astype(np.uint32)
        key = (k, O.pack_mask(rng.random(n) < rng.choice([0, 0, 0.01, 0.2])) if rng.random() < 0.5 else None, kd)
        nv = int(rng.integers(1, 5))
        vkind = O.F64 if rng.random() < 0.6 else O.I64
        masked = rng.random() < 0.4
        vals = []
        for _ in range(nv):
            if vkind == O.F64:
                v = rng.normal(50, 20, n) if rng.random() < 0.7 else rng.integers(-4, 5, n).astype(np.float64) / 2.0
                if rng.random() < 0.3: v[rng.random(n) < 0.001] = np.nan
                if rng.random() < 0.2: v[rng.random(n) < 0.001] = np.inf
            else:
                v = rng.integers(-10**6, 10**6, n).astype(np.int64)
            vals.append((v, O.pack_mask(rng.random(n) < rng.choice([0.0, 0.1, 0.9])) if masked else None, vkind))
        opset = [(O.SUM,), (O.SUM, O.MEAN), (O.SUM, O.MIN, O.MAX), (O.SUM, O.MEAN, O.MIN, O.MAX), (O.MIN, O.MAX)][int(rng.integers(0, 5))]
        if vkind == O.I64 and opset == (O.MIN, O.MAX): opset = (O.SUM, O.MIN, O.MAX)
        aggs = [(c, op) for c in range(nv) for op in opset] + ([(0, O.COUNT)] if rng.random() < 0.5 else [])
        ctx.set_option("no_absorb", -1); ctx.set_option("no_direct", int(rng.random() < 0.5)); ctx.set_option("no_small", 1); ctx.set_option("no_hot_image", int(rng.random() < 0.3))
        try:
            got = ctx.groupby_agg([key], n, vals, aggs)
            t = ctx.timings()
        finally:
            ctx.set_option("no_absorb", 0); ctx.set_option("no_direct", 0); ctx.set_option("no_small", 0); ctx.set_option("no_hot_image", 0)
        want = O.groupby_agg([key], n, vals, aggs)
        exact = [i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT) or (vkind == O.I64 and op == O.SUM)]
        assert_groupby_equal(got, want, [kd], int_exact_rows=exact, rtol=1e-9)
        taken += t["absorbed_rows"] > 0
        print("ok   %3d n=%d g=%d hot=%d share=%.2f kd=%d nv=%d vkind=%d masked=%d ops=%s absorbed=%d P=%d" %
              (case, n, g, hot_keys, share, kd, nv, vkind, masked, opset, t["absorbed_rows"], t["n_partitions"]), flush=True)
    except Exception:
        fails += 1; print("FAIL %3d" % case); traceback.print_exc()
print("fuzz_absorb done: %d cases, %d took the absorb pass, %d failures" % (n_cases, taken, fails))
