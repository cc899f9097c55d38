#!/usr/bin/env python3
"""Randomised parity of the hot-key absorb-and-spill pass (absorb.hip) against the oracle: forced on (no_absorb = -1) over
random sizes, key dtypes, skews, null patterns and uniform aggregate profiles.  GPU box only.
usage: fuzz_absorb.py [n_cases] [seed]
FUZZ_COMPACT=1: the shapes of the COMPACT spill (a hot set in front of a long tail: hundreds of thousands to millions of groups), with
a second key column, forced slices, a forced low tail estimate and a tiny radix level drawn at random (the nested runs of round 4)."""
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
for case in range(int(os.environ.get("FUZZ_FIRST", "0")), n_cases):          # replay: FUZZ_FIRST=59 fuzz_absorb.py 60 31 runs case 59 alone
    rng = np.random.default_rng(seed0 * 7919 + case)
    try:
        compact = os.environ.get("FUZZ_COMPACT") == "1"
        n = int(rng.choice([70_000, 300_000, 1_200_000, 3_000_000, 6_000_000] if not compact else [2_000_000, 5_000_000, 9_000_000]))
        g = int(rng.choice([3_000, 9_000, 40_000, 150_000] if not compact else [400_000, 1_000_000, 3_000_000]))
        hot_keys = int(rng.choice([1, 50, 800, 2_000]))
        share = float(rng.choice([0.0, 0.5, 0.8, 0.97] if not compact else [0.7, 0.85, 0.95]))
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
        keys, kds = [key], [kd]
        extra = {}
        if compact:
            if rng.random() < 0.4 and kd != O.F64:      # a composite key: (ids % 1000, ids // 1000) packs into one cell
                keys = [((ids % 1000).astype(np.int64) * 3 - 700, key[1], O.I64), ((ids // 1000).astype(np.uint32), None, O.U32CODE)]
                kds = [O.I64, O.U32CODE]
            extra = {"slice_rows": int(rng.choice([0, 0, 30_000])), "tail_groups_hint": int(rng.choice([0, 0, 20_000])), "p_max": int(rng.choice([0, 0, 24]))}
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
        ctx.set_option("no_absorb", -1); ctx.set_option("no_direct", int(rng.random() < 0.5)); ctx.set_option("no_small", 1); ctx.set_option("no_hot_image", 0 if compact else int(rng.random() < 0.3))
        for name, val in extra.items(): ctx.set_option(name, val)
        try:
            got = ctx.groupby_agg(keys, n, vals, aggs)
            t = ctx.timings()
        finally:
            ctx.set_option("no_absorb", 0); ctx.set_option("no_direct", 0); ctx.set_option("no_small", 0); ctx.set_option("no_hot_image", 0)
            for name in extra: ctx.set_option(name, 0)
        want = O.groupby_agg(keys, n, vals, aggs)
        exact = [i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT) or (vkind == O.I64 and op == O.SUM)]
        assert_groupby_equal(got, want, kds, int_exact_rows=exact, rtol=1e-9)
        taken += t["absorbed_rows"] > 0
        print("ok   %3d n=%d g=%d hot=%d share=%.2f kd=%d nk=%d nv=%d vkind=%d masked=%d ops=%s %s absorbed=%d P=%d retries=%d" %
              (case, n, g, hot_keys, share, kd, len(keys), nv, vkind, masked, opset, extra, t["absorbed_rows"], t["n_partitions"], t["retries"]), flush=True)
    except Exception:
        fails += 1; print("FAIL %3d" % case); traceback.print_exc()
print("fuzz_absorb done: %d cases, %d took the absorb pass, %d failures" % (n_cases, taken, fails))
