#!/usr/bin/env python3
"""Randomised parity sweep aimed at the clustered-rows pass (clustered.hip, groupby.hip::run_clustered): rows sorted by key, in runs,
sorted in blocks, sorted with stray rows; every key dtype with and without null masks; the pass's aggregate profiles over 1-4 columns;
forced chunk lengths (tables that fill up -> the call is handed back), the short-run bar moved.  HIP engine through the C ABI vs the CPU
oracle.  GPU box only.   usage: fuzz_clustered.py [n_cases] [seed]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandrs_amd as pa
from oracle import oracle as O
from tests.helpers import assert_groupby_equal

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
first_case = int(os.environ.get("FUZZ_FIRST", "0"))
ctx = pa.Context(0)
ctx.set_option("no_small", 1)
fails = taken = n_bursts = 0
for case in range(first_case, n_cases):
    rng = np.random.default_rng(seed0 * 100003 + case)
    try:
        n = int(rng.integers(1_050_000, 6_000_000))
        g = int(rng.choice([3, 500, 20_000, 300_000, 2_000_000]))
        kd = int(rng.choice([O.I64, O.I64, O.F64, O.U32CODE]))
        layout = str(rng.choice(["sorted", "runs", "runs", "blocks", "stray", "nearly", "nearly"]))
        run = int(rng.choice([8, 12, 30, 100, 2000]))
        if layout == "runs":
            ids = np.repeat(rng.integers(0, g, (n + run - 1) // run), run)[:n]
        else:
            ids = np.sort(rng.integers(0, g, n))
            if layout == "blocks":                      # sorted inside blocks of ~n/7 rows: every key comes back in every block
                for b in np.array_split(np.arange(n), 7): ids[b] = np.sort(rng.integers(0, g, len(b)))
            elif layout == "nearly":                    # keys local in position: bucket of (i + noise)
                jit = int(rng.choice([5, 50, 3000]))
                ids = (np.clip(np.arange(n) + rng.integers(-jit, jit + 1, n), 0, n - 1) // max(run, 2)) % max(g, 2)
            elif layout == "stray":                     # 1 % of the rows carry a random key
                stray = rng.random(n) < 0.01
                ids[stray] = rng.integers(0, g, int(stray.sum()))
        if kd == O.I64:
            kdata = (ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
            if rng.random() < 0.3: kdata[rng.integers(0, n, 3)[0]:][:rng.integers(1, 500)] = -1          # a run of the table's sentinel bits
        elif kd == O.F64:
            pool = np.concatenate([rng.normal(size=max(g - 4, 1)), [0.0, -0.0, np.nan, np.inf]])
            kdata = pool[ids % len(pool)]
        else:
            kdata = ids.astype(np.uint32)
        p_null = float(rng.choice([0, 0, 0.01]))
        kmask = O.pack_mask(np.sort(rng.random(n)) < p_null) if p_null and rng.random() < 0.5 else (O.pack_mask(rng.random(n) < p_null) if p_null else None)
        keys, kdts = [(kdata, kmask, kd)], [kd]
        if rng.random() < 0.15:
            keys.append(((ids % 3).astype(np.uint32), None, O.U32CODE)); kdts.append(O.U32CODE)
        nv = int(rng.integers(1, 5))
        kind = O.F64 if rng.random() < 0.7 else O.I64
        prof = str(rng.choice(["sum", "all", "minmax", "min", "max"])) if kind == O.F64 else str(rng.choice(["sum", "all"]))
        v_null = float(rng.choice([0, 0, 0.1])) if kind == O.F64 and prof in ("sum", "all") else 0.0
        vals = []
        for _ in range(nv):
            if kind == O.F64:
                x = rng.normal(50, 20, n) if rng.random() < 0.8 else rng.integers(-4, 5, n).astype(np.float64) / 2.0
                if rng.random() < 0.3: x[rng.integers(0, n, 5)] = np.nan
                if rng.random() < 0.3: x[rng.integers(0, n, 5)] = -np.inf
            else:
                x = rng.integers(-10**9, 10**9, n).astype(np.int64)
            vals.append((x, O.pack_mask(rng.random(n) < v_null) if v_null else None, kind))
        ops = {"sum": [O.SUM, O.MEAN], "all": [O.SUM, O.MEAN, O.MIN, O.MAX], "minmax": [O.MIN, O.MAX], "min": [O.MIN], "max": [O.MAX]}[prof]
        aggs = [(c, op) for c in range(nv) for op in (ops if prof != "sum" else ops[:1 + int(rng.random() < 0.5)])]
        if prof in ("sum", "all") and rng.random() < 0.5: aggs.append((0, O.COUNT))
        opts = {"clustered_chunk": int(rng.choice([0, 0, 0, 4096, 65536, 1 << 20])), "clustered_max_runs_pct": int(rng.choice([0, 0, 45])),
                "no_clustered": int(rng.random() < 0.05), "fold_min_multi": 0, "no_burst_kernel": int(rng.random() < 0.15),
                "slice_rows": int(rng.choice([0, 0, 0, 30_000])), "p_max": int(rng.choice([0, 0, 0, 24]))}
        for k, v in opts.items(): ctx.set_option(k, v)
        try:
            got = ctx.groupby_agg(keys, n, vals, aggs)
            t = ctx.timings()
        finally:
            for k in opts: ctx.set_option(k, 0)
        want = O.groupby_agg(keys, n, vals, aggs)
        exact = [i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT) or (kind == O.I64 and op == O.SUM)]
        assert_groupby_equal(got, want, kdts, int_exact_rows=exact, rtol=1e-9)
        took = t["n_partitions"] == -2
        taken += took
        n_bursts += t["n_partitions"] > 0
        print("ok   %3d n=%d g=%d kd=%d %s run=%d nk=%d nv=%d kind=%d %s knull=%g vnull=%g opts=%s groups=%d clustered=%d" %
              (case, n, g, kd, layout, run, len(keys), nv, kind, prof, p_null, v_null, opts, got[0].shape[1], took), flush=True)
    except Exception:
        fails += 1
        print("FAIL %3d" % case, flush=True)
        traceback.print_exc()
print("fuzz_clustered done: %d cases, %d took the clustered-rows pass, %d a radix partition, %d failures" % (n_cases - first_case, taken, n_bursts, fails))
