#!/usr/bin/env python3
"""Randomised parity sweep: HIP engine (through the C ABI) vs the CPU oracle over random shapes,
dtypes, null rates, skews, aggregate sets, join types and engine knobs.  GPU box only.
usage: fuzz_parity.py [n_cases] [seed]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandrs_amd as pa
from oracle import oracle as O
from oracle import oracle_np as ONP
from tests.helpers import assert_groupby_equal

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = pa.Context(0)
n_clustered = 0
OPS_MERGEABLE = [O.SUM, O.MEAN, O.MIN, O.MAX, O.COUNT]
OPS_ALL = OPS_MERGEABLE + [O.STD, O.VAR, O.FIRST, O.LAST, O.MEDIAN, O.MEDIAN, O.NUNIQUE, O.NUNIQUE]

def rand_key(rng, n, dtype, g, skew):
    ids = rng.integers(0, g, n)
    if skew == "hot":
        ids[rng.random(n) < 0.6] = 0
    elif skew == "8020":
        hot = rng.random(n) < 0.8
        ids = np.where(hot, rng.integers(0, max(g // 5, 1), n), ids)
    if dtype == O.I64:
        k = (ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
        if rng.random() < 0.3: k[rng.random(n) < 0.05] = -1          # table sentinel bits
        return k
    if dtype == O.F64:
        pool = np.concatenate([rng.normal(size=max(g - 4, 1)), [0.0, -0.0, np.nan, np.inf]])
        return pool[ids % len(pool)]
    if dtype == O.U32CODE:
        return ids.astype(np.uint32)
    return np.packbits(ids % 2 == 0, bitorder="little")

def mask(rng, n, p):
    return O.pack_mask(rng.random(n) < p) if p > 0 else None

SCALE = int(os.environ.get("FUZZ_SCALE", "1"))
fails = 0
first_case = int(os.environ.get("FUZZ_FIRST", "0"))          # replay: FUZZ_FIRST=781 fuzz_parity.py 782 77001 runs case 781 alone
for case in range(first_case, n_cases):
    rng = np.random.default_rng(seed0 * 100003 + case)
    try:
        if rng.random() < float(os.environ.get("FUZZ_GROUPBY_FRAC", "0.7")):      # ---------------- groupby
            n = int(rng.choice([1, 7, 1000, 70_000, 300_000, 1_200_000, 5_000_000])) * (SCALE if rng.random() < 0.5 else 1)
            kd = int(rng.choice([O.I64, O.I64, O.F64, O.U32CODE, O.BOOLBITS]))
            g = int(rng.choice([1, 3, 50, 2000, 60_000, 900_000]))
            skew = rng.choice(["uniform", "hot", "8020"])
            nk = 1 if rng.random() < 0.8 else 2
            kdata = rand_key(rng, n, kd, g, skew)
            layout = rng.choice(os.environ["FUZZ_LAYOUT"].split(",") if "FUZZ_LAYOUT" in os.environ else ["any", "any", "any", "sorted", "runs"])
            if kd != O.BOOLBITS and n > 1:
                if layout == "sorted":
                    kdata = np.sort(kdata) if kd != O.F64 else kdata[np.argsort(kdata.view(np.uint64), kind="stable")]
                elif layout == "runs":
                    kdata = np.repeat(kdata[: n // 20 + 1], 20)[:n]
            keys = [(kdata, mask(rng, n, rng.choice([0, 0, 0.01, 0.3])), kd)]
            kdts = [kd]
            if nk == 2:
                keys.append((rng.integers(0, 5, n).astype(np.uint32), mask(rng, n, rng.choice([0, 0.05])), O.U32CODE)); kdts.append(O.U32CODE)
                if kd in (O.I64, O.F64) and rng.random() < 0.6: keys[0] = (rng.integers(-100, 100, n).astype(np.int64), keys[0][1], O.I64); kdts[0] = O.I64
            nv = int(rng.integers(1, 4))
            vals = []
            for _ in range(nv):
                r = rng.random()
                if r < 0.15: vals.append((rng.integers(-4, 5, n).astype(np.float64) / 2.0, mask(rng, n, rng.choice([0, 0, 0.1])), O.F64))   # few distinct values, +-0.0
                elif r < 0.25: vals.append((rng.integers(-3, 4, n).astype(np.int64), mask(rng, n, rng.choice([0, 0.2])), O.I64))
                elif r < 0.65: vals.append((rng.normal(50, 20, n), mask(rng, n, rng.choice([0, 0, 0.1])), O.F64))
                else: vals.append((rng.integers(-10**6, 10**6, n).astype(np.int64), mask(rng, n, rng.choice([0, 0.2])), O.I64))
            ops = OPS_ALL if rng.random() < 0.4 else OPS_MERGEABLE
            aggs = [(int(rng.integers(0, nv)), int(rng.choice(ops))) for _ in range(int(rng.integers(1, 9)))]
            if rng.random() < 0.2:         # a WIDE uniform aggregation: 5-12 columns of one kind, the same ops on each (the lean kernel in rounds)
                nv = int(rng.integers(5, 13))
                wkind = O.F64 if rng.random() < 0.7 else O.I64
                wmask = float(rng.choice([0, 0, 0.1])) if wkind == O.F64 else 0.0
                mixed_kinds = rng.random() < 0.3          # f64 and i64 columns side by side: no uniform profile, the older kernel in rounds of 4
                kinds = [(O.I64 if (mixed_kinds and rng.random() < 0.5) else wkind) for _ in range(nv)]
                vals = [((rng.normal(50, 20, n) if kd_ == O.F64 else rng.integers(-10**6, 10**6, n).astype(np.int64)), mask(rng, n, wmask) if (wmask and kd_ == O.F64) else None, kd_) for kd_ in kinds]
                wops = [[O.SUM], [O.SUM, O.MEAN], [O.SUM, O.MIN, O.MAX], [O.MIN, O.MAX], [O.MAX]][int(rng.integers(0, 5))]
                if (wkind == O.I64 or mixed_kinds) and wops in ([O.MIN, O.MAX], [O.MAX]): wops = [O.SUM, O.MIN, O.MAX]
                wops = wops if nv * (len([o for o in wops if o != O.MEAN]) + (1 if wmask else 0)) <= 38 else [O.SUM]
                aggs = [(c, op) for c in range(nv) for op in wops] + ([(0, O.COUNT)] if rng.random() < 0.5 else [])
            opts = {"no_direct": int(rng.random() < 0.3), "slice_rows": int(rng.choice([0, 0, 20_000])),
                    "p_max": int(rng.choice([0, 0, 0, 24])), "generic_aggregate": int(rng.random() < 0.2),
                    "scatter_staged": int(rng.random() < 0.9), "shared_cursors": int(rng.random() < 0.9),
                    "agg_v1": int(rng.random() < 0.15), "exact_partition": int(rng.random() < 0.2), "deterministic": int(rng.random() < 0.15),
                    "no_small": int(rng.random() < 0.4), "no_absorb": int(rng.choice([0, 0, -1, -1, 1])), "no_hot_image": int(rng.random() < 0.3), "scatter_wide": int(rng.choice([0, 1, 1, -1])), "two_pass_min_p": int(rng.choice([0, 0, 96])), "sorted_dictionary": int(rng.random() < 0.3), "wide_slices": int(rng.random() < 0.3), "no_census": int(rng.random() < 0.3),
                    "no_table_order": int(rng.random() < 0.25), "p_target": int(rng.choice([0, 0, 3072, 64])),
                    "no_lean_rounds": int(rng.random() < 0.2), "no_profile_rounds": int(rng.random() < 0.25), "no_burst_kernel": int(rng.random() < 0.2), "no_window_bound": int(rng.random() < 0.1), "no_clustered": int(rng.random() < 0.15), "clustered_chunk": int(rng.choice([0, 0, 0, 4096, 1 << 20])), "clustered_max_runs_pct": int(rng.choice([0, 0, 45]))}
            for k, v in opts.items(): ctx.set_option(k, v)
            try:
                got = ctx.groupby_agg(keys, n, vals, aggs)
                absorbed = ctx.timings()["absorbed_rows"]
                n_clustered += ctx.timings()["n_partitions"] == -2
            finally:
                for k, v in {"no_direct": 0, "slice_rows": 0, "p_max": 0, "generic_aggregate": 0, "scatter_staged": 1, "shared_cursors": 1,
                             "agg_v1": 0, "exact_partition": 0, "deterministic": 0, "no_small": 0, "no_absorb": 0, "no_hot_image": 0, "scatter_wide": 0, "two_pass_min_p": 0, "sorted_dictionary": 0, "wide_slices": 0, "no_census": 0, "no_table_order": 0, "p_target": 0, "no_lean_rounds": 0, "no_profile_rounds": 0, "no_burst_kernel": 0, "no_window_bound": 0, "no_clustered": 0, "clustered_chunk": 0, "clustered_max_runs_pct": 0}.items(): ctx.set_option(k, v)
            want = O.groupby_agg(keys, n, vals, aggs)
            exact = [i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT, O.FIRST, O.LAST, O.MEDIAN, O.NUNIQUE) or (vals[c][2] == O.I64 and op == O.SUM)]
            if opts["deterministic"] and not any(np.isnan(np.asarray(v[0], np.float64)).any() or np.isinf(np.asarray(v[0], np.float64)).any() for v in vals if v[2] == O.F64):
                exact = list(range(len(aggs)))          # ascending-row-order fold: every aggregate bit for bit
            assert_groupby_equal(got, want, kdts, int_exact_rows=exact, rtol=1e-9)
            if rng.random() < 0.3:      # group_by's own result on the same keys: a complete characterisation
                cells, nulls, off, rows = ctx.groupby_indices(keys, n)
                gcount = cells.shape[1]
                assert off[0] == 0 and off[-1] == n and (gcount == 0 or np.all(np.diff(off) > 0))
                np.testing.assert_array_equal(np.sort(rows), np.arange(n))
                gid = np.repeat(np.arange(gcount), np.diff(off))
                inner = np.ones(n, bool); inner[off[:-1]] = False
                assert np.all(np.diff(rows)[inner[1:]] > 0)
                for kk, col in enumerate(keys):
                    nul, cell = ONP.key_cells(col, n)
                    np.testing.assert_array_equal(nulls[kk][gid], nul[rows])
                    np.testing.assert_array_equal(np.where(nulls[kk][gid] == 1, 0, cells[kk][gid]), cell[rows])
                assert gcount == want[0].shape[1]
            desc = "groupby n=%d kd=%d g=%d %s %s nk=%d aggs=%s opts=%s absorbed=%d" % (n, kd, g, skew, layout, nk, aggs, opts, absorbed)
        elif rng.random() < 0.35:   # ---------------- fused join -> groupby-sum (LDS multimap / L2 regions / general)
            nl = int(rng.choice([0, 7, 4000, 300_000, 2_000_000])) * (SCALE if rng.random() < 0.5 else 1); nr = int(rng.choice([0, 5, 3000, 120_000, 700_000]))
            unique = rng.random() < 0.6
            space = max(nr * 3, 8)
            rcells = (rng.permutation(space)[:nr] if unique else rng.integers(0, max(nr // 2, 3), nr)).astype(np.int64) * 7919 - 5
            if nr and rng.random() < 0.3: rcells[rng.integers(0, nr)] = -1                    # the table sentinel's bit pattern
            lcells = (rcells[rng.integers(0, nr, nl)] if nr else rng.integers(0, 9, nl).astype(np.int64)).copy()
            if nl: lcells[rng.random(nl) < 0.15] = 123456789                                 # misses
            gdt = int(rng.choice([O.I64, O.U32CODE])); vdt = int(rng.choice([O.I64, O.F64]))
            rg = rng.integers(0, int(rng.choice([3, 400, 60_000])), nr)
            rg = rg.astype(np.uint32) if gdt == O.U32CODE else rg.astype(np.int64) - 7
            lv = rng.integers(-1000, 1000, nl).astype(np.int64) if vdt == O.I64 else rng.normal(10, 5, nl)
            args = ((lcells, mask(rng, nl, rng.choice([0, 0.02])), O.I64), (lv, mask(rng, nl, rng.choice([0, 0.05])), vdt), nl,
                    (rcells, mask(rng, nr, rng.choice([0, 0.02])), O.I64), (rg, mask(rng, nr, rng.choice([0, 0.02])), gdt), nr)
            l2 = int(rng.choice([-1, -1, 0, 1]))
            tp = int(rng.choice([0, 0, 96]))                          # two-pass partition of both sides from 96 partitions up (default: 6144)
            ctx.set_option("join_no_l2", l2); ctx.set_option("two_pass_min_p", tp)
            try:
                got = ctx.join_groupby_sum(*args)
            finally:
                ctx.set_option("join_no_l2", 0); ctx.set_option("two_pass_min_p", 0)
            want = O.join_groupby_sum(*args)
            assert_groupby_equal(got, want, [gdt], int_exact_rows=[0] if vdt == O.I64 else [], rtol=1e-9)
            desc = "fused join nl=%d nr=%d unique=%d g=%d v=%d l2=%d two_pass_from=%d -> %d groups" % (nl, nr, unique, gdt, vdt, l2, tp, want[0].shape[1])
        else:                       # ---------------- join
            nl = int(rng.choice([0, 5, 3000, 200_000, 1_500_000])) * (SCALE if rng.random() < 0.5 else 1); nr = int(rng.choice([0, 4, 2500, 150_000, 900_000])) * (SCALE if rng.random() < 0.5 else 1)
            kd = int(rng.choice([O.I64, O.I64, O.U32CODE, O.F64]))
            space = int(rng.choice([3, 500, 100_000, 5_000_000]))
            if nl * nr / max(space, 1) > 3e7: space = 100_000     # keep the output (and the oracle's run time) bounded
            lk = (rand_key(rng, nl, kd, space, "uniform"), mask(rng, nl, rng.choice([0, 0.02])), kd)
            rk = (rand_key(rng, nr, kd, space, "uniform"), mask(rng, nr, rng.choice([0, 0.02])), kd)
            how = int(rng.integers(0, 4))
            jg = int(rng.random() < 0.25)
            ctx.set_option("join_generic", jg)
            ctx.set_option("join_one_pass", int(rng.random() < 0.25))
            try:
                gl, gr = ctx.join_indices(lk, nl, rk, nr, how)
            finally:
                ctx.set_option("join_generic", 0)
                ctx.set_option("join_one_pass", 0)
            wl, wr = O.join_indices(lk, nl, rk, nr, how)
            np.testing.assert_array_equal(gl, wl); np.testing.assert_array_equal(gr, wr)
            if rng.random() < 0.5:      # the joined frame's columns through the retained pairs (join.rs:286-552): a payload of either side, the key column
                side = int(rng.integers(0, 2))
                n_src = nr if side else nl
                pay = rng.normal(size=n_src); pmask = mask(rng, n_src, rng.choice([0, 0.1]))
                got_p = ctx.join_gather((pay, pmask, O.F64), n_src, len(gl), side, 0.0)
                np.testing.assert_array_equal(got_p, O.gather(pay, pmask, wr if side else wl, 0.0, O.F64))
                if kd in (O.I64, O.U32CODE):
                    fillv = 0 if kd == O.I64 else 0xFFFFFFFF
                    got_k = ctx.join_gather(lk, nl, len(gl), 0, fillv, key_right=rk, n_right=nr)
                    a_, b_ = O.gather(lk[0], lk[1], wl, fillv, kd), O.gather(rk[0], rk[1], wr, fillv, kd)
                    np.testing.assert_array_equal(got_k, np.where(np.asarray(wl) >= 0, a_, b_))
            desc = "join nl=%d nr=%d kd=%d space=%d how=%d generic=%d -> %d rows" % (nl, nr, kd, space, how, jg, len(gl))
        print("ok   %3d %s" % (case, desc), flush=True)
    except pa.PandrsHipError as e:
        if "does not fit" in str(e) or "more than 64 bits" in str(e) or "2^32-row" in str(e):      # documented limits, reported loudly
            print("skip %3d %s" % (case, str(e)[:100]), flush=True)
        else:
            fails += 1; print("FAIL %3d" % case); traceback.print_exc()
    except Exception:
        fails += 1; print("FAIL %3d" % case); traceback.print_exc()
print("fuzz done: %d cases, %d failures (%d through the clustered-rows pass)" % (n_cases, fails, n_clustered))
sys.exit(1 if fails else 0)
