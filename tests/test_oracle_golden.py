"""Pins the CPU oracle against the reference's own known-answer tests (tests/golden/) and
cross-checks its three restatements (typed C, faithful string-keyed C, pure-Python twin)."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from oracle import oracle_np as ON
from tests.helpers import assert_groupby_equal, codes_of

OPS = {"sum": O.SUM, "mean": O.MEAN, "min": O.MIN, "max": O.MAX, "count": O.COUNT,
       "std": O.STD, "var": O.VAR, "median": O.MEDIAN, "first": O.FIRST, "last": O.LAST}


def _fl(x):
    if x == "nan":
        return float("nan")
    if x == "inf":
        return float("inf")
    if x == "-inf":
        return float("-inf")
    return float(x)


@pytest.mark.parametrize("faithful", [False, True])
def test_groupby_known_answers(golden, faithful):
    for case in golden["groupby"]:
        if "key_strings" in case:
            codes, pool = codes_of(case["key_strings"])
            key = (codes, None, O.U32CODE)
            name_of = lambda cell: pool[int(cell)]
            pools = [pool]
        else:
            key = (np.array(case["key_i64"], np.int64), None, O.I64)
            name_of = lambda cell: str(int(np.int64(np.uint64(cell))))
            pools = None
        n = len(key[0])
        if "values_i64" in case:
            val = (np.array(case["values_i64"], np.int64), None, O.I64)
        else:
            mask = O.pack_mask(case["value_nulls"]) if "value_nulls" in case else None
            val = (np.array(case["values_f64"], np.float64), mask, O.F64)
        ops = sorted({op for e in case["expect"].values() for op in e})
        aggs = [(0, OPS[o]) for o in ops]
        kc, kn, oa = O.groupby_agg([key], n, [val], aggs, faithful=faithful, pools=pools)
        assert kc.shape[1] == len(case["expect"]), case["cite"]
        for g in range(kc.shape[1]):
            exp = case["expect"][name_of(kc[0, g])]
            for a, o in enumerate(ops):
                if o in exp:
                    assert oa[a, g] == pytest.approx(exp[o], rel=1e-12, abs=1e-3 if o == "std" else 0), \
                        (case["cite"], o)


def test_groupby_two_keys_known_answer(golden):
    c = golden["groupby_two_keys"]
    k1, _ = codes_of(c["key1_strings"])
    k2, _ = codes_of(c["key2_strings"])
    v = (np.array(c["values_f64"]), None, O.F64)
    for faithful in (False, True):
        kc, kn, oa = O.groupby_agg([(k1, None, O.U32CODE), (k2, None, O.U32CODE)], 4, [v],
                                   [(0, O.SUM)], faithful=faithful)
        assert kc.shape[1] == c["expect_n_groups"]


def test_join_known_answers(golden):
    c = golden["join_string_key"]
    both = c["left_keys"] + c["right_keys"]
    codes, pool = codes_of(both)
    lk = (codes[:4], None, O.U32CODE)
    rk = (codes[4:], None, O.U32CODE)
    for how_name, how in (("inner", O.INNER), ("left", O.LEFT), ("right", O.RIGHT), ("outer", O.OUTER)):
        li, ri = O.join_indices(lk, 4, rk, 4, how)
        assert li.tolist() == c[how_name]["left_idx"], how_name
        assert ri.tolist() == c[how_name]["right_idx"], how_name
        # key column of the result: left key, else right key (join.rs:364-472)
        keys = [c["left_keys"][l] if l >= 0 else c["right_keys"][r] for l, r in zip(li, ri)]
        assert keys == c[how_name]["keys"]
        li2, ri2 = ON.join_indices(lk, 4, rk, 4, how)
        assert li2.tolist() == li.tolist() and ri2.tolist() == ri.tolist()
    # value columns for the inner case (merge.rs:320-337)
    li, ri = O.join_indices(lk, 4, rk, 4, O.INNER)
    v1 = O.gather(np.array(c["left_value1"]), None, li, 0.0, O.F64)
    v2 = O.gather(np.array(c["right_value2"]), None, ri, 0.0, O.F64)
    assert v1.tolist() == [2.0, 3.0, 4.0] and v2.tolist() == [20.0, 30.0, 40.0]
    # optimized path fills misses with 0.0, not NaN (join.rs:304-307, :319-322)
    li, ri = O.join_indices(lk, 4, rk, 4, O.LEFT)
    assert O.gather(np.array(c["right_value2"]), None, ri, 0.0, O.F64).tolist() == [0.0, 20.0, 30.0, 40.0]


def test_join_numeric_key_known_answer(golden):
    c = golden["join_numeric_key"]
    lk = (np.array(c["left_keys_f64"]), None, O.F64)
    rk = (np.array(c["right_keys_f64"]), None, O.F64)
    li, ri = O.join_indices(lk, 3, rk, 3, O.INNER)
    assert li.tolist() == c["inner"]["left_idx"] and ri.tolist() == c["inner"]["right_idx"]
    assert [c["left_names"][i] for i in li] == c["inner"]["names"]
    assert O.gather(np.array(c["right_scores"]), None, ri, 0.0, O.F64).tolist() == c["inner"]["scores"]


def test_join_optimized_counts(golden):
    c = golden["join_optimized"]
    lk = (np.array(c["left_ids"], np.int64), None, O.I64)
    rk = (np.array(c["right_ids"], np.int64), None, O.I64)
    for name, how in (("inner", O.INNER), ("left", O.LEFT), ("right", O.RIGHT), ("outer", O.OUTER)):
        li, ri = O.join_indices(lk, 4, rk, 4, how)
        assert len(li) == c["rows"][name]
    d = c["disjoint"]
    li, ri = O.join_indices((np.array(d["left_ids"], np.int64), None, O.I64), 3,
                            (np.array(d["right_ids"], np.int64), None, O.I64), 3, O.INNER)
    assert len(li) == d["inner_rows"]
    with pytest.raises(O.OracleError):        # key dtype mismatch, join.rs:98-104
        O.join_indices(lk, 4, (np.array([1.0]), None, O.F64), 1, O.INNER)


def test_reduction_known_answers(golden):
    for c in golden["reductions"]:
        if "f64_range" in c:
            lo, hi = c["f64_range"]
            col = (np.arange(lo, hi + 1, dtype=np.float64), None, O.F64)
        elif "f64" in c:
            col = (np.array(c["f64"], np.float64), None, O.F64)
        else:
            col = (np.array(c["i64"], np.int64), None, O.I64)
        out, cnt = O.reduce_column(col, len(col[0]))
        for i, name in enumerate(("sum", "mean", "min", "max")):
            if name in c:
                assert out[i] == pytest.approx(_fl(c[name]), abs=1e-10), (c["cite"], name)


# ---------------------------------------------------------------- cross-checks + source-derived quirks

def _random_case(rng, n, g, with_nulls):
    keys = rng.integers(-g // 2, g // 2 + 1, n).astype(np.int64) * 7919
    vf = rng.normal(100, 10, n)
    vi = rng.integers(-1000, 1000, n).astype(np.int64)
    km = vm = None
    if with_nulls:
        km = O.pack_mask(rng.random(n) < 0.05)
        vm = O.pack_mask(rng.random(n) < 0.1)
    return (keys, km, O.I64), [(vf, vm, O.F64), (vi, vm, O.I64)]


ALL_AGGS = [(c, op) for c in (0, 1) for op in
            (O.SUM, O.MEAN, O.MIN, O.MAX, O.COUNT, O.STD, O.VAR, O.MEDIAN, O.FIRST, O.LAST)]


@pytest.mark.parametrize("with_nulls", [False, True])
def test_three_restatements_agree(with_nulls):
    rng = np.random.default_rng(7)
    key, vals = _random_case(rng, 3000, 40, with_nulls)
    typed = O.groupby_agg([key], 3000, vals, ALL_AGGS)
    faithful = O.groupby_agg([key], 3000, vals, ALL_AGGS, faithful=True)
    twin = ON.groupby_agg_arrays([key], 3000, vals, ALL_AGGS)
    exact = range(len(ALL_AGGS))    # same fold order everywhere => bit-identical
    assert_groupby_equal(faithful, typed, [O.I64], int_exact_rows=exact)
    assert_groupby_equal(twin, typed, [O.I64], int_exact_rows=exact)


def test_f64_and_bool_and_multi_keys_agree():
    rng = np.random.default_rng(11)
    n = 500
    kf = rng.choice(np.array([0.0, -0.0, 1.5, float("nan"), float("inf"), -2.25]), n)
    # a second NaN payload must land in the same group ("NaN", grouping.rs:79)
    kf_bits = kf.view(np.uint64).copy()
    kf_bits[np.isnan(kf) & (rng.random(n) < 0.5)] = 0x7FF8000000000123
    kf = kf_bits.view(np.float64)
    kb = np.packbits(rng.random(n) < 0.5, bitorder="little")
    v = (rng.normal(size=n), None, O.F64)
    keys = [(kf, None, O.F64), (kb, O.pack_mask(rng.random(n) < 0.1), O.BOOLBITS)]
    aggs = [(0, O.SUM), (0, O.COUNT), (0, O.MIN)]
    typed = O.groupby_agg(keys, n, [v], aggs)
    faithful = O.groupby_agg(keys, n, [v], aggs, faithful=True)
    twin = ON.groupby_agg_arrays(keys, n, [v], aggs)
    assert typed[0].shape[1] == 6 * 3          # 6 float groups x {false,true,NULL}
    assert_groupby_equal(faithful, typed, [O.F64, O.BOOLBITS], int_exact_rows=range(3))
    assert_groupby_equal(twin, typed, [O.F64, O.BOOLBITS], int_exact_rows=range(3))


def test_source_derived_quirks():
    """No asserting test in the reference; pinned by the cited source lines."""
    key = (np.array([1, 1, 2, 2, 3], np.int64), None, O.I64)
    # all-null group: Mean 0.0 (aggregation.rs:527), Min/Max sentinel => 0.0 (:538, :551),
    # Count still counts null rows (:743), First/Last null => 0.0 (:611, :621)
    vi = (np.array([5, 7, 0, 0, 9], np.int64), O.pack_mask([0, 0, 1, 1, 0]), O.I64)
    vf = (np.array([5.0, 7.0, 0, 0, np.inf]), O.pack_mask([0, 0, 1, 1, 0]), O.F64)
    aggs = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (0, O.MAX), (0, O.COUNT), (0, O.FIRST), (0, O.LAST),
            (1, O.MIN), (1, O.MAX), (1, O.MEAN)]
    kc, kn, oa = O.groupby_agg([key], 5, [vi, vf], aggs)
    assert kc[0].tolist() == [1, 2, 3]
    assert oa[:, 1].tolist() == [0.0, 0.0, 0.0, 0.0, 2.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    # f64 Min of a group whose only value is +inf is reported as 0.0 (min == INFINITY, :656)
    assert oa[7, 2] == 0.0 and oa[8, 2] == np.inf
    # i64 Sum wraps (release build semantics of `sum += val`, :511)
    big = (np.array([2**62, 2**62, 2**62], np.int64), None, O.I64)
    _, _, o2 = O.groupby_agg([(np.zeros(3, np.int64), None, O.I64)], 3, [big], [(0, O.SUM)])
    assert o2[0, 0] == float(np.int64(np.uint64(3 * 2**62 % 2**64)))
    # NaN value propagates through Sum but is ignored by Min/Max (f64::min, :653)
    vn = (np.array([1.0, np.nan, 3.0]), None, O.F64)
    _, _, o3 = O.groupby_agg([(np.zeros(3, np.int64), None, O.I64)], 3, [vn],
                             [(0, O.SUM), (0, O.MIN), (0, O.MAX)])
    assert math.isnan(o3[0, 0]) and o3[1, 0] == 1.0 and o3[2, 0] == 3.0
    # unsupported: numeric op on a string-code / bool column (:748), Custom (:744)
    with pytest.raises(O.OracleError):
        O.groupby_agg([key], 5, [(np.zeros(5, np.uint32), None, O.U32CODE)], [(0, O.SUM)])
    with pytest.raises(O.OracleError):
        O.groupby_agg([key], 5, [vi], [(0, O.CUSTOM)])
    # ... but Count works on any dtype (:743)
    _, _, o4 = O.groupby_agg([key], 5, [(np.zeros(5, np.uint32), None, O.U32CODE)], [(0, O.COUNT)])
    assert o4[0].tolist() == [2.0, 2.0, 1.0]


def test_nunique_source_derived():
    """AggFunc::Nunique of the legacy frame (src/dataframe/groupby.rs:455-467, :514-519): the group's
    parseable values are sorted, `dedup`-ed (==) and counted; no value => 0.0.  No asserting test in the
    reference — pinned by the cited lines (parity unpinned for this op)."""
    key = (np.array([1, 1, 1, 1, 2, 2, 3, 3, 3], np.int64), None, O.I64)
    vf = (np.array([2.5, 2.5, -0.0, 0.0, 7.0, 0.0, np.nan, np.nan, 1.0]), O.pack_mask([0, 0, 0, 0, 1, 1, 0, 0, 0]), O.F64)
    vi = (np.array([4, -4, 4, 9, 1, 1, 5, 5, 5], np.int64), O.pack_mask([0, 0, 0, 1, 0, 0, 0, 0, 0]), O.I64)
    for faithful in (False, True):
        kc, kn, oa = O.groupby_agg([key], 9, [vf, vi], [(0, O.NUNIQUE), (1, O.NUNIQUE), (0, O.COUNT)], faithful=faithful)
        got = {int(k): oa[:, j].tolist() for j, k in enumerate(kc[0].view(np.int64))}
        assert got == {1: [2.0, 2.0, 4.0],      # {2.5, 0.0 (== -0.0)}; {4, -4} with the null row skipped
                       2: [0.0, 1.0, 2.0],      # every value null => 0.0 (:467)
                       3: [3.0, 1.0, 3.0]}      # NaN != NaN: dedup keeps both, plus 1.0


def test_join_quirks_source_derived():
    # null keys never match; null LEFT keys vanish even from left/outer (join.rs:152);
    # null RIGHT keys are unmatched => appended by right/outer (join.rs:211-224);
    # duplicate right keys come out ascending (join.rs:156-158)
    lk = (np.array([5, 7, 5, 9], np.int64), O.pack_mask([0, 1, 0, 0]), O.I64)
    rk = (np.array([5, 5, 8, 9, 5], np.int64), O.pack_mask([0, 0, 0, 1, 0]), O.I64)
    li, ri = O.join_indices(lk, 4, rk, 5, O.OUTER)
    assert li.tolist() == [0, 0, 0, 2, 2, 2, 3, -1, -1]
    assert ri.tolist() == [0, 1, 4, 0, 1, 4, -1, 2, 3]
    li2, ri2 = ON.join_indices(lk, 4, rk, 5, O.OUTER)
    assert li2.tolist() == li.tolist() and ri2.tolist() == ri.tolist()
    # empty inputs do not fail (tests/edge_cases_test.rs:46-80)
    e = (np.zeros(0, np.int64), None, O.I64)
    for how in (O.INNER, O.LEFT, O.RIGHT, O.OUTER):
        li, ri = O.join_indices(e, 0, e, 0, how)
        assert len(li) == 0
    kc, kn, oa = O.groupby_agg([e], 0, [e], [(0, O.SUM)])
    assert kc.shape == (1, 0)


def test_join_random_vs_twin():
    rng = np.random.default_rng(3)
    lk = (rng.integers(0, 50, 400).astype(np.int64), O.pack_mask(rng.random(400) < 0.05), O.I64)
    rk = (rng.integers(0, 60, 300).astype(np.int64), O.pack_mask(rng.random(300) < 0.05), O.I64)
    for how in (O.INNER, O.LEFT, O.RIGHT, O.OUTER):
        a = O.join_indices(lk, 400, rk, 300, how)
        b = ON.join_indices(lk, 400, rk, 300, how)
        assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


def test_join_faithful_string_keyed_shape_gives_the_same_pairs(golden):
    """oracle_join_indices_ref (HashMap<String, Vec<usize>> build + per-row formatted probe, join.rs:107-224) — the shape
    bench.py times as the join's cpu_baseline — pair for pair against the typed restatement and the reference's vectors."""
    rng = np.random.default_rng(11)
    cases = [((rng.integers(0, 50, 400).astype(np.int64), O.pack_mask(rng.random(400) < 0.05), O.I64),
              (rng.integers(0, 60, 300).astype(np.int64), O.pack_mask(rng.random(300) < 0.05), O.I64)),
             ((rng.integers(-3, 4, 500).astype(np.float64) / 2, None, O.F64), (np.array([0.0, -0.0, 0.5, np.nan, -1.5, 0.5]), None, O.F64)),
             ((np.array([np.nan, 0.5, np.nan]), None, O.F64), (np.array([np.nan, 2.0]), None, O.F64)),
             ((rng.integers(0, 9, 200).astype(np.uint32), None, O.U32CODE), (rng.integers(0, 12, 50).astype(np.uint32), O.pack_mask(rng.random(50) < 0.1), O.U32CODE)),
             ((np.zeros(0, np.int64), None, O.I64), (np.arange(4, dtype=np.int64), None, O.I64))]
    for lk, rk in cases:
        for how in (O.INNER, O.LEFT, O.RIGHT, O.OUTER):
            a = O.join_indices(lk, len(lk[0]), rk, len(rk[0]), how)
            b = O.join_indices(lk, len(lk[0]), rk, len(rk[0]), how, faithful=True)
            assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist(), how
    case = golden["join_optimized"]          # tests/optimized_join_test.rs:6-164: ids [1,2,3,4] x [1,2,5,6]
    lk = (np.array(case["left_ids"], np.int64), None, O.I64)
    rk = (np.array(case["right_ids"], np.int64), None, O.I64)
    for how, rows in ((O.INNER, case["rows"]["inner"]), (O.LEFT, case["rows"]["left"]), (O.RIGHT, case["rows"]["right"]), (O.OUTER, case["rows"]["outer"])):
        assert len(O.join_indices(lk, 4, rk, 4, how, faithful=True)[0]) == rows


def test_fused_join_groupby_matches_composition():
    rng = np.random.default_rng(5)
    nb, npb = 200, 2000
    rkeys = rng.permutation(10_000)[:nb].astype(np.int64)
    rg = rng.integers(0, 17, nb).astype(np.int64)
    lkeys = rng.choice(np.concatenate([rkeys, np.array([-1, -2, -3])]), npb).astype(np.int64)
    lv = rng.normal(size=npb)
    kc, kn, oa = O.join_groupby_sum((lkeys, None, O.I64), (lv, None, O.F64), npb,
                                    (rkeys, None, O.I64), (rg, None, O.I64), nb)
    # independent: dictionary lookup + numpy bincount in row order
    lut = dict(zip(rkeys.tolist(), rg.tolist()))
    sums = {}
    for k, v in zip(lkeys.tolist(), lv.tolist()):
        if k in lut:
            sums[lut[k]] = sums.get(lut[k], 0.0) + v
    assert kc[0].astype(np.int64).tolist() == sorted(sums)
    np.testing.assert_allclose(oa[0], [sums[k] for k in sorted(sums)], rtol=1e-12)


def test_typed_multithreaded_baseline_agrees_with_oracle():
    """bench.py's second CPU baseline (typed keys, all cores) computes the same
    aggregates as the oracle on the C2 shape."""
    rng = np.random.default_rng(3)
    n, g = 200_000, 5_000
    keys = (rng.integers(0, g, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    vals = [rng.normal(100, 10, n) for _ in range(2)]
    k, st = O.groupby_typed_mt(keys, vals, 4)
    aggs = [(0, O.COUNT)] + [(c, op) for c in range(2) for op in (O.SUM, O.MIN, O.MAX)]
    wk, wn, wa = O.groupby_agg([(keys, None, O.I64)], n, [(v, None, O.F64) for v in vals], aggs)
    assert len(k) == wk.shape[1] == len(np.unique(keys))
    o1, o2 = np.argsort(k), np.argsort(wk[0])
    np.testing.assert_array_equal(k[o1], wk[0][o2])
    np.testing.assert_allclose(st[o1], wa[:, o2].T, rtol=1e-12)


def test_parallel_faithful_restatement_equals_the_serial_one():
    """bench.py's third CPU figure: par_groupby / par_aggregate shaped restatement on all cores.  Same
    groups, same row lists (ascending after the chunk merge), hence bit-identical aggregates."""
    from tests.helpers import sort_groups
    rng = np.random.default_rng(12)
    n = 120_000
    keys = [((rng.integers(0, 3000, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), O.pack_mask(rng.random(n) < 0.01), O.I64),
            (rng.integers(0, 4, n).astype(np.uint32), None, O.U32CODE)]
    vals = [(rng.normal(0, 1, n), O.pack_mask(rng.random(n) < 0.1), O.F64), (rng.integers(-5, 5, n).astype(np.int64), None, O.I64)]
    aggs = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (1, O.MAX), (1, O.COUNT), (0, O.MEDIAN), (1, O.FIRST), (0, O.LAST), (0, O.STD)]
    a = sort_groups(*O.groupby_agg(keys, n, vals, aggs, faithful=True), [O.I64, O.U32CODE])
    b = sort_groups(*O.groupby_agg(keys, n, vals, aggs, faithful=True, threads=5), [O.I64, O.U32CODE])
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_k1_families_restated(golden):
    """K1's three families (oracle_k1_stats): the reference's own vectors (simd.rs:451-506, parallel.rs:354-372) through
    family (C) and (A), and the source-derived corner cases the reference holds no test for."""
    for c in golden["reductions"]:
        if "f64_range" in c:
            col = (np.arange(c["f64_range"][0], c["f64_range"][1] + 1, dtype=np.float64), None, O.F64)
        elif "f64" in c:
            col = (np.array(c["f64"], np.float64), None, O.F64)
        else:
            col = (np.array(c["i64"], np.int64), None, O.I64)
        k = O.k1_stats(col, len(col[0]))
        f64 = col[2] == O.F64
        got = {"sum": k.c_sum_f64 if f64 else float(k.c_sum_i64), "mean": k.c_mean_f64 if f64 else float(k.c_mean_i64),
               "min": k.c_min_f64 if f64 else float(k.c_min_i64), "max": k.c_max_f64 if f64 else float(k.c_max_i64)}
        for name in ("sum", "mean", "min", "max"):
            if name in c:
                assert got[name] == pytest.approx(float(c[name]), abs=1e-10), (c["cite"], name)
        if "sum" in c and len(col[0]):
            assert k.a_sum == pytest.approx(float(c["sum"]), abs=1e-10)
    # source-derived: simd_mean_i64 truncates toward zero (simd.rs:77-82)
    k = O.k1_stats((np.array([-7, 2, 2], np.int64), None, O.I64), 3)
    assert k.c_mean_i64 == -1 and k.b_mean == -1.0 and k.b_sum_i64 == -3
    k = O.k1_stats((np.array([7, -2, -2, 0], np.int64), None, O.I64), 4)
    assert k.c_mean_i64 == 0 and k.b_mean == 0.75
    # Float64Column::min/max skip non-finite values, None when nothing finite is left (float64_column.rs:147-199);
    # the frame-level fold keeps infinities and drops NaN (aggregate.rs:119-215)
    k = O.k1_stats((np.array([1.0, np.inf, -2.0, np.nan, -np.inf]), None, O.F64), 5)
    assert (k.b_min, k.b_max, k.b_minmax_none) == (-2.0, 1.0, 0) and (k.a_min, k.a_max) == (-np.inf, np.inf)
    k = O.k1_stats((np.array([np.inf, np.nan]), None, O.F64), 2)
    assert k.b_minmax_none == 1 and k.b_mean_none == 0 and k.a_empty == 0 and k.a_max == np.inf
    # all null: frame sum 0.0 and Err(Empty) for the rest; column mean / min / max None
    k = O.k1_stats((np.array([3.0, 4.0]), O.pack_mask([1, 1]), O.F64), 2)
    assert k.a_empty == 1 and k.a_sum == 0.0 and k.b_mean_none == 1 and k.b_minmax_none == 1
    k = O.k1_stats((np.array([], np.int64), None, O.I64), 0)
    assert k.a_empty == 1 and k.b_data_empty == 1 and (k.c_min_i64, k.c_max_i64, k.c_mean_i64) == (2**63 - 1, -2**63, 0)
    # Int64: the frame sums (v as f64), the column wraps in i64
    big = np.array([2**62, 2**62, 2**62], np.int64)
    k = O.k1_stats((big, None, O.I64), 3)
    assert k.a_sum == 3.0 * 2.0**62 and k.b_sum_i64 == np.int64(np.uint64(3 * 2**62 % 2**64).astype(np.int64))


def test_direct_aggregation_known_answers(golden):
    """/root/reference/src/optimized/direct_aggregations.rs:363-593 through the oracle's K1 restatement: the *_direct methods are
    the column folds (family B), the *_simd methods the slice folds (family C); the reference asserts their values on a
    5-element frame (15 / 150, 3 / 30, 5 / 50, 1 / 10), their equality, and a 10 000-element case within 1e-10."""
    c = golden["direct_aggregations"]
    for name, dtype, np_t in (("float_col", O.F64, np.float64), ("int_col", O.I64, np.int64)):
        data = np.array(c[name], np_t)
        k = O.k1_stats((data, None, dtype), len(data))
        e = c["expect"][name]
        f64 = dtype == O.F64
        direct = {"sum": k.b_sum_f64 if f64 else float(k.b_sum_i64), "mean": k.b_mean,
                  "max": k.b_max if f64 else float(k.b_max_i64), "min": k.b_min if f64 else float(k.b_min_i64)}
        simd = {"sum": k.c_sum_f64 if f64 else float(k.c_sum_i64), "mean": k.c_mean_f64 if f64 else float(k.c_mean_i64),
                "max": k.c_max_f64 if f64 else float(k.c_max_i64), "min": k.c_min_f64 if f64 else float(k.c_min_i64)}
        for op in ("sum", "mean", "max", "min"):
            assert direct[op] == e[op], (name, op)                      # assert_eq! in the reference: exact
            assert simd[op] == direct[op], (name, op)                   # :505-545
        assert len(data) == e["count"] and k.b_data_empty == 0
    big = c["large"]
    i = np.arange(1, big["n"] + 1)
    fl, it = i.astype(np.float64) * big["float_scale"], (i * big["int_scale"]).astype(np.int64)
    seq = 0.0
    for x in fl.tolist():
        seq += x                                                        # `iter().sum()`: the sequential f64 sum (:565)
    kf, ki = O.k1_stats((fl, None, O.F64), len(fl)), O.k1_stats((it, None, O.I64), len(it))
    assert abs(kf.c_sum_f64 - seq) < big["abs_tolerance"] and abs(kf.c_mean_f64 - seq / len(fl)) < big["abs_tolerance"]
    assert abs(kf.b_sum_f64 - seq) < big["abs_tolerance"]
    assert float(ki.c_max_i64) == float(ki.b_max_i64) == big["int_max"] and float(ki.c_min_i64) == float(ki.b_min_i64) == big["int_min"]
