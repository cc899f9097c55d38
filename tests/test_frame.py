"""Host mirror of the reference API (pandrs_amd/frame.py): these read like the reference's own
tests (tests/optimized_groupby_test.rs, tests/optimized_join_test.rs, tests/optimized_lazy_test.rs,
examples/optimized_groupby_example.rs) but assert exact values against the oracle / golden data."""
import math

import numpy as np
import pytest

from pandrs_amd.frame import (AggregateOp, BooleanColumn, ColumnNotFound, Float64Column, GLOBAL_STRING_POOL,
                              Int64Column, JoinType, LazyFrame, OptimizedDataFrame, StringColumn,
                              rust_f64_to_string)


def _df_values_keys():
    df = OptimizedDataFrame()
    df.add_column("values", Int64Column([10, 20, 30, 40, 50]))
    df.add_column("keys", StringColumn(["A", "B", "A", "B", "C"]))
    return df


# ------------------------------------------------------------------------------ CPU-side behaviour
def test_columns_and_pool():
    a = StringColumn(["x", "y", "x"])
    b = StringColumn(["y", "x"])
    assert a.data[0] == a.data[2] == b.data[1] and a.data[1] == b.data[0]    # equal string <=> equal code
    assert GLOBAL_STRING_POOL.get(int(a.data[1])) == "y"
    c = Int64Column.with_nulls([1, 2, 3], [False, True, False])
    assert c.get(1) is None and c.get(2) == 3 and c.null_mask.tolist() == [2]
    assert Int64Column([1, 2]).null_mask is None
    bc = BooleanColumn([True, False, True, True, False, False, False, False, True])
    assert bc.data.tolist() == [0b00001101, 1] and bc.get(8) is True and bc.len() == 9
    assert [rust_f64_to_string(v) for v in (1.0, 0.1, -0.0, float("nan"), float("inf"), 1e21, 1.5e-7)] == \
        ["1", "0.1", "-0", "NaN", "inf", "1000000000000000000000", "0.00000015"]


def test_errors_before_any_device_work():
    df = _df_values_keys()
    with pytest.raises(ColumnNotFound):                   # grouping.rs:53-57
        df.group_by(["nope"])
    with pytest.raises(ColumnNotFound):                   # aggregation.rs:770-774
        df.group_by(["keys"]).aggregate([("nope", AggregateOp.Sum, "s")])
    with pytest.raises(ColumnNotFound):                   # tests/optimized_join_test.rs:206-212
        df.inner_join(df, "id", "keys")
    with pytest.raises(ValueError):
        df.add_column("values", Int64Column([1, 2, 3, 4, 5]))
    with pytest.raises(ValueError):
        df.add_column("short", Int64Column([1]))


# ------------------------------------------------------------------------------ device behaviour
@pytest.mark.gpu
def test_lazy_aggregate_multiple(golden):
    """tests/optimized_groupby_test.rs:85-133 (shape) + tests/groupby_test.rs:18-67 (values)."""
    res = LazyFrame.new(_df_values_keys()).aggregate(
        ["keys"], [("values", AggregateOp.Count, "count"), ("values", AggregateOp.Sum, "sum"),
                   ("values", AggregateOp.Mean, "mean"), ("values", AggregateOp.Min, "min"),
                   ("values", AggregateOp.Max, "max")]).execute()
    assert res.column_count() == 6 and res.row_count() == 3
    assert res.column_names == ["keys", "count", "sum", "mean", "min", "max"]
    rows = {k: i for i, k in enumerate(res.column("keys").to_list())}
    exp = golden["groupby"][0]["expect"]
    for k, e in exp.items():
        assert res.column("count").data[rows[k]] == e["count"]
        assert res.column("sum").data[rows[k]] == e["sum"]
        assert res.column("mean").data[rows[k]] == e["mean"]
    assert res.column("min").data[rows["A"]] == 10 and res.column("max").data[rows["B"]] == 40


@pytest.mark.gpu
def test_groupby_example_and_shortcuts(golden):
    """examples/optimized_groupby_example.rs:22-33 data; aliases per operations.rs:438-521."""
    case = golden["groupby"][5]
    df = OptimizedDataFrame()
    df.add_column("values", Int64Column(case["values_i64"]))
    df.add_column("category", StringColumn(case["key_strings"]))
    gb = df.group_by(["category"])
    res = gb.agg([("values", AggregateOp.Count), ("values", AggregateOp.Sum), ("values", AggregateOp.Mean),
                  ("values", AggregateOp.Min), ("values", AggregateOp.Max)])
    assert res.column_names == ["category", "values_count", "values_sum", "values_mean", "values_min", "values_max"]
    rows = {k: i for i, k in enumerate(res.column("category").to_list())}
    for k, e in case["expect"].items():
        for op in ("count", "sum", "mean", "min", "max"):
            assert res.column("values_" + op).data[rows[k]] == pytest.approx(e[op], rel=1e-15)
    assert gb.sum("values").column_names == ["category", "values_sum"]
    # numeric and null keys are stringified exactly like the reference ("NULL", decimal i64)
    df2 = OptimizedDataFrame()
    df2.add_column("k", Int64Column.with_nulls([-5, 7, -5, 0], [False, False, False, True]))
    df2.add_column("v", Float64Column([1.0, 2.0, 3.0, 4.0]))
    r2 = df2.group_by("k").sum("v")
    got = dict(zip(r2.column("k").to_list(), r2.column("v_sum").data.tolist()))
    assert got == {"-5": 4.0, "7": 2.0, "NULL": 4.0}
    from pandrs_amd import OperationFailed
    with pytest.raises(OperationFailed):                  # lazy.rs:377-382
        LazyFrame.new(df2).aggregate(["k"], [("v", AggregateOp.Median, "m")]).execute()


@pytest.mark.gpu
def test_two_key_groupby_like_reference():
    """tests/optimized_groupby_test.rs:138-186: two keys => 3 columns (no multi-index), 4 groups;
    group_by() itself builds a multi-index instead of key columns (aggregation.rs:812-853)."""
    df = OptimizedDataFrame()
    df.add_column("cat1", StringColumn(["A", "A", "B", "B"]))
    df.add_column("cat2", StringColumn(["X", "Y", "X", "Y"]))
    df.add_column("value", Float64Column([1.0, 2.0, 3.0, 4.0]))
    res = df.group_by_with_options(["cat1", "cat2"], False).aggregate([("value", AggregateOp.Sum, "total")])
    assert res.column_count() == 3 and res.row_count() == 4
    got = {(a, b): v for a, b, v in zip(res.column("cat1").to_list(), res.column("cat2").to_list(), res.column("total").data)}
    assert got == {("A", "X"): 1.0, ("A", "Y"): 2.0, ("B", "X"): 3.0, ("B", "Y"): 4.0}
    res = df.group_by(["cat1", "cat2"]).sum("value")
    assert res.column_names == ["value_sum"] and sorted(res.multi_index) == [("A", "X"), ("A", "Y"), ("B", "X"), ("B", "Y")]
    assert res.multi_index_names == ["cat1", "cat2"]


@pytest.mark.gpu
def test_lazy_two_keys_keep_their_key_columns():
    """tests/optimized_groupby_test.rs:138-186 through LazyFrame: `column_count() == 3` for two keys (:184).  The lazy arm builds
    its frame inline and never a multi-index (lazy.rs:390-394) — round 3's mirror routed it through group_by(), which does."""
    df = OptimizedDataFrame()
    df.add_column("category", StringColumn(["A", "A", "B", "B", "A"]))
    df.add_column("group", StringColumn(["X", "Y", "X", "Y", "X"]))
    df.add_column("values", Int64Column([10, 20, 30, 40, 50]))
    res = LazyFrame.new(df).aggregate(["category", "group"], [("values", AggregateOp.Sum, "sum")]).execute()
    assert res.row_count() == 4 and res.column_count() == 3 and res.multi_index is None
    assert res.column_names == ["category", "group", "sum"]
    got = {(a, b): v for a, b, v in zip(res.column("category").to_list(), res.column("group").to_list(), res.column("sum").data)}
    assert got == {("A", "X"): 60.0, ("A", "Y"): 20.0, ("B", "X"): 30.0, ("B", "Y"): 40.0}


@pytest.mark.gpu
def test_lazy_join_then_aggregate_runs_the_fused_operator(monkeypatch):
    """LazyFrame: Join(Inner) immediately followed by Aggregate([g], [(v, Sum, alias)]) (lazy.rs:405-425, then :186) with v from the
    left frame and g from the right one is BASELINE config 5: the mirror answers it with pandrs_hip_join_groupby_sum (no joined
    rows).  Same frame as the oracle's join -> gather -> groupby and as the two arms run one after the other."""
    from oracle import oracle as O
    from pandrs_amd import frame as F
    rng = np.random.default_rng(31)
    n_left, n_right = 300_000, 20_000
    rid = rng.permutation(n_right * 3)[:n_right].astype(np.int64) - 7
    g = rng.integers(-3, 40, n_right).astype(np.int64)
    lid = rng.choice(np.concatenate([rid, np.arange(10**9, 10**9 + 500)]), n_left)      # ~97 % hits
    v = rng.normal(10, 3, n_left)
    left, right = OptimizedDataFrame(), OptimizedDataFrame()
    left.add_column("id", Int64Column(lid)); left.add_column("v", Float64Column.with_nulls(v, rng.random(n_left) < 0.01))
    left.add_column("g", Int64Column(np.arange(n_left)))
    right.add_column("rid", Int64Column(rid)); right.add_column("g", Int64Column(g)); right.add_column("name", StringColumn(["n%d" % (x % 11) for x in g]))
    calls = []
    real = F.Context.join_groupby_sum
    monkeypatch.setattr(F.Context, "join_groupby_sum", lambda self, *a: (calls.append(1), real(self, *a))[1])
    for gname, rcol in (("g_right", "g"), ("name", "name")):                             # `g` clashes with a left column -> "g_right" (join.rs:478-482)
        calls.clear()
        res = LazyFrame.new(left).join(right, "id", "rid", JoinType.Inner).aggregate([gname], [("v", AggregateOp.Sum, "total")]).execute()
        assert calls == [1]                                                              # the fused operator answered
        assert res.column_names == [gname, "total"]
        two = left.inner_join(right, "id", "rid").group_by([gname]).aggregate([("v", AggregateOp.Sum, "total")])
        a = dict(zip(res.column(gname).to_list(), res.column("total").data.tolist()))
        b = dict(zip(two.column(gname).to_list(), two.column("total").data.tolist()))
        assert a.keys() == b.keys() and all(abs(a[k] - b[k]) <= 1e-9 * abs(b[k]) for k in b)
        rc = right.column(rcol)
        kc, kn, sums = O.join_groupby_sum(left.column("id").view(), left.column("v").view(), n_left, right.column("rid").view(), rc.view(), n_right)
        want = dict(zip(F._key_strings(rc.dtype, kc[0], kn[0]), sums[0].tolist()))
        assert a.keys() == want.keys() and all(abs(a[k] - want[k]) <= 1e-9 * abs(want[k]) for k in want)
    # shapes that must NOT fuse: another join type, two aggregations, grouping by a left column, a null in g
    calls.clear()
    LazyFrame.new(left).join(right, "id", "rid", JoinType.Left).aggregate(["name"], [("v", AggregateOp.Sum, "t")]).execute()
    LazyFrame.new(left).join(right, "id", "rid", JoinType.Inner).aggregate(["name"], [("v", AggregateOp.Sum, "t"), ("v", AggregateOp.Count, "n")]).execute()
    LazyFrame.new(left).join(right, "id", "rid", JoinType.Inner).aggregate(["g"], [("v", AggregateOp.Sum, "t")]).execute()     # "g" is the LEFT column
    rn = OptimizedDataFrame()
    rn.add_column("rid", Int64Column(rid)); rn.add_column("w", Int64Column.with_nulls(g, np.arange(n_right) % 1000 == 0))
    res = LazyFrame.new(left).join(rn, "id", "rid", JoinType.Inner).aggregate(["w"], [("v", AggregateOp.Sum, "t")]).execute()
    assert calls == []
    assert "NULL" not in res.column("w").to_list()                                       # a null g is the join's fill value 0 (join.rs:304-307)


@pytest.mark.gpu
def test_join_columns_are_gathered_through_the_retained_pairs():
    """pandrs_hip_join_gather / _join_gather_key (the Rust shim's join seam, patch 0002): every column of the joined frame is one
    gather through the pairs the context retains — resident or host source, host output — equal to the oracle's gathers over the
    fetched pairs (join.rs:286-552: misses and nulls become the fill value; the key column takes the right key where there is no
    left row)."""
    import ctypes as C
    from oracle import oracle as O
    from pandrs_amd import _lib as L
    from pandrs_amd.frame import get_context
    rng = np.random.default_rng(77)
    ctx = get_context()
    n_left, n_right = 50_000, 30_000
    lk = rng.integers(0, 40_000, n_left).astype(np.int64)
    rk = rng.permutation(60_000)[:n_right].astype(np.int64)
    lmask, rmask = O.pack_mask(rng.random(n_left) < 0.02), O.pack_mask(rng.random(n_right) < 0.02)
    lpay = rng.normal(0, 1, n_left)
    lpay_mask = O.pack_mask(rng.random(n_left) < 0.1)
    rcodes = rng.integers(0, 50, n_right).astype(np.uint32)
    rbits = np.packbits(rng.random(n_right) < 0.5, bitorder="little")
    for how in (0, 1, 2, 3):
        li, ri = ctx.join_indices((lk, lmask, L.I64), n_left, (rk, rmask, L.I64), n_right, how)
        li, ri = np.asarray(li.cpu() if hasattr(li, "cpu") else li), np.asarray(ri.cpu() if hasattr(ri, "cpu") else ri)
        n = len(li)
        wl, wr = O.join_indices((lk, lmask, O.I64), n_left, (rk, rmask, O.I64), n_right, how)
        assert np.array_equal(li, wl) and np.array_equal(ri, wr)

        def gather(col, n_src, side, fill, out, key_right=None, n_key_right=0, resident=False):
            keep = []
            cc, sp = ctx._cols([col], keep)
            if resident:
                up = ctx.upload_column(*col)
                keep = [up]
                cc, sp = ctx._cols([tuple(up)], keep)
            if key_right is None:
                st = ctx.lib.pandrs_hip_join_gather(ctx.h, sp, cc, n_src, side, fill, L.MEM_HOST, out.ctypes.data_as(C.c_void_p))
            else:
                kc, _ = ctx._cols([key_right], keep)
                st = ctx.lib.pandrs_hip_join_gather_key(ctx.h, sp, cc, n_src, kc, n_key_right, fill, L.MEM_HOST, out.ctypes.data_as(C.c_void_p))
            assert st == 0, ctx.lib.pandrs_hip_last_error()
            if resident:
                up.release()
            return out

        for resident in (False, True):
            got = gather((lpay, lpay_mask, L.F64), n_left, 0, 0, np.full(n, -1.0), resident=resident)
            assert np.array_equal(got, O.gather(lpay, lpay_mask, li, 0.0, O.F64))
        got = gather((rcodes, rmask, L.U32CODE), n_right, 1, 0xFFFFFFFF, np.zeros(n, np.uint32))
        assert np.array_equal(got, O.gather(rcodes, rmask, ri, 0xFFFFFFFF, O.U32CODE))
        got = gather((rbits, None, L.BOOLBITS), n_right, 1, 0, np.full(n, 7, np.uint8))
        assert np.array_equal(got, O.gather(rbits, None, ri, 0, O.BOOLBITS))
        # the key column: left value (null -> fill), else the right key's value
        got = gather((lk, lmask, L.I64), n_left, 0, 0, np.full(n, -1, np.int64), key_right=(rk, rmask, L.I64), n_key_right=n_right)
        a, b = O.gather(lk, lmask, li, 0, O.I64), O.gather(rk, rmask, ri, 0, O.I64)
        assert np.array_equal(got, np.where(li >= 0, a, b))


@pytest.mark.gpu
def test_joins_like_reference_tests(golden):
    """tests/optimized_join_test.rs:6-232: row / column counts, plus exact contents."""
    left = OptimizedDataFrame()
    left.add_column("id", Int64Column([1, 2, 3, 4]))
    left.add_column("name", StringColumn(["Alice", "Bob", "Charlie", "Dave"]))
    right = OptimizedDataFrame()
    right.add_column("id", Int64Column([1, 2, 5, 6]))
    right.add_column("value", Int64Column([100, 200, 500, 600]))
    exp = golden["join_optimized"]["rows"]
    j = left.inner_join(right, "id", "id")
    assert (j.row_count(), j.column_count()) == (exp["inner"], 3) and j.column_names == ["name", "id", "value"]
    assert j.column("name").to_list() == ["Alice", "Bob"] and j.column("value").data.tolist() == [100, 200]
    j = left.left_join(right, "id", "id")
    assert j.row_count() == exp["left"] and j.column("value").data.tolist() == [100, 200, 0, 0]      # fill 0, not null
    j = left.right_join(right, "id", "id")
    assert j.row_count() == exp["right"] and j.column("id").data.tolist() == [1, 2, 5, 6]
    assert j.column("name").to_list() == ["Alice", "Bob", "", ""]
    j = left.outer_join(right, "id", "id")
    assert j.row_count() == exp["outer"] and j.column("id").data.tolist() == [1, 2, 3, 4, 5, 6]
    # different key names: the key column keeps the LEFT name (optimized_join_test.rs:166-203)
    r2 = OptimizedDataFrame()
    r2.add_column("right_id", Int64Column([1, 2, 5, 6]))
    r2.add_column("name", StringColumn(["a", "b", "c", "d"]))
    j = left.inner_join(r2, "id", "right_id")
    assert j.column_names == ["name", "id", "name_right"]                                           # join.rs:478-482
    # disjoint keys: empty frame WITHOUT the key column (join.rs:227-284)
    r3 = OptimizedDataFrame()
    r3.add_column("id", Int64Column([7, 8]))
    r3.add_column("value", Float64Column([1.0, 2.0]))
    j = left.inner_join(r3, "id", "id")
    assert j.row_count() == 0 and j.column_names == ["name", "value"]
    # key dtype mismatch (join.rs:98-104)
    from pandrs_amd import ColumnTypeMismatch
    r4 = OptimizedDataFrame()
    r4.add_column("id", Float64Column([1.0]))
    with pytest.raises(ColumnTypeMismatch):
        left.inner_join(r4, "id", "id")
    # LazyFrame join dispatch (lazy.rs:405-425)
    j = LazyFrame.new(left).join(right, "id", "id", JoinType.Outer).execute()
    assert j.row_count() == exp["outer"]


@pytest.mark.gpu
def test_string_key_merge_vectors(golden):
    """src/dataframe/pandas_compat/merge.rs:271-411 on the optimized frame (fills are 0.0, not NaN)."""
    c = golden["join_string_key"]
    left = OptimizedDataFrame()
    left.add_column("key", StringColumn(c["left_keys"]))
    left.add_column("value1", Float64Column(c["left_value1"]))
    right = OptimizedDataFrame()
    right.add_column("key", StringColumn(c["right_keys"]))
    right.add_column("value2", Float64Column(c["right_value2"]))
    j = left.outer_join(right, "key", "key")
    assert j.column("key").to_list() == c["outer"]["keys"]
    assert j.column("value1").data.tolist() == [1.0, 2.0, 3.0, 4.0, 0.0]
    assert j.column("value2").data.tolist() == [0.0, 20.0, 30.0, 40.0, 50.0]
    j = left.inner_join(right, "key", "key")
    assert j.column("key").to_list() == c["inner"]["keys"] and j.column("value2").data.tolist() == [20.0, 30.0, 40.0]


@pytest.mark.gpu
def test_whole_column_reductions():
    df = OptimizedDataFrame()
    df.add_column("v", Float64Column(np.arange(1, 1001, dtype=np.float64)))
    assert df.sum("v") == 500500.0 and df.mean("v") == 500.5 and df.min("v") == 1.0 and df.max("v") == 1000.0


@pytest.mark.gpu
def test_concurrent_callers_share_one_frame():
    """tests/concurrency_test.rs:351-398: 4 threads x 5 iterations on one shared frame, 4 groups."""
    import threading
    rng = np.random.default_rng(1)
    df = OptimizedDataFrame()
    df.add_column("k", StringColumn([["a", "b", "c", "d"][i] for i in rng.integers(0, 4, 20_000)]))
    df.add_column("v", Float64Column(rng.normal(size=20_000)))
    out, errs = [], []

    def work():
        try:
            for _ in range(5):
                out.append(df.group_by(["k"]).sum("v").row_count())
        except Exception as e:       # noqa
            errs.append(e)
    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs and out == [4] * 20


@pytest.mark.gpu
def test_empty_frames_do_not_fail():
    """tests/edge_cases_test.rs:46-80: joins / groupby on empty frames must not panic."""
    e1 = OptimizedDataFrame()
    e1.add_column("id", Int64Column([]))
    e1.add_column("v", Float64Column([]))
    e2 = OptimizedDataFrame()
    e2.add_column("id", Int64Column([]))
    e2.add_column("w", StringColumn([]))
    for how in ("inner_join", "left_join", "right_join", "outer_join"):
        j = getattr(e1, how)(e2, "id", "id")
        assert j.row_count() == 0 and j.column_names == ["v", "w"]
    g = e1.group_by(["id"]).agg([("v", AggregateOp.Sum), ("v", AggregateOp.Count)])
    assert g.row_count() == 0 and g.column_names == ["id", "v_sum", "v_count"]
    full = OptimizedDataFrame()
    full.add_column("id", Int64Column([1, 2]))
    full.add_column("v", Float64Column([1.0, 2.0]))
    j = full.left_join(e2, "id", "id")
    assert j.row_count() == 2 and j.column("w").to_list() == ["", ""] and j.column("id").data.tolist() == [1, 2]


@pytest.mark.gpu
def test_lazy_pipeline_join_then_aggregate():
    """tests/optimized_lazy_test.rs shape: a join followed by an aggregate in one LazyFrame."""
    left = OptimizedDataFrame()
    left.add_column("id", Int64Column([1, 2, 3, 4, 5, 6]))
    left.add_column("amount", Float64Column([10.0, 20.0, 30.0, 40.0, 50.0, 60.0]))
    right = OptimizedDataFrame()
    right.add_column("id", Int64Column([1, 2, 3, 4, 5]))
    right.add_column("dept", StringColumn(["a", "b", "a", "b", "a"]))
    res = (LazyFrame.new(left).join(right, "id", "id", JoinType.Inner)
           .aggregate(["dept"], [("amount", AggregateOp.Sum, "total"), ("amount", AggregateOp.Mean, "avg")]).execute())
    got = {k: (t, a) for k, t, a in zip(res.column("dept").to_list(), res.column("total").data, res.column("avg").data)}
    assert got == {"a": (90.0, 30.0), "b": (60.0, 30.0)}


@pytest.mark.gpu
def test_groups_median_custom_filter_par_groupby():
    """group_by's own result and its closure-based consumers: GroupBy.groups (grouping.rs:62-104),
    median (operations.rs:480), custom aggregation (aggregation.rs:391-497, jit/groupby.rs tests use
    closures over &[f64]), filter (operations.rs:51-74), par_groupby (grouping.rs:124-331)."""
    df = _df_values_keys()
    gb = df.group_by(["keys"])
    assert gb.groups == {("A",): [0, 2], ("B",): [1, 3], ("C",): [4]}       # tests/groupby_test.rs:18-40 sizes 2/2/1
    med = gb.median("values")
    assert dict(zip(med.column("keys").to_list(), med.column("values_median").data.tolist())) == {"A": 20.0, "B": 30.0, "C": 50.0}
    rng_ = gb.custom("values", "range", lambda v: max(v) - min(v))
    assert dict(zip(rng_.column("keys").to_list(), rng_.column("range").data.tolist())) == {"A": 20.0, "B": 20.0, "C": 0.0}
    big = gb.filter(lambda g: g.row_count() >= 2)
    assert big.row_count() == 4 and sorted(big.column("values").data.tolist()) == [10, 20, 30, 40]
    assert set(big.column("keys").to_list()) == {"A", "B"}
    # par_groupby: parts joined with "_", a null part is "NA"; sub-frames hold every column, nulls filled
    df2 = OptimizedDataFrame()
    df2.add_column("k1", StringColumn(["x", "y", "x", "y", "x"]))
    df2.add_column("k2", Int64Column.with_nulls([1, 2, 1, 0, 3], [False, False, False, True, False]))
    df2.add_column("v", Float64Column.with_nulls([1.0, 2.0, 3.0, 4.0, 5.0], [False, False, True, False, False]))
    parts = df2.par_groupby(["k1", "k2"])
    assert set(parts) == {"x_1", "y_2", "y_NA", "x_3"}
    assert parts["x_1"].row_count() == 2 and parts["x_1"].column("v").data.tolist() == [1.0, 0.0]
    assert parts["x_1"].column_names == ["k1", "k2", "v"] and parts["y_NA"].column("k2").data.tolist() == [0]
    with pytest.raises(ColumnNotFound):
        df2.par_groupby(["nope"])
    # transform (operations.rs:132-276): per-group frames concatenated after the first result's schema
    def demean(g):
        out = OptimizedDataFrame()
        vals = g.column("values").data.astype(np.float64)
        out.add_column("keys", g.column("keys"))
        out.add_column("centered", Float64Column(vals - vals.mean()))
        return out
    t = gb.transform(demean)
    assert t.column_names == ["keys", "centered"] and t.row_count() == 5
    got = sorted(zip(t.column("keys").to_list(), t.column("centered").data.tolist()))
    assert got == [("A", -10.0), ("A", 10.0), ("B", -10.0), ("B", 10.0), ("C", 0.0)]


# ------------------------------------------------------------------------------ JIT extension (G7)
def test_ready_made_aggregations_known_answers():
    """src/optimized/jit/groupby.rs:455-520: the reference's own known answers for its aggregation closures."""
    from pandrs_amd import aggregations as A
    assert abs(A.weighted_mean([1.0, 2.0, 3.0, 4.0, 5.0]) - 55.0 / 15.0) < 1e-10          # test_jit_aggregation
    assert abs(A.geometric_mean([1.0, 2.0, 4.0, 8.0]) - 64.0 ** 0.25) < 1e-10              # test_geometric_mean
    assert abs(A.harmonic_mean([1.0, 2.0, 4.0]) - 3.0 / 1.75) < 1e-10                      # test_harmonic_mean
    assert A.value_range([1.0, 5.0, 3.0, 9.0, 2.0]) == 8.0                                 # test_range
    assert abs(A.coefficient_of_variation([10.0, 12.0, 14.0, 16.0, 18.0]) - 0.22587697572631278) < 1e-10
    # the closures' guards: empty input, no admissible value, single value, zero mean
    for f in (A.weighted_mean, A.geometric_mean, A.harmonic_mean, A.value_range, A.coefficient_of_variation,
              A.kahan_sum, A.kahan_mean, A.kahan_std, A.population_std):
        assert f([]) == 0.0
    assert A.geometric_mean([-1.0, 0.0]) == 0.0 and A.harmonic_mean([0.0, 0.0]) == 0.0
    assert A.coefficient_of_variation([3.0]) == 0.0 and A.coefficient_of_variation([-1.0, 1.0]) == 0.0
    tiny = [1.0] + [1e-16] * 10                              # each addend is below half an ulp of the running sum
    assert sum(tiny) == 1.0 and A.kahan_sum(tiny) == math.fsum(tiny) > 1.0
    assert abs(A.population_std([2.0, 4.0, 4.0, 4.0, 5.0, 5.0, 7.0, 9.0]) - 2.0) < 1e-12


@pytest.mark.gpu
def test_groupby_jit_extension_matches_its_closures():
    """GroupByJitExt (jit/groupby.rs:68-290): sum/mean/std/min/max_jit and the parallel_* variants come from the
    device aggregates; each must equal the reference's closure run over the group's non-null values."""
    from pandrs_amd import aggregations as A
    rng = np.random.default_rng(5)
    n = 20_000
    df = OptimizedDataFrame()
    df.add_column("k", Int64Column(rng.integers(0, 300, n)))
    vnull = rng.random(n) < 0.2
    vnull[:] |= (df.column("k").data == 7)                  # group 7: no non-null value at all
    df.add_column("v", Float64Column.with_nulls(rng.normal(100, 10, n), vnull))
    df.add_column("i", Int64Column(rng.integers(-1000, 1000, n)))
    gb = df.group_by(["k"])
    fmin = lambda v: min(v) if v else float("inf")
    fmax = lambda v: max(v) if v else float("-inf")
    cases = [("sum_jit", A.kahan_sum), ("mean_jit", A.kahan_mean), ("std_jit", A.kahan_std), ("min_jit", fmin),
             ("max_jit", fmax), ("parallel_sum_jit", A.kahan_sum), ("parallel_mean_jit", A.kahan_mean),
             ("parallel_std_jit", A.population_std)]
    for col in ("v", "i"):
        for name, closure in cases:
            got = getattr(gb, name)(col, "r")
            want = gb.aggregate_jit(col, closure, "r")
            g = dict(zip(got.column("k").to_list(), got.column("r").data.tolist()))
            w = dict(zip(want.column("k").to_list(), want.column("r").data.tolist()))
            assert g.keys() == w.keys() and len(g) == 300
            for key in w:
                assert g[key] == pytest.approx(w[key], rel=1e-9, abs=1e-9), (name, col, key)   # 1e-9: the f64 tolerance
    assert gb.min_jit("v", "r").column("r").data[gb.min_jit("v", "r").column("k").to_list().index("7")] == float("inf")
    # a ready-made closure through the custom path, against numpy on the same groups
    cv = gb.aggregate_jit("i", A.coefficient_of_variation, "cv")
    kcol = df.column("k").data
    for key, val in zip(cv.column("k").to_list()[:20], cv.column("cv").data[:20]):
        x = df.column("i").data[kcol == int(key)].astype(np.float64)
        assert val == pytest.approx(x.std(ddof=1) / abs(x.mean()), rel=1e-9)
