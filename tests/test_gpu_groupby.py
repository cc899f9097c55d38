"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs, against the reference's golden vectors, and — at BASELINE scale — through
size-independent properties.  Tolerances: keys / counts / min / max bit-exact; f64 sums and means
within 1e-9 relative (BASELINE.json north_star)."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import assert_groupby_equal, codes_of

pytestmark = pytest.mark.gpu

OPS = {"sum": O.SUM, "mean": O.MEAN, "min": O.MIN, "max": O.MAX, "count": O.COUNT}
FIVE = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (0, O.MAX), (0, O.COUNT)]
EXACT5 = (2, 3, 4)      # min, max, count rows of FIVE are bit-exact


@pytest.fixture(scope="module")
def ctx():
    import pandrs_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def sparse_keys(rng, n, g):
    """keys uniform over g groups, bit-mixed to sparse i64."""
    ids = rng.integers(0, g, n).astype(np.uint64)
    return (ids * np.uint64(0x9E3779B97F4A7C15) ^ np.uint64(0x5555AAAA5555AAAA)).view(np.int64)


def sparse_keys_from(ids):
    return (np.asarray(ids).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)


def check(ctx, keys, n, vals, aggs, key_dtypes, exact=(), rtol=1e-9):
    keys = keys if isinstance(keys, list) else [keys]
    got = ctx.groupby_agg(keys, n, vals, aggs)
    want = O.groupby_agg(keys, n, vals, aggs)
    assert_groupby_equal(got, want, key_dtypes, int_exact_rows=exact, rtol=rtol)
    return got


def test_golden_known_answers(ctx, golden):
    for case in golden["groupby"]:
        if "key_strings" in case:
            codes, pool = codes_of(case["key_strings"])
            key = (codes, None, O.U32CODE)
            name_of = lambda cell: pool[int(cell)]
        else:
            key = (np.array(case["key_i64"], np.int64), None, O.I64)
            name_of = lambda cell: str(int(np.int64(np.uint64(cell))))
        n = len(key[0])
        if "values_i64" in case:
            val = (np.array(case["values_i64"], np.int64), None, O.I64)
        else:
            mask = O.pack_mask(case["value_nulls"]) if "value_nulls" in case else None
            val = (np.array(case["values_f64"], np.float64), mask, O.F64)
        ops = sorted({op for e in case["expect"].values() for op in e if op in OPS})
        kc, kn, oa = ctx.groupby_agg([key], n, [val], [(0, OPS[o]) for o in ops])
        assert kc.shape[1] == len(case["expect"]), case["cite"]
        for g in range(kc.shape[1]):
            exp = case["expect"][name_of(kc[0, g])]
            for a, o in enumerate(ops):
                if o in exp:
                    assert oa[a, g] == pytest.approx(exp[o], rel=1e-12), (case["cite"], o)


@pytest.mark.parametrize("n,g", [(1, 1), (5, 3), (1000, 7), (65_537, 1000), (300_000, 50_000),
                                 (1_000_000, 1_000), (2_000_000, 1_000_000)])
def test_i64_key_f64_value(ctx, n, g):
    rng = np.random.default_rng(n + g)
    keys = [(sparse_keys(rng, n, g), None, O.I64)]
    vals = [(rng.normal(100, 10, n), None, O.F64)]
    check(ctx, keys, n, vals, FIVE, [O.I64], exact=EXACT5)


def test_config2_shape_small(ctx):
    """C2's shape (4 f64 columns x sum/mean/min/max) at a size the oracle finishes in seconds."""
    rng = np.random.default_rng(42 + 1)
    n, g = 1_500_000, 40_000
    keys = [(sparse_keys(rng, n, g), None, O.I64)]
    vals = [(rng.normal(100, 10, n), None, O.F64) for _ in range(4)]
    aggs = [(c, op) for c in range(4) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)]
    exact = [i for i, (_, op) in enumerate(aggs) if op in (O.MIN, O.MAX)]
    check(ctx, keys, n, vals, aggs, [O.I64], exact=exact)


def test_nulls_everywhere(ctx):
    rng = np.random.default_rng(9)
    n, g = 400_000, 3_000
    km = O.pack_mask(rng.random(n) < 0.01)
    keys = [(sparse_keys(rng, n, g), km, O.I64)]
    vals = [(rng.normal(0, 1, n) + 50, O.pack_mask(rng.random(n) < 0.01), O.F64),
            (rng.integers(-10**6, 10**6, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.3), O.I64)]
    aggs = [(c, op) for c in (0, 1) for op in (O.SUM, O.MEAN, O.MIN, O.MAX, O.COUNT)]
    # i64 sums/min/max/count are integer work: bit-exact
    exact = [i for i, (c, op) in enumerate(aggs) if c == 1 and op != O.MEAN or op in (O.MIN, O.MAX, O.COUNT)]
    got = check(ctx, keys, n, vals, aggs, [O.I64], exact=exact)
    assert got[1].sum() == 1            # exactly one NULL group (grouping.rs:74)


def test_all_null_groups_and_sentinels(ctx):
    """Source-derived quirks: empty => 0.0, sentinel min/max => 0.0, count
    includes nulls, i64 sum wraps, NaN propagates through sum and is ignored by min/max."""
    key = [(np.array([1, 1, 2, 2, 3, -1, -1], np.int64), None, O.I64)]     # -1 == table sentinel bits
    vi = (np.array([5, 7, 0, 0, 9, 2**62, 2**62], np.int64), O.pack_mask([0, 0, 1, 1, 0, 0, 0]), O.I64)
    vf = (np.array([5.0, np.nan, 0, 0, np.inf, -0.0, 0.0]), O.pack_mask([0, 0, 1, 1, 0, 0, 0]), O.F64)
    aggs = [(c, op) for c in (0, 1) for op in (O.SUM, O.MEAN, O.MIN, O.MAX, O.COUNT)]
    check(ctx, key, 7, [vi, vf], aggs, [O.I64], exact=range(len(aggs)))
    big = (np.array([2**62] * 4 + [np.iinfo(np.int64).max, np.iinfo(np.int64).min], np.int64), None, O.I64)
    k2 = [(np.array([0, 0, 0, 0, 1, 2], np.int64), None, O.I64)]
    check(ctx, k2, 6, [big], FIVE, [O.I64], exact=range(5))


def test_other_key_dtypes(ctx):
    rng = np.random.default_rng(21)
    n = 200_000
    v = [(rng.normal(10, 3, n), None, O.F64)]
    codes = rng.integers(0, 10_000, n).astype(np.uint32)
    check(ctx, [(codes, O.pack_mask(rng.random(n) < 0.001), O.U32CODE)], n, v, FIVE, [O.U32CODE], exact=EXACT5)
    kf = rng.choice(np.array([0.0, -0.0, 1.5, np.nan, np.inf, -2.25, 1e300]), n)
    bits = kf.view(np.uint64).copy()
    bits[np.isnan(kf) & (rng.random(n) < 0.5)] = 0xFFF8000000000123      # another NaN payload
    check(ctx, [(bits.view(np.float64), None, O.F64)], n, v, FIVE, [O.F64], exact=EXACT5)
    kb = np.packbits(rng.random(n) < 0.3, bitorder="little")
    check(ctx, [(kb, O.pack_mask(rng.random(n) < 0.05), O.BOOLBITS)], n, v, FIVE, [O.BOOLBITS], exact=EXACT5)


def test_skew_and_extremes(ctx):
    rng = np.random.default_rng(33)
    n = 600_000
    v = [(rng.normal(100, 10, n), None, O.F64)]
    # 80/20 skew (reference benches/enhanced_comprehensive_benchmark.rs:53-59)
    g = 20_000
    hot = rng.random(n) < 0.8
    ids = np.where(hot, rng.integers(0, g // 5, n), rng.integers(0, g, n)).astype(np.uint64)
    keys = (ids * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    check(ctx, [(keys, None, O.I64)], n, v, FIVE, [O.I64], exact=EXACT5)
    # one key for every row; every row its own key
    check(ctx, [(np.full(n, 12345, np.int64), None, O.I64)], n, v, FIVE, [O.I64], exact=EXACT5)
    check(ctx, [(np.arange(n, dtype=np.int64) * 1024, None, O.I64)], n, v, FIVE, [O.I64], exact=EXACT5)


@pytest.mark.parametrize("no_direct", [0, 1])
def test_low_cardinality_direct_and_partitioned_paths_agree(ctx, no_direct):
    """Few groups: the partition-free direct path (default) and the radix path must both be exact,
    including NULL keys, the sentinel-valued key (-1), null values and one dominant key."""
    rng = np.random.default_rng(101)
    n = 5_000_000
    k = rng.integers(-3, 40, n).astype(np.int64)
    k[rng.random(n) < 0.5] = 7                                   # dominant key
    keys = [(k, O.pack_mask(rng.random(n) < 0.01), O.I64)]
    vals = [(rng.normal(100, 10, n), O.pack_mask(rng.random(n) < 0.05), O.F64),
            (rng.integers(-99, 99, n).astype(np.int64), None, O.I64)]
    aggs = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (0, O.MAX), (0, O.COUNT), (1, O.SUM), (1, O.MIN), (1, O.MAX), (1, O.MEAN)]
    ctx.set_option("no_direct", no_direct)
    try:
        check(ctx, keys, n, vals, aggs, [O.I64], exact=[2, 3, 4, 5, 6, 7])
        # the path actually taken: the direct path reports 0 radix partitions, the partitioned one >= 1
        assert (ctx.timings()["n_partitions"] == 0) == (no_direct == 0), ctx.timings()
        kb = [(np.packbits(rng.random(n) < 0.3, bitorder="little"), None, O.BOOLBITS)]
        check(ctx, kb, n, vals, aggs, [O.BOOLBITS], exact=[2, 3, 4, 5, 6, 7])
    finally:
        ctx.set_option("no_direct", 0)


def test_hot_key_partitions_are_sliced(ctx):
    """A dominant key inside a high-cardinality column: its radix partition is cut into row slices
    handled by several workgroups and merged; results must stay exact."""
    rng = np.random.default_rng(202)
    n, g = 3_000_000, 400_000
    k = sparse_keys(rng, n, g)
    k[rng.random(n) < 0.6] = 424242                              # 60 % of the rows share one key
    k[rng.random(n) < 0.1] = -1                                  # and 10 % the table-sentinel value
    keys = [(k, O.pack_mask(rng.random(n) < 0.05), O.I64)]       # 5 % null keys: a large NULL partition
    vals = [(rng.normal(100, 10, n), O.pack_mask(rng.random(n) < 0.05), O.F64),
            (rng.integers(-99, 99, n).astype(np.int64), None, O.I64)]
    aggs = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (0, O.MAX), (0, O.COUNT), (1, O.SUM), (1, O.MIN), (1, O.MAX)]
    # (pieces of the average partition's size by default; as long as the cutting threshold with wide_slices; a forced length)
    for slice_rows, wide in ((0, 0), (0, 1), (50_000, 0)):
        ctx.set_option("slice_rows", slice_rows); ctx.set_option("wide_slices", wide)
        try:
            check(ctx, keys, n, vals, aggs, [O.I64], exact=[2, 3, 4, 5, 6, 7])
        finally:
            ctx.set_option("slice_rows", 0); ctx.set_option("wide_slices", 0)
    # partials of a sliced input stay mergeable
    ctx.set_option("slice_rows", 50_000)
    try:
        ng, ns = ctx.groupby_partials(keys, n, vals, aggs)
        rec, counts = ctx.partials_split(1)
        ctx.groupby_merge(O.I64, rec, [O.F64, O.I64], [True, False], aggs)
        got = ctx.groupby_fetch(to_device=False)
    finally:
        ctx.set_option("slice_rows", 0)
    want = O.groupby_agg(keys, n, vals, aggs)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[2, 3, 4, 5, 6, 7])


def test_a_dominant_key_is_not_read_as_clustered_rows(ctx):
    """90 % of the rows on ONE key in random order: neighbours share their key most of the time, but so do rows far apart (the
    estimate's far-pair control) — the call keeps the lean aggregate / the absorb pass instead of the clustered-rows kernels;
    the same rows SORTED by key must give the same groups."""
    rng = np.random.default_rng(207)
    n, g = 20_000_000, 300_000
    ids = rng.integers(0, g, n)
    ids[rng.random(n) < 0.9] = 7
    keys = [(sparse_keys_from(ids), None, O.I64)]
    vals = [(rng.normal(3, 2, n), None, O.F64), (rng.normal(-1, 5, n), None, O.F64)]
    aggs = [(c, op) for c in range(2) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
    exact = [2, 3, 6, 7, 8]
    check(ctx, keys, n, vals, aggs, [O.I64], exact=exact)
    t = ctx.timings()
    assert t["absorbed_rows"] > 0.8 * n, t             # (the absorb pass took the hot key: only possible when the rows did not read as clustered)
    order = np.argsort(ids, kind="stable")
    keys_sorted = [(sparse_keys_from(ids[order]), None, O.I64)]
    # (sorted: the hot key's run is far longer than the far pairs reach, so it still reads as a dominant key — any path must stay exact)
    check(ctx, keys_sorted, n, [(vals[0][0][order], None, O.F64), (vals[1][0][order], None, O.F64)], aggs, [O.I64], exact=exact)


def test_first_last_of_a_hot_key_are_sliced_too(ctx):
    """First / Last keep the min / max row index per group: those merge across the row slices of an oversized partition like any
    other state, and the value is looked up behind the merge (one workgroup used to walk the hot key's whole partition)."""
    rng = np.random.default_rng(203)
    n, g = 3_000_000, 400_000
    k = sparse_keys(rng, n, g)
    k[rng.random(n) < 0.5] = 424242
    keys = [(k, O.pack_mask(rng.random(n) < 0.02), O.I64)]
    vals = [(rng.normal(100, 10, n), O.pack_mask(rng.random(n) < 0.3), O.F64),
            (rng.integers(-99, 99, n).astype(np.int64), None, O.I64)]
    aggs = [(0, O.FIRST), (0, O.LAST), (1, O.FIRST), (1, O.LAST), (0, O.SUM), (1, O.MAX), (0, O.COUNT)]
    for slice_rows in (0, 40_000):
        ctx.set_option("slice_rows", slice_rows)
        try:
            check(ctx, keys, n, vals, aggs, [O.I64], exact=[0, 1, 2, 3, 5, 6])
        finally:
            ctx.set_option("slice_rows", 0)
    # (partial records for other shards still refuse First / Last: a row index means nothing there)
    import pandrs_amd as pa
    with pytest.raises(pa.OperationFailed):
        ctx.groupby_partials(keys, n, vals, aggs)


def test_std_var_of_a_hot_key_are_sliced_too(ctx):
    """Std / Var: every row slice of an oversized partition runs the reference's two passes over its own rows; the merge of the
    slices' records adds the between-slice term n_i (m_i - m)^2 in a second pass of its own (aggregate.hpp, MergeVar)."""
    rng = np.random.default_rng(204)
    n, g = 3_000_000, 400_000
    k = sparse_keys(rng, n, g)
    k[rng.random(n) < 0.5] = 424242
    k[rng.random(n) < 0.05] = -1
    keys = [(k, O.pack_mask(rng.random(n) < 0.02), O.I64)]
    drift = np.linspace(0.0, 50.0, n)                 # slice means differ: the between-slice term matters
    vals = [(1e6 + drift + rng.normal(0, 1, n), O.pack_mask(rng.random(n) < 0.3), O.F64),
            (rng.integers(-99, 99, n).astype(np.int64) + (np.arange(n) // 100_000), None, O.I64),
            (rng.normal(0, 1, n), None, O.F64)]
    aggs = [(0, O.STD), (0, O.VAR), (1, O.STD), (1, O.VAR), (2, O.STD), (0, O.MEAN), (1, O.SUM), (2, O.FIRST), (0, O.COUNT)]
    for slice_rows in (0, 40_000):
        ctx.set_option("slice_rows", slice_rows)
        try:
            check(ctx, keys, n, vals, aggs, [O.I64], exact=[6, 7, 8])
        finally:
            ctx.set_option("slice_rows", 0)


@pytest.mark.parametrize("skew", [False, True])
def test_mid_cardinality_takes_few_sliced_partitions(ctx, skew):
    """>= 16 M rows, >= 4 states, a few thousand groups: the engine picks 16-64 large partitions and
    cuts them into ~512 row slices whose partial records are merged ('Mid cardinalities');
    C3's shape (u32 codes, 80/20 skew, nulls in one column) against the oracle."""
    rng = np.random.default_rng(77 + skew)
    n, g = 17_000_000, 6_000
    ids = rng.integers(0, g, n)
    if skew:
        hot = rng.random(n) < 0.8
        ids = np.where(hot, rng.integers(0, g // 5, n), ids)
    keys = [(ids.astype(np.uint32), None, O.U32CODE)]
    vals = [(rng.normal(100, 10, n), None, O.F64),
            (rng.integers(-10**6, 10**6, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.1), O.I64)]
    aggs = [(c, op) for c in range(2) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
    # (an f64 and a masked i64 column have no uniform profile: since late round 4 the lean kernel takes them in rounds grouped by profile,
    # at its own fan-out — also checked; the mid-cardinality plan is the older kernel's, `no_profile_rounds`)
    for no_profile_rounds in (1, 0):
        ctx.set_option("no_profile_rounds", no_profile_rounds)
        try:
            check(ctx, keys, n, vals, aggs, [O.U32CODE], exact=[2, 3, 4, 6, 7, 8])
            t = ctx.timings()
        finally:
            ctx.set_option("no_profile_rounds", 0)
        if no_profile_rounds:
            assert 16 <= t["n_partitions"] <= 64, t["n_partitions"]


def test_mid_cardinality_sliced_partitions_with_null_keys_and_a_hot_key(ctx):
    """The same rule on sparse i64 keys with 3 % NULL keys (their own partition), the table-sentinel key, one key on
    30 % of the rows (its partition is sliced again by the hot-key rule) and wrapping i64 sums."""
    rng = np.random.default_rng(1234)
    n, g = 16_800_000, 20_000
    ids = rng.integers(0, g, n)
    ids[rng.random(n) < 0.3] = 11
    k = sparse_keys_from(ids)
    k[ids == 12] = -1
    keys = [(k, O.pack_mask(rng.random(n) < 0.03), O.I64)]
    vals = [(rng.integers(-2**62, 2**62, n).astype(np.int64), None, O.I64),
            (rng.normal(0, 1e6, n), O.pack_mask(rng.random(n) < 0.5), O.F64)]
    aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (1, O.SUM), (1, O.MEAN), (1, O.MIN), (1, O.MAX), (1, O.COUNT)]
    for no_profile_rounds in (1, 0):            # (mixed kinds: the older kernel's mid-cardinality plan, and the lean kernel's rounds grouped by profile)
        ctx.set_option("no_profile_rounds", no_profile_rounds)
        try:
            check(ctx, keys, n, vals, aggs, [O.I64], exact=[0, 1, 2, 5, 6, 7])
            t = ctx.timings()
        finally:
            ctx.set_option("no_profile_rounds", 0)
        if no_profile_rounds:
            assert 16 <= t["n_partitions"] <= 64, t


def test_two_level_for_huge_cardinality(ctx):
    """More groups than one radix level holds (forced here with a small partition cap): rows are
    split by an independent hash into super-partitions, the engine runs per super-partition and the
    group lists are concatenated.  Every op, null keys, null values, and the merge path."""
    rng = np.random.default_rng(303)
    n, g = 2_000_000, 1_500_000
    keys = [(sparse_keys(rng, n, g), O.pack_mask(rng.random(n) < 0.01), O.I64)]
    vals = [(rng.normal(100, 10, n), O.pack_mask(rng.random(n) < 0.05), O.F64),
            (rng.integers(-99, 99, n).astype(np.int64), None, O.I64)]
    aggs = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (0, O.MAX), (0, O.COUNT), (1, O.SUM), (1, O.MIN), (1, O.MAX),
            (0, O.STD), (0, O.FIRST), (1, O.LAST), (1, O.VAR)]
    exact = [2, 3, 4, 5, 6, 7, 9, 10]
    ctx.set_option("p_max", 48)
    try:
        got = check(ctx, keys, n, vals, aggs, [O.I64], exact=exact)
        assert ctx.timings()["n_partitions"] >= 2 and got[1].sum() == 1
        # mergeable subset through partials -> split -> merge, both stages two-level
        m_aggs = aggs[:8]
        ng, ns = ctx.groupby_partials(keys, n, vals, m_aggs)
        rec, counts = ctx.partials_split(1)
        ctx.groupby_merge(O.I64, rec, [O.F64, O.I64], [True, False], m_aggs)
        merged = ctx.groupby_fetch(to_device=False)
    finally:
        ctx.set_option("p_max", 0)
    want = O.groupby_agg(keys, n, vals, aggs[:8])
    assert_groupby_equal(merged, want, [O.I64], int_exact_rows=[2, 3, 4, 5, 6, 7])


def test_std_var_first_last(ctx, golden):
    """aggregation.rs:557-624, :675-742 + :881-903: two-pass Bessel variance, value at first/last row."""
    rng = np.random.default_rng(17)
    n, g = 500_000, 7_000
    keys = [(sparse_keys(rng, n, g), O.pack_mask(rng.random(n) < 0.001), O.I64)]
    vf = (rng.normal(1e6, 3.0, n), O.pack_mask(rng.random(n) < 0.1), O.F64)       # large mean: two-pass matters
    vi = (rng.integers(-10**5, 10**5, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.3), O.I64)
    vp = (rng.normal(0, 1, n), None, O.F64)
    aggs = [(0, O.STD), (0, O.VAR), (0, O.FIRST), (0, O.LAST), (0, O.MEAN),
            (1, O.STD), (1, O.VAR), (1, O.FIRST), (1, O.LAST), (1, O.SUM),
            (2, O.VAR), (2, O.FIRST), (2, O.COUNT)]
    exact = [i for i, (_, op) in enumerate(aggs) if op in (O.FIRST, O.LAST, O.COUNT)] + [9]
    check(ctx, keys, n, [vf, vi, vp], aggs, [O.I64], exact=exact)
    # singletons and all-null groups: var/std 0.0, first/last of a null => 0.0
    k = [(np.array([1, 2, 2, 3, 3, 3], np.int64), None, O.I64)]
    v = (np.array([5.0, 1.0, 2.0, 0.0, 0.0, 0.0]), O.pack_mask([0, 0, 0, 1, 1, 1]), O.F64)
    check(ctx, k, 6, [v], [(0, O.STD), (0, O.VAR), (0, O.FIRST), (0, O.LAST)], [O.I64], exact=range(4))
    # the reference's own known answers: std(A)=20, first 10/20, last 50/40
    case = golden["groupby"][2]
    codes, pool = codes_of(case["key_strings"])
    kc, kn, oa = ctx.groupby_agg([(codes, None, O.U32CODE)], 5, [(np.array(case["values_f64"]), None, O.F64)],
                                 [(0, O.STD), (0, O.FIRST), (0, O.LAST)])
    for gi in range(2):
        e = case["expect"][pool[int(kc[0, gi])]]
        if "std" in e:
            assert oa[0, gi] == pytest.approx(e["std"], abs=1e-3)
        assert oa[1, gi] == e["first"] and oa[2, gi] == e["last"]
    import pandrs_amd as pa
    with pytest.raises(pa.OperationFailed):          # non-mergeable ops cannot produce partials
        ctx.groupby_partials(k, 6, [v], [(0, O.STD)])
    with pytest.raises(pa.OperationFailed):
        ctx.groupby_partials(k, 6, [v], [(0, O.MEDIAN)])


def test_median_small_and_edge_cases(ctx):
    """aggregation.rs:585-604 / :703-722: middle of the sorted non-null values; even count = mean of the
    two middles, for Int64 with the ADD IN i64 (wraps); no non-null value => 0.0; null keys are a group."""
    k = (np.array([1, 1, 1, 2, 2, 3, 3, 3, 3, -1, 9, 9], np.int64), O.pack_mask([0] * 10 + [1, 1]), O.I64)
    vf = (np.array([5.0, 1.0, 3.0, 2.0, 8.0, 7.0, 7.0, -1.0, 4.0, 6.5, 10.0, 20.0]), None, O.F64)
    vi = (np.array([5, 1, 3, 2, 9, 7, 7, -1, 4, 6, 2**62, 2**62], np.int64),
          O.pack_mask([0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]), O.I64)       # group 2: all null -> 0.0; NULL group: i64 add wraps
    got = check(ctx, k, 12, [vf, vi], [(0, O.MEDIAN), (1, O.MEDIAN), (0, O.SUM), (1, O.COUNT)], [O.I64], exact=[0, 1, 3])
    by_key = {(int(np.int64(c)), int(nl)): (a, b) for c, nl, a, b in zip(got[0][0], got[1][0], got[2][0], got[2][1])}
    assert by_key[(1, 0)] == (3.0, 3.0) and by_key[(2, 0)] == (5.0, 0.0) and by_key[(3, 0)] == (5.5, 5.5)
    assert [v for (c, nl), v in by_key.items() if nl == 1] == [(15.0, float(np.int64(-2**63)) / 2.0)]


@pytest.mark.parametrize("n,g,skew", [(200_000, 1_000, False), (3_000_000, 40_000, False), (2_000_000, 50, True),
                                     (9_000_000, 200_000, True)])     # a 5.4 M-row hot partition: split once more, then selected
def test_median_random(ctx, n, g, skew):
    """Groups from a few rows to far beyond one LDS tile (the skewed case: 50 groups, one with ~60 % of
    the rows => multi-tile partitions, merge passes), f64 and masked i64 columns, next to other ops."""
    rng = np.random.default_rng(n % 1000 + g)
    ids = rng.integers(0, g, n)
    if skew:
        ids[rng.random(n) < 0.6] = 7
    k = (sparse_keys_from(ids), O.pack_mask(rng.random(n) < 0.001), O.I64)
    vf = (np.round(rng.normal(50, 100, n), 1) + 0.0, None, O.F64)            # rounded: many ties (+ 0.0: no -0.0, whose place among the zeros is unspecified in the reference's sort)
    vi = (rng.integers(-10**6, 10**6, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.1), O.I64)
    for generic in (0, 1):       # 0: LDS group-sort fast path (+ general path for flagged partitions), 1: general path only
        ctx.set_option("median_generic", generic)
        try:
            check(ctx, k, n, [vf, vi], [(0, O.MEDIAN), (1, O.MEDIAN), (0, O.MIN), (1, O.MAX), (1, O.COUNT)], [O.I64], exact=[0, 1, 2, 3, 4])
        finally:
            ctx.set_option("median_generic", 0)


@pytest.mark.parametrize("n,g,skew", [(300_000, 2_000, False), (2_500_000, 30_000, False), (2_000_000, 40, True)])
def test_nunique_random(ctx, n, g, skew):
    """PANDRS_HIP_AGG_NUNIQUE (legacy AggFunc::Nunique): distinct non-null values per group, exact; few distinct
    values per group (so duplicates abound), nulls, NaNs, signed zeros, a hot key, beside other aggregates."""
    rng = np.random.default_rng(n + g)
    ids = rng.integers(0, g, n)
    if skew:
        ids[rng.random(n) < 0.7] = 3
    keys = [(sparse_keys_from(ids), O.pack_mask(rng.random(n) < 0.02), O.I64)]
    vf = rng.integers(-6, 6, n).astype(np.float64) / 2.0
    vf[rng.random(n) < 0.01] = np.nan
    vf[rng.random(n) < 0.05] = -0.0
    vi = rng.integers(-2**62, 2**62, n)
    small = rng.integers(0, 7, n).astype(np.int64) - 3
    vals = [(vf, O.pack_mask(rng.random(n) < 0.1), O.F64), (np.where(rng.random(n) < 0.6, small, vi), None, O.I64)]
    aggs = [(0, O.NUNIQUE), (1, O.NUNIQUE), (0, O.COUNT), (1, O.MIN), (1, O.MEDIAN)]    # (Median of a NaN-holding column is unspecified)
    for generic in (0, 1):
        ctx.set_option("median_generic", generic)
        try:
            check(ctx, keys, n, vals, aggs, [O.I64], exact=[0, 1, 2, 3])
        finally:
            ctx.set_option("median_generic", 0)


def test_sort_based_aggregates_fast_path_shapes(ctx):
    """The LDS group-sort path of Median / Nunique at its edges: runs of every length around the wave /
    workgroup hand-over (GS_BIG = 512), a partition-filling group, more distinct keys per partition than the
    LDS table takes (=> general path), the key ~0 and a NULL-key group that fills several partitions' worth."""
    rng = np.random.default_rng(4242)
    sizes = [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 257, 511, 512, 513, 1000, 1023, 1025, 2047, 4097, 9000, 15360, 15361, 40_000]
    ids = np.repeat(np.arange(len(sizes)), sizes)
    n = len(ids)
    perm = rng.permutation(n)
    k = sparse_keys_from(ids)[perm]
    k[ids[perm] == 5] = -1                                     # one group keyed by the table sentinel ~0
    vals = [(np.round(rng.normal(0, 50, n), 0) + 0.0, None, O.F64), (rng.integers(-40, 40, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.3), O.I64)]
    aggs = [(0, O.MEDIAN), (1, O.MEDIAN), (0, O.NUNIQUE), (1, O.NUNIQUE), (0, O.COUNT)]
    check(ctx, [(k, O.pack_mask(rng.random(n) < 0.2), O.I64)], n, vals, aggs, [O.I64], exact=[0, 1, 2, 3, 4])
    # nearly unique keys: ~2 rows per group, far more groups per partition than the LDS table holds
    n2 = 600_000
    k2 = sparse_keys_from(rng.integers(0, n2 // 2, n2))
    v2 = [(rng.integers(0, 3, n2).astype(np.float64), None, O.F64)]
    check(ctx, [(k2, None, O.I64)], n2, v2, [(0, O.MEDIAN), (0, O.NUNIQUE)], [O.I64], exact=[0, 1])


@pytest.mark.parametrize("g", [1, 7, 60])
def test_median_of_few_huge_groups_takes_the_selection_path(ctx, g):
    """Low-cardinality keys: a group is far larger than an LDS partition and sits alone in its hash partition —
    single_key_median_kernel (radix select, no sort); with 60 keys some partitions hold two keys and stay on the
    general path.  Constant and two-valued columns (every lane in one histogram bin), a masked i64 column, a
    large NULL-key group, odd and even group sizes."""
    rng = np.random.default_rng(900 + g)
    n = 2_500_001
    ids = rng.integers(0, g, n)
    keys = [(sparse_keys_from(ids), O.pack_mask(rng.random(n) < 0.15), O.I64)]
    vals = [(np.round(rng.normal(0, 1000, n), 2) + 0.0, None, O.F64),
            (rng.integers(-2**62, 2**62, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.3), O.I64),
            (np.full(n, 42.5), None, O.F64),
            ((rng.random(n) < 0.5).astype(np.int64) * 7, None, O.I64)]
    aggs = [(0, O.MEDIAN), (1, O.MEDIAN), (2, O.MEDIAN), (3, O.MEDIAN), (0, O.COUNT), (0, O.NUNIQUE)]
    for generic in (0, 1):
        ctx.set_option("median_generic", generic)
        try:
            check(ctx, keys, n, vals, aggs, [O.I64], exact=[0, 1, 2, 3, 4, 5])
        finally:
            ctx.set_option("median_generic", 0)


def test_nunique_edge_cases_and_frame_shortcut(ctx):
    key = (np.array([1, 1, 1, 1, 2, 2, 3, 3, 3], np.int64), None, O.I64)
    vf = (np.array([2.5, 2.5, -0.0, 0.0, 7.0, 0.0, np.nan, np.nan, 1.0]), O.pack_mask([0, 0, 0, 0, 1, 1, 0, 0, 0]), O.F64)
    kc, kn, oa = ctx.groupby_agg([key], 9, [vf], [(0, O.NUNIQUE)])
    assert dict(zip(kc[0].view(np.int64).tolist(), oa[0].tolist())) == {1: 2.0, 2: 0.0, 3: 3.0}
    # empty input, one group of one row
    kc, kn, oa = ctx.groupby_agg([(np.zeros(0, np.int64), None, O.I64)], 0, [(np.zeros(0), None, O.F64)], [(0, O.NUNIQUE)])
    assert oa.shape[1] == 0
    kc, kn, oa = ctx.groupby_agg([(np.array([5], np.int64), None, O.I64)], 1, [(np.array([1.5]), None, O.F64)], [(0, O.NUNIQUE)])
    assert oa[0].tolist() == [1.0]
    # string column: OperationFailed like every numeric op (aggregation.rs:748)
    import pandrs_amd as pa
    with pytest.raises(pa.OperationFailed):
        ctx.groupby_agg([key], 9, [(np.zeros(9, np.uint32), None, O.U32CODE)], [(0, O.NUNIQUE)])
    # partial states of a sort-based aggregate are not mergeable
    with pytest.raises(pa.OperationFailed):
        ctx.groupby_partials([key], 9, [vf], [(0, O.NUNIQUE)])
    from pandrs_amd.frame import OptimizedDataFrame, Int64Column, StringColumn
    df = OptimizedDataFrame()
    df.add_column("k", StringColumn(["a", "b", "a", "a", "b"]))
    df.add_column("v", Int64Column([3, 3, 3, 4, 3]))
    r = df.group_by(["k"]).nunique("v")                      # legacy GroupBy::nunique: alias "{col}_nunique"
    assert dict(zip(r.column("k").to_list(), r.column("v_nunique").data.tolist())) == {"a": 2.0, "b": 1.0}


def test_median_multi_key_and_string_codes(ctx):
    rng = np.random.default_rng(99)
    n = 300_000
    k0 = (rng.integers(0, 30, n).astype(np.uint32), None, O.U32CODE)
    k1 = (rng.integers(-5, 5, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.05), O.I64)
    v = (rng.normal(5, 2, n), O.pack_mask(rng.random(n) < 0.3), O.F64)
    check(ctx, [k0, k1], n, [v], [(0, O.MEDIAN), (0, O.COUNT)], [O.U32CODE, O.I64], exact=[0, 1])


def test_multi_key_packed(ctx, golden):
    """Vec<String> keys of the reference (grouping.rs:62-104) as packed composite cells."""
    rng = np.random.default_rng(23)
    n = 300_000
    k1 = (rng.integers(0, 300, n).astype(np.uint32), O.pack_mask(rng.random(n) < 0.01), O.U32CODE)
    k2 = (rng.integers(-50, 50, n).astype(np.int64) * 1000 - 7, None, O.I64)
    k3 = (np.packbits(rng.random(n) < 0.5, bitorder="little"), O.pack_mask(rng.random(n) < 0.05), O.BOOLBITS)
    v = [(rng.normal(100, 10, n), None, O.F64)]
    got = ctx.groupby_agg([k1, k2, k3], n, v, FIVE)
    want = O.groupby_agg([k1, k2, k3], n, v, FIVE)
    assert_groupby_equal(got, want, [O.U32CODE, O.I64, O.BOOLBITS], int_exact_rows=EXACT5)
    # two full-range i64 keys do not fit 64 bits of codes: dictionary-encoded (every row its own group here)
    kw = [(rng.integers(-2**62, 2**62, n), None, O.I64), (rng.integers(-2**62, 2**62, n), None, O.I64)]
    assert_groupby_equal(ctx.groupby_agg(kw, n, v, FIVE), O.groupby_agg(kw, n, v, FIVE), [O.I64, O.I64], int_exact_rows=EXACT5)
    # many rows, few composite groups: the packed cells must survive the partition-free direct path
    n2 = 5_000_000
    ka = (rng.integers(-100, 100, n2).astype(np.int64), None, O.I64)
    kb = (rng.integers(0, 5, n2).astype(np.uint32), None, O.U32CODE)
    v2 = [(rng.normal(50, 20, n2), O.pack_mask(rng.random(n2) < 0.1), O.F64)]
    got = ctx.groupby_agg([ka, kb], n2, v2, [(0, O.MAX), (0, O.MEAN)])
    want = O.groupby_agg([ka, kb], n2, v2, [(0, O.MAX), (0, O.MEAN)])
    assert_groupby_equal(got, want, [O.I64, O.U32CODE], int_exact_rows=[0])
    c = golden["groupby_two_keys"]
    a, _ = codes_of(c["key1_strings"])
    b, _ = codes_of(c["key2_strings"])
    kc, kn, oa = ctx.groupby_agg([(a, None, O.U32CODE), (b, None, O.U32CODE)], 4,
                                 [(np.array(c["values_f64"]), None, O.F64)], [(0, O.SUM)])
    assert kc.shape == (2, c["expect_n_groups"]) and sorted(oa[0].tolist()) == [1.0, 2.0, 3.0, 4.0]


def test_empty_and_errors(ctx):
    import pandrs_amd as pa
    e = np.zeros(0, np.int64)
    kc, kn, oa = ctx.groupby_agg([(e, None, O.I64)], 0, [(np.zeros(0), None, O.F64)], FIVE)
    assert kc.shape == (1, 0) and oa.shape == (5, 0)        # tests/edge_cases_test.rs:46-80
    k = [(np.array([1, 2, 1], np.int64), None, O.I64)]
    with pytest.raises(pa.OperationFailed):                # aggregation.rs:748
        ctx.groupby_agg(k, 3, [(np.zeros(3, np.uint32), None, O.U32CODE)], [(0, O.SUM)])
    with pytest.raises(pa.OperationFailed):                # aggregation.rs:744
        ctx.groupby_agg(k, 3, [(np.zeros(3), None, O.F64)], [(0, O.CUSTOM)])
    with pytest.raises(pa.PandrsHipError):
        ctx.groupby_agg(k, 3, [(np.zeros(3), None, O.F64)], [(5, O.SUM)])
    # Count works on any dtype (aggregation.rs:743)
    kc, kn, oa = ctx.groupby_agg(k, 3, [(np.zeros(3, np.uint32), None, O.U32CODE)], [(0, O.COUNT)])
    assert sorted(oa[0].tolist()) == [1.0, 2.0]


def test_overflow_retry_path(ctx):
    """A wrong cardinality hint makes LDS tables overflow; the engine must retry with more
    partitions and still be exact."""
    rng = np.random.default_rng(5)
    n, g = 500_000, 200_000
    keys = [(sparse_keys(rng, n, g), None, O.I64)]
    vals = [(rng.normal(100, 10, n), None, O.F64)]
    ctx.set_option("groups_hint", 10)
    ctx.set_option("partitions", 1)
    try:
        check(ctx, keys, n, vals, FIVE, [O.I64], exact=EXACT5)
        assert ctx.timings()["retries"] >= 1
    finally:
        ctx.set_option("groups_hint", 0)
        ctx.set_option("partitions", 0)


def test_unstaged_scatter_variant(ctx):
    rng = np.random.default_rng(6)
    n, g = 300_000, 30_000
    keys = [(sparse_keys(rng, n, g), None, O.I64)]
    vals = [(rng.normal(100, 10, n), O.pack_mask(rng.random(n) < 0.1), O.F64)]
    ctx.set_option("scatter_staged", 0)
    try:
        check(ctx, keys, n, vals, FIVE, [O.I64], exact=EXACT5)
    finally:
        ctx.set_option("scatter_staged", 1)


def test_device_resident_columns(ctx):
    import torch
    rng = np.random.default_rng(8)
    n, g = 700_000, 9_000
    k = sparse_keys(rng, n, g)
    v = rng.normal(100, 10, n)
    m = O.pack_mask(rng.random(n) < 0.02)
    d = "cuda:0"
    got = ctx.groupby_agg([(torch.from_numpy(k).to(d), None, O.I64)], n,
                          [(torch.from_numpy(v).to(d), torch.from_numpy(m).to(d), O.F64)], FIVE)
    got = tuple(t.cpu().numpy() for t in got)
    got = (got[0].view(np.uint64), got[1], got[2])
    want = O.groupby_agg([(k, None, O.I64)], n, [(v, m, O.F64)], FIVE)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=EXACT5)


def test_partials_split_merge_roundtrip(ctx):
    """Two row-range shards -> partials -> owner split -> merge == oracle on the whole input
    (the single-process rehearsal of the multi-GPU exchange)."""
    rng = np.random.default_rng(12)
    n, g, ranks = 400_000, 25_000, 2
    k = sparse_keys(rng, n, g)
    km = O.pack_mask(rng.random(n) < 0.001)
    v0 = rng.normal(100, 10, n)
    v1 = rng.integers(-1000, 1000, n).astype(np.int64)
    m1 = O.pack_mask(rng.random(n) < 0.2)
    aggs = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (0, O.MAX), (0, O.COUNT), (1, O.SUM), (1, O.MEAN), (1, O.MIN), (1, O.MAX)]
    half = n // 2 // 8 * 8
    inbox = [[] for _ in range(ranks)]
    for lo, hi in ((0, half), (half, n)):
        sl = slice(lo, hi)
        kmask = O.pack_mask(np.unpackbits(km, bitorder="little")[:n][sl])
        vmask = O.pack_mask(np.unpackbits(m1, bitorder="little")[:n][sl])
        ng, ns = ctx.groupby_partials([(k[sl], kmask, O.I64)], hi - lo,
                                      [(v0[sl], None, O.F64), (v1[sl], vmask, O.I64)], aggs)
        rec, counts = ctx.partials_split(ranks)
        assert sum(counts) == ng and rec.shape == (ng, 2 + ns)
        off = 0
        for r, c in enumerate(counts):
            inbox[r].append(rec[off:off + c])
            off += c
    outs = []
    for r in range(ranks):
        ctx.groupby_merge(O.I64, np.concatenate(inbox[r]), [O.F64, O.I64], [False, True], aggs)
        outs.append(ctx.groupby_fetch(to_device=False))
    got = tuple(np.concatenate([o[i] for o in outs], axis=1) for i in range(3))
    want = O.groupby_agg([(k, km, O.I64)], n, [(v0, None, O.F64), (v1, m1, O.I64)], aggs)
    exact = [i for i, (c, op) in enumerate(aggs) if (c == 1 and op != O.MEAN) or op in (O.MIN, O.MAX, O.COUNT)]
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=exact)


def test_reduce_column(ctx, golden):
    rng = np.random.default_rng(4)
    n = 1_000_003
    for col in [(rng.normal(5, 2, n), O.pack_mask(rng.random(n) < 0.1), O.F64),
                (rng.integers(-10**9, 10**9, n).astype(np.int64), None, O.I64)]:
        got, gc = ctx.reduce_column(col, n)
        want, wc = O.reduce_column(col, n)
        assert gc == wc
        np.testing.assert_allclose(got[:2], want[:2], rtol=1e-9)
        assert got[2] == want[2] and got[3] == want[3]
    for c in golden["reductions"]:
        if "f64_range" in c:
            col = (np.arange(c["f64_range"][0], c["f64_range"][1] + 1, dtype=np.float64), None, O.F64)
        elif "f64" in c:
            col = (np.array(c["f64"], np.float64), None, O.F64)
        else:
            col = (np.array(c["i64"], np.int64), None, O.I64)
        got, _ = ctx.reduce_column(col, len(col[0]))
        for i, name in enumerate(("sum", "mean", "min", "max")):
            if name in c:
                assert got[i] == pytest.approx(float(c[name]), abs=1e-10), (c["cite"], name)


def test_full_size_properties():
    """BASELINE C2 scale (100 M rows, 1 M groups, 4 f64 columns): too big for the oracle, so
    check size-independent properties: group count, sum of counts == N, sum of sums == column
    total (linearity, 1e-9), global min/max == min/max of per-group min/max, mean*count == sum."""
    import torch
    import pandrs_amd as pa
    n, g = 100_000_000, 1_000_000
    d = "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(43)
    ids = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64)
    keys = ids * -7046029254386353131 ^ 0x5555AAAA5555AAAA      # 0x9E3779B97F4A7C15 as i64, wraps
    vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(4)]
    aggs = [(c, op) for c in range(4) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
    c = pa.Context(0)
    try:
        ng = c.groupby_compute([(keys, None, O.I64)], n, [(v, None, O.F64) for v in vals], aggs)
        kc, kn, oa = c.groupby_fetch()
        assert ng == torch.unique(ids).numel()
        assert torch.unique(kc[0]).numel() == ng and int(kn.sum()) == 0
        cnt = oa[16]
        assert float(cnt.sum()) == n
        for col in range(4):
            s, mean, mn, mx = oa[4 * col:4 * col + 4]
            tot = float(vals[col].sum())
            assert abs(float(s.sum()) - tot) <= 1e-9 * abs(tot)
            assert float(mn.min()) == float(vals[col].min()) and float(mx.max()) == float(vals[col].max())
            assert torch.allclose(mean * cnt, s, rtol=1e-12, atol=0)
        # idempotence: grouping the group keys again yields every key once
        ng2 = c.groupby_compute([(kc[0].contiguous(), None, O.I64)], ng, [(oa[0].contiguous(), None, O.F64)], [(0, O.COUNT)])
        assert ng2 == ng
    finally:
        c.close()


@pytest.mark.parametrize("hot_share,hot_keys,g", [(0.8, 200_000, 1_000_000), (0.5, 500_000, 5_000_000)])
def test_config2_skew_full_size_properties(hot_share, hot_keys, g):
    """C2's own 80/20 variant (/root/reference/benches/enhanced_comprehensive_benchmark.rs:53-59) at full size — 100 M rows,
    80 % of them on a fifth of the 1 M keys — through the same size-independent properties as the uniform case above.  The strided
    sample cannot size the tail behind the 200 K hot keys (round 3: estimate 297 K, a third of the groups through an overflow run,
    4.9 ms); the estimate's second stage — a hash-slice census of a tenth of the rows — does, so the call is planned once with the
    right fan-out: no retry, no overflow run, estimate within 15 % of the truth; with the census off the old behaviour is still exact.
    The second shape is the low-repeat end of the same trigger: half the rows on 500 K keys in front of 5 M, where the sub-sample holds
    only ~30 triple sightings and they are still 3 sigma more than its doubletons predict (14.5 ms and three attempts without the
    census, 5.9 ms and one with it)."""
    import torch
    import pandrs_amd as pa
    n = 100_000_000
    d = "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(47)
    hot = torch.rand(n, device=d, generator=gen) < hot_share
    ids = torch.where(hot, torch.randint(0, hot_keys, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen))
    del hot
    keys = ids * -7046029254386353131 ^ 0x5555AAAA5555AAAA
    vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(4)]
    aggs = [(c, op) for c in range(4) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
    true_groups = torch.unique(ids).numel()
    c = pa.Context(0)
    try:
        for census in (1, 0):
            c.set_option("no_census", 1 - census)
            ng = c.groupby_compute([(keys, None, O.I64)], n, [(v, None, O.F64) for v in vals], aggs)
            t = c.timings()
            kc, kn, oa = c.groupby_fetch()
            assert ng == true_groups
            assert torch.unique(kc[0]).numel() == ng and int(kn.sum()) == 0
            cnt = oa[16]
            assert float(cnt.sum()) == n
            for col in range(4):
                s, mean, mn, mx = oa[4 * col:4 * col + 4]
                tot = float(vals[col].sum())
                assert abs(float(s.sum()) - tot) <= 1e-9 * abs(tot)
                assert float(mn.min()) == float(vals[col].min()) and float(mx.max()) == float(vals[col].max())
                assert torch.allclose(mean * cnt, s, rtol=1e-12, atol=0)
            if census:
                assert t["retries"] == 0, t                                  # neither a retry nor an overflow run
                assert 0.85 * true_groups <= t["estimated_groups"] <= 1.15 * true_groups, (t["estimated_groups"], true_groups)
            else:
                assert t["estimated_groups"] < 0.5 * true_groups, t         # what the strided sample alone makes of it
            del kc, kn, oa
    finally:
        c.set_option("no_census", 0)
        c.close()


def test_census_is_not_taken_on_uniform_keys_and_sizes_a_long_tail(ctx):
    """The census trigger (partition.hip, estimate_groups): singletons that the sample's repeating keys cannot explain.  Uniform keys
    never trigger it (their estimate is the model's, as before); a broad hot class in front of a long tail does, and the estimate is
    then within 15 % at a size the oracle checks too."""
    rng = np.random.default_rng(321)
    n = 12_000_000
    v = [(rng.normal(100, 10, n), None, O.F64)]
    aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.COUNT)]
    for g in (5_000, 120_000, 2_000_000):
        keys = [(sparse_keys(rng, n, g), None, O.I64)]
        ctx.set_option("no_census", 1)
        ctx.groupby_compute(keys, n, v, aggs)
        e0 = ctx.timings()["estimated_groups"]
        ctx.set_option("no_census", 0)
        ctx.groupby_compute(keys, n, v, aggs)
        assert ctx.timings()["estimated_groups"] == e0, g           # (the estimate is deterministic: same sample, no second stage)
    g = 240_000
    ids = np.where(rng.random(n) < 0.8, rng.integers(0, g // 10, n), rng.integers(0, g, n))    # 24 K keys with 80 % of the rows + 240 K
    # every key layout the census reads through key_cell / key_is_null: sparse i64 cells, string-pool codes behind a null mask
    # (the NULL group and the rows' nulls are not part of the slice), f64 cells
    for kd, kdata, kmask in ((O.I64, sparse_keys_from(ids), None), (O.U32CODE, ids.astype(np.uint32), O.pack_mask(rng.random(n) < 0.01)),
                             (O.F64, ids.astype(np.float64) * 0.25 - 1000.0, None)):
        keys = [(kdata, kmask, kd)]
        want = O.groupby_agg(keys, n, v, aggs)
        true_groups = want[0].shape[1]
        got = ctx.groupby_agg(keys, n, v, aggs)
        t = ctx.timings()
        assert_groupby_equal(got, want, [kd], int_exact_rows=[1, 2, 3])
        assert 0.85 * true_groups <= t["estimated_groups"] <= 1.15 * true_groups, (kd, t["estimated_groups"], true_groups)
        ctx.set_option("no_census", 1)
        try:
            ctx.groupby_compute(keys, n, v, aggs)
            assert ctx.timings()["estimated_groups"] < 0.6 * true_groups
        finally:
            ctx.set_option("no_census", 0)


def test_gpu_config_is_honoured():
    """pandrs_hip_config mirrors GpuConfig (reference src/gpu/mod.rs:18-44): enabled / memory_limit."""
    import ctypes as C
    import pandrs_amd as pa
    from pandrs_amd import _lib as L
    lib = L.load()
    try:
        cfg = L.Config(enabled=0, device_id=0, memory_limit=0, fallback_to_cpu=1, use_pinned_memory=0, min_size_threshold=10_000)
        assert lib.pandrs_hip_init(C.byref(cfg)) == 0
        with pytest.raises(pa.PandrsHipError) as e:
            pa.Context(0)
        assert e.value.status == L.ERR_NOT_INITIALIZED
        cfg.enabled, cfg.memory_limit = 1, 8 << 20          # 8 MiB: far too small for 2 M rows
        assert lib.pandrs_hip_init(C.byref(cfg)) == 0
        c = pa.Context(0)
        rng = np.random.default_rng(0)
        with pytest.raises(pa.PandrsHipError) as e:
            c.groupby_agg([(rng.integers(0, 10**6, 2_000_000), None, O.I64)], 2_000_000,
                          [(rng.normal(size=2_000_000), None, O.F64)], [(0, O.SUM)])
        assert e.value.status == L.ERR_OUT_OF_MEMORY and "memory_limit" in str(e.value)
        c.close()
        # min_size_threshold: below it the frame-level calls answer "keep your CPU path" (status 7) and compute nothing;
        # use_pinned_memory: host columns are page-locked for the call, results unchanged
        cfg.memory_limit, cfg.min_size_threshold, cfg.use_pinned_memory = 0, 10_000, 1
        assert lib.pandrs_hip_init(C.byref(cfg)) == 0
        c = pa.Context(0)
        small = (np.arange(100, dtype=np.int64), None, O.I64)
        for call in (lambda: c.groupby_agg([small], 100, [(np.ones(100), None, O.F64)], [(0, O.SUM)]),
                     lambda: c.join_indices(small, 100, small, 100, O.INNER),
                     lambda: c.column_stats((np.ones(100), None, O.F64), 100),
                     lambda: c.groupby_indices([small], 100)):
            with pytest.raises(pa.BelowThreshold) as e:
                call()
            assert e.value.status == L.ERR_BELOW_THRESHOLD
        n = 3_000_000
        k = (rng.integers(0, 1000, n), None, O.I64)
        v = (rng.normal(100, 10, n), None, O.F64)
        got = c.groupby_agg([k], n, [v], FIVE)
        assert_groupby_equal(got, O.groupby_agg([k], n, [v], FIVE), [O.I64], int_exact_rows=EXACT5)
        c.close()
    finally:
        lib.pandrs_hip_init(None)    # back to the library defaults (no explicit config: no threshold)
        pa.Context(0).close()        # resets the limit


# ---- group_by's own result: row -> group assignment (G1, grouping.rs:22-115) -------------------------
def check_group_indices(ctx, keys, n):
    """Complete characterisation of the reference's HashMap<key, Vec<usize>>: the rows are a
    permutation of 0..n, every group's rows are ascending and all carry the group's key, and there
    are exactly as many groups as distinct keys."""
    from oracle import oracle_np as ONP
    cells, nulls, off, rows = ctx.groupby_indices(keys, n)
    g = cells.shape[1]
    assert off[0] == 0 and off[-1] == n and np.all(np.diff(off) > 0)
    np.testing.assert_array_equal(np.sort(rows), np.arange(n))
    gid = np.repeat(np.arange(g), np.diff(off))
    inner = np.ones(n, bool)
    inner[off[:-1]] = False                                  # first row of every group
    assert np.all(np.diff(rows)[inner[1:]] > 0), "rows of a group must ascend"
    comp = []
    for k, col in enumerate(keys):
        nul, cell = ONP.key_cells(col, n)
        np.testing.assert_array_equal(nulls[k][gid], nul[rows])
        np.testing.assert_array_equal(np.where(nulls[k][gid] == 1, 0, cells[k][gid]), cell[rows])
        comp += [nul.astype(np.uint64), cell]
    assert len(np.unique(np.stack(comp, 1), axis=0)) == g
    return cells, nulls, off, rows


def test_group_indices_small_matches_reference_semantics(ctx, golden):
    from oracle import oracle_np as ONP
    case = golden["groupby"][0]
    codes, pool = codes_of(case["key_strings"])                # tests/groupby_test.rs:18-82: sizes A 2, B 2, C 1
    cells, nulls, off, rows = ctx.groupby_indices([(codes, None, O.U32CODE)], len(codes))
    want = ONP.group_indices([(codes, None, O.U32CODE)], len(codes), pools=[pool])
    got = {(pool[int(cells[0, g])],): rows[off[g]:off[g + 1]].tolist() for g in range(cells.shape[1])}
    assert got == want
    k = (np.array([5, -1, 5, 0, -1, 7, 0], np.int64), O.pack_mask([0, 0, 0, 1, 0, 0, 1]), O.I64)
    cells, nulls, off, rows = check_group_indices(ctx, [k], 7)
    got = {("NULL" if nulls[0, g] else str(int(np.int64(cells[0, g]))),): rows[off[g]:off[g + 1]].tolist() for g in range(cells.shape[1])}
    assert got == ONP.group_indices([k], 7) == {("5",): [0, 2], ("-1",): [1, 4], ("NULL",): [3, 6], ("7",): [5]}
    check_group_indices(ctx, [(np.zeros(0, np.int64), None, O.I64)], 0)


@pytest.mark.parametrize("n,g,kd", [(300_000, 2_000, O.I64), (2_000_000, 700_000, O.I64), (1_500_000, 40, O.F64),
                                    (1_000_000, 3, O.U32CODE), (500_000, 2, O.BOOLBITS)])
def test_group_indices_random(ctx, n, g, kd):
    rng = np.random.default_rng(n // 1000 + g)
    ids = rng.integers(0, g, n)
    if kd == O.I64:
        data = sparse_keys_from(ids)
    elif kd == O.F64:
        data = np.concatenate([rng.normal(size=g - 4), [0.0, -0.0, np.nan, np.inf]])[ids]
    elif kd == O.U32CODE:
        data = ids.astype(np.uint32)
    else:
        data = np.packbits(ids % 2 == 0, bitorder="little")
    check_group_indices(ctx, [(data, O.pack_mask(rng.random(n) < 0.01), kd)], n)


def test_group_indices_multi_key(ctx):
    rng = np.random.default_rng(31)
    n = 400_000
    k0 = (rng.integers(0, 30, n).astype(np.uint32), None, O.U32CODE)
    k1 = (rng.integers(-5, 5, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.05), O.I64)
    check_group_indices(ctx, [k0, k1], n)


def test_multi_key_wider_than_64_bits_is_dictionary_encoded(ctx):
    """Two hashed i64 ids (each spans the full 64-bit range) + a small nullable key: 129+ bits of codes.
    The widest columns get dense per-column group ids (<= 32 bits) until the packed cell fits."""
    rng = np.random.default_rng(2024)
    n = 600_000
    a = sparse_keys_from(rng.integers(0, 3_000, n))
    b = sparse_keys_from(rng.integers(0, 700, n) + 10_000)
    k0 = (a, O.pack_mask(rng.random(n) < 0.01), O.I64)
    k1 = (b, None, O.I64)
    k2 = (rng.integers(0, 3, n).astype(np.uint32), O.pack_mask(rng.random(n) < 0.1), O.U32CODE)
    v = (rng.normal(1, 2, n), None, O.F64)
    a[rng.random(n) < 0.01] = -1                              # the table-sentinel cell as a key value
    f = (rng.choice(np.array([0.0, -0.0, np.nan, 1e300, -1e300, 5e-324]), n), None, O.F64)   # full-range f64 key
    # the codes come from a list of the column's distinct cells + a hashed look-up per row, or (any cardinality) from ordering its rows
    for sorted_dictionary in (0, 1):
        ctx.set_option("sorted_dictionary", sorted_dictionary)
        try:
            check(ctx, [k0, k1, k2], n, [v], [(0, O.SUM), (0, O.MIN), (0, O.COUNT), (0, O.MEDIAN)], [O.I64, O.I64, O.U32CODE], exact=[1, 2, 3])
            check_group_indices(ctx, [k0, k1], n)
            check(ctx, [f, k1], n, [v], [(0, O.MAX), (0, O.COUNT)], [O.F64, O.I64], exact=[0, 1])
        finally:
            ctx.set_option("sorted_dictionary", 0)


def test_wide_low_cardinality_key_column_of_a_composite_key(ctx):
    """Three keys whose f64 column (0.0 / 1.0 / 2.0: 63 bits of span) pushes the codes past 64 bits: its dictionary is the engine's
    own list of the column's distinct cells (ordering 4 M rows into three groups was the slow way), and a column with
    many distinct cells (900 K hashed ids) takes the same route."""
    rng = np.random.default_rng(2025)
    n = 4_000_000
    ids = rng.integers(0, 60_000, n)
    k0 = ((ids // 21) * 11, None, O.I64)
    k1 = ((ids % 7).astype(np.uint32), None, O.U32CODE)
    k2 = (((ids // 7) % 3).astype(np.float64), O.pack_mask(rng.random(n) < 0.001), O.F64)
    v = (rng.normal(1, 2, n), None, O.F64)
    check(ctx, [k0, k1, k2], n, [v], [(0, O.SUM), (0, O.MAX), (0, O.COUNT)], [O.I64, O.U32CODE, O.F64], exact=[1, 2])
    big = (sparse_keys_from(rng.integers(0, 900_000, n)), None, O.I64)
    big2 = (sparse_keys_from(rng.integers(0, 5, n) + 7), O.pack_mask(rng.random(n) < 0.01), O.I64)
    check(ctx, [big, big2], n, [v], [(0, O.MIN), (0, O.COUNT)], [O.I64, O.I64], exact=[0, 1])


def test_column_population_std_like_parallel_std(ctx):
    """jit/parallel.rs:374-380: std of [1,2,3,4,5] = 1.4142135623730951 (population); :354-372 sum 1..1000 / mean."""
    std, var = ctx.column_std((np.array([1.0, 2.0, 3.0, 4.0, 5.0]), None, O.F64), 5)
    assert abs(std - 1.4142135623730951) < 1e-10 and abs(var - 2.0) < 1e-10
    assert ctx.column_std((np.array([7.0]), None, O.F64), 1) == (0.0, 0.0)
    rng = np.random.default_rng(1)
    n = 2_000_003
    x = rng.normal(3, 2, n)
    m = rng.random(n) < 0.1
    std, _ = ctx.column_std((x, O.pack_mask(m), O.F64), n)
    assert std == pytest.approx(np.std(x[~m]), rel=1e-9)
    xi = rng.integers(-1000, 1000, n).astype(np.int64)
    std, _ = ctx.column_std((xi, None, O.I64), n)
    assert std == pytest.approx(np.std(xi.astype(np.float64)), rel=1e-9)


@pytest.mark.parametrize("layout", ["sorted", "runs", "sorted_masked_i64"])
def test_clustered_rows_fold_runs_inside_the_wave(ctx, layout):
    """Rows sorted / grouped by key: the estimator's adjacency signal switches the aggregate to the
    run-folding kernels (segmented scan over the lanes, one table update per run)."""
    rng = np.random.default_rng({"sorted": 1, "runs": 2, "sorted_masked_i64": 3}[layout])
    n, g = 3_000_000, 40_000
    ids = rng.integers(0, g, n)
    if layout == "runs":
        ids = np.repeat(rng.integers(0, g, n // 50 + 1), rng.integers(1, 100, n // 50 + 1))[:n]
        n = len(ids)
    else:
        ids = np.sort(ids)
    k = (sparse_keys_from(ids), O.pack_mask(rng.random(n) < 0.001), O.I64)
    if layout == "sorted_masked_i64":
        v = [(rng.integers(-10**9, 10**9, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.2), O.I64)]
        aggs, exact = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (0, O.MAX), (0, O.COUNT)], [0, 2, 3, 4]
    else:
        pool = np.concatenate([rng.normal(size=1000), [np.nan, np.inf, -np.inf, -0.0]])
        v = [(pool[rng.integers(0, len(pool), n)], None, O.F64), (rng.normal(5, 1, n), None, O.F64)]
        aggs, exact = [(c, op) for c in range(2) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)], [2, 3, 6, 7, 8]
    check(ctx, k, n, v, aggs, [O.I64], exact=exact)
    ctx.set_option("no_runs", 1)
    try:
        check(ctx, k, n, v, aggs, [O.I64], exact=exact)
    finally:
        ctx.set_option("no_runs", 0)


@pytest.mark.parametrize("ncols,ops", [(12, (O.SUM, O.MEAN, O.MIN, O.MAX)), (16, (O.SUM, O.MAX)), (7, (O.STD, O.MIN, O.MEDIAN))])
def test_many_value_columns_take_several_rounds(ctx, ncols, ops):
    """Wide aggregations: the states of all columns do not fit one LDS table, so the aggregate processes
    the columns in rounds (key table kept, state arrays reused); mixed dtypes and masks across columns."""
    rng = np.random.default_rng(ncols)
    n, g = 700_000, 30_000
    k = (sparse_keys_from(rng.integers(0, g, n)), O.pack_mask(rng.random(n) < 0.001), O.I64)
    vals = []
    for c in range(ncols):
        if c % 3 == 2:
            vals.append((rng.integers(-10**6, 10**6, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.1) if c % 2 else None, O.I64))
        else:
            vals.append((rng.normal(c, 1 + c, n), O.pack_mask(rng.random(n) < 0.05) if c % 4 == 0 else None, O.F64))
    aggs = [(c, op) for c in range(ncols) for op in ops]
    exact = [i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.MEDIAN) or (op == O.SUM and vals[c][2] == O.I64)]
    check(ctx, k, n, vals, aggs, [O.I64], exact=exact)


def test_chunked_groupby_beyond_one_call(ctx):
    """Rows processed in chunks (the way a caller goes past the 2^32-row per-call limit or bounds the
    workspace): partial states per chunk + one merge == one call over everything."""
    rng = np.random.default_rng(64)
    n, g = 1_000_003, 20_000
    k = (sparse_keys_from(rng.integers(0, g, n)), O.pack_mask(rng.random(n) < 0.001), O.I64)
    v0 = (rng.normal(40, 5, n), O.pack_mask(rng.random(n) < 0.1), O.F64)
    v1 = (rng.integers(-100, 100, n).astype(np.int64), None, O.I64)
    aggs = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (1, O.SUM), (1, O.MAX), (1, O.COUNT)]
    got = ctx.groupby_agg_chunked([k], n, [v0, v1], aggs, chunk_rows=131_072)
    want = O.groupby_agg([k], n, [v0, v1], aggs)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[2, 3, 4, 5])
    with pytest.raises(ValueError):
        ctx.groupby_agg_chunked([k], n, [v0], [(0, O.SUM)], chunk_rows=1001)


def test_device_resident_inputs_for_the_newer_entry_points(ctx):
    """Median, group indices and multi-key dictionary encoding with DEVICE pointers (torch tensors):
    same results as with host pointers."""
    import torch
    rng = np.random.default_rng(808)
    n = 400_000
    k0 = sparse_keys_from(rng.integers(0, 3000, n))
    m0 = O.pack_mask(rng.random(n) < 0.01)
    k1 = rng.integers(-2**62, 2**62, n)
    v = rng.normal(0, 1, n)
    vm = O.pack_mask(rng.random(n) < 0.1)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    aggs = [(0, O.MEDIAN), (0, O.SUM), (0, O.COUNT)]
    got = ctx.groupby_agg([(dev(k0), dev(m0), O.I64)], n, [(dev(v), dev(vm), O.F64)], aggs)
    got = tuple(x.cpu().numpy() for x in got)
    want = O.groupby_agg([(k0, m0, O.I64)], n, [(v, vm, O.F64)], aggs)
    assert_groupby_equal((got[0].view(np.uint64), got[1], got[2]), want, [O.I64], int_exact_rows=[0, 2])
    cells, nulls, off, rows = ctx.groupby_indices([(dev(k0), dev(m0), O.I64), (dev(k1), None, O.I64)], n)
    h = ctx.groupby_indices([(k0, m0, O.I64), (k1, None, O.I64)], n)
    assert cells.shape == h[0].shape and int(off[-1]) == n
    a = np.lexsort((cells.cpu().numpy()[1], cells.cpu().numpy()[0], nulls.cpu().numpy()[0]))
    b = np.lexsort((h[0].view(np.int64)[1], h[0].view(np.int64)[0], h[1][0]))
    np.testing.assert_array_equal(cells.cpu().numpy()[:, a], h[0].view(np.int64)[:, b])
    np.testing.assert_array_equal(np.diff(off.cpu().numpy())[a], np.diff(h[2])[b])


# ---- round 2: the sampled-capacity partition (no histogram pass) and the lean aggregate kernel ----------
def _took_sampled_partition(t):
    """The capacity-mode partition has no scan phase (its regions come from a sample); the exact one does."""
    return t["n_partitions"] > 0 and "scan" not in t["phase_ms"] and "scatter" in t["phase_ms"]


@pytest.mark.parametrize("nulls", [False, True])
def test_sampled_partition_matches_oracle(ctx, nulls):
    """20 M rows / 600 K groups: enough rows per (partition, XCD group) region for the sampled-capacity
    partition.  Same answers as the oracle, and as the exact-histogram path."""
    rng = np.random.default_rng(4242 + nulls)
    n, g = 20_000_000, 600_000
    k = sparse_keys(rng, n, g)
    k[rng.random(n) < 0.001] = -1                                 # the table-sentinel value among the keys
    keys = [(k, O.pack_mask(rng.random(n) < 0.01) if nulls else None, O.I64)]
    vals = [(rng.normal(100, 10, n), O.pack_mask(rng.random(n) < 0.05) if nulls else None, O.F64),
            (rng.normal(-3, 1, n), O.pack_mask(rng.random(n) < 0.05) if nulls else None, O.F64)]
    aggs = [(c, op) for c in range(2) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
    got = ctx.groupby_agg(keys, n, vals, aggs)
    assert _took_sampled_partition(ctx.timings()), ctx.timings()
    want = O.groupby_agg(keys, n, vals, aggs)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[2, 3, 6, 7, 8])
    ctx.set_option("exact_partition", 1)
    try:
        got2 = ctx.groupby_agg(keys, n, vals, aggs)
        assert not _took_sampled_partition(ctx.timings())
    finally:
        ctx.set_option("exact_partition", 0)
    assert_groupby_equal(got2, want, [O.I64], int_exact_rows=[2, 3, 6, 7, 8])


def test_sampled_partition_falls_back_when_the_sample_misleads(ctx):
    """Keys whose set depends on the tile index (mod 8) AND on a slow drift: the 1-in-8 block sample and the
    1/8 split over the XCD groups both mislead, some region overflows, the scatter drops the run, and the call
    must come back exact through the histogram path (retried inside the library)."""
    rng = np.random.default_rng(99)
    n, g = 24_000_000, 500_000
    tile = np.arange(n) // 8192
    ids = rng.integers(0, g // 16, n) + (tile % 8) * (g // 16)    # group g of tiles only ever sees its own 1/8 of half the keys
    ids[n // 2:] += g // 2                                       # and the second half of the table uses the other half
    keys = [(sparse_keys_from(ids), None, O.I64)]
    vals = [(rng.normal(0, 1, n), None, O.F64)]
    check(ctx, keys, n, vals, FIVE, [O.I64], exact=EXACT5)
    assert "scan" in ctx.timings()["phase_ms"], "the exact histogram path did not run: the input no longer defeats the sample"


@pytest.mark.parametrize("ncols,ops", [(1, (O.SUM,)), (2, (O.MIN, O.MAX)), (3, (O.SUM, O.MIN, O.MAX)), (4, (O.SUM, O.MEAN, O.MIN, O.MAX))])
def test_lean_aggregate_kernel_agrees_with_the_generic_one(ctx, ncols, ops):
    """aggregate2 (Swiss-table lookup, retry queue, negated max states) against the round-1 kernel (agg_v1) and
    the oracle, with NaN / +-inf / -0.0 values, int64 extremes, a hot key and the sentinel key."""
    rng = np.random.default_rng(7 + ncols)
    n, g = 3_000_000, 200_000
    k = sparse_keys(rng, n, g)
    k[rng.random(n) < 0.2] = 777                                  # hot key -> sliced partition (multi tables)
    k[rng.random(n) < 0.01] = -1
    keys = [(k, O.pack_mask(rng.random(n) < 0.01), O.I64)]
    def col(i):
        v = rng.normal(0, 1e3, n)
        v[rng.random(n) < 0.01] = np.nan
        v[rng.random(n) < 0.005] = np.inf
        v[rng.random(n) < 0.005] = -np.inf
        v[rng.random(n) < 0.01] = -0.0 if i % 2 else 0.0
        return (v, O.pack_mask(rng.random(n) < 0.03) if i % 2 else None, O.F64)
    for kind in ("f64", "i64"):
        if kind == "f64":
            vals = [col(i) for i in range(ncols)]
        else:
            vals = [(rng.integers(-2**62, 2**62, n).astype(np.int64) if i == 0 else rng.integers(-1000, 1000, n).astype(np.int64),
                     None, O.I64) for i in range(ncols)]
        aggs = [(c, op) for c in range(ncols) for op in ops] + [(0, O.COUNT)]
        exact = [i for i, (_, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT) or kind == "i64" and op == O.SUM]
        want = O.groupby_agg(keys, n, vals, aggs)
        got = ctx.groupby_agg(keys, n, vals, aggs)
        assert_groupby_equal(got, want, [O.I64], int_exact_rows=exact)
        ctx.set_option("agg_v1", 1)
        try:
            got1 = ctx.groupby_agg(keys, n, vals, aggs)
        finally:
            ctx.set_option("agg_v1", 0)
        assert_groupby_equal(got1, want, [O.I64], int_exact_rows=exact)


def _full_size_properties(ctx, keys_t, key_dtype, ids, n, vals, aggs_per_col, torch):
    """Size-independent checks for inputs too large for the oracle (see test_full_size_properties)."""
    import pandrs_amd as pa
    ncol = len(vals)
    aggs = [(c, op) for c in range(ncol) for op in aggs_per_col] + [(0, O.COUNT)]
    ng = ctx.groupby_compute([(keys_t, None, key_dtype)], n, [(v, None, O.F64) for v in vals], aggs)
    kc, kn, oa = ctx.groupby_fetch()
    assert ng == torch.unique(ids).numel()
    assert torch.unique(kc[0]).numel() == ng and int(kn.sum()) == 0
    cnt = oa[-1]
    assert float(cnt.sum()) == n
    w = len(aggs_per_col)
    for col in range(ncol):
        row = dict(zip(aggs_per_col, oa[w * col:w * col + w]))
        tot = float(vals[col].sum())
        if O.SUM in row:
            assert abs(float(row[O.SUM].sum()) - tot) <= 1e-9 * abs(tot)
        if O.MIN in row:
            assert float(row[O.MIN].min()) == float(vals[col].min()) and float(row[O.MAX].max()) == float(vals[col].max())
        if O.MEAN in row and O.SUM in row:
            assert torch.allclose(row[O.MEAN] * cnt, row[O.SUM], rtol=1e-12, atol=0)
    return ng, kc, oa


def test_config3_full_size_properties():
    """BASELINE C3 at full size: 100 M rows, u32 string-pool codes, 10 K groups with 80/20 skew, 2 f64 columns x
    sum/mean/min/max + count.  Property run (group count, sum of counts, linearity of the sums, global extremes,
    mean x count = sum); the same shape is oracle-checked at 17 M rows in test_mid_cardinality_*."""
    import torch
    import pandrs_amd as pa
    n, g, d = 100_000_000, 10_000, "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(45)
    hot = torch.rand(n, device=d, generator=gen) < 0.8
    ids = torch.where(hot, torch.randint(0, g // 5, (n,), device=d, generator=gen), torch.randint(0, g, (n,), device=d, generator=gen))
    del hot
    codes = ids.to(torch.int32)
    vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(2)]
    c = pa.Context(0)
    try:
        _full_size_properties(c, codes, O.U32CODE, ids, n, vals, (O.SUM, O.MEAN, O.MIN, O.MAX), torch)
    finally:
        c.close()


def test_config4_shard_full_size_properties_and_scaled_oracle():
    """BASELINE C4's per-GPU shard at full size: 125 M rows, sparse i64 key, 10 M groups, sum + count.
    Property run at full size; the same plan (fan-out forced to the full-size one) against the oracle at 1/50 scale."""
    import torch
    import pandrs_amd as pa
    n, g, d = 125_000_000, 10_000_000, "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(46)
    ids = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64)
    keys = ids * -7046029254386353131 ^ 0x5555AAAA5555AAAA
    v = torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100
    c = pa.Context(0)
    try:
        aggs = [(0, O.SUM), (0, O.COUNT)]
        ng = c.groupby_compute([(keys, None, O.I64)], n, [(v, None, O.F64)], aggs)
        t_full = c.timings()
        kc, kn, oa = c.groupby_fetch()
        assert ng == torch.unique(ids).numel()
        assert torch.unique(kc[0]).numel() == ng and int(kn.sum()) == 0
        assert float(oa[1].sum()) == n
        tot = float(v.sum())
        assert abs(float(oa[0].sum()) - tot) <= 1e-9 * abs(tot)
        del ids, keys, v, kc, kn, oa
        torch.cuda.empty_cache()
        # 1/50 scale on the same plan: same fan-out, so the same kernels and table load per partition
        rng = np.random.default_rng(47)
        n2, g2 = n // 50, g // 50
        k2 = sparse_keys(rng, n2, g2)
        v2 = rng.normal(100, 10, n2)
        c.set_option("partitions", int(t_full["n_partitions"]))
        try:
            got = c.groupby_agg([(k2, None, O.I64)], n2, [(v2, None, O.F64)], aggs)
            assert c.timings()["n_partitions"] == t_full["n_partitions"]
        finally:
            c.set_option("partitions", 0)
        want = O.groupby_agg([(k2, None, O.I64)], n2, [(v2, None, O.F64)], aggs)
        assert_groupby_equal(got, want, [O.I64], int_exact_rows=[1])
    finally:
        c.close()



def test_config4_full_size_on_one_gpu():
    """BASELINE C4 WHOLE (1 B rows, sparse i64 key, 10 M groups, sum + count) on one MI355X: 16 GB of input fits the
    288 GB several times over.  Once through ONE call (N < 2^32) and once through groupby_agg_chunked (4 x 250 M rows:
    partial states per chunk + one merge), both checked through size-independent properties and against each other."""
    import torch
    import pandrs_amd as pa
    n, g, d = 1_000_000_000, 10_000_000, "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(48)
    ids = torch.randint(0, g, (n,), device=d, generator=gen, dtype=torch.int64)
    keys = ids * -7046029254386353131 ^ 0x5555AAAA5555AAAA
    v = torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100
    c = pa.Context(0)
    try:
        aggs = [(0, O.SUM), (0, O.COUNT)]
        ng = c.groupby_compute([(keys, None, O.I64)], n, [(v, None, O.F64)], aggs)
        t = c.timings()
        print("C4 on one GPU: %.2f ms, %d partitions, %d groups" % (t["total_ms"], t["n_partitions"], ng))
        kc, kn, oa = c.groupby_fetch()
        assert ng == g                                   # 100 rows per group on average: every id is drawn
        assert torch.unique(kc[0]).numel() == ng and int(kn.sum()) == 0
        assert float(oa[1].sum()) == n
        tot = float(v.sum())
        assert abs(float(oa[0].sum()) - tot) <= 1e-9 * abs(tot)
        # a handful of groups against a direct masked reduction over the 1 B rows
        order = torch.argsort(kc[0])
        skeys, ssum, scnt = kc[0][order], oa[0][order], oa[1][order]
        for gid in (0, 1, 4_999_999, 9_999_999, 1234567):
            key = gid * -7046029254386353131 ^ 0x5555AAAA5555AAAA
            key = (key + 2**63) % 2**64 - 2**63
            m = ids == gid
            want_n, want_s = int(m.sum()), float(v[m].sum())
            pos = int(torch.searchsorted(skeys, torch.tensor([key], device=d, dtype=torch.int64))[0])
            assert int(skeys[pos]) == key and float(scnt[pos]) == want_n
            assert abs(float(ssum[pos]) - want_s) <= 1e-9 * max(1.0, abs(want_s))
        del ids, m, kc, kn, oa, order
        torch.cuda.empty_cache()
        kc2, kn2, oa2 = c.groupby_agg_chunked([(keys, None, O.I64)], n, [(v, None, O.F64)], aggs, 250_000_000)
        assert kc2.shape[1] == g and int(kn2.sum()) == 0
        order2 = torch.argsort(kc2[0])
        assert torch.equal(kc2[0][order2], skeys)
        assert torch.equal(oa2[1][order2], scnt)                                   # counts exact
        rel = ((oa2[0][order2] - ssum).abs() / ssum.abs().clamp_min(1.0)).max()
        assert float(rel) <= 1e-9
    finally:
        c.close()


# ---- K1: the three reference families from one device pass (pandrs_hip_reduce_stats) -----------------------
def _k1_check(ctx, col, n):
    st = ctx.column_stats(col, n)
    k = O.k1_stats(col, n)
    f64 = col[2] == O.F64
    assert st["count"] == (0 if k.a_empty else st["count"])
    # (A) frame level
    if k.a_empty:
        assert st["count"] == 0 and st["sum_f64"] == 0.0
    else:
        if math.isnan(k.a_sum) or math.isinf(k.a_sum):
            assert str(st["sum_f64"]) == str(k.a_sum)
        else:
            assert st["sum_f64"] == pytest.approx(k.a_sum, rel=1e-9, abs=1e-9 * abs(k.a_sum) + 1e-300)
        assert st["min"] == k.a_min and st["max"] == k.a_max
    # (B) column level
    if f64:
        assert (st["count_finite"] == 0) == bool(k.b_minmax_none) or n == 0
        if not k.b_minmax_none:
            assert st["min_finite"] == k.b_min and st["max_finite"] == k.b_max
    else:
        assert st["sum_i64"] == k.b_sum_i64
        if not k.b_minmax_none:
            assert st["min_i64"] == k.b_min_i64 and st["max_i64"] == k.b_max_i64
    return st, k


def test_k1_statistics_match_the_three_reference_families(ctx):
    import math
    rng = np.random.default_rng(11)
    for n in (0, 1, 2, 7, 1000, 1_000_003):
        v = rng.normal(5, 2, n)
        if n >= 7:
            v[rng.integers(0, n, max(n // 50, 1))] = np.inf
            v[rng.integers(0, n, max(n // 50, 1))] = -np.inf
            v[rng.integers(0, n, max(n // 50, 1))] = np.nan
            v[rng.integers(0, n, 2)] = [0.0, -0.0]
        for mask in (None, O.pack_mask(rng.random(n) < 0.2) if n else None):
            _k1_check(ctx, (v, mask, O.F64), n)
        q = rng.integers(-2**62, 2**62, n).astype(np.int64)
        for mask in (None, O.pack_mask(rng.random(n) < 0.2) if n else None):
            _k1_check(ctx, (q, mask, O.I64), n)
    # only non-finite values: column min / max are None, the frame keeps the infinities
    st, k = _k1_check(ctx, (np.array([np.inf, np.nan, -np.inf]), None, O.F64), 3)
    assert st["count_finite"] == 0 and st["min"] == -np.inf and st["max"] == np.inf
    # all null
    st, k = _k1_check(ctx, (np.array([1.0, 2.0, 3.0]), O.pack_mask([1, 1, 1]), O.F64), 3)
    assert st["count"] == 0
    # a column that starts 8 bytes off a 16-byte boundary (sliced buffer), with a mask
    import torch
    base = torch.arange(0, 100_001, dtype=torch.float64, device="cuda:0")
    view = base[1:]
    st = ctx.column_stats((view, None, O.F64), 100_000)
    assert st["sum_f64"] == 100_000 * 100_001 / 2 and st["min"] == 1.0 and st["max"] == 100_000.0


def test_k1_mirrors_follow_the_reference(ctx):
    """frame level: Err(Empty) / Error::Type / `v as f64` sums (aggregate.rs:21-215); column level: None where the
    reference returns None, non-finite values skipped; slice level: simd_mean_i64's integer division."""
    import pandrs_amd as pa
    from pandrs_amd import frame as F, simd as S
    df = F.OptimizedDataFrame()
    df.add_column("x", F.Float64Column([1.0, np.inf, -2.0, np.nan], [False, False, False, False]))
    df.add_column("e", F.Float64Column([1.0, 2.0, 3.0, 4.0], [True, True, True, True]))
    df.add_column("q", F.Int64Column([2**62, 2**62, 2**62, 1]))
    df.add_column("s", F.StringColumn(["a", "b", "c", "d"]))
    assert df.max("x") == np.inf and df.min("x") == -2.0 and math.isnan(df.sum("x"))
    assert df.sum("e") == 0.0
    for fn in (df.mean, df.min, df.max):
        with pytest.raises(pa.EmptyError):
            fn("e")
    with pytest.raises(pa.ColumnTypeMismatch):
        df.sum("s")
    assert df.sum("q") == 3.0 * 2.0**62 + 1.0                       # no i64 wrap at frame level
    x, e, q = df.column("x"), df.column("e"), df.column("q")
    assert (x.min(), x.max()) == (-2.0, 1.0) and e.mean() is None and e.min() is None and e.max() is None
    assert q.sum() == int(np.int64(np.uint64((3 * 2**62 + 1) % 2**64).astype(np.int64)))
    assert F.Float64Column([]).mean() is None and F.Int64Column([]).min() is None and F.Float64Column([]).sum() == 0.0
    assert S.simd_mean_i64(np.array([-7, 2, 2], np.int64)) == -1 and S.simd_mean_i64(np.array([], np.int64)) == 0
    assert S.simd_min_i64(np.array([], np.int64)) == 2**63 - 1 and S.simd_max_f64(np.array([], np.float64)) == -np.inf
    assert S.simd_sum_f64(np.arange(1.0, 9.0)) == 36.0 and S.simd_mean_f64(np.array([1.0, 2, 3, 4, 5])) == 3.0


def test_direct_aggregation_known_answers_on_the_device(ctx, golden):
    """/root/reference/src/optimized/direct_aggregations.rs:363-593 on the device path: sum_direct / mean_direct / max_direct /
    min_direct are the column folds (Float64Column / Int64Column::{sum, mean, min, max}), the *_simd twins the slice folds; both
    mirrors read the one-pass pandrs_hip_reduce_stats.  Values exact on the 5-element frame, within 1e-10 on the 10 000-element one."""
    from pandrs_amd import frame as F, simd as S
    c = golden["direct_aggregations"]
    for name, col, mk in (("float_col", F.Float64Column(c["float_col"]), np.float64), ("int_col", F.Int64Column(c["int_col"]), np.int64)):
        e = c["expect"][name]
        assert float(col.sum()) == e["sum"] and col.mean() == e["mean"] and float(col.max()) == e["max"] and float(col.min()) == e["min"]
        assert col.len() == e["count"]
        data = np.array(c[name], mk)
        sfx = "f64" if mk is np.float64 else "i64"
        assert float(getattr(S, "simd_sum_" + sfx)(data)) == e["sum"] and float(getattr(S, "simd_mean_" + sfx)(data)) == e["mean"]
        assert float(getattr(S, "simd_max_" + sfx)(data)) == e["max"] and float(getattr(S, "simd_min_" + sfx)(data)) == e["min"]
    big = c["large"]
    i = np.arange(1, big["n"] + 1)
    fl, it = i.astype(np.float64) * big["float_scale"], (i * big["int_scale"]).astype(np.int64)
    seq = 0.0
    for x in fl.tolist():
        seq += x
    assert abs(S.simd_sum_f64(fl) - seq) < big["abs_tolerance"] and abs(S.simd_mean_f64(fl) - seq / len(fl)) < big["abs_tolerance"]
    assert abs(F.Float64Column(fl).sum() - seq) < big["abs_tolerance"]
    assert float(S.simd_max_i64(it)) == float(F.Int64Column(it).max()) == big["int_max"]
    assert float(S.simd_min_i64(it)) == float(F.Int64Column(it).min()) == big["int_min"]


def test_k1_stream_rate():
    """100 M f64 (0.8 GB): the rewritten reduce kernel must stream at >= 4 TB/s (round 1: 2.6 TB/s)."""
    import torch
    import pandrs_amd as pa
    c = pa.Context(0)
    try:
        x = torch.randn(100_000_000, dtype=torch.float64, device="cuda:0")
        best = 1e9
        for _ in range(5):
            st = c.column_stats((x, None, O.F64), x.numel())
            best = min(best, c.timings()["phase_ms"]["other"])
        assert st["count"] == x.numel() and st["min"] == float(x.min()) and st["max"] == float(x.max())
        assert abs(st["sum_f64"] - float(x.sum())) <= 1e-9 * abs(float(x.abs().sum()))
        assert 0.8 / best >= 4.0, "reduce kernel %.3f ms = %.2f TB/s" % (best, 0.8 / best)
    finally:
        c.close()


def test_deterministic_mode_is_bit_identical_to_the_sequential_fold(ctx):
    """"deterministic" = 1: f64 Sum / Mean and Std / Var are folded in ascending row order per group, the reference's
    own loop (aggregation.rs:625-648, :881-903) — EVERY aggregate then equals the oracle's sequential fold bit for bit
    (default mode: sums within 1e-9 only, and not the same from run to run)."""
    rng = np.random.default_rng(31)
    n, g = 3_000_000, 120_000
    k = sparse_keys(rng, n, g)
    k[rng.random(n) < 0.05] = 4242                                # a hot key
    k[rng.random(n) < 0.002] = -1                                 # the table-sentinel value
    keys = [(k, O.pack_mask(rng.random(n) < 0.01), O.I64)]
    vals = [(rng.normal(0, 1e6, n) * rng.choice([1e-9, 1.0, 1e9], n), O.pack_mask(rng.random(n) < 0.05), O.F64),
            (rng.integers(-10**12, 10**12, n).astype(np.int64), None, O.I64),
            (rng.normal(3, 2, n), None, O.F64)]
    aggs = [(0, O.SUM), (0, O.MEAN), (0, O.STD), (0, O.VAR), (0, O.MIN), (0, O.MAX), (0, O.COUNT),
            (1, O.SUM), (1, O.MEAN), (1, O.STD), (1, O.VAR), (2, O.SUM), (2, O.MEAN)]
    want = O.groupby_agg(keys, n, vals, aggs)
    ctx.set_option("deterministic", 1)
    try:
        got = ctx.groupby_agg(keys, n, vals, aggs)
        assert_groupby_equal(got, want, [O.I64], int_exact_rows=list(range(len(aggs))))
        got2 = ctx.groupby_agg(keys, n, vals, aggs)                # and reproducible
        assert_groupby_equal(got2, got, [O.I64], int_exact_rows=list(range(len(aggs))))
        # two key columns (packed cells) and string-pool codes
        k2 = rng.integers(0, 7, n).astype(np.uint32)
        keys2 = [(k, None, O.I64), (k2, None, O.U32CODE)]
        got3 = ctx.groupby_agg(keys2, n, vals[:1], [(0, O.SUM), (0, O.MEAN)])
        want3 = O.groupby_agg(keys2, n, vals[:1], [(0, O.SUM), (0, O.MEAN)])
        assert_groupby_equal(got3, want3, [O.I64, O.U32CODE], int_exact_rows=[0, 1])
    finally:
        ctx.set_option("deterministic", 0)


@pytest.mark.parametrize("shape", ["c1", "nulls", "i64", "two_cols", "tiny"])
def test_small_call_path_matches_oracle_and_general_path(ctx, shape):
    """calls of <= 2 M rows take two launches (fold into LDS tables + a context-owned global table,
    then the output kernel).  Same results as the oracle and as the general path; n_partitions == 0 proves it ran."""
    rng = np.random.default_rng(77)
    n, g = (1_000_000, 1_000) if shape != "tiny" else (777, 13)
    kv = sparse_keys(rng, n, g)
    kv[rng.integers(0, n, 5)] = -1                      # the table sentinel's bit pattern as a key
    km = O.pack_mask(rng.random(n) < 0.01) if shape in ("nulls", "tiny") else None
    keys = [(kv, km, O.I64)]
    if shape == "i64":
        vals = [(rng.integers(-10**9, 10**9, n).astype(np.int64), None, O.I64)]
        aggs, exact = FIVE, range(5)
    elif shape == "two_cols":
        vals = [(rng.normal(100, 10, n), None, O.F64), (rng.normal(-5, 1, n), None, O.F64)]
        aggs = [(c, op) for c in (0, 1) for op in (O.SUM, O.MEAN, O.MIN, O.MAX, O.COUNT)]
        exact = [i for i, (_, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT)]
    elif shape in ("nulls", "tiny"):
        vals = [(rng.normal(100, 10, n), O.pack_mask(rng.random(n) < 0.2), O.F64)]
        aggs, exact = FIVE, EXACT5
    else:
        vals = [(rng.normal(100, 10, n), None, O.F64)]
        aggs, exact = [(0, O.SUM)], []
    ctx.set_option("no_small", 0)           # (also clears the back-off an earlier test's oversized call may have left)
    got = check(ctx, keys, n, vals, aggs, [O.I64], exact=exact)
    assert ctx.timings()["n_partitions"] == 0, "the small path did not run"
    ctx.set_option("no_small", 1)
    try:
        ref = check(ctx, keys, n, vals, aggs, [O.I64], exact=exact)
        assert ctx.timings()["n_partitions"] > 0 or n < 5000
    finally:
        ctx.set_option("no_small", 0)
    assert got[0].shape == ref[0].shape


def test_small_call_path_falls_back_when_groups_outgrow_its_tables(ctx):
    rng = np.random.default_rng(78)
    n = 1_500_000
    vals = [(rng.normal(100, 10, n), None, O.F64)]
    for g in (12_000, 200_000):          # 12 K: fits the global table but overflows an LDS table; 200 K: outgrows both
        ctx.set_option("no_small", 0)
        check(ctx, [(sparse_keys(rng, n, g), None, O.I64)], n, vals, FIVE, [O.I64], exact=EXACT5)
        assert ctx.timings()["n_partitions"] > 0
    # a call that did not fit makes the next ones skip the attempt ...
    check(ctx, [(sparse_keys(rng, 50_000, 10), None, O.I64)], 50_000, [(vals[0][0][:50_000], None, O.F64)], FIVE, [O.I64], exact=EXACT5)
    assert ctx.timings()["n_partitions"] > 0
    # ... and the armed global table is clean afterwards
    ctx.set_option("no_small", 0)
    check(ctx, [(sparse_keys(rng, 50_000, 10), None, O.I64)], 50_000, [(vals[0][0][:50_000], None, O.F64)], FIVE, [O.I64], exact=EXACT5)
    assert ctx.timings()["n_partitions"] == 0


# ---- hot-key absorb-and-spill in front of the radix path (absorb.hip) --------------------------------------------------
def _skewed_ids(rng, n, g, hot_share=0.8, hot_keys=None):
    """80 % of the rows on the first fifth of the keys (benches/enhanced_comprehensive_benchmark.rs:53-59)"""
    hot = rng.random(n) < hot_share
    return np.where(hot, rng.integers(0, hot_keys or max(g // 5, 1), n), rng.integers(0, g, n))


@pytest.mark.parametrize("shape", ["c3_codes", "i64_nulls", "i64_sum_only", "uniform_forced", "hot_set_long_tail"])
def test_absorb_and_spill_matches_oracle(ctx, shape):
    """One pass over the original columns folds the rows whose key found a slot in a workgroup's LDS table, the rest is
    spilled and goes through the radix path; both halves are merged.  Same answers as the oracle, bit for bit where
    the reference is (counts, min / max), 1e-9 on f64 sums."""
    rng = np.random.default_rng({"c3_codes": 1, "i64_nulls": 2, "i64_sum_only": 3, "uniform_forced": 4, "hot_set_long_tail": 5}[shape])
    n = 17_000_000
    if shape == "c3_codes":
        g = 10_000
        key = (_skewed_ids(rng, n, g).astype(np.uint32), None, O.U32CODE)
        vals = [(rng.standard_normal(n), None, O.F64), (rng.standard_normal(n) * 10 + 100, None, O.F64)]
        aggs = [(c, op) for c in range(2) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
        exact, kd = (2, 3, 6, 7, 8), O.U32CODE
    elif shape == "i64_nulls":
        g = 30_000
        ids = _skewed_ids(rng, n, g, 0.9, 1_000)
        k = sparse_keys_from(ids)
        k[::100_003] = -1                                        # the key equal to the table sentinel
        key = (k, O.pack_mask(rng.random(n) < 0.001), O.I64)      # and a NULL-key group
        v = rng.standard_normal(n)
        v[::50_021] = np.nan
        vals = [(v, O.pack_mask(rng.random(n) < 0.02), O.F64), (rng.standard_normal(n), O.pack_mask(rng.random(n) < 0.5), O.F64)]
        aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.MEAN), (1, O.SUM), (1, O.MIN), (1, O.MAX), (1, O.COUNT)]
        exact, kd = (1, 2, 5, 6, 7), O.I64
    elif shape == "i64_sum_only":
        g = 200_000
        key = (sparse_keys_from(_skewed_ids(rng, n, g, 0.95, 4_000)), None, O.I64)
        vals = [(rng.integers(-1000, 1000, n).astype(np.int64), None, O.I64)]
        aggs = [(0, O.SUM), (0, O.COUNT)]
        exact, kd = (0, 1), O.I64
    elif shape == "hot_set_long_tail":
        # 85 % of the rows on 800 keys in front of 2 M others: far more groups than the spill tables hold -> COMPACT spill (the rows the
        # tables do not take are closed up and go through the ordinary engine as partial states; one merge joins both halves)
        g = 2_000_000
        k = sparse_keys_from(_skewed_ids(rng, n, g, 0.85, 800))
        k[::100_003] = -1
        key = (k, O.pack_mask(rng.random(n) < 0.001), O.I64)
        v = rng.standard_normal(n)
        v[::50_021] = np.nan
        vals = [(v, O.pack_mask(rng.random(n) < 0.02), O.F64), (rng.standard_normal(n), O.pack_mask(rng.random(n) < 0.3), O.F64)]
        aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.MEAN), (1, O.SUM), (1, O.MIN), (1, O.MAX), (1, O.COUNT)]
        exact, kd = (1, 2, 5, 6, 7), O.I64
    else:
        g = 12_000                                                # uniform keys: almost everything spills (forced)
        key = (sparse_keys(rng, n, g), None, O.I64)
        vals = [(rng.standard_normal(n), None, O.F64)]
        aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX)]
        exact, kd = (1, 2), O.I64
    want = O.groupby_agg([key], n, vals, aggs)
    if shape == "uniform_forced":
        ctx.set_option("no_absorb", -1)
    try:
        got = ctx.groupby_agg([key], n, vals, aggs)
        t = ctx.timings()
    finally:
        ctx.set_option("no_absorb", 0)
    assert t["absorbed_rows"] > 0, t
    if shape == "hot_set_long_tail":
        assert t["n_partitions"] == -1, t                        # the compact spill
    if shape != "uniform_forced":
        assert t["absorbed_rows"] > 0.6 * n, t
    else:
        assert t["absorbed_rows"] < 0.5 * n, t
    assert_groupby_equal(got, want, [kd], int_exact_rows=exact)
    # the ordinary path gives the same groups (and does not absorb)
    ctx.set_option("no_absorb", 1)
    try:
        got2 = ctx.groupby_agg([key], n, vals, aggs)
        assert ctx.timings()["absorbed_rows"] == 0
    finally:
        ctx.set_option("no_absorb", 0)
    assert_groupby_equal(got2, want, [kd], int_exact_rows=exact)
    # ... and so do absorb tables that start EMPTY (first come, first served) instead of from the sample's hot keys; the
    # image is what keeps the cold keys out of the slots: it never absorbs less
    ctx.set_option("no_hot_image", 1)
    if shape == "uniform_forced":
        ctx.set_option("no_absorb", -1)
    try:
        got3 = ctx.groupby_agg([key], n, vals, aggs)
        t3 = ctx.timings()
    finally:
        ctx.set_option("no_hot_image", 0)
        ctx.set_option("no_absorb", 0)
    if shape == "hot_set_long_tail":
        assert t3["absorbed_rows"] == 0                      # (the compact spill needs the image: its tables take no other key)
    else:
        assert t3["absorbed_rows"] > 0
    assert_groupby_equal(got3, want, [kd], int_exact_rows=exact)
    if shape == "c3_codes":
        assert t["absorbed_rows"] > t3["absorbed_rows"] + 0.03 * n, (t["absorbed_rows"], t3["absorbed_rows"])


@pytest.mark.parametrize("case", ["sliced_tail", "underestimated_tail", "late_hot_key", "two_keys"])
def test_compact_spill_with_nested_runs_and_composite_keys(ctx, case):
    """The compact spill (a hot set in front of a long tail) groups the spilled rows as the call's RESULT and appends the absorbed hot
    groups behind them.  The tail's run may itself nest — the slice merge of an oversized partition, the overflow run of a tail whose
    own estimate is too low — into the result slot the absorbed groups used to sit in (round-3 advisor finding: the hot groups were
    lost and the nested groups appended twice); and with a composite key the tail's result must be laid out for ALL key columns
    (it was laid out for one: the unpack wrote past the key arrays).  Same groups as the oracle in every case."""
    rng = np.random.default_rng({"sliced_tail": 11, "underestimated_tail": 12, "late_hot_key": 13, "two_keys": 14}[case])
    n, g = 17_000_000, 2_000_000
    opts = []
    ids = _skewed_ids(rng, n, g, 0.85, 800)
    if case == "late_hot_key":
        # one more heavy key that only shows up in the last tenth of the rows (image or not, it is the same call); a forced tiny
        # slice size also cuts its partition of the tail into many slices -> nested merge
        ids[-n // 10:][rng.random(n // 10) < 0.6] = 5_000_000
    if case in ("sliced_tail", "late_hot_key"):
        opts = [("slice_rows", 30_000)]
    if case == "underestimated_tail":
        opts = [("tail_groups_hint", 50_000)]                   # ~1.4 M groups in the tail: its LDS tables fill up -> overflow run
    if case == "two_keys":
        k0 = ((ids % 1000).astype(np.int64) * 7 - 3000, O.pack_mask(rng.random(n) < 0.0005), O.I64)
        k1 = ((ids // 1000).astype(np.uint32), None, O.U32CODE)
        keys, kds = [k0, k1], [O.I64, O.U32CODE]
    else:
        k = sparse_keys_from(ids)
        k[::100_003] = -1
        keys, kds = [(k, O.pack_mask(rng.random(n) < 0.001), O.I64)], [O.I64]
    v = rng.standard_normal(n)
    v[::50_021] = np.nan
    vals = [(v, O.pack_mask(rng.random(n) < 0.02), O.F64), (rng.standard_normal(n), O.pack_mask(rng.random(n) < 0.3), O.F64)]
    aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.MEAN), (1, O.SUM), (1, O.MIN), (1, O.MAX), (1, O.COUNT)]
    want = O.groupby_agg(keys, n, vals, aggs)
    for name, val in opts:
        ctx.set_option(name, val)
    try:
        got = ctx.groupby_agg(keys, n, vals, aggs)
        t = ctx.timings()
    finally:
        for name, _ in opts:
            ctx.set_option(name, 0)
    assert t["n_partitions"] == -1 and t["absorbed_rows"] > 0.5 * n, t          # the compact spill answered
    assert_groupby_equal(got, want, kds, int_exact_rows=(1, 2, 5, 6, 7))


def test_absorb_is_not_tried_on_uniform_keys(ctx):
    """The decision comes from the estimate's own sample (share of the rows on the most frequent keys): uniform keys
    must keep the ordinary path, at the cost of two small kernels."""
    rng = np.random.default_rng(9)
    n, g = 17_000_000, 12_000
    key = (sparse_keys(rng, n, g), None, O.I64)
    vals = [(rng.standard_normal(n), None, O.F64)]
    ctx.groupby_compute([key], n, vals, [(0, O.SUM), (0, O.MIN), (0, O.MAX)])
    assert ctx.timings()["absorbed_rows"] == 0


def test_two_pass_partition_under_median_and_group_by_row_lists(ctx):
    """The pair partition of Median / Nunique and group_by's row lists take the two-pass radix partition at fan-outs >= 6144
    (100 M rows); forced here at 132 partitions.  Same medians, same row lists (ascending rows per group) as the oracle."""
    rng = np.random.default_rng(2024)
    n = 9000 * 131 + 17                                        # 132 partitions of ~9 K rows: first-pass buckets = partition id >> 1
    keys = [(sparse_keys(rng, n, 20_000), O.pack_mask(rng.random(n) < 0.002), O.I64)]
    # no value nulls: the pair partition sees all n rows -> as many partitions.  (+ 0.0: no -0.0 among the values — a group holding both
    # zeros has a Median whose SIGN depends on the rows' order in the reference's stable sort)
    vals = [(np.round(rng.normal(500, 100, n), 1) + 0.0, None, O.F64)]
    aggs = [(0, O.MEDIAN), (0, O.NUNIQUE), (0, O.SUM)]
    want = O.groupby_agg(keys, n, vals, aggs)
    ctx.set_option("two_pass_min_p", 64)
    try:
        got = ctx.groupby_agg(keys, n, vals, aggs)                  # (its inner passes are not timed by phase)
        cells, nulls, off, rows = ctx.groupby_indices(keys, n)
        assert ctx.timings()["phase_ms"].get("prepartition", 0) > 0
    finally:
        ctx.set_option("two_pass_min_p", 0)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[0, 1])
    off, rows = np.asarray(off), np.asarray(rows)
    assert off[0] == 0 and off[-1] == n and np.array_equal(np.sort(rows), np.arange(n))
    k = keys[0][0].copy()
    null = np.unpackbits(keys[0][1], bitorder="little")[:n].astype(bool)
    first = rows[off[:-1]]
    for g in rng.integers(0, len(off) - 1, 200):
        r = rows[off[g]:off[g + 1]]
        assert np.all(np.diff(r) > 0)
        assert np.all(null[r]) if null[first[g]] else (not null[r].any() and np.all(k[r] == k[first[g]]))


@pytest.mark.parametrize("exact,tail", [(0, 5003), (1, 5003), (0, 12001), (1, 8192)])
def test_wide_scatter_tile_gives_the_same_groups(ctx, exact, tail):
    """From a fan-out of ~1 K the scatter ranks 16 K rows per workgroup at once and stages them in two halves (scatter_tile_wide,
    partition.hip).  Forced here (scatter_wide = 1) on both layouts — sampled capacity regions and the exact histogram — with NULL
    keys, the sentinel-valued key, masked value columns (validity bytes ride as byte columns) and a ragged last tile."""
    rng = np.random.default_rng(4100 + exact + tail)
    n = 16384 * 37 + tail                                       # full wide tiles + a ragged last one (one or two ordinary tiles)
    k = sparse_keys(rng, n, 60_000)
    k[::70_001] = -1
    keys = [(k, O.pack_mask(rng.random(n) < 0.003), O.I64)]
    vals = [(rng.normal(100, 10, n), O.pack_mask(rng.random(n) < 0.1), O.F64), (rng.normal(5, 1, n), None, O.F64),
            (rng.integers(-1000, 1000, n).astype(np.int64), O.pack_mask(rng.random(n) < 0.5), O.I64)]
    aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (1, O.MEAN), (1, O.MAX), (2, O.SUM), (2, O.MIN), (2, O.COUNT)]
    want = O.groupby_agg(keys, n, vals, aggs)
    for name, val in (("scatter_wide", 1), ("exact_partition", exact), ("no_direct", 1), ("no_absorb", 1), ("no_small", 1)):
        ctx.set_option(name, val)
    try:
        got = ctx.groupby_agg(keys, n, vals, aggs)
        assert ctx.timings()["n_partitions"] >= 16
        ctx.set_option("scatter_wide", -1)
        got1 = ctx.groupby_agg(keys, n, vals, aggs)
    finally:
        for name in ("scatter_wide", "exact_partition", "no_direct", "no_absorb", "no_small"):
            ctx.set_option(name, 0)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[1, 2, 4, 5, 6, 7])
    assert_groupby_equal(got1, want, [O.I64], int_exact_rows=[1, 2, 4, 5, 6, 7])


def test_a_few_hot_keys_in_front_of_a_long_tail_are_planned_without_a_retry(ctx):
    """The sampled group estimate: 2 K keys holding 80 % of the rows + 600 K others.  The uniform-occupancy model reads the sample
    (mostly repeats of the hot keys) as ~60 K groups, every LDS table overflows and the call used to be retried with 4 x the
    fan-out up to three times (C2's shape: 12-29 ms instead of ~4).  The Chao1 term from the sample's singletons and doubletons
    sees the tail: no retry, estimate within 2 x, same answers as the oracle."""
    rng = np.random.default_rng(808)
    n, hot, cold = 9_000_000, 2_000, 600_000
    ids = np.where(rng.random(n) < 0.8, rng.integers(0, hot, n), hot + rng.integers(0, cold, n))
    keys = [(sparse_keys_from(ids), None, O.I64)]
    vals = [(rng.normal(100, 10, n), None, O.F64), (rng.normal(5, 1, n), None, O.F64)]
    aggs = [(0, O.SUM), (0, O.MIN), (1, O.MAX), (1, O.MEAN), (0, O.COUNT)]
    want = O.groupby_agg(keys, n, vals, aggs)
    true_groups = want[0].shape[1]
    ctx.set_option("no_absorb", 1)                              # the radix path itself
    try:
        got = ctx.groupby_agg(keys, n, vals, aggs)
        t = ctx.timings()
        ctx.set_option("no_chao", 1)
        ctx.groupby_compute(keys, n, vals, aggs)
        t_model = ctx.timings()
    finally:
        ctx.set_option("no_absorb", 0); ctx.set_option("no_chao", 0)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[1, 2, 4])
    assert t["retries"] == 0, t
    assert 0.5 * true_groups <= t["estimated_groups"] <= 2.0 * true_groups, (t["estimated_groups"], true_groups)
    assert t_model["estimated_groups"] < 0.3 * true_groups      # what the model alone made of the same sample


def test_full_tables_hand_their_unplaced_rows_to_a_run_of_their_own(ctx):
    """An estimate that is too low (here: a forced fan-out of 128 for 300 K groups) fills LDS tables.  The rows whose key found no
    slot are a disjoint sub-problem — their keys are in no table — so they are grouped in a run of their own and appended, instead
    of the whole call starting over with 4 x the fan-out (C2's own 80/20 variant: 7.8 -> 4.9 ms).  Same answers as the oracle, with
    masked values and the NULL / sentinel keys around; with the path switched off the retry answers as before."""
    rng = np.random.default_rng(515)
    n, g = 6_000_000, 300_000
    k = sparse_keys(rng, n, g)
    k[::99_991] = -1
    keys = [(k, O.pack_mask(rng.random(n) < 0.001), O.I64)]
    # (one uniform profile — the lean aggregate's — over both columns: sum / min / max with null masks)
    vals = [(rng.normal(100, 10, n), O.pack_mask(rng.random(n) < 0.1), O.F64), (rng.normal(5, 1, n), O.pack_mask(rng.random(n) < 0.3), O.F64)]
    aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (1, O.MEAN), (1, O.MAX), (1, O.MIN), (0, O.COUNT)]
    want = O.groupby_agg(keys, n, vals, aggs)
    for name, val in (("partitions", 128), ("no_absorb", 1), ("no_direct", 1)):     # ~2340 groups per table of ~2050 slots
        ctx.set_option(name, val)
    try:
        got = ctx.groupby_agg(keys, n, vals, aggs)
        t = ctx.timings()
        ctx.set_option("no_overflow_run", 1)
        got1 = ctx.groupby_agg(keys, n, vals, aggs)
        t1 = ctx.timings()
    finally:
        for name in ("partitions", "no_absorb", "no_direct", "no_overflow_run"):
            ctx.set_option(name, 0)
    assert t["retries"] >= 100, t                               # an overflow run answered
    assert 1 <= t1["retries"] < 100, t1                         # the retry with more partitions
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[1, 2, 4, 5, 6])
    assert_groupby_equal(got1, want, [O.I64], int_exact_rows=[1, 2, 4, 5, 6])


def test_an_overflow_run_that_outgrows_one_radix_level_fails_the_attempt_not_the_call(ctx):
    """Found by the round-4 fuzz (case 133 of seed 93001): the run over a full table's unplaced rows is a nested run and cannot take
    the two-level path; when those rows alone hold more groups than one radix level takes it returned "group cardinality exceeds the
    radix capacity" and the whole CALL failed with it.  It now fails the ATTEMPT: the call goes on to more partitions or the
    two-level path.  Forced here with a tiny level (p_max = 24: 24 tables) under 900 K groups behind a hot key."""
    rng = np.random.default_rng(133)
    n, g = 1_200_000, 900_000
    ids = rng.integers(0, g, n)
    ids[rng.random(n) < 0.6] = 0
    keys = [(sparse_keys_from(ids), None, O.I64)]
    vals = [(rng.normal(50, 20, n), None, O.F64)]
    aggs = [(0, O.MEAN), (0, O.MAX), (0, O.MIN)]
    want = O.groupby_agg(keys, n, vals, aggs)
    opts = {"p_max": 24, "exact_partition": 1, "no_small": 1, "no_absorb": -1, "scatter_wide": 1, "wide_slices": 1}
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        got = ctx.groupby_agg(keys, n, vals, aggs)
    finally:
        for k in opts:
            ctx.set_option(k, 0)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[1, 2])


@pytest.mark.parametrize("vkind", [O.F64, O.I64])
def test_slices_of_a_hot_partition_fold_their_rows_per_wave(ctx, vkind):
    """A slice of an oversized partition is mostly ONE key (a hot key, the NULL group): when all placed rows of a wave sit in one
    slot the lean aggregate reduces them on the VALU (DPP row shifts / broadcasts) and one lane updates the table (aggregate2.hip,
    wave_fold).  Half the rows on one key, a tenth on the NULL key, the rest on 50 K keys; masked values, NaN / inf among them; small
    forced slices so that many tables are slices.  Same answers as the oracle (integer sums and min / max bit for bit)."""
    rng = np.random.default_rng(91 + vkind)
    n = 3_000_000
    k = sparse_keys(rng, n, 50_000)
    k[rng.random(n) < 0.5] = 424_242_424_242
    keys = [(k, O.pack_mask(rng.random(n) < 0.1), O.I64)]
    if vkind == O.F64:
        v0 = rng.normal(100, 10, n); v0[::70_001] = np.nan; v0[::90_001] = np.inf
        v1 = rng.normal(-5, 3, n)
    else:
        v0 = rng.integers(-10**9, 10**9, n).astype(np.int64); v1 = rng.integers(-50, 50, n).astype(np.int64)
    vals = [(v0, O.pack_mask(rng.random(n) < 0.2), vkind), (v1, O.pack_mask(rng.random(n) < 0.01), vkind)]
    aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.MEAN), (1, O.SUM), (1, O.MIN), (1, O.MAX), (1, O.COUNT)]
    want = O.groupby_agg(keys, n, vals, aggs)
    for name, val in (("slice_rows", 40_000), ("no_absorb", 1), ("no_direct", 1)):
        ctx.set_option(name, val)
    try:
        got = ctx.groupby_agg(keys, n, vals, aggs)
    finally:
        for name in ("slice_rows", "no_absorb", "no_direct"):
            ctx.set_option(name, 0)
    exact = [1, 2, 5, 6, 7] + ([0, 4] if vkind == O.I64 else [])
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=exact)


def test_tables_of_unequal_size_are_drawn_largest_first(ctx):
    """The lean aggregate's workgroups draw their tables from a ticket counter, in the order build_tables_kernel sorts them
    (groupby.hip: 64 size classes, largest first), so partitions that hold a hot key — not far enough above the average to be cut —
    start first and do not become the kernel's tail (Zipf(0.8) over 5 M keys, C2's shape: aggregate 7.8 -> 1.6 ms together with the
    one-round plan below).  Zipf-like keys over 1.5 M groups, 12 states: with the order, in partition order (`no_table_order`), and
    with small forced pieces (every table a slice, order over pieces) — the oracle's answers each time."""
    rng = np.random.default_rng(404)
    n, g, a = 17_000_000, 1_500_000, 0.8
    u = rng.random(n)
    ids = np.clip((((g ** (1 - a) - 1) * u + 1) ** (1 / (1 - a))).astype(np.int64), 1, g)
    keys = [(sparse_keys_from(ids), None, O.I64)]
    vals = [(rng.normal(100, 10, n), None, O.F64) for _ in range(4)]
    aggs = [(c, op) for c in range(4) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
    want = O.groupby_agg(keys, n, vals, aggs)
    exact = [2, 3, 6, 7, 10, 11, 14, 15, 16]
    for opts in ({}, {"no_table_order": 1}, {"slice_rows": 50_000}):
        for name, val in opts.items():
            ctx.set_option(name, val)
        try:
            got = ctx.groupby_agg(keys, n, vals, aggs)
            t = ctx.timings()
        finally:
            for name in opts:
                ctx.set_option(name, 0)
        assert_groupby_equal(got, want, [O.I64], int_exact_rows=exact)
        assert t["n_partitions"] >= 1024, t                # (the radix path with the lean kernel, not the direct or absorb paths)


def test_one_lean_round_is_preferred_to_rounds_up_to_the_scatter_limit():
    """The rounds heuristic (groupby.hip, P_TARGET; experiments/p_target_sweep.py).  C2's 12 states over 5 M uniform groups: the lean
    kernel in two rounds (two columns per launch over the same 2816 partitions, 2528-slot tables) — the default since the lean kernel has
    rounds; one lean round at a fan-out of 4864 (1408 slots) with `p_target` = 8192, and without lean rounds (`no_lean_rounds`: one lean
    round up to the scatter's limit, the plan before); the older kernel's two rounds (2396 slots) with `p_target` = 3072 on top.  Same
    group count, counts, sums and extremes every way."""
    import torch
    import pandrs_amd as pa
    n, g, d = 60_000_000, 5_000_000, "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(5)
    ids = torch.randint(0, g, (n,), device=d, generator=gen)
    keys = ids * -7046029254386353131
    vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) for _ in range(4)]
    aggs = [(c, op) for c in range(4) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
    true_groups = torch.unique(ids).numel()
    c = pa.Context(0)
    try:
        for p_target, no_lean_rounds, slots in ((8192, 0, 1408), (0, 0, 2528), (0, 1, 1408), (3072, 1, 2396)):
            c.set_option("p_target", p_target)
            c.set_option("no_lean_rounds", no_lean_rounds)
            ng = c.groupby_compute([(keys, None, O.I64)], n, [(v, None, O.F64) for v in vals], aggs)
            t = c.timings()
            kc, kn, oa = c.groupby_fetch()
            assert ng == true_groups and torch.unique(kc[0]).numel() == ng
            assert float(oa[16].sum()) == n
            for col in range(4):
                tot = float(vals[col].sum())
                assert abs(float(oa[4 * col].sum()) - tot) <= 1e-9 * max(abs(tot), 1.0)
                assert float(oa[4 * col + 2].min()) == float(vals[col].min()) and float(oa[4 * col + 3].max()) == float(vals[col].max())
            assert t["table_slots"] == slots, t
            assert (t["n_partitions"] > 4096) == (slots == 1408), t
            del kc, kn, oa
    finally:
        c.set_option("p_target", 0)
        c.set_option("no_lean_rounds", 0)
        c.close()


@pytest.mark.parametrize("g", [300, 10_000, 400_000])
def test_eight_columns_of_sum_min_max_never_need_a_merge_they_cannot_have(ctx, g):
    """8 value columns x sum / mean / min / max = 24 partial states; a merge of partial records takes every state as a source of its own
    and at most 16 of them, so the paths that end in a merge (few-groups direct path, sliced partitions) used to FAIL such a call with
    "too many states to merge (24)" (round 4 cliff hunt: 10 K groups, 50 M rows).  They are not taken beyond 16 states: the radix path's
    rounds answer.  Few, mid and many groups, oracle-compared."""
    rng = np.random.default_rng(808 + g)
    n = 4_500_000
    keys = [(sparse_keys(rng, n, g), None, O.I64)]
    vals = [(rng.normal(50 + 10 * c, 3, n), O.pack_mask(rng.random(n) < 0.05) if c % 3 == 0 else None, O.F64) for c in range(8)]
    aggs = [(c, op) for c in range(8) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)]
    want = O.groupby_agg(keys, n, vals, aggs)
    got = ctx.groupby_agg(keys, n, vals, aggs)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX)])


@pytest.mark.parametrize("ncol,kind,masked", [(5, "f64", False), (6, "f64", True), (7, "i64", False), (9, "f64", True), (13, "f64", False), (16, "f64sum", False)])
def test_wide_aggregations_run_the_lean_kernel_in_rounds(ctx, ncol, kind, masked):
    """More than 4 uniform columns: the lean kernel folds them 4 at a time, one launch per round over the same partitions; launch 0's key
    table and output positions are what the later launches start from, so every round's outputs land on the same group rows
    (aggregate2.hip, snap_*).  Sum / mean / min / max (+ count) over 5-16 columns, null masks, i64 columns, NULL keys and the table's
    sentinel bits among the keys — the oracle's answers; the same with the older kernel's rounds (`no_lean_rounds`)."""
    rng = np.random.default_rng(900 + ncol)
    n, g = 3_000_000, 150_000
    k = sparse_keys(rng, n, g)
    k[rng.random(n) < 0.001] = -1                     # the table sentinel's bits
    keys = [(k, O.pack_mask(rng.random(n) < 0.01), O.I64)]
    dt = O.I64 if kind == "i64" else O.F64
    vals = []
    for c in range(ncol):
        x = rng.integers(-10**6, 10**6, n).astype(np.int64) if kind == "i64" else rng.normal(50 + 5 * c, 2, n)
        vals.append((x, O.pack_mask(rng.random(n) < 0.1) if masked else None, dt))
    ops = {"f64": (O.SUM, O.MEAN, O.MIN, O.MAX), "i64": (O.SUM, O.MIN, O.MAX), "f64sum": (O.SUM, O.MEAN)}[kind]       # (at most 40 states per call)
    aggs = [(c, op) for c in range(ncol) for op in ops][:60] + [(ncol - 1, O.COUNT)]
    want = O.groupby_agg(keys, n, vals, aggs)
    exact = [i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT) or (kind == "i64" and op == O.SUM)]
    for no_lean in (0, 1):
        ctx.set_option("no_lean_rounds", no_lean)
        try:
            got = ctx.groupby_agg(keys, n, vals, aggs)
            t = ctx.timings()
        finally:
            ctx.set_option("no_lean_rounds", 0)
        assert_groupby_equal(got, want, [O.I64], int_exact_rows=exact)
        assert t["n_partitions"] > 0, t                     # the radix path (the NULL group's 1 % of the rows may read as a dominant key:
                                                            # then it is one round of the older kernel with slices — either plan must be exact)


def test_a_full_table_in_the_first_round_fails_the_attempt_for_all_rounds(ctx):
    """Rounds of the lean kernel with an estimate far too low (`groups_hint`): launch 0's tables fill up, its snapshots are incomplete,
    the later launches must not touch them — the attempt is repeated with more partitions and the answers are the oracle's."""
    rng = np.random.default_rng(77)
    n, g = 2_500_000, 600_000
    keys = [(sparse_keys(rng, n, g), None, O.I64)]
    vals = [(rng.normal(100, 10, n), None, O.F64) for _ in range(6)]
    aggs = [(c, op) for c in range(6) for op in (O.SUM, O.MIN, O.MAX)]
    want = O.groupby_agg(keys, n, vals, aggs)
    ctx.set_option("groups_hint", 120_000)
    ctx.set_option("no_small", 1)
    try:
        got = ctx.groupby_agg(keys, n, vals, aggs)
        t = ctx.timings()
    finally:
        ctx.set_option("groups_hint", 0)
        ctx.set_option("no_small", 0)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[i for i, (c, op) in enumerate(aggs) if op != O.SUM])
    assert t["retries"] >= 1, t


def test_wide_aggregations_of_mixed_kinds_take_rounds_of_four(ctx):
    """f64 and i64 columns side by side, sums on some and min / max on others: no uniform profile, so the older kernel answers — in rounds of
    at most 4 columns since the round-4 cliff hunt (its catch-all instantiation for more sources keeps register arrays for 16 and spills:
    4 f64 + 4 i64 columns x sum, 50 M rows: 8.3 ms in one round, 3.8 in two).  9 columns, the oracle's answers."""
    rng = np.random.default_rng(4242)
    n, g = 3_200_000, 200_000
    keys = [(sparse_keys(rng, n, g), O.pack_mask(rng.random(n) < 0.01), O.I64)]
    vals, aggs = [], []
    for c in range(9):
        if c % 2 == 0:
            vals.append((rng.normal(50 + c, 3, n), O.pack_mask(rng.random(n) < 0.1) if c % 4 == 0 else None, O.F64))
            aggs += [(c, O.SUM), (c, O.MEAN)] if c % 3 else [(c, O.MIN), (c, O.MAX)]
        else:
            vals.append((rng.integers(-10**7, 10**7, n).astype(np.int64), None, O.I64))
            aggs += [(c, O.SUM), (c, O.MAX)]
    aggs.append((0, O.COUNT))
    want = O.groupby_agg(keys, n, vals, aggs)
    got = ctx.groupby_agg(keys, n, vals, aggs)
    exact = [i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT) or (vals[c][2] == O.I64 and op == O.SUM)]
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=exact)


@pytest.mark.parametrize("share,ncol,kind", [(0.5, 8, "f64x3"), (0.12, 6, "mixed")])
def test_wide_aggregations_with_a_dominant_key_keep_their_slices(ctx, share, ncol, kind):
    """A dominant key's partition must be cut into row slices whatever the plan — in rounds it was not, and became ONE workgroup's job (half
    the rows on one key, 8 columns, 50 M rows: 222 ms).  The lean kernel's rounds cut it now (a piece's partial record is filled in round
    by round at the position launch 0 recorded; one merge at the end — of up to 39 states, it took 16 and such a call used to FAIL with
    "too many states to merge"); columns of mixed kinds stay in one round of the older kernel with slices when the estimate's far pairs
    show a dominant key.  Both shapes, the oracle's answers, one attempt."""
    rng = np.random.default_rng(515)
    n, g = 4_500_000, 300_000
    ids = rng.integers(1, g, n)
    ids[rng.random(n) < share] = 0
    keys = [(sparse_keys_from(ids), None, O.I64)]
    if kind == "f64x3":
        vals = [(rng.normal(50 + c, 2, n), None, O.F64) for c in range(ncol)]
        aggs = [(c, op) for c in range(ncol) for op in (O.SUM, O.MIN, O.MAX)]
    else:
        vals = [((rng.normal(50 + c, 2, n), None, O.F64) if c % 2 else (rng.integers(-10**6, 10**6, n).astype(np.int64), None, O.I64)) for c in range(ncol)]
        aggs = [(c, O.SUM) for c in range(ncol)] + [(1, O.MAX), (0, O.COUNT)]
    want = O.groupby_agg(keys, n, vals, aggs)
    got = ctx.groupby_agg(keys, n, vals, aggs)
    t = ctx.timings()
    exact = [i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT) or (vals[c][2] == O.I64 and op == O.SUM)]
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=exact)
    assert t["retries"] == 0, t
