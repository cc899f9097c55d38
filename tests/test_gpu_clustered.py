"""GPU parity tests for rows CLUSTERED by key (sorted input, input grouped by key): the one-pass path of
pandrs_amd/csrc/clustered.hip + groupby.hip::run_clustered against the CPU oracle.  The reference has no such case of its own — its
group_by walks the rows in order into a HashMap whatever their order (grouping.rs:22-115) — so the expected answers are the oracle's
on the same inputs; what is tested is that the row order changes the device's plan and never the result.
Tolerances as everywhere: keys / counts / min / max bit-exact, f64 sums and means within 1e-9 relative."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import assert_groupby_equal

pytestmark = pytest.mark.gpu

CLUSTERED = -2          # timings()["n_partitions"] of a call the clustered-rows pass answered


@pytest.fixture(scope="module")
def ctx():
    import pandrs_amd as pa
    c = pa.Context(0)
    c.set_option("no_small", 1)            # (calls of <= 2 M rows would take the two-launch path before any estimate)
    yield c
    c.close()


def mixed(ids):
    return (np.asarray(ids).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)


def runs_of(rng, n, g, run):
    """rows in runs of `run` equal keys, the runs' keys random: a key comes back in several runs."""
    return np.repeat(rng.integers(0, g, (n + run - 1) // run), run)[:n]


def run_and_check(ctx, keys, n, vals, aggs, key_dtypes, exact, expect_clustered=True):
    want = O.groupby_agg(keys, n, vals, aggs)
    got = ctx.groupby_agg(keys, n, vals, aggs)
    t = ctx.timings()
    assert_groupby_equal(got, want, key_dtypes, int_exact_rows=exact)
    if expect_clustered is not None:
        assert (t["n_partitions"] == CLUSTERED) == expect_clustered, t
    return t


@pytest.mark.parametrize("layout", ["sorted", "sorted_mixed_bits", "runs64", "runs12"])
def test_sorted_and_grouped_rows_take_one_pass(ctx, layout):
    """C2's aggregate set over rows sorted by key, sorted by the keys' mixed bits (runs in random key order), and in runs of 64 / 12
    equal keys whose keys come back later — the same answers as for any other row order, from the clustered-rows pass."""
    rng = np.random.default_rng(11)
    n, g = 3_000_001, 40_000
    ids = {"sorted": lambda: np.sort(rng.integers(0, g, n)), "sorted_mixed_bits": lambda: rng.integers(0, g, n),
           "runs64": lambda: runs_of(rng, n, g, 64), "runs12": lambda: runs_of(rng, n, g, 12)}[layout]()
    k = np.sort(mixed(ids)) if layout == "sorted_mixed_bits" else mixed(ids)
    if layout == "sorted":
        k[100_000:100_300] = -1                  # a run of the table's sentinel bits
    vals = [(rng.normal(100, 10, n), None, O.F64) for _ in range(4)]
    vals[1][0][::50_001] = np.nan
    vals[2][0][::70_001] = np.inf
    aggs = [(c, op) for c in range(4) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
    run_and_check(ctx, [(k, None, O.I64)], n, vals, aggs, [O.I64], exact=[2, 3, 6, 7, 10, 11, 14, 15, 16])


@pytest.mark.parametrize("kd", ["f64", "codes", "bool"])
def test_clustered_rows_with_every_key_dtype_and_null_keys(ctx, kd):
    """f64 keys (every NaN one group, -0.0 and 0.0 two; grouping.rs:79), string-pool codes and bool bits, each behind a null mask
    (the NULL group: grouping.rs:74), sorted; values behind null masks too (sum / mean / min / max skip them, count does not)."""
    rng = np.random.default_rng(23)
    n = 2_500_003
    if kd == "f64":
        pool = np.concatenate([rng.normal(size=29_996), [0.0, -0.0, np.nan, np.inf]])
        ids = np.sort(rng.integers(0, len(pool), n))
        key = (pool[ids], O.pack_mask(np.sort(rng.random(n)) < 0.01), O.F64)
    elif kd == "codes":
        ids = np.sort(rng.integers(0, 50_000, n))
        key = (ids.astype(np.uint32), O.pack_mask(np.sort(rng.random(n)) > 0.99), O.U32CODE)
    else:
        ids = np.sort(rng.integers(0, 2, n))
        key = (np.packbits(ids == 1, bitorder="little"), O.pack_mask(np.sort(rng.random(n)) < 0.02), O.BOOLBITS)
    vals = [(rng.normal(40, 5, n), O.pack_mask(rng.random(n) < 0.1), O.F64), (rng.normal(3, 1, n), O.pack_mask(rng.random(n) < 0.5), O.F64)]
    aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.MEAN), (1, O.SUM), (1, O.MIN), (1, O.MAX), (1, O.COUNT)]
    dt = {"f64": O.F64, "codes": O.U32CODE, "bool": O.BOOLBITS}[kd]
    # (two bool groups in two long runs read as "a dominant key", not as clustered rows: the few-groups path answers)
    run_and_check(ctx, [key], n, vals, aggs, [dt], exact=[1, 2, 5, 6, 7], expect_clustered=None if kd == "bool" else True)


@pytest.mark.parametrize("profile", ["f64_sum", "f64_minmax", "f64_max", "i64_sum", "i64_all"])
def test_clustered_rows_in_every_instantiated_profile(ctx, profile):
    """The kernel's other instantiations: one f64 sum (the north-star shape), min + max alone, i64 sums (exact) and i64 sum / min / max
    over three columns."""
    rng = np.random.default_rng(31)
    n = 2_300_000
    k = mixed(np.sort(rng.integers(0, 70_000, n)))
    if profile == "f64_sum":
        vals, aggs, exact = [(rng.normal(100, 10, n), None, O.F64)], [(0, O.SUM)], []
    elif profile == "f64_minmax":
        vals, aggs, exact = [(rng.normal(100, 10, n), None, O.F64), (rng.normal(100, 10, n), None, O.F64)], [(0, O.MIN), (0, O.MAX), (1, O.MIN), (1, O.MAX)], [0, 1, 2, 3]
    elif profile == "f64_max":
        vals, aggs, exact = [(rng.normal(100, 10, n), None, O.F64)], [(0, O.MAX)], [0]
    elif profile == "i64_sum":
        vals, aggs, exact = [(rng.integers(-10**12, 10**12, n), None, O.I64)], [(0, O.SUM), (0, O.COUNT)], [0, 1]
    else:
        vals = [(rng.integers(-10**9, 10**9, n), None, O.I64) for _ in range(3)]
        aggs, exact = [(c, op) for c in range(3) for op in (O.SUM, O.MIN, O.MAX)], list(range(9))
    run_and_check(ctx, [(k, None, O.I64)], n, vals, aggs, [O.I64], exact=exact)


def test_clustered_rows_with_two_key_columns(ctx):
    """Rows sorted by (a, b): the packed composite cell is what the pass groups on; the result has both key columns."""
    rng = np.random.default_rng(41)
    n = 2_200_000
    a, b = rng.integers(0, 300, n), rng.integers(0, 200, n)
    idx = np.lexsort((b, a))
    a, b = a[idx], b[idx]
    keys = [(a.astype(np.int64), None, O.I64), (b.astype(np.uint32), None, O.U32CODE)]
    vals = [(rng.normal(100, 10, n), None, O.F64)]
    aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.COUNT)]
    run_and_check(ctx, keys, n, vals, aggs, [O.I64, O.U32CODE], exact=[1, 2, 3])


def test_columns_off_the_16_byte_grid_take_the_scalar_loads():
    """Resident columns that start 8 bytes into an allocation (a view): the kernel's 16-byte loads are off, its answers are not."""
    import torch
    import pandrs_amd as pa
    n, d = 4_000_000, "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(8)
    ids = torch.sort(torch.randint(0, 30_000, (n + 1,), device=d, generator=gen))[0]
    k_all = ids * -7046029254386353131
    v_all = torch.randn(n + 1, device=d, generator=gen, dtype=torch.float64)
    c = pa.Context(0)
    try:
        res = {}
        for off in (0, 1):
            k, v = k_all[off:off + n], v_all[off:off + n]
            assert (k.data_ptr() % 16 == 0) == (off == 0)
            c.groupby_compute([(k, None, O.I64)], n, [(v, None, O.F64)], [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.COUNT)])
            assert c.timings()["n_partitions"] == CLUSTERED
            kc, kn, oa = c.groupby_fetch()
            order = torch.argsort(kc[0])
            uk, inv = torch.unique(k, return_inverse=True)
            assert torch.equal(kc[0][order], uk)
            cnt = torch.bincount(inv, minlength=uk.numel()).to(torch.float64)
            assert torch.equal(oa[3][order], cnt)
            s = torch.zeros(uk.numel(), device=d, dtype=torch.float64).index_add_(0, inv, v)
            assert torch.allclose(oa[0][order], s, rtol=1e-9, atol=1e-9)
            mn = torch.full((uk.numel(),), float("inf"), device=d, dtype=torch.float64).scatter_reduce_(0, inv, v, "amin")
            mx = torch.full((uk.numel(),), float("-inf"), device=d, dtype=torch.float64).scatter_reduce_(0, inv, v, "amax")
            assert torch.equal(oa[1][order], mn) and torch.equal(oa[2][order], mx)
            res[off] = int(kc.shape[1])
        assert res[0] > 0 and res[1] > 0
    finally:
        c.close()


def test_a_chunk_with_more_runs_than_its_table_hands_the_call_back(ctx):
    """The chunk length comes from the sample's average run length; rows whose runs are much shorter somewhere (here: forced, one chunk
    of 4 M rows for 150 K runs) fill a chunk's table — the pass gives up and the ordinary path answers, same result."""
    rng = np.random.default_rng(57)
    n = 2_400_000
    k = mixed(runs_of(rng, n, 900_000, 16))
    vals = [(rng.normal(100, 10, n), None, O.F64)]
    aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.COUNT)]
    ctx.set_option("clustered_chunk", 1 << 22)
    try:
        run_and_check(ctx, [(k, None, O.I64)], n, vals, aggs, [O.I64], exact=[1, 2, 3], expect_clustered=False)
    finally:
        ctx.set_option("clustered_chunk", 0)
    # the same rows with the chunk the sample asks for
    run_and_check(ctx, [(k, None, O.I64)], n, vals, aggs, [O.I64], exact=[1, 2, 3], expect_clustered=True)
    # and with the pass switched off
    ctx.set_option("no_clustered", 1)
    try:
        run_and_check(ctx, [(k, None, O.I64)], n, vals, aggs, [O.I64], exact=[1, 2, 3], expect_clustered=False)
    finally:
        ctx.set_option("no_clustered", 0)


def test_sorted_config2_at_full_size():
    """C2's shape with the rows sorted by key — 100 M rows, 1 M groups, 4 f64 columns x sum / mean / min / max — through the
    size-independent properties of the random-order test: group count, sum of counts, linearity of the sums, global extremes,
    mean x count = sum."""
    import torch
    import pandrs_amd as pa
    n, g, d = 100_000_000, 1_000_000, "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(77)
    ids = torch.sort(torch.randint(0, g, (n,), device=d, generator=gen))[0]
    keys = ids * -7046029254386353131
    true_groups = torch.unique_consecutive(ids).numel()
    del ids
    vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(4)]
    aggs = [(c, op) for c in range(4) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
    c = pa.Context(0)
    try:
        ng = c.groupby_compute([(keys, None, O.I64)], n, [(v, None, O.F64) for v in vals], aggs)
        t = c.timings()
        kc, kn, oa = c.groupby_fetch()
        assert t["n_partitions"] == CLUSTERED, t
        assert ng == true_groups and torch.unique(kc[0]).numel() == ng and int(kn.sum()) == 0
        cnt = oa[16]
        assert float(cnt.sum()) == n
        for col in range(4):
            s, mean, mn, mx = oa[4 * col:4 * col + 4]
            tot = float(vals[col].sum())
            assert abs(float(s.sum()) - tot) <= 1e-9 * abs(tot)
            assert float(mn.min()) == float(vals[col].min()) and float(mx.max()) == float(vals[col].max())
            assert torch.allclose(mean * cnt, s, rtol=1e-12, atol=0)
    finally:
        c.close()


@pytest.mark.parametrize("run", [3, 4])
def test_fixed_length_runs_do_not_alias_with_the_sample(ctx, run):
    """Every key exactly `run` consecutive rows (three measurements per subject, …) with the sample's stride a multiple of the run length:
    a fixed-stride sample only ever saw the FIRST row of a run, read "no adjacent pair differs", and the run bound cut the estimate to
    the sample's own distinct count (100 M rows, runs of 3: 230 K for 1 M groups, overflowing tables, 49 ms).  The sampled row now
    sits at a pseudo-random offset inside its stride: the estimate is the truth's size, one attempt; and rows in runs this short take
    the lean kernel behind the exact partition, not the RUNS instantiation of the older one."""
    rng = np.random.default_rng(60 + run)
    n = 262_144 * 12                       # the estimate samples 262 144 rows: stride 12
    k = mixed(np.repeat(rng.integers(0, 1 << 40, n // run + 1), run)[:n])
    vals = [(rng.normal(100, 10, n), None, O.F64) for _ in range(4)]
    aggs = [(c, op) for c in range(4) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)]
    want = O.groupby_agg([(k, None, O.I64)], n, vals, aggs)
    got = ctx.groupby_agg([(k, None, O.I64)], n, vals, aggs)
    t = ctx.timings()
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[2, 3, 6, 7, 10, 11, 14, 15])
    true_groups = want[0].shape[1]
    assert 0.8 * true_groups <= t["estimated_groups"] <= 1.3 * true_groups, (t, true_groups)
    assert t["retries"] == 0 and t["n_partitions"] > 0, t
    assert t["table_slots"] == 1408, t                 # the lean kernel's 109-byte slots (the older kernel: 20 + 8 x states bytes, in rounds)


@pytest.mark.parametrize("rows_per_key,jitter", [(100, 50), (10, 40), (300, 20_000)])
def test_nearly_sorted_keys_are_sized_by_their_windows(ctx, rows_per_key, jitter):
    """Event times arriving slightly out of order: key = bucket of (i + noise).  One sampled row per stride never sees such a key twice
    (its rows lie inside one or two strides), the sample reads "all distinct" and the model extrapolated 8 x beyond the truth (100 M
    rows, 100 rows per key within +-50: 8.2 M for 1 M groups; 10 rows per key: 91 M, the two-level path).  The estimate now counts the
    distinct keys inside windows of consecutive rows — a bound in any row order, tight when keys are local in position — and such rows
    skip the sampled region plan (which assumes random order) for the exact histogram.  The oracle's answers, one attempt, and an
    estimate of the truth's size."""
    rng = np.random.default_rng(1000 + rows_per_key)
    n = 4_600_000
    pos = np.clip(np.arange(n) + rng.integers(-jitter, jitter + 1, n), 0, n - 1)
    k = mixed(pos // rows_per_key)
    vals = [(rng.normal(100, 10, n), None, O.F64) for _ in range(2)]       # (sums far from zero: the 1e-9 bar is relative)
    aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (1, O.SUM), (1, O.MIN), (1, O.MAX), (0, O.COUNT)]
    want = O.groupby_agg([(k, None, O.I64)], n, vals, aggs)
    got = ctx.groupby_agg([(k, None, O.I64)], n, vals, aggs)
    t = ctx.timings()
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[1, 2, 4, 5, 6])
    true_groups = want[0].shape[1]
    assert 0.9 * true_groups <= t["estimated_groups"] <= 1.6 * true_groups, (t, true_groups)
    assert t["retries"] in (0, 100), t                      # (100: a handful of rows through the overflow run — still one attempt)


def test_nearly_sorted_config2_at_full_size():
    """The same layout where it bites — 100 M rows, 100 rows per key arriving within +-50 rows of their place, C2's aggregates: the
    sample's stride (381 rows) is wider than a key's span, so without the windows the estimate is 8 x the truth.  Group count, sum of
    counts, linearity of the sums, global extremes; the estimate within 1.3 x of the truth with the windows and beyond 3 x without."""
    import torch
    import pandrs_amd as pa
    n, d = 100_000_000, "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(99)
    pos = (torch.arange(n, device=d) + torch.randint(-50, 51, (n,), device=d, generator=gen)).clamp_(0, n - 1)
    ids = pos // 100
    del pos
    keys = ids * -7046029254386353131
    true_groups = torch.unique(ids).numel()
    del ids
    vals = [torch.randn(n, device=d, generator=gen, dtype=torch.float64) * 10 + 100 for _ in range(4)]
    aggs = [(c, op) for c in range(4) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(0, O.COUNT)]
    c = pa.Context(0)
    try:
        for windows in (1, 0):
            c.set_option("no_window_bound", 1 - windows)
            ng = c.groupby_compute([(keys, None, O.I64)], n, [(v, None, O.F64) for v in vals], aggs)
            t = c.timings()
            kc, kn, oa = c.groupby_fetch()
            assert ng == true_groups and torch.unique(kc[0]).numel() == ng
            assert float(oa[16].sum()) == n
            for col in range(4):
                tot = float(vals[col].sum())
                assert abs(float(oa[4 * col].sum()) - tot) <= 1e-9 * abs(tot)
                assert float(oa[4 * col + 2].min()) == float(vals[col].min()) and float(oa[4 * col + 3].max()) == float(vals[col].max())
            if windows:
                assert true_groups <= t["estimated_groups"] <= 1.3 * true_groups, (t, true_groups)
            else:
                assert t["estimated_groups"] > 3 * true_groups, (t, true_groups)
            del kc, kn, oa
    finally:
        c.set_option("no_window_bound", 0)
        c.close()


@pytest.mark.parametrize("layout", ["sorted", "runs5", "nearly"])
def test_wide_aggregations_over_clustered_rows_run_the_burst_kernel_in_rounds(ctx, layout):
    """More than 4 columns over rows that are sorted, in short runs or nearly sorted: the one-pass path takes at most 4 columns, so these
    go to the exact partition and the burst kernel (clustered.hip, PARTS) in rounds of 4 columns — launch 0's key tables and output
    positions handed on as in the lean kernel (sorted rows, 8 columns x 4 aggregates, 50 M rows: 9.6 ms with the older kernel's 24
    states in one table -> 3.5).  7 columns with null masks, NULL keys, the sentinel's bits: the oracle's answers, also with the
    lean kernel in its place (`no_burst_kernel`)."""
    rng = np.random.default_rng(2024)
    n, g = 4_400_000, 120_000
    if layout == "sorted":
        ids = np.sort(rng.integers(0, g, n))
    elif layout == "runs5":
        ids = runs_of(rng, n, g, 5)
    else:
        ids = np.clip(np.arange(n) + rng.integers(-40, 41, n), 0, n - 1) // 37
    k = mixed(ids)
    k[1000:1200] = -1
    keys = [(k, O.pack_mask(np.sort(rng.random(n)) < 0.003), O.I64)]
    vals = [(rng.normal(50 + 3 * c, 2, n), O.pack_mask(rng.random(n) < 0.1), O.F64) for c in range(7)]
    aggs = [(c, op) for c in range(7) for op in (O.SUM, O.MEAN, O.MIN, O.MAX)] + [(3, O.COUNT)]
    want = O.groupby_agg(keys, n, vals, aggs)
    exact = [i for i, (c, op) in enumerate(aggs) if op in (O.MIN, O.MAX, O.COUNT)]
    for no_burst in (0, 1):
        ctx.set_option("no_burst_kernel", no_burst)
        try:
            got = ctx.groupby_agg(keys, n, vals, aggs)
            t = ctx.timings()
        finally:
            ctx.set_option("no_burst_kernel", 0)
        assert_groupby_equal(got, want, [O.I64], int_exact_rows=exact)
        assert t["n_partitions"] > 0, t
