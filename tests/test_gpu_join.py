"""GPU parity tests for the hash join: index pairs must equal the oracle's IN ORDER (the reference's
join order is deterministic), gathers bit-exact, fused join->groupby within 1e-9."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import assert_groupby_equal, codes_of

pytestmark = pytest.mark.gpu
HOWS = [("inner", O.INNER), ("left", O.LEFT), ("right", O.RIGHT), ("outer", O.OUTER)]


@pytest.fixture(scope="module")
def ctx():
    import pandrs_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def check_join(ctx, lk, nl, rk, nr, hows=HOWS):
    for name, how in hows:
        gl, gr = ctx.join_indices(lk, nl, rk, nr, how)
        wl, wr = O.join_indices(lk, nl, rk, nr, how)
        assert len(gl) == len(wl), (name, len(gl), len(wl))
        np.testing.assert_array_equal(gl, wl, err_msg=name)
        np.testing.assert_array_equal(gr, wr, err_msg=name)


def test_golden_join_vectors(ctx, golden):
    c = golden["join_string_key"]
    codes, _ = codes_of(c["left_keys"] + c["right_keys"])
    lk, rk = (codes[:4], None, O.U32CODE), (codes[4:], None, O.U32CODE)
    for name, how in HOWS:
        li, ri = ctx.join_indices(lk, 4, rk, 4, how)
        assert li.tolist() == c[name]["left_idx"] and ri.tolist() == c[name]["right_idx"], name
    n = golden["join_numeric_key"]
    li, ri = ctx.join_indices((np.array(n["left_keys_f64"]), None, O.F64), 3,
                              (np.array(n["right_keys_f64"]), None, O.F64), 3, O.INNER)
    assert li.tolist() == n["inner"]["left_idx"] and ri.tolist() == n["inner"]["right_idx"]
    o = golden["join_optimized"]
    lk = (np.array(o["left_ids"], np.int64), None, O.I64)
    rk = (np.array(o["right_ids"], np.int64), None, O.I64)
    for name, how in HOWS:
        assert len(ctx.join_indices(lk, 4, rk, 4, how)[0]) == o["rows"][name]
    d = o["disjoint"]
    li, _ = ctx.join_indices((np.array(d["left_ids"], np.int64), None, O.I64), 3,
                             (np.array(d["right_ids"], np.int64), None, O.I64), 3, O.INNER)
    assert len(li) == 0


def test_type_mismatch_and_empty(ctx):
    import pandrs_amd as pa
    with pytest.raises(pa.ColumnTypeMismatch):      # join.rs:98-104
        ctx.join_indices((np.zeros(3, np.int64), None, O.I64), 3, (np.zeros(3), None, O.F64), 3, O.INNER)
    e = (np.zeros(0, np.int64), None, O.I64)
    k = (np.array([3, 1, 2], np.int64), None, O.I64)
    check_join(ctx, e, 0, e, 0)
    check_join(ctx, k, 3, e, 0)
    check_join(ctx, e, 0, k, 3)


@pytest.mark.parametrize("nl,nr,space", [(1000, 800, 500), (200_000, 50_000, 40_000), (300_000, 300_000, 1_000_000),
                                         (50_000, 400_000, 30_000)])
def test_random_i64_with_dups_and_nulls(ctx, nl, nr, space):
    rng = np.random.default_rng(nl + nr)
    mix = lambda x: (x.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    lk = (mix(rng.integers(0, space, nl)), O.pack_mask(rng.random(nl) < 0.02), O.I64)
    rk = (mix(rng.integers(0, space, nr)), O.pack_mask(rng.random(nr) < 0.02), O.I64)
    check_join(ctx, lk, nl, rk, nr)


def test_single_pass_probe_matches(ctx):
    """The optional single-pass probe (lookup + decoupled look-back scan + emit in one kernel; measured
    within a few percent of the default lookup / scan / emit kernels, so it stays opt-in)."""
    rng = np.random.default_rng(404)
    nl, nr = 700_000, 90_000
    lk = (rng.integers(0, 60_000, nl).astype(np.int64), O.pack_mask(rng.random(nl) < 0.01), O.I64)
    rk = (rng.integers(0, 70_000, nr).astype(np.int64), O.pack_mask(rng.random(nr) < 0.01), O.I64)
    ctx.set_option("join_one_pass", 1)
    try:
        check_join(ctx, lk, nl, rk, nr)
    finally:
        ctx.set_option("join_one_pass", 0)
    check_join(ctx, lk, nl, rk, nr)


def test_sentinel_valued_key_on_either_side(ctx):
    """i64 -1 has the bit pattern of the hash tables' empty marker: it owns a dedicated entry that must
    read as "absent" when only the probe side holds the key (found by the randomised sweep)."""
    rng = np.random.default_rng(88)
    nl, nr = 200_000, 60_000
    base_l = rng.integers(0, 5000, nl).astype(np.int64)
    base_r = rng.integers(0, 5000, nr).astype(np.int64)
    for left_has, right_has in ((True, False), (False, True), (True, True)):
        l, r = base_l.copy(), base_r.copy()
        if left_has:
            l[rng.random(nl) < 0.05] = -1
        if right_has:
            r[rng.random(nr) < 0.0005] = -1
        check_join(ctx, (l, None, O.I64), nl, (r, None, O.I64), nr)


def test_unique_build_side_large(ctx):
    """C5's shape scaled down: unique build keys, probe keys uniform over them (+10 % misses)."""
    rng = np.random.default_rng(55)
    nr, nl = 1_000_000, 4_000_000
    rkeys = (rng.permutation(nr * 4)[:nr].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    pick = rng.integers(0, nr, nl)
    lkeys = rkeys[pick].copy()
    miss = rng.random(nl) < 0.1
    lkeys[miss] = rng.integers(-10**9, -1, miss.sum())
    check_join(ctx, (lkeys, None, O.I64), nl, (rkeys, None, O.I64), nr, hows=[("inner", O.INNER), ("outer", O.OUTER)])


def test_other_key_dtypes_and_heavy_dups(ctx):
    rng = np.random.default_rng(66)
    nl, nr = 30_000, 20_000
    lc = rng.integers(0, 300, nl).astype(np.uint32)
    rc = rng.integers(0, 300, nr).astype(np.uint32)          # ~67 duplicates per key on the build side
    check_join(ctx, (lc, O.pack_mask(rng.random(nl) < 0.01), O.U32CODE), nl, (rc, None, O.U32CODE), nr)
    lf = rng.choice(np.array([0.0, -0.0, 1.5, np.nan, np.inf, 7.25]), 2000)
    rf = rng.choice(np.array([0.0, -0.0, np.nan, 2.5, 7.25]), 50)
    check_join(ctx, (lf, None, O.F64), 2000, (rf, None, O.F64), 50)
    lb = np.packbits(rng.random(500) < 0.5, bitorder="little")
    rb = np.packbits(np.array([True, False, True]), bitorder="little")
    check_join(ctx, (lb, None, O.BOOLBITS), 500, (rb, None, O.BOOLBITS), 3)


def test_hot_build_key_beyond_one_lds_partition(ctx):
    """One build key with more duplicates than an LDS partition holds (20 000 > 8192): the general
    segmented sort (chunk sort + merge passes) takes over; matches stay in ascending right-row order."""
    rng = np.random.default_rng(8)
    nl, nr = 300, 60_000
    rk = rng.integers(0, 1000, nr).astype(np.int64)
    rk[rng.random(nr) < 0.35] = 77                              # ~21 000 duplicates of one key
    rk[rng.random(nr) < 0.2] = -1                               # and ~9 000 of the sentinel-valued key
    lk = rng.integers(-1, 1100, nl).astype(np.int64)
    lk[:5] = 77
    check_join(ctx, (lk, O.pack_mask(rng.random(nl) < 0.02), O.I64), nl, (rk, O.pack_mask(rng.random(nr) < 0.01), O.I64), nr)
    assert ctx.timings()["retries"] >= 1


@pytest.mark.parametrize("parts", [1, 5, 64])
def test_general_segmented_sort_build_path(ctx, parts):
    """The general build path forced on ordinary inputs, with few partitions so that every partition
    spans many 8192-row tiles (1 partition of 200 000 rows = 25 tiles, 5 merge passes)."""
    rng = np.random.default_rng(100 + parts)
    nl, nr = 150_000, 200_000
    rk = rng.integers(0, 120_000, nr).astype(np.int64) * 1_000_003
    lk = rng.integers(0, 130_000, nl).astype(np.int64) * 1_000_003
    ctx.set_option("join_generic", 1)
    ctx.set_option("partitions", parts)
    try:
        check_join(ctx, (lk, O.pack_mask(rng.random(nl) < 0.01), O.I64), nl, (rk, O.pack_mask(rng.random(nr) < 0.01), O.I64), nr)
    finally:
        ctx.set_option("join_generic", 0)
        ctx.set_option("partitions", 0)


def test_gathers_match_reference_fill(ctx):
    import torch
    rng = np.random.default_rng(9)
    nl, nr = 5000, 3000
    lk = (rng.integers(0, 2000, nl).astype(np.int64), None, O.I64)
    rk = (rng.integers(0, 2000, nr).astype(np.int64), None, O.I64)
    li, ri = ctx.join_indices(lk, nl, rk, nr, O.OUTER)
    d = "cuda:0"
    tl, tr = torch.from_numpy(li).to(d), torch.from_numpy(ri).to(d)
    srcs = [(rng.normal(size=nr), O.F64, 0.0), (rng.integers(-9, 9, nr).astype(np.int64), O.I64, 0),
            (rng.integers(0, 50, nr).astype(np.uint32), O.U32CODE, 0)]
    mask = O.pack_mask(rng.random(nr) < 0.2)
    for data, dt, fill in srcs:
        src_t = torch.from_numpy(data.view(np.int32) if dt == O.U32CODE else data).to(d)
        got = ctx.gather(src_t, torch.from_numpy(mask).to(d), tr, fill, dt).cpu().numpy()
        want = O.gather(data, mask, ri, fill, dt)
        np.testing.assert_array_equal(got.view(want.dtype), want)
    bits = np.packbits(rng.random(nl) < 0.5, bitorder="little")
    got = ctx.gather(torch.from_numpy(bits).to(d), None, tl, 0, O.BOOLBITS).cpu().numpy()
    np.testing.assert_array_equal(got, O.gather(bits, None, li, 0, O.BOOLBITS))


def test_fused_join_few_groups_uses_direct_path(ctx):
    """Few distinct group values after the join: the pair buffers must survive the partition-free path."""
    rng = np.random.default_rng(77)
    nb, npb = 100_000, 6_000_000
    rkeys = (rng.permutation(nb * 3)[:nb].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rg = rng.integers(0, 12, nb).astype(np.int64)
    lkeys = rkeys[rng.integers(0, nb, npb)].copy()
    lv = rng.normal(100, 10, npb)
    args = ((lkeys, None, O.I64), (lv, None, O.F64), npb, (rkeys, None, O.I64), (rg, None, O.I64), nb)
    assert_groupby_equal(ctx.join_groupby_sum(*args), O.join_groupby_sum(*args), [O.I64])


@pytest.mark.parametrize("gdtype", [O.I64, O.U32CODE])
def test_fused_join_groupby_sum(ctx, gdtype):
    rng = np.random.default_rng(5 + gdtype)
    nb, npb = 300_000, 2_000_000
    rkeys = (rng.permutation(nb * 3)[:nb].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rg = rng.integers(0, 5000, nb)
    rg = rg.astype(np.uint32) if gdtype == O.U32CODE else rg.astype(np.int64)
    lkeys = rkeys[rng.integers(0, nb, npb)].copy()
    lkeys[rng.random(npb) < 0.1] = -7                         # 10 % misses
    lv = rng.normal(100, 10, npb)
    args = ((lkeys, O.pack_mask(rng.random(npb) < 0.01), O.I64), (lv, O.pack_mask(rng.random(npb) < 0.05), O.F64), npb,
            (rkeys, None, O.I64), (rg, O.pack_mask(rng.random(nb) < 0.01), gdtype), nb)
    got = ctx.join_groupby_sum(*args)
    want = O.join_groupby_sum(*args)
    assert_groupby_equal(got, want, [gdtype])


def test_fused_join_duplicate_build_keys_and_sentinel(ctx):
    """Duplicate build keys give more pairs than probe rows (the pair buffer is re-sized and the probe
    repeated); the key equal to the table sentinel (-1) and null build keys take part as well."""
    rng = np.random.default_rng(321)
    nb, npb = 150_000, 700_000
    rkeys = rng.integers(0, 40_000, nb).astype(np.int64)       # ~3.75 duplicates per key
    rkeys[rng.random(nb) < 0.01] = -1
    rg = rng.integers(-50, 50, nb).astype(np.int64)
    lkeys = rng.integers(-1, 45_000, npb).astype(np.int64)
    lv = rng.integers(-1000, 1000, npb).astype(np.int64)       # i64 payload: sums must be bit-exact
    args = ((lkeys, O.pack_mask(rng.random(npb) < 0.01), O.I64), (lv, None, O.I64), npb,
            (rkeys, O.pack_mask(rng.random(nb) < 0.02), O.I64), (rg, None, O.I64), nb)
    got = ctx.join_groupby_sum(*args)
    want = O.join_groupby_sum(*args)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[0])


def test_join_output_beyond_row_limit_is_reported(ctx):
    """300 000 x 20 000 rows of one key = 6e9 output rows: a clear error, never a wrapped 32-bit count."""
    import pandrs_amd as pa
    lk = (np.zeros(300_000, np.int64), None, O.I64)
    rk = (np.zeros(20_000, np.int64), None, O.I64)
    with pytest.raises(pa.PandrsHipError) as e:
        ctx.join_indices(lk, 300_000, rk, 20_000, O.INNER)
    assert "2^32" in str(e.value)


def test_fused_join_general_fallback(ctx):
    """A build key with 20 000 duplicates does not fit a workgroup's LDS multimap: the fused entry falls
    back to the general join + pair gather; same for a forced run on ordinary data with null g / v."""
    rng = np.random.default_rng(55)
    nb, npb = 60_000, 40_000
    rkeys = rng.integers(0, 5000, nb).astype(np.int64)
    rkeys[rng.random(nb) < 0.34] = 4242
    rg = rng.integers(0, 40, nb).astype(np.int64)
    lkeys = rng.integers(0, 6000, npb).astype(np.int64)
    lkeys[:50] = 4242
    lv = rng.integers(-100, 100, npb).astype(np.int64)
    args = ((lkeys, None, O.I64), (lv, None, O.I64), npb, (rkeys, None, O.I64), (rg, None, O.I64), nb)
    assert_groupby_equal(ctx.join_groupby_sum(*args), O.join_groupby_sum(*args), [O.I64], int_exact_rows=[0])
    ctx.set_option("join_generic", 1)
    try:
        nb, npb = 50_000, 400_000
        rkeys = sparse(rng.permutation(nb * 2)[:nb])
        rgc = rng.integers(0, 300, nb).astype(np.uint32)
        lkeys = rkeys[rng.integers(0, nb, npb)].copy()
        args = ((lkeys, O.pack_mask(rng.random(npb) < 0.01), O.I64), (rng.normal(3, 1, npb), O.pack_mask(rng.random(npb) < 0.1), O.F64), npb,
                (rkeys, None, O.I64), (rgc, O.pack_mask(rng.random(nb) < 0.02), O.U32CODE), nb)
        assert_groupby_equal(ctx.join_groupby_sum(*args), O.join_groupby_sum(*args), [O.U32CODE])
    finally:
        ctx.set_option("join_generic", 0)


def sparse(ids):
    return (np.asarray(ids).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)


def test_join_full_size_properties():
    """BASELINE config 5's shape at a size the oracle cannot reach (100 M probe x 10 M unique build rows):
    size-independent properties.  Every probe row matches exactly the build row it was drawn from, in
    left order; a 10 % miss variant drops exactly the misses; the fused join -> groupby(g).sum(v) conserves
    the total of v over the matched rows and the per-group sums agree with a scatter-add of the same pairs."""
    import torch
    import pandrs_amd as pa
    d = "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(47)
    nb, npb, g = 10_000_000, 100_000_000, 100_000
    rk = torch.randperm(nb * 2, device=d, generator=gen)[:nb].to(torch.int64) * -7046029254386353131
    rg = torch.randint(0, g, (nb,), device=d, generator=gen, dtype=torch.int64)
    pick = torch.randint(0, nb, (npb,), device=d, generator=gen)
    lk = rk[pick]
    lv = torch.randn(npb, device=d, generator=gen, dtype=torch.float64) * 10 + 100
    c = pa.Context(0)
    try:
        li, ri = c.join_indices((lk, None, pa.I64), npb, (rk, None, pa.I64), nb, pa.INNER)
        assert li.numel() == npb and bool((li == torch.arange(npb, device=d)).all()) and bool((ri == pick).all())
        miss = torch.rand(npb, device=d, generator=gen) < 0.1
        lk2 = torch.where(miss, lk ^ 1, lk)                     # odd multiplier => rk are distinct mod 2 patterns; most flips miss
        li2, ri2 = c.join_indices((lk2, None, pa.I64), npb, (rk, None, pa.I64), nb, pa.LEFT)
        assert li2.numel() == npb and bool((li2 == torch.arange(npb, device=d)).all())
        hit = ri2 >= 0
        assert bool((rk[ri2[hit]] == lk2[hit]).all())
        assert int((~hit).sum()) > 0.05 * npb
        del li, ri, li2, ri2, lk2, miss, hit
        kc, kn, oa = c.join_groupby_sum((lk, None, pa.I64), (lv, None, pa.F64), npb, (rk, None, pa.I64), (rg, None, pa.I64), nb)
        assert int(kn.sum()) == 0 and kc.shape[1] == torch.unique(rg[pick]).numel()
        want = torch.zeros(g, dtype=torch.float64, device=d).index_add_(0, rg[pick], lv)
        got = torch.zeros(g, dtype=torch.float64, device=d)
        got[kc[0]] = oa[0]
        assert float((got - want).abs().max()) <= 1e-9 * float(want.abs().max())
        assert abs(float(oa[0].sum()) - float(lv.sum())) <= 1e-9 * abs(float(lv.sum()))
    finally:
        c.close()


@pytest.mark.parametrize("nb", [50_000_000, 60_000_000])
def test_config5_shard_full_size_properties(nb):
    """BASELINE config 5's per-GPU shard after the build side's all-gather: 62.5 M probe rows x 50 M unique build rows
    through the fused join -> groupby-sum — right at the ~55 M-row limit of the LDS-partition path — and 60 M build
    rows, which must take the general-join fallback (limits).  Size-independent properties: the groups are
    exactly the g values hit, the total of v is conserved, the per-group sums equal a scatter-add of the same pairs."""
    import torch
    import pandrs_amd as pa
    d = "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(48 + nb % 7)
    npb, g = 62_500_000, 100_000
    rk = torch.randperm(nb, device=d, generator=gen) * -7046029254386353131        # odd multiplier: unique keys
    rg = torch.randint(0, g, (nb,), device=d, generator=gen, dtype=torch.int64)
    pick = torch.randint(0, nb, (npb,), device=d, generator=gen)
    lk = rk[pick]
    miss = torch.rand(npb, device=d, generator=gen) < 0.1                          # the 90 %-hit variant of SURVEY 8d
    lk = torch.where(miss, lk ^ 1, lk)
    hit = ~miss | (lk == rk[pick])
    lv = torch.randn(npb, device=d, generator=gen, dtype=torch.float64) * 10 + 100
    c = pa.Context(0)
    try:
        kc, kn, oa = c.join_groupby_sum((lk, None, pa.I64), (lv, None, pa.F64), npb, (rk, None, pa.I64), (rg, None, pa.I64), nb)
        # a flipped key may by chance equal another build key: settle the truth with the index join on a sample
        want = torch.zeros(g, dtype=torch.float64, device=d).index_add_(0, rg[pick][hit], lv[hit])
        got = torch.zeros(g, dtype=torch.float64, device=d)
        got[kc[0]] = oa[0]
        assert int(kn.sum()) == 0 and kc.shape[1] == torch.unique(kc[0]).numel()
        stray = float((got - want).abs().max())
        assert stray <= 1e-9 * float(want.abs().max()) + 200.0 * 3, stray          # <= a handful of chance hits among 6 M flipped keys
        assert abs(float(oa[0].sum()) - float(lv[hit].sum())) <= 1e-9 * abs(float(lv.sum())) + 200.0 * 3
    finally:
        c.close()


@pytest.mark.parametrize("case", ["unique", "sentinel_nulls_u32", "duplicates_decline", "sampled_probe_side", "pair_partitions", "pair_regions_overflow"])
def test_fused_join_large_build_path_with_l2_resident_table_regions(ctx, case):
    """build sides beyond ~1024 LDS partitions keep their hash-table regions in global memory
    (8 fine regions = one coarse partition, probed by one XCD group out of its L2) and partition the probe side only
    coarsely.  Forced here at sizes the oracle finishes in seconds (join_no_l2 = -1); unique build keys only — a
    duplicate build key makes the path decline and the LDS-multimap path answers."""
    rng = np.random.default_rng(900 + len(case))
    big = case in ("sampled_probe_side", "pair_partitions", "pair_regions_overflow")
    nb, npb = (300_000, 9_000_000) if big else (400_000, 3_000_000)
    if case in ("pair_partitions", "pair_regions_overflow"):
        nb = 1_500_000                      # enough build rows per (pair partition, XCD group) region for the sampled plan to hold
    rkeys = sparse(rng.permutation(nb * 3)[:nb])
    gdt = O.U32CODE if case == "sentinel_nulls_u32" else O.I64
    rg = rng.integers(0, 5000, nb)
    rg = rg.astype(np.uint32) if gdt == O.U32CODE else rg.astype(np.int64)
    rmask = lmask = vmask = gmask = None
    if case == "sentinel_nulls_u32":
        rkeys[7] = -1                                               # the table sentinel's bit pattern as a build key
        rmask = O.pack_mask(rng.random(nb) < 0.01)
        gmask = O.pack_mask(rng.random(nb) < 0.01)
    if case == "duplicates_decline":
        rkeys[1000:1010] = rkeys[5]
    if case == "pair_partitions":
        rg[::1000] = -1                                             # a GROUP key with the table sentinel's bit pattern, through the partitioned pair output
    lkeys = rkeys[rng.integers(0, nb, npb)].copy()
    lkeys[rng.random(npb) < 0.1] = -7                               # 10 % misses
    if case == "sentinel_nulls_u32":
        lkeys[rng.random(npb) < 0.001] = -1
        lmask = O.pack_mask(rng.random(npb) < 0.01)
        vmask = O.pack_mask(rng.random(npb) < 0.05)
    lv = rng.integers(-1000, 1000, npb).astype(np.int64) if case != "sampled_probe_side" else rng.normal(100, 10, npb)
    vdt = O.I64 if case != "sampled_probe_side" else O.F64
    # from 8.4 M probe rows on, the probe writes its (g, v) pairs straight into the groupby engine's partitions (regions sized from
    # a 1-in-64 sample); "pair_regions_overflow" plans them for an eighth of the rows: they overflow and the plain emission answers
    pairpart = {"pair_partitions": 0, "pair_regions_overflow": 2}.get(case, 1 if case == "sampled_probe_side" else 0)
    args = ((lkeys, lmask, O.I64), (lv, vmask, vdt), npb, (rkeys, rmask, O.I64), (rg, gmask, gdt), nb)
    want = O.join_groupby_sum(*args)
    ctx.set_option("join_no_l2", -1)
    ctx.set_option("join_no_pairpart", pairpart)
    try:
        got = ctx.join_groupby_sum(*args)
        parts = ctx.timings()["n_partitions"]
        assert ctx.timings()["retries"] == (1 if case == "pair_regions_overflow" else 0)
    finally:
        ctx.set_option("join_no_l2", 0)
        ctx.set_option("join_no_pairpart", 0)
    assert_groupby_equal(got, want, [gdt], int_exact_rows=[0] if vdt == O.I64 else [])
    # the L2 path reports the probe side's coarse fan-out (a multiple of 8); a declined call reports the LDS path's fan-out
    if case == "duplicates_decline":
        assert parts % 8 != 0
    else:
        assert parts % 8 == 0 and parts >= nb // 6144 // 8


def test_config5_full_size_on_one_gpu_takes_the_l2_path():
    """BASELINE config 5 whole on ONE GPU (500 M probe rows x 50 M unique build rows -> 100 K groups): the probe side is
    ten times the build side, so the fused join takes the L2-resident-region path.  Properties: exactly the g values hit
    are groups, the per-group sums equal a scatter-add of the same pairs (f64 sums in another order: 1e-9 relative), and
    the LDS-multimap path gives the same answer."""
    import torch
    import pandrs_amd as pa
    d = "cuda:0"
    gen = torch.Generator(device=d)
    gen.manual_seed(4848)
    nb, npb, g = 50_000_000, 500_000_000, 100_000
    rk = torch.randperm(nb, device=d, generator=gen) * -7046029254386353131
    rg = torch.randint(0, g, (nb,), device=d, generator=gen, dtype=torch.int64)
    pick = torch.randint(0, nb, (npb,), device=d, generator=gen)
    lk = rk[pick]
    lv = torch.randn(npb, device=d, generator=gen, dtype=torch.float64) * 10 + 100
    want = torch.zeros(g, dtype=torch.float64, device=d).index_add_(0, rg[pick], lv)
    del pick
    c = pa.Context(0)
    try:
        res = {}
        for mode in (0, 1):
            c.set_option("join_no_l2", mode)
            kc, kn, oa = c.join_groupby_sum((lk, None, pa.I64), (lv, None, pa.F64), npb, (rk, None, pa.I64), (rg, None, pa.I64), nb)
            t = c.timings()
            assert int(kn.sum()) == 0 and kc.shape[1] == g == torch.unique(kc[0]).numel()
            got = torch.zeros(g, dtype=torch.float64, device=d)
            got[kc[0]] = oa[0]
            assert float((got - want).abs().max()) <= 1e-9 * float(want.abs().max())
            res[mode] = (got, t["n_partitions"], t["total_ms"])
        assert res[0][1] == 1024 and res[1][1] == 8192                 # the L2 path's coarse probe-side fan-out vs the LDS path's
        assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-9 * float(want.abs().max())
        print("C5 on one GPU: L2 path %.2f ms, LDS-multimap path %.2f ms" % (res[0][2], res[1][2]))
    finally:
        c.close()


def test_fused_join_prepartitioned_pairs_fall_back_when_the_group_estimate_is_far_too_low():
    """with pairs written straight into the groupby engine's partitions, a full LDS table used to
    surface as PANDRS_HIP_ERR_COMPUTATION.  Build side: 60 % of the rows in 16 hot groups, the rest one group EACH
    (3.2 M groups) — the strided sample sees mostly repeats and estimates ~0.1 M groups, the pair fan-out stays at 256,
    every table overflows.  The call must answer (through the plain pair emission), exactly."""
    import torch
    import pandrs_amd as pa
    d = "cuda:0"
    nr, nl = 8_000_000, 17_000_000
    rng = np.random.default_rng(31)
    rk = (rng.permutation(nr).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    hot = rng.random(nr) < 0.6
    rg = np.where(hot, rng.integers(0, 16, nr), 1_000 + np.arange(nr)).astype(np.int64)
    pick = rng.integers(0, nr, nl)
    lk = rk[pick]
    lv = rng.integers(-8, 9, nl).astype(np.float64)          # small integers: the sums are exact in any order
    dev = lambda a: torch.from_numpy(a).to(d)
    c = pa.Context(0)
    groups, inv = np.unique(rg[pick], return_inverse=True)
    want = np.bincount(inv, weights=lv, minlength=len(groups))
    try:
        # round 3, late: the full tables hand their unplaced pairs to a run of their own and the call answers at once (retries 0);
        # with that switched off the engine refuses the pre-partitioned pairs and the whole call is repeated with the plain emission (2)
        # (and the estimate's Chao1 term now sees the 3.2 M groups in the first place: the model alone — no_chao — reads ~0.1 M)
        for no_chao, no_overflow_run, retries in ((0, 0, 0), (1, 0, 0), (1, 1, 2)):
            c.set_option("no_chao", no_chao); c.set_option("no_overflow_run", no_overflow_run)
            kc, kn, oa = c.join_groupby_sum((dev(lk), None, pa.I64), (dev(lv), None, pa.F64), nl, (dev(rk), None, pa.I64), (dev(rg), None, pa.I64), nr)
            t = c.timings()
            assert t["n_partitions"] > 0                      # the L2-region path (not the general fallback)
            assert t["retries"] == retries, t
            got_k = kc[0].cpu().numpy().view(np.int64)
            order = np.argsort(got_k)
            np.testing.assert_array_equal(got_k[order], groups)
            np.testing.assert_array_equal(oa[0].cpu().numpy()[order], want)
            assert int(kn.sum()) == 0
    finally:
        c.close()


def test_two_pass_partition_gives_the_same_join_and_groups(ctx):
    """From a fan-out of 6144 the exact radix partition moves the rows twice (64 buckets, then the rest over the bucket-sorted
    rows; partition.hip).  Forced here at small sizes (explicit threshold + forced fan-out) on both sides of the fused join, with
    NULL keys, the sentinel-valued key and masked payloads, against the oracle and against the single pass."""
    rng = np.random.default_rng(77)
    nb, npb = 500_000, 900_000
    rkeys = (rng.permutation(nb * 3)[:nb].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rkeys[7] = -1
    rg = rng.integers(0, 3000, nb).astype(np.int64)
    lkeys = rkeys[rng.integers(0, nb, npb)].copy()
    lkeys[rng.random(npb) < 0.1] = -7
    lv = rng.integers(-1000, 1000, npb).astype(np.int64)       # i64 payload: sums must be bit-exact
    args = ((lkeys, O.pack_mask(rng.random(npb) < 0.01), O.I64), (lv, O.pack_mask(rng.random(npb) < 0.05), O.I64), npb,
            (rkeys, O.pack_mask(rng.random(nb) < 0.01), O.I64), (rg, O.pack_mask(rng.random(nb) < 0.01), O.I64), nb)
    want = O.join_groupby_sum(*args)
    ctx.set_option("partitions", 1000)                          # (any fan-out: the first pass's buckets are partition ids shifted down)
    try:
        ctx.set_option("two_pass_min_p", 512)
        got = ctx.join_groupby_sum(*args)
        assert ctx.timings()["phase_ms"].get("prepartition", 0) > 0
        ctx.set_option("two_pass", -1)
        got1 = ctx.join_groupby_sum(*args)
        assert ctx.timings()["phase_ms"].get("prepartition", 0) == 0
    finally:
        ctx.set_option("partitions", 0); ctx.set_option("two_pass_min_p", 0); ctx.set_option("two_pass", 0)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[0])
    assert_groupby_equal(got1, want, [O.I64], int_exact_rows=[0])


def test_a_hot_probe_key_reroutes_the_fused_join_to_the_l2_region_path(ctx):
    """The LDS-multimap path gives every partition ONE workgroup: half the probe rows on one build key is one workgroup walking
    them alone (62.5 M x 50 M: 46 ms instead of 4.3).  The probe kernel flags such a partition and skips it, and the call is
    repeated on the L2-region path, whose probe tiles are dealt by ticket (12 ms).  Same sums as the oracle; retries = 0 (the
    L2 path's own figure) and the fan-out reported is the L2 path's coarse one."""
    rng = np.random.default_rng(4242)
    nb, npb = 3_000_000, 4_400_000
    rkeys = (rng.permutation(nb * 2)[:nb].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rg = rng.integers(0, 5000, nb).astype(np.int64)
    pick = np.where(rng.random(npb) < 0.5, 12_345, rng.integers(0, nb, npb))
    lkeys = rkeys[pick].copy()
    lkeys[rng.random(npb) < 0.05] = -7                          # 5 % misses
    lv = rng.integers(-1000, 1000, npb).astype(np.int64)       # i64 payload: sums must be bit-exact
    args = ((lkeys, O.pack_mask(rng.random(npb) < 0.01), O.I64), (lv, None, O.I64), npb,
            (rkeys, None, O.I64), (rg, O.pack_mask(rng.random(nb) < 0.01), O.I64), nb)
    want = O.join_groupby_sum(*args)
    got = ctx.join_groupby_sum(*args)
    t = ctx.timings()
    ctx.set_option("join_no_l2", 1)                             # never the L2 path: the multimap path walks the hot partition after all
    try:
        got1 = ctx.join_groupby_sum(*args)
        t1 = ctx.timings()
    finally:
        ctx.set_option("join_no_l2", 0)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[0])
    assert_groupby_equal(got1, want, [O.I64], int_exact_rows=[0])
    assert t["n_partitions"] != t1["n_partitions"], (t["n_partitions"], t1["n_partitions"])     # coarse L2 fan-out vs the multimap's
