"""The multi-GPU drivers (pandrs_amd/dist.py) over the real HIP engine and RCCL, rehearsed with the
one GPU a test box has: a world-size-1 "nccl" process group (all-gather / all-to-all degenerate to
device copies, every other line of the driver runs as on 8 GPUs).  The N > 1 routing itself is
covered on CPU by tests/test_dist_gloo.py (gloo, world 2 and 3)."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import assert_groupby_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    yield dist
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def ctx():
    import pandrs_amd
    c = pandrs_amd.Context(0)
    yield c
    c.close()


def _dev(a):
    import torch
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_distributed_groupby_world1_rccl(ctx, pg):
    from pandrs_amd.dist import DistributedGroupBy
    rng = np.random.default_rng(5)
    n, g = 2_000_000, 150_000
    k = (rng.integers(0, g, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    km = O.pack_mask(rng.random(n) < 0.001)
    v = rng.normal(100, 10, n)
    aggs = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (0, O.MAX), (0, O.COUNT)]
    d = DistributedGroupBy(ctx, pg, "cuda:0")
    kc, kn, oa = d.groupby_agg([(_dev(k), _dev(km), O.I64)], n, [(_dev(v), None, O.F64)], aggs)
    got = (kc.cpu().numpy().view(np.uint64), kn.cpu().numpy(), oa.cpu().numpy())
    want = O.groupby_agg([(k, km, O.I64)], n, [(v, None, O.F64)], aggs)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[2, 3, 4])
    assert d.last_timings["records_sent"] == want[0].shape[1]


def test_distributed_join_groupby_world1_rccl(ctx, pg):
    from pandrs_amd.dist import DistributedJoinGroupBy
    rng = np.random.default_rng(6)
    nl, nr, g = 3_000_001, 300_003, 20_000      # nr not a multiple of 8: build shard gets NULL-key padding
    rk = (rng.permutation(4 * nr)[:nr].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rg = rng.integers(0, g, nr).astype(np.int64)
    rgm = O.pack_mask(rng.random(nr) < 0.001)
    lk = np.where(rng.random(nl) < 0.9, rk[rng.integers(0, nr, nl)], rng.integers(1, 1 << 40, nl))
    lkm = O.pack_mask(rng.random(nl) < 0.001)
    lv = rng.normal(10, 3, nl)
    d = DistributedJoinGroupBy(ctx, pg, "cuda:0")
    kc, kn, oa = d.join_groupby_sum((_dev(lk), _dev(lkm), O.I64), (_dev(lv), None, O.F64), nl,
                                    (_dev(rk), None, O.I64), (_dev(rg), _dev(rgm), O.I64), nr)
    got = (kc.cpu().numpy().view(np.uint64), kn.cpu().numpy(), oa.cpu().numpy())
    want = O.join_groupby_sum((lk, lkm, O.I64), (lv, None, O.F64), nl, (rk, None, O.I64), (rg, rgm, O.I64), nr)
    assert got[0].shape[1] == want[0].shape[1]
    assert_groupby_equal(got, want, [O.I64])
    assert set(d.last_wall_ms) == {"allgather_build", "local_join_groupby", "exchange_merge"}


def test_shuffle_split_buckets_rows_by_owner(ctx):
    """pandrs_hip_shuffle_split: every input row appears exactly once (null keys on the last rank or
    dropped), rank-contiguous, all rows of one key on one rank, payload and null bytes travel with
    their row; bytes_to_bitmap round-trips."""
    from oracle import oracle_np as ONP
    rng = np.random.default_rng(17)
    n, ranks = 500_003, 5
    k = (rng.integers(0, 20_000, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    km = O.pack_mask(rng.random(n) < 0.01)
    p0 = np.arange(n, dtype=np.int64)                            # the original row index as payload
    p1 = rng.normal(size=n)
    m1 = O.pack_mask(rng.random(n) < 0.2)
    p2 = rng.integers(0, 1000, n).astype(np.uint32)
    for drop in (False, True):
        cells, knull, pays, pnull, counts = ctx.shuffle_split((k, km, O.I64), [(p0, None, O.I64), (p1, m1, O.F64), (p2, None, O.U32CODE)],
                                                              n, ranks, drop_null_keys=drop)
        nul, cell = ONP.key_cells((k, km, O.I64), n)
        rows = pays[0].view(np.int64)
        assert sum(counts) == len(cells) == (n - int(nul.sum()) if drop else n)
        np.testing.assert_array_equal(np.sort(rows), np.flatnonzero(nul == 0) if drop else np.arange(n))
        np.testing.assert_array_equal(knull, nul[rows])
        np.testing.assert_array_equal(cells[knull == 0], cell[rows][knull == 0])
        np.testing.assert_array_equal(pays[1].view(np.float64), p1[rows])
        np.testing.assert_array_equal(pays[2], p2[rows].astype(np.uint64))
        np.testing.assert_array_equal(pnull[1], np.unpackbits(m1, bitorder="little")[:n][rows])
        assert pnull[0] is None and pnull[2] is None
        owner = np.repeat(np.arange(ranks), counts)
        nn = knull == 0
        first_owner = {}
        ks, os_ = cells[nn], owner[nn]
        o = np.argsort(ks, kind="stable")
        same = ks[o][1:] == ks[o][:-1]
        assert np.all(os_[o][1:][same] == os_[o][:-1][same]), "a key's rows must share one owner"
        assert np.all(owner[~nn] == ranks - 1)
        assert min(counts) > 0.5 * n / ranks                      # balanced
    flags = (rng.random(1003) < 0.3).astype(np.uint8) * 7
    np.testing.assert_array_equal(ctx.bytes_to_bitmap(flags), np.packbits(flags != 0, bitorder="little"))


def test_distributed_non_mergeable_aggregates_world1_rccl(ctx, pg):
    """Median / Std / Var across ranks go through the row shuffle; here with one rank over RCCL."""
    from pandrs_amd.dist import DistributedGroupBy
    rng = np.random.default_rng(8)
    n, g = 1_000_000, 30_000
    k = (rng.integers(0, g, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    km = O.pack_mask(rng.random(n) < 0.001)
    v0 = np.round(rng.normal(100, 10, n), 2)
    v1 = rng.integers(-1000, 1000, n).astype(np.int64)
    m1 = O.pack_mask(rng.random(n) < 0.1)
    aggs = [(0, O.MEDIAN), (1, O.MEDIAN), (0, O.STD), (1, O.VAR), (0, O.SUM), (1, O.COUNT)]
    d = DistributedGroupBy(ctx, pg, "cuda:0")
    kc, kn, oa = d.groupby_agg([(_dev(k), _dev(km), O.I64)], n, [(_dev(v0), None, O.F64), (_dev(v1), _dev(m1), O.I64)], aggs)
    got = (kc.cpu().numpy().view(np.uint64), kn.cpu().numpy(), oa.cpu().numpy())
    want = O.groupby_agg([(k, km, O.I64)], n, [(v0, None, O.F64), (v1, m1, O.I64)], aggs)
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[0, 1, 5])


def test_distributed_join_shuffle_strategy_world1_rccl(ctx, pg):
    from pandrs_amd.dist import DistributedJoinGroupBy
    rng = np.random.default_rng(9)
    nl, nr, g = 2_000_001, 200_003, 5_000
    rk = (rng.permutation(4 * nr)[:nr].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rg = rng.integers(0, g, nr).astype(np.uint32)               # string-pool codes as the group column
    rgm = O.pack_mask(rng.random(nr) < 0.001)
    lk = np.where(rng.random(nl) < 0.9, rk[rng.integers(0, nr, nl)], rng.integers(1, 1 << 40, nl))
    lkm = O.pack_mask(rng.random(nl) < 0.001)
    lv = rng.normal(10, 3, nl)
    lvm = O.pack_mask(rng.random(nl) < 0.01)
    d = DistributedJoinGroupBy(ctx, pg, "cuda:0")
    kc, kn, oa = d.join_groupby_sum((_dev(lk), _dev(lkm), O.I64), (_dev(lv), _dev(lvm), O.F64), nl,
                                    (_dev(rk), None, O.I64), (_dev(rg.view(np.int32)), _dev(rgm), O.U32CODE), nr, strategy="shuffle")
    got = (kc.cpu().numpy().view(np.uint64), kn.cpu().numpy(), oa.cpu().numpy())
    want = O.join_groupby_sum((lk, lkm, O.I64), (lv, lvm, O.F64), nl, (rk, None, O.I64), (rg, rgm, O.U32CODE), nr)
    assert got[0].shape[1] == want[0].shape[1]
    assert_groupby_equal(got, want, [O.U32CODE])
    assert "shuffle_rows" in d.last_wall_ms


def test_distributed_multi_key_world1_rccl(ctx, pg):
    """Composite keys across ranks: shuffled on pandrs_hip_key_hash_cells of the tuple, key columns as payload."""
    from pandrs_amd.dist import DistributedGroupBy
    rng = np.random.default_rng(10)
    n = 800_000
    k0 = (rng.integers(0, 500, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    m0 = O.pack_mask(rng.random(n) < 0.01)
    k1 = rng.integers(0, 9, n).astype(np.uint32)
    k2 = rng.choice(np.array([0.5, -0.0, 0.0, np.nan, 3.25]), n)
    v = rng.normal(0, 1, n)
    aggs = [(0, O.SUM), (0, O.COUNT), (0, O.MEDIAN), (0, O.STD)]
    d = DistributedGroupBy(ctx, pg, "cuda:0")
    kc, kn, oa = d.groupby_agg([(_dev(k0), _dev(m0), O.I64), (_dev(k1.view(np.int32)), None, O.U32CODE), (_dev(k2), None, O.F64)],
                               n, [(_dev(v), None, O.F64)], aggs)
    got = (kc.cpu().numpy().view(np.uint64), kn.cpu().numpy(), oa.cpu().numpy())
    want = O.groupby_agg([(k0, m0, O.I64), (k1, None, O.U32CODE), (k2, None, O.F64)], n, [(v, None, O.F64)], aggs)
    assert got[0].shape[1] == want[0].shape[1]
    assert_groupby_equal(got, want, [O.I64, O.U32CODE, O.F64], int_exact_rows=[1, 2])
    # the hash cells are a function of the tuple alone (every rank computes the same owner)
    h = ctx.key_hash_cells([(k0, m0, O.I64), (k1, None, O.U32CODE), (k2, None, O.F64)], n)
    from oracle import oracle_np as ONP
    comp = np.stack([c for col in [(k0, m0, O.I64), (k1, None, O.U32CODE), (k2, None, O.F64)] for c in (ONP.key_cells(col, n)[0].astype(np.uint64), ONP.key_cells(col, n)[1])], 1)
    _, inv = np.unique(comp, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    first = np.zeros(inv.max() + 1, np.uint64)
    first[inv] = h
    np.testing.assert_array_equal(first[inv], h)


# ---- two REAL ranks on the one GPU: the HIP engine on both sides, gloo as the transport ---------------
def _two_rank_worker(rank, world, port, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import pandrs_amd
    from pandrs_amd.dist import DistributedGroupBy, DistributedJoinGroupBy
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = pandrs_amd.Context(0)
    rng = np.random.default_rng(2024)
    n, g = 600_000, 40_000
    k = (rng.integers(0, g, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    km = rng.random(n) < 0.002
    v0 = np.round(rng.normal(100, 10, n), 1)
    v1 = rng.integers(-1000, 1000, n).astype(np.int64)
    m1 = rng.random(n) < 0.1
    lo, hi = (n // world // 8 * 8) * rank, n if rank == world - 1 else (n // world // 8 * 8) * (rank + 1)
    pb = lambda m: np.packbits(m, bitorder="little")
    d = DistributedGroupBy(c, dist, "cpu")
    out = {}
    for name, aggs in (("merge", [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 4)]), ("shuffle", [(0, 7), (1, 7), (0, 5), (1, 0)])):
        kc, kn, oa = d.groupby_agg([(k[lo:hi], pb(km[lo:hi]), 0)], hi - lo, [(v0[lo:hi], None, 1), (v1[lo:hi], pb(m1[lo:hi]), 0)], aggs)
        out[name + "_kc"], out[name + "_kn"], out[name + "_oa"] = np.asarray(kc).view(np.uint64), np.asarray(kn), np.asarray(oa)
    nr = 50_000
    rk = (rng.permutation(4 * nr)[:nr].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rg = rng.integers(0, 900, nr).astype(np.int64)
    lk = np.where(rng.random(n) < 0.9, rk[rng.integers(0, nr, n)], rng.integers(1, 1 << 40, n))
    r0, r1 = nr * rank // world, nr * (rank + 1) // world
    for strategy in ("allgather", "shuffle"):
        j = DistributedJoinGroupBy(c, dist, "cpu")
        kc, kn, oa = j.join_groupby_sum((lk[lo:hi], None, 0), (v0[lo:hi], None, 1), hi - lo, (rk[r0:r1], None, 0), (rg[r0:r1], None, 0), r1 - r0, strategy=strategy)
        out["join_" + strategy + "_kc"], out["join_" + strategy + "_kn"], out["join_" + strategy + "_oa"] = np.asarray(kc).view(np.uint64), np.asarray(kn), np.asarray(oa)
    np.savez(os.path.join(outdir, "t%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()
    c.close()


def test_two_real_ranks_share_the_gpu_over_gloo(tmp_path):
    """Two processes, each with its own HIP context on cuda:0 and its own row range (host columns), exchange
    over gloo: partial-state merge, row shuffle (Median / Std) and both distributed join strategies with the
    REAL engine on both sides of a real exchange.  (RCCL needs one GPU per rank; the driver has those.)"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    parts = [np.load(os.path.join(tmp_path, "t%d.npz" % r)) for r in range(2)]
    cat = lambda name: tuple(np.concatenate([p[name + s_] for p in parts], axis=1) for s_ in ("_kc", "_kn", "_oa"))
    rng = np.random.default_rng(2024)
    n, g = 600_000, 40_000
    k = (rng.integers(0, g, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    km = rng.random(n) < 0.002
    v0 = np.round(rng.normal(100, 10, n), 1)
    v1 = rng.integers(-1000, 1000, n).astype(np.int64)
    m1 = rng.random(n) < 0.1
    keys, vals = [(k, O.pack_mask(km), O.I64)], [(v0, None, O.F64), (v1, O.pack_mask(m1), O.I64)]
    aggs = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 4)]
    assert_groupby_equal(cat("merge"), O.groupby_agg(keys, n, vals, aggs), [O.I64], int_exact_rows=[2, 3, 4, 5])
    aggs = [(0, 7), (1, 7), (0, 5), (1, 0)]
    assert_groupby_equal(cat("shuffle"), O.groupby_agg(keys, n, vals, aggs), [O.I64], int_exact_rows=[0, 1, 3])
    nr = 50_000
    rk = (rng.permutation(4 * nr)[:nr].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rg = rng.integers(0, 900, nr).astype(np.int64)
    lk = np.where(rng.random(n) < 0.9, rk[rng.integers(0, nr, n)], rng.integers(1, 1 << 40, n))
    want = O.join_groupby_sum((lk, None, O.I64), (v0, None, O.F64), n, (rk, None, O.I64), (rg, None, O.I64), nr)
    for strategy in ("allgather", "shuffle"):
        got = cat("join_" + strategy)
        assert got[0].shape[1] == want[0].shape[1], strategy
        assert_groupby_equal(got, want, [O.I64])


# ---- the exchange INSIDE the library (pandrs_hip_dist_*: RCCL behind the C ABI), world-size-1 rehearsal -------------
@pytest.fixture(scope="module")
def cctx():
    import pandrs_amd
    c = pandrs_amd.Context(0)
    c.comm_init(pandrs_amd.Context.comm_unique_id(), 0, 1)
    yield c
    c.close()


def test_in_library_exchange_groupby_world1(cctx):
    """pandrs_hip_dist_groupby_agg: partials -> owner split -> count all-gather -> grouped ncclSend / ncclRecv -> merge,
    all inside libpandrs_hip.so.  One value column with a null mask, one without, a null-key group."""
    rng = np.random.default_rng(15)
    n, g = 2_000_000, 150_000
    k = (rng.integers(0, g, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    km = O.pack_mask(rng.random(n) < 0.001)
    v, w = rng.normal(100, 10, n), rng.normal(-3, 1, n)
    wm = O.pack_mask(rng.random(n) < 0.05)
    aggs = [(0, O.SUM), (0, O.MEAN), (0, O.MIN), (0, O.MAX), (0, O.COUNT), (1, O.MEAN), (1, O.MAX)]
    for dev in (True, False):
        cols = ([(_dev(k), _dev(km), O.I64)], [(_dev(v), None, O.F64), (_dev(w), _dev(wm), O.F64)]) if dev else \
               ([(k, km, O.I64)], [(v, None, O.F64), (w, wm, O.F64)])
        cctx.dist_groupby_compute(cols[0], n, cols[1], aggs)
        kc, kn, oa = cctx.groupby_fetch(to_device=False)
        want = O.groupby_agg([(k, km, O.I64)], n, [(v, None, O.F64), (w, wm, O.F64)], aggs)
        assert_groupby_equal((kc, kn, oa), want, [O.I64], int_exact_rows=[2, 3, 4, 6])
    # dist.py is a thin caller of it when the engine holds a communicator
    from pandrs_amd.dist import DistributedGroupBy

    class _NoDist:                      # the library does the exchange: torch.distributed must not be touched
        def get_world_size(self): return 1
        def get_rank(self): return 0
    d = DistributedGroupBy(cctx, _NoDist(), "cuda:0")
    kc, kn, oa = d.groupby_agg([(_dev(k), _dev(km), O.I64)], n, [(_dev(v), None, O.F64), (_dev(w), _dev(wm), O.F64)], aggs)
    assert_groupby_equal((kc.cpu().numpy().view(np.uint64), kn.cpu().numpy(), oa.cpu().numpy()), want, [O.I64], int_exact_rows=[2, 3, 4, 6])
    # what partial states cannot express takes the row shuffle inside the same entry point (world 1: RCCL sends to itself)
    gen = [(0, O.MEDIAN), (1, O.STD), (0, O.NUNIQUE), (1, O.SUM)]
    cctx.dist_groupby_compute([(_dev(k), _dev(km), O.I64)], n, [(_dev(v), None, O.F64), (_dev(w), _dev(wm), O.F64)], gen)
    assert_groupby_equal(cctx.groupby_fetch(to_device=False), O.groupby_agg([(k, km, O.I64)], n, [(v, None, O.F64), (w, wm, O.F64)], gen),
                         [O.I64], int_exact_rows=[0, 2])
    codes = (np.abs(k) % 5).astype(np.uint32)
    two = [(0, O.SUM), (0, O.COUNT)]
    cctx.dist_groupby_compute([(codes, None, O.U32CODE), (k % 100, km, O.I64)], n, [(v, None, O.F64)], two)       # host shards, two keys
    assert_groupby_equal(cctx.groupby_fetch(to_device=False), O.groupby_agg([(codes, None, O.U32CODE), (k % 100, km, O.I64)], n, [(v, None, O.F64)], two),
                         [O.U32CODE, O.I64], int_exact_rows=[1])
    with pytest.raises(Exception):
        cctx.dist_groupby_compute([(k, None, O.I64)], n, [(v, None, O.F64)], [(0, O.FIRST)])      # needs the global row order


def test_in_library_exchange_join_groupby_world1(cctx):
    """pandrs_hip_dist_join_groupby_sum: build side all-gathered (padding rows are NULL keys), local fused join ->
    groupby-sum, partial sums through the in-library exchange."""
    rng = np.random.default_rng(16)
    nl, nr, g = 1_500_003, 120_001, 700                       # lengths that need padding to a multiple of 8
    rkeys = (rng.permutation(4 * nr)[:nr].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rgrp = rng.integers(0, g, nr).astype(np.int64)
    lkeys = np.where(rng.random(nl) < 0.9, rkeys[rng.integers(0, nr, nl)], rng.integers(1, 1 << 40, nl))
    lval = rng.normal(10, 3, nl)
    rkm = O.pack_mask(rng.random(nr) < 0.01)
    kc, kn, oa = cctx.dist_join_groupby_sum((_dev(lkeys), None, O.I64), (_dev(lval), None, O.F64), nl,
                                            (_dev(rkeys), _dev(rkm), O.I64), (_dev(rgrp), None, O.I64), nr)
    want = O.join_groupby_sum((lkeys, None, O.I64), (lval, None, O.F64), nl, (rkeys, rkm, O.I64), (rgrp, None, O.I64), nr)
    got = (kc.cpu().numpy().view(np.uint64), kn.cpu().numpy(), oa.cpu().numpy())
    assert_groupby_equal(got, want, [O.I64])


# ---- the in-library exchange at world 2 and 3 on ONE GPU: host-callback transport (gloo underneath) ------------------
# exchange_records / dist_groupby_impl / dist_join_groupby_sum had only ever run with world = 1, where a
# rank sends to itself.  pandrs_hip_comm_adopt_transport puts the same C++ code over any fabric; here every rank is a fresh
# process with its own HIP context on cuda:0 and the collectives travel over a gloo process group.
def _shard_bounds(n, world, uneven):
    if not uneven:
        return [n * r // world for r in range(world + 1)]
    cuts = [0]
    for r in range(world):
        share = 0 if (uneven == "empty" and r == 1) else (r + 1) * 3 + 1
        cuts.append(cuts[-1] + share)
    total = cuts[-1]
    return [n * c // total for c in cuts]


def _gen_case(seed):
    rng = np.random.default_rng(seed)
    n, g = 700_001, 30_000
    k = (rng.integers(0, g, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    km = rng.random(n) < 0.002
    v0 = np.round(rng.normal(100, 10, n), 1)
    v1 = rng.integers(-1000, 1000, n).astype(np.int64)
    m1 = rng.random(n) < 0.1
    m1[: int(0.4 * n)] = False              # rank 0's shard has no null here: it passes NO mask while the other ranks do
    nr = 60_003
    rk = (rng.permutation(4 * nr)[:nr].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rg = rng.integers(0, 900, nr).astype(np.int64)
    rgm = rng.random(nr) < 0.01
    rgm[: int(0.4 * nr)] = False            # likewise: rank 0's build shard carries no group mask
    lk = np.where(rng.random(n) < 0.9, rk[rng.integers(0, nr, n)], rng.integers(1, 1 << 40, n))
    return dict(n=n, k=k, km=km, v0=v0, v1=v1, m1=m1, nr=nr, rk=rk, rg=rg, rgm=rgm, lk=lk)


def _transport_worker(rank, world, port, outdir, uneven):
    import datetime
    import pickle
    import torch
    import torch.distributed as dist
    import pandrs_amd as pa
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    torch.cuda.set_device(0)
    ctx = pa.Context(0)

    def all_gather(b):
        parts = [None] * world
        dist.all_gather_object(parts, b)
        return parts

    def all_reduce_max(vals):
        t = torch.tensor(vals, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    def all_to_all_v(parts):
        everyone = [None] * world
        dist.all_gather_object(everyone, parts)          # test-sized payloads: the whole matrix travels
        return [everyone[src][rank] for src in range(world)]

    ctx.comm_adopt_transport(all_gather, all_reduce_max, all_to_all_v, rank, world)
    c = _gen_case(77)
    b = _shard_bounds(c["n"], world, uneven)
    lo, hi = b[rank], b[rank + 1]
    dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    bits = lambda m, lo, hi: O.pack_mask(m[lo:hi])
    out = {}
    # groupby: rank 0 passes NO null mask for v1 although others do (layout agreement), device and host shards
    keys = [(dev(c["k"][lo:hi]), dev(bits(c["km"], lo, hi)), O.I64)]
    assert rank != 0 or not c["m1"][lo:hi].any()
    m1 = None if rank == 0 else dev(bits(c["m1"], lo, hi))
    vals = [(dev(c["v0"][lo:hi]), None, O.F64), (dev(c["v1"][lo:hi]), m1, O.I64)]
    aggs = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 4)]
    ctx.dist_groupby_compute(keys, hi - lo, vals, aggs)
    kc, kn, oa = ctx.groupby_fetch(to_device=False)
    out.update(gb_kc=kc, gb_kn=kn, gb_oa=oa)
    # steady state: the same call again allocates nothing on the device
    ctx.dist_groupby_compute(keys, hi - lo, vals, aggs)
    a0 = pa.Context.alloc_events()
    ctx.dist_groupby_compute(keys, hi - lo, vals, aggs)
    out["gb_allocs_in_steady_state"] = np.array([pa.Context.alloc_events() - a0])
    # host shards through the same exchange
    hk = [(c["k"][lo:hi], bits(c["km"], lo, hi), O.I64)]
    hv = [(c["v0"][lo:hi], None, O.F64), (c["v1"][lo:hi], bits(c["m1"], lo, hi), O.I64)]
    ctx.dist_groupby_compute(hk, hi - lo, hv, aggs)
    kc, kn, oa = ctx.groupby_fetch(to_device=False)
    out.update(gbh_kc=kc, gbh_kn=kn, gbh_oa=oa)
    # join: uneven build shards (padding rows with NULL keys in the all-gather), one possibly empty
    rb = _shard_bounds(c["nr"], world, uneven)
    rlo, rhi = rb[rank], rb[rank + 1]
    assert rank != 0 or not c["rgm"][rlo:rhi].any()
    rgm = dev(O.pack_mask(c["rgm"][rlo:rhi])) if rank != 0 else None
    for rep in range(3):
        if rep == 2:
            a0 = pa.Context.alloc_events()
        kc, kn, oa = ctx.dist_join_groupby_sum((dev(c["lk"][lo:hi]), None, O.I64), (dev(c["v0"][lo:hi]), None, O.F64), hi - lo,
                                               (dev(c["rk"][rlo:rhi]), None, O.I64), (dev(c["rg"][rlo:rhi]), rgm, O.I64), rhi - rlo)
    out["join_allocs_in_steady_state"] = np.array([pa.Context.alloc_events() - a0])
    out.update(jn_kc=kc.cpu().numpy().view(np.uint64), jn_kn=kn.cpu().numpy(), jn_oa=oa.cpu().numpy())
    # the GENERAL exchange inside the library (row shuffle by key owner): Median / Std / Nunique, and a composite key
    gen_aggs = [(0, O.MEDIAN), (1, O.MEDIAN), (0, O.STD), (1, O.SUM), (0, O.NUNIQUE)]
    ctx.dist_groupby_compute(keys, hi - lo, vals, gen_aggs)
    kc, kn, oa = ctx.groupby_fetch(to_device=False)
    out.update(gen_kc=kc, gen_kn=kn, gen_oa=oa)
    k2 = (c["v1"][lo:hi] % 7).astype(np.uint32)
    mk = [(dev(c["k"][lo:hi] % 1000), None, O.I64), (dev(k2), m1, O.U32CODE)]        # the code column's mask: absent on rank 0 only
    ctx.dist_groupby_compute(mk, hi - lo, vals[:1], [(0, O.SUM), (0, O.MEDIAN), (0, O.COUNT)])
    kc, kn, oa = ctx.groupby_fetch(to_device=False)
    out.update(mk_kc=kc, mk_kn=kn, mk_oa=oa)
    # a rank-local failure between collectives: rank `world - 1` passes a value column nothing can sum; EVERY rank must
    # come back with an error instead of waiting in the count exchange
    bad = [(dev(c["v1"][lo:hi].astype(np.uint32)), None, O.U32CODE)] if rank == world - 1 else [(dev(c["v0"][lo:hi]), None, O.F64)]
    try:
        ctx.dist_groupby_compute(keys, hi - lo, bad, [(0, 0)])
        out["failure_status"] = np.array([0])
    except pa.engine.PandrsHipError as e:
        out["failure_status"] = np.array([e.status])
    # ... and the communicator still works afterwards
    ctx.dist_groupby_compute(keys, hi - lo, vals[:1], [(0, 0)])
    out["after_failure_groups"] = np.array([ctx.groupby_fetch(to_device=False)[0].shape[1]])
    np.savez(os.path.join(outdir, "x%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.parametrize("world,uneven", [(2, "uneven"), (3, "empty"), (5, "uneven")])
def test_in_library_exchange_with_real_ranks_over_a_host_transport(tmp_path, world, uneven):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_transport_worker, args=(world, port, str(tmp_path), uneven), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, "x%d.npz" % r)) for r in range(world)]
    cat = lambda name: tuple(np.concatenate([p[name + s_] for p in parts], axis=1) for s_ in ("_kc", "_kn", "_oa"))
    c = _gen_case(77)
    keys = [(c["k"], O.pack_mask(c["km"]), O.I64)]
    vals = [(c["v0"], None, O.F64), (c["v1"], O.pack_mask(c["m1"]), O.I64)]
    aggs = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 4)]
    want = O.groupby_agg(keys, c["n"], vals, aggs)
    for name in ("gb", "gbh"):
        got = cat(name)
        assert got[0].shape[1] == want[0].shape[1], name            # owners are disjoint: no key on two ranks
        assert_groupby_equal(got, want, [O.I64], int_exact_rows=[2, 3, 4, 6])
    wantj = O.join_groupby_sum((c["lk"], None, O.I64), (c["v0"], None, O.F64), c["n"], (c["rk"], None, O.I64),
                               (c["rg"], O.pack_mask(c["rgm"]), O.I64), c["nr"])
    gotj = cat("jn")
    assert gotj[0].shape[1] == wantj[0].shape[1]
    assert_groupby_equal(gotj, wantj, [O.I64])
    gen_aggs = [(0, O.MEDIAN), (1, O.MEDIAN), (0, O.STD), (1, O.SUM), (0, O.NUNIQUE)]
    wantg = O.groupby_agg(keys, c["n"], vals, gen_aggs)
    gotg = cat("gen")
    assert gotg[0].shape[1] == wantg[0].shape[1]
    assert_groupby_equal(gotg, wantg, [O.I64], int_exact_rows=[0, 1, 3, 4])
    mk = [(c["k"] % 1000, None, O.I64), ((c["v1"] % 7).astype(np.uint32), O.pack_mask(c["m1"]), O.U32CODE)]
    wantm = O.groupby_agg(mk, c["n"], vals[:1], [(0, O.SUM), (0, O.MEDIAN), (0, O.COUNT)])
    gotm = cat("mk")
    assert gotm[0].shape[1] == wantm[0].shape[1]
    assert_groupby_equal(gotm, wantm, [O.I64, O.U32CODE], int_exact_rows=[1, 2])
    for p in parts:
        assert int(p["gb_allocs_in_steady_state"][0]) == 0 and int(p["join_allocs_in_steady_state"][0]) == 0
        assert int(p["failure_status"][0]) != 0                     # every rank, not only the failing one
    assert sum(int(p["after_failure_groups"][0]) for p in parts) == want[0].shape[1]
