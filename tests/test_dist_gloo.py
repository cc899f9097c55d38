"""world_size-2 gloo rehearsal of the multi-GPU groupby exchange (pandrs_amd/dist.py): partials ->
owner split -> count exchange + ONE all_to_all -> merge.  The local engine is a numpy stand-in
(tests/cpu_engine.py) because there is no GPU here; the GPU box runs the same driver over
pandrs_amd.Context in tests/test_gpu_groupby.py::test_partials_split_merge_roundtrip and bench.py."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(n=60_000, g=5_000):
    rng = np.random.default_rng(77)
    ids = rng.integers(0, g, n).astype(np.uint64)
    keys = (ids * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    km = np.packbits(rng.random(n) < 0.002, bitorder="little")
    v0 = rng.normal(100, 10, n)
    v1 = rng.normal(-5, 3, n)
    m1 = np.packbits(rng.random(n) < 0.1, bitorder="little")
    return keys, km, v0, v1, m1


AGGS = [(0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (1, 0), (1, 1), (1, 2), (1, 3)]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from pandrs_amd.dist import DistributedGroupBy
    from tests.cpu_engine import NumpyEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys, km, v0, v1, m1 = _data()
    n = len(keys)
    lo, hi = (n // world // 8 * 8) * rank, n if rank == world - 1 else (n // world // 8 * 8) * (rank + 1)
    bits = lambda m: np.packbits(np.unpackbits(m, bitorder="little")[:n][lo:hi], bitorder="little")
    d = DistributedGroupBy(NumpyEngine(), dist, "cpu")
    kc, kn, oa = d.groupby_agg([(keys[lo:hi], bits(km), 0)], hi - lo,
                               [(v0[lo:hi], None, 1), (v1[lo:hi], bits(m1), 1)], AGGS)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), kc=kc, kn=kn, oa=oa)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_matches_oracle(world):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from tests.helpers import assert_groupby_equal
    port = _free_port()
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, port, outdir), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, "r%d.npz" % r)) for r in range(world)]
    # key ownership is disjoint across ranks
    allk = np.concatenate([np.stack([p["kn"][0].astype(np.uint64), p["kc"][0]], 1) for p in parts])
    assert len(np.unique(allk, axis=0)) == len(allk)
    got = tuple(np.concatenate([p[name] for p in parts], axis=1) for name in ("kc", "kn", "oa"))
    keys, km, v0, v1, m1 = _data()
    want = O.groupby_agg([(keys, km, O.I64)], len(keys), [(v0, None, O.F64), (v1, m1, O.F64)], AGGS)
    exact = [i for i, (_, op) in enumerate(AGGS) if op in (O.MIN, O.MAX, O.COUNT)]
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=exact)


# ---- config 5 across ranks: all-gather the build side, local fused join->groupby, exchange ----------
def _join_data(nl=50_003, nr=6_001, g=300):
    rng = np.random.default_rng(91)
    rkeys = (rng.permutation(4 * nr)[:nr].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rgrp = rng.integers(0, g, nr).astype(np.int64)
    rk_null = rng.random(nr) < 0.01
    rg_null = rng.random(nr) < 0.01
    pick = rng.integers(0, nr, nl)
    lkeys = np.where(rng.random(nl) < 0.9, rkeys[pick], rng.integers(1, 1 << 40, nl))
    lk_null = rng.random(nl) < 0.005
    lval = rng.normal(10, 3, nl)
    return lkeys, lk_null, lval, rkeys, rk_null, rgrp, rg_null


def _shard(n, rank, world):
    # deliberately not byte-aligned: exercises the null-key padding of the build shards
    cut = [n * r // world for r in range(world + 1)]
    return cut[rank], cut[rank + 1]


def _join_worker(rank, world, port, outdir, strategy="allgather"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from pandrs_amd.dist import DistributedJoinGroupBy
    from tests.cpu_engine import NumpyEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lkeys, lk_null, lval, rkeys, rk_null, rgrp, rg_null = _join_data()
    pb = lambda m: np.packbits(m, bitorder="little")
    l0, l1 = _shard(len(lkeys), rank, world)
    r0, r1 = _shard(len(rkeys), rank, world)
    d = DistributedJoinGroupBy(NumpyEngine(), dist, "cpu")
    kc, kn, oa = d.join_groupby_sum((lkeys[l0:l1], pb(lk_null[l0:l1]), 0), (lval[l0:l1], None, 1), l1 - l0,
                                    (rkeys[r0:r1], pb(rk_null[r0:r1]), 0), (rgrp[r0:r1], pb(rg_null[r0:r1]), 0),
                                    r1 - r0, strategy=strategy)
    np.savez(os.path.join(outdir, "j%d.npz" % rank), kc=kc, kn=kn, oa=oa)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,strategy", [(2, "allgather"), (3, "allgather"), (2, "shuffle"), (3, "shuffle"), (2, "auto")])
def test_distributed_join_groupby_matches_oracle(world, strategy):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from tests.helpers import assert_groupby_equal
    port = _free_port()
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_join_worker, args=(world, port, outdir, strategy), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, "j%d.npz" % r)) for r in range(world)]
    got = tuple(np.concatenate([p[name] for p in parts], axis=1) for name in ("kc", "kn", "oa"))
    lkeys, lk_null, lval, rkeys, rk_null, rgrp, rg_null = _join_data()
    pb = lambda m: np.packbits(m, bitorder="little")
    want = O.join_groupby_sum((lkeys, pb(lk_null), O.I64), (lval, None, O.F64), len(lkeys),
                              (rkeys, pb(rk_null), O.I64), (rgrp, pb(rg_null), O.I64), len(rkeys))
    assert got[0].shape[1] == want[0].shape[1]
    assert_groupby_equal(got, want, [O.I64])


# ---- non-mergeable aggregates (Std / Var / Median) across ranks: the row shuffle by key owner -----------
AGGS_ANY = [(0, 7), (0, 5), (1, 6), (1, 7), (0, 0), (1, 4)]      # median, std, var, median, sum, count


def _shuffle_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from pandrs_amd.dist import DistributedGroupBy
    from tests.cpu_engine import NumpyEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys, km, v0, v1, m1 = _data(n=40_000, g=900)
    v1 = np.round(v1 * 100).astype(np.int64)
    n = len(keys)
    lo, hi = (n // world // 8 * 8) * rank, n if rank == world - 1 else (n // world // 8 * 8) * (rank + 1)
    bits = lambda m: np.packbits(np.unpackbits(m, bitorder="little")[:n][lo:hi], bitorder="little")
    d = DistributedGroupBy(NumpyEngine(), dist, "cpu")
    kc, kn, oa = d.groupby_agg([(keys[lo:hi], bits(km), 0)], hi - lo,
                               [(v0[lo:hi], None, 1), (v1[lo:hi], bits(m1), 0)], AGGS_ANY)
    np.savez(os.path.join(outdir, "s%d.npz" % rank), kc=kc, kn=kn, oa=oa)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shuffle_path_for_non_mergeable_aggregates(world):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from tests.helpers import assert_groupby_equal
    port = _free_port()
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_shuffle_worker, args=(world, port, outdir), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, "s%d.npz" % r)) for r in range(world)]
    got = tuple(np.concatenate([p[name] for p in parts], axis=1) for name in ("kc", "kn", "oa"))
    keys, km, v0, v1, m1 = _data(n=40_000, g=900)
    v1 = np.round(v1 * 100).astype(np.int64)
    want = O.groupby_agg([(keys, km, O.I64)], len(keys), [(v0, None, O.F64), (v1, m1, O.I64)], AGGS_ANY)
    assert got[0].shape[1] == want[0].shape[1]
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=[0, 3, 5])


# ---- multi-key groupby across ranks: shuffle on a hash cell of the whole tuple ---------------------------
def _mk_data(n=30_000):
    rng = np.random.default_rng(5)
    k0 = (rng.integers(0, 40, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    m0 = np.packbits(rng.random(n) < 0.02, bitorder="little")
    k1 = rng.integers(0, 7, n).astype(np.uint32)
    k2 = rng.choice(np.array([0.5, -0.0, 0.0, np.nan, 3.25]), n)
    v = rng.normal(0, 1, n)
    return k0, m0, k1, k2, v


def _mk_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from pandrs_amd.dist import DistributedGroupBy
    from tests.cpu_engine import NumpyEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k0, m0, k1, k2, v = _mk_data()
    n = len(k0)
    lo, hi = (n // world // 8 * 8) * rank, n if rank == world - 1 else (n // world // 8 * 8) * (rank + 1)
    bits = lambda m: np.packbits(np.unpackbits(m, bitorder="little")[:n][lo:hi], bitorder="little")
    d = DistributedGroupBy(NumpyEngine(), dist, "cpu")
    kc, kn, oa = d.groupby_agg([(k0[lo:hi], bits(m0), 0), (k1[lo:hi], None, 2), (k2[lo:hi], None, 1)], hi - lo,
                               [(v[lo:hi], None, 1)], [(0, 0), (0, 4), (0, 7)])
    np.savez(os.path.join(outdir, "m%d.npz" % rank), kc=kc, kn=kn, oa=oa)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_multi_key_groupby_across_ranks(world):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from tests.helpers import assert_groupby_equal
    port = _free_port()
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_mk_worker, args=(world, port, outdir), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, "m%d.npz" % r)) for r in range(world)]
    got = tuple(np.concatenate([p[name] for p in parts], axis=1) for name in ("kc", "kn", "oa"))
    k0, m0, k1, k2, v = _mk_data()
    want = O.groupby_agg([(k0, m0, O.I64), (k1, None, O.U32CODE), (k2, None, O.F64)], len(k0), [(v, None, O.F64)],
                         [(0, O.SUM), (0, O.COUNT), (0, O.MEDIAN)])
    assert got[0].shape[1] == want[0].shape[1]
    assert_groupby_equal(got, want, [O.I64, O.U32CODE, O.F64], int_exact_rows=[1, 2])


# ---- null masks on ONE rank only: every rank must still plan the same partial-record layout -----------
def _one_sided_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from pandrs_amd.dist import DistributedGroupBy
    from tests.cpu_engine import NumpyEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys, km, v0, v1, m1 = _data()
    n = len(keys)
    lo, hi = (n // world // 8 * 8) * rank, n if rank == world - 1 else (n // world // 8 * 8) * (rank + 1)
    bits = lambda m: np.packbits(np.unpackbits(m, bitorder="little")[:n][lo:hi], bitorder="little")
    d = DistributedGroupBy(NumpyEngine(), dist, "cpu")
    # rank 0 passes masks, the others pass None (what io.py does for a shard with null_count == 0)
    vmask = bits(m1) if rank == 0 else None
    kmask = bits(km) if rank == 0 else None
    kc, kn, oa = d.groupby_agg([(keys[lo:hi], kmask, 0)], hi - lo, [(v0[lo:hi], None, 1), (v1[lo:hi], vmask, 1)], AGGS)
    np.savez(os.path.join(outdir, "o%d.npz" % rank), kc=kc, kn=kn, oa=oa)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_null_masks_on_one_rank_only(world):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    from tests.helpers import assert_groupby_equal
    port = _free_port()
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_one_sided_worker, args=(world, port, outdir), nprocs=world, join=True)
        parts = [np.load(os.path.join(outdir, "o%d.npz" % r)) for r in range(world)]
    got = tuple(np.concatenate([p[name] for p in parts], axis=1) for name in ("kc", "kn", "oa"))
    keys, km, v0, v1, m1 = _data()
    n = len(keys)
    cut = n // world // 8 * 8
    # the masks only cover rank 0's rows
    keep = np.arange(n) < cut
    km2 = np.packbits(np.unpackbits(km, bitorder="little")[:n] & keep, bitorder="little")
    m12 = np.packbits(np.unpackbits(m1, bitorder="little")[:n] & keep, bitorder="little")
    want = O.groupby_agg([(keys, km2, O.I64)], n, [(v0, None, O.F64), (v1, m12, O.F64)], AGGS)
    exact = [i for i, (_, op) in enumerate(AGGS) if op in (O.MIN, O.MAX, O.COUNT)]
    assert_groupby_equal(got, want, [O.I64], int_exact_rows=exact)
