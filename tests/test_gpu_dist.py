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
