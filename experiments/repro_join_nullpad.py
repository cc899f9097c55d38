"""fused join -> groupby-sum with a block of NULL build keys in the middle (what the multi-rank all-gather's padding looks like)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import pandrs_amd as pa
from oracle import oracle as O
from tests.helpers import assert_groupby_equal
rng = np.random.default_rng(77)
n, nr = 700_001, 60_003
rk = (rng.permutation(4 * nr)[:nr].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
rg = rng.integers(0, 900, nr).astype(np.int64)
rgm = rng.random(nr) < 0.01
lk = np.where(rng.random(n) < 0.9, rk[rng.integers(0, nr, n)], rng.integers(1, 1 << 40, n))
v0 = np.round(rng.normal(100, 10, n), 1)
ctx = pa.Context(0)
dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
for pad_at, pad in ((None, 0), (21819, 16549), (0, 5000), (nr, 7)):
    if pad_at is None:
        k2, g2, km, gm = rk, rg, None, rgm
    else:
        k2 = np.concatenate([rk[:pad_at], np.zeros(pad, np.int64), rk[pad_at:]])
        g2 = np.concatenate([rg[:pad_at], np.zeros(pad, np.int64), rg[pad_at:]])
        km = np.concatenate([np.zeros(pad_at, bool), np.ones(pad, bool), np.zeros(nr - pad_at, bool)])
        gm = np.concatenate([rgm[:pad_at], np.zeros(pad, bool), rgm[pad_at:]])
    want = O.join_groupby_sum((lk, None, O.I64), (v0, None, O.F64), n, (k2, None if km is None else O.pack_mask(km), O.I64), (g2, O.pack_mask(gm), O.I64), len(k2))
    got = ctx.join_groupby_sum((dev(lk), None, O.I64), (dev(v0), None, O.F64), n, (dev(k2), None if km is None else dev(O.pack_mask(km)), O.I64), (dev(g2), dev(O.pack_mask(gm)), O.I64), len(k2))
    got = (got[0].cpu().numpy().view(np.uint64), got[1].cpu().numpy(), got[2].cpu().numpy())
    try:
        assert_groupby_equal(got, want, [O.I64])
        print("pad", pad_at, pad, "ok", ctx.timings()["n_partitions"])
    except AssertionError as e:
        print("pad", pad_at, pad, "MISMATCH", str(e)[:300])
