import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandrs_amd as pa
from oracle import oracle as O
ctx = pa.Context(0)
rng = np.random.default_rng(1)
for nb, npb, missp in ((1000, 3_000_000, 0.0), (30_000, 3_000_000, 0.0), (400_000, 3_000_000, 0.0), (400_000, 3_000_000, 0.1), (400_000, 100_000, 0.0)):
    rkeys = (rng.permutation(nb * 3)[:nb].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    rg = rng.integers(0, 5000, nb).astype(np.int64)
    pick = rng.integers(0, nb, npb)
    lkeys = rkeys[pick].copy()
    miss = rng.random(npb) < missp
    lkeys[miss] = -7
    lv = np.ones(npb, np.int64)
    args = ((lkeys, None, O.I64), (lv, None, O.I64), npb, (rkeys, None, O.I64), (rg, None, O.I64), nb)
    ctx.set_option("join_no_l2", -1)
    k, n, a = ctx.join_groupby_sum(*args)
    t = ctx.timings()
    print("nb", nb, "npb", npb, "miss", missp, "total", a[0].sum(), "expected", (~miss).sum(), "parts", t["n_partitions"], "phases", list(t["phase_ms"]), flush=True)
