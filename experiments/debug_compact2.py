import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandrs_amd as pa
from oracle import oracle as O
rng = np.random.default_rng(5)
n, g = 17_000_000, 2_000_000
hot = rng.random(n) < 0.85
ids = np.where(hot, rng.integers(0, 800, n), rng.integers(0, g, n))
k = (ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
k[::100_003] = -1
key = (k, O.pack_mask(rng.random(n) < 0.001), O.I64)
v = rng.standard_normal(n); v[::50_021] = np.nan
vals = [(v, O.pack_mask(rng.random(n) < 0.02), O.F64), (rng.standard_normal(n), O.pack_mask(rng.random(n) < 0.3), O.F64)]
aggs = [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.MEAN), (1, O.SUM), (1, O.MIN), (1, O.MAX), (1, O.COUNT)]
ctx = pa.Context(0)
for i in range(2):
    got = ctx.groupby_agg([key], n, vals, aggs)
    print(ctx.timings(), got[0].shape, flush=True)
