import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandrs_amd as pa
from oracle import oracle as O
rng = np.random.default_rng(5)
n, g = 17_000_000, 2_000_000
hot = rng.random(n) < 0.85
ids = np.where(hot, rng.integers(0, 800, n), rng.integers(0, g, n))
k = (ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
v = rng.standard_normal(n)
ctx = pa.Context(0)
try:
    ctx.groupby_compute([(k, None, pa.I64)], n, [(v, None, pa.F64)], [(0, pa.SUM), (0, pa.MIN), (0, pa.MAX)])
    print(ctx.timings())
except Exception as e:
    print("ERR", e)
