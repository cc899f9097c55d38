import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandrs_amd as pa
from oracle import oracle as O
ctx = pa.Context(0); ctx.set_option("no_small", 1)
rng = np.random.default_rng(57)
n = 2_400_000
ids = np.repeat(rng.integers(0, 900_000, (n + 15) // 16), 16)[:n]
k = (ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
vals = [(rng.normal(size=n), None, O.F64)]
for aggs in ([(0, O.SUM), (0, O.MAX), (0, O.COUNT)], [(0, O.SUM), (0, O.MIN), (0, O.MAX), (0, O.COUNT)], [(0, O.SUM)]):
    ctx.groupby_compute([(k, None, O.I64)], n, vals, aggs)
    print(aggs, ctx.timings())
