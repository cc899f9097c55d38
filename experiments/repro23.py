import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandrs_amd as pa
from oracle import oracle as O
from tests.helpers import assert_groupby_equal
ctx = pa.Context(0)
rng = np.random.default_rng(5)
n = 5_000_000
k1 = rng.integers(-100, 100, n).astype(np.int64); k2 = rng.integers(0, 5, n).astype(np.uint32)
v = rng.normal(50, 20, n); m = O.pack_mask(rng.random(n) < 0.1)
for nk in (1, 2):
    keys = [(k1, None, O.I64), (k2, None, O.U32CODE)][:nk]
    kd = [O.I64, O.U32CODE][:nk]
    want = O.groupby_agg(keys, n, [(v, m, O.F64)], [(0, O.MAX), (0, O.MEAN)])
    for opts in [dict(no_direct=0, slice_rows=0), dict(no_direct=0, slice_rows=20000), dict(no_direct=1, slice_rows=20000), dict(no_direct=1, slice_rows=0), dict(no_direct=0, slice_rows=20000, generic_aggregate=1)]:
        for k_, v_ in dict(no_direct=0, slice_rows=0, generic_aggregate=0).items(): ctx.set_option(k_, v_)
        for k_, v_ in opts.items(): ctx.set_option(k_, v_)
        bad = 0
        for rep in range(3):
            got = ctx.groupby_agg(keys, n, [(v, m, O.F64)], [(0, O.MAX), (0, O.MEAN)])
            try:
                assert_groupby_equal(got, want, kd, int_exact_rows=[0])
            except AssertionError as e:
                bad += 1
        print("nk", nk, opts, "bad runs:", bad, "of 3", flush=True)
