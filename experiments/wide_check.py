#!/usr/bin/env python3
"""The scatter's tile variants against each other on one input: same groups, same aggregates (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandrs_amd as pa
from tests.helpers import assert_groupby_equal
rng = np.random.default_rng(3)
n = 16384 * 61 + 777
k = (rng.integers(0, 80_000, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
vals = [(rng.normal(100, 10, n), None, pa.F64), (rng.integers(-99, 99, n).astype(np.int64), np.packbits(rng.random(n) < 0.2, bitorder="little"), pa.I64)]
aggs = [(0, pa.SUM), (0, pa.MIN), (1, pa.SUM), (1, pa.MAX), (1, pa.COUNT)]
ctx = pa.Context(0)
for name in ("no_direct", "no_absorb", "no_small"): ctx.set_option(name, 1)
res = {}
for w in (-1, 1):
    for ex in (0, 1):
        ctx.set_option("scatter_wide", w); ctx.set_option("exact_partition", ex)
        res[(w, ex)] = ctx.groupby_agg([(k, None, pa.I64)], n, vals, aggs)
base = res[(-1, 0)]
for key, r in res.items():
    assert_groupby_equal(r, base, [pa.I64], int_exact_rows=[1, 2, 3, 4])
    print("variant", key, "ok", r[0].shape)
