"""Shared helpers for parity tests (test-side only)."""
import numpy as np

from oracle import oracle as O


def codes_of(strings):
    """Equal string <=> equal code (reference: GLOBAL_STRING_POOL, src/column/string_pool.rs:28-53)."""
    pool, codes = {}, []
    for s in strings:
        codes.append(pool.setdefault(s, len(pool)))
    inv = [None] * len(pool)
    for s, c in pool.items():
        inv[c] = s
    return np.array(codes, np.uint32), inv


def sort_groups(kc, kn, aggs, key_dtypes):
    """Order-insensitive comparison: sort groups by (null, sortable key) per key column."""
    g = kc.shape[1]
    if g == 0:
        return kc, kn, aggs
    cols = []
    for k in reversed(range(kc.shape[0])):
        cell = kc[k].astype(np.uint64)
        dt = key_dtypes[k]
        if dt == O.I64:
            s = cell ^ np.uint64(0x8000000000000000)
        elif dt == O.F64:
            neg = (cell >> np.uint64(63)).astype(bool)
            s = np.where(neg, ~cell, cell | np.uint64(0x8000000000000000))
        else:
            s = cell
        cols.append(s)
        cols.append(kn[k])
    order = np.lexsort(cols)
    return kc[:, order], kn[:, order], aggs[:, order]


def assert_groupby_equal(got, want, key_dtypes, int_exact_rows=(), rtol=1e-9):
    """got/want = (kc, kn, aggs).  Keys and null flags bit-exact; aggregates within rtol relative
    (BASELINE: 1e-9 for f64 sums/means), rows listed in int_exact_rows compared bit-exact."""
    gk, gn, ga = sort_groups(*got, key_dtypes)
    wk, wn, wa = sort_groups(*want, key_dtypes)
    assert gk.shape == wk.shape, (gk.shape, wk.shape)
    np.testing.assert_array_equal(gn, wn)
    np.testing.assert_array_equal(gk, wk)
    assert ga.shape == wa.shape
    for a in range(ga.shape[0]):
        if a in int_exact_rows:
            np.testing.assert_array_equal(ga[a].view(np.uint64), wa[a].view(np.uint64))
        else:
            nan_g, nan_w = np.isnan(ga[a]), np.isnan(wa[a])
            np.testing.assert_array_equal(nan_g, nan_w)
            np.testing.assert_allclose(ga[a][~nan_g], wa[a][~nan_w], rtol=rtol, atol=0)
