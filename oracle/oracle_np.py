"""Pure-Python/numpy twin of the oracle — a second, independent restatement used to cross-check
oracle/pandrs_oracle.c on small inputs and to generate tests/golden fixtures.

TEST INFRASTRUCTURE ONLY (same rule as oracle.py).

It follows the reference literally: keys are formed as strings per row
(src/optimized/split_dataframe/group/grouping.rs:62-104), rows are collected per group in a dict
(HashMap<Vec<String>, Vec<usize>>), and each aggregate is a sequential fold over the group's row
list (src/optimized/split_dataframe/group/aggregation.rs:500-754).  Python loops: small cases only.
"""
import math
import struct

import numpy as np

I64, F64, U32CODE, BOOLBITS = 0, 1, 2, 3
SUM, MEAN, MIN, MAX, COUNT, STD, VAR, MEDIAN, FIRST, LAST, CUSTOM = range(11)
INNER, LEFT, RIGHT, OUTER = range(4)


def _null(mask, i):
    return mask is not None and (int(mask[i >> 3]) >> (i & 7)) & 1 == 1


def _get(col, i):
    data, mask, dt = col
    if _null(mask, i):
        return None
    if dt == BOOLBITS:
        return bool((int(data[i >> 3]) >> (i & 7)) & 1)
    if dt == F64:
        return float(data[i])
    return int(data[i])


def _key_string(col, i, pool=None):
    v = _get(col, i)
    if v is None:
        return "NULL"                                    # grouping.rs:74
    dt = col[2]
    if dt == F64:
        if math.isnan(v):
            return "NaN"
        return repr(v)                                   # injective, like Rust's shortest repr
    if dt == BOOLBITS:
        return "true" if v else "false"
    if dt == U32CODE and pool is not None:
        return pool[v]
    return str(v)


def _cell(col, i):
    v = _get(col, i)
    dt = col[2]
    if v is None:
        return 0
    if dt == F64:
        if math.isnan(v):
            return 0x7FF8000000000000
        return struct.unpack("<Q", struct.pack("<d", v))[0]
    if dt == I64:
        return v & 0xFFFFFFFFFFFFFFFF
    return int(v)


def _wrap_i64(x):
    x &= 0xFFFFFFFFFFFFFFFF
    return x - (1 << 64) if x >= (1 << 63) else x


def _variance(vals):
    if not vals:
        return 0.0
    n = float(len(vals))
    s = 0.0
    for v in vals:
        s += v
    mean = s / n
    ss = 0.0
    for v in vals:
        d = v - mean
        ss += d * d
    return ss / (n - 1.0) if len(vals) > 1 else 0.0


def _rmin(a, b):
    if math.isnan(a):
        return b
    if math.isnan(b):
        return a
    if a == b:
        return a if math.copysign(1.0, a) < 0 else b
    return a if a < b else b


def _rmax(a, b):
    if math.isnan(a):
        return b
    if math.isnan(b):
        return a
    if a == b:
        return b if math.copysign(1.0, a) < 0 else a
    return a if a > b else b


class OperationFailed(Exception):
    pass


def fold(col, op, rows):
    """GroupBy::calculate_aggregation (aggregation.rs:500-754)."""
    if op == COUNT:
        return float(len(rows))
    if op == CUSTOM:
        raise OperationFailed()
    dt = col[2]
    vals = [v for v in (_get(col, r) for r in rows) if v is not None]
    if dt == I64:
        if op == SUM:
            return float(_wrap_i64(sum(vals)))
        if op == MEAN:
            return float(_wrap_i64(sum(vals))) / float(len(vals)) if vals else 0.0
        if op == MIN:
            m = min(vals + [2**63 - 1])
            return 0.0 if m == 2**63 - 1 else float(m)
        if op == MAX:
            m = max(vals + [-2**63])
            return 0.0 if m == -2**63 else float(m)
        if op in (STD, VAR):
            var = _variance([float(v) for v in vals])
            return math.sqrt(var) if op == STD else var
        if op == MEDIAN:
            if not vals:
                return 0.0
            s = sorted(vals)
            mid = len(s) // 2
            return float(_wrap_i64(s[mid - 1] + s[mid])) / 2.0 if len(s) % 2 == 0 else float(s[mid])
        if op == FIRST:
            v = _get(col, rows[0]) if rows else None
            return float(v) if v is not None else 0.0
        if op == LAST:
            v = _get(col, rows[-1]) if rows else None
            return float(v) if v is not None else 0.0
    elif dt == F64:
        if op == SUM:
            s = 0.0
            for v in vals:
                s += v
            return s
        if op == MEAN:
            s = 0.0
            for v in vals:
                s += v
            return s / float(len(vals)) if vals else 0.0
        if op == MIN:
            m = math.inf
            for v in vals:
                m = _rmin(m, v)
            return 0.0 if m == math.inf else m
        if op == MAX:
            m = -math.inf
            for v in vals:
                m = _rmax(m, v)
            return 0.0 if m == -math.inf else m
        if op in (STD, VAR):
            var = _variance(vals)
            return math.sqrt(var) if op == STD else var
        if op == MEDIAN:
            if not vals:
                return 0.0
            s = sorted(vals)
            mid = len(s) // 2
            return (s[mid - 1] + s[mid]) / 2.0 if len(s) % 2 == 0 else s[mid]
        if op == FIRST:
            v = _get(col, rows[0]) if rows else None
            return v if v is not None else 0.0
        if op == LAST:
            v = _get(col, rows[-1]) if rows else None
            return v if v is not None else 0.0
    raise OperationFailed()


def groupby_agg(keys, n_rows, vals, aggs, pools=None):
    """-> dict: tuple(key strings) -> (first_row, [agg values...]); insertion (first-seen) order."""
    groups = {}
    for r in range(n_rows):
        k = tuple(_key_string(c, r, pools[i] if pools else None) for i, c in enumerate(keys))
        groups.setdefault(k, []).append(r)
    out = {}
    for k, rows in groups.items():
        out[k] = (rows[0], [fold(vals[c], op, rows) for c, op in aggs])
    return out


def group_indices(keys, n_rows, pools=None):
    """group_by's result (grouping.rs:62-104): tuple(key strings) -> its row indices, ascending
    (rows are visited in order and pushed, :98-103); a null key is the string "NULL" (:74)."""
    groups = {}
    for r in range(n_rows):
        k = tuple(_key_string(c, r, pools[i] if pools else None) for i, c in enumerate(keys))
        groups.setdefault(k, []).append(r)
    return groups


def key_cells(col, n_rows):
    """Vectorised (null flag, 8-byte cell) of every row of one key column: equal cells <=> equal key
    strings (NaNs collapsed, 0.0 != -0.0, grouping.rs:72-79); null rows get cell 0."""
    data, mask, dt = col
    nul = np.zeros(n_rows, np.uint8) if mask is None else np.unpackbits(np.asarray(mask, np.uint8), bitorder="little")[:n_rows]
    if dt == I64:
        cell = np.asarray(data, np.int64)[:n_rows].view(np.uint64).copy()
    elif dt == F64:
        d = np.asarray(data, np.float64)[:n_rows]
        cell = d.view(np.uint64).copy()
        cell[np.isnan(d)] = np.uint64(0x7FF8000000000000)
    elif dt == U32CODE:
        cell = np.asarray(data, np.uint32)[:n_rows].astype(np.uint64)
    else:
        cell = np.unpackbits(np.asarray(data, np.uint8), bitorder="little")[:n_rows].astype(np.uint64)
    cell[nul.astype(bool)] = 0
    return nul, cell


def groupby_agg_arrays(keys, n_rows, vals, aggs):
    """Same result as oracle.groupby_agg(..., faithful=True) layout, first-seen order."""
    res = groupby_agg(keys, n_rows, vals, aggs)
    g = len(res)
    kc = np.zeros((len(keys), g), np.uint64)
    kn = np.zeros((len(keys), g), np.uint8)
    oa = np.zeros((len(aggs), g), np.float64)
    for gi, (_, (first, av)) in enumerate(res.items()):
        for ki, c in enumerate(keys):
            kn[ki, gi] = 1 if _null(c[1], first) else 0
            kc[ki, gi] = _cell(c, first)
        oa[:, gi] = av
    return kc, kn, oa


def join_indices(lkey, n_left, rkey, n_right, how):
    """join_impl up to join_indices (join.rs:106-224)."""
    right = {}
    for i in range(n_right):
        if not _null(rkey[1], i):
            right.setdefault(_key_string(rkey, i), []).append(i)
    li, ri = [], []
    for i in range(n_left):
        if _null(lkey[1], i):
            continue
        m = right.get(_key_string(lkey, i))
        if m is not None:
            for r in m:
                li.append(i)
                ri.append(r)
        elif how in (LEFT, OUTER):
            li.append(i)
            ri.append(-1)
    if how in (RIGHT, OUTER):
        matched = set(r for r in ri if r >= 0)
        for i in range(n_right):
            if i not in matched:
                li.append(-1)
                ri.append(i)
    return np.array(li, np.int64), np.array(ri, np.int64)
