"""ctypes front-end of the CPU oracle (oracle/pandrs_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under pandrs_amd/ may import this module.

Columns are passed as ``(data, null_mask, dtype)`` triples:
  data       np.int64 / np.float64 / np.uint32 array, or np.uint8 bit-packed array for bools
  null_mask  None or np.uint8 LSB-first bitmap, bit 1 = null (reference: src/core/column.rs:163-177)
  dtype      I64 / F64 / U32CODE / BOOLBITS
"""
import ctypes as C
import os
import subprocess

import numpy as np

I64, F64, U32CODE, BOOLBITS, CELL64 = 0, 1, 2, 3, 4
SUM, MEAN, MIN, MAX, COUNT, STD, VAR, MEDIAN, FIRST, LAST, CUSTOM, NUNIQUE = range(12)
INNER, LEFT, RIGHT, OUTER = range(4)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpandrs_oracle.so")


class _Col(C.Structure):
    _fields_ = [("data", C.c_void_p), ("null_mask", C.c_void_p),
                ("dtype", C.c_int32), ("reserved", C.c_int32)]


class _Agg(C.Structure):
    _fields_ = [("col", C.c_int32), ("op", C.c_int32)]


def build():
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "pandrs_oracle.c")
    if (not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_free.argtypes = [C.c_void_p]
        _lib.oracle_free.restype = None
    return _lib


_NP_OF = {I64: np.int64, F64: np.float64, U32CODE: np.uint32, BOOLBITS: np.uint8, CELL64: np.uint64}


def _cols(cols, keep):
    arr = (_Col * max(len(cols), 1))()
    for i, (data, mask, dt) in enumerate(cols):
        data = np.ascontiguousarray(data, dtype=_NP_OF[dt])
        keep.append(data)
        arr[i].data = data.ctypes.data
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            keep.append(mask)
            arr[i].null_mask = mask.ctypes.data
        else:
            arr[i].null_mask = None
        arr[i].dtype = dt
    return arr


def _aggs(aggs):
    arr = (_Agg * max(len(aggs), 1))()
    for i, (c, op) in enumerate(aggs):
        arr[i].col, arr[i].op = int(c), int(op)
    return arr


class OracleError(RuntimeError):
    def __init__(self, status):
        super().__init__("oracle status %d" % status)
        self.status = status


def _take(ptr, shape, np_dtype):
    n = int(np.prod(shape))
    if not ptr or n == 0:
        out = np.zeros(shape, dtype=np_dtype)
    else:
        ct = np.ctypeslib.as_ctypes_type(np_dtype)
        out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy().reshape(shape)
    if ptr:
        lib().oracle_free(ptr)
    return out


def groupby_agg(keys, n_rows, vals, aggs, faithful=False, pools=None, threads=1):
    """-> (key_cells[n_keys, G] u64, key_null[n_keys, G] u8, aggs[n_aggs, G] f64).

    faithful=False: typed restatement, output sorted by key (nulls last).
    faithful=True : string-keyed HashMap shape of the reference (lazy.rs:186-404); output in
                    table order.  `pools[k]` = list of bytes/str for U32CODE key column k.
                    threads > 1: the reference's parallel shape (par_groupby's chunk-map + serial merge,
                    grouping.rs:203-280, then par_aggregate's parallel folds).
    """
    keep = []
    kc, vc, ag = _cols(keys, keep), _cols(vals, keep), _aggs(aggs)
    ng = C.c_int64(0)
    pk, pn, pa = C.c_void_p(), C.c_void_p(), C.c_void_p()
    L = lib()
    if faithful:
        pool_arr = None
        if pools is not None:
            arrs = []
            for p in pools:
                if p is None:
                    arrs.append(None)
                    continue
                enc = [s.encode() if isinstance(s, str) else s for s in p]
                keep.append(enc)
                a = (C.c_char_p * len(enc))(*enc)
                keep.append(a)
                arrs.append(C.cast(a, C.c_void_p))
            pool_arr = (C.c_void_p * len(arrs))(*[a if a is not None else None for a in arrs])
        if threads > 1:
            rc = L.oracle_groupby_agg_ref_mt(kc, C.c_int(len(keys)), pool_arr, C.c_int64(n_rows),
                                             vc, C.c_int(len(vals)), ag, C.c_int(len(aggs)), C.c_int(int(threads)),
                                             C.byref(ng), C.byref(pk), C.byref(pn), C.byref(pa))
        else:
            rc = L.oracle_groupby_agg_ref(kc, C.c_int(len(keys)), pool_arr, C.c_int64(n_rows),
                                          vc, C.c_int(len(vals)), ag, C.c_int(len(aggs)),
                                          C.byref(ng), C.byref(pk), C.byref(pn), C.byref(pa))
    else:
        rc = L.oracle_groupby_agg(kc, C.c_int(len(keys)), C.c_int64(n_rows),
                                  vc, C.c_int(len(vals)), ag, C.c_int(len(aggs)),
                                  C.byref(ng), C.byref(pk), C.byref(pn), C.byref(pa))
    if rc:
        raise OracleError(rc)
    g = ng.value
    return (_take(pk.value, (len(keys), g), np.uint64),
            _take(pn.value, (len(keys), g), np.uint8),
            _take(pa.value, (len(aggs), g), np.float64))


def join_indices(lkey, n_left, rkey, n_right, how, faithful=False):
    """faithful=True: the reference's own shape (HashMap<String, Vec<usize>> over the right side, one formatted
    String + SipHash per probed left row, join.rs:107-224); same pairs, same order."""
    keep = []
    lc, rc_ = _cols([lkey], keep), _cols([rkey], keep)
    n = C.c_int64(0)
    pl, pr = C.c_void_p(), C.c_void_p()
    fn = lib().oracle_join_indices_ref if faithful else lib().oracle_join_indices
    rc = fn(lc, C.c_int64(n_left), rc_, C.c_int64(n_right), C.c_int(how),
                                   C.byref(n), C.byref(pl), C.byref(pr))
    if rc:
        raise OracleError(rc)
    return _take(pl.value, (n.value,), np.int64), _take(pr.value, (n.value,), np.int64)


def gather(src, mask, idx, fill, dtype):
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    n = len(idx)
    src = np.ascontiguousarray(src, dtype=_NP_OF[dtype])
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    mp = None if m is None else C.c_void_p(m.ctypes.data)
    L = lib()
    if dtype == I64:
        out = np.empty(n, np.int64)
        L.oracle_gather_i64(C.c_void_p(src.ctypes.data), mp, C.c_void_p(idx.ctypes.data),
                            C.c_int64(n), C.c_int64(int(fill)), C.c_void_p(out.ctypes.data))
    elif dtype == F64:
        out = np.empty(n, np.float64)
        L.oracle_gather_f64(C.c_void_p(src.ctypes.data), mp, C.c_void_p(idx.ctypes.data),
                            C.c_int64(n), C.c_double(float(fill)), C.c_void_p(out.ctypes.data))
    elif dtype == U32CODE:
        out = np.empty(n, np.uint32)
        L.oracle_gather_u32(C.c_void_p(src.ctypes.data), mp, C.c_void_p(idx.ctypes.data),
                            C.c_int64(n), C.c_uint32(int(fill)), C.c_void_p(out.ctypes.data))
    else:
        out = np.empty(n, np.uint8)
        L.oracle_gather_bool(C.c_void_p(src.ctypes.data), mp, C.c_void_p(idx.ctypes.data),
                             C.c_int64(n), C.c_uint8(int(fill)), C.c_void_p(out.ctypes.data))
    return out


def reduce_column(col, n):
    keep = []
    cc = _cols([col], keep)
    out = (C.c_double * 4)()
    cnt = C.c_int64(0)
    rc = lib().oracle_reduce_column(cc, C.c_int64(n), out, C.byref(cnt))
    if rc:
        raise OracleError(rc)
    return np.array(list(out)), cnt.value


class K1(C.Structure):
    """oracle_k1 (pandrs_oracle.c): the three K1 families restated."""
    _fields_ = [("a_empty", C.c_int32), ("a_sum", C.c_double), ("a_mean", C.c_double), ("a_min", C.c_double), ("a_max", C.c_double),
                ("b_data_empty", C.c_int32), ("b_mean_none", C.c_int32), ("b_minmax_none", C.c_int32),
                ("b_sum_f64", C.c_double), ("b_mean", C.c_double), ("b_min", C.c_double), ("b_max", C.c_double),
                ("b_sum_i64", C.c_int64), ("b_min_i64", C.c_int64), ("b_max_i64", C.c_int64),
                ("c_sum_f64", C.c_double), ("c_mean_f64", C.c_double), ("c_min_f64", C.c_double), ("c_max_f64", C.c_double),
                ("c_sum_i64", C.c_int64), ("c_mean_i64", C.c_int64), ("c_min_i64", C.c_int64), ("c_max_i64", C.c_int64)]


def k1_stats(col, n):
    keep = []
    cc = _cols([col], keep)
    out = K1()
    rc = lib().oracle_k1_stats(cc, C.c_int64(n), C.byref(out))
    if rc:
        raise OracleError(rc)
    return out


def join_groupby_sum(lkey, lval, n_left, rkey, rgroup, n_right):
    keep = []
    a, b, c, d = (_cols([x], keep) for x in (lkey, lval, rkey, rgroup))
    ng = C.c_int64(0)
    pk, pn, pa = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = lib().oracle_join_groupby_sum(a, b, C.c_int64(n_left), c, d, C.c_int64(n_right),
                                       C.byref(ng), C.byref(pk), C.byref(pn), C.byref(pa))
    if rc:
        raise OracleError(rc)
    g = ng.value
    return (_take(pk.value, (1, g), np.uint64), _take(pn.value, (1, g), np.uint8),
            _take(pa.value, (1, g), np.float64))


def groupby_typed_mt(keys_i64, vals_f64, n_threads):
    """Fair typed CPU baseline, NOT the reference's algorithm: one i64 key, f64
    columns, all cores.  -> (keys u64[G], stats[G, 1 + 3 * n_vals]: count, then sum/min/max per column)."""
    keys_i64 = np.ascontiguousarray(keys_i64, np.int64)
    vals = [np.ascontiguousarray(v, np.float64) for v in vals_f64]
    n, nv = len(keys_i64), len(vals)
    ptrs = (C.c_void_p * max(nv, 1))(*[v.ctypes.data for v in vals])
    ng = C.c_int64(0)
    pk, ps = C.c_void_p(), C.c_void_p()
    rc = lib().oracle_groupby_typed_mt(C.c_void_p(keys_i64.ctypes.data), C.c_int64(n), ptrs, C.c_int(nv),
                                       C.c_int(int(n_threads)), C.byref(ng), C.byref(pk), C.byref(ps))
    if rc:
        raise OracleError(rc)
    g = ng.value
    return _take(pk.value, (g,), np.uint64), _take(ps.value, (g, 1 + 3 * nv), np.float64)


def pack_mask(nulls):
    """bool array -> LSB-first bitmap (reference: create_bitmask, src/core/column.rs:163-177)."""
    nulls = np.asarray(nulls, dtype=bool)
    return np.packbits(nulls, bitorder="little")
