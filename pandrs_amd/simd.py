"""The reference's slice-level reductions (src/optimized/jit/simd.rs:9-112) on the device: one HBM pass
(pandrs_hip_reduce_stats) behind the same names and the same corner cases — simd_mean_i64 is an INTEGER
division (:77-82), empty slices give 0 / 0.0 and the fold identities (+-inf, i64::MAX / MIN, tests :493-506).
`data` is a numpy array (staged over PCIe) or a torch CUDA tensor (read in place)."""
from . import _lib as L

I64_MAX, I64_MIN = 2**63 - 1, -2**63


def _stats(data, dtype):
    from .frame import get_context
    n = int(data.shape[0])
    return n, get_context().column_stats((data, None, dtype), n)


def simd_sum_f64(data):
    return float(_stats(data, L.F64)[1]["sum_f64"])


def simd_mean_f64(data):
    n, st = _stats(data, L.F64)
    return float(st["sum_f64"]) / n if n else 0.0


def simd_min_f64(data):
    return float(_stats(data, L.F64)[1]["min"])        # +inf when empty (fold identity)


def simd_max_f64(data):
    return float(_stats(data, L.F64)[1]["max"])


def simd_sum_i64(data):
    return int(_stats(data, L.I64)[1]["sum_i64"])


def simd_mean_i64(data):
    n, st = _stats(data, L.I64)
    if n == 0:
        return 0
    s = int(st["sum_i64"])
    q = abs(s) // n                                    # Rust's `/` truncates toward zero
    return q if s >= 0 else -q


def simd_min_i64(data):
    n, st = _stats(data, L.I64)
    return int(st["min_i64"]) if n else I64_MAX


def simd_max_i64(data):
    n, st = _stats(data, L.I64)
    return int(st["max_i64"]) if n else I64_MIN
