"""ctypes binding of libpandrs_hip.so — the exact stub a host-language shim would generate from
include/pandrs_hip.h (see INTEGRATION.md for the Rust `extern "C"` equivalent).

There is NO CPU fallback in this package: if the HIP library is missing, or no gfx950 device is
present, loading / context creation fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PANDRS_HIP_LIB: another build of the same library (A/B runs of two kernel versions on one box)
LIB_PATH = os.environ.get("PANDRS_HIP_LIB") or os.path.join(_HERE, "libpandrs_hip.so")

MAX_PHASES = 12
PHASE_NAMES = ["stage_in", "estimate", "histogram", "scan", "scatter", "aggregate",
               "build", "probe", "gather", "other", "prepartition", ""]

# enums (include/pandrs_hip.h)
I64, F64, U32CODE, BOOLBITS, CELL64 = 0, 1, 2, 3, 4
SUM, MEAN, MIN, MAX, COUNT, STD, VAR, MEDIAN, FIRST, LAST, CUSTOM, NUNIQUE = range(12)
INNER, LEFT, RIGHT, OUTER = range(4)
MEM_HOST, MEM_DEVICE = 0, 1
OK, ERR_INVALID_ARGUMENT, ERR_TYPE_MISMATCH, ERR_OPERATION_FAILED, ERR_COMPUTATION, \
    ERR_OUT_OF_MEMORY, ERR_NOT_INITIALIZED, ERR_BELOW_THRESHOLD = range(8)


class Config(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("device_id", C.c_int32), ("memory_limit", C.c_int64),
                ("fallback_to_cpu", C.c_int32), ("use_pinned_memory", C.c_int32),
                ("min_size_threshold", C.c_int64)]


class Column(C.Structure):
    _fields_ = [("data", C.c_void_p), ("null_mask", C.c_void_p),
                ("dtype", C.c_int32), ("reserved", C.c_int32)]


class AggSpec(C.Structure):
    _fields_ = [("col", C.c_int32), ("op", C.c_int32)]


class ColumnStats(C.Structure):            # pandrs_hip_column_stats
    _fields_ = [("count", C.c_int64), ("count_finite", C.c_int64), ("sum_f64", C.c_double), ("sum_sq", C.c_double),
                ("sum_i64", C.c_int64), ("min_i64", C.c_int64), ("max_i64", C.c_int64),
                ("min", C.c_double), ("max", C.c_double), ("min_finite", C.c_double), ("max_finite", C.c_double)]


class Timings(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("phase_ms", C.c_double * MAX_PHASES),
                ("algorithmic_bytes", C.c_int64), ("n_partitions", C.c_int64),
                ("table_slots", C.c_int64), ("retries", C.c_int64), ("estimated_groups", C.c_int64),
                ("absorbed_rows", C.c_int64)]


# pandrs_hip_transport: the exchange's collectives as host callbacks (pandrs_hip_comm_adopt_transport)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
ALL_REDUCE_MAX_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_int64), C.c_int32)
ALL_TO_ALL_V_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                              C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", ALL_GATHER_FN), ("all_reduce_max_i64", ALL_REDUCE_MAX_FN),
                ("all_to_all_v", ALL_TO_ALL_V_FN)]


# every symbol include/pandrs_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "pandrs_hip_abi_version": (C.c_int32, []),
    "pandrs_hip_init": (C.c_int32, [C.POINTER(Config)]),
    "pandrs_hip_shutdown": (C.c_int32, []),
    "pandrs_hip_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "pandrs_hip_last_error": (C.c_char_p, []),
    "pandrs_hip_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(_P)]),
    "pandrs_hip_ctx_destroy": (C.c_int32, [_P]),
    "pandrs_hip_ctx_synchronize": (C.c_int32, [_P]),
    "pandrs_hip_ctx_reserve": (C.c_int32, [_P, C.c_int64]),
    "pandrs_hip_alloc_events": (C.c_int32, [C.POINTER(C.c_int64)]),
    "pandrs_hip_ctx_set_option": (C.c_int32, [_P, C.c_char_p, C.c_int64]),
    "pandrs_hip_get_timings": (C.c_int32, [_P, C.POINTER(Timings)]),
    "pandrs_hip_column_upload": (C.c_int32, [_P, C.POINTER(Column), C.c_int64, C.POINTER(Column)]),
    "pandrs_hip_column_release": (C.c_int32, [_P, C.POINTER(Column)]),
    "pandrs_hip_resident_bytes": (C.c_int32, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pandrs_hip_groupby_agg": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int32, C.c_int64,
                                           C.POINTER(Column), C.c_int32, C.POINTER(AggSpec), C.c_int32,
                                           C.POINTER(C.c_int64)]),
    "pandrs_hip_groupby_fetch": (C.c_int32, [_P, C.c_int32, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "pandrs_hip_groupby_partials": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int32, C.c_int64,
                                                C.POINTER(Column), C.c_int32, C.POINTER(AggSpec), C.c_int32,
                                                C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "pandrs_hip_partials_split": (C.c_int32, [_P, C.c_int32, C.c_int32, _P, C.POINTER(C.c_int64)]),
    "pandrs_hip_groupby_merge": (C.c_int32, [_P, C.c_int32, C.c_int32, _P, C.c_int64,
                                             C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_uint8),
                                             C.POINTER(AggSpec), C.c_int32, C.POINTER(C.c_int64)]),
    "pandrs_hip_groupby_indices": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int32, C.c_int64,
                                               C.POINTER(C.c_int64)]),
    "pandrs_hip_groupby_indices_fetch": (C.c_int32, [_P, C.c_int32, C.POINTER(_P), C.POINTER(_P), _P, _P]),
    "pandrs_hip_shuffle_split": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.POINTER(Column), C.c_int32, C.c_int64,
                                             C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pandrs_hip_shuffle_fetch": (C.c_int32, [_P, C.c_int32, _P, _P, C.POINTER(_P), C.POINTER(_P)]),
    "pandrs_hip_bytes_to_bitmap": (C.c_int32, [_P, C.c_int32, _P, C.c_int64, _P]),
    "pandrs_hip_key_hash_cells": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int32, C.c_int64, _P]),
    "pandrs_hip_gather_column": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int64, _P, C.c_int64, C.c_uint64, _P]),
    "pandrs_hip_join_gather": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int64, C.c_int32, C.c_uint64, C.c_int32, _P]),
    "pandrs_hip_join_gather_key": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int64, C.POINTER(Column), C.c_int64, C.c_uint64, C.c_int32, _P]),
    "pandrs_hip_reduce_moments": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int64, C.POINTER(C.c_double),
                                              C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pandrs_hip_join_indices": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int64,
                                            C.POINTER(Column), C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    "pandrs_hip_join_fetch": (C.c_int32, [_P, C.c_int32, _P, _P]),
    "pandrs_hip_gather_i64": (C.c_int32, [_P, C.c_int32, _P, _P, _P, C.c_int64, C.c_int64, _P]),
    "pandrs_hip_gather_f64": (C.c_int32, [_P, C.c_int32, _P, _P, _P, C.c_int64, C.c_double, _P]),
    "pandrs_hip_gather_u32": (C.c_int32, [_P, C.c_int32, _P, _P, _P, C.c_int64, C.c_uint32, _P]),
    "pandrs_hip_gather_bool": (C.c_int32, [_P, C.c_int32, _P, _P, _P, C.c_int64, C.c_uint8, _P]),
    "pandrs_hip_join_groupby_sum": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.POINTER(Column), C.c_int64,
                                                C.POINTER(Column), C.POINTER(Column), C.c_int64,
                                                C.POINTER(C.c_int64)]),
    "pandrs_hip_reduce_column": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int64,
                                             C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pandrs_hip_comm_unique_id": (C.c_int32, [C.c_char_p]),
    "pandrs_hip_comm_init": (C.c_int32, [_P, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "pandrs_hip_comm_adopt": (C.c_int32, [_P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "pandrs_hip_comm_adopt_transport": (C.c_int32, [C.POINTER(Transport), C.c_int32, C.c_int32, C.POINTER(_P)]),
    "pandrs_hip_comm_destroy": (C.c_int32, [_P]),
    "pandrs_hip_dist_groupby_agg": (C.c_int32, [_P, _P, C.c_int32, C.POINTER(Column), C.c_int32, C.c_int64,
                                                C.POINTER(Column), C.c_int32, C.POINTER(AggSpec), C.c_int32,
                                                C.POINTER(C.c_int64)]),
    "pandrs_hip_dist_join_groupby_sum": (C.c_int32, [_P, _P, C.c_int32, C.POINTER(Column), C.POINTER(Column), C.c_int64,
                                                     C.POINTER(Column), C.POINTER(Column), C.c_int64, C.POINTER(C.c_int64)]),
    "pandrs_hip_reduce_stats": (C.c_int32, [_P, C.c_int32, C.POINTER(Column), C.c_int64, C.POINTER(ColumnStats)]),
}

_lib = None


class LibraryMissing(ImportError):
    pass


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (same soname, libamdhip64.so.7, as /opt/rocm's); when torch is installed we map
    its copy first so libpandrs_hip.so's DT_NEEDED resolves to it and device pointers, streams and
    RCCL buffers are shared.  Without torch the system ROCm runtime is used.  torch itself is NOT
    imported here."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """dlopen the in-tree library and type every entry point.  Raises LibraryMissing if the
    HIP extension has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            "%s not found — build the HIP extension first (make -C pandrs_amd/csrc, or "
            "__graft_entry__.build()).  pandrs_amd has no CPU fallback." % LIB_PATH)
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.pandrs_hip_abi_version() != 1:
        raise LibraryMissing("ABI version mismatch: library %d, binding 1" % lib.pandrs_hip_abi_version())
    _lib = lib
    return lib


def last_error():
    return load().pandrs_hip_last_error().decode(errors="replace")
