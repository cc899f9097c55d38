"""pandrs_amd — MI355X-native groupby-aggregate / hash-join engine behind PandRS's GroupBy / agg /
join API.  The compute path is libpandrs_hip.so (hand-written HIP for gfx950) reached through the
C ABI in include/pandrs_hip.h; this package is the host-side mirror of the reference interface.
There is no CPU fallback: without the built library or a gfx950 device, calls fail loudly."""
from . import _lib
from ._lib import (I64, F64, U32CODE, BOOLBITS, CELL64, SUM, MEAN, MIN, MAX, COUNT, STD, VAR, MEDIAN, FIRST,
                   LAST, CUSTOM, NUNIQUE, INNER, LEFT, RIGHT, OUTER)
from .engine import Context, PandrsHipError, ColumnTypeMismatch, OperationFailed, EmptyError, BelowThreshold

__all__ = ["Context", "PandrsHipError", "ColumnTypeMismatch", "OperationFailed", "EmptyError", "BelowThreshold", "_lib",
           "I64", "F64", "U32CODE", "BOOLBITS", "CELL64", "SUM", "MEAN", "MIN", "MAX", "COUNT", "STD", "VAR",
           "MEDIAN", "FIRST", "LAST", "CUSTOM", "NUNIQUE", "INNER", "LEFT", "RIGHT", "OUTER"]
