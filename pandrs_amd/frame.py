"""Host-side mirror of the reference's interface for the hot path — same names, argument meaning and
error behaviour — so the parity tests read like the reference's own tests.

  reference (file:line)                                               here
  OptimizedDataFrame  src/optimized/split_dataframe/core.rs:14-25     OptimizedDataFrame
  Int64Column ...     src/column/*.rs                                 Int64Column / Float64Column / StringColumn / BooleanColumn
  GLOBAL_STRING_POOL  src/column/string_pool.rs:6-53                  GLOBAL_STRING_POOL
  group_by            src/optimized/split_dataframe/group/grouping.rs:22      OptimizedDataFrame.group_by
  GroupBy.aggregate   src/optimized/split_dataframe/group/aggregation.rs:763  GroupBy.aggregate
  sum/mean/../agg     src/optimized/split_dataframe/group/operations.rs:438-521  GroupBy.sum ... GroupBy.agg
  *_join / join_impl  src/optimized/split_dataframe/join.rs:32-555    OptimizedDataFrame.inner_join ...
  LazyFrame           src/optimized/lazy.rs:98-170, :186-425          LazyFrame
  AggregateOp         src/optimized/split_dataframe/group/types.rs:11-34   AggregateOp
  JoinType            src/optimized/split_dataframe/join.rs:11-20     JoinType

This is what the Rust call-outs of INTEGRATION.md §3 do: adapt columns to the C ABI, call the
device engine, stringify group keys, name result columns.  All arithmetic happens in
libpandrs_hip.so; there is no CPU implementation of groupby or join here.
"""
from decimal import Decimal
from enum import IntEnum

import numpy as np

from . import _lib as L
from .engine import Context, PandrsHipError, ColumnTypeMismatch, OperationFailed, EmptyError


class AggregateOp(IntEnum):          # types.rs:11-34, same order
    Sum = 0
    Mean = 1
    Min = 2
    Max = 3
    Count = 4
    Std = 5
    Var = 6
    Median = 7
    First = 8
    Last = 9
    Custom = 10
    Nunique = 11                     # not in the reference's enum: the legacy AggFunc::Nunique (src/dataframe/groupby.rs:41)


class JoinType(IntEnum):             # join.rs:11-20
    Inner = 0
    Left = 1
    Right = 2
    Outer = 3


class ColumnNotFound(KeyError):      # Error::ColumnNotFound (grouping.rs:55, join.rs:87)
    pass


class DuplicateColumnName(ValueError):
    pass


class InconsistentRowCount(ValueError):
    pass


_OP_NAME = {AggregateOp.Sum: "sum", AggregateOp.Mean: "mean", AggregateOp.Min: "min", AggregateOp.Max: "max",
            AggregateOp.Count: "count", AggregateOp.Std: "std", AggregateOp.Var: "var",
            AggregateOp.Median: "median", AggregateOp.First: "first", AggregateOp.Last: "last",
            AggregateOp.Custom: "custom", AggregateOp.Nunique: "nunique"}       # operations.rs:501-514


# ---------------------------------------------------------------------------------------------- columns
class StringPool:
    """Process-global string pool: equal string <=> equal u32 code (string_pool.rs:28-53)."""

    def __init__(self):
        self._code = {}
        self._strings = []

    def get_or_insert(self, s):
        c = self._code.get(s)
        if c is None:
            c = len(self._strings)
            self._code[s] = c
            self._strings.append(s)
        return c

    def get(self, code):
        return self._strings[code]

    def __len__(self):
        return len(self._strings)


GLOBAL_STRING_POOL = StringPool()


def create_bitmask(nulls):
    """bool list -> LSB-first bitmap, 1 = null (src/core/column.rs:163-177)."""
    return np.packbits(np.asarray(nulls, dtype=bool), bitorder="little")


class _Column:
    dtype = None
    type_name = None

    def __init__(self, data, null_mask, length):
        self.data = data
        self.null_mask = null_mask
        self.length = length

    def len(self):
        return self.length

    def __len__(self):
        return self.length

    def column_type(self):
        return self.type_name

    def is_null(self, i):
        return self.null_mask is not None and bool((self.null_mask[i >> 3] >> (i & 7)) & 1)

    def view(self):
        """(data, null_mask, dtype) triple for the engine."""
        return (self.data, self.null_mask, self.dtype)

    @staticmethod
    def _mask(nulls):
        if nulls is None or not np.any(nulls):
            return None
        return create_bitmask(nulls)


class _NumericColumn(_Column):
    """The null-skipping column folds of src/column/{int64,float64}_column.rs:100-199 on the device
    (pandrs_hip_reduce_stats): None exactly where the reference returns None."""

    def _stats(self):
        return get_context().column_stats(self.view(), self.length)

    def sum(self):
        if self.length == 0:
            return 0 if self.dtype == L.I64 else 0.0
        st = self._stats()
        return int(st["sum_i64"]) if self.dtype == L.I64 else float(st["sum_f64"])

    def mean(self):
        if self.length == 0:
            return None
        st = self._stats()
        if st["count"] == 0:
            return None
        return (float(st["sum_i64"]) if self.dtype == L.I64 else st["sum_f64"]) / st["count"]

    def _extreme(self, which):
        if self.length == 0:
            return None
        st = self._stats()
        if self.dtype == L.I64:
            return None if st["count"] == 0 else int(st[which + "_i64"])
        return None if st["count_finite"] == 0 else float(st[which + "_finite"])      # non-finite values are skipped

    def min(self):
        return self._extreme("min")

    def max(self):
        return self._extreme("max")


class Int64Column(_NumericColumn):   # src/column/int64_column.rs:52-66
    dtype, type_name = L.I64, "Int64"

    def __init__(self, data, nulls=None):
        data = np.ascontiguousarray(data, dtype=np.int64)
        super().__init__(data, self._mask(nulls), len(data))

    @classmethod
    def with_nulls(cls, data, nulls):
        return cls(data, nulls)

    def get(self, i):
        return None if self.is_null(i) else int(self.data[i])


class Float64Column(_NumericColumn): # src/column/float64_column.rs:9-13
    dtype, type_name = L.F64, "Float64"

    def __init__(self, data, nulls=None):
        data = np.ascontiguousarray(data, dtype=np.float64)
        super().__init__(data, self._mask(nulls), len(data))

    @classmethod
    def with_nulls(cls, data, nulls):
        return cls(data, nulls)

    def get(self, i):
        return None if self.is_null(i) else float(self.data[i])


class StringColumn(_Column):         # src/column/string_column.rs:26-72 (GlobalPool mode)
    dtype, type_name = L.U32CODE, "String"

    def __init__(self, values, nulls=None, _codes=None):
        if _codes is None:
            _codes = np.fromiter((GLOBAL_STRING_POOL.get_or_insert(s) for s in values), dtype=np.uint32,
                                 count=len(values))
        super().__init__(np.ascontiguousarray(_codes, dtype=np.uint32), self._mask(nulls), len(_codes))

    @classmethod
    def with_nulls(cls, values, nulls):
        return cls(values, nulls)

    @classmethod
    def from_codes(cls, codes, null_mask=None):
        c = cls([], _codes=codes)
        c.null_mask = null_mask
        return c

    def get(self, i):
        return None if self.is_null(i) else GLOBAL_STRING_POOL.get(int(self.data[i]))

    def to_list(self):
        return [self.get(i) for i in range(self.length)]


class BooleanColumn(_Column):        # src/column/boolean_column.rs:10-15 (bit-packed)
    dtype, type_name = L.BOOLBITS, "Boolean"

    def __init__(self, data, nulls=None, _bits=None, _length=None):
        if _bits is None:
            data = np.asarray(data, dtype=bool)
            _bits, _length = np.packbits(data, bitorder="little"), len(data)
        super().__init__(_bits, self._mask(nulls), _length)

    @classmethod
    def with_nulls(cls, data, nulls):
        return cls(data, nulls)

    def get(self, i):
        return None if self.is_null(i) else bool((self.data[i >> 3] >> (i & 7)) & 1)


def rust_f64_to_string(v):
    """f64::to_string(): shortest round-trip digits, never an exponent ("1", "0.1", "NaN", "inf", "-0")."""
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "inf" if v > 0 else "-inf"
    s = format(Decimal(repr(float(v))), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    if s in ("0", "-0"):
        return "-0" if np.signbit(v) else "0"
    return s


def _key_strings(dtype, cells, nulls, null_string="NULL"):
    """Group-key cells -> the strings the reference's result frame holds (grouping.rs:69-98)."""
    out = []
    for cell, nul in zip(cells.tolist(), nulls.tolist()):
        if nul:
            out.append(null_string)
        elif dtype == L.I64:
            out.append(str(int(np.int64(np.uint64(cell)))))
        elif dtype == L.F64:
            out.append(rust_f64_to_string(float(np.uint64(cell).view(np.float64))))
        elif dtype == L.U32CODE:
            out.append(GLOBAL_STRING_POOL.get(int(cell)))
        else:
            out.append("true" if cell else "false")
    return out


# ---------------------------------------------------------------------------------------------- context
_default_ctx = None


def get_context():
    """Lazily created process-wide engine context (cf. get_gpu_manager, src/gpu/mod.rs:249-282)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ---------------------------------------------------------------------------------------------- frame
class OptimizedDataFrame:
    def __init__(self):
        self.columns = []
        self.column_names = []
        self.column_indices = {}
        self._row_count = 0
        self.multi_index = None           # list of key tuples when a multi-key group_by built a multi-index
        self.multi_index_names = None

    # -- construction / inspection (core.rs) ----------------------------------------------------------
    def add_column(self, name, column):
        if name in self.column_indices:
            raise DuplicateColumnName(name)
        if self.columns and column.len() != self._row_count:
            raise InconsistentRowCount("expected %d rows, found %d" % (self._row_count, column.len()))
        self.column_indices[name] = len(self.columns)
        self.column_names.append(name)
        self.columns.append(column)
        self._row_count = column.len()
        return self

    def column(self, name):
        if name not in self.column_indices:
            raise ColumnNotFound(name)
        return self.columns[self.column_indices[name]]

    def contains_column(self, name):
        return name in self.column_indices

    def row_count(self):
        return self._row_count

    def column_count(self):
        return len(self.columns)

    # -- groupby (grouping.rs:22-115) -------------------------------------------------------------------
    def group_by(self, columns):
        return self.group_by_with_options(columns, True)      # grouping.rs:22-28

    def group_by_with_options(self, columns, as_multi_index):
        columns = [columns] if isinstance(columns, str) else list(columns)
        for c in columns:
            if c not in self.column_indices:
                raise ColumnNotFound(c)                       # grouping.rs:53-57
        return GroupBy(self, columns, as_multi_index and len(columns) > 1)   # grouping.rs:107

    def par_groupby(self, group_by_columns):
        """par_groupby (grouping.rs:124-331): {joined key -> sub-frame}.  Keys are the parts joined
        with "_" (:186), a null part is "NA" (:158); tuples that collide after joining share a
        group, as in the reference.  The row lists come from the device (groupby_indices), every
        sub-frame is a device gather of all columns (:286-328)."""
        cols = [group_by_columns] if isinstance(group_by_columns, str) else list(group_by_columns)
        for c in cols:
            if c not in self.column_indices:
                raise ColumnNotFound(c)
        gb = GroupBy(self, cols, False)
        merged = {}
        for key, rows in gb._group_rows(null_string="NA").items():
            name = "_".join(key)
            if name in merged:
                merged[name] = np.sort(np.concatenate([merged[name], rows]))
            else:
                merged[name] = rows
        return {name: self.filter_by_indices(rows) for name, rows in merged.items()}

    def filter_by_indices(self, indices):
        """data_ops.rs:124-209: row gather of every column; nulls become 0 / 0.0 / "" / false and the
        result carries no masks; out-of-range indices are dropped."""
        idx = np.asarray(indices, dtype=np.int64)
        idx = idx[(idx >= 0) & (idx < self._row_count)]
        result = OptimizedDataFrame()
        if not self.columns:
            return result
        g = _Gatherer(get_context(), idx, idx)
        for name in self.column_names:
            result.add_column(name, g.take(self.column(name), left=True))
        return result

    # -- whole-column reductions (K1: split_dataframe/aggregate.rs:21-215) ----------------------------------
    def _stats(self, name):
        col = self.column(name)
        if col.dtype not in (L.I64, L.F64):        # Error::Type (aggregate.rs:57)
            raise ColumnTypeMismatch(L.ERR_TYPE_MISMATCH, "Column '%s' is not a numeric type" % name)
        return get_context().column_stats(col.view(), col.len())

    def sum(self, name):
        """aggregate.rs:21-62: the non-null values as f64 (`v as f64` for Int64), 0.0 when there is none."""
        return float(self._stats(name)["sum_f64"])

    def mean(self, name):
        st = self._stats(name)
        if st["count"] == 0:                       # aggregate.rs:87, :99
            raise EmptyError(L.ERR_OPERATION_FAILED, "Column '%s' is empty" % name)
        return float(st["sum_f64"]) / st["count"]

    def min(self, name):
        st = self._stats(name)
        if st["count"] == 0:                       # aggregate.rs:185, :198
            raise EmptyError(L.ERR_OPERATION_FAILED, "Column '%s' is empty" % name)
        return float(st["min"])                    # fold(+inf, f64::min): NaN dropped, infinities kept

    def max(self, name):
        st = self._stats(name)
        if st["count"] == 0:                       # aggregate.rs:135, :148
            raise EmptyError(L.ERR_OPERATION_FAILED, "Column '%s' is empty" % name)
        return float(st["max"])

    # -- joins (join.rs:32-73) -----------------------------------------------------------------------------
    def inner_join(self, other, left_on, right_on):
        return self._join_impl(other, left_on, right_on, JoinType.Inner)

    def left_join(self, other, left_on, right_on):
        return self._join_impl(other, left_on, right_on, JoinType.Left)

    def right_join(self, other, left_on, right_on):
        return self._join_impl(other, left_on, right_on, JoinType.Right)

    def outer_join(self, other, left_on, right_on):
        return self._join_impl(other, left_on, right_on, JoinType.Outer)

    def _join_impl(self, other, left_on, right_on, how):
        if left_on not in self.column_indices:
            raise ColumnNotFound(left_on)                     # join.rs:84-87
        if right_on not in other.column_indices:
            raise ColumnNotFound(right_on)                    # join.rs:89-92
        lcol, rcol = self.column(left_on), other.column(right_on)
        ctx = get_context()
        # dtype mismatch => ColumnTypeMismatch raised by the library (join.rs:98-104)
        li, ri = ctx.join_indices(lcol.view(), lcol.len(), rcol.view(), rcol.len(), int(how))
        result = OptimizedDataFrame()
        if len(li) == 0:
            # empty result: NON-KEY columns only, suffix decided against the LEFT frame (join.rs:227-284)
            for name in self.column_names:
                if name != left_on:
                    result.add_column(name, _empty_like(self.column(name)))
            for name in other.column_names:
                if name != right_on:
                    new = name + "_right" if name in self.column_indices else name
                    result.add_column(new, _empty_like(other.column(name)))
            return result
        g = _Gatherer(ctx, li, ri)
        for name in self.column_names:                        # left non-key columns (join.rs:290-361)
            if name != left_on:
                result.add_column(name, g.take(self.column(name), left=True))
        result.add_column(left_on, g.take_key(lcol, rcol))    # key: left value, else right (join.rs:364-472)
        for name in other.column_names:                       # right non-key columns (join.rs:475-552)
            if name != right_on:
                new = name + "_right" if name in result.column_indices else name
                result.add_column(new, g.take(other.column(name), left=False))
        return result


def _empty_like(col):
    if isinstance(col, Int64Column):
        return Int64Column([])
    if isinstance(col, Float64Column):
        return Float64Column([])
    if isinstance(col, StringColumn):
        return StringColumn([])
    return BooleanColumn([])


class _Gatherer:
    """Column gathers of join_impl on the device: misses / nulls become 0 / 0.0 / "" / false and the
    result columns carry NO null mask (join.rs:304-307, :319-322)."""

    def __init__(self, ctx, li, ri):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.dev = "cuda:%d" % ctx.device
        self.li = torch.from_numpy(np.ascontiguousarray(li)).to(self.dev)
        self.ri = torch.from_numpy(np.ascontiguousarray(ri)).to(self.dev)

    def _up(self, a):
        if a is None:
            return None
        if a.dtype == np.uint32:
            a = a.view(np.int32)
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def _gather(self, col, idx):
        fill = GLOBAL_STRING_POOL.get_or_insert("") if col.dtype == L.U32CODE else 0
        if col.len() == 0:
            n = idx.numel()
            return {L.I64: np.zeros(n, np.int64), L.F64: np.zeros(n, np.float64),
                    L.U32CODE: np.full(n, fill, np.uint32), L.BOOLBITS: np.zeros(n, np.uint8)}[col.dtype]
        out = self.ctx.gather(self._up(col.data), self._up(col.null_mask), idx, fill, col.dtype).cpu().numpy()
        return out.view(np.uint32) if col.dtype == L.U32CODE else out

    @staticmethod
    def _wrap(col, arr):
        if col.dtype == L.I64:
            return Int64Column(arr)
        if col.dtype == L.F64:
            return Float64Column(arr)
        if col.dtype == L.U32CODE:
            return StringColumn.from_codes(arr)
        return BooleanColumn(None, _bits=np.packbits(arr.astype(bool), bitorder="little"), _length=len(arr))

    def take(self, col, left):
        return self._wrap(col, self._gather(col, self.li if left else self.ri))

    def take_key(self, lcol, rcol):
        a = self._gather(lcol, self.li)
        b = self._gather(rcol, self.ri)
        from_left = (self.li >= 0).cpu().numpy()
        return self._wrap(lcol, np.where(from_left, a, b))


# ---------------------------------------------------------------------------------------------- GroupBy
class GroupBy:
    """GroupBy<'a> (group/types.rs:46-55).  Aggregations run on the device straight from the
    columns; the row-index map `groups` (a pub field in the reference, types.rs:52) is built on the
    device too, but only when something reads it (closures: filter / custom aggregations)."""

    def __init__(self, df, group_by_columns, create_multi_index=False):
        self.df = df
        self.group_by_columns = group_by_columns
        self.create_multi_index = create_multi_index
        self._groups = None

    def _group_rows(self, null_string="NULL"):
        """{tuple(key strings) -> ascending row indices (numpy)} from pandrs_hip_groupby_indices."""
        key_cols = [self.df.column(k) for k in self.group_by_columns]
        cells, nulls, off, rows = get_context().groupby_indices([k.view() for k in key_cols], self.df.row_count())
        strs = [_key_strings(k.dtype, cells[i], nulls[i], null_string) for i, k in enumerate(key_cols)]
        return {key: rows[off[g]:off[g + 1]] for g, key in enumerate(zip(*strs))}

    @property
    def groups(self):
        """HashMap<Vec<String>, Vec<usize>> of the reference (grouping.rs:62-104)."""
        if self._groups is None:
            self._groups = {k: v.tolist() for k, v in self._group_rows().items()}
        return self._groups

    def _values(self, column, rows):
        """The group's non-null values of a numeric column as f64 (aggregation.rs:440-455)."""
        col = self.df.column(column)
        if col.dtype not in (L.I64, L.F64):
            raise OperationFailed(L.ERR_OPERATION_FAILED, "column '%s' is not numeric" % column)
        out = []
        for r in rows:
            v = col.get(r)
            if v is not None:
                out.append(float(v))
        return out

    def aggregate_custom(self, aggregations):
        """aggregations: iterable of (column, fn(list[float]) -> float, result_name)
        (CustomAggregation, types.rs:58-67; aggregate_custom, aggregation.rs:391-497): the closure
        runs on the host over the group's non-null values (Int64 cast to f64), groups from the device."""
        aggregations = list(aggregations)
        for column, fn, _ in aggregations:
            if fn is None:
                raise OperationFailed(L.ERR_OPERATION_FAILED,
                                      "Custom aggregation function is required for AggregateOp::Custom")
            if column not in self.df.column_indices:
                raise ColumnNotFound(column)
        result = OptimizedDataFrame()
        groups = self.groups
        for i, name in enumerate(self.group_by_columns):
            result.add_column(name, StringColumn([k[i] for k in groups]))
        for column, fn, result_name in aggregations:
            result.add_column(result_name, Float64Column([float(fn(self._values(column, rows))) for rows in groups.values()]))
        return result

    def custom(self, column, result_name, func):          # operations.rs:606
        return self.aggregate_custom([(column, func, result_name)])

    par_custom = custom
    par_aggregate_custom = aggregate_custom

    def filter(self, filter_fn):
        """operations.rs:51-74: keep the rows of the groups whose sub-frame passes filter_fn."""
        keep = [np.asarray(rows, np.int64) for rows in self.groups.values()
                if filter_fn(self.df.filter_by_indices(rows))]
        return self.df.filter_by_indices(np.concatenate(keep) if keep else np.zeros(0, np.int64))

    par_filter = filter

    def transform(self, transform_fn):
        """operations.rs:132-276: transform_fn(group sub-frame) -> frame, for every group; the results are
        concatenated column by column after the first result's schema (columns matched by POSITION, a
        column of another type contributes nothing, like the reference's `if let Some(Column::X(..))`)."""
        outs = [transform_fn(self.df.filter_by_indices(rows)) for rows in self.groups.values()]
        result = OptimizedDataFrame()
        if not outs:
            return result
        template = outs[0]
        for ci, tcol in enumerate(template.columns):
            values = []
            for df in outs:
                if ci < len(df.columns) and type(df.columns[ci]) is type(tcol):
                    col = df.columns[ci]
                    values.extend(col.get(i) for i in range(col.len()))
            nulls = [v is None for v in values]
            fill = {Int64Column: 0, Float64Column: 0.0, StringColumn: "", BooleanColumn: False}[type(tcol)]
            data = [fill if v is None else v for v in values]
            result.add_column(template.column_names[ci], type(tcol)(data, nulls if any(nulls) else None))
        return result

    par_transform = transform

    def aggregate(self, aggregations):
        """aggregations: iterable of (column, AggregateOp, alias)  (aggregation.rs:763-871)."""
        aggregations = [(c, AggregateOp(int(op)), alias) for c, op, alias in aggregations]
        for col_name, _, _ in aggregations:
            if col_name not in self.df.column_indices:
                raise ColumnNotFound(col_name)                # aggregation.rs:770-774
        key_cols = [self.df.column(k) for k in self.group_by_columns]
        val_names = []
        for col_name, _, _ in aggregations:
            if col_name not in val_names:
                val_names.append(col_name)
        vals = [self.df.column(n).view() for n in val_names]
        specs = [(val_names.index(c), int(op)) for c, op, _ in aggregations]
        ctx = get_context()
        kc, kn, oa = ctx.groupby_agg([k.view() for k in key_cols], self.df.row_count(), vals, specs)
        result = OptimizedDataFrame()
        key_strings = [_key_strings(k.dtype, kc[i], kn[i]) for i, k in enumerate(key_cols)]
        if self.create_multi_index:
            # > 1 key with the multi-index option: no key columns, a StringMultiIndex of tuples instead
            # (aggregation.rs:812-853)
            result.multi_index = list(zip(*key_strings))
            result.multi_index_names = list(self.group_by_columns)
        else:
            # key columns as strings (aggregation.rs:856-860; their relative order is HashMap order in
            # the reference, group_by order here), then one Float64Column per alias in request order (:863-867)
            for name, strs in zip(self.group_by_columns, key_strings):
                result.add_column(name, StringColumn(strs))
        seen = {}
        for a, (_, _, alias) in enumerate(aggregations):
            seen[alias] = a          # the reference keeps one Vec per alias in a HashMap: last writer wins
        for a, (_, _, alias) in enumerate(aggregations):
            if alias in result.column_indices:
                raise DuplicateColumnName(alias)
            result.add_column(alias, Float64Column(oa[seen[alias]]))
        return result

    def agg(self, aggs):
        """[(column, op)] -> aliases "{col}_{op}" (operations.rs:498-521)."""
        return self.aggregate([(c, op, "%s_%s" % (c, _OP_NAME[AggregateOp(int(op))])) for c, op in aggs])

    par_aggregate = aggregate        # G6: the reference's parallel variant has a known row race
    par_agg = agg

    def _short(self, column, op):
        return self.aggregate([(column, op, "%s_%s" % (column, _OP_NAME[op]))])

    def sum(self, column):
        return self._short(column, AggregateOp.Sum)

    def mean(self, column):
        return self._short(column, AggregateOp.Mean)

    def min(self, column):
        return self._short(column, AggregateOp.Min)

    def max(self, column):
        return self._short(column, AggregateOp.Max)

    def count(self, column):
        return self._short(column, AggregateOp.Count)

    def std(self, column):
        return self._short(column, AggregateOp.Std)

    def var(self, column):
        return self._short(column, AggregateOp.Var)

    def median(self, column):
        return self._short(column, AggregateOp.Median)

    def first(self, column):
        return self._short(column, AggregateOp.First)

    def last(self, column):
        return self._short(column, AggregateOp.Last)

    def nunique(self, column):       # legacy GroupBy::nunique, src/dataframe/groupby.rs:386-393, alias "{col}_nunique"
        return self._short(column, AggregateOp.Nunique)

    # ---- GroupByJitExt (src/optimized/jit/groupby.rs:68-290).  The reference wraps fixed closures in
    # CustomAggregation and runs them through aggregate_custom over each group's non-null f64 values:
    # Kahan sum / mean / std, plain min / max folds, and the chunked "parallel" variants.  The same
    # quantities come out of the device aggregates; what differs from the AggregateOp path is only the
    # value for a group WITHOUT non-null values (min / max folds stay at +-inf instead of 0.0) and the
    # parallel std being the POPULATION std from E[x^2] - E[x]^2 (jit/parallel.rs:222-233).  Kahan and the
    # device's sums agree to ~1e-13 relative on benign data; tests compare with the closures themselves.
    def _jit(self, column, result_name, op, fix=None):
        col = self.df.column(column) if column in self.df.column_indices else None
        if col is None:
            raise ColumnNotFound(column)
        if col.dtype not in (L.I64, L.F64):
            raise OperationFailed(L.ERR_OPERATION_FAILED, "column '%s' is not numeric" % column)
        key_cols = [self.df.column(k) for k in self.group_by_columns]
        n = self.df.row_count()
        vals = [col.view()]
        specs = [(0, int(op)), (0, int(AggregateOp.Count))]
        if col.null_mask is not None:        # nulls per group: the sum of an indicator column
            ind = np.unpackbits(col.null_mask, bitorder="little")[:n].astype(np.int64)
            vals.append((ind, None, L.I64))
            specs.append((1, int(AggregateOp.Sum)))
        kc, kn, oa = get_context().groupby_agg([k.view() for k in key_cols], n, vals, specs)
        nn = oa[1] - (oa[2] if len(specs) == 3 else 0.0)          # non-null values per group
        out = fix(oa[0], nn) if fix else oa[0]
        result = OptimizedDataFrame()
        for i, (name, k) in enumerate(zip(self.group_by_columns, key_cols)):
            result.add_column(name, StringColumn(_key_strings(k.dtype, kc[i], kn[i])))
        result.add_column(result_name, Float64Column(np.asarray(out, np.float64)))
        return result

    def sum_jit(self, column, result_name):                # jit/groupby.rs:69-96
        return self._jit(column, result_name, AggregateOp.Sum)

    def mean_jit(self, column, result_name):               # :98-128, empty -> 0.0
        return self._jit(column, result_name, AggregateOp.Mean)

    def std_jit(self, column, result_name):                # :130-172, n <= 1 -> 0.0, n - 1 denominator
        return self._jit(column, result_name, AggregateOp.Std)

    def var_jit(self, column, result_name):
        return self._jit(column, result_name, AggregateOp.Var)

    def min_jit(self, column, result_name):                # :174-190, fold from +inf: empty stays +inf
        return self._jit(column, result_name, AggregateOp.Min, lambda v, nn: np.where(nn == 0, np.inf, v))

    def max_jit(self, column, result_name):                # :192-208
        return self._jit(column, result_name, AggregateOp.Max, lambda v, nn: np.where(nn == 0, -np.inf, v))

    def parallel_sum_jit(self, column, result_name, config=None):     # :210-228 (Kahan per chunk + Kahan combine)
        return self._jit(column, result_name, AggregateOp.Sum)

    def parallel_mean_jit(self, column, result_name, config=None):    # :230-247, count 0 -> 0.0
        return self._jit(column, result_name, AggregateOp.Mean)

    def parallel_std_jit(self, column, result_name, config=None):     # :249-266 -> parallel_std_f64_value
        def population(v, nn):                                        # sample variance * (n - 1) / n, n <= 1 -> 0.0
            return np.sqrt(np.where(nn > 1, v * (nn - 1) / np.maximum(nn, 1), 0.0))
        return self._jit(column, result_name, AggregateOp.Var, population)

    def aggregate_jit(self, column, func, result_name):               # :268-290: any closure -> host custom path
        return self.aggregate_custom([(column, func, result_name)])

    # operations.rs:550-594: the par_* shortcuts share the exact path (see par_aggregate above)
    par_sum, par_mean, par_min, par_max, par_count = sum, mean, min, max, count
    par_std, par_var, par_median = std, var, median


# ---------------------------------------------------------------------------------------------- LazyFrame
class LazyFrame:
    """LazyFrame (lazy.rs:98-170): only the two operations on the hot path are mirrored."""

    def __init__(self, df):
        self.source = df
        self.operations = []

    @classmethod
    def new(cls, df):
        return cls(df)

    def aggregate(self, group_by, aggregations):
        self.operations.append(("aggregate", list(group_by), list(aggregations)))
        return self

    def join(self, right, left_on, right_on, join_type):
        self.operations.append(("join", right, left_on, right_on, JoinType(int(join_type))))
        return self

    def execute(self):
        df = self.source
        ops = list(self.operations)
        i = 0
        while i < len(ops):
            op = ops[i]
            i += 1
            if op[0] == "aggregate":
                _, group_by, aggregations = op
                for _, agg_op, _ in aggregations:             # lazy.rs:377-382: only these five ops
                    if AggregateOp(int(agg_op)) not in (AggregateOp.Sum, AggregateOp.Mean, AggregateOp.Min,
                                                        AggregateOp.Max, AggregateOp.Count):
                        raise OperationFailed(L.ERR_OPERATION_FAILED,
                                              "Aggregation operation %s is not supported" % AggregateOp(int(agg_op)).name)
                # the arm builds its result inline and never a multi-index: key columns always (lazy.rs:390-394;
                # tests/optimized_groupby_test.rs:184 asserts 3 columns for two keys)
                df = df.group_by_with_options(group_by, False).aggregate(aggregations)
            else:
                _, right, left_on, right_on, jt = op
                # Join(Inner) immediately followed by Aggregate([g], [(v, Sum, alias)]) — lazy.rs:405-425 then :186 — with v a
                # left column and g a right column is BASELINE config 5: one fused device operator
                # (pandrs_hip_join_groupby_sum), the joined rows are never materialised
                if jt == JoinType.Inner and i < len(ops) and ops[i][0] == "aggregate":
                    fused = _fused_join_groupby_sum(df, right, left_on, right_on, ops[i][1], ops[i][2])
                    if fused is not None:
                        df = fused
                        i += 1
                        continue
                df = df._join_impl(right, left_on, right_on, jt)     # lazy.rs:405-425
        return df


def _fused_join_groupby_sum(left, right, left_on, right_on, group_by, aggregations):
    """The shape test of hip_shim.rs `lazy_join_groupby_sum_hip` (same conditions, same result frame); None = not
    that shape, the two arms run one after the other."""
    if len(group_by) != 1 or len(aggregations) != 1 or AggregateOp(int(aggregations[0][1])) != AggregateOp.Sum:
        return None
    group_col, (value_col, _, alias) = group_by[0], aggregations[0]
    if left_on not in left.column_indices or right_on not in right.column_indices:
        return None                                          # the join arm raises ColumnNotFound
    if value_col == left_on or not left.contains_column(value_col) or left.contains_column(group_col):
        return None
    if group_col.endswith("_right") and left.contains_column(group_col[:-6]) and right.contains_column(group_col[:-6]):
        right_name = group_col[:-6]                          # join.rs:478-482
    elif right.contains_column(group_col):
        right_name = group_col
    else:
        return None
    if right_name == right_on:
        return None
    lk, lv, rk, rg = left.column(left_on), left.column(value_col), right.column(right_on), right.column(right_name)
    if lk.dtype != rk.dtype or lv.dtype not in (L.I64, L.F64) or rg.dtype == L.BOOLBITS or rg.null_mask is not None:
        return None
    kc, kn, sums = get_context().join_groupby_sum(lk.view(), lv.view(), left.row_count(), rk.view(), rg.view(), right.row_count())
    result = OptimizedDataFrame()
    result.add_column(group_col, StringColumn(_key_strings(rg.dtype, kc[0], kn[0])))
    result.add_column(alias, Float64Column(sums[0]))
    return result
