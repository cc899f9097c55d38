"""The legacy string-typed frame's groupby and merge on the device.

`src/dataframe/groupby.rs` groups a `DataFrame` of stringified cells on `Vec<String>` keys (:188-212) and, for
every aggregate of every group, re-parses the group's cells with `parse::<f64>()` (:444-463): cells that do
not parse are DROPPED, a group without parseable cells gives 0.0 (:465-467), and the results are written back
as strings (:283).  The same answers come from the typed engine without the per-row string work:

  * a key column's strings go through the global string pool once — equal string <=> equal code, which is all
    a `Vec<String>` key is compared on — and the key columns are grouped as u32 codes;
  * a value column is parsed ONCE (Rust's float grammar, not Python's) and its unparseable rows are compacted
    away, so one device groupby over the remaining rows gives every AggFunc its `group_values` semantics
    directly: Count is the number of parseable cells (:476), First / Last the first / last parseable cell
    (:511-512), Min / Max plain folds, Std / Var two-pass with n - 1, Median by sort, Nunique by sort + dedup;
  * groups that lost all their rows to the compaction are put back with 0.0 from the key set of all rows.

Differences kept on purpose: the order of the groups (HashMap order there, device order here), and a group
whose every value is +inf (-inf) has Min (Max) 0.0 — the optimized path's sentinel rule (aggregation.rs:656)
— where the legacy fold returns the infinity itself.  Custom closures run on the host over device-built groups.
"""
import re
from enum import IntEnum

import numpy as np

from . import _lib as L
from .frame import ColumnNotFound, GLOBAL_STRING_POOL, get_context, rust_f64_to_string


class AggFunc(IntEnum):              # src/dataframe/groupby.rs:29-43, same order
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
    Nunique = 10
    Custom = 11

    def as_str(self):                # :46-62
        return self.name.lower()


_ENGINE_OP = {AggFunc.Sum: L.SUM, AggFunc.Mean: L.MEAN, AggFunc.Min: L.MIN, AggFunc.Max: L.MAX, AggFunc.Count: L.COUNT,
              AggFunc.Std: L.STD, AggFunc.Var: L.VAR, AggFunc.Median: L.MEDIAN, AggFunc.First: L.FIRST,
              AggFunc.Last: L.LAST, AggFunc.Nunique: L.NUNIQUE}

# what `str::parse::<f64>` accepts: optional sign, then "inf" / "infinity" / "nan" in any case, or decimal digits
# with an optional point and exponent (at least one digit before the exponent); nothing else — no blanks,
# no "_" separators, no hex (Python's float() takes the first two)
# ASCII digits only and the WHOLE cell (Rust's parse rejects "1.5\n" and Arabic-Indic digits; `$` and `\d` would not)
_RUST_F64 = re.compile(r"[+-]?(?:inf|infinity|nan|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?)", re.IGNORECASE)


def parse_f64_cells(cells):
    """-> (values f64[n], parsed bool[n]) for a column of strings, cell by cell like `.parse::<f64>().ok()`."""
    n = len(cells)
    vals = np.zeros(n, np.float64)
    ok = np.zeros(n, bool)
    for i, s in enumerate(cells):
        if _RUST_F64.fullmatch(s):
            vals[i] = float(s)
            ok[i] = True
    return vals, ok


class InvalidValue(ValueError):      # Error::InvalidValue
    pass


class NamedAgg:                      # :68-107
    def __init__(self, column, func, alias, custom_fn=None):
        self.column, self.func, self.alias, self.custom_fn = column, AggFunc(int(func)), alias, custom_fn

    @classmethod
    def custom(cls, column, alias, func):
        return cls(column, AggFunc.Custom, alias, func)


class ColumnAggBuilder:              # :120-176
    def __init__(self, column):
        self.column = column
        self.aggregations = []

    def agg(self, func, alias):
        self.aggregations.append((AggFunc(int(func)), alias, None))
        return self

    def custom(self, alias, func):
        self.aggregations.append((AggFunc.Custom, alias, func))
        return self

    def build(self):
        return [NamedAgg(self.column, f, a, c) for f, a, c in self.aggregations]


class DataFrame:
    """The legacy frame as far as this path reads it: named columns of cells, every cell seen as the string
    `get_column_string_values` returns (src/dataframe/base.rs); numbers are stringified the way Rust does."""

    def __init__(self):
        self.column_names = []
        self._cells = {}

    @staticmethod
    def _cell(v):
        if isinstance(v, str):
            return v
        if isinstance(v, (bool, np.bool_)):
            return "true" if v else "false"
        if isinstance(v, (float, np.floating)):
            return rust_f64_to_string(float(v))
        return str(v)

    def add_column(self, name, values):
        if name in self._cells:
            raise ValueError("duplicate column name '%s'" % name)
        cells = [self._cell(v) for v in values]
        if self.column_names and len(cells) != self.row_count():
            raise ValueError("inconsistent row count")
        self.column_names.append(name)
        self._cells[name] = cells

    def contains_column(self, name):
        return name in self._cells

    def row_count(self):
        return len(self._cells[self.column_names[0]]) if self.column_names else 0

    def column_count(self):
        return len(self.column_names)

    def get_column_string_values(self, name):
        if name not in self._cells:
            raise ColumnNotFound(name)
        return self._cells[name]

    def groupby(self, columns):                      # GroupByExt::groupby (:606-617)
        return DataFrameGroupBy(self, list(columns))

    def groupby_single(self, column):                # :620-622
        return DataFrameGroupBy(self, [column])

    def to_optimized(self):
        """from_standard_dataframe (src/optimized/convert.rs:13-91): a column is Int64 if every non-empty cell parses
        as i64 (empty => 0), else Float64 if every non-empty cell parses as f64 (empty => 0.0), else Boolean for
        true / false / 1 / 0 in any case (empty => false), else String."""
        from .frame import BooleanColumn, Float64Column, Int64Column, OptimizedDataFrame, StringColumn
        out = OptimizedDataFrame()
        rust_i64 = re.compile(r"[+-]?[0-9]+")
        for name in self.column_names:
            cells = self._cells[name]

            def is_i64(s):
                return bool(rust_i64.fullmatch(s)) and -2**63 <= int(s) < 2**63
            if all(s == "" or is_i64(s) for s in cells):
                out.add_column(name, Int64Column([int(s) if s else 0 for s in cells]))
            elif all(s == "" or _RUST_F64.fullmatch(s) for s in cells):
                out.add_column(name, Float64Column([float(s) if s else 0.0 for s in cells]))
            elif all(s.lower() in ("", "true", "false", "1", "0") for s in cells):
                out.add_column(name, BooleanColumn([s.lower() in ("true", "1") for s in cells]))
            else:
                out.add_column(name, StringColumn(cells))
        return out


class DataFrameGroupBy:
    """DataFrameGroupBy (src/dataframe/groupby.rs:178-603)."""

    def __init__(self, df, group_by_columns):
        for c in group_by_columns:
            if not df.contains_column(c):
                raise ColumnNotFound(c)              # :185-190
        self.df = df
        self.group_by_columns = group_by_columns
        self._codes = None
        self._all = None
        self._groups = None

    # ---- device plumbing
    def _key_codes(self):
        if self._codes is None:
            self._codes = [np.fromiter((GLOBAL_STRING_POOL.get_or_insert(s) for s in self.df.get_column_string_values(c)),
                                       dtype=np.uint32, count=self.df.row_count()) for c in self.group_by_columns]
        return self._codes

    def _key_views(self, rows=None):
        return [((k if rows is None else k[rows]), None, L.U32CODE) for k in self._key_codes()]

    def _all_groups(self):
        """(key code tuples in result order, group sizes): one Count over every row."""
        if self._all is None:
            n = self.df.row_count()
            # (Count takes any column: the first key's codes stand in as the value column)
            kc, kn, oa = get_context().groupby_agg(self._key_views(), n, [self._key_views()[0]], [(0, L.COUNT)]) if n else \
                (np.zeros((len(self.group_by_columns), 0), np.uint64), None, np.zeros((1, 0)))
            keys = list(zip(*[kc[i].astype(np.uint32).tolist() for i in range(kc.shape[0])])) if kc.shape[1] else []
            self._all = (keys, oa[0].astype(np.int64))
        return self._all

    @property
    def groups(self):
        """HashMap<Vec<String>, Vec<usize>> (:199-211), from the device-built row lists."""
        if self._groups is None:
            n = self.df.row_count()
            self._groups = {}
            if n:
                cells, nulls, off, rows = get_context().groupby_indices(self._key_views(), n)
                for g in range(cells.shape[1]):
                    key = tuple(GLOBAL_STRING_POOL.get(int(cells[i, g])) for i in range(cells.shape[0]))
                    self._groups[key] = rows[off[g]:off[g + 1]].tolist()
        return self._groups

    def ngroups(self):                               # :220-222
        return len(self._all_groups()[0])

    def size(self):                                  # :225-247: "group" = the key joined with "_", "size"
        keys, sizes = self._all_groups()
        out = DataFrame()
        out.add_column("group", ["_".join(GLOBAL_STRING_POOL.get(c) for c in k) for k in keys])
        out.add_column("size", [str(int(s)) for s in sizes])
        return out

    # ---- aggregation
    def agg(self, named_aggs):                       # :250-292
        named_aggs = list(named_aggs)
        if not named_aggs:
            raise InvalidValue("At least one aggregation must be specified")
        for a in named_aggs:
            if not self.df.contains_column(a.column):
                raise ColumnNotFound(a.column)
            if a.func == AggFunc.Custom and a.custom_fn is None:
                raise InvalidValue("Custom function not provided")         # :520-528
        keys, _ = self._all_groups()
        pos = {k: i for i, k in enumerate(keys)}
        results = {}
        ctx = get_context()
        for column in dict.fromkeys(a.column for a in named_aggs):
            mine = [(j, a) for j, a in enumerate(named_aggs) if a.column == column]
            vals, ok = parse_f64_cells(self.df.get_column_string_values(column))
            dev = [(j, a) for j, a in mine if a.func != AggFunc.Custom]
            if dev:
                rows = np.flatnonzero(ok)                                   # the parseable cells, in row order
                out = np.zeros((len(dev), len(keys)), np.float64)           # no parseable cell => 0.0 (:465-467)
                if len(rows):
                    kc, kn, oa = ctx.groupby_agg(self._key_views(rows), len(rows), [(vals[rows], None, L.F64)],
                                                 [(0, _ENGINE_OP[a.func]) for _, a in dev])
                    at = np.fromiter((pos[k] for k in zip(*[kc[i].astype(np.uint32).tolist() for i in range(kc.shape[0])])),
                                     dtype=np.int64, count=kc.shape[1])
                    out[:, at] = oa
                for r, (j, _) in enumerate(dev):
                    results[j] = out[r]
            for j, a in mine:
                if a.func == AggFunc.Custom:                                # the closure sees the group's parseable values
                    col = np.zeros(len(keys), np.float64)
                    for key, idx in self.groups.items():
                        idx = np.asarray(idx, np.int64)
                        gv = vals[idx][ok[idx]]
                        code_key = tuple(GLOBAL_STRING_POOL.get_or_insert(s) for s in key)
                        col[pos[code_key]] = float(a.custom_fn(gv.tolist())) if len(gv) else 0.0
                    results[j] = col
        result = DataFrame()
        for i, name in enumerate(self.group_by_columns):                    # key columns first (:266-273)
            result.add_column(name, [GLOBAL_STRING_POOL.get(k[i]) for k in keys])
        for j, a in enumerate(named_aggs):                                  # then one column per alias, stringified (:276-289)
            result.add_column(a.alias, [rust_f64_to_string(float(x)) for x in results[j]])
        return result

    def agg_multi(self, builders):                   # :295-303
        return self.agg([a for b in builders for a in b.build()])

    def agg_dict(self, agg_spec):                    # :306-316
        return self.agg([NamedAgg(c, f, alias) for c, specs in agg_spec.items() for f, alias in specs])

    def _short(self, column, func):                  # :319-393: alias "{column}_{func}"
        return self.agg([NamedAgg(column, func, "%s_%s" % (column, func.as_str()))])

    def sum(self, column):
        return self._short(column, AggFunc.Sum)

    def mean(self, column):
        return self._short(column, AggFunc.Mean)

    def count(self, column):
        return self._short(column, AggFunc.Count)

    def min(self, column):
        return self._short(column, AggFunc.Min)

    def max(self, column):
        return self._short(column, AggFunc.Max)

    def std(self, column):
        return self._short(column, AggFunc.Std)

    def var(self, column):
        return self._short(column, AggFunc.Var)

    def median(self, column):
        return self._short(column, AggFunc.Median)

    def nunique(self, column):
        return self._short(column, AggFunc.Nunique)

    def apply(self, column, alias, func):            # :396-402
        return self.agg([NamedAgg.custom(column, alias, func)])

    def _subset(self, indices):                      # create_subset_dataframe (:554-576)
        out = DataFrame()
        for name in self.df.column_names:
            cells = self.df.get_column_string_values(name)
            out.add_column(name, [cells[i] for i in indices])
        return out

    def filter(self, condition):                     # :405-423: rows of the groups whose sub-frame passes, group by group
        keep = []
        for idx in self.groups.values():
            if condition(self._subset(idx)):
                keep.extend(idx)
        return self._subset(keep)

    def transform(self, func):                       # :426-440: per-group frames, concatenated after the first one's columns
        parts = [func(self._subset(idx)) for idx in self.groups.values()]
        out = DataFrame()
        if parts:
            for name in parts[0].column_names:
                out.add_column(name, [c for p in parts for c in p.get_column_string_values(name)])
        return out


# ------------------------------------------------------------------------------------------------- merge
class PandasGroupBy:
    """pandas_compat::DataFrameGroupBy (src/dataframe/pandas_compat/groupby.rs:11-455, `groupby_multi` :467-475) on the device.

    The reference groups the rows on the key cells joined with "|||" (:63) — so two key tuples whose joined strings
    coincide are ONE group, labelled with the tuple seen first (:68) — and folds every numeric column (a column whose
    every cell is a number, `get_column_numeric_values`) per group with NaN meaning "missing":
    sum of the non-NaN values (0.0 for none), mean / std / var NaN for none (n <= 1 for std / var, n - 1 divisor), min / max
    the fold's +inf / -inf for none, count = size = rows of the group, first / last the group's first / last ROW (NaN and
    string columns included), `agg` with "{col}_{agg}" columns where count is the number of non-NaN values and an unknown
    aggregate name gives NaN (:333-409).  Here: the joined strings become pool codes, every numeric column goes to the
    device with its NaN cells as nulls (one groupby for all columns and aggregates, plus a sum of 0/1 flags for the
    non-NaN counts), and the few cases where the engine's null rules differ from this file's NaN rules are patched from
    those counts.  Group order: the device's (the reference's is HashMap order)."""

    def __init__(self, df, by):
        by = list(by)
        if not by:
            raise InvalidValue("GroupBy requires at least one column")                         # :23-27
        for c in by:
            if not df.contains_column(c):
                raise InvalidValue("Column '%s' not found in DataFrame" % c)                 # :30-37
        self.df, self.group_columns = df, by
        n = df.row_count()
        cols = [df.get_column_string_values(c) for c in by]
        joined = ["|||".join(parts) for parts in zip(*cols)] if n else []
        self._codes = np.fromiter((GLOBAL_STRING_POOL.get_or_insert(s) for s in joined), dtype=np.uint32, count=n)
        self._labels = {}                                                                      # joined key -> parts of its first row
        for parts, j in zip(zip(*cols), joined):
            self._labels.setdefault(j, parts)
        self._rows = None

    def _key(self):
        return [(self._codes, None, L.U32CODE)]

    def _row_lists(self):
        if self._rows is None:
            n = self.df.row_count()
            if n:
                cells, _, off, rows = get_context().groupby_indices(self._key(), n)
                self._rows = (cells[0].astype(np.uint32), np.asarray(off), np.asarray(rows))
            else:
                self._rows = (np.zeros(0, np.uint32), np.zeros(1, np.int64), np.zeros(0, np.int64))
        return self._rows

    def ngroups(self):                                                                         # :83-85
        return len(self._labels)

    def _numeric_columns(self):
        out = []
        for c in self.df.column_names:
            if c not in self.group_columns:
                v = _numeric_column(self.df, c)
                if v is not None:
                    out.append((c, v))
        return out

    def _group_frame(self, codes):
        out = DataFrame()
        labels = [self._labels[GLOBAL_STRING_POOL.get(int(c))] for c in codes]
        for i, name in enumerate(self.group_columns):
            out.add_column(name, [lab[i] for lab in labels])
        return out

    def _fold(self, requests):
        """requests: [(values f64[n], op name)] -> (group codes, [f64[G] per request]) with this file's NaN rules."""
        n = self.df.row_count()
        if n == 0:
            return np.zeros(0, np.uint32), [np.zeros(0) for _ in requests]
        ops = {"sum": L.SUM, "mean": L.MEAN, "min": L.MIN, "max": L.MAX, "std": L.STD, "var": L.VAR}
        cols, col_of, aggs, slots = [], {}, [], []
        for vals, op in requests:
            if id(vals) not in col_of:
                nan = np.isnan(vals)
                col_of[id(vals)] = len(cols)
                cols.append((np.where(nan, 0.0, vals), np.packbits(nan, bitorder="little") if nan.any() else None, L.F64))
                cols.append(((~nan).astype(np.float64), None, L.F64))                          # non-NaN flags: their sum is the count
            c = col_of[id(vals)]
            slots.append((len(aggs) if op in ops else -1, len(aggs) + (1 if op in ops else 0)))
            if op in ops:
                aggs.append((c, ops[op]))
            aggs.append((c + 1, L.SUM))
        kc, _, oa = get_context().groupby_agg(self._key(), n, cols, aggs)
        out = []
        for (vals, op), (a, cnt_slot) in zip(requests, slots):
            nn = oa[cnt_slot]
            if op == "count":
                r = nn.copy()
            elif op == "sum":
                r = oa[a].copy()
            elif op == "mean":
                r = np.where(nn > 0, oa[a], np.nan)
            elif op == "min":
                r = np.where(nn > 0, oa[a], np.inf)
            elif op == "max":
                r = np.where(nn > 0, oa[a], -np.inf)
            elif op in ("std", "var"):
                r = np.where(nn > 1, oa[a], np.nan)
            else:
                r = np.full(kc.shape[1], np.nan)                                               # unknown aggregate name (:405)
            out.append(r)
        return kc[0].astype(np.uint32), out

    def _aggregate(self, op):                                                                  # aggregate (:203-262): "{col}" columns
        numeric = self._numeric_columns()
        if not numeric:
            codes, _ = self._sizes()
            return self._group_frame(codes)
        codes, res = self._fold([(v, op) for _, v in numeric])
        out = self._group_frame(codes)
        for (name, _), r in zip(numeric, res):
            out.add_column(name, [float(x) for x in r])
        return out

    def _sizes(self):
        n = self.df.row_count()
        if n == 0:
            return np.zeros(0, np.uint32), np.zeros(0)
        kc, _, oa = get_context().groupby_agg(self._key(), n, [self._key()[0]], [(0, L.COUNT)])
        return kc[0].astype(np.uint32), oa[0]

    def size(self):                                                                            # :88-117
        codes, sizes = self._sizes()
        out = self._group_frame(codes)
        out.add_column("size", [float(x) for x in sizes])
        return out

    def count(self):                                                                           # :120-122
        return self.size()

    def sum(self):
        return self._aggregate("sum")

    def mean(self):
        return self._aggregate("mean")

    def min(self):
        return self._aggregate("min")

    def max(self):
        return self._aggregate("max")

    def std(self):
        return self._aggregate("std")

    def var(self):
        return self._aggregate("var")

    def _first_last(self, first):                                                              # aggregate_first_last (:264-330)
        codes, off, rows = self._row_lists()
        pick = rows[off[:-1]] if first else rows[off[1:] - 1]
        out = self._group_frame(codes)
        for c in self.df.column_names:
            if c not in self.group_columns:
                cells = self.df.get_column_string_values(c)
                num = _numeric_column(self.df, c)
                out.add_column(c, [float(num[i]) for i in pick] if num is not None else [cells[i] for i in pick])
        return out

    def first(self):
        return self._first_last(True)

    def last(self):
        return self._first_last(False)

    def agg(self, aggs):                                                                       # :333-455
        aggs = [(c, a) for c, a in aggs if self.df.contains_column(c)]
        numeric = {c: _numeric_column(self.df, c) for c, _ in aggs}
        todo = [(c, a) for c, a in aggs if numeric[c] is not None]
        fl = [(c, a) for c, a in todo if a in ("first", "last")]
        fold = [(c, a) for c, a in todo if a not in ("first", "last")]
        codes, res = self._fold([(numeric[c], a) for c, a in fold]) if fold else (self._sizes()[0], [])
        by_name = {"%s_%s" % (c, a): r for (c, a), r in zip(fold, res)}
        if fl:
            gcodes, off, rows = self._row_lists()
            order = {int(k): i for i, k in enumerate(gcodes)}
            at = np.array([order[int(k)] for k in codes], np.int64)
            for c, a in fl:
                pick = rows[off[:-1]] if a == "first" else rows[off[1:] - 1]
                by_name["%s_%s" % (c, a)] = numeric[c][pick][at]
        out = self._group_frame(codes)
        for c, a in todo:
            name = "%s_%s" % (c, a)
            if not out.contains_column(name):
                out.add_column(name, [float(x) for x in by_name[name]])
        return out


def groupby_multi(df, by):
    """PandasGroupByExt::groupby_multi (src/dataframe/pandas_compat/groupby.rs:467-475)."""
    return PandasGroupBy(df, by)


class JoinType(IntEnum):             # src/dataframe/pandas_compat/merge.rs:11-20 (same order as the engine's)
    Inner = 0
    Left = 1
    Right = 2
    Outer = 3


def _numeric_column(df, name):
    """get_column_numeric_values (src/dataframe/base.rs:542-…): the column as f64, or None unless EVERY cell parses."""
    vals, ok = parse_f64_cells(df.get_column_string_values(name))
    return vals if ok.all() else None


def merge(left, right, on, how, suffixes=("_x", "_y")):
    """pandas_compat::merge (src/dataframe/pandas_compat/merge.rs:34-265) with the match pairs from
    pandrs_hip_join_indices: the build side is the right frame, left rows in order, a left row's matches in right
    order, unmatched left rows for Left / Outer, unmatched right rows appended for Right / Outer (:70-118) — the
    order contract of the optimized join, and these frames hold no nulls.  Numeric join columns compare by
    `f64::to_bits` (:56-59): the bit patterns are joined as i64 cells, so -0.0 != 0.0 and NaN payloads count;
    anything else compares as strings (pool codes).  A missing side is NaN in numeric columns and "" in string
    columns (:150, :172); overlapping non-key columns get the suffixes, left then right (:128-133, :236-262)."""
    for side, df in (("left", left), ("right", right)):
        if not df.contains_column(on):
            raise InvalidValue("Join column '%s' not found in %s DataFrame" % (on, side))
    ln, rn = _numeric_column(left, on), _numeric_column(right, on)
    if ln is not None and rn is not None:
        lk, rk = (ln.view(np.int64), None, L.I64), (rn.view(np.int64), None, L.I64)
    else:                                            # the reference's join strings: the bits as decimal text, or the cell
        def join_strings(df, num):
            return [str(int(b)) for b in num.view(np.uint64)] if num is not None else df.get_column_string_values(on)

        def codes(strs):
            return (np.fromiter((GLOBAL_STRING_POOL.get_or_insert(s) for s in strs), dtype=np.uint32, count=len(strs)), None, L.U32CODE)
        lk, rk = codes(join_strings(left, ln)), codes(join_strings(right, rn))
    li, ri = get_context().join_indices(lk, left.row_count(), rk, right.row_count(), int(how))
    li, ri = np.asarray(li), np.asarray(ri)

    def take(df, name, idx, other=None, other_idx=None):
        num = _numeric_column(df, name)
        miss = idx < 0
        if num is not None:
            out = np.where(miss, np.nan, num[np.where(miss, 0, idx)]) if len(num) else np.full(len(idx), np.nan)
            if other is not None:                     # the join key: the right value where the left row is missing
                onum = _numeric_column(other, name)
                if onum is not None and len(onum):
                    out = np.where(miss & (other_idx >= 0), onum[np.where(other_idx < 0, 0, other_idx)], out)
            return [rust_f64_to_string(float(x)) for x in out]
        cells = df.get_column_string_values(name)
        out = ["" if i < 0 else cells[i] for i in idx.tolist()]
        if other is not None:
            ocells = other.get_column_string_values(name)
            out = [ocells[j] if (i < 0 <= j) else c for c, i, j in zip(out, idx.tolist(), other_idx.tolist())]
        return out

    overlapping = [c for c in right.column_names if c != on and c in left.column_names]
    result = DataFrame()
    for name in left.column_names:
        new = name + suffixes[0] if name in overlapping else name
        result.add_column(new, take(left, name, li, right, ri) if name == on else take(left, name, li))
    for name in right.column_names:
        if name == on:
            continue
        result.add_column(name + suffixes[1] if name in overlapping else name, take(right, name, ri))
    return result
