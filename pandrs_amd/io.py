"""Arrow / Parquet ingest straight into the typed columns the engine consumes.

The reference's Arrow bridge (src/arrow_integration.rs:80-103, :170-275) and its Parquet reader
(src/io/parquet.rs:175-367) go through the legacy string-typed DataFrame: every value is formatted
to a String and re-parsed later.  Here a RecordBatch column becomes the column layout of
src/column/* directly — the layouts already agree almost bit for bit:

  Arrow Int64 / Float64   values buffer           == Int64Column.data / Float64Column.data (Arc<[T]>)
  Arrow Boolean           LSB-first packed bits   == BooleanColumn.data (BitMask)
  Arrow validity bitmap   LSB-first, 1 = VALID    == the reference's null mask inverted (1 = null,
                                                     src/core/column.rs:163-177)
  Arrow Utf8 / dictionary dictionary-encoded, then the dictionary (not the rows) goes through
                          GLOBAL_STRING_POOL.get_or_insert (string_pool.rs:28-53): codes by table lookup

so a Parquet file reaches pandrs_hip_groupby_agg / pandrs_hip_join_indices without a stringify round trip.
pyarrow is a host-side dependency of this module only."""
import numpy as np

from .frame import (BooleanColumn, Float64Column, GLOBAL_STRING_POOL, Int64Column, OptimizedDataFrame,
                    StringColumn)


def _null_mask(arr):
    """Arrow validity bitmap (1 = valid, bit `offset` first) -> reference null bitmap (1 = null) or None."""
    if arr.null_count == 0:
        return None
    valid = np.asarray(arr.is_valid())          # handles slices / offsets
    return np.packbits(~valid, bitorder="little")


def _values(arr, np_dtype):
    """The values buffer of a primitive array as numpy (zero-copy when the array is not sliced)."""
    import pyarrow as pa
    buf = arr.buffers()[1]
    out = np.frombuffer(buf, dtype=np_dtype, count=arr.offset + len(arr))[arr.offset:]
    return out


def column_from_arrow(arr):
    """One pyarrow Array / ChunkedArray -> Int64Column / Float64Column / StringColumn / BooleanColumn."""
    import pyarrow as pa
    import pyarrow.compute as pc
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
    t = arr.type
    if pa.types.is_integer(t):
        if not pa.types.is_int64(t):
            arr = arr.cast(pa.int64())
        col = Int64Column(_values(arr, np.int64))
    elif pa.types.is_floating(t):
        if not pa.types.is_float64(t):
            arr = arr.cast(pa.float64())
        col = Float64Column(_values(arr, np.float64))
    elif pa.types.is_boolean(t):
        bits = np.packbits(np.asarray(arr.fill_null(False)), bitorder="little")
        col = BooleanColumn(None, _bits=bits, _length=len(arr))
    elif pa.types.is_string(t) or pa.types.is_large_string(t) or pa.types.is_dictionary(t):
        d = arr if pa.types.is_dictionary(t) else pc.dictionary_encode(arr)
        pool_codes = np.fromiter((GLOBAL_STRING_POOL.get_or_insert(s) for s in d.dictionary.to_pylist()),
                                 dtype=np.uint32, count=len(d.dictionary))
        idx = np.asarray(d.indices.fill_null(0)).astype(np.int64)
        # null rows hold the code of "" like StringColumn::with_nulls's placeholder; the mask decides
        codes = pool_codes[idx] if len(pool_codes) else np.full(len(arr), GLOBAL_STRING_POOL.get_or_insert(""), np.uint32)
        col = StringColumn.from_codes(codes)
    else:
        raise TypeError("Arrow type %s has no typed column in the reference (src/column/mod.rs)" % t)
    col.null_mask = _null_mask(arr)
    return col


def from_arrow(table):
    """pyarrow Table / RecordBatch -> OptimizedDataFrame (column order kept, src/arrow_integration.rs:93-101)."""
    df = OptimizedDataFrame()
    for name, col in zip(table.schema.names, table.columns):
        df.add_column(name, column_from_arrow(col))
    return df


def to_arrow(df):
    """OptimizedDataFrame -> pyarrow Table (nulls restored from the masks)."""
    import pyarrow as pa
    arrays = []
    for col in df.columns:
        n = col.len()
        nulls = None
        if col.null_mask is not None:
            nulls = np.unpackbits(col.null_mask, bitorder="little")[:n].astype(bool)
        if isinstance(col, StringColumn):
            vals = [GLOBAL_STRING_POOL.get(int(c)) for c in col.data]
            arrays.append(pa.array(vals, type=pa.string(), mask=nulls))
        elif isinstance(col, BooleanColumn):
            arrays.append(pa.array(np.unpackbits(col.data, bitorder="little")[:n].astype(bool), mask=nulls))
        else:
            arrays.append(pa.array(col.data, mask=nulls))
    return pa.Table.from_arrays(arrays, names=list(df.column_names))


def read_parquet(path, columns=None):
    """Parquet file -> OptimizedDataFrame (reference: read_parquet, src/io/parquet.rs:175)."""
    import pyarrow.parquet as pq
    return from_arrow(pq.read_table(path, columns=columns))


def write_parquet(df, path):
    """reference: write_parquet, src/io/parquet.rs:369."""
    import pyarrow.parquet as pq
    pq.write_table(to_arrow(df), path)
