"""Arrow / Parquet ingest into the typed columns (pandrs_amd/io.py): layouts on
CPU, and on the GPU a groupby + join over a frame that came out of a Parquet file, against the oracle."""
import os

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")

from oracle import oracle as O
from pandrs_amd.frame import AggregateOp, GLOBAL_STRING_POOL
from pandrs_amd.io import from_arrow, read_parquet, to_arrow, write_parquet


def _table(n=1000, seed=0):
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, 37, n)
    return pa.table({
        "k": pa.array(ids, mask=rng.random(n) < 0.02),
        "name": pa.array(["cat_%02d" % i for i in ids % 11], mask=rng.random(n) < 0.03),
        "v": pa.array(rng.normal(10, 3, n), mask=rng.random(n) < 0.1),
        "small": pa.array(rng.integers(-5, 5, n).astype(np.int32)),
        "flag": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.05),
    })


def test_arrow_layouts_map_onto_reference_columns():
    t = _table()
    df = from_arrow(t)
    assert df.column_names == ["k", "name", "v", "small", "flag"] and df.row_count() == 1000
    assert [c.column_type() for c in df.columns] == ["Int64", "String", "Float64", "Int64", "Boolean"]
    for name in df.column_names:
        col, want = df.column(name), t.column(name).to_pylist()
        got = [col.get(i) for i in range(col.len())]
        assert got == want, name                                      # values and nulls, element by element
    k = df.column("k")
    nulls = np.array([x is None for x in t.column("k").to_pylist()])
    np.testing.assert_array_equal(k.null_mask, np.packbits(nulls, bitorder="little"))    # 1 = null, LSB first
    assert df.column("small").null_mask is None
    # equal strings share one pool code (string_pool.rs:28-53)
    name = df.column("name")
    codes = {GLOBAL_STRING_POOL.get(int(c)) for c, nl in zip(name.data, [name.is_null(i) for i in range(1000)]) if not nl}
    assert codes == {"cat_%02d" % i for i in range(11)}
    # sliced (offset) arrays and chunked columns
    sl = from_arrow(t.slice(13, 200))
    assert [sl.column("v").get(i) for i in range(200)] == t.column("v").to_pylist()[13:213]
    ch = from_arrow(pa.concat_tables([t.slice(0, 400), t.slice(400)]))
    assert [ch.column("flag").get(i) for i in range(1000)] == t.column("flag").to_pylist()
    back = to_arrow(df)
    assert back.column("name").to_pylist() == t.column("name").to_pylist()
    assert back.column("v").to_pylist() == t.column("v").to_pylist()
    with pytest.raises(TypeError):
        from_arrow(pa.table({"d": pa.array([1, 2], type=pa.date32())}))


@pytest.mark.gpu
def test_parquet_file_to_device_groupby_and_join(tmp_path):
    t = _table(n=200_000, seed=5)
    path = os.path.join(tmp_path, "t.parquet")
    write_parquet(from_arrow(t), path)
    df = read_parquet(path)
    res = df.group_by(["name"]).aggregate([("v", AggregateOp.Sum, "s"), ("v", AggregateOp.Median, "m"), ("k", AggregateOp.Count, "c")])
    name, v, k = df.column("name"), df.column("v"), df.column("k")
    wk, wn, wa = O.groupby_agg([name.view()], df.row_count(), [v.view(), k.view()], [(0, O.SUM), (0, O.MEDIAN), (1, O.COUNT)])
    want = {("NULL" if wn[0, g] else GLOBAL_STRING_POOL.get(int(wk[0, g]))): wa[:, g] for g in range(wk.shape[1])}
    assert res.row_count() == len(want)
    for i, key in enumerate(res.column("name").to_list()):
        np.testing.assert_allclose([res.column(c).data[i] for c in ("s", "m", "c")], want[key], rtol=1e-9)
    dims = from_arrow(pa.table({"name": ["cat_%02d" % i for i in range(0, 11, 2)], "w": np.arange(6, dtype=np.float64)}))
    j = df.inner_join(dims, "name", "name")
    li, ri = O.join_indices(name.view(), name.len(), dims.column("name").view(), 6, O.INNER)
    assert j.row_count() == len(li)
    np.testing.assert_array_equal(j.column("w").data, dims.column("w").data[ri])
