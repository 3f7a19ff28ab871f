"""Page-index pruning in the Parquet scan (the reference turns on DataFusion's page-index pruning in parquet_exec.rs): the rows a
pushed-down filter can still be true for are worked out from ColumnIndex / OffsetIndex, pages outside them are never decompressed, pages
partly inside them are decoded from the first kept row on — for every column, whatever its own page boundaries — and the scan emits only
the kept rows, in file order.  Checked against pyarrow's reader: the scan's output is exactly the rows of the pages that survive, every
row the filter accepts is among them, Filter(scan) equals the filtered table, and with the index off nothing changes but the row count."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S
from tests.test_parquet_gpu import _types

pytestmark = pytest.mark.gpu


def _table(n, seed):
    rng = np.random.default_rng(seed)
    k = np.sort(rng.integers(0, 1_000_000, n))                 # the filter column: sorted, so its pages have disjoint ranges
    m = lambda p: rng.random(n) < p
    words = np.array(["", "a", "lineitem", "MI355X", "x" * 30], dtype=object)
    return pa.table({
        "k": pa.array(k, pa.int64()),
        "kn": pa.array(k, pa.int64(), mask=m(0.2)),            # same values with NULLs: value indices differ from row indices
        "v": pa.array(rng.integers(-10**9, 10**9, n), pa.int64(), mask=m(0.1)),
        "f": pa.array(rng.standard_normal(n)),                 # PLAIN doubles (dictionary overflows): other page boundaries than k's
        "low": pa.array(rng.integers(0, 7, n), pa.int32(), mask=m(0.3)),
        "s": pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=m(0.15)),
        "b": pa.array(rng.random(n) < 0.5, pa.bool_(), mask=m(0.1)),
        "ps": pa.array(["row%07d" % i for i in range(n)], pa.utf8()),      # PLAIN strings (unique values)
    })


def _run(path, table, filters, config=None, with_filter=None):
    plan = S.native_scan([path], table.schema.names, _types(table.schema), data_filters=filters)
    if with_filter is not None:
        plan = S.filter_(plan, with_filter)
    it = native.CometExecIterator([], table.num_columns, plan.encode(), batch_size=0, config=S.config_map(config or {}))
    batches = []
    while True:
        b = native.Native.executePlan(it.handle, table.num_columns)
        if b is None:
            break
        batches.append(b)
    m = S.decode_metric_node(it.metrics())
    it.close()
    while m[1]:
        m = m[1][0]
    out = pa.Table.from_batches(batches) if batches else table.schema.empty_table()
    return out, m[0]


@pytest.mark.parametrize("version,codec,dictionary", [("1.0", "snappy", True), ("2.0", "zstd", False), ("1.0", "none", True)])
def test_pages_ruled_out_by_the_index_are_not_decoded(built, tmp_path, version, codec, dictionary):
    n = 400_000
    t = _table(n, 21)
    path = str(tmp_path / "indexed.parquet")
    papq.write_table(t, path, row_group_size=150_000, data_page_size=32 << 10, write_page_index=True, data_page_version=version,
                     compression=None if codec == "none" else codec, use_dictionary=dictionary)
    I64 = S.T_INT64
    k, kn = S.col(0, I64), S.col(1, I64)
    lo, hi = 300_000, 420_000
    pred = S.and_(S.gt_eq(k, S.lit(lo, I64)), S.lt(k, S.lit(hi, I64)))
    filters = [S.gt_eq(k, S.lit(lo, I64)), S.lt(k, S.lit(hi, I64))]
    cfg = {"spark.comet.gpu.scan.deviceDecompress": "true"}
    want = t.filter(pc.and_(pc.greater_equal(t.column("k"), lo), pc.less(t.column("k"), hi)))
    # the scan alone: a superset of the qualifying rows, a sub-sequence of the file, far fewer rows than the file
    got, m = _run(path, t, filters, cfg)
    assert m["page_index_rows_pruned"] > 0 and got.num_rows < n // 3
    ks = np.asarray(got.column(0))
    assert ks.min() <= lo and ks.max() >= hi - 1
    # kept rows are whole runs of the file: compare with the same slice(s) of the source table, column by column
    pos = np.array([int(x[3:]) for x in got.column(7).to_pylist()])          # "row%07d": the file row each kept row came from
    assert len(pos) == got.num_rows and (np.diff(pos) > 0).all()
    src = t.take(pa.array(pos))
    for i, name in enumerate(t.schema.names):
        assert got.column(i).combine_chunks().equals(src.column(name).combine_chunks()), name
    # Filter above the scan: exactly the filtered table
    got_f, _ = _run(path, t, filters, cfg, with_filter=pred)
    for i, name in enumerate(t.schema.names):
        assert got_f.column(i).combine_chunks().equals(want.column(name).combine_chunks()), name
    # the nullable twin of the column: IsNotNull + range; pages of kn hold NULLs, so their value indices are not their row indices
    filters2 = [S.is_not_null(kn), S.gt(kn, S.lit(900_000, I64))]
    pred2 = S.and_(S.is_not_null(kn), S.gt(kn, S.lit(900_000, I64)))
    got2, m2 = _run(path, t, filters2, cfg, with_filter=pred2)
    want2 = t.filter(pc.greater(t.column("kn"), 900_000))
    assert m2["page_index_rows_pruned"] > 0
    for i, name in enumerate(t.schema.names):
        assert got2.column(i).combine_chunks().equals(want2.column(name).combine_chunks()), name
    # index off: same answer, no pruning below the row-group level
    got_off, m_off = _run(path, t, filters, dict(cfg, **{"spark.comet.gpu.scan.pageIndex": "false"}), with_filter=pred)
    assert m_off["page_index_rows_pruned"] == 0
    for i, name in enumerate(t.schema.names):
        assert got_off.column(i).combine_chunks().equals(want.column(name).combine_chunks()), name


def test_or_of_ranges_and_files_without_an_index(built, tmp_path):
    n = 200_000
    t = _table(n, 22)
    I64 = S.T_INT64
    k = S.col(0, I64)
    either = S.or_(S.lt(k, S.lit(50_000, I64)), S.gt(k, S.lit(950_000, I64)))
    want = t.filter(pc.or_(pc.less(t.column("k"), 50_000), pc.greater(t.column("k"), 950_000)))
    for name, index in (("with.parquet", True), ("without.parquet", False)):
        path = str(tmp_path / name)
        papq.write_table(t, path, row_group_size=n, data_page_size=16 << 10, write_page_index=index)
        got, m = _run(path, t, [either], with_filter=either)
        assert (m["page_index_rows_pruned"] > n // 2) == index
        for i, nm in enumerate(t.schema.names):
            assert got.column(i).combine_chunks().equals(want.column(nm).combine_chunks()), nm
