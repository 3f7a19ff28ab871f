"""GPU parity for the Parquet scan: footer + page walk on the host, levels / dictionary
indices / values decoded on the device, against pyarrow (parquet-cpp) as the independent decoder.
Files are written here with pyarrow: PLAIN and RLE_DICTIONARY, uncompressed / snappy / zstd, NULLs, many pages,
several row groups, byte-range splits, decimals as FLBA and as INT64 (how Spark writes decimal(12,2))."""
import os
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu


def _types(schema: pa.Schema):
    out = []
    for f in schema:
        t = f.type
        if pa.types.is_decimal(t):
            out.append(S.decimal(t.precision, t.scale))
        else:
            out.append({pa.int32(): S.T_INT32, pa.int64(): S.T_INT64, pa.float64(): S.T_DOUBLE, pa.float32(): S.T_FLOAT, pa.date32(): S.T_DATE,
                        pa.utf8(): S.T_STRING, pa.bool_(): S.T_BOOL, pa.int16(): S.T_INT16, pa.int8(): S.T_INT8}[t])
    return out


def _mixed_table(n, seed, nulls=True):
    rng = np.random.default_rng(seed)
    m = (lambda: rng.random(n) < 0.15) if nulls else (lambda: None)
    words = np.array(["", "a", "bb", "lineitem", "MI355X", "naïve", "x" * 40], dtype=object)
    return pa.table({
        "i32": pa.array(rng.integers(-2**31, 2**31 - 1, n), pa.int32(), mask=m()),
        "i64": pa.array(rng.integers(-2**62, 2**62, n), pa.int64(), mask=m()),
        "f64": pa.array(rng.standard_normal(n), pa.float64(), mask=m()),
        "f32": pa.array(rng.standard_normal(n).astype(np.float32), pa.float32(), mask=m()),
        "d": pa.array(rng.integers(8000, 11000, n), pa.int32(), mask=m()).cast(pa.date32()),
        "dec": pa.array([Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**11, 10**11, n)], pa.decimal128(12, 2), mask=m()),
        "dec38": pa.array([Decimal(int(v) * 10**15).scaleb(-6) for v in rng.integers(-10**18, 10**18, n)], pa.decimal128(38, 6), mask=m()),
        "lowcard": pa.array(rng.integers(0, 5, n), pa.int64()),                      # dictionary, narrow bit width, no nulls
        "s": pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=m()),
        "b": pa.array(rng.random(n) < 0.5, pa.bool_(), mask=m()),
    })


def _scan(path_or_splits, table, **kw):
    files = path_or_splits if isinstance(path_or_splits, list) else [path_or_splits]
    plan = S.native_scan(files, table.schema.names, _types(table.schema))
    out = native.execute_to_table([], table.num_columns, plan.encode(), batch_size=0, **kw)
    return pa.Table.from_batches(out) if out else None


def _assert_same(got: pa.Table, want: pa.Table):
    assert got.num_rows == want.num_rows
    for i, name in enumerate(want.schema.names):
        g, w = got.column(i).combine_chunks(), want.column(name).combine_chunks()
        assert g.type == w.type, name
        assert g.equals(w), f"column {name} differs"


@pytest.mark.parametrize("compression", ["NONE", "SNAPPY", "ZSTD", "LZ4", "GZIP"])
@pytest.mark.parametrize("use_dictionary", [True, False])
def test_roundtrip_all_types(built, tmp_path, compression, use_dictionary):
    t = _mixed_table(30_000, seed=1)
    path = str(tmp_path / "t.parquet")
    papq.write_table(t, path, compression=compression, use_dictionary=use_dictionary, data_page_size=16 * 1024, row_group_size=9_000)
    _assert_same(_scan(path, t), papq.read_table(path))


@pytest.mark.parametrize("version,codec", [("2.0", "snappy"), ("1.0", "zstd")])
def test_delta_and_byte_stream_split_encodings(built, tmp_path, version, codec):
    """what data-page-v2 writers emit instead of PLAIN: DELTA_BINARY_PACKED ints / dates, DELTA_LENGTH_BYTE_ARRAY and prefix-compressed DELTA_BYTE_ARRAY strings, BYTE_STREAM_SPLIT
    floats (rewritten as PLAIN on the host, tests/test_parquet_encodings_cpu.py), next to dictionary columns, with NULLs and several row groups"""
    t = _mixed_table(40_000, 23)
    rng = np.random.default_rng(5)
    t = t.append_column("p", pa.array(sorted("Customer#%09d/BUILDING" % int(k) for k in rng.integers(0, 10**6, t.num_rows)), mask=rng.random(t.num_rows) < 0.1))
    path = str(tmp_path / "v2.parquet")
    enc = {"i32": "DELTA_BINARY_PACKED", "i64": "DELTA_BINARY_PACKED", "d": "DELTA_BINARY_PACKED", "s": "DELTA_LENGTH_BYTE_ARRAY", "p": "DELTA_BYTE_ARRAY", "f64": "BYTE_STREAM_SPLIT",
           "f32": "BYTE_STREAM_SPLIT"}
    papq.write_table(t, path, use_dictionary=["lowcard"], column_encoding=enc, data_page_version=version, compression=codec, row_group_size=15_000, data_page_size=1 << 13)
    _assert_same(_scan(path, t), t)


def test_required_columns_and_single_page(built, tmp_path):
    t = _mixed_table(2_000, seed=2, nulls=False)
    schema = pa.schema([pa.field(f.name, f.type, nullable=False) for f in t.schema])
    t = t.cast(schema)
    path = str(tmp_path / "req.parquet")
    papq.write_table(t, path, compression="NONE")
    _assert_same(_scan(path, t), papq.read_table(path))


def test_byte_range_splits_partition_row_groups_exactly_once(built, tmp_path):
    # Spark file splits: a row group belongs to the split that contains its midpoint
    t = _mixed_table(40_000, seed=3).select(["i64", "dec", "s"])
    path = str(tmp_path / "split.parquet")
    papq.write_table(t, path, compression="SNAPPY", row_group_size=5_000)
    size = os.path.getsize(path)
    cuts = [0, size // 3, 2 * size // 3, size]
    parts = [_scan([(path, cuts[k], cuts[k + 1] - cuts[k], size)], t) for k in range(3)]
    parts = [p for p in parts if p is not None]
    assert sum(p.num_rows for p in parts) == t.num_rows
    _assert_same(pa.concat_tables(parts), papq.read_table(path))


def test_spark_style_decimal_as_int64_and_q6_over_parquet(built, tmp_path):
    from oracle import oracle as O
    table = tpch.lineitem_q6(200_000, seed=14)
    path = str(tmp_path / "lineitem.parquet")
    try:
        papq.write_table(table, path, compression="ZSTD", row_group_size=50_000, store_decimal_as_integer=True)
    except TypeError:
        papq.write_table(table, path, compression="ZSTD", row_group_size=50_000)
    DEC = S.decimal(12, 2)
    scan = S.native_scan([path], table.schema.names, [DEC, DEC, DEC, S.T_DATE])
    plan = tpch.q6_plan()
    # replace the Scan leaf of Q6 by the Parquet scan
    node = plan
    while node.children[0].kind != "scan":
        node = node.children[0]
    node.children[0] = scan
    got = pa.Table.from_batches(native.execute_to_table([], 2, plan.encode()))
    want = O.run_plan_to_arrow(S, tpch.q6_plan(), table)
    assert got.column(0).to_pylist() == want.column(0).to_pylist()
    # and through the oracle's own parquet leaf (pyarrow decoder)
    want2 = O.run_plan_to_arrow(S, plan, None)
    assert got.column(0).to_pylist() == want2.column(0).to_pylist()


def test_empty_file_list_and_missing_column(built, tmp_path):
    t = _mixed_table(10, seed=5).select(["i32"])
    plan = S.native_scan([], ["i32"], [S.T_INT32])
    assert native.execute_to_table([], 1, plan.encode()) == []
    path = str(tmp_path / "m.parquet")
    papq.write_table(t, path)
    # a column the file does not have reads as NULL (schema evolution), like the reference's schema adapter
    missing = S.native_scan([path], ["nope", "i32"], [S.T_INT32, S.T_INT32])
    got = pa.Table.from_batches(native.execute_to_table([], 2, missing.encode(), batch_size=0))
    assert got.column(0).null_count == 10 and got.column(1).equals(t.column("i32"))


def test_schema_adaptation(built, tmp_path):
    """Per-file schema reconciliation (parquet/schema_adapter.rs, parquet_support.rs:141-240): widening promotions,
    decimal precision/scale widening, legacy INT96 timestamps, a column the file does not have (→ NULL), case-insensitive names."""
    import datetime
    n = 40_000
    rng = np.random.default_rng(8)
    ts = pa.array(rng.integers(0, 2 * 10**15, n), pa.int64(), mask=rng.random(n) < 0.1).cast(pa.timestamp("us"))
    t = pa.table({
        "I": pa.array(rng.integers(-2**31, 2**31 - 1, n), pa.int32(), mask=rng.random(n) < 0.1),
        "f": pa.array(rng.standard_normal(n).astype(np.float32), pa.float32(), mask=rng.random(n) < 0.1),
        "d": pa.array([Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**8, 10**8, n)], pa.decimal128(9, 2), mask=rng.random(n) < 0.1),
        "ts": ts,
    })
    path = str(tmp_path / "adapt.parquet")
    papq.write_table(t, path, row_group_size=15_000, use_deprecated_int96_timestamps=True, store_decimal_as_integer=True)
    assert papq.ParquetFile(path).schema.column(3).physical_type == "INT96"
    names = ["i", "F", "i", "d", "TS", "not_in_file", "missing_str"]       # read twice with different promotions; names differ in case
    types = [S.T_INT64, S.T_DOUBLE, S.T_DOUBLE, S.decimal(20, 6), S.DataType(S.TIMESTAMP_NTZ), S.T_INT32, S.T_STRING]
    plan = S.native_scan([path], names, types, case_sensitive=False)
    got = pa.Table.from_batches(native.execute_to_table([], len(names), plan.encode(), batch_size=0))
    want = [t.column("I").cast(pa.int64()), t.column("f").cast(pa.float64()), t.column("I").cast(pa.float64()), t.column("d").cast(pa.decimal128(20, 6)),
            t.column("ts")]
    for i, w in enumerate(want):
        g = got.column(i).combine_chunks()
        assert g.type == w.type or (i == 4 and pa.types.is_timestamp(g.type)), (i, g.type, w.type)
        assert g.cast(w.type).equals(w.combine_chunks()), f"column {names[i]}"
    assert got.column(5).null_count == n and got.column(6).null_count == n and got.column(5).type == pa.int32()
    # a case-sensitive scan does not find "i" / "TS": they read as NULL columns
    plan_cs = S.native_scan([path], ["i", "f"], [S.T_INT64, S.T_DOUBLE], case_sensitive=True)
    got_cs = pa.Table.from_batches(native.execute_to_table([], 2, plan_cs.encode(), batch_size=0))
    assert got_cs.column(0).null_count == n and got_cs.column(1).null_count == t.column("f").null_count
    # narrowing is refused
    with pytest.raises(native.CometNativeException, match="without losing digits"):
        native.execute_to_table([], 1, S.native_scan([path], ["d"], [S.decimal(9, 1)]).encode())


def test_row_group_pruning_from_statistics(built, tmp_path):
    """data_filters pushed into the scan prune row groups by min/max statistics (parquet_exec.rs:60-211 → ParquetSource): same
    answer as without them, fewer bytes scanned; the Filter above still decides every row."""
    n = 200_000
    rng = np.random.default_rng(12)
    ship = np.sort(rng.integers(tpch.days(1992, 1, 1), tpch.days(1998, 12, 1), n)).astype(np.int32)
    t = pa.table({"d": pa.array(ship, pa.int32()).cast(pa.date32()), "q": tpch._dec128_array(rng.integers(100, 5100, n), 12, 2),
                  "k": pa.array(rng.integers(0, 1000, n), pa.int64(), mask=rng.random(n) < 0.1)})
    path = str(tmp_path / "sorted.parquet")
    papq.write_table(t, path, row_group_size=10_000, store_decimal_as_integer=True)
    D = S.decimal(12, 2)
    types = [S.T_DATE, D, S.T_INT64]
    d, q, k = S.col(0, S.T_DATE), S.col(1, D), S.col(2, S.T_INT64)
    pred = S.and_(S.and_(S.gt_eq(d, S.lit(tpch.days(1994, 1, 1), S.T_DATE)), S.lt(d, S.lit(tpch.days(1995, 1, 1), S.T_DATE))), S.lt(q, S.lit(2400, D)))

    def run(filters):
        plan = S.filter_(S.native_scan([path], t.schema.names, types, data_filters=filters), pred)
        it = native.CometExecIterator([], 3, plan.encode(), batch_size=0)
        out = pa.Table.from_batches(list(it_all(it)))
        metrics = S.decode_metric_node(it.metrics())
        it.close()
        return out, metrics

    def it_all(it):
        while True:
            b = native.Native.executePlan(it.handle, 3)
            if b is None:
                return
            yield b

    def scan_metrics(m):     # (metrics dict, [children]); the NativeScan is the leaf
        node = m
        while node[1]:
            node = node[1][0]
        return node[0]

    full, m_full = run([])
    pruned, m_pruned = run([S.gt_eq(d, S.lit(tpch.days(1994, 1, 1), S.T_DATE)), S.lt(d, S.lit(tpch.days(1995, 1, 1), S.T_DATE)), S.is_not_null(k)])
    assert pruned.equals(full) and full.num_rows > 1000
    assert scan_metrics(m_pruned)["row_groups_pruned_statistics"] >= 15 and scan_metrics(m_full)["row_groups_pruned_statistics"] == 0
    assert scan_metrics(m_pruned)["bytes_scanned"] < scan_metrics(m_full)["bytes_scanned"] // 4
    # a filter that excludes everything prunes every row group: empty result
    nothing = S.native_scan([path], t.schema.names, types, data_filters=[S.gt(S.lit(tpch.days(1990, 1, 1), S.T_DATE), d)])
    assert native.execute_to_table([], 3, nothing.encode()) == []


def test_row_group_pruning_from_bloom_filters(built, tmp_path):
    """`column = literal` / `column IN (...)` pushed into the scan rule row groups out by the chunks' Bloom filters where min / max cannot
    (datafusion.execution.parquet.bloom_filter_on_read, parquet_exec.rs:251-252; the decision itself: tests/test_parquet_bloom_cpu.py): same answer as
    without them, fewer bytes scanned, counted under DataFusion's metric name"""
    import pyarrow.compute as pc
    n = 80_000
    rng = np.random.default_rng(15)
    kv = rng.integers(0, 1_000_000, n)
    t = pa.table({"k": pa.array(kv, pa.int64()), "s": pa.array(["name-%06d" % v for v in rng.integers(0, 500_000, n)]), "x": pa.array(rng.integers(0, 100, n), pa.int32())})
    path = str(tmp_path / "bloom.parquet")
    papq.write_table(t, path, row_group_size=10_000, bloom_filter_options={"k": {"ndv": 10_000, "fpp": 0.01}, "s": {"ndv": 10_000, "fpp": 0.01}})
    types = [S.T_INT64, S.T_STRING, S.T_INT32]
    k, s = S.col(0, S.T_INT64), S.col(1, S.T_STRING)
    k_here, s_here = int(kv[12_345]), t.column("s")[54_321].as_py()
    pred = S.or_(S.eq(k, S.lit(k_here, S.T_INT64)), S.eq(s, S.lit(s_here, S.T_STRING)))

    def run(filters, predicate=pred):
        plan = S.filter_(S.native_scan([path], t.schema.names, types, data_filters=filters), predicate)
        it = native.CometExecIterator([], 3, plan.encode(), batch_size=0)
        batches = []
        while True:
            b = native.Native.executePlan(it.handle, 3)
            if b is None:
                break
            batches.append(b)
        node = S.decode_metric_node(it.metrics())
        it.close()
        while node[1]:
            node = node[1][0]
        return (pa.Table.from_batches(batches) if batches else None), node[0]

    full, m_full = run([])
    pruned, m_pruned = run([pred])
    want = t.filter(pc.or_(pc.equal(t.column("k"), k_here), pc.equal(t.column("s"), s_here)))
    assert full.num_rows == want.num_rows >= 2 and pruned.equals(full)
    assert sorted(pruned.column(0).to_pylist()) == sorted(want.column("k").to_pylist())
    assert m_pruned["row_groups_pruned_bloom_filter"] >= 4 and m_full["row_groups_pruned_bloom_filter"] == 0 and m_pruned["row_groups_pruned_statistics"] == 0
    assert m_pruned["bytes_scanned"] < m_full["bytes_scanned"]
    # IN over the string column, one literal present
    inp = S.in_(s, [S.lit("no such name", S.T_STRING), S.lit(s_here, S.T_STRING)])
    got, m_in = run([inp], inp)
    assert got.num_rows == pc.sum(pc.equal(t.column("s"), s_here)).as_py() and m_in["row_groups_pruned_bloom_filter"] >= 4


def test_hive_partition_columns(built, tmp_path):
    """Partition columns are appended after the file columns as one constant per file (operator.proto:103-109, planner.rs:1558-1575),
    NULL partition values included; a filter / aggregate above sees them like any column."""
    rng = np.random.default_rng(14)
    files, parts, want_tabs = [], [], []
    pvals = [(2023, "eu-west", tpch.days(2023, 1, 5)), (2024, None, tpch.days(2024, 2, 6)), (2025, "a much longer region name than fifteen bytes", None)]
    for i, pv in enumerate(pvals):
        n = 5000 + 777 * i
        t = pa.table({"x": pa.array(rng.integers(0, 100, n), pa.int64()), "s": pa.array(["v%d" % (j % 7) for j in range(n)])})
        path = str(tmp_path / f"part{i}.parquet")
        papq.write_table(t, path, row_group_size=2000)
        files.append(path)
        parts.append(pv)
        want_tabs.append(t.append_column("year", pa.array([pv[0]] * n, pa.int32())).append_column("region", pa.array([pv[1]] * n, pa.utf8()))
                         .append_column("day", pa.array([pv[2]] * n, pa.int32()).cast(pa.date32())))
    want = pa.concat_tables(want_tabs)
    pf = [("year", S.T_INT32), ("region", S.T_STRING), ("day", S.T_DATE)]
    scan = S.native_scan(files, ["x", "s"], [S.T_INT64, S.T_STRING], partition_fields=pf, partition_values=parts)
    got = pa.Table.from_batches(native.execute_to_table([], 5, scan.encode(), batch_size=0))
    for i, name in enumerate(want.schema.names):
        assert got.column(i).combine_chunks().equals(want.column(name).combine_chunks()), name
    # filter on a partition column + aggregate grouped by it
    plan = S.hash_agg(S.filter_(scan, S.gt(S.col(2, S.T_INT32), S.lit(2023, S.T_INT32))), [S.col(2, S.T_INT32)], [S.count(S.col(0, S.T_INT64)), S.sum_(S.col(0, S.T_INT64), S.T_INT64)])
    res = pa.Table.from_batches(native.execute_to_table([], 3, plan.encode(), batch_size=0))
    rows = sorted(zip(*[res.column(i).to_pylist() for i in range(3)]))
    exp = sorted((y, c.num_rows, sum(c.column("x").to_pylist())) for y, c in ((2024, want_tabs[1]), (2025, want_tabs[2])))
    assert rows == exp
