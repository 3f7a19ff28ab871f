"""Parquet decode pinned on the REFERENCE's own fixture files (spark/src/test/resources/test-data/*.parquet, copied to
tests/golden/parquet/): files written by parquet-mr 1.10 / 1.12 and by a third-party writer, i.e. not by the pyarrow that checks them.

  * every fixture is decoded on the GPU through NativeScan and compared with pyarrow (parquet-cpp), column by column;
  * the values the reference's Scala suite asserts are asserted here too (ParquetReadSuite.scala:1448-1487: dec-in-fixed-len =
    id % 10 as decimal(10,2); the first / last rows of the two decimal32-written-as-64-bit files; :1962-1982: eight dates before 1582 read
    WITHOUT rebasing, as the reference documents for Comet);
  * TIMESTAMP_MILLIS columns come back as microseconds, INT96 as microseconds (dictionary- and plain-encoded columns in every file).

Then the schema-adapter behaviours the reference implements in parquet/schema_adapter.rs (:76-250 field ids, :352-525 missing columns and
default values, :749-771 type promotion gating, :843-860 LTZ → NTZ) on files written here."""
import glob
import json
import os
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
FIXTURES = os.path.join(os.path.dirname(__file__), "golden", "parquet")
NAMES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(FIXTURES, "*.parquet")))


def spark_type(t: pa.DataType):
    if pa.types.is_decimal(t):
        return S.decimal(t.precision, t.scale)
    if pa.types.is_timestamp(t):
        return S.T_TIMESTAMP
    return {pa.int32(): S.T_INT32, pa.int64(): S.T_INT64, pa.float64(): S.T_DOUBLE, pa.float32(): S.T_FLOAT, pa.date32(): S.T_DATE,
            pa.utf8(): S.T_STRING, pa.bool_(): S.T_BOOL, pa.int16(): S.T_INT16, pa.int8(): S.T_INT8}[t]


def scan(files, names, types, **kw):
    plan = S.native_scan(files, names, types, **kw)
    out = native.execute_to_table([], len(names), plan.encode(), batch_size=0)
    return pa.Table.from_batches(out) if out else None


def as_python(col: pa.ChunkedArray):
    col = col.combine_chunks()
    if pa.types.is_timestamp(col.type):
        return col.cast(pa.int64()).to_pylist()          # microseconds since the epoch
    return col.to_pylist()


def expected(path):
    t = papq.read_table(path, coerce_int96_timestamp_unit="us")
    cols = []
    for c in t.columns:
        if pa.types.is_timestamp(c.type) and c.type.unit != "us":
            c = c.cast(pa.timestamp("us", tz=c.type.tz))   # what Spark's TimestampType holds
        cols.append(c)
    return pa.table(cols, names=t.schema.names)


def test_all_18_reference_fixtures_are_present():
    assert len(NAMES) == 18


@pytest.mark.parametrize("name", NAMES)
def test_reference_fixture_matches_pyarrow(built, name):
    path = os.path.join(FIXTURES, name)
    want = expected(path)
    got = scan([path], want.schema.names, [spark_type(f.type) for f in want.schema])
    assert got.num_rows == want.num_rows
    for i, f in enumerate(want.schema):
        assert as_python(got.column(i)) == as_python(want.column(i)), f"{name}: column {f.name}"


def test_reference_scala_assertions_on_the_decimal_fixtures(built):
    t = scan([os.path.join(FIXTURES, "dec-in-fixed-len.parquet")], ["fixed_len_dec"], [S.decimal(10, 2)])
    assert t.column(0).to_pylist() == [Decimal(i % 10).scaleb(0).quantize(Decimal("0.01")) for i in range(16)]
    t = scan([os.path.join(FIXTURES, "decimal32-written-as-64-bit.snappy.parquet")], ["_c0"], [S.decimal(9, 1)])
    unscaled = [None if v is None else int(v.scaleb(1)) for v in t.column(0).to_pylist()]
    assert unscaled == [792059492, 986842987, 540247998, None, 357991078, 494131059, 92536396, 426847157, -999999999, 204486094]
    t = scan([os.path.join(FIXTURES, "decimal32-written-as-64-bit-dict.snappy.parquet")], ["_c0"], [S.decimal(3, 1)])
    unscaled = [None if v is None else int(v.scaleb(1)) for v in t.column(0).to_pylist()]
    assert len(unscaled) == 2048
    assert unscaled[:10] == [751, 937, 511, None, 337, 467, 84, 403, -999, 190]
    assert unscaled[-10:] == [866, 20, 492, 76, 824, 604, 343, 820, 864, 243]


def test_reference_ancient_dates_are_read_without_rebase(built):
    t = scan([os.path.join(FIXTURES, "before_1582_date_v3_2_0.snappy.parquet")], ["dict", "plain"], [S.T_DATE, S.T_DATE])
    assert t.num_rows == 8
    for d in t.column(0).to_pylist() + t.column(1).to_pylist():
        assert d.year < 1582


def test_timestamp_millis_is_scaled_to_micros_and_pruning_ignores_its_statistics(built, tmp_path):
    ms = np.array([0, 1, -1, 1_600_000_000_123, -62_135_596_800_000, 253_402_300_799_999], dtype=np.int64)
    t = pa.table({"ts": pa.array(ms, pa.timestamp("ms", tz="UTC")), "k": pa.array(range(len(ms)), pa.int32())})
    path = str(tmp_path / "ms.parquet")
    papq.write_table(t, path)
    got = scan([path], ["ts", "k"], [S.T_TIMESTAMP, S.T_INT32])
    assert as_python(got.column(0)) == [int(v) * 1000 for v in ms]
    # a microsecond literal that lies above every MILLISECOND statistic but below most microsecond values: nothing may be pruned
    flt = S.gt(S.col(0, S.T_TIMESTAMP), S.lit(1_600_000_000_123 * 10, S.T_TIMESTAMP))
    got = scan([path], ["ts", "k"], [S.T_TIMESTAMP, S.T_INT32], data_filters=[flt])
    assert got.num_rows == len(ms)


def test_unsigned_and_nanos_annotations(built, tmp_path):
    t = pa.table({"u8": pa.array([0, 200, 255], pa.uint8()), "u16": pa.array([0, 40000, 65535], pa.uint16()),
                  "u32": pa.array([0, 3_000_000_000, 4_294_967_295], pa.uint32()), "u64": pa.array([0, 2**63, 2**64 - 1], pa.uint64())})
    path = str(tmp_path / "u.parquet")
    papq.write_table(t, path)
    got = scan([path], ["u8", "u16", "u32", "u64"], [S.T_INT16, S.T_INT32, S.T_INT64, S.decimal(20, 0)])
    assert got.column(0).to_pylist() == [0, 200, 255] and got.column(1).to_pylist() == [0, 40000, 65535]
    assert got.column(2).to_pylist() == [0, 3_000_000_000, 4_294_967_295]
    assert [int(v) for v in got.column(3).to_pylist()] == [0, 2**63, 2**64 - 1]
    with pytest.raises(native.CometNativeException, match="UINT_32"):      # never a silent signed reinterpretation
        scan([path], ["u32"], [S.T_INT32])
    tn = pa.table({"ts": pa.array([1, 2], pa.timestamp("ns"))})
    pn = str(tmp_path / "ns.parquet")
    papq.write_table(tn, pn, version="2.6")
    with pytest.raises(native.CometNativeException, match="NANOS"):
        scan([pn], ["ts"], [S.T_TIMESTAMP])


def _write_with_ids(path, cols):
    """cols: [(name, array, field_id or None)]"""
    fields = [pa.field(n, a.type, metadata=({b"PARQUET:field_id": str(i).encode()} if i is not None else None)) for n, a, i in cols]
    papq.write_table(pa.Table.from_arrays([a for _, a, _ in cols], schema=pa.schema(fields)), path)


def test_field_id_matching(built, tmp_path):
    path = str(tmp_path / "ids.parquet")
    _write_with_ids(path, [("x", pa.array([1, 2, 3], pa.int64()), 1), ("y", pa.array([10, 20, 30], pa.int64()), 2), ("z", pa.array([7, 8, 9], pa.int64()), None)])
    I = S.T_INT64
    # renamed columns resolve through their ids; an id-bearing column whose id the file lacks is NULL even though its NAME exists;
    # a column without an id still matches by name
    got = scan([path], ["renamed_y", "x", "z"], [I, I, I], field_ids=[2, 99, None], use_field_id=True)
    assert got.column(0).to_pylist() == [10, 20, 30] and got.column(1).to_pylist() == [None] * 3 and got.column(2).to_pylist() == [7, 8, 9]
    # ids are ignored unless use_field_id is set: plain name matching
    got = scan([path], ["renamed_y", "x"], [I, I], field_ids=[2, 99])
    assert got.column(0).to_pylist() == [None] * 3 and got.column(1).to_pylist() == [1, 2, 3]
    dup = str(tmp_path / "dup.parquet")
    _write_with_ids(dup, [("a", pa.array([1], pa.int64()), 5), ("b", pa.array([2], pa.int64()), 5)])
    with pytest.raises(native.CometQueryExecutionException) as e:
        scan([dup], ["q"], [I], field_ids=[5], use_field_id=True)
    err = json.loads(str(e.value))
    assert err["errorType"] == "DuplicateFieldByFieldId" and err["params"] == {"requiredId": 5, "matchedFields": "a, b"}
    noids = str(tmp_path / "noids.parquet")
    papq.write_table(pa.table({"x": pa.array([1, 2], pa.int64())}), noids)
    with pytest.raises(native.CometQueryExecutionException, match="ParquetMissingFieldIds"):
        scan([noids], ["x"], [I], field_ids=[1], use_field_id=True)
    got = scan([noids], ["x"], [I], field_ids=[1], use_field_id=True, ignore_missing_field_id=True)
    assert got.column(0).to_pylist() == [None, None]


def test_case_insensitive_duplicates_are_an_error(built, tmp_path):
    path = str(tmp_path / "case.parquet")
    papq.write_table(pa.table({"a": pa.array([1], pa.int64()), "A": pa.array([2], pa.int64()), "b": pa.array([3], pa.int64())}), path)
    assert scan([path], ["B"], [S.T_INT64], case_sensitive=False).column(0).to_pylist() == [3]
    with pytest.raises(native.CometQueryExecutionException) as e:
        scan([path], ["a"], [S.T_INT64], case_sensitive=False)
    err = json.loads(str(e.value))
    assert err["errorType"] == "DuplicateFieldCaseInsensitive" and err["params"] == {"requiredFieldName": "a", "matchedOrcFields": "[a, A]"}
    assert scan([path], ["A"], [S.T_INT64], case_sensitive=True).column(0).to_pylist() == [2]


def test_missing_columns_defaults_and_per_file_schema_evolution(built, tmp_path):
    """One partition, three files: the oldest lacks `added` and `s`, the middle one stores `v` as INT32, the newest has everything.
    `added` has a default value (Spark's ALTER TABLE ADD COLUMN … DEFAULT), `s` does not."""
    old, mid, new = (str(tmp_path / f"{n}.parquet") for n in ("old", "mid", "new"))
    papq.write_table(pa.table({"k": pa.array([1, 2], pa.int64()), "v": pa.array([10, 20], pa.int64())}), old)
    papq.write_table(pa.table({"k": pa.array([3], pa.int64()), "v": pa.array([30], pa.int32()), "added": pa.array([Decimal("7.50")], pa.decimal128(9, 2)),
                               "s": pa.array(["mid"], pa.utf8())}), mid)
    papq.write_table(pa.table({"k": pa.array([4, 5], pa.int64()), "v": pa.array([40, None], pa.int64()), "added": pa.array([None, Decimal("1.25")], pa.decimal128(9, 2)),
                               "s": pa.array([None, "new"], pa.utf8()), "flag": pa.array([True, False])}), new)
    names, types = ["k", "v", "added", "s", "flag", "d"], [S.T_INT64, S.T_INT64, S.decimal(9, 2), S.T_STRING, S.T_BOOL, S.T_DATE]
    got = scan([old, mid, new], names, types, default_values={2: 999, 4: True})
    assert got.column(0).to_pylist() == [1, 2, 3, 4, 5]
    assert got.column(1).to_pylist() == [10, 20, 30, 40, None]
    assert got.column(2).to_pylist() == [Decimal("9.99"), Decimal("9.99"), Decimal("7.50"), None, Decimal("1.25")]
    assert got.column(3).to_pylist() == [None, None, "mid", None, "new"]
    assert got.column(4).to_pylist() == [True, True, True, True, False]
    assert got.column(5).to_pylist() == [None] * 5           # in no file, no default
    # a default on a column that no file has, of every fixed-width kind and a string
    got = scan([old], ["k", "i", "f", "s", "ts"], [S.T_INT64, S.T_INT32, S.T_DOUBLE, S.T_STRING, S.T_TIMESTAMP], default_values={1: -5, 2: 2.5, 3: "dflt", 4: 123456})
    assert got.column(1).to_pylist() == [-5, -5] and got.column(2).to_pylist() == [2.5, 2.5] and got.column(3).to_pylist() == ["dflt", "dflt"]
    assert as_python(got.column(4)) == [123456, 123456]


def test_type_promotion_and_ltz_to_ntz_gates(built, tmp_path):
    path = str(tmp_path / "p.parquet")
    papq.write_table(pa.table({"i": pa.array([1, 2], pa.int32()), "f": pa.array([1.5, 2.5], pa.float32()), "ts": pa.array([1, 2], pa.timestamp("us", tz="UTC")),
                               "ntz": pa.array([3, 4], pa.timestamp("us"))}), path)
    got = scan([path], ["i", "f"], [S.T_INT64, S.T_DOUBLE])
    assert got.column(0).to_pylist() == [1, 2] and got.column(1).to_pylist() == [1.5, 2.5]
    for name, t, found, want in (("i", S.T_INT64, "INT32", "bigint"), ("f", S.T_DOUBLE, "FLOAT", "double"), ("i", S.T_DOUBLE, "INT32", "double")):
        with pytest.raises(native.CometQueryExecutionException) as e:
            scan([path], [name], [t], allow_type_promotion=False)
        err = json.loads(str(e.value))
        assert err["errorType"] == "ParquetSchemaConvert"
        assert err["params"] == {"filePath": "", "column": f"[{name}]", "physicalType": found, "sparkType": want}
    NTZ = S.DataType(S.TIMESTAMP_NTZ)
    assert as_python(scan([path], ["ts", "ntz"], [NTZ, NTZ]).column(0)) == [1, 2]
    with pytest.raises(native.CometQueryExecutionException, match="timestamp_ntz"):
        scan([path], ["ts"], [NTZ], allow_timestamp_ltz_to_ntz=False)
    assert as_python(scan([path], ["ntz"], [NTZ], allow_timestamp_ltz_to_ntz=False).column(0)) == [3, 4]
