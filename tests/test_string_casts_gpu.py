"""Casts from and to strings on the GPU (conversion_funcs/cast.rs:228-420, string.rs, numeric.rs:593-704) against the oracle's
restatement, which tests/test_string_casts_cpu.py pins on the reference's own vectors: string → boolean / tinyint / smallint / int / bigint /
decimal / date in LEGACY, TRY and ANSI mode, and integers / booleans / decimals / dates / timestamps → string as output columns."""
import json
import os
import random

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
STR, I32 = S.T_STRING, S.T_INT32
KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")))["string_casts"]


def _run(plan, table, ncols, **kw):
    return pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(table)], ncols, plan.encode(), batch_size=0, **kw))


def _check(plan, table, ncols):
    from oracle import oracle as O
    got, want = _run(plan, table, ncols), O.run_plan_to_arrow(S, plan, table)
    assert got.num_rows == want.num_rows
    for i in range(ncols):
        gc, wc = got.column(i), want.column(i)
        assert gc.type == wc.type, (i, gc.type, wc.type)
        if pa.types.is_date32(gc.type):       # (years beyond datetime.date's: compare the epoch days)
            gc, wc = gc.cast(pa.int32()), wc.cast(pa.int32())
        g, w = gc.to_pylist(), wc.to_pylist()
        if g != w:
            k = next(j for j in range(len(g)) if g[j] != w[j])
            raise AssertionError(f"output {i}, row {k}: got {g[k]!r}, want {w[k]!r}, input {[c[k].as_py() for c in table.columns]!r}")
    return got


def _strings(n, seed):
    import importlib.util
    spec = importlib.util.spec_from_file_location("sc_cpu", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_string_casts_cpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    vals = [b.decode() for b in m._strings(random.Random(seed))]
    vals += [s for k in ("date_ok_18262", "date_invalid", "date_null_every_mode") for s in KATS[k]] + [s for s, _ in KATS["date_values"]]
    rng = np.random.default_rng(seed)
    pick = rng.integers(0, len(vals), n)
    return pa.table({"s": pa.array([vals[i] for i in pick], pa.utf8(), mask=rng.random(n) < 0.05), "k": pa.array(rng.integers(0, 100, n), pa.int32())})


TARGETS = [S.T_BOOL, S.T_INT8, S.T_INT16, S.T_INT32, S.T_INT64, S.decimal(5, 0), S.decimal(10, 2), S.decimal(18, 6), S.decimal(38, 10), S.decimal(38, 38), S.decimal(20, 19), S.T_DATE]


@pytest.mark.parametrize("mode", [S.LEGACY, S.TRY])
def test_string_to_values(built, mode):
    t = _strings(30_000, 11)
    s = S.col(0, STR)
    plan = S.project(S.scan([STR, I32]), [S.cast(s, to, mode) for to in TARGETS] + [S.col(1, I32)])
    _check(plan, t, len(TARGETS) + 1)


def test_ansi_raises_where_the_reference_raises(built):
    from oracle import oracle as O
    s = S.col(0, STR)
    good = pa.table({"s": pa.array([" 12 ", "-7", None, "+0", "127"]), "k": pa.array(np.arange(5, dtype=np.int32))})
    plan = S.project(S.scan([STR, I32]), [S.cast(s, S.T_INT8, S.ANSI), S.cast(s, S.decimal(10, 2), S.ANSI)])
    _check(plan, good, 2)
    dates = pa.table({"s": pa.array(["2020-01-01", " 2020-1-1T", None, "262143-01-01"]), "k": pa.array(np.arange(4, dtype=np.int32))})
    _check(S.project(S.scan([STR, I32]), [S.cast(s, S.T_DATE, S.ANSI)]), dates, 1)
    for bad, to, what in [("128", S.T_INT8, "CAST_INVALID_INPUT"), ("1.5", S.T_INT32, "CAST_INVALID_INPUT"), ("abc", S.T_BOOL, "CAST_INVALID_INPUT"), ("1e", S.decimal(10, 2), "CAST_INVALID_INPUT"),
                         ("123456789", S.decimal(5, 0), "NUMERIC_VALUE_OUT_OF_RANGE"), ("2020-02-30", S.T_DATE, "CAST_INVALID_INPUT"), ("", S.T_DATE, "CAST_INVALID_INPUT")]:
        tb = pa.table({"s": pa.array(["1", bad, None]), "k": pa.array(np.arange(3, dtype=np.int32))})
        p = S.project(S.scan([STR, I32]), [S.cast(s, to, S.ANSI)])
        with pytest.raises(O.OracleError, match=what):
            O.run_plan_to_arrow(S, p, tb)
        with pytest.raises(native.CometQueryExecutionException, match=what):
            _run(p, tb, 1)


def test_parsed_values_feed_filters_and_arithmetic(built):
    """a cast of a string column is an ordinary operand: compared, added, filtered on"""
    t = _strings(20_000, 12)
    s = S.col(0, STR)
    as_int = S.cast(s, S.T_INT32)
    src = S.filter_(S.scan([STR, I32]), S.gt(as_int, S.lit(0, I32)))
    got = _check(S.project(src, [S.math("add", as_int, S.col(1, I32), I32), S.cast(s, S.T_DATE), s]), t, 3)
    assert 0 < got.num_rows < t.num_rows


def _values_table(n, seed):
    rng = np.random.default_rng(seed)
    py = random.Random(seed)
    i64 = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    i64[:8] = [0, 1, -1, 2**63 - 1, -2**63, 10**18, -10**18, 9]
    dec = [py.randrange(-10**py.randrange(1, 38), 10**py.randrange(1, 38)) for _ in range(n)]
    dec[:4] = [0, 1, -1, 10**37]
    small = [py.randrange(-10**py.randrange(1, 9), 10**py.randrange(1, 9)) for _ in range(n)]
    small[:3] = [0, 1, -5]
    days = rng.integers(-800_000, 3_100_000, n).astype(np.int32)
    days[:6] = [0, -1, 18262, -719528, -719529, 2932897]
    us = rng.integers(-6 * 10**16, 3 * 10**17, n, dtype=np.int64)
    us[:6] = [0, 1, -1, 1_500_000, 86_399_999_999, -62_135_596_800_000_000]
    us[6:n // 2] = us[6:n // 2] // 1000 * 1000
    import decimal
    decimal.getcontext().prec = 60
    mask = lambda: rng.random(n) < 0.05
    return pa.table({
        "i8": pa.array(rng.integers(-128, 128, n).astype(np.int8), mask=mask()), "i16": pa.array(rng.integers(-2**15, 2**15, n).astype(np.int16)),
        "i32": pa.array(rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)), "i64": pa.array(i64, mask=mask()),
        "b": pa.array(rng.random(n) < 0.5, mask=mask()),
        "d38_10": pa.array([decimal.Decimal(v).scaleb(-10) for v in dec], pa.decimal128(38, 10), mask=mask()),
        "d38_38": pa.array([decimal.Decimal(v).scaleb(-38) for v in dec], pa.decimal128(38, 38)),
        "d12_2": pa.array([decimal.Decimal(v).scaleb(-2) for v in small], pa.decimal128(12, 2)),
        "d9_9": pa.array([decimal.Decimal(v).scaleb(-9) for v in small], pa.decimal128(9, 9), mask=mask()),
        "d10_0": pa.array([decimal.Decimal(v) for v in small], pa.decimal128(10, 0)),
        "date": pa.array(days, pa.int32(), mask=mask()).cast(pa.date32()),
        "ts": pa.array(us, pa.timestamp("us", tz="UTC"), mask=mask()),
        "ntz": pa.array(us, pa.timestamp("us")),
    })


def test_values_to_strings(built):
    t = _values_table(20_000, 21)
    types = [S.T_INT8, S.T_INT16, S.T_INT32, S.T_INT64, S.T_BOOL, S.decimal(38, 10), S.decimal(38, 38), S.decimal(12, 2), S.decimal(9, 9), S.decimal(10, 0), S.T_DATE, S.T_TIMESTAMP,
             S.DataType(S.TIMESTAMP_NTZ)]
    cols = [S.col(i, ty) for i, ty in enumerate(types)]
    exprs = [S.cast(c, STR) for c in cols]
    exprs += [S.cast(cols[5], STR, S.TRY), S.cast(cols[6], STR, S.ANSI), S.cast(cols[8], STR, S.TRY)]           # plain notation instead of BigDecimal.toString
    exprs += [S.cast(cols[11], STR, S.LEGACY, "+05:30"), S.cast(cols[11], STR, S.LEGACY, "-08:00")]
    exprs += [S.cast(S.math("add", cols[2], S.lit(1, I32), I32), STR), cols[0]]
    got = _check(S.project(S.scan(types), exprs), t, len(exprs))
    assert got.column(3).to_pylist()[3] == str(2**63 - 1)
    # below a filter, and with no surviving row
    src = S.filter_(S.scan(types), S.gt(cols[2], S.lit(0, I32)))
    _check(S.project(src, exprs[:13]), t, 13)
    none = S.filter_(S.scan(types), S.and_(S.gt(cols[2], S.lit(0, I32)), S.lt(cols[2], S.lit(0, I32))))
    assert native.execute_to_table([native.HostInput.from_table(t)], 13, S.project(none, exprs[:13]).encode(), batch_size=0) == []


def test_floats_to_strings_and_decimals(built):
    """Float → String (numeric.rs:137-221: the shortest digits, Java's notation) and Float → Decimal (numeric.rs:884-990: the shortest digits rounded
    HALF_UP, not the binary value — 0.5153125 at scale 6 is 0.515313); the Ryu routine is checked on the host against Python's repr() and numpy
    (tests/test_ryu_cpu.py)."""
    rng = np.random.default_rng(31)
    n = 30_000
    d = rng.standard_normal(n) * 10.0 ** rng.integers(-12, 12, n)
    d[:14] = [0.0, -0.0, 1.0, 0.001, 1e-4, 1e7, 9999999.0, 5e-324, float("nan"), float("inf"), float("-inf"), 0.5153125, 1.7976931348623157e308, 123456.789]
    bits = rng.integers(0, 2**63, n // 4).astype(np.uint64)
    d[100:100 + n // 4] = bits.view(np.float64)
    f = (rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n)).astype(np.float32)
    f[:8] = [0.0, 1.0, 0.1, 1e7, 3.4028235e38, 1.4e-45, 16777216.0, float("nan")]
    t = pa.table({"d": pa.array(d, mask=rng.random(n) < 0.05), "f": pa.array(f, mask=rng.random(n) < 0.05)})
    D, F = S.col(0, S.T_DOUBLE), S.col(1, S.T_FLOAT)
    exprs = [S.cast(D, STR), S.cast(F, STR), S.cast(D, S.decimal(38, 18)), S.cast(D, S.decimal(18, 2)), S.cast(D, S.decimal(10, 6)), S.cast(F, S.decimal(20, 10)), S.cast(D, S.decimal(38, 6), S.TRY)]
    got = _check(S.project(S.scan([S.T_DOUBLE, S.T_FLOAT]), exprs), t, len(exprs))
    assert got.column(0).to_pylist()[:8] == ["0.0", "-0.0", "1.0", "0.001", "1.0E-4", "1.0E7", "9999999.0", "4.9E-324"]
    from oracle import oracle as O
    bad = pa.table({"d": pa.array([1.0, 1e30]), "f": pa.array(np.array([1.0, 2.0], np.float32))})
    p = S.project(S.scan([S.T_DOUBLE, S.T_FLOAT]), [S.cast(D, S.decimal(10, 2), S.ANSI)])
    with pytest.raises(O.OracleError, match="NUMERIC_VALUE_OUT_OF_RANGE"):
        O.run_plan_to_arrow(S, p, bad)
    with pytest.raises(native.CometQueryExecutionException, match="NUMERIC_VALUE_OUT_OF_RANGE"):
        _run(p, bad, 1)


def test_strings_to_floats(built):
    """cast_string_to_float (string.rs:177-258): correctly rounded straight to the target's width (tests/test_strtod_cpu.py checks the routine against
    Python's float() and exact rationals), String.trim, inf / nan, one trailing d / f; bits compared, NaN included"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sd_cpu", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_strtod_cpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    vals = [v for v in m._numbers(random.Random(5), 6000) if len(v) < 400]
    vals += ["inf", "+INF", "-Infinity", "NaN", " 1.5 ", "1.5d", "1.5F", "1e5f", "nand", "", ".", "1e", "1.5x", "0x10", "１２", "1.5\x7f", "-0", "1.", ".5", "1e999", "1e-999", "abc"]
    rng = np.random.default_rng(6)
    n = 20_000
    t = pa.table({"s": pa.array([vals[i] for i in rng.integers(0, len(vals), n)], pa.utf8(), mask=rng.random(n) < 0.05), "k": pa.array(rng.integers(0, 9, n), pa.int32())})
    from oracle import oracle as O
    s = S.col(0, STR)
    for mode in (S.LEGACY, S.TRY):
        plan = S.project(S.scan([STR, I32]), [S.cast(s, S.T_DOUBLE, mode), S.cast(s, S.T_FLOAT, mode)])
        got, want = _run(plan, t, 2), O.run_plan_to_arrow(S, plan, t)
        for i, (ty, w) in enumerate([(np.uint64, 8), (np.uint32, 4)]):
            g, x = got.column(i).combine_chunks(), want.column(i).combine_chunks()
            assert g.is_valid().equals(x.is_valid()), i
            gb, xb = np.frombuffer(g.buffers()[1], ty)[:n], np.frombuffer(x.buffers()[1], ty)[:n]
            ok = np.asarray(g.is_valid())
            nan = np.isnan(np.asarray(x.fill_null(0)))
            bad = np.nonzero(ok & ~nan & (gb != xb))[0]
            assert len(bad) == 0, (i, t.column(0)[int(bad[0])].as_py(), hex(int(gb[bad[0]])), hex(int(xb[bad[0]])))
            assert (np.isnan(np.asarray(g.fill_null(0))) == nan).all()
    bad = pa.table({"s": pa.array(["1.5", "1.5x"]), "k": pa.array(np.arange(2, dtype=np.int32))})
    with pytest.raises(native.CometQueryExecutionException, match="CAST_INVALID_INPUT"):
        _run(S.project(S.scan([STR, I32]), [S.cast(s, S.T_DOUBLE, S.ANSI)]), bad, 1)


def _timestamp_strings(n, seed):
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("ts_cpu", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_string_timestamps_cpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    vals = m._values(random.Random(seed), 6000) + [v[1] for v in m.KATS]
    # what the kernel refuses instead of answering is kept out of the parity run (and checked below): a zone NAME inside the value, a time of day without a date
    vals = [v for v in vals if "/" not in v and not re.match(r"^\+?([Tt]|\d{1,2}:)", v.strip().lstrip("\u3000"))]
    rng = np.random.default_rng(seed)
    return pa.table({"s": pa.array([vals[i] for i in rng.integers(0, len(vals), n)], pa.utf8(), mask=rng.random(n) < 0.05), "k": pa.array(rng.integers(0, 9, n), pa.int32())})


@pytest.mark.parametrize("tz", ["UTC", "America/New_York", "Asia/Kolkata", "Pacific/Apia", "+05:30"])
def test_strings_to_timestamps(built, tz):
    """cast_string_to_timestamp / _ntz (string.rs:798-852, 1406-1853; tests/test_string_timestamps_cpu.py pins the oracle and the kernel's parser on the
    reference's 103 vectors): every shape, zone suffix and the session zone's gaps and overlaps, LEGACY and TRY, Spark 3 and 4 readings"""
    t = _timestamp_strings(20_000, 21)
    s = S.col(0, STR)
    for mode in (S.LEGACY, S.TRY):
        plan = S.project(S.scan([STR, I32]), [S.cast(s, S.T_TIMESTAMP, mode, tz), S.cast(s, S.T_TIMESTAMP, mode, tz, is_spark4_plus=True), S.cast(s, S.DataType(S.TIMESTAMP_NTZ), mode), S.col(1, I32)])
        from oracle import oracle as O
        got, want = _run(plan, t, 4), O.run_plan_to_arrow(S, plan, t)
        for i in range(3):
            g, w = got.column(i).combine_chunks(), want.column(i).combine_chunks()
            assert g.type == w.type, (i, g.type, w.type)
            g, w = g.cast(pa.int64()).to_pylist(), w.cast(pa.int64()).to_pylist()
            if g != w:
                k = next(j for j in range(len(g)) if g[j] != w[j])
                raise AssertionError(f"{tz} mode {mode} output {i}: got {g[k]!r}, want {w[k]!r}, input {t.column(0)[k].as_py()!r}")


def test_timestamp_strings_under_ansi_and_the_refusals(built):
    from oracle import oracle as O
    s = S.col(0, STR)
    good = pa.table({"s": pa.array(["2020-01-01T12:34:56.123456", " 2020-03-08 02:30:00 ", None, "2021-06-01 UTC+07:30", ""]), "k": pa.array(np.arange(5, dtype=np.int32))})
    plan = S.project(S.scan([STR, I32]), [S.cast(s, S.T_TIMESTAMP, S.ANSI, "America/New_York"), S.cast(s, S.DataType(S.TIMESTAMP_NTZ), S.ANSI)])
    got, want = _run(plan, good, 2), O.run_plan_to_arrow(S, plan, good)
    for i in range(2):
        assert got.column(i).cast(pa.int64()).to_pylist() == want.column(i).cast(pa.int64()).to_pylist(), i
    for bad, to in [("2020-13-01", S.T_TIMESTAMP), ("2020-01-01T25:00:00", S.DataType(S.TIMESTAMP_NTZ)), ("yesterday", S.T_TIMESTAMP), ("0119704", S.T_TIMESTAMP)]:
        tb = pa.table({"s": pa.array(["2020-01-01", bad, None]), "k": pa.array(np.arange(3, dtype=np.int32))})
        p = S.project(S.scan([STR, I32]), [S.cast(s, to, S.ANSI, "Asia/Kolkata")])
        with pytest.raises(O.OracleError, match="CAST_INVALID_INPUT"):
            O.run_plan_to_arrow(S, p, tb)
        with pytest.raises(native.CometQueryExecutionException, match='"toType":"TIMESTAMP' + ('_NTZ"' if to.type_id == S.TIMESTAMP_NTZ else '"')):
            _run(p, tb, 1)
    for refused in ["2020-01-01T12:34:56 Europe/Moscow", "T12:34", "12:34:56"]:
        tb = pa.table({"s": pa.array(["2020-01-01", refused]), "k": pa.array(np.arange(2, dtype=np.int32))})
        with pytest.raises(native.CometNativeException, match="names a time zone inside the value"):
            _run(S.project(S.scan([STR, I32]), [S.cast(s, S.T_TIMESTAMP, S.LEGACY, "UTC")]), tb, 1)


def test_unknown_time_zones_are_refused_by_name(built):
    """region zones come from the time-zone database (tests/test_temporal_casts_gpu.py); a name it does not hold fails createPlan"""
    t = _values_table(16, 3)
    plan = S.project(S.scan([S.T_TIMESTAMP]), [S.cast(S.col(0, S.T_TIMESTAMP), STR, S.LEGACY, "Mars/Olympus")])
    with pytest.raises(native.CometNativeException, match="Mars/Olympus"):
        _run(plan, t.select(["ts"]), 1)


def test_errors_name_the_offending_value(built):
    """ANSI errors carry what the JVM side reads back (ShimSparkErrorConverter.scala: params("value"), "precision", "scale", "fromType", "toType"):
    the kernel leaves the raise site and the value's bits — or the string's bytes — in its error block (kparams.h), csrc/err_sites.cpp formats
    them like the reference's raise sites do (tests/test_error_json_cpu.py holds the formats)"""
    import json
    from decimal import Decimal
    from datafusion_comet_amd.tpch import _dec128_array
    D = S.decimal(12, 2)
    fields = [S.T_INT64, S.T_DOUBLE, D, STR]
    i64, f64, d, s = (S.col(i, t) for i, t in enumerate(fields))

    def table(a=1, b=1.0, c=100, text="2020"):
        # (three rows; the middle one is the offender, its neighbours are harmless)
        return pa.table({"a": pa.array([1, a, None], pa.int64()), "b": pa.array([1.0, b, None], pa.float64()), "c": _dec128_array(np.array([100, c, 100], np.int64), 12, 2),
                         "s": pa.array(["2020", text, None], pa.utf8())})      # ("2020" is an integer, a decimal, a date and a timestamp)

    cases = [
        (S.cast(s, I32, S.ANSI), table(text="12x"), "CastInvalidValue", {"value": "12x", "fromType": "STRING", "toType": "INT"}),
        (S.cast(s, S.T_DATE, S.ANSI), table(text=" 2020-13-01 "), "InvalidInputInCastToDatetime", {"value": " 2020-13-01 ", "fromType": "STRING", "toType": "DATE"}),
        (S.cast(s, S.T_TIMESTAMP, S.ANSI, "Asia/Kolkata"), table(text='yester"day"'), "InvalidInputInCastToDatetime", {"value": 'yester"day"', "fromType": "STRING", "toType": "TIMESTAMP"}),
        (S.cast(s, S.decimal(5, 0), S.ANSI), table(text="123456789"), "NumericValueOutOfRange", {"value": "123456789", "precision": 5, "scale": 0}),
        (S.cast(s, S.decimal(10, 2), S.ANSI), table(text="1e"), "CastInvalidValue", {"value": "1e", "fromType": "STRING", "toType": "DECIMAL(10,2)"}),
        (S.cast(i64, I32, S.ANSI), table(a=2147483648), "CastOverFlow", {"value": "2147483648L", "fromType": "BIGINT", "toType": "INT"}),
        (S.cast(f64, I32, S.ANSI), table(b=3e9), "CastOverFlow", {"value": "3E9D", "fromType": "DOUBLE", "toType": "INT"}),
        (S.cast(d, I32, S.ANSI), table(c=999999999999), "CastOverFlow", {"value": "9999999999.99BD", "fromType": "DECIMAL(12,2)", "toType": "INT"}),
        (S.cast(i64, S.decimal(3, 2), S.ANSI), table(a=1000), "NumericValueOutOfRange", {"value": "1000", "precision": 3, "scale": 2}),
        (S.cast(f64, S.decimal(5, 2), S.ANSI), table(b=4242.42), "NumericValueOutOfRange", {"value": "4242.42", "precision": 5, "scale": 2}),
        (S.check_overflow(S.math("multiply", d, d, S.decimal(25, 4)), S.decimal(20, 4), True), table(c=999999999999), "NumericValueOutOfRange",
         {"value": str(999999999999 ** 2), "precision": 20, "scale": 4}),
    ]
    for expr, t, etype, params in cases:
        with pytest.raises(native.CometQueryExecutionException) as ei:
            _run(S.project(S.scan(fields), [expr]), t, 1)
        j = json.loads(str(ei.value))
        assert j["errorType"] == etype and j["params"] == params and "context" not in j, (etype, j)
    # an expression that carries Spark's SQLQueryContext (expr.proto:103-141) raises with it (SparkErrorWithContext::to_json, error.rs:806-831) —
    # an error that names a value and one that does not
    sql = "SELECT CAST(s AS INT), a % b FROM t"
    cast = S.with_context(S.cast(s, I32, S.ANSI), 7, sql_text=sql, start_index=7, stop_index=20, line=1, start_position=7, object_type="VIEW", object_name="v1")
    with pytest.raises(native.CometQueryExecutionException) as ei:
        _run(S.project(S.scan(fields), [cast]), table(text="12x"), 1)
    j = json.loads(str(ei.value))
    assert j["params"]["value"] == "12x" and j["context"]["sqlText"] == sql and j["context"]["objectName"] == "v1"
    assert j["summary"] == "== SQL of VIEW v1 (line 1, position 8) ==\n" + sql + "\n" + " " * 7 + "^" * 14
    rem = S.with_context(S.math("remainder", i64, S.math("subtract", i64, i64, S.T_INT64), S.T_INT64, S.ANSI), 8, sql_text_idx=0, start_index=23, stop_index=27, line=1, start_position=23)
    plan = S.project(S.scan(fields), [rem])
    plan.sql_text_pool = [sql]
    with pytest.raises(native.CometQueryExecutionException) as ei:
        _run(plan, table(a=5), 1)
    j = json.loads(str(ei.value))
    assert j["errorType"] == "RemainderByZero" and j["params"] == {} and j["context"]["sqlText"] == sql and j["summary"].endswith(" " * 23 + "^" * 5)
