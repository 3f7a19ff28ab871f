"""More of the reference's scalar functions in the fused kernels (QueryPlanSerde.scala:117-174, 300-352; comet_scalar_funcs.rs): the Float64 functions DataFusion /
datafusion-spark evaluate, pow / spark_log (math_funcs/{pow,log}.rs), factorial, bitwise_not / bit_count / bit_get / shiftrightunsigned, greatest / least,
last_day / date_trunc / next_day / make_date / date_from_unix_date / datepart isodow + week, seconds_to_timestamp, TruncTimestamp, UnixTimestamp.
Against the oracle: bit-exact for integers, dates and timestamps (the calendar code is the source tests/test_dates_cpu.py walks on the host); within
8 ulp for what both sides hand to a libm (ocml here, the platform's libm behind Rust's std there; numpy's in the oracle), exact for degrees / radians /
rint / pi."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
I8, I16, I32, I64, F64, D, STR, TS, NTZ = S.T_INT8, S.T_INT16, S.T_INT32, S.T_INT64, S.T_DOUBLE, S.T_DATE, S.T_STRING, S.T_TIMESTAMP, S.DataType(S.TIMESTAMP_NTZ)
f = S.scalar_func


def _run(plan, table, ncols, **kw):
    return pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(table)], ncols, plan.encode(), batch_size=0, **kw))


def _ulps(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ia, ib = a.view(np.int64).copy(), b.view(np.int64).copy()
    ia = np.where(ia < 0, np.int64(-2**63) - ia, ia)      # order-preserving integers: neighbouring doubles are neighbours
    ib = np.where(ib < 0, np.int64(-2**63) - ib, ib)
    d = np.abs(ia.astype(object) - ib.astype(object)).astype(np.float64)      # (exact integers: a float64 of a 62-bit integer is only good to 1024)
    return np.where(np.isnan(a) & np.isnan(b), 0.0, np.where(np.isnan(a) | np.isnan(b), np.inf, d))


def _check(exprs, types, table, tol=None):
    from oracle import oracle as O
    plan = S.project(S.scan(types), exprs)
    got, want = _run(plan, table, len(exprs)), O.run_plan_to_arrow(S, plan, table)
    for i in range(len(exprs)):
        g, w = got.column(i).combine_chunks(), want.column(i).combine_chunks()
        assert g.is_valid().to_pylist() == w.is_valid().to_pylist(), f"output {i}: validity"
        if tol is not None and tol[i]:
            gv, wv = g.fill_null(0.0).to_numpy(zero_copy_only=False), w.fill_null(0.0).to_numpy(zero_copy_only=False)
            worst = _ulps(gv, wv)
            assert worst.max() <= tol[i], f"output {i}: {worst.max()} ulp at {gv[worst.argmax()]!r} vs {wv[worst.argmax()]!r}"
        else:
            assert g.to_pylist() == w.to_pylist() or all((a == b) or (a != a and b != b) for a, b in zip(g.to_pylist(), w.to_pylist())), f"output {i}"
    return got


def _doubles(n, seed):
    rng = np.random.default_rng(seed)
    x = np.concatenate([rng.normal(0, 10, n // 2), rng.uniform(-1, 1, n // 4), rng.uniform(0, 700, n // 8), 10.0 ** rng.uniform(-300, 5, n - n // 2 - n // 4 - n // 8)])
    x[:12] = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.5, 2.0, 1e5, -1e5, 1e-320]
    rng.shuffle(x)
    return pa.array(x, pa.float64(), mask=rng.random(n) < 0.05)


UNARY = ["acos", "acosh", "asin", "asinh", "atan", "atanh", "cbrt", "cos", "cosh", "exp", "expm1", "ln", "log2", "log10", "sin", "sinh", "tan", "tanh", "cot", "csc", "sec", "degrees", "radians", "rint"]
EXACT = {"degrees", "radians", "rint"}


def test_float64_functions(built):
    t = pa.table({"x": _doubles(40_000, 31), "y": _doubles(40_000, 32)})
    x, y = S.col(0, F64), S.col(1, F64)
    for k in range(0, len(UNARY), 12):
        names = UNARY[k:k + 12]
        _check([f(n, [x], F64) for n in names], [F64, F64], t, tol=[0 if n in EXACT else 8 for n in names])
    _check([f("atan2", [x, y], F64), f("pow", [x, y], F64), f("spark_log", [x, y], F64), f("pi", [], F64), f("pow", [x, S.lit(float("inf"), F64)], F64), f("pow", [S.lit(0.0, F64), y], F64),
            f("greatest", [x, y, S.lit(1.5, F64)], F64), f("least", [x, y], F64)], [F64, F64], t, tol=[8, 8, 8, 0, 0, 0, 0, 0])


def test_integers_and_bits(built):
    rng = np.random.default_rng(33)
    n = 30_000
    m = lambda: rng.random(n) < 0.05
    t = pa.table({"a": pa.array(rng.integers(-5, 30, n), pa.int32(), mask=m()), "b": pa.array(rng.integers(-2**63, 2**63 - 1, n), pa.int64(), mask=m()),
                  "c": pa.array(rng.integers(-2**31, 2**31 - 1, n), pa.int32(), mask=m()), "d": pa.array(rng.integers(-128, 127, n), pa.int8(), mask=m()), "e": pa.array(rng.integers(-2**15, 2**15 - 1, n), pa.int16(), mask=m())})
    a, b, c, d, e = S.col(0, I32), S.col(1, I64), S.col(2, I32), S.col(3, I8), S.col(4, I16)
    _check([f("factorial", [a], I64), f("bitwise_not", [c], I32), f("bitwise_not", [b], I64), f("bitwise_not", [d], I8), f("bitwise_not", [e], I16), f("bit_count", [c], I32), f("bit_count", [b], I32),
            f("bit_count", [d], I32), f("bit_get", [b, S.lit(63, I32)], I8), f("bit_get", [c, S.lit(0, I32)], I8), f("shiftrightunsigned", [c, S.lit(3, I32)], I32), f("shiftrightunsigned", [b, a], I64),
            f("greatest", [a, c, S.lit(7, I32)], I32), f("least", [b, S.lit(0, I64)], I64)], [I32, I64, I32, I8, I16], t)


def test_dates(built):
    rng = np.random.default_rng(34)
    n = 30_000
    days = np.concatenate([rng.integers(-135_000, 160_000, n - 6), [0, -1, 19782, 11016, -719162, 2932880]]).astype(np.int32)      # 1600 … 2400, the epoch, a leap day, the year 1 and 9999
    t = pa.table({"d": pa.array(days, pa.int32(), mask=rng.random(n) < 0.05).cast(pa.date32()), "y": pa.array(rng.integers(1, 9999, n), pa.int32()), "m": pa.array(rng.integers(-1, 15, n), pa.int32()),
                  "dd": pa.array(rng.integers(-1, 34, n), pa.int32(), mask=rng.random(n) < 0.05)})
    d, y, m, dd = S.col(0, D), S.col(1, I32), S.col(2, I32), S.col(3, I32)
    L = lambda s: S.lit(s, STR)
    _check([f("last_day", [d], D), f("date_trunc", [d, L("year")], D), f("date_trunc", [d, L("QUARTER")], D), f("date_trunc", [d, L("mon")], D), f("date_trunc", [d, L("week")], D), f("next_day", [d, L("tue")], D),
            f("next_day", [d, L("SUNDAY")], D), f("next_day", [d, L("xyz")], D), f("make_date", [y, m, dd], D), f("date_from_unix_date", [y], D), S.date_part("isodow", d), S.date_part("week", d)], [D, I32, I32, I32], t)


def test_timestamps(built):
    rng = np.random.default_rng(35)
    n = 20_000
    us = rng.integers(-6 * 10**15, 6 * 10**15, n)
    us[:4] = [0, -1, 1, 86_399_999_999]
    t = pa.table({"ts": pa.array(us, pa.int64(), mask=rng.random(n) < 0.05).cast(pa.timestamp("us", tz="UTC")), "ntz": pa.array(us, pa.int64()).cast(pa.timestamp("us")),
                  "d": pa.array(rng.integers(-40_000, 40_000, n), pa.int32()).cast(pa.date32()), "s": pa.array(rng.integers(-2**31, 2**31 - 1, n), pa.int32()), "x": _doubles(n, 36)})
    ts, ntz, d, s, x = S.col(0, TS), S.col(1, NTZ), S.col(2, D), S.col(3, I32), S.col(4, F64)
    types = [TS, NTZ, D, I32, F64]
    units = ["year", "quarter", "MONTH", "week", "day", "hour", "minute", "second", "millisecond", "microsecond"]
    _check([S.trunc_timestamp(ts, u) for u in units], types, t)
    _check([S.trunc_timestamp(ts, u, "+05:30") for u in ("week", "DD", "hour")] + [S.trunc_timestamp(ntz, u) for u in ("yyyy", "mm", "week", "minute")], types, t)
    _check([S.unix_timestamp(ts), S.unix_timestamp(ntz), S.unix_timestamp(d), S.unix_timestamp(d, "America/Los_Angeles"), S.unix_timestamp(d, "+05:30"), f("seconds_to_timestamp", [s], TS),
            f("seconds_to_timestamp", [x], TS), f("seconds_to_timestamp", [S.cast(s, I64)], TS)], types, t)
    # one child in several zones inside ONE kernel: the zone belongs to the expression's identity (the generator once merged these)
    _check([S.time_part("hour", ts, "UTC"), S.time_part("hour", ts, "+05:30"), S.time_part("hour", ts, "America/Los_Angeles"), S.cast(ts, D, timezone="UTC"), S.cast(ts, D, timezone="Asia/Tokyo")], types, t)


def test_refusals_and_errors(built):
    t = pa.table({"a": pa.array([1, 2], pa.int64()), "ts": pa.array([0, 1], pa.int64()).cast(pa.timestamp("us", tz="UTC")), "d": pa.array([0, 1], pa.int32()).cast(pa.date32())})
    types = [I64, TS, D]
    for e, why in ((S.trunc_timestamp(S.col(1, TS), "day", "Europe/Berlin"), "transitions"), (S.trunc_timestamp(S.col(1, TS), "fortnight"), "Unsupported format"),
                   (f("date_trunc", [S.col(2, D), S.lit("day", STR)], D), "Unsupported format"), (f("bit_get", [S.col(0, I64), S.lit(64, I32)], I8), "literal position"),
                   (f("make_date", [S.lit(1, I32), S.lit(1, I32), S.lit(1, I32)], D, fail_on_error=True), "ANSI"), (f("sin", [S.col(0, I64)], F64), "Float64")):
        with pytest.raises(native.CometNativeException, match=why):
            _run(S.project(S.scan(types), [e]), t, 1)
    big = pa.table({"a": pa.array([5, 9_223_372_036_855], pa.int64()), "ts": t.column(1), "d": t.column(2)})
    with pytest.raises(Exception, match="long overflow"):
        _run(S.project(S.scan(types), [f("seconds_to_timestamp", [S.col(0, I64)], TS)]), big, 1)


def test_scalar_subqueries(built):
    """Subquery{id, datatype} (expr.proto:513-516; expressions/subquery.rs): the values registered for the plan (comet_plan_set_subquery here, CometScalarSubquery's
    static methods under the JVM — tests/test_jni_shim_cpu.py) become literals of the kernels at the first executePlan; a second plan with other values must not
    run the first one's kernels (the values are part of the plan's hash)."""
    from oracle import oracle as O
    rng = np.random.default_rng(37)
    n = 20_000
    DEC = S.decimal(12, 2)
    t = pa.table({"a": pa.array(rng.integers(-50, 50, n), pa.int64(), mask=rng.random(n) < 0.05), "x": _doubles(n, 38), "s": pa.array(np.array(["ab", "cd", "a much longer string than 15 bytes"], dtype=object)[rng.integers(0, 3, n)]),
                  "d": pa.array(rng.integers(18_000, 20_000, n), pa.int32()).cast(pa.date32())})
    types = [I64, F64, STR, D]
    a, x, s, d = S.col(0, I64), S.col(1, F64), S.col(2, STR), S.col(3, D)
    plan = S.project(S.filter_(S.scan(types), S.gt(a, S.subquery(1, I64))),
                     [S.math("add", a, S.subquery(2, I64), I64), S.math("multiply", x, S.subquery(3, F64), F64), S.subquery(4, DEC), S.eq(s, S.subquery(5, STR)), S.subquery(6, I32), S.lt(d, S.subquery(7, D)),
                      S.and_(S.subquery(8, S.T_BOOL), S.is_not_null(a))])
    for vals in ({1: 10, 2: 1000, 3: 0.5, 4: 123456, 5: "a much longer string than 15 bytes", 6: None, 7: 19_000, 8: True},
                 {1: -20, 2: -1, 3: -3.0, 4: -99, 5: "cd", 6: 7, 7: 18_500, 8: False}):
        tys = {1: I64, 2: I64, 3: F64, 4: DEC, 5: STR, 6: I32, 7: D, 8: S.T_BOOL}
        O.SUBQUERIES = dict(vals)
        try:
            got = _run(plan, t, 7, subqueries={k: native.subquery_value(v, tys[k]) for k, v in vals.items()})
            want = O.run_plan_to_arrow(S, plan, t)
        finally:
            O.SUBQUERIES = {}
        assert got.num_rows == want.num_rows > 0
        for i in range(7):
            assert all((p == q) or (p != p and q != q) for p, q in zip(got.column(i).to_pylist(), want.column(i).to_pylist())), f"output {i}"
