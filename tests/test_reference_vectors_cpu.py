"""The oracle against more of the reference's own unit-test vectors (casts, date parts, round / ceil / floor / abs, checked arithmetic): float / double / decimal → timestamp (conversion_funcs/numeric.rs:1753-1860
test_cast_decimal_to_timestamp, test_cast_float_to_timestamp) and the rules of :87-135, 1184-1208; tests/test_temporal_casts_gpu.py runs the
same casts on the GPU against this oracle."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import serde as S
from datafusion_comet_amd.tpch import _dec128_array
from oracle import oracle as O

TS, NTZ = S.T_TIMESTAMP, S.DataType(S.TIMESTAMP_NTZ)


def _cast(arr, frm, to, mode=S.LEGACY, tz="UTC"):
    plan = S.project(S.scan([frm]), [S.cast(S.col(0, frm), to, mode, tz)])
    return O.run_plan_to_arrow(S, plan, pa.table({"v": arr})).column(0).cast(pa.int64()).to_pylist()


def test_the_references_decimal_vectors():
    for to in (TS, NTZ):
        for tz in ("UTC", "America/Los_Angeles"):      # (the zone only labels the result: the value is epoch microseconds)
            # numeric.rs:1765-1782: Decimal128(18,6) unscaled 0, 1e6, -1e6, 1.5e6, 123456789 → the same microseconds
            got = _cast(_dec128_array(np.array([0, 1_000_000, -1_000_000, 1_500_000, 123_456_789], np.int64), 18, 6), S.decimal(18, 6), to, tz=tz)
            assert got == [0, 1_000_000, -1_000_000, 1_500_000, 123_456_789]
            # :1786-1800: Decimal128(10,2) unscaled 100, 150, -250 → 1.0 s, 1.5 s, -2.5 s
            assert _cast(_dec128_array(np.array([100, 150, -250], np.int64), 10, 2), S.decimal(10, 2), to, tz=tz) == [1_000_000, 1_500_000, -2_500_000]
    # truncation toward zero below a microsecond (scale 8), wrapping to 64 bits like `as_i128() as i64` (scale 0, 10^13 s)
    assert _cast(_dec128_array(np.array([199, -199, 100], np.int64), 10, 8), S.decimal(10, 8), TS) == [1, -1, 1]
    assert _cast(_dec128_array(np.array([10**13], np.int64), 20, 0), S.decimal(20, 0), TS) == [(10**19 + 2**63) % 2**64 - 2**63]


def test_the_references_float_vectors():
    for mode in (S.LEGACY, S.ANSI, S.TRY):
        for to in (TS, NTZ):
            # numeric.rs:1815-1832 / :1836-1848
            assert _cast(pa.array([0.0, 1.0, -1.0, 1.5, 0.000001, None], pa.float64()), S.T_DOUBLE, to, mode) == [0, 1_000_000, -1_000_000, 1_500_000, 1, None]
            assert _cast(pa.array([0.0, 1.0, -1.0, None], pa.float32()), S.T_FLOAT, to, mode) == [0, 1_000_000, -1_000_000, None]
    # :1852-1858: NaN and infinity raise under ANSI, are NULL otherwise; so is a product beyond a bigint
    for bad in (float("nan"), float("inf"), float("-inf")):
        assert _cast(pa.array([bad]), S.T_DOUBLE, TS) == [None]
        with pytest.raises(O.OracleError, match="CAST_INVALID_INPUT"):
            _cast(pa.array([bad]), S.T_DOUBLE, TS, S.ANSI)
    assert _cast(pa.array([1e13, -1e13, 9.3e12, 1e303]), S.T_DOUBLE, TS) == [None, None, None, None]
    with pytest.raises(O.OracleError, match="CAST_OVERFLOW"):
        _cast(pa.array([1e13]), S.T_DOUBLE, TS, S.ANSI)
    # the edge: 9223372036854.775 s · 10^6 rounds to 2^63 as a double — floor(micros) ≤ i64::MAX as f64 holds, `as i64` saturates
    assert _cast(pa.array([9223372036854.775, -9223372036854.775]), S.T_DOUBLE, TS) == [2**63 - 1, -2**63]


def test_the_references_decimal_to_boolean_vector():
    # numeric.rs:1286-1298: Decimal128(10,2) 0, 100, -100, NULL → false, true, true, NULL
    arr = pa.array([0, 100, -100, None], pa.int64())
    from decimal import Decimal
    d = pa.array([None if v is None else Decimal(v).scaleb(-2) for v in arr.to_pylist()], pa.decimal128(10, 2))
    plan = S.project(S.scan([S.decimal(10, 2)]), [S.cast(S.col(0, S.decimal(10, 2)), S.T_BOOL)])
    assert O.run_plan_to_arrow(S, plan, pa.table({"v": d})).column(0).to_pylist() == [False, True, True, None]


def test_date_to_int_is_the_day_number():
    # cast.rs:273-276 (Date32 → Int32 reinterprets); the plan compiles for gfx950 and the oracle gives the days
    from datafusion_comet_amd import native
    plan = S.project(S.scan([S.T_DATE]), [S.cast(S.col(0, S.T_DATE), S.T_INT32)])
    native.compile_plan(plan.encode())
    t = pa.table({"d": pa.array([0, 19723, None, -5], pa.int32()).cast(pa.date32())})
    assert O.run_plan_to_arrow(S, plan, t).column(0).to_pylist() == [0, 19723, None, -5]


def test_the_references_hour_minute_second_vectors():
    """datetime_funcs/extract_date_part.rs:128-189: 2024-01-15 18:30:45 UTC — a TIMESTAMP shows hour 10 in a Los Angeles session, a TIMESTAMP_NTZ is a
    wall clock already: hour 18 / minute 30 / second 45 whatever the session zone (issue #3180 of the reference)"""
    MICROS = 1_705_343_445_000_000

    def part(kind, dtype, arrow_type, tz):
        plan = S.project(S.scan([dtype]), [S.time_part(kind, S.col(0, dtype), tz)])
        return O.run_plan_to_arrow(S, plan, pa.table({"t": pa.array([MICROS], arrow_type)})).column(0).to_pylist()[0]
    assert part("hour", TS, pa.timestamp("us", tz="UTC"), "America/Los_Angeles") == 10
    for tz in ("UTC", "America/Los_Angeles", "Asia/Tokyo"):
        assert part("hour", NTZ, pa.timestamp("us"), tz) == 18
    assert part("minute", NTZ, pa.timestamp("us"), "Asia/Tokyo") == 30 and part("second", NTZ, pa.timestamp("us"), "Asia/Tokyo") == 45
    assert part("hour", TS, pa.timestamp("us", tz="UTC"), "Asia/Tokyo") == 3          # (18:30 UTC is 03:30 the next day in Tokyo)


def test_more_of_the_references_cast_vectors():
    from decimal import Decimal

    def cast(arr, frm, to, mode=S.LEGACY):
        plan = S.project(S.scan([frm]), [S.cast(S.col(0, frm), to, mode)])
        return O.run_plan_to_arrow(S, plan, pa.table({"v": arr})).column(0)
    # numeric.rs:1243-1263 test_spark_cast_int_to_int_overflow: LEGACY keeps the low bits, ANSI raises
    assert cast(pa.array([2**63 - 1, -2**63, 100], pa.int64()), S.T_INT64, S.T_INT32).to_pylist() == [-1, 0, 100]
    with pytest.raises(O.OracleError, match="CAST_OVERFLOW"):
        cast(pa.array([2**63 - 1], pa.int64()), S.T_INT64, S.T_INT32, S.ANSI)
    # :1301-1317 int → decimal(10,2); :1319-1360 overflow is NULL in LEGACY and TRY; :1380-1404 an error under ANSI
    assert cast(pa.array([100, -100, None], pa.int32()), S.T_INT32, S.decimal(10, 2)).to_pylist() == [Decimal("100.00"), Decimal("-100.00"), None]
    for mode in (S.LEGACY, S.TRY):
        assert cast(pa.array([9, 1000, None, -9], pa.int32()), S.T_INT32, S.decimal(3, 2), mode).to_pylist() == [Decimal("9.00"), None, None, Decimal("-9.00")]
    with pytest.raises(O.OracleError, match="NUMERIC_VALUE_OUT_OF_RANGE"):
        cast(pa.array([9, 1000], pa.int32()), S.T_INT32, S.decimal(3, 2), S.ANSI)
    # boolean.rs:205-230: true is one microsecond, false the epoch
    for to in (TS, NTZ):
        assert cast(pa.array([True, False, None]), S.T_BOOL, to).cast(pa.int64()).to_pylist() == [1, 0, None]
    # boolean.rs:90-172: true / false as 1 / 0 in every numeric type
    for to, one in ((S.T_INT8, 1), (S.T_INT16, 1), (S.T_INT32, 1), (S.T_INT64, 1), (S.T_FLOAT, 1.0), (S.T_DOUBLE, 1.0)):
        assert cast(pa.array([True, False, None]), S.T_BOOL, to).to_pylist() == [one, 0, None]


def test_the_references_integer_round_vectors():
    """math_funcs/round.rs:381-460 — bigint rounded at positions beyond its nineteen digits: at -19 the halves ±5·10^18 round to ±10^19, which wraps
    to its low 64 bits in LEGACY (WRAPPED_1E19 = 10^19 as i64) and is an overflow under ANSI; at -20 and below everything is 0 in both modes.
    (The kernels refuse positions below -18 by name; these pin the oracle.)"""
    def rnd(vals, scale, **kw):
        plan = S.project(S.scan([S.T_INT64]), [S.scalar_func("round", [S.col(0, S.T_INT64), S.lit(scale, S.T_INT64)], S.T_INT64, **kw)])
        return O.run_plan_to_arrow(S, plan, pa.table({"v": pa.array(vals, pa.int64())})).column(0).to_pylist()
    wrapped = 10**19 - 2**64
    assert rnd([5 * 10**18, -5 * 10**18, 5 * 10**18 - 1, 0, 2**63 - 1, -2**63], -19) == [wrapped, -wrapped, 0, 0, wrapped, -wrapped]
    with pytest.raises(O.OracleError, match="ARITHMETIC_OVERFLOW"):
        rnd([5 * 10**18], -19, fail_on_error=True)
    for fail in (False, True):
        assert rnd([2**63 - 1, -2**63, 0, 10**18], -20, fail_on_error=fail) == [0, 0, 0, 0]
        assert rnd([2**63 - 1, -2**63, 0, 5 * 10**18], -40, fail_on_error=fail) == [0, 0, 0, 0]


def test_the_references_ceil_and_floor_vectors():
    """math_funcs/ceil.rs and floor.rs (their array tests): floats and doubles to bigint, bigints unchanged, decimal(5,2) to decimal(4,0)"""
    from decimal import Decimal

    def f(name, arr, ty, rt):
        plan = S.project(S.scan([ty]), [S.scalar_func(name, [S.col(0, ty)], rt)])
        return O.run_plan_to_arrow(S, plan, pa.table({"v": arr})).column(0).to_pylist()
    up, down = [125.2345, 15.0001, 0.1, -0.9, -1.1, 123.0], [125.9345, 15.9999, 0.9, -0.1, -1.999, 123.0]
    for ty, arr in ((S.T_DOUBLE, lambda v: pa.array(v, pa.float64())), (S.T_FLOAT, lambda v: pa.array(np.array(v, np.float32)))):
        assert f("ceil", arr(up), ty, S.T_INT64) == [126, 16, 1, 0, -1, 123]
        assert f("floor", arr(down), ty, S.T_INT64) == [125, 15, 0, -1, -2, 123]
    for name in ("ceil", "floor"):
        assert f(name, pa.array([-1, 0, 1, None], pa.int64()), S.T_INT64, S.T_INT64) == [-1, 0, 1, None]
    d = pa.array([Decimal("123.45"), Decimal("125.00"), Decimal("-129.99")], pa.decimal128(5, 2))
    assert f("ceil", d, S.decimal(5, 2), S.decimal(4, 0)) == [Decimal("124"), Decimal("125"), Decimal("-129")]
    assert f("floor", d, S.decimal(5, 2), S.decimal(4, 0)) == [Decimal("123"), Decimal("125"), Decimal("-130")]


def test_the_references_checked_arithmetic_vectors():
    """math_funcs/checked_arithmetic.rs (its six tests): NULLs propagate, an overflow is NULL in TRY mode and an error under ANSI, and a NULL row
    whose value slot holds garbage does not raise"""
    I32 = S.T_INT32

    def op(name, l, r, mode):
        plan = S.project(S.scan([I32, I32]), [S.math(name, S.col(0, I32), S.col(1, I32), I32, mode)])
        return O.run_plan_to_arrow(S, plan, pa.table({"l": l, "r": r})).column(0).to_pylist()
    a32 = lambda v: pa.array(v, pa.int32())
    mx, mn = 2**31 - 1, -2**31
    assert op("add", a32([1, None, 3, None]), a32([10, 20, None, None]), S.TRY) == [11, None, None, None]
    assert op("add", a32([mx, 1]), a32([1, 1]), S.TRY) == [None, 2]
    with pytest.raises(O.OracleError, match="ARITHMETIC_OVERFLOW"):
        op("add", a32([mx]), a32([1]), S.ANSI)
    assert op("subtract", a32([mn, 5]), a32([1, 3]), S.TRY) == [None, 2]
    assert op("multiply", a32([mx, 5]), a32([2, 3]), S.TRY) == [None, 15]
    # a NULL row with i32::MAX in its value slot: no error under ANSI
    garbage = pa.Array.from_buffers(pa.int32(), 2, [pa.py_buffer(bytes([0b10])), pa.py_buffer(np.array([mx, 1], np.int32).tobytes())])
    assert op("add", garbage, a32([1, 1]), S.ANSI) == [None, 2]


def test_the_references_abs_vectors():
    """math_funcs/abs.rs (its array tests): [-1, MIN, MAX, NULL] → [1, MIN, MAX, NULL] in LEGACY — the minimum wraps onto itself — and an
    ARITHMETIC_OVERFLOW with fail_on_error"""
    for ty, arrow, bits in ((S.T_INT8, pa.int8(), 8), (S.T_INT16, pa.int16(), 16), (S.T_INT32, pa.int32(), 32), (S.T_INT64, pa.int64(), 64)):
        mn, mx = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
        t = pa.table({"v": pa.array([-1, mn, mx, None], arrow)})
        legacy = S.project(S.scan([ty]), [S.scalar_func("abs", [S.col(0, ty), S.lit(False, S.T_BOOL)], ty)])
        assert O.run_plan_to_arrow(S, legacy, t).column(0).to_pylist() == [1, mn, mx, None]
        ansi = S.project(S.scan([ty]), [S.scalar_func("abs", [S.col(0, ty), S.lit(True, S.T_BOOL)], ty)])
        with pytest.raises(O.OracleError, match="ARITHMETIC_OVERFLOW"):
            O.run_plan_to_arrow(S, ansi, t)
        assert O.run_plan_to_arrow(S, ansi, pa.table({"v": pa.array([-1, mx, None], arrow)})).column(0).to_pylist() == [1, mx, None]


def test_the_references_date_diff_and_negative_vectors():
    """datetime_funcs/date_diff.rs (basic, and the i32 wrap of extreme inputs); math_funcs/negative.rs: LEGACY wraps the minimum onto itself, ANSI
    raises, and a minimum sitting in a NULL slot does not"""
    D = S.T_DATE

    def diff(end, start):
        plan = S.project(S.scan([D, D]), [S.scalar_func("date_diff", [S.col(0, D), S.col(1, D)], S.T_INT32)])
        t = pa.table({"e": pa.array([end], pa.int32()).cast(pa.date32()), "s": pa.array([start], pa.int32()).cast(pa.date32())})
        return O.run_plan_to_arrow(S, plan, t).column(0).to_pylist()[0]
    mx, mn = 2**31 - 1, -2**31
    assert diff(18263, 18262) == 1 and diff(18262, 18263) == -1
    assert diff(mx, mn) == -1 and diff(mn, mx) == 1          # i32::MAX.wrapping_sub(i32::MIN), i32::MIN.wrapping_sub(i32::MAX)
    for ty, arrow, bits in ((S.T_INT8, pa.int8(), 8), (S.T_INT16, pa.int16(), 16), (S.T_INT32, pa.int32(), 32), (S.T_INT64, pa.int64(), 64)):
        lo = -(1 << (bits - 1))
        neg = lambda fail: S.project(S.scan([ty]), [S.Expr("unary_minus", [S.col(0, ty)], fail_on_error=fail)])
        assert O.run_plan_to_arrow(S, neg(False), pa.table({"v": pa.array([lo, 7, None], arrow)})).column(0).to_pylist() == [lo, -7, None]
        with pytest.raises(O.OracleError, match="ARITHMETIC_OVERFLOW"):
            O.run_plan_to_arrow(S, neg(True), pa.table({"v": pa.array([lo], arrow)}))
        hidden = pa.Array.from_buffers(arrow, 2, [pa.py_buffer(bytes([0b10])), pa.py_buffer(np.array([lo, 7]).astype(arrow.to_pandas_dtype()).tobytes())])
        assert O.run_plan_to_arrow(S, neg(True), pa.table({"v": hidden})).column(0).to_pylist() == [None, -7]
