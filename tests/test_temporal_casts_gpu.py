"""Casts between dates, timestamps and integers, timestamps written out as strings, and hour / minute / second — in region time zones (SURVEY
§8 a6; conversion_funcs/temporal.rs:37-78, cast.rs:395-415, utils.rs:62-330, datetime_funcs/extract_date_part.rs): the zone's table is a
constant of the fused kernel (csrc/tz.cpp), checked on the CPU against zoneinfo and the reference's vectors (tests/test_time_zones_cpu.py)."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
TS, NTZ, D, I64, STR = S.T_TIMESTAMP, S.DataType(S.TIMESTAMP_NTZ), S.T_DATE, S.T_INT64, S.T_STRING
ZONES = ["UTC", "America/Los_Angeles", "Asia/Kolkata", "Europe/London", "Australia/Lord_Howe", "Pacific/Apia", "+05:30"]


def _table(n, seed):
    rng = np.random.default_rng(seed)
    us = rng.integers(-5 * 10**15, 6 * 10**15, n, dtype=np.int64)          # 1811 … 2160
    # around the daylight-saving switches of 2024 in the zones above, to the second
    edges = np.array([1710064800, 1710068400, 1730624400, 1730628000, 1711846800, 1729990800, 1712419200, 1728142200, 1325239200, 1325325600], np.int64) * 1_000_000
    k = len(edges) * 5
    us[:k] = np.repeat(edges, 5) + np.tile(np.array([-1_000_001, -1, 0, 1, 1_800_000_000]), len(edges))
    days = rng.integers(-60_000, 70_000, n).astype(np.int32)
    days[:6] = [0, 19723, 19793, 19800, 20029, 15337]
    secs = rng.integers(-10**10, 10**10, n)
    secs[:4] = [0, -1, 2**62, -2**62]
    m = lambda: rng.random(n) < 0.05
    return pa.table({"ts": pa.array(us, pa.timestamp("us", tz="UTC"), mask=m()), "ntz": pa.array(us, pa.timestamp("us"), mask=m()),
                     "d": pa.array(days, pa.int32(), mask=m()).cast(pa.date32()), "s": pa.array(secs, pa.int64(), mask=m()), "b": pa.array(rng.random(n) < 0.5)})


FIELDS = [TS, NTZ, D, I64, S.T_BOOL]


def _check(exprs, t):
    from oracle import oracle as O
    plan = S.project(S.scan(FIELDS), exprs)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], len(exprs), plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, t)
    for i in range(len(exprs)):
        g, w = got.column(i), want.column(i)
        if pa.types.is_timestamp(g.type) or pa.types.is_date(g.type):
            g, w = g.cast(pa.int64() if pa.types.is_timestamp(g.type) else pa.int32()), w.cast(pa.int64() if pa.types.is_timestamp(w.type) else pa.int32())
        gl, wl = g.to_pylist(), w.to_pylist()
        if gl != wl:
            k = next(j for j in range(len(gl)) if gl[j] != wl[j])
            raise AssertionError(f"output {i}, row {k}: got {gl[k]!r}, want {wl[k]!r}, input {[c[k].as_py() for c in t.columns]!r}")


@pytest.mark.parametrize("tz", ZONES)
def test_casts_and_time_parts_in_a_time_zone(built, tz):
    t = _table(20_000, 33)
    ts, ntz, d, s, b = (S.col(i, ty) for i, ty in enumerate(FIELDS))
    c = lambda x, to: S.cast(x, to, S.LEGACY, tz)
    _check([c(ts, D), c(ntz, D), c(d, TS), c(d, NTZ), c(ts, I64), c(ntz, I64), c(s, TS), c(b, TS), c(ts, NTZ), c(ntz, TS), c(ts, STR), c(ntz, STR),
            S.time_part("hour", ts, tz), S.time_part("minute", ts, tz), S.time_part("second", ts, tz), S.time_part("hour", ntz, tz),
            S.cast(c(ts, D), STR), c(c(d, TS), STR)], t)


def test_the_references_date_to_timestamp_vectors(built):
    """temporal.rs test_cast_date_to_timestamp"""
    t = pa.table({"ts": pa.array([0, 0, 0], pa.timestamp("us", tz="UTC")), "ntz": pa.array([0, 0, 0], pa.timestamp("us")), "d": pa.array([0, 19723, 19793], pa.int32()).cast(pa.date32()),
                  "s": pa.array([0, 0, 0], pa.int64()), "b": pa.array([True, False, True])})
    non_dst, dst = 1704067200000000, 1710115200000000
    for zone, want in [("UTC", [0, non_dst, dst]), ("America/Los_Angeles", [28800000000, non_dst + 28800000000, dst + 25200000000]),
                       ("America/Phoenix", [25200000000, non_dst + 25200000000, dst + 25200000000])]:
        plan = S.project(S.scan(FIELDS), [S.cast(S.col(2, D), TS, S.LEGACY, zone)])
        got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 1, plan.encode(), batch_size=0))
        assert got.column(0).cast(pa.int64()).to_pylist() == want, zone


def test_unknown_zones_and_instants_behind_the_tables_end(built):
    t = _table(64, 1)
    with pytest.raises(native.CometNativeException, match="Mars/Olympus"):
        native.compile_plan(S.project(S.scan(FIELDS), [S.cast(S.col(0, TS), D, S.LEGACY, "Mars/Olympus")]).encode())
    # behind a zone table's end the last rule goes on (the calendar repeats every 400 years: comet_device.hpp tz_fold) — the years 2413, 9999, 150000
    rng = np.random.default_rng(4)
    us = np.concatenate([rng.integers(14_000_000_000, 14_100_000_000, 24), rng.integers(253_300_000_000, 253_400_000_000, 20), rng.integers(4_670_000_000_000, 4_680_000_000_000, 20)]) * 1_000_000 + 123
    far = t.set_column(0, "ts", pa.array(us, pa.timestamp("us", tz="UTC"))).set_column(1, "ntz", pa.array(us, pa.timestamp("us")))
    ts, ntz = S.col(0, TS), S.col(1, NTZ)
    for tz in ("Europe/Berlin", "America/Los_Angeles", "Australia/Sydney", "Asia/Tokyo"):
        c = lambda x, to: S.cast(x, to, S.LEGACY, tz)
        _check([c(ts, D), c(ts, NTZ), c(ntz, TS), c(ts, STR), S.time_part("hour", ts, tz)], far)


def test_floats_and_decimals_to_timestamps(built):
    """cast_float_to_timestamp / cast_decimal_to_timestamp (numeric.rs:87-135, 1184-1233; tests/test_reference_vectors_cpu.py pins the oracle on
    the reference's vectors): seconds → microseconds, NaN / ±Infinity / beyond a bigint → NULL (ANSI: the reference's two errors), decimals
    truncated toward zero and wrapped to 64 bits"""
    import json
    from datafusion_comet_amd.tpch import _dec128_array
    from oracle import oracle as O
    rng = np.random.default_rng(12)
    n = 4000
    f = np.concatenate([[0.0, 1.0, -1.0, 1.5, 0.000001, float("nan"), float("inf"), float("-inf"), 1e13, -1e13, 9.3e12, 1e303, 9223372036854.775, -9223372036854.775, -0.0],
                        rng.normal(0, 1e9, n), rng.normal(0, 1e13, n // 4), rng.random(n // 4) * 1e-5])
    m = len(f)
    lo = np.concatenate([[0, 1_000_000, -1_000_000, 1_500_000, 123_456_789, 199, -199], rng.integers(-10**17, 10**17, m - 7)]).astype(np.int64)
    with np.errstate(over="ignore", invalid="ignore"):
        f32 = f.astype(np.float32)
    mask = rng.random(m) < 0.05
    big = [int(x) * 10**15 + 7 for x in lo]
    wide = pa.Array.from_buffers(pa.decimal128(38, 0), m, [None, pa.py_buffer(np.array([[v & (2**64 - 1), (v >> 64) & (2**64 - 1)] for v in big], np.uint64).tobytes())])
    t = pa.table({"f": pa.array(f, mask=mask), "g": pa.array(f32, mask=mask), "d6": _dec128_array(lo, 18, 6), "d2": _dec128_array(lo, 18, 2), "d8": _dec128_array(lo, 18, 8), "w": wide})
    fields = [S.T_DOUBLE, S.T_FLOAT, S.decimal(18, 6), S.decimal(18, 2), S.decimal(18, 8), S.decimal(38, 0)]
    cols = [S.col(k, ty) for k, ty in enumerate(fields)]
    for mode in (S.LEGACY, S.TRY):
        outs = [S.cast(c, to, mode, "America/Los_Angeles") for c in cols for to in (TS, NTZ)]
        plan = S.project(S.scan(fields), outs)
        got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], len(outs), plan.encode(), batch_size=0))
        want = O.run_plan_to_arrow(S, plan, t)
        for k in range(len(outs)):
            g, w = got.column(k).cast(pa.int64()).to_pylist(), want.column(k).cast(pa.int64()).to_pylist()
            if g != w:
                j = next(x for x in range(m) if g[x] != w[x])
                raise AssertionError(f"output {k}, row {j}: got {g[j]!r}, want {w[j]!r}, input {[c[j].as_py() for c in t.columns]!r}")
        assert got.column(0).cast(pa.int64()).to_pylist()[:5] == [0, 1_000_000, -1_000_000, 1_500_000, 1]          # numeric.rs:1815-1832
    for bad, etype, params in [(float("nan"), "CastInvalidValue", {"value": "NaN", "fromType": "DOUBLE", "toType": "TIMESTAMP"}),
                               (float("-inf"), "CastInvalidValue", {"value": "-inf", "fromType": "DOUBLE", "toType": "TIMESTAMP"}),
                               (1e13, "CastOverFlow", {"value": "1E19D", "fromType": "DOUBLE", "toType": "BIGINT"}),
                               (1e303, "CastOverFlow", {"value": "Infinity", "fromType": "DOUBLE", "toType": "BIGINT"})]:
        tb = pa.table({"f": pa.array([1.0, bad, None])})
        with pytest.raises(native.CometQueryExecutionException) as ei:
            native.execute_to_table([native.HostInput.from_table(tb)], 1, S.project(S.scan([S.T_DOUBLE]), [S.cast(S.col(0, S.T_DOUBLE), TS, S.ANSI)]).encode(), batch_size=0)
        j = json.loads(str(ei.value))
        assert j["errorType"] == etype and j["params"] == params, j


def test_cast_date_as_int(built):
    """cast.rs:273-277 `(Date32, Int32)`: the days since the epoch, reinterpreted — every mode, NULLs kept, extremes of the type included; and as an
    operand of later arithmetic and of a filter (the value must be the day number there too)."""
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    n = 50_001
    days = rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64, endpoint=True).astype(np.int32)
    days[:7] = [0, -1, 1, 19723, -719162, 2**31 - 1, -2**31]
    mask = rng.random(n) < 0.1
    mask[:7] = False
    t = pa.table({"d": pa.array(days, pa.int32(), mask=mask).cast(pa.date32())})
    d = S.col(0, D)
    outs = [S.cast(d, S.T_INT32, mode) for mode in (S.LEGACY, S.TRY, S.ANSI)]
    outs.append(S.math("add", S.cast(S.cast(d, S.T_INT32), I64), S.lit(1, I64), I64))
    plan = S.project(S.scan([D]), outs)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], len(outs), plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, t)
    for k in range(len(outs)):
        assert got.column(k).combine_chunks().equals(want.column(k).combine_chunks()), k
    assert got.column(0).to_pylist()[:7] == [0, -1, 1, 19723, -719162, 2**31 - 1, -2**31]
    fplan = S.project(S.filter_(S.scan([D]), S.gt(S.cast(d, S.T_INT32), S.lit(19000, S.T_INT32))), [S.cast(d, S.T_INT32)])
    g2 = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 1, fplan.encode(), batch_size=0))
    w2 = O.run_plan_to_arrow(S, fplan, t)
    assert g2.num_rows == w2.num_rows > 0 and g2.column(0).combine_chunks().equals(w2.column(0).combine_chunks())
