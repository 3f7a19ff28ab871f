"""ScanExec casts a stream column whose Arrow type differs from the declared Scan field (operators/scan.rs:281-291: arrow's cast_with_options
with the default, safe CastOptions — what the target cannot hold becomes NULL).  pyarrow.compute.cast(safe=False) plus explicit range checks is
the independent statement of those semantics here; arrow-rs and Arrow C++ agree on this subset (integer widths, floats, Date64, timestamp units,
decimal precision / scale, LargeUtf8), except that arrow-rs NULLs where C++ would wrap or raise."""
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu


def scan_as(table, declared):
    plan = S.project(S.scan(declared), [S.col(i, t) for i, t in enumerate(declared)])
    out = native.execute_to_table([native.HostInput.from_table(table, 1000)], len(declared), plan.encode(), batch_size=0)
    return pa.Table.from_batches(out)


def test_integer_and_float_width_casts(built):
    rng = np.random.default_rng(1)
    n = 5000
    big = rng.integers(-2**40, 2**40, n)
    small = rng.integers(-100, 100, n)
    t = pa.table({"i64_to_i32": pa.array(big, pa.int64(), mask=rng.random(n) < 0.1), "i8_to_i64": pa.array(small, pa.int8()),
                  "u32_to_i32": pa.array(rng.integers(0, 2**32, n), pa.uint32()), "f64_to_i32": pa.array(rng.standard_normal(n) * 3e9),
                  "i32_to_f64": pa.array(rng.integers(-2**31, 2**31, n), pa.int32()), "f32_to_f64": pa.array(rng.standard_normal(n).astype(np.float32)),
                  "f64_to_f32": pa.array(rng.standard_normal(n))})
    got = scan_as(t, [S.T_INT32, S.T_INT64, S.T_INT32, S.T_INT32, S.T_DOUBLE, S.T_DOUBLE, S.T_FLOAT])

    def narrowed(col, lo, hi, trunc=False):
        out = []
        for v in col.to_pylist():
            if v is None or v != v:
                out.append(None)
                continue
            w = int(v) if trunc else v            # int() truncates toward zero
            out.append(w if lo <= w <= hi else None)
        return out
    assert got.column(0).to_pylist() == narrowed(t.column(0), -2**31, 2**31 - 1)
    assert got.column(1).to_pylist() == t.column(1).to_pylist()
    assert got.column(2).to_pylist() == narrowed(t.column(2), -2**31, 2**31 - 1)
    assert got.column(3).to_pylist() == narrowed(t.column(3), -2**31, 2**31 - 1, trunc=True)
    assert got.column(4).to_pylist() == [float(v) for v in t.column(4).to_pylist()]
    assert got.column(5).to_pylist() == t.column(5).to_pylist()
    assert got.column(6).to_pylist() == t.column(6).cast(pa.float32()).to_pylist()
    assert [f.type for f in got.schema] == [pa.int32(), pa.int64(), pa.int32(), pa.int32(), pa.float64(), pa.float64(), pa.float32()]


def test_temporal_decimal_and_large_string_casts(built):
    ms = [0, 1, -1, 86_400_000, -86_400_001, 1_600_000_000_123]
    ns = [0, 999, -999, 1_000, -1_001, 1_600_000_000_123_456_789]
    sec = [0, 1, -1, 2**62, -(2**62), 1_600_000_000]
    t = pa.table({"ms": pa.array(ms, pa.timestamp("ms", tz="UTC")), "ns": pa.array(ns, pa.timestamp("ns")), "s": pa.array(sec, pa.timestamp("s")),
                  "d64": pa.array(ms, pa.date64()),
                  "dec": pa.array([Decimal("1.005"), Decimal("-1.005"), Decimal("99999.994"), Decimal("99999.995"), None, Decimal("0.004")], pa.decimal128(10, 3)),
                  "up": pa.array([Decimal("12.5"), Decimal("-0.1"), Decimal("9999999.9"), None, Decimal("0.0"), Decimal("1.0")], pa.decimal128(8, 1)),
                  "ls": pa.array(["", "a", None, "héllo", "x" * 50, "end"], pa.large_utf8())})
    got = scan_as(t, [S.T_TIMESTAMP, S.T_TIMESTAMP, S.T_TIMESTAMP, S.T_DATE, S.decimal(7, 2), S.decimal(12, 4), S.T_STRING])
    us = lambda c: c.combine_chunks().cast(pa.int64()).to_pylist()
    assert us(got.column(0)) == [v * 1000 for v in ms]
    assert us(got.column(1)) == [int(v / 1000) for v in ns]                               # truncating division, like arrow's unit cast
    assert us(got.column(2)) == [v * 1_000_000 if abs(v * 1_000_000) < 2**63 else None for v in sec]   # overflow → NULL
    assert got.column(3).cast(pa.int32()).to_pylist() == [int(v / 86_400_000) for v in ms]
    # scale 3 → 2 rounds half away from zero; decimal(7,2) holds up to 99999.99
    assert got.column(4).to_pylist() == [Decimal("1.01"), Decimal("-1.01"), Decimal("99999.99"), None, None, Decimal("0.00")]
    assert got.column(5).to_pylist() == [Decimal("12.5000"), Decimal("-0.1000"), Decimal("9999999.9000"), None, Decimal("0.0000"), Decimal("1.0000")]
    assert got.column(6).to_pylist() == t.column(6).to_pylist() and got.schema.field(6).type == pa.utf8()


def test_uncastable_pairs_are_still_an_error(built):
    t = pa.table({"s": pa.array(["1", "2"])})
    with pytest.raises(native.CometNativeException, match="Arrow format 'u'"):
        scan_as(t, [S.T_INT64])
