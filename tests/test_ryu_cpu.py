"""Float → String and Float → Decimal: the device routine (csrc/device/ryu.hpp + the generated csrc/ryu_tables.hpp), compiled for the host.
Shortest digits against Python's repr() (doubles: David Gay's shortest mode) and numpy's unique formatting (floats: Dragon4) — every algorithm
that returns the shortest decimal reading back as the same float, closest to the exact value, returns the same digits, and that is what the
`ryu` crate the reference links (numeric.rs:970) and Rust's Display (numeric.rs:182, 207) do.  Then the reference's formatting rules
(numeric.rs:137-221) and its Decimal(Double.toString(d)).setScale(HALF_UP) semantics (numeric.rs:955-990) with its own examples."""
import ctypes
import decimal
import os
import random
import struct
import subprocess

import numpy as np
import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "..", "datafusion-comet_amd", "csrc")


@pytest.fixture(scope="module")
def dev(tmp_path_factory):
    shim = """
typedef long long i64; typedef int i32; typedef unsigned int u32; typedef unsigned char u8; typedef unsigned long long u64; typedef __int128 i128; typedef unsigned __int128 u128;
#define CDEV static inline
#include "device/ryu.hpp"
extern "C" {
void t_d2d(u64 bits, u64* mant, i32* exp) { RyuDec r = ryu_d2d(bits); *mant = r.mant; *exp = r.exp; }
void t_f2d(u32 bits, u64* mant, i32* exp) { RyuDec r = ryu_f2d(bits); *mant = r.mant; *exp = r.exp; }
i32 t_fmt64(u64 bits, u8* o) { return fmt_f64_bits(bits, o); }
i32 t_fmt32(u32 bits, u8* o) { return fmt_f32_bits(bits, o); }
int t_dec(u64 bits, int p, int s, u64* lo, i64* hi) { i128 v = 0; int rc = f64_bits_to_decimal(bits, p, s, v); *lo = (u64)v; *hi = (i64)(v >> 64); return rc; }
}
"""
    d = tmp_path_factory.mktemp("ryu")
    (d / "r.cpp").write_text(shim)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-I", _CSRC, "-o", str(d / "libryu.so"), str(d / "r.cpp")])
    return ctypes.CDLL(str(d / "libryu.so"))


def _bits64(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def _digits(text):
    t = decimal.Decimal(text).as_tuple()
    m = int("".join(map(str, t.digits)))
    e = t.exponent
    while m and m % 10 == 0:
        m //= 10
        e += 1
    return m, e


def _doubles(rng, n):
    out = [5e-324, 2.2250738585072014e-308, 2.225073858507201e-308, 1.7976931348623157e308, 1.0, 0.1, 0.3, 1e22, 1e23, 9007199254740993.0, 4.35, 0.5153125, 1e-7, 123456789.0, 2.0**-1074 * 3,
           9.5367431640625e-07, 1e21, 5e-324 * 2, 0.001, 9999999.999999998, 1e7, 1.2345678901234567e-300]
    out += [2.0**k for k in range(-1074, 1024, 7)] + [float(10**k) for k in range(0, 23)] + [10.0**-k for k in range(1, 40, 3)]
    while len(out) < n:
        k = rng.randrange(4)
        if k == 0:
            b = rng.getrandbits(63)
            if (b >> 52) == 0x7FF or b == 0:
                continue
            out.append(struct.unpack("<d", struct.pack("<Q", b))[0])
        elif k == 1:
            out.append(rng.randrange(1, 10**rng.randrange(1, 18)) / 10**rng.randrange(0, 12))
        elif k == 2:
            out.append(float(rng.randrange(1, 2**53)))
        else:
            out.append(rng.random() * 10**rng.randrange(-10, 10))
    return out


def test_shortest_digits_of_doubles(dev):
    rng = random.Random(11)
    m, e = ctypes.c_uint64(), ctypes.c_int32()
    for x in _doubles(rng, 200_000):
        dev.t_d2d(ctypes.c_uint64(_bits64(x)), ctypes.byref(m), ctypes.byref(e))
        assert (m.value, e.value) == _digits(repr(x)), x


def test_shortest_digits_of_floats(dev):
    rng = np.random.default_rng(12)
    bits = np.concatenate([rng.integers(1, 0x7F800000, 120_000, dtype=np.uint32),
                           (np.arange(1, 255, dtype=np.uint32) << 23),                 # every power of two (the interval below is half as wide there)
                           (np.arange(1, 255, dtype=np.uint32) << 23) + 1, (np.arange(1, 255, dtype=np.uint32) << 23) - 1,
                           np.array([1, 2, 3, 0x7FFFFF, 0x800000, 0x7F7FFFFF, 0x3F800000, 0x3DCCCCCD, 0x4B189680, 0x3A83126F], np.uint32)])
    vals = bits.view(np.float32)
    m, e = ctypes.c_uint64(), ctypes.c_int32()
    for b, v in zip(bits.tolist(), vals):
        dev.t_f2d(ctypes.c_uint32(b), ctypes.byref(m), ctypes.byref(e))
        assert (m.value, e.value) == _digits(np.format_float_scientific(v, unique=True)), (hex(b), float(v))


def _java_like(x, is32):
    """the reference's rules (numeric.rs:137-221) over Python's / numpy's shortest digits"""
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Infinity" if x > 0 else "-Infinity"
    neg = np.signbit(x)
    a = abs(float(x))
    if a == 0:
        return "-0.0" if neg else "0.0"
    tiny = float(np.float32(1.4e-45)) if is32 else 5e-324
    if a == tiny:
        return ("-" if neg else "") + ("1.4E-45" if is32 else "4.9E-324")
    m, e = _digits(np.format_float_scientific(np.float32(x), unique=True) if is32 else repr(a))
    d = str(m)
    n = len(d)
    lo, hi = (float(np.float32(0.001)), float(np.float32(1e7))) if is32 else (0.001, 1e7)
    if lo <= a < hi:
        point = n + e
        if point <= 0:
            s = "0." + "0" * -point + d
        elif point >= n:
            s = d + "0" * (point - n) + ".0"
        else:
            s = d[:point] + "." + d[point:]
    else:
        s = d[0] + "." + (d[1:] if n > 1 else "0") + "E" + str(e + n - 1)
    return ("-" if neg else "") + s


def test_float_to_string(dev):
    """spot values Java / Spark print (Double.toString), then the rules over random values"""
    buf = ctypes.create_string_buffer(40)
    def f64(x):
        n = dev.t_fmt64(ctypes.c_uint64(_bits64(x)), buf)
        return buf.raw[:n].decode()

    def f32(x):
        n = dev.t_fmt32(ctypes.c_uint32(int(np.float32(x).view(np.uint32))), buf)
        return buf.raw[:n].decode()
    for x, want in [(0.0, "0.0"), (-0.0, "-0.0"), (1.0, "1.0"), (100.0, "100.0"), (1.5, "1.5"), (0.001, "0.001"), (1e-4, "1.0E-4"), (1.234e-5, "1.234E-5"), (1e7, "1.0E7"), (9999999.0, "9999999.0"),
                    (1.2345678e7, "1.2345678E7"), (1e21, "1.0E21"), (-2.5e-10, "-2.5E-10"), (5e-324, "4.9E-324"), (-5e-324, "-4.9E-324"), (float("nan"), "NaN"), (float("inf"), "Infinity"),
                    (float("-inf"), "-Infinity"), (1.7976931348623157e308, "1.7976931348623157E308"), (123456.789, "123456.789"), (0.1 + 0.2, "0.30000000000000004")]:
        assert f64(x) == want, x
    for x, want in [(1.0, "1.0"), (0.1, "0.1"), (1e7, "1.0E7"), (3.4028235e38, "3.4028235E38"), (1.4e-45, "1.4E-45"), (16777216.0, "1.6777216E7"), (0.001, "0.001"), (9.999e-4, "9.999E-4"), (-1.5, "-1.5")]:
        assert f32(x) == want, x
    rng = random.Random(13)
    for x in _doubles(rng, 30_000):
        for v in (x, -x):
            assert f64(v) == _java_like(v, False), v
    fb = np.random.default_rng(14).integers(1, 0x7F800000, 30_000, dtype=np.uint32)
    for v in fb.view(np.float32):
        assert f32(v) == _java_like(v, True), float(v)


def test_float_to_decimal(dev):
    """numeric.rs:955-990 float_to_decimal128: the SHORTEST string form is rounded HALF_UP (0.5153125 at scale 6 is 0.515313, though the binary
    value lies just below the tie), NaN / infinity → NULL, results beyond the precision → NULL"""
    lo, hi = ctypes.c_uint64(), ctypes.c_int64()

    def dec(x, p, s):
        rc = dev.t_dec(ctypes.c_uint64(_bits64(x)), p, s, ctypes.byref(lo), ctypes.byref(hi))
        return rc, (hi.value << 64) | lo.value
    assert dec(0.5153125, 10, 6) == (0, 515313)
    assert dec(-0.5153125, 10, 6)[1] == -515313
    assert dec(float("nan"), 10, 2)[0] == 2 and dec(float("inf"), 10, 2)[0] == 2
    assert dec(123.456, 5, 2) == (0, 12346) and dec(1234.56, 5, 2)[0] == 3
    assert dec(0.0, 5, 2) == (0, 0) and dec(1e-40, 38, 18) == (0, 0)
    rng = random.Random(15)
    decimal.getcontext().prec = 100
    for x in _doubles(rng, 40_000):
        for p, s in ((38, 18), (18, 2), (10, 0), (38, 0), (20, 10)):
            for v in (x, -x):
                want = decimal.Decimal(repr(v)).scaleb(s).quantize(decimal.Decimal(1), rounding=decimal.ROUND_HALF_UP) if abs(v) < 1e60 else None
                rc, got = dec(v, p, s)
                if want is None or abs(int(want)) >= 10**p:
                    assert rc == 3, (v, p, s)
                else:
                    assert (rc, got) == (0, int(want)), (v, p, s)
