"""String → Float / Double: the device routine (csrc/device/strtod.hpp), compiled for the host, against correctly rounded conversions — Python's
float() for doubles, an exact rational rounding for floats (numpy parses through a double: rounded twice) — and the reference's parse rules
(string.rs:177-258: String.trim, inf / infinity / nan, one trailing d / D / f / F, Rust's float grammar)."""
import ctypes
import os
import random
import struct
import subprocess
from fractions import Fraction

import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "..", "datafusion-comet_amd", "csrc")


@pytest.fixture(scope="module")
def dev(tmp_path_factory):
    shim = """
typedef long long i64; typedef int i32; typedef unsigned int u32; typedef unsigned char u8; typedef unsigned long long u64;
#define CDEV static inline
#include "device/strtod.hpp"
extern "C" int t_parse(const u8* p, i32 n, int is32, u64* out) { return str_to_float_bits(p, n, is32 != 0, *out); }
"""
    d = tmp_path_factory.mktemp("strtod")
    (d / "s.cpp").write_text(shim)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-I", _CSRC, "-o", str(d / "libsd.so"), str(d / "s.cpp")])
    return ctypes.CDLL(str(d / "libsd.so"))


def _parse(dev, s, is32=False):
    b = s.encode()
    out = ctypes.c_uint64()
    rc = dev.t_parse(b, len(b), 1 if is32 else 0, ctypes.byref(out))
    return None if rc else out.value


def _bits64(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def _f32_bits_exact(text):
    """the float nearest to the decimal `text` (ties to even), by exact rational arithmetic"""
    q = Fraction(text)
    neg = q < 0
    q = abs(q)
    if q == 0:
        return 0x80000000 if text.strip().startswith("-") else 0
    e = q.numerator.bit_length() - q.denominator.bit_length()
    if Fraction(2) ** e > q:
        e -= 1
    e = max(e, -126)                                  # subnormals share the smallest exponent
    scaled = q / Fraction(2) ** (e - 23)                # the significand in units of the last place
    m = scaled.numerator // scaled.denominator
    rem = scaled - m
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and m & 1):
        m += 1
    if m == 1 << 24:
        m >>= 1
        e += 1
    if e > 127:
        bits = 0x7F800000
    elif m < (1 << 23):
        bits = m                                        # subnormal (or zero)
    else:
        bits = ((e + 127) << 23) | (m - (1 << 23))
    return bits | (0x80000000 if neg else 0)


def _numbers(rng, n):
    out = ["0", "-0", "1", "1.0", ".5", "5.", "0.1", "1e10", "1E-10", "123456.789", "1.7976931348623157e308", "1.7976931348623159e308", "1e309", "4.9e-324", "2.4703282292062327e-324",
           "2.4703282292062328e-324", "1e-400", "9007199254740993", "9007199254740992.5", "0.000000000000000000000000000000001", "1" + "0" * 400 + "e-400", "0." + "0" * 400 + "1e400",
           "8.5e-46", "7.006492321624085e-46", "7.006492321624086e-46", "3.4028235677973366e38", "3.4028235e38", "1.00000017881393432617187499", "1.00000017881393432617187501",
           "16777217", "16777216.99999", "2.2250738585072011e-308", "2.2250738585072014e-308", "179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365510704342711559699508093042880177904174497791.999"]
    while len(out) < n:
        k = rng.randrange(5)
        if k == 0:
            out.append(repr(struct.unpack("<d", struct.pack("<Q", rng.getrandbits(63) % 0x7FF0000000000000))[0]))
        elif k == 1:
            digits = "".join(rng.choice("0123456789") for _ in range(rng.randrange(1, 40)))
            i = rng.randrange(len(digits) + 1)
            out.append(digits[:i] + "." + digits[i:] if rng.random() < 0.7 else digits)
        elif k == 2:
            out.append("%d.%de%d" % (rng.randrange(10), rng.randrange(10**rng.randrange(1, 25)), rng.randrange(-340, 320)))
        elif k == 3:
            # a double's exact halfway point to its neighbour, and digits just around it
            x = struct.unpack("<d", struct.pack("<Q", rng.getrandbits(62) % 0x7FE0000000000000 + 1))[0]
            y = struct.unpack("<d", struct.pack("<Q", _bits64(x) + 1))[0]
            mid = (Fraction(x) + Fraction(y)) / 2
            from decimal import Decimal, getcontext
            getcontext().prec = 1200
            t = str(Decimal(mid.numerator) / Decimal(mid.denominator))
            out.append(t)
            if "E" not in t:
                out.append(t + "1")
        else:
            out.append(str(rng.randrange(-10**6, 10**6)) + rng.choice(["", ".0", ".5", "e2", "E-2", "e+3"]))
    return out


def test_doubles_are_correctly_rounded(dev):
    rng = random.Random(21)
    for t in _numbers(rng, 60_000):
        for s in (t, "-" + t if not t.startswith("-") else t[1:]):
            assert _parse(dev, s) == _bits64(float(s)), s


def test_floats_are_correctly_rounded_once(dev):
    rng = random.Random(22)
    for t in _numbers(rng, 8_000):
        if len(t) > 200:
            continue
        assert _parse(dev, t, True) == _f32_bits_exact(t), t


def test_the_references_parse_rules(dev):
    inf, ninf, nan = _bits64(float("inf")), _bits64(float("-inf")), 0x7FF8000000000000
    for s, want in [("inf", inf), ("+INF", inf), ("Infinity", inf), ("+infinity", inf), ("-inf", ninf), ("-Infinity", ninf), ("NaN", nan), ("nan", nan), (" 1.5 ", _bits64(1.5)), ("\t1.5\n", _bits64(1.5)),
                    ("\x001.5\x00", _bits64(1.5)), ("1.5d", _bits64(1.5)), ("1.5D", _bits64(1.5)), ("1.5f", _bits64(1.5)), ("1.5F", _bits64(1.5)), ("1e5f", _bits64(1e5)), ("infd", inf), ("nanF", nan), ("nand", nan),
                    ("-nan", nan | (1 << 63)), ("+nan", nan), ("1.", _bits64(1.0)), (".5", _bits64(0.5)), ("+.5e1", _bits64(5.0)), ("1e+2", _bits64(100.0)), ("0e999999999999", 0), ("1e999999999999", inf),
                    ("1e-999999999999", 0), ("-0", 1 << 63), ("-0.0f", 1 << 63)]:
        assert _parse(dev, s) == want, s
    for s in ["", " ", ".", "+", "-", "e5", "1e", "1e+", "1.5x", "1.5dd", "1.5 d", "d", "f", "1..5", "1.5.", "0x10", "1_000", "１２", "1.5\x7f", "\x7f1.5", "1 .5", "--1", "+-1", "infx", "in", "nanx", "1,5",
              "1e5.5", "1.5 f"]:
        assert _parse(dev, s) is None, repr(s)
