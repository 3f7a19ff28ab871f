"""String casts: the oracle's restatement (oracle/strcast.py) against the reference's own unit-test vectors, and the DEVICE source of the
same casts (csrc/device/comet_device.hpp, the section between "string casts: begin / end") compiled for the host and compared with the oracle
on those vectors and on a seeded fuzz — a CPU-side check of the exact text the GPU pipelines JIT."""
import ctypes
import json
import os
import random
import subprocess

import pytest

from oracle import strcast as C

_HERE = os.path.dirname(os.path.abspath(__file__))
_HDR = os.path.join(_HERE, "..", "datafusion-comet_amd", "csrc", "device", "comet_device.hpp")
KATS = json.load(open(os.path.join(_HERE, "golden", "reference_kats.json")))["string_casts"]
MODES = (C.LEGACY, C.ANSI, C.TRY)
MODE_ID = {C.LEGACY: 0, C.ANSI: 1, C.TRY: 2}


def test_oracle_on_the_references_vectors():
    for s in KATS["date_ok_18262"]:
        for m in MODES:
            assert C.string_to_date(s.encode(), m) == (18262, None), (s, m)
    for s in KATS["date_invalid"]:
        for m in (C.LEGACY, C.TRY):
            assert C.string_to_date(s.encode(), m) == (None, None), (s, m)
        assert C.string_to_date(s.encode(), C.ANSI)[1] is not None, s
    for s in KATS["date_null_every_mode"]:
        for m in MODES:
            assert C.string_to_date(s.encode(), m) == (None, None), (s, m)
    for s, v in KATS["date_values"]:
        for m in MODES:
            assert C.string_to_date(s.encode(), m) == (v, None), (s, m)
    for s, m, want in KATS["i8"]:
        got = C.string_to_int(s.encode(), m, 8)
        assert (got[1] is not None) if want == "error" else got == (want, None), (s, m, got)
    for s, p, sc, want in KATS["decimal_boundary"]:
        got = C.string_to_decimal(s.encode(), p, sc, C.ANSI)
        assert (got[1] is not None) == (want == "error"), (s, got)
    for u, sc, want in KATS["java_string"]:
        assert C.decimal_to_string(int(u), sc, C.LEGACY) == want


def test_oracle_trim_regimes():
    """test_cast_string_to_{boolean,int,date,float_and_decimal}_trim_parity (string.rs:2205-2308): trimAll drops 0x00-0x20 and 0x7F, String.trim
    keeps 0x7F; neither drops non-ASCII whitespace."""
    for pad in [" ", "\t", "\n", "\r", "\x0b", "\x0c", "\x00", "\x1f", "\x7f"]:
        for s in (pad + "1", "1" + pad, pad + "1" + pad):
            assert C.string_to_bool(s.encode(), C.LEGACY) == (True, None)
            assert C.string_to_int(s.encode(), C.LEGACY, 32) == (1, None)
            want = (None, None) if pad == "\x7f" else (1, None)
            assert C.string_to_decimal(s.encode(), 5, 0, C.LEGACY) == want, repr(s)
        assert C.string_to_date((pad + "2020-01-01" + pad).encode(), C.LEGACY) == (18262, None)
    for pad in [" ", " ", "　"]:
        assert C.string_to_int((pad + "1").encode(), C.LEGACY, 32) == (None, None)
        assert C.string_to_decimal((pad + "1").encode(), 5, 0, C.LEGACY) == (None, None)
    assert C.string_to_decimal("１２.５".encode(), 5, 1, C.LEGACY) == (125, None)      # fullwidth digits (string.rs:472-495)


@pytest.fixture(scope="module")
def dev(tmp_path_factory):
    src = open(_HDR).read()
    a = src.index("// ---- string casts: begin")
    b = src.index("// ---- string casts: end")
    shim = """
#include <stdint.h>
#include <string.h>
typedef long long i64; typedef unsigned long long u64; typedef int i32; typedef unsigned int u32; typedef short i16; typedef signed char i8;
typedef unsigned char u8; typedef __int128 i128; typedef unsigned __int128 u128;
#define CDEV static inline
#define COMET_GLOBAL
""" + src[a:b] + """
extern "C" {
int t_bool(const u8* p, i32 n, int* out) { bool o = false; int rc = str_to_bool(p, n, o); *out = o; return rc; }
int t_int(const u8* p, i32 n, int mode, int bits, i64* out) { return str_to_int(p, n, mode, bits, *out); }
int t_dec(const u8* p, i32 n, int precision, int scale, u64* lo, i64* hi) { i128 v = 0; int rc = str_to_decimal(p, n, precision, scale, v); *lo = (u64)v; *hi = (i64)(v >> 64); return rc; }
int t_date(const u8* p, i32 n, i32* out) { return str_to_date(p, n, *out); }
i32 f_i64(i64 v, u8* o) { return fmt_i64(v, o); }
i32 f_bool(int v, u8* o) { return fmt_bool(v != 0, o); }
i32 f_dec(u64 lo, i64 hi, int scale, int java, u8* o) { return fmt_decimal((i128)(((u128)(u64)hi << 64) | lo), scale, java != 0, o); }
i32 f_date(i64 d, u8* o) { return fmt_date(d, o); }
i32 f_ts(i64 us, i64 off, u8* o) { return fmt_timestamp(us, off, o); }
}
"""
    d = tmp_path_factory.mktemp("strcast")
    c = d / "dev_strcast.cpp"
    c.write_text(shim)
    so = d / "libdevstrcast.so"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-I", os.path.dirname(_HDR), "-o", str(so), str(c)])      # (-I: the section includes dates.hpp)
    return ctypes.CDLL(str(so))


def _dev_parse(dev, kind, b, mode, *a):
    """(value, error) in the oracle's vocabulary from the device function's return code."""
    n = len(b)
    if kind == "bool":
        o = ctypes.c_int()
        rc = dev.t_bool(b, n, ctypes.byref(o))
        v = bool(o.value)
    elif kind == "int":
        o = ctypes.c_int64()
        rc = dev.t_int(b, n, MODE_ID[mode], a[0], ctypes.byref(o))
        v = o.value
    elif kind == "dec":
        lo, hi = ctypes.c_uint64(), ctypes.c_int64()
        rc = dev.t_dec(b, n, a[0], a[1], ctypes.byref(lo), ctypes.byref(hi))
        v = (hi.value << 64) | lo.value
    else:
        o = ctypes.c_int32()
        rc = dev.t_date(b, n, ctypes.byref(o))
        v = o.value
    if rc == 0:
        return v, None
    if rc == 2 or mode != C.ANSI:
        return None, None
    return None, (C.NUMERIC_OUT_OF_RANGE if rc == 3 else C.CAST_INVALID)


def _fmt(dev, fn, *a):
    buf = ctypes.create_string_buffer(64)
    n = fn(*a, buf)
    assert 0 < n <= 48
    return buf.raw[:n].decode()


def _strings(rng):
    """values around what the parsers accept: numbers, dates, words, with the padding, signs, dots, exponents and junk that decide the edge cases"""
    pads = ["", " ", "\t", "\n ", "\x00", "\x7f", " ", "\x1f\r"]
    words = ["true", "FALSE", "t", "Y", "yes", "no", "N", "1", "0", "tru", "inf", "-Infinity", "NaN", "+inf", "nan ", "", "+", "-", ".", "-.", "+.5", "e5", "1e", "1e+", "1E-3", "--1", "1-", "1.2.3"]
    out = []
    for _ in range(6000):
        k = rng.randrange(8)
        if k == 0:
            s = rng.choice(words)
        elif k == 1:
            s = str(rng.choice([0, 1, -1, 127, 128, -128, -129, 32767, 32768, -32768, 2**31 - 1, 2**31, -2**31, -2**31 - 1, 2**63 - 1, 2**63, -2**63, -2**63 - 1, 10**18, rng.randrange(-10**20, 10**20)]))
            if rng.random() < 0.3:
                s = "+" + s if not s.startswith("-") else s
            if rng.random() < 0.3:
                s += "." + "".join(rng.choice("0123456789x") for _ in range(rng.randrange(4)))
        elif k == 2:
            s = "".join(rng.choice("0123456789") for _ in range(rng.randrange(1, 45)))
            if rng.random() < 0.7:
                i = rng.randrange(len(s) + 1)
                s = s[:i] + "." + s[i:]
            if rng.random() < 0.4:
                s += rng.choice("eE") + rng.choice(["", "+", "-"]) + str(rng.choice([0, 1, 5, 37, 38, 39, 40, 77, 2**31 - 1, 2**31, 10**12]))
            if rng.random() < 0.5:
                s = rng.choice("+-") + s
        elif k == 3:
            y = rng.choice([0, 1, 4, 100, 1582, 1900, 1970, 2000, 2020, 2024, 9999, 10000, 262142, 262143, 5881580, 5881581, 9999999, rng.randrange(0, 300000)])
            s = rng.choice(["", "", "-", "+"]) + str(y).zfill(rng.choice([4, 4, 5, 7, 8]))
            if rng.random() < 0.9:
                s += "-" + str(rng.randrange(0, 14)).zfill(rng.choice([1, 2, 3]))
                if rng.random() < 0.9:
                    s += "-" + str(rng.randrange(0, 33)).zfill(rng.choice([1, 2, 3]))
                    s += rng.choice(["", "", "T", " ", "T12:00", " x", "x"])
        elif k == 4:
            s = "".join(rng.choice("0123456789-+.eE TtrueFALSyn\x00 ０１９") for _ in range(rng.randrange(0, 14)))
        elif k == 5:
            s = "".join(rng.choice("０１２３４５６７８９") for _ in range(rng.randrange(1, 6))) + rng.choice(["", ".５", "e２", "x"])
        elif k == 6:
            s = "%04d-%02d-%02d" % (rng.randrange(0, 10000), rng.randrange(0, 14), rng.randrange(0, 33))
        else:
            s = str(rng.randrange(-10**6, 10**6)) + rng.choice(["", ".0", ".5", ".49999", "e2", "E-2"])
        out.append((rng.choice(pads) + s + rng.choice(pads)).encode())
    return out


def test_device_source_parsers_agree_with_the_oracle(dev):
    rng = random.Random(20260924)
    vals = _strings(rng) + [s.encode() for k in ("date_ok_18262", "date_invalid", "date_null_every_mode") for s in KATS[k]] + [s.encode() for s, _ in KATS["date_values"]]
    for b in vals:
        for m in MODES:
            assert _dev_parse(dev, "bool", b, m) == C.string_to_bool(b, m), (b, m)
            for bits in (8, 16, 32, 64):
                assert _dev_parse(dev, "int", b, m, bits) == C.string_to_int(b, m, bits), (b, m, bits)
            for p, s in ((5, 0), (10, 2), (18, 6), (38, 0), (38, 10), (38, 38), (3, 3), (20, 19)):
                assert _dev_parse(dev, "dec", b, m, p, s) == C.string_to_decimal(b, p, s, m), (b, m, p, s)
            assert _dev_parse(dev, "date", b, m) == C.string_to_date(b, m), (b, m)


def test_device_source_formatters_agree_with_the_oracle(dev):
    rng = random.Random(7)
    for u, sc, want in KATS["java_string"]:
        u = int(u)
        assert _fmt(dev, dev.f_dec, ctypes.c_uint64(u & (2**64 - 1)), ctypes.c_int64(u >> 64), sc, 1) == want
    ints = [0, 1, -1, 9, 10, -10, 2**31 - 1, -2**31, 2**63 - 1, -2**63, 10**18, -10**18] + [rng.randrange(-2**63, 2**63) for _ in range(500)]
    for v in ints:
        assert _fmt(dev, dev.f_i64, ctypes.c_int64(v)) == C.int_to_string(v)
    assert _fmt(dev, dev.f_bool, 1) == "true" and _fmt(dev, dev.f_bool, 0) == "false"
    for _ in range(3000):
        nd = rng.randrange(1, 39)
        u = rng.randrange(-10**nd, 10**nd)
        sc = rng.randrange(0, 39)
        for java in (0, 1):
            want = C.decimal_to_string(u, sc, C.LEGACY if java else C.TRY)
            assert _fmt(dev, dev.f_dec, ctypes.c_uint64(u & (2**64 - 1)), ctypes.c_int64(u >> 64), sc, java) == want, (u, sc, java)
    days = [0, -1, 1, 18262, -719528, -719529, 2932896, 2932897, -2**31, 2**31 - 1, 11016, 11017] + [rng.randrange(-10**6, 4 * 10**6) for _ in range(2000)]
    for d in days:
        assert _fmt(dev, dev.f_date, ctypes.c_int64(d)) == C.date_to_string(d), d
    for _ in range(3000):
        us = rng.choice([0, 1, -1, 10**6, 1_500_000, 86_400_000_000 - 1, rng.randrange(-2**62, 2**62), rng.randrange(-10**16, 10**16), rng.randrange(0, 4 * 10**15) // 1000 * 1000])
        off = rng.choice([0, 0, 3600, -8 * 3600, 19800, -34200])
        assert _fmt(dev, dev.f_ts, ctypes.c_int64(us), ctypes.c_int64(off)) == C.timestamp_to_string(us, off), (us, off)


def test_date_format_agrees_with_python_in_the_common_era():
    import datetime
    for d in (0, 18262, -719162, 2932896, 11016):
        assert C.date_to_string(d) == (datetime.date(1970, 1, 1) + datetime.timedelta(days=d)).isoformat()
