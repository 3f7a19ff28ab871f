"""String → Timestamp / Timestamp_NTZ: the oracle's restatement (oracle/strcast.py string_to_timestamp*) against the reference's own vectors
(tests/golden/timestamp_kats.json, transcribed from string.rs' test module by tools/extract_timestamp_kats.py), and the DEVICE routine
(csrc/device/strts.hpp) compiled for the host against the oracle on those vectors and on a seeded fuzz over the shapes, separators, zone suffixes,
signs and junk that decide the edge cases, in three zones."""
import ctypes
import json
import os
import random
import subprocess

import pytest

from datafusion_comet_amd import native
from oracle import strcast as C

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "..", "datafusion-comet_amd", "csrc")
KATS = json.load(open(os.path.join(_HERE, "golden", "timestamp_kats.json")))["vectors"]
NOW = 1_700_000_000_000_000


def _check(got, exp, what):
    if exp == "err":
        assert got[1] is not None, what
    elif exp == "some":
        assert got[0] is not None and got[1] is None, what
    elif exp in ("none", None):
        assert got == (None, None), what
    else:
        assert got == (exp, None), what


def test_oracle_on_the_references_vectors():
    for f, val, mode, a4, a5, exp in KATS:
        if f == "timestamp_parser":
            got = C.string_to_timestamp(val.encode(), mode, a4, a5, now_us=NOW)
        else:
            got = C.string_to_timestamp_ntz(val.encode(), mode, a4 == "true")
        _check(got, exp, (f, val, mode, a4, a5))


@pytest.fixture(scope="module")
def dev(tmp_path_factory):
    src = open(os.path.join(_CSRC, "device", "comet_device.hpp")).read()
    sc = src[src.index("// ---- string casts: begin"):src.index("// ---- string casts: end")]
    tz = src[src.index("// ---- time zones: begin"):src.index("// ---- time zones: end")]
    shim = """
#include <stdint.h>
typedef long long i64; typedef unsigned long long u64; typedef int i32; typedef unsigned int u32; typedef unsigned char u8; typedef __int128 i128; typedef unsigned __int128 u128;
#define CDEV static inline
#define COMET_GLOBAL
""" + sc + tz + """
#include "device/strts.hpp"
extern "C" {
int t_ts(const u8* p, i32 n, const i64* zt, int spark4, i64 now_us, i64* out) { return str_to_timestamp(p, n, zt, spark4 != 0, now_us, *out); }
int t_ntz(const u8* p, i32 n, i64* out) { return str_to_timestamp_ntz(p, n, *out); }
}
"""
    d = tmp_path_factory.mktemp("strts")
    (d / "t.cpp").write_text(shim)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-I", _CSRC, "-I", os.path.join(_CSRC, "device"), "-o", str(d / "libts.so"), str(d / "t.cpp")])
    return ctypes.CDLL(str(d / "libts.so"))


_TABLES = {}


def _dev_ts(dev, b, mode, tz, spark4):
    if tz not in _TABLES:
        _TABLES[tz] = native.zone_table(tz)
    t = _TABLES[tz]
    out = ctypes.c_int64()
    rc = dev.t_ts(b, len(b), t.ctypes.data_as(ctypes.c_void_p), 1 if spark4 else 0, ctypes.c_int64(NOW), ctypes.byref(out))
    if rc == 0:
        return out.value, None
    if rc == 1:
        return None, (C.CAST_INVALID if mode == C.ANSI else None)
    return ("unsupported" if rc == 4 else "beyond" if rc == 5 else None), None


def _dev_ntz(dev, b, mode):
    out = ctypes.c_int64()
    rc = dev.t_ntz(b, len(b), ctypes.byref(out))
    if rc == 0:
        return out.value, None
    if rc == 1:
        return None, (C.CAST_INVALID if mode == C.ANSI else None)
    return ("unsupported" if rc == 4 else None), None


def test_device_source_on_the_references_vectors(dev):
    for f, val, mode, a4, a5, exp in KATS:
        if f == "timestamp_parser":
            got = _dev_ts(dev, val.encode(), mode, a4, a5)
            if got[0] == "unsupported":          # a NAMED zone inside the value: the device holds the session zone's table only — refused, not guessed
                assert "/" in val.split(" ")[-1], val
                continue
            _check(got, exp, (f, val, mode, a4, a5))
        elif a4 == "true":
            _check(_dev_ntz(dev, val.encode(), mode), exp, (f, val, mode))


def _values(rng, n):
    years = ["2020", "0100", "10000", "1970", "2024", "9999", "262142", "262143", "294247", "-290308", "-0001", "0000", "2021", "202", "20200", "0002020", "12345678", "1582", "1900", "２０２０", "٢٠٢٠"]
    out = ["", " ", "T2", "T2:30", " T2", "\tT2:30", "12:34", "12:34:56.7", "+12:12:12", "0119704", "2024001", "invalid", "2020-01-01T12:34:56.123456", "2020-03-08 02:30:00", "2020-11-01 01:30:00",
           "2024-03-10 02:30:00", "2023-02-29 00:00:00", "294247-01-10T04:00:54.775807Z", "-290308-12-21T19:59:05.224192Z", "294247-01-10T04:00:54.775808Z", "2020-01-01T12:34:56 Europe/Moscow",
           "2020-01-01T12:34:56 Mars/Olympus", "2020-01-01 12:34:56　", " 2020-01-01", "2020-01-01T25:00:00", "2020-01-01T12:60:00", "2020-13-01", "2020-02-30T00:00:00", "2011-12-30 10:00:00"]
    sfx = ["", "", "", "Z", " UTC", "UTC", "UTC+0", " UTC+07:30", "GMT-8", " GMT", "UT", " UT+1", "+05:30", "-08:00", "+0530", "-1:0", "+8:", "+19:00", "+5", "-05", " EST", "MST", " HST", "+05:60", " +08:00",
           "+00:00:00", "Zz", " America/New_York", "-20:0", "UTC-", "+", "-"]
    while len(out) < n:
        y = rng.choice(years)
        s = rng.choice(["", "", "", "-", "+"]) + y
        k = rng.randrange(8)
        if k >= 1:
            s += "-" + rng.choice(["01", "02", "12", "13", "00", "1", "011", "٠١"])
        if k >= 2:
            s += "-" + rng.choice(["01", "28", "29", "30", "31", "32", "00", "1", "٠١"])
        if k >= 3:
            s += rng.choice(["T", " ", "t", "  "]) + rng.choice(["00", "12", "23", "24", "2", "7", "012", "１２"])
        if k >= 4:
            s += ":" + rng.choice(["00", "34", "59", "60", "5"])
        if k >= 5:
            s += ":" + rng.choice(["00", "56", "59", "60", "6"])
        if k >= 6:
            s += "." + rng.choice(["1", "12", "123", "123456", "1234567", "123456789012", "000001", "٣", "5x"])
        if k == 7:
            s = rng.choice(["T", "", "T"]) + rng.choice(["1", "12", "24", "123"]) + rng.choice(["", ":3", ":34", ":60"]) + rng.choice(["", ":5", ":56", ":56.789", ":56.1234567"])
        s += rng.choice(sfx)
        out.append(rng.choice(["", "", "", " ", "\t", " "]) + s + rng.choice(["", "", " ", "\n", "　"]))
    return out


@pytest.mark.parametrize("tz", ["UTC", "America/New_York", "Asia/Kolkata", "Pacific/Apia", "+05:30"])
def test_device_source_agrees_with_the_oracle(dev, tz):
    rng = random.Random(hash(tz) & 0xFFFF)
    some = 0
    for v in _values(rng, 5000):
        b = v.encode()
        for mode in (C.LEGACY, C.ANSI):
            for spark4 in (True, False):
                got = _dev_ts(dev, b, mode, tz, spark4)
                if got[0] in ("unsupported", "beyond"):
                    assert "/" in v.strip().split(" ")[-1] or got[0] == "beyond", v      # a named zone in the value: refused, not guessed
                    continue
                want = C.string_to_timestamp(b, mode, tz, spark4, now_us=NOW)
                assert got == want, (v, mode, tz, spark4, got, want)
                some += got[0] is not None
            g = _dev_ntz(dev, b, mode)
            if g[0] != "unsupported":
                assert g == C.string_to_timestamp_ntz(b, mode, True), (v, mode)
    assert some > 1000


def test_what_the_kernel_refuses_without_a_today(dev):
    """the product passes no "today" (codegen.cpp cast_from_string): a time-only value is rc 6, a zone name inside a value rc 4, and the values the GPU
    parity run keeps (tests/test_string_casts_gpu.py _timestamp_strings) are answered, never refused"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sc_gpu", os.path.join(_HERE, "test_string_casts_gpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    t = native.zone_table("America/New_York")
    out = ctypes.c_int64()

    def rc(s, spark4=False):
        b = s.encode()
        return dev.t_ts(b, len(b), t.ctypes.data_as(ctypes.c_void_p), 1 if spark4 else 0, ctypes.c_int64(-(1 << 63)), ctypes.byref(out))
    assert rc("T12:34") == 6 and rc("12:34:56") == 6 and rc("2020-01-01T12:34:56 Europe/Moscow") == 4
    for s in set(m._timestamp_strings(20_000, 21).column(0).to_pylist()) - {None}:
        assert rc(s) in (0, 1, 2, 5) and rc(s, True) in (0, 1, 2, 5), s
