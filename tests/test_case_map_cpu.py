"""upper / lower: the device routine (csrc/device/case_map.hpp + the generated csrc/case_tables.hpp), compiled for the host, against
  * Rust's standard library itself for to_lowercase — the `tokenizers` wheel's Lowercase normalizer, every scalar value;
  * Python's str.lower() / str.upper() (Unicode 13: the same full case mapping, SpecialCasing's unconditional entries and the Final_Sigma rule)
    on every character Python knows and on strings that put Σ into every context the rule distinguishes."""
import ctypes
import os
import random
import subprocess
import unicodedata

import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "..", "datafusion-comet_amd", "csrc")


@pytest.fixture(scope="module")
def dev(tmp_path_factory):
    shim = """
typedef long long i64; typedef int i32; typedef unsigned int u32; typedef unsigned char u8;
#define CDEV static inline
#include "device/case_map.hpp"
extern "C" i32 t_map(const u8* p, i32 n, int mode, u8* o) { return case_map_value(p, n, mode, o); }
"""
    d = tmp_path_factory.mktemp("casemap")
    (d / "cm.cpp").write_text(shim)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-I", _CSRC, "-o", str(d / "libcm.so"), str(d / "cm.cpp")])
    return ctypes.CDLL(str(d / "libcm.so"))


def _map(dev, s, mode):
    b = s.encode()
    n = dev.t_map(b, len(b), mode, None)
    out = ctypes.create_string_buffer(n + 1)
    assert dev.t_map(b, len(b), mode, out) == n
    return out.raw[:n].decode()


def test_lowercase_of_every_scalar_value_is_rusts(dev):
    from tokenizers.normalizers import Lowercase
    low = Lowercase()
    for c in range(0x110000):
        if 0xD800 <= c <= 0xDFFF or c == 0x3A3:      # (Σ alone is not word-final in Rust's str::to_lowercase either: σ — checked below)
            continue
        ch = chr(c)
        want = low.normalize_str(ch)
        if want != ch or c < 0x3000:
            assert _map(dev, ch, 1) == want, hex(c)
    assert _map(dev, "Σ", 1) == "σ"


def test_python_agrees_on_every_character_it_knows(dev):
    for c in range(0x110000):
        if 0xD800 <= c <= 0xDFFF:
            continue
        ch = chr(c)
        if unicodedata.category(ch) == "Cn":
            continue
        if ch.upper() != ch or ch.lower() != ch or c < 0x3000:
            assert _map(dev, ch, 2) == ch.upper(), hex(c)
            assert _map(dev, ch, 1) == ch.lower(), hex(c)
    # the letters of Unicode 14-16 (beyond Python's tables): pairs that invert each other
    for lo, up in [(0x10D70, 0x10D50), (0x1C8A, 0x1C89), (0xA7CD, 0xA7CC), (0x10597, 0x10570)]:
        assert _map(dev, chr(up), 1) == chr(lo) and _map(dev, chr(lo), 2) == chr(up)


def test_strings_and_the_final_sigma_rule(dev):
    rng = random.Random(3)
    pool = ["Σ", "σ", "ς", "Α", "α", "a", "B", " ", ".", "'", "́", "­", "ß", "İ", "ı", "ǅ", "ﬁ", "ŉ", "日", "1", "-", "ΟΔΥΣΣΕΥΣ", "Straße", "ǰ", "ΐ", "ᾳ", "ᾼ", "😀"]
    fixed = ["ΟΔΥΣΣΕΥΣ", "Σ", "ΑΣ", "ΑΣ.", "ΑΣ'", "ΑΣ'Β", "ΑΣΒ", "Σ'", "'Σ", "Α'Σ", "ΆΣ́", "1Σ", "ΑΣ1", "ΑΣ Β", "Α.Σ", "straße", "İstanbul", "ǅemal", "ﬁn", "ŉ", "", "tschüß"]
    for s in fixed + ["".join(rng.choice(pool) for _ in range(rng.randrange(0, 8))) for _ in range(4000)]:
        assert _map(dev, s, 1) == s.lower(), repr(s)
        assert _map(dev, s, 2) == s.upper(), repr(s)
