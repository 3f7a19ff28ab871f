"""The device snappy decompressor (csrc/device/snappy_inflate.hpp) run on the CPU: the SAME source the gfx950 kernel compiles, driven by a
64-lane host emulation (tests/emu/snappy_emu.cpp) in which a lane sees another lane's writes only across a wave primitive.  Streams come
from pyarrow's snappy (the Google C++ library) plus hand-built ones for what that compressor never emits: copies reaching back more than the
64 KiB LDS history ring, 4-byte-offset copies, and corrupt input."""
import ctypes
import os
import subprocess

import numpy as np
import pyarrow as pa
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("snappy_emu") / "libsnappy_emu.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "datafusion-comet_amd", "csrc"),
                    os.path.join(ROOT, "tests", "emu", "snappy_emu.cpp"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.emu_snappy_inflate.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib.emu_snappy_inflate.restype = ctypes.c_int

    def inflate(stream: bytes, n: int):
        out = np.zeros(max(n, 1), np.uint8)
        rc = lib.emu_snappy_inflate(stream, len(stream), out.ctypes.data, n)
        return rc, out[:n].tobytes()
    return inflate


def varint(n):
    out = b""
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


def literal(data: bytes):
    n = len(data) - 1
    if n < 60:
        return bytes([n << 2]) + data
    nb = (n.bit_length() + 7) // 8
    return bytes([(59 + nb) << 2]) + n.to_bytes(nb, "little") + data


def copy(length, offset, wide=False):
    if not wide and 4 <= length <= 11 and offset < 2048:
        return bytes([1 | ((length - 4) << 2) | ((offset >> 8) << 5), offset & 0xFF])
    if not wide and offset < 65536:
        return bytes([2 | ((length - 1) << 2)]) + offset.to_bytes(2, "little")
    return bytes([3 | ((length - 1) << 2)]) + offset.to_bytes(4, "little")


def reference(stream_elems):
    """apply (kind, payload) elements the slow way"""
    out = bytearray()
    for kind, v in stream_elems:
        if kind == "lit":
            out += v
        else:
            length, offset = v
            for _ in range(length):
                out.append(out[-offset])
    return bytes(out)


def build(stream_elems, wide=False):
    body = b"".join(literal(v) if k == "lit" else copy(v[0], v[1], wide) for k, v in stream_elems)
    raw = reference(stream_elems)
    return varint(len(raw)) + body, raw


CASES = {
    "empty": b"",
    "one byte": b"a",
    "short text": b"hello hello hello hello hello hello",
    "zeros": bytes(70_000),
    "period 7": b"abcdefg" * 9000,
}


@pytest.mark.parametrize("name", list(CASES))
def test_small_streams(emu, name):
    raw = CASES[name]
    rc, got = emu(pa.compress(raw, codec="snappy", asbytes=True), len(raw))
    assert rc == 0 and got == raw


def test_columnar_pages_like_the_scan_sees(emu):
    """PLAIN pages of 8-byte decimals (2-byte literal + 6-byte copy per value, offsets up to the block size), 4-byte ints, doubles
    (incompressible: 64 KiB literals), dictionary-like text"""
    rng = np.random.default_rng(5)
    pages = [
        rng.integers(90_000, 10_000_000, 20_000).astype(np.int64).tobytes(),
        rng.integers(0, 50, 40_000).astype(np.int32).tobytes(),
        rng.standard_normal(20_000).tobytes(),
        " ".join(rng.choice(["alpha", "beta", "gamma", "lineitem", "orders", "MI355X"], 20_000)).encode(),
        b"".join([rng.integers(0, 256, 70_000, dtype=np.uint8).tobytes(), b"xyz" * 1000, rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()]),
    ]
    for raw in pages:
        stream = pa.compress(raw, codec="snappy", asbytes=True)
        rc, got = emu(stream, len(raw))
        assert rc == 0 and got == raw, (len(raw), len(stream), rc)


def test_copies_beyond_the_history_ring_and_wide_offsets(emu):
    rng = np.random.default_rng(6)
    noise = lambda n: ("lit", rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    elems = [noise(40_000), noise(50_000), noise(33),
             ("copy", (64, 90_000)),          # further back than the 64 KiB ring: read from the flushed output
             ("copy", (11, 60_000)),          # inside the ring's span but beyond its safe part
             noise(5), ("copy", (20, 3)),     # overlaps its own output (period 3)
             ("copy", (64, 1)),               # run of one byte
             noise(70_000), ("copy", (64, 150_000)), ("copy", (7, 70_064)), noise(1)]
    for wide in (False, True):
        stream, raw = build(elems, wide)
        rc, got = emu(stream, len(raw))
        assert rc == 0 and got == raw


def test_dependency_chains_inside_one_window(emu):
    """every copy reads what the element before it produced — the rounds degenerate to one element each and must still be exact"""
    elems = [("lit", b"0123456789abcdef")]
    for k in range(300):
        elems.append(("copy", (4 + k % 8, 1 + k % 13)))
        if k % 5 == 0:
            elems.append(("lit", bytes([k & 0xFF, (k * 7) & 0xFF])))
    stream, raw = build(elems)
    rc, got = emu(stream, len(raw))
    assert rc == 0 and got == raw


def test_corrupt_streams_are_reported_not_followed(emu):
    raw = np.random.default_rng(7).integers(0, 1000, 5000).astype(np.int64).tobytes()
    good = pa.compress(raw, codec="snappy", asbytes=True)
    assert emu(good, len(raw))[0] == 0
    assert emu(good, len(raw) + 1)[0] == 1                        # length in the preamble differs from the page header's
    assert emu(good[:-3], len(raw))[0] in (2, 5)                  # truncated
    assert emu(good + b"\x00a", len(raw))[0] == 4                 # trailing element overruns the page
    bad_off, _ = build([("lit", b"abcd"), ("copy", (4, 4))])
    broken = bytearray(bad_off)
    broken[-1] = 9                                                # offset 9 with only 4 bytes written
    assert emu(bytes(broken), 8)[0] == 3
    zero = bytearray(bad_off)
    zero[-1] = 0
    assert emu(bytes(zero), 8)[0] == 3                            # offset 0
    assert emu(b"\x80\x80\x80\x80\x80\x80", 5)[0] == 1            # endless varint
