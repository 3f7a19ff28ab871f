"""The multi-kernel snappy pipeline (csrc/device/snappy2.hpp: window transfer functions → chunk functions → page chain → element list →
pointer jumping per 64 KiB fragment) run on the CPU: the same phases the gfx950 kernels call, with the threads of a workgroup looped
(tests/emu/snappy2_emu.cpp).  Streams: pyarrow's snappy (the Google C++ library, 64 KiB blocks — what Parquet writers emit) over the page
shapes a scan meets, several pages per call; hand-built streams for the corners (long literals that end deep inside a chunk, 4-byte
offsets, self-overlapping copies, chains of copies thousands deep); streams that are legal but not fragment-shaped must come back FLAGGED
for the one-wave kernel, corrupt ones as errors — never as wrong bytes."""
import ctypes
import os
import subprocess

import numpy as np
import pyarrow as pa
import pytest

from tests.test_snappy_emu_cpu import build, varint, literal, copy, reference      # noqa: F401  (stream builders)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ST_OK, ST_FALLBACK, ST_ERR = 0, 1, 16


@pytest.fixture(scope="module")
def emu2(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("snappy2_emu") / "libsnappy2_emu.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "datafusion-comet_amd", "csrc"),
                    os.path.join(ROOT, "tests", "emu", "snappy2_emu.cpp"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.sn2_emu_inflate_pages.restype = ctypes.c_int64

    def inflate(streams, page_lens):
        n = len(streams)
        slen = np.array([len(s) for s in streams], np.int32)
        soff = np.zeros(n, np.int64)
        soff[1:] = np.cumsum(slen[:-1], dtype=np.int64)
        blob = np.frombuffer(b"".join(streams) + b"\0", np.uint8)
        plen = np.array(page_lens, np.int32)
        ooff = np.zeros(n, np.int64)
        ooff[1:] = np.cumsum(plen[:-1], dtype=np.int64)
        out = np.zeros(int(plen.sum()) + 1, np.uint8)
        status = np.zeros(n, np.uint32)
        rounds = ctypes.c_int32(0)
        lib.sn2_emu_inflate_pages(ctypes.c_void_p(blob.ctypes.data), ctypes.c_void_p(soff.ctypes.data), ctypes.c_void_p(slen.ctypes.data),
                                  ctypes.c_void_p(plen.ctypes.data), ctypes.c_int32(n), ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ooff.ctypes.data),
                                  ctypes.c_void_p(status.ctypes.data), ctypes.byref(rounds))
        return [int(s) for s in status], [out[int(o):int(o) + int(l)].tobytes() for o, l in zip(ooff, plen)], rounds.value
    return inflate


def test_pages_as_the_scan_sees_them_several_per_call(emu2):
    rng = np.random.default_rng(5)
    pages = [
        b"", b"a", b"hello hello hello hello hello hello", bytes(70_000), b"abcdefg" * 9000,
        rng.integers(90_000, 10_000_000, 140_000).astype(np.int64).tobytes(),          # 1.1 MB of decimal(12,2)-as-INT64: 18 fragments, deep chains
        rng.integers(0, 50, 40_000).astype(np.int32).tobytes(),
        rng.standard_normal(20_000).tobytes(),                                           # incompressible: 64 KiB literals
        " ".join(rng.choice(["alpha", "beta", "gamma", "lineitem", "orders", "MI355X"], 20_000)).encode(),
        b"".join([rng.integers(0, 256, 70_000, dtype=np.uint8).tobytes(), b"xyz" * 1000, rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()]),
        (np.arange(200_000, dtype=np.int64) * 1000).tobytes(),                           # sorted keys: long copies at offset 8
    ]
    streams = [pa.compress(p, codec="snappy", asbytes=True) for p in pages]
    status, got, rounds = emu2(streams, [len(p) for p in pages])
    assert status == [ST_OK] * len(pages)
    for g, p in zip(got, pages):
        assert g == p
    assert 1 <= rounds <= 17          # (the looped threads of the emulation see each other's updates inside a round: fewer rounds than log2 of the chain depth)


def test_hand_built_corners(emu2):
    rng = np.random.default_rng(6)
    noise = lambda n: ("lit", rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    cases = []
    # a chain of copies thousands deep inside ONE fragment, copies overlapping their own output, runs of one byte
    e = [("lit", b"0123456789abcdef")]
    for k in range(6000):
        e.append(("copy", (4 + k % 8, 1 + k % 13)))
        if k % 5 == 0:
            e.append(("lit", bytes([k & 0xFF, (k * 7) & 0xFF])))
    cases.append(build(e))
    # literals longer than a window / a chunk that end deep inside a chunk (kernel B's slow path), then small elements up to the chunk's end
    e = [noise(5000), ("copy", (10, 77)), noise(61), noise(3), ("copy", (64, 5000)), noise(300)] + [("copy", (5, 9)), noise(2)] * 900
    cases.append(build(e))
    # 4-byte offsets (wide), literal length encodings of 1, 2 and 3 bytes
    e = [noise(60), noise(61), noise(256), noise(257), noise(65_536 - 60 - 61 - 256 - 257), noise(100), ("copy", (64, 90)), ("copy", (11, 100))]
    cases.append(build(e, wide=True))
    # exactly one fragment, and one byte more
    cases.append(build([noise(65_536)]))
    cases.append(build([noise(65_536), noise(1)]))
    streams = [c[0] for c in cases]
    status, got, _ = emu2(streams, [len(c[1]) for c in cases])
    assert status == [ST_OK] * len(cases)
    for g, (_, raw) in zip(got, cases):
        assert g == raw


def test_legal_streams_that_are_not_fragment_shaped_are_flagged_for_the_fallback(emu2):
    rng = np.random.default_rng(8)
    noise = lambda n: ("lit", rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    crossing_literal = build([noise(65_000), noise(1000), noise(10)])                 # an element straddles the 64 KiB boundary
    crossing_copy = build([noise(65_530), ("copy", (20, 100)), noise(10)])
    far_copy = build([noise(65_536), noise(100), ("copy", (30, 2000)), noise(5)])      # reads the previous fragment
    good = build([noise(500), ("copy", (40, 100))])
    cases = [crossing_literal, crossing_copy, far_copy, good]
    status, got, _ = emu2([c[0] for c in cases], [len(c[1]) for c in cases])
    assert status == [ST_FALLBACK, ST_FALLBACK, ST_FALLBACK, ST_OK]
    assert got[3] == good[1]


def test_corrupt_streams_are_errors(emu2):
    raw = np.random.default_rng(7).integers(0, 1000, 50_000).astype(np.int64).tobytes()
    good = pa.compress(raw, codec="snappy", asbytes=True)
    bad_len = varint(len(raw) + 1) + good[len(varint(len(raw))):]
    truncated = good[:-7]
    zero_offset = build([("lit", b"abcdefgh")])[0][:-0 or None]
    zero_offset = varint(12) + literal(b"abcdefgh") + bytes([1 | (0 << 2), 0])          # copy of 4 bytes with offset 0
    before_start = varint(12) + literal(b"abcdefgh") + copy(4, 9)                         # reaches before the page's first byte
    overrun = varint(10) + literal(b"abcdefgh") + copy(8, 2)                              # produces more than declared
    streams = [bad_len, truncated, zero_offset, before_start, overrun, good]
    lens = [len(raw) + 1, len(raw), 12, 12, 10, len(raw)]
    status, got, _ = emu2(streams, lens)
    assert all(s >= ST_ERR for s in status[:5]), status
    assert status[5] == ST_OK and got[5] == raw


def test_random_streams_against_the_slow_reference(emu2):
    """random element soups (literals of every length class, copies of every form) cut into fragment-shaped pieces"""
    rng = np.random.default_rng(11)
    cases = []
    for _ in range(12):
        elems, frag_fill, total = [], 0, 0
        while total < 200_000:
            room = 65_536 - frag_fill
            if rng.random() < 0.45 or frag_fill == 0:
                n = int(min(room, rng.choice([1, 2, 5, 59, 60, 61, 255, 256, 257, 4000])))
                elems.append(("lit", rng.integers(0, 256, n, dtype=np.uint8).tobytes()))
            else:
                n = int(min(room, rng.integers(1, 65)))
                off = int(rng.integers(1, min(frag_fill, 65_535) + 1))
                elems.append(("copy", (n, off)))
            frag_fill = (frag_fill + n) % 65_536
            total += n
        cases.append(build(elems, wide=bool(rng.random() < 0.3)))
    status, got, _ = emu2([c[0] for c in cases], [len(c[1]) for c in cases])
    assert status == [ST_OK] * len(cases)
    for g, (_, raw) in zip(got, cases):
        assert g == raw
