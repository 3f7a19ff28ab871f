"""The scan's host page codecs (csrc/parquet_meta.cpp, comet_page_decompress) against pyarrow's codecs — no GPU needed.  The snappy decoder
is hand-written (fixed-size fast paths for the tiny elements columnar pages compress to), so it gets the same streams as the device
kernel's emulation: pyarrow-compressed pages of every shape plus hand-built ones (overlapping copies, 8 <= offset < 16 copies that read
what the first half of the move just wrote, 4-byte offsets) and corrupt input."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native
from tests.test_snappy_emu_cpu import build

SNAPPY, GZIP, ZSTD, LZ4_RAW = 1, 2, 6, 7


def pages():
    rng = np.random.default_rng(11)
    return [b"", b"a", b"hello hello hello hello hello hello", bytes(70_000), b"abcdefg" * 9000, b"0123456789" * 5000,
            rng.integers(90_000, 10_000_000, 131_072).astype(np.int64).tobytes(), rng.integers(0, 50, 40_000).astype(np.int32).tobytes(),
            rng.standard_normal(50_000).tobytes(), " ".join(rng.choice(["alpha", "beta", "gamma", "lineitem", "orders"], 50_000)).encode(),
            rng.integers(0, 4, 100_000, dtype=np.uint8).tobytes()]


@pytest.mark.parametrize("codec,name", [(SNAPPY, "snappy"), (GZIP, "gzip"), (ZSTD, "zstd"), (LZ4_RAW, "lz4_raw")])
def test_codecs_round_trip_pyarrow_pages(built, codec, name):
    for raw in pages():
        if not raw and name != "snappy":
            continue
        comp = pa.compress(raw, codec=name, asbytes=True)
        assert native.page_decompress(codec, comp, len(raw)) == raw, (name, len(raw))


def test_snappy_hand_built_streams(built):
    rng = np.random.default_rng(12)
    noise = lambda n: ("lit", rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    elems = [noise(40), ("copy", (16, 8)), ("copy", (16, 9)), ("copy", (13, 15)), ("copy", (16, 16)), ("copy", (12, 7)), ("copy", (4, 1)),
             noise(3), ("copy", (64, 3)), noise(16), noise(17), noise(61), noise(70_000), ("copy", (64, 69_000)), ("copy", (11, 12)), noise(1)]
    for k in range(200):
        elems.append(("copy", (4 + k % 13, 1 + k % 19)))
        if k % 3 == 0:
            elems.append(noise(1 + k % 18))
    for wide in (False, True):
        stream, raw = build(elems, wide)
        assert native.page_decompress(SNAPPY, stream, len(raw)) == raw


def test_snappy_corrupt_streams(built):
    raw = np.random.default_rng(13).integers(0, 1000, 5000).astype(np.int64).tobytes()
    good = pa.compress(raw, codec="snappy", asbytes=True)
    for stream, n in [(good, len(raw) + 1), (good[:-3], len(raw)), (good + b"\x00a", len(raw)), (b"\x80\x80\x80\x80\x80\x80\x80", 5)]:
        with pytest.raises(native.CometNativeException, match="snappy"):
            native.page_decompress(SNAPPY, stream, n)
    bad = bytearray(build([("lit", b"abcd"), ("copy", (4, 4))])[0])
    bad[-1] = 9
    with pytest.raises(native.CometNativeException, match="bad copy"):
        native.page_decompress(SNAPPY, bytes(bad), 8)


@pytest.mark.parametrize("codec,name", [(SNAPPY, "snappy"), (GZIP, "gzip"), (ZSTD, "zstd"), (LZ4_RAW, "lz4_raw")])
def test_corrupted_page_bodies_fail_cleanly(built, codec, name):
    """page bodies come from files: overwritten, truncated and extended compressed bytes, and a wrong declared size, give an exception or
    exactly `uncompressed_size` bytes — the hand-written snappy / LZ4 decoders never write past the output or read past the input"""
    import random
    rng = random.Random(40 + codec)
    srcs = [pages()[2], pages()[4][:20_000], pages()[7][:30_000], pages()[9][:25_000]]
    outcomes = {"ok": 0, "error": 0}
    for raw in srcs:
        comp = pa.compress(raw, codec=name, asbytes=True) if name != "lz4_raw" else pa.Codec("lz4_raw").compress(raw, asbytes=True)
        for trial in range(150):
            bad = bytearray(comp)
            k = trial % 4
            size = len(raw)
            if k == 0:
                bad = bad[: rng.randrange(0, len(bad))]
            elif k == 1:
                size = max(0, len(raw) + rng.choice([-1, 1, -100, 100, -len(raw) // 2]))
            else:
                for _ in range(rng.randrange(1, 4)):
                    bad[rng.randrange(0, len(bad))] = rng.choice([0, 0xFF, 0x80, rng.randrange(256)])
            # straight through the C entry, with canaries behind the output: nothing may be written past `size` bytes
            buf = np.full(size + 64, 0xA5, np.uint8)
            data = bytes(bad)
            rc = native.lib().comet_page_decompress(codec, data, len(data), buf.ctypes.data, size)
            assert (buf[size:] == 0xA5).all(), (name, trial, "wrote past the output")
            outcomes["ok" if rc == 0 else "error"] += 1
    assert outcomes["error"] > 0, outcomes


def test_sparse_reads_through_a_snappy_stream():
    """pq::SnappyView (comet_snappy_view_read): bytes of a snappy stream's OUTPUT read without producing it — what the scan does for the run
    headers of dictionary-encoded pages it ships compressed.  Against pyarrow's decompression: incompressible data (bit-packed indices: a few
    long literals), text with overlapping copies, runs (offset 1); too many elements → -1 (the scan inflates that page on the host)."""
    import ctypes
    import numpy as np
    import pyarrow as pa
    from datafusion_comet_amd import native
    lib = native.lib()
    lib.comet_snappy_view_read.restype = ctypes.c_int64
    rng = np.random.default_rng(3)
    raws = [rng.integers(0, 256, 300_000, dtype=np.uint8).tobytes(),
            b"".join([b"abcabcabd" * 3000, bytes(5000), rng.integers(0, 64, 20_000, dtype=np.uint8).tobytes(), b"x" * 70_000]),
            np.repeat(rng.integers(0, 50, 3000), 40).astype(np.int32).tobytes()]
    for raw in raws:
        stream = pa.compress(raw, codec="snappy", asbytes=True)
        offs = np.concatenate([np.arange(0, min(len(raw), 5000)), rng.integers(0, len(raw), 4000), [len(raw) - 1]]).astype(np.int64)
        out = np.zeros(len(offs), np.uint8)
        rc = lib.comet_snappy_view_read(stream, ctypes.c_size_t(len(stream)), 1 << 20, ctypes.c_void_p(offs.ctypes.data), len(offs), ctypes.c_void_p(out.ctypes.data))
        assert rc == len(raw)
        assert out.tobytes() == bytes(raw[int(o)] for o in offs)
        few = lib.comet_snappy_view_read(stream, ctypes.c_size_t(len(stream)), 2, ctypes.c_void_p(offs.ctypes.data), 1, ctypes.c_void_p(out.ctypes.data))
        assert few in (-1, len(raw))
    stream = pa.compress(raws[1], codec="snappy", asbytes=True)
    assert lib.comet_snappy_view_read(stream, ctypes.c_size_t(len(stream)), 8, ctypes.c_void_p(offs.ctypes.data), 1, ctypes.c_void_p(out.ctypes.data)) == -1
    assert lib.comet_snappy_view_read(stream[:-3], ctypes.c_size_t(len(stream) - 3), 1 << 20, ctypes.c_void_p(offs.ctypes.data), 1, ctypes.c_void_p(out.ctypes.data)) == -1
