"""The deprecated Parquet codec LZ4 (CompressionCodec 5; parquet-format Compression.md): Hadoop's framing — a big-endian decompressed size, then (big-endian
compressed size, raw LZ4 block) until the chunk is covered, chunks repeating — which parquet-mr writes, and the single raw block some older writers put under the
same codec number (the Parquet readers fall back to it; so does this one).  Page bodies are built here from raw blocks pyarrow compresses; a whole FILE under codec 5
is a pyarrow LZ4_RAW file whose footer says LZ4 (one byte per column chunk patched: the fallback path end to end, against pyarrow reading the same patched file)."""
import struct

import numpy as np
import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S

LZ4, LZ4_RAW = 5, 7


def _raw(b):
    return pa.compress(b, codec="lz4_raw", asbytes=True)


def _data(n, seed):
    rng = np.random.default_rng(seed)
    words = [b"lineitem ", b"MAIL", b"REG AIR", b"\x00\x01\x02", b"deliver in person "]
    return b"".join(words[int(k)] for k in rng.integers(0, len(words), n))


def test_hadoop_frames(built):
    a, b, c = _data(3000, 1), _data(10, 2), _data(70_000, 3)
    one = struct.pack(">II", len(a), len(_raw(a))) + _raw(a)
    assert native.page_decompress(LZ4, one, len(a)) == a
    # chunks repeat (what a writer with a small block size produces)
    many = b"".join(struct.pack(">II", len(x), len(_raw(x))) + _raw(x) for x in (a, b, c))
    assert native.page_decompress(LZ4, many, len(a) + len(b) + len(c)) == a + b + c
    # one chunk, several blocks (BlockCompressorStream splits a write larger than its buffer)
    halves = (c[:30_000], c[30_000:])
    split = struct.pack(">I", len(c)) + b"".join(struct.pack(">I", len(_raw(h))) + _raw(h) for h in halves)
    assert native.page_decompress(LZ4, split, len(c)) == c
    assert native.page_decompress(LZ4, b"", 0) == b""


def test_a_raw_block_under_the_old_codec_number(built):
    a = _data(5000, 4)
    assert native.page_decompress(LZ4, _raw(a), len(a)) == a
    assert native.page_decompress(LZ4_RAW, _raw(a), len(a)) == a


def test_what_is_neither_is_an_error(built):
    a = _data(3000, 5)
    framed = struct.pack(">II", len(a), len(_raw(a))) + _raw(a)
    for bad in (framed[:-7], framed + b"\x00\x00", struct.pack(">II", len(a) + 1, len(_raw(a))) + _raw(a), framed[:12] + bytes(len(framed) - 12), b"\xff" * 40):
        with pytest.raises(native.CometNativeException, match="neither Hadoop-framed nor one raw block"):
            native.page_decompress(LZ4, bad, len(a))
    with pytest.raises(native.CometNativeException, match="neither"):
        native.page_decompress(LZ4, framed, len(a) - 1)


def test_a_file_whose_footer_says_lz4(built, tmp_path):
    rng = np.random.default_rng(6)
    n = 40_000
    t = pa.table({"k": pa.array(rng.integers(0, 1000, n), pa.int64(), mask=rng.random(n) < 0.1), "v": pa.array(rng.integers(-2**40, 2**40, n), pa.int64()),
                  "s": pa.array(["name-%d" % v for v in rng.integers(0, 5000, n)])})
    path = str(tmp_path / "lz4.parquet")
    papq.write_table(t, path, compression="lz4", row_group_size=15_000, use_dictionary=["s"])
    raw = bytearray(open(path, "rb").read())
    flen = struct.unpack("<I", raw[-8:-4])[0]
    start = len(raw) - 8 - flen
    footer = bytes(raw[start:-8])
    # ColumnMetaData.codec: field 4 after the path list, an i32 — header 0x15, zigzag(7) = 0x0e → zigzag(5) = 0x0a
    nchunks = 3 * 3
    assert footer.count(b"\x15\x0e") == nchunks, footer.count(b"\x15\x0e")
    raw[start:-8] = footer.replace(b"\x15\x0e", b"\x15\x0a")
    open(path, "wb").write(raw)
    back = papq.read_table(path)          # (parquet-cpp's own fallback reads it)
    assert back.equals(t)
    plan = S.native_scan([path], t.schema.names, [S.T_INT64, S.T_INT64, S.T_STRING]).encode()
    for c, name in enumerate(("k", "v")):
        assert native.parquet_host_plain_values(plan, c) == t.column(name).combine_chunks().drop_null().to_numpy().tobytes(), name
