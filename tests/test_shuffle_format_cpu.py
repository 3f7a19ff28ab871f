"""Shuffle block format on the host: the library's hand-written Arrow IPC writer / reader and block codecs against
pyarrow — an independent implementation of the same specifications — in both directions, through the C ABI
(comet_encode_shuffle_block / comet_decode_shuffle_block = Native.decodeShuffleBlock).  No GPU involved: these entries frame
host bytes; the ShuffleWriter operator that feeds them from the GPU is covered by tests/test_shuffle_gpu.py."""
import decimal
import struct

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native
from oracle import shuffle_oracle as SO


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()


def _batch(n, seed=5, nulls=True):
    rng = np.random.default_rng(seed)
    m = (lambda p: rng.random(n) < p) if nulls else (lambda p: None)
    return pa.record_batch({
        "i64": pa.array(rng.integers(-2**62, 2**62, n), pa.int64(), mask=m(0.1)),
        "i32": pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32), mask=m(0.5)),
        "i16": pa.array(rng.integers(-2**15, 2**15, n).astype(np.int16)),
        "i8": pa.array(rng.integers(-128, 128, n).astype(np.int8), mask=m(0.02)),
        "f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=m(0.1)),
        "f64": pa.array(rng.standard_normal(n), mask=m(0.1)),
        "bool": pa.array(rng.random(n) < 0.5, mask=m(0.3)),
        "str": pa.array([None if nulls and i % 7 == 0 else ("" if i % 5 == 0 else "v%d" % (i * 7919 % 1000)) * (i % 4) for i in range(n)], pa.string()),
        "bin": pa.array([None if nulls and i % 11 == 0 else bytes([i % 251]) * (i % 9) for i in range(n)], pa.binary()),
        "dec": pa.array([decimal.Decimal(int(x)).scaleb(-2) for x in rng.integers(-10**11, 10**11, n)], pa.decimal128(12, 2)),
        "wide": pa.array([None if nulls and i % 13 == 0 else decimal.Decimal(int(x) * 10**19 + 7).scaleb(-6) for i, x in enumerate(rng.integers(-10**17, 10**17, n))], pa.decimal128(38, 6)),
        "date": pa.array(rng.integers(0, 20000, n).astype(np.int32), pa.date32(), mask=m(0.1)),
        "ts": pa.array(rng.integers(0, 2 * 10**15, n), pa.timestamp("us", tz="UTC")),
        "ntz": pa.array(rng.integers(0, 2 * 10**15, n), pa.timestamp("us"), mask=m(0.1)),
    })


def _same(a: pa.RecordBatch, b: pa.RecordBatch):
    assert a.num_rows == b.num_rows and a.num_columns == b.num_columns
    for i in range(a.num_columns):
        assert a.column(i).type == b.column(i).type, i
        assert a.column(i).equals(b.column(i)), (i, a.schema.field(i).name if a.schema.names else i)


def test_checksum_known_answers():
    assert SO.crc32c(b"123456789") == 0xE3069283            # the CRC-32C check value (RFC 3720 appendix B.4)
    assert SO._mask(SO.crc32c(b"")) == 0xA282EAD8


@pytest.mark.parametrize("codec", [0, 1, 2, 3])
@pytest.mark.parametrize("n,first,rows", [(1, 0, 1), (3000, 0, 3000), (20000, 13, 7777), (70000, 64, 65000)])
def test_library_blocks_are_read_by_pyarrow(built, codec, n, first, rows):
    b = _batch(n).slice(first, rows)
    blk = native.encode_shuffle_block(b, codec)
    length, nfields = struct.unpack_from("<qq", blk, 0)
    assert length == len(blk) - 8 and nfields == b.num_columns and blk[16:20] == SO.TAGS[codec]
    _same(SO.decode_block(blk[16:]), b)


@pytest.mark.parametrize("codec", [0, 1, 2, 3])
@pytest.mark.parametrize("nulls", [True, False])
def test_pyarrow_blocks_are_read_by_the_library(built, codec, nulls):
    b = _batch(40000, seed=9, nulls=nulls).slice(5, 33333)
    blk = SO.encode_block(b, codec)
    got = native.decode_shuffle_block(blk[16:], b.num_columns)
    _same(got, pa.record_batch(b.columns, names=got.schema.names))


@pytest.mark.parametrize("codec", [0, 1, 2, 3])
def test_round_trip_through_the_library(built, codec):
    b = _batch(50000, seed=11)
    got = native.decode_shuffle_block(native.encode_shuffle_block(b, codec)[16:], b.num_columns)
    _same(got, pa.record_batch(b.columns, names=got.schema.names))


def test_compressible_data_shrinks(built):
    b = pa.record_batch({"k": pa.array(np.arange(200000) % 17, pa.int64()), "s": pa.array(["abcabcabc"] * 200000)})
    raw = len(native.encode_shuffle_block(b, 0))
    for codec in (1, 2, 3):
        assert len(native.encode_shuffle_block(b, codec)) < raw / 4, codec


def test_dictionary_encoded_columns_are_unpacked(built):
    """shuffle_scan.rs:175-183: native shuffle may dictionary-encode string columns; readers get plain arrays."""
    n = 5000
    s = pa.array([None if i % 9 == 0 else "k%d" % (i % 23) for i in range(n)]).dictionary_encode()
    v = pa.array(np.arange(n) % 7, pa.int64()).dictionary_encode()
    b = pa.record_batch({"s": s, "v": v, "x": pa.array(np.arange(n), pa.int32())})
    for codec in (0, 1):
        got = native.decode_shuffle_block(SO.encode_block(b, codec)[16:], 3)
        assert got.column(0).equals(s.dictionary_decode()) and got.column(1).equals(v.dictionary_decode())
        assert got.column(2).equals(b.column(2))


def test_zero_rows_write_no_block(built):
    assert native.encode_shuffle_block(_batch(10).slice(0, 0), 1) == b""     # shuffle_block_writer.rs:185-187


def test_decode_errors(built):
    with pytest.raises(native.CometNativeException, match="invalid compression codec"):
        native.decode_shuffle_block(b"GZIPxxxxxxxx", 1)
    blk = native.encode_shuffle_block(_batch(100), 0)[16:]
    with pytest.raises(native.CometNativeException):
        native.decode_shuffle_block(blk[:200], 14)
    with pytest.raises(native.CometNativeException, match="Output column count mismatch"):
        native.decode_shuffle_block(blk, 3)
    bad = bytearray(native.encode_shuffle_block(_batch(3000), 3)[16:])
    bad[len(bad) // 2] ^= 0x40
    with pytest.raises(native.CometNativeException):
        native.decode_shuffle_block(bytes(bad), 14)


@pytest.mark.parametrize("codec", [0, 1, 2, 3])
def test_corrupted_blocks_fail_cleanly(built, codec):
    """shuffle blocks come back from disk / the network: overwritten, truncated or extended bytes of the codec frame, the IPC flatbuffers or the
    buffers give an exception or a batch — never a crash or a read outside the block (offsets, lengths and vtables are all checked)"""
    import random
    b = _batch(300, seed=9)
    blk = native.encode_shuffle_block(b, codec)[16:]
    rng = random.Random(1000 + codec)
    outcomes = {"ok": 0, "error": 0}
    for trial in range(400):
        bad = bytearray(blk)
        k = trial % 4
        if k == 0:
            bad = bad[: rng.randrange(0, len(bad))]
        elif k == 1:
            bad += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
        else:
            lim = len(bad) if k == 2 else min(len(bad), 700)          # everywhere / concentrated on the frame header and the flatbuffers
            for _ in range(rng.randrange(1, 4)):
                bad[rng.randrange(0, lim)] = rng.choice([0, 0xFF, 0x80, rng.randrange(256)])
        try:
            native.decode_shuffle_block(bytes(bad), b.num_columns).validate()      # what comes back is structurally sound Arrow
            outcomes["ok"] += 1
        except native.CometNativeException:
            outcomes["error"] += 1
    assert outcomes["error"] > 0, outcomes
