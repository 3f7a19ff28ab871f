"""CPU restatement of Comet's native shuffle files — TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu_baseline).

Follows the reference's writer and reader, with pyarrow standing in for the arrow-ipc crate (both implement the same
Arrow IPC stream specification) and for the zstd / lz4-frame / raw-snappy codecs:

* partition ids: Spark murmur3 (seed 42) chained over the hash expressions → pmod
  (native/shuffle/src/partitioners/multi_partition.rs:296-312, comet_partitioning.rs:51-57); "round robin" = the same hash
  over the first max_hash_columns columns (multi_partition.rs:386-437); single partition = everything in partition 0.
* row order: rows of a partition keep their input order and are cut into blocks of at most batch_size rows
  (multi_partition.rs:54-103, partitioned_batch_iterator.rs:100-124).
* block: u64le length of the rest | u64le field count | 4-byte codec tag | Arrow IPC stream (schema, one record batch, EOS)
  raw / as one zstd frame / as one LZ4 frame / in Snappy framing format; zero-row batches write nothing
  (writers/shuffle_block_writer.rs:86-137,179-238).
* files: data = the partitions back to back, index = num_partitions + 1 little-endian i64 offsets
  (writers/local/local_partition_writer.rs:255-295).
* reader: the tag selects the decoder, the FIRST record batch of the stream is the block's content (native/shuffle/src/ipc.rs:23-52).

Parity pinning: the block layout constants above are the reference's own (`header_bytes`, tags, `to_le_bytes`); the IPC payload
is checked in both directions against pyarrow (tests/test_shuffle_format_cpu.py), i.e. against an independent implementation of
the same specification the reference's arrow-ipc crate implements.  The reference's Rust tests for this path
(shuffle_writer.rs:700-1430) are round trips through its own reader and carry no byte-level golden vectors.
"""
import io
import struct
from typing import List, Sequence

import numpy as np
import pyarrow as pa

from . import oracle as O

TAGS = {0: b"NONE", 1: b"ZSTD", 2: b"LZ4_", 3: b"SNAP"}
_SNAPPY_STREAM_ID = b"\xff\x06\x00\x00sNaPpY"


def _crc32c_table():
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t.append(c)
    return t


_CRC = _crc32c_table()


def crc32c(b: bytes) -> int:
    c = 0xFFFFFFFF
    for x in b:
        c = _CRC[(c ^ x) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _mask(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def snappy_frame_encode(raw: bytes) -> bytes:
    """Snappy framing format (framing_format.txt): stream identifier + chunks of ≤ 65536 uncompressed bytes."""
    out = [_SNAPPY_STREAM_ID]
    for off in range(0, len(raw), 65536):
        chunk = raw[off:off + 65536]
        comp = pa.compress(chunk, codec="snappy", asbytes=True)
        body = struct.pack("<I", _mask(crc32c(chunk))) + comp
        out.append(b"\x00" + struct.pack("<I", len(body))[:3] + body)
    return b"".join(out)


def snappy_frame_decode(buf: bytes) -> bytes:
    out, p = [], 0
    while p < len(buf):
        kind = buf[p]
        n = int.from_bytes(buf[p + 1:p + 4], "little")
        body = buf[p + 4:p + 4 + n]
        p += 4 + n
        if kind == 0xFF:
            assert body == b"sNaPpY"
        elif kind in (0, 1):
            want, = struct.unpack_from("<I", body, 0)
            if kind == 1:
                chunk = body[4:]
            else:
                # raw snappy carries its uncompressed length as a varint preamble
                ulen, shift, k = 0, 0, 4
                while True:
                    b = body[k]
                    k += 1
                    ulen |= (b & 0x7F) << shift
                    if not b & 0x80:
                        break
                    shift += 7
                chunk = pa.decompress(body[4:], decompressed_size=ulen, codec="snappy", asbytes=True)
            assert _mask(crc32c(chunk)) == want, "snappy chunk checksum"
            out.append(chunk)
        elif 2 <= kind <= 0x7F:
            raise ValueError("reserved unskippable snappy chunk")
    return b"".join(out)


def ipc_stream(batch: pa.RecordBatch) -> bytes:
    sink = io.BytesIO()
    with pa.ipc.new_stream(sink, batch.schema) as w:
        w.write_batch(batch)
    return sink.getvalue()


def encode_block(batch: pa.RecordBatch, codec: int = 0, level: int = 1) -> bytes:
    """ShuffleBlockWriter::write_batch (shuffle_block_writer.rs:179-238)."""
    if batch.num_rows == 0:
        return b""
    raw = ipc_stream(batch)
    if codec == 0:
        payload = raw
    elif codec == 1:
        sink = pa.BufferOutputStream()   # streaming encoder, like zstd::Encoder: the frame does not record its content size
        z = pa.CompressedOutputStream(sink, "zstd")
        z.write(raw)
        z.close()
        payload = sink.getvalue().to_pybytes()
    elif codec == 2:
        sink = pa.BufferOutputStream()
        z = pa.CompressedOutputStream(sink, "lz4")   # LZ4 frame format
        z.write(raw)
        z.close()
        payload = sink.getvalue().to_pybytes()
    else:
        payload = snappy_frame_encode(raw)
    rest = struct.pack("<q", batch.num_columns) + TAGS[codec] + payload
    return struct.pack("<q", len(rest)) + rest


def decode_block(block: bytes) -> pa.RecordBatch:
    """read_ipc_compressed (ipc.rs:23-52); `block` starts at the codec tag."""
    tag, payload = block[:4], block[4:]
    if tag == b"ZSTD":
        payload = pa.CompressedInputStream(pa.BufferReader(payload), "zstd").read()
    elif tag == b"LZ4_":
        payload = pa.CompressedInputStream(pa.BufferReader(payload), "lz4").read()
    elif tag == b"SNAP":
        payload = snappy_frame_decode(payload)
    elif tag != b"NONE":
        raise ValueError(f"Failed to decode batch: invalid compression codec: {tag!r}")
    return pa.ipc.open_stream(payload).read_next_batch()


def _cmp_values(a, b, descending: bool, nulls_last: bool) -> int:
    """Order of two values under one SortOrder, the way arrow's RowConverter orders them (NULL placement is independent of the
    direction; floats: −0.0 = 0.0 is not assumed — IEEE total order like the row format; strings by UTF-8 bytes)."""
    if a is None or b is None:
        if a is None and b is None:
            return 0
        first = -1 if not nulls_last else 1
        return first if a is None else -first
    if isinstance(a, str):
        a, b = a.encode(), b.encode()
    if isinstance(a, float):
        key = lambda x: (lambda bits: bits ^ 0xFFFFFFFFFFFFFFFF if bits >> 63 else bits | (1 << 63))(struct.unpack("<Q", struct.pack("<d", x))[0])
        a, b = key(a), key(b)
    c = (a > b) - (a < b)
    return -c if descending else c


def range_partition_ids(keys: Sequence[Sequence], orders: Sequence[tuple], bounds: Sequence[Sequence]) -> np.ndarray:
    """multi_partition.rs:332-366: partition id = bounds.partition_point(|bound| bound <= row).  keys: one list of values per sort
    order; orders: (descending, nulls_last) per sort order; bounds: boundary rows, ascending."""
    n = len(keys[0]) if keys else 0

    def cmp_rows(x, y):
        for (d, nl), a, b in zip(orders, x, y):
            c = _cmp_values(a, b, d, nl)
            if c:
                return c
        return 0
    out = np.zeros(n, np.int32)
    for i in range(n):
        row = [k[i] for k in keys]
        lo, hi = 0, len(bounds)
        while lo < hi:
            mid = (lo + hi) // 2
            if cmp_rows(bounds[mid], row) <= 0:
                lo = mid + 1
            else:
                hi = mid
        out[i] = lo
    return out


def partition_rows(S, table: pa.Table, partitioning: str, key_cols: Sequence[int], num_partitions: int, max_hash_columns: int = 0):
    """→ (starts[P+1], row_indices[n]) of the shuffle writer for this input."""
    n = table.num_rows
    if partitioning == "single":
        return np.array([0, n], np.int64), np.arange(n, dtype=np.uint32)
    if partitioning == "round_robin":
        k = table.num_columns if max_hash_columns <= 0 else min(max_hash_columns, table.num_columns)
        key_cols = list(range(k))
    pids = O.hash_partition_ids(S, table, key_cols, num_partitions)[:n]
    starts, idx = O.partition_starts_and_indices(pids, num_partitions)
    return starts.astype(np.int64), idx


def shuffle_write(S, table: pa.Table, partitioning: str, key_cols: Sequence[int], num_partitions: int, batch_size: int, codec: int = 0,
                  level: int = 1, max_hash_columns: int = 0):
    """→ (data file bytes, index file bytes, per-partition list of row-index arrays)."""
    P = 1 if partitioning == "single" else num_partitions
    starts, idx = partition_rows(S, table, partitioning, key_cols, P, max_hash_columns)
    parts, offsets, rows = [], [0], []
    for p in range(P):
        sel = idx[starts[p]:starts[p + 1]]
        rows.append(sel)
        chunk = b""
        for r in range(0, len(sel), batch_size):
            b = table.take(pa.array(sel[r:r + batch_size])).combine_chunks()
            chunk += encode_block(b.to_batches()[0] if b.num_rows else pa.RecordBatch.from_pylist([], b.schema), codec, level)
        parts.append(chunk)
        offsets.append(offsets[-1] + len(chunk))
    return b"".join(parts), struct.pack("<%dq" % (P + 1), *offsets), rows


def read_partition(data: bytes, index: bytes, partition: int) -> List[pa.RecordBatch]:
    offs = struct.unpack("<%dq" % (len(index) // 8), index)
    buf, p, out = data[offs[partition]:offs[partition + 1]], 0, []
    while p < len(buf):
        n, = struct.unpack_from("<q", buf, p)
        out.append(decode_block(buf[p + 16:p + 8 + n]))
        p += 8 + n
    return out


# --------------------------------------------------------------------------- columnar → UnsafeRow (columnar_to_row.rs:949-1345)


def _dec_bytes(unscaled: int) -> bytes:
    nbytes = max(1, (unscaled.bit_length() + 8) // 8) if unscaled >= 0 else max(1, ((-unscaled - 1).bit_length() + 8) // 8)
    return unscaled.to_bytes(nbytes, "big", signed=True)


def _fixed_slot(t, arr, i):
    """the 8-byte slot of a fixed-width value in a row / nested struct (get_field_value :1356-1400), or None for variable-length types"""
    v = arr[i]
    if pa.types.is_boolean(t):
        return int(v.as_py())
    if pa.types.is_integer(t):
        return v.as_py() & 0xFFFFFFFFFFFFFFFF
    if pa.types.is_date32(t) or pa.types.is_timestamp(t):
        return v.value & 0xFFFFFFFFFFFFFFFF
    if pa.types.is_float32(t):
        return struct.unpack("<I", struct.pack("<f", v.as_py()))[0]
    if pa.types.is_float64(t):
        return struct.unpack("<Q", struct.pack("<d", v.as_py()))[0]
    if pa.types.is_decimal(t) and t.precision <= 18:
        return int(v.as_py().scaleb(t.scale).to_integral_exact()) & 0xFFFFFFFFFFFFFFFF
    return None


def _nested_value(arr, i) -> bytes:
    """write_nested_variable_to_buffer (:1841-1900) for value i of `arr`: the unpadded bytes of a variable-length value"""
    t = arr.type
    v = arr[i]
    if pa.types.is_string(t):
        return v.as_py().encode()
    if pa.types.is_binary(t):
        return v.as_py()
    if pa.types.is_decimal(t):
        return _dec_bytes(int(v.as_py().scaleb(t.scale).to_integral_exact()))
    if pa.types.is_struct(t):
        # write_struct_to_buffer (:1653-1730): a nested row — null bitset | 8-byte slots | variable part, offsets relative to the struct's start
        nf = t.num_fields
        bitset = ((nf + 63) // 64) * 8
        fixed, var = bytearray(bitset + 8 * nf), bytearray()
        for f in range(nf):
            child = arr.field(f)
            if not child[i].is_valid:
                fixed[f // 64 * 8 + (f % 64) // 8] |= 1 << (f % 8)
                continue
            slot = _fixed_slot(child.type, child, i)
            if slot is None:
                data = _nested_value(child, i)
                slot = 0
                if len(data) > 0:
                    off = len(fixed) + len(var)
                    var += data + b"\0" * (-len(data) % 8)
                    slot = (off << 32) | len(data)
            fixed[bitset + 8 * f:bitset + 8 * f + 8] = struct.pack("<Q", slot)
        return bytes(fixed) + bytes(var)
    if pa.types.is_list(t) or pa.types.is_large_list(t):
        offs = arr.offsets.to_pylist()
        return _array_range(arr.values, offs[i], offs[i + 1] - offs[i])
    if pa.types.is_map(t):
        # write_map_to_buffer (:1788-1836): 8-byte size of the key array | key array | value array
        offs = arr.offsets.to_pylist()
        keys = _array_range(arr.keys, offs[i], offs[i + 1] - offs[i])
        vals = _array_range(arr.items, offs[i], offs[i + 1] - offs[i])
        return struct.pack("<q", len(keys)) + keys + vals
    raise NotImplementedError(str(t))


def _array_range(values, start, n) -> bytes:
    """UnsafeArrayData of values[start:start+n] (write_range_to_buffer :570-616): element count | null bitset | elements at their natural
    width, rounded up to 8 | variable part (offsets relative to the array's start).  Primitive elements are copied in bulk — the bytes of a
    NULL slot are whatever the Arrow buffer holds (zeros for arrays built by pyarrow) — every other type leaves a NULL slot zero."""
    t = values.type
    bitset = ((n + 63) // 64) * 8
    if pa.types.is_boolean(t) or pa.types.is_int8(t):
        esize = 1
    elif pa.types.is_int16(t):
        esize = 2
    elif pa.types.is_int32(t) or pa.types.is_float32(t) or pa.types.is_date32(t):
        esize = 4
    else:
        esize = 8
    head = bytearray(struct.pack("<q", n)) + bytearray(bitset)
    elems = bytearray(-(-n * esize // 8) * 8)
    var = bytearray()
    base = 8 + bitset + len(elems)
    for k in range(n):
        v = values[start + k]
        if not v.is_valid:
            head[8 + k // 8] |= 1 << (k % 8)
            continue
        slot = _fixed_slot(t, values, start + k)
        if slot is None:
            data = _nested_value(values, start + k)
            slot = 0
            if len(data) > 0:
                off = base + len(var)
                var += data + b"\0" * (-len(data) % 8)
                slot = (off << 32) | len(data)
            elems[k * 8:k * 8 + 8] = struct.pack("<Q", slot)
        else:
            elems[k * esize:(k + 1) * esize] = struct.pack("<Q", slot)[:esize]
    return bytes(head) + bytes(elems) + bytes(var)


def unsafe_rows(batch: pa.RecordBatch) -> List[bytes]:
    """The reference's ColumnarToRowContext::convert restated: null bitset | 8-byte slots | variable-length data padded to 8.
    Integers sign-extend into the slot, floats store their bits (f32 zero-extended), Decimal128(p ≤ 18) the unscaled long; Utf8 / Binary /
    wide decimals (minimal big-endian two's complement, i128_to_spark_decimal_bytes :1532-1558) go to the variable part with
    (offset << 32) | length in the slot; NULL and zero-length values leave the slot 0."""
    import decimal
    ncols, n = batch.num_columns, batch.num_rows
    bitset = ((ncols + 63) // 64) * 8
    cols = [batch.column(i) for i in range(ncols)]
    out = []
    for r in range(n):
        fixed = bytearray(bitset + 8 * ncols)
        var = bytearray()
        for c, arr in enumerate(cols):
            t = arr.type
            v = arr[r]
            if not v.is_valid:
                fixed[c // 64 * 8 + (c % 64) // 8] |= 1 << (c % 8)
                continue
            slot = 0
            data = None
            if pa.types.is_boolean(t):
                slot = int(v.as_py())
            elif pa.types.is_integer(t):
                slot = v.as_py() & 0xFFFFFFFFFFFFFFFF
            elif pa.types.is_date32(t) or pa.types.is_timestamp(t):
                slot = v.value & 0xFFFFFFFFFFFFFFFF
            elif pa.types.is_float32(t):
                slot = struct.unpack("<I", struct.pack("<f", v.as_py()))[0]
            elif pa.types.is_float64(t):
                slot = struct.unpack("<Q", struct.pack("<d", v.as_py()))[0]
            elif pa.types.is_decimal(t):
                unscaled = int(v.as_py().scaleb(t.scale).to_integral_exact())
                if t.precision <= 18:
                    slot = unscaled & 0xFFFFFFFFFFFFFFFF
                else:
                    nbytes = max(1, (unscaled.bit_length() + 8) // 8) if unscaled >= 0 else max(1, ((-unscaled - 1).bit_length() + 8) // 8)
                    data = unscaled.to_bytes(nbytes, "big", signed=True)
            elif pa.types.is_string(t) or pa.types.is_binary(t):
                data = v.as_py().encode() if pa.types.is_string(t) else v.as_py()
            elif pa.types.is_struct(t) or pa.types.is_list(t) or pa.types.is_large_list(t) or pa.types.is_map(t):
                data = _nested_value(arr, r)
            else:
                raise NotImplementedError(str(t))
            if data is not None:
                if len(data) > 0:
                    off = len(fixed) + len(var)
                    var += data + b"\0" * (-len(data) % 8)
                    slot = (off << 32) | len(data)
            fixed[bitset + 8 * c:bitset + 8 * c + 8] = struct.pack("<Q", slot)
        out.append(bytes(fixed) + bytes(var))
    return out
