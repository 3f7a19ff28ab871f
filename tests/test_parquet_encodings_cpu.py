"""Value encodings the device kernels do not read — DELTA_BINARY_PACKED, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY, BYTE_STREAM_SPLIT (what data-page-v2
writers emit) — are rewritten as PLAIN on the host before the pages are staged (parquet_meta.cpp; the reference reads them through arrow-rs,
parquet/parquet_exec.rs:60-211).  The staged bytes (comet_parquet_host_plain_values: decode_chunk_host, no device) must be the PLAIN bytes of
the column's non-NULL values as pyarrow reads them back from the same file, for both page versions, every codec, NULLs, several row groups,
wrapping deltas and the extreme values of both integer widths."""
import struct

import numpy as np
import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S

TYPES = {"i32": S.T_INT32, "i64": S.T_INT64, "d32": S.T_DATE, "ts": S.T_TIMESTAMP, "s": S.T_STRING, "p": S.T_STRING, "f32": S.T_FLOAT, "f64": S.T_DOUBLE}


def _table(n, seed):
    rng = np.random.default_rng(seed)
    i32 = rng.integers(-2**31, 2**31, n).astype(np.int32)                 # deltas wrap in 32 bits
    i32[:4] = [2**31 - 1, -2**31, 0, -1]
    i64 = np.cumsum(rng.integers(0, 1000, n)).astype(np.int64)            # small deltas: narrow miniblocks
    i64[-3:] = [2**63 - 1, -2**63, 5]                                     # … and the widest ones at the end
    words = ["", "a", "né", "delta-length-byte-array", "x" * 300]
    return pa.table({
        "i32": pa.array(i32, mask=rng.random(n) < 0.1),
        "i64": pa.array(i64),
        "d32": pa.array(rng.integers(0, 20000, n).astype(np.int32), pa.int32(), mask=rng.random(n) < 0.5).cast(pa.date32()),
        "ts": pa.array(np.sort(rng.integers(0, 2**50, n)), pa.timestamp("us", tz="UTC")),
        "s": pa.array([words[i] + str(i) for i in rng.integers(0, 5, n)], mask=rng.random(n) < 0.2),
        # sorted keys with long shared prefixes: DELTA_BYTE_ARRAY decodes to many times its page size
        "p": pa.array(sorted("Customer#%09d/segment-%s" % (int(k), "BUILDING" if k % 3 else "") for k in rng.integers(0, 10**6, n)), mask=rng.random(n) < 0.1),
        "f32": pa.array(rng.standard_normal(n).astype(np.float32), mask=rng.random(n) < 0.1),
        "f64": pa.array(rng.standard_normal(n)),
    })


def _plain_bytes(col):
    """PLAIN encoding of the non-NULL values of a pyarrow column"""
    col = col.combine_chunks().drop_null()
    t = col.type
    if pa.types.is_string(t):
        return b"".join(struct.pack("<I", len(b)) + b for b in (v.as_py().encode() for v in col))
    if pa.types.is_date32(t):
        return col.cast(pa.int32()).to_numpy().astype("<i4").tobytes()
    if pa.types.is_timestamp(t):
        return col.cast(pa.int64()).to_numpy().astype("<i8").tobytes()
    return col.to_numpy().tobytes()


ENC = {"i32": "DELTA_BINARY_PACKED", "i64": "DELTA_BINARY_PACKED", "d32": "DELTA_BINARY_PACKED", "ts": "DELTA_BINARY_PACKED", "s": "DELTA_LENGTH_BYTE_ARRAY", "p": "DELTA_BYTE_ARRAY",
       "f32": "BYTE_STREAM_SPLIT", "f64": "BYTE_STREAM_SPLIT"}


@pytest.mark.parametrize("version,codec", [("1.0", "NONE"), ("2.0", "SNAPPY"), ("2.0", "ZSTD"), ("1.0", "LZ4"), ("2.0", "GZIP")])
def test_delta_and_split_pages_stage_as_plain(built, tmp_path, version, codec):
    t = _table(30_000 if codec != "GZIP" else 5_000, 7)
    path = str(tmp_path / "enc.parquet")
    papq.write_table(t, path, use_dictionary=False, column_encoding=ENC, data_page_version=version, compression=codec, row_group_size=11_000, data_page_size=1 << 12)
    md = papq.ParquetFile(path).metadata
    for c, name in enumerate(t.schema.names):
        assert ENC[name] in md.row_group(0).column(c).encodings, (name, md.row_group(0).column(c).encodings)
    back = papq.read_table(path)
    plan = S.native_scan([path], t.schema.names, [TYPES[n] for n in t.schema.names]).encode()
    for c, name in enumerate(t.schema.names):
        got = native.parquet_host_plain_values(plan, c)
        want = _plain_bytes(back.column(name))
        assert len(got) == len(want) and got == want, name


def test_edge_shapes(built, tmp_path):
    """one value, all NULLs, exactly one block, one more than a block, constant columns (bit width 0)"""
    for k, vals in enumerate(([5], [None, None, None], list(range(128)), list(range(129)), [7] * 1000, [None] + [3] * 40 + [None], [-2**63, 2**63 - 1] * 65)):
        strs = [None if v is None else "s" * (abs(v) % 9) for v in vals]
        t = pa.table({"v": pa.array(vals, pa.int64()), "s": pa.array(strs, pa.string()), "p": pa.array(strs, pa.string())})
        path = str(tmp_path / f"edge{k}.parquet")
        papq.write_table(t, path, use_dictionary=False, column_encoding={"v": "DELTA_BINARY_PACKED", "s": "DELTA_LENGTH_BYTE_ARRAY", "p": "DELTA_BYTE_ARRAY"}, data_page_version="2.0")
        plan = S.native_scan([path], ["v", "s", "p"], [S.T_INT64, S.T_STRING, S.T_STRING]).encode()
        back = papq.read_table(path)
        for c, name in enumerate(["v", "s", "p"]):
            assert native.parquet_host_plain_values(plan, c) == _plain_bytes(back.column(name)), (k, name)


def test_prefix_compressed_strings_outgrow_their_pages(built, tmp_path):
    """DELTA_BYTE_ARRAY: 20 000 sorted 60-byte keys share almost everything with their predecessor — the pages hold a fraction of the PLAIN
    size, which the scan measures from the length blocks before it sizes the staging slot"""
    vals = ["warehouse/region-europe/country-france/city-paris/customer-%08d" % i for i in range(20_000)]
    t = pa.table({"p": pa.array(vals)})
    path = str(tmp_path / "dba.parquet")
    papq.write_table(t, path, use_dictionary=False, column_encoding={"p": "DELTA_BYTE_ARRAY"}, compression="NONE")
    md = papq.ParquetFile(path).metadata.row_group(0).column(0)
    got = native.parquet_host_plain_values(S.native_scan([path], ["p"], [S.T_STRING]).encode(), 0)
    assert got == _plain_bytes(t.column("p"))
    assert len(got) > 4 * md.total_uncompressed_size          # the footer's size says nothing about the decoded one


def test_corrupt_delta_pages_fail_cleanly(built, tmp_path):
    """bytes of the data pages overwritten at random: the decoder answers with an exception or with bytes, never with a crash or an
    out-of-bounds read (lengths, bit widths and counts are all checked against the page)"""
    rng = np.random.default_rng(5)
    t = pa.table({"v": pa.array(np.cumsum(rng.integers(-50, 5000, 4000)), pa.int64()), "s": pa.array(["w" * int(i) for i in rng.integers(0, 40, 4000)]),
                  "p": pa.array(sorted("key-%06d" % int(i) for i in rng.integers(0, 10**5, 4000)))})
    path = str(tmp_path / "ok.parquet")
    papq.write_table(t, path, use_dictionary=False, column_encoding={"v": "DELTA_BINARY_PACKED", "s": "DELTA_LENGTH_BYTE_ARRAY", "p": "DELTA_BYTE_ARRAY"}, compression="NONE",
                     data_page_size=1 << 11)
    raw = bytearray(open(path, "rb").read())
    md = papq.ParquetFile(path).metadata
    lo = min(md.row_group(0).column(c).data_page_offset for c in range(3))
    hi = max(md.row_group(0).column(c).data_page_offset + md.row_group(0).column(c).total_compressed_size for c in range(3))
    plan_of = lambda p: S.native_scan([p], ["v", "s", "p"], [S.T_INT64, S.T_STRING, S.T_STRING]).encode()
    outcomes = {"ok": 0, "error": 0}
    for trial in range(150):
        bad = bytearray(raw)
        for pos in rng.integers(lo, hi, 3):
            bad[int(pos)] = int(rng.integers(0, 256))
        q = str(tmp_path / "bad.parquet")
        open(q, "wb").write(bad)
        for c in range(3):
            try:
                native.parquet_host_plain_values(plan_of(q), c)
                outcomes["ok"] += 1
            except native.CometNativeException:
                outcomes["error"] += 1
    assert outcomes["error"] > 0 and outcomes["ok"] > 0, outcomes
