"""The device's walk over RLE / bit-packed hybrid run headers (csrc/device/pq_runs.hpp; Parquet Encodings.md "Run Length Encoding / Bit-Packing
Hybrid") on the CPU: the same source the gfx950 kernels compile, against an independent Python reading of the format and against pyarrow-written
dictionary pages (their index sections located with the page header parser of the scan).  What the scan relies on: byte offsets of bit-packed runs,
value starts, counts (whole groups), RLE values of every width, empty runs skipped, max_values, truncated sections reported — never wrong runs."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def walk(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("pq_runs_emu") / "libpq_runs_emu.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "datafusion-comet_amd", "csrc"),
                    os.path.join(ROOT, "tests", "emu", "pq_runs_emu.cpp"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.pq_runs_emu.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int]

    def run(section: bytes, bw: int, max_values: int = -1, lead: int = 3, tail: bytes = b"\xff" * 16):
        buf = np.frombuffer(b"\xee" * lead + section + tail, np.uint8)      # the section sits inside a larger buffer, like a page in its column
        out = np.zeros(5 * 4096, np.int64)
        n = lib.pq_runs_emu(buf.ctypes.data, lead, lead + len(section), bw, max_values, out.ctypes.data, 4096)
        if n < 0:
            return n
        return [tuple(int(x) for x in out[5 * k: 5 * k + 5]) for k in range(n)]
    return run


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def reference(section: bytes, bw: int, max_values: int = -1, lead: int = 3):
    """the format read straight off Encodings.md; byte offsets in the coordinates of the enclosing buffer (lead bytes in front)"""
    pos, vstart, runs, vb = 0, 0, [], (bw + 7) // 8
    while pos < len(section) and (max_values < 0 or vstart < max_values):
        h, sh = 0, 0
        while True:
            if pos >= len(section):
                return -1
            b = section[pos]
            pos += 1
            h |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                break
        if h & 1:
            groups = h >> 1
            if groups > (2**31 - 1 - vstart) // 8:
                return -4                      # a count that would leave the range of a value index (round 6, advisor finding)
            if pos + groups * bw > len(section):
                # the bit-packed payload must end inside the section — or, when the page's value count is known, at least the values still wanted must
                wanted = (max_values - vstart) if max_values >= 0 else groups * 8
                if max_values < 0 or pos + (wanted * bw + 7) // 8 > len(section):
                    return -5
            if groups:
                runs.append((lead + pos, vstart, groups * 8, 0, 0))
            pos += groups * bw
            vstart += groups * 8
        else:
            if (h >> 1) > 2**31 - 1 - vstart:
                return -4
            if pos + vb > len(section):
                return -2
            v = int.from_bytes(section[pos:pos + vb], "little")
            pos += vb
            if h >> 1:
                runs.append((0, vstart, h >> 1, 1, v))
            vstart += h >> 1
    return runs


def test_hand_built_sections(walk):
    rle = lambda count, v, bw: varint(count << 1) + int(v).to_bytes((bw + 7) // 8, "little")
    packed = lambda groups, bw, fill=0x5A: varint((groups << 1) | 1) + bytes([fill]) * (groups * bw)
    cases = [
        (b"", 5), (rle(10, 3, 5), 5), (packed(1, 1), 1), (packed(63, 6), 6), (packed(63, 12) * 40, 12), (rle(1, 0, 1) * 100, 1),
        (rle(200, 255, 8) + packed(2, 8) + rle(9, 1, 8), 8), (rle(70000, 65535, 16) + packed(300, 16), 16),      # multi-byte varints
        (rle(5, 0xABCDEF, 24) + packed(1, 24) + rle(2**20, 1, 24), 24), (rle(3, 0xDEADBEEF, 32) + packed(2, 32) + rle(7, 0x80000001, 32), 32),
        (rle(0, 9, 4) + packed(0, 4) + rle(4, 9, 4), 4),                                                          # empty runs are skipped
        (packed(1, 3) + rle(8, 5, 3) + packed(2, 3) + rle(1000, 7, 3), 3),
        (rle(8, 0, 0) + varint((5 << 1) | 1), 0),                                                                  # bit width 0: no payload bytes at all
    ]
    for sec, bw in cases:
        assert walk(sec, bw) == reference(sec, bw), (sec[:16], bw)
    # a section that ends flush with its buffer's readable bytes (nothing may be read behind `end` as data) and one at offset 0
    assert walk(rle(10, 3, 5), 5, lead=0, tail=b"") == reference(rle(10, 3, 5), 5, lead=0)
    assert walk(packed(2, 7) + rle(3, 1, 7), 7, lead=0, tail=b"") == reference(packed(2, 7) + rle(3, 1, 7), 7, lead=0)
    # max_values: the walk stops behind the run that reaches it
    sec = rle(10, 1, 4) + packed(2, 4) + rle(10, 2, 4) + rle(10, 3, 4)
    for mv in (1, 10, 11, 26, 27, 36, 1000):
        assert walk(sec, 4, mv) == reference(sec, 4, mv), mv
    # truncated header (continuation bit on the last byte), truncated RLE value, bad width
    assert walk(rle(10, 1, 4) + b"\x80", 4) == -1 and reference(rle(10, 1, 4) + b"\x80", 4) == -1
    assert walk(varint(10 << 1) + b"\x01", 16) == -2 and reference(varint(10 << 1) + b"\x01", 16) == -2
    assert walk(rle(1, 1, 8), 33) == -3


def test_random_sections(walk):
    rng = np.random.default_rng(77)
    for trial in range(3000):
        bw = int(rng.integers(0, 33))
        sec = bytearray()
        for _ in range(int(rng.integers(0, 40))):
            if rng.random() < 0.5:
                groups = int(rng.integers(0, 70)) if rng.random() < 0.9 else int(rng.integers(0, 3000))
                sec += varint((groups << 1) | 1) + rng.integers(0, 256, groups * bw, dtype=np.uint8).tobytes()
            else:
                count = int(rng.integers(0, 300)) if rng.random() < 0.8 else int(rng.integers(0, 2**28))
                sec += varint(count << 1) + rng.integers(0, 256, (bw + 7) // 8, dtype=np.uint8).tobytes()
        sec = bytes(sec)
        mv = -1 if rng.random() < 0.7 else int(rng.integers(0, 5000))
        lead = int(rng.integers(0, 9))
        got, want = walk(sec, bw, mv, lead=lead, tail=b"" if rng.random() < 0.3 else b"\xff" * 9), reference(sec, bw, mv, lead=lead)
        assert got == want, (trial, bw, mv)
        # … and cut anywhere: the same runs as the reference up to the cut, or the same error
        if sec and rng.random() < 0.5:
            cut = sec[: int(rng.integers(0, len(sec)))]
            assert walk(cut, bw, mv, lead=lead) == reference(cut, bw, mv, lead=lead), (trial, "cut")


def test_pyarrow_dictionary_pages(walk, tmp_path):
    """index sections of real dictionary-encoded pages (pyarrow: parquet-cpp's RleEncoder): located through the scan's own footer / page-header
    parser on uncompressed files, walked, and the indices they describe decoded with numpy against the column's values"""
    import pyarrow as pa
    import pyarrow.parquet as pq
    rng = np.random.default_rng(5)
    n = 50_000
    cols = {"few": rng.integers(0, 11, n), "runs": np.repeat(rng.integers(0, 50, n // 100), 100), "many": rng.integers(0, 2500, n)}
    for name, vals in cols.items():
        path = str(tmp_path / f"{name}.parquet")
        pq.write_table(pa.table({name: pa.array(vals.astype(np.int32))}), path, compression="none", use_dictionary=True, data_page_size=64 << 10, write_statistics=False)
        raw = open(path, "rb").read()
        md = pq.ParquetFile(path).metadata
        col = md.row_group(0).column(0)
        dict_off, data_off, end = col.dictionary_page_offset, col.data_page_offset, col.dictionary_page_offset + col.total_compressed_size
        # walk the thrift page headers with pyarrow's own low-level reader is not exposed: parse the few fields needed (compact protocol)
        pos, pages = dict_off, []
        while pos < end:
            hdr, body_at = _page_header(raw, pos)
            pages.append((hdr, body_at))
            pos = body_at + hdr["compressed"]
        dictionary = np.frombuffer(raw, np.int32, pages[0][0]["num_values"], pages[0][1])
        decoded = []
        for hdr, at in pages[1:]:
            body = raw[at: at + hdr["compressed"]]
            dl = int.from_bytes(body[:4], "little")              # v1 page of an optional column: definition levels first
            sec_at = 4 + dl
            bw = body[sec_at]
            sec = body[sec_at + 1:]
            runs = walk(sec, bw, -1, lead=0)
            assert runs == reference(sec, bw, -1, lead=0) and isinstance(runs, list)
            idx = np.zeros(0, np.int64)
            for byte_off, vstart, count, is_rle, v in runs:
                assert vstart == len(idx)
                if is_rle:
                    idx = np.concatenate([idx, np.full(count, v)])
                else:
                    bits = np.unpackbits(np.frombuffer(sec, np.uint8, count * bw // 8, byte_off), bitorder="little")
                    idx = np.concatenate([idx, (bits.reshape(count, bw).astype(np.int64) << np.arange(bw)).sum(axis=1)])
            decoded.append(dictionary[idx[: hdr["num_values"]]])
        assert np.array_equal(np.concatenate(decoded), vals.astype(np.int32)), name


def _page_header(raw, pos):
    """the few PageHeader fields this test needs, thrift compact protocol: type (1), uncompressed (2), compressed (3), num_values of the
    data_page_header (5) / dictionary_page_header (7)"""
    def uvar(p):
        v, sh = 0, 0
        while True:
            b = raw[p]
            p += 1
            v |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                return v, p

    def zz(p):
        v, p = uvar(p)
        return (v >> 1) ^ -(v & 1), p

    def skip(p, t):
        if t in (1, 2):
            return p
        if t in (3,):
            return p + 1
        if t in (4, 5, 6):
            return uvar(p)[1]
        if t == 7:
            return p + 8
        if t == 8:
            ln, p = uvar(p)
            return p + ln
        if t == 12:
            return struct(p, None)[1]
        if t in (9, 10):
            h = raw[p]
            p += 1
            cnt = h >> 4
            if cnt == 15:
                cnt, p = uvar(p)
            for _ in range(cnt):
                p = skip(p, h & 15)
            return p
        raise AssertionError(f"thrift type {t}")

    def struct(p, want):
        out, fid = {}, 0
        while True:
            b = raw[p]
            p += 1
            if b == 0:
                return out, p
            d, t = b >> 4, b & 15
            if d == 0:
                fid, p = zz(p)
            else:
                fid += d
            if want is not None and fid in want and t in (4, 5, 6):
                out[fid], p = zz(p)
            elif want is not None and fid in want and t == 12:
                out[fid], p = struct(p, {1})
            else:
                p = skip(p, t)

    f, p = struct(pos, {1, 2, 3, 5, 7})
    sub = f.get(5) or f.get(7) or {}
    return {"type": f[1], "uncompressed": f[2], "compressed": f[3], "num_values": sub.get(1, 0)}, p
