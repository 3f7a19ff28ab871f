"""Device-side snappy decompression of Parquet data pages (csrc/snappy_kernels.hip): the kernel alone on raw streams — the
same hand-built and pyarrow-compressed streams the CPU suite runs through the 64-lane emulation (tests/test_snappy_emu_cpu.py) — and the
scan with PLAIN snappy pages (v1 with the levels inside the stream, v2 with the levels outside) against pyarrow's reader, with the host
decompression path as a second opinion."""
import numpy as np
import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S
from tests.test_parquet_gpu import _assert_same, _mixed_table, _types
from tests.test_snappy_emu_cpu import build

pytestmark = pytest.mark.gpu


def test_kernel_on_raw_streams(built):
    rng = np.random.default_rng(5)
    noise = lambda n: ("lit", rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    far = [noise(40_000), noise(50_000), noise(33), ("copy", (64, 90_000)), ("copy", (11, 60_000)), noise(5), ("copy", (20, 3)), ("copy", (64, 1)),
           noise(70_000), ("copy", (64, 150_000)), ("copy", (7, 70_064)), noise(1)]
    chain = [("lit", b"0123456789abcdef")]
    for k in range(300):
        chain.append(("copy", (4 + k % 8, 1 + k % 13)))
        if k % 5 == 0:
            chain.append(("lit", bytes([k & 0xFF, (k * 7) & 0xFF])))
    pages = [b"", b"a", b"hello hello hello hello hello hello", bytes(70_000), b"abcdefg" * 9000,
             rng.integers(90_000, 10_000_000, 131_072).astype(np.int64).tobytes(),          # a 1 MiB page of decimal(12,2)-as-INT64
             rng.integers(0, 50, 40_000).astype(np.int32).tobytes(), rng.standard_normal(131_072).tobytes(),
             " ".join(rng.choice(["alpha", "beta", "gamma", "lineitem", "orders", "MI355X"], 50_000)).encode()]
    streams = [pa.compress(p, codec="snappy", asbytes=True) for p in pages]
    for elems in (far, chain):
        for wide in (False, True):
            s, raw = build(elems, wide)
            streams.append(s)
            pages.append(raw)
    got, ms = native.snappy_inflate_pages(streams, [len(p) for p in pages])
    for i, (g, w) in enumerate(zip(got, pages)):
        assert g == w, f"page {i} ({len(w)} bytes) differs"
    print(f"{len(pages)} pages, {sum(map(len, pages))} bytes: {ms:.3f} ms")
    # the multi-kernel pipeline over the same streams: pyarrow's pages are fragment-shaped and decoded by it, the hand-built ones with copies
    # across 64 KiB boundaries are handed to the one-wave kernel inside the same call — every page comes back exact either way
    got2, ms2, status = native.snappy2_inflate_pages(streams, [len(p) for p in pages])
    for i, (g, w) in enumerate(zip(got2, pages)):
        assert g == w, f"pipeline: page {i} ({len(w)} bytes, status {status[i]}) differs"
    routed = lambda st, pg: 1 if (len(st) * 100 >= len(pg) * 97 and len(pg) >= 4096) else 0       # incompressible pages go straight to the one-wave kernel
    assert status[:9] == [routed(st, pg) for st, pg in zip(streams[:9], pages[:9])] and status[5] == 0 and 1 in status[9:]
    print(f"pipeline: {ms2:.3f} ms, pages to the fallback: {sum(1 for x in status if x == 1)}")


def test_pipeline_on_many_large_pages_and_the_emulation_corners(built):
    """sizes the CPU emulation cannot afford: 96 pages of 1 MiB (decimal-as-INT64, doubles, low-cardinality ints, sorted keys), plus the
    corner streams of tests/test_snappy2_emu_cpu.py on the real kernels"""
    from tests.test_snappy_emu_cpu import literal, copy, varint
    rng = np.random.default_rng(15)
    n8 = (1 << 20) // 8
    gens = [lambda: rng.integers(90_000, 10_000_000, n8).astype(np.int64).tobytes(), lambda: rng.standard_normal(n8).tobytes(),
            lambda: rng.integers(0, 50, (1 << 20) // 4).astype(np.int32).tobytes(), lambda: (np.arange(n8, dtype=np.int64) * 1000 + int(rng.integers(0, 10**9))).tobytes()]
    pages = [gens[i % 4]() for i in range(96)]
    streams = [pa.compress(p, codec="snappy", asbytes=True) for p in pages]
    noise = lambda n: ("lit", rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    deep = [("lit", b"0123456789abcdef")]
    for k in range(6000):
        deep.append(("copy", (4 + k % 8, 1 + k % 13)))
        if k % 5 == 0:
            deep.append(("lit", bytes([k & 0xFF, (k * 7) & 0xFF])))
    corners = [build(deep), build([noise(5000), ("copy", (10, 77)), noise(61), noise(3), ("copy", (64, 5000)), noise(300)] + [("copy", (5, 9)), noise(2)] * 900),
               build([noise(60), noise(61), noise(256), noise(257), noise(65_536 - 60 - 61 - 256 - 257), noise(100), ("copy", (64, 90)), ("copy", (11, 100))], wide=True),
               build([noise(65_536)]), build([noise(65_536), noise(1)]), build([noise(65_000), noise(1000), noise(10)])]
    for s_, raw in corners:
        streams.append(s_)
        pages.append(raw)
    got, ms, status = native.snappy2_inflate_pages(streams, [len(p) for p in pages])
    for i, (g, w) in enumerate(zip(got, pages)):
        assert g == w, f"page {i} ({len(w)} bytes, status {status[i]}) differs"
    assert status[:96] == [1 if i % 4 == 1 else 0 for i in range(96)]            # the pages of doubles are routed to the one-wave kernel, the rest decoded by the pipeline
    assert status[-1] == 1 and status[-6] == 0 and status[-5] == 0
    total = sum(map(len, pages))
    print(f"pipeline: {len(pages)} pages, {total} bytes in {ms:.3f} ms = {total / ms / 1e6:.1f} GB/s")
    # corrupt pages: the pipeline's checks, reported through the same error word
    raw = rng.integers(0, 1000, 50_000).astype(np.int64).tobytes()
    good = pa.compress(raw, codec="snappy", asbytes=True)
    for stream, n in [(varint(len(raw) + 1) + good[len(varint(len(raw))):], len(raw) + 1), (good[:-7], len(raw)),
                      (varint(12) + literal(b"abcdefgh") + bytes([1, 0]), 12), (varint(12) + literal(b"abcdefgh") + copy(4, 9), 12)]:
        with pytest.raises(native.CometNativeException, match="page 1: code"):
            native.snappy2_inflate_pages([good, stream], [len(raw), n])


def test_kernel_reports_corrupt_pages(built):
    raw = np.random.default_rng(7).integers(0, 1000, 5000).astype(np.int64).tobytes()
    good = pa.compress(raw, codec="snappy", asbytes=True)
    for stream, n, code in [(good, len(raw) + 1, 1), (good + b"\x00a", len(raw), 4)]:
        with pytest.raises(native.CometNativeException, match=f"code {code}"):
            native.snappy_inflate_pages([good, stream], [len(raw), n])
    broken = bytearray(build([("lit", b"abcd"), ("copy", (4, 4))])[0])
    broken[-1] = 9
    with pytest.raises(native.CometNativeException, match="page 1: code 3"):
        native.snappy_inflate_pages([good, bytes(broken)], [len(raw), 8])


def _scan_with_metrics(path, table, device):
    plan = S.native_scan([path], table.schema.names, _types(table.schema))
    it = native.CometExecIterator([], table.num_columns, plan.encode(), batch_size=0,
                                  config=S.config_map({"spark.comet.gpu.scan.deviceDecompress": "true" if device else "false"}))
    batches = []
    while True:
        b = native.Native.executePlan(it.handle, table.num_columns)
        if b is None:
            break
        batches.append(b)
    m = S.decode_metric_node(it.metrics())
    it.close()
    while m[1]:
        m = m[1][0]
    return pa.Table.from_batches(batches), m[0]


@pytest.mark.parametrize("version", ["1.0", "2.0"])
def test_scan_of_plain_snappy_pages(built, tmp_path, version):
    """PLAIN pages (dictionary off) so every fixed-width column takes the device path: NULLs put definition levels in front of the values
    inside a v1 page's stream (the host reads just that prefix), outside the stream in a v2 page"""
    t = _mixed_table(300_000, 31)
    path = str(tmp_path / f"plain_v{version[0]}.parquet")
    papq.write_table(t, path, compression="snappy", use_dictionary=False, data_page_version=version, row_group_size=120_000, data_page_size=256 << 10)
    want = papq.read_table(path)
    got, m = _scan_with_metrics(path, t, True)
    _assert_same(got, want)
    assert m["pages_decompressed_on_device"] > 20
    host, mh = _scan_with_metrics(path, t, False)
    _assert_same(host, want)
    assert mh["pages_decompressed_on_device"] == 0


def test_mixed_chunks_dictionary_then_plain(built, tmp_path):
    """pyarrow falls back from dictionary to PLAIN pages once the dictionary page is full: one column chunk then holds host-decoded
    dictionary pages followed by device-decompressed PLAIN pages — the TPC-H l_extendedprice layout"""
    rng = np.random.default_rng(32)
    n = 1_500_000
    t = pa.table({"price": pa.array(rng.integers(90_000, 10_000_000, n), pa.int64()),
                  "qty": pa.array(rng.integers(1, 51, n), pa.int64(), mask=rng.random(n) < 0.05),
                  "f": pa.array(rng.standard_normal(n))})
    path = str(tmp_path / "mixed.parquet")
    papq.write_table(t, path, compression="snappy", use_dictionary=True, row_group_size=1 << 20, data_page_size=1 << 20)
    got, m = _scan_with_metrics(path, t, True)
    _assert_same(got, papq.read_table(path))
    assert m["pages_decompressed_on_device"] >= 8


def _scan_conf(path, table, conf):
    plan = S.native_scan([path], table.schema.names, _types(table.schema))
    it = native.CometExecIterator([], table.num_columns, plan.encode(), batch_size=0, config=S.config_map(conf))
    batches = []
    while True:
        b = native.Native.executePlan(it.handle, table.num_columns)
        if b is None:
            break
        batches.append(b)
    m = S.decode_metric_node(it.metrics())
    it.close()
    while m[1]:
        m = m[1][0]
    return pa.Table.from_batches(batches), m[0]


@pytest.mark.parametrize("version", ["1.0", "2.0"])
def test_dictionary_encoded_snappy_pages_walked_on_the_device(built, tmp_path, version, monkeypatch):
    """round 5: a task with few scan threads does not look through the compressed stream for a dictionary-encoded page's run headers — the
    page is registered as pending, the device inflates it and walks the headers where they land (device/pq_runs.hpp).  Forced on and off here,
    both against pyarrow: bit widths 0 / 4 / 6 / 12, NULLs in front of the indices (v1) or outside the stream (v2), pages of NULLs only, long RLE runs"""
    from tests.test_device_zstd_gpu import _dict_table
    t = _dict_table(1_200_000, 37)
    path = str(tmp_path / f"dict_snappy_v{version[0]}.parquet")
    papq.write_table(t, path, compression="snappy", use_dictionary=True, data_page_version=version, row_group_size=500_000, data_page_size=64 << 10)
    want = papq.read_table(path)
    monkeypatch.setenv("COMET_DEVICE_RUNS_SNAPPY", "1")
    got, m = _scan_conf(path, t, {"spark.comet.gpu.scan.deviceDecompress": "true"})
    _assert_same(got, want)
    monkeypatch.setenv("COMET_DEVICE_RUNS_SNAPPY", "0")
    host, mh = _scan_conf(path, t, {"spark.comet.gpu.scan.deviceDecompress": "true"})
    _assert_same(host, want)
    # (a page that compresses into more than 2048 elements is inflated on the host when the host looks through it, on the device otherwise)
    assert m["pages_decompressed_on_device"] >= mh["pages_decompressed_on_device"] > 20


@pytest.mark.parametrize("codec", ["snappy", "zstd"])
def test_one_scan_thread_task(built, tmp_path, codec):
    """the executor's shape: spark.comet.gpu.scanThreads=1 — the device takes the dictionary-encoded pages of either codec (auto), the task's own
    thread reads pieces while it waits for its scan thread; several row groups, dictionary → PLAIN fallback inside a chunk, NULLs"""
    from tests.test_device_zstd_gpu import _dict_table
    t = _dict_table(900_000, 38)
    path = str(tmp_path / f"one_thread_{codec}.parquet")
    papq.write_table(t, path, compression=codec, use_dictionary=True, row_group_size=200_000, data_page_size=128 << 10)
    want = papq.read_table(path)
    got, m = _scan_conf(path, t, {"spark.comet.gpu.scanThreads": "1"})
    _assert_same(got, want)
    assert m["pages_decompressed_on_device"] > 20
    many, mm = _scan_conf(path, t, {"spark.comet.gpu.scanThreads": "16"})
    _assert_same(many, want)
    big = _mixed_table(400_000, 39)
    path2 = str(tmp_path / f"one_thread_plain_{codec}.parquet")
    papq.write_table(big, path2, compression=codec, use_dictionary=False, row_group_size=50_000, data_page_size=64 << 10)
    got2, _ = _scan_conf(path2, big, {"spark.comet.gpu.scanThreads": "1"})
    _assert_same(got2, papq.read_table(path2))
