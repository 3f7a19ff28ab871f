"""zstd pages on the device (csrc/device/zstd2.hpp, zstd2_kernels.hip): the pipeline alone on raw frames — the same libzstd
and hand-built frames the CPU suite runs through the host emulation (tests/test_zstd2_emu_cpu.py), plus many 1 MiB pages at once — and the
scan with PLAIN zstd pages (v1 with the levels inside the frame: the host decodes just that prefix; v2 with the levels outside) against
pyarrow's reader, with the host decompression path as a second opinion."""
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S
from tests.test_parquet_gpu import _assert_same, _mixed_table, _types
from tests.test_zstd2_emu_cpu import frame, scan_pages

pytestmark = pytest.mark.gpu


def test_pipeline_on_libzstd_frames_at_every_level(built):
    pages = scan_pages()
    for level in (-5, 1, 3, 9, 19):
        codec = pa.Codec("zstd", compression_level=level)
        streams = [codec.compress(p, asbytes=True) for p in pages]
        got, ms, status = native.zstd2_inflate_pages(streams, [len(p) for p in pages])
        assert status == [0] * len(pages), (level, status)
        for i, (g, p) in enumerate(zip(got, pages)):
            assert g == p, (level, i)


def test_many_large_pages_and_streaming_frames(built):
    rng = np.random.default_rng(41)
    pages = [rng.integers(90_000, 10_000_000, 131072).astype(np.int64).tobytes() for _ in range(40)] + \
            [rng.integers(0, 50, 262144).astype(np.int32).tobytes() for _ in range(30)] + \
            [rng.standard_normal(131072).tobytes() for _ in range(10)] + [(np.arange(131072, dtype=np.int64) * 7 + k).tobytes() for k in range(16)]
    streams = [pa.compress(p, codec="zstd", asbytes=True) for p in pages]
    for k in range(0, len(pages), 5):                     # every fifth as a streaming frame (no content size)
        sink = pa.BufferOutputStream()
        out = pa.CompressedOutputStream(sink, "zstd")
        out.write(pages[k])
        out.close()
        streams[k] = sink.getvalue().to_pybytes()
    got, ms, status = native.zstd2_inflate_pages(streams, [len(p) for p in pages])
    assert status == [0] * len(pages)
    assert got == pages


def test_hand_built_frames_and_frames_kept_on_the_host(built):
    rle_lits = bytes([(27 << 3) | 1, ord("x"), 0])
    body = [("raw", b"0123456789" * 7000), ("rle", (ord("z"), 70_000)), ("comp", rle_lits), ("raw", b"tail")]
    page = b"0123456789" * 7000 + b"z" * 70_000 + b"x" * 27 + b"tail"
    good = pa.compress(page, codec="zstd", asbytes=True)
    got, ms, status = native.zstd2_inflate_pages([frame(body), frame(body, content_size=len(page)), good + good, good], [len(page), len(page), 2 * len(page), len(page)])
    assert status == [0, 0, 1, 0]
    assert got[0] == page and got[1] == page and got[3] == page


def test_corrupt_frames_are_errors_naming_the_page(built):
    rng = np.random.default_rng(42)
    raw = rng.integers(0, 50, 100_000).astype(np.int32).tobytes()
    good = pa.compress(raw, codec="zstd", asbytes=True)
    hit = 0
    for seed in range(12):
        r = np.random.default_rng(seed)
        bad = bytearray(good)
        for _ in range(6):
            bad[int(r.integers(20, len(bad)))] ^= 1 << int(r.integers(0, 8))
        try:
            got, ms, status = native.zstd2_inflate_pages([good, bytes(bad)], [len(raw), len(raw)])
            assert got[0] == raw                          # (damage the decoder cannot see: the frame carries no checksum — page 0 is still right)
        except native.CometNativeException as e:
            assert "zstd page 1" in str(e)
            hit += 1
    assert hit >= 6


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
def test_scan_of_plain_zstd_pages(built, tmp_path, version):
    """PLAIN pages (dictionary off) so every fixed-width column takes the device path: NULLs put definition levels in front of the values
    inside a v1 page's frame (the host decodes just that prefix), outside the frame in a v2 page"""
    t = _mixed_table(300_000, 33)
    path = str(tmp_path / f"plain_zstd_v{version[0]}.parquet")
    papq.write_table(t, path, compression="zstd", use_dictionary=False, data_page_version=version, row_group_size=120_000, data_page_size=256 << 10)
    want = papq.read_table(path)
    got, m = _scan_with_metrics(path, t, True)
    _assert_same(got, want)
    assert m["pages_decompressed_on_device"] > 20
    host, mh = _scan_with_metrics(path, t, False)
    _assert_same(host, want)
    assert mh["pages_decompressed_on_device"] == 0


@pytest.mark.parametrize("level", [1, 9])
def test_mixed_chunks_dictionary_then_plain_zstd(built, tmp_path, level):
    """one column chunk holds host-inflated dictionary pages followed by device-inflated PLAIN pages — the TPC-H l_extendedprice layout"""
    rng = np.random.default_rng(34)
    n = 1_500_000
    t = pa.table({"price": pa.array(rng.integers(90_000, 10_000_000, n), pa.int64()),
                  "qty": pa.array(rng.integers(1, 51, n), pa.int64(), mask=rng.random(n) < 0.05),
                  "f": pa.array(rng.standard_normal(n)),
                  "k": pa.array(np.arange(n, dtype=np.int64) * 3)})
    path = str(tmp_path / "mixed_zstd.parquet")
    papq.write_table(t, path, compression="zstd", compression_level=level, use_dictionary=True, row_group_size=1 << 20, data_page_size=1 << 20)
    got, m = _scan_with_metrics(path, t, True)
    _assert_same(got, papq.read_table(path))
    assert m["pages_decompressed_on_device"] >= 8


def _dict_table(n, seed):
    rng = np.random.default_rng(seed)
    runs = np.repeat(rng.integers(0, 7, n // 50 + 1), 50)[:n]                         # long RLE runs between bit-packed ones
    nulls_block = np.zeros(n, dtype=bool)
    nulls_block[n // 3: n // 3 + 150_000] = True                                      # whole pages of NULLs only
    return pa.table({
        "disc": pa.array(rng.integers(0, 11, n), pa.int64(), mask=rng.random(n) < 0.1),       # 4-bit indices, NULLs: levels in front of them in a v1 page
        "qty": pa.array(rng.integers(1, 51, n).astype(np.int32), pa.int32()),                 # 6-bit, required
        "date": pa.array(rng.integers(8000, 10500, n), pa.int32()).cast(pa.date32()),         # 12-bit
        "runs": pa.array(runs, pa.int64()),
        "one": pa.array(np.full(n, 42, dtype=np.int64), pa.int64(), mask=rng.random(n) < 0.5),     # one dictionary entry: bit width 0
        "holes": pa.array(rng.integers(0, 100, n), pa.int64(), mask=nulls_block),
        "f": pa.array(rng.integers(0, 300, n) * 0.25, pa.float64(), mask=rng.random(n) < 0.02),
        "dec": pa.array([Decimal(int(v)).scaleb(-2) for v in rng.integers(0, 11, n)], pa.decimal128(12, 2)),
        "k": pa.array(np.arange(n, dtype=np.int64)),                                          # falls back to PLAIN after the first pages
    })


@pytest.mark.parametrize("version", ["1.0", "2.0"])
def test_scan_of_dictionary_encoded_zstd_pages(built, tmp_path, version, monkeypatch):
    """dictionary-encoded pages under zstd: the device inflates them, their index sections come back and the host reads the run headers
    there (read_columns "deferred": COMET_DEVICE_ZSTD_DICT=1); by default host threads inflate them — both against pyarrow"""
    t = _dict_table(1_200_000, 35)
    path = str(tmp_path / f"dict_zstd_v{version[0]}.parquet")
    papq.write_table(t, path, compression="zstd", use_dictionary=True, data_page_version=version, row_group_size=500_000, data_page_size=64 << 10)
    want = papq.read_table(path)
    monkeypatch.setenv("COMET_DEVICE_ZSTD_DICT", "1")
    got, m = _scan_with_metrics(path, t, True)
    _assert_same(got, want)
    monkeypatch.setenv("COMET_DEVICE_ZSTD_DICT", "0")
    host, mh = _scan_with_metrics(path, t, True)
    _assert_same(host, want)
    assert m["pages_decompressed_on_device"] > mh["pages_decompressed_on_device"] + 20


def test_dictionary_encoded_zstd_pages_under_a_pruned_scan_stay_on_the_host(built, tmp_path, monkeypatch):
    """a scan that keeps only some row ranges of a chunk clips each page's runs on the host, so it needs them there"""
    from tests.test_parquet_page_index_gpu import _run
    monkeypatch.setenv("COMET_DEVICE_ZSTD_DICT", "1")
    t = _dict_table(600_000, 36)
    path = str(tmp_path / "dict_zstd_pruned.parquet")
    papq.write_table(t, path, compression="zstd", use_dictionary=True, row_group_size=300_000, data_page_size=32 << 10, write_page_index=True)
    k = S.col(8, S.T_INT64)
    lo, hi = 100_000, 140_000
    filters = [S.gt_eq(k, S.lit(lo, S.T_INT64)), S.lt(k, S.lit(hi, S.T_INT64))]
    got, m = _run(path, t, filters, {"spark.comet.gpu.scan.deviceDecompress": "true"}, with_filter=S.and_(*filters))
    assert m["page_index_rows_pruned"] > 0
    want = papq.read_table(path).slice(lo, hi - lo)
    _assert_same(got, want)
