"""Corrupted Parquet files on the CPU: whatever bytes of the footer (Thrift FileMetaData), of the page index, of the page headers or of the
pages themselves are overwritten, the host side of the scan — footer parse, row-group / page-index selection, page walk, decompression,
host-side decoding — answers with an exception or with data, never with a crash, a hang or an out-of-bounds read.  (The reference surfaces
these as CometError::Parquet → ParquetRuntimeException; here the same happens through comet_last_error.)"""
import numpy as np
import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S


def _file(tmp_path, codec):
    rng = np.random.default_rng(1)
    n = 6000
    t = pa.table({"k": pa.array(np.sort(rng.integers(0, 10**6, n)), pa.int64()), "v": pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.1),
                  "s": pa.array(["str-%d" % int(i) for i in rng.integers(0, 50, n)])})
    path = str(tmp_path / "ok.parquet")
    papq.write_table(t, path, compression=codec, row_group_size=2500, data_page_size=1 << 11, write_page_index=True, use_dictionary=["s"],
                     column_encoding={"k": "DELTA_BINARY_PACKED", "v": "PLAIN"})
    return path


def _touch_everything(path):
    filt = [S.gt(S.col(0, S.T_INT64), S.lit(500_000, S.T_INT64))]
    plan = S.native_scan([path], ["k", "v", "s"], [S.T_INT64, S.T_DOUBLE, S.T_STRING], data_filters=filt).encode()
    native.parquet_prune_report(plan, True)
    for c in range(3):
        try:
            native.parquet_host_plain_values(plan, c)
        except native.CometNativeException as e:
            # column 2 is dictionary encoded: its pages are walked (dictionary page, index runs) and the entry then refuses it by design
            if c != 2 or "dictionary-encoded" not in str(e):
                raise


@pytest.mark.parametrize("codec,region", [("NONE", "footer"), ("SNAPPY", "footer"), ("NONE", "pages"), ("SNAPPY", "pages"), ("ZSTD", "pages"), ("NONE", "anywhere")])
def test_random_corruption_never_crashes(built, tmp_path, codec, region):
    path = _file(tmp_path, codec)
    raw = bytearray(open(path, "rb").read())
    _touch_everything(path)                 # the pristine file reads
    footer_len = int.from_bytes(raw[-8:-4], "little")
    lo, hi = {"footer": (len(raw) - 8 - footer_len, len(raw) - 4), "pages": (4, len(raw) - 8 - footer_len), "anywhere": (0, len(raw))}[region]
    rng = np.random.default_rng(sum((codec + region).encode()))
    outcomes = {"ok": 0, "error": 0}
    bad_path = str(tmp_path / "bad.parquet")
    for trial in range(300):
        bad = bytearray(raw)
        for pos in rng.integers(lo, hi, int(rng.integers(1, 5))):
            bad[int(pos)] = int(rng.integers(0, 256))
        if trial % 10 == 0:                 # … and truncation
            bad = bad[: int(rng.integers(8, len(bad)))]
        open(bad_path, "wb").write(bad)
        try:
            _touch_everything(bad_path)
            outcomes["ok"] += 1
        except (native.CometNativeException, native.CometQueryExecutionException):      # (a footer that cannot be read is Spark's FAILED_READ_FILE: below)
            outcomes["error"] += 1
    assert outcomes["error"] > 0, outcomes


def test_corrupted_bloom_filters_never_crash(built, tmp_path):
    """the filters' headers and bitsets (read by the row-group selection for `column = literal` / IN), and the footer fields that say where they are"""
    rng = np.random.default_rng(2)
    n = 6000
    t = pa.table({"k": pa.array(rng.integers(0, 10**6, n), pa.int64()), "s": pa.array(["str-%d" % int(i) for i in rng.integers(0, 5000, n)])})
    path = str(tmp_path / "bloom.parquet")
    papq.write_table(t, path, row_group_size=1500, bloom_filter_options={"k": {"ndv": 1500}, "s": {"ndv": 1500}})
    raw = bytearray(open(path, "rb").read())
    k, sc = S.col(0, S.T_INT64), S.col(1, S.T_STRING)
    plan = S.native_scan([path], ["k", "s"], [S.T_INT64, S.T_STRING], data_filters=[S.or_(S.eq(k, S.lit(77, S.T_INT64)), S.in_(sc, [S.lit("str-1", S.T_STRING), S.lit("nope", S.T_STRING)]))])
    assert native.parquet_prune_report(plan.encode(), False)["row_groups_pruned_bloom_filter"] >= 1
    md = papq.ParquetFile(path).metadata
    footer_len = int.from_bytes(raw[-8:-4], "little")
    last_page_end = max(md.row_group(g).column(c).data_page_offset + md.row_group(g).column(c).total_compressed_size for g in range(md.num_row_groups) for c in range(2))
    regions = [(last_page_end, len(raw) - 8 - footer_len), (len(raw) - 8 - footer_len, len(raw) - 4)]      # the filters; the footer
    assert regions[0][1] - regions[0][0] > 8000
    bad_path = str(tmp_path / "bad.parquet")
    open(bad_path, "wb").write(raw)
    bad_plan = S.native_scan([bad_path], ["k", "s"], [S.T_INT64, S.T_STRING], data_filters=plan.data_filters).encode()
    outcomes = {"ok": 0, "error": 0}
    for trial in range(400):
        lo, hi = regions[trial % 2]
        bad = bytearray(raw)
        for pos in rng.integers(lo, hi, int(rng.integers(1, 6))):
            bad[int(pos)] = int(rng.integers(0, 256))
        if trial % 4 == 0:      # (a filter is mostly bitset: aim some hits at the first filter's header)
            bad[regions[0][0] + int(rng.integers(0, 40))] = int(rng.integers(0, 256))
        open(bad_path, "wb").write(bad)
        try:
            native.parquet_prune_report(bad_plan, False)
            outcomes["ok"] += 1
        except (native.CometNativeException, native.CometQueryExecutionException):
            outcomes["error"] += 1
    assert outcomes["ok"] > 0 and outcomes["error"] > 0, outcomes


def test_missing_and_unreadable_files_are_classified_like_the_reference(built, tmp_path):
    """jni-bridge/src/errors.rs:600-735 (try_classify_file_read_error, cannot_read_file_message): a file that is not there is FileNotFound { message }
    with object_store's "Object at location <path> not found" (ShimSparkErrorConverter cuts the path out of it for Spark's
    readCurrentFileNotFoundError); a footer that cannot be read is CannotReadFile { filePath, message } — Spark's FAILED_READ_FILE — and a
    bad magic says "is not a Parquet file" as Spark's own reader does"""
    import json
    scan = lambda p: S.native_scan([p], ["k"], [S.T_INT64]).encode()
    import os
    missing = str(tmp_path / "nope.parquet")
    open(missing, "wb").write(b"x")          # (the plan records the file's size: it is there when the plan is made and gone when the task runs)
    plan = scan(missing)
    os.unlink(missing)
    with pytest.raises(native.CometQueryExecutionException) as ei:
        native.parquet_prune_report(plan, False)
    j = json.loads(str(ei.value))
    assert j["errorType"] == "FileNotFound" and j["params"] == {"message": f"Object at location {missing} not found"}
    garbage = str(tmp_path / "garbage.parquet")
    open(garbage, "wb").write(b"this is not a parquet file at all, just text" * 3)
    with pytest.raises(native.CometQueryExecutionException) as ei:
        native.parquet_prune_report(scan(garbage), False)
    j = json.loads(str(ei.value))
    assert j["errorType"] == "CannotReadFile" and j["params"]["filePath"].endswith("garbage.parquet") and "is not a Parquet file" in j["params"]["message"]
    good = _file(tmp_path, "SNAPPY")
    raw = open(good, "rb").read()
    cut = str(tmp_path / "cut.parquet")
    open(cut, "wb").write(raw[: len(raw) - 100])          # the tail with the footer's length and magic is gone
    with pytest.raises(native.CometQueryExecutionException) as ei:
        native.parquet_prune_report(scan(cut), False)
    assert json.loads(str(ei.value))["errorType"] == "CannotReadFile"
    tiny = str(tmp_path / "tiny.parquet")
    open(tiny, "wb").write(b"PAR1")
    with pytest.raises(native.CometQueryExecutionException) as ei:
        native.parquet_prune_report(scan(tiny), False)
    assert "is not a Parquet file" in json.loads(str(ei.value))["params"]["message"]
