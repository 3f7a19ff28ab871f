"""CPU check of the hand-written Thrift/footer parser against pyarrow's metadata for files written here."""
import ctypes

import numpy as np
import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native


def _describe(path):
    lib = native.lib()
    lib.comet_parquet_describe.restype = ctypes.c_int32
    lib.comet_parquet_describe.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    buf = ctypes.create_string_buffer(1 << 16)
    rc = lib.comet_parquet_describe(str(path).encode(), buf, len(buf))
    return rc, buf.value.decode()


@pytest.mark.parametrize("compression", ["NONE", "SNAPPY", "ZSTD"])
def test_footer_matches_pyarrow_metadata(built, tmp_path, compression):
    rng = np.random.default_rng(0)
    n = 25_000
    t = pa.table({"a": pa.array(rng.integers(0, 100, n), pa.int64()), "s": pa.array([f"v{i % 37}" for i in range(n)]),
                  "d": pa.array(rng.random(n), mask=rng.random(n) < 0.1)})
    path = str(tmp_path / "f.parquet")
    papq.write_table(t, path, compression=compression, row_group_size=7_000)
    rc, text = _describe(path)
    assert rc == 0, native.lib().comet_last_error(0)
    md = papq.ParquetFile(path).metadata
    assert f"rows={md.num_rows};row_groups={md.num_row_groups};" in text
    fields = text.split(";")
    schema = fields[2][len("schema="):].strip(",").split(",")
    assert [s.split(":")[0] for s in schema] == ["a", "s", "d"]
    assert [int(s.split(":")[1]) for s in schema] == [2, 6, 5]          # INT64, BYTE_ARRAY, DOUBLE
    rgs = [f for f in fields if f.startswith("rg=")]
    codec = {"NONE": 0, "SNAPPY": 1, "ZSTD": 6}[compression]
    assert len(rgs) == md.num_row_groups
    for g, rg_text in enumerate(rgs):
        rg = md.row_group(g)
        assert rg_text.startswith(f"rg={rg.num_rows}[")
        cols = rg_text[rg_text.index("[") + 1:-1].strip(",").split(",")
        for c, ctext in enumerate(cols):
            cc = rg.column(c)
            got = [int(x) for x in ctext.split(":")]
            assert got[0] == codec and got[1] == cc.num_values and got[2] == cc.data_page_offset
            assert got[4] == cc.total_compressed_size and got[5] == cc.total_uncompressed_size


def test_not_a_parquet_file(built, tmp_path):
    p = tmp_path / "x.bin"
    p.write_bytes(b"hello world, not parquet")
    rc, _ = _describe(p)
    assert rc == -2
    assert b"PAR1" in native.lib().comet_last_error(0)
