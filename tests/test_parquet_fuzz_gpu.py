"""Randomised Parquet scans against pyarrow's reader: random column subsets and types, NULL fractions from none to almost all, dictionary on or
off, page sizes from a few hundred bytes to a megabyte, data page v1 / v2, every supported codec, page index on or off, device or host
decompression, several row groups and byte-range splits — read plainly (the run-at-a-time decode kernel, dense for columns with NULLs) and
under a pushed-down range filter on a sorted column (page-index pruning → pieces of pages → the row-at-a-time kernel), the latter compared
with the filtered table.  Seeds are fixed: a failure names its case."""
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S
from tests.test_parquet_gpu import _types

pytestmark = pytest.mark.gpu


def _column(kind, n, rng, null_frac):
    mask = (rng.random(n) < null_frac) if null_frac > 0 else None
    card = int(rng.choice([3, 200, 10**9]))
    if kind == "i32":
        return pa.array(rng.integers(-min(card, 2**31 - 1), min(card, 2**31 - 1), n).astype(np.int32), pa.int32(), mask=mask)
    if kind == "i64":
        return pa.array(rng.integers(-card, card, n), pa.int64(), mask=mask)
    if kind == "f64":
        return pa.array(rng.integers(0, card, n).astype(np.float64) / 7.0, pa.float64(), mask=mask)
    if kind == "f32":
        return pa.array((rng.integers(0, card, n) / 3.0).astype(np.float32), pa.float32(), mask=mask)
    if kind == "date":
        return pa.array(rng.integers(0, min(card, 40000), n).astype(np.int32), pa.int32(), mask=mask).cast(pa.date32())
    if kind == "dec":
        return pa.array([Decimal(int(v)).scaleb(-2) for v in rng.integers(-min(card, 10**11), min(card, 10**11), n)], pa.decimal128(12, 2), mask=mask)
    if kind == "dec38":
        return pa.array([Decimal(int(v) * 10**18).scaleb(-6) for v in rng.integers(-min(card, 10**15), min(card, 10**15), n)], pa.decimal128(38, 6), mask=mask)
    if kind == "bool":
        return pa.array(rng.random(n) < 0.5, pa.bool_(), mask=mask)
    if kind == "str":
        words = np.array(["", "a", "lineitem", "MI355X", "naïve", "x" * 33] + ["w%d" % i for i in range(min(card, 300))], dtype=object)
        return pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=mask)
    if kind == "i16":
        return pa.array(rng.integers(-min(card, 30000), min(card, 30000), n).astype(np.int16), pa.int16(), mask=mask)
    raise AssertionError(kind)


KINDS = ["i32", "i64", "f64", "f32", "date", "dec", "dec38", "bool", "str", "i16"]


@pytest.mark.parametrize("seed", range(48))
def test_random_files(built, tmp_path, seed):
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([1, 700, 20_000, 90_000]))
    kinds = list(rng.choice(KINDS, size=int(rng.integers(1, 6)), replace=False))
    null_frac = float(rng.choice([0.0, 0.0, 0.02, 0.5, 0.97]))
    cols = {"key": pa.array(np.sort(rng.integers(0, 1_000_000, n)), pa.int64())}
    for i, kd in enumerate(kinds):
        cols[f"c{i}_{kd}"] = _column(kd, n, rng, null_frac if rng.random() < 0.7 else 0.0)
    t = pa.table(cols)
    codec = str(rng.choice(["none", "snappy", "zstd", "gzip", "lz4"]))
    version = str(rng.choice(["1.0", "2.0"]))
    opts = dict(row_group_size=int(rng.choice([n + 1, max(1, n // 3), 5000])), data_page_size=int(rng.choice([300, 4 << 10, 64 << 10, 1 << 20])),
                use_dictionary=bool(rng.random() < 0.6), data_page_version=version, compression=None if codec == "none" else codec,
                write_page_index=bool(rng.random() < 0.7), store_decimal_as_integer=bool(rng.random() < 0.5))
    path = str(tmp_path / f"fuzz{seed}.parquet")
    papq.write_table(t, path, **opts)
    want = papq.read_table(path)
    cfg = S.config_map({"spark.comet.gpu.scan.deviceDecompress": str(rng.choice(["true", "false", "auto"]))})
    case = f"seed {seed}: n={n} kinds={kinds} nulls={null_frac} codec={codec} v{version} {opts}"
    types = _types(t.schema)
    # 1. the plain scan
    out = native.execute_to_table([], t.num_columns, S.native_scan([path], t.schema.names, types).encode(), batch_size=0, config=cfg)
    got = pa.Table.from_batches(out) if out else want.slice(0, 0)
    assert got.num_rows == want.num_rows, case
    for i, name in enumerate(t.schema.names):
        assert got.column(i).combine_chunks().equals(want.column(name).combine_chunks()), f"{case}: column {name}"
    # 2. a pushed-down range on the sorted key, the Filter above it
    lo, hi = sorted(int(x) for x in rng.integers(0, 1_000_000, 2))
    k = S.col(0, S.T_INT64)
    pred = S.and_(S.gt_eq(k, S.lit(lo, S.T_INT64)), S.lt_eq(k, S.lit(hi, S.T_INT64)))
    plan = S.filter_(S.native_scan([path], t.schema.names, types, data_filters=[S.gt_eq(k, S.lit(lo, S.T_INT64)), S.lt_eq(k, S.lit(hi, S.T_INT64))]), pred)
    out = native.execute_to_table([], t.num_columns, plan.encode(), batch_size=0, config=cfg)
    wantf = want.filter(pc.and_(pc.greater_equal(want.column("key"), lo), pc.less_equal(want.column("key"), hi)))
    gotf = pa.Table.from_batches(out) if out else wantf.slice(0, 0)
    assert gotf.num_rows == wantf.num_rows, f"{case} filter [{lo}, {hi}]"
    for i, name in enumerate(t.schema.names):
        assert gotf.column(i).combine_chunks().equals(wantf.column(name).combine_chunks()), f"{case} filter [{lo}, {hi}]: column {name}"
    # 3. two byte-range splits of the file read separately cover it exactly once
    size = __import__("os").path.getsize(path)
    parts = []
    for start, length in ((0, size // 2), (size // 2, size - size // 2)):
        out = native.execute_to_table([], t.num_columns, S.native_scan([(path, start, length, size)], t.schema.names, types).encode(), batch_size=0, config=cfg)
        parts += out
    assert sum(b.num_rows for b in parts) == want.num_rows, f"{case}: splits"
