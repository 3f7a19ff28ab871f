"""Row groups ruled out by the column chunks' Bloom filters (DataFusion's ParquetSource probes them for `column = literal` and `column IN (literals)` when
datafusion.execution.parquet.bloom_filter_on_read is on, the default the reference carries through: parquet_exec.rs:251-252).  Decided on the CPU
(comet_parquet_prune_report: footers and filters are read, no page is): the XXH64 the filters are keyed with against the xxhash package, the split-block test
against pyarrow's filters — a row group that HOLDS the value is never ruled out, whatever the column's type, and most row groups that do not hold it are
(the min / max statistics of these files overlap everywhere, so they rule nothing out).  Files written by pyarrow (parquet-cpp)."""
import datetime
from decimal import Decimal

import numpy as np
import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S

I32, I64, STR, DATE, TS = S.T_INT32, S.T_INT64, S.T_STRING, S.T_DATE, S.T_TIMESTAMP
D = S.decimal(12, 2)
D30 = S.decimal(30, 4)
RG, NRG = 4000, 8
NAMES = ["i", "l", "s", "d", "ts", "dec", "big"]
TYPES = [I32, I64, STR, DATE, TS, D, D30]


def _table(seed=7):
    rng = np.random.default_rng(seed)
    n = RG * NRG
    i = rng.integers(-50_000, 50_000, n).astype(np.int32)
    l = rng.integers(-2**40, 2**40, n)
    s = np.array([f"key-{v:07d}" if v % 3 else "日本-" + "x" * (v % 40) + str(v) for v in rng.integers(0, 300_000, n)], dtype=object)
    d = rng.integers(0, 20_000, n).astype(np.int32)
    ts = rng.integers(0, 1_600_000_000_000_000, n)
    dec = rng.integers(-10**9, 10**9, n)
    big = [Decimal(int(a)) * 10**12 + Decimal(int(b)) for a, b in zip(rng.integers(-10**9, 10**9, n), rng.integers(0, 10**9, n))]
    return pa.table({"i": pa.array(i, pa.int32(), mask=rng.random(n) < 0.05), "l": pa.array(l, pa.int64()), "s": pa.array(s, pa.utf8(), mask=rng.random(n) < 0.05),
                     "d": pa.array(d, pa.int32()).cast(pa.date32()), "ts": pa.array(ts, pa.int64()).cast(pa.timestamp("us", tz="UTC")),
                     "dec": pa.array([Decimal(int(v)).scaleb(-2) for v in dec], pa.decimal128(12, 2)),
                     "big": pa.array([v.scaleb(-4) for v in big], pa.decimal128(30, 4))})


def _write(tmp_path, name, t, **kw):
    path = str(tmp_path / name)
    papq.write_table(t, path, row_group_size=RG, bloom_filter_options={c: {"ndv": RG, "fpp": 0.01} for c in t.schema.names}, **kw)
    md = papq.ParquetFile(path).metadata
    assert md.num_row_groups == NRG
    return path


def _lit(col, v):
    if col == "d":
        return S.lit((v - datetime.date(1970, 1, 1)).days, DATE)
    if col == "ts":
        return S.lit(int(v.value // 1000) if hasattr(v, "value") else int((v - datetime.datetime(1970, 1, 1, tzinfo=datetime.timezone.utc)) / datetime.timedelta(microseconds=1)), TS)
    if col == "dec":
        return S.lit(int(v.scaleb(2)), D)
    if col == "big":
        return S.lit(int(v.scaleb(4)), D30)
    return S.lit(v, TYPES[NAMES.index(col)])


def _report(path, filters, **kw):
    return native.parquet_prune_report(S.native_scan([path], NAMES, TYPES, data_filters=filters).encode(), False, **kw)


_HOLD = {}


def _groups_holding(t, col, v):
    key = (id(t), col)
    if key not in _HOLD:
        m = {}
        for k, x in enumerate(t.column(col).to_pylist()):
            m.setdefault(x, set()).add(k // RG)
        _HOLD[key] = (t, m)          # (t kept alive: its id is the key)
    return _HOLD[key][1].get(v, set())


def test_xxh64_is_the_xxhash_packages(built):
    import xxhash
    rng = np.random.default_rng(1)
    for n in list(range(0, 70)) + [95, 96, 97, 127, 128, 1000, 4097]:
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for seed in (0, 42, 2**63 + 12345):
            assert native.xxh64(b, seed) == xxhash.xxh64(b, seed=seed).intdigest(), (n, seed)
    # parquet-format BloomFilter.md's reading of a value: the PLAIN encoding — four little-endian bytes of an INT32
    assert native.xxh64((1).to_bytes(4, "little")) == xxhash.xxh64((1).to_bytes(4, "little")).intdigest()


@pytest.mark.parametrize("as_integer", [False, True])
def test_a_row_group_that_holds_the_value_is_never_ruled_out(built, tmp_path, as_integer):
    """dec is written as FIXED_LEN_BYTE_ARRAY (big-endian two's complement) or, store_decimal_as_integer, as INT64 — the hash follows the file's encoding"""
    t = _table()
    path = _write(tmp_path, "bloom.parquet", t, store_decimal_as_integer=as_integer)
    rng = np.random.default_rng(5)
    pruned_when_present = 0
    for ci, col in enumerate(NAMES):
        vals = t.column(col).to_pylist()
        for k in rng.integers(0, t.num_rows, 40):
            v = vals[int(k)]
            if v is None:
                continue
            rep = _report(path, [S.eq(S.col(ci, TYPES[ci]), _lit(col, v))])
            kept = {rg["row_group"] for rg in rep["row_groups"]}
            holding = _groups_holding(t, col, v)
            assert holding <= kept, (col, v, holding, kept)
            assert rep["row_groups_pruned"] == NRG - len(kept) >= rep["row_groups_pruned_bloom_filter"]      # (now and then the min / max rule one out first)
            pruned_when_present += NRG - len(kept)
    assert pruned_when_present > 0.7 * 7 * 40 * (NRG - 1.2)      # …and most of the other row groups are (fpp 0.01)


def test_absent_values_rule_out_almost_every_row_group(built, tmp_path):
    t = _table(8)
    path = _write(tmp_path, "absent.parquet", t)
    present = {c: set(t.column(c).to_pylist()) for c in ("i", "l", "s")}
    rng = np.random.default_rng(6)
    probes = pruned = 0
    for ci, col, make in ((0, "i", lambda: int(rng.integers(-50_000, 50_000))), (1, "l", lambda: int(rng.integers(-2**40, 2**40))), (2, "s", lambda: f"key-{int(rng.integers(300_000, 900_000)):07d}")):
        for _ in range(60):
            v = make()
            if v in present[col]:
                continue
            rep = _report(path, [S.eq(S.col(ci, TYPES[ci]), S.lit(v, TYPES[ci]))])
            probes += NRG
            pruned += rep["row_groups_pruned_bloom_filter"]
            assert rep["rows"] == RG * (NRG - rep["row_groups_pruned"])
    assert probes > 1000 and pruned > 0.95 * probes
    # off: spark.comet.datafusion.execution.parquet.bloom_filter_on_read=false — nothing is ruled out (the statistics overlap)
    off = _report(path, [S.eq(S.col(0, I32), S.lit(123_456, I32))], bloom_filters=True)
    assert off["row_groups_pruned"] == NRG          # (beyond min / max: the statistics do that one)
    v = next(x for x in range(0, 50_000) if x not in present["i"])          # (in the middle of every row group's range)
    assert _report(path, [S.eq(S.col(0, I32), S.lit(v, I32))], bloom_filters=False)["row_groups_pruned"] == 0
    assert _report(path, [S.eq(S.col(0, I32), S.lit(v, I32))])["row_groups_pruned_bloom_filter"] >= NRG - 1


def test_in_lists_and_or_and_what_is_left_alone(built, tmp_path):
    t = _table(9)
    path = _write(tmp_path, "logic.parquet", t)
    iv = t.column("i").to_pylist()
    ivs = set(iv)
    absent = [x for x in range(0, 2000) if x not in ivs][:6]
    here = next(x for x in iv if x is not None)
    i, l, s = S.col(0, I32), S.col(1, I64), S.col(2, STR)
    L = lambda v: S.lit(v, I32)
    # IN: every literal must be absent; a NULL in the list matches nothing
    rep = _report(path, [S.in_(i, [L(absent[0]), L(absent[1]), S.lit(None, I32)])])
    assert rep["row_groups_pruned_bloom_filter"] >= NRG - 1
    rep = _report(path, [S.in_(i, [L(absent[0]), L(here)])])
    assert _groups_holding(t, "i", here) <= {rg["row_group"] for rg in rep["row_groups"]}
    assert _report(path, [S.in_(i, [L(absent[0])], negated=True)])["row_groups_pruned"] == 0
    # OR: both sides must be ruled out; AND: one is enough
    both = _report(path, [S.or_(S.eq(i, L(absent[2])), S.eq(s, S.lit("no such key", STR)))])
    assert both["row_groups_pruned_bloom_filter"] >= NRG - 1
    one = _report(path, [S.or_(S.eq(i, L(absent[2])), S.eq(i, L(here)))])
    assert _groups_holding(t, "i", here) <= {rg["row_group"] for rg in one["row_groups"]}
    assert _report(path, [S.and_(S.eq(i, L(here)), S.eq(s, S.lit("no such key", STR)))])["row_groups_pruned"] >= NRG - 1
    # left alone: inequalities, a literal of another type than the column's, a decimal of another scale, values before the Gregorian cutover
    assert _report(path, [S.not_(S.eq(i, L(absent[3])))])["row_groups_pruned"] == 0
    assert _report(path, [S.gt(l, S.lit(0, I64))])["row_groups_pruned"] == 0
    assert _report(path, [S.eq(S.col(5, D), S.lit(12345, S.decimal(12, 3)))])["row_groups_pruned"] == 0
    assert _report(path, [S.eq(S.col(3, DATE), S.lit(-200_000, DATE))])["row_groups_pruned_bloom_filter"] == 0


def test_files_without_filters_and_promoted_columns(built, tmp_path):
    t = _table(10)
    path = str(tmp_path / "plain.parquet")
    papq.write_table(t, path, row_group_size=RG)
    iv = set(t.column("i").to_pylist())
    v = next(x for x in range(0, 50_000) if x not in iv)
    assert _report(path, [S.eq(S.col(0, I32), S.lit(v, I32))])["row_groups_pruned"] == 0
    # an INT32 column read as bigint (type promotion): the literal is hashed as the FILE's four bytes
    bpath = _write(tmp_path, "promoted.parquet", t)
    types = [I64] + TYPES[1:]
    rep = native.parquet_prune_report(S.native_scan([bpath], NAMES, types, data_filters=[S.eq(S.col(0, I64), S.lit(v, I64))]).encode(), False)
    assert rep["row_groups_pruned_bloom_filter"] >= NRG - 1
    here = next(x for x in t.column("i").to_pylist() if x is not None)
    rep = native.parquet_prune_report(S.native_scan([bpath], NAMES, types, data_filters=[S.eq(S.col(0, I64), S.lit(here, I64))]).encode(), False)
    assert _groups_holding(t, "i", here) <= {rg["row_group"] for rg in rep["row_groups"]}
    assert native.parquet_prune_report(S.native_scan([bpath], NAMES, types, data_filters=[S.eq(S.col(0, I64), S.lit(2**40, I64))]).encode(), False)["row_groups_pruned_bloom_filter"] == 0


def test_the_split_block_test_against_a_filter_built_here(built):
    """parquet-format BloomFilter.md's insert, restated: what is inserted is found, in every block size"""
    salt = [0x47b6137b, 0x44974d91, 0x8824ad5b, 0xa2b7289d, 0x705495c7, 0x2df1424b, 0x9efc4947, 0x5c6bfb31]
    rng = np.random.default_rng(2)
    for nblocks in (1, 3, 64):
        words = np.zeros(nblocks * 8, np.uint32)
        hashes = [int(x) for x in rng.integers(0, 2**64, 200, dtype=np.uint64)]
        for h in hashes[:100]:
            b = ((h >> 32) * nblocks) >> 32
            for k in range(8):
                words[b * 8 + k] |= np.uint32(1 << (((h & 0xffffffff) * salt[k] & 0xffffffff) >> 27))
        bits = words.tobytes()
        assert all(native.sbbf_might_contain(bits, h) for h in hashes[:100])
        if nblocks == 64:
            assert sum(native.sbbf_might_contain(bits, h) for h in hashes[100:]) < 10
