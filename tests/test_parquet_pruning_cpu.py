"""Row-group and page-index pruning decided on the CPU (comet_parquet_prune_report: footers and ColumnIndex / OffsetIndex are read, no page
is, no GPU needed): whatever the scan decides to skip must not hold a row the pushed-down filter accepts, the kept ranges are sorted and
disjoint, selective filters on a sorted column prune most of the file, AND intersects and OR unites, a nullable column's IsNotNull /
comparison pages prune by their null counts, and a file without an index prunes row groups only.  Files written by pyarrow (parquet-cpp
writes the same ColumnIndex / OffsetIndex structures parquet-mr does)."""
import numpy as np
import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import native, serde as S

I64 = S.T_INT64


def _file(tmp_path, name, index=True, n=300_000, seed=3):
    rng = np.random.default_rng(seed)
    k = np.sort(rng.integers(0, 1_000_000, n))
    blocks = np.repeat(rng.random(n // 5000 + 1) < 0.3, 5000)[:n]           # runs of 5000 NULLs: some pages of kn hold nothing but NULLs
    t = pa.table({"k": pa.array(k, pa.int64()), "kn": pa.array(k, pa.int64(), mask=blocks), "u": pa.array(rng.integers(0, 1_000_000, n), pa.int64())})
    path = str(tmp_path / name)
    # write_batch_size = 1000 with a tiny page size: a page per 1000 rows, so a run of 5000 NULLs fills whole pages
    papq.write_table(t, path, row_group_size=100_000, data_page_size=1 << 10, write_batch_size=1000, write_page_index=index)
    return path, t


def _report(path, t, filters, page_index=True):
    plan = S.native_scan([path], t.schema.names, [I64, I64, I64], data_filters=filters)
    return native.parquet_prune_report(plan.encode(), page_index)


def _kept_mask(rep, n, rg_rows=100_000):
    m = np.zeros(n, bool)
    for rg in rep["row_groups"]:
        base = rg["row_group"] * rg_rows
        last = -1
        for a, b in rg["keep"]:
            assert 0 <= a < b <= rg["num_rows"] and a >= last          # sorted, disjoint, inside the row group
            last = b
            m[base + a:base + b] = True
    return m


def test_selective_range_on_a_sorted_column(built, tmp_path):
    path, t = _file(tmp_path, "sorted.parquet")
    k = S.col(0, I64)
    rep = _report(path, t, [S.gt_eq(k, S.lit(400_000, I64)), S.lt(k, S.lit(420_000, I64))])
    kv = np.asarray(t.column("k"))
    want = (kv >= 400_000) & (kv < 420_000)
    kept = _kept_mask(rep, t.num_rows)
    assert not (want & ~kept).any()                                   # nothing the filter accepts was ruled out
    assert kept.sum() < 4 * want.sum() + 5000 and rep["rows"] == int(kept.sum())
    assert rep["row_groups_pruned"] >= 1 and rep["page_index_rows_pruned"] > 0
    off = _report(path, t, [S.gt_eq(k, S.lit(400_000, I64)), S.lt(k, S.lit(420_000, I64))], page_index=False)
    assert off["page_index_rows_pruned"] == 0 and off["rows"] % 100_000 == 0 and off["rows"] > rep["rows"]


def test_and_or_and_unprunable_leaves(built, tmp_path):
    path, t = _file(tmp_path, "logic.parquet")
    k, u = S.col(0, I64), S.col(2, I64)
    kv, uv = np.asarray(t.column("k")), np.asarray(t.column("u"))
    either = S.or_(S.lt(k, S.lit(50_000, I64)), S.gt(k, S.lit(950_000, I64)))
    kept = _kept_mask(_report(path, t, [either]), t.num_rows)
    want = (kv < 50_000) | (kv > 950_000)
    assert not (want & ~kept).any() and kept.sum() < t.num_rows // 3
    # a conjunct on an unsorted column prunes nothing by itself but must not stop the other conjunct from pruning
    both = [S.lt(k, S.lit(50_000, I64)), S.gt(u, S.lit(10, I64))]
    kept2 = _kept_mask(_report(path, t, both), t.num_rows)
    assert not (((kv < 50_000) & (uv > 10)) & ~kept2).any() and kept2.sum() < t.num_rows // 3
    # OR with a branch statistics cannot decide (not a column-vs-literal comparison): everything stays
    undecidable = S.or_(S.lt(k, S.lit(50_000, I64)), S.gt(S.math("add", k, u, I64), S.lit(0, I64)))
    assert _report(path, t, [undecidable])["rows"] == t.num_rows


def test_null_pages(built, tmp_path):
    path, t = _file(tmp_path, "nulls.parquet")
    kn = S.col(1, I64)
    valid = np.asarray(t.column("kn").is_valid())
    kept = _kept_mask(_report(path, t, [S.is_not_null(kn)]), t.num_rows)
    assert not (valid & ~kept).any() and kept.sum() < t.num_rows                     # pages holding only NULLs are ruled out
    vals = np.asarray(t.column("kn").fill_null(-1))
    kept2 = _kept_mask(_report(path, t, [S.gt(kn, S.lit(900_000, I64))]), t.num_rows)
    assert not ((valid & (vals > 900_000)) & ~kept2).any() and kept2.sum() < t.num_rows // 4


def test_file_without_an_index_prunes_row_groups_only(built, tmp_path):
    path, t = _file(tmp_path, "plain.parquet", index=False)
    k = S.col(0, I64)
    rep = _report(path, t, [S.lt(k, S.lit(50_000, I64))])
    assert rep["page_index_rows_pruned"] == 0 and rep["row_groups_pruned"] >= 1
    assert all(rg["keep"] == [[0, rg["num_rows"]]] for rg in rep["row_groups"])


def test_string_columns_prune_by_unsigned_byte_order(built, tmp_path):
    """min / max of a BYTE_ARRAY column are in unsigned bytewise order — Spark's string order: names that start with bytes ≥ 0x80 ("é…", "日本…") sort
    BEHIND the ASCII ones; a signed reading would put them first and rule out the wrong row groups.  Row groups and pages, every comparison, a shortened
    statistic (pyarrow cuts the ColumnIndex's min / max of long values) still a bound."""
    STR = S.T_STRING
    rng = np.random.default_rng(4)
    names = sorted({"name-%07d" % v for v in rng.integers(0, 2_000_000, 120_000)} | {"é-%06d" % v for v in rng.integers(0, 900_000, 40_000)} |
                   {"日本-%06d-" % v + "x" * 80 for v in rng.integers(0, 900_000, 40_000)}, key=lambda x: x.encode())
    n = len(names)
    t = pa.table({"s": pa.array(names, pa.utf8(), mask=rng.random(n) < 0.02), "u": pa.array(rng.integers(0, 100, n), pa.int64())})
    path = str(tmp_path / "strings.parquet")
    papq.write_table(t, path, row_group_size=20_000, data_page_size=1 << 12, write_batch_size=500, write_page_index=True)
    sv = np.array([x.encode() if x is not None else None for x in t.column("s").to_pylist()], dtype=object)
    valid = np.array([x is not None for x in sv])
    s = S.col(0, STR)

    def kept(filters, page_index=True):
        rep = native.parquet_prune_report(S.native_scan([path], ["s", "u"], [STR, I64], data_filters=filters).encode(), page_index)
        return _kept_mask(rep, n, rg_rows=20_000), rep

    import operator
    for lit in ("name-1000000", "é-450000", "日本-450000-" + "x" * 80, "a", "zzzz", "\U0001F600"):
        b = lit.encode()
        for make, op in ((S.eq, operator.eq), (S.lt, operator.lt), (S.lt_eq, operator.le), (S.gt, operator.gt), (S.gt_eq, operator.ge)):
            want = np.array([v is not None and op(v, b) for v in sv])
            for pi in (True, False):
                m, rep = kept([make(s, S.lit(lit, STR))], pi)
                assert not (want & ~m).any(), (lit, make.__name__, pi)
                if pi:
                    assert m.sum() <= want.sum() + 2 * 20_000 or m.sum() < 0.6 * n, (lit, make.__name__, int(m.sum()), int(want.sum()))
    # equality deep inside the non-ASCII range: one row group, a page or two
    m, rep = kept([S.eq(s, S.lit("é-450000", STR))])
    assert rep["row_groups_pruned"] >= n // 20_000 - 1 and m.sum() < 3000
    # a literal of another type decides nothing
    assert kept([S.eq(s, S.lit(5, I64))])[1]["rows"] == n


def test_in_lists_prune_by_statistics(built, tmp_path):
    """column IN (literals): a row group is ruled out when min / max rule out EVERY literal (NULLs in the list match nothing); NOT IN decides nothing"""
    path, t = _file(tmp_path, "inlist.parquet", index=False)
    k = S.col(0, I64)
    kv = np.asarray(t.column("k"))
    L = lambda v: S.lit(v, I64)
    for items in ([10, 20, 999_990], [500_000], [-5, 2_000_000], [10, None]):
        rep = _report(path, t, [S.in_(k, [L(v) for v in items])], page_index=False)
        kept = _kept_mask(rep, t.num_rows)
        want = np.isin(kv, [v for v in items if v is not None])
        assert not (want & ~kept).any(), items
        assert rep["row_groups_pruned"] >= 1 and rep["row_groups_pruned_bloom_filter"] == 0, items
    assert _report(path, t, [S.in_(k, [L(-5), L(2_000_000)])], page_index=False)["rows"] == 0
    assert _report(path, t, [S.in_(k, [L(-5)], negated=True)], page_index=False)["rows"] == t.num_rows
    # a list with something that is not a literal decides nothing
    assert _report(path, t, [S.in_(k, [L(-5), S.col(2, I64)])], page_index=False)["rows"] == t.num_rows
