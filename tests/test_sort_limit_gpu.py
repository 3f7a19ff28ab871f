"""GPU parity for Sort (with fetch = TopK, skip), Limit and operators ABOVE an aggregate (planner.rs:1436-1522):
order-preserving key bytes + LSD radix sort on the device against the oracle's comparison sort.  Ties may come out in any
order (SortExec is not stable either): tests compare the sort-key columns exactly and the payload as a multiset per key."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, parallel, serde as S, tpch

pytestmark = pytest.mark.gpu


def _run(plan, tables, ncols, **kw):
    out = native.execute_to_table([native.HostInput.from_table(t) for t in tables], ncols, plan.encode(), batch_size=0, **kw)
    return pa.Table.from_batches(out) if out else None


def _oracle(plan, tables):
    from oracle import oracle as O
    return O.run_plan_to_arrow(S, plan, tables)


def _cols(t, idx):
    return list(zip(*[t.column(i).to_pylist() for i in idx]))


def _table(n, seed):
    rng = np.random.default_rng(seed)
    f = rng.standard_normal(n)
    f[::97] = np.nan
    f[1::101] = -0.0
    return pa.table({
        "k": pa.array(rng.integers(-5, 6, n), pa.int32(), mask=rng.random(n) < 0.1),
        "d": tpch._dec128_array(rng.integers(-10**15, 10**15, n), 20, 4),
        "f": pa.array(f, mask=rng.random(n) < 0.05),
        "dt": pa.array(rng.integers(8000, 8100, n), pa.int32()).cast(pa.date32()),
        "id": pa.array(np.arange(n, dtype=np.int64)),
    })


FIELDS = [S.T_INT32, S.decimal(20, 4), S.T_DOUBLE, S.T_DATE, S.T_INT64]


@pytest.mark.parametrize("orders", [
    [(0, False, False), (1, True, True)],          # k ASC NULLS FIRST, d DESC NULLS LAST
    [(0, True, False), (2, False, True), (4, False, False)],   # k DESC NULLS FIRST, f ASC NULLS LAST (NaN, -0.0), id
    [(3, True, True), (2, True, False), (4, True, True)],
])
def test_sort_matches_oracle(built, orders):
    t = _table(50_000, 3)
    so = [(S.col(i, FIELDS[i]), desc, nl) for i, desc, nl in orders]
    plan = S.sort(S.scan(FIELDS), so)
    got, want = _run(plan, [t], 5), _oracle(plan, [t])
    keys = [i for i, _, _ in orders]
    fix = lambda rows: [tuple("nan" if isinstance(x, float) and x != x else (repr(x) if isinstance(x, float) else x) for x in r) for r in rows]
    assert fix(_cols(got, keys)) == fix(_cols(want, keys))
    assert sorted(got.column(4).to_pylist()) == list(range(t.num_rows))          # a permutation: nothing lost or duplicated
    if 4 in keys:                                                                # total order → identical tables
        assert got.column(4).to_pylist() == want.column(4).to_pylist()


def test_topk_fetch_skip_and_limit(built):
    t = _table(200_000, 4)
    so = [(S.col(1, FIELDS[1]), True, True), (S.col(4, FIELDS[4]), False, False)]
    plan = S.sort(S.filter_(S.scan(FIELDS), S.is_not_null(S.col(0, S.T_INT32))), so, fetch=100, skip=7)
    got, want = _run(plan, [t], 5), _oracle(plan, [t])
    assert got.num_rows == 93 and got.column(4).to_pylist() == want.column(4).to_pylist()
    lim = S.limit(S.filter_(S.scan(FIELDS), S.gt(S.col(0, S.T_INT32), S.lit(0, S.T_INT32))), 1000, 10)
    got, want = _run(lim, [t], 5), _oracle(lim, [t])
    assert got.num_rows == 990 and got.column(4).to_pylist() == want.column(4).to_pylist()     # FilterExec keeps the input order
    # projection on top of a limit on top of a sort
    p = S.project(S.limit(S.sort(S.scan(FIELDS), [(S.col(4, S.T_INT64), True, True)]), 5), [S.col(4, S.T_INT64), S.col(3, S.T_DATE)])
    got = _run(p, [t], 2)
    assert got.column(0).to_pylist() == [199_999, 199_998, 199_997, 199_996, 199_995]
    assert _run(S.sort(S.scan(FIELDS), so, fetch=0), [t], 5) is None and _run(S.limit(S.scan(FIELDS), 5), [t.slice(0, 0)], 5) is None


def test_tpch_q3_in_one_native_plan_with_take_ordered(built):
    """Partial aggregate → Final aggregate → Sort(revenue DESC, o_orderdate ASC) fetch 10 in ONE plan: the aggregates below the
    sort are materialised in HBM by nested execution contexts (TakeOrderedAndProject, q3.sql)."""
    customer, orders, lineitem = tpch.q3_tables(30_000, seed=9)
    partial = tpch.q3_plan()
    from oracle import oracle as O
    pstates = O.run_plan_to_arrow(S, partial, [customer, orders, lineitem])
    f = S.final_of(partial, pstates.schema)
    final_over_partial = S.hash_agg(partial, f.exprs, f.aggs, S.FINAL)
    top = S.sort(final_over_partial, [(S.col(3, S.decimal(36, 4)), True, True), (S.col(1, S.T_DATE), False, False), (S.col(0, S.T_INT64), False, False)], fetch=10)
    got = _run(top, [customer, orders, lineitem], 4)
    final = O.run_plan_to_arrow(S, f, pstates)
    want = parallel.q3_top10(final)
    assert _cols(got, [0, 1, 2, 3]) == want and got.num_rows == 10


@pytest.mark.parametrize("fetch", [None, 37])
def test_sort_by_strings_of_any_length(built, fetch):
    """ORDER BY Utf8 columns (TPC-H Q1's returnflag / linestatus, names, comments): key bytes = the value zero-padded to the column's longest
    value + its length — unsigned byte order like UTF8String.compareTo, prefixes first — sorted by the same LSD radix passes over the planes
    that vary; DESC, NULLS LAST, empty strings, embedded multi-byte characters, a computed substring key, and TopK."""
    rng = np.random.default_rng(33)
    n = 30_000
    words = ["", "a", "ab", "ab\x00", "abc", "abd", "b", "Customer#000000001", "Customer#000000002", "Customer#00000001", "über", "ue", "zebra", "日本", "日本語", "Zebra",
             "a much longer string that shares a long common prefix with others — 1", "a much longer string that shares a long common prefix with others — 2"]
    s1 = [None if rng.random() < 0.05 else words[int(i)] for i in rng.integers(0, len(words), n)]
    s2 = [None if rng.random() < 0.05 else "%s-%04d" % (words[int(i) % 7], int(j)) for i, j in zip(rng.integers(0, 50, n), rng.integers(0, 3000, n))]
    t = pa.table({"s1": pa.array(s1, pa.string()), "s2": pa.array(s2, pa.string()), "id": pa.array(np.arange(n, dtype=np.int64)), "k": pa.array(rng.integers(0, 4, n), pa.int32())})
    fields = [S.T_STRING, S.T_STRING, S.T_INT64, S.T_INT32]
    a, b, i, k = (S.col(j, ty) for j, ty in enumerate(fields))
    sub = S.scalar_func("substring", [b, S.lit(1, S.T_INT32), S.lit(3, S.T_INT32)], S.T_STRING)
    for orders in ([(a, False, False), (b, True, True), (i, False, False)], [(k, True, False), (a, True, False), (i, True, True)], [(sub, False, True), (a, False, True), (i, False, False)]):
        plan = S.sort(S.scan(fields), orders, fetch=fetch)
        got, want = _run(plan, [t], 4), _oracle(plan, [t])
        assert got.column(2).to_pylist() == want.column(2).to_pylist(), [o[1:] for o in orders]      # id makes the order total
        assert got.column(0).to_pylist() == want.column(0).to_pylist()
