"""TPC-DS Q36's shape — one plan exercising the operators around the hot path together: a hash join against a filtered dimension,
GROUP BY ROLLUP(category, class) (Expand + Partial / Final aggregates with Utf8 keys), a decimal quotient of two sums, the grouping-id
arithmetic, rank() OVER (PARTITION BY lochierarchy, CASE … ORDER BY margin) and ORDER BY … LIMIT 100.

    select sum(ss_net_profit)/sum(ss_ext_sales_price) as gross_margin, i_category, i_class,
           grouping(i_category)+grouping(i_class) as lochierarchy,
           rank() over (partition by grouping(i_category)+grouping(i_class), case when grouping(i_class) = 0 then i_category end
                        order by sum(ss_net_profit)/sum(ss_ext_sales_price) asc) as rank_within_parent
    from store_sales, item, store where … s_state in ('TN') group by rollup(i_category, i_class)
    order by lochierarchy desc, case when lochierarchy = 0 then i_category end, rank_within_parent limit 100
"""
import decimal

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu

D = S.decimal(7, 2)
I64, I32, STR = S.T_INT64, S.T_INT32, S.T_STRING
CATS = ["Books", "Children", "Electronics", "Home", "Jewelry", "Men", "Music", "Shoes", "Sports", "Women"]      # TPC-DS i_category values


def _tables(n=60_000, nitems=600, seed=36):
    rng = np.random.default_rng(seed)
    item = pa.table({"i_item_sk": pa.array(np.arange(1, nitems + 1, dtype=np.int64)),
                     "i_category": pa.array([None if rng.random() < 0.02 else CATS[int(i)] for i in rng.integers(0, len(CATS), nitems)]),
                     "i_class": pa.array(["class %02d of the catalogue" % int(i) for i in rng.integers(0, 12, nitems)])})
    store = pa.table({"s_store_sk": pa.array(np.arange(1, 21, dtype=np.int64)), "s_state": pa.array([["TN", "GA", "OH", "TN"][i % 4] for i in range(20)])})
    sales = pa.table({"ss_item_sk": pa.array(rng.integers(1, nitems + 1, n)), "ss_store_sk": pa.array(rng.integers(1, 21, n), mask=rng.random(n) < 0.02),
                      "ss_net_profit": tpch._dec128_array(rng.integers(-50_000, 90_000, n), 7, 2), "ss_ext_sales_price": tpch._dec128_array(rng.integers(100, 200_000, n), 7, 2)})
    return sales, item, store


def _stage_a():
    c = S.col
    st = S.project(S.filter_(S.scan([I64, STR]), S.eq(c(1, STR), S.lit("TN", STR))), [c(0, I64)])
    j1 = S.project(S.hash_join(S.scan([I64, I64, D, D]), st, [c(1, I64)], [c(0, I64)], S.INNER, S.BUILD_RIGHT), [c(0, I64), c(2, D), c(3, D)])
    j2 = S.project(S.hash_join(j1, S.scan([I64, STR, STR]), [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_RIGHT), [c(1, D), c(2, D), c(4, STR), c(5, STR)])   # profit, price, category, class
    NS = S.lit(None, STR)
    gid = lambda x: S.lit(x, I32)
    ex = S.expand(j2, [[c(0, D), c(1, D), c(2, STR), c(3, STR), gid(0)], [c(0, D), c(1, D), c(2, STR), NS, gid(1)], [c(0, D), c(1, D), NS, NS, gid(3)]])
    SD = S.decimal(17, 2)
    return S.hash_agg(ex, [c(2, STR), c(3, STR), c(4, I32)], [S.sum_(c(0, D), SD), S.sum_(c(1, D), SD)], S.PARTIAL), SD


def _stage_b(state_schema, partial, SD):
    c = S.col
    fin = S.final_of(partial, state_schema)                      # category, class, gid, sum(profit), sum(price)
    M = S.decimal(37, 20)
    nz = S.if_(S.eq(c(4, SD), S.lit(0, SD)), S.lit(None, SD), c(4, SD))
    margin = S.check_overflow(S.math("divide", c(3, SD), nz, M), M)
    # grouping(i_category) + grouping(i_class) from the grouping id: bit 1 = category rolled up, bit 0 = class rolled up
    loch = S.case_when([(S.eq(c(2, I32), S.lit(0, I32)), S.lit(0, I32)), (S.eq(c(2, I32), S.lit(1, I32)), S.lit(1, I32))], S.lit(2, I32))
    parent = S.case_when([(S.eq(c(2, I32), S.lit(0, I32)), c(0, STR))], S.lit(None, STR))     # case when grouping(i_class) = 0 then i_category end
    p = S.project(fin, [margin, c(0, STR), c(1, STR), loch, parent])
    pre = S.sort(p, [(c(3, I32), False, False), (c(4, STR), False, False), (c(0, M), False, False)])
    w = S.window(pre, [c(3, I32), c(4, STR)], [(c(0, M), False, False)], [("rank", [], I32)])
    out = S.project(w, [c(0, M), c(1, STR), c(2, STR), c(3, I32), c(5, I32), S.case_when([(S.eq(c(3, I32), S.lit(0, I32)), c(1, STR))], S.lit(None, STR))])
    return S.sort(out, [(c(3, I32), True, True), (c(5, STR), False, False), (c(4, I32), False, False), (c(2, STR), False, False)], fetch=100)


def test_q36_rollup_rank(built):
    from oracle import oracle as O
    sales, item, store = _tables()
    partial, SD = _stage_a()
    run = lambda plan, tbs, nc: pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(x) for x in tbs], nc, plan.encode(), batch_size=0))
    tabs = [sales, store, item]
    st, want_st = run(partial, tabs, 7), O.run_plan_to_arrow(S, partial, tabs)
    key = lambda tb: sorted(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]), key=lambda r: tuple((x is None, str(x)) for x in r[:3]))
    assert key(st) == key(want_st)
    plan_b = _stage_b(st.schema, partial, SD)
    got, want = run(plan_b, [st], 6), O.run_plan_to_arrow(S, plan_b, [st])
    assert got.schema.types == want.schema.types
    assert got.to_pylist() == want.to_pylist()
    assert got.num_rows == 100 or got.num_rows == st.num_rows
    # spot check against a direct evaluation: the grand-total row (lochierarchy 2) carries the overall margin, rank 1
    tn = {k for k, s in zip(store.column(0).to_pylist(), store.column(1).to_pylist()) if s == "TN"}
    prof = price = decimal.Decimal(0)
    for sk, p, x in zip(sales.column(1).to_pylist(), sales.column(2).to_pylist(), sales.column(3).to_pylist()):
        if sk in tn:
            prof += p
            price += x
    top = got.slice(0, 1).to_pylist()[0]
    exact = (prof / price).quantize(decimal.Decimal(1).scaleb(-20), rounding=decimal.ROUND_HALF_UP)
    assert top["col_3"] == 2 and top["col_4"] == 1 and top["col_0"] == exact
