"""TPC-H Q10 shape end to end (returned-item reporting): Utf8 equality filter, three hash joins, a grouped aggregate keyed by seven
columns of which five are strings up to 117 bytes long (c_name, c_phone, n_name, c_address, c_comment), a wide-decimal product, and
ORDER BY revenue DESC LIMIT 20 above the Final aggregate.  Stage A (joins + Partial aggregate) and stage B (Final aggregate below a
Sort with fetch) are native plans; both are checked against the oracle and the result against a direct Python evaluation."""
import decimal

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu

D = S.decimal(12, 2)
NATIONS = ["ALGERIA", "ARGENTINA", "BRAZIL", "CANADA", "EGYPT", "ETHIOPIA", "FRANCE", "GERMANY", "INDIA", "INDONESIA", "IRAN", "IRAQ", "JAPAN", "JORDAN", "KENYA",
           "MOROCCO", "MOZAMBIQUE", "PERU", "CHINA", "ROMANIA", "SAUDI ARABIA", "VIETNAM", "RUSSIA", "UNITED KINGDOM", "UNITED STATES"]


def _tables(nc=3000, no=30_000, seed=10):
    rng = np.random.default_rng(seed)
    words = ["furiously", "carefully", "pending", "express", "deposits", "accounts", "ironic", "packages", "über", "requests", "sleep", "quickly"]
    comment = lambda: " ".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(3, 14))))
    customer = pa.table({
        "c_custkey": pa.array(np.arange(1, nc + 1, dtype=np.int64)),
        "c_name": pa.array(["Customer#%09d" % i for i in range(1, nc + 1)]),
        "c_address": pa.array(["".join(chr(65 + int(x)) for x in rng.integers(0, 26, int(rng.integers(10, 41)))) for _ in range(nc)]),
        "c_nationkey": pa.array(rng.integers(0, 25, nc).astype(np.int32)),
        "c_phone": pa.array(["%02d-%03d-%03d-%04d" % tuple(rng.integers(10, 99, 4)) for _ in range(nc)]),
        "c_acctbal": tpch._dec128_array(rng.integers(-99999, 999999, nc), 12, 2),
        "c_comment": pa.array([comment() for _ in range(nc)]),
    })
    orders = pa.table({
        "o_orderkey": pa.array(np.arange(1, no + 1, dtype=np.int64) * 4),
        "o_custkey": pa.array(rng.integers(1, nc + 1, no)),
        "o_orderdate": pa.array(rng.integers(tpch.days(1993, 1, 1), tpch.days(1995, 1, 1), no).astype(np.int32), pa.int32()).cast(pa.date32()),
    })
    items = rng.integers(1, 8, no)
    lo = np.repeat(np.asarray(orders.column(0)), items)
    nl = len(lo)
    lineitem = pa.table({
        "l_orderkey": pa.array(lo),
        "l_extendedprice": tpch._dec128_array(rng.integers(90_000, 10_000_000, nl), 12, 2),
        "l_discount": tpch._dec128_array(rng.integers(0, 11, nl), 12, 2),
        "l_returnflag": pa.array([["R", "A", "N"][int(i)] for i in rng.integers(0, 3, nl)]),
    })
    nation = pa.table({"n_nationkey": pa.array(np.arange(25, dtype=np.int32)), "n_name": pa.array(NATIONS)})
    return customer, orders, lineitem, nation


def _plans():
    c = S.col
    I64, I32, STR, DATE = S.T_INT64, S.T_INT32, S.T_STRING, S.T_DATE
    cust_f = [I64, STR, STR, I32, STR, D, STR]
    ord_f, li_f, nat_f = [I64, I64, DATE], [I64, D, D, STR], [I32, STR]
    o = S.project(S.filter_(S.scan(ord_f), S.and_(S.gt_eq(c(2, DATE), S.lit(tpch.days(1993, 10, 1), DATE)), S.lt(c(2, DATE), S.lit(tpch.days(1994, 1, 1), DATE)))),
                  [c(0, I64), c(1, I64)])
    # customer ⋈ orders on custkey (build: the filtered orders)
    co = S.hash_join(S.scan(cust_f), o, [c(0, I64)], [c(1, I64)], S.INNER, S.BUILD_RIGHT)              # 7 customer cols ++ (o_orderkey, o_custkey)
    li = S.project(S.filter_(S.scan(li_f), S.eq(c(3, STR), S.lit("R", STR))), [c(0, I64), c(1, D), c(2, D)])
    col = S.hash_join(co, li, [c(7, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT)                          # … ++ (l_orderkey, price, discount)
    cn = S.hash_join(col, S.scan(nat_f), [c(3, I32)], [c(0, I32)], S.INNER, S.BUILD_RIGHT)              # … ++ (n_nationkey, n_name)
    one_minus = S.check_overflow(S.math("subtract", S.lit(100, D), c(11, D), S.decimal(13, 2)), S.decimal(13, 2))
    rev = S.check_overflow(S.math("multiply", c(10, D), one_minus, S.decimal(26, 4)), S.decimal(26, 4))
    p = S.project(cn, [c(0, I64), c(1, STR), c(5, D), c(4, STR), c(13, STR), c(2, STR), c(6, STR), rev])
    keys_t = [I64, STR, D, STR, STR, STR, STR]
    keys = [c(i, t) for i, t in enumerate(keys_t)]
    partial = S.hash_agg(p, keys, [S.sum_(c(7, S.decimal(26, 4)), S.decimal(36, 4))], S.PARTIAL)
    return partial, keys_t


def _direct(customer, orders, lineitem, nation):
    d0, d1 = tpch.days(1993, 10, 1), tpch.days(1994, 1, 1)
    import datetime
    epoch = datetime.date(1970, 1, 1)
    ok_orders = {k: ck for k, ck, d in zip(orders.column(0).to_pylist(), orders.column(1).to_pylist(), orders.column(2).to_pylist()) if d0 <= (d - epoch).days < d1}
    rev = {}
    for k, p, dsc, f in zip(lineitem.column(0).to_pylist(), lineitem.column(1).to_pylist(), lineitem.column(2).to_pylist(), lineitem.column(3).to_pylist()):
        if f == "R" and k in ok_orders:
            rev[ok_orders[k]] = rev.get(ok_orders[k], decimal.Decimal(0)) + p * (1 - dsc)
    cust = {r[0]: r for r in zip(*[customer.column(i).to_pylist() for i in range(7)])}
    rows = [(ck, cust[ck][1], v, cust[ck][5], NATIONS[cust[ck][3]], cust[ck][2], cust[ck][4], cust[ck][6]) for ck, v in rev.items()]
    rows.sort(key=lambda r: (-r[2], r[0]))
    return rows[:20]


def test_q10_two_stages(built):
    from oracle import oracle as O
    customer, orders, lineitem, nation = _tables()
    partial, keys_t = _plans()
    tables = [customer, orders, lineitem, nation]
    run = lambda plan, tbs, nc: pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(x) for x in tbs], nc, plan.encode(), batch_size=0))
    st = run(partial, tables, 9)
    want_st = O.run_plan_to_arrow(S, partial, tables)
    canon = lambda tb: sorted(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]), key=lambda r: r[0])
    assert canon(st) == canon(want_st)
    # stage B: Final aggregate, ORDER BY revenue DESC, custkey LIMIT 20 — the aggregate's Utf8 keys stay on the device below the Sort
    final = S.final_of(partial, st.schema)
    plan_b = S.sort(final, [(S.col(7, S.decimal(36, 4)), True), (S.col(0, S.T_INT64), False)], fetch=20)
    got = run(plan_b, [st], 8)
    want = O.run_plan_to_arrow(S, plan_b, [st])
    assert got.to_pylist() == want.to_pylist()
    direct = _direct(customer, orders, lineitem, nation)
    got_rows = list(zip(*[got.column(i).to_pylist() for i in range(8)]))
    assert [(r[0], r[1], r[7], r[2], r[4], r[5], r[3], r[6]) for r in got_rows] == direct
