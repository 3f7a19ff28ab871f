"""More TPC-H query shapes end to end on the GPU, each against the oracle and a direct Python evaluation:
Q12 (IN on strings, column-to-column date comparisons, conditional sums, ORDER BY a string key), Q14 (LIKE inside CASE WHEN, a wide-decimal
product, a decimal quotient of two aggregates) and Q19 (an OR of three conjunctions mixing string IN, BETWEEN and equality)."""
import datetime
import decimal

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu

D = S.decimal(12, 2)
I64, I32, STR, DATE = S.T_INT64, S.T_INT32, S.T_STRING, S.T_DATE
MODES = ["MAIL", "SHIP", "AIR", "REG AIR", "RAIL", "TRUCK", "FOB"]
PRIOS = ["1-URGENT", "2-HIGH", "3-MEDIUM", "4-NOT SPECIFIED", "5-LOW"]
TYPES = ["PROMO BRUSHED COPPER", "STANDARD POLISHED BRASS", "PROMO PLATED STEEL", "ECONOMY ANODIZED TIN", "MEDIUM BURNISHED NICKEL", "PROMO"]
BRANDS = ["Brand#12", "Brand#23", "Brand#34", "Brand#45"]
CONTAINERS = ["SM CASE", "SM BOX", "SM PACK", "SM PKG", "MED BAG", "MED BOX", "MED PKG", "MED PACK", "LG CASE", "LG BOX", "LG PACK", "LG PKG", "JUMBO JAR"]
INSTRUCT = ["DELIVER IN PERSON", "COLLECT COD", "NONE", "TAKE BACK RETURN"]


def _tables(no=20_000, nparts=2_000, seed=12):
    rng = np.random.default_rng(seed)
    orders = pa.table({"o_orderkey": pa.array(np.arange(1, no + 1, dtype=np.int64)), "o_orderpriority": pa.array([PRIOS[int(i)] for i in rng.integers(0, 5, no)])})
    items = rng.integers(1, 8, no)
    lo = np.repeat(np.arange(1, no + 1, dtype=np.int64), items)
    n = len(lo)
    ship = rng.integers(tpch.days(1993, 6, 1), tpch.days(1995, 6, 1), n).astype(np.int32)
    commit = (ship + rng.integers(-40, 40, n)).astype(np.int32)
    receipt = (ship + rng.integers(1, 40, n)).astype(np.int32)
    d32 = lambda a: pa.array(a, pa.int32()).cast(pa.date32())
    lineitem = pa.table({
        "l_orderkey": pa.array(lo), "l_partkey": pa.array(rng.integers(1, nparts + 1, n)),
        "l_quantity": tpch._dec128_array(rng.integers(1, 51, n) * 100, 12, 2), "l_extendedprice": tpch._dec128_array(rng.integers(90_000, 10_000_000, n), 12, 2),
        "l_discount": tpch._dec128_array(rng.integers(0, 11, n), 12, 2), "l_shipdate": d32(ship), "l_commitdate": d32(commit), "l_receiptdate": d32(receipt),
        "l_shipmode": pa.array([None if rng.random() < 0.01 else MODES[int(i)] for i in rng.integers(0, len(MODES), n)]),
        "l_shipinstruct": pa.array([INSTRUCT[int(i)] for i in rng.integers(0, 4, n)]),
    })
    part = pa.table({"p_partkey": pa.array(np.arange(1, nparts + 1, dtype=np.int64)), "p_type": pa.array([TYPES[int(i)] for i in rng.integers(0, len(TYPES), nparts)]),
                     "p_brand": pa.array([BRANDS[int(i)] for i in rng.integers(0, 4, nparts)]), "p_container": pa.array([CONTAINERS[int(i)] for i in rng.integers(0, len(CONTAINERS), nparts)]),
                     "p_size": pa.array(rng.integers(1, 51, nparts).astype(np.int32))})
    return orders, lineitem, part


LI = [I64, I64, D, D, D, DATE, DATE, DATE, STR, STR]
run = lambda plan, tbs, nc: pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(x) for x in tbs], nc, plan.encode(), batch_size=0))
rows = lambda tb: list(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]))
date = lambda y, m, d: (datetime.date(y, m, d) - datetime.date(1970, 1, 1)).days
c = S.col
L = lambda s: S.lit(s, STR)


def _revenue(price, disc):
    one_minus = S.check_overflow(S.math("subtract", S.lit(100, D), disc, S.decimal(13, 2)), S.decimal(13, 2))
    return S.check_overflow(S.math("multiply", price, one_minus, S.decimal(26, 4)), S.decimal(26, 4))


def q12_partial_plan():
    """TPC-H Q12 up to the partial aggregate; inputs: orders[o_orderkey, o_orderpriority], lineitem (the LI layout)"""
    li = S.filter_(S.scan(LI), S.and_(S.and_(S.in_(c(8, STR), [L("MAIL"), L("SHIP")]), S.and_(S.lt(c(6, DATE), c(7, DATE)), S.lt(c(5, DATE), c(6, DATE)))),
                                      S.and_(S.gt_eq(c(7, DATE), S.lit(date(1994, 1, 1), DATE)), S.lt(c(7, DATE), S.lit(date(1995, 1, 1), DATE)))))
    j = S.hash_join(S.scan([I64, STR]), S.project(li, [c(0, I64), c(8, STR)]), [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_RIGHT)   # o_orderkey, prio, l_orderkey, shipmode
    urgent = S.or_(S.eq(c(1, STR), L("1-URGENT")), S.eq(c(1, STR), L("2-HIGH")))
    one, zero = S.lit(1, I32), S.lit(0, I32)
    p = S.project(j, [c(3, STR), S.case_when([(urgent, one)], zero), S.case_when([(S.and_(S.neq(c(1, STR), L("1-URGENT")), S.neq(c(1, STR), L("2-HIGH"))), one)], zero)])
    return S.hash_agg(p, [c(0, STR)], [S.sum_(c(1, I32), I64), S.sum_(c(2, I32), I64)], S.PARTIAL)


def q12_final_plan(partial, state_schema):
    return S.sort(S.final_of(partial, state_schema), [(c(0, STR), False, False)])


def test_q12_shipping_modes(built):
    from oracle import oracle as O
    orders, lineitem, _ = _tables()
    partial = q12_partial_plan()
    st = run(partial, [orders, lineitem], 3)
    final = q12_final_plan(partial, st.schema)
    got, want = run(final, [st], 3), O.run_plan_to_arrow(S, final, [O.run_plan_to_arrow(S, partial, [orders, lineitem])])
    assert rows(got) == rows(want)
    prio = dict(zip(orders.column(0).to_pylist(), orders.column(1).to_pylist()))
    ref = {}
    for k, sd, cd, rd, m in zip(*[lineitem.column(i).to_pylist() for i in (0, 5, 6, 7, 8)]):
        if m in ("MAIL", "SHIP") and cd < rd and sd < cd and datetime.date(1994, 1, 1) <= rd < datetime.date(1995, 1, 1):
            hi = prio[k] in ("1-URGENT", "2-HIGH")
            a = ref.setdefault(m, [0, 0])
            a[0 if hi else 1] += 1
    assert rows(got) == [(m, ref[m][0], ref[m][1]) for m in sorted(ref)]


def q14_partial_plan(d0, d1):
    """TPC-H Q14 up to the partial aggregate, shipdate in [d0, d1); inputs: lineitem (the LI layout), part[p_partkey, p_type, p_brand, p_container, p_size]"""
    li = S.project(S.filter_(S.scan(LI), S.and_(S.gt_eq(c(5, DATE), S.lit(d0, DATE)), S.lt(c(5, DATE), S.lit(d1, DATE)))), [c(1, I64), c(3, D), c(4, D)])
    j = S.hash_join(li, S.project(S.scan([I64, STR, STR, STR, I32]), [c(0, I64), c(1, STR)]), [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_RIGHT)   # partkey, price, disc, p_partkey, p_type
    rev = _revenue(c(1, D), c(2, D))
    R = S.decimal(26, 4)
    p = S.project(j, [S.case_when([(S.like(c(4, STR), L("PROMO%")), rev)], S.lit(decimal.Decimal("0.0000"), R)), rev])
    return S.hash_agg(p, [], [S.sum_(c(0, R), S.decimal(36, 4)), S.sum_(c(1, R), S.decimal(36, 4))], S.PARTIAL)


def q14_final_plan(partial, state_schema):
    SD = S.decimal(36, 4)
    # 100.00 * promo / total: Spark types the product decimal(38,6) and the quotient decimal(38,6)
    fin = S.final_of(partial, state_schema)
    hundred = S.lit(decimal.Decimal("100.00"), S.decimal(5, 2))
    prod = S.check_overflow(S.math("multiply", hundred, c(0, SD), S.decimal(38, 6)), S.decimal(38, 6))
    quot = S.check_overflow(S.math("divide", prod, c(1, SD), S.decimal(38, 6)), S.decimal(38, 6))
    return S.project(fin, [quot])


def test_q14_promotion_effect(built):
    from oracle import oracle as O
    _, lineitem, part = _tables()
    partial = q14_partial_plan(date(1995, 1, 1), date(1995, 2, 1))
    st = run(partial, [lineitem, part], 4)
    assert rows(st) == rows(O.run_plan_to_arrow(S, partial, [lineitem, part]))
    plan_b = q14_final_plan(partial, st.schema)
    got, want = run(plan_b, [st], 1), O.run_plan_to_arrow(S, plan_b, [st])
    assert rows(got) == rows(want)
    ptype = dict(zip(part.column(0).to_pylist(), part.column(1).to_pylist()))
    promo = total = decimal.Decimal(0)
    for pk, price, disc, sd in zip(*[lineitem.column(i).to_pylist() for i in (1, 3, 4, 5)]):
        if datetime.date(1995, 1, 1) <= sd < datetime.date(1995, 2, 1):
            r = price * (1 - disc)
            total += r
            if ptype[pk].startswith("PROMO"):
                promo += r
    exact = (decimal.Decimal(100) * promo / total).quantize(decimal.Decimal("0.000001"), rounding=decimal.ROUND_HALF_UP)
    assert got.column(0).to_pylist() == [exact]


def q4_partial_plan(d0, d1):
    """TPC-H Q4 up to the partial aggregate: orders of a quarter that have a line received after its commit date (LeftSemi), counted by
    priority; inputs: orders[o_orderkey, o_orderdate, o_orderpriority], lineitem[l_orderkey, l_commitdate, l_receiptdate]"""
    o = S.filter_(S.scan([I64, DATE, STR]), S.and_(S.gt_eq(c(1, DATE), S.lit(d0, DATE)), S.lt(c(1, DATE), S.lit(d1, DATE))))
    li = S.project(S.filter_(S.scan([I64, DATE, DATE]), S.lt(c(1, DATE), c(2, DATE))), [c(0, I64)])
    j = S.hash_join(o, li, [c(0, I64)], [c(0, I64)], S.LEFT_SEMI, S.BUILD_RIGHT)
    return S.hash_agg(S.project(j, [c(2, STR)]), [c(0, STR)], [S.count(S.lit(1, I32))], S.PARTIAL)


def q5_partial_plan(d0, d1, region_name="ASIA"):
    """TPC-H Q5 up to the partial aggregate: region ⋈ nation ⋈ customer ⋈ orders ⋈ lineitem ⋈ supplier (the last on the supplier key AND
    the customer's nation), revenue by nation.  Scan leaves in order: region[r_regionkey, r_name], nation[n_nationkey, n_name, n_regionkey],
    customer[c_custkey, c_nationkey], orders[o_orderkey, o_custkey, o_orderdate], lineitem[l_orderkey, l_suppkey, l_extendedprice, l_discount],
    supplier[s_suppkey, s_nationkey]"""
    reg = S.project(S.filter_(S.scan([I32, STR]), S.eq(c(1, STR), L(region_name))), [c(0, I32)])
    a = S.project(S.hash_join(reg, S.scan([I32, STR, I32]), [c(0, I32)], [c(2, I32)], S.INNER, S.BUILD_LEFT), [c(1, I32), c(2, STR)])                       # n_nationkey, n_name
    b = S.project(S.hash_join(a, S.scan([I64, I32]), [c(0, I32)], [c(1, I32)], S.INNER, S.BUILD_LEFT), [c(2, I64), c(3, I32), c(1, STR)])                   # c_custkey, c_nationkey, n_name
    o = S.project(S.filter_(S.scan([I64, I64, DATE]), S.and_(S.gt_eq(c(2, DATE), S.lit(d0, DATE)), S.lt(c(2, DATE), S.lit(d1, DATE)))), [c(0, I64), c(1, I64)])
    cj = S.project(S.hash_join(b, o, [c(0, I64)], [c(1, I64)], S.INNER, S.BUILD_LEFT), [c(3, I64), c(1, I32), c(2, STR)])                                   # o_orderkey, c_nationkey, n_name
    dj = S.project(S.hash_join(cj, S.scan([I64, I64, D, D]), [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT), [c(4, I64), c(1, I32), c(2, STR), c(5, D), c(6, D)])   # l_suppkey, c_nationkey, n_name, price, disc
    e = S.hash_join(dj, S.scan([I64, I32]), [c(0, I64), c(1, I32)], [c(0, I64), c(1, I32)], S.INNER, S.BUILD_RIGHT)                                         # … s_suppkey, s_nationkey
    return S.hash_agg(S.project(e, [c(2, STR), _revenue(c(3, D), c(4, D))]), [c(0, STR)], [S.sum_(c(1, S.decimal(26, 4)), S.decimal(36, 4))], S.PARTIAL)


def q5_final_plan(partial, state_schema):
    return S.sort(S.final_of(partial, state_schema), [(c(1, S.decimal(36, 4)), True, True)])


def _two_nations():
    return S.project(S.filter_(S.scan([I32, STR, I32]), S.in_(c(1, STR), [L("FRANCE"), L("GERMANY")])), [c(0, I32), c(1, STR)])       # n_nationkey, n_name


def q7_partial_plan(d0, d1):
    """TPC-H Q7 up to the partial aggregate: trade volume between two nations by year.  Scan leaves in order: nation, customer[c_custkey,
    c_nationkey], orders[o_orderkey, o_custkey], nation, supplier[s_suppkey, s_nationkey], lineitem[l_orderkey, l_suppkey, l_extendedprice,
    l_discount, l_shipdate]"""
    cus = S.project(S.hash_join(_two_nations(), S.scan([I64, I32]), [c(0, I32)], [c(1, I32)], S.INNER, S.BUILD_LEFT), [c(2, I64), c(1, STR)])                 # c_custkey, cust_nation
    ordj = S.project(S.hash_join(cus, S.scan([I64, I64]), [c(0, I64)], [c(1, I64)], S.INNER, S.BUILD_LEFT), [c(2, I64), c(1, STR)])                          # o_orderkey, cust_nation
    sup = S.project(S.hash_join(_two_nations(), S.scan([I64, I32]), [c(0, I32)], [c(1, I32)], S.INNER, S.BUILD_LEFT), [c(2, I64), c(1, STR)])                 # s_suppkey, supp_nation
    li = S.filter_(S.scan([I64, I64, D, D, DATE]), S.and_(S.gt_eq(c(4, DATE), S.lit(d0, DATE)), S.lt_eq(c(4, DATE), S.lit(d1, DATE))))
    j1 = S.project(S.hash_join(sup, li, [c(0, I64)], [c(1, I64)], S.INNER, S.BUILD_LEFT), [c(2, I64), c(1, STR), c(4, D), c(5, D), c(6, DATE)])                # l_orderkey, supp_nation, price, disc, shipdate
    pair = S.or_(S.and_(S.eq(c(3, STR), L("FRANCE")), S.eq(c(1, STR), L("GERMANY"))), S.and_(S.eq(c(3, STR), L("GERMANY")), S.eq(c(1, STR), L("FRANCE"))))
    j2 = S.hash_join(ordj, j1, [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT, condition=pair)            # o_orderkey, cust_nation, l_orderkey, supp_nation, price, disc, shipdate
    p = S.project(j2, [c(3, STR), c(1, STR), S.date_part("year", c(6, DATE)), _revenue(c(4, D), c(5, D))])
    return S.hash_agg(p, [c(0, STR), c(1, STR), c(2, I32)], [S.sum_(c(3, S.decimal(26, 4)), S.decimal(36, 4))], S.PARTIAL)


def q7_final_plan(partial, state_schema):
    return S.sort(S.final_of(partial, state_schema), [(c(0, STR), False, False), (c(1, STR), False, False), (c(2, I32), False, False)])


def q8_partial_plan(d0, d1, ptype="ECONOMY ANODIZED STEEL", region_name="AMERICA", nation_name="BRAZIL"):
    """TPC-H Q8 up to the partial aggregate: a nation's share of a region's market for one part type, by year.  Scan leaves in order: region,
    nation, customer[c_custkey, c_nationkey], orders[o_orderkey, o_custkey, o_orderdate], nation, supplier[s_suppkey, s_nationkey],
    part[p_partkey, p_type, p_brand, p_container, p_size], lineitem[l_orderkey, l_partkey, l_suppkey, l_extendedprice, l_discount]"""
    reg = S.project(S.filter_(S.scan([I32, STR]), S.eq(c(1, STR), L(region_name))), [c(0, I32)])
    n1 = S.project(S.hash_join(reg, S.scan([I32, STR, I32]), [c(0, I32)], [c(2, I32)], S.INNER, S.BUILD_LEFT), [c(1, I32)])
    cus = S.project(S.hash_join(n1, S.scan([I64, I32]), [c(0, I32)], [c(1, I32)], S.INNER, S.BUILD_LEFT), [c(1, I64)])
    o = S.filter_(S.scan([I64, I64, DATE]), S.and_(S.gt_eq(c(2, DATE), S.lit(d0, DATE)), S.lt_eq(c(2, DATE), S.lit(d1, DATE))))
    ordj = S.project(S.hash_join(cus, o, [c(0, I64)], [c(1, I64)], S.INNER, S.BUILD_LEFT), [c(1, I64), c(3, DATE)])                                           # o_orderkey, o_orderdate
    supn = S.project(S.hash_join(S.scan([I32, STR, I32]), S.scan([I64, I32]), [c(0, I32)], [c(1, I32)], S.INNER, S.BUILD_LEFT), [c(3, I64), c(1, STR)])       # s_suppkey, nation
    partf = S.project(S.filter_(S.scan([I64, STR, STR, STR, I32]), S.eq(c(1, STR), L(ptype))), [c(0, I64)])
    lp = S.project(S.hash_join(partf, S.scan([I64, I64, I64, D, D]), [c(0, I64)], [c(1, I64)], S.INNER, S.BUILD_LEFT), [c(1, I64), c(3, I64), c(4, D), c(5, D)])  # l_orderkey, l_suppkey, price, disc
    j1 = S.project(S.hash_join(supn, lp, [c(0, I64)], [c(1, I64)], S.INNER, S.BUILD_LEFT), [c(2, I64), c(4, D), c(5, D), c(1, STR)])                          # l_orderkey, price, disc, nation
    j2 = S.hash_join(ordj, j1, [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT)                              # o_orderkey, o_orderdate, l_orderkey, price, disc, nation
    R = S.decimal(26, 4)
    vol = _revenue(c(3, D), c(4, D))
    p = S.project(j2, [S.date_part("year", c(1, DATE)), S.case_when([(S.eq(c(5, STR), L(nation_name)), vol)], S.lit(decimal.Decimal("0.0000"), R)), vol])
    return S.hash_agg(p, [c(0, I32)], [S.sum_(c(1, R), S.decimal(36, 4)), S.sum_(c(2, R), S.decimal(36, 4))], S.PARTIAL)


def q8_final_plan(partial, state_schema):
    SD = S.decimal(36, 4)
    fin = S.final_of(partial, state_schema)
    share = S.check_overflow(S.math("divide", c(1, SD), c(2, SD), S.decimal(38, 6)), S.decimal(38, 6))
    return S.sort(S.project(fin, [c(0, I32), share]), [(c(0, I32), False, False)])


def q18_partial_plan(threshold="300.00"):
    """TPC-H Q18 up to the partial aggregate: orders whose lines add up to more than `threshold` items (a Final over a Partial aggregate of all
    of lineitem, filtered, inside the plan), joined to orders, customer and their lines.  Scan leaves in order: lineitem[l_orderkey, l_quantity],
    orders[o_orderkey, o_custkey, o_orderdate, o_totalprice], customer[c_custkey, c_name], lineitem[l_orderkey, l_quantity]"""
    D22 = S.decimal(22, 2)
    per_order = S.hash_agg(S.scan([I64, D]), [c(0, I64)], [S.sum_(c(1, D), D22)], S.PARTIAL)
    totals = S.hash_agg(per_order, [c(0, I64)], per_order.aggs, S.FINAL)                                     # l_orderkey, sum(l_quantity)
    big = S.project(S.filter_(totals, S.gt(c(1, D22), S.lit(decimal.Decimal(threshold), D22))), [c(0, I64)])
    jo = S.project(S.hash_join(big, S.scan([I64, I64, DATE, D]), [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT), [c(1, I64), c(2, I64), c(3, DATE), c(4, D)])     # o_orderkey, o_custkey, o_orderdate, o_totalprice
    jc = S.project(S.hash_join(jo, S.scan([I64, STR]), [c(1, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT), [c(5, STR), c(4, I64), c(0, I64), c(2, DATE), c(3, D)])    # c_name, c_custkey, o_orderkey, o_orderdate, o_totalprice
    jl = S.hash_join(jc, S.scan([I64, D]), [c(2, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT)                  # … l_orderkey, l_quantity
    return S.hash_agg(jl, [c(0, STR), c(1, I64), c(2, I64), c(3, DATE), c(4, D)], [S.sum_(c(6, D), D22)], S.PARTIAL)


def q18_final_plan(partial, state_schema):
    return S.sort(S.final_of(partial, state_schema), [(c(4, D), True, True), (c(3, DATE), False, False)], fetch=100)


def q21_partial_plan(nation_name="SAUDI ARABIA"):
    """TPC-H Q21 up to the partial aggregate: suppliers of a nation who alone kept a multi-supplier order of status 'F' waiting — a LeftSemi and a
    LeftAnti join on the order key, each with the residual condition "another supplier".  Scan leaves in order: nation, supplier[s_suppkey,
    s_nationkey, s_name], lineitem l1[l_orderkey, l_suppkey, l_commitdate, l_receiptdate], orders[o_orderkey, o_orderstatus],
    lineitem l2[l_orderkey, l_suppkey], lineitem l3[l_orderkey, l_suppkey, l_commitdate, l_receiptdate]"""
    nat = S.project(S.filter_(S.scan([I32, STR, I32]), S.eq(c(1, STR), L(nation_name))), [c(0, I32)])
    sup = S.project(S.hash_join(nat, S.scan([I64, I32, STR]), [c(0, I32)], [c(1, I32)], S.INNER, S.BUILD_LEFT), [c(1, I64), c(3, STR)])                       # s_suppkey, s_name
    late = lambda: S.project(S.filter_(S.scan([I64, I64, DATE, DATE]), S.gt(c(3, DATE), c(2, DATE))), [c(0, I64), c(1, I64)])                                # l_orderkey, l_suppkey of late lines
    j1 = S.project(S.hash_join(sup, late(), [c(0, I64)], [c(1, I64)], S.INNER, S.BUILD_LEFT), [c(1, STR), c(2, I64), c(3, I64)])                             # s_name, l_orderkey, l_suppkey
    of = S.project(S.filter_(S.scan([I64, STR]), S.eq(c(1, STR), L("F"))), [c(0, I64)])
    j2 = S.project(S.hash_join(j1, of, [c(1, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT), [c(0, STR), c(1, I64), c(2, I64)])                                  # s_name, l_orderkey, l_suppkey
    other = S.neq(c(4, I64), c(2, I64))                                                                       # (left ++ right: s_name, l_orderkey, l_suppkey | l_orderkey, l_suppkey)
    semi = S.hash_join(j2, S.scan([I64, I64]), [c(1, I64)], [c(0, I64)], S.LEFT_SEMI, S.BUILD_RIGHT, condition=other)
    anti = S.hash_join(semi, late(), [c(1, I64)], [c(0, I64)], S.LEFT_ANTI, S.BUILD_RIGHT, condition=other)
    return S.hash_agg(S.project(anti, [c(0, STR)]), [c(0, STR)], [S.count(S.lit(1, I32))], S.PARTIAL)


def q21_final_plan(partial, state_schema):
    return S.sort(S.final_of(partial, state_schema), [(c(1, I64), True, True), (c(0, STR), False, False)], fetch=100)


Q22_CODES = ["13", "31", "23", "29", "30", "18", "17"]


def _q22_code():
    return S.scalar_func("substring", [c(1, STR), S.lit(1, I32), S.lit(2, I32)], STR)


def q22_average_plan():
    """the scalar subquery of TPC-H Q22: the average positive balance of the customers of seven country codes; input: customer[c_custkey, c_phone, c_acctbal]"""
    f = S.filter_(S.scan([I64, STR, D]), S.and_(S.gt(c(2, D), S.lit(decimal.Decimal("0.00"), D)), S.in_(_q22_code(), [L(x) for x in Q22_CODES])))
    return S.hash_agg(f, [], [S.avg(c(2, D), S.decimal(16, 6), S.decimal(22, 2))], S.PARTIAL)


def q22_partial_plan(average):
    """TPC-H Q22 up to the partial aggregate: customers of those codes with a balance above `average` (a decimal(16,6) literal, what Spark
    puts in place of the subquery) and no orders (LeftAnti); inputs: customer[c_custkey, c_phone, c_acctbal], orders[o_custkey]"""
    A = S.decimal(16, 6)
    f = S.filter_(S.scan([I64, STR, D]), S.and_(S.in_(_q22_code(), [L(x) for x in Q22_CODES]), S.gt(S.cast(c(2, D), A), S.lit(average, A))))
    anti = S.hash_join(f, S.scan([I64]), [c(0, I64)], [c(0, I64)], S.LEFT_ANTI, S.BUILD_RIGHT)
    return S.hash_agg(S.project(anti, [_q22_code(), c(2, D)]), [c(0, STR)], [S.count(S.lit(1, I32)), S.sum_(c(1, D), S.decimal(22, 2))], S.PARTIAL)


def q17_partial_plan(brand="Brand#23", container="MED BOX"):
    """TPC-H Q17 up to the partial aggregate: lines of one brand's parts in one container whose quantity is below a fifth of the part's average
    quantity (a Final over a Partial average per part inside the plan, joined back on the part key with the comparison as the join's residual
    condition).  Scan leaves in order: part, lineitem[l_partkey, l_quantity] (for the averages), part, lineitem[l_partkey, l_quantity, l_extendedprice]"""
    A, T = S.decimal(16, 6), S.decimal(18, 7)
    parts = lambda: S.project(S.filter_(S.scan([I64, STR, STR, STR, I32]), S.and_(S.eq(c(2, STR), L(brand)), S.eq(c(3, STR), L(container)))), [c(0, I64)])
    lq = S.project(S.hash_join(parts(), S.scan([I64, D]), [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT), [c(1, I64), c(2, D)])                             # l_partkey, l_quantity
    per_part = S.hash_agg(lq, [c(0, I64)], [S.avg(c(1, D), A, S.decimal(22, 2))], S.PARTIAL)
    avgs = S.hash_agg(per_part, [c(0, I64)], per_part.aggs, S.FINAL)                                         # l_partkey, avg(l_quantity)
    fifth = S.check_overflow(S.math("multiply", S.lit(decimal.Decimal("0.2"), S.decimal(1, 1)), c(1, A), T), T)
    limit = S.project(avgs, [c(0, I64), fifth])                                                              # l_partkey, 0.2 * avg
    lines = S.project(S.hash_join(parts(), S.scan([I64, D, D]), [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT), [c(1, I64), c(2, D), c(3, D)])               # l_partkey, l_quantity, l_extendedprice
    small = S.lt(S.cast(c(3, D), T), c(1, T))                                                                # (left ++ right: l_partkey, limit | l_partkey, l_quantity, l_extendedprice)
    j = S.hash_join(limit, lines, [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_LEFT, condition=small)
    return S.hash_agg(S.project(j, [c(4, D)]), [], [S.sum_(c(0, D), S.decimal(22, 2))], S.PARTIAL)


def q17_final_plan(partial, state_schema):
    fin = S.final_of(partial, state_schema)
    Q = S.decimal(27, 6)
    return S.project(fin, [S.check_overflow(S.math("divide", c(0, S.decimal(22, 2)), S.lit(decimal.Decimal("7.0"), S.decimal(2, 1)), Q), Q)])


def q11_partial_plan(nation_name="GERMANY", grouped=True):
    """TPC-H Q11 up to the partial aggregate: the stock value (supply cost × available quantity) of a nation's suppliers, by part — or, not
    grouped, in total (the scalar subquery).  Scan leaves in order: nation, supplier[s_suppkey, s_nationkey], partsupp[ps_partkey, ps_suppkey,
    ps_availqty, ps_supplycost]"""
    nat = S.project(S.filter_(S.scan([I32, STR, I32]), S.eq(c(1, STR), L(nation_name))), [c(0, I32)])
    sup = S.project(S.hash_join(nat, S.scan([I64, I32]), [c(0, I32)], [c(1, I32)], S.INNER, S.BUILD_LEFT), [c(1, I64)])
    j = S.hash_join(sup, S.scan([I64, I64, I32, D]), [c(0, I64)], [c(1, I64)], S.INNER, S.BUILD_LEFT)          # s_suppkey, ps_partkey, ps_suppkey, ps_availqty, ps_supplycost
    V = S.decimal(23, 2)
    value = S.check_overflow(S.math("multiply", c(4, D), S.cast(c(3, I32), S.decimal(10, 0)), V), V)
    p = S.project(j, [c(1, I64), value])
    return S.hash_agg(p, [c(0, I64)] if grouped else [], [S.sum_(c(1, V), S.decimal(33, 2))], S.PARTIAL)


def q11_final_plan(partial, state_schema, threshold):
    """the parts whose value exceeds `threshold` (what Spark computes from the subquery: its total × 0.0001), most valuable first"""
    T = S.decimal(33, 2)
    return S.sort(S.filter_(S.final_of(partial, state_schema), S.gt(c(1, T), S.lit(threshold, T))), [(c(1, T), True, True)])


def q19_partial_plan(modes=("AIR", "REG AIR")):
    """TPC-H Q19 up to the partial aggregate; inputs: lineitem (the LI layout), part[p_partkey, p_type, p_brand, p_container, p_size].  (The
    benchmark's text asks for the modes 'AIR' and 'AIR REG'; no row carries the latter.)"""
    li = S.project(S.filter_(S.scan(LI), S.and_(S.in_(c(8, STR), [L(m) for m in modes]), S.eq(c(9, STR), L("DELIVER IN PERSON")))), [c(1, I64), c(2, D), c(3, D), c(4, D)])
    j = S.hash_join(li, S.scan([I64, STR, STR, STR, I32]), [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_RIGHT)   # partkey, qty, price, disc | p_partkey, type, brand, container, size
    qty, brand, cont, size = c(1, D), c(6, STR), c(7, STR), c(8, I32)
    dq = lambda v: S.lit(decimal.Decimal(v), D)
    between = lambda x, lo, hi: S.and_(S.gt_eq(x, lo), S.lt_eq(x, hi))
    branch = lambda b, conts, q0, q1, s1: S.and_(S.and_(S.eq(brand, L(b)), S.in_(cont, [L(x) for x in conts])), S.and_(between(qty, dq(q0), dq(q1)), between(size, S.lit(1, I32), S.lit(s1, I32))))
    cond = S.or_(S.or_(branch("Brand#12", ["SM CASE", "SM BOX", "SM PACK", "SM PKG"], "1.00", "11.00", 5), branch("Brand#23", ["MED BAG", "MED BOX", "MED PKG", "MED PACK"], "10.00", "20.00", 10)),
                 branch("Brand#34", ["LG CASE", "LG BOX", "LG PACK", "LG PKG"], "20.00", "30.00", 15))
    return S.hash_agg(S.project(S.filter_(j, cond), [_revenue(c(2, D), c(3, D))]), [], [S.sum_(c(0, S.decimal(26, 4)), S.decimal(36, 4))], S.PARTIAL)


def test_q19_discounted_revenue(built):
    from oracle import oracle as O
    _, lineitem, part = _tables()
    partial = q19_partial_plan()
    st = run(partial, [lineitem, part], 2)
    got = run(S.final_of(partial, st.schema), [st], 1)
    assert rows(st) == rows(O.run_plan_to_arrow(S, partial, [lineitem, part]))
    pinfo = {r[0]: r for r in rows(part)}
    tot, any_ = decimal.Decimal(0), False
    for pk, q, price, disc, m, ins in zip(*[lineitem.column(i).to_pylist() for i in (1, 2, 3, 4, 8, 9)]):
        if m in ("AIR", "REG AIR") and ins == "DELIVER IN PERSON":
            _, _, b, ct, sz = pinfo[pk]
            if ((b == "Brand#12" and ct in ("SM CASE", "SM BOX", "SM PACK", "SM PKG") and 1 <= q <= 11 and 1 <= sz <= 5) or
                    (b == "Brand#23" and ct in ("MED BAG", "MED BOX", "MED PKG", "MED PACK") and 10 <= q <= 20 and 1 <= sz <= 10) or
                    (b == "Brand#34" and ct in ("LG CASE", "LG BOX", "LG PACK", "LG PKG") and 20 <= q <= 30 and 1 <= sz <= 15)):
                tot += price * (1 - disc)
                any_ = True
    assert got.column(0).to_pylist() == [tot if any_ else None] and any_


# ---- plans run against the reference's SF1 answers only (tests/test_tpch_golden_cpu.py on the oracle, tests/test_tpch_golden_gpu.py on the GPU) ----

def q9_partial_plan(word="green"):
    """TPC-H Q9 up to the partial aggregate: the profit on the parts whose name holds a word, by the supplier's nation and the order's year.  Scan leaves in
    order: nation, supplier[s_suppkey, s_nationkey], part[p_partkey, p_name], lineitem[l_orderkey, l_partkey, l_suppkey, l_quantity, l_extendedprice,
    l_discount], partsupp[ps_partkey, ps_suppkey, ps_availqty, ps_supplycost], orders[o_orderkey, o_orderdate]"""
    parts = S.project(S.filter_(S.scan([I64, STR]), S.like(c(1, STR), L("%" + word + "%"))), [c(0, I64)])
    lp = S.project(S.hash_join(parts, S.scan([I64, I64, I64, D, D, D]), [c(0, I64)], [c(1, I64)], S.INNER, S.BUILD_LEFT),
                   [c(1, I64), c(2, I64), c(3, I64), c(4, D), c(5, D), c(6, D)])                             # l_orderkey, l_partkey, l_suppkey, qty, price, disc
    supn = S.project(S.hash_join(S.scan([I32, STR, I32]), S.scan([I64, I32]), [c(0, I32)], [c(1, I32)], S.INNER, S.BUILD_LEFT), [c(3, I64), c(1, STR)])        # s_suppkey, nation
    j1 = S.project(S.hash_join(supn, lp, [c(0, I64)], [c(2, I64)], S.INNER, S.BUILD_LEFT), [c(2, I64), c(3, I64), c(4, I64), c(5, D), c(6, D), c(7, D), c(1, STR)])
    # l_orderkey, l_partkey, l_suppkey, qty, price, disc, nation | ps_partkey, ps_suppkey, ps_availqty, ps_supplycost
    j2 = S.project(S.hash_join(j1, S.scan([I64, I64, I32, D]), [c(1, I64), c(2, I64)], [c(0, I64), c(1, I64)], S.INNER, S.BUILD_RIGHT),
                   [c(0, I64), c(3, D), c(4, D), c(5, D), c(6, STR), c(10, D)])                              # l_orderkey, qty, price, disc, nation, supplycost
    j3 = S.hash_join(j2, S.scan([I64, DATE]), [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_RIGHT)               # … o_orderkey, o_orderdate
    cost = S.check_overflow(S.math("multiply", c(5, D), c(1, D), S.decimal(25, 4)), S.decimal(25, 4))
    A = S.decimal(27, 4)
    amount = S.check_overflow(S.math("subtract", _revenue(c(2, D), c(3, D)), cost, A), A)
    p = S.project(j3, [c(4, STR), S.date_part("year", c(7, DATE)), amount])
    return S.hash_agg(p, [c(0, STR), c(1, I32)], [S.sum_(c(2, A), S.decimal(37, 4))], S.PARTIAL)


def q9_final_plan(partial, state_schema):
    return S.sort(S.final_of(partial, state_schema), [(c(0, STR), False, False), (c(1, I32), True, True)])


def q15_revenue_plan(d0, d1):
    """the view of TPC-H Q15 up to the partial aggregate: revenue by supplier over a quarter; input: lineitem[l_suppkey, l_extendedprice, l_discount, l_shipdate]"""
    f = S.filter_(S.scan([I64, D, D, DATE]), S.and_(S.gt_eq(c(3, DATE), S.lit(d0, DATE)), S.lt(c(3, DATE), S.lit(d1, DATE))))
    return S.hash_agg(S.project(f, [c(0, I64), _revenue(c(1, D), c(2, D))]), [c(0, I64)], [S.sum_(c(1, S.decimal(26, 4)), S.decimal(36, 4))], S.PARTIAL)


def q15_max_plan():
    """the scalar subquery of Q15: the largest revenue; input: the view's rows [supplier_no, total_revenue]"""
    R = S.decimal(36, 4)
    return S.hash_agg(S.scan([I64, R]), [], [S.max_(c(1, R), R)], S.PARTIAL)


def q15_top_plan(best):
    """Q15's outer query: the supplier(s) whose revenue equals `best` (what Spark puts in place of the subquery), by key.  Scan leaves in order:
    supplier[s_suppkey, s_name, s_address, s_phone], the view's rows [supplier_no, total_revenue]"""
    R = S.decimal(36, 4)
    top = S.filter_(S.scan([I64, R]), S.eq(c(1, R), S.lit(best, R)))
    j = S.hash_join(S.scan([I64, STR, STR, STR]), top, [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_RIGHT)
    return S.sort(S.project(j, [c(0, I64), c(1, STR), c(2, STR), c(3, STR), c(5, R)]), [(c(0, I64), False, False)])


def q16_partial_plan(brand="Brand#45", type_prefix="MEDIUM POLISHED", sizes=(49, 14, 23, 45, 19, 3, 36, 9)):
    """TPC-H Q16 up to the partial aggregate of the outer count: suppliers without complaints (LeftAnti) per brand, type and size, each counted once — the
    distinct count as Spark plans it, an aggregate over (brand, type, size, supplier) without functions under the counting one.  Scan leaves in order:
    partsupp[ps_partkey, ps_suppkey], supplier[s_suppkey, s_complaints] (the outcome of s_comment LIKE '%Customer%Complaints%', see dbgen.supplier),
    part[p_partkey, p_type, p_brand, p_container, p_size]"""
    bad = S.project(S.filter_(S.scan([I64, S.T_BOOL]), c(1, S.T_BOOL)), [c(0, I64)])
    ps = S.hash_join(S.scan([I64, I64]), bad, [c(1, I64)], [c(0, I64)], S.LEFT_ANTI, S.BUILD_RIGHT)
    pf = S.filter_(S.scan([I64, STR, STR, STR, I32]), S.and_(S.and_(S.neq(c(2, STR), L(brand)), S.not_(S.like(c(1, STR), L(type_prefix + "%")))),
                                                              S.in_(c(4, I32), [S.lit(int(x), I32) for x in sizes])))
    j = S.hash_join(ps, S.project(pf, [c(0, I64), c(2, STR), c(1, STR), c(4, I32)]), [c(0, I64)], [c(0, I64)], S.INNER, S.BUILD_RIGHT)   # ps_partkey, ps_suppkey | p_partkey, brand, type, size
    keys = S.project(j, [c(3, STR), c(4, STR), c(5, I32), c(1, I64)])
    once_p = S.hash_agg(keys, [c(0, STR), c(1, STR), c(2, I32), c(3, I64)], [], S.PARTIAL)
    once = S.hash_agg(once_p, [c(0, STR), c(1, STR), c(2, I32), c(3, I64)], [], S.FINAL)
    return S.hash_agg(once, [c(0, STR), c(1, STR), c(2, I32)], [S.count(c(3, I64))], S.PARTIAL)


def q16_final_plan(partial, state_schema):
    return S.sort(S.final_of(partial, state_schema), [(c(3, I64), True, True), (c(0, STR), False, False), (c(1, STR), False, False), (c(2, I32), False, False)])


def q20_plan(d0, d1, prefix="forest", nation_name="CANADA"):
    """TPC-H Q20: the suppliers of a nation holding more of a part whose name starts with a word than half of what they shipped of it in a year (the
    correlated sum: a Final over a Partial aggregate by part and supplier inside the plan, joined back with the comparison as the residual condition).
    Scan leaves in order: nation, supplier[s_suppkey, s_nationkey, s_name, s_address], partsupp[ps_partkey, ps_suppkey, ps_availqty, ps_supplycost],
    part[p_partkey, p_name], lineitem[l_partkey, l_suppkey, l_quantity, l_shipdate]"""
    D22, H = S.decimal(22, 2), S.decimal(24, 3)
    nat = S.project(S.filter_(S.scan([I32, STR, I32]), S.eq(c(1, STR), L(nation_name))), [c(0, I32)])
    sup = S.project(S.hash_join(nat, S.scan([I64, I32, STR, STR]), [c(0, I32)], [c(1, I32)], S.INNER, S.BUILD_LEFT), [c(1, I64), c(3, STR), c(4, STR)])       # s_suppkey, s_name, s_address
    parts = S.project(S.filter_(S.scan([I64, STR]), S.like(c(1, STR), L(prefix + "%"))), [c(0, I64)])
    stock = S.project(S.hash_join(S.scan([I64, I64, I32, D]), parts, [c(0, I64)], [c(0, I64)], S.LEFT_SEMI, S.BUILD_RIGHT), [c(0, I64), c(1, I64), c(2, I32)])   # ps_partkey, ps_suppkey, ps_availqty
    f = S.filter_(S.scan([I64, I64, D, DATE]), S.and_(S.gt_eq(c(3, DATE), S.lit(d0, DATE)), S.lt(c(3, DATE), S.lit(d1, DATE))))
    sold_p = S.hash_agg(S.project(f, [c(0, I64), c(1, I64), c(2, D)]), [c(0, I64), c(1, I64)], [S.sum_(c(2, D), D22)], S.PARTIAL)
    sold = S.hash_agg(sold_p, [c(0, I64), c(1, I64)], sold_p.aggs, S.FINAL)                                   # l_partkey, l_suppkey, sum(l_quantity)
    half = S.project(sold, [c(0, I64), c(1, I64), S.check_overflow(S.math("multiply", S.lit(decimal.Decimal("0.5"), S.decimal(1, 1)), c(2, D22), H), H)])
    more = S.gt(S.cast(S.cast(c(2, I32), S.decimal(10, 0)), H), c(5, H))                                      # (left ++ right: ps_partkey, ps_suppkey, ps_availqty | l_partkey, l_suppkey, half)
    holders = S.project(S.hash_join(stock, half, [c(0, I64), c(1, I64)], [c(0, I64), c(1, I64)], S.INNER, S.BUILD_RIGHT, condition=more), [c(1, I64)])
    return S.project(S.hash_join(sup, holders, [c(0, I64)], [c(0, I64)], S.LEFT_SEMI, S.BUILD_RIGHT), [c(1, STR), c(2, STR)])


def q20_sort_plan():
    return S.sort(S.scan([STR, STR]), [(c(0, STR), False, False)])
