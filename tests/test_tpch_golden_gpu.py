"""The GPU path against the reference's OWN golden answers: TPC-H scale factor 1, Q1 / Q3 / Q6 / Q4 / Q5 / Q7 / Q8 / Q12 / Q14 / Q18 / Q19 — the numbers in
spark/src/test/resources/tpch-query-results/q{1,3,4,5,6,7,8,12,14,18,19}.sql.out of apache/datafusion-comet (copies under tests/golden/tpch_sf1/), over tables
regenerated with dbgen's random streams (datafusion-comet_amd/dbgen.py; tests/test_tpch_golden_cpu.py pins the generator and the oracle on
the same files).  Q6 goes in through Parquet (snappy and zstd, pages inflated on the device, and the host path), Q1 and Q3 over
HBM-resident columns; every stage runs through the C ABI, the Final aggregates included."""
import os

import pyarrow as pa
import pyarrow.parquet as papq
import pytest

from datafusion_comet_amd import dbgen, native, parallel, serde as S, tpch
from tests.test_tpch_golden_cpu import GOLD, more_layout, q1_rows, q3_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sf1():
    orders, lineitem = dbgen.orders_and_lineitem(1)
    return dbgen.customer(1), orders, lineitem


def _final(plan, partial, ncols):
    out = native.execute_to_table([native.HostInput.from_table(partial)], ncols, S.final_of(plan, partial.schema).encode(), batch_size=0)
    return pa.Table.from_batches(out)


@pytest.mark.parametrize("codec,device", [("snappy", "true"), ("zstd", "true"), ("snappy", "false")])
def test_q6_from_parquet_gives_the_references_answer(built, sf1, tmp_path, codec, device):
    t = sf1[2].select(["l_quantity", "l_extendedprice", "l_discount", "l_shipdate"])
    path = str(tmp_path / f"lineitem_sf1_{codec}.parquet")
    papq.write_table(t, path, row_group_size=1 << 20, compression=codec, store_decimal_as_integer=True, data_page_size=1 << 20)
    src = S.native_scan([path], t.schema.names, [tpch.DEC, tpch.DEC, tpch.DEC, S.T_DATE])
    plan = tpch.q6_plan(source=src)
    it = native.CometExecIterator([], tpch.Q6_NUM_OUTPUT_COLS, plan.encode(), batch_size=0, config=S.config_map({"spark.comet.gpu.scan.deviceDecompress": device}))
    partial = pa.Table.from_batches([native.Native.executePlan(it.handle, tpch.Q6_NUM_OUTPUT_COLS)])
    m = S.decode_metric_node(it.metrics())
    it.close()
    while m[1]:
        m = m[1][0]
    assert (m[0]["pages_decompressed_on_device"] > 0) == (device == "true")
    final = _final(tpch.q6_plan(), partial, 1)
    assert [[str(final.column(0)[0].as_py())]] == dbgen.parse_golden(os.path.join(GOLD, "q6.sql.out"))      # 123141078.2283


def test_q1_gives_the_references_answer(built, sf1):
    t = sf1[2].select(["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"])
    dev = native.DeviceTable.from_arrow(t)
    partial = pa.Table.from_batches(native.execute_to_table([native.DeviceInput(dev)], tpch.Q1_NUM_OUTPUT_COLS, tpch.q1_plan().encode(), batch_size=0))
    final = _final(tpch.q1_plan(), partial, 10)
    assert q1_rows(final) == dbgen.parse_golden(os.path.join(GOLD, "q1.sql.out"))       # every sum, every average, to the last digit


def test_q3_gives_the_references_answer(built, sf1):
    customer, orders, lineitem = sf1
    li = lineitem.select(["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
    orders = orders.select(["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    customer = customer.select(["c_custkey", "c_mktsegment"])
    engine = parallel.GpuEngine(0)
    top, groups = parallel.run_q3_single(engine, native.DeviceTable.from_arrow(customer), native.DeviceTable.from_arrow(orders), native.DeviceTable.from_arrow(li))
    assert groups == 11620                                                                # rows of the full Q3 answer at SF1 (TPC-H answer set)
    assert q3_rows(top) == dbgen.parse_golden(os.path.join(GOLD, "q3.sql.out"))


def test_q4_q5_q7_q8_q12_q14_q18_q19_give_the_references_answers(built, sf1):
    from tests import test_tpch_more_gpu as M
    _, orders, lineitem = sf1
    o2, li, pt = more_layout(orders, lineitem, dbgen.part(1))
    partial = M.q12_partial_plan()
    st = M.run(partial, [o2, li], 3)
    final = M.run(M.q12_final_plan(partial, st.schema), [st], 3)
    assert [[str(v) for v in r] for r in M.rows(final)] == dbgen.parse_golden(os.path.join(GOLD, "q12.sql.out"))      # MAIL 6202 9324 / SHIP 6200 9262
    partial = M.q14_partial_plan(tpch.days(1995, 9, 1), tpch.days(1995, 10, 1))
    st = M.run(partial, [li, pt], 4)
    final = M.run(M.q14_final_plan(partial, st.schema), [st], 1)
    assert [[str(final.column(0)[0].as_py())]] == dbgen.parse_golden(os.path.join(GOLD, "q14.sql.out"))                # 16.380779
    partial = M.q19_partial_plan(("AIR", "AIR REG"))
    st = M.run(partial, [li, pt], 2)
    final = M.run(S.final_of(partial, st.schema), [st], 1)
    assert [[str(final.column(0)[0].as_py())]] == dbgen.parse_golden(os.path.join(GOLD, "q19.sql.out"))                # 3083843.0578
    partial = M.q4_partial_plan(tpch.days(1993, 7, 1), tpch.days(1993, 10, 1))
    o4, l4 = orders.select(["o_orderkey", "o_orderdate", "o_orderpriority"]), lineitem.select(["l_orderkey", "l_commitdate", "l_receiptdate"])
    st = M.run(partial, [o4, l4], 2)
    final = M.run(M.q12_final_plan(partial, st.schema), [st], 2)
    assert [[str(v) for v in r] for r in M.rows(final)] == dbgen.parse_golden(os.path.join(GOLD, "q4.sql.out"))       # five priorities, ≈ 10 500 orders each
    customer = sf1[0]
    q5_in = [dbgen.region(), dbgen.nation(), customer.select(["c_custkey", "c_nationkey"]), orders.select(["o_orderkey", "o_custkey", "o_orderdate"]),
             lineitem.select(["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount"]), dbgen.supplier(1)]
    partial = M.q5_partial_plan(tpch.days(1994, 1, 1), tpch.days(1995, 1, 1))
    st = M.run(partial, q5_in, 3)
    final = M.run(M.q5_final_plan(partial, st.schema), [st], 2)
    assert [[str(v) for v in r] for r in M.rows(final)] == dbgen.parse_golden(os.path.join(GOLD, "q5.sql.out"))       # five Asian nations by revenue
    cn, sp = customer.select(["c_custkey", "c_nationkey"]), dbgen.supplier(1)
    q7_in = [dbgen.nation(), cn, orders.select(["o_orderkey", "o_custkey"]), dbgen.nation(), sp,
             lineitem.select(["l_orderkey", "l_suppkey", "l_extendedprice", "l_discount", "l_shipdate"])]
    q8_in = [dbgen.region(), dbgen.nation(), cn, orders.select(["o_orderkey", "o_custkey", "o_orderdate"]), dbgen.nation(), sp, pt,
             lineitem.select(["l_orderkey", "l_partkey", "l_suppkey", "l_extendedprice", "l_discount"])]
    partial = M.q7_partial_plan(tpch.days(1995, 1, 1), tpch.days(1996, 12, 31))
    st = M.run(partial, q7_in, 5)
    final = M.run(M.q7_final_plan(partial, st.schema), [st], 4)
    assert [[str(v) for v in r] for r in M.rows(final)] == dbgen.parse_golden(os.path.join(GOLD, "q7.sql.out"))       # FRANCE ↔ GERMANY, 1995 and 1996
    partial = M.q8_partial_plan(tpch.days(1995, 1, 1), tpch.days(1996, 12, 31))
    st = M.run(partial, q8_in, 5)
    final = M.run(M.q8_final_plan(partial, st.schema), [st], 2)
    assert [[str(v) for v in r] for r in M.rows(final)] == dbgen.parse_golden(os.path.join(GOLD, "q8.sql.out"))       # 1995 0.034436 / 1996 0.041486
    lq = lineitem.select(["l_orderkey", "l_quantity"])
    q18_in = [lq, orders.select(["o_orderkey", "o_custkey", "o_orderdate", "o_totalprice"]), customer.select(["c_custkey", "c_name"]), lq]
    partial = M.q18_partial_plan()
    st = M.run(partial, q18_in, 7)
    final = M.run(M.q18_final_plan(partial, st.schema), [st], 6)
    import re      # (the reference's suite writes every "#<digits>" as "#x" into its result files: CometTPCHQuerySuite's normalisation)
    assert [[re.sub(r"#\d+", "#x", str(v)) for v in r] for r in M.rows(final)] == dbgen.parse_golden(os.path.join(GOLD, "q18.sql.out"))      # the 57 orders of more than 300 items
