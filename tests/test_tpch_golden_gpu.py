"""The GPU path against the reference's OWN golden answers: TPC-H scale factor 1, Q1 / Q3 / Q6 / Q4 / Q5 / Q7 / Q8 / Q9 / Q11 / Q12 / Q14 / Q15 / Q16 / Q17 / Q18 / Q19 / Q20 / Q21 / Q22 — the numbers in
spark/src/test/resources/tpch-query-results/q{1,3,4,5,6,7,8,9,11,12,14,15,16,17,18,19,20,21,22}.sql.out of apache/datafusion-comet (copies under tests/golden/tpch_sf1/), over tables
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


@pytest.mark.parametrize("q", ["q4", "q5", "q7", "q8", "q11", "q12", "q14", "q17", "q18", "q19", "q21", "q22"])
def test_more_queries_give_the_references_answers(built, sf1, q):
    from tests import test_tpch_more_gpu as M
    from tests.test_tpch_golden_cpu import _more_inputs, golden_case
    ncols = {"q4": (2, 2), "q5": (3, 2), "q7": (5, 4), "q8": (5, 2), "q11": (3, 2), "q12": (3, 3), "q14": (4, 1), "q17": (2, 1), "q18": (7, 6), "q19": (2, 1), "q21": (2, 2), "q22": (4, 3)}[q]
    state = {"n": 0}

    def run_partial(plan, tables):
        # (the first partial plan of Q22 is its scalar subquery: an ungrouped average, two state columns)
        n = 2 if (q in ("q22", "q11") and state["n"] == 0) else ncols[0]
        state["n"] += 1
        return M.run(plan, tables, n)

    def run_final(plan, tables):
        n = 1 if (q in ("q22", "q11") and state["n"] == 1) else ncols[1]
        return M.run(plan, tables, n)
    got, want = golden_case(q, _more_inputs(sf1), run_partial, run_final)
    assert got == want


@pytest.mark.parametrize("q", ["q9", "q15", "q16", "q20"])
def test_queries_with_generated_names_and_addresses_give_the_references_answers(built, sf1, q):
    """Q9 (part names), Q15 (a view aggregated twice, printed addresses and phones), Q16 (a distinct count as two aggregates, 18 314 rows) and Q20 (a correlated
    sum joined back); output columns of each stage in the order golden_case runs them"""
    from tests import test_tpch_more_gpu as M
    from tests.test_tpch_golden_cpu import _more_inputs, golden_case
    partial_cols, final_cols = {"q9": ([4], [3]), "q15": ([3, 1], [2, 1, 5]), "q16": ([4], [4]), "q20": ([2], [2])}[q]
    calls = {"p": 0, "f": 0}

    def run_partial(plan, tables):
        calls["p"] += 1
        return M.run(plan, tables, partial_cols[calls["p"] - 1])

    def run_final(plan, tables):
        calls["f"] += 1
        return M.run(plan, tables, final_cols[calls["f"] - 1])
    got, want = golden_case(q, _more_inputs(sf1), run_partial, run_final)
    assert got == want
