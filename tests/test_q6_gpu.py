"""GPU parity: TPC-H Q6 stage 1 (scan→filter→project→partial SumDecimal) through the C ABI vs the oracle.
Integer/decimal work must be bit-exact."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu


def _oracle(plan, table):
    from oracle import oracle as O
    return O.run_plan_to_arrow(S, plan, table)


def _run(plan, inputs, ncols, **kw):
    out = native.execute_to_table(inputs, ncols, plan.encode(), **kw)
    return pa.Table.from_batches(out) if out else None


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 8192, 100_003, 1 << 20])
def test_q6_host_stream_matches_oracle(built, n):
    table = tpch.lineitem_q6(n, seed=n)
    plan = tpch.q6_plan()
    got = _run(plan, [native.HostInput.from_table(table)], tpch.Q6_NUM_OUTPUT_COLS)
    want = _oracle(plan, table)
    assert got.num_rows == 1
    assert got.column(0).to_pylist() == want.column(0).to_pylist()
    assert got.column(1).to_pylist() == want.column(1).to_pylist()
    assert got.schema.field(0).type == pa.decimal128(35, 4)


def test_q6_device_resident_matches_oracle(built):
    table = tpch.lineitem_q6(3_000_017, seed=11)
    plan = tpch.q6_plan()
    dev = native.DeviceTable.from_arrow(table, "cuda:0")
    got = _run(plan, [native.DeviceInput(dev)], tpch.Q6_NUM_OUTPUT_COLS)
    want = _oracle(plan, table)
    assert got.column(0).to_pylist() == want.column(0).to_pylist()
    assert got.column(1).to_pylist() == [False]


def test_q6_with_nulls_matches_oracle(built):
    table = tpch.lineitem_q6(200_000, seed=5, null_frac=0.1)
    plan = tpch.q6_plan()
    got = _run(plan, [native.HostInput.from_table(table, batch_rows=4096)], tpch.Q6_NUM_OUTPUT_COLS)
    want = _oracle(plan, table)
    assert got.column(0).to_pylist() == want.column(0).to_pylist()
    assert got.column(1).to_pylist() == want.column(1).to_pylist()


def test_q6_empty_input_emits_one_state_row(built):
    table = tpch.lineitem_q6(0)
    plan = tpch.q6_plan()
    got = _run(plan, [native.HostInput.from_table(table)], tpch.Q6_NUM_OUTPUT_COLS)
    # SumDecimal partial state of no rows: (0, is_empty = true)  (sum_decimal.rs:185-197)
    assert got.num_rows == 1
    assert got.column(0).to_pylist()[0] == 0
    assert got.column(1).to_pylist() == [True]


def test_q6_nothing_passes_filter(built):
    table = tpch.lineitem_q6(50_000, seed=3)
    ship = pa.array(np.full(50_000, tpch.days(1999, 1, 1), np.int32), pa.int32()).cast(pa.date32())
    table = table.set_column(3, "l_shipdate", ship)
    got = _run(tpch.q6_plan(), [native.HostInput.from_table(table)], tpch.Q6_NUM_OUTPUT_COLS)
    assert got.column(1).to_pylist() == [True]


def test_q6_chunked_execution_equals_single_chunk(built):
    table = tpch.lineitem_q6(300_000, seed=21)
    plan = tpch.q6_plan()
    cfg = S.config_map({"spark.comet.gpu.chunkRows": 20_000})
    a = _run(plan, [native.HostInput.from_table(table)], 2, config=cfg)
    b = _run(plan, [native.HostInput.from_table(table)], 2)
    assert a.column(0).to_pylist() == b.column(0).to_pylist()


def test_column_count_mismatch_is_an_error(built):
    table = tpch.lineitem_q6(100)
    with pytest.raises(native.CometNativeException, match="column count mismatch"):
        _run(tpch.q6_plan(), [native.HostInput.from_table(table)], 3)


def test_metrics_tree_mirrors_operator_tree(built):
    # NativeMetricNode tree has one node per Operator (metrics/utils.rs:30-45): agg <- project <- filter <- scan
    table = tpch.lineitem_q6(10_000)
    it = native.CometExecIterator([native.HostInput.from_table(table)], 2, tpch.q6_plan().encode())
    batches = []
    while True:
        b = native.Native.executePlan(it.handle, 2)
        if b is None:
            break
        batches.append(b)
    metrics, children = S.decode_metric_node(it.metrics())
    it.close()
    assert metrics["output_rows"] == 1 and metrics["elapsed_compute"] > 0
    depth = 0
    while children:
        assert len(children) == 1
        _, children = children[0]
        depth += 1
    assert depth == 3
