"""GPU parity for BASELINE config 1: ProjectExec + FilterExec over a 1M-row (int64, float64) batch, and the
reference's own planner test shape (`col = 3` over n % 4, planner.rs:4652-4660 → 25 of 100 rows).
FilterExec preserves row order, so outputs are compared position by position, bit for bit."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu


def _oracle(plan, table):
    from oracle import oracle as O
    return O.run_plan_to_arrow(S, plan, table)


def _run(plan, table, ncols, **kw):
    out = native.execute_to_table([native.HostInput.from_table(table)], ncols, plan.encode(), **kw)
    return out


def _config1_table(n, nulls=False, seed=42):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1_000_000, n, dtype=np.int64)
    b = rng.random(n)
    if not nulls:
        return pa.table({"a": pa.array(a), "b": pa.array(b)})
    r2 = np.random.default_rng(43)
    return pa.table({"a": pa.array(a, mask=r2.random(n) < 0.1), "b": pa.array(b, mask=r2.random(n) < 0.1)})


def _config1_plan():
    a, b = S.col(0, S.T_INT64), S.col(1, S.T_DOUBLE)
    f = S.filter_(S.scan([S.T_INT64, S.T_DOUBLE]), S.and_(S.lt(a, S.lit(500_000, S.T_INT64)), S.is_not_null(b)))
    return S.project(f, [S.math("add", a, S.lit(1, S.T_INT64), S.T_INT64), S.math("multiply", b, S.lit(2.0, S.T_DOUBLE), S.T_DOUBLE), a])


@pytest.mark.parametrize("nulls", [False, True])
def test_config1_project_filter_1m_rows(built, nulls):
    table = _config1_table(1_000_000, nulls)
    plan = _config1_plan()
    batches = _run(plan, table, 3)
    assert all(b.num_rows <= 8192 for b in batches)          # output batches respect spark.comet.batchSize
    got = pa.Table.from_batches(batches)
    want = _oracle(plan, table)
    assert got.num_rows == want.num_rows
    for i in range(3):
        assert got.column(i).combine_chunks().equals(want.column(i).combine_chunks()), f"column {i}"


def test_reference_planner_case_col_eq_3(built):
    # planner.rs:4637-4699 test_unpack_dictionary_primitive expects 25 of 100 rows from `col = 3` over n % 4
    table = pa.table({"c": pa.array([i % 4 for i in range(100)], pa.int32())})
    plan = S.filter_(S.scan([S.T_INT32]), S.eq(S.col(0, S.T_INT32), S.lit(3, S.T_INT32)))
    got = pa.Table.from_batches(_run(plan, table, 1))
    assert got.num_rows == 25
    assert got.column(0).to_pylist() == [3] * 25


def test_empty_input_gives_empty_output(built):
    # planner.rs:4778-4800: empty input → end of stream without batches
    table = pa.table({"c": pa.array([], pa.int32())})
    plan = S.filter_(S.scan([S.T_INT32]), S.eq(S.col(0, S.T_INT32), S.lit(3, S.T_INT32)))
    assert _run(plan, table, 1) == []


def test_filter_keeps_only_true_and_valid(built):
    # three-valued logic: NULL predicate rows are dropped, OR with a TRUE side survives a NULL side
    x = pa.array([1, None, 3, None, 5, 6], pa.int32())
    y = pa.array([None, 2, 3, None, 0, 7], pa.int32())
    table = pa.table({"x": x, "y": y})
    cx, cy = S.col(0, S.T_INT32), S.col(1, S.T_INT32)
    pred = S.or_(S.gt(cx, S.lit(4, S.T_INT32)), S.eq(cy, S.lit(3, S.T_INT32)))
    plan = S.filter_(S.scan([S.T_INT32, S.T_INT32]), pred)
    got = pa.Table.from_batches(_run(plan, table, 2))
    want = _oracle(plan, table)
    assert got.column(0).to_pylist() == want.column(0).to_pylist() == [3, 5, 6]
    assert got.column(1).to_pylist() == want.column(1).to_pylist()


def test_projection_only_int_wrapping_and_float(built):
    n = 70_001
    rng = np.random.default_rng(1)
    a = rng.integers(-2**63, 2**63 - 1, n, dtype=np.int64)
    b = rng.standard_normal(n) * 1e300
    table = pa.table({"a": pa.array(a), "b": pa.array(b)})
    ca, cb = S.col(0, S.T_INT64), S.col(1, S.T_DOUBLE)
    plan = S.project(S.scan([S.T_INT64, S.T_DOUBLE]),
                     [S.math("multiply", ca, S.lit(3, S.T_INT64), S.T_INT64),      # LEGACY: wraps
                      S.math("add", S.math("multiply", cb, cb, S.T_DOUBLE), cb, S.T_DOUBLE),  # must NOT contract into an FMA
                      S.math("divide", cb, S.lit(0.0, S.T_DOUBLE), S.T_DOUBLE)])
    got = pa.Table.from_batches(_run(plan, table, 3))
    want = _oracle(plan, table)
    for i in range(3):
        g, w = got.column(i).combine_chunks(), want.column(i).combine_chunks()
        assert g.to_numpy(zero_copy_only=False).tobytes() == w.to_numpy(zero_copy_only=False).tobytes(), f"column {i}"


def test_decimal_projection_narrow_and_wide(built):
    from datafusion_comet_amd import tpch
    table = tpch.lineitem_q1(50_000, seed=12).select([1, 2, 3])   # price, disc, tax
    DEC = S.decimal(12, 2)
    price, disc, tax = (S.col(i, DEC) for i in range(3))
    one = S.lit(100, DEC)
    om = S.check_overflow(S.math("subtract", one, disc, S.decimal(13, 2)), S.decimal(13, 2))
    op = S.check_overflow(S.math("add", one, tax, S.decimal(13, 2)), S.decimal(13, 2))
    dp = S.check_overflow(S.math("multiply", price, om, S.decimal(26, 4)), S.decimal(26, 4))
    ch = S.check_overflow(S.math("multiply", dp, op, S.decimal(38, 6)), S.decimal(38, 6))
    plan = S.project(S.scan([DEC, DEC, DEC]), [dp, ch])
    got = pa.Table.from_batches(_run(plan, table, 2))
    want = _oracle(plan, table)
    assert got.column(0).combine_chunks().equals(want.column(0).combine_chunks())
    assert got.column(1).combine_chunks().equals(want.column(1).combine_chunks())
    assert got.schema.field(1).type == pa.decimal128(38, 6)
