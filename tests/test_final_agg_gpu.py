"""GPU parity for Final-mode aggregation (merge_batch + evaluate of each accumulator): per-shard Partial plans on
the GPU, their state rows concatenated (what the exchange between Spark stages does), then the Final plan on
the GPU — compared with the oracle's Final over the same states AND with the oracle's single-pass answer."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu


def _run(plan, table, ncols, **kw):
    out = native.execute_to_table([native.HostInput.from_table(table)], ncols, plan.encode(), **kw)
    return pa.Table.from_batches(out) if out else None


def _final_plan(partial_plan: S.Operator, state_schema: pa.Schema) -> S.Operator:
    """HashAgg(Final) over a Scan of the partial output (group columns then state columns)."""
    def to_type(t):
        if pa.types.is_decimal(t):
            return S.decimal(t.precision, t.scale)
        return {pa.int64(): S.T_INT64, pa.bool_(): S.T_BOOL, pa.utf8(): S.T_STRING, pa.float64(): S.T_DOUBLE, pa.int32(): S.T_INT32}[t]
    fields = [to_type(f.type) for f in state_schema]
    ng = len(partial_plan.exprs)
    return S.hash_agg(S.scan(fields), [S.col(i, fields[i]) for i in range(ng)], partial_plan.aggs, S.FINAL)


def _shards(table, k):
    n = table.num_rows
    cuts = [n * i // k for i in range(k + 1)]
    return [table.slice(cuts[i], cuts[i + 1] - cuts[i]) for i in range(k)]


def test_q6_partial_then_final(built):
    from oracle import oracle as O
    table = tpch.lineitem_q6(400_000, seed=31)
    partial = tpch.q6_plan()
    states = pa.concat_tables([_run(partial, sh, 2) for sh in _shards(table, 3)])
    fplan = _final_plan(partial, states.schema)
    got = _run(fplan, states, 1)
    want_states = O.run_plan_to_arrow(S, fplan, states)
    assert got.column(0).to_pylist() == want_states.column(0).to_pylist()
    assert got.schema.field(0).type == pa.decimal128(35, 4)
    # and equals the single-pass answer over the whole table
    one = O.run_plan_to_arrow(S, partial, table)
    assert got.column(0).to_pylist() == one.column(0).to_pylist()


def test_q6_final_of_empty_partials_is_null(built):
    partial = tpch.q6_plan()
    states = pa.concat_tables([_run(partial, tpch.lineitem_q6(0), 2) for _ in range(2)])
    got = _run(_final_plan(partial, states.schema), states, 1)
    assert got.column(0).to_pylist() == [None]      # SumDecimal over no rows is NULL (sum_decimal.rs:264-279)


def test_q1_partial_then_final(built):
    from oracle import oracle as O
    table = tpch.lineitem_q1(300_000, seed=32)
    partial = tpch.q1_plan()
    states = pa.concat_tables([_run(partial, sh, tpch.Q1_NUM_OUTPUT_COLS) for sh in _shards(table, 4)])
    fplan = _final_plan(partial, states.schema)
    got = _run(fplan, states, 10)
    want = O.run_plan_to_arrow(S, fplan, states)
    rows = lambda t: sorted(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]), key=lambda r: (r[0], r[1]))
    assert rows(got) == rows(want)
    # golden schema of TPC-H Q1 (spark/src/test/resources/tpch-query-results/q1.sql.out:3-4)
    types = [f.type for f in got.schema]
    assert types[2:] == [pa.decimal128(22, 2), pa.decimal128(22, 2), pa.decimal128(36, 4), pa.decimal128(38, 6),
                         pa.decimal128(16, 6), pa.decimal128(16, 6), pa.decimal128(16, 6), pa.int64()]
    # cross-check avg = HALF_UP(sum / count) with exact Python ints
    from decimal import Decimal
    for r in rows(got):
        sum_qty, cnt, avg_qty = r[2], r[9], r[6]
        from oracle import pyint
        assert int(avg_qty.scaleb(6)) == pyint.avg_decimal(int(sum_qty.scaleb(2)), cnt, 16, 6, 2)


def test_final_with_overflowed_partial_is_null(built):
    # a partial state (sum NULL, is_empty false) means "overflowed": sticky through the merge (sum_decimal.rs:322-327)
    from decimal import Decimal
    D = pa.decimal128(10, 2)
    states = pa.table({"s": pa.array([Decimal("1.00"), None, Decimal("2.00")], D), "e": pa.array([False, False, False])})
    fplan = S.hash_agg(S.scan([S.decimal(10, 2), S.T_BOOL]), [], [S.sum_(S.col(0, S.decimal(10, 2)), S.decimal(10, 2))], S.FINAL)
    assert _run(fplan, states, 1).column(0).to_pylist() == [None]
    states2 = pa.table({"s": pa.array([Decimal("1.00"), None, Decimal("2.00")], D), "e": pa.array([False, True, False])})
    assert _run(fplan, states2, 1).column(0).to_pylist() == [Decimal("3.00")]
    # sum beyond the precision → NULL
    states3 = pa.table({"s": pa.array([Decimal("99999999.99"), Decimal("0.01")], D), "e": pa.array([False, False])})
    assert _run(fplan, states3, 1).column(0).to_pylist() == [None]


def _merge_plan(partial_plan, state_schema, mode):
    f = S.final_of(partial_plan, state_schema)
    return S.hash_agg(f.children[0], f.exprs, f.aggs, mode)


@pytest.mark.parametrize("grouped", [True, False])
def test_partial_merge_then_final(built, grouped):
    """Partial → PartialMerge (merge_batch + state) → Final equals Partial → Final and the single-pass answer; the PartialMerge
    output itself is compared with the oracle's restatement of each accumulator's state() after merge_batch."""
    from oracle import oracle as O
    from decimal import Decimal
    rng = np.random.default_rng(44)
    n = 120_000
    D = S.decimal(12, 2)
    table = pa.table({"k": pa.array(rng.integers(0, 7, n), pa.int32()),
                      "v": tpch._dec128_array(rng.integers(-10**9, 10**9, n), 12, 2),
                      "f": pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.2),
                      "i": pa.array(rng.integers(-10**6, 10**6, n), pa.int64(), mask=rng.random(n) < 0.3)})
    fields = [S.T_INT32, D, S.T_DOUBLE, S.T_INT64]
    ck, cv, cf, ci = (S.col(i, t) for i, t in enumerate(fields))
    aggs = [S.sum_(cv, S.decimal(22, 2)), S.avg(cv, S.decimal(16, 6), S.decimal(22, 2)), S.avg(cf, S.T_DOUBLE, S.T_DOUBLE), S.sum_(ci, S.T_INT64),
            S.count(ci), S.min_(cv, D), S.max_(cf, S.T_DOUBLE)]
    partial = S.hash_agg(S.scan(fields), [ck] if grouped else [], aggs)
    ncols_state = (1 if grouped else 0) + 2 + 2 + 2 + 1 + 1 + 1 + 1
    shards = _shards(table, 6)
    states = [_run(partial, sh, ncols_state, batch_size=0) for sh in shards]
    pm = _merge_plan(partial, states[0].schema, S.PARTIAL_MERGE)
    merged = [_run(pm, pa.concat_tables(states[:3]), ncols_state, batch_size=0), _run(pm, pa.concat_tables(states[3:]), ncols_state, batch_size=0)]
    rows = lambda t: sorted(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]), key=lambda r: tuple((x is None, str(x)) for x in r[:1]))
    want_merged = O.run_plan_to_arrow(S, pm, pa.concat_tables(states[:3]))
    if grouped:
        got_m, want_m = rows(merged[0]), rows(want_merged)
        assert len(got_m) == len(want_m) == 7
        for g, w in zip(got_m, want_m):
            assert g[:3] == w[:3] and g[3:5] == w[3:5] and g[6:] == w[6:]          # decimals, ints, counts, min/max: exact
            assert g[5] == pytest.approx(w[5], rel=1e-12)                          # float sums: order-dependent
    assert [f.type for f in merged[0].schema] == [f.type for f in states[0].schema]
    fplan = _merge_plan(partial, states[0].schema, S.FINAL)
    nfinal = (1 if grouped else 0) + 7
    via_merge = _run(fplan, pa.concat_tables(merged), nfinal, batch_size=0)
    direct = _run(fplan, pa.concat_tables(states), nfinal, batch_size=0)
    one = O.run_plan_to_arrow(S, fplan, O.run_plan_to_arrow(S, partial, table))
    fa = 3 if grouped else 2        # position of avg(f) in the final output
    for a, b in zip(rows(via_merge), rows(direct)):
        assert a[:fa] == b[:fa] and a[fa + 1:] == b[fa + 1:]
        assert a[fa] == pytest.approx(b[fa], rel=1e-12)
    for a, b in zip(rows(via_merge), rows(one)):
        assert a[:fa] == b[:fa] and a[fa + 1:-1] == b[fa + 1:-1]
        assert a[fa] == pytest.approx(b[fa], rel=1e-9) and a[-1] == pytest.approx(b[-1])
