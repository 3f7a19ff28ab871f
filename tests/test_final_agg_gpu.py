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


@pytest.mark.parametrize("grouped", [False, True])
def test_count_distinct_rewrite_with_mixed_mode_aggregate(built, grouped):
    """Spark plans `count(DISTINCT a), sum(b), avg(f)` as four aggregates (TPC-DS Q95's tail): Partial by (g, a) → PartialMerge by (g, a) →
    a Partial-mode aggregate by g whose sum/avg are PartialMerge (states read from initial_input_buffer_offset) while count(a) starts
    counting (HashAggregate.expr_modes, planner.rs:1274-1345, merge_as_partial.rs) → Final.  Every stage on the GPU against the oracle, and
    the end result against a direct computation."""
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    n = 120_000
    g = rng.integers(0, 9, n)
    a = rng.integers(0, 3000, n)
    b = rng.integers(-10**9, 10**9, n)
    f = rng.standard_normal(n)
    am = rng.random(n) < 0.03
    t = pa.table({"g": pa.array(g, pa.int32()), "a": pa.array(a, pa.int64(), mask=am), "b": tpch._dec128_array(b, 12, 2),
                  "f": pa.array(f, mask=rng.random(n) < 0.1)})
    D, SD = S.decimal(12, 2), S.decimal(22, 2)
    fields = [S.T_INT32, S.T_INT64, D, S.T_DOUBLE]
    keys1 = ([S.col(0, S.T_INT32)] if grouped else []) + [S.col(1, S.T_INT64)]
    aggs = [S.sum_(S.col(2, D), SD), S.avg(S.col(3, S.T_DOUBLE), S.T_DOUBLE, S.T_DOUBLE)]
    s1 = S.hash_agg(S.scan(fields), keys1, aggs, S.PARTIAL)
    nk = len(keys1)

    def stage(plan, tables, ncols):
        return pa.concat_tables([_run(plan, tb, ncols) for tb in tables])

    def canon(tb, nkeys):
        rows = list(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]))
        return sorted(rows, key=lambda r: tuple((x is None, x if x is not None else 0) for x in r[:nkeys]) if nkeys else 0)

    # stage 1 on three shards (as three map tasks), stage 2 merges them per (g, a)
    st1 = stage(s1, _shards(t, 3), nk + 4)
    want1 = pa.concat_tables([O.run_plan_to_arrow(S, s1, sh) for sh in _shards(t, 3)])
    close = lambda x, y: abs(x - y) <= 1e-9 * max(1.0, abs(y))       # float sums: summation order is unspecified
    for r, w in zip(canon(st1, nk)[:500], canon(want1, nk)[:500]):
        assert r[:nk + 2] == w[:nk + 2] and r[nk + 3] == w[nk + 3] and close(r[nk + 2], w[nk + 2])
    f1 = [S.from_arrow_type(x.type) for x in st1.schema]
    s2 = S.hash_agg(S.scan(f1), [S.col(i, f1[i]) for i in range(nk)], aggs, S.PARTIAL_MERGE)
    st2 = _run(s2, st1, st1.num_columns)
    want2 = O.run_plan_to_arrow(S, s2, st1)
    c2, w2 = canon(st2, nk), canon(want2, nk)
    assert len(c2) == len(w2)
    for r, w in zip(c2, w2):
        assert r[:nk + 2] == w[:nk + 2] and r[nk + 3] == w[nk + 3] and abs(r[nk + 2] - w[nk + 2]) <= 1e-9 * max(1.0, abs(w[nk + 2]))
    # stage 3: mixed modes — sum / avg merge their states, count(a) is a fresh Partial count over the (now distinct) a values
    f2 = [S.from_arrow_type(x.type) for x in st2.schema]
    keys3 = [S.col(0, S.T_INT32)] if grouped else []
    s3 = S.hash_agg(S.scan(f2), keys3, aggs + [S.count(S.col(nk - 1, S.T_INT64))], S.PARTIAL,
                    expr_modes=[S.PARTIAL_MERGE, S.PARTIAL_MERGE, S.PARTIAL], initial_input_buffer_offset=nk)
    want3 = O.run_plan_to_arrow(S, s3, st2)
    st3 = _run(s3, st2, want3.num_columns)
    nk3 = len(keys3)
    c3, w3 = canon(st3, nk3), canon(want3, nk3)
    assert len(c3) == len(w3)
    for r, w in zip(c3, w3):
        assert r[:nk3 + 2] == w[:nk3 + 2] and r[nk3 + 3:] == w[nk3 + 3:] and abs(r[nk3 + 2] - w[nk3 + 2]) <= 1e-9 * max(1.0, abs(w[nk3 + 2]))
    # stage 4: Final
    f3 = [S.from_arrow_type(x.type) for x in st3.schema]
    s4 = S.hash_agg(S.scan(f3), [S.col(i, f3[i]) for i in range(nk3)], aggs + [S.count(S.col(0, S.T_INT64))], S.FINAL)
    res = _run(s4, st3, nk3 + 3)
    # direct computation
    import decimal
    groups = sorted(set(g.tolist())) if grouped else [None]
    got_rows = {(r[0] if grouped else None): r[nk3:] for r in canon(res, nk3)}
    for gv in groups:
        sel = (g == gv) if grouped else np.ones(n, bool)
        want_sum = decimal.Decimal(int(b[sel].sum())).scaleb(-2)
        fv = np.asarray(t.column(3).is_valid()) & sel
        want_avg = float(f[fv].sum() / fv.sum())
        want_cnt = len(set(a[sel & ~am].tolist()))
        s_, a_, c_ = got_rows[gv]
        assert s_ == want_sum and c_ == want_cnt and abs(a_ - want_avg) <= 1e-9 * max(1.0, abs(want_avg)), (gv, s_, want_sum, c_, want_cnt)


def test_ansi_decimal_sums_raise_where_legacy_ones_turn_null(built):
    """sum / avg of decimals under ANSI: an overflow fails the query with DecimalSumOverflow (sum_decimal.rs:211-215, 352-358, 427-431, 594-600;
    avg_decimal.rs:366-380, 610-616; error.rs:75-76, 374-377) where the LEGACY aggregate gives NULL — Partial, Final and grouped"""
    import json
    from decimal import Decimal
    from oracle import oracle as O
    D10 = S.decimal(10, 2)
    big = pa.table({"k": pa.array([1, 1, 2], pa.int32()), "v": pa.array([Decimal("99999999.99"), Decimal("0.01"), Decimal("5.00")], pa.decimal128(10, 2))})
    k, v = S.col(0, S.T_INT32), S.col(1, D10)
    for grouping in ([], [k]):
        for agg, fn in ((lambda m: S.sum_(v, D10, m), "sum"), (lambda m: S.avg(v, S.decimal(14, 6), D10, m), "avg")):
            legacy = S.hash_agg(S.scan([S.T_INT32, D10]), grouping, [agg(S.LEGACY)], S.PARTIAL)
            got = _run(legacy, big, len(grouping) + 2)
            assert got.column(len(grouping)).null_count == 1, (fn, grouping)          # the overflowed state: sum NULL
            ansi = S.hash_agg(S.scan([S.T_INT32, D10]), grouping, [agg(S.ANSI)], S.PARTIAL)
            if fn == "avg":      # an average notes the overflow in its Partial state and raises when the states are merged / evaluated
                assert _run(ansi, big, len(grouping) + 2).column(len(grouping)).null_count == 1
                continue
            with pytest.raises(native.CometQueryExecutionException) as ei:
                _run(ansi, big, len(grouping) + 2)
            assert json.loads(str(ei.value)) == {"errorType": "DecimalSumOverflow", "errorClass": "ARITHMETIC_OVERFLOW", "params": {"functionName": fn}}
            with pytest.raises(O.OracleError, match="ARITHMETIC_OVERFLOW"):
                O.run_plan_to_arrow(S, ansi, big)
    # Final: two partial sums that do not fit together
    states = pa.table({"s": pa.array([Decimal("99999999.99"), Decimal("0.01")], pa.decimal128(10, 2)), "e": pa.array([False, False])})
    fin = lambda m: S.hash_agg(S.scan([D10, S.T_BOOL]), [], [S.sum_(S.col(0, D10), D10, m)], S.FINAL)
    assert _run(fin(S.LEGACY), states, 1).column(0).to_pylist() == [None]
    with pytest.raises(native.CometQueryExecutionException, match='DecimalSumOverflow.*"functionName":"sum"'):
        _run(fin(S.ANSI), states, 1)
    # a GROUPED average whose partial state overflowed (sum NULL under a count): NULL, or the error under ANSI (avg_decimal.rs:542-636); the
    # ungrouped accumulator skips NULL partial sums instead (:331-356) and gives 1.00 / 3 in both modes
    avg_states = pa.table({"k": pa.array([7, 7], pa.int32()), "s": pa.array([None, Decimal("1.00")], pa.decimal128(10, 2)), "c": pa.array([2, 1], pa.int64())})
    favg = lambda m, g: (S.hash_agg(S.scan([S.T_INT32, D10, S.T_INT64]), [S.col(0, S.T_INT32)], [S.avg(S.col(1, D10), S.decimal(14, 6), D10, m)], S.FINAL) if g else
                         S.hash_agg(S.scan([D10, S.T_INT64]), [], [S.avg(S.col(0, D10), S.decimal(14, 6), D10, m)], S.FINAL))
    assert _run(favg(S.LEGACY, True), avg_states, 2).column(1).to_pylist() == [None] == O.run_plan_to_arrow(S, favg(S.LEGACY, True), avg_states).column(1).to_pylist()
    with pytest.raises(native.CometQueryExecutionException, match='DecimalSumOverflow.*"functionName":"avg"'):
        _run(favg(S.ANSI, True), avg_states, 2)
    with pytest.raises(O.OracleError, match="ARITHMETIC_OVERFLOW avg"):
        O.run_plan_to_arrow(S, favg(S.ANSI, True), avg_states)
    flat = avg_states.select(["s", "c"])
    for m in (S.LEGACY, S.ANSI):
        assert _run(favg(m, False), flat, 1).column(0).to_pylist() == [Decimal("0.333333")] == O.run_plan_to_arrow(S, favg(m, False), flat).column(0).to_pylist()
    # … and ANSI sums that fit are the LEGACY sums
    ok = pa.table({"k": pa.array([1, 1, 2], pa.int32()), "v": pa.array([Decimal("1.50"), Decimal("2.25"), None], pa.decimal128(10, 2))})
    a = _run(S.hash_agg(S.scan([S.T_INT32, D10]), [k], [S.sum_(v, D10, S.ANSI)], S.PARTIAL), ok, 3)
    b = _run(S.hash_agg(S.scan([S.T_INT32, D10]), [k], [S.sum_(v, D10, S.LEGACY)], S.PARTIAL), ok, 3)
    assert sorted(zip(*[c.to_pylist() for c in a.columns]), key=str) == sorted(zip(*[c.to_pylist() for c in b.columns]), key=str)


def _metrics_run(plan, dev_or_host_inputs, ncols):
    it = native.CometExecIterator(dev_or_host_inputs, ncols, plan.encode(), batch_size=0)
    batches = []
    while True:
        b = native.Native.executePlan(it.handle, ncols)
        if b is None:
            break
        batches.append(b)
    m = S.decode_metric_node(it.metrics())[0]
    it.close()
    return (pa.Table.from_batches(batches) if batches else None), m


_MERGE_CASE = []


def _merge_case():
    """states of 40 K groups (a fifth of them in two or three rows), the Final plan and the oracle's answers — computed once (the oracle's Final is a Python loop per group)"""
    if not _MERGE_CASE:
        from oracle import oracle as O
        rng = np.random.default_rng(91)
        n = 48_000
        k0 = rng.integers(0, 40_000, n).astype(np.int64) * 7919
        k1 = (k0 % 5).astype(np.int32)
        table = pa.table({"k0": pa.array(k0, mask=rng.random(n) < 0.001), "k1": pa.array(k1),
                          "m": tpch._dec128_array(rng.integers(-10**9, 10**9, n), 12, 2), "q": pa.array(rng.integers(-1000, 1000, n), pa.int64(), mask=rng.random(n) < 0.05)})
        D = S.decimal(12, 2)
        partial = S.hash_agg(S.scan([S.T_INT64, S.T_INT32, D, S.T_INT64]), [S.col(0, S.T_INT64), S.col(1, S.T_INT32)],
                             [S.sum_(S.col(2, D), S.decimal(22, 2)), S.count(S.col(3, S.T_INT64)), S.min_(S.col(3, S.T_INT64), S.T_INT64), S.max_(S.col(3, S.T_INT64), S.T_INT64),
                              S.avg(S.col(2, D), S.decimal(16, 6), S.decimal(22, 2))], S.PARTIAL)
        states = pa.concat_tables([O.run_plan_to_arrow(S, partial, sh) for sh in _shards(table, 3)]).combine_chunks()
        fplan = _final_plan(partial, states.schema)
        top = S.sort(fplan, [(S.col(2, S.decimal(22, 2)), True, True), (S.col(0, S.T_INT64), False, False), (S.col(1, S.T_INT32), False, False)], fetch=10)
        want = O.run_plan_to_arrow(S, fplan, states)
        srt = want.rename_columns([f"c{i}" for i in range(want.num_columns)]).combine_chunks().sort_by([("c2", "descending"), ("c0", "ascending"), ("c1", "ascending")])
        _MERGE_CASE.append((states, fplan, top, want, srt.slice(0, 10)))
    return _MERGE_CASE[0]


@pytest.mark.parametrize("shape", ["final_over_device_table", "top10_above_final", "host_batches"])
def test_merging_aggregate_of_many_groups_runs_partitioned(built, shape):
    """A Final aggregate whose input is about one state row per group (the Final stage of a high-cardinality GROUP BY: SF100 Q3 has 1.13 M): when the whole input is ONE
    device-resident chunk it is partitioned by key hash, merged per partition in LDS and emitted from there (comet_device.hpp template C''; metric agg_partitioned_merges) —
    no table in HBM.  40 K groups, a fifth of them in two or three state rows (shards of one Partial), NULL group keys, sum / count / min / max / avg states;
    bit-exact against the oracle's Final, under a Sort + fetch too (the aggregate is then a nested one); host batches keep the table path."""
    states, fplan, top, want, want10 = _merge_case()
    assert states.num_rows > 36_000
    nout = want.num_columns
    srt = lambda t: t.rename_columns([f"c{i}" for i in range(t.num_columns)]).combine_chunks().sort_by([("c0", "ascending"), ("c1", "ascending")])
    if shape == "final_over_device_table":
        dev = native.DeviceTable.from_arrow(states, "cuda:0")
        got, m = _metrics_run(fplan, [native.DeviceInput(dev)], nout)
        assert m["agg_partitioned_merges"] == 1, m
        assert got.num_rows == want.num_rows and srt(got).equals(srt(want))
    elif shape == "top10_above_final":
        dev = native.DeviceTable.from_arrow(states, "cuda:0")
        got, m = _metrics_run(top, [native.DeviceInput(dev)], nout)
        assert m["agg_partitioned_merges"] == 1, m
        assert [got.column(i).to_pylist() for i in range(nout)] == [want10.column(i).to_pylist() for i in range(nout)]
    else:
        got, m = _metrics_run(fplan, [native.HostInput.from_table(states)], nout)
        assert m["agg_partitioned_merges"] == 0, m
        assert got.num_rows == want.num_rows and srt(got).equals(srt(want))


def test_merging_aggregate_with_wide_states_runs_partitioned(built):
    """Eight decimal sums, a count and a min per group: a slot of the partition's LDS table is wider than 192 bytes, so the table holds 128 slots — fewer than the
    workgroup has threads (comet_device.hpp AggPart::kCap; the emit sweep's tail threads hold no slot).  ROCm 7.2's compiler refused the kernel as first written
    (a zero-length array for that shape; ROCm 7.0's accepted it and would have emitted nothing).  40 K groups, a fifth of them in two or three state rows."""
    from oracle import oracle as O
    rng = np.random.default_rng(97)
    n = 48_000
    k0 = rng.integers(0, 40_000, n).astype(np.int64) * 6151
    cols = {"k0": pa.array(k0, mask=rng.random(n) < 0.001), "q": pa.array(rng.integers(-1000, 1000, n), pa.int64(), mask=rng.random(n) < 0.05)}
    for i in range(8):
        cols[f"m{i}"] = tpch._dec128_array(rng.integers(-10**9, 10**9, n), 12, 2)
    table = pa.table(cols)
    D = S.decimal(12, 2)
    partial = S.hash_agg(S.scan([S.T_INT64, S.T_INT64] + [D] * 8), [S.col(0, S.T_INT64)],
                         [S.sum_(S.col(2 + i, D), S.decimal(22, 2)) for i in range(8)] + [S.count(S.col(1, S.T_INT64)), S.min_(S.col(1, S.T_INT64), S.T_INT64)], S.PARTIAL)
    states = pa.concat_tables([O.run_plan_to_arrow(S, partial, sh) for sh in _shards(table, 3)]).combine_chunks()
    fplan = _final_plan(partial, states.schema)
    want = O.run_plan_to_arrow(S, fplan, states)
    srt = lambda t: t.rename_columns([f"c{i}" for i in range(t.num_columns)]).combine_chunks().sort_by([("c0", "ascending")])
    assert states.num_rows > 36_000      # (the partitioned path starts at 32 768 rows)
    dev = native.DeviceTable.from_arrow(states, "cuda:0")
    got, m = _metrics_run(fplan, [native.DeviceInput(dev)], want.num_columns)
    assert m["agg_partitioned_merges"] == 1, m
    assert got.num_rows == want.num_rows and srt(got).equals(srt(want))


@pytest.mark.parametrize("groups", [60_000, 4])
def test_partial_aggregate_over_a_joins_output_partitions_when_its_groups_are_many(built, groups):
    """A grouped aggregate whose whole input is ONE modest chunk — here a join's output — may take the partition → LDS merge → emit path; whether the groups are many is
    read off the partition sizes after the counting pass: 60 K groups of ≈ 3 rows run partitioned (metric 1), 4 groups of 50 K rows fall back to the table path
    (metric 0).  Partial states bit-equal to the oracle's either way."""
    from oracle import oracle as O
    rng = np.random.default_rng(93)
    n = 200_000
    k = rng.integers(0, groups, n).astype(np.int64) * 104_729
    table = pa.table({"k": pa.array(k, mask=rng.random(n) < 0.002), "m": tpch._dec128_array(rng.integers(-10**9, 10**9, n), 12, 2),
                      "q": pa.array(rng.integers(-1000, 1000, n), pa.int64(), mask=rng.random(n) < 0.05), "j": pa.array(np.zeros(n, np.int32))})
    dim = pa.table({"j": pa.array(np.zeros(1, np.int32)), "w": pa.array(np.ones(1, np.int64))})
    D = S.decimal(12, 2)
    join = S.hash_join(S.scan([S.T_INT64, D, S.T_INT64, S.T_INT32]), S.scan([S.T_INT32, S.T_INT64]), [S.col(3, S.T_INT32)], [S.col(0, S.T_INT32)], S.INNER, S.BUILD_RIGHT)
    partial = S.hash_agg(join, [S.col(0, S.T_INT64)],
                         [S.sum_(S.col(1, D), S.decimal(22, 2)), S.count(S.col(2, S.T_INT64)), S.min_(S.col(2, S.T_INT64), S.T_INT64), S.avg(S.col(1, D), S.decimal(16, 6), S.decimal(22, 2)),
                          S.sum_(S.col(5, S.T_INT64), S.T_INT64)], S.PARTIAL)
    want = O.run_plan_to_arrow(S, partial, [table, dim])
    got, m = _metrics_run(partial, [native.HostInput.from_table(table), native.HostInput.from_table(dim)], want.num_columns)
    assert m["agg_partitioned_merges"] == (1 if groups > 100 else 0), m
    srt = lambda t: t.rename_columns([f"c{i}" for i in range(t.num_columns)]).combine_chunks().sort_by([("c0", "ascending")])
    assert got.num_rows == want.num_rows and srt(got).equals(srt(want))
