"""Expand (operator 107 — grouping sets / rollup / cube; planner.rs:1913-1948, operators/expand.rs): every input row yields one row per
projection.  On the GPU each projection is a fused Projection writing into one set of output buffers at its row offset; NULL markers of
Utf8 columns are not generated code.  Checked stand-alone and in its natural habitat: GROUP BY ROLLUP(flag, status) below a Partial aggregate."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu


def _table(n, seed=4):
    rng = np.random.default_rng(seed)
    return pa.table({"flag": pa.array([None if rng.random() < 0.04 else ["A", "N", "R"][int(i)] for i in rng.integers(0, 3, n)]),
                     "store": pa.array(["Store number %03d of the chain" % int(i) for i in rng.integers(0, 40, n)]),
                     "k": pa.array(rng.integers(0, 6, n).astype(np.int32), mask=rng.random(n) < 0.05),
                     "v": tpch._dec128_array(rng.integers(-10**7, 10**7, n), 12, 2),
                     "f": pa.array(rng.random(n), mask=rng.random(n) < 0.1)})


D = S.decimal(12, 2)
FIELDS = [S.T_STRING, S.T_STRING, S.T_INT32, D, S.T_DOUBLE]


def _rows(tb):
    return sorted(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]), key=lambda r: tuple((x is None, str(x)) for x in r))


def _projections():
    flag, store, k, v, f = (S.col(i, t) for i, t in enumerate(FIELDS))
    NS, NI = S.lit(None, S.T_STRING), S.lit(None, S.T_INT32)
    gid = lambda x: S.lit(x, S.T_INT32)
    # rollup(flag, store, k): (flag, store, k), (flag, store), (flag), ()
    return [[v, f, flag, store, k, gid(0)], [v, f, flag, store, NI, gid(1)], [v, f, flag, NS, NI, gid(3)], [v, f, NS, NS, NI, gid(7)]]


def test_expand_alone(built):
    from oracle import oracle as O
    t = _table(30_000)
    plan = S.expand(S.filter_(S.scan(FIELDS), S.is_not_null(S.col(4, S.T_DOUBLE))), _projections())
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 6, plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, [t])
    assert got.num_rows == want.num_rows == 4 * sum(1 for x in t.column(4).to_pylist() if x is not None)
    assert got.schema.types == want.schema.types
    assert _rows(got) == _rows(want)


def test_rollup_aggregate_over_expand(built):
    """GROUP BY ROLLUP(flag, store, k): Expand below a Partial aggregate keyed by (flag, store, k, grouping id) — long Utf8 keys included —
    then Final; the total row (grouping id 7) must equal a direct sum."""
    from oracle import oracle as O
    t = _table(50_000, seed=9)
    ex = S.expand(S.scan(FIELDS), _projections())
    SD = S.decimal(22, 2)
    keys = [S.col(2, S.T_STRING), S.col(3, S.T_STRING), S.col(4, S.T_INT32), S.col(5, S.T_INT32)]
    partial = S.hash_agg(ex, keys, [S.sum_(S.col(0, D), SD), S.count(S.col(1, S.T_DOUBLE))], S.PARTIAL)
    run = lambda plan, tb, nc: pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(tb)], nc, plan.encode(), batch_size=0))
    st, want_st = run(partial, t, 7), O.run_plan_to_arrow(S, partial, [t])
    assert _rows(st) == _rows(want_st)
    final = S.final_of(partial, st.schema)
    got, want = run(final, st, 6), O.run_plan_to_arrow(S, final, [st])
    assert _rows(got) == _rows(want)
    total = [r for r in _rows(got) if r[3] == 7]
    assert len(total) == 1 and total[0][5] == sum(1 for x in t.column(4).to_pylist() if x is not None)
    import decimal
    assert total[0][4] == sum(t.column(3).to_pylist(), decimal.Decimal(0))
