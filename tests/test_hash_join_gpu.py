"""GPU parity for the hash join: Inner / LeftSemi / LeftAnti, NULL keys never match, duplicates on
both sides, residual condition, and TPC-H Q3's two joins + grouped aggregate in one native plan.
Join output order is unspecified in the reference → multiset comparison; all values bit-exact."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu


def _oracle(plan, tables):
    from oracle import oracle as O
    return O.run_plan_to_arrow(S, plan, tables)


def _run(plan, tables, ncols, **kw):
    out = native.execute_to_table([native.HostInput.from_table(t) for t in tables], ncols, plan.encode(), **kw)
    return pa.Table.from_batches(out) if out else None


def _rows(t):
    if t is None:
        return []
    return sorted(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]), key=lambda r: tuple((x is None, str(x)) for x in r))


def _tables(nl, nr, seed, nulls=True):
    rng = np.random.default_rng(seed)
    lk = pa.array(rng.integers(0, 50, nl), pa.int64(), mask=(rng.random(nl) < 0.1) if nulls else None)
    lv = pa.array(rng.integers(-1000, 1000, nl), pa.int32())
    rk = pa.array(rng.integers(0, 60, nr), pa.int64(), mask=(rng.random(nr) < 0.1) if nulls else None)
    rv = pa.array(rng.random(nr))
    return pa.table({"k": lk, "v": lv}), pa.table({"k": rk, "w": rv})


@pytest.mark.parametrize("build", [S.BUILD_LEFT, S.BUILD_RIGHT])
def test_inner_join_with_duplicates_and_null_keys(built, build):
    left, right = _tables(3000, 2000, 1)
    plan = S.hash_join(S.scan([S.T_INT64, S.T_INT32]), S.scan([S.T_INT64, S.T_DOUBLE]), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)],
                       S.INNER, build)
    got, want = _run(plan, [left, right], 4, batch_size=0), _oracle(plan, [left, right])
    assert got.num_rows == want.num_rows and got.num_rows > 10_000
    assert _rows(got) == _rows(want)


@pytest.mark.parametrize("build", [S.BUILD_LEFT, S.BUILD_RIGHT])
@pytest.mark.parametrize("jt", [S.LEFT_SEMI, S.LEFT_ANTI])
def test_semi_and_anti_join(built, jt, build):
    left, right = _tables(5000, 700, 2)
    plan = S.hash_join(S.scan([S.T_INT64, S.T_INT32]), S.scan([S.T_INT64, S.T_DOUBLE]), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)],
                       jt, build)
    got, want = _run(plan, [left, right], 2, batch_size=0), _oracle(plan, [left, right])
    assert _rows(got) == _rows(want)
    assert got.num_rows > 0


def test_inner_join_with_residual_condition_and_chains(built):
    left, right = _tables(4000, 4000, 3, nulls=False)
    l = S.filter_(S.scan([S.T_INT64, S.T_INT32]), S.gt(S.col(1, S.T_INT32), S.lit(-500, S.T_INT32)))
    r = S.project(S.scan([S.T_INT64, S.T_DOUBLE]), [S.col(0, S.T_INT64), S.math("multiply", S.col(1, S.T_DOUBLE), S.lit(2.0, S.T_DOUBLE), S.T_DOUBLE)])
    # condition over left ++ right: l.v (col 1) > 0 OR r.w2 (col 3) < 1.0
    cond = S.or_(S.gt(S.col(1, S.T_INT32), S.lit(0, S.T_INT32)), S.lt(S.col(3, S.T_DOUBLE), S.lit(1.0, S.T_DOUBLE)))
    j = S.hash_join(l, r, [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.INNER, S.BUILD_RIGHT, cond)
    plan = S.project(j, [S.col(0, S.T_INT64), S.col(1, S.T_INT32), S.col(3, S.T_DOUBLE)])
    got, want = _run(plan, [left, right], 3, batch_size=0), _oracle(plan, [left, right])
    assert got.num_rows == want.num_rows > 0
    assert _rows(got) == _rows(want)


def test_join_with_empty_side(built):
    left, right = _tables(100, 0, 4)
    plan = S.hash_join(S.scan([S.T_INT64, S.T_INT32]), S.scan([S.T_INT64, S.T_DOUBLE]), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)],
                       S.INNER, S.BUILD_RIGHT)
    assert _run(plan, [left, right], 4) is None
    anti = S.hash_join(S.scan([S.T_INT64, S.T_INT32]), S.scan([S.T_INT64, S.T_DOUBLE]), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)],
                       S.LEFT_ANTI, S.BUILD_RIGHT)
    assert _run(anti, [left, right], 2, batch_size=0).num_rows == 100


def test_tpch_q3_two_joins_and_grouped_aggregate(built):
    customer, orders, lineitem = tpch.q3_tables(20_000, seed=3)
    plan = tpch.q3_plan()
    got = _run(plan, [customer, orders, lineitem], tpch.Q3_NUM_OUTPUT_COLS, batch_size=0)
    want = _oracle(plan, [customer, orders, lineitem])
    assert got.num_rows == want.num_rows > 100
    assert _rows(got) == _rows(want)
    assert got.schema.field(3).type == pa.decimal128(36, 4)


@pytest.mark.parametrize("jt", [S.LEFT_OUTER, S.RIGHT_OUTER, S.FULL_OUTER])
@pytest.mark.parametrize("build", [S.BUILD_LEFT, S.BUILD_RIGHT])
def test_outer_joins(built, jt, build):
    # planner.rs:2448-2460: the preserved side's unmatched rows (NULL keys included) come out NULL-extended, whichever side is built
    left, right = _tables(4000, 3000, 11)
    plan = S.hash_join(S.scan([S.T_INT64, S.T_INT32]), S.scan([S.T_INT64, S.T_DOUBLE]), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, build)
    got, want = _run(plan, [left, right], 4, batch_size=0), _oracle(plan, [left, right])
    assert got.num_rows == want.num_rows
    assert _rows(got) == _rows(want)
    nulls_right = sum(1 for r in _rows(got) if r[2] is None and r[3] is None)
    nulls_left = sum(1 for r in _rows(got) if r[0] is None and r[1] is None)
    assert (nulls_right > 0) == (jt in (S.LEFT_OUTER, S.FULL_OUTER)) and (nulls_left > 0) == (jt in (S.RIGHT_OUTER, S.FULL_OUTER))


def test_outer_join_with_condition_empty_sides_and_projection_on_top(built):
    left, right = _tables(3000, 2500, 12, nulls=False)
    cond = S.gt(S.col(1, S.T_INT32), S.lit(0, S.T_INT32))          # residual over left ++ right: left.v > 0
    j = S.hash_join(S.scan([S.T_INT64, S.T_INT32]), S.scan([S.T_INT64, S.T_DOUBLE]), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)],
                    S.FULL_OUTER, S.BUILD_RIGHT, cond)
    plan = S.project(j, [S.col(0, S.T_INT64), S.col(3, S.T_DOUBLE), S.is_null(S.col(2, S.T_INT64))])
    got, want = _run(plan, [left, right], 3, batch_size=0), _oracle(plan, [left, right])
    assert _rows(got) == _rows(want) and got.num_rows > 3000
    # an empty build side: every probe row survives a probe-preserving outer join
    lo = S.hash_join(S.scan([S.T_INT64, S.T_INT32]), S.scan([S.T_INT64, S.T_DOUBLE]), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.LEFT_OUTER, S.BUILD_RIGHT)
    got = _run(lo, [left, right.slice(0, 0)], 4, batch_size=0)
    assert got.num_rows == left.num_rows and got.column(3).null_count == left.num_rows
    # an empty probe side: every build row survives a build-preserving outer join
    ro = S.hash_join(S.scan([S.T_INT64, S.T_INT32]), S.scan([S.T_INT64, S.T_DOUBLE]), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.RIGHT_OUTER, S.BUILD_RIGHT)
    got = _run(ro, [left.slice(0, 0), right], 4, batch_size=0)
    assert got.num_rows == right.num_rows and got.column(0).null_count == right.num_rows


@pytest.mark.parametrize("jt", [S.INNER, S.LEFT_OUTER, S.RIGHT_OUTER, S.FULL_OUTER, S.LEFT_SEMI, S.LEFT_ANTI])
def test_sort_merge_join_operator(built, jt):
    """SortMergeJoin (operator.proto:765-771, planner.rs:2126-2191) runs as hash join + sort of the output by the join keys:
    same rows as the oracle, and the preserved side's key column comes out in key order."""
    left, right = _tables(5000, 4000, 21)
    plan = S.sort_merge_join(S.scan([S.T_INT64, S.T_INT32]), S.scan([S.T_INT64, S.T_DOUBLE]), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt)
    ncols = 2 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 4
    got, want = _run(plan, [left, right], ncols, batch_size=0), _oracle(plan, [left, right])
    assert _rows(got) == _rows(want) and got.num_rows > 0
    kcol = 2 if jt == S.RIGHT_OUTER else 0
    assert got.column(kcol).to_pylist() == want.column(kcol).to_pylist()       # NULLS FIRST, ascending


@pytest.mark.parametrize("jt,build", [(S.INNER, S.BUILD_RIGHT), (S.LEFT_OUTER, S.BUILD_RIGHT), (S.LEFT_SEMI, S.BUILD_RIGHT), (S.LEFT_ANTI, S.BUILD_RIGHT), (S.INNER, S.BUILD_LEFT),
                                      (S.FULL_OUTER, S.BUILD_LEFT)])
def test_broadcast_nested_loop_join(built, jt, build):
    """BroadcastNestedLoopJoin (117): joins without an equality — a range condition between the sides, and a plain cross join."""
    import numpy as np
    import pyarrow as pa
    from oracle import oracle as O
    rng = np.random.default_rng(51)
    nl, nr = 3000, 40
    left = pa.table({"x": pa.array(rng.integers(0, 1000, nl), pa.int64(), mask=rng.random(nl) < 0.05), "id": pa.array(np.arange(nl), pa.int64())})
    right = pa.table({"lo": pa.array(rng.integers(0, 900, nr), pa.int64()), "hi": pa.array(rng.integers(100, 1000, nr), pa.int64(), mask=rng.random(nr) < 0.1), "tag": pa.array(["band-%02d" % i for i in range(nr)])})
    lf, rf = [S.T_INT64, S.T_INT64], [S.T_INT64, S.T_INT64, S.T_STRING]
    cond = S.and_(S.gt_eq(S.col(0, S.T_INT64), S.col(2, S.T_INT64)), S.lt(S.col(0, S.T_INT64), S.col(3, S.T_INT64)))      # left.x in [right.lo, right.hi)
    plan = S.nested_loop_join(S.scan(lf), S.scan(rf), jt, build, cond)
    ncols = 2 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 5
    run = lambda p, nc: pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(left), native.HostInput.from_table(right)], nc, p.encode(), batch_size=0))
    got, want = run(plan, ncols), O.run_plan_to_arrow(S, plan, [left, right])
    key = lambda tb: sorted(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]), key=lambda r: tuple((v is None, str(v)) for v in r))
    assert got.num_rows == want.num_rows > 0 and key(got) == key(want)
    if jt == S.INNER and build == S.BUILD_RIGHT:
        cross = S.nested_loop_join(S.scan(lf), S.scan(rf), S.INNER, S.BUILD_RIGHT, None)
        assert run(cross, 5).num_rows == nl * nr


# ---- probe-side fusion: the probe child's Filter / Projection chain runs inside the probe kernel (codegen.hpp JoinFusion) ----
def _fusion_tables(seed):
    rng = np.random.default_rng(seed)
    nl, nr = 6000, 2500
    left = pa.table({"k": pa.array(rng.integers(0, 80, nl), pa.int64(), mask=rng.random(nl) < 0.08),
                     "v": pa.array(rng.integers(-1000, 1000, nl), pa.int32(), mask=rng.random(nl) < 0.05),
                     "s": pa.array([None if rng.random() < 0.1 else "payload-string-number-%d" % int(x) for x in rng.integers(0, 400, nl)]),
                     "d": tpch._dec128_array(rng.integers(-10**9, 10**9, nl), 12, 2)})
    right = pa.table({"k": pa.array(rng.integers(0, 90, nr), pa.int64(), mask=rng.random(nr) < 0.08), "w": pa.array(rng.random(nr))})
    return left, right


LFIELDS = [S.T_INT64, S.T_INT32, S.T_STRING, S.decimal(12, 2)]
RFIELDS = [S.T_INT64, S.T_DOUBLE]


def _probe_chain():
    """Filter → Project (reordered columns, a computed decimal, the key shifted by a computed expression) → Filter over the left scan"""
    D = S.decimal(12, 2)
    f1 = S.filter_(S.scan(LFIELDS), S.gt(S.col(1, S.T_INT32), S.lit(-600, S.T_INT32)))                       # NULL v fails too
    p = S.project(f1, [S.col(2, S.T_STRING), S.math("add", S.col(0, S.T_INT64), S.lit(3, S.T_INT64), S.T_INT64), S.col(3, D),
                       S.check_overflow(S.math("add", S.col(3, D), S.lit(__import__("decimal").Decimal("1.50"), D), S.decimal(13, 2)), S.decimal(13, 2))])
    return S.filter_(p, S.lt(S.col(2, D), S.lit(__import__("decimal").Decimal("5000000.00"), D)))            # → (s, k + 3, d, d + 1.50)


@pytest.mark.parametrize("jt", [S.INNER, S.LEFT_OUTER, S.RIGHT_OUTER, S.FULL_OUTER, S.LEFT_SEMI, S.LEFT_ANTI])
def test_fused_probe_chain_matches_oracle_and_the_unfused_join(built, jt):
    """The probe child (left, build = right) is a chain: rows its Filters drop are not part of the join at all (not NULL-extended by
    the outer joins, not kept by the anti join), its computed columns are evaluated for emitted rows only, its Utf8 payload is gathered
    from the SOURCE table.  Same multiset as the oracle and as the engine with spark.comet.gpu.join.fuseProbe=false."""
    left, right = _fusion_tables(21)
    j = S.hash_join(_probe_chain(), S.scan(RFIELDS), [S.col(1, S.T_INT64)], [S.col(0, S.T_INT64)], jt, S.BUILD_RIGHT)
    ncols = 4 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 6
    explain = native.compile_plan(j.encode())
    assert "probe side fused" in explain and "probe filter (fused into the probe kernel)" in explain
    got = _run(j, [left, right], ncols, batch_size=0)
    unfused = _run(j, [left, right], ncols, batch_size=0, config=S.config_map({"spark.comet.gpu.join.fuseProbe": "false"}))
    want = _oracle(j, [left, right])
    assert got.schema.types == want.schema.types == unfused.schema.types
    assert _rows(got) == _rows(want) == _rows(unfused)
    assert got.num_rows > 100


def test_fused_probe_on_the_right_with_a_residual_condition(built):
    left, right = _fusion_tables(22)
    # build = left (plain scan), probe = right chain; the condition reads a computed probe column and a build column
    rchain = S.project(S.filter_(S.scan(RFIELDS), S.lt(S.col(1, S.T_DOUBLE), S.lit(0.8, S.T_DOUBLE))),
                       [S.math("multiply", S.col(1, S.T_DOUBLE), S.lit(100.0, S.T_DOUBLE), S.T_DOUBLE), S.col(0, S.T_INT64)])
    cond = S.gt(S.cast(S.col(1, S.T_INT32), S.T_DOUBLE), S.col(4, S.T_DOUBLE))          # left.v > right.w * 100
    for jt in (S.INNER, S.LEFT_OUTER, S.FULL_OUTER):
        j = S.hash_join(S.scan(LFIELDS), rchain, [S.col(0, S.T_INT64)], [S.col(1, S.T_INT64)], jt, S.BUILD_LEFT, cond)
        assert "probe side fused" in native.compile_plan(j.encode())
        got, want = _run(j, [left, right], 6, batch_size=0), _oracle(j, [left, right])
        assert _rows(got) == _rows(want) and got.num_rows > 0


def test_fused_probe_whose_filter_keeps_nothing_and_device_resident_inputs(built):
    left, right = _fusion_tables(23)
    none = S.filter_(S.scan(LFIELDS), S.gt(S.col(1, S.T_INT32), S.lit(5000, S.T_INT32)))
    j = S.hash_join(none, S.scan(RFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.LEFT_OUTER, S.BUILD_RIGHT)
    assert _run(j, [left, right], 6, batch_size=0) is None
    # RightOuter over an empty probe side: every build row, NULL-extended
    j = S.hash_join(none, S.scan(RFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.RIGHT_OUTER, S.BUILD_RIGHT)
    got = _run(j, [left, right], 6, batch_size=0)
    assert got.num_rows == right.num_rows and got.column(0).null_count == right.num_rows
    # HBM-resident inputs take the same fused kernel (the scan is zero-copy: the probe kernel reads the caller's buffers)
    nn_left = pa.table({"k": pa.array(np.arange(5000, dtype=np.int64) % 97), "v": pa.array(np.arange(5000, dtype=np.int32) - 2500)})
    nn_right = pa.table({"k": pa.array(np.arange(60, dtype=np.int64)), "w": pa.array(np.arange(60, dtype=np.float64))})
    chain = S.project(S.filter_(S.scan([S.T_INT64, S.T_INT32]), S.gt(S.col(1, S.T_INT32), S.lit(0, S.T_INT32))), [S.col(0, S.T_INT64), S.col(1, S.T_INT32)])
    j = S.hash_join(chain, S.scan(RFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.INNER, S.BUILD_RIGHT)
    dl, dr = native.DeviceTable.from_arrow(nn_left, "cuda:0"), native.DeviceTable.from_arrow(nn_right, "cuda:0")
    out = native.execute_to_table([native.DeviceInput(dl), native.DeviceInput(dr)], 4, j.encode(), batch_size=0)
    assert _rows(pa.Table.from_batches(out)) == _rows(_oracle(j, [nn_left, nn_right]))


# ---- runs of equal keys on the build side (comet_device.hpp "Runs of equal keys"), semi-join reduction ----
def _clustered(n, seed, keys_per_run=4.5, null_frac=0.03):
    """a fact-table-shaped side: rows of one key are neighbours (runs of random length, crossing the 64-row wave boundaries), NULL keys
    sprinkled INSIDE runs, a second column that differs inside a run (what a residual condition looks at)"""
    rng = np.random.default_rng(seed)
    lens = rng.geometric(1.0 / keys_per_run, n)
    key = np.repeat(np.arange(len(lens), dtype=np.int64) * 3 + 7, lens)[:n]
    return pa.table({"k": pa.array(key, mask=rng.random(n) < null_frac), "w": pa.array(rng.integers(0, 4, n).astype(np.int32), mask=rng.random(n) < null_frac),
                     "id": pa.array(np.arange(n, dtype=np.int64))})


CFIELDS = [S.T_INT64, S.T_INT32, S.T_INT64]


@pytest.mark.parametrize("jt", [S.INNER, S.LEFT_OUTER, S.FULL_OUTER, S.LEFT_SEMI, S.LEFT_ANTI])
@pytest.mark.parametrize("build", [S.BUILD_LEFT, S.BUILD_RIGHT])
@pytest.mark.parametrize("cond", [False, True])
def test_clustered_keys_on_both_sides(built, jt, build, cond):
    """Self-join-shaped inputs: both sides clustered by the key, about 4.5 rows per key, a residual condition that holds for some rows of
    a run and not for others (the ws_wh self-join of TPC-DS Q95: w1 <> w2).  The build inserts one row per run; the probe walks the run."""
    left, right = _clustered(9000, 31), _clustered(8000, 32)
    c = S.not_(S.eq(S.col(1, S.T_INT32), S.col(4, S.T_INT32))) if cond else None
    j = S.hash_join(S.scan(CFIELDS), S.scan(CFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, build, c)
    ncols = 3 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 6
    got, want = _run(j, [left, right], ncols, batch_size=0), _oracle(j, [left, right])
    assert got.num_rows == want.num_rows > 1000
    assert _rows(got) == _rows(want)


def test_build_side_above_a_million_rows_is_sized_by_its_runs(built):
    """≥ 2^20 build rows: the runs are counted first (k_jbcnt) and the bucket array sized by them; 4.5 rows per key, NULL keys, a condition"""
    build_t, probe_t = _clustered(1_100_000, 41), _clustered(3000, 42)
    probe_t = probe_t.set_column(0, "k", pa.array(np.asarray(build_t.column(0).fill_null(7))[::366][:3000].astype(np.int64)))     # keys that exist
    c = S.not_(S.eq(S.col(1, S.T_INT32), S.col(4, S.T_INT32)))
    for jt in (S.INNER, S.LEFT_SEMI):
        j = S.hash_join(S.scan(CFIELDS), S.scan(CFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, S.BUILD_RIGHT, c)
        got, want = _run(j, [probe_t, build_t], 6 if jt == S.INNER else 3, batch_size=0), _oracle(j, [probe_t, build_t])
        assert got.num_rows == want.num_rows > 2000
        assert _rows(got) == _rows(want)


@pytest.mark.parametrize("side", ["left", "right"])
def test_semi_join_reduction_keeps_the_answer(built, side):
    """x IN (SELECT k FROM b JOIN c ON …): the Inner join under the semi join's build side is projected onto one of its sides, so the
    engine runs it as a LeftSemi join (children swapped when the projection keeps the right side) — same rows as the plan as written
    (spark.comet.gpu.join.semiReduction=false) and as the oracle; input streams stay bound to their Scan leaves."""
    a, b, c_ = _clustered(5000, 51), _clustered(6000, 52), _clustered(4000, 53)
    cond = S.not_(S.eq(S.col(1, S.T_INT32), S.col(4, S.T_INT32)))
    inner = S.hash_join(S.scan(CFIELDS), S.scan(CFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.INNER, S.BUILD_RIGHT, cond)
    proj = S.project(inner, [S.col(0, S.T_INT64)] if side == "left" else [S.math("add", S.col(3, S.T_INT64), S.lit(0, S.T_INT64), S.T_INT64)])
    plan = S.hash_join(S.scan(CFIELDS), proj, [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.LEFT_SEMI, S.BUILD_RIGHT)
    explain = native.compile_plan(plan.encode())
    assert explain.count("LeftSemi") == 2 and "Inner" not in explain
    got = _run(plan, [a, b, c_], 3, batch_size=0)
    asis = _run(plan, [a, b, c_], 3, batch_size=0, config=S.config_map({"spark.comet.gpu.join.semiReduction": "false"}))
    want = _oracle(plan, [a, b, c_])
    assert _rows(got) == _rows(want) == _rows(asis) and 100 < got.num_rows < 5000
    # the inner join's output multiplicity matters to an aggregate above it: no reduction there
    agg = S.hash_agg(proj, [S.col(0, S.T_INT64)], [S.count(S.col(0, S.T_INT64))])
    assert "Inner" in native.compile_plan(agg.encode())
    got, want = _run(agg, [b, c_], 2, batch_size=0), _oracle(agg, [b, c_])
    assert _rows(got) == _rows(want)


def _sorted(t):
    """the table's rows in one canonical order (large outputs: sorted by Arrow, not as Python tuples)"""
    t = t.rename_columns([f"c{i}" for i in range(t.num_columns)]).combine_chunks()
    return t.sort_by([(n, "ascending") for n in t.column_names])


def _join_metrics(plan, tables, ncols, config=b""):
    it = native.CometExecIterator([native.HostInput.from_table(t) for t in tables], ncols, plan.encode(), config=config, batch_size=0)
    batches = []
    while True:
        b = native.Native.executePlan(it.handle, ncols)
        if b is None:
            break
        batches.append(b)
    m = S.decode_metric_node(it.metrics())[0]
    it.close()
    return (pa.Table.from_batches(batches) if batches else None), m


@pytest.mark.parametrize("jt", [S.INNER, S.LEFT_SEMI, S.LEFT_ANTI, S.LEFT_OUTER, S.RIGHT_OUTER, S.FULL_OUTER])
def test_unique_integer_build_key_goes_through_the_direct_map(built, jt):
    """A build side of ≥ 2^20 rows whose one integer key is unique and dense enough (a primary key) is probed through the direct map — key bitmap,
    keys below each 128-bit block, build rows in key order — instead of a hash table (comet_device.hpp JoinDirectTable): every join type, NULL
    keys on both sides, probe keys below / above / inside the holes of the build side's range, with and without a residual condition."""
    rng = np.random.default_rng(61)
    nb, npr = 1_200_000, 400_000
    keys = rng.permutation(np.arange(1000, 1000 + 3 * nb, dtype=np.int64))[:nb]                    # unique, every third key of the range, shuffled
    build_t = pa.table({"k": pa.array(keys, mask=rng.random(nb) < 0.01), "v": pa.array(rng.integers(-1000, 1000, nb), pa.int32()), "id": pa.array(np.arange(nb, dtype=np.int64))})
    pk = rng.integers(0, 1000 + 3 * nb + 2000, npr).astype(np.int64)
    pk[:4] = [-5, 0, 999, 1000 + 3 * nb + 1999]
    probe_t = pa.table({"k": pa.array(pk, mask=rng.random(npr) < 0.02), "v": pa.array(rng.integers(-1000, 1000, npr), pa.int32()), "id": pa.array(np.arange(npr, dtype=np.int64))})
    for cond in (None, S.lt(S.col(1, S.T_INT32), S.col(4, S.T_INT32))):
        j = S.hash_join(S.scan(CFIELDS), S.scan(CFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, S.BUILD_RIGHT, cond)
        ncols = 3 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 6
        got, m = _join_metrics(j, [probe_t, build_t], ncols)
        want = _oracle(j, [probe_t, build_t])
        if cond is None and jt in (S.LEFT_SEMI, S.LEFT_ANTI):      # (only the key's existence matters: the bitmap alone answers, no rows[] needed)
            assert m["join_bitmap_only"] == 1 and m["join_direct_maps"] == 0, m
        else:
            assert m["join_direct_maps"] == 1, m
        assert got.num_rows == want.num_rows > 1000
        assert _sorted(got).equals(_sorted(want))


def test_a_key_that_comes_twice_falls_back_to_the_hash_table(built):
    rng = np.random.default_rng(62)
    nb = 1_100_000
    keys = rng.permutation(np.arange(0, 2 * nb, dtype=np.int64))[:nb]
    keys[nb // 2] = keys[7]                                                                          # far apart: no run, found by the bitmap's build pass
    build_t = pa.table({"k": pa.array(keys), "v": pa.array(rng.integers(-1000, 1000, nb), pa.int32()), "id": pa.array(np.arange(nb, dtype=np.int64))})
    pk = np.concatenate([rng.integers(0, 2 * nb, 50_000), np.full(5, keys[7])]).astype(np.int64)
    probe_t = pa.table({"k": pa.array(pk), "v": pa.array(rng.integers(-1000, 1000, len(pk)), pa.int32()), "id": pa.array(np.arange(len(pk), dtype=np.int64))})
    j = S.hash_join(S.scan(CFIELDS), S.scan(CFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.INNER, S.BUILD_RIGHT)
    got, m = _join_metrics(j, [probe_t, build_t], 6)
    want = _oracle(j, [probe_t, build_t])
    assert m["join_direct_maps"] == 0
    assert got.num_rows == want.num_rows and _sorted(got).equals(_sorted(want))


# ---- the bucket table (comet_device.hpp template D'': partitioned build into LDS tables, one random access per probe key) and the bitmap-only join ----
ALL_TYPES = [S.INNER, S.LEFT_SEMI, S.LEFT_ANTI, S.LEFT_OUTER, S.RIGHT_OUTER, S.FULL_OUTER]
KFIELDS = [S.T_INT64, S.T_INT32, S.T_INT32, S.T_INT64]      # k, k2, v, id


def _sparse_dup_sides(nb, npr, seed, key_space=None):
    """build side: sparse 64-bit keys (no range a bitmap could cover), about a third of them twice or more — SCATTERED duplicates (separate runs) and CLUSTERED ones
    (neighbouring rows: one entry with a run length), a second key column with few values, NULLs in both; probe side: half its keys from the build side"""
    rng = np.random.default_rng(seed)
    key_space = key_space or nb
    base = rng.integers(-(1 << 62), 1 << 62, key_space).astype(np.int64)
    bk = base[rng.integers(0, key_space, nb)]
    runs = rng.random(nb) < 0.25                                     # a quarter of the rows repeat their predecessor's keys: runs, also across wave boundaries
    idx = np.arange(nb)
    idx[runs] = 0
    idx = np.maximum.accumulate(idx)
    bk = bk[idx]
    bk2 = rng.integers(0, 3, nb).astype(np.int32)[idx]
    build_t = pa.table({"k": pa.array(bk, mask=rng.random(nb) < 0.02), "k2": pa.array(bk2, mask=rng.random(nb) < 0.02),
                        "v": pa.array(rng.integers(-1000, 1000, nb), pa.int32()), "id": pa.array(np.arange(nb, dtype=np.int64))})
    pk = np.where(rng.random(npr) < 0.5, base[rng.integers(0, key_space, npr)], rng.integers(-(1 << 62), 1 << 62, npr)).astype(np.int64)
    probe_t = pa.table({"k": pa.array(pk, mask=rng.random(npr) < 0.02), "k2": pa.array(rng.integers(0, 3, npr).astype(np.int32), mask=rng.random(npr) < 0.02),
                        "v": pa.array(rng.integers(-1000, 1000, npr), pa.int32()), "id": pa.array(np.arange(npr, dtype=np.int64))})
    return probe_t, build_t


@pytest.mark.parametrize("jt", ALL_TYPES)
@pytest.mark.parametrize("two_cols", [False, True])
def test_bucket_table_join_on_sparse_duplicate_keys(built, jt, two_cols):
    """Every join type over the bucket table: sparse keys with scattered and clustered duplicates, one key column (the entry's signature IS the key) and
    two (the leader is verified by P::match), NULL keys on both sides, with and without a residual condition, built on either side."""
    probe_t, build_t = _sparse_dup_sides(150_000, 120_000, 71)
    lk = [S.col(0, S.T_INT64)] + ([S.col(1, S.T_INT32)] if two_cols else [])
    ncols = 4 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 8
    for cond in (None, S.lt(S.col(2, S.T_INT32), S.col(6, S.T_INT32))):
        for build in (S.BUILD_RIGHT, S.BUILD_LEFT):
            tables = [probe_t, build_t] if build == S.BUILD_RIGHT else [build_t, probe_t]
            j = S.hash_join(S.scan(KFIELDS), S.scan(KFIELDS), lk, lk, jt, build, cond)
            got, m = _join_metrics(j, tables, ncols)
            want = _oracle(j, tables)
            assert m["join_bucket_tables"] == 1 and m["join_direct_maps"] == 0, m
            assert got.num_rows == want.num_rows > 1000, (jt, two_cols, build)
            assert _sorted(got).equals(_sorted(want)), (jt, two_cols, build, cond is not None)


@pytest.mark.parametrize("jt", ALL_TYPES)
def test_bucket_table_at_ten_million_build_rows(built, jt):
    """≥ 10 M build rows, two key columns, sparse keys, duplicates both scattered and in runs (about 2,400 partitions of the table): all six join types
    against the oracle, bit-exact as multisets."""
    probe_t, build_t = _sparse_dup_sides(10_500_000, 1_500_000, 72, key_space=6_000_000)
    lk = [S.col(0, S.T_INT64), S.col(1, S.T_INT32)]
    ncols = 4 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 8
    cond = S.lt(S.col(2, S.T_INT32), S.col(6, S.T_INT32)) if jt in (S.INNER, S.LEFT_SEMI, S.FULL_OUTER) else None
    j = S.hash_join(S.scan(KFIELDS), S.scan(KFIELDS), lk, lk, jt, S.BUILD_RIGHT, cond)
    got, m = _join_metrics(j, [probe_t, build_t], ncols)
    want = _oracle(j, [probe_t, build_t])
    assert m["join_bucket_tables"] == 1, m
    assert got.num_rows == want.num_rows > 100_000
    assert _sorted(got).equals(_sorted(want))


@pytest.mark.parametrize("jt", [S.LEFT_SEMI, S.LEFT_ANTI])
def test_semi_and_anti_join_answered_by_the_key_bitmap_alone(built, jt):
    """A semi / anti join without a residual condition on one integer key of a foreign key's shape: the build side is its key bitmap — no table
    (metric join_bitmap_only).  Duplicate and NULL build keys, probe keys below / above / in the holes of the range, a probe chain fused in."""
    rng = np.random.default_rng(73)
    nb, npr = 300_000, 500_000
    bk = (rng.integers(0, nb, nb) * 3 + 1000).astype(np.int64)
    build_t = pa.table({"k": pa.array(bk, mask=rng.random(nb) < 0.01), "v": pa.array(rng.integers(-1000, 1000, nb), pa.int32()), "id": pa.array(np.arange(nb, dtype=np.int64))})
    pk = rng.integers(0, 1000 + 3 * nb + 2000, npr).astype(np.int64)
    pk[:4] = [-5, 0, 999, 1000 + 3 * nb + 1999]
    probe_t = pa.table({"k": pa.array(pk, mask=rng.random(npr) < 0.02), "v": pa.array(rng.integers(-1000, 1000, npr), pa.int32()), "id": pa.array(np.arange(npr, dtype=np.int64))})
    chain = S.project(S.filter_(S.scan(CFIELDS), S.gt(S.col(1, S.T_INT32), S.lit(-900, S.T_INT32))), [S.col(0, S.T_INT64), S.col(1, S.T_INT32), S.col(2, S.T_INT64)])
    for left in (S.scan(CFIELDS), chain):
        j = S.hash_join(left, S.scan(CFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, S.BUILD_RIGHT)
        got, m = _join_metrics(j, [probe_t, build_t], 3)
        want = _oracle(j, [probe_t, build_t])
        assert m["join_bitmap_only"] == 1 and m["join_bucket_tables"] == 0, m
        assert got.num_rows == want.num_rows > 10_000
        assert _sorted(got).equals(_sorted(want))
    # with a residual condition the rows of the build side matter: the bucket table
    j = S.hash_join(S.scan(CFIELDS), S.scan(CFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, S.BUILD_RIGHT, S.lt(S.col(1, S.T_INT32), S.col(4, S.T_INT32)))
    got, m = _join_metrics(j, [probe_t, build_t], 3)
    want = _oracle(j, [probe_t, build_t])
    assert m["join_bitmap_only"] == 0 and m["join_bucket_tables"] == 1, m
    assert _sorted(got).equals(_sorted(want))


def test_bucket_table_partition_overflow_takes_the_chained_table(built):
    """Three keys, each in tens of thousands of SEPARATE runs: every entry of a key lands in one partition, which cannot hold them — the build says so
    and the join runs over the chained table (metric join_bucket_tables stays 0); the answer is the oracle's."""
    rng = np.random.default_rng(74)
    nb = 150_000
    build_t = pa.table({"k": pa.array(rng.integers(0, 3, nb).astype(np.int64) * (1 << 40)), "k2": pa.array(np.zeros(nb, np.int32)),
                        "v": pa.array(rng.integers(-1000, 1000, nb), pa.int32()), "id": pa.array(np.arange(nb, dtype=np.int64))})
    probe_t = pa.table({"k": pa.array(np.array([0, 1 << 40, 5, 2 << 40, 1 << 40], dtype=np.int64)), "k2": pa.array(np.zeros(5, np.int32)),
                        "v": pa.array(np.arange(5, dtype=np.int32)), "id": pa.array(np.arange(5, dtype=np.int64))})
    j = S.hash_join(S.scan(KFIELDS), S.scan(KFIELDS), [S.col(0, S.T_INT64), S.col(1, S.T_INT32)], [S.col(0, S.T_INT64), S.col(1, S.T_INT32)], S.INNER, S.BUILD_RIGHT)
    got, m = _join_metrics(j, [probe_t, build_t], 8)
    want = _oracle(j, [probe_t, build_t])
    assert m["join_bucket_tables"] == 0, m
    assert got.num_rows == want.num_rows > 100_000
    assert _sorted(got).equals(_sorted(want))


def test_monotone_hash_is_dropped_for_keys_that_do_not_spread_over_their_range(built):
    """One integer key → the bucket table tries the order-preserving hash (slots in key order: clustered, sorted fact tables then stream).  Keys 0 .. 200 000 plus
    ONE key at 2^40: under that hash all but one entry would land in the first partition — the partition sizes say so before the table is built, and the join runs
    with the scrambling hash (metrics: a bucket table, not a monotone one).  Dense keys with NULLs and duplicates keep the monotone hash."""
    rng = np.random.default_rng(75)
    nb, npr = 200_000, 150_000
    bk = rng.permutation(np.arange(nb, dtype=np.int64))
    bk[5] = 1 << 40
    bk[1000:1100] = bk[900:1000]                                   # duplicates, far apart: no direct map
    build_t = pa.table({"k": pa.array(bk, mask=rng.random(nb) < 0.01), "v": pa.array(rng.integers(-1000, 1000, nb), pa.int32()), "id": pa.array(np.arange(nb, dtype=np.int64))})
    pk = rng.integers(-10, nb + 10, npr).astype(np.int64)
    pk[:3] = [1 << 40, (1 << 40) + 1, -(1 << 50)]
    probe_t = pa.table({"k": pa.array(pk, mask=rng.random(npr) < 0.02), "v": pa.array(rng.integers(-1000, 1000, npr), pa.int32()), "id": pa.array(np.arange(npr, dtype=np.int64))})
    cond = S.lt(S.col(1, S.T_INT32), S.col(4, S.T_INT32))
    for jt in (S.INNER, S.LEFT_ANTI, S.FULL_OUTER):
        j = S.hash_join(S.scan(CFIELDS), S.scan(CFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, S.BUILD_RIGHT, cond)
        got, m = _join_metrics(j, [probe_t, build_t], 3 if jt == S.LEFT_ANTI else 6)
        want = _oracle(j, [probe_t, build_t])
        assert m["join_bucket_tables"] == 1 and m["join_mono_tables"] == 0, m
        assert got.num_rows == want.num_rows > 1000 and _sorted(got).equals(_sorted(want))
    # the same sides without the outlier: dense keys → the monotone hash; probe keys below, above and inside the range
    bk2 = bk.copy()
    bk2[5] = 17
    build2 = build_t.set_column(0, "k", pa.array(bk2, mask=rng.random(nb) < 0.01))
    for jt in (S.INNER, S.LEFT_SEMI, S.RIGHT_OUTER):
        j = S.hash_join(S.scan(CFIELDS), S.scan(CFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, S.BUILD_RIGHT, cond)
        got, m = _join_metrics(j, [probe_t, build2], 3 if jt == S.LEFT_SEMI else 6)
        want = _oracle(j, [probe_t, build2])
        assert m["join_bucket_tables"] == 1 and m["join_mono_tables"] == 1, m
        assert got.num_rows == want.num_rows > 1000 and _sorted(got).equals(_sorted(want))


# ---- build-side chain fusion (round 6): the build passes read the chain's Scan table; rows its Filters drop are no build rows ----
@pytest.mark.parametrize("jt", ALL_TYPES)
@pytest.mark.parametrize("size", ["lds", "bucket"])
def test_fused_build_chain_matches_oracle_and_the_unfused_join(built, jt, size):
    """The BUILD child (right) is Filter → Project (reordered columns, a computed key, a computed payload) → Filter over a Scan: the table, the bitmap and an outer
    join's tail of unmatched build rows see only the rows the chain keeps (P::bkeep), output columns are the chain's expressions over the source row.  Small build
    side (the per-block LDS table) and large (bucket table); probe side fused too or a bare scan; residual condition over a computed build column.  Same multiset as
    the oracle and as the engine with spark.comet.gpu.join.fuseBuild=false."""
    rng = np.random.default_rng(81 if size == "lds" else 82)
    nb, npr = (4000, 6000) if size == "lds" else (180_000, 90_000)
    keys = max(50, nb // 3)
    build_t = pa.table({"w": pa.array(rng.integers(-1000, 1000, nb), pa.int32(), mask=rng.random(nb) < 0.05), "k": pa.array(rng.integers(0, keys, nb), pa.int64(), mask=rng.random(nb) < 0.03),
                        "id": pa.array(np.arange(nb, dtype=np.int64))})
    probe_t = pa.table({"k": pa.array(rng.integers(0, keys + keys // 4, npr), pa.int64(), mask=rng.random(npr) < 0.03), "v": pa.array(rng.integers(-1000, 1000, npr), pa.int32()),
                        "id": pa.array(np.arange(npr, dtype=np.int64))})
    BF = [S.T_INT32, S.T_INT64, S.T_INT64]
    f1 = S.filter_(S.scan(BF), S.and_(S.gt(S.col(0, S.T_INT32), S.lit(-700, S.T_INT32)), S.is_not_null(S.col(1, S.T_INT64))))       # NULL w fails too
    pr = S.project(f1, [S.math("add", S.col(1, S.T_INT64), S.lit(0, S.T_INT64), S.T_INT64), S.col(2, S.T_INT64), S.math("multiply", S.col(0, S.T_INT32), S.lit(2, S.T_INT32), S.T_INT32)])
    bchain = S.filter_(pr, S.lt(S.col(2, S.T_INT32), S.lit(1600, S.T_INT32)))                                                          # → (k + 0, id, 2 w)
    pchain = S.project(S.filter_(S.scan(CFIELDS), S.gt(S.col(1, S.T_INT32), S.lit(-900, S.T_INT32))), [S.col(0, S.T_INT64), S.col(1, S.T_INT32), S.col(2, S.T_INT64)])
    ncols = 3 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 6
    for left in (S.scan(CFIELDS), pchain):
        for cond in (None, S.lt(S.col(1, S.T_INT32), S.col(5, S.T_INT32))):      # probe.v < 2 · build.w
            j = S.hash_join(left, bchain, [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, S.BUILD_RIGHT, cond)
            always = S.config_map({"spark.comet.gpu.join.fuseBuild": "always"})      # (by default only chains whose Filters merely drop NULLs are fused)
            assert "build side fused with its chain" not in native.compile_plan(j.encode())
            got, m = _join_metrics(j, [probe_t, build_t], ncols, config=always)
            assert m["join_fused_builds"] == 1, m
            unfused = _run(j, [probe_t, build_t], ncols, batch_size=0, config=S.config_map({"spark.comet.gpu.join.fuseBuild": "false"}))
            want = _oracle(j, [probe_t, build_t])
            assert got.num_rows == want.num_rows == unfused.num_rows > 100, (jt, size, cond is not None)
            assert _sorted(got).equals(_sorted(want)) and _sorted(unfused).equals(_sorted(want)), (jt, size, cond is not None)


def test_fused_build_on_the_left_and_a_filter_that_keeps_nothing(built):
    rng = np.random.default_rng(83)
    nb, npr = 3000, 2000
    left = pa.table({"k": pa.array(rng.integers(0, 500, nb), pa.int64()), "v": pa.array(rng.integers(-1000, 1000, nb), pa.int32()), "id": pa.array(np.arange(nb, dtype=np.int64))})
    right = pa.table({"k": pa.array(rng.integers(0, 600, npr), pa.int64()), "v": pa.array(rng.integers(-1000, 1000, npr), pa.int32()), "id": pa.array(np.arange(npr, dtype=np.int64))})
    lchain = S.filter_(S.scan(CFIELDS), S.gt(S.col(1, S.T_INT32), S.lit(0, S.T_INT32)))
    for jt in (S.INNER, S.LEFT_SEMI, S.LEFT_ANTI, S.LEFT_OUTER, S.FULL_OUTER):      # LeftSemi / LeftAnti built on the left: the output is a subset of the BUILD rows the chain keeps
        j = S.hash_join(lchain, S.scan(CFIELDS), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, S.BUILD_LEFT)
        always = S.config_map({"spark.comet.gpu.join.fuseBuild": "always"})
        (got, m), want = _join_metrics(j, [left, right], 3 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 6, config=always), _oracle(j, [left, right])
        assert m["join_fused_builds"] == 1, m
        assert got.num_rows == want.num_rows > 0 and _sorted(got).equals(_sorted(want)), jt
    none = S.filter_(S.scan(CFIELDS), S.gt(S.col(1, S.T_INT32), S.lit(5000, S.T_INT32)))
    always = S.config_map({"spark.comet.gpu.join.fuseBuild": "always"})
    j = S.hash_join(S.scan(CFIELDS), none, [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.LEFT_OUTER, S.BUILD_RIGHT)
    got = _run(j, [left, right], 6, batch_size=0, config=always)
    assert got.num_rows == left.num_rows and got.column(3).null_count == left.num_rows
    j = S.hash_join(S.scan(CFIELDS), none, [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.RIGHT_OUTER, S.BUILD_RIGHT)
    assert _run(j, [left, right], 6, batch_size=0, config=always) is None
    # isnotnull-only chains are fused by default
    nn = S.project(S.filter_(S.scan(CFIELDS), S.is_not_null(S.col(0, S.T_INT64))), [S.col(0, S.T_INT64), S.col(2, S.T_INT64)])
    j = S.hash_join(S.scan(CFIELDS), nn, [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.INNER, S.BUILD_RIGHT)
    assert "build side fused with its chain" in native.compile_plan(j.encode())
    got, want = _run(j, [left, right], 5, batch_size=0), _oracle(j, [left, right])
    assert got.num_rows == want.num_rows > 0 and _sorted(got).equals(_sorted(want))
