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


def test_case_when_projection_and_conditional_sum(built):
    # CASE WHEN (expr.proto:473-483, planner.rs:677-704): first TRUE branch wins, NULL conditions fall through, no ELSE → NULL;
    # and the TPC-H Q14 shape sum(CASE WHEN c THEN x ELSE 0 END) as a grouped aggregate input
    from datafusion_comet_amd import tpch
    n = 60_000
    rng = np.random.default_rng(21)
    k = pa.array(rng.integers(0, 10, n), pa.int32(), mask=rng.random(n) < 0.1)
    v = tpch._dec128_array(rng.integers(-10**6, 10**6, n), 12, 2)
    f = pa.array(rng.standard_normal(n), pa.float64(), mask=rng.random(n) < 0.1)
    table = pa.table({"k": k, "v": v, "f": f})
    D = S.decimal(12, 2)
    ck, cv, cf = S.col(0, S.T_INT32), S.col(1, D), S.col(2, S.T_DOUBLE)
    i32 = lambda x: S.lit(x, S.T_INT32)
    e1 = S.case_when([(S.lt(ck, i32(3)), cv), (S.lt(ck, i32(6)), S.lit(12345, D))], S.lit(0, D))        # with ELSE
    e2 = S.case_when([(S.gt(cf, S.lit(0.5, S.T_DOUBLE)), cf), (S.is_null(ck), S.lit(-1.0, S.T_DOUBLE))])   # no ELSE → NULL
    e3 = S.case_when([(S.eq(ck, i32(1)), i32(10)), (S.eq(ck, i32(1)), i32(20)), (S.gt_eq(ck, i32(8)), ck)], S.lit(None, S.T_INT32))
    plan = S.project(S.scan([S.T_INT32, D, S.T_DOUBLE]), [e1, e2, e3])
    got = pa.Table.from_batches(_run(plan, table, 3))
    want = _oracle(plan, table)
    for i in range(3):
        assert got.column(i).combine_chunks().equals(want.column(i).combine_chunks()), f"column {i}"
    assert got.column(1).null_count > 0 and got.column(2).null_count > 0
    agg = S.hash_agg(S.project(S.scan([S.T_INT32, D, S.T_DOUBLE]), [ck, e1]), [S.col(0, S.T_INT32)], [S.sum_(S.col(1, D), S.decimal(22, 2))])
    rows = lambda t: sorted(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]), key=lambda r: (r[0] is None, r[0] or 0))
    assert rows(pa.Table.from_batches(_run(agg, table, 3, batch_size=0))) == rows(_oracle(agg, table))


def test_decimal_division(built):
    # decimal_div (spark-expr/src/math_funcs/div.rs:71-165): result scale via (s2 + s3 + 1) widening, HALF_UP at the last
    # digit, zero divisor → NULL through the If the Scala serde wraps around it (nullIfWhenPrimitive), ANSI → DIVIDE_BY_ZERO
    from datafusion_comet_amd import tpch
    n = 30_000
    rng = np.random.default_rng(33)
    D = S.decimal(12, 2)
    lv = rng.integers(-10**11, 10**11, n)
    rv = rng.integers(-10**6, 10**6, n)
    rv[::17] = 0
    lv[:4] = [15, -15, 10**12 - 1, -(10**12 - 1)]
    rv[:4] = [200, 200, 1, 3]
    table = pa.table({"l": tpch._dec128_array(lv, 12, 2), "r": tpch._dec128_array(rv, 12, 2),
                      "w": pa.array([__import__("decimal").Decimal(int(x)).scaleb(-6) for x in rng.integers(-10**17, 10**17, n)], pa.decimal128(38, 6))})
    W = S.decimal(38, 6)
    cl, cr, cw = S.col(0, D), S.col(1, D), S.col(2, W)
    nz = lambda c, t: S.if_(S.eq(c, S.lit(0, t)), S.lit(None, t), c)            # what Spark's serde emits for a non-ANSI divisor
    R1 = S.decimal(27, 15)      # decimal(12,2) / decimal(12,2)
    R2 = S.decimal(38, 6)       # decimal(38,6) / decimal(12,2), precision-capped by Spark
    e1 = S.check_overflow(S.math("divide", cl, nz(cr, D), R1), R1)
    e2 = S.check_overflow(S.math("divide", cw, nz(cr, D), R2), R2)
    plan = S.project(S.scan([D, D, W]), [e1, e2])
    got = pa.Table.from_batches(_run(plan, table, 2))
    want = _oracle(plan, table)
    for i in range(2):
        assert got.column(i).combine_chunks().equals(want.column(i).combine_chunks()), f"column {i}"
    from decimal import Decimal
    assert got.column(0)[0].as_py() == Decimal("0.075000000000000") and got.column(0)[1].as_py() == Decimal("-0.075000000000000")
    assert got.column(0).null_count >= n // 17
    # exact-integer cross oracle for the HALF_UP rule
    for i in (2, 3, 100, 101):
        l, r = int(lv[i]), int(rv[i])
        if r:
            num, den = abs(l) * 10**15, abs(r)
            q = (2 * num + den) // (2 * den) * (-1 if (l < 0) != (r < 0) else 1)
            assert int(got.column(0)[i].as_py().scaleb(15)) == q
    # ANSI: a zero divisor raises
    ansi = S.project(S.scan([D, D, W]), [S.math("divide", cl, cr, R1, S.ANSI)])
    with pytest.raises(native.CometQueryExecutionException, match="DIVIDE_BY_ZERO"):
        _run(ansi, table, 1)


def test_scalar_functions_exact_subset(built):
    """ScalarFunc (expr.proto:466-471): ceil / floor (Float → Int64 with Rust's saturating `as i64`, Decimal → div_ceil/div_floor),
    abs (wrapping / ANSI error), sqrt, signum, isnan (NULL → false), datepart(year|month|day|quarter|dow|doy) as Comet's serde
    emits it for year()/month()/… — all exactly defined, so bit-identical to the oracle."""
    from datafusion_comet_amd import tpch
    n = 50_000
    rng = np.random.default_rng(77)
    f = rng.standard_normal(n) * 1e6
    f[:8] = [np.nan, np.inf, -np.inf, 9.3e18, -9.3e18, -0.0, 0.5, -0.5]
    i32 = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    i32[0] = -2**31
    t = pa.table({"f": pa.array(f, mask=rng.random(n) < 0.1),
                  "d": pa.Array.from_buffers(pa.decimal128(12, 2), n, [pa.py_buffer(np.packbits(rng.random(n) >= 0.1, bitorder="little").tobytes()),
                                                                      tpch._dec128_array(rng.integers(-10**9, 10**9, n), 12, 2).buffers()[1]]),
                  "i": pa.array(i32, mask=rng.random(n) < 0.1),
                  "dt": pa.array(rng.integers(-30000, 60000, n), pa.int32(), mask=rng.random(n) < 0.1).cast(pa.date32())})
    D = S.decimal(12, 2)
    F, Dc, I, DT = S.col(0, S.T_DOUBLE), S.col(1, D), S.col(2, S.T_INT32), S.col(3, S.T_DATE)
    sf = S.scalar_func
    outs = [sf("ceil", [F], S.T_INT64), sf("floor", [F], S.T_INT64), sf("ceil", [Dc], S.decimal(11, 0)), sf("floor", [Dc], S.decimal(11, 0)),
            sf("abs", [F], S.T_DOUBLE), sf("abs", [Dc], D), sf("abs", [I, S.lit(False, S.T_BOOL)], S.T_INT32),
            sf("sqrt", [sf("abs", [F], S.T_DOUBLE)], S.T_DOUBLE), sf("signum", [F], S.T_DOUBLE), sf("isnan", [F], S.T_BOOL)]
    outs += [S.date_part(p, DT) for p in ("year", "month", "day", "quarter", "dow", "doy")]
    plan = S.project(S.scan([S.T_DOUBLE, D, S.T_INT32, S.T_DATE]), outs)
    got = pa.Table.from_batches(_run(plan, table=t, ncols=len(outs), batch_size=0))
    want = _oracle(plan, t)
    for i in range(len(outs)):
        g, w = got.column(i).combine_chunks(), want.column(i).combine_chunks()
        assert g.type == w.type, i
        if pa.types.is_floating(g.type):
            assert g.is_valid().equals(w.is_valid()) and g.fill_null(0).to_numpy().tobytes() == w.fill_null(0).to_numpy().tobytes(), f"column {i}"
        else:
            assert g.equals(w), f"column {i}"
    assert got.column(9).null_count == 0 and got.column(10)[5].as_py() is not None
    # year() in a predicate + group key (TPC-H Q7/Q8/Q9 shape), ANSI abs overflow, unknown function
    agg = S.hash_agg(S.filter_(S.scan([S.T_DOUBLE, D, S.T_INT32, S.T_DATE]), S.gt_eq(S.date_part("year", DT), S.lit(1995, S.T_INT32))),
                     [S.date_part("year", DT)], [S.count(S.col(3, S.T_DATE)), S.sum_(Dc, S.decimal(22, 2))])
    rows = lambda tb: sorted(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]), key=lambda r: (r[0] is None, r[0] or 0))
    assert rows(pa.Table.from_batches(_run(agg, table=t, ncols=4, batch_size=0))) == rows(_oracle(agg, t))
    ansi = S.project(S.scan([S.T_DOUBLE, D, S.T_INT32, S.T_DATE]), [sf("abs", [I, S.lit(True, S.T_BOOL)], S.T_INT32)])
    with pytest.raises(native.CometQueryExecutionException, match='ARITHMETIC_OVERFLOW.*"fromType":"Int32"'):      # abs.rs:238
        _run(ansi, table=t, ncols=1)
    with pytest.raises(native.CometNativeException, match="levenshtein"):
        native.compile_plan(S.project(S.scan([S.T_DOUBLE]), [sf("levenshtein", [S.col(0, S.T_DOUBLE)], S.T_INT32)]).encode())


def test_remainder_int_and_float(built):
    # create_modulo_expr (math_funcs/modulo_expr.rs): zero divisor → NULL (ANSI: error), sign of the dividend, MIN % -1 = 0, fmod for floats
    n = 40_000
    rng = np.random.default_rng(88)
    a = rng.integers(-2**31, 2**31 - 1, n).astype(np.int32)
    b = rng.integers(-50, 50, n).astype(np.int32)
    a[:3] = [-2**31, 7, -7]
    b[:3] = [-1, 0, 3]
    l = rng.integers(-2**62, 2**62, n)
    f = rng.standard_normal(n) * 1000
    g = rng.standard_normal(n)
    g[5] = 0.0
    g[6] = -0.0
    t = pa.table({"a": pa.array(a, mask=rng.random(n) < 0.1), "b": pa.array(b), "l": pa.array(l), "f": pa.array(f), "g": pa.array(g, mask=rng.random(n) < 0.1)})
    fields = [S.T_INT32, S.T_INT32, S.T_INT64, S.T_DOUBLE, S.T_DOUBLE]
    A, B, L, F, G = (S.col(i, ty) for i, ty in enumerate(fields))
    outs = [S.math("remainder", A, B, S.T_INT32), S.math("remainder", L, S.lit(-7, S.T_INT64), S.T_INT64), S.math("remainder", L, S.cast(B, S.T_INT64), S.T_INT64),
            S.math("remainder", F, G, S.T_DOUBLE), S.math("remainder", F, S.lit(2.5, S.T_DOUBLE), S.T_DOUBLE)]
    plan = S.project(S.scan(fields), outs)
    got = pa.Table.from_batches(_run(plan, table=t, ncols=5, batch_size=0))
    want = _oracle(plan, t)
    for i in range(5):
        g_, w_ = got.column(i).combine_chunks(), want.column(i).combine_chunks()
        assert g_.is_valid().equals(w_.is_valid()), i
        assert g_.fill_null(0).to_numpy().tobytes() == w_.fill_null(0).to_numpy().tobytes(), f"column {i}"
    assert got.column(0)[0].as_py() == 0 and got.column(0)[1].as_py() is None and got.column(0)[2].as_py() == -1
    ansi = S.project(S.scan(fields), [S.math("remainder", A, B, S.T_INT32, S.ANSI)])
    with pytest.raises(native.CometQueryExecutionException, match="REMAINDER_BY_ZERO"):
        _run(ansi, table=t, ncols=1)


def test_remainder_decimal(built):
    """Decimal % Decimal (create_modulo_expr, math_funcs/modulo_expr.rs:137-206): operands at the larger scale — through 256 bits where the reference
    casts to Decimal256 —, sign of the dividend, zero divisor → NULL (ANSI: error); first the reference's own test_modulo_basic_decimal vector."""
    import decimal
    decimal.getcontext().prec = 80
    D184 = S.decimal(18, 4)
    from datafusion_comet_amd import tpch
    ref = pa.table({"a": tpch._dec128_array(np.array([3000000000000000000, 2000000000000000000]), 18, 4), "b": tpch._dec128_array(np.array([1000000000000000000, 5000000000000000000]), 18, 4)})
    got = pa.Table.from_batches(_run(S.project(S.scan([D184, D184]), [S.math("remainder", S.col(0, D184), S.col(1, D184), D184)]), table=ref, ncols=1, batch_size=0))
    assert np.frombuffer(got.column(0).combine_chunks().buffers()[1], np.int64)[::2].tolist() == [0, 2000000000000000000]
    n = 30_000
    rng = np.random.default_rng(5)
    py = __import__("random").Random(5)
    def dec(p, s, zeros=False):
        vals = [py.randrange(-10**py.randrange(1, p + 1), 10**py.randrange(1, p + 1)) for _ in range(n)]
        if zeros:
            for k in range(0, n, 97):
                vals[k] = 0
        return pa.array([decimal.Decimal(v).scaleb(-s) for v in vals], pa.decimal128(p, s), mask=rng.random(n) < 0.05)
    cases = [((12, 2), (12, 2)), ((18, 4), (10, 0)), ((10, 0), (18, 6)), ((38, 10), (38, 10)), ((38, 0), (38, 38)), ((38, 38), (38, 0)), ((38, 6), (20, 18)), ((20, 2), (38, 30))]
    cols, fields, outs = {}, [], []
    for k, ((p1, s1), (p2, s2)) in enumerate(cases):
        cols[f"a{k}"], cols[f"b{k}"] = dec(p1, s1), dec(p2, s2, zeros=True)
        fields += [S.decimal(p1, s1), S.decimal(p2, s2)]
        rt = S.decimal(min(p1 - s1, p2 - s2) + max(s1, s2), max(s1, s2))       # Spark's result type of Remainder (DecimalPrecision)
        outs.append(S.math("remainder", S.col(2 * k, fields[-2]), S.col(2 * k + 1, fields[-1]), rt))
    t = pa.table(cols)
    plan = S.project(S.scan(fields), outs)
    got = pa.Table.from_batches(_run(plan, table=t, ncols=len(outs), batch_size=0))
    want = _oracle(plan, t)
    for i in range(len(outs)):
        assert got.column(i).to_pylist() == want.column(i).to_pylist(), f"case {cases[i]}"
    assert got.column(0).null_count > t.column(0).null_count        # zero divisors became NULL
    ansi = S.project(S.scan(fields), [S.math("remainder", S.col(0, fields[0]), S.col(1, fields[1]), S.decimal(12, 2), S.ANSI)])
    with pytest.raises(native.CometQueryExecutionException, match="REMAINDER_BY_ZERO"):
        _run(ansi, table=t, ncols=1)


def test_more_casts(built):
    """Float/Decimal → integral (Rust `as`: saturating for floats, truncating for decimals; narrow types via i32 — numeric.rs:311-560),
    Decimal → Float, Double → Float, Boolean ↔ numeric; ANSI overflow → CAST_OVERFLOW."""
    from datafusion_comet_amd import tpch
    n = 30_000
    rng = np.random.default_rng(99)
    f = rng.standard_normal(n) * 1e5
    f[:10] = [np.nan, np.inf, -np.inf, 3e9, -3e9, 1e19, -1e19, 127.9, -128.9, 40000.5]
    t = pa.table({"f": pa.array(f, mask=rng.random(n) < 0.1), "d": tpch._dec128_array(rng.integers(-10**11, 10**11, n), 12, 2),
                  "w": pa.array([__import__("decimal").Decimal(int(x) * 10**12).scaleb(-6) for x in rng.integers(-10**17, 10**17, n)], pa.decimal128(38, 6)),
                  "b": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1), "i": pa.array(rng.integers(-3, 4, n), pa.int32())})
    D, W = S.decimal(12, 2), S.decimal(38, 6)
    fields = [S.T_DOUBLE, D, W, S.T_BOOL, S.T_INT32]
    F, Dc, Wc, B, I = (S.col(i, ty) for i, ty in enumerate(fields))
    outs = [S.cast(F, S.T_INT8), S.cast(F, S.T_INT16), S.cast(F, S.T_INT32), S.cast(F, S.T_INT64), S.cast(F, S.T_FLOAT),
            S.cast(Dc, S.T_INT8), S.cast(Dc, S.T_INT32), S.cast(Dc, S.T_INT64), S.cast(Wc, S.T_INT16), S.cast(Wc, S.T_INT64),
            S.cast(Dc, S.T_DOUBLE), S.cast(Wc, S.T_DOUBLE), S.cast(Dc, S.T_FLOAT),
            S.cast(B, S.T_INT32), S.cast(B, S.T_DOUBLE), S.cast(I, S.T_BOOL), S.cast(F, S.T_BOOL), S.cast(Dc, S.T_BOOL), S.cast(Wc, S.T_BOOL)]
    plan = S.project(S.scan(fields), outs)
    got = pa.Table.from_batches(_run(plan, table=t, ncols=len(outs), batch_size=0))
    want = _oracle(plan, t)
    for i in range(len(outs)):
        g_, w_ = got.column(i).combine_chunks(), want.column(i).combine_chunks()
        assert g_.type == w_.type, i
        assert g_.is_valid().equals(w_.is_valid()), i
        if pa.types.is_boolean(g_.type):
            assert g_.equals(w_), i
        else:
            assert g_.fill_null(0).to_numpy().tobytes() == w_.fill_null(0).to_numpy().tobytes(), f"column {i}"
    for e in (S.cast(F, S.T_INT32, S.ANSI), S.cast(Wc, S.T_INT16, S.ANSI)):
        with pytest.raises(native.CometQueryExecutionException, match="CAST_OVERFLOW"):
            _run(S.project(S.scan(fields), [e]), table=t, ncols=1)


def test_round_and_date_arithmetic(built):
    """spark_round (math_funcs/round.rs:160-260: HALF_UP on Decimal128 at positive / zero / negative positions, Int32 / Int64 at negative
    positions incl. wrap-around and ANSI overflow) and date_add / date_sub / date_diff (wrapping day arithmetic)."""
    from datafusion_comet_amd import tpch
    import decimal
    n = 40_000
    rng = np.random.default_rng(5)
    i64 = rng.integers(-10**12, 10**12, n)
    i64[:6] = [2**63 - 1, -2**63, 5, -5, 15, -15]
    i32 = rng.integers(-2**31, 2**31, n).astype(np.int32)
    i32[:4] = [2**31 - 1, -2**31, 2147483645, -2147483645]
    t = pa.table({"d": tpch._dec128_array(rng.integers(-10**11, 10**11, n), 12, 2),
                  "w": pa.array([decimal.Decimal(int(x) * 10**11 + 55555).scaleb(-6) for x in rng.integers(-10**17, 10**17, n)], pa.decimal128(38, 6)),
                  "l": pa.array(i64, mask=rng.random(n) < 0.05), "i": pa.array(i32),
                  "dt": pa.array(rng.integers(-20000, 40000, n).astype(np.int32), pa.int32()).cast(pa.date32()),
                  "dt2": pa.array(rng.integers(0, 20000, n).astype(np.int32), pa.int32(), mask=rng.random(n) < 0.1).cast(pa.date32()),
                  "k": pa.array(rng.integers(-400, 400, n).astype(np.int32))})
    D, W = S.decimal(12, 2), S.decimal(38, 6)
    fields = [D, W, S.T_INT64, S.T_INT32, S.T_DATE, S.T_DATE, S.T_INT32]
    d, w, l, i, dt, dt2, k = (S.col(j, ty) for j, ty in enumerate(fields))
    P = lambda v: S.lit(v, S.T_INT64)
    rnd = lambda x, p, ty, **kw: S.scalar_func("round", [x, P(p)], ty, **kw)
    outs = [rnd(d, 1, S.decimal(12, 1)), rnd(d, 0, S.decimal(11, 0)), rnd(d, 2, S.decimal(12, 2)), rnd(d, 5, S.decimal(12, 2)), rnd(d, -2, S.decimal(11, 0)),
            rnd(w, 3, S.decimal(36, 3)), rnd(w, -4, S.decimal(33, 0)), rnd(w, 0, S.decimal(33, 0)),
            rnd(l, -1, S.T_INT64), rnd(l, -3, S.T_INT64), rnd(l, -18, S.T_INT64), rnd(i, -2, S.T_INT32), rnd(i, -9, S.T_INT32),
            S.scalar_func("date_add", [dt, k], S.T_DATE), S.scalar_func("date_sub", [dt, k], S.T_DATE), S.scalar_func("date_diff", [dt, dt2], S.T_INT32)]
    for at in range(0, len(outs), 8):
        chunk = outs[at:at + 8]
        plan = S.project(S.scan(fields), chunk)
        got = pa.Table.from_batches(_run(plan, table=t, ncols=len(chunk), batch_size=0))
        want = _oracle(plan, t)
        for c in range(len(chunk)):
            assert got.column(c).combine_chunks().equals(want.column(c).combine_chunks()), at + c
    # ANSI: rounding Long.MaxValue to tens leaves the type (an Int32 cannot overflow at the positions its powers of ten allow)
    from oracle import oracle as O
    e = rnd(l, -1, S.T_INT64, fail_on_error=True)
    with pytest.raises(native.CometQueryExecutionException, match="ARITHMETIC_OVERFLOW"):
        _run(S.project(S.scan(fields), [e]), table=t, ncols=1)
    with pytest.raises(O.OracleError, match="ARITHMETIC_OVERFLOW"):
        _oracle(S.project(S.scan(fields), [e]), t)
    ok = rnd(i, -2, S.T_INT32, fail_on_error=True)
    assert pa.Table.from_batches(_run(S.project(S.scan(fields), [ok]), table=t, ncols=1)).num_rows == n


def test_bitwise_and_shifts(built):
    """BitwiseAnd / BitwiseOr / BitwiseXor (expr.proto:46-48 tags 34-36) and ShiftRight / ShiftLeft (tags 42-43): Java semantics — the
    count is taken modulo the value's width, >> is arithmetic, Byte / Short results wrap to their width; NULL if either side is."""
    n = 30_000
    rng = np.random.default_rng(9)
    i64 = rng.integers(-2**63, 2**63 - 1, n)
    i64[:4] = [2**63 - 1, -2**63, -1, 0]
    i32 = rng.integers(-2**31, 2**31, n).astype(np.int32)
    t = pa.table({"l": pa.array(i64, mask=rng.random(n) < 0.05), "l2": pa.array(rng.integers(-2**63, 2**63 - 1, n)),
                  "i": pa.array(i32), "i2": pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32), mask=rng.random(n) < 0.05),
                  "s": pa.array(rng.integers(-2**15, 2**15, n).astype(np.int16)), "s2": pa.array(rng.integers(-2**15, 2**15, n).astype(np.int16)),
                  "k": pa.array(rng.integers(-70, 70, n).astype(np.int32))})
    fields = [S.T_INT64, S.T_INT64, S.T_INT32, S.T_INT32, S.T_INT16, S.T_INT16, S.T_INT32]
    l, l2, i, i2, s, s2, k = (S.col(j, ty) for j, ty in enumerate(fields))
    outs = [S.bit_and(l, l2), S.bit_or(l, l2), S.bit_xor(l, l2), S.shift_left(l, k), S.shift_right(l, k),
            S.bit_and(i, i2), S.bit_or(i, i2), S.bit_xor(i, i2), S.shift_left(i, k), S.shift_right(i, k),
            S.bit_and(s, s2), S.bit_or(s, s2), S.bit_xor(s, s2), S.bit_and(i, S.lit(0xFF, S.T_INT32))]
    for at in range(0, len(outs), 7):
        chunk = outs[at:at + 7]
        plan = S.project(S.scan(fields), chunk)
        got = pa.Table.from_batches(_run(plan, table=t, ncols=len(chunk), batch_size=0))
        want = _oracle(plan, t)
        for c in range(len(chunk)):
            assert got.column(c).combine_chunks().equals(want.column(c).combine_chunks()), at + c
    # and as a predicate: flags & 4 != 0
    plan = S.filter_(S.scan(fields), S.neq(S.bit_and(i, S.lit(4, S.T_INT32)), S.lit(0, S.T_INT32)))
    got = pa.Table.from_batches(_run(plan, table=t, ncols=len(fields), batch_size=0))
    assert got.equals(_oracle(plan, t))


def test_integral_divide(built):
    """Spark `a div b` in the shape CometIntegralDivide emits (serde/arithmetic.scala:283-345 → decimal_integral_div, div.rs:40-165):
    truncation toward zero, zero divisor → NULL (legacy) / DIVIDE_BY_ZERO (ANSI), Long.MinValue div -1 wraps (legacy) or raises with
    check_divide_overflow under ANSI; decimal operands of different scales."""
    from datafusion_comet_amd import tpch
    from oracle import oracle as O
    n = 30_000
    rng = np.random.default_rng(13)
    a = rng.integers(-2**63, 2**63 - 1, n)
    b = rng.integers(-1000, 1000, n)
    a[:6] = [-2**63, 2**63 - 1, 7, -7, 7, -7]
    b[:6] = [-1, -1, 2, 2, -2, -2]
    b[6:40] = 0
    t = pa.table({"a": pa.array(a, mask=rng.random(n) < 0.03), "b": pa.array(b, mask=rng.random(n) < 0.03),
                  "i": pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32)), "j": pa.array(rng.integers(-50, 50, n).astype(np.int32)),
                  "d": tpch._dec128_array(rng.integers(-10**11, 10**11, n), 12, 2), "e": tpch._dec128_array(rng.integers(-10**6, 10**6, n), 7, 3)})
    D, E = S.decimal(12, 2), S.decimal(7, 3)
    fields = [S.T_INT64, S.T_INT64, S.T_INT32, S.T_INT32, D, E]
    ca, cb, ci, cj, cd, ce = (S.col(k, ty) for k, ty in enumerate(fields))
    outs = [S.integral_divide(ca, S.T_INT64, cb, S.T_INT64), S.integral_divide(ci, S.T_INT32, cj, S.T_INT32),
            S.integral_divide(cd, D, ce, E), S.integral_divide(ce, E, cd, D), S.integral_divide(ca, S.T_INT64, S.lit(10, S.T_INT64), S.T_INT64)]
    plan = S.project(S.scan(fields), outs)
    got = pa.Table.from_batches(_run(plan, table=t, ncols=len(outs), batch_size=0))
    want = _oracle(plan, t)
    for c in range(len(outs)):
        assert got.column(c).combine_chunks().equals(want.column(c).combine_chunks()), c
    # pinned by hand: truncation toward zero, MIN div -1 wraps, x div 0 is NULL
    assert got.column(0).slice(0, 7).to_pylist() == [-2**63, -(2**63 - 1), 3, -3, -3, 3, None]
    # ANSI: zero divisor raises; with the divisor kept away from zero MIN div -1 raises only when check_divide_overflow is set
    ansi = S.project(S.scan(fields), [S.integral_divide(ca, S.T_INT64, cb, S.T_INT64, eval_mode=S.ANSI)])
    with pytest.raises(native.CometQueryExecutionException, match="DIVIDE_BY_ZERO"):
        _run(ansi, table=t, ncols=1)
    with pytest.raises(O.OracleError, match="DIVIDE_BY_ZERO"):
        _oracle(ansi, t)
    t2 = t.set_column(1, "b", pa.array(np.where(b == 0, 3, b)))
    got = pa.Table.from_batches(_run(ansi, table=t2, ncols=1, batch_size=0))
    assert got.column(0).combine_chunks().equals(_oracle(ansi, t2).column(0).combine_chunks())
    chk = S.project(S.scan(fields), [S.integral_divide(ca, S.T_INT64, cb, S.T_INT64, eval_mode=S.ANSI, check_divide_overflow=True)])
    with pytest.raises(native.CometQueryExecutionException, match="ARITHMETIC_OVERFLOW"):
        _run(chk, table=t2, ncols=1)
    with pytest.raises(O.OracleError, match="ARITHMETIC_OVERFLOW"):
        _oracle(chk, t2)


def test_murmur3_hash_expression(built):
    """Spark's hash(...) = murmur3_hash(cols..., seed literal) and xxhash64(cols..., seed) (hash_funcs/murmur3.rs:24-70): every non-NULL value folds into the running
    hash in argument order, per-type encodings as in the partitioning hash, never NULL; also over a computed operand."""
    from datafusion_comet_amd import tpch
    n = 40_000
    rng = np.random.default_rng(21)
    f64 = rng.standard_normal(n)
    f64[:3] = [0.0, -0.0, np.nan]
    t = pa.table({"b": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1), "i8": pa.array(rng.integers(-128, 128, n).astype(np.int8)),
                  "i32": pa.array(rng.integers(-2**31, 2**31, n).astype(np.int32), mask=rng.random(n) < 0.1),
                  "i64": pa.array(rng.integers(-2**63, 2**63 - 1, n), mask=rng.random(n) < 0.1),
                  "f32": pa.array(rng.standard_normal(n).astype(np.float32)), "f64": pa.array(f64, mask=rng.random(n) < 0.1),
                  "d": tpch._dec128_array(rng.integers(-10**11, 10**11, n), 12, 2), "w": tpch._dec128_array(rng.integers(-10**17, 10**17, n), 30, 4),
                  "dt": pa.array(rng.integers(-20000, 40000, n).astype(np.int32), pa.int32()).cast(pa.date32())})
    D, W = S.decimal(12, 2), S.decimal(30, 4)
    fields = [S.T_BOOL, S.T_INT8, S.T_INT32, S.T_INT64, S.T_FLOAT, S.T_DOUBLE, D, W, S.T_DATE]
    c = [S.col(j, ty) for j, ty in enumerate(fields)]
    seed = S.lit(42, S.T_INT32)
    h = lambda *xs, s=seed: S.scalar_func("murmur3_hash", list(xs) + [s], S.T_INT32)
    outs = [h(c[0]), h(c[1]), h(c[2]), h(c[3]), h(c[4]), h(c[5]), h(c[6]), h(c[7]), h(c[8]), h(*c), h(c[2], c[3], s=S.lit(-7, S.T_INT32)),
            h(S.math("add", c[3], S.lit(1, S.T_INT64), S.T_INT64))]
    # xxhash64 (hash_funcs/xxhash64.rs:31-82): Int64 seed and result, the same value encodings
    x = lambda *xs, s=S.lit(42, S.T_INT64): S.scalar_func("xxhash64", list(xs) + [s], S.T_INT64)
    outs += [x(c[0]), x(c[1]), x(c[2]), x(c[3]), x(c[4]), x(c[5]), x(c[6]), x(c[7]), x(c[8]), x(*c), x(c[2], c[3], s=S.lit(-7, S.T_INT64)), x(h(c[3]))]
    for at in range(0, len(outs), 6):
        chunk = outs[at:at + 6]
        plan = S.project(S.scan(fields), chunk)
        got = pa.Table.from_batches(_run(plan, table=t, ncols=len(chunk), batch_size=0))
        want = _oracle(plan, t)
        for k in range(len(chunk)):
            assert got.column(k).null_count == 0
            assert got.column(k).combine_chunks().equals(want.column(k).combine_chunks()), at + k
    # pmod(hash(keys), n) as computed in a projection equals the shuffle writer's partition ids for the same keys
    from oracle import oracle as O
    ids = O.hash_partition_ids(S, t, [2, 3], 200)
    plan = S.project(S.scan(fields), [h(c[2], c[3])])
    hv = pa.Table.from_batches(_run(plan, table=t, ncols=1, batch_size=0)).column(0).to_numpy().astype(np.int64)
    assert ((hv % 200 + 200) % 200 == ids).all()


def test_coalesce(built):
    """coalesce(a, b, …): the first non-NULL argument — integers, doubles, decimals whose arguments sit in different widths, dates, packed
    strings, a literal fallback that makes the result non-nullable, and as a filter operand."""
    from decimal import Decimal
    from oracle import oracle as O
    rng = np.random.default_rng(23)
    n = 50_000
    m = lambda p: rng.random(n) < p
    D = S.decimal(12, 2)
    t = pa.table({"a": pa.array(rng.integers(-10**9, 10**9, n), pa.int64(), mask=m(0.5)), "b": pa.array(rng.integers(-10**9, 10**9, n), pa.int64(), mask=m(0.5)),
                  "x": pa.array(rng.standard_normal(n), mask=m(0.4)), "y": pa.array(rng.standard_normal(n), mask=m(0.4)),
                  "d": pa.array([Decimal(int(v)).scaleb(-2) for v in rng.integers(-10**11, 10**11, n)], pa.decimal128(12, 2), mask=m(0.6)),
                  "e": pa.array([Decimal(int(v)).scaleb(-2) for v in rng.integers(-100, 100, n)], pa.decimal128(12, 2), mask=m(0.3)),
                  "s": pa.array(np.array(["", "A", "N", "lineitem"], dtype=object)[rng.integers(0, 4, n)], pa.utf8(), mask=m(0.5))})
    types = [S.T_INT64, S.T_INT64, S.T_DOUBLE, S.T_DOUBLE, D, D, S.T_STRING]
    c = lambda i: S.col(i, types[i])
    co = lambda args, ty: S.scalar_func("coalesce", args, ty)
    exprs = [co([c(0), c(1)], S.T_INT64), co([c(0), c(1), S.lit(-1, S.T_INT64)], S.T_INT64), co([c(2), c(3)], S.T_DOUBLE), co([c(4), c(5)], D),
             co([c(4), S.math("add", c(5), c(5), D)], D), co([c(6), S.lit("none", S.T_STRING)], S.T_STRING), co([c(0)], S.T_INT64)]
    plan = S.project(S.filter_(S.scan(types), S.gt(co([c(0), c(1), S.lit(0, S.T_INT64)], S.T_INT64), S.lit(-5 * 10**8, S.T_INT64))), exprs)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], len(exprs), plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, t)
    assert got.num_rows == want.num_rows and 0 < got.num_rows < n
    for i in range(len(exprs)):
        assert got.column(i).to_pylist() == want.column(i).to_pylist(), f"expression {i}"
    assert got.column(1).null_count == 0


def test_the_references_modulo_vectors(built):
    """modulo_expr.rs:360-985 — the cases tests/test_modulo_kats_cpu.py transcribes — through the C ABI: -0.0 is a zero divisor, NULL operands
    never raise, NaN / ±Infinity dividends, literal operands; ANSI raises REMAINDER_BY_ZERO (common/src/error.rs:81-82, 684)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("modulo_kats", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_modulo_kats_cpu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for args, kw, want in m.CASES:
        plan, table = m.build(*args, **kw)
        if want == "raise":
            with pytest.raises(native.CometQueryExecutionException, match='"errorClass":"REMAINDER_BY_ZERO"'):
                _run(plan, table, 1)
        else:
            got = pa.Table.from_batches(_run(plan, table, 1)).column(0).to_pylist()
            assert m.same(got, want), (args, kw, got)


def test_try_casts_to_integers(built):
    """try_cast(number AS integer type): the reference leaves these to arrow's cast with safe = true (cast.rs:284-293, 311-326 `if eval_mode != Try`,
    :236-241, 401-407) — the value truncated toward zero, NULL when that does not fit the target type, NaN included; LEGACY wraps / saturates and
    ANSI raises (test_more_casts)"""
    from datafusion_comet_amd.tpch import _dec128_array
    ints = [0, 1, -1, 127, 128, -128, -129, 32767, 32768, -32768, -32769, 2**31 - 1, 2**31, -2**31, -2**31 - 1, 2**63 - 1, -2**63, 123456789012]
    floats = [0.0, -0.0, 0.9, -0.9, 127.9, 128.0, -128.9, -129.0, 32767.5, 32768.0, -32768.9, -32769.0, 2147483647.9, 2147483648.0, -2147483648.9, -2147483649.0,
              9.2233720368547748e18, 9.223372036854775807e18, -9.223372036854775808e18, -9.3e18, 1e300, -1e300, float("inf"), float("-inf"), float("nan"), 1.5e10]
    n = max(len(ints), len(floats))
    pad = lambda v: v + [v[0]] * (n - len(v))
    decs = [0, 99, -99, 12799, 12800, -12899, -12900, 3276799, 3276800, 214748364799, 214748364800, -214748364899, -214748364900, 922337203685477580799, 922337203685477580800,
            -922337203685477580899, -922337203685477580900, 10**30]
    W = S.decimal(38, 2)
    lo = np.array([v & (2**64 - 1) for v in pad(decs)], np.uint64)
    hi = np.array([(v >> 64) & (2**64 - 1) for v in pad(decs)], np.uint64)
    wide = pa.Array.from_buffers(pa.decimal128(38, 2), n, [None, pa.py_buffer(np.stack([lo, hi], axis=1).tobytes())])
    with np.errstate(over="ignore"):      # (1e300 as a float is infinity: meant)
        f32 = np.array(pad(floats), np.float32)
    t = pa.table({"i": pa.array(pad(ints), pa.int64()), "f": pa.array(pad(floats), pa.float64()), "g": pa.array(f32), "d": wide})
    fields = [S.T_INT64, S.T_DOUBLE, S.T_FLOAT, W]
    i, f, g, d = (S.col(k, ty) for k, ty in enumerate(fields))
    outs = [S.cast(src, to, S.TRY) for src in (i, f, g, d) for to in (S.T_INT8, S.T_INT16, S.T_INT32, S.T_INT64) if not (src is i and to == S.T_INT64)]
    plan = S.project(S.scan(fields), outs)
    got, want = pa.Table.from_batches(_run(plan, t, len(outs), batch_size=0)), _oracle(plan, t)
    for k in range(len(outs)):
        assert got.column(k).to_pylist() == want.column(k).to_pylist(), k
    # spot checks of the rule itself: 128 does not fit a tinyint, -128.9 truncates to -128, NaN is NULL, 2^63 as a double does not fit a bigint
    col = lambda k: got.column(k).to_pylist()
    assert col(0)[3:7] == [127, None, -128, None]
    f8 = col(3)
    assert f8[4:8] == [127, None, -128, None] and f8[24] is None and f8[22] is None
    assert col(6)[16:20] == [9223372036854774784, None, -2**63, None]      # (the largest double below 2^63 fits; 2^63 itself does not)


def test_unary_minus(built):
    """NegativeExpr (math_funcs/negative.rs:100-160) in the generated kernels vs the oracle: tinyint / smallint / int / bigint wrap IN THEIR OWN WIDTH in
    LEGACY (the minimum negates onto itself), floats flip the sign bit (-0.0, NaN, infinities), narrow and wide decimals change sign; with
    fail_on_error a VALID minimum raises ARITHMETIC_OVERFLOW naming "byte" / "short" / "integer" / "long" (negative.rs:136-150), a minimum that
    sits in a NULL slot does not."""
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    n = 70_001

    def dec(vals, p, sc):
        lo = np.array([v & (2**64 - 1) for v in vals], np.uint64)
        hi = np.array([(v >> 64) & (2**64 - 1) for v in vals], np.uint64)
        return pa.Array.from_buffers(pa.decimal128(p, sc), len(vals), [None, pa.py_buffer(np.stack([lo, hi], axis=1).tobytes())])
    m = lambda: rng.random(n) < 0.1

    def ints(np_t, bits):
        lo, hi = -(1 << (bits - 1)), (1 << (bits - 1)) - 1
        v = rng.integers(lo, hi, n, dtype=np.int64, endpoint=True)
        v[:6] = [lo, hi, 0, -1, 1, lo + 1]
        v[rng.integers(6, n, 50)] = lo
        return v.astype(np_t)
    f64 = rng.standard_normal(n) * 1e9
    f64[:7] = [0.0, -0.0, float("inf"), float("-inf"), float("nan"), 5e-324, -1.7976931348623157e308]
    with np.errstate(over="ignore"):
        f32 = f64.astype(np.float32)
    narrow = [int(x) for x in rng.integers(-10**11, 10**11, n)]
    wide = [int(x) * 10**19 + int(y) for x, y in zip(rng.integers(-10**18, 10**18, n), rng.integers(0, 10**18, n))]
    wide[:3] = [10**38 - 1, -(10**38 - 1), 0]
    DN, DW = S.decimal(12, 2), S.decimal(38, 4)
    fields = [S.T_INT8, S.T_INT16, S.T_INT32, S.T_INT64, S.T_FLOAT, S.T_DOUBLE, DN, DW]

    def with_mask(arr, mask):
        return pa.Array.from_buffers(arr.type, len(arr), [pa.array(~mask).buffers()[1]] + arr.buffers()[1:])
    t = pa.table({"b": pa.array(ints(np.int8, 8), mask=m()), "s": pa.array(ints(np.int16, 16), mask=m()), "i": pa.array(ints(np.int32, 32), mask=m()),
                  "l": pa.array(ints(np.int64, 64), mask=m()), "f": pa.array(f32, mask=m()), "d": pa.array(f64, mask=m()),
                  "n": with_mask(dec(narrow, 12, 2), m()), "w": with_mask(dec(wide, 38, 4), m())})
    cols = [S.col(k, ty) for k, ty in enumerate(fields)]
    outs = [S.Expr("unary_minus", [c]) for c in cols]
    # (negation inside a larger expression: the narrow wrap must happen BEFORE the widening cast, -(-128 as tinyint) = -128, then +1 as int = -127)
    outs.append(S.math("add", S.cast(S.Expr("unary_minus", [cols[0]]), S.T_INT32), S.lit(1, S.T_INT32), S.T_INT32))
    outs.append(S.Expr("unary_minus", [S.Expr("unary_minus", [cols[1]])]))
    plan = S.project(S.scan(fields), outs)
    got, want = pa.Table.from_batches(_run(plan, t, len(outs), batch_size=0)), _oracle(plan, t)
    for k in range(len(outs)):
        g, w = got.column(k).combine_chunks(), want.column(k).combine_chunks()
        if pa.types.is_floating(g.type):        # bit for bit, NaN sign included
            it = np.int32 if g.type == pa.float32() else np.int64
            assert g.is_valid().equals(w.is_valid()), k
            gm, wm = g.fill_null(0).to_numpy().view(it), w.fill_null(0).to_numpy().view(it)
            assert np.array_equal(gm, wm), k
        else:
            assert g.equals(w), k
    # as a filter predicate and below a filter: -x > 100 keeps exactly the oracle's rows
    fplan = S.project(S.filter_(S.scan(fields), S.gt(S.Expr("unary_minus", [cols[2]]), S.lit(100, S.T_INT32))), [cols[2], S.Expr("unary_minus", [cols[3]])])
    g2, w2 = pa.Table.from_batches(_run(fplan, t, 2, batch_size=0)), _oracle(fplan, t)
    assert g2.num_rows == w2.num_rows and g2.column(0).combine_chunks().equals(w2.column(0).combine_chunks()) and g2.column(1).combine_chunks().equals(w2.column(1).combine_chunks())
    # ANSI: each integer width raises on a valid minimum with the reference's type name, and not on a hidden one
    for ty, arrow, bits, name in ((S.T_INT8, pa.int8(), 8, "byte"), (S.T_INT16, pa.int16(), 16, "short"), (S.T_INT32, pa.int32(), 32, "integer"), (S.T_INT64, pa.int64(), 64, "long")):
        lo = -(1 << (bits - 1))
        neg = S.project(S.scan([ty]), [S.Expr("unary_minus", [S.col(0, ty)], fail_on_error=True)])
        bad = pa.table({"v": pa.array([5, lo, 7], arrow)})
        with pytest.raises(native.CometQueryExecutionException, match=f'ARITHMETIC_OVERFLOW.*"fromType":"{name}"'):
            _run(neg, bad, 1)
        with pytest.raises(O.OracleError, match="ARITHMETIC_OVERFLOW"):
            _oracle(neg, bad)
        hidden = pa.Array.from_buffers(arrow, 3, [pa.py_buffer(bytes([0b101])), pa.py_buffer(np.array([lo + 1, lo, 7]).astype(arrow.to_pandas_dtype()).tobytes())])
        ok = pa.table({"v": hidden})
        assert pa.Table.from_batches(_run(neg, ok, 1)).column(0).to_pylist() == _oracle(neg, ok).column(0).to_pylist() == [-(lo + 1), None, -7]
    # ANSI leaves floats and decimals alone
    fneg = S.project(S.scan([S.T_DOUBLE, DW]), [S.Expr("unary_minus", [S.col(0, S.T_DOUBLE)], fail_on_error=True), S.Expr("unary_minus", [S.col(1, DW)], fail_on_error=True)])
    ft = pa.table({"d": t.column("d"), "w": t.column("w")})
    g3, w3 = pa.Table.from_batches(_run(fneg, ft, 2, batch_size=0)), _oracle(fneg, ft)
    assert g3.column(1).combine_chunks().equals(w3.column(1).combine_chunks())
