"""Utf8 columns of ANY length carried through the GPU operators (first step): the emit kernels write source row
indices, the strings are gathered afterwards (lengths → scan → copy).  Filter/Project, hash joins (payload on both sides, outer
NULL extension), Sort/TopK, Limit, Parquet scans and device-resident outputs, against the oracle."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu

WORDS = np.array(["", "a", "Customer#000000001", "the quick brown fox jumps over the lazy dog " * 3, "naïve café ☕", "x" * 300, "BUILDING"], dtype=object)


def _strings(rng, n, null_frac=0.1):
    return pa.array(WORDS[rng.integers(0, len(WORDS), n)], pa.utf8(), mask=(rng.random(n) < null_frac) if null_frac else None)


def _run(plan, tables, ncols, **kw):
    out = native.execute_to_table([native.HostInput.from_table(t) for t in tables], ncols, plan.encode(), batch_size=0, **kw)
    return pa.Table.from_batches(out) if out else None


def _oracle(plan, tables):
    from oracle import oracle as O
    return O.run_plan_to_arrow(S, plan, tables)


def _rows(t):
    return sorted(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]), key=lambda r: tuple((x is None, str(x)) for x in r))


def test_filter_project_carries_long_strings(built):
    rng = np.random.default_rng(1)
    n = 120_000
    t = pa.table({"k": pa.array(rng.integers(0, 100, n), pa.int32()), "s": _strings(rng, n), "s2": _strings(rng, n, 0)})
    plan = S.project(S.filter_(S.scan([S.T_INT32, S.T_STRING, S.T_STRING]), S.lt(S.col(0, S.T_INT32), S.lit(37, S.T_INT32))),
                     [S.col(2, S.T_STRING), S.math("add", S.col(0, S.T_INT32), S.lit(1, S.T_INT32), S.T_INT32), S.col(1, S.T_STRING)])
    want = _oracle(plan, [t])
    got = _run(plan, [t], 3)
    assert got.num_rows == want.num_rows > 40_000
    for i in range(3):
        assert got.column(i).combine_chunks().equals(want.column(i).combine_chunks()), i      # FilterExec keeps the input order
    # device-resident input and output
    dev = native.execute_to_device([native.DeviceInput(native.DeviceTable.from_arrow(t))], 3, plan.encode())
    assert dev.to_arrow().column(2).combine_chunks().equals(want.column(2).combine_chunks())
    # no filter (projection only) and an empty result
    p2 = S.project(S.scan([S.T_INT32, S.T_STRING, S.T_STRING]), [S.col(1, S.T_STRING)])
    assert _run(p2, [t], 1).column(0).combine_chunks().equals(t.column("s").combine_chunks())
    none = S.filter_(S.scan([S.T_INT32, S.T_STRING, S.T_STRING]), S.lt(S.col(0, S.T_INT32), S.lit(-1, S.T_INT32)))
    assert _run(none, [t], 3) is None


@pytest.mark.parametrize("jt,build", [(S.INNER, S.BUILD_LEFT), (S.INNER, S.BUILD_RIGHT), (S.LEFT_OUTER, S.BUILD_RIGHT), (S.FULL_OUTER, S.BUILD_LEFT),
                                      (S.LEFT_SEMI, S.BUILD_RIGHT), (S.LEFT_ANTI, S.BUILD_LEFT)])
def test_join_with_string_payload(built, jt, build):
    rng = np.random.default_rng(2)
    nl, nr = 6000, 4000
    left = pa.table({"k": pa.array(rng.integers(0, 3000, nl), pa.int64(), mask=rng.random(nl) < 0.05), "name": _strings(rng, nl)})
    right = pa.table({"k": pa.array(rng.integers(0, 3500, nr), pa.int64()), "comment": _strings(rng, nr, 0), "v": pa.array(rng.random(nr))})
    plan = S.hash_join(S.scan([S.T_INT64, S.T_STRING]), S.scan([S.T_INT64, S.T_STRING, S.T_DOUBLE]), [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], jt, build)
    ncols = 2 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 5
    got, want = _run(plan, [left, right], ncols), _oracle(plan, [left, right])
    assert got.num_rows == want.num_rows > 0
    assert _rows(got) == _rows(want)


def test_sort_limit_and_parquet_with_strings(built, tmp_path):
    import pyarrow.parquet as papq
    rng = np.random.default_rng(3)
    n = 30_000
    t = pa.table({"id": pa.array(rng.permutation(n), pa.int64()), "s": _strings(rng, n), "d": pa.array(rng.integers(8000, 9000, n), pa.int32()).cast(pa.date32())})
    fields = [S.T_INT64, S.T_STRING, S.T_DATE]
    top = S.sort(S.scan(fields), [(S.col(0, S.T_INT64), True, True)], fetch=500, skip=3)
    got, want = _run(top, [t], 3), _oracle(top, [t])
    assert got.equals(want.rename_columns(got.column_names)) if got.schema.equals(want.schema) else [c.to_pylist() for c in got.columns] == [c.to_pylist() for c in want.columns]
    full = S.sort(S.scan(fields), [(S.col(2, S.T_DATE), False, False), (S.col(0, S.T_INT64), False, False)])
    got, want = _run(full, [t], 3), _oracle(full, [t])
    assert [c.to_pylist() for c in got.columns] == [c.to_pylist() for c in want.columns]
    lim = S.limit(S.scan(fields), 77, 5)
    assert [c.to_pylist() for c in _run(lim, [t], 3).columns] == [c.to_pylist() for c in t.slice(5, 72).columns]
    # Parquet scan (dictionary + plain strings) → filter → project with the string carried along
    path = str(tmp_path / "s.parquet")
    papq.write_table(t, path, row_group_size=7000, compression="snappy")
    src = S.native_scan([path], t.schema.names, fields)
    plan = S.project(S.filter_(src, S.gt(S.col(0, S.T_INT64), S.lit(n // 2, S.T_INT64))), [S.col(1, S.T_STRING), S.col(0, S.T_INT64)])
    got = pa.Table.from_batches(native.execute_to_table([], 2, plan.encode(), batch_size=0))
    keep = t.filter(pa.compute.greater(t.column("id"), n // 2))
    assert got.column(0).combine_chunks().equals(keep.column("s").combine_chunks()) and got.column(1).equals(keep.column("id"))


def test_string_predicates_any_length(built):
    """=, <>, <, <=, >, >=, <=>, IN between a Utf8 column and literals / another column: bytes compared in place (unsigned
    lexicographic, like Spark's UTF8String and arrow-ord), no 15-byte limit, NULLs three-valued."""
    rng = np.random.default_rng(5)
    n = 80_000
    t = pa.table({"a": _strings(rng, n), "b": _strings(rng, n, 0.05), "id": pa.array(np.arange(n), pa.int64())})
    A, B = S.col(0, S.T_STRING), S.col(1, S.T_STRING)
    lit = lambda x: S.lit(x, S.T_STRING)
    long_lit = "the quick brown fox jumps over the lazy dog " * 3
    preds = [S.eq(A, lit("Customer#000000001")), S.neq(A, lit("BUILDING")), S.gt_eq(A, lit("a")), S.lt(A, lit(long_lit)), S.gt(lit("n"), A),
             S.lt_eq(A, B), S.neq(A, B), S.eq(B, A), S.in_(A, [lit("BUILDING"), lit(long_lit), lit("")]), S.in_(A, [lit("x" * 300)], negated=True),
             S.or_(S.eq(A, lit("naïve café ☕")), S.is_null(A)),
             S.eq_null_safe(A, B), S.not_(S.eq_null_safe(A, lit("a")))]
    for i, pred in enumerate(preds):
        plan = S.project(S.filter_(S.scan([S.T_STRING, S.T_STRING, S.T_INT64]), pred), [S.col(2, S.T_INT64)])
        got, want = _run(plan, [t], 1), _oracle(plan, [t])
        assert (got.column(0).to_pylist() if got is not None else []) == want.column(0).to_pylist(), f"predicate {i}"
        assert want.num_rows > 0


def test_like_and_string_predicates(built):
    """LIKE (Expr.like = 26) with %, _ and \\-escapes over multi-byte UTF-8, the prefix / suffix / substring fast paths, starts_with /
    ends_with / contains and the length functions — evaluated on the column bytes in place; oracle: Python's regex engine over code points."""
    from oracle import oracle as O
    rng = np.random.default_rng(41)
    words = ["PROMO BRUSHED COPPER", "STANDARD POLISHED BRASS", "special requests", "green almond", "forest green", "", "%", "a_b", "a%b", "50% off",
             "über-größe", "日本語テキスト", "naïve café", "x", "xy", "line\nbreak", "special packages requests", "Customer Complaints", "ab", "abc", "a\\b"]
    n = 40_000
    vals = [None if rng.random() < 0.05 else words[int(i)] + ("" if rng.random() < 0.7 else str(int(rng.integers(0, 50)))) for i in rng.integers(0, len(words), n)]
    t = pa.table({"s": pa.array(vals, pa.string()), "id": pa.array(np.arange(n), pa.int64())})
    fields = [S.T_STRING, S.T_INT64]
    s, i = S.col(0, S.T_STRING), S.col(1, S.T_INT64)
    L = lambda p: S.lit(p, S.T_STRING)
    patterns = ["%green%", "PROMO%", "%BRASS", "%special%requests%", "a_b", "a\\_b", "a\\%b", "%", "%%", "", "_", "__", "_%", "%_", "日本%", "%テ_スト", "über-gr__e",
                "%é", "x%y", "%\n%", "50\\% off", "%off%", "abc", "ab_", "%a%b%c%", "a\\\\b", "_b%", "%b_"]
    preds = [S.like(s, L(p)) for p in patterns]
    preds += [S.scalar_func(f, [s, L(x)], S.T_BOOL) for f in ("starts_with", "ends_with", "contains") for x in ("", "a", "green", "é", "语")]
    lens = [S.scalar_func(f, [s], S.T_INT32) for f in ("length", "octet_length", "bit_length")]
    exprs = preds + lens
    labels = patterns + ["fn"] * (len(exprs) - len(patterns))
    for at in range(0, len(exprs), 10):        # a pipeline carries a bounded number of output columns
        chunk = exprs[at:at + 10]
        plan = S.project(S.scan(fields), chunk + [i])
        got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], len(chunk) + 1, plan.encode(), batch_size=0))
        want = O.run_plan_to_arrow(S, plan, [t])
        for c in range(len(chunk) + 1):
            assert got.column(c).combine_chunks().equals(want.column(c).combine_chunks()), (at + c, labels[at + c] if at + c < len(labels) else "id")
    # as filters (Kleene: NULL rows drop), TPC-H style: p_type LIKE 'PROMO%', o_comment NOT LIKE '%special%requests%'
    fplan = S.project(S.filter_(S.scan(fields), S.and_(S.not_(S.like(s, L("%special%requests%"))), S.or_(S.like(s, L("PROMO%")), S.like(s, L("%green%"))))), [i, s])
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 2, fplan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, fplan, [t])
    assert got.column(0).to_pylist() == want.column(0).to_pylist() and got.column(1).to_pylist() == want.column(1).to_pylist()
    assert 0 < got.num_rows < n


def _key_rows(tb, nkeys):
    rows = list(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]))
    return sorted(rows, key=lambda r: tuple((x is None, x if x is not None else "") if isinstance(x, (str, type(None))) else (False, x) for x in r[:nkeys]))


@pytest.mark.parametrize("mode", ["partial", "final", "below_sort"])
def test_group_by_strings_of_any_length(built, mode):
    """Utf8 group keys longer than the 15 bytes that fit the packed key words (TPC-H Q10: c_name, c_address, c_comment; TPC-DS item ids):
    the keys travel as representative row indices (strdict_kernels.hip) and come back as gathered strings."""
    from oracle import oracle as O
    rng = np.random.default_rng(17)
    n = 60_000
    names = ["Customer#%09d" % i for i in range(400)]                                  # 18 bytes
    addrs = ["".join(chr(97 + (i * 7 + j) % 26) for j in range(5 + i % 36)) + " Straße %d" % i for i in range(300)]   # 15..50 bytes, multi-byte
    ni, ai = rng.integers(0, len(names), n), rng.integers(0, len(addrs), n)
    t = pa.table({
        "name": pa.array([None if rng.random() < 0.03 else names[int(i)] for i in ni], pa.string()),
        "addr": pa.array([None if rng.random() < 0.03 else addrs[int(i)] for i in ai], pa.string()),
        "flag": pa.array([["A", "N", "R"][int(i) % 3] for i in ni], pa.string()),
        "v": tpch._dec128_array(rng.integers(-10**8, 10**8, n), 12, 2),
    })
    D, SD = S.decimal(12, 2), S.decimal(22, 2)
    fields = [S.T_STRING, S.T_STRING, S.T_STRING, D]
    keys = [S.col(0, S.T_STRING), S.col(2, S.T_STRING), S.col(1, S.T_STRING)]
    aggs = [S.sum_(S.col(3, D), SD), S.count(S.col(3, D))]
    partial = S.hash_agg(S.scan(fields), keys, aggs, S.PARTIAL)
    run = lambda plan, tb, nc: pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(tb)], nc, plan.encode(), batch_size=0))
    if mode == "partial":
        got, want = run(partial, t, 6), O.run_plan_to_arrow(S, partial, [t])
        assert _key_rows(got, 3) == _key_rows(want, 3)
        return
    if mode == "final":
        states = pa.concat_tables([run(partial, t.slice(0, n // 2), 6), run(partial, t.slice(n // 2), 6)])
        final = S.final_of(partial, states.schema)
        got, want = run(final, states, 5), O.run_plan_to_arrow(S, final, [states])
        assert _key_rows(got, 3) == _key_rows(want, 3)
        one = O.run_plan_to_arrow(S, S.final_of(partial, states.schema), [O.run_plan_to_arrow(S, partial, [t])])
        assert _key_rows(got, 3) == _key_rows(one, 3)
        return
    # an aggregate below a Sort (nested context, device-resident result): ORDER BY sum DESC LIMIT 25 (sums are distinct here)
    plan = S.sort(partial, [(S.col(3, SD), True)], fetch=25)
    got, want = run(plan, t, 6), O.run_plan_to_arrow(S, plan, [t])
    assert got.to_pylist() == want.to_pylist()


@pytest.mark.parametrize("jt,build", [(S.INNER, S.BUILD_RIGHT), (S.LEFT_OUTER, S.BUILD_RIGHT), (S.FULL_OUTER, S.BUILD_LEFT), (S.LEFT_SEMI, S.BUILD_RIGHT),
                                      (S.LEFT_ANTI, S.BUILD_LEFT), (S.RIGHT_OUTER, S.BUILD_LEFT)])
def test_join_on_strings_of_any_length(built, jt, build):
    """Join keys that are Utf8 columns longer than 15 bytes (TPC-DS item ids, names): dictionary over the right column, lookup of the left,
    join on row indices; with a second (integer) key and a residual condition that addresses columns of both sides."""
    rng = np.random.default_rng(23)
    nl, nr = 9000, 5000
    ids = ["AAAAAAAA%08dXYZ-é" % i for i in range(1200)]          # 21 bytes, multi-byte tail
    pick = lambda n, hi: [None if rng.random() < 0.04 else ids[int(i)] for i in rng.integers(0, hi, n)]
    left = pa.table({"item": pa.array(pick(nl, 1000), pa.string()), "g": pa.array(rng.integers(0, 3, nl), pa.int32()), "q": pa.array(rng.integers(0, 100, nl), pa.int64())})
    right = pa.table({"w": pa.array(rng.integers(0, 100, nr), pa.int64()), "g": pa.array(rng.integers(0, 3, nr), pa.int32()), "item": pa.array(pick(nr, 1200), pa.string())})
    lf, rf = [S.T_STRING, S.T_INT32, S.T_INT64], [S.T_INT64, S.T_INT32, S.T_STRING]
    cond = S.lt(S.col(2, S.T_INT64), S.col(3, S.T_INT64)) if jt in (S.INNER, S.LEFT_SEMI) else None        # left.q < right.w
    plan = S.hash_join(S.scan(lf), S.scan(rf), [S.col(0, S.T_STRING), S.col(1, S.T_INT32)], [S.col(2, S.T_STRING), S.col(1, S.T_INT32)], jt, build, condition=cond)
    ncols = 3 if jt in (S.LEFT_SEMI, S.LEFT_ANTI) else 6
    got, want = _run(plan, [left, right], ncols), _oracle(plan, [left, right])
    assert got.num_rows == want.num_rows > 0
    assert got.schema.types == want.schema.types
    assert _rows(got) == _rows(want)


def test_substring_as_filter_and_group_key(built):
    """TPC-H Q22's shape: WHERE substring(c_phone, 1, 2) IN (…) GROUP BY substring(c_phone, 1, 2) — a computed string of ≤ 15 bytes built
    from the column bytes (Spark's 1-based, character-wise substring incl. pos 0, negative pos and clipping), used as predicate and key."""
    from oracle import oracle as O
    rng = np.random.default_rng(22)
    n = 50_000
    phones = ["%02d-%03d-%03d-%04d" % tuple(rng.integers(10, 35, 1).tolist() + rng.integers(100, 999, 3).tolist()) for _ in range(n)]
    odd = ["", "7", "ü1-x", "日本語テキスト-long-tail-beyond-fifteen-bytes", "ab"]
    vals = [None if rng.random() < 0.03 else (odd[int(rng.integers(0, len(odd)))] if rng.random() < 0.05 else phones[i]) for i in range(n)]
    t = pa.table({"c_phone": pa.array(vals, pa.string()), "c_acctbal": tpch._dec128_array(rng.integers(-99999, 999999, n), 12, 2)})
    D, SD = S.decimal(12, 2), S.decimal(22, 2)
    s, I = S.col(0, S.T_STRING), (lambda v: S.lit(v, S.T_INT32))
    code = S.scalar_func("substring", [s, I(1), I(2)], S.T_STRING)
    flt = S.in_(code, [S.lit(x, S.T_STRING) for x in ("13", "31", "23", "29", "30", "18", "17", "日本")])
    plan = S.hash_agg(S.filter_(S.scan([S.T_STRING, D]), S.and_(flt, S.gt(S.col(1, D), S.lit(__import__("decimal").Decimal("0.00"), D)))),
                      [code], [S.count(S.col(1, D)), S.sum_(S.col(1, D), SD)], S.PARTIAL)
    run = lambda p, nc: pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], nc, p.encode(), batch_size=0))
    got, want = run(plan, 4), O.run_plan_to_arrow(S, plan, [t])
    assert _rows(got) == _rows(want) and got.num_rows >= 7
    for pos, ln in ((0, 2), (-4, 3), (4, 100), (-100, 3), (2, 0), (3, -1), (-2, 15)):
        key = S.scalar_func("substring", [s, I(pos), I(ln)], S.T_STRING)
        keep = S.lt(S.scalar_func("octet_length", [s], S.T_INT32), S.lit(16, S.T_INT32))      # results stay within the packed 15 bytes
        p2 = S.hash_agg(S.filter_(S.scan([S.T_STRING, D]), keep), [key], [S.count(S.col(1, D))], S.PARTIAL)
        got, want = run(p2, 2), O.run_plan_to_arrow(S, p2, [t])
        assert _rows(got) == _rows(want), (pos, ln)


def test_project_computed_strings(built):
    """String literals, substring results and CASE / IF over them as OUTPUT columns (TPC-DS: `'store channel' as channel`, bucket labels):
    packed in registers, stored as 16 bytes and expanded to offsets + bytes on the device — through Projection, above a join, and as an
    aggregate's grouping column coming back from a nested context."""
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    n = 40_000
    t = pa.table({"phone": pa.array([None if rng.random() < 0.05 else "%02d-%03d-%04d" % tuple(rng.integers(10, 99, 1).tolist() + rng.integers(100, 999, 1).tolist() + rng.integers(1000, 9999, 1).tolist()) for _ in range(n)]),
                  "k": pa.array(rng.integers(0, 5, n), pa.int32(), mask=rng.random(n) < 0.1), "v": pa.array(rng.integers(0, 1000, n), pa.int64())})
    fields = [S.T_STRING, S.T_INT32, S.T_INT64]
    s, k, v = (S.col(i, ty) for i, ty in enumerate(fields))
    L, I = (lambda x: S.lit(x, S.T_STRING)), (lambda x: S.lit(x, S.T_INT32))
    code = S.scalar_func("substring", [s, I(1), I(2)], S.T_STRING)
    label = S.case_when([(S.eq(k, I(0)), L("store channel")), (S.eq(k, I(1)), L("catalog channel")), (S.lt(k, I(4)), L(""))], L("web channel"))
    plan = S.project(S.filter_(S.scan(fields), S.gt(v, S.lit(100, S.T_INT64))), [v, code, L("constant"), label, S.if_(S.is_null(k), S.lit(None, S.T_STRING), code), s])
    run = lambda p, nc: pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], nc, p.encode(), batch_size=0))
    got, want = run(plan, 6), O.run_plan_to_arrow(S, plan, [t])
    assert got.schema.types == want.schema.types
    for c in range(6):
        assert got.column(c).combine_chunks().equals(want.column(c).combine_chunks()), c
    # grouped by the computed label, then sorted (the aggregate's result stays on the device below the Sort)
    agg = S.hash_agg(S.project(S.scan(fields), [label, code, v]), [S.col(0, S.T_STRING), S.col(1, S.T_STRING)], [S.sum_(S.col(2, S.T_INT64), S.T_INT64)], S.PARTIAL)
    p2 = S.sort(agg, [(S.col(0, S.T_STRING), False, False), (S.col(1, S.T_STRING), True, True)])
    got2, want2 = run(p2, 3), O.run_plan_to_arrow(S, p2, [t])
    assert got2.to_pylist() == want2.to_pylist()
