"""String functions whose result is a slice of a Utf8 column plus padding — substring, trim / ltrim / rtrim, rpad / lpad and the read-side padding
of CHAR(n) columns (static_invoke/char_varchar_utils/read_side_padding.rs) — as OUTPUT columns of any length: the projection's
kernel describes each result as (source row, byte slice, pad characters) and the executor assembles the column.  Against the oracle, whose
restatements are pinned on Spark's documented answers and the reference's own cases (tests/test_oracle_semantics_cpu.py)."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
STR, I32 = S.T_STRING, S.T_INT32


def _table(n, seed=4):
    rng = np.random.default_rng(seed)
    words = np.array(["", " ", "hi", "  padded  ", "日本語テキスト", "naïve café ☕", "Customer#000000001", "x" * 70, "a much longer value that never fitted fifteen bytes"], dtype=object)
    return pa.table({"s": pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=rng.random(n) < 0.1),
                     "k": pa.array(rng.integers(0, 100, n), pa.int32())})


def _run(plan, table, ncols, **kw):
    return pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(table)], ncols, plan.encode(), batch_size=0, **kw))


def _check(exprs, table, source=None):
    from oracle import oracle as O
    plan = S.project(source if source is not None else S.scan([STR, I32]), exprs)
    got, want = _run(plan, table, len(exprs)), O.run_plan_to_arrow(S, plan, table)
    for i in range(len(exprs)):
        assert got.column(i).to_pylist() == want.column(i).to_pylist(), f"output {i}"
    return got


def test_results_of_any_length(built):
    t = _table(20_000)
    s, I, L = S.col(0, STR), lambda v: S.lit(v, I32), lambda v: S.lit(v, STR)
    f = lambda name, *a: S.scalar_func(name, [s] + list(a), STR)
    got = _check([f("substring", I(3)), f("substring", I(-20), I(30)), f("substring", I(2), I(40)), f("trim"), f("ltrim"), f("rtrim"),
                  f("rpad", I(25)), f("rpad", I(3), L("ab")), f("lpad", I(80), L("é☕")), f("read_side_padding", I(20)), f("read_side_padding", I(0)), s, S.col(1, I32)], t)
    assert max(len(v) for v in got.column(8).to_pylist() if v is not None) == 80


def test_below_a_filter_and_with_no_rows(built):
    t = _table(50_000, 5)
    s = S.col(0, STR)
    src = S.filter_(S.scan([STR, I32]), S.lt(S.col(1, I32), S.lit(30, I32)))
    exprs = [S.scalar_func("read_side_padding", [s, S.lit(32, I32)], STR), S.scalar_func("substring", [s, S.lit(5, I32), S.lit(1000, I32)], STR), S.col(1, I32)]
    got = _check(exprs, t, src)
    assert 0 < got.num_rows < t.num_rows
    none = S.filter_(S.scan([STR, I32]), S.lt(S.col(1, I32), S.lit(-1, I32)))
    assert native.execute_to_table([native.HostInput.from_table(t)], 3, S.project(none, exprs).encode(), batch_size=0) == []
    # CHAR(n) read-side padding of a column that already has its length: the TPC-H flag columns
    flags = pa.table({"s": pa.array(np.array(["A", "N", "R"], dtype=object)[np.random.default_rng(1).integers(0, 3, 10_000)]), "k": pa.array(np.zeros(10_000, np.int32))})
    _check([S.scalar_func("read_side_padding", [s, S.lit(1, I32)], STR)], flags)


def test_hbm_resident_input_and_sliced_tables(built):
    """device-resident source (the uniform-length shortcut of the Utf8 accessors must not leak into the offset-based view functions)"""
    from oracle import oracle as O
    t = pa.table({"s": pa.array(["ab", "cd", "ef", "gh"] * 5000), "k": pa.array(np.arange(20_000, dtype=np.int32))})
    plan = S.project(S.scan([STR, I32]), [S.scalar_func("rpad", [S.col(0, STR), S.lit(5, I32), S.lit("*", STR)], STR), S.col(1, I32)])
    dt = native.DeviceTable.from_arrow(t)
    got = native.execute_to_device([native.DeviceInput(dt)], 2, plan.encode()).to_arrow()
    want = O.run_plan_to_arrow(S, plan, t)
    assert got.column(0).to_pylist() == want.column(0).to_pylist()


def test_long_pad_strings_are_refused(built):
    t = _table(10)
    plan = S.project(S.scan([STR, I32]), [S.scalar_func("rpad", [S.col(0, STR), S.lit(200, I32), S.lit("0123456789" * 4, STR)], STR)])
    with pytest.raises(native.CometNativeException, match="32 characters"):
        _run(plan, t, 1)


def test_concat_of_columns_and_literals(built):
    """Spark's Concat (datafusion-spark's SparkConcat, jni_api.rs:70): the arguments' bytes one after the other, NULL as soon as one argument is
    NULL — Utf8 columns of any length and literals, as an output column; below a filter; with no surviving row; over a device-resident table."""
    from oracle import oracle as O
    rng = np.random.default_rng(8)
    n = 30_000
    words = np.array(["", "a", "日本語", "naïve café ☕", "Customer#000000001", "x" * 70, "a much longer value that never fitted fifteen bytes"], dtype=object)
    t = pa.table({"s": pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=rng.random(n) < 0.1), "k": pa.array(rng.integers(0, 100, n), pa.int32()),
                  "u": pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=rng.random(n) < 0.05)})
    fields = [STR, I32, STR]
    s, u, L = S.col(0, STR), S.col(2, STR), lambda v: S.lit(v, STR)
    cc = lambda *a: S.scalar_func("concat", list(a), STR)
    exprs = [cc(s, u), cc(s, L("-"), u, L(" ☕ "), s), cc(L("id: "), u), cc(s), cc(u, u, u, u, u, u, u, u), S.col(1, I32)]
    plan = S.project(S.scan(fields), exprs)
    got, want = _run(plan, t, len(exprs)), O.run_plan_to_arrow(S, plan, t)
    for i in range(len(exprs)):
        assert got.column(i).to_pylist() == want.column(i).to_pylist(), f"output {i}"
    assert got.column(0).null_count > t.column(0).null_count
    src = S.filter_(S.scan(fields), S.lt(S.col(1, I32), S.lit(30, I32)))
    plan = S.project(src, exprs[:3])
    got, want = _run(plan, t, 3), O.run_plan_to_arrow(S, plan, t)
    assert 0 < got.num_rows < n and all(got.column(i).to_pylist() == want.column(i).to_pylist() for i in range(3))
    none = S.filter_(S.scan(fields), S.lt(S.col(1, I32), S.lit(-1, I32)))
    assert native.execute_to_table([native.HostInput.from_table(t)], 3, S.project(none, exprs[:3]).encode(), batch_size=0) == []
    dt = native.DeviceTable.from_arrow(pa.table({"s": pa.array(["ab", "cd", "ef", "gh"] * 5000), "k": pa.array(np.arange(20_000, dtype=np.int32)), "u": pa.array(["xy"] * 20_000)}))
    plan = S.project(S.scan(fields), [cc(s, L("/"), u)])
    got = native.execute_to_device([native.DeviceInput(dt)], 1, plan.encode()).to_arrow()
    assert got.column(0).to_pylist() == ["ab/xy", "cd/xy", "ef/xy", "gh/xy"] * 5000
    with pytest.raises(native.CometNativeException, match="eight"):
        native.compile_plan(S.project(S.scan(fields), [cc(*([s] * 9))]).encode())


def test_upper_and_lower(built):
    """DataFusion's upper / lower = Rust's str::to_uppercase / to_lowercase (the reference's Upper / Lower under
    spark.comet.caseConversion.enabled): full case mapping — ß → SS, İ → i̇, ﬁ → FI — and the Final_Sigma rule, results longer and shorter than
    their sources; the tables are Rust's own (tests/test_case_map_cpu.py), the oracle here is Python's str.upper() / str.lower()."""
    from oracle import oracle as O
    rng = np.random.default_rng(12)
    n = 30_000
    words = np.array(["", "Hello World", "straße", "Straße", "ΟΔΥΣΣΕΥΣ", "ΑΣ.", "Σ", "İstanbul", "ıI", "ǅemal", "ﬁnal ﬂight", "ŉ", "日本語テキスト", "naïve café ☕", "tschüß",
                      "Customer#000000001", "x" * 70 + "ß", "MiXeD 123 _-", "ǰ ΐ ᾳ ᾼ", "😀 emoji"], dtype=object)
    t = pa.table({"s": pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=rng.random(n) < 0.1), "k": pa.array(rng.integers(0, 100, n), pa.int32())})
    s = S.col(0, STR)
    exprs = [S.scalar_func("upper", [s], STR), S.scalar_func("lower", [s], STR), s, S.col(1, I32)]
    got = _check(exprs, t)
    assert "STRASSE" in got.column(0).to_pylist() and "οδυσσευς" in got.column(1).to_pylist()
    _check(exprs[:2], t, S.filter_(S.scan([STR, I32]), S.lt(S.col(1, I32), S.lit(20, I32))))
