"""regexp_extract(subject, pattern, idx) as an output column (string_funcs/regexp_extract.rs; strings.scala:464-495: pattern and idx are literals):
the generated kernel runs comet_regex_vm.hpp's matcher — the source tests/test_regexp_extract_cpu.py walks on the host against the reference's
vectors and a backtracking engine — per row and describes the group's span as a view of the source value; the executor assembles the column.
Against the oracle (Python's `re` standing in for the crate on the syntax both read alike)."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
STR, I32 = S.T_STRING, S.T_INT32


def _table(n, seed=11):
    rng = np.random.default_rng(seed)
    words = np.array(["", "100-200", "foo-bar", "nodelim", "abc123def456", "order 66, item 7", "joe@site.com", "ann.lee@mail.example.org", "日本語 2024-06-30 テキスト",
                      "naïve café №٣٤", "<a><b>", "x" * 40 + "9-8" + "y" * 40, "key=value;other=thing", "  padded  ", "2023-11-05T08:09:10"], dtype=object)
    return pa.table({"s": pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=rng.random(n) < 0.1), "k": pa.array(rng.integers(0, 100, n), pa.int32())})


def _run(plan, table, ncols, **kw):
    return pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(table)], ncols, plan.encode(), batch_size=0, **kw))


def _check(exprs, table, source=None):
    from oracle import oracle as O
    plan = S.project(source if source is not None else S.scan([STR, I32]), exprs)
    got, want = _run(plan, table, len(exprs)), O.run_plan_to_arrow(S, plan, table)
    for i in range(len(exprs)):
        assert got.column(i).to_pylist() == want.column(i).to_pylist(), f"output {i}"
    return got


def _rx(pattern, idx=None):
    args = [S.col(0, STR), S.lit(pattern, STR)] + ([S.lit(idx, I32)] if idx is not None else [])
    return S.scalar_func("regexp_extract", args, STR)


def test_the_references_vectors(built):
    t = pa.table({"s": pa.array(["100-200", "foo-bar", "nodelim", None, "abc123def456", "foo", "a1b", "c2d"]), "k": pa.array(np.arange(8, dtype=np.int32))})
    got = _check([_rx(r"(\d+)-(\d+)", 1), _rx(r"\d+", 0), _rx(r"(\d+)-(\d+)"), _rx(r"(foo)(bar)?", 2), _rx(r"(\d)", 1)], t)
    assert got.column(0).to_pylist()[:4] == ["100", "", "", None]
    assert got.column(1).to_pylist()[4] == "123"
    assert got.column(3).to_pylist()[5] == ""
    assert got.column(4).to_pylist()[6:] == ["1", "2"]


PATTERNS = [(r"(\d+)-(\d+)", 2), (r"(\w+)@(\w+)\.", 2), (r"<(.+?)>", 1), (r"<(.+)>", 1), (r"(\d{4})-(\d{2})-(\d{2})", 0), (r"\b(\w+)$", 1), (r"(?i)(ORDER|ITEM)\s+(\d+)", 2),
            (r"^\s*(\S+)", 1), (r"([^=;]+)=([^;]*)", 2), (r"(a|ab)(c|bcd)?(\d*)", 3), (r"(\d+)", 1), (r"(é|ï)", 1)]


def test_groups_classes_and_preferences(built):
    t = _table(30_000)
    _check([_rx(p, i) for p, i in PATTERNS] + [S.col(0, STR), S.col(1, I32)], t)


def test_below_a_filter_with_nulls_and_no_rows(built):
    t = _table(40_000, 12)
    src = S.filter_(S.scan([STR, I32]), S.lt(S.col(1, I32), S.lit(25, I32)))
    got = _check([_rx(r"(\d+)", 1), S.col(1, I32), _rx(r"(é|ï)", 1)], t, src)
    assert 0 < got.num_rows < t.num_rows
    none = S.filter_(S.scan([STR, I32]), S.lt(S.col(1, I32), S.lit(-1, I32)))
    assert native.execute_to_table([native.HostInput.from_table(t)], 1, S.project(none, [_rx(r"(\d+)", 1)]).encode(), batch_size=0) == []


def test_null_pattern_or_index_is_null_everywhere(built):
    t = _table(1000, 13)
    nul = lambda ty: S.lit(None, ty)
    got = _check([S.scalar_func("regexp_extract", [S.col(0, STR), nul(STR), S.lit(1, I32)], STR), S.scalar_func("regexp_extract", [S.col(0, STR), S.lit("(a)", STR), nul(I32)], STR)], t)
    assert got.column(0).null_count == 1000 and got.column(1).null_count == 1000


def test_errors_of_the_reference_and_refusals(built):
    t = _table(10)
    for pat, idx, why in ((r"(a)(b)", 3, "Expects group index between 0 and 2, but got 3"), (r"(a)", -1, "but got -1"), (r"(unclosed", 0, "unclosed group"),
                          (r"(a*)*", 1, "empty string"), (r"\p{Greek}+", 0, "not supported")):
        with pytest.raises(native.CometNativeException, match=why):
            _run(S.project(S.scan([STR, I32]), [_rx(pat, idx)]), t, 1)
