"""split(str, pattern, limit) (string_funcs/split.rs; strings.scala:598-631 sends it under spark.comet.expression.StringSplit.allowIncompatible) as a
list<string> column DERIVED from the source table: two passes of the matcher per row on the device (regex_kernels.hip: the source
tests/test_regexp_extract_cpu.py walks on the host), the pieces assembled from views, the lists passed through the chain by row index — and
exploded.  Against the oracle's restatement (Regex::split / find_iter's rule for empty matches, limit > 0 / = 0 / < 0)."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
STR, I32 = S.T_STRING, S.T_INT32
LS = S.list_type(STR, False)


def _table(n, seed=21):
    rng = np.random.default_rng(seed)
    words = np.array(["", "a,b,c", "a,b,c,,", ",,,", "x", "one, two ,three", "k1=v1;k2=v2;;k3", "日本,語テ,キスト", "naïve  café   au lait", "foo123bar456baz", "2024-06-30", "a" * 50 + "," + "b" * 60,
                      " leading and trailing "], dtype=object)
    return pa.table({"s": pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=rng.random(n) < 0.1), "k": pa.array(rng.integers(0, 100, n), pa.int32())})


def _sp(pattern, *limit):
    return S.scalar_func("split", [S.col(0, STR), S.lit(pattern, STR)] + [S.lit(l, I32) for l in limit], LS)


def _run(plan, table, ncols, **kw):
    return pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(table)], ncols, plan.encode(), batch_size=0, **kw))


def _check(exprs, table, source=None):
    from oracle import oracle as O
    plan = S.project(source if source is not None else S.scan([STR, I32]), exprs)
    got, want = _run(plan, table, len(exprs)), O.run_plan_to_arrow(S, plan, table)
    for i in range(len(exprs)):
        assert got.column(i).to_pylist() == want.column(i).to_pylist(), f"output {i}"
    return got


def test_the_references_vectors(built):
    t = pa.table({"s": pa.array(["foo123bar456baz", "a,b,c,d,e", "a,b,c,,", "", None, "x,y"]), "k": pa.array(np.arange(6, dtype=np.int32))})
    got = _check([_sp(r"\d+"), _sp(",", 3), _sp(",", 0), _sp(",", -1), _sp(",")], t)
    assert got.column(0).to_pylist()[0] == ["foo", "bar", "baz"]
    assert got.column(1).to_pylist()[1] == ["a", "b", "c,d,e"]
    assert got.column(2).to_pylist()[2] == ["a", "b", "c"]
    assert got.column(3).to_pylist()[2:5] == [["a", "b", "c", "", ""], [""], None]


def test_patterns_limits_and_nulls(built):
    t = _table(30_000)
    got = _check([_sp(","), _sp(",", 0), _sp(",", 2), _sp(r"\s*[,;]\s*"), _sp(r"\s+", 0), _sp(""), _sp(r"\b", 4), _sp("[=;]", -1), S.col(0, STR), S.col(1, I32), _sp(",", 1)], t)
    assert got.column(0).null_count == t.column(0).null_count > 0


def test_below_a_filter_and_with_no_rows(built):
    t = _table(40_000, 22)
    src = S.filter_(S.scan([STR, I32]), S.lt(S.col(1, I32), S.lit(20, I32)))
    got = _check([_sp(","), S.col(1, I32), S.scalar_func("regexp_extract", [S.col(0, STR), S.lit(r"(\w+)", STR), S.lit(1, I32)], STR)], t, src)
    assert 0 < got.num_rows < t.num_rows
    none = S.filter_(S.scan([STR, I32]), S.lt(S.col(1, I32), S.lit(-1, I32)))
    assert native.execute_to_table([native.HostInput.from_table(t)], 1, S.project(none, [_sp(",")]).encode(), batch_size=0) == []


def test_explode_of_a_split(built):
    """SELECT k, explode(split(s, ',')) — the plan Spark writes for it: Explode over the Projection that holds the split"""
    from oracle import oracle as O
    t = _table(10_000, 23)
    proj = S.project(S.scan([STR, I32]), [S.col(1, I32), _sp(r"\s*,\s*")])
    plan = S.explode(proj, S.col(1, LS), [S.col(0, I32)], outer=False, position=True)
    got = _run(plan, t, 3)
    exp = []
    for s, k in zip(t.column(0).to_pylist(), t.column(1).to_pylist()):
        if s is not None:
            for pos, piece in enumerate(O.split_like_the_crate(r"\s*,\s*", s, -1)):
                exp.append((k, pos, piece))
    assert list(zip(*[got.column(c).to_pylist() for c in range(3)])) == exp


def test_what_split_refuses(built):
    t = _table(10)
    up = S.scalar_func("upper", [S.col(0, STR)], STR)
    for plan, why in ((S.project(S.scan([STR, I32]), [S.scalar_func("split", [up, S.lit(",", STR)], LS)]), "Utf8 COLUMN"), (S.project(S.scan([STR, I32]), [_sp(r"\p{L}")]), "not supported"),
                      (S.project(S.scan([STR, I32]), [_sp("(a*)*")]), "empty string")):
        with pytest.raises(native.CometNativeException, match=why):
            _run(plan, t, 1)


def _all(pattern, *idx):
    return S.scalar_func("regexp_extract_all", [S.col(0, STR), S.lit(pattern, STR)] + [S.lit(i, I32) for i in idx], S.list_type(STR, True))


def test_regexp_extract_all(built):
    """string_funcs/regexp_extract_all.rs: group idx of every match, through the same derived-column path (the group-0 program drives the iteration, the
    group's program reports)"""
    ref = pa.table({"s": pa.array(["100-200, 300-400", "foo-bar", "nodelim", None, "abc123def456", "foo foo", "1 2 3"]), "k": pa.array(np.arange(7, dtype=np.int32))})
    got = _check([_all(r"(\d+)-(\d+)", 1), _all(r"(\d+)-(\d+)", 2), _all(r"\d+", 0), _all(r"(foo)(bar)?", 2), _all(r"(\d)")], ref)
    assert got.column(0).to_pylist()[:4] == [["100", "300"], [], [], None] and got.column(3).to_pylist()[5] == ["", ""]
    t = _table(20_000, 24)
    _check([_all(r"(\w+)=(\w*)", 2), _all(r"[^,;\s]+", 0), _all(r"(\d+)", 1), _all(r"x*", 0), S.col(1, I32)], t, S.filter_(S.scan([STR, I32]), S.lt(S.col(1, I32), S.lit(60, I32))))
    with pytest.raises(native.CometNativeException, match="Expects group index between 0 and 1, but got 2"):
        _run(S.project(S.scan([STR, I32]), [_all(r"(a)", 2)]), ref, 1)
