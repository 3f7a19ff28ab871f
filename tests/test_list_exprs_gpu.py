"""Expressions over list columns in the fused kernels: size (array_funcs/size.rs), GetArrayItem / element_at (array_funcs/list_extract.rs, ListExtract),
array_contains (datafusion-spark's, Spark's three-valued answer), IS [NOT] NULL of a nested column — over lists that arrive through a Scan and over the
lists split() derives (split(s, ',')[0], element_at(split(...), -1), array_contains(split(...), 'x')).  The chain's source table carries every list's
element column as a column of its own; an element of a list of strings is gathered by the executor.  Against the oracle (Python lists)."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
I32, I64, STR, D = S.T_INT32, S.T_INT64, S.T_STRING, S.T_DATE
LI, LS, LD = S.list_type(I64, True), S.list_type(STR, True), S.list_type(D, True)
f = S.scalar_func


def _table(n, seed=51):
    rng = np.random.default_rng(seed)
    words = ["", "a", "bb", "x", "日本", "a much longer element than fifteen bytes", "k=v"]

    def lists(make, null_elem=0.1):
        out = []
        for _ in range(n):
            r = rng.random()
            if r < 0.08:
                out.append(None)
            else:
                out.append([None if rng.random() < null_elem else make() for _ in range(int(rng.integers(0, 6)))])
        return out
    import datetime
    return pa.table({"k": pa.array(rng.integers(-3, 7, n), pa.int32(), mask=rng.random(n) < 0.05),
                     "li": pa.array(lists(lambda: int(rng.integers(-5, 5))), pa.list_(pa.int64())),
                     "ls": pa.array(lists(lambda: words[int(rng.integers(0, len(words)))]), pa.list_(pa.utf8())),
                     "ld": pa.array(lists(lambda: datetime.date(1970, 1, 1) + datetime.timedelta(days=int(rng.integers(-3, 3)))), pa.list_(pa.date32())),
                     "s": pa.array(np.array(["a,b,c", "x", "", "k=v,x,,", "one,two"], dtype=object)[rng.integers(0, 5, n)], pa.utf8(), mask=rng.random(n) < 0.05)})


TYPES = [I32, LI, LS, LD, STR]


def _run(plan, table, ncols, **kw):
    return pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(table)], ncols, plan.encode(), batch_size=0, **kw))


def _check(exprs, table, source=None):
    from oracle import oracle as O
    plan = S.project(source if source is not None else S.scan(TYPES), exprs)
    got, want = _run(plan, table, len(exprs)), O.run_plan_to_arrow(S, plan, table)
    for i in range(len(exprs)):
        assert got.column(i).to_pylist() == want.column(i).to_pylist(), f"output {i}"
    return got


def test_size_and_nullness(built):
    t = _table(20_000)
    k, li, ls, ld = S.col(0, I32), S.col(1, LI), S.col(2, LS), S.col(3, LD)
    got = _check([f("size", [li], I32), f("size", [ls], I32), S.is_null(li), S.is_not_null(ls), S.case_when([(S.is_not_null(ld), f("size", [ld], I32))], S.lit(None, I32)), k], t)
    assert -1 in got.column(0).to_pylist()


def test_elements_by_position(built):
    t = _table(20_000, 52)
    k, li, ls, ld = S.col(0, I32), S.col(1, LI), S.col(2, LS), S.col(3, LD)
    L = lambda v: S.lit(v, I32)
    _check([S.list_extract(li, L(0)), S.list_extract(li, L(2)), S.list_extract(li, k), S.list_extract(li, L(1), one_based=True), S.list_extract(li, L(-1), one_based=True),
            S.list_extract(ld, L(1)), S.list_extract(ls, L(0)), S.list_extract(ls, L(-2), one_based=True), S.list_extract(ls, k), S.math("add", S.list_extract(li, L(0)), S.lit(1, I64), I64)], t)
    # below a Filter that reads an element
    _check([S.col(0, I32), S.list_extract(ls, L(1))], t, S.filter_(S.scan(TYPES), S.gt(S.list_extract(li, L(0)), S.lit(0, I64))))


def test_array_contains(built):
    t = _table(20_000, 53)
    k, li, ls, ld = S.col(0, I32), S.col(1, LI), S.col(2, LS), S.col(3, LD)
    _check([f("array_contains", [li, S.lit(3, I64)], S.T_BOOL), f("array_contains", [li, S.cast(k, I64)], S.T_BOOL), f("array_contains", [ls, S.lit("x", STR)], S.T_BOOL),
            f("array_contains", [ls, S.lit("a much longer element than fifteen bytes", STR)], S.T_BOOL), f("array_contains", [ld, S.lit(1, D)], S.T_BOOL), f("array_contains", [li, S.lit(None, I64)], S.T_BOOL)], t)


def test_elements_of_a_split(built):
    t = _table(20_000, 54)
    s = S.col(4, STR)
    sp = f("split", [s, S.lit(",", STR), S.lit(-1, I32)], S.list_type(STR, False))
    L = lambda v: S.lit(v, I32)
    _check([S.list_extract(sp, L(0)), S.list_extract(sp, L(-1), one_based=True), S.list_extract(sp, L(5)), f("size", [sp], I32), f("array_contains", [sp, S.lit("x", STR)], S.T_BOOL), sp, s], t)


def test_errors_of_the_reference_and_refusals(built):
    t = _table(200, 55)
    li, ls = S.col(1, LI), S.col(2, LS)
    L = lambda v: S.lit(v, I32)
    with pytest.raises(native.CometQueryExecutionException, match="INVALID_INDEX_OF_ZERO"):
        _run(S.project(S.scan(TYPES), [S.list_extract(li, L(0), one_based=True)]), t, 1)
    with pytest.raises(native.CometQueryExecutionException, match=r'INVALID_ARRAY_INDEX.*"indexValue":9,"arraySize":\d'):
        _run(S.project(S.scan(TYPES), [S.list_extract(li, L(9), fail_on_error=True)]), t, 1)
    with pytest.raises(native.CometQueryExecutionException, match=r'INVALID_ARRAY_INDEX_IN_ELEMENT_AT.*"indexValue":-9'):
        _run(S.project(S.scan(TYPES), [S.list_extract(li, L(-9), one_based=True, fail_on_error=True)]), t, 1)
    for e, why in ((S.gt(S.list_extract(ls, L(0)), S.lit("a", STR)), "OUTPUT column"), (f("array_contains", [li, S.lit(1, I32)], S.T_BOOL), "key's type"),
                   (f("size", [S.col(0, I32)], I32), "list / map COLUMN")):
        with pytest.raises(native.CometNativeException, match=why):
            _run(S.project(S.scan(TYPES), [e]), t, 1)
