"""String functions with new bytes and digests as DERIVED Utf8 columns of a chain's source (reverse, repeat, replace, substring_index, md5 / sha1 / sha2:
strfn_kernels.hip over device/strfn.hpp — the source tests/test_strfn_cpu.py runs on the host against hashlib, zlib and Python's str), usable as outputs AND
as operands (a comparison, LIKE, length of any length), and instr / ascii / crc32 in the fused kernel.  Against the oracle."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
STR, I32, I64 = S.T_STRING, S.T_INT32, S.T_INT64
f = S.scalar_func


def _table(n, seed=61):
    rng = np.random.default_rng(seed)
    words = np.array(["", "a", "abc", "www.apache.org", "a.b.c.d.e", "日本語テキスト", "naïve café", "xxyxx", "aaa", "The quick brown fox jumps over the lazy dog", "x" * 200, "55-56 bytes: " + "p" * 43,
                      "a,b,,c", "abcabcabc"], dtype=object)
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


s = S.col(0, STR)
L = lambda v: S.lit(v, STR)


def test_new_bytes_and_digests(built):
    t = _table(30_000)
    b = S.cast(s, S.DataType(S.BYTES))
    got = _check([f("reverse", [s], STR), f("repeat", [s, S.lit(3, I64)], STR), f("replace", [s, L("a"), L("ZZ")], STR), f("replace", [s, L(""), L("-")], STR), f("replace", [s, L("abc")], STR),
                  f("substring_index", [s, L("."), S.lit(2, I64)], STR), f("substring_index", [s, L("."), S.lit(-2, I64)], STR), f("md5", [b], STR), f("sha1", [b], STR)], t)
    assert got.column(7).to_pylist()[0] is None or len(got.column(7).to_pylist()[0]) == 32
    _check([f("sha2", [b, S.lit(224, I32)], STR), f("sha2", [b, S.lit(256, I32)], STR), f("sha2", [b, S.lit(0, I32)], STR), f("sha2", [b, S.lit(384, I32)], STR), f("sha2", [s, S.lit(512, I32)], STR),
            f("crc32", [b], I64), f("instr", [s, L("c")], I32), f("instr", [s, L("語")], I32), f("ascii", [s], I32), s, S.col(1, I32)], t)


def test_as_operands_and_below_a_filter(built):
    t = _table(40_000, 62)
    rev = f("reverse", [s], STR)
    src = S.filter_(S.scan([STR, I32]), S.and_(S.like(rev, L("%a")), S.lt(S.col(1, I32), S.lit(70, I32))))
    got = _check([rev, S.eq(f("replace", [s, L("a"), L("b")], STR), L("bbc")), f("length", [f("repeat", [s, S.lit(2, I64)], STR)], I32), S.gt(f("md5", [s], STR), L("8")), S.col(1, I32)], t, src)
    assert 0 < got.num_rows < t.num_rows
    none = S.filter_(S.scan([STR, I32]), S.lt(S.col(1, I32), S.lit(-1, I32)))
    assert native.execute_to_table([native.HostInput.from_table(t)], 1, S.project(none, [rev]).encode(), batch_size=0) == []


def test_refusals(built):
    t = _table(10)
    up = f("upper", [s], STR)
    for e, why in ((f("reverse", [up], STR), "Utf8 COLUMN"), (f("repeat", [s, S.lit(-1, I64)], STR), "negative"), (f("sha2", [s, S.lit(100, I32)], STR), "bit length"),
                   (f("replace", [s, s, L("x")], STR), "string literal")):
        with pytest.raises(native.CometNativeException, match=why):
            _run(S.project(S.scan([STR, I32]), [e]), t, 1)
