"""RLIKE on the GPU (expr.proto rlike = 30 → predicate_funcs/rlike.rs): the pattern's search automaton is compiled at createPlan
(csrc/regex.cpp, pinned against a backtracking engine in tests/test_rlike_cpu.py) and walked per value by the generated kernel — as a
projected Boolean and as a filter predicate, with NULLs, UTF-8 text and values far longer than the packed-string limit; patterns outside the
exactly reproducible subset fail at createPlan by name."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
STR, I32 = S.T_STRING, S.T_INT32


def _table(n, seed=9):
    rng = np.random.default_rng(seed)
    words = np.array(["", "Rose", "Robert", "rose", "a,b", "a,b,c", "日本語テキスト", "café", "line one\nline two", "Customer#000000001", "x" * 90 + "Rz",
                      "special requests: deposits", "1.5", "colour", "color"], dtype=object)
    return pa.table({"s": pa.array(words[rng.integers(0, len(words), n)], pa.utf8(), mask=rng.random(n) < 0.1), "k": pa.array(rng.integers(0, 100, n), pa.int32())})


@pytest.mark.parametrize("pattern", ["R[a-z]+", "^R", "e$", "^[^,]+,[^,]+$", "colou?r", "日本", "é$", "^.{4}$", "(ab|c)+d?", "special.*requests", "^$", "x{50,}R", "1\\.5", "line.two",
                                     # the Perl classes (the crate's Unicode 16 tables; on these words Python's re agrees)
                                     "^\\d\\.\\d$", "^\\w+$", "\\w\\W\\w", "^[\\w#]+\\d{9}$", "^\\D+$",
                                     # word boundaries
                                     "\\bRose\\b", "\\b\\w{5}\\b", "e\\B", "\\bline\\b.*\\btwo\\b"])
def test_projection_and_filter_match_the_oracle(built, pattern):
    from oracle import oracle as O
    t = _table(30_000)
    pred = S.rlike(S.col(0, STR), S.lit(pattern, STR))
    proj = S.project(S.scan([STR, I32]), [pred, S.col(1, I32)])
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 2, proj.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, proj, t)
    assert got.column(0).to_pylist() == want.column(0).to_pylist()
    filt = S.filter_(S.scan([STR, I32]), S.and_(pred, S.lt(S.col(1, I32), S.lit(50, I32))))
    out = native.execute_to_table([native.HostInput.from_table(t)], 2, filt.encode(), batch_size=0)
    wantf = O.run_plan_to_arrow(S, filt, t)
    gotf = pa.Table.from_batches(out) if out else wantf.slice(0, 0)
    assert gotf.num_rows == wantf.num_rows and gotf.column(0).to_pylist() == wantf.column(0).to_pylist()


def test_unsupported_patterns_fail_at_create_plan(built):
    t = _table(10)
    for pattern in ("(?m)a\\b", "a(?i)rose", "\\p{L}+"):
        plan = S.project(S.scan([STR, I32]), [S.rlike(S.col(0, STR), S.lit(pattern, STR))])
        with pytest.raises(native.CometNativeException, match="RLIKE pattern .* is not supported"):
            native.execute_to_table([native.HostInput.from_table(t)], 1, plan.encode())
