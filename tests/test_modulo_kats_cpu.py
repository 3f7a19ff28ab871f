"""The reference's own remainder vectors (native/spark-expr/src/math_funcs/modulo_expr.rs:360-985, 25 tests) on the oracle: Int32 and
Decimal128(18,4) basics, a zero divisor → NULL outside ANSI mode and REMAINDER_BY_ZERO (common/src/error.rs:81-82, 684) in it, and the Float64
cases — -0.0 is a zero divisor, a NULL dividend or divisor never raises, NaN / ±Infinity dividends raise with a zero divisor and give NaN
with another, a NaN divisor gives NaN.  tests/test_filter_project_gpu.py runs the same operator on the GPU against this oracle."""
import math

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import serde as S
from oracle import oracle as O

F64, I32 = S.T_DOUBLE, S.T_INT32
NAN, INF = float("nan"), float("inf")


def build(lhs, rhs, mode, lit_divisor=None, lit_dividend=None, dtype=F64, arrow=pa.float64()):
    """→ (plan, table) of `a % b`; a one-element lit_* list makes that operand a literal (the reference's Array/Scalar pairs)"""
    cols, fields = {}, []
    if lit_dividend is None:
        cols["a"] = pa.array(lhs, arrow)
        fields.append(dtype)
        a = S.col(0, dtype)
    else:
        a = S.lit(lit_dividend[0], dtype)
    if lit_divisor is None:
        cols["b"] = pa.array(rhs, arrow)
        b = S.col(len(fields), dtype)
        fields.append(dtype)
    else:
        b = S.lit(lit_divisor[0], dtype)
    return S.project(S.scan(fields), [S.math("remainder", a, b, dtype, mode)]), pa.table(cols)


def _mod(*args, **kw):
    plan, table = build(*args, **kw)
    return O.run_plan_to_arrow(S, plan, table).column(0).to_pylist()


def _raises(*args, **kw):
    with pytest.raises(O.OracleError, match="REMAINDER_BY_ZERO"):
        _mod(*args, **kw)


# (arguments of build, expected values — None = NULL, NaN = any NaN — or "raise"): modulo_expr.rs line numbers beside each
I32KW = dict(dtype=I32, arrow=pa.int32())
CASES = [
    (([3, 2, -2**31], [1, 5, -1], S.ANSI), I32KW, [0, 2, 0]), (([3, 2, -2**31], [1, 5, -1], S.LEGACY), I32KW, [0, 2, 0]),      # :361-392
    (([3], [0], S.LEGACY), I32KW, [None]), (([3], [0], S.ANSI), I32KW, "raise"),                                             # :441-470
    (([1.0], [0.0], S.ANSI), {}, "raise"),                                                       # :545-556
    (([1.0], [-0.0], S.ANSI), {}, "raise"),                                                      # :558-569
    (([1.0], [0.0], S.ANSI), dict(dtype=S.T_FLOAT, arrow=pa.float32()), "raise"),               # :571-581
    (([1.0], [0.0], S.LEGACY), {}, [None]),                                                      # :583-595
    (([None], [0.0], S.ANSI), {}, [None]),                                                       # :597-609
    (([None, 1.0], [0.0, 0.0], S.ANSI), {}, "raise"),                                            # :611-623
    (([None, None], [0.0, 0.0], S.ANSI), {}, [None, None]),                                      # :625-637
    (([1.0, 2.0], None, S.ANSI), dict(lit_divisor=[0.0]), "raise"),                              # :666-676
    (([1.0], None, S.ANSI), dict(lit_divisor=[-0.0]), "raise"),                                  # :678-687
    (([None, None], None, S.ANSI), dict(lit_divisor=[0.0]), [None, None]),                       # :689-700
    (([5.0, None], None, S.LEGACY), dict(lit_divisor=[2.0]), [1.0, None]),                       # :702-712
    (([5.0, None], None, S.ANSI), dict(lit_divisor=[2.0]), [1.0, None]),
    ((None, [2.0, 0.0], S.ANSI), dict(lit_dividend=[1.0]), "raise"),                             # :741-752
    ((None, [0.0], S.ANSI), dict(lit_dividend=[None]), [None]),                                  # :754-764
    (([1.0, 3.0, None, 5.0], [2.0, 0.0, 0.0, 1.5], S.ANSI), {}, "raise"),                        # :902-924
    (([5.0, None, 7.0, None, 9.0], [2.0, 0.0, 4.0, -0.0, 2.0], S.ANSI), {}, [1.0, None, 3.0, None, 1.0]),      # :926-956
    (([1.0, 2.0], [None, 2.0], S.ANSI), {}, [None, 0.0]),                                        # :958-969
    (([NAN, INF, -INF, 0.0, 5.0], [2.0] * 5, S.ANSI), {}, [NAN, NAN, NAN, 0.0, 1.0]),            # :874-900
    (([1.0], [NAN], S.ANSI), {}, [NAN]),                                                         # :971-985
]
for _d in (NAN, INF, -INF, 0.0, -0.0):                                                           # :819-843: only the divisor decides
    CASES += [(([_d], [0.0], S.ANSI), {}, "raise"), (([_d], None, S.ANSI), dict(lit_divisor=[0.0]), "raise")]


def same(got, want):
    return len(got) == len(want) and all((g is None and w is None) or (g is not None and w is not None and ((math.isnan(w) and math.isnan(g)) or g == w))
                                          for g, w in zip(got, want))


def test_every_case_on_the_oracle():
    for args, kw, want in CASES:
        if want == "raise":
            _raises(*args, **kw)
        else:
            assert same(_mod(*args, **kw), want), (args, kw)


def test_integers_and_decimals():
    for mode in (S.ANSI, S.LEGACY):                                      # modulo_expr.rs:361-438
        assert _mod([3, 2, -2**31], [1, 5, -1], mode, dtype=I32, arrow=pa.int32()) == [0, 2, 0]
    from datafusion_comet_amd.tpch import _dec128_array
    D = S.decimal(18, 4)          # (the reference's unscaled values have 19 digits: arrow-rs does not validate them, the buffers are built directly)
    t = pa.table({"a": _dec128_array(np.array([3000000000000000000, 2000000000000000000], np.int64), 18, 4),
                  "b": _dec128_array(np.array([1000000000000000000, 5000000000000000000], np.int64), 18, 4)})
    for mode in (S.ANSI, S.LEGACY):
        got = O.run_plan_to_arrow(S, S.project(S.scan([D, D]), [S.math("remainder", S.col(0, D), S.col(1, D), D, mode)]), t).column(0).combine_chunks()
        assert np.frombuffer(got.buffers()[1], np.int64)[:4:2].tolist() == [0, 2000000000000000000]
    assert _mod([3], [0], S.LEGACY, dtype=I32, arrow=pa.int32()) == [None]      # :441-470
    _raises([3], [0], S.ANSI, dtype=I32, arrow=pa.int32())


def test_float_zero_divisors():
    _raises([1.0], [0.0], S.ANSI)                                        # :545-556
    _raises([1.0], [-0.0], S.ANSI)                                       # :558-569
    _raises([1.0], [0.0], S.ANSI, dtype=S.T_FLOAT, arrow=pa.float32())   # :571-581
    assert _mod([1.0], [0.0], S.LEGACY) == [None]                        # :583-595
    assert _mod([None], [0.0], S.ANSI) == [None]                         # :597-609
    _raises([None, 1.0], [0.0, 0.0], S.ANSI)                             # :611-623
    assert _mod([None, None], [0.0, 0.0], S.ANSI) == [None, None]        # :625-637
    _raises([1.0, 2.0], None, S.ANSI, lit_divisor=[0.0])                 # :666-676
    _raises([1.0], None, S.ANSI, lit_divisor=[-0.0])                     # :678-687
    assert _mod([None, None], None, S.ANSI, lit_divisor=[0.0]) == [None, None]      # :689-700
    assert _mod([5.0, None], None, S.LEGACY, lit_divisor=[2.0]) == [1.0, None]      # :702-712
    assert _mod([5.0, None], None, S.ANSI, lit_divisor=[2.0]) == [1.0, None]
    _raises(None, [2.0, 0.0], S.ANSI, lit_dividend=[1.0])                # :741-752
    assert _mod(None, [0.0], S.ANSI, lit_dividend=[None]) == [None]      # :754-764
    _raises([1.0, 3.0, None, 5.0], [2.0, 0.0, 0.0, 1.5], S.ANSI)         # :902-924
    assert _mod([5.0, None, 7.0, None, 9.0], [2.0, 0.0, 4.0, -0.0, 2.0], S.ANSI) == [1.0, None, 3.0, None, 1.0]      # :926-956
    assert _mod([1.0, 2.0], [None, 2.0], S.ANSI) == [None, 0.0]          # :958-969


def test_float_special_values():
    for dividend in (NAN, INF, -INF, 0.0, -0.0):                         # :819-843: only the divisor decides
        _raises([dividend], [0.0], S.ANSI)
        _raises([dividend], None, S.ANSI, lit_divisor=[0.0])
    got = _mod([NAN, INF, -INF, 0.0, 5.0], [2.0] * 5, S.ANSI)            # :874-900
    assert all(math.isnan(v) for v in got[:3]) and got[3:] == [0.0, 1.0]
    got = _mod([1.0], [NAN], S.ANSI)                                     # :971-985
    assert math.isnan(got[0])
