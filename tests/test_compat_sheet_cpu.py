"""COMPAT.md — which Spark operators / expressions libcomet.so accepts and the Comet configuration for the rest — is generated
(tools/compat_sheet.py) and probed: every expression class the sheet lists as accepted has a probe plan that comet_check_plan accepts
(decode + plan + generate, no compilation, no GPU); function names behind a sample of the classes it tells the JVM to keep are refused BY
NAME; and the committed file equals the generator's output."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import compat_sheet as C                                        # noqa: E402
from datafusion_comet_amd import native, serde as S             # noqa: E402


def test_every_accepted_expression_class_is_accepted_by_check_plan(built):
    P = C.probes()
    listed = {n for names in C.REFERENCE_EXPRESSIONS.values() for n in names.split()}
    assert set(P) <= listed, set(P) - listed                    # the sheet only speaks about classes the reference serializes
    for cls, (expr, _) in P.items():
        ok, msg = native.check_plan(C.probe_plan(expr).encode())
        assert ok, (cls, msg)


@pytest.mark.parametrize("cls,func,args", [("Hex", "hex", "i"), ("InitCap", "initcap", "s"), ("StringTranslate", "translate", "s"), ("Chr", "char", "i"), ("StringSpace", "space", "i"),
                                           ("SoundEx", "soundex", "s"), ("Levenshtein", "levenshtein", "s")])
def test_functions_the_sheet_disables_are_refused_by_name(built, cls, func, args):
    assert cls not in C.probes()
    col = {"s": S.col(4, S.T_STRING), "f": S.col(2, S.T_DOUBLE), "i": S.col(1, S.T_INT64)}
    rt = S.T_STRING if args[0] == "s" else S.T_DOUBLE
    ok, msg = native.check_plan(C.probe_plan(S.scalar_func(func, [col[a] for a in args], rt)).encode())
    assert not ok and func in msg, msg
    assert f"spark.comet.expression.{cls}.enabled=false" in C.render()


def test_refusals_below_the_class_level_and_unsupported_operators(built):
    # Cast is ONE class: most casts run, timestamp → double is refused at createPlan — what comet_check_plan exists for
    ok, _ = native.check_plan(C.probe_plan(S.cast(S.col(0, S.T_INT32), S.T_DOUBLE)).encode())
    assert ok
    ok, _ = native.check_plan(C.probe_plan(S.cast(S.col(0, S.T_INT32), S.T_STRING)).encode())
    assert ok
    ok, _ = native.check_plan(C.probe_plan(S.cast(S.col(4, S.T_STRING), S.T_TIMESTAMP)).encode())
    assert ok
    ok, _ = native.check_plan(C.probe_plan(S.cast(S.col(2, S.T_DOUBLE), S.T_TIMESTAMP)).encode())
    assert ok
    ok, msg = native.check_plan(C.probe_plan(S.cast(S.cast(S.col(2, S.T_DOUBLE), S.T_TIMESTAMP), S.T_DOUBLE)).encode())
    assert not ok and "Cast" in msg
    # an operator the engine does not run (ParquetWriter = 113) is refused by name; Explode (114) runs since round 5 — over a list COLUMN:
    # without its child expression, and over a column that is not a list, it is refused with the reason
    plan = S.project(S.scan([S.T_INT32]), [S.col(0, S.T_INT32)]).encode()
    ok, msg = native.check_plan(S._f_msg(1, plan) + S._f_msg(113, b""))
    assert not ok and "ParquetWriter" in msg
    ok, msg = native.check_plan(S._f_msg(1, plan) + S._f_msg(114, b""))
    assert not ok and "Explode" in msg
    ok, msg = native.check_plan(S.explode(S.scan([S.T_INT32]), S.col(0, S.T_INT32)).encode())
    assert not ok and "not a list" in msg
    ok, _ = native.check_plan(S.explode(S.scan([S.T_INT32, S.list_type(S.T_STRING)]), S.col(1, S.list_type(S.T_STRING)), [S.col(0, S.T_INT32)], outer=True, position=True).encode())
    assert ok
    text = C.render()
    assert "--conf spark.comet.exec.explode.enabled=false" not in text and "--conf spark.comet.exec.sample.enabled=false" in text


def test_the_committed_sheet_is_the_generators_output(built):
    with open(os.path.join(ROOT, "COMPAT.md")) as f:
        assert f.read() == C.render(), "COMPAT.md is stale: python tools/compat_sheet.py --write"
