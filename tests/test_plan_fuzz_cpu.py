"""Corrupted plan bytes on the CPU: whatever bytes of a serialized Operator tree are overwritten, dropped or appended, the hand-written proto3
decoder (csrc/proto.cpp) and the planner behind it (compile_plan: decode → schema inference → code generation, no GPU) answer with an error
or with a plan, never with a crash, a hang or unbounded memory — createPlan receives these bytes across the JNI boundary."""
import random

import pytest

from datafusion_comet_amd import native


def _corpus():
    import test_proto_wire_cpu as W
    plans = W.corpus()
    return {k: plans[k].encode() for k in ("tpch_0", "tpch_q3_single", "tpcds_q95_a", "sort_limit", "expand", "window", "window_range_offsets", "window_first_last_nth", "joins",
                                           "literals", "shuffle_writer_range", "native_scan")}


@pytest.mark.parametrize("name", ["tpch_0", "tpch_q3_single", "tpcds_q95_a", "sort_limit", "expand", "window", "window_range_offsets", "window_first_last_nth", "joins", "literals",
                                  "shuffle_writer_range", "native_scan"])
def test_corrupted_plans_fail_cleanly(built, name):
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    good = _corpus()[name]
    rng = random.Random(sum(name.encode()))
    outcomes = {"ok": 0, "error": 0}
    for trial in range(250):
        b = bytearray(good)
        kind = trial % 5
        if kind == 0:                                   # truncated
            b = b[: rng.randrange(0, len(b))]
        elif kind == 1:                                 # a run of bytes dropped
            a = rng.randrange(0, len(b))
            del b[a: a + rng.randrange(1, 6)]
        elif kind == 2:                                 # garbage appended
            b += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 12)))
        else:                                           # bytes overwritten (varint lengths, tags, payloads)
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(0, len(b))] = rng.choice([0x00, 0x7F, 0x80, 0xFF, rng.randrange(256)])
        try:
            native.compile_plan(bytes(b))
            outcomes["ok"] += 1
        except native.CometNativeException:
            outcomes["error"] += 1
    assert outcomes["error"] > 0, outcomes


def test_nesting_is_bounded_like_prost(built):
    """prost, which decodes these bytes in the reference, refuses messages nested more than 100 levels ("recursion limit reached"); the
    hand-written decoder applies the same bound, so a plan of 100 000 nested NOTs is an error, not a stack overflow"""
    from datafusion_comet_amd import serde as S

    def varint(n):
        out = b""
        while True:
            b, n = n & 0x7F, n >> 7
            out += bytes([b | 0x80]) if n else bytes([b])
            if not n:
                return out
    fmsg = lambda f, payload: varint((f << 3) | 2) + varint(len(payload)) + payload
    scan = S.scan([S.T_BOOL]).encode()
    for depth, ok in ((10, True), (45, True), (60, False), (100_000, False)):
        e = S.lit(True, S.T_BOOL).encode()
        for _ in range(depth):
            e = fmsg(40, fmsg(1, e))                       # Expr{not = 40: UnaryExpr{child = 1: Expr}}: two levels per NOT
        plan = fmsg(1, scan) + fmsg(102, fmsg(1, e))       # Operator{children = 1, filter = 102: Filter{predicate = 1}}
        if ok:
            assert native.compile_plan(plan)
        else:
            with pytest.raises(native.CometNativeException, match="recursion limit reached"):
                native.compile_plan(plan)
