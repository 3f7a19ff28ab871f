"""Pins the CPU oracle (oracle/comet_oracle.c + oracle/oracle.py) against every known-answer vector the
reference's own tests hold for the hot path — tests/golden/reference_kats.json — and
cross-checks the C code against an independent exact-integer Python restatement on random inputs."""
import ctypes
import json
import os
import random
from decimal import Decimal

import numpy as np
import pytest

from datafusion_comet_amd import serde as S
from oracle import oracle as O
from oracle import pyint

K = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
C = O.C


def _hash_with_nulls(fn, values, expected):
    """test_hashes_with_nulls! (hash_funcs/utils.rs:1021-1048): plain, then with NULLs inserted at 0 and len/2
    whose hash must stay the seed 42."""
    got = fn(values, None)
    assert got == expected
    n = len(values)
    vals, exp = list(values), list(expected)
    vals.insert(0, None)
    vals.insert(n // 2, None)
    exp.insert(0, 42)
    exp.insert(n // 2, 42)
    assert fn(vals, [v is not None for v in vals]) == exp


def _mm3(kind):
    def run(values, valid):
        n = len(values)
        h = np.full(n, 42, np.uint32)
        vb = None if valid is None else np.array(valid, np.uint8)
        filled = [0 if v is None else v for v in values]
        if kind in ("i8", "i32"):
            a = np.array(filled, np.int32)
            C.o_murmur3_i32(O._p(a), O._p(vb), ctypes.c_int64(n), O._p(h))
        elif kind == "i64":
            a = np.array(filled, np.int64)
            C.o_murmur3_i64(O._p(a), O._p(vb), ctypes.c_int64(n), O._p(h))
        elif kind == "f32":
            a = np.array(filled, np.float32)
            C.o_murmur3_f32(O._p(a), O._p(vb), ctypes.c_int64(n), O._p(h))
        elif kind == "f64":
            a = np.array(filled, np.float64)
            C.o_murmur3_f64(O._p(a), O._p(vb), ctypes.c_int64(n), O._p(h))
        elif kind == "str":
            enc = [("" if v is None else v).encode() for v in values]
            offs = np.zeros(n + 1, np.int32)
            offs[1:] = np.cumsum([len(e) for e in enc])
            data = np.frombuffer(b"".join(enc) + b"\0", np.uint8).copy()
            C.o_murmur3_utf8(O._p(offs), O._p(data), O._p(vb), ctypes.c_int64(n), O._p(h))
        return [int(x) for x in h]
    return run


@pytest.mark.parametrize("kind", ["i8", "i32", "i64", "f32", "f64", "str"])
def test_murmur3_reference_kats(kind):
    k = K["murmur3"][kind]
    _hash_with_nulls(_mm3(kind), k["values"], k["expected"])


def test_pmod_reference_kat():
    k = K["pmod"]
    C.o_pmod.restype = ctypes.c_int32
    assert [C.o_pmod(ctypes.c_uint32(h), k["n"]) for h in k["hashes"]] == k["expected"]


def _wide(op, l, s1, r, s2, p_out, s_out):
    n = len(l)
    out = np.zeros(n, O.DEC128)
    ok = np.zeros(n, np.uint8)
    C.o_wide_decimal({"add": 0, "subtract": 1, "multiply": 2}[op], O._p(O.ints_to_dec(l)), s1, O._p(O.ints_to_dec(r)), s2, p_out, s_out,
                     O._p(out), O._p(ok), ctypes.c_int64(n))
    return [O.dec_to_int(out, i) if ok[i] else None for i in range(n)]   # dec_to_int is signed (hi limb is int64)


def _signed(a, i):
    v = O.dec_to_int(a, i)
    return v


@pytest.mark.parametrize("case", K["wide_decimal"], ids=lambda c: f"{c['op']}_{c['l'][0]}")
def test_wide_decimal_reference_kats(case):
    got = _wide(case["op"], case["l"], case["s1"], case["r"], case["s2"], case["p_out"], case["s_out"])
    assert got == case["expected"]
    # independent exact-int restatement agrees too
    assert [pyint.wide_decimal(case["op"], a, case["s1"], b, case["s2"], case["p_out"], case["s_out"])
            for a, b in zip(case["l"], case["r"])] == case["expected"]


@pytest.mark.parametrize("case", K["check_overflow"], ids=lambda c: str(c["values"]))
def test_check_overflow_reference_kats(case):
    vals = [0 if v is None else v for v in case["values"]]
    ok = np.zeros(len(vals), np.uint8)
    C.o_check_overflow(O._p(O.ints_to_dec(vals)), case["p"], O._p(ok), ctypes.c_int64(len(vals)))
    got = [v if (v is not None and ok[i]) else None for i, v in enumerate(case["values"])]
    assert got == case["expected"]


@pytest.mark.parametrize("case", K["rescale_check"], ids=lambda c: str(c["values"]))
def test_rescale_check_reference_kats(case):
    vals = [0 if v is None else v for v in case["values"]]
    n = len(vals)
    out = np.zeros(n, O.DEC128)
    ok = np.zeros(n, np.uint8)
    C.o_rescale_check(O._p(O.ints_to_dec(vals)), case["s_in"], case["p_out"], case["s_out"], O._p(out), O._p(ok), ctypes.c_int64(n))
    got = [(O.dec_to_int(out, i) if ok[i] else None) if v is not None else None for i, v in enumerate(case["values"])]
    assert got == case["expected"]
    assert [None if v is None else pyint.rescale_check(v, case["s_in"], case["p_out"], case["s_out"]) for v in case["values"]] == case["expected"]


def test_sum_decimal_reference_kats():
    k = K["sum_decimal"]
    for name in ("update_with_filter", "update_filter_null_excluded"):
        c = k[name]
        n = len(c["values"])
        st = (O.SumDecState * 1)()
        C.o_sumdec_init(ctypes.byref(st[0]))
        valid = np.array([1 if f else 0 for f in c["filter"]], np.uint8)   # NULL filter excludes the row
        C.o_sumdec_update_groups(st, O._p(O.ints_to_dec(c["values"])), O._p(valid), O._p(np.zeros(n, np.int64)), ctypes.c_int64(n),
                                 c["precision"], 0)
        out = (ctypes.c_uint64 * 2)()
        assert C.o_sumdec_evaluate(ctypes.byref(st[0]), c["precision"], out) == 1
        assert O._limbs_to_int(out) == c["expected"]
    c = k["merge_multi_row"]
    st = O.SumDecState()
    C.o_sumdec_init(ctypes.byref(st))
    for s, e in zip(c["sums"], c["is_empty"]):
        assert C.o_sumdec_merge(ctypes.byref(st), ctypes.byref(O._i128(s or 0)), int(s is not None), int(e), c["precision"], 0) == 0
    out = (ctypes.c_uint64 * 2)()
    assert C.o_sumdec_evaluate(ctypes.byref(st), c["precision"], out) == 1
    assert O._limbs_to_int(out) == c["expected"]


def test_sum_int_reference_kats():
    k = K["sum_int"]
    for name in ("legacy_filter", "legacy_filter_null", "no_filter"):
        c = k[name]
        n = len(c["values"])
        sums, has = np.zeros(1, np.int64), np.zeros(1, np.uint8)
        valid = None if "filter" not in c else np.array([1 if f else 0 for f in c["filter"]], np.uint8)
        C.o_sumint_update_groups(O._p(sums), O._p(has), O._p(np.array(c["values"], np.int64)), O._p(valid), O._p(np.zeros(n, np.int64)),
                                 ctypes.c_int64(n))
        assert has[0] == 1 and sums[0] == c["expected"]
    c = k["merge_multi_row"]
    sums, has = np.zeros(1, np.int64), np.zeros(1, np.uint8)
    vals = np.array([0 if v is None else v for v in c["states"]], np.int64)
    valid = np.array([v is not None for v in c["states"]], np.uint8)
    C.o_sumint_update_groups(O._p(sums), O._p(has), O._p(vals), O._p(valid), O._p(np.zeros(len(vals), np.int64)), ctypes.c_int64(len(vals)))
    assert sums[0] == c["expected"]


def test_avg_decimal_pinned_by_golden_tpch_q1():
    for c in K["avg_decimal_golden_q1"]:
        s = int(Decimal(c["sum"]).scaleb(2))
        out = (ctypes.c_uint64 * 2)()
        st = O.AvgDecState()
        C.o_avgdec_init(ctypes.byref(st))
        assert C.o_avgdec_merge(ctypes.byref(st), ctypes.byref(O._i128(s)), 1, ctypes.c_int64(c["count"]), 1, 22, 0) == 0
        assert C.o_avgdec_evaluate(ctypes.byref(st), 16, 6, 2, out) == 1
        assert O._limbs_to_int(out) == int(Decimal(c["avg"]).scaleb(6))
        assert pyint.avg_decimal(s, c["count"], 16, 6, 2) == int(Decimal(c["avg"]).scaleb(6))


def test_reference_planner_filter_case_through_the_oracle():
    import pyarrow as pa
    k = K["planner_filter_case"]
    t = pa.table({"c": pa.array([i % k["modulus"] for i in range(k["rows"])], pa.int32())})
    plan = S.filter_(S.scan([S.T_INT32]), S.eq(S.col(0, S.T_INT32), S.lit(k["equals"], S.T_INT32)))
    assert O.run_plan_to_arrow(S, plan, t).num_rows == k["expected_rows"]


def test_c_oracle_matches_exact_python_ints_on_random_inputs():
    rnd = random.Random(7)
    for _ in range(300):
        op = rnd.choice(["add", "subtract", "multiply"])
        s1, s2 = rnd.randint(0, 12), rnd.randint(0, 12)
        p_out = rnd.randint(20, 38)
        s_out = rnd.randint(0, min(p_out, 14))
        mag = rnd.choice([10**6, 10**18, 10**27, 10**37])
        l = [rnd.randint(-mag, mag) for _ in range(8)]
        r = [rnd.randint(-mag, mag) for _ in range(8)]
        want = [pyint.wide_decimal(op, a, s1, b, s2, p_out, s_out) for a, b in zip(l, r)]
        assert _wide(op, l, s1, r, s2, p_out, s_out) == want, (op, s1, s2, p_out, s_out, l, r)
    for _ in range(300):
        s_in, s_out, p_out = rnd.randint(0, 12), rnd.randint(0, 12), rnd.randint(1, 38)
        vals = [rnd.randint(-10**rnd.randint(1, 37), 10**rnd.randint(1, 37)) for _ in range(8)]
        n = len(vals)
        out, ok = np.zeros(n, O.DEC128), np.zeros(n, np.uint8)
        C.o_rescale_check(O._p(O.ints_to_dec(vals)), s_in, p_out, s_out, O._p(out), O._p(ok), ctypes.c_int64(n))
        got = []
        for i in range(n):
            got.append(O.dec_to_int(out, i) if ok[i] else None)
        assert got == [pyint.rescale_check(v, s_in, p_out, s_out) for v in vals]
    for _ in range(300):
        s = rnd.randint(-10**rnd.randint(1, 30), 10**rnd.randint(1, 30))
        cnt = rnd.randint(1, 10**9)
        out = (ctypes.c_uint64 * 2)()
        st = O.AvgDecState()
        C.o_avgdec_init(ctypes.byref(st))
        C.o_avgdec_merge(ctypes.byref(st), ctypes.byref(O._i128(s)), 1, ctypes.c_int64(cnt), 1, 38, 0)
        has = C.o_avgdec_evaluate(ctypes.byref(st), 38, 6, 2, out)
        assert (O._limbs_to_int(out) if has else None) == pyint.avg_decimal(s, cnt, 38, 6, 2)


def test_oracle_q6_agrees_with_pyarrow_compute():
    """Independent engine check: pyarrow's decimal multiply/sum over the same filter."""
    import datetime
    import pyarrow as pa
    import pyarrow.compute as pc
    from datafusion_comet_amd import tpch
    t = tpch.lineitem_q6(100_000, seed=3)
    out = O.run_plan_to_arrow(S, tpch.q6_plan(), t)
    d = lambda x: pa.scalar(Decimal(x), pa.decimal128(12, 2))
    m = pc.and_(pc.and_(pc.greater_equal(t["l_shipdate"], datetime.date(1994, 1, 1)), pc.less(t["l_shipdate"], datetime.date(1995, 1, 1))),
                pc.and_(pc.and_(pc.greater_equal(t["l_discount"], d("0.05")), pc.less_equal(t["l_discount"], d("0.07"))),
                        pc.less(t["l_quantity"], d("24.00"))))
    f = t.filter(m)
    assert out.column(0)[0].as_py() == pc.sum(pc.multiply(f["l_extendedprice"], f["l_discount"])).as_py()


def test_partition_starts_and_indices_reference_example():
    # multi_partition.rs:78-84
    k = K["partition_indices"]
    starts, idx = O.partition_starts_and_indices(np.array(k["partition_ids"]), k["num_partitions"])
    assert starts.tolist() == k["partition_starts"]
    assert idx.tolist() == k["partition_row_indices"]
    # ≡ a stable sort by partition id
    rng = np.random.default_rng(5)
    pids = rng.integers(0, 13, 10_000)
    starts, idx = O.partition_starts_and_indices(pids, 13)
    assert np.array_equal(idx, np.argsort(pids, kind="stable"))
    assert np.array_equal(starts, np.concatenate([[0], np.cumsum(np.bincount(pids, minlength=13))]))
