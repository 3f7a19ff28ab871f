"""Float64 sum / avg held to the tolerance north_star states ("within 1 ULP for floating-point aggregates"), with the ULP distance computed.

The reference adds doubles one after the other in row order (agg_funcs/avg.rs:239-280; DataFusion's sum for Float64), so ITS result moves with
batch and partition boundaries: any two orders differ by up to (n − 1)·ε·Σ|x|.  The GPU path sums in fixed point (comet_device.hpp "Exact
Float64 sums"): the state it emits is the EXACT real sum rounded once — at most half an ULP from the truth, the same bits for every grid,
chunking and row order.  So the checks are:
  * bit-equality with math.fsum (the correctly rounded exact sum) — 0 ULP — per group, through Partial, PartialMerge and Final;
  * against the oracle's sequential restatement of the reference: ≤ 1 ULP wherever the sequential sum is itself exact (integer-valued
    doubles, small well-conditioned sums), and inside the reference's own a-priori error bound elsewhere, with the ULP distance reported;
  * the result does not change when the rows are shuffled, the input arrives in small chunks, or a later chunk widens the exponent window.
"""
import math

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S

pytestmark = pytest.mark.gpu
F64, I32 = S.T_DOUBLE, S.T_INT32
EPS = 2.0 ** -53


def ulp_distance(a: float, b: float) -> int:
    """number of representable doubles between a and b (0 = same value; −0.0 and +0.0 are 0 apart)"""
    if math.isnan(a) or math.isnan(b):
        return 0 if (math.isnan(a) and math.isnan(b)) else 1 << 62

    def key(x):
        (i,) = np.array([x], dtype=np.float64).view(np.int64)
        i = int(i)
        return i if i >= 0 else -(i & 0x7FFFFFFFFFFFFFFF)
    return abs(key(a) - key(b))


def run(plan, table, ncols, batch_rows=8192, config=None):
    kw = {"config": S.config_map(config)} if config else {}
    out = native.execute_to_table([native.HostInput.from_table(table, batch_rows)], ncols, plan.encode(), batch_size=0, **kw)
    return pa.Table.from_batches(out)


def seq_sum(xs):
    s = 0.0
    for x in xs:
        s += x
    return s


def check_sum(got, xs, label):
    """got must be the correctly rounded exact sum; the sequential (reference-order) sum must lie within its own error bound of it."""
    xs = [float(x) for x in xs]
    want = math.fsum(xs)
    assert got == want, f"{label}: got {got!r}, exact sum rounds to {want!r} ({ulp_distance(got, want)} ULP apart)"
    seq = seq_sum(xs)
    bound = (len(xs) - 1) * EPS * math.fsum(abs(x) for x in xs) * (1 + 1e-9)
    assert abs(seq - got) <= bound + 5e-324, f"{label}: sequential {seq!r} vs {got!r} outside the reference's own bound {bound!r}"
    return ulp_distance(got, seq)


def test_ungrouped_sum_and_avg_are_the_correctly_rounded_exact_sum(built):
    rng = np.random.default_rng(1)
    n = 1_000_003
    x = rng.standard_normal(n) * 1e3
    k = rng.integers(0, 100, n).astype(np.int32)
    table = pa.table({"x": pa.array(x, mask=rng.random(n) < 0.1), "k": pa.array(k)})
    pred = S.lt(S.col(1, I32), S.lit(70, I32))
    plan = S.hash_agg(S.filter_(S.scan([F64, I32]), pred), [], [S.sum_(S.col(0, F64), F64), S.avg(S.col(0, F64), F64, F64)])
    got = run(plan, table, 3)
    xs = [v for v, kk in zip(table.column(0).to_pylist(), k) if v is not None and kk < 70]
    d = check_sum(got.column(0)[0].as_py(), xs, "sum")
    assert got.column(1)[0].as_py() == math.fsum(xs) and got.column(2)[0].as_py() == len(xs)
    print(f"ULP distance to the sequential sum over {len(xs)} rows: {d}")
    # Final: avg = sum / count, one more correctly rounded operation
    fin = run(S.final_of(plan, got.schema), got, 2)
    assert fin.column(0)[0].as_py() == math.fsum(xs) and fin.column(1)[0].as_py() == math.fsum(xs) / len(xs)


@pytest.mark.parametrize("ngroups", [5, 40_000])
def test_grouped_sums_low_and_high_cardinality(built, ngroups):
    rng = np.random.default_rng(2 + ngroups)
    n = 600_000
    x = rng.standard_normal(n) * rng.choice([1e-3, 1.0, 1e6], n)
    g = rng.integers(0, ngroups, n).astype(np.int32)
    table = pa.table({"x": pa.array(x), "g": pa.array(g)})
    plan = S.hash_agg(S.scan([F64, I32]), [S.col(1, I32)], [S.sum_(S.col(0, F64), F64), S.avg(S.col(0, F64), F64, F64)])
    got = run(plan, table, 4)
    order = np.argsort(g, kind="stable")
    gs, xs = g[order], x[order]
    starts = np.flatnonzero(np.r_[True, gs[1:] != gs[:-1]])
    groups = {int(gs[a]): xs[a:b] for a, b in zip(starts, np.r_[starts[1:], len(gs)])}
    assert got.num_rows == len(groups)
    worst = 0
    for key, s, a, c in zip(*[got.column(i).to_pylist() for i in range(4)]):
        worst = max(worst, check_sum(s, groups[key], f"group {key}"))
        assert a == s and c == len(groups[key])
    print(f"{len(groups)} groups: largest ULP distance to the per-group sequential sum = {worst}")
    # the same table shuffled, in other batch and chunk sizes: identical bits
    perm = rng.permutation(n)
    again = run(plan, pa.table({"x": pa.array(x[perm]), "g": pa.array(g[perm])}), 4, batch_rows=5000, config={"spark.comet.gpu.chunkRows": 70_000})
    a = dict(zip(got.column(0).to_pylist(), got.column(1).to_pylist()))
    b = dict(zip(again.column(0).to_pylist(), again.column(1).to_pylist()))
    assert a == b


def test_within_one_ulp_of_the_sequential_reference_where_that_is_exact(built):
    """Integer-valued doubles below 2^53: every order of additions is exact, so the reference's sequential sum IS the true sum and the GPU must
    equal it bit for bit (0 ULP ≤ 1 ULP); small well-conditioned sums of arbitrary doubles stay within 1 ULP of the sequential order."""
    from oracle import oracle as O
    rng = np.random.default_rng(3)
    n = 200_000
    table = pa.table({"x": pa.array(rng.integers(-2**30, 2**30, n).astype(np.float64)), "g": pa.array(rng.integers(0, 9, n).astype(np.int32))})
    plan = S.hash_agg(S.scan([F64, I32]), [S.col(1, I32)], [S.sum_(S.col(0, F64), F64), S.avg(S.col(0, F64), F64, F64)])
    got, want = run(plan, table, 4), O.run_plan_to_arrow(S, plan, table)
    rows = lambda t: sorted(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]))
    assert rows(got) == rows(want)
    small = pa.table({"x": pa.array(rng.random(40) + 0.5), "g": pa.array(np.zeros(40, np.int32))})
    got, want = run(plan, small, 4), O.run_plan_to_arrow(S, plan, small)
    assert ulp_distance(got.column(1)[0].as_py(), want.column(1)[0].as_py()) <= 1


def test_exponent_window_moves_with_the_data(built):
    """The default window holds 2^-94 … 2^64.  Tiny values move it down before anything is accumulated (exact); a later chunk with huge values
    moves it up and the earlier accumulators follow; a range wider than the accumulator keeps the large end exact to the last bit it can."""
    rng = np.random.default_rng(4)
    tiny = rng.standard_normal(50_000) * 1e-60
    plan = S.hash_agg(S.scan([F64]), [], [S.sum_(S.col(0, F64), F64)])
    assert run(plan, pa.table({"x": pa.array(tiny)}), 1).column(0)[0].as_py() == math.fsum(tiny.tolist())
    # chunk 1: values around 1; chunk 2: values around 1e40 — the window has to move up between chunks (host path, 8192-row chunks)
    x = np.concatenate([rng.standard_normal(8192), rng.standard_normal(8192) * 1e40, rng.standard_normal(8192) * 1e25])
    got = run(plan, pa.table({"x": pa.array(x)}), 1, config={"spark.comet.gpu.chunkRows": 8192}).column(0)[0].as_py()
    want = math.fsum(x.tolist())
    # after the move the first chunk's bits below 2^(top + 10 − 158) are gone: bounded by rows · 2^s, far below one ULP of the result here
    assert ulp_distance(got, want) <= 1
    gplan = S.hash_agg(S.scan([F64, I32]), [S.col(1, I32)], [S.sum_(S.col(0, F64), F64)])
    g = np.tile(np.arange(3, dtype=np.int32), len(x) // 3)
    gt = run(gplan, pa.table({"x": pa.array(x), "g": pa.array(g)}), 2, config={"spark.comet.gpu.chunkRows": 8192})
    for key, s in zip(gt.column(0).to_pylist(), gt.column(1).to_pylist()):
        assert ulp_distance(s, math.fsum(x[g == key].tolist())) <= 1
    # 400 binary orders of magnitude in one sum: the large values dominate, the result is still the correctly rounded sum
    wide = np.concatenate([rng.standard_normal(1000) * 1e60, rng.standard_normal(1000) * 1e-60])
    got = run(plan, pa.table({"x": pa.array(wide)}), 1).column(0)[0].as_py()
    assert ulp_distance(got, math.fsum(wide.tolist())) <= 1


def test_non_finite_addends_follow_ieee(built):
    plan = S.hash_agg(S.scan([F64, I32]), [S.col(1, I32)], [S.sum_(S.col(0, F64), F64)])
    inf, nan = float("inf"), float("nan")
    x = [1.0, inf, 2.0, -inf, 5.0, inf, -inf, nan, 1.5, 2.5, -0.0, 0.0]
    g = [0, 0, 1, 1, 2, 3, 3, 4, 5, 5, 6, 6]
    got = run(plan, pa.table({"x": pa.array(x), "g": pa.array(g, pa.int32())}), 2)
    res = dict(zip(got.column(0).to_pylist(), got.column(1).to_pylist()))
    assert res[0] == inf and res[1] == -inf and res[2] == 5.0 and math.isnan(res[3]) and math.isnan(res[4]) and res[5] == 4.0 and res[6] == 0.0
    ug = run(S.hash_agg(S.scan([F64]), [], [S.sum_(S.col(0, F64), F64)]), pa.table({"x": pa.array([1e308, 1e308, -1e308])}), 1)
    assert ug.column(0)[0].as_py() == 1e308          # the exact sum is finite even though a running double sum overflows


@pytest.mark.parametrize("grouped", [False, True])
def test_seven_float_sums_and_averages_in_one_aggregate(built, grouped):
    """Up to EIGHT exact Float64 sums per aggregate (round 6; four before: the scales of sums 4-7 travel in a second kernel argument word): four sums and three
    averages over seven columns of very different magnitude (2^-30 … 2^40, so their fixed-point windows differ), NULLs, Partial → Final, grouped and ungrouped —
    every sum bit-equal to math.fsum, every average to the correctly rounded sum divided by the count (avg.rs:239-280)."""
    rng = np.random.default_rng(97)
    n = 120_000
    mags = [2.0 ** -30, 1.0, 2.0 ** 10, 2.0 ** 20, 2.0 ** 40, 2.0 ** -5, 2.0 ** 15]
    cols = {f"x{i}": pa.array(rng.standard_normal(n) * m, mask=rng.random(n) < 0.03) for i, m in enumerate(mags)}
    g = rng.integers(0, 5, n).astype(np.int32)
    table = pa.table({"g": pa.array(g), **cols})
    fields = [I32] + [F64] * 7
    aggs = [S.sum_(S.col(1 + i, F64), F64) for i in range(4)] + [S.avg(S.col(5 + i, F64), F64, F64) for i in range(3)]
    groups = [S.col(0, I32)] if grouped else []
    partial = S.hash_agg(S.scan(fields), groups, aggs, S.PARTIAL)
    ng = len(groups)
    nstate = ng + 4 + 3 * 2
    states = run(partial, table, nstate, batch_rows=20_000)
    sfields = [I32] * ng + [F64] * 4 + [F64, S.T_INT64] * 3
    final = S.hash_agg(S.scan(sfields), [S.col(i, I32) for i in range(ng)], aggs, S.FINAL)
    out = run(final, states, ng + 7)
    keys = sorted(set(g.tolist())) if grouped else [None]
    assert out.num_rows == len(keys)
    by_key = {(out.column(0)[r].as_py() if grouped else None): r for r in range(out.num_rows)}
    for k in keys:
        sel = (g == k) if grouped else np.ones(n, bool)
        r = by_key[k]
        for i in range(7):
            xs = [v for v, keep in zip(table.column(1 + i).to_pylist(), sel) if keep and v is not None]
            got = out.column(ng + i)[r].as_py()
            want = math.fsum(xs) if i < 4 else math.fsum(xs) / len(xs)
            assert got == want, (k, i, got, want, ulp_distance(got, want))


def test_nine_float_sums_are_refused_by_name(built):
    fields = [F64] * 9
    plan = S.hash_agg(S.scan(fields), [], [S.sum_(S.col(i, F64), F64) for i in range(9)], S.PARTIAL)
    with pytest.raises(native.CometNativeException, match="more than 8 distinct Float64 sums"):
        native.compile_plan(plan.encode())
