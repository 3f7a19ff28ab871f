"""Seeded differential fuzzing: random Filter / Project / HashAggregate plans over nullable int, decimal, float, bool and date
columns, GPU engine against the oracle.  Elementwise results must be bit-identical (integers wrap like LEGACY mode, decimals
follow Spark's result types with CheckOverflow, float ops are single IEEE operations); only float SUMs get a tolerance."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu

I32, I64, F64, B, DATE = S.T_INT32, S.T_INT64, S.T_DOUBLE, S.T_BOOL, S.T_DATE
D = S.decimal(12, 2)
FIELDS = [I32, I64, D, D, F64, F64, B, DATE, I32]      # last column: small-cardinality group key


def _table(n, seed):
    rng = np.random.default_rng(seed)
    m = lambda p=0.12: rng.random(n) < p
    f = rng.standard_normal(n) * 100
    f[::53] = 0.0
    return pa.table({
        "i": pa.array(rng.integers(-2**31, 2**31 - 1, n), pa.int32(), mask=m()),
        "l": pa.array(rng.integers(-2**62, 2**62, n), pa.int64(), mask=m()),
        "d1": pa.Array.from_buffers(pa.decimal128(12, 2), n, [pa.py_buffer(np.packbits(~m(), bitorder="little").tobytes()),
                                                             tpch._dec128_array(rng.integers(-10**11, 10**11, n), 12, 2).buffers()[1]]),
        "d2": tpch._dec128_array(rng.integers(-10**6, 10**6, n), 12, 2),
        "f": pa.array(f, mask=m()),
        "g": pa.array(rng.standard_normal(n)),
        "b": pa.array(rng.random(n) < 0.5, mask=m()),
        "dt": pa.array(rng.integers(0, 20000, n), pa.int32(), mask=m()).cast(pa.date32()),
        "k": pa.array(rng.integers(0, 9, n), pa.int32(), mask=m(0.05)),
    })


class Gen:
    def __init__(self, rng):
        self.rng = rng

    def pick(self, xs):
        return xs[int(self.rng.integers(0, len(xs)))]

    def int32(self, d):
        if d == 0 or self.rng.random() < 0.3:
            return self.pick([S.col(0, I32), S.col(8, I32), S.lit(int(self.rng.integers(-1000, 1000)), I32)])
        k = self.pick(["add", "subtract", "multiply", "if", "case", "neg"])
        if k == "neg":      # NegativeExpr, LEGACY wrap (math_funcs/negative.rs:100-160)
            return S.Expr("unary_minus", [self.int32(d - 1)])
        if k == "if":
            return S.if_(self.boolean(d - 1), self.int32(d - 1), self.int32(d - 1))
        if k == "case":
            return S.case_when([(self.boolean(d - 1), self.int32(d - 1)), (self.boolean(d - 1), self.int32(d - 1))],
                               self.int32(d - 1) if self.rng.random() < 0.5 else None)
        return S.math(k, self.int32(d - 1), self.int32(d - 1), I32)

    def int64(self, d):
        if d == 0 or self.rng.random() < 0.3:
            return self.pick([S.col(1, I64), S.lit(int(self.rng.integers(-10**12, 10**12)), I64), S.cast(S.col(0, I32), I64)])
        k = self.pick(["add", "subtract", "multiply", "if", "neg"])
        if k == "neg":
            return S.Expr("unary_minus", [self.int64(d - 1)])
        if k == "if":
            return S.if_(self.boolean(d - 1), self.int64(d - 1), self.int64(d - 1))
        return S.math(k, self.int64(d - 1), self.int64(d - 1), I64)

    def f64(self, d):
        if d == 0 or self.rng.random() < 0.3:
            return self.pick([S.col(4, F64), S.col(5, F64), S.lit(float(self.rng.integers(-50, 50)) / 4, F64)])
        k = self.pick(["add", "subtract", "multiply", "divide", "if", "neg"])
        if k == "neg":
            return S.Expr("unary_minus", [self.f64(d - 1)])
        if k == "if":
            return S.if_(self.boolean(d - 1), self.f64(d - 1), self.f64(d - 1))
        return S.math(k, self.f64(d - 1), self.f64(d - 1), F64)

    def dec(self, d):
        """returns (expr, precision, scale); Spark's result types (DecimalPrecision.scala) while they stay ≤ 38"""
        if d == 0 or self.rng.random() < 0.35:
            c = self.pick([(S.col(2, D), 12, 2), (S.col(3, D), 12, 2), (S.lit(int(self.rng.integers(-99999, 99999)), D), 12, 2)])
            return c
        if self.rng.random() < 0.15:
            a, p1, s1 = self.dec(d - 1)
            return S.Expr("unary_minus", [a]), p1, s1
        (a, p1, s1), (b, p2, s2) = self.dec(d - 1), self.dec(d - 1)
        k = self.pick(["add", "subtract", "multiply"])
        if k == "multiply":
            p, s = p1 + p2 + 1, s1 + s2
        else:
            s = max(s1, s2)
            p = max(p1 - s1, p2 - s2) + s + 1
        if p > 38:
            return a, p1, s1
        t = S.decimal(p, s)
        return S.check_overflow(S.math(k, a, b, t), t), p, s

    def boolean(self, d):
        if d == 0 or self.rng.random() < 0.25:
            return self.pick([S.col(6, B), S.is_null(S.col(0, I32)), S.is_not_null(S.col(4, F64)), S.lt(S.col(7, DATE), S.lit(int(self.rng.integers(0, 20000)), DATE))])
        k = self.pick(["cmp_i", "cmp_l", "cmp_f", "cmp_d", "and", "or", "not", "in"])
        ops = [S.eq, S.neq, S.lt, S.lt_eq, S.gt, S.gt_eq]
        if k == "cmp_i":
            return self.pick(ops)(self.int32(d - 1), self.int32(d - 1))
        if k == "cmp_l":
            return self.pick(ops)(self.int64(d - 1), self.int64(d - 1))
        if k == "cmp_f":
            return self.pick(ops)(self.f64(d - 1), self.f64(d - 1))
        if k == "cmp_d":
            (a, p1, s1), (b, p2, s2) = self.dec(d - 1), self.dec(d - 1)
            if s1 != s2:
                return S.is_null(a)
            return self.pick(ops)(a, b)
        if k == "and":
            return S.and_(self.boolean(d - 1), self.boolean(d - 1))
        if k == "or":
            return S.or_(self.boolean(d - 1), self.boolean(d - 1))
        if k == "not":
            return S.not_(self.boolean(d - 1))
        return S.in_(S.col(8, I32), [S.lit(int(x), I32) for x in self.rng.integers(0, 9, 3)] + ([S.lit(None, I32)] if self.rng.random() < 0.3 else []),
                     negated=bool(self.rng.random() < 0.3))


def _bits(col):
    c = col.combine_chunks()
    if pa.types.is_floating(c.type):
        v = c.to_numpy(zero_copy_only=False)
        return [None if x is None or (isinstance(x, float) and x != x and not ok) else (np.float64(x).view(np.int64).item() if ok else None)
                for x, ok in zip(v, np.asarray(c.is_valid()))]
    return c.to_pylist()


@pytest.mark.parametrize("seed", range(64))
def test_random_filter_project(built, seed):
    from oracle import oracle as O
    rng = np.random.default_rng(1000 + seed)
    g = Gen(rng)
    t = _table(6000, seed)
    outs = [g.int32(3), g.int64(2), g.f64(3), g.dec(3)[0], g.boolean(3), S.col(7, DATE)]
    plan = S.project(S.filter_(S.scan(FIELDS), g.boolean(3)), outs) if rng.random() < 0.8 else S.project(S.scan(FIELDS), outs)
    want = O.run_plan_to_arrow(S, plan, t)
    got = native.execute_to_table([native.HostInput.from_table(t)], len(outs), plan.encode(), batch_size=0)
    got = pa.Table.from_batches(got) if got else None
    assert (got.num_rows if got is not None else 0) == want.num_rows
    if got is None:
        return
    for i in range(len(outs)):
        assert got.column(i).type == want.column(i).type, i
        assert _bits(got.column(i)) == _bits(want.column(i)), f"seed {seed} column {i}"


@pytest.mark.parametrize("seed", range(24))
def test_random_grouped_aggregate(built, seed):
    from oracle import oracle as O
    rng = np.random.default_rng(2000 + seed)
    g = Gen(rng)
    t = _table(8000, 100 + seed)
    de, p, s = g.dec(2)
    proj = [S.col(8, I32), g.int32(2), de, g.f64(2), g.boolean(2)]
    child = S.project(S.filter_(S.scan(FIELDS), g.boolean(2)), proj)
    DT = S.decimal(p, s)
    aggs = [S.count(S.col(1, I32)), S.sum_(S.cast(S.col(1, I32), I64), I64), S.sum_(S.col(2, DT), S.decimal(min(38, p + 10), s)), (S.min_(S.col(2, DT), DT) if p <= 18 else S.count(S.col(2, DT))),   # grouped min/max of decimal(>18): known gap
            S.max_(S.col(1, I32), I32), S.sum_(S.col(3, F64), F64), S.count(S.col(4, B))]
    plan = S.hash_agg(child, [S.col(0, I32)], aggs)
    want = O.run_plan_to_arrow(S, plan, t)
    got = native.execute_to_table([native.HostInput.from_table(t)], want.num_columns, plan.encode(), batch_size=0)
    got = pa.Table.from_batches(got) if got else None
    rows = lambda tb: sorted(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]), key=lambda r: (r[0] is None, r[0] or 0))
    gr, wr = rows(got) if got is not None else [], rows(want)
    assert len(gr) == len(wr)
    fcol = [i for i, f in enumerate(want.schema) if pa.types.is_floating(f.type)]
    for a, b in zip(gr, wr):
        for i, (x, y) in enumerate(zip(a, b)):
            if i in fcol and x is not None and y is not None:
                assert x == pytest.approx(y, rel=1e-9, abs=1e-9) or (x != x and y != y), (seed, i)
            else:
                assert x == y, (seed, i, a, b)
