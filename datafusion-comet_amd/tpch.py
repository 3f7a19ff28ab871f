"""TPC-H plan shapes (as Comet serialises them) and seeded synthetic lineitem data.

Plans follow the reference's native stage-1 shapes: Q6 SURVEY §3.3, Q1 SURVEY §3.4; decimal result types
are Spark's (Appendix B; golden schema spark/src/test/resources/tpch-query-results/q1.sql.out:3-4).
"""
from __future__ import annotations

import datetime

import numpy as np
import pyarrow as pa

from . import serde as S

DEC = S.decimal(12, 2)
EPOCH = datetime.date(1970, 1, 1)


def days(y, m, d) -> int:
    return (datetime.date(y, m, d) - EPOCH).days


# ------------------------------------------------------------------ data

def _dec128_array(int64_values: np.ndarray, p: int, s: int) -> pa.Array:
    """int64 unscaled values → Decimal128(p, s) Arrow array (16 B little-endian two's complement)."""
    n = len(int64_values)
    buf = np.empty((n, 2), dtype=np.int64)
    buf[:, 0] = int64_values
    buf[:, 1] = int64_values >> 63
    return pa.Array.from_buffers(pa.decimal128(p, s), n, [None, pa.py_buffer(buf.tobytes())])


def lineitem_q6(n: int, seed: int = 6, null_frac: float = 0.0) -> pa.Table:
    """Columns in scan order: l_quantity, l_extendedprice, l_discount (decimal(12,2)), l_shipdate (date32)."""
    rng = np.random.default_rng(seed)
    qty = rng.integers(1, 51, n, dtype=np.int64)
    price = qty * rng.integers(90000, 210001, n, dtype=np.int64)      # qty × U[900.00, 2100.00]
    disc = rng.integers(0, 11, n, dtype=np.int64)                      # 0.00 .. 0.10
    ship = rng.integers(days(1992, 1, 2), days(1998, 12, 1) + 1, n, dtype=np.int64).astype(np.int32)
    cols = [_dec128_array(qty * 100, 12, 2), _dec128_array(price, 12, 2), _dec128_array(disc, 12, 2),
            pa.array(ship, type=pa.int32()).cast(pa.date32())]
    if null_frac > 0:
        out = []
        for i, c in enumerate(cols):
            mask = rng.random(n) < null_frac
            out.append(pa.Array.from_buffers(c.type, n, [pa.py_buffer(np.packbits(~mask, bitorder="little").tobytes()),
                                                        c.buffers()[1]], null_count=int(mask.sum())))
        cols = out
    return pa.table(cols, names=["l_quantity", "l_extendedprice", "l_discount", "l_shipdate"])


def lineitem_q1(n: int, seed: int = 1) -> pa.Table:
    """l_quantity, l_extendedprice, l_discount, l_tax (decimal(12,2)), l_returnflag, l_linestatus (utf8), l_shipdate."""
    rng = np.random.default_rng(seed)
    qty = rng.integers(1, 51, n, dtype=np.int64)
    price = qty * rng.integers(90000, 210001, n, dtype=np.int64)
    disc = rng.integers(0, 11, n, dtype=np.int64)
    tax = rng.integers(0, 9, n, dtype=np.int64)
    ship = rng.integers(days(1992, 1, 2), days(1998, 12, 1) + 1, n, dtype=np.int64).astype(np.int32)
    # dbgen correlation: shipped after 1995-06-17 → 'N'/'O'; earlier → 'R' or 'A' / 'F'
    cutoff = days(1995, 6, 17)
    late = ship > cutoff
    rf = np.where(late, ord("N"), np.where(rng.random(n) < 0.5, ord("R"), ord("A"))).astype(np.uint8)
    ls = np.where(late, ord("O"), ord("F")).astype(np.uint8)
    # a sliver of 'N','F' like the real data (orders straddling the cutoff)
    straddle = (~late) & (ship > cutoff - 60) & (rng.random(n) < 0.3)
    rf = np.where(straddle, ord("N"), rf).astype(np.uint8)
    offs = np.arange(n + 1, dtype=np.int32)

    def utf8(bytes_arr):
        return pa.Array.from_buffers(pa.utf8(), n, [None, pa.py_buffer(offs.tobytes()), pa.py_buffer(bytes_arr.tobytes())])

    return pa.table([_dec128_array(qty * 100, 12, 2), _dec128_array(price, 12, 2), _dec128_array(disc, 12, 2),
                     _dec128_array(tax, 12, 2), utf8(rf), utf8(ls), pa.array(ship, type=pa.int32()).cast(pa.date32())],
                    names=["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"])


# ------------------------------------------------------------------ plans

def q6_plan(mode: int = S.PARTIAL, source: "S.Operator" = None) -> S.Operator:
    """TPC-H Q6 stage 1; `source` replaces the Scan leaf (e.g. a NativeScan over Parquet files):
    HashAgg(Partial, sum(CheckOverflow(price*disc → dec(25,4))) : dec(35,4))
      ← Project[price, disc] ← Filter(shipdate >= 1994-01-01 AND shipdate < 1995-01-01 AND
                                       disc >= 0.05 AND disc <= 0.07 AND qty < 24.00) ← Scan."""
    fields = [DEC, DEC, DEC, S.T_DATE]
    qty, price, disc, ship = (S.col(i, t) for i, t in enumerate(fields))
    pred = S.and_(S.and_(S.and_(S.and_(
        S.gt_eq(ship, S.lit(days(1994, 1, 1), S.T_DATE)),
        S.lt(ship, S.lit(days(1995, 1, 1), S.T_DATE))),
        S.gt_eq(disc, S.lit(5, DEC))),
        S.lt_eq(disc, S.lit(7, DEC))),
        S.lt(qty, S.lit(2400, DEC)))
    f = S.filter_(source if source is not None else S.scan(fields), pred)
    p = S.project(f, [price, disc])
    revenue = S.check_overflow(S.math("multiply", S.col(0, DEC), S.col(1, DEC), S.decimal(25, 4)), S.decimal(25, 4))
    return S.hash_agg(p, [], [S.sum_(revenue, S.decimal(35, 4))], mode)


Q6_NUM_OUTPUT_COLS = 2  # (sum dec(35,4), is_empty bool)
Q6_BYTES_PER_ROW = 4 + 16 + 16 + 16  # Arrow layout of the four referenced columns


def q1_plan(mode: int = S.PARTIAL, source: "S.Operator" = None) -> S.Operator:
    """TPC-H Q1 stage 1.  Scan order: qty, price, disc, tax, returnflag, linestatus, shipdate.
    HashAgg(Partial, keys=[returnflag, linestatus],
            [sum(qty) d(22,2), sum(price) d(22,2), sum(disc_price) d(36,4), sum(charge) d(38,6),
             avg(qty), avg(price), avg(disc) → d(16,6) with sum type d(22,2), count(1)])
      ← Project[returnflag, linestatus, qty, price, disc,
                disc_price = CheckOverflow(price * CheckOverflow(1 - disc → d(13,2)) → d(26,4)),
                charge     = wide (disc_price d(26,4)) * CheckOverflow(1 + tax → d(13,2)) → d(38,6)]
      ← Filter(shipdate <= 1998-09-02) ← Scan."""
    fields = [DEC, DEC, DEC, DEC, S.T_STRING, S.T_STRING, S.T_DATE]
    qty, price, disc, tax, rf, ls, ship = (S.col(i, t) for i, t in enumerate(fields))
    f = S.filter_(source if source is not None else S.scan(fields), S.lt_eq(ship, S.lit(days(1998, 9, 2), S.T_DATE)))
    one = S.lit(100, S.decimal(12, 2))   # Spark promotes the literal 1 to decimal(12,2)? it sends Decimal(1,0) cast; keep (12,2)
    one_minus = S.check_overflow(S.math("subtract", one, disc, S.decimal(13, 2)), S.decimal(13, 2))
    one_plus = S.check_overflow(S.math("add", one, tax, S.decimal(13, 2)), S.decimal(13, 2))
    disc_price = S.check_overflow(S.math("multiply", price, one_minus, S.decimal(26, 4)), S.decimal(26, 4))
    charge = S.check_overflow(S.math("multiply", disc_price, one_plus, S.decimal(38, 6)), S.decimal(38, 6))
    p = S.project(f, [rf, ls, qty, price, disc, disc_price, charge])
    D22, D16 = S.decimal(22, 2), S.decimal(16, 6)
    c = lambda i, t: S.col(i, t)
    aggs = [S.sum_(c(2, DEC), D22), S.sum_(c(3, DEC), D22), S.sum_(c(5, S.decimal(26, 4)), S.decimal(36, 4)),
            S.sum_(c(6, S.decimal(38, 6)), S.decimal(38, 6)),
            S.avg(c(2, DEC), D16, D22), S.avg(c(3, DEC), D16, D22), S.avg(c(4, DEC), D16, D22),
            S.count(S.lit(1, S.T_INT32))]
    return S.hash_agg(p, [c(0, S.T_STRING), c(1, S.T_STRING)], aggs, mode)


Q1_NUM_OUTPUT_COLS = 2 + 4 * 2 + 3 * 2 + 1  # keys + (sum,is_empty)×4 + (sum,count)×3 + count
Q1_BYTES_PER_ROW = 4 + 4 * 16 + 2 * (4 + 1)


def warm_plans():
    """Plans whose fused kernels build() pre-compiles into the code-object cache."""
    return [q6_plan(), q1_plan(), q3_plan()] + [pl for pl, _, _ in q3_stage_plans().values()]


# ------------------------------------------------------------------ Q3

SEGMENTS = [b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"MACHINERY", b"HOUSEHOLD"]


def q3_tables(n_orders: int, seed: int = 3):
    """customer (n_orders/10 rows), orders (n_orders), lineitem (~4 per order) shaped like dbgen:
    customer[c_custkey int64, c_mktsegment utf8], orders[o_orderkey, o_custkey int64, o_orderdate date32,
    o_shippriority int32], lineitem[l_orderkey int64, l_extendedprice, l_discount dec(12,2), l_shipdate date32]."""
    rng = np.random.default_rng(seed)
    n_c = max(1, n_orders // 10)
    ckey = np.arange(1, n_c + 1, dtype=np.int64)
    seg = rng.integers(0, 5, n_c)
    lens = np.array([len(s) for s in SEGMENTS], np.int32)
    offs = np.zeros(n_c + 1, np.int32)
    offs[1:] = np.cumsum(lens[seg])
    data = b"".join(SEGMENTS[i] for i in seg)
    customer = pa.table([pa.array(ckey), pa.Array.from_buffers(pa.utf8(), n_c, [None, pa.py_buffer(offs.tobytes()), pa.py_buffer(data)])],
                        names=["c_custkey", "c_mktsegment"])
    okey = (np.arange(n_orders, dtype=np.int64) // 8) * 32 + (np.arange(n_orders, dtype=np.int64) % 8) + 1   # dbgen's sparse order keys
    ocust = rng.integers(1, n_c + 1, n_orders, dtype=np.int64)
    odate = rng.integers(days(1992, 1, 1), days(1998, 8, 2) + 1, n_orders, dtype=np.int64).astype(np.int32)
    orders = pa.table([pa.array(okey), pa.array(ocust), pa.array(odate, pa.int32()).cast(pa.date32()), pa.array(np.zeros(n_orders, np.int32))],
                      names=["o_orderkey", "o_custkey", "o_orderdate", "o_shippriority"])
    per = rng.integers(1, 8, n_orders)
    lo = np.repeat(np.arange(n_orders), per)
    n_l = len(lo)
    lkey = okey[lo]
    qty = rng.integers(1, 51, n_l, dtype=np.int64)
    price = qty * rng.integers(90000, 210001, n_l, dtype=np.int64)
    disc = rng.integers(0, 11, n_l, dtype=np.int64)
    ship = (odate[lo].astype(np.int64) + rng.integers(1, 122, n_l)).astype(np.int32)
    lineitem = pa.table([pa.array(lkey), _dec128_array(price, 12, 2), _dec128_array(disc, 12, 2), pa.array(ship, pa.int32()).cast(pa.date32())],
                        names=["l_orderkey", "l_extendedprice", "l_discount", "l_shipdate"])
    return customer, orders, lineitem


def q3_plan() -> S.Operator:
    """TPC-H Q3 as one native stage: customer ⋈ orders ⋈ lineitem (Inner, build = left), project, partial aggregate.
    Scan leaves in depth-first order: customer, orders, lineitem."""
    cutoff = days(1995, 3, 15)
    cust = S.project(S.filter_(S.scan([S.T_INT64, S.T_STRING]), S.eq(S.col(1, S.T_STRING), S.lit("BUILDING", S.T_STRING))), [S.col(0, S.T_INT64)])
    ofields = [S.T_INT64, S.T_INT64, S.T_DATE, S.T_INT32]
    orders = S.filter_(S.scan(ofields), S.lt(S.col(2, S.T_DATE), S.lit(cutoff, S.T_DATE)))
    j1 = S.hash_join(cust, orders, [S.col(0, S.T_INT64)], [S.col(1, S.T_INT64)], S.INNER, S.BUILD_LEFT)
    # j1 output: c_custkey, o_orderkey, o_custkey, o_orderdate, o_shippriority
    j1p = S.project(j1, [S.col(1, S.T_INT64), S.col(3, S.T_DATE), S.col(4, S.T_INT32)])
    lfields = [S.T_INT64, DEC, DEC, S.T_DATE]
    li = S.project(S.filter_(S.scan(lfields), S.gt(S.col(3, S.T_DATE), S.lit(cutoff, S.T_DATE))),
                   [S.col(0, S.T_INT64), S.col(1, DEC), S.col(2, DEC)])
    j2 = S.hash_join(j1p, li, [S.col(0, S.T_INT64)], [S.col(0, S.T_INT64)], S.INNER, S.BUILD_LEFT)
    # j2 output: o_orderkey, o_orderdate, o_shippriority, l_orderkey, l_extendedprice, l_discount
    one_minus = S.check_overflow(S.math("subtract", S.lit(100, DEC), S.col(5, DEC), S.decimal(13, 2)), S.decimal(13, 2))
    rev = S.check_overflow(S.math("multiply", S.col(4, DEC), one_minus, S.decimal(26, 4)), S.decimal(26, 4))
    p = S.project(j2, [S.col(3, S.T_INT64), S.col(1, S.T_DATE), S.col(2, S.T_INT32), rev])
    return S.hash_agg(p, [S.col(0, S.T_INT64), S.col(1, S.T_DATE), S.col(2, S.T_INT32)], [S.sum_(S.col(3, S.decimal(26, 4)), S.decimal(36, 4))])


Q3_NUM_OUTPUT_COLS = 3 + 2


def q3_stage_plans() -> dict:
    """TPC-H Q3 cut at its exchanges, the way Spark plans it for a partitioned run: every stage is one native
    plan per partition; `exchange` names the hash-partitioning key column of the stage's output (None = no exchange).

      customer:  Scan → Filter(c_mktsegment = 'BUILDING') → Project(c_custkey)                       ⇒ hash(c_custkey)
      orders:    Scan → Filter(o_orderdate < 1995-03-15)                                             ⇒ hash(o_custkey)
      join1:     customer' ⋈ orders' on custkey → Project(o_orderkey, o_orderdate, o_shippriority)   ⇒ hash(o_orderkey)
      lineitem:  Scan → Filter(l_shipdate > 1995-03-15) → Project(l_orderkey, price, discount)       ⇒ hash(l_orderkey)
      join2agg:  join1' ⋈ lineitem' on orderkey → Project → HashAggregate(Partial)   (groups are partition-local)
    """
    cutoff = days(1995, 3, 15)
    I64, DATE, I32 = S.T_INT64, S.T_DATE, S.T_INT32
    customer = S.project(S.filter_(S.scan([I64, S.T_STRING]), S.eq(S.col(1, S.T_STRING), S.lit("BUILDING", S.T_STRING))), [S.col(0, I64)])
    ofields = [I64, I64, DATE, I32]
    orders = S.project(S.filter_(S.scan(ofields), S.lt(S.col(2, DATE), S.lit(cutoff, DATE))), [S.col(i, t) for i, t in enumerate(ofields)])
    j1 = S.hash_join(S.scan([I64]), S.scan(ofields), [S.col(0, I64)], [S.col(1, I64)], S.INNER, S.BUILD_LEFT)
    join1 = S.project(j1, [S.col(1, I64), S.col(3, DATE), S.col(4, I32)])
    lineitem = S.project(S.filter_(S.scan([I64, DEC, DEC, DATE]), S.gt(S.col(3, DATE), S.lit(cutoff, DATE))),
                         [S.col(0, I64), S.col(1, DEC), S.col(2, DEC)])
    j2 = S.hash_join(S.scan([I64, DATE, I32]), S.scan([I64, DEC, DEC]), [S.col(0, I64)], [S.col(0, I64)], S.INNER, S.BUILD_LEFT)
    one_minus = S.check_overflow(S.math("subtract", S.lit(100, DEC), S.col(5, DEC), S.decimal(13, 2)), S.decimal(13, 2))
    rev = S.check_overflow(S.math("multiply", S.col(4, DEC), one_minus, S.decimal(26, 4)), S.decimal(26, 4))
    p = S.project(j2, [S.col(3, I64), S.col(1, DATE), S.col(2, I32), rev])
    join2agg = S.hash_agg(p, [S.col(0, I64), S.col(1, DATE), S.col(2, I32)], [S.sum_(S.col(3, S.decimal(26, 4)), S.decimal(36, 4))])
    return {"customer": (customer, 1, 0), "orders": (orders, 4, 1), "join1": (join1, 3, 0), "lineitem": (lineitem, 3, 0),
            "join2agg": (join2agg, Q3_NUM_OUTPUT_COLS, None)}   # name → (plan, output columns, exchange key column)


def lineitem_q1_device(n: int, device="cuda:0", seed: int = 1):
    """The Q1 lineitem columns generated directly in HBM with torch (SF100 = 600 M rows = 46.8 GB does not have to pass
    through host memory).  Same distributions as lineitem_q1; returns a native.DeviceTable."""
    import torch
    from .native import DeviceTable
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def dec(v):  # int64 unscaled (non-negative here) → Decimal128 little-endian limbs
        buf = torch.zeros((n, 2), dtype=torch.int64, device=device)
        buf[:, 0] = v
        return buf.view(torch.uint8).reshape(-1)

    qty = torch.randint(1, 51, (n,), generator=g, device=device, dtype=torch.int64)
    price = qty * torch.randint(90000, 210001, (n,), generator=g, device=device, dtype=torch.int64)
    disc = torch.randint(0, 11, (n,), generator=g, device=device, dtype=torch.int64)
    tax = torch.randint(0, 9, (n,), generator=g, device=device, dtype=torch.int64)
    ship = torch.randint(days(1992, 1, 2), days(1998, 12, 1) + 1, (n,), generator=g, device=device, dtype=torch.int32)
    late = ship > days(1995, 6, 17)
    coin = torch.rand((n,), generator=g, device=device) < 0.5
    rf = torch.where(late, torch.tensor(ord("N"), device=device, dtype=torch.uint8),
                     torch.where(coin, torch.tensor(ord("R"), device=device, dtype=torch.uint8), torch.tensor(ord("A"), device=device, dtype=torch.uint8)))
    ls = torch.where(late, torch.tensor(ord("O"), device=device, dtype=torch.uint8), torch.tensor(ord("F"), device=device, dtype=torch.uint8))
    # dbgen's fourth group: a sliver of ('N','F') from orders straddling the cutoff (q1.sql.out:6-9 has 38 854 of 6 M rows)
    straddle = (~late) & (ship > days(1995, 6, 17) - 60) & (torch.rand((n,), generator=g, device=device) < 0.3)
    rf = torch.where(straddle, torch.tensor(ord("N"), device=device, dtype=torch.uint8), rf)
    del late, coin, straddle
    offs = torch.arange(n + 1, device=device, dtype=torch.int32).view(torch.uint8).reshape(-1)
    schema = pa.schema([("l_quantity", pa.decimal128(12, 2)), ("l_extendedprice", pa.decimal128(12, 2)), ("l_discount", pa.decimal128(12, 2)),
                        ("l_tax", pa.decimal128(12, 2)), ("l_returnflag", pa.utf8()), ("l_linestatus", pa.utf8()), ("l_shipdate", pa.date32())])
    values = [dec(qty * 100), dec(price), dec(disc), dec(tax), offs, offs.clone(), ship.view(torch.uint8).reshape(-1)]
    aux = [None, None, None, None, rf, ls, None]
    checks = {"qty": qty, "price": price, "disc": disc, "tax": tax, "ship": ship, "rf": rf, "ls": ls}
    return DeviceTable(schema, n, values, [None] * 7, device, aux), checks


def q3_tables_device(n_orders: int, world: int = 1, rank: int = 0, device="cuda:0", seed: int = 3):
    """This rank's shard of the Q3 tables generated directly in HBM (same shapes as q3_tables; SF100 = 150 M orders):
    orders rows shard_range(n_orders), their lineitems (1..7 each), customers shard_range(n_orders/10).  Deterministic in
    (seed, world, rank).  Returns (customer, orders, lineitem DeviceTables, raw) where raw holds the generating tensors
    for an independent torch cross-check."""
    import torch
    from .native import DeviceTable
    from .parallel import shard_range
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1000 + rank)
    n_c_total = max(1, n_orders // 10)
    c0, nc = shard_range(n_c_total, world, rank)
    o0, no = shard_range(n_orders, world, rank)
    u8 = lambda t: t.contiguous().view(torch.uint8).reshape(-1)

    def dec(v):
        buf = torch.zeros((v.numel(), 2), dtype=torch.int64, device=device)
        buf[:, 0] = v
        return buf.view(torch.uint8).reshape(-1)

    # customer
    ckey = torch.arange(c0 + 1, c0 + nc + 1, device=device, dtype=torch.int64)
    seg = torch.randint(0, 5, (nc,), generator=g, device=device)
    width = max(len(s) for s in SEGMENTS)
    tab = torch.zeros((5, width), dtype=torch.uint8)
    lens = torch.zeros(5, dtype=torch.int32)
    for i, s in enumerate(SEGMENTS):
        tab[i, :len(s)] = torch.tensor(list(s), dtype=torch.uint8)
        lens[i] = len(s)
    tab, lens = tab.to(device), lens.to(device)
    offs = torch.zeros(nc + 1, dtype=torch.int32, device=device)
    offs[1:] = torch.cumsum(lens[seg], 0, dtype=torch.int32)
    mask = torch.arange(width, device=device)[None, :] < lens[seg][:, None]
    cbytes = tab[seg][mask]
    if cbytes.numel() == 0:
        cbytes = torch.zeros(1, dtype=torch.uint8, device=device)
    customer = DeviceTable(pa.schema([("c_custkey", pa.int64()), ("c_mktsegment", pa.utf8())]), nc, [u8(ckey), u8(offs)], [None, None], device,
                           [None, cbytes])
    # orders
    oi = torch.arange(o0, o0 + no, device=device, dtype=torch.int64)
    okey = (oi // 8) * 32 + (oi % 8) + 1
    ocust = torch.randint(1, n_c_total + 1, (no,), generator=g, device=device, dtype=torch.int64)
    odate = torch.randint(days(1992, 1, 1), days(1998, 8, 2) + 1, (no,), generator=g, device=device, dtype=torch.int32)
    oprio = torch.zeros(no, dtype=torch.int32, device=device)
    orders = DeviceTable(pa.schema([("o_orderkey", pa.int64()), ("o_custkey", pa.int64()), ("o_orderdate", pa.date32()), ("o_shippriority", pa.int32())]),
                         no, [u8(okey), u8(ocust), u8(odate), u8(oprio)], [None] * 4, device)
    # lineitem
    per = torch.randint(1, 8, (no,), generator=g, device=device)
    lo = torch.repeat_interleave(torch.arange(no, device=device), per)
    nl = lo.numel()
    lkey = okey[lo]
    qty = torch.randint(1, 51, (nl,), generator=g, device=device, dtype=torch.int64)
    price = qty * torch.randint(90000, 210001, (nl,), generator=g, device=device, dtype=torch.int64)
    disc = torch.randint(0, 11, (nl,), generator=g, device=device, dtype=torch.int64)
    ship = (odate[lo] + torch.randint(1, 122, (nl,), generator=g, device=device, dtype=torch.int32)).to(torch.int32)
    lineitem = DeviceTable(pa.schema([("l_orderkey", pa.int64()), ("l_extendedprice", pa.decimal128(12, 2)), ("l_discount", pa.decimal128(12, 2)),
                                      ("l_shipdate", pa.date32())]), nl, [u8(lkey), dec(price), dec(disc), u8(ship)], [None] * 4, device)
    raw = {"c0": c0, "seg": seg, "o0": o0, "ocust": ocust, "odate": odate, "lo": lo, "price": price, "disc": disc, "ship": ship}
    return customer, orders, lineitem, raw


def q3_torch_reference(n_orders: int, world: int, device="cuda:0", seed: int = 3, top: int = 10):
    """Independent answer for the tables q3_tables_device generates (all ranks' shards regenerated one after the other):
    dense lookups instead of hash joins, exact int64 arithmetic.  Returns (top rows as (orderkey, orderdate days,
    shippriority, unscaled revenue at scale 4), number of result groups)."""
    import torch
    cutoff = days(1995, 3, 15)
    n_c_total = max(1, n_orders // 10)
    building = torch.zeros(n_c_total + 1, dtype=torch.bool, device=device)
    for r in range(world):      # pass 1: which customers are in the BUILDING segment
        _, _, _, raw = q3_tables_device(n_orders, world, r, device, seed)
        nc = raw["seg"].numel()
        building[raw["c0"] + 1: raw["c0"] + nc + 1] = raw["seg"] == SEGMENTS.index(b"BUILDING")
        del raw
    revenue = torch.zeros(n_orders, dtype=torch.int64, device=device)
    hit = torch.zeros(n_orders, dtype=torch.bool, device=device)
    odate_all = torch.zeros(n_orders, dtype=torch.int32, device=device)
    for r in range(world):      # pass 2: one shard resident at a time
        _, _, _, raw = q3_tables_device(n_orders, world, r, device, seed)
        no = raw["ocust"].numel()
        ok_order = building[raw["ocust"]] & (raw["odate"] < cutoff)
        odate_all[raw["o0"]: raw["o0"] + no] = raw["odate"]
        keep = ok_order[raw["lo"]] & (raw["ship"] > cutoff)
        idx = raw["lo"][keep] + raw["o0"]
        revenue.index_add_(0, idx, raw["price"][keep] * (100 - raw["disc"][keep]))
        hit[idx] = True
        del raw, ok_order, keep, idx
    groups = int(hit.sum().item())
    cand = torch.nonzero(hit).reshape(-1)
    rev = revenue[cand]
    k = min(top * 50, cand.numel())
    _, pos = torch.topk(rev, k)
    rows = []
    for p in pos.tolist():
        oi = int(cand[p].item())
        rows.append(((oi // 8) * 32 + (oi % 8) + 1, int(odate_all[oi].item()), 0, int(rev[p].item())))
    rows.sort(key=lambda r: (-r[3], r[1], r[0]))
    return rows[:top], groups


def q1_check_against_torch(out: pa.Table, chk: dict) -> list:
    """Compare a Q1 stage-1 result (Partial states, q1_plan() column order) with independent torch reductions of the generating
    tensors — ALL eight aggregates of every group, exact: count(1), sum_qty, sum_base_price, sum_disc_price = Σ price·(100 − disc)
    (scale 4), sum_charge = Σ price·(100 − disc)·(100 + tax) (scale 6; exceeds int64 at SF100, so the per-row disc_price is split
    into a high and a low 16-bit part summed separately and recombined with Python ints), and the three avg states (sum, count).
    Returns a list of mismatch descriptions (empty = all good)."""
    import torch
    keep = chk["ship"] <= days(1998, 9, 2)
    rows = {(r[0], r[1]): r for r in zip(*[out.column(i).to_pylist() for i in range(out.num_columns)])}
    problems = []
    seen = set()
    for rf in "ANR":
        for ls in "FO":
            m = keep & (chk["rf"] == ord(rf)) & (chk["ls"] == ord(ls))
            cnt = int(m.sum().item())
            if cnt == 0:
                if (rf, ls) in rows:
                    problems.append(f"group {rf}{ls} should not exist")
                continue
            if (rf, ls) not in rows:
                problems.append(f"group {rf}{ls} missing")
                continue
            seen.add((rf, ls))
            r = rows[(rf, ls)]
            mi = m.to(torch.int64)
            sq = int((chk["qty"] * mi).sum().item()) * 100
            sp = int((chk["price"] * mi).sum().item())
            sd = int((chk["disc"] * mi).sum().item())
            dp = chk["price"] * (100 - chk["disc"]) * mi                  # ≤ 1.05e9 per row
            sdp = int(dp.sum().item())
            f = 100 + chk["tax"]
            sch = (int(((dp >> 16) * f).sum().item()) << 16) + int(((dp & 0xFFFF) * f).sum().item())
            del dp, f, mi
            # columns: rf, ls, (sum, is_empty)×4, (sum, count)×3, count
            got = {"sum_qty": int(r[2].scaleb(2)), "sum_base_price": int(r[4].scaleb(2)), "sum_disc_price": int(r[6].scaleb(4)),
                   "sum_charge": int(r[8].scaleb(6)), "avg_qty.sum": int(r[10].scaleb(2)), "avg_qty.count": r[11],
                   "avg_price.sum": int(r[12].scaleb(2)), "avg_price.count": r[13], "avg_disc.sum": int(r[14].scaleb(2)),
                   "avg_disc.count": r[15], "count": r[16], "is_empty": (r[3], r[5], r[7], r[9])}
            want = {"sum_qty": sq, "sum_base_price": sp, "sum_disc_price": sdp, "sum_charge": sch, "avg_qty.sum": sq, "avg_qty.count": cnt,
                    "avg_price.sum": sp, "avg_price.count": cnt, "avg_disc.sum": sd, "avg_disc.count": cnt, "count": cnt,
                    "is_empty": (False, False, False, False)}
            for k in want:
                if got[k] != want[k]:
                    problems.append(f"group {rf}{ls} {k}: got {got[k]}, want {want[k]}")
    for k in rows:
        if k not in seen:
            problems.append(f"unexpected group {k}")
    return problems


def lineitem_q6_device(n: int, device="cuda:0", seed: int = 6):
    """Q6's four lineitem columns generated in HBM (same distributions as lineitem_q6) + the generating tensors."""
    import torch
    from .native import DeviceTable
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def dec(v):
        buf = torch.zeros((n, 2), dtype=torch.int64, device=device)
        buf[:, 0] = v
        return buf.view(torch.uint8).reshape(-1)
    qty = torch.randint(1, 51, (n,), generator=g, device=device, dtype=torch.int64)
    price = qty * torch.randint(90000, 210001, (n,), generator=g, device=device, dtype=torch.int64)
    disc = torch.randint(0, 11, (n,), generator=g, device=device, dtype=torch.int64)
    ship = torch.randint(days(1992, 1, 2), days(1998, 12, 1) + 1, (n,), generator=g, device=device, dtype=torch.int32)
    schema = pa.schema([("l_quantity", pa.decimal128(12, 2)), ("l_extendedprice", pa.decimal128(12, 2)), ("l_discount", pa.decimal128(12, 2)),
                        ("l_shipdate", pa.date32())])
    t = DeviceTable(schema, n, [dec(qty * 100), dec(price), dec(disc), ship.view(torch.uint8).reshape(-1)], [None] * 4, device)
    return t, {"qty": qty, "price": price, "disc": disc, "ship": ship}


def q6_torch_reference(chk: dict) -> int:
    """Unscaled (scale 4) Q6 revenue of the generating tensors, exact int64."""
    m = (chk["ship"] >= days(1994, 1, 1)) & (chk["ship"] < days(1995, 1, 1)) & (chk["disc"] >= 5) & (chk["disc"] <= 7) & (chk["qty"] < 24)
    return int((chk["price"][m] * chk["disc"][m]).sum().item())
