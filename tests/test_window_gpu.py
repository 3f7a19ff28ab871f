"""Window operator (110): ranking functions, ntile, lag and lead over input sorted by (partition keys, order keys) — the Sort Spark plans
below every Window.  Flags on adjacent rows' order-preserving key bytes, two prefix sums and closed forms per function (window_kernels.hip)
against the oracle's row-by-row evaluation.  Ties: Sort is not stable, so only tie-invariant functions are compared on tied keys."""
import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu

D = S.decimal(12, 2)
FIELDS = [S.T_STRING, S.T_INT32, D, S.T_INT64, S.T_STRING, S.T_DOUBLE]


def _table(n, seed, unique_order):
    rng = np.random.default_rng(seed)
    cat = [None if rng.random() < 0.03 else ["Books", "Music", "Shoes", "Home"][int(i)] for i in rng.integers(0, 4, n)]
    store = rng.integers(0, 5, n).astype(np.int32)
    amount = rng.integers(0, 60, n) * 100 if not unique_order else rng.permutation(n) * 7 - 1000
    return pa.table({"cat": pa.array(cat, pa.string()), "store": pa.array(store, mask=rng.random(n) < 0.03),
                     "amount": tpch._dec128_array(np.asarray(amount, dtype=np.int64), 12, 2) if unique_order else
                     pa.array([None if rng.random() < 0.05 else __import__("decimal").Decimal(int(a)).scaleb(-2) for a in amount], pa.decimal128(12, 2)),
                     "id": pa.array(np.arange(n, dtype=np.int64)),
                     "label": pa.array([None if rng.random() < 0.1 else "label-%d-with-a-long-tail" % int(i) for i in rng.integers(0, 500, n)]),
                     "f": pa.array(rng.random(n), mask=rng.random(n) < 0.1)})


def _plan(fns):
    cat, store, amount = S.col(0, S.T_STRING), S.col(1, S.T_INT32), S.col(2, D)
    order = [(amount, True, True)]
    sorted_child = S.sort(S.scan(FIELDS), [(cat, False, False), (store, False, False)] + order)
    return S.window(sorted_child, [cat, store], order, fns)


def _rows(tb):
    return sorted(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]), key=lambda r: r[3])


def test_ranking_with_ties(built):
    from oracle import oracle as O
    t = _table(40_000, 7, unique_order=False)
    fns = [("rank", [], S.T_INT32), ("dense_rank", [], S.T_INT32), ("percent_rank", [], S.T_DOUBLE), ("cume_dist", [], S.T_DOUBLE)]
    plan = _plan(fns)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 10, plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, [t])
    assert got.schema.types == want.schema.types
    assert _rows(got) == _rows(want)
    assert max(got.column(6).to_pylist()) > 50 and max(got.column(7).to_pylist()) > 20


def test_all_functions_on_unique_order_keys(built):
    from oracle import oracle as O
    t = _table(30_000, 11, unique_order=True)
    fns = [("row_number", [], S.T_INT32), ("rank", [], S.T_INT32), ("ntile", [S.lit(4, S.T_INT32)], S.T_INT32),
           ("lag", [S.col(2, D), S.lit(1, S.T_INT32), S.lit(None, D)], D), ("lead", [S.col(4, S.T_STRING), S.lit(2, S.T_INT32), S.lit(None, S.T_STRING)], S.T_STRING),
           ("lag", [S.col(5, S.T_DOUBLE), S.lit(3, S.T_INT32), S.lit(None, S.T_DOUBLE)], S.T_DOUBLE), ("lead", [S.col(3, S.T_INT64)], S.T_INT64),
           ("lag", [S.col(2, D), S.lit(2, S.T_INT32), S.lit(__import__("decimal").Decimal("-1.50"), D)], D), ("lead", [S.col(5, S.T_DOUBLE), S.lit(1, S.T_INT32), S.lit(0.25, S.T_DOUBLE)], S.T_DOUBLE)]
    plan = _plan(fns)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 15, plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, [t])
    assert got.schema.types == want.schema.types
    assert _rows(got) == _rows(want)
    # Window below other operators: top 3 per (cat, store) by amount — filter on the rank column above the Window
    top = S.project(S.filter_(plan, S.lt_eq(S.col(6, S.T_INT32), S.lit(3, S.T_INT32))), [S.col(0, S.T_STRING), S.col(1, S.T_INT32), S.col(3, S.T_INT64), S.col(6, S.T_INT32)])
    got2 = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 4, top.encode(), batch_size=0))
    want2 = O.run_plan_to_arrow(S, top, [t])
    key = lambda tb: sorted(zip(*[tb.column(i).to_pylist() for i in range(4)]), key=lambda r: r[2])
    assert key(got2) == key(want2) and got2.num_rows <= 3 * 5 * 6


def test_window_without_partition_or_order(built):
    from oracle import oracle as O
    t = _table(5_000, 3, unique_order=True)
    plan = S.window(S.sort(S.scan(FIELDS), [(S.col(3, S.T_INT64), False, False)]), [], [(S.col(3, S.T_INT64), False, False)],
                    [("row_number", [], S.T_INT32), ("lag", [S.col(3, S.T_INT64), S.lit(1, S.T_INT32)], S.T_INT64)])
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 8, plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, [t])
    assert _rows(got) == _rows(want)
    assert got.column(6).to_pylist() == list(range(1, 5001))


def test_aggregates_over_partition_and_running_frames(built):
    """SUM / COUNT / AVG OVER (PARTITION BY … [ORDER BY …]) — TPC-DS q12 / q20 / q98's sum(…) over (partition by i_class), q47 / q57 / q89's
    avg(…) over a partition, q51's running sums: whole-partition frames, ROWS … CURRENT ROW and RANGE … CURRENT ROW (peers included),
    computed from one 128-bit prefix sum per argument column; NULL arguments, empty frames, decimal precision overflow → NULL."""
    from oracle import oracle as O
    import decimal
    t = _table(30_000, 13, unique_order=False)
    big = pa.array([None if i % 97 == 0 else decimal.Decimal(10**37 // 3 + i).scaleb(-2) for i in range(t.num_rows)], pa.decimal128(38, 2))
    t = t.append_column("big", big)
    fields = FIELDS + [S.decimal(38, 2)]
    cat, store, amount = S.col(0, S.T_STRING), S.col(1, S.T_INT32), S.col(2, D)
    order = [(amount, True, True)]
    child = S.sort(S.scan(fields), [(cat, False, False), (store, False, False)] + order)
    SD, AD = S.decimal(22, 2), S.decimal(16, 6)
    whole, rows_cur, range_cur = ("rows", "unbounded", "unbounded"), ("rows", "unbounded", "current"), ("range", "unbounded", "current")
    fns = [("agg", S.sum_(amount, SD), SD, whole), ("agg", S.sum_(amount, SD), SD, range_cur), ("agg", S.count(amount), S.T_INT64, whole),
           ("agg", S.count(S.lit(1, S.T_INT32)), S.T_INT64, range_cur), ("agg", S.avg(amount, AD, SD), AD, whole), ("agg", S.avg(amount, AD, SD), AD, range_cur),
           ("agg", S.sum_(S.col(3, S.T_INT64), S.T_INT64), S.T_INT64, whole), ("agg", S.sum_(S.col(1, S.T_INT32), S.T_INT64), S.T_INT64, range_cur),
           ("agg", S.sum_(S.col(6, S.decimal(38, 2)), S.decimal(38, 2)), S.decimal(38, 2), whole), ("rank", [], S.T_INT32)]
    plan = S.window(child, [cat, store], order, fns)
    ncols = len(fields) + len(fns)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], ncols, plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, [t])
    assert got.schema.types == want.schema.types
    assert _rows(got) == _rows(want)
    assert got.column(len(fields) + 8).null_count > 0          # the decimal(38,2) partition sums overflow their precision → NULL
    # ROWS … CURRENT ROW depends on the position among peers: checked on unique order keys
    t2 = _table(20_000, 14, unique_order=True)
    child2 = S.sort(S.scan(FIELDS), [(cat, False, False), (store, False, False)] + order)
    plan2 = S.window(child2, [cat, store], order, [("agg", S.sum_(amount, SD), SD, rows_cur), ("agg", S.count(S.col(5, S.T_DOUBLE)), S.T_INT64, rows_cur), ("row_number", [], S.T_INT32)])
    got2 = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t2)], 9, plan2.encode(), batch_size=0))
    want2 = O.run_plan_to_arrow(S, plan2, [t2])
    assert _rows(got2) == _rows(want2)


def test_sliding_frames_and_min_max(built):
    """ROWS frames with literal offsets on either side (n PRECEDING / n FOLLOWING, both bounds on one side of the row, frames that fall off the
    partition and come back empty), frames that start at the current row, and MIN / MAX over every frame shape (planner.rs:2953-3100): sums
    and counts from prefix-sum differences, extremes from running extremes per partition (frames touching a partition edge) or a walk of the
    frame (bounded on both sides).  Unique order keys: ROWS frames depend on the position among peers."""
    from oracle import oracle as O
    t = _table(25_000, 15, unique_order=True)
    t = t.set_column(2, "amount", pa.array([None if i % 11 == 0 else v for i, v in enumerate(t.column(2).to_pylist())], pa.decimal128(12, 2)))
    # order by the id-like unique column instead of amount (which now has NULLs): keep amount as the argument
    cat, store, amount, ident = S.col(0, S.T_STRING), S.col(1, S.T_INT32), S.col(2, D), S.col(3, S.T_INT64)
    order = [(ident, True, True)]
    child = S.sort(S.scan(FIELDS), [(cat, False, False), (store, False, False)] + order)
    SD, AD = S.decimal(22, 2), S.decimal(16, 6)
    frames = [("rows", -2, 2), ("rows", -3, "current"), ("rows", "current", 4), ("rows", -5, -2), ("rows", 1, 3), ("rows", "unbounded", 1), ("rows", -1, "unbounded"),
              ("rows", "current", "unbounded"), ("rows", "current", "current"), ("range", "current", "unbounded"), ("rows", -2000, 2000)]
    fns = []
    for fr in frames:
        fns += [("agg", S.sum_(amount, SD), SD, fr), ("agg", S.count(amount), S.T_INT64, fr), ("agg", S.min_(amount, D), D, fr), ("agg", S.max_(ident, S.T_INT64), S.T_INT64, fr)]
    fns += [("agg", S.avg(amount, AD, SD), AD, ("rows", -2, 2)), ("agg", S.min_(S.col(1, S.T_INT32), S.T_INT32), S.T_INT32, ("rows", "unbounded", "unbounded")),
            ("agg", S.max_(amount, D), D, ("range", "unbounded", "current"))]
    plan = S.window(child, [cat, store], order, fns)
    ncols = len(FIELDS) + len(fns)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], ncols, plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, [t])
    assert got.schema.types == want.schema.types
    assert _rows(got) == _rows(want)
    assert got.column(len(FIELDS) + 4 * 3).null_count > 0        # SUM over (5 PRECEDING, 2 PRECEDING): empty at the head of every partition → NULL


@pytest.mark.parametrize("desc,nulls_last", [(False, False), (True, True), (False, True)])
def test_range_frames_with_value_offsets(built, desc, nulls_last):
    """RANGE BETWEEN a PRECEDING AND b FOLLOWING over one integer ORDER BY key (planner.rs:3031-3037, 3090-3096; the JVM side sends magnitudes,
    CometWindowExec.scala:588-632): the frame holds the partition's rows whose key lies within [key − a, key + b] in sort order — ties are
    peers, so every function is tie invariant; NULL keys frame their NULL peers; an offset on one side combines with UNBOUNDED / CURRENT ROW
    on the other; keys near the type's limits wrap in the key's own width like the reference's ScalarValue arithmetic."""
    from oracle import oracle as O
    n = 3000                                                                  # ~100 rows per partition: the oracle scans a partition per row and function
    rng = np.random.default_rng(31 + int(desc) + 2 * int(nulls_last))
    day = rng.integers(0, 60, n).astype(np.int32)
    day[:8] = np.int32(2**31 - 1) - np.arange(8, dtype=np.int32)               # day + 10 wraps for these
    t = pa.table({"g": pa.array(rng.integers(0, 30, n).astype(np.int32)), "day": pa.array(day, mask=rng.random(n) < 0.04),
                  "amount": tpch._dec128_array(rng.integers(-5000, 5000, n), 12, 2), "id": pa.array(np.arange(n, dtype=np.int64)),
                  "small": pa.array(rng.integers(-120, 120, n).astype(np.int8))})
    fields = [S.T_INT32, S.T_INT32, D, S.T_INT64, S.T_INT8]
    g, dayc, amount, ident = S.col(0, S.T_INT32), S.col(1, S.T_INT32), S.col(2, D), S.col(3, S.T_INT64)
    order = [(dayc, desc, nulls_last)]
    child = S.sort(S.scan(fields), [(g, False, False)] + order)
    v = lambda k: ("value", S.lit(k, S.T_INT32))
    frames = [("range", v(3), v(3)), ("range", v(10), "current"), ("range", "current", v(7)), ("range", "unbounded", v(2)), ("range", v(0), "unbounded"), ("range", v(0), v(0)),
              ("range", v(40), v(40))]
    SD = S.decimal(22, 2)
    fns = []
    for fr in frames:
        fns += [("agg", S.sum_(amount, SD), SD, fr), ("agg", S.count(amount), S.T_INT64, fr), ("agg", S.min_(amount, D), D, fr), ("agg", S.max_(ident, S.T_INT64), S.T_INT64, fr)]
    plan = S.window(child, [g], order, fns)
    ncols = len(fields) + len(fns)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], ncols, plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, [t])
    assert got.schema.types == want.schema.types
    assert _rows(got) == _rows(want)
    # an int8 key: the target wraps at 8 bits (−120 − 20 → +116), whatever the row's position
    small = S.col(4, S.T_INT8)
    order8 = [(small, desc, nulls_last)]
    plan8 = S.window(S.sort(S.scan(fields), [(g, False, False)] + order8), [g], order8,
                     [("agg", S.count(amount), S.T_INT64, ("range", ("value", S.lit(20, S.T_INT8)), ("value", S.lit(5, S.T_INT8))))])
    got8 = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], len(fields) + 1, plan8.encode(), batch_size=0))
    assert _rows(got8) == _rows(O.run_plan_to_arrow(S, plan8, [t]))


def test_first_last_and_nth_value(built):
    """FIRST_VALUE / LAST_VALUE (the First / Last aggregates over a frame, planner.rs:3243-3251) and nth_value (CometWindowExec.scala:293-306),
    respecting and ignoring NULLs, over ROWS frames, frames touching the partition edges and a RANGE frame with value offsets; lag / lead
    IGNORE NULLS; decimal,
    string and integer arguments.  Unique order keys: which row is first depends on the position among peers."""
    from oracle import oracle as O
    n = 4000
    rng = np.random.default_rng(91)
    t = pa.table({"g": pa.array(rng.integers(0, 40, n).astype(np.int32)), "k": pa.array(rng.permutation(n).astype(np.int32) * 3),
                  "amount": pa.array([None if rng.random() < 0.3 else __import__("decimal").Decimal(int(a)).scaleb(-2) for a in rng.integers(-9000, 9000, n)], pa.decimal128(12, 2)),
                  "label": pa.array([None if rng.random() < 0.25 else "label-%d-with-a-long-tail" % int(i) for i in rng.integers(0, 500, n)]),
                  "id": pa.array(np.arange(n, dtype=np.int64))})
    fields = [S.T_INT32, S.T_INT32, D, S.T_STRING, S.T_INT64]
    g, k, amount, label, ident = S.col(0, S.T_INT32), S.col(1, S.T_INT32), S.col(2, D), S.col(3, S.T_STRING), S.col(4, S.T_INT64)
    order = [(k, False, False)]
    child = S.sort(S.scan(fields), [(g, False, False)] + order)
    v = lambda x: ("value", S.lit(x, S.T_INT32))
    frames = [("rows", -2, 2), ("rows", "unbounded", "current"), ("rows", "current", "unbounded"), ("rows", 1, 3), ("range", v(30), v(60)), ("rows", "unbounded", "unbounded")]
    fns = []
    for fr in frames:
        for ign in (False, True):
            fns += [("agg", S.first_(amount, D, ign), D, fr), ("agg", S.last_(label, S.T_STRING, ign), S.T_STRING, fr), ("agg", S.last_(ident, S.T_INT64, ign), S.T_INT64, fr),
                    ("nth_value", [amount, S.lit(2, S.T_INT64)], D, fr, ign), ("nth_value", [label, S.lit(3, S.T_INT64)], S.T_STRING, fr, ign)]
    # lag / lead IGNORE NULLS: the k-th non-NULL row before / after, with and without a default
    whole = ("rows", "unbounded", "unbounded")
    fns += [("lag", [amount, S.lit(1, S.T_INT32)], D, whole, True), ("lead", [amount, S.lit(2, S.T_INT32), S.lit(__import__("decimal").Decimal("-1.00"), D)], D, whole, True),
            ("lag", [label, S.lit(3, S.T_INT32)], S.T_STRING, whole, True), ("lead", [amount, S.lit(1, S.T_INT32)], D, whole, False),
            # a negative offset looks the other way, IGNORE NULLS included (it used to fall through to the plain offset path)
            ("lag", [amount, S.lit(-2, S.T_INT32)], D, whole, True), ("lead", [label, S.lit(-1, S.T_INT32)], S.T_STRING, whole, True)]
    plan = S.window(child, [g], order, fns)
    ncols = len(fields) + len(fns)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], ncols, plan.encode(), batch_size=0))
    want = O.run_plan_to_arrow(S, plan, [t])
    assert got.schema.types == want.schema.types
    key = lambda tb: sorted(zip(*[tb.column(i).to_pylist() for i in range(tb.num_columns)]), key=lambda r: r[4])
    assert key(got) == key(want)
    # IGNORE NULLS changes answers on this data (a third of the amounts are NULL)
    assert got.column(len(fields)).to_pylist() != got.column(len(fields) + 5).to_pylist()


def test_frames_the_engine_refuses(built):
    t = _table(100, 16, unique_order=True)
    cat, store, amount = S.col(0, S.T_STRING), S.col(1, S.T_INT32), S.col(2, D)
    order = [(amount, True, True)]
    child = S.sort(S.scan(FIELDS), [(cat, False, False), (store, False, False)] + order)
    for fn, msg in ((("agg", S.sum_(amount, S.decimal(22, 2)), S.decimal(22, 2), ("range", -2, "current")), "RANGE frames with a value offset"),
                    (("agg", S.min_(amount, D), D, ("rows", -5000, 5000)), "wider than 4096"),
                    (("lag", [amount, S.lit(0, S.T_INT32)], D, ("rows", "unbounded", "unbounded"), True), "offset 0"),
                    (("agg", S.min_(S.col(5, S.T_DOUBLE), S.T_DOUBLE), S.T_DOUBLE, ("rows", "unbounded", "current")), "not supported yet")):
        with pytest.raises(native.CometNativeException, match=msg):
            native.execute_to_table([native.HostInput.from_table(t)], len(FIELDS) + 1, S.window(child, [cat, store], order, [fn]).encode(), batch_size=0)


def test_range_offset_wider_than_the_order_key_is_refused(built):
    """key ± offset is evaluated in the key's width, so an Int64 literal of 2^32 over an Int32 ORDER BY key would wrap to offset 0
    instead of covering the partition — refused at createPlan; the largest offset the key type holds still runs."""
    t = pa.table({"g": pa.array(np.zeros(50, np.int32)), "k": pa.array(np.arange(50, dtype=np.int32)), "v": pa.array(np.arange(50, dtype=np.int64))})
    g, k, v = S.col(0, S.T_INT32), S.col(1, S.T_INT32), S.col(2, S.T_INT64)
    order = [(k, False, False)]
    child = S.sort(S.scan([S.T_INT32, S.T_INT32, S.T_INT64]), [(g, False, False)] + order)

    def plan(lit):
        return S.window(child, [g], order, [("agg", S.count(v), S.T_INT64, ("range", ("value", lit), "current"))])
    with pytest.raises(native.CometNativeException, match="does not fit the ORDER BY column's type"):
        native.execute_to_table([native.HostInput.from_table(t)], 4, plan(S.lit(1 << 32, S.T_INT64)).encode(), batch_size=0)
    got = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(t)], 4, plan(S.lit((1 << 31) - 1, S.T_INT64)).encode(), batch_size=0))
    assert sorted(got.column(3).to_pylist()) == list(range(1, 51))
