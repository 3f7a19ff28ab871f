"""Many plan handles driven concurrently from different threads — what a Spark executor does with its task threads
(SURVEY §8b threading contract, jni_api.rs:133-170): createPlan/executePlan/releasePlan of ONE handle stay on one thread, but
many handles are live at once and share the process-wide plan cache, code-object cache, buffer/stream pools and scan threads."""
import threading

import numpy as np
import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch

pytestmark = pytest.mark.gpu


def _rows(t):
    return sorted(zip(*[t.column(i).to_pylist() for i in range(t.num_columns)]), key=lambda r: tuple((x is None, str(x)) for x in r))


def test_concurrent_tasks_share_caches_and_pools(built, tmp_path):
    from oracle import oracle as O
    import pyarrow.parquet as papq
    q6_t, q1_t = tpch.lineitem_q6(300_000, seed=51), tpch.lineitem_q1(200_000, seed=52)
    customer, orders, lineitem = tpch.q3_tables(8_000, seed=53)
    path = str(tmp_path / "c.parquet")
    papq.write_table(q6_t, path, row_group_size=50_000, compression="snappy", store_decimal_as_integer=True)
    pq_plan = tpch.q6_plan(source=S.native_scan([path], q6_t.schema.names, [tpch.DEC, tpch.DEC, tpch.DEC, S.T_DATE]))
    jobs = {
        "q6_host": (tpch.q6_plan(), [q6_t], tpch.Q6_NUM_OUTPUT_COLS),
        "q6_parquet": (pq_plan, [], tpch.Q6_NUM_OUTPUT_COLS),
        "q1": (tpch.q1_plan(), [q1_t], tpch.Q1_NUM_OUTPUT_COLS),
        "q3": (tpch.q3_plan(), [customer, orders, lineitem], tpch.Q3_NUM_OUTPUT_COLS),
    }
    want = {}
    for name, (plan, tables, ncols) in jobs.items():
        src = tables if len(tables) != 1 else tables[0]
        want[name] = _rows(O.run_plan_to_arrow(S, tpch.q6_plan() if name == "q6_parquet" else plan, q6_t if name == "q6_parquet" else src))
    errors, done = [], []

    def worker(tid):
        try:
            names = list(jobs)
            for it in range(6):
                name = names[(tid + it) % len(names)]
                plan, tables, ncols = jobs[name]
                out = native.execute_to_table([native.HostInput.from_table(t) for t in tables], ncols, plan.encode(), batch_size=0)
                got = _rows(pa.Table.from_batches(out))
                if got != want[name]:
                    errors.append(f"thread {tid} iteration {it}: {name} differs")
                done.append(name)
        except Exception as e:  # pragma: no cover
            errors.append(f"thread {tid}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[:3]
    assert len(done) == 48


def test_interleaved_handles_on_one_thread_and_handles_moved_between_threads(built):
    """What the contract allows beyond "one handle, one thread": a task thread may hold several live handles and step them alternately (a
    stage with two native plans in one task), and Spark may run the calls of ONE handle on different threads over time as long as they do not
    overlap (jni_api.rs:133-170: the context sits behind a pointer, not a thread-local).  Streaming plans, batch by batch."""
    from oracle import oracle as O
    t1, t2 = tpch.lineitem_q6(60_000, seed=61), tpch.lineitem_q6(45_000, seed=62)
    D = tpch.DEC
    filt = S.filter_(S.scan([D, D, D, S.T_DATE]), S.lt(S.col(0, D), S.lit(2400, D)))
    want1, want2 = _rows(O.run_plan_to_arrow(S, filt, t1)), _rows(O.run_plan_to_arrow(S, filt, t2))
    cfg = S.config_map({"spark.comet.gpu.chunkRows": 8192})
    a = native.CometExecIterator([native.HostInput.from_table(t1, 4096)], 4, filt.encode(), batch_size=4096, config=cfg)
    b = native.CometExecIterator([native.HostInput.from_table(t2, 4096)], 4, filt.encode(), batch_size=4096, config=cfg)
    got = {id(a): [], id(b): []}
    live = [a, b]
    k = 0
    while live:                       # alternate between the two handles on this thread
        it = live[k % len(live)]
        batch = native.Native.executePlan(it.handle, 4)
        if batch is None:
            live.remove(it)
        else:
            got[id(it)].append(batch)
        k += 1
    assert _rows(pa.Table.from_batches(got[id(a)])) == want1 and _rows(pa.Table.from_batches(got[id(b)])) == want2
    a.close()
    b.close()
    # one handle, its calls handed from thread to thread (never overlapping)
    c = native.CometExecIterator([native.HostInput.from_table(t1, 4096)], 4, filt.encode(), batch_size=4096, config=cfg)
    out, errs = [], []

    def step():
        try:
            out.append(native.Native.executePlan(c.handle, 4))
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    while not out or out[-1] is not None:
        th = threading.Thread(target=step)
        th.start()
        th.join(120)
        assert not errs, errs
    rel = threading.Thread(target=c.close)
    rel.start()
    rel.join(60)
    assert _rows(pa.Table.from_batches([x for x in out if x is not None])) == want1


def test_failing_and_abandoned_plans_do_not_disturb_their_neighbours(built):
    """A task that fails (ANSI overflow), a task that is released half-way through its stream, and healthy tasks share the pools: the healthy
    ones keep producing the right answer, and nothing the failed ones held is lost to the pools (many rounds)."""
    from oracle import oracle as O
    import decimal
    q6_t = tpch.lineitem_q6(120_000, seed=71)
    want = _rows(O.run_plan_to_arrow(S, tpch.q6_plan(), q6_t))
    D38 = S.decimal(38, 0)
    bad_t = pa.table({"a": pa.array([decimal.Decimal(10**38 - 1)] * 1000, pa.decimal128(38, 0)), "b": pa.array([decimal.Decimal(1)] * 1000, pa.decimal128(38, 0))})
    bad = S.project(S.scan([D38, D38]), [S.math("add", S.col(0, D38), S.col(1, D38), D38, eval_mode=S.ANSI)])
    D = tpch.DEC
    stream = S.filter_(S.scan([D, D, D, S.T_DATE]), S.is_not_null(S.col(0, D)))
    errors = []

    def healthy(tid):
        try:
            for _ in range(10):
                out = native.execute_to_table([native.HostInput.from_table(q6_t)], tpch.Q6_NUM_OUTPUT_COLS, tpch.q6_plan().encode(), batch_size=0)
                if _rows(pa.Table.from_batches(out)) != want:
                    errors.append(f"healthy {tid}: wrong answer")
        except Exception as e:      # noqa: BLE001
            errors.append(f"healthy {tid}: {e!r}")

    def failing(tid):
        for _ in range(10):
            try:
                native.execute_to_table([native.HostInput.from_table(bad_t)], 1, bad.encode(), batch_size=0)
                errors.append(f"failing {tid}: no overflow reported")
            except native.CometQueryExecutionException:
                pass
            except Exception as e:      # noqa: BLE001
                errors.append(f"failing {tid}: {e!r}")

    def abandoning(tid):
        try:
            for _ in range(10):
                it = native.CometExecIterator([native.HostInput.from_table(q6_t, 4096)], 4, stream.encode(), batch_size=4096, config=S.config_map({"spark.comet.gpu.chunkRows": 8192}))
                assert native.Native.executePlan(it.handle, 4) is not None       # one batch, then walk away mid-stream
                it.close()
        except Exception as e:      # noqa: BLE001
            errors.append(f"abandoning {tid}: {e!r}")

    threads = [threading.Thread(target=f, args=(i,)) for i, f in enumerate([healthy, failing, abandoning, healthy, failing, abandoning])]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[:3]
