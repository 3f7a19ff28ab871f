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
