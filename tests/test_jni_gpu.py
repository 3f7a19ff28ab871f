"""End to end through the JNI exports on the GPU: what CometExecIterator does (createPlan → executePlan until -1 →
releasePlan, CometExecIterator.scala:109,158,236), with a JVM-less JNIEnv standing in for the JVM."""
import ctypes

import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch
from tests.jni_mock import Jvm

pytestmark = pytest.mark.gpu


def test_q6_through_jni_exports_matches_oracle(built):
    from oracle import oracle as O
    jvm = Jvm(native.lib())
    table = tpch.lineitem_q6(300_000, seed=8)
    plan = tpch.q6_plan()
    inp = native.HostInput.from_table(table)
    node = jvm.m.mock_metrics_node()
    h = jvm.create_plan([inp.address], plan.encode(), metrics_node=node, task_attempt_id=0)
    assert h > 0, jvm.exception()
    arrays = [native.ArrowArrayC() for _ in range(2)]
    schemas = [native.ArrowSchemaC() for _ in range(2)]
    rows = jvm.execute_plan(h, [ctypes.addressof(a) for a in arrays], [ctypes.addressof(s) for s in schemas])
    assert rows == 1, jvm.exception()
    cols = [pa.Array._import_from_c(ctypes.addressof(a), ctypes.addressof(s)) for a, s in zip(arrays, schemas)]
    want = O.run_plan_to_arrow(S, plan, table)
    assert cols[0].to_pylist() == want.column(0).to_pylist()
    assert cols[1].to_pylist() == want.column(1).to_pylist()
    arrays2 = [native.ArrowArrayC() for _ in range(2)]
    schemas2 = [native.ArrowSchemaC() for _ in range(2)]
    assert jvm.execute_plan(h, [ctypes.addressof(a) for a in arrays2], [ctypes.addressof(s) for s in schemas2]) == -1
    jvm.release_plan(h)
    metrics, _ = S.decode_metric_node(ctypes.string_at(jvm.m.mock_metrics_bytes(node), jvm.m.mock_metrics_len(node)))
    assert metrics["output_rows"] == 1
    assert jvm.m.mock_live_global_refs() == 0 and not inp._c.release


def test_ansi_overflow_raises_query_execution_exception(built):
    jvm = Jvm(native.lib())
    # decimal(38,0) + decimal(38,0) → decimal(38,0) in ANSI mode overflows for 9…9 + 1 (wide path, eval_mode=ANSI)
    D = S.decimal(38, 0)
    big = 10**38 - 1
    arr = pa.array([__import__("decimal").Decimal(big), __import__("decimal").Decimal(5)], pa.decimal128(38, 0))
    table = pa.table({"a": arr, "b": pa.array([__import__("decimal").Decimal(1)] * 2, pa.decimal128(38, 0))})
    plan = S.project(S.scan([D, D]), [S.math("add", S.col(0, D), S.col(1, D), D, eval_mode=S.ANSI)])
    inp = native.HostInput.from_table(table)
    h = jvm.create_plan([inp.address], plan.encode())
    a, s = native.ArrowArrayC(), native.ArrowSchemaC()
    rows = jvm.execute_plan(h, [ctypes.addressof(a)], [ctypes.addressof(s)])
    assert rows == 0
    cls, msg = jvm.exception()
    assert cls == "org/apache/comet/exceptions/CometQueryExecutionException"
    assert "ARITHMETIC_OVERFLOW" in msg
    jvm.release_plan(h)
    # LEGACY: same data gives NULL for the overflowing row
    plan2 = S.project(S.scan([D, D]), [S.math("add", S.col(0, D), S.col(1, D), D)])
    out = pa.Table.from_batches(native.execute_to_table([native.HostInput.from_table(table)], 1, plan2.encode()))
    assert out.column(0).to_pylist() == [None, __import__("decimal").Decimal(6)]


def test_shuffle_scan_through_jni_block_iterator(built):
    """A reduce-side plan as Spark runs it: iterators[0] is a CometShuffleBlockIterator whose hasNext()/getBuffer() the native
    side calls from executePlan (shuffle_scan.rs:139-171); the blocks here are written by the oracle's (pyarrow) writer."""
    import numpy as np
    from oracle import oracle as O, shuffle_oracle as SO
    jvm = Jvm(native.lib())
    rng = np.random.default_rng(3)
    n = 30_000
    t = pa.table({"k": pa.array(rng.integers(0, 50, n), pa.int64(), mask=rng.random(n) < 0.05),
                  "v": tpch._dec128_array(rng.integers(-10**9, 10**9, n), 12, 2)})
    fields = [S.T_INT64, S.decimal(12, 2)]
    blocks = [SO.encode_block(b, c)[16:] for c, b in zip([1, 2, 3, 0, 1, 2, 3, 0], t.to_batches(max_chunksize=4000))]
    plan = S.hash_agg(S.shuffle_scan(fields), [S.col(0, S.T_INT64)], [S.sum_(S.col(1, fields[1]), S.decimal(22, 2)), S.count(S.col(1, fields[1]))], S.PARTIAL)
    h = jvm.create_plan([], plan.encode(), iterator_objects=[jvm.block_iterator(blocks)], batch_size=0)
    assert h > 0, jvm.exception()
    arrays = [native.ArrowArrayC() for _ in range(4)]
    schemas = [native.ArrowSchemaC() for _ in range(4)]
    rows = jvm.execute_plan(h, [ctypes.addressof(a) for a in arrays], [ctypes.addressof(s) for s in schemas])
    assert rows == 51, jvm.exception()
    cols = [pa.Array._import_from_c(ctypes.addressof(a), ctypes.addressof(s)) for a, s in zip(arrays, schemas)]
    want = O.run_plan_to_arrow(S, plan, [t])
    key = lambda r: (r[0] is None, r[0] or 0)
    got_rows = sorted(zip(*[c.to_pylist() for c in cols]), key=key)
    want_rows = sorted(zip(*[want.column(i).to_pylist() for i in range(4)]), key=key)
    assert got_rows == want_rows
    jvm.release_plan(h)
    assert jvm.m.mock_live_global_refs() == 0


def test_metrics_are_pushed_periodically_while_batches_flow(built):
    """createPlan's metricsUpdateInterval (jni_api.rs:897-909): CometMetricNode.set_all_from_bytes is called while the stream is being
    drained — at most once per interval — and once more at the end, not only at release."""
    import time
    import numpy as np
    jvm = Jvm(native.lib())
    n = 100_000
    table = pa.table({"a": pa.array(np.arange(n, dtype=np.int64))})
    plan = S.filter_(S.scan([S.T_INT64]), S.gt_eq(S.col(0, S.T_INT64), S.lit(0, S.T_INT64)))
    inp = native.HostInput.from_table(table)
    node = jvm.m.mock_metrics_node()
    h = jvm.create_plan([inp.address], plan.encode(), metrics_node=node, batch_size=8192, metrics_interval_ms=2)
    assert h > 0, jvm.exception()
    batches = total = 0
    while True:
        arrays, schemas = [native.ArrowArrayC()], [native.ArrowSchemaC()]
        rows = jvm.execute_plan(h, [ctypes.addressof(arrays[0])], [ctypes.addressof(schemas[0])])
        if rows == -1:
            break
        assert rows > 0, jvm.exception()
        total += pa.Array._import_from_c(ctypes.addressof(arrays[0]), ctypes.addressof(schemas[0])).__len__()
        batches += 1
        if batches == 3:
            mid_stream = jvm.m.mock_metrics_pushes(node)
        time.sleep(0.004)
    assert total == n and batches >= 10
    assert mid_stream >= 1                                   # pushed before the stream ended
    assert 2 <= jvm.m.mock_metrics_pushes(node) <= batches + 1
    jvm.release_plan(h)
    metrics, _ = S.decode_metric_node(ctypes.string_at(jvm.m.mock_metrics_bytes(node), jvm.m.mock_metrics_len(node)))
    assert metrics["output_rows"] == n
