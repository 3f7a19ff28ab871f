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


def test_parquet_native_reader_reads_a_reference_fixture_batch_by_batch(built, tmp_path):
    """org.apache.comet.parquet.Native through its JNI exports (parquet/mod.rs:135-330): initRecordBatchReader with Arrow IPC schema bytes,
    byte ranges and a pushed filter, readNextRecordBatch until 0, currentColumnBatch moving every column of a batch, closeRecordBatchReader —
    over one of the reference's own fixture files and over a larger multi-row-group file."""
    import os
    import numpy as np
    import pyarrow.parquet as papq
    jvm = Jvm(native.lib())
    L = jvm.lib
    init = L.Java_org_apache_comet_parquet_Native_initRecordBatchReader
    init.restype = ctypes.c_int64
    init.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] + [ctypes.c_void_p] * 6 + [ctypes.c_int32, ctypes.c_uint8, ctypes.c_uint8] + [ctypes.c_void_p] * 3
    nxt = L.Java_org_apache_comet_parquet_Native_readNextRecordBatch
    nxt.restype = ctypes.c_int32
    nxt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    col = L.Java_org_apache_comet_parquet_Native_currentColumnBatch
    col.restype = None
    col.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64]
    close = L.Java_org_apache_comet_parquet_Native_closeRecordBatchReader
    close.restype = None
    close.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]

    def read_all(path, required: pa.Schema, data: pa.Schema, ranges, batch_size, flt=None):
        size = os.path.getsize(path)
        st = (ctypes.c_int64 * len(ranges))(*[r[0] for r in ranges])
        ln = (ctypes.c_int64 * len(ranges))(*[r[1] for r in ranges])
        rs, ds = required.serialize().to_pybytes(), data.serialize().to_pybytes()
        fb = flt.encode() if flt is not None else None
        h = init(jvm.env, None, jvm.m.mock_string(("file://" + path).encode()), size, jvm.m.mock_longs(st, len(ranges)), jvm.m.mock_longs(ln, len(ranges)),
                 jvm.m.mock_bytes(fb, len(fb)) if fb else None, jvm.m.mock_bytes(rs, len(rs)), jvm.m.mock_bytes(ds, len(ds)), jvm.m.mock_string(b"UTC"),
                 batch_size, 1, 0, None, None, None)
        assert h > 0, jvm.exception()
        batches = []
        while True:
            rows = nxt(jvm.env, None, h)
            assert not jvm.m.mock_exception_pending(), jvm.exception()
            if rows == 0:
                break
            cols = []
            for i in range(len(required)):
                a, s = native.ArrowArrayC(), native.ArrowSchemaC()
                col(jvm.env, None, h, i, ctypes.addressof(a), ctypes.addressof(s))
                assert not jvm.m.mock_exception_pending(), jvm.exception()
                cols.append(pa.Array._import_from_c(ctypes.addressof(a), ctypes.addressof(s)))
                assert len(cols[-1]) == rows
            batches.append(cols)
        # taking a column when no batch is current is the reference's "There is no more data to read"
        a, s = native.ArrowArrayC(), native.ArrowSchemaC()
        col(jvm.env, None, h, 0, ctypes.addressof(a), ctypes.addressof(s))
        exc = jvm.exception()
        assert exc is not None and "no more data" in exc[1]
        close(jvm.env, None, h)
        return batches

    fx = os.path.join(os.path.dirname(__file__), "golden", "parquet", "decimal32-written-as-64-bit-dict.snappy.parquet")
    want = papq.read_table(fx)
    got = read_all(fx, want.schema, want.schema, [(0, os.path.getsize(fx))], 500)
    assert len(got) == 5 and [len(b[0]) for b in got] == [500, 500, 500, 500, 48]
    assert pa.concat_arrays([b[0] for b in got]).to_pylist() == want.column(0).to_pylist()

    rng = np.random.default_rng(9)
    n = 50_000
    t = pa.table({"k": pa.array(np.arange(n, dtype=np.int64)), "v": pa.array(rng.standard_normal(n)), "s": pa.array([f"r{i % 97}" for i in range(n)])})
    path = str(tmp_path / "rg.parquet")
    papq.write_table(t, path, row_group_size=10_000)
    required = pa.schema([t.schema.field("s"), t.schema.field("k")])              # a projection in another order than the file
    got = read_all(path, required, t.schema, [(0, os.path.getsize(path))], 8192)
    assert pa.concat_arrays([b[0] for b in got]).to_pylist() == t.column("s").to_pylist()
    assert pa.concat_arrays([b[1] for b in got]).to_pylist() == t.column("k").to_pylist()
    # a pushed filter bound to the DATA schema (k is column 0 there, column 1 in the required schema) prunes whole row groups
    flt = S.gt_eq(S.col(0, S.T_INT64), S.lit(30_000, S.T_INT64))
    got = read_all(path, required, t.schema, [(0, os.path.getsize(path))], 8192, flt)
    assert pa.concat_arrays([b[1] for b in got]).to_pylist() == list(range(30_000, n))
    # byte ranges: the first half of the file selects the row groups whose midpoint lies there
    half = read_all(path, required, t.schema, [(0, os.path.getsize(path) // 2)], 8192)
    ks = pa.concat_arrays([b[1] for b in half]).to_pylist()
    assert 0 < len(ks) < n and len(ks) % 10_000 == 0 and ks == list(range(len(ks)))
    assert jvm.m.mock_live_global_refs() == 0
    # a file that is not Parquet: the format error surfaces as the reference's ParquetRuntimeException (errors.rs:333-336)
    junk = str(tmp_path / "junk.parquet")
    with open(junk, "wb") as f:
        f.write(b"PAR1" + b"\x00" * 64 + (1 << 20).to_bytes(4, "little") + b"PAR1")
    rs = required.serialize().to_pybytes()
    h = init(jvm.env, None, jvm.m.mock_string(("file://" + junk).encode()), os.path.getsize(junk), jvm.m.mock_longs((ctypes.c_int64 * 1)(0), 1),
             jvm.m.mock_longs((ctypes.c_int64 * 1)(os.path.getsize(junk)), 1), None, jvm.m.mock_bytes(rs, len(rs)), jvm.m.mock_bytes(rs, len(rs)), jvm.m.mock_string(b"UTC"),
             8192, 1, 0, None, None, None)
    rows = nxt(jvm.env, None, h) if h > 0 else 0
    cls, msg = jvm.exception()
    assert rows == 0 and cls == "org/apache/comet/ParquetRuntimeException" and "parquet:" in msg
    jvm.m.mock_exception_clear()
    if h > 0:
        close(jvm.env, None, h)


def test_pinned_staging_is_charged_to_the_task_memory_manager(built):
    """createPlan's taskMemoryManager (unified_pool.rs:64-150): every growth of the plan's pinned host staging goes through
    CometTaskMemoryManager.acquireMemory on the task thread, everything is handed back by releasePlan, and a manager that grants less than
    asked fails the task with the reference's message (after the partial grant is released)."""
    jvm = Jvm(native.lib())
    table = tpch.lineitem_q6(400_000, seed=9)
    plan = tpch.q6_plan()

    def run(mm):
        inp = native.HostInput.from_table(table)
        h = jvm.create_plan([inp.address], plan.encode(), task_attempt_id=77, memory_manager=mm.handle)
        assert h > 0, jvm.exception()
        arrays = [native.ArrowArrayC() for _ in range(2)]
        schemas = [native.ArrowSchemaC() for _ in range(2)]
        rows = jvm.execute_plan(h, [ctypes.addressof(a) for a in arrays], [ctypes.addressof(s) for s in schemas])
        stats = (ctypes.c_int64 * 4)()
        native.lib().comet_plan_memory_stats(h, stats)
        during = mm.stats()
        jvm.release_plan(h)
        return rows, during, list(stats)

    roomy = jvm.memory_manager(1 << 40)
    rows, during, stats = run(roomy)
    assert rows == 1, jvm.exception()
    assert during["acquires"] > 0 and during["peak"] >= 400_000 * 8          # at least one column of the chunk staged in pinned memory
    assert stats[1] == during["peak"] and stats[3] > 0                       # the plan's own counters agree with what Spark granted
    after = roomy.stats()
    assert after["used"] == 0 and after["releases"] > 0 and after["refused"] == 0

    tight = jvm.memory_manager(64 << 10)                                     # 64 KiB: the first staging block does not fit
    rows, during, _ = run(tight)
    assert rows == 0
    cls, msg = jvm.exception()
    assert cls == "org/apache/comet/CometNativeException"
    assert "Task 77 failed to acquire" in msg and "only got" in msg
    jvm.m.mock_exception_clear()
    assert tight.stats()["used"] == 0 and tight.stats()["refused"] > 0       # the partial grant went back; nothing leaks
    assert jvm.m.mock_live_global_refs() == 0


def test_hbm_budget_of_a_plan(built):
    """spark.comet.gpu.memory.limit: HBM is not Spark's to grant, so a plan holds it against its own budget and fails loudly over it;
    comet_plan_memory_stats reports what a run peaked at."""
    table = tpch.lineitem_q1(300_000, seed=5)
    plan = tpch.q1_plan().encode()
    with pytest.raises(native.CometNativeException, match="GPU memory budget exceeded"):
        native.execute_to_table([native.HostInput.from_table(table)], tpch.Q1_NUM_OUTPUT_COLS, plan, config=S.config_map({"spark.comet.gpu.memory.limit": str(1 << 20)}))
    it = native.CometExecIterator([native.HostInput.from_table(table)], tpch.Q1_NUM_OUTPUT_COLS, plan, config=S.config_map({"spark.comet.gpu.memory.limit": str(8 << 30)}))
    assert native.Native.executePlan(it.handle, tpch.Q1_NUM_OUTPUT_COLS).num_rows == 4
    stats = (ctypes.c_int64 * 4)()
    native.lib().comet_plan_memory_stats(it.handle, stats)
    assert 0 < stats[3] < (8 << 30) and stats[1] > 0
    it.close()
