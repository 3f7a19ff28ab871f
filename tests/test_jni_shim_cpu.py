"""The JNI exports (what Spark's Native.scala binds) driven through a JVM-less JNIEnv: argument marshalling,
stream ownership, global-ref balance and the Java exception classes thrown on failure
(native/jni-bridge/src/errors.rs:473-560).  No GPU work happens here: plans fail at planning time."""
import ctypes

import pyarrow as pa
import pytest

from datafusion_comet_amd import native, serde as S, tpch
from tests.jni_mock import Jvm


@pytest.fixture()
def jvm(built):
    return Jvm(ctypes.CDLL(native.LIB_PATH))


def test_jni_symbols_exported(built):
    lib = ctypes.CDLL(native.LIB_PATH)
    for n in ("NativeBase_init", "NativeBase_isFeatureEnabled", "Native_createPlan", "Native_executePlan", "Native_releasePlan",
              "Native_traceBegin", "Native_traceEnd", "Native_logMemoryUsage", "Native_getRustThreadId",
              "Native_writeSortedFileNative", "Native_sortRowPartitionsNative", "Native_decodeShuffleBlock",
              "Native_columnarToRowInit", "Native_columnarToRowConvert", "Native_columnarToRowClose",
              "NativeBase_release", "NativeBase_isObjectStoreSchemeSupported",
              # org.apache.comet.parquet.Native (native/core/src/parquet/mod.rs:135,250,295,318): with these all 21 JNI exports of the reference resolve
              "parquet_Native_initRecordBatchReader", "parquet_Native_readNextRecordBatch", "parquet_Native_currentColumnBatch",
              "parquet_Native_closeRecordBatchReader"):
        assert hasattr(lib, "Java_org_apache_comet_" + n)


def test_out_of_scope_entry_points_throw_instead_of_unsatisfied_link(jvm):
    f = jvm.lib.Java_org_apache_comet_Native_sortRowPartitionsNative
    f.restype = None
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint8]
    f(jvm.env, None, 0, 0, 0)
    cls, msg = jvm.exception()
    assert cls == "org/apache/comet/CometNativeException" and "sortRowPartitionsNative" in msg


def test_create_release_balances_global_refs_and_releases_stream(jvm):
    t = tpch.lineitem_q6(100)
    inp = native.HostInput.from_table(t)
    node = jvm.m.mock_metrics_node()
    h = jvm.create_plan([inp.address], tpch.q6_plan().encode(), metrics_node=node)
    assert h > 0 and jvm.exception() is None
    assert jvm.m.mock_live_global_refs() == 2          # iterator + metrics node
    assert inp._c.release                              # not touched before the first executePlan (jni_api.rs:795-797)
    jvm.release_plan(h)
    assert jvm.m.mock_live_global_refs() == 0
    assert not inp._c.release                          # dropped plan released the ArrowArrayStream (scan.rs:41-44)
    assert jvm.m.mock_metrics_len(node) > 0            # final metrics push (jni_api.rs:961-990)
    raw = ctypes.string_at(jvm.m.mock_metrics_bytes(node), jvm.m.mock_metrics_len(node))
    metrics, children = S.decode_metric_node(raw)
    assert "output_rows" in metrics and len(children) == 1


def test_unsupported_plan_throws_comet_native_exception(jvm):
    t = pa.table({"a": pa.array([1], pa.int32())})
    inp = native.HostInput.from_table(t)
    plan = S.Operator("raw", [S.scan([S.T_INT32])], raw_tag=114)   # Explode
    h = jvm.create_plan([inp.address], plan.encode())
    assert h == 0                                       # JNIDefault zero value (errors.rs:390-450)
    cls, msg = jvm.exception()
    assert cls == "org/apache/comet/CometNativeException" and "Explode" in msg
    assert jvm.m.mock_live_global_refs() == 0
    assert not inp._c.release


def test_non_stream_iterator_is_rejected(jvm):
    h = jvm.create_plan([], tpch.q6_plan().encode(), iterator_objects=[jvm.m.mock_plain_object()])
    assert h == 0
    cls, msg = jvm.exception()
    assert cls == "org/apache/comet/CometNativeException" and "ArrowArrayStream" in msg


def test_execute_with_mismatched_address_arrays(jvm):
    t = tpch.lineitem_q6(10)
    inp = native.HostInput.from_table(t)
    h = jvm.create_plan([inp.address], tpch.q6_plan().encode())
    rows = jvm.execute_plan(h, [1, 2], [1])
    assert rows == 0
    cls, _ = jvm.exception()
    assert cls == "org/apache/comet/CometNativeException"
    jvm.release_plan(h)


def test_invalid_handle(jvm):
    rows = jvm.execute_plan(987654, [], [])
    assert rows == 0 and jvm.exception()[0] == "org/apache/comet/CometNativeException"


def test_decode_shuffle_block_through_jni(jvm):
    """Native.decodeShuffleBlock (jni_api.rs:1163-1181): direct ByteBuffer in, Arrow C Data structs out; errors become
    CometNativeException."""
    import numpy as np
    from oracle import shuffle_oracle as SO
    b = pa.record_batch({"k": pa.array(np.arange(1000), pa.int64()), "s": pa.array([None if i % 5 == 0 else "v%d" % i for i in range(1000)])})
    for codec in (0, 1, 2, 3):
        blk = SO.encode_block(b, codec)[16:]
        arrays = [native.ArrowArrayC() for _ in range(2)]
        schemas = [native.ArrowSchemaC() for _ in range(2)]
        rows = jvm.decode_shuffle_block(blk, [ctypes.addressof(a) for a in arrays], [ctypes.addressof(s) for s in schemas])
        assert rows == 1000, jvm.exception()
        cols = [pa.Array._import_from_c(ctypes.addressof(a), ctypes.addressof(s)) for a, s in zip(arrays, schemas)]
        assert cols[0].equals(b.column(0)) and cols[1].equals(b.column(1))
    a, s = native.ArrowArrayC(), native.ArrowSchemaC()
    assert jvm.decode_shuffle_block(b"BZIP" + b"\0" * 64, [ctypes.addressof(a)], [ctypes.addressof(s)]) == 0
    cls, msg = jvm.exception()
    assert cls == "org/apache/comet/CometNativeException" and "invalid compression codec" in msg


def test_shuffle_block_iterator_is_accepted_for_a_shuffle_scan_leaf(jvm):
    """createPlan binds a CometShuffleBlockIterator (hasNext()I / getBuffer()) to a ShuffleScan leaf; global refs balance."""
    it = jvm.block_iterator([])
    plan = S.filter_(S.shuffle_scan([S.T_INT64]), S.is_not_null(S.col(0, S.T_INT64)))
    h = jvm.create_plan([], plan.encode(), iterator_objects=[it])
    assert h > 0, jvm.exception()
    assert jvm.m.mock_live_global_refs() == 1
    jvm.release_plan(h)
    assert jvm.m.mock_live_global_refs() == 0


def test_task_memory_manager_is_bound_and_released(jvm):
    """createPlan takes a global ref on the CometTaskMemoryManager and binds acquireMemory(J)J / releaseMemory(J)V
    (comet_task_memory_manager.rs:32-60); releasePlan drops it.  An object without those methods is ignored, not an error."""
    mm = jvm.memory_manager(1 << 30)
    inp = native.HostInput.from_table(tpch.lineitem_q6(10))
    h = jvm.create_plan([inp.address], tpch.q6_plan().encode(), memory_manager=mm.handle)
    assert h > 0 and jvm.exception() is None
    assert jvm.m.mock_live_global_refs() == 2          # iterator + memory manager
    stats = (ctypes.c_int64 * 4)(9, 9, 9, 9)
    jvm.lib.comet_plan_memory_stats.argtypes = [ctypes.c_int64, ctypes.c_void_p]
    jvm.lib.comet_plan_memory_stats(h, stats)
    assert list(stats) == [0, 0, 0, 0]                 # nothing staged before the first executePlan
    jvm.release_plan(h)
    assert jvm.m.mock_live_global_refs() == 0 and mm.stats()["used"] == 0
    inp2 = native.HostInput.from_table(tpch.lineitem_q6(10))
    h2 = jvm.create_plan([inp2.address], tpch.q6_plan().encode(), memory_manager=jvm.m.mock_plain_object())
    assert h2 > 0 and jvm.exception() is None and jvm.m.mock_live_global_refs() == 1
    jvm.release_plan(h2)


def test_scalar_subqueries_are_asked_of_the_jvm_at_the_first_execute(jvm):
    """Subquery{id, datatype} (expr.proto:513-516): CometScalarSubquery.isNull / get<Type>(planId, id) with createPlan's plan id (jni-bridge/src/comet_exec.rs:54-126),
    on the first executePlan — not at createPlan (operators.scala registers the subqueries behind the iterator's construction).  No GPU here: the call sequence is
    what is checked; the values' way into the kernels is tests/test_scalar_batch_gpu.py's."""
    import struct
    jvm.m.mock_static_calls_clear()
    t = pa.table({"a": pa.array([1, 2, 3], pa.int64()), "s": pa.array(["x", "y", "z"])})
    inp = native.HostInput.from_table(t)
    D = S.decimal(10, 2)
    plan = S.project(S.filter_(S.scan([S.T_INT64, S.T_STRING]), S.gt(S.col(0, S.T_INT64), S.subquery(5, S.T_INT64))),
                     [S.subquery(6, S.T_DOUBLE), S.subquery(7, D), S.subquery(8, S.T_STRING), S.subquery(9, S.T_INT32), S.subquery(10, S.T_BOOL), S.subquery(11, S.T_DATE)])
    jvm.m.mock_set_subquery(1, 5, 0, 2, 0.0, b"", 0)
    jvm.m.mock_set_subquery(1, 6, 0, 0, 2.5, b"", 0)
    jvm.m.mock_set_subquery(1, 7, 0, 0, 0.0, (12345).to_bytes(2, "big", signed=True), 2)
    jvm.m.mock_set_subquery(1, 8, 0, 0, 0.0, "h\xc3\xa9".encode("latin-1"), 3)
    jvm.m.mock_set_subquery(1, 9, 1, 0, 0.0, b"", 0)
    jvm.m.mock_set_subquery(1, 10, 0, 1, 0.0, b"", 0)
    jvm.m.mock_set_subquery(1, 11, 0, 19000, 0.0, b"", 0)
    h = jvm.create_plan([inp.address], plan.encode())
    assert h > 0 and jvm.exception() is None
    assert jvm.m.mock_static_calls() == b""                 # nothing is asked at createPlan
    arrays = [native.new_arrow_array() for _ in range(7)] if hasattr(native, "new_arrow_array") else None
    jvm.execute_plan(h, [], [])                               # (fails behind the subqueries without a GPU, or on the column count with one: either way they were asked)
    calls = jvm.m.mock_static_calls().decode()
    assert calls == ("isNull(1,5);getLong(1,5);isNull(1,6);getDouble(1,6);isNull(1,7);getDecimal(1,7);isNull(1,8);getString(1,8);isNull(1,9);isNull(1,10);getBoolean(1,10);"
                     "isNull(1,11);getInt(1,11);"), calls
    jvm.m.mock_exception_clear()
    jvm.release_plan(h)
    # a subquery nobody registered: the mock answers isNull = true (like a JVM whose map has no entry would throw: here NULL) — and a plan whose subquery id the
    # C ABI's table does not hold fails by name
    t2 = pa.table({"a": pa.array([1], pa.int64())})
    it = native.CometExecIterator([native.HostInput.from_table(t2)], 1, S.project(S.scan([S.T_INT64]), [S.subquery(3, S.T_INT64)]).encode())
    with pytest.raises(native.CometNativeException, match="Subquery 3 is not registered"):
        next(it)
    it.close()
