"""ctypes driver for the JVM-less JNIEnv mock (tests/jni_mock/jni_mock.c) — calls libcomet's
Java_org_apache_comet_Native_* exports exactly as a JVM would."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libjnimock.so")


def load():
    src = os.path.join(_HERE, "jni_mock.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-Wall", "-o", _SO, src])
    m = ctypes.CDLL(_SO)
    vp = ctypes.c_void_p
    m.mock_env.restype = vp
    for f in ("mock_bytes", "mock_longs", "mock_objs", "mock_stream", "mock_plain_object", "mock_metrics_node", "mock_string", "mock_block_iterator", "mock_ints", "mock_array_data"):
        getattr(m, f).restype = vp
    m.mock_bytes.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    m.mock_longs.argtypes = [vp, ctypes.c_int64]
    m.mock_objs.argtypes = [vp, ctypes.c_int64]
    m.mock_ints.argtypes = [vp, ctypes.c_int64]
    m.mock_array_data.argtypes = [vp]
    m.mock_array_len.restype = ctypes.c_int64
    m.mock_array_len.argtypes = [vp]
    m.mock_stream.argtypes = [ctypes.c_int64]
    m.mock_block_iterator.argtypes = [vp, ctypes.c_int64]
    for f in ("mock_info_address", "mock_info_rows"):
        getattr(m, f).restype = ctypes.c_int64
        getattr(m, f).argtypes = [vp]
    for f in ("mock_info_offsets", "mock_info_lengths"):
        getattr(m, f).restype = vp
        getattr(m, f).argtypes = [vp]
    m.mock_string.argtypes = [ctypes.c_char_p]
    m.mock_memory_manager.restype = vp
    m.mock_memory_manager.argtypes = [ctypes.c_int64]
    m.mock_memory_stats.restype = ctypes.POINTER(ctypes.c_int64)
    m.mock_memory_stats.argtypes = [vp]
    m.mock_metrics_len.restype = ctypes.c_int64
    m.mock_metrics_len.argtypes = [vp]
    m.mock_metrics_pushes.restype = ctypes.c_int64
    m.mock_metrics_pushes.argtypes = [vp]
    m.mock_metrics_bytes.restype = vp
    m.mock_metrics_bytes.argtypes = [vp]
    m.mock_set_subquery.restype = None
    m.mock_set_subquery.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_char_p, ctypes.c_int64]
    m.mock_static_calls.restype = ctypes.c_char_p
    m.mock_exception_class.restype = ctypes.c_char_p
    m.mock_exception_msg.restype = ctypes.c_char_p
    return m


class Jvm:
    """Calls the JNI exports with the exact argument lists of Native.scala:60-111."""

    def __init__(self, libcomet: ctypes.CDLL):
        self.m = load()
        self.lib = libcomet
        self.env = self.m.mock_env()
        vp, i64, i32, u8 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint8
        f = libcomet.Java_org_apache_comet_Native_createPlan
        f.restype = i64
        f.argtypes = [vp, vp, i64, vp, vp, vp, i32, vp, i64, vp, vp, i32, u8, vp, i64, i64, i64, i64, vp, vp, vp]
        g = libcomet.Java_org_apache_comet_Native_executePlan
        g.restype = i64
        g.argtypes = [vp, vp, i32, i32, i64, vp, vp]
        r = libcomet.Java_org_apache_comet_Native_releasePlan
        r.restype = None
        r.argtypes = [vp, vp, i64]

    def create_plan(self, stream_addrs, plan: bytes, config: bytes = b"", batch_size=8192, metrics_node=None, task_attempt_id=0,
                    iterator_objects=None, metrics_interval_ms=1000, memory_manager=None):
        objs = iterator_objects if iterator_objects is not None else [self.m.mock_stream(a) for a in stream_addrs]
        arr = (ctypes.c_void_p * max(len(objs), 1))(*objs)
        its = self.m.mock_objs(arr, len(objs))
        return self.lib.Java_org_apache_comet_Native_createPlan(
            self.env, None, 1, its, self.m.mock_bytes(plan, len(plan)), self.m.mock_bytes(config, len(config)) if config else None, 1,
            metrics_node, metrics_interval_ms, memory_manager, None, batch_size, 1, None, 0, 0, task_attempt_id, 1, None, None, None)

    def memory_manager(self, limit: int):
        """a CometTaskMemoryManager with `limit` bytes to grant; .stats() → dict"""
        obj = self.m.mock_memory_manager(limit)
        m = self.m

        class _MM:
            handle = obj

            @staticmethod
            def stats():
                st = m.mock_memory_stats(obj)
                return dict(limit=st[0], used=st[1], peak=st[2], acquires=st[3], releases=st[4], refused=st[5])
        return _MM

    def execute_plan(self, handle, array_addrs, schema_addrs):
        a = (ctypes.c_int64 * max(len(array_addrs), 1))(*array_addrs)
        s = (ctypes.c_int64 * max(len(schema_addrs), 1))(*schema_addrs)
        return self.lib.Java_org_apache_comet_Native_executePlan(self.env, None, 0, 0, handle, self.m.mock_longs(a, len(array_addrs)),
                                                                 self.m.mock_longs(s, len(schema_addrs)))

    def block_iterator(self, blocks):
        """A CometShuffleBlockIterator stand-in over `blocks` (each from its codec tag on)."""
        objs = [self.m.mock_bytes(b, len(b)) for b in blocks]
        arr = (ctypes.c_void_p * max(len(objs), 1))(*objs)
        return self.m.mock_block_iterator(arr, len(objs))

    def decode_shuffle_block(self, block: bytes, array_addrs, schema_addrs):
        f = self.lib.Java_org_apache_comet_Native_decodeShuffleBlock
        f.restype = ctypes.c_int64
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint8]
        a = (ctypes.c_int64 * max(len(array_addrs), 1))(*array_addrs)
        s = (ctypes.c_int64 * max(len(schema_addrs), 1))(*schema_addrs)
        return f(self.env, None, self.m.mock_bytes(block, len(block)), len(block), self.m.mock_longs(a, len(array_addrs)),
                 self.m.mock_longs(s, len(schema_addrs)), 0)

    def columnar_to_row(self, batch):
        """columnarToRowInit → Convert → Close; returns the rows as bytes objects read from the NativeColumnarToRowInfo the shim built."""
        import pyarrow as pa  # noqa: F401
        from datafusion_comet_amd import native
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
        init = self.lib.Java_org_apache_comet_Native_columnarToRowInit
        init.restype, init.argtypes = i64, [vp, vp, vp, i32]
        conv = self.lib.Java_org_apache_comet_Native_columnarToRowConvert
        conv.restype, conv.argtypes = vp, [vp, vp, i64, vp, vp, i32]
        close = self.lib.Java_org_apache_comet_Native_columnarToRowClose
        close.restype, close.argtypes = None, [vp, vp, i64]
        h = init(self.env, None, None, 8192)
        n = batch.num_columns
        arrays = [native.ArrowArrayC() for _ in range(n)]
        schemas = [native.ArrowSchemaC() for _ in range(n)]
        for i in range(n):
            batch.column(i)._export_to_c(ctypes.addressof(arrays[i]), ctypes.addressof(schemas[i]))
        a = (ctypes.c_int64 * max(n, 1))(*[ctypes.addressof(x) for x in arrays])
        s = (ctypes.c_int64 * max(n, 1))(*[ctypes.addressof(x) for x in schemas])
        info = conv(self.env, None, h, self.m.mock_longs(a, n), self.m.mock_longs(s, n), batch.num_rows)
        rows = None
        if info:
            k = self.m.mock_info_rows(info)
            base = self.m.mock_info_address(info)
            offs = (ctypes.c_int32 * k).from_address(self.m.mock_info_offsets(info))
            lens = (ctypes.c_int32 * k).from_address(self.m.mock_info_lengths(info))
            rows = [ctypes.string_at(base + offs[i], lens[i]) for i in range(k)]
        close(self.env, None, h)
        return rows

    def sort_row_partitions(self, address: int, size: int):
        f = self.lib.Java_org_apache_comet_Native_sortRowPartitionsNative
        f.restype, f.argtypes = None, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint8]
        f(self.env, None, address, size, 0)

    def write_sorted_file(self, addresses, row_sizes, datatypes, path: str, batch_size: int, checksum_enabled: bool, checksum_algo: int,
                          current_checksum: int, codec: str, level: int = 1, prefer_dictionary_ratio: float = 10.0):
        """Native.writeSortedFileNative with the argument list of Native.scala:146-158; returns the three longs or None (exception pending)."""
        vp, i64, i32, u8 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint8
        f = self.lib.Java_org_apache_comet_Native_writeSortedFileNative
        f.restype = vp
        f.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_double, i32, u8, i32, i64, vp, i32, u8]
        a = (i64 * max(len(addresses), 1))(*addresses)
        s = (i32 * max(len(row_sizes), 1))(*row_sizes)
        objs = [self.m.mock_bytes(t, len(t)) for t in datatypes]
        arr = (vp * max(len(objs), 1))(*objs)
        out = f(self.env, None, self.m.mock_longs(a, len(addresses)), self.m.mock_ints(s, len(row_sizes)), self.m.mock_objs(arr, len(objs)),
                self.m.mock_string(path.encode()), prefer_dictionary_ratio, batch_size, 1 if checksum_enabled else 0, checksum_algo, current_checksum,
                self.m.mock_string(codec.encode()), level, 0)
        if not out:
            return None
        k = self.m.mock_array_len(out)
        return list((i64 * k).from_address(self.m.mock_array_data(out)))

    def release_plan(self, handle):
        self.lib.Java_org_apache_comet_Native_releasePlan(self.env, None, handle)

    def exception(self):
        if not self.m.mock_exception_pending():
            return None
        e = (self.m.mock_exception_class().decode(), self.m.mock_exception_msg().decode())
        self.m.mock_exception_clear()
        return e
