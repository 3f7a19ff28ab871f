"""Host-side mirror of the reference's JVM boundary, over the C ABI of libcomet.so.

``Native`` has the three calls of ``org.apache.comet.Native`` (spark/src/main/scala/org/apache/comet/
Native.scala:60-111) and ``CometExecIterator`` mirrors the Scala class of the same name
(spark/src/main/scala/org/apache/comet/CometExecIterator.scala:64-256): createPlan once, executePlan per
output batch until -1, releasePlan on close.  Inputs are exported through the Arrow C Stream interface
(host memory, what the JVM does) or the Arrow C Device Stream interface (buffers already in HBM).

There is no CPU fallback here: if libcomet.so is missing or fails to load, import raises.
"""
from __future__ import annotations

import ctypes
import os
import sys
from typing import Iterator, List, Optional, Sequence

import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcomet.so")


class CometNativeException(RuntimeError):
    """org.apache.comet.CometNativeException"""


class CometQueryExecutionException(RuntimeError):
    """org.apache.comet.exceptions.CometQueryExecutionException (message is the Spark error JSON)"""


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the native engine)")
    # The harness keeps device buffers in torch tensors; torch bundles its own copy of libamdhip64.so.7.
    # Import it first so that libcomet.so (NEEDED libamdhip64.so.7) binds to the SAME runtime instance —
    # two HIP runtimes in one process do not share a device context.  (Under the JVM there is no torch and
    # the system ROCm runtime is used.)
    import datafusion_comet_amd as _pkg
    if _pkg.SYSTEM_COMGR is None and os.environ.get("COMET_SYSTEM_COMGR", "1") != "0" and "torch" in sys.modules and os.path.exists("/opt/rocm/lib/libamd_comgr.so.3"):
        sys.stderr.write("datafusion_comet_amd: torch was imported first — the JIT compiles with the torch wheel's bundled ROCm compiler, not the installed one "
                         "(import datafusion_comet_amd before torch; see datafusion_comet_amd/__init__.py)\n")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    c = ctypes
    lib.comet_create_plan.restype = c.c_int64
    lib.comet_create_plan.argtypes = [c.c_void_p, c.c_size_t, c.c_void_p, c.c_size_t, c.POINTER(c.c_void_p),
                                      c.POINTER(c.c_int32), c.c_int32, c.c_int32, c.c_int32, c.c_int32]
    lib.comet_execute_plan.restype = c.c_int64
    lib.comet_execute_plan.argtypes = [c.c_int64, c.POINTER(c.c_void_p), c.POINTER(c.c_void_p), c.c_int32]
    lib.comet_release_plan.restype = None
    lib.comet_release_plan.argtypes = [c.c_int64]
    lib.comet_last_error.restype = c.c_char_p
    lib.comet_last_error.argtypes = [c.c_int64]
    lib.comet_last_error_kind.restype = c.c_int32
    lib.comet_last_error_kind.argtypes = [c.c_int64]
    lib.comet_plan_metrics.restype = c.c_int64
    lib.comet_plan_metrics.argtypes = [c.c_int64, c.c_void_p, c.c_size_t]
    lib.comet_explain.restype = c.c_char_p
    lib.comet_explain.argtypes = [c.c_int64]
    lib.comet_plan_kernel_stats.restype = None
    lib.comet_plan_kernel_stats.argtypes = [c.c_int64, c.POINTER(c.c_double), c.POINTER(c.c_int64), c.POINTER(c.c_int64)]
    lib.comet_set_kernel_times.restype = None
    lib.comet_set_kernel_times.argtypes = [c.c_int32]
    lib.comet_plan_kernel_times.restype = c.c_int64
    lib.comet_plan_kernel_times.argtypes = [c.c_int64, c.c_char_p, c.c_size_t]
    lib.comet_plan_aux_kernel_stats.restype = None
    lib.comet_plan_aux_kernel_stats.argtypes = [c.c_int64, c.POINTER(c.c_double), c.POINTER(c.c_int64)]
    lib.comet_compile_plan.restype = c.c_int32
    lib.comet_compile_plan.argtypes = [c.c_void_p, c.c_size_t, c.c_char_p, c.c_size_t]
    lib.comet_check_plan.restype = c.c_int32
    lib.comet_check_plan.argtypes = [c.c_void_p, c.c_size_t, c.c_char_p, c.c_size_t]
    lib.comet_murmur3_column.restype = c.c_int32
    lib.comet_murmur3_column.argtypes = [c.c_int32, c.c_int32, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
    lib.comet_pmod_partition.restype = c.c_int32
    lib.comet_pmod_partition.argtypes = [c.c_void_p, c.c_int64, c.c_int32, c.c_void_p, c.c_void_p]
    lib.comet_version.restype = c.c_char_p
    lib.comet_execute_plan_device.restype = c.c_int64
    lib.comet_execute_plan_device.argtypes = [c.c_int64, c.POINTER(c.c_void_p), c.POINTER(c.c_void_p), c.c_int32]
    lib.comet_partition_indices.restype = c.c_int32
    lib.comet_partition_indices.argtypes = [c.c_void_p, c.c_int64, c.c_int32, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.comet_take_column.restype = c.c_int32
    lib.comet_take_column.argtypes = [c.c_int32, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
    lib.comet_take_utf8_offsets.restype = c.c_int64
    lib.comet_take_utf8_offsets.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p]
    lib.comet_take_utf8_bytes.restype = c.c_int32
    lib.comet_take_utf8_bytes.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.comet_decode_shuffle_block.restype = c.c_int64
    lib.comet_decode_shuffle_block.argtypes = [c.c_void_p, c.c_int64, c.POINTER(c.c_void_p), c.POINTER(c.c_void_p), c.c_int32]
    lib.comet_encode_shuffle_block.restype = c.c_int32
    lib.comet_encode_shuffle_block.argtypes = [c.POINTER(c.c_void_p), c.POINTER(c.c_void_p), c.c_int32, c.c_int32, c.c_int32,
                                               c.POINTER(c.c_void_p), c.POINTER(c.c_int64)]
    lib.comet_concat_nested_column.restype = c.c_int64
    lib.comet_concat_nested_column.argtypes = [c.POINTER(c.c_void_p), c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p]
    lib.comet_sort_row_partitions.restype = None
    lib.comet_sort_row_partitions.argtypes = [c.c_void_p, c.c_int64]
    lib.comet_write_sorted_rows.restype = c.c_int32
    lib.comet_write_sorted_rows.argtypes = [c.c_void_p, c.c_void_p, c.c_int64, c.POINTER(c.c_char_p), c.c_void_p, c.c_int32, c.c_char_p, c.c_int32,
                                            c.c_int32, c.c_int32, c.c_int64, c.c_char_p, c.c_int32, c.POINTER(c.c_int64)]
    lib.comet_columnar_to_row_init.restype = c.c_int64
    lib.comet_columnar_to_row_init.argtypes = [c.c_int32, c.c_int32]
    lib.comet_columnar_to_row_convert.restype = c.c_int32
    lib.comet_columnar_to_row_convert.argtypes = [c.c_int64, c.POINTER(c.c_void_p), c.POINTER(c.c_void_p), c.c_int32, c.c_int64, c.POINTER(c.c_void_p),
                                                  c.POINTER(c.c_void_p), c.POINTER(c.c_void_p)]
    lib.comet_columnar_to_row_close.restype = None
    lib.comet_columnar_to_row_close.argtypes = [c.c_int64]
    lib.comet_columnar_to_row_error.restype = c.c_char_p
    lib.comet_columnar_to_row_error.argtypes = [c.c_int64]
    lib.comet_comm_unique_id.restype = c.c_int32
    lib.comet_comm_unique_id.argtypes = [c.c_void_p]
    lib.comet_comm_init_rank.restype = c.c_int64
    lib.comet_comm_init_rank.argtypes = [c.c_void_p, c.c_int32, c.c_int32, c.c_int32]
    lib.comet_comm_init_local.restype = c.c_int64
    lib.comet_comm_init_local.argtypes = [c.c_int64, c.c_int32, c.c_int32, c.c_int32]
    lib.comet_comm_init_tcp.restype = c.c_int64
    lib.comet_comm_init_tcp.argtypes = [c.c_char_p, c.c_int32, c.c_int32, c.c_int32, c.c_int32]
    lib.comet_comm_transport.restype = c.c_char_p
    lib.comet_comm_transport.argtypes = [c.c_int64]
    lib.comet_comm_destroy.restype = None
    lib.comet_comm_destroy.argtypes = [c.c_int64]
    lib.comet_exchange.restype = c.c_int64
    lib.comet_exchange.argtypes = [c.c_int64, c.c_int32, c.c_void_p, c.c_int64, c.POINTER(c.c_int32), c.c_int32]
    lib.comet_exchange_result_rows.restype = c.c_int64
    lib.comet_exchange_result_rows.argtypes = [c.c_int64]
    lib.comet_exchange_result_column.restype = c.c_int32
    lib.comet_exchange_result_column.argtypes = [c.c_int64, c.c_int32, c.POINTER(c.c_void_p), c.POINTER(c.c_void_p)]
    lib.comet_parquet_host_plain_values.restype = c.c_int64
    lib.comet_parquet_host_plain_values.argtypes = [c.c_char_p, c.c_size_t, c.c_int32, c.c_void_p, c.c_size_t]
    lib.comet_exchange_result_aux.restype = c.c_int32
    lib.comet_exchange_result_aux.argtypes = [c.c_int64, c.c_int32, c.POINTER(c.c_void_p), c.POINTER(c.c_int64)]
    lib.comet_exchange_result_release.restype = None
    lib.comet_exchange_result_release.argtypes = [c.c_int64]
    lib.comet_exchange_last_error.restype = c.c_char_p
    lib.comet_free_buffer.restype = None
    lib.comet_free_buffer.argtypes = [c.c_void_p]
    lib.comet_plan_memory_stats.restype = None
    lib.comet_plan_memory_stats.argtypes = [c.c_int64, c.c_void_p]
    lib.comet_plan_set_memory_manager.restype = c.c_int32
    lib.comet_plan_set_memory_manager.argtypes = [c.c_int64, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int64]
    lib.comet_parquet_prune_report.restype = c.c_int64
    lib.comet_parquet_prune_report.argtypes = [c.c_char_p, c.c_size_t, c.c_int32, c.c_char_p, c.c_size_t]
    lib.comet_error_json.restype = c.c_int64
    lib.comet_error_json.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p, c.c_char_p, c.c_int32, c.c_int32, c.c_int32, c.c_char_p, c.c_uint64, c.c_uint64, c.c_char_p,
                                     c.c_int64, c.c_char_p, c.c_int64]
    lib.comet_plan_error_json.restype = c.c_int64
    lib.comet_plan_error_json.argtypes = [c.c_char_p, c.c_size_t, c.c_int32, c.c_uint64, c.c_uint64, c.c_char_p, c.c_int64, c.c_char_p, c.c_int64]
    lib.comet_zone_table.restype = c.c_int64
    lib.comet_zone_table.argtypes = [c.c_char_p, c.c_void_p, c.c_int64]
    lib.comet_rlike_match.restype = c.c_int32
    lib.comet_rlike_match.argtypes = [c.c_char_p, c.c_char_p, c.c_size_t]
    lib.comet_regexp_extract_host.restype = c.c_int32
    lib.comet_regexp_extract_host.argtypes = [c.c_char_p, c.c_int32, c.c_char_p, c.c_size_t, c.POINTER(c.c_int32), c.POINTER(c.c_int32)]
    lib.comet_extract_all_host.restype = c.c_int32
    lib.comet_extract_all_host.argtypes = [c.c_char_p, c.c_int32, c.c_char_p, c.c_size_t, c.POINTER(c.c_int32), c.POINTER(c.c_int32), c.c_int32]
    lib.comet_strfn_host.restype = c.c_int64
    lib.comet_strfn_host.argtypes = [c.c_int32, c.c_char_p, c.c_int32, c.c_char_p, c.c_int32, c.c_char_p, c.c_int32, c.c_int64, c.c_void_p, c.c_int64]
    lib.comet_plan_codegen.restype = c.c_int64
    lib.comet_plan_codegen.argtypes = [c.c_char_p, c.c_size_t, c.c_char_p, c.c_int32, c.c_void_p, c.c_int64]
    lib.comet_error_site_json.restype = c.c_int64
    lib.comet_error_site_json.argtypes = [c.c_uint32, c.c_uint64, c.c_uint64, c.c_char_p, c.c_int64, c.c_void_p, c.c_int64]
    lib.comet_plan_site_error_json.restype = c.c_int64
    lib.comet_plan_site_error_json.argtypes = [c.c_char_p, c.c_size_t, c.c_uint32, c.c_uint64, c.c_uint64, c.c_char_p, c.c_int64, c.c_void_p, c.c_int64]
    lib.comet_embedded_header.restype = c.c_int64
    lib.comet_embedded_header.argtypes = [c.c_char_p, c.c_void_p, c.c_int64]
    lib.comet_plan_set_subquery.restype = c.c_int32
    lib.comet_plan_set_subquery.argtypes = [c.c_int64, c.c_int64, c.c_int32, c.c_char_p, c.c_size_t]
    lib.comet_split_host.restype = c.c_int32
    lib.comet_split_host.argtypes = [c.c_char_p, c.c_int32, c.c_char_p, c.c_size_t, c.POINTER(c.c_int32), c.POINTER(c.c_int32), c.c_int32]
    lib.comet_date_fn_host.restype = c.c_int32
    lib.comet_date_fn_host.argtypes = [c.c_int32, c.c_int64, c.c_int64, c.c_int64, c.POINTER(c.c_int64)]
    lib.comet_page_decompress.restype = c.c_int32
    lib.comet_page_decompress.argtypes = [c.c_int32, c.c_char_p, c.c_size_t, c.c_void_p, c.c_size_t]
    lib.comet_snappy_inflate_pages.restype = c.c_int64
    lib.comet_snappy_inflate_pages.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_void_p, c.c_void_p, c.c_int32, c.POINTER(c.c_double)]
    return lib


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


# --------------------------------------------------------------------------- Arrow C structs


class ArrowSchemaC(ctypes.Structure):
    pass


class ArrowArrayC(ctypes.Structure):
    pass


ArrowSchemaC._fields_ = [("format", ctypes.c_char_p), ("name", ctypes.c_char_p), ("metadata", ctypes.c_char_p),
                         ("flags", ctypes.c_int64), ("n_children", ctypes.c_int64),
                         ("children", ctypes.POINTER(ctypes.POINTER(ArrowSchemaC))),
                         ("dictionary", ctypes.POINTER(ArrowSchemaC)), ("release", ctypes.c_void_p),
                         ("private_data", ctypes.c_void_p)]
ArrowArrayC._fields_ = [("length", ctypes.c_int64), ("null_count", ctypes.c_int64), ("offset", ctypes.c_int64),
                        ("n_buffers", ctypes.c_int64), ("n_children", ctypes.c_int64),
                        ("buffers", ctypes.POINTER(ctypes.c_void_p)),
                        ("children", ctypes.POINTER(ctypes.POINTER(ArrowArrayC))),
                        ("dictionary", ctypes.POINTER(ArrowArrayC)), ("release", ctypes.c_void_p),
                        ("private_data", ctypes.c_void_p)]


class ArrowArrayStreamC(ctypes.Structure):
    _fields_ = [("get_schema", ctypes.c_void_p), ("get_next", ctypes.c_void_p), ("get_last_error", ctypes.c_void_p),
                ("release", ctypes.c_void_p), ("private_data", ctypes.c_void_p)]


class ArrowDeviceArrayC(ctypes.Structure):
    _fields_ = [("array", ArrowArrayC), ("device_id", ctypes.c_int64), ("device_type", ctypes.c_int32),
                ("sync_event", ctypes.c_void_p), ("reserved", ctypes.c_int64 * 3)]


class ArrowDeviceArrayStreamC(ctypes.Structure):
    _fields_ = [("device_type", ctypes.c_int32), ("get_schema", ctypes.c_void_p), ("get_next", ctypes.c_void_p),
                ("get_last_error", ctypes.c_void_p), ("release", ctypes.c_void_p), ("private_data", ctypes.c_void_p)]


ARROW_DEVICE_ROCM = 10

_GET_SCHEMA_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
_GET_NEXT_T = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
_LAST_ERR_T = ctypes.CFUNCTYPE(ctypes.c_char_p, ctypes.c_void_p)
_RELEASE_T = ctypes.CFUNCTYPE(None, ctypes.c_void_p)


class HostInput:
    """An input handed over as struct ArrowArrayStream* (what CometNativeArrowSource exports)."""

    kind = 0

    def __init__(self, reader: pa.RecordBatchReader):
        self._c = ArrowArrayStreamC()
        reader._export_to_c(ctypes.addressof(self._c))

    @staticmethod
    def from_table(table: pa.Table, batch_rows: int = 8192) -> "HostInput":
        batches = table.to_batches(max_chunksize=batch_rows)
        return HostInput(pa.RecordBatchReader.from_batches(table.schema, batches))

    @property
    def address(self) -> int:
        return ctypes.addressof(self._c)


_FIXED_WIDTH = {pa.int8(): 1, pa.int16(): 2, pa.int32(): 4, pa.int64(): 8, pa.float32(): 4, pa.float64(): 8,
                pa.date32(): 4}


class DeviceTable:
    """Arrow-layout columns resident in HBM (torch tensors own the memory)."""

    def __init__(self, schema: pa.Schema, num_rows: int, values, validity, device, aux=None):
        self.schema, self.num_rows, self.values, self.validity, self.device = schema, num_rows, values, validity, device
        self.aux = aux if aux is not None else [None] * len(values)   # Utf8: data bytes (values = int32 offsets)

    @staticmethod
    def from_arrow(table: pa.Table, device="cuda:0") -> "DeviceTable":
        import numpy as np
        import torch
        table = table.combine_chunks()
        vals, valid, aux = [], [], []
        for name in table.schema.names:
            arr = table.column(name).chunk(0) if table.num_rows else pa.array([], type=table.schema.field(name).type)
            if arr.offset != 0:
                arr = pa.concat_arrays([arr])
            bufs = arr.buffers()
            data = bufs[1]
            host = np.frombuffer(data, dtype=np.uint8) if data is not None and data.size else np.zeros(0, np.uint8)
            # torch allocations are 256 B aligned → Decimal128 loads are 16 B aligned
            vals.append(torch.from_numpy(host.copy()).to(device))
            if arr.null_count and bufs[0] is not None:
                valid.append(torch.from_numpy(np.frombuffer(bufs[0], dtype=np.uint8).copy()).to(device))
            else:
                valid.append(None)
            if pa.types.is_string(arr.type) or pa.types.is_binary(arr.type):
                d = bufs[2]
                hd = np.frombuffer(d, dtype=np.uint8) if d is not None and d.size else np.zeros(1, np.uint8)
                aux.append(torch.from_numpy(hd.copy()).to(device))
            else:
                aux.append(None)
        return DeviceTable(table.schema, table.num_rows, vals, valid, device, aux)

    def with_string_hints(self) -> "DeviceTable":
        """The same resident table with field metadata ``comet:utf8_fixed_len`` = L on every Utf8 column whose values all occupy exactly L ≤ 15
        bytes (TPC-H's CHAR(1) flags).  The owner of an immutable resident table measures this ONCE here; the engine then checks only the end
        points of each batch instead of re-reading 4 B/row of offsets in every task (exec_input.cpp pull_device_table).  The metadata is an
        assertion by the producer: set it only on buffers that do not change while the table is in use."""
        import torch
        fields = []
        for i, f in enumerate(self.schema):
            if pa.types.is_string(f.type) and self.num_rows > 0:
                offs = self.values[i].view(torch.int32)[: self.num_rows + 1]
                total = int(offs[-1]) - int(offs[0])
                L = total // self.num_rows
                if int(offs[0]) == 0 and total == L * self.num_rows and L <= 15:
                    ok = True
                    step = 1 << 26
                    for a in range(0, self.num_rows, step):     # in slices: no n-sized temporaries next to a table that fills the HBM
                        b = min(self.num_rows, a + step)
                        ok = ok and bool(((offs[a + 1:b + 1] - offs[a:b]) == L).all())
                        if not ok:
                            break
                    if ok:
                        md = dict(f.metadata or {})
                        md[b"comet:utf8_fixed_len"] = str(L).encode()
                        f = f.with_metadata(md)
            fields.append(f)
        return DeviceTable(pa.schema(fields, metadata=self.schema.metadata), self.num_rows, self.values, self.validity, self.device, self.aux)

    def nbytes(self) -> int:
        return (sum(v.numel() for v in self.values) + sum(v.numel() for v in self.validity if v is not None)
                + sum(v.numel() for v in self.aux if v is not None))

    def slice(self, start: int, length: int) -> "DeviceTable":
        """Zero-copy row range (columns without validity only; Utf8 keeps its whole data buffer, offsets stay absolute)."""
        vals = []
        for f, v, valid in zip(self.schema, self.values, self.validity):
            if valid is not None:
                raise ValueError("DeviceTable.slice: columns with validity bitmaps are not supported")
            if pa.types.is_string(f.type) or pa.types.is_binary(f.type):
                vals.append(v[start * 4:(start + length + 1) * 4])
            else:
                w = value_width(f.type)
                vals.append(v[start * w:(start + length) * w])
        return DeviceTable(self.schema, length, vals, [None] * len(vals), self.device, list(self.aux))

    def to_arrow(self) -> pa.Table:
        """Copy back to host memory as a pyarrow Table (tests / small results only)."""
        import numpy as np
        cols = []
        for i, f in enumerate(self.schema):
            host = lambda t: pa.py_buffer(t.detach().cpu().numpy().tobytes()) if t is not None else None
            bufs = [host(self.validity[i]), host(self.values[i])]
            if pa.types.is_string(f.type) or pa.types.is_binary(f.type):
                bufs.append(host(self.aux[i]))
            cols.append(pa.Array.from_buffers(f.type, self.num_rows, bufs, null_count=-1 if self.validity[i] is not None else 0))
        return pa.Table.from_arrays(cols, schema=self.schema)


def value_width(t: pa.DataType) -> int:
    """Bytes per value of the values buffer (0 = bit-packed Boolean); Utf8 is not fixed width."""
    if pa.types.is_boolean(t):
        return 0
    if pa.types.is_decimal(t):
        return 16
    if t in _FIXED_WIDTH:
        return _FIXED_WIDTH[t]
    if pa.types.is_timestamp(t):
        return 8
    raise CometNativeException(f"no fixed value width for {t}")


class _ExportedBatch:
    """Owns the ArrowDeviceArrays a plan exported (comet_execute_plan_device); releases them when collected."""

    def __init__(self, arrays):
        self.arrays = arrays

    def __del__(self):
        for a in self.arrays:
            try:
                if a.array.release:
                    _RELEASE_T(a.array.release)(ctypes.addressof(a.array))
            except Exception:
                pass


class _DeviceBuffer:
    """One exported HBM buffer, visible to torch through __cuda_array_interface__ (zero copy; keeps its batch alive)."""

    def __init__(self, owner, ptr: int, nbytes: int):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}


class DeviceInput:
    """An input handed over as struct ArrowDeviceArrayStream* with ARROW_DEVICE_ROCM buffers.

    The stream yields the table as ONE batch per ``get_next`` (optionally split into ``splits`` row ranges);
    it can be rewound with ``rewind()`` so a bench loop can re-run the plan on the same resident data.
    """

    kind = 1

    def __init__(self, table: DeviceTable, device_id: int = 0):
        self.table = table
        self.device_id = device_id
        self._emitted = False
        self._keep = []  # ctypes objects referenced from C structs
        self._c = ArrowDeviceArrayStreamC()
        self._cb_schema = _GET_SCHEMA_T(self._get_schema)
        self._cb_next = _GET_NEXT_T(self._get_next)
        self._cb_err = _LAST_ERR_T(lambda _s: b"")
        self._cb_release = _RELEASE_T(self._release)
        self._cb_arr_release = _RELEASE_T(self._array_release)
        self._c.device_type = ARROW_DEVICE_ROCM
        self._c.get_schema = ctypes.cast(self._cb_schema, ctypes.c_void_p)
        self._c.get_next = ctypes.cast(self._cb_next, ctypes.c_void_p)
        self._c.get_last_error = ctypes.cast(self._cb_err, ctypes.c_void_p)
        self._c.release = ctypes.cast(self._cb_release, ctypes.c_void_p)
        self.released = False

    @property
    def address(self) -> int:
        return ctypes.addressof(self._c)

    def _get_schema(self, _self, out) -> int:
        self.table.schema._export_to_c(out)
        return 0

    def _array_release(self, arr_ptr):
        a = ctypes.cast(arr_ptr, ctypes.POINTER(ArrowArrayC)).contents
        a.release = None

    def _get_next(self, _self, out) -> int:
        dev = ctypes.cast(out, ctypes.POINTER(ArrowDeviceArrayC)).contents
        ctypes.memset(out, 0, ctypes.sizeof(ArrowDeviceArrayC))
        if self._emitted:
            return 0  # release == NULL → end of stream
        self._emitted = True
        t = self.table
        n = len(t.values)
        kids = (ctypes.POINTER(ArrowArrayC) * n)()
        release = ctypes.cast(self._cb_arr_release, ctypes.c_void_p)
        for i in range(n):
            child = ArrowArrayC()
            nb = 3 if t.aux[i] is not None else 2
            bufs = (ctypes.c_void_p * nb)()
            bufs[0] = t.validity[i].data_ptr() if t.validity[i] is not None else None
            bufs[1] = t.values[i].data_ptr() if t.values[i].numel() else None
            if nb == 3:
                bufs[2] = t.aux[i].data_ptr()
            child.length = t.num_rows
            child.null_count = -1 if t.validity[i] is not None else 0
            child.offset = 0
            child.n_buffers = nb
            child.n_children = 0
            child.buffers = bufs
            child.release = release
            kids[i] = ctypes.pointer(child)
            self._keep += [child, bufs]
        top_bufs = (ctypes.c_void_p * 1)()
        top_bufs[0] = None
        dev.array.length = t.num_rows
        dev.array.null_count = 0
        dev.array.offset = 0
        dev.array.n_buffers = 1
        dev.array.n_children = n
        dev.array.buffers = top_bufs
        dev.array.children = kids
        dev.array.release = release
        dev.device_id = self.device_id
        dev.device_type = ARROW_DEVICE_ROCM
        self._keep += [kids, top_bufs]
        return 0

    def _release(self, _self):
        self.released = True
        self._c.release = None

    def close(self) -> None:
        """Drop the table and the ctypes callbacks.  The callbacks are bound methods, i.e. reference cycles through self:
        without this the HBM tensors of a finished stage would live until the cyclic GC happens to run."""
        self.table = None
        self._keep = []
        self._cb_schema = self._cb_next = self._cb_err = self._cb_release = self._cb_arr_release = None

    def rearm(self) -> "DeviceInput":
        """Make the (released) stream usable for another plan over the same resident table (bench loops)."""
        self._emitted = False
        self._keep = []
        self.released = False
        self._c.release = ctypes.cast(self._cb_release, ctypes.c_void_p)
        return self


# --------------------------------------------------------------------------- Native (Native.scala)


class CometShuffleBlockStreamC(ctypes.Structure):
    _fields_ = [("next_block", ctypes.c_void_p), ("get_last_error", ctypes.c_void_p), ("release", ctypes.c_void_p),
                ("private_data", ctypes.c_void_p)]


class ShuffleBlockInput:
    """Input of a ShuffleScan leaf: plays org.apache.comet.CometShuffleBlockIterator — hands the library one shuffle block at a
    time, each starting at its codec tag (struct CometShuffleBlockStream, include/comet_amd.h)."""
    kind = 2  # COMET_INPUT_SHUFFLE_BLOCKS

    def __init__(self, blocks: Sequence[bytes]):
        self._blocks = list(blocks)
        self._pos = 0
        self._cur = None
        self._next_t = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p))
        self._err_t = ctypes.CFUNCTYPE(ctypes.c_char_p, ctypes.c_void_p)
        self._rel_t = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
        self._cbs = (self._next_t(self._next), self._err_t(lambda _s: None), self._rel_t(lambda _s: None))
        self._c = CometShuffleBlockStreamC()
        self._c.next_block = ctypes.cast(self._cbs[0], ctypes.c_void_p).value
        self._c.get_last_error = ctypes.cast(self._cbs[1], ctypes.c_void_p).value
        self._c.release = ctypes.cast(self._cbs[2], ctypes.c_void_p).value

    @staticmethod
    def from_files(data_file: str, index_file: str, partition: int) -> "ShuffleBlockInput":
        """The blocks of one reduce partition of a (data, index) pair written by ShuffleWriter, headers stripped the way the JVM
        reader strips them (8-byte length and 8-byte field count)."""
        return ShuffleBlockInput(read_shuffle_partition(data_file, index_file, partition))

    @property
    def address(self) -> int:
        return ctypes.addressof(self._c)

    def _next(self, _self, out) -> int:
        if self._pos >= len(self._blocks):
            return -1
        b = self._blocks[self._pos]
        self._pos += 1
        self._cur = ctypes.create_string_buffer(b, len(b))
        out[0] = ctypes.addressof(self._cur)
        return len(b)


def read_shuffle_partition(data_file: str, index_file: str, partition: int) -> List[bytes]:
    """Splits partition `partition` of a shuffle data file into blocks (each returned from its codec tag on): the index file
    holds num_partitions + 1 little-endian int64 offsets (local_partition_writer.rs:255-295), every block starts with its
    u64le length and u64le field count (shuffle_block_writer.rs:86-137)."""
    import struct
    idx = open(index_file, "rb").read()
    offs = struct.unpack("<%dq" % (len(idx) // 8), idx)
    with open(data_file, "rb") as f:
        f.seek(offs[partition])
        buf = f.read(offs[partition + 1] - offs[partition])
    out, p = [], 0
    while p < len(buf):
        n, = struct.unpack_from("<q", buf, p)
        out.append(buf[p + 16:p + 8 + n])
        p += 8 + n
    return out


def decode_shuffle_block(block: bytes, num_cols: int) -> pa.RecordBatch:
    """Native.decodeShuffleBlock: one block (from its codec tag) → a record batch."""
    l = lib()
    arrays = [ArrowArrayC() for _ in range(num_cols)]
    schemas = [ArrowSchemaC() for _ in range(num_cols)]
    aaddr = (ctypes.c_void_p * max(num_cols, 1))(*[ctypes.addressof(a) for a in arrays])
    saddr = (ctypes.c_void_p * max(num_cols, 1))(*[ctypes.addressof(s) for s in schemas])
    rows = l.comet_decode_shuffle_block(block, len(block), aaddr, saddr, num_cols)
    if rows < 0:
        _raise_last(0)
    cols = [pa.Array._import_from_c(ctypes.addressof(a), ctypes.addressof(s)) for a, s in zip(arrays, schemas)]
    return pa.RecordBatch.from_arrays(cols, names=[f"c{i}" for i in range(len(cols))])


def concat_nested_column(chunks: Sequence[pa.Array]) -> pa.Array:
    """comet_concat_nested_column: what a Scan leaf makes of the batches of one chunk of a struct / list column before uploading it."""
    l = lib()
    n = len(chunks)
    arrays = [ArrowArrayC() for _ in range(n)]
    schemas = [ArrowSchemaC() for _ in range(n)]
    for i, ch in enumerate(chunks):
        ch._export_to_c(ctypes.addressof(arrays[i]), ctypes.addressof(schemas[i]))
    aaddr = (ctypes.c_void_p * max(n, 1))(*[ctypes.addressof(a) for a in arrays])
    out_a, out_s = ArrowArrayC(), ArrowSchemaC()
    rows = l.comet_concat_nested_column(aaddr, ctypes.addressof(schemas[0]), n, ctypes.addressof(out_a), ctypes.addressof(out_s))
    for a, s in zip(arrays, schemas):
        pa.Array._import_from_c(ctypes.addressof(a), ctypes.addressof(s))
    if rows < 0:
        _raise_last(0)
    return pa.Array._import_from_c(ctypes.addressof(out_a), ctypes.addressof(out_s))


def encode_shuffle_block(batch: pa.RecordBatch, codec: int = 0, level: int = 1) -> bytes:
    """ShuffleBlockWriter::write_batch on host columns: the complete block including its length and field-count words."""
    l = lib()
    n = batch.num_columns
    arrays = [ArrowArrayC() for _ in range(n)]
    schemas = [ArrowSchemaC() for _ in range(n)]
    for i in range(n):
        batch.column(i)._export_to_c(ctypes.addressof(arrays[i]), ctypes.addressof(schemas[i]))
    aaddr = (ctypes.c_void_p * max(n, 1))(*[ctypes.addressof(a) for a in arrays])
    saddr = (ctypes.c_void_p * max(n, 1))(*[ctypes.addressof(s) for s in schemas])
    out = ctypes.c_void_p()
    out_len = ctypes.c_int64()
    rc = l.comet_encode_shuffle_block(aaddr, saddr, n, codec, level, ctypes.byref(out), ctypes.byref(out_len))
    # give the exported structs back to pyarrow so that their release callbacks run
    for a, s in zip(arrays, schemas):
        pa.Array._import_from_c(ctypes.addressof(a), ctypes.addressof(s))
    if rc != 0:
        _raise_last(0)
    data = ctypes.string_at(out.value, out_len.value) if out_len.value else b""
    l.comet_free_buffer(out)
    return data


NO_CHECKSUM = -2**63   # Long.MinValue: "no checksum yet" / "checksums disabled" (jni_api.rs:1078-1083, 1110-1116)


def sort_row_partitions(records) -> None:
    """Native.sortRowPartitionsNative: sorts a contiguous int64 numpy array in place, ascending."""
    import numpy as np
    assert records.dtype == np.int64 and records.flags["C_CONTIGUOUS"]
    lib().comet_sort_row_partitions(records.ctypes.data, len(records))


def write_sorted_rows(row_addresses, row_sizes, datatypes, path: str, batch_size: int, codec: str = "lz4", level: int = 1,
                      checksum_enabled: bool = False, checksum_algo: int = 0, current_checksum: int = NO_CHECKSUM):
    """Native.writeSortedFileNative: UnsafeRows at `row_addresses` (int64 array) / `row_sizes` (int32 array) with the serde DataTypes
    `datatypes` are appended to `path` as shuffle blocks; returns (bytes written, checksum or NO_CHECKSUM, encode nanoseconds)."""
    import numpy as np
    l = lib()
    a = np.ascontiguousarray(row_addresses, dtype=np.int64)
    s = np.ascontiguousarray(row_sizes, dtype=np.int32)
    enc = [t.encode() for t in datatypes]
    tp = (ctypes.c_char_p * max(len(enc), 1))(*enc)
    tl = np.array([len(e) for e in enc] or [0], dtype=np.int32)
    out = (ctypes.c_int64 * 3)()
    rc = l.comet_write_sorted_rows(a.ctypes.data, s.ctypes.data, len(a), tp, tl.ctypes.data, len(enc), path.encode(), batch_size,
                                   1 if checksum_enabled else 0, checksum_algo, current_checksum, codec.encode(), level, out)
    if rc != 0:
        _raise_last(0)
    return out[0], out[1], out[2]


class ColumnarToRow:
    """Native.columnarToRowInit / Convert / Close: Arrow columns → Spark UnsafeRow bytes (one bytes object per row)."""

    def __init__(self, batch_size: int = 8192, device_id: int = 0):
        self.handle = lib().comet_columnar_to_row_init(batch_size, device_id)
        if not self.handle:
            raise CometNativeException("columnarToRowInit failed")

    def convert(self, batch: pa.RecordBatch, num_rows: Optional[int] = None) -> List[bytes]:
        l = lib()
        n = batch.num_columns
        rows = batch.num_rows if num_rows is None else num_rows
        arrays = [ArrowArrayC() for _ in range(n)]
        schemas = [ArrowSchemaC() for _ in range(n)]
        for i in range(n):
            batch.column(i)._export_to_c(ctypes.addressof(arrays[i]), ctypes.addressof(schemas[i]))
        aaddr = (ctypes.c_void_p * max(n, 1))(*[ctypes.addressof(a) for a in arrays])
        saddr = (ctypes.c_void_p * max(n, 1))(*[ctypes.addressof(s) for s in schemas])
        buf, offs, lens = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        rc = l.comet_columnar_to_row_convert(self.handle, aaddr, saddr, n, rows, ctypes.byref(buf), ctypes.byref(offs), ctypes.byref(lens))
        if rc != 0:
            raise CometNativeException((l.comet_columnar_to_row_error(self.handle) or b"").decode(errors="replace"))
        if rows == 0:
            return []
        o = (ctypes.c_int32 * rows).from_address(offs.value)
        ln = (ctypes.c_int32 * rows).from_address(lens.value)
        return [ctypes.string_at(buf.value + o[i], ln[i]) for i in range(rows)]

    def close(self):
        if self.handle:
            lib().comet_columnar_to_row_close(self.handle)
            self.handle = 0


def _raise_last(handle: int):
    l = lib()
    msg = (l.comet_last_error(handle) or b"").decode(errors="replace")
    kind = l.comet_last_error_kind(handle)
    if kind == 1:
        raise CometQueryExecutionException(msg)
    raise CometNativeException(msg)


class Native:
    """The three calls of org.apache.comet.Native (Native.scala:60-111)."""

    @staticmethod
    def createPlan(inputs: Sequence, plan: bytes, config: bytes = b"", partition_count: int = 1, batch_size: int = 8192,
                   device_id: int = 0) -> int:
        l = lib()
        n = len(inputs)
        addrs = (ctypes.c_void_p * max(n, 1))(*[i.address for i in inputs])
        kinds = (ctypes.c_int32 * max(n, 1))(*[i.kind for i in inputs])
        h = l.comet_create_plan(plan, len(plan), config if config else None, len(config), addrs, kinds, n, partition_count,
                                batch_size, device_id)
        if h == 0:
            _raise_last(0)
        return h

    @staticmethod
    def executePlan(handle: int, num_output_cols: int) -> Optional[pa.RecordBatch]:
        """Returns the next output batch, or None at end of stream (executePlan == -1)."""
        l = lib()
        arrays = [ArrowArrayC() for _ in range(num_output_cols)]
        schemas = [ArrowSchemaC() for _ in range(num_output_cols)]
        aaddr = (ctypes.c_void_p * max(num_output_cols, 1))(*[ctypes.addressof(a) for a in arrays])
        saddr = (ctypes.c_void_p * max(num_output_cols, 1))(*[ctypes.addressof(s) for s in schemas])
        rows = l.comet_execute_plan(handle, aaddr, saddr, num_output_cols)
        if rows == -1:
            return None
        if rows < 0:
            _raise_last(handle)
        cols = [pa.Array._import_from_c(ctypes.addressof(a), ctypes.addressof(s)) for a, s in zip(arrays, schemas)]
        if not cols:
            return pa.RecordBatch.from_arrays([], names=[])
        return pa.RecordBatch.from_arrays(cols, names=[f"col_{i}" for i in range(len(cols))])

    @staticmethod
    def executePlanDevice(handle: int, num_output_cols: int) -> Optional["DeviceTable"]:
        """comet_execute_plan_device: the whole result as ONE batch that stays in HBM (None at end of stream).  The returned
        DeviceTable's tensors alias the library's buffers (zero copy) and keep them alive."""
        import torch
        l = lib()
        arrays = [ArrowDeviceArrayC() for _ in range(num_output_cols)]
        schemas = [ArrowSchemaC() for _ in range(num_output_cols)]
        aaddr = (ctypes.c_void_p * max(num_output_cols, 1))(*[ctypes.addressof(a) for a in arrays])
        saddr = (ctypes.c_void_p * max(num_output_cols, 1))(*[ctypes.addressof(s) for s in schemas])
        rows = l.comet_execute_plan_device(handle, aaddr, saddr, num_output_cols)
        if rows == -1:
            return None
        if rows < 0:
            _raise_last(handle)
        owner = _ExportedBatch(arrays)
        fields = [pa.Field._import_from_c(ctypes.addressof(s)).with_name(f"col_{i}") for i, s in enumerate(schemas)]
        device = f"cuda:{arrays[0].device_id}" if arrays else "cuda:0"

        def wrap(ptr, nbytes):
            if not ptr or nbytes == 0:
                return torch.empty(0, dtype=torch.uint8, device=device)
            return torch.as_tensor(_DeviceBuffer(owner, ptr, nbytes), device=device)

        vals, valid, aux = [], [], []
        for a, f in zip(arrays, fields):
            if a.device_type != ARROW_DEVICE_ROCM:
                raise CometNativeException(f"unexpected device type {a.device_type}")
            b = a.array.buffers
            if pa.types.is_string(f.type) or pa.types.is_binary(f.type):
                offs = wrap(b[1], (rows + 1) * 4)
                total = int(offs.view(torch.int32)[-1].item()) if rows else 0
                vals.append(offs)
                aux.append(wrap(b[2], max(total, 1)))
            else:
                w = value_width(f.type)
                vals.append(wrap(b[1], (rows + 7) // 8 if w == 0 else rows * w))
                aux.append(None)
            valid.append(wrap(b[0], (rows + 7) // 8) if b[0] else None)
        return DeviceTable(pa.schema(fields), rows, vals, valid, device, aux)

    @staticmethod
    def releasePlan(handle: int) -> None:
        if _kernel_times_sink is not None:      # measurement only (collect_kernel_times): the plan's per-kernel totals before it goes
            buf = ctypes.create_string_buffer(1 << 16)
            if lib().comet_plan_kernel_times(handle, buf, len(buf)) > 2:
                import json
                for k, v in json.loads(buf.value.decode()).items():
                    e = _kernel_times_sink.setdefault(k, {"ms": 0.0, "calls": 0})
                    e["ms"] += v["ms"]
                    e["calls"] += v["calls"]
        lib().comet_release_plan(handle)


_kernel_times_sink = None


class collect_kernel_times:
    """with collect_kernel_times() as kt: …  — every plan created and released inside the block times each of its generated-kernel launches
    with its own HIP event pair (comet_set_kernel_times); kt.times = {kernel name: {"ms", "calls"}} summed over those plans.  A measurement
    switch for the bench legs' rooflines: the event pairs cost a few microseconds per launch, so timed loops run with it off."""

    def __enter__(self):
        global _kernel_times_sink
        self.times = {}
        _kernel_times_sink = self.times
        lib().comet_set_kernel_times(1)
        return self

    def __exit__(self, *exc):
        global _kernel_times_sink
        lib().comet_set_kernel_times(0)
        _kernel_times_sink = None
        return False


class CometExecIterator:
    """Per-task driver: createPlan → executePlan* → releasePlan (CometExecIterator.scala:109,158,236)."""

    def __init__(self, inputs: Sequence, num_output_cols: int, plan: bytes, config: bytes = b"", batch_size: int = 8192,
                 device_id: int = 0, subqueries=None):
        self._inputs = list(inputs)  # keep the exported streams alive
        self.num_output_cols = num_output_cols
        self.handle = Native.createPlan(self._inputs, plan, config, 1, batch_size, device_id)
        self._closed = False
        # scalar subqueries: {id: value bytes or None for NULL} (comet_plan_set_subquery; what CometScalarSubquery.setSubquery registers on the JVM side)
        for sid, val in (subqueries or {}).items():
            if lib().comet_plan_set_subquery(self.handle, sid, 1 if val is None else 0, val or b"", len(val or b"")) != 0:
                _raise_last(self.handle)

    def __iter__(self) -> Iterator[pa.RecordBatch]:
        return self

    def __next__(self) -> pa.RecordBatch:
        if self._closed:
            raise StopIteration
        b = Native.executePlan(self.handle, self.num_output_cols)
        if b is None:
            self.close()
            raise StopIteration
        return b

    def kernel_stats(self):
        ms, launches, rows = ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
        lib().comet_plan_kernel_stats(self.handle, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(rows))
        return ms.value, launches.value, rows.value

    def kernel_times(self) -> dict:
        """{kernel name: {"ms", "calls"}} of this plan's generated-kernel launches; empty unless COMET_KERNEL_TIMES=1 was set before the library loaded"""
        import json
        buf = ctypes.create_string_buffer(1 << 16)
        n = lib().comet_plan_kernel_times(self.handle, buf, len(buf))
        return json.loads(buf.value.decode()) if n > 0 else {}

    def aux_kernel_stats(self):
        """(ms, launches) of the input-verification kernels (utf8_uniform_kernel) this plan ran ahead of its main kernels"""
        ms, launches = ctypes.c_double(), ctypes.c_int64()
        lib().comet_plan_aux_kernel_stats(self.handle, ctypes.byref(ms), ctypes.byref(launches))
        return ms.value, launches.value

    def metrics(self) -> bytes:
        n = lib().comet_plan_metrics(self.handle, None, 0)
        buf = ctypes.create_string_buffer(max(int(n), 1))
        lib().comet_plan_metrics(self.handle, buf, n)
        return buf.raw[:n]

    def explain(self) -> str:
        return (lib().comet_explain(self.handle) or b"").decode()

    def close(self):
        if not self._closed:
            self._closed = True
            Native.releasePlan(self.handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def execute_to_table(inputs: Sequence, num_output_cols: int, plan: bytes, **kw) -> List[pa.RecordBatch]:
    it = CometExecIterator(inputs, num_output_cols, plan, **kw)
    try:
        return list(it)
    finally:
        it.close()


def execute_to_device(inputs: Sequence, num_output_cols: int, plan: bytes, device_id: int = 0, config: bytes = b"") -> DeviceTable:
    """createPlan → executePlanDevice → releasePlan; the result keeps its buffers alive after the plan is released."""
    import time
    trace = os.environ.get("COMET_TRACE_STAGES")
    keep = list(inputs)
    t0 = time.perf_counter()
    h = Native.createPlan(keep, plan, config, 1, 0, device_id)
    t1 = time.perf_counter()
    try:
        t = Native.executePlanDevice(h, num_output_cols)
        t2 = time.perf_counter()
        if t is None:
            raise CometNativeException("plan produced no device batch")
        return t
    finally:
        Native.releasePlan(h)
        if trace:
            t3 = time.perf_counter()
            print(f"[comet] createPlan {1e3 * (t1 - t0):.2f} ms, executePlanDevice {1e3 * (t2 - t1):.2f} ms, releasePlan {1e3 * (t3 - t2):.2f} ms", flush=True)


def _stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def partition_ids(table: DeviceTable, key_cols: Sequence[int], num_partitions: int):
    """Spark hash partitioning on device: murmur3 (seed 42, chained over key_cols) then pmod (comet_murmur3_column,
    comet_pmod_partition).  Returns an int32 torch tensor of table.num_rows partition ids."""
    import torch
    from . import serde as S
    n = table.num_rows
    hashes = torch.full((max(n, 1),), 42, dtype=torch.int32, device=table.device)   # same bits as u32 42
    st = _stream_ptr()
    for c in key_cols:
        t = table.schema.field(c).type
        tid, prec = S.arrow_type_id(t)
        rc = lib().comet_murmur3_column(tid, prec, table.values[c].data_ptr() if table.values[c].numel() else None,
                                        table.validity[c].data_ptr() if table.validity[c] is not None else None,
                                        table.aux[c].data_ptr() if table.aux[c] is not None else None, n, hashes.data_ptr(), st)
        if rc != 0:
            _raise_last(0)
    pids = torch.empty((max(n, 1),), dtype=torch.int32, device=table.device)
    if lib().comet_pmod_partition(hashes.data_ptr(), n, num_partitions, pids.data_ptr(), st) != 0:
        _raise_last(0)
    return pids[:n]


def partition_table(table: DeviceTable, pids, num_partitions: int):
    """Group the rows of `table` by partition id (comet_partition_indices + comet_take_column per buffer).
    Returns (DeviceTable with rows grouped by partition, partition_starts as a Python list of P+1 ints)."""
    import torch
    n = table.num_rows
    dev = table.device
    st = _stream_ptr()
    starts = torch.empty((num_partitions + 1,), dtype=torch.int64, device=dev)
    idx = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    if lib().comet_partition_indices(pids.data_ptr() if n else None, n, num_partitions, starts.data_ptr(), idx.data_ptr(), st) != 0:
        _raise_last(0)
    vals, valid, aux = [], [], []
    for i, f in enumerate(table.schema):
        if pa.types.is_string(f.type) or pa.types.is_binary(f.type):
            # Utf8: new offsets first (returns the byte total), then the bytes
            vptr = table.validity[i].data_ptr() if table.validity[i] is not None else None
            offs = torch.empty(((n + 1) * 4,), dtype=torch.uint8, device=dev)
            total = lib().comet_take_utf8_offsets(table.values[i].data_ptr() if table.values[i].numel() else None, vptr, idx.data_ptr(), n, offs.data_ptr(), st)
            if total < 0:
                _raise_last(0)
            data = torch.empty((max(int(total), 1),), dtype=torch.uint8, device=dev)
            if n and lib().comet_take_utf8_bytes(table.values[i].data_ptr(), table.aux[i].data_ptr(), vptr, idx.data_ptr(), n, offs.data_ptr(), data.data_ptr(), st) != 0:
                _raise_last(0)
            vals.append(offs)
            aux.append(data)
            if table.validity[i] is not None:
                vb = torch.empty(((n + 7) // 8,), dtype=torch.uint8, device=dev)
                if n and lib().comet_take_column(0, table.validity[i].data_ptr(), idx.data_ptr(), n, vb.data_ptr(), st) != 0:
                    _raise_last(0)
                valid.append(vb)
            else:
                valid.append(None)
            continue
        w = value_width(f.type)
        out = torch.empty(((n + 7) // 8 if w == 0 else n * w,), dtype=torch.uint8, device=dev)
        if n and lib().comet_take_column(w, table.values[i].data_ptr(), idx.data_ptr(), n, out.data_ptr(), st) != 0:
            _raise_last(0)
        vals.append(out)
        aux.append(None)
        if table.validity[i] is not None:
            vb = torch.empty(((n + 7) // 8,), dtype=torch.uint8, device=dev)
            if n and lib().comet_take_column(0, table.validity[i].data_ptr(), idx.data_ptr(), n, vb.data_ptr(), st) != 0:
                _raise_last(0)
            valid.append(vb)
        else:
            valid.append(None)
    torch.cuda.current_stream().synchronize()
    return DeviceTable(table.schema, n, vals, valid, dev, aux), [int(x) for x in starts.cpu().tolist()]


def check_plan(plan: bytes):
    """comet_check_plan: would createPlan accept this plan?  → (True, description) or (False, the refusal naming the operator / expression).
    Nothing is compiled and no GPU is touched."""
    buf = ctypes.create_string_buffer(1 << 16)
    rc = lib().comet_check_plan(plan, len(plan), buf, len(buf))
    return rc == 0, buf.value.decode(errors="replace")


def jit_toolchain() -> str:
    """which compiler this process's createPlan uses (comet_jit_toolchain): "hiprtc X.Y; comgr X.Y /path/libamd_comgr…" """
    f = lib().comet_jit_toolchain
    f.restype = ctypes.c_char_p
    f.argtypes = []
    return (f() or b"").decode()


def compile_plan(plan: bytes) -> str:
    """Decode, plan, generate and hiprtc-compile for gfx950 without a GPU; returns the fused-pipeline text."""
    buf = ctypes.create_string_buffer(1 << 16)
    rc = lib().comet_compile_plan(plan, len(plan), buf, len(buf))
    if rc != 0:
        _raise_last(0)
    return buf.value.decode()


# --------------------------------------------------------------------------- in-library exchange (csrc/exchange.cpp)


class CometExchangeColumnC(ctypes.Structure):
    _fields_ = [("type_id", ctypes.c_int32), ("precision", ctypes.c_int32), ("values", ctypes.c_void_p), ("validity", ctypes.c_void_p),
                ("aux", ctypes.c_void_p)]


class _ExchangeResult:
    """Owns a comet_exchange result; the tensors built over its buffers keep it alive."""

    def __init__(self, handle):
        self.handle = handle

    def __del__(self):
        try:
            lib().comet_exchange_result_release(self.handle)
        except Exception:
            pass


class NativeComm:
    """One rank of an in-library communicator: RCCL between processes (`unique_id` from NativeComm.unique_id() on rank 0, moved to the
    other ranks by the launcher), or the in-process transport between the task threads of one process (`local_group`)."""

    def __init__(self, world: int, rank: int, device_id: int = 0, unique_id: Optional[bytes] = None, local_group: Optional[int] = None,
                 tcp_peers: Optional[str] = None, timeout_ms: int = 0):
        if tcp_peers is not None:
            # one process per rank over TCP ("host:port,host:port,…", one entry per rank): host memory on the wire
            self.handle = lib().comet_comm_init_tcp(tcp_peers.encode(), world, rank, device_id, timeout_ms)
        elif local_group is not None:
            self.handle = lib().comet_comm_init_local(local_group, world, rank, device_id)
        else:
            buf = ctypes.create_string_buffer(unique_id if unique_id is not None else bytes(128), 128)
            self.handle = lib().comet_comm_init_rank(buf, world, rank, device_id)
        if not self.handle:
            raise CometNativeException((lib().comet_exchange_last_error() or b"").decode())
        self.world, self.rank, self.device_id = world, rank, device_id

    @property
    def transport(self) -> str:
        """"rccl" | "tcp" | "in-process" | "none (1 rank)": the wire this communicator moves its slices over"""
        return (lib().comet_comm_transport(self.handle) or b"").decode()

    def stats(self) -> dict:
        """What the wire itself reports: ranks of the communicator (RCCL: ncclCommCount), this rank there, bytes sent to / received from other ranks."""
        out = (ctypes.c_int64 * 4)()
        fn = lib().comet_comm_stats
        fn.restype = ctypes.c_int32
        fn.argtypes = [ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
        if fn(self.handle, out) != 0:
            raise CometNativeException((lib().comet_exchange_last_error() or b"").decode())
        return {"comm_count": int(out[0]), "comm_rank": int(out[1]), "bytes_sent": int(out[2]), "bytes_received": int(out[3])}

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        if lib().comet_comm_unique_id(buf) != 0:
            raise CometNativeException((lib().comet_exchange_last_error() or b"").decode())
        return buf.raw

    def close(self):
        if self.handle:
            lib().comet_comm_destroy(self.handle)
            self.handle = 0

    def exchange(self, table: "DeviceTable", key_cols: Sequence[int]) -> "DeviceTable":
        """Collective hash exchange of this rank's shard on `key_cols` (fixed-width, Boolean, Utf8 / Binary columns); returns
        partition `rank`."""
        import torch
        from . import serde as S
        torch.cuda.current_stream(torch.device(table.device)).synchronize()     # the producer's work is complete before libcomet's stream reads
        n = len(table.values)
        cols = (CometExchangeColumnC * max(n, 1))()
        for i, f in enumerate(table.schema):
            t = S.from_arrow_type(f.type)
            cols[i].type_id, cols[i].precision = t.type_id, t.precision
            cols[i].values = table.values[i].data_ptr() if table.values[i].numel() else None
            cols[i].validity = table.validity[i].data_ptr() if table.validity[i] is not None else None
            cols[i].aux = table.aux[i].data_ptr() if table.aux[i] is not None and table.aux[i].numel() else None
        keys = (ctypes.c_int32 * max(len(key_cols), 1))(*key_cols)
        h = lib().comet_exchange(self.handle, n, cols, table.num_rows, keys, len(key_cols))
        if not h:
            raise CometNativeException((lib().comet_exchange_last_error() or b"").decode())
        owner = _ExchangeResult(h)
        rows = lib().comet_exchange_result_rows(h)
        vals, valid, aux = [], [], []
        empty = lambda: torch.empty(0, dtype=torch.uint8, device=table.device)
        for i, f in enumerate(table.schema):
            pv, pb = ctypes.c_void_p(), ctypes.c_void_p()
            lib().comet_exchange_result_column(h, i, ctypes.byref(pv), ctypes.byref(pb))
            is_str = pa.types.is_string(f.type) or pa.types.is_binary(f.type)
            if is_str:
                nb = (rows + 1) * 4                      # the rebuilt offsets (always rows + 1 of them)
            elif pa.types.is_boolean(f.type):
                nb = (rows + 7) // 8
            else:
                nb = rows * value_width(f.type)
            vals.append(torch.as_tensor(_DeviceBuffer(owner, pv.value, nb), device=table.device) if nb else empty())
            valid.append(torch.as_tensor(_DeviceBuffer(owner, pb.value, (rows + 7) // 8), device=table.device) if (pb.value and rows) else None)
            if is_str:
                pd_, nbytes = ctypes.c_void_p(), ctypes.c_int64()
                lib().comet_exchange_result_aux(h, i, ctypes.byref(pd_), ctypes.byref(nbytes))
                aux.append(torch.as_tensor(_DeviceBuffer(owner, pd_.value, nbytes.value), device=table.device) if nbytes.value
                           else torch.zeros(1, dtype=torch.uint8, device=table.device))
            else:
                aux.append(None)
        return DeviceTable(table.schema, rows, vals, valid, table.device, aux)


def snappy2_inflate_pages(streams, page_lens, device_id: int = 0):
    """The multi-kernel snappy pipeline (csrc/snappy2.cpp) over raw snappy streams held in host memory — comet_snappy2_inflate_pages.
    Returns (pages, kernel_ms, status per page: 0 decoded by the pipeline, 1 by the one-wave fallback)."""
    import numpy as np
    n = len(streams)
    slen = np.array([len(s) for s in streams], np.int32)
    soff = np.zeros(n, np.int64)
    soff[1:] = np.cumsum(slen[:-1], dtype=np.int64)
    blob = np.frombuffer(b"".join(streams) + b"\0", np.uint8)
    plen = np.array(page_lens, np.int32)
    ooff = np.zeros(n, np.int64)
    ooff[1:] = np.cumsum(plen[:-1], dtype=np.int64)
    out = np.zeros(int(plen.sum()) + 1, np.uint8)
    status = np.zeros(max(n, 1), np.uint32)
    ms = ctypes.c_double(0.0)
    f = lib().comet_snappy2_inflate_pages
    f.restype = ctypes.c_int64
    rc = f(ctypes.c_void_p(blob.ctypes.data), ctypes.c_void_p(soff.ctypes.data), ctypes.c_void_p(slen.ctypes.data), ctypes.c_void_p(plen.ctypes.data), ctypes.c_int32(n),
           ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ooff.ctypes.data), ctypes.c_int32(device_id), ctypes.byref(ms), ctypes.c_void_p(status.ctypes.data))
    if rc != 0:
        raise CometNativeException(f"snappy page {rc >> 8}: code {rc & 0xff}" if rc > 0 else "HIP error in comet_snappy2_inflate_pages")
    return [out[int(o):int(o) + int(l)].tobytes() for o, l in zip(ooff, plen)], ms.value, [int(x) for x in status[:n]]


def zstd2_inflate_pages(streams, page_lens, device_id: int = 0):
    """The zstd pipeline (csrc/zstd2.cpp) over single-frame zstd streams held in host memory — comet_zstd2_inflate_pages.
    Returns (pages, kernel_ms, status per page: 0 decoded on the device, 1 kept on the host by the frame walk — its page comes back as zeros);
    raises CometNativeException naming the first corrupt page."""
    import numpy as np
    n = len(streams)
    slen = np.array([len(s) for s in streams], np.int32)
    soff = np.zeros(n, np.int64)
    soff[1:] = np.cumsum(slen[:-1], dtype=np.int64)
    blob = np.frombuffer(b"".join(streams) + b"\0", np.uint8)
    plen = np.array(page_lens, np.int32)
    ooff = np.zeros(n, np.int64)
    ooff[1:] = np.cumsum(plen[:-1], dtype=np.int64)
    out = np.zeros(int(plen.sum()) + 1, np.uint8)
    status = np.zeros(max(n, 1), np.uint32)
    ms = ctypes.c_double(0.0)
    f = lib().comet_zstd2_inflate_pages
    f.restype = ctypes.c_int64
    rc = f(ctypes.c_void_p(blob.ctypes.data), ctypes.c_void_p(soff.ctypes.data), ctypes.c_void_p(slen.ctypes.data), ctypes.c_void_p(plen.ctypes.data), ctypes.c_int32(n),
           ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(ooff.ctypes.data), ctypes.c_int32(device_id), ctypes.byref(ms), ctypes.c_void_p(status.ctypes.data))
    if rc != 0:
        raise CometNativeException(f"zstd page {rc >> 8}: code {rc & 0xff}" if rc > 0 else "HIP error in comet_zstd2_inflate_pages")
    return [out[int(o):int(o) + int(l)].tobytes() for o, l in zip(ooff, plen)], ms.value, [int(x) for x in status[:n]]


def snappy_inflate_pages(streams, page_lens, device_id: int = 0):
    """Run the device snappy kernel (csrc/snappy_kernels.hip) over raw snappy streams held in host memory — the diagnostic entry
    comet_snappy_inflate_pages.  Returns (pages, kernel_ms); raises CometNativeException naming the first corrupt page."""
    import numpy as np
    n = len(streams)
    slen = np.array([len(s) for s in streams], np.int32)
    soff = np.zeros(n, np.int64)
    soff[1:] = np.cumsum(slen[:-1], dtype=np.int64)
    blob = np.frombuffer(b"".join(streams) + b"\0", np.uint8)
    plen = np.array(page_lens, np.int32)
    ooff = np.zeros(n, np.int64)
    ooff[1:] = np.cumsum(plen[:-1], dtype=np.int64)
    out = np.zeros(int(plen.sum()) + 1, np.uint8)
    ms = ctypes.c_double(0.0)
    rc = lib().comet_snappy_inflate_pages(blob.ctypes.data, soff.ctypes.data, slen.ctypes.data, plen.ctypes.data, n, out.ctypes.data, ooff.ctypes.data,
                                          device_id, ctypes.byref(ms))
    if rc != 0:
        raise CometNativeException(f"snappy page {rc >> 8}: code {rc & 0xff}" if rc > 0 else "HIP error in comet_snappy_inflate_pages")
    return [out[int(o):int(o) + int(l)].tobytes() for o, l in zip(ooff, plen)], ms.value


def page_decompress(codec: int, data: bytes, uncompressed_size: int) -> bytes:
    """one Parquet page body through the scan's HOST codecs (comet_page_decompress); codec = Parquet CompressionCodec number"""
    import numpy as np
    out = np.zeros(max(uncompressed_size, 1), np.uint8)
    if lib().comet_page_decompress(codec, data, len(data), out.ctypes.data, uncompressed_size) != 0:
        _raise_last(0)
    return out[:uncompressed_size].tobytes()


def error_json(error_type: str, error_class: str, value_kind: int, lo: int = 0, hi: int = 0, from_type: str = "", to_type: str = "", precision: int = 0, scale: int = 0,
               suffix: str = "", string: bytes = b"") -> dict:
    """the Spark error JSON of one raise site and the value a kernel left (comet_error_json; csrc/err_sites.cpp), parsed"""
    import json
    buf = ctypes.create_string_buffer(4096)
    n = lib().comet_error_json(error_type.encode(), error_class.encode(), from_type.encode(), to_type.encode(), precision, scale, value_kind, suffix.encode(),
                               lo & (2**64 - 1), hi & (2**64 - 1), string, len(string), buf, len(buf))
    assert 0 < n < len(buf)
    return json.loads(buf.value.decode())


def plan_error_json(plan: bytes, site_index: int, lo: int = 0, hi: int = 0, string: bytes = b"") -> dict:
    """the error JSON of the site_index-th raise site with a QueryContext of a plan's pipeline (comet_plan_error_json), parsed"""
    import json
    buf = ctypes.create_string_buffer(1 << 16)
    n = lib().comet_plan_error_json(plan, len(plan), site_index, lo & (2**64 - 1), hi & (2**64 - 1), string, len(string), buf, len(buf))
    if n < 0:
        _raise_last(0)
    return json.loads(buf.value.decode())


def zone_table(zone: str):
    """the (instant, offset) table a time zone is planned with (comet_zone_table): numpy int64 { n, first offset, limit, at[n], off[n] }"""
    import numpy as np
    n = lib().comet_zone_table(zone.encode(), None, 0)
    if n < 0:
        _raise_last(0)
    out = np.zeros(n, np.int64)
    if lib().comet_zone_table(zone.encode(), out.ctypes.data_as(ctypes.c_void_p), n) != n:
        _raise_last(0)
    return out


def rlike_match(pattern: str, value: str) -> bool:
    """compile `pattern` like the planner does for RLike and walk the tables over `value` on the host (comet_rlike_match)"""
    v = value.encode()
    rc = lib().comet_rlike_match(pattern.encode(), v, len(v))
    if rc < 0:
        _raise_last(0)
    return rc == 1


def regexp_extract_host(pattern: str, group: int, value: str):
    """the device's capture matcher on the host (comet_regexp_extract_host): → (matched, the group's text)"""
    v = value.encode()
    a, b = ctypes.c_int32(0), ctypes.c_int32(0)
    rc = lib().comet_regexp_extract_host(pattern.encode(), group, v, len(v), ctypes.byref(a), ctypes.byref(b))
    if rc < 0:
        _raise_last(0)
    return rc == 1, v[a.value:a.value + b.value].decode()


def extract_all_host(pattern: str, group: int, value: str):
    """regexp_extract_all(value, pattern, group) by the device's two passes on the host (comet_extract_all_host)"""
    v = value.encode()
    cap = len(v) + 2
    a, b = (ctypes.c_int32 * cap)(), (ctypes.c_int32 * cap)()
    k = lib().comet_extract_all_host(pattern.encode(), group, v, len(v), a, b, cap)
    if k < 0:
        _raise_last(0)
    return [v[a[i]:a[i] + b[i]].decode() for i in range(k)]


def strfn_host(op: int, value: bytes, a: bytes = b"", b: bytes = b"", k: int = 0):
    """csrc/device/strfn.hpp on the host (comet_strfn_host): ops 1-15 → the result's bytes; 20 crc32 / 21 instr / 22 ascii → the value"""
    if op >= 20:
        return lib().comet_strfn_host(op, value, len(value), a, len(a), b, len(b), k, None, 0)
    n = lib().comet_strfn_host(op, value, len(value), a, len(a), b, len(b), k, None, 0)
    if n < 0:
        _raise_last(0)
    buf = ctypes.create_string_buffer(max(n, 1))
    lib().comet_strfn_host(op, value, len(value), a, len(a), b, len(b), k, buf, n)
    return buf.raw[:n]


def plan_codegen(plan: bytes, has_valid: Sequence[bool]) -> dict:
    """the generated HIP source of a chain over one Scan leaf and its output descriptors (comet_plan_codegen)"""
    import json
    hv = bytes(1 if v else 0 for v in has_valid)
    n = lib().comet_plan_codegen(plan, len(plan), hv, len(hv), None, 0)
    if n < 0:
        _raise_last(0)
    buf = ctypes.create_string_buffer(n + 1)
    lib().comet_plan_codegen(plan, len(plan), hv, len(hv), buf, n + 1)
    return json.loads(buf.value.decode())


def error_site_json(site_id: int, lo: int, hi: int, text: bytes = b"") -> str:
    buf = ctypes.create_string_buffer(1 << 14)
    n = lib().comet_error_site_json(site_id, lo, hi, text, len(text), buf, len(buf))
    if n < 0:
        _raise_last(0)
    return buf.value.decode()


def plan_site_error_json(plan: bytes, site_id: int, lo: int, hi: int, text: bytes = b"") -> str:
    """comet_plan_site_error_json: the raise site's JSON with the QueryContext the plan gives it — check_device_errors' text"""
    buf = ctypes.create_string_buffer(1 << 16)
    n = lib().comet_plan_site_error_json(plan, len(plan), site_id, lo, hi, text, len(text), buf, len(buf))
    if n < 0:
        _raise_last(0)
    return buf.value.decode()


def embedded_header(name: str) -> str:
    n = lib().comet_embedded_header(name.encode(), None, 0)
    if n < 0:
        _raise_last(0)
    buf = ctypes.create_string_buffer(n + 1)
    lib().comet_embedded_header(name.encode(), buf, n + 1)
    return buf.value.decode()


def subquery_value(v, dtype) -> Optional[bytes]:
    """a scalar subquery's value in comet_plan_set_subquery's encoding (include/comet_amd.h)"""
    import struct
    from . import serde as S
    if v is None:
        return None
    t = dtype.type_id
    if t == S.BOOL:
        return b"\x01" if v else b"\x00"
    if t in (S.FLOAT, S.DOUBLE):
        return struct.pack("<d", float(v))
    if t == S.DECIMAL:
        u = int(v)
        return u.to_bytes(max(1, (u.bit_length() + 8) // 8), "big", signed=True)
    if t in (S.STRING, S.BYTES):
        return v.encode() if isinstance(v, str) else bytes(v)
    return struct.pack("<q", int(v))


def split_host(pattern: str, limit: int, value: str):
    """split(value, pattern, limit) by the device's two passes on the host (comet_split_host): → the pieces"""
    v = value.encode()
    cap = len(v) + 2
    a, b = (ctypes.c_int32 * cap)(), (ctypes.c_int32 * cap)()
    k = lib().comet_split_host(pattern.encode(), limit, v, len(v), a, b, cap)
    if k < 0:
        _raise_last(0)
    return [v[a[i]:a[i] + b[i]].decode() for i in range(k)]


def date_fn_host(fn: int, a: int, b: int = 0, c: int = 0):
    """the generated kernels' calendar functions on the host (comet_date_fn_host): → the value, or None for NULL"""
    out = ctypes.c_int64(0)
    rc = lib().comet_date_fn_host(fn, a, b, c, ctypes.byref(out))
    if rc < 0:
        _raise_last(0)
    return out.value if rc == 1 else None


def parquet_host_plain_values(plan: bytes, column: int) -> bytes:
    """PLAIN value bytes the scan stages on the host for one column of the plan's NativeScan (comet_parquet_host_plain_values; host only)"""
    n = lib().comet_parquet_host_plain_values(plan, len(plan), column, None, 0)
    if n < 0:
        _raise_last(0)
    buf = ctypes.create_string_buffer(max(int(n), 1))
    if lib().comet_parquet_host_plain_values(plan, len(plan), column, buf, n) < 0:
        _raise_last(0)
    return buf.raw[:n]


def xxh64(data: bytes, seed: int = 0) -> int:
    l = lib()
    l.comet_xxh64.restype = ctypes.c_uint64
    l.comet_xxh64.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]
    return l.comet_xxh64(data, len(data), seed)


def sbbf_might_contain(bitset: bytes, h: int) -> bool:
    l = lib()
    l.comet_sbbf_might_contain.restype = ctypes.c_int32
    l.comet_sbbf_might_contain.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64]
    return bool(l.comet_sbbf_might_contain(bitset, len(bitset), h))


def parquet_prune_report(plan: bytes, page_index: bool = True, bloom_filters: bool = True) -> dict:
    """row-group / Bloom-filter / page-index selection of the plan's NativeScan (comet_parquet_prune_report; host only)"""
    import json
    buf = ctypes.create_string_buffer(1 << 20)
    n = lib().comet_parquet_prune_report(plan, len(plan), (1 if page_index else 0) | (0 if bloom_filters else 2), buf, len(buf))
    if n < 0:
        _raise_last(0)
    return json.loads(buf.value.decode())
