"""TEST INFRASTRUCTURE: run the per-row code the generator writes for a Filter / Projection chain — and every device helper it calls — on the HOST.

comet_plan_codegen hands out the HIP source of a plan; it is compiled with g++ against the very header texts hiprtc uses (comet_embedded_header), made host-
compilable by tests/emu/hip_host_shim.hpp (qualifiers and intrinsics as no-ops / one-lane stand-ins) and two textual patches (the address-space qualifiers).
The kernel BODIES are not emulated — ballots, LDS tables and the ordered compaction only have to compile; the driver below walks the rows itself: keep_tile for
every thread slot of a tile, positions in row order (what the device's single-pass compaction produces), emit_tile.  The executor's part behind the kernel
(gathering Utf8 values by row index, assembling string views) is done here in Python for the output kinds that need it; kinds the emulator does not know
raise Unsupported.  Nothing under datafusion-comet_amd/ imports this; it is the CPU suite's way to check generated code against the oracle without a GPU."""
import ctypes
import hashlib
import os
import re
import subprocess
import threading
import tempfile

import numpy as np
import pyarrow as pa
import pyarrow.compute

from datafusion_comet_amd import native, serde as S

HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = None
_CACHE = {}


class Unsupported(Exception):
    pass


_RUN_LOCK = threading.Lock()


DRIVER = r"""
// ---- aggregate sinks: the generated per-row code (keys, private words, fold, combine, emit / finalize) with a std::map where the device has its LDS and global hash
// tables.  A tile's rows meet in a tile-local table of PRIVATE words (P::pop), which is folded (P::fold) into the contribution the global accumulators take
// (P::op) — the two steps of the device's block; the kernels' own machinery (slots, tickets, rehash, float exponent windows) is not run.
typedef std::map<std::vector<u64>, std::vector<u64>> EmuTable;
static EmuTable* g_emu_local = nullptr;
template <class P, class G, class K, class V>
static void emu_group_update(const G&, bool active, const K* key, const V* pv) {
  if (!active) return;
  std::vector<u64> k(key, key + P::NK);
  auto it = g_emu_local->find(k);
  if (it == g_emu_local->end()) {
    std::vector<u64> w(P::NPW);
    for (int i = 0; i < P::NPW; i++) w[i] = P::pidentity(i);
    it = g_emu_local->emplace(k, w).first;
  }
  for (int i = 0; i < P::NPW; i++) it->second[i] = comet::pword_combine<P>(i, it->second[i], (u64)pv[i]);
}
template <class P>
static long long emu_grouped(CometKParams& prm) {
  constexpr i64 kRows = (i64)P::R * comet::kBlock;
  EmuTable global, local;
  std::vector<std::vector<u64>> order;
  comet::GroupCtx<P> grp{};
  u64 kacc[P::NKW > 0 ? P::NKW : 1];
  P::kinit(kacc);
  g_emu_local = &local;
  for (i64 base = 0; base < prm.n; base += kRows) {
    local.clear();
    for (unsigned t = 0; t < 256; t++) {
      threadIdx.x = t;
      typename P::L ld;
      P::tile_load(prm, base, prm.n, ld);
      P::tile_grouped(prm, base, prm.n, ld, grp, kacc);
    }
    for (auto& kv : local) {
      auto it = global.find(kv.first);
      if (it == global.end()) {
        std::vector<u64> acc(P::NW);
        P::init(acc.data());
        it = global.emplace(kv.first, acc).first;
        order.push_back(kv.first);
      }
      u64 val[P::NW];
      P::fold(kv.second.data(), val);
      comet::slot_apply_private<P>(it->second.data(), val);
    }
  }
  threadIdx.x = 0;
  // the kernel-level accumulators (value bounds for the overflow proof) land behind the error block's flags, as the kernel's last step leaves them
  { unsigned long long* aux = (unsigned long long*)prm.out[2] + 2; for (int k = 0; k < P::NKW; k++) aux[k] = P::kop(k) == comet::G_UMAX64 ? (kacc[k] > aux[k] ? kacc[k] : aux[k]) : (aux[k] | kacc[k]); }
  i64 pos = 0;
  for (auto& k : order) P::emit_group(prm, k.data(), global[k].data(), pos++);
  return pos;
}
template <class P>
static long long emu_ungrouped(CometKParams& prm) {
  constexpr i64 kRows = (i64)P::R * comet::kBlock;
  u64 acc[P::NW];
  P::init(acc);
  for (i64 base = 0; base < prm.n; base += kRows)
    for (unsigned t = 0; t < 256; t++) {
      threadIdx.x = t;
      u64 a[P::NW];
      P::init(a);
      typename P::L ld;
      P::tile_load(prm, base, prm.n, ld);
      P::tile(prm, base, prm.n, ld, a);
      P::combine(acc, a);
    }
  threadIdx.x = 0;
  P::kexport(prm, acc);
  P::finalize(prm, acc);
  return 1;
}

template <class P>
static long long emu_impl(const CometKParams* prm_in) {
  CometKParams prm = *prm_in;
  const i64 n = prm.n;
  if constexpr (requires { P::NPW; P::emit_group(prm, (const u64*)nullptr, (const u64*)nullptr, (i64)0); }) {
    if constexpr (requires(typename P::L& ld, const comet::GroupCtx<P>& g) { P::tile_grouped(prm, (i64)0, n, ld, g, (u64*)nullptr); }) return emu_grouped<P>(prm);
    else return -1;
  } else if constexpr (requires { P::finalize(prm, (const u64*)nullptr); }) {
    if constexpr (requires(typename P::L& ld) { P::tile(prm, (i64)0, n, ld, (u64*)nullptr); }) return emu_ungrouped<P>(prm);
    else return -1;
  } else if constexpr (requires { P::keep_tile(prm, (i64)0, n, (bool*)nullptr); }) {
    constexpr int R = P::R;
    constexpr i64 kRows = (i64)R * comet::kBlock;
    i64 total = 0;
    static bool keep[R * 256];
    static i64 posn[R * 256];
    for (i64 base = 0; base < n; base += kRows) {
      for (unsigned t = 0; t < 256; t++) {
        threadIdx.x = t;
        bool k[R];
        P::keep_tile(prm, base, n, k);
        for (int r = 0; r < R; r++) keep[r * 256 + t] = k[r];
      }
      for (int s = 0; s < R * 256; s++) { posn[s] = total; total += keep[s] ? 1 : 0; }      // slot order (r, t) IS row order
      for (unsigned t = 0; t < 256; t++) {
        threadIdx.x = t;
        bool k[R];
        i64 idx[R], pos[R];
        for (int r = 0; r < R; r++) { k[r] = keep[r * 256 + t]; idx[r] = base + (i64)r * 256 + t; pos[r] = posn[r * 256 + t]; }
        P::emit_tile(prm, k, idx, pos);
      }
    }
    return total;
  } else {
    threadIdx.x = 0;
    for (i64 i = 0; i < n; i++) P::emit(prm, i, i);
    return n;
  }
}
// (every plan's library defines a struct P, its statics and emu_run: hidden visibility, no unique symbols, symbolic binding keep each library to itself)
extern "C" __attribute__((visibility("default"))) long long emu_run(const CometKParams* prm_in) { return emu_impl<P>(prm_in); }
"""


def _workdir():
    global _DIR
    if _DIR is None:
        _DIR = tempfile.mkdtemp(prefix="comet_emu_")
        dev = native.embedded_header("comet_device.hpp")
        dev = re.sub(r"#define COMET_GLOBAL [^\n]*", "#define COMET_GLOBAL", dev)
        dev = re.sub(r"#define COMET_LDS [^\n]*", "#define COMET_LDS", dev)
        dev = dev.replace("template <> struct as_i64<COMET_LDS u64*> { typedef COMET_LDS i64* type; };", "")
        dev = dev.replace('__asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");', "")      # (the aggregate kernels' bodies only have to COMPILE here)
        open(os.path.join(_DIR, "comet_device.hpp"), "w").write(dev)
        for name in ("kparams.h", "comet_ryu.hpp", "comet_strtod.hpp", "comet_strts.hpp", "comet_regex_vm.hpp"):
            open(os.path.join(_DIR, name), "w").write(native.embedded_header(name))
    return _DIR


_FLAGS = ["-std=c++20", "-O1", "-fPIC", "-w", "-ffp-contract=off", "-fvisibility=hidden", "-fno-gnu-unique"]
_PCH = {}


def _prefix_header(nt: bool) -> str:
    """the shim, the containers the aggregate driver uses and the device header, precompiled once per process (a plan then compiles in 0.4 s instead of 0.65):
    with COMET_LD_NT = 1 for the sources that define it (aggregate sinks), without for the others — what the source's own first lines would have done"""
    name = "emu_prefix_nt.hpp" if nt else "emu_prefix.hpp"
    if name not in _PCH:
        d = _workdir()
        hdr = os.path.join(d, name)
        open(hdr, "w").write('#include "hip_host_shim.hpp"\n#include <map>\n#include <vector>\n' + ("#define COMET_LD_NT 1\n" if nt else "") + '#include "comet_device.hpp"\n')
        r = subprocess.run(["g++"] + _FLAGS + ["-I", HERE, "-I", d, "-x", "c++-header", hdr, "-o", hdr + ".gch"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("the device header does not compile for the host:\n" + r.stderr[:3000])
        _PCH[name] = True
    return name


def _compiled(source: str):
    key = hashlib.sha1(source.encode()).hexdigest()[:20]
    if key not in _CACHE:
        d = _workdir()
        cpp = os.path.join(d, key + ".cpp")
        # (the grouped sink's rows go to the driver's table instead of the device's: declared ahead of the generated struct, defined behind it)
        prefix = ('#include "%s"\n' % _prefix_header("#define COMET_LD_NT 1" in source) +
                  'template <class P, class G, class K, class V> static void emu_group_update(const G&, bool, const K*, const V*);\n')
        open(cpp, "w").write(prefix + source.replace("comet::group_update<P>(", "emu_group_update<P>(") + DRIVER)
        so = os.path.join(d, key + ".so")
        r = subprocess.run(["g++"] + _FLAGS + ["-shared", "-Wl,-Bsymbolic", "-I", HERE, "-I", d, "-x", "c++", cpp, "-o", so], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("the generated source does not compile for the host:\n" + "\n".join(l for l in r.stderr.splitlines() if "error" in l)[:3000])
        lib = ctypes.CDLL(so)
        lib.emu_run.restype = ctypes.c_longlong
        lib.emu_run.argtypes = [ctypes.c_void_p]
        _CACHE[key] = lib
    return _CACHE[key]


class _Col(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("valid", ctypes.c_void_p), ("aux", ctypes.c_void_p), ("offset", ctypes.c_longlong)]


class _Params(ctypes.Structure):
    _fields_ = [("n", ctypes.c_longlong), ("iarg", ctypes.c_longlong * 7), ("inp", _Col * 24), ("out", ctypes.c_void_p * 48)]


_NP = {S.BOOL: np.uint8, S.INT8: np.int8, S.INT16: np.int16, S.INT32: np.int32, S.DATE: np.int32, S.INT64: np.int64, S.TIMESTAMP: np.int64, S.TIMESTAMP_NTZ: np.int64, S.FLOAT: np.float32,
       S.DOUBLE: np.float64}
_PA = {S.BOOL: pa.bool_(), S.INT8: pa.int8(), S.INT16: pa.int16(), S.INT32: pa.int32(), S.DATE: pa.date32(), S.INT64: pa.int64(), S.TIMESTAMP: pa.timestamp("us", tz="UTC"),
       S.TIMESTAMP_NTZ: pa.timestamp("us"), S.FLOAT: pa.float32(), S.DOUBLE: pa.float64()}


def _addr(buf):
    return buf.address if buf is not None else None


def run_chain(plan, table: pa.Table) -> pa.Table:
    """Evaluate a Filter / Projection chain over ONE Scan leaf on the host, through the generated code.  Output kinds beyond fixed-width values, gathered Utf8
    columns and plain string views raise Unsupported."""
    cols = [table.column(i).combine_chunks() if isinstance(table.column(i), pa.ChunkedArray) else table.column(i) for i in range(table.num_columns)]
    # struct fields and the elements of lists of flat values are columns of their own behind the real ones (exec.cpp extend_struct_fields); a field is NULL where
    # its struct is (what the executor's import makes of pyarrow's layout)
    virt = []
    for c in cols:
        if pa.types.is_list(c.type) and not pa.types.is_nested(c.type.value_type):
            virt.append(c.values)
        elif pa.types.is_struct(c.type):
            for k in range(c.type.num_fields):
                f = c.field(k)
                if c.null_count:
                    f = pa.compute.if_else(c.is_valid(), f, pa.scalar(None, f.type))
                virt.append(f)
    bound = cols + virt
    has_valid = [c.null_count > 0 for c in bound]
    desc = native.plan_codegen(plan if isinstance(plan, (bytes, bytearray)) else plan.encode(), has_valid)
    if desc["sink"] not in (0, 1, 2):
        raise Unsupported("sink %d is not emulated" % desc["sink"])
    if desc["derived"]:
        raise Unsupported("derived columns are computed by the executor's own kernels")
    lib = _compiled(desc["source"])
    n = table.num_rows
    prm = _Params()
    prm.n = n
    keep = []
    if len(bound) > 24:
        raise Unsupported("more than COMET_MAX_IN columns")
    for i, c in enumerate(bound):
        bufs = c.buffers()
        t = c.type
        prm.inp[i].offset = c.offset
        prm.inp[i].valid = _addr(bufs[0]) if has_valid[i] else None
        if pa.types.is_string(t) or pa.types.is_binary(t):
            prm.inp[i].data = _addr(bufs[1])
            prm.inp[i].aux = _addr(bufs[2]) if bufs[2] is not None else ctypes.addressof(ctypes.create_string_buffer(1))
        elif pa.types.is_list(t):
            prm.inp[i].data = _addr(bufs[1])      # the offsets; the elements are a column of their own
        elif pa.types.is_struct(t):
            prm.inp[i].data = None
        elif pa.types.is_nested(t) or pa.types.is_dictionary(t):
            raise Unsupported("map / dictionary inputs")
        else:
            prm.inp[i].data = _addr(bufs[1])
        keep.append((c, bufs))
    cols = bound
    errbuf = np.zeros(512, np.uint8)
    scratch0, scratch1 = np.zeros(8 * (n // 2048 + 4), np.uint8), np.zeros(64, np.uint8)
    prm.out[0], prm.out[1], prm.out[2] = scratch0.ctypes.data, scratch1.ctypes.data, errbuf.ctypes.data
    outs = []
    for j, oc in enumerate(desc["out"]):
        width = 16 if (oc["view_src"] >= 0 or oc["fmt_kind"] or oc["type"] == S.DECIMAL or oc["packed_string"]) else 4 if (oc["gather_src"] >= 0 or oc["concat"]) else np.dtype(_NP[oc["type"]]).itemsize
        vals = np.zeros((n + 1) * width + 16, np.uint8)
        ok = np.ones(n + 16, np.uint8)
        prm.out[4 + 2 * j] = vals.ctypes.data
        prm.out[5 + 2 * j] = ok.ctypes.data
        outs.append((vals, ok, width))
    # exact Float64 sums: the executor's scale pass (exec_pipeline.cpp: run, read the exponent range the kernel left in the aux words, move the fixed-point window,
    # run again — three times at most; one chunk here, so nothing has been accumulated before and the window may move down as well as up)
    FIX_W, scales = 158, [-94] * len(desc.get("fix_sums", []))
    for attempt in range(4):
        prm.iarg[6] = sum((sc & 0xffff) << (16 * f) for f, sc in enumerate(scales[:4]))
        prm.iarg[4] = sum((sc & 0xffff) << (16 * f) for f, sc in enumerate(scales[4:8]))      # (sums 4-7: codegen.hpp kFixScaleArg2)
        with _RUN_LOCK:      # (threadIdx and the aggregate driver's table are globals of the shim: one emulated launch at a time)
            rows = lib.emu_run(ctypes.byref(prm))
        if not scales or attempt >= 3:
            break
        aux = errbuf[16:].view(np.uint64)
        moved = False
        for f, fs in enumerate(desc["fix_sums"]):
            hi, lo = int(aux[fs["aux_hi"]]), int(aux[fs["aux_lo"]])
            if hi == 0:
                continue
            top, low, sc = hi - 1200, 1200 - lo, scales[f]
            target = top + 10 - FIX_W if top > sc + FIX_W else ((low if top - low <= FIX_W - 10 else top + 2 - FIX_W) if low < sc else sc)
            target = max(target, -1300)
            if target != sc:
                scales[f], moved = target, True
        if not moved:
            break
    if rows < 0:
        raise Unsupported("an aggregate sink of another shape than tile / tile_grouped")
    flags = int(errbuf[:4].view(np.uint32)[0])
    if flags:
        _raise_like_the_executor(flags, errbuf, plan)
    arrays = []
    for (vals, ok, width), oc in zip(outs, desc["out"]):
        mask = None if not oc["nullable"] else (ok[:rows] == 0)
        if mask is not None and not mask.any():
            mask = None
        if oc["fmt_kind"] or oc["concat"] or oc["case_mode"] or oc["pad"]:
            raise Unsupported("output columns the executor formats / concatenates / case-maps / pads")
        if oc["packed_string"]:      # str16: bytes 0-14 and the length in the last byte (what the executor expands into offsets + data, exec_pipeline.cpp)
            raw = vals[:rows * 16].reshape(-1, 16)
            arrays.append(pa.array([None if (mask is not None and mask[r]) else raw[r, :raw[r, 15]].tobytes().decode() for r in range(rows)], pa.utf8()))
            continue
        if oc["gather_src"] >= 0:
            src = cols[oc["gather_src"]].to_pylist()
            idx = vals[:rows * 4].view(np.uint32)
            arrays.append(pa.array([None if (mask is not None and mask[r]) else src[idx[r]] for r in range(rows)], cols[oc["gather_src"]].type))
        elif oc["view_src"] >= 0:
            src = [None if v is None else v.encode() for v in cols[oc["view_src"]].to_pylist()]
            v = vals[:rows * 16].view(np.uint32).reshape(-1, 4)      # strview {row, start, len, pad}
            arrays.append(pa.array([None if (mask is not None and mask[r]) else src[v[r, 0]][v[r, 1]:v[r, 1] + v[r, 2]].decode() for r in range(rows)], pa.utf8()))
        elif oc["type"] == S.DECIMAL:
            raw = vals[:rows * 16].tobytes()
            vb = None if mask is None else pa.py_buffer(np.packbits(~mask, bitorder="little").tobytes())
            arrays.append(pa.Array.from_buffers(pa.decimal128(oc["precision"], oc["scale"]), rows, [vb, pa.py_buffer(raw)], null_count=0 if mask is None else int(mask.sum())))
        elif oc["type"] == S.BOOL:
            arrays.append(pa.array(vals[:rows].astype(bool), pa.bool_(), mask=mask))
        else:
            x = vals[:rows * width].view(_NP[oc["type"]])
            base = {S.DATE: pa.int32(), S.TIMESTAMP: pa.int64(), S.TIMESTAMP_NTZ: pa.int64()}.get(oc["type"])
            a = pa.array(x, base or _PA[oc["type"]], mask=mask)
            arrays.append(a.cast(_PA[oc["type"]]) if base else a)
    return pa.table(arrays, names=[f"col_{i}" for i in range(len(arrays))])


_FLAG_JSON = [(1, '{"errorType":"ArithmeticOverflow","errorClass":"ARITHMETIC_OVERFLOW","params":{"fromType":"decimal"}}'),
              (2, '{"errorType":"ArithmeticOverflow","errorClass":"ARITHMETIC_OVERFLOW","params":{"fromType":"integer"}}'), (4, '{"errorType":"CastOverFlow","errorClass":"CAST_OVERFLOW","params":{}}'),
              (8, '{"errorType":"NumericValueOutOfRange","errorClass":"NUMERIC_VALUE_OUT_OF_RANGE.WITH_SUGGESTION","params":{}}'),
              (512, '{"errorType":"CastInvalidValue","errorClass":"CAST_INVALID_INPUT","params":{"fromType":"STRING"}}'),
              (1024, '{"errorType":"InvalidInputInCastToDatetime","errorClass":"CAST_INVALID_INPUT","params":{"fromType":"STRING","toType":"DATE"}}'),
              (8192, '{"errorType":"InvalidInputInCastToDatetime","errorClass":"CAST_INVALID_INPUT","params":{"fromType":"STRING","toType":"TIMESTAMP"}}'),
              (16384, '{"errorType":"InvalidInputInCastToDatetime","errorClass":"CAST_INVALID_INPUT","params":{"fromType":"STRING","toType":"TIMESTAMP_NTZ"}}'),
              (256, '{"errorType":"DivideByZero","errorClass":"DIVIDE_BY_ZERO","params":{}}'), (32768, '{"errorType":"RemainderByZero","errorClass":"REMAINDER_BY_ZERO","params":{}}')]


def _raise_like_the_executor(flags, block, plan):
    """exec_pipeline.cpp check_device_errors, for the flags a Filter / Projection chain can raise: the site's JSON when a site left its detail, else the flag's"""
    detail = block[192:].view(np.uint64)
    if detail[0] != 0:
        raise native.CometQueryExecutionException(native.plan_site_error_json(plan, int(detail[0]) - 1, int(detail[1]), int(detail[2]), bytes(block[192 + 32:192 + 32 + 224])))
    for bit, js in _FLAG_JSON:
        if flags & bit:
            raise native.CometQueryExecutionException(js)
    if flags & (65536 | 131072):      # the ANSI decimal sum / average of an aggregate sink overflowed (decimal_sum_overflow_json with the aggregate's SQL context)
        import json
        raise native.CometQueryExecutionException(json.dumps(native.plan_error_json(plan, -1 if flags & 65536 else -2), separators=(",", ":"), ensure_ascii=False))
    if flags & 64:
        raise Unsupported("Utf8 group keys longer than 15 bytes take another path of the executor")
    if flags & 262144:
        raise native.CometNativeException("Arrow error: Compute error: long overflow")
    if flags & 4096:
        raise native.CometNativeException("a string cast to a timestamp names a time zone inside the value, or holds a time of day without a date (which takes the current date): not supported by the MI355X native engine")
    if flags & 2048:
        raise native.CometNativeException("a timestamp lies behind the end of its time zone's table (the year 2400): not supported by the MI355X native engine")
    raise DeviceError(flags, block)


class DeviceError(Exception):
    """the generated code raised an error flag (what check_device_errors turns into a Spark error on the device path)"""

    def __init__(self, flags, block):
        super().__init__("device error flags %d" % flags)
        self.flags = flags
        self.block = block


class _HostInput:
    def __init__(self, table):
        self.table = table

    @staticmethod
    def from_table(table, batch_rows=8192):
        return _HostInput(table)


def run_gpu_test_on_host(module: str, fn: str, **params):
    """Run a GPU parity test's Python with the EMULATOR standing in for the device: the plan's generated code, compiled for the host, evaluates the rows; what the
    test compares it with (the oracle) is untouched.  → "ok", or raises (Unsupported: the plan needs something the emulator does not do)."""
    import importlib
    mod = importlib.import_module(module)
    saved = (native.HostInput, native.execute_to_table)

    def execute(inputs, ncols, plan_bytes, **kw):
        if kw.get("subqueries"):
            raise Unsupported("scalar subqueries are resolved by the executor")
        if len(inputs) != 1:
            raise Unsupported("plans with several inputs")
        out = run_chain(plan_bytes, inputs[0].table)
        assert out.num_columns == ncols, (out.num_columns, ncols)
        return out.to_batches(max_chunksize=kw.get("batch_size", 8192) or None) if out.num_rows else []      # spark.comet.batchSize bounds an output batch

    native.HostInput, native.execute_to_table = _HostInput, execute
    try:
        getattr(mod, fn)(None, **params)
    finally:
        native.HostInput, native.execute_to_table = saved
    return "ok"
