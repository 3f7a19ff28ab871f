// extern "C" boundary (include/comet_amd.h).  Every entry catches all C++ exceptions: the reference
// wraps each JNI entry in try_unwrap_or_throw (native/jni-bridge/src/errors.rs:832-850).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>

#include "../../include/comet_amd.h"
#include "exec.hpp"
#include "shuffle_format.hpp"
#include "parquet_meta.hpp"
#include "regex.hpp"
namespace comet {
std::vector<DType> extend_struct_field_types(const std::vector<DType>& types);      // exec.cpp
}
// (jit.cpp: the header texts hiprtc compiles against)
namespace comet {
extern const char* const kEmbeddedDeviceHeader;
extern const char* const kEmbeddedKParamsHeader;
extern const char* const kEmbeddedRyuHeader;
extern const char* const kEmbeddedStrtodHeader;
extern const char* const kEmbeddedStrtsHeader;
extern const char* const kEmbeddedRegexVmHeader;
}
// (device/strfn.hpp, as the host sees it: comet_strfn_host below)
namespace comet_strfn_host_ns {
#include "device/strfn.hpp"
}
#include "row_shuffle.hpp"
#include "tz.hpp"

using namespace comet;

// kernels_static.hip
extern "C" int comet_launch_murmur3(int type_id, int precision, const void* values, const uint8_t* validity,
                                    const void* aux, int64_t n, uint32_t* hashes, void* stream);
extern "C" int comet_launch_pmod(const uint32_t* hashes, int64_t n, int32_t np, int32_t* out, void* stream);

// exchange_kernels.hip
extern "C" int64_t comet_partition_tiles(int64_t n);
extern "C" int64_t comet_partition_scratch_bytes(int64_t n, int32_t P);
extern "C" int comet_launch_partition_indices(const int32_t* pids, int64_t n, int32_t P, uint64_t* hist, uint32_t* bad, int64_t* starts,
                                              uint32_t* row_indices, void* stream);
extern "C" int comet_launch_take(int width, const void* src, const uint32_t* idx, int64_t n, void* dst, void* stream);

extern "C" int comet_launch_take_utf8_lengths(const int32_t* offs, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits, int64_t n,
                                              uint32_t* lengths, void* stream);
extern "C" int comet_launch_take_utf8_copy(const int32_t* offs, const uint8_t* bytes, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits,
                                           int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream);
extern "C" void pq_launch_u32_scan(const uint32_t* in, int64_t n, uint64_t* tiles, int32_t* out, void* st);

namespace {

std::mutex g_mu;
std::map<int64_t, std::shared_ptr<ExecutionContext>> g_ctx;
int64_t g_next = 1;
thread_local std::string t_last_error;
thread_local int t_last_kind = 0;

uint64_t plan_bytes_hash(const uint8_t* plan, size_t plan_len) {   // FNV-1a: key of the process-wide plan cache
  uint64_t ph = 0xcbf29ce484222325ull;
  for (size_t i = 0; i < plan_len; i++) { ph ^= plan[i]; ph *= 0x100000001b3ull; }
  return ph ^ ((uint64_t)plan_len << 48);
}

std::shared_ptr<ExecutionContext> lookup(int64_t h) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_ctx.find(h);
  return it == g_ctx.end() ? nullptr : it->second;
}

template <class F>
auto guarded(ExecutionContext* ctx, decltype(std::declval<F>()()) err_value, F f) -> decltype(f()) {
  try {
    return f();
  } catch (const CometError& e) {
    if (ctx) { ctx->last_error = e.what(); ctx->last_error_kind = e.kind; }
    t_last_error = e.what();
    t_last_kind = e.kind;
  } catch (const std::exception& e) {
    if (ctx) { ctx->last_error = e.what(); ctx->last_error_kind = 0; }
    t_last_error = e.what();
    t_last_kind = 0;
  } catch (...) {
    if (ctx) { ctx->last_error = "unknown native error"; ctx->last_error_kind = 0; }
    t_last_error = "unknown native error";
    t_last_kind = 0;
  }
  return err_value;
}

}  // namespace


namespace comet { namespace detail {
void plan_execution_begins(); void plan_execution_ends(); int plans_executing();
bool nested_schema_matches(const ArrowSchema* f, const DType& t);
void append_nested_rows(HostColumn& dst, const ArrowArray* a, const DType& t, int64_t off, int64_t len);
} }
// COMET_TRACE_STAGES: every plan call with its begin on the process clock (the same clock the scan traces use), so that the calls of
// concurrent tasks can be laid side by side
struct ApiTrace {
  const char* what; int64_t handle; double t0; bool on;
  ApiTrace(const char* w, int64_t h) : what(w), handle(h), t0(0), on(false) {
    static const bool trace = getenv("COMET_TRACE_STAGES") != nullptr;
    on = trace;
    if (on) t0 = comet::process_clock_ms();
  }
  ~ApiTrace() { if (on) fprintf(stderr, "[comet] api: %s of plan %lld began at %.2f ms of the process clock, took %.2f ms\n", what, (long long)handle, t0, comet::process_clock_ms() - t0); }
};

extern "C" {

int64_t comet_create_plan(const uint8_t* plan, size_t plan_len, const uint8_t* config, size_t config_len, void** inputs,
                          const int32_t* input_kinds, int32_t n_inputs, int32_t partition_count, int32_t batch_size,
                          int32_t device_id) {
  (void)partition_count;
  ApiTrace api_trace("createPlan", 0);
  return guarded(nullptr, (int64_t)0, [&]() -> int64_t {
    // Ownership of every input stream passes to the library with this call (the reference takes the C structs over as soon as it has
    // their addresses): whatever fails below — plan decoding, an unknown input kind, planning itself — each stream is released once.
    struct Guard {
      void** raw; const int32_t* kinds; int32_t n;
      std::vector<InputSource> ins;      // streams already wrapped (a shuffle-block wrapper owns its block stream)
      int32_t wrapped = 0;               // raw inputs [0, wrapped) are represented in `ins`
      bool armed = true;
      ~Guard() {
        if (!armed) return;
        for (auto& s : ins) {
          if (s.host && s.host->release) s.host->release(s.host);
          if (s.dev && s.dev->release) s.dev->release(s.dev);
        }
        for (int32_t i = wrapped; i < n; i++) {
          const int32_t k = kinds ? kinds[i] : 0;
          if (!raw || !raw[i]) continue;
          if (k == COMET_INPUT_HOST_STREAM) { auto* h = (ArrowArrayStream*)raw[i]; if (h->release) h->release(h); }
          else if (k == COMET_INPUT_DEVICE_STREAM) { auto* d = (ArrowDeviceArrayStream*)raw[i]; if (d->release) d->release(d); }
          else if (k == COMET_INPUT_SHUFFLE_BLOCKS) { auto* b = (comet::CometShuffleBlockStreamC*)raw[i]; if (b->release) b->release(b); }
        }
      }
    } guard{inputs, input_kinds, n_inputs};
    if (!plan || plan_len == 0) throw CometError("empty plan");
    OperatorP op = decode_operator(plan, plan_len);
    auto cfg = (config && config_len) ? decode_config_map(config, config_len) : std::vector<std::pair<std::string, std::string>>();
    std::vector<InputSource>& ins = guard.ins;
    for (int i = 0; i < n_inputs; i++) {
      InputSource s;
      s.kind = input_kinds ? input_kinds[i] : 0;
      if (s.kind == COMET_INPUT_HOST_STREAM) s.host = (ArrowArrayStream*)inputs[i];
      else if (s.kind == COMET_INPUT_DEVICE_STREAM) s.dev = (ArrowDeviceArrayStream*)inputs[i];
      else if (s.kind == COMET_INPUT_SHUFFLE_BLOCKS) {
        // a ShuffleScan leaf: blocks are decoded on the host into Arrow batches and then take the host-stream path
        std::vector<const Operator*> leaves;
        std::function<void(const Operator&)> walk = [&](const Operator& o) {
          if (o.kind == OpKind::Scan) leaves.push_back(&o);
          for (auto& c : o.children) walk(*c);
        };
        walk(*op);
        if ((size_t)i >= leaves.size()) throw CometError("more inputs than Scan leaves");
        auto* bs = (comet::CometShuffleBlockStreamC*)inputs[i];   // same layout as struct CometShuffleBlockStream
        s.kind = COMET_INPUT_HOST_STREAM;
        s.host = shuffle_blocks_as_arrow_stream(bs, leaves[(size_t)i]->scan_fields);
      } else throw CometError("unknown input kind " + std::to_string(s.kind));
      ins.push_back(s);
      guard.wrapped = i + 1;
    }
    std::shared_ptr<ExecutionContext> ctx = std::make_shared<ExecutionContext>(op, plan_bytes_hash(plan, plan_len), cfg, ins, batch_size, device_id);
    guard.armed = false;                 // the context owns the streams from here on
    std::lock_guard<std::mutex> lk(g_mu);
    int64_t h = g_next++;
    g_ctx[h] = ctx;
    return h;
  });
}

// ---- org.apache.comet.parquet.Native: the record-batch reader the iceberg-compat scan drives (native/core/src/parquet/mod.rs:133-330).
// One file (byte ranges select its row groups), schemas as Arrow IPC schema messages, an optional pushed filter; every batch is decoded
// by the same NativeScan machinery as Operator.native_scan (parquet_scan.cpp), columns are handed out one at a time. ----
namespace {
struct ParquetReader {
  std::shared_ptr<ExecutionContext> ctx;
  std::vector<ArrowArray> arrays;
  std::vector<ArrowSchema> schemas;
  bool have_batch = false, done = false;
  std::string error;
  void drop_batch() {
    for (auto& a : arrays) if (a.release) a.release(&a);
    for (auto& sc : schemas) if (sc.release) sc.release(&sc);
    have_batch = false;
  }
  ~ParquetReader() { drop_batch(); }
};
std::mutex g_pq_mu;
std::map<int64_t, std::shared_ptr<ParquetReader>> g_pq;
int64_t g_pq_next = 1;
std::shared_ptr<ParquetReader> pq_lookup(int64_t h) {
  std::lock_guard<std::mutex> lk(g_pq_mu);
  auto it = g_pq.find(h);
  return it == g_pq.end() ? nullptr : it->second;
}
// the pushed filter is bound to data_schema positions; the scan's pruning binds to required_schema — re-bind by name, or drop the filter
// (it is a pruning hint only: parquet_exec.rs:166-176)
ExprP rebind_filter(const ExprP& e, const std::vector<StructField>& data, const std::vector<StructField>& required, bool& ok) {
  if (!e) return e;
  auto n = std::make_shared<Expr>(*e);
  if (e->kind == ExprKind::Bound) {
    n->bound_index = -1;
    if (e->bound_index >= 0 && (size_t)e->bound_index < data.size())
      for (size_t i = 0; i < required.size(); i++)
        if (required[i].name == data[(size_t)e->bound_index].name) n->bound_index = (int)i;
    if (n->bound_index < 0) ok = false;
    return n;
  }
  for (auto& c : n->children) c = rebind_filter(c, data, required, ok);
  return n;
}
}  // namespace

int64_t comet_parquet_reader_init(const char* file_path, int64_t file_size, const int64_t* starts, const int64_t* lengths, int32_t n_ranges,
                                  const uint8_t* filter, size_t filter_len, const uint8_t* required_schema_ipc, size_t required_len,
                                  const uint8_t* data_schema_ipc, size_t data_len, const char* session_timezone, int32_t batch_size,
                                  int32_t case_sensitive, int32_t device_id) {
  return guarded(nullptr, (int64_t)0, [&]() -> int64_t {
    if (!file_path || !required_schema_ipc) throw CometError("initRecordBatchReader: file path and required schema are mandatory");
    auto op = std::make_shared<Operator>();
    op->kind = OpKind::NativeScan;
    op->reader_api = true;      // (format errors stay "parquet: …" → ParquetRuntimeException, parquet/mod.rs; a plan's NativeScan classifies them for Spark)
    op->proto_tag = 111;
    op->required_schema = decode_ipc_schema(required_schema_ipc, required_len);
    op->data_schema = data_schema_ipc && data_len ? decode_ipc_schema(data_schema_ipc, data_len) : op->required_schema;
    for (auto& f : op->required_schema) op->scan_fields.push_back(f.dtype);
    for (size_t i = 0; i < op->required_schema.size(); i++) op->projection_vector.push_back((int64_t)i);
    op->session_timezone = session_timezone ? session_timezone : "UTC";
    op->case_sensitive = case_sensitive != 0;
    op->allow_type_promotion = true;            // the JVM side already validated the types (TypeUtil.checkParquetType), mod.rs:228-229
    op->allow_timestamp_ltz_to_ntz = true;
    const std::string path = file_path;
    if (n_ranges <= 0) {
      PartitionedFile pf;
      pf.file_path = path; pf.start = 0; pf.length = file_size; pf.file_size = file_size;
      op->files.push_back(pf);
    }
    for (int32_t i = 0; i < n_ranges; i++) {
      PartitionedFile pf;
      pf.file_path = path; pf.start = starts[i]; pf.length = lengths[i]; pf.file_size = file_size;
      op->files.push_back(pf);
    }
    if (filter && filter_len) {
      bool ok = true;
      ExprP f = rebind_filter(decode_expr_bytes(filter, filter_len), op->data_schema, op->required_schema, ok);
      if (ok) op->data_filters.push_back(f);
    }
    uint64_t h = plan_bytes_hash(required_schema_ipc, required_len) ^ 0x70617271ull;
    auto r = std::make_shared<ParquetReader>();
    r->ctx = std::make_shared<ExecutionContext>(op, h, std::vector<std::pair<std::string, std::string>>(), std::vector<InputSource>(), batch_size, device_id);
    r->arrays.resize(op->required_schema.size());
    r->schemas.resize(op->required_schema.size());
    for (auto& a : r->arrays) memset(&a, 0, sizeof a);
    for (auto& sc : r->schemas) memset(&sc, 0, sizeof sc);
    std::lock_guard<std::mutex> lk(g_pq_mu);
    int64_t handle = g_pq_next++;
    g_pq[handle] = r;
    return handle;
  });
}

int32_t comet_parquet_reader_next(int64_t handle) {
  auto r = pq_lookup(handle);
  if (!r) { t_last_error = "invalid parquet reader handle"; return -2; }
  return guarded(r->ctx.get(), (int32_t)-2, [&]() -> int32_t {
    r->drop_batch();
    if (r->done) return 0;
    std::vector<ArrowArray*> ap;
    std::vector<ArrowSchema*> sp;
    for (auto& a : r->arrays) ap.push_back(&a);
    for (auto& sc : r->schemas) sp.push_back(&sc);
    const int64_t rows = r->ctx->execute(ap.data(), sp.data(), (int)ap.size());
    if (rows < 0) { r->done = true; return 0; }          // end of file (mod.rs:268-276: rows_read stays 0)
    if (rows > INT32_MAX) throw CometError("readNextRecordBatch: batch larger than 2^31 rows");
    r->have_batch = true;
    return (int32_t)rows;
  });
}

int32_t comet_parquet_reader_column(int64_t handle, int32_t column, struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
  auto r = pq_lookup(handle);
  if (!r) { t_last_error = "invalid parquet reader handle"; return -2; }
  return guarded(r->ctx.get(), (int32_t)-2, [&]() -> int32_t {
    if (!r->have_batch) throw CometError("There is no more data to read");      // mod.rs:305-307
    if (column < 0 || (size_t)column >= r->arrays.size()) throw CometError("currentColumnBatch: column index out of range");
    if (!r->arrays[(size_t)column].release) throw CometError("currentColumnBatch: this column of the current batch was already taken");
    *out_array = r->arrays[(size_t)column];                  // move (Arrow C data interface: copy the struct, mark the source released)
    *out_schema = r->schemas[(size_t)column];
    r->arrays[(size_t)column].release = nullptr;
    r->schemas[(size_t)column].release = nullptr;
    return 0;
  });
}

void comet_parquet_reader_close(int64_t handle) {
  std::shared_ptr<ParquetReader> r;
  {
    std::lock_guard<std::mutex> lk(g_pq_mu);
    auto it = g_pq.find(handle);
    if (it == g_pq.end()) return;
    r = it->second;
    g_pq.erase(it);
  }
  r.reset();
}

int32_t comet_plan_set_subquery(int64_t handle, int64_t id, int32_t is_null, const uint8_t* value, size_t value_len) {
  auto ctx = lookup(handle);
  if (!ctx) {
    t_last_error = "invalid plan handle";
    return -2;
  }
  return guarded(ctx.get(), (int32_t)-2, [&]() -> int32_t {
    ctx->set_subquery_value(id, is_null != 0, std::string((const char*)value, value ? value_len : 0));
    return 0;
  });
}

int32_t comet_plan_set_subquery_provider(int64_t handle, comet_subquery_provider provider, void* provider_ctx) {
  auto ctx = lookup(handle);
  if (!ctx) {
    t_last_error = "invalid plan handle";
    return -2;
  }
  return guarded(ctx.get(), (int32_t)-2, [&]() -> int32_t {
    ctx->set_subquery_provider([provider, provider_ctx](int64_t id, const DType& type, bool& is_null, std::string& value) -> bool {
      if (!provider) return false;
      int32_t nul = 1;
      int64_t len = 0;
      value.assign(64, '\0');
      int32_t rc = provider(provider_ctx, id, (int32_t)type.id, &nul, (uint8_t*)&value[0], (int64_t)value.size(), &len);
      if (rc == 1 && len > (int64_t)value.size()) {      // a longer string / binary: once more with room
        value.assign((size_t)len, '\0');
        rc = provider(provider_ctx, id, (int32_t)type.id, &nul, (uint8_t*)&value[0], (int64_t)value.size(), &len);
      }
      if (rc < 0) throw CometError("the scalar subquery provider failed for subquery " + std::to_string(id));
      if (rc == 0) return false;
      is_null = nul != 0;
      value.resize((size_t)std::max<int64_t>(0, std::min<int64_t>(len, (int64_t)value.size())));
      return true;
    });
    return 0;
  });
}

int64_t comet_execute_plan(int64_t handle, struct ArrowArray** out_arrays, struct ArrowSchema** out_schemas, int32_t n_out) {
  auto ctx = lookup(handle);
  if (!ctx) {
    t_last_error = "invalid plan handle";
    return -2;
  }
  ApiTrace api_trace("executePlan", handle);
  struct Executing { Executing() { comet::detail::plan_execution_begins(); } ~Executing() { comet::detail::plan_execution_ends(); } } executing;
  return guarded(ctx.get(), (int64_t)-2, [&]() -> int64_t { return ctx->execute(out_arrays, out_schemas, n_out); });
}

int64_t comet_execute_plan_device(int64_t handle, struct ArrowDeviceArray** out_arrays, struct ArrowSchema** out_schemas, int32_t n_out) {
  auto ctx = lookup(handle);
  if (!ctx) {
    t_last_error = "invalid plan handle";
    return -2;
  }
  struct Executing { Executing() { comet::detail::plan_execution_begins(); } ~Executing() { comet::detail::plan_execution_ends(); } } executing;
  return guarded(ctx.get(), (int64_t)-2, [&]() -> int64_t { return ctx->execute_device(out_arrays, out_schemas, n_out); });
}

void comet_release_plan(int64_t handle) {
  std::shared_ptr<ExecutionContext> ctx;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(handle);
    if (it == g_ctx.end()) return;
    ctx = it->second;
    g_ctx.erase(it);
  }
  ApiTrace api_trace("releasePlan", handle);
  guarded(nullptr, 0, [&]() -> int {
    std::shared_ptr<MemAccount> mem = ctx->memory_account();
    ctx.reset();          // (another thread still inside a call keeps the context alive until it returns)
    mem->detach();        // whatever outlives the plan is no longer charged to the task; the manager is not called again
    return 0;
  });
}

int32_t comet_plan_set_memory_manager(int64_t handle, int64_t (*acquire)(void*, int64_t), void (*release)(void*, int64_t), void* ctx_, int64_t task_id) {
  auto ctx = lookup(handle);
  if (!ctx) return -2;
  ctx->set_memory_manager(acquire, release, ctx_, (long long)task_id);
  return 0;
}

void comet_plan_memory_stats(int64_t handle, int64_t* out4) {
  auto ctx = lookup(handle);
  if (!ctx || !out4) return;
  ctx->memory_stats(out4);
}

const char* comet_last_error(int64_t handle) {
  if (handle == 0) return t_last_error.c_str();
  auto ctx = lookup(handle);
  if (!ctx) return t_last_error.c_str();
  // keep the string alive in thread-local storage: the context may be released concurrently
  t_last_error = ctx->last_error;
  return t_last_error.c_str();
}

int32_t comet_last_error_kind(int64_t handle) {
  if (handle == 0) return t_last_kind;
  auto ctx = lookup(handle);
  return ctx ? ctx->last_error_kind : t_last_kind;
}

int64_t comet_plan_metrics(int64_t handle, uint8_t* buf, size_t cap) {
  auto ctx = lookup(handle);
  if (!ctx) return -2;
  return guarded(ctx.get(), (int64_t)-2, [&]() -> int64_t {
    std::string s = ctx->metrics_proto();
    if (buf && cap) memcpy(buf, s.data(), std::min(cap, s.size()));
    return (int64_t)s.size();
  });
}

const char* comet_explain(int64_t handle) {
  auto ctx = lookup(handle);
  if (!ctx) return "";
  static thread_local std::string s;
  s = ctx->explain();
  return s.c_str();
}

void comet_plan_kernel_stats(int64_t handle, double* kernel_ms, int64_t* launches, int64_t* input_rows) {
  auto ctx = lookup(handle);
  if (!ctx) return;
  if (kernel_ms) *kernel_ms = ctx->last_kernel_ms;
  if (launches) *launches = ctx->last_kernel_launches;
  if (input_rows) *input_rows = ctx->input_rows;
}

void comet_set_kernel_times(int32_t on) { g_kernel_times.store(on ? 1 : 0); }

int64_t comet_plan_kernel_times(int64_t handle, char* buf, size_t cap) {
  auto ctx = lookup(handle);
  if (!ctx) return -1;
  return guarded(ctx.get(), (int64_t)-2, [&]() -> int64_t {
    std::string s = ctx->kernel_times_json();
    if (buf && cap) {
      size_t n = std::min(cap - 1, s.size());
      memcpy(buf, s.data(), n);
      buf[n] = 0;
    }
    return (int64_t)s.size();
  });
}

void comet_plan_aux_kernel_stats(int64_t handle, double* aux_ms, int64_t* aux_launches) {
  auto ctx = lookup(handle);
  if (!ctx) return;
  if (aux_ms) *aux_ms = ctx->last_aux_ms;
  if (aux_launches) *aux_launches = ctx->last_aux_launches;
}

int32_t comet_compile_plan(const uint8_t* plan, size_t plan_len, char* out, size_t cap) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    OperatorP op = decode_operator(plan, plan_len);
    std::string ex = ExecutionContext::compile_only(op, plan_bytes_hash(plan, plan_len));
    if (out && cap) {
      size_t n = std::min(cap - 1, ex.size());
      memcpy(out, ex.data(), n);
      out[n] = 0;
    }
    return 0;
  });
}

static std::string json_str(const std::string& s) {
  std::string o = "\"";
  for (unsigned char ch : s) {
    if (ch == '"' || ch == '\\') { o += '\\'; o += (char)ch; }
    else if (ch == '\n') o += "\\n";
    else if (ch == '\t') o += "\\t";
    else if (ch == '\r') o += "\\r";
    else if (ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", ch); o += b; }
    else o += (char)ch;
  }
  return o + "\"";
}
int64_t comet_plan_codegen(const uint8_t* plan, size_t plan_len, const uint8_t* has_valid, int32_t n_valid, char* out, int64_t cap) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    OperatorP op = decode_operator(plan, plan_len);
    const Operator* leaf = op.get();
    while (!leaf->children.empty()) leaf = leaf->children[0].get();
    if (leaf->kind != OpKind::Scan) throw CometError("comet_plan_codegen: a Filter / Projection / HashAggregate chain over ONE Scan leaf is expected");
    // (a source with struct / list columns: the chain sees their fields and elements as columns behind the real ones, like over a materialised source)
    bool nested = false;
    for (auto& t : leaf->scan_fields) nested = nested || t.is_nested();
    const std::vector<DType> types = nested ? extend_struct_field_types(leaf->scan_fields) : leaf->scan_fields;
    std::vector<bool> hv(types.size(), false);
    for (int32_t k = 0; k < n_valid && (size_t)k < hv.size(); k++) hv[(size_t)k] = has_valid && has_valid[k] != 0;
    PipelineDesc d = nested ? generate_pipeline(*op, hv, &types) : generate_pipeline(*op, hv);
    std::string j = "{\"sink\":" + std::to_string((int)d.sink) + ",\"has_filter\":" + (d.has_filter ? "true" : "false") + ",\"R\":" + std::to_string(d.R) + ",\"derived\":" + std::to_string(d.derived.size()) +
                    ",\"kernels\":[";
    for (size_t k = 0; k < d.kernels.size(); k++) j += (k ? "," : "") + json_str(d.kernels[k]);
    j += "],\"out\":[";
    for (size_t k = 0; k < d.out_cols.size(); k++) {
      const OutCol& oc = d.out_cols[k];
      j += std::string(k ? "," : "") + "{\"type\":" + std::to_string((int)oc.type.id) + ",\"precision\":" + std::to_string(oc.type.precision) + ",\"scale\":" + std::to_string(oc.type.scale) +
           ",\"nullable\":" + (oc.nullable ? "true" : "false") + ",\"gather_src\":" + std::to_string(oc.gather_src) + ",\"view_src\":" + std::to_string(oc.view_src) + ",\"fmt_kind\":" +
           std::to_string(oc.fmt_kind) + ",\"packed_string\":" + (oc.packed_string ? "true" : "false") + ",\"concat\":" + std::to_string(oc.concat_cols.size()) + ",\"case_mode\":" +
           std::to_string(oc.case_mode) + ",\"pad\":" + json_str(oc.pad_pattern) + ",\"pad_left\":" + (oc.pad_left ? "true" : "false") + "}";
    }
    j += "],\"fix_sums\":[";      // exact Float64 sums: the accumulator word and the aux words (exponent range seen) of each, for the scale pass (exec_pipeline.cpp adjust_fix_scales)
    for (size_t k = 0; k < d.fix_sums.size(); k++)
      j += std::string(k ? "," : "") + "{\"word\":" + std::to_string(d.fix_sums[k].word) + ",\"aux_hi\":" + std::to_string(d.fix_sums[k].aux_hi) + ",\"aux_lo\":" + std::to_string(d.fix_sums[k].aux_lo) + "}";
    j += "],\"source\":" + json_str(d.source) + "}";
    if (out && cap > (int64_t)j.size()) memcpy(out, j.c_str(), j.size() + 1);
    return (int64_t)j.size();
  });
}
int64_t comet_error_site_json(uint32_t site_id, uint64_t lo, uint64_t hi, const uint8_t* str, int64_t str_avail, char* out, int64_t cap) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    ErrSite site;
    if (!lookup_err_site(site_id, site)) throw CometError("no raise site with id " + std::to_string(site_id) + " is registered in this process");
    const std::string j = err_site_json(site, lo, hi, str, str_avail < 0 ? 0 : (size_t)str_avail, nullptr);
    if (out && cap > (int64_t)j.size()) memcpy(out, j.c_str(), j.size() + 1);
    return (int64_t)j.size();
  });
}
int64_t comet_embedded_header(const char* name, char* out, int64_t cap) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    const std::string n = name ? name : "";
    const char* t = n == "comet_device.hpp" ? kEmbeddedDeviceHeader : n == "kparams.h" ? kEmbeddedKParamsHeader : n == "comet_ryu.hpp" ? kEmbeddedRyuHeader : n == "comet_strtod.hpp" ? kEmbeddedStrtodHeader
                    : n == "comet_strts.hpp" ? kEmbeddedStrtsHeader : n == "comet_regex_vm.hpp" ? kEmbeddedRegexVmHeader : nullptr;
    if (!t) throw CometError("no embedded header named '" + n + "'");
    const int64_t len = (int64_t)strlen(t);
    if (out && cap > len) memcpy(out, t, (size_t)len + 1);
    return len;
  });
}

int32_t comet_check_plan(const uint8_t* plan, size_t plan_len, char* out, size_t cap) {
  int32_t rc = guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    OperatorP op = decode_operator(plan, plan_len);
    std::string ex = ExecutionContext::check_only(op, plan_bytes_hash(plan, plan_len));
    if (out && cap) {
      size_t n = std::min(cap - 1, ex.size());
      memcpy(out, ex.data(), n);
      out[n] = 0;
    }
    return 0;
  });
  if (rc != 0 && out && cap) {          // the reason, where the description would have gone
    const char* why = comet_last_error(0);
    size_t n = std::min(cap - 1, strlen(why));
    memcpy(out, why, n);
    out[n] = 0;
  }
  return rc;
}

int64_t comet_parquet_prune_report(const uint8_t* plan, size_t plan_len, int32_t page_index, char* out, size_t cap) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    OperatorP op = decode_operator(plan, plan_len);
    const Operator* scan = op.get();
    while (scan && scan->kind != OpKind::NativeScan) scan = scan->children.empty() ? nullptr : scan->children[0].get();
    if (!scan) throw CometError("comet_parquet_prune_report: the plan holds no NativeScan");
    const std::string j = parquet_prune_report(*scan, (page_index & 1) != 0, (page_index & 2) == 0);
    if (out && cap) {
      const size_t n = std::min(cap - 1, j.size());
      memcpy(out, j.data(), n);
      out[n] = 0;
    }
    return (int64_t)j.size();
  });
}

uint64_t comet_xxh64(const uint8_t* data, size_t len, uint64_t seed) { return pq::xxh64(data, len, seed); }
int32_t comet_sbbf_might_contain(const uint8_t* bitset, size_t nbytes, uint64_t hash) { return pq::sbbf_might_contain(bitset, nbytes, hash) ? 1 : 0; }

int64_t comet_parquet_host_plain_values(const uint8_t* plan, size_t plan_len, int32_t column, uint8_t* out, size_t cap) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    OperatorP op = decode_operator(plan, plan_len);
    const Operator* scan = op.get();
    while (scan && scan->kind != OpKind::NativeScan) scan = scan->children.empty() ? nullptr : scan->children[0].get();
    if (!scan) throw CometError("comet_parquet_host_plain_values: the plan holds no NativeScan");
    if (column < 0) throw CometError("comet_parquet_host_plain_values: negative column");
    const std::vector<uint8_t> v = parquet_host_plain_values(*scan, (size_t)column);
    if (out && cap) memcpy(out, v.data(), std::min(cap, v.size()));
    return (int64_t)v.size();
  });
}

int64_t comet_error_json(const char* error_type, const char* error_class, const char* from_type, const char* to_type, int32_t precision, int32_t scale,
                         int32_t value_kind, const char* suffix, uint64_t lo, uint64_t hi, const uint8_t* str, int64_t str_avail, char* out, int64_t cap) {
  comet::ErrSite s;
  s.error_type = error_type ? error_type : "";
  s.error_class = error_class ? error_class : "";
  s.from_type = from_type ? from_type : "";
  s.to_type = to_type ? to_type : "";
  s.precision = precision;
  s.scale = scale;
  s.value = value_kind;
  s.suffix = suffix ? suffix : "";
  const std::string j = comet::err_site_json(s, lo, hi, str, str_avail < 0 ? 0 : (size_t)str_avail);
  if (out && cap > (int64_t)j.size()) memcpy(out, j.c_str(), j.size() + 1);
  return (int64_t)j.size();
}

int64_t comet_plan_error_json(const uint8_t* plan, size_t plan_len, int32_t site_index, uint64_t lo, uint64_t hi, const uint8_t* str, int64_t str_avail,
                              char* out, int64_t cap) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    OperatorP op = decode_operator(plan, plan_len);
    const Operator* leaf = op.get();
    while (!leaf->children.empty()) leaf = leaf->children[0].get();
    std::vector<bool> none(leaf->scan_fields.size(), false);
    const PipelineDesc d = generate_pipeline(*op, none);
    if (site_index == -1 || site_index == -2) {      // the pipeline's ANSI decimal sum (-1) / average (-2) overflow
      const std::string j = decimal_sum_overflow_json(-1 - site_index, d.agg_ctx[-1 - site_index].get());
      if (out && cap > (int64_t)j.size()) memcpy(out, j.c_str(), j.size() + 1);
      return (int64_t)j.size();
    }
    if (site_index < 0 || (size_t)site_index >= d.site_contexts.size())
      throw CometError("the plan's pipeline has " + std::to_string(d.site_contexts.size()) + " raise sites with a QueryContext");
    ErrSite site;
    if (!lookup_err_site(d.site_contexts[(size_t)site_index].first, site)) throw CometError("internal: a raise site is not registered");
    const std::string j = err_site_json(site, lo, hi, str, str_avail < 0 ? 0 : (size_t)str_avail, d.site_contexts[(size_t)site_index].second.get());
    if (out && cap > (int64_t)j.size()) memcpy(out, j.c_str(), j.size() + 1);
    return (int64_t)j.size();
  });
}

int64_t comet_plan_site_error_json(const uint8_t* plan, size_t plan_len, uint32_t site_id, uint64_t lo, uint64_t hi, const uint8_t* str, int64_t str_avail,
                                   char* out, int64_t cap) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    OperatorP op = decode_operator(plan, plan_len);
    const Operator* leaf = op.get();
    while (!leaf->children.empty()) leaf = leaf->children[0].get();
    bool nested = false;      // (as in comet_plan_codegen: fields and elements are columns behind the real ones)
    for (auto& t : leaf->scan_fields) nested = nested || t.is_nested();
    const std::vector<DType> types = nested ? extend_struct_field_types(leaf->scan_fields) : leaf->scan_fields;
    std::vector<bool> none(types.size(), false);
    const PipelineDesc d = nested ? generate_pipeline(*op, none, &types) : generate_pipeline(*op, none);
    ErrSite site;
    if (!lookup_err_site(site_id, site)) throw CometError("no raise site with id " + std::to_string(site_id) + " is registered in this process");
    const QueryContext* ctx = nullptr;      // what ExecutionContext::site_context answers for the pipeline's kernels
    for (auto& kv : d.site_contexts)
      if (kv.first == site_id) { ctx = kv.second.get(); break; }
    const std::string j = err_site_json(site, lo, hi, str, str_avail < 0 ? 0 : (size_t)str_avail, ctx);
    if (out && cap > (int64_t)j.size()) memcpy(out, j.c_str(), j.size() + 1);
    return (int64_t)j.size();
  });
}

int64_t comet_zone_table(const char* zone, int64_t* out, int64_t cap) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    const std::vector<int64_t> f = load_zone(zone ? zone : "")->flat();
    if (out && cap >= (int64_t)f.size()) memcpy(out, f.data(), f.size() * 8);
    return (int64_t)f.size();
  });
}

int32_t comet_rlike_match(const char* pattern, const uint8_t* value, size_t value_len) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    // (the last few patterns stay compiled: a test walks thousands of values through one pattern)
    static std::mutex mu;
    static std::map<std::string, std::shared_ptr<const RegexDfa>> cache;
    const std::string key = pattern ? pattern : "";
    std::shared_ptr<const RegexDfa> d;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = cache.find(key);
      if (it != cache.end()) d = it->second;
    }
    if (!d) {
      d = std::make_shared<const RegexDfa>(compile_rlike(key));
      std::lock_guard<std::mutex> lk(mu);
      if (cache.size() >= 64) cache.clear();
      cache[key] = d;
    }
    return regex_dfa_match(*d, value, value_len) ? 1 : 0;
  });
}

int32_t comet_regexp_extract_host(const char* pattern, int32_t group, const uint8_t* value, size_t value_len, int32_t* start, int32_t* len) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    static std::mutex mu;
    static std::map<std::pair<std::string, int>, std::shared_ptr<const RegexProg>> cache;
    const std::pair<std::string, int> key(pattern ? pattern : "", (int)group);
    std::shared_ptr<const RegexProg> p;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = cache.find(key);
      if (it != cache.end()) p = it->second;
    }
    if (!p) {
      p = std::make_shared<const RegexProg>(compile_regex_captures(key.first, group, "regexp_extract"));
      std::lock_guard<std::mutex> lk(mu);
      if (cache.size() >= 64) cache.clear();
      cache[key] = p;
    }
    return regex_prog_extract(*p, value, value_len, start, len) ? 1 : 0;
  });
}

int64_t comet_strfn_host(int32_t op, const uint8_t* value, int32_t n, const uint8_t* a, int32_t na, const uint8_t* b, int32_t nb, int64_t k, uint8_t* out, int64_t cap) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    using namespace comet_strfn_host_ns;
    if (op == 20) return (int64_t)sf_crc32(value, (sf_i64)n);
    if (op == 21) return (int64_t)sf_instr(value, n, a, na);
    if (op == 22) return (int64_t)sf_ascii(value, n);
    const int64_t len = sf_len(op, value, n, a, na, b, nb, k);
    if (len <= cap && out) sf_write(op, value, n, a, na, b, nb, k, out);
    return len;
  });
}

int32_t comet_extract_all_host(const char* pattern, int32_t group, const uint8_t* value, size_t value_len, int32_t* starts, int32_t* lens, int32_t cap) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    const RegexProg whole = compile_regex_captures(pattern ? pattern : "", 0, "regexp_extract_all");
    const RegexProg grp = group == 0 ? RegexProg() : compile_regex_captures(pattern ? pattern : "", group, "regexp_extract_all");
    const auto spans = regex_prog_find_all(whole, group == 0 ? whole : grp, value, value_len);
    for (size_t k = 0; k < spans.size() && (int32_t)k < cap; k++) { starts[k] = spans[k].first; lens[k] = spans[k].second; }
    return (int32_t)spans.size();
  });
}

int32_t comet_split_host(const char* pattern, int32_t limit, const uint8_t* value, size_t value_len, int32_t* starts, int32_t* lens, int32_t cap) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    const RegexProg prog = compile_regex_captures(pattern ? pattern : "", 0, "split");
    const auto pieces = regex_prog_split(prog, value, value_len, limit);
    for (size_t k = 0; k < pieces.size() && (int32_t)k < cap; k++) { starts[k] = pieces[k].first; lens[k] = pieces[k].second; }
    return (int32_t)pieces.size();
  });
}

namespace comet_dates_host {
#include "device/dates.hpp"
}
int32_t comet_date_fn_host(int32_t fn, int64_t a, int64_t b, int64_t c, int64_t* out) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    using namespace comet_dates_host;
    switch (fn) {
      case 0: *out = date_part((i32)a, (int)b); return 1;
      case 1: *out = date_weekday_mon0((i32)a) + 1; return 1;
      case 2: *out = date_iso_week((i32)a); return 1;
      case 3: if (!date_in_chrono_range((i32)a)) return 0; *out = date_trunc_days((i32)a, (int)b); return 1;
      case 4: if (!date_in_chrono_range((i32)a)) return 0; *out = date_last_day((i32)a); return 1;
      case 5: if (!date_in_chrono_range((i32)a)) return 0; *out = date_next_day((i32)a, (int)b); return 1;
      case 6: { i32 o = 0; if (!date_make((i32)a, (i32)b, (i32)c, o)) return 0; *out = o; return 1; }
      case 7: *out = ts_trunc_local_us(a, (int)b); return 1;
      default: throw CometError("comet_date_fn_host: unknown function");
    }
  });
}

int64_t comet_snappy_view_read(const uint8_t* src, size_t src_len, int32_t max_elems, const int64_t* offsets, int32_t n, uint8_t* out) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    pq::SnappyView v;
    if (!v.build(src, src_len, (size_t)max_elems)) return -1;
    for (int32_t i = 0; i < n; i++) out[i] = v.at((size_t)offsets[i]);
    return (int64_t)v.out_len;
  });
}

int32_t comet_page_decompress(int32_t codec, const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    pq::decompress(codec, src, src_len, dst, dst_len);
    return 0;
  });
}

int32_t comet_murmur3_column(int32_t type_id, int32_t precision, const void* values, const uint8_t* validity,
                             const void* aux_bytes, int64_t n, uint32_t* hashes, void* hip_stream) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    int rc = comet_launch_murmur3(type_id, precision, values, validity, aux_bytes, n, hashes, hip_stream);
    if (rc != 0) throw CometError("murmur3: unsupported column type " + std::to_string(type_id));
    return 0;
  });
}

int32_t comet_pmod_partition(const uint32_t* hashes, int64_t n, int32_t num_partitions, int32_t* partition_ids, void* hip_stream) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    if (num_partitions <= 0) throw CometError("pmod: num_partitions must be positive");
    if (comet_launch_pmod(hashes, n, num_partitions, partition_ids, hip_stream) != 0) throw CometError("pmod launch failed");
    return 0;
  });
}

int32_t comet_partition_indices(const int32_t* partition_ids, int64_t n, int32_t num_partitions, int64_t* partition_starts,
                                uint32_t* partition_row_indices, void* hip_stream) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    if (num_partitions <= 0 || num_partitions > 4096) throw CometError("partition_indices: num_partitions must be in 1..4096");
    if (n < 0 || n >= (int64_t)1 << 32) throw CometError("partition_indices: row count must fit 32 bits (multi_partition.rs uses u32 row indices)");
    hipStream_t st = (hipStream_t)hip_stream;
    const int64_t W = comet_partition_tiles(n);
    DevBuf scratch;   // histogram + one flag word; returned to the pool once the stream is idle
    const size_t hist_bytes = ((size_t)num_partitions * (size_t)W + 1) * 8;
    scratch.ensure((size_t)comet_partition_scratch_bytes(n, num_partitions));
    uint32_t* bad = (uint32_t*)((char*)scratch.p + hist_bytes);
    if (hipMemsetAsync(bad, 0, 4, st) != hipSuccess) throw CometError("partition_indices: memset failed");
    if (comet_launch_partition_indices(partition_ids, n, num_partitions, (uint64_t*)scratch.p, bad, partition_starts, partition_row_indices, st) != 0)
      throw CometError("partition_indices: launch failed");
    uint32_t flag = 0;
    if (hipMemcpyAsync(&flag, bad, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      throw CometError(std::string("partition_indices: ") + hipGetErrorString(hipGetLastError()));
    if (flag) throw CometError("partition_indices: a partition id is outside [0, num_partitions)");
    return 0;
  });
}

int32_t comet_take_column(int32_t width_bytes, const void* src, const uint32_t* row_indices, int64_t n, void* dst, void* hip_stream) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    if (comet_launch_take(width_bytes, src, row_indices, n, dst, hip_stream) != 0)
      throw CometError("take_column: unsupported value width " + std::to_string(width_bytes));
    return 0;
  });
}

int64_t comet_take_utf8_offsets(const int32_t* offsets, const uint8_t* validity_bits, const uint32_t* row_indices, int64_t n,
                                int32_t* out_offsets, void* hip_stream) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    hipStream_t st = (hipStream_t)hip_stream;
    if (n <= 0) {
      if (hipMemsetAsync(out_offsets, 0, 4, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) throw CometError("take_utf8_offsets: memset failed");
      return 0;
    }
    DevBuf lengths, tiles;
    PinnedBuf host;
    lengths.ensure((size_t)n * 4 + 16);
    tiles.ensure((size_t)((n + 1023) / 1024 + 2) * 8);
    host.ensure(64);
    if (comet_launch_take_utf8_lengths(offsets, row_indices, nullptr, validity_bits, n, (uint32_t*)lengths.p, st) != 0) throw CometError("take_utf8_offsets: launch failed");
    pq_launch_u32_scan((const uint32_t*)lengths.p, n, (uint64_t*)tiles.p, out_offsets, st);
    if (hipMemcpyAsync(host.p, out_offsets + n, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      throw CometError(std::string("take_utf8_offsets: ") + hipGetErrorString(hipGetLastError()));
    const int32_t total = *(const int32_t*)host.p;
    if (total < 0) throw CometError("Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
    return (int64_t)total;
  });
}

int32_t comet_take_utf8_bytes(const int32_t* offsets, const uint8_t* bytes, const uint8_t* validity_bits, const uint32_t* row_indices, int64_t n,
                              const int32_t* out_offsets, uint8_t* out_bytes, void* hip_stream) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    if (comet_launch_take_utf8_copy(offsets, bytes, row_indices, nullptr, validity_bits, n, out_offsets, out_bytes, hip_stream) != 0)
      throw CometError("take_utf8_bytes: launch failed");
    return 0;
  });
}

const char* comet_version(void) { return "comet-mi355x 0.1.0 (gfx950)"; }

const char* comet_jit_toolchain(void) {
  static const std::string s = comet::jit_toolchain();
  return s.c_str();
}

}  // extern "C"

// ---- Parquet footer description (host-only; used by tests to pin the Thrift/footer parser against pyarrow) ----
#include <fstream>
#include <iterator>

#include "parquet_meta.hpp"
extern "C" int64_t comet_decode_shuffle_block(const uint8_t* block, int64_t len, struct ArrowArray** out_arrays, struct ArrowSchema** out_schemas,
                                   int32_t n_out) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    if (!block || len < 0) throw CometError("decodeShuffleBlock: null block");
    HostBatch b = decode_shuffle_block(block, (size_t)len);
    const int64_t rows = b.rows;
    export_host_batch(b, out_arrays, out_schemas, n_out);
    return rows;
  });
}

namespace {
DType dtype_from_format(const char* f) {
  std::string s = f ? f : "";
  if (s == "b") return DType::of(TypeId::Bool);
  if (s == "c") return DType::of(TypeId::Int8);
  if (s == "s") return DType::of(TypeId::Int16);
  if (s == "i") return DType::of(TypeId::Int32);
  if (s == "l") return DType::of(TypeId::Int64);
  if (s == "f") return DType::of(TypeId::Float);
  if (s == "g") return DType::of(TypeId::Double);
  if (s == "u") return DType::of(TypeId::String);
  if (s == "z") return DType::of(TypeId::Bytes);
  if (s == "tdD") return DType::of(TypeId::Date);
  if (s == "tsu:") return DType::of(TypeId::TimestampNtz);
  if (s.rfind("tsu:", 0) == 0) return DType::of(TypeId::Timestamp);
  if (s.rfind("d:", 0) == 0) {
    int p = 0, sc = 0, bits = 128;
    if (sscanf(s.c_str(), "d:%d,%d,%d", &p, &sc, &bits) >= 2 && bits == 128) return DType::decimal(p, sc);
  }
  throw CometError("encodeShuffleBlock: Arrow format '" + s + "' is not supported");
}
// an Arrow C Data array (children included) as a ColumnSlice; child arrays must start at their own row 0 (what every exporter of a fresh
// array produces)
DType dtype_of_schema(const ArrowSchema* s) {
  const std::string f = s->format ? s->format : "";
  if (f != "+s" && f != "+l" && f != "+m") return dtype_from_format(s->format);
  DType t = DType::of(f == "+s" ? TypeId::Struct : f == "+l" ? TypeId::List : TypeId::Map);
  if (f != "+s" && s->n_children != 1) throw CometError("encodeShuffleBlock: a list / map schema with " + std::to_string(s->n_children) + " children");
  for (int64_t k = 0; k < s->n_children; k++) {
    t.kids.push_back(dtype_of_schema(s->children[k]));
    t.kid_names.push_back(f == "+l" ? std::string("element") : f == "+m" ? std::string("entries") : std::string(s->children[k]->name ? s->children[k]->name : ""));
    t.kid_nullable.push_back((s->children[k]->flags & ARROW_FLAG_NULLABLE) ? 1 : 0);
  }
  if (f == "+m") {      // (whatever the producer called them: the entries' two fields are the key and the value)
    if (t.kids[0].id != TypeId::Struct || t.kids[0].kids.size() != 2) throw CometError("encodeShuffleBlock: a map whose entries are not (key, value) structs");
    t.kids[0].kid_names = {"key", "value"};
  }
  return t;
}
void slice_of_array(const ArrowArray* a, const ArrowSchema* s, bool top, ColumnSlice& c) {
  if (a->dictionary) throw CometError("encodeShuffleBlock: dictionary-encoded input is not supported");
  if (!top && a->offset != 0) throw CometError("encodeShuffleBlock: a child array with a non-zero offset is not supported");
  c.type = dtype_of_schema(s);
  c.validity = a->null_count != 0 ? (const uint8_t*)a->buffers[0] : nullptr;
  c.first = a->offset;
  if (c.type.id == TypeId::Struct) {
    if (a->n_children != (int64_t)c.type.kids.size()) throw CometError("encodeShuffleBlock: struct array and schema differ in their children");
    c.kids.resize((size_t)a->n_children);
    for (int64_t k = 0; k < a->n_children; k++) slice_of_array(a->children[k], s->children[k], false, c.kids[(size_t)k]);
    return;
  }
  c.values = a->n_buffers > 1 ? a->buffers[1] : nullptr;
  if (c.type.is_listlike()) {
    if (a->n_children != 1) throw CometError("encodeShuffleBlock: list array without its elements");
    c.kids.resize(1);
    slice_of_array(a->children[0], s->children[0], false, c.kids[0]);
    return;
  }
  c.data = a->n_buffers > 2 ? (const uint8_t*)a->buffers[2] : nullptr;
}
}  // namespace

int32_t comet_encode_shuffle_block(struct ArrowArray** arrays, struct ArrowSchema** schemas, int32_t n_cols, int32_t codec,
                                   int32_t compression_level, uint8_t** out, int64_t* out_len) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    if (!out || !out_len) throw CometError("encodeShuffleBlock: null output");
    if (codec < 0 || codec > 3) throw CometError("Unsupported shuffle compression codec: " + std::to_string(codec));
    std::vector<ColumnSlice> cols((size_t)n_cols);
    int64_t rows = n_cols ? arrays[0]->length : 0;
    for (int i = 0; i < n_cols; i++) {
      const ArrowArray* a = arrays[i];
      if (a->length != rows) throw CometError("encodeShuffleBlock: columns differ in length");
      slice_of_array(a, schemas[i], true, cols[(size_t)i]);
    }
    std::vector<uint8_t> bytes;
    encode_shuffle_block(cols, rows, (ShuffleCodec)codec, compression_level, bytes);
    *out_len = (int64_t)bytes.size();
    *out = nullptr;
    if (!bytes.empty()) {
      *out = (uint8_t*)malloc(bytes.size());
      if (!*out) throw CometError("encodeShuffleBlock: out of memory");
      memcpy(*out, bytes.data(), bytes.size());
    }
    return 0;
  });
}

int64_t comet_concat_nested_column(struct ArrowArray** arrays, struct ArrowSchema* schema, int32_t n, struct ArrowArray* out, struct ArrowSchema* out_schema) {
  return guarded(nullptr, (int64_t)-2, [&]() -> int64_t {
    if (!arrays || !schema || !out || !out_schema || n < 0) throw CometError("concatNestedColumn: bad arguments");
    const DType t = dtype_of_schema(schema);
    if (!t.is_nested()) throw CometError("concatNestedColumn: " + t.str() + " is not a nested type");
    if (!comet::detail::nested_schema_matches(schema, t)) throw CometError("concatNestedColumn: the schema does not describe " + t.str());
    HostColumn h;
    h.type = t;
    std::function<void(HostColumn&, const DType&)> shape = [&](HostColumn& x, const DType& tt) {
      x.type = tt;
      if (tt.id == TypeId::Struct) { x.children.resize(tt.kids.size()); for (size_t k = 0; k < tt.kids.size(); k++) shape(x.children[k], tt.kids[k]); }
      else if (tt.is_listlike()) { x.children.resize(1); shape(x.children[0], tt.kids.at(0)); }
    };
    shape(h, t);
    for (int32_t i = 0; i < n; i++) comet::detail::append_nested_rows(h, arrays[i], t, 0, arrays[i]->length);
    // what the upload does: NULLs are counted from the bitmaps (fields were masked after their own counts), all-valid bitmaps are dropped
    std::function<void(HostColumn&)> finish = [&](HostColumn& x) {
      int64_t nulls = 0;
      for (int64_t r = 0; r < x.length && !x.validity.empty(); r++) nulls += !((x.validity[(size_t)(r >> 3)] >> (r & 7)) & 1);
      x.null_count = nulls;
      if (!nulls) x.validity.clear();
      if ((x.type.is_listlike() || x.type.id == TypeId::String || x.type.id == TypeId::Bytes) && x.values.empty()) x.values.assign(4, 0);
      for (auto& k : x.children) finish(k);
    };
    finish(h);
    const int64_t rows = h.length;
    HostBatch b;
    b.rows = rows;
    b.cols.push_back(std::move(h));
    ArrowArray* oa[1] = {out};
    ArrowSchema* os[1] = {out_schema};
    export_host_batch(b, oa, os, 1);
    return rows;
  });
}

void comet_free_buffer(uint8_t* p) { free(p); }

void comet_sort_row_partitions(int64_t* records, int64_t n) {
  if (records && n > 1) comet::sort_row_partitions(records, (size_t)n);
}

int32_t comet_write_sorted_rows(const int64_t* row_addresses, const int32_t* row_sizes, int64_t row_num, const uint8_t* const* serialized_datatypes,
                                const int32_t* datatype_lens, int32_t n_cols, const char* file_path, int32_t batch_size, int32_t checksum_enabled,
                                int32_t checksum_algo, int64_t current_checksum, const char* compression_codec, int32_t compression_level,
                                int64_t out_result[3]) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    if (!file_path || !out_result || row_num < 0 || n_cols < 0) throw CometError("writeSortedFileNative: bad arguments");
    std::vector<DType> schema((size_t)n_cols);
    for (int i = 0; i < n_cols; i++) schema[(size_t)i] = comet::decode_datatype_bytes(serialized_datatypes[i], (size_t)datatype_lens[i]);
    const std::string codec = compression_codec ? compression_codec : "";
    // jni_api.rs:1086-1091: unknown names fall back to the LZ4 frame codec
    const ShuffleCodec c = codec == "zstd" ? ShuffleCodec::Zstd : codec == "snappy" ? ShuffleCodec::Snappy : ShuffleCodec::Lz4;
    const bool has_initial = current_checksum != INT64_MIN;
    comet::SortedFileResult r = comet::write_sorted_rows(row_addresses, row_sizes, (size_t)row_num, schema, file_path, (size_t)batch_size,
                                                         checksum_enabled != 0, checksum_algo, has_initial, (uint32_t)current_checksum, c,
                                                         compression_level);
    out_result[0] = r.written;
    out_result[1] = r.has_checksum ? (int64_t)r.checksum : INT64_MIN;
    out_result[2] = r.encode_nanos;
    return 0;
  });
}

int32_t comet_parquet_describe(const char* path, char* out, size_t cap) {
  return guarded(nullptr, (int32_t)-2, [&]() -> int32_t {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw CometError(std::string("cannot open ") + path);
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    comet::pq::FileMeta fm = comet::pq::parse_footer(data.data(), data.size());
    std::string s = "rows=" + std::to_string(fm.num_rows) + ";row_groups=" + std::to_string(fm.row_groups.size()) + ";schema=";
    for (size_t i = 1; i < fm.schema.size(); i++)
      s += fm.schema[i].name + ":" + std::to_string(fm.schema[i].type) + ":" + std::to_string(fm.schema[i].repetition) + ":" +
           std::to_string(fm.schema[i].type_length) + ":" + std::to_string(fm.schema[i].precision) + ":" + std::to_string(fm.schema[i].scale) + ",";
    for (auto& rg : fm.row_groups) {
      s += ";rg=" + std::to_string(rg.num_rows) + "[";
      for (auto& c : rg.columns)
        s += std::to_string(c.codec) + ":" + std::to_string(c.num_values) + ":" + std::to_string(c.data_page_offset) + ":" +
             std::to_string(c.dictionary_page_offset) + ":" + std::to_string(c.total_compressed) + ":" + std::to_string(c.total_uncompressed) + ",";
      s += "]";
    }
    if (out && cap) {
      size_t n = std::min(cap - 1, s.size());
      memcpy(out, s.data(), n);
      out[n] = 0;
    }
    return 0;
  });
}
