// Execution context: the MI355X counterpart of the reference's ExecutionContext
// (native/core/src/execution/jni_api.rs:306-365) — one per Spark task / plan handle.
#pragma once
#include <mutex>
#include <thread>
#include <atomic>
#include <hip/hip_runtime_api.h>

#include <deque>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <vector>

#include "arrow_abi.hpp"
#include "codegen.hpp"
#include "jit.hpp"
#include "kparams.h"
#include "plan.hpp"

namespace comet {
double process_clock_ms();      // exec_util.cpp

// Native memory one plan (one Spark task) holds, the way the reference's CometUnifiedMemoryPool reports it
// (native/core/src/execution/memory_pools/unified_pool.rs:64-150): every growth of PINNED HOST staging asks the host's memory manager
// (CometTaskMemoryManager.acquireMemory over JNI; comet_plan_set_memory_manager over the C ABI) and fails with the reference's
// "failed to acquire" error when less than the request is granted; every release hands the bytes back.  HBM is not Spark's to grant: it is
// held against the plan's own budget (spark.comet.gpu.memory.limit, bytes; 0 = the device's capacity).  A buffer remembers the account it
// was charged to, so it is credited back wherever and whenever it dies; releases that happen off the task thread are queued and handed
// to the manager at the task thread's next call (the JNI up-call needs that thread's JNIEnv).
struct MemAccount {
  int64_t (*acquire)(void* ctx, int64_t bytes) = nullptr;
  void (*release)(void* ctx, int64_t bytes) = nullptr;
  void* ctx = nullptr;
  long long task_id = 0;
  int64_t dev_limit = 0;
  std::atomic<int64_t> host_used{0}, host_peak{0}, dev_used{0}, dev_peak{0}, pending_release{0};
  std::thread::id owner;
  void grow_host(int64_t n);     // throws CometError when the manager grants less
  void shrink_host(int64_t n);
  void grow_dev(int64_t n);      // throws CometError over the budget
  void shrink_dev(int64_t n);
  void flush();                  // task thread: hand queued releases to the manager
  void detach();                 // the plan is gone: give back whatever is still charged and stop calling the manager
  std::mutex cb_mu;              // serialises up-calls against detach()
  bool detached = false;
};
// the account buffers allocated on this thread are charged to (set while a plan runs on it)
struct AccountScope {
  std::shared_ptr<MemAccount> prev;
  explicit AccountScope(std::shared_ptr<MemAccount> a);
  ~AccountScope();
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int dev = 0;
  std::shared_ptr<MemAccount> acct;
  void ensure(size_t n);
  void release();
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};
struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  std::shared_ptr<MemAccount> acct;
  void ensure(size_t n);
  void release();
  ~PinnedBuf() { release(); }
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
};

// one host-side result batch, buffers malloc'ed, exported through the Arrow C Data interface
struct HostColumn {
  DType type;
  int64_t length = 0;
  int64_t null_count = 0;
  std::vector<uint8_t> validity;  // bitmap, empty if null_count == 0
  std::vector<uint8_t> values;    // fixed width values / bit-packed booleans / int32 offsets (Utf8, List)
  std::vector<uint8_t> data;      // Utf8 bytes
  std::vector<HostColumn> children;   // Struct: one per field (each `length` rows); List: the one element column (offsets[length] elements)
};
struct HostBatch {
  int64_t rows = 0;
  std::vector<HostColumn> cols;
};

struct InputSource {
  int kind = 0;  // 0 = ArrowArrayStream (host memory), 1 = ArrowDeviceArrayStream (HBM resident)
  ArrowArrayStream* host = nullptr;
  ArrowDeviceArrayStream* dev = nullptr;
  bool exhausted = false;
};

struct DeviceColumnView {
  const void* data = nullptr;
  const uint8_t* valid = nullptr;
  const void* aux = nullptr;
  int64_t offset = 0;
  int fixed_len = -1;   // Utf8: every value has this byte length (aux then addresses the bytes directly); -1 = variable
  // Nested columns (DType::kids): a Struct has no buffers of its own but `valid`; its fields are `kids`, each as long as the struct.  A List's
  // `data` are its int32 offsets (rows + 1), kids[0] its elements — kid_rows of them.  kid_has_valid[i]: kids[i].valid is a bitmap.
  std::vector<DeviceColumnView> kids;
  std::vector<char> kid_has_valid;
  int64_t kid_rows = 0;
};

// a table resident in HBM (Arrow layout); owners keep pooled buffers / producer arrays alive
struct DevTable {
  int64_t rows = 0;
  std::vector<DType> types;
  std::vector<DeviceColumnView> cols;
  std::vector<bool> has_valid;
  std::vector<std::shared_ptr<void>> owners;
};

struct Staging {  // host→device staging of one input stream (one chunk at a time)
  std::vector<std::unique_ptr<PinnedBuf>> stage_vals, stage_valid, stage_aux;
  std::vector<std::unique_ptr<DevBuf>> dev_vals, dev_valid, dev_aux;
  std::vector<int> dict_index_width;   // per column: byte width of dictionary indices (0 = not dictionary-encoded)
  std::vector<std::shared_ptr<void>> nested_keep;   // device buffers of this chunk's nested columns (children, offsets, bitmaps): live as long as the set
  hipEvent_t busy = nullptr;           // recorded behind the last GPU work that reads this set
};

struct Variant {  // one JIT specialisation of the pipeline (per input-validity pattern)
  PipelineDesc desc;
  std::shared_ptr<LoadedModule> mod;
};

// parquet_scan.cpp: run fn(0..n-1) on the process-wide scan threads and wait
void scan_pool_parallel(size_t n, const std::function<void(size_t)>& fn);
// row-group / page-index selection of a NativeScan as JSON (parquet_scan.cpp; host only)
std::string parquet_prune_report(const Operator& native_scan, bool page_index, bool bloom_filters = true);
// host-staged PLAIN value bytes of one column of a NativeScan (parquet_scan.cpp; host only, a test hook)
std::vector<uint8_t> parquet_host_plain_values(const Operator& native_scan, size_t column);
// queue one task on the same threads (FIFO) without waiting
void scan_pool_submit(std::function<void()> fn);

extern std::atomic<int> g_kernel_times;      // COMET_KERNEL_TIMES / comet_set_kernel_times: per-kernel event pairs in ExecutionContext::launch

class ExecutionContext {
 public:
  ExecutionContext(OperatorP plan, uint64_t plan_hash, std::vector<std::pair<std::string, std::string>> config,
                   std::vector<InputSource> inputs, int batch_size, int device_id);
  ~ExecutionContext();

  // returns rows of the exported batch, or -1 at end of stream
  int64_t execute(ArrowArray** out_arrays, ArrowSchema** out_schemas, int n_out);
  // same, but the (single) result batch stays in HBM and is exported as ARROW_DEVICE_ROCM arrays
  int64_t execute_device(ArrowDeviceArray** out_arrays, ArrowSchema** out_schemas, int n_out);
  std::string metrics_proto();
  // the host's memory manager for this plan's pinned staging (see MemAccount); stats: host used / peak, device used / peak
  void set_memory_manager(int64_t (*acquire)(void*, int64_t), void (*release)(void*, int64_t), void* ctx, long long task_id);
  // Scalar subqueries (expr.proto:513-516; expressions/subquery.rs:72-180): the reference asks the JVM for a subquery's value when the expression is first
  // evaluated (CometScalarSubquery.isNull / getInt / … (planId, id)) — here every Subquery node of the plan becomes a Literal at the first executePlan.  The value
  // comes from the provider (JNI: those static methods; C ABI: comet_plan_set_subquery's table).  → false: no such subquery.  Value bytes: integers / dates /
  // timestamps 8-byte little-endian, booleans 1 byte, floats and doubles an 8-byte double, decimals BigInteger.toByteArray, strings / binary as they are.
  typedef std::function<bool(int64_t id, const DType& type, bool& is_null, std::string& value)> SubqueryProvider;
  void set_subquery_provider(SubqueryProvider p) { subquery_provider_ = std::move(p); }
  void set_subquery_value(int64_t id, bool is_null, std::string value) { subquery_values_[id] = {is_null, std::move(value)}; }
  void resolve_subqueries();
  void memory_stats(int64_t out[4]);
  std::shared_ptr<MemAccount> memory_account() const { return mem_; }
  const std::string& explain();
  // CPU-only: plan + generate + hiprtc-compile the all-valid variant (used by build()/tests w/o GPU)
  static std::string compile_only(OperatorP plan, uint64_t plan_hash);
  static std::string check_only(OperatorP plan, uint64_t plan_hash);

  std::string last_error;
  int last_error_kind = 0;

  // timing of the last run (for bench.py): device time of the main kernels measured with HIP events
  double last_kernel_ms = 0;
  int64_t last_kernel_launches = 0;
  // input-verification launches ahead of the main kernels (utf8_uniform_kernel over a device input's Utf8 offsets): timed apart, so a
  // roofline can charge a column's bytes to the kernel that reads them
  std::string kernel_times_json();
  double last_aux_ms = 0;
  int64_t last_aux_launches = 0;
  int64_t input_rows = 0;

 private:
  void run_to_completion();
  void start();
  std::vector<std::vector<int>> fixed_len_hint_;   // per device input, per column: producer-asserted uniform Utf8 length, -1 = none
  Variant& variant_for(const std::vector<bool>& has_valid, const std::vector<int>& str_fixed_len);
  void process_chunk(const std::vector<DeviceColumnView>& cols, const std::vector<bool>& has_valid, int64_t n);
  void finish_aggregate();
  void finish_grouped();
  DevTable grouped_to_device();
  std::vector<DType> infer_schema(const Operator& op);
  DevTable materialize(const Operator& op);
  DevTable scan_parquet(const Operator& native_scan);
  int64_t launch_fused_filter(Variant& v, CometKParams& prm, int64_t n);
  DevTable run_chain_to_device(const Operator& top, const DevTable& in);
  void extend_derived(DevTable& in, const std::vector<DerivedCol>& derived);
  void extend_derived_strfn(DevTable& in, const DerivedCol& dc);
  DevTable hash_join(const Operator& j, const DevTable& l, const DevTable& r);
  DevTable hash_join_impl(const Operator& node, const Operator& j, const DevTable& l, const DevTable& r, const std::string& key_suffix,
                          const JoinFusion* fused_probe = nullptr, const JoinFusion* fused_build = nullptr);
  // Probe-side fusion (codegen.hpp JoinFusion): decided per join at createPlan (infer_schema)
  struct FusedProbe { const Operator* source = nullptr; JoinFusion fu; };
  bool plan_fused_probe(const Operator& join, const std::vector<DType>& build_types, PipelineDesc& desc);
  bool plan_fused_build(const Operator& join, FusedProbe& fb);
  DevTable sort_table(const Operator& s, const DevTable& in);
  std::shared_ptr<DevBuf> sort_key_planes(const Operator& s, const DevTable& in, int& W, std::vector<int64_t>* str_len = nullptr, bool measure_only = false);
  DevTable literal_table(const std::vector<std::vector<ExprP>>& rows, const std::vector<DType>& types);
  DevTable take_rows(const DevTable& in, const uint32_t* dev_perm, int64_t first, int64_t rows, std::shared_ptr<DevBuf> perm_owner);
  DevTable nested_aggregate(const Operator& agg);
  DevTable write_shuffle(const Operator& sw);
  DevTable expand(const Operator& ex, const DevTable& in);
  DevTable explode(const Operator& ex, const DevTable& in);
  DevTable window(const Operator& w, const DevTable& in);
  void prepare_dict_keys(DevTable& src);
  static bool is_source(const Operator& op, const Operator* chain_top);
  typedef std::function<std::pair<const DevTable*, int>(int)> GatherSource;   // OutCol::gather_src → (table, column)
  DevTable outputs_to_table(Variant& v, const std::vector<std::shared_ptr<DevBuf>>& vals, const std::vector<std::shared_ptr<DevBuf>>& valid_bytes,
                            int64_t rows, const GatherSource& gather_source = nullptr);
  // rows idx[0 … rows) of a column of ANY type (nested ones with their children) → a new resident column; ok_bytes (optional): rows whose byte is 0 come out NULL / empty
  DeviceColumnView take_column(const DeviceColumnView& src, const DType& t, bool has_valid, const uint32_t* idx, const uint8_t* ok_bytes, int64_t rows, bool& out_has_valid,
                               std::vector<std::shared_ptr<void>>& owners);
  void take_utf8(const DeviceColumnView& src, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits, int64_t rows,
                 DeviceColumnView& out, std::vector<std::shared_ptr<void>>& owners);
  void table_to_host_batches(const DevTable& t);
  DevTable host_batch_to_table(const HostBatch& b);
  bool pull_host_chunk();
  bool pull_device_batch();
  bool pull_host_table(size_t input, const std::vector<DType>& types, int64_t max_rows, std::vector<DeviceColumnView>& views,
                       std::vector<bool>& has_valid, int64_t& rows);
  bool pull_device_table(size_t input, const std::vector<DType>& types, std::vector<DeviceColumnView>& views, std::vector<bool>& has_valid,
                         int64_t& rows, std::shared_ptr<void>& keepalive);
  void validate_input_schema(size_t input, const std::vector<DType>& types);
  void export_batch(HostBatch& b, ArrowArray** out_arrays, ArrowSchema** out_schemas, int n_out);
  void check_device_errors();
  void raise_device_errors(uint32_t flags);
  // the QueryContexts of a pipeline's raise sites (PipelineDesc::site_contexts), kept for the errors this context's kernels may raise
  // A site's id is its content + its ordinal INSIDE its pipeline (the kernel text stays independent of the plan around it), so two pipelines of
  // one plan can hold sites with the same id and different SQL fragments.  Errors are checked after each pipeline's launches: the
  // pipeline noted last answers first; an id that two pipelines disagree on and that the last one does not hold reports no fragment at all
  // rather than the wrong one.
  void note_sites(const PipelineDesc& d) {
    site_ctx_last_.clear();
    for (auto& kv : d.site_contexts) {
      site_ctx_last_[kv.first] = kv.second;
      auto it = site_ctx_.find(kv.first);
      if (it == site_ctx_.end()) site_ctx_[kv.first] = kv.second;
      else if (it->second != kv.second && !(it->second && kv.second && it->second->sql_text == kv.second->sql_text && it->second->start_index == kv.second->start_index &&
                                            it->second->stop_index == kv.second->stop_index))
        site_ctx_ambiguous_.insert(kv.first);
    }
    for (int k = 0; k < 2; k++) if (d.agg_ctx[k]) agg_ctx_[k] = d.agg_ctx[k];
  }
  const QueryContext* site_context(uint32_t id) const {
    auto a = site_ctx_last_.find(id);
    if (a != site_ctx_last_.end()) return a->second.get();
    if (site_ctx_ambiguous_.count(id)) return nullptr;
    auto b = site_ctx_.find(id);
    return b == site_ctx_.end() ? nullptr : b->second.get();
  }
  std::map<uint32_t, std::shared_ptr<QueryContext>> site_ctx_last_;
  std::set<uint32_t> site_ctx_ambiguous_;
  std::shared_ptr<QueryContext> agg_ctx_[2];      // … and of its ANSI decimal sum / average (PipelineDesc::agg_ctx)
  std::map<uint32_t, std::shared_ptr<QueryContext>> site_ctx_;
  void read_small(void* dst, const void* dev_src, size_t n);
  void write_small(void* dev_dst, const void* src, size_t n);
  void timed_begin();
  void timed_end();
  void collect_timings();
  void launch(Variant& v, const char* kernel, int grid, CometKParams& prm, int block = 256);

  OperatorP plan_;
  uint64_t plan_hash_ = 0;
  std::vector<std::pair<std::string, std::string>> config_;
  std::vector<InputSource> inputs_;
  int batch_size_;
  int device_id_;
  int64_t chunk_rows_;
  hipStream_t stream_ = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> timed_;   // (start, stop) around each main-kernel launch
  size_t timed_done_ = 0;
  struct KtPending { std::string name; hipEvent_t a, b; };
  std::vector<KtPending> kt_pending_;                        // COMET_KERNEL_TIMES=1: one event pair per generated-kernel launch
  std::map<std::string, std::pair<double, int64_t>> kernel_times_;
  hipEvent_t aux_ev_[2] = {nullptr, nullptr};              // (start, stop) around pull_device_table's verification launches; aux_n_ of them
  int aux_n_ = 0;
  bool started_ = false, finished_ = false;
  std::vector<DType> in_types_;
  std::map<std::string, Variant> variants_;
  Variant* agg_variant_ = nullptr;   // variant whose accumulator layout the partials follow
  std::string explain_;
  SinkKind sink_ = SinkKind::Output;
  bool has_join_ = false;
  SubqueryProvider subquery_provider_;
  std::map<int64_t, std::pair<bool, std::string>> subquery_values_;
  bool subqueries_resolved_ = false;
  int64_t strfn_rows_ = 0;          // rows that went through the derived string functions' kernels
  int64_t split_rows_ = 0;          // rows that went through split's kernels (tests: the device path ran)
  bool materialize_root_ = false;   // plain Scan chain whose outputs need the materialising path (Utf8 pass-through)
  bool compile_in_infer_ = false;
  std::vector<const Operator*> nested_aggs_;
  struct ExpandPart {
    OperatorP proj;                 // Projection(non-NULL-string expressions) over a synthetic Scan of the child's schema
    std::vector<int> out_col;       // functor output j → Expand output column
    std::vector<int> null_cols;     // Expand output columns that are an untyped / Utf8 NULL literal in this projection
  };
  struct ExpandInfo {
    std::vector<ExpandPart> parts;
    std::vector<OutCol> out_cols;   // unified output schema
  };
  std::map<const Operator*, ExpandInfo> expand_info_;
  std::map<const Operator*, OperatorP> explode_proj_;   // Explode → the synthetic Projection of the columns it carries alongside
  std::map<const Operator*, OperatorP> window_psort_, window_osort_;   // Window → synthetic Sorts describing its partition / order keys
  std::map<const Operator*, OperatorP> range_sort_, range_bsort_;   // ShuffleWriter(range) → synthetic Sort over its rows / its boundary rows
  std::shared_ptr<void> planes_owner_;
  std::map<const Operator*, OperatorP> shuffle_projs_;   // ShuffleWriter with computed hash expressions → synthetic Projection(child ++ hash exprs)
  int64_t shuffle_bytes_written_ = 0, shuffle_data_size_ = 0, shuffle_staged_peak_ = 0;
  double shuffle_repart_ns_ = 0, shuffle_write_ns_ = 0;
  std::vector<int> dict_id_col_;                   // per source column: appended row-index column standing in for a long Utf8 group key
  bool device_result_ = false;                     // the grouped result stays in HBM (nested aggregate / execute_device)
  DevTable dict_src_;                              // the aggregate's input while such keys are in flight
  std::map<const Operator*, FusedProbe> fused_probe_;   // joins that read their probe chain's source table directly
  std::map<const Operator*, FusedProbe> fused_build_;   // … and their build chain's (a Scan leaf's table)
  int64_t ramp_rows_ = 0;                          // host streams: size of the next chunk while the pipeline fills (doubles up to chunk_rows_)
  bool fuse_probe_ = true;                         // spark.comet.gpu.join.fuseProbe
  int fuse_build_ = 1;                             // spark.comet.gpu.join.fuseBuild: 0 false, 1 true (chains whose Filters only drop NULLs), 2 "always"
  std::set<const Operator*> smj_needs_sort_;       // sort-merge joins whose output order is observable (others skip the sort)
  std::map<const Operator*, OperatorP> smj_sorts_;  // SortMergeJoin node → synthetic Sort over its output       // aggregates that are not the plan root (materialised by sub-contexts)
  const Operator* root_source_ = nullptr;          // Scan or HashJoin the root chain reads from
  std::map<const Operator*, int> node_id_;          // preorder ordinal (plan-cache key of sub-pipelines)
  std::map<const Operator*, size_t> scan_input_;    // Scan leaf → input stream index
  int64_t join_build_rows_ = 0, join_probe_rows_ = 0, join_keymap_bytes_ = 0, join_direct_maps_ = 0, join_bucket_tables_ = 0, join_bitmap_only_ = 0, join_mono_tables_ = 0, join_fused_builds_ = 0;
  int small_write_slot_ = 0;         // write_small's ring of staging slots
  // the partitioned merging aggregate (exec_pipeline.cpp try_partitioned_merge): its input is known to be ONE chunk; the emitted result
  bool single_chunk_hint_ = false, part_result_ready_ = false;
  DevTable part_result_;
  int64_t part_merges_ = 0;
  bool try_partitioned_merge(Variant& v, CometKParams& prm, int64_t n);
  struct DevPending {      // a device batch pulled one ahead of its turn
    std::vector<DeviceColumnView> views;
    std::vector<bool> has_valid;
    int64_t rows = 0;
    std::shared_ptr<void> keep;
    bool valid = false;
  };
  DevPending dev_pending_;
  bool dev_stream_done_ = false;
  int64_t dev_chunks_seen_ = 0;
  bool join_no_bucket_ = false;      // set while a join whose bucket table overflowed re-runs over the chained table
  int64_t bytes_scanned_ = 0;
  int64_t row_groups_pruned_ = 0;
  int64_t row_groups_pruned_bloom_ = 0;    // … of them, by a column chunk's Bloom filter (the statistics had not ruled them out)
  std::shared_ptr<MemAccount> mem_ = std::make_shared<MemAccount>();
  int64_t rows_pruned_page_index_ = 0;     // rows the Parquet page index ruled out (never decoded)
  int64_t pages_inflated_on_device_ = 0;   // data pages decompressed by snappy_kernels.hip

  std::vector<std::unique_ptr<Staging>> staging_;   // per input stream
  int stage_parity_ = 0;                            // which of the two staging sets the next streamed chunk uses
  std::vector<bool> schema_checked_;                // per input stream
  // ScanExec casts a stream column whose Arrow type differs from the declared one (operators/scan.rs:281-291, arrow cast_with_options,
  // safe mode: a value that does not fit becomes NULL).  Per input stream and column: the source column's Arrow format when a cast is
  // needed ("" = the types agree); the conversion happens in the pinned staging copy every host batch goes through anyway
  std::vector<std::vector<std::string>> scan_cast_from_;

  // aggregate state
  DevBuf partials_;
  int64_t n_partials_ = 0;
  // exact Float64 sums: the fixed-point scale of every sum of the aggregate (codegen.hpp kFixDefaultScale until the data say otherwise)
  std::vector<int> fix_scales_;
  bool fix_has_state_ = false;        // an earlier chunk already contributed to the accumulators at these scales
  int fix_attempts_ = 0;
  uint64_t groups_committed_ = 0;     // grouped: groups in the global table after the last completed chunk
  long long packed_fix_scales(const PipelineDesc& d, int first = 0);
  bool adjust_fix_scales(const PipelineDesc& d, const uint64_t* aux, std::vector<int>& shift_right);
  DevBuf err_flags_;
  PinnedBuf result_host_;
  PinnedBuf small_host_;   // 4 KiB pinned scratch for flag / count read-backs
  PinnedBuf export_host_;  // a small result's buffers land here side by side (one copy kernel instead of a hipMemcpy per buffer)
  DevBuf group_table_, group_backup_;
  int64_t group_cap_ = 0;
  bool group_table_clear_ = false;    // grouped: the global table holds nothing but zeroes (no chunk has gone into it): its checkpoint is a memset
  DevBuf scratch_mask_, scratch_counts_;
  std::vector<std::unique_ptr<DevBuf>> out_vals_, out_valid_;
  DevBuf emit_arena_;   // finish_grouped: values + validity bytes of every output column of one emit

  std::deque<HostBatch> ready_;
  int64_t output_rows_ = 0;
  double elapsed_compute_ns_ = 0;
};

}  // namespace comet
