// Internal to the execution engine's translation units (exec.cpp and exec_*.cpp): kernel launchers of the static .hip files, buffer / stream
// pools, small host helpers, the planned-pipeline cache.  Not part of any interface.
#pragma once
#include "exec.hpp"
#include "shuffle_format.hpp"

#include <cerrno>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>

extern "C" int comet_launch_dict_gather_fixed(const void* idx, int iw, const uint8_t* idx_valid, const uint8_t* dict, const uint8_t* dict_valid,
                                              int width, int64_t n, uint8_t* out, uint8_t* out_valid_bytes, void* stream);
extern "C" int comet_launch_dict_gather_str_len(const void* idx, int iw, const uint8_t* idx_valid, const int32_t* dict_offs, const uint8_t* dict_valid,
                                                int64_t n, uint32_t* lengths, uint8_t* out_valid_bytes, void* stream);
extern "C" int comet_launch_dict_gather_str_copy(const void* idx, int iw, const uint8_t* valid_bytes, const int32_t* dict_offs, const uint8_t* dict_bytes,
                                                 int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream);
extern "C" void pq_launch_u32_scan(const uint32_t* in, int64_t n, uint64_t* tiles, int32_t* out, void* st);
extern "C" void pq_launch_pack(const uint8_t* bytes, uint8_t* bitmap, int64_t n, void* st);
extern "C" int comet_launch_window_default(int width, const uint8_t* inside, int64_t n, const void* value, void* data, uint8_t* ok_bytes, void* stream);
extern "C" int comet_launch_window_widen(int width, const void* src, const uint8_t* valid_bits, int64_t n, void* out128, void* hi128, uint32_t* ok, void* stream);
extern "C" int comet_launch_scan128(const void* in128, int64_t n, void* tiles, void* out128, void* stream);
extern "C" int comet_launch_window_running_extreme(const void* vals128, const uint32_t* ok, const int32_t* sp, int64_t n, int backward, int is_max, void* local, void* tiles, void* out_v,
                                                   uint8_t* out_has, void* stream);
extern "C" int comet_launch_window_minmax(int is_max, int lo_kind, int64_t lo_off, int hi_kind, int64_t hi_off, const void* vals128, const uint32_t* ok, const void* P, const uint8_t* Ph,
                                          const void* Q, const uint8_t* Qh, const int32_t* sp, const int32_t* sg, const uint32_t* first_part, const uint32_t* first_peer, int64_t n,
                                          int out_width, void* out, uint8_t* out_ok, void* stream);
extern "C" int comet_launch_window_range_bounds(int width, const void* keys, const uint8_t* valid, const int32_t* sp, const uint32_t* first_part, int64_t n, int desc, int nulls_first,
                                                int has_lo, int64_t dlo, int has_hi, int64_t dhi, int32_t* out_lo, int32_t* out_hi, void* stream);
extern "C" int comet_launch_window_valid_flags(const uint8_t* valid, int64_t n, uint32_t* flags, void* stream);
extern "C" int comet_launch_window_pick(int mode, int64_t nth, int lo_kind, int64_t lo_off, int hi_kind, int64_t hi_off, const int32_t* C, const int32_t* sp, const int32_t* sg,
                                        const uint32_t* first_part, const uint32_t* first_peer, int64_t n, uint32_t* idx, uint8_t* ok, void* stream);
extern "C" int comet_launch_window_agg(int fn, int lo_kind, int64_t lo_off, int hi_kind, int64_t hi_off, const void* S128, const void* SH128, const int32_t* C, const int32_t* sp, const int32_t* sg, const uint32_t* first_part,
                                       const uint32_t* first_peer, int64_t n, const void* bound16, const void* scaler16, const void* avg_bound16, void* out, uint8_t* out_ok,
                                       void* stream);
extern "C" int comet_launch_window_flags(const uint8_t* part_planes, int Wp, const uint8_t* order_planes, int Wo, int64_t n, uint32_t* fpart, uint32_t* fpeer, void* stream);
extern "C" int comet_launch_window_first(const uint32_t* fpart, const int32_t* sp, const uint32_t* fpeer, const int32_t* sg, int64_t n, uint32_t* first_part,
                                         uint32_t* first_peer, void* stream);
extern "C" int comet_launch_window_rank(int kind, int64_t arg, const int32_t* sp, const int32_t* sg, const uint32_t* first_part, const uint32_t* first_peer, int64_t n,
                                        void* out, void* stream);
extern "C" int comet_launch_window_offset(int64_t shift, const int32_t* sp, const uint32_t* first_part, int64_t n, uint32_t* idx, uint8_t* ok, void* stream);
extern "C" int comet_launch_window_offset_valid(const uint32_t* idx, const uint8_t* ok, const uint8_t* src_valid_bits, int64_t n, uint8_t* out_ok, void* stream);
struct CometConcatArgs {      // the parts of a concat (exchange_kernels.hip): a Utf8 column (offsets / bytes / first row) or a literal (bytes = the literal, offsets = NULL)
  int32_t n;
  int32_t lit_len[8];
  const int32_t* offs[8];
  const uint8_t* bytes[8];
  int64_t first[8];
};
extern "C" int comet_launch_concat_lengths(const CometConcatArgs* a, const uint32_t* rows, const uint8_t* ok_bytes, int64_t n, uint32_t* lengths, void* stream);
extern "C" int comet_launch_concat_copy(const CometConcatArgs* a, const uint32_t* rows, const uint8_t* ok_bytes, int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream);
extern "C" int comet_launch_strcase_lengths(const void* views, const uint8_t* ok_bytes, const int32_t* src_offs, const uint8_t* src_bytes, int64_t n, int mode, uint32_t* lengths, void* stream);
extern "C" int comet_launch_strcase_write(const void* views, const uint8_t* ok_bytes, const int32_t* src_offs, const uint8_t* src_bytes, int64_t n, int mode, const int32_t* out_offs,
                                          uint8_t* out_bytes, void* stream);
extern "C" int comet_launch_join_part_scan(uint32_t* cnt, int g, int np, uint32_t* tot, uint32_t limit, uint64_t* over, void* stream);
extern "C" int comet_launch_popcount128(const void* blocks, int64_t n, uint32_t* counts, void* stream);
extern "C" int comet_launch_strfmt_lengths(int kind, long long arg, const void* vals128, const uint8_t* ok_bytes, int64_t n, uint32_t* lengths, void* stream);
extern "C" int comet_launch_strfmt_write(int kind, long long arg, const void* vals128, const uint8_t* ok_bytes, int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream);
extern "C" int comet_launch_strfn_len(int op, const int32_t* offs, const uint8_t* bytes, const uint8_t* valid_bits, int64_t valid_first, int64_t n, const uint8_t* a, int32_t na, const uint8_t* b,
                                      int32_t nb, int64_t k, uint32_t* lengths, uint32_t* too_long, void* stream);
extern "C" int comet_launch_strfn_write(int op, const int32_t* offs, const uint8_t* bytes, const uint8_t* valid_bits, int64_t valid_first, int64_t n, const uint8_t* a, int32_t na, const uint8_t* b,
                                        int32_t nb, int64_t k, const int32_t* out_offs, uint8_t* out, void* stream);
extern "C" int comet_launch_split_count(const int32_t* offs, const uint8_t* bytes, const uint8_t* valid_bits, int64_t valid_first, int64_t n, const uint32_t* prog, const uint32_t* prog2, int32_t limit,
                                        uint32_t* counts, void* stream);
extern "C" int comet_launch_split_write(const int32_t* offs, const uint8_t* bytes, const uint8_t* valid_bits, int64_t valid_first, int64_t n, const uint32_t* prog, const uint32_t* prog2, int32_t limit,
                                        const int32_t* list_offs, void* views, void* stream);
extern "C" int comet_launch_strview_lengths(const void* views, const uint8_t* ok_bytes, int64_t n, const uint8_t* pattern, int32_t pattern_bytes, uint32_t* lengths, void* stream);
extern "C" int comet_launch_strview_copy(const void* views, const uint8_t* ok_bytes, const int32_t* src_offs, const uint8_t* src_bytes, int64_t n, const uint8_t* pattern,
                                         int32_t pattern_bytes, int pad_left, const int32_t* out_offs, uint8_t* out_bytes, void* stream);
extern "C" int comet_launch_str16_lengths(const void* packed, const uint8_t* ok_bytes, int64_t n, uint32_t* lengths, void* stream);
extern "C" int comet_launch_str16_copy(const void* packed, const int32_t* offsets, int64_t n, uint8_t* bytes, void* stream);
extern "C" int comet_launch_str_max_len(const int32_t* offs, int64_t n, uint32_t* out_max, void* stream);
extern "C" int comet_launch_str_dict_build(const int32_t* offs, const uint8_t* bytes, const uint8_t* valid_bits, int64_t n, uint32_t* table, int64_t slots,
                                           int64_t* rep, void* stream);
extern "C" int comet_launch_str_dict_lookup(const int32_t* build_offs, const uint8_t* build_bytes, const uint32_t* table, int64_t slots, const int32_t* offs,
                                            const uint8_t* bytes, const uint8_t* valid_bits, int64_t n, int64_t* rep, uint8_t* ok, void* stream);
extern "C" int64_t comet_partition_tiles(int64_t n);
extern "C" int64_t comet_partition_scratch_bytes(int64_t n, int32_t P);
extern "C" int comet_launch_fill(int width, void* dst, int64_t n, const void* value, void* stream);
extern "C" int comet_launch_copy_small(const void* descs, int n, uint64_t longest, void* stream);   // descs: pinned { const uint8_t* src; uint8_t* dst; uint64_t len; }
extern "C" int comet_launch_murmur3(int type_id, int precision, const void* values, const uint8_t* validity, const void* aux, int64_t n, uint32_t* hashes, void* stream);
extern "C" int comet_launch_pmod(const uint32_t* hashes, int64_t n, int32_t np, int32_t* out, void* stream);
extern "C" int comet_launch_partition_indices(const int32_t* pids, int64_t n, int32_t P, uint64_t* hist, uint32_t* bad, int64_t* starts,
                                              uint32_t* row_indices, void* stream);
extern "C" int comet_launch_take(int width, const void* src, const uint32_t* idx, int64_t n, void* dst, void* stream);
extern "C" int comet_launch_take_utf8_lengths(const int32_t* offs, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits, int64_t n,
                                              uint32_t* lengths, void* stream);
extern "C" int comet_launch_explode_counts(const int32_t* offs, const uint8_t* valid_bits, int64_t n, int outer, uint32_t* counts, void* stream);
extern "C" int comet_launch_explode_indices(const int32_t* offs, const uint8_t* valid_bits, const uint8_t* elem_valid_bits, int64_t n, const int32_t* out_offs, uint32_t* row_idx,
                                            uint32_t* elem_idx, uint8_t* has_elem, int32_t* pos, uint8_t* elem_ok, void* stream);
extern "C" int comet_launch_take_list_indices(const int32_t* offs, const uint32_t* idx, int64_t n, const int32_t* out_offs, uint32_t* elem_idx, void* stream);
extern "C" int comet_launch_take_utf8_copy(const int32_t* offs, const uint8_t* bytes, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits,
                                           int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream);
extern "C" void pq_launch_u32_scan(const uint32_t* in, int64_t n, uint64_t* tiles, int32_t* out, void* st);
extern "C" int comet_launch_sort_small(const uint8_t* planes, int64_t n, const uint32_t* cand, int m, const int* plane_idx, int nv, uint32_t* out, void* stream);
extern "C" int comet_launch_sort_iota(uint32_t* perm, int64_t n, uint32_t first, void* stream);
extern "C" int comet_launch_sort_gather_digit(const uint8_t* plane, const uint32_t* perm, int64_t n, int32_t* digit, void* stream);
extern "C" int comet_launch_range_partition_ids(const uint8_t* planes, int64_t n, int W, const uint8_t* bkeys, int B, int32_t* pids, void* stream);
extern "C" int comet_launch_sort_plane_varies(const uint8_t* planes, int64_t n, int W, uint32_t* flags, void* stream);
extern "C" int comet_launch_sort_hist256(const uint8_t* plane, const uint32_t* cand, int64_t m, uint64_t* hist, void* stream);
extern "C" int comet_launch_sort_select(const uint8_t* plane, const uint32_t* cand, int64_t m, int dstar, uint32_t* sure, uint32_t* next_cand, uint32_t* counters,
                                        void* stream);
extern "C" int comet_launch_fix_rescale(uint64_t* base, int64_t count, int64_t stride_words, int32_t word_off, int32_t shift, void* stream);
extern "C" int comet_launch_utf8_uniform(const int32_t* offsets, int64_t n, int32_t L, uint32_t* flag, void* stream);

namespace comet {
// a resident column of any type, children included, into host memory (exec.cpp: nested results and nested shuffle payloads)
HostColumn download_column(const DeviceColumnView& v, const DType& t, bool has_valid, int64_t rows, hipStream_t st);

namespace detail {
void plan_execution_begins();      // exec_memory.cpp: how many plans execute right now decides how threads wait for the device
void plan_execution_ends();
int plans_executing();

// per-process stream / event pools (exec_memory.cpp)
hipStream_t pool_get_stream(int dev);
hipStream_t shared_copy_stream(int dev);      // one per device for all scans: never returned to the pool, never synchronised as a whole
void pool_put_stream(int dev, hipStream_t s);
hipEvent_t pool_get_event(int dev);
void pool_put_event(int dev, hipEvent_t e);

// small host helpers (exec_util.cpp)
int fixed_width(const DType& t);
inline int out_width(const OutCol& oc) { return (oc.view_src >= 0 || oc.fmt_kind) ? 16 : (oc.gather_src >= 0 || !oc.concat_cols.empty()) ? 4 : oc.packed_string ? 16 : (oc.type.id == TypeId::Bool ? 1 : fixed_width(oc.type)); }
void bit_append(uint8_t* dst, int64_t dst_off, const uint8_t* src, int64_t src_off, int64_t n);
bool nested_schema_matches(const ArrowSchema* f, const DType& t);
void append_nested_rows(HostColumn& dst, const ArrowArray* a, const DType& t, int64_t off, int64_t len);
void bit_fill_ones(uint8_t* dst, int64_t dst_off, int64_t n);
bool format_matches(const char* fmt, const DType& t);
struct SrcFmt {
  enum Cls { Unknown, Int, UInt, F32, F64, Date32, Date64, Ts, Dec, Utf8, LargeUtf8, Bool } cls = Unknown;
  int width = 0;
  int64_t per_second = 0;   // Ts: ticks per second
  int p = 0, s = 0;         // Dec
};
SrcFmt parse_src_format(const char* fmt);
bool scan_cast_supported(const SrcFmt& f, const DType& t);
bool scan_cast_value(const SrcFmt& f, const char* src, int64_t i, const DType& t, char* dst);
const Operator* find_scan(const Operator* op);
std::string validity_key(const std::vector<bool>& v);
u128 pow10_u128_host(int p);
struct Timer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ns() const { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

// Planned pipelines are shared by every task that runs the same plan bytes (exec.cpp)
struct PlannedVariant {
  PipelineDesc desc;
  std::shared_ptr<CodeObject> code;
};
extern std::mutex g_plan_mu;                                                    // guards the cache (join variants are cached under their own keys: exec_join.cpp)
extern std::map<std::string, std::shared_ptr<PlannedVariant>> g_plan_cache;
std::shared_ptr<PlannedVariant> planned_variant(const Operator& plan, uint64_t plan_hash, const std::vector<bool>& has_valid, bool compile,
                                                const std::vector<DType>* source_types = nullptr, const std::vector<int>* str_fixed_len = nullptr,
                                                const std::vector<int>* dict_id_col = nullptr);

}  // namespace detail
using namespace detail;
}  // namespace comet
