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

namespace comet {

// ---------------------------------------------------------------------------------------------
// buffers
// ---------------------------------------------------------------------------------------------
// Per-process pools.  A Spark executor runs thousands of short tasks with the same plan: hipMalloc/hipFree
// (device-synchronising), hipHostMalloc and stream/event creation per task would dominate a 0.2 ms kernel.
// Blocks are recycled by power-of-two size class and keyed by device.
namespace {
struct Pools {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void*>> dev_free;   // (device, class bytes) → blocks
  std::map<size_t, std::vector<void*>> pinned_free;
  // idle blocks are kept for reuse up to these caps (a long-lived executor must not pin HBM it no longer uses);
  // COMET_POOL_MAX_BYTES / COMET_PINNED_POOL_MAX_BYTES override (bytes)
  std::map<int, size_t> dev_cached;
  size_t pinned_cached = 0;
  size_t dev_cap = (size_t)96 << 30, pinned_cap = (size_t)16 << 30;
  Pools() {
    if (const char* e = getenv("COMET_POOL_MAX_BYTES")) dev_cap = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("COMET_PINNED_POOL_MAX_BYTES")) pinned_cap = (size_t)strtoull(e, nullptr, 10);
  }
  std::map<int, std::vector<hipStream_t>> streams;
  std::map<int, std::vector<hipEvent_t>> events;
};
Pools& pools() {
  static Pools* p = new Pools();  // intentionally leaked: HIP may already be torn down at process exit
  return *p;
}
size_t size_class(size_t n) {
  size_t c = 256;
  while (c < n) c <<= 1;
  return c;
}
int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d;
}
}  // namespace

namespace {
thread_local std::shared_ptr<MemAccount> t_account;
void raise_peak(std::atomic<int64_t>& peak, int64_t v) {
  int64_t p = peak.load(std::memory_order_relaxed);
  while (v > p && !peak.compare_exchange_weak(p, v, std::memory_order_relaxed)) {}
}
}  // namespace
AccountScope::AccountScope(std::shared_ptr<MemAccount> a) : prev(t_account) { t_account = std::move(a); }
AccountScope::~AccountScope() { t_account = prev; }
void MemAccount::flush() {
  if (std::this_thread::get_id() != owner) return;
  std::lock_guard<std::mutex> lk(cb_mu);
  const int64_t n = pending_release.exchange(0);
  if (n > 0 && release) release(ctx, n);
}
void MemAccount::detach() {
  std::lock_guard<std::mutex> lk(cb_mu);
  pending_release.store(0);
  const int64_t left = host_used.load();
  if (left > 0 && release) release(ctx, left);     // buffers that outlive the plan (exported batches) are no longer the task's
  acquire = nullptr;
  release = nullptr;
  detached = true;
}
void MemAccount::grow_host(int64_t n) {
  if (n <= 0) return;
  flush();
  {
    std::lock_guard<std::mutex> lk(cb_mu);
    if (acquire && std::this_thread::get_id() == owner) {
      const int64_t got = acquire(ctx, n);
      if (got < n) {
        if (got > 0 && release) release(ctx, got);
        throw CometError("Task " + std::to_string(task_id) + " failed to acquire " + std::to_string(n) + " bytes, only got " + std::to_string(got < 0 ? 0 : got) +
                         ". Reserved: " + std::to_string(host_used.load()));
      }
    }
  }
  raise_peak(host_peak, host_used.fetch_add(n) + n);
}
void MemAccount::shrink_host(int64_t n) {
  if (n <= 0) return;
  host_used.fetch_sub(n);
  std::lock_guard<std::mutex> lk(cb_mu);
  if (!release || detached) return;
  if (std::this_thread::get_id() == owner) {
    const int64_t queued = pending_release.exchange(0);
    release(ctx, n + queued);
  } else {
    pending_release.fetch_add(n);
  }
}
void MemAccount::grow_dev(int64_t n) {
  if (n <= 0) return;
  const int64_t now = dev_used.fetch_add(n) + n;
  if (dev_limit > 0 && now > dev_limit) {
    dev_used.fetch_sub(n);
    throw CometError("Task " + std::to_string(task_id) + ": GPU memory budget exceeded (spark.comet.gpu.memory.limit = " + std::to_string(dev_limit) +
                     " bytes, " + std::to_string(now - n) + " in use, " + std::to_string(n) + " more requested)");
  }
  raise_peak(dev_peak, now);
}
void MemAccount::shrink_dev(int64_t n) {
  if (n > 0) dev_used.fetch_sub(n);
}

void DevBuf::ensure(size_t n) {
  if (n <= cap) return;
  release();
  const size_t cls = size_class(n);
  dev = current_device();
  if (t_account) {
    t_account->grow_dev((int64_t)cls);      // may throw: over the plan's HBM budget
    acct = t_account;
  }
  try {
    {
      std::lock_guard<std::mutex> lk(pools().mu);
      auto& fl = pools().dev_free[{dev, cls}];
      if (!fl.empty()) {
        p = fl.back();
        fl.pop_back();
        pools().dev_cached[dev] -= cls;
        cap = cls;
        return;
      }
    }
    if (hipMalloc(&p, cls) != hipSuccess) {
      // out of memory: hand every idle block of this device back to the driver and try once more
      (void)hipGetLastError();
      std::vector<void*> victims;
      {
        std::lock_guard<std::mutex> lk(pools().mu);
        for (auto& kv : pools().dev_free)
          if (kv.first.first == dev) {
            victims.insert(victims.end(), kv.second.begin(), kv.second.end());
            kv.second.clear();
          }
        pools().dev_cached[dev] = 0;
      }
      for (void* v : victims) (void)hipFree(v);
      p = nullptr;
      HIP_CHECK(hipMalloc(&p, cls));
    }
    cap = cls;
  } catch (...) {
    if (acct) acct->shrink_dev((int64_t)cls);
    acct.reset();
    p = nullptr;
    throw;
  }
}
void DevBuf::release() {
  if (p) {
    bool keep;
    {
      std::lock_guard<std::mutex> lk(pools().mu);
      keep = pools().dev_cached[dev] + cap <= pools().dev_cap;
      if (keep) {
        pools().dev_free[{dev, cap}].push_back(p);
        pools().dev_cached[dev] += cap;
      }
    }
    if (!keep) (void)hipFree(p);   // over the cap: give the block back to the driver (hipFree waits for the device)
    if (acct) acct->shrink_dev((int64_t)cap);
  }
  acct.reset();
  p = nullptr;
  cap = 0;
}
void PinnedBuf::ensure(size_t n) {
  if (n <= cap) return;
  release();
  const size_t cls = size_class(n);
  if (t_account) {
    t_account->grow_host((int64_t)cls);     // may throw: the host's memory manager granted less
    acct = t_account;
  }
  try {
    {
      std::lock_guard<std::mutex> lk(pools().mu);
      auto& fl = pools().pinned_free[cls];
      if (!fl.empty()) {
        p = fl.back();
        fl.pop_back();
        pools().pinned_cached -= cls;
        cap = cls;
        return;
      }
    }
    HIP_CHECK(hipHostMalloc(&p, cls, hipHostMallocDefault));
    cap = cls;
  } catch (...) {
    if (acct) acct->shrink_host((int64_t)cls);
    acct.reset();
    p = nullptr;
    throw;
  }
}
void PinnedBuf::release() {
  if (p) {
    bool keep;
    {
      std::lock_guard<std::mutex> lk(pools().mu);
      keep = pools().pinned_cached + cap <= pools().pinned_cap;
      if (keep) {
        pools().pinned_free[cap].push_back(p);
        pools().pinned_cached += cap;
      }
    }
    if (!keep) (void)hipHostFree(p);
    if (acct) acct->shrink_host((int64_t)cap);
  }
  acct.reset();
  p = nullptr;
  cap = 0;
}

static hipStream_t pool_get_stream(int dev) {
  {
    std::lock_guard<std::mutex> lk(pools().mu);
    auto& v = pools().streams[dev];
    if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
  }
  hipStream_t s;
  HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  return s;
}
static void pool_put_stream(int dev, hipStream_t s) {
  std::lock_guard<std::mutex> lk(pools().mu);
  pools().streams[dev].push_back(s);
}
static hipEvent_t pool_get_event(int dev) {
  {
    std::lock_guard<std::mutex> lk(pools().mu);
    auto& v = pools().events[dev];
    if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; }
  }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}
static void pool_put_event(int dev, hipEvent_t e) {
  std::lock_guard<std::mutex> lk(pools().mu);
  pools().events[dev].push_back(e);
}

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
extern "C" int comet_launch_murmur3(int type_id, int precision, const void* values, const uint8_t* validity, const void* aux, int64_t n, uint32_t* hashes, void* stream);
extern "C" int comet_launch_pmod(const uint32_t* hashes, int64_t n, int32_t np, int32_t* out, void* stream);
extern "C" int comet_launch_partition_indices(const int32_t* pids, int64_t n, int32_t P, uint64_t* hist, uint32_t* bad, int64_t* starts,
                                              uint32_t* row_indices, void* stream);
extern "C" int comet_launch_take(int width, const void* src, const uint32_t* idx, int64_t n, void* dst, void* stream);
extern "C" int comet_launch_take_utf8_lengths(const int32_t* offs, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits, int64_t n,
                                              uint32_t* lengths, void* stream);
extern "C" int comet_launch_take_utf8_copy(const int32_t* offs, const uint8_t* bytes, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits,
                                           int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream);
extern "C" void pq_launch_u32_scan(const uint32_t* in, int64_t n, uint64_t* tiles, int32_t* out, void* st);
extern "C" int comet_launch_sort_iota(uint32_t* perm, int64_t n, uint32_t first, void* stream);
extern "C" int comet_launch_sort_gather_digit(const uint8_t* plane, const uint32_t* perm, int64_t n, int32_t* digit, void* stream);
extern "C" int comet_launch_range_partition_ids(const uint8_t* planes, int64_t n, int W, const uint8_t* bkeys, int B, int32_t* pids, void* stream);
extern "C" int comet_launch_sort_plane_varies(const uint8_t* planes, int64_t n, int W, uint32_t* flags, void* stream);
extern "C" int comet_launch_sort_hist256(const uint8_t* plane, const uint32_t* cand, int64_t m, uint64_t* hist, void* stream);
extern "C" int comet_launch_sort_select(const uint8_t* plane, const uint32_t* cand, int64_t m, int dstar, uint32_t* sure, uint32_t* next_cand, uint32_t* counters,
                                        void* stream);
extern "C" int comet_launch_fix_rescale(uint64_t* base, int64_t count, int64_t stride_words, int32_t word_off, int32_t shift, void* stream);
extern "C" int comet_launch_utf8_uniform(const int32_t* offsets, int64_t n, int32_t L, uint32_t* flag, void* stream);

namespace {

int fixed_width(const DType& t);
int out_width(const OutCol& oc) { return oc.view_src >= 0 ? 16 : oc.gather_src >= 0 ? 4 : oc.packed_string ? 16 : (oc.type.id == TypeId::Bool ? 1 : fixed_width(oc.type)); }

int fixed_width(const DType& t) {
  switch (t.id) {
    case TypeId::Int8: return 1;
    case TypeId::Int16: return 2;
    case TypeId::Int32: case TypeId::Date: case TypeId::Float: return 4;
    case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: case TypeId::Double: return 8;
    case TypeId::Decimal: return 16;
    case TypeId::Bool: return 0;  // bit-packed
    default: throw CometError("Unsupported column type in GPU scan: " + t.str());
  }
}

// append n bits from src (starting at bit src_off) to dst at bit dst_off
void bit_append(uint8_t* dst, int64_t dst_off, const uint8_t* src, int64_t src_off, int64_t n) {
  if (n <= 0) return;
  if ((dst_off & 7) == 0 && (src_off & 7) == 0) {
    int64_t full = n >> 3;
    memcpy(dst + (dst_off >> 3), src + (src_off >> 3), (size_t)full);
    int64_t rem = n & 7;
    if (rem) {
      uint8_t m = (uint8_t)((1u << rem) - 1);
      uint8_t& d = dst[(dst_off >> 3) + full];
      d = (uint8_t)((d & ~m) | (src[(src_off >> 3) + full] & m));
    }
    return;
  }
  for (int64_t i = 0; i < n; i++) {
    int64_t s = src_off + i, d = dst_off + i;
    uint8_t bit = (src[s >> 3] >> (s & 7)) & 1;
    if (bit) dst[d >> 3] |= (uint8_t)(1u << (d & 7));
    else dst[d >> 3] &= (uint8_t)~(1u << (d & 7));
  }
}
void bit_fill_ones(uint8_t* dst, int64_t dst_off, int64_t n) {
  for (int64_t i = 0; i < n;) {
    int64_t d = dst_off + i;
    if ((d & 7) == 0 && n - i >= 8) {
      int64_t full = (n - i) >> 3;
      memset(dst + (d >> 3), 0xff, (size_t)full);
      i += full * 8;
    } else {
      dst[d >> 3] |= (uint8_t)(1u << (d & 7));
      i++;
    }
  }
}

bool format_matches(const char* fmt, const DType& t) {
  if (!fmt) return false;
  std::string f = fmt;
  if (t.id == TypeId::Timestamp) return f.rfind("tsu:", 0) == 0 && f.size() > 4;
  if (t.id == TypeId::Decimal) {
    std::string e = expected_format(t);
    return f == e || f == e + ",128";
  }
  return f == expected_format(t);
}

// ---- ScanExec's cast of stream columns to the declared types (operators/scan.rs:281-291 → arrow::compute::cast_with_options with the
// default CastOptions: safe, i.e. a value the target cannot hold becomes NULL) — the numeric / temporal / decimal subset the JVM side can
// produce: integer widths and signedness, float widths, int ↔ float, Date64, timestamp units, decimal precision / scale, LargeUtf8 ----
struct SrcFmt {
  enum Cls { Unknown, Int, UInt, F32, F64, Date32, Date64, Ts, Dec, Utf8, LargeUtf8, Bool } cls = Unknown;
  int width = 0;
  int64_t per_second = 0;   // Ts: ticks per second
  int p = 0, s = 0;         // Dec
};
SrcFmt parse_src_format(const char* fmt) {
  SrcFmt f;
  if (!fmt) return f;
  const std::string x = fmt;
  auto intw = [&](char c) { return c == 'c' || c == 'C' ? 1 : c == 's' || c == 'S' ? 2 : c == 'i' || c == 'I' ? 4 : 8; };
  if (x.size() == 1 && strchr("csil", x[0])) { f.cls = SrcFmt::Int; f.width = intw(x[0]); }
  else if (x.size() == 1 && strchr("CSIL", x[0])) { f.cls = SrcFmt::UInt; f.width = intw(x[0]); }
  else if (x == "f") { f.cls = SrcFmt::F32; f.width = 4; }
  else if (x == "g") { f.cls = SrcFmt::F64; f.width = 8; }
  else if (x == "b") { f.cls = SrcFmt::Bool; }
  else if (x == "u" || x == "z") { f.cls = SrcFmt::Utf8; }
  else if (x == "U" || x == "Z") { f.cls = SrcFmt::LargeUtf8; }
  else if (x == "tdD") { f.cls = SrcFmt::Date32; f.width = 4; }
  else if (x == "tdm") { f.cls = SrcFmt::Date64; f.width = 8; }
  else if (x.rfind("ts", 0) == 0 && x.size() >= 4 && x[3] == ':') {
    f.cls = SrcFmt::Ts; f.width = 8;
    f.per_second = x[2] == 's' ? 1 : x[2] == 'm' ? 1000 : x[2] == 'u' ? 1000000 : x[2] == 'n' ? 1000000000 : 0;
    if (!f.per_second) f.cls = SrcFmt::Unknown;
  } else if (x.rfind("d:", 0) == 0) {
    int bits = 128;
    if (sscanf(x.c_str(), "d:%d,%d,%d", &f.p, &f.s, &bits) >= 2 && bits == 128) { f.cls = SrcFmt::Dec; f.width = 16; }
  }
  return f;
}
bool scan_cast_supported(const SrcFmt& f, const DType& t) {
  const bool tint = t.is_integer(), tflt = t.is_float();
  switch (f.cls) {
    case SrcFmt::Int: case SrcFmt::UInt: return tint || tflt || t.id == TypeId::Decimal || (f.cls == SrcFmt::Int && f.width == 4 && t.id == TypeId::Date) ||
                                                (f.cls == SrcFmt::Int && f.width == 8 && (t.id == TypeId::Timestamp || t.id == TypeId::TimestampNtz));
    case SrcFmt::F32: case SrcFmt::F64: return tint || tflt;
    case SrcFmt::Date32: return t.id == TypeId::Int32 || t.id == TypeId::Int64;
    case SrcFmt::Date64: return t.id == TypeId::Date || t.id == TypeId::Int64;
    case SrcFmt::Ts: return t.id == TypeId::Timestamp || t.id == TypeId::TimestampNtz || t.id == TypeId::Int64;
    case SrcFmt::Dec: return t.id == TypeId::Decimal || tint;
    case SrcFmt::LargeUtf8: return t.id == TypeId::String || t.id == TypeId::Bytes;
    default: return false;
  }
}
i128 cast_pow10(int e) { i128 r = 1; for (int i = 0; i < e; i++) r *= 10; return r; }
// one source value (row `i` of a column buffer) → the declared type at dst; false = NULL (safe cast)
bool scan_cast_value(const SrcFmt& f, const char* src, int64_t i, const DType& t, char* dst) {
  // read
  int64_t iv = 0; uint64_t uv = 0; double dv = 0; i128 xv = 0;
  enum { I, U, D, X } k = I;
  switch (f.cls) {
    case SrcFmt::Int: case SrcFmt::Date32: case SrcFmt::Date64: case SrcFmt::Ts:
      switch (f.width) { case 1: iv = ((const int8_t*)src)[i]; break; case 2: iv = ((const int16_t*)src)[i]; break; case 4: { int32_t v; memcpy(&v, src + i * 4, 4); iv = v; break; }
                         default: memcpy(&iv, src + i * 8, 8); }
      break;
    case SrcFmt::UInt:
      switch (f.width) { case 1: uv = ((const uint8_t*)src)[i]; break; case 2: { uint16_t v; memcpy(&v, src + i * 2, 2); uv = v; break; } case 4: { uint32_t v; memcpy(&v, src + i * 4, 4); uv = v; break; }
                         default: memcpy(&uv, src + i * 8, 8); }
      k = U;
      break;
    case SrcFmt::F32: { float v; memcpy(&v, src + i * 4, 4); dv = v; k = D; break; }
    case SrcFmt::F64: memcpy(&dv, src + i * 8, 8); k = D; break;
    case SrcFmt::Dec: memcpy(&xv, src + i * 16, 16); k = X; break;
    default: return false;
  }
  // temporal rescaling first (integers)
  if (f.cls == SrcFmt::Date64 && t.id == TypeId::Date) iv = iv / 86400000;            // arrow: ms / MILLISECONDS_IN_DAY (truncating)
  if (f.cls == SrcFmt::Ts && (t.id == TypeId::Timestamp || t.id == TypeId::TimestampNtz) && f.per_second != 1000000) {
    if (f.per_second > 1000000) iv = iv / (f.per_second / 1000000);                     // finer → µs: truncating division (arrow unary `/`)
    else if (__builtin_mul_overflow(iv, (int64_t)(1000000 / f.per_second), &iv)) return false;   // coarser → µs: checked multiply
  }
  auto store_int = [&](i128 v) -> bool {   // range-checked narrowing (num::cast): out of range → NULL
    switch (t.id) {
      case TypeId::Int8: if (v < -128 || v > 127) return false; { int8_t o = (int8_t)v; memcpy(dst, &o, 1); } return true;
      case TypeId::Int16: if (v < -32768 || v > 32767) return false; { int16_t o = (int16_t)v; memcpy(dst, &o, 2); } return true;
      case TypeId::Int32: case TypeId::Date: if (v < INT32_MIN || v > INT32_MAX) return false; { int32_t o = (int32_t)v; memcpy(dst, &o, 4); } return true;
      default: if (v < (i128)INT64_MIN || v > (i128)INT64_MAX) return false; { int64_t o = (int64_t)v; memcpy(dst, &o, 8); } return true;
    }
  };
  if (t.is_integer() || t.id == TypeId::Date || t.id == TypeId::Timestamp || t.id == TypeId::TimestampNtz) {
    if (k == I) return store_int(iv);
    if (k == U) return store_int((i128)uv);
    if (k == D) {                          // float → int: truncate toward zero; NaN / out of range → NULL
      if (!(dv == dv)) return false;
      const double tr = dv < 0 ? ceil(dv) : floor(dv);
      if (tr < -9223372036854775808.0 || tr >= 9223372036854775808.0) return false;
      return store_int((i128)(int64_t)tr);
    }
    // decimal → int: unscaled / 10^scale (truncating), then the range check
    return store_int(xv / cast_pow10(f.s));
  }
  if (t.id == TypeId::Float || t.id == TypeId::Double) {
    const double v = k == I ? (double)iv : k == U ? (double)uv : dv;
    if (t.id == TypeId::Float) { float o = k == I ? (float)iv : k == U ? (float)uv : (float)dv; memcpy(dst, &o, 4); }
    else memcpy(dst, &v, 8);
    return true;
  }
  if (t.id == TypeId::Decimal) {
    i128 v;
    const i128 bound = cast_pow10(t.precision) - 1;
    if (k == X) {
      const int up = t.scale - f.s;
      if (up >= 0) { if (__builtin_mul_overflow(xv, cast_pow10(up), &v)) return false; }
      else {
        const i128 div = cast_pow10(-up), half = div / 2, d = xv / div, r = xv % div;   // round half away from zero
        v = xv >= 0 ? (r >= half ? d + 1 : d) : (r <= -half ? d - 1 : d);
      }
    } else if (k == I || k == U) {
      if (__builtin_mul_overflow(k == I ? (i128)iv : (i128)uv, cast_pow10(t.scale), &v)) return false;
    } else return false;
    if (v > bound || v < -bound) return false;
    memcpy(dst, &v, 16);
    return true;
  }
  return false;
}

const Operator* find_scan(const Operator* op) {
  while (op && op->kind != OpKind::Scan) {
    if (op->children.empty()) return nullptr;
    op = op->children[0].get();
  }
  return op;
}

std::string validity_key(const std::vector<bool>& v) {
  std::string k;
  for (bool b : v) k.push_back(b ? '1' : '0');
  return k;
}

struct Timer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ns() const { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

}  // namespace

// Planned pipelines are shared by every task that runs the same plan bytes (a Spark stage = thousands of
// identical createPlan calls): (plan hash, validity pattern) → generated source + compiled code object.
namespace {
struct PlannedVariant {
  PipelineDesc desc;
  std::shared_ptr<CodeObject> code;
};
std::mutex g_plan_mu;
std::map<std::string, std::shared_ptr<PlannedVariant>> g_plan_cache;
}  // namespace

static std::shared_ptr<PlannedVariant> planned_variant(const Operator& plan, uint64_t plan_hash, const std::vector<bool>& has_valid,
                                                       bool compile, const std::vector<DType>* source_types = nullptr,
                                                       const std::vector<int>* str_fixed_len = nullptr,
                                                       const std::vector<int>* dict_id_col = nullptr) {
  std::string key = std::to_string(plan_hash) + ":" + validity_key(has_valid);
  if (dict_id_col) {
    key += ":D";
    for (int c : *dict_id_col) key += std::to_string(c) + ",";
  }
  if (str_fixed_len) {
    key += ":L";
    for (int l : *str_fixed_len) key += std::to_string(l) + ",";
  }
  if (source_types) {
    key += ":";
    for (auto& t : *source_types) key += t.str() + ",";
  }
  std::shared_ptr<PlannedVariant> pv;
  {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    if (it != g_plan_cache.end()) pv = it->second;
  }
  if (!pv) {
    pv = std::make_shared<PlannedVariant>();
    pv->desc = generate_pipeline(plan, has_valid, source_types, str_fixed_len, dict_id_col);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto res = g_plan_cache.emplace(key, pv);
    pv = res.first->second;
  }
  if (compile && !pv->code) {
    auto co = jit_compile(pv->desc.source);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (!pv->code) pv->code = co;
  }
  return pv;
}

// ---------------------------------------------------------------------------------------------
ExecutionContext::ExecutionContext(OperatorP plan, uint64_t plan_hash, std::vector<std::pair<std::string, std::string>> config,
                                   std::vector<InputSource> inputs, int batch_size, int device_id)
    : plan_(std::move(plan)), plan_hash_(plan_hash), config_(std::move(config)), inputs_(std::move(inputs)), batch_size_(batch_size),
      device_id_(device_id) {
  chunk_rows_ = 4 << 20;
  mem_->owner = std::this_thread::get_id();
  for (auto& kv : config_) {
    if (kv.first == "spark.comet.gpu.chunkRows") chunk_rows_ = std::max<long long>(1024, atoll(kv.second.c_str()));
    if (kv.first == "spark.comet.gpu.memory.limit") mem_->dev_limit = std::max<long long>(0, atoll(kv.second.c_str()));
  }
  if (const char* e = getenv("COMET_GPU_CHUNK_ROWS")) chunk_rows_ = std::max<long long>(1024, atoll(e));
  // Scan leaves in depth-first, left-before-right order map to the input streams (planner.rs:1726, :2391)
  std::function<void(const Operator&)> walk = [&](const Operator& op) {
    node_id_[&op] = (int)node_id_.size();
    if (op.kind == OpKind::Scan) scan_input_[&op] = scan_input_.size();
    if (op.kind == OpKind::HashJoin || op.kind == OpKind::NativeScan || op.kind == OpKind::Sort || op.kind == OpKind::Limit || op.kind == OpKind::ShuffleWriter ||
        op.kind == OpKind::Expand || op.kind == OpKind::Window)
      has_join_ = true;
    if (op.kind == OpKind::Window) {
      for (int t = 0; t < 2; t++) {
        auto so = std::make_shared<Operator>();
        so->kind = OpKind::Sort;
        so->proto_tag = 103;
        (t == 0 ? window_psort_ : window_osort_)[&op] = so;
        node_id_[so.get()] = (int)node_id_.size();
      }
    }   // sources materialised in HBM
    if (op.kind == OpKind::ShuffleWriter) {
      if (&op != plan_.get()) throw CometError("ShuffleWriter must be the root of a native plan");
      bool computed = false;
      for (auto& e : op.shuffle_hash_exprs) computed |= e->kind != ExprKind::Bound;
      for (auto& k : op.shuffle_sort_orders) computed |= k.child->kind != ExprKind::Bound;
      if (op.shuffle_partitioning == Operator::Partitioning::Range) {
        // range partitioning compares order-preserving key bytes: one synthetic Sort describes the rows' keys, one the boundaries'
        for (int t = 0; t < 2; t++) {
          auto so = std::make_shared<Operator>();
          so->kind = OpKind::Sort;
          so->proto_tag = 103;
          (t == 0 ? range_sort_ : range_bsort_)[&op] = so;
          node_id_[so.get()] = (int)node_id_.size();
        }
      }
      if (computed) {
        // hash expressions that are not plain column references: evaluated by a Projection(child columns ++ expressions) fused
        // over the child (its project_list is filled in when the child's schema is known)
        auto pr = std::make_shared<Operator>();
        pr->kind = OpKind::Projection;
        pr->proto_tag = 101;
        pr->children = op.children;
        shuffle_projs_[&op] = pr;
        node_id_[pr.get()] = (int)node_id_.size();
      }
    }
    if (op.kind == OpKind::HashAgg && &op != plan_.get()) {
      // an aggregate below other operators: materialised too, unless it is the sink of the root chain (checked below)
      nested_aggs_.push_back(&op);
    }
    if (op.kind == OpKind::Unsupported)
      throw CometError(std::string("Operator ") + op_name(op.proto_tag) + " is not supported by the MI355X native engine");
    if (op.kind == OpKind::HashJoin && op.smj) {
      // synthetic Sort over the join output: left keys keep their column indices in left ++ right; a RightOuter join is ordered
      // by the right keys (shifted past the left columns, resolved when the schema is known)
      auto so = std::make_shared<Operator>();
      so->kind = OpKind::Sort;
      so->proto_tag = 103;
      smj_sorts_[&op] = so;
      node_id_[so.get()] = (int)node_id_.size();
    }
    for (auto& c : op.children) walk(*c);
  };
  walk(*plan_);
  // Which sort-merge joins must really deliver their output in key order?  Only those whose row order can reach something that
  // looks at it: the plan's output, a Limit, a shuffle file.  Aggregates, sorts and every join here (sort-merge joins run as hash
  // joins and do not need sorted inputs) forget the order of their inputs, so the sort below them would be wasted work
  // (TPC-DS Q95: five sorts of up to 68 M rows).
  std::function<void(const Operator&, bool)> mark = [&](const Operator& op, bool order_visible) {
    if (op.kind == OpKind::HashJoin && op.smj && order_visible) smj_needs_sort_.insert(&op);
    for (size_t i = 0; i < op.children.size(); i++) {
      bool v = order_visible;
      switch (op.kind) {
        case OpKind::HashAgg: case OpKind::Sort: v = false; break;
        case OpKind::HashJoin:
          // a plain hash join streams its probe side: the probe order shows in the output; a sort-merge join re-sorts (or not)
          v = !op.smj && order_visible && (op.build_side == BuildSide::Left ? i == 1 : i == 0);
          break;
        default: break;   // Projection / Filter / Limit / ShuffleWriter keep their input's order
      }
      mark(*op.children[i], v);
    }
  };
  mark(*plan_, true);
  if (scan_input_.empty() && !has_join_) throw CometError("Plan has no Scan leaf: only Scan-rooted pipelines are supported by the MI355X native engine");
  if (inputs_.size() != scan_input_.size())
    throw CometError("Plan has " + std::to_string(scan_input_.size()) + " Scan leaves but " + std::to_string(inputs_.size()) + " input streams were given");
  // the root chain ends at a Scan or at the first join below it
  root_source_ = plan_.get();
  while (!is_source(*root_source_, plan_.get())) {
    if (root_source_->children.size() != 1) throw CometError(std::string(op_name(root_source_->proto_tag)) + " expects exactly one child");
    root_source_ = root_source_->children[0].get();
  }
  if (root_source_->kind == OpKind::HashAgg) has_join_ = true;   // operators above an aggregate: the aggregate is materialised
  // Validate the plan shape eagerly (generated, not compiled) so that unsupported operators fail at createPlan
  // like the reference's planner would on first execute.
  if (!has_join_) {
    in_types_ = root_source_->scan_fields;
    std::vector<bool> none(in_types_.size(), false);
    auto pv = planned_variant(*plan_, plan_hash_, none, false);
    explain_ = pv->desc.explain;
    sink_ = pv->desc.sink;
    if (sink_ == SinkKind::Output)
      for (auto& oc : pv->desc.out_cols) materialize_root_ |= oc.gather_src >= 0 || oc.packed_string || oc.view_src >= 0;   // Utf8 outputs are finished on the device (gather / unpack)
    // a grouped aggregate keyed by Utf8 columns sees its whole input at once (like a join input): only then can strings longer
    // than the packed 15 bytes be swapped for representative row indices (prepare_dict_keys)
    if (sink_ == SinkKind::AggGrouped && !pv->desc.str_key_cols.empty()) has_join_ = true;
  } else {
    in_types_ = infer_schema(*root_source_);
    if (plan_.get() != root_source_) {
      std::vector<bool> none(in_types_.size(), false);
      auto pv = planned_variant(*plan_, plan_hash_, none, false, &in_types_);
      explain_ = explain_ + pv->desc.explain;
      sink_ = pv->desc.sink;
    } else {
      sink_ = SinkKind::Output;
    }
  }
}

// Output schema of a sub-plan (no data needed): Scan fields, chain outputs, join = left ++ right (semi/anti: left)
// Where a fused chain stops: Scan leaves and everything whose result is materialised in HBM — joins, Parquet scans, sorts,
// limits, and an aggregate that is not the top of the chain being fused.
bool ExecutionContext::is_source(const Operator& op, const Operator* chain_top) {
  switch (op.kind) {
    case OpKind::Scan: case OpKind::HashJoin: case OpKind::NativeScan: case OpKind::Sort: case OpKind::Limit: case OpKind::ShuffleWriter: case OpKind::Expand: case OpKind::Window: return true;
    case OpKind::HashAgg: return &op != chain_top;
    default: return false;
  }
}

std::vector<DType> ExecutionContext::infer_schema(const Operator& op) {
  if (op.kind == OpKind::Scan) return op.scan_fields;
  if (op.kind == OpKind::Sort || op.kind == OpKind::Limit) {
    if (op.children.size() != 1) throw CometError(std::string(op_name(op.proto_tag)) + " expects exactly one child");
    std::vector<DType> st = infer_schema(*op.children[0]);
    if (op.kind == OpKind::Sort) {
      std::vector<bool> none(st.size(), false);
      PipelineDesc d = generate_sort_keys(op, st, none);   // validates the sort expressions
      if (compile_in_infer_) jit_compile(d.source);
      explain_ += d.explain;
    } else {
      if (op.limit != -1 && op.offset > op.limit)
        throw CometError("Invalid limit/offset combination: [" + std::to_string(op.limit) + ". " + std::to_string(op.offset) + "]");
      explain_ += "  limit " + std::to_string(op.limit) + " offset " + std::to_string(op.offset) + "\n";
    }
    return st;
  }
  if (op.kind == OpKind::Window) {
    // WindowAggExec / BoundedWindowAggExec (planner.rs:2267-2379): child columns ++ one column per window expression
    if (op.children.size() != 1) throw CometError("Window expects exactly one child");
    std::vector<DType> st = infer_schema(*op.children[0]);
    Operator& ps = *window_psort_.at(&op);
    Operator& os = *window_osort_.at(&op);
    ps.sort_orders.clear();
    for (auto& e : op.window_partition) {
      Operator::SortKey k;
      k.child = e;
      ps.sort_orders.push_back(k);
    }
    os.sort_orders = op.window_order;
    std::vector<bool> none(st.size(), false);
    if (!ps.sort_orders.empty()) { PipelineDesc d = generate_sort_keys(ps, st, none); if (compile_in_infer_) jit_compile(d.source); }
    if (!os.sort_orders.empty()) { PipelineDesc d = generate_sort_keys(os, st, none); if (compile_in_infer_) jit_compile(d.source); }
    std::vector<DType> out = st;
    for (auto& fn : op.window_fns) {
      if (fn.is_agg) {
        // SUM / COUNT / AVG of exact types over a frame that starts at the partition start (whole partition, or up to the current row /
        // peer group) — the frames the reference runs with its own Spark-exact accumulators (planner.rs:2953-2972)
        const AggExpr& a = fn.agg;
        // frames: every combination of UNBOUNDED / CURRENT ROW bounds, and ROWS frames with literal offsets (n PRECEDING / n FOLLOWING);
        // RANGE frames with value offsets need a search over the order key and are not supported
        if (fn.frame_range_literal || (!fn.frame_rows && (fn.frame_lower == 1 || fn.frame_upper == 1)))
          throw CometError("Window: RANGE frames with a value offset (RANGE BETWEEN x PRECEDING …) are not supported yet");
        const bool minmax = a.kind == AggKind::Min || a.kind == AggKind::Max;
        if (minmax && fn.frame_lower == 1 && fn.frame_upper == 1 && fn.frame_upper_off - fn.frame_lower_off > 4096)
          throw CometError("Window: MIN / MAX over a sliding frame wider than 4096 rows is not supported yet");
        if (a.children.size() != 1) throw CometError("Window: aggregate window functions take one argument");
        const ExprP& arg = a.children[0];
        const bool lit = arg->kind == ExprKind::Literal;
        if (!lit && (arg->kind != ExprKind::Bound || arg->bound_index < 0 || (size_t)arg->bound_index >= st.size()))
          throw CometError("Window: the argument of an aggregate window function must be a column (or a literal for COUNT)");
        const DType at = lit ? arg->dtype : st[(size_t)arg->bound_index];
        if (a.kind == AggKind::Count) out.push_back(DType::of(TypeId::Int64));
        else if (lit) throw CometError("Window: SUM / AVG / MIN / MAX of a literal is not supported");
        else if (minmax && (at.is_integer() || at.id == TypeId::Decimal || at.id == TypeId::Date || at.id == TypeId::Timestamp || at.id == TypeId::TimestampNtz)) out.push_back(at);
        else if (a.kind == AggKind::Sum && at.id == TypeId::Decimal && a.dtype.id == TypeId::Decimal) out.push_back(a.dtype);
        else if (a.kind == AggKind::Sum && at.is_integer()) out.push_back(DType::of(TypeId::Int64));
        else if (a.kind == AggKind::Avg && at.id == TypeId::Decimal && a.dtype.id == TypeId::Decimal) out.push_back(a.dtype);
        else throw CometError("Window: aggregate (tag " + std::to_string(a.proto_tag) + ") over " + at.str() + " is not supported yet (SUM / AVG of decimals, SUM of integers, COUNT, MIN / MAX of integers, decimals, dates and timestamps are)");
        continue;
      }
      const std::string& f = fn.func;
      auto int_lit = [](const ExprP& x) { return x->kind == ExprKind::Literal && !x->lit_null && x->dtype.is_integer(); };
      if (f == "row_number" || f == "rank" || f == "dense_rank") out.push_back(DType::of(TypeId::Int32));
      else if (f == "percent_rank" || f == "cume_dist") out.push_back(DType::of(TypeId::Double));
      else if (f == "ntile") {
        if (fn.args.size() != 1 || !int_lit(fn.args[0]) || fn.args[0]->lit_i64 <= 0) throw CometError("ntile expects a positive literal bucket count");
        out.push_back(DType::of(TypeId::Int32));
      } else if (f == "lag" || f == "lead") {
        if (fn.args.size() < 1 || fn.args.size() > 3 || fn.args[0]->kind != ExprKind::Bound || fn.args[0]->bound_index < 0 || (size_t)fn.args[0]->bound_index >= st.size())
          throw CometError(f + " is supported for a column argument");
        if (fn.args.size() >= 2 && !int_lit(fn.args[1])) throw CometError(f + " expects a literal offset");
        if (fn.args.size() == 3 && fn.args[2]->kind != ExprKind::Literal) throw CometError(f + " default value must be a literal");
        if (fn.args.size() == 3 && !fn.args[2]->lit_null) {
          const DType& at = st[(size_t)fn.args[0]->bound_index];
          if (at.id == TypeId::String || at.id == TypeId::Bytes || at.id == TypeId::Bool) throw CometError(f + " with a non-NULL default value over " + at.str() + " is not supported yet");
        }
        if (fn.ignore_nulls) throw CometError(f + " IGNORE NULLS is not supported yet");
        out.push_back(st[(size_t)fn.args[0]->bound_index]);
      } else {
        throw CometError(f + " not supported for window function");
      }
    }
    explain_ += "  window: " + std::to_string(op.window_fns.size()) + " function(s), " + std::to_string(op.window_partition.size()) + " partition key(s), " +
                std::to_string(op.window_order.size()) + " order key(s)\n";
    return out;
  }
  if (op.kind == OpKind::Expand) {
    // ExpandExec (operators/expand.rs; planner.rs:1913-1948): every input row yields one output row per projection (grouping sets /
    // rollup / cube, the count(DISTINCT …) rewrite over several columns).  Each projection is a fused Projection over the resident child;
    // all of them write into ONE set of output buffers at their row offset.  NULL literals of Utf8 (or unknown) type — the "not in this
    // grouping set" marker — are not generated code: their rows are simply marked invalid.
    if (op.children.size() != 1) throw CometError("Expand expects exactly one child");
    std::vector<DType> st = infer_schema(*op.children[0]);
    const size_t ncol = op.expand_projections.empty() ? 0 : op.expand_projections[0].size();
    ExpandInfo info;
    std::vector<std::vector<DType>> types(op.expand_projections.size(), std::vector<DType>(ncol));
    std::vector<std::vector<bool>> known(op.expand_projections.size(), std::vector<bool>(ncol, false));
    std::vector<std::vector<int>> gsrc(op.expand_projections.size(), std::vector<int>(ncol, -1));
    auto is_null_lit = [](const ExprP& e) { return e->kind == ExprKind::Literal && e->lit_null; };
    auto scan_of = [&]() {
      auto sc = std::make_shared<Operator>();
      sc->kind = OpKind::Scan;
      sc->proto_tag = 100;
      sc->scan_fields = st;
      return sc;
    };
    // pass 1: types of everything that is not a NULL literal
    for (size_t p = 0; p < op.expand_projections.size(); p++) {
      auto pr = std::make_shared<Operator>();
      pr->kind = OpKind::Projection;
      pr->proto_tag = 101;
      pr->children.push_back(scan_of());
      std::vector<size_t> at;
      for (size_t j = 0; j < ncol; j++)
        if (!is_null_lit(op.expand_projections[p][j])) { pr->project_list.push_back(op.expand_projections[p][j]); at.push_back(j); }
      if (pr->project_list.empty()) continue;
      std::vector<bool> none(st.size(), false);
      PipelineDesc d = generate_pipeline(*pr, none, &st);
      for (size_t k = 0; k < at.size(); k++) {
        types[p][at[k]] = d.out_cols[k].type;
        known[p][at[k]] = true;
        if (d.out_cols[k].view_src >= 0) throw CometError("Expand: string functions with results of any length are not supported inside a grouping-set projection yet");
        gsrc[p][at[k]] = d.out_cols[k].packed_string ? -2 : d.out_cols[k].gather_src;   // −2: a computed (packed) string
      }
    }
    for (size_t j = 0; j < ncol; j++) {
      OutCol oc;
      bool have = false;
      for (size_t p = 0; p < op.expand_projections.size(); p++) {
        if (!known[p][j]) continue;
        if (!have) { oc.type = types[p][j]; oc.gather_src = gsrc[p][j] == -2 ? -1 : gsrc[p][j]; oc.packed_string = gsrc[p][j] == -2; have = true; }
        else if (types[p][j] != oc.type || gsrc[p][j] != (oc.packed_string ? -2 : oc.gather_src))
          throw CometError("Expand: column " + std::to_string(j) + " differs between the projections (" + oc.type.str() + " vs " + types[p][j].str() + ")");
      }
      if (!have) {
        const DType& lt = op.expand_projections[0][j]->dtype;
        if (lt.id == TypeId::Null || lt.id == TypeId::Unknown) throw CometError("Expand: column " + std::to_string(j) + " is NULL in every projection and carries no type");
        if (lt.id == TypeId::String || lt.id == TypeId::Bytes) throw CometError("Expand: a Utf8 column that is NULL in every projection is not supported yet");
        oc.type = lt;
      }
      oc.nullable = true;
      info.out_cols.push_back(oc);
    }
    // pass 2: the projections as executed
    for (size_t p = 0; p < op.expand_projections.size(); p++) {
      ExpandPart part;
      part.proj = std::make_shared<Operator>();
      part.proj->kind = OpKind::Projection;
      part.proj->proto_tag = 101;
      part.proj->children.push_back(scan_of());
      for (size_t j = 0; j < ncol; j++) {
        const ExprP& e = op.expand_projections[p][j];
        const DType& ut = info.out_cols[j].type;
        if (is_null_lit(e) && (ut.id == TypeId::String || ut.id == TypeId::Bytes)) { part.null_cols.push_back((int)j); continue; }
        if (is_null_lit(e)) {
          auto typed = std::make_shared<Expr>(*e);
          typed->dtype = ut;
          typed->has_dtype = true;
          part.proj->project_list.push_back(typed);
        } else {
          part.proj->project_list.push_back(e);
        }
        part.out_col.push_back((int)j);
      }
      node_id_[part.proj.get()] = (int)node_id_.size();
      if (!part.proj->project_list.empty()) {
        std::vector<bool> none(st.size(), false);
        PipelineDesc d = generate_pipeline(*part.proj, none, &st);
        if (compile_in_infer_) jit_compile(d.source);
      }
      info.parts.push_back(part);
    }
    explain_ += "  expand: " + std::to_string(info.parts.size()) + " projection(s) of " + std::to_string(ncol) + " column(s)\n";
    std::vector<DType> out;
    for (auto& oc : info.out_cols) out.push_back(oc.type);
    expand_info_[&op] = info;
    return out;
  }
  if (op.kind == OpKind::ShuffleWriter) {
    // ShuffleWriterExec (shuffle_writer.rs:60-110): consumes its child, writes the data + index files, yields no batches
    if (op.children.size() != 1) throw CometError("ShuffleWriter expects exactly one child");
    std::vector<DType> st = infer_schema(*op.children[0]);
    if (op.shuffle_num_partitions < 1) throw CometError("ShuffleWriter: num_partitions must be positive");
    if (op.shuffle_num_partitions > 4096) throw CometError("ShuffleWriter: more than 4096 output partitions are not supported yet");
    if (op.shuffle_codec < 0 || op.shuffle_codec > 3) throw CometError("Unsupported shuffle compression codec: " + std::to_string(op.shuffle_codec));
    if (op.shuffle_data_file.empty() || op.shuffle_index_file.empty()) throw CometError("ShuffleWriter: output_data_file / output_index_file missing");
    for (auto& e : op.shuffle_hash_exprs)
      if (e->kind == ExprKind::Bound && (e->bound_index < 0 || (size_t)e->bound_index >= st.size()))
        throw CometError("ShuffleWriter: hash expression references column " + std::to_string(e->bound_index) + " of " + std::to_string(st.size()));
    for (auto& t : st)
      if (expected_format(t) == "?") throw CometError("ShuffleWriter: column type " + t.str() + " is not supported");
    auto sp = shuffle_projs_.find(&op);
    if (sp != shuffle_projs_.end()) {
      Operator& pr = *sp->second;
      pr.project_list.clear();
      for (size_t i = 0; i < st.size(); i++) {
        auto b = std::make_shared<Expr>();
        b->kind = ExprKind::Bound;
        b->proto_tag = 3;
        b->bound_index = (int)i;
        b->dtype = st[i];
        b->has_dtype = true;
        pr.project_list.push_back(b);
      }
      for (auto& e : op.shuffle_hash_exprs)
        if (e->kind != ExprKind::Bound) pr.project_list.push_back(e);
      for (auto& k : op.shuffle_sort_orders)
        if (k.child->kind != ExprKind::Bound) pr.project_list.push_back(k.child);
      st = infer_schema(pr);   // validates (and, under compile_only, compiles) the fused chain; st now includes the computed key columns
    }
    if (op.shuffle_partitioning == Operator::Partitioning::Range) {
      // RangePartition (partitioning.proto:52-56; planner.rs:3298-3358): rows are compared with the boundary rows under sort_orders
      if (op.shuffle_sort_orders.empty()) throw CometError("ShuffleWriter: range partitioning without sort orders");
      for (auto& row : op.shuffle_bounds) {
        if (row.size() != op.shuffle_sort_orders.size()) throw CometError("ShuffleWriter: a range boundary row has " + std::to_string(row.size()) + " values for " +
                                                                           std::to_string(op.shuffle_sort_orders.size()) + " sort orders");
        for (auto& e : row)
          if (e->kind != ExprKind::Literal) throw CometError("ShuffleWriter: range boundaries must be literals");
      }
      if ((int)op.shuffle_bounds.size() + 1 > op.shuffle_num_partitions)
        throw CometError("ShuffleWriter: " + std::to_string(op.shuffle_bounds.size()) + " range boundaries need more than " + std::to_string(op.shuffle_num_partitions) + " partitions");
      Operator& so = *range_sort_.at(&op);
      Operator& sb = *range_bsort_.at(&op);
      so.sort_orders.clear();
      sb.sort_orders.clear();
      std::vector<DType> btypes;
      size_t next = st.size();
      for (auto& k : op.shuffle_sort_orders) next -= k.child->kind != ExprKind::Bound;
      size_t computed_at = next;
      for (size_t i = 0; i < op.shuffle_sort_orders.size(); i++) {
        const Operator::SortKey& k = op.shuffle_sort_orders[i];
        const int col = k.child->kind == ExprKind::Bound ? k.child->bound_index : (int)computed_at++;
        if (col < 0 || (size_t)col >= st.size()) throw CometError("ShuffleWriter: range sort order references column " + std::to_string(col));
        auto mk = [&](int idx, const DType& t) {
          auto b = std::make_shared<Expr>();
          b->kind = ExprKind::Bound;
          b->proto_tag = 3;
          b->bound_index = idx;
          b->dtype = t;
          b->has_dtype = true;
          return b;
        };
        Operator::SortKey a = k, b = k;
        a.child = mk(col, st[(size_t)col]);
        b.child = mk((int)i, st[(size_t)col]);
        so.sort_orders.push_back(a);
        sb.sort_orders.push_back(b);
        btypes.push_back(st[(size_t)col]);
      }
      std::vector<bool> none(st.size(), false), bnone(btypes.size(), true);
      PipelineDesc d1 = generate_sort_keys(so, st, none), d2 = generate_sort_keys(sb, btypes, bnone);
      if (d1.sort_key_bytes != d2.sort_key_bytes) throw CometError("internal: range boundary keys and row keys differ in width");
      if (compile_in_infer_) { jit_compile(d1.source); jit_compile(d2.source); }
    }
    explain_ += "  shuffle writer: " + std::to_string(op.shuffle_num_partitions) + " partition(s), codec " + std::to_string(op.shuffle_codec) + "\n";
    return {};
  }
  if (op.kind == OpKind::NativeScan) {
    std::vector<DType> out;
    for (auto& f : op.required_schema) out.push_back(f.dtype);
    for (auto& f : op.partition_schema) out.push_back(f.dtype);   // Hive partition columns follow the file columns
    explain_ += "  parquet scan: " + std::to_string(op.files.size()) + " file(s), " + std::to_string(out.size()) + " column(s)\n";
    return out;
  }
  if (op.kind == OpKind::HashJoin) {
    if (op.children.size() != 2) throw CometError("HashJoin expects two children");
    std::vector<DType> l = infer_schema(*op.children[0]), r = infer_schema(*op.children[1]);
    std::vector<bool> lv(l.size(), false), rv(r.size(), false);
    PipelineDesc d = generate_join(op, l, r, lv, rv);   // validates keys / join type
    if (compile_in_infer_) jit_compile(d.source);
    explain_ += d.explain;
    if (op.smj) {
      Operator& so = *smj_sorts_.at(&op);
      so.sort_orders.clear();
      const bool by_right = op.join_type == JoinType::RightOuter;
      const std::vector<ExprP>& keys = by_right ? op.right_keys : op.left_keys;
      std::function<ExprP(const ExprP&)> shift = [&](const ExprP& e) -> ExprP {
        auto n = std::make_shared<Expr>(*e);
        if (e->kind == ExprKind::Bound) n->bound_index = e->bound_index + (int)l.size();
        for (auto& c : n->children) c = shift(c);
        return n;
      };
      for (size_t k = 0; k < keys.size(); k++) {
        Operator::SortKey sk;
        sk.child = by_right ? shift(keys[k]) : keys[k];
        if (k < op.smj_sort_options.size()) { sk.descending = op.smj_sort_options[k].first; sk.nulls_last = op.smj_sort_options[k].second; }
        so.sort_orders.push_back(sk);
      }
      explain_ += "  (sort-merge join: output ordered by the join keys)\n";
    }
    std::vector<DType> out;
    for (auto& c : d.out_cols) out.push_back(c.type);
    return out;
  }
  const Operator* src = &op;
  do {
    if (src->children.size() != 1) throw CometError(std::string(op_name(src->proto_tag)) + " expects exactly one child");
    src = src->children[0].get();
  } while (!is_source(*src, &op));
  std::vector<DType> st = infer_schema(*src);
  std::vector<bool> none(st.size(), false);
  PipelineDesc d = generate_pipeline(op, none, &st);
  if (compile_in_infer_) jit_compile(d.source);
  explain_ += d.explain;
  std::vector<DType> out;
  for (auto& c : d.out_cols) out.push_back(c.type);
  return out;
}

ExecutionContext::~ExecutionContext() {
  mem_->owner = std::this_thread::get_id();   // the buffers die on this thread (after this body): their bytes go back to the manager from here
  // Dropping the context releases the input streams back to their producer (scan.rs:41-44).
  for (auto& in : inputs_) {
    if (in.host && in.host->release) in.host->release(in.host);
    if (in.dev && in.dev->release) in.dev->release(in.dev);
  }
  for (auto& st : staging_)
    if (st && st->busy) {
      (void)hipEventSynchronize(st->busy);
      pool_put_event(device_id_, st->busy);
      st->busy = nullptr;
    }
  if (stream_) {
    (void)hipStreamSynchronize(stream_);  // pooled buffers go back only once the stream is idle
    for (auto& pr : timed_) { pool_put_event(device_id_, pr.first); pool_put_event(device_id_, pr.second); }
    pool_put_stream(device_id_, stream_);
  }
}

const std::string& ExecutionContext::explain() { return explain_; }

std::string ExecutionContext::compile_only(OperatorP plan, uint64_t plan_hash) {
  // count Scan leaves to fabricate the (never used) input list
  size_t nscan = 0;
  std::function<void(const Operator&)> cnt = [&](const Operator& op) {
    if (op.kind == OpKind::Scan) nscan++;
    for (auto& c : op.children) cnt(*c);
  };
  cnt(*plan);
  std::vector<InputSource> ins(nscan);
  ExecutionContext ctx(plan, plan_hash, {}, ins, 8192, 0);
  std::vector<bool> none(ctx.in_types_.size(), false);
  if (ctx.has_join_) {
    ctx.compile_in_infer_ = true;
    ctx.explain_.clear();
    ctx.infer_schema(*ctx.root_source_);
    if (ctx.plan_.get() != ctx.root_source_) {
      auto pv = planned_variant(*ctx.plan_, ctx.plan_hash_, none, true, &ctx.in_types_);
      ctx.explain_ += pv->desc.explain;
    }
  } else {
    planned_variant(*ctx.plan_, ctx.plan_hash_, none, true);
  }
  return ctx.explain_;
}

Variant& ExecutionContext::variant_for(const std::vector<bool>& has_valid, const std::vector<int>& str_fixed_len) {
  std::string key = validity_key(has_valid) + ":";
  bool any_fixed = false;
  for (int l : str_fixed_len) { key += std::to_string(l) + ","; any_fixed |= l >= 0; }
  auto it = variants_.find(key);
  if (it != variants_.end()) return it->second;
  auto pv = planned_variant(*plan_, plan_hash_, has_valid, true, has_join_ ? &in_types_ : nullptr, any_fixed ? &str_fixed_len : nullptr,
                            dict_id_col_.empty() ? nullptr : &dict_id_col_);
  Variant v;
  v.desc = pv->desc;
  v.mod = jit_load(pv->code);
  auto res = variants_.emplace(key, std::move(v));
  return res.first->second;
}

void ExecutionContext::launch(Variant& v, const char* kernel, int grid, CometKParams& prm) {
  hipFunction_t fn = v.mod->fn(kernel);
  void* args[] = {&prm};
  HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, stream_, args, nullptr));
}

// Utf8 group keys longer than the 15 bytes that fit the packed key words: replace them by representative row indices
// (strdict_kernels.hip).  `src` is the aggregate's complete, resident input; an Int64 index column is appended per long key column
// and the pipeline is generated with dict_id_col so that it groups on the index and emits it as the gather index of the string.
void ExecutionContext::prepare_dict_keys(DevTable& src) {
  if (src.rows == 0) return;
  std::vector<bool> none(in_types_.size(), false);
  auto pv = planned_variant(*plan_, plan_hash_, none, false, &in_types_);
  std::vector<int> key_cols = pv->desc.str_key_cols;
  std::sort(key_cols.begin(), key_cols.end());
  key_cols.erase(std::unique(key_cols.begin(), key_cols.end()), key_cols.end());
  if (key_cols.empty()) return;
  const int64_t n = src.rows;
  std::vector<int> id_col(src.cols.size(), -1);
  bool any = false;
  // Packed keys (≤ 15 bytes) are cheaper, but one long column — or a result that must stay on the device, where packed strings
  // cannot be expanded by the host — switches every Utf8 key column of the aggregate to row indices.
  bool need = device_result_;
  for (int c : key_cols) {
    const DeviceColumnView& sc = src.cols[(size_t)c];
    if (sc.offset != 0) return;     // sliced producer arrays keep the packed path (and its 15-byte limit)
    if (need) break;
    if (sc.fixed_len >= 0 && sc.fixed_len <= 15) continue;   // already known to hold values of one short length (TPC-H Q1's flag columns): nothing to measure
    uint32_t* mx = (uint32_t*)err_flags_.p + (kErrBytes / 4 - 1);   // last word of the error/aux block: scratch
    HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
    if (comet_launch_str_max_len((const int32_t*)sc.data, n, mx, stream_) != 0) throw CometError("string keys: launch failed");
    uint32_t longest = 0;
    read_small(&longest, mx, 4);
    HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
    need = longest > 15;
  }
  if (!need || src.cols.size() + key_cols.size() > COMET_MAX_IN) return;
  for (int c : key_cols) {
    const DeviceColumnView& sc = src.cols[(size_t)c];
    if (n >= ((int64_t)1 << 32) - 1) throw CometError("Utf8 group keys longer than 15 bytes over more than 2^32 rows are not supported");
    int64_t slots = 1024;
    while (slots < 2 * n) slots <<= 1;
    DevBuf table;   // only needed while the indices are computed
    table.ensure((size_t)slots * 4);
    HIP_CHECK(hipMemsetAsync(table.p, 0, (size_t)slots * 4, stream_));
    auto rep = std::make_shared<DevBuf>();
    rep->ensure((size_t)n * 8 + 16);
    if (comet_launch_str_dict_build((const int32_t*)sc.data, (const uint8_t*)sc.aux, src.has_valid[(size_t)c] ? sc.valid : nullptr, n, (uint32_t*)table.p, slots,
                                    (int64_t*)rep->p, stream_) != 0)
      throw CometError("string keys: launch failed");
    HIP_CHECK(hipStreamSynchronize(stream_));   // `table` goes back to the pool here
    DeviceColumnView idv;
    idv.data = rep->p;
    idv.valid = sc.valid;
    id_col[(size_t)c] = (int)src.cols.size();
    src.types.push_back(DType::of(TypeId::Int64));
    src.cols.push_back(idv);
    src.has_valid.push_back(src.has_valid[(size_t)c]);
    src.owners.push_back(rep);
    any = true;
  }
  if (!any) return;
  id_col.resize(src.cols.size(), -1);
  dict_id_col_ = id_col;
  in_types_ = src.types;
  dict_src_ = src;   // the emit step gathers the key strings from here
}

// one chunk of input rows resident in HBM → run the fused pipeline on it
void ExecutionContext::process_chunk(const std::vector<DeviceColumnView>& cols, const std::vector<bool>& has_valid, int64_t n) {
  if (n == 0) return;
  std::vector<int> fixed_lens(cols.size(), -1);
  for (size_t i = 0; i < cols.size(); i++) fixed_lens[i] = cols[i].fixed_len;
  Variant& v = variant_for(has_valid, fixed_lens);
  const PipelineDesc& d = v.desc;
  if (d.max_rows_exact && input_rows + n > d.max_rows_exact)
    throw CometError("decimal sum over more rows than the exactness bound allows (" + std::to_string(d.max_rows_exact) + ")");
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  prm.n = n;
  for (size_t i = 0; i < cols.size(); i++) {
    prm.in[i].data = cols[i].data;
    prm.in[i].valid = has_valid[i] ? cols[i].valid : nullptr;
    prm.in[i].aux = cols[i].aux;
    prm.in[i].offset = cols[i].offset;
  }
  err_flags_.ensure(kErrBytes);
  prm.out[kOutErr] = err_flags_.p;
  if (!has_join_) input_rows += n;

  if (d.sink == SinkKind::AggNoGroup) {
    if (agg_variant_ && agg_variant_->desc.NW != d.NW) throw CometError("internal: accumulator layout differs between variants");
    agg_variant_ = &v;
    const int64_t tile = (int64_t)d.R * 256;
    int grid = (int)std::min<int64_t>((n + tile - 1) / tile, 256 * 8);
    size_t need = (size_t)(n_partials_ + grid) * d.NW * 8;
    if (need > partials_.cap) {
      DevBuf bigger;
      bigger.ensure(std::max(need * 2, (size_t)(4096 * d.NW * 8)));
      if (n_partials_) HIP_CHECK(hipMemcpyAsync(bigger.p, partials_.p, (size_t)n_partials_ * d.NW * 8, hipMemcpyDeviceToDevice, stream_));
      HIP_CHECK(hipStreamSynchronize(stream_));
      std::swap(partials_.p, bigger.p);
      std::swap(partials_.cap, bigger.cap);
    }
    prm.out[kOutPartials] = (char*)partials_.p + (size_t)n_partials_ * d.NW * 8;
    for (int attempt = 0;; attempt++) {
      prm.iarg[kFixScaleArg] = packed_fix_scales(d);
      timed_begin();
      launch(v, "k_agg", grid, prm);
      timed_end();
      if (d.fix_sums.empty()) break;
      uint64_t aux[2 + 2 * kFixMaxSums];
      read_small(aux, err_flags_.p, sizeof aux);
      std::vector<int> shift;
      if (attempt >= 3 || !adjust_fix_scales(d, aux + 2, shift)) break;
      // the window moved: earlier chunks' partials follow it, this chunk's partials are simply overwritten by the re-run
      for (size_t f = 0; f < shift.size(); f++)
        if (shift[f] > 0 && n_partials_ > 0 &&
            comet_launch_fix_rescale((uint64_t*)partials_.p, n_partials_, d.NW, d.fix_sums[f].word, shift[f], stream_) != 0)
          throw CometError("float sum rescale: launch failed");
    }
    fix_has_state_ = fix_has_state_ || !d.fix_sums.empty();
    n_partials_ += grid;
    return;
  }

  if (d.sink == SinkKind::AggGrouped) {
    if (agg_variant_ && (agg_variant_->desc.NW != d.NW || agg_variant_->desc.NK != d.NK))
      throw CometError("internal: group slot layout differs between variants");
    agg_variant_ = &v;
    const size_t slot_bytes = 8 + 8 * (size_t)(d.NK + d.NW);
    auto alloc_table = [&](DevBuf& buf, int64_t cap) {
      buf.ensure((size_t)cap * slot_bytes);
      HIP_CHECK(hipMemsetAsync(buf.p, 0, (size_t)cap * slot_bytes, stream_));
    };
    if (group_cap_ == 0) {
      // start small: low-cardinality group-bys (TPC-H Q1: 4 groups) must not pay for a table sized by the row count
      int64_t want = 1 << 16;
      if (d.merges_states) {
        // merging Partial states: roughly one input row per group (SF100 Q3's Final aggregate: 1.13 M rows, 1.13 M groups) — size the table for
        // the chunk at once instead of filling and growing it twice (3.9 ms → one pass)
        while (want < 2 * n && want < ((int64_t)1 << 26)) want <<= 1;
      }
      group_cap_ = want;
      alloc_table(group_table_, group_cap_);
      HIP_CHECK(hipMemsetAsync((char*)err_flags_.p + 8, 0, 8, stream_));
    }
    const int64_t tile = (int64_t)d.R * 256;
    int grid_mult = 4;
    if (const char* e = getenv("COMET_GROUPED_GRID_MULT")) grid_mult = std::max(1, atoi(e));
    int grid = (int)std::min<int64_t>((n + tile - 1) / tile, 256 * grid_mult);
    // carry-save LDS accumulation bounds the rows one block may add (comet::kMaxRowsPerBlock = 2^19)
    const int64_t per_block_cap = ((int64_t)1 << 19) - 2 * tile;
    grid = (int)std::max<int64_t>(grid, (n + per_block_cap - 1) / per_block_cap);
    while (true) {
      // checkpoint: if the table fills up mid-chunk some rows are dropped, so the chunk is re-run from the checkpoint
      group_backup_.ensure((size_t)group_cap_ * slot_bytes);
      HIP_CHECK(hipMemcpyAsync(group_backup_.p, group_table_.p, (size_t)group_cap_ * slot_bytes, hipMemcpyDeviceToDevice, stream_));
      prm.out[0] = group_table_.p;
      prm.iarg[0] = group_cap_;
      prm.iarg[kFixScaleArg] = packed_fix_scales(d);
      timed_begin();
      launch(v, "k_gagg", grid, prm);
      timed_end();
      uint64_t head[2 + (kErrBytes - 16) / 8];
      uint32_t flags[4];
      read_small(head, err_flags_.p, d.fix_sums.empty() ? 16 : sizeof head);
      memcpy(flags, head, 16);
      uint64_t groups_now;
      memcpy(&groups_now, &flags[2], 8);
      const bool full = (flags[0] & 32u) != 0;
      if (!full && !d.fix_sums.empty() && fix_attempts_ < 3) {
        std::vector<int> shift;
        if (adjust_fix_scales(d, head + 2, shift)) {
          // a float sum's window moved: back to the checkpoint (table and group counter), shift what earlier chunks accumulated, run again
          fix_attempts_++;
          HIP_CHECK(hipMemcpyAsync(group_table_.p, group_backup_.p, (size_t)group_cap_ * slot_bytes, hipMemcpyDeviceToDevice, stream_));
          uint32_t restore[4] = {flags[0], flags[1], 0, 0};
          memcpy(&restore[2], &groups_committed_, 8);
          write_small(err_flags_.p, restore, 16);
          for (size_t f = 0; f < shift.size(); f++)
            if (shift[f] > 0 && comet_launch_fix_rescale((uint64_t*)group_table_.p, group_cap_, (int64_t)(slot_bytes / 8), 1 + d.NK + d.fix_sums[f].word, shift[f], stream_) != 0)
              throw CometError("float sum rescale: launch failed");
          continue;
        }
      }
      if (!full && (int64_t)groups_now * 2 <= group_cap_) { groups_committed_ = groups_now; break; }
      // grow ×8 and rehash; after a "full" event restart this chunk from the checkpoint
      int64_t new_cap = group_cap_ * 8;
      if (new_cap > ((int64_t)1 << 28)) throw CometError("group table would exceed 2^28 slots");
      DevBuf bigger;
      alloc_table(bigger, new_cap);
      uint32_t zero4[4] = {flags[0] & ~32u, flags[1], 0, 0};
      write_small(err_flags_.p, zero4, 16);
      CometKParams rp;
      memset(&rp, 0, sizeof rp);
      rp.out[0] = bigger.p;
      rp.iarg[0] = new_cap;
      rp.out[2] = err_flags_.p;
      rp.out[3] = full ? group_backup_.p : group_table_.p;
      rp.iarg[1] = group_cap_;
      launch(v, "k_grehash", (int)std::min<int64_t>((group_cap_ + 255) / 256, 256 * 8), rp);
      HIP_CHECK(hipStreamSynchronize(stream_));
      std::swap(group_table_.p, bigger.p);
      std::swap(group_table_.cap, bigger.cap);
      group_cap_ = new_cap;
      if (!full) { groups_committed_ = groups_now; break; }
    }
    fix_attempts_ = 0;
    fix_has_state_ = fix_has_state_ || !d.fix_sums.empty();
    return;
  }

  if (d.sink == SinkKind::Output) {
    for (auto& oc : d.out_cols)
      if (oc.gather_src >= 0 || oc.view_src >= 0) throw CometError("internal: gathered Utf8 outputs must go through the materialising path");
    const size_t ncol = d.out_cols.size();
    int64_t out_rows = n;
    if (out_vals_.size() < ncol) {
      out_vals_.resize(ncol);
      out_valid_.resize(ncol);
      for (size_t j = 0; j < ncol; j++) {
        if (!out_vals_[j]) out_vals_[j].reset(new DevBuf());
        if (!out_valid_[j]) out_valid_[j].reset(new DevBuf());
      }
    }
    auto bind_outputs = [&](int64_t rows_cap) {
      for (size_t j = 0; j < ncol; j++) {
        int w = d.out_cols[j].type.id == TypeId::Bool ? 1 : fixed_width(d.out_cols[j].type);
        out_vals_[j]->ensure((size_t)rows_cap * w + 16);
        prm.out[kOutFirstCol + 2 * j] = out_vals_[j]->p;
        if (d.out_cols[j].nullable) {
          out_valid_[j]->ensure((size_t)rows_cap + 16);
          prm.out[kOutFirstCol + 2 * j + 1] = out_valid_[j]->p;
        }
      }
    };
    timed_begin();
    if (d.has_filter) {
      // one pass: the survivor count is only known afterwards, so the outputs are sized for the whole chunk
      bind_outputs(n);
      out_rows = launch_fused_filter(v, prm, n);
    } else {
      bind_outputs(n);
      int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
      launch(v, "k_emit", grid, prm);
    }
    timed_end();
    check_device_errors();
    if (out_rows == 0) return;
    // device → host, then cut into batches of at most batch_size rows (FilterExec coalesces toward
    // the configured batch size; planner.rs:4688-4689)
    std::vector<std::vector<uint8_t>> hv(ncol), hk(ncol);
    for (size_t j = 0; j < ncol; j++) {
      int w = d.out_cols[j].type.id == TypeId::Bool ? 1 : fixed_width(d.out_cols[j].type);
      hv[j].resize((size_t)out_rows * w);
      HIP_CHECK(hipMemcpyAsync(hv[j].data(), out_vals_[j]->p, hv[j].size(), hipMemcpyDeviceToHost, stream_));
      if (d.out_cols[j].nullable) {
        hk[j].resize((size_t)out_rows);
        HIP_CHECK(hipMemcpyAsync(hk[j].data(), out_valid_[j]->p, hk[j].size(), hipMemcpyDeviceToHost, stream_));
      }
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    const int64_t bs = batch_size_ > 0 ? batch_size_ : out_rows;
    for (int64_t off = 0; off < out_rows; off += bs) {
      int64_t len = std::min(bs, out_rows - off);
      HostBatch b;
      b.rows = len;
      for (size_t j = 0; j < ncol; j++) {
        HostColumn c;
        c.type = d.out_cols[j].type;
        c.length = len;
        if (c.type.id == TypeId::Bool) {
          c.values.assign((size_t)((len + 7) / 8), 0);
          for (int64_t i = 0; i < len; i++)
            if (hv[j][(size_t)(off + i)]) c.values[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
        } else {
          int w = fixed_width(c.type);
          c.values.assign(hv[j].begin() + (size_t)off * w, hv[j].begin() + (size_t)(off + len) * w);
        }
        if (d.out_cols[j].nullable) {
          int64_t nulls = 0;
          std::vector<uint8_t> bm((size_t)((len + 7) / 8), 0);
          for (int64_t i = 0; i < len; i++) {
            if (hk[j][(size_t)(off + i)]) bm[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
            else nulls++;
          }
          c.null_count = nulls;
          if (nulls) c.validity = std::move(bm);
        }
        b.cols.push_back(std::move(c));
      }
      ready_.push_back(std::move(b));
    }
    return;
  }
  throw CometError("internal: unsupported sink");
}

// Small host↔device transfers go through a pinned scratch block: a copy to/from PAGEABLE memory makes the runtime set up
// staging for the stream, which was measured at 9–24 ms on the first such copy of each plan (profiles/r1_q3_*).
void ExecutionContext::read_small(void* dst, const void* dev_src, size_t n) {
  small_host_.ensure(4096);
  HIP_CHECK(hipMemcpyAsync(small_host_.p, dev_src, n, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
  memcpy(dst, small_host_.p, n);
}
void ExecutionContext::write_small(void* dev_dst, const void* src, size_t n) {
  small_host_.ensure(4096);
  HIP_CHECK(hipStreamSynchronize(stream_));   // the scratch may still be the source of an earlier async copy
  memcpy((char*)small_host_.p + 2048, src, n);
  HIP_CHECK(hipMemcpyAsync(dev_dst, (char*)small_host_.p + 2048, n, hipMemcpyHostToDevice, stream_));
}

void ExecutionContext::timed_begin() {
  hipEvent_t a = pool_get_event(device_id_), b = pool_get_event(device_id_);
  timed_.emplace_back(a, b);
  HIP_CHECK(hipEventRecord(a, stream_));
}
void ExecutionContext::timed_end() { HIP_CHECK(hipEventRecord(timed_.back().second, stream_)); }
// resolve the recorded event pairs (the stream must be idle)
void ExecutionContext::collect_timings() {
  for (; timed_done_ < timed_.size(); timed_done_++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, timed_[timed_done_].first, timed_[timed_done_].second) == hipSuccess) {
      last_kernel_ms += ms;
      last_kernel_launches++;
    }
  }
}

void ExecutionContext::check_device_errors() {
  if (!err_flags_.p) return;
  uint32_t flags[4] = {0, 0, 0, 0};
  read_small(flags, err_flags_.p, 16);
  collect_timings();
  raise_device_errors(flags[0]);
}

void ExecutionContext::raise_device_errors(uint32_t f) {
  if (!f) return;
  // Spark error JSON as thrown through CometQueryExecutionException (native/common/src/error.rs:806-831)
  if (f & 1u) throw CometError("{\"errorType\":\"ArithmeticOverflow\",\"errorClass\":\"ARITHMETIC_OVERFLOW\",\"params\":{\"fromType\":\"decimal\"}}", 1);
  if (f & 2u) throw CometError("{\"errorType\":\"ArithmeticOverflow\",\"errorClass\":\"ARITHMETIC_OVERFLOW\",\"params\":{\"fromType\":\"integer\"}}", 1);
  if (f & 4u) throw CometError("{\"errorType\":\"CastOverFlow\",\"errorClass\":\"CAST_OVERFLOW\",\"params\":{}}", 1);
  if (f & 8u) throw CometError("{\"errorType\":\"NumericValueOutOfRange\",\"errorClass\":\"NUMERIC_VALUE_OUT_OF_RANGE\",\"params\":{}}", 1);
  if (f & 256u) throw CometError("{\"errorType\":\"DivideByZero\",\"errorClass\":\"DIVIDE_BY_ZERO\",\"params\":{}}", 1);
  if (f & 64u) throw CometError("Utf8 group keys longer than 15 bytes are not supported by the GPU hash aggregate yet");
  if (f & 16u)
    throw CometError("decimal sum overflow cannot be decided order-independently for this input (mixed signs beyond the precision bound); "
                     "exact sequential evaluation is not implemented");
  throw CometError("device error flags " + std::to_string(f));
}

void ExecutionContext::finish_aggregate() {
  // AggregateExec emits one state row even for empty input
  std::vector<bool> none(in_types_.size(), false);
  Variant& v = agg_variant_ ? *agg_variant_ : variant_for(none, std::vector<int>(in_types_.size(), -1));
  const PipelineDesc& d = v.desc;
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  partials_.ensure(64);
  const size_t ncol = d.out_cols.size();
  // one result block: [kErrBytes error/aux words][32 B per output column: 16 B value, 1 B validity] → ONE D2H copy
  const size_t block_bytes = kErrBytes + ncol * 32;
  if (err_flags_.cap < block_bytes) {
    DevBuf bigger;
    bigger.ensure(block_bytes);
    HIP_CHECK(hipMemcpyAsync(bigger.p, err_flags_.p, kErrBytes, hipMemcpyDeviceToDevice, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
    std::swap(err_flags_.p, bigger.p);
    std::swap(err_flags_.cap, bigger.cap);
  }
  prm.out[kOutPartials] = partials_.p;
  prm.out[kOutErr] = err_flags_.p;
  prm.iarg[0] = n_partials_;
  char* base = (char*)err_flags_.p + kErrBytes;
  HIP_CHECK(hipMemsetAsync(base, 1, ncol * 32, stream_));  // validity defaults to 1
  for (size_t j = 0; j < ncol; j++) {
    prm.out[kOutFirstCol + 2 * j] = base + j * 32;
    prm.out[kOutFirstCol + 2 * j + 1] = base + j * 32 + 16;
  }
  prm.iarg[kFixScaleArg] = packed_fix_scales(d);
  launch(v, "k_agg_final", 1, prm);
  result_host_.ensure(block_bytes);
  HIP_CHECK(hipMemcpyAsync(result_host_.p, err_flags_.p, block_bytes, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
  collect_timings();
  raise_device_errors(((const uint32_t*)result_host_.p)[0]);
  const uint8_t* hb = (const uint8_t*)result_host_.p + kErrBytes;
  HostBatch b;
  b.rows = 1;
  for (size_t j = 0; j < ncol; j++) {
    HostColumn c;
    c.type = d.out_cols[j].type;
    c.length = 1;
    const uint8_t* val = hb + j * 32;
    if (c.type.id == TypeId::Bool) {
      c.values.assign(1, val[0] ? 1 : 0);
    } else {
      int w = fixed_width(c.type);
      c.values.assign(val, val + w);
    }
    if (d.out_cols[j].nullable && val[16] == 0) {
      c.null_count = 1;
      c.validity.assign(1, 0);
    }
    b.cols.push_back(std::move(c));
  }
  ready_.push_back(std::move(b));
}

// Grouped aggregate result left in HBM (stage boundary of a multi-GPU plan: Partial states feed the next stage's exchange
// or Final aggregate without touching the host).  Utf8 group keys are not supported on this path yet.
DevTable ExecutionContext::grouped_to_device() {
  DevTable empty;
  if (!agg_variant_) {   // no input rows → no groups: an empty table with the plan's output types
    std::vector<bool> none(in_types_.size(), false);
    auto pv = planned_variant(*plan_, plan_hash_, none, false, has_join_ ? &in_types_ : nullptr);
    for (auto& oc : pv->desc.out_cols) {
      empty.types.push_back(oc.type);
      empty.cols.push_back(DeviceColumnView());
      empty.has_valid.push_back(false);
    }
    return empty;
  }
  Variant& v = *agg_variant_;
  const PipelineDesc& d = v.desc;
  uint64_t ngroups = 0;
  read_small(&ngroups, (char*)err_flags_.p + 8, 8);
  check_device_errors();
  const size_t ncol = d.out_cols.size();
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  prm.out[0] = group_table_.p;
  prm.iarg[0] = group_cap_;
  scratch_counts_.ensure(64);
  HIP_CHECK(hipMemsetAsync(scratch_counts_.p, 0, 8, stream_));
  prm.out[1] = scratch_counts_.p;
  prm.out[kOutErr] = err_flags_.p;
  std::vector<std::shared_ptr<DevBuf>> vals(ncol), vbytes(ncol);
  for (size_t j = 0; j < ncol; j++) {
    vals[j] = std::make_shared<DevBuf>();
    vbytes[j] = std::make_shared<DevBuf>();
    vals[j]->ensure((size_t)std::max<uint64_t>(ngroups, 1) * out_width(d.out_cols[j]) + 16);
    vbytes[j]->ensure((size_t)std::max<uint64_t>(ngroups, 1) + 16);
    HIP_CHECK(hipMemsetAsync(vbytes[j]->p, 1, (size_t)std::max<uint64_t>(ngroups, 1), stream_));
    prm.out[kOutFirstCol + 2 * j] = vals[j]->p;
    prm.out[kOutFirstCol + 2 * j + 1] = vbytes[j]->p;
  }
  prm.iarg[kFixScaleArg] = packed_fix_scales(d);
  if (ngroups) launch(v, "k_gemit", (int)std::min<int64_t>((group_cap_ + 255) / 256, 256 * 8), prm);
  GatherSource gs = nullptr;
  if (!dict_id_col_.empty()) gs = [this](int c) { return std::make_pair((const DevTable*)&dict_src_, c); };
  DevTable t = outputs_to_table(v, vals, vbytes, (int64_t)ngroups, gs);
  t.owners.push_back(v.mod);
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  return t;
}

void ExecutionContext::finish_grouped() {
  if (!agg_variant_) return;  // no input rows → no groups → no output batch
  Variant& v = *agg_variant_;
  const PipelineDesc& d = v.desc;
  uint64_t ngroups = 0;
  read_small(&ngroups, (char*)err_flags_.p + 8, 8);
  check_device_errors();
  if (ngroups == 0) return;
  if (!dict_id_col_.empty()) {
    // keys that travelled as row indices: gather the strings on the device, then copy the finished table out
    DevTable t = grouped_to_device();
    table_to_host_batches(t);
    return;
  }
  const size_t ncol = d.out_cols.size();
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  prm.out[0] = group_table_.p;
  prm.iarg[0] = group_cap_;
  scratch_counts_.ensure(64);
  HIP_CHECK(hipMemsetAsync(scratch_counts_.p, 0, 8, stream_));
  prm.out[1] = scratch_counts_.p;
  prm.out[kOutErr] = err_flags_.p;
  out_vals_.resize(ncol);
  out_valid_.resize(ncol);
  std::vector<int> widths(ncol);
  for (size_t j = 0; j < ncol; j++) {
    if (!out_vals_[j]) out_vals_[j].reset(new DevBuf());
    if (!out_valid_[j]) out_valid_[j].reset(new DevBuf());
    const OutCol& oc = d.out_cols[j];
    widths[j] = oc.packed_string ? 16 : (oc.type.id == TypeId::Bool ? 1 : fixed_width(oc.type));
    out_vals_[j]->ensure((size_t)ngroups * widths[j] + 16);
    out_valid_[j]->ensure((size_t)ngroups + 16);
    HIP_CHECK(hipMemsetAsync(out_valid_[j]->p, 1, (size_t)ngroups, stream_));
    prm.out[kOutFirstCol + 2 * j] = out_vals_[j]->p;
    prm.out[kOutFirstCol + 2 * j + 1] = out_valid_[j]->p;
  }
  prm.iarg[kFixScaleArg] = packed_fix_scales(d);
  launch(v, "k_gemit", (int)std::min<int64_t>((group_cap_ + 255) / 256, 256 * 8), prm);
  // results come back through pooled pinned buffers (a pageable destination would be staged by the runtime at a fraction of the rate)
  struct HostSpan {
    PinnedBuf buf;
    size_t n = 0;
    const uint8_t* data() const { return (const uint8_t*)buf.p; }
    const uint8_t* begin() const { return data(); }
    uint8_t operator[](size_t i) const { return data()[i]; }
  };
  std::vector<HostSpan> hv(ncol), hk(ncol);
  for (size_t j = 0; j < ncol; j++) {
    hv[j].n = (size_t)ngroups * widths[j];
    hk[j].n = (size_t)ngroups;
    hv[j].buf.ensure(hv[j].n + 16);
    hk[j].buf.ensure(hk[j].n + 16);
    HIP_CHECK(hipMemcpyAsync(hv[j].buf.p, out_vals_[j]->p, hv[j].n, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipMemcpyAsync(hk[j].buf.p, out_valid_[j]->p, hk[j].n, hipMemcpyDeviceToHost, stream_));
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  const int64_t total = (int64_t)ngroups;
  const int64_t bs = batch_size_ > 0 ? batch_size_ : total;
  for (int64_t off = 0; off < total; off += bs) {
    const int64_t len = std::min(bs, total - off);
    HostBatch b;
    b.rows = len;
    for (size_t j = 0; j < ncol; j++) {
      const OutCol& oc = d.out_cols[j];
      HostColumn c;
      c.type = oc.type;
      c.length = len;
      if (oc.packed_string) {
        // expand str16 (bytes 0-7 | bytes 8-14 + length byte) into Arrow Utf8 offsets + data
        c.values.resize((size_t)(len + 1) * 4);
        int32_t* offs = (int32_t*)c.values.data();
        offs[0] = 0;
        for (int64_t i = 0; i < len; i++) {
          uint64_t w[2];
          memcpy(w, hv[j].data() + (size_t)(off + i) * 16, 16);
          int slen = hk[j][(size_t)(off + i)] ? (int)(w[1] >> 56) : 0;
          for (int k = 0; k < slen; k++) c.data.push_back((uint8_t)(k < 8 ? (w[0] >> (8 * k)) : (w[1] >> (8 * (k - 8)))));
          offs[i + 1] = (int32_t)c.data.size();
        }
      } else if (c.type.id == TypeId::Bool) {
        c.values.assign((size_t)((len + 7) / 8), 0);
        for (int64_t i = 0; i < len; i++)
          if (hv[j][(size_t)(off + i)]) c.values[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
      } else {
        int w = widths[j];
        c.values.assign(hv[j].begin() + (size_t)off * w, hv[j].begin() + (size_t)(off + len) * w);
      }
      if (oc.nullable) {
        const uint8_t* vb = hk[j].data() + off;
        int64_t valid = 0;
        for (int64_t i = 0; i < len; i++) valid += vb[i] != 0;
        c.null_count = len - valid;
        if (c.null_count) {
          std::vector<uint8_t> bm((size_t)((len + 7) / 8), 0);
          for (int64_t i = 0; i < len; i++)
            if (vb[i]) bm[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
          c.validity = std::move(bm);
        }
      }
      b.cols.push_back(std::move(c));
    }
    ready_.push_back(std::move(b));
  }
}

// Pull host batches from the JVM stream until a chunk is full; copy through pinned staging to HBM.
// Gather host batches of input `input` (up to max_rows rows) into one chunk resident in HBM.
// Returns false when nothing was read (stream exhausted); `rows` may be 0 with more to come only for empty batches.
// The stream's schema must be what the Scan declares (the reference casts mismatching inputs to the declared types,
// operators/scan.rs:134-164; casting is not implemented here, so a mismatch is an error instead of garbage).
void ExecutionContext::validate_input_schema(size_t input, const std::vector<DType>& types) {
  if (schema_checked_.size() <= input) schema_checked_.resize(input + 1, false);
  if (schema_checked_[input]) return;
  InputSource& in = inputs_[input];
  ArrowSchema sch;
  memset(&sch, 0, sizeof sch);
  int rc = in.kind == 0 ? in.host->get_schema(in.host, &sch) : in.dev->get_schema(in.dev, &sch);
  if (rc != 0 || !sch.release) throw CometError("input stream: get_schema failed");
  std::string err;
  if ((size_t)sch.n_children != types.size()) {
    err = "Scan declares " + std::to_string(types.size()) + " field(s) but the input stream has " + std::to_string(sch.n_children);
  } else {
    for (size_t c = 0; c < types.size() && err.empty(); c++) {
      const ArrowSchema* f = sch.children[c];
      const char* fmt = f->dictionary ? f->dictionary->format : f->format;
      if (format_matches(fmt, types[c])) continue;
      if (in.kind == 0 && !f->dictionary && scan_cast_supported(parse_src_format(fmt), types[c])) {
        if (scan_cast_from_.size() <= input) scan_cast_from_.resize(input + 1);
        scan_cast_from_[input].resize(types.size());
        scan_cast_from_[input][c] = fmt;
        continue;
      }
      err = "Scan input column " + std::to_string(c) + " has Arrow format '" + (fmt ? fmt : "?") + "' but the plan declares " + types[c].str() +
            " (this cast of a scan input is not supported by the MI355X native engine" + (in.kind == 0 ? ")" : "; device-resident inputs are never cast)");
    }
  }
  sch.release(&sch);
  if (!err.empty()) throw CometError(err);
  schema_checked_[input] = true;
}

bool ExecutionContext::pull_host_table(size_t input, const std::vector<DType>& in_types_, int64_t max_rows,
                                       std::vector<DeviceColumnView>& views, std::vector<bool>& has_valid, int64_t& rows_out) {
  InputSource& in = inputs_[input];
  rows_out = 0;
  if (in.exhausted) return false;
  validate_input_schema(input, in_types_);
  // two staging sets per input: while the GPU still reads chunk k (H2D + kernel are asynchronous) the host fills the other set
  // with chunk k+1; a set is reused only after the event recorded behind its last consumer has fired
  const size_t slot = input * 2 + (size_t)(stage_parity_ & 1);
  if (staging_.size() <= slot) staging_.resize(slot + 1);
  if (!staging_[slot]) staging_[slot].reset(new Staging());
  Staging& stg = *staging_[slot];
  if (stg.busy) {
    HIP_CHECK(hipEventSynchronize(stg.busy));
    pool_put_event(device_id_, stg.busy);
    stg.busy = nullptr;
  }
  auto& stage_vals_ = stg.stage_vals;
  auto& stage_valid_ = stg.stage_valid;
  auto& stage_aux_ = stg.stage_aux;
  auto& dev_vals_ = stg.dev_vals;
  auto& dev_valid_ = stg.dev_valid;
  auto& dev_aux_ = stg.dev_aux;
  const size_t nc = in_types_.size();
  if (stage_vals_.size() != nc) {
    stage_vals_.resize(nc);
    stage_valid_.resize(nc);
    stage_aux_.resize(nc);
    dev_vals_.resize(nc);
    dev_valid_.resize(nc);
    dev_aux_.resize(nc);
    for (size_t c = 0; c < nc; c++) {
      stage_vals_[c].reset(new PinnedBuf());
      stage_valid_[c].reset(new PinnedBuf());
      stage_aux_[c].reset(new PinnedBuf());
      dev_vals_[c].reset(new DevBuf());
      dev_valid_[c].reset(new DevBuf());
      dev_aux_[c].reset(new DevBuf());
    }
  }
  int64_t rows = 0;
  has_valid.assign(nc, false);
  std::vector<ArrowArray> held;
  // gather batches first so that staging buffers can be sized once
  while (rows < max_rows) {
    ArrowArray arr;
    memset(&arr, 0, sizeof arr);
    int rc = in.host->get_next(in.host, &arr);
    if (rc != 0) {
      const char* m = in.host->get_last_error ? in.host->get_last_error(in.host) : nullptr;
      for (auto& a : held) if (a.release) a.release(&a);
      throw CometError(std::string("input ArrowArrayStream.get_next failed: ") + (m ? m : "unknown error"));
    }
    if (!arr.release) {  // end of stream
      in.exhausted = true;
      break;
    }
    if ((size_t)arr.n_children != nc) {
      std::string msg = "input batch has " + std::to_string(arr.n_children) + " columns, Scan declares " + std::to_string(nc);
      arr.release(&arr);
      for (auto& a : held) if (a.release) a.release(&a);
      throw CometError(msg);
    }
    rows += arr.length;
    held.push_back(arr);
  }
  if (rows == 0) {
    for (auto& a : held) if (a.release) a.release(&a);
    return !in.exhausted;
  }
  for (auto& a : held)
    for (size_t c = 0; c < nc; c++)
      if (a.children[c]->null_count != 0 && a.children[c]->buffers[0]) has_valid[c] = true;
  std::vector<SrcFmt> cast_from(nc);
  if (scan_cast_from_.size() > input)
    for (size_t c = 0; c < nc && c < scan_cast_from_[input].size(); c++)
      if (!scan_cast_from_[input][c].empty()) {
        cast_from[c] = parse_src_format(scan_cast_from_[input][c].c_str());
        if (cast_from[c].cls != SrcFmt::LargeUtf8) has_valid[c] = true;   // a safe cast turns what does not fit into NULL
      }
  std::vector<size_t> aux_bytes(nc, 0);
  std::vector<int> str_uniform_(nc, -1);
  // index width of dictionary-encoded columns comes from the stream schema (fetched once per input)
  if (stg.dict_index_width.empty()) {
    stg.dict_index_width.assign(nc, 0);
    bool any_dict = false;
    for (auto& a : held)
      for (size_t c = 0; c < nc; c++) any_dict |= a.children[c]->dictionary != nullptr;
    if (any_dict) {
      ArrowSchema sch;
      memset(&sch, 0, sizeof sch);
      if (in.host->get_schema(in.host, &sch) != 0 || !sch.release) throw CometError("input stream: get_schema failed");
      for (size_t c = 0; c < nc && c < (size_t)sch.n_children; c++) {
        const ArrowSchema* f = sch.children[c];
        if (f->dictionary && f->format) {
          int w = f->format[0] == 'c' || f->format[0] == 'C' ? 1 : f->format[0] == 's' || f->format[0] == 'S' ? 2 : f->format[0] == 'i' || f->format[0] == 'I' ? 4 : 8;
          stg.dict_index_width[c] = w;
        }
      }
      sch.release(&sch);
    }
  }
  std::vector<std::shared_ptr<DevBuf>> dict_keep;
  std::vector<bool> dict_done(nc, false);
  for (size_t c = 0; c < nc; c++) {
    bool is_dict = false;
    for (auto& a : held) is_dict |= a.children[c]->dictionary != nullptr;
    if (!is_dict) continue;
    // ---- dictionary unpack on the device (K1): indices + dictionary go up, a gather kernel writes the plain column
    const DType& t = in_types_[c];
    const int iw = stg.dict_index_width[c];
    if (!iw) throw CometError("dictionary-encoded column without an index type in the stream schema");
    const bool is_str = t.id == TypeId::String || t.id == TypeId::Bytes;
    const int w = is_str ? 0 : (t.id == TypeId::Bool ? -1 : fixed_width(t));
    if (w < 0) throw CometError("dictionary-encoded boolean columns are not supported yet");
    auto vbytes = std::make_shared<DevBuf>();
    vbytes->ensure((size_t)rows + 16);
    dict_keep.push_back(vbytes);
    auto upload = [&](const void* src, size_t n) {
      auto d = std::make_shared<DevBuf>();
      d->ensure(n + 16);
      if (n) HIP_CHECK(hipMemcpy(d->p, src, n, hipMemcpyHostToDevice));
      dict_keep.push_back(d);
      return d;
    };
    struct Part { std::shared_ptr<DevBuf> idx, doffs, dbytes; int64_t at, len; };
    std::vector<Part> parts;
    auto lengths = std::make_shared<DevBuf>();
    if (is_str) lengths->ensure((size_t)rows * 4 + 16);
    else dev_vals_[c]->ensure((size_t)rows * w + 16);
    int64_t at = 0;
    bool any_null = false;
    for (auto& a : held) {
      const ArrowArray* col = a.children[c];
      const ArrowArray* dict = col->dictionary;
      if (!dict) throw CometError("a column mixes dictionary-encoded and plain batches");
      const int64_t len = col->length;
      auto d_idx = upload((const char*)col->buffers[1] + (size_t)col->offset * iw, (size_t)len * iw);
      std::shared_ptr<DevBuf> d_iv, d_dv;
      if (col->null_count != 0 && col->buffers[0]) {
        std::vector<uint8_t> bm((size_t)((len + 7) / 8) + 1, 0);
        bit_append(bm.data(), 0, (const uint8_t*)col->buffers[0], col->offset, len);
        d_iv = upload(bm.data(), bm.size());
        any_null = true;
      }
      if (dict->null_count != 0 && dict->buffers[0]) {
        std::vector<uint8_t> bm((size_t)((dict->length + 7) / 8) + 1, 0);
        bit_append(bm.data(), 0, (const uint8_t*)dict->buffers[0], dict->offset, dict->length);
        d_dv = upload(bm.data(), bm.size());
        any_null = true;
      }
      if (!is_str) {
        auto d_vals = upload((const char*)dict->buffers[1] + (size_t)dict->offset * w, (size_t)dict->length * w);
        comet_launch_dict_gather_fixed(d_idx->p, iw, d_iv ? (const uint8_t*)d_iv->p : nullptr, (const uint8_t*)d_vals->p,
                                       d_dv ? (const uint8_t*)d_dv->p : nullptr, w, len, (uint8_t*)dev_vals_[c]->p + (size_t)at * w,
                                       (uint8_t*)vbytes->p + at, stream_);
      } else {
        const int32_t* off = (const int32_t*)dict->buffers[1] + dict->offset;
        std::vector<int32_t> ro((size_t)dict->length + 1);
        for (int64_t k = 0; k <= dict->length; k++) ro[(size_t)k] = off[k] - off[0];
        auto d_off = upload(ro.data(), ro.size() * 4);
        auto d_bytes = upload((const char*)dict->buffers[2] + off[0], (size_t)ro[(size_t)dict->length]);
        comet_launch_dict_gather_str_len(d_idx->p, iw, d_iv ? (const uint8_t*)d_iv->p : nullptr, (const int32_t*)d_off->p,
                                         d_dv ? (const uint8_t*)d_dv->p : nullptr, len, (uint32_t*)lengths->p + at, (uint8_t*)vbytes->p + at, stream_);
        parts.push_back({d_idx, d_off, d_bytes, at, len});
      }
      at += len;
    }
    if (is_str) {
      auto tiles = std::make_shared<DevBuf>();
      tiles->ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
      dict_keep.push_back(tiles);
      dict_keep.push_back(lengths);
      dev_vals_[c]->ensure((size_t)(rows + 1) * 4 + 16);
      pq_launch_u32_scan((const uint32_t*)lengths->p, rows, (uint64_t*)tiles->p, (int32_t*)dev_vals_[c]->p, stream_);
      int32_t total = 0;
      read_small(&total, (char*)dev_vals_[c]->p + (size_t)rows * 4, 4);
      dev_aux_[c]->ensure((size_t)std::max(total, 1) + 16);
      for (auto& pt : parts)
        comet_launch_dict_gather_str_copy(pt.idx->p, iw, (const uint8_t*)vbytes->p + pt.at, (const int32_t*)pt.doffs->p, (const uint8_t*)pt.dbytes->p, pt.len,
                                          (const int32_t*)dev_vals_[c]->p + pt.at, (uint8_t*)dev_aux_[c]->p, stream_);
    }
    if (any_null) {
      has_valid[c] = true;
      dev_valid_[c]->ensure((size_t)((rows + 7) / 8) + 16);
      pq_launch_pack((const uint8_t*)vbytes->p, (uint8_t*)dev_valid_[c]->p, rows, stream_);
    } else {
      has_valid[c] = false;
    }
    dict_done[c] = true;
  }
  if (!dict_keep.empty()) HIP_CHECK(hipStreamSynchronize(stream_));   // uploaded indices/dictionaries are released below
  dict_keep.clear();
  for (size_t c = 0; c < nc; c++) {
    if (dict_done[c]) continue;
    const DType& t = in_types_[c];
    if (t.id == TypeId::String || t.id == TypeId::Bytes) {
      // Utf8: int32 offsets rebased to the chunk + concatenated bytes
      size_t total_bytes = 0;
      const bool large = cast_from[c].cls == SrcFmt::LargeUtf8;     // LargeUtf8 / LargeBinary: int64 offsets, cast to the declared Utf8
      auto off_at = [large](const ArrowArray* col, int64_t i) -> int64_t {
        return large ? ((const int64_t*)col->buffers[1])[col->offset + i] : (int64_t)((const int32_t*)col->buffers[1])[col->offset + i];
      };
      for (auto& a : held) {
        const ArrowArray* col = a.children[c];
        total_bytes += (size_t)(off_at(col, col->length) - off_at(col, 0));
      }
      if (total_bytes > 0x7fffffffull) throw CometError("Utf8 chunk exceeds 2 GiB of string bytes; lower spark.comet.gpu.chunkRows");
      stage_vals_[c]->ensure((size_t)(rows + 1) * 4 + 16);
      stage_aux_[c]->ensure(total_bytes + 16);
      if (has_valid[c]) stage_valid_[c]->ensure((size_t)((rows + 7) / 8) + 16);
      int32_t* so = (int32_t*)stage_vals_[c]->p;
      // one job per input batch: where its rows and bytes land is a running sum over the batches; rebasing the offsets, copying the
      // bytes and noticing whether all values share one length are independent per batch and spread over the scan threads (a single
      // thread walking 4 M offsets per chunk was what held the Utf8 columns of the host path below the PCIe rate)
      struct StrJob { const ArrowArray* col; int64_t at; int32_t pos; int uniform; };
      std::vector<StrJob> sjobs;
      int64_t at = 0;
      int32_t pos = 0;
      for (auto& a : held) {
        const ArrowArray* col = a.children[c];
        if (col->dictionary) throw CometError("dictionary-encoded input columns are not unpacked on the GPU path yet");
        sjobs.push_back({col, at, pos, -2});
        pos += (int32_t)(off_at(col, col->length) - off_at(col, 0));
        at += col->length;
      }
      auto run_job = [&](StrJob& j) {
        const ArrowArray* col = j.col;
        const int64_t base = off_at(col, 0);
        int uniform = -2;   // -2 no value seen yet, -1 lengths differ, else the common length
        int32_t* dst = so + j.at;
        for (int64_t i = 0; i < col->length; i++) {
          const int64_t o = off_at(col, i);
          dst[i] = j.pos + (int32_t)(o - base);
          const int len = (int)(off_at(col, i + 1) - o);
          if (uniform == -2) uniform = len;
          else if (uniform != len) uniform = -1;
        }
        j.uniform = uniform;
        const size_t nb = (size_t)(off_at(col, col->length) - base);
        if (nb) memcpy((char*)stage_aux_[c]->p + j.pos, (const char*)col->buffers[2] + base, nb);
      };
      if (rows >= (1 << 20) && sjobs.size() > 1) {
        const size_t parts = std::min<size_t>(16, sjobs.size());
        scan_pool_parallel(parts, [&](size_t pidx) {
          for (size_t k = pidx; k < sjobs.size(); k += parts) run_job(sjobs[k]);
        });
      } else {
        for (auto& j : sjobs) run_job(j);
      }
      int uniform = -2;
      for (auto& j : sjobs) {
        if (j.col->length == 0) continue;
        if (uniform == -2) uniform = j.uniform;
        else if (uniform != j.uniform) uniform = -1;
      }
      if (has_valid[c])       // bitmaps are small and batches need not start on a byte boundary: appended in order on this thread
        for (auto& j : sjobs) {
          if (j.col->null_count != 0 && j.col->buffers[0]) bit_append((uint8_t*)stage_valid_[c]->p, j.at, (const uint8_t*)j.col->buffers[0], j.col->offset, j.col->length);
          else bit_fill_ones((uint8_t*)stage_valid_[c]->p, j.at, j.col->length);
        }
      so[rows] = pos;
      dev_vals_[c]->ensure((size_t)(rows + 1) * 4 + 16);
      dev_aux_[c]->ensure(total_bytes + 16);
      HIP_CHECK(hipMemcpyAsync(dev_vals_[c]->p, stage_vals_[c]->p, (size_t)(rows + 1) * 4, hipMemcpyHostToDevice, stream_));
      if (total_bytes) HIP_CHECK(hipMemcpyAsync(dev_aux_[c]->p, stage_aux_[c]->p, total_bytes, hipMemcpyHostToDevice, stream_));
      if (has_valid[c]) {
        size_t kb = (size_t)((rows + 7) / 8);
        dev_valid_[c]->ensure(kb + 16);
        HIP_CHECK(hipMemcpyAsync(dev_valid_[c]->p, stage_valid_[c]->p, kb, hipMemcpyHostToDevice, stream_));
      }
      aux_bytes[c] = total_bytes;
      str_uniform_[c] = (uniform >= 0 && uniform <= 15) ? uniform : -1;
      continue;
    }
    const int w = fixed_width(t);
    size_t vbytes = w ? (size_t)rows * w : (size_t)((rows + 7) / 8);
    stage_vals_[c]->ensure(vbytes + 16);
    if (has_valid[c]) stage_valid_[c]->ensure((size_t)((rows + 7) / 8) + 16);
    int64_t at = 0;
    struct CopyJob { char* dst; const char* src; size_t n; };
    std::vector<CopyJob> jobs;
    for (auto& a : held) {
      const ArrowArray* col = a.children[c];
      if (col->dictionary) throw CometError("dictionary-encoded input columns are not unpacked on the GPU path yet");
      const int64_t len = col->length, off = col->offset;
      if (len != a.length) throw CometError("ragged input batch");
      if (cast_from[c].cls != SrcFmt::Unknown) {
        // ScanExec's cast to the declared type, fused into the staging copy; validity = source validity AND "the value fits"
        if (col->null_count != 0 && col->buffers[0]) bit_append((uint8_t*)stage_valid_[c]->p, at, (const uint8_t*)col->buffers[0], off, len);
        else bit_fill_ones((uint8_t*)stage_valid_[c]->p, at, len);
        if (!w) throw CometError("casting a scan input to Boolean is not supported");
        const char* src = (const char*)col->buffers[1] + (size_t)off * (size_t)cast_from[c].width;
        char* dst = (char*)stage_vals_[c]->p + (size_t)at * w;
        uint8_t* vb = (uint8_t*)stage_valid_[c]->p;
        for (int64_t i = 0; i < len; i++) {
          const int64_t bit = at + i;
          const bool ok = ((vb[bit >> 3] >> (bit & 7)) & 1) && scan_cast_value(cast_from[c], src, i, t, dst + (size_t)i * w);
          if (!ok) {
            vb[bit >> 3] &= (uint8_t)~(1u << (bit & 7));
            memset(dst + (size_t)i * w, 0, (size_t)w);
          }
        }
        at += len;
        continue;
      }
      if (w) {
        // Decimal128 buffers from the JVM may be only 8-byte aligned (aligned_stream_reader.rs:95-107);
        // the staging copy realigns them.
        jobs.push_back({(char*)stage_vals_[c]->p + (size_t)at * w, (const char*)col->buffers[1] + (size_t)off * w, (size_t)len * w});
      } else {
        bit_append((uint8_t*)stage_vals_[c]->p, at, (const uint8_t*)col->buffers[1], off, len);
      }
      if (has_valid[c]) {
        if (col->null_count != 0 && col->buffers[0]) bit_append((uint8_t*)stage_valid_[c]->p, at, (const uint8_t*)col->buffers[0], off, len);
        else bit_fill_ones((uint8_t*)stage_valid_[c]->p, at, len);
      }
      at += len;
    }
    if (vbytes >= ((size_t)8 << 20)) {
      // split big single copies so that one huge batch is spread too
      std::vector<CopyJob> pieces;
      const size_t kPiece = (size_t)4 << 20;
      for (auto& j : jobs)
        for (size_t o = 0; o < j.n; o += kPiece) pieces.push_back({j.dst + o, j.src + o, std::min(kPiece, j.n - o)});
      jobs.swap(pieces);
    }
    if (vbytes >= ((size_t)8 << 20) && jobs.size() > 1) {
      // a large column: the batch copies are spread over the scan threads (one thread tops out near 10–15 GB/s, PCIe needs 45+)
      const size_t parts = std::min<size_t>(16, jobs.size());
      scan_pool_parallel(parts, [&](size_t pidx) {
        for (size_t j = pidx; j < jobs.size(); j += parts) memcpy(jobs[j].dst, jobs[j].src, jobs[j].n);
      });
    } else {
      for (auto& j : jobs) memcpy(j.dst, j.src, j.n);
    }
    dev_vals_[c]->ensure(vbytes + 16);
    HIP_CHECK(hipMemcpyAsync(dev_vals_[c]->p, stage_vals_[c]->p, vbytes, hipMemcpyHostToDevice, stream_));
    if (has_valid[c]) {
      size_t kb = (size_t)((rows + 7) / 8);
      dev_valid_[c]->ensure(kb + 16);
      HIP_CHECK(hipMemcpyAsync(dev_valid_[c]->p, stage_valid_[c]->p, kb, hipMemcpyHostToDevice, stream_));
    }
  }
  for (auto& a : held) if (a.release) a.release(&a);
  views.assign(nc, DeviceColumnView());
  for (size_t c = 0; c < nc; c++) {
    views[c].data = dev_vals_[c]->p;
    views[c].valid = has_valid[c] ? (const uint8_t*)dev_valid_[c]->p : nullptr;
    views[c].aux = dev_aux_[c]->p;
    views[c].fixed_len = str_uniform_[c];   // staged bytes are contiguous from 0, offsets rebased
  }
  rows_out = rows;
  return true;
}

bool ExecutionContext::pull_host_chunk() {
  std::vector<DeviceColumnView> views;
  std::vector<bool> has_valid;
  int64_t rows = 0;
  if (!pull_host_table(0, in_types_, chunk_rows_, views, has_valid, rows)) return false;
  if (rows > 0) {
    process_chunk(views, has_valid, rows);
    // no host-side wait: mark this staging set busy until the work queued so far is done, and switch to the other set
    Staging& stg = *staging_[(size_t)(stage_parity_ & 1)];
    stg.busy = pool_get_event(device_id_);
    HIP_CHECK(hipEventRecord(stg.busy, stream_));
    stage_parity_ ^= 1;
  }
  return !inputs_[0].exhausted;
}

bool ExecutionContext::pull_device_table(size_t input, const std::vector<DType>& types, std::vector<DeviceColumnView>& views,
                                         std::vector<bool>& has_valid, int64_t& rows, std::shared_ptr<void>& keepalive) {
  InputSource& in = inputs_[input];
  rows = 0;
  if (in.exhausted) return false;
  static const bool trace = getenv("COMET_TRACE_STAGES") != nullptr;
  Timer tm;
  validate_input_schema(input, types);
  const double t_schema = tm.ns();
  auto da = std::make_shared<ArrowDeviceArray>();
  memset(da.get(), 0, sizeof(ArrowDeviceArray));
  int rc = in.dev->get_next(in.dev, da.get());
  if (trace) fprintf(stderr, "[comet] device input %zu: get_schema %.3f ms, get_next %.3f ms\n", input, t_schema / 1e6, (tm.ns() - t_schema) / 1e6);
  if (rc != 0) {
    const char* m = in.dev->get_last_error ? in.dev->get_last_error(in.dev) : nullptr;
    throw CometError(std::string("input ArrowDeviceArrayStream.get_next failed: ") + (m ? m : "unknown error"));
  }
  if (!da->array.release) {
    in.exhausted = true;
    return false;
  }
  // the producer's buffers stay alive until the keepalive is dropped
  keepalive = std::shared_ptr<void>(da.get(), [da](void*) mutable {
    if (da->array.release) da->array.release(&da->array);
  });
  if (da->device_type != ARROW_DEVICE_ROCM && da->device_type != ARROW_DEVICE_ROCM_HOST)
    throw CometError("device input stream must carry ARROW_DEVICE_ROCM memory");
  if (da->sync_event) HIP_CHECK(hipStreamWaitEvent(stream_, *(hipEvent_t*)da->sync_event, 0));
  const size_t nc = types.size();
  if ((size_t)da->array.n_children != nc) throw CometError("device batch column count does not match Scan fields");
  views.assign(nc, DeviceColumnView());
  has_valid.assign(nc, false);
  for (size_t c = 0; c < nc; c++) {
    const ArrowArray* col = da->array.children[c];
    if (col->dictionary) throw CometError("dictionary-encoded device columns are not supported yet");
    views[c].data = col->buffers[1];
    views[c].offset = col->offset;
    if (types[c].id == TypeId::String || types[c].id == TypeId::Bytes) views[c].aux = col->buffers[2];
    if (types[c].id == TypeId::Decimal && (((uintptr_t)col->buffers[1]) & 15))
      throw CometError("device Decimal128 buffers must be 16-byte aligned");
    if (col->null_count != 0 && col->buffers[0]) {
      has_valid[c] = true;
      views[c].valid = (const uint8_t*)col->buffers[0];
    }
  }
  rows = da->array.length;
  // Utf8 columns: check on the device whether all values share one length (one pass over the offsets, 4 B/row); if so the fused
  // kernels skip the offsets and the dependent byte load altogether.  All columns at once: the first / last offsets of every column
  // come back with ONE synchronisation, the verification launches run back to back, their flags come back with a second one.
  std::vector<size_t> scols;
  for (size_t c = 0; c < nc && rows > 0; c++)
    if (types[c].id == TypeId::String) scols.push_back(c);
  if (!scols.empty() && scols.size() <= 16) {
    small_host_.ensure(4096);
    int32_t* ends = (int32_t*)small_host_.p;                 // [2 k], [2 k + 1] = first / last offset of string column k
    for (size_t k = 0; k < scols.size(); k++) {
      const ArrowArray* col = da->array.children[scols[k]];
      const int32_t* off = (const int32_t*)col->buffers[1] + col->offset;
      HIP_CHECK(hipMemcpyAsync(ends + 2 * k, off, 4, hipMemcpyDeviceToHost, stream_));
      HIP_CHECK(hipMemcpyAsync(ends + 2 * k + 1, off + rows, 4, hipMemcpyDeviceToHost, stream_));
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    std::vector<int32_t> first(scols.size()), len(scols.size(), -1);
    uint32_t* flags = (uint32_t*)err_flags_.p + (kErrBytes / 4 - 16);   // last 16 words of the error/aux block: scratch
    HIP_CHECK(hipMemsetAsync(flags, 0, 64, stream_));
    bool any = false;
    for (size_t k = 0; k < scols.size(); k++) {
      first[k] = ends[2 * k];
      const int64_t total = (int64_t)ends[2 * k + 1] - ends[2 * k];
      if (total % rows != 0 || total / rows > 15 || total < 0) continue;
      const ArrowArray* col = da->array.children[scols[k]];
      const int32_t* off = (const int32_t*)col->buffers[1] + col->offset;
      if (comet_launch_utf8_uniform(off, rows, (int32_t)(total / rows), flags + k, stream_) != 0) continue;
      len[k] = (int32_t)(total / rows);
      any = true;
    }
    if (any) {
      uint32_t f[16];
      read_small(f, flags, 64);
      HIP_CHECK(hipMemsetAsync(flags, 0, 64, stream_));
      for (size_t k = 0; k < scols.size(); k++) {
        if (len[k] < 0 || f[k] != 0) continue;
        const size_t c = scols[k];
        const ArrowArray* col = da->array.children[c];
        // value i then sits at aux + (offset + i)·len.  Only claimed when that base IS the data buffer (an unsliced column), because the
        // offset-based accessors (substring, LIKE, views …) of the same kernels keep addressing aux + offsets[i]
        if ((int64_t)first[k] != (int64_t)col->offset * len[k]) continue;
        views[c].fixed_len = len[k];
      }
    }
  }
  return true;
}

// HBM-resident input (Arrow C Device stream, ARROW_DEVICE_ROCM): zero copy.
bool ExecutionContext::pull_device_batch() {
  std::vector<DeviceColumnView> views;
  std::vector<bool> has_valid;
  int64_t rows = 0;
  std::shared_ptr<void> keep;
  if (!pull_device_table(0, in_types_, views, has_valid, rows, keep)) return false;
  process_chunk(views, has_valid, rows);
  HIP_CHECK(hipStreamSynchronize(stream_));
  return true;
}

// ---------------------------------------------------------------------------------------------
// Plans with joins: every join input is materialised in HBM (chains are fused pipelines, joins are the
// materialisation points), then the root chain streams over the top join's output.
// ---------------------------------------------------------------------------------------------

// dense outputs written by an emit kernel (values + validity BYTES) → Arrow-layout device table (validity bitmaps)
// out row k = source string idx[k] (Arrow Utf8: int32 offsets + bytes), any length: lengths → scan → copy
void ExecutionContext::take_utf8(const DeviceColumnView& src, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits, int64_t rows,
                                 DeviceColumnView& out, std::vector<std::shared_ptr<void>>& owners) {
  auto offsets = std::make_shared<DevBuf>();
  offsets->ensure((size_t)(rows + 1) * 4 + 16);
  auto data = std::make_shared<DevBuf>();
  if (rows == 0) {
    HIP_CHECK(hipMemsetAsync(offsets->p, 0, 4, stream_));
    data->ensure(16);
  } else {
    DevBuf lengths, tiles;
    lengths.ensure((size_t)rows * 4 + 16);
    tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
    const int32_t* offs = (const int32_t*)src.data + src.offset;
    if (comet_launch_take_utf8_lengths(offs, idx, ok_bytes, src_valid_bits, rows, (uint32_t*)lengths.p, stream_) != 0) throw CometError("take_utf8: launch failed");
    pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)offsets->p, stream_);
    int32_t total = 0;
    read_small(&total, (char*)offsets->p + (size_t)rows * 4, 4);
    if (total < 0) throw CometError("Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
    data->ensure((size_t)std::max(total, 1) + 16);
    if (comet_launch_take_utf8_copy(offs, (const uint8_t*)src.aux, idx, ok_bytes, src_valid_bits, rows, (const int32_t*)offsets->p, (uint8_t*)data->p, stream_) != 0)
      throw CometError("take_utf8: launch failed");
    HIP_CHECK(hipStreamSynchronize(stream_));   // lengths / tiles go back to the pool
  }
  out.data = offsets->p;
  out.aux = data->p;
  out.offset = 0;
  owners.push_back(offsets);
  owners.push_back(data);
}

DevTable ExecutionContext::outputs_to_table(Variant& v, const std::vector<std::shared_ptr<DevBuf>>& vals,
                                            const std::vector<std::shared_ptr<DevBuf>>& valid_bytes, int64_t rows, const GatherSource& gather_source) {
  const PipelineDesc& d = v.desc;
  DevTable t;
  t.rows = rows;
  for (size_t j = 0; j < d.out_cols.size(); j++) {
    const OutCol& oc = d.out_cols[j];
    DeviceColumnView cv;
    cv.data = vals[j]->p;
    t.owners.push_back(vals[j]);
    bool hv = false;
    if (oc.packed_string) {
      // packed ≤ 15-byte strings (computed values, short group keys) → offsets + bytes: lengths, prefix sum, copy
      DevBuf lengths, tiles;
      auto offsets = std::make_shared<DevBuf>(), bytes = std::make_shared<DevBuf>();
      lengths.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
      tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
      offsets->ensure((size_t)(rows + 2) * 4);
      if (rows == 0) HIP_CHECK(hipMemsetAsync(offsets->p, 0, 8, stream_));
      if (comet_launch_str16_lengths(vals[j]->p, (oc.nullable && rows) ? (const uint8_t*)valid_bytes[j]->p : nullptr, rows, (uint32_t*)lengths.p, stream_) != 0)
        throw CometError("packed strings: launch failed");
      if (rows) pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)offsets->p, stream_);
      int32_t total = 0;
      if (rows) read_small(&total, (char*)offsets->p + (size_t)rows * 4, 4);
      bytes->ensure((size_t)total + 16);
      if (comet_launch_str16_copy(vals[j]->p, (const int32_t*)offsets->p, rows, (uint8_t*)bytes->p, stream_) != 0) throw CometError("packed strings: launch failed");
      HIP_CHECK(hipStreamSynchronize(stream_));   // lengths / tiles go back to the pool
      cv.data = offsets->p;
      cv.aux = bytes->p;
      t.owners.push_back(offsets);
      t.owners.push_back(bytes);
    }
    if (oc.view_src >= 0) {
      // the emit kernel wrote one strview per row (source row, byte slice, pad characters): lengths, prefix sum, copy
      if (!gather_source) throw CometError("internal: string-view column without a source table");
      auto src = gather_source(oc.view_src);
      const DeviceColumnView& sc = src.first->cols[(size_t)src.second];
      if (sc.fixed_len >= 0 && !sc.data) throw CometError("internal: string view over a column without offsets");
      const uint8_t* okb = (oc.nullable && rows) ? (const uint8_t*)valid_bytes[j]->p : nullptr;
      const uint8_t* pat = (const uint8_t*)oc.pad_pattern.data();
      const int32_t patn = (int32_t)oc.pad_pattern.size();
      DevBuf lengths, tiles;
      auto offsets = std::make_shared<DevBuf>(), bytes = std::make_shared<DevBuf>();
      lengths.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
      tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
      offsets->ensure((size_t)(rows + 2) * 4);
      if (rows == 0) HIP_CHECK(hipMemsetAsync(offsets->p, 0, 8, stream_));
      if (comet_launch_strview_lengths(vals[j]->p, okb, rows, pat, patn, (uint32_t*)lengths.p, stream_) != 0) throw CometError("string view: launch failed");
      if (rows) pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)offsets->p, stream_);
      int32_t total = 0;
      if (rows) read_small(&total, (char*)offsets->p + (size_t)rows * 4, 4);
      if (total < 0) throw CometError("Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
      bytes->ensure((size_t)total + 16);
      if (comet_launch_strview_copy(vals[j]->p, okb, (const int32_t*)sc.data + sc.offset, (const uint8_t*)sc.aux, rows, pat, patn, oc.pad_left ? 1 : 0,
                                    (const int32_t*)offsets->p, (uint8_t*)bytes->p, stream_) != 0)
        throw CometError("string view: launch failed");
      HIP_CHECK(hipStreamSynchronize(stream_));   // lengths / tiles go back to the pool
      cv.data = offsets->p;
      cv.aux = bytes->p;
      t.owners.push_back(offsets);
      t.owners.push_back(bytes);
    }
    if (oc.gather_src >= 0) {
      // the emit kernel wrote source row indices: gather the strings now
      if (!gather_source) throw CometError("internal: gathered Utf8 column without a source table");
      auto src = gather_source(oc.gather_src);
      const DeviceColumnView& sc = src.first->cols[(size_t)src.second];
      take_utf8(sc, (const uint32_t*)vals[j]->p, (oc.nullable && rows) ? (const uint8_t*)valid_bytes[j]->p : nullptr, nullptr, rows, cv, t.owners);
    }
    if (oc.type.id == TypeId::Bool) {
      // kernels store booleans as bytes; Arrow wants bits
      auto bits = std::make_shared<DevBuf>();
      bits->ensure((size_t)((rows + 7) / 8) + 16);
      CometKParams pk;
      memset(&pk, 0, sizeof pk);
      pk.n = rows;
      pk.out[0] = vals[j]->p;
      pk.out[1] = bits->p;
      if (rows) launch(v, "k_pack", (int)std::min<int64_t>((rows + 255) / 256, 2048), pk);
      cv.data = bits->p;
      t.owners.push_back(bits);
    }
    if (oc.nullable && rows) {
      auto bm = std::make_shared<DevBuf>();
      bm->ensure((size_t)((rows + 7) / 8) + 16);
      CometKParams pk;
      memset(&pk, 0, sizeof pk);
      pk.n = rows;
      pk.out[0] = valid_bytes[j]->p;
      pk.out[1] = bm->p;
      launch(v, "k_pack", (int)std::min<int64_t>((rows + 255) / 256, 2048), pk);
      cv.valid = (const uint8_t*)bm->p;
      t.owners.push_back(bm);
      t.owners.push_back(valid_bytes[j]);
      hv = true;
    }
    t.types.push_back(oc.type);
    t.cols.push_back(cv);
    t.has_valid.push_back(hv);
  }
  return t;
}

// ---- exact Float64 sums: window bookkeeping (device side: comet_device.hpp "Exact Float64 sums") ----

long long ExecutionContext::packed_fix_scales(const PipelineDesc& d) {
  if (fix_scales_.size() != d.fix_sums.size()) fix_scales_.assign(d.fix_sums.size(), kFixDefaultScale);
  uint64_t p = 0;
  for (size_t f = 0; f < fix_scales_.size(); f++) p |= (uint64_t)(uint16_t)(int16_t)fix_scales_[f] << (16 * f);
  return (long long)p;
}

// After a chunk: do the addends seen so far (aux words: 1200 + top and 1200 − low, maxima over every chunk of this execution) fit the
// fixed-point window [2^s, 2^(s + kFixW)) of every sum?  Returns true when scales changed and the chunk has to be run again;
// shift_right[f] > 0 means accumulators of earlier chunks must first be shifted right by that many bits.
//   * a value at or above 2^(s + kFixW) would lose HIGH bits: the window moves up (with 10 bits of slack), always;
//   * bits below 2^s are only truncated (error < rows · 2^s): the window moves down when nothing has been accumulated yet —
//     to the lowest bit seen when the whole range fits (then the sum is exact), else as low as the top value allows.
bool ExecutionContext::adjust_fix_scales(const PipelineDesc& d, const uint64_t* aux, std::vector<int>& shift_right) {
  bool rerun = false;
  shift_right.assign(d.fix_sums.size(), 0);
  for (size_t f = 0; f < d.fix_sums.size(); f++) {
    const uint64_t hi = aux[d.fix_sums[f].aux_hi], lo = aux[d.fix_sums[f].aux_lo];
    if (hi == 0) continue;                      // no finite non-zero addend yet
    const int top = (int)hi - 1200, low = 1200 - (int)lo, s = fix_scales_[f];
    int target = s;
    if (top > s + kFixW) target = top + 10 - kFixW;
    else if (low < s && !fix_has_state_) target = (top - low <= kFixW - 10) ? low : top + 2 - kFixW;
    if (target < -1300) target = -1300;
    if (target == s) continue;
    if (target > s && fix_has_state_) shift_right[f] = target - s;
    fix_scales_[f] = target;
    rerun = true;
  }
  return rerun;
}

// Single-pass filter + compaction (comet_device.hpp filter_fused_body): tile status words and the ticket / total block are zeroed,
// one launch, then the survivor count comes back.  Outputs must already be bound with room for n rows.
int64_t ExecutionContext::launch_fused_filter(Variant& v, CometKParams& prm, int64_t n) {
  const int64_t tile_rows = 256 * (int64_t)v.desc.R;   // P::R row slots per thread
  const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
  scratch_mask_.ensure((size_t)ntiles * 8 + 64);
  scratch_counts_.ensure(64);
  HIP_CHECK(hipMemsetAsync(scratch_mask_.p, 0, (size_t)ntiles * 8, stream_));
  HIP_CHECK(hipMemsetAsync(scratch_counts_.p, 0, 16, stream_));
  prm.out[0] = scratch_mask_.p;
  prm.out[1] = scratch_counts_.p;
  launch(v, "k_filter", (int)std::min<int64_t>(ntiles, 256 * 8), prm);
  uint64_t total = 0;
  read_small(&total, (char*)scratch_counts_.p + 8, 8);
  return (int64_t)total;
}

// Filter/Project chain `top` over the resident table `in` → resident table
DevTable ExecutionContext::run_chain_to_device(const Operator& top, const DevTable& in) {
  auto pv = planned_variant(top, plan_hash_ ^ (0x9E3779B97F4A7C15ull * (uint64_t)(node_id_[&top] + 1)), in.has_valid, true, &in.types);
  if (pv->desc.sink != SinkKind::Output) throw CometError("internal: run_chain_to_device on an aggregate chain");
  Variant v;
  v.desc = pv->desc;
  v.mod = jit_load(pv->code);
  const PipelineDesc& d = v.desc;
  const int64_t n = in.rows;
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  prm.n = n;
  for (size_t i = 0; i < in.cols.size(); i++) {
    prm.in[i].data = in.cols[i].data;
    prm.in[i].valid = in.has_valid[i] ? in.cols[i].valid : nullptr;
    prm.in[i].aux = in.cols[i].aux;
    prm.in[i].offset = in.cols[i].offset;
  }
  prm.out[kOutErr] = err_flags_.p;
  const size_t ncol = d.out_cols.size();
  std::vector<std::shared_ptr<DevBuf>> vals(ncol), vbytes(ncol);
  auto bind = [&](int64_t cap) {
    for (size_t j = 0; j < ncol; j++) {
      vals[j] = std::make_shared<DevBuf>();
      vals[j]->ensure((size_t)cap * out_width(d.out_cols[j]) + 16);
      prm.out[kOutFirstCol + 2 * j] = vals[j]->p;
      vbytes[j] = std::make_shared<DevBuf>();
      if (d.out_cols[j].nullable) {
        vbytes[j]->ensure((size_t)cap + 16);
        prm.out[kOutFirstCol + 2 * j + 1] = vbytes[j]->p;
      }
    }
  };
  int64_t out_rows = n;
  timed_begin();
  if (n == 0) {
    bind(1);
    out_rows = 0;
  } else if (d.has_filter) {
    bind(n);
    out_rows = launch_fused_filter(v, prm, n);
  } else {
    bind(n);
    launch(v, "k_emit", (int)std::min<int64_t>((n + 255) / 256, 256 * 8), prm);
  }
  timed_end();
  DevTable out = outputs_to_table(v, vals, vbytes, out_rows, [&](int c) { return std::make_pair(&in, c); });
  out.owners.push_back(v.mod);
  return out;
}

// Join keys that are Utf8 columns with values longer than the 15 bytes of a packed key: an exact string dictionary is built over the
// right column (strdict_kernels.hip), the left column is looked up in it, and the join runs on the two Int64 row-index columns
// instead (a left string without a partner gets a NULL index: NULL keys never match, outer joins still emit the row).  The index
// columns are appended to the inputs and dropped from the result.
DevTable ExecutionContext::hash_join(const Operator& j, const DevTable& L, const DevTable& R) {
  auto is_str = [](const DType& t) { return t.id == TypeId::String || t.id == TypeId::Bytes; };
  std::vector<size_t> sk;
  for (size_t k = 0; k < j.left_keys.size() && k < j.right_keys.size(); k++) {
    const ExprP &a = j.left_keys[k], &b = j.right_keys[k];
    if (a->kind == ExprKind::Bound && b->kind == ExprKind::Bound && a->bound_index >= 0 && b->bound_index >= 0 && (size_t)a->bound_index < L.types.size() &&
        (size_t)b->bound_index < R.types.size() && is_str(L.types[(size_t)a->bound_index]) && is_str(R.types[(size_t)b->bound_index]) &&
        L.cols[(size_t)a->bound_index].offset == 0 && R.cols[(size_t)b->bound_index].offset == 0)
      sk.push_back(k);
  }
  if (sk.empty() || L.rows == 0 || R.rows == 0 || L.cols.size() + R.cols.size() + 2 * sk.size() > COMET_MAX_IN) return hash_join_impl(j, j, L, R, "");
  auto longest = [&](const DevTable& t, int c) {
    uint32_t* mx = (uint32_t*)err_flags_.p + (kErrBytes / 4 - 1);
    HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
    if (comet_launch_str_max_len((const int32_t*)t.cols[(size_t)c].data, t.rows, mx, stream_) != 0) throw CometError("string keys: launch failed");
    uint32_t v = 0;
    read_small(&v, mx, 4);
    HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
    return v;
  };
  bool need = false;
  for (size_t k : sk) need = need || longest(L, j.left_keys[k]->bound_index) > 15 || longest(R, j.right_keys[k]->bound_index) > 15;
  if (!need) return hash_join_impl(j, j, L, R, "");
  if (R.rows >= ((int64_t)1 << 32) - 1) throw CometError("Utf8 join keys longer than 15 bytes over more than 2^32 rows are not supported");
  DevTable l2 = L, r2 = R;
  Operator jj = j;
  auto bound = [](int idx) {
    auto e = std::make_shared<Expr>();
    e->kind = ExprKind::Bound;
    e->proto_tag = 3;
    e->bound_index = idx;
    e->dtype = DType::of(TypeId::Int64);
    e->has_dtype = true;
    return e;
  };
  for (size_t k : sk) {
    const int lc = j.left_keys[k]->bound_index, rc = j.right_keys[k]->bound_index;
    const DeviceColumnView &lv = L.cols[(size_t)lc], &rv = R.cols[(size_t)rc];
    int64_t slots = 1024;
    while (slots < 2 * R.rows) slots <<= 1;
    DevBuf table;
    table.ensure((size_t)slots * 4);
    HIP_CHECK(hipMemsetAsync(table.p, 0, (size_t)slots * 4, stream_));
    auto rrep = std::make_shared<DevBuf>(), lrep = std::make_shared<DevBuf>(), lbits = std::make_shared<DevBuf>();
    DevBuf lok;
    rrep->ensure((size_t)R.rows * 8 + 16);
    lrep->ensure((size_t)L.rows * 8 + 16);
    lok.ensure((size_t)L.rows + 16);
    lbits->ensure((size_t)((L.rows + 7) / 8) + 16);
    if (comet_launch_str_dict_build((const int32_t*)rv.data, (const uint8_t*)rv.aux, R.has_valid[(size_t)rc] ? rv.valid : nullptr, R.rows, (uint32_t*)table.p, slots,
                                    (int64_t*)rrep->p, stream_) != 0 ||
        comet_launch_str_dict_lookup((const int32_t*)rv.data, (const uint8_t*)rv.aux, (const uint32_t*)table.p, slots, (const int32_t*)lv.data, (const uint8_t*)lv.aux,
                                     L.has_valid[(size_t)lc] ? lv.valid : nullptr, L.rows, (int64_t*)lrep->p, (uint8_t*)lok.p, stream_) != 0)
      throw CometError("string keys: launch failed");
    pq_launch_pack((const uint8_t*)lok.p, (uint8_t*)lbits->p, L.rows, stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));   // `table` and `lok` go back to the pool
    DeviceColumnView rid, lid;
    rid.data = rrep->p;
    rid.valid = rv.valid;
    lid.data = lrep->p;
    lid.valid = (const uint8_t*)lbits->p;
    jj.right_keys[k] = bound((int)r2.cols.size());
    jj.left_keys[k] = bound((int)l2.cols.size());
    r2.types.push_back(DType::of(TypeId::Int64));
    r2.cols.push_back(rid);
    r2.has_valid.push_back(R.has_valid[(size_t)rc]);
    r2.owners.push_back(rrep);
    l2.types.push_back(DType::of(TypeId::Int64));
    l2.cols.push_back(lid);
    l2.has_valid.push_back(true);
    l2.owners.push_back(lrep);
    l2.owners.push_back(lbits);
  }
  const size_t nl = L.cols.size(), nr = R.cols.size(), extra = sk.size();
  if (jj.join_condition) {
    // the residual condition addresses left ++ right: the right columns moved up by the index columns appended to the left
    std::function<ExprP(const ExprP&)> shift = [&](const ExprP& e) -> ExprP {
      auto c = std::make_shared<Expr>(*e);
      if (e->kind == ExprKind::Bound && (size_t)e->bound_index >= nl) c->bound_index = e->bound_index + (int)extra;
      for (auto& ch : c->children) ch = shift(ch);
      return c;
    };
    jj.join_condition = shift(j.join_condition);
  }
  DevTable out = hash_join_impl(j, jj, l2, r2, ":SD");
  // drop the index columns: the result is left' ++ right' (semi / anti joins: left' only)
  auto drop = [&](size_t first, size_t count) {
    if (first + count > out.cols.size()) return;
    out.types.erase(out.types.begin() + (long)first, out.types.begin() + (long)(first + count));
    out.cols.erase(out.cols.begin() + (long)first, out.cols.begin() + (long)(first + count));
    out.has_valid.erase(out.has_valid.begin() + (long)first, out.has_valid.begin() + (long)(first + count));
  };
  if (out.cols.size() == nl + extra + nr + extra) drop(nl + extra + nr, extra);
  else if (out.cols.size() != nl + extra) throw CometError("internal: unexpected join output width with string keys");
  drop(nl, extra);
  return out;
}

DevTable ExecutionContext::hash_join_impl(const Operator& node, const Operator& j, const DevTable& L, const DevTable& R, const std::string& key_suffix) {
  // planned once per (join node, validity patterns)
  std::string key = std::to_string(plan_hash_ ^ (0x9E3779B97F4A7C15ull * (uint64_t)(node_id_[&node] + 1))) + ":J:" + validity_key(L.has_valid) + "|" +
                    validity_key(R.has_valid) + key_suffix;
  std::shared_ptr<PlannedVariant> pv;
  {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    if (it != g_plan_cache.end()) pv = it->second;
  }
  if (!pv) {
    pv = std::make_shared<PlannedVariant>();
    pv->desc = generate_join(j, L.types, R.types, L.has_valid, R.has_valid);
    pv->code = jit_compile(pv->desc.source);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plan_cache[key] = pv;
  }
  Variant v;
  v.desc = pv->desc;
  v.mod = jit_load(pv->code);
  const PipelineDesc& d = v.desc;
  const bool build_left = j.build_side == BuildSide::Left;
  const DevTable& B = build_left ? L : R;
  const DevTable& P = build_left ? R : L;
  if (B.rows >= ((int64_t)1 << 31)) throw CometError("hash join build side exceeds 2^31 rows");
  const size_t nb = B.cols.size(), np = P.cols.size();
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  for (size_t i = 0; i < nb; i++) {
    prm.in[i].data = B.cols[i].data;
    prm.in[i].valid = B.has_valid[i] ? B.cols[i].valid : nullptr;
    prm.in[i].aux = B.cols[i].aux;
    prm.in[i].offset = B.cols[i].offset;
  }
  for (size_t i = 0; i < np; i++) {
    prm.in[nb + i].data = P.cols[i].data;
    prm.in[nb + i].valid = P.has_valid[i] ? P.cols[i].valid : nullptr;
    prm.in[nb + i].aux = P.cols[i].aux;
    prm.in[nb + i].offset = P.cols[i].offset;
  }
  int64_t cap = 1024;
  while (cap < 2 * B.rows) cap <<= 1;
  const int64_t n = P.rows;
  DevBuf head, next, matched, btiles;
  head.ensure((size_t)cap * 4);
  next.ensure((size_t)std::max<int64_t>(B.rows, 1) * 4);
  HIP_CHECK(hipMemsetAsync(head.p, 0xff, (size_t)cap * 4, stream_));
  const bool outer_build = d.join_outer_build;
  const int64_t nbtiles = (B.rows + 1023) / 1024;
  if (outer_build) {
    matched.ensure((size_t)std::max<int64_t>(B.rows, 1));
    btiles.ensure((size_t)(nbtiles + 1) * 8);
    HIP_CHECK(hipMemsetAsync(matched.p, 0, (size_t)std::max<int64_t>(B.rows, 1), stream_));
    prm.out[45] = matched.p;
    prm.out[46] = btiles.p;
    prm.iarg[3] = nbtiles;
  }
  prm.n = n;
  prm.iarg[0] = cap;
  prm.iarg[1] = B.rows;
  prm.out[0] = head.p;
  prm.out[1] = next.p;
  prm.out[kOutErr] = err_flags_.p;
  timed_begin();
  int64_t out_rows = 0, tail_rows = 0;
  const size_t ncol = d.out_cols.size();
  std::vector<std::shared_ptr<DevBuf>> vals(ncol), vbytes(ncol);
  auto bind_outputs = [&](int64_t rows_cap) {
    for (size_t c = 0; c < ncol; c++) {
      if (!vals[c]) vals[c] = std::make_shared<DevBuf>();
      vals[c]->ensure((size_t)std::max<int64_t>(rows_cap, 1) * out_width(d.out_cols[c]) + 16);
      prm.out[kOutFirstCol + 2 * c] = vals[c]->p;
      if (!vbytes[c]) vbytes[c] = std::make_shared<DevBuf>();
      if (d.out_cols[c].nullable) {
        vbytes[c]->ensure((size_t)std::max<int64_t>(rows_cap, 1) + 16);
        prm.out[kOutFirstCol + 2 * c + 1] = vbytes[c]->p;
      }
    }
  };
  // ---- single-pass probe (comet_device.hpp template D'): a small build side is hashed into LDS by every block, a large one into the
  // chained global table; either way the probe counts and emits in one launch, reserving output ranges with one atomic per tile ----
  const bool use_lds = B.rows > 0 && B.rows <= 6144 && getenv("COMET_JOIN_GLOBAL_TABLE") == nullptr;
  if (!use_lds && B.rows) launch(v, "k_jbuild", (int)std::min<int64_t>((B.rows + 255) / 256, 256 * 8), prm);
  DevBuf emitted_buf;
  emitted_buf.ensure(64);
  prm.out[47] = emitted_buf.p;
  // FK-shaped joins emit at most one row per probe row; anything beyond the capacity is counted, not written, and the probe re-run
  int64_t out_cap = d.join_build_only ? 1 : n + 1024;
  for (int attempt = 0; n > 0; attempt++) {
    bind_outputs(out_cap);
    prm.iarg[6] = out_cap;
    HIP_CHECK(hipMemsetAsync(emitted_buf.p, 0, 8, stream_));
    const int64_t ptiles = (n + 2047) / 2048;
    launch(v, use_lds ? "k_jlds" : "k_jprobe", (int)std::min<int64_t>(ptiles, use_lds ? 256 * 3 : 256 * 8), prm);
    uint64_t emitted = 0;
    read_small(&emitted, emitted_buf.p, 8);
    out_rows = d.join_build_only ? 0 : (int64_t)emitted;
    if (out_rows <= out_cap) break;
    if (attempt == 1) throw CometError("internal: hash join output exceeded its exact size");
    out_cap = out_rows;
  }
  const int64_t probe_capacity = n > 0 ? out_cap : 0;
  if (outer_build && B.rows > 0) {
    // build rows without a match follow the probe-driven rows
    launch(v, "k_jbcount", (int)std::min<int64_t>(nbtiles, 256 * 8), prm);
    launch(v, "k_jbscan", 1, prm);
    uint64_t total = 0;
    read_small(&total, (char*)btiles.p + (size_t)nbtiles * 8, 8);
    tail_rows = (int64_t)total;
    prm.iarg[4] = out_rows;
  }
  const int64_t all_rows = out_rows + tail_rows;
  if (all_rows > probe_capacity || (ncol > 0 && !vals[0])) {
    // the unmatched build rows follow the probe-driven rows: grow the output buffers, keeping what the probe wrote
    std::vector<std::shared_ptr<DevBuf>> ov = vals, ob = vbytes;
    for (size_t c = 0; c < ncol; c++) { vals[c].reset(); vbytes[c].reset(); }
    bind_outputs(all_rows);
    for (size_t c = 0; c < ncol && out_rows > 0; c++) {
      HIP_CHECK(hipMemcpyAsync(vals[c]->p, ov[c]->p, (size_t)out_rows * out_width(d.out_cols[c]), hipMemcpyDeviceToDevice, stream_));
      if (d.out_cols[c].nullable) HIP_CHECK(hipMemcpyAsync(vbytes[c]->p, ob[c]->p, (size_t)out_rows, hipMemcpyDeviceToDevice, stream_));
    }
    HIP_CHECK(hipStreamSynchronize(stream_));   // the old buffers return to the pool
  }
  if (tail_rows > 0) launch(v, "k_jbemit", (int)std::min<int64_t>(nbtiles, 256 * 8), prm);
  timed_end();
  out_rows = all_rows;
  const int nleft = (int)L.cols.size();
  DevTable out = outputs_to_table(v, vals, vbytes, out_rows, [&](int c) { return c < nleft ? std::make_pair(&L, c) : std::make_pair(&R, c - nleft); });
  HIP_CHECK(hipStreamSynchronize(stream_));  // head/next/counts go back to the pool when this frame ends
  out.owners.push_back(v.mod);
  join_build_rows_ += B.rows;
  join_probe_rows_ += P.rows;
  return out;
}

DevTable ExecutionContext::materialize(const Operator& op) {
  static const bool trace = getenv("COMET_TRACE_STAGES") != nullptr;
  Timer tm;
  struct Report {
    const Operator& op; Timer& tm; bool on;
    ~Report() { if (on) fprintf(stderr, "[comet] materialize %s: %.3f ms (incl. children)\n", op_name(op.proto_tag), tm.ns() / 1e6); }
  } report{op, tm, trace};
  if (op.kind == OpKind::Scan) {
    const size_t input = scan_input_.at(&op);
    DevTable t;
    t.types = op.scan_fields;
    if (inputs_[input].kind == 0) {
      // host stream: the whole input becomes one resident chunk (joins need their inputs complete)
      int64_t rows = 0;
      if (pull_host_table(input, op.scan_fields, INT64_MAX, t.cols, t.has_valid, rows)) t.rows = rows;
      if (t.cols.empty()) {
        t.cols.assign(op.scan_fields.size(), DeviceColumnView());
        t.has_valid.assign(op.scan_fields.size(), false);
      }
      HIP_CHECK(hipStreamSynchronize(stream_));
    } else {
      std::shared_ptr<void> keep;
      int64_t rows = 0;
      if (pull_device_table(input, op.scan_fields, t.cols, t.has_valid, rows, keep)) {
        t.rows = rows;
        t.owners.push_back(keep);
        std::vector<DeviceColumnView> c2;
        std::vector<bool> v2;
        std::shared_ptr<void> k2;
        int64_t r2 = 0;
        if (pull_device_table(input, op.scan_fields, c2, v2, r2, k2) && r2 > 0)
          throw CometError("a device input stream feeding a join must deliver a single batch");
      } else {
        t.cols.assign(op.scan_fields.size(), DeviceColumnView());
        t.has_valid.assign(op.scan_fields.size(), false);
      }
    }
    input_rows += t.rows;
    return t;
  }
  if (op.kind == OpKind::NativeScan) {
    DevTable t = scan_parquet(op);
    input_rows += t.rows;
    return t;
  }
  if (op.kind == OpKind::HashJoin) {
    DevTable l = materialize(*op.children[0]);
    DevTable r = materialize(*op.children[1]);
    DevTable j = hash_join(op, l, r);
    if (!op.smj || !smj_needs_sort_.count(&op)) return j;
    // SortMergeJoin: its output is ordered by the join keys (SortMergeJoinExec streams the sorted inputs, planner.rs:2126-2191)
    return sort_table(*smj_sorts_.at(&op), j);
  }
  if (op.kind == OpKind::Sort) {
    DevTable in = materialize(*op.children[0]);
    return sort_table(op, in);
  }
  if (op.kind == OpKind::Limit) {
    // LocalLimitExec / GlobalLimitExec (planner.rs:1436-1470): rows [offset, limit) of the child, in its order
    DevTable in = materialize(*op.children[0]);
    const int64_t off = std::min<int64_t>(std::max(0, op.offset), in.rows);
    const int64_t end = op.limit < 0 ? in.rows : std::min<int64_t>(in.rows, op.limit);
    return take_rows(in, nullptr, off, std::max<int64_t>(0, end - off), nullptr);
  }
  if (op.kind == OpKind::ShuffleWriter) return write_shuffle(op);
  if (op.kind == OpKind::Expand) {
    DevTable in = materialize(*op.children[0]);
    return expand(op, in);
  }
  if (op.kind == OpKind::Window) {
    DevTable in = materialize(*op.children[0]);
    return window(op, in);
  }
  if (op.kind == OpKind::HashAgg) return nested_aggregate(op);   // an aggregate below other operators
  // Filter / Projection chain: fused over its source
  const Operator* src = &op;
  do src = src->children[0].get(); while (!is_source(*src, &op));
  DevTable in = materialize(*src);
  DevTable out = run_chain_to_device(op, in);
  HIP_CHECK(hipStreamSynchronize(stream_));
  return out;
}



static u128 pow10_u128_host(int p) {
  u128 r = 1;
  for (int i = 0; i < p; i++) r *= 10;
  return r;
}

// Window: ranking / ntile / lag / lead and prefix-sum aggregates over input sorted by (partition keys, order keys) — see window_kernels.hip
DevTable ExecutionContext::window(const Operator& w, const DevTable& in) {
  const int64_t n = in.rows;
  if (n >= ((int64_t)1 << 31)) throw CometError("Window: more than 2^31 rows in one partition of the plan");
  DevTable out = in;
  auto add_col = [&](const DType& t, std::shared_ptr<DevBuf> data, std::shared_ptr<DevBuf> valid_bits, std::shared_ptr<DevBuf> aux = nullptr) {
    DeviceColumnView v;
    v.data = data ? data->p : nullptr;
    v.valid = valid_bits ? (const uint8_t*)valid_bits->p : nullptr;
    v.aux = aux ? aux->p : nullptr;
    out.types.push_back(t);
    out.cols.push_back(v);
    out.has_valid.push_back(valid_bits != nullptr);
    if (data) out.owners.push_back(data);
    if (valid_bits) out.owners.push_back(valid_bits);
    if (aux) out.owners.push_back(aux);
  };
  if (n == 0) {
    for (size_t k = 0; k < w.window_fns.size(); k++) {
      const std::string& f = w.window_fns[k].func;
      if (w.window_fns[k].is_agg) {
        const AggExpr& a = w.window_fns[k].agg;
        const bool dec = a.dtype.id == TypeId::Decimal && a.kind != AggKind::Count;
        if (a.kind == AggKind::Min || a.kind == AggKind::Max) add_col(in.types[(size_t)a.children[0]->bound_index], nullptr, nullptr);
        else add_col(dec ? a.dtype : DType::of(TypeId::Int64), nullptr, nullptr);
        continue;
      }
      DType t = (f == "percent_rank" || f == "cume_dist") ? DType::of(TypeId::Double) : (f == "lag" || f == "lead") ? in.types[(size_t)w.window_fns[k].args[0]->bound_index] : DType::of(TypeId::Int32);
      add_col(t, nullptr, nullptr);
    }
    return out;
  }
  timed_begin();
  int Wp = 0, Wo = 0;
  std::shared_ptr<DevBuf> pp, po;
  if (!window_psort_.at(&w)->sort_orders.empty()) pp = sort_key_planes(*window_psort_.at(&w), in, Wp);
  if (!window_osort_.at(&w)->sort_orders.empty()) po = sort_key_planes(*window_osort_.at(&w), in, Wo);
  DevBuf fpart, fpeer, tiles;
  auto sp = std::make_shared<DevBuf>(), sg = std::make_shared<DevBuf>(), first_part = std::make_shared<DevBuf>(), first_peer = std::make_shared<DevBuf>();
  fpart.ensure((size_t)n * 4 + 16);
  fpeer.ensure((size_t)n * 4 + 16);
  tiles.ensure((size_t)((n + 1023) / 1024 + 2) * 8);
  sp->ensure((size_t)(n + 2) * 4);
  sg->ensure((size_t)(n + 2) * 4);
  first_part->ensure((size_t)(n + 2) * 4);
  first_peer->ensure((size_t)(n + 2) * 4);
  if (comet_launch_window_flags(pp ? (const uint8_t*)pp->p : nullptr, Wp, po ? (const uint8_t*)po->p : nullptr, Wo, n, (uint32_t*)fpart.p, (uint32_t*)fpeer.p, stream_) != 0)
    throw CometError("window: launch failed");
  pq_launch_u32_scan((const uint32_t*)fpart.p, n, (uint64_t*)tiles.p, (int32_t*)sp->p, stream_);
  pq_launch_u32_scan((const uint32_t*)fpeer.p, n, (uint64_t*)tiles.p, (int32_t*)sg->p, stream_);
  if (comet_launch_window_first((const uint32_t*)fpart.p, (const int32_t*)sp->p, (const uint32_t*)fpeer.p, (const int32_t*)sg->p, n, (uint32_t*)first_part->p,
                                (uint32_t*)first_peer->p, stream_) != 0)
    throw CometError("window: launch failed");
  struct Prefix { std::shared_ptr<DevBuf> S, SH, C; };   // 128-bit inclusive sums (low part), sums of the high 64 bits (wide decimals only), non-NULL prefix counts
  std::map<int, Prefix> prefix;                           // by argument column
  for (auto& fn : w.window_fns) {
    if (fn.is_agg) {
      const AggExpr& a = fn.agg;
      const ExprP& arg = a.children[0];
      auto bound_kind = [&](int k, bool upper) { return k == 0 ? 0 : k == 1 ? 3 : (fn.frame_rows ? 1 : 2); (void)upper; };   // → WB_* of window_kernels.hip
      const int lo_kind = bound_kind(fn.frame_lower, false), hi_kind = bound_kind(fn.frame_upper, true);
      if (a.kind == AggKind::Min || a.kind == AggKind::Max) {
        // the frame's extreme: running extremes per partition from its start (P) and towards its end (Q) — two segmented scans — answer
        // every frame that touches a partition edge; a frame bounded on both sides is walked row by row (≤ 4097 rows)
        const int cc = arg->bound_index;
        const DeviceColumnView& sc = in.cols[(size_t)cc];
        if (sc.offset != 0) throw CometError("Window: aggregate over a column with a non-zero Arrow offset is not supported yet");
        const DType& at = in.types[(size_t)cc];
        const int width = at.id == TypeId::Decimal ? 16 : fixed_width(at);
        const int is_max = a.kind == AggKind::Max ? 1 : 0;
        DevBuf wide, okf, local, tl, P, Ph, Q, Qh;
        wide.ensure((size_t)n * 16 + 16);
        okf.ensure((size_t)n * 4 + 16);
        if (comet_launch_window_widen(width, sc.data, in.has_valid[(size_t)cc] ? sc.valid : nullptr, n, wide.p, nullptr, (uint32_t*)okf.p, stream_) != 0) throw CometError("window: launch failed");
        const bool need_p = lo_kind == 0, need_q = lo_kind != 0 && hi_kind == 0;
        if (need_p || need_q) {
          local.ensure((size_t)n * 32 + 64);
          tl.ensure((size_t)((n + 1023) / 1024 + 2) * 64 + 64);
          DevBuf& V = need_p ? P : Q;
          DevBuf& H = need_p ? Ph : Qh;
          V.ensure((size_t)n * 16 + 16);
          H.ensure((size_t)n + 16);
          if (comet_launch_window_running_extreme(wide.p, (const uint32_t*)okf.p, (const int32_t*)sp->p, n, need_p ? 0 : 1, is_max, local.p, tl.p, V.p, (uint8_t*)H.p, stream_) != 0)
            throw CometError("window: launch failed");
        }
        auto data = std::make_shared<DevBuf>(), okb = std::make_shared<DevBuf>(), bits = std::make_shared<DevBuf>();
        data->ensure((size_t)n * (size_t)width + 16);
        okb->ensure((size_t)n + 16);
        bits->ensure((size_t)((n + 7) / 8) + 16);
        if (comet_launch_window_minmax(is_max, lo_kind, fn.frame_lower_off, hi_kind, fn.frame_upper_off, wide.p, (const uint32_t*)okf.p, P.p, (const uint8_t*)Ph.p, Q.p, (const uint8_t*)Qh.p,
                                       (const int32_t*)sp->p, (const int32_t*)sg->p, (const uint32_t*)first_part->p, (const uint32_t*)first_peer->p, n, width, data->p, (uint8_t*)okb->p,
                                       stream_) != 0)
          throw CometError("window: launch failed");
        pq_launch_pack((const uint8_t*)okb->p, (uint8_t*)bits->p, n, stream_);
        HIP_CHECK(hipStreamSynchronize(stream_));   // scratch goes back to the pool
        add_col(at, data, bits);
        out.owners.push_back(okb);
        continue;
      }
      const int c = arg->kind == ExprKind::Bound ? arg->bound_index : -1 - (int)(arg->lit_null ? 1 : 0);   // literals: −1 non-NULL, −2 NULL
      auto it = prefix.find(c);
      if (it == prefix.end()) {
        auto S = std::make_shared<DevBuf>(), C = std::make_shared<DevBuf>();
        std::shared_ptr<DevBuf> SH;
        DevBuf wide, wide_hi, okf, t128, t32;
        wide.ensure((size_t)n * 16 + 16);
        okf.ensure((size_t)n * 4 + 16);
        S->ensure((size_t)n * 16 + 16);
        C->ensure((size_t)(n + 2) * 4);
        t128.ensure((size_t)((n + 2047) / 2048 + 2) * 16);
        t32.ensure((size_t)((n + 1023) / 1024 + 2) * 8);
        const void* src = nullptr;
        const uint8_t* vb = nullptr;
        int width = 8;
        if (c >= 0) {
          const DeviceColumnView& sc = in.cols[(size_t)c];
          if (sc.offset != 0) throw CometError("Window: aggregate over a column with a non-zero Arrow offset is not supported yet");
          src = sc.data;
          vb = in.has_valid[(size_t)c] ? sc.valid : nullptr;
          width = in.types[(size_t)c].id == TypeId::Decimal ? 16 : fixed_width(in.types[(size_t)c]);
        }
        DevBuf zero_bits;
        if (c == -2) {   // COUNT(NULL literal): no row counts — an all-zero validity bitmap
          zero_bits.ensure((size_t)((n + 7) / 8) + 16);
          HIP_CHECK(hipMemsetAsync(zero_bits.p, 0, (size_t)((n + 7) / 8), stream_));
          vb = (const uint8_t*)zero_bits.p;
        }
        const bool split = c >= 0 && in.types[(size_t)c].id == TypeId::Decimal && in.types[(size_t)c].precision > 18;
        if (split) {
          wide_hi.ensure((size_t)n * 16 + 16);
          SH = std::make_shared<DevBuf>();
          SH->ensure((size_t)n * 16 + 16);
        }
        if (comet_launch_window_widen(width, src, vb, n, wide.p, split ? wide_hi.p : nullptr, (uint32_t*)okf.p, stream_) != 0 ||
            comet_launch_scan128(wide.p, n, t128.p, S->p, stream_) != 0 || (split && comet_launch_scan128(wide_hi.p, n, t128.p, SH->p, stream_) != 0))
          throw CometError("window: launch failed");
        pq_launch_u32_scan((const uint32_t*)okf.p, n, (uint64_t*)t32.p, (int32_t*)C->p, stream_);
        HIP_CHECK(hipStreamSynchronize(stream_));   // scratch goes back to the pool
        it = prefix.emplace(c, Prefix{S, SH, C}).first;
      }
      const DType at = c >= 0 ? in.types[(size_t)c] : arg->dtype;
      int fnk = a.kind == AggKind::Count ? 2 : a.kind == AggKind::Avg ? 3 : (at.id == TypeId::Decimal ? 0 : 1);
      const DType rt = fnk == 2 || fnk == 1 ? DType::of(TypeId::Int64) : a.dtype;
      // precision bounds: SUM checks the result type; AVG checks the sum type, scales by 10^(result scale − sum scale) and checks the result type
      const DType sum_t = fnk == 3 ? a.sum_dtype : a.dtype;
      u128 bound = fnk == 0 || fnk == 3 ? pow10_u128_host(sum_t.precision) - 1 : 0, avg_bound = fnk == 3 ? pow10_u128_host(a.dtype.precision) - 1 : 0;
      i128 scaler = fnk == 3 ? (i128)pow10_u128_host(std::max(0, a.dtype.scale - sum_t.scale)) : 1;
      auto data = std::make_shared<DevBuf>(), okb = std::make_shared<DevBuf>(), bits = std::make_shared<DevBuf>();
      data->ensure((size_t)n * (fnk == 0 || fnk == 3 ? 16 : 8) + 16);
      okb->ensure((size_t)n + 16);
      bits->ensure((size_t)((n + 7) / 8) + 16);
      if (comet_launch_window_agg(fnk, lo_kind, fn.frame_lower_off, hi_kind, fn.frame_upper_off, it->second.S->p, it->second.SH ? it->second.SH->p : nullptr, (const int32_t*)it->second.C->p, (const int32_t*)sp->p, (const int32_t*)sg->p,
                                  (const uint32_t*)first_part->p, (const uint32_t*)first_peer->p, n, &bound, &scaler, &avg_bound, data->p, (uint8_t*)okb->p, stream_) != 0)
        throw CometError("window: launch failed");
      pq_launch_pack((const uint8_t*)okb->p, (uint8_t*)bits->p, n, stream_);
      add_col(rt, data, fnk == 2 ? nullptr : bits);
      if (fnk == 2) out.owners.push_back(bits);
      out.owners.push_back(okb);
      continue;
    }
    const std::string& f = fn.func;
    int kind = f == "row_number" ? 0 : f == "rank" ? 1 : f == "dense_rank" ? 2 : f == "percent_rank" ? 3 : f == "cume_dist" ? 4 : f == "ntile" ? 5 : -1;
    if (kind >= 0) {
      const bool dbl = kind == 3 || kind == 4;
      auto data = std::make_shared<DevBuf>();
      data->ensure((size_t)n * (dbl ? 8 : 4) + 16);
      if (comet_launch_window_rank(kind, kind == 5 ? fn.args[0]->lit_i64 : 0, (const int32_t*)sp->p, (const int32_t*)sg->p, (const uint32_t*)first_part->p,
                                   (const uint32_t*)first_peer->p, n, data->p, stream_) != 0)
        throw CometError("window: launch failed");
      add_col(DType::of(dbl ? TypeId::Double : TypeId::Int32), data, nullptr);
      continue;
    }
    // lag / lead: a gather with NULL outside the partition
    const int c = fn.args[0]->bound_index;
    const int64_t k = fn.args.size() >= 2 ? fn.args[1]->lit_i64 : 1;
    const int64_t shift = f == "lag" ? -k : k;
    const DType& t = in.types[(size_t)c];
    const DeviceColumnView& sc = in.cols[(size_t)c];
    if (sc.offset != 0) throw CometError(f + " over a column with a non-zero Arrow offset is not supported yet");
    DevBuf ok;
    auto idx = std::make_shared<DevBuf>(), okv = std::make_shared<DevBuf>(), bits = std::make_shared<DevBuf>();
    idx->ensure((size_t)n * 4 + 16);
    ok.ensure((size_t)n + 16);
    okv->ensure((size_t)n + 16);
    bits->ensure((size_t)((n + 7) / 8) + 16);
    if (comet_launch_window_offset(shift, (const int32_t*)sp->p, (const uint32_t*)first_part->p, n, (uint32_t*)idx->p, (uint8_t*)ok.p, stream_) != 0 ||
        comet_launch_window_offset_valid((const uint32_t*)idx->p, (const uint8_t*)ok.p, in.has_valid[(size_t)c] ? sc.valid : nullptr, n, (uint8_t*)okv->p, stream_) != 0)
      throw CometError("window: launch failed");
    const bool has_default = fn.args.size() == 3 && !fn.args[2]->lit_null;
    if (!has_default) pq_launch_pack((const uint8_t*)okv->p, (uint8_t*)bits->p, n, stream_);
    if (t.id == TypeId::String || t.id == TypeId::Bytes) {
      DeviceColumnView ov;
      take_utf8(sc, (const uint32_t*)idx->p, (const uint8_t*)okv->p, nullptr, n, ov, out.owners);
      ov.valid = (const uint8_t*)bits->p;
      out.types.push_back(t);
      out.cols.push_back(ov);
      out.has_valid.push_back(true);
      out.owners.push_back(bits);
    } else {
      const int wd = t.id == TypeId::Bool ? 0 : fixed_width(t);
      auto data = std::make_shared<DevBuf>();
      data->ensure((wd ? (size_t)n * (size_t)wd : (size_t)((n + 7) / 8)) + 16);
      if (comet_launch_take(wd, sc.data, (const uint32_t*)idx->p, n, data->p, stream_) != 0) throw CometError("window: take failed");
      if (has_default) {
        // rows whose offset row is outside the partition take the literal default (lag(x, k, d))
        const Expr& lit = *fn.args[2];
        uint8_t buf[16] = {0};
        if (t.id == TypeId::Decimal) { i128 v = lit.lit_dec; memcpy(buf, &v, 16); }
        else if (t.id == TypeId::Double) { double v = lit.lit_f64; memcpy(buf, &v, 8); }
        else if (t.id == TypeId::Float) { float v = (float)lit.lit_f64; memcpy(buf, &v, 4); }
        else { int64_t v = lit.lit_i64; memcpy(buf, &v, 8); }
        if (comet_launch_window_default(wd, (const uint8_t*)ok.p, n, buf, data->p, (uint8_t*)okv->p, stream_) != 0) throw CometError("window: launch failed");
        pq_launch_pack((const uint8_t*)okv->p, (uint8_t*)bits->p, n, stream_);
      }
      add_col(t, data, bits);
    }
    HIP_CHECK(hipStreamSynchronize(stream_));   // `ok` goes back to the pool; idx / okv are released with this scope
  }
  timed_end();
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  return out;
}

DevTable ExecutionContext::expand(const Operator& ex, const DevTable& in) {
  const ExpandInfo& info = expand_info_.at(&ex);
  const size_t ncol = info.out_cols.size(), P = info.parts.size();
  const int64_t n = in.rows, total = n * (int64_t)P;
  if (total >= ((int64_t)1 << 32)) throw CometError("Expand: more than 2^32 output rows in one partition");
  Variant u;   // unified description of the output columns; k_pack comes from the first generated projection
  u.desc.out_cols = info.out_cols;
  std::vector<std::shared_ptr<DevBuf>> vals(ncol), vbytes(ncol);
  for (size_t c = 0; c < ncol; c++) {
    vals[c] = std::make_shared<DevBuf>();
    vbytes[c] = std::make_shared<DevBuf>();
    vals[c]->ensure((size_t)std::max<int64_t>(total, 1) * (size_t)out_width(info.out_cols[c]) + 16);
    vbytes[c]->ensure((size_t)std::max<int64_t>(total, 1) + 16);
    HIP_CHECK(hipMemsetAsync(vbytes[c]->p, 1, (size_t)std::max<int64_t>(total, 1), stream_));   // outputs the kernels treat as non-nullable stay valid
  }
  timed_begin();
  for (size_t p = 0; p < P; p++) {
    const ExpandPart& part = info.parts[p];
    const int64_t base = (int64_t)p * n;
    if (!part.proj->project_list.empty()) {
      auto pv = planned_variant(*part.proj, plan_hash_ ^ (0x9E3779B97F4A7C15ull * (uint64_t)(node_id_[part.proj.get()] + 1)), in.has_valid, true, &in.types);
      Variant v;
      v.desc = pv->desc;
      v.mod = jit_load(pv->code);
      if (!u.mod) u.mod = v.mod;
      CometKParams prm;
      memset(&prm, 0, sizeof prm);
      prm.n = n;
      for (size_t i = 0; i < in.cols.size(); i++) {
        prm.in[i].data = in.cols[i].data;
        prm.in[i].valid = in.has_valid[i] ? in.cols[i].valid : nullptr;
        prm.in[i].aux = in.cols[i].aux;
        prm.in[i].offset = in.cols[i].offset;
      }
      prm.out[kOutErr] = err_flags_.p;
      for (size_t k = 0; k < part.out_col.size(); k++) {
        const size_t c = (size_t)part.out_col[k];
        prm.out[kOutFirstCol + 2 * k] = (char*)vals[c]->p + (size_t)base * (size_t)out_width(info.out_cols[c]);
        prm.out[kOutFirstCol + 2 * k + 1] = (char*)vbytes[c]->p + (size_t)base;
      }
      if (n) launch(v, "k_emit", (int)std::min<int64_t>((n + 255) / 256, 256 * 8), prm);
      u.desc.kernels = v.desc.kernels;
      HIP_CHECK(hipStreamSynchronize(stream_));   // v (and its module reference) goes out of scope
    }
    for (int c : part.null_cols) {
      if (!n) continue;
      HIP_CHECK(hipMemsetAsync((char*)vbytes[(size_t)c]->p + (size_t)base, 0, (size_t)n, stream_));
      HIP_CHECK(hipMemsetAsync((char*)vals[(size_t)c]->p + (size_t)base * (size_t)out_width(info.out_cols[(size_t)c]), 0,
                               (size_t)n * (size_t)out_width(info.out_cols[(size_t)c]), stream_));
    }
  }
  timed_end();
  if (!u.mod) throw CometError("Expand: every projection consists of NULL literals only");
  DevTable out = outputs_to_table(u, vals, vbytes, total, [&](int c) { return std::make_pair(&in, c); });
  out.owners.push_back(u.mod);
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  return out;
}

// ShuffleWriter (native/shuffle/src/shuffle_writer.rs:166-300, partitioners/multi_partition.rs:265-457, single_partition.rs):
// the child's whole output is resident in HBM; partition ids (Spark murmur3 seed 42 chained over the hash expressions → pmod),
// the stable per-partition row order and the per-column gathers all run on the GPU (the exchange kernels), ONE download brings the
// partition-major table to pinned host memory, and the host threads frame it: per partition, blocks of ≤ batch_size rows in input
// order (partitioned_batch_iterator.rs:100-124), each an Arrow IPC stream behind the 20-byte header, codec applied per block
// (shuffle_block_writer.rs:179-238).  Data file = partitions back to back; index file = num_partitions + 1 little-endian i64 offsets
// (writers/local/local_partition_writer.rs:255-295).
DevTable ExecutionContext::write_shuffle(const Operator& sw) {
  static const bool trace = getenv("COMET_TRACE_STAGES") != nullptr;
  Timer tm;
  double t_last = 0;
  auto lap = [&](const char* what) {
    if (!trace) return;
    const double now = tm.ns() / 1e6;
    fprintf(stderr, "[comet] shuffle write: %s %.3f ms\n", what, now - t_last);
    t_last = now;
  };
  auto sp = shuffle_projs_.find(&sw);
  const Operator& child = sp != shuffle_projs_.end() ? *sp->second : *sw.children[0];
  DevTable in = materialize(child);
  const int64_t n = in.rows;
  lap("child");
  const int P = sw.shuffle_partitioning == Operator::Partitioning::Single ? 1 : sw.shuffle_num_partitions;
  size_t n_payload = in.cols.size();
  std::vector<int> key_cols;
  {
    size_t appended = 0;   // computed key expressions sit behind the payload columns (the synthetic projection of the constructor)
    for (auto& e : sw.shuffle_hash_exprs) appended += e->kind != ExprKind::Bound;
    for (auto& k : sw.shuffle_sort_orders) appended += k.child->kind != ExprKind::Bound;
    n_payload -= appended;
  }
  if (sw.shuffle_partitioning == Operator::Partitioning::Hash) {
    size_t next = n_payload;
    for (auto& e : sw.shuffle_hash_exprs) key_cols.push_back(e->kind == ExprKind::Bound ? e->bound_index : (int)next++);
  } else if (sw.shuffle_partitioning == Operator::Partitioning::RoundRobin) {
    // "round robin" = hash of the first max_hash_columns columns (multi_partition.rs:386-437)
    const size_t k = sw.shuffle_max_hash_columns <= 0 ? n_payload : std::min<size_t>((size_t)sw.shuffle_max_hash_columns, n_payload);
    for (size_t i = 0; i < k; i++) key_cols.push_back((int)i);
  }
  if (n >= (int64_t)1 << 32) throw CometError("ShuffleWriter: more than 2^32 rows in one task are not supported (u32 row indices, multi_partition.rs)");
  std::vector<int64_t> starts((size_t)P + 1, 0);
  starts[(size_t)P] = n;
  DevTable grouped;
  if (P > 1 && n > 0) {
    DevBuf hashes, pids, dstarts, hist;
    auto ridx = std::make_shared<DevBuf>();
    hashes.ensure((size_t)n * 4);
    pids.ensure((size_t)n * 4);
    ridx->ensure((size_t)n * 4 + 16);
    dstarts.ensure(((size_t)P + 1) * 8);
    const bool by_range = sw.shuffle_partitioning == Operator::Partitioning::Range;
    std::shared_ptr<DevBuf> planes;
    DevBuf bkeys;
    if (by_range) {
      // order-preserving key bytes of every row and of every boundary row (same generated kernel, same widths), then an
      // upper-bound search per row: partition = number of boundaries ≤ row (multi_partition.rs:352-358)
      int W = 0, Wb = 0;
      const int B = (int)sw.shuffle_bounds.size();
      std::vector<DType> btypes;
      for (auto& k : range_sort_.at(&sw)->sort_orders) btypes.push_back(k.child->dtype);
      DevTable bt = literal_table(sw.shuffle_bounds, btypes);
      // Utf8 keys: rows and boundaries must be padded to the same length — the longer of the two
      std::vector<int64_t> lr, lb;
      {
        int w0 = 0;
        (void)sort_key_planes(*range_sort_.at(&sw), in, w0, &lr, true);              // measure only
        if (B > 0) (void)sort_key_planes(*range_bsort_.at(&sw), bt, w0, &lb, true);
        for (size_t s = 0; s < lr.size(); s++) lr[s] = std::max<int64_t>(lr[s], s < lb.size() ? lb[s] : 0);
        lb = lr;
      }
      planes = sort_key_planes(*range_sort_.at(&sw), in, W, &lr);
      std::vector<uint8_t> rowmajor((size_t)std::max(B, 1) * (size_t)std::max(W, 1), 0);
      if (B > 0) {
        auto bplanes = sort_key_planes(*range_bsort_.at(&sw), bt, Wb, &lb);
        if (Wb != W) throw CometError("internal: range boundary keys and row keys differ in width");
        std::vector<uint8_t> pl((size_t)W * (size_t)B);
        HIP_CHECK(hipMemcpyAsync(pl.data(), bplanes->p, pl.size(), hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        for (int b = 0; b < B; b++)
          for (int p = 0; p < W; p++) rowmajor[(size_t)b * W + p] = pl[(size_t)p * B + b];
        for (int b = 1; b < B; b++)
          if (memcmp(&rowmajor[(size_t)(b - 1) * W], &rowmajor[(size_t)b * W], (size_t)W) > 0) throw CometError("ShuffleWriter: range boundaries are not in ascending order");
      }
      bkeys.ensure(rowmajor.size() + 16);
      HIP_CHECK(hipMemcpyAsync(bkeys.p, rowmajor.data(), rowmajor.size(), hipMemcpyHostToDevice, stream_));
      if (comet_launch_range_partition_ids((const uint8_t*)planes->p, n, W, (const uint8_t*)bkeys.p, B, (int32_t*)pids.p, stream_) != 0)
        throw CometError("shuffle: launch failed");
      HIP_CHECK(hipStreamSynchronize(stream_));   // rowmajor (pageable) must outlive the upload
    }
    const uint32_t seed = 42;
    if (!by_range && comet_launch_fill(4, hashes.p, n, &seed, stream_) != 0) throw CometError("shuffle: launch failed");
    for (int c : key_cols) {
      const DeviceColumnView& v = in.cols[(size_t)c];
      if (v.offset != 0) throw CometError("ShuffleWriter: hash key column with a non-zero Arrow offset is not supported yet");
      if (comet_launch_murmur3((int)in.types[(size_t)c].id, in.types[(size_t)c].precision, v.data, in.has_valid[(size_t)c] ? v.valid : nullptr, v.aux, n,
                               (uint32_t*)hashes.p, stream_) != 0)
        throw CometError("ShuffleWriter: cannot hash a column of type " + in.types[(size_t)c].str());
    }
    const int64_t W = comet_partition_tiles(n);
    const size_t hist_bytes = ((size_t)P * (size_t)W + 1) * 8;
    hist.ensure((size_t)comet_partition_scratch_bytes(n, P));
    uint32_t* bad = (uint32_t*)((char*)hist.p + hist_bytes);
    HIP_CHECK(hipMemsetAsync(bad, 0, 4, stream_));
    if ((!by_range && comet_launch_pmod((const uint32_t*)hashes.p, n, P, (int32_t*)pids.p, stream_) != 0) ||
        comet_launch_partition_indices((const int32_t*)pids.p, n, P, (uint64_t*)hist.p, bad, (int64_t*)dstarts.p, (uint32_t*)ridx->p, stream_) != 0)
      throw CometError("shuffle: launch failed");
    HIP_CHECK(hipMemcpyAsync(starts.data(), dstarts.p, ((size_t)P + 1) * 8, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
    DevTable payload = in;
    payload.types.resize(n_payload);
    payload.cols.resize(n_payload);
    payload.has_valid.resize(n_payload);
    grouped = take_rows(payload, (const uint32_t*)ridx->p, 0, n, ridx);
  } else {
    grouped = in;
    grouped.types.resize(n_payload);
    grouped.cols.resize(n_payload);
    grouped.has_valid.resize(n_payload);
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  shuffle_repart_ns_ += tm.ns();
  lap("partition (murmur3, pmod, indices, takes)");

  // one download of the partition-major table
  std::vector<std::unique_ptr<PinnedBuf>> hv(n_payload), hb(n_payload), hd(n_payload);
  for (size_t j = 0; j < n_payload && n > 0; j++) {
    const DType& ty = grouped.types[j];
    const DeviceColumnView& v = grouped.cols[j];
    if (v.offset != 0) throw CometError("ShuffleWriter: input column with a non-zero Arrow offset is not supported yet");
    const bool is_str = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    const size_t bytes = is_str ? (size_t)(n + 1) * 4 : ty.id == TypeId::Bool ? (size_t)((n + 7) / 8) : (size_t)n * fixed_width(ty);
    hv[j].reset(new PinnedBuf());
    hv[j]->ensure(bytes + 8);
    HIP_CHECK(hipMemcpyAsync(hv[j]->p, v.data, bytes, hipMemcpyDeviceToHost, stream_));
    if (grouped.has_valid[j]) {
      hb[j].reset(new PinnedBuf());
      hb[j]->ensure((size_t)((n + 7) / 8) + 8);
      HIP_CHECK(hipMemcpyAsync(hb[j]->p, v.valid, (size_t)((n + 7) / 8), hipMemcpyDeviceToHost, stream_));
    }
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  for (size_t j = 0; j < n_payload && n > 0; j++) {
    const DType& ty = grouped.types[j];
    if (ty.id != TypeId::String && ty.id != TypeId::Bytes) continue;
    const int32_t* offs = (const int32_t*)hv[j]->p;
    if (offs[0] != 0) throw CometError("ShuffleWriter: Utf8 column whose offsets do not start at 0");
    hd[j].reset(new PinnedBuf());
    hd[j]->ensure((size_t)offs[n] + 8);
    if (offs[n]) HIP_CHECK(hipMemcpyAsync(hd[j]->p, grouped.cols[j].aux, (size_t)offs[n], hipMemcpyDeviceToHost, stream_));
  }
  HIP_CHECK(hipStreamSynchronize(stream_));

  lap("download");
  // frame every partition on the host threads
  const int64_t bs = batch_size_ > 0 ? batch_size_ : std::max<int64_t>(n, 1);
  const ShuffleCodec codec = (ShuffleCodec)sw.shuffle_codec;
  const double write_t0 = tm.ns();
  // Blocks in file order — (partition, first row, rows) — grouped into runs of consecutive blocks of ≈4 MiB of column data.  The
  // scan threads encode whole runs (one output buffer per run, allocated once); this thread writes finished runs to the data
  // file in order while later runs are still being encoded.
  struct BlockTask { int p; int64_t first, rows; };
  std::vector<BlockTask> tasks;
  for (int p = 0; p < P; p++)
    for (int64_t r = starts[(size_t)p]; r < starts[(size_t)p + 1]; r += bs) tasks.push_back({p, r, std::min(bs, starts[(size_t)p + 1] - r)});
  size_t row_bytes = 0;
  for (size_t j = 0; j < n_payload; j++) {
    const DType& ty = grouped.types[j];
    const bool is_str = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    row_bytes += is_str ? 4 + (n > 0 ? (size_t)(((const int32_t*)hv[j]->p)[n] / n) + 1 : 0) : ty.id == TypeId::Bool ? 1 : (size_t)fixed_width(ty);
    if (hb[j]) row_bytes += 1;
  }
  struct Run {
    size_t first = 0, last = 0;      // tasks [first, last)
    std::vector<uint8_t> bytes;
    std::vector<size_t> block_size;  // per task
    std::string error;
    bool done = false;
    double encode_ms = 0;
  };
  std::vector<Run> runs;
  for (size_t t = 0; t < tasks.size();) {
    size_t e = t, acc = 0;
    while (e < tasks.size() && acc < (size_t)(4 << 20)) acc += (size_t)tasks[e++].rows * std::max<size_t>(row_bytes, 1);
    Run r;
    r.first = t;
    r.last = e;
    runs.push_back(std::move(r));
    t = e;
  }
  std::mutex mu;
  std::condition_variable cv;
  for (size_t ri = 0; ri < runs.size(); ri++) {
    scan_pool_submit([&, ri]() {
      Run& r = runs[ri];
      Timer rt;
      try {
        size_t est = 0;
        for (size_t t = r.first; t < r.last; t++) est += (size_t)tasks[t].rows * row_bytes + 2048;
        r.bytes.reserve(est + est / 8 + (64 << 10));
        std::vector<ColumnSlice> cols(n_payload);
        for (size_t j = 0; j < n_payload; j++) {
          cols[j].type = grouped.types[j];
          cols[j].validity = hb[j] ? (const uint8_t*)hb[j]->p : nullptr;
          cols[j].values = hv[j]->p;
          cols[j].data = hd[j] ? (const uint8_t*)hd[j]->p : nullptr;
        }
        for (size_t t = r.first; t < r.last; t++) {
          for (auto& c : cols) c.first = tasks[t].first;
          r.block_size.push_back(encode_shuffle_block(cols, tasks[t].rows, codec, sw.shuffle_compression_level, r.bytes));
        }
      } catch (const std::exception& e) {
        r.error = e.what();
      } catch (...) {
        r.error = "shuffle writer: unknown error while encoding a block";
      }
      r.encode_ms = rt.ns() / 1e6;
      {
        std::lock_guard<std::mutex> lk(mu);
        r.done = true;
      }
      cv.notify_all();
    });
  }
  const int fd = open(sw.shuffle_data_file.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  std::string failure;
  if (fd < 0) failure = "shuffle write error: cannot create " + sw.shuffle_data_file + ": " + strerror(errno);
  std::vector<int64_t> offsets((size_t)P + 1, 0);
  int64_t file_pos = 0;
  int next_p = 0;
  double wait_ms = 0, write_ms = 0, enc_sum = 0, enc_max = 0;
  for (size_t ri = 0; ri < runs.size(); ri++) {   // every run is waited for, also after a failure: the tasks reference this frame
    Run& r = runs[ri];
    {
      Timer wt;
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return r.done; });
      wait_ms += wt.ns() / 1e6;
    }
    enc_sum += r.encode_ms;
    enc_max = std::max(enc_max, r.encode_ms);
    Timer wrt;
    if (failure.empty() && !r.error.empty()) failure = r.error;
    if (!failure.empty()) continue;
    int64_t pos = file_pos;
    for (size_t t = r.first; t < r.last; t++) {
      while (next_p <= tasks[t].p) offsets[(size_t)next_p++] = pos;
      pos += (int64_t)r.block_size[t - r.first];
    }
    size_t done = 0;
    while (done < r.bytes.size()) {
      const ssize_t w = write(fd, r.bytes.data() + done, r.bytes.size() - done);
      if (w <= 0) {
        failure = "shuffle write error: " + std::string(strerror(errno)) + " (" + sw.shuffle_data_file + ")";
        break;
      }
      done += (size_t)w;
    }
    file_pos = pos;
    std::vector<uint8_t>().swap(r.bytes);
    write_ms += wrt.ns() / 1e6;
  }
  if (trace)
    fprintf(stderr, "[comet] shuffle write: %zu runs, encode cpu %.1f ms total (max %.2f ms/run), writer waited %.1f ms, wrote for %.1f ms\n", runs.size(),
            enc_sum, enc_max, wait_ms, write_ms);
  while (next_p <= P) offsets[(size_t)next_p++] = file_pos;
  if (fd >= 0 && close(fd) != 0 && failure.empty()) failure = "shuffle write error: closing " + sw.shuffle_data_file + " failed";
  if (!failure.empty()) throw CometError(failure);
  lap("encode blocks + write data file (overlapped)");
  FILE* xf = fopen(sw.shuffle_index_file.c_str(), "wb");
  if (!xf) throw CometError("shuffle write error: cannot create " + sw.shuffle_index_file + ": " + strerror(errno));
  const bool ok = fwrite(offsets.data(), 8, offsets.size(), xf) == offsets.size();
  if (fclose(xf) != 0 || !ok) throw CometError("shuffle write error: writing " + sw.shuffle_index_file + " failed");
  shuffle_bytes_written_ += offsets[(size_t)P];
  shuffle_write_ns_ += tm.ns() - write_t0;
  shuffle_data_size_ += (int64_t)row_bytes * n;
  lap("write files");
  DevTable none;
  return none;
}

// rows [first, first + rows) of `in` in the order given by dev_perm (nullptr = identity) → a new resident table
DevTable ExecutionContext::take_rows(const DevTable& in, const uint32_t* dev_perm, int64_t first, int64_t rows, std::shared_ptr<DevBuf> perm_owner) {
  DevTable out;
  out.rows = rows;
  out.types = in.types;
  out.has_valid = in.has_valid;
  out.cols.assign(in.cols.size(), DeviceColumnView());
  std::shared_ptr<DevBuf> perm = perm_owner;
  if (!dev_perm) {
    perm = std::make_shared<DevBuf>();
    perm->ensure((size_t)std::max<int64_t>(rows, 1) * 4);
    if (comet_launch_sort_iota((uint32_t*)perm->p, rows, (uint32_t)first, stream_) != 0) throw CometError("limit: launch failed");
    dev_perm = (const uint32_t*)perm->p;
    first = 0;
  }
  const uint32_t* idx = dev_perm + first;
  for (size_t c = 0; c < in.cols.size(); c++) {
    const DType& t = in.types[c];
    if (t.id == TypeId::String || t.id == TypeId::Bytes) {
      if (in.cols[c].offset != 0 && in.has_valid[c]) throw CometError("Sort / Limit over a nullable Utf8 column with a non-zero Arrow offset is not supported yet");
      take_utf8(in.cols[c], idx, nullptr, in.has_valid[c] ? in.cols[c].valid : nullptr, rows, out.cols[c], out.owners);
      if (in.has_valid[c]) {
        auto bm = std::make_shared<DevBuf>();
        bm->ensure((size_t)((rows + 7) / 8) + 16);
        if (rows && comet_launch_take(0, in.cols[c].valid, idx, rows, bm->p, stream_) != 0) throw CometError("take: validity");
        out.cols[c].valid = (const uint8_t*)bm->p;
        out.owners.push_back(bm);
      }
      continue;
    }
    if (in.cols[c].offset != 0) throw CometError("Sort / Limit over a column with a non-zero Arrow offset is not supported yet");
    const int w = t.id == TypeId::Bool ? 0 : fixed_width(t);
    auto vals = std::make_shared<DevBuf>();
    vals->ensure((w ? (size_t)std::max<int64_t>(rows, 1) * w : (size_t)((rows + 7) / 8)) + 16);
    if (rows && comet_launch_take(w, in.cols[c].data, idx, rows, vals->p, stream_) != 0) throw CometError("take: unsupported width");
    out.cols[c].data = vals->p;
    out.owners.push_back(vals);
    if (in.has_valid[c]) {
      auto bm = std::make_shared<DevBuf>();
      bm->ensure((size_t)((rows + 7) / 8) + 16);
      if (rows && comet_launch_take(0, in.cols[c].valid, idx, rows, bm->p, stream_) != 0) throw CometError("take: validity");
      out.cols[c].valid = (const uint8_t*)bm->p;
      out.owners.push_back(bm);
    }
  }
  HIP_CHECK(hipStreamSynchronize(stream_));   // `in` (and the permutation) may be released by the caller
  return out;
}

// Sort (planner.rs:1488-1522 → SortExec with fetch / skip): order-preserving key bytes per row (generated kernel), LSD radix
// sort of a row permutation over the byte planes that actually vary, then one take per column of rows [skip, skip+fetch).
// order-preserving key bytes of every row of `in` under sop.sort_orders, as W byte planes of n rows (plane p of row i at p·n + i)
std::shared_ptr<DevBuf> ExecutionContext::sort_key_planes(const Operator& sop, const DevTable& in, int& W, std::vector<int64_t>* str_len, bool measure_only) {
  const int64_t n = in.rows;
  std::string key = std::to_string(plan_hash_ ^ (0x9E3779B97F4A7C15ull * (uint64_t)(node_id_[&sop] + 1))) + ":S:" + validity_key(in.has_valid);
  std::shared_ptr<PlannedVariant> pv;
  {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    if (it != g_plan_cache.end()) pv = it->second;
  }
  if (!pv) {
    pv = std::make_shared<PlannedVariant>();
    pv->desc = generate_sort_keys(sop, in.types, in.has_valid);
    pv->code = jit_compile(pv->desc.source);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plan_cache[key] = pv;
  }
  Variant v;
  v.desc = pv->desc;
  v.mod = jit_load(pv->code);
  W = v.desc.sort_key_bytes;
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  // Utf8 sort keys: padded to the longest value of the column (measured here, or imposed by the caller when two tables must share
  // one key layout — range-partition boundaries)
  std::vector<int64_t> lens;
  for (size_t s = 0; s < v.desc.sort_str_cols.size(); s++) {
    int64_t L = 0;
    if (str_len && s < str_len->size() && !measure_only) L = (*str_len)[s];
    else if (n > 0) {
      const DeviceColumnView& sc = in.cols[(size_t)v.desc.sort_str_cols[s]];
      uint32_t* mx = (uint32_t*)err_flags_.p + (kErrBytes / 4 - 1);
      HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
      if (comet_launch_str_max_len((const int32_t*)sc.data + sc.offset, n, mx, stream_) != 0) throw CometError("sort: launch failed");
      uint32_t longest = 0;
      read_small(&longest, mx, 4);
      HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
      L = longest;
    }
    lens.push_back(L);
    prm.iarg[1 + s] = L + 4;
    W += (int)(L + 4);
  }
  if (str_len) *str_len = lens;
  if (measure_only) return nullptr;
  if (W > 1000) throw CometError("Sort key wider than 1000 bytes (Utf8 sort keys are padded to their longest value)");
  auto planes = std::make_shared<DevBuf>();
  planes->ensure((size_t)W * (size_t)std::max<int64_t>(n, 1) + 16);
  if (n == 0) return planes;
  prm.n = n;
  for (size_t i = 0; i < in.cols.size(); i++) {
    prm.in[i].data = in.cols[i].data;
    prm.in[i].valid = in.has_valid[i] ? in.cols[i].valid : nullptr;
    prm.in[i].aux = in.cols[i].aux;
    prm.in[i].offset = in.cols[i].offset;
  }
  prm.out[0] = planes->p;
  prm.out[kOutErr] = err_flags_.p;
  launch(v, "k_sortkey", (int)std::min<int64_t>((n + 255) / 256, 256 * 8), prm);
  planes_owner_ = v.mod;   // the module must stay loaded until the launch has run; callers synchronise before returning
  return planes;
}

// a small resident table from literal rows (range-partition boundaries): one column per entry of `types`
DevTable ExecutionContext::literal_table(const std::vector<std::vector<ExprP>>& rows, const std::vector<DType>& types) {
  DevTable t;
  const int64_t n = (int64_t)rows.size();
  t.rows = n;
  for (size_t c = 0; c < types.size(); c++) {
    const DType& ty = types[c];
    const bool is_str = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    std::vector<uint8_t> vals, data, valid((size_t)((n + 7) / 8) + 1, 0);
    std::vector<int32_t> offs(1, 0);
    const int w = is_str ? 0 : ty.id == TypeId::Bool ? 0 : fixed_width(ty);
    if (ty.id == TypeId::Bool) vals.assign((size_t)((n + 7) / 8) + 1, 0);
    for (int64_t r = 0; r < n; r++) {
      const Expr& e = *rows[(size_t)r][c];
      if (!e.lit_null) valid[(size_t)(r >> 3)] |= (uint8_t)(1u << (r & 7));
      if (is_str) {
        if (!e.lit_null) data.insert(data.end(), e.lit_bytes.begin(), e.lit_bytes.end());
        offs.push_back((int32_t)data.size());
      } else if (ty.id == TypeId::Bool) {
        if (!e.lit_null && e.lit_bool) vals[(size_t)(r >> 3)] |= (uint8_t)(1u << (r & 7));
      } else {
        uint8_t buf[16] = {0};
        if (!e.lit_null) {
          if (ty.id == TypeId::Decimal) { i128 v = e.lit_dec; memcpy(buf, &v, 16); }
          else if (ty.id == TypeId::Double) { double v = e.lit_f64; memcpy(buf, &v, 8); }
          else if (ty.id == TypeId::Float) { float v = (float)e.lit_f64; memcpy(buf, &v, 4); }
          else { int64_t v = e.lit_i64; memcpy(buf, &v, 8); }   // little endian: the low `w` bytes are the narrower integer
        }
        vals.insert(vals.end(), buf, buf + w);
      }
    }
    auto up = [&](const void* p, size_t bytes) {
      auto b = std::make_shared<DevBuf>();
      b->ensure(bytes + 16);
      if (bytes) HIP_CHECK(hipMemcpy(b->p, p, bytes, hipMemcpyHostToDevice));
      t.owners.push_back(b);
      return b->p;
    };
    DeviceColumnView v;
    if (is_str) {
      v.data = up(offs.data(), offs.size() * 4);
      v.aux = up(data.data(), data.size());
    } else {
      v.data = up(vals.data(), vals.size());
    }
    v.valid = (const uint8_t*)up(valid.data(), valid.size());
    t.types.push_back(ty);
    t.cols.push_back(v);
    t.has_valid.push_back(true);
  }
  return t;
}

DevTable ExecutionContext::sort_table(const Operator& sop, const DevTable& in) {
  const int64_t n = in.rows;
  if (n >= ((int64_t)1 << 32)) throw CometError("Sort: more than 2^32 rows in one partition");
  const int64_t skip = std::min<int64_t>(std::max(0, sop.skip), n);
  const int64_t keep = sop.fetch >= 0 ? std::min<int64_t>(n, sop.fetch) : n;     // fetch counts from the first row (GlobalLimit(skip) on top)
  const int64_t out_rows = std::max<int64_t>(0, keep - skip);
  if (n == 0 || out_rows == 0) return take_rows(in, nullptr, 0, 0, nullptr);
  timed_begin();
  int W = 0;
  auto planes = sort_key_planes(sop, in, W);
  // which planes vary at all?
  DevBuf flags;
  flags.ensure((size_t)W * 4 + 16);
  HIP_CHECK(hipMemsetAsync(flags.p, 0, (size_t)W * 4, stream_));
  if (comet_launch_sort_plane_varies((const uint8_t*)planes->p, n, W, (uint32_t*)flags.p, stream_) != 0) throw CometError("sort: launch failed");
  std::vector<uint32_t> varies((size_t)W);
  small_host_.ensure(std::max<size_t>(4096, (size_t)W * 4));
  HIP_CHECK(hipMemcpyAsync(small_host_.p, flags.p, (size_t)W * 4, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
  memcpy(varies.data(), small_host_.p, (size_t)W * 4);
  auto perm = std::make_shared<DevBuf>();
  auto perm2 = std::make_shared<DevBuf>();
  perm->ensure((size_t)n * 4 + 16);
  perm2->ensure((size_t)n * 4 + 16);
  if (comet_launch_sort_iota((uint32_t*)perm->p, n, 0, stream_) != 0) throw CometError("sort: launch failed");
  int64_t ns = n;   // rows that take part in the full sort
  int select_passes = 0;
  if (sop.fetch >= 0 && keep * 8 < n) {
    // TopK: radix select from the most significant varying plane down.  `sure` rows are certainly among the first `keep`;
    // only the bucket that straddles the K-th position stays a candidate.  What is left (sure ∪ candidates) is sorted.
    auto sure = std::make_shared<DevBuf>();
    sure->ensure((size_t)n * 4 + 16);
    DevBuf sel;   // [0..255] u64 histogram, then two u32 counters
    sel.ensure(256 * 8 + 16);
    HIP_CHECK(hipMemsetAsync((char*)sel.p + 256 * 8, 0, 8, stream_));
    uint32_t* counters = (uint32_t*)((char*)sel.p + 256 * 8);
    int64_t m = n, need = keep, nsure = 0;
    for (int b = 0; b < W && m > std::max<int64_t>(4096, need); b++) {
      if (!varies[(size_t)b]) continue;
      const uint8_t* plane = (const uint8_t*)planes->p + (size_t)b * (size_t)n;
      HIP_CHECK(hipMemsetAsync(sel.p, 0, 256 * 8, stream_));
      if (comet_launch_sort_hist256(plane, (const uint32_t*)perm->p, m, (uint64_t*)sel.p, stream_) != 0) throw CometError("sort: launch failed");
      uint64_t h[256];
      HIP_CHECK(hipMemcpyAsync(small_host_.p, sel.p, 256 * 8, hipMemcpyDeviceToHost, stream_));
      HIP_CHECK(hipStreamSynchronize(stream_));
      memcpy(h, small_host_.p, sizeof h);
      int dstar = 255;
      int64_t below = 0;
      for (int dgt = 0; dgt < 256; dgt++) {
        if (below + (int64_t)h[dgt] >= need) { dstar = dgt; break; }
        below += (int64_t)h[dgt];
      }
      uint32_t cnt2[2] = {(uint32_t)nsure, 0};
      write_small(counters, cnt2, 8);
      if (comet_launch_sort_select(plane, (const uint32_t*)perm->p, m, dstar, (uint32_t*)sure->p, (uint32_t*)perm2->p, counters, stream_) != 0)
        throw CometError("sort: launch failed");
      std::swap(perm, perm2);
      nsure += below;
      need -= below;
      m = (int64_t)h[dstar];
      select_passes++;
    }
    // rows to sort = sure ++ remaining candidates
    if (select_passes) {
      HIP_CHECK(hipMemcpyAsync((char*)sure->p + (size_t)nsure * 4, perm->p, (size_t)m * 4, hipMemcpyDeviceToDevice, stream_));
      HIP_CHECK(hipStreamSynchronize(stream_));
      perm = sure;
      ns = nsure + m;
    }
  }
  DevBuf digit, ridx, hist, starts;
  digit.ensure((size_t)ns * 4 + 16);
  ridx.ensure((size_t)ns * 4 + 16);
  const int64_t Wt = comet_partition_tiles(ns);
  hist.ensure((size_t)comet_partition_scratch_bytes(ns, 256));
  starts.ensure(257 * 8);
  uint32_t* bad = (uint32_t*)((char*)hist.p + ((size_t)256 * (size_t)Wt + 1) * 8);
  HIP_CHECK(hipMemsetAsync(bad, 0, 4, stream_));
  int passes = 0;
  for (int b = W - 1; b >= 0; b--) {
    if (!varies[(size_t)b]) continue;
    const uint8_t* plane = (const uint8_t*)planes->p + (size_t)b * (size_t)n;
    if (comet_launch_sort_gather_digit(plane, (const uint32_t*)perm->p, ns, (int32_t*)digit.p, stream_) != 0 ||
        comet_launch_partition_indices((const int32_t*)digit.p, ns, 256, (uint64_t*)hist.p, bad, (int64_t*)starts.p, (uint32_t*)ridx.p, stream_) != 0 ||
        comet_launch_take(4, perm->p, (const uint32_t*)ridx.p, ns, perm2->p, stream_) != 0)
      throw CometError("sort: launch failed");
    std::swap(perm, perm2);
    passes++;
  }
  timed_end();
  if (getenv("COMET_TRACE_STAGES"))
    fprintf(stderr, "[comet] sort: %lld rows, key %d bytes, %d select passes -> %lld rows sorted in %d radix passes\n", (long long)n, W, select_passes,
            (long long)ns, passes);
  DevTable out = take_rows(in, (const uint32_t*)perm->p, skip, out_rows, perm);   // synchronises the stream: the key kernel has run
  return out;
}

// An aggregate below other operators (Sort / Project / Filter / join over a HashAggregate): it runs as its own execution
// context over the input streams of its sub-tree, and its grouped result is handed over resident in HBM.
DevTable ExecutionContext::nested_aggregate(const Operator& agg) {
  // the Scan leaves of the sub-tree, in depth-first order, are a contiguous range of this context's inputs
  std::vector<size_t> idx;
  std::function<void(const Operator&)> walk = [&](const Operator& op) {
    if (op.kind == OpKind::Scan) idx.push_back(scan_input_.at(&op));
    for (auto& c : op.children) walk(*c);
  };
  walk(agg);
  std::vector<InputSource> sub_inputs;
  for (size_t i : idx) {
    sub_inputs.push_back(inputs_[i]);
    inputs_[i].host = nullptr;      // ownership moves to the sub-context (it releases the streams)
    inputs_[i].dev = nullptr;
    inputs_[i].exhausted = true;
  }
  // the sub-plan shares the Operator nodes: wrap the node in a non-owning shared_ptr
  OperatorP sub_plan(const_cast<Operator*>(&agg), [](Operator*) {});
  ExecutionContext sub(sub_plan, plan_hash_ ^ (0x9E3779B97F4A7C15ull * (uint64_t)(node_id_[&agg] + 1)), config_, sub_inputs, 0, device_id_);
  if (sub.sink_ != SinkKind::AggGrouped && sub.sink_ != SinkKind::AggNoGroup) throw CometError("internal: nested aggregate without an aggregate sink");
  sub.device_result_ = true;
  sub.start();
  sub.run_to_completion();
  DevTable t;
  if (sub.sink_ == SinkKind::AggNoGroup) {
    // an ungrouped aggregate yields exactly one row (TPC-H Q14 / Q17 / Q19 compute on it): finish it the usual way, then put that row
    // back into HBM for the operators above
    sub.finish_aggregate();
    if (sub.ready_.empty()) throw CometError("internal: ungrouped aggregate produced no row");
    t = host_batch_to_table(sub.ready_.front());
    sub.ready_.clear();
  } else {
    t = sub.grouped_to_device();
  }
  input_rows += sub.input_rows;
  sub.collect_timings();
  last_kernel_ms += sub.last_kernel_ms;
  last_kernel_launches += sub.last_kernel_launches;
  return t;
}

// a (small) host batch → resident table
DevTable ExecutionContext::host_batch_to_table(const HostBatch& b) {
  DevTable t;
  t.rows = b.rows;
  auto up = [&](const std::vector<uint8_t>& v) -> const void* {
    auto d = std::make_shared<DevBuf>();
    d->ensure(v.size() + 16);
    if (!v.empty()) HIP_CHECK(hipMemcpy(d->p, v.data(), v.size(), hipMemcpyHostToDevice));
    t.owners.push_back(d);
    return d->p;
  };
  for (const HostColumn& c : b.cols) {
    DeviceColumnView v;
    v.data = up(c.values);
    const bool is_str = c.type.id == TypeId::String || c.type.id == TypeId::Bytes;
    if (is_str) v.aux = up(c.data);
    const bool hv = c.null_count > 0 && !c.validity.empty();
    if (hv) v.valid = (const uint8_t*)up(c.validity);
    t.types.push_back(c.type);
    t.cols.push_back(v);
    t.has_valid.push_back(hv);
  }
  return t;
}

// resident table → host batches of ≤ batch_size rows (root of a plan that ends in a join)
void ExecutionContext::table_to_host_batches(const DevTable& t) {
  if (t.rows == 0) return;
  const size_t ncol = t.cols.size();
  std::vector<std::vector<uint8_t>> hv(ncol), hb(ncol);
  std::vector<std::vector<uint8_t>> hd(ncol);
  for (size_t j = 0; j < ncol; j++) {
    const DType& ty = t.types[j];
    const bool is_str = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    size_t bytes = is_str ? (size_t)(t.rows + 1) * 4 : ty.id == TypeId::Bool ? (size_t)((t.rows + 7) / 8) : (size_t)t.rows * fixed_width(ty);
    hv[j].resize(bytes);
    HIP_CHECK(hipMemcpyAsync(hv[j].data(), t.cols[j].data, bytes, hipMemcpyDeviceToHost, stream_));
    if (t.has_valid[j]) {
      hb[j].resize((size_t)((t.rows + 7) / 8));
      HIP_CHECK(hipMemcpyAsync(hb[j].data(), t.cols[j].valid, hb[j].size(), hipMemcpyDeviceToHost, stream_));
    }
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  for (size_t j = 0; j < ncol; j++) {
    if (t.types[j].id == TypeId::String || t.types[j].id == TypeId::Bytes) {
      const int32_t* offs = (const int32_t*)hv[j].data();
      hd[j].resize((size_t)offs[t.rows]);
      if (!hd[j].empty()) HIP_CHECK(hipMemcpyAsync(hd[j].data(), t.cols[j].aux, hd[j].size(), hipMemcpyDeviceToHost, stream_));
    }
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  const int64_t bs = batch_size_ > 0 ? batch_size_ : t.rows;
  auto getbit = [](const std::vector<uint8_t>& b, int64_t i) { return (b[(size_t)(i >> 3)] >> (i & 7)) & 1; };
  for (int64_t off = 0; off < t.rows; off += bs) {
    const int64_t len = std::min(bs, t.rows - off);
    HostBatch b;
    b.rows = len;
    for (size_t j = 0; j < ncol; j++) {
      HostColumn c;
      c.type = t.types[j];
      c.length = len;
      if (c.type.id == TypeId::String || c.type.id == TypeId::Bytes) {
        const int32_t* offs = (const int32_t*)hv[j].data();
        c.values.resize((size_t)(len + 1) * 4);
        int32_t* o = (int32_t*)c.values.data();
        for (int64_t i = 0; i <= len; i++) o[i] = offs[off + i] - offs[off];
        c.data.assign(hd[j].begin() + offs[off], hd[j].begin() + offs[off + len]);
      } else if (c.type.id == TypeId::Bool) {
        c.values.assign((size_t)((len + 7) / 8), 0);
        for (int64_t i = 0; i < len; i++)
          if (getbit(hv[j], off + i)) c.values[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
      } else {
        int w = fixed_width(c.type);
        c.values.assign(hv[j].begin() + (size_t)off * w, hv[j].begin() + (size_t)(off + len) * w);
      }
      if (t.has_valid[j]) {
        int64_t nulls = 0;
        std::vector<uint8_t> bm((size_t)((len + 7) / 8), 0);
        for (int64_t i = 0; i < len; i++) {
          if (getbit(hb[j], off + i)) bm[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
          else nulls++;
        }
        c.null_count = nulls;
        if (nulls) c.validity = std::move(bm);
      }
      b.cols.push_back(std::move(c));
    }
    ready_.push_back(std::move(b));
  }
}

void ExecutionContext::run_to_completion() {
  if (has_join_) {
    DevTable src = materialize(*root_source_);
    if (plan_.get() == root_source_) throw CometError("internal: bare join root");
    if (sink_ == SinkKind::AggGrouped) prepare_dict_keys(src);
    process_chunk(src.cols, src.has_valid, src.rows);
    HIP_CHECK(hipStreamSynchronize(stream_));
    return;
  }
  // ungrouped / grouped aggregates are pipeline breakers: drain the input completely
  while (true) {
    bool more = inputs_[0].kind == 0 ? pull_host_chunk() : pull_device_batch();
    if (!more) break;
  }
}

void ExecutionContext::start() {
  if (!started_) {
    // Lazy start like the reference (jni_api.rs:795-872): nothing touches the input before the first executePlan.
    HIP_CHECK(hipSetDevice(device_id_));
    stream_ = pool_get_stream(device_id_);
    err_flags_.ensure(kErrBytes);
    HIP_CHECK(hipMemsetAsync(err_flags_.p, 0, kErrBytes, stream_));
    started_ = true;
  } else {
    HIP_CHECK(hipSetDevice(device_id_));
  }
}

namespace {
struct ExportedDeviceColumn {
  std::vector<std::shared_ptr<void>> owners;   // pooled buffers / producer arrays the pointers live in
  const void* buffers[3];
};
void release_device_array(ArrowArray* a) {
  delete (ExportedDeviceColumn*)a->private_data;
  a->release = nullptr;
}
void release_fmt_schema(ArrowSchema* s) {
  delete (std::string*)s->private_data;
  s->release = nullptr;
}
}  // namespace

// The stage boundary for multi-GPU plans: a Filter/Project/HashJoin plan's output stays resident so that the
// exchange (murmur3 → pmod → partition scatter → RCCL all-to-all) never touches the host.  One batch = the whole result.
void ExecutionContext::set_memory_manager(int64_t (*acquire)(void*, int64_t), void (*release)(void*, int64_t), void* ctx, long long task_id) {
  mem_->acquire = acquire;
  mem_->release = release;
  mem_->ctx = ctx;
  mem_->task_id = task_id;
}
void ExecutionContext::memory_stats(int64_t out[4]) {
  out[0] = mem_->host_used.load();
  out[1] = mem_->host_peak.load();
  out[2] = mem_->dev_used.load();
  out[3] = mem_->dev_peak.load();
}

int64_t ExecutionContext::execute_device(ArrowDeviceArray** out_arrays, ArrowSchema** out_schemas, int n_out) {
  mem_->owner = std::this_thread::get_id();      // Spark may move a task's calls between threads; the up-calls go with the caller
  AccountScope account(mem_);
  mem_->flush();
  Timer t;
  start();
  if (sink_ == SinkKind::AggNoGroup)
    throw CometError("comet_execute_plan_device: an ungrouped aggregate result is one row and is exported through comet_execute_plan");
  if (finished_) return -1;
  DevTable tab;
  if (sink_ == SinkKind::AggGrouped) {
    device_result_ = true;
    run_to_completion();
    tab = grouped_to_device();
  } else {
    tab = materialize(*plan_);
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  collect_timings();
  finished_ = true;
  if ((size_t)n_out != tab.cols.size())
    throw CometError("Output column count mismatch: expected " + std::to_string(n_out) + ", got " + std::to_string(tab.cols.size()));
  for (int j = 0; j < n_out; j++) {
    const DType& ty = tab.types[(size_t)j];
    const bool is_str = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    if (tab.cols[(size_t)j].offset != 0) throw CometError("comet_execute_plan_device: input column with a non-zero Arrow offset passed through");
    auto* ec = new ExportedDeviceColumn();
    ec->owners = tab.owners;
    ec->buffers[0] = tab.has_valid[(size_t)j] ? tab.cols[(size_t)j].valid : nullptr;
    ec->buffers[1] = tab.cols[(size_t)j].data;
    ec->buffers[2] = is_str ? tab.cols[(size_t)j].aux : nullptr;
    ArrowDeviceArray* d = out_arrays[j];
    memset(d, 0, sizeof *d);
    d->array.length = tab.rows;
    d->array.null_count = tab.has_valid[(size_t)j] ? -1 : 0;
    d->array.n_buffers = is_str ? 3 : 2;
    d->array.buffers = ec->buffers;
    d->array.private_data = ec;
    d->array.release = release_device_array;
    d->device_id = device_id_;
    d->device_type = ARROW_DEVICE_ROCM;
    d->sync_event = nullptr;   // the plan's stream was synchronised above
    ArrowSchema* s = out_schemas[j];
    memset(s, 0, sizeof *s);
    auto* fmt = new std::string(expected_format(ty));
    s->format = fmt->c_str();
    s->name = "";
    s->flags = ARROW_FLAG_NULLABLE;
    s->private_data = fmt;
    s->release = release_fmt_schema;
  }
  output_rows_ += tab.rows;
  elapsed_compute_ns_ += t.ns();
  return tab.rows;
}

int64_t ExecutionContext::execute(ArrowArray** out_arrays, ArrowSchema** out_schemas, int n_out) {
  mem_->owner = std::this_thread::get_id();
  AccountScope account(mem_);
  mem_->flush();
  Timer t;
  start();
  const bool is_agg = sink_ != SinkKind::Output;
  if (!finished_) {
    if (is_agg) {
      run_to_completion();
      if (sink_ == SinkKind::AggGrouped) finish_grouped();
      else finish_aggregate();
      finished_ = true;
    } else {
      if (has_join_ || materialize_root_) {
        // a plan over materialised sources (joins, Parquet scans, sorts, limits, nested aggregates) or one that passes Utf8
        // columns through: the whole result is produced resident in HBM, then copied out in batches
        DevTable t = materialize(*plan_);
        HIP_CHECK(hipStreamSynchronize(stream_));
        table_to_host_batches(t);
        finished_ = true;
      }
      while (ready_.empty() && !finished_) {
        bool more = inputs_[0].kind == 0 ? pull_host_chunk() : pull_device_batch();
        if (!more) finished_ = true;
      }
    }
  }
  elapsed_compute_ns_ += t.ns();
  if (ready_.empty()) return -1;
  HostBatch b = std::move(ready_.front());
  ready_.pop_front();
  export_batch(b, out_arrays, out_schemas, n_out);
  output_rows_ += b.rows;
  return b.rows;
}

// ---------------------------------------------------------------------------------------------
// Arrow C Data export (prepare_output, jni_api.rs:674-742): one moved ArrowArray + ArrowSchema per
// output column, offset 0, buffers owned by the array until the consumer calls release.
// ---------------------------------------------------------------------------------------------
void ExecutionContext::export_batch(HostBatch& b, ArrowArray** out_arrays, ArrowSchema** out_schemas, int n_out) {
  export_host_batch(b, out_arrays, out_schemas, n_out);
}

std::string ExecutionContext::metrics_proto() {
  // tree mirrors the Operator tree (metrics/utils.rs:30-45); per-node attribution of a fused pipeline:
  // the root carries the measured values, fused children report zero time and their row counts unknown (0).
  std::function<MetricNode(const Operator&, bool)> build = [&](const Operator& op, bool root) {
    MetricNode n;
    n.metrics.emplace_back("output_rows", root ? output_rows_ : 0);
    n.metrics.emplace_back("elapsed_compute", root ? (int64_t)elapsed_compute_ns_ : 0);
    if (op.kind == OpKind::ShuffleWriter) {   // ShufflePartitionerMetrics (native/shuffle/src/metrics.rs:24-71)
      n.metrics.emplace_back("data_size", shuffle_data_size_);
      n.metrics.emplace_back("repart_time", (int64_t)shuffle_repart_ns_);
      n.metrics.emplace_back("write_time", (int64_t)shuffle_write_ns_);
      n.metrics.emplace_back("spill_count", 0);
      n.metrics.emplace_back("spilled_bytes", 0);
    }
    if (op.kind == OpKind::NativeScan) {
      n.metrics.emplace_back("bytes_scanned", bytes_scanned_);
      n.metrics.emplace_back("row_groups_pruned_statistics", row_groups_pruned_);
      n.metrics.emplace_back("pages_decompressed_on_device", pages_inflated_on_device_);
      n.metrics.emplace_back("page_index_rows_pruned", rows_pruned_page_index_);
    }
    for (auto& c : op.children) n.children.push_back(build(*c, false));
    return n;
  };
  return encode_metric_node(build(*plan_, true));
}

}  // namespace comet
