// Device / pinned buffer pools, per-plan memory accounting (MemAccount), stream and event pools.
#include "exec_internal.hpp"

namespace comet {

// An executor runs 8-16 task threads against one GPU, each with a handful of streams (its plan's, a copy stream, two decompression-group
// streams).  ROCm maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues — four by default — and a stream that waits for an event
// holds up whatever shares its queue: sixteen concurrent zstd scans took 80 ms with four queues and 62 ms with sixteen (profiles/r5_executor_shape.md).
// The runtime reads the variable when it initialises (the first HIP call), so loading the library is early enough in a JVM; a value the
// operator has set stays.
__attribute__((constructor)) static void comet_hip_env_defaults() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

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

// what the pools could NOT serve (process-wide): calls and nanoseconds spent in hipMalloc / hipHostMalloc — a miss is a driver call that takes
// a lock every other HIP call of the process waits for (COMET_TRACE_STAGES prints the deltas per scan)
std::atomic<int64_t> g_dev_alloc_calls{0}, g_dev_alloc_ns{0}, g_pinned_alloc_calls{0}, g_pinned_alloc_ns{0};
void pool_miss_counters(int64_t out[4]) {
  out[0] = g_dev_alloc_calls.load(); out[1] = g_dev_alloc_ns.load(); out[2] = g_pinned_alloc_calls.load(); out[3] = g_pinned_alloc_ns.load();
}
namespace {
struct AllocTimer {
  std::atomic<int64_t>&calls, &ns;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  AllocTimer(std::atomic<int64_t>& c, std::atomic<int64_t>& n) : calls(c), ns(n) {}
  ~AllocTimer() { calls.fetch_add(1); ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count()); }
};
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
    AllocTimer alloc_timer(g_dev_alloc_calls, g_dev_alloc_ns);
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
    AllocTimer alloc_timer(g_pinned_alloc_calls, g_pinned_alloc_ns);
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

hipStream_t detail::pool_get_stream(int dev) {
  {
    std::lock_guard<std::mutex> lk(pools().mu);
    auto& v = pools().streams[dev];
    if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
  }
  hipStream_t s;
  HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  return s;
}
void detail::pool_put_stream(int dev, hipStream_t s) {
  std::lock_guard<std::mutex> lk(pools().mu);
  pools().streams[dev].push_back(s);
}
// Plans executing at this moment (executePlan calls in flight): a scan decides by it what it leaves to the device — with many tasks at once
// the device's decompression kernels are the bound and host threads take what they can (parquet_scan.cpp).
namespace {
std::atomic<int> g_plans_executing{0};
}  // namespace
void detail::plan_execution_begins() { g_plans_executing.fetch_add(1); }
void detail::plan_execution_ends() { g_plans_executing.fetch_sub(1); }
int detail::plans_executing() { return g_plans_executing.load(); }

// COMET_PQ_SHARED_COPY_STREAM=1: one host → device copy stream per device for every Parquet scan of the process (the link is one FIFO whoever
// queues the copies, and every stream with pending work wants a hardware queue: GPU_MAX_HW_QUEUES of them, and 32 queues time-slice the
// runlist — 74 ms for a wave that takes 20).  Measured and not the default: one submission that stalls then holds every task's copies.
hipStream_t detail::shared_copy_stream(int dev) {
  std::lock_guard<std::mutex> lk(pools().mu);
  static std::map<int, hipStream_t> shared;
  auto it = shared.find(dev);
  if (it != shared.end()) return it->second;
  hipStream_t s;
  HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  shared[dev] = s;
  return s;
}

hipEvent_t detail::pool_get_event(int dev) {
  {
    std::lock_guard<std::mutex> lk(pools().mu);
    auto& v = pools().events[dev];
    if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; }
  }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}
void detail::pool_put_event(int dev, hipEvent_t e) {
  std::lock_guard<std::mutex> lk(pools().mu);
  pools().events[dev].push_back(e);
}

}  // namespace comet
