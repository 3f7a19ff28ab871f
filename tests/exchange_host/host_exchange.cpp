// TEST INFRASTRUCTURE — not part of libcomet.so, never loaded by the product.
//
// The hash exchange of csrc/exchange_core.hpp (the orchestration the RCCL path runs between GPUs) instantiated over HOST memory so that
// world-size-2 / -4 runs can execute on a box without a GPU: the product's Ops are HIP kernels over HBM buffers (exchange.cpp HipOps);
// this stand-in does the same per-buffer operations with plain loops over malloc'ed buffers.  The wire is the product's TCP transport
// (csrc/exchange_tcp.hpp, the same header libcomet.so compiles).  Partition ids are NOT computed here: the caller passes the ids the
// oracle computed (oracle.hash_partition_ids — the pinned restatement of the reference's murmur3 + pmod), so the only murmur3 on the CPU
// stays the oracle's.  What the tests pin with it: the count exchange, the splits, the order of the collectives, validity-on-any-rank,
// the Utf8 byte split, the offset rebuild, and the transport's behaviour when a peer dies or goes silent.
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../datafusion-comet_amd/csrc/exchange_core.hpp"
#include "../../datafusion-comet_amd/csrc/exchange_rccl.hpp"
#include "../../datafusion-comet_amd/csrc/exchange_tcp.hpp"

using namespace comet::xchg;

namespace {

struct HostBufT {
  void* p = nullptr;
  size_t cap = 0;
  void ensure(size_t n) {
    if (n <= cap) return;
    free(p);
    p = calloc(n, 1);
    if (!p) throw Error("out of memory");
    cap = n;
  }
  ~HostBufT() { free(p); }
  HostBufT() = default;
  HostBufT(const HostBufT&) = delete;
  HostBufT& operator=(const HostBufT&) = delete;
};

struct HostOps {
  using Buf = HostBufT;
  using HostBuf = HostBufT;
  static constexpr bool kDeviceMemory = false;
  const int32_t* oracle_pids = nullptr;      // this rank's rows → partition, from oracle.hash_partition_ids
  void fill_u32(uint32_t* dst, int64_t n, uint32_t v) { for (int64_t i = 0; i < n; i++) dst[i] = v; }
  void murmur3(const CometExchangeColumn&, int64_t, uint32_t*) {}                       // the oracle's ids are used instead
  void pmod(const uint32_t*, int64_t rows, int world, int32_t* pids) {
    for (int64_t i = 0; i < rows; i++) {
      if (oracle_pids[i] < 0 || oracle_pids[i] >= world) throw Error("partition id out of range");
      pids[i] = oracle_pids[i];
    }
  }
  // multi_partition.rs:54-103: counts → exclusive prefix → stable replay
  void partition_indices(const int32_t* pids, int64_t rows, int world, int64_t* starts, uint32_t* idx) {
    std::vector<int64_t> cnt((size_t)world + 1, 0);
    for (int64_t i = 0; i < rows; i++) cnt[(size_t)pids[i] + 1]++;
    for (int p = 0; p < world; p++) cnt[(size_t)p + 1] += cnt[(size_t)p];
    for (int p = 0; p <= world; p++) starts[p] = cnt[(size_t)p];
    std::vector<int64_t> at(cnt.begin(), cnt.end() - 1);
    for (int64_t i = 0; i < rows; i++) idx[at[(size_t)pids[i]]++] = (uint32_t)i;
  }
  void take(int w, const void* src, const uint32_t* idx, int64_t n, void* dst) {
    for (int64_t i = 0; i < n; i++) memcpy((char*)dst + (size_t)i * (size_t)w, (const char*)src + (size_t)idx[i] * (size_t)w, (size_t)w);
  }
  static bool bit(const uint8_t* bits, uint32_t i) { return (bits[i >> 3] >> (i & 7)) & 1; }
  void take_valid_bytes(const uint8_t* bits, const uint32_t* idx, int64_t n, uint8_t* out) {
    for (int64_t i = 0; i < n; i++) out[i] = bit(bits, idx[i]) ? 1 : 0;
  }
  void take_utf8_lengths(const int32_t* offs, const uint32_t* idx, const uint8_t* valid_bits, int64_t n, uint32_t* lengths) {
    for (int64_t i = 0; i < n; i++) lengths[i] = (valid_bits && !bit(valid_bits, idx[i])) ? 0u : (uint32_t)(offs[idx[i] + 1] - offs[idx[i]]);
  }
  void take_utf8_copy(const int32_t* offs, const uint8_t* bytes, const uint32_t* idx, const uint8_t* valid_bits, int64_t n, const int32_t* out_offs, uint8_t* out) {
    for (int64_t i = 0; i < n; i++) {
      if (valid_bits && !bit(valid_bits, idx[i])) continue;
      memcpy(out + out_offs[i], bytes + offs[idx[i]], (size_t)(offs[idx[i] + 1] - offs[idx[i]]));
    }
  }
  void scan_u32(const uint32_t* lengths, int64_t n, int32_t* offsets) {
    int64_t run = 0;
    for (int64_t i = 0; i < n; i++) {
      offsets[i] = (int32_t)run;
      run += lengths[i];
    }
    offsets[n] = (int32_t)run;
  }
  void pack(const uint8_t* bytes, uint8_t* bitmap, int64_t n) {
    memset(bitmap, 0, (size_t)((n + 7) / 8));
    for (int64_t i = 0; i < n; i++)
      if (bytes[i]) bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
  }
  void set_bytes(void* p, int v, size_t n) { memset(p, v, n); }
  void read_i32_at(const int32_t* base, const int64_t* positions, int count, int32_t* out) {
    for (int k = 0; k < count; k++) out[k] = base[positions[k]];
  }
  void copy(void* dst, const void* src, size_t n) { memcpy(dst, src, n); }
  void to_host(void*, const void*, size_t) {}
  void from_host(void*, const void*, size_t) {}
  void before_transport() {}
  void sync() {}
};

// the product's RCCL transport (csrc/exchange_rccl.hpp — the text libcomet.so compiles) over host memory: "device" buffers are malloc'ed, the
// stream is nothing.  Whatever librccl COMET_RCCL_LIBRARY names answers it; the tests name tests/fake_rccl/.
struct HostMem {
  using Buf = HostBufT;
  using HostBuf = HostBufT;
  static void h2d(void* dst, const void* src, size_t n, void*) { memcpy(dst, src, n); }
  static void d2h(void* dst, const void* src, size_t n, void*) { memcpy(dst, src, n); }
  static void sync(void*) {}
};
struct RcclComm {
  ncclComm_t nccl = nullptr;
  int world = 1, rank = 0;
  std::atomic<int64_t> sent{0}, received{0};
  std::unique_ptr<RcclTransportT<HostMem>> t;
};

std::mutex g_mu;
std::map<int64_t, std::unique_ptr<RcclComm>> g_rccl;
std::map<int64_t, std::unique_ptr<TcpTransport>> g_comms;
std::map<int64_t, std::unique_ptr<Result<HostOps>>> g_results;
int64_t g_next = 1;
thread_local std::string t_error;

}  // namespace

extern "C" {

const char* xh_last_error(void) { return t_error.c_str(); }

int64_t xh_comm_init_tcp(const char* peers, int32_t world, int32_t rank, int32_t timeout_ms) {
  try {
    std::unique_ptr<TcpTransport> t(new TcpTransport(peers, world, rank, timeout_ms));
    std::lock_guard<std::mutex> lk(g_mu);
    const int64_t h = g_next++;
    g_comms[h] = std::move(t);
    return h;
  } catch (const std::exception& e) {
    t_error = e.what();
    return 0;
  }
}

// ---- the RCCL wire: the same three steps as comet_comm_unique_id / comet_comm_init_rank / comet_comm_stats (csrc/exchange.cpp) ----
int32_t xh_comm_unique_id(uint8_t* out128) {
  try {
    NcclUniqueId id;
    Rccl& r = Rccl::get();
    r.check(r.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out128, id.internal, 128);
    return 0;
  } catch (const std::exception& e) {
    t_error = e.what();
    return -2;
  }
}

int64_t xh_comm_init_rccl(const uint8_t* id128, int32_t world, int32_t rank) {
  try {
    Rccl& r = Rccl::get();
    std::unique_ptr<RcclComm> c(new RcclComm());
    c->world = world;
    c->rank = rank;
    NcclUniqueId id;
    memcpy(id.internal, id128, 128);
    r.check(r.CommInitRank(&c->nccl, world, id, rank), "ncclCommInitRank");
    c->t.reset(new RcclTransportT<HostMem>(c->nccl, world, rank, nullptr, &c->sent, &c->received));
    std::lock_guard<std::mutex> lk(g_mu);
    const int64_t h = g_next++;
    g_rccl[h] = std::move(c);
    return h;
  } catch (const std::exception& e) {
    t_error = e.what();
    return 0;
  }
}

// out[0] = ncclCommCount, out[1] = ncclCommUserRank, out[2] / out[3] = bytes sent to / received from other ranks (comet_comm_stats's four)
int32_t xh_comm_stats(int64_t comm, int64_t* out4) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_rccl.find(comm);
  if (it == g_rccl.end()) return -1;
  Rccl& r = Rccl::get();
  int v = -1;
  out4[0] = (r.CommCount && r.CommCount(it->second->nccl, &v) == 0) ? v : -1;
  v = -1;
  out4[1] = (r.CommUserRank && r.CommUserRank(it->second->nccl, &v) == 0) ? v : -1;
  out4[2] = it->second->sent.load();
  out4[3] = it->second->received.load();
  return 0;
}

// the stand-in library's own log of the calls it saw on this communicator (fake_rccl_counters; 7 values) — 0 when the loaded library has none
int32_t xh_comm_wire_counters(int64_t comm, int64_t* out7) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_rccl.find(comm);
  if (it == g_rccl.end()) return -1;
  auto fn = (void (*)(ncclComm_t, int64_t*))dlsym(Rccl::get().lib, "fake_rccl_counters");
  if (!fn) return 0;
  fn(it->second->nccl, out7);
  return 7;
}

void xh_comm_destroy(int64_t comm) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_comms.erase(comm);
  auto it = g_rccl.find(comm);
  if (it != g_rccl.end()) {
    if (it->second->nccl) Rccl::get().CommDestroy(it->second->nccl);
    g_rccl.erase(it);
  }
}

int64_t xh_exchange(int64_t comm, int32_t n_cols, const CometExchangeColumn* cols, int64_t rows, const int32_t* oracle_pids) {
  try {
    Transport* t;
    {
      std::lock_guard<std::mutex> lk(g_mu);
      auto it = g_comms.find(comm);
      auto ir = g_rccl.find(comm);
      if (it != g_comms.end()) t = it->second.get();
      else if (ir != g_rccl.end()) t = ir->second->t.get();
      else throw Error("invalid communicator handle");
    }
    HostOps ops;
    ops.oracle_pids = oracle_pids;
    std::unique_ptr<Result<HostOps>> res(new Result<HostOps>());
    const int32_t no_key = 0;
    run(ops, *t, n_cols, cols, rows, &no_key, 0, *res);
    std::lock_guard<std::mutex> lk(g_mu);
    const int64_t h = g_next++;
    g_results[h] = std::move(res);
    return h;
  } catch (const std::exception& e) {
    t_error = e.what();
    return 0;
  }
}

int64_t xh_result_rows(int64_t r) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_results.find(r);
  return it == g_results.end() ? -1 : it->second->rows;
}
int32_t xh_result_column(int64_t r, int32_t col, void** values, void** validity, void** aux, int64_t* aux_bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_results.find(r);
  if (it == g_results.end() || col < 0 || (size_t)col >= it->second->values.size()) return -2;
  Result<HostOps>& x = *it->second;
  *values = x.values[(size_t)col]->p;
  *validity = x.validity[(size_t)col] ? x.validity[(size_t)col]->p : nullptr;
  *aux = x.aux[(size_t)col] ? x.aux[(size_t)col]->p : nullptr;
  *aux_bytes = x.aux_bytes[(size_t)col];
  return 0;
}
void xh_result_release(int64_t r) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_results.erase(r);
}

}  // extern "C"
