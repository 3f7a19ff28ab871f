// In-library hash exchange between GPUs ("the multi-GPU path lives inside one native plan execution, or across the N
// concurrent task contexts of one executor, coordinated in-library").  One communicator per GPU:
//   * RCCL transport — one PROCESS per GPU (bench.py under torchrun, a multi-executor deployment): librccl is dlopen'ed, the 128-byte
//     unique id travels out of band (the control plane: the launcher's rendezvous), ncclCommInitRank binds rank ↔ device; the data
//     plane is a count exchange (ncclAllGather of the world send counts) and, per buffer, ONE group of ncclSend / ncclRecv pairs —
//     xGMI is point to point, so the all-to-all maps one to one onto the links;
//   * local transport — N task THREADS of one process, one GPU each (the Spark-executor shape: COMET_GPU_DEVICES), or several ranks on
//     one GPU in tests: the ranks meet at an in-process rendezvous, publish their partitioned send buffers and pull their slices with
//     hipMemcpyAsync (peer copies over xGMI between devices).
// comet_exchange itself is transport independent: murmur3 (seed 42) chained over the key columns → pmod(world) → partition_starts /
// partition_row_indices exactly as the reference's shuffle writer computes them (multi_partition.rs:54-103) → one `take` per buffer
// into partition order → transport → the received slices, sender after sender in rank order, each sender's rows in their input
// order.  Validity travels one byte per row (partition boundaries are not byte aligned) and is packed again on arrival.
// Fixed-width columns (ints, floats, dates, timestamps, decimals), Boolean (bit-packed values travel one byte per row like validity)
// and Utf8 / Binary: the lengths travel one int32 per row, the bytes — gathered into partition order — with per-partition BYTE counts
// (a second count exchange), and the receiver rebuilds its int32 offsets with one prefix sum over the received lengths.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/comet_amd.h"
#include "exchange_core.hpp"
#include "exchange_rccl.hpp"
#include "exchange_tcp.hpp"
#include "exec.hpp"
#include "plan.hpp"

extern "C" int comet_launch_partition_indices(const int32_t* pids, int64_t n, int32_t P, uint64_t* hist, uint32_t* bad, int64_t* starts,
                                              uint32_t* row_indices, void* stream);
extern "C" int64_t comet_partition_scratch_bytes(int64_t n, int32_t P);
extern "C" int64_t comet_partition_tiles(int64_t n);
extern "C" int comet_launch_take(int width, const void* src, const uint32_t* idx, int64_t n, void* dst, void* stream);
extern "C" int comet_launch_take_valid_bytes(const uint8_t* valid_bits, const uint32_t* idx, int64_t n, uint8_t* out_bytes, void* stream);
extern "C" int comet_launch_fill(int width, void* dst, int64_t n, const void* value, void* stream);
extern "C" void pq_launch_pack(const uint8_t* bytes, uint8_t* bitmap, int64_t n, void* st);
extern "C" void pq_launch_u32_scan(const uint32_t* lengths, int64_t n, uint64_t* tiles, int32_t* offsets, void* st);
extern "C" int comet_launch_take_utf8_lengths(const int32_t* offs, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits, int64_t n,
                                              uint32_t* lengths, void* stream);
extern "C" int comet_launch_take_utf8_copy(const int32_t* offs, const uint8_t* bytes, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits,
                                           int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream);

namespace comet {
namespace {

#define XHIP(call)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) throw CometError(std::string("exchange: ") + #call + ": " + hipGetErrorString(e_)); \
  } while (0)

// ---- RCCL through dlopen (exchange_rccl.hpp: no link-time dependency, a single-GPU deployment never loads it) over HBM ----
using xchg::NcclUniqueId;
using xchg::Rccl;
using xchg::ncclComm_t;
struct HipMem {
  using Buf = DevBuf;
  using HostBuf = PinnedBuf;
  static void h2d(void* dst, const void* src, size_t n, void* st) { XHIP(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (hipStream_t)st)); }
  static void d2h(void* dst, const void* src, size_t n, void* st) { XHIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, (hipStream_t)st)); }
  static void sync(void* st) { XHIP(hipStreamSynchronize((hipStream_t)st)); }
};
using RcclTransport = xchg::RcclTransportT<HipMem>;      // one process per GPU; everything is enqueued on the communicator's stream

// ---- in-process rendezvous of the local transport ----
struct LocalGroup {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t epoch = 0;
  std::vector<const void*> send_ptr;             // per rank: the buffer currently being exchanged (partition order)
  std::vector<std::vector<int64_t>> starts;      // per rank: partition_starts (world + 1) of the current table
  std::vector<std::vector<int64_t>> flags;       // per rank: which columns carry a validity bitmap
  std::vector<hipEvent_t> ready;                 // per rank: recorded when its send buffer is complete on its stream
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t e = epoch;
    if (++arrived == world) {
      arrived = 0;
      epoch++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return epoch != e; });
    }
  }
};
std::mutex g_groups_mu;
std::map<int64_t, std::shared_ptr<LocalGroup>> g_groups;

// ---- transports (exchange_core.hpp Transport); the RCCL one is exchange_rccl.hpp's ----
// N task threads of one process: publish through the group's slots, pull the slices with peer copies
class LocalTransport : public xchg::Transport {
 public:
  LocalTransport(std::shared_ptr<LocalGroup> g, int rank, hipStream_t st, hipEvent_t ready) : g_(std::move(g)), rank_(rank), st_(st), ready_(ready) {}
  int world() const override { return g_->world; }
  int rank() const override { return rank_; }
  bool host_memory() const override { return false; }
  void allgather_i64(const int64_t* mine, int n, int64_t* all) override {
    LocalGroup& g = *g_;
    { std::lock_guard<std::mutex> lk(g.mu); g.starts[(size_t)rank_].assign(mine, mine + n); }
    g.barrier();
    { std::lock_guard<std::mutex> lk(g.mu);
      for (int s = 0; s < g.world; s++) {
        if ((int)g.starts[(size_t)s].size() != n) throw CometError("exchange: local ranks disagree about the collective they are in");
        memcpy(all + (size_t)s * (size_t)n, g.starts[(size_t)s].data(), (size_t)n * 8);
      } }
    g.barrier();      // everyone has read: the slot may be published again
  }
  void alltoallv(const void* send_buf, void* recv_buf, int w, const xchg::Split& sp) override {
    LocalGroup& g = *g_;
    XHIP(hipEventRecord(ready_, st_));
    { std::lock_guard<std::mutex> lk(g.mu); g.send_ptr[(size_t)rank_] = send_buf; }
    g.barrier();                                              // every send buffer is published (and its event recorded)
    for (int s = 0; s < g.world; s++) {
      const void* src;
      hipEvent_t ev;
      { std::lock_guard<std::mutex> lk(g.mu); src = g.send_ptr[(size_t)s]; ev = g.ready[(size_t)s]; }
      if (!sp.recv[(size_t)s]) continue;
      XHIP(hipStreamWaitEvent(st_, ev, 0));
      XHIP(hipMemcpyAsync((char*)recv_buf + (size_t)sp.roff[(size_t)s] * (size_t)w, (const char*)src + (size_t)sp.peer_off[(size_t)s] * (size_t)w,
                          (size_t)sp.recv[(size_t)s] * (size_t)w, hipMemcpyDeviceToDevice, st_));
    }
    XHIP(hipStreamSynchronize(st_));                          // my pulls are done …
    g.barrier();                                              // … and so are everybody's: the send buffers may be reused
  }

 private:
  std::shared_ptr<LocalGroup> g_;
  int rank_;
  hipStream_t st_;
  hipEvent_t ready_;
};

// one rank, no wire
class SelfTransport : public xchg::Transport {
 public:
  int world() const override { return 1; }
  int rank() const override { return 0; }
  bool host_memory() const override { return false; }
  bool self_only() const override { return true; }
  void allgather_i64(const int64_t* mine, int n, int64_t* all) override { memcpy(all, mine, (size_t)n * 8); }
  void alltoallv(const void*, void*, int, const xchg::Split&) override { throw CometError("exchange: internal: self transport asked to move bytes"); }
};

struct Comm {
  int world = 1, rank = 0, device = 0;
  ncclComm_t nccl = nullptr;
  std::shared_ptr<LocalGroup> local;
  std::unique_ptr<xchg::TcpTransport> tcp;
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr;
  std::atomic<int64_t> bytes_sent{0}, bytes_received{0};      // over the RCCL wire, to / from OTHER ranks
  const char* transport_name() const { return nccl ? "rccl" : tcp ? "tcp" : local ? "in-process" : "none (1 rank)"; }
};
std::mutex g_comm_mu;
std::map<int64_t, std::shared_ptr<Comm>> g_comms;
int64_t g_next_comm = 1;

// ---- the memory space of the product: HBM buffers and the partition / take / scan / pack kernels ----
struct HipOps {
  using Buf = DevBuf;
  using HostBuf = PinnedBuf;
  static constexpr bool kDeviceMemory = true;
  hipStream_t st;
  DevBuf scratch, tiles, dstarts;
  PinnedBuf hs;
  void fill_u32(uint32_t* dst, int64_t n, uint32_t v) {
    if (comet_launch_fill(4, dst, n, &v, st) != 0) throw CometError("exchange: launch failed");
  }
  void murmur3(const CometExchangeColumn& kc, int64_t rows, uint32_t* hashes) {
    if (comet_murmur3_column(kc.type_id, kc.precision, kc.values, kc.validity, kc.aux, rows, hashes, st) != 0)
      throw CometError(std::string("exchange: murmur3: ") + comet_last_error(0));
  }
  void pmod(const uint32_t* hashes, int64_t rows, int world, int32_t* pids) {
    if (comet_pmod_partition(hashes, rows, world, pids, st) != 0) throw CometError("exchange: pmod failed");
  }
  void partition_indices(const int32_t* pids, int64_t rows, int world, int64_t* host_starts, uint32_t* idx) {
    scratch.ensure((size_t)comet_partition_scratch_bytes(rows, world));
    dstarts.ensure((size_t)(world + 1) * 8 + 16);
    const size_t hist_bytes = ((size_t)world * (size_t)comet_partition_tiles(rows) + 1) * 8;
    uint32_t* bad = (uint32_t*)((char*)scratch.p + hist_bytes);
    XHIP(hipMemsetAsync(bad, 0, 4, st));
    if (comet_launch_partition_indices(pids, rows, world, (uint64_t*)scratch.p, bad, (int64_t*)dstarts.p, idx, st) != 0)
      throw CometError("exchange: partition launch failed");
    hs.ensure((size_t)(world + 1) * 8 + 16);
    XHIP(hipMemcpyAsync(hs.p, dstarts.p, (size_t)(world + 1) * 8, hipMemcpyDeviceToHost, st));
    XHIP(hipStreamSynchronize(st));
    memcpy(host_starts, hs.p, (size_t)(world + 1) * 8);
  }
  void take(int w, const void* src, const uint32_t* idx, int64_t n, void* dst) {
    if (comet_launch_take(w, src, idx, n, dst, st) != 0) throw CometError("exchange: take failed");
  }
  void take_valid_bytes(const uint8_t* bits, const uint32_t* idx, int64_t n, uint8_t* out) {
    if (comet_launch_take_valid_bytes(bits, idx, n, out, st) != 0) throw CometError("exchange: take failed");
  }
  void take_utf8_lengths(const int32_t* offs, const uint32_t* idx, const uint8_t* valid_bits, int64_t n, uint32_t* lengths) {
    if (comet_launch_take_utf8_lengths(offs, idx, nullptr, valid_bits, n, lengths, st) != 0) throw CometError("exchange: take failed");
  }
  void take_utf8_copy(const int32_t* offs, const uint8_t* bytes, const uint32_t* idx, const uint8_t* valid_bits, int64_t n, const int32_t* out_offs, uint8_t* out) {
    if (comet_launch_take_utf8_copy(offs, bytes, idx, nullptr, valid_bits, n, out_offs, out, st) != 0) throw CometError("exchange: take failed");
  }
  void scan_u32(const uint32_t* lengths, int64_t n, int32_t* offsets) {
    tiles.ensure((size_t)((n + 1023) / 1024 + 2) * 8);
    pq_launch_u32_scan(lengths, n, (uint64_t*)tiles.p, offsets, st);
  }
  void pack(const uint8_t* bytes, uint8_t* bitmap, int64_t n) { pq_launch_pack(bytes, bitmap, n, st); }
  void set_bytes(void* p, int v, size_t n) { XHIP(hipMemsetAsync(p, v, n, st)); }
  void read_i32_at(const int32_t* base, const int64_t* positions, int count, int32_t* out_host) {
    hs.ensure((size_t)count * 4 + 16);
    for (int k = 0; k < count; k++) XHIP(hipMemcpyAsync((char*)hs.p + (size_t)k * 4, base + positions[k], 4, hipMemcpyDeviceToHost, st));
    XHIP(hipStreamSynchronize(st));
    memcpy(out_host, hs.p, (size_t)count * 4);
  }
  void copy(void* dst, const void* src, size_t n) { XHIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, st)); }
  void to_host(void* host, const void* src, size_t n) {
    XHIP(hipMemcpyAsync(host, src, n, hipMemcpyDeviceToHost, st));
    XHIP(hipStreamSynchronize(st));
  }
  void from_host(void* dst, const void* host, size_t n) {
    XHIP(hipMemcpyAsync(dst, host, n, hipMemcpyHostToDevice, st));
    XHIP(hipStreamSynchronize(st));      // the staging buffer is reused by the next column
  }
  void before_transport() {}             // RCCL and the local transport order themselves on the stream
  void sync() { XHIP(hipStreamSynchronize(st)); }
};

struct ExchangeResult {
  xchg::Result<HipOps> r;
  int device = 0;
};
std::mutex g_res_mu;
std::map<int64_t, std::shared_ptr<ExchangeResult>> g_results;
int64_t g_next_res = 1;

thread_local std::string t_error;

std::shared_ptr<Comm> find_comm(int64_t h) {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  auto it = g_comms.find(h);
  if (it == g_comms.end()) throw CometError("exchange: invalid communicator handle");
  return it->second;
}

template <class F>
auto guarded(F f, decltype(f()) err) -> decltype(f()) {
  try {
    return f();
  } catch (const std::exception& e) {
    t_error = e.what();
  } catch (...) {
    t_error = "unknown native error";
  }
  return err;
}

}  // namespace
}  // namespace comet

using namespace comet;

extern "C" {

const char* comet_exchange_last_error(void) { return t_error.c_str(); }

int32_t comet_comm_unique_id(uint8_t* out128) {
  return guarded([&]() -> int32_t {
    NcclUniqueId id;
    Rccl& r = Rccl::get();
    r.check(r.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out128, id.internal, 128);
    return 0;
  }, (int32_t)-2);
}

static int64_t register_comm(std::shared_ptr<Comm> c) {
  XHIP(hipSetDevice(c->device));
  XHIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  XHIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  std::lock_guard<std::mutex> lk(g_comm_mu);
  int64_t h = g_next_comm++;
  g_comms[h] = c;
  return h;
}

int64_t comet_comm_init_rank(const uint8_t* id128, int32_t world, int32_t rank, int32_t device_id) {
  return guarded([&]() -> int64_t {
    if (world < 1 || rank < 0 || rank >= world) throw CometError("exchange: bad rank / world");
    auto c = std::make_shared<Comm>();
    c->world = world; c->rank = rank; c->device = device_id;
    XHIP(hipSetDevice(device_id));
    if (world > 1 || getenv("COMET_EXCHANGE_FORCE_RCCL")) {   // a 1-rank RCCL communicator is legal (single-GPU tests of this transport)
      Rccl& r = Rccl::get();
      NcclUniqueId id;
      memcpy(id.internal, id128, 128);
      r.check(r.CommInitRank(&c->nccl, world, id, rank), "ncclCommInitRank");
    }
    return register_comm(c);
  }, (int64_t)0);
}

int64_t comet_comm_init_local(int64_t group_id, int32_t world, int32_t rank, int32_t device_id) {
  return guarded([&]() -> int64_t {
    if (world < 1 || rank < 0 || rank >= world) throw CometError("exchange: bad rank / world");
    auto c = std::make_shared<Comm>();
    c->world = world; c->rank = rank; c->device = device_id;
    {
      std::lock_guard<std::mutex> lk(g_groups_mu);
      auto& g = g_groups[group_id];
      if (!g) {
        g = std::make_shared<LocalGroup>();
        g->world = world;
        g->send_ptr.assign((size_t)world, nullptr);
        g->starts.assign((size_t)world, {});
        g->flags.assign((size_t)world, {});
        g->ready.assign((size_t)world, nullptr);
      }
      if (g->world != world) throw CometError("exchange: local group joined with a different world size");
      c->local = g;
    }
    int64_t h = register_comm(c);
    c->local->ready[(size_t)rank] = c->ready;
    return h;
  }, (int64_t)0);
}

int64_t comet_comm_init_tcp(const char* peers, int32_t world, int32_t rank, int32_t device_id, int32_t timeout_ms) {
  return guarded([&]() -> int64_t {
    if (world < 1 || rank < 0 || rank >= world) throw CometError("exchange: bad rank / world");
    if (!peers) throw CometError("exchange: tcp transport needs the peer list");
    auto c = std::make_shared<Comm>();
    c->world = world; c->rank = rank; c->device = device_id;
    c->tcp.reset(new xchg::TcpTransport(peers, world, rank, timeout_ms));     // blocks until every pair of ranks is connected (or the deadline)
    return register_comm(c);
  }, (int64_t)0);
}

void comet_comm_destroy(int64_t comm) {
  std::shared_ptr<Comm> c;
  {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    auto it = g_comms.find(comm);
    if (it == g_comms.end()) return;
    c = it->second;
    g_comms.erase(it);
  }
  (void)hipSetDevice(c->device);
  if (c->nccl) { try { Rccl::get().CommDestroy(c->nccl); } catch (...) {} }
  if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  if (c->ready) (void)hipEventDestroy(c->ready);
}

int64_t comet_exchange(int64_t comm, int32_t n_cols, const CometExchangeColumn* cols, int64_t rows, const int32_t* key_cols, int32_t n_keys) {
  return guarded([&]() -> int64_t {
    auto c = find_comm(comm);
    XHIP(hipSetDevice(c->device));
    auto res = std::make_shared<ExchangeResult>();
    res->device = c->device;
    HipOps ops;
    ops.st = c->stream;
    // the orchestration is exchange_core.hpp's, whatever the wire
    if (c->nccl) {
      RcclTransport t(c->nccl, c->world, c->rank, c->stream, &c->bytes_sent, &c->bytes_received);
      xchg::run(ops, t, n_cols, cols, rows, key_cols, n_keys, res->r);
    } else if (c->tcp) {
      xchg::run(ops, *c->tcp, n_cols, cols, rows, key_cols, n_keys, res->r);
    } else if (c->world > 1) {
      LocalTransport t(c->local, c->rank, c->stream, c->ready);
      xchg::run(ops, t, n_cols, cols, rows, key_cols, n_keys, res->r);
    } else {
      SelfTransport t;
      xchg::run(ops, t, n_cols, cols, rows, key_cols, n_keys, res->r);
    }
    std::lock_guard<std::mutex> lk(g_res_mu);
    int64_t h = g_next_res++;
    g_results[h] = res;
    return h;
  }, (int64_t)0);
}

const char* comet_comm_transport(int64_t comm) {
  std::lock_guard<std::mutex> lk(g_comm_mu);
  auto it = g_comms.find(comm);
  return it == g_comms.end() ? "" : it->second->transport_name();
}

// out[0] = ranks of the communicator AS THE WIRE REPORTS IT (RCCL: ncclCommCount; −1 when the library has no such entry), out[1] = this
// rank there (ncclCommUserRank), out[2] / out[3] = bytes this rank has sent to / received from other ranks over RCCL since the
// communicator was made.  A bench line that prints these proves how many ranks RCCL saw without anyone reading logs.
int32_t comet_comm_stats(int64_t comm, int64_t* out4) {
  return guarded([&]() -> int32_t {
    auto c = find_comm(comm);
    out4[0] = c->world;
    out4[1] = c->rank;
    if (c->nccl) {
      Rccl& r = Rccl::get();
      int v = -1;
      out4[0] = (r.CommCount && r.CommCount(c->nccl, &v) == 0) ? v : -1;
      v = -1;
      out4[1] = (r.CommUserRank && r.CommUserRank(c->nccl, &v) == 0) ? v : -1;
    }
    out4[2] = c->bytes_sent.load();
    out4[3] = c->bytes_received.load();
    return 0;
  }, (int32_t)-1);
}

int64_t comet_exchange_result_rows(int64_t result) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  auto it = g_results.find(result);
  return it == g_results.end() ? -1 : it->second->r.rows;
}

int32_t comet_exchange_result_column(int64_t result, int32_t col, void** values, void** validity) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  auto it = g_results.find(result);
  if (it == g_results.end() || col < 0 || (size_t)col >= it->second->r.values.size()) return -2;
  *values = it->second->r.values[(size_t)col]->p;
  *validity = it->second->r.validity[(size_t)col] ? it->second->r.validity[(size_t)col]->p : nullptr;
  return 0;
}

int32_t comet_exchange_result_aux(int64_t result, int32_t col, void** bytes, int64_t* n_bytes) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  auto it = g_results.find(result);
  if (it == g_results.end() || col < 0 || (size_t)col >= it->second->r.values.size()) return -2;
  *bytes = it->second->r.aux[(size_t)col] ? it->second->r.aux[(size_t)col]->p : nullptr;
  *n_bytes = it->second->r.aux_bytes[(size_t)col];
  return 0;
}

void comet_exchange_result_release(int64_t result) {
  std::shared_ptr<ExchangeResult> r;
  {
    std::lock_guard<std::mutex> lk(g_res_mu);
    auto it = g_results.find(result);
    if (it == g_results.end()) return;
    r = it->second;
    g_results.erase(it);
  }
  (void)hipSetDevice(r->device);
  r.reset();
}

}  // extern "C"
