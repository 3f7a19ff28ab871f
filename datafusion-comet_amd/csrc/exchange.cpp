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
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/comet_amd.h"
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

// ---- RCCL through dlopen (no link-time dependency: a single-GPU deployment never loads it) ----
typedef struct ncclComm* ncclComm_t;
struct NcclUniqueId { char internal[128]; };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  static Rccl& get() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [&]() {
      const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char* n : names)       // a copy the process already holds (torch ships one) is reused: two RCCL instances do not share state
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
      for (const char* n : names)
        if (!r.lib && (r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
      if (!r.lib) return;
      auto sym = [&](const char* s) { return dlsym(r.lib, s); };
      r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
      r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
      r.Send = (decltype(r.Send))sym("ncclSend");
      r.Recv = (decltype(r.Recv))sym("ncclRecv");
      r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
      r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
      r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
      r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.Send || !r.Recv || !r.AllGather || !r.GroupStart || !r.GroupEnd)
      throw CometError("exchange: librccl.so could not be loaded (needed for the multi-process RCCL transport)");
    return r;
  }
  void check(int rc, const char* what) {
    if (rc != 0) throw CometError(std::string("exchange: ") + what + ": " + (GetErrorString ? GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
  }
};
constexpr int kNcclInt64 = 4, kNcclUint8 = 1;   // ncclDataType_t (nccl.h): ncclUint8 = 1, ncclInt64 = 4

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

struct Comm {
  int world = 1, rank = 0, device = 0;
  ncclComm_t nccl = nullptr;
  std::shared_ptr<LocalGroup> local;
  hipStream_t stream = nullptr;
  hipEvent_t ready = nullptr;
};
std::mutex g_comm_mu;
std::map<int64_t, std::shared_ptr<Comm>> g_comms;
int64_t g_next_comm = 1;

struct ExchangeResult {
  int64_t rows = 0;
  std::vector<std::unique_ptr<DevBuf>> values, validity, aux;   // validity[c] null ⇔ column arrives without a bitmap; aux[c]: Utf8 bytes
  std::vector<int64_t> aux_bytes;
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

constexpr int kUtf8Column = 0, kBoolColumn = -1;   // value_width of the two kinds that are not fixed-width byte columns
int value_width(int type_id) {
  switch ((TypeId)type_id) {
    case TypeId::Bool: return kBoolColumn;
    case TypeId::String: case TypeId::Bytes: return kUtf8Column;
    case TypeId::Int8: return 1;
    case TypeId::Int16: return 2;
    case TypeId::Int32: case TypeId::Date: case TypeId::Float: return 4;
    case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: case TypeId::Double: return 8;
    case TypeId::Decimal: return 16;
    default: throw CometError("exchange: column type " + std::to_string(type_id) + " is not supported by the in-library exchange yet");
  }
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
    hipStream_t st = c->stream;
    const int world = c->world;
    if (rows < 0 || rows >= ((int64_t)1 << 31)) throw CometError("exchange: row count must be below 2^31");
    auto res = std::make_shared<ExchangeResult>();
    res->device = c->device;
    res->values.resize((size_t)n_cols);
    res->validity.resize((size_t)n_cols);
    res->aux.resize((size_t)n_cols);
    res->aux_bytes.assign((size_t)n_cols, 0);
    std::vector<int> width((size_t)n_cols);
    for (int i = 0; i < n_cols; i++) width[(size_t)i] = value_width(cols[i].type_id);

    // 1. partition ids: Spark's murmur3 (seed 42) chained over the key columns, then pmod
    DevBuf hashes, pids, idx, dstarts, scratch;
    hashes.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
    pids.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
    idx.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
    dstarts.ensure((size_t)(world + 1) * 8 + 16);
    std::vector<int64_t> starts((size_t)world + 1, 0);
    if (rows > 0) {
      uint32_t seed = 42;
      if (comet_launch_fill(4, hashes.p, rows, &seed, st) != 0) throw CometError("exchange: launch failed");
      for (int k = 0; k < n_keys; k++) {
        const CometExchangeColumn& kc = cols[key_cols[k]];
        if (comet_murmur3_column(kc.type_id, kc.precision, kc.values, kc.validity, kc.aux, rows, (uint32_t*)hashes.p, st) != 0)
          throw CometError(std::string("exchange: murmur3: ") + comet_last_error(0));
      }
      if (comet_pmod_partition((const uint32_t*)hashes.p, rows, world, (int32_t*)pids.p, st) != 0) throw CometError("exchange: pmod failed");
      // 2. partition_starts / partition_row_indices (stable inside every partition)
      scratch.ensure((size_t)comet_partition_scratch_bytes(rows, world));
      const size_t hist_bytes = ((size_t)world * (size_t)comet_partition_tiles(rows) + 1) * 8;
      uint32_t* bad = (uint32_t*)((char*)scratch.p + hist_bytes);
      XHIP(hipMemsetAsync(bad, 0, 4, st));
      if (comet_launch_partition_indices((const int32_t*)pids.p, rows, world, (uint64_t*)scratch.p, bad, (int64_t*)dstarts.p, (uint32_t*)idx.p, st) != 0)
        throw CometError("exchange: partition launch failed");
      PinnedBuf hs;
      hs.ensure((size_t)(world + 1) * 8 + 16);
      XHIP(hipMemcpyAsync(hs.p, dstarts.p, (size_t)(world + 1) * 8, hipMemcpyDeviceToHost, st));
      XHIP(hipStreamSynchronize(st));
      memcpy(starts.data(), hs.p, (size_t)(world + 1) * 8);
    }
    // 3. counts: who sends how many units (rows, or bytes of a Utf8 column) to whom.  `my_starts` = world + 1 unit offsets of my send
    //    buffer in partition order; collective — every rank calls it the same number of times, in the same order.
    struct Split {
      std::vector<int64_t> starts, send, recv, roff, peer_off;   // peer_off[s]: where my slice begins in sender s's buffer (local transport)
      int64_t total = 0;                                         // units this rank receives
    };
    auto make_split = [&](const std::vector<int64_t>& my_starts) {
      Split sp;
      sp.starts = my_starts;
      sp.send.assign((size_t)world, 0); sp.recv.assign((size_t)world, 0); sp.peer_off.assign((size_t)world, 0);
      for (int p = 0; p < world; p++) sp.send[(size_t)p] = my_starts[(size_t)p + 1] - my_starts[(size_t)p];
      if (world == 1 && !c->nccl) {
        sp.recv[0] = sp.send[0];
      } else if (c->nccl) {
        Rccl& r = Rccl::get();
        DevBuf dsend, dall;
        PinnedBuf hsend, hall;
        dsend.ensure((size_t)world * 8 + 16);
        dall.ensure((size_t)world * world * 8 + 16);
        hsend.ensure((size_t)world * 8 + 16);
        hall.ensure((size_t)world * world * 8 + 16);
        memcpy(hsend.p, sp.send.data(), (size_t)world * 8);
        XHIP(hipMemcpyAsync(dsend.p, hsend.p, (size_t)world * 8, hipMemcpyHostToDevice, st));
        r.check(r.AllGather(dsend.p, dall.p, (size_t)world, kNcclInt64, c->nccl, st), "ncclAllGather");
        XHIP(hipMemcpyAsync(hall.p, dall.p, (size_t)world * world * 8, hipMemcpyDeviceToHost, st));
        XHIP(hipStreamSynchronize(st));
        const int64_t* m = (const int64_t*)hall.p;   // m[s · world + d] = units rank s sends to rank d
        for (int s = 0; s < world; s++) sp.recv[(size_t)s] = m[(size_t)s * world + c->rank];
      } else {
        LocalGroup& g = *c->local;
        { std::lock_guard<std::mutex> lk(g.mu); g.starts[(size_t)c->rank] = my_starts; }
        g.barrier();
        { std::lock_guard<std::mutex> lk(g.mu);
          for (int s = 0; s < world; s++) {
            sp.peer_off[(size_t)s] = g.starts[(size_t)s][(size_t)c->rank];
            sp.recv[(size_t)s] = g.starts[(size_t)s][(size_t)c->rank + 1] - sp.peer_off[(size_t)s];
          } }
        g.barrier();      // everyone has read: the slot may be published again (the byte counts of a Utf8 column, the next exchange)
      }
      sp.roff.assign((size_t)world + 1, 0);
      for (int s = 0; s < world; s++) { sp.roff[(size_t)s] = sp.total; sp.total += sp.recv[(size_t)s]; }
      sp.roff[(size_t)world] = sp.total;
      return sp;
    };
    const Split R = make_split(starts);
    const int64_t n_out = R.total;
    res->rows = n_out;
    if (n_out >= ((int64_t)1 << 31)) throw CometError("exchange: a rank would receive 2^31 rows or more");

    // 4. every buffer: take into partition order, then move the slices
    auto move = [&](const void* send_buf, void* recv_buf, int w, const Split& sp) {   // w bytes per unit
      if (world == 1 && !c->nccl) {
        if (sp.total) XHIP(hipMemcpyAsync(recv_buf, send_buf, (size_t)sp.total * (size_t)w, hipMemcpyDeviceToDevice, st));
      } else if (c->nccl) {
        Rccl& r = Rccl::get();
        r.check(r.GroupStart(), "ncclGroupStart");
        for (int p = 0; p < world; p++) {
          if (sp.send[(size_t)p]) r.check(r.Send((const char*)send_buf + (size_t)sp.starts[(size_t)p] * (size_t)w, (size_t)sp.send[(size_t)p] * (size_t)w, kNcclUint8, p, c->nccl, st), "ncclSend");
          if (sp.recv[(size_t)p]) r.check(r.Recv((char*)recv_buf + (size_t)sp.roff[(size_t)p] * (size_t)w, (size_t)sp.recv[(size_t)p] * (size_t)w, kNcclUint8, p, c->nccl, st), "ncclRecv");
        }
        r.check(r.GroupEnd(), "ncclGroupEnd");
      } else {
        LocalGroup& g = *c->local;
        XHIP(hipEventRecord(c->ready, st));
        { std::lock_guard<std::mutex> lk(g.mu); g.send_ptr[(size_t)c->rank] = send_buf; }
        g.barrier();                                              // every send buffer is published (and its event recorded)
        for (int s = 0; s < world; s++) {
          const void* src;
          hipEvent_t ev;
          { std::lock_guard<std::mutex> lk(g.mu); src = g.send_ptr[(size_t)s]; ev = g.ready[(size_t)s]; }
          if (!sp.recv[(size_t)s]) continue;
          XHIP(hipStreamWaitEvent(st, ev, 0));
          XHIP(hipMemcpyAsync((char*)recv_buf + (size_t)sp.roff[(size_t)s] * (size_t)w, (const char*)src + (size_t)sp.peer_off[(size_t)s] * (size_t)w,
                              (size_t)sp.recv[(size_t)s] * (size_t)w, hipMemcpyDeviceToDevice, st));
        }
        XHIP(hipStreamSynchronize(st));                           // my pulls are done …
        g.barrier();                                              // … and so are everybody's: the send buffers may be reused
      }
    };
    // does the column carry validity on ANY rank?  (a rank without NULLs still has to send validity bytes then)
    std::vector<int64_t> has_valid((size_t)n_cols, 0);
    for (int i = 0; i < n_cols; i++) has_valid[(size_t)i] = cols[i].validity ? 1 : 0;
    if ((world > 1 || c->nccl) && n_cols > 0) {
      if (c->nccl) {
        Rccl& r = Rccl::get();
        DevBuf d1, d2;
        PinnedBuf h1, h2;
        d1.ensure((size_t)n_cols * 8 + 16); d2.ensure((size_t)n_cols * world * 8 + 16);
        h1.ensure((size_t)n_cols * 8 + 16); h2.ensure((size_t)n_cols * world * 8 + 16);
        memcpy(h1.p, has_valid.data(), (size_t)n_cols * 8);
        XHIP(hipMemcpyAsync(d1.p, h1.p, (size_t)n_cols * 8, hipMemcpyHostToDevice, st));
        r.check(r.AllGather(d1.p, d2.p, (size_t)n_cols, kNcclInt64, c->nccl, st), "ncclAllGather");
        XHIP(hipMemcpyAsync(h2.p, d2.p, (size_t)n_cols * world * 8, hipMemcpyDeviceToHost, st));
        XHIP(hipStreamSynchronize(st));
        for (int s = 0; s < world; s++)
          for (int i = 0; i < n_cols; i++) has_valid[(size_t)i] |= ((const int64_t*)h2.p)[(size_t)s * n_cols + i];
      } else {
        LocalGroup& g = *c->local;
        { std::lock_guard<std::mutex> lk(g.mu); g.flags[(size_t)c->rank] = has_valid; }
        g.barrier();
        { std::lock_guard<std::mutex> lk(g.mu);
          for (int s = 0; s < world; s++)
            for (int i = 0; i < n_cols && (size_t)i < g.flags[(size_t)s].size(); i++) has_valid[(size_t)i] |= g.flags[(size_t)s][(size_t)i]; }
        g.barrier();      // nobody overwrites its flags (next exchange) before everyone has read them
      }
    }
    DevBuf send_buf, vbytes_send, vbytes_recv, lengths, send_offs, recv_lengths, tiles;
    const size_t rows1 = (size_t)std::max<int64_t>(rows, 1), out1 = (size_t)std::max<int64_t>(n_out, 1);
    for (int i = 0; i < n_cols; i++) {
      const int w = width[(size_t)i];
      res->values[(size_t)i].reset(new DevBuf());
      if (w > 0) {
        send_buf.ensure(rows1 * (size_t)w + 16);
        if (rows > 0 && comet_launch_take(w, cols[i].values, (const uint32_t*)idx.p, rows, send_buf.p, st) != 0) throw CometError("exchange: take failed");
        res->values[(size_t)i]->ensure(out1 * (size_t)w + 16);
        move(send_buf.p, res->values[(size_t)i]->p, w, R);
      } else if (w == kBoolColumn) {
        // bit-packed values: one byte per row on the wire (partition boundaries are not byte aligned), packed again on arrival
        vbytes_send.ensure(rows1 + 16);
        vbytes_recv.ensure(out1 + 16);
        if (rows > 0 && comet_launch_take_valid_bytes((const uint8_t*)cols[i].values, (const uint32_t*)idx.p, rows, (uint8_t*)vbytes_send.p, st) != 0)
          throw CometError("exchange: take failed");
        move(vbytes_send.p, vbytes_recv.p, 1, R);
        res->values[(size_t)i]->ensure((size_t)((n_out + 7) / 8) + 16);
        if (n_out > 0) pq_launch_pack((const uint8_t*)vbytes_recv.p, (uint8_t*)res->values[(size_t)i]->p, n_out, st);
      } else {
        // Utf8 / Binary: lengths (0 for NULL rows) → offsets of my send bytes → the bytes in partition order
        const int32_t* offs = (const int32_t*)cols[i].values;
        lengths.ensure(rows1 * 4 + 16);
        send_offs.ensure((rows1 + 1) * 4 + 16);
        tiles.ensure((size_t)((std::max(rows, n_out) + 1023) / 1024 + 2) * 8);
        std::vector<int64_t> bstarts((size_t)world + 1, 0);
        if (rows > 0) {
          if (comet_launch_take_utf8_lengths(offs, (const uint32_t*)idx.p, nullptr, cols[i].validity, rows, (uint32_t*)lengths.p, st) != 0)
            throw CometError("exchange: take failed");
          pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)send_offs.p, st);
          PinnedBuf hb;
          hb.ensure((size_t)(world + 1) * 4 + 16);
          for (int p = 0; p <= world; p++)
            XHIP(hipMemcpyAsync((char*)hb.p + (size_t)p * 4, (const char*)send_offs.p + (size_t)starts[(size_t)p] * 4, 4, hipMemcpyDeviceToHost, st));
          XHIP(hipStreamSynchronize(st));
          for (int p = 0; p <= world; p++) bstarts[(size_t)p] = ((const int32_t*)hb.p)[p];
          if (bstarts[(size_t)world] < 0) throw CometError("exchange: Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
        }
        send_buf.ensure((size_t)std::max<int64_t>(bstarts[(size_t)world], 1) + 16);
        if (rows > 0 && comet_launch_take_utf8_copy(offs, cols[i].aux, (const uint32_t*)idx.p, nullptr, cols[i].validity, rows, (const int32_t*)send_offs.p,
                                                    (uint8_t*)send_buf.p, st) != 0)
          throw CometError("exchange: take failed");
        const Split B = make_split(bstarts);
        if (B.total >= ((int64_t)1 << 31)) throw CometError("exchange: a rank would receive 2 GiB or more of one Utf8 column");
        recv_lengths.ensure(out1 * 4 + 16);
        move(lengths.p, recv_lengths.p, 4, R);
        res->aux[(size_t)i].reset(new DevBuf());
        res->aux[(size_t)i]->ensure((size_t)std::max<int64_t>(B.total, 1) + 16);
        move(send_buf.p, res->aux[(size_t)i]->p, 1, B);
        res->aux_bytes[(size_t)i] = B.total;
        // the received slices arrive sender after sender, each in row order: one prefix sum over the lengths is the offsets buffer
        res->values[(size_t)i]->ensure((out1 + 1) * 4 + 16);
        if (n_out > 0) pq_launch_u32_scan((const uint32_t*)recv_lengths.p, n_out, (uint64_t*)tiles.p, (int32_t*)res->values[(size_t)i]->p, st);
        else XHIP(hipMemsetAsync(res->values[(size_t)i]->p, 0, 4, st));
      }
      if (has_valid[(size_t)i]) {
        vbytes_send.ensure(rows1 + 16);
        vbytes_recv.ensure(out1 + 16);
        if (rows > 0) {
          if (cols[i].validity) {
            if (comet_launch_take_valid_bytes(cols[i].validity, (const uint32_t*)idx.p, rows, (uint8_t*)vbytes_send.p, st) != 0) throw CometError("exchange: take failed");
          } else {
            XHIP(hipMemsetAsync(vbytes_send.p, 1, (size_t)rows, st));
          }
        }
        move(vbytes_send.p, vbytes_recv.p, 1, R);
        res->validity[(size_t)i].reset(new DevBuf());
        res->validity[(size_t)i]->ensure((size_t)((n_out + 7) / 8) + 16);
        if (n_out > 0) pq_launch_pack((const uint8_t*)vbytes_recv.p, (uint8_t*)res->validity[(size_t)i]->p, n_out, st);
      }
    }
    XHIP(hipStreamSynchronize(st));     // scratch buffers return to the pool; the result is complete
    std::lock_guard<std::mutex> lk(g_res_mu);
    int64_t h = g_next_res++;
    g_results[h] = res;
    return h;
  }, (int64_t)0);
}

int64_t comet_exchange_result_rows(int64_t result) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  auto it = g_results.find(result);
  return it == g_results.end() ? -1 : it->second->rows;
}

int32_t comet_exchange_result_column(int64_t result, int32_t col, void** values, void** validity) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  auto it = g_results.find(result);
  if (it == g_results.end() || col < 0 || (size_t)col >= it->second->values.size()) return -2;
  *values = it->second->values[(size_t)col]->p;
  *validity = it->second->validity[(size_t)col] ? it->second->validity[(size_t)col]->p : nullptr;
  return 0;
}

int32_t comet_exchange_result_aux(int64_t result, int32_t col, void** bytes, int64_t* n_bytes) {
  std::lock_guard<std::mutex> lk(g_res_mu);
  auto it = g_results.find(result);
  if (it == g_results.end() || col < 0 || (size_t)col >= it->second->values.size()) return -2;
  *bytes = it->second->aux[(size_t)col] ? it->second->aux[(size_t)col]->p : nullptr;
  *n_bytes = it->second->aux_bytes[(size_t)col];
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
