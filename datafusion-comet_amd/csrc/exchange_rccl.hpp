// The RCCL wire of the hash exchange (north_star: "RCCL all-to-all of partitioned build rows over xGMI"): librccl through dlopen and
// the transport that drives it — ONE ncclAllGather of int64 counts per split, ONE ncclGroupStart / ncclSend… / ncclRecv… / ncclGroupEnd per
// buffer (xGMI is point to point: the all-to-all maps one to one onto the links).
//
// A header of its own, and a template over the memory space `Mem`, for one reason: the exact call sequence — argument order, the
// ncclDataType_t values, zero-length slices, a rank sending to itself inside the group, what crosses through device memory — must be
// exercisable WITHOUT eight GPUs.  libcomet.so instantiates it over HBM (exchange.cpp HipMem: hipMemcpyAsync + stream synchronisation);
// tests/exchange_host/ instantiates the same text over host memory against a stand-in librccl (tests/fake_rccl/, test infrastructure) in
// 2 and 8 processes on a CPU-only box.  COMET_RCCL_LIBRARY names the library to load (default: the librccl.so the process already holds —
// torch ships one — or the one under /opt/rocm/lib).
#pragma once
#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "exchange_core.hpp"

namespace comet {
namespace xchg {

typedef struct ncclComm* ncclComm_t;
struct NcclUniqueId { char internal[128]; };
constexpr int kNcclInt64 = 4, kNcclUint8 = 1;   // ncclDataType_t (nccl.h): ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommCount)(const ncclComm_t, int*) = nullptr;          // (optional: what the communicator itself says about its size / this rank)
  int (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, void*) = nullptr;      // (…, ncclDataType_t, peer, comm, hipStream_t)
  int (*Recv)(void*, size_t, int, int, ncclComm_t, void*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  static Rccl& get() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [&]() {
      const char* named = getenv("COMET_RCCL_LIBRARY");
      if (named && *named) {
        r.lib = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
      } else {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)       // a copy the process already holds (torch ships one) is reused: two RCCL instances do not share state
          if ((r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        for (const char* n : names)
          if (!r.lib && (r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
      }
      if (!r.lib) return;
      auto sym = [&](const char* s) { return dlsym(r.lib, s); };
      r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
      r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
      r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
      r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
      r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
      r.Send = (decltype(r.Send))sym("ncclSend");
      r.Recv = (decltype(r.Recv))sym("ncclRecv");
      r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
      r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
      r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
      r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.Send || !r.Recv || !r.AllGather || !r.GroupStart || !r.GroupEnd)
      throw Error("exchange: librccl.so could not be loaded (needed for the multi-process RCCL transport)");
    return r;
  }
  void check(int rc, const char* what) {
    if (rc != 0) throw Error(std::string("exchange: ") + what + ": " + (GetErrorString ? GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
  }
};

// Mem: `Buf` (memory RCCL may address: HBM) and `HostBuf` (memory the host fills / reads; pinned in the product), each with ensure(bytes) and
// `p`; h2d / d2h enqueue a copy on stream `st`, sync(st) waits for the stream.
template <class Mem>
class RcclTransportT : public Transport {
 public:
  RcclTransportT(ncclComm_t comm, int world, int rank, void* st, std::atomic<int64_t>* sent = nullptr, std::atomic<int64_t>* received = nullptr)
      : comm_(comm), world_(world), rank_(rank), st_(st), sent_(sent), received_(received) {}
  int world() const override { return world_; }
  int rank() const override { return rank_; }
  bool host_memory() const override { return false; }
  // everything is enqueued on the communicator's stream; the counts cross through device memory
  void allgather_i64(const int64_t* mine, int n, int64_t* all) override {
    Rccl& r = Rccl::get();
    typename Mem::Buf dsend, dall;
    typename Mem::HostBuf hsend, hall;
    dsend.ensure((size_t)n * 8 + 16);
    dall.ensure((size_t)n * world_ * 8 + 16);
    hsend.ensure((size_t)n * 8 + 16);
    hall.ensure((size_t)n * world_ * 8 + 16);
    memcpy(hsend.p, mine, (size_t)n * 8);
    Mem::h2d(dsend.p, hsend.p, (size_t)n * 8, st_);
    r.check(r.AllGather(dsend.p, dall.p, (size_t)n, kNcclInt64, comm_, st_), "ncclAllGather");      // sendcount = ELEMENTS per rank
    Mem::d2h(hall.p, dall.p, (size_t)n * world_ * 8, st_);
    Mem::sync(st_);
    memcpy(all, hall.p, (size_t)n * world_ * 8);
  }
  void alltoallv(const void* send_buf, void* recv_buf, int w, const Split& sp) override {
    // ONE group of send / recv pairs per buffer; a slice of zero units is neither sent nor posted (both sides know the counts)
    Rccl& r = Rccl::get();
    r.check(r.GroupStart(), "ncclGroupStart");
    for (int p = 0; p < world_; p++) {
      if (sent_ && p != rank_) sent_->fetch_add((int64_t)sp.send[(size_t)p] * w);           // bytes that leave this GPU (the rank's own partition stays)
      if (received_ && p != rank_) received_->fetch_add((int64_t)sp.recv[(size_t)p] * w);
      if (sp.send[(size_t)p]) r.check(r.Send((const char*)send_buf + (size_t)sp.starts[(size_t)p] * (size_t)w, (size_t)sp.send[(size_t)p] * (size_t)w, kNcclUint8, p, comm_, st_), "ncclSend");
      if (sp.recv[(size_t)p]) r.check(r.Recv((char*)recv_buf + (size_t)sp.roff[(size_t)p] * (size_t)w, (size_t)sp.recv[(size_t)p] * (size_t)w, kNcclUint8, p, comm_, st_), "ncclRecv");
    }
    r.check(r.GroupEnd(), "ncclGroupEnd");
  }

 private:
  ncclComm_t comm_;
  int world_, rank_;
  void* st_;
  std::atomic<int64_t>*sent_, *received_;
};

}  // namespace xchg
}  // namespace comet
