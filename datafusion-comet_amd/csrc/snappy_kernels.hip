// Device-side snappy decompression of Parquet data pages: one 64-lane workgroup per page (device/snappy_inflate.hpp holds the algorithm;
// this file gives it the gfx950 wave primitives).  The host ships the COMPRESSED page bodies (parquet_scan.cpp) and this kernel writes the
// decompressed pages into the column's staging area in HBM, where the decode kernels read them — the bytes cross PCIe compressed and the
// host's cores never touch them (the reference decompresses on the task's CPU core: parquet/read/... via the parquet crate's codecs).
//
// Occupancy: 66 KiB of LDS per workgroup (64 KiB history ring + 2 KiB input ring) → 2 workgroups per CU, 512 pages in flight per GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SNAPPY_FN __device__ __forceinline__
#define SNAPPY_LDS __attribute__((address_space(3)))
#ifdef COMET_SNAPPY_PROF
#define SNAPPY_TICK(i) w.tick(i)
#endif
#include "device/snappy_inflate.hpp"
#include "parquet_dev.h"

namespace {
using namespace comet_snappy;

struct DevWave {
#ifdef COMET_SNAPPY_PROF
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0, count = 0;
  __device__ __forceinline__ void tick(int i) { const unsigned long long t = __builtin_readcyclecounter(); prof[i] += t - last; last = t; if (i == 0) count++; }
#endif
  __device__ __forceinline__ int lane() const { return (int)threadIdx.x; }
  __device__ __forceinline__ u64 ballot(bool b) const { return __ballot(b); }
  __device__ __forceinline__ u32 readlane(u32 v, int src) const { return (u32)__builtin_amdgcn_readlane((int)v, src); }
  // wave64 inclusive prefix sum on the DPP path: row_shr 1/2/4/8 inside each row of 16, then row_bcast 15 / 31 across rows
  __device__ __forceinline__ u32 incl_scan_add(u32 v) const {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (u32)x;
  }
  __device__ __forceinline__ u32 gather(u32 v, u32 srclane) const { return (u32)__builtin_amdgcn_ds_bpermute((int)(srclane << 2), (int)v); }
  __device__ __forceinline__ u32 wave_min(u32 v) const {
    u32 x = v, y;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x111, 0xf, 0xf, false); x = y < x ? y : x;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x112, 0xf, 0xf, false); x = y < x ? y : x;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x114, 0xf, 0xf, false); x = y < x ? y : x;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x118, 0xf, 0xf, false); x = y < x ? y : x;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x142, 0xa, 0xf, false); x = y < x ? y : x;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x143, 0xc, 0xf, false); x = y < x ? y : x;
    return (u32)__builtin_amdgcn_readlane((int)x, 63);
  }
  // LDS operations of one wave execute in order; this only stops the compiler from moving them across the point
  __device__ __forceinline__ void lds_sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __device__ __forceinline__ void release_stores() const { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
  __device__ __forceinline__ int ctz64(u64 m) const { return __builtin_ctzll(m); }
  __device__ __forceinline__ u8 load_coherent_byte(const u8* p) const {
    const uintptr_t a = (uintptr_t)p;
    const u32 v = __hip_atomic_load((const u32*)(a & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (u8)(v >> (8 * (a & 3)));
  }
};

__global__ __launch_bounds__(64) void pq_snappy_kernel(const PqInflate* __restrict__ jobs, int njobs, uint8_t* bytes, uint32_t* err) {
  __shared__ Lds s_lds;
  const int j = (int)blockIdx.x;
  if (j >= njobs) return;
  const PqInflate job = jobs[j];
  DevWave w;
  const int rc = inflate_page(w, (SNAPPY_LDS Lds*)&s_lds, bytes + job.src_off, job.src_len, bytes + job.dst_off, job.dst_len);
#ifdef COMET_SNAPPY_PROF
  if (j == 0 && threadIdx.x == 0) printf("snappy prof: windows %llu | loop-top %llu refill %llu parse %llu chain %llu scan %llu rounds %llu flush %llu\n", w.count, w.prof[0], w.prof[1], w.prof[2], w.prof[3], w.prof[4], w.prof[5], w.prof[6]);
#endif
  if (rc != 0 && threadIdx.x == 0) atomicCAS(err, 0u, ((uint32_t)j << 8) | (uint32_t)rc);
}
// the pages the multi-kernel pipeline (snappy2_kernels.hip) flagged: status 1 = a legal stream that is not fragment-shaped → decompress it
// here, one wave per page; status ≥ 16 = corrupt → report it
__global__ __launch_bounds__(64) void pq_snappy_fallback_kernel(const PqInflate* __restrict__ jobs, int njobs, uint8_t* bytes, const uint32_t* __restrict__ status, uint32_t* err) {
  __shared__ Lds s_lds;
  const int j = (int)blockIdx.x;
  if (j >= njobs) return;
  const uint32_t st = status[j];
  if (st == 0) return;
  if (st >= 16) {
    if (threadIdx.x == 0) atomicCAS(err, 0u, ((uint32_t)j << 8) | (st == 16 ? (uint32_t)ERR_PREAMBLE : st == 18 ? (uint32_t)ERR_BAD_COPY : st == 19 ? (uint32_t)ERR_OVERRUN : (uint32_t)ERR_TRUNCATED));
    return;
  }
  const PqInflate job = jobs[j];
  DevWave w;
  const int rc = inflate_page(w, (SNAPPY_LDS Lds*)&s_lds, bytes + job.src_off, job.src_len, bytes + job.dst_off, job.dst_len);
  if (rc != 0 && threadIdx.x == 0) atomicCAS(err, 0u, ((uint32_t)j << 8) | (uint32_t)rc);
}
}  // namespace

extern "C" void pq_launch_snappy_fallback(const PqInflate* jobs, int njobs, uint8_t* bytes, const uint32_t* status, uint32_t* err, void* st) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(pq_snappy_fallback_kernel, njobs, 64, 0, (hipStream_t)st, jobs, njobs, bytes, status, err);
}

extern "C" void pq_launch_snappy(const PqInflate* jobs, int njobs, uint8_t* bytes, uint32_t* err, void* st) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(pq_snappy_kernel, njobs, 64, 0, (hipStream_t)st, jobs, njobs, bytes, err);
}

// Diagnostic / test entry (include/comet_amd.h): decompress `npages` raw snappy streams held in host memory with the kernel above and
// return the pages to host memory.  Not a data path — the scan calls pq_launch_snappy on buffers that are already in HBM.  Returns 0, or
// (page << 8 | code) of the first page that failed, or -1 for a HIP error; *kernel_ms = the kernel's time by HIP events.
extern "C" int64_t comet_snappy_inflate_pages(const uint8_t* streams, const int64_t* stream_off, const int32_t* stream_len, const int32_t* page_len,
                                              int32_t npages, uint8_t* out, const int64_t* out_off, int32_t device_id, double* kernel_ms) {
  if (npages <= 0) return 0;
  if (hipSetDevice(device_id) != hipSuccess) return -1;
  auto up16 = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
  int64_t in_total = 0, out_total = 0;
  PqInflate* jobs = (PqInflate*)malloc(sizeof(PqInflate) * (size_t)npages);
  for (int i = 0; i < npages; i++) {
    jobs[i].src_off = in_total;
    jobs[i].src_len = stream_len[i];
    in_total = up16(in_total + stream_len[i]) + 16;
  }
  for (int i = 0; i < npages; i++) {
    jobs[i].dst_off = in_total + out_total;
    jobs[i].dst_len = page_len[i];
    out_total = up16(out_total + page_len[i]) + 16;
  }
  uint8_t* d_bytes = nullptr;
  PqInflate* d_jobs = nullptr;
  uint32_t* d_err = nullptr;
  uint32_t h_err = 0;
  int64_t rc = -1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.f;
  if (hipMalloc(&d_bytes, (size_t)(in_total + out_total) + 1024) != hipSuccess) goto done;
  if (hipMalloc(&d_jobs, sizeof(PqInflate) * (size_t)npages) != hipSuccess) goto done;
  if (hipMalloc(&d_err, 4) != hipSuccess) goto done;
  if (hipMemset(d_bytes, 0, (size_t)in_total) != hipSuccess || hipMemset(d_err, 0, 4) != hipSuccess) goto done;
  for (int i = 0; i < npages; i++)
    if (hipMemcpy(d_bytes + jobs[i].src_off, streams + stream_off[i], (size_t)stream_len[i], hipMemcpyHostToDevice) != hipSuccess) goto done;
  if (hipMemcpy(d_jobs, jobs, sizeof(PqInflate) * (size_t)npages, hipMemcpyHostToDevice) != hipSuccess) goto done;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) goto done;
  if (hipEventRecord(e0, nullptr) != hipSuccess) goto done;
  hipLaunchKernelGGL(pq_snappy_kernel, npages, 64, 0, nullptr, d_jobs, npages, d_bytes, d_err);
  if (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) goto done;
  if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) goto done;
  if (kernel_ms) *kernel_ms = (double)ms;
  if (hipMemcpy(&h_err, d_err, 4, hipMemcpyDeviceToHost) != hipSuccess) goto done;
  if (h_err) { rc = (int64_t)h_err; goto done; }
  for (int i = 0; i < npages; i++)
    if (page_len[i] && hipMemcpy(out + out_off[i], d_bytes + jobs[i].dst_off, (size_t)page_len[i], hipMemcpyDeviceToHost) != hipSuccess) goto done;
  rc = 0;
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (d_bytes) (void)hipFree(d_bytes);
  if (d_jobs) (void)hipFree(d_jobs);
  if (d_err) (void)hipFree(d_err);
  free(jobs);
  return rc;
}
