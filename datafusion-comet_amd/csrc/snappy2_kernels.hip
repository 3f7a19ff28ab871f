// The multi-kernel snappy pipeline on gfx950 (device/snappy2.hpp holds the algorithm as phases; this file gives the phases their
// workgroups, barriers and workgroup memory).  Kernel A: one 64-thread workgroup per 4 KiB chunk of compressed bytes; kernel B: one
// thread per page; kernel C: as A; kernel D: one 1024-thread workgroup per 64 KiB fragment of output, 134 KiB of LDS (one per CU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#define SN2_FN __device__ __forceinline__
#define SN2_LDS __attribute__((address_space(3)))
#define SN2_ATOMIC_OR_U32(p, v) __hip_atomic_fetch_or((uint32_t*)(p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SN2_ATOMIC_ADD_U32(p, v) __hip_atomic_fetch_add((uint32_t*)(p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SN2_ATOMIC_ADD_LDS(p, v) __hip_atomic_fetch_add((uint32_t*)(p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SN2_ATOMIC_MIN_LDS(p, v) __hip_atomic_fetch_min((uint32_t*)(p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#include "device/snappy2.hpp"

namespace {
using namespace comet_snappy2;

__global__ __launch_bounds__(64) void sn2_window_kernel(const Page* __restrict__ pages, const i32* __restrict__ chunk_page, const u8* __restrict__ bytes,
                                                        ChunkFn* __restrict__ fns, const u32* __restrict__ status) {
  __shared__ ChunkLds s;
  SN2_LDS ChunkLds* L = (SN2_LDS ChunkLds*)&s;
  const i64 c = blockIdx.x;
  if (status[chunk_page[c]] != ST_OK) return;
  const Page pg = pages[chunk_page[c]];
  const i64 chunk_pos = (i64)pg.body + (c - pg.chunk_first) * (i64)kChunk;
  const int t = (int)threadIdx.x;
  chunk_stage(L, bytes + pg.src_off, pg.src_len, chunk_pos, t);
  __syncthreads();
  chunk_tables(L, pg.src_len, chunk_pos, t);
  __syncthreads();
  fns[c * kWin + t] = chunk_compose(L, t, (i64)pg.src_len - chunk_pos);
}

__global__ __launch_bounds__(64) void sn2_chain_kernel(Page* pages, int npages, const u8* __restrict__ bytes, const ChunkFn* __restrict__ fns, ChunkIn* ins, i32* frag_chunk, u32* status) {
  const int i = (int)(blockIdx.x * 64 + threadIdx.x);
  if (i >= npages) return;
  page_chain(&pages[i], bytes, fns, ins, frag_chunk, status, i);
}

__global__ __launch_bounds__(64) void sn2_emit_kernel(const Page* __restrict__ pages, const i32* __restrict__ chunk_page, const u8* __restrict__ bytes,
                                                      const ChunkIn* __restrict__ ins, Elem* elems, const u32* __restrict__ status) {
  __shared__ ChunkLds s;
  SN2_LDS ChunkLds* L = (SN2_LDS ChunkLds*)&s;
  const i64 c = blockIdx.x;
  const int pi = chunk_page[c];
  if (status[pi] != ST_OK) return;
  const ChunkIn in = ins[c];
  if (in.entry == kNoEntry) return;
  const Page pg = pages[pi];
  const i64 chunk_pos = (i64)pg.body + (c - pg.chunk_first) * (i64)kChunk;
  const int t = (int)threadIdx.x;
  chunk_stage(L, bytes + pg.src_off, pg.src_len, chunk_pos, t);
  __syncthreads();
  chunk_tables(L, pg.src_len, chunk_pos, t);
  __syncthreads();
  if (t == 0) chunk_window_entries(L, in, (i64)pg.src_len - chunk_pos);
  __syncthreads();
  chunk_emit(L, chunk_pos, pg.src_len, elems + pg.elem_first, t);
}

__global__ __launch_bounds__(1024) void sn2_exec_kernel(const Page* __restrict__ pages, const i32* __restrict__ frag_page, u8* bytes, const Elem* __restrict__ elems_all,
                                                        const ChunkIn* __restrict__ ins, const i32* __restrict__ frag_chunk, u32* status, int debug_skip) {
  __shared__ ExecLds s;
  SN2_LDS ExecLds* L = (SN2_LDS ExecLds*)&s;
  const i64 f = blockIdx.x;
  const int pi = frag_page[f];
  if (status[pi] != (u32)ST_OK) return;                // routed to the one-wave kernel, flagged by another fragment, or corrupt
  const Page pg = pages[pi];
  const i32 fl = (i32)(f - pg.frag_first);
  const u32 frag_out = (u32)fl * (u32)kFrag;
  const u32 frag_end = frag_out + (u32)kFrag < (u32)pg.dst_len ? frag_out + (u32)kFrag : (u32)pg.dst_len;
  const u32 frag_len = frag_end - frag_out;
  const Elem* elems = elems_all + pg.elem_first;
  const int tid = (int)threadIdx.x;
  if (tid == 0) {
    s.nbig = 0;
    s.covered = 0;
    s.changed = 0;
    s.flags = 0;
    s.lo = 0xffffffffu;
    s.hi = 0xffffffffu;
  }
  __syncthreads();
  frag_range_search(L, elems, pg.nelems, ins + pg.chunk_first, pg.nchunks, frag_chunk[f], fl + 1 < pg.nfrags ? frag_chunk[f + 1] : -1, frag_out, frag_end, tid, kExecThreads);
  __syncthreads();
  const u32 s_lo = s.lo, s_hi = s.hi;
  const u8* src = bytes + pg.src_off;
  u8* dst = bytes + pg.dst_off;
  if (debug_skip & 4) return;                          // (COMET_SN2_DEBUG_SKIP: phase timing only — the output is wrong)
  frag_scatter(L, elems, s_lo, s_hi, frag_out, frag_end, src, dst, tid, kExecThreads);
  __syncthreads();
  frag_big_literals(L, frag_out, src, dst, tid, kExecThreads);
  __syncthreads();
  if ((s.flags & 3u) || s.covered != frag_len) {
    if (tid == 0) atomicMax(&status[pi], (s.flags & 2u) ? (u32)ST_ERR_BAD_COPY : (s.flags & 1u) ? (u32)ST_FALLBACK : (u32)ST_ERR_LENGTH);
    return;
  }
  if (!(s.flags & 4u)) return;                          // literals only: everything is in place
  if (debug_skip & 8) return;
  for (int round = 0; round < ((debug_skip & 1) ? 0 : 18); round++) {
    const bool moved = frag_jump(L, frag_len, tid, kExecThreads);
    if (moved) SN2_ATOMIC_OR_U32(&L->changed, 1u);
    __syncthreads();
    const u32 any = s.changed;
    __syncthreads();
    if (!any) break;
    if (tid == 0) s.changed = 0;
    __syncthreads();
  }
  // (no device-scope fence: the literal bytes were stored by threads of THIS workgroup, and __syncthreads orders a workgroup's global
  // accesses — a __threadfence() here cost 3.4 of the kernel's 4.7 ms, it writes the L2 back)
  __syncthreads();
  if (debug_skip & 2) return;
  frag_resolve(L, frag_out, frag_len, dst, tid, kExecThreads);
}
}  // namespace

extern "C" {
void sn2_launch_window(const void* pages, const int32_t* chunk_page, const uint8_t* bytes, void* fns, const uint32_t* status, int64_t nchunks, void* st) {
  if (nchunks > 0) hipLaunchKernelGGL(sn2_window_kernel, (unsigned)nchunks, 64, 0, (hipStream_t)st, (const Page*)pages, chunk_page, bytes, (ChunkFn*)fns, status);
}
void sn2_launch_chain(void* pages, int npages, const uint8_t* bytes, const void* fns, void* ins, int32_t* frag_chunk, uint32_t* status, void* st) {
  if (npages > 0) hipLaunchKernelGGL(sn2_chain_kernel, (unsigned)((npages + 63) / 64), 64, 0, (hipStream_t)st, (Page*)pages, npages, bytes, (const ChunkFn*)fns, (ChunkIn*)ins, frag_chunk, status);
}
void sn2_launch_emit(const void* pages, const int32_t* chunk_page, const uint8_t* bytes, const void* ins, void* elems, const uint32_t* status, int64_t nchunks, void* st) {
  if (nchunks > 0) hipLaunchKernelGGL(sn2_emit_kernel, (unsigned)nchunks, 64, 0, (hipStream_t)st, (const Page*)pages, chunk_page, bytes, (const ChunkIn*)ins, (Elem*)elems, status);
}
void sn2_launch_exec(const void* pages, const int32_t* frag_page, uint8_t* bytes, const void* elems, const void* ins, const int32_t* frag_chunk, uint32_t* status, int64_t nfrags, void* st) {
  static const int debug_skip = getenv("COMET_SN2_DEBUG_SKIP") ? atoi(getenv("COMET_SN2_DEBUG_SKIP")) : 0;
  if (nfrags > 0) hipLaunchKernelGGL(sn2_exec_kernel, (unsigned)nfrags, 1024, 0, (hipStream_t)st, (const Page*)pages, frag_page, bytes, (const Elem*)elems, (const ChunkIn*)ins, frag_chunk, status, debug_skip);
}
}
