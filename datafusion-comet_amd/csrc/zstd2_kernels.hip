// The zstd pipeline on gfx950 (device/zstd2.hpp holds the algorithm as phases; this file gives the phases their workgroups, barriers and
// workgroup memory).  Kernels A1 / A2: one wave per block — the literals (Huffman: a lane per stream) / the sequences of four blocks (a lane per block), decoding
// in rounds out of a window of the bitstream in workgroup memory; ≈ 14 / 52 KiB of LDS.  Kernel B: one thread per page.  Kernel C: one 256-thread workgroup per block, a pass over its records.
// Kernel D: one 1024-thread workgroup per page, its 64 KiB fragments in order, 144 KiB of LDS (one per CU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#define SN2_FN __device__ __forceinline__
#define SN2_LDS __attribute__((address_space(3)))
#define SN2_ATOMIC_OR_U32(p, v) __hip_atomic_fetch_or((uint32_t*)(p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SN2_ATOMIC_ADD_U32(p, v) __hip_atomic_fetch_add((uint32_t*)(p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SN2_ATOMIC_ADD_LDS(p, v) __hip_atomic_fetch_add((uint32_t*)(p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SN2_ATOMIC_MIN_LDS(p, v) __hip_atomic_fetch_min((uint32_t*)(p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define ZS2_DEVICE_ONLY
#ifndef ZS_SEQ_TIMING_SKIP
#define ZS_SEQ_TIMING_SKIP 0       // measurement builds only (tools/build_zstd_variants.sh): 1 skips the values phase, 2 the history / record phase — wrong output, timing of what is left
#endif
#include "device/zstd2.hpp"

namespace {
using namespace comet_zstd2;

__global__ __launch_bounds__(64) void zs2_literals_kernel(const ZPage* __restrict__ pages, const ZBlock* __restrict__ blocks, const i32* __restrict__ block_page,
                                                          const u8* __restrict__ bytes, u8* lits_all, u32* status) {
  __shared__ LitLds s;
  ZS_LDS LitLds* L = (ZS_LDS LitLds*)&s;
  const i64 bi = blockIdx.x;
  const int pi = block_page[bi];
  const ZPage pg = pages[pi];
  const ZBlock blk = blocks[bi];
  const u8* src = bytes + pg.src_off;
  u8* lits = lits_all + pg.lit_first;
  const int t = (int)threadIdx.x;
  const u32 rounds = lit_rounds(blk);
  if (!rounds) {                                            // raw / RLE literals, raw / RLE blocks: a copy / a fill
    lit_plain(src, blk, lits, t);
    return;
  }
  if (t == 0) { s.status = 0; s.huf_log = 0; s.nstreams = 0; }
  __syncthreads();
  lit_stage(L, src, blk, t);
  __syncthreads();
  if (t == 0) lit_table(L, blk);
  __syncthreads();
  lit_fill(L, src, blk, pg.src_len, t);
  __syncthreads();
  LitState st;
  const bool decodes = t < (int)s.nstreams && s.status == 0;
  if (decodes) { lit_fill_done(L, t); lit_begin(L, st, t); }
  for (u32 r = 0; r < rounds; r++) {
    if (decodes) lit_round(L, st, t);
    __syncthreads();
    if (s.status) break;
    lit_flush(L, blk, lits, r, t);
    lit_fill(L, src, blk, pg.src_len, t);
    __syncthreads();
    if (decodes) lit_fill_done(L, t);
  }
  if (t == 0 && s.status) atomicMax(&status[pi], s.status);
}

__global__ __launch_bounds__(64) void zs2_sequences_kernel(const ZPage* __restrict__ pages, ZBlock* blocks, const i32* __restrict__ block_page, const i32* __restrict__ order, i64 nblocks,
                                                           const u8* __restrict__ bytes, ZRec* recs_all, u32* status) {
  __shared__ SeqLds s;
  ZS_LDS SeqLds* L = (ZS_LDS SeqLds*)&s;
  const int t = (int)threadIdx.x, k = t / kSeqGroup, tt = t % kSeqGroup;
  const i64 slot = (i64)blockIdx.x * kSeqLanes + k;        // this group's block: the launch takes them longest first
  const bool have = slot < nblocks;
  const i64 bi = have ? (i64)order[slot] : 0;
  ZBlock blk;
  ZPage pg;
  int pi = 0;
  if (have) {
    pi = block_page[bi];
    pg = pages[pi];
    blk = blocks[bi];
  } else {                                                 // (the last workgroup's spare groups: a block without a stream, nothing to write)
    __builtin_memset(&blk, 0, sizeof blk);
    __builtin_memset(&pg, 0, sizeof pg);
    blk.type = 3;
  }
  const u8* src = bytes + pg.src_off;
  ZRec* recs = recs_all + pg.rec_first + blk.rec_first;
  seq_stage(L, k, src, blk, (u32)pg.src_len, tt);
  __syncthreads();
  if (tt == 0) seq_tables(L, k, blk);
  __syncthreads();
  seq_fill(L, k, src, blk, pg.src_len, tt);
  __syncthreads();
  SeqState st;
  if (tt == 0 && seq_block_has_stream(blk)) { seq_fill_done(L, k); seq_start(L, k, st, blk); }
  u32 rounds = 0;
  for (int j = 0; j < kSeqLanes; j++) rounds = seq_rounds_of(L, j) > rounds ? seq_rounds_of(L, j) : rounds;
  for (u32 r = 0; r < rounds; r++) {
    if (tt == 0) seq_chain_round(L, k, st, blk);
    __syncthreads();
#if !(ZS_SEQ_TIMING_SKIP & 1)
    seq_values(L, k, tt);
    __syncthreads();
#endif
#if !(ZS_SEQ_TIMING_SKIP & 2)
    seq_history_local(L, k, tt);
    __syncthreads();
    for (int p = 0; p < kSeqScanSteps; p++) {
      seq_history_step(L, k, p, tt);
      __syncthreads();
    }
    seq_history_apply(L, k, r, recs, tt);
    __syncthreads();
#endif
    if (tt == 0) seq_round_check(L, k, r);
    seq_fill(L, k, src, blk, pg.src_len, tt);
    __syncthreads();
    if (tt == 0) seq_fill_done(L, k);
  }
  if (tt == 0 && have) {
    seq_finish(L, k, st, &blocks[bi], recs, rounds);
    if (seq_status(L, k)) atomicMax(&status[pi], seq_status(L, k));
  }
}

__global__ __launch_bounds__(64) void zs2_blocks_kernel(const ZPage* __restrict__ pages, int npages, ZBlock* blocks, u32* status) {
  const int i = (int)(blockIdx.x * 64 + threadIdx.x);
  if (i >= npages) return;
  page_blocks(pages[i], blocks, status, i);
}

__global__ __launch_bounds__(kScanThreads) void zs2_records_kernel(const ZPage* __restrict__ pages, const ZBlock* __restrict__ blocks, const i32* __restrict__ block_page,
                                                                   ZRec* recs, u32* status) {
  const i64 bi = blockIdx.x;
  const int pi = block_page[bi];
  if (status[pi] != (u32)ST_OK) return;
  const ZPage pg = pages[pi];
  const ZBlock blk = blocks[bi];
  const u32 st = fix_records(recs + pg.rec_first + blk.rec_first, blk.nseq + 1, blk, (int)threadIdx.x, kScanThreads);
  if (st) atomicMax(&status[pi], st);
}

__global__ __launch_bounds__(kExecThreads) void zs2_exec_kernel(const ZPage* __restrict__ pages, u8* bytes, const u8* __restrict__ lits_all, const ZRec* __restrict__ recs_all, u32* status) {
  __shared__ ZExecLds s;
  ZS_LDS ZExecLds* L = (ZS_LDS ZExecLds*)&s;
  const int pi = (int)blockIdx.x;
  if (status[pi] != (u32)ST_OK) return;
  const ZPage pg = pages[pi];
  const ZRec* recs = recs_all + pg.rec_first;
  const u8* lits = lits_all + pg.lit_first;
  u8* dst = bytes + pg.dst_off;
  const int tid = (int)threadIdx.x;
  u32 lo = 0;
  for (u32 f0 = 0; f0 < (u32)pg.dst_len; f0 += kFrag) {
    const u32 f1 = f0 + kFrag < (u32)pg.dst_len ? f0 + kFrag : (u32)pg.dst_len, frag_len = f1 - f0;
    if (tid == 0) {
      s.e.covered = 0;
      s.e.changed = 0;
      s.e.flags = 0;
      s.nq = 0;
      s.next_lo = pg.nrecs;
    }
    __syncthreads();
    zfrag_scatter(L, recs, pg.nrecs, lo, f0, f1, lits, dst, tid, kExecThreads);
    __syncthreads();
    zfrag_long_parts(L, f0, lits, dst, tid, kExecThreads);
    __syncthreads();
    if ((s.e.flags & 3u) || s.e.covered != frag_len) {
      if (tid == 0) atomicMax(&status[pi], (s.e.flags & 2u) ? (u32)ST_ERR_OFFSET : (u32)ST_ERR_LENGTH);
      return;
    }
    lo = s.next_lo;
    if (s.e.flags & 4u) {
      for (int round = 0; round < 20; round++) {
        const bool moved = comet_snappy2::frag_jump(&L->e, frag_len, tid, kExecThreads);
        if (moved) SN2_ATOMIC_OR_U32(&L->e.changed, 1u);
        __syncthreads();
        const u32 any = s.e.changed;
        __syncthreads();
        if (!any) break;
        if (tid == 0) s.e.changed = 0;
        __syncthreads();
      }
      __syncthreads();
      comet_snappy2::frag_resolve(&L->e, f0, frag_len, dst, tid, kExecThreads);
    }
    __syncthreads();      // the fragment's bytes are final (and this workgroup's stores ordered) before the next fragment reads them
  }
}

// pages that failed: the first one into the column's error word as (page << 8 | code)
__global__ __launch_bounds__(64) void zs2_report_kernel(const u32* __restrict__ status, int npages, u32* err) {
  const int i = (int)(blockIdx.x * 64 + threadIdx.x);
  if (i >= npages) return;
  const u32 st = status[i];
  if (st >= 16u) atomicCAS(err, 0u, ((u32)i << 8) | st);
}
}  // namespace

extern "C" {
void zs2_launch_entropy(const void* pages, void* blocks, const int32_t* block_page, const int32_t* order, const uint8_t* bytes, uint8_t* lits, void* recs, uint32_t* status, int64_t nblocks, void* st) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(zs2_sequences_kernel, (unsigned)((nblocks + kSeqLanes - 1) / kSeqLanes), 64, 0, (hipStream_t)st, (const ZPage*)pages, (ZBlock*)blocks, block_page, order, (i64)nblocks, bytes,
                     (ZRec*)recs, status);
  hipLaunchKernelGGL(zs2_literals_kernel, (unsigned)nblocks, 64, 0, (hipStream_t)st, (const ZPage*)pages, (const ZBlock*)blocks, block_page, bytes, lits, status);
}
void zs2_launch_blocks(const void* pages, int npages, void* blocks, uint32_t* status, void* st) {
  if (npages > 0) hipLaunchKernelGGL(zs2_blocks_kernel, (unsigned)((npages + 63) / 64), 64, 0, (hipStream_t)st, (const ZPage*)pages, npages, (ZBlock*)blocks, status);
}
void zs2_launch_scan(const void* pages, const void* blocks, const int32_t* block_page, void* recs, uint32_t* status, int64_t nblocks, void* st) {
  if (nblocks > 0) hipLaunchKernelGGL(zs2_records_kernel, (unsigned)nblocks, kScanThreads, 0, (hipStream_t)st, (const ZPage*)pages, (const ZBlock*)blocks, block_page, (ZRec*)recs, status);
}
void zs2_launch_exec(const void* pages, int npages, uint8_t* bytes, const uint8_t* lits, const void* recs, uint32_t* status, void* st) {
  if (npages > 0) hipLaunchKernelGGL(zs2_exec_kernel, (unsigned)npages, kExecThreads, 0, (hipStream_t)st, (const ZPage*)pages, bytes, lits, (const ZRec*)recs, status);
}
void zs2_launch_report(const uint32_t* status, int npages, uint32_t* err, void* st) {
  if (npages > 0) hipLaunchKernelGGL(zs2_report_kernel, (unsigned)((npages + 63) / 64), 64, 0, (hipStream_t)st, status, npages, err);
}
}
