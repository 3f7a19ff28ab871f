// Parquet page decode on the device: definition levels → validity, RLE/bit-packed hybrid
// dictionary indices, PLAIN and RLE_DICTIONARY values → Arrow buffers.  One lane per ROW of a column chunk; the
// host only parses page headers and hybrid run headers (PqPage / PqRun tables), never a value.
// Format: Apache Parquet spec (Encodings.md: PLAIN = 0, RLE = 3, RLE_DICTIONARY = 8; hybrid runs are
// <varint header><payload>, header LSB 1 = bit-packed groups of 8, LSB 0 = RLE).  The reference delegates this to the
// `parquet` 58.4.0 crate (native/core/src/parquet/parquet_exec.rs:145-147); pyarrow is the independent checker.
#include <hip/hip_runtime.h>
#include <algorithm>

#include "device/comet_device.hpp"
#include "parquet_dev.h"
#include "device/pq_runs.hpp"

using namespace comet;

// last page whose row_start <= row
__device__ __forceinline__ int pq_find_page(const PqPage* pages, int npages, i64 row) {
  int lo = 0, hi = npages - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (pages[mid].row_start <= row) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}
// value `v` of a hybrid section described by runs[first .. first+count)
__device__ __forceinline__ u32 pq_hybrid_value(const PqRun* runs, int first, int count, const u8* bytes, int bw, i32 v) {
  int lo = first, hi = first + count - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (runs[mid].value_start <= v) lo = mid;
    else hi = mid - 1;
  }
  const PqRun r = runs[lo];
  if (r.is_rle) return r.rle_value;
  const i64 bit = (i64)(v - r.value_start) * bw;
  const u8* p = bytes + r.byte_off + (bit >> 3);
  // bw <= 32: the value spans at most 5 bytes
  u64 w = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) w |= (u64)p[k] << (8 * k);
  return (u32)((w >> (bit & 7)) & ((bw >= 32) ? 0xffffffffull : ((1ull << bw) - 1)));
}

// 1. definition levels → per-row validity byte
__global__ __launch_bounds__(256) void pq_validity_kernel(PqDecodeArgs a) {
  for (i64 row = (i64)blockIdx.x * 256 + threadIdx.x; row < a.n_rows; row += (i64)gridDim.x * 256) {
    u8 valid = 1;
    if (a.max_def > 0) {
      const PqPage pg = a.pages[pq_find_page(a.pages, a.npages, row)];
      if (pg.def_run_count > 0) {
        int bw = a.max_def == 1 ? 1 : (32 - __clz(a.max_def));
        u32 lvl = pq_hybrid_value(a.def_runs, pg.def_run_first, pg.def_run_count, a.bytes, bw, (i32)(row - pg.row_start) + pg.lvl_skip);
        valid = lvl == (u32)a.max_def;
      }
    }
    a.valid_out[row] = valid;
  }
}

// 1a. Nested leaves (a struct's field, a list's element): the LEVELS themselves, one byte per entry of the column — what the assembly kernels
// below turn into struct validity, list offsets, list validity and element validity (Dremel's definition / repetition levels:
// parquet-format LogicalTypes.md "Nested Types"; the reference reads them through the parquet crate's record reader).
__global__ __launch_bounds__(256) void pq_levels_kernel(PqDecodeArgs a, int which, u8* __restrict__ out) {
  const int maxl = which ? a.max_rep : a.max_def;
  const int bw = maxl <= 1 ? 1 : (32 - __clz(maxl));
  for (i64 row = (i64)blockIdx.x * 256 + threadIdx.x; row < a.n_rows; row += (i64)gridDim.x * 256) {
    const PqPage pg = a.pages[pq_find_page(a.pages, a.npages, row)];
    const i32 first = which ? pg.rep_run_first : pg.def_run_first, count = which ? pg.rep_run_count : pg.def_run_count;
    u32 lvl = which ? 0u : (u32)maxl;      // no runs: every repetition level 0 / every value defined
    if (maxl > 0 && count > 0) lvl = pq_hybrid_value(which ? a.rep_runs : a.def_runs, first, count, a.bytes, bw, (i32)(row - pg.row_start) + pg.lvl_skip);
    out[row] = (u8)lvl;
  }
}
// out[i] = level[i] >= thr
__global__ __launch_bounds__(256) void pq_level_ge_kernel(const u8* __restrict__ lv, i64 n, int thr, u8* __restrict__ out) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) out[i] = lv[i] >= (u8)thr;
}
// list leaf, pass 1: which entries start a row (repetition level 0) and which hold an element slot (definition level >= def_slot)
__global__ __launch_bounds__(256) void pq_list_flags_kernel(const u8* __restrict__ def, const u8* __restrict__ rep, i64 n, int def_slot, u32* __restrict__ starts, u32* __restrict__ elems) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    starts[i] = rep[i] == 0;
    elems[i] = def[i] >= (u8)def_slot;
  }
}
// list leaf, pass 2 (behind the two prefix sums): offsets and validity of the rows, validity and values of the elements.
// start_idx / elem_idx: exclusive prefix counts (n + 1 entries).  A file whose row starts do not add up to `rows` sets *err.
__global__ __launch_bounds__(256) void pq_list_assemble_kernel(const u8* __restrict__ def, const u8* __restrict__ rep, i64 n, i64 rows, int def_list, int def_slot, int max_def,
                                                               const i32* __restrict__ start_idx, const i32* __restrict__ elem_idx, const u8* __restrict__ values, int width,
                                                               i32* __restrict__ offsets, u8* __restrict__ list_valid, u8* __restrict__ elem_valid, u8* __restrict__ elem_values, u32* err) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i <= n; i += (i64)gridDim.x * 256) {
    if (i == n) {
      if ((i64)start_idx[n] != rows) atomicCAS(err, 0u, 0xD1u);
      else offsets[rows] = elem_idx[n];
      continue;
    }
    const u8 d = def[i];
    if (rep[i] == 0) {
      const i64 r = start_idx[i];
      if (r < rows) { offsets[r] = elem_idx[i]; list_valid[r] = d >= (u8)def_list; }
    }
    if (d >= (u8)def_slot) {
      const i64 e = elem_idx[i];
      elem_valid[e] = d == (u8)max_def;
      const u8* src = values + i * (i64)width;
      u8* dst = elem_values + e * (i64)width;
      for (int k = 0; k < width; k++) dst[k] = d == (u8)max_def ? src[k] : (u8)0;
    }
  }
}

// list leaf whose elements are not fixed-width (strings, booleans): the entry of every element slot, so that the element column can be TAKEN
// out of the leaf's column over entries (exec.cpp take_column) instead of copied value by value
__global__ __launch_bounds__(256) void pq_list_elem_entries_kernel(const u8* __restrict__ def, i64 n, int def_slot, const i32* __restrict__ elem_idx, u32* __restrict__ entries) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256)
    if (def[i] >= (u8)def_slot) entries[elem_idx[i]] = (u32)i;
}

// 1b. run headers of index sections the device inflated (device/pq_runs.hpp): one lane per page.  Pass 1 counts a page's runs, a prefix sum
// places them, pass 2 walks again and writes the PqRun entries behind the column's host-parsed runs and the page's (first, count).
// A malformed section leaves (page job << 8 | 0xE0 + status) in *err, like the decompression kernels.
__global__ __launch_bounds__(256) void pq_count_runs_kernel(const PqPendingRuns* __restrict__ pend, int n, const u8* __restrict__ bytes, u32* __restrict__ counts, u32* err) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= n) return;
  const PqPendingRuns p = pend[i];
  i32 runs = 0;
  const int st = pq_walk_runs(bytes, p.begin, p.end, p.bit_width, p.max_values, &runs, [](i64, i32, i32, int, u32) {});
  if (st != PQ_RUNS_OK) { atomicCAS(err, 0u, ((u32)i << 8) | (0xE0u + (u32)st)); runs = 0; }
  counts[i] = runs > 0 ? (u32)runs : 1u;      // (a page of NULLs only has no run: one RLE run of zeros stands for it, as on the host)
}
// one word from device memory to wherever `dst` points — pinned host memory: the scan reads a count without queueing a copy command
__global__ void pq_store_u32_kernel(const u32* __restrict__ src, u32* dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __hip_atomic_store(dst, *src, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(256) void pq_write_runs_kernel(const PqPendingRuns* __restrict__ pend, int n, const u8* __restrict__ bytes, const i32* __restrict__ offsets, i32 run_base,
                                                            PqRun* __restrict__ runs, PqPage* __restrict__ pages) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= n) return;
  const PqPendingRuns p = pend[i];
  const i32 first = run_base + offsets[i];
  const i32 reserved = offsets[i + 1] - offsets[i];      // what pass 1 counted for this page (the scan writes n + 1 prefix sums): a walk that fails half way — pass 1 then
  PqRun* out = runs + first;                             // reserved ONE slot — must not spill the runs before the error into the next page's slots
  i32 k = 0;
  i32 nruns = 0;
  const int st = pq_walk_runs(bytes, p.begin, p.end, p.bit_width, p.max_values, &nruns, [&](i64 byte_off, i32 value_start, i32 count, int is_rle, u32 rle_value) {
    PqRun r;
    r.byte_off = byte_off;
    r.value_start = value_start;
    r.count = count;
    r.is_rle = is_rle;
    r.rle_value = rle_value;
    r.page = p.page;
    r.pad = 0;
    if (k < reserved) out[k] = r;
    k++;
  });
  if (st != PQ_RUNS_OK || k == 0 || k > reserved) {      // (a malformed section was reported by pass 1; its page decodes as zeros and the scan fails on the error word)
    PqRun r;
    r.byte_off = 0;
    r.value_start = 0;
    r.count = pages[p.page].num_values;
    r.is_rle = 1;
    r.rle_value = 0;
    r.page = p.page;
    r.pad = 0;
    out[0] = r;
    k = 1;
  }
  pages[p.page].idx_run_first = first;
  pages[p.page].idx_run_count = k;
}

// 2. exclusive prefix count of valid rows (per 1024-row tile: count, then scan of tile counts, then apply)
__global__ __launch_bounds__(256) void pq_tile_count_kernel(const u8* valid, i64 n, u64* tile_counts) {
  const i64 ntiles = (n + 1023) / 1024;
  __shared__ u32 s_cnt;
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    u32 local = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      i64 i = t * 1024 + r * 256 + threadIdx.x;
      u64 b = __ballot(i < n && valid[i] != 0);
      if (lane_id() == 0) local += (u32)__popcll(b);
    }
    if (lane_id() == 0) atomicAdd(&s_cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[t] = s_cnt;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void pq_tile_scan_kernel(u64* counts, i64 ntiles) { tile_scan_body(counts, ntiles); }
__global__ __launch_bounds__(256) void pq_vidx_kernel(const u8* valid, i64 n, const u64* tile_off, u32* vidx) {
  const i64 ntiles = (n + 1023) / 1024;
  __shared__ u32 s_wave[4];
  __shared__ u32 s_run;
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int r = 0; r < 4; r++) {
      i64 i = t * 1024 + r * 256 + threadIdx.x;
      bool v = i < n && valid[i] != 0;
      u64 b = __ballot(v);
      u32 below = (u32)__popcll(b & ((1ull << lane_id()) - 1));
      if (lane_id() == 0) s_wave[wave_id()] = (u32)__popcll(b);
      __syncthreads();
      u32 woff = 0;
      for (int w = 0; w < wave_id(); w++) woff += s_wave[w];
      const u32 run = s_run;
      if (i < n) vidx[i] = (u32)tile_off[t] + run + woff + below;
      __syncthreads();
      if (threadIdx.x == 0) s_run = run + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
      __syncthreads();
    }
  }
}

// 3. values: PLAIN or dictionary → typed Arrow values
__device__ __forceinline__ i128 pq_flba_to_i128(const u8* p, int len) {
  // big-endian two's complement of `len` bytes
  u128 v = (p[0] & 0x80) ? ~(u128)0 : 0;
  for (int k = 0; k < len; k++) v = (v << 8) | p[k];
  return (i128)v;
}
// little-endian loads from arbitrarily aligned page bytes (gfx9+ global loads may be unaligned)
__device__ __forceinline__ u32 pq_ld32(const u8* p) { u32 x; __builtin_memcpy(&x, p, 4); return x; }
__device__ __forceinline__ u64 pq_ld64(const u8* p) { u64 x; __builtin_memcpy(&x, p, 8); return x; }
__device__ __forceinline__ i128 pq_pow10(int k) {
  i128 r = 1;
  for (int i = 0; i < k; i++) r *= 10;
  return r;
}
__device__ __forceinline__ i64 pq_uniform_i64(i64 v) {
  return (i64)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)((u64)v >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)v));
}

// Every block decodes one CONTIGUOUS slice of the column; a wave takes 64·R consecutive rows per step (every lane R consecutive rows
// of them), so the page and the hybrid run of a wave's first row only ever move FORWARD.
//   * The wave keeps its current page and run — and the row / value index at which the NEXT page / run starts — in wave-uniform
//     scalars: the common step needs no table load at all, only "am I still before the next boundary" compares.  (A PqPage / PqRun
//     struct copy lives in scratch memory: measured, every step paid scratch round trips.)
//   * R rows per lane would give R independent index → dictionary → store chains per lane and let a lane's bit-packed indices come out
//     of ONE 8-byte load; measured on MI355X (profiles/r2_parquet_decode.txt) R = 4 was no faster than R = 1 — with 256 rows per wave
//     step every second step crosses a ≤ 504-value run and sends lanes down the per-element path — so R = 1 is what runs.
//   * Elements beyond the wave's page or run (a boundary inside the 256 rows) find their page / run by stepping forward from the wave's.
// The value of an element is produced as a 128-bit payload whose low out_width bytes are stored.
struct PqElem { i128 val; bool ok; };

__device__ __forceinline__ i128 pq_convert(int kind, const u8* src, int width, int dec_up, u32 boolbit) {
  switch (kind) {
    case PQ_COPY4: return (i128)(u128)pq_ld32(src);
    case PQ_COPY8: return (i128)(u128)pq_ld64(src);
    case PQ_I32_TO_I64: return (i128)(i64)(i32)pq_ld32(src);
    case PQ_I32_TO_I16: return (i128)(i16)(i32)pq_ld32(src);
    case PQ_I32_TO_I8: return (i128)(i8)(i32)pq_ld32(src);
    case PQ_I32_TO_DEC: return (i128)(i32)pq_ld32(src) * pq_pow10(dec_up);
    case PQ_I64_TO_DEC: return (i128)(i64)pq_ld64(src) * pq_pow10(dec_up);
    case PQ_FLBA_TO_DEC: return pq_flba_to_i128(src, width) * pq_pow10(dec_up);
    case PQ_F32_TO_F64: { const double d = (double)__uint_as_float(pq_ld32(src)); return (i128)(u128)(u64)__double_as_longlong(d); }
    case PQ_I32_TO_F64: { const double d = (double)(i32)pq_ld32(src); return (i128)(u128)(u64)__double_as_longlong(d); }
    case PQ_INT96_TO_TS_MICROS:
      // INT96 = 8 bytes nanoseconds of day (LE) + 4 bytes Julian day (LE); 2440588 = Julian day of 1970-01-01
      return (i128)(((i64)(i32)pq_ld32(src + 8) - 2440588) * 86400000000ll + (i64)(pq_ld64(src) / 1000ull));
    case PQ_I64_MILLIS_TO_MICROS: return (i128)(i64)((u64)pq_ld64(src) * 1000ull);   // wrapping like arrow's cast kernel multiply
    case PQ_U32_TO_I64: return (i128)(u128)pq_ld32(src);
    case PQ_U64_TO_DEC: return (i128)(u128)(u64)pq_ld64(src) * pq_pow10(dec_up);
    case PQ_BOOL: return (i128)boolbit;
    case PQ_COPY1: return (i128)(u128)src[0];
    default: return 0;
  }
}

__global__ __launch_bounds__(256) void pq_decode_fixed_kernel(PqDecodeArgs a) {
  constexpr int R = 1;                                   // consecutive rows per lane
  constexpr i64 kTile = 256 * R;                         // rows per block and step (256 per wave)
  const i64 per_block = (((a.n_rows + gridDim.x - 1) / gridDim.x) + kTile - 1) / kTile * kTile;
  const i64 begin = (i64)blockIdx.x * per_block;
  const i64 end = begin + per_block < a.n_rows ? begin + per_block : a.n_rows;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = (int)(threadIdx.x & 63u);
  constexpr i64 kNever = 0x7fffffffffffffffll;
  const u8* __restrict__ bytes = a.bytes;
  const u8* __restrict__ dict = a.dict;
  int p = -1;                                   // the wave's page …
  i64 pg_row_start = 0, pg_values_off = 0, pg_dict_off = 0, next_page_row = -1;   // … and the first row of the page after it
  i32 pg_encoding = 0, pg_bit_width = 0, pg_kind = 0, pg_width = 0, pg_dec_up = 0, pg_idx_first = 0, pg_idx_count = 0, pg_val_skip = 0;
  int r = -1, run_page = -1;                    // the wave's run (dictionary pages) …
  i64 rn_byte_off = 0;
  i32 rn_value_start = 0, rn_is_rle = 0, next_run_val = 0;   // … and the first value index of the run after it (within the page)
  u32 rn_rle_value = 0;
  for (i64 base = begin; base < end; base += kTile) {
    const i64 row0 = base + (i64)wave * (64 * R);          // first of this wave's 256 rows (wave-uniform)
    if (row0 >= end) break;
    if (p < 0 || row0 >= next_page_row) {
      if (p < 0) p = pq_find_page(a.pages, a.npages, row0);
      else
        while (p + 1 < a.npages && a.pages[p + 1].row_start <= row0) p++;
      const PqPage* q = a.pages + p;
      pg_row_start = q->row_start; pg_values_off = q->values_off; pg_dict_off = q->dict_off;
      pg_encoding = q->encoding; pg_bit_width = q->bit_width; pg_kind = q->kind; pg_width = q->width; pg_dec_up = q->dec_scale_up;
      pg_idx_first = q->idx_run_first; pg_idx_count = q->idx_run_count; pg_val_skip = q->val_skip;
      next_page_row = p + 1 < a.npages ? a.pages[p + 1].row_start : kNever;
    }
    if (pg_encoding == 1) {
      // the run holding the wave's first value
      const i32 v0 = (a.max_def > 0 ? (i32)(a.vidx[row0] - a.vidx[pg_row_start]) : (i32)(row0 - pg_row_start)) + pg_val_skip;
      const int last0 = pg_idx_first + pg_idx_count - 1;
      bool reload = false;
      if (run_page != p) {
        int lo = pg_idx_first, hi = last0;
        while (lo < hi) {
          int mid = (lo + hi + 1) >> 1;
          if (a.idx_runs[mid].value_start <= v0) lo = mid;
          else hi = mid - 1;
        }
        r = lo;
        run_page = p;
        reload = true;
      } else if (v0 >= next_run_val) {
        while (r < last0 && a.idx_runs[r + 1].value_start <= v0) r++;
        reload = true;
      }
      if (reload) {
        const PqRun* q = a.idx_runs + r;
        rn_byte_off = q->byte_off; rn_value_start = q->value_start; rn_is_rle = q->is_rle; rn_rle_value = q->rle_value;
        next_run_val = r < last0 ? a.idx_runs[r + 1].value_start : 0x7fffffff;
      }
    }
    const i64 lrow = row0 + (i64)lane * R;                 // this lane's first row
    // ---- phase 1: where does every element's value come from (index loads happen here) ----
    const u8* src[R];
    u32 boolbit[R];
    i32 kind[R], width[R], dec_up[R];
    bool ok[R];
    // fast path: the lane's rows lie in the wave's page and bit-packed run and the column has no NULLs → ONE load holds all indices
    const bool no_nulls = a.max_def == 0;
    const i32 lv0 = (i32)(lrow - pg_row_start) + pg_val_skip;
    const bool all_here = lrow + R <= end && lrow + R <= next_page_row;
    const bool packed_fast = pg_encoding == 1 && no_nulls && all_here && !rn_is_rle && lv0 + R <= next_run_val && pg_bit_width * R <= 56;
    u64 packed = 0;
    if (packed_fast) {
      const i64 bit = (i64)(lv0 - rn_value_start) * pg_bit_width;
      packed = pq_ld64(bytes + rn_byte_off + (bit >> 3)) >> (bit & 7);
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
      const i64 row = lrow + k;
      ok[k] = row < end && (!a.valid_out || a.valid_out[row] != 0);
      src[k] = nullptr;
      boolbit[k] = 0;
      kind[k] = pg_kind; width[k] = pg_width; dec_up[k] = pg_dec_up;
      if (!ok[k]) continue;
      if (packed_fast) {
        const u32 idx = (u32)((packed >> (k * pg_bit_width)) & ((1ull << pg_bit_width) - 1));
        src[k] = dict + pg_dict_off + (i64)idx * pg_width;
        continue;
      }
      // the element's own page: the wave's, unless a page boundary falls inside this step
      i64 l_row_start = pg_row_start, l_values_off = pg_values_off, l_dict_off = pg_dict_off;
      i32 l_encoding = pg_encoding, l_bit_width = pg_bit_width, l_idx_first = pg_idx_first, l_idx_count = pg_idx_count, l_val_skip = pg_val_skip;
      const bool here = row < next_page_row;
      if (!here) {
        int qi = p;
        while (qi + 1 < a.npages && a.pages[qi + 1].row_start <= row) qi++;
        const PqPage* q = a.pages + qi;
        l_row_start = q->row_start; l_values_off = q->values_off; l_dict_off = q->dict_off;
        l_encoding = q->encoding; l_bit_width = q->bit_width; kind[k] = q->kind; width[k] = q->width; dec_up[k] = q->dec_scale_up;
        l_idx_first = q->idx_run_first; l_idx_count = q->idx_run_count; l_val_skip = q->val_skip;
      }
      const i32 v = (a.max_def > 0 ? (i32)(a.vidx[row] - a.vidx[l_row_start]) : (i32)(row - l_row_start)) + l_val_skip;
      if (l_encoding == 1) {
        u32 idx;
        if (here && v < next_run_val) {            // v ≥ rn_value_start holds: v ≥ v0 ≥ the run's first value
          if (rn_is_rle) {
            idx = rn_rle_value;
          } else {
            const int bw = l_bit_width;
            const i64 bit = (i64)(v - rn_value_start) * bw;
            const u64 w = pq_ld64(bytes + rn_byte_off + (bit >> 3));   // bw <= 32 → the value fits in 5 bytes; staging is padded
            idx = (u32)((w >> (bit & 7)) & ((bw >= 32) ? 0xffffffffull : ((1ull << bw) - 1)));
          }
        } else if (here) {
          // a later run of the wave's page: step forward from the wave's run
          int rr = r;
          const int last = l_idx_first + l_idx_count - 1;
          while (rr < last && a.idx_runs[rr + 1].value_start <= v) rr++;
          const PqRun* q = a.idx_runs + rr;
          if (q->is_rle) {
            idx = q->rle_value;
          } else {
            const int bw = l_bit_width;
            const i64 bit = (i64)(v - q->value_start) * bw;
            const u64 w = pq_ld64(bytes + q->byte_off + (bit >> 3));
            idx = (u32)((w >> (bit & 7)) & ((bw >= 32) ? 0xffffffffull : ((1ull << bw) - 1)));
          }
        } else {
          idx = pq_hybrid_value(a.idx_runs, l_idx_first, l_idx_count, bytes, l_bit_width, v);
        }
        src[k] = dict + l_dict_off + (i64)idx * width[k];
      } else if (kind[k] == PQ_BOOL) {
        boolbit[k] = (bytes[l_values_off + (v >> 3)] >> (v & 7)) & 1;
      } else {
        src[k] = bytes + l_values_off + (i64)v * width[k];
      }
    }
    // ---- phase 2: the values (dictionary / page loads), all issued before any store ----
    i128 val[R];
#pragma unroll
    for (int k = 0; k < R; k++) val[k] = ok[k] ? pq_convert(kind[k], src[k], width[k], dec_up[k], boolbit[k]) : (i128)0;   // NULL rows store zeros
    // ---- phase 3: stores (a lane's four values are contiguous) ----
#pragma unroll
    for (int k = 0; k < R; k++) {
      const i64 row = lrow + k;
      if (row >= end) continue;
      switch (a.out_width) {
        case 1: ((u8*)a.values_out)[row] = (u8)val[k]; break;
        case 2: ((u16*)a.values_out)[row] = (u16)val[k]; break;
        case 4: ((u32*)a.values_out)[row] = (u32)val[k]; break;
        case 8: ((u64*)a.values_out)[row] = (u64)val[k]; break;
        default: ((i128*)a.values_out)[row] = val[k]; break;
      }
    }
  }
}

// 3b. values, a RUN at a time (columns without NULLs): one wave per unit of work — a bit-packed run of dictionary indices (≤ 504 values as
// the writers emit them), an RLE run, or a 4096-value chunk of a PLAIN page.  Everything about the unit is wave-uniform (one PqRun and one
// PqPage, loaded once through the scalar path); lane l decodes values l, l + 64, l + 128, … of the unit, eight per pass with all index and
// dictionary loads issued before the first store, and consecutive lanes store consecutive rows.  The row-at-a-time kernel above spends a
// dependent index → dictionary → store chain per 64 rows and drops to a per-element path at every run boundary; here there are no
// boundaries inside a unit.
// (decoded columns are written once and read by a LATER kernel: streaming stores, which do not claim L2 lines the dictionary and the packed
// indices want; COMET_PQ_STORE=plain restores ordinary stores)
template <int OW>
__device__ __forceinline__ void pq_store(void* out, i64 row, i128 v) {
  if (OW == 1) ((u8*)out)[row] = (u8)v;
  else if (OW == 2) ((u16*)out)[row] = (u16)v;
  else if (OW == 4) __builtin_nontemporal_store((u32)v, (u32*)out + row);
  else if (OW == 8) __builtin_nontemporal_store((u64)v, (u64*)out + row);
  else {
    typedef u32 V4S __attribute__((ext_vector_type(4)));
    V4S x;
    x[0] = (u32)v; x[1] = (u32)((u128)v >> 32); x[2] = (u32)((u128)v >> 64); x[3] = (u32)((u128)v >> 96);
    __builtin_nontemporal_store(x, (V4S*)((i128*)out + row));
  }
}
// CV: how a source value becomes the stored value — the common conversions get straight-line code (the generic pq_convert pays a
// switch and, for decimals, a 128-bit multiply by 10^0 per value): 1 copy 4 bytes, 2 copy 8 bytes, 3 INT64 → Decimal128 without
// rescaling, 4 INT32 → Decimal128 without rescaling, 0 everything else
template <int CV>
__device__ __forceinline__ i128 pq_cv(int kind, const u8* src, int width, int dec_up) {
  if (CV == 1) return (i128)(u128)pq_ld32(src);
  if (CV == 2) return (i128)(u128)pq_ld64(src);
  if (CV == 3) return (i128)(i64)pq_ld64(src);
  if (CV == 4) return (i128)(i32)pq_ld32(src);
  return pq_convert(kind, src, width, dec_up, 0);
}
#define PQ_LDS __attribute__((address_space(3)))
constexpr int kUnitLdsBytes = 2048;        // a wave's LDS slice for one unit's packed indices (504 values × 32 bits = 2016 bytes)
template <int OW, int CV>
__device__ __forceinline__ void pq_decode_unit(const PqDecodeArgs& a, void* out, const PqRun& rn, const PqPage& pg, int lane, PQ_LDS u32* lds) {
  const u8* __restrict__ bytes = a.bytes;
  const u8* __restrict__ dict = a.dict + pg.dict_off;
  // where value 0 of the page goes: its first row — or, for a column with NULLs, its ordinal among the column's non-NULL values (the
  // caller put that in pg.row_start and passes the dense buffer as `out`)
  const i64 row0 = pg.row_start + rn.value_start;
  i32 count = rn.count;
  if (rn.value_start + count > pg.value_count) count = pg.value_count - rn.value_start;   // the last bit-packed group of a page is padded to 8 values
  const int kind = pg.kind, width = pg.width, dec_up = pg.dec_scale_up;
  // 4- and 8-byte outputs: a lane that stores ONE value per instruction moves 256 / 512 bytes per wave store — a quarter / half of what
  // the 16-byte outputs move.  When the unit's first row is 16-byte aligned in the output, every lane takes VN CONSECUTIVE values instead
  // and stores them as one 16-byte vector (1 KiB per wave store); l_shipdate-like columns (INT32 → Date32) ran at half the rate of the
  // Decimal128 ones before.
  constexpr int VN = OW == 4 ? 4 : OW == 8 ? 2 : 1;
  typedef u32 V4 __attribute__((ext_vector_type(4)));
  const bool vec = VN > 1 && ((((uintptr_t)out + (uintptr_t)row0 * (uintptr_t)OW) & 15u) == 0);
  auto store_vec = [&](i32 j0, const i128* v, i32 nvalid) {      // values j0 … j0 + VN − 1 of the unit (nvalid of them exist)
    if (nvalid >= VN) {
      V4 x;
      if (OW == 4) { x[0] = (u32)v[0]; x[1] = (u32)v[1]; x[2] = (u32)v[2]; x[3] = (u32)v[3]; }
      else { const u64 a = (u64)v[0], b = (u64)v[VN - 1]; x[0] = (u32)a; x[1] = (u32)(a >> 32); x[2] = (u32)b; x[3] = (u32)(b >> 32); }
      __builtin_nontemporal_store(x, (V4*)((u8*)out + (row0 + j0) * (i64)OW));
    } else {
      for (int k = 0; k < nvalid; k++) pq_store<OW>(out, row0 + j0 + k, v[k]);
    }
  };
  if (rn.is_rle == 1) {
    const i128 v = pq_cv<CV>(kind, dict + (i64)rn.rle_value * width, width, dec_up);
    if (vec) {
      i128 vv[VN];
      for (int k = 0; k < VN; k++) vv[k] = v;
      for (i32 j0 = lane * VN; j0 < count; j0 += 64 * VN) store_vec(j0, vv, count - j0 < VN ? count - j0 : VN);
      return;
    }
    for (i32 j = lane; j < count; j += 64) pq_store<OW>(out, row0 + j, v);
    return;
  }
  constexpr int U = 4;     // values per lane per pass: 4 keeps the kernel at 68 VGPRs (7 waves per SIMD); 8 needs 103 (4 waves) and measured slower
  if (rn.is_rle == 2) {
    // PLAIN values (booleans: one bit per value)
    const u8* src = bytes + rn.byte_off;
    if (vec && !(CV == 0 && kind == PQ_BOOL)) {
      constexpr int G = U / VN > 0 ? U / VN : 1;       // vectors per lane and pass
      for (i32 base = 0; base < count; base += 64 * VN * G) {
        i128 v[G][VN];
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
          for (int k = 0; k < VN; k++) {
            const i32 j = base + (g * 64 + lane) * VN + k;
            v[g][k] = j < count ? pq_cv<CV>(kind, src + (i64)j * width, width, dec_up) : (i128)0;
          }
#pragma unroll
        for (int g = 0; g < G; g++) {
          const i32 j0 = base + (g * 64 + lane) * VN;
          if (j0 < count) store_vec(j0, v[g], count - j0 < VN ? count - j0 : VN);
        }
      }
      return;
    }
    for (i32 base = 0; base < count; base += 64 * U) {
      i128 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const i32 j = base + u * 64 + lane;
        v[u] = 0;
        if (j < count) v[u] = (CV == 0 && kind == PQ_BOOL) ? (i128)((src[(j + rn.pad) >> 3] >> ((j + rn.pad) & 7)) & 1) : pq_cv<CV>(kind, src + (i64)j * width, width, dec_up);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const i32 j = base + u * 64 + lane;
        if (j < count) pq_store<OW>(out, row0 + j, v[u]);
      }
    }
    return;
  }
  // bit-packed dictionary indices.  What limits this path is the number of loads in flight, not bytes: one 4-byte load per 6-bit index
  // wastes a load slot on 6 bits.  So the wave first copies the unit's packed bytes (≤ 2 KiB for 504 values) into its LDS slice with
  // one or two 16-byte loads per lane, then every lane picks its indices out of LDS (values lane, lane + 64, …: consecutive lanes
  // still store consecutive rows).
  const int bw = pg.bit_width;
  const u8* packed = bytes + rn.byte_off;
  const i32 nbytes = (i32)(((i64)count * bw + rn.pad + 7) >> 3);
  if (nbytes <= kUnitLdsBytes) {
    for (i32 o = lane * 16; o < nbytes; o += 64 * 16) {
      u64 lo = pq_ld64(packed + o), hi = pq_ld64(packed + o + 8);      // (reads up to 15 bytes past the run: the staging is padded)
      lds[(o >> 2) + 0] = (u32)lo;
      lds[(o >> 2) + 1] = (u32)(lo >> 32);
      lds[(o >> 2) + 2] = (u32)hi;
      lds[(o >> 2) + 3] = (u32)(hi >> 32);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const u64 mask = bw >= 32 ? 0xffffffffull : ((1ull << bw) - 1);
    if (vec) {
      constexpr int G = U / VN > 0 ? U / VN : 1;
      for (i32 base = 0; base < count; base += 64 * VN * G) {
        u32 idx[G][VN];
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
          for (int k = 0; k < VN; k++) {
            const i32 j = base + (g * 64 + lane) * VN + k;
            const u32 bit = (u32)j * (u32)bw + (u32)rn.pad;
            const u32 wi = bit >> 5;
            idx[g][k] = j < count ? (u32)(((((u64)lds[wi + 1]) << 32 | lds[wi]) >> (bit & 31)) & mask) : 0u;
          }
        i128 v[G][VN];
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
          for (int k = 0; k < VN; k++) v[g][k] = pq_cv<CV>(kind, dict + (i64)idx[g][k] * width, width, dec_up);
#pragma unroll
        for (int g = 0; g < G; g++) {
          const i32 j0 = base + (g * 64 + lane) * VN;
          if (j0 < count) store_vec(j0, v[g], count - j0 < VN ? count - j0 : VN);
        }
      }
      __builtin_amdgcn_wave_barrier();      // the slice is overwritten by the wave's next unit
      return;
    }
    for (i32 base = 0; base < count; base += 64 * U) {
      u32 idx[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const i32 j = base + u * 64 + lane;
        const u32 bit = (u32)j * (u32)bw + (u32)rn.pad;
        const u32 wi = bit >> 5;
        idx[u] = j < count ? (u32)(((((u64)lds[wi + 1]) << 32 | lds[wi]) >> (bit & 31)) & mask) : 0u;
      }
      i128 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = pq_cv<CV>(kind, dict + (i64)idx[u] * width, width, dec_up);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const i32 j = base + u * 64 + lane;
        if (j < count) pq_store<OW>(out, row0 + j, v[u]);
      }
    }
    __builtin_amdgcn_wave_barrier();      // the slice is overwritten by the wave's next unit
    return;
  }
  // a run longer than the LDS slice (writers do not emit them; a hand-made file could): indices straight from memory
  for (i32 base = 0; base < count; base += 64 * U) {
    u32 idx[U];
    const u64 mask = bw >= 32 ? 0xffffffffull : ((1ull << bw) - 1);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const i32 j = base + u * 64 + lane;
      const i64 bit = (i64)j * bw + rn.pad;
      idx[u] = j < count ? (u32)((pq_ld64(packed + (bit >> 3)) >> (bit & 7)) & mask) : 0u;
    }
    i128 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = pq_cv<CV>(kind, dict + (i64)idx[u] * width, width, dec_up);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const i32 j = base + u * 64 + lane;
      if (j < count) pq_store<OW>(out, row0 + j, v[u]);
    }
  }
}
__device__ __forceinline__ PqRun pq_load_run_uniform(const PqRun* rp) {
  PqRun rn;
  rn.byte_off = pq_uniform_i64(rp->byte_off);
  rn.value_start = __builtin_amdgcn_readfirstlane(rp->value_start);
  rn.count = __builtin_amdgcn_readfirstlane(rp->count);
  rn.is_rle = __builtin_amdgcn_readfirstlane(rp->is_rle);
  rn.rle_value = (u32)__builtin_amdgcn_readfirstlane((int)rp->rle_value);
  rn.page = __builtin_amdgcn_readfirstlane(rp->page);
  rn.pad = __builtin_amdgcn_readfirstlane(rp->pad);     // first bit of the unit inside its first byte (units clipped to a kept piece of a pruned page)
  return rn;
}
__global__ __launch_bounds__(256) void pq_decode_runs_kernel(PqDecodeArgs a) {
  const int lane = (int)(threadIdx.x & 63u);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // every wave takes a CONTIGUOUS range of units: consecutive units share their page (its entry is reloaded only when the page changes),
  // read adjacent index bytes and write adjacent rows; the next unit's table entry is loaded while the current unit is being decoded,
  // so the only dependent memory accesses left per unit are index → dictionary → store
  __shared__ u32 s_idx[4][kUnitLdsBytes / 4 + 8];
  PQ_LDS u32* lds = (PQ_LDS u32*)s_idx[wave];      // address_space(3): ds_read / ds_write, not FLAT (a FLAT access waits for every outstanding global load)
  const i64 nwaves = (i64)gridDim.x * 4;
  const i64 per = ((i64)a.n_idx_runs + nwaves - 1) / nwaves;
  const i64 w = (i64)blockIdx.x * 4 + wave;
  const i64 begin = w * per, end = begin + per < a.n_idx_runs ? begin + per : (i64)a.n_idx_runs;
  if (begin >= end) return;
  int cur_page = -1;
  PqPage pg;
  void* out = a.dense_out ? a.dense_out : a.values_out;
  PqRun next = pq_load_run_uniform(a.idx_runs + begin);
  for (i64 r = begin; r < end; r++) {
    const PqRun rn = next;
    if (r + 1 < end) next = pq_load_run_uniform(a.idx_runs + r + 1);
    if (rn.page != cur_page) {
      const PqPage* pp = a.pages + rn.page;
      pg.row_start = pq_uniform_i64(pp->row_start);
      if (a.dense_out) pg.row_start = (i64)(u32)__builtin_amdgcn_readfirstlane((int)a.vidx[pg.row_start]);   // non-NULL values before the page
      pg.dict_off = pq_uniform_i64(pp->dict_off);
      pg.value_count = __builtin_amdgcn_readfirstlane(pp->value_count);
      pg.bit_width = __builtin_amdgcn_readfirstlane(pp->bit_width);
      pg.kind = __builtin_amdgcn_readfirstlane(pp->kind);
      pg.width = __builtin_amdgcn_readfirstlane(pp->width);
      pg.dec_scale_up = __builtin_amdgcn_readfirstlane(pp->dec_scale_up);
      cur_page = rn.page;
    }
    // the conversion class is wave-uniform: one branch per unit, straight-line code inside
    const int cv = pg.kind == PQ_COPY4 ? 1 : pg.kind == PQ_COPY8 ? 2 : (pg.kind == PQ_I64_TO_DEC && pg.dec_scale_up == 0) ? 3 : (pg.kind == PQ_I32_TO_DEC && pg.dec_scale_up == 0) ? 4 : 0;
    if (cv == 1 && a.out_width == 4) pq_decode_unit<4, 1>(a, out, rn, pg, lane, lds);
    else if (cv == 2 && a.out_width == 8) pq_decode_unit<8, 2>(a, out, rn, pg, lane, lds);
    else if (cv == 3 && a.out_width == 16) pq_decode_unit<16, 3>(a, out, rn, pg, lane, lds);
    else if (cv == 4 && a.out_width == 16) pq_decode_unit<16, 4>(a, out, rn, pg, lane, lds);
    else
      switch (a.out_width) {
        case 1: pq_decode_unit<1, 0>(a, out, rn, pg, lane, lds); break;
        case 2: pq_decode_unit<2, 0>(a, out, rn, pg, lane, lds); break;
        case 4: pq_decode_unit<4, 0>(a, out, rn, pg, lane, lds); break;
        case 8: pq_decode_unit<8, 0>(a, out, rn, pg, lane, lds); break;
        default: pq_decode_unit<16, 0>(a, out, rn, pg, lane, lds); break;
      }
  }
}

// a column with NULLs decoded run by run: the values sit densely in dense_out by their ordinal; row r takes value vidx[r] if it is valid
__global__ __launch_bounds__(256) void pq_expand_nulls_kernel(PqDecodeArgs a) {
  for (i64 row = (i64)blockIdx.x * 256 + threadIdx.x; row < a.n_rows; row += (i64)gridDim.x * 256) {
    const bool ok = a.valid_out[row] != 0;
    const i64 v = a.vidx[row];
    switch (a.out_width) {
      case 1: ((u8*)a.values_out)[row] = ok ? ((const u8*)a.dense_out)[v] : (u8)0; break;
      case 2: ((u16*)a.values_out)[row] = ok ? ((const u16*)a.dense_out)[v] : (u16)0; break;
      case 4: ((u32*)a.values_out)[row] = ok ? ((const u32*)a.dense_out)[v] : 0u; break;
      case 8: ((u64*)a.values_out)[row] = ok ? ((const u64*)a.dense_out)[v] : 0ull; break;
      default: ((i128*)a.values_out)[row] = ok ? ((const i128*)a.dense_out)[v] : (i128)0; break;
    }
  }
}

// 4. strings: lengths, then (after the host-driven offset scan) bytes
__device__ __forceinline__ void pq_string_ref(const PqDecodeArgs& a, i64 row, const u8*& p, u32& len) {
  const PqPage pg = a.pages[pq_find_page(a.pages, a.npages, row)];
  const i32 v = (a.max_def > 0 ? (i32)(a.vidx[row] - a.vidx[pg.row_start]) : (i32)(row - pg.row_start)) + pg.val_skip;
  if (pg.encoding == 1) {
    u32 idx = pq_hybrid_value(a.idx_runs, pg.idx_run_first, pg.idx_run_count, a.bytes, pg.bit_width, v);
    const i32* doffs = a.dict_offs + pg.dict_offs_first;
    p = a.dict + pg.dict_off + doffs[idx];
    len = (u32)(doffs[idx + 1] - doffs[idx]);
  } else {
    // PLAIN BYTE_ARRAY: value bytes start 4 bytes after the previous value's end; offsets were prescanned on the host
    const i64 o = a.plain_str_offs[pg.str_first + v];
    p = a.bytes + o;
    len = (u32)a.bytes[o - 4] | ((u32)a.bytes[o - 3] << 8) | ((u32)a.bytes[o - 2] << 16) | ((u32)a.bytes[o - 1] << 24);
  }
}
__global__ __launch_bounds__(256) void pq_string_lengths_kernel(PqDecodeArgs a) {
  for (i64 row = (i64)blockIdx.x * 256 + threadIdx.x; row < a.n_rows; row += (i64)gridDim.x * 256) {
    u32 len = 0;
    if (!a.valid_out || a.valid_out[row]) {
      const u8* p;
      pq_string_ref(a, row, p, len);
    }
    a.lengths_out[row] = len;
  }
}
__global__ __launch_bounds__(256) void pq_string_copy_kernel(PqDecodeArgs a) {
  for (i64 row = (i64)blockIdx.x * 256 + threadIdx.x; row < a.n_rows; row += (i64)gridDim.x * 256) {
    if (a.valid_out && !a.valid_out[row]) continue;
    const u8* p;
    u32 len;
    pq_string_ref(a, row, p, len);
    u8* dst = a.str_bytes_out + a.str_offsets[row];
    for (u32 k = 0; k < len; k++) dst[k] = p[k];
  }
}

// generic exclusive scan of u32 → i32 offsets (n+1 entries), tiles of 1024
__global__ __launch_bounds__(256) void pq_u32_tile_sum_kernel(const u32* in, i64 n, u64* tile_sums) {
  const i64 ntiles = (n + 1023) / 1024;
  __shared__ u64 s_sum;
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    u64 local = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      i64 i = t * 1024 + r * 256 + threadIdx.x;
      if (i < n) local += in[i];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) local += shfl_xor_u64(local, m);
    if (lane_id() == 0) atomicAdd((unsigned long long*)&s_sum, (unsigned long long)local);
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[t] = s_sum;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void pq_u32_scan_apply_kernel(const u32* in, i64 n, const u64* tile_off, i32* out) {
  const i64 ntiles = (n + 1023) / 1024;
  __shared__ u64 s_wave[4];
  __shared__ u64 s_run;
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int r = 0; r < 4; r++) {
      i64 i = t * 1024 + r * 256 + threadIdx.x;
      u64 v = i < n ? in[i] : 0, x = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        u64 y = shfl_u64(x, lane_id() - d < 0 ? 0 : lane_id() - d);
        if (lane_id() >= d) x += y;
      }
      if (lane_id() == 63) s_wave[wave_id()] = x;
      __syncthreads();
      u64 woff = 0;
      for (int w = 0; w < wave_id(); w++) woff += s_wave[w];
      const u64 run = s_run;
      if (i < n) out[i] = (i32)(tile_off[t] + run + woff + x - v);
      __syncthreads();
      if (threadIdx.x == 255) s_run = run + woff + x;
      __syncthreads();
    }
  }
  // the grand total is kept in 64 bits: more than INT32_MAX comes back as -1 (every caller turns a negative total into its "exceeds
  // 2 GiB" error) instead of wrapping to a small positive size
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tile_off[ntiles] > 0x7fffffffull ? (i32)-1 : (i32)tile_off[ntiles];
}
__global__ __launch_bounds__(256) void pq_pack_kernel(const u8* bytes, u8* bitmap, i64 n) { pack_validity_body(bytes, bitmap, n); }

static int grid_slices(i64 n) {   // contiguous slices of >= 4 tiles, enough blocks to fill 256 CUs several times over
  i64 g = (n + 1023) / 1024;
  return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}
static int grid_rows(i64 n) {
  i64 g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
static int grid_tiles(i64 n) {
  i64 g = (n + 1023) / 1024;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// Pulls slices of pinned host memory into HBM: workgroup (x, y) moves the x-th 64 KiB piece of slice y — sixteen 16-byte loads per thread
// in flight across PCIe, then sixteen stores.  One launch moves every slice that is ready (parquet_scan.cpp: a hipMemcpyAsync per slice
// costs ≈ 40 µs of latency whatever its size).
__global__ __launch_bounds__(256) void pq_upload_kernel(const PqCopyDesc* __restrict__ descs) {
  typedef unsigned V4 __attribute__((vector_size(16)));
  const PqCopyDesc d = descs[blockIdx.y];
  const u64 piece = (u64)blockIdx.x << 16;
  if (piece >= d.len) return;
  const u64 nvec = (d.len - piece < 65536 ? d.len - piece : 65536) >> 4;
  const V4* src = (const V4*)(d.src + piece);
  V4* dst = (V4*)(d.dst + piece);
  V4 v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const u64 i = (u64)threadIdx.x + (u64)k * 256;
    if (i < nvec) v[k] = __builtin_nontemporal_load(src + i);
  }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const u64 i = (u64)threadIdx.x + (u64)k * 256;
    if (i < nvec) dst[i] = v[k];
  }
}

extern "C" {
void pq_launch_validity(const PqDecodeArgs* a, void* st) { hipLaunchKernelGGL(pq_validity_kernel, grid_rows(a->n_rows), 256, 0, (hipStream_t)st, *a); }
void pq_launch_levels(const PqDecodeArgs* a, int which, uint8_t* out, void* st) { hipLaunchKernelGGL(pq_levels_kernel, grid_rows(a->n_rows), 256, 0, (hipStream_t)st, *a, which, (u8*)out); }
void pq_launch_level_ge(const uint8_t* lv, int64_t n, int thr, uint8_t* out, void* st) {
  if (n > 0) hipLaunchKernelGGL(pq_level_ge_kernel, grid_rows(n), 256, 0, (hipStream_t)st, (const u8*)lv, (i64)n, thr, (u8*)out);
}
void pq_launch_list_flags(const uint8_t* def, const uint8_t* rep, int64_t n, int def_slot, uint32_t* starts, uint32_t* elems, void* st) {
  if (n > 0) hipLaunchKernelGGL(pq_list_flags_kernel, grid_rows(n), 256, 0, (hipStream_t)st, (const u8*)def, (const u8*)rep, (i64)n, def_slot, (u32*)starts, (u32*)elems);
}
void pq_launch_list_assemble(const uint8_t* def, const uint8_t* rep, int64_t n, int64_t rows, int def_list, int def_slot, int max_def, const int32_t* start_idx, const int32_t* elem_idx,
                             const uint8_t* values, int width, int32_t* offsets, uint8_t* list_valid, uint8_t* elem_valid, uint8_t* elem_values, uint32_t* err, void* st) {
  hipLaunchKernelGGL(pq_list_assemble_kernel, grid_rows(n + 1), 256, 0, (hipStream_t)st, (const u8*)def, (const u8*)rep, (i64)n, (i64)rows, def_list, def_slot, max_def, (const i32*)start_idx,
                     (const i32*)elem_idx, (const u8*)values, width, (i32*)offsets, (u8*)list_valid, (u8*)elem_valid, (u8*)elem_values, (u32*)err);
}
void pq_launch_list_elem_entries(const uint8_t* def, int64_t n, int def_slot, const int32_t* elem_idx, uint32_t* entries, void* st) {
  if (n > 0) hipLaunchKernelGGL(pq_list_elem_entries_kernel, grid_rows(n), 256, 0, (hipStream_t)st, (const u8*)def, (i64)n, def_slot, (const i32*)elem_idx, (u32*)entries);
}
void pq_launch_vidx(const uint8_t* valid, int64_t n, uint64_t* tiles, uint32_t* vidx, void* st) {
  hipStream_t s = (hipStream_t)st;
  hipLaunchKernelGGL(pq_tile_count_kernel, grid_tiles(n), 256, 0, s, valid, (i64)n, (u64*)tiles);
  hipLaunchKernelGGL(pq_tile_scan_kernel, 1, 256, 0, s, (u64*)tiles, (i64)((n + 1023) / 1024));
  hipLaunchKernelGGL(pq_vidx_kernel, grid_tiles(n), 256, 0, s, valid, (i64)n, (const u64*)tiles, (u32*)vidx);
}
void pq_launch_store_u32(const uint32_t* src, uint32_t* dst, void* st) { hipLaunchKernelGGL(pq_store_u32_kernel, 1, 64, 0, (hipStream_t)st, (const u32*)src, (u32*)dst); }
void pq_launch_count_runs(const PqPendingRuns* pend, int n, const uint8_t* bytes, uint32_t* counts, uint32_t* err, void* st) {
  if (n > 0) hipLaunchKernelGGL(pq_count_runs_kernel, (n + 255) / 256, 256, 0, (hipStream_t)st, pend, n, (const u8*)bytes, (u32*)counts, (u32*)err);
}
void pq_launch_write_runs(const PqPendingRuns* pend, int n, const uint8_t* bytes, const int32_t* offsets, int32_t run_base, PqRun* runs, PqPage* pages, void* st) {
  if (n > 0) hipLaunchKernelGGL(pq_write_runs_kernel, (n + 255) / 256, 256, 0, (hipStream_t)st, pend, n, (const u8*)bytes, (const i32*)offsets, (i32)run_base, runs, pages);
}
void pq_launch_decode_runs(const PqDecodeArgs* a, void* st) {
  if (a->n_idx_runs <= 0) return;
  static const int grid_mul = getenv("COMET_PQ_GRID_MUL") ? std::max(1, atoi(getenv("COMET_PQ_GRID_MUL"))) : 16;
  const int blocks = (int)std::min<i64>(((i64)a->n_idx_runs + 3) / 4, (i64)256 * 16 * grid_mul);
  hipLaunchKernelGGL(pq_decode_runs_kernel, blocks, 256, 0, (hipStream_t)st, *a);
}
void pq_launch_expand_nulls(const PqDecodeArgs* a, void* st) { hipLaunchKernelGGL(pq_expand_nulls_kernel, grid_rows(a->n_rows), 256, 0, (hipStream_t)st, *a); }
void pq_launch_decode_fixed(const PqDecodeArgs* a, void* st) { hipLaunchKernelGGL(pq_decode_fixed_kernel, grid_slices(a->n_rows), 256, 0, (hipStream_t)st, *a); }
void pq_launch_string_lengths(const PqDecodeArgs* a, void* st) { hipLaunchKernelGGL(pq_string_lengths_kernel, grid_rows(a->n_rows), 256, 0, (hipStream_t)st, *a); }
void pq_launch_string_copy(const PqDecodeArgs* a, void* st) { hipLaunchKernelGGL(pq_string_copy_kernel, grid_rows(a->n_rows), 256, 0, (hipStream_t)st, *a); }
void pq_launch_u32_scan(const uint32_t* in, int64_t n, uint64_t* tiles, int32_t* out, void* st) {
  hipStream_t s = (hipStream_t)st;
  hipLaunchKernelGGL(pq_u32_tile_sum_kernel, grid_tiles(n), 256, 0, s, in, (i64)n, (u64*)tiles);
  hipLaunchKernelGGL(pq_tile_scan_kernel, 1, 256, 0, s, (u64*)tiles, (i64)((n + 1023) / 1024));
  hipLaunchKernelGGL(pq_u32_scan_apply_kernel, grid_tiles(n), 256, 0, s, in, (i64)n, (const u64*)tiles, (i32*)out);
}
void pq_launch_pack(const uint8_t* bytes, uint8_t* bitmap, int64_t n, void* st) {
  hipLaunchKernelGGL(pq_pack_kernel, grid_rows(n), 256, 0, (hipStream_t)st, bytes, bitmap, (i64)n);
}
void pq_launch_upload(const PqCopyDesc* descs, int n, void* st) {
  if (n <= 0) return;
  uint64_t longest = 0;
  for (int i = 0; i < n; i++) longest = std::max<uint64_t>(longest, descs[i].len);      // (the descriptors sit in pinned host memory: the host reads them too)
  if (!longest) return;
  hipLaunchKernelGGL(pq_upload_kernel, dim3((unsigned)((longest + 65535) >> 16), (unsigned)n), 256, 0, (hipStream_t)st, descs);
}
}
