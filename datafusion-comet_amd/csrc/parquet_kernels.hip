// Parquet page decode on the device: definition levels → validity, RLE/bit-packed hybrid
// dictionary indices, PLAIN and RLE_DICTIONARY values → Arrow buffers.  One lane per ROW of a column chunk; the
// host only parses page headers and hybrid run headers (PqPage / PqRun tables), never a value.
// Format: Apache Parquet spec (Encodings.md: PLAIN = 0, RLE = 3, RLE_DICTIONARY = 8; hybrid runs are
// <varint header><payload>, header LSB 1 = bit-packed groups of 8, LSB 0 = RLE).  The reference delegates this to the
// `parquet` 58.4.0 crate (native/core/src/parquet/parquet_exec.rs:145-147); pyarrow is the independent checker.
#include <hip/hip_runtime.h>

#include "device/comet_device.hpp"
#include "parquet_dev.h"

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
        u32 lvl = pq_hybrid_value(a.def_runs, pg.def_run_first, pg.def_run_count, a.bytes, bw, (i32)(row - pg.row_start));
        valid = lvl == (u32)a.max_def;
      }
    }
    a.valid_out[row] = valid;
  }
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

// Every block decodes one CONTIGUOUS slice of the column, tile after tile, and the 64 rows of a wave are consecutive.
// So the page and the hybrid run of a wave's first row only ever move FORWARD: they are found by binary search once per
// block and then advanced with wave-uniform loads (amortised O(1) per tile), and each lane steps forward from the wave's
// run to its own.  (One lane per row with two binary searches per row was latency-bound at ~1 TB/s.)
__global__ __launch_bounds__(256) void pq_decode_fixed_kernel(PqDecodeArgs a) {
  const i64 per_block = (((a.n_rows + gridDim.x - 1) / gridDim.x) + 255) / 256 * 256;
  const i64 begin = (i64)blockIdx.x * per_block;
  const i64 end = begin + per_block < a.n_rows ? begin + per_block : a.n_rows;
  const int wave_first = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
  int p0 = -1, run0 = -1, run_page = -1;
  for (i64 base = begin; base < end; base += 256) {
    const i64 row0 = base + wave_first;                      // first row of this wave's 64 (wave-uniform)
    if (row0 >= end) break;
    const i64 row = base + threadIdx.x;
    const bool in_range = row < end;
    if (p0 < 0) p0 = pq_find_page(a.pages, a.npages, row0);
    else
      while (p0 + 1 < a.npages && a.pages[p0 + 1].row_start <= row0) p0++;
    int p = p0;
    if (in_range)
      while (p + 1 < a.npages && a.pages[p + 1].row_start <= row) p++;
    const bool valid = in_range && (!a.valid_out || a.valid_out[row] != 0);   // valid_out == NULL: the column has no NULLs
    const u8* src = nullptr;
    u32 boolbit = 0;
    const PqPage pg0 = a.pages[p0];
    if (pg0.encoding == 1) {
      const i32 v0 = a.max_def > 0 ? (i32)(a.vidx[row0] - a.vidx[pg0.row_start]) : (i32)(row0 - pg0.row_start);
      const int last0 = pg0.idx_run_first + pg0.idx_run_count - 1;
      if (run_page != p0) {
        int lo = pg0.idx_run_first, hi = last0;
        while (lo < hi) {
          int mid = (lo + hi + 1) >> 1;
          if (a.idx_runs[mid].value_start <= v0) lo = mid;
          else hi = mid - 1;
        }
        run0 = lo;
        run_page = p0;
      } else {
        while (run0 < last0 && a.idx_runs[run0 + 1].value_start <= v0) run0++;
      }
    } else {
      run_page = -1;
    }
    int kind = 0, dec_up = 0;
    if (valid) {
      const PqPage pg = a.pages[p];
      kind = pg.kind;
      dec_up = pg.dec_scale_up;
      const i32 v = a.max_def > 0 ? (i32)(a.vidx[row] - a.vidx[pg.row_start]) : (i32)(row - pg.row_start);
      if (pg.encoding == 1) {
        u32 idx;
        if (p == p0 && run_page == p0) {
          int r = run0;
          const int last = pg.idx_run_first + pg.idx_run_count - 1;
          while (r < last && a.idx_runs[r + 1].value_start <= v) r++;
          const PqRun rn = a.idx_runs[r];
          if (rn.is_rle) {
            idx = rn.rle_value;
          } else {
            const int bw = pg.bit_width;
            const i64 bit = (i64)(v - rn.value_start) * bw;
            const u64 w = pq_ld64(a.bytes + rn.byte_off + (bit >> 3));   // bw <= 32 → the value fits in 5 bytes; staging is padded
            idx = (u32)((w >> (bit & 7)) & ((bw >= 32) ? 0xffffffffull : ((1ull << bw) - 1)));
          }
        } else {
          idx = pq_hybrid_value(a.idx_runs, pg.idx_run_first, pg.idx_run_count, a.bytes, pg.bit_width, v);
        }
        src = a.dict + pg.dict_off + (i64)idx * pg.width;
      } else if (pg.kind == PQ_BOOL) {
        boolbit = (a.bytes[pg.values_off + (v >> 3)] >> (v & 7)) & 1;
      } else {
        src = a.bytes + pg.values_off + (i64)v * pg.width;
      }
    }
    if (!in_range) continue;
    if (!valid) {
      // NULL rows (and rows of pages that carry no value) store zeros of the column's output width
      switch (a.out_width) {
        case 1: ((u8*)a.values_out)[row] = 0; break;
        case 2: ((u16*)a.values_out)[row] = 0; break;
        case 4: ((u32*)a.values_out)[row] = 0u; break;
        case 8: ((u64*)a.values_out)[row] = 0ull; break;
        default: ((i128*)a.values_out)[row] = (i128)0; break;
      }
      continue;
    }
    switch (kind) {
      case PQ_COPY4: ((u32*)a.values_out)[row] = pq_ld32(src); break;
      case PQ_COPY8: ((u64*)a.values_out)[row] = pq_ld64(src); break;
      case PQ_I32_TO_I64: ((i64*)a.values_out)[row] = (i64)(i32)pq_ld32(src); break;
      case PQ_I32_TO_I16: ((i16*)a.values_out)[row] = (i16)(i32)pq_ld32(src); break;
      case PQ_I32_TO_I8: ((i8*)a.values_out)[row] = (i8)(i32)pq_ld32(src); break;
      case PQ_I32_TO_DEC: ((i128*)a.values_out)[row] = (i128)(i32)pq_ld32(src) * pq_pow10(dec_up); break;
      case PQ_I64_TO_DEC: ((i128*)a.values_out)[row] = (i128)(i64)pq_ld64(src) * pq_pow10(dec_up); break;
      case PQ_FLBA_TO_DEC: ((i128*)a.values_out)[row] = pq_flba_to_i128(src, a.pages[p].width) * pq_pow10(dec_up); break;
      case PQ_F32_TO_F64: ((double*)a.values_out)[row] = (double)__uint_as_float(pq_ld32(src)); break;
      case PQ_I32_TO_F64: ((double*)a.values_out)[row] = (double)(i32)pq_ld32(src); break;
      case PQ_INT96_TO_TS_MICROS: {
        // INT96 = 8 bytes nanoseconds of day (LE) + 4 bytes Julian day (LE); 2440588 = Julian day of 1970-01-01
        ((i64*)a.values_out)[row] = ((i64)(i32)pq_ld32(src + 8) - 2440588) * 86400000000ll + (i64)(pq_ld64(src) / 1000ull);
        break;
      }
      case PQ_I64_MILLIS_TO_MICROS: ((i64*)a.values_out)[row] = (i64)((u64)pq_ld64(src) * 1000ull); break;   // wrapping like arrow's cast kernel multiply
      case PQ_U32_TO_I64: ((i64*)a.values_out)[row] = (i64)(u64)pq_ld32(src); break;
      case PQ_U64_TO_DEC: ((i128*)a.values_out)[row] = (i128)(u128)(u64)pq_ld64(src) * pq_pow10(dec_up); break;
      case PQ_BOOL: ((u8*)a.values_out)[row] = (u8)boolbit; break;
      default: break;
    }
  }
}

// 4. strings: lengths, then (after the host-driven offset scan) bytes
__device__ __forceinline__ void pq_string_ref(const PqDecodeArgs& a, i64 row, const u8*& p, u32& len) {
  const PqPage pg = a.pages[pq_find_page(a.pages, a.npages, row)];
  const i32 v = a.max_def > 0 ? (i32)(a.vidx[row] - a.vidx[pg.row_start]) : (i32)(row - pg.row_start);
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

extern "C" {
void pq_launch_validity(const PqDecodeArgs* a, void* st) { hipLaunchKernelGGL(pq_validity_kernel, grid_rows(a->n_rows), 256, 0, (hipStream_t)st, *a); }
void pq_launch_vidx(const uint8_t* valid, int64_t n, uint64_t* tiles, uint32_t* vidx, void* st) {
  hipStream_t s = (hipStream_t)st;
  hipLaunchKernelGGL(pq_tile_count_kernel, grid_tiles(n), 256, 0, s, valid, (i64)n, (u64*)tiles);
  hipLaunchKernelGGL(pq_tile_scan_kernel, 1, 256, 0, s, (u64*)tiles, (i64)((n + 1023) / 1024));
  hipLaunchKernelGGL(pq_vidx_kernel, grid_tiles(n), 256, 0, s, valid, (i64)n, (const u64*)tiles, (u32*)vidx);
}
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
}
