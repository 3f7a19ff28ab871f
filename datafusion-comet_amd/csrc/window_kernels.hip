// Window operator (operator.proto:857-862; planner.rs:2267-2379): ranking and offset functions over input that arrives sorted by
// (partition keys, order keys) — Spark plans the Sort below the Window, the reference's WindowAggExec relies on it as well.
//
// Everything derives from two flag columns computed on the order-preserving key bytes of adjacent rows (the Sort key kernel):
//   new_part[i]  row i starts a partition     new_peer[i]  row i starts a peer group (a partition start, or its ORDER BY key differs)
// Exclusive prefix sums of the flags give every row its partition index and peer-group index; scattering the start rows by those
// indices gives first_part[] / first_peer[] (with n as the sentinel one past the end).  All ranking functions are then closed forms:
//   row_number = i − ps + 1        rank = peer_start − ps + 1        dense_rank = g − g(ps) + 1
//   percent_rank = (rank − 1) / (rows − 1)      cume_dist = (peer_end − ps) / rows      ntile(k): Spark's bucket formula
// and lag / lead are a gather with index i ∓ k when that row lies inside the partition (otherwise NULL).
#include <hip/hip_runtime.h>

#include <cstring>

#include "device/comet_device.hpp"

using namespace comet;

namespace {

int grid_for(i64 n) {
  i64 g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

__global__ __launch_bounds__(256) void window_flags_kernel(const u8* __restrict__ pp, int Wp, const u8* __restrict__ po, int Wo, i64 n,
                                                           u32* __restrict__ fpart, u32* __restrict__ fpeer) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    bool np = i == 0;
    for (int p = 0; p < Wp && !np; p++) np = pp[(size_t)p * (size_t)n + (size_t)i] != pp[(size_t)p * (size_t)n + (size_t)i - 1];
    bool ng = np;
    for (int p = 0; p < Wo && !ng; p++) ng = po[(size_t)p * (size_t)n + (size_t)i] != po[(size_t)p * (size_t)n + (size_t)i - 1];
    fpart[i] = np ? 1u : 0u;
    fpeer[i] = ng ? 1u : 0u;
  }
}

// sp / sg: exclusive prefix sums (n + 1 entries) of the flags
__global__ __launch_bounds__(256) void window_first_kernel(const u32* __restrict__ fpart, const i32* __restrict__ sp, const u32* __restrict__ fpeer,
                                                           const i32* __restrict__ sg, i64 n, u32* __restrict__ first_part, u32* __restrict__ first_peer) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i <= n; i += (i64)gridDim.x * 256) {
    if (i == n) {   // sentinels one past the last partition / peer group
      first_part[sp[n]] = (u32)n;
      first_peer[sg[n]] = (u32)n;
      continue;
    }
    if (fpart[i]) first_part[sp[i + 1] - 1] = (u32)i;
    if (fpeer[i]) first_peer[sg[i + 1] - 1] = (u32)i;
  }
}

enum { W_ROW_NUMBER = 0, W_RANK = 1, W_DENSE_RANK = 2, W_PERCENT_RANK = 3, W_CUME_DIST = 4, W_NTILE = 5 };

__global__ __launch_bounds__(256) void window_rank_kernel(int kind, i64 arg, const i32* __restrict__ sp, const i32* __restrict__ sg,
                                                          const u32* __restrict__ first_part, const u32* __restrict__ first_peer, i64 n,
                                                          void* __restrict__ out) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const i32 p = sp[i + 1] - 1, g = sg[i + 1] - 1;
    const i64 ps = first_part[p], pe = first_part[p + 1], gs = first_peer[g], ge = first_peer[g + 1];
    const i64 rows = pe - ps, rank = gs - ps + 1;
    switch (kind) {
      case W_ROW_NUMBER: ((i32*)out)[i] = (i32)(i - ps + 1); break;
      case W_RANK: ((i32*)out)[i] = (i32)rank; break;
      case W_DENSE_RANK: ((i32*)out)[i] = (i32)(g - (sg[ps + 1] - 1) + 1); break;
      case W_PERCENT_RANK: ((double*)out)[i] = rows > 1 ? (double)(rank - 1) / (double)(rows - 1) : 0.0; break;
      case W_CUME_DIST: ((double*)out)[i] = (double)(ge - ps) / (double)rows; break;
      case W_NTILE: {
        // Spark NTile: the first (rows % k) buckets hold one row more than the others
        const i64 k = arg, i0 = i - ps, q = rows / k, r = rows % k, thr = r * (q + 1);
        ((i32*)out)[i] = (i32)(i0 < thr ? i0 / (q + 1) + 1 : (q == 0 ? i0 + 1 : (i0 - thr) / q + r + 1));
        break;
      }
    }
  }
}

// idx[i] = i + shift if that row is in row i's partition (ok[i] = 1), else ok[i] = 0 — lag: shift = −k, lead: shift = +k
__global__ __launch_bounds__(256) void window_offset_kernel(i64 shift, const i32* __restrict__ sp, const u32* __restrict__ first_part, i64 n,
                                                            u32* __restrict__ idx, u8* __restrict__ ok) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const i32 p = sp[i + 1] - 1;
    const i64 ps = first_part[p], pe = first_part[p + 1], j = i + shift;
    const bool in = j >= ps && j < pe;
    idx[i] = in ? (u32)j : 0u;
    ok[i] = in ? 1 : 0;
  }
}

// out_valid_byte[i] = ok[i] && source row idx[i] is valid
__global__ __launch_bounds__(256) void window_offset_valid_kernel(const u32* __restrict__ idx, const u8* __restrict__ ok, const u8* __restrict__ src_valid_bits, i64 n,
                                                                  u8* __restrict__ out_ok) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const u32 j = idx[i];
    out_ok[i] = ok[i] && (!src_valid_bits || ((src_valid_bits[j >> 3] >> (j & 7)) & 1)) ? 1 : 0;
  }
}

// ---- aggregates over frames that start at the partition start (the only frames Comet hands to its own Spark-exact accumulators,
// planner.rs:2953-2972): SUM / COUNT / AVG of exact types come from ONE inclusive prefix sum of the argument widened to 128 bits
// (wrap-around cancels in the difference S[end−1] − S[start−1]) and one prefix count of its non-NULL rows:
//   whole partition            [ps, pe)          ROWS  … CURRENT ROW   [ps, i + 1)          RANGE … CURRENT ROW   [ps, end of i's peer group)
__global__ __launch_bounds__(256) void window_widen_kernel(int width, const void* __restrict__ src, const u8* __restrict__ valid_bits, i64 n,
                                                           i128* __restrict__ out, i128* __restrict__ hi, u32* __restrict__ ok) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const bool v = !valid_bits || ((valid_bits[i >> 3] >> (i & 7)) & 1);
    i128 x = 0;
    if (v && src) {
      switch (width) {
        case 1: x = ((const i8*)src)[i]; break;
        case 2: x = ((const i16*)src)[i]; break;
        case 4: x = ((const i32*)src)[i]; break;
        case 8: x = ((const i64*)src)[i]; break;
        default: x = ((const i128*)src)[i]; break;
      }
    }
    if (hi) {   // wide decimals: the high and the low 64 bits are summed separately, so that no partition sum can wrap unnoticed
      out[i] = (i128)(u64)x;
      hi[i] = (i128)(i64)(x >> 64);
    } else {
      out[i] = x;
    }
    ok[i] = v ? 1u : 0u;
  }
}

constexpr int kScanTile = 2048;   // rows per block in the 128-bit scan (256 threads × 8)

__global__ __launch_bounds__(256) void scan128_tile_sum_kernel(const i128* __restrict__ in, i64 n, u128* __restrict__ tiles) {
  __shared__ u128 part[256];
  const i64 base = (i64)blockIdx.x * kScanTile;
  u128 s = 0;
  for (int k = 0; k < kScanTile / 256; k++) {
    const i64 i = base + (i64)threadIdx.x * (kScanTile / 256) + k;
    if (i < n) s += (u128)in[i];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) tiles[blockIdx.x] = part[0];
}
__global__ __launch_bounds__(256) void scan128_tiles_kernel(u128* tiles, i64 ntiles) {   // one block: exclusive scan of the tile sums
  __shared__ u128 part[256];
  const i64 per = (ntiles + 255) / 256, lo = (i64)threadIdx.x * per, hi = lo + per < ntiles ? lo + per : ntiles;
  u128 s = 0;
  for (i64 t = lo; t < hi; t++) s += tiles[t];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int st = 1; st < 256; st <<= 1) {
    u128 add = (int)threadIdx.x >= st ? part[threadIdx.x - st] : (u128)0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  u128 run = threadIdx.x ? part[threadIdx.x - 1] : (u128)0;
  for (i64 t = lo; t < hi; t++) {
    const u128 v = tiles[t];
    tiles[t] = run;
    run += v;
  }
}
__global__ __launch_bounds__(256) void scan128_apply_kernel(const i128* __restrict__ in, i64 n, const u128* __restrict__ tiles, i128* __restrict__ out) {
  __shared__ u128 part[256];
  const int per = kScanTile / 256;
  const i64 base = (i64)blockIdx.x * kScanTile + (i64)threadIdx.x * per;
  u128 loc[kScanTile / 256];
  u128 s = 0;
  for (int k = 0; k < per; k++) {
    const i64 i = base + k;
    s += i < n ? (u128)in[i] : (u128)0;
    loc[k] = s;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  // exclusive scan of the 256 per-thread sums (Hillis–Steele on LDS)
  for (int st = 1; st < 256; st <<= 1) {
    u128 add = (int)threadIdx.x >= st ? part[threadIdx.x - st] : (u128)0;
    __syncthreads();
    part[threadIdx.x] += add;
    __syncthreads();
  }
  const u128 before = tiles[blockIdx.x] + (threadIdx.x ? part[threadIdx.x - 1] : (u128)0);
  for (int k = 0; k < per; k++) {
    const i64 i = base + k;
    if (i < n) out[i] = (i128)(before + loc[k]);
  }
}

enum { WA_SUM_DEC = 0, WA_SUM_INT = 1, WA_COUNT = 2, WA_AVG_DEC = 3 };
// A frame bound: the partition's edge, the current row (ROWS) or its peer group (RANGE), or the current row ± a literal number of rows
// (negative = PRECEDING, positive = FOLLOWING — the sign convention of the plan, planner.rs:3016-3030).
// WB_RANGE_OFFSET: a RANGE frame with a value offset — the bound was searched per row over the ORDER BY key (window_range_bounds_kernel);
// the frame then carries the device array of row positions in place of the offset.
enum { WB_UNBOUNDED = 0, WB_CURRENT_ROW = 1, WB_CURRENT_RANGE = 2, WB_ROWS_OFFSET = 3, WB_RANGE_OFFSET = 4 };
struct WFrame { int lo_kind, hi_kind; i64 lo_off, hi_off; };
// rows [start, end) of row i's frame, clipped to its partition [ps, pe); empty frames come back with end == start
__device__ __forceinline__ void frame_bounds(const WFrame& f, i64 i, i64 ps, i64 pe, i64 gs, i64 ge, i64& start, i64& end) {
  switch (f.lo_kind) {
    case WB_UNBOUNDED: start = ps; break;
    case WB_CURRENT_ROW: start = i; break;
    case WB_CURRENT_RANGE: start = gs; break;
    case WB_RANGE_OFFSET: start = (i64)((const i32*)f.lo_off)[i]; break;
    default: start = i + f.lo_off; break;
  }
  switch (f.hi_kind) {
    case WB_UNBOUNDED: end = pe; break;
    case WB_CURRENT_ROW: end = i + 1; break;
    case WB_CURRENT_RANGE: end = ge; break;
    case WB_RANGE_OFFSET: end = (i64)((const i32*)f.hi_off)[i]; break;
    default: end = i + f.hi_off + 1; break;
  }
  if (start < ps) start = ps;
  if (end > pe) end = pe;
  if (end < start) end = start;
}

// S: inclusive 128-bit prefix sums of the argument, C: exclusive prefix counts (n + 1) of its non-NULL rows
__global__ __launch_bounds__(256) void window_agg_kernel(int fn, WFrame frame, const i128* __restrict__ S, const i128* __restrict__ SH, const i32* __restrict__ C, const i32* __restrict__ sp,
                                                         const i32* __restrict__ sg, const u32* __restrict__ first_part, const u32* __restrict__ first_peer, i64 n,
                                                         u128 bound, i128 scaler, u128 avg_bound, void* __restrict__ out, u8* __restrict__ out_ok) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const i32 p = sp[i + 1] - 1, g = sg[i + 1] - 1;
    i64 start, end;
    frame_bounds(frame, i, (i64)first_part[p], (i64)first_part[p + 1], (i64)first_peer[g], (i64)first_peer[g + 1], start, end);
    const bool empty = end <= start;
    i128 total = empty ? (i128)0 : (i128)((u128)S[end - 1] - (start ? (u128)S[start - 1] : (u128)0));
    bool wrapped = false;
    if (SH && !empty) {
      // total = hi·2^64 + lo with both parts exact; anything that does not fit 127 bits is far beyond every decimal precision
      const i128 hs = SH[end - 1] - (start ? SH[start - 1] : (i128)0);
      if (hs >= ((i128)1 << 63) || hs < -((i128)1 << 63)) wrapped = true;
      else wrapped = __builtin_add_overflow((i128)((u128)hs << 64), total, &total);
    }
    const i64 cnt = (wrapped || empty) ? 0 : (i64)C[end] - (i64)C[start];   // a wrapped sum evaluates to NULL like an overflowed one
    switch (fn) {
      case WA_SUM_DEC: {   // SumDecimal evaluate (sum_decimal.rs:264-279): NULL if no value or out of precision
        const bool ok = cnt > 0 && dec_fits(total, bound);
        ((i128*)out)[i] = ok ? total : (i128)0;
        out_ok[i] = ok ? 1 : 0;
        break;
      }
      case WA_SUM_INT: ((i64*)out)[i] = cnt > 0 ? (i64)total : 0; out_ok[i] = cnt > 0 ? 1 : 0; break;   // SumInteger: wrapping, NULL if no value
      case WA_COUNT: ((i64*)out)[i] = cnt; out_ok[i] = 1; break;
      case WA_AVG_DEC: {   // AvgDecimal evaluate (avg_decimal.rs:597-636, 670-689)
        i128 v = 0;
        const bool ok = cnt > 0 && dec_fits(total, bound) && dec_avg(total, cnt, scaler, avg_bound, v);
        ((i128*)out)[i] = ok ? v : (i128)0;
        out_ok[i] = ok ? 1 : 0;
        break;
      }
    }
  }
}

// ---- RANGE frames with value offsets -----------------------------------------------------------------------------------------------
// RANGE BETWEEN a PRECEDING AND b FOLLOWING over ONE integer ORDER BY key (the shape the JVM side sends: magnitudes only, the lower bound
// always PRECEDING, the upper always FOLLOWING — CometWindowExec.scala:588-632 → planner.rs:3031-3037,3090-3096): the frame of row i holds
// the partition's rows whose key lies within [key_i − a, key_i + b] in SORT order (descending keys: [key_i + a, key_i − b]).  As in
// DataFusion's WindowFrameStateRange the target is computed in the key's own width with wrapping arithmetic, a NULL key's target is NULL
// (its frame is its NULL peers), and each bound is the first row that does not sort before (lower) / sorts after (upper) the target.  Rows
// are sorted, so both are binary searches over the partition.  The reference advances its bounds monotonically along a partition, which
// differs from this stateless search only when key ± offset wraps.
__device__ __forceinline__ i64 wr_key(const void* keys, int width, i64 k) {
  switch (width) {
    case 1: return (i64)((const signed char*)keys)[k];
    case 2: return (i64)((const short*)keys)[k];
    case 4: return (i64)((const i32*)keys)[k];
    default: return ((const i64*)keys)[k];
  }
}
__device__ __forceinline__ i64 wr_wrap(i64 x, int width) {
  switch (width) {
    case 1: return (i64)(signed char)x;
    case 2: return (i64)(short)x;
    case 4: return (i64)(i32)x;
    default: return x;
  }
}
// does row k sort strictly before (dir = 0) / strictly after (dir = 1) the target?
__device__ __forceinline__ bool wr_cmp(const void* keys, const u8* valid, int width, i64 k, bool t_null, i64 t, int desc, int nulls_first, int dir) {
  const bool k_null = valid && !((valid[k >> 3] >> (k & 7)) & 1);
  if (k_null || t_null) {
    if (k_null && t_null) return false;
    const bool k_first = k_null ? nulls_first != 0 : nulls_first == 0;   // the row sorts before the target
    return dir == 0 ? k_first : !k_first;
  }
  const i64 v = wr_key(keys, width, k);
  const bool before = desc ? v > t : v < t, after = desc ? v < t : v > t;
  return dir == 0 ? before : after;
}
__global__ __launch_bounds__(256) void window_range_bounds_kernel(int width, const void* __restrict__ keys, const u8* __restrict__ valid, const i32* __restrict__ sp,
                                                                  const u32* __restrict__ first_part, i64 n, int desc, int nulls_first, int has_lo, i64 dlo, int has_hi, i64 dhi,
                                                                  i32* __restrict__ out_lo, i32* __restrict__ out_hi) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const i32 p = sp[i + 1] - 1;
    const i64 ps = (i64)first_part[p], pe = (i64)first_part[p + 1];
    const bool i_null = valid && !((valid[i >> 3] >> (i & 7)) & 1);
    const i64 v = i_null ? 0 : wr_key(keys, width, i);
    if (has_lo) {
      const i64 t = wr_wrap((i64)(desc ? (u64)v + (u64)dlo : (u64)v - (u64)dlo), width);
      i64 a = ps, b = pe;                       // first row in [ps, pe) that does not sort before the target
      while (a < b) {
        const i64 m = (a + b) >> 1;
        if (wr_cmp(keys, valid, width, m, i_null, t, desc, nulls_first, 0)) a = m + 1; else b = m;
      }
      out_lo[i] = (i32)a;
    }
    if (has_hi) {
      const i64 t = wr_wrap((i64)(desc ? (u64)v - (u64)dhi : (u64)v + (u64)dhi), width);
      i64 a = ps, b = pe;                       // first row that sorts after the target
      while (a < b) {
        const i64 m = (a + b) >> 1;
        if (!wr_cmp(keys, valid, width, m, i_null, t, desc, nulls_first, 1)) a = m + 1; else b = m;
      }
      out_hi[i] = (i32)a;
    }
  }
}

// ---- FIRST_VALUE / LAST_VALUE / nth_value over a frame ---------------------------------------------------------------------------------
// The row whose value the function returns: the frame's first / last / n-th row, or — IGNORE NULLS, C = exclusive prefix counts of the
// column's non-NULL rows — its first / last / n-th non-NULL row: the smallest k in the frame with C[k + 1] ≥ C[start] + want (C is
// monotone: a binary search).  No such row → ok = 0 (the result is NULL); the gather kernels of lag / lead take it from there.
// WP_LAG / WP_LEAD (IGNORE NULLS, the frame is the whole partition): the nth non-NULL row strictly before / after the current one
enum { WP_FIRST = 0, WP_LAST = 1, WP_NTH = 2, WP_LAG = 3, WP_LEAD = 4 };
__global__ __launch_bounds__(256) void window_valid_flags_kernel(const u8* __restrict__ valid, i64 n, u32* __restrict__ flags) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) flags[i] = (valid[i >> 3] >> (i & 7)) & 1u;
}
__global__ __launch_bounds__(256) void window_pick_kernel(int mode, i64 nth, WFrame frame, const i32* __restrict__ C, const i32* __restrict__ sp, const i32* __restrict__ sg,
                                                          const u32* __restrict__ first_part, const u32* __restrict__ first_peer, i64 n, u32* __restrict__ idx,
                                                          u8* __restrict__ ok) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const i32 p = sp[i + 1] - 1, g = sg[i + 1] - 1;
    i64 start, end;
    frame_bounds(frame, i, (i64)first_part[p], (i64)first_part[p + 1], (i64)first_peer[g], (i64)first_peer[g + 1], start, end);
    i64 pick = -1;
    if (end > start) {
      if (!C) {
        pick = mode == WP_FIRST ? start : mode == WP_LAST ? end - 1 : mode == WP_LAG ? (i - nth >= start ? i - nth : -1) : mode == WP_LEAD ? (i + nth < end ? i + nth : -1)
                                                                   : (start + nth - 1 < end ? start + nth - 1 : -1);
      } else {
        const i64 base = (i64)C[start], total = (i64)C[end] - base;
        // the ordinal, among the frame's non-NULL rows, of the one wanted
        const i64 want = mode == WP_FIRST ? 1 : mode == WP_LAST ? total : mode == WP_LAG ? (i64)C[i] - base - nth + 1 : mode == WP_LEAD ? (i64)C[i + 1] - base + nth : nth;
        if (want >= 1 && want <= total) {
          i64 a = start, b = end - 1;
          while (a < b) {
            const i64 m = (a + b) >> 1;
            if ((i64)C[m + 1] >= base + want) b = m; else a = m + 1;
          }
          pick = a;
        }
      }
    }
    idx[i] = pick >= 0 ? (u32)pick : 0u;
    ok[i] = pick >= 0 ? 1 : 0;
  }
}

// ---- MIN / MAX over frames -------------------------------------------------------------------------------------------------------
// Running extremes per partition: P[i] = extreme of the non-NULL values of rows [partition start, i], Q[i] = of rows [i, partition end).
// A segmented inclusive scan in three launches (tile scan in LDS → carries across tiles → apply), run forwards for P and backwards for Q.
// An element is (partition ordinal, value, "holds a value"); combining left ⊕ right keeps right when the partitions differ.
struct MM { i32 pid; u32 has; i128 v; };
__device__ __forceinline__ MM mm_combine(const MM& a, const MM& b, int is_max) {
  if (a.pid != b.pid || !a.has) return b;
  if (!b.has) { MM r = a; r.pid = b.pid; return r; }
  MM r = b;
  r.v = is_max ? (a.v > b.v ? a.v : b.v) : (a.v < b.v ? a.v : b.v);
  return r;
}
constexpr int kMMTile = 1024;
// element at scan position j (forwards: row j; backwards: row n − 1 − j)
__device__ __forceinline__ MM mm_load(const i128* vals, const u32* ok, const i32* sp, i64 n, i64 j, int backward) {
  MM e;
  if (j >= n) { e.pid = -1; e.has = 0; e.v = 0; return e; }
  const i64 i = backward ? n - 1 - j : j;
  e.pid = sp[i + 1] - 1;
  e.has = ok[i] ? 1u : 0u;
  e.v = vals[i];
  return e;
}
__global__ __launch_bounds__(256) void mm_tile_kernel(const i128* __restrict__ vals, const u32* __restrict__ ok, const i32* __restrict__ sp, i64 n, int backward, int is_max,
                                                      MM* __restrict__ local, MM* __restrict__ tile_last) {
  __shared__ MM part[256];
  const i64 base = (i64)blockIdx.x * kMMTile + (i64)threadIdx.x * 4;
  MM loc[4];
  MM run = mm_load(vals, ok, sp, n, base, backward);
  loc[0] = run;
  for (int k = 1; k < 4; k++) {
    run = mm_combine(run, mm_load(vals, ok, sp, n, base + k, backward), is_max);
    loc[k] = run;
  }
  part[threadIdx.x] = run;
  __syncthreads();
  for (int st = 1; st < 256; st <<= 1) {
    MM left = part[threadIdx.x];
    if ((int)threadIdx.x >= st) left = mm_combine(part[threadIdx.x - st], part[threadIdx.x], is_max);
    __syncthreads();
    part[threadIdx.x] = left;
    __syncthreads();
  }
  for (int k = 0; k < 4; k++) {
    const i64 j = base + k;
    if (j >= n) break;
    local[j] = threadIdx.x ? mm_combine(part[threadIdx.x - 1], loc[k], is_max) : loc[k];
  }
  if (threadIdx.x == 255) tile_last[blockIdx.x] = part[255];
}
// carries: tile_carry[t] = the running element at the end of tile t − 1 (pid −1 for tile 0); one thread — n / 1024 steps
__global__ void mm_carry_kernel(const MM* __restrict__ tile_last, i64 ntiles, int is_max, MM* __restrict__ tile_carry) {
  MM run;
  run.pid = -1; run.has = 0; run.v = 0;
  for (i64 t = 0; t < ntiles; t++) {
    tile_carry[t] = run;
    const MM last = tile_last[t];
    run = last.pid < 0 ? run : mm_combine(run, last, is_max);
  }
}
__global__ __launch_bounds__(256) void mm_apply_kernel(const MM* __restrict__ local, const MM* __restrict__ tile_carry, i64 n, int backward, int is_max, i128* __restrict__ out_v,
                                                       u8* __restrict__ out_has) {
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < n; j += (i64)gridDim.x * 256) {
    const MM r = mm_combine(tile_carry[j / kMMTile], local[j], is_max);
    const i64 i = backward ? n - 1 - j : j;
    out_v[i] = r.v;
    out_has[i] = (u8)r.has;
  }
}
// the frame's extreme from the running extremes; a frame bounded on both sides is walked (its width is capped by the planner)
__global__ __launch_bounds__(256) void window_minmax_kernel(int is_max, WFrame frame, const i128* __restrict__ vals, const u32* __restrict__ ok, const i128* __restrict__ P,
                                                            const u8* __restrict__ Ph, const i128* __restrict__ Q, const u8* __restrict__ Qh, const i32* __restrict__ sp,
                                                            const i32* __restrict__ sg, const u32* __restrict__ first_part, const u32* __restrict__ first_peer, i64 n, int out_width,
                                                            void* __restrict__ out, u8* __restrict__ out_ok) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const i32 p = sp[i + 1] - 1, g = sg[i + 1] - 1;
    i64 start, end;
    frame_bounds(frame, i, (i64)first_part[p], (i64)first_part[p + 1], (i64)first_peer[g], (i64)first_peer[g + 1], start, end);
    i128 v = 0;
    bool has = false;
    if (end > start) {
      if (frame.lo_kind == WB_UNBOUNDED) { v = P[end - 1]; has = Ph[end - 1] != 0; }
      else if (frame.hi_kind == WB_UNBOUNDED) { v = Q[start]; has = Qh[start] != 0; }
      else
        for (i64 k = start; k < end; k++)
          if (ok[k]) {
            const i128 x = vals[k];
            if (!has || (is_max ? x > v : x < v)) v = x;
            has = true;
          }
    }
    if (!has) v = 0;
    switch (out_width) {
      case 1: ((u8*)out)[i] = (u8)v; break;
      case 2: ((unsigned short*)out)[i] = (unsigned short)v; break;
      case 4: ((u32*)out)[i] = (u32)v; break;
      case 8: ((u64*)out)[i] = (u64)v; break;
      default: ((i128*)out)[i] = v; break;
    }
    out_ok[i] = has ? 1 : 0;
  }
}

// lag / lead with a non-NULL default: rows whose offset row lies outside the partition take the default value (a NULL source value stays NULL)
template <class T>
__global__ __launch_bounds__(256) void window_default_kernel(const u8* __restrict__ inside, i64 n, T value, T* __restrict__ data, u8* __restrict__ ok_bytes) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256)
    if (!inside[i]) { data[i] = value; ok_bytes[i] = 1; }
}

}  // namespace

extern "C" {

int comet_launch_window_default(int width, const uint8_t* inside, int64_t n, const void* value, void* data, uint8_t* ok_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  switch (width) {
    case 1: hipLaunchKernelGGL(window_default_kernel<u8>, grid_for(n), 256, 0, st, inside, (i64)n, *(const u8*)value, (u8*)data, ok_bytes); break;
    case 2: hipLaunchKernelGGL(window_default_kernel<unsigned short>, grid_for(n), 256, 0, st, inside, (i64)n, *(const unsigned short*)value, (unsigned short*)data, ok_bytes); break;
    case 4: hipLaunchKernelGGL(window_default_kernel<u32>, grid_for(n), 256, 0, st, inside, (i64)n, *(const u32*)value, (u32*)data, ok_bytes); break;
    case 8: hipLaunchKernelGGL(window_default_kernel<u64>, grid_for(n), 256, 0, st, inside, (i64)n, *(const u64*)value, (u64*)data, ok_bytes); break;
    case 16: { i128 v; memcpy(&v, value, 16); hipLaunchKernelGGL(window_default_kernel<i128>, grid_for(n), 256, 0, st, inside, (i64)n, v, (i128*)data, ok_bytes); break; }
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int comet_launch_window_widen(int width, const void* src, const uint8_t* valid_bits, int64_t n, void* out128, void* hi128, uint32_t* ok, void* stream) {
  if (n > 0) hipLaunchKernelGGL(window_widen_kernel, grid_for(n), 256, 0, (hipStream_t)stream, width, src, valid_bits, (i64)n, (i128*)out128, (i128*)hi128, ok);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// inclusive prefix sums of n 128-bit integers (wrapping); tiles: (n / 2048 + 2) × 16 bytes of scratch
int comet_launch_scan128(const void* in128, int64_t n, void* tiles, void* out128, void* stream) {
  if (n <= 0) return 0;
  const int64_t nt = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(scan128_tile_sum_kernel, (int)nt, 256, 0, (hipStream_t)stream, (const i128*)in128, (i64)n, (u128*)tiles);
  hipLaunchKernelGGL(scan128_tiles_kernel, 1, 256, 0, (hipStream_t)stream, (u128*)tiles, (i64)nt);
  hipLaunchKernelGGL(scan128_apply_kernel, (int)nt, 256, 0, (hipStream_t)stream, (const i128*)in128, (i64)n, (const u128*)tiles, (i128*)out128);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// one direction of the running extremes: scratch `local` n × 32 bytes, `tiles` 2 × (n / 1024 + 1) × 32 bytes
int comet_launch_window_running_extreme(const void* vals128, const uint32_t* ok, const int32_t* sp, int64_t n, int backward, int is_max, void* local, void* tiles, void* out_v,
                                        uint8_t* out_has, void* stream) {
  if (n <= 0) return 0;
  const int64_t nt = (n + kMMTile - 1) / kMMTile;
  MM* tl = (MM*)tiles;
  hipLaunchKernelGGL(mm_tile_kernel, (int)nt, 256, 0, (hipStream_t)stream, (const i128*)vals128, ok, sp, (i64)n, backward, is_max, (MM*)local, tl);
  hipLaunchKernelGGL(mm_carry_kernel, 1, 1, 0, (hipStream_t)stream, (const MM*)tl, (i64)nt, is_max, tl + nt);
  hipLaunchKernelGGL(mm_apply_kernel, grid_for(n), 256, 0, (hipStream_t)stream, (const MM*)local, (const MM*)(tl + nt), (i64)n, backward, is_max, (i128*)out_v, out_has);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_window_minmax(int is_max, int lo_kind, int64_t lo_off, int hi_kind, int64_t hi_off, const void* vals128, const uint32_t* ok, const void* P, const uint8_t* Ph,
                               const void* Q, const uint8_t* Qh, const int32_t* sp, const int32_t* sg, const uint32_t* first_part, const uint32_t* first_peer, int64_t n,
                               int out_width, void* out, uint8_t* out_ok, void* stream) {
  WFrame f{lo_kind, hi_kind, (i64)lo_off, (i64)hi_off};
  if (n > 0)
    hipLaunchKernelGGL(window_minmax_kernel, grid_for(n), 256, 0, (hipStream_t)stream, is_max, f, (const i128*)vals128, ok, (const i128*)P, Ph, (const i128*)Q, Qh, sp, sg, first_part,
                       first_peer, (i64)n, out_width, out, out_ok);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_window_range_bounds(int width, const void* keys, const uint8_t* valid, const int32_t* sp, const uint32_t* first_part, int64_t n, int desc, int nulls_first,
                                     int has_lo, int64_t dlo, int has_hi, int64_t dhi, int32_t* out_lo, int32_t* out_hi, void* stream) {
  if (n > 0)
    hipLaunchKernelGGL(window_range_bounds_kernel, grid_for(n), 256, 0, (hipStream_t)stream, width, keys, valid, sp, first_part, (i64)n, desc, nulls_first, has_lo, (i64)dlo, has_hi,
                       (i64)dhi, out_lo, out_hi);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_window_valid_flags(const uint8_t* valid, int64_t n, uint32_t* flags, void* stream) {
  if (n > 0) hipLaunchKernelGGL(window_valid_flags_kernel, grid_for(n), 256, 0, (hipStream_t)stream, valid, (i64)n, flags);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_window_pick(int mode, int64_t nth, int lo_kind, int64_t lo_off, int hi_kind, int64_t hi_off, const int32_t* C, const int32_t* sp, const int32_t* sg,
                             const uint32_t* first_part, const uint32_t* first_peer, int64_t n, uint32_t* idx, uint8_t* ok, void* stream) {
  if (n > 0)
    hipLaunchKernelGGL(window_pick_kernel, grid_for(n), 256, 0, (hipStream_t)stream, mode, (i64)nth, WFrame{lo_kind, hi_kind, (i64)lo_off, (i64)hi_off}, C, sp, sg, first_part, first_peer,
                       (i64)n, idx, ok);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_window_agg(int fn, int lo_kind, int64_t lo_off, int hi_kind, int64_t hi_off, const void* S128, const void* SH128, const int32_t* C, const int32_t* sp, const int32_t* sg, const uint32_t* first_part,
                            const uint32_t* first_peer, int64_t n, const void* bound16, const void* scaler16, const void* avg_bound16, void* out, uint8_t* out_ok,
                            void* stream) {
  u128 bound, avg_bound;
  i128 scaler;
  memcpy(&bound, bound16, 16);
  memcpy(&scaler, scaler16, 16);
  memcpy(&avg_bound, avg_bound16, 16);
  if (n > 0)
    hipLaunchKernelGGL(window_agg_kernel, grid_for(n), 256, 0, (hipStream_t)stream, fn, WFrame{lo_kind, hi_kind, (i64)lo_off, (i64)hi_off}, (const i128*)S128, (const i128*)SH128, C, sp, sg, first_part, first_peer, (i64)n, bound, scaler,
                       avg_bound, out, out_ok);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}


int comet_launch_window_flags(const uint8_t* part_planes, int Wp, const uint8_t* order_planes, int Wo, int64_t n, uint32_t* fpart, uint32_t* fpeer, void* stream) {
  if (n > 0) hipLaunchKernelGGL(window_flags_kernel, grid_for(n), 256, 0, (hipStream_t)stream, part_planes, Wp, order_planes, Wo, (i64)n, fpart, fpeer);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_window_first(const uint32_t* fpart, const int32_t* sp, const uint32_t* fpeer, const int32_t* sg, int64_t n, uint32_t* first_part,
                              uint32_t* first_peer, void* stream) {
  hipLaunchKernelGGL(window_first_kernel, grid_for(n + 1), 256, 0, (hipStream_t)stream, fpart, sp, fpeer, sg, (i64)n, first_part, first_peer);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_window_rank(int kind, int64_t arg, const int32_t* sp, const int32_t* sg, const uint32_t* first_part, const uint32_t* first_peer, int64_t n,
                             void* out, void* stream) {
  if (n > 0) hipLaunchKernelGGL(window_rank_kernel, grid_for(n), 256, 0, (hipStream_t)stream, kind, (i64)arg, sp, sg, first_part, first_peer, (i64)n, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_window_offset(int64_t shift, const int32_t* sp, const uint32_t* first_part, int64_t n, uint32_t* idx, uint8_t* ok, void* stream) {
  if (n > 0) hipLaunchKernelGGL(window_offset_kernel, grid_for(n), 256, 0, (hipStream_t)stream, (i64)shift, sp, first_part, (i64)n, idx, ok);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_window_offset_valid(const uint32_t* idx, const uint8_t* ok, const uint8_t* src_valid_bits, int64_t n, uint8_t* out_ok, void* stream) {
  if (n > 0) hipLaunchKernelGGL(window_offset_valid_kernel, grid_for(n), 256, 0, (hipStream_t)stream, idx, ok, src_valid_bits, (i64)n, out_ok);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // extern "C"
