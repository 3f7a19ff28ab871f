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

}  // namespace

extern "C" {

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
