// Ahead-of-time kernels (hipcc --offload-arch=gfx950): the pieces of the hot path that do not depend on
// the plan — Spark murmur3 row hashing + pmod partitioning for the exchange step — and a
// hand-written instantiation of the fused aggregate template for TPC-H Q6, which keeps the templates of
// comet_device.hpp under the ahead-of-time compiler as well as hiprtc.
#include <hip/hip_runtime.h>

#include "device/comet_device.hpp"

using namespace comet;

// ---------------------------------------------------------------------------------------------
// murmur3: one lane per row, hashes[] chained across key columns (create_hashes_internal!,
// native/spark-expr/src/hash_funcs/utils.rs:573-760): NULL rows leave the running hash untouched.
// ---------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256) void mm3_kernel(const void* values, const u8* validity, i64 n, u32* hashes, F f) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    if (validity && !((validity[i >> 3] >> (i & 7)) & 1)) continue;
    hashes[i] = f(values, i, hashes[i]);
  }
}

struct HashBool { __device__ u32 operator()(const void* v, i64 i, u32 s) const { return mm3_hash_i32((((const u8*)v)[i >> 3] >> (i & 7)) & 1, s); } };
struct HashI8 { __device__ u32 operator()(const void* v, i64 i, u32 s) const { return mm3_hash_i32((i32)((const i8*)v)[i], s); } };
struct HashI16 { __device__ u32 operator()(const void* v, i64 i, u32 s) const { return mm3_hash_i32((i32)((const i16*)v)[i], s); } };
struct HashI32 { __device__ u32 operator()(const void* v, i64 i, u32 s) const { return mm3_hash_i32(((const i32*)v)[i], s); } };
struct HashI64 { __device__ u32 operator()(const void* v, i64 i, u32 s) const { return mm3_hash_i64(((const i64*)v)[i], s); } };
struct HashF32 { __device__ u32 operator()(const void* v, i64 i, u32 s) const { return mm3_hash_f32(((const float*)v)[i], s); } };
struct HashF64 { __device__ u32 operator()(const void* v, i64 i, u32 s) const { return mm3_hash_f64(((const double*)v)[i], s); } };
// decimal(p ≤ 18) hashes as i64, wider as the 16 little-endian bytes (hash_funcs/utils.rs:154-158)
struct HashDecSmall { __device__ u32 operator()(const void* v, i64 i, u32 s) const { return mm3_hash_i64(((const i64*)v)[2 * i], s); } };
struct HashDecWide { __device__ u32 operator()(const void* v, i64 i, u32 s) const { return mm3_hash_i128(((const i128*)v)[i], s); } };
struct HashUtf8 {
  const u8* bytes;
  __device__ u32 operator()(const void* v, i64 i, u32 s) const {
    const i32* off = (const i32*)v;
    return mm3_hash_bytes(bytes + off[i], off[i + 1] - off[i], s);
  }
};

__global__ __launch_bounds__(256) void pmod_kernel(const u32* hashes, i64 n, i32 np, i32* out) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) out[i] = pmod(hashes[i], np);
}

// Are all n Utf8 values exactly L bytes long?  (lets the fused kernels address the bytes directly, see ld_str_fixed)
// Streaming read of the offsets: four independent 16-byte loads per lane and iteration; offset i must equal off[0] + i·L, so no value
// depends on its neighbour and nothing is exchanged between lanes.
__global__ __launch_bounds__(256) void utf8_uniform_kernel(const i32* off, i64 n, i32 L, u32* flag) {
  bool bad = false;
  const i32 o0 = off[0];
  const i64 total = n + 1;                                   // offsets to check
  i64 head = 0;                                              // elements before the first 16-byte boundary
  while (head < total && ((((uintptr_t)(off + head)) & 15) != 0)) head++;
  const int4* v4 = (const int4*)(off + head);
  const i64 ngroups = (total - head) / 4;
  constexpr int U = 4;
  for (i64 g0 = ((i64)blockIdx.x * 256 + threadIdx.x) * U; g0 < ngroups; g0 += (i64)gridDim.x * 256 * U) {
    int4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = g0 + u < ngroups ? v4[g0 + u] : make_int4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (g0 + u < ngroups) {
        const i64 i = head + 4 * (g0 + u);
        const i32 e = o0 + (i32)(i * L);                      // wraps like the offsets would; compared for equality only
        bad |= (v[u].x != e) | (v[u].y != e + L) | (v[u].z != e + 2 * L) | (v[u].w != e + 3 * L);
      }
    }
  }
  if (blockIdx.x == 0) {
    for (i64 i = threadIdx.x; i < head; i += 256) bad |= off[i] != o0 + (i32)(i * L);
    for (i64 i = head + ngroups * 4 + threadIdx.x; i < total; i += 256) bad |= off[i] != o0 + (i32)(i * L);
  }
  if (__any(bad) && lane_id() == 0) atomicOr(flag, 1u);
}

// ---------------------------------------------------------------------------------------------
// Dictionary unpack (ScanExec always hands plain arrays downstream: operators/scan.rs:98-106, copy.rs:69-93):
// out[i] = dict[idx[i]], validity = index validity AND dictionary-value validity.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ i64 dict_index(const void* idx, int iw, i64 i) {
  switch (iw) {
    case 1: return ((const i8*)idx)[i];
    case 2: return ((const i16*)idx)[i];
    case 4: return ((const i32*)idx)[i];
    default: return ((const i64*)idx)[i];
  }
}
__global__ __launch_bounds__(256) void dict_gather_fixed_kernel(const void* idx, int iw, const u8* idx_valid, const u8* dict, const u8* dict_valid,
                                                                int width, i64 n, u8* out, u8* out_valid_bytes) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    bool ok = !idx_valid || ((idx_valid[i >> 3] >> (i & 7)) & 1);
    i64 k = ok ? dict_index(idx, iw, i) : 0;
    if (ok && dict_valid) ok = (dict_valid[k >> 3] >> (k & 7)) & 1;
    for (int b = 0; b < width; b++) out[i * width + b] = ok ? dict[k * width + b] : 0;
    out_valid_bytes[i] = ok ? 1 : 0;
  }
}
__global__ __launch_bounds__(256) void dict_gather_str_len_kernel(const void* idx, int iw, const u8* idx_valid, const i32* dict_offs, const u8* dict_valid,
                                                                  i64 n, u32* lengths, u8* out_valid_bytes) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    bool ok = !idx_valid || ((idx_valid[i >> 3] >> (i & 7)) & 1);
    i64 k = ok ? dict_index(idx, iw, i) : 0;
    if (ok && dict_valid) ok = (dict_valid[k >> 3] >> (k & 7)) & 1;
    lengths[i] = ok ? (u32)(dict_offs[k + 1] - dict_offs[k]) : 0;
    out_valid_bytes[i] = ok ? 1 : 0;
  }
}
__global__ __launch_bounds__(256) void dict_gather_str_copy_kernel(const void* idx, int iw, const u8* valid_bytes, const i32* dict_offs, const u8* dict_bytes,
                                                                   i64 n, const i32* out_offs, u8* out_bytes) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    if (!valid_bytes[i]) continue;
    i64 k = dict_index(idx, iw, i);
    const u8* src = dict_bytes + dict_offs[k];
    i32 len = dict_offs[k + 1] - dict_offs[k];
    u8* dst = out_bytes + out_offs[i];
    for (i32 b = 0; b < len; b++) dst[b] = src[b];
  }
}

static int grid_for(i64 n) {
  i64 g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 256 * 8 ? 256 * 8 : g));
}

extern "C" int comet_launch_murmur3(int type_id, int precision, const void* values, const uint8_t* validity, const void* aux,
                                    int64_t n, uint32_t* hashes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  int g = grid_for(n);
  switch (type_id) {
    case 0: hipLaunchKernelGGL(mm3_kernel<HashBool>, g, 256, 0, st, values, validity, (i64)n, hashes, HashBool()); break;
    case 1: hipLaunchKernelGGL(mm3_kernel<HashI8>, g, 256, 0, st, values, validity, (i64)n, hashes, HashI8()); break;
    case 2: hipLaunchKernelGGL(mm3_kernel<HashI16>, g, 256, 0, st, values, validity, (i64)n, hashes, HashI16()); break;
    case 3: case 12: hipLaunchKernelGGL(mm3_kernel<HashI32>, g, 256, 0, st, values, validity, (i64)n, hashes, HashI32()); break;
    case 4: case 9: case 11: hipLaunchKernelGGL(mm3_kernel<HashI64>, g, 256, 0, st, values, validity, (i64)n, hashes, HashI64()); break;
    case 5: hipLaunchKernelGGL(mm3_kernel<HashF32>, g, 256, 0, st, values, validity, (i64)n, hashes, HashF32()); break;
    case 6: hipLaunchKernelGGL(mm3_kernel<HashF64>, g, 256, 0, st, values, validity, (i64)n, hashes, HashF64()); break;
    case 7: case 8: {
      HashUtf8 h;
      h.bytes = (const u8*)aux;
      hipLaunchKernelGGL(mm3_kernel<HashUtf8>, g, 256, 0, st, values, validity, (i64)n, hashes, h);
      break;
    }
    case 10:
      if (precision <= 18) hipLaunchKernelGGL(mm3_kernel<HashDecSmall>, g, 256, 0, st, values, validity, (i64)n, hashes, HashDecSmall());
      else hipLaunchKernelGGL(mm3_kernel<HashDecWide>, g, 256, 0, st, values, validity, (i64)n, hashes, HashDecWide());
      break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// exact Float64 sums: the fixed-point window of a sum moved up by `shift` bits — arithmetic right shift of the 192-bit accumulator at
// word `word_off` of every record (partials of an ungrouped aggregate, slots of a group table)
__global__ __launch_bounds__(256) void fix_rescale_kernel(u64* base, i64 count, i64 stride, i32 word_off, i32 shift) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < count; i += (i64)gridDim.x * 256) {
    u64* t = base + i * stride + word_off;
    const u64 w[3] = {t[0], t[1], t[2]};
    const u64 sign = (w[2] >> 63) ? ~0ull : 0ull;
    u64 r[3];
    const int ws = shift >> 6, bs = shift & 63;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const u64 lo = k + ws < 3 ? w[k + ws] : sign;
      const u64 hi = k + ws + 1 < 3 ? w[k + ws + 1] : sign;
      r[k] = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
    }
    t[0] = r[0]; t[1] = r[1]; t[2] = r[2];
  }
}
extern "C" int comet_launch_fix_rescale(uint64_t* base, int64_t count, int64_t stride_words, int32_t word_off, int32_t shift, void* stream) {
  if (count <= 0 || shift <= 0) return 0;
  hipLaunchKernelGGL(fix_rescale_kernel, grid_for(count), 256, 0, (hipStream_t)stream, (u64*)base, (i64)count, (i64)stride_words, word_off, shift > 191 ? 191 : shift);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// Calibration of rocprofv3's FETCH_SIZE on this GPU: a streaming read of `n_bytes` with 4, 8 or 16 bytes per lane and load (the widths
// the fused pipelines issue for Date32 / Int64-or-narrow-decimal / Decimal128 columns).  tools/pmc_calibrate.py compares the counter with
// the known byte count (MI355X_MICROARCH.md calibrates the ×2 correction for 16 B/lane only).
template <class T>
__global__ __launch_bounds__(256) void calib_read_kernel(const T* p, i64 n, u64* sink) {
  u64 acc = 0;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const T v = p[i];
    const u32* w = (const u32*)&v;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); k++) acc += w[k];
  }
  if (acc == 0x123456789abcdefull) *sink = acc;   // never true in practice: keeps the loads alive
}
struct CalibB16 { u32 w[4]; };
// the access pattern of ld_dec_lo: the low 8 bytes of every 16-byte Decimal128 value (every cache line is touched, half of its bytes used)
__global__ __launch_bounds__(256) void calib_read_lo8_kernel(const u64* p, i64 n, u64* sink) {
  u64 acc = 0;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) acc += p[2 * i];
  if (acc == 0x123456789abcdefull) *sink = acc;
}
extern "C" int comet_calib_read(const void* p, int64_t n_bytes, int32_t lane_bytes, uint64_t* sink, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int grid = 256 * 8;
  if (lane_bytes == 4) hipLaunchKernelGGL(calib_read_kernel<u32>, grid, 256, 0, st, (const u32*)p, (i64)(n_bytes / 4), (u64*)sink);
  else if (lane_bytes == 8) hipLaunchKernelGGL(calib_read_kernel<u64>, grid, 256, 0, st, (const u64*)p, (i64)(n_bytes / 8), (u64*)sink);
  else if (lane_bytes == 16) hipLaunchKernelGGL(calib_read_kernel<CalibB16>, grid, 256, 0, st, (const CalibB16*)p, (i64)(n_bytes / 16), (u64*)sink);
  else if (lane_bytes == 816) hipLaunchKernelGGL(calib_read_lo8_kernel, grid, 256, 0, st, (const u64*)p, (i64)(n_bytes / 16), (u64*)sink);
  else return -1;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int comet_launch_utf8_uniform(const int32_t* offsets, int64_t n, int32_t L, uint32_t* flag, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(utf8_uniform_kernel, grid_for(n), 256, 0, (hipStream_t)stream, offsets, (i64)n, L, flag);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int comet_launch_dict_gather_fixed(const void* idx, int iw, const uint8_t* idx_valid, const uint8_t* dict, const uint8_t* dict_valid,
                                              int width, int64_t n, uint8_t* out, uint8_t* out_valid_bytes, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(dict_gather_fixed_kernel, grid_for(n), 256, 0, (hipStream_t)stream, idx, iw, idx_valid, dict, dict_valid, width, (i64)n, out, out_valid_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int comet_launch_dict_gather_str_len(const void* idx, int iw, const uint8_t* idx_valid, const int32_t* dict_offs, const uint8_t* dict_valid,
                                                int64_t n, uint32_t* lengths, uint8_t* out_valid_bytes, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(dict_gather_str_len_kernel, grid_for(n), 256, 0, (hipStream_t)stream, idx, iw, idx_valid, dict_offs, dict_valid, (i64)n, lengths, out_valid_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int comet_launch_dict_gather_str_copy(const void* idx, int iw, const uint8_t* valid_bytes, const int32_t* dict_offs, const uint8_t* dict_bytes,
                                                 int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(dict_gather_str_copy_kernel, grid_for(n), 256, 0, (hipStream_t)stream, idx, iw, valid_bytes, dict_offs, dict_bytes, (i64)n, out_offs, out_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// A result's buffers → one block of pinned host memory, in ONE launch (exec.cpp table_to_host_batches): a query's last batch is a handful of
// small buffers — values and validity of each column — and a hipMemcpy per buffer costs 20–70 µs of latency each, fifteen of them 0.7 ms
// behind a 15 ms TPC-H Q3 (profiles/r4_q3_timeline.txt) and a tenth of a TPC-H Q1 task.  descs (pinned, read by the device): n × { src, dst, bytes };
// workgroup y copies piece x of buffer y, bytes one by one at the unaligned ends, words in between.
struct CometCopyDesc { const uint8_t* src; uint8_t* dst; uint64_t len; };
__global__ __launch_bounds__(256) void copy_small_kernel(const CometCopyDesc* __restrict__ descs) {
  const CometCopyDesc d = descs[blockIdx.y];
  const u64 lo = (u64)blockIdx.x << 14, hi = lo + 16384 < d.len ? lo + 16384 : d.len;
  if (lo >= d.len) return;
  if ((((u64)d.src | (u64)d.dst) & 3u) == 0) {
    const u64 words = (hi - lo) >> 2;
    const u32* s4 = (const u32*)(d.src + lo);
    u32* d4 = (u32*)(d.dst + lo);
    for (u64 i = threadIdx.x; i < words; i += 256) d4[i] = s4[i];
    for (u64 i = lo + (words << 2) + threadIdx.x; i < hi; i += 256) d.dst[i] = d.src[i];
  } else {
    for (u64 i = lo + threadIdx.x; i < hi; i += 256) d.dst[i] = d.src[i];
  }
}
extern "C" int comet_launch_copy_small(const void* descs, int n, uint64_t longest, void* stream) {
  if (n <= 0 || !longest) return 0;
  hipLaunchKernelGGL(copy_small_kernel, dim3((unsigned)((longest + 16383) >> 14), (unsigned)n), 256, 0, (hipStream_t)stream, (const CometCopyDesc*)descs);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int comet_launch_pmod(const uint32_t* hashes, int64_t n, int32_t np, int32_t* out, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(pmod_kernel, grid_for(n), 256, 0, (hipStream_t)stream, hashes, (i64)n, np, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------------------------
// Hand-written TPC-H Q6 functor (what codegen emits for the plan, written out by hand):
//   in[0] l_quantity dec(12,2)  in[1] l_extendedprice dec(12,2)  in[2] l_discount dec(12,2)  in[3] l_shipdate date32
//   WHERE shipdate >= d0 AND shipdate < d1 AND discount BETWEEN lo AND hi AND quantity < q
//   SUM(extendedprice * discount) : Decimal(35,4) state (sum, is_empty)
//   iarg[1]=d0 iarg[2]=d1 iarg[3]=disc_lo iarg[4]=disc_hi iarg[5]=qty_lt
// ---------------------------------------------------------------------------------------------
struct Q6Static {
  static constexpr int R = 4;
  static constexpr int NW = 3;  // [0] rows, [1..2] sum128
  static constexpr bool PIPELINED = false;
  static __device__ __forceinline__ void init(u64* a) { a[0] = a[1] = a[2] = 0; }
  static __device__ __forceinline__ void combine(u64* a, const u64* b) {
    acc_add64(a, b);
    acc_add128(a + 1, b + 1);
  }
  static __device__ __forceinline__ void tile(const CometKParams& prm, i64 base, i64 n, u64* acc) {
    bool k[R];
    i64 idx[R];
    i32 sd[R];
    i64 disc[R], qty[R], price[R];
#pragma unroll
    for (int r = 0; r < R; r++) { idx[r] = base + (i64)r * kBlock + threadIdx.x; k[r] = idx[r] < n; }
#pragma unroll
    for (int r = 0; r < R; r++) if (k[r]) sd[r] = ld<i32>(prm.in[3], idx[r]);
#pragma unroll
    for (int r = 0; r < R; r++) if (k[r]) k[r] = sd[r] >= (i32)prm.iarg[1] && sd[r] < (i32)prm.iarg[2];
#pragma unroll
    for (int r = 0; r < R; r++) if (k[r]) disc[r] = ld_dec_lo(prm.in[2], idx[r]);
#pragma unroll
    for (int r = 0; r < R; r++) if (k[r]) k[r] = disc[r] >= prm.iarg[3] && disc[r] <= prm.iarg[4];
#pragma unroll
    for (int r = 0; r < R; r++) if (k[r]) qty[r] = ld_dec_lo(prm.in[0], idx[r]);
#pragma unroll
    for (int r = 0; r < R; r++) if (k[r]) k[r] = qty[r] < prm.iarg[5];
#pragma unroll
    for (int r = 0; r < R; r++) if (k[r]) price[r] = ld_dec_lo(prm.in[1], idx[r]);
#pragma unroll
    for (int r = 0; r < R; r++) if (k[r]) {
      acc[0] += 1;
      acc_feed_i128(acc + 1, (i128)price[r] * (i128)disc[r]);
    }
  }
  static __device__ __forceinline__ void kexport(const CometKParams&, const u64*) {}
  static __device__ __forceinline__ void finalize(const CometKParams& prm, const u64* acc) {
    ((i128*)prm.out[4])[0] = mk128(acc[2], acc[1]);
    ((u8*)prm.out[5])[0] = 1;
    ((u8*)prm.out[6])[0] = acc[0] == 0 ? 1 : 0;
  }
};

extern "C" __global__ __launch_bounds__(256) void comet_q6_static_agg(const CometKParams prm) { agg_nogroup_body<Q6Static>(prm); }
extern "C" __global__ __launch_bounds__(256) void comet_q6_static_final(const CometKParams prm) { agg_nogroup_final_body<Q6Static>(prm); }

// ---- direct-map joins (comet_device.hpp JoinDirectTable): keys set per 128-bit block of a build side's key bitmap
__global__ __launch_bounds__(256) void popcount128_kernel(const uint4* blocks, long long n, unsigned int* counts) {
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) {
    const uint4 w = blocks[k];
    counts[k] = (unsigned int)(__popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w));
  }
}
extern "C" int comet_launch_popcount128(const void* blocks, int64_t n, uint32_t* counts, void* stream) {
  if (n > 0) {
    const long long g = (n + 255) / 256;
    hipLaunchKernelGGL(popcount128_kernel, dim3((unsigned)(g < 256 * 16 ? g : 256 * 16)), dim3(256), 0, (hipStream_t)stream, (const uint4*)blocks, (long long)n, counts);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- bucket-table joins (comet_device.hpp template D''): cnt[b][p] = leaders of partition p that block b of the partition passes saw → in place its
// exclusive prefix over the blocks (where block b's records of partition p start inside the partition), tot[p] = the partition's size
__global__ __launch_bounds__(256) void join_part_scan_kernel(unsigned int* cnt, int g, int np, unsigned int* tot, unsigned int limit, unsigned long long* over) {
  const int p = (int)(blockIdx.x * 256 + threadIdx.x);
  if (p >= np) return;
  unsigned int run = 0;
  int b = 0;
  for (; b + 8 <= g; b += 8) {
    unsigned int c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = cnt[(long long)(b + k) * np + p];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      cnt[(long long)(b + k) * np + p] = run;
      run += c[k];
    }
  }
  for (; b < g; b++) {
    const unsigned int c = cnt[(long long)b * np + p];
    cnt[(long long)b * np + p] = run;
    run += c;
  }
  tot[p] = run;
  if (run > limit) *(volatile unsigned long long*)over = 1ull;      // the partition would not fit its table (the executor asks before it builds with the monotone hash)
}
extern "C" int comet_launch_join_part_scan(uint32_t* cnt, int g, int np, uint32_t* tot, uint32_t limit, uint64_t* over, void* stream) {
  if (g > 0 && np > 0)
    hipLaunchKernelGGL(join_part_scan_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cnt, g, np, tot, (unsigned int)limit, (unsigned long long*)over);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
