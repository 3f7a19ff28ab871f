// Utf8 keys of any length for group-by and join (SURVEY §8 f2 "Utf8 keys at scale").
//
// The fused aggregate / join kernels carry keys as 64-bit words; strings of up to 15 bytes travel packed in two of them.  Longer
// strings are first replaced by a REPRESENTATIVE ROW INDEX: an open-addressing table of u32 slots (row + 1, 0 = empty) keyed by the
// string bytes maps every row to the first-inserted row that holds an equal string.  Equality is decided on the bytes (the hash only
// picks the starting slot), so the mapping is exact; the aggregate then groups on the 8-byte index and the emit step gathers the
// string of each group from its representative row.  For joins the build side fills the table and the probe side looks its
// strings up in it (miss → no partner).  NULL strings never enter the table: the index column shares the string column's validity.
//
// The slot IS the whole entry (strings are read from the immutable input column), so one compare-and-swap publishes an insert:
// no payload to wait for.  A plain load runs first and the CAS only on empty slots — with few distinct strings every later row
// sees the winner in cache instead of hammering one address with atomics.
#include <hip/hip_runtime.h>

#include "device/comet_device.hpp"

using namespace comet;

namespace {

__device__ __forceinline__ u64 hash_bytes(const u8* p, i32 n) {
  // 8 bytes per step: multiply-xorshift mixing (the hash only spreads rows over slots; equality is decided on the bytes)
  u64 h = 0x9E3779B97F4A7C15ull ^ (u64)n;
  i32 k = 0;
  for (; k + 8 <= n; k += 8) {
    u64 w = 0;
    for (int b = 0; b < 8; b++) w |= (u64)p[k + b] << (8 * b);
    h = (h ^ w) * 0xFF51AFD7ED558CCDull;
    h ^= h >> 32;
  }
  u64 w = 0;
  for (int b = 0; k < n; k++, b++) w |= (u64)p[k] << (8 * b);
  h = (h ^ w) * 0xC4CEB9FE1A85EC53ull;
  h ^= h >> 29;
  return h;
}

__device__ __forceinline__ bool same_bytes(const u8* a, const u8* b, i32 n) {
  for (i32 k = 0; k < n; k++)
    if (a[k] != b[k]) return false;
  return true;
}

__global__ __launch_bounds__(256) void str_max_len_kernel(const i32* __restrict__ offs, i64 n, u32* __restrict__ out) {
  u32 m = 0;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const u32 l = (u32)(offs[i + 1] - offs[i]);
    m = l > m ? l : m;
  }
  for (int s = 32; s >= 1; s >>= 1) {
    const u32 o = (u32)__shfl_xor((int)m, s, kWave);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & (kWave - 1)) == 0 && m) atomicMax(out, m);
}

// rep[i] = row index of the representative of row i's string; rows with a NULL string get 0 (their validity bit says NULL)
__global__ __launch_bounds__(256) void str_dict_build_kernel(const i32* __restrict__ offs, const u8* __restrict__ bytes,
                                                             const u8* __restrict__ valid_bits, i64 n, u32* __restrict__ table, u64 mask,
                                                             i64* __restrict__ rep) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    if (valid_bits && !((valid_bits[i >> 3] >> (i & 7)) & 1)) { rep[i] = 0; continue; }
    const i32 lo = offs[i], len = offs[i + 1] - lo;
    const u8* p = bytes + lo;
    u64 pos = hash_bytes(p, len) & mask;
    while (true) {
      u32 cur = __hip_atomic_load(&table[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == 0) {
        cur = atomicCAS(&table[pos], 0u, (u32)i + 1u);
        if (cur == 0) { rep[i] = i; break; }
      }
      const i64 r = (i64)cur - 1;
      const i32 rlo = offs[r];
      if (offs[r + 1] - rlo == len && same_bytes(bytes + rlo, p, len)) { rep[i] = r; break; }
      pos = (pos + 1) & mask;
    }
  }
}

// probe side of a join: rep[j] = representative BUILD row of an equal string, ok[j] = 1; no equal build string (or NULL) → ok[j] = 0
__global__ __launch_bounds__(256) void str_dict_lookup_kernel(const i32* __restrict__ boffs, const u8* __restrict__ bbytes,
                                                              const u32* __restrict__ table, u64 mask, const i32* __restrict__ offs,
                                                              const u8* __restrict__ bytes, const u8* __restrict__ valid_bits, i64 n,
                                                              i64* __restrict__ rep, u8* __restrict__ ok) {
  for (i64 j = (i64)blockIdx.x * 256 + threadIdx.x; j < n; j += (i64)gridDim.x * 256) {
    rep[j] = 0;
    ok[j] = 0;
    if (valid_bits && !((valid_bits[j >> 3] >> (j & 7)) & 1)) continue;
    const i32 lo = offs[j], len = offs[j + 1] - lo;
    const u8* p = bytes + lo;
    u64 pos = hash_bytes(p, len) & mask;
    while (true) {
      const u32 cur = table[pos];
      if (cur == 0) break;
      const i64 r = (i64)cur - 1;
      const i32 rlo = boffs[r];
      if (boffs[r + 1] - rlo == len && same_bytes(bbytes + rlo, p, len)) { rep[j] = r; ok[j] = 1; break; }
      pos = (pos + 1) & mask;
    }
  }
}

// packed strings (≤ 15 bytes: bytes 0-7 in word a, 8-14 in the low 56 bits of word b, length in b's top byte) → Arrow lengths / bytes
__global__ __launch_bounds__(256) void str16_lengths_kernel(const u64* __restrict__ packed, const u8* __restrict__ ok, i64 n, u32* __restrict__ lengths) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) lengths[i] = (ok && !ok[i]) ? 0u : (u32)(packed[2 * i + 1] >> 56);
}
__global__ __launch_bounds__(256) void str16_copy_kernel(const u64* __restrict__ packed, const i32* __restrict__ offs, i64 n, u8* __restrict__ bytes) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const i32 lo = offs[i], len = offs[i + 1] - lo;
    const u64 a = packed[2 * i], b = packed[2 * i + 1];
    for (i32 k = 0; k < len; k++) bytes[lo + k] = (u8)(k < 8 ? a >> (8 * k) : b >> (8 * (k - 8)));
  }
}

int grid_for(i64 n) {
  i64 g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace

extern "C" {

// *out_max (device u32, zeroed by the caller) = longest value of the column in bytes
int comet_launch_str_max_len(const int32_t* offs, int64_t n, uint32_t* out_max, void* stream) {
  if (n > 0) hipLaunchKernelGGL(str_max_len_kernel, grid_for(n), 256, 0, (hipStream_t)stream, offs, (i64)n, out_max);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// table: `slots` zeroed u32 entries, slots a power of two ≥ 2·n; n < 2^32 − 1
int comet_launch_str_dict_build(const int32_t* offs, const uint8_t* bytes, const uint8_t* valid_bits, int64_t n, uint32_t* table, int64_t slots,
                                int64_t* rep, void* stream) {
  if (n > 0)
    hipLaunchKernelGGL(str_dict_build_kernel, grid_for(n), 256, 0, (hipStream_t)stream, offs, bytes, valid_bits, (i64)n, table, (u64)slots - 1, (i64*)rep);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_str_dict_lookup(const int32_t* build_offs, const uint8_t* build_bytes, const uint32_t* table, int64_t slots, const int32_t* offs,
                                 const uint8_t* bytes, const uint8_t* valid_bits, int64_t n, int64_t* rep, uint8_t* ok, void* stream) {
  if (n > 0)
    hipLaunchKernelGGL(str_dict_lookup_kernel, grid_for(n), 256, 0, (hipStream_t)stream, build_offs, build_bytes, table, (u64)slots - 1, offs, bytes,
                       valid_bits, (i64)n, (i64*)rep, ok);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int comet_launch_str16_lengths(const void* packed, const uint8_t* ok_bytes, int64_t n, uint32_t* lengths, void* stream) {
  if (n > 0) hipLaunchKernelGGL(str16_lengths_kernel, grid_for(n), 256, 0, (hipStream_t)stream, (const u64*)packed, ok_bytes, (i64)n, lengths);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_str16_copy(const void* packed, const int32_t* offsets, int64_t n, uint8_t* bytes, void* stream) {
  if (n > 0) hipLaunchKernelGGL(str16_copy_kernel, grid_for(n), 256, 0, (hipStream_t)stream, (const u64*)packed, offsets, (i64)n, bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // extern "C"
