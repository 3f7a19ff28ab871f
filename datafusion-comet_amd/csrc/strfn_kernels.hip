// Derived Utf8 columns (codegen.hpp DerivedCol kind 3): a string function of a source column whose result is new bytes — reverse, repeat, replace,
// substring_index, md5 / sha1 / sha2 as hexadecimal digits (device/strfn.hpp) — computed over the chain's source table before the fused kernel runs: one pass
// for every row's result length, a prefix sum (the column's offsets), one pass that writes.  One thread per row.
#include <hip/hip_runtime.h>

#include "device/comet_device.hpp"      // (includes device/strfn.hpp)

using namespace comet;

namespace {

__global__ __launch_bounds__(256) void strfn_len_kernel(int op, const i32* offs, const u8* bytes, const u8* valid_bits, i64 vfirst, i64 n, const u8* a, i32 na, const u8* b, i32 nb, i64 k, u32* lengths,
                                                        u32* too_long) {
  for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < n; r += (i64)gridDim.x * 256) {
    const bool ok = !valid_bits || ((valid_bits[(vfirst + r) >> 3] >> ((vfirst + r) & 7)) & 1);
    i64 len = 0;
    if (ok) {
      const i32 lo = offs[r];
      len = sf_len(op, bytes + lo, offs[r + 1] - lo, a, na, b, nb, k);
      if (len > 0x7fffffffll) { atomicOr(too_long, 1u); len = 0; }
    }
    lengths[r] = (u32)len;
  }
}

__global__ __launch_bounds__(256) void strfn_write_kernel(int op, const i32* offs, const u8* bytes, const u8* valid_bits, i64 vfirst, i64 n, const u8* a, i32 na, const u8* b, i32 nb, i64 k,
                                                          const i32* out_offs, u8* out) {
  for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < n; r += (i64)gridDim.x * 256) {
    const bool ok = !valid_bits || ((valid_bits[(vfirst + r) >> 3] >> ((vfirst + r) & 7)) & 1);
    if (!ok || out_offs[r + 1] == out_offs[r]) continue;
    const i32 lo = offs[r];
    sf_write(op, bytes + lo, offs[r + 1] - lo, a, na, b, nb, k, out + out_offs[r]);
  }
}

inline dim3 grid_rows(i64 n) { return dim3((unsigned)((n + 255) / 256 < 256 * 16 ? (n + 255) / 256 : 256 * 16)); }

}  // namespace

// offs: the column's offsets at its first row; row r's validity is bit valid_first + r of valid_bits (null: every row is valid); a / b: the literal arguments in
// device memory; too_long: set when a row's result would pass 2 GiB
extern "C" int comet_launch_strfn_len(int op, const int32_t* offs, const uint8_t* bytes, const uint8_t* valid_bits, int64_t valid_first, int64_t n, const uint8_t* a, int32_t na, const uint8_t* b,
                                      int32_t nb, int64_t k, uint32_t* lengths, uint32_t* too_long, void* stream) {
  if (n > 0) hipLaunchKernelGGL(strfn_len_kernel, grid_rows(n), 256, 0, (hipStream_t)stream, op, offs, bytes, valid_bits, (i64)valid_first, (i64)n, a, na, b, nb, (i64)k, lengths, too_long);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int comet_launch_strfn_write(int op, const int32_t* offs, const uint8_t* bytes, const uint8_t* valid_bits, int64_t valid_first, int64_t n, const uint8_t* a, int32_t na, const uint8_t* b,
                                        int32_t nb, int64_t k, const int32_t* out_offs, uint8_t* out, void* stream) {
  if (n > 0) hipLaunchKernelGGL(strfn_write_kernel, grid_rows(n), 256, 0, (hipStream_t)stream, op, offs, bytes, valid_bits, (i64)valid_first, (i64)n, a, na, b, nb, (i64)k, out_offs, out);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
