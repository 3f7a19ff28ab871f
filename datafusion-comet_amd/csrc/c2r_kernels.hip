// Columnar → Spark UnsafeRow (native/core/src/execution/columnar_to_row.rs:949-1345): the off-ramp where Comet hands rows
// to operators Spark runs itself.  Row layout: null bitset (one bit per field, 8-byte words) | one 8-byte slot per field | variable-length
// data, each value padded to 8 bytes; a variable-length slot holds (offset from the row start << 32) | length.  Following the reference:
// integers are sign-extended into their slot, floats store their bits, Decimal128 of precision ≤ 18 stores the unscaled long, wider
// decimals store their minimal big-endian two's-complement bytes as variable-length data, NULL and zero-length values leave the slot 0.
// Two kernels over the resident columns: row sizes (then a prefix sum), and one thread per row writing its bytes.
#include <hip/hip_runtime.h>

#include "device/comet_device.hpp"

using namespace comet;

extern "C" {
typedef struct C2RCol {
  const void* values;      // fixed-width values / boolean bits / int32 offsets
  const u8* valid_bits;    // Arrow validity bitmap or NULL
  const u8* data;          // Utf8 / Binary bytes
  int kind;                // 0 bool, 1 int8, 2 int16, 3 int32 (date32), 4 int64 (timestamp), 5 float, 6 double, 7 decimal ≤ 18, 8 decimal > 18, 9 utf8 / binary
  int pad;
} C2RCol;
}

namespace {

__device__ __forceinline__ bool col_valid(const C2RCol& c, i64 i) { return !c.valid_bits || ((c.valid_bits[i >> 3] >> (i & 7)) & 1); }

// minimal big-endian two's-complement length of a 128-bit value (i128_to_spark_decimal_bytes, columnar_to_row.rs:1532-1558)
__device__ __forceinline__ int dec_be_len(i128 v) {
  int start = 0;
  const u8 sign = v < 0 ? 0xFF : 0x00;
  while (start < 15) {
    const u8 b = (u8)((u128)v >> (8 * (15 - start)));
    const u8 nb = (u8)((u128)v >> (8 * (14 - start)));
    if (b != sign || ((nb & 0x80) != 0) != (v < 0)) break;
    start++;
  }
  return 16 - start;
}

__device__ __forceinline__ i32 var_len(const C2RCol& c, i64 i) {
  if (c.kind == 9) {
    const i32* off = (const i32*)c.values;
    return off[i + 1] - off[i];
  }
  return dec_be_len(((const i128*)c.values)[i]);
}

__global__ __launch_bounds__(256) void c2r_sizes_kernel(const C2RCol* __restrict__ cols, int ncols, i64 n, int fixed_size, u32* __restrict__ sizes) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    u32 s = (u32)fixed_size;
    for (int c = 0; c < ncols; c++) {
      const C2RCol& col = cols[c];
      if (col.kind < 8 || !col_valid(col, i)) continue;
      s += ((u32)var_len(col, i) + 7u) & ~7u;
    }
    sizes[i] = s;
  }
}

__global__ __launch_bounds__(256) void c2r_write_kernel(const C2RCol* __restrict__ cols, int ncols, i64 n, int bitset_bytes, const i32* __restrict__ row_off,
                                                        u8* __restrict__ out, i32* __restrict__ lengths) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const i32 start = row_off[i], end = row_off[i + 1];
    u8* row = out + start;
    u64* words = (u64*)row;                       // rows start 8-aligned: every size is a multiple of 8
    const int fixed_words = bitset_bytes / 8 + ncols;
    for (int w = 0; w < fixed_words; w++) words[w] = 0;
    i32 cursor = fixed_words * 8;
    for (int c = 0; c < ncols; c++) {
      const C2RCol& col = cols[c];
      if (!col_valid(col, i)) {
        words[c >> 6] |= 1ull << (c & 63);
        continue;
      }
      u64 slot = 0;
      switch (col.kind) {
        case 0: slot = (((const u8*)col.values)[i >> 3] >> (i & 7)) & 1; break;
        case 1: slot = (u64)(i64)((const i8*)col.values)[i]; break;
        case 2: slot = (u64)(i64)((const i16*)col.values)[i]; break;
        case 3: slot = (u64)(i64)((const i32*)col.values)[i]; break;
        case 4: slot = (u64)((const i64*)col.values)[i]; break;
        case 5: slot = (u64)((const u32*)col.values)[i]; break;            // f32::to_bits() as i64
        case 6: slot = ((const u64*)col.values)[i]; break;
        case 7: slot = (u64)(i64)((const i128*)col.values)[i]; break;      // unscaled value as long
        case 8: {
          const i128 v = ((const i128*)col.values)[i];
          const int len = dec_be_len(v);
          for (int b = 0; b < len; b++) row[cursor + b] = (u8)((u128)v >> (8 * (len - 1 - b)));
          const int padded = (len + 7) & ~7;
          for (int b = len; b < padded; b++) row[cursor + b] = 0;
          slot = ((u64)(u32)cursor << 32) | (u64)(u32)len;
          cursor += padded;
          break;
        }
        default: {
          const i32* off = (const i32*)col.values;
          const i32 lo = off[i], len = off[i + 1] - lo;
          if (len > 0) {
            const u8* src = col.data + lo;
            for (i32 b = 0; b < len; b++) row[cursor + b] = src[b];
            const i32 padded = (len + 7) & ~7;
            for (i32 b = len; b < padded; b++) row[cursor + b] = 0;
            slot = ((u64)(u32)cursor << 32) | (u64)(u32)len;
            cursor += padded;
          }
          break;
        }
      }
      words[bitset_bytes / 8 + c] = slot;
    }
    lengths[i] = end - start;
  }
}

int grid_for(i64 n) {
  i64 g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

}  // namespace

extern "C" {

int comet_launch_c2r_sizes(const C2RCol* dev_cols, int ncols, int64_t n, int fixed_size, uint32_t* sizes, void* stream) {
  if (n > 0) hipLaunchKernelGGL(c2r_sizes_kernel, grid_for(n), 256, 0, (hipStream_t)stream, dev_cols, ncols, (i64)n, fixed_size, sizes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_c2r_write(const C2RCol* dev_cols, int ncols, int64_t n, int bitset_bytes, const int32_t* row_offsets, uint8_t* out, int32_t* lengths, void* stream) {
  if (n > 0) hipLaunchKernelGGL(c2r_write_kernel, grid_for(n), 256, 0, (hipStream_t)stream, dev_cols, ncols, (i64)n, bitset_bytes, row_offsets, out, lengths);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // extern "C"
