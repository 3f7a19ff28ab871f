// split(str, pattern, limit) over a Utf8 column (string_funcs/split.rs:434-472): the column's rows become lists of pieces.  Two passes of
// device/regex_vm.hpp's rx_split per row — count the pieces, then (behind the prefix sum over the counts) describe each piece as a view of
// its source value (comet_device.hpp strview: source row, first byte, byte count); the executor assembles the element column from the
// views with the kernels every string view uses (exchange_kernels.hip).  The pattern is a group-0 program of the matcher in device memory.
#include <hip/hip_runtime.h>

#include "device/comet_device.hpp"
#define RXVM_FN __device__ inline
#define RXVM_ENTRY __device__ __noinline__
#include "device/regex_vm.hpp"

using namespace comet;

namespace {

// (prog2 set: regexp_extract_all — prog drives the iteration over the matches, prog2 reports the wanted group; string_funcs/regexp_extract_all.rs:79-108)
__global__ __launch_bounds__(256) void split_count_kernel(const i32* offs, const u8* bytes, const u8* valid_bits, i64 vfirst, i64 n, const u32* prog, const u32* prog2, i32 limit, u32* counts) {
  for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < n; r += (i64)gridDim.x * 256) {
    const bool ok = !valid_bits || ((valid_bits[(vfirst + r) >> 3] >> ((vfirst + r) & 7)) & 1);
    u32 c = 0;
    if (ok) {
      const i32 lo = offs[r], len = offs[r + 1] - lo;
      c = prog2 ? (u32)rx_find_all(prog, prog2, bytes + lo, len, 0, [](i32, i32, i32) {}) : (u32)rx_split(prog, bytes + lo, len, limit, 0, [](i32, i32, i32) {});
    }
    counts[r] = c;
  }
}

__global__ __launch_bounds__(256) void split_write_kernel(const i32* offs, const u8* bytes, const u8* valid_bits, i64 vfirst, i64 n, const u32* prog, const u32* prog2, i32 limit, const i32* list_offs,
                                                          strview* views) {
  for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < n; r += (i64)gridDim.x * 256) {
    const bool ok = !valid_bits || ((valid_bits[(vfirst + r) >> 3] >> ((vfirst + r) & 7)) & 1);
    if (!ok) continue;
    const i32 first = list_offs[r], cnt = list_offs[r + 1] - first;
    const i32 lo = offs[r], len = offs[r + 1] - lo;
    strview* out = views + first;
    auto put = [&](i32 k, i32 a, i32 m) { out[k] = strview{(u32)r, (u32)a, (u32)m, 0u}; };
    if (prog2) rx_find_all(prog, prog2, bytes + lo, len, cnt, put);
    else rx_split(prog, bytes + lo, len, limit, cnt, put);
  }
}

inline dim3 grid_rows(i64 n) { return dim3((unsigned)((n + 255) / 256 < 256 * 16 ? (n + 255) / 256 : 256 * 16)); }

}  // namespace

// offs: the column's offsets at its first row (the view's offset applied by the caller); row r's validity is bit valid_first + r of valid_bits (null: every row is valid)
extern "C" int comet_launch_split_count(const int32_t* offs, const uint8_t* bytes, const uint8_t* valid_bits, int64_t valid_first, int64_t n, const uint32_t* prog, const uint32_t* prog2, int32_t limit, uint32_t* counts,
                                        void* stream) {
  if (n > 0) hipLaunchKernelGGL(split_count_kernel, grid_rows(n), 256, 0, (hipStream_t)stream, offs, bytes, valid_bits, (i64)valid_first, (i64)n, prog, prog2, limit, counts);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int comet_launch_split_write(const int32_t* offs, const uint8_t* bytes, const uint8_t* valid_bits, int64_t valid_first, int64_t n, const uint32_t* prog, const uint32_t* prog2, int32_t limit,
                                        const int32_t* list_offs, void* views, void* stream) {
  if (n > 0) hipLaunchKernelGGL(split_write_kernel, grid_rows(n), 256, 0, (hipStream_t)stream, offs, bytes, valid_bits, (i64)valid_first, (i64)n, prog, prog2, limit, list_offs, (strview*)views);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
