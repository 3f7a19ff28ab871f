// TEST INFRASTRUCTURE: csrc/device/pq_runs.hpp — the walk the gfx950 kernels pq_count_runs_kernel / pq_write_runs_kernel run, one lane per page —
// compiled for the CPU.  tests/test_parquet_runs_emu_cpu.py feeds it hand-built and random RLE / bit-packed hybrid sections and compares the
// runs with an independent Python reading of the format.
#define PQ_RUNS_HOST 1
#include "device/pq_runs.hpp"

extern "C" int pq_runs_emu(const uint8_t* bytes, int64_t begin, int64_t end, int bw, int32_t max_values, int64_t* out5, int cap) {
  int32_t n = 0;
  int k = 0;
  const int st = pq_walk_runs(bytes, begin, end, bw, max_values, &n, [&](int64_t byte_off, int32_t value_start, int32_t count, int is_rle, uint32_t rle_value) {
    if (k < cap) {
      out5[5 * k + 0] = byte_off;
      out5[5 * k + 1] = value_start;
      out5[5 * k + 2] = count;
      out5[5 * k + 3] = is_rle;
      out5[5 * k + 4] = (int64_t)rle_value;
    }
    k++;
  });
  if (st != PQ_RUNS_OK) return -st;
  return n == k ? n : -100;
}
