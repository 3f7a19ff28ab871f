// The run headers of an RLE / bit-packed hybrid section (Parquet Encodings.md: <varint header><payload>; header LSB 1 = bit-packed groups of
// eight values, LSB 0 = one repeated value of ceil(bit width / 8) bytes), walked WHERE THE SECTION LIES.  The host walks the sections it can
// see (parquet_scan.cpp parse_hybrid_runs: uncompressed pages, snappy pages read through the compressed stream); the index sections of
// dictionary-encoded pages the DEVICE inflates (zstd: entropy-coded, nothing to see through) used to come back over PCIe for this walk —
// tens of MB per scan.  This is the same walk as a device function: one lane per page (pq_count_runs_kernel / pq_write_runs_kernel), every
// header fetched with eight independent byte loads, so a run costs one memory latency; a page's few dozen to few hundred runs are a
// chain of that many latencies, thousands of pages walk side by side.
//
// One source for both sides: PQ_RUNS_HOST compiles it for the CPU, where tests/test_page_codecs_cpu.py checks it against the host parser
// on random sections (tests/emu/pq_runs_emu.cpp).
#pragma once
#include <stdint.h>

#include "../parquet_dev.h"

#ifdef PQ_RUNS_HOST
#define PQ_RUNS_FN static inline
#else
#define PQ_RUNS_FN __device__ __forceinline__
#endif

// status of a walk
enum { PQ_RUNS_OK = 0, PQ_RUNS_TRUNCATED_HEADER = 1, PQ_RUNS_TRUNCATED_RLE = 2, PQ_RUNS_BAD_WIDTH = 3, PQ_RUNS_BAD_COUNT = 4, PQ_RUNS_TRUNCATED_PACKED = 5 };

// Walks the section bytes[begin, end) and calls emit(byte_off, value_start, count, is_rle, rle_value) for every run that holds values, in
// order; stops behind max_values values when max_values >= 0.  Mirrors parse_hybrid_runs_from (parquet_scan.cpp) decision for decision:
// a bit-packed run counts whole groups (the last group's padding included), an empty run is skipped, a header or an RLE value that runs past
// `end` is an error.  Returns the status; *n_runs = runs emitted.
template <class Emit>
PQ_RUNS_FN int pq_walk_runs(const uint8_t* bytes, int64_t begin, int64_t end, int bw, int32_t max_values, int32_t* n_runs, Emit emit) {
  int64_t pos = begin;
  int32_t vstart = 0, n = 0;
  *n_runs = 0;
  if (bw < 0 || bw > 32) return PQ_RUNS_BAD_WIDTH;
  const int vbytes = (bw + 7) >> 3;
  while (pos < end && (max_values < 0 || vstart < max_values)) {
    // the next eight bytes, as far as the section goes (independent loads: one latency)
    uint64_t w = 0;
    const int avail = end - pos < 8 ? (int)(end - pos) : 8;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (k < avail) w |= (uint64_t)bytes[pos + k] << (8 * k);
    // varint header: at most five bytes for a 32-bit count (the host accepts longer ones; no writer emits them — treated as truncated)
    uint64_t h = 0;
    int used = 0;
    bool done = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (!done) {
        if (k >= avail) return PQ_RUNS_TRUNCATED_HEADER;
        const uint32_t b = (uint32_t)(w >> (8 * k)) & 0xffu;
        h |= (uint64_t)(b & 0x7fu) << (7 * k);
        used = k + 1;
        done = !(b & 0x80u);
      }
    }
    if (!done) return PQ_RUNS_TRUNCATED_HEADER;
    pos += used;
    int32_t count;
    if (h & 1) {
      const int64_t groups = (int64_t)(h >> 1);
      if (groups > (int64_t)(INT32_MAX - vstart) / 8) return PQ_RUNS_BAD_COUNT;      // (a header of up to 56 bits: the count must stay a value index)
      count = (int32_t)(groups * 8);
      // the payload ends inside the section — or, when the page's value count is known, at least the values still wanted do (some writers cut
      // the last group's padding)
      if (pos + groups * bw > end) {
        const int64_t wanted = max_values >= 0 ? (int64_t)(max_values - vstart) : (int64_t)count;
        if (max_values < 0 || pos + (wanted * bw + 7) / 8 > end) return PQ_RUNS_TRUNCATED_PACKED;
      }
      if (count != 0) { emit(pos, vstart, count, 0, 0u); n++; }
      pos += groups * bw;
    } else {
      if ((h >> 1) > (uint64_t)(INT32_MAX - vstart)) return PQ_RUNS_BAD_COUNT;
      count = (int32_t)(h >> 1);
      if (pos + vbytes > end) return PQ_RUNS_TRUNCATED_RLE;
      uint32_t v = 0;
      if (used + vbytes <= avail) {      // the value came with the header's eight bytes
        v = (uint32_t)((w >> (8 * used)) & (vbytes >= 4 ? 0xffffffffull : ((1ull << (8 * vbytes)) - 1)));
      } else {
        for (int k = 0; k < vbytes; k++) v |= (uint32_t)bytes[pos + k] << (8 * k);
      }
      pos += vbytes;
      if (count != 0) { emit((int64_t)0, vstart, count, 1, v); n++; }
    }
    vstart += count;
  }
  *n_runs = n;
  return PQ_RUNS_OK;
}
