// Exchange (shuffle) step on device: partition ids → partition_starts + partition_row_indices → per-column take.
// This is the GPU form of the reference's hash repartitioner scratch computation
// (native/shuffle/src/partitioners/multi_partition.rs:54-103: count per partition, prefix sum, row indices grouped by
// partition in ascending row order) followed by the per-partition `take` that builds the outgoing batches
// (multi_partition.rs:457-520).  Partition ids come from comet_murmur3_column + comet_pmod_partition.
//
// Layout of the work: one wave owns a tile of kPartTile consecutive rows.  Pass 1 counts rows per partition per tile
// into hist[p * W + tile]; an exclusive scan over that partition-major array yields, for every (partition, tile), the
// first output slot; pass 2 replays the tile in row order and writes row i to its slot, so rows keep their input
// order inside each partition (same result as the reference's reverse fill).
#include <hip/hip_runtime.h>

#include <cstring>

#include "device/comet_device.hpp"

using namespace comet;

namespace {

constexpr int kPartTileMax = 8192;  // rows per wave tile for large inputs
// rows per wave tile: large tiles keep the histogram small, but a small input must still fill the GPU (≥ ~4096 waves)
__host__ __device__ inline i64 part_tile_rows(i64 n) {
  i64 t = n / 4096;
  t = (t + 63) / 64 * 64;
  return t < 512 ? 512 : (t > kPartTileMax ? kPartTileMax : t);
}

// Visit the groups of equal partition id among the active lanes of this wave, in lane order of their first member.
// f(leader_lane, pid_of_group, member_mask) is called uniformly by the whole wave.
template <class F>
__device__ __forceinline__ void for_each_group(i32 pid, bool active, F f) {
  u64 remaining = __ballot(active);
  while (remaining) {
    const int l = __ffsll((unsigned long long)remaining) - 1;
    const i32 lp = __shfl(pid, l, kWave);
    const u64 m = __ballot(active && pid == lp);
    f(l, lp, m);
    remaining &= ~m;
  }
}

__global__ __launch_bounds__(256) void part_hist_kernel(const i32* pids, i64 n, i32 P, i64 W, u64* hist, u32* bad) {
  extern __shared__ u32 s_cnt[];  // [4][P]
  const i64 tile = part_tile_rows(n);
  u32* mine = s_cnt + wave_id() * P;
  const i64 block_tiles = (W + 3) / 4;
  for (i64 bt = blockIdx.x; bt < block_tiles; bt += gridDim.x) {
    const i64 g = bt * 4 + wave_id();
    for (int p = lane_id(); p < P; p += kWave) mine[p] = 0;
    __syncthreads();
    if (g < W) {
      const i64 lo = g * tile, hi = lo + tile < n ? lo + tile : n;
      for (i64 base = lo; base < hi; base += kWave) {
        const i64 i = base + lane_id();
        const bool active = i < hi;
        i32 pid = active ? pids[i] : 0;
        if (active && (pid < 0 || pid >= P)) { *bad = 1; pid = 0; }
        for_each_group(pid, active, [&](int l, i32 lp, u64 m) {
          if (lane_id() == l) atomicAdd(&mine[lp], (u32)__popcll(m));
        });
      }
    }
    __syncthreads();
    if (g < W)
      for (int p = lane_id(); p < P; p += kWave) hist[(i64)p * W + g] = mine[p];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void part_scan_kernel(u64* hist, i64 len) { tile_scan_body(hist, len); }

// The same exclusive scan spread over many blocks for long histograms (a radix-sort pass over 20 M rows scans 1 M counters; one block
// needs 240 µs for that): chunk sums → scan of the sums (one block) → every block scans its chunk of 4096 counters from its carry.
constexpr i64 kScanChunk = 4096;
__global__ __launch_bounds__(256) void part_scan_sums_kernel(const u64* __restrict__ hist, i64 len, u64* __restrict__ sums) {
  __shared__ u64 s_part[256];
  const i64 first = (i64)blockIdx.x * kScanChunk + (i64)threadIdx.x * 16;
  u64 s = 0;
  for (int j = 0; j < 16; j++) s += first + j < len ? hist[first + j] : 0;
  s_part[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st) s_part[threadIdx.x] += s_part[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = s_part[0];
}
__global__ __launch_bounds__(256) void part_scan_apply_kernel(u64* __restrict__ hist, i64 len, const u64* __restrict__ sums, i64 nblocks) {
  __shared__ u64 s_wave[kBlock / kWave];
  const i64 first = (i64)blockIdx.x * kScanChunk + (i64)threadIdx.x * 16;
  u64 v[16];
  u64 sum = 0;
  for (int j = 0; j < 16; j++) {
    v[j] = first + j < len ? hist[first + j] : 0;
    sum += v[j];
  }
  u64 x = sum;
  for (int d = 1; d < kWave; d <<= 1) {
    u32 lo = __shfl_up((u32)x, d, kWave), hi = __shfl_up((u32)(x >> 32), d, kWave);
    u64 y = ((u64)hi << 32) | lo;
    if (lane_id() >= d) x += y;
  }
  if (lane_id() == kWave - 1) s_wave[wave_id()] = x;
  __syncthreads();
  u64 woff = 0;
  for (int w = 0; w < wave_id(); w++) woff += s_wave[w];
  u64 run = sums[blockIdx.x] + woff + x - sum;
  for (int j = 0; j < 16; j++) {
    if (first + j < len) hist[first + j] = run;
    run += v[j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) hist[len] = sums[nblocks];
}

__global__ __launch_bounds__(256) void part_starts_kernel(const u64* scanned, i32 P, i64 W, i64 n, i64* starts) {
  for (int p = threadIdx.x; p <= P; p += 256) starts[p] = p == P ? n : (i64)scanned[(i64)p * W];
}

__global__ __launch_bounds__(256) void part_index_kernel(const i32* pids, i64 n, i32 P, i64 W, const u64* scanned, u32* row_indices) {
  extern __shared__ u32 s_cnt[];
  const i64 tile = part_tile_rows(n);
  u32* run = s_cnt + wave_id() * P;
  const i64 block_tiles = (W + 3) / 4;
  const u64 lt = (1ull << lane_id()) - 1;
  for (i64 bt = blockIdx.x; bt < block_tiles; bt += gridDim.x) {
    const i64 g = bt * 4 + wave_id();
    if (g < W)
      for (int p = lane_id(); p < P; p += kWave) run[p] = (u32)scanned[(i64)p * W + g];
    __syncthreads();
    if (g < W) {
      const i64 lo = g * tile, hi = lo + tile < n ? lo + tile : n;
      for (i64 base = lo; base < hi; base += kWave) {
        const i64 i = base + lane_id();
        const bool active = i < hi;
        i32 pid = active ? pids[i] : 0;
        if (pid < 0 || pid >= P) pid = 0;
        u32 slot = 0;
        for_each_group(pid, active, [&](int l, i32 lp, u64 m) {
          u32 b = 0;
          if (lane_id() == l) b = atomicAdd(&run[lp], (u32)__popcll(m));
          b = __shfl(b, l, kWave);
          if (active && pid == lp) slot = b + (u32)__popcll(m & lt);
        });
        if (active) row_indices[slot] = (u32)i;
      }
    }
    __syncthreads();
  }
}

// take: dst[k] = src[idx[k]]
template <class T>
__global__ __launch_bounds__(256) void take_kernel(const T* src, const u32* idx, i64 n, T* dst) {
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < n; k += (i64)gridDim.x * 256) dst[k] = src[idx[k]];
}
// bit-packed source (validity bitmaps, Boolean values): one output byte (8 rows) per lane
__global__ __launch_bounds__(256) void take_bits_kernel(const u8* src, const u32* idx, i64 n, u8* dst) {
  const i64 nbytes = (n + 7) / 8;
  for (i64 b = (i64)blockIdx.x * 256 + threadIdx.x; b < nbytes; b += (i64)gridDim.x * 256) {
    u32 v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const i64 k = b * 8 + j;
      if (k < n) {
        const u32 r = idx[k];
        v |= (u32)((src[r >> 3] >> (r & 7)) & 1) << j;
      }
    }
    dst[b] = (u8)v;
  }
}

// ---- LSD radix sort over key byte planes (Sort operator): one pass = gather the pass's digit through the current permutation,
// group stably by digit with the partition kernels above (P = 256), compose the permutations with a take.
__global__ __launch_bounds__(256) void sort_iota_kernel(u32* perm, i64 n, u32 first) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) perm[i] = first + (u32)i;
}
__global__ __launch_bounds__(256) void sort_gather_digit_kernel(const u8* plane, const u32* perm, i64 n, i32* digit) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) digit[i] = (i32)plane[perm[i]];
}
// flags[b] = 1 iff plane b holds more than one distinct byte (constant planes need no pass)
__global__ __launch_bounds__(256) void sort_plane_varies_kernel(const u8* planes, i64 n, int W, u32* flags) {
  const int b = blockIdx.y;
  if (b >= W) return;
  const u8* plane = planes + (i64)b * n;
  const u8 first = plane[0];
  bool diff = false;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n && !diff; i += (i64)gridDim.x * 256) diff = plane[i] != first;
  if (__ballot(diff) != 0 && lane_id() == 0) flags[b] = 1;
}

// ---- TopK pre-selection (Sort with fetch ≪ rows): radix select from the most significant varying plane down.
// hist[d] = number of candidates whose digit is d
__global__ __launch_bounds__(256) void sort_hist256_kernel(const u8* plane, const u32* cand, i64 m, unsigned long long* hist) {
  __shared__ u32 s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < m; i += (i64)gridDim.x * 256) atomicAdd(&s_h[plane[cand[i]]], 1u);
  __syncthreads();
  if (s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
}
// candidates with digit < dstar are certainly among the first K: append them to `sure`; digit == dstar stay candidates
__global__ __launch_bounds__(256) void sort_select_kernel(const u8* plane, const u32* cand, i64 m, int dstar, u32* sure, u32* next_cand, u32* counters) {
  for (i64 base = (i64)blockIdx.x * 256; base < m; base += (i64)gridDim.x * 256) {
    const i64 i = base + threadIdx.x;
    const bool in = i < m;
    const u32 row = in ? cand[i] : 0;
    const int d = in ? (int)plane[row] : 256;
    const bool a = d < dstar, b = d == dstar;
    const u64 ma = __ballot(a), mb = __ballot(b);
    u32 offa = 0, offb = 0;
    if (lane_id() == 0) {
      if (ma) offa = atomicAdd(&counters[0], (u32)__popcll(ma));
      if (mb) offb = atomicAdd(&counters[1], (u32)__popcll(mb));
    }
    offa = __shfl(offa, 0, kWave);
    offb = __shfl(offb, 0, kWave);
    const u64 lt = (1ull << lane_id()) - 1;
    if (a) sure[offa + (u32)__popcll(ma & lt)] = row;
    if (b) next_cand[offb + (u32)__popcll(mb & lt)] = row;
  }
}

// What a TopK's radix select leaves — at most 4096 candidate rows — sorted by ONE workgroup: the varying key bytes of each row (at most 16:
// the planes that do not vary say nothing) packed big-endian into two words, a bitonic network over (key, row) in workgroup memory.  The
// LSD radix sort it replaces for such inputs took five launches per key byte — fifty for TPC-H Q3's top ten, 0.5 ms of launch latency.
struct SortSmallArgs { const u8* planes; i64 n; const u32* cand; u32* out; int m; int nv; int plane[16]; };
__global__ __launch_bounds__(1024) void sort_small_kernel(const SortSmallArgs a) {
  __shared__ u64 s_hi[4096], s_lo[4096];
  __shared__ u32 s_row[4096];
  int P = 2;
  while (P < a.m) P <<= 1;
  for (int i = threadIdx.x; i < P; i += 1024) {
    u64 hi = ~0ull, lo = ~0ull;
    u32 row = 0xffffffffu;
    if (i < a.m) {
      row = a.cand[i];
      hi = lo = 0;
      for (int b = 0; b < 16; b++) {
        const u64 v = b < a.nv ? (u64)a.planes[(i64)a.plane[b] * a.n + (i64)row] : 0ull;
        if (b < 8) hi |= v << (8 * (7 - b));
        else lo |= v << (8 * (15 - b));
      }
    }
    s_hi[i] = hi;
    s_lo[i] = lo;
    s_row[i] = row;
  }
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (P >> 1); t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), x = i | j;
        const bool up = (i & k) == 0;
        const u64 ah = s_hi[i], al = s_lo[i], bh = s_hi[x], bl = s_lo[x];
        const u32 ar = s_row[i], br = s_row[x];
        const bool gt = ah != bh ? ah > bh : al != bl ? al > bl : ar > br;       // (rows break ties: the order of equal keys is the input's)
        if (gt == up) {
          s_hi[i] = bh; s_lo[i] = bl; s_row[i] = br;
          s_hi[x] = ah; s_lo[x] = al; s_row[x] = ar;
        }
      }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < a.m; i += 1024) a.out[i] = s_row[i];
}

// ---- Utf8 take: out row k = source row idx[k].  A row is NULL (→ empty) when its byte in `ok_bytes` is 0 (per OUTPUT row, as
// written by an emit kernel) or its bit in `src_valid_bits` is 0 (per SOURCE row); either may be null.  Three steps:
// lengths → exclusive scan into the new offsets → byte copy.
__device__ __forceinline__ bool utf8_row_valid(const u8* ok_bytes, const u8* src_valid_bits, i64 k, u32 r) {
  if (ok_bytes && !ok_bytes[k]) return false;
  if (src_valid_bits && !((src_valid_bits[r >> 3] >> (r & 7)) & 1)) return false;
  return true;
}
__global__ __launch_bounds__(256) void take_utf8_lengths_kernel(const i32* offs, const u32* idx, const u8* ok_bytes, const u8* src_valid_bits, i64 n,
                                                                u32* lengths) {
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < n; k += (i64)gridDim.x * 256) {
    const u32 r = idx[k];
    lengths[k] = utf8_row_valid(ok_bytes, src_valid_bits, k, r) ? (u32)(offs[r + 1] - offs[r]) : 0u;
  }
}
// rows of a List column taken by index: output row k's elements are source elements offs[idx[k]] …; the element indices, one per output element
// (eight lanes per row, like the byte copy below) — the element column is then taken by them, whatever its type
__global__ __launch_bounds__(256) void take_list_indices_kernel(const i32* offs, const u32* idx, i64 n, const i32* out_offs, u32* elem_idx) {
  const int sub = threadIdx.x & 7;
  for (i64 k = ((i64)blockIdx.x * 256 + threadIdx.x) >> 3; k < n; k += ((i64)gridDim.x * 256) >> 3) {
    const i32 lo = out_offs[k], len = out_offs[k + 1] - lo;
    const i32 src = offs[idx[k]];
    for (i32 j = sub; j < len; j += 8) elem_idx[lo + j] = (u32)(src + j);
  }
}
// Explode (operators: UnnestExec in the reference, planner.rs:1949-2110): rows of a List column → one output row per element.  counts[r] = the
// row's elements — a NULL list has none — or ONE for an empty / NULL list under explode_outer (its element is NULL).
__global__ __launch_bounds__(256) void explode_counts_kernel(const i32* offs, const u8* valid_bits, i64 n, int outer, u32* counts) {
  for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < n; r += (i64)gridDim.x * 256) {
    const bool ok = !valid_bits || ((valid_bits[r >> 3] >> (r & 7)) & 1);
    const u32 len = ok ? (u32)(offs[r + 1] - offs[r]) : 0u;
    counts[r] = (len == 0 && outer) ? 1u : len;
  }
}
// per output row: its input row, its element (source index in the element column), whether there is one (0: the NULL row of explode_outer),
// its position in the list, and the element's validity byte (there is one AND its own validity bit says so)
__global__ __launch_bounds__(256) void explode_indices_kernel(const i32* offs, const u8* valid_bits, const u8* elem_valid_bits, i64 n, const i32* out_offs, u32* row_idx, u32* elem_idx,
                                                              u8* has_elem, i32* pos, u8* elem_ok) {
  const int sub = threadIdx.x & 7;
  for (i64 r = ((i64)blockIdx.x * 256 + threadIdx.x) >> 3; r < n; r += ((i64)gridDim.x * 256) >> 3) {
    const i32 lo = out_offs[r], cnt = out_offs[r + 1] - lo;
    const bool ok = !valid_bits || ((valid_bits[r >> 3] >> (r & 7)) & 1);
    const i32 src = offs[r], len = ok ? offs[r + 1] - src : 0;
    for (i32 j = sub; j < cnt; j += 8) {
      const bool real = j < len;
      const u32 e = real ? (u32)(src + j) : 0u;
      row_idx[lo + j] = (u32)r;
      elem_idx[lo + j] = e;
      has_elem[lo + j] = real;
      pos[lo + j] = real ? j : 0;
      elem_ok[lo + j] = real && (!elem_valid_bits || ((elem_valid_bits[e >> 3] >> (e & 7)) & 1));
    }
  }
}
// eight lanes copy one value: they write consecutive bytes, so a wave stores 8 contiguous runs instead of 64 scattered bytes per step
__global__ __launch_bounds__(256) void take_utf8_copy_kernel(const i32* offs, const u8* bytes, const u32* idx, const u8* ok_bytes, const u8* src_valid_bits,
                                                             i64 n, const i32* out_offs, u8* out_bytes) {
  const int sub = threadIdx.x & 7;
  for (i64 k = ((i64)blockIdx.x * 256 + threadIdx.x) >> 3; k < n; k += ((i64)gridDim.x * 256) >> 3) {
    const i32 lo = out_offs[k], len = out_offs[k + 1] - lo;
    if (len <= 0) continue;
    const u8* src = bytes + offs[idx[k]];
    u8* dst = out_bytes + lo;
    for (i32 b = sub; b < len; b += 8) dst[b] = src[b];
  }
}

// ---- string views (comet_device.hpp strview): a slice of a source value plus pad characters from a repeating pattern ----
struct StrView { u32 row, start, len, pad; };
struct PadPattern {          // ≤ 32 characters / 64 bytes; char_off[i] = byte offset of the first i characters of one repetition
  u32 nchars, nbytes;
  u8 bytes[64];
  u8 char_off[36];
};
__device__ __forceinline__ u32 pad_byte_count(const PadPattern& pp, u32 chars) {
  if (pp.nchars == 0) return 0;
  return (chars / pp.nchars) * pp.nbytes + pp.char_off[chars % pp.nchars];
}
__global__ __launch_bounds__(256) void strview_lengths_kernel(const StrView* v, const u8* ok_bytes, i64 n, PadPattern pp, u32* lengths) {
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < n; k += (i64)gridDim.x * 256)
    lengths[k] = (!ok_bytes || ok_bytes[k]) ? v[k].len + pad_byte_count(pp, v[k].pad) : 0u;
}
__global__ __launch_bounds__(256) void strview_copy_kernel(const StrView* v, const u8* ok_bytes, const i32* src_offs, const u8* src_bytes, i64 n, PadPattern pp,
                                                           int pad_left, const i32* out_offs, u8* out_bytes) {
  const int sub = threadIdx.x & 7;
  for (i64 k = ((i64)blockIdx.x * 256 + threadIdx.x) >> 3; k < n; k += ((i64)gridDim.x * 256) >> 3) {
    const i32 lo = out_offs[k], total = out_offs[k + 1] - lo;
    if (total <= 0) continue;
    const StrView sv = v[k];
    const i32 padb = total - (i32)sv.len;
    const u8* src = src_bytes + src_offs[sv.row] + sv.start;
    u8* dst = out_bytes + lo;
    const i32 val_at = pad_left ? padb : 0, pad_at = pad_left ? 0 : (i32)sv.len;
    for (i32 b = sub; b < (i32)sv.len; b += 8) dst[val_at + b] = src[b];
    for (i32 b = sub; b < padb; b += 8) dst[pad_at + b] = pp.bytes[(u32)b % pp.nbytes];
  }
}

// ---- upper / lower (OutCol::case_mode): Rust's str::to_uppercase / to_lowercase over a view's bytes (device/case_map.hpp) ----
#define CASE_TABLE_QUAL __device__ static const
#include "device/case_map.hpp"
__global__ __launch_bounds__(256) void strcase_lengths_kernel(const StrView* v, const u8* ok_bytes, const i32* src_offs, const u8* src_bytes, i64 n, int mode, u32* lengths) {
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < n; k += (i64)gridDim.x * 256) {
    u32 len = 0;
    if (!ok_bytes || ok_bytes[k]) {
      const StrView sv = v[k];
      len = (u32)case_map_value(src_bytes + src_offs[sv.row] + sv.start, (i32)sv.len, mode, nullptr);
    }
    lengths[k] = len;
  }
}
__global__ __launch_bounds__(256) void strcase_write_kernel(const StrView* v, const u8* ok_bytes, const i32* src_offs, const u8* src_bytes, i64 n, int mode, const i32* out_offs, u8* out_bytes) {
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < n; k += (i64)gridDim.x * 256) {
    if (ok_bytes && !ok_bytes[k]) continue;
    const StrView sv = v[k];
    (void)case_map_value(src_bytes + src_offs[sv.row] + sv.start, (i32)sv.len, mode, out_bytes + out_offs[k]);
  }
}

// ---- concat (OutCol::concat_cols): per output row the source row; the parts are Utf8 columns of that row and literals ----
struct ConcatArgs { i32 n; i32 lit_len[8]; const i32* offs[8]; const u8* bytes[8]; i64 first[8]; };
__global__ __launch_bounds__(256) void concat_lengths_kernel(ConcatArgs a, const u32* rows, const u8* ok_bytes, i64 n, u32* lengths) {
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < n; k += (i64)gridDim.x * 256) {
    u32 len = 0;
    if (!ok_bytes || ok_bytes[k]) {
      const i64 r = rows[k];
      for (int p = 0; p < a.n; p++) len += a.offs[p] ? (u32)(a.offs[p][a.first[p] + r + 1] - a.offs[p][a.first[p] + r]) : (u32)a.lit_len[p];
    }
    lengths[k] = len;
  }
}
__global__ __launch_bounds__(256) void concat_copy_kernel(ConcatArgs a, const u32* rows, const u8* ok_bytes, i64 n, const i32* out_offs, u8* out_bytes) {
  const int sub = threadIdx.x & 7;
  for (i64 k = ((i64)blockIdx.x * 256 + threadIdx.x) >> 3; k < n; k += ((i64)gridDim.x * 256) >> 3) {
    if (ok_bytes && !ok_bytes[k]) continue;
    const i64 r = rows[k];
    u8* dst = out_bytes + out_offs[k];
    for (int p = 0; p < a.n; p++) {
      const u8* src;
      i32 len;
      if (a.offs[p]) {
        const i32 lo = a.offs[p][a.first[p] + r];
        len = a.offs[p][a.first[p] + r + 1] - lo;
        src = a.bytes[p] + lo;
      } else {
        len = a.lit_len[p];
        src = a.bytes[p];
      }
      for (i32 b = sub; b < len; b += 8) dst[b] = src[b];
      dst += len;
    }
  }
}

#define RYU_TABLE_QUAL __device__ static const
#include "device/ryu.hpp"
// ---- formatted values (Cast … AS STRING, OutCol::fmt_kind): one i128 per row in, digits out (comet_device.hpp "values to strings") ----
__device__ __forceinline__ i32 strfmt_one(int kind, long long arg, i128 v, u8* o) {
  switch (kind) {
    case 1: return fmt_i64((i64)v, o);
    case 2: return fmt_bool(v != 0, o);
    case 3: return fmt_decimal(v, (int)arg, false, o);
    case 4: return fmt_decimal(v, (int)arg, true, o);
    case 5: return fmt_date((i64)v, o);
    case 7: return fmt_f64_bits((u64)v, o);
    case 8: return fmt_f32_bits((u32)v, o);
    default: return fmt_timestamp((i64)v, (i64)arg, o);
  }
}
__global__ __launch_bounds__(256) void strfmt_lengths_kernel(int kind, long long arg, const i128* v, const u8* ok_bytes, i64 n, u32* lengths) {
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < n; k += (i64)gridDim.x * 256) {
    u8 buf[48];
    lengths[k] = (!ok_bytes || ok_bytes[k]) ? (u32)strfmt_one(kind, arg, v[k], buf) : 0u;
  }
}
__global__ __launch_bounds__(256) void strfmt_write_kernel(int kind, long long arg, const i128* v, const u8* ok_bytes, i64 n, const i32* out_offs, u8* out_bytes) {
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < n; k += (i64)gridDim.x * 256) {
    if (ok_bytes && !ok_bytes[k]) continue;
    u8 buf[48];
    const i32 len = strfmt_one(kind, arg, v[k], buf);
    u8* dst = out_bytes + out_offs[k];
    for (i32 b = 0; b < len; b++) dst[b] = buf[b];
  }
}

// ---- constant columns (Hive partition values of a Parquet scan): dst[first .. first+n) = value
template <class T>
__global__ __launch_bounds__(256) void fill_kernel(T* dst, i64 n, T value) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) dst[i] = value;
}
// Utf8 constant: offsets[i] = base + i·len for i in [0, n]; bytes = the value repeated n times
__global__ __launch_bounds__(256) void fill_utf8_kernel(i32* offsets, u8* bytes, i64 n, i32 base, i32 len, const u8* value) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i <= n; i += (i64)gridDim.x * 256) {
    offsets[i] = base + (i32)i * len;
    if (i < n)
      for (i32 b = 0; b < len; b++) bytes[(i64)base + i * len + b] = value[b];
  }
}

__global__ __launch_bounds__(256) void take_valid_bytes_kernel(const u8* bits, const u32* idx, i64 n, u8* out) {
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < n; k += (i64)gridDim.x * 256) {
    const u32 r = idx[k];
    out[k] = (bits[r >> 3] >> (r & 7)) & 1;
  }
}

int grid_for(i64 n) {
  i64 g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 256 * 16 ? 256 * 16 : g));
}

__global__ __launch_bounds__(256) void range_pid_kernel(const u8* __restrict__ planes, i64 n, int W, const u8* __restrict__ bkeys, int B,
                                                        i32* __restrict__ pids) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    int lo = 0, hi = B;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      const u8* b = bkeys + (size_t)mid * (size_t)W;
      int c = 0;   // compare(bound, row)
      for (int p = 0; p < W; p++) {
        const u8 x = b[p], y = planes[(size_t)p * (size_t)n + (size_t)i];
        if (x != y) { c = x < y ? -1 : 1; break; }
      }
      if (c <= 0) lo = mid + 1;
      else hi = mid;
    }
    pids[i] = lo;
  }
}

}  // namespace

extern "C" {

int64_t comet_partition_tiles(int64_t n) { return n <= 0 ? 1 : (n + part_tile_rows(n) - 1) / part_tile_rows(n); }
// bytes of the `hist` scratch of comet_launch_partition_indices: P·W counters + their total, one flag word (at byte offset (P·W + 1)·8), and
// the chunk sums of the multi-block scan
int64_t comet_partition_scratch_bytes(int64_t n, int32_t P) {
  const int64_t len = (int64_t)P * comet_partition_tiles(n);
  return (len + 2 + (len + kScanChunk - 1) / kScanChunk + 2) * 8;
}

// hist: (P*W + 1) u64 scratch; bad: one zeroed u32; starts: P+1 i64; row_indices: n u32
int comet_launch_partition_indices(const int32_t* pids, int64_t n, int32_t P, uint64_t* hist, uint32_t* bad, int64_t* starts,
                                   uint32_t* row_indices, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const i64 W = comet_partition_tiles(n);
  const i64 block_tiles = (W + 3) / 4;
  const int grid = (int)(block_tiles > 256 * 16 ? 256 * 16 : block_tiles);
  const size_t lds = (size_t)4 * (size_t)P * sizeof(u32);
  hipLaunchKernelGGL(part_hist_kernel, grid, 256, lds, st, pids, (i64)n, P, W, (u64*)hist, bad);
  const i64 len = (i64)P * W, nb = (len + kScanChunk - 1) / kScanChunk;
  if (nb <= 8) {
    hipLaunchKernelGGL(part_scan_kernel, 1, 256, 0, st, (u64*)hist, len);
  } else {
    u64* sums = (u64*)hist + len + 2;   // behind the counters, their total and the flag word (comet_partition_scratch_bytes)
    hipLaunchKernelGGL(part_scan_sums_kernel, (int)nb, 256, 0, st, (const u64*)hist, len, sums);
    hipLaunchKernelGGL(part_scan_kernel, 1, 256, 0, st, sums, nb);
    hipLaunchKernelGGL(part_scan_apply_kernel, (int)nb, 256, 0, st, (u64*)hist, len, (const u64*)sums, nb);
  }
  hipLaunchKernelGGL(part_starts_kernel, 1, 256, 0, st, (const u64*)hist, P, W, (i64)n, (i64*)starts);
  if (n > 0) hipLaunchKernelGGL(part_index_kernel, grid, 256, lds, st, pids, (i64)n, P, W, (const u64*)hist, row_indices);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// out_bytes[k] = validity bit of row idx[k] as one byte (validity crosses the exchange one byte per row: partition boundaries are not
// byte aligned)
int comet_launch_take_valid_bytes(const uint8_t* valid_bits, const uint32_t* idx, int64_t n, uint8_t* out_bytes, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(take_valid_bytes_kernel, grid_for(n), 256, 0, (hipStream_t)stream, valid_bits, idx, (i64)n, out_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// width: bytes per value (1, 2, 4, 8, 16) or 0 for bit-packed data
int comet_launch_take(int width, const void* src, const uint32_t* idx, int64_t n, void* dst, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  switch (width) {
    case 0: hipLaunchKernelGGL(take_bits_kernel, grid_for((n + 7) / 8), 256, 0, st, (const u8*)src, idx, (i64)n, (u8*)dst); break;
    case 1: hipLaunchKernelGGL(take_kernel<u8>, grid_for(n), 256, 0, st, (const u8*)src, idx, (i64)n, (u8*)dst); break;
    case 2: hipLaunchKernelGGL(take_kernel<unsigned short>, grid_for(n), 256, 0, st, (const unsigned short*)src, idx, (i64)n, (unsigned short*)dst); break;
    case 4: hipLaunchKernelGGL(take_kernel<u32>, grid_for(n), 256, 0, st, (const u32*)src, idx, (i64)n, (u32*)dst); break;
    case 8: hipLaunchKernelGGL(take_kernel<u64>, grid_for(n), 256, 0, st, (const u64*)src, idx, (i64)n, (u64*)dst); break;
    case 16: hipLaunchKernelGGL(take_kernel<i128>, grid_for(n), 256, 0, st, (const i128*)src, idx, (i64)n, (i128*)dst); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int comet_launch_take_utf8_lengths(const int32_t* offs, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits, int64_t n,
                                   uint32_t* lengths, void* stream) {
  if (n > 0) hipLaunchKernelGGL(take_utf8_lengths_kernel, grid_for(n), 256, 0, (hipStream_t)stream, offs, idx, ok_bytes, src_valid_bits, (i64)n, lengths);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_explode_counts(const int32_t* offs, const uint8_t* valid_bits, int64_t n, int outer, uint32_t* counts, void* stream) {
  if (n > 0) hipLaunchKernelGGL(explode_counts_kernel, grid_for(n), 256, 0, (hipStream_t)stream, offs, valid_bits, (i64)n, outer, counts);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_explode_indices(const int32_t* offs, const uint8_t* valid_bits, const uint8_t* elem_valid_bits, int64_t n, const int32_t* out_offs, uint32_t* row_idx,
                                 uint32_t* elem_idx, uint8_t* has_elem, int32_t* pos, uint8_t* elem_ok, void* stream) {
  if (n > 0)
    hipLaunchKernelGGL(explode_indices_kernel, grid_for(n * 8), 256, 0, (hipStream_t)stream, offs, valid_bits, elem_valid_bits, (i64)n, out_offs, row_idx, elem_idx, has_elem, pos, elem_ok);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_take_list_indices(const int32_t* offs, const uint32_t* idx, int64_t n, const int32_t* out_offs, uint32_t* elem_idx, void* stream) {
  if (n > 0) hipLaunchKernelGGL(take_list_indices_kernel, grid_for(n * 8), 256, 0, (hipStream_t)stream, offs, idx, (i64)n, out_offs, elem_idx);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_take_utf8_copy(const int32_t* offs, const uint8_t* bytes, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits,
                                int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream) {
  if (n > 0)
    hipLaunchKernelGGL(take_utf8_copy_kernel, grid_for(n * 8), 256, 0, (hipStream_t)stream, offs, bytes, idx, ok_bytes, src_valid_bits, (i64)n, out_offs, out_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// pattern: the pad string's bytes (≤ 64 bytes, ≤ 32 characters; validated by the caller)
static PadPattern make_pattern(const uint8_t* pattern, int32_t nbytes) {
  PadPattern pp;
  memset(&pp, 0, sizeof pp);
  pp.nbytes = (u32)nbytes;
  u32 ch = 0;
  for (int32_t b = 0; b < nbytes; b++) {
    pp.bytes[b] = pattern[b];
    if ((pattern[b] & 0xC0) != 0x80) pp.char_off[ch++] = (u8)b;
  }
  pp.char_off[ch] = (u8)nbytes;
  pp.nchars = ch;
  return pp;
}
int comet_launch_strcase_lengths(const void* views, const uint8_t* ok_bytes, const int32_t* src_offs, const uint8_t* src_bytes, int64_t n, int mode, uint32_t* lengths, void* stream) {
  if (n > 0) hipLaunchKernelGGL(strcase_lengths_kernel, grid_for(n), 256, 0, (hipStream_t)stream, (const StrView*)views, ok_bytes, src_offs, src_bytes, (i64)n, mode, lengths);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_strcase_write(const void* views, const uint8_t* ok_bytes, const int32_t* src_offs, const uint8_t* src_bytes, int64_t n, int mode, const int32_t* out_offs, uint8_t* out_bytes,
                               void* stream) {
  if (n > 0) hipLaunchKernelGGL(strcase_write_kernel, grid_for(n), 256, 0, (hipStream_t)stream, (const StrView*)views, ok_bytes, src_offs, src_bytes, (i64)n, mode, out_offs, out_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_concat_lengths(const void* a, const uint32_t* rows, const uint8_t* ok_bytes, int64_t n, uint32_t* lengths, void* stream) {
  if (n > 0) hipLaunchKernelGGL(concat_lengths_kernel, grid_for(n), 256, 0, (hipStream_t)stream, *(const ConcatArgs*)a, rows, ok_bytes, (i64)n, lengths);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_concat_copy(const void* a, const uint32_t* rows, const uint8_t* ok_bytes, int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream) {
  if (n > 0) hipLaunchKernelGGL(concat_copy_kernel, grid_for(n * 8), 256, 0, (hipStream_t)stream, *(const ConcatArgs*)a, rows, ok_bytes, (i64)n, out_offs, out_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_strfmt_lengths(int kind, long long arg, const void* vals128, const uint8_t* ok_bytes, int64_t n, uint32_t* lengths, void* stream) {
  if (n > 0) hipLaunchKernelGGL(strfmt_lengths_kernel, grid_for(n), 256, 0, (hipStream_t)stream, kind, arg, (const i128*)vals128, ok_bytes, (i64)n, lengths);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_strfmt_write(int kind, long long arg, const void* vals128, const uint8_t* ok_bytes, int64_t n, const int32_t* out_offs, uint8_t* out_bytes, void* stream) {
  if (n > 0) hipLaunchKernelGGL(strfmt_write_kernel, grid_for(n), 256, 0, (hipStream_t)stream, kind, arg, (const i128*)vals128, ok_bytes, (i64)n, out_offs, out_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_strview_lengths(const void* views, const uint8_t* ok_bytes, int64_t n, const uint8_t* pattern, int32_t pattern_bytes, uint32_t* lengths, void* stream) {
  if (pattern_bytes < 0 || pattern_bytes > 64) return -1;
  const PadPattern pp = make_pattern(pattern, pattern_bytes);
  if (pp.nchars > 32) return -1;
  if (n > 0) hipLaunchKernelGGL(strview_lengths_kernel, grid_for(n), 256, 0, (hipStream_t)stream, (const StrView*)views, ok_bytes, (i64)n, pp, lengths);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_strview_copy(const void* views, const uint8_t* ok_bytes, const int32_t* src_offs, const uint8_t* src_bytes, int64_t n, const uint8_t* pattern,
                              int32_t pattern_bytes, int pad_left, const int32_t* out_offs, uint8_t* out_bytes, void* stream) {
  if (pattern_bytes < 0 || pattern_bytes > 64) return -1;
  const PadPattern pp = make_pattern(pattern, pattern_bytes);
  if (n > 0)
    hipLaunchKernelGGL(strview_copy_kernel, grid_for(n * 8), 256, 0, (hipStream_t)stream, (const StrView*)views, ok_bytes, src_offs, src_bytes, (i64)n, pp, pad_left,
                       out_offs, out_bytes);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// width ∈ {1,2,4,8,16}: value points to `width` bytes on the HOST
int comet_launch_fill(int width, void* dst, int64_t n, const void* value, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  switch (width) {
    case 1: hipLaunchKernelGGL(fill_kernel<u8>, grid_for(n), 256, 0, st, (u8*)dst, (i64)n, *(const u8*)value); break;
    case 2: hipLaunchKernelGGL(fill_kernel<unsigned short>, grid_for(n), 256, 0, st, (unsigned short*)dst, (i64)n, *(const unsigned short*)value); break;
    case 4: hipLaunchKernelGGL(fill_kernel<u32>, grid_for(n), 256, 0, st, (u32*)dst, (i64)n, *(const u32*)value); break;
    case 8: hipLaunchKernelGGL(fill_kernel<u64>, grid_for(n), 256, 0, st, (u64*)dst, (i64)n, *(const u64*)value); break;
    case 16: { i128 v; memcpy(&v, value, 16); hipLaunchKernelGGL(fill_kernel<i128>, grid_for(n), 256, 0, st, (i128*)dst, (i64)n, v); break; }
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_fill_utf8(int32_t* offsets, uint8_t* bytes, int64_t n, int32_t base, int32_t len, const uint8_t* dev_value, void* stream) {
  hipLaunchKernelGGL(fill_utf8_kernel, grid_for(n + 1), 256, 0, (hipStream_t)stream, offsets, bytes, (i64)n, base, len, dev_value);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_sort_iota(uint32_t* perm, int64_t n, uint32_t first, void* stream) {
  if (n > 0) hipLaunchKernelGGL(sort_iota_kernel, grid_for(n), 256, 0, (hipStream_t)stream, perm, (i64)n, first);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_sort_gather_digit(const uint8_t* plane, const uint32_t* perm, int64_t n, int32_t* digit, void* stream) {
  if (n > 0) hipLaunchKernelGGL(sort_gather_digit_kernel, grid_for(n), 256, 0, (hipStream_t)stream, plane, perm, (i64)n, digit);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_sort_hist256(const uint8_t* plane, const uint32_t* cand, int64_t m, uint64_t* hist, void* stream) {
  if (m > 0) hipLaunchKernelGGL(sort_hist256_kernel, grid_for(m), 256, 0, (hipStream_t)stream, plane, cand, (i64)m, (unsigned long long*)hist);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_sort_select(const uint8_t* plane, const uint32_t* cand, int64_t m, int dstar, uint32_t* sure, uint32_t* next_cand, uint32_t* counters,
                             void* stream) {
  if (m > 0) hipLaunchKernelGGL(sort_select_kernel, grid_for(m), 256, 0, (hipStream_t)stream, plane, cand, (i64)m, dstar, sure, next_cand, counters);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// Range partitioning (multi_partition.rs:332-366): partition id = number of boundary rows ≤ the row under the sort order, i.e. an
// upper-bound binary search over the boundaries' order-preserving key bytes.  planes: W byte planes of n rows (k_sortkey layout),
// bkeys: B boundary keys of W bytes each, row-major, ascending.
int comet_launch_range_partition_ids(const uint8_t* planes, int64_t n, int W, const uint8_t* bkeys, int B, int32_t* pids, void* stream) {
  if (n > 0) hipLaunchKernelGGL(range_pid_kernel, grid_for(n), 256, 0, (hipStream_t)stream, planes, (i64)n, W, bkeys, B, pids);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// cand[0 … m) (m ≤ 4096) sorted by the nv ≤ 16 key planes listed (most significant first) → out
int comet_launch_sort_small(const uint8_t* planes, int64_t n, const uint32_t* cand, int m, const int* plane_idx, int nv, uint32_t* out, void* stream) {
  if (m <= 0) return 0;
  if (m > 4096 || nv > 16) return -1;
  SortSmallArgs a;
  a.planes = planes; a.n = (i64)n; a.cand = cand; a.out = out; a.m = m; a.nv = nv;
  for (int b = 0; b < 16; b++) a.plane[b] = b < nv ? plane_idx[b] : 0;
  hipLaunchKernelGGL(sort_small_kernel, 1, 1024, 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int comet_launch_sort_plane_varies(const uint8_t* planes, int64_t n, int W, uint32_t* flags, void* stream) {
  if (n > 0 && W > 0) {
    i64 g = (n + 255) / 256;
    dim3 grid((unsigned)(g > 512 ? 512 : g), (unsigned)W);
    hipLaunchKernelGGL(sort_plane_varies_kernel, grid, 256, 0, (hipStream_t)stream, planes, (i64)n, W, flags);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // extern "C"
