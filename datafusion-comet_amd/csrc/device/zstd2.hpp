// Zstandard decompression of Parquet pages on the GPU (round 3) — the format of RFC 8878, restated for many lanes.
//
// A zstd frame is a list of blocks of ≤ 128 KiB; a compressed block is (a) its literals, Huffman-coded in one or four backward
// bitstreams, (b) its sequences — (literal length, match length, offset) triples, three interleaved FSE states over one backward
// bitstream — and (c) the execution: copy `ll` literals, then `ml` bytes from `offset` back, for every sequence.  (a) and (b) are serial
// per stream but independent across streams and blocks: a page of 1 MiB holds 8 blocks → 32 Huffman streams + 8 sequence streams, a
// scan holds hundreds of pages.  (c) is the copy machinery of the snappy pipeline (device/snappy2.hpp): pointer jumping over the
// 65 536 two-byte pointers of a 64 KiB output fragment in workgroup memory — except that zstd matches reach back over the whole page,
// so a page's fragments are resolved in order by ONE workgroup, and a match byte whose source lies in an earlier fragment is fetched
// from the (final) output directly.
//
//   kernel A1 one wave per block: the Huffman table, then the literal streams a lane each, 256 symbols a round, into the page's literal
//             scratch;   kernel A2 one wave per block: the three FSE tables, then one lane decodes the sequences, 64 a round, into
//             records (ll, ml, offset).  The decoding lanes read their bitstream from a window in workgroup memory and write into a
//             buffer there; between rounds the whole wave slides the window and writes the buffer out — a serial lane never waits on
//             global memory (a first version that loaded and stored from the lane itself ran 30 times slower: on this hardware a load
//             returns behind every store issued before it).  Repeat offsets that reach back over the block's start stay SYMBOLIC
//             (index into the block's initial history, minus a delta), so blocks do not wait for each other.
//   kernel B  one lane per page: block output positions, the repeat-offset history handed from block to block, total length.
//   kernel C  one workgroup per block: prefix sums over its records (output position, literal position), symbolic offsets resolved
//             and checked.
//   kernel D  one workgroup per page: fragments in order — scatter (literals and far matches to the output, near matches as
//             pointers), pointer jumping, resolve.
//
// The HOST walks the frame and block headers only (sizes, modes, where the table descriptions and bitstreams sit: host `scan_page`);
// it never touches an entropy-coded byte.  A page the walk does not like — dictionary id, several frames, skippable frames, a window
// larger than the page — is not sent here (the scan inflates it on the host as before).
//
// Like snappy2.hpp every kernel is a sequence of PHASES (plain functions of the thread index that talk through workgroup memory between
// barriers, no wave intrinsics), so tests/emu/zstd2_emu.cpp runs the same source on the host with the threads looped.
#pragma once
#include <stdint.h>

#include "snappy2.hpp"

#define ZS_FN SN2_FN
#define ZS_LDS SN2_LDS

namespace comet_zstd2 {

using comet_snappy2::i32;
using comet_snappy2::i64;
using comet_snappy2::u16;
using comet_snappy2::u32;
using comet_snappy2::u64;
using comet_snappy2::u8;
typedef int16_t i16;

constexpr u32 kBlockMax = 131072;

constexpr int kScanThreads = 256;               // kernel C
constexpr int kExecThreads = comet_snappy2::kExecThreads;
constexpr u32 kFrag = (u32)comet_snappy2::kFrag;
constexpr u32 kBigPart = 256;                   // literal runs / matches from this length on are copied by the whole workgroup
constexpr u32 kBigQueue = 640;
constexpr int kHufLogMax = 11;

enum { ST_OK = 0, ST_ERR_HEADER = 16, ST_ERR_HUF = 17, ST_ERR_FSE = 18, ST_ERR_BITS = 19, ST_ERR_SEQ = 20, ST_ERR_OFFSET = 21, ST_ERR_LENGTH = 22 };
enum { BT_RAW = 0, BT_RLE = 1, BT_COMPRESSED = 2 };
enum { LT_RAW = 0, LT_RLE = 1, LT_HUF = 2 };   // (treeless = LT_HUF with an inherited tree description)
enum { TM_PREDEF = 0, TM_RLE = 1, TM_FSE = 2 }; // (repeat = whatever the earlier block had, resolved by the host walk)

struct ZPage {
  i64 src_off, dst_off;       // offsets into the column's byte buffer
  i32 src_len, dst_len;
  i32 block_first, nblocks;   // its blocks in the global block array
  i64 rec_first;              // its records in the global record array
  i64 lit_first;              // its literal bytes in the literal scratch
  u32 nrecs, pad;
};
struct ZBlock {
  // filled by the host walk (offsets are page relative)
  u32 pos, size;              // content: raw bytes / the RLE byte (size = run length) / the compressed block
  u8 type, lit_type, lit_streams, pad0;
  u8 tab_mode[3], pad1;       // LL, OF, ML
  u32 lit_regen;              // literal bytes of the block (raw / RLE blocks: their `size`, the whole block is one literal run)
  u32 lit_pos, lit_len;       // raw: the bytes; RLE: the byte; Huffman: jump table + streams
  u32 huf_desc, huf_desc_len; // the tree description to use (its own or an earlier block's)
  u32 nseq;
  u32 tab_desc[3];            // FSE: normalized counts; RLE: the symbol
  u32 bits_pos, bits_len;     // the sequence bitstream
  u32 rec_first;              // page-relative index of its first record (nseq + 1 records: the last holds the trailing literals)
  u32 lit_first;              // page-relative offset of its literals in the scratch
  // filled by the device
  u32 out_size;               // kernel A
  i32 rep_out[3];             // kernel A: history after the block (> 0: an offset; ≤ 0: symbolic, see rep_symbolic)
  u32 out_base;               // kernel B
  i32 rep_in[3];              // kernel B
};
struct ZRec { u32 out_pos, lit_pos, ll, ml; i32 off; };   // kernel A: ll, ml, off; kernel C: out_pos (page relative), lit_pos (page relative), off resolved

ZS_FN int highbit(u32 v) { return 31 - __builtin_clz(v); }

// ---- symbolic repeat offsets: "entry idx of the history at the block's start, minus delta" = −(1 + idx + 3·delta) ----
ZS_FN i32 rep_symbolic(int idx) { return -(1 + idx); }
ZS_FN i32 rep_minus_one(i32 v) { return v > 0 ? v - 1 : v - 3; }                 // (an offset of 0 is caught when it is resolved / checked)
ZS_FN i32 rep_resolve(i32 v, const i32* init) {
  if (v > 0) return v;
  const i32 t = -v - 1;
  return init[t % 3] - t / 3;
}

// ---- forward (little-endian) bit reader: table descriptions.  P: a byte pointer (global memory, workgroup memory or host memory) ----
template <class P>
struct FwdBits {
  P p;
  u32 len, bit;
  bool over;
  ZS_FN void init(P q, u32 n) { p = q; len = n; bit = 0; over = false; }
  ZS_FN u32 peek(int n) const {            // n ≤ 16
    const u32 b = bit >> 3;
    u32 v = 0;
    for (u32 k = 0; k < 4; k++) v |= (b + k < len ? (u32)p[b + k] : 0u) << (8 * k);
    return (v >> (bit & 7)) & ((1u << n) - 1u);
  }
  ZS_FN void skip(int n) { bit += (u32)n; if (bit > 8 * len) over = true; }
  ZS_FN u32 read(int n) { const u32 v = peek(n); skip(n); return v; }
  ZS_FN u32 bytes() const { return (bit + 7) >> 3; }
};

// normalized counts of an FSE table description → norm[0 … nsym), accuracy log; returns the bytes consumed, 0 = malformed
template <class P, class NormPtr>
ZS_FN u32 fse_read_ncount(P p, u32 avail, int max_sym /* inclusive */, int max_log, NormPtr norm, int& nsym, int& log) {
  FwdBits<P> in;
  in.init(p, avail);
  log = 5 + (int)in.read(4);
  if (log > max_log) return 0;
  i32 remaining = 1 << log;
  int s = 0;
  while (remaining > 0 && s <= max_sym) {
    const int bits = highbit((u32)remaining + 1u) + 1;
    u32 val = in.peek(bits);
    const u32 lower = (1u << (bits - 1)) - 1u;
    const u32 thresh = (1u << bits) - 1u - ((u32)remaining + 1u);
    if ((val & lower) < thresh) {
      in.skip(bits - 1);
      val &= lower;
    } else {
      in.skip(bits);
      if (val > lower) val -= thresh;
    }
    const i32 proba = (i32)val - 1;
    remaining -= proba < 0 ? -proba : proba;
    norm[s++] = (i16)proba;
    if (proba == 0) {
      u32 rep = in.read(2);
      for (;;) {
        for (u32 k = 0; k < rep && s <= max_sym; k++) norm[s++] = 0;
        if (rep == 3) rep = in.read(2);
        else break;
      }
    }
    if (in.over) return 0;
  }
  if (remaining != 0 || in.over) return 0;
  nsym = s;
  return in.bytes();
}
// decoding table: entry = symbol | bits to read << 8 | base of the next state << 16.  `next` is scratch of nsym entries.
template <class NormPtr, class TabPtr, class NextPtr>
ZS_FN bool fse_build(NormPtr norm, int nsym, int log, TabPtr tab, NextPtr next) {
  const u32 size = 1u << log;
  u32 high = size;
  for (int s = 0; s < nsym; s++)
    if (norm[s] == -1) { tab[--high] = (u32)s; next[s] = 1; }
  const u32 step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
  u32 pos = 0;
  for (int s = 0; s < nsym; s++) {
    const int n = norm[s];
    if (n <= 0) continue;
    next[s] = (u16)n;
    for (int i = 0; i < n; i++) {
      tab[pos] = (u32)s;
      do pos = (pos + step) & mask; while (pos >= high);
    }
  }
  if (pos != 0) return false;
  for (u32 i = 0; i < size; i++) {
    const u32 s = tab[i];
    const u32 ns = next[s]++;
    const u32 nb = (u32)log - (u32)highbit(ns);
    tab[i] = s | (nb << 8) | (((ns << nb) - size) << 16);
  }
  return true;
}
template <class TabPtr>
ZS_FN void fse_build_rle(TabPtr tab, u32 sym) { tab[0] = sym; }        // log 0: one state, no bits

// ---- backward bit reader: Huffman and FSE streams.  The stream's last byte holds a marker bit above its last data bit; bits are taken
// from there towards the stream's first byte.  Three 8-byte registers cover 24 bytes below the cursor: a load is issued a register's worth
// of decoding before its bytes are needed.  Src: where the stream's bytes come from —
//   PtrBytes   a pointer, bounds checked byte by byte (the host's look at a page's first bytes; the few dozen bytes of FSE-coded weights)
//   RingBytes  a window of the stream in workgroup memory that the workgroup slides down between rounds of decoding (the device's streams:
//              a lane that decodes serially must not wait on global memory — a load there returns behind every store issued before it) ----
template <class P>
struct PtrBytes {
  P p;
  i32 len;
  ZS_FN u64 load8(i32 pos) const {           // bytes [pos, pos + 8) of the stream; outside it: zero
    u64 v = 0;
    for (int k = 0; k < 8; k++) {
      const i32 q = pos + k;
      if (q >= 0 && q < len) v |= (u64)p[q] << (8 * k);
    }
    return v;
  }
};
constexpr u32 kRing = 2048;                  // bytes of a stream window (power of two)
// The reader only ever loads at positions ≡ the stream's length (mod 8) — its registers step down 8 bytes at a time from the stream's end —
// so the window is laid out with a BIAS: the byte at stream position q sits at ring byte (q + bias) mod kRing, bias = (−length) mod 8, and
// every load is one aligned 8-byte read.
ZS_FN u32 ring_bias(u32 len) { return (8u - (len & 7u)) & 7u; }
template <class WP, bool kUnused>
struct RingBytes {
  WP w;                                      // kRing / 4 words
  u32 bias;
  ZS_FN u64 load8(i32 pos) const {
    const u32 i = (((u32)pos + bias) & (kRing - 1)) >> 2;
    const u32 w0 = w[i], w1 = w[i + 1];
    return (u64)w0 | ((u64)w1 << 32);
  }
};
// the window slides: stream positions [from, to) are loaded, a word per thread and step; from + bias and to + bias are multiples of 4.
// Positions outside [lim_lo, lim_hi) — before the page's first byte or behind its padded end — read as zero.
template <class WP>
ZS_FN void ring_fill(WP w, u32 bias, const u8* stream, i32 from, i32 to, i32 lim_lo, i32 lim_hi, int t, int nthreads) {
  for (i32 q = from + 4 * t; q < to; q += 4 * nthreads) {
    u32 v = 0;
    if (q >= lim_lo && q + 4 <= lim_hi) __builtin_memcpy(&v, stream + q, 4);
    else
      for (int k = 0; k < 4; k++)
        if (q + k >= lim_lo && q + k < lim_hi) v |= (u32)stream[q + k] << (8 * k);
    w[(((u32)q + bias) & (kRing - 1)) >> 2] = v;
  }
}
// how far down the window must reach when decoding resumes at that byte: 1 KiB below it (a word boundary of the biased layout)
ZS_FN i32 ring_low_for(i32 cursor_byte, u32 bias) {
  const i32 v = ((cursor_byte + (i32)bias - 1024) & ~3) - (i32)bias;
  const i32 floor_ = -40 - (((i32)bias - 40) & 3);          // nothing below the stream's first byte is ever consumed; the registers reach 31 bytes below it
  return v < floor_ ? floor_ : v;
}
ZS_FN i32 ring_top_for(u32 len, u32 bias) { return (i32)(((len + bias + 8 + 3) & ~3u) - bias); }      // the first fill's upper edge: the stream's end + 8, a word boundary

template <class Src>
struct BackBitsT {
  Src src;
  i32 bitpos;          // bits left: the next read takes bits [bitpos − n, bitpos) of the stream
  i32 wb;              // hi = bytes [wb − 8, wb), lo = [wb − 16, wb − 8), nx = [wb − 24, wb − 16)
  u64 hi, lo, nx;
  ZS_FN bool init(const Src& s, u32 len) {
    src = s;
    bitpos = 0;
    wb = (i32)len;
    hi = lo = nx = 0;
    if (len == 0) return false;
    hi = src.load8(wb - 8);
    const u32 last = (u32)(hi >> 56);
    if (last == 0) return false;
    bitpos = 8 * ((i32)len - 1) + highbit(last);
    lo = src.load8(wb - 16);
    nx = src.load8(wb - 24);
    return true;
  }
  ZS_FN u32 peek(int n) const {            // n ≤ 32; bits beyond the stream's beginning read as 0
    if (n == 0 || bitpos <= 0) return 0;
    const int m = bitpos < n ? bitpos : n;                  // (not recursive: a recursive function is not inlined, and a call spills the reader)
    const i32 sft = bitpos - 8 * (wb - 16) - m;             // lowest wanted bit within (hi:lo); the cursor stays inside hi: 32 < sft < 128
    const u64 v = sft >= 64 ? hi >> (sft - 64) : (hi << (64 - sft)) | (lo >> sft);
    return ((u32)v & (m >= 32 ? 0xffffffffu : ((1u << m) - 1u))) << (n - m);
  }
  ZS_FN void consume(int n) {              // n ≤ 64
    bitpos -= n;
    if (bitpos - 8 * (wb - 16) <= 64) {
      hi = lo;
      lo = nx;
      wb -= 8;
      nx = src.load8(wb - 24);
      if (bitpos - 8 * (wb - 16) <= 64) {    // (more than 32 bits at once can step over a whole register)
        hi = lo;
        lo = nx;
        wb -= 8;
        nx = src.load8(wb - 24);
      }
    }
  }
  // The 64 bits below the cursor, the next bit on top; fields are then taken off its top.  No branch, and no condition on where the
  // cursor is: the registers step down with it below the stream's first byte too (there the source yields bytes nobody may consume — a
  // stream that does is caught by its final bit count).
  ZS_FN u64 window() const {
    const u32 sft = (u32)(bitpos - 8 * (wb - 16) - 64);     // 0 < sft ≤ 64 (the register invariant)
    return (hi << ((64u - sft) & 63u)) | ((lo >> (sft - 1u)) >> 1);
  }
  // consume without a branch (n ≤ 64: one register step at most): the candidate for nx is loaded whether or not the registers move —
  // lanes of one wave decode different streams, and a branch that some take costs all of them both paths
  ZS_FN void consume_sel(u32 n) {
    bitpos -= (i32)n;
    const bool step = bitpos - 8 * (wb - 16) <= 64;
    const u64 cand = src.load8(wb - 32);
    hi = step ? lo : hi;
    lo = step ? nx : lo;
    nx = step ? cand : nx;
    wb -= step ? 8 : 0;
  }
  ZS_FN static u32 take(u64& w, u32 n) {     // n ≤ 32 (n = 0 → 0): the top word shifted up by n, what crosses into the next word
    const u32 v = (u32)(((u64)(u32)(w >> 32) << n) >> 32);
    w <<= n;
    return v;
  }
  ZS_FN u32 read(int n) { const u32 v = peek(n); consume(n); return v; }
  ZS_FN i32 cursor_byte() const { return bitpos > 0 ? (bitpos + 7) >> 3 : 0; }
};

// ---- Huffman ----
// Tree description at p: weights of symbols 0 … nw − 1 into w[] (the last symbol's weight is implied), → bytes consumed, 0 = malformed.
// wtab: scratch for the weights' own FSE table (64 entries), norm / next: scratch of ≥ 16 entries.
template <class P, class WPtr, class TabPtr, class NormPtr, class NextPtr>
ZS_FN u32 huf_read_weights(P p, u32 avail, WPtr w, int& nw, TabPtr wtab, NormPtr norm, NextPtr next) {
  if (avail < 1) return 0;
  const u32 hb = p[0];
  if (hb >= 128) {
    nw = (int)hb - 127;
    const u32 bytes = ((u32)nw + 1) / 2;
    if (1 + bytes > avail) return 0;
    for (int i = 0; i < nw; i++) w[i] = (i & 1) ? (u8)(p[1 + i / 2] & 15) : (u8)(p[1 + i / 2] >> 4);
    return 1 + bytes;
  }
  if (hb == 0 || 1 + hb > avail) return 0;
  int nsym = 0, log = 0;
  const u32 used = fse_read_ncount(p + 1, hb, 12, 6, norm, nsym, log);
  if (used == 0 || used >= hb) return 0;
  if (!fse_build(norm, nsym, log, wtab, next)) return 0;
  BackBitsT<PtrBytes<P>> b;
  PtrBytes<P> src;
  src.p = p + 1 + used;
  src.len = (i32)(hb - used);
  if (!b.init(src, hb - used)) return 0;
  u32 s1 = b.read(log), s2 = b.read(log);
  int n = 0;
  for (;;) {
    if (n >= 254) return 0;
    const u32 e1 = wtab[s1];
    w[n++] = (u8)(e1 & 0xff);
    s1 = (e1 >> 16) + b.read((int)((e1 >> 8) & 0xff));
    if (b.bitpos < 0) { w[n++] = (u8)(wtab[s2] & 0xff); break; }
    const u32 e2 = wtab[s2];
    w[n++] = (u8)(e2 & 0xff);
    s2 = (e2 >> 16) + b.read((int)((e2 >> 8) & 0xff));
    if (b.bitpos < 0) { w[n++] = (u8)(wtab[s1] & 0xff); break; }
  }
  nw = n;
  return 1 + hb;
}
// decoding table of 2^log entries (symbol | code length << 8) from the weights; → log, 0 = malformed
template <class WPtr, class TabPtr>
ZS_FN int huf_build(WPtr w, int nw, TabPtr tab) {
  if (nw < 1 || nw > 255) return 0;
  u32 sum = 0;
  for (int i = 0; i < nw; i++) {
    if (w[i] > kHufLogMax) return 0;
    sum += w[i] ? 1u << (w[i] - 1) : 0u;
  }
  if (sum == 0) return 0;
  const int log = highbit(sum) + 1;
  if (log > kHufLogMax) return 0;
  const u32 left = (1u << log) - sum;
  if (left & (left - 1)) return 0;                          // the implied weight must make a power of two
  w[nw] = (u8)(highbit(left) + 1);
  const int nsym = nw + 1;
  // entries by ascending weight (the longest codes first), within a weight by symbol
  u32 count[kHufLogMax + 2];
  for (int k = 0; k <= kHufLogMax + 1; k++) count[k] = 0;
  for (int i = 0; i < nsym; i++) count[w[i]]++;
  u32 start[kHufLogMax + 2];
  u32 pos = 0;
  for (int k = 1; k <= log; k++) { start[k] = pos; pos += count[k] << (k - 1); }
  if (pos != (1u << log)) return 0;
  for (int i = 0; i < nsym; i++) {
    const u32 wt = w[i];
    if (!wt) continue;
    const u32 len = 1u << (wt - 1), e = (u32)i | (((u32)log + 1u - wt) << 8);
    u32 a = start[wt];
    start[wt] = a + len;
    for (u32 k = 0; k < len; k++) tab[a + k] = (u16)e;
  }
  return log;
}
// `count` symbols from the reader into out[]
template <class BB, class TabPtr, class OutPtr>
ZS_FN void huf_decode_symbols(BB& b, TabPtr tab, int log, OutPtr out, u32 count) {
  u32 i = 0;
  // five codes (≤ 11 bits each) off one 64-bit window, one consume for the five
  for (; i + 5 <= count; i += 5) {
    u64 w = b.window();
    u32 used = 0;
    for (int k = 0; k < 5; k++) {
      const u32 e = tab[(u32)(w >> (64 - log))];
      const u32 n = e >> 8;
      w <<= n;
      used += n;
      out[i + k] = (u8)e;
    }
    b.consume_sel(used);
  }
  for (; i < count; i++) {
    const u32 e = tab[b.peek(log)];
    b.consume((int)(e >> 8));
    out[i] = (u8)e;
  }
}

// ---- sequence codes: value = base + extra bits; entry = base | bits << 24 ----
ZS_FN u32 ll_code_entry(u32 c) {
  const u32 base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40, 48, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536};
  const u32 bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  return base[c] | (bits[c] << 24);
}
ZS_FN u32 ml_code_entry(u32 c) {
  const u32 base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
                        35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 131, 259, 515, 1027, 2051, 4099, 8195, 16387, 32771, 65539};
  const u32 bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                        1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
  return base[c] | (bits[c] << 24);
}
// predefined distributions (RFC 8878 §3.1.1.3.2.2): kind 0 LL (log 6), 1 OF (log 5), 2 ML (log 6)
template <class NormPtr>
ZS_FN void predefined_norm(int kind, NormPtr norm, int& nsym, int& log) {
  const i16 ll[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
  const i16 of[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
  const i16 ml[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                      1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
  if (kind == 0) { nsym = 36; log = 6; for (int i = 0; i < 36; i++) norm[i] = ll[i]; }
  else if (kind == 1) { nsym = 29; log = 5; for (int i = 0; i < 29; i++) norm[i] = of[i]; }
  else { nsym = 53; log = 6; for (int i = 0; i < 53; i++) norm[i] = ml[i]; }
}
constexpr int kMaxSym[3] = {35, 31, 52};
constexpr int kMaxLog[3] = {9, 8, 9};
// one of a block's three tables; → its accuracy log, −1 = malformed
template <class P, class TabPtr, class NormPtr, class NextPtr>
ZS_FN int seq_table(int kind, int mode, P desc, u32 avail, TabPtr tab, NormPtr norm, NextPtr next) {
  int nsym = 0, log = 0;
  if (mode == TM_RLE) {
    if (avail < 1 || desc[0] > kMaxSym[kind]) return -1;
    fse_build_rle(tab, desc[0]);
    return 0;
  }
  if (mode == TM_PREDEF) predefined_norm(kind, norm, nsym, log);
  else if (fse_read_ncount(desc, avail, kMaxSym[kind], kMaxLog[kind], norm, nsym, log) == 0) return -1;
  return fse_build(norm, nsym, log, tab, next) ? log : -1;
}

// A sequence table as the decoder walks it: two words per state — [0] the value's base (literal length / match length / offset value
// 1 << code), [1] bits of the next state | extra bits of the value << 8 | base of the next state << 16.  Expanded in place from
// fse_build's one word per state (tab holds 2 << log words; from the top down, so no entry is overwritten before it is read).
// (llc / mlc: ll_code_entry / ml_code_entry of every code, in fast memory — the constant arrays behind those functions sit in global memory)
template <class TabPtr, class CodePtr>
ZS_FN void seq_table_expand(int kind, int log, TabPtr tab, CodePtr llc, CodePtr mlc) {
  for (i32 i = (i32)(1u << log) - 1; i >= 0; i--) {
    const u32 e = tab[i], sym = e & 0xffu;
    u32 base, extra;
    if (kind == 1) { base = 1u << sym; extra = sym; }
    else { const u32 ce = kind == 0 ? llc[sym] : mlc[sym]; base = ce & 0xffffffu; extra = ce >> 24; }
    tab[2 * i] = base;
    tab[2 * i + 1] = ((e >> 8) & 0xffu) | (extra << 8) | ((e >> 16) << 16);
  }
}

// ---- the sequences of one block: the decoder's state between sequences, and one step ----
struct SeqCore {
  u32 sll, sof, sml;          // the three FSE states
  i32 r0, r1, r2;             // repeat-offset history (symbolic until an offset of this block replaces an entry)
  u64 sum_ll, sum_ml;
  u32 done;                   // sequences decoded so far
};
template <class BB>
ZS_FN void seq_begin(BB& b, SeqCore& c, int lll, int lof, int lml) {
  c.sll = b.read(lll);
  c.sof = b.read(lof);
  c.sml = b.read(lml);
  c.r0 = rep_symbolic(0);
  c.r1 = rep_symbolic(1);
  c.r2 = rep_symbolic(2);
  c.sum_ll = c.sum_ml = 0;
  c.done = 0;
}
// the next sequence (`last`: no state update behind it); → ST_OK or an error code.  Tables: seq_table_expand's two words per state.
// Straight-line code: the three values' extra bits (≤ 31 + 16 + 16) come off one 64-bit window, the three states' bits (≤ 9 + 9 + 8) off a
// second, the repeat-offset rules are selects.  (Several lanes of a wave decode different blocks: every branch one lane takes is paid by all.)
template <class BB, class TabPtr>
ZS_FN u32 seq_step(BB& b, SeqCore& c, TabPtr tll, TabPtr tof, TabPtr tml, bool last, u32& ll, u32& ml, i32& off) {
  const u32 lb = tll[2 * c.sll], lh = tll[2 * c.sll + 1];
  const u32 ob = tof[2 * c.sof], oh = tof[2 * c.sof + 1];
  const u32 mb = tml[2 * c.sml], mh = tml[2 * c.sml + 1];
  const u32 oe = (oh >> 8) & 0xffu, me = (mh >> 8) & 0xffu, le = (lh >> 8) & 0xffu;
  const u32 ln = lh & 0xffu, mn = mh & 0xffu, on = oh & 0xffu;
  u64 w = b.window();
  const u32 ov = ob + BB::take(w, oe);
  ml = mb + BB::take(w, me);
  ll = lb + BB::take(w, le);
  b.consume_sel(oe + me + le);
  w = b.window();
  c.sll = (lh >> 16) + BB::take(w, ln);                    // (behind the last sequence: never looked at, and nothing is consumed for them)
  c.sml = (mh >> 16) + BB::take(w, mn);
  c.sof = (oh >> 16) + BB::take(w, on);
  b.consume_sel(last ? 0u : ln + mn + on);
  // offset value > 3: a new offset, pushed onto the history; 1 … 3: an entry of the history (shifted by one when there are no literals,
  // the fourth choice then being "the newest entry minus one"), moved to the front
  const bool rep = ov <= 3u;
  const u32 idx = ov - 1u + (ll == 0 ? 1u : 0u);
  const i32 picked = idx == 0 ? c.r0 : idx == 1 ? c.r1 : idx == 2 ? c.r2 : rep_minus_one(c.r0);
  off = rep ? picked : (i32)(ov - 3u);
  const bool keep = rep && idx == 0;
  const i32 n2 = (!rep || idx > 1) ? c.r1 : c.r2;
  c.r2 = keep ? c.r2 : n2;
  c.r1 = keep ? c.r1 : c.r0;
  c.r0 = keep ? c.r0 : off;
  c.sum_ll += ll;
  c.sum_ml += ml;
  c.done++;
  return (!rep && off <= 0) ? (u32)ST_ERR_OFFSET : (u32)ST_OK;
}
// ---- the sequence kernel's bit reader.  seq_step above (the host's prefix decoder uses it) keeps a 192-bit register window alive; the
// device carries NOTHING of the stream between fields: the cursor is a bit position, and the bits below any bit position are aligned words
// of the block's window in workgroup memory, funnel-shifted (v_alignbit_b32). ----
ZS_FN u32 funnel32(u32 hi, u32 lo, u32 sh) { return (u32)((((u64)hi << 32) | (u64)lo) >> (sh & 31u)); }        // (hi:lo) >> sh, sh < 32
// the top n bits of a word, n ≤ 31 (n = 0 → 0): one bit-field extract on the device
#if defined(__HIP_DEVICE_COMPILE__)
ZS_FN u32 top_field(u32 h, u32 n) { return __builtin_amdgcn_ubfe(h, 32u - n, n); }
#else
ZS_FN u32 top_field(u32 h, u32 n) { return n ? h >> (32u - n) : 0u; }
#endif
// a choice that must compile to a select: lanes of a wave decode different blocks, a branch that one lane takes is paid by all of them
#if defined(__clang__)
#define ZS_SEL(c, a, b) (__builtin_unpredictable(c) ? (a) : (b))
#else
#define ZS_SEL(c, a, b) ((c) ? (a) : (b))
#endif
struct SeqBits {
  ZS_LDS u32* w;           // kRing / 4 + 2 words: the window, its first two words repeated behind its last (seq_fill_done)
  u32 bias8;               // stream bit b sits at window bit b + bias8 + 64 (the 64: a window starts two words below its top word)
  i32 bitpos;              // bits left: the next field ends just below stream bit `bitpos`
  ZS_FN u64 below(i32 top) const {               // the 64 stream bits below bit `top`, the highest on top
    const u32 t = (u32)top + bias8;              // (a cursor that ran below the stream's first bit wraps — in the window too: reads stay inside it)
    const u32 q = (t >> 5) & (u32)(kRing / 4 - 1);
    const u32 d0 = w[q], d1 = w[q + 1], d2 = w[q + 2];
    return ((u64)funnel32(d2, d1, t) << 32) | (u64)funnel32(d1, d0, t);
  }
  ZS_FN u32 get(u32 n) {                         // n ≤ 31
    const u32 v = top_field((u32)(below(bitpos) >> 32), n);
    bitpos -= (i32)n;
    return v;
  }
  ZS_FN bool init(ZS_LDS u32* ring, u32 bias, u32 len) {
    w = ring;
    bias8 = 8u * bias - 64u;
    bitpos = 0;
    if (len == 0) return false;
    const u32 at = (len - 1u + bias) & (kRing - 1);
    const u32 last = (w[at >> 2] >> (8u * (at & 3u))) & 0xffu;
    if (last == 0) return false;
    bitpos = 8 * ((i32)len - 1) + highbit(last);
    return true;
  }
  ZS_FN i32 cursor_byte() const { return bitpos > 0 ? (bitpos + 7) >> 3 : 0; }
};
// ---- kernel A1: the literals of one block — one 64-thread workgroup.  Threads 0 … 3 decode a Huffman stream each, 256 symbols a round,
// out of their stream's window in workgroup memory into a buffer there; between rounds ALL threads slide the windows and write the
// buffers out.  Raw / RLE literals (and raw / RLE blocks) are copied or filled by all threads. ----
constexpr u32 kLitRound = 256;
struct LitLds {
  u16 huf[1 << kHufLogMax];
  u32 ring[4][kRing / 4];
  u8 obuf[4][kLitRound];
  u8 hdr[160];               // the tree description, staged
  u32 wfse[64];
  u8 weights[256];
  i16 norm[64];
  u16 next[64];
  i32 huf_log;
  i32 low[4], cursor[4];     // per stream: the window's lowest loaded position, the byte decoding resumes at
  u32 bias[4];
  u32 s_off[4], s_len[4], s_cnt[4];   // per stream: offset from lit_pos, length, symbols
  u32 nstreams;
  u32 status;
};
typedef BackBitsT<RingBytes<ZS_LDS u32*, false>> RingReader;
struct LitState { RingReader b; u32 left; };                // a decoding thread's registers between rounds
// phase 1 (all threads): stage the tree description; thread 0: the streams' extents
ZS_FN void lit_stage(ZS_LDS LitLds* L, const u8* src, const ZBlock& b, int t) {
  if (b.type != BT_COMPRESSED || b.lit_type != LT_HUF) return;
  for (u32 i = (u32)t; i < 160u; i += 64) L->hdr[i] = i < b.huf_desc_len ? src[b.huf_desc + i] : (u8)0;
  if (t != 0) return;
  const u8* s = src + b.lit_pos;
  if (b.lit_streams == 1) {
    L->nstreams = 1;
    L->s_off[0] = 0;
    L->s_len[0] = b.lit_len;
    L->s_cnt[0] = b.lit_regen;
    return;
  }
  L->nstreams = 4;
  if (b.lit_len < 10) { L->status = ST_ERR_HUF; L->nstreams = 0; return; }
  const u32 l1 = (u32)s[0] | ((u32)s[1] << 8), l2 = (u32)s[2] | ((u32)s[3] << 8), l3 = (u32)s[4] | ((u32)s[5] << 8);
  const u32 seg = (b.lit_regen + 3) / 4;
  if (6 + l1 + l2 + l3 >= b.lit_len || 3 * seg > b.lit_regen) { L->status = ST_ERR_HUF; L->nstreams = 0; return; }
  L->s_off[0] = 6;            L->s_len[0] = l1; L->s_cnt[0] = seg;
  L->s_off[1] = 6 + l1;       L->s_len[1] = l2; L->s_cnt[1] = seg;
  L->s_off[2] = 6 + l1 + l2;  L->s_len[2] = l3; L->s_cnt[2] = seg;
  L->s_off[3] = 6 + l1 + l2 + l3; L->s_len[3] = b.lit_len - 6 - l1 - l2 - l3; L->s_cnt[3] = b.lit_regen - 3 * seg;
}
// phase 2 (thread 0): the table; (all threads, after it): the streams' first windows
ZS_FN void lit_table(ZS_LDS LitLds* L, const ZBlock& b) {
  if (b.type != BT_COMPRESSED || b.lit_type != LT_HUF || L->status) return;
  int nw = 0;
  const u32 used = huf_read_weights((ZS_LDS u8*)L->hdr, b.huf_desc_len < 160u ? b.huf_desc_len : 160u, L->weights, nw, L->wfse, L->norm, L->next);
  const int log = used ? huf_build(L->weights, nw, L->huf) : 0;
  L->huf_log = log;
  if (!log) L->status = ST_ERR_HUF;
  for (u32 k = 0; k < L->nstreams; k++) {
    L->bias[k] = ring_bias(L->s_len[k]);
    L->cursor[k] = (i32)L->s_len[k];
    L->low[k] = ring_top_for(L->s_len[k], L->bias[k]);
  }
}
// (all threads) slide stream k's window down to where its next round may reach — 16 threads per stream
ZS_FN void lit_fill(ZS_LDS LitLds* L, const u8* src, const ZBlock& b, i32 page_len, int t) {
  const u32 k = (u32)t >> 4;
  if (k >= L->nstreams) return;
  const i32 spos = (i32)(b.lit_pos + L->s_off[k]);
  const i32 want = ring_low_for(L->cursor[k], L->bias[k]);
  if (want < L->low[k]) ring_fill((ZS_LDS u32*)L->ring[k], L->bias[k], src + spos, want, L->low[k], -spos, page_len + 16 - spos, t & 15, 16);
}
ZS_FN void lit_fill_done(ZS_LDS LitLds* L, int k) {           // (the stream's thread, after the barrier behind lit_fill)
  const i32 want = ring_low_for(L->cursor[k], L->bias[k]);
  if (want < L->low[k]) L->low[k] = want;
}
// a decoding thread starts its stream / decodes its next ≤ 256 symbols into obuf
ZS_FN void lit_begin(ZS_LDS LitLds* L, LitState& st, int k) {
  RingBytes<ZS_LDS u32*, false> src;
  src.w = (ZS_LDS u32*)L->ring[k];
  src.bias = L->bias[k];
  st.left = L->s_cnt[k];
  if (!st.b.init(src, L->s_len[k])) { L->status = ST_ERR_HUF; st.left = 0; }
}
ZS_FN void lit_round(ZS_LDS LitLds* L, LitState& st, int k) {
  const u32 n = st.left < kLitRound ? st.left : kLitRound;
  huf_decode_symbols(st.b, (ZS_LDS u16*)L->huf, L->huf_log, (ZS_LDS u8*)L->obuf[k], n);
  st.left -= n;
  L->cursor[k] = st.b.cursor_byte();
  if (st.left == 0 && n && st.b.bitpos != 0) L->status = ST_ERR_HUF;       // the stream must end exactly with its last symbol
}
// (all threads) round r's symbols to the literal scratch: 16 bytes per thread
ZS_FN void lit_flush(const ZS_LDS LitLds* L, const ZBlock& b, u8* lits, u32 r, int t) {
  const u32 k = (u32)t >> 4, j = ((u32)t & 15u) * 16u;
  if (k >= L->nstreams) return;
  const u32 seg = L->nstreams == 1 ? 0u : (b.lit_regen + 3) / 4;
  const u32 done = r * kLitRound;                              // symbols of the stream written before this round
  if (done >= L->s_cnt[k]) return;
  const u32 n = L->s_cnt[k] - done < kLitRound ? L->s_cnt[k] - done : kLitRound;
  u8* out = lits + b.lit_first + k * seg + done;
  for (u32 i = j; i < j + 16 && i < n; i++) out[i] = L->obuf[k][i];
}
// raw / RLE literals, raw / RLE blocks (all threads)
ZS_FN void lit_plain(const u8* src, const ZBlock& b, u8* lits, int t) {
  u8* out = lits + b.lit_first;
  if (b.type == BT_RAW || (b.type == BT_COMPRESSED && b.lit_type == LT_RAW)) {
    const u8* from = src + (b.type == BT_RAW ? b.pos : b.lit_pos);
    for (u32 i = (u32)t; i < b.lit_regen; i += 64) out[i] = from[i];
  } else if (b.type == BT_RLE || b.lit_type == LT_RLE) {
    const u8 v = src[b.type == BT_RLE ? b.pos : b.lit_pos];
    for (u32 i = (u32)t; i < b.lit_regen; i += 64) out[i] = v;
  }
}
ZS_FN u32 lit_rounds(const ZBlock& b) {                        // rounds of a Huffman block (uniform over the workgroup)
  if (b.type != BT_COMPRESSED || b.lit_type != LT_HUF) return 0;
  const u32 longest = b.lit_streams == 1 ? b.lit_regen : (b.lit_regen + 3) / 4;
  return (longest + kLitRound - 1) / kLitRound;
}

// ---- kernel A2 (round 4): the sequences of FOUR blocks — one 64-thread workgroup, 16 threads per block; the serial chain carries ONLY what
// is serial.  A block's sequences are one chain (three interleaved FSE states over one backward bitstream), and one wave issues about one
// instruction per four cycles whatever the instruction does for how many lanes, so a block costs (instructions per sequence) × 4 cycles ×
// its sequence count — with every block of a launch in flight at once that IS the kernel's duration.  Round 3's step spent ≈ 100 instructions
// per sequence in the decoding lane: values, repeat offsets, sums and records all sat in the chain.  Here a round of 64 sequences is
//   (1) chain    — the group's thread: per sequence only "three table entries → how many bits → the next three states"; it notes the
//                  entries and the cursor (16 bytes) and moves on: ≈ 40 instructions, one workgroup-memory latency;
//   (2) values   — the group's 16 threads, four sequences each: literal length, match length and offset value from the noted entries and
//                  cursor (the bitstream's window is still there);
//   (3) history  — the repeat-offset rules over the 64 offset values: a recurrence, but one whose steps compose — a scan over the group (below);
//   (4) records  — with (3): every thread writes its four sequences' records (lengths, offset, positions inside the block); the window slides.
// Table entries are one word — bits of the next state [0, 5) | extra bits of the value [5, 11) | symbol [11, 17) | word address of the
// next state's base entry [17, 32) — so that ONE three-operand add of a sequence's entries yields both bit totals (neither field can carry:
// ≤ 26 and ≤ 63) and the next state's address is an add and a shift.  10 KiB of workgroup memory per block, four workgroups per CU: every
// block of a 480-page scan is in flight at once. ----
#ifndef ZS_SEQ_LANES
#define ZS_SEQ_LANES 4
#endif
constexpr int kSeqLanes = ZS_SEQ_LANES;
constexpr int kSeqGroup = 64 / kSeqLanes;         // threads per block's group
constexpr u32 kSeqRound = 64;
static_assert(kSeqRound * 89 / 8 + 16 <= 1024, "a round's sequences (≤ 89 bits each) and the chain's four-word window must stay inside the 1 KiB the window keeps below the cursor");
constexpr u32 kTabLL = 0, kTabML = 512, kTabOF = 1024, kTabWords = 1280;       // accuracy logs ≤ 9 / 9 / 8
struct alignas(16) SeqQuad { u32 x, y, z, w; };
ZS_FN SeqQuad quad_load(const ZS_LDS SeqQuad* p) { SeqQuad v; v.x = p->x; v.y = p->y; v.z = p->z; v.w = p->w; return v; }       // (one 16-byte access on the device)
ZS_FN void quad_store(ZS_LDS SeqQuad* p, u32 x, u32 y, u32 z, u32 w) { p->x = x; p->y = y; p->z = z; p->w = w; }
struct alignas(16) SeqBlockLds {
  u32 ring[kRing / 4 + 4];              // first: the chain's four window words are one address + immediate offsets.  (+ 3: the first three words again, so that four words in a row never wrap)
  u32 tab[kTabWords];
  union {
    SeqQuad chain[kSeqRound];           // (1) → (2): the sequence's three table entries and the cursor in front of it
    u8 hdr[3][128];                     // the table descriptions, staged (read before the first round)
    struct {                            // (3), when the chain's notes have been read:
      SeqQuad scan[2][64 / ZS_SEQ_LANES];     // the threads' history functions (three slots + a "malformed" mark), double-buffered for the scan
      u32 psum[2][64 / ZS_SEQ_LANES][2];      // the threads' literal / match byte counts, scanned with the functions
    } h;
  } u;
  SeqQuad vals[kSeqRound];              // (2) → (3): ll, ml, offset value → offset relative to the thread's run, unused
  i32 hist[2][4];                       // the repeat-offset history in front of round r: hist[r & 1] (symbolic until an offset of this block replaces an entry)
  u32 sums[2][2];                       // literal / match bytes in front of round r: sums[r & 1] (checked every round: no overflow in between)
  i16 norm[64];
  u16 next[64];
  i32 fse_log[3];
  i32 low, cursor;
  u32 bias, rcount, rounds, status;
  u32 pad[3];
};
static_assert(sizeof(SeqBlockLds) * 4 + 512 <= 40960 || ZS_SEQ_LANES != 4, "four workgroups of four blocks must fit a CU's 160 KiB: every block of a 480-page launch in flight at once");
struct SeqLds {
  SeqBlockLds b[kSeqLanes];
  u32 llc[36], mlc[53];                 // the literal-length / match-length codes' (base | extra bits << 24)
};
// A table entry names its successor's base by ADDRESS: on the device the address workgroup memory itself uses (no base register to add in
// the chain); on the host (emulation) the offset from the workgroup's struct.
#if defined(ZS2_DEVICE_ONLY)
ZS_FN u32 seq_lds_addr(const ZS_LDS SeqLds*, const ZS_LDS void* p) { return (u32)(__UINTPTR_TYPE__)p; }
ZS_FN u32 seq_lds_word(const ZS_LDS SeqLds*, u32 addr) { return *(const ZS_LDS u32*)(__UINTPTR_TYPE__)addr; }
#else
ZS_FN u32 seq_lds_addr(const SeqLds* L, const void* p) { return (u32)((const u8*)p - (const u8*)L); }
ZS_FN u32 seq_lds_word(const SeqLds* L, u32 addr) { u32 v; __builtin_memcpy(&v, (const u8*)L + addr, 4); return v; }
#endif
ZS_FN u32 rotr32(u32 v, u32 n) { return funnel32(v, v, n); }                      // by n mod 32 (one v_alignbit_b32)
#if defined(__HIP_DEVICE_COMPILE__)
ZS_FN u32 low_field(u32 v, u32 n) { return __builtin_amdgcn_ubfe(v, 0u, n); }    // the low (n mod 32) bits: the width operand is the entry itself
#else
ZS_FN u32 low_field(u32 v, u32 n) { return v & ((1u << (n & 31u)) - 1u); }
#endif
struct SeqState {
  u32 aL, aM, aO;             // addresses of the three states' entries
  u32 tq;                     // the cursor as a window bit: stream bit b ↔ b + 8·bias − 96 (the chain's window = words (tq >> 5) … + 3)
  u32 done;
};
ZS_FN bool seq_block_has_stream(const ZBlock& b) { return b.type == BT_COMPRESSED && b.nseq > 0; }
ZS_FN u32 seq_rounds(const ZBlock& b) { return seq_block_has_stream(b) ? (b.nseq + kSeqRound - 1) / kSeqRound : 0; }
ZS_FN i32 seq_bitpos(const ZS_LDS SeqBlockLds* B, u32 tq) { return (i32)(tq + 96u - 8u * B->bias); }
// k: the block's group, tt: the thread within the group; "the group's thread" = its thread 0
// phase 1 (the group): stage the table descriptions (this block's, or — repeat mode — an earlier block's)
ZS_FN void seq_stage(ZS_LDS SeqLds* L, int k, const u8* src, const ZBlock& b, u32 src_len, int tt) {
  ZS_LDS SeqBlockLds* B = &L->b[k];
  if (tt == 0) { B->status = 0; B->rcount = 0; B->rounds = seq_rounds(b); B->sums[0][0] = B->sums[0][1] = 0; }
  const u32 t = (u32)k * (u32)kSeqGroup + (u32)tt;                       // (the workgroup's 64 threads fill the code tables once)
  if (t < 36u) L->llc[t] = ll_code_entry(t);
  if (t < 53u) L->mlc[t] = ml_code_entry(t);
  if (!seq_block_has_stream(b)) return;
  for (u32 i = (u32)tt; i < 3u * 128u; i += (u32)kSeqGroup) {
    const u32 kind = i >> 7, j = i & 127u;
    const u32 q = b.tab_desc[kind] + j;
    B->u.hdr[kind][j] = (b.tab_mode[kind] != TM_PREDEF && q < src_len) ? src[q] : (u8)0;
  }
  if (tt == 0) { B->bias = ring_bias(b.bits_len); B->cursor = (i32)b.bits_len; B->low = ring_top_for(b.bits_len, B->bias); }
}
ZS_FN u32 seq_tab_first(int kind) { return kind == 0 ? kTabLL : kind == 1 ? kTabOF : kTabML; }       // (kinds in the format's order: LL, OF, ML)
// phase 2 (the group's thread): the block's three tables, fse_build's entries repacked for the chain
ZS_FN void seq_tables(ZS_LDS SeqLds* L, int k, const ZBlock& b) {
  if (!seq_block_has_stream(b)) return;
  ZS_LDS SeqBlockLds* B = &L->b[k];
  for (int kind = 0; kind < 3; kind++) {
    ZS_LDS u32* tab = (ZS_LDS u32*)B->tab + seq_tab_first(kind);
    const int log = seq_table(kind, b.tab_mode[kind], (ZS_LDS u8*)B->u.hdr[kind], 128u, tab, (ZS_LDS i16*)B->norm, (ZS_LDS u16*)B->next);
    B->fse_log[kind] = log;
    if (log < 0) { B->status = ST_ERR_FSE; continue; }
    const u32 first = seq_lds_addr(L, tab) >> 2;
    for (u32 i = 0; i < (1u << log); i++) {
      const u32 e = tab[i], sym = e & 0xffu, nb = (e >> 8) & 0xffu, nx = e >> 16;
      const u32 extra = kind == 1 ? sym : ((kind == 0 ? L->llc[sym] : L->mlc[sym]) >> 24);
      tab[i] = nb | (extra << 5) | (sym << 11) | ((first + nx) << 17);
    }
  }
}
// (the group) slide the bitstream's window down
ZS_FN void seq_fill(ZS_LDS SeqLds* L, int k, const u8* src, const ZBlock& b, i32 page_len, int tt) {
  if (!seq_block_has_stream(b)) return;
  ZS_LDS SeqBlockLds* B = &L->b[k];
  const i32 want = ring_low_for(B->cursor, B->bias);
  if (want < B->low) ring_fill((ZS_LDS u32*)B->ring, B->bias, src + b.bits_pos, want, B->low, -(i32)b.bits_pos, page_len + 16 - (i32)b.bits_pos, tt, kSeqGroup);
}
ZS_FN void seq_fill_done(ZS_LDS SeqLds* L, int k) {          // (the group's thread, behind the barrier that follows seq_fill)
  ZS_LDS SeqBlockLds* B = &L->b[k];
  const i32 want = ring_low_for(B->cursor, B->bias);
  if (want < B->low) B->low = want;
  for (int j = 0; j < 3; j++) B->ring[kRing / 4 + j] = B->ring[j];
}
ZS_FN void seq_start(ZS_LDS SeqLds* L, int k, SeqState& st, const ZBlock& b) {        // the group's thread, behind the first fill
  ZS_LDS SeqBlockLds* B = &L->b[k];
  st.done = 0;
  if (B->status) return;
  SeqBits rd;
  if (!rd.init((ZS_LDS u32*)B->ring, B->bias, b.bits_len)) { B->status = ST_ERR_BITS; return; }
  const u32 sll = rd.get((u32)B->fse_log[0]), sof = rd.get((u32)B->fse_log[1]), sml = rd.get((u32)B->fse_log[2]);
  st.aL = seq_lds_addr(L, &B->tab[kTabLL + sll]);
  st.aO = seq_lds_addr(L, &B->tab[kTabOF + sof]);
  st.aM = seq_lds_addr(L, &B->tab[kTabML + sml]);
  st.tq = (u32)rd.bitpos + 8u * B->bias - 96u;
  for (int j = 0; j < 3; j++) B->hist[0][j] = rep_symbolic(j);
}
// (1) the group's thread: the chain over the block's next ≤ 64 sequences
ZS_FN void seq_chain_round(ZS_LDS SeqLds* L, int k, SeqState& st, const ZBlock& b) {
  ZS_LDS SeqBlockLds* B = &L->b[k];
  B->rcount = 0;
  if (B->status || !seq_block_has_stream(b)) return;
  const u32 left = b.nseq - st.done, n = left < kSeqRound ? left : kSeqRound;
  u32 aL = st.aL, aM = st.aM, aO = st.aO, tq = st.tq, nst = 0;
  for (u32 i = 0; i < n; i++) {
    const u32 eL = seq_lds_word(L, aL), eM = seq_lds_word(L, aM), eO = seq_lds_word(L, aO);
    const u32 q = (tq >> 5) & (u32)(kRing / 4 - 1);
    const u32 w0 = B->ring[q], w1 = B->ring[q + 1], w2 = B->ring[q + 2], w3 = B->ring[q + 3];    // w3 holds stream-window bit tq + 96: the cursor
    quad_store(&B->u.chain[i], eL, eM, eO, tq);
    const u32 t = eL + eM + eO;
    nst = t & 31u;                                   // bits of the three next states (≤ 9 + 9 + 8)
    const u32 val = (t >> 5) & 63u;                  // extra bits of the three values (≤ 16 + 16 + 31): skipped here, read in (2)
    // S: the 32 window bits below the values' bits — bit offset u = 1 … 95 of the four words
    const u32 u = (tq & 31u) + 64u - val;
    const u32 lo = ZS_SEL(u < 32u, w0, ZS_SEL(u < 64u, w1, w2)), hi = ZS_SEL(u < 32u, w1, ZS_SEL(u < 64u, w2, w3));
    const u32 S = funnel32(hi, lo, u);
    // the states' fields sit at the TOP of S, literal length first: rotate each down to bit 0 and cut it with its own entry as the width
    const u32 sL = 0u - eL, sM = sL - eM, sO = sM - eO;          // (only the low five bits count: −(bits so far))
    const u32 bL = low_field(rotr32(S, sL), eL), bM = low_field(rotr32(S, sM), eM), bO = low_field(rotr32(S, sO), eO);
    aL = ((eL >> 17) + bL) << 2;
    aM = ((eM >> 17) + bM) << 2;
    aO = ((eO >> 17) + bO) << 2;
    tq -= val + nst;
  }
  st.done += n;
  if (st.done == b.nseq) tq += nst;                  // behind the last sequence no state is updated: its bits were never there
  st.aL = aL; st.aM = aM; st.aO = aO; st.tq = tq;
  if (seq_bitpos(B, tq) < 0) { B->status = ST_ERR_BITS; return; }           // the stream ran out: nothing more to decode from it
  B->rcount = n;
  const i32 bp = seq_bitpos(B, tq);
  B->cursor = bp > 0 ? (bp + 7) >> 3 : 0;
}
// (2) the group: the round's values from the noted entries and cursors
ZS_FN void seq_values(ZS_LDS SeqLds* L, int k, int tt) {
  ZS_LDS SeqBlockLds* B = &L->b[k];
  const u32 n = B->rcount;
  for (u32 i = (u32)tt; i < n; i += (u32)kSeqGroup) {
    const SeqQuad c = quad_load(&B->u.chain[i]);
    const u32 le = (c.x >> 5) & 63u, me = (c.y >> 5) & 63u, oe = (c.z >> 5) & 63u;
    const u32 cl = L->llc[(c.x >> 11) & 63u], cm = L->mlc[(c.y >> 11) & 63u];
    const u32 q = (c.w >> 5) & (u32)(kRing / 4 - 1);
    const u32 w1 = B->ring[q + 1], w2 = B->ring[q + 2], w3 = B->ring[q + 3];
    u64 wv = ((u64)funnel32(w3, w2, c.w) << 32) | (u64)funnel32(w2, w1, c.w);       // the 64 bits below the cursor
    const u32 ov = (1u << oe) + top_field((u32)(wv >> 32), oe);                    // offset value first, then match length, then literal length
    wv <<= oe;
    const u32 ml = (cm & 0xffffffu) + top_field((u32)(wv >> 32), me);
    wv <<= me;
    const u32 ll = (cl & 0xffffffu) + top_field((u32)(wv >> 32), le);
    quad_store(&B->vals[i], ll, ml, ov, 0u);
  }
}
// (3) the repeat-offset rules, as a scan.  The rules are a recurrence over three history entries — serial, 27 instructions per sequence, a
// third of the kernel when the group's thread walked them alone.  But a run of sequences acts on the history as a FUNCTION whose three
// outputs are each "a real offset" or "input entry j minus d" — exactly the symbolic values blocks already start with (rep_symbolic:
// −(1 + j + 3·d)) — and such functions compose.  So: (3a) every thread of the group walks ITS four sequences from the symbolic history
// (−1, −2, −3): the history it ends with is its run's function, the offsets it notes are relative to the run's start; (3b) an inclusive
// scan over the group's functions (log₂ 16 steps); (3c) every thread evaluates the function of the runs before it on the round's real
// history → the history its run starts from → its noted offsets become real (or block-symbolic, where the block's own start shows through).
// The rules themselves (seq_step's, every candidate computed, every choice a select): a value > 3 is a new offset, pushed onto the history;
// 1 … 3 names an entry (shifted by one when there are no literals, the fourth choice being "the newest entry minus one"), which moves to
// the front.
constexpr u32 kSeqPer = kSeqRound / (u32)kSeqGroup;         // sequences per thread and round
constexpr int kSeqScanSteps = kSeqGroup == 16 ? 4 : kSeqGroup == 8 ? 3 : kSeqGroup == 32 ? 5 : 6;
static_assert((1 << kSeqScanSteps) == kSeqGroup, "the history scan runs over the group's threads");
// a history value `a` minus d: a real offset stays real (and must stay positive: `bad`), a symbolic one adds d to its delta
ZS_FN i32 hist_minus(i32 a, u32 d, u32& bad) {
  const i32 real = (i32)((u32)a - d), sym = (i32)((u32)a - 3u * d);      // (unsigned: the unused branch of hist_eval may wrap)
  bad |= ((a > 0) & (real <= 0)) ? 1u : 0u;
  return ZS_SEL(a > 0, real, sym);
}
// what slot s (> 0: a real offset; ≤ 0: −(1 + j + 3·d) = input entry j minus d) is worth on the inputs a0, a1, a2
ZS_FN i32 hist_eval(i32 s, i32 a0, i32 a1, i32 a2, u32& bad) {
  const u32 t = ~(u32)s, d = (u32)(((u64)t * 0xAAAAAAABull) >> 33), j = t - 3u * d;       // (s > 0: t wraps, the result below is not used)
  const i32 a = ZS_SEL(j == 0u, a0, ZS_SEL(j == 1u, a1, a2));
  u32 b = 0;
  const i32 v = hist_minus(a, d, b);
  bad |= ZS_SEL(s > 0, 0u, b);
  return ZS_SEL(s > 0, s, v);
}
// (3a) (the group) every thread: the rules over its own sequences, from the symbolic history
ZS_FN void seq_history_local(ZS_LDS SeqLds* L, int k, int tt) {
  ZS_LDS SeqBlockLds* B = &L->b[k];
  const u32 n = B->rcount, i0 = (u32)tt * kSeqPer;
  i32 r0 = rep_symbolic(0), r1 = rep_symbolic(1), r2 = rep_symbolic(2), worst = 1;
  u32 sum_ll = 0, sum_ml = 0;
  for (u32 i = i0; i < i0 + kSeqPer && i < n; i++) {
    const SeqQuad v = quad_load(&B->vals[i]);
    const u32 ll = v.x, ov = v.z;
    const bool rep = ov <= 3u;
    const u32 idx = ov - 1u + (ll == 0 ? 1u : 0u);
    const i32 less = r0 + ZS_SEL(r0 > 0, -1, -3);
    i32 picked = ZS_SEL(idx == 2u, r2, less);
    picked = ZS_SEL(idx == 1u, r1, picked);
    picked = ZS_SEL(idx == 0u, r0, picked);
    const i32 off = ZS_SEL(rep, picked, (i32)(ov - 3u));
    const bool front = rep & (idx == 0u), second = rep & (idx <= 1u);
    r2 = ZS_SEL(second, r2, r1);
    r1 = ZS_SEL(front, r1, r0);
    r0 = off;                                                  // (front: off IS r0)
    const i32 fresh = ZS_SEL(rep, ZS_SEL(off == 0, 0, 1), off);       // a new offset must be positive; "one less" must not reach zero
    worst = fresh < worst ? fresh : worst;
    sum_ll += ll;
    sum_ml += v.y;
    B->vals[i].z = (u32)off;
  }
  quad_store(&B->u.h.scan[0][tt], (u32)r0, (u32)r1, (u32)r2, worst <= 0 ? 1u : 0u);
  B->u.h.psum[0][tt][0] = sum_ll;
  B->u.h.psum[0][tt][1] = sum_ml;
}
// (3b) (the group) step p of the inclusive scan: the function of the runs up to and including this thread's
ZS_FN void seq_history_step(ZS_LDS SeqLds* L, int k, int p, int tt) {
  ZS_LDS SeqBlockLds* B = &L->b[k];
  const int from = p & 1, d = 1 << p;
  SeqQuad t = quad_load(&B->u.h.scan[from][tt]);
  u32 pl = B->u.h.psum[from][tt][0], pm = B->u.h.psum[from][tt][1];
  if (tt >= d) {
    pl += B->u.h.psum[from][tt - d][0];
    pm += B->u.h.psum[from][tt - d][1];
    const SeqQuad e = quad_load(&B->u.h.scan[from][tt - d]);       // the earlier runs first
    u32 bad = t.w | e.w;
    const i32 s0 = hist_eval((i32)t.x, (i32)e.x, (i32)e.y, (i32)e.z, bad), s1 = hist_eval((i32)t.y, (i32)e.x, (i32)e.y, (i32)e.z, bad),
              s2 = hist_eval((i32)t.z, (i32)e.x, (i32)e.y, (i32)e.z, bad);
    t.x = (u32)s0; t.y = (u32)s1; t.z = (u32)s2; t.w = bad;
  }
  quad_store(&B->u.h.scan[from ^ 1][tt], t.x, t.y, t.z, t.w);
  B->u.h.psum[from ^ 1][tt][0] = pl;
  B->u.h.psum[from ^ 1][tt][1] = pm;
}
// (3c) + (4) (the group) every thread: its run's starting history and positions; its sequences become records — literal length, match
// length, the noted offset made real (or block-symbolic), and where the sequence's output and literals start INSIDE the block (kernel C adds
// the block's own position once kernel B knows it) — written straight to global memory, a thread's four records side by side.  The round's
// last run leaves the history and the byte counts behind.
ZS_FN void seq_history_apply(ZS_LDS SeqLds* L, int k, u32 round, ZRec* recs_block, int tt) {
  ZS_LDS SeqBlockLds* B = &L->b[k];
  const u32 n = B->rcount, i0 = (u32)tt * kSeqPer, fin = (u32)(kSeqScanSteps & 1);
  if (i0 >= n) return;
  const i32 h0 = B->hist[round & 1u][0], h1 = B->hist[round & 1u][1], h2 = B->hist[round & 1u][2];
  const u32 cl = B->sums[round & 1u][0], cm = B->sums[round & 1u][1];
  u32 bad = 0;
  i32 a0 = h0, a1 = h1, a2 = h2;
  u32 lit = cl, out = cl + cm;
  if (tt) {
    const SeqQuad e = quad_load(&B->u.h.scan[fin][tt - 1]);
    bad = e.w;
    a0 = hist_eval((i32)e.x, h0, h1, h2, bad);
    a1 = hist_eval((i32)e.y, h0, h1, h2, bad);
    a2 = hist_eval((i32)e.z, h0, h1, h2, bad);
    lit += B->u.h.psum[fin][tt - 1][0];
    out += B->u.h.psum[fin][tt - 1][0] + B->u.h.psum[fin][tt - 1][1];
  }
  ZRec* r = recs_block + round * kSeqRound;
  for (u32 i = i0; i < i0 + kSeqPer && i < n; i++) {
    const SeqQuad v = quad_load(&B->vals[i]);
    r[i].out_pos = out;
    r[i].lit_pos = lit;
    r[i].ll = v.x;
    r[i].ml = v.y;
    r[i].off = hist_eval((i32)v.z, a0, a1, a2, bad);
    out += v.x + v.y;
    lit += v.x;
  }
  if (i0 + kSeqPer >= n) {                                     // the run that holds the round's last sequence
    const SeqQuad e = quad_load(&B->u.h.scan[fin][tt]);
    bad |= e.w;
    B->hist[(round + 1u) & 1u][0] = hist_eval((i32)e.x, h0, h1, h2, bad);
    B->hist[(round + 1u) & 1u][1] = hist_eval((i32)e.y, h0, h1, h2, bad);
    B->hist[(round + 1u) & 1u][2] = hist_eval((i32)e.z, h0, h1, h2, bad);
    B->sums[(round + 1u) & 1u][0] = cl + B->u.h.psum[fin][tt][0];
    B->sums[(round + 1u) & 1u][1] = cm + B->u.h.psum[fin][tt][1];
  }
  if (bad) B->status = ST_ERR_OFFSET;
}
// (the group's thread) behind (3c): the round's checks
ZS_FN void seq_round_check(ZS_LDS SeqLds* L, int k, u32 round) {
  ZS_LDS SeqBlockLds* B = &L->b[k];
  if (!B->rcount) {                                            // (a block that is through, or failed: nothing moved — the counts stay where finish looks for them)
    B->sums[(round + 1u) & 1u][0] = B->sums[round & 1u][0];
    B->sums[(round + 1u) & 1u][1] = B->sums[round & 1u][1];
    for (int j = 0; j < 3; j++) B->hist[(round + 1u) & 1u][j] = B->hist[round & 1u][j];
    return;
  }
  if (B->status) { B->rcount = 0; return; }
  if (B->sums[(round + 1u) & 1u][0] > kBlockMax || B->sums[(round + 1u) & 1u][1] > kBlockMax) { B->status = ST_ERR_LENGTH; B->rcount = 0; }     // (every round: the counts cannot wrap in between)
}
// the group's thread, behind the last round: the stream must be used up; the trailing literals; the block's size and history on exit
ZS_FN void seq_finish(ZS_LDS SeqLds* L, int k, SeqState& st, ZBlock* b, ZRec* recs_block, u32 rounds_run) {
  ZS_LDS SeqBlockLds* B = &L->b[k];
  if (!seq_block_has_stream(*b)) {                           // a raw / RLE block, or a block of literals only: one run of literals
    const u32 n = b->type == BT_COMPRESSED ? b->lit_regen : b->size;
    recs_block[0].out_pos = 0;
    recs_block[0].lit_pos = 0;
    recs_block[0].ll = n;
    recs_block[0].ml = 0;
    recs_block[0].off = 0;
    b->out_size = n;
    for (int j = 0; j < 3; j++) b->rep_out[j] = rep_symbolic(j);
    return;
  }
  if (B->status) return;
  const u32 sum_ll = B->sums[rounds_run & 1u][0], sum_ml = B->sums[rounds_run & 1u][1];
  if (st.done != b->nseq || seq_bitpos(B, st.tq) != 0) { B->status = ST_ERR_BITS; return; }
  if (sum_ll > b->lit_regen || sum_ll + sum_ml > kBlockMax) { B->status = ST_ERR_LENGTH; return; }
  recs_block[b->nseq].out_pos = sum_ll + sum_ml;
  recs_block[b->nseq].lit_pos = sum_ll;
  recs_block[b->nseq].ll = b->lit_regen - sum_ll;
  recs_block[b->nseq].ml = 0;
  recs_block[b->nseq].off = 0;
  b->out_size = b->lit_regen + sum_ml;
  for (int j = 0; j < 3; j++) b->rep_out[j] = B->hist[rounds_run & 1u][j];
}
ZS_FN u32 seq_status(const ZS_LDS SeqLds* L, int k) { return L->b[k].status; }
ZS_FN u32 seq_rounds_of(const ZS_LDS SeqLds* L, int k) { return L->b[k].rounds; }

// ---- kernel B: one lane per page ----
ZS_FN void page_blocks(const ZPage& pg, ZBlock* blocks, u32* status, int page_index) {
  if (status[page_index] != ST_OK) return;
  u64 out = 0;
  i32 rep[3] = {1, 4, 8};
  for (i32 k = 0; k < pg.nblocks; k++) {
    ZBlock& b = blocks[pg.block_first + k];
    b.out_base = (u32)out;
    for (int j = 0; j < 3; j++) b.rep_in[j] = rep[j];
    out += b.out_size;
    if (out > (u64)pg.dst_len) { status[page_index] = ST_ERR_LENGTH; return; }
    i32 nr[3];
    for (int j = 0; j < 3; j++) nr[j] = rep_resolve(b.rep_out[j], rep);
    for (int j = 0; j < 3; j++) rep[j] = nr[j];
  }
  if (out != (u64)pg.dst_len) status[page_index] = ST_ERR_LENGTH;
}

// ---- kernel C: one workgroup per block, a pass over its records: kernel A left positions relative to the block and offsets that may
// name the block's initial history; kernel B has since placed the block in its page and handed the history down ----
ZS_FN u32 fix_records(ZRec* r, u32 n, const ZBlock& b, int t, int nthreads) {
  u32 status = 0;
  for (u32 i = (u32)t; i < n; i += (u32)nthreads) {
    ZRec x = r[i];
    x.out_pos += b.out_base;
    x.lit_pos += b.lit_first;
    if (x.ml) {
      const i32 off = rep_resolve(x.off, b.rep_in);
      if (off <= 0 || (u64)(u32)off > (u64)x.out_pos + x.ll) status = ST_ERR_OFFSET;      // reaches before the page's first byte (no dictionaries here)
      x.off = off;
    }
    r[i] = x;
  }
  return status;
}

// ---- kernel D: one workgroup per page, its fragments in order ----
struct ZExecLds {
  comet_snappy2::ExecLds e;     // src[] (the fragment's pointers), flags / changed / covered
  u32 q[kBigQueue][4];          // long parts: fragment-relative position, length, kind (0 literal / 1 match), literal position / offset
  u32 nq;
  u32 next_lo;                  // the record the next fragment starts with
};
// phase 1: the records that overlap the fragment [f0, f1) — from record `lo` on, every thread its share until a record starts at or beyond f1
ZS_FN void zfrag_scatter(ZS_LDS ZExecLds* L, const ZRec* recs, u32 nrecs, u32 lo, u32 f0, u32 f1, const u8* lits, u8* dst, int tid, int nthreads) {
  u32 mine = 0;
  bool copies = false;
  auto queue = [&](u32 x, u32 len, u32 kind, u32 v) {
    const u32 q = SN2_ATOMIC_ADD_LDS(&L->nq, 1u);
    if (q < kBigQueue) { L->q[q][0] = x; L->q[q][1] = len; L->q[q][2] = kind; L->q[q][3] = v; }
    else SN2_ATOMIC_OR_U32(&L->e.flags, 2u);                                       // (cannot happen: ≤ 2 · 65536 / 256 + 2 long parts per fragment)
  };
  for (u32 i = lo + (u32)tid; i < nrecs; i += (u32)nthreads) {
    const ZRec r = recs[i];
    if (r.out_pos >= f1) { SN2_ATOMIC_MIN_LDS(&L->next_lo, i); break; }
    const u32 lit_end = r.out_pos + r.ll, end = lit_end + r.ml;
    if (end > f1) SN2_ATOMIC_MIN_LDS(&L->next_lo, i);                              // straddles the fragment's end: the next fragment starts with it
    if (end <= f0) continue;
    // literals
    const u32 a0 = r.out_pos > f0 ? r.out_pos : f0, a1 = lit_end < f1 ? lit_end : f1;
    if (a1 > a0) {
      const u32 n = a1 - a0, lp = r.lit_pos + (a0 - r.out_pos);
      mine += n;
      if (n >= kBigPart) queue(a0 - f0, n, 0u, lp);
      else for (u32 k = 0; k < n; k++) { dst[a0 + k] = lits[lp + k]; L->e.src[a0 - f0 + k] = (u16)(a0 - f0 + k); }
    }
    // match
    const u32 m0 = lit_end > f0 ? lit_end : f0, m1 = end < f1 ? end : f1;
    if (m1 > m0) {
      const u32 n = m1 - m0, off = (u32)r.off;
      mine += n;
      if (off == 0 || off > m0) { SN2_ATOMIC_OR_U32(&L->e.flags, 2u); continue; }
      if (n >= kBigPart) queue(m0 - f0, n, 1u, off);
      else
        for (u32 k = 0; k < n; k++) {
          const u32 x = m0 + k, s = x - off;
          if (s >= f0) { L->e.src[x - f0] = (u16)(s - f0); copies = true; }
          else { dst[x] = dst[s]; L->e.src[x - f0] = (u16)(x - f0); }              // an earlier fragment: final bytes
        }
    }
  }
  if (mine) SN2_ATOMIC_ADD_U32(&L->e.covered, mine);
  if (copies) SN2_ATOMIC_OR_U32(&L->e.flags, 4u);
}
// phase 2: the long parts, all threads together
ZS_FN void zfrag_long_parts(ZS_LDS ZExecLds* L, u32 f0, const u8* lits, u8* dst, int tid, int nthreads) {
  const u32 n = L->nq < kBigQueue ? L->nq : kBigQueue;
  bool copies = false;
  for (u32 q = 0; q < n; q++) {
    const u32 x0 = L->q[q][0], len = L->q[q][1], kind = L->q[q][2], v = L->q[q][3];
    if (kind == 0) {
      for (u32 k0 = (u32)tid; k0 < len; k0 += 8u * (u32)nthreads) {
        u8 b[8];
        for (int u = 0; u < 8; u++) { const u32 k = k0 + (u32)u * (u32)nthreads; b[u] = k < len ? lits[v + k] : (u8)0; }
        for (int u = 0; u < 8; u++) {
          const u32 k = k0 + (u32)u * (u32)nthreads;
          if (k < len) { dst[f0 + x0 + k] = b[u]; L->e.src[x0 + k] = (u16)(x0 + k); }
        }
      }
    } else {
      for (u32 k = (u32)tid; k < len; k += (u32)nthreads) {
        const u32 x = f0 + x0 + k, s = x - v;
        if (s >= f0) { L->e.src[x - f0] = (u16)(s - f0); copies = true; }
        else { dst[x] = dst[s]; L->e.src[x - f0] = (u16)(x - f0); }
      }
    }
  }
  if (copies) SN2_ATOMIC_OR_U32(&L->e.flags, 4u);
}

}  // namespace comet_zstd2

// ---- host side: the walk over a page's frame and block headers (plain C++; csrc/zstd2.cpp, the scan and the host emulation) ----
#ifndef ZS2_DEVICE_ONLY      // (the .hip file compiles the phases as __device__ functions: the walk, plain host code, cannot call them there)
#include <vector>
namespace comet_zstd2 {
struct PageWalk {
  std::vector<ZBlock> blocks;
  u32 nrecs = 0;            // Σ (nseq + 1)
  u32 nlits = 0;            // Σ literal bytes
  // what the walk met (tests assert that their inputs reach every branch): raw / RLE / compressed blocks; raw / RLE / Huffman (own tree) /
  // treeless literals; one / four streams; FSE-compressed / direct weights; predefined / RLE / described / repeated sequence tables
  u32 seen[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};
enum { SEEN_RAW_BLOCK = 0, SEEN_RLE_BLOCK, SEEN_COMPRESSED_BLOCK, SEEN_LIT_RAW, SEEN_LIT_RLE, SEEN_LIT_HUF, SEEN_LIT_TREELESS, SEEN_ONE_STREAM, SEEN_FOUR_STREAMS,
       SEEN_WEIGHTS_FSE, SEEN_WEIGHTS_DIRECT, SEEN_TAB_PREDEF, SEEN_TAB_RLE, SEEN_TAB_FSE, SEEN_TAB_REPEAT, SEEN_NO_SEQUENCES };
// → true: the page is ONE frame of `expect_out` bytes the device pipeline decodes; false: not for the device (the caller inflates it on
// the host, which also reports what is wrong with it if it is corrupt)
inline bool scan_page(const u8* p, u32 len, u32 expect_out, PageWalk& w) {
  w.blocks.clear();
  w.nrecs = 0;
  w.nlits = 0;
  for (u32& x : w.seen) x = 0;
  if (len < 9 || p[0] != 0x28 || p[1] != 0xb5 || p[2] != 0x2f || p[3] != 0xfd) return false;
  const u32 fhd = p[4];
  const u32 fcs_flag = fhd >> 6, single = (fhd >> 5) & 1u, checksum = (fhd >> 2) & 1u, dict_flag = fhd & 3u;
  if ((fhd & 0x08u) || dict_flag) return false;             // reserved bit; dictionaries are not used by Parquet writers
  if (checksum) return false;                               // a frame that carries a content checksum is verified where libzstd verifies it: on the host
  u32 pos = 5;
  u64 window = 0;
  if (!single) {
    const u32 wd = p[pos++];
    const u32 wlog = 10 + (wd >> 3);
    if (wlog > 31) return false;
    window = (1ull << wlog) + ((1ull << wlog) >> 3) * (wd & 7u);
  }
  const u32 fcs_bytes = fcs_flag == 0 ? (single ? 1u : 0u) : fcs_flag == 1 ? 2u : fcs_flag == 2 ? 4u : 8u;
  if (pos + fcs_bytes > len) return false;
  if (fcs_bytes) {
    u64 fcs = 0;
    for (u32 k = 0; k < fcs_bytes; k++) fcs |= (u64)p[pos + k] << (8 * k);
    if (fcs_bytes == 2) fcs += 256;
    if (fcs != expect_out) return false;
    pos += fcs_bytes;
    if (single) window = fcs;
  }
  (void)window;                                             // matches are checked against the page's first byte, whatever the window says
  i64 huf_desc = -1, huf_desc_len = 0;
  int tab_mode[3] = {-1, -1, -1};
  u32 tab_desc[3] = {0, 0, 0};
  u64 out_known = 0;
  for (;;) {
    if (pos + 3 > len) return false;
    const u32 bh = (u32)p[pos] | ((u32)p[pos + 1] << 8) | ((u32)p[pos + 2] << 16);
    pos += 3;
    const u32 last = bh & 1u, type = (bh >> 1) & 3u, size = bh >> 3;
    if (type == 3 || size > kBlockMax) return false;
    ZBlock b;
    __builtin_memset(&b, 0, sizeof b);
    b.pos = pos;
    b.size = size;
    b.type = (u8)type;
    b.rec_first = w.nrecs;
    b.lit_first = w.nlits;
    w.seen[type == BT_RAW ? SEEN_RAW_BLOCK : type == BT_RLE ? SEEN_RLE_BLOCK : SEEN_COMPRESSED_BLOCK]++;
    if (type == BT_RAW) {
      if (pos + size > len) return false;
      b.lit_regen = size;
      pos += size;
      out_known += size;
    } else if (type == BT_RLE) {
      if (pos + 1 > len) return false;
      b.lit_regen = size;
      pos += 1;
      out_known += size;
    } else {
      if (size < 2 || pos + size > len) return false;
      const u8* c = p + pos;
      const u32 lt = c[0] & 3u, sf = (c[0] >> 2) & 3u;
      u32 hdr, regen, comp = 0, streams = 1;
      if (lt < 2) {
        if (sf == 0 || sf == 2) { hdr = 1; regen = c[0] >> 3; }
        else if (sf == 1) { hdr = 2; if (size < 2) return false; regen = (c[0] >> 4) | ((u32)c[1] << 4); }
        else { hdr = 3; if (size < 3) return false; regen = (c[0] >> 4) | ((u32)c[1] << 4) | ((u32)c[2] << 12); }
        comp = lt == LT_RAW ? regen : 1;
      } else {
        if (size < 5) return false;
        const u64 v = (u64)c[0] | ((u64)c[1] << 8) | ((u64)c[2] << 16) | ((u64)c[3] << 24) | ((u64)c[4] << 32);
        if (sf == 0) { hdr = 3; regen = (u32)(v >> 4) & 0x3ffu; comp = (u32)(v >> 14) & 0x3ffu; streams = 1; }
        else if (sf == 1) { hdr = 3; regen = (u32)(v >> 4) & 0x3ffu; comp = (u32)(v >> 14) & 0x3ffu; streams = 4; }
        else if (sf == 2) { hdr = 4; regen = (u32)(v >> 4) & 0x3fffu; comp = (u32)(v >> 18) & 0x3fffu; streams = 4; }
        else { hdr = 5; regen = (u32)(v >> 4) & 0x3ffffu; comp = (u32)(v >> 22) & 0x3ffffu; streams = 4; }
      }
      if (regen > kBlockMax || hdr + comp > size) return false;
      b.lit_regen = regen;
      b.lit_streams = (u8)streams;
      b.lit_pos = pos + hdr;
      w.seen[lt == 0 ? SEEN_LIT_RAW : lt == 1 ? SEEN_LIT_RLE : lt == 2 ? SEEN_LIT_HUF : SEEN_LIT_TREELESS]++;
      if (lt >= 2) w.seen[streams == 1 ? SEEN_ONE_STREAM : SEEN_FOUR_STREAMS]++;
      if (lt == 0) b.lit_type = LT_RAW;
      else if (lt == 1) b.lit_type = LT_RLE;
      else {
        b.lit_type = LT_HUF;
        u32 tree = 0;
        if (lt == 2) {                                      // its own tree: the description's length from its header byte
          if (comp < 1) return false;
          const u32 hb = c[hdr];
          tree = hb < 128 ? 1 + hb : 1 + ((hb - 127) + 1) / 2;
          w.seen[hb < 128 ? SEEN_WEIGHTS_FSE : SEEN_WEIGHTS_DIRECT]++;
          if (tree > comp) return false;
          huf_desc = (i64)pos + hdr;
          huf_desc_len = tree;
        } else if (huf_desc < 0) return false;              // treeless without an earlier tree (a dictionary's): not for the device
        b.huf_desc = (u32)huf_desc;
        b.huf_desc_len = (u32)huf_desc_len;
        b.lit_pos = pos + hdr + tree;
        b.lit_len = comp - tree;
      }
      // sequences section
      u32 sp = hdr + comp;
      if (sp >= size) {
        if (sp != size) return false;
        b.nseq = 0;                                         // (a block may end with its literals: zstd always writes the count, accept both)
      } else {
        const u32 b0 = c[sp++];
        u32 nseq;
        if (b0 == 0) nseq = 0;
        else if (b0 < 128) nseq = b0;
        else if (b0 < 255) { if (sp + 1 > size) return false; nseq = ((b0 - 128) << 8) + c[sp++]; }
        else { if (sp + 2 > size) return false; nseq = (u32)c[sp] + ((u32)c[sp + 1] << 8) + 0x7f00u; sp += 2; }
        b.nseq = nseq;
        if (nseq == 0) {
          w.seen[SEEN_NO_SEQUENCES]++;
          if (sp != size) return false;
        } else {
          if (sp + 1 > size) return false;
          const u32 modes = c[sp++];
          if (modes & 3u) return false;
          const u32 m[3] = {modes >> 6, (modes >> 4) & 3u, (modes >> 2) & 3u};
          for (int k = 0; k < 3; k++) {
            w.seen[m[k] == 0 ? SEEN_TAB_PREDEF : m[k] == 1 ? SEEN_TAB_RLE : m[k] == 2 ? SEEN_TAB_FSE : SEEN_TAB_REPEAT]++;
            if (m[k] == 0) { tab_mode[k] = TM_PREDEF; tab_desc[k] = 0; }
            else if (m[k] == 1) {
              if (sp + 1 > size) return false;
              tab_mode[k] = TM_RLE;
              tab_desc[k] = pos + sp;
              sp += 1;
            } else if (m[k] == 2) {
              i16 norm[64];
              int nsym, log;
              const u32 used = fse_read_ncount(c + sp, size - sp, kMaxSym[k], kMaxLog[k], norm, nsym, log);
              if (!used) return false;
              tab_mode[k] = TM_FSE;
              tab_desc[k] = pos + sp;
              sp += used;
            } else if (tab_mode[k] < 0) return false;       // repeat without an earlier table
            b.tab_mode[k] = (u8)tab_mode[k];
            b.tab_desc[k] = tab_desc[k];
          }
          if (sp >= size) return false;
          b.bits_pos = pos + sp;
          b.bits_len = size - sp;
        }
      }
      pos += size;
    }
    w.nrecs += b.nseq + 1;
    w.nlits += b.lit_regen;
    w.blocks.push_back(b);
    if (last) break;
    if (w.blocks.size() > 65536) return false;
  }
  if (pos != len) return false;                             // another frame (or garbage) behind the first
  if (out_known > expect_out) return false;
  return true;
}

// The first n bytes of a page's content — a v1 data page's definition levels sit there, and the host needs them for its run tables —
// decoded on the host from the walk's descriptors with the primitives above: the literals and sequences of the first block(s) only as far
// as those bytes reach (a Huffman stream yields its symbols in order from its END, so a prefix is cheap).  → bytes produced (< n: the
// page is malformed or shorter; the caller then inflates it on the host, which reports what is wrong).
inline size_t host_prefix(const u8* p, u32 len, const PageWalk& w, u8* out, size_t n) {
  size_t produced = 0;
  i32 rep[3] = {1, 4, 8};
  std::vector<u8> lits;
  std::vector<ZRec> recs;
  std::vector<u16> huf((size_t)1 << kHufLogMax);
  std::vector<u32> fse(3 * 1024);
  u32 wfse[64], llc[36], mlc[53];
  for (u32 c = 0; c < 36; c++) llc[c] = ll_code_entry(c);
  for (u32 c = 0; c < 53; c++) mlc[c] = ml_code_entry(c);
  u8 weights[256];
  i16 norm[64];
  u16 next[64];
  for (const ZBlock& b : w.blocks) {
    if (produced >= n) break;
    const size_t need = n - produced;
    if (b.type == BT_RAW) { const size_t k = need < b.size ? need : b.size; __builtin_memcpy(out + produced, p + b.pos, k); produced += k; continue; }
    if (b.type == BT_RLE) { const size_t k = need < b.size ? need : b.size; __builtin_memset(out + produced, p[b.pos], k); produced += k; continue; }
    const u32 nl = (u32)(need < b.lit_regen ? need : b.lit_regen);        // a prefix of `need` bytes holds at most `need` literals
    lits.assign((size_t)nl + 8, 0);
    if (b.lit_type == LT_RAW) __builtin_memcpy(lits.data(), p + b.lit_pos, nl);
    else if (b.lit_type == LT_RLE) __builtin_memset(lits.data(), p[b.lit_pos], nl);
    else {
      int nw = 0;
      if (!huf_read_weights(p + b.huf_desc, b.huf_desc_len, weights, nw, wfse, norm, next)) return produced;
      const int log = huf_build(weights, nw, huf.data());
      if (!log) return produced;
      const u8* s = p + b.lit_pos;
      auto stream = [&](const u8* at, u32 slen, u8* to, u32 cnt, bool whole) {
        BackBitsT<PtrBytes<const u8*>> bb;
        PtrBytes<const u8*> src;
        src.p = at;
        src.len = (i32)slen;
        if (!bb.init(src, slen)) return false;
        huf_decode_symbols(bb, huf.data(), log, to, cnt);
        return whole ? bb.bitpos == 0 : bb.bitpos >= 0;
      };
      if (b.lit_streams == 1) {
        if (!stream(s, b.lit_len, lits.data(), nl, nl == b.lit_regen)) return produced;
      } else {
        if (b.lit_len < 10) return produced;
        const u32 l[3] = {(u32)s[0] | ((u32)s[1] << 8), (u32)s[2] | ((u32)s[3] << 8), (u32)s[4] | ((u32)s[5] << 8)};
        if (6 + l[0] + l[1] + l[2] >= b.lit_len) return produced;
        const u32 seg = (b.lit_regen + 3) / 4;
        if (3 * seg > b.lit_regen) return produced;
        u32 off = 6;
        for (u32 k = 0; k < 4 && k * seg < nl; k++) {
          const u32 slen = k < 3 ? l[k] : b.lit_len - off, full = k < 3 ? seg : b.lit_regen - 3 * seg;
          const u32 cnt = nl - k * seg < full ? nl - k * seg : full;
          if (!stream(s + off, slen, lits.data() + k * seg, cnt, cnt == full)) return produced;
          off += slen;
        }
      }
    }
    if (b.nseq == 0) { __builtin_memcpy(out + produced, lits.data(), nl); produced += nl; continue; }
    int logs[3];
    for (int k = 0; k < 3; k++) {
      const u32 d = b.tab_desc[k];
      logs[k] = seq_table(k, b.tab_mode[k], p + d, d < len ? len - d : 0, fse.data() + 1024 * k, norm, next);
      if (logs[k] < 0) return produced;
      seq_table_expand(k, logs[k], fse.data() + 1024 * k, llc, mlc);
    }
    // (a sequence yields at least its 3 match bytes: `need` bytes take at most need / 3 + 1 of them)
    recs.resize((size_t)(b.nseq < need / 3 + 2 ? b.nseq : need / 3 + 2) + 1);
    u32 ndone = 0;
    i32 rep_out[3];
    {
      BackBitsT<PtrBytes<const u8*>> bb;
      PtrBytes<const u8*> src;
      src.p = p + b.bits_pos;
      src.len = (i32)b.bits_len;
      if (!bb.init(src, b.bits_len)) return produced;
      SeqCore c;
      seq_begin(bb, c, logs[0], logs[1], logs[2]);
      while (c.done < b.nseq && c.sum_ll + c.sum_ml < need) {
        ZRec& r = recs[c.done];
        if (seq_step(bb, c, fse.data(), fse.data() + 1024, fse.data() + 2048, c.done + 1 == b.nseq, r.ll, r.ml, r.off) != ST_OK) return produced;
      }
      ndone = c.done;
      if (c.sum_ll > b.lit_regen) return produced;
      if (ndone == b.nseq) {
        if (bb.bitpos != 0) return produced;
        recs[b.nseq].ll = b.lit_regen - (u32)c.sum_ll;
        recs[b.nseq].ml = 0;
        recs[b.nseq].off = 0;
      }
      rep_out[0] = c.r0;
      rep_out[1] = c.r1;
      rep_out[2] = c.r2;
    }
    u32 lp = 0;
    const u32 nrec = ndone == b.nseq ? b.nseq + 1 : ndone;              // the whole block: its trailing literals too
    for (u32 i = 0; i < nrec && produced < n; i++) {
      const ZRec& r = recs[i];
      for (u32 k = 0; k < r.ll && produced < n; k++) { if (lp >= nl) return produced; out[produced++] = lits[lp++]; }
      if (!r.ml || produced >= n) continue;
      const i32 off = rep_resolve(r.off, rep);
      if (off <= 0 || (size_t)off > produced) return produced;
      for (u32 k = 0; k < r.ml && produced < n; k++) { out[produced] = out[produced - (size_t)off]; produced++; }
    }
    if (ndone == b.nseq) {
      i32 nr[3];
      for (int j = 0; j < 3; j++) nr[j] = rep_resolve(rep_out[j], rep);
      for (int j = 0; j < 3; j++) rep[j] = nr[j];
    }
  }
  return produced;
}
}  // namespace comet_zstd2
#endif
