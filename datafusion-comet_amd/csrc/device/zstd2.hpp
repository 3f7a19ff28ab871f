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
//   kernel A  one workgroup per block, two waves: wave 0 builds the Huffman table and decodes the literal streams (a lane per stream)
//             into the page's literal scratch; wave 1 builds the three FSE tables and one lane decodes the sequences into records
//             (ll, ml, offset).  Repeat offsets that reach back over the block's start stay SYMBOLIC (index into the block's
//             initial history, minus a delta), so blocks do not wait for each other.
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
constexpr int kEntThreads = 128;                // kernel A: threads 0 … 63 literals, 64 … 127 sequences
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

// ---- forward (little-endian) bit reader: table descriptions ----
struct FwdBits {
  const u8* p;
  u32 len, bit;
  bool over;
  ZS_FN void init(const u8* q, u32 n) { p = q; len = n; bit = 0; over = false; }
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
template <class NormPtr>
ZS_FN u32 fse_read_ncount(const u8* p, u32 avail, int max_sym /* inclusive */, int max_log, NormPtr norm, int& nsym, int& log) {
  FwdBits in;
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
// of decoding before its bytes are needed. ----
struct BackBits {
  const u8* s;
  const u8* fl;        // lowest readable address (the page's first byte)
  i32 bitpos;          // bits left: the next read takes bits [bitpos − n, bitpos) of the stream
  i32 wb;              // hi = bytes [wb − 8, wb), lo = [wb − 16, wb − 8), nx = [wb − 24, wb − 16)
  u64 hi, lo, nx;
  ZS_FN static u64 ld(const u8* p, const u8* floor_) {
    u64 v;
    if (p >= floor_) { __builtin_memcpy(&v, p, 8); return v; }
    const i64 d = floor_ - p;                               // bytes below the page: zero (never legitimately consumed)
    if (d >= 8) return 0;
    __builtin_memcpy(&v, floor_, 8);
    return v << (8 * d);
  }
  ZS_FN bool init(const u8* stream, u32 len, const u8* floor_) {
    s = stream;
    fl = floor_;
    bitpos = 0;
    wb = (i32)len;
    hi = lo = nx = 0;
    if (len == 0) return false;
    const u8 last = stream[len - 1];
    if (last == 0) return false;
    bitpos = 8 * ((i32)len - 1) + highbit(last);
    hi = ld(s + wb - 8, fl);
    lo = ld(s + wb - 16, fl);
    nx = ld(s + wb - 24, fl);
    return true;
  }
  ZS_FN u32 peek(int n) const {            // n ≤ 32; bits beyond the stream's beginning read as 0
    if (n == 0) return 0;
    if (bitpos < n) return bitpos > 0 ? peek(bitpos) << (n - bitpos) : 0u;
    const i32 sft = bitpos - 8 * (wb - 16) - n;             // lowest wanted bit within (hi:lo); the cursor stays inside hi: 32 < sft < 128
    const u64 v = sft >= 64 ? hi >> (sft - 64) : (hi << (64 - sft)) | (lo >> sft);
    return (u32)v & (n >= 32 ? 0xffffffffu : ((1u << n) - 1u));
  }
  ZS_FN void consume(int n) {
    bitpos -= n;
    if (bitpos - 8 * (wb - 16) <= 64) {
      hi = lo;
      lo = nx;
      wb -= 8;
      nx = ld(s + wb - 24, fl);
    }
  }
  ZS_FN u32 read(int n) { const u32 v = peek(n); consume(n); return v; }
};

// ---- Huffman ----
// Tree description at p: weights of symbols 0 … nw − 1 into w[] (the last symbol's weight is implied), → bytes consumed, 0 = malformed.
// wtab: scratch for the weights' own FSE table (64 entries), norm / next: scratch of ≥ 16 entries.
template <class WPtr, class TabPtr, class NormPtr, class NextPtr>
ZS_FN u32 huf_read_weights(const u8* p, u32 avail, const u8* floor_, WPtr w, int& nw, TabPtr wtab, NormPtr norm, NextPtr next) {
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
  BackBits b;
  if (!b.init(p + 1 + used, hb - used, floor_)) return 0;
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
// `count` symbols of one stream; the stream must end exactly
template <class TabPtr>
ZS_FN bool huf_decode_stream(const u8* stream, u32 len, const u8* floor_, TabPtr tab, int log, u8* out, u32 count) {
  BackBits b;
  if (!b.init(stream, len, floor_)) return false;
  u32 i = 0;
  for (; i + 4 <= count; i += 4) {
    u32 word = 0;
    for (int k = 0; k < 4; k++) {
      const u32 e = tab[b.peek(log)];
      b.consume((int)(e >> 8));
      word |= (e & 0xffu) << (8 * k);
    }
    __builtin_memcpy(out + i, &word, 4);
  }
  for (; i < count; i++) {
    const u32 e = tab[b.peek(log)];
    b.consume((int)(e >> 8));
    out[i] = (u8)e;
  }
  return b.bitpos == 0;
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
template <class TabPtr, class NormPtr, class NextPtr>
ZS_FN int seq_table(int kind, int mode, const u8* desc, u32 avail, TabPtr tab, NormPtr norm, NextPtr next) {
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

// ---- the sequences of one block: one lane.  Writes recs[0 … nseq] (ll, ml, off) — the last record holds the trailing literals —, the
// block's output size and its history on exit.  → ST_OK or an error code ----
template <class TabPtr, class CodePtr>
ZS_FN u32 seq_decode(const u8* bits, u32 bits_len, const u8* floor_, TabPtr tll, int lll, TabPtr tof, int lof, TabPtr tml, int lml, CodePtr llc, CodePtr mlc, u32 nseq,
                     u32 lit_regen, ZRec* recs, u32& out_size, i32* rep_out) {
  BackBits b;
  if (!b.init(bits, bits_len, floor_)) return ST_ERR_BITS;
  u32 sll = b.read(lll), sof = b.read(lof), sml = b.read(lml);
  i32 r0 = rep_symbolic(0), r1 = rep_symbolic(1), r2 = rep_symbolic(2);
  u64 sum_ll = 0, sum_ml = 0;
  for (u32 i = 0; i < nseq; i++) {
    const u32 ell = tll[sll], eof = tof[sof], eml = tml[sml];
    const u32 ofc = eof & 0xffu, mc = eml & 0xffu, lc = ell & 0xffu;
    if (ofc > 31u || mc > 52u || lc > 35u) return ST_ERR_SEQ;
    const u32 ov = (ofc ? (1u << ofc) : 1u) + b.read((int)ofc);
    const u32 me = mlc[mc], le = llc[lc];
    const u32 ml = (me & 0xffffffu) + b.read((int)(me >> 24));
    const u32 ll = (le & 0xffffffu) + b.read((int)(le >> 24));
    i32 off;
    if (ov > 3u) {
      off = (i32)(ov - 3u);
      if (off <= 0) return ST_ERR_OFFSET;
      r2 = r1; r1 = r0; r0 = off;
    } else {
      const u32 idx = ov - 1u + (ll == 0 ? 1u : 0u);
      if (idx == 0) off = r0;
      else {
        off = idx == 1 ? r1 : idx == 2 ? r2 : rep_minus_one(r0);
        if (idx > 1) r2 = r1;
        r1 = r0;
        r0 = off;
      }
    }
    recs[i].ll = ll;
    recs[i].ml = ml;
    recs[i].off = off;
    sum_ll += ll;
    sum_ml += ml;
    if (i + 1 < nseq) {
      sll = (ell >> 16) + b.read((int)((ell >> 8) & 0xffu));
      sml = (eml >> 16) + b.read((int)((eml >> 8) & 0xffu));
      sof = (eof >> 16) + b.read((int)((eof >> 8) & 0xffu));
    }
  }
  if (b.bitpos != 0) return ST_ERR_BITS;
  if (sum_ll > lit_regen || sum_ll + sum_ml > kBlockMax) return ST_ERR_LENGTH;
  recs[nseq].ll = lit_regen - (u32)sum_ll;
  recs[nseq].ml = 0;
  recs[nseq].off = 0;
  out_size = lit_regen + (u32)sum_ml;
  rep_out[0] = r0;
  rep_out[1] = r1;
  rep_out[2] = r2;
  return ST_OK;
}

// ---- kernel A: workgroup memory and phases ----
struct EntLds {
  u16 huf[1 << kHufLogMax];
  u32 fse[3][512];
  u32 wfse[64];
  u32 llc[36], mlc[53];
  u8 weights[256];
  i16 norm[2][64];           // [0]: the literal wave's, [1]: the sequence wave's
  u16 next[2][64];
  i32 huf_log, fse_log[3];
  u32 status;
};
// literal wave, thread 0: tree description → table
ZS_FN void ent_huf_table(ZS_LDS EntLds* L, const u8* src, const ZBlock& b) {
  if (b.type != BT_COMPRESSED || b.lit_type != LT_HUF) return;
  int nw = 0;
  const u32 used = huf_read_weights(src + b.huf_desc, b.huf_desc_len, src, L->weights, nw, L->wfse, L->norm[0], L->next[0]);
  const int log = used ? huf_build(L->weights, nw, L->huf) : 0;
  L->huf_log = log;
  if (!log) L->status = ST_ERR_HUF;
}
// sequence wave, thread 64: code tables and the block's three FSE tables
ZS_FN void ent_seq_tables(ZS_LDS EntLds* L, const u8* src, const ZBlock& b, u32 src_len) {
  for (u32 c = 0; c < 36; c++) L->llc[c] = ll_code_entry(c);
  for (u32 c = 0; c < 53; c++) L->mlc[c] = ml_code_entry(c);
  if (b.type != BT_COMPRESSED || b.nseq == 0) return;
  for (int k = 0; k < 3; k++) {
    const u32 d = b.tab_desc[k];
    const int log = seq_table(k, b.tab_mode[k], src + d, d < src_len ? src_len - d : 0, L->fse[k], L->norm[1], L->next[1]);
    L->fse_log[k] = log;
    if (log < 0) L->status = ST_ERR_FSE;
  }
}
// literal wave: the block's literals into the scratch.  t = thread within the wave (0 … 63)
ZS_FN void ent_literals(ZS_LDS EntLds* L, const u8* src, const ZBlock& b, u8* lits, int t) {
  u8* out = lits + b.lit_first;
  if (b.type == BT_RAW || (b.type == BT_COMPRESSED && b.lit_type == LT_RAW)) {
    const u8* from = src + (b.type == BT_RAW ? b.pos : b.lit_pos);
    for (u32 i = (u32)t; i < b.lit_regen; i += 64) out[i] = from[i];
    return;
  }
  if (b.type == BT_RLE || b.lit_type == LT_RLE) {
    const u8 v = src[b.type == BT_RLE ? b.pos : b.lit_pos];
    for (u32 i = (u32)t; i < b.lit_regen; i += 64) out[i] = v;
    return;
  }
  if (L->huf_log <= 0) return;
  const u8* s = src + b.lit_pos;
  if (b.lit_streams == 1) {
    if (t == 0 && !huf_decode_stream(s, b.lit_len, src, L->huf, L->huf_log, out, b.lit_regen)) L->status = ST_ERR_HUF;
    return;
  }
  if (t >= 4) return;
  if (b.lit_len < 10) { L->status = ST_ERR_HUF; return; }
  const u32 l1 = (u32)s[0] | ((u32)s[1] << 8), l2 = (u32)s[2] | ((u32)s[3] << 8), l3 = (u32)s[4] | ((u32)s[5] << 8);
  if (6 + l1 + l2 + l3 >= b.lit_len) { L->status = ST_ERR_HUF; return; }
  const u32 l4 = b.lit_len - 6 - l1 - l2 - l3;
  const u32 seg = (b.lit_regen + 3) / 4;
  if (3 * seg > b.lit_regen) { L->status = ST_ERR_HUF; return; }
  const u32 off = t == 0 ? 6u : t == 1 ? 6u + l1 : t == 2 ? 6u + l1 + l2 : 6u + l1 + l2 + l3;
  const u32 len = t == 0 ? l1 : t == 1 ? l2 : t == 2 ? l3 : l4;
  const u32 cnt = t < 3 ? seg : b.lit_regen - 3 * seg;
  if (!huf_decode_stream(s + off, len, src, L->huf, L->huf_log, out + (u32)t * seg, cnt)) L->status = ST_ERR_HUF;
}
// sequence wave, thread 64: the block's records
ZS_FN void ent_sequences(ZS_LDS EntLds* L, const u8* src, ZBlock* b, ZRec* recs) {
  ZRec* r = recs + b->rec_first;
  if (b->type != BT_COMPRESSED) {                         // a raw / RLE block is one run of literals
    r[0].ll = b->size;
    r[0].ml = 0;
    r[0].off = 0;
    b->out_size = b->size;
    for (int k = 0; k < 3; k++) b->rep_out[k] = rep_symbolic(k);
    return;
  }
  if (b->nseq == 0) {
    r[0].ll = b->lit_regen;
    r[0].ml = 0;
    r[0].off = 0;
    b->out_size = b->lit_regen;
    for (int k = 0; k < 3; k++) b->rep_out[k] = rep_symbolic(k);
    return;
  }
  if (L->status) return;
  u32 out_size = 0;
  i32 rep[3];
  const u32 rc = seq_decode(src + b->bits_pos, b->bits_len, src, L->fse[0], L->fse_log[0], L->fse[1], L->fse_log[1], L->fse[2], L->fse_log[2], L->llc, L->mlc, b->nseq,
                            b->lit_regen, r, out_size, rep);
  if (rc != ST_OK) { L->status = rc; return; }
  b->out_size = out_size;
  for (int k = 0; k < 3; k++) b->rep_out[k] = rep[k];
}

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

// ---- kernel C: one workgroup per block: output and literal positions of its records (prefix sums), offsets resolved and checked ----
struct ScanLds {
  u32 part_out[2][kScanThreads], part_lit[2][kScanThreads];      // double-buffered: scan step k reads [k & 1], writes [(k + 1) & 1]
  u32 carry_out, carry_lit;
  u32 status;
};
constexpr u32 kScanPer = 8;                                 // records per thread and tile
constexpr int kScanSteps = 8;                               // log2(kScanThreads)
// a tile = kScanThreads · kScanPer records.  Phases: sums (every thread its records) → kScanSteps steps of an inclusive scan over the
// threads' sums → write (positions from the thread's exclusive prefix) → carry (thread 0).  A barrier between any two.
ZS_FN void scan_tile_sums(ZS_LDS ScanLds* L, const ZRec* r, u32 n, u32 tile, int t) {
  const u32 a = tile + (u32)t * kScanPer;
  u32 so = 0, sl = 0;
  for (u32 k = 0; k < kScanPer; k++)
    if (a + k < n) { so += r[a + k].ll + r[a + k].ml; sl += r[a + k].ll; }
  L->part_out[0][t] = so;
  L->part_lit[0][t] = sl;
}
ZS_FN void scan_tile_step(ZS_LDS ScanLds* L, int step, int t) {
  const int from = step & 1, to = from ^ 1, d = 1 << step;
  L->part_out[to][t] = L->part_out[from][t] + (t >= d ? L->part_out[from][t - d] : 0u);
  L->part_lit[to][t] = L->part_lit[from][t] + (t >= d ? L->part_lit[from][t - d] : 0u);
}
ZS_FN void scan_tile_carry(ZS_LDS ScanLds* L) {
  L->carry_out += L->part_out[kScanSteps & 1][kScanThreads - 1];
  L->carry_lit += L->part_lit[kScanSteps & 1][kScanThreads - 1];
}
ZS_FN void scan_tile_write(ZS_LDS ScanLds* L, ZRec* r, u32 n, u32 tile, const ZBlock& b, int t) {
  const u32 a = tile + (u32)t * kScanPer;
  u32 o = L->carry_out + (t ? L->part_out[kScanSteps & 1][t - 1] : 0u), l = L->carry_lit + (t ? L->part_lit[kScanSteps & 1][t - 1] : 0u);
  for (u32 k = 0; k < kScanPer; k++) {
    if (a + k >= n) break;
    ZRec x = r[a + k];
    x.out_pos = o;
    x.lit_pos = l;
    if (x.ml) {
      const i32 off = rep_resolve(x.off, b.rep_in);
      if (off <= 0 || (u64)(u32)off > (u64)o + x.ll) L->status = ST_ERR_OFFSET;      // reaches before the page's first byte (no dictionaries here)
      x.off = off;
    }
    r[a + k] = x;
    o += x.ll + x.ml;
    l += x.ll;
  }
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
#ifndef __HIP_DEVICE_COMPILE__
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
  if (checksum) pos += 4;
  if (pos != len) return false;                             // another frame (or garbage) behind the first
  if (out_known > expect_out) return false;
  return true;
}
}  // namespace comet_zstd2
#endif
