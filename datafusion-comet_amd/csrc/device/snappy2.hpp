// Snappy (raw format) decompression of Parquet pages with EVERY lane of the GPU — the multi-kernel pipeline (round 3).
//
// The format is a serial chain of elements (tag byte: literal of n bytes follows / copy n bytes from `offset` back) whose start positions
// depend on every earlier element, and whose copies read what earlier elements wrote.  device/snappy_inflate.hpp walks that chain with ONE
// wave per page: a 1 MiB page of 8-byte decimals takes 18 ms, and a scan holds only a few hundred pages — the GPU idles.  Here both chains
// are broken with tables instead of being walked:
//
//   where elements START — a transfer function per 64-byte WINDOW of compressed bytes: for every position p of the window, "if an element
//   started at p, where does the chain leave the window, and how many output bytes and elements does it produce on the way?"  A backward
//   pass over the 64 positions gives all 64 answers at once (position p's answer is position (p + size)'s plus its own element), one
//   lane per window, no lane talking to another.  64 windows compose into a 4 KiB CHUNK's function (kernel A); one lane per page then hops
//   from chunk to chunk (kernel B: a few hundred dependent steps instead of a few hundred thousand); the chunks, now knowing where they
//   are entered, list their elements with their output positions (kernel C).
//
//   what copies READ — the standard compressors (C++ snappy behind pyarrow and snappy-java / parquet-mr, aircompressor) compress 64 KiB
//   of input at a time and never match across that boundary, so every 64 KiB FRAGMENT of output is self-contained.  Kernel D gives a
//   fragment to one workgroup: literal bytes go straight to the output and point at themselves, copy bytes point `offset` back, and
//   pointer jumping over the fragment's 65536 two-byte pointers in LDS (src[x] = src[src[x]], ~log2(chain depth) rounds — a column of
//   8-byte decimals chains every value to its predecessor, depth ≈ 8000) turns every pointer into the literal byte it finally names.
//
// A stream that does not have that shape (an element or a copy crossing a 64 KiB output boundary: legal, no known writer emits it) is
// flagged and decompressed by the one-wave kernel afterwards; a corrupt stream is an error naming the page.
//
// Every kernel is a sequence of PHASES: a phase is a plain function of the thread index, threads of a workgroup communicate only
// through workgroup memory between phases.  No wave intrinsics — so the same source runs on the host with the threads of a workgroup
// looped one after the other (tests/emu/snappy2_emu.cpp), where the CPU-only suite checks it against pyarrow-compressed pages.
#pragma once
#include <stdint.h>

#ifndef SN2_FN
#define SN2_FN inline
#endif
#ifndef SN2_LDS
#define SN2_LDS
#endif
#ifndef SN2_ATOMIC_OR_U32
#define SN2_ATOMIC_OR_U32(p, v) (*(p) |= (v))
#define SN2_ATOMIC_ADD_U32(p, v) (*(p) += (v))
#define SN2_ATOMIC_ADD_LDS(p, v) sn2_host_fetch_add((p), (v))
static inline uint32_t sn2_host_fetch_add(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
#endif

namespace comet_snappy2 {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

constexpr int kWin = 64;                       // compressed bytes per window
constexpr int kWinPad = 68;                    // a window in workgroup memory: its 64 bytes + the 4 bytes a tag at byte 63 may need (17 words: no bank conflicts)
constexpr int kWins = 64;                      // windows per chunk = threads of kernels A and C
constexpr int kChunk = kWin * kWins;           // 4096
constexpr int kFrag = 65536;                   // output bytes per fragment
constexpr int kExecThreads = 1024;             // threads of kernel D
constexpr u32 kNoEntry = 0xffffffffu;
constexpr int kBigLiteral = 256;               // literals from this length on are copied by the whole workgroup
constexpr int kBigQueue = 512;

// page status (u32 per page): 0 fine; kFallback: well-formed as far as seen but not fragment-shaped → the one-wave kernel decodes it;
// ≥ kErrBase: corrupt
enum { ST_OK = 0, ST_FALLBACK = 1, ST_ERR_PREAMBLE = 16, ST_ERR_TRUNCATED = 17, ST_ERR_BAD_COPY = 18, ST_ERR_LENGTH = 19 };

struct Page {                // one compressed page body, offsets into the column's byte buffer (PqInflate) + what the pipeline learns about it
  i64 src_off, dst_off;
  i32 src_len, dst_len;
  i32 body;                  // compressed offset of the first element (after the varint preamble)
  i32 chunk_first, nchunks;  // its chunks in the global chunk arrays; chunk c covers compressed bytes [body + c·4096, +4096)
  i32 frag_first, nfrags;    // its fragments in the global fragment numbering
  i64 elem_first;            // its elements in the global element array (after kernel B)
  u32 nelems;
  u32 pad;
};
struct ChunkFn { u32 exit, out, cnt; };          // entered at byte e of the chunk's first window: compressed bytes past the chunk's end where the chain lands, output bytes, elements
struct ChunkIn { u32 entry, out, elem; };        // kernel B's verdict per chunk: entry byte (chunk relative, kNoEntry = jumped over), output position and element index at entry
struct Elem { u32 out_pos, len, src, kind; };    // kind 0: literal, src = page-relative compressed offset of its bytes; 1: copy, src = offset back

// ---- the element at compressed position p of a page (bytes through get(p); positions ≥ the page's length read as 0) ----
template <class Get>
SN2_FN void parse_element(const Get& get, i64 p, u32& size, u32& outlen, u32& kind, u32& src) {
  const u32 tag = get(p);
  const u32 k = tag & 3u;
  if (k == 0) {
    u32 len = tag >> 2;
    u32 extra = 0;
    if (len >= 60) {
      extra = len - 59;                                  // 60 → 1 … 63 → 4 length bytes, little endian, value = length − 1
      len = 0;
      for (u32 b = 0; b < extra; b++) len |= (u32)get(p + 1 + b) << (8 * b);
    }
    outlen = len + 1;                                    // (a 4-byte length of 0xffffffff wraps to 0: caught by the length checks)
    size = 1 + extra + outlen;
    kind = 0;
    src = (u32)(p + 1 + extra);
  } else if (k == 1) {
    outlen = 4 + ((tag >> 2) & 7u);
    src = ((tag >> 5) << 8) | (u32)get(p + 1);
    size = 2;
    kind = 1;
  } else if (k == 2) {
    outlen = 1 + (tag >> 2);
    src = (u32)get(p + 1) | ((u32)get(p + 2) << 8);
    size = 3;
    kind = 1;
  } else {
    outlen = 1 + (tag >> 2);
    src = (u32)get(p + 1) | ((u32)get(p + 2) << 8) | ((u32)get(p + 3) << 16) | ((u32)get(p + 4) << 24);
    size = 5;
    kind = 1;
  }
}

// ---- workgroup memory of kernels A and C: one chunk ----
struct ChunkLds {
  u8 bytes[kWins * kWinPad];                   // window t at t · 68
  u32 exit[kWin][kWins];                       // [position][window]: transposed, lane t touches column t
  u32 out[kWin][kWins];
  u8 cnt[kWin][kWins];
  u32 went[kWins], wout[kWins], welem[kWins];  // kernel C: every window's true entry (kNoEntry = none), output position and element index
};

// phase 1 of A and C: thread t stages window t of chunk c (global → workgroup memory) and runs the backward pass over its 64 positions.
// Positions at or beyond the END of the stream hold no element: a chain that lands there has ended (well-formed streams land exactly on
// the end; kernel B checks that).  Differences are kept modulo 2^32: `exit` of a chain that ends inside the window is "negative".
SN2_FN void chunk_tables(SN2_LDS ChunkLds* L, const u8* src, i32 src_len, i64 chunk_pos, int t) {
  const i64 wpos = chunk_pos + (i64)t * kWin;
  SN2_LDS u8* wb = L->bytes + t * kWinPad;
  for (int b = 0; b < kWinPad; b++) wb[b] = (wpos + b < src_len) ? src[wpos + b] : (u8)0;
  const i64 lim64 = (i64)src_len - wpos;
  const u32 limit = lim64 <= 0 ? 0u : lim64 >= kWin ? (u32)kWin : (u32)lim64;       // window-relative end of the stream, clamped to the window
  auto get = [&](i64 p) -> u32 { return wb[p]; };                  // p = position inside the staged window (0 … 67)
  for (int p = kWin - 1; p >= 0; p--) {
    if ((u32)p >= limit) {                                         // nothing starts here: the chain has ended AT p
      L->exit[p][t] = (u32)p - (u32)kWin;
      L->out[p][t] = 0;
      L->cnt[p][t] = 0;
      continue;
    }
    u32 size, outlen, kind, s;
    parse_element(get, p, size, outlen, kind, s);
    (void)kind; (void)s;
    if (size > 0x40000000u) size = 0x40000000u;                    // a length no page can hold: lands far beyond the stream's end, kernel B reports it
    const u32 nxt = (u32)p + size;
    if (nxt >= limit) {                                            // leaves the window, or ends the stream inside it
      L->exit[p][t] = nxt - (u32)kWin;
      L->out[p][t] = outlen;
      L->cnt[p][t] = 1;
    } else {
      L->exit[p][t] = L->exit[nxt][t];
      L->out[p][t] = outlen + L->out[nxt][t];
      L->cnt[p][t] = (u8)(1 + L->cnt[nxt][t]);
    }
  }
}

// phase 2 of A: thread e composes the chunk's function for entry byte e of its first window; `limit` = chunk-relative end of the stream
SN2_FN ChunkFn chunk_compose(const SN2_LDS ChunkLds* L, int e, i64 limit) {
  u32 g = (u32)e, out = 0, cnt = 0;
  while (g < (u32)kChunk && (i64)g < limit) {
    const u32 w = g >> 6, o = g & 63u;
    out += L->out[o][w];
    cnt += L->cnt[o][w];
    g = (w + 1) * (u32)kWin + L->exit[o][w];                       // (modulo 2^32: a chain ending inside window w comes back to w·64 + its end)
  }
  ChunkFn f;
  f.exit = g - (u32)kChunk;                                        // "negative" when the stream ends inside the chunk; kernel B adds it back
  f.out = out;
  f.cnt = cnt;
  return f;
}

// ---- kernel B: one thread per page hops over its chunks ----
// Enters chunk c at byte `entry` (chunk relative).  entry < 64: the chunk's function answers; deeper (a long literal ended there): the
// elements up to the chunk's end are parsed one by one (rare: once per literal longer than a window that ends inside a chunk).
SN2_FN void page_chain(Page* pg, const u8* bytes, const ChunkFn* fns, ChunkIn* ins, u32* status, int page_index) {
  const u8* src = bytes + pg->src_off;
  const i32 src_len = pg->src_len;
  // preamble: varint uncompressed length
  u32 ulen = 0;
  int sh = 0, p = 0;
  for (;;) {
    if (p >= src_len || sh > 28) { status[page_index] = ST_ERR_PREAMBLE; return; }
    const u32 b = src[p++];
    ulen |= (b & 0x7fu) << sh;
    if (!(b & 0x80u)) break;
    sh += 7;
  }
  if (ulen != (u32)pg->dst_len || p != pg->body) { status[page_index] = ST_ERR_PREAMBLE; return; }
  auto get = [&](i64 q) -> u32 { return q < src_len ? src[q] : 0u; };
  i64 g = pg->body;                       // page-relative compressed position of the next element
  u64 out = 0, elem = 0;
  for (i32 c = 0; c < pg->nchunks; c++) {
    const i64 cpos = (i64)pg->body + (i64)c * kChunk;
    ChunkIn& in = ins[pg->chunk_first + c];
    if (g >= cpos + kChunk || g >= src_len) { in.entry = kNoEntry; in.out = (u32)out; in.elem = (u32)elem; continue; }
    const u32 entry = (u32)(g - cpos);
    in.entry = entry;
    in.out = (u32)out;
    in.elem = (u32)elem;
    if (entry < (u32)kWin) {
      const ChunkFn f = fns[(i64)(pg->chunk_first + c) * kWin + entry];
      out += f.out;
      elem += f.cnt;
      g = cpos + kChunk + (i64)(i32)f.exit;                          // f.exit is a 32-bit difference: negative = the stream ended inside the chunk
    } else {
      while (g < cpos + kChunk && g < src_len) {
        u32 size, outlen, kind, s;
        parse_element(get, g, size, outlen, kind, s);
        out += outlen;
        elem++;
        g += size;
      }
    }
    if (out > (u64)pg->dst_len) { status[page_index] = ST_ERR_LENGTH; return; }
  }
  if (g != src_len) { status[page_index] = ST_ERR_TRUNCATED; return; }
  if (out != (u64)pg->dst_len) { status[page_index] = ST_ERR_LENGTH; return; }
  pg->nelems = (u32)elem;
}

// ---- kernel C ----
// phase 2: one thread walks the chunk's windows from the true entry (64 dependent steps at most)
SN2_FN void chunk_window_entries(SN2_LDS ChunkLds* L, const ChunkIn& in, i64 limit) {
  for (int w = 0; w < kWins; w++) L->went[w] = kNoEntry;
  if (in.entry == kNoEntry) return;
  u32 g = in.entry, out = in.out, elem = in.elem;
  // an entry deeper than the first window is handled by the same walk: the tables cover every position of every window
  while (g < (u32)kChunk && (i64)g < limit) {
    const u32 w = g >> 6, o = g & 63u;
    L->went[w] = o;
    L->wout[w] = out;
    L->welem[w] = elem;
    out += L->out[o][w];
    elem += L->cnt[o][w];
    g = (w + 1) * (u32)kWin + L->exit[o][w];
  }
}
// phase 3: thread t lists the elements that start in window t
SN2_FN void chunk_emit(const SN2_LDS ChunkLds* L, i64 chunk_pos, i32 src_len, Elem* elems, int t) {
  if (L->went[t] == kNoEntry) return;
  const i64 lim64 = (i64)src_len - (chunk_pos + (i64)t * kWin);
  const u32 limit = lim64 <= 0 ? 0u : lim64 >= kWin ? (u32)kWin : (u32)lim64;
  const SN2_LDS u8* wb = L->bytes + t * kWinPad;
  auto get = [&](i64 p) -> u32 { return wb[p]; };
  u32 p = L->went[t], out = L->wout[t];
  Elem* dst = elems + L->welem[t];
  const i64 wpos = chunk_pos + (i64)t * kWin;
  while (p < limit) {
    u32 size, outlen, kind, s;
    parse_element(get, p, size, outlen, kind, s);
    Elem e;
    e.out_pos = out;
    e.len = outlen;
    e.kind = kind;
    e.src = kind == 0 ? (u32)(wpos + s) : s;            // literal: page-relative compressed offset of its bytes
    *dst++ = e;
    out += outlen;
    const u32 np = p + size;
    if (np < p) break;
    p = np;
  }
}

// ---- kernel D: one workgroup per 64 KiB fragment of output ----
struct ExecLds {
  u16 src[kFrag];                 // per output byte of the fragment: the fragment-relative byte it copies (itself: a literal byte)
  u32 big[kBigQueue][3];          // literals the whole workgroup copies: (fragment-relative output position, length, page-relative compressed offset)
  u32 nbig;
  u32 covered;                    // output bytes the fragment's elements account for
  u32 changed;
  u32 flags;                      // bit 0: not fragment-shaped (fall back); bit 1: bad copy (corrupt)
};

// the fragment's elements: [lo, hi) of the page's list (sorted by out_pos)
SN2_FN void frag_range(const Elem* elems, u32 nelems, u32 frag_out, u32 frag_end, u32& lo, u32& hi) {
  u32 a = 0, b = nelems;
  while (a < b) { const u32 m = (a + b) >> 1; if (elems[m].out_pos < frag_out) a = m + 1; else b = m; }
  lo = a;
  b = nelems;
  while (a < b) { const u32 m = (a + b) >> 1; if (elems[m].out_pos < frag_end) a = m + 1; else b = m; }
  hi = a;
}
// phase 1: elements → literal bytes to the output, pointers to workgroup memory
SN2_FN void frag_scatter(SN2_LDS ExecLds* L, const Elem* elems, u32 lo, u32 hi, u32 frag_out, u32 frag_end, const u8* src, u8* dst, int tid, int nthreads) {
  if (tid == 0 && (lo >= hi || elems[lo].out_pos != frag_out)) SN2_ATOMIC_OR_U32(&L->flags, 1u);      // an element straddles the fragment's start
  u32 mine = 0;
  for (u32 i = lo + (u32)tid; i < hi; i += (u32)nthreads) {
    const Elem e = elems[i];
    const u32 x = e.out_pos - frag_out;
    if (e.out_pos + e.len > frag_end || e.out_pos + e.len < e.out_pos) { SN2_ATOMIC_OR_U32(&L->flags, 1u); continue; }   // … or its end
    mine += e.len;
    if (e.kind == 0) {
      if (e.len >= (u32)kBigLiteral) {
        const u32 q = SN2_ATOMIC_ADD_LDS(&L->nbig, 1u);
        if (q < (u32)kBigQueue) { L->big[q][0] = x; L->big[q][1] = e.len; L->big[q][2] = e.src; }
        else {                                                         // (cannot happen: 65536 / 256 literals at most)
          for (u32 k = 0; k < e.len; k++) { dst[e.out_pos + k] = src[e.src + k]; L->src[x + k] = (u16)(x + k); }
        }
        continue;
      }
      for (u32 k = 0; k < e.len; k++) {
        dst[e.out_pos + k] = src[e.src + k];
        L->src[x + k] = (u16)(x + k);
      }
    } else {
      if (e.src == 0) { SN2_ATOMIC_OR_U32(&L->flags, 2u); continue; }
      if (e.src > x) { SN2_ATOMIC_OR_U32(&L->flags, e.src > e.out_pos ? 2u : 1u); continue; }    // before the page: corrupt; before the fragment: fall back
      for (u32 k = 0; k < e.len; k++) L->src[x + k] = (u16)(x + k - e.src);
    }
  }
  if (mine) SN2_ATOMIC_ADD_U32(&L->covered, mine);
}
// phase 2: the long literals, all threads together
SN2_FN void frag_big_literals(SN2_LDS ExecLds* L, u32 frag_out, const u8* src, u8* dst, int tid, int nthreads) {
  const u32 n = L->nbig < (u32)kBigQueue ? L->nbig : (u32)kBigQueue;
  for (u32 q = 0; q < n; q++) {
    const u32 x = L->big[q][0], len = L->big[q][1], s = L->big[q][2];
    for (u32 k = (u32)tid; k < len; k += (u32)nthreads) {
      dst[frag_out + x + k] = src[s + k];
      L->src[x + k] = (u16)(x + k);
    }
  }
}
// phase 3 (repeated): one round of pointer jumping over the thread's share of the fragment; → did anything move?
SN2_FN bool frag_jump(SN2_LDS ExecLds* L, u32 frag_len, int tid, int nthreads) {
  bool moved = false;
  for (u32 x = (u32)tid; x < frag_len; x += (u32)nthreads) {
    const u16 s = L->src[x];
    const u16 r = L->src[s];
    if (r != s) { L->src[x] = r; moved = true; }
  }
  return moved;
}
// phase 4: copy bytes take the literal byte their pointer has reached
SN2_FN void frag_resolve(const SN2_LDS ExecLds* L, u32 frag_out, u32 frag_len, u8* dst, int tid, int nthreads) {
  for (u32 x = (u32)tid; x < frag_len; x += (u32)nthreads) {
    const u16 s = L->src[x];
    if (s != (u16)x) dst[frag_out + x] = dst[frag_out + s];
  }
}

}  // namespace comet_snappy2

// ---- host side of the pipeline (plain C++; shared by csrc/snappy2.cpp and the host emulation) ----
#ifndef __HIP_DEVICE_COMPILE__
#include <vector>
namespace comet_snappy2 {
struct Plan {
  std::vector<Page> pages;
  std::vector<i32> chunk_page, frag_page;       // global chunk / fragment index → page
  i64 nchunks = 0, nfrags = 0;
};
// jobs: (src_off, dst_off, src_len, dst_len) per page; body[i] = length of page i's varint preamble (the host sees the compressed bytes)
inline Plan make_plan(const i64* src_off, const i64* dst_off, const i32* src_len, const i32* dst_len, const i32* body, int npages) {
  Plan pl;
  pl.pages.resize((size_t)npages);
  for (int i = 0; i < npages; i++) {
    Page& p = pl.pages[(size_t)i];
    p.src_off = src_off[i];
    p.dst_off = dst_off[i];
    p.src_len = src_len[i];
    p.dst_len = dst_len[i];
    p.body = body[i];
    const i64 rest = (i64)src_len[i] - body[i];
    p.chunk_first = (i32)pl.nchunks;
    p.nchunks = rest > 0 ? (i32)((rest + kChunk - 1) / kChunk) : 0;
    p.frag_first = (i32)pl.nfrags;
    p.nfrags = dst_len[i] > 0 ? (i32)(((i64)dst_len[i] + kFrag - 1) / kFrag) : 0;
    p.elem_first = 0;
    p.nelems = 0;
    p.pad = 0;
    for (i32 c = 0; c < p.nchunks; c++) pl.chunk_page.push_back(i);
    for (i32 f = 0; f < p.nfrags; f++) pl.frag_page.push_back(i);
    pl.nchunks += p.nchunks;
    pl.nfrags += p.nfrags;
  }
  return pl;
}
// length of the varint preamble of a raw snappy stream (0 = malformed)
inline i32 preamble_length(const u8* s, i32 n) {
  for (i32 p = 0; p < n && p < 5; p++)
    if (!(s[p] & 0x80u)) return p + 1;
  return 0;
}
}  // namespace comet_snappy2
#endif
