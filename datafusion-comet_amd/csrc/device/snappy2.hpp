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
#define SN2_ATOMIC_MIN_LDS(p, v) (*(p) = *(p) < (v) ? *(p) : (v))
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
    outlen = len + 1;
    size = outlen ? 1 + extra + outlen : 0xfffffff0u;    // a 4-byte length of 0xffffffff (+ 1 wraps to 0): no page holds it, lands beyond the stream
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

// The same element, sizes only, without a branch: tag byte + the four bytes behind it (kernels A and C parse 64 positions per lane).
SN2_FN void parse_sizes(u32 tag, u32 next4, u32& size, u32& outlen) {
  const u32 k = tag & 3u, n = tag >> 2;
  const u32 extra = n >= 60u ? n - 59u : 0u;                              // literal: 60 → 1 … 63 → 4 length bytes
  const u32 mask = extra >= 4u ? 0xffffffffu : ((1u << (8u * extra)) - 1u);
  const u32 lit_len = (extra ? (next4 & mask) : n) + 1u;                   // (0xffffffff + 1 wraps to 0: caught by the length checks)
  const u32 copy_len = k == 1u ? 4u + (n & 7u) : n + 1u;
  const u32 copy_size = k == 1u ? 2u : k == 2u ? 3u : 5u;
  outlen = k == 0u ? lit_len : copy_len;
  size = k == 0u ? (lit_len ? 1u + extra + lit_len : 0xfffffff0u) : copy_size;   // a length field of 2^32 − 1 (+ 1 wraps to 0): no page holds it
}

// ---- workgroup memory of kernels A and C: one chunk ----
struct ChunkLds {
  u8 bytes[kWins * kWinPad];                   // window t at t · 68
  // per (position, window), transposed so that lane t touches column t: where the chain from that position LANDS, counted from the window's
  // first byte (≥ 64: beyond the window; < 64: the stream ended there), its output bytes and its elements.  16 bits each — a chain that
  // holds a literal of 64 KiB or more does not fit: kBig in both, and whoever needs the numbers walks the window's elements (window_walk)
  u16 land[kWin][kWins];
  u16 out[kWin][kWins];
  u8 cnt[kWin][kWins];
  u32 went[kWins], wout[kWins], welem[kWins];  // kernel C: every window's true entry (kNoEntry = none), output position and element index
};
constexpr u16 kBig = 0xffffu;

// phase 1 of A and C: the workgroup stages the chunk's 4096 (+ 4) compressed bytes — aligned 16-byte loads, consecutive threads consecutive
// vectors — into the padded windows of workgroup memory (window w at w · 68; its bytes 64 … 67 repeat the next window's first four, for a tag at
// byte 63).  Bytes at or beyond the end of the stream are zero.
SN2_FN void chunk_stage(SN2_LDS ChunkLds* L, const u8* src, i32 src_len, i64 chunk_pos, int t) {
  const u8* start = src + chunk_pos;
  const int shift = (int)((uintptr_t)start & 15u);
  const u8* base = start - shift;                                  // 16-byte aligned (the page's bytes start 16-byte aligned)
  const int nvec = (kChunk + 4 + shift + 15) >> 4;
  const i64 readable = (((i64)src_len + 15) & ~(i64)15) - chunk_pos + shift;    // bytes from `base` that belong to the page's padded extent
  for (int v = t; v < nvec; v += kWins) {
    u8 b[16];
    if ((i64)v * 16 < readable) __builtin_memcpy(b, base + (i64)v * 16, 16);
    else __builtin_memset(b, 0, 16);
    for (int k = 0; k < 16; k++) {
      const int i = v * 16 + k - shift;                            // chunk-relative byte
      if (i < 0 || i >= kChunk + 4) continue;
      const u8 x = (chunk_pos + i < src_len) ? b[k] : (u8)0;
      const int w = i >> 6, o = i & 63;
      if (w < kWins) L->bytes[w * kWinPad + o] = x;
      if (o < 4 && w > 0) L->bytes[(w - 1) * kWinPad + kWin + o] = x;
    }
  }
}
// phase 2 of A and C: thread t runs the backward pass over the 64 positions of window t.
// Positions at or beyond the END of the stream hold no element: a chain that lands there has ended (well-formed streams land exactly on
// the end; kernel B checks that).
SN2_FN void chunk_tables(SN2_LDS ChunkLds* L, i32 src_len, i64 chunk_pos, int t) {
  const i64 wpos = chunk_pos + (i64)t * kWin;
  SN2_LDS u8* wb = L->bytes + t * kWinPad;
  const i64 lim64 = (i64)src_len - wpos;
  const u32 limit = lim64 <= 0 ? 0u : lim64 >= kWin ? (u32)kWin : (u32)lim64;       // window-relative end of the stream, clamped to the window
  // pass 1: every position on its own — the element that would start there.  The window's 68 bytes sit in 17 registers and every position
  // is a compile-time constant of the unrolled loop: tag and length bytes are shifts of registers, the parse is branch-free.
  u32 wv[kWinPad / 4];
#pragma unroll
  for (int k = 0; k < kWinPad / 4; k++) {
    const SN2_LDS u8* q = wb + 4 * k;
    wv[k] = (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16) | ((u32)q[3] << 24);
  }
  u8 nx[kWin];                                 // the successor inside the window, or 0xff: held in registers (the loops are unrolled)
#pragma unroll
  for (int p = 0; p < kWin; p++) {
    const u64 five = ((((u64)wv[(p >> 2) + 1]) << 32) | (u64)wv[p >> 2]) >> ((p & 3) * 8);      // bytes p … p + 4
    u32 size, outlen;
    parse_sizes((u32)five & 0xffu, (u32)(five >> 8), size, outlen);
    const bool none = (u32)p >= limit;                             // nothing starts at or beyond the end of the stream: the chain has ended AT p
    const u32 landing = none ? (u32)p : (size >= 0xfff0u ? 0xffffu : (u32)p + size);
    const bool big = !none && (landing >= 0xffffu || outlen >= 0xffffu);
    L->land[p][t] = big ? kBig : (u16)landing;
    L->out[p][t] = none ? (u16)0 : big ? kBig : (u16)outlen;
    L->cnt[p][t] = none ? (u8)0 : (u8)1;
    nx[p] = (!none && !big && landing < limit) ? (u8)landing : (u8)0xff;
  }
  // pass 2: backwards, an element that ends inside the window (and inside the stream) continues with its successor's answer
#pragma unroll
  for (int p = kWin - 1; p >= 0; p--) {
    if (nx[p] != 0xff) {
      const u32 n = nx[p];
      const u16 nl = L->land[n][t], no = L->out[n][t];
      const u32 so = (u32)L->out[p][t] + (u32)no;
      const bool big = nl == kBig || no == kBig || so >= 0xffffu;
      L->land[p][t] = big ? kBig : nl;
      L->out[p][t] = big ? kBig : (u16)so;
      L->cnt[p][t] = (u8)(1 + L->cnt[n][t]);
    }
  }
}
// the exact numbers of a chain the tables could not hold: from position o of window w, element by element, to where it leaves the window
SN2_FN void window_walk(const SN2_LDS ChunkLds* L, u32 w, u32 o, i64 limit_chunk, u64& landing, u32& out, u32& cnt) {
  const SN2_LDS u8* wb = L->bytes + w * kWinPad;
  const i64 lim64 = limit_chunk - (i64)w * kWin;
  const u32 limit = lim64 <= 0 ? 0u : lim64 >= kWin ? (u32)kWin : (u32)lim64;
  u64 p = o;
  out = 0;
  cnt = 0;
  while (p < limit) {
    const u32 tag = wb[p];
    const u32 next4 = (u32)wb[p + 1] | ((u32)wb[p + 2] << 8) | ((u32)wb[p + 3] << 16) | ((u32)wb[p + 4] << 24);
    u32 size, outlen;
    parse_sizes(tag, next4, size, outlen);
    out += outlen;
    cnt++;
    p += size;
  }
  landing = p;
}
// one step of a walk over the chunk's windows: from chunk-relative byte g to where its chain leaves window g / 64
SN2_FN void chunk_step(const SN2_LDS ChunkLds* L, u64& g, u32& out, u32& cnt, i64 limit_chunk) {
  const u32 w = (u32)(g >> 6), o = (u32)g & 63u;
  const u16 l = L->land[o][w];
  if (l != kBig) {
    out += L->out[o][w];
    cnt += L->cnt[o][w];
    g = (u64)w * kWin + l;
  } else {
    u64 landing;
    u32 wo, wc;
    window_walk(L, w, o, limit_chunk, landing, wo, wc);
    out += wo;
    cnt += wc;
    g = (u64)w * kWin + landing;
  }
}

// phase 3 of A: thread e composes the chunk's function for entry byte e of its first window; `limit` = chunk-relative end of the stream
SN2_FN ChunkFn chunk_compose(const SN2_LDS ChunkLds* L, int e, i64 limit) {
  u64 g = (u64)e;
  u32 out = 0, cnt = 0;
  while (g < (u64)kChunk && (i64)g < limit) chunk_step(L, g, out, cnt, limit);
  ChunkFn f;
  const u64 over = g - (u64)kChunk;                                // "negative" (mod 2^64) when the stream ends inside the chunk
  f.exit = g >= (u64)kChunk && over > 0x7fffffffull ? 0x7fffffffu : (u32)over;      // beyond any page: kernel B reports the truncation
  f.out = out;
  f.cnt = cnt;
  return f;
}

// ---- kernel B: one thread per page hops over its chunks ----
// Enters chunk c at byte `entry` (chunk relative).  entry < 64: the chunk's function answers; deeper (a long literal ended there): the
// elements up to the chunk's end are parsed one by one (rare: once per literal longer than a window that ends inside a chunk).
SN2_FN void page_chain(Page* pg, const u8* bytes, const ChunkFn* fns, ChunkIn* ins, i32* frag_chunk, u32* status, int page_index) {
  if (status[page_index] != ST_OK) return;        // routed to the one-wave kernel by the host (incompressible pages: one long literal per block)
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
  i32 nf = 0;                             // fragments whose first element's chunk is known: fragment f starts in the chunk where `out` reaches f · 65536
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
    while (nf < pg->nfrags && (u64)nf * kFrag < out) frag_chunk[pg->frag_first + nf++] = c;     // the element that starts fragment nf starts inside chunk c
  }
  if (g != src_len) { status[page_index] = ST_ERR_TRUNCATED; return; }
  if (out != (u64)pg->dst_len) { status[page_index] = ST_ERR_LENGTH; return; }
  while (nf < pg->nfrags) frag_chunk[pg->frag_first + nf++] = pg->nchunks > 0 ? pg->nchunks - 1 : 0;
  pg->nelems = (u32)elem;
}

// ---- kernel C ----
// phase 2: one thread walks the chunk's windows from the true entry (64 dependent steps at most)
SN2_FN void chunk_window_entries(SN2_LDS ChunkLds* L, const ChunkIn& in, i64 limit) {
  for (int w = 0; w < kWins; w++) L->went[w] = kNoEntry;
  if (in.entry == kNoEntry) return;
  u64 g = in.entry;
  u32 out = in.out, elem = in.elem;
  // an entry deeper than the first window is handled by the same walk: the tables cover every position of every window
  while (g < (u64)kChunk && (i64)g < limit) {
    const u32 w = (u32)(g >> 6);
    L->went[w] = (u32)g & 63u;
    L->wout[w] = out;
    L->welem[w] = elem;
    chunk_step(L, g, out, elem, limit);
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
  u32 lo, hi;                     // the fragment's elements
  u32 nbig;
  u32 covered;                    // output bytes the fragment's elements account for
  u32 changed;
  u32 flags;                      // bit 0: not fragment-shaped (fall back); bit 1: bad copy (corrupt); bit 2: holds copies
};

// the fragment's elements: [lo, hi) of the page's list (sorted by out_pos).  Kernel B has named the chunk in which the fragment's first
// element starts (and the next fragment's): the elements of those chunks are looked at by all threads at once — phase 0a clears, 0b
// searches (workgroup-memory atomicMin), the caller reads L->lo / L->hi after the next barrier.
SN2_FN void frag_range_search(SN2_LDS ExecLds* L, const Elem* elems, u32 nelems, const ChunkIn* page_ins, i32 nchunks, i32 c_lo, i32 c_hi /* -1: last fragment */,
                              u32 frag_out, u32 frag_end, int tid, int nthreads) {
  auto first_at_or_after = [&](i32 c, u32 bound, SN2_LDS u32* slot) {
    // elements that start in chunk c: [page_ins[c].elem, page_ins[c + 1].elem) — the answer is one of them, or the element right behind them
    const u32 a = page_ins[c].elem, b = c + 1 < nchunks ? page_ins[c + 1].elem : nelems;
    for (u32 i = a + (u32)tid; i < b; i += (u32)nthreads)
      if (elems[i].out_pos >= bound) { SN2_ATOMIC_MIN_LDS(slot, i); break; }
    if (tid == 0) SN2_ATOMIC_MIN_LDS(slot, b);
  };
  first_at_or_after(c_lo, frag_out, &L->lo);
  if (c_hi >= 0) first_at_or_after(c_hi, frag_end, &L->hi);
  else if (tid == 0) L->hi = nelems;
}
// phase 1: elements → literal bytes to the output, pointers to workgroup memory.  Four elements per thread and step: their records, then
// the (short) literals' bytes, are loaded together — a loop that loads, waits and stores per element spends its time waiting.
SN2_FN void frag_scatter(SN2_LDS ExecLds* L, const Elem* elems, u32 lo, u32 hi, u32 frag_out, u32 frag_end, const u8* src, u8* dst, int tid, int nthreads) {
  if (tid == 0 && (lo >= hi || elems[lo].out_pos != frag_out)) SN2_ATOMIC_OR_U32(&L->flags, 1u);      // an element straddles the fragment's start
  u32 mine = 0;
  bool copies = false;
  constexpr int B = 4, S = 8;                      // elements per step; literal bytes fetched ahead per element
  for (u32 i0 = lo + (u32)tid; i0 < hi; i0 += (u32)B * (u32)nthreads) {
    Elem e[B];
    bool ok[B];
    for (int u = 0; u < B; u++) {
      const u32 i = i0 + (u32)u * (u32)nthreads;
      ok[u] = i < hi;
      if (ok[u]) e[u] = elems[i];
      else { e[u].out_pos = frag_out; e[u].len = 0; e[u].src = 1; e[u].kind = 1; }
    }
    u8 lit[B][S];
    for (int u = 0; u < B; u++) {
      if (ok[u] && (e[u].out_pos + e[u].len > frag_end || e[u].out_pos + e[u].len < e[u].out_pos)) {         // … or its end
        SN2_ATOMIC_OR_U32(&L->flags, 1u);
        ok[u] = false;
      }
      for (int k = 0; k < S; k++) lit[u][k] = (ok[u] && e[u].kind == 0 && (u32)k < e[u].len && e[u].len < (u32)kBigLiteral) ? src[e[u].src + k] : (u8)0;
    }
    for (int u = 0; u < B; u++) {
      if (!ok[u]) continue;
      const u32 x = e[u].out_pos - frag_out, len = e[u].len;
      mine += len;
      if (e[u].kind == 0) {
        if (len >= (u32)kBigLiteral) {
          const u32 q = SN2_ATOMIC_ADD_LDS(&L->nbig, 1u);
          if (q < (u32)kBigQueue) { L->big[q][0] = x; L->big[q][1] = len; L->big[q][2] = e[u].src; }
          else                                                          // (cannot happen: 65536 / 256 literals at most)
            for (u32 k = 0; k < len; k++) { dst[e[u].out_pos + k] = src[e[u].src + k]; L->src[x + k] = (u16)(x + k); }
          continue;
        }
        for (u32 k = 0; k < len && k < (u32)S; k++) { dst[e[u].out_pos + k] = lit[u][k]; L->src[x + k] = (u16)(x + k); }
        for (u32 k = (u32)S; k < len; k++) { dst[e[u].out_pos + k] = src[e[u].src + k]; L->src[x + k] = (u16)(x + k); }
      } else {
        if (e[u].src == 0) { SN2_ATOMIC_OR_U32(&L->flags, 2u); continue; }
        if (e[u].src > x) { SN2_ATOMIC_OR_U32(&L->flags, e[u].src > e[u].out_pos ? 2u : 1u); continue; }    // before the page: corrupt; before the fragment: fall back
        for (u32 k = 0; k < len; k++) L->src[x + k] = (u16)(x + k - e[u].src);
        copies = true;
      }
    }
  }
  if (mine) SN2_ATOMIC_ADD_U32(&L->covered, mine);
  if (copies) SN2_ATOMIC_OR_U32(&L->flags, 4u);                    // bit 2: the fragment holds copies (else there is nothing to resolve)
}
// phase 2: the long literals, all threads together (eight bytes in flight per thread: a byte loop waits for every load)
SN2_FN void frag_big_literals(SN2_LDS ExecLds* L, u32 frag_out, const u8* src, u8* dst, int tid, int nthreads) {
  const u32 n = L->nbig < (u32)kBigQueue ? L->nbig : (u32)kBigQueue;
  for (u32 q = 0; q < n; q++) {
    const u32 x = L->big[q][0], len = L->big[q][1], s = L->big[q][2];
    for (u32 k0 = (u32)tid; k0 < len; k0 += 8u * (u32)nthreads) {
      u8 b[8];
      for (int u = 0; u < 8; u++) { const u32 k = k0 + (u32)u * (u32)nthreads; b[u] = k < len ? src[s + k] : (u8)0; }
      for (int u = 0; u < 8; u++) {
        const u32 k = k0 + (u32)u * (u32)nthreads;
        if (k < len) { dst[frag_out + x + k] = b[u]; L->src[x + k] = (u16)(x + k); }
      }
    }
  }
}
// phase 3 (repeated): one round of pointer jumping over the thread's share of the fragment; → did anything move?
SN2_FN bool frag_jump(SN2_LDS ExecLds* L, u32 frag_len, int tid, int nthreads) {
  bool moved = false;
  for (u32 x0 = (u32)tid; x0 < frag_len; x0 += 8u * (u32)nthreads) {
    u16 s[8], r[8];
    for (int u = 0; u < 8; u++) { const u32 x = x0 + (u32)u * (u32)nthreads; s[u] = x < frag_len ? L->src[x] : (u16)0; }
    for (int u = 0; u < 8; u++) r[u] = L->src[s[u]];
    for (int u = 0; u < 8; u++) r[u] = L->src[r[u]];               // two hops per round: the distance to the root shrinks four-fold, half the rounds
    for (int u = 0; u < 8; u++) {
      const u32 x = x0 + (u32)u * (u32)nthreads;
      if (x < frag_len && r[u] != s[u]) { L->src[x] = r[u]; moved = true; }
    }
  }
  return moved;
}
// phase 4: every byte takes the literal byte its pointer has reached (a literal byte: itself) — four consecutive bytes per thread and step,
// gathered with all loads in flight and stored as one word
SN2_FN void frag_resolve(const SN2_LDS ExecLds* L, u32 frag_out, u32 frag_len, u8* dst, int tid, int nthreads) {
  u8* fd = dst + frag_out;
  const u32 head = (u32)((4u - ((uintptr_t)fd & 3u)) & 3u) < frag_len ? (u32)((4u - ((uintptr_t)fd & 3u)) & 3u) : frag_len;   // bytes before the first aligned word
  const u32 nwords = (frag_len - head) >> 2, tail = head + (nwords << 2);
  for (u32 x = (u32)tid; x < head; x += (u32)nthreads) { const u16 s = L->src[x]; if (s != (u16)x) fd[x] = fd[s]; }
  for (u32 x = tail + (u32)tid; x < frag_len; x += (u32)nthreads) { const u16 s = L->src[x]; if (s != (u16)x) fd[x] = fd[s]; }
  for (u32 w0 = (u32)tid; w0 < nwords; w0 += 4u * (u32)nthreads) {
    u32 v[4];
    for (int u = 0; u < 4; u++) {
      const u32 w = w0 + (u32)u * (u32)nthreads;
      v[u] = 0;
      if (w < nwords) {
        const u32 x = head + (w << 2);
        const u32 b0 = fd[L->src[x]], b1 = fd[L->src[x + 1]], b2 = fd[L->src[x + 2]], b3 = fd[L->src[x + 3]];
        v[u] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
      }
    }
    for (int u = 0; u < 4; u++) {
      const u32 w = w0 + (u32)u * (u32)nthreads;
      if (w < nwords) *(u32*)(fd + head + (w << 2)) = v[u];
    }
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
