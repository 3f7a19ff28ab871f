// Snappy (raw format) decompression of ONE Parquet page by ONE 64-lane wavefront.
//
// The format (google/snappy format_description.txt) is a serial chain of elements — a tag byte says "literal of n bytes follows" or "copy
// n bytes from `offset` back" — whose start positions depend on every earlier element, and Parquet's PLAIN pages of 8-byte values compress
// to ~4 output bytes per element, so one lane walking the chain decodes a 1 MiB page in ~15 ms.  This kernel breaks the chain per WINDOW of
// 64 compressed bytes:
//   1. every lane decodes the element that WOULD start at its byte (tag, length, offset, compressed size) — 64 speculative parses at once;
//   2. the true starts are the orbit of lane 0 under lane → lane + size, followed with scalar v_readlane steps (≈10 cycles per element);
//   3. a DPP prefix sum over the true starts' lengths gives every element its output position;
//   4. all literals and every copy whose source lies below the bytes already written run AT ONCE, each on its own lane, 8 bytes per pass;
//      a copy that reads bytes produced inside the window waits a round (rounds = dependency depth, 1–2 for columnar data).
// The last 64 KiB of output live in an LDS ring (snappy offsets stay below 64 KiB in practice: compressors work in 64 KiB blocks), so copy
// sources never touch HBM; the ring drains to HBM in aligned 1 KiB stores.  Offsets beyond the ring read the flushed output back with
// device-coherent loads.  Compressed input is prefetched 1 KiB ahead into a 2 KiB LDS ring.
//
// The body is written against a small wave interface W so the SAME source compiles for gfx950 (snappy_kernels.hip: DPP / v_readlane /
// LDS) and for a 64-thread host emulation (tests/emu/snappy_emu.cpp) that the CPU-only suite runs against pyarrow-compressed streams.
#pragma once
#include <stdint.h>

namespace comet_snappy {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int kRing = 65536, kRingMask = kRing - 1;
constexpr int kIn = 2048, kInMask = kIn - 1;
constexpr int kHalf = 1024;                 // refill / flush granule
constexpr u32 kNearMax = kRing - 8192;      // copies reaching further back than this read the flushed output instead of the ring
constexpr u32 kStopBig = 0x100, kStopEnd = 0x200;

enum { OK = 0, ERR_PREAMBLE = 1, ERR_TRUNCATED = 2, ERR_BAD_COPY = 3, ERR_OVERRUN = 4, ERR_SHORT = 5 };

typedef u32 V4 __attribute__((vector_size(16)));   // one 16-byte load / store (gcc and clang)

struct Lds {
  alignas(16) u8 ring[kRing];
  alignas(16) u8 in[kIn];
};

#ifndef SNAPPY_TICK
#define SNAPPY_TICK(i)
#endif
#ifndef SNAPPY_FN
#define SNAPPY_FN inline
#endif
#ifndef SNAPPY_LDS
#define SNAPPY_LDS
#endif

// src: any alignment, readable up to the next multiple of 16 past src_len.  dst: 16-byte aligned.  Returns an ERR_* code (uniform).
template <class W>
SNAPPY_FN int inflate_page(W& w, SNAPPY_LDS Lds* lds, const u8* src, int src_len, u8* dst, int dst_len) {
  const int lane = w.lane();
  const int src_len16 = (src_len + 15) & ~15;
  SNAPPY_LDS u8* ring = lds->ring;
  SNAPPY_LDS u8* in = lds->in;

  auto load_half = [&](int base) {
    V4 v = {0, 0, 0, 0};
    const int a = base + lane * 16;
    if (a < src_len16) __builtin_memcpy(&v, src + a, 16);      // any alignment: a body read in place sits where the file put it
    return v;
  };
  int in_hi = 0;                // compressed bytes [in_hi - kIn, in_hi) are in `in`
  V4 pf = load_half(0);
  auto refill = [&]() {
    *(SNAPPY_LDS V4*)(in + ((in_hi + lane * 16) & kInMask)) = pf;
    in_hi += kHalf;
    pf = load_half(in_hi);
    w.lds_sync();
  };
  refill();
  refill();

  // preamble: varint uncompressed length
  int p = 0;
  {
    u32 ulen = 0;
    int sh = 0;
    for (;;) {
      if (p >= src_len || sh > 28) return ERR_PREAMBLE;
      const u32 b = in[p & kInMask];
      p++;
      ulen |= (b & 0x7f) << sh;
      if (!(b & 0x80)) break;
      sh += 7;
    }
    if (ulen != (u32)dst_len) return ERR_PREAMBLE;
  }

  int o = 0, flushed = 0;       // output bytes [0, o) are final; [0, flushed) are in dst
  bool bad = false;             // this lane saw a copy with an impossible offset
  auto flush = [&]() {
    while (flushed + kHalf <= o) {
      const V4 v = *(SNAPPY_LDS const V4*)(ring + ((flushed + lane * 16) & kRingMask));
      *(V4*)(dst + flushed + lane * 16) = v;
      flushed += kHalf;
    }
  };

  while (p < src_len) {
    if (p >= in_hi) {           // a long literal jumped past the staged input: restart the prefetch at p
      in_hi = p & ~(kHalf - 1);
      pf = load_half(in_hi);
      refill();
      refill();
    }
    SNAPPY_TICK(0);
    while (in_hi < p + 128 && in_hi < src_len) refill();
    SNAPPY_TICK(1);

    // 1. speculative parse: the element that would start at byte p + lane
    const int a = p + lane;
    const u32 tag = in[a & kInMask];
    const u32 x = (u32)in[(a + 1) & kInMask] | ((u32)in[(a + 2) & kInMask] << 8) | ((u32)in[(a + 3) & kInMask] << 16) | ((u32)in[(a + 4) & kInMask] << 24);
    const u32 kind = tag & 3;
    u32 len, off = 0, hdr;
    bool big = false;
    if (kind == 0) {
      const u32 n = tag >> 2;
      if (n < 60) {
        len = n + 1;
        hdr = 1;
      } else {
        const u32 nb = n - 59;
        const u32 m = nb == 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1);
        len = (x & m);
        len = len >= 0x3fffffffu ? 0x3fffffffu : len + 1;
        hdr = 1 + nb;
        big = true;
      }
    } else if (kind == 1) {
      len = ((tag >> 2) & 7) + 4;
      off = ((tag >> 5) << 8) | (x & 0xff);
      hdr = 2;
    } else if (kind == 2) {
      len = (tag >> 2) + 1;
      off = x & 0xffff;
      hdr = 3;
    } else {
      len = (tag >> 2) + 1;
      off = x;
      hdr = 5;
    }
    // lane → the lane (≥ 64: past the window) where the element after this one starts; two out-of-band values end the walk:
    // kStopBig + lane = "a long literal starts here", kStopEnd + lane = "this byte is past the end of the stream"
    const u32 esize = hdr + (kind == 0 ? len : 0);
    const u32 nx = a >= src_len ? kStopEnd + (u32)lane : big ? kStopBig + (u32)lane : (u32)lane + esize;

    SNAPPY_TICK(2);
    // 2. the true element starts: follow lane → nx from lane 0.  A v_readlane → scalar → next v_readlane step costs ~60 cycles however
    //    little it does, so the walk takes FOUR elements per step: nx∘nx, nx∘nx∘nx and nx⁴ come from three lane gathers (exit codes stick),
    //    and the four readlanes of a step are independent of each other.
    const u32 c1 = nx;
    const u32 g2 = w.gather(c1, c1 & 63);
    const u32 c2 = c1 < 64 ? g2 : c1;
    const u32 g3 = w.gather(c1, c2 & 63), g4 = w.gather(c2, c2 & 63);
    const u32 c3 = c2 < 64 ? g3 : c2;
    const u32 c4 = c2 < 64 ? g4 : c2;
    u64 starts = 0;
    u32 qq = 0;
    while (qq < 64) {
      const u32 n1 = w.readlane(c1, (int)qq), n2 = w.readlane(c2, (int)qq), n3 = w.readlane(c3, (int)qq), n4 = w.readlane(c4, (int)qq);
      starts |= ((u64)1 << qq) | (n1 < 64 ? (u64)1 << n1 : 0) | (n2 < 64 ? (u64)1 << n2 : 0) | (n3 < 64 ? (u64)1 << n3 : 0);
      qq = n4;
    }
    SNAPPY_TICK(3);
    bool stop_big = false;
    if (qq >= kStopEnd) {          // walked onto the end of the stream (exactly, or an element ran past it: checked below)
      qq -= kStopEnd;
      starts &= ~((u64)1 << qq);
    } else if (qq >= kStopBig) {
      qq -= kStopBig;
      starts &= ~((u64)1 << qq);
      stop_big = true;
    }
    const int q = (int)qq;
    if (stop_big && q == 0) {
      // a literal of more than 60 bytes: the whole wave copies it, a flush granule at a time
      const u32 blen = w.readlane(len, 0);
      int sp = p + (int)w.readlane(hdr, 0);
      if ((u64)sp + blen > (u64)src_len) return ERR_TRUNCATED;
      if ((u64)o + blen > (u64)dst_len) return ERR_OVERRUN;
      u32 rem = blen;
      while (rem) {
        const u32 room = (u32)(kHalf - (o & (kHalf - 1)));
        const u32 n = rem < room ? rem : room;
        for (u32 k = (u32)lane; k < n; k += 64) ring[(o + (int)k) & kRingMask] = src[sp + (int)k];
        w.lds_sync();
        o += (int)n;
        sp += (int)n;
        rem -= n;
        flush();
      }
      p = sp;
      continue;
    }
    if (p + q > src_len) return ERR_TRUNCATED;

    // 3. output positions
    const bool is_start = (starts >> lane) & 1;
    const u32 mylen = is_start ? len : 0;
    const u32 incl = w.incl_scan_add(mylen);
    const u32 total = w.readlane(incl, 63);
    const u32 opos = (u32)o + incl - mylen;
    if ((u64)o + total > (u64)dst_len) return ERR_OVERRUN;
    const bool is_copy = is_start && kind != 0;
    if (is_copy && (off == 0 || off > opos)) {       // reported at the end of the page; until then the element copies harmless bytes
      bad = true;
      off = 1;
    }
    const bool far = is_copy && off > kNearMax;
    const u32 s = opos - off;                        // copies: first source byte
    const u32 need = s + (len < off ? len : off);    // copies: one past the last source byte that is not the copy's own output

#ifdef SNAPPY_TRACE
    SNAPPY_TRACE(lane, p, q, o, total, starts, is_start, kind, len, off, opos);
#endif
    SNAPPY_TICK(4);
    // 4. the copies.  One pass moves up to 8 bytes of every element it is given.  Literals (from the input ring) and copies (from the
    //    history ring) share one path: both sources are LDS bytes at base + ((pos + k) & mask); the 8 loads are unconditional (any LDS
    //    byte may be read), so they issue back to back and the wave waits for LDS once per pass.  A copy that overlaps its own output
    //    (offset < length) repeats its first `offset` source bytes.
    const bool lit = kind == 0;
    const bool overlap = !lit && off < len;
    SNAPPY_LDS const u8* sbase = lit ? (SNAPPY_LDS const u8*)in : (SNAPPY_LDS const u8*)ring;
    const u32 smask = lit ? (u32)kInMask : (u32)kRingMask;
    const u32 spos = lit ? (u32)(a + 1) : s;
    auto pass8 = [&](bool act, u32 c, u32& cmod) {
      if (act) {
        const u32 n = len - c;
        u32 i0 = c, i1 = c + 1, i2 = c + 2, i3 = c + 3, i4 = c + 4, i5 = c + 5, i6 = c + 6, i7 = c + 7;
        if (overlap) {
          u32 x = cmod;
          i0 = x; x = x + 1 == off ? 0 : x + 1;
          i1 = x; x = x + 1 == off ? 0 : x + 1;
          i2 = x; x = x + 1 == off ? 0 : x + 1;
          i3 = x; x = x + 1 == off ? 0 : x + 1;
          i4 = x; x = x + 1 == off ? 0 : x + 1;
          i5 = x; x = x + 1 == off ? 0 : x + 1;
          i6 = x; x = x + 1 == off ? 0 : x + 1;
          i7 = x; x = x + 1 == off ? 0 : x + 1;
          cmod = x;
        }
        const u8 t0 = sbase[(spos + i0) & smask], t1 = sbase[(spos + i1) & smask], t2 = sbase[(spos + i2) & smask], t3 = sbase[(spos + i3) & smask];
        const u8 t4 = sbase[(spos + i4) & smask], t5 = sbase[(spos + i5) & smask], t6 = sbase[(spos + i6) & smask], t7 = sbase[(spos + i7) & smask];
        const u32 d = opos + c;
        ring[d & kRingMask] = t0;
        if (n > 1) ring[(d + 1) & kRingMask] = t1;
        if (n > 2) ring[(d + 2) & kRingMask] = t2;
        if (n > 3) ring[(d + 3) & kRingMask] = t3;
        if (n > 4) ring[(d + 4) & kRingMask] = t4;
        if (n > 5) ring[(d + 5) & kRingMask] = t5;
        if (n > 6) ring[(d + 6) & kRingMask] = t6;
        if (n > 7) ring[(d + 7) & kRingMask] = t7;
      }
    };
    // Two straight-line rounds take what columnar pages are made of — elements of at most 8 bytes — without a single wave-uniform
    // decision in between: (A) literals and copies whose source lies below the window's first output byte; (B) copies whose source ends
    // below the first element (A) left unwritten.  Whatever remains (longer elements, chains of dependent copies, copies reaching
    // beyond the ring) goes through the general rounds below.
    const bool small = is_start && len <= 8 && !far;
    const bool ready_a = small && (lit || need <= (u32)o);
    {
      u32 cm = 0;
      pass8(ready_a, 0, cm);
    }
    w.lds_sync();
    const bool left_a = is_start && !ready_a;
    const u32 done_b = w.wave_min(left_a ? opos : 0xffffffffu);
    const bool ready_b = left_a && small && need <= done_b;
    {
      u32 cm = 0;
      pass8(ready_b, 0, cm);
    }
    w.lds_sync();
    u64 pending = w.ballot(left_a && !ready_b);
    u32 done = pending ? w.readlane(opos, w.ctz64(pending)) : (u32)o + total;
    while (pending) {
      const bool mine = (pending >> lane) & 1;
      const bool ready = mine && (lit || (!far && need <= done));
      const u64 rmask = w.ballot(ready);
      const int first = w.ctz64(pending);
      if (!((rmask >> first) & 1)) {
        // the oldest pending element reaches beyond the ring: read the flushed output back (coherently), whole wave on one copy
        const u32 fl = w.readlane(len, first), fo = w.readlane(off, first), fp = w.readlane(opos, first);
        w.release_stores();
        for (u32 k = (u32)lane; k < fl; k += 64) ring[(fp + k) & kRingMask] = w.load_coherent_byte(dst + (fp - fo) + (k % fo));
        w.lds_sync();
        pending &= ~((u64)1 << first);
        done = pending ? w.readlane(opos, w.ctz64(pending)) : (u32)o + total;   // later elements may have run already
        continue;
      }
      u32 cmod = 0;                                   // overlapping copies: (bytes already copied) mod offset
      for (u32 c = 0; w.ballot(ready && len > c) != 0; c += 8) pass8(ready && len > c, c, cmod);
      w.lds_sync();
      pending &= ~rmask;
      done = pending ? w.readlane(opos, w.ctz64(pending)) : (u32)o + total;
    }
    SNAPPY_TICK(5);
    o += (int)total;
    p += q;
    flush();
    SNAPPY_TICK(6);
  }
  if (w.ballot(bad) != 0) return ERR_BAD_COPY;
  if (o != dst_len) return ERR_SHORT;
  for (int k = flushed + lane; k < o; k += 64) dst[k] = ring[k & kRingMask];
  return OK;
}

}  // namespace comet_snappy
