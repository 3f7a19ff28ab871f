// comet_device.hpp — hand-written HIP device library for gfx950 (CDNA4, wave64).
//
// This header holds every kernel template of the engine and the arithmetic they share.  It is
// compiled two ways: ahead of time by hipcc (static instantiations in kernels_static.hip) and at
// plan time by hiprtc, where codegen.cpp supplies only the per-row expression functor `P` of a fused
// Scan→Filter→Project→(Aggregate|Output) pipeline.  The grid/wave/LDS structure below is fixed.
//
// Semantics restated from the reference (cited per function):
//   native/spark-expr/src/math_funcs/wide_decimal_binary_expr.rs   (i256 add/sub/mul, HALF_UP)
//   native/spark-expr/src/math_funcs/internal/checkoverflow.rs     (precision bound → null)
//   native/spark-expr/src/math_funcs/internal/decimal_rescale_check.rs
//   native/spark-expr/src/agg_funcs/{sum_decimal,avg_decimal,avg,sum_int}.rs
//   native/spark-expr/src/hash_funcs/murmur3.rs, native/shuffle/src/comet_partitioning.rs
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include "kparams.h"

// string functions with new bytes, digests, instr / ascii / crc32 (plain C++ the host runs too)
#ifndef STRFN
#define STRFN __device__ inline
#endif
#include "strfn.hpp"

// regexp_extract's matcher (comet_regex_vm.hpp, included by the generated sources that call utf8_view_regex)
template <class P>
__device__ bool rx_search(const unsigned int* w, P text, int n, int from, int& m0, int& m1);

namespace comet {

typedef long long i64;
typedef unsigned long long u64;
typedef int i32;
typedef unsigned int u32;
typedef short i16;
typedef unsigned short u16;
typedef signed char i8;
typedef unsigned char u8;
typedef __int128 i128;
typedef unsigned __int128 u128;

#define CDEV __device__ __forceinline__

constexpr int kBlock = 256;   // 4 waves of 64; one block per SIMD-quad, ≥8 blocks/CU resident
constexpr int kWave = 64;

CDEV constexpr i128 mk128(u64 hi, u64 lo) { return (i128)(((u128)hi << 64) | (u128)lo); }
CDEV u64 lo64(i128 v) { return (u64)(u128)v; }
CDEV u64 hi64(i128 v) { return (u64)((u128)v >> 64); }
CDEV u128 uabs128(i128 v) { return v < 0 ? (u128)0 - (u128)v : (u128)v; }

// ---------------------------------------------------------------------------------------------
// Error detail (kparams.h: the block behind out[2]).  The Spark error a raised flag becomes names the offending value
// (common/src/error.rs params_as_json; the JVM side reads params("value") …): the FIRST lane to report leaves the raise site's id and the
// value — its bits, or a string's bytes — for the host to format.  Cold code: reached on the row that fails the task.
// ---------------------------------------------------------------------------------------------
CDEV void err_detail(void* errbuf, u32 site, u64 lo, u64 hi) {
  unsigned long long* d = (unsigned long long*)errbuf + COMET_ERR_DETAIL_WORD;
  if (atomicCAS(d, 0ull, (unsigned long long)site + 1ull) == 0ull) { d[1] = lo; d[2] = hi; }
}
CDEV void err_detail_str(void* errbuf, u32 site, const u8* p, i32 n) {
  unsigned long long* d = (unsigned long long*)errbuf + COMET_ERR_DETAIL_WORD;
  if (atomicCAS(d, 0ull, (unsigned long long)site + 1ull) == 0ull) {
    d[1] = (unsigned long long)(n < 0 ? 0 : n);
    u8* o = (u8*)(d + 4);
    for (i32 k = 0; k < n && k < COMET_ERR_DETAIL_STR_BYTES; k++) o[k] = p[k];
  }
}

// ---------------------------------------------------------------------------------------------
// Column access.  Lane l of a wave reads row base+l: every load instruction is one contiguous
// 64×sizeof(T) segment (1 KiB for Decimal128).
// ---------------------------------------------------------------------------------------------
// Column buffers are always global memory: say so (address_space(1)), otherwise the `const void*` in the
// kernel-argument struct makes the compiler emit FLAT loads, which bump lgkmcnt as well as vmcnt and thereby
// couple every LDS wait to the outstanding HBM loads.
#define COMET_GLOBAL __attribute__((address_space(1)))
// COMET_LD_NT (set by the generator for aggregate sinks; COMET_LD_NT=0/1 in the environment overrides it for experiments): column loads as
// non-temporal loads — the lines are marked for early eviction in the L2 instead of displacing what a kernel re-reads
#if defined(COMET_LD_NT) && COMET_LD_NT
template <class T>
CDEV T ld_stream(const COMET_GLOBAL T* p) { return __builtin_nontemporal_load(p); }
template <>
CDEV i128 ld_stream<i128>(const COMET_GLOBAL i128* p) {
  const u64 lo = __builtin_nontemporal_load((const COMET_GLOBAL u64*)p), hi = __builtin_nontemporal_load((const COMET_GLOBAL u64*)p + 1);
  return (i128)(((u128)hi << 64) | lo);
}
#else
template <class T>
CDEV T ld_stream(const COMET_GLOBAL T* p) { return *p; }
#endif
template <class T>
CDEV T ld(const CometCol& c, i64 i) { return ld_stream(&((const COMET_GLOBAL T*)c.data)[c.offset + i]); }
CDEV bool ld_valid(const CometCol& c, i64 i) {
  i64 j = c.offset + i;
  return (((const COMET_GLOBAL u8*)c.valid)[j >> 3] >> (j & 7)) & 1;
}
// Boolean values are bit-packed in Arrow.
CDEV bool ld_bool(const CometCol& c, i64 i) {
  i64 j = c.offset + i;
  return (((const COMET_GLOBAL u8*)c.data)[j >> 3] >> (j & 7)) & 1;
}
// Decimal128 whose precision ≤ 18: the upper limb is sign extension, read only the lower one.
CDEV i64 ld_dec_lo(const CometCol& c, i64 i) { return ld_stream(&((const COMET_GLOBAL i64*)c.data)[2 * (c.offset + i)]); }

// Utf8 values of ≤ 15 bytes packed into two words (bytes 0-7 in a, bytes 8-14 in the low 56 bits of b,
// length in the top byte of b): injective, so equality and grouping on the packed form are exact.
// Longer strings set `toolong` (the host turns it into an explicit "not supported yet" error).
struct str16 {
  u64 a, b;
};
CDEV str16 ld_str16(const CometCol& c, i64 i, bool& toolong) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  i64 j = c.offset + i;
  i32 lo = off[j], len = off[j + 1] - lo;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  str16 r;
  r.a = 0;
  r.b = 0;
  if (len > 15) { toolong = true; len = 15; }
  for (i32 k = 0; k < len; k++) {
    u64 byte = p[k];
    if (k < 8) r.a |= byte << (8 * k);
    else r.b |= byte << (8 * (k - 8));
  }
  r.b |= (u64)len << 56;
  return r;
}

// substring(str, pos, len) with Spark's rules (UTF8String.substringSQL): 1-based, CHARACTER positions; pos 0 behaves like 1; a negative
// pos counts from the end; the window [start, start + len) is clipped to the string.  The result is built straight from the column
// bytes into the packed form, so the source may have any length — only a RESULT longer than 15 bytes sets `toolong`.
CDEV str16 utf8_substr16(const CometCol& c, i64 i, i32 pos, i32 len, bool& toolong) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j], nbytes = off[j + 1] - lo;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  str16 r;
  r.a = 0;
  r.b = 0;
  i64 start;
  if (pos > 0) start = (i64)pos - 1;
  else if (pos < 0) {
    i32 nchars = 0;
    for (i32 k = 0; k < nbytes; k++) nchars += (p[k] & 0xC0) != 0x80;
    start = (i64)nchars + pos;
  } else start = 0;
  const i64 until = start + (i64)len;
  if (until <= start || start >= nbytes) return r;   // empty string (length byte 0)
  // byte range of characters [start, until)
  i32 k = 0;
  i64 ch = 0;
  while (k < nbytes && ch < start) { k++; while (k < nbytes && (p[k] & 0xC0) == 0x80) k++; ch++; }
  const i32 b0 = k;
  while (k < nbytes && ch < until) { k++; while (k < nbytes && (p[k] & 0xC0) == 0x80) k++; ch++; }
  i32 n = k - b0;
  if (n > 15) { toolong = true; n = 15; }
  for (i32 q = 0; q < n; q++) {
    const u64 byte = p[b0 + q];
    if (q < 8) r.a |= byte << (8 * q);
    else r.b |= byte << (8 * (q - 8));
  }
  r.b |= (u64)n << 56;
  return r;
}

// A string RESULT of any length that is a slice of a source value plus padding: (source row, first byte and byte count inside that
// value, number of pad CHARACTERS).  The projection's kernel writes one of these per output row; the executor then sizes and writes
// the Utf8 column (strview kernels, exchange_kernels.hip).  substring / trim / rpad / lpad / read-side padding all reduce to it.
struct strview { u32 row, start, len, pad; };
CDEV strview utf8_view_substr(const CometCol& c, i64 i, i32 pos, i32 len) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j], nbytes = off[j + 1] - lo;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  strview r = {(u32)i, 0u, 0u, 0u};
  i64 start;
  if (pos > 0) start = (i64)pos - 1;
  else if (pos < 0) {
    i32 nchars = 0;
    for (i32 k = 0; k < nbytes; k++) nchars += (p[k] & 0xC0) != 0x80;
    start = (i64)nchars + pos;
  } else start = 0;
  // Spark's substringSQL: a window that starts before the string is clipped to it (its end stays where it was)
  i64 until = start + (i64)len;
  if (start < 0) start = 0;
  if (until <= start || start >= nbytes) return r;
  i32 k = 0;
  i64 ch = 0;
  while (k < nbytes && ch < start) { k++; while (k < nbytes && (p[k] & 0xC0) == 0x80) k++; ch++; }
  const i32 b0 = k;
  while (k < nbytes && ch < until) { k++; while (k < nbytes && (p[k] & 0xC0) == 0x80) k++; ch++; }
  r.start = (u32)b0;
  r.len = (u32)(k - b0);
  return r;
}
// mode: 1 leading, 2 trailing, 3 both — the space character U+0020 only (Spark's trim without a trim string)
CDEV strview utf8_view_trim(const CometCol& c, i64 i, int mode) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j], nbytes = off[j + 1] - lo;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  i32 a = 0, b = nbytes;
  if (mode & 1) while (a < b && p[a] == 0x20) a++;
  if (mode & 2) while (b > a && p[b - 1] == 0x20) b--;
  strview r = {(u32)i, (u32)a, (u32)(b - a), 0u};
  return r;
}
// regexp_extract: the span of one group in the leftmost match (comet_regex_vm.hpp's matcher over the program the host compiled); no match
// or an unset group = the empty string.  The generated source includes the matcher's header before it calls this (a template: instantiated there).
template <class W>
CDEV strview utf8_view_regex(const CometCol& c, i64 i, const W* prog) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j], nbytes = off[j + 1] - lo;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  i32 m0 = -1, m1 = -1;
  rx_search(prog, p, nbytes, 0, m0, m1);
  strview r = {(u32)i, m0 < 0 ? 0u : (u32)m0, m0 < 0 ? 0u : (u32)(m1 - m0), 0u};
  return r;
}
// instr / ascii / crc32 of a Utf8 value (device/strfn.hpp, the source the host tests run)
CDEV i32 utf8_instr_lit(const CometCol& c, i64 i, const char* lit, i32 m) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j];
  return sf_instr((const COMET_GLOBAL u8*)c.aux + lo, off[j + 1] - lo, (const u8*)lit, m);
}
CDEV i32 utf8_ascii(const CometCol& c, i64 i) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j];
  return sf_ascii((const COMET_GLOBAL u8*)c.aux + lo, off[j + 1] - lo);
}
CDEV i64 utf8_crc32(const CometCol& c, i64 i) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j];
  return (i64)sf_crc32((const COMET_GLOBAL u8*)c.aux + lo, (long long)(off[j + 1] - lo));
}
// pad (or, with `truncate`, cut) the value to `target` characters: rpad / lpad truncate, read-side padding of CHAR(n) columns does not
// (static_invoke/char_varchar_utils/read_side_padding.rs:35-47)
CDEV strview utf8_view_pad(const CometCol& c, i64 i, i32 target, bool truncate) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j], nbytes = off[j + 1] - lo;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  strview r = {(u32)i, 0u, (u32)nbytes, 0u};
  if (target < 0) target = 0;
  i32 k = 0, ch = 0;
  while (k < nbytes && ch < target) { k++; while (k < nbytes && (p[k] & 0xC0) == 0x80) k++; ch++; }
  if (k < nbytes) {            // longer than the target: first `target` characters, or the whole value
    if (truncate) r.len = (u32)k;
  } else {
    r.pad = (u32)(target - ch);
  }
  return r;
}

// ---------------------------------------------------------------------------------------------
// String casts (spark-expr/src/conversion_funcs/string.rs, trim.rs, numeric.rs).  Plain byte and integer arithmetic over one value's
// bytes: this section is also compiled for the host (tests/test_string_casts_cpu.py, g++) and checked there against the reference's own
// vectors and the oracle's restatement.  Parsers return 0 = a value, 1 = invalid input (NULL, or CAST_INVALID_INPUT under ANSI),
// 2 = NULL in every eval mode, 3 = out of range (NULL, or NUMERIC_VALUE_OUT_OF_RANGE under ANSI).  mode: 0 LEGACY, 1 ANSI, 2 TRY.
// ---- string casts: begin
typedef const COMET_GLOBAL u8* strp;
// UTF8String.trimAll: bytes ≤ 0x20 and 0x7F (trim.rs:47-49)
CDEV void str_trim_all(strp p, i32& a, i32& b) {
  while (a < b && (p[a] <= 0x20 || p[a] == 0x7F)) a++;
  while (b > a && (p[b - 1] <= 0x20 || p[b - 1] == 0x7F)) b--;
}
// java.lang.String.trim: bytes ≤ 0x20 (trim.rs:58-61)
CDEV void str_trim_java(strp p, i32& a, i32& b) {
  while (a < b && p[a] <= 0x20) a++;
  while (b > a && p[b - 1] <= 0x20) b--;
}
// p[a, b) equals the lower-case ASCII word w, ignoring ASCII case
CDEV bool str_is_word(strp p, i32 a, i32 b, const char* w, i32 wn) {
  if (b - a != wn) return false;
  for (i32 k = 0; k < wn; k++) {
    u8 ch = p[a + k];
    if (ch >= 'A' && ch <= 'Z') ch = (u8)(ch + 32);
    if (ch != (u8)w[k]) return false;
  }
  return true;
}
// spark_cast_utf8_to_boolean (string.rs:260-312)
CDEV int str_to_bool(strp p, i32 n, bool& out) {
  i32 a = 0, b = n;
  str_trim_all(p, a, b);
  if (str_is_word(p, a, b, "t", 1) || str_is_word(p, a, b, "true", 4) || str_is_word(p, a, b, "y", 1) || str_is_word(p, a, b, "yes", 3) || str_is_word(p, a, b, "1", 1)) { out = true; return 0; }
  if (str_is_word(p, a, b, "f", 1) || str_is_word(p, a, b, "false", 5) || str_is_word(p, a, b, "n", 1) || str_is_word(p, a, b, "no", 2) || str_is_word(p, a, b, "0", 1)) { out = false; return 0; }
  return 1;
}
// do_parse_string_to_int_{legacy,ansi,try} (string.rs:942-1060): accumulated NEGATIVE with the reference's own overflow tests; Int8 / Int16
// parse as i32 and are range-checked afterwards (string.rs:1063-1103)
CDEV int str_to_int(strp p, i32 n, int mode, int bits, i64& out) {
  i32 a = 0, b = n;
  str_trim_all(p, a, b);
  if (a == b) return 1;
  bool neg = false;
  if ((p[a] == '-' || p[a] == '+') && b - a > 1) { neg = p[a] == '-'; a++; }
  const i64 lo = bits == 64 ? (i64)0x8000000000000000ull : -(i64)2147483648ll;
  const i64 stop = lo / 10;                 // truncates toward zero, like Rust
  i64 r = 0;
  i32 i = a;
  for (; i < b; i++) {
    const u8 ch = p[i];
    if (ch == '.') {
      if (mode != 0) return 1;
      i++;
      break;
    }
    if (ch < '0' || ch > '9') return 1;
    if (r < stop) return 1;
    const i64 v = r * 10, d = (i64)(ch - '0');
    if (v < lo + d) return 1;               // checked_sub
    r = v - d;
  }
  for (; i < b; i++)                        // LEGACY: the fraction is validated and ignored
    if (p[i] < '0' || p[i] > '9') return 1;
  if (!neg) {
    if (r == lo) return 1;                  // checked_neg
    r = -r;
  }
  if (bits == 8 && (r < -128 || r > 127)) return 1;
  if (bits == 16 && (r < -32768 || r > 32767)) return 1;
  out = r;
  return 0;
}
CDEV u128 str_pow10(int e) {
  u128 v = 1;
  for (int k = 0; k < e; k++) v *= 10;
  return v;
}
// One byte of the value with fullwidth digits (U+FF10..U+FF19 = EF BC 90..99) read as ASCII digits (normalize_fullwidth_digits, string.rs:472-495)
CDEV u8 str_norm_next(strp p, i32& i, i32 b) {
  if (p[i] == 0xEF && i + 2 < b && p[i + 1] == 0xBC && p[i + 2] >= 0x90 && p[i + 2] <= 0x99) {
    const u8 d = (u8)(p[i + 2] - 0x60);
    i += 3;
    return d;
  }
  return p[i++];
}
// the normalized value equals the lower-case ASCII word w, ignoring ASCII case
CDEV bool str_norm_is_word(strp p, i32 a, i32 b, const char* w, i32 wn) {
  i32 i = a, k = 0;
  while (i < b) {
    u8 ch = str_norm_next(p, i, b);
    if (ch >= 'A' && ch <= 'Z') ch = (u8)(ch + 32);
    if (k >= wn || ch != (u8)w[k]) return false;
    k++;
  }
  return k == wn;
}
// parse_string_to_decimal + parse_decimal_str (string.rs:579-758) into an unscaled value at (precision, scale)
CDEV int str_to_decimal(strp p, i32 n, int precision, int scale, i128& out) {
  const u128 i128_max = ((u128)1 << 127) - 1;
  i32 a = 0, b = n;
  str_trim_java(p, a, b);
  if (a == b) return 1;
  {
    // inf / nan spellings (string.rs:549-576)
    const u8 f = p[a] | 0x20, g = (a + 1 < b) ? (u8)(p[a + 1] | 0x20) : 0;
    if (f == 'i' || f == 'n' || ((p[a] == '+' || p[a] == '-') && g == 'i')) {
      if (str_norm_is_word(p, a, b, "inf", 3) || str_norm_is_word(p, a, b, "+inf", 4) || str_norm_is_word(p, a, b, "-inf", 4) || str_norm_is_word(p, a, b, "infinity", 8) ||
          str_norm_is_word(p, a, b, "+infinity", 9) || str_norm_is_word(p, a, b, "-infinity", 9) || str_norm_is_word(p, a, b, "nan", 3))
        return 1;
    }
  }
  i32 i = a;
  bool neg = false;
  if (p[i] == '-') { neg = true; i++; }
  else if (p[i] == '+') i++;
  u128 ip = 0, fp = 0;
  i32 nint = 0, nfrac = 0;
  bool dot = false, has_exp = false;
  while (i < b) {
    i32 j = i;
    const u8 ch = str_norm_next(p, j, b);
    if (ch >= '0' && ch <= '9') {
      u128& acc = dot ? fp : ip;
      if (acc > (i128_max - (u128)(ch - '0')) / 10) return 1;        // digits_to_i128 leaves i128
      acc = acc * 10 + (u128)(ch - '0');
      if (dot) nfrac++; else nint++;
    } else if (ch == '.' && !dot) {
      dot = true;
    } else if (ch == 'e' || ch == 'E') {
      has_exp = true;
      i = j;
      break;
    } else {
      return 1;
    }
    i = j;
  }
  i64 exponent = 0;
  if (has_exp) {
    // str::parse::<i32>: an optional sign, at least one digit, no overflow
    bool eneg = false;
    if (i < b && (p[i] == '+' || p[i] == '-')) { eneg = p[i] == '-'; i++; }   // (the sign is ASCII: fullwidth forms are digits only)
    if (i >= b) return 1;
    i64 ev = 0;
    while (i < b) {
      const u8 ch = str_norm_next(p, i, b);
      if (ch < '0' || ch > '9') return 1;
      ev = ev * 10 + (i64)(ch - '0');
      if (ev > ((i64)1 << 40)) ev = (i64)1 << 40;
    }
    exponent = eneg ? -ev : ev;
    if (exponent > 2147483647ll || exponent < -2147483648ll) return 1;
  }
  if (nint == 0 && nfrac == 0) return 1;
  if (nfrac > 38) return 1;
  const u128 pf = str_pow10(nfrac);
  if (ip != 0 && ip > i128_max / pf) return 1;
  u128 mant = ip * pf;
  if (mant > i128_max - fp) return 1;
  mant += fp;
  const i64 final_scale = (i64)nfrac - exponent;
  if (mant == 0) {
    if (final_scale < -37) return 3;
    out = 0;
    return 0;
  }
  const i64 adj = (i64)scale - final_scale;
  u128 mag;
  if (adj >= 0) {
    if (adj > 38) return 1;
    const u128 m = str_pow10((int)adj);
    // i128 checked_mul: −2^127 is representable
    const u128 lim = neg ? ((u128)1 << 127) : i128_max;
    if (mant > lim / m) return 3;
    mag = mant * m;
  } else {
    if (-adj > 38) { out = 0; return 0; }
    const u128 d = str_pow10((int)-adj);
    mag = mant / d;
    if (mant % d >= d / 2) mag++;           // HALF_UP (string.rs:520-530)
  }
  if (mag >= str_pow10(precision)) return 3;
  out = neg ? -(i128)mag : (i128)mag;
  return 0;
}
#include "dates.hpp"

// resolve_epoch_day (string.rs:1929-1955)
CDEV int str_resolve_date(i64 y, i64 m, i64 d, i32& out) {
  if (m < 1 || m > 12) return 1;
  const bool leap = y % 4 == 0 && (y % 100 != 0 || y % 400 == 0);
  const i64 mx = m == 2 ? (leap ? 29 : 28) : (m == 4 || m == 6 || m == 9 || m == 11) ? 30 : 31;
  if (d < 1 || d > mx) return 1;
  const i64 days = str_days_from_civil(y, m, d);
  if (days < -2147483648ll || days > 2147483647ll) return 1;
  if (y < -262143 || y > 262142) return 2;     // beyond chrono's years: NULL in every mode
  out = (i32)days;
  return 0;
}
// date_parser (string.rs:1896-2046)
CDEV int str_to_date(strp p, i32 n, i32& out) {
  if (n == 0) return 1;
  i32 j = 0, end = n;
  str_trim_all(p, j, end);
  if (j == end) return 1;
  auto dig = [&](i32 k) { return p[k] >= '0' && p[k] <= '9'; };
  if (end - j == 10 && p[j + 4] == '-' && p[j + 7] == '-' && dig(j) && dig(j + 1) && dig(j + 2) && dig(j + 3) && dig(j + 5) && dig(j + 6) && dig(j + 8) && dig(j + 9)) {
    const i64 y = (p[j] - '0') * 1000 + (p[j + 1] - '0') * 100 + (p[j + 2] - '0') * 10 + (p[j + 3] - '0');
    return str_resolve_date(y, (p[j + 5] - '0') * 10 + (p[j + 6] - '0'), (p[j + 8] - '0') * 10 + (p[j + 9] - '0'), out);
  }
  i32 seg[3] = {1, 1, 1};
  i32 sign = 1, cur = 0, digits = 0;
  u32 val = 0;                                 // Wrapping<i32>
  if (p[j] == '-') { sign = -1; j++; }
  else if (p[j] == '+') j++;
  auto valid_digits = [](i32 s, i32 nd) { return (s == 0 && nd >= 4 && nd <= 7) || (s != 0 && nd > 0 && nd <= 2); };
  while (j < end && cur < 3 && !(p[j] == ' ' || p[j] == 'T')) {
    const u8 ch = p[j];
    if (cur < 2 && ch == '-') {
      if (!valid_digits(cur, digits)) return 1;
      seg[cur] = (i32)val;
      val = 0;
      digits = 0;
      cur++;
    } else if (ch < '0' || ch > '9') {
      return 1;
    } else {
      val = val * 10u + (u32)(ch - '0');
      digits++;
    }
    j++;
  }
  if (!valid_digits(cur, digits)) return 1;
  if (cur < 2 && j < end) return 1;
  seg[cur] = (i32)val;
  return str_resolve_date((i64)(i32)((u32)sign * (u32)seg[0]), seg[1], seg[2], out);
}

// ---- values to strings: each writes at most 48 bytes to `o` and returns the byte count
CDEV i32 fmt_u128_digits(u128 v, u8* o) {         // decimal digits, no sign
  u8 tmp[40];
  i32 k = 0;
  while (v > (u128)0xFFFFFFFFFFFFFFFFull) { tmp[k++] = (u8)('0' + (int)(v % 10)); v /= 10; }
  u64 w = (u64)v;
  do { tmp[k++] = (u8)('0' + (int)(w % 10)); w /= 10; } while (w);
  for (i32 q = 0; q < k; q++) o[q] = tmp[k - 1 - q];
  return k;
}
// arrow-cast integer → Utf8 (numeric.rs:35-47)
CDEV i32 fmt_i64(i64 v, u8* o) {
  i32 k = 0;
  u64 m = (u64)v;
  if (v < 0) { o[k++] = '-'; m = 0ull - m; }
  return k + fmt_u128_digits((u128)m, o + k);
}
// arrow-cast boolean → Utf8
CDEV i32 fmt_bool(bool v, u8* o) {
  const char* s = v ? "true" : "false";
  const i32 n = v ? 4 : 5;
  for (i32 k = 0; k < n; k++) o[k] = (u8)s[k];
  return n;
}
// Decimal128 → Utf8.  java_string: decimal128_to_java_string (numeric.rs:660-704, BigDecimal.toString, LEGACY); else arrow-cast's plain notation
CDEV i32 fmt_decimal(i128 v, int scale, bool java_string, u8* o) {
  u8 dg[40];
  const u128 mag = v < 0 ? (u128)0 - (u128)v : (u128)v;
  const i32 nd = fmt_u128_digits(mag, dg);
  const i64 adj = -(i64)scale + (nd - 1);
  i32 k = 0;
  if (v < 0) o[k++] = '-';
  if (!java_string || (scale >= 0 && adj >= -6)) {
    if (scale <= 0) {
      for (i32 q = 0; q < nd; q++) o[k++] = dg[q];
      if (!java_string) for (i32 q = 0; q < -scale && k < 48; q++) o[k++] = '0';
    } else if (nd > scale) {
      for (i32 q = 0; q < nd - scale; q++) o[k++] = dg[q];
      o[k++] = '.';
      for (i32 q = nd - scale; q < nd; q++) o[k++] = dg[q];
    } else {
      o[k++] = '0';
      o[k++] = '.';
      for (i32 q = 0; q < scale - nd; q++) o[k++] = '0';
      for (i32 q = 0; q < nd; q++) o[k++] = dg[q];
    }
    return k;
  }
  o[k++] = dg[0];
  if (nd > 1) {
    o[k++] = '.';
    for (i32 q = 1; q < nd; q++) o[k++] = dg[q];
  }
  o[k++] = 'E';
  if (adj > 0) o[k++] = '+';
  return k + fmt_i64(adj, o + k);
}
// civil date of an epoch day (the proleptic Gregorian calendar chrono uses)
CDEV void fmt_civil(i64 z, i64& y, i32& m, i32& d) {
  z += 719468;
  const i64 era = (z >= 0 ? z : z - 146096) / 146097;
  const i64 doe = z - era * 146097;
  const i64 yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const i64 doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const i64 mp = (5 * doy + 2) / 153;
  d = (i32)(doy - (153 * mp + 2) / 5 + 1);
  m = (i32)(mp < 10 ? mp + 3 : mp - 9);
  y = yoe + era * 400 + (m <= 2 ? 1 : 0);
}
CDEV i32 fmt_2(i32 v, u8* o) { o[0] = (u8)('0' + v / 10); o[1] = (u8)('0' + v % 10); return 2; }
// chrono's "%Y-%m-%d": four digits for years 0..9999, otherwise a sign and at least four digits
CDEV i32 fmt_date(i64 days, u8* o) {
  i64 y; i32 m, d;
  fmt_civil(days, y, m, d);
  i32 k = 0;
  if (y < 0 || y > 9999) o[k++] = y < 0 ? '-' : '+';
  const u64 ay = (u64)(y < 0 ? -y : y);
  u8 dg[24];
  const i32 nd = fmt_u128_digits((u128)ay, dg);
  for (i32 q = nd; q < 4; q++) o[k++] = '0';
  for (i32 q = 0; q < nd; q++) o[k++] = dg[q];
  o[k++] = '-';
  k += fmt_2(m, o + k);
  o[k++] = '-';
  k += fmt_2(d, o + k);
  return k;
}
// Timestamp(µs) → Utf8 in a fixed-offset zone: "%Y-%m-%d %H:%M:%S%.f" (cast.rs:71) with the fraction's trailing zeroes removed (utils.rs:88-113)
CDEV i32 fmt_timestamp(i64 micros, i64 offset_seconds, u8* o) {
  const i128 local = (i128)micros + (i128)offset_seconds * 1000000;
  const i128 day_us = (i128)86400 * 1000000;
  i128 days = local / day_us, rem = local % day_us;
  if (rem < 0) { rem += day_us; days -= 1; }
  const i64 secs = (i64)(rem / 1000000);
  i32 us = (i32)(rem % 1000000);
  i32 k = fmt_date((i64)days, o);
  o[k++] = ' ';
  k += fmt_2((i32)(secs / 3600), o + k);
  o[k++] = ':';
  k += fmt_2((i32)(secs / 60 % 60), o + k);
  o[k++] = ':';
  k += fmt_2((i32)(secs % 60), o + k);
  if (us) {
    o[k++] = '.';
    i32 nd = 6;
    while (us % 10 == 0) { us /= 10; nd--; }
    for (i32 q = nd - 1; q >= 0; q--) { o[k + q] = (u8)('0' + us % 10); us /= 10; }
    k += nd;
  }
  return k;
}
// ---- string casts: end
// the bytes of value i of a Utf8 column
CDEV strp utf8_bytes(const CometCol& c, i64 i, i32& n) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  n = off[j + 1] - off[j];
  return (strp)c.aux + off[j];
}

// ---------------------------------------------------------------------------------------------
// Time zones (csrc/tz.cpp flattens a zone into zt = { n, first_off, limit, at[0..n) UTC seconds, off[0..n) seconds east }): the offset in force at
// an instant, and the instant of a local wall-clock time as chrono-tz's from_local_datetime and the reference's resolve_local_datetime
// (spark-expr/src/utils.rs:184-205) decide it.  Plain integer code, also compiled for the host (tests/test_time_zones_cpu.py).
// ---- time zones: begin
typedef const i64* tzp;      // (a kernel's zone tables are its own constant arrays: generic pointers)
CDEV i64 tz_floor_div(i64 a, i64 b) { const i64 q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
// transitions at or before the instant
CDEV i64 tz_span_of(tzp zt, i64 utc_s) {
  i64 lo = 0, hi = zt[0];
  while (lo < hi) {
    const i64 mid = (lo + hi) >> 1;
    if (zt[3 + mid] <= utc_s) lo = mid + 1; else hi = mid;
  }
  return lo;
}
CDEV i64 tz_span_offset(tzp zt, i64 span) { return span == 0 ? zt[1] : zt[3 + zt[0] + span - 1]; }
// Behind the table's end (zt[2]) the zone's daylight-saving rule goes on for ever, and the Gregorian calendar — weekdays included — repeats every
// 400 years = 146097 days: an instant there is read 400-year periods earlier, inside the table's last, rule-generated 400 years (csrc/tz.cpp
// expands the rule that far).  java.time's ZoneRules — Spark's answers — apply the last rule the same way.
CDEV i64 tz_fold(i64 limit, i64 s) {
  const i64 period = 146097ll * 86400ll;
  return s < limit ? s : s - ((s - limit) / period + 1) * period;
}
CDEV i64 tz_offset_at(tzp zt, i64 utc_s) { return tz_span_offset(zt, tz_span_of(zt, tz_fold(zt[2], utc_s))); }
// (a wall-clock second folds a day earlier than an instant: its spans' instants lie within a day of it)
CDEV i64 tz_fold_local(tzp zt, i64 L) { return tz_fold(zt[2] == (i64)0x7fffffffffffffffll ? zt[2] : zt[2] - 86400, L); }
// UTC µs → the zone's wall clock as µs (what Timestamp → Date / String / hour() look at)
CDEV i64 tz_utc_to_local_us(tzp zt, i64 us, bool& beyond) {
  const i64 s = tz_floor_div(us, 1000000);
  beyond = false;
  return us + tz_offset_at(zt, s) * 1000000;
}
// spans whose wall clock shows local second L: 0 (a gap), 1, or 2 (an overlap: `off` is the EARLIER span's, chrono's Ambiguous(earliest, _))
CDEV int tz_local_spans(tzp zt, i64 L, i64& off) {
  const i64 n = zt[0];
  const i64 g = tz_span_of(zt, L);
  int cnt = 0;
  for (i64 j = g - 2; j <= g + 2; j++) {
    if (j < 0 || j > n) continue;
    const i64 o = tz_span_offset(zt, j);
    const i64 u = L - o;
    if ((j == 0 || zt[3 + j - 1] <= u) && (j == n || u < zt[3 + j])) {
      if (cnt == 0) off = o;
      cnt++;
    }
  }
  return cnt;
}
// local wall-clock µs → UTC µs (resolve_local_datetime: an overlap takes the earlier instant; a gap takes the offset in force three hours before)
CDEV i64 tz_local_to_utc_us(tzp zt, i64 local_us, bool& beyond) {
  const i64 L = tz_fold_local(zt, tz_floor_div(local_us, 1000000));
  i64 off = 0;
  if (tz_local_spans(zt, L, off) == 0 && tz_local_spans(zt, L - 10800, off) == 0) off = 0;
  beyond = false;
  return local_us - off * 1000000;
}
// ---- time zones: end

// ---------------------------------------------------------------------------------------------
// Scalar functions (ScalarFunc, expr.proto:466-471): the exact, integer/IEEE-defined subset.
// ---------------------------------------------------------------------------------------------
// Rust `x as i64`: saturating, NaN → 0 (spark_ceil / spark_floor: math_funcs/ceil.rs:31-40, floor.rs)
// Java's Math.rint (datafusion-spark SparkRint): the closest integral double, ties to even — the current rounding mode's rint
CDEV double f64_rint(double x) { return rint(x); }
CDEV i64 f64_to_i64_sat(double x) {
  if (x != x) return 0;
  if (x >= 9223372036854775808.0) return (i64)0x7fffffffffffffffll;
  if (x <= -9223372036854775808.0) return (i64)0x8000000000000000ull;
  return (i64)x;
}
// Rust `x as i32` for a float: saturating, NaN → 0 (conversion_funcs/numeric.rs cast_float_to_int*)
CDEV i32 f64_to_i32_sat(double x) {
  if (x != x) return 0;
  if (x >= 2147483648.0) return (i32)0x7fffffff;
  if (x <= -2147483648.0) return (i32)0x80000000;
  return (i32)x;
}
// div_ceil / div_floor of the unscaled value by 10^scale (decimal_ceil_f / decimal_floor_f)
CDEV i128 dec_div_ceil(i128 x, i128 d) { i128 q = x / d, r = x % d; return (r > 0) ? q + 1 : q; }
CDEV i128 dec_div_floor(i128 x, i128 d) { i128 q = x / d, r = x % d; return (r < 0) ? q - 1 : q; }
// Direct Utf8 comparisons for predicates: byte-wise unsigned lexicographic order (Spark's UTF8String binary compare, arrow-ord's
// string kernels), any length, no packing.  Returns <0, 0, >0.
CDEV int utf8_cmp_lit(const CometCol& c, i64 i, const char* lit, i32 n) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j], len = off[j + 1] - lo;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  const i32 m = len < n ? len : n;
  for (i32 k = 0; k < m; k++) {
    const u8 x = p[k], y = (u8)lit[k];
    if (x != y) return x < y ? -1 : 1;
  }
  return len < n ? -1 : (len > n ? 1 : 0);
}
CDEV bool utf8_eq_lit(const CometCol& c, i64 i, const char* lit, i32 n) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j];
  if (off[j + 1] - lo != n) return false;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  for (i32 k = 0; k < n; k++)
    if (p[k] != (u8)lit[k]) return false;
  return true;
}
// ---- byte-wise string predicates against a literal (UTF8_BINARY collation: the reference compares raw bytes too,
// spark/src/main/scala/org/apache/comet/serde/strings.scala:343-360) ----
CDEV bool utf8_starts_with_lit(const CometCol& c, i64 i, const char* lit, i32 n) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j];
  if (off[j + 1] - lo < n) return false;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  for (i32 k = 0; k < n; k++)
    if (p[k] != (u8)lit[k]) return false;
  return true;
}
CDEV bool utf8_ends_with_lit(const CometCol& c, i64 i, const char* lit, i32 n) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 hi = off[j + 1];
  if (hi - off[j] < n) return false;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + (hi - n);
  for (i32 k = 0; k < n; k++)
    if (p[k] != (u8)lit[k]) return false;
  return true;
}
CDEV bool utf8_contains_lit(const CometCol& c, i64 i, const char* lit, i32 n) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j], len = off[j + 1] - lo;
  if (n == 0) return true;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  const u8 first = (u8)lit[0];
  for (i32 s = 0; s + n <= len; s++) {
    if (p[s] != first) continue;
    i32 k = 1;
    while (k < n && p[s + k] == (u8)lit[k]) k++;
    if (k == n) return true;
  }
  return false;
}
// SQL LIKE with `%` (any run of characters), `_` (exactly one CHARACTER — a whole UTF-8 code point) and `\` escaping the next
// pattern byte (Spark's default escape; strings.scala:314-341 rejects any other).  Iterative matcher that backtracks to the last `%`.
CDEV i32 utf8_next_char(const COMET_GLOBAL u8* p, i32 at, i32 len) {
  at++;
  while (at < len && (p[at] & 0xC0) == 0x80) at++;
  return at;
}
// RLIKE: walk the search automaton the host compiled from the pattern (csrc/regex.cpp): trans[state][byte] → state; flags bit 0 = a match
// has been found (answer true at once), bit 1 = a match if the text ends in this state.  One table lookup per byte of the value.
// (the table steps on byte CLASSES — bytes no state tells apart — and holds 16-bit states: regex.hpp RegexDfa)
CDEV bool utf8_rlike(const CometCol& c, i64 i, const char* trans, const char* flags, const char* classes, u32 nclasses) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j], nbytes = off[j + 1] - lo;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  if (flags[0] & 1) return true;
  u32 st = 0;
  for (i32 k = 0; k < nbytes; k++) {
    const u32 at = (st * nclasses + (u8)classes[p[k]]) * 2u;
    st = (u32)(u8)trans[at] | ((u32)(u8)trans[at + 1] << 8);
    if (flags[st] & 1) return true;
  }
  return (flags[st] & 2) != 0;
}
CDEV bool utf8_like_lit(const CometCol& c, i64 i, const char* pat, i32 m) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j], len = off[j + 1] - lo;
  const COMET_GLOBAL u8* s = (const COMET_GLOBAL u8*)c.aux + lo;
  i32 si = 0, pi = 0, star_p = -1, star_s = 0;
  while (si < len) {
    if (pi < m) {
      const u8 pc = (u8)pat[pi];
      if (pc == '%') { star_p = ++pi; star_s = si; continue; }
      if (pc == '_') { si = utf8_next_char(s, si, len); pi++; continue; }
      const bool esc = pc == '\\' && pi + 1 < m;
      const u8 want = esc ? (u8)pat[pi + 1] : pc;
      if (s[si] == want) { si++; pi += esc ? 2 : 1; continue; }
    }
    if (star_p < 0) return false;
    pi = star_p;
    star_s = utf8_next_char(s, star_s, len);
    si = star_s;
  }
  while (pi < m && pat[pi] == '%') pi++;
  return pi == m;
}
// number of characters (code points): bytes that are not UTF-8 continuation bytes (DataFusion character_length → Int32)
CDEV i32 utf8_char_length(const CometCol& c, i64 i) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  const i32 lo = off[j], len = off[j + 1] - lo;
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + lo;
  i32 n = 0;
  for (i32 k = 0; k < len; k++) n += (p[k] & 0xC0) != 0x80;
  return n;
}
CDEV i32 utf8_octet_length(const CometCol& c, i64 i) {
  const COMET_GLOBAL i32* off = (const COMET_GLOBAL i32*)c.data;
  const i64 j = c.offset + i;
  return off[j + 1] - off[j];
}

CDEV int utf8_cmp(const CometCol& a, i64 i, const CometCol& b, i64 j) {
  const COMET_GLOBAL i32* oa = (const COMET_GLOBAL i32*)a.data;
  const COMET_GLOBAL i32* ob = (const COMET_GLOBAL i32*)b.data;
  const i64 ia = a.offset + i, jb = b.offset + j;
  const i32 la = oa[ia + 1] - oa[ia], lb = ob[jb + 1] - ob[jb];
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)a.aux + oa[ia];
  const COMET_GLOBAL u8* q = (const COMET_GLOBAL u8*)b.aux + ob[jb];
  const i32 m = la < lb ? la : lb;
  for (i32 k = 0; k < m; k++) {
    const u8 x = p[k], y = q[k];
    if (x != y) return x < y ? -1 : 1;
  }
  return la < lb ? -1 : (la > lb ? 1 : 0);
}

// Utf8 column whose values all have the same length LEN (≤ 15; verified by the executor before this variant is
// chosen): the bytes of row i sit at aux + (offset + i)·LEN, so neither the int32 offsets nor a dependent load is
// needed (TPC-H flag/status columns: LEN = 1).
template <int LEN>
CDEV str16 ld_str_fixed(const CometCol& c, i64 i) {
  const COMET_GLOBAL u8* p = (const COMET_GLOBAL u8*)c.aux + (c.offset + i) * LEN;
  str16 r;
  r.a = 0;
  r.b = 0;
#pragma unroll
  for (int k = 0; k < LEN; k++) {
    u64 byte = p[k];
    if (k < 8) r.a |= byte << (8 * k);
    else r.b |= byte << (8 * (k - 8));
  }
  r.b |= (u64)LEN << 56;
  return r;
}

// ---------------------------------------------------------------------------------------------
// 256-bit two's-complement integer (four little-endian u64 limbs) for the wide-decimal path.
// ---------------------------------------------------------------------------------------------
struct i256 {
  u64 w[4];
};
CDEV i256 i256_from_i128(i128 v) {
  i256 r;
  r.w[0] = lo64(v);
  r.w[1] = hi64(v);
  u64 s = v < 0 ? ~0ull : 0ull;
  r.w[2] = s;
  r.w[3] = s;
  return r;
}
CDEV bool i256_neg(const i256& a) { return (a.w[3] >> 63) != 0; }
CDEV i256 i256_add(const i256& a, const i256& b) {
  i256 r;
  u128 c = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    c += (u128)a.w[k] + b.w[k];
    r.w[k] = (u64)c;
    c >>= 64;
  }
  return r;
}
CDEV i256 i256_negate(const i256& a) {
  i256 r;
  u128 c = 1;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    c += (u128)(~a.w[k]);
    r.w[k] = (u64)c;
    c >>= 64;
  }
  return r;
}
CDEV i256 i256_sub(const i256& a, const i256& b) { return i256_add(a, i256_negate(b)); }
CDEV i256 i256_abs(const i256& a) { return i256_neg(a) ? i256_negate(a) : a; }
// unsigned compare of magnitudes
CDEV int u256_cmp(const i256& a, const i256& b) {
#pragma unroll
  for (int k = 3; k >= 0; k--) {
    if (a.w[k] != b.w[k]) return a.w[k] < b.w[k] ? -1 : 1;
  }
  return 0;
}
// magnitude(u128) × magnitude(u128) → u256
CDEV i256 u128_mul_u128(u128 a, u128 b) {
  u64 a0 = (u64)a, a1 = (u64)(a >> 64), b0 = (u64)b, b1 = (u64)(b >> 64);
  u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
  i256 r;
  r.w[0] = (u64)p00;
  u128 mid = (p00 >> 64) + (u64)p01 + (u64)p10;
  r.w[1] = (u64)mid;
  u128 hi = (mid >> 64) + (p01 >> 64) + (p10 >> 64) + (u64)p11;
  r.w[2] = (u64)hi;
  r.w[3] = (u64)((hi >> 64) + (p11 >> 64));
  return r;
}
// wrapping signed 128×128 → 256 (i256::from_i128(l).wrapping_mul(i256::from_i128(r)),
// wide_decimal_binary_expr.rs:276; the true product of two i128 always fits in 256 bits)
CDEV i256 i128_mul_i128(i128 a, i128 b) {
  i256 m = u128_mul_u128(uabs128(a), uabs128(b));
  return ((a < 0) != (b < 0)) ? i256_negate(m) : m;
}
// |a| < 2^127, |b| < 2^63: exact product and the 10^p-1 bound check without forming all 256 bits
// (same result as i128_mul_i128 + i256_fits_bound; TPC-H Q1's charge = disc_price(26,4) × (1+tax)(13,2))
CDEV bool i128_mul_i64_fits(i128 a, i64 b, u128 bound, i128& out) {
  const u128 ua = uabs128(a);
  const u64 ub = (u64)(b < 0 ? -(u64)b : (u64)b);
  const u128 lo = (u128)(u64)ua * ub;           // a0·b
  const u128 hi = (u128)(u64)(ua >> 64) * ub;   // a1·b (to be shifted by 64)
  const u128 mid = hi + (lo >> 64);
  const bool fits = (mid >> 64) == 0;
  const u128 mag = (mid << 64) | (u64)lo;
  out = ((a < 0) != (b < 0)) ? (i128)((u128)0 - mag) : (i128)mag;
  return fits && mag <= bound;
}
// u256 × u128 keeping the low 256 bits (wrapping_mul by a power of ten)
CDEV i256 u256_mul_u128_wrapping(const i256& a, u128 b) {
  u64 bl[2] = {(u64)b, (u64)(b >> 64)};
  u64 r[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 carry = 0;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (i + j < 4) {
        u128 t = (u128)a.w[i] * bl[j] + r[i + j] + carry;
        r[i + j] = (u64)t;
        carry = t >> 64;
      }
    }
    if (i + 2 < 4) {
      u128 t = (u128)r[i + 2] + carry;
      r[i + 2] = (u64)t;
      if (i + 3 < 4) r[i + 3] += (u64)(t >> 64);
    }
  }
  i256 o;
  o.w[0] = r[0]; o.w[1] = r[1]; o.w[2] = r[2]; o.w[3] = r[3];
  return o;
}
// magnitude division u256 / u128 → quotient (u256) and remainder (u128); shift-subtract, only used
// when a wide-decimal result has to be scaled DOWN (rare: Spark picks s_out = s1+s2 unless p > 38).
CDEV void u256_divmod_u128(const i256& n, u128 d, i256& q, u128& rem) {
  q.w[0] = q.w[1] = q.w[2] = q.w[3] = 0;
  u128 r = 0;
  bool rtop = false;  // 129th bit of the running remainder
  for (int bit = 255; bit >= 0; bit--) {
    rtop = (r >> 127) != 0;
    r = (r << 1) | ((n.w[bit >> 6] >> (bit & 63)) & 1);
    if (rtop || r >= d) {
      r -= d;
      q.w[bit >> 6] |= 1ull << (bit & 63);
    }
  }
  rem = r;
}
// div_round_half_up(value, divisor) with divisor = 10^k > 0 (wide_decimal_binary_expr.rs:121-144)
CDEV i256 i256_div_pow10_half_up(const i256& v, u128 divisor) {
  bool neg = i256_neg(v);
  i256 mag = i256_abs(v), q;
  u128 rem;
  u256_divmod_u128(mag, divisor, q, rem);
  // |rem|*2 >= divisor → round away from zero
  bool round = (rem >= divisor - rem);
  if (round) {
    i256 one;
    one.w[0] = 1; one.w[1] = one.w[2] = one.w[3] = 0;
    q = i256_add(q, one);
  }
  return neg ? i256_negate(q) : q;
}
// Spark decimal division (spark-expr/src/math_funcs/div.rs:71-165, non-integral): with L = l·10^l_exp, R = r·10^r_exp
//   div = trunc(L / R);  res = (div + sign(div)·5) / 10 (truncating)  →  round-half-up at the result scale;
// a quotient that does not fit i128 becomes i128::MAX (quotient_to_i128, div.rs:57-68); R = 0 yields 0 here (the caller
// raises DIVIDE_BY_ZERO in ANSI mode; in the other modes Spark has already replaced zero divisors by NULL).
// lmul = 10^l_exp, rmul = 10^r_exp; requires |r|·rmul < 2^127 (checked at plan time through the precisions).
CDEV i128 dec_div(i128 l, i128 r, u128 lmul, u128 rmul, bool& div_by_zero, bool integral = false) {
  const u128 R = uabs128(r) * rmul;
  div_by_zero = R == 0;
  if (div_by_zero) return 0;
  const bool neg = (l < 0) != (r < 0);
  i256 L = u128_mul_u128(uabs128(l), lmul), q;
  u128 rem;
  u256_divmod_u128(L, R, q, rem);
  i256 five;
  five.w[0] = 5; five.w[1] = five.w[2] = five.w[3] = 0;
  i256 q10;
  if (integral) five.w[0] = 0;   // decimal_integral_div truncates: (l / r) / 10 without the ±5
  u256_divmod_u128(i256_add(q, five), 10, q10, rem);
  const i128 kMax = (i128)(((u128)1 << 127) - 1);
  if (q10.w[3] != 0 || q10.w[2] != 0 || (q10.w[1] >> 63) != 0) return kMax;   // to_i128() failed → i128::MAX
  const i128 mag = (i128)(((u128)q10.w[1] << 64) | q10.w[0]);
  return neg ? -mag : mag;
}

// Decimal remainder (create_modulo_expr → arrow-arith's decimal `rem`, math_funcs/modulo_expr.rs:137-206): l·lmul % r·rmul at the larger of the
// two scales — one of lmul / rmul is 1 — with the sign of the dividend.  The rescaled operand may need 256 bits (where the reference casts both
// sides to Decimal256); the remainder itself is smaller than either operand's magnitude, so it is an i128 again.
CDEV i128 dec_rem(i128 l, i128 r, u128 lmul, u128 rmul, bool& div_by_zero) {
  const u128 rm = uabs128(r), lm = uabs128(l);
  div_by_zero = rm == 0;
  if (div_by_zero) return 0;
  u128 rem;
  if (rmul != 1) {
    const i256 R = u128_mul_u128(rm, rmul);
    if (R.w[2] | R.w[3]) rem = lm;                            // the divisor exceeds every 128-bit dividend
    else rem = lm % (((u128)R.w[1] << 64) | R.w[0]);
  } else if (lmul != 1) {
    i256 q;
    u256_divmod_u128(u128_mul_u128(lm, lmul), rm, q, rem);
  } else {
    rem = lm % rm;
  }
  return l < 0 ? -(i128)rem : (i128)rem;
}

// result > bound || result < -bound with bound = 10^p - 1 < 2^127 (check_overflow_and_convert,
// wide_decimal_binary_expr.rs:335-350).  On success the value fits in i128.
CDEV bool i256_fits_bound(const i256& v, u128 bound, i128& out) {
  i256 mag = i256_abs(v);
  bool ok = (mag.w[3] == 0 && mag.w[2] == 0) && ((((u128)mag.w[1] << 64) | mag.w[0]) <= bound);
  out = (i128)(((u128)v.w[1] << 64) | v.w[0]);
  return ok;
}

// ---------------------------------------------------------------------------------------------
// Decimal value checks and narrow-path helpers.
// ---------------------------------------------------------------------------------------------
// Decimal128Type::is_valid_decimal_precision(v, p): |v| <= 10^p - 1 (checkoverflow.rs:128-160)
CDEV bool dec_fits(i128 v, u128 bound) { return uabs128(v) <= bound; }
CDEV bool dec_fits64(i64 v, u64 bound) { return (u64)(v < 0 ? -(u64)v : (u64)v) <= bound; }

// rescale_and_check (decimal_rescale_check.rs:108-150): delta>0 checked_mul, delta<0 HALF_UP by
// adding sign*half before the truncating division.
CDEV bool dec_rescale_up(i128 v, i128 factor, u128 bound, i128& out) {
  i128 r;
  if (__builtin_mul_overflow(v, factor, &r)) return false;
  out = r;
  return dec_fits(r, bound);
}
CDEV bool dec_rescale_down(i128 v, i128 divisor, u128 bound, i128& out) {
  i128 half = divisor / 2;
  i128 sign = (v > 0) - (v < 0);
  i128 r = (v + sign * half) / divisor;
  out = r;
  return dec_fits(r, bound);
}

// AvgDecimal final division (avg_decimal.rs:670-689): sum*scaler / count, ROUND_HALF_UP, bound check.
CDEV bool dec_avg(i128 sum, i64 count, i128 scaler, u128 bound, i128& out) {
  i128 value;
  if (__builtin_mul_overflow(sum, scaler, &value)) return false;
  i128 c = (i128)count;
  i128 div = value / c, rem = value % c;
  i128 half = (c + 1) / 2;  // div_ceil(count, 2), count > 0
  i128 nv = div;
  if (value >= 0) {
    if (rem >= half) nv = div + 1;
  } else {
    if (rem <= -half) nv = div - 1;
  }
  out = nv;
  return dec_fits(nv, bound);
}

// ---------------------------------------------------------------------------------------------
// Spark murmur3_x86_32 (hash_funcs/murmur3.rs:73-142) and pmod (comet_partitioning.rs:51-57).
// ---------------------------------------------------------------------------------------------
CDEV u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
CDEV u32 mm3_mix_k1(u32 k1) { k1 *= 0xcc9e2d51u; k1 = rotl32(k1, 15); k1 *= 0x1b873593u; return k1; }
CDEV u32 mm3_mix_h1(u32 h1, u32 k1) { h1 ^= k1; h1 = rotl32(h1, 13); return h1 * 5u + 0xe6546b64u; }
CDEV u32 mm3_fmix(u32 h1, u32 len) {
  h1 ^= len; h1 ^= h1 >> 16; h1 *= 0x85ebca6bu; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35u; h1 ^= h1 >> 16;
  return h1;
}
CDEV u32 mm3_hash_i32(i32 v, u32 seed) { return mm3_fmix(mm3_mix_h1(seed, mm3_mix_k1((u32)v)), 4u); }
CDEV u32 mm3_hash_i64(i64 v, u32 seed) {
  u32 h = mm3_mix_h1(seed, mm3_mix_k1((u32)(u64)v));
  h = mm3_mix_h1(h, mm3_mix_k1((u32)((u64)v >> 32)));
  return mm3_fmix(h, 8u);
}
CDEV u32 mm3_hash_i128(i128 v, u32 seed) {  // decimal(p>18): 16 LE bytes (hash_funcs/utils.rs hash_array_decimal)
  u32 h = seed;
  u128 u = (u128)v;
#pragma unroll
  for (int k = 0; k < 4; k++) h = mm3_mix_h1(h, mm3_mix_k1((u32)(u >> (32 * k))));
  return mm3_fmix(h, 16u);
}
CDEV u32 mm3_hash_f64(double d, u32 seed) {  // -0.0 hashes as 0 (hash_array_primitive_float)
  i64 bits = (d == 0.0) ? 0 : __double_as_longlong(d);
  return mm3_hash_i64(bits, seed);
}
CDEV u32 mm3_hash_f32(float f, u32 seed) {
  i32 bits = (f == 0.0f) ? 0 : __float_as_int(f);
  return mm3_hash_i32(bits, seed);
}
CDEV u32 mm3_hash_bytes(const u8* p, i32 len, u32 seed) {
  u32 h = seed;
  i32 aligned = len & ~3;
  for (i32 i = 0; i < aligned; i += 4) {
    u32 w = (u32)p[i] | ((u32)p[i + 1] << 8) | ((u32)p[i + 2] << 16) | ((u32)p[i + 3] << 24);
    h = mm3_mix_h1(h, mm3_mix_k1(w));
  }
  for (i32 i = aligned; i < len; i++) h = mm3_mix_h1(h, mm3_mix_k1((u32)(i32)(i8)p[i]));  // sign-extended tail bytes
  return mm3_fmix(h, (u32)len);
}
// XXH64 of 4, 8 or 16 little-endian bytes (twox-hash XxHash64::oneshot as called by hash_funcs/xxhash64.rs:80-82; inputs shorter than
// 32 bytes take the short path: seed + PRIME5 + len, then 8-byte, 4-byte rounds and the avalanche).
CDEV u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
#define COMET_XXH_P1 0x9E3779B185EBCA87ull
#define COMET_XXH_P2 0xC2B2AE3D27D4EB4Full
#define COMET_XXH_P3 0x165667B19E3779F9ull
#define COMET_XXH_P4 0x85EBCA77C2B2AE63ull
#define COMET_XXH_P5 0x27D4EB2F165667C5ull
CDEV u64 xxh64_word(u64 h, u64 w) {
  const u64 k = rotl64(w * COMET_XXH_P2, 31) * COMET_XXH_P1;
  return rotl64(h ^ k, 27) * COMET_XXH_P1 + COMET_XXH_P4;
}
CDEV u64 xxh64_avalanche(u64 h) {
  h ^= h >> 33; h *= COMET_XXH_P2; h ^= h >> 29; h *= COMET_XXH_P3; h ^= h >> 32;
  return h;
}
CDEV u64 xxh64_hash_i32(i32 v, u64 seed) {
  u64 h = seed + COMET_XXH_P5 + 4ull;
  h ^= (u64)(u32)v * COMET_XXH_P1;
  h = rotl64(h, 23) * COMET_XXH_P2 + COMET_XXH_P3;
  return xxh64_avalanche(h);
}
CDEV u64 xxh64_hash_i64(i64 v, u64 seed) { return xxh64_avalanche(xxh64_word(seed + COMET_XXH_P5 + 8ull, (u64)v)); }
CDEV u64 xxh64_hash_i128(i128 v, u64 seed) {
  const u128 u = (u128)v;
  return xxh64_avalanche(xxh64_word(xxh64_word(seed + COMET_XXH_P5 + 16ull, (u64)u), (u64)(u >> 64)));
}
CDEV u64 xxh64_hash_f64(double d, u64 seed) { return xxh64_hash_i64((d == 0.0) ? 0 : __double_as_longlong(d), seed); }   // -0.0 hashes as 0
CDEV u64 xxh64_hash_f32(float f, u64 seed) { return xxh64_hash_i32((f == 0.0f) ? 0 : __float_as_int(f), seed); }
CDEV i32 pmod(u32 hash, i32 n) {
  i32 r = (i32)hash % n;
  return r < 0 ? (r + n) % n : r;
}

// ---------------------------------------------------------------------------------------------
// Wave / block primitives (wave = 64 lanes).
// ---------------------------------------------------------------------------------------------
CDEV u64 shfl_xor_u64(u64 v, int m) {
  u32 lo = (u32)v, hi = (u32)(v >> 32);
  lo = __shfl_xor(lo, m, kWave);
  hi = __shfl_xor(hi, m, kWave);
  return ((u64)hi << 32) | lo;
}
CDEV u64 shfl_u64(u64 v, int src) {
  u32 lo = (u32)v, hi = (u32)(v >> 32);
  lo = __shfl(lo, src, kWave);
  hi = __shfl(hi, src, kWave);
  return ((u64)hi << 32) | lo;
}
CDEV int lane_id() { return threadIdx.x & (kWave - 1); }
CDEV int wave_id() { return threadIdx.x >> 6; }

// Accumulator word ops.  An accumulator is a flat array of u64 words; codegen assigns each
// aggregate primitive a word range and emits the matching combine below.
CDEV void acc_add64(u64* a, const u64* b) { a[0] += b[0]; }
CDEV void acc_add128(u64* a, const u64* b) {
  u128 s = (((u128)a[1] << 64) | a[0]) + (((u128)b[1] << 64) | b[0]);
  a[0] = (u64)s;
  a[1] = (u64)(s >> 64);
}
CDEV void acc_add192(u64* a, const u64* b) {  // 3 limbs, two's complement
  u128 c = (u128)a[0] + b[0];
  a[0] = (u64)c;
  c = (c >> 64) + a[1] + b[1];
  a[1] = (u64)c;
  a[2] = a[2] + b[2] + (u64)(c >> 64);
}
CDEV void acc_umax128(u64* a, const u64* b) {
  bool take = (b[1] > a[1]) || (b[1] == a[1] && b[0] > a[0]);
  if (take) { a[0] = b[0]; a[1] = b[1]; }
}
CDEV void acc_or64(u64* a, const u64* b) { a[0] |= b[0]; }
CDEV void acc_fadd64(u64* a, const u64* b) {
  a[0] = (u64)__double_as_longlong(__longlong_as_double((i64)a[0]) + __longlong_as_double((i64)b[0]));
}
CDEV void acc_imin64(u64* a, const u64* b) { if ((i64)b[0] < (i64)a[0]) a[0] = b[0]; }
CDEV void acc_imax64(u64* a, const u64* b) { if ((i64)b[0] > (i64)a[0]) a[0] = b[0]; }
CDEV void acc_imin128(u64* a, const u64* b) {
  i128 x = mk128(a[1], a[0]), y = mk128(b[1], b[0]);
  if (y < x) { a[0] = b[0]; a[1] = b[1]; }
}
CDEV void acc_imax128(u64* a, const u64* b) {
  i128 x = mk128(a[1], a[0]), y = mk128(b[1], b[0]);
  if (y > x) { a[0] = b[0]; a[1] = b[1]; }
}
// per-row feeders
CDEV void acc_feed_i128(u64* a, i128 v) { u64 t[2] = {lo64(v), hi64(v)}; acc_add128(a, t); }
CDEV void acc_feed_i192(u64* a, i128 v) {
  u64 t[3] = {lo64(v), hi64(v), v < 0 ? ~0ull : 0ull};
  acc_add192(a, t);
}
CDEV void acc_feed_amax(u64* a, i128 v) {
  u128 m = uabs128(v);
  u64 t[2] = {(u64)m, (u64)(m >> 64)};
  acc_umax128(a, t);
}

// float helpers.  The translation unit is compiled with -ffp-contract=off so a*b+c rounds twice like
// the CPU path; these wrappers keep that explicit at the call sites.
CDEV double fp_add(double a, double b) { return __dadd_rn(a, b); }
CDEV double fp_sub(double a, double b) { return __dsub_rn(a, b); }
CDEV double fp_mul(double a, double b) { return __dmul_rn(a, b); }
CDEV double fp_div(double a, double b) { return a / b; }
CDEV float fp_add(float a, float b) { return __fadd_rn(a, b); }
CDEV float fp_sub(float a, float b) { return __fsub_rn(a, b); }
CDEV float fp_mul(float a, float b) { return __fmul_rn(a, b); }
CDEV float fp_div(float a, float b) { return a / b; }
// IEEE-754 totalOrder keys (arrow-ord compares floats this way: -NaN < -inf < … < -0 < +0 < … < +inf < NaN)
CDEV i64 f64_total_key(double d) {
  i64 b = __double_as_longlong(d);
  return b ^ (i64)((u64)(b >> 63) >> 1);
}
CDEV i32 f32_total_key(float f) {
  i32 b = __float_as_int(f);
  return b ^ (i32)((u32)(b >> 31) >> 1);
}
// NormalizeNaNAndZero (spark-expr/src/math_funcs/internal/normalize_nan.rs:93-101): NaN → canonical NaN, -0.0 → 0.0
CDEV double normalize_nan_zero_f64(double d) {
  if (d != d) return __longlong_as_double(0x7ff8000000000000ll);
  return d == 0.0 ? 0.0 : d;
}
CDEV float normalize_nan_zero_f32(float f) {
  if (f != f) return __int_as_float(0x7fc00000);
  return f == 0.0f ? 0.0f : f;
}
CDEV void acc_fmin64(u64* a, const u64* b) {
  if (f64_total_key(__longlong_as_double((i64)b[0])) < f64_total_key(__longlong_as_double((i64)a[0]))) a[0] = b[0];
}
CDEV void acc_fmax64(u64* a, const u64* b) {
  if (f64_total_key(__longlong_as_double((i64)b[0])) > f64_total_key(__longlong_as_double((i64)a[0]))) a[0] = b[0];
}

// ---------------------------------------------------------------------------------------------
// Exact Float64 sums.  The reference adds doubles one after the other in row order (agg_funcs/avg.rs:239-280, DataFusion's sum), so
// its result depends on batch and partition boundaries; a parallel sum cannot reproduce one particular order.  Instead every value is
// converted to FIXED POINT — an integer multiple of 2^s — and summed with the integer machinery (192-bit accumulators, the 43-bit limbs
// of the grouped path): integer addition is associative, so the result is the same for every grid size and every order, and it is the
// EXACT real sum whenever all addends lie in the window  2^s ≤ lowest set bit,  |x| < 2^(s + kFixW).  The final state is that exact sum
// rounded once to nearest-even, i.e. within half an ULP of the true sum — which is what any ordering of the reference's additions
// approximates within its own (n−1)·ε·Σ|x| error.  The executor picks s per sum from the exponents it observes (tracked below) and
// re-runs a chunk whose values did not fit; addends whose low bits fall below 2^s are truncated toward zero (only when one sum spans
// more than kFixW − 53 = 105 binary orders of magnitude), which bounds the error by rows · 2^s.
// ±inf and NaN do not enter the fixed-point sum; a class word remembers them and the result follows IEEE (inf − inf = NaN).
// ---------------------------------------------------------------------------------------------
constexpr int kFixW = 158;      // 158 value bits + 33 bits of row-count headroom + sign = 192
CDEV int fix_scale(i64 packed, int f) { return (int)(i16)(u16)((u64)packed >> (16 * f)); }
// finite non-zero x = ±m · 2^q with 0 < m < 2^53
CDEV bool f64_parts(double x, u64& m, int& q) {
  const u64 b = (u64)__double_as_longlong(x);
  const int be = (int)((b >> 52) & 0x7ff);
  const u64 frac = b & ((1ull << 52) - 1);
  if (be == 0x7ff) return false;
  if (be == 0) { m = frac; q = -1074; } else { m = frac | (1ull << 52); q = be - 1075; }
  return m != 0;
}
CDEV u64 f64_class(double x) {            // 1 = +inf, 2 = −inf, 4 = NaN
  const u64 b = (u64)__double_as_longlong(x);
  if (((b >> 52) & 0x7ff) != 0x7ff) return 0;
  if (b & ((1ull << 52) - 1)) return 4;
  return (b >> 63) ? 2 : 1;
}
// exponent tracking words, both monotone under an unsigned max (0 = no finite non-zero value yet):
//   hi = 1200 + top  where |x| < 2^top;   lo = 1200 − low  where 2^low is x's lowest set bit
CDEV u64 f64_exp_hi(double x) { u64 m; int q; return f64_parts(x, m, q) ? (u64)(1200 + q + 64 - __builtin_clzll(m)) : 0ull; }
CDEV u64 f64_exp_lo(double x) { u64 m; int q; return f64_parts(x, m, q) ? (u64)(1200 - (q + __builtin_ctzll(m))) : 0ull; }
// limb j (kLimbBits = 43 bits, carrying x's sign) of trunc(x / 2^s)
CDEV u64 f64_fix_limb(double x, int s, int j) {
  u64 m; int q;
  if (!f64_parts(x, m, q)) return 0;
  const int sh = q - s - 43 * j;
  u64 v;
  if (sh >= 43 || sh <= -53) v = 0;
  else if (sh >= 0) v = (m << sh) & ((1ull << 43) - 1);
  else v = (m >> -sh) & ((1ull << 43) - 1);
  return x < 0 ? (u64)(-(i64)v) : v;
}
// acc (192-bit two's complement) += trunc(x / 2^s)
CDEV void acc_feed_fix192(u64* a, double x, int s) {
  u64 m; int q;
  if (!f64_parts(x, m, q)) return;
  u64 t[3] = {0, 0, 0};
  const int sh = q - s;
  if (sh < 0) {
    if (sh > -53) t[0] = m >> -sh;
  } else {
    const int ws = sh >> 6, bs = sh & 63;
    if (ws < 3) t[ws] = m << bs;
    if (bs && ws + 1 < 3) t[ws + 1] = m >> (64 - bs);
  }
  if (x < 0) {
    t[0] = ~t[0]; t[1] = ~t[1]; t[2] = ~t[2];
    if (++t[0] == 0) { if (++t[1] == 0) ++t[2]; }
  }
  acc_add192(a, t);
}
// the accumulated integer t (192-bit) · 2^s → nearest double (ties to even), with the IEEE outcome of any inf / NaN addends
CDEV double fix192_to_f64(const u64* t, int s, u64 cls) {
  if ((cls & 4) || ((cls & 1) && (cls & 2))) return __longlong_as_double(0x7ff8000000000000ll);
  if (cls & 1) return __longlong_as_double(0x7ff0000000000000ll);
  if (cls & 2) return __longlong_as_double((i64)0xfff0000000000000ull);
  u64 w0 = t[0], w1 = t[1], w2 = t[2];
  const bool neg = (w2 >> 63) != 0;
  if (neg) {
    w0 = ~w0; w1 = ~w1; w2 = ~w2;
    if (++w0 == 0) { if (++w1 == 0) ++w2; }
  }
  const int L = w2 ? 192 - __builtin_clzll(w2) : (w1 ? 128 - __builtin_clzll(w1) : (w0 ? 64 - __builtin_clzll(w0) : 0));
  if (L == 0) return 0.0;
  int e_lsb = s + L - 53;                 // weight of the result's last mantissa bit …
  if (e_lsb < -1074) e_lsb = -1074;       // … but never below the subnormal grid
  const int shift = e_lsb - s;
  u64 mant;
  if (shift <= 0) {
    mant = w0;                            // the whole integer fits the mantissa: exact
    e_lsb = s;
  } else {
    // mant = t >> shift, rounded to nearest even on the dropped bits
    const int ws = shift >> 6, bs = shift & 63;
    const u64 w[4] = {w0, w1, w2, 0};
    mant = w[ws] >> bs;
    if (bs) mant |= w[ws + 1] << (64 - bs);
    const int gb = shift - 1;             // guard bit
    const bool guard = (w[gb >> 6] >> (gb & 63)) & 1;
    bool sticky = false;
    for (int k = 0; k < (gb >> 6); k++) sticky |= w[k] != 0;
    sticky |= (w[gb >> 6] & ((1ull << (gb & 63)) - 1)) != 0;
    if (guard && (sticky || (mant & 1))) mant += 1;
  }
  const double r = ldexp((double)mant, e_lsb);   // mant ≤ 2^53: exact conversion, one exact scaling (or overflow to inf)
  return neg ? -r : r;
}

// SumDecimal overflow is prefix-order dependent in the reference (sum_decimal.rs:417-438: once a running
// sum leaves the precision it stays NULL).  A parallel reduction only sees totals, so decide exactness
// from order-independent facts:
//   1. cnt · max|v| ≤ bound            → no prefix can overflow, the total is the answer
//      (max|v| is tracked as one monotone word, amax_enc, so that it reduces with an unsigned max)
//   2. all values share one sign       → prefixes are monotone, overflow ⇔ |total| > bound (192-bit total)
//   3. otherwise                       → cannot be decided without the row order: flag err bit 4
// max|v| travels as ONE monotone word so that it can be reduced with an unsigned max: bit length in the top byte, the leading 56
// bits below it, rounded UP — amax_dec(amax_enc(a)) ≥ a with a relative slack < 2^-55 (0 = no value yet).
CDEV u64 amax_enc(u128 a) {
  if (a == 0) a = 1;
  const u64 hi = (u64)(a >> 64), lo = (u64)a;
  int L = hi ? 128 - __builtin_clzll(hi) : 64 - __builtin_clzll(lo);
  u64 mant;
  if (L <= 56) {
    mant = lo << (56 - L);
  } else {
    const int sh = L - 56;
    mant = (u64)(a >> sh);
    if ((a & ((((u128)1) << sh) - 1)) != 0) mant += 1;
    if (mant >> 56) { mant >>= 1; L += 1; }
  }
  return ((u64)L << 56) | mant;
}
CDEV u128 amax_dec(u64 e) {
  const int L = (int)(e >> 56);
  const u64 mant = e & ((1ull << 56) - 1);
  if (L <= 56) return (u128)(mant >> (56 - L));
  if (L > 128) return ~(u128)0;
  return (u128)mant << (L - 56);
}

CDEV void sum_overflow_decide(const u64* sum192, u64 amax_word, u64 signflags, u64 cnt, u128 bound, bool& ovf,
                              unsigned int* err) {
  ovf = false;
  if (cnt == 0 || amax_word == 0) return;
  // case 1: cnt · max|v| ≤ bound — no prefix of any order can leave the precision
  // (cnt · max|v| ≤ bound without the 128-bit division a quotient would cost per group: the 192-bit product's upper limb must be 0)
  {
    const u128 a = amax_dec(amax_word);
    const u128 p0 = (u128)(u64)a * (u128)cnt, p1 = (u128)(u64)(a >> 64) * (u128)cnt;
    const u128 mid = (p0 >> 64) + (u128)(u64)p1;
    if ((u64)(p1 >> 64) + (u64)(mid >> 64) == 0 && (((u128)(u64)mid << 64) | (u128)(u64)p0) <= bound) return;
  }
  // |total| from the three limbs
  bool neg = (sum192[2] >> 63) != 0;
  u64 l0 = sum192[0], l1 = sum192[1], l2 = sum192[2];
  if (neg) {
    l0 = ~l0; l1 = ~l1; l2 = ~l2;
    l0 += 1;
    if (l0 == 0) { l1 += 1; if (l1 == 0) l2 += 1; }
  }
  bool over = l2 != 0 || ((((u128)l1 << 64) | l0) > bound);
  if (signflags != 3) { ovf = over; return; }  // case 2
  if (over) { ovf = true; return; }            // mixed signs but even the total is out of range
  atomicOr(err, 16u);                          // case 3
}

// ---------------------------------------------------------------------------------------------
// Kernel template A — ungrouped aggregate over a fused scan→filter→project pipeline.
//   P::R            rows per thread per tile (tile = 256·R rows)
//   P::NW           accumulator words
//   P::init(acc), P::tile(prm, base, n, acc), P::combine(a, b), P::finalize(prm, acc)
// Each block grid-strides over tiles, keeps its accumulator in registers, reduces across the wave
// with shuffles, across waves through LDS, and writes ONE partial per block.  A second 1-block
// launch folds the partials and writes the aggregate state row (Partial mode) or value (Final).
//   prm.out[0] = partials (gridDim.x × NW u64);  prm.iarg[0] = number of partials
// ---------------------------------------------------------------------------------------------
template <class P>
CDEV void block_reduce_acc(u64* acc) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    u64 other[P::NW];
#pragma unroll
    for (int k = 0; k < P::NW; k++) other[k] = shfl_xor_u64(acc[k], m);
    P::combine(acc, other);
  }
  __shared__ u64 s_part[kBlock / kWave][P::NW];
  const int lane = lane_id(), wv = wave_id();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < P::NW; k++) s_part[wv][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kBlock / kWave; w++) P::combine(acc, s_part[w]);
  }
}

template <class P>
CDEV void agg_nogroup_body(const CometKParams& prm) {
  u64 acc[P::NW];
  P::init(acc);
  const i64 n = prm.n;
  const i64 tile = (i64)P::R * kBlock;
  const i64 stride = (i64)gridDim.x * tile;
  if constexpr (P::PIPELINED) {
    // software pipeline: tile t+1's first-stage loads are issued before tile t is computed
    typename P::L cur, nxt;
    i64 base = (i64)blockIdx.x * tile;
    if (base < n) P::tile_load(prm, base, n, cur);
    for (; base < n; base += stride) {
      if (base + stride < n) P::tile_load(prm, base + stride, n, nxt);
      P::tile(prm, base, n, cur, acc);
      cur = nxt;
    }
  } else {
    for (i64 base = (i64)blockIdx.x * tile; base < n; base += stride) P::tile(prm, base, n, acc);
  }
  block_reduce_acc<P>(acc);
  if (threadIdx.x == 0) {
    u64* dst = (u64*)prm.out[0] + (i64)blockIdx.x * P::NW;
#pragma unroll
    for (int k = 0; k < P::NW; k++) dst[k] = acc[k];
    P::kexport(prm, acc);   // exponent range of exact float sums → aux words (the executor sizes the fixed-point window from them)
  }
}

template <class P>
CDEV void agg_nogroup_final_body(const CometKParams& prm) {
  u64 acc[P::NW];
  P::init(acc);
  const i64 np = prm.iarg[0];
  const u64* src = (const u64*)prm.out[0];
  for (i64 b = threadIdx.x; b < np; b += kBlock) {
    u64 t[P::NW];
#pragma unroll
    for (int k = 0; k < P::NW; k++) t[k] = src[b * P::NW + k];
    P::combine(acc, t);
  }
  block_reduce_acc<P>(acc);
  if (threadIdx.x == 0) P::finalize(prm, acc);
}

// ---------------------------------------------------------------------------------------------
// Kernel template B — filter (+project) with ORDER-PRESERVING compaction, as FilterExec keeps row
// order (reference: planner.rs:1230-1247 → DataFusion FilterExec).  Three launches:
//   1. filter_mask:   predicate → one ballot word per 64 rows + per-tile survivor count
//   2. tile_scan:     exclusive scan of tile counts (single block)
//   3. filter_emit:   survivors evaluate the projection and scatter to their dense position
//   P::keep(prm, i)            predicate is TRUE and valid for row i (loads only what it needs)
//   P::emit(prm, i, pos)       evaluate outputs for row i, store at dense index pos
//   prm.out[0] = mask words (u64, ceil(n/64)); prm.out[1] = tile counts/offsets (u64, ntiles+1)
// ---------------------------------------------------------------------------------------------
constexpr int kMaskTileWords = 16;                 // 16 ballot words = 1024 rows per tile
constexpr int kMaskTileRows = kMaskTileWords * 64;

template <class P>
CDEV void filter_mask_body(const CometKParams& prm) {
  const i64 n = prm.n;
  u64* mask = (u64*)prm.out[0];
  u64* counts = (u64*)prm.out[1];
  const i64 ntiles = (n + kMaskTileRows - 1) / kMaskTileRows;
  __shared__ u32 s_cnt;
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    u32 local = 0;
#pragma unroll
    for (int r = 0; r < kMaskTileRows / kBlock; r++) {
      i64 i = t * kMaskTileRows + r * kBlock + threadIdx.x;
      bool k = (i < n) && P::keep(prm, i);
      u64 b = __ballot(k);
      if (lane_id() == 0) {
        i64 w = i >> 6;
        if (w * 64 < n) mask[w] = b;
        local += (u32)__popcll(b);
      }
    }
    if (lane_id() == 0) atomicAdd(&s_cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) counts[t] = s_cnt;
    __syncthreads();
  }
}

// exclusive scan over counts[0..ntiles) in place; counts[ntiles] = total.  One block; every thread owns kScanItems
// consecutive elements per round, so a round covers 4096 elements between barriers.
constexpr int kScanItems = 16;
CDEV void tile_scan_body(u64* counts, i64 ntiles) {
  __shared__ u64 s_wave[kBlock / kWave];
  __shared__ u64 s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const i64 round = (i64)kBlock * kScanItems;
  for (i64 base = 0; base < ntiles; base += round) {
    const i64 first = base + (i64)threadIdx.x * kScanItems;
    u64 v[kScanItems];
    u64 sum = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
      v[j] = first + j < ntiles ? counts[first + j] : 0;
      sum += v[j];
    }
    u64 x = sum;  // inclusive scan of the per-thread sums within the wave
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      u32 lo = __shfl_up((u32)x, d, kWave), hi = __shfl_up((u32)(x >> 32), d, kWave);
      u64 y = ((u64)hi << 32) | lo;
      if (lane_id() >= d) x += y;
    }
    if (lane_id() == kWave - 1) s_wave[wave_id()] = x;
    __syncthreads();
    u64 woff = 0;
    for (int w = 0; w < wave_id(); w++) woff += s_wave[w];
    const u64 carry = s_carry;
    u64 run = carry + woff + x - sum;
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
      if (first + j < ntiles) counts[first + j] = run;
      run += v[j];
    }
    __syncthreads();
    if (threadIdx.x == kBlock - 1) s_carry = carry + woff + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[ntiles] = s_carry;
}

template <class P>
CDEV void filter_emit_body(const CometKParams& prm) {
  const i64 n = prm.n;
  const u64* mask = (const u64*)prm.out[0];
  const u64* offs = (const u64*)prm.out[1];
  const i64 ntiles = (n + kMaskTileRows - 1) / kMaskTileRows;
  __shared__ u32 s_pref[kMaskTileWords];
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (threadIdx.x < kMaskTileWords) {
      i64 w = t * kMaskTileWords + threadIdx.x;
      u32 c = (w * 64 < n) ? (u32)__popcll(mask[w]) : 0;
      s_pref[threadIdx.x] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      u32 run = 0;
#pragma unroll
      for (int k = 0; k < kMaskTileWords; k++) { u32 c = s_pref[k]; s_pref[k] = run; run += c; }
    }
    __syncthreads();
    const u64 tile_off = offs[t];
#pragma unroll
    for (int r = 0; r < kMaskTileRows / kBlock; r++) {
      i64 i = t * kMaskTileRows + r * kBlock + threadIdx.x;
      if (i < n) {
        int wi = r * (kBlock / 64) + wave_id();
        u64 m = mask[i >> 6];
        int l = lane_id();
        if ((m >> l) & 1) {
          u64 below = m & ((1ull << l) - 1);
          i64 pos = (i64)(tile_off + s_pref[wi] + (u32)__popcll(below));
          P::emit(prm, i, pos);
        }
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel template B' — filter (+project) with ORDER-PRESERVING compaction in ONE pass (decoupled look-back).
// Every tile (P::R·256 consecutive rows) is claimed through a ticket counter, so a tile's predecessors are always held by blocks
// that are already running: evaluate the predicate (ballot word per wave and row slot), scan the tile's 32 wave counts in LDS,
// publish the tile's survivor count in its status word, look back over the predecessors' words (a whole wave reads 64 of them at a
// time) until one carries an inclusive prefix, publish this tile's inclusive prefix, then the survivors evaluate the projection and
// store at their dense position.  Every input column is fetched once (the projection's loads of predicate columns hit the cache the
// predicate just filled); nothing but the outputs is written.  Status word = flag (bits 63..62: 1 aggregate, 2 inclusive prefix) |
// count; one relaxed agent-scope atomic publishes flag and value together, so no fence is needed (the per-XCD L2s are bypassed
// for these words only).
//   prm.out[0] = status words (u64 × ntiles, zeroed);  prm.out[1] = { u32 ticket; u32 pad; u64 total } (zeroed)
// ---------------------------------------------------------------------------------------------
constexpr u64 kTileAgg = 1ull << 62, kTileIncl = 2ull << 62, kTileVal = (1ull << 62) - 1;

//   P::R                                   row slots per thread (tile = R·256 consecutive rows; slot r of thread t is row base + r·256 + t)
//   P::keep_tile(prm, base, n, k[R])       k[r] = the predicate is TRUE and valid for slot r (staged: later conjuncts load only for survivors)
//   P::emit_tile(prm, k[R], idx[R], pos[R]) evaluate the outputs of the surviving slots and store them at pos[r]
template <class P>
CDEV void filter_fused_body(const CometKParams& prm) {
  constexpr int R = P::R;
  static_assert(R <= 16, "one wave scans the R·4 (slot, wave) counts of a tile");
  constexpr i64 kRows = (i64)R * kBlock;
  const i64 n = prm.n;
  u64* status = (u64*)prm.out[0];
  u32* ticket = (u32*)prm.out[1];
  u64* total_out = (u64*)prm.out[1] + 1;
  const i64 ntiles = (n + kRows - 1) / kRows;
  constexpr int NC = R * (kBlock / kWave);             // (slot, wave) counts per tile, in row order (≤ 32)
  __shared__ u32 s_cnt[NC];
  __shared__ u32 s_tile;
  __shared__ u64 s_excl;
  const int lane = lane_id(), wv = wave_id();
  const u64 lt = (1ull << lane) - 1;
  for (;;) {
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const i64 tile = (i64)s_tile;
    if (tile >= ntiles) break;
    const i64 base = tile * kRows;
    bool k[R];
    P::keep_tile(prm, base, n, k);
    u32 below[(R + 3) / 4] = {};                       // 8 bits per slot: survivors in lower lanes of my wave
#pragma unroll
    for (int r = 0; r < R; r++) {
      const u64 b = __ballot(k[r]);
      if (lane == 0) s_cnt[r * (kBlock / kWave) + wv] = (u32)__popcll(b);
      below[r >> 2] |= (u32)__popcll(b & lt) << (8 * (r & 3));
    }
    __syncthreads();
    if (wv == 0) {
      u32 c = lane < NC ? s_cnt[lane] : 0u;
      u32 x = c;                                       // inclusive scan across the first NC lanes
#pragma unroll
      for (int d = 1; d < NC; d <<= 1) {
        u32 y = __shfl_up(x, d, kWave);
        if (lane >= d) x += y;
      }
      if (lane < NC) s_cnt[lane] = x - c;
      const u64 tile_total = (u64)__shfl(x, NC - 1, kWave);
      u64 excl = 0;
      if (tile == 0) {
        if (lane == 0) __hip_atomic_store(&status[0], kTileIncl | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (lane == 0) __hip_atomic_store(&status[tile], kTileAgg | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        i64 look = tile - 1;                           // lane l inspects tile look - l
        for (;;) {
          const i64 t = look - lane;
          u64 st;
          do {
            st = t >= 0 ? __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kTileIncl;   // before tile 0: prefix 0
          } while (__ballot((st >> 62) == 0) != 0);    // a predecessor has not published yet: it is running, poll again
          const u64 incl = __ballot((st >> 62) == 2);
          const int first = incl ? __ffsll((unsigned long long)incl) - 1 : kWave;   // nearest tile carrying an inclusive prefix
          u64 v = lane <= first ? (st & kTileVal) : 0ull;
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
          excl += v;
          if (incl) break;
          look -= kWave;
        }
        if (lane == 0) __hip_atomic_store(&status[tile], kTileIncl | (excl + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (lane == 0) {
        s_excl = excl;
        if (tile == ntiles - 1) *total_out = excl + tile_total;
      }
    }
    __syncthreads();
    const u64 tile_off = s_excl;
    i64 idx[R], pos[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      idx[r] = base + (i64)r * kBlock + threadIdx.x;
      pos[r] = (i64)(tile_off + s_cnt[r * (kBlock / kWave) + wv] + ((below[r >> 2] >> (8 * (r & 3))) & 0xff));
    }
    P::emit_tile(prm, k, idx, pos);
    __syncthreads();
  }
}

// Pure projection (no filter): dense, position = row.
template <class P>
CDEV void project_body(const CometKParams& prm) {
  const i64 n = prm.n;
  for (i64 i = (i64)blockIdx.x * kBlock + threadIdx.x; i < n; i += (i64)gridDim.x * kBlock) P::emit(prm, i, i);
}

// byte-per-row validity → Arrow bitmap (n rows).  prm.out[0]=bytes, prm.out[1]=bitmap
CDEV void pack_validity_body(const u8* bytes, u8* bitmap, i64 n) {
  // one lane per row, ballot gives 64 bits = 8 bitmap bytes
  const i64 nround = (n + 63) & ~63ll;
  for (i64 i = (i64)blockIdx.x * kBlock + threadIdx.x; i < nround; i += (i64)gridDim.x * kBlock) {
    bool v = (i < n) && bytes[i] != 0;
    u64 b = __ballot(v);
    if (lane_id() == 0) {
      i64 byte0 = i >> 3;
      i64 nbytes = (n + 7) >> 3;
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (byte0 + k < nbytes) bitmap[byte0 + k] = (u8)(b >> (8 * k));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel template C — grouped hash aggregate.
// Two-level open addressing: a per-block LDS table absorbs the hot groups (low cardinality: Q1 has
// 4), rows whose group does not fit go straight to the global table; at block exit the LDS table
// is merged into the global table.  All accumulator primitives are commutative (wrapping adds,
// max, or), so the merge order is irrelevant and integer results are bit-exact.
//   P::NK   key words (u64) per group — key columns packed by codegen, NULLs carried in a flag word
//   P::NW   accumulator words per group
//   P::LDS_CAP  LDS table capacity (power of two, 0 = disabled)
//   P::tile_grouped(prm, base, n, tbl)   calls tbl.update(key, vals, active) per row
//   P::merge_word(k, dst, src)  combine op for accumulator word k (by primitive kind)
// Global table layout (prm.out[0]): capacity = prm.iarg[0] slots of
//   { u32 state; u32 pad; u64 key[NK]; u64 acc[NW] }
//   prm.out[1] = u32 error flag (1 = table full)
// ---------------------------------------------------------------------------------------------
constexpr u32 kSlotEmpty = 0, kSlotBusy = 1, kSlotReady = 2;

template <int NK>
CDEV u64 hash_key(const u64* key) {
  u64 h = 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int k = 0; k < NK; k++) {
    h ^= key[k];
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 32;
  }
  return h;
}

// which atomic combine an accumulator word uses
enum AccOp : int { OP_ADD = 0, OP_ADDC_LO = 1, OP_ADDC_HI = 2, OP_UMAX = 3, OP_OR = 4, OP_FADD = 5, OP_IMIN = 6, OP_IMAX = 7, OP_ADDC_MID = 8 };

template <int NK, int NW>
struct Slot {
  u32 state;
  u32 pad;
  u64 key[NK];
  u64 acc[NW];
};

// find-or-insert in the global table.  No lane ever waits for another lane while holding a claim, so lanes of one
// wave probing the same slot cannot deadlock.  A lane that finds the slot BUSY goes round the probe loop (`continue`) — it must not spin in
// place: an inner `while (state == BUSY) reload` can be scheduled before the publishing branch of the winner IN THE SAME WAVE and then
// never ends (tried with a claim-first variant in round 4: the aggregate tests hung on the GPU).
//
// Memory ordering without agent-scope fences: an acquire/release pair at agent scope costs an L2 invalidate
// (buffer_inv sc1) and an L2 write-back (buffer_wbl2 sc1) PER ROW on CDNA3/4 — the per-XCD L2s are not coherent for
// ordinary accesses — which made the high-cardinality path ~10× slower than its atomics.  Instead EVERY access to a
// slot is a relaxed agent-scope atomic (sc1: performed at the coherence point, coherent per location), and the
// owner orders "key + initial accumulators" before "state = READY" by waiting for its stores to be acknowledged
// (s_waitcnt vmcnt(0)) in between.  A reader only looks at the key after it has seen READY (control dependency).
template <int NK, int NW, class InitFn>
CDEV Slot<NK, NW>* table_find_or_insert(Slot<NK, NW>* tbl, u64 cap, const u64* key, InitFn init, u64 max_probes,
                                        u32* insert_counter = nullptr, bool* fresh = nullptr) {
  u64 h = hash_key<NK>(key) & (cap - 1);
  for (u64 probes = 0; probes < max_probes;) {
    Slot<NK, NW>* s = &tbl[h];
    u32 st = __hip_atomic_load(&s->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st == kSlotEmpty) {
      u32 expected = kSlotEmpty;
      if (__hip_atomic_compare_exchange_strong(&s->state, &expected, kSlotBusy, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)) {
        u64 acc0[NW];
        init(acc0);
#pragma unroll
        // (an empty slot is all zeroes — the host clears tables with a memset — so a word that is zero is not stored: every access to a slot is
        // an operation at the device's coherence point, and their NUMBER bounds a table with as many groups as rows; the null-flag key word
        // and the upper limbs of a sum are zero nearly always)
        for (int k = 0; k < NK; k++) if (key[k]) __hip_atomic_store(&s->key[k], key[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < NW; k++) if (acc0[k]) __hip_atomic_store(&s->acc[k], acc0[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&s->state, kSlotReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (insert_counter) (*insert_counter)++;   // per-lane count, added to the table's group counter once per wave
        if (fresh) *fresh = true;
        return s;
      }
      continue;  // lost the race: re-read this slot
    }
    if (st == kSlotBusy) continue;  // owner is publishing the key; re-read
    __asm__ volatile("" ::: "memory");
    bool eq = true;
#pragma unroll
    for (int k = 0; k < NK; k++) eq &= (__hip_atomic_load(&s->key[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == key[k]);
    if (eq) return s;
    h = (h + 1) & (cap - 1);
    probes++;
  }
  return nullptr;
}

// ---- atomics that work on both global (agent scope) and LDS (address_space(3), workgroup scope) words.
// Going through address_space(3) pointers matters: a generic pointer makes the compiler emit FLAT
// instructions for the LDS table, and a FLAT access has to wait for every outstanding global load
// (vmcnt AND lgkmcnt), which serialises the table probe behind the column loads.
#define COMET_LDS __attribute__((address_space(3)))

template <int SCOPE, class WordPtr>
CDEV void atomic_add_limbs(WordPtr dst, const u64* v, int limbs) {
  // multi-limb wrapping add with explicit carries: each limb add returns the old value, so the carry out of
  // THIS add is exact, and carries commute — after all updates the limbs equal the true sum mod 2^(64·L).
  u64 carry = 0;
  for (int k = 0; k < limbs; k++) {
    u64 add = v[k] + carry;
    u64 c = (add < carry) ? 1 : 0;  // v[k] + carry wrapped (only when v[k] = 2^64-1 and carry = 1)
    if (add != 0) {
      u64 old = __hip_atomic_fetch_add(dst + k, add, __ATOMIC_RELAXED, SCOPE);
      if (old + add < old) c = 1;
    }
    carry = c;
  }
}
template <int SCOPE, class WordPtr>
CDEV void atomic_cas_combine_f64(WordPtr dst, u64 v, int op) {
  u64 old = __hip_atomic_load(dst, __ATOMIC_RELAXED, SCOPE);
  while (true) {
    u64 nv = old;
    if (op == OP_FADD) nv = (u64)__double_as_longlong(fp_add(__longlong_as_double((i64)old), __longlong_as_double((i64)v)));
    else { u64 t[1] = {old}; u64 w[1] = {v}; if (op == OP_IMIN) acc_fmin64(t, w); else acc_fmax64(t, w); nv = t[0]; }
    if (nv == old) return;
    if (__hip_atomic_compare_exchange_strong(dst, &old, nv, __ATOMIC_RELAXED, __ATOMIC_RELAXED, SCOPE)) return;
  }
}

// accumulator word kinds of the grouped path (P::op(k))
enum GOp : int { G_ADD64 = 0, G_ADD128 = 1, G_ADD192 = 2, G_UMAX64 = 3, G_OR64 = 4, G_FADD64 = 5, G_IMIN64 = 6, G_IMAX64 = 7, G_FMIN64 = 8, G_FMAX64 = 9, G_CONT = 10 };

template <class T> struct as_i64;
template <> struct as_i64<u64*> { typedef i64* type; };
template <> struct as_i64<COMET_LDS u64*> { typedef COMET_LDS i64* type; };

// apply one contribution (NW words) to a slot's accumulators
template <class P, int SCOPE, class WordPtr>
CDEV void slot_apply(WordPtr acc, const u64* val) {
#pragma unroll
  for (int k = 0; k < P::NW; k++) {
    switch (P::op(k)) {
      case G_ADD64: if (val[k]) __hip_atomic_fetch_add(acc + k, val[k], __ATOMIC_RELAXED, SCOPE); break;
      case G_ADD128: atomic_add_limbs<SCOPE>(acc + k, val + k, 2); break;
      case G_ADD192: atomic_add_limbs<SCOPE>(acc + k, val + k, 3); break;
      case G_UMAX64: if (val[k]) __hip_atomic_fetch_max(acc + k, val[k], __ATOMIC_RELAXED, SCOPE); break;
      case G_OR64: if (val[k]) __hip_atomic_fetch_or(acc + k, val[k], __ATOMIC_RELAXED, SCOPE); break;
      case G_IMIN64: __hip_atomic_fetch_min((typename as_i64<WordPtr>::type)(acc + k), (i64)val[k], __ATOMIC_RELAXED, SCOPE); break;
      case G_IMAX64: __hip_atomic_fetch_max((typename as_i64<WordPtr>::type)(acc + k), (i64)val[k], __ATOMIC_RELAXED, SCOPE); break;
      case G_FADD64: atomic_cas_combine_f64<SCOPE>(acc + k, val[k], OP_FADD); break;
      case G_FMIN64: atomic_cas_combine_f64<SCOPE>(acc + k, val[k], OP_IMIN); break;
      case G_FMAX64: atomic_cas_combine_f64<SCOPE>(acc + k, val[k], OP_IMAX); break;
      default: break;  // G_CONT: continuation limb of a multi-limb add
    }
  }
}

// the same contribution applied to accumulators nobody else can see yet (registers): what slot_apply does, without atomics
template <class P>
CDEV void slot_apply_private(u64* acc, const u64* val) {
#pragma unroll
  for (int k = 0; k < P::NW; k++) {
    switch (P::op(k)) {
      case G_ADD64: acc[k] += val[k]; break;
      case G_ADD128: case G_ADD192: {
        const int limbs = P::op(k) == G_ADD128 ? 2 : 3;
        u64 carry = 0;
        for (int j = 0; j < limbs; j++) {
          const u64 a = acc[k + j], b = val[k + j];
          const u64 t = a + b, r = t + carry;
          carry = (t < a || r < t) ? 1 : 0;
          acc[k + j] = r;
        }
        break;
      }
      case G_UMAX64: if (val[k] > acc[k]) acc[k] = val[k]; break;
      case G_OR64: acc[k] |= val[k]; break;
      case G_IMIN64: if ((i64)val[k] < (i64)acc[k]) acc[k] = val[k]; break;
      case G_IMAX64: if ((i64)val[k] > (i64)acc[k]) acc[k] = val[k]; break;
      case G_FADD64: acc[k] = (u64)__double_as_longlong(fp_add(__longlong_as_double((i64)acc[k]), __longlong_as_double((i64)val[k]))); break;
      case G_FMIN64: acc_fmin64(acc + k, val + k); break;
      case G_FMAX64: acc_fmax64(acc + k, val + k); break;
      default: break;  // G_CONT
    }
  }
}
// A slot's first contribution travels with its key: the lane that wins an empty slot stores identity ⊕ its own value before it publishes the
// slot, instead of the identity followed by NW device-scope read-modify-writes (each of them a round trip to the coherence point, the limbs
// of a wide sum one after the other).  With as many groups as rows — SF100 Q3's 1.13 M — that is every row.
template <class P>
struct SlotInitWith {
  const u64* val;
  CDEV void operator()(u64* acc) const { P::init(acc); slot_apply_private<P>(acc, val); }
};

// ---------------------------------------------------------------------------------------------
// Block-level accumulation in LDS, "carry-save" form.
// A per-row 128-bit add into a shared accumulator would need a returning atomic per limb (carry) and, for a
// low-cardinality GROUP BY, 64 lanes hammering the same address.  Instead every integer sum is split into
// LIMBS of kLimbBits bits, each limb accumulated in its own 64-bit LDS word by a NON-returning ds_add_u64:
// no carries, no dependent LDS round trips.  A block adds at most kMaxRowsPerBlock rows, so a word cannot
// overflow; the limbs are recombined (P::fold) once per block.  The first P::GC groups a block meets get
// P::COPIES private copies of their accumulator words (copy = lane % COPIES, layout [group][word][copy] so a
// wave's lanes hit consecutive banks) which removes the same-address serialisation; other groups use the
// single copy in their LDS table slot; groups that do not fit the LDS table go to the global table.
// ---------------------------------------------------------------------------------------------
constexpr int kLimbBits = 43;
constexpr i64 kMaxRowsPerBlock = (i64)1 << 19;  // 2^19 rows · 2^43 per limb < 2^63

// limb j (of nl) of a signed value: lower limbs unsigned kLimbBits bits, top limb signed
CDEV u64 limb_of(i128 v, int j, int nl) {
  if (j == nl - 1) return (u64)(i64)(v >> (kLimbBits * j));
  return (u64)(v >> (kLimbBits * j)) & ((1ull << kLimbBits) - 1);
}
// Σ_j sext(w[j]) · 2^(43·j) as a 192-bit two's-complement number
CDEV void limbs_to_i192(const u64* w, int nl, u64* out3) {
  out3[0] = out3[1] = out3[2] = 0;
  for (int j = 0; j < nl; j++) {
    i64 sw = (i64)w[j];
    u64 t[3] = {(u64)sw, sw < 0 ? ~0ull : 0ull, sw < 0 ? ~0ull : 0ull};
    int sh = kLimbBits * j;
    // 192-bit left shift by sh (< 192)
    u64 r[3] = {0, 0, 0};
    int ws = sh >> 6, bs = sh & 63;
    for (int k = 2; k >= 0; k--) {
      int src = k - ws;
      if (src < 0) continue;
      u64 v = t[src] << bs;
      if (bs && src > 0) v |= t[src - 1] >> (64 - bs);
      r[k] = v;
    }
    acc_add192(out3, r);
  }
}

// LDS slot (level 1): limb-form accumulators; padded to an ODD number of 8-byte words so consecutive slots
// start on different banks.
template <int NK, int NPW>
struct LSlot {
  u32 state;
  u32 ord;
  u64 key[NK];
  u64 acc[NPW];
  u64 pad[((1 + NK + NPW) % 2 == 0) ? 1 : 0];
};

template <class P>
struct GroupCtx {
  COMET_LDS LSlot<P::NK, P::NPW>* lds;  // per-block table (LDS)
  COMET_LDS u32* lds_count;             // number of groups inserted into the LDS table (dense ordinals)
  COMET_LDS u32* ord_slot;              // ordinal (< P::GC) → LDS slot index
  COMET_LDS u64* priv;                  // [GC][NPW][COPIES] private accumulator copies
  Slot<P::NK, P::NW>* glb;
  u64 glb_cap;
  unsigned int* err;                    // err[0] flags; ((u64*)err)[1] = number of groups in the global table
  // the partition passes of a merging aggregate (template C'', below): 0 = none, 1 = count my row's partition, 2 = write my row's record
  int part_mode = 0;
  COMET_LDS u32* part = nullptr;        // the block's per-partition counters (mode 1) / cursors (mode 2)
  u64* recs = nullptr;                  // mode 2: records { key[NK], val[NW] }, partition-major
  u32 np = 0;
  mutable u32 inserted = 0;             // groups this lane inserted into the global table (flushed by flush_inserted)
  CDEV u32* counter() const { return &inserted; }
  CDEV bool table_full() const { return (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 32u) != 0; }
  CDEV void set_full() const { if (!table_full()) atomicOr(err, 32u); }
};

// one atomic per wave instead of one per inserted group (a single hot address serialises at the L2)
CDEV void flush_inserted(unsigned int* err, u32 inserted) {
  u32 v = inserted;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  if (lane_id() == 0 && v) atomicAdd((unsigned long long*)err + 1, (unsigned long long)v);
}
constexpr u64 kMaxGlobalProbes = 128;  // beyond this the table counts as full (host grows it and re-runs)
constexpr u32 kNoOrdinal = 0xffffffffu;

template <class P>
struct SlotInit {
  CDEV void operator()(u64* acc) const { P::init(acc); }
};

// one private word ← one contribution (non-returning LDS atomic)
template <class P>
CDEV void lds_word_apply(COMET_LDS u64* w, int k, u64 v) {
  switch (P::pop(k)) {
    case G_ADD64: if (v) __hip_atomic_fetch_add(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
    case G_UMAX64: if (v) __hip_atomic_fetch_max(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
    case G_OR64: if (v) __hip_atomic_fetch_or(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
    case G_IMIN64: __hip_atomic_fetch_min((COMET_LDS i64*)w, (i64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
    case G_IMAX64: __hip_atomic_fetch_max((COMET_LDS i64*)w, (i64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
    case G_FADD64: __hip_atomic_fetch_add((COMET_LDS double*)w, __longlong_as_double((i64)v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
    default: break;
  }
}
// plain (non-atomic) combine of two private words of kind P::pop(k)
template <class P>
CDEV u64 pword_combine(int k, u64 a, u64 b) {
  switch (P::pop(k)) {
    case G_ADD64: return a + b;
    case G_UMAX64: return a > b ? a : b;
    case G_OR64: return a | b;
    case G_IMIN64: return (i64)b < (i64)a ? b : a;
    case G_IMAX64: return (i64)b > (i64)a ? b : a;
    case G_FADD64: return (u64)__double_as_longlong(fp_add(__longlong_as_double((i64)a), __longlong_as_double((i64)b)));
    default: return a;
  }
}

// LDS-table lookup: workgroup scope, relaxed — the table lives and dies inside one block, so no
// agent-scope acquire/release (= L1/L2 invalidate + write-back per row) is needed here.
// Returns the slot and its dense ordinal (order of first insertion within the block).
template <class P>
CDEV COMET_LDS LSlot<P::NK, P::NPW>* lds_find_or_insert(const GroupCtx<P>& g, const u64* key, u32& ordinal) {
  typedef COMET_LDS LSlot<P::NK, P::NPW> S;
  u32 h = (u32)hash_key<P::NK>(key) & (u32)(P::LDS_CAP - 1);
  for (int probes = 0; probes < 16;) {
    S* s = g.lds + h;
    u32 st = __hip_atomic_load(&s->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (st == kSlotEmpty) {
      u32 expected = kSlotEmpty;
      if (__hip_atomic_compare_exchange_strong(&s->state, &expected, kSlotBusy, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP)) {
#pragma unroll
        for (int k = 0; k < P::NK; k++) s->key[k] = key[k];
#pragma unroll
        for (int k = 0; k < P::NPW; k++) s->acc[k] = P::pidentity(k);
        u32 ord = __hip_atomic_fetch_add(g.lds_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        s->ord = ord;
        if (ord < (u32)P::GC) g.ord_slot[ord] = h;
        __hip_atomic_store(&s->state, kSlotReady, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        ordinal = ord;
        return s;
      }
      continue;  // lost the race: re-read this slot
    }
    if (st == kSlotBusy) continue;  // owner is publishing; it never waits on us
    bool eq = true;
#pragma unroll
    for (int k = 0; k < P::NK; k++) eq &= (s->key[k] == key[k]);
    if (eq) {
      ordinal = s->ord;
      return s;
    }
    h = (h + 1) & (u32)(P::LDS_CAP - 1);
    probes++;
  }
  ordinal = kNoOrdinal;
  return nullptr;
}

// One row per lane: key words + limb-form contribution pv[NPW].
template <class P>
CDEV void group_update(const GroupCtx<P>& g, bool active, const u64* key, const u64* pv) {
  if (!active) return;
#if defined(COMET_EXPERIMENT) && COMET_EXPERIMENT == 1
  {  // floor measurement: keep key/pv alive, no probe, no update
    u64 x = 0;
    for (int k = 0; k < P::NK; k++) x ^= key[k];
    for (int k = 0; k < P::NPW; k++) x += pv[k];
    if (x == 0x123456789abcdefull) atomicOr(g.err, 128u);
    return;
  }
#endif
  if (g.part_mode) {      // (uniform: a partition pass of template C'')
    const u32 p = (u32)__umul64hi(hash_key<P::NK>(key), (u64)g.np);
    if (g.part_mode == 1) {
      __hip_atomic_fetch_add(g.part + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return;
    }
    u64 val[P::NW];
    P::fold(pv, val);
    const u32 pos = __hip_atomic_fetch_add(g.part + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    u64* rec = g.recs + (u64)pos * (u64)(P::NK + P::NW);
#pragma unroll
    for (int k = 0; k < P::NK; k++) rec[k] = key[k];
#pragma unroll
    for (int k = 0; k < P::NW; k++) rec[P::NK + k] = val[k];
    return;
  }
  u32 ord = kNoOrdinal;
  COMET_LDS LSlot<P::NK, P::NPW>* ls = nullptr;
  if (P::LDS_CAP > 0) ls = lds_find_or_insert<P>(g, key, ord);
  if (ord < (u32)P::GC) {
    COMET_LDS u64* base = g.priv + (u32)(ord * P::NPW) * P::COPIES + (threadIdx.x & (P::COPIES - 1));
#pragma unroll
    for (int k = 0; k < P::NPW; k++) lds_word_apply<P>(base + k * P::COPIES, k, pv[k]);
    return;
  }
  if (ls) {
#pragma unroll
    for (int k = 0; k < P::NPW; k++) lds_word_apply<P>(&ls->acc[k], k, pv[k]);
    return;
  }
  if (g.table_full()) return;   // this pass is void: the host grows the table and re-runs the chunk
  u64 val[P::NW];
  P::fold(pv, val);
  bool fresh = false;
  Slot<P::NK, P::NW>* s = table_find_or_insert<P::NK, P::NW>(g.glb, g.glb_cap, key, SlotInitWith<P>{val}, kMaxGlobalProbes, g.counter(), &fresh);
  if (!s) { g.set_full(); return; }
  if (!fresh) slot_apply<P, __HIP_MEMORY_SCOPE_AGENT>(&s->acc[0], val);
}

// Kernel template C body.  prm.out[0] = global table, prm.iarg[0] = its capacity, prm.out[2] = err/aux words.
template <class P>
CDEV void agg_grouped_body(const CometKParams& prm) {
  typedef Slot<P::NK, P::NW> S;
  typedef LSlot<P::NK, P::NPW> LS;
  __shared__ LS s_tbl[P::LDS_CAP > 0 ? P::LDS_CAP : 1];
  __shared__ u64 s_priv[P::GC * P::NPW * P::COPIES];
  __shared__ u32 s_count;
  __shared__ u32 s_ord_slot[P::GC];
  if (P::LDS_CAP > 0) {
    for (int i = threadIdx.x; i < P::LDS_CAP; i += kBlock) s_tbl[i].state = kSlotEmpty;
  }
  for (int i = threadIdx.x; i < P::GC * P::NPW * P::COPIES; i += kBlock) s_priv[i] = P::pidentity((i / P::COPIES) % P::NPW);
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  GroupCtx<P> g;
  g.lds = (COMET_LDS LS*)s_tbl;
  g.lds_count = (COMET_LDS u32*)&s_count;
  g.ord_slot = (COMET_LDS u32*)s_ord_slot;
  g.priv = (COMET_LDS u64*)s_priv;
  g.glb = (S*)prm.out[0];
  g.glb_cap = (u64)prm.iarg[0];
  g.err = (unsigned int*)prm.out[2];
  u64 kacc[P::NKW > 0 ? P::NKW : 1];  // kernel-level (not per-group) accumulators: value bounds for the overflow proof
  P::kinit(kacc);
  const i64 n = prm.n;
  const i64 tile = (i64)P::R * kBlock;
  const i64 stride = (i64)gridDim.x * tile;
  if constexpr (P::PIPELINED) {
    typename P::L cur, nxt;
    i64 base = (i64)blockIdx.x * tile;
    if (base < n) P::tile_load(prm, base, n, cur);
    for (u32 it = 0; base < n; base += stride, it++) {
      if (base + stride < n) P::tile_load(prm, base + stride, n, nxt);
      P::tile_grouped(prm, base, n, cur, g, kacc);
      cur = nxt;
      // the pass is void once any lane found the table full (the host grows it and re-runs the chunk): stop early
      if ((it < 8 || (it & 7) == 7) && g.table_full()) break;
    }
  } else {
    u32 it = 0;
    for (i64 base = (i64)blockIdx.x * tile; base < n; base += stride, it++) {
      P::tile_grouped(prm, base, n, g, kacc);
      if ((it < 8 || (it & 7) == 7) && g.table_full()) break;
    }
  }
  __syncthreads();
  // level 0 → level 1: sum the private copies into the group's LDS slot (one thread per (group, word))
  const u32 ngrp = s_count < (u32)P::GC ? s_count : (u32)P::GC;
  for (u32 t = threadIdx.x; t < ngrp * P::NPW; t += kBlock) {
    const u32 gi = t / P::NPW, k = t % P::NPW;
    u64 a = P::pidentity(k);
    for (int c = 0; c < P::COPIES; c++) a = pword_combine<P>(k, a, s_priv[(gi * P::NPW + k) * P::COPIES + c]);
    LS* ls = &s_tbl[s_ord_slot[gi]];
    ls->acc[k] = pword_combine<P>(k, ls->acc[k], a);
  }
  __syncthreads();
  // level 1 → level 2: fold limbs to canonical words and merge into the global table
  for (int i = threadIdx.x; i < P::LDS_CAP; i += kBlock) {
    LS* ls = &s_tbl[i];
    if (ls->state == kSlotReady) {
      u64 key[P::NK], pw[P::NPW], val[P::NW];
#pragma unroll
      for (int k = 0; k < P::NK; k++) key[k] = ls->key[k];
#pragma unroll
      for (int k = 0; k < P::NPW; k++) pw[k] = ls->acc[k];
      P::fold(pw, val);
      bool fresh = false;
      S* gs = g.table_full() ? nullptr : table_find_or_insert<P::NK, P::NW>(g.glb, g.glb_cap, key, SlotInitWith<P>{val}, kMaxGlobalProbes, g.counter(), &fresh);
      if (!gs) g.set_full();
      else if (!fresh) slot_apply<P, __HIP_MEMORY_SCOPE_AGENT>(&gs->acc[0], val);
    }
  }
  flush_inserted(g.err, g.inserted);
  // kernel-level accumulators: wave reduce, one global atomic per wave
  if (P::NKW > 0) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
      for (int k = 0; k < P::NKW; k++) {
        u64 o = shfl_xor_u64(kacc[k], m);
        kacc[k] = P::kop(k) == G_UMAX64 ? (o > kacc[k] ? o : kacc[k]) : (kacc[k] | o);
      }
    }
    if (lane_id() == 0) {
      unsigned long long* aux = (unsigned long long*)prm.out[2] + 2;
#pragma unroll
      for (int k = 0; k < P::NKW; k++) {
        if (P::kop(k) == G_UMAX64) atomicMax(aux + k, (unsigned long long)kacc[k]);
        else atomicOr(aux + k, (unsigned long long)kacc[k]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Kernel template C'' (round 6) — a MERGING aggregate (Final / PartialMerge: about one input row per group) without the global table.
// Template C gives every row a find-or-insert in a table in HBM: device-scope compare-and-swap, key and accumulator stores, the publishing store — all
// executed at the memory side, ≈ 0.55 ns per row whatever the kernel does (SF100 Q3's Final aggregate: 1.13 M rows = 1.13 M groups, 0.62 ms, plus a
// 0.15 ms emit pass over a table of 4 M slots and the memset that clears it).  The bucket table's recipe (template D'') fits: the rows are PARTITIONED on
// their key hash (counts per block → column scan → scatter of { key, folded contribution } records; the same per-row code as template C runs, its
// group_update sees part_mode and counts / writes instead of probing), one workgroup per partition merges its records in an LDS table (the find-or-insert
// of template C over workgroup memory) and EMITS its groups from there — P::emit_group, positions from one atomic per partition; the group counter of the
// error block ends as the row count.  No table in HBM, no emit pass over empty slots.
//   out[1] = records, out[3] = { u32 pstart[…kJoinPartMax + 16], u32 tot[kJoinPartMax], u32 cnt[G][NP] } (the join's layout and scan kernel), iarg[2] = NP,
//   iarg[5] = rows per block (a multiple of the tile); the output columns as for k_gemit.  A partition with more groups than its table holds raises
//   the "table full" flag (32): the executor resets the counter and takes template C.
// ---------------------------------------------------------------------------------------------
constexpr int kJoinPartMax = 16384;                 // partitions at most (the 64 KB LDS histogram / cursor array of the partition passes; template D'' shares the layout)
constexpr int kJoinPartTotOff = kJoinPartMax + 16;  // u32 index of tot[] in out[3]
constexpr int kJoinPartCntOff = 2 * kJoinPartMax + 16;
template <class P>
struct AggPart {
  static constexpr int kSlotBytes = 8 + 8 * (P::NK + P::NW);
  static constexpr int kCap = kSlotBytes <= 48 ? 1024 : kSlotBytes <= 96 ? 512 : kSlotBytes <= 192 ? 256 : 128;      // ≤ 48 KB of workgroup memory
};

template <class P>
CDEV void agg_publish_kacc(const CometKParams& prm, u64* kacc) {
  if (P::NKW > 0) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
      for (int k = 0; k < P::NKW; k++) {
        u64 o = shfl_xor_u64(kacc[k], m);
        kacc[k] = P::kop(k) == G_UMAX64 ? (o > kacc[k] ? o : kacc[k]) : (kacc[k] | o);
      }
    }
    if (lane_id() == 0) {
      unsigned long long* aux = (unsigned long long*)prm.out[2] + 2;
#pragma unroll
      for (int k = 0; k < P::NKW; k++) {
        if (P::kop(k) == G_UMAX64) atomicMax(aux + k, (unsigned long long)kacc[k]);
        else atomicOr(aux + k, (unsigned long long)kacc[k]);
      }
    }
  }
}

template <class P, int MODE>
CDEV void agg_part_pass_body(const CometKParams& prm) {
  __shared__ u32 s_part[kJoinPartMax];
  __shared__ u32 s_wsum[kBlock / kWave];
  __shared__ u32 s_carry;
  const i64 n = prm.n, chunk = prm.iarg[5];
  const int np = (int)prm.iarg[2];
  u32* pstart = (u32*)prm.out[3];
  u32* cnt = pstart + kJoinPartCntOff + (i64)blockIdx.x * np;
  if (MODE == 1) {
    for (int p = threadIdx.x; p < np; p += kBlock) s_part[p] = 0;
  } else {
    // cursors: exclusive scan of tot[] (256 partitions at a time, a carry between them) + where this block's records start inside each partition
    const u32* tot = pstart + kJoinPartTotOff;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < np; b0 += kBlock) {
      const int p = b0 + (int)threadIdx.x;
      const u32 v = p < np ? tot[p] : 0u;
      u32 x = v;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const u32 y = __shfl_up(x, d, kWave);
        if (lane_id() >= d) x += y;
      }
      if (lane_id() == kWave - 1) s_wsum[wave_id()] = x;
      __syncthreads();
      u32 excl = s_carry + x - v;
      for (int w = 0; w < wave_id(); w++) excl += s_wsum[w];
      if (p < np) {
        s_part[p] = excl + cnt[p];
        if (blockIdx.x == 0) pstart[p] = excl;
      }
      __syncthreads();
      if (threadIdx.x == kBlock - 1) s_carry = excl + v;      // (thread 255 sits at or behind the round's last partition: the running total)
      __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) pstart[np] = s_carry;
  }
  __syncthreads();
  GroupCtx<P> g;
  g.lds = nullptr;
  g.lds_count = nullptr;
  g.ord_slot = nullptr;
  g.priv = nullptr;
  g.glb = nullptr;
  g.glb_cap = 0;
  g.err = (unsigned int*)prm.out[2];
  g.part_mode = MODE;
  g.part = (COMET_LDS u32*)s_part;
  g.recs = (u64*)prm.out[1];
  g.np = (u32)np;
  u64 kacc[P::NKW > 0 ? P::NKW : 1];
  P::kinit(kacc);
  const i64 tile = (i64)P::R * kBlock;
  const i64 r0 = (i64)blockIdx.x * chunk, r1 = r0 + chunk < n ? r0 + chunk : n;
  for (i64 base = r0; base < r1; base += tile) {
    if constexpr (P::PIPELINED) {
      typename P::L ld;
      P::tile_load(prm, base, r1, ld);
      P::tile_grouped(prm, base, r1, ld, g, kacc);
    } else {
      P::tile_grouped(prm, base, r1, g, kacc);
    }
  }
  __syncthreads();
  if (MODE == 1) {
    for (int p = threadIdx.x; p < np; p += kBlock) cnt[p] = s_part[p];
  } else {
    agg_publish_kacc<P>(prm, kacc);      // (the value bounds of the overflow proof: once, in the pass that keeps the rows)
  }
}

template <class P>
CDEV void agg_part_merge_body(const CometKParams& prm) {
  typedef Slot<P::NK, P::NW> S;
  constexpr int kCap = AggPart<P>::kCap;
  constexpr int kPer = kCap >= kBlock ? kCap / kBlock : 1;      // slots per thread in the emit sweep (a table of wide slots is smaller than the block: its tail threads have none)
  __shared__ S s_tbl[kCap];
  __shared__ u32 s_wave[kBlock / kWave];
  __shared__ unsigned long long s_base;
  const int np = (int)prm.iarg[2];
  const u32* pstart = (const u32*)prm.out[3];
  const u64* recs = (const u64*)prm.out[1];
  unsigned int* err = (unsigned int*)prm.out[2];
  const int lane = lane_id(), wv = wave_id();
  for (int p = blockIdx.x; p < np; p += gridDim.x) {
    for (int i = threadIdx.x; i < kCap * (int)(sizeof(S) / 8); i += kBlock) ((u64*)s_tbl)[i] = 0ull;
    __syncthreads();
    const u32 r0 = pstart[p], r1 = pstart[p + 1];
    u32 dummy = 0;
    for (u32 r = r0 + threadIdx.x; r < r1; r += kBlock) {
      const u64* rec = recs + (u64)r * (u64)(P::NK + P::NW);
      u64 key[P::NK], val[P::NW];
#pragma unroll
      for (int k = 0; k < P::NK; k++) key[k] = rec[k];
#pragma unroll
      for (int k = 0; k < P::NW; k++) val[k] = rec[P::NK + k];
      bool fresh = false;
      S* gs = table_find_or_insert<P::NK, P::NW>((S*)s_tbl, (u64)kCap, key, SlotInitWith<P>{val}, (u64)kCap, &dummy, &fresh);
      if (!gs) atomicOr(err, 32u);      // more groups than the partition's table holds: the executor takes template C
      else if (!fresh) slot_apply<P, __HIP_MEMORY_SCOPE_WORKGROUP>(&gs->acc[0], val);
    }
    __syncthreads();
    // emit: this partition's groups in slot order, ONE atomic for their rows
    u32 before[kPer], ready = 0, run = 0;
#pragma unroll
    for (int r = 0; r < kPer; r++) {
      const int si = r * kBlock + (int)threadIdx.x;
      const bool rd = si < kCap && s_tbl[si < kCap ? si : 0].state == kSlotReady;
      const u64 b = __ballot(rd);
      before[r] = run + (u32)__popcll(b & ((1ull << lane) - 1ull));
      run += (u32)__popcll(b);
      if (rd) ready |= 1u << r;
    }
    if (lane == 0) s_wave[wv] = run;
    __syncthreads();
    u32 woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; w++) {
      if (w < wv) woff += s_wave[w];
      total += s_wave[w];
    }
    if (threadIdx.x == 0 && total) s_base = atomicAdd((unsigned long long*)err + 1, (unsigned long long)total);
    __syncthreads();
    if (ready) {
#pragma unroll
      for (int r = 0; r < kPer; r++)
        if ((ready >> r) & 1u) {
          const S* sl = &s_tbl[r * kBlock + (int)threadIdx.x];
          u64 key[P::NK], acc[P::NW];
#pragma unroll
          for (int k = 0; k < P::NK; k++) key[k] = sl->key[k];
#pragma unroll
          for (int k = 0; k < P::NW; k++) acc[k] = sl->acc[k];
          P::emit_group(prm, key, acc, (i64)s_base + woff + before[r]);
        }
    }
    __syncthreads();
  }
}

// Emit one output row per occupied slot (order unspecified, like the reference's hash aggregate).
//   prm.out[0] table, iarg[0] capacity, prm.out[1] = u64 row counter
template <class P>
CDEV void agg_grouped_emit_body(const CometKParams& prm) {
  typedef Slot<P::NK, P::NW> S;
  const S* tbl = (const S*)prm.out[0];
  const i64 cap = prm.iarg[0];
  // ONE atomic per 2048 slots reserves the output rows of their ready groups: atomics on one address serialise at the L2 (~13 ns each —
  // one per wave was 65 K of them, 0.83 ms for the 1.13 M groups of SF100 Q3 whatever else the kernel did); the groups' order is
  // unspecified either way
  const int lane = lane_id(), wv = wave_id();
  constexpr int R = 8;
  __shared__ u32 s_wave[kBlock / kWave];
  __shared__ unsigned long long s_base;
  for (i64 base = (i64)blockIdx.x * kBlock * R; base < cap; base += (i64)gridDim.x * kBlock * R) {
    u32 ready = 0, before[R];
    u32 run = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      const i64 i = base + (i64)r * kBlock + threadIdx.x;
      const bool rd = i < cap && tbl[i].state == kSlotReady;
      const u64 b = __ballot(rd);
      before[r] = run + (u32)__popcll(b & ((1ull << lane) - 1ull));
      run += (u32)__popcll(b);
      ready |= rd ? 1u << r : 0u;
    }
    if (lane == 0) s_wave[wv] = run;
    __syncthreads();
    u32 woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; w++) {
      if (w < wv) woff += s_wave[w];
      total += s_wave[w];
    }
    if (threadIdx.x == 0 && total) s_base = atomicAdd((unsigned long long*)prm.out[1], (unsigned long long)total);
    __syncthreads();
    if (ready) {
      const i64 at = (i64)s_base + woff;
#pragma unroll
      for (int r = 0; r < R; r++)
        if ((ready >> r) & 1u) {
          const i64 i = base + (i64)r * kBlock + threadIdx.x;
          P::emit_group(prm, tbl[i].key, tbl[i].acc, at + before[r]);
        }
    }
    __syncthreads();   // s_wave / s_base are reused
  }
}

template <class P>
CDEV void agg_grouped_rehash_body(const CometKParams& prm) {
  typedef Slot<P::NK, P::NW> S;
  const S* old = (const S*)prm.out[3];
  S* nt = (S*)prm.out[0];
  u32 inserted = 0;
  for (i64 i = (i64)blockIdx.x * kBlock + threadIdx.x; i < prm.iarg[1]; i += (i64)gridDim.x * kBlock) {
    if (old[i].state == kSlotReady) {
      u64 key[P::NK];
#pragma unroll
      for (int k = 0; k < P::NK; k++) key[k] = old[i].key[k];
      S* gs = table_find_or_insert<P::NK, P::NW>(nt, (u64)prm.iarg[0], key, SlotInit<P>(), (u64)prm.iarg[0], &inserted);
      if (gs) {
#pragma unroll
        for (int k = 0; k < P::NW; k++) gs->acc[k] = old[i].acc[k];  // unique key per old slot: plain copy
      } else {
        atomicOr((unsigned int*)prm.out[2], 32u);
      }
    }
  }
  flush_inserted((unsigned int*)prm.out[2], inserted);
}

// ---------------------------------------------------------------------------------------------
// Kernel template D — hash join (reference: planner.rs:2192-2266 → DataFusion HashJoinExec; NULL keys never
// match, planner.rs:2225-2227).  Build side: bucket-chained table over the build rows,
//   head[cap] (u32, 0xffffffff = empty: bits [0, B) = newest build row of the bucket, B = iarg[2] = bits needed for a build row index;
//   bits [B, 31) = tag bits of that row's hash; bit 31 = "the chain holds more than one row") and next[n_build]; insertion is one
//   atomicExch per row, duplicates chain.  The tag and the flag let the probe settle most rows with ONE random access: an empty bucket,
//   or a single-row bucket whose tag differs, is a miss without touching the build keys or next[] (a probe-side FK join misses or hits
//   single-row buckets almost always).  The table stays 4 bytes per bucket — an 8-byte head with a 31-bit tag was measured too: the
//   probe of SF100 Q3's second join got SLOWER (the 256 MB head array no longer sits in the 256 MB Infinity Cache).
// Probe side: template D' below (single pass; counts and emits together).
//   P::pkeep(prm,j)                       the probe row exists at all: the Filters of a probe-side chain fused into the kernel
//   P::bvalid(prm,i) / P::pvalid(prm,j)   all key columns non-NULL
//   P::bhash(prm,i)  / P::phash(prm,j)    64-bit hash of the key words
//   P::match(prm,i,j)                     keys equal (and residual join condition TRUE)
//   P::emit(prm,i,j,pos)                  write output row pos from build row i / probe row j (i = -1: no build row)
//   P::MODE                               0 inner, 1 semi (probe row kept if it matches), 2 anti (kept if it does not)
// prm.in[0 .. NB) = build columns, prm.in[NB ..) = probe columns; prm.iarg[0] = table capacity (power of two),
// iarg[1] = build rows, prm.n = probe rows; out[0] = head, out[1] = next, out[2] = err, out[3] = per-row counts (u32),
// out[kJoinTileCounts] = tile counts/offsets.
// Outer joins: P::OUTER_PROBE keeps probe rows without a match (build columns NULL, P::emit_probe_only); P::OUTER_BUILD keeps
// build rows no probe row matched — the count pass marks matched build rows in out[kJoinMatched] (one byte per build row),
// join_build_unmatched_count / _emit then append them after the matched output (probe columns NULL, P::emit_build_only);
// out[kJoinBuildTiles] = their tile counts/offsets, iarg[3] = number of build tiles, iarg[4] = first output row of that tail.
// ---------------------------------------------------------------------------------------------
constexpr int kJoinTileCounts = 44;
constexpr int kJoinMatched = 45;
constexpr int kJoinBuildTiles = 46;
// The build side's KEY BITMAP (round 4).  A join on one integer key whose build keys span a range not much larger than their number
// (foreign keys: customer keys, order keys) gets one bit per possible key: out[kJoinKeyMap] = { i64 smallest key, u64 bits, u32 words[] },
// or NULL.  The probe asks it first — a probe row whose key the build side does not hold never touches the hash table, and for the usual
// shapes the bit is a cache hit: a dimension's key range is a few MB (L2), a fact table probing in key order walks the bitmap
// sequentially.  SF100 Q3: 80 % of the orders probing the customers of one market segment, and 95 % of the line items probing the open
// orders, end here instead of at a random 128-byte line of the bucket array (profiles/r4_q3_probe_pmc.txt).
constexpr int kJoinKeyMap = 44;
CDEV bool join_keymap_has(const u64* km, u64 key) {
  const u64 idx = key - km[0];
  return idx < km[1] && ((((const u32*)(km + 2))[idx >> 5] >> (idx & 31u)) & 1u) != 0;
}

constexpr u32 kJoinNoRow = 0xffffffffu;
constexpr u32 kJoinChainBit = 1u << 31;
constexpr i32 kJoinFollower = -2;     // next[] of a build row that continues the run of equal keys its predecessor started
// the bucket index uses the hash's low bits, the tag its high bits; B index bits leave 31 − B tag bits (B ≤ 31; a row index is never all
// ones in B bits, so no entry equals the empty marker)
CDEV u32 join_head_entry(u64 h, u32 row, int ib) { return ((u32)(h >> 33) << ib) & 0x7fffffffu | row; }

// Runs of equal keys.  Fact tables arrive clustered by their join key (the lines of an order, the items of a ticket): the rows of one key
// are NEIGHBOURS, and a wave holds 64 consecutive rows.  A row whose key equals its left neighbour's (inside the wave) is a FOLLOWER: it
// does not touch the table at all — next[row] = kJoinFollower marks it — and only the run's first row, its LEADER, is inserted (one
// atomicExch per run instead of one per row, and no two lanes of a wave ever hit the same bucket with the same key).  The probe, having
// reached a leader, walks the followers that sit right behind it (contiguous rows: the same cache lines).  Keys whose rows are scattered
// simply form runs of length one.  P::DEDUP_BUILD (semi / anti joins without a residual condition: only the key's existence matters):
// followers are dropped altogether.
//   P::NKW, P::bkeys(prm, i, kw)   the key words of build row i (what bhash hashes)
template <class P>
CDEV bool join_build_classify(const CometKParams& prm, i64 i, i64 nb, u64& h, bool& has_follower) {   // → leader?; writes next[] of non-leaders
  i32* next = (i32*)prm.out[1];
  const int lane = lane_id();
  const bool valid = i < nb && P::bvalid(prm, i);
  u64 kw[P::NKW];
#pragma unroll
  for (int w = 0; w < P::NKW; w++) kw[w] = 0;
  if (valid) P::bkeys(prm, i, kw);
  bool same = valid && lane > 0;
#pragma unroll
  for (int w = 0; w < P::NKW; w++) {
    const u32 lo = __shfl_up((u32)kw[w], 1, kWave), hi = __shfl_up((u32)(kw[w] >> 32), 1, kWave);
    same = same && (((u64)hi << 32) | lo) == kw[w];
  }
  const int prev_valid = __shfl_up(valid ? 1 : 0, 1, kWave);      // (its own statement: as the right operand of && only the lanes whose `same` still held would
  same = same && prev_valid != 0;                                 //  execute the shuffle, and a shuffle FROM a lane that sits the instruction out returns anything)
  const u64 followers = __ballot(same);
  has_follower = lane < kWave - 1 && ((followers >> (lane + 1)) & 1ull) != 0 && !P::DEDUP_BUILD;
  h = hash_key<P::NKW>(kw);
  if (i < nb && !P::DEDUP_BUILD) {
    if (same) next[i] = kJoinFollower;
    else if (!valid) next[i] = -1;          // a NULL key: never in the table, and it ends the run before it
  }
  return valid && !same;
}

template <class P>
CDEV void join_build_body(const CometKParams& prm) {
  u32* head = (u32*)prm.out[0];
  i32* next = (i32*)prm.out[1];
  const u64 mask = (u64)prm.iarg[0] - 1;
  const i64 nb = prm.iarg[1];
  const int ib = (int)prm.iarg[2];
  // wave-uniform loop: lane l of a wave holds row wbase + l
  for (i64 wbase = (i64)blockIdx.x * kBlock + (i64)wave_id() * kWave; wbase < nb; wbase += (i64)gridDim.x * kBlock) {
    const i64 i = wbase + lane_id();
    u64 h;
    bool has_follower;
    if (!join_build_classify<P>(prm, i, nb, h, has_follower)) continue;
    // the chain bit says "this bucket holds more than one ROW": another leader, or followers behind this one
    const u32 old = atomicExch(&head[h & mask], join_head_entry(h, (u32)i, ib) | (has_follower ? kJoinChainBit : 0u));
    next[i] = old == kJoinNoRow ? -1 : (i32)(old & ((1u << ib) - 1u));
    // a bucket that already held a row: whatever its head is from now on, its chain is longer than one (monotone, so the OR may land
    // on a newer head)
    if (old != kJoinNoRow) atomicOr(&head[h & mask], kJoinChainBit);
  }
}

// how many leaders the build will insert: sizes the bucket array by KEYS-ish instead of rows (a clustered fact table has several rows
// per key; its bucket array then fits the caches four times better).  out[0] = { u64 leaders }
template <class P>
CDEV void join_build_count_body(const CometKParams& prm) {
  const i64 nb = prm.iarg[1];
  unsigned long long* total = (unsigned long long*)prm.out[0];      // { leaders, smallest key, largest key, rows with a key } — the keys order-preserving as u64 (sign bit flipped)
  u32 mine = 0, keyed = 0;
  u64 kmin = ~0ull, kmax = 0;
  for (i64 wbase = (i64)blockIdx.x * kBlock + (i64)wave_id() * kWave; wbase < nb; wbase += (i64)gridDim.x * kBlock) {
    const i64 i = wbase + lane_id();
    const bool valid = i < nb && P::bvalid(prm, i);
    u64 kw[P::NKW];
#pragma unroll
    for (int w = 0; w < P::NKW; w++) kw[w] = 0;
    if (valid) P::bkeys(prm, i, kw);
    if (P::KEYMAP && valid) {
      const u64 o = kw[0] ^ (1ull << 63);
      kmin = o < kmin ? o : kmin;
      kmax = o > kmax ? o : kmax;
    }
    bool same = valid && lane_id() > 0;
#pragma unroll
    for (int w = 0; w < P::NKW; w++) {
      const u32 lo = __shfl_up((u32)kw[w], 1, kWave), hi = __shfl_up((u32)(kw[w] >> 32), 1, kWave);
      same = same && (((u64)hi << 32) | lo) == kw[w];
    }
    const int prev_valid = __shfl_up(valid ? 1 : 0, 1, kWave);      // (its own statement: as the right operand of && only the lanes whose `same` still held would
  same = same && prev_valid != 0;                                 //  execute the shuffle, and a shuffle FROM a lane that sits the instruction out returns anything)
    mine += (u32)__popcll(__ballot(valid && !same));
    keyed += (u32)__popcll(__ballot(valid));
  }
  // one set of atomics per BLOCK: thousands of waves adding to, and taking the minimum / maximum of, the same three words queue up
  // behind one another (measured: 0.11 → 0.30 ms per launch when every wave did its own)
  __shared__ u32 s_cnt[kBlock / kWave], s_keyed[kBlock / kWave];
  __shared__ u64 s_min[kBlock / kWave], s_max[kBlock / kWave];
  if (P::KEYMAP) {
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
      const u64 a = ((u64)__shfl_xor((u32)(kmin >> 32), d, kWave) << 32) | __shfl_xor((u32)kmin, d, kWave);
      const u64 b = ((u64)__shfl_xor((u32)(kmax >> 32), d, kWave) << 32) | __shfl_xor((u32)kmax, d, kWave);
      kmin = a < kmin ? a : kmin;
      kmax = b > kmax ? b : kmax;
    }
  }
  if (lane_id() == 0) { s_cnt[wave_id()] = mine; s_keyed[wave_id()] = keyed; s_min[wave_id()] = kmin; s_max[wave_id()] = kmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 c = 0, kd = 0;
    u64 lo = ~0ull, hi = 0;
    for (int w = 0; w < kBlock / kWave; w++) {
      c += s_cnt[w];
      kd += s_keyed[w];
      lo = s_min[w] < lo ? s_min[w] : lo;
      hi = s_max[w] > hi ? s_max[w] : hi;
    }
    if (c) atomicAdd(total, (unsigned long long)c);
    if (kd) atomicAdd(total + 3, (unsigned long long)kd);
    if (P::KEYMAP && lo <= hi) {
      atomicMin(total + 1, (unsigned long long)lo);
      atomicMax(total + 2, (unsigned long long)hi);
    }
  }
}

// build rows nobody matched (outer joins that preserve the build side; LeftAnti built on the left) — or, with
// P::BUILD_KEEP_MATCHED, the build rows that WERE matched (LeftSemi built on the left): count per tile, then emit
template <class P>
CDEV void join_build_unmatched_count_body(const CometKParams& prm) {
  const i64 nb = prm.iarg[1];
  const u8* matched = (const u8*)prm.out[kJoinMatched];
  u64* tile_counts = (u64*)prm.out[kJoinBuildTiles];
  const i64 ntiles = (nb + kMaskTileRows - 1) / kMaskTileRows;
  __shared__ u32 s_cnt;
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    u32 local = 0;
#pragma unroll
    for (int r = 0; r < kMaskTileRows / kBlock; r++) {
      i64 i = t * kMaskTileRows + r * kBlock + threadIdx.x;
      local += (u32)__popcll(__ballot(i < nb && (matched[i] != 0) == P::BUILD_KEEP_MATCHED && P::bkeep(prm, i))) * (lane_id() == 0 ? 1u : 0u);      // (bkeep: a fused build chain's Filters — a source row they drop is no build row)
    }
    if (lane_id() == 0) atomicAdd(&s_cnt, local);
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[t] = s_cnt;
    __syncthreads();
  }
}
template <class P>
CDEV void join_build_unmatched_emit_body(const CometKParams& prm) {
  const i64 nb = prm.iarg[1];
  const u8* matched = (const u8*)prm.out[kJoinMatched];
  const u64* tile_off = (const u64*)prm.out[kJoinBuildTiles];
  const i64 ntiles = (nb + kMaskTileRows - 1) / kMaskTileRows;
  __shared__ u32 s_wave[kBlock / kWave];
  __shared__ u32 s_run;
  for (i64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int r = 0; r < kMaskTileRows / kBlock; r++) {
      i64 i = t * kMaskTileRows + r * kBlock + threadIdx.x;
      const bool keep = i < nb && (matched[i] != 0) == P::BUILD_KEEP_MATCHED && P::bkeep(prm, i);
      const u64 b = __ballot(keep);
      const u32 below = (u32)__popcll(b & ((1ull << lane_id()) - 1));
      if (lane_id() == 0) s_wave[wave_id()] = (u32)__popcll(b);
      __syncthreads();
      u32 woff = 0;
      for (int w = 0; w < wave_id(); w++) woff += s_wave[w];
      const u32 run = s_run;
      if (keep) P::emit_build_only(prm, i, prm.iarg[4] + (i64)tile_off[t] + run + woff + below);
      __syncthreads();
      if (threadIdx.x == 0) s_run = run + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Kernel template D' — single-pass probe.  Template D walks every probe row's chain twice (count, then emit) with a scan launch in
// between, because FilterExec-style ordered output needs exact positions.  A join's output order is unspecified in the reference
// (HashJoinExec emits per probe batch in hash-table order), so one pass suffices: every thread probes R rows and remembers per row
// the number of matches and the first matching build row; the tile's total is one block reduction + ONE global atomic that reserves
// its output range; rows with a single match (the common, FK-shaped case) are emitted from the remembered row, only multi-match rows
// walk their chain again.  The output buffers have a capacity: beyond it rows are counted, not written, and the executor re-runs the
// kernel with the exact size (only many-to-many joins ever need that).
// Two sources of candidates share the tile logic:
//   * join_probe_fused_body: the chained global table of template D (build side of any size);
//   * join_probe_lds_body: a build side of ≤ kJoinLdsMaxBuild rows (dimension tables, broadcast joins) is hashed ONCE per block into
//     an open-addressing table in LDS (u32 row + u16 hash tag per slot, linear probing) and probed there — no HBM access per probe row
//     except the key / condition check of a slot whose tag already matched ("hash-join probe staged through LDS open-addressing
//     tables", north_star).  Radix-partitioning BOTH sides so that larger build sides fit LDS was built and measured in round 2
//     (DESIGN §4c): the 4096-way scatter of 324 M probe pairs ran at 1.4 TB/s — one 8-byte write transaction per row — which made the
//     partitioned join slower than this single-pass probe of the global table, so it was removed.
//   out[47] = { u64 emitted },  iarg[6] = output capacity
// ---------------------------------------------------------------------------------------------
constexpr int kJoinLdsCap = 8192;
constexpr int kJoinLdsMaxBuild = 6144;
constexpr u32 kJoinEmpty = 0xffffffffu;
#ifndef COMET_JOIN_R
#define COMET_JOIN_R 4
#endif
#ifndef COMET_JOIN_R0
#define COMET_JOIN_R0 16
#endif
#ifndef COMET_JOIN_UNCOND
#define COMET_JOIN_UNCOND 1
#endif
#ifndef COMET_JOIN_PRE_RUN
#define COMET_JOIN_PRE_RUN 2
#endif
#ifndef COMET_JOIN_FILTER_UNCOND
#define COMET_JOIN_FILTER_UNCOND 1
#endif
#ifndef COMET_JOIN_KM_UNCOND
#define COMET_JOIN_KM_UNCOND 1
#endif
#ifndef COMET_JOIN_DIRECT_UNCOND
#define COMET_JOIN_DIRECT_UNCOND 0      // (measured on SF100 Q3: the direct map's few survivors per tile gain nothing from the unconditional sweeps' extra instructions)
#endif
constexpr int kJoinR = COMET_JOIN_R;                    // slices of a wave's survivors probed together (eight random accesses in flight per lane)
// (kJoinR0 = COMET_JOIN_R0, a template parameter of join_probe_tiles since round 6: the direct map's probe — nearly all of its work is the filter / bitmap phase —
// takes 24 rows per thread (SF100 Q3: 4.09 → 3.90 ms), the bucket table's 16 (24 cost TPC-DS Q95 2 ms: registers))
#ifndef COMET_JOIN_R0_DIRECT
#define COMET_JOIN_R0_DIRECT 32      // (24 under ROCm 7.0's compiler; under ROCm 7.2's, SF100 Q3's k_jdprobe: 16 → 4.00, 24 → 4.15–4.23, 32 → 3.98–4.01 ms)
#endif
// Waves per SIMD the register allocator must leave room for (the second argument of __launch_bounds__ in the generated kernels' declarations; 1 = no
// request).  The two compilers this header has met differ by a few registers on the same source, and a few registers decide an occupancy step (512 VGPRs
// per SIMD in units of 8): ROCm 7.2's clang 22 gives k_jdprobe 174 where ROCm 7.0's clang 20 gave 139 (two waves instead of three), k_jprobe_b 97 for 91
// (four for five), k_gagg 86 for 80 (five for six) — and none of the requests pays (profiles/r6_jit_compiler.md: k_gagg held to six waves 9.44 ms for 7.65,
// k_jdprobe held to three 5.22 ms for 4.23, k_jprobe_b held to five or six: no change), so the defaults ask for nothing.
#ifndef COMET_WAVES_JDPROBE
#define COMET_WAVES_JDPROBE 1
#endif
#ifndef COMET_WAVES_JPROBE_B
#define COMET_WAVES_JPROBE_B 1
#endif
#ifndef COMET_WAVES_JPROBE_BKM
#define COMET_WAVES_JPROBE_BKM 1
#endif
#ifndef COMET_WAVES_JLDS
#define COMET_WAVES_JLDS 1
#endif
#ifndef COMET_WAVES_GAGG
#define COMET_WAVES_GAGG 1
#endif
// probe rows per thread and tile: the filter / key bitmap phase runs over twice as many rows as one probe batch holds —
                                             // most rows end there, and a wave's handful of survivors costs the same latency whatever the tile's size

// candidates of probe row j in the global chained table
template <class P>
struct JoinGlobalTable {
  static constexpr bool BY_KEY = false;      // peek() takes the key's hash
  typedef u32 Entry;                         // what peek() hands out: the bucket's head
  static CDEV Entry none() { return kJoinNoRow; }
  static constexpr bool UNCONDITIONAL = false;      // peek() / prefetch() only for lanes that hold a probe row with a key
  CDEV u64 key_of(const CometKParams& prm, i64 j) const { return P::phash(prm, j); }      // what peek() takes
  const u32* head;
  const i32* next;
  u64 mask;
  int ib;
  i64 nb;
  // the one access every probe row needs; the tile loads it for all its rows before it looks at any of them
  CDEV u32 peek(u64 h) const { return head[h & mask]; }
  // … and then, for all its rows again, what the first candidate costs: the key (and condition) check, the chain successor and the marker of
  // the row behind it — loads that do not depend on one another, issued together
  struct Pre {
    bool m;     // the first candidate matches (keys and residual condition)
    i32 nx;     // its successor in the bucket's chain, < 0 = none
    i32 nf;     // next[] of the row right behind it: kJoinFollower = a run continues there
  };
  CDEV Pre prefetch(const CometKParams& prm, i64 j, u64 h, u32 e) const {
    Pre p{false, -1, 0};
    if (e == kJoinNoRow) return p;
    const u32 rowmask = (1u << ib) - 1u, first = e & rowmask;
    const bool chained = (e & kJoinChainBit) != 0;
    if (!chained && ((e ^ join_head_entry(h, 0, ib)) & ~rowmask) != 0) return p;      // one row in the bucket, another tag: a miss, nothing to read
    p.m = P::match(prm, (i64)first, j);
    if (chained) {
      p.nx = next[first];
      if (!P::DEDUP_BUILD && (i64)first + 1 < nb) p.nf = next[first + 1];
    }
    return p;
  }
  template <class F>
  CDEV void for_each(const CometKParams& prm, i64 j, u64 h, u32 e, const Pre& pre, F f) const {   // f(build row) returns false to stop
    (void)h;
    if (e == kJoinNoRow) return;
    const u32 rowmask = (1u << ib) - 1u;
    i32 i = (i32)(e & rowmask);
    if (!(e & kJoinChainBit)) {
      if (pre.m) f((u32)i);
      return;
    }
    bool mi = pre.m;
    i32 nx = pre.nx, nf = pre.nf;
    for (;;) {
      if (mi && !f((u32)i)) return;
      if (!P::DEDUP_BUILD) {                            // the leader's followers: the rows right behind it (same key, maybe another condition)
        i64 k = (i64)i + 1;
        i32 fk = nf;
        while (k < nb && fk == kJoinFollower) {
          const i32 fnext = k + 1 < nb ? next[k + 1] : 0;      // the next marker travels with this follower's key / condition loads: one latency per follower
          if (P::match(prm, k, j) && !f((u32)k)) return;
          k++;
          fk = fnext;
        }
      }
      if (nx < 0) return;
      i = nx;
      mi = P::match(prm, (i64)i, j);
      nx = next[i];
      nf = (!P::DEDUP_BUILD && (i64)i + 1 < nb) ? next[i + 1] : 0;
    }
  }
};
// candidates in the block's LDS table
template <class P>
struct JoinLdsTable {
  static constexpr bool BY_KEY = false;
  static constexpr bool UNCONDITIONAL = false;
  typedef u32 Entry;
  static CDEV Entry none() { return kJoinEmpty; }
  CDEV u64 key_of(const CometKParams& prm, i64 j) const { return P::phash(prm, j); }
  const COMET_LDS u32* rows;             // address_space(3): ds_read, not FLAT (a FLAT access waits for every outstanding global load)
  const COMET_LDS unsigned short* tags;
  CDEV u32 peek(u64 h) const { return rows[((u32)(h >> 32)) & (kJoinLdsCap - 1)]; }
  struct Pre {};
  CDEV Pre prefetch(const CometKParams&, i64, u64, u32) const { return Pre{}; }
  template <class F>
  CDEV void for_each(const CometKParams& prm, i64 j, u64 h, u32 e, const Pre&, F f) const {
    if (e == kJoinEmpty) return;
    const u32 hi = (u32)(h >> 16);
    u32 slot = (hi >> 16) & (kJoinLdsCap - 1);
    for (u32 row; (row = rows[slot]) != kJoinEmpty; slot = (slot + 1) & (kJoinLdsCap - 1))
      if (tags[slot] == (unsigned short)hi && P::match(prm, (i64)row, j) && !f(row)) break;
  }
};

// The DIRECT MAP (round 4): a build side whose one integer key is UNIQUE and spans a foreign key's range (a primary key: the customers of a
// segment, the open orders) needs no hash table at all.  Its key bitmap (above) is completed by the number of keys below every 128-bit block
// (ranks[]) and by the build rows in KEY order (rows[]): a probe row whose bit is set finds its build row at rows[ranks[block] + bits below
// it in the block] — no hash, no tag, no chain, and no look at the build side's key column (the position IS the key; a residual condition
// is still evaluated).  A fact table probing in key order (the line items of the open orders) walks all three arrays sequentially; a
// dimension's arrays are a few MB.  out[0] = ranks (u32 per block), out[1] = rows (u32 per build row), out[kJoinKeyMap] = the bitmap.
template <class P>
struct JoinDirectTable {
  static constexpr bool BY_KEY = true;       // peek() takes the key itself
  static constexpr bool UNCONDITIONAL = COMET_JOIN_DIRECT_UNCOND != 0;      // (peek clamps keys outside the bitmap, prefetch clamps the row it reads back)
  typedef u32 Entry;
  static CDEV Entry none() { return kJoinNoRow; }
  CDEV u64 key_of(const CometKParams& prm, i64 j) const { return P::pkey0(prm, j); }
  const u64* km;
  const u32* ranks;
  const u32* rows;
  CDEV u32 peek(u64 key) const {             // position of the key among the build side's keys (its bit is set: the tile's filter phase looked)
    u64 idx = key - km[0];
    if (UNCONDITIONAL) idx = idx < km[1] ? idx : 0ull;      // (a lane without a probe row asks with whatever key it has)
    const u64 blk = idx >> 7;
    const uint4 w = ((const uint4*)(km + 2))[blk];
    const u32 b = (u32)idx & 127u;
    u32 c = 0;
    c += b >= 32 ? (u32)__popc(w.x) : (u32)__popc(w.x & ((1u << (b & 31u)) - 1u));
    if (b >= 32) c += b >= 64 ? (u32)__popc(w.y) : (u32)__popc(w.y & ((1u << (b & 31u)) - 1u));
    if (b >= 64) c += b >= 96 ? (u32)__popc(w.z) : (u32)__popc(w.z & ((1u << (b & 31u)) - 1u));
    if (b >= 96) c += (u32)__popc(w.w & ((1u << (b & 31u)) - 1u));
    return ranks[blk] + c;
  }
  struct Pre { u32 row; bool m; };
  CDEV Pre prefetch(const CometKParams& prm, i64 j, u64, u32 e) const {
    Pre p{rows[e], true};
    if (UNCONDITIONAL && p.row >= (u32)prm.iarg[1]) p.row = 0;      // (a position behind the last key: rows[] holds nothing there; such a lane's answer is dropped)
    if (P::HAS_COND) p.m = P::match(prm, (i64)p.row, j);
    return p;
  }
  template <class F>
  CDEV void for_each(const CometKParams&, i64, u64, u32, const Pre& pre, F f) const {
    if (pre.m) f(pre.row);
  }
};

// One tile = kJoinR × 256 probe rows; every WAVE owns the 8 × 64 of them that its lanes load (coalesced), and works on its own:
//   0. filter: P::pkeep (the Filters of a fused probe chain) and P::pvalid per row; the rows that take part are COMPACTED into the wave's
//      list in LDS (ballot + popcount, no barrier: the list is the wave's own) — after a 50 % filter the probe phase runs with full
//      waves instead of half-empty ones;
//   1. probe, 64 list entries at a time, in three sweeps over the wave's (up to 8) slices: all key loads and hashes, then ALL bucket-head
//      loads — up to eight random accesses in flight per lane, where one-row-at-a-time probing has one and spends 90 % of its cycles
//      waiting (SQ_WAIT_ANY, profiles/r3_q95_join_pmc.txt) —, then the rows are settled one by one (most need nothing more: an empty
//      bucket, or a single row with another tag);
//   2. output positions in (wave, slice, lane) order — a wave's consecutive list entries are consecutive input rows, so a clustered
//      probe side gives a (locally) clustered join output, which the next join's build exploits (runs); one atomic per tile reserves
//      the range.
// KM (compile time): a key bitmap may be present (k_jprobe_km, k_jdprobe).  The bitmap's sweeps hold sixteen keys and sixteen bitmap words per
// thread; a kernel that CAN take them is allocated the registers for them whether the bitmap exists at run time or not, and a probe that lives
// on random accesses (TPC-DS Q95's self-joins: 72 M rows, duplicate-heavy keys, no bitmap) lost a fifth of its speed to the lower occupancy
// (round-5 bisect: 22.7 → 26.7 ms, profiles/r5_q95_bisect.txt).  So the table probe without a bitmap is its own kernel with the plain filter loop.
template <class P, class T, bool KM, int kJoinR0 = COMET_JOIN_R0>
CDEV void join_probe_tiles(const CometKParams& prm, const T& table) {
  static_assert(kJoinR0 >= 1 && kJoinR0 <= 32, "a tile's rows per thread are tracked in 32-bit masks (alive_bits, can_bits)");
  const i64 n = prm.n;
  const i64 cap_out = prm.iarg[6];
  unsigned long long* emitted = (unsigned long long*)prm.out[47];
  u8* matched = (u8*)prm.out[kJoinMatched];
  const u64* keymap = (const u64*)prm.out[kJoinKeyMap];
  __shared__ unsigned short s_list[kBlock / kWave][kJoinR0 * kWave];
  __shared__ u32 s_wave[kBlock / kWave];
  __shared__ unsigned long long s_base;
  const int lane = lane_id(), wv = wave_id();
  const u64 lt = (1ull << lane) - 1ull;
  COMET_LDS unsigned short* list = (COMET_LDS unsigned short*)s_list[wv];
  constexpr i64 kTile = (i64)kJoinR0 * kBlock;
  for (i64 base = (i64)blockIdx.x * kTile; base < n; base += (i64)gridDim.x * kTile) {
    // ---- 0. filter + wave-local compaction ----
    // (three sweeps over the thread's sixteen rows — the chain's filters, the keys, the bitmap words — so that each sweep's loads are in flight
    // together: one loop that filtered, looked up and compacted row after row spent its time in sixteen dependent load latencies)
    u32 m = 0;
    u32 alive_bits = 0, can_bits = 0;
    if (!(KM && P::KEYMAP)) {
#pragma unroll
      for (int r = 0; r < kJoinR0; r++) {
        const i64 j = base + (i64)r * kBlock + threadIdx.x;
#if COMET_JOIN_FILTER_UNCOND
        const i64 js = j < n ? j : n - 1;                      // (a row that exists: the loads below carry no lane-dependent branch of their own)
        const bool keep = P::pkeep(prm, js), pv = P::pvalid(prm, js);
        const bool alive = keep & (j < n);
        if (alive) alive_bits |= 1u << r;
        if (alive & pv) can_bits |= 1u << r;
#else
        const bool alive = j < n && P::pkeep(prm, j);
        if (alive) alive_bits |= 1u << r;
        if (alive && P::pvalid(prm, j)) can_bits |= 1u << r;
#endif
      }
    } else {
#if COMET_JOIN_KM_UNCOND
    // (no lane-dependent branch around a load — see JoinBucketTable::UNCONDITIONAL: rows past the end read the table's last row, rows the chain's filter
    // dropped still load their key, keys outside the bitmap read its first word; the answers are masked afterwards)
#pragma unroll
    for (int r = 0; r < kJoinR0; r++) {
      const i64 j = base + (i64)r * kBlock + threadIdx.x;
      const bool keep = P::pkeep(prm, j < n ? j : n - 1);
      if (keep & (j < n)) alive_bits |= 1u << r;
    }
    if (keymap) {
      u64 keys[kJoinR0];
#pragma unroll
      for (int r = 0; r < kJoinR0; r++) {
        const i64 j = base + (i64)r * kBlock + threadIdx.x;
        const i64 js = j < n ? j : n - 1;
        const bool pv = P::pvalid(prm, js);
        keys[r] = P::pkey0(prm, js);
        if (((alive_bits >> r) & 1u) & pv) can_bits |= 1u << r;
      }
      const u64 first = keymap[0], bits = keymap[1];
      u32 words[kJoinR0];
#pragma unroll
      for (int r = 0; r < kJoinR0; r++) {
        const u64 idx = keys[r] - first;
        // (rows the chain's filter dropped read their word too — sending them to word 0 instead was measured and is no faster: SF100 Q3 3.99 against 3.87–3.92 ms)
        const u32 w = ((const u32*)(keymap + 2))[idx < bits ? idx >> 5 : 0ull];
        // (a MASK, not a select: the compiler sinks a load whose value is only wanted under a condition into that branch — and waits for it there, sixteen
        // dependent bitmap reads per thread and tile)
        const u32 use = 0u - (u32)(((can_bits >> r) & 1u) & (idx < bits ? 1u : 0u));
        words[r] = w & use;
      }
#else
#pragma unroll
    for (int r = 0; r < kJoinR0; r++) {
      const i64 j = base + (i64)r * kBlock + threadIdx.x;
      if (j < n && P::pkeep(prm, j)) alive_bits |= 1u << r;
    }
    if (keymap) {
      u64 keys[kJoinR0];
#pragma unroll
      for (int r = 0; r < kJoinR0; r++) {
        const i64 j = base + (i64)r * kBlock + threadIdx.x;
        keys[r] = 0;
        if (((alive_bits >> r) & 1u) && P::pvalid(prm, j)) { can_bits |= 1u << r; keys[r] = P::pkey0(prm, j); }
      }
      const u64 first = keymap[0], bits = keymap[1];
      u32 words[kJoinR0];
#pragma unroll
      for (int r = 0; r < kJoinR0; r++) {
        const u64 idx = keys[r] - first;
        words[r] = (((can_bits >> r) & 1u) && idx < bits) ? ((const u32*)(keymap + 2))[idx >> 5] : 0u;
      }
#endif
#pragma unroll
      for (int r = 0; r < kJoinR0; r++)
        if (!((words[r] >> ((u32)(keys[r] - first) & 31u)) & 1u)) can_bits &= ~(1u << r);      // the build side does not hold the key: settled
    } else {
#pragma unroll
      for (int r = 0; r < kJoinR0; r++) {
        const i64 j = base + (i64)r * kBlock + threadIdx.x;
        if (((alive_bits >> r) & 1u) && P::pvalid(prm, j)) can_bits |= 1u << r;
      }
    }
    }
#pragma unroll
    for (int r = 0; r < kJoinR0; r++) {
      const bool alive = ((alive_bits >> r) & 1u) != 0, can_match = ((can_bits >> r) & 1u) != 0;
      // a row that cannot match only stays where the join still has to say something about it (the preserved side of an outer join, an anti join)
      const bool keep = alive && (can_match || P::OUTER_PROBE || P::MODE == 2);
      const u64 b = __ballot(keep);
      if (keep) list[m + (u32)__popcll(b & lt)] = (unsigned short)((u32)(r * kBlock + (int)threadIdx.x) | (can_match ? 0u : 0x8000u));
      m += (u32)__popcll(b);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int nslice_all = (int)((m + kWave - 1) / kWave);  // wave-uniform
    // the survivors, eight slices at a time (after a selective filter or key bitmap: one batch, mostly one slice)
    for (int q0 = 0; q0 < kJoinR0; q0 += kJoinR) {
    if (!__syncthreads_or(q0 < nslice_all)) break;          // no wave of the block has a slice left (also orders the reuse of s_wave / s_base)
    const int nslice = nslice_all - q0 < kJoinR ? (nslice_all - q0 < 0 ? 0 : nslice_all - q0) : kJoinR;
    COMET_LDS unsigned short* const list0 = list;
    COMET_LDS unsigned short* list = list0 + (u32)q0 * kWave;
    const u32 m_all = m;
    const u32 m = m_all > (u32)q0 * kWave ? m_all - (u32)q0 * kWave : 0u;
    // ---- 1. probe: key loads + hashes of every slice, then every bucket head, then row by row ----
    u64 hs[kJoinR];
    typename T::Entry he[kJoinR];
    u32 keyed = 0;                                   // bit q: my entry of slice q exists and has a non-NULL key (it can match)
    typename T::Pre pre[kJoinR];
    if (T::UNCONDITIONAL) {
      // No branch around any load: a lane without an entry (or with a NULL key) computes on the tile's first row and drops the result.  Three straight-line
      // sweeps — keys, buckets, the runs' checks — each with all its loads in flight (slices that do not exist cost their instructions, not their latency).
      i64 js[kJoinR];
#pragma unroll
      for (int q = 0; q < kJoinR; q++) {
        const u32 k = (u32)q * kWave + (u32)lane;
        const u32 ent = list[k < m ? k : 0u];
        js[q] = k < m ? base + (i64)(ent & 0x7fffu) : base;
        if (k < m && !(ent & 0x8000u)) keyed |= 1u << q;
      }
#pragma unroll
      for (int q = 0; q < kJoinR; q++) hs[q] = table.key_of(prm, js[q]);
#pragma unroll
      for (int q = 0; q < kJoinR; q++) he[q] = table.peek(hs[q]);
#pragma unroll
      for (int q = 0; q < kJoinR; q++) pre[q] = table.prefetch(prm, js[q], hs[q], he[q]);
#pragma unroll
      for (int q = 0; q < kJoinR; q++)
        if (!((keyed >> q) & 1u)) he[q] = T::none();
    } else {
#pragma unroll
    for (int q = 0; q < kJoinR; q++) {
      hs[q] = 0;
      if (q < nslice) {
        const u32 k = (u32)q * kWave + (u32)lane;
        if (k < m) {
          const u32 ent = list[k];
          if (!(ent & 0x8000u)) {
            hs[q] = table.key_of(prm, base + (i64)(ent & 0x7fffu));
            keyed |= 1u << q;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kJoinR; q++) he[q] = (q < nslice && ((keyed >> q) & 1u)) ? table.peek(hs[q]) : T::none();
#pragma unroll
    for (int q = 0; q < kJoinR; q++) {
      pre[q] = typename T::Pre();
      if (q < nslice && ((keyed >> q) & 1u)) pre[q] = table.prefetch(prm, base + (i64)(list[(u32)q * kWave + (u32)lane] & 0x7fffu), hs[q], he[q]);
    }
    }
    // settle the rows; positions as we go: slice after slice inside the wave (exclusive scan of the emit counts + the wave's running total)
    u32 first[kJoinR], excl[kJoinR];
    u32 cls = 0;                                     // 2 bits per slice: 0 nothing to emit, 1 probe row alone, 2 one build row (first[q]), 3 several
    u32 run = 0;
#pragma unroll
    for (int q = 0; q < kJoinR; q++) {
      first[q] = kJoinEmpty;
      excl[q] = 0;
      if (q < nslice) {                              // wave-uniform
        const u32 k = (u32)q * kWave + (u32)lane;
        u32 eq = 0;
        if (k < m) {
          u32 c = 0, f0 = kJoinEmpty;
          if ((keyed >> q) & 1u) {
            const i64 j = base + (i64)(list[k] & 0x7fffu);
            table.for_each(prm, j, hs[q], he[q], pre[q], [&](u32 row) {
              if (c == 0) f0 = row;
              c++;
              if (P::OUTER_BUILD) matched[row] = 1;    // racing stores of the same value
              return P::MODE == 0;                     // semi / anti only need existence
            });
          }
          first[q] = f0;
          if (P::BUILD_ONLY) eq = 0;
          else if (P::MODE == 1) eq = c ? 1u : 0u;
          else if (P::MODE == 2) eq = c ? 0u : 1u;
          else eq = (c == 0 && P::OUTER_PROBE) ? 1u : c;
          const u32 kind = eq == 0 ? 0u : (P::MODE != 0 || c == 0) ? 1u : c == 1 ? 2u : 3u;
          cls |= kind << (2 * q);
        }
        u32 x = eq;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
          const u32 y = __shfl_up(x, d, kWave);
          if (lane >= d) x += y;
        }
        excl[q] = run + x - eq;
        run += __shfl(x, kWave - 1, kWave);
      }
    }
    // ---- 2. wave after wave inside the tile, ONE global reservation ----
    if (lane == 0) s_wave[wv] = run;
    __syncthreads();
    u32 woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; w++) {
      if (w < wv) woff += s_wave[w];
      total += s_wave[w];
    }
    if (threadIdx.x == 0 && total) s_base = atomicAdd(emitted, (unsigned long long)total);
    __syncthreads();
    // ---- 3. emit ----
    if (total && cls) {
#pragma unroll
      for (int q = 0; q < kJoinR; q++) {
        const u32 kind = (cls >> (2 * q)) & 3u;
        if (q >= nslice || !kind) continue;
        const i64 j = base + (i64)(list[(u32)q * kWave + (u32)lane] & 0x7fffu);
        i64 pos = (i64)s_base + woff + excl[q];
        if (kind == 1) {
          if (pos < cap_out) {
            if (P::MODE != 0) P::emit(prm, -1, j, pos);
            else P::emit_probe_only(prm, j, pos);
          }
        } else if (kind == 2) {
          if (pos < cap_out) P::emit(prm, (i64)first[q], j, pos);
        } else {
          const u64 h = table.key_of(prm, j);            // several matches (rare outside many-to-many joins): walk the candidates again
          const typename T::Entry e2 = table.peek(h);
          table.for_each(prm, j, h, e2, table.prefetch(prm, j, h, e2), [&](u32 row) {
            if (pos < cap_out) P::emit(prm, (i64)row, j, pos);
            pos++;
            return true;
          });
        }
      }
    }
    }                  // batches of slices
    __syncthreads();   // s_wave / s_base / the lists are reused by the next tile
  }
}

// The key bitmap only pays when probe rows MISS: a sample of the probe side (iarg[5] rows, evenly spaced, through the fused chain's filters)
// is looked up in the finished table first — out[47] = { u64 rows alive, u64 rows with a match } — and the executor builds the bitmap
// (join_keymap_build_body: one more pass over the build keys) only when fewer than half of them found a partner.  TPC-DS Q95's self-joins,
// where nearly every probe row has one, skip it (measured with the bitmap always on: 22.6 → 25.4 ms).
template <class P, class T>
CDEV void join_sample_table(const CometKParams& prm, const T& t) {
  unsigned long long* cnt = (unsigned long long*)prm.out[47];
  const i64 ns = prm.iarg[5], n = prm.n;
  const i64 s = (i64)blockIdx.x * kBlock + threadIdx.x;
  bool alive = false, hit = false;
  if (s < ns) {
    const i64 j = (i64)((unsigned __int128)s * (unsigned __int128)n / (unsigned __int128)ns);
    alive = j < n && P::pkeep(prm, j) && P::pvalid(prm, j);
    if (alive) {
      const u64 h = t.key_of(prm, j);
      const typename T::Entry e = t.peek(h);
      t.for_each(prm, j, h, e, t.prefetch(prm, j, h, e), [&](u32) { hit = true; return false; });
    }
  }
  const u64 ba = __ballot(alive), bh = __ballot(hit);
  if (lane_id() == 0) {
    if (ba) atomicAdd(cnt, (unsigned long long)__popcll(ba));
    if (bh) atomicAdd(cnt + 1, (unsigned long long)__popcll(bh));
  }
}
template <class P>
CDEV void join_sample_body(const CometKParams& prm) {
  JoinGlobalTable<P> t{(const u32*)prm.out[0], (const i32*)prm.out[1], (u64)prm.iarg[0] - 1, (int)prm.iarg[2], prm.iarg[1]};
  join_sample_table<P, JoinGlobalTable<P>>(prm, t);
}
template <class P>
CDEV void join_keymap_build_body(const CometKParams& prm) {
  if (!P::KEYMAP) return;
  u64* km = (u64*)prm.out[kJoinKeyMap];
  const i64 nb = prm.iarg[1];
  if (prm.iarg[5] == 0) {
    // nobody asks whether a key came twice (the bitmap filters probe rows, or IS the build side of a semi / anti join): the lanes of a wave whose keys
    // fall into the same word (neighbouring rows of a clustered build side: every one of them) OR their bits together first — one atomic per
    // wave and word instead of one per key (5.6 M device-scope atomics at the memory side cost Q95's three bitmaps 2.9 ms)
    const int lane = lane_id();
    for (i64 wbase = (i64)blockIdx.x * kBlock + (i64)wave_id() * kWave; wbase < nb; wbase += (i64)gridDim.x * kBlock) {
      const i64 i = wbase + lane;
      u64 kw[P::NKW];
      u64 idx = ~0ull;
      if (i < nb && P::bvalid(prm, i)) {
        P::bkeys(prm, i, kw);
        idx = kw[0] - km[0];
      }
      const bool in = idx < km[1];
      u32 word = in ? (u32)(idx >> 5) : 0xffffffffu, bits = in ? 1u << (idx & 31u) : 0u;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {               // lanes further on with MY word (wherever they sit: one OR more into the same word is harmless)
        const u32 ow = __shfl_down(word, d, kWave), ob = __shfl_down(bits, d, kWave);
        if (lane + d < kWave && ow == word) bits |= ob;
      }
      const u32 pw = __shfl_up(word, 1, kWave);
      if (in && (lane == 0 || pw != word)) {
        u32* w = (u32*)(km + 2) + word;
        if ((*w & bits) != bits) atomicOr(w, bits);
      }
    }
    return;
  }
  for (i64 i = (i64)blockIdx.x * kBlock + threadIdx.x; i < nb; i += (i64)gridDim.x * kBlock) {
    if (!P::bvalid(prm, i)) continue;
    u64 kw[P::NKW];
    P::bkeys(prm, i, kw);
    const u64 idx = kw[0] - km[0];
    if (idx < km[1]) {
      u32* w = (u32*)(km + 2) + (idx >> 5);
      const u32 bit = 1u << (idx & 31u);
      // (runs of equal keys: the bit is usually there already) — a bit that was there means the key is NOT unique: out[47][4] says so
      // (one store per WAVE that saw one, and only while the flag is still clear: with duplicate-heavy keys — Q95's 7 M returned order lines — every
      // lane storing into the one word made this pass 1.13 ms instead of 0.1)
      const bool dup = (*w & bit) || (atomicOr(w, bit) & bit);
      if (dup) {
        volatile unsigned long long* flag = (volatile unsigned long long*)prm.out[47] + 4;
        const u64 who = __ballot(true);
        if (lane_id() == (int)__builtin_ctzll(who) && *flag == 0ull) *flag = 1ull;
      }
    }
  }
}
// rows[] of the direct map: every build row at the rank of its key
template <class P>
CDEV void join_direct_rows_body(const CometKParams& prm) {
  if (!P::KEYMAP) return;
  const u64* km = (const u64*)prm.out[kJoinKeyMap];
  const JoinDirectTable<P> t{km, (const u32*)prm.out[0], (const u32*)prm.out[1]};
  u32* rows = (u32*)prm.out[1];
  const i64 nb = prm.iarg[1];
  // "a key came twice" = fewer bits in the bitmap than rows with a key (iarg[6]): the scan behind the bitmap counted the bits — ranks[number of blocks] —, so
  // the bitmap's own build pass need not watch for duplicates and can OR a wave's neighbouring keys together (join_keymap_build_body)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const u64 nblocks = (km[1] + 127) >> 7;
    if ((i64)((const u32*)prm.out[0])[nblocks] != prm.iarg[6]) ((volatile unsigned long long*)prm.out[47])[4] = 1ull;
  }
  for (i64 i = (i64)blockIdx.x * kBlock + threadIdx.x; i < nb; i += (i64)gridDim.x * kBlock) {
    if (!P::bvalid(prm, i)) continue;
    u64 kw[P::NKW];
    P::bkeys(prm, i, kw);
    if (kw[0] - km[0] < km[1]) rows[t.peek(kw[0])] = (u32)i;
  }
}
template <class P>
CDEV void join_probe_direct_body(const CometKParams& prm) {
  const JoinDirectTable<P> t{(const u64*)prm.out[kJoinKeyMap], (const u32*)prm.out[0], (const u32*)prm.out[1]};
  join_probe_tiles<P, JoinDirectTable<P>, true, COMET_JOIN_R0_DIRECT>(prm, t);
}

template <class P, bool KM = false>
CDEV void join_probe_fused_body(const CometKParams& prm) {
  JoinGlobalTable<P> t{(const u32*)prm.out[0], (const i32*)prm.out[1], (u64)prm.iarg[0] - 1, (int)prm.iarg[2], prm.iarg[1]};
  join_probe_tiles<P, JoinGlobalTable<P>, KM>(prm, t);
}

template <class P>
CDEV void join_probe_lds_body(const CometKParams& prm) {
  const i64 nbuild = prm.iarg[1];
  __shared__ u32 s_rows[kJoinLdsCap];
  __shared__ unsigned short s_tags[kJoinLdsCap];
  for (int s2 = threadIdx.x; s2 < kJoinLdsCap; s2 += kBlock) s_rows[s2] = kJoinEmpty;
  __syncthreads();
  for (i64 i = threadIdx.x; i < nbuild; i += kBlock) {
    if (!P::bvalid(prm, i)) continue;
    const u32 hi = (u32)(P::bhash(prm, i) >> 16);
    u32 slot = (hi >> 16) & (kJoinLdsCap - 1);
    for (;;) {
      const u32 old = atomicCAS(&s_rows[slot], kJoinEmpty, (u32)i);
      if (old == kJoinEmpty) { s_tags[slot] = (unsigned short)hi; break; }
      slot = (slot + 1) & (kJoinLdsCap - 1);
    }
  }
  __syncthreads();
  JoinLdsTable<P> t{(const COMET_LDS u32*)s_rows, (const COMET_LDS unsigned short*)s_tags};
  join_probe_tiles<P, JoinLdsTable<P>, false>(prm, t);
}

// ---------------------------------------------------------------------------------------------
// Kernel template D'' (round 6) — the BUCKET TABLE: the general hash join as streaming passes plus ONE random 16-byte access per probe key.
// The chained table above pays, per build run, a device-scope atomicExch on a random word of a 128 MB array (executed at the memory side: a
// 64-byte read-modify-write each — k_jbuild ran at 0.95 TB/s) and, per probe key, up to four dependent random lines (head, build key, next[],
// the residual's column).  Here the build side is PARTITIONED on the bits of its hash that pick the slot, each partition's open-addressing
// table is built in LDS by one workgroup ("hash-join probe staged through LDS open-addressing tables", north_star — the staging is the BUILD:
// no global atomic anywhere) and written out with coalesced 16-byte stores:
//   entry = { u64 sig = the key's 64-bit hash, u32 row = the run's leader, u32 cnt = rows of the run }      row = kJoinNoRow: empty
//   slot(h) = umulhi(h, NP · S) — partition = slot / S, linear probing wraps inside the partition (S = kJoinPartSlots slots, ≈ half full)
// hash_key<1> is a bijection of its one key word, so for single-word keys (every integer / date / float key, decimals ≤ 18 digits) sig == h IS
// key equality: a probe row reads its entry and knows — no build-key gather, no next[] — and touches the build side only for a residual
// condition (P::cond) or for the columns it emits.  Keys of several words verify the leader with P::match (one gather, on a 2^-64 collision or
// a true match only).  A run of equal neighbouring keys (fact tables clustered on the key) is one entry; its followers sit right behind the
// leader, `cnt` says how many: next[] does not exist on this path.
//   k_jphist   G blocks × 1024 threads, block b owns rows [b·chunk, (b+1)·chunk): classify runs, LDS histogram over the NP partitions → cnt[b][p]
//   (static)   comet_launch_join_part_scan: per partition the exclusive prefix over the blocks, in place; tot[p]
//   k_jpscat   same blocks: exclusive scan of tot[] (every block, in LDS) + its own prefix = its cursors; records {sig,row,cnt} scattered to
//              recs[] — partition-major, contiguous per partition; block 0 leaves pstart[0 .. NP]
//   k_jtbuild  one 256-thread block per partition: LDS table (sig / row / cnt planes, 64 KB), LDS atomicCAS claims, coalesced write-out
//   k_jprobe_b join_probe_tiles over JoinBucketTable (k_jprobe_bkm: behind the key bitmap)
// A partition that would fill its table beyond 7/8 (many separate runs of ONE key: a many-to-many join on a scattered low-cardinality key) raises
// out[47][6]; the executor then runs the join over the chained table instead.
// out[0] = table (uint4[NP·S]), out[1] = recs (uint4[leaders]), out[3] = { u32 pstart[NP + 1 … padded to kJoinPartMax + 16], u32 tot[kJoinPartMax], u32 cnt[G][NP] };
// iarg[0] = NP · S, iarg[1] = build rows, iarg[2] = NP, iarg[5] = chunk (rows per block, a multiple of 1024).
// ---------------------------------------------------------------------------------------------
constexpr int kJoinPartSlots = 4096;
constexpr int kJoinPartBlock = 1024;

// The MONOTONE hash.  One integer key whose build values span [kmin, kmin + range): h = (key − kmin) · ⌊2^64 / range⌋ keeps the keys' ORDER — slot(h) = umulhi(h,
// slots) grows with the key — and stays injective (so sig == h is still key equality).  Fact tables arrive clustered AND sorted on their keys (the lines of an
// order follow the order numbers): with this hash the partition pass writes its records nearly in sequence, neighbouring waves of the probe read neighbouring lines
// of the table, and both sides of TPC-DS Q95's 72 M × 72 M self-joins stream where a scrambling hash sent every run to a random line.  Keys in any other order
// lose nothing.  A key distribution that is not spread evenly over its range overflows a partition; the build says so before the table is built and the executor
// takes the scrambling hash.  out[3] u32 words kJoinPartMono…: u64 { mult (0: scrambling hash), kmin, range }.
constexpr int kJoinPartMono = kJoinPartMax + 4;
CDEV u64 join_mono_hash(const u64* mono, u64 key) {
  const u64 idx = key - mono[1];
  return idx < mono[2] ? idx * mono[0] : ~0ull;      // (idx · mult ≤ 2^64 − 1 − mult: all ones is no key's hash — a probe key outside the build side's range finds nothing)
}
// either hash of a key, chosen WITHOUT a branch (the choice is uniform, but a branch would cut the probe tile's straight-line load sweeps into blocks — and the
// compiler waits for a block's loads at its end): both are computed, a mask picks
CDEV u64 join_pick_hash(const u64* mono, u64 key0, u64 scrambled) {
  const u64 pick = mono[0] ? ~0ull : 0ull;
  return (join_mono_hash(mono, key0) & pick) | (scrambled & ~pick);
}
template <class P>
CDEV u64 join_bucket_hash(const CometKParams& prm, const u64* kw) {
  const u64* mono = (const u64*)((const u32*)prm.out[3] + kJoinPartMono);
  if (P::KEYMAP) return join_pick_hash(mono, kw[0], hash_key<P::NKW>(kw));
  return hash_key<P::NKW>(kw);
}

// the run a build row leads: like join_build_classify, but the run's LENGTH comes back instead of next[] markers
template <class P>
CDEV bool join_classify_run(const CometKParams& prm, i64 i, i64 nb, u64& h, u32& cnt) {
  const int lane = (int)(threadIdx.x & (kWave - 1));
  const bool valid = i < nb && P::bvalid(prm, i);
  u64 kw[P::NKW];
#pragma unroll
  for (int w = 0; w < P::NKW; w++) kw[w] = 0;
  if (valid) P::bkeys(prm, i, kw);
  bool same = valid && lane > 0;
#pragma unroll
  for (int w = 0; w < P::NKW; w++) {
    const u32 lo = __shfl_up((u32)kw[w], 1, kWave), hi = __shfl_up((u32)(kw[w] >> 32), 1, kWave);
    same = same && (((u64)hi << 32) | lo) == kw[w];
  }
  const int prev_valid = __shfl_up(valid ? 1 : 0, 1, kWave);      // (its own statement: as the right operand of && only the lanes whose `same` still held would
  same = same && prev_valid != 0;                                 //  execute the shuffle, and a shuffle FROM a lane that sits the instruction out returns anything)
  const u64 followers = __ballot(same);
  const u64 after = lane < kWave - 1 ? followers >> (lane + 1) : 0ull;        // the lanes behind me that continue a run: mine, while the bits are ones
  cnt = P::DEDUP_BUILD ? 1u : 1u + (u32)__builtin_ctzll(~after);
  h = join_bucket_hash<P>(prm, kw);
  return valid && !same;
}

template <class P>
CDEV void join_part_hist_body(const CometKParams& prm) {
  __shared__ u32 s_hist[kJoinPartMax];
  const i64 nb = prm.iarg[1], chunk = prm.iarg[5];
  const u64 slots = (u64)prm.iarg[0];
  const int np = (int)prm.iarg[2];
  u32* cnt = (u32*)prm.out[3] + kJoinPartCntOff;
  for (int p = threadIdx.x; p < np; p += kJoinPartBlock) s_hist[p] = 0;
  __syncthreads();
  const i64 r0 = (i64)blockIdx.x * chunk, r1 = r0 + chunk < nb ? r0 + chunk : nb;
  for (i64 wbase = r0 + (i64)(threadIdx.x & ~(kWave - 1)); wbase < r1; wbase += kJoinPartBlock) {      // wave-uniform: lane l holds row wbase + l
    u64 h;
    u32 c;
    if (join_classify_run<P>(prm, wbase + (threadIdx.x & (kWave - 1)), nb, h, c)) atomicAdd(&s_hist[(u32)(__umul64hi(h, slots) / kJoinPartSlots)], 1u);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < np; p += kJoinPartBlock) cnt[(i64)blockIdx.x * np + p] = s_hist[p];
}

template <class P>
CDEV void join_part_scatter_body(const CometKParams& prm) {
  __shared__ u32 s_cur[kJoinPartMax];
  __shared__ u32 s_wsum[kJoinPartBlock / kWave];
  const i64 nb = prm.iarg[1], chunk = prm.iarg[5];
  const u64 slots = (u64)prm.iarg[0];
  const int np = (int)prm.iarg[2];
  u32* pstart = (u32*)prm.out[3];
  const u32* tot = pstart + kJoinPartTotOff;
  const u32* cnt = pstart + kJoinPartCntOff + (i64)blockIdx.x * np;
  uint4* recs = (uint4*)prm.out[1];
  // exclusive scan of tot[0 .. np): sixteen consecutive partitions per thread
  constexpr int kPer = kJoinPartMax / kJoinPartBlock;
  const int lane = (int)(threadIdx.x & (kWave - 1)), wv = (int)(threadIdx.x >> 6);
  u32 v[kPer], sum = 0;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int p = (int)threadIdx.x * kPer + k;
    v[k] = p < np ? tot[p] : 0u;
    sum += v[k];
  }
  u32 x = sum;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const u32 y = __shfl_up(x, d, kWave);
    if (lane >= d) x += y;
  }
  if (lane == kWave - 1) s_wsum[wv] = x;
  __syncthreads();
  u32 run = x - sum;
  for (int w = 0; w < wv; w++) run += s_wsum[w];
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    const int p = (int)threadIdx.x * kPer + k;
    if (p < np) {
      s_cur[p] = run + cnt[p];
      if (blockIdx.x == 0) pstart[p] = run;
    }
    run += v[k];
    if (blockIdx.x == 0 && p == np - 1) pstart[np] = run;
  }
  __syncthreads();
  const i64 r0 = (i64)blockIdx.x * chunk, r1 = r0 + chunk < nb ? r0 + chunk : nb;
  for (i64 wbase = r0 + (i64)(threadIdx.x & ~(kWave - 1)); wbase < r1; wbase += kJoinPartBlock) {
    const i64 i = wbase + lane;
    u64 h;
    u32 c;
    if (join_classify_run<P>(prm, i, nb, h, c)) {
      const u32 pos = atomicAdd(&s_cur[(u32)(__umul64hi(h, slots) / kJoinPartSlots)], 1u);
      recs[pos] = uint4{(u32)h, (u32)(h >> 32), (u32)i, c};
    }
  }
}

template <class P>
CDEV void join_table_build_body(const CometKParams& prm) {
  __shared__ u64 s_sig[kJoinPartSlots];
  __shared__ u32 s_row[kJoinPartSlots];
  __shared__ u32 s_cnt[kJoinPartSlots];
  const u64 slots = (u64)prm.iarg[0];
  const int np = (int)prm.iarg[2];
  const u32* pstart = (const u32*)prm.out[3];
  const uint4* recs = (const uint4*)prm.out[1];
  uint4* tab = (uint4*)prm.out[0];
  for (int p = blockIdx.x; p < np; p += gridDim.x) {
    for (int s2 = threadIdx.x; s2 < kJoinPartSlots; s2 += kBlock) s_row[s2] = kJoinNoRow;
    __syncthreads();
    const u32 r0 = pstart[p], r1 = pstart[p + 1];
    if (r1 - r0 > (u32)(kJoinPartSlots - kJoinPartSlots / 8)) {
      if (threadIdx.x == 0) ((volatile unsigned long long*)prm.out[47])[6] = 1ull;       // too full to probe in a few steps: the executor takes the chained table
    } else {
      for (u32 r = r0 + threadIdx.x; r < r1; r += kBlock) {
        const uint4 rec = recs[r];
        const u64 h = ((u64)rec.y << 32) | rec.x;
        u32 s2 = (u32)__umul64hi(h, slots) & (u32)(kJoinPartSlots - 1);
        while (atomicCAS(&s_row[s2], kJoinNoRow, rec.z) != kJoinNoRow) s2 = (s2 + 1) & (u32)(kJoinPartSlots - 1);
        s_sig[s2] = h;
        s_cnt[s2] = rec.w;
      }
    }
    __syncthreads();
    for (int s2 = threadIdx.x; s2 < kJoinPartSlots; s2 += kBlock) {
      const u32 row = s_row[s2];
      const u64 sg = row != kJoinNoRow ? s_sig[s2] : 0ull;
      tab[(i64)p * kJoinPartSlots + s2] = uint4{(u32)sg, (u32)(sg >> 32), row, row != kJoinNoRow ? s_cnt[s2] : 0u};
    }
    __syncthreads();
  }
}

template <class P>
struct JoinBucketTable {
  static constexpr bool BY_KEY = false;
  // peek() and prefetch() are safe for ANY lane (any hash lands on a slot; rows are clamped into the run, or to row 0): the tile calls them without a per-lane
  // branch.  A load inside a lane-dependent branch is waited for before the branch ends (s_waitcnt vmcnt(0) in its block) — the eight "independent" bucket loads
  // of a lane then run one after the other: 442 full waits for 610 loads in the first build of this kernel, 84 % of all wave cycles waiting
  // (profiles/r6_q95_join_pmc.txt).  Straight-line loads are issued together and waited for once.
  static constexpr bool UNCONDITIONAL = COMET_JOIN_UNCOND != 0;
  typedef uint4 Entry;
  static CDEV Entry none() { return uint4{0u, 0u, kJoinNoRow, 0u}; }
  const uint4* tab;
  u64 slots;
  const u64* mono;
  CDEV u64 key_of(const CometKParams& prm, i64 j) const {
    if (P::KEYMAP) return join_pick_hash(mono, P::pkey0(prm, j), P::phash(prm, j));
    return P::phash(prm, j);
  }
  CDEV Entry peek(u64 h) const { return tab[__umul64hi(h, slots)]; }
  struct Pre {
    uint4 e2;   // the slot behind the home slot (linear probing: about every second lookup needs it)
    u32 m;      // bit k: row k of the home slot's run holds the key and passes the residual condition (k < kPreRun; bit 0 = the leader)
  };
  // rows of a run whose key / condition checks travel with the prefetch sweep: independent loads, ONE latency for the whole run — walked one after
  // the other (a semi join whose condition fails for every row of the run walks all of it: TPC-DS Q95's single-warehouse orders) they were 5.7
  // dependent latencies per probe row
  static constexpr u32 kPreRun = COMET_JOIN_PRE_RUN;
  static CDEV bool holds(const uint4& e, u64 h) { return e.x == (u32)h && e.y == (u32)(h >> 32); }
  // the leader of an entry whose signature matched: single-word keys are equal already
  static CDEV bool leader_ok(const CometKParams& prm, u32 row, i64 j) { return P::NKW == 1 ? P::cond(prm, (i64)row, j) : P::match(prm, (i64)row, j); }
  CDEV Pre prefetch(const CometKParams& prm, i64 j, u64 h, const Entry& e) const {      // (j: any valid probe row; e: an entry or none())
    Pre p;
    const u64 g = __umul64hi(h, slots);
    if (!UNCONDITIONAL) {
      p.e2 = none();
      p.m = 0;
      if (e.z == kJoinNoRow) return p;
      p.e2 = tab[(g & ~(u64)(kJoinPartSlots - 1)) | ((g + 1) & (u64)(kJoinPartSlots - 1))];
      if (holds(e, h)) {
        p.m = leader_ok(prm, e.z, j) ? 1u : 0u;
        if (P::HAS_COND || P::NKW != 1) {
#pragma unroll
          for (u32 k = 1; k < kPreRun; k++)
            if (k < e.w && follower_ok(prm, e.z + k, j)) p.m |= 1u << k;
        }
      }
      return p;
    }
    p.e2 = tab[(g & ~(u64)(kJoinPartSlots - 1)) | ((g + 1) & (u64)(kJoinPartSlots - 1))];
    const bool hit = e.z != kJoinNoRow && holds(e, h);
    u32 m = hit ? 1u : 0u;
    if (P::HAS_COND || P::NKW != 1) {
      // the run's rows (the first kPreRun of them), every lane: a lane without a hit, or with a shorter run, checks row 0 / the run's last row again and drops the answer
      const u32 row0 = hit ? e.z : 0u, last = hit ? e.z + e.w - 1u : 0u;
      m = 0;
#pragma unroll
      for (u32 k = 0; k < kPreRun; k++) {
        const u32 r = row0 + k <= last ? row0 + k : last;
        const bool ok = k == 0 ? leader_ok(prm, r, j) : follower_ok(prm, r, j);
        m |= (ok ? 1u : 0u) << k;
      }
      m &= hit ? (e.w >= kPreRun ? (1u << kPreRun) - 1u : (1u << e.w) - 1u) : 0u;
    }
    p.m = m;
    return p;
  }
  static CDEV bool follower_ok(const CometKParams& prm, u32 row, i64 j) { return P::NKW == 1 ? (!P::HAS_COND || P::cond(prm, (i64)row, j)) : P::match(prm, (i64)row, j); }
  template <class F>
  CDEV void for_each(const CometKParams& prm, i64 j, u64 h, const Entry& e, const Pre& pre, F f) const {
    if (e.z == kJoinNoRow) return;
    const u64 g = __umul64hi(h, slots);
    uint4 cur = e;
    for (u32 t = 0; t < (u32)kJoinPartSlots; t++) {
      if (holds(cur, h)) {
        const bool mi = t == 0 ? (pre.m & 1u) != 0 : leader_ok(prm, cur.z, j);
        if (mi && !f(cur.z)) return;
        for (u32 k = 1; k < cur.w; k++) {               // the run's followers: the rows right behind the leader (same key; another residual, maybe)
          const bool mk = (!P::HAS_COND && P::NKW == 1) ? true : (t == 0 && k < kPreRun) ? ((pre.m >> k) & 1u) != 0 : follower_ok(prm, cur.z + k, j);
          if (mk && !f(cur.z + k)) return;
        }
      }
      cur = t == 0 ? pre.e2 : tab[(g & ~(u64)(kJoinPartSlots - 1)) | ((g + t + 1) & (u64)(kJoinPartSlots - 1))];
      if (cur.z == kJoinNoRow) return;
    }
  }
};
template <class P, bool KM = false>
CDEV void join_probe_bucket_body(const CometKParams& prm) {
  JoinBucketTable<P> t{(const uint4*)prm.out[0], (u64)prm.iarg[0], (const u64*)((const u32*)prm.out[3] + kJoinPartMono)};
  join_probe_tiles<P, JoinBucketTable<P>, KM>(prm, t);
}
template <class P>
CDEV void join_sample_bucket_body(const CometKParams& prm) {
  JoinBucketTable<P> t{(const uint4*)prm.out[0], (u64)prm.iarg[0], (const u64*)((const u32*)prm.out[3] + kJoinPartMono)};
  join_sample_table<P, JoinBucketTable<P>>(prm, t);
}

// The BITMAP-ONLY join (round 6): a LeftSemi / LeftAnti join that keeps probe rows, has no residual condition (P::DEDUP_BUILD) and joins on one
// integer key whose build values span a foreign key's range asks nothing of its build side but "is the key there" — the key bitmap IS the
// answer.  No table, no build rows: the tile's filter phase looks the bit up (k_jbmap built it), every row that is still "keyed" matches.
template <class P>
struct JoinBitmapTable {
  static constexpr bool BY_KEY = true;
  static constexpr bool UNCONDITIONAL = false;
  typedef u32 Entry;
  static CDEV Entry none() { return kJoinNoRow; }
  CDEV u64 key_of(const CometKParams&, i64) const { return 0ull; }
  CDEV u32 peek(u64) const { return 0u; }
  struct Pre {};
  CDEV Pre prefetch(const CometKParams&, i64, u64, u32) const { return Pre{}; }
  template <class F>
  CDEV void for_each(const CometKParams&, i64, u64, u32, const Pre&, F f) const { f(0u); }
};
template <class P>
CDEV void join_probe_bitmap_body(const CometKParams& prm) {
  join_probe_tiles<P, JoinBitmapTable<P>, true>(prm, JoinBitmapTable<P>{});
}

}  // namespace comet

