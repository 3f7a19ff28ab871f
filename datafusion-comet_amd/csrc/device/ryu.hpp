// Shortest round-trip decimal digits of a float — the Ryu algorithm (Ulf Adams, PLDI 2018), which the reference links as the `ryu` crate for
// Float → Decimal (conversion_funcs/numeric.rs:965-990 float_to_decimal128) and whose digits Rust's float Display (Float → String,
// numeric.rs:137-221) equals — and the two casts built on it.  Tables: ryu_tables.hpp (generated).  Plain integer code: compiled by hipcc for
// exchange_kernels.hip / the fused kernels and by g++ for tests/test_ryu_cpu.py, where it is compared with Python's repr() (doubles) and numpy's
// unique formatting (floats).
#pragma once
#include "../ryu_tables.hpp"
#ifndef RYU_ENTRY
#define RYU_ENTRY CDEV      // (the fused kernels' copy makes the entry points real calls: inlined into every unrolled row they took seconds to compile)
#endif

struct RyuDec { u64 mant; i32 exp; };      // value = mant · 10^exp, mant without trailing zeroes (0 for zero)

CDEV u32 ryu_pow5bits(i32 e) { return (u32)(((u32)e * 1217359u) >> 19) + 1u; }
CDEV u32 ryu_log10pow2(i32 e) { return ((u32)e * 78913u) >> 18; }
CDEV u32 ryu_log10pow5(i32 e) { return ((u32)e * 732923u) >> 20; }
CDEV u32 ryu_pow5factor(u64 v) { u32 c = 0; while (v != 0 && v % 5 == 0) { v /= 5; c++; } return c; }
CDEV bool ryu_mult_pow5(u64 v, u32 p) { return ryu_pow5factor(v) >= p; }
CDEV bool ryu_mult_pow2(u64 v, u32 p) { return (v & (((u64)1 << p) - 1)) == 0; }
CDEV u64 ryu_mulshift64(u64 m, const unsigned long long* mul, i32 j) {
  const u128 b0 = (u128)m * mul[0], b2 = (u128)m * mul[1];
  return (u64)(((b0 >> 64) + b2) >> (j - 64));
}
CDEV u32 ryu_mulshift32(u32 m, u64 factor, i32 shift) {
  const u64 bits0 = (u64)m * (u32)factor, bits1 = (u64)m * (u32)(factor >> 32);
  return (u32)(((bits0 >> 32) + bits1) >> (shift - 32));
}
// the digit-removal loop both widths share (d2s.c / f2s.c step 4)
CDEV RyuDec ryu_shorten(u64 vr, u64 vp, u64 vm, i32 e10, bool vm_tz, bool vr_tz, bool accept, u32 last) {
  i32 removed = 0;
  u64 out;
  if (vm_tz || vr_tz) {
    while (vp / 10 > vm / 10) {
      vm_tz = vm_tz && vm % 10 == 0;
      vr_tz = vr_tz && last == 0;
      last = (u32)(vr % 10);
      vr /= 10; vp /= 10; vm /= 10;
      removed++;
    }
    if (vm_tz) {
      while (vm % 10 == 0) {
        vr_tz = vr_tz && last == 0;
        last = (u32)(vr % 10);
        vr /= 10; vp /= 10; vm /= 10;
        removed++;
      }
    }
    if (vr_tz && last == 5 && vr % 2 == 0) last = 4;      // exactly one half: round to even
    out = vr + (((vr == vm && (!accept || !vm_tz)) || last >= 5) ? 1 : 0);
  } else {
    while (vp / 10 > vm / 10) {
      last = (u32)(vr % 10);
      vr /= 10; vp /= 10; vm /= 10;
      removed++;
    }
    out = vr + ((vr == vm || last >= 5) ? 1 : 0);
  }
  RyuDec r = {out, e10 + removed};
  return r;
}
// a finite, non-zero double's shortest digits (sign dropped)
CDEV RyuDec ryu_d2d(u64 bits) {
  const u64 ieee_m = bits & (((u64)1 << 52) - 1);
  const u32 ieee_e = (u32)((bits >> 52) & 0x7FFu);
  i32 e2;
  u64 m2;
  if (ieee_e == 0) { e2 = 1 - 1023 - 52 - 2; m2 = ieee_m; }
  else { e2 = (i32)ieee_e - 1023 - 52 - 2; m2 = ((u64)1 << 52) | ieee_m; }
  const bool accept = (m2 & 1) == 0;
  const u64 mv = 4 * m2;
  const u32 mm_shift = (ieee_m != 0 || ieee_e <= 1) ? 1u : 0u;
  u64 vr, vp, vm;
  i32 e10;
  bool vm_tz = false, vr_tz = false;
  if (e2 >= 0) {
    const u32 q = ryu_log10pow2(e2) - (e2 > 3 ? 1u : 0u);
    e10 = (i32)q;
    const i32 k = 125 + (i32)ryu_pow5bits((i32)q) - 1;
    const i32 i = -e2 + (i32)q + k;
    vr = ryu_mulshift64(4 * m2, kRyuDoublePow5InvSplit[q], i);
    vp = ryu_mulshift64(4 * m2 + 2, kRyuDoublePow5InvSplit[q], i);
    vm = ryu_mulshift64(4 * m2 - 1 - mm_shift, kRyuDoublePow5InvSplit[q], i);
    if (q <= 21) {
      if (mv % 5 == 0) vr_tz = ryu_mult_pow5(mv, q);
      else if (accept) vm_tz = ryu_mult_pow5(mv - 1 - mm_shift, q);
      else vp -= ryu_mult_pow5(mv + 2, q) ? 1 : 0;
    }
  } else {
    const u32 q = ryu_log10pow5(-e2) - (-e2 > 1 ? 1u : 0u);
    e10 = (i32)q + e2;
    const i32 i = -e2 - (i32)q;
    const i32 k = (i32)ryu_pow5bits(i) - 125;
    const i32 j = (i32)q - k;
    vr = ryu_mulshift64(4 * m2, kRyuDoublePow5Split[i], j);
    vp = ryu_mulshift64(4 * m2 + 2, kRyuDoublePow5Split[i], j);
    vm = ryu_mulshift64(4 * m2 - 1 - mm_shift, kRyuDoublePow5Split[i], j);
    if (q <= 1) {
      vr_tz = true;
      if (accept) vm_tz = mm_shift == 1;
      else vp--;
    } else if (q < 63) {
      vr_tz = ryu_mult_pow2(mv, q);
    }
  }
  RyuDec r = ryu_shorten(vr, vp, vm, e10, vm_tz, vr_tz, accept, 0);
  while (r.mant % 10 == 0) { r.mant /= 10; r.exp++; }      // (an integer-valued double: the digits carry no trailing zeroes)
  return r;
}
// … and a finite, non-zero float's
CDEV RyuDec ryu_f2d(u32 bits) {
  const u32 ieee_m = bits & ((1u << 23) - 1);
  const u32 ieee_e = (bits >> 23) & 0xFFu;
  i32 e2;
  u32 m2;
  if (ieee_e == 0) { e2 = 1 - 127 - 23 - 2; m2 = ieee_m; }
  else { e2 = (i32)ieee_e - 127 - 23 - 2; m2 = (1u << 23) | ieee_m; }
  const bool accept = (m2 & 1) == 0;
  const u32 mv = 4 * m2, mp = 4 * m2 + 2;
  const u32 mm_shift = (ieee_m != 0 || ieee_e <= 1) ? 1u : 0u;
  const u32 mm = 4 * m2 - 1 - mm_shift;
  u32 vr, vp, vm, last = 0;
  i32 e10;
  bool vm_tz = false, vr_tz = false;
  if (e2 >= 0) {
    const u32 q = ryu_log10pow2(e2);
    e10 = (i32)q;
    const i32 k = 59 + (i32)ryu_pow5bits((i32)q) - 1;
    const i32 i = -e2 + (i32)q + k;
    vr = ryu_mulshift32(mv, kRyuFloatPow5InvSplit[q], i);
    vp = ryu_mulshift32(mp, kRyuFloatPow5InvSplit[q], i);
    vm = ryu_mulshift32(mm, kRyuFloatPow5InvSplit[q], i);
    if (q != 0 && (vp - 1) / 10 <= vm / 10) {
      // the digit the loop below will not see: the last one removed
      const i32 l = 59 + (i32)ryu_pow5bits((i32)q - 1) - 1;
      last = ryu_mulshift32(mv, kRyuFloatPow5InvSplit[q - 1], -e2 + (i32)q - 1 + l) % 10;
    }
    if (q <= 9) {
      if (mv % 5 == 0) vr_tz = ryu_mult_pow5(mv, q);
      else if (accept) vm_tz = ryu_mult_pow5(mm, q);
      else vp -= ryu_mult_pow5(mp, q) ? 1 : 0;
    }
  } else {
    const u32 q = ryu_log10pow5(-e2);
    e10 = (i32)q + e2;
    const i32 i = -e2 - (i32)q;
    const i32 k = (i32)ryu_pow5bits(i) - 61;
    i32 j = (i32)q - k;
    vr = ryu_mulshift32(mv, kRyuFloatPow5Split[i], j);
    vp = ryu_mulshift32(mp, kRyuFloatPow5Split[i], j);
    vm = ryu_mulshift32(mm, kRyuFloatPow5Split[i], j);
    if (q != 0 && (vp - 1) / 10 <= vm / 10) {
      j = (i32)q - 1 - ((i32)ryu_pow5bits(i + 1) - 61);
      last = ryu_mulshift32(mv, kRyuFloatPow5Split[i + 1], j) % 10;
    }
    if (q <= 1) {
      vr_tz = true;
      if (accept) vm_tz = mm_shift == 1;
      else vp--;
    } else if (q < 31) {
      vr_tz = ryu_mult_pow2(mv, q - 1);
    }
  }
  RyuDec r = ryu_shorten(vr, vp, vm, e10, vm_tz, vr_tz, accept, last);
  while (r.mant % 10 == 0) { r.mant /= 10; r.exp++; }
  return r;
}

// ---- Float → String (numeric.rs:137-221): Rust's `{}` between 10⁻³ and 10⁷ (and for zero) with ".0" behind whole numbers, `{:E}` with a
// fractional digit elsewhere, Java's spelling of the smallest subnormal.  Writes at most 32 bytes, returns the count.
CDEV i32 ryu_digits(u64 m, u8* d) {
  u8 t[20];
  i32 k = 0;
  do { t[k++] = (u8)('0' + (int)(m % 10)); m /= 10; } while (m);
  for (i32 q = 0; q < k; q++) d[q] = t[k - 1 - q];
  return k;
}
CDEV i32 ryu_format(bool neg, bool is_zero, bool plain, RyuDec v, u8* o) {
  i32 k = 0;
  if (neg) o[k++] = '-';
  if (is_zero) { o[k++] = '0'; o[k++] = '.'; o[k++] = '0'; return k; }
  u8 d[20];
  const i32 n = ryu_digits(v.mant, d);
  if (plain) {
    const i32 point = n + v.exp;                    // digits in front of the decimal point
    if (point <= 0) {
      o[k++] = '0'; o[k++] = '.';
      for (i32 q = 0; q < -point; q++) o[k++] = '0';
      for (i32 q = 0; q < n; q++) o[k++] = d[q];
    } else if (point >= n) {
      for (i32 q = 0; q < n; q++) o[k++] = d[q];
      for (i32 q = n; q < point; q++) o[k++] = '0';
      o[k++] = '.'; o[k++] = '0';
    } else {
      for (i32 q = 0; q < point; q++) o[k++] = d[q];
      o[k++] = '.';
      for (i32 q = point; q < n; q++) o[k++] = d[q];
    }
    return k;
  }
  o[k++] = d[0];
  o[k++] = '.';
  if (n == 1) o[k++] = '0';
  for (i32 q = 1; q < n; q++) o[k++] = d[q];
  o[k++] = 'E';
  i32 e = v.exp + n - 1;
  if (e < 0) { o[k++] = '-'; e = -e; }
  u8 ed[20];
  const i32 en = ryu_digits((u64)e, ed);
  for (i32 q = 0; q < en; q++) o[k++] = ed[q];
  return k;
}
CDEV i32 ryu_lit(const char* s, u8* o) { i32 k = 0; while (s[k]) { o[k] = (u8)s[k]; k++; } return k; }
CDEV i32 fmt_f64_bits(u64 bits, u8* o) {
  const bool neg = (bits >> 63) != 0;
  const u64 a = bits & ~((u64)1 << 63);
  if (a > 0x7FF0000000000000ull) return ryu_lit("NaN", o);
  if (a == 0x7FF0000000000000ull) return ryu_lit(neg ? "-Infinity" : "Infinity", o);
  if (a == 1) return ryu_lit(neg ? "-4.9E-324" : "4.9E-324", o);
  RyuDec z = {0, 0};
  if (a == 0) return ryu_format(neg, true, true, z, o);
  // 0.001 ≤ |v| < 10⁷ as doubles: 0x3F50624DD2F1A9FC is 0.001
  const bool plain = a >= 0x3F50624DD2F1A9FCull && a < 0x416312D000000000ull;
  return ryu_format(neg, false, plain, ryu_d2d(a), o);
}
CDEV i32 fmt_f32_bits(u32 bits, u8* o) {
  const bool neg = (bits >> 31) != 0;
  const u32 a = bits & 0x7FFFFFFFu;
  if (a > 0x7F800000u) return ryu_lit("NaN", o);
  if (a == 0x7F800000u) return ryu_lit(neg ? "-Infinity" : "Infinity", o);
  if (a == 1) return ryu_lit(neg ? "-1.4E-45" : "1.4E-45", o);
  RyuDec z = {0, 0};
  if (a == 0) return ryu_format(neg, true, true, z, o);
  // 0.001f32 = 0x3A83126F, 1e7f32 = 0x4B189680
  const bool plain = a >= 0x3A83126Fu && a < 0x4B189680u;
  return ryu_format(neg, false, plain, ryu_f2d(a), o);
}

// ---- Float → Decimal (numeric.rs:884-990): BigDecimal(Double.toString(d)).setScale(scale, HALF_UP) — the SHORTEST digits are rounded, not the
// binary value.  A float is widened to double first.  0 = a value, 2 = NULL in every mode (NaN / infinity), 3 = does not fit the precision.
RYU_ENTRY int f64_bits_to_decimal(u64 bits, int precision, int scale, i128& out) {
  const bool neg = (bits >> 63) != 0;
  const u64 a = bits & ~((u64)1 << 63);
  if (a >= 0x7FF0000000000000ull) return 2;
  if (a == 0) { out = 0; return 0; }
  const RyuDec v = ryu_d2d(a);
  const i32 shift = v.exp + scale;
  u128 mag;
  if (shift >= 0) {
    if (shift > 38) return 3;
    u128 p = 1;
    for (i32 k = 0; k < shift; k++) p *= 10;
    const u128 lim = (((u128)1 << 127) - 1) / p;
    if ((u128)v.mant > lim) return 3;
    mag = (u128)v.mant * p;
  } else if (-shift > 38) {
    mag = 0;
  } else {
    u128 p = 1;
    for (i32 k = 0; k < -shift; k++) p *= 10;
    mag = (u128)v.mant / p;
    if ((u128)v.mant % p >= p / 2) mag++;
  }
  u128 bound = 1;
  for (int k = 0; k < precision; k++) bound *= 10;
  if (mag >= bound) return 3;
  out = neg ? -(i128)mag : (i128)mag;
  return 0;
}
