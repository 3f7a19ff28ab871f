// String → Float / Double (conversion_funcs/string.rs:177-258): String.trim, the spellings of infinity and NaN, one trailing d / D / f / F, then
// Rust's `str::parse::<f32 / f64>` — a correctly rounded conversion.  Every correctly rounded conversion returns the same float, so this one is
// the shift-by-powers-of-two decimal algorithm (the slow path of Rust's dec2flt, Go's strconv.decimal): the digits live in an array (767
// significant digits decide any double; what lies beyond only says "not zero"), the value is scaled into [0.5, 1) by exact multiplications
// and divisions by powers of two, 24 / 53 bits are extracted and rounded half to even.  No table of powers of ten, no floating-point
// arithmetic.  Compiled by hipcc for the fused kernels and by g++ for tests/test_strtod_cpu.py (against Python's float() and exact rationals).
#pragma once
#include "../strtod_tables.hpp"
#ifndef STRTOD_ENTRY
#define STRTOD_ENTRY CDEV
#endif

struct SdDecimal {
  u8 d[800];        // digits, most significant first
  i32 nd, dp;       // digits used; position of the decimal point (value = 0.d[0]d[1]… × 10^dp)
  bool trunc;       // non-zero digits were dropped behind d[nd − 1]
};
CDEV void sd_trim(SdDecimal& a) {
  while (a.nd > 0 && a.d[a.nd - 1] == 0) a.nd--;
  if (a.nd == 0) a.dp = 0;
}
CDEV void sd_right_shift(SdDecimal& a, u32 k) {      // a /= 2^k, k ≤ 60
  i32 r = 0, w = 0;
  u64 n = 0;
  for (; (n >> k) == 0; r++) {
    if (r >= a.nd) {
      if (n == 0) { a.nd = 0; return; }
      while ((n >> k) == 0) { n *= 10; r++; }
      break;
    }
    n = n * 10 + a.d[r];
  }
  a.dp -= r - 1;
  const u64 mask = ((u64)1 << k) - 1;
  for (; r < a.nd; r++) {
    const u64 dig = n >> k;
    n &= mask;
    a.d[w++] = (u8)dig;
    n = n * 10 + a.d[r];
  }
  while (n > 0) {
    const u64 dig = n >> k;
    n &= mask;
    if (w < 800) a.d[w++] = (u8)dig;
    else if (dig > 0) a.trunc = true;
    n *= 10;
  }
  a.nd = w;
  sd_trim(a);
}
CDEV void sd_left_shift(SdDecimal& a, u32 k) {       // a *= 2^k, k ≤ 60
  i32 delta = kLeftDelta[k];
  {
    const i32 c0 = kLeftCutoffAt[k], cn = kLeftCutoffAt[k + 1] - c0;
    bool less = false;
    i32 i = 0;
    for (; i < cn; i++) {
      if (i >= a.nd) { less = true; break; }
      const u8 c = (u8)(kLeftCutoff[c0 + i] - '0');
      if (a.d[i] != c) { less = a.d[i] < c; break; }
    }
    if (less) delta--;
  }
  i32 r = a.nd, w = a.nd + delta;
  u64 n = 0;
  for (r--; r >= 0; r--) {
    n += (u64)a.d[r] << k;
    const u64 quo = n / 10, rem = n - 10 * quo;
    w--;
    if (w < 800) a.d[w] = (u8)rem;
    else if (rem != 0) a.trunc = true;
    n = quo;
  }
  while (n > 0) {
    const u64 quo = n / 10, rem = n - 10 * quo;
    w--;
    if (w < 800) a.d[w] = (u8)rem;
    else if (rem != 0) a.trunc = true;
    n = quo;
  }
  a.nd += delta;
  if (a.nd >= 800) a.nd = 800;
  a.dp += delta;
  sd_trim(a);
}
CDEV void sd_shift(SdDecimal& a, i32 k) {
  if (a.nd == 0) return;
  while (k > 60) { sd_left_shift(a, 60); k -= 60; }
  if (k > 0) sd_left_shift(a, (u32)k);
  while (k < -60) { sd_right_shift(a, 60); k += 60; }
  if (k < 0) sd_right_shift(a, (u32)-k);
}
CDEV u64 sd_rounded_integer(const SdDecimal& a) {     // the integer part, the fraction rounded half to even
  if (a.dp > 20) return ~(u64)0;
  i32 i = 0;
  u64 n = 0;
  for (; i < a.dp && i < a.nd; i++) n = n * 10 + a.d[i];
  for (; i < a.dp; i++) n *= 10;
  bool up = false;
  if (a.dp >= 0 && a.dp < a.nd) {
    if (a.d[a.dp] == 5 && a.dp + 1 == a.nd) up = a.trunc || (a.dp > 0 && (a.d[a.dp - 1] & 1) != 0);      // exactly halfway: to even
    else up = a.d[a.dp] >= 5;
  }
  return n + (up ? 1 : 0);
}
// the decimal → the float's bits (mantbits 52 / 23, expbits 11 / 8, bias −1023 / −127); too large → infinity
CDEV u64 sd_float_bits(SdDecimal& a, int mantbits, int expbits, int bias) {
  const u64 inf = (((u64)1 << expbits) - 1) << mantbits;
  i32 exp = 0;
  u64 mant = 0;
  if (a.nd == 0) return 0;
  if (a.dp > 310) return inf;
  if (a.dp < -330) return 0;
  {
    const i32 powtab[9] = {1, 3, 6, 9, 13, 16, 19, 23, 26};
    while (a.dp > 0) {
      const i32 n = a.dp >= 9 ? 27 : powtab[a.dp];
      sd_shift(a, -n);
      exp += n;
    }
    while (a.dp < 0 || (a.dp == 0 && a.d[0] < 5)) {
      const i32 n = -a.dp >= 9 ? 27 : powtab[-a.dp];
      sd_shift(a, n);
      exp -= n;
    }
  }
  exp--;                                  // [0.5, 1) here, [1, 2) in the float
  if (exp < bias + 1) {                   // below the smallest normal exponent: a subnormal's scale
    const i32 n = bias + 1 - exp;
    sd_shift(a, -n);
    exp += n;
  }
  if (exp - bias >= (1 << expbits) - 1) return inf;
  sd_shift(a, 1 + mantbits);
  mant = sd_rounded_integer(a);
  if (mant == ((u64)2 << mantbits)) {     // rounding carried into the next binade
    mant >>= 1;
    exp++;
    if (exp - bias >= (1 << expbits) - 1) return inf;
  }
  if ((mant & ((u64)1 << mantbits)) == 0) exp = bias;      // subnormal
  return (mant & (((u64)1 << mantbits) - 1)) | ((u64)(exp - bias) << mantbits);
}
CDEV bool sd_word(const u8* p, i32 a, i32 b, const char* w, i32 wn) {
  if (b - a != wn) return false;
  for (i32 k = 0; k < wn; k++) {
    u8 ch = p[a + k];
    if (ch >= 'A' && ch <= 'Z') ch = (u8)(ch + 32);
    if (ch != (u8)w[k]) return false;
  }
  return true;
}
// → 0 and the float's bits (sign included), or 1: not a number (NULL, CAST_INVALID_INPUT under ANSI)
STRTOD_ENTRY int str_to_float_bits(const u8* p, i32 n, bool is32, u64& out) {
  const int mantbits = is32 ? 23 : 52, expbits = is32 ? 8 : 11, bias = is32 ? -127 : -1023;
  const u64 inf = (((u64)1 << expbits) - 1) << mantbits, signbit = (u64)1 << (mantbits + expbits);
  i32 a = 0, b = n;
  while (a < b && p[a] <= 0x20) a++;
  while (b > a && p[b - 1] <= 0x20) b--;
  if (sd_word(p, a, b, "inf", 3) || sd_word(p, a, b, "+inf", 4) || sd_word(p, a, b, "infinity", 8) || sd_word(p, a, b, "+infinity", 9)) { out = inf; return 0; }
  if (sd_word(p, a, b, "-inf", 4) || sd_word(p, a, b, "-infinity", 9)) { out = inf | signbit; return 0; }
  const u64 qnan = inf | ((u64)1 << (mantbits - 1));
  if (sd_word(p, a, b, "nan", 3)) { out = qnan; return 0; }
  if (b > a && (p[b - 1] == 'd' || p[b - 1] == 'D' || p[b - 1] == 'f' || p[b - 1] == 'F')) b--;
  // Rust's float grammar: Sign? ( inf | infinity | nan | Digit* ( '.' Digit* )? ( [eE] Sign? Digit+ )? ) with at least one digit in the number
  if (a == b) return 1;
  bool neg = false;
  if (p[a] == '+' || p[a] == '-') { neg = p[a] == '-'; a++; }
  if (sd_word(p, a, b, "inf", 3) || sd_word(p, a, b, "infinity", 8)) { out = inf | (neg ? signbit : 0); return 0; }
  if (sd_word(p, a, b, "nan", 3)) { out = qnan | (neg ? signbit : 0); return 0; }      // (the sign of a NaN: Rust keeps it)
  SdDecimal dec;
  dec.nd = 0;
  dec.dp = 0;
  dec.trunc = false;
  bool saw_digit = false, saw_dot = false;
  i32 nall = 0;                        // digits behind the leading zeroes, kept or not
  // The mantissa's end first (digits and points), then a loop over it that nothing leaves early.  One loop that both `continue`d (leading zeroes) and `break`ed
  // (the first other character) handed ROCm 7.2's device compiler a wrong exit index — every number with an exponent came back as "not a number", on the GPU only
  // (the same compiler's host build, UBSan / ASan / MSan clean, and ROCm 7.0's device build were right; profiles/r6_jit_compiler.md).
  i32 mend = a;
  while (mend < b && (p[mend] == '.' || (p[mend] >= '0' && p[mend] <= '9'))) mend++;
  i32 i = a;
  for (; i < mend; i++) {
    const u8 ch = p[i];
    if (ch == '.') {
      if (saw_dot) return 1;
      saw_dot = true;
      dec.dp += nall;                  // (leading zeroes behind the point have been counted negatively below)
    } else if (ch >= '0' && ch <= '9') {
      saw_digit = true;
      if (ch == '0' && nall == 0) {    // leading zeroes: before the point they mean nothing, behind it they move it
        if (saw_dot) dec.dp--;
        continue;
      }
      nall++;
      if (dec.nd < 800) dec.d[dec.nd++] = (u8)(ch - '0');
      else if (ch != '0') dec.trunc = true;
    }
  }
  if (!saw_digit) return 1;
  if (!saw_dot) dec.dp = nall;
  if (i < b) {
    if (p[i] != 'e' && p[i] != 'E') return 1;
    i++;
    if (i >= b) return 1;
    bool eneg = false;
    if (p[i] == '+' || p[i] == '-') { eneg = p[i] == '-'; i++; }
    if (i >= b) return 1;
    i64 e = 0;
    for (; i < b; i++) {
      if (p[i] < '0' || p[i] > '9') return 1;
      if (e < 100000) e = e * 10 + (p[i] - '0');
    }
    dec.dp += (i32)(eneg ? -e : e);
  }
  sd_trim(dec);
  out = sd_float_bits(dec, mantbits, expbits, bias) | (neg ? signbit : 0);
  return 0;
}
