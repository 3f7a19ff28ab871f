// Rust's str::to_lowercase / str::to_uppercase over one value's UTF-8 bytes (what DataFusion's `lower` / `upper` — the reference's Lower / Upper —
// evaluate): full Unicode case mapping without locale (ß → SS, İ → i̇, ŉ → ʼN), and the Final_Sigma rule of to_lowercase (library/alloc/src/str.rs
// map_uppercase_sigma: Σ becomes ς when a cased letter precedes it — case-ignorable characters skipped — and none follows).  Tables:
// case_tables.hpp (generated, Unicode 16.0: tools/gen_case_tables.py).  Plain integer code: compiled by hipcc for exchange_kernels.hip and by
// g++ for tests/test_case_map_cpu.py (CDEV = static inline there).
#pragma once
#include "../case_tables.hpp"

CDEV u32 case_decode(const u8* p, i32 n, i32& i) {          // one scalar value of valid UTF-8 at p[i]
  const u32 c = p[i];
  if (c < 0x80 || i + 1 >= n) { i++; return c; }
  if (c < 0xE0) { const u32 v = ((c & 0x1F) << 6) | (p[i + 1] & 0x3F); i += 2; return v; }
  if (c < 0xF0) { if (i + 2 >= n) { i = n; return 0xFFFD; } const u32 v = ((c & 0x0F) << 12) | ((u32)(p[i + 1] & 0x3F) << 6) | (p[i + 2] & 0x3F); i += 3; return v; }
  if (i + 3 >= n) { i = n; return 0xFFFD; }
  const u32 v = ((c & 0x07) << 18) | ((u32)(p[i + 1] & 0x3F) << 12) | ((u32)(p[i + 2] & 0x3F) << 6) | (p[i + 3] & 0x3F);
  i += 4;
  return v;
}
CDEV i32 case_encode(u32 cp, u8* o) {                        // o == nullptr: the length only
  if (cp < 0x80) { if (o) o[0] = (u8)cp; return 1; }
  if (cp < 0x800) { if (o) { o[0] = (u8)(0xC0 | (cp >> 6)); o[1] = (u8)(0x80 | (cp & 0x3F)); } return 2; }
  if (cp < 0x10000) { if (o) { o[0] = (u8)(0xE0 | (cp >> 12)); o[1] = (u8)(0x80 | ((cp >> 6) & 0x3F)); o[2] = (u8)(0x80 | (cp & 0x3F)); } return 3; }
  if (o) { o[0] = (u8)(0xF0 | (cp >> 18)); o[1] = (u8)(0x80 | ((cp >> 12) & 0x3F)); o[2] = (u8)(0x80 | ((cp >> 6) & 0x3F)); o[3] = (u8)(0x80 | (cp & 0x3F)); }
  return 4;
}
CDEV bool case_in_ranges(const unsigned int (*t)[2], i32 n, u32 cp) {
  i32 lo = 0, hi = n;
  while (lo < hi) {
    const i32 mid = (lo + hi) >> 1;
    if (t[mid][1] < cp) lo = mid + 1; else hi = mid;
  }
  return lo < n && t[lo][0] <= cp;
}
CDEV i32 case_find(const unsigned int* keys, i32 n, u32 cp) {
  i32 lo = 0, hi = n;
  while (lo < hi) {
    const i32 mid = (lo + hi) >> 1;
    if (keys[mid] < cp) lo = mid + 1; else hi = mid;
  }
  return (lo < n && keys[lo] == cp) ? lo : -1;
}
#define CASE_COUNT(a) ((i32)(sizeof(a) / sizeof(a[0])))
CDEV bool case_is_ignorable(u32 cp) { return case_in_ranges(kCaseIgnorable, CASE_COUNT(kCaseIgnorable), cp); }
CDEV bool case_is_cased(u32 cp) { return case_in_ranges(kCaseCased, CASE_COUNT(kCaseCased), cp); }
// Σ at byte `at` of p[0, n): word-final?
CDEV bool case_final_sigma(const u8* p, i32 n, i32 at) {
  bool before = false;
  for (i32 j = at; j > 0;) {
    i32 k = j - 1;
    while (k > 0 && (p[k] & 0xC0) == 0x80) k--;
    i32 q = k;
    const u32 cp = case_decode(p, n, q);
    j = k;
    if (case_is_ignorable(cp)) continue;
    before = case_is_cased(cp);
    break;
  }
  if (!before) return false;
  for (i32 j = at + 2; j < n;) {
    const u32 cp = case_decode(p, n, j);
    if (case_is_ignorable(cp)) continue;
    return !case_is_cased(cp);
  }
  return true;
}
// mode 1: to_lowercase, 2: to_uppercase.  Writes to `o` when it is not NULL; returns the mapped value's byte length.
CDEV i32 case_map_value(const u8* p, i32 n, int mode, u8* o) {
  i32 len = 0;
  for (i32 i = 0; i < n;) {
    const i32 at = i;
    const u32 cp = case_decode(p, n, i);
    if (cp < 0x80) {
      u32 m = cp;
      if (mode == 1 && cp >= 'A' && cp <= 'Z') m = cp + 32;
      if (mode == 2 && cp >= 'a' && cp <= 'z') m = cp - 32;
      if (o) o[len] = (u8)m;
      len++;
      continue;
    }
    if (mode == 1 && cp == 0x3A3) {
      len += case_encode(case_final_sigma(p, n, at) ? 0x3C2u : 0x3C3u, o ? o + len : nullptr);
      continue;
    }
    const i32 k = mode == 1 ? case_find(kCaseLowerKeys, CASE_COUNT(kCaseLowerKeys), cp) : case_find(kCaseUpperKeys, CASE_COUNT(kCaseUpperKeys), cp);
    if (k < 0) {
      for (i32 b = at; b < i; b++) { if (o) o[len] = p[b]; len++; }      // unchanged: the bytes as they are
      continue;
    }
    const unsigned int* m = mode == 1 ? kCaseLowerVals[k] : kCaseUpperVals[k];
    for (int q = 0; q < 3 && m[q]; q++) len += case_encode(m[q], o ? o + len : nullptr);
  }
  return len;
}
