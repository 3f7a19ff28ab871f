// String → Timestamp / Timestamp_NTZ (conversion_funcs/string.rs:798-852, 1125-1900).  The reference matches fourteen regular expressions
// (Unicode \d), splits the value on [T -:.] and lets integer parses that fail fall back to defaults; a zone suffix (Z, UTC±h:mm, GMT…, EST / MST /
// HST, ±hh:mm, ±hhmm) replaces the session zone.  Restated here shape by shape over the value's bytes.  Needs the "time zones" section of
// comet_device.hpp (tz_local_spans …) and regex_unicode_tables.hpp (\d).  Return codes: 0 a value, 1 invalid (NULL, CAST_INVALID_INPUT under ANSI),
// 2 NULL in every mode, 4 a NAMED zone inside the value (" Europe/Moscow": the device holds the session zone's table only), 6 a time-only value when the caller passes no "today".  Time-only values ("T12:34", "12:34:56") take today's date in the zone: `now_us` says what today is.
#pragma once
#include "../regex_unicode_tables.hpp"
#ifndef STRTS_ENTRY
#define STRTS_ENTRY CDEV
#endif

CDEV u32 ts_cp_at(const u8* p, i32 b, i32& i) {        // one scalar value of valid UTF-8, i advances
  const u32 c = p[i];
  if (c < 0x80) { i++; return c; }
  if (c < 0xE0 && i + 1 < b) { const u32 v = ((c & 0x1F) << 6) | (p[i + 1] & 0x3F); i += 2; return v; }
  if (c < 0xF0 && i + 2 < b) { const u32 v = ((c & 0x0F) << 12) | ((u32)(p[i + 1] & 0x3F) << 6) | (p[i + 2] & 0x3F); i += 3; return v; }
  if (i + 3 < b) { const u32 v = ((c & 0x07) << 18) | ((u32)(p[i + 1] & 0x3F) << 12) | ((u32)(p[i + 2] & 0x3F) << 6) | (p[i + 3] & 0x3F); i += 4; return v; }
  i = b;
  return 0xFFFD;
}
CDEV bool ts_is_ws(u32 c) {       // White_Space: what str::trim removes
  return (c >= 9 && c <= 13) || c == 32 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
CDEV bool ts_is_nd(u32 c) {       // \d of the regex crate: General_Category = Nd
  if (c < 0x80) return c >= '0' && c <= '9';
  i32 lo = 0, hi = (i32)(sizeof(kPerlDigit) / sizeof(kPerlDigit[0]));
  while (lo < hi) {
    const i32 mid = (lo + hi) >> 1;
    if ((u32)kPerlDigit[mid][1] < c) lo = mid + 1; else hi = mid;
  }
  return lo < (i32)(sizeof(kPerlDigit) / sizeof(kPerlDigit[0])) && (u32)kPerlDigit[lo][0] <= c;
}
CDEV void ts_trim(const u8* p, i32& a, i32& b, bool left, bool right) {
  while (left && a < b) { i32 j = a; if (!ts_is_ws(ts_cp_at(p, b, j))) break; a = j; }
  while (right && b > a) {
    i32 k = b - 1;
    while (k > a && (p[k] & 0xC0) == 0x80) k--;
    i32 j = k;
    if (!ts_is_ws(ts_cp_at(p, b, j))) break;
    b = k;
  }
}
// \d{lo,hi} at i: the count taken (greedy; the shapes are anchored so greedy is what the regex does), −1 when fewer than lo
CDEV i32 ts_digits(const u8* p, i32 b, i32& i, i32 lo, i32 hi) {
  i32 n = 0;
  while (i < b && n < hi) { i32 j = i; if (!ts_is_nd(ts_cp_at(p, b, j))) break; i = j; n++; }
  return n >= lo ? n : -1;
}
// shapes: 0 year, 1 month, 2 day, 3 hour, 4 minute, 5 second, 6 microsecond, 7..10 T h / hm / hms / hmsu, 11..13 bare hm / hms / hmsu; −1 none
CDEV int ts_time_tail(const u8* p, i32 i, i32 b, int base) {      // behind the hour digits of a time-only value: (:\d{1,2}(:\d{1,2}(\.\d+)?)?)?
  if (i == b) return base;
  if (p[i] != ':') return -1;
  i++;
  if (ts_digits(p, b, i, 1, 2) < 0) return -1;
  if (i == b) return base + 1;
  if (p[i] != ':') return -1;
  i++;
  if (ts_digits(p, b, i, 1, 2) < 0) return -1;
  if (i == b) return base + 2;
  if (p[i] != '.') return -1;
  i++;
  if (ts_digits(p, b, i, 1, 0x7fffffff) < 0) return -1;
  return i == b ? base + 3 : -1;
}
CDEV int ts_shape(const u8* p, i32 a, i32 b) {
  if (a >= b) return -1;
  i32 i = a;
  if (p[i] == 'T') {
    i++;
    if (ts_digits(p, b, i, 1, 2) < 0) return -1;
    return ts_time_tail(p, i, b, 7);
  }
  if (p[i] == '-') i++;
  const i32 i0 = i;
  const i32 n = ts_digits(p, b, i, 1, 7);
  if (n < 0) return -1;
  if (i == b) return (n >= 4 && n <= 6) ? 0 : -1;
  if (p[i] == ':' && p[a] != '-' && n <= 2) {       // a bare time: \d{1,2}:\d{1,2}…
    i32 j = i0;
    (void)ts_digits(p, b, j, 1, 2);
    const int t = ts_time_tail(p, j, b, 10);          // base 10 + (1: hm → 11, 2: hms → 12, 3: hmsu → 13)
    return t >= 11 ? t : -1;
  }
  if (n < 4 || p[i] != '-') return -1;
  i++;
  if (ts_digits(p, b, i, 2, 2) < 0) return -1;
  if (i == b) return 1;
  if (p[i] != '-') return -1;
  i++;
  if (ts_digits(p, b, i, 2, 2) < 0) return -1;
  if (i == b) return 2;
  if (p[i] != 'T' && p[i] != ' ') return -1;
  i++;
  const i32 h0 = i;
  const i32 nh = ts_digits(p, b, i, 1, 2);
  if (nh < 0) return -1;
  if (i == b) return 3;
  if (nh != 2 || p[i] != ':') return -1;
  (void)h0;
  i++;
  if (ts_digits(p, b, i, 2, 2) < 0) return -1;
  if (i == b) return 4;
  if (p[i] != ':') return -1;
  i++;
  if (ts_digits(p, b, i, 2, 2) < 0) return -1;
  if (i == b) return 5;
  if (p[i] != '.') return -1;
  i++;
  if (ts_digits(p, b, i, 1, 0x7fffffff) < 0) return -1;
  return i == b ? 6 : -1;
}
// str::parse::<i32 / u32> of p[a, b): ASCII digits behind an optional sign; false when it fails (the callers then take their default)
CDEV bool ts_int(const u8* p, i32 a, i32 b, bool is_signed, i64& out) {
  if (a >= b) return false;
  bool neg = false;
  if (p[a] == '+' || (is_signed && p[a] == '-')) { neg = p[a] == '-'; a++; }
  if (a >= b) return false;
  i64 v = 0;
  for (i32 k = a; k < b; k++) {
    if (p[k] < '0' || p[k] > '9') return false;
    v = v * 10 + (p[k] - '0');
    if (v > 4294967295ll) return false;
  }
  if (neg) v = -v;
  if (is_signed ? (v < -2147483648ll || v > 2147483647ll) : (v < 0 || v > 4294967295ll)) return false;
  out = v;
  return true;
}
struct TsInfo { i64 f[7]; };      // year, month, day, hour, minute, second, microsecond
// parse_to_timestamp_info (string.rs:1125-1205): the value split on [T -:.] behind one leading '-'; false: the year is out of the range that can be a timestamp
CDEV bool ts_info(const u8* p, i32 a, i32 b, int kind, TsInfo& out) {
  i64 sign = 1;
  if (a < b && p[a] == '-') { sign = -1; a++; }
  i64 got[7] = {0, 1, 1, 0, 0, 0, 0};
  i32 seg = 0, s0 = a;
  for (i32 k = a; k <= b && seg < 7; k++) {
    const bool sep = k == b || p[k] == 'T' || p[k] == ' ' || p[k] == '-' || p[k] == ':' || p[k] == '.';
    if (!sep) continue;
    i64 v;
    if (seg == 0) got[0] = sign * (ts_int(p, s0, k, true, v) ? v : 0);
    else if (seg < 6) { if (ts_int(p, s0, k, false, v)) got[seg] = v; }
    else {
      i32 e = k - s0 > 6 ? s0 + 6 : k;                // the first six BYTES of the fraction
      i64 scale = 1;
      for (i32 q = e - s0; q < 6; q++) scale *= 10;
      got[6] = ts_int(p, s0, e, false, v) ? v * scale : 0;
      // (the seventh part ends at the next separator: "….123.456" cannot pass the shape, so k == b here)
    }
    seg++;
    s0 = k + 1;
  }
  if (got[0] < -290309 || got[0] > 294248) return false;
  const i64 dflt[7] = {1, 1, 1, 0, 0, 0, 0};
  for (int k = 0; k < 7; k++) out.f[k] = k <= kind ? got[k] : dflt[k];
  return true;
}
CDEV bool ts_epoch_day(i64 y, i64 m, i64 d, i64& days) {      // ymd_to_epoch_day (string.rs:1237-1247)
  if (m < 1 || m > 12) return false;
  const bool leap = y % 4 == 0 && (y % 100 != 0 || y % 400 == 0);
  const i64 mx = m == 2 ? (leap ? 29 : 28) : (m == 4 || m == 6 || m == 9 || m == 11) ? 30 : 31;
  if (d < 1 || d > mx) return false;
  days = str_days_from_civil(y, m, d);
  return true;
}
// the zone a value is read in: the session zone's table, or a fixed offset taken from the value's suffix
struct TsZone { tzp zt; bool fixed; i64 off; };
CDEV int ts_zone_spans(const TsZone& z, i64 L, i64& off) {
  if (z.fixed) { off = z.off; return 1; }
  return tz_local_spans(z.zt, tz_fold_local(z.zt, L), off);
}
// parse_timestamp_to_micros (string.rs:1249-1348)
CDEV int ts_to_micros(const TsInfo& t, const TsZone& z, i64& out) {
  const i64 y = t.f[0], h = t.f[3], mi = t.f[4], s = t.f[5];
  if (h >= 24 || mi >= 60 || s >= 60) return 1;
  i64 days = 0;
  const bool valid = ts_epoch_day(y, t.f[1], t.f[2], days);
  if (valid && y >= -262143 && y <= 262142) {
    const i64 L = days * 86400 + h * 3600 + mi * 60 + s;
    i64 off = 0;
    if (ts_zone_spans(z, L, off) == 0 && ts_zone_spans(z, L - 10800, off) == 0) return 1;
    out = (L - off) * 1000000 + t.f[6];
    return 0;
  }
  if ((y >= -262144 && y <= 262143) || !valid) return 1;
  i64 off = 0;
  if (ts_zone_spans(z, 0, off) == 0) off = 0;
  const i128 us = ((i128)days * 86400 + h * 3600 + mi * 60 + s - off) * 1000000 + t.f[6];
  if (us < -(i128)0x7fffffffffffffffll - 1 || us > (i128)0x7fffffffffffffffll) return 1;
  out = (i64)us;
  return 0;
}
// parse_sign_offset (string.rs:1493-1531) of p[a, b)
CDEV bool ts_sign_offset(const u8* p, i32 a, i32 b, i64& secs) {
  if (a == b) { secs = 0; return true; }
  i64 sign;
  if (p[a] == '+') sign = 1; else if (p[a] == '-') sign = -1; else return false;
  a++;
  if (a == b) return false;
  i32 colon = -1;
  for (i32 k = a; k < b; k++) if (p[k] == ':') { colon = k; break; }
  i64 h = 0, m = 0;
  if (colon >= 0) {
    if (colon + 1 == b) return false;
    if (!ts_int(p, a, colon, true, h) || !ts_int(p, colon + 1, b, true, m)) return false;
  } else if (b - a == 1 || b - a == 2) {
    if (!ts_int(p, a, b, true, h)) return false;
  } else if (b - a == 4) {
    for (i32 k = a; k < b; k++) if (p[k] >= 0x80) return false;      // (a slice through a character: the reference would not get here with ASCII digits)
    if (!ts_int(p, a, a + 2, true, h) || !ts_int(p, a + 2, b, true, m)) return false;
  } else {
    return false;
  }
  if (h < 0 || h > 18 || m < 0 || m > 59) return false;
  secs = sign * (h * 3600 + m * 60);
  return true;
}
CDEV i32 ts_rfind(const u8* p, i32 a, i32 b, const char* w, i32 wn) {
  for (i32 k = b - wn; k >= a; k--) {
    bool eq = true;
    for (i32 q = 0; q < wn && eq; q++) eq = p[k + q] == (u8)w[q];
    if (eq) return k;
  }
  return -1;
}
// extract_offset_suffix (string.rs:1566-1641): → 1 and the prefix's end + the offset; 0 no suffix; 4 a named zone
CDEV int ts_suffix(const u8* p, i32 a, i32 b, i32& end, i64& secs) {
  if (b > a && p[b - 1] == 'Z') { end = b - 1; secs = 0; return 1; }
  {
    const char* pre[6] = {" UTC", "UTC", " GMT", "GMT", " UT", "UT"};
    const i32 len[6] = {4, 3, 4, 3, 3, 2};
    for (int k = 0; k < 6; k++) {
      const i32 pos = ts_rfind(p, a, b, pre[k], len[k]);
      if (pos >= 0 && ts_sign_offset(p, pos + len[k], b, secs)) { end = pos; return 1; }
    }
  }
  {
    const char* ab[6] = {" EST", "EST", " MST", "MST", " HST", "HST"};
    const i32 len[6] = {4, 3, 4, 3, 4, 3};
    const i64 off[6] = {-18000, -18000, -25200, -25200, -36000, -36000};
    for (int k = 0; k < 6; k++) {
      const i32 pos = ts_rfind(p, a, b, ab[k], len[k]);
      if (pos >= 0 && pos + len[k] == b) { end = pos; secs = off[k]; return 1; }
    }
  }
  {
    i32 sp = -1;
    for (i32 k = b - 1; k >= a; k--) if (p[k] == ' ') { sp = k; break; }
    if (sp >= 0)
      for (i32 k = sp + 1; k < b; k++) if (p[k] == '/') return 4;
  }
  i32 pos = -1;
  for (i32 k = b - 1; k >= a; k--) if (p[k] == '+' || p[k] == '-') { pos = k; break; }
  if (pos >= 0 && ts_sign_offset(p, pos, b, secs)) { end = pos; return 1; }
  return 0;
}
// a leading '+' (string.rs:1440-1450): "+2020-…" loses it, anything else that starts with '+' is NULL
CDEV bool ts_leading_plus(const u8* p, i32& a, i32 b) {
  if (a >= b || p[a] != '+') return true;
  i32 k = a + 1;
  while (k < b && p[k] >= '0' && p[k] <= '9') k++;
  if (k < b && k - (a + 1) >= 1 && p[k] == '-') { a++; return true; }
  return false;
}
// parse_str_to_time_only_timestamp (string.rs:1855-1893)
CDEV int ts_time_only(const u8* p, i32 a, i32 b, const TsZone& z, i64 now_us, i64& out) {
  if (a < b && p[a] == 'T') a++;
  i64 part[3] = {0, 0, 0}, us = 0;
  i32 s0 = a, seg = 0;
  for (i32 k = a; k <= b && seg < 3; k++) {
    if (k < b && !(p[k] == ':' && seg < 2)) continue;
    i32 e = k;
    if (seg == 2) {
      i32 dot = -1;
      for (i32 q = s0; q < k; q++) if (p[q] == '.') { dot = q; break; }
      if (dot >= 0) {
        e = dot;
        i32 fe = k - (dot + 1) > 6 ? dot + 1 + 6 : k;
        i64 v, scale = 1;
        for (i32 q = fe - (dot + 1); q < 6; q++) scale *= 10;
        us = ts_int(p, dot + 1, fe, false, v) ? v * scale : 0;
      }
    }
    i64 v;
    part[seg] = ts_int(p, s0, e, false, v) ? v : 0;
    seg++;
    s0 = k + 1;
  }
  if (part[0] >= 24 || part[1] >= 60 || part[2] >= 60) return 1;
  i64 off_now = 0;
  if (z.fixed) off_now = z.off; else off_now = tz_offset_at(z.zt, tz_floor_div(now_us, 1000000));
  const i64 day = tz_floor_div(now_us + off_now * 1000000, 86400000000ll);
  const i64 L = day * 86400 + part[0] * 3600 + part[1] * 60 + part[2];
  i64 off = 0;
  if (ts_zone_spans(z, L, off) != 1) return 1;      // `.single()`: a gap or an overlap is None
  out = (L - off) * 1000000 + us;
  return 0;
}
STRTS_ENTRY int str_to_timestamp(const u8* p, i32 n, tzp zt, bool spark4, i64 now_us, i64& out) {
  i32 a = 0, b = n;
  ts_trim(p, a, b, false, true);                     // the cast's trim_end
  const i32 a_raw = a;
  ts_trim(p, a, b, true, true);
  if (a == b) return 2;
  if (spark4 && a > a_raw) {
    const int k = ts_shape(p, a, b);
    if (k >= 7 && k <= 10) return 1;
  }
  if (!ts_leading_plus(p, a, b)) return 2;
  TsZone z = {zt, false, 0};
  int kind = ts_shape(p, a, b);
  if (kind < 0) {
    i32 end = b;
    i64 secs = 0;
    const int rc = ts_suffix(p, a, b, end, secs);
    if (rc == 4) return 4;
    if (rc == 1) { b = end; z.fixed = true; z.off = secs; kind = ts_shape(p, a, b); }
  }
  if (kind < 0) return 1;
  if (kind <= 6) {
    TsInfo t;
    if (!ts_info(p, a, b, kind, t)) return 1;
    return ts_to_micros(t, z, out);
  }
  if (now_us == (i64)0x8000000000000000ull) return 6;      // the caller has no "today" to give: a time-only value is refused
  return ts_time_only(p, a, b, z, now_us, out);
}
STRTS_ENTRY int str_to_timestamp_ntz(const u8* p, i32 n, i64& out) {
  i32 a = 0, b = n;
  ts_trim(p, a, b, true, true);
  if (a == b) return 2;
  if (!ts_leading_plus(p, a, b)) return 2;
  int kind = ts_shape(p, a, b);
  if (kind >= 7) return 1;
  if (kind < 0) {
    i32 end = b;
    i64 secs = 0;
    const int rc = ts_suffix(p, a, b, end, secs);
    if (rc == 4) return 4;
    if (rc == 1) { b = end; ts_trim(p, a, b, false, true); kind = ts_shape(p, a, b); }
  }
  if (kind < 0 || kind > 6) return 1;
  TsInfo t;
  if (!ts_info(p, a, b, kind, t)) return 2;
  i64 days = 0;
  if (!ts_epoch_day(t.f[0], t.f[1], t.f[2], days) || t.f[3] >= 24 || t.f[4] >= 60 || t.f[5] >= 60) return 1;
  const i128 us = ((i128)days * 86400 + t.f[3] * 3600 + t.f[4] * 60 + t.f[5]) * 1000000 + t.f[6];
  if (us < -(i128)0x7fffffffffffffffll - 1 || us > (i128)0x7fffffffffffffffll) return 1;
  out = (i64)us;
  return 0;
}
