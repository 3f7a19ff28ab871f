// Calendar arithmetic on epoch days and wall-clock microseconds (the proleptic Gregorian calendar chrono uses): what the date functions of the
// generated kernels call — datetime_funcs/{date_trunc,next_day,make_date}.rs, kernels/temporal.rs:63-100, 179-270, DataFusion's date_part,
// datafusion-spark's last_day.  Plain C++: comet_device.hpp includes it for the device, the host runs the same source for the CPU tests
// (capi.cpp comet_date_fn_host) against Python's datetime.
#pragma once
#ifndef CDEV
#define CDEV inline
typedef long long i64;
typedef int i32;
typedef unsigned int u32;
#endif

// days_from_civil (string.rs:1221-1228)
CDEV i64 str_days_from_civil(i64 y, i64 m, i64 d) {
  if (m <= 2) { y -= 1; m += 9; } else m -= 3;
  const i64 era = (y >= 0 ? y : y - 399) / 400;
  const i64 yoe = y - era * 400;
  const i64 doy = (153 * m + 2) / 5 + d - 1;
  const i64 doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + doe - 719468;
}
// proleptic Gregorian civil date from days since 1970-01-01 (Howard Hinnant's algorithm; what chrono / arrow's date_part use)
CDEV void civil_from_days(i32 z0, i32& y, i32& m, i32& d) {
  i64 z = (i64)z0 + 719468;
  const i64 era = (z >= 0 ? z : z - 146096) / 146097;
  const i64 doe = z - era * 146097;
  const i64 yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const i64 doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const i64 mp = (5 * doy + 2) / 153;
  d = (i32)(doy - (153 * mp + 2) / 5 + 1);
  m = (i32)(mp < 10 ? mp + 3 : mp - 9);
  y = (i32)(yoe + era * 400 + (m <= 2 ? 1 : 0));
}
CDEV i32 date_part(i32 days, int part) {   // 0 year, 1 month, 2 day, 3 quarter, 4 dow (Sunday = 0), 5 doy (1-based)
  i32 y, m, d;
  civil_from_days(days, y, m, d);
  switch (part) {
    case 0: return y;
    case 1: return m;
    case 2: return d;
    case 3: return (m - 1) / 3 + 1;
    case 4: { i64 w = ((i64)days + 4) % 7; return (i32)(w < 0 ? w + 7 : w); }   // 1970-01-01 was a Thursday
    default: {
      const bool leap = (y % 4 == 0 && y % 100 != 0) || y % 400 == 0;
      const int cum[12] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334};
      return cum[m - 1] + d + ((leap && m > 2) ? 1 : 0);
    }
  }
}

// ---- more of the date functions (datetime_funcs/{date_trunc,next_day,make_date}.rs, kernels/temporal.rs:63-100, datafusion-spark's last_day) ----
// chrono holds the years -262143 ..= 262142: a date beyond them is NULL to every function that reads it as a calendar date
CDEV bool date_in_chrono_range(i32 days) { return days >= -96465292 && days <= 95026236; }
CDEV i32 date_weekday_mon0(i32 days) { const i64 w = ((i64)days + 3) % 7; return (i32)(w < 0 ? w + 7 : w); }   // Monday = 0 (1970-01-01 was a Thursday)
CDEV i32 date_days_in_month(i32 y, i32 m) {
  const bool leap = (y % 4 == 0 && y % 100 != 0) || y % 400 == 0;
  return m == 2 ? (leap ? 29 : 28) : (m == 4 || m == 6 || m == 9 || m == 11) ? 30 : 31;
}
// ISO-8601 week of the year (DataFusion date_part 'week' = chrono's iso_week().week(); Spark's weekofyear): the week of the date's Thursday
CDEV i32 date_iso_week(i32 days) {
  const i32 thu = days - date_weekday_mon0(days) + 3;
  return (date_part(thu, 5) - 1) / 7 + 1;
}
// unit: 0 year, 1 quarter, 2 month, 3 week (Monday) — trunc_days_to_* (kernels/temporal.rs:63-100)
CDEV i32 date_trunc_days(i32 days, int unit) {
  if (unit == 3) return days - date_weekday_mon0(days);
  i32 y, m, d;
  civil_from_days(days, y, m, d);
  if (unit == 2) return days - (d - 1);
  if (unit == 0) return days - (date_part(days, 5) - 1);
  return (i32)str_days_from_civil(y, ((m - 1) / 3) * 3 + 1, 1);
}
CDEV i32 date_last_day(i32 days) {
  i32 y, m, d;
  civil_from_days(days, y, m, d);
  return days - d + date_days_in_month(y, m);
}
// next_date_for_day_of_week (next_day.rs:64-68): the first date LATER than `days` that falls on the weekday (Monday = 0)
CDEV i32 date_next_day(i32 days, int target_mon0) {
  const int since = (date_weekday_mon0(days) - target_mon0 + 7) % 7;
  return (i32)((u32)days + (u32)(7 - since));
}
// make_date (make_date.rs:87-96): chrono's from_ymd_opt — month 1..12, a day the month has, a year chrono holds
CDEV bool date_make(i32 y, i32 m, i32 d, i32& out) {
  if (m < 1 || m > 12 || d < 1 || d > 31 || y < -262143 || y > 262142 || d > date_days_in_month(y, m)) return false;
  out = (i32)str_days_from_civil(y, m, d);
  return true;
}
CDEV i64 floor_div_i64(i64 a, i64 b) { const i64 q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
// timestamp_trunc on the zone's wall clock (kernels/temporal.rs:179-270): unit 0 year, 1 quarter, 2 month, 3 week, 4 day, 5 hour, 6 minute, 7 second,
// 8 millisecond, 9 microsecond
CDEV i64 ts_trunc_local_us(i64 us, int unit) {
  if (unit == 9) return us;
  if (unit >= 5) {
    const i64 q = unit == 5 ? 3600000000ll : unit == 6 ? 60000000ll : unit == 7 ? 1000000ll : 1000ll;
    return floor_div_i64(us, q) * q;
  }
  const i64 day = floor_div_i64(us, 86400000000ll);
  if (unit == 4) return day * 86400000000ll;
  return (i64)date_trunc_days((i32)day, unit) * 86400000000ll;
}
