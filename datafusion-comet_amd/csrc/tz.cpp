// See tz.hpp.  TZif (RFC 8536): the 64-bit block of a version 2+ file — transition instants, the local time type in force after each, and the
// POSIX TZ string of the footer that describes the time after the last transition ("slim" files carry everything after 2007 there).
#include "tz.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>

#include "plan.hpp"

namespace comet {

bool fixed_zone_offset(const std::string& tz, long long& secs) {
  if (tz.empty() || tz == "UTC" || tz == "Z" || tz == "GMT" || tz == "Etc/UTC" || tz == "Etc/GMT" || tz == "UCT" || tz == "Etc/UCT" || tz == "Zulu" || tz == "Etc/Zulu" || tz == "Universal" ||
      tz == "Etc/Universal" || tz == "UT") {
    secs = 0;
    return true;
  }
  std::string s = tz;
  if (s.rfind("UTC", 0) == 0 || s.rfind("GMT", 0) == 0) s = s.substr(3);
  else if (s.rfind("UT", 0) == 0) s = s.substr(2);
  if (s.size() < 2 || (s[0] != '+' && s[0] != '-')) return false;
  int part[3] = {0, 0, 0}, np = 0, nd = 0;
  for (size_t i = 1; i < s.size(); i++) {
    if (s[i] == ':') { if (nd == 0 || ++np > 2) return false; nd = 0; continue; }
    if (s[i] < '0' || s[i] > '9' || ++nd > 2) return false;
    part[np] = part[np] * 10 + (s[i] - '0');
  }
  if (nd == 0 || part[0] > 18 || part[1] > 59 || part[2] > 59) return false;
  secs = (long long)part[0] * 3600 + part[1] * 60 + part[2];
  if (s[0] == '-') secs = -secs;
  return true;
}

std::vector<int64_t> ZoneTable::flat() const {
  std::vector<int64_t> f;
  f.reserve(3 + 2 * at.size());
  f.push_back((int64_t)at.size());
  f.push_back(first_off);
  f.push_back(limit);
  f.insert(f.end(), at.begin(), at.end());
  for (int32_t o : off) f.push_back(o);
  return f;
}

namespace {

int64_t days_from_civil(int64_t y, int64_t m, int64_t d) {
  y -= m <= 2;
  const int64_t era = (y >= 0 ? y : y - 399) / 400;
  const int64_t yoe = y - era * 400;
  const int64_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
  return era * 146097 + yoe * 365 + yoe / 4 - yoe / 100 + doy - 719468;
}
bool leap(int64_t y) { return y % 4 == 0 && (y % 100 != 0 || y % 400 == 0); }
int64_t year_of(int64_t utc_s) {
  int64_t z = (utc_s >= 0 ? utc_s : utc_s - 86399) / 86400 + 719468;
  const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  const int64_t doe = z - era * 146097;
  const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const int64_t mp = (5 * doy + 2) / 153;
  return yoe + era * 400 + (mp >= 10 ? 1 : 0);
}

// ---- POSIX TZ string: std offset [dst [offset] [, start [/time], end [/time]]]
struct Rule { int kind = 0; int m = 0, w = 0, d = 0; int64_t time = 7200; };      // kind 0: Mm.w.d, 1: Jn (1..365, no leap day), 2: n (0..365)
struct Posix { int32_t std_off = 0, dst_off = 0; bool has_dst = false; Rule start, end; };

struct P {
  const std::string& s;
  size_t i = 0;
  bool name() {
    if (i < s.size() && s[i] == '<') {
      const size_t e = s.find('>', i);
      if (e == std::string::npos) return false;
      i = e + 1;
      return true;
    }
    const size_t b = i;
    while (i < s.size() && ((s[i] >= 'A' && s[i] <= 'Z') || (s[i] >= 'a' && s[i] <= 'z'))) i++;
    return i - b >= 3;
  }
  bool hms(int64_t& out) {      // [+-]hh[:mm[:ss]]
    int sign = 1;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) { sign = s[i] == '-' ? -1 : 1; i++; }
    int64_t v[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
      const size_t b = i;
      while (i < s.size() && s[i] >= '0' && s[i] <= '9') v[k] = v[k] * 10 + (s[i++] - '0');
      if (i == b) return false;
      if (i < s.size() && s[i] == ':' && k < 2) i++;
      else break;
    }
    out = sign * (v[0] * 3600 + v[1] * 60 + v[2]);
    return true;
  }
  bool rule(Rule& r) {
    if (i >= s.size()) return false;
    auto num = [&](int& out) {
      const size_t b = i;
      out = 0;
      while (i < s.size() && s[i] >= '0' && s[i] <= '9') out = out * 10 + (s[i++] - '0');
      return i > b;
    };
    if (s[i] == 'M') {
      i++;
      r.kind = 0;
      if (!num(r.m) || i >= s.size() || s[i++] != '.' || !num(r.w) || i >= s.size() || s[i++] != '.' || !num(r.d)) return false;
      if (r.m < 1 || r.m > 12 || r.w < 1 || r.w > 5 || r.d > 6) return false;
    } else if (s[i] == 'J') {
      i++;
      r.kind = 1;
      if (!num(r.d) || r.d < 1 || r.d > 365) return false;
    } else {
      r.kind = 2;
      if (!num(r.d) || r.d > 365) return false;
    }
    r.time = 7200;
    if (i < s.size() && s[i] == '/') { i++; if (!hms(r.time)) return false; }
    return true;
  }
};

bool parse_posix(const std::string& s, Posix& out) {
  P p{s};
  int64_t v;
  if (!p.name() || !p.hms(v)) return false;
  out.std_off = (int32_t)-v;              // POSIX counts west of Greenwich positive
  if (p.i >= s.size()) return true;
  if (!p.name()) return false;
  out.has_dst = true;
  out.dst_off = out.std_off + 3600;
  if (p.i < s.size() && s[p.i] != ',') { if (!p.hms(v)) return false; out.dst_off = (int32_t)-v; }
  if (p.i >= s.size()) return false;       // a DST name without rules: not something the database writes
  if (s[p.i++] != ',' || !p.rule(out.start) || p.i >= s.size() || s[p.i++] != ',' || !p.rule(out.end)) return false;
  return p.i == s.size();
}
// the rule's day of `year` as seconds since the epoch of that day's local midnight
int64_t rule_day(const Rule& r, int64_t year) {
  if (r.kind == 1) {
    int d = r.d;                                  // 1..365, February 29 never counted
    if (leap(year) && d >= 60) d++;
    return (days_from_civil(year, 1, 1) + d - 1) * 86400;
  }
  if (r.kind == 2) return (days_from_civil(year, 1, 1) + r.d) * 86400;
  const int64_t first = days_from_civil(year, r.m, 1);
  const int wd = (int)(((first % 7) + 11) % 7);   // 1970-01-01 was a Thursday (4): weekday of the month's first day, 0 = Sunday
  int64_t day = first + ((r.d - wd) % 7 + 7) % 7 + (int64_t)(r.w - 1) * 7;
  static const int dim[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  const int64_t last = first + dim[r.m - 1] + (r.m == 2 && leap(year) ? 1 : 0) - 1;
  while (day > last) day -= 7;                    // week 5 = the last such weekday of the month
  return day * 86400;
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int64_t be64(const uint8_t* p) { return (int64_t)(((uint64_t)be32(p) << 32) | be32(p + 4)); }

std::shared_ptr<const ZoneTable> parse_tzif(const std::vector<uint8_t>& b, const std::string& name) {
  auto bad = [&]() -> CometError { return CometError("time zone '" + name + "': not a TZif file of version 2 or later"); };
  if (b.size() < 44 || memcmp(b.data(), "TZif", 4) != 0 || b[4] < '2') throw bad();
  auto counts = [&](size_t o, uint32_t c[6]) { for (int k = 0; k < 6; k++) c[k] = be32(b.data() + o + 20 + 4 * k); };      // isut, isstd, leap, time, type, char
  uint32_t c[6];
  counts(0, c);
  const size_t v1 = 44 + (size_t)c[3] * 5 + (size_t)c[4] * 6 + c[5] + (size_t)c[2] * 8 + c[1] + c[0];
  if (b.size() < v1 + 44 || memcmp(b.data() + v1, "TZif", 4) != 0) throw bad();
  counts(v1, c);
  const size_t timecnt = c[3], typecnt = c[4];
  size_t o = v1 + 44;
  const size_t need = timecnt * 9 + typecnt * 6 + c[5] + (size_t)c[2] * 12 + c[1] + c[0];
  if (typecnt == 0 || b.size() < o + need) throw bad();
  const uint8_t* times = b.data() + o;
  const uint8_t* idx = times + timecnt * 8;
  const uint8_t* types = idx + timecnt;
  auto z = std::make_shared<ZoneTable>();
  auto type_off = [&](size_t t) {
    if (t >= typecnt) throw bad();
    return (int32_t)be32(types + t * 6);
  };
  z->first_off = type_off(0);         // RFC 8536 §3.2: the time before the first transition is type 0's
  for (size_t k = 0; k < timecnt; k++) {
    const int64_t t = be64(times + k * 8);
    if (!z->at.empty() && t <= z->at.back()) throw bad();
    z->at.push_back(t);
    z->off.push_back(type_off(idx[k]));
  }
  // the footer: "\n" TZ string "\n"
  o += need;
  std::string tzs;
  if (o < b.size() && b[o] == '\n') {
    const uint8_t* e = (const uint8_t*)memchr(b.data() + o + 1, '\n', b.size() - o - 1);
    if (e) tzs.assign((const char*)b.data() + o + 1, (const char*)e);
  }
  if (tzs.empty()) return z;
  Posix px;
  if (!parse_posix(tzs, px)) throw CometError("time zone '" + name + "': cannot read the rule '" + tzs + "' of its TZif footer");
  const int64_t last = z->at.empty() ? INT64_MIN : z->at.back();
  if (!px.has_dst) {
    if (z->at.empty()) z->first_off = px.std_off;
    else if (z->off.back() != px.std_off) { /* the file's last type and its footer disagree: the footer describes the time after the last transition */ z->at.push_back(last + 1); z->off.push_back(px.std_off); }
    return z;
  }
  // a zone with daylight-saving rules: the rule expanded over 400 years behind the file's last transition and a little more — the device reads a
  // later instant 400-year periods earlier (comet_device.hpp tz_fold), inside [limit − 400 years, limit), which must be all the rule's
  const int64_t kLastYear = z->at.empty() ? 2400 : std::max<int64_t>(2400, year_of(last) + 403);
  const int64_t y0 = z->at.empty() ? 1900 : std::max<int64_t>(year_of(last) - 1, 1800);
  for (int64_t y = y0; y < kLastYear; y++) {
    // DST starts at the rule's wall time on the STANDARD clock and ends at its wall time on the DST clock
    std::pair<int64_t, int32_t> ev[2] = {{rule_day(px.start, y) + px.start.time - px.std_off, px.dst_off}, {rule_day(px.end, y) + px.end.time - px.dst_off, px.std_off}};
    if (ev[1].first < ev[0].first) std::swap(ev[0], ev[1]);
    for (auto& e : ev) {
      if (e.first <= last) continue;
      if (!z->off.empty() && z->off.back() == e.second) continue;      // (the first generated event repeats the offset already in force)
      if (z->off.empty() && z->first_off == e.second) continue;
      z->at.push_back(e.first);
      z->off.push_back(e.second);
    }
  }
  z->limit = days_from_civil(kLastYear, 1, 1) * 86400 - 86400 * 2;
  return z;
}

}  // namespace

std::shared_ptr<const ZoneTable> load_zone(const std::string& name) {
  static std::mutex mu;
  static std::map<std::string, std::shared_ptr<const ZoneTable>> cache;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(name);
    if (it != cache.end()) return it->second;
  }
  std::shared_ptr<const ZoneTable> z;
  long long secs = 0;
  if (fixed_zone_offset(name, secs)) {
    auto f = std::make_shared<ZoneTable>();
    f->first_off = (int32_t)secs;
    z = f;
  } else {
    // a region name: letters, digits, '_', '-', '+', '/' — and no way out of the database's directory
    if (name.empty() || name[0] == '/' || name.find("..") != std::string::npos) throw CometError("time zone '" + name + "' is not a zone name");
    for (char ch : name)
      if (!((ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z') || (ch >= '0' && ch <= '9') || ch == '_' || ch == '-' || ch == '+' || ch == '/'))
        throw CometError("time zone '" + name + "' is not a zone name");
    std::vector<std::string> dirs;
    if (const char* e = getenv("TZDIR")) dirs.push_back(e);
    dirs.insert(dirs.end(), {"/usr/share/zoneinfo", "/usr/lib/zoneinfo", "/usr/share/lib/zoneinfo", "/etc/zoneinfo"});
    std::vector<uint8_t> bytes;
    for (const std::string& d : dirs) {
      std::ifstream in(d + "/" + name, std::ios::binary);
      if (!in) continue;
      bytes.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
      if (bytes.size() >= 44) break;
      bytes.clear();
    }
    if (bytes.empty()) throw CometError("time zone '" + name + "' was not found in the time-zone database ($TZDIR, /usr/share/zoneinfo)");
    z = parse_tzif(bytes, name);
  }
  std::lock_guard<std::mutex> lk(mu);
  cache[name] = z;
  return z;
}

}  // namespace comet
