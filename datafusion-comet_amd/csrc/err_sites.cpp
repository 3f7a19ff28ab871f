// Raise sites whose Spark error names the offending value: the process-wide registry the generated kernels' site ids point into, and the
// formatting of what the device leaves in the error block (kparams.h) into the reference's error JSON — the strings
// native/common/src/error.rs:318-380 (params_as_json) builds and spark/…/ShimSparkErrorConverter.scala reads back (params("value"),
// params("precision") …: a missing key is a NoSuchElementException in the JVM instead of the Spark error).
#include <charconv>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

#include <algorithm>

#include "codegen.hpp"

namespace comet {

namespace {

std::mutex g_mu;
std::map<uint32_t, ErrSite> g_sites;
std::map<uint32_t, std::string> g_site_canon;      // id → canonical text + ordinal of the site that owns it

std::string canon(const ErrSite& s) {
  return s.error_type + "|" + s.error_class + "|" + s.from_type + "|" + s.to_type + "|" + std::to_string(s.precision) + "|" + std::to_string(s.scale) + "|" +
         std::to_string(s.value) + "|" + s.suffix;
}

std::string json_escape(const std::string& v) {
  std::string o;
  for (unsigned char ch : v) {
    switch (ch) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      default:
        if (ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", ch); o += b; }
        else o += (char)ch;
    }
  }
  return o;
}

std::string i128_str(__int128 v) {
  if (v == 0) return "0";
  const bool neg = v < 0;
  unsigned __int128 u = neg ? (unsigned __int128)0 - (unsigned __int128)v : (unsigned __int128)v;
  std::string d;
  while (u) { d += (char)('0' + (int)(u % 10)); u /= 10; }
  if (neg) d += '-';
  return std::string(d.rbegin(), d.rend());
}

// format_decimal_str (conversion_funcs/numeric.rs:564-585): the unscaled digits cut to `precision`, the point `scale` digits from the right
std::string decimal_str(__int128 unscaled, int precision, int scale) {
  std::string v = i128_str(unscaled);
  const std::string sign = v[0] == '-' ? "-" : "";
  const std::string rest = v.substr(sign.size());
  v = v.substr(0, std::min<size_t>((size_t)precision, rest.size()) + sign.size());
  if (scale == 0) return v;
  if (scale < 0) return v + std::string((size_t)-scale, '0');
  if (rest.size() > (size_t)scale) return v.substr(0, v.size() - (size_t)scale) + "." + v.substr(v.size() - (size_t)scale);
  return sign + "0." + std::string((size_t)scale - rest.size(), '0') + rest;
}

// Rust's `{:e}` (LowerExp) of a float: the shortest digits that read back, "d.ddde-x" without a plus sign or padding; inf / NaN by name
template <class F>
std::string rust_lower_exp(F x) {
  if (std::isnan(x)) return "NaN";
  if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
  if (x == 0) return std::signbit(x) ? "-0e0" : "0e0";
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::scientific);
  std::string s(buf, r.ptr);
  const size_t e = s.find('e');
  std::string mant = s.substr(0, e), ex = s.substr(e + 1);
  const bool eneg = ex[0] == '-';
  ex = ex.substr(1);
  while (ex.size() > 1 && ex[0] == '0') ex.erase(0, 1);
  return mant + "e" + (eneg ? "-" : "") + ex;
}

// Rust's Display of an f64: the shortest digits in positional notation, never an exponent
std::string rust_display(double x) {
  if (std::isnan(x)) return "NaN";
  if (std::isinf(x)) return x < 0 ? "-inf" : "inf";
  char buf[400];
  auto r = std::to_chars(buf, buf + sizeof buf, x, std::chars_format::fixed);
  return std::string(buf, r.ptr);
}

}  // namespace

uint32_t register_err_site(const ErrSite& s, int ordinal) {
  const std::string c = canon(s) + "#" + std::to_string(ordinal);
  uint32_t h = 2166136261u;
  for (unsigned char ch : c) { h ^= ch; h *= 16777619u; }
  h &= 0x7fffffffu;
  std::lock_guard<std::mutex> lk(g_mu);
  // two different sites that hash alike must not share an id (the second would be reported with the first one's class, types and value
  // kind): the later one probes to the next free id.  The id goes into the kernel text, so the text — and with it the code-object cache
  // key — follows whatever id was handed out; nothing else depends on it.
  for (;; h = (h + 1) & 0x7fffffffu) {
    auto it = g_sites.find(h);
    if (it == g_sites.end()) { g_sites.emplace(h, s); g_site_canon.emplace(h, c); return h; }
    if (g_site_canon[h] == c) return h;
  }
}

bool lookup_err_site(uint32_t id, ErrSite& out) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_sites.find(id);
  if (it == g_sites.end()) return false;
  out = it->second;
  return true;
}

// SparkErrorWithContext::to_json (error.rs:806-831): "context" with the QueryContext's fields and "summary" = QueryContext::format_summary
// (query_context.rs:104-158: "== SQL [of TYPE [NAME]] (line L, position P+1) ==", the SQL text, and carets under the fragment)
static std::string context_json(const QueryContext& c) {
  auto chars = [](const std::string& t) { size_t n = 0; for (unsigned char ch : t) n += (ch & 0xC0) != 0x80; return n; };
  auto byte_of_char = [](const std::string& t, size_t k, size_t& out) {      // char_index_to_byte_offset: None behind the last character
    size_t n = 0;
    for (size_t i = 0; i < t.size(); i++)
      if (((unsigned char)t[i] & 0xC0) != 0x80) { if (n == k) { out = i; return true; } n++; }
    return false;
  };
  const size_t start_char = (size_t)std::max(c.start_index, 0), stop_char = (size_t)std::max(c.stop_index + 1, 0);
  size_t b0 = 0, b1 = 0;
  bool ok0 = byte_of_char(c.sql_text, start_char, b0), ok1 = byte_of_char(c.sql_text, stop_char, b1);
  if (!ok1 && stop_char == chars(c.sql_text)) { b1 = c.sql_text.size(); ok1 = true; }
  const std::string fragment = (ok0 && ok1 && b1 >= b0) ? c.sql_text.substr(b0, b1 - b0) : "";
  std::string summary = "== SQL";
  if (c.has_object_type && !c.object_type.empty()) {
    summary += " of " + c.object_type;
    if (c.has_object_name && !c.object_name.empty()) summary += " " + c.object_name;
  }
  summary += " (line " + std::to_string(c.line) + ", position " + std::to_string(c.start_position + 1) + ") ==\n" + c.sql_text + "\n" +
             std::string((size_t)std::max(c.start_position, 0), ' ') + std::string(std::max<size_t>(chars(fragment), 1), '^');
  std::string j = ",\"context\":{\"sqlText\":\"" + json_escape(c.sql_text) + "\",\"startIndex\":" + std::to_string(c.start_index) + ",\"stopIndex\":" +
                  std::to_string(c.stop_index) + ",\"objectType\":" + (c.has_object_type ? "\"" + json_escape(c.object_type) + "\"" : std::string("null")) +
                  ",\"objectName\":" + (c.has_object_name ? "\"" + json_escape(c.object_name) + "\"" : std::string("null")) + ",\"line\":" + std::to_string(c.line) +
                  ",\"startPosition\":" + std::to_string(c.start_position) + "},\"summary\":\"" + json_escape(summary) + "\"";
  return j;
}

std::string err_site_json(const ErrSite& s, uint64_t lo, uint64_t hi, const uint8_t* str, size_t str_avail, const QueryContext* ctx) {
  const std::string tail = (ctx ? context_json(*ctx) : std::string()) + "}";
  if (s.value == ErrSite::FunctionName)      // DecimalSumOverflow { function_name } (error.rs:75-76, 374-377); the name travels in from_type
    return "{\"errorType\":\"" + s.error_type + "\",\"errorClass\":\"" + s.error_class + "\",\"params\":{\"functionName\":\"" + s.from_type + "\"}" + tail;
  if (s.value == ErrSite::IndexAndSize)      // InvalidArrayIndex / InvalidElementAtIndex { index_value, array_size } (error.rs:397-414)
    return "{\"errorType\":\"" + s.error_type + "\",\"errorClass\":\"" + s.error_class + "\",\"params\":{\"indexValue\":" + std::to_string((long long)(int64_t)lo) + ",\"arraySize\":" +
           std::to_string((long long)(int64_t)hi) + "}" + tail;
  if (s.value == ErrSite::NoValue)      // ArithmeticOverflow { from_type } (error.rs:369-373): which type overflowed, no value
    return "{\"errorType\":\"" + s.error_type + "\",\"errorClass\":\"" + s.error_class + "\",\"params\":{" +
           (s.from_type.empty() ? std::string() : "\"fromType\":\"" + s.from_type + "\"") + "}" + tail;
  std::string value;
  const __int128 v128 = (__int128)(((unsigned __int128)hi << 64) | lo);
  switch (s.value) {
    case ErrSite::Unscaled128: value = i128_str(v128); break;                                  // decimal_overflow_error: value.to_string() of the i128
    case ErrSite::Int64: value = std::to_string((long long)lo) + s.suffix; break;              // cast_int_to_int_macro: value.to_string() + suffix
    case ErrSite::Int64Plain: value = std::to_string((long long)lo); break;                    // cast_int_to_decimal128: v.to_string()
    case ErrSite::F64: { double d; memcpy(&d, &lo, 8); value = rust_lower_exp(d) + "D"; break; }      // "{:e}D" with e → E below
    case ErrSite::F32: { float f; uint32_t b = (uint32_t)lo; memcpy(&f, &b, 4); value = rust_lower_exp(f); break; }
    case ErrSite::F64Display: { double d; memcpy(&d, &lo, 8); value = rust_display(d); break; }      // cast_float_to_decimal128: input_value.to_string()
    case ErrSite::DecimalBD: value = decimal_str(v128, s.precision, s.scale) + "BD"; break;   // cast_decimal_to_int*: "{}BD"
    case ErrSite::NoValue: case ErrSite::FunctionName: case ErrSite::IndexAndSize: break;
    case ErrSite::F64Micros: {      // cast_float_to_timestamp (numeric.rs:111-127): format!("{:e}", micros).to_uppercase() + "D", infinities by their Java names
      double d;
      memcpy(&d, &lo, 8);
      if (std::isinf(d)) value = d < 0 ? "-Infinity" : "Infinity";
      else if (std::isnan(d)) value = "NaN";
      else { value = rust_lower_exp(d) + "D"; for (char& ch : value) if (ch == 'e') ch = 'E'; }
      break;
    }
    case ErrSite::Str: {
      const size_t n = (size_t)lo, have = std::min(n, str_avail);
      value.assign((const char*)str, have);
      if (have < n) value += "...";      // (the error block keeps the first COMET_ERR_DETAIL_STR_BYTES bytes of a longer value)
      break;
    }
  }
  if (s.value == ErrSite::F64 || s.value == ErrSite::F32)
    for (char& ch : value) if (ch == 'e') ch = 'E';
  std::string j = "{\"errorType\":\"" + s.error_type + "\",\"errorClass\":\"" + s.error_class + "\",\"params\":{\"value\":\"" + json_escape(value) + "\"";
  if (s.error_type == "NumericValueOutOfRange") j += ",\"precision\":" + std::to_string(s.precision) + ",\"scale\":" + std::to_string(s.scale);
  else j += ",\"fromType\":\"" + s.from_type + "\",\"toType\":\"" + s.to_type + "\"";
  return j + "}" + tail;
}

std::string decimal_sum_overflow_json(int kind, const QueryContext* ctx) {
  ErrSite s;      // decimal_sum_overflow_error("sum" / "avg") (spark-expr/src/lib.rs:131-135)
  s.error_type = "DecimalSumOverflow";
  s.error_class = "ARITHMETIC_OVERFLOW";
  s.from_type = kind == 0 ? "sum" : "avg";
  s.value = ErrSite::FunctionName;
  return err_site_json(s, 0, 0, nullptr, 0, ctx);
}

}  // namespace comet
