#include <chrono>
// Small host helpers shared by the execution engine: type widths, bit copies, Arrow format checks, ScanExec's casts.
#include "exec_internal.hpp"

namespace comet {

// milliseconds since the first call in this process (COMET_TRACE_STAGES lines of concurrent tasks share this clock)
double process_clock_ms() {
  static const auto epoch = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - epoch).count();
}
namespace detail {


int fixed_width(const DType& t) {
  switch (t.id) {
    case TypeId::Int8: return 1;
    case TypeId::Int16: return 2;
    case TypeId::Int32: case TypeId::Date: case TypeId::Float: return 4;
    case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: case TypeId::Double: return 8;
    case TypeId::Decimal: return 16;
    case TypeId::Bool: return 0;  // bit-packed
    default: throw CometError("Unsupported column type in GPU scan: " + t.str());
  }
}

// append n bits from src (starting at bit src_off) to dst at bit dst_off
void bit_append(uint8_t* dst, int64_t dst_off, const uint8_t* src, int64_t src_off, int64_t n) {
  if (n <= 0) return;
  if ((dst_off & 7) == 0 && (src_off & 7) == 0) {
    int64_t full = n >> 3;
    memcpy(dst + (dst_off >> 3), src + (src_off >> 3), (size_t)full);
    int64_t rem = n & 7;
    if (rem) {
      uint8_t m = (uint8_t)((1u << rem) - 1);
      uint8_t& d = dst[(dst_off >> 3) + full];
      d = (uint8_t)((d & ~m) | (src[(src_off >> 3) + full] & m));
    }
    return;
  }
  for (int64_t i = 0; i < n; i++) {
    int64_t s = src_off + i, d = dst_off + i;
    uint8_t bit = (src[s >> 3] >> (s & 7)) & 1;
    if (bit) dst[d >> 3] |= (uint8_t)(1u << (d & 7));
    else dst[d >> 3] &= (uint8_t)~(1u << (d & 7));
  }
}
void bit_fill_ones(uint8_t* dst, int64_t dst_off, int64_t n) {
  for (int64_t i = 0; i < n;) {
    int64_t d = dst_off + i;
    if ((d & 7) == 0 && n - i >= 8) {
      int64_t full = (n - i) >> 3;
      memset(dst + (d >> 3), 0xff, (size_t)full);
      i += full * 8;
    } else {
      dst[d >> 3] |= (uint8_t)(1u << (d & 7));
      i++;
    }
  }
}

// a nested input column: the stream's field (children and all) against the declared type.  Dictionary-encoded children are not unpacked.
bool nested_schema_matches(const ArrowSchema* f, const DType& t) {
  if (!f || !f->format || f->dictionary) return false;
  const std::string fmt = f->format;
  if (t.id == TypeId::Struct) {
    if (fmt != "+s" || (size_t)f->n_children != t.kids.size()) return false;
    for (size_t k = 0; k < t.kids.size(); k++)
      if (!nested_schema_matches(f->children[k], t.kids[k])) return false;
    return true;
  }
  if (t.id == TypeId::List) return fmt == "+l" && f->n_children == 1 && t.kids.size() == 1 && nested_schema_matches(f->children[0], t.kids[0]);
  if (t.id == TypeId::Map) return fmt == "+m" && f->n_children == 1 && t.kids.size() == 1 && nested_schema_matches(f->children[0], t.kids[0]);
  if (t.is_nested()) return false;
  return format_matches(f->format, t);
}

// Arrow C arrays of one nested column (the batches of a chunk) → ONE host column: rows [off, off + len) of `a` appended to `dst`, children
// and all (struct: the same rows of every field; list: offsets rebased onto what `dst` holds, the addressed elements appended).  Validity
// bitmaps are always built (all ones where a batch has none) and dropped by the caller when nothing is NULL.
void append_nested_rows(HostColumn& dst, const ArrowArray* a, const DType& t, int64_t off, int64_t len) {
  if (a->dictionary) throw CometError("dictionary-encoded fields inside a nested input column are not supported");
  const int64_t src0 = a->offset + off;
  const int64_t at = dst.length;
  dst.type = t;
  dst.validity.resize((size_t)((at + len + 7) / 8) + 8, 0);
  if (a->null_count != 0 && a->n_buffers > 0 && a->buffers[0]) {
    bit_append(dst.validity.data(), at, (const uint8_t*)a->buffers[0], src0, len);
    const uint8_t* vb = (const uint8_t*)a->buffers[0];
    for (int64_t i = 0; i < len; i++) dst.null_count += !((vb[(size_t)((src0 + i) >> 3)] >> ((src0 + i) & 7)) & 1);
  } else {
    bit_fill_ones(dst.validity.data(), at, len);
  }
  dst.length = at + len;
  if (t.id == TypeId::Struct) {
    if ((size_t)a->n_children != t.kids.size()) throw CometError("nested input column: struct array and declared type differ in their fields");
    if (dst.children.size() != t.kids.size()) dst.children.resize(t.kids.size());
    for (size_t k = 0; k < t.kids.size(); k++) append_nested_rows(dst.children[k], a->children[k], t.kids[k], src0, len);
    // Arrow leaves a field's slot under a NULL struct undefined (pyarrow writes a VALID zero there); in HBM a field's validity says NULL
    // wherever its struct is NULL — what the Parquet scan's levels give, and what GetStructField reads as the field's own validity
    std::function<void(HostColumn&, const HostColumn&)> mask = [&](HostColumn& kid, const HostColumn& parent) {
      for (int64_t i = at; i < at + len; i++)
        if (!((parent.validity[(size_t)(i >> 3)] >> (i & 7)) & 1)) kid.validity[(size_t)(i >> 3)] &= (uint8_t)~(1u << (i & 7));
      if (kid.type.id == TypeId::Struct)
        for (HostColumn& g : kid.children) mask(g, kid);
    };
    if (dst.null_count > 0)
      for (HostColumn& k : dst.children) mask(k, dst);
    return;
  }
  if (t.is_listlike()) {
    if (a->n_children != 1 || t.kids.size() != 1) throw CometError("nested input column: list array without its elements");
    if (dst.children.empty()) dst.children.resize(1);
    const int32_t* o = (const int32_t*)a->buffers[1] + src0;
    if (dst.values.empty()) dst.values.assign(4, 0);                     // offsets[0] = 0
    const int32_t base = (int32_t)dst.children[0].length;
    const size_t was = dst.values.size();
    dst.values.resize(was + (size_t)len * 4);
    int32_t* w = (int32_t*)(dst.values.data() + was);
    for (int64_t i = 0; i < len; i++) w[i] = base + (o[i + 1] - o[0]);
    if ((int64_t)base + (o[len] - o[0]) > 0x7fffffffll) throw CometError("nested input column: more than 2^31 list elements in one chunk");
    append_nested_rows(dst.children[0], a->children[0], t.kids[0], o[0], o[len] - o[0]);
    return;
  }
  if (t.id == TypeId::String || t.id == TypeId::Bytes) {
    const int32_t* o = (const int32_t*)a->buffers[1] + src0;
    if (dst.values.empty()) dst.values.assign(4, 0);
    const int32_t base = (int32_t)dst.data.size();
    const size_t was = dst.values.size();
    dst.values.resize(was + (size_t)len * 4);
    int32_t* w = (int32_t*)(dst.values.data() + was);
    for (int64_t i = 0; i < len; i++) w[i] = base + (o[i + 1] - o[0]);
    if (o[len] > o[0]) dst.data.insert(dst.data.end(), (const uint8_t*)a->buffers[2] + o[0], (const uint8_t*)a->buffers[2] + o[len]);
    return;
  }
  if (t.id == TypeId::Bool) {
    dst.values.resize((size_t)((at + len + 7) / 8) + 8, 0);
    bit_append(dst.values.data(), at, (const uint8_t*)a->buffers[1], src0, len);
    return;
  }
  const size_t w = (size_t)fixed_width(t);
  const uint8_t* src = (const uint8_t*)a->buffers[1] + (size_t)src0 * w;
  dst.values.insert(dst.values.end(), src, src + (size_t)len * w);
}

bool format_matches(const char* fmt, const DType& t) {
  if (!fmt || t.is_nested()) return false;      // (nested columns are matched field by field: nested_schema_matches)
  std::string f = fmt;
  if (t.id == TypeId::Timestamp) return f.rfind("tsu:", 0) == 0 && f.size() > 4;
  if (t.id == TypeId::Decimal) {
    std::string e = expected_format(t);
    return f == e || f == e + ",128";
  }
  return f == expected_format(t);
}

// ---- ScanExec's cast of stream columns to the declared types (operators/scan.rs:281-291 → arrow::compute::cast_with_options with the
// default CastOptions: safe, i.e. a value the target cannot hold becomes NULL) — the numeric / temporal / decimal subset the JVM side can
// produce: integer widths and signedness, float widths, int ↔ float, Date64, timestamp units, decimal precision / scale, LargeUtf8 ----

SrcFmt parse_src_format(const char* fmt) {
  SrcFmt f;
  if (!fmt) return f;
  const std::string x = fmt;
  auto intw = [&](char c) { return c == 'c' || c == 'C' ? 1 : c == 's' || c == 'S' ? 2 : c == 'i' || c == 'I' ? 4 : 8; };
  if (x.size() == 1 && strchr("csil", x[0])) { f.cls = SrcFmt::Int; f.width = intw(x[0]); }
  else if (x.size() == 1 && strchr("CSIL", x[0])) { f.cls = SrcFmt::UInt; f.width = intw(x[0]); }
  else if (x == "f") { f.cls = SrcFmt::F32; f.width = 4; }
  else if (x == "g") { f.cls = SrcFmt::F64; f.width = 8; }
  else if (x == "b") { f.cls = SrcFmt::Bool; }
  else if (x == "u" || x == "z") { f.cls = SrcFmt::Utf8; }
  else if (x == "U" || x == "Z") { f.cls = SrcFmt::LargeUtf8; }
  else if (x == "tdD") { f.cls = SrcFmt::Date32; f.width = 4; }
  else if (x == "tdm") { f.cls = SrcFmt::Date64; f.width = 8; }
  else if (x.rfind("ts", 0) == 0 && x.size() >= 4 && x[3] == ':') {
    f.cls = SrcFmt::Ts; f.width = 8;
    f.per_second = x[2] == 's' ? 1 : x[2] == 'm' ? 1000 : x[2] == 'u' ? 1000000 : x[2] == 'n' ? 1000000000 : 0;
    if (!f.per_second) f.cls = SrcFmt::Unknown;
  } else if (x.rfind("d:", 0) == 0) {
    int bits = 128;
    if (sscanf(x.c_str(), "d:%d,%d,%d", &f.p, &f.s, &bits) >= 2 && bits == 128) { f.cls = SrcFmt::Dec; f.width = 16; }
  }
  return f;
}
bool scan_cast_supported(const SrcFmt& f, const DType& t) {
  const bool tint = t.is_integer(), tflt = t.is_float();
  switch (f.cls) {
    case SrcFmt::Int: case SrcFmt::UInt: return tint || tflt || t.id == TypeId::Decimal || (f.cls == SrcFmt::Int && f.width == 4 && t.id == TypeId::Date) ||
                                                (f.cls == SrcFmt::Int && f.width == 8 && (t.id == TypeId::Timestamp || t.id == TypeId::TimestampNtz));
    case SrcFmt::F32: case SrcFmt::F64: return tint || tflt;
    case SrcFmt::Date32: return t.id == TypeId::Int32 || t.id == TypeId::Int64;
    case SrcFmt::Date64: return t.id == TypeId::Date || t.id == TypeId::Int64;
    case SrcFmt::Ts: return t.id == TypeId::Timestamp || t.id == TypeId::TimestampNtz || t.id == TypeId::Int64;
    case SrcFmt::Dec: return t.id == TypeId::Decimal || tint;
    case SrcFmt::LargeUtf8: return t.id == TypeId::String || t.id == TypeId::Bytes;
    default: return false;
  }
}
i128 cast_pow10(int e) { i128 r = 1; for (int i = 0; i < e; i++) r *= 10; return r; }
// one source value (row `i` of a column buffer) → the declared type at dst; false = NULL (safe cast)
bool scan_cast_value(const SrcFmt& f, const char* src, int64_t i, const DType& t, char* dst) {
  // read
  int64_t iv = 0; uint64_t uv = 0; double dv = 0; i128 xv = 0;
  enum { I, U, D, X } k = I;
  switch (f.cls) {
    case SrcFmt::Int: case SrcFmt::Date32: case SrcFmt::Date64: case SrcFmt::Ts:
      switch (f.width) { case 1: iv = ((const int8_t*)src)[i]; break; case 2: iv = ((const int16_t*)src)[i]; break; case 4: { int32_t v; memcpy(&v, src + i * 4, 4); iv = v; break; }
                         default: memcpy(&iv, src + i * 8, 8); }
      break;
    case SrcFmt::UInt:
      switch (f.width) { case 1: uv = ((const uint8_t*)src)[i]; break; case 2: { uint16_t v; memcpy(&v, src + i * 2, 2); uv = v; break; } case 4: { uint32_t v; memcpy(&v, src + i * 4, 4); uv = v; break; }
                         default: memcpy(&uv, src + i * 8, 8); }
      k = U;
      break;
    case SrcFmt::F32: { float v; memcpy(&v, src + i * 4, 4); dv = v; k = D; break; }
    case SrcFmt::F64: memcpy(&dv, src + i * 8, 8); k = D; break;
    case SrcFmt::Dec: memcpy(&xv, src + i * 16, 16); k = X; break;
    default: return false;
  }
  // temporal rescaling first (integers)
  if (f.cls == SrcFmt::Date64 && t.id == TypeId::Date) iv = iv / 86400000;            // arrow: ms / MILLISECONDS_IN_DAY (truncating)
  if (f.cls == SrcFmt::Ts && (t.id == TypeId::Timestamp || t.id == TypeId::TimestampNtz) && f.per_second != 1000000) {
    if (f.per_second > 1000000) iv = iv / (f.per_second / 1000000);                     // finer → µs: truncating division (arrow unary `/`)
    else if (__builtin_mul_overflow(iv, (int64_t)(1000000 / f.per_second), &iv)) return false;   // coarser → µs: checked multiply
  }
  auto store_int = [&](i128 v) -> bool {   // range-checked narrowing (num::cast): out of range → NULL
    switch (t.id) {
      case TypeId::Int8: if (v < -128 || v > 127) return false; { int8_t o = (int8_t)v; memcpy(dst, &o, 1); } return true;
      case TypeId::Int16: if (v < -32768 || v > 32767) return false; { int16_t o = (int16_t)v; memcpy(dst, &o, 2); } return true;
      case TypeId::Int32: case TypeId::Date: if (v < INT32_MIN || v > INT32_MAX) return false; { int32_t o = (int32_t)v; memcpy(dst, &o, 4); } return true;
      default: if (v < (i128)INT64_MIN || v > (i128)INT64_MAX) return false; { int64_t o = (int64_t)v; memcpy(dst, &o, 8); } return true;
    }
  };
  if (t.is_integer() || t.id == TypeId::Date || t.id == TypeId::Timestamp || t.id == TypeId::TimestampNtz) {
    if (k == I) return store_int(iv);
    if (k == U) return store_int((i128)uv);
    if (k == D) {                          // float → int: truncate toward zero; NaN / out of range → NULL
      if (!(dv == dv)) return false;
      const double tr = dv < 0 ? ceil(dv) : floor(dv);
      if (tr < -9223372036854775808.0 || tr >= 9223372036854775808.0) return false;
      return store_int((i128)(int64_t)tr);
    }
    // decimal → int: unscaled / 10^scale (truncating), then the range check
    return store_int(xv / cast_pow10(f.s));
  }
  if (t.id == TypeId::Float || t.id == TypeId::Double) {
    const double v = k == I ? (double)iv : k == U ? (double)uv : dv;
    if (t.id == TypeId::Float) { float o = k == I ? (float)iv : k == U ? (float)uv : (float)dv; memcpy(dst, &o, 4); }
    else memcpy(dst, &v, 8);
    return true;
  }
  if (t.id == TypeId::Decimal) {
    i128 v;
    const i128 bound = cast_pow10(t.precision) - 1;
    if (k == X) {
      const int up = t.scale - f.s;
      if (up >= 0) { if (__builtin_mul_overflow(xv, cast_pow10(up), &v)) return false; }
      else {
        const i128 div = cast_pow10(-up), half = div / 2, d = xv / div, r = xv % div;   // round half away from zero
        v = xv >= 0 ? (r >= half ? d + 1 : d) : (r <= -half ? d - 1 : d);
      }
    } else if (k == I || k == U) {
      if (__builtin_mul_overflow(k == I ? (i128)iv : (i128)uv, cast_pow10(t.scale), &v)) return false;
    } else return false;
    if (v > bound || v < -bound) return false;
    memcpy(dst, &v, 16);
    return true;
  }
  return false;
}

const Operator* find_scan(const Operator* op) {
  while (op && op->kind != OpKind::Scan) {
    if (op->children.empty()) return nullptr;
    op = op->children[0].get();
  }
  return op;
}

std::string validity_key(const std::vector<bool>& v) {
  std::string k;
  for (bool b : v) k.push_back(b ? '1' : '0');
  return k;
}



}  // namespace detail

}  // namespace comet
