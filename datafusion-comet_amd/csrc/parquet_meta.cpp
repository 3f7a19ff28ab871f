#include <algorithm>
#include "parquet_meta.hpp"
#include "shuffle_format.hpp"

#include <dlfcn.h>

#include <cstring>

#include "plan.hpp"

namespace comet {
namespace pq {

namespace {

// ---- Thrift compact protocol reader -------------------------------------------------------------
struct TReader {
  const uint8_t* p;
  const uint8_t* end;
  TReader(const uint8_t* d, size_t n) : p(d), end(d + n) {}
  uint8_t byte() {
    if (p >= end) throw CometError("parquet: truncated thrift data");
    return *p++;
  }
  uint64_t varint() {
    uint64_t v = 0;
    int sh = 0;
    while (true) {
      uint8_t b = byte();
      v |= (uint64_t)(b & 0x7f) << sh;
      if (!(b & 0x80)) return v;
      sh += 7;
      if (sh > 63) throw CometError("parquet: bad varint");
    }
  }
  int64_t zigzag() {
    uint64_t u = varint();
    return (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
  }
  std::string binary() {
    uint64_t n = varint();
    if ((uint64_t)(end - p) < n) throw CometError("parquet: truncated thrift binary");
    std::string s((const char*)p, (size_t)n);
    p += n;
    return s;
  }
  // returns field type (0 = STOP); updates field id
  int field(int16_t& fid) {
    uint8_t h = byte();
    if (h == 0) return 0;
    int type = h & 0x0f;
    int delta = h >> 4;
    if (delta == 0) fid = (int16_t)zigzag();
    else fid = (int16_t)(fid + delta);
    return type;
  }
  void list_header(int& elem_type, uint32_t& size) {
    uint8_t h = byte();
    elem_type = h & 0x0f;
    size = h >> 4;
    if (size == 15) size = (uint32_t)varint();
  }
  void skip(int type) {
    switch (type) {
      case 1: case 2: break;                 // bool encoded in the field header
      case 3: byte(); break;
      case 4: case 5: case 6: zigzag(); break;
      case 7: if (end - p < 8) throw CometError("parquet: truncated double"); p += 8; break;
      case 8: binary(); break;
      case 9: case 10: {
        int et;
        uint32_t n;
        list_header(et, n);
        for (uint32_t i = 0; i < n; i++) {
          if (et == 1 || et == 2) byte();     // bools in lists take a byte each
          else skip(et);
        }
        break;
      }
      case 11: {
        uint64_t n = varint();
        if (n) {
          uint8_t kv = byte();
          for (uint64_t i = 0; i < n; i++) { skip(kv >> 4); skip(kv & 0x0f); }
        }
        break;
      }
      case 12: {
        int16_t fid = 0;
        while (int t = field(fid)) skip(t);
        break;
      }
      default: throw CometError("parquet: unknown thrift type " + std::to_string(type));
    }
  }
};

SchemaElement read_schema_element(TReader& r) {
  SchemaElement e;
  int16_t fid = 0;
  while (int t = r.field(fid)) {
    switch (fid) {
      case 1: e.type = (int)r.zigzag(); break;
      case 2: e.type_length = (int)r.zigzag(); break;
      case 3: e.repetition = (int)r.zigzag(); break;
      case 4: e.name = r.binary(); break;
      case 5: e.num_children = (int)r.zigzag(); break;
      case 6: e.converted_type = (int)r.zigzag(); break;
      case 7: e.scale = (int)r.zigzag(); break;
      case 8: e.precision = (int)r.zigzag(); break;
      case 9: e.field_id = (int)r.zigzag(); break;
      case 10: {   // LogicalType union (parquet.thrift): 5 DECIMAL, 7 TIME, 8 TIMESTAMP, 10 INTEGER; the rest carry nothing the scan needs
        int16_t f2 = 0;
        while (int t2 = r.field(f2)) {
          if (t2 != 12) { r.skip(t2); continue; }
          int16_t f3 = 0;
          if (f2 == 8 || f2 == 7) {           // TimestampType / TimeType { 1: isAdjustedToUTC, 2: TimeUnit unit }
            int unit = 0;
            bool utc = true;
            while (int t3 = r.field(f3)) {
              if (f3 == 1 && (t3 == 1 || t3 == 2)) utc = t3 == 1;
              else if (f3 == 2 && t3 == 12) {   // TimeUnit union: 1 MILLIS, 2 MICROS, 3 NANOS (empty structs)
                int16_t f4 = 0;
                while (int t4 = r.field(f4)) { unit = f4; r.skip(t4); }
              } else r.skip(t3);
            }
            if (f2 == 8) { e.ts_unit = unit; e.ts_utc = utc; }
            else e.is_time = true;
          } else if (f2 == 10) {              // IntType { 1: i8 bitWidth, 2: bool isSigned }
            while (int t3 = r.field(f3)) {
              if (f3 == 1 && t3 == 3) e.int_bits = (int)(int8_t)r.byte();
              else if (f3 == 2 && (t3 == 1 || t3 == 2)) e.int_signed = t3 == 1;
              else r.skip(t3);
            }
          } else if (f2 == 5) {               // DecimalType { 1: scale, 2: precision }
            while (int t3 = r.field(f3)) {
              if (f3 == 1) e.scale = (int)r.zigzag();
              else if (f3 == 2) e.precision = (int)r.zigzag();
              else r.skip(t3);
            }
          } else {
            while (int t3 = r.field(f3)) r.skip(t3);
          }
        }
        break;
      }
      default: r.skip(t);
    }
  }
  // older writers only set converted_type (parquet.thrift ConvertedType)
  switch (e.converted_type) {
    case 7: case 8: e.is_time = true; break;                       // TIME_MILLIS / TIME_MICROS
    case 9: if (!e.ts_unit) e.ts_unit = 1; break;                  // TIMESTAMP_MILLIS
    case 10: if (!e.ts_unit) e.ts_unit = 2; break;                 // TIMESTAMP_MICROS
    case 11: case 12: case 13: case 14:                            // UINT_8 / 16 / 32 / 64
      if (!e.int_bits) { e.int_bits = 8 << (e.converted_type - 11); e.int_signed = false; }
      break;
    case 15: case 16: case 17: case 18:                            // INT_8 / 16 / 32 / 64
      if (!e.int_bits) { e.int_bits = 8 << (e.converted_type - 15); e.int_signed = true; }
      break;
    default: break;
  }
  return e;
}

ColumnMeta read_column_meta(TReader& r) {
  ColumnMeta m;
  int16_t fid = 0;
  while (int t = r.field(fid)) {
    switch (fid) {
      case 1: m.type = (int)r.zigzag(); break;
      case 2: {   // encodings used anywhere in the chunk
        int et;
        uint32_t n;
        r.list_header(et, n);
        for (uint32_t i = 0; i < n; i++) {
          const int e = (int)r.zigzag();
          if (e == DELTA_BINARY_PACKED || e == DELTA_LENGTH_BYTE_ARRAY || e == DELTA_BYTE_ARRAY) m.delta_encoded = true;
          if (e == DELTA_BYTE_ARRAY) m.prefix_encoded = true;
        }
        break;
      }
      case 3: {
        int et;
        uint32_t n;
        r.list_header(et, n);
        for (uint32_t i = 0; i < n; i++) m.path.push_back(r.binary());
        break;
      }
      case 4: m.codec = (int)r.zigzag(); break;
      case 5: m.num_values = r.zigzag(); break;
      case 6: m.total_uncompressed = r.zigzag(); break;
      case 7: m.total_compressed = r.zigzag(); break;
      case 9: m.data_page_offset = r.zigzag(); break;
      case 11: m.dictionary_page_offset = r.zigzag(); break;
      case 14: if (t == 6) m.bloom_filter_offset = r.zigzag(); else r.skip(t); break;
      case 15: if (t == 5) m.bloom_filter_length = (int32_t)r.zigzag(); else r.skip(t); break;
      case 12: {   // Statistics
        int16_t f2 = 0;
        std::string mn, mx, mn_old, mx_old;
        bool has_new = false, has_old = false;
        while (int t2 = r.field(f2)) {
          switch (f2) {
            case 1: mx_old = r.binary(); has_old = true; break;
            case 2: mn_old = r.binary(); break;
            case 3: m.null_count = r.zigzag(); break;
            case 5: mx = r.binary(); has_new = true; break;
            case 6: mn = r.binary(); break;
            default: r.skip(t2);
          }
        }
        if (has_new) { m.min_value = mn; m.max_value = mx; m.has_min_max = true; m.stats_typed_order = true; }
        else if (has_old) { m.min_value = mn_old; m.max_value = mx_old; m.has_min_max = true; }   // signed physical types only (checked by the user)
        break;
      }
      default: r.skip(t);
    }
  }
  return m;
}

RowGroup read_row_group(TReader& r) {
  RowGroup g;
  int16_t fid = 0;
  while (int t = r.field(fid)) {
    switch (fid) {
      case 1: {
        int et;
        uint32_t n;
        r.list_header(et, n);
        for (uint32_t i = 0; i < n; i++) {
          // ColumnChunk { 1 file_path, 2 file_offset, 3 meta_data, 4 offset_index_offset, 5 offset_index_length, 6 column_index_offset,
          //               7 column_index_length }
          int16_t f2 = 0;
          ColumnMeta cm;
          int64_t oio = 0, cio = 0;
          int32_t oil = 0, cil = 0;
          while (int t2 = r.field(f2)) {
            if (f2 == 3 && t2 == 12) cm = read_column_meta(r);
            else if (f2 == 4 && t2 == 6) oio = r.zigzag();
            else if (f2 == 5 && t2 == 5) oil = (int32_t)r.zigzag();
            else if (f2 == 6 && t2 == 6) cio = r.zigzag();
            else if (f2 == 7 && t2 == 5) cil = (int32_t)r.zigzag();
            else r.skip(t2);
          }
          cm.offset_index_offset = oio; cm.offset_index_length = oil;
          cm.column_index_offset = cio; cm.column_index_length = cil;
          g.columns.push_back(std::move(cm));
        }
        break;
      }
      case 2: g.total_byte_size = r.zigzag(); break;
      case 3: g.num_rows = r.zigzag(); break;
      case 6: g.total_compressed = r.zigzag(); break;
      default: r.skip(t);
    }
  }
  return g;
}

// ---- snappy (raw format) — https://github.com/google/snappy/blob/main/format_description.txt -------
// Columnar pages compress to millions of tiny elements (a PLAIN page of 8-byte decimals is ~4 output bytes per element), so the loop is
// built around fixed-size moves: while 16 bytes of slack remain on both sides, a literal of up to 16 bytes and a copy of up to 16 bytes
// from at least 8 bytes back are two 8-byte loads and stores each, whatever their length; everything else takes the careful path.
void snappy_decompress(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_len) {
  size_t i = 0;
  uint64_t ulen = 0;
  int sh = 0;
  while (true) {
    if (i >= n) throw CometError("snappy: truncated preamble");
    uint8_t b = src[i++];
    ulen |= (uint64_t)(b & 0x7f) << sh;
    if (!(b & 0x80)) break;
    sh += 7;
    if (sh > 35) throw CometError("snappy: bad preamble");
  }
  if (ulen != dst_len) throw CometError("snappy: uncompressed length mismatch");
  size_t o = 0;
  auto move16 = [](uint8_t* d, const uint8_t* s) {
    uint64_t a, b;
    memcpy(&a, s, 8);
    memcpy(&b, s + 8, 8);
    memcpy(d, &a, 8);
    memcpy(d + 8, &b, 8);
  };
  while (i < n) {
    const uint8_t tag = src[i++];
    uint32_t len, off;
    const uint32_t kind = tag & 3;
    if (kind == 0) {
      len = (tag >> 2) + 1;
      if (len <= 16 && i + 16 <= n && o + 16 <= dst_len) {
        move16(dst + o, src + i);
        i += len;
        o += len;
        continue;
      }
      if (len > 60) {
        uint32_t nb = len - 60;
        if (i + nb > n) throw CometError("snappy: truncated literal length");
        len = 0;
        for (uint32_t k = 0; k < nb; k++) len |= (uint32_t)src[i + k] << (8 * k);
        len += 1;
        i += nb;
      }
      if (i + len > n || o + len > dst_len || i + len < i) throw CometError("snappy: literal overruns buffer");
      memcpy(dst + o, src + i, len);
      i += len;
      o += len;
      continue;
    }
    if (kind == 1) {
      len = ((tag >> 2) & 7) + 4;
      if (i >= n) throw CometError("snappy: truncated copy");
      off = ((uint32_t)(tag >> 5) << 8) | src[i++];
    } else if (kind == 2) {
      len = (tag >> 2) + 1;
      if (i + 2 > n) throw CometError("snappy: truncated copy");
      off = src[i] | ((uint32_t)src[i + 1] << 8);
      i += 2;
    } else {
      len = (tag >> 2) + 1;
      if (i + 4 > n) throw CometError("snappy: truncated copy");
      off = src[i] | ((uint32_t)src[i + 1] << 8) | ((uint32_t)src[i + 2] << 16) | ((uint32_t)src[i + 3] << 24);
      i += 4;
    }
    if (off == 0 || off > o || o + len > dst_len) throw CometError("snappy: bad copy");
    if (len <= 16 && off >= 8 && o + 16 <= dst_len) {
      // two 8-byte moves IN ORDER: with 8 <= offset < 16 the second one reads bytes the first one just wrote
      uint64_t a;
      memcpy(&a, dst + o - off, 8);
      memcpy(dst + o, &a, 8);
      memcpy(&a, dst + o - off + 8, 8);
      memcpy(dst + o + 8, &a, 8);
    } else if (off >= len) {
      memcpy(dst + o, dst + o - off, len);
    } else {
      for (uint32_t k = 0; k < len; k++) dst[o + k] = dst[o + k - off];  // overlaps its own output: byte by byte
    }
    o += len;
  }
  if (o != dst_len) throw CometError("snappy: short output");
}

// The first `want` bytes of a snappy stream (fewer if the stream is shorter): what the host needs of a v1 data page whose body the device
// decompresses — the definition levels in front of the values.  Returns the bytes produced.
size_t snappy_prefix_impl(const uint8_t* src, size_t n, uint8_t* dst, size_t want) {
  size_t i = 0;
  while (true) {
    if (i >= n) throw CometError("snappy: truncated preamble");
    if (!(src[i++] & 0x80)) break;
  }
  size_t o = 0;
  while (i < n && o < want) {
    uint8_t tag = src[i++];
    uint32_t len, off;
    switch (tag & 3) {
      case 0: {
        len = (tag >> 2) + 1;
        if (len > 60) {
          uint32_t nb = len - 60;
          if (i + nb > n) throw CometError("snappy: truncated literal length");
          len = 0;
          for (uint32_t k = 0; k < nb; k++) len |= (uint32_t)src[i + k] << (8 * k);
          len += 1;
          i += nb;
        }
        if (i + len > n) throw CometError("snappy: literal overruns buffer");
        const size_t take = std::min<size_t>(len, want - o);
        memcpy(dst + o, src + i, take);
        i += len;
        o += take;
        continue;
      }
      case 1:
        len = ((tag >> 2) & 7) + 4;
        if (i >= n) throw CometError("snappy: truncated copy");
        off = ((uint32_t)(tag >> 5) << 8) | src[i++];
        break;
      case 2:
        len = (tag >> 2) + 1;
        if (i + 2 > n) throw CometError("snappy: truncated copy");
        off = src[i] | ((uint32_t)src[i + 1] << 8);
        i += 2;
        break;
      default:
        len = (tag >> 2) + 1;
        if (i + 4 > n) throw CometError("snappy: truncated copy");
        off = src[i] | ((uint32_t)src[i + 1] << 8) | ((uint32_t)src[i + 2] << 16) | ((uint32_t)src[i + 3] << 24);
        i += 4;
    }
    if (off == 0 || off > o) throw CometError("snappy: bad copy");
    const size_t take = std::min<size_t>(len, want - o);
    for (size_t k = 0; k < take; k++) dst[o + k] = dst[o + k - off];
    o += take;
  }
  return o;
}

typedef size_t (*zstd_decompress_fn)(void*, size_t, const void*, size_t);
typedef unsigned (*zstd_iserror_fn)(size_t);

}  // namespace

FileMeta parse_footer(const uint8_t* file, size_t size) {
  if (size < 12 || memcmp(file + size - 4, "PAR1", 4) != 0 || memcmp(file, "PAR1", 4) != 0)
    throw CometError("not a Parquet file (missing PAR1 magic; encrypted footers are not supported)");
  uint32_t mlen;
  memcpy(&mlen, file + size - 8, 4);
  if ((size_t)mlen + 8 > size) throw CometError("parquet: bad footer length");
  TReader r(file + size - 8 - mlen, mlen);
  FileMeta fm;
  int16_t fid = 0;
  while (int t = r.field(fid)) {
    switch (fid) {
      case 2: {
        int et;
        uint32_t n;
        r.list_header(et, n);
        for (uint32_t i = 0; i < n; i++) fm.schema.push_back(read_schema_element(r));
        break;
      }
      case 3: fm.num_rows = r.zigzag(); break;
      case 4: {
        int et;
        uint32_t n;
        r.list_header(et, n);
        for (uint32_t i = 0; i < n; i++) fm.row_groups.push_back(read_row_group(r));
        break;
      }
      default: r.skip(t);
    }
  }
  return fm;
}

PageHeader parse_page_header(const uint8_t* p, size_t avail) {
  TReader r(p, avail);
  PageHeader h;
  int16_t fid = 0;
  while (int t = r.field(fid)) {
    switch (fid) {
      case 1: h.type = (int)r.zigzag(); break;
      case 2: h.uncompressed_size = (int32_t)r.zigzag(); break;
      case 3: h.compressed_size = (int32_t)r.zigzag(); break;
      case 5: {  // DataPageHeader
        int16_t f2 = 0;
        while (int t2 = r.field(f2)) {
          switch (f2) {
            case 1: h.num_values = (int32_t)r.zigzag(); break;
            case 2: h.encoding = (int)r.zigzag(); break;
            case 3: h.def_encoding = (int)r.zigzag(); break;
            case 4: h.rep_encoding = (int)r.zigzag(); break;
            default: r.skip(t2);
          }
        }
        break;
      }
      case 7: {  // DictionaryPageHeader
        int16_t f2 = 0;
        while (int t2 = r.field(f2)) {
          switch (f2) {
            case 1: h.num_values = (int32_t)r.zigzag(); break;
            case 2: h.encoding = (int)r.zigzag(); break;
            default: r.skip(t2);
          }
        }
        break;
      }
      case 8: {  // DataPageHeaderV2
        int16_t f2 = 0;
        while (int t2 = r.field(f2)) {
          switch (f2) {
            case 1: h.num_values = (int32_t)r.zigzag(); break;
            case 2: h.num_nulls = (int32_t)r.zigzag(); break;
            case 3: h.num_rows = (int32_t)r.zigzag(); break;
            case 4: h.encoding = (int)r.zigzag(); break;
            case 5: h.def_bytes = (int32_t)r.zigzag(); break;
            case 6: h.rep_bytes = (int32_t)r.zigzag(); break;
            case 7: h.v2_compressed = (t2 == 1); break;
            default: r.skip(t2);
          }
        }
        break;
      }
      default: r.skip(t);
    }
  }
  h.header_len = (size_t)(r.p - p);
  return h;
}

size_t snappy_prefix(const uint8_t* src, size_t n, uint8_t* dst, size_t want) { return snappy_prefix_impl(src, n, dst, want); }

bool SnappyView::build(const uint8_t* src, size_t n, size_t max_elems) {
  stream = src;
  stream_len = n;
  els.clear();
  last = 0;
  size_t i = 0;
  uint64_t ulen = 0;
  for (int sh = 0;; sh += 7) {
    if (i >= n || sh > 28) return false;
    const uint8_t b = src[i++];
    ulen |= (uint64_t)(b & 0x7f) << sh;
    if (!(b & 0x80)) break;
  }
  out_len = (size_t)ulen;
  uint64_t o = 0;
  while (i < n) {
    if (els.size() >= max_elems) return false;
    const uint8_t tag = src[i++];
    El e;
    e.out_pos = (uint32_t)o;
    switch (tag & 3) {
      case 0: {
        uint32_t len = (tag >> 2) + 1;
        if (len > 60) {
          const uint32_t nb = len - 60;
          if (i + nb > n) return false;
          len = 0;
          for (uint32_t k = 0; k < nb; k++) len |= (uint32_t)src[i + k] << (8 * k);
          if (len == 0xffffffffu) return false;
          len += 1;
          i += nb;
        }
        if (i + len > n) return false;
        e.len = len; e.src = (uint32_t)i; e.copy = 0;
        i += len;
        break;
      }
      case 1:
        if (i >= n) return false;
        e.len = ((tag >> 2) & 7) + 4; e.src = ((uint32_t)(tag >> 5) << 8) | src[i]; e.copy = 1;
        i += 1;
        break;
      case 2:
        if (i + 2 > n) return false;
        e.len = (tag >> 2) + 1; e.src = (uint32_t)src[i] | ((uint32_t)src[i + 1] << 8); e.copy = 1;
        i += 2;
        break;
      default:
        if (i + 4 > n) return false;
        e.len = (tag >> 2) + 1; e.src = (uint32_t)src[i] | ((uint32_t)src[i + 1] << 8) | ((uint32_t)src[i + 2] << 16) | ((uint32_t)src[i + 3] << 24); e.copy = 1;
        i += 4;
        break;
    }
    if (e.copy && (e.src == 0 || e.src > o)) return false;
    o += e.len;
    if (o > ulen) return false;
    els.push_back(e);
  }
  return o == ulen;
}

uint8_t SnappyView::at(size_t o) const {
  for (int depth = 0; depth < 4096; depth++) {
    if (o >= out_len || els.empty()) throw CometError("snappy: read beyond the page");
    size_t k = last;
    if (k >= els.size() || els[k].out_pos > o || (size_t)els[k].out_pos + els[k].len <= o) {
      if (k + 1 < els.size() && els[k + 1].out_pos <= o && (size_t)els[k + 1].out_pos + els[k + 1].len > o) k = k + 1;     // the usual case: reading forwards
      else {
        size_t a = 0, b = els.size();
        while (a + 1 < b) { const size_t m = (a + b) / 2; if (els[m].out_pos <= o) a = m; else b = m; }
        k = a;
      }
    }
    const El& e = els[k];
    if (!e.copy) { last = k; return stream[e.src + (o - e.out_pos)]; }
    o = (size_t)e.out_pos - e.src + ((o - e.out_pos) % e.src);     // a copy: the byte `offset` back — for a copy that overlaps itself (offset < length) taken modulo the offset, so that every step lands BEFORE the element: at most one step per element
  }
  throw CometError("snappy: copy chain too deep for a sparse read");
}

// ---- Bloom filters (parquet-format BloomFilter.md; parquet.thrift BloomFilterHeader { 1 numBytes, 2 algorithm, 3 hash, 4 compression }, each of the three a union
// with ONE arm so far: BLOCK, XXHASH, UNCOMPRESSED) ------------------------------------------------------------------------------------------------------------
size_t parse_bloom_header(const uint8_t* data, size_t len, int32_t& num_bytes) {
  TReader r(data, len);
  int16_t fid = 0;
  num_bytes = -1;
  bool seen[5] = {false, false, false, false, false};
  while (int t = r.field(fid)) {
    if (fid == 1 && t == 5) { num_bytes = (int32_t)r.zigzag(); seen[1] = true; }
    else if (fid >= 2 && fid <= 4 && t == 12) {      // a union: exactly one field, whose id names the arm
      int16_t f2 = 0;
      int arms = 0;
      while (int t2 = r.field(f2)) {
        if (f2 != 1) throw CometError("parquet: a Bloom filter with an algorithm / hash / compression this reader does not know");
        arms++;
        r.skip(t2);
      }
      if (arms != 1) throw CometError("parquet: malformed Bloom filter header");
      seen[fid] = true;
    } else r.skip(t);
  }
  if (!seen[1] || !seen[2] || !seen[3] || !seen[4] || num_bytes < 32 || (num_bytes & 31)) throw CometError("parquet: malformed Bloom filter header");
  return (size_t)(r.p - data);
}

uint64_t xxh64(const void* data, size_t len, uint64_t seed) {
  constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull, P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
  auto rotl = [](uint64_t x, int r) { return (x << r) | (x >> (64 - r)); };
  auto rd64 = [](const uint8_t* q) { uint64_t v; memcpy(&v, q, 8); return v; };
  auto rd32 = [](const uint8_t* q) { uint32_t v; memcpy(&v, q, 4); return (uint64_t)v; };
  auto round = [&](uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; };
  auto merge = [&](uint64_t h, uint64_t v) { return (h ^ round(0, v)) * P1 + P4; };
  const uint8_t* q = (const uint8_t*)data;
  const uint8_t* const e = q + len;
  uint64_t h;
  if (len >= 32) {
    uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    for (; e - q >= 32; q += 32) { v1 = round(v1, rd64(q)); v2 = round(v2, rd64(q + 8)); v3 = round(v3, rd64(q + 16)); v4 = round(v4, rd64(q + 24)); }
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    h = merge(merge(merge(merge(h, v1), v2), v3), v4);
  } else {
    h = seed + P5;
  }
  h += (uint64_t)len;
  for (; e - q >= 8; q += 8) h = rotl(h ^ round(0, rd64(q)), 27) * P1 + P4;
  if (e - q >= 4) { h = rotl(h ^ (rd32(q) * P1), 23) * P2 + P3; q += 4; }
  for (; q < e; q++) h = rotl(h ^ (*q * P5), 11) * P1;
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

bool sbbf_might_contain(const uint8_t* bits, size_t nbytes, uint64_t hash) {
  static const uint32_t kSalt[8] = {0x47b6137bu, 0x44974d91u, 0x8824ad5bu, 0xa2b7289du, 0x705495c7u, 0x2df1424bu, 0x9efc4947u, 0x5c6bfb31u};
  const uint64_t nblocks = nbytes / 32;
  if (nblocks == 0) return true;
  const uint8_t* block = bits + (size_t)(((hash >> 32) * nblocks) >> 32) * 32;
  const uint32_t key = (uint32_t)hash;
  for (int i = 0; i < 8; i++) {
    uint32_t w;
    memcpy(&w, block + 4 * i, 4);
    if (!(w & (1u << ((key * kSalt[i]) >> 27)))) return false;
  }
  return true;
}

PageIndex parse_page_index(const uint8_t* column_index, size_t ci_len, const uint8_t* offset_index, size_t oi_len) {
  PageIndex pi;
  {
    // OffsetIndex { 1: list<PageLocation{1 offset, 2 compressed_page_size, 3 first_row_index}> page_locations }
    TReader r(offset_index, oi_len);
    int16_t fid = 0;
    while (int t = r.field(fid)) {
      if (fid == 1 && t == 9) {
        int et;
        uint32_t n;
        r.list_header(et, n);
        for (uint32_t i = 0; i < n; i++) {
          int16_t f2 = 0;
          int64_t first = 0;
          while (int t2 = r.field(f2)) {
            if (f2 == 3) first = r.zigzag();
            else r.skip(t2);
          }
          pi.first_row.push_back(first);
        }
      } else r.skip(t);
    }
  }
  {
    // ColumnIndex { 1: list<bool> null_pages, 2: list<binary> min_values, 3: list<binary> max_values, 4: boundary_order, 5: list<i64> null_counts }
    TReader r(column_index, ci_len);
    int16_t fid = 0;
    while (int t = r.field(fid)) {
      int et;
      uint32_t n;
      if (fid == 1 && t == 9) {
        r.list_header(et, n);
        for (uint32_t i = 0; i < n; i++) pi.null_page.push_back(r.byte() == 1);     // compact protocol: a bool in a list is one byte, 1 = true
      } else if ((fid == 2 || fid == 3) && t == 9) {
        r.list_header(et, n);
        for (uint32_t i = 0; i < n; i++) (fid == 2 ? pi.min_value : pi.max_value).push_back(r.binary());
      } else if (fid == 5 && t == 9) {
        r.list_header(et, n);
        for (uint32_t i = 0; i < n; i++) pi.null_count.push_back(r.zigzag());
      } else r.skip(t);
    }
  }
  const size_t n = pi.first_row.size();
  if (pi.null_page.size() != n || pi.min_value.size() != n || pi.max_value.size() != n || (!pi.null_count.empty() && pi.null_count.size() != n))
    throw CometError("parquet: ColumnIndex and OffsetIndex disagree on the number of pages");
  return pi;
}

void decompress(int codec, const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len) {
  switch (codec) {
    case UNCOMPRESSED:
      if (src_len != dst_len) throw CometError("parquet: uncompressed page size mismatch");
      memcpy(dst, src, dst_len);
      return;
    case SNAPPY: snappy_decompress(src, src_len, dst, dst_len); return;
    case ZSTD: {
      struct Zstd {   // resolved once; magic statics make this safe when scan threads race here
        zstd_decompress_fn fn = nullptr;
        zstd_iserror_fn iserr = nullptr;
        Zstd() {
          void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
          if (!h) return;
          fn = (zstd_decompress_fn)dlsym(h, "ZSTD_decompress");
          iserr = (zstd_iserror_fn)dlsym(h, "ZSTD_isError");
        }
      };
      static const Zstd z;
      if (!z.fn || !z.iserr) throw CometError("parquet: zstd pages need libzstd.so.1 (ZSTD_decompress), which could not be loaded");
      const zstd_decompress_fn fn = z.fn;
      const zstd_iserror_fn iserr = z.iserr;
      size_t rc = fn(dst, dst_len, src, src_len);
      if (iserr(rc) || rc != dst_len) throw CometError("parquet: zstd decompression failed");
      return;
    }
    case LZ4_RAW: {   // one raw LZ4 block per page (parquet-format Compression.md)
      const size_t got = lz4_decompress_block(src, src_len, dst, dst_len, 0);
      if (got != dst_len) throw CometError("parquet: lz4 page decompressed to " + std::to_string(got) + " bytes, expected " + std::to_string(dst_len));
      return;
    }
    case LZ4: {
      // The deprecated codec (parquet-format Compression.md "LZ4"): Hadoop's BlockCompressorStream framing — a big-endian u32 with the decompressed size
      // of a chunk, then (big-endian u32 compressed size, raw LZ4 block) until the chunk is covered; chunks repeat.  parquet-mr and parquet-cpp (pyarrow's
      // compression="lz4") write it; some older writers put one raw block there instead — like the Parquet readers, fall back to that when the framing does not add up.
      auto be32 = [](const uint8_t* q) { return ((size_t)q[0] << 24) | ((size_t)q[1] << 16) | ((size_t)q[2] << 8) | (size_t)q[3]; };
      bool framed = true;
      size_t i = 0, o = 0;
      try {
        while (framed && i < src_len) {
          if (src_len - i < 8) { framed = false; break; }
          const size_t chunk = be32(src + i);
          i += 4;
          if (chunk > dst_len - o) { framed = false; break; }
          const size_t chunk_end = o + chunk;
          while (o < chunk_end) {
            if (src_len - i < 4) { framed = false; break; }
            const size_t clen = be32(src + i);
            i += 4;
            if (clen > src_len - i) { framed = false; break; }
            const size_t got = lz4_decompress_block(src + i, clen, dst, chunk_end, o);
            if (got <= o || got > chunk_end) { framed = false; break; }
            o = got;
            i += clen;
          }
        }
      } catch (const CometError&) {
        framed = false;
      }
      if (framed && o == dst_len) return;
      size_t got = 0;
      try {
        got = lz4_decompress_block(src, src_len, dst, dst_len, 0);
      } catch (const CometError&) {
        got = (size_t)-1;
      }
      if (got != dst_len) throw CometError("parquet: an LZ4 page is neither Hadoop-framed nor one raw block of the expected size (" + std::to_string(dst_len) + " bytes)");
      return;
    }
    case GZIP: {
      // zlib through dlopen (no headers in the image: z_stream restated; the layout is part of zlib's stable ABI)
      struct ZStream {
        const uint8_t* next_in; unsigned avail_in; unsigned long total_in;
        uint8_t* next_out; unsigned avail_out; unsigned long total_out;
        const char* msg; void* state; void* zalloc; void* zfree; void* opaque;
        int data_type; unsigned long adler; unsigned long reserved;
      };
      struct Zlib {
        int (*init2)(ZStream*, int, const char*, int) = nullptr;
        int (*inflate)(ZStream*, int) = nullptr;
        int (*end)(ZStream*) = nullptr;
        Zlib() {
          void* h = dlopen("libz.so.1", RTLD_NOW | RTLD_GLOBAL);
          if (!h) return;
          init2 = (decltype(init2))dlsym(h, "inflateInit2_");
          inflate = (decltype(inflate))dlsym(h, "inflate");
          end = (decltype(end))dlsym(h, "inflateEnd");
        }
      };
      static const Zlib z;
      if (!z.init2 || !z.inflate || !z.end) throw CometError("parquet: gzip pages need libz.so.1, which could not be loaded");
      ZStream st;
      memset(&st, 0, sizeof st);
      if (z.init2(&st, 15 + 32 /* zlib or gzip header */, "1.2.11", (int)sizeof st) != 0) throw CometError("parquet: inflateInit2 failed");
      st.next_in = src;
      st.avail_in = (unsigned)src_len;
      st.next_out = dst;
      st.avail_out = (unsigned)dst_len;
      const int rc = z.inflate(&st, 4 /* Z_FINISH */);
      const unsigned long produced = st.total_out;
      z.end(&st);
      if (rc != 1 /* Z_STREAM_END */ || produced != dst_len) throw CometError("parquet: gzip decompression failed");
      return;
    }
    default: throw CometError("parquet: compression codec " + std::to_string(codec) + " is not supported (UNCOMPRESSED, SNAPPY, GZIP, ZSTD, LZ4, LZ4_RAW are)");
  }
}

// ---- DELTA_BINARY_PACKED / DELTA_LENGTH_BYTE_ARRAY / BYTE_STREAM_SPLIT → PLAIN (parquet-format Encodings.md) ----------------------------
namespace {
uint64_t delta_uleb(const uint8_t*& p, const uint8_t* end) {
  uint64_t v = 0;
  for (int shift = 0; shift < 70; shift += 7) {
    if (p >= end) throw CometError("parquet: truncated DELTA_BINARY_PACKED header");
    const uint8_t b = *p++;
    v |= (uint64_t)(b & 0x7f) << (shift < 64 ? shift : 63);
    if (!(b & 0x80)) return v;
  }
  throw CometError("parquet: malformed varint in a DELTA_BINARY_PACKED page");
}
int64_t delta_zigzag(uint64_t u) { return (int64_t)(u >> 1) ^ -(int64_t)(u & 1); }
}  // namespace

// header: block size, miniblocks per block, total value count (ULEB128), first value (zigzag); per block: min delta (zigzag), one bit
// width per miniblock, then the miniblocks — (value − previous − min delta) bit-packed LSB first, always values_per_miniblock of them
// (the last one padded); miniblocks wholly behind the last value are not stored.  Sums wrap (the spec computes in the column's width).
std::vector<int64_t> delta_binary_unpack(const uint8_t* src, size_t len, int64_t max_values, size_t* consumed) {
  const uint8_t* p = src;
  const uint8_t* end = src + len;
  const uint64_t block = delta_uleb(p, end), mini = delta_uleb(p, end), total = delta_uleb(p, end);
  const int64_t first = delta_zigzag(delta_uleb(p, end));
  if (block == 0 || block % 128 || mini == 0 || block % mini || (block / mini) % 32 || block > (1u << 20))
    throw CometError("parquet: DELTA_BINARY_PACKED block layout " + std::to_string(block) + " / " + std::to_string(mini) + " is not valid");
  if (total > (uint64_t)std::max<int64_t>(max_values, 0)) throw CometError("parquet: DELTA_BINARY_PACKED block holds " + std::to_string(total) + " values, its page only " + std::to_string(max_values));
  const uint64_t vpm = block / mini;
  std::vector<int64_t> out;
  out.reserve((size_t)total);
  if (total == 0) { if (consumed) *consumed = (size_t)(p - src); return out; }
  out.push_back(first);
  uint64_t prev = (uint64_t)first;
  while (out.size() < total) {
    const int64_t min_delta = delta_zigzag(delta_uleb(p, end));
    if ((uint64_t)(end - p) < mini) throw CometError("parquet: truncated DELTA_BINARY_PACKED block");
    const uint8_t* widths = p;
    p += mini;
    for (uint64_t m = 0; m < mini && out.size() < total; m++) {
      const int bw = widths[m];
      if (bw > 64) throw CometError("parquet: DELTA_BINARY_PACKED bit width " + std::to_string(bw));
      const size_t bytes = (size_t)(vpm * (uint64_t)bw / 8);
      if ((size_t)(end - p) < bytes) throw CometError("parquet: truncated DELTA_BINARY_PACKED miniblock");
      const uint64_t mask = bw == 64 ? ~0ull : ((1ull << bw) - 1);
      uint64_t bitpos = 0;
      for (uint64_t v = 0; v < vpm && out.size() < total; v++, bitpos += (uint64_t)bw) {
        uint64_t x = 0;
        if (bw) {
          const size_t b0 = (size_t)(bitpos >> 3);
          const int sh = (int)(bitpos & 7);
          uint8_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          memcpy(w, p + b0, std::min<size_t>(9, bytes - b0));
          uint64_t lo;
          memcpy(&lo, w, 8);
          x = sh ? (lo >> sh) | ((uint64_t)w[8] << (64 - sh)) : lo;
          x &= mask;
        }
        prev = prev + (uint64_t)min_delta + x;
        out.push_back((int64_t)prev);
      }
      p += bytes;
    }
  }
  if (consumed) *consumed = (size_t)(p - src);
  return out;
}

void delta_binary_to_plain(const uint8_t* src, size_t len, int width, int64_t max_values, std::vector<uint8_t>& out) {
  const std::vector<int64_t> v = delta_binary_unpack(src, len, max_values, nullptr);
  const size_t at = out.size();
  out.resize(at + v.size() * (size_t)width);
  if (width == 8) {
    if (!v.empty()) memcpy(out.data() + at, v.data(), v.size() * 8);
  } else if (width == 4) {
    for (size_t i = 0; i < v.size(); i++) {
      const int32_t x = (int32_t)v[i];
      memcpy(out.data() + at + i * 4, &x, 4);
    }
  } else {
    throw CometError("parquet: DELTA_BINARY_PACKED over a " + std::to_string(width) + "-byte physical type");
  }
}

void delta_length_byte_array_to_plain(const uint8_t* src, size_t len, int64_t max_values, std::vector<uint8_t>& out) {
  size_t used = 0;
  const std::vector<int64_t> lens = delta_binary_unpack(src, len, max_values, &used);
  size_t p = used;
  for (int64_t l : lens) {
    if (l < 0 || l > 0x7fffffff || (size_t)l > len - p) throw CometError("parquet: DELTA_LENGTH_BYTE_ARRAY value runs past its page");
    const uint32_t l32 = (uint32_t)l;
    const size_t at = out.size();
    out.resize(at + 4 + (size_t)l);
    memcpy(out.data() + at, &l32, 4);
    if (l) memcpy(out.data() + at + 4, src + p, (size_t)l);
    p += (size_t)l;
  }
}

void byte_stream_split_to_plain(const uint8_t* src, size_t len, int width, std::vector<uint8_t>& out) {
  if (width <= 0 || len % (size_t)width) throw CometError("parquet: BYTE_STREAM_SPLIT page size is not a multiple of the value width");
  const size_t n = len / (size_t)width, at = out.size();
  out.resize(at + len);
  uint8_t* dst = out.data() + at;
  for (int k = 0; k < width; k++) {
    const uint8_t* stream = src + (size_t)k * n;
    for (size_t i = 0; i < n; i++) dst[i * (size_t)width + (size_t)k] = stream[i];
  }
}

size_t delta_byte_array_plain_size(const uint8_t* src, size_t len, int64_t max_values) {
  size_t used = 0, used2 = 0;
  const std::vector<int64_t> prefix = delta_binary_unpack(src, len, max_values, &used);
  const std::vector<int64_t> suffix = delta_binary_unpack(src + used, len - used, max_values, &used2);
  if (prefix.size() != suffix.size()) throw CometError("parquet: DELTA_BYTE_ARRAY prefix and suffix counts differ");
  size_t total = 0;
  for (size_t i = 0; i < prefix.size(); i++) {
    if (prefix[i] < 0 || suffix[i] < 0 || prefix[i] > 0x7fffffff || suffix[i] > 0x7fffffff) throw CometError("parquet: DELTA_BYTE_ARRAY length out of range");
    total += 4 + (size_t)prefix[i] + (size_t)suffix[i];
    if (total > ((size_t)1 << 40)) throw CometError("parquet: DELTA_BYTE_ARRAY page decodes to more than 1 TiB");
  }
  return total;
}

void delta_byte_array_to_plain(const uint8_t* src, size_t len, int64_t max_values, std::vector<uint8_t>& out) {
  size_t used = 0, used2 = 0;
  const std::vector<int64_t> prefix = delta_binary_unpack(src, len, max_values, &used);
  const std::vector<int64_t> suffix = delta_binary_unpack(src + used, len - used, max_values, &used2);
  if (prefix.size() != suffix.size()) throw CometError("parquet: DELTA_BYTE_ARRAY prefix and suffix counts differ");
  size_t p = used + used2;          // the suffix bytes, back to back
  size_t prev_at = 0, prev_len = 0; // the previous value inside `out` (its bytes, behind its length word)
  for (size_t i = 0; i < prefix.size(); i++) {
    const int64_t pl = prefix[i], sl = suffix[i];
    if (pl < 0 || sl < 0 || (size_t)pl > prev_len || (size_t)sl > len - p) throw CometError("parquet: DELTA_BYTE_ARRAY value " + std::to_string(i) + " is inconsistent with its page");
    const uint32_t l32 = (uint32_t)(pl + sl);
    const size_t at = out.size();
    out.resize(at + 4 + (size_t)l32);
    memcpy(out.data() + at, &l32, 4);
    if (pl) memcpy(out.data() + at + 4, out.data() + prev_at, (size_t)pl);     // out may have moved: addressed by offset
    if (sl) memcpy(out.data() + at + 4 + (size_t)pl, src + p, (size_t)sl);
    p += (size_t)sl;
    prev_at = at + 4;
    prev_len = l32;
  }
}

}  // namespace pq
}  // namespace comet
