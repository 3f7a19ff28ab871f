// See row_shuffle.hpp.  UnsafeRow layout (spark_unsafe/row.rs:60-140, unsafe_object.rs:40-127): a null bitset of ceil(fields/64) 8-byte
// words, one 8-byte slot per field, then the variable-length region; a variable-length field's slot holds (offset << 32) | size with the
// offset counted from the row's first byte.  Decimals up to 18 digits sit in the slot as the unscaled long, wider ones as big-endian
// two's-complement bytes in the variable region (bytes_to_i128, native/common/src/utils.rs:22-39).
#include "row_shuffle.hpp"

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstring>

namespace comet {

void sort_row_partitions(int64_t* a, size_t n) {
  if (n < 2) return;
  if (n < 256) {
    std::sort(a, a + n);
    return;
  }
  // LSD radix over the bytes that actually vary; the top byte is biased so that negative values come first
  std::vector<size_t> hist(8 * 256, 0);
  for (size_t i = 0; i < n; i++) {
    const uint64_t v = (uint64_t)a[i] ^ 0x8000000000000000ull;
    for (int b = 0; b < 8; b++) hist[(size_t)b * 256 + ((v >> (8 * b)) & 0xFF)]++;
  }
  std::vector<int64_t> tmp(n);
  int64_t *src = a, *dst = tmp.data();
  for (int b = 0; b < 8; b++) {
    size_t* h = &hist[(size_t)b * 256];
    bool single = false;
    for (int k = 0; k < 256; k++)
      if (h[k] == n) single = true;
    if (single) continue;   // every record has the same byte here
    size_t at = 0;
    for (int k = 0; k < 256; k++) {
      const size_t c = h[k];
      h[k] = at;
      at += c;
    }
    const int shift = 8 * b;
    for (size_t i = 0; i < n; i++) {
      const uint64_t v = (uint64_t)src[i] ^ 0x8000000000000000ull;
      dst[h[(v >> shift) & 0xFF]++] = src[i];
    }
    std::swap(src, dst);
  }
  if (src != a) memcpy(a, src, n * sizeof(int64_t));
}

namespace {
struct Crc32Tables {
  uint32_t t[8][256];
  Crc32Tables() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
      for (int s = 1; s < 8; s++) t[s][i] = t[0][t[s - 1][i] & 0xFF] ^ (t[s - 1][i] >> 8);
  }
};
}  // namespace

uint32_t crc32_ieee(const uint8_t* p, size_t n, uint32_t init) {
  static const Crc32Tables T;
  uint32_t c = init ^ 0xFFFFFFFFu;
  while (n >= 8) {   // slicing-by-8
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = T.t[7][lo & 0xFF] ^ T.t[6][(lo >> 8) & 0xFF] ^ T.t[5][(lo >> 16) & 0xFF] ^ T.t[4][lo >> 24] ^ T.t[3][hi & 0xFF] ^ T.t[2][(hi >> 8) & 0xFF] ^
        T.t[1][(hi >> 16) & 0xFF] ^ T.t[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

uint32_t adler32(const uint8_t* p, size_t n, uint32_t init) {
  uint32_t a = init & 0xFFFF, b = init >> 16;
  while (n) {
    const size_t k = std::min<size_t>(n, 5552);   // the largest run for which b cannot overflow 32 bits
    for (size_t i = 0; i < k; i++) {
      a += p[i];
      b += a;
    }
    a %= 65521;
    b %= 65521;
    p += k;
    n -= k;
  }
  return (b << 16) | a;
}

namespace {

// one output column being assembled for a batch
struct ColumnBuild {
  DType type;
  std::vector<uint8_t> validity;   // bitmap
  std::vector<uint8_t> values;     // fixed-width values / boolean bits / int32 offsets
  std::vector<uint8_t> data;       // Utf8 / Binary bytes
  bool any_null = false;
  size_t width = 0;                // bytes per value, 0 for Bool / variable-length
};

inline bool row_is_null(const uint8_t* row, size_t idx) {
  uint64_t w;
  memcpy(&w, row + (idx >> 6) * 8, 8);
  return (w >> (idx & 63)) & 1;
}

size_t value_width(const DType& t) {
  switch (t.id) {
    case TypeId::Int8: return 1;
    case TypeId::Int16: return 2;
    case TypeId::Int32: case TypeId::Float: case TypeId::Date: return 4;
    case TypeId::Int64: case TypeId::Double: case TypeId::Timestamp: case TypeId::TimestampNtz: return 8;
    case TypeId::Decimal: return 16;
    case TypeId::Bool: case TypeId::String: case TypeId::Bytes: return 0;
    case TypeId::Struct: case TypeId::List: case TypeId::Map: return 0;      // (nested columns: NestedBuild below)
    default: throw CometError("writeSortedFileNative: unsupported column type " + t.str());
  }
}

// ---- nested columns: a struct is a nested UnsafeRow (null bitset | 8-byte slots | variable part, offsets from the struct's first byte), a list
// an UnsafeArrayData (element count | null bitset | elements at their natural width, the region rounded up to 8 | variable part, offsets from
// the array's first byte) — spark_unsafe/row.rs:140-330 + list.rs / map.rs read these; columnar_to_row.rs:570-830 writes them.  One
// NestedBuild per column and child column; values are appended row by row.
struct NestedBuild {
  DType type;
  std::vector<uint8_t> validity, values, data;
  std::vector<NestedBuild> kids;
  int64_t length = 0, nulls = 0;
  size_t width = 0;
};
void nested_init(NestedBuild& b, const DType& t) {
  b = NestedBuild();
  b.type = t;
  b.width = value_width(t);
  if (t.id == TypeId::String || t.id == TypeId::Bytes || t.is_listlike()) b.values.assign(4, 0);      // offsets[0] = 0
  if (t.id == TypeId::Struct || t.is_listlike()) {
    b.kids.resize(t.kids.size());
    for (size_t k = 0; k < t.kids.size(); k++) nested_init(b.kids[k], t.kids[k]);
  }
}
void nested_push_validity(NestedBuild& b, bool valid) {
  const int64_t i = b.length;
  if ((size_t)((i + 8) / 8) > b.validity.size()) b.validity.resize((size_t)((i + 8) / 8) + 64, 0);
  if (valid) b.validity[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
  else b.nulls++;
}
int32_t nested_last_offset(const NestedBuild& b) {
  int32_t v;
  memcpy(&v, b.values.data() + b.values.size() - 4, 4);
  return v;
}
void nested_append_null(NestedBuild& b) {
  nested_push_validity(b, false);
  switch (b.type.id) {
    case TypeId::Struct: for (auto& k : b.kids) nested_append_null(k); break;
    case TypeId::List: case TypeId::Map: case TypeId::String: case TypeId::Bytes: { const int32_t e = nested_last_offset(b); b.values.insert(b.values.end(), (const uint8_t*)&e, (const uint8_t*)&e + 4); break; }
    case TypeId::Bool: if ((size_t)((b.length + 8) / 8) > b.values.size()) b.values.resize((size_t)((b.length + 8) / 8) + 64, 0); break;
    default: b.values.insert(b.values.end(), b.width, 0);
  }
  b.length++;
}
size_t array_element_size(const DType& t) {      // UnsafeArrayData: primitive elements at their natural width, everything else an 8-byte slot
  switch (t.id) {
    case TypeId::Bool: case TypeId::Int8: return 1;
    case TypeId::Int16: return 2;
    case TypeId::Int32: case TypeId::Float: case TypeId::Date: return 4;
    default: return 8;
  }
}
// one non-NULL value: `slot` holds the value itself (a primitive: its low `width` bytes) or (offset << 32) | size of its bytes, the offset counted
// from `base` (the enclosing row / struct / array), which is `size` bytes long
void nested_append_value(NestedBuild& b, const uint8_t* base, size_t size, const uint8_t* slot, size_t slot_width, int depth = 0) {
  if (depth > 16) throw CometError("writeSortedFileNative: types nested deeper than 16 levels");
  auto var_part = [&](size_t& off, size_t& len) {
    if (slot_width < 8) throw CometError("writeSortedFileNative: a variable-length value in a slot of fewer than 8 bytes");
    uint64_t os;
    memcpy(&os, slot, 8);
    off = (size_t)(os >> 32);
    len = (size_t)(os & 0xFFFFFFFFu);
    if (off + len > size) throw CometError("writeSortedFileNative: variable-length field points outside its row");
  };
  nested_push_validity(b, true);
  switch (b.type.id) {
    case TypeId::Bool:
      if ((size_t)((b.length + 8) / 8) > b.values.size()) b.values.resize((size_t)((b.length + 8) / 8) + 64, 0);
      if (slot[0]) b.values[(size_t)(b.length >> 3)] |= (uint8_t)(1u << (b.length & 7));
      break;
    case TypeId::String: case TypeId::Bytes: {
      size_t off, len;
      var_part(off, len);
      const int64_t e = (int64_t)nested_last_offset(b) + (int64_t)len;
      if (e > INT32_MAX) throw CometError("writeSortedFileNative: more than 2 GiB of string data in one batch");
      b.data.insert(b.data.end(), base + off, base + off + len);
      const int32_t e32 = (int32_t)e;
      b.values.insert(b.values.end(), (const uint8_t*)&e32, (const uint8_t*)&e32 + 4);
      break;
    }
    case TypeId::Decimal: {
      i128 v;
      if (b.type.precision <= 18) {
        if (slot_width < 8) throw CometError("writeSortedFileNative: a decimal in a slot of fewer than 8 bytes");
        int64_t x;
        memcpy(&x, slot, 8);
        v = x;
      } else {
        size_t off, len;
        var_part(off, len);
        if (len > 16) throw CometError("writeSortedFileNative: bad wide-decimal field");
        const uint8_t* p = base + off;
        u128 u = (len && (p[0] & 0x80)) ? ~(u128)0 : 0;
        for (size_t k = 0; k < len; k++) u = (u << 8) | p[k];
        v = (i128)u;
      }
      b.values.insert(b.values.end(), (const uint8_t*)&v, (const uint8_t*)&v + 16);
      break;
    }
    case TypeId::Struct: {
      size_t off, len;
      var_part(off, len);
      const uint8_t* row = base + off;
      const size_t nf = b.kids.size(), bitset = ((nf + 63) / 64) * 8;
      if (bitset + nf * 8 > len) throw CometError("writeSortedFileNative: a nested struct is shorter than its fixed-width region");
      for (size_t k = 0; k < nf; k++) {
        if (row_is_null(row, k)) nested_append_null(b.kids[k]);
        else nested_append_value(b.kids[k], row, len, row + bitset + k * 8, 8, depth + 1);
      }
      break;
    }
    case TypeId::Map: {
      // UnsafeMapData (map.rs; written by columnar_to_row.rs:1788-1836): 8-byte size of the key array | key array | value array, both
      // UnsafeArrayData of the same element count — appended entry by entry into the entries struct's (key, value) children
      size_t off, len;
      var_part(off, len);
      NestedBuild& en = b.kids.at(0);
      if (en.kids.size() != 2) throw CometError("writeSortedFileNative: a map type without (key, value) entries");
      if (len) {
        const uint8_t* m = base + off;
        if (len < 8) throw CometError("writeSortedFileNative: a map shorter than its key-array size");
        int64_t ksz;
        memcpy(&ksz, m, 8);
        if (ksz < 8 || (uint64_t)ksz + 8 > (uint64_t)len) throw CometError("writeSortedFileNative: bad map key-array size");
        const uint8_t* ka = m + 8;
        const uint8_t* va = ka + ksz;
        const size_t vsz = len - 8 - (size_t)ksz;
        if (vsz < 8) throw CometError("writeSortedFileNative: a map without its value array");
        int64_t nk, nv;
        memcpy(&nk, ka, 8);
        memcpy(&nv, va, 8);
        if (nk < 0 || nk != nv || (uint64_t)nk > (uint64_t)len) throw CometError("writeSortedFileNative: a map whose key and value arrays differ in length");
        const size_t kes = array_element_size(en.kids[0].type), ves = array_element_size(en.kids[1].type), bitset = (((size_t)nk + 63) / 64) * 8;
        if (8 + bitset + (size_t)nk * kes > (size_t)ksz || 8 + bitset + (size_t)nk * ves > vsz) throw CometError("writeSortedFileNative: a map array is shorter than its elements");
        for (int64_t j = 0; j < nk; j++) {
          nested_push_validity(en, true);      // an entry is never NULL
          en.length++;
          if (row_is_null(ka + 8, (size_t)j)) throw CometError("writeSortedFileNative: a NULL map key");
          nested_append_value(en.kids[0], ka, (size_t)ksz, ka + 8 + bitset + (size_t)j * kes, kes, depth + 1);
          if (row_is_null(va + 8, (size_t)j)) nested_append_null(en.kids[1]);
          else nested_append_value(en.kids[1], va, vsz, va + 8 + bitset + (size_t)j * ves, ves, depth + 1);
        }
      }
      if (en.length > INT32_MAX) throw CometError("writeSortedFileNative: more than 2^31 map entries in one batch");
      const int32_t e = (int32_t)en.length;
      b.values.insert(b.values.end(), (const uint8_t*)&e, (const uint8_t*)&e + 4);
      break;
    }
    case TypeId::List: {
      size_t off, len;
      var_part(off, len);
      NestedBuild& el = b.kids.at(0);
      if (len) {      // (a zero slot: no bytes at all — read as an empty array)
        const uint8_t* arr = base + off;
        if (len < 8) throw CometError("writeSortedFileNative: an array shorter than its element count");
        int64_t n;
        memcpy(&n, arr, 8);
        const size_t esize = array_element_size(el.type);
        if (n < 0 || (uint64_t)n > (uint64_t)len) throw CometError("writeSortedFileNative: bad array element count");
        const size_t bitset = (((size_t)n + 63) / 64) * 8;
        if (8 + bitset + (size_t)n * esize > len) throw CometError("writeSortedFileNative: an array is shorter than its elements");
        const uint8_t* elems = arr + 8 + bitset;
        for (int64_t j = 0; j < n; j++) {
          if (row_is_null(arr + 8, (size_t)j)) nested_append_null(el);
          else nested_append_value(el, arr, len, elems + (size_t)j * esize, esize, depth + 1);
        }
      }
      if (el.length > INT32_MAX) throw CometError("writeSortedFileNative: more than 2^31 list elements in one batch");
      const int32_t e = (int32_t)el.length;
      b.values.insert(b.values.end(), (const uint8_t*)&e, (const uint8_t*)&e + 4);
      break;
    }
    default:
      if (slot_width < b.width) throw CometError("writeSortedFileNative: a value wider than its slot");
      b.values.insert(b.values.end(), slot, slot + b.width);      // little-endian: the low bytes are the value
  }
  b.length++;
}
void nested_slice(const NestedBuild& b, ColumnSlice& s) {
  s = ColumnSlice();
  s.type = b.type;
  s.validity = b.nulls ? b.validity.data() : nullptr;
  s.values = b.values.empty() ? nullptr : b.values.data();
  s.data = b.data.data();
  s.first = 0;
  s.kids.resize(b.kids.size());
  for (size_t k = 0; k < b.kids.size(); k++) nested_slice(b.kids[k], s.kids[k]);
}

void append_column(ColumnBuild& c, size_t idx, size_t nfields, const int64_t* addrs, const int32_t* sizes, size_t first, size_t n) {
  const size_t bitset = ((nfields + 63) / 64) * 8;
  const size_t slot_at = bitset + idx * 8;
  c.validity.assign((n + 7) / 8, 0);
  c.any_null = false;
  c.data.clear();
  const bool var = c.type.id == TypeId::String || c.type.id == TypeId::Bytes;
  if (var) c.values.assign((n + 1) * 4, 0);
  else if (c.type.id == TypeId::Bool) c.values.assign((n + 7) / 8, 0);
  else c.values.assign(n * c.width, 0);
  for (size_t r = 0; r < n; r++) {
    const uint8_t* row = (const uint8_t*)(uintptr_t)addrs[first + r];
    const size_t row_size = (size_t)sizes[first + r];
    if (slot_at + 8 > row_size) throw CometError("writeSortedFileNative: row " + std::to_string(first + r) + " is shorter than its fixed-width region");
    const bool is_null = row_is_null(row, idx);
    if (var) {
      int32_t end;
      memcpy(&end, c.values.data() + r * 4, 4);
      if (!is_null) {
        uint64_t os;
        memcpy(&os, row + slot_at, 8);
        const size_t off = (size_t)(os >> 32), len = (size_t)(os & 0xFFFFFFFFu);
        if (off + len > row_size) throw CometError("writeSortedFileNative: variable-length field points outside its row");
        if ((uint64_t)end + len > (uint64_t)INT32_MAX) throw CometError("writeSortedFileNative: more than 2 GiB of string data in one batch");
        c.data.insert(c.data.end(), row + off, row + off + len);
        end += (int32_t)len;
      }
      memcpy(c.values.data() + (r + 1) * 4, &end, 4);
    } else if (!is_null) {
      const uint8_t* slot = row + slot_at;
      switch (c.type.id) {
        case TypeId::Bool:
          if (slot[0]) c.values[r >> 3] |= (uint8_t)(1u << (r & 7));
          break;
        case TypeId::Decimal: {
          i128 v;
          if (c.type.precision <= 18) {
            int64_t x;
            memcpy(&x, slot, 8);
            v = x;
          } else {
            uint64_t os;
            memcpy(&os, slot, 8);
            const size_t off = (size_t)(os >> 32), len = (size_t)(os & 0xFFFFFFFFu);
            if (off + len > row_size || len > 16) throw CometError("writeSortedFileNative: bad wide-decimal field");
            const uint8_t* b = row + off;
            u128 u = (len && (b[0] & 0x80)) ? ~(u128)0 : 0;   // sign extension of the big-endian bytes
            for (size_t k = 0; k < len; k++) u = (u << 8) | b[k];
            v = (i128)u;
          }
          memcpy(c.values.data() + r * 16, &v, 16);
          break;
        }
        default:
          memcpy(c.values.data() + r * c.width, slot, c.width);   // little-endian slot: the low bytes are the value
      }
    }
    if (is_null) c.any_null = true;
    else c.validity[r >> 3] |= (uint8_t)(1u << (r & 7));
  }
}

void write_all(int fd, const uint8_t* p, size_t n, const std::string& path) {
  while (n) {
    const ssize_t w = ::write(fd, p, n);
    if (w < 0) throw CometError("writeSortedFileNative: write to " + path + " failed: " + strerror(errno));
    p += w;
    n -= (size_t)w;
  }
}

}  // namespace

SortedFileResult write_sorted_rows(const int64_t* row_addresses, const int32_t* row_sizes, size_t row_num, const std::vector<DType>& schema,
                                   const std::string& path, size_t batch_size, bool checksum_enabled, int checksum_algo, bool has_initial,
                                   uint32_t initial_checksum, ShuffleCodec codec, int level) {
  if (checksum_enabled && (checksum_algo < 0 || checksum_algo > 2)) throw CometError("Unsupported checksum algorithm");
  if (batch_size == 0) throw CometError("writeSortedFileNative: batch size must be positive");
  SortedFileResult res;
  res.has_checksum = checksum_enabled;
  // fresh state: CRC32 / CRC32C start at 0, Adler32 at 1 (checksum.rs:41-68)
  uint32_t sum = has_initial ? initial_checksum : (checksum_algo == 1 ? 1u : 0u);
  std::vector<ColumnBuild> cols(schema.size());
  std::vector<NestedBuild> nested(schema.size());
  for (size_t i = 0; i < schema.size(); i++) {
    cols[i].type = schema[i];
    cols[i].width = value_width(schema[i]);
  }
  const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0644);
  if (fd < 0) throw CometError("writeSortedFileNative: cannot open " + path + ": " + strerror(errno));
  try {
    std::vector<uint8_t> frozen;
    std::vector<ColumnSlice> slices(cols.size());
    for (size_t at = 0; at < row_num; at += batch_size) {
      const size_t n = std::min(batch_size, row_num - at);
      const auto t0 = std::chrono::steady_clock::now();
      for (size_t i = 0; i < cols.size(); i++) {
        if (schema[i].is_nested()) {
          // a struct / list column: its slot addresses a nested row / array inside the row; appended value by value
          nested_init(nested[i], schema[i]);
          const size_t bitset = ((cols.size() + 63) / 64) * 8, slot_at = bitset + i * 8;
          for (size_t r = 0; r < n; r++) {
            const uint8_t* row = (const uint8_t*)(uintptr_t)row_addresses[at + r];
            const size_t row_size = (size_t)row_sizes[at + r];
            if (slot_at + 8 > row_size) throw CometError("writeSortedFileNative: row " + std::to_string(at + r) + " is shorter than its fixed-width region");
            if (row_is_null(row, i)) nested_append_null(nested[i]);
            else nested_append_value(nested[i], row, row_size, row + slot_at, 8);
          }
          nested_slice(nested[i], slices[i]);
          continue;
        }
        append_column(cols[i], i, cols.size(), row_addresses, row_sizes, at, n);
        ColumnSlice& s = slices[i];
        s.type = cols[i].type;
        s.validity = cols[i].any_null ? cols[i].validity.data() : nullptr;
        s.values = cols[i].values.data();
        s.data = cols[i].data.data();
        s.first = 0;
      }
      frozen.clear();
      res.written += (int64_t)encode_shuffle_block(slices, (int64_t)n, codec, level, frozen);
      res.encode_nanos += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      if (checksum_enabled)
        sum = checksum_algo == 0 ? crc32_ieee(frozen.data(), frozen.size(), sum)
              : checksum_algo == 1 ? adler32(frozen.data(), frozen.size(), sum) : crc32c(frozen.data(), frozen.size(), sum);
      write_all(fd, frozen.data(), frozen.size(), path);
    }
  } catch (...) {
    ::close(fd);
    throw;
  }
  ::close(fd);
  res.checksum = sum;
  return res;
}

}  // namespace comet
