// Columnar → UnsafeRow conversion behind Native.columnarToRow{Init,Convert,Close} (native/core/src/execution/jni_api.rs:1253-1377,
// columnar_to_row.rs:866-1345).  The host arrays the JVM exports are uploaded, converted by the kernels of c2r_kernels.hip and the row
// buffer is copied back into pinned host memory that stays valid until the next convert / close — the contract of
// ColumnarToRowContext::convert (buffer pointer + per-row offsets and lengths).
#include <cstring>
#include <climits>
#include <map>
#include <memory>
#include <mutex>

#include "../../include/comet_amd.h"
#include "exec.hpp"

using namespace comet;

extern "C" {
typedef struct C2RCol {
  const void* values;
  const uint8_t* valid_bits;
  const uint8_t* data;
  int kind;
  int pad;
} C2RCol;
int comet_launch_c2r_sizes(const C2RCol* dev_cols, int ncols, int64_t n, int fixed_size, uint32_t* sizes, void* stream);
int comet_launch_c2r_write(const C2RCol* dev_cols, int ncols, int64_t n, int bitset_bytes, const int32_t* row_offsets, uint8_t* out, int32_t* lengths, void* stream);
void pq_launch_u32_scan(const uint32_t* in, int64_t n, uint64_t* tiles, int32_t* out, void* st);
}

namespace {

#define C2R_HIP(x)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) throw CometError(std::string("columnarToRow: ") + hipGetErrorString(e_));    \
  } while (0)

struct C2RContext {
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<std::unique_ptr<DevBuf>> col_bufs;
  DevBuf dev_cols, sizes, tiles, row_off, out, lengths;
  PinnedBuf host_out, host_off, host_len;
  std::string error;
  ~C2RContext() {
    if (stream) (void)hipStreamDestroy(stream);
  }
};

std::mutex g_mu;
std::map<int64_t, std::shared_ptr<C2RContext>> g_ctx;
int64_t g_next = 1;
thread_local std::string t_error;

int kind_of(const char* fmt) {
  const std::string f = fmt ? fmt : "";
  if (f == "b") return 0;
  if (f == "c") return 1;
  if (f == "s") return 2;
  if (f == "i" || f == "tdD") return 3;
  if (f == "l" || f.rfind("tsu:", 0) == 0) return 4;
  if (f == "f") return 5;
  if (f == "g") return 6;
  if (f == "u" || f == "z") return 9;
  if (f.rfind("d:", 0) == 0) {
    int p = 0, s = 0, bits = 128;
    if (sscanf(f.c_str(), "d:%d,%d,%d", &p, &s, &bits) >= 2 && bits == 128) return p <= 18 ? 7 : 8;
  }
  throw CometError("Unsupported data type for columnar to row conversion: Arrow format '" + f + "'");
}
int width_of(int kind) { return kind == 1 ? 1 : kind == 2 ? 2 : kind == 3 || kind == 5 ? 4 : kind == 4 || kind == 6 ? 8 : 16; }

// ---- nested types (struct / list / map; columnar_to_row.rs:570-830, 1602-1900) ------------------------------------------------------------
// A batch with a nested column is converted on the HOST: the input arrays are host memory handed over by the JVM, the rows go back to host
// memory, and a nested value is a variable-length tree (a nested row per struct, UnsafeArrayData per list, two of them per map) that the
// one-thread-per-row device kernel has no layout for.  The flat columns of such a batch take the same host writer.
struct HType {
  enum Kind { Bool, I8, I16, I32, I64, F32, F64, DecLong, DecWide, Bin, LargeBin, Struct, List, LargeList, Map } kind = I64;
  std::vector<HType> kids;
  bool nested() const { return kind >= Struct; }
};
HType htype_of(const ArrowSchema* s) {
  HType t;
  const std::string f = s->format ? s->format : "";
  if (s->dictionary) throw CometError("columnarToRowConvert: dictionary-encoded children of nested columns are not supported yet");
  if (f == "+s") {
    t.kind = HType::Struct;
    for (int64_t k = 0; k < s->n_children; k++) t.kids.push_back(htype_of(s->children[k]));
    return t;
  }
  if (f == "+l" || f == "+L") {
    t.kind = f == "+l" ? HType::List : HType::LargeList;
    if (s->n_children != 1) throw CometError("columnarToRowConvert: malformed list type");
    t.kids.push_back(htype_of(s->children[0]));
    return t;
  }
  if (f == "+m") {
    t.kind = HType::Map;
    if (s->n_children != 1 || s->children[0]->n_children != 2) throw CometError("columnarToRowConvert: malformed map type");
    t.kids.push_back(htype_of(s->children[0]->children[0]));
    t.kids.push_back(htype_of(s->children[0]->children[1]));
    return t;
  }
  if (f == "U" || f == "Z") { t.kind = HType::LargeBin; return t; }
  switch (kind_of(s->format)) {
    case 0: t.kind = HType::Bool; break;
    case 1: t.kind = HType::I8; break;
    case 2: t.kind = HType::I16; break;
    case 3: t.kind = HType::I32; break;
    case 4: t.kind = HType::I64; break;
    case 5: t.kind = HType::F32; break;
    case 6: t.kind = HType::F64; break;
    case 7: t.kind = HType::DecLong; break;
    case 8: t.kind = HType::DecWide; break;
    default: t.kind = HType::Bin; break;
  }
  return t;
}
bool any_nested(const HType& t) { return t.nested() || t.kind == HType::LargeBin; }

inline bool h_valid(const ArrowArray* a, int64_t i) {
  if (a->null_count == 0 || a->n_buffers < 1 || !a->buffers[0]) return true;
  const int64_t b = a->offset + i;
  return (((const uint8_t*)a->buffers[0])[b >> 3] >> (b & 7)) & 1;
}
inline size_t round8(size_t n) { return (n + 7) & ~(size_t)7; }
inline void put_u64(std::vector<uint8_t>& buf, size_t at, uint64_t v) { memcpy(buf.data() + at, &v, 8); }
inline void set_null(std::vector<uint8_t>& buf, size_t bitset_start, int64_t idx) { buf[bitset_start + (size_t)(idx >> 3)] |= (uint8_t)(1u << (idx & 7)); }

// the 8-byte slot of a fixed-width value (get_field_value :1356-1400); false for variable-length types
bool h_fixed_slot(const HType& t, const ArrowArray* a, int64_t i, uint64_t& slot) {
  const int64_t j = a->offset + i;
  switch (t.kind) {
    case HType::Bool: slot = (((const uint8_t*)a->buffers[1])[j >> 3] >> (j & 7)) & 1; return true;
    case HType::I8: slot = (uint64_t)(int64_t)((const int8_t*)a->buffers[1])[j]; return true;
    case HType::I16: slot = (uint64_t)(int64_t)((const int16_t*)a->buffers[1])[j]; return true;
    case HType::I32: slot = (uint64_t)(int64_t)((const int32_t*)a->buffers[1])[j]; return true;
    case HType::I64: case HType::F64: slot = ((const uint64_t*)a->buffers[1])[j]; return true;
    case HType::F32: slot = ((const uint32_t*)a->buffers[1])[j]; return true;
    case HType::DecLong: slot = ((const uint64_t*)a->buffers[1])[2 * j]; return true;      // the low 64 bits of the unscaled value
    default: return false;
  }
}
size_t h_write_array_range(std::vector<uint8_t>& buf, const HType& et, const ArrowArray* values, int64_t start, int64_t n);

// appends the bytes of variable-length value i of `a` (padded to 8) and returns the UNPADDED length (write_nested_variable_to_buffer :1841-1900)
size_t h_write_value(std::vector<uint8_t>& buf, const HType& t, const ArrowArray* a, int64_t i) {
  const int64_t j = a->offset + i;
  auto append_padded = [&](const uint8_t* p, size_t n) {
    const size_t at = buf.size();
    buf.resize(at + round8(n), 0);
    if (n) memcpy(buf.data() + at, p, n);
    return n;
  };
  switch (t.kind) {
    case HType::Bin: {
      const int32_t* off = (const int32_t*)a->buffers[1];
      return append_padded((const uint8_t*)a->buffers[2] + off[j], (size_t)(off[j + 1] - off[j]));
    }
    case HType::LargeBin: {
      const int64_t* off = (const int64_t*)a->buffers[1];
      return append_padded((const uint8_t*)a->buffers[2] + off[j], (size_t)(off[j + 1] - off[j]));
    }
    case HType::DecWide: {
      // minimal big-endian two's complement (i128_to_spark_decimal_bytes :1532-1558)
      const uint8_t* le = (const uint8_t*)a->buffers[1] + 16 * (size_t)j;
      uint8_t be[16];
      for (int k = 0; k < 16; k++) be[k] = le[15 - k];
      const uint8_t sign = (be[0] & 0x80) ? 0xFF : 0x00;
      int startb = 0;
      while (startb < 15 && be[startb] == sign && ((be[startb + 1] & 0x80) == (sign & 0x80))) startb++;
      return append_padded(be + startb, (size_t)(16 - startb));
    }
    case HType::Struct: {
      // a nested row: null bitset | 8-byte slots | variable part; offsets relative to the struct's start (write_struct_to_buffer :1653-1730)
      const size_t nf = t.kids.size(), bitset = ((nf + 63) / 64) * 8, start = buf.size();
      buf.resize(start + bitset + 8 * nf, 0);
      for (size_t f = 0; f < nf; f++) {
        const ArrowArray* child = a->children[f];
        if (!h_valid(child, j)) { set_null(buf, start, (int64_t)f); continue; }
        uint64_t slot = 0;
        if (!h_fixed_slot(t.kids[f], child, j, slot)) {
          const size_t len = h_write_value(buf, t.kids[f], child, j);
          if (len > 0) slot = ((uint64_t)(buf.size() - round8(len) - start) << 32) | (uint64_t)len;
        }
        put_u64(buf, start + bitset + 8 * f, slot);
      }
      return buf.size() - start;
    }
    case HType::List: {
      const int32_t* off = (const int32_t*)a->buffers[1];
      return h_write_array_range(buf, t.kids[0], a->children[0], off[j], off[j + 1] - off[j]);
    }
    case HType::LargeList: {
      const int64_t* off = (const int64_t*)a->buffers[1];
      return h_write_array_range(buf, t.kids[0], a->children[0], off[j], off[j + 1] - off[j]);
    }
    case HType::Map: {
      // 8-byte size of the key array | key array | value array (write_map_to_buffer :1788-1836)
      const int32_t* off = (const int32_t*)a->buffers[1];
      const ArrowArray* entries = a->children[0];
      const int64_t first = entries->offset + off[j], cnt = off[j + 1] - off[j];
      const size_t start = buf.size();
      buf.resize(start + 8, 0);
      const size_t ksize = h_write_array_range(buf, t.kids[0], entries->children[0], first, cnt);
      put_u64(buf, start, (uint64_t)ksize);
      h_write_array_range(buf, t.kids[1], entries->children[1], first, cnt);
      return buf.size() - start;
    }
    default: throw CometError("internal: fixed-width type in the variable-length writer");
  }
}

// UnsafeArrayData of values[start, start + n): element count | null bitset | elements at their natural width (rounded up to 8) | variable
// part, offsets relative to the array's start (write_range_to_buffer :570-616).  Primitive elements are copied whatever their validity.
size_t h_write_array_range(std::vector<uint8_t>& buf, const HType& et, const ArrowArray* values, int64_t start, int64_t n) {
  const size_t arr_start = buf.size(), bitset = (size_t)((n + 63) / 64) * 8;
  size_t esize = 8;
  switch (et.kind) {
    case HType::Bool: case HType::I8: esize = 1; break;
    case HType::I16: esize = 2; break;
    case HType::I32: case HType::F32: esize = 4; break;
    default: esize = 8; break;
  }
  buf.resize(arr_start + 8 + bitset + round8((size_t)n * esize), 0);
  put_u64(buf, arr_start, (uint64_t)n);
  const size_t bits_at = arr_start + 8, elems_at = bits_at + bitset;
  for (int64_t k = 0; k < n; k++) {
    const int64_t i = start + k;
    const bool valid = h_valid(values, i);
    if (!valid) set_null(buf, bits_at, k);
    uint64_t slot = 0;
    const bool primitive = et.kind == HType::I8 || et.kind == HType::I16 || et.kind == HType::I32 || et.kind == HType::I64 || et.kind == HType::F32 || et.kind == HType::F64;
    if (primitive) {                      // bulk-copied by the reference: the slot holds the buffer's bytes even when the element is NULL
      h_fixed_slot(et, values, i, slot);
      memcpy(buf.data() + elems_at + (size_t)k * esize, &slot, esize);
      continue;
    }
    if (!valid) continue;
    if (h_fixed_slot(et, values, i, slot)) {
      memcpy(buf.data() + elems_at + (size_t)k * esize, &slot, esize);
      continue;
    }
    const size_t len = h_write_value(buf, et, values, i);
    if (len > 0) slot = ((uint64_t)(buf.size() - round8(len) - arr_start) << 32) | (uint64_t)len;
    put_u64(buf, elems_at + (size_t)k * 8, slot);
  }
  return buf.size() - arr_start;
}

// every row of the batch, host side; returns total bytes
size_t host_rows(const std::vector<HType>& types, struct ArrowArray** arrays, int n_cols, int64_t n, std::vector<uint8_t>& buf, int32_t* offs, int32_t* lens) {
  const size_t bitset = (size_t)((n_cols + 63) / 64) * 8, fixed = bitset + 8 * (size_t)n_cols;
  buf.clear();
  for (int64_t r = 0; r < n; r++) {
    const size_t start = buf.size();
    buf.resize(start + fixed, 0);
    for (int c = 0; c < n_cols; c++) {
      const ArrowArray* a = arrays[c];
      if (!h_valid(a, r)) { set_null(buf, start, c); continue; }
      uint64_t slot = 0;
      if (!h_fixed_slot(types[(size_t)c], a, r, slot)) {
        const size_t len = h_write_value(buf, types[(size_t)c], a, r);
        if (len > 0) slot = ((uint64_t)(buf.size() - round8(len) - start) << 32) | (uint64_t)len;
      }
      put_u64(buf, start + bitset + 8 * (size_t)c, slot);
    }
    if (buf.size() > (size_t)INT32_MAX) throw CometError("columnarToRow: the rows of one batch exceed 2 GiB");
    offs[r] = (int32_t)start;
    lens[r] = (int32_t)(buf.size() - start);
  }
  offs[n] = (int32_t)buf.size();
  return buf.size();
}

}  // namespace

extern "C" {

int64_t comet_columnar_to_row_init(int32_t batch_size, int32_t device_id) {
  (void)batch_size;
  try {
    auto c = std::make_shared<C2RContext>();
    c->device = device_id;
    std::lock_guard<std::mutex> lk(g_mu);
    const int64_t h = g_next++;
    g_ctx[h] = c;
    return h;
  } catch (const std::exception& e) {
    t_error = e.what();
    return 0;
  }
}

const char* comet_columnar_to_row_error(int64_t handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_ctx.find(handle);
  if (it != g_ctx.end() && !it->second->error.empty()) return it->second->error.c_str();
  return t_error.c_str();
}

int32_t comet_columnar_to_row_convert(int64_t handle, struct ArrowArray** arrays, struct ArrowSchema** schemas, int32_t n_cols, int64_t num_rows,
                                      const uint8_t** out_buffer, const int32_t** out_offsets, const int32_t** out_lengths) {
  std::shared_ptr<C2RContext> ctx;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(handle);
    if (it != g_ctx.end()) ctx = it->second;
  }
  // ownership of the Arrow structs moves to this call (jni_api.rs:1310-1318): released on every path
  auto release_all = [&]() {
    for (int i = 0; i < n_cols; i++) {
      if (arrays && arrays[i] && arrays[i]->release) arrays[i]->release(arrays[i]);
      if (schemas && schemas[i] && schemas[i]->release) schemas[i]->release(schemas[i]);
    }
  };
  if (!ctx) {
    t_error = "Null columnar to row context";
    release_all();
    return -2;
  }
  try {
    C2RContext& c = *ctx;
    C2R_HIP(hipSetDevice(c.device));
    if (!c.stream) C2R_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    const int64_t n = num_rows;
    if (n < 0) throw CometError("columnarToRowConvert: num_rows is negative");
    {
      // a nested column (or a 64-bit-offset string column) anywhere: the whole batch takes the host writer
      bool nested = false;
      for (int i = 0; i < n_cols && !nested; i++) {
        const std::string f = schemas[i]->format ? schemas[i]->format : "";
        nested = !schemas[i]->dictionary && (f == "+s" || f == "+l" || f == "+L" || f == "+m" || f == "U" || f == "Z");
      }
      if (nested) {
        std::vector<HType> types;
        for (int i = 0; i < n_cols; i++) {
          if (schemas[i]->dictionary) throw CometError("columnarToRowConvert: a dictionary-encoded column next to a nested column is not supported yet");
          if (arrays[i]->length < n) throw CometError("columnarToRowConvert: column " + std::to_string(i) + " has fewer rows than num_rows");
          types.push_back(htype_of(schemas[i]));
        }
        c.host_off.ensure((size_t)(n + 2) * 4);
        c.host_len.ensure((size_t)std::max<int64_t>(n, 1) * 4 + 16);
        std::vector<uint8_t> rows;
        const size_t total = host_rows(types, arrays, n_cols, n, rows, (int32_t*)c.host_off.p, (int32_t*)c.host_len.p);
        c.host_out.ensure(total + 16);
        if (total) memcpy(c.host_out.p, rows.data(), total);
        *out_buffer = (const uint8_t*)c.host_out.p;
        *out_offsets = (const int32_t*)c.host_off.p;
        *out_lengths = (const int32_t*)c.host_len.p;
        release_all();
        return 0;
      }
    }
    std::vector<C2RCol> cols((size_t)n_cols);
    c.col_bufs.clear();
    auto upload = [&](const void* p, size_t bytes) -> const void* {
      auto b = std::make_unique<DevBuf>();
      b->dev = c.device;
      b->ensure(bytes + 16);
      if (bytes) C2R_HIP(hipMemcpyAsync(b->p, p, bytes, hipMemcpyHostToDevice, c.stream));
      const void* d = b->p;
      c.col_bufs.push_back(std::move(b));
      return d;
    };
    // Host-side normalisation before the upload: a sliced array (offset != 0) and a dictionary-encoded array (the reference unpacks
    // dictionaries first, columnar_to_row.rs "maybe_cast_dictionary") become plain zero-offset buffers; the temporaries live until
    // the uploads have completed (first stream synchronisation below).
    struct Plain { std::vector<uint8_t> valid, values, data; };
    std::vector<std::unique_ptr<Plain>> temps;
    auto get_bit = [](const uint8_t* bits, int64_t i) { return bits ? ((bits[i >> 3] >> (i & 7)) & 1) != 0 : true; };
    for (int i = 0; i < n_cols; i++) {
      const ArrowArray* a = arrays[i];
      if (a->length < n) throw CometError("columnarToRowConvert: column " + std::to_string(i) + " has fewer rows than num_rows");
      C2RCol& col = cols[(size_t)i];
      col.pad = 0;
      col.data = nullptr;
      const uint8_t* in_valid = (a->null_count != 0 && a->n_buffers > 0) ? (const uint8_t*)a->buffers[0] : nullptr;
      const int64_t off0 = a->offset;
      if (a->dictionary) {
        const ArrowArray* d = a->dictionary;
        if (!schemas[i]->dictionary) throw CometError("columnarToRowConvert: dictionary array without a dictionary schema");
        col.kind = kind_of(schemas[i]->dictionary->format);
        const std::string ifmt = schemas[i]->format ? schemas[i]->format : "";
        const int iw = (ifmt == "c" || ifmt == "C") ? 1 : (ifmt == "s" || ifmt == "S") ? 2 : (ifmt == "i" || ifmt == "I") ? 4 : (ifmt == "l" || ifmt == "L") ? 8 : 0;
        if (!iw) throw CometError("columnarToRowConvert: dictionary index type '" + ifmt + "' is not supported");
        const bool isigned = ifmt[0] >= 'a';
        const uint8_t* ip = (const uint8_t*)a->buffers[1];
        const uint8_t* dvalid = (d->null_count != 0 && d->n_buffers > 0) ? (const uint8_t*)d->buffers[0] : nullptr;
        auto index_at = [&](int64_t r) -> int64_t {
          const uint8_t* q = ip + (size_t)(off0 + r) * (size_t)iw;
          switch (iw) {
            case 1: return isigned ? (int64_t) * (const int8_t*)q : (int64_t)*q;
            case 2: { uint16_t v; memcpy(&v, q, 2); return isigned ? (int64_t)(int16_t)v : (int64_t)v; }
            case 4: { uint32_t v; memcpy(&v, q, 4); return isigned ? (int64_t)(int32_t)v : (int64_t)v; }
            default: { int64_t v; memcpy(&v, q, 8); return v; }
          }
        };
        auto t = std::make_unique<Plain>();
        t->valid.assign((size_t)((n + 7) / 8) + 1, 0);
        bool any_null = false;
        std::vector<int64_t> idx((size_t)n, -1);
        for (int64_t r = 0; r < n; r++) {
          bool ok = get_bit(in_valid, off0 + r);
          int64_t k = -1;
          if (ok) {
            k = index_at(r);
            if (k < 0 || k >= d->length) throw CometError("columnarToRowConvert: dictionary index out of range");
            ok = get_bit(dvalid, d->offset + k);
          }
          if (ok) { t->valid[(size_t)(r >> 3)] |= (uint8_t)(1u << (r & 7)); idx[(size_t)r] = d->offset + k; }
          else any_null = true;
        }
        if (col.kind == 9) {
          const int32_t* doff = (const int32_t*)d->buffers[1];
          const uint8_t* dbytes = (const uint8_t*)d->buffers[2];
          t->values.resize((size_t)(n + 1) * 4);
          int32_t* o = (int32_t*)t->values.data();
          int64_t pos = 0;
          for (int64_t r = 0; r < n; r++) {
            o[r] = (int32_t)pos;
            if (idx[(size_t)r] >= 0) pos += doff[idx[(size_t)r] + 1] - doff[idx[(size_t)r]];
            if (pos > INT32_MAX) throw CometError("columnarToRow: the strings of one dictionary column exceed 2 GiB");
          }
          o[n] = (int32_t)pos;
          t->data.resize((size_t)pos + 1);
          for (int64_t r = 0; r < n; r++)
            if (idx[(size_t)r] >= 0) memcpy(t->data.data() + o[r], dbytes + doff[idx[(size_t)r]], (size_t)(o[r + 1] - o[r]));
          col.values = upload(t->values.data(), (size_t)(n + 1) * 4);
          col.data = (const uint8_t*)upload(t->data.data(), (size_t)pos);
        } else if (col.kind == 0) {
          t->values.assign((size_t)((n + 7) / 8) + 1, 0);
          const uint8_t* dv = (const uint8_t*)d->buffers[1];
          for (int64_t r = 0; r < n; r++)
            if (idx[(size_t)r] >= 0 && get_bit(dv, idx[(size_t)r])) t->values[(size_t)(r >> 3)] |= (uint8_t)(1u << (r & 7));
          col.values = upload(t->values.data(), (size_t)((n + 7) / 8));
        } else {
          const size_t w = (size_t)width_of(col.kind);
          t->values.assign((size_t)n * w + 16, 0);
          const uint8_t* dv = (const uint8_t*)d->buffers[1];
          for (int64_t r = 0; r < n; r++)
            if (idx[(size_t)r] >= 0) memcpy(t->values.data() + (size_t)r * w, dv + (size_t)idx[(size_t)r] * w, w);
          col.values = upload(t->values.data(), (size_t)n * w);
        }
        col.valid_bits = any_null ? (const uint8_t*)upload(t->valid.data(), (size_t)((n + 7) / 8)) : nullptr;
        temps.push_back(std::move(t));
        continue;
      }
      col.kind = kind_of(schemas[i]->format);
      // validity and Boolean values are bit-addressed: a sliced array needs them re-packed from bit `offset`
      auto repack = [&](const uint8_t* bits) -> const uint8_t* {
        if (off0 == 0) return bits;
        auto t = std::make_unique<Plain>();
        t->valid.assign((size_t)((n + 7) / 8) + 1, 0);
        for (int64_t r = 0; r < n; r++)
          if (get_bit(bits, off0 + r)) t->valid[(size_t)(r >> 3)] |= (uint8_t)(1u << (r & 7));
        const uint8_t* q = t->valid.data();
        temps.push_back(std::move(t));
        return q;
      };
      col.valid_bits = in_valid ? (const uint8_t*)upload(repack(in_valid), (size_t)((n + 7) / 8)) : nullptr;
      if (col.kind == 9) {
        const int32_t* off = (const int32_t*)a->buffers[1] + off0;      // offsets stay absolute into the data buffer
        col.values = upload(off, (size_t)(n + 1) * 4);
        const size_t total = n > 0 ? (size_t)off[n] : 0;
        col.data = (const uint8_t*)upload(a->buffers[2], total);
      } else if (col.kind == 0) {
        col.values = upload(repack((const uint8_t*)a->buffers[1]), (size_t)((n + 7) / 8));
      } else {
        const size_t w = (size_t)width_of(col.kind);
        col.values = upload((const uint8_t*)a->buffers[1] + (size_t)off0 * w, (size_t)n * w);
      }
    }
    const int bitset_bytes = ((n_cols + 63) / 64) * 8, fixed_size = bitset_bytes + 8 * n_cols;
    c.dev_cols.dev = c.sizes.dev = c.tiles.dev = c.row_off.dev = c.out.dev = c.lengths.dev = c.device;
    c.dev_cols.ensure((size_t)std::max(n_cols, 1) * sizeof(C2RCol));
    c.sizes.ensure((size_t)std::max<int64_t>(n, 1) * 4 + 16);
    c.tiles.ensure((size_t)((n + 1023) / 1024 + 2) * 8);
    c.row_off.ensure((size_t)(n + 2) * 4);
    c.lengths.ensure((size_t)std::max<int64_t>(n, 1) * 4 + 16);
    c.host_off.ensure((size_t)(n + 2) * 4);
    c.host_len.ensure((size_t)std::max<int64_t>(n, 1) * 4 + 16);
    int64_t total = 0;
    if (n > 0) {
      C2R_HIP(hipMemcpyAsync(c.dev_cols.p, cols.data(), cols.size() * sizeof(C2RCol), hipMemcpyHostToDevice, c.stream));
      if (comet_launch_c2r_sizes((const C2RCol*)c.dev_cols.p, n_cols, n, fixed_size, (uint32_t*)c.sizes.p, c.stream) != 0) throw CometError("columnarToRow: launch failed");
      pq_launch_u32_scan((const uint32_t*)c.sizes.p, n, (uint64_t*)c.tiles.p, (int32_t*)c.row_off.p, c.stream);
      C2R_HIP(hipMemcpyAsync(c.host_off.p, c.row_off.p, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, c.stream));
      C2R_HIP(hipStreamSynchronize(c.stream));
      total = ((const int32_t*)c.host_off.p)[n];
      if (total < 0) throw CometError("columnarToRow: the rows of one batch exceed 2 GiB");
      c.out.ensure((size_t)total + 16);
      c.host_out.ensure((size_t)total + 16);
      if (comet_launch_c2r_write((const C2RCol*)c.dev_cols.p, n_cols, n, bitset_bytes, (const int32_t*)c.row_off.p, (uint8_t*)c.out.p, (int32_t*)c.lengths.p, c.stream) != 0)
        throw CometError("columnarToRow: launch failed");
      C2R_HIP(hipMemcpyAsync(c.host_out.p, c.out.p, (size_t)total, hipMemcpyDeviceToHost, c.stream));
      C2R_HIP(hipMemcpyAsync(c.host_len.p, c.lengths.p, (size_t)n * 4, hipMemcpyDeviceToHost, c.stream));
      C2R_HIP(hipStreamSynchronize(c.stream));
    } else {
      c.host_out.ensure(16);
    }
    c.col_bufs.clear();
    *out_buffer = (const uint8_t*)c.host_out.p;
    *out_offsets = (const int32_t*)c.host_off.p;
    *out_lengths = (const int32_t*)c.host_len.p;
    release_all();
    return 0;
  } catch (const std::exception& e) {
    ctx->error = e.what();
    t_error = e.what();
    release_all();
    return -2;
  }
}

void comet_columnar_to_row_close(int64_t handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_ctx.erase(handle);
}

}  // extern "C"
