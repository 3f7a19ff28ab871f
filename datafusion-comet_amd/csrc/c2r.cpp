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
