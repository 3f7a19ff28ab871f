// Plan inputs: host ArrowArrayStreams staged through pinned memory (casts, dictionaries, Utf8 rebasing), device streams consumed in place.
#include "exec_internal.hpp"
#include <emmintrin.h>
#include <mutex>
#include <set>
#include <tuple>

namespace comet {

// Copy into pinned staging memory with streaming (non-temporal) stores: the destination is written once and next read by the DMA engine,
// so pulling its cache lines in first (what ordinary stores do) only costs memory bandwidth — the staging copy, not PCIe, was what held
// the host-stream path at 37–40 GB/s (one read + read-for-ownership + write per byte; glibc switches to streaming stores only for copies
// far larger than the 128 KiB an 8192-row batch contributes per column).
static void stream_copy(void* dst, const void* src, size_t n) {
  char* d = (char*)dst;
  const char* s = (const char*)src;
  if (n < 4096) { memcpy(d, s, n); return; }
  const size_t head = (size_t)(-(intptr_t)d) & 15;              // up to the destination's next 16-byte boundary
  if (head) { memcpy(d, s, head); d += head; s += head; n -= head; }
  size_t blocks = n / 64;
  for (; blocks; blocks--, d += 64, s += 64) {
    const __m128i a = _mm_loadu_si128((const __m128i*)s), b = _mm_loadu_si128((const __m128i*)(s + 16)), c = _mm_loadu_si128((const __m128i*)(s + 32)),
                  e = _mm_loadu_si128((const __m128i*)(s + 48));
    _mm_stream_si128((__m128i*)d, a);
    _mm_stream_si128((__m128i*)(d + 16), b);
    _mm_stream_si128((__m128i*)(d + 32), c);
    _mm_stream_si128((__m128i*)(d + 48), e);
  }
  _mm_sfence();
  if (n & 63) memcpy(d, s, n & 63);
}

// Pull host batches from the JVM stream until a chunk is full; copy through pinned staging to HBM.
// Gather host batches of input `input` (up to max_rows rows) into one chunk resident in HBM.
// Returns false when nothing was read (stream exhausted); `rows` may be 0 with more to come only for empty batches.
// The stream's schema must be what the Scan declares (the reference casts mismatching inputs to the declared types,
// operators/scan.rs:134-164; casting is not implemented here, so a mismatch is an error instead of garbage).
void ExecutionContext::validate_input_schema(size_t input, const std::vector<DType>& types) {
  if (schema_checked_.size() <= input) schema_checked_.resize(input + 1, false);
  if (schema_checked_[input]) return;
  InputSource& in = inputs_[input];
  ArrowSchema sch;
  memset(&sch, 0, sizeof sch);
  int rc = in.kind == 0 ? in.host->get_schema(in.host, &sch) : in.dev->get_schema(in.dev, &sch);
  if (rc != 0 || !sch.release) throw CometError("input stream: get_schema failed");
  std::string err;
  if (in.kind == 1 && (size_t)sch.n_children == types.size()) {
    // Producer hint on a device-resident Utf8 field: metadata "comet:utf8_fixed_len" = L asserts that every value occupies exactly L
    // bytes (the owner of an immutable resident table measures that once — native.DeviceTable.with_string_hints — instead of every
    // task re-reading 4 B/row of offsets).  The end points are still checked per batch (pull_device_table).
    if (fixed_len_hint_.size() <= input) fixed_len_hint_.resize(input + 1);
    fixed_len_hint_[input].assign(types.size(), -1);
    for (size_t c = 0; c < types.size(); c++) {
      const char* md = sch.children[c]->metadata;
      if (!md || types[c].id != TypeId::String) continue;
      int32_t npairs;
      memcpy(&npairs, md, 4);
      const char* q = md + 4;
      for (int32_t k = 0; k < npairs && k < 4096; k++) {
        int32_t kl, vl;
        memcpy(&kl, q, 4);
        if (kl < 0 || kl > (1 << 20)) break;       // not a metadata block this reader walks any further
        const char* key = q + 4;
        memcpy(&vl, key + kl, 4);
        if (vl < 0 || vl > (1 << 24)) break;
        const char* val = key + kl + 4;
        if (kl == 20 && !memcmp(key, "comet:utf8_fixed_len", 20) && vl > 0 && vl < 4) {
          const int L = atoi(std::string(val, (size_t)vl).c_str());
          if (L >= 0 && L <= 15) fixed_len_hint_[input][c] = L;
        }
        q = val + vl;
      }
    }
  }
  if ((size_t)sch.n_children != types.size()) {
    err = "Scan declares " + std::to_string(types.size()) + " field(s) but the input stream has " + std::to_string(sch.n_children);
  } else {
    for (size_t c = 0; c < types.size() && err.empty(); c++) {
      const ArrowSchema* f = sch.children[c];
      const char* fmt = f->dictionary ? f->dictionary->format : f->format;
      if (types[c].is_nested()) {
        // a struct / list column of a host stream (a JVM-side scan's batches, a shuffle block): matched field by field, uploaded with its children
        if (in.kind == 0 && nested_schema_matches(f, types[c])) continue;
        err = "Scan input column " + std::to_string(c) + " (Arrow format '" + (fmt ? fmt : "?") + "') does not match the declared " + types[c].str() +
              (in.kind == 0 ? " field by field (nested input columns are not cast)" : ": nested columns of device-resident inputs are not supported");
        break;
      }
      if (format_matches(fmt, types[c])) continue;
      if (in.kind == 0 && !f->dictionary && scan_cast_supported(parse_src_format(fmt), types[c])) {
        if (scan_cast_from_.size() <= input) scan_cast_from_.resize(input + 1);
        scan_cast_from_[input].resize(types.size());
        scan_cast_from_[input][c] = fmt;
        continue;
      }
      err = "Scan input column " + std::to_string(c) + " has Arrow format '" + (fmt ? fmt : "?") + "' but the plan declares " + types[c].str() +
            " (this cast of a scan input is not supported by the MI355X native engine" + (in.kind == 0 ? ")" : "; device-resident inputs are never cast)");
    }
  }
  sch.release(&sch);
  if (!err.empty()) throw CometError(err);
  schema_checked_[input] = true;
}

bool ExecutionContext::pull_host_table(size_t input, const std::vector<DType>& in_types_, int64_t max_rows,
                                       std::vector<DeviceColumnView>& views, std::vector<bool>& has_valid, int64_t& rows_out) {
  InputSource& in = inputs_[input];
  rows_out = 0;
  if (in.exhausted) return false;
  validate_input_schema(input, in_types_);
  // two staging sets per input: while the GPU still reads chunk k (H2D + kernel are asynchronous) the host fills the other set
  // with chunk k+1; a set is reused only after the event recorded behind its last consumer has fired
  const size_t slot = input * 2 + (size_t)(stage_parity_ & 1);
  if (staging_.size() <= slot) staging_.resize(slot + 1);
  if (!staging_[slot]) staging_[slot].reset(new Staging());
  Staging& stg = *staging_[slot];
  if (stg.busy) {
    HIP_CHECK(hipEventSynchronize(stg.busy));
    pool_put_event(device_id_, stg.busy);
    stg.busy = nullptr;
  }
  auto& stage_vals_ = stg.stage_vals;
  auto& stage_valid_ = stg.stage_valid;
  auto& stage_aux_ = stg.stage_aux;
  auto& dev_vals_ = stg.dev_vals;
  auto& dev_valid_ = stg.dev_valid;
  auto& dev_aux_ = stg.dev_aux;
  const size_t nc = in_types_.size();
  if (stage_vals_.size() != nc) {
    stage_vals_.resize(nc);
    stage_valid_.resize(nc);
    stage_aux_.resize(nc);
    dev_vals_.resize(nc);
    dev_valid_.resize(nc);
    dev_aux_.resize(nc);
    for (size_t c = 0; c < nc; c++) {
      stage_vals_[c].reset(new PinnedBuf());
      stage_valid_[c].reset(new PinnedBuf());
      stage_aux_[c].reset(new PinnedBuf());
      dev_vals_[c].reset(new DevBuf());
      dev_valid_[c].reset(new DevBuf());
      dev_aux_[c].reset(new DevBuf());
    }
  }
  int64_t rows = 0;
  has_valid.assign(nc, false);
  std::vector<ArrowArray> held;
  // gather batches first so that staging buffers can be sized once
  while (rows < max_rows) {
    ArrowArray arr;
    memset(&arr, 0, sizeof arr);
    int rc = in.host->get_next(in.host, &arr);
    if (rc != 0) {
      const char* m = in.host->get_last_error ? in.host->get_last_error(in.host) : nullptr;
      for (auto& a : held) if (a.release) a.release(&a);
      throw CometError(std::string("input ArrowArrayStream.get_next failed: ") + (m ? m : "unknown error"));
    }
    if (!arr.release) {  // end of stream
      in.exhausted = true;
      break;
    }
    if ((size_t)arr.n_children != nc) {
      std::string msg = "input batch has " + std::to_string(arr.n_children) + " columns, Scan declares " + std::to_string(nc);
      arr.release(&arr);
      for (auto& a : held) if (a.release) a.release(&a);
      throw CometError(msg);
    }
    rows += arr.length;
    held.push_back(arr);
  }
  if (rows == 0) {
    for (auto& a : held) if (a.release) a.release(&a);
    return !in.exhausted;
  }
  for (auto& a : held)
    for (size_t c = 0; c < nc; c++)
      if (a.children[c]->null_count != 0 && a.children[c]->buffers[0]) has_valid[c] = true;
  std::vector<SrcFmt> cast_from(nc);
  if (scan_cast_from_.size() > input)
    for (size_t c = 0; c < nc && c < scan_cast_from_[input].size(); c++)
      if (!scan_cast_from_[input][c].empty()) {
        cast_from[c] = parse_src_format(scan_cast_from_[input][c].c_str());
        if (cast_from[c].cls != SrcFmt::LargeUtf8) has_valid[c] = true;   // a safe cast turns what does not fit into NULL
      }
  std::vector<size_t> aux_bytes(nc, 0);
  std::vector<int> str_uniform_(nc, -1);
  // index width of dictionary-encoded columns comes from the stream schema (fetched once per input)
  if (stg.dict_index_width.empty()) {
    stg.dict_index_width.assign(nc, 0);
    bool any_dict = false;
    for (auto& a : held)
      for (size_t c = 0; c < nc; c++) any_dict |= a.children[c]->dictionary != nullptr;
    if (any_dict) {
      ArrowSchema sch;
      memset(&sch, 0, sizeof sch);
      if (in.host->get_schema(in.host, &sch) != 0 || !sch.release) throw CometError("input stream: get_schema failed");
      for (size_t c = 0; c < nc && c < (size_t)sch.n_children; c++) {
        const ArrowSchema* f = sch.children[c];
        if (f->dictionary && f->format) {
          int w = f->format[0] == 'c' || f->format[0] == 'C' ? 1 : f->format[0] == 's' || f->format[0] == 'S' ? 2 : f->format[0] == 'i' || f->format[0] == 'I' ? 4 : 8;
          stg.dict_index_width[c] = w;
        }
      }
      sch.release(&sch);
    }
  }
  std::vector<std::shared_ptr<DevBuf>> dict_keep;
  std::vector<bool> dict_done(nc, false);
  for (size_t c = 0; c < nc; c++) {
    bool is_dict = false;
    for (auto& a : held) is_dict |= a.children[c]->dictionary != nullptr;
    if (!is_dict) continue;
    // ---- dictionary unpack on the device (K1): indices + dictionary go up, a gather kernel writes the plain column
    const DType& t = in_types_[c];
    const int iw = stg.dict_index_width[c];
    if (!iw) throw CometError("dictionary-encoded column without an index type in the stream schema");
    const bool is_str = t.id == TypeId::String || t.id == TypeId::Bytes;
    const int w = is_str ? 0 : (t.id == TypeId::Bool ? -1 : fixed_width(t));
    if (w < 0) throw CometError("dictionary-encoded boolean columns are not supported yet");
    auto vbytes = std::make_shared<DevBuf>();
    vbytes->ensure((size_t)rows + 16);
    dict_keep.push_back(vbytes);
    auto upload = [&](const void* src, size_t n) {
      auto d = std::make_shared<DevBuf>();
      d->ensure(n + 16);
      if (n) HIP_CHECK(hipMemcpy(d->p, src, n, hipMemcpyHostToDevice));
      dict_keep.push_back(d);
      return d;
    };
    struct Part { std::shared_ptr<DevBuf> idx, doffs, dbytes; int64_t at, len; };
    std::vector<Part> parts;
    auto lengths = std::make_shared<DevBuf>();
    if (is_str) lengths->ensure((size_t)rows * 4 + 16);
    else dev_vals_[c]->ensure((size_t)rows * w + 16);
    int64_t at = 0;
    bool any_null = false;
    for (auto& a : held) {
      const ArrowArray* col = a.children[c];
      const ArrowArray* dict = col->dictionary;
      if (!dict) throw CometError("a column mixes dictionary-encoded and plain batches");
      const int64_t len = col->length;
      auto d_idx = upload((const char*)col->buffers[1] + (size_t)col->offset * iw, (size_t)len * iw);
      std::shared_ptr<DevBuf> d_iv, d_dv;
      if (col->null_count != 0 && col->buffers[0]) {
        std::vector<uint8_t> bm((size_t)((len + 7) / 8) + 1, 0);
        bit_append(bm.data(), 0, (const uint8_t*)col->buffers[0], col->offset, len);
        d_iv = upload(bm.data(), bm.size());
        any_null = true;
      }
      if (dict->null_count != 0 && dict->buffers[0]) {
        std::vector<uint8_t> bm((size_t)((dict->length + 7) / 8) + 1, 0);
        bit_append(bm.data(), 0, (const uint8_t*)dict->buffers[0], dict->offset, dict->length);
        d_dv = upload(bm.data(), bm.size());
        any_null = true;
      }
      if (!is_str) {
        auto d_vals = upload((const char*)dict->buffers[1] + (size_t)dict->offset * w, (size_t)dict->length * w);
        comet_launch_dict_gather_fixed(d_idx->p, iw, d_iv ? (const uint8_t*)d_iv->p : nullptr, (const uint8_t*)d_vals->p,
                                       d_dv ? (const uint8_t*)d_dv->p : nullptr, w, len, (uint8_t*)dev_vals_[c]->p + (size_t)at * w,
                                       (uint8_t*)vbytes->p + at, stream_);
      } else {
        const int32_t* off = (const int32_t*)dict->buffers[1] + dict->offset;
        std::vector<int32_t> ro((size_t)dict->length + 1);
        for (int64_t k = 0; k <= dict->length; k++) ro[(size_t)k] = off[k] - off[0];
        auto d_off = upload(ro.data(), ro.size() * 4);
        auto d_bytes = upload((const char*)dict->buffers[2] + off[0], (size_t)ro[(size_t)dict->length]);
        comet_launch_dict_gather_str_len(d_idx->p, iw, d_iv ? (const uint8_t*)d_iv->p : nullptr, (const int32_t*)d_off->p,
                                         d_dv ? (const uint8_t*)d_dv->p : nullptr, len, (uint32_t*)lengths->p + at, (uint8_t*)vbytes->p + at, stream_);
        parts.push_back({d_idx, d_off, d_bytes, at, len});
      }
      at += len;
    }
    if (is_str) {
      auto tiles = std::make_shared<DevBuf>();
      tiles->ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
      dict_keep.push_back(tiles);
      dict_keep.push_back(lengths);
      dev_vals_[c]->ensure((size_t)(rows + 1) * 4 + 16);
      pq_launch_u32_scan((const uint32_t*)lengths->p, rows, (uint64_t*)tiles->p, (int32_t*)dev_vals_[c]->p, stream_);
      int32_t total = 0;
      read_small(&total, (char*)dev_vals_[c]->p + (size_t)rows * 4, 4);
      dev_aux_[c]->ensure((size_t)std::max(total, 1) + 16);
      for (auto& pt : parts)
        comet_launch_dict_gather_str_copy(pt.idx->p, iw, (const uint8_t*)vbytes->p + pt.at, (const int32_t*)pt.doffs->p, (const uint8_t*)pt.dbytes->p, pt.len,
                                          (const int32_t*)dev_vals_[c]->p + pt.at, (uint8_t*)dev_aux_[c]->p, stream_);
    }
    if (any_null) {
      has_valid[c] = true;
      dev_valid_[c]->ensure((size_t)((rows + 7) / 8) + 16);
      pq_launch_pack((const uint8_t*)vbytes->p, (uint8_t*)dev_valid_[c]->p, rows, stream_);
    } else {
      has_valid[c] = false;
    }
    dict_done[c] = true;
  }
  if (!dict_keep.empty()) HIP_CHECK(hipStreamSynchronize(stream_));   // uploaded indices/dictionaries are released below
  dict_keep.clear();
  // Nested columns: the chunk's batches are concatenated on the host (offsets rebased, children appended) and go up buffer by buffer; the
  // device buffers live with this staging set.  Not the streaming-store path of the flat columns: nested inputs are not the hot path.
  stg.nested_keep.clear();
  std::vector<DeviceColumnView> nested_views(nc);
  std::function<DeviceColumnView(const HostColumn&, bool&)> upload_nested = [&](const HostColumn& h, bool& hv) -> DeviceColumnView {
    DeviceColumnView v;
    auto up = [&](const void* src, size_t n) -> const void* {
      auto d = std::make_shared<DevBuf>();
      d->ensure(n + 16);
      if (n) HIP_CHECK(hipMemcpyAsync(d->p, src, n, hipMemcpyHostToDevice, stream_));
      stg.nested_keep.push_back(d);
      return d->p;
    };
    // (NULLs are counted here: a struct's fields were masked by their struct's validity after their own counts were taken)
    int64_t nulls = 0;
    for (int64_t i = 0; i < h.length && !h.validity.empty(); i++) nulls += !((h.validity[(size_t)(i >> 3)] >> (i & 7)) & 1);
    hv = nulls > 0;
    if (hv) v.valid = (const uint8_t*)up(h.validity.data(), (size_t)((h.length + 7) / 8));
    if (h.type.id == TypeId::Struct) {
      for (const HostColumn& k : h.children) {
        bool khv = false;
        v.kids.push_back(upload_nested(k, khv));
        v.kid_has_valid.push_back(khv ? 1 : 0);
      }
      v.kid_rows = h.length;
    } else if (h.type.is_listlike()) {
      static const int32_t zero = 0;
      v.data = h.values.empty() ? up(&zero, 4) : up(h.values.data(), (size_t)(h.length + 1) * 4);
      bool khv = false;
      v.kids.push_back(upload_nested(h.children.at(0), khv));
      v.kid_has_valid.push_back(khv ? 1 : 0);
      v.kid_rows = h.children[0].length;
    } else if (h.type.id == TypeId::String || h.type.id == TypeId::Bytes) {
      static const int32_t zero = 0;
      v.data = h.values.empty() ? up(&zero, 4) : up(h.values.data(), (size_t)(h.length + 1) * 4);
      v.aux = up(h.data.data(), h.data.size());
    } else if (h.type.id == TypeId::Bool) {
      v.data = up(h.values.data(), (size_t)((h.length + 7) / 8));
    } else {
      v.data = up(h.values.data(), (size_t)h.length * (size_t)fixed_width(h.type));
    }
    return v;
  };
  std::vector<HostColumn> nested_host;      // (alive until the copies queued above have run: synchronised below)
  nested_host.reserve(nc);
  for (size_t c = 0; c < nc; c++) {
    if (!in_types_[c].is_nested()) continue;
    nested_host.emplace_back();
    HostColumn& h = nested_host.back();
    h.type = in_types_[c];
    // (an empty struct / list still needs its children's shape)
    std::function<void(HostColumn&, const DType&)> shape = [&](HostColumn& x, const DType& t) {
      x.type = t;
      if (t.id == TypeId::Struct) { x.children.resize(t.kids.size()); for (size_t k = 0; k < t.kids.size(); k++) shape(x.children[k], t.kids[k]); }
      else if (t.is_listlike()) { x.children.resize(1); shape(x.children[0], t.kids.at(0)); }
    };
    shape(h, in_types_[c]);
    for (auto& a : held) {
      if (a.children[c]->length != a.length) throw CometError("ragged input batch");
      append_nested_rows(h, a.children[c], in_types_[c], 0, a.children[c]->length);
    }
    bool hv = false;
    nested_views[c] = upload_nested(h, hv);
    has_valid[c] = hv;
    dict_done[c] = true;
  }
  if (!nested_host.empty()) HIP_CHECK(hipStreamSynchronize(stream_));
  nested_host.clear();
  for (size_t c = 0; c < nc; c++) {
    if (dict_done[c]) continue;
    const DType& t = in_types_[c];
    if (t.id == TypeId::String || t.id == TypeId::Bytes) {
      // Utf8: int32 offsets rebased to the chunk + concatenated bytes
      size_t total_bytes = 0;
      const bool large = cast_from[c].cls == SrcFmt::LargeUtf8;     // LargeUtf8 / LargeBinary: int64 offsets, cast to the declared Utf8
      auto off_at = [large](const ArrowArray* col, int64_t i) -> int64_t {
        return large ? ((const int64_t*)col->buffers[1])[col->offset + i] : (int64_t)((const int32_t*)col->buffers[1])[col->offset + i];
      };
      for (auto& a : held) {
        const ArrowArray* col = a.children[c];
        total_bytes += (size_t)(off_at(col, col->length) - off_at(col, 0));
      }
      if (total_bytes > 0x7fffffffull) throw CometError("Utf8 chunk exceeds 2 GiB of string bytes; lower spark.comet.gpu.chunkRows");
      stage_vals_[c]->ensure((size_t)(rows + 1) * 4 + 16);
      stage_aux_[c]->ensure(total_bytes + 16);
      if (has_valid[c]) stage_valid_[c]->ensure((size_t)((rows + 7) / 8) + 16);
      int32_t* so = (int32_t*)stage_vals_[c]->p;
      // one job per input batch: where its rows and bytes land is a running sum over the batches; rebasing the offsets, copying the
      // bytes and noticing whether all values share one length are independent per batch and spread over the scan threads (a single
      // thread walking 4 M offsets per chunk was what held the Utf8 columns of the host path below the PCIe rate)
      struct StrJob { const ArrowArray* col; int64_t at; int32_t pos; int uniform; };
      std::vector<StrJob> sjobs;
      int64_t at = 0;
      int32_t pos = 0;
      for (auto& a : held) {
        const ArrowArray* col = a.children[c];
        if (col->dictionary) throw CometError("dictionary-encoded input columns are not unpacked on the GPU path yet");
        sjobs.push_back({col, at, pos, -2});
        pos += (int32_t)(off_at(col, col->length) - off_at(col, 0));
        at += col->length;
      }
      auto run_job = [&](StrJob& j) {
        const ArrowArray* col = j.col;
        const int64_t base = off_at(col, 0);
        int uniform = -2;   // -2 no value seen yet, -1 lengths differ, else the common length
        int32_t* dst = so + j.at;
        for (int64_t i = 0; i < col->length; i++) {
          const int64_t o = off_at(col, i);
          dst[i] = j.pos + (int32_t)(o - base);
          const int len = (int)(off_at(col, i + 1) - o);
          if (uniform == -2) uniform = len;
          else if (uniform != len) uniform = -1;
        }
        j.uniform = uniform;
        const size_t nb = (size_t)(off_at(col, col->length) - base);
        if (nb) stream_copy((char*)stage_aux_[c]->p + j.pos, (const char*)col->buffers[2] + base, nb);
      };
      if (rows >= (1 << 20) && sjobs.size() > 1) {
        const size_t parts = std::min<size_t>(16, sjobs.size());
        scan_pool_parallel(parts, [&](size_t pidx) {
          for (size_t k = pidx; k < sjobs.size(); k += parts) run_job(sjobs[k]);
        });
      } else {
        for (auto& j : sjobs) run_job(j);
      }
      int uniform = -2;
      for (auto& j : sjobs) {
        if (j.col->length == 0) continue;
        if (uniform == -2) uniform = j.uniform;
        else if (uniform != j.uniform) uniform = -1;
      }
      if (has_valid[c])       // bitmaps are small and batches need not start on a byte boundary: appended in order on this thread
        for (auto& j : sjobs) {
          if (j.col->null_count != 0 && j.col->buffers[0]) bit_append((uint8_t*)stage_valid_[c]->p, j.at, (const uint8_t*)j.col->buffers[0], j.col->offset, j.col->length);
          else bit_fill_ones((uint8_t*)stage_valid_[c]->p, j.at, j.col->length);
        }
      so[rows] = pos;
      dev_vals_[c]->ensure((size_t)(rows + 1) * 4 + 16);
      dev_aux_[c]->ensure(total_bytes + 16);
      HIP_CHECK(hipMemcpyAsync(dev_vals_[c]->p, stage_vals_[c]->p, (size_t)(rows + 1) * 4, hipMemcpyHostToDevice, stream_));
      if (total_bytes) HIP_CHECK(hipMemcpyAsync(dev_aux_[c]->p, stage_aux_[c]->p, total_bytes, hipMemcpyHostToDevice, stream_));
      if (has_valid[c]) {
        size_t kb = (size_t)((rows + 7) / 8);
        dev_valid_[c]->ensure(kb + 16);
        HIP_CHECK(hipMemcpyAsync(dev_valid_[c]->p, stage_valid_[c]->p, kb, hipMemcpyHostToDevice, stream_));
      }
      aux_bytes[c] = total_bytes;
      str_uniform_[c] = (uniform >= 0 && uniform <= 15) ? uniform : -1;
      continue;
    }
    const int w = fixed_width(t);
    size_t vbytes = w ? (size_t)rows * w : (size_t)((rows + 7) / 8);
    stage_vals_[c]->ensure(vbytes + 16);
    if (has_valid[c]) stage_valid_[c]->ensure((size_t)((rows + 7) / 8) + 16);
    int64_t at = 0;
    struct CopyJob { char* dst; const char* src; size_t n; };
    std::vector<CopyJob> jobs;
    for (auto& a : held) {
      const ArrowArray* col = a.children[c];
      if (col->dictionary) throw CometError("dictionary-encoded input columns are not unpacked on the GPU path yet");
      const int64_t len = col->length, off = col->offset;
      if (len != a.length) throw CometError("ragged input batch");
      if (cast_from[c].cls != SrcFmt::Unknown) {
        // ScanExec's cast to the declared type, fused into the staging copy; validity = source validity AND "the value fits"
        if (col->null_count != 0 && col->buffers[0]) bit_append((uint8_t*)stage_valid_[c]->p, at, (const uint8_t*)col->buffers[0], off, len);
        else bit_fill_ones((uint8_t*)stage_valid_[c]->p, at, len);
        if (!w) throw CometError("casting a scan input to Boolean is not supported");
        const char* src = (const char*)col->buffers[1] + (size_t)off * (size_t)cast_from[c].width;
        char* dst = (char*)stage_vals_[c]->p + (size_t)at * w;
        uint8_t* vb = (uint8_t*)stage_valid_[c]->p;
        for (int64_t i = 0; i < len; i++) {
          const int64_t bit = at + i;
          const bool ok = ((vb[bit >> 3] >> (bit & 7)) & 1) && scan_cast_value(cast_from[c], src, i, t, dst + (size_t)i * w);
          if (!ok) {
            vb[bit >> 3] &= (uint8_t)~(1u << (bit & 7));
            memset(dst + (size_t)i * w, 0, (size_t)w);
          }
        }
        at += len;
        continue;
      }
      if (w) {
        // Decimal128 buffers from the JVM may be only 8-byte aligned (aligned_stream_reader.rs:95-107);
        // the staging copy realigns them.
        jobs.push_back({(char*)stage_vals_[c]->p + (size_t)at * w, (const char*)col->buffers[1] + (size_t)off * w, (size_t)len * w});
      } else {
        bit_append((uint8_t*)stage_vals_[c]->p, at, (const uint8_t*)col->buffers[1], off, len);
      }
      if (has_valid[c]) {
        if (col->null_count != 0 && col->buffers[0]) bit_append((uint8_t*)stage_valid_[c]->p, at, (const uint8_t*)col->buffers[0], off, len);
        else bit_fill_ones((uint8_t*)stage_valid_[c]->p, at, len);
      }
      at += len;
    }
    if (vbytes >= ((size_t)8 << 20)) {
      // split big single copies so that one huge batch is spread too
      std::vector<CopyJob> pieces;
      const size_t kPiece = (size_t)4 << 20;
      for (auto& j : jobs)
        for (size_t o = 0; o < j.n; o += kPiece) pieces.push_back({j.dst + o, j.src + o, std::min(kPiece, j.n - o)});
      jobs.swap(pieces);
    }
    if (vbytes >= ((size_t)8 << 20) && jobs.size() > 1) {
      // a large column: the batch copies are spread over the scan threads (one thread tops out near 10–15 GB/s, PCIe needs 45+)
      const size_t parts = std::min<size_t>(16, jobs.size());
      scan_pool_parallel(parts, [&](size_t pidx) {
        for (size_t j = pidx; j < jobs.size(); j += parts) stream_copy(jobs[j].dst, jobs[j].src, jobs[j].n);
      });
    } else {
      for (auto& j : jobs) stream_copy(j.dst, j.src, j.n);
    }
    dev_vals_[c]->ensure(vbytes + 16);
    HIP_CHECK(hipMemcpyAsync(dev_vals_[c]->p, stage_vals_[c]->p, vbytes, hipMemcpyHostToDevice, stream_));
    if (has_valid[c]) {
      size_t kb = (size_t)((rows + 7) / 8);
      dev_valid_[c]->ensure(kb + 16);
      HIP_CHECK(hipMemcpyAsync(dev_valid_[c]->p, stage_valid_[c]->p, kb, hipMemcpyHostToDevice, stream_));
    }
  }
  for (auto& a : held) if (a.release) a.release(&a);
  views.assign(nc, DeviceColumnView());
  for (size_t c = 0; c < nc; c++) {
    if (in_types_[c].is_nested()) { views[c] = nested_views[c]; continue; }
    views[c].data = dev_vals_[c]->p;
    views[c].valid = has_valid[c] ? (const uint8_t*)dev_valid_[c]->p : nullptr;
    views[c].aux = dev_aux_[c]->p;
    views[c].fixed_len = str_uniform_[c];   // staged bytes are contiguous from 0, offsets rebased
  }
  rows_out = rows;
  return true;
}

bool ExecutionContext::pull_host_chunk() {
  std::vector<DeviceColumnView> views;
  std::vector<bool> has_valid;
  int64_t rows = 0;
  // The first chunks are small and double up to chunkRows: nothing can overlap the staging of the FIRST chunk (the GPU and the link idle
  // while it is gathered), so it should be short; from then on chunk k + 1 is staged while chunk k crosses PCIe and is consumed.
  if (ramp_rows_ <= 0) ramp_rows_ = std::max<int64_t>(chunk_rows_ / 4, std::min<int64_t>(chunk_rows_, 65536));
  const int64_t want_rows = std::min<int64_t>(chunk_rows_, ramp_rows_);
  ramp_rows_ = std::min<int64_t>(chunk_rows_, ramp_rows_ * 2);
  if (!pull_host_table(0, in_types_, want_rows, views, has_valid, rows)) return false;
  if (rows > 0) {
    process_chunk(views, has_valid, rows);
    // no host-side wait: mark this staging set busy until the work queued so far is done, and switch to the other set
    Staging& stg = *staging_[(size_t)(stage_parity_ & 1)];
    stg.busy = pool_get_event(device_id_);
    HIP_CHECK(hipEventRecord(stg.busy, stream_));
    stage_parity_ ^= 1;
  }
  return !inputs_[0].exhausted;
}


// Verdicts of utf8_uniform_kernel for DECLARED columns (comet:utf8_fixed_len), process-wide: the declaration promises an immutable
// buffer, so (offsets address, rows, length, first offset) identifies what was verified.  Bounded; cleared when full.
namespace {
struct UniformKey {
  const void* off; int64_t rows; int32_t len; int32_t first;
  bool operator<(const UniformKey& o) const { return std::tie(off, rows, len, first) < std::tie(o.off, o.rows, o.len, o.first); }
};
std::mutex g_uniform_mu;
std::set<UniformKey> g_uniform_ok;
bool uniform_verdict_known(const void* off, int64_t rows, int32_t len, int32_t first) {
  std::lock_guard<std::mutex> g(g_uniform_mu);
  return g_uniform_ok.count(UniformKey{off, rows, len, first}) != 0;
}
void remember_uniform_verdict(const void* off, int64_t rows, int32_t len, int32_t first) {
  std::lock_guard<std::mutex> g(g_uniform_mu);
  if (g_uniform_ok.size() >= 4096) g_uniform_ok.clear();
  g_uniform_ok.insert(UniformKey{off, rows, len, first});
}
}  // namespace

bool ExecutionContext::pull_device_table(size_t input, const std::vector<DType>& types, std::vector<DeviceColumnView>& views,
                                         std::vector<bool>& has_valid, int64_t& rows, std::shared_ptr<void>& keepalive) {
  InputSource& in = inputs_[input];
  rows = 0;
  if (in.exhausted) return false;
  static const bool trace = getenv("COMET_TRACE_STAGES") != nullptr;
  Timer tm;
  validate_input_schema(input, types);
  const double t_schema = tm.ns();
  auto da = std::make_shared<ArrowDeviceArray>();
  memset(da.get(), 0, sizeof(ArrowDeviceArray));
  int rc = in.dev->get_next(in.dev, da.get());
  if (trace) fprintf(stderr, "[comet] device input %zu: get_schema %.3f ms, get_next %.3f ms\n", input, t_schema / 1e6, (tm.ns() - t_schema) / 1e6);
  if (rc != 0) {
    const char* m = in.dev->get_last_error ? in.dev->get_last_error(in.dev) : nullptr;
    throw CometError(std::string("input ArrowDeviceArrayStream.get_next failed: ") + (m ? m : "unknown error"));
  }
  if (!da->array.release) {
    in.exhausted = true;
    return false;
  }
  // the producer's buffers stay alive until the keepalive is dropped
  keepalive = std::shared_ptr<void>(da.get(), [da](void*) mutable {
    if (da->array.release) da->array.release(&da->array);
  });
  if (da->device_type != ARROW_DEVICE_ROCM && da->device_type != ARROW_DEVICE_ROCM_HOST)
    throw CometError("device input stream must carry ARROW_DEVICE_ROCM memory");
  if (da->sync_event) HIP_CHECK(hipStreamWaitEvent(stream_, *(hipEvent_t*)da->sync_event, 0));
  const size_t nc = types.size();
  if ((size_t)da->array.n_children != nc) throw CometError("device batch column count does not match Scan fields");
  views.assign(nc, DeviceColumnView());
  has_valid.assign(nc, false);
  for (size_t c = 0; c < nc; c++) {
    const ArrowArray* col = da->array.children[c];
    if (col->dictionary) throw CometError("dictionary-encoded device columns are not supported yet");
    views[c].data = col->buffers[1];
    views[c].offset = col->offset;
    if (types[c].id == TypeId::String || types[c].id == TypeId::Bytes) views[c].aux = col->buffers[2];
    if (types[c].id == TypeId::Decimal && (((uintptr_t)col->buffers[1]) & 15))
      throw CometError("device Decimal128 buffers must be 16-byte aligned");
    if (col->null_count != 0 && col->buffers[0]) {
      has_valid[c] = true;
      views[c].valid = (const uint8_t*)col->buffers[0];
    }
  }
  rows = da->array.length;
  // Utf8 columns: check on the device whether all values share one length (one pass over the offsets, 4 B/row); if so the fused
  // kernels skip the offsets and the dependent byte load altogether.  All columns at once: the first / last offsets of every column
  // come back with ONE synchronisation, the verification launches run back to back, their flags come back with a second one.
  std::vector<size_t> scols;
  for (size_t c = 0; c < nc && rows > 0; c++)
    if (types[c].id == TypeId::String) scols.push_back(c);
  if (!scols.empty() && scols.size() <= 16) {
    small_host_.ensure(4096);
    int32_t* ends = (int32_t*)small_host_.p;                 // [2 k], [2 k + 1] = first / last offset of string column k
    for (size_t k = 0; k < scols.size(); k++) {
      const ArrowArray* col = da->array.children[scols[k]];
      const int32_t* off = (const int32_t*)col->buffers[1] + col->offset;
      HIP_CHECK(hipMemcpyAsync(ends + 2 * k, off, 4, hipMemcpyDeviceToHost, stream_));
      HIP_CHECK(hipMemcpyAsync(ends + 2 * k + 1, off + rows, 4, hipMemcpyDeviceToHost, stream_));
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    std::vector<int32_t> first(scols.size()), len(scols.size(), -1);
    std::vector<bool> declared(scols.size(), false);
    uint32_t* flags = (uint32_t*)err_flags_.p + (kErrBytes / 4 - 16);   // last 16 words of the error/aux block: scratch
    HIP_CHECK(hipMemsetAsync(flags, 0, 64, stream_));
    bool any = false;
    if (!aux_ev_[0]) { aux_ev_[0] = pool_get_event(device_id_); aux_ev_[1] = pool_get_event(device_id_); }
    HIP_CHECK(hipEventRecord(aux_ev_[0], stream_));
    int aux_launched = 0;
    for (size_t k = 0; k < scols.size(); k++) {
      first[k] = ends[2 * k];
      const int64_t total = (int64_t)ends[2 * k + 1] - ends[2 * k];
      const int hint = input < fixed_len_hint_.size() && scols[k] < fixed_len_hint_[input].size() ? fixed_len_hint_[input][scols[k]] : -1;
      const ArrowArray* col = da->array.children[scols[k]];
      const int32_t* off = (const int32_t*)col->buffers[1] + col->offset;
      if (hint >= 0) {
        // A declaration is a promise about an IMMUTABLE buffer, and it is still verified: the end points on every batch, every offset in
        // between by utf8_uniform_kernel the first time this (offsets address, rows, length) is seen in the process — the verdict is
        // remembered, so the tasks after the first pay nothing.  A batch that contradicts the declaration is refused, never mis-read.
        if (total != (int64_t)hint * rows)
          throw CometError("device input column " + std::to_string(scols[k]) + " is declared comet:utf8_fixed_len=" + std::to_string(hint) + " but holds " +
                           std::to_string(total) + " bytes in " + std::to_string(rows) + " rows");
        if ((int64_t)first[k] != (int64_t)col->offset * hint) continue;      // a sliced column: offset-based accessors only
        if (uniform_verdict_known(off, rows, hint, first[k])) {
          views[scols[k]].fixed_len = hint;
          continue;
        }
        if (comet_launch_utf8_uniform(off, rows, hint, flags + k, stream_) != 0) continue;
        aux_launched++;
        len[k] = hint;
        declared[k] = true;
        any = true;
        continue;
      }
      if (total % rows != 0 || total / rows > 15 || total < 0) continue;
      if (comet_launch_utf8_uniform(off, rows, (int32_t)(total / rows), flags + k, stream_) != 0) continue;
      aux_launched++;
      len[k] = (int32_t)(total / rows);
      any = true;
    }
    if (any) {
      HIP_CHECK(hipEventRecord(aux_ev_[1], stream_));
      uint32_t f[16];
      read_small(f, flags, 64);      // (synchronises the stream: the event pair is complete)
      float aux = 0;
      if (hipEventElapsedTime(&aux, aux_ev_[0], aux_ev_[1]) == hipSuccess) { last_aux_ms += aux; last_aux_launches += aux_launched; }
      HIP_CHECK(hipMemsetAsync(flags, 0, 64, stream_));
      for (size_t k = 0; k < scols.size(); k++) {
        if (len[k] < 0) continue;
        const size_t c = scols[k];
        const ArrowArray* col = da->array.children[c];
        if (declared[k]) {
          if (f[k] != 0)
            throw CometError("device input column " + std::to_string(c) + " is declared comet:utf8_fixed_len=" + std::to_string(len[k]) +
                             " but its offsets are not " + std::to_string(len[k]) + " bytes apart (the byte total alone matches)");
          remember_uniform_verdict((const int32_t*)col->buffers[1] + col->offset, rows, len[k], first[k]);
          views[c].fixed_len = len[k];
          continue;
        }
        if (f[k] != 0) continue;
        // value i then sits at aux + (offset + i)·len.  Only claimed when that base IS the data buffer (an unsliced column), because the
        // offset-based accessors (substring, LIKE, views …) of the same kernels keep addressing aux + offsets[i]
        if ((int64_t)first[k] != (int64_t)col->offset * len[k]) continue;
        views[c].fixed_len = len[k];
      }
    }
  }
  return true;
}

// HBM-resident input (Arrow C Device stream, ARROW_DEVICE_ROCM): zero copy.
bool ExecutionContext::pull_device_batch() {
  DevPending cur;
  if (dev_pending_.valid) {
    cur = std::move(dev_pending_);
    dev_pending_ = DevPending();
  } else {
    if (dev_stream_done_ || !pull_device_table(0, in_types_, cur.views, cur.has_valid, cur.rows, cur.keep)) return false;
  }
  // A grouped aggregate looks ONE batch ahead before its first chunk: a merging aggregate whose whole input is that chunk can run partitioned
  // (exec_pipeline.cpp try_partitioned_merge) — the usual shape of a Final aggregate over a device-resident table of Partial states
  if (sink_ == SinkKind::AggGrouped && dev_chunks_seen_ == 0 && !dev_stream_done_) {
    dev_pending_.valid = pull_device_table(0, in_types_, dev_pending_.views, dev_pending_.has_valid, dev_pending_.rows, dev_pending_.keep);
    if (!dev_pending_.valid) dev_stream_done_ = true;
    single_chunk_hint_ = !dev_pending_.valid;
  } else {
    single_chunk_hint_ = false;
  }
  dev_chunks_seen_++;
  process_chunk(cur.views, cur.has_valid, cur.rows);
  HIP_CHECK(hipStreamSynchronize(stream_));
  return true;
}

// ---------------------------------------------------------------------------------------------
// Plans with joins: every join input is materialised in HBM (chains are fused pipelines, joins are the
// materialisation points), then the root chain streams over the top join's output.
// ---------------------------------------------------------------------------------------------

}  // namespace comet
