// Sort / TopK / Limit: order-preserving key bytes, LSD radix sort over the varying planes, row gathers.
#include "exec_internal.hpp"

namespace comet {
// rows [first, first + rows) of `in` in the order given by dev_perm (nullptr = identity) → a new resident table
DevTable ExecutionContext::take_rows(const DevTable& in, const uint32_t* dev_perm, int64_t first, int64_t rows, std::shared_ptr<DevBuf> perm_owner) {
  DevTable out;
  out.rows = rows;
  out.types = in.types;
  out.has_valid = in.has_valid;
  out.cols.assign(in.cols.size(), DeviceColumnView());
  std::shared_ptr<DevBuf> perm = perm_owner;
  if (!dev_perm) {
    perm = std::make_shared<DevBuf>();
    perm->ensure((size_t)std::max<int64_t>(rows, 1) * 4);
    if (comet_launch_sort_iota((uint32_t*)perm->p, rows, (uint32_t)first, stream_) != 0) throw CometError("limit: launch failed");
    dev_perm = (const uint32_t*)perm->p;
    first = 0;
  }
  const uint32_t* idx = dev_perm + first;
  for (size_t c = 0; c < in.cols.size(); c++) {
    const DType& t = in.types[c];
    if (t.is_nested()) {      // children and all (exec.cpp take_column)
      bool hv = false;
      out.cols[c] = take_column(in.cols[c], t, in.has_valid[c], idx, nullptr, rows, hv, out.owners);
      out.has_valid[c] = hv;
      continue;
    }
    if (t.id == TypeId::String || t.id == TypeId::Bytes) {
      if (in.cols[c].offset != 0 && in.has_valid[c]) throw CometError("Sort / Limit over a nullable Utf8 column with a non-zero Arrow offset is not supported yet");
      take_utf8(in.cols[c], idx, nullptr, in.has_valid[c] ? in.cols[c].valid : nullptr, rows, out.cols[c], out.owners);
      if (in.has_valid[c]) {
        auto bm = std::make_shared<DevBuf>();
        bm->ensure((size_t)((rows + 7) / 8) + 16);
        if (rows && comet_launch_take(0, in.cols[c].valid, idx, rows, bm->p, stream_) != 0) throw CometError("take: validity");
        out.cols[c].valid = (const uint8_t*)bm->p;
        out.owners.push_back(bm);
      }
      continue;
    }
    if (in.cols[c].offset != 0) throw CometError("Sort / Limit over a column with a non-zero Arrow offset is not supported yet");
    const int w = t.id == TypeId::Bool ? 0 : fixed_width(t);
    auto vals = std::make_shared<DevBuf>();
    vals->ensure((w ? (size_t)std::max<int64_t>(rows, 1) * w : (size_t)((rows + 7) / 8)) + 16);
    if (rows && comet_launch_take(w, in.cols[c].data, idx, rows, vals->p, stream_) != 0) throw CometError("take: unsupported width");
    out.cols[c].data = vals->p;
    out.owners.push_back(vals);
    if (in.has_valid[c]) {
      auto bm = std::make_shared<DevBuf>();
      bm->ensure((size_t)((rows + 7) / 8) + 16);
      if (rows && comet_launch_take(0, in.cols[c].valid, idx, rows, bm->p, stream_) != 0) throw CometError("take: validity");
      out.cols[c].valid = (const uint8_t*)bm->p;
      out.owners.push_back(bm);
    }
  }
  HIP_CHECK(hipStreamSynchronize(stream_));   // `in` (and the permutation) may be released by the caller
  return out;
}

// Explode: one output row per element of a List column (see infer_schema for the plan shape).  counts → prefix sum → per output row its
// input row / element / position (exchange_kernels.hip explode_*), then the carried columns taken by input row and the element column taken
// by element index — any element type, children and all.
DevTable ExecutionContext::explode(const Operator& ex, const DevTable& in) {
  const int lc = ex.explode_child->bound_index;
  const DeviceColumnView& lv = in.cols.at((size_t)lc);
  const DType& lt = in.types.at((size_t)lc);
  if (lt.id != TypeId::List || lv.kids.size() != 1) throw CometError("Explode: the exploded column is not a resident list");
  // (the explode kernels index offsets and validity from the buffers' first element: a sliced list column would read other rows')
  if (lv.offset != 0 || lv.kids[0].offset != 0) throw CometError("Explode: a sliced (offset != 0) list column is not supported");
  const int64_t n = in.rows;
  const uint8_t* lvalid = in.has_valid[(size_t)lc] ? lv.valid : nullptr;
  const bool ehv = !lv.kid_has_valid.empty() && lv.kid_has_valid[0] && lv.kids[0].valid;
  // the carried columns, once per input row
  DevTable carried;
  auto pit = explode_proj_.find(&ex);
  if (pit != explode_proj_.end()) carried = run_chain_to_device(*pit->second, in);
  DevBuf counts, tiles;
  auto out_offs = std::make_shared<DevBuf>();
  counts.ensure((size_t)std::max<int64_t>(n, 1) * 4 + 16);
  tiles.ensure((size_t)((n + 1023) / 1024 + 2) * 8);
  out_offs->ensure((size_t)(n + 1) * 4 + 16);
  int32_t total = 0;
  if (n > 0) {
    if (comet_launch_explode_counts((const int32_t*)lv.data, lvalid, n, ex.explode_outer ? 1 : 0, (uint32_t*)counts.p, stream_) != 0) throw CometError("explode: launch failed");
    pq_launch_u32_scan((const uint32_t*)counts.p, n, (uint64_t*)tiles.p, (int32_t*)out_offs->p, stream_);
    read_small(&total, (char*)out_offs->p + (size_t)n * 4, 4);
    if (total < 0) throw CometError("Explode: more than 2^31 output rows in one partition");
  }
  auto row_idx = std::make_shared<DevBuf>(), elem_idx = std::make_shared<DevBuf>(), has_elem = std::make_shared<DevBuf>(), pos = std::make_shared<DevBuf>(),
       elem_ok = std::make_shared<DevBuf>();
  const size_t cap = (size_t)std::max(total, 1);
  row_idx->ensure(cap * 4 + 16);
  elem_idx->ensure(cap * 4 + 16);
  has_elem->ensure(cap + 16);
  pos->ensure(cap * 4 + 16);
  elem_ok->ensure(cap + 16);
  if (total > 0 &&
      comet_launch_explode_indices((const int32_t*)lv.data, lvalid, ehv ? lv.kids[0].valid : nullptr, n, (const int32_t*)out_offs->p, (uint32_t*)row_idx->p, (uint32_t*)elem_idx->p,
                                   (uint8_t*)has_elem->p, (int32_t*)pos->p, (uint8_t*)elem_ok->p, stream_) != 0)
    throw CometError("explode: launch failed");
  DevTable out;
  out.rows = total;
  for (size_t c = 0; c < carried.cols.size(); c++) {
    bool hv = false;
    out.cols.push_back(take_column(carried.cols[c], carried.types[c], carried.has_valid[c], (const uint32_t*)row_idx->p, nullptr, total, hv, out.owners));
    out.types.push_back(carried.types[c]);
    out.has_valid.push_back(hv);
  }
  auto pack = [&](const std::shared_ptr<DevBuf>& bytes) -> const uint8_t* {
    auto bm = std::make_shared<DevBuf>();
    bm->ensure((size_t)((total + 7) / 8) + 16);
    if (total > 0) pq_launch_pack((const uint8_t*)bytes->p, (uint8_t*)bm->p, total, stream_);
    out.owners.push_back(bm);
    return (const uint8_t*)bm->p;
  };
  if (ex.explode_position) {
    DeviceColumnView pv;
    pv.data = pos->p;
    if (ex.explode_outer) pv.valid = pack(has_elem);      // the NULL row of an empty / NULL list has no position
    out.cols.push_back(pv);
    out.types.push_back(DType::of(TypeId::Int32));
    out.has_valid.push_back(ex.explode_outer);
    out.owners.push_back(pos);
  }
  {
    // the element: taken by its index; its validity is "there is one, and its own bit says valid" (elem_ok), whatever its type
    bool hv_unused = false;
    DeviceColumnView ev = take_column(lv.kids[0], lt.kids[0], false, (const uint32_t*)elem_idx->p, (const uint8_t*)has_elem->p, total, hv_unused, out.owners);
    const bool nullable = ex.explode_outer || ehv;
    if (nullable) ev.valid = pack(elem_ok);
    out.cols.push_back(ev);
    out.types.push_back(lt.kids[0]);
    out.has_valid.push_back(nullable);
  }
  HIP_CHECK(hipStreamSynchronize(stream_));      // counts / tiles / index buffers may go back to their pools; `in` may be released by the caller
  out.owners.push_back(out_offs);
  for (auto& o : carried.owners) out.owners.push_back(o);
  return out;
}

// Sort (planner.rs:1488-1522 → SortExec with fetch / skip): order-preserving key bytes per row (generated kernel), LSD radix
// sort of a row permutation over the byte planes that actually vary, then one take per column of rows [skip, skip+fetch).
// order-preserving key bytes of every row of `in` under sop.sort_orders, as W byte planes of n rows (plane p of row i at p·n + i)
std::shared_ptr<DevBuf> ExecutionContext::sort_key_planes(const Operator& sop, const DevTable& in, int& W, std::vector<int64_t>* str_len, bool measure_only) {
  const int64_t n = in.rows;
  std::string key = std::to_string(plan_hash_ ^ (0x9E3779B97F4A7C15ull * (uint64_t)(node_id_[&sop] + 1))) + ":S:" + validity_key(in.has_valid);
  std::shared_ptr<PlannedVariant> pv;
  {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    if (it != g_plan_cache.end()) pv = it->second;
  }
  if (!pv) {
    pv = std::make_shared<PlannedVariant>();
    pv->desc = generate_sort_keys(sop, in.types, in.has_valid);
    pv->code = jit_compile(pv->desc.source);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plan_cache[key] = pv;
  }
  Variant v;
  v.desc = pv->desc;
  note_sites(v.desc);
  v.mod = jit_load(pv->code);
  W = v.desc.sort_key_bytes;
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  // Utf8 sort keys: padded to the longest value of the column (measured here, or imposed by the caller when two tables must share
  // one key layout — range-partition boundaries)
  std::vector<int64_t> lens;
  for (size_t s = 0; s < v.desc.sort_str_cols.size(); s++) {
    int64_t L = 0;
    if (str_len && s < str_len->size() && !measure_only) L = (*str_len)[s];
    else if (n > 0) {
      const DeviceColumnView& sc = in.cols[(size_t)v.desc.sort_str_cols[s]];
      uint32_t* mx = (uint32_t*)err_flags_.p + (kErrBytes / 4 - 1);
      HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
      if (comet_launch_str_max_len((const int32_t*)sc.data + sc.offset, n, mx, stream_) != 0) throw CometError("sort: launch failed");
      uint32_t longest = 0;
      read_small(&longest, mx, 4);
      HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
      L = longest;
    }
    lens.push_back(L);
    prm.iarg[1 + s] = L + 4;
    W += (int)(L + 4);
  }
  if (str_len) *str_len = lens;
  if (measure_only) return nullptr;
  if (W > 1000) throw CometError("Sort key wider than 1000 bytes (Utf8 sort keys are padded to their longest value)");
  auto planes = std::make_shared<DevBuf>();
  planes->ensure((size_t)W * (size_t)std::max<int64_t>(n, 1) + 16);
  if (n == 0) return planes;
  prm.n = n;
  for (size_t i = 0; i < in.cols.size(); i++) {
    prm.in[i].data = in.cols[i].data;
    prm.in[i].valid = in.has_valid[i] ? in.cols[i].valid : nullptr;
    prm.in[i].aux = in.cols[i].aux;
    prm.in[i].offset = in.cols[i].offset;
  }
  prm.out[0] = planes->p;
  prm.out[kOutErr] = err_flags_.p;
  launch(v, "k_sortkey", (int)std::min<int64_t>((n + 255) / 256, 256 * 8), prm);
  planes_owner_ = v.mod;   // the module must stay loaded until the launch has run; callers synchronise before returning
  return planes;
}

// a small resident table from literal rows (range-partition boundaries): one column per entry of `types`
DevTable ExecutionContext::literal_table(const std::vector<std::vector<ExprP>>& rows, const std::vector<DType>& types) {
  DevTable t;
  const int64_t n = (int64_t)rows.size();
  t.rows = n;
  for (size_t c = 0; c < types.size(); c++) {
    const DType& ty = types[c];
    const bool is_str = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    std::vector<uint8_t> vals, data, valid((size_t)((n + 7) / 8) + 1, 0);
    std::vector<int32_t> offs(1, 0);
    const int w = is_str ? 0 : ty.id == TypeId::Bool ? 0 : fixed_width(ty);
    if (ty.id == TypeId::Bool) vals.assign((size_t)((n + 7) / 8) + 1, 0);
    for (int64_t r = 0; r < n; r++) {
      const Expr& e = *rows[(size_t)r][c];
      if (!e.lit_null) valid[(size_t)(r >> 3)] |= (uint8_t)(1u << (r & 7));
      if (is_str) {
        if (!e.lit_null) data.insert(data.end(), e.lit_bytes.begin(), e.lit_bytes.end());
        offs.push_back((int32_t)data.size());
      } else if (ty.id == TypeId::Bool) {
        if (!e.lit_null && e.lit_bool) vals[(size_t)(r >> 3)] |= (uint8_t)(1u << (r & 7));
      } else {
        uint8_t buf[16] = {0};
        if (!e.lit_null) {
          if (ty.id == TypeId::Decimal) { i128 v = e.lit_dec; memcpy(buf, &v, 16); }
          else if (ty.id == TypeId::Double) { double v = e.lit_f64; memcpy(buf, &v, 8); }
          else if (ty.id == TypeId::Float) { float v = (float)e.lit_f64; memcpy(buf, &v, 4); }
          else { int64_t v = e.lit_i64; memcpy(buf, &v, 8); }   // little endian: the low `w` bytes are the narrower integer
        }
        vals.insert(vals.end(), buf, buf + w);
      }
    }
    auto up = [&](const void* p, size_t bytes) {
      auto b = std::make_shared<DevBuf>();
      b->ensure(bytes + 16);
      if (bytes) HIP_CHECK(hipMemcpy(b->p, p, bytes, hipMemcpyHostToDevice));
      t.owners.push_back(b);
      return b->p;
    };
    DeviceColumnView v;
    if (is_str) {
      v.data = up(offs.data(), offs.size() * 4);
      v.aux = up(data.data(), data.size());
    } else {
      v.data = up(vals.data(), vals.size());
    }
    v.valid = (const uint8_t*)up(valid.data(), valid.size());
    t.types.push_back(ty);
    t.cols.push_back(v);
    t.has_valid.push_back(true);
  }
  return t;
}

DevTable ExecutionContext::sort_table(const Operator& sop, const DevTable& in) {
  const int64_t n = in.rows;
  if (n >= ((int64_t)1 << 32)) throw CometError("Sort: more than 2^32 rows in one partition");
  const int64_t skip = std::min<int64_t>(std::max(0, sop.skip), n);
  const int64_t keep = sop.fetch >= 0 ? std::min<int64_t>(n, sop.fetch) : n;     // fetch counts from the first row (GlobalLimit(skip) on top)
  const int64_t out_rows = std::max<int64_t>(0, keep - skip);
  if (n == 0 || out_rows == 0) return take_rows(in, nullptr, 0, 0, nullptr);
  timed_begin();
  int W = 0;
  auto planes = sort_key_planes(sop, in, W);
  // which planes vary at all?
  DevBuf flags;
  flags.ensure((size_t)W * 4 + 16);
  HIP_CHECK(hipMemsetAsync(flags.p, 0, (size_t)W * 4, stream_));
  if (comet_launch_sort_plane_varies((const uint8_t*)planes->p, n, W, (uint32_t*)flags.p, stream_) != 0) throw CometError("sort: launch failed");
  std::vector<uint32_t> varies((size_t)W);
  if ((size_t)W * 4 > small_host_.cap) HIP_CHECK(hipStreamSynchronize(stream_));      // (growing the staging frees the old block: no copy out of write_small's slots may be in flight)
  small_host_.ensure(std::max<size_t>(4096, (size_t)W * 4));
  HIP_CHECK(hipMemcpyAsync(small_host_.p, flags.p, (size_t)W * 4, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
  memcpy(varies.data(), small_host_.p, (size_t)W * 4);
  auto perm = std::make_shared<DevBuf>();
  auto perm2 = std::make_shared<DevBuf>();
  perm->ensure((size_t)n * 4 + 16);
  perm2->ensure((size_t)n * 4 + 16);
  if (comet_launch_sort_iota((uint32_t*)perm->p, n, 0, stream_) != 0) throw CometError("sort: launch failed");
  int64_t ns = n;   // rows that take part in the full sort
  int select_passes = 0;
  if (sop.fetch >= 0 && keep * 8 < n) {
    // TopK: radix select from the most significant varying plane down.  `sure` rows are certainly among the first `keep`;
    // only the bucket that straddles the K-th position stays a candidate.  What is left (sure ∪ candidates) is sorted.
    auto sure = std::make_shared<DevBuf>();
    sure->ensure((size_t)n * 4 + 16);
    DevBuf sel;   // [0..255] u64 histogram, then two u32 counters
    sel.ensure(256 * 8 + 16);
    HIP_CHECK(hipMemsetAsync((char*)sel.p + 256 * 8, 0, 8, stream_));
    uint32_t* counters = (uint32_t*)((char*)sel.p + 256 * 8);
    int64_t m = n, need = keep, nsure = 0;
    for (int b = 0; b < W && m > std::max<int64_t>(4096, need); b++) {
      if (!varies[(size_t)b]) continue;
      const uint8_t* plane = (const uint8_t*)planes->p + (size_t)b * (size_t)n;
      HIP_CHECK(hipMemsetAsync(sel.p, 0, 256 * 8, stream_));
      if (comet_launch_sort_hist256(plane, (const uint32_t*)perm->p, m, (uint64_t*)sel.p, stream_) != 0) throw CometError("sort: launch failed");
      uint64_t h[256];
      HIP_CHECK(hipMemcpyAsync(small_host_.p, sel.p, 256 * 8, hipMemcpyDeviceToHost, stream_));
      HIP_CHECK(hipStreamSynchronize(stream_));
      memcpy(h, small_host_.p, sizeof h);
      int dstar = 255;
      int64_t below = 0;
      for (int dgt = 0; dgt < 256; dgt++) {
        if (below + (int64_t)h[dgt] >= need) { dstar = dgt; break; }
        below += (int64_t)h[dgt];
      }
      uint32_t cnt2[2] = {(uint32_t)nsure, 0};
      write_small(counters, cnt2, 8);
      if (comet_launch_sort_select(plane, (const uint32_t*)perm->p, m, dstar, (uint32_t*)sure->p, (uint32_t*)perm2->p, counters, stream_) != 0)
        throw CometError("sort: launch failed");
      std::swap(perm, perm2);
      nsure += below;
      need -= below;
      m = (int64_t)h[dstar];
      select_passes++;
    }
    // rows to sort = sure ++ remaining candidates
    if (select_passes) {
      HIP_CHECK(hipMemcpyAsync((char*)sure->p + (size_t)nsure * 4, perm->p, (size_t)m * 4, hipMemcpyDeviceToDevice, stream_));
      HIP_CHECK(hipStreamSynchronize(stream_));
      perm = sure;
      ns = nsure + m;
    }
  }
  // few rows left (a TopK after its select passes, or a small input): one workgroup sorts them by their varying key bytes
  std::vector<int> vplanes;
  for (int b = 0; b < W; b++)
    if (varies[(size_t)b]) vplanes.push_back(b);
  static const bool small_sort = getenv("COMET_SORT_SMALL") == nullptr || atoi(getenv("COMET_SORT_SMALL")) != 0;
  if (small_sort && ns <= 4096 && vplanes.size() <= 16) {
    if (comet_launch_sort_small((const uint8_t*)planes->p, n, (const uint32_t*)perm->p, (int)ns, vplanes.data(), (int)vplanes.size(), (uint32_t*)perm2->p, stream_) != 0)
      throw CometError("sort: launch failed");
    timed_end();
    if (getenv("COMET_TRACE_STAGES"))
      fprintf(stderr, "[comet] sort: %lld rows, key %d bytes, %d select passes -> %lld rows sorted by one workgroup\n", (long long)n, W, select_passes, (long long)ns);
    return take_rows(in, (const uint32_t*)perm2->p, skip, out_rows, perm2);   // synchronises the stream
  }
  DevBuf digit, ridx, hist, starts;
  digit.ensure((size_t)ns * 4 + 16);
  ridx.ensure((size_t)ns * 4 + 16);
  const int64_t Wt = comet_partition_tiles(ns);
  hist.ensure((size_t)comet_partition_scratch_bytes(ns, 256));
  starts.ensure(257 * 8);
  uint32_t* bad = (uint32_t*)((char*)hist.p + ((size_t)256 * (size_t)Wt + 1) * 8);
  HIP_CHECK(hipMemsetAsync(bad, 0, 4, stream_));
  int passes = 0;
  for (int b = W - 1; b >= 0; b--) {
    if (!varies[(size_t)b]) continue;
    const uint8_t* plane = (const uint8_t*)planes->p + (size_t)b * (size_t)n;
    if (comet_launch_sort_gather_digit(plane, (const uint32_t*)perm->p, ns, (int32_t*)digit.p, stream_) != 0 ||
        comet_launch_partition_indices((const int32_t*)digit.p, ns, 256, (uint64_t*)hist.p, bad, (int64_t*)starts.p, (uint32_t*)ridx.p, stream_) != 0 ||
        comet_launch_take(4, perm->p, (const uint32_t*)ridx.p, ns, perm2->p, stream_) != 0)
      throw CometError("sort: launch failed");
    std::swap(perm, perm2);
    passes++;
  }
  timed_end();
  if (getenv("COMET_TRACE_STAGES"))
    fprintf(stderr, "[comet] sort: %lld rows, key %d bytes, %d select passes -> %lld rows sorted in %d radix passes\n", (long long)n, W, select_passes,
            (long long)ns, passes);
  DevTable out = take_rows(in, (const uint32_t*)perm->p, skip, out_rows, perm);   // synchronises the stream: the key kernel has run
  return out;
}

}  // namespace comet
