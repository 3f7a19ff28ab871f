// Hash joins of materialised tables (build, single-pass probe, outer-build tail).
#include "exec_internal.hpp"

namespace comet {
// Join keys that are Utf8 columns with values longer than the 15 bytes of a packed key: an exact string dictionary is built over the
// right column (strdict_kernels.hip), the left column is looked up in it, and the join runs on the two Int64 row-index columns
// instead (a left string without a partner gets a NULL index: NULL keys never match, outer joins still emit the row).  The index
// columns are appended to the inputs and dropped from the result.
DevTable ExecutionContext::hash_join(const Operator& j, const DevTable& L, const DevTable& R) {
  auto is_str = [](const DType& t) { return t.id == TypeId::String || t.id == TypeId::Bytes; };
  std::vector<size_t> sk;
  for (size_t k = 0; k < j.left_keys.size() && k < j.right_keys.size(); k++) {
    const ExprP &a = j.left_keys[k], &b = j.right_keys[k];
    if (a->kind == ExprKind::Bound && b->kind == ExprKind::Bound && a->bound_index >= 0 && b->bound_index >= 0 && (size_t)a->bound_index < L.types.size() &&
        (size_t)b->bound_index < R.types.size() && is_str(L.types[(size_t)a->bound_index]) && is_str(R.types[(size_t)b->bound_index]) &&
        L.cols[(size_t)a->bound_index].offset == 0 && R.cols[(size_t)b->bound_index].offset == 0)
      sk.push_back(k);
  }
  if (sk.empty() || L.rows == 0 || R.rows == 0 || L.cols.size() + R.cols.size() + 2 * sk.size() > COMET_MAX_IN) return hash_join_impl(j, j, L, R, "");
  auto longest = [&](const DevTable& t, int c) {
    uint32_t* mx = (uint32_t*)err_flags_.p + (kErrBytes / 4 - 1);
    HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
    if (comet_launch_str_max_len((const int32_t*)t.cols[(size_t)c].data, t.rows, mx, stream_) != 0) throw CometError("string keys: launch failed");
    uint32_t v = 0;
    read_small(&v, mx, 4);
    HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
    return v;
  };
  bool need = false;
  for (size_t k : sk) need = need || longest(L, j.left_keys[k]->bound_index) > 15 || longest(R, j.right_keys[k]->bound_index) > 15;
  if (!need) return hash_join_impl(j, j, L, R, "");
  if (R.rows >= ((int64_t)1 << 32) - 1) throw CometError("Utf8 join keys longer than 15 bytes over more than 2^32 rows are not supported");
  DevTable l2 = L, r2 = R;
  Operator jj = j;
  auto bound = [](int idx) {
    auto e = std::make_shared<Expr>();
    e->kind = ExprKind::Bound;
    e->proto_tag = 3;
    e->bound_index = idx;
    e->dtype = DType::of(TypeId::Int64);
    e->has_dtype = true;
    return e;
  };
  for (size_t k : sk) {
    const int lc = j.left_keys[k]->bound_index, rc = j.right_keys[k]->bound_index;
    const DeviceColumnView &lv = L.cols[(size_t)lc], &rv = R.cols[(size_t)rc];
    int64_t slots = 1024;
    while (slots < 2 * R.rows) slots <<= 1;
    DevBuf table;
    table.ensure((size_t)slots * 4);
    HIP_CHECK(hipMemsetAsync(table.p, 0, (size_t)slots * 4, stream_));
    auto rrep = std::make_shared<DevBuf>(), lrep = std::make_shared<DevBuf>(), lbits = std::make_shared<DevBuf>();
    DevBuf lok;
    rrep->ensure((size_t)R.rows * 8 + 16);
    lrep->ensure((size_t)L.rows * 8 + 16);
    lok.ensure((size_t)L.rows + 16);
    lbits->ensure((size_t)((L.rows + 7) / 8) + 16);
    if (comet_launch_str_dict_build((const int32_t*)rv.data, (const uint8_t*)rv.aux, R.has_valid[(size_t)rc] ? rv.valid : nullptr, R.rows, (uint32_t*)table.p, slots,
                                    (int64_t*)rrep->p, stream_) != 0 ||
        comet_launch_str_dict_lookup((const int32_t*)rv.data, (const uint8_t*)rv.aux, (const uint32_t*)table.p, slots, (const int32_t*)lv.data, (const uint8_t*)lv.aux,
                                     L.has_valid[(size_t)lc] ? lv.valid : nullptr, L.rows, (int64_t*)lrep->p, (uint8_t*)lok.p, stream_) != 0)
      throw CometError("string keys: launch failed");
    pq_launch_pack((const uint8_t*)lok.p, (uint8_t*)lbits->p, L.rows, stream_);
    HIP_CHECK(hipStreamSynchronize(stream_));   // `table` and `lok` go back to the pool
    DeviceColumnView rid, lid;
    rid.data = rrep->p;
    rid.valid = rv.valid;
    lid.data = lrep->p;
    lid.valid = (const uint8_t*)lbits->p;
    jj.right_keys[k] = bound((int)r2.cols.size());
    jj.left_keys[k] = bound((int)l2.cols.size());
    r2.types.push_back(DType::of(TypeId::Int64));
    r2.cols.push_back(rid);
    r2.has_valid.push_back(R.has_valid[(size_t)rc]);
    r2.owners.push_back(rrep);
    l2.types.push_back(DType::of(TypeId::Int64));
    l2.cols.push_back(lid);
    l2.has_valid.push_back(true);
    l2.owners.push_back(lrep);
    l2.owners.push_back(lbits);
  }
  const size_t nl = L.cols.size(), nr = R.cols.size(), extra = sk.size();
  if (jj.join_condition) {
    // the residual condition addresses left ++ right: the right columns moved up by the index columns appended to the left
    std::function<ExprP(const ExprP&)> shift = [&](const ExprP& e) -> ExprP {
      auto c = std::make_shared<Expr>(*e);
      if (e->kind == ExprKind::Bound && (size_t)e->bound_index >= nl) c->bound_index = e->bound_index + (int)extra;
      for (auto& ch : c->children) ch = shift(ch);
      return c;
    };
    jj.join_condition = shift(j.join_condition);
  }
  DevTable out = hash_join_impl(j, jj, l2, r2, ":SD");
  // drop the index columns: the result is left' ++ right' (semi / anti joins: left' only)
  auto drop = [&](size_t first, size_t count) {
    if (first + count > out.cols.size()) return;
    out.types.erase(out.types.begin() + (long)first, out.types.begin() + (long)(first + count));
    out.cols.erase(out.cols.begin() + (long)first, out.cols.begin() + (long)(first + count));
    out.has_valid.erase(out.has_valid.begin() + (long)first, out.has_valid.begin() + (long)(first + count));
  };
  if (out.cols.size() == nl + extra + nr + extra) drop(nl + extra + nr, extra);
  else if (out.cols.size() != nl + extra) throw CometError("internal: unexpected join output width with string keys");
  drop(nl, extra);
  return out;
}

DevTable ExecutionContext::hash_join_impl(const Operator& node, const Operator& j, const DevTable& L, const DevTable& R, const std::string& key_suffix,
                                          const JoinFusion* fused_probe, const JoinFusion* fused_build) {
  // with `fused_build` the build-side table (L when the build side is left) is the SOURCE of the build chain
  // with `fused_probe` the probe-side table (R when the build side is left) is the SOURCE of the probe chain
  // planned once per (join node, validity patterns)
  std::string key = std::to_string(plan_hash_ ^ (0x9E3779B97F4A7C15ull * (uint64_t)(node_id_[&node] + 1))) + ":J:" + validity_key(L.has_valid) + "|" +
                    validity_key(R.has_valid) + key_suffix;
  std::shared_ptr<PlannedVariant> pv;
  {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    if (it != g_plan_cache.end()) pv = it->second;
  }
  if (!pv) {
    pv = std::make_shared<PlannedVariant>();
    pv->desc = generate_join(j, L.types, R.types, L.has_valid, R.has_valid, fused_probe, fused_build);
    pv->code = jit_compile(pv->desc.source);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plan_cache[key] = pv;
  }
  Variant v;
  v.desc = pv->desc;
  note_sites(v.desc);
  v.mod = jit_load(pv->code);
  const PipelineDesc& d = v.desc;
  const bool build_left = j.build_side == BuildSide::Left;
  const DevTable& B = build_left ? L : R;
  const DevTable& P = build_left ? R : L;
  if (B.rows >= ((int64_t)1 << 31)) throw CometError("hash join build side exceeds 2^31 rows");
  const size_t nb = B.cols.size(), np = P.cols.size();
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  for (size_t i = 0; i < nb; i++) {
    prm.in[i].data = B.cols[i].data;
    prm.in[i].valid = B.has_valid[i] ? B.cols[i].valid : nullptr;
    prm.in[i].aux = B.cols[i].aux;
    prm.in[i].offset = B.cols[i].offset;
  }
  for (size_t i = 0; i < np; i++) {
    prm.in[nb + i].data = P.cols[i].data;
    prm.in[nb + i].valid = P.has_valid[i] ? P.cols[i].valid : nullptr;
    prm.in[nb + i].aux = P.cols[i].aux;
    prm.in[nb + i].offset = P.cols[i].offset;
  }
  const int64_t n = P.rows;
  const bool use_lds = B.rows > 0 && B.rows <= 6144 && getenv("COMET_JOIN_GLOBAL_TABLE") == nullptr;
  DevBuf head, next, matched, btiles, emitted_buf;
  emitted_buf.ensure(64);      // u64 words: [0] emitted rows / leaders, [1] smallest key, [2] largest key, [3] keyed rows, [4] "a key came twice", [6] "a bucket-table partition overflowed"
  HIP_CHECK(hipMemsetAsync(emitted_buf.p, 0, 64, stream_));
  prm.n = n;
  prm.iarg[1] = B.rows;
  // The bucket array holds one entry per RUN of equal neighbouring keys (comet_device.hpp "Runs of equal keys"), not one per row: a large
  // build side is counted first (one coalesced pass over its key columns) so that a clustered fact table — several rows per key — gets
  // a bucket array sized by its runs, which stays in the caches several times better than one sized by its rows.
  int64_t entries = B.rows, keyed_rows = -1;
  static const bool count_runs = getenv("COMET_JOIN_COUNT_RUNS") == nullptr || atoi(getenv("COMET_JOIN_COUNT_RUNS")) != 0;
  DevBuf keymap;
  uint64_t keymap_first = 0, keymap_range = 0;       // a candidate for the key bitmap: one integer key whose values span a foreign key's range
  uint64_t key_first = 0, key_range = 0;             // that key's range whatever its density (the monotone hash of the bucket table; 0: none / unknown)
  // The general table (comet_device.hpp template D''): partitioned build into LDS, 16-byte entries, one random access per probe key.  Small build
  // sides keep the chained table (it sits in L2, and one launch builds it); so does a build side with more runs than 16384 partitions hold.
  static const int64_t bucket_min_rows = getenv("COMET_JOIN_BUCKET_MIN_ROWS") ? atoll(getenv("COMET_JOIN_BUCKET_MIN_ROWS")) : 65536;
  static const int bitmap_only_mode = getenv("COMET_JOIN_BITMAP_ONLY") ? atoi(getenv("COMET_JOIN_BITMAP_ONLY")) : 1;
  bool bucket = !use_lds && !join_no_bucket_ && bucket_min_rows >= 0 && B.rows >= bucket_min_rows;
  if (!use_lds && count_runs && B.rows >= (bucket ? 65536 : (1 << 20))) {
    // leaders, smallest key, largest key (order-preserving u64), rows with a non-NULL key
    HIP_CHECK(hipMemsetAsync((char*)emitted_buf.p + 8, 0xff, 8, stream_));
    prm.out[0] = emitted_buf.p;
    prm.out[kOutErr] = err_flags_.p;
    launch(v, "k_jbcnt", (int)std::min<int64_t>((B.rows + 255) / 256, 256 * 8), prm);
    uint64_t got[4] = {0, 0, 0, 0};
    read_small(got, emitted_buf.p, sizeof got);
    entries = std::min<int64_t>(B.rows, (int64_t)got[0]);
    keyed_rows = (int64_t)got[3];
    if (getenv("COMET_JOIN_DEBUG"))
      fprintf(stderr, "[join %s] build rows %lld: leaders %llu, keyed rows %llu, key min %lld max %lld\n", key_suffix.c_str(), (long long)B.rows, (unsigned long long)got[0],
              (unsigned long long)got[3], (long long)(got[1] ^ ((uint64_t)1 << 63)), (long long)(got[2] ^ ((uint64_t)1 << 63)));
    // The key bitmap (comet_device.hpp kJoinKeyMap): one integer key whose values span at most 64 bits per build row and 2^31 bits — a
    // foreign key's shape.  k_jbcnt leaves min / max untouched when the kernel has no such key (KEYMAP false).
    if (got[1] <= got[2]) {
      const uint64_t range = got[2] - got[1] + 1;      // (never 0: the keys are 64-bit values of ≤ 2^31 rows … guarded below anyway)
      if (range != 0) {
        key_first = got[1] ^ ((uint64_t)1 << 63);
        key_range = range;
      }
      if (range != 0 && range <= ((uint64_t)1 << 31) && range <= (uint64_t)B.rows * 64) {
        keymap_first = got[1] ^ ((uint64_t)1 << 63);
        keymap_range = range;
      }
    }
  }
  // The DIRECT MAP (comet_device.hpp JoinDirectTable): when that key is also UNIQUE — every build row leads its own run, and the bitmap's build
  // pass sees no bit twice — the bitmap, the keys below each of its 128-bit blocks and the build rows in key order replace the hash table.
  static const int direct_mode = getenv("COMET_JOIN_DIRECT") ? atoi(getenv("COMET_JOIN_DIRECT")) : 1;      // 0: never
  bool direct = false, keymap_built = false;
  DevBuf dranks, drows, dtiles, dcounts;
  auto build_keymap = [&](bool want_dup_flag) {
    const size_t words = (size_t)((keymap_range + 127) / 128) * 4;       // whole 128-bit blocks
    keymap.ensure(16 + words * 4 + 16);
    HIP_CHECK(hipMemsetAsync((char*)keymap.p + 16, 0, words * 4 + 16, stream_));
    const uint64_t hdr[2] = {keymap_first, keymap_range};
    write_small(keymap.p, hdr, sizeof hdr);
    HIP_CHECK(hipMemsetAsync((char*)emitted_buf.p + 32, 0, 8, stream_));     // "a key came twice"
    prm.iarg[5] = want_dup_flag ? 1 : 0;                                  // (0: the wave-combined build, which cannot tell)
    prm.out[44] = keymap.p;
    prm.out[47] = emitted_buf.p;
    prm.out[kOutErr] = err_flags_.p;
    launch(v, "k_jbmap", (int)std::min<int64_t>((B.rows + 255) / 256, 256 * 8), prm);
    join_keymap_bytes_ += (int64_t)words * 4;
    keymap_built = true;
  };
  // A semi / anti join that only asks whether the key exists (no residual, probe rows kept) over such a key: the bitmap is the whole build side.
  const bool bitmap_only = keymap_range != 0 && d.join_dedup_build && bitmap_only_mode != 0 && n > 0;
  if (bitmap_only) {
    build_keymap(false);
    bucket = false;
    join_bitmap_only_++;
  } else if (keymap_range && direct_mode != 0 && entries == keyed_rows) {       // (no two neighbouring rows share a key: a clustered fact table is spared the pass)
    build_keymap(false);      // (the wave-combined build: duplicates show as missing bits, counted below)
    // ranks and rows[] are built right behind the bitmap, BEFORE the host knows whether a key came twice (one wait instead of three; with a duplicate the
    // two small passes were for nothing — their writes stay inside their buffers either way)
    {
      const int64_t nblocks = (int64_t)((keymap_range + 127) / 128);
      dcounts.ensure((size_t)nblocks * 4 + 16);
      dranks.ensure((size_t)(nblocks + 2) * 4);
      dtiles.ensure((size_t)((nblocks + 1023) / 1024 + 2) * 8);
      drows.ensure((size_t)B.rows * 4 + 16);
      if (comet_launch_popcount128((const char*)keymap.p + 16, nblocks, (uint32_t*)dcounts.p, stream_) != 0) throw CometError("hash join: launch failed");
      pq_launch_u32_scan((const uint32_t*)dcounts.p, nblocks, (uint64_t*)dtiles.p, (int32_t*)dranks.p, stream_);
      prm.out[0] = dranks.p;
      prm.out[1] = drows.p;
      prm.iarg[6] = keyed_rows;       // k_jdrows compares the bitmap's bit count with it: fewer bits = a key came twice
      launch(v, "k_jdrows", (int)std::min<int64_t>((B.rows + 255) / 256, 256 * 8), prm);
    }
    uint64_t dup = 1;
    read_small(&dup, (char*)emitted_buf.p + 32, 8);
    if (!dup) {
      direct = true;
      join_direct_maps_++;
    }
  }
  // ---- the bucket table: partition passes, LDS build ----
  DevBuf btab, brecs, bpart;
  int64_t bslots = 0;
  if (direct || bitmap_only || B.rows == 0) bucket = false;
  if (bucket) {
    constexpr int64_t kS = 4096, kMaxP = 16384;            // comet_device.hpp kJoinPartSlots, kJoinPartMax
    const int64_t np = std::max<int64_t>(1, (std::max<int64_t>(entries, 1) + kS / 2 - 1) / (kS / 2));
    if (np > kMaxP) bucket = false;
    else {
      const int64_t g = std::min<int64_t>(256, (B.rows + 4095) / 4096);
      const int64_t chunk = ((B.rows + g - 1) / g + 1023) / 1024 * 1024;
      bslots = np * kS;
      btab.ensure((size_t)bslots * 16);
      brecs.ensure((size_t)std::max<int64_t>(entries, 1) * 16 + 16);
      bpart.ensure((size_t)(2 * kMaxP + 16 + g * np) * 4 + 16);
      prm.iarg[0] = bslots;
      prm.iarg[2] = np;
      prm.iarg[5] = chunk;
      prm.out[0] = btab.p;
      prm.out[1] = brecs.p;
      prm.out[3] = bpart.p;
      prm.out[47] = emitted_buf.p;
      prm.out[kOutErr] = err_flags_.p;
      // The monotone hash (comet_device.hpp join_mono_hash) where the key is one integer with a known range: slots in key order.  Whether the keys are spread
      // evenly enough is asked BEFORE the table is built (a partition above 7/8 of its slots: the scan kernel's flag, one word read back) — if not, the
      // scrambling hash, whose own overflow (many separate runs of one key) is only read with the probe's result.
      static const int mono_mode = getenv("COMET_JOIN_MONO") ? atoi(getenv("COMET_JOIN_MONO")) : 1;
      bool mono = mono_mode != 0 && key_range != 0;
      for (int attempt = 0;; attempt++) {
        const uint64_t mp[3] = {mono ? (~(uint64_t)0) / key_range : 0, key_first, key_range};      // (⌊(2^64 − 1) / range⌋ ≤ ⌊2^64 / range⌋: still injective, still below all ones)
        write_small((uint32_t*)bpart.p + (kMaxP + 4), mp, sizeof mp);
        launch(v, "k_jphist", (int)g, prm, 1024);
        if (comet_launch_join_part_scan((uint32_t*)bpart.p + (2 * kMaxP + 16), (int)g, (int)np, (uint32_t*)bpart.p + (kMaxP + 16), (uint32_t)(kS - kS / 8),
                                        (uint64_t*)emitted_buf.p + 6, stream_) != 0)
          throw CometError("hash join: launch failed");
        if (!mono) break;
        uint64_t over = 0;
        read_small(&over, (char*)emitted_buf.p + 48, 8);
        if (!over) break;
        mono = false;
        HIP_CHECK(hipMemsetAsync((char*)emitted_buf.p + 48, 0, 8, stream_));
      }
      launch(v, "k_jpscat", (int)g, prm, 1024);
      launch(v, "k_jtbuild", (int)np, prm);
      join_bucket_tables_++;
      if (mono) join_mono_tables_++;
    }
  }
  int64_t cap = 1024;
  if (!direct && !bucket && !bitmap_only) {
    while (cap < 2 * entries) cap <<= 1;
    head.ensure((size_t)cap * 4);     // u32 per bucket: newest run leader | tag | "more than one row" flag (comet_device.hpp template D)
    next.ensure((size_t)std::max<int64_t>(B.rows, 1) * 4 + 16);
    HIP_CHECK(hipMemsetAsync(head.p, 0xff, (size_t)cap * 4, stream_));
  }
  const bool outer_build = d.join_outer_build;
  const int64_t nbtiles = (B.rows + 1023) / 1024;
  if (outer_build) {
    matched.ensure((size_t)std::max<int64_t>(B.rows, 1));
    btiles.ensure((size_t)(nbtiles + 1) * 8);
    HIP_CHECK(hipMemsetAsync(matched.p, 0, (size_t)std::max<int64_t>(B.rows, 1), stream_));
    prm.out[45] = matched.p;
    prm.out[46] = btiles.p;
    prm.iarg[3] = nbtiles;
  }
  if (!bucket) prm.iarg[0] = cap;
  if (!bucket) {
    int ib = 1;                                   // bits of a build row index: rows ≤ 2^ib − 1, so an index is never all ones
    while (ib < 31 && ((int64_t)1 << ib) - 1 < std::max<int64_t>(B.rows, 1)) ib++;
    if (((int64_t)1 << ib) - 1 < B.rows) throw CometError("HashJoin: build sides of 2^31 rows or more are not supported");
    prm.iarg[2] = ib;
  }
  if (!direct && !bucket && !bitmap_only) {
    prm.out[0] = head.p;
    prm.out[1] = next.p;
  }
  prm.out[kOutErr] = err_flags_.p;
  timed_begin();
  int64_t out_rows = 0, tail_rows = 0;
  const size_t ncol = d.out_cols.size();
  std::vector<std::shared_ptr<DevBuf>> vals(ncol), vbytes(ncol);
  auto bind_outputs = [&](int64_t rows_cap) {
    for (size_t c = 0; c < ncol; c++) {
      if (!vals[c]) vals[c] = std::make_shared<DevBuf>();
      vals[c]->ensure((size_t)std::max<int64_t>(rows_cap, 1) * out_width(d.out_cols[c]) + 16);
      prm.out[kOutFirstCol + 2 * c] = vals[c]->p;
      if (!vbytes[c]) vbytes[c] = std::make_shared<DevBuf>();
      if (d.out_cols[c].nullable) {
        vbytes[c]->ensure((size_t)std::max<int64_t>(rows_cap, 1) + 16);
        prm.out[kOutFirstCol + 2 * c + 1] = vbytes[c]->p;
      }
    }
  };
  // ---- single-pass probe (comet_device.hpp template D'): a small build side is hashed into LDS by every block, a large one into the
  // chained global table; either way the probe counts and emits in one launch, reserving output ranges with one atomic per tile ----
  if (!use_lds && !direct && !bucket && !bitmap_only && B.rows) launch(v, "k_jbuild", (int)std::min<int64_t>((B.rows + 255) / 256, 256 * 8), prm);
  prm.out[47] = emitted_buf.p;
  // The key bitmap (comet_device.hpp kJoinKeyMap) pays when probe rows miss: 8192 probe rows, evenly spaced, go through the finished table
  // first; fewer than half with a partner → one more pass over the build keys sets the bits, and the probe asks them before the table.
  static const int keymap_mode = getenv("COMET_JOIN_KEYMAP") ? atoi(getenv("COMET_JOIN_KEYMAP")) : -1;      // 0 never, 1 always, default: by the sample
  // (… and only where the probe side is several times the build side: the bitmap's build pass costs what the build side's atomics cost —
  // 2.2 ms for each of TPC-DS Q95's 70 M-row builds, whose equally large probe sides it did not speed up)
  if (bitmap_only) {
    // (the bitmap is the table)
  } else if (keymap_built && !direct) {
    // (the bitmap exists already — the build side turned out not to be unique — and filters the probe rows as below)
  } else if (!direct && keymap_range && keymap_mode != 0 && n >= (1 << 20) && (keymap_mode == 1 || n >= 4 * B.rows)) {
    bool wanted = keymap_mode == 1;
    if (!wanted) {
      HIP_CHECK(hipMemsetAsync(emitted_buf.p, 0, 16, stream_));
      const int64_t ns = 8192;
      prm.iarg[5] = ns;
      launch(v, bucket ? "k_jsample_b" : "k_jsample", (int)((ns + 255) / 256), prm);
      uint64_t cnt[2] = {0, 0};
      read_small(cnt, emitted_buf.p, sizeof cnt);
      wanted = cnt[0] >= 64 && cnt[1] * 2 < cnt[0];
    }
    if (wanted) build_keymap(false);
  }
  // FK-shaped joins emit at most one row per probe row; anything beyond the capacity is counted, not written, and the probe re-run
  int64_t out_cap = d.join_build_only ? 1 : n + 1024;
  for (int attempt = 0; n > 0; attempt++) {
    bind_outputs(out_cap);
    prm.iarg[6] = out_cap;
    HIP_CHECK(hipMemsetAsync(emitted_buf.p, 0, 8, stream_));
    const int64_t ptiles = (n + 4095) / 4096;      // kJoinR0 × 256 probe rows per tile (comet_device.hpp)
    // (k_jprobe_km: the table probe that asks the key bitmap first; without a bitmap the leaner k_jprobe — comet_device.hpp join_probe_tiles)
    launch(v, use_lds ? "k_jlds" : bitmap_only ? "k_jprobe_bm" : direct ? "k_jdprobe" : bucket ? (keymap_built ? "k_jprobe_bkm" : "k_jprobe_b") : keymap_built ? "k_jprobe_km" : "k_jprobe",
           (int)std::min<int64_t>(ptiles, use_lds ? 256 * 3 : 256 * 8), prm);
    uint64_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    read_small(words, emitted_buf.p, sizeof words);
    const uint64_t emitted = words[0];
    if (bucket && words[6]) {
      // a partition of the bucket table overflowed (many separate runs of one key): this join runs over the chained table
      timed_end();
      HIP_CHECK(hipStreamSynchronize(stream_));
      join_bucket_tables_--;
      join_no_bucket_ = true;
      try {
        DevTable r = hash_join_impl(node, j, L, R, key_suffix, fused_probe, fused_build);
        join_no_bucket_ = false;
        return r;
      } catch (...) {
        join_no_bucket_ = false;
        throw;
      }
    }
    out_rows = d.join_build_only ? 0 : (int64_t)emitted;
    if (out_rows <= out_cap) break;
    if (attempt == 1) throw CometError("internal: hash join output exceeded its exact size");
    out_cap = out_rows;
  }
  const int64_t probe_capacity = n > 0 ? out_cap : 0;
  if (outer_build && B.rows > 0) {
    // build rows without a match follow the probe-driven rows
    launch(v, "k_jbcount", (int)std::min<int64_t>(nbtiles, 256 * 8), prm);
    launch(v, "k_jbscan", 1, prm);
    uint64_t total = 0;
    read_small(&total, (char*)btiles.p + (size_t)nbtiles * 8, 8);
    tail_rows = (int64_t)total;
    prm.iarg[4] = out_rows;
  }
  const int64_t all_rows = out_rows + tail_rows;
  if (all_rows > probe_capacity || (ncol > 0 && !vals[0])) {
    // the unmatched build rows follow the probe-driven rows: grow the output buffers, keeping what the probe wrote
    std::vector<std::shared_ptr<DevBuf>> ov = vals, ob = vbytes;
    for (size_t c = 0; c < ncol; c++) { vals[c].reset(); vbytes[c].reset(); }
    bind_outputs(all_rows);
    for (size_t c = 0; c < ncol && out_rows > 0; c++) {
      HIP_CHECK(hipMemcpyAsync(vals[c]->p, ov[c]->p, (size_t)out_rows * out_width(d.out_cols[c]), hipMemcpyDeviceToDevice, stream_));
      if (d.out_cols[c].nullable) HIP_CHECK(hipMemcpyAsync(vbytes[c]->p, ob[c]->p, (size_t)out_rows, hipMemcpyDeviceToDevice, stream_));
    }
    HIP_CHECK(hipStreamSynchronize(stream_));   // the old buffers return to the pool
  }
  if (tail_rows > 0) launch(v, "k_jbemit", (int)std::min<int64_t>(nbtiles, 256 * 8), prm);
  timed_end();
  out_rows = all_rows;
  // gathered Utf8 payload columns name their source by its kernel-argument index: build columns, then probe columns
  const int nbuild = (int)nb;
  DevTable out = outputs_to_table(v, vals, vbytes, out_rows, [&](int c) { return c < nbuild ? std::make_pair(&B, c) : std::make_pair(&P, c - nbuild); });
  if (fused_probe || fused_build) check_device_errors();     // the chain's expressions may raise ANSI errors; an unfused chain checks after its own launch
  // the tables of this frame (heads, records, bitmap, ranks …) go back to the pool when it ends: nothing queued may still read them.  The probe's result was
  // waited for (read_small) and only the output's own buffers are touched after it — except by the build-side tail of an outer join, and when no probe ran
  if (tail_rows > 0 || n == 0) HIP_CHECK(hipStreamSynchronize(stream_));
  out.owners.push_back(v.mod);
  join_build_rows_ += B.rows;
  join_probe_rows_ += P.rows;
  return out;
}

}  // namespace comet
