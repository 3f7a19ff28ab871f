// ShuffleWriter: partition on the device, frame and write blocks on the host.
#include "exec_internal.hpp"

namespace comet {
// ShuffleWriter (native/shuffle/src/shuffle_writer.rs:166-300, partitioners/multi_partition.rs:265-457, single_partition.rs):
// the child's whole output is resident in HBM; partition ids (Spark murmur3 seed 42 chained over the hash expressions → pmod),
// the stable per-partition row order and the per-column gathers all run on the GPU (the exchange kernels), ONE download brings the
// partition-major table to pinned host memory, and the host threads frame it: per partition, blocks of ≤ batch_size rows in input
// order (partitioned_batch_iterator.rs:100-124), each an Arrow IPC stream behind the 20-byte header, codec applied per block
// (shuffle_block_writer.rs:179-238).  Data file = partitions back to back; index file = num_partitions + 1 little-endian i64 offsets
// (writers/local/local_partition_writer.rs:255-295).
DevTable ExecutionContext::write_shuffle(const Operator& sw) {
  static const bool trace = getenv("COMET_TRACE_STAGES") != nullptr;
  Timer tm;
  double t_last = 0;
  auto lap = [&](const char* what) {
    if (!trace) return;
    const double now = tm.ns() / 1e6;
    fprintf(stderr, "[comet] shuffle write: %s %.3f ms\n", what, now - t_last);
    t_last = now;
  };
  auto sp = shuffle_projs_.find(&sw);
  const Operator& child = sp != shuffle_projs_.end() ? *sp->second : *sw.children[0];
  DevTable in = materialize(child);
  const int64_t n = in.rows;
  lap("child");
  const int P = sw.shuffle_partitioning == Operator::Partitioning::Single ? 1 : sw.shuffle_num_partitions;
  size_t n_payload = in.cols.size();
  std::vector<int> key_cols;
  {
    size_t appended = 0;   // computed key expressions sit behind the payload columns (the synthetic projection of the constructor)
    for (auto& e : sw.shuffle_hash_exprs) appended += e->kind != ExprKind::Bound;
    for (auto& k : sw.shuffle_sort_orders) appended += k.child->kind != ExprKind::Bound;
    n_payload -= appended;
  }
  if (sw.shuffle_partitioning == Operator::Partitioning::Hash) {
    size_t next = n_payload;
    for (auto& e : sw.shuffle_hash_exprs) key_cols.push_back(e->kind == ExprKind::Bound ? e->bound_index : (int)next++);
  } else if (sw.shuffle_partitioning == Operator::Partitioning::RoundRobin) {
    // "round robin" = hash of the first max_hash_columns columns (multi_partition.rs:386-437)
    const size_t k = sw.shuffle_max_hash_columns <= 0 ? n_payload : std::min<size_t>((size_t)sw.shuffle_max_hash_columns, n_payload);
    for (size_t i = 0; i < k; i++)
      if (!in.types[i].is_nested()) key_cols.push_back((int)i);      // (nested columns are not hashed here: any assignment is a round robin)
  }
  if (n >= (int64_t)1 << 32) throw CometError("ShuffleWriter: more than 2^32 rows in one task are not supported (u32 row indices, multi_partition.rs)");
  std::vector<int64_t> starts((size_t)P + 1, 0);
  starts[(size_t)P] = n;
  DevTable grouped;
  if (P > 1 && n > 0) {
    DevBuf hashes, pids, dstarts, hist;
    auto ridx = std::make_shared<DevBuf>();
    hashes.ensure((size_t)n * 4);
    pids.ensure((size_t)n * 4);
    ridx->ensure((size_t)n * 4 + 16);
    dstarts.ensure(((size_t)P + 1) * 8);
    const bool by_range = sw.shuffle_partitioning == Operator::Partitioning::Range;
    std::shared_ptr<DevBuf> planes;
    DevBuf bkeys;
    if (by_range) {
      // order-preserving key bytes of every row and of every boundary row (same generated kernel, same widths), then an
      // upper-bound search per row: partition = number of boundaries ≤ row (multi_partition.rs:352-358)
      int W = 0, Wb = 0;
      const int B = (int)sw.shuffle_bounds.size();
      std::vector<DType> btypes;
      for (auto& k : range_sort_.at(&sw)->sort_orders) btypes.push_back(k.child->dtype);
      DevTable bt = literal_table(sw.shuffle_bounds, btypes);
      // Utf8 keys: rows and boundaries must be padded to the same length — the longer of the two
      std::vector<int64_t> lr, lb;
      {
        int w0 = 0;
        (void)sort_key_planes(*range_sort_.at(&sw), in, w0, &lr, true);              // measure only
        if (B > 0) (void)sort_key_planes(*range_bsort_.at(&sw), bt, w0, &lb, true);
        for (size_t s = 0; s < lr.size(); s++) lr[s] = std::max<int64_t>(lr[s], s < lb.size() ? lb[s] : 0);
        lb = lr;
      }
      planes = sort_key_planes(*range_sort_.at(&sw), in, W, &lr);
      std::vector<uint8_t> rowmajor((size_t)std::max(B, 1) * (size_t)std::max(W, 1), 0);
      if (B > 0) {
        auto bplanes = sort_key_planes(*range_bsort_.at(&sw), bt, Wb, &lb);
        if (Wb != W) throw CometError("internal: range boundary keys and row keys differ in width");
        std::vector<uint8_t> pl((size_t)W * (size_t)B);
        HIP_CHECK(hipMemcpyAsync(pl.data(), bplanes->p, pl.size(), hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipStreamSynchronize(stream_));
        for (int b = 0; b < B; b++)
          for (int p = 0; p < W; p++) rowmajor[(size_t)b * W + p] = pl[(size_t)p * B + b];
        for (int b = 1; b < B; b++)
          if (memcmp(&rowmajor[(size_t)(b - 1) * W], &rowmajor[(size_t)b * W], (size_t)W) > 0) throw CometError("ShuffleWriter: range boundaries are not in ascending order");
      }
      bkeys.ensure(rowmajor.size() + 16);
      HIP_CHECK(hipMemcpyAsync(bkeys.p, rowmajor.data(), rowmajor.size(), hipMemcpyHostToDevice, stream_));
      if (comet_launch_range_partition_ids((const uint8_t*)planes->p, n, W, (const uint8_t*)bkeys.p, B, (int32_t*)pids.p, stream_) != 0)
        throw CometError("shuffle: launch failed");
      HIP_CHECK(hipStreamSynchronize(stream_));   // rowmajor (pageable) must outlive the upload
    }
    const uint32_t seed = 42;
    if (!by_range && comet_launch_fill(4, hashes.p, n, &seed, stream_) != 0) throw CometError("shuffle: launch failed");
    for (int c : key_cols) {
      const DeviceColumnView& v = in.cols[(size_t)c];
      if (v.offset != 0) throw CometError("ShuffleWriter: hash key column with a non-zero Arrow offset is not supported yet");
      if (comet_launch_murmur3((int)in.types[(size_t)c].id, in.types[(size_t)c].precision, v.data, in.has_valid[(size_t)c] ? v.valid : nullptr, v.aux, n,
                               (uint32_t*)hashes.p, stream_) != 0)
        throw CometError("ShuffleWriter: cannot hash a column of type " + in.types[(size_t)c].str());
    }
    const int64_t W = comet_partition_tiles(n);
    const size_t hist_bytes = ((size_t)P * (size_t)W + 1) * 8;
    hist.ensure((size_t)comet_partition_scratch_bytes(n, P));
    uint32_t* bad = (uint32_t*)((char*)hist.p + hist_bytes);
    HIP_CHECK(hipMemsetAsync(bad, 0, 4, stream_));
    if ((!by_range && comet_launch_pmod((const uint32_t*)hashes.p, n, P, (int32_t*)pids.p, stream_) != 0) ||
        comet_launch_partition_indices((const int32_t*)pids.p, n, P, (uint64_t*)hist.p, bad, (int64_t*)dstarts.p, (uint32_t*)ridx->p, stream_) != 0)
      throw CometError("shuffle: launch failed");
    HIP_CHECK(hipMemcpyAsync(starts.data(), dstarts.p, ((size_t)P + 1) * 8, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
    DevTable payload = in;
    payload.types.resize(n_payload);
    payload.cols.resize(n_payload);
    payload.has_valid.resize(n_payload);
    grouped = take_rows(payload, (const uint32_t*)ridx->p, 0, n, ridx);
  } else {
    grouped = in;
    grouped.types.resize(n_payload);
    grouped.cols.resize(n_payload);
    grouped.has_valid.resize(n_payload);
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  shuffle_repart_ns_ += tm.ns();
  lap("partition (murmur3, pmod, indices, takes)");

  // The partition-major table stays in HBM; it crosses to the host in SLABS of consecutive blocks (≈ staging_bytes of column data, two
  // pinned staging sets: slab k+1 is downloading while slab k is framed and written), so the writer's host footprint is bounded by the
  // staging size whatever the task's output — the role spilling plays in the reference's writer (multi_partition.rs:439-457), without
  // the temporary files: partitions are contiguous row ranges here, so the data file is written front to back in one pass.
  const int64_t bs = batch_size_ > 0 ? batch_size_ : std::max<int64_t>(n, 1);
  const ShuffleCodec codec = (ShuffleCodec)sw.shuffle_codec;
  size_t staging_bytes = (size_t)256 << 20;
  for (auto& kv : config_)
    if (kv.first == "spark.comet.gpu.shuffle.stagingBytes") staging_bytes = (size_t)std::max<long long>(1 << 16, atoll(kv.second.c_str()));
  if (const char* e = getenv("COMET_SHUFFLE_STAGING_BYTES")) staging_bytes = (size_t)std::max<long long>(1 << 16, atoll(e));
  size_t row_bytes = 0;
  std::vector<char> is_str(n_payload, 0);
  // Nested columns (structs, lists) come to the host WHOLE, once — their buffers hang off each other by offsets, which a slab of rows does not
  // cut cleanly; the flat columns next to them keep the slabs.  (A task whose nested columns outgrow host memory is not bounded by the staging
  // size: the one place the writer's footprint follows the data.)
  std::vector<HostColumn> nested_host(n_payload);
  std::vector<ColumnSlice> nested_slice(n_payload);
  std::function<void(const HostColumn&, ColumnSlice&)> slice_of = [&](const HostColumn& h, ColumnSlice& c) {
    c.type = h.type;
    c.validity = h.validity.empty() ? nullptr : h.validity.data();
    c.values = h.values.empty() ? nullptr : h.values.data();
    c.data = h.data.empty() ? nullptr : h.data.data();
    c.first = 0;
    c.kids.resize(h.children.size());
    for (size_t k = 0; k < h.children.size(); k++) slice_of(h.children[k], c.kids[k]);
  };
  for (size_t j = 0; j < n_payload; j++) {
    if (!grouped.types[j].is_nested()) continue;
    nested_host[j] = download_column(grouped.cols[j], grouped.types[j], grouped.has_valid[j], n, stream_);
    slice_of(nested_host[j], nested_slice[j]);
    row_bytes += 32;
  }
  for (size_t j = 0; j < n_payload; j++) {
    const DType& ty = grouped.types[j];
    const DeviceColumnView& v = grouped.cols[j];
    if (ty.is_nested()) continue;
    if (n > 0 && v.offset != 0) throw CometError("ShuffleWriter: input column with a non-zero Arrow offset is not supported yet");
    is_str[j] = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    if (is_str[j]) {
      int32_t ends[2] = {0, 0};
      if (n > 0) {
        read_small(&ends[0], v.data, 4);
        read_small(&ends[1], (const char*)v.data + (size_t)n * 4, 4);
        if (ends[0] != 0) throw CometError("ShuffleWriter: Utf8 column whose offsets do not start at 0");
      }
      row_bytes += 4 + (n > 0 ? (size_t)(ends[1] / n) + 1 : 0);
    } else {
      row_bytes += ty.id == TypeId::Bool ? 1 : (size_t)fixed_width(ty);
    }
    if (grouped.has_valid[j]) row_bytes += 1;
  }
  const double write_t0 = tm.ns();
  // Blocks in file order — (partition, first row, rows) — grouped into runs of consecutive blocks of ≈4 MiB of column data (the unit a
  // scan thread encodes into one output buffer), runs grouped into slabs.
  struct BlockTask { int p; int64_t first, rows; };
  std::vector<BlockTask> tasks;
  for (int p = 0; p < P; p++)
    for (int64_t r = starts[(size_t)p]; r < starts[(size_t)p + 1]; r += bs) tasks.push_back({p, r, std::min(bs, starts[(size_t)p + 1] - r)});
  struct Run {
    size_t first = 0, last = 0;      // tasks [first, last)
    std::vector<uint8_t> bytes;
    std::vector<size_t> block_size;  // per task
    std::string error;
    bool done = false;
    double encode_ms = 0;
  };
  struct Slab {
    size_t run0 = 0, run1 = 0;       // runs [run0, run1)
    int64_t base = 0, end = 0;       // rows [base, end) are staged; base is a multiple of 8 (bitmaps are copied from a byte boundary)
  };
  std::vector<Run> runs;
  std::vector<Slab> slabs;
  {
    const size_t rb = std::max<size_t>(row_bytes, 1);
    const size_t run_target = std::min<size_t>((size_t)4 << 20, std::max<size_t>(staging_bytes / 64, (size_t)64 << 10));   // ≥ 64 runs per slab keep every scan thread busy
    size_t slab_acc = 0;
    for (size_t t = 0; t < tasks.size();) {
      size_t e = t, acc = 0;
      while (e < tasks.size() && acc < run_target) acc += (size_t)tasks[e++].rows * rb;
      Run r;
      r.first = t;
      r.last = e;
      if (slabs.empty() || slab_acc + acc > staging_bytes) {
        Slab sl;
        sl.run0 = runs.size();
        sl.base = tasks[t].first & ~(int64_t)7;
        slabs.push_back(sl);
        slab_acc = 0;
      }
      slab_acc += acc;
      runs.push_back(std::move(r));
      slabs.back().run1 = runs.size();
      slabs.back().end = tasks[e - 1].first + tasks[e - 1].rows;
      t = e;
    }
  }
  // Utf8 columns: the byte range of every slab (offsets at its first and last row), fetched up front so a slab is ONE group of copies
  std::vector<std::vector<int32_t>> str_lo(n_payload), str_hi(n_payload);
  {
    PinnedBuf hb;
    size_t n_str = 0;
    for (size_t j = 0; j < n_payload; j++) n_str += is_str[j];
    hb.ensure(n_str * slabs.size() * 8 + 16);
    size_t k = 0;
    for (size_t j = 0; j < n_payload; j++) {
      if (!is_str[j]) continue;
      for (auto& sl : slabs) {
        HIP_CHECK(hipMemcpyAsync((char*)hb.p + k * 4, (const char*)grouped.cols[j].data + (size_t)sl.base * 4, 4, hipMemcpyDeviceToHost, stream_));
        HIP_CHECK(hipMemcpyAsync((char*)hb.p + k * 4 + 4, (const char*)grouped.cols[j].data + (size_t)sl.end * 4, 4, hipMemcpyDeviceToHost, stream_));
        k += 2;
      }
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    k = 0;
    for (size_t j = 0; j < n_payload; j++) {
      if (!is_str[j]) continue;
      for (size_t q = 0; q < slabs.size(); q++, k += 2) {
        str_lo[j].push_back(((const int32_t*)hb.p)[k]);
        str_hi[j].push_back(((const int32_t*)hb.p)[k + 1]);
      }
    }
  }
  struct Staging {
    std::vector<std::unique_ptr<PinnedBuf>> hv, hb, hd;
    hipEvent_t ready = nullptr;
  };
  Staging stage[2];
  for (auto& sg : stage) {
    sg.hv.resize(n_payload); sg.hb.resize(n_payload); sg.hd.resize(n_payload);
    HIP_CHECK(hipEventCreateWithFlags(&sg.ready, hipEventDisableTiming));
  }
  struct EventGuard { Staging* s; ~EventGuard() { for (int i = 0; i < 2; i++) if (s[i].ready) (void)hipEventDestroy(s[i].ready); } } event_guard{stage};
  size_t staged_peak = 0;
  auto download = [&](size_t q) {
    const Slab& sl = slabs[q];
    Staging& sg = stage[q & 1];
    const int64_t rows = sl.end - sl.base;
    size_t total = 0;
    for (size_t j = 0; j < n_payload; j++) {
      const DType& ty = grouped.types[j];
      const DeviceColumnView& v = grouped.cols[j];
      if (ty.is_nested()) continue;
      size_t skip, bytes;
      if (is_str[j]) { skip = (size_t)sl.base * 4; bytes = (size_t)(rows + 1) * 4; }
      else if (ty.id == TypeId::Bool) { skip = (size_t)(sl.base / 8); bytes = (size_t)((rows + 7) / 8); }
      else { skip = (size_t)sl.base * (size_t)fixed_width(ty); bytes = (size_t)rows * (size_t)fixed_width(ty); }
      if (!sg.hv[j]) sg.hv[j].reset(new PinnedBuf());
      sg.hv[j]->ensure(bytes + 8);
      HIP_CHECK(hipMemcpyAsync(sg.hv[j]->p, (const char*)v.data + skip, bytes, hipMemcpyDeviceToHost, stream_));
      total += bytes;
      if (grouped.has_valid[j]) {
        if (!sg.hb[j]) sg.hb[j].reset(new PinnedBuf());
        sg.hb[j]->ensure((size_t)((rows + 7) / 8) + 8);
        HIP_CHECK(hipMemcpyAsync(sg.hb[j]->p, (const char*)v.valid + (size_t)(sl.base / 8), (size_t)((rows + 7) / 8), hipMemcpyDeviceToHost, stream_));
        total += (size_t)((rows + 7) / 8);
      }
      if (is_str[j]) {
        const size_t nb = (size_t)(str_hi[j][q] - str_lo[j][q]);
        if (!sg.hd[j]) sg.hd[j].reset(new PinnedBuf());
        sg.hd[j]->ensure(nb + 8);
        if (nb) HIP_CHECK(hipMemcpyAsync(sg.hd[j]->p, (const char*)v.aux + (size_t)str_lo[j][q], nb, hipMemcpyDeviceToHost, stream_));
        total += nb;
      }
    }
    HIP_CHECK(hipEventRecord(sg.ready, stream_));
    staged_peak = std::max(staged_peak, total);
  };
  std::mutex mu;
  std::condition_variable cv;
  auto submit = [&](size_t q) {      // the slab's copies are complete: its runs go to the scan threads
    const Slab& sl = slabs[q];
    const Staging* sg = &stage[q & 1];
    for (size_t ri = sl.run0; ri < sl.run1; ri++) {
      scan_pool_submit([&, ri, sg, q, base = sl.base]() {
        Run& r = runs[ri];
        Timer rt;
        try {
          size_t est = 0;
          for (size_t t = r.first; t < r.last; t++) est += (size_t)tasks[t].rows * row_bytes + 2048;
          r.bytes.reserve(est + est / 8 + (64 << 10));
          std::vector<ColumnSlice> cols(n_payload);
          for (size_t j = 0; j < n_payload; j++) {
            if (grouped.types[j].is_nested()) { cols[j] = nested_slice[j]; continue; }
            cols[j].type = grouped.types[j];
            cols[j].validity = sg->hb[j] ? (const uint8_t*)sg->hb[j]->p : nullptr;
            cols[j].values = sg->hv[j]->p;
            cols[j].data = sg->hd[j] ? (const uint8_t*)sg->hd[j]->p : nullptr;
            cols[j].data_origin = is_str[j] ? str_lo[j][q] : 0;
          }
          for (size_t t = r.first; t < r.last; t++) {
            for (auto& c : cols) c.first = c.type.is_nested() ? tasks[t].first : tasks[t].first - base;      // (nested columns are addressed from the task's row 0)
            r.block_size.push_back(encode_shuffle_block(cols, tasks[t].rows, codec, sw.shuffle_compression_level, r.bytes));
          }
        } catch (const std::exception& e) {
          r.error = e.what();
        } catch (...) {
          r.error = "shuffle writer: unknown error while encoding a block";
        }
        r.encode_ms = rt.ns() / 1e6;
        {
          std::lock_guard<std::mutex> lk(mu);
          r.done = true;
        }
        cv.notify_all();
      });
    }
  };
  const int fd = open(sw.shuffle_data_file.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  std::string failure;
  if (fd < 0) failure = "shuffle write error: cannot create " + sw.shuffle_data_file + ": " + strerror(errno);
  std::vector<int64_t> offsets((size_t)P + 1, 0);
  int64_t file_pos = 0;
  int next_p = 0;
  double wait_ms = 0, write_ms = 0, enc_sum = 0, enc_max = 0, copy_wait_ms = 0;
  size_t submitted = 0;                   // slabs whose runs have been handed to the scan threads (every one of them must be waited for)
  auto wait_runs = [&](size_t q) {        // write slab q's runs to the data file in order, as they finish
    for (size_t ri = slabs[q].run0; ri < slabs[q].run1; ri++) {
      Run& r = runs[ri];
      {
        Timer wt;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return r.done; });
        wait_ms += wt.ns() / 1e6;
      }
      enc_sum += r.encode_ms;
      enc_max = std::max(enc_max, r.encode_ms);
      Timer wrt;
      if (failure.empty() && !r.error.empty()) failure = r.error;
      if (!failure.empty()) continue;
      int64_t pos = file_pos;
      for (size_t t = r.first; t < r.last; t++) {
        while (next_p <= tasks[t].p) offsets[(size_t)next_p++] = pos;
        pos += (int64_t)r.block_size[t - r.first];
      }
      size_t done = 0;
      while (done < r.bytes.size()) {
        const ssize_t w = write(fd, r.bytes.data() + done, r.bytes.size() - done);
        if (w <= 0) {
          failure = "shuffle write error: " + std::string(strerror(errno)) + " (" + sw.shuffle_data_file + ")";
          break;
        }
        done += (size_t)w;
      }
      file_pos = pos;
      std::vector<uint8_t>().swap(r.bytes);
      write_ms += wrt.ns() / 1e6;
    }
  };
  try {
    if (!slabs.empty() && failure.empty()) download(0);
    for (size_t q = 0; q < slabs.size() && failure.empty(); q++) {
      {
        Timer ct;
        HIP_CHECK(hipEventSynchronize(stage[q & 1].ready));
        copy_wait_ms += ct.ns() / 1e6;
      }
      submit(q);
      submitted = q + 1;
      if (q >= 1) wait_runs(q - 1);                      // staging set (q+1)&1 is free again …
      if (q + 1 < slabs.size() && failure.empty()) download(q + 1);   // … and refilled while slab q is being framed
    }
    if (submitted) wait_runs(submitted - 1);
  } catch (...) {
    // the scan threads reference this frame: every submitted run is waited for before the error leaves
    for (size_t q = 0; q < submitted; q++)
      for (size_t ri = slabs[q].run0; ri < slabs[q].run1; ri++) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return runs[ri].done; });
      }
    (void)hipStreamSynchronize(stream_);
    if (fd >= 0) close(fd);
    throw;
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  if (trace)
    fprintf(stderr, "[comet] shuffle write: %zu slabs (≤ %.1f MiB staged), %zu runs, encode cpu %.1f ms total (max %.2f ms/run), writer waited %.1f ms for runs, "
            "%.1f ms for copies, wrote for %.1f ms\n", slabs.size(), staged_peak / 1048576.0, runs.size(), enc_sum, enc_max, wait_ms, copy_wait_ms, write_ms);
  shuffle_staged_peak_ = std::max<int64_t>(shuffle_staged_peak_, (int64_t)staged_peak);
  while (next_p <= P) offsets[(size_t)next_p++] = file_pos;
  if (fd >= 0 && close(fd) != 0 && failure.empty()) failure = "shuffle write error: closing " + sw.shuffle_data_file + " failed";
  if (!failure.empty()) throw CometError(failure);
  lap("encode blocks + write data file (overlapped)");
  FILE* xf = fopen(sw.shuffle_index_file.c_str(), "wb");
  if (!xf) throw CometError("shuffle write error: cannot create " + sw.shuffle_index_file + ": " + strerror(errno));
  const bool ok = fwrite(offsets.data(), 8, offsets.size(), xf) == offsets.size();
  if (fclose(xf) != 0 || !ok) throw CometError("shuffle write error: writing " + sw.shuffle_index_file + " failed");
  shuffle_bytes_written_ += offsets[(size_t)P];
  shuffle_write_ns_ += tm.ns() - write_t0;
  shuffle_data_size_ += (int64_t)row_bytes * n;
  lap("write files");
  DevTable none;
  return none;
}

}  // namespace comet
