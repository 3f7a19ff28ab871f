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
    for (size_t i = 0; i < k; i++) key_cols.push_back((int)i);
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

  // one download of the partition-major table
  std::vector<std::unique_ptr<PinnedBuf>> hv(n_payload), hb(n_payload), hd(n_payload);
  for (size_t j = 0; j < n_payload && n > 0; j++) {
    const DType& ty = grouped.types[j];
    const DeviceColumnView& v = grouped.cols[j];
    if (v.offset != 0) throw CometError("ShuffleWriter: input column with a non-zero Arrow offset is not supported yet");
    const bool is_str = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    const size_t bytes = is_str ? (size_t)(n + 1) * 4 : ty.id == TypeId::Bool ? (size_t)((n + 7) / 8) : (size_t)n * fixed_width(ty);
    hv[j].reset(new PinnedBuf());
    hv[j]->ensure(bytes + 8);
    HIP_CHECK(hipMemcpyAsync(hv[j]->p, v.data, bytes, hipMemcpyDeviceToHost, stream_));
    if (grouped.has_valid[j]) {
      hb[j].reset(new PinnedBuf());
      hb[j]->ensure((size_t)((n + 7) / 8) + 8);
      HIP_CHECK(hipMemcpyAsync(hb[j]->p, v.valid, (size_t)((n + 7) / 8), hipMemcpyDeviceToHost, stream_));
    }
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  for (size_t j = 0; j < n_payload && n > 0; j++) {
    const DType& ty = grouped.types[j];
    if (ty.id != TypeId::String && ty.id != TypeId::Bytes) continue;
    const int32_t* offs = (const int32_t*)hv[j]->p;
    if (offs[0] != 0) throw CometError("ShuffleWriter: Utf8 column whose offsets do not start at 0");
    hd[j].reset(new PinnedBuf());
    hd[j]->ensure((size_t)offs[n] + 8);
    if (offs[n]) HIP_CHECK(hipMemcpyAsync(hd[j]->p, grouped.cols[j].aux, (size_t)offs[n], hipMemcpyDeviceToHost, stream_));
  }
  HIP_CHECK(hipStreamSynchronize(stream_));

  lap("download");
  // frame every partition on the host threads
  const int64_t bs = batch_size_ > 0 ? batch_size_ : std::max<int64_t>(n, 1);
  const ShuffleCodec codec = (ShuffleCodec)sw.shuffle_codec;
  const double write_t0 = tm.ns();
  // Blocks in file order — (partition, first row, rows) — grouped into runs of consecutive blocks of ≈4 MiB of column data.  The
  // scan threads encode whole runs (one output buffer per run, allocated once); this thread writes finished runs to the data
  // file in order while later runs are still being encoded.
  struct BlockTask { int p; int64_t first, rows; };
  std::vector<BlockTask> tasks;
  for (int p = 0; p < P; p++)
    for (int64_t r = starts[(size_t)p]; r < starts[(size_t)p + 1]; r += bs) tasks.push_back({p, r, std::min(bs, starts[(size_t)p + 1] - r)});
  size_t row_bytes = 0;
  for (size_t j = 0; j < n_payload; j++) {
    const DType& ty = grouped.types[j];
    const bool is_str = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    row_bytes += is_str ? 4 + (n > 0 ? (size_t)(((const int32_t*)hv[j]->p)[n] / n) + 1 : 0) : ty.id == TypeId::Bool ? 1 : (size_t)fixed_width(ty);
    if (hb[j]) row_bytes += 1;
  }
  struct Run {
    size_t first = 0, last = 0;      // tasks [first, last)
    std::vector<uint8_t> bytes;
    std::vector<size_t> block_size;  // per task
    std::string error;
    bool done = false;
    double encode_ms = 0;
  };
  std::vector<Run> runs;
  for (size_t t = 0; t < tasks.size();) {
    size_t e = t, acc = 0;
    while (e < tasks.size() && acc < (size_t)(4 << 20)) acc += (size_t)tasks[e++].rows * std::max<size_t>(row_bytes, 1);
    Run r;
    r.first = t;
    r.last = e;
    runs.push_back(std::move(r));
    t = e;
  }
  std::mutex mu;
  std::condition_variable cv;
  for (size_t ri = 0; ri < runs.size(); ri++) {
    scan_pool_submit([&, ri]() {
      Run& r = runs[ri];
      Timer rt;
      try {
        size_t est = 0;
        for (size_t t = r.first; t < r.last; t++) est += (size_t)tasks[t].rows * row_bytes + 2048;
        r.bytes.reserve(est + est / 8 + (64 << 10));
        std::vector<ColumnSlice> cols(n_payload);
        for (size_t j = 0; j < n_payload; j++) {
          cols[j].type = grouped.types[j];
          cols[j].validity = hb[j] ? (const uint8_t*)hb[j]->p : nullptr;
          cols[j].values = hv[j]->p;
          cols[j].data = hd[j] ? (const uint8_t*)hd[j]->p : nullptr;
        }
        for (size_t t = r.first; t < r.last; t++) {
          for (auto& c : cols) c.first = tasks[t].first;
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
  const int fd = open(sw.shuffle_data_file.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  std::string failure;
  if (fd < 0) failure = "shuffle write error: cannot create " + sw.shuffle_data_file + ": " + strerror(errno);
  std::vector<int64_t> offsets((size_t)P + 1, 0);
  int64_t file_pos = 0;
  int next_p = 0;
  double wait_ms = 0, write_ms = 0, enc_sum = 0, enc_max = 0;
  for (size_t ri = 0; ri < runs.size(); ri++) {   // every run is waited for, also after a failure: the tasks reference this frame
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
  if (trace)
    fprintf(stderr, "[comet] shuffle write: %zu runs, encode cpu %.1f ms total (max %.2f ms/run), writer waited %.1f ms, wrote for %.1f ms\n", runs.size(),
            enc_sum, enc_max, wait_ms, write_ms);
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
