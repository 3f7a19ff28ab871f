// One chunk through a fused pipeline: launches, the aggregate sinks (ungrouped / grouped, exact float windows), error flags.
#include "exec_internal.hpp"

namespace comet {
// one chunk of input rows resident in HBM → run the fused pipeline on it
void ExecutionContext::process_chunk(const std::vector<DeviceColumnView>& cols, const std::vector<bool>& has_valid, int64_t n) {
  if (n == 0) return;
  std::vector<int> fixed_lens(cols.size(), -1);
  for (size_t i = 0; i < cols.size(); i++) fixed_lens[i] = cols[i].fixed_len;
  Variant& v = variant_for(has_valid, fixed_lens);
  const PipelineDesc& d = v.desc;
  if (d.max_rows_exact && input_rows + n > d.max_rows_exact)
    throw CometError("decimal sum over more rows than the exactness bound allows (" + std::to_string(d.max_rows_exact) + ")");
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  prm.n = n;
  for (size_t i = 0; i < cols.size(); i++) {
    prm.in[i].data = cols[i].data;
    prm.in[i].valid = has_valid[i] ? cols[i].valid : nullptr;
    prm.in[i].aux = cols[i].aux;
    prm.in[i].offset = cols[i].offset;
  }
  err_flags_.ensure(kErrBytes);
  prm.out[kOutErr] = err_flags_.p;
  if (!has_join_) input_rows += n;

  if (d.sink == SinkKind::AggNoGroup) {
    if (agg_variant_ && agg_variant_->desc.NW != d.NW) throw CometError("internal: accumulator layout differs between variants");
    agg_variant_ = &v;
    const int64_t tile = (int64_t)d.R * 256;
    int grid = (int)std::min<int64_t>((n + tile - 1) / tile, 256 * 8);
    size_t need = (size_t)(n_partials_ + grid) * d.NW * 8;
    if (need > partials_.cap) {
      DevBuf bigger;
      bigger.ensure(std::max(need * 2, (size_t)(4096 * d.NW * 8)));
      if (n_partials_) HIP_CHECK(hipMemcpyAsync(bigger.p, partials_.p, (size_t)n_partials_ * d.NW * 8, hipMemcpyDeviceToDevice, stream_));
      HIP_CHECK(hipStreamSynchronize(stream_));
      std::swap(partials_.p, bigger.p);
      std::swap(partials_.cap, bigger.cap);
    }
    prm.out[kOutPartials] = (char*)partials_.p + (size_t)n_partials_ * d.NW * 8;
    for (int attempt = 0;; attempt++) {
      prm.iarg[kFixScaleArg] = packed_fix_scales(d), prm.iarg[kFixScaleArg2] = packed_fix_scales(d, 4);
      timed_begin();
      launch(v, "k_agg", grid, prm);
      timed_end();
      if (d.fix_sums.empty()) break;
      uint64_t aux[2 + 2 * kFixMaxSums];
      read_small(aux, err_flags_.p, sizeof aux);
      std::vector<int> shift;
      if (attempt >= 3 || !adjust_fix_scales(d, aux + 2, shift)) break;
      // the window moved: earlier chunks' partials follow it, this chunk's partials are simply overwritten by the re-run
      for (size_t f = 0; f < shift.size(); f++)
        if (shift[f] > 0 && n_partials_ > 0 &&
            comet_launch_fix_rescale((uint64_t*)partials_.p, n_partials_, d.NW, d.fix_sums[f].word, shift[f], stream_) != 0)
          throw CometError("float sum rescale: launch failed");
    }
    fix_has_state_ = fix_has_state_ || !d.fix_sums.empty();
    n_partials_ += grid;
    return;
  }

  if (d.sink == SinkKind::AggGrouped) {
    if (agg_variant_ && (agg_variant_->desc.NW != d.NW || agg_variant_->desc.NK != d.NK))
      throw CometError("internal: group slot layout differs between variants");
    agg_variant_ = &v;
    if (try_partitioned_merge(v, prm, n)) return;
    const size_t slot_bytes = 8 + 8 * (size_t)(d.NK + d.NW);
    auto alloc_table = [&](DevBuf& buf, int64_t cap) {
      buf.ensure((size_t)cap * slot_bytes);
      HIP_CHECK(hipMemsetAsync(buf.p, 0, (size_t)cap * slot_bytes, stream_));
    };
    if (group_cap_ == 0) {
      // start small: low-cardinality group-bys (TPC-H Q1: 4 groups) must not pay for a table sized by the row count
      int64_t want = 1 << 16;
      if (d.merges_states) {
        // merging Partial states: roughly one input row per group (SF100 Q3's Final aggregate: 1.13 M rows, 1.13 M groups) — size the table for
        // the chunk at once instead of filling and growing it twice (3.9 ms → one pass)
        while (want < 2 * n && want < ((int64_t)1 << 26)) want <<= 1;
      } else if (n <= ((int64_t)1 << 24)) {
        // a modest chunk (a join's output rather than a fact-table scan): at most n groups, and a table of up to 2^22 slots costs less to
        // clear and checkpoint than ONE voided pass over the chunk (SF100 Q3's Partial aggregate: 3 M rows, 1.13 M groups)
        while (want < 2 * n && want < ((int64_t)1 << 22)) want <<= 1;
      }
      group_cap_ = want;
      alloc_table(group_table_, group_cap_);
      group_table_clear_ = true;
      HIP_CHECK(hipMemsetAsync((char*)err_flags_.p + 8, 0, 8, stream_));
    }
    const int64_t tile = (int64_t)d.R * 256;
    int grid_mult = 4;
    if (const char* e = getenv("COMET_GROUPED_GRID_MULT")) grid_mult = std::max(1, atoi(e));
    int grid = (int)std::min<int64_t>((n + tile - 1) / tile, 256 * grid_mult);
    // carry-save LDS accumulation bounds the rows one block may add (comet::kMaxRowsPerBlock = 2^19)
    const int64_t per_block_cap = ((int64_t)1 << 19) - 2 * tile;
    grid = (int)std::max<int64_t>(grid, (n + per_block_cap - 1) / per_block_cap);
    while (true) {
      // checkpoint: if the table fills up mid-chunk some rows are dropped, so the chunk is re-run from the checkpoint.  The checkpoint of a
      // table nothing has gone into is "all zeroes": no copy (SF100 Q3: 369 MB, 0.15 ms of a 1.9 ms Final stage) — going back is a memset
      const bool clear = group_table_clear_;
      if (!clear) {
        group_backup_.ensure((size_t)group_cap_ * slot_bytes);
        HIP_CHECK(hipMemcpyAsync(group_backup_.p, group_table_.p, (size_t)group_cap_ * slot_bytes, hipMemcpyDeviceToDevice, stream_));
      }
      prm.out[0] = group_table_.p;
      prm.iarg[0] = group_cap_;
      prm.iarg[kFixScaleArg] = packed_fix_scales(d), prm.iarg[kFixScaleArg2] = packed_fix_scales(d, 4);
      timed_begin();
      launch(v, "k_gagg", grid, prm);
      timed_end();
      uint64_t head[2 + (kErrBytes - 16) / 8];
      uint32_t flags[4];
      read_small(head, err_flags_.p, d.fix_sums.empty() ? 16 : sizeof head);
      memcpy(flags, head, 16);
      uint64_t groups_now;
      memcpy(&groups_now, &flags[2], 8);
      const bool full = (flags[0] & 32u) != 0;
      if (!full && !d.fix_sums.empty() && fix_attempts_ < 3) {
        std::vector<int> shift;
        if (adjust_fix_scales(d, head + 2, shift)) {
          // a float sum's window moved: back to the checkpoint (table and group counter), shift what earlier chunks accumulated, run again
          fix_attempts_++;
          if (clear) HIP_CHECK(hipMemsetAsync(group_table_.p, 0, (size_t)group_cap_ * slot_bytes, stream_));
          else HIP_CHECK(hipMemcpyAsync(group_table_.p, group_backup_.p, (size_t)group_cap_ * slot_bytes, hipMemcpyDeviceToDevice, stream_));
          uint32_t restore[4] = {flags[0], flags[1], 0, 0};
          memcpy(&restore[2], &groups_committed_, 8);
          write_small(err_flags_.p, restore, 16);
          for (size_t f = 0; f < shift.size(); f++)
            if (shift[f] > 0 && comet_launch_fix_rescale((uint64_t*)group_table_.p, group_cap_, (int64_t)(slot_bytes / 8), 1 + d.NK + d.fix_sums[f].word, shift[f], stream_) != 0)
              throw CometError("float sum rescale: launch failed");
          continue;
        }
      }
      if (!full && (int64_t)groups_now * 2 <= group_cap_) { groups_committed_ = groups_now; group_table_clear_ = false; break; }
      // grow and rehash; after a "full" event restart this chunk from the checkpoint.  A table that is merely more than half full grows ×8.
      // A FULL table voided a whole pass over the chunk (blocks merge their LDS tables at their end, so nobody notices early) and says the
      // chunk holds far more groups than slots: grow ×64, but no further than twice the chunk's rows ever need (SF100 Q3's second join
      // feeds 30 M rows / 1.13 M groups into a 2^16-slot table: one voided pass instead of two)
      int64_t new_cap = group_cap_ * 8;
      if (full) {
        int64_t need = 1 << 16;
        while (need < 2 * n && need < ((int64_t)1 << 28)) need <<= 1;
        new_cap = std::max(new_cap, std::min(group_cap_ * 64, need));
      }
      if (new_cap > ((int64_t)1 << 28)) throw CometError("group table would exceed 2^28 slots");
      DevBuf bigger;
      alloc_table(bigger, new_cap);
      uint32_t zero4[4] = {flags[0] & ~32u, flags[1], 0, 0};
      write_small(err_flags_.p, zero4, 16);
      CometKParams rp;
      memset(&rp, 0, sizeof rp);
      rp.out[0] = bigger.p;
      rp.iarg[0] = new_cap;
      rp.out[2] = err_flags_.p;
      rp.out[3] = full ? group_backup_.p : group_table_.p;
      rp.iarg[1] = group_cap_;
      if (!(full && clear)) launch(v, "k_grehash", (int)std::min<int64_t>((group_cap_ + 255) / 256, 256 * 8), rp);      // (a void pass over a clear table: nothing to carry over)
      HIP_CHECK(hipStreamSynchronize(stream_));
      std::swap(group_table_.p, bigger.p);
      std::swap(group_table_.cap, bigger.cap);
      group_cap_ = new_cap;
      if (!full) { groups_committed_ = groups_now; group_table_clear_ = false; break; }
    }
    fix_attempts_ = 0;
    fix_has_state_ = fix_has_state_ || !d.fix_sums.empty();
    return;
  }

  if (d.sink == SinkKind::Output) {
    for (auto& oc : d.out_cols)
      if (oc.gather_src >= 0 || oc.view_src >= 0 || oc.fmt_kind || !oc.concat_cols.empty()) throw CometError("internal: gathered Utf8 outputs must go through the materialising path");
    const size_t ncol = d.out_cols.size();
    int64_t out_rows = n;
    if (out_vals_.size() < ncol) {
      out_vals_.resize(ncol);
      out_valid_.resize(ncol);
      for (size_t j = 0; j < ncol; j++) {
        if (!out_vals_[j]) out_vals_[j].reset(new DevBuf());
        if (!out_valid_[j]) out_valid_[j].reset(new DevBuf());
      }
    }
    auto bind_outputs = [&](int64_t rows_cap) {
      for (size_t j = 0; j < ncol; j++) {
        int w = d.out_cols[j].type.id == TypeId::Bool ? 1 : fixed_width(d.out_cols[j].type);
        out_vals_[j]->ensure((size_t)rows_cap * w + 16);
        prm.out[kOutFirstCol + 2 * j] = out_vals_[j]->p;
        if (d.out_cols[j].nullable) {
          out_valid_[j]->ensure((size_t)rows_cap + 16);
          prm.out[kOutFirstCol + 2 * j + 1] = out_valid_[j]->p;
        }
      }
    };
    timed_begin();
    if (d.has_filter) {
      // one pass: the survivor count is only known afterwards, so the outputs are sized for the whole chunk
      bind_outputs(n);
      out_rows = launch_fused_filter(v, prm, n);
    } else {
      bind_outputs(n);
      int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
      launch(v, "k_emit", grid, prm);
    }
    timed_end();
    check_device_errors();
    if (out_rows == 0) return;
    // device → host, then cut into batches of at most batch_size rows (FilterExec coalesces toward
    // the configured batch size; planner.rs:4688-4689)
    std::vector<std::vector<uint8_t>> hv(ncol), hk(ncol);
    for (size_t j = 0; j < ncol; j++) {
      int w = d.out_cols[j].type.id == TypeId::Bool ? 1 : fixed_width(d.out_cols[j].type);
      hv[j].resize((size_t)out_rows * w);
      HIP_CHECK(hipMemcpyAsync(hv[j].data(), out_vals_[j]->p, hv[j].size(), hipMemcpyDeviceToHost, stream_));
      if (d.out_cols[j].nullable) {
        hk[j].resize((size_t)out_rows);
        HIP_CHECK(hipMemcpyAsync(hk[j].data(), out_valid_[j]->p, hk[j].size(), hipMemcpyDeviceToHost, stream_));
      }
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    const int64_t bs = batch_size_ > 0 ? batch_size_ : out_rows;
    for (int64_t off = 0; off < out_rows; off += bs) {
      int64_t len = std::min(bs, out_rows - off);
      HostBatch b;
      b.rows = len;
      for (size_t j = 0; j < ncol; j++) {
        HostColumn c;
        c.type = d.out_cols[j].type;
        c.length = len;
        if (c.type.id == TypeId::Bool) {
          c.values.assign((size_t)((len + 7) / 8), 0);
          for (int64_t i = 0; i < len; i++)
            if (hv[j][(size_t)(off + i)]) c.values[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
        } else {
          int w = fixed_width(c.type);
          c.values.assign(hv[j].begin() + (size_t)off * w, hv[j].begin() + (size_t)(off + len) * w);
        }
        if (d.out_cols[j].nullable) {
          int64_t nulls = 0;
          std::vector<uint8_t> bm((size_t)((len + 7) / 8), 0);
          for (int64_t i = 0; i < len; i++) {
            if (hk[j][(size_t)(off + i)]) bm[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
            else nulls++;
          }
          c.null_count = nulls;
          if (nulls) c.validity = std::move(bm);
        }
        b.cols.push_back(std::move(c));
      }
      ready_.push_back(std::move(b));
    }
    return;
  }
  throw CometError("internal: unsupported sink");
}

// Small host↔device transfers go through a pinned scratch block: a copy to/from PAGEABLE memory makes the runtime set up
// staging for the stream, which was measured at 9–24 ms on the first such copy of each plan (profiles/r1_q3_*).
void ExecutionContext::read_small(void* dst, const void* dev_src, size_t n) {
  if (n > 4096) throw CometError("internal: read_small of " + std::to_string(n) + " bytes (its staging holds 4096)");
  small_host_.ensure(4096);
  HIP_CHECK(hipMemcpyAsync(small_host_.p, dev_src, n, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
  memcpy(dst, small_host_.p, n);
}
void ExecutionContext::write_small(void* dev_dst, const void* src, size_t n) {
  if (n > 2048) throw CometError("internal: write_small of " + std::to_string(n) + " bytes (its staging holds 2048)");
  small_host_.ensure(4096);
  // a few words (table headers, hash parameters): sixteen 128-byte slots of the staging take turns, the stream is waited for only when they wrap — a slot may still
  // be the source of an earlier async copy, but not of one sixteen writes (and at least one synchronisation) back
  if (n <= 128) {
    if (small_write_slot_ == 16) {
      HIP_CHECK(hipStreamSynchronize(stream_));
      small_write_slot_ = 0;
    }
    char* slot = (char*)small_host_.p + 2048 + 128 * (size_t)small_write_slot_++;
    memcpy(slot, src, n);
    HIP_CHECK(hipMemcpyAsync(dev_dst, slot, n, hipMemcpyHostToDevice, stream_));
    return;
  }
  HIP_CHECK(hipStreamSynchronize(stream_));   // the scratch may still be the source of an earlier async copy
  small_write_slot_ = 0;
  memcpy((char*)small_host_.p + 2048, src, n);
  HIP_CHECK(hipMemcpyAsync(dev_dst, (char*)small_host_.p + 2048, n, hipMemcpyHostToDevice, stream_));
  small_write_slot_ = 16;                     // (the large write owns the whole area until the stream has been waited for)
}

void ExecutionContext::timed_begin() {
  hipEvent_t a = pool_get_event(device_id_), b = pool_get_event(device_id_);
  timed_.emplace_back(a, b);
  HIP_CHECK(hipEventRecord(a, stream_));
}
void ExecutionContext::timed_end() { HIP_CHECK(hipEventRecord(timed_.back().second, stream_)); }
// resolve the recorded event pairs (the stream must be idle)
void ExecutionContext::collect_timings() {
  for (; timed_done_ < timed_.size(); timed_done_++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, timed_[timed_done_].first, timed_[timed_done_].second) == hipSuccess) {
      last_kernel_ms += ms;
      last_kernel_launches++;
    }
  }
}

void ExecutionContext::check_device_errors() {
  if (!err_flags_.p) return;
  uint32_t flags[4] = {0, 0, 0, 0};
  read_small(flags, err_flags_.p, 16);
  collect_timings();
  raise_device_errors(flags[0]);
}

void ExecutionContext::raise_device_errors(uint32_t f) {
  if (!f) return;
  // Spark error JSON as thrown through CometQueryExecutionException (native/common/src/error.rs:806-831).  An error that names the offending
  // value left its raise site and the value in the detail words of the error block (kparams.h; err_sites.cpp formats them): the JVM side reads
  // params("value"), params("precision") … back (ShimSparkErrorConverter.scala), a missing key would be a NoSuchElementException there.
  {
    uint64_t detail[4 + kErrDetailStrBytes / 8];
    memset(detail, 0, sizeof detail);
    read_small(detail, (const char*)err_flags_.p + 8 * kErrDetailWord, sizeof detail);
    ErrSite site;
    if (detail[0] != 0 && lookup_err_site((uint32_t)(detail[0] - 1), site)) {
      // (the flag that is raised and the site that won the detail words can differ when two kinds of error meet in one launch: the site's own
      // error is reported — it did occur)
      // the SQL fragment of the expression, when the plan carried one
      throw CometError(err_site_json(site, detail[1], detail[2], (const uint8_t*)(detail + 4), (size_t)kErrDetailStrBytes, site_context((uint32_t)(detail[0] - 1))), 1);
    }
  }
  if (f & 1u) throw CometError("{\"errorType\":\"ArithmeticOverflow\",\"errorClass\":\"ARITHMETIC_OVERFLOW\",\"params\":{\"fromType\":\"decimal\"}}", 1);
  if (f & 2u) throw CometError("{\"errorType\":\"ArithmeticOverflow\",\"errorClass\":\"ARITHMETIC_OVERFLOW\",\"params\":{\"fromType\":\"integer\"}}", 1);
  if (f & 4u) throw CometError("{\"errorType\":\"CastOverFlow\",\"errorClass\":\"CAST_OVERFLOW\",\"params\":{}}", 1);
  if (f & 8u) throw CometError("{\"errorType\":\"NumericValueOutOfRange\",\"errorClass\":\"NUMERIC_VALUE_OUT_OF_RANGE.WITH_SUGGESTION\",\"params\":{}}", 1);
  if (f & 512u) throw CometError("{\"errorType\":\"CastInvalidValue\",\"errorClass\":\"CAST_INVALID_INPUT\",\"params\":{\"fromType\":\"STRING\"}}", 1);
  if (f & 1024u) throw CometError("{\"errorType\":\"InvalidInputInCastToDatetime\",\"errorClass\":\"CAST_INVALID_INPUT\",\"params\":{\"fromType\":\"STRING\",\"toType\":\"DATE\"}}", 1);
  if (f & 8192u) throw CometError("{\"errorType\":\"InvalidInputInCastToDatetime\",\"errorClass\":\"CAST_INVALID_INPUT\",\"params\":{\"fromType\":\"STRING\",\"toType\":\"TIMESTAMP\"}}", 1);
  if (f & 16384u) throw CometError("{\"errorType\":\"InvalidInputInCastToDatetime\",\"errorClass\":\"CAST_INVALID_INPUT\",\"params\":{\"fromType\":\"STRING\",\"toType\":\"TIMESTAMP_NTZ\"}}", 1);
  if (f & 4096u) throw CometError("a string cast to a timestamp names a time zone inside the value, or holds a time of day without a date (which takes the current date): not supported by the MI355X native engine");
  if (f & 2048u) throw CometError("a timestamp lies behind the end of its time zone's table (the year 2400): not supported by the MI355X native engine");
  if (f & 262144u) throw CometError("Arrow error: Compute error: long overflow");      // (seconds_to_timestamp of an Int64 beyond i64 / 10^6: seconds_to_timestamp.rs:70-75)
  if (f & 256u) throw CometError("{\"errorType\":\"DivideByZero\",\"errorClass\":\"DIVIDE_BY_ZERO\",\"params\":{}}", 1);
  // decimal_sum_overflow_error (spark-expr/src/lib.rs:131-135; error.rs:75-76, 374-377): ANSI sum / avg of decimals
  if (f & (65536u | 131072u)) {
    const int kind = (f & 65536u) ? 0 : 1;
    throw CometError(decimal_sum_overflow_json(kind, agg_ctx_[kind].get()), 1);
  }
  if (f & 32768u) throw CometError("{\"errorType\":\"RemainderByZero\",\"errorClass\":\"REMAINDER_BY_ZERO\",\"params\":{}}", 1);      // (common/src/error.rs:81-82, 684)
  if (f & 64u) throw CometError("Utf8 group keys longer than 15 bytes are not supported by the GPU hash aggregate yet");
  if (f & 16u)
    throw CometError("decimal sum overflow cannot be decided order-independently for this input (mixed signs beyond the precision bound); "
                     "exact sequential evaluation is not implemented");
  throw CometError("device error flags " + std::to_string(f));
}

// A merging aggregate (Final / PartialMerge) whose whole input is this one chunk: partition → merge in LDS → emit (comet_device.hpp template C'').  → true when
// the result table is ready (part_result_); false: the caller runs template C (not eligible, or a partition held more groups than its LDS table).
bool ExecutionContext::try_partitioned_merge(Variant& v, CometKParams& prm, int64_t n) {
  const PipelineDesc& d = v.desc;
  static const int64_t min_rows = getenv("COMET_AGG_PARTITIONED_MIN_ROWS") ? atoll(getenv("COMET_AGG_PARTITIONED_MIN_ROWS")) : 32768;
  // … and since the end of round 6 ANY grouped aggregate over one modest chunk (a join's output: SF100 Q3's Partial aggregate, 3 M rows into 1.13 M groups, was 0.52 ms of
  // memory-side atomics): whether the keys are many is not known beforehand, so the partition sizes are looked at after the counting pass — a few huge partitions
  // (few groups, many rows each) send the chunk to template C at the price of that one pass
  static const int64_t max_rows_any = getenv("COMET_AGG_PARTITIONED_MAX_ROWS") ? atoll(getenv("COMET_AGG_PARTITIONED_MAX_ROWS")) : ((int64_t)1 << 22);
  if (!single_chunk_hint_ || group_cap_ != 0 || part_result_ready_ || !d.fix_sums.empty() || !dict_id_col_.empty() || min_rows < 0 || n < min_rows) return false;
  if (!d.merges_states && n > max_rows_any) return false;
  if (std::find(d.kernels.begin(), d.kernels.end(), std::string("k_gpmerge")) == d.kernels.end()) return false;
  const int slot_bytes = 8 + 8 * (d.NK + d.NW);
  const int64_t cap = slot_bytes <= 48 ? 1024 : slot_bytes <= 96 ? 512 : slot_bytes <= 192 ? 256 : 128;      // comet_device.hpp AggPart::kCap
  constexpr int64_t kMaxP = 16384;                                                                            // kJoinPartMax
  const int64_t np = (n + cap / 2 - 1) / (cap / 2);
  if (np > kMaxP) return false;
  const int64_t tile = (int64_t)d.R * 256;
  const int64_t g = std::max<int64_t>(1, std::min<int64_t>(1024, (n + 4 * tile - 1) / (4 * tile)));
  const int64_t chunk = ((n + g - 1) / g + tile - 1) / tile * tile;
  const int64_t rec_words = d.NK + d.NW;
  DevBuf recs, part;
  recs.ensure((size_t)n * (size_t)rec_words * 8 + 16);
  part.ensure((size_t)(2 * kMaxP + 16 + g * np) * 4 + 16);
  const size_t ncol = d.out_cols.size();
  std::vector<std::shared_ptr<DevBuf>> vals(ncol), vbytes(ncol);
  for (size_t j = 0; j < ncol; j++) {
    vals[j] = std::make_shared<DevBuf>();
    vbytes[j] = std::make_shared<DevBuf>();
    vals[j]->ensure((size_t)n * out_width(d.out_cols[j]) + 16);
    vbytes[j]->ensure((size_t)n + 16);
    HIP_CHECK(hipMemsetAsync(vbytes[j]->p, 1, (size_t)n, stream_));
    prm.out[kOutFirstCol + 2 * j] = vals[j]->p;
    prm.out[kOutFirstCol + 2 * j + 1] = vbytes[j]->p;
  }
  HIP_CHECK(hipMemsetAsync((char*)err_flags_.p + 8, 0, 8, stream_));      // the group counter: the merge's row positions come from it
  prm.out[1] = recs.p;
  prm.out[3] = part.p;
  prm.iarg[2] = np;
  prm.iarg[5] = chunk;
  prm.iarg[kFixScaleArg] = packed_fix_scales(d), prm.iarg[kFixScaleArg2] = packed_fix_scales(d, 4);
  timed_begin();
  launch(v, "k_gphist", (int)g, prm);
  uint64_t* skew_flag = (uint64_t*)((char*)err_flags_.p + kErrBytes - 8);
  HIP_CHECK(hipMemsetAsync(skew_flag, 0, 8, stream_));
  if (comet_launch_join_part_scan((uint32_t*)part.p + (2 * kMaxP + 16), (int)g, (int)np, (uint32_t*)part.p + (kMaxP + 16), d.merges_states ? 0xffffffffu : (uint32_t)(4 * cap), skew_flag,
                                  stream_) != 0)
    throw CometError("partitioned aggregate: launch failed");
  if (!d.merges_states) {
    uint64_t skew = 0;
    read_small(&skew, skew_flag, 8);
    if (skew) {      // a partition with more than eight times its share of the rows: few groups, many rows each — template C's LDS pre-aggregation is the right tool
      timed_end();
      return false;
    }
  }
  launch(v, "k_gpscat", (int)g, prm);
  launch(v, "k_gpmerge", (int)std::min<int64_t>(np, 256 * 16), prm);
  timed_end();
  uint32_t hdr[4] = {0, 0, 0, 0};
  read_small(hdr, err_flags_.p, 16);
  if (hdr[0] & 32u) {
    // a partition with more groups than its LDS table: forget the attempt (flag and counter), template C takes the chunk
    uint32_t reset[4] = {hdr[0] & ~32u, hdr[1], 0, 0};
    write_small(err_flags_.p, reset, 16);
    HIP_CHECK(hipStreamSynchronize(stream_));
    return false;
  }
  collect_timings();
  raise_device_errors(hdr[0]);
  uint64_t ngroups = 0;
  memcpy(&ngroups, hdr + 2, 8);
  GatherSource gs = nullptr;
  part_result_ = outputs_to_table(v, vals, vbytes, (int64_t)ngroups, gs);
  part_result_.owners.push_back(v.mod);
  part_result_ready_ = true;
  part_merges_++;
  groups_committed_ = ngroups;
  // (the records and the partition counters go back to the pool with this frame: the merge kernel that read them was waited for by the read-back above, and what
  // outputs_to_table queued behind it touches the output's own buffers only)
  return true;
}

void ExecutionContext::finish_aggregate() {
  // AggregateExec emits one state row even for empty input
  std::vector<bool> none(in_types_.size(), false);
  Variant& v = agg_variant_ ? *agg_variant_ : variant_for(none, std::vector<int>(in_types_.size(), -1));
  const PipelineDesc& d = v.desc;
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  partials_.ensure(64);
  const size_t ncol = d.out_cols.size();
  // one result block: [kErrBytes error/aux words][32 B per output column: 16 B value, 1 B validity] → ONE D2H copy
  const size_t block_bytes = kErrBytes + ncol * 32;
  if (err_flags_.cap < block_bytes) {
    DevBuf bigger;
    bigger.ensure(block_bytes);
    HIP_CHECK(hipMemcpyAsync(bigger.p, err_flags_.p, kErrBytes, hipMemcpyDeviceToDevice, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
    std::swap(err_flags_.p, bigger.p);
    std::swap(err_flags_.cap, bigger.cap);
  }
  prm.out[kOutPartials] = partials_.p;
  prm.out[kOutErr] = err_flags_.p;
  prm.iarg[0] = n_partials_;
  char* base = (char*)err_flags_.p + kErrBytes;
  HIP_CHECK(hipMemsetAsync(base, 1, ncol * 32, stream_));  // validity defaults to 1
  for (size_t j = 0; j < ncol; j++) {
    prm.out[kOutFirstCol + 2 * j] = base + j * 32;
    prm.out[kOutFirstCol + 2 * j + 1] = base + j * 32 + 16;
  }
  prm.iarg[kFixScaleArg] = packed_fix_scales(d), prm.iarg[kFixScaleArg2] = packed_fix_scales(d, 4);
  launch(v, "k_agg_final", 1, prm);
  result_host_.ensure(block_bytes);
  HIP_CHECK(hipMemcpyAsync(result_host_.p, err_flags_.p, block_bytes, hipMemcpyDeviceToHost, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));
  collect_timings();
  raise_device_errors(((const uint32_t*)result_host_.p)[0]);
  const uint8_t* hb = (const uint8_t*)result_host_.p + kErrBytes;
  HostBatch b;
  b.rows = 1;
  for (size_t j = 0; j < ncol; j++) {
    HostColumn c;
    c.type = d.out_cols[j].type;
    c.length = 1;
    const uint8_t* val = hb + j * 32;
    if (c.type.id == TypeId::Bool) {
      c.values.assign(1, val[0] ? 1 : 0);
    } else {
      int w = fixed_width(c.type);
      c.values.assign(val, val + w);
    }
    if (d.out_cols[j].nullable && val[16] == 0) {
      c.null_count = 1;
      c.validity.assign(1, 0);
    }
    b.cols.push_back(std::move(c));
  }
  ready_.push_back(std::move(b));
}

// Grouped aggregate result left in HBM (stage boundary of a multi-GPU plan: Partial states feed the next stage's exchange
// or Final aggregate without touching the host).  Utf8 group keys are not supported on this path yet.
DevTable ExecutionContext::grouped_to_device() {
  if (part_result_ready_) {      // the partitioned merge emitted already (and its error flags were read with the group count)
    part_result_ready_ = false;
    return std::move(part_result_);
  }
  DevTable empty;
  if (!agg_variant_) {   // no input rows → no groups: an empty table with the plan's output types
    std::vector<bool> none(in_types_.size(), false);
    auto pv = planned_variant(*plan_, plan_hash_, none, false, has_join_ ? &in_types_ : nullptr);
    for (auto& oc : pv->desc.out_cols) {
      empty.types.push_back(oc.type);
      empty.cols.push_back(DeviceColumnView());
      empty.has_valid.push_back(false);
    }
    return empty;
  }
  Variant& v = *agg_variant_;
  const PipelineDesc& d = v.desc;
  // the error flags and the group counter share the head of the error block: one read-back, one wait
  uint64_t ngroups = 0;
  {
    uint32_t hdr[4] = {0, 0, 0, 0};
    read_small(hdr, err_flags_.p, 16);
    collect_timings();
    memcpy(&ngroups, hdr + 2, 8);
    raise_device_errors(hdr[0]);
  }
  if (getenv("COMET_TRACE_STAGES")) fprintf(stderr, "[comet] grouped result: %llu groups in a table of %lld slots x %zu bytes\n", (unsigned long long)ngroups, (long long)group_cap_, (size_t)(8 + 8 * (d.NK + d.NW)));
  const size_t ncol = d.out_cols.size();
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  prm.out[0] = group_table_.p;
  prm.iarg[0] = group_cap_;
  scratch_counts_.ensure(64);
  HIP_CHECK(hipMemsetAsync(scratch_counts_.p, 0, 8, stream_));
  prm.out[1] = scratch_counts_.p;
  prm.out[kOutErr] = err_flags_.p;
  std::vector<std::shared_ptr<DevBuf>> vals(ncol), vbytes(ncol);
  for (size_t j = 0; j < ncol; j++) {
    vals[j] = std::make_shared<DevBuf>();
    vbytes[j] = std::make_shared<DevBuf>();
    vals[j]->ensure((size_t)std::max<uint64_t>(ngroups, 1) * out_width(d.out_cols[j]) + 16);
    vbytes[j]->ensure((size_t)std::max<uint64_t>(ngroups, 1) + 16);
    HIP_CHECK(hipMemsetAsync(vbytes[j]->p, 1, (size_t)std::max<uint64_t>(ngroups, 1), stream_));
    prm.out[kOutFirstCol + 2 * j] = vals[j]->p;
    prm.out[kOutFirstCol + 2 * j + 1] = vbytes[j]->p;
  }
  prm.iarg[kFixScaleArg] = packed_fix_scales(d), prm.iarg[kFixScaleArg2] = packed_fix_scales(d, 4);
  if (ngroups) launch(v, "k_gemit", (int)std::min<int64_t>((group_cap_ + 255) / 256, 256 * 8), prm);
  GatherSource gs = nullptr;
  if (!dict_id_col_.empty()) gs = [this](int c) { return std::make_pair((const DevTable*)&dict_src_, c); };
  DevTable t = outputs_to_table(v, vals, vbytes, (int64_t)ngroups, gs);
  t.owners.push_back(v.mod);
  check_device_errors();      // (its read-back waits for the stream: the emit's scratch goes back to the pool behind it)
  return t;
}

void ExecutionContext::finish_grouped() {
  if (!agg_variant_) return;  // no input rows → no groups → no output batch
  if (part_result_ready_) {
    DevTable t = grouped_to_device();
    table_to_host_batches(t);
    return;
  }
  Variant& v = *agg_variant_;
  const PipelineDesc& d = v.desc;
  uint64_t ngroups = 0;
  {
    // error flags (word 0) and the group count (bytes 8-15) come back with ONE small copy
    uint32_t head[4] = {0, 0, 0, 0};
    read_small(head, err_flags_.p, 16);
    collect_timings();
    raise_device_errors(head[0]);
    memcpy(&ngroups, head + 2, 8);
  }
  if (ngroups == 0) return;
  if (!dict_id_col_.empty()) {
    // keys that travelled as row indices: gather the strings on the device, then copy the finished table out
    DevTable t = grouped_to_device();
    table_to_host_batches(t);
    return;
  }
  const size_t ncol = d.out_cols.size();
  CometKParams prm;
  memset(&prm, 0, sizeof prm);
  prm.out[0] = group_table_.p;
  prm.iarg[0] = group_cap_;
  scratch_counts_.ensure(64);
  HIP_CHECK(hipMemsetAsync(scratch_counts_.p, 0, 8, stream_));
  prm.out[1] = scratch_counts_.p;
  prm.out[kOutErr] = err_flags_.p;
  // Every output column of the emit lives in ONE device arena — [values of column 0 … values of column n-1][validity bytes of all
  // columns] — so the validity defaults are one memset and the results come back with one D2H copy into one pinned buffer (a Q1-shaped
  // aggregate emits 10 columns of a handful of groups: 20 memsets + 20 copies of a few bytes each were ~0.25 ms of a 7.8 ms task).
  std::vector<int> widths(ncol);
  std::vector<size_t> val_off(ncol);
  size_t arena = 0;
  for (size_t j = 0; j < ncol; j++) {
    const OutCol& oc = d.out_cols[j];
    widths[j] = oc.packed_string ? 16 : (oc.type.id == TypeId::Bool ? 1 : fixed_width(oc.type));
    val_off[j] = arena;
    arena += ((size_t)ngroups * (size_t)widths[j] + 15) & ~(size_t)15;
  }
  const size_t valid_base = arena, valid_stride = ((size_t)ngroups + 15) & ~(size_t)15;
  arena += valid_stride * ncol;
  emit_arena_.ensure(arena + 16);
  HIP_CHECK(hipMemsetAsync((char*)emit_arena_.p + valid_base, 1, valid_stride * ncol, stream_));
  for (size_t j = 0; j < ncol; j++) {
    prm.out[kOutFirstCol + 2 * j] = (char*)emit_arena_.p + val_off[j];
    prm.out[kOutFirstCol + 2 * j + 1] = (char*)emit_arena_.p + valid_base + j * valid_stride;
  }
  prm.iarg[kFixScaleArg] = packed_fix_scales(d), prm.iarg[kFixScaleArg2] = packed_fix_scales(d, 4);
  launch(v, "k_gemit", (int)std::min<int64_t>((group_cap_ + 255) / 256, 256 * 8), prm);
  // results come back through a pooled pinned buffer (a pageable destination would be staged by the runtime at a fraction of the rate)
  struct HostSpan {
    const uint8_t* p = nullptr;
    size_t n = 0;
    const uint8_t* data() const { return p; }
    const uint8_t* begin() const { return p; }
    uint8_t operator[](size_t i) const { return p[i]; }
  };
  PinnedBuf host_arena;
  host_arena.ensure(arena + 16);
  HIP_CHECK(hipMemcpyAsync(host_arena.p, emit_arena_.p, arena, hipMemcpyDeviceToHost, stream_));
  std::vector<HostSpan> hv(ncol), hk(ncol);
  for (size_t j = 0; j < ncol; j++) {
    hv[j].p = (const uint8_t*)host_arena.p + val_off[j];
    hv[j].n = (size_t)ngroups * widths[j];
    hk[j].p = (const uint8_t*)host_arena.p + valid_base + j * valid_stride;
    hk[j].n = (size_t)ngroups;
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  const int64_t total = (int64_t)ngroups;
  const int64_t bs = batch_size_ > 0 ? batch_size_ : total;
  for (int64_t off = 0; off < total; off += bs) {
    const int64_t len = std::min(bs, total - off);
    HostBatch b;
    b.rows = len;
    for (size_t j = 0; j < ncol; j++) {
      const OutCol& oc = d.out_cols[j];
      HostColumn c;
      c.type = oc.type;
      c.length = len;
      if (oc.packed_string) {
        // expand str16 (bytes 0-7 | bytes 8-14 + length byte) into Arrow Utf8 offsets + data
        c.values.resize((size_t)(len + 1) * 4);
        int32_t* offs = (int32_t*)c.values.data();
        offs[0] = 0;
        for (int64_t i = 0; i < len; i++) {
          uint64_t w[2];
          memcpy(w, hv[j].data() + (size_t)(off + i) * 16, 16);
          int slen = hk[j][(size_t)(off + i)] ? (int)(w[1] >> 56) : 0;
          for (int k = 0; k < slen; k++) c.data.push_back((uint8_t)(k < 8 ? (w[0] >> (8 * k)) : (w[1] >> (8 * (k - 8)))));
          offs[i + 1] = (int32_t)c.data.size();
        }
      } else if (c.type.id == TypeId::Bool) {
        c.values.assign((size_t)((len + 7) / 8), 0);
        for (int64_t i = 0; i < len; i++)
          if (hv[j][(size_t)(off + i)]) c.values[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
      } else {
        int w = widths[j];
        c.values.assign(hv[j].begin() + (size_t)off * w, hv[j].begin() + (size_t)(off + len) * w);
      }
      if (oc.nullable) {
        const uint8_t* vb = hk[j].data() + off;
        int64_t valid = 0;
        for (int64_t i = 0; i < len; i++) valid += vb[i] != 0;
        c.null_count = len - valid;
        if (c.null_count) {
          std::vector<uint8_t> bm((size_t)((len + 7) / 8), 0);
          for (int64_t i = 0; i < len; i++)
            if (vb[i]) bm[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
          c.validity = std::move(bm);
        }
      }
      b.cols.push_back(std::move(c));
    }
    ready_.push_back(std::move(b));
  }
}

long long ExecutionContext::packed_fix_scales(const PipelineDesc& d, int first) {      // sums first … first + 3, 16 bits each
  if (fix_scales_.size() != d.fix_sums.size()) fix_scales_.assign(d.fix_sums.size(), kFixDefaultScale);
  uint64_t p = 0;
  for (size_t f = (size_t)first; f < fix_scales_.size() && f < (size_t)first + 4; f++) p |= (uint64_t)(uint16_t)(int16_t)fix_scales_[f] << (16 * (f - (size_t)first));
  return (long long)p;
}

// After a chunk: do the addends seen so far (aux words: 1200 + top and 1200 − low, maxima over every chunk of this execution) fit the
// fixed-point window [2^s, 2^(s + kFixW)) of every sum?  Returns true when scales changed and the chunk has to be run again;
// shift_right[f] > 0 means accumulators of earlier chunks must first be shifted right by that many bits.
//   * a value at or above 2^(s + kFixW) would lose HIGH bits: the window moves up (with 10 bits of slack), always;
//   * bits below 2^s are only truncated (error < rows · 2^s): the window moves down when nothing has been accumulated yet —
//     to the lowest bit seen when the whole range fits (then the sum is exact), else as low as the top value allows.
bool ExecutionContext::adjust_fix_scales(const PipelineDesc& d, const uint64_t* aux, std::vector<int>& shift_right) {
  bool rerun = false;
  shift_right.assign(d.fix_sums.size(), 0);
  for (size_t f = 0; f < d.fix_sums.size(); f++) {
    const uint64_t hi = aux[d.fix_sums[f].aux_hi], lo = aux[d.fix_sums[f].aux_lo];
    if (hi == 0) continue;                      // no finite non-zero addend yet
    const int top = (int)hi - 1200, low = 1200 - (int)lo, s = fix_scales_[f];
    int target = s;
    if (top > s + kFixW) target = top + 10 - kFixW;
    else if (low < s && !fix_has_state_) target = (top - low <= kFixW - 10) ? low : top + 2 - kFixW;
    if (target < -1300) target = -1300;
    if (target == s) continue;
    if (target > s && fix_has_state_) shift_right[f] = target - s;
    fix_scales_[f] = target;
    rerun = true;
  }
  return rerun;
}

// Single-pass filter + compaction (comet_device.hpp filter_fused_body): tile status words and the ticket / total block are zeroed,
// one launch, then the survivor count comes back.  Outputs must already be bound with room for n rows.
int64_t ExecutionContext::launch_fused_filter(Variant& v, CometKParams& prm, int64_t n) {
  const int64_t tile_rows = 256 * (int64_t)v.desc.R;   // P::R row slots per thread
  const int64_t ntiles = (n + tile_rows - 1) / tile_rows;
  scratch_mask_.ensure((size_t)ntiles * 8 + 64);
  scratch_counts_.ensure(64);
  HIP_CHECK(hipMemsetAsync(scratch_mask_.p, 0, (size_t)ntiles * 8, stream_));
  HIP_CHECK(hipMemsetAsync(scratch_counts_.p, 0, 16, stream_));
  prm.out[0] = scratch_mask_.p;
  prm.out[1] = scratch_counts_.p;
  launch(v, "k_filter", (int)std::min<int64_t>(ntiles, 256 * 8), prm);
  uint64_t total = 0;
  read_small(&total, (char*)scratch_counts_.p + 8, 8);
  return (int64_t)total;
}

}  // namespace comet
