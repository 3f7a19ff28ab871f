// Window and Expand operators.
#include "exec_internal.hpp"

namespace comet {
u128 detail::pow10_u128_host(int p) {
  u128 r = 1;
  for (int i = 0; i < p; i++) r *= 10;
  return r;
}

// Window: ranking / ntile / lag / lead and prefix-sum aggregates over input sorted by (partition keys, order keys) — see window_kernels.hip
DevTable ExecutionContext::window(const Operator& w, const DevTable& in) {
  const int64_t n = in.rows;
  if (n >= ((int64_t)1 << 31)) throw CometError("Window: more than 2^31 rows in one partition of the plan");
  DevTable out = in;
  auto add_col = [&](const DType& t, std::shared_ptr<DevBuf> data, std::shared_ptr<DevBuf> valid_bits, std::shared_ptr<DevBuf> aux = nullptr) {
    DeviceColumnView v;
    v.data = data ? data->p : nullptr;
    v.valid = valid_bits ? (const uint8_t*)valid_bits->p : nullptr;
    v.aux = aux ? aux->p : nullptr;
    out.types.push_back(t);
    out.cols.push_back(v);
    out.has_valid.push_back(valid_bits != nullptr);
    if (data) out.owners.push_back(data);
    if (valid_bits) out.owners.push_back(valid_bits);
    if (aux) out.owners.push_back(aux);
  };
  if (n == 0) {
    for (size_t k = 0; k < w.window_fns.size(); k++) {
      const std::string& f = w.window_fns[k].func;
      if (w.window_fns[k].is_agg) {
        const AggExpr& a = w.window_fns[k].agg;
        const bool dec = a.dtype.id == TypeId::Decimal && a.kind != AggKind::Count;
        if (a.kind == AggKind::Min || a.kind == AggKind::Max || a.kind == AggKind::First || a.kind == AggKind::Last) add_col(in.types[(size_t)a.children[0]->bound_index], nullptr, nullptr);
        else add_col(dec ? a.dtype : DType::of(TypeId::Int64), nullptr, nullptr);
        continue;
      }
      DType t = (f == "percent_rank" || f == "cume_dist") ? DType::of(TypeId::Double) : (f == "lag" || f == "lead" || f == "nth_value") ? in.types[(size_t)w.window_fns[k].args[0]->bound_index] : DType::of(TypeId::Int32);
      add_col(t, nullptr, nullptr);
    }
    return out;
  }
  timed_begin();
  int Wp = 0, Wo = 0;
  std::shared_ptr<DevBuf> pp, po;
  if (!window_psort_.at(&w)->sort_orders.empty()) pp = sort_key_planes(*window_psort_.at(&w), in, Wp);
  if (!window_osort_.at(&w)->sort_orders.empty()) po = sort_key_planes(*window_osort_.at(&w), in, Wo);
  DevBuf fpart, fpeer, tiles;
  auto sp = std::make_shared<DevBuf>(), sg = std::make_shared<DevBuf>(), first_part = std::make_shared<DevBuf>(), first_peer = std::make_shared<DevBuf>();
  fpart.ensure((size_t)n * 4 + 16);
  fpeer.ensure((size_t)n * 4 + 16);
  tiles.ensure((size_t)((n + 1023) / 1024 + 2) * 8);
  sp->ensure((size_t)(n + 2) * 4);
  sg->ensure((size_t)(n + 2) * 4);
  first_part->ensure((size_t)(n + 2) * 4);
  first_peer->ensure((size_t)(n + 2) * 4);
  if (comet_launch_window_flags(pp ? (const uint8_t*)pp->p : nullptr, Wp, po ? (const uint8_t*)po->p : nullptr, Wo, n, (uint32_t*)fpart.p, (uint32_t*)fpeer.p, stream_) != 0)
    throw CometError("window: launch failed");
  pq_launch_u32_scan((const uint32_t*)fpart.p, n, (uint64_t*)tiles.p, (int32_t*)sp->p, stream_);
  pq_launch_u32_scan((const uint32_t*)fpeer.p, n, (uint64_t*)tiles.p, (int32_t*)sg->p, stream_);
  if (comet_launch_window_first((const uint32_t*)fpart.p, (const int32_t*)sp->p, (const uint32_t*)fpeer.p, (const int32_t*)sg->p, n, (uint32_t*)first_part->p,
                                (uint32_t*)first_peer->p, stream_) != 0)
    throw CometError("window: launch failed");
  // out row i = row idx[i] of column c where ok[i], else NULL (or the literal `dflt`): the tail of lag / lead / first / last / nth_value
  auto gather_rows = [&](int c, const std::shared_ptr<DevBuf>& idx, DevBuf& ok, const Expr* dflt) {
    const DType& t = in.types[(size_t)c];
    const DeviceColumnView& sc = in.cols[(size_t)c];
    auto okv = std::make_shared<DevBuf>(), bits = std::make_shared<DevBuf>();
    okv->ensure((size_t)n + 16);
    bits->ensure((size_t)((n + 7) / 8) + 16);
    if (comet_launch_window_offset_valid((const uint32_t*)idx->p, (const uint8_t*)ok.p, in.has_valid[(size_t)c] ? sc.valid : nullptr, n, (uint8_t*)okv->p, stream_) != 0)
      throw CometError("window: launch failed");
    const bool has_default = dflt != nullptr;
    if (!has_default) pq_launch_pack((const uint8_t*)okv->p, (uint8_t*)bits->p, n, stream_);
    if (t.id == TypeId::String || t.id == TypeId::Bytes) {
      DeviceColumnView ov;
      take_utf8(sc, (const uint32_t*)idx->p, (const uint8_t*)okv->p, nullptr, n, ov, out.owners);
      ov.valid = (const uint8_t*)bits->p;
      out.types.push_back(t);
      out.cols.push_back(ov);
      out.has_valid.push_back(true);
      out.owners.push_back(bits);
    } else {
      const int wd = t.id == TypeId::Bool ? 0 : fixed_width(t);
      auto data = std::make_shared<DevBuf>();
      data->ensure((wd ? (size_t)n * (size_t)wd : (size_t)((n + 7) / 8)) + 16);
      if (comet_launch_take(wd, sc.data, (const uint32_t*)idx->p, n, data->p, stream_) != 0) throw CometError("window: take failed");
      if (has_default) {
        // rows whose offset row is outside the partition take the literal default (lag(x, k, d))
        const Expr& lit = *dflt;
        uint8_t buf[16] = {0};
        if (t.id == TypeId::Decimal) { i128 v = lit.lit_dec; memcpy(buf, &v, 16); }
        else if (t.id == TypeId::Double) { double v = lit.lit_f64; memcpy(buf, &v, 8); }
        else if (t.id == TypeId::Float) { float v = (float)lit.lit_f64; memcpy(buf, &v, 4); }
        else { int64_t v = lit.lit_i64; memcpy(buf, &v, 8); }
        if (comet_launch_window_default(wd, (const uint8_t*)ok.p, n, buf, data->p, (uint8_t*)okv->p, stream_) != 0) throw CometError("window: launch failed");
        pq_launch_pack((const uint8_t*)okv->p, (uint8_t*)bits->p, n, stream_);
      }
      add_col(t, data, bits);
    }
    out.owners.push_back(okv);
    out.owners.push_back(idx);
  };
  // FIRST_VALUE / LAST_VALUE / nth_value: the frame's first / last / n-th row — or non-NULL row (IGNORE NULLS: a binary search over the
  // column's non-NULL prefix counts inside the frame) — then the same gather as lag / lead
  auto pick_and_gather = [&](int c, int mode, int64_t nth, bool ignore_nulls, int lo_kind, int64_t lo_off, int hi_kind, int64_t hi_off) {
    const DeviceColumnView& sc = in.cols[(size_t)c];
    if (sc.offset != 0) throw CometError("Window: a column with a non-zero Arrow offset is not supported yet");
    DevBuf ok, flags, t32, counts;
    auto idx = std::make_shared<DevBuf>();
    idx->ensure((size_t)n * 4 + 16);
    ok.ensure((size_t)n + 16);
    const int32_t* C = nullptr;
    if (ignore_nulls && in.has_valid[(size_t)c]) {
      flags.ensure((size_t)n * 4 + 16);
      counts.ensure((size_t)(n + 2) * 4);
      t32.ensure((size_t)((n + 1023) / 1024 + 2) * 8);
      if (comet_launch_window_valid_flags(sc.valid, n, (uint32_t*)flags.p, stream_) != 0) throw CometError("window: launch failed");
      pq_launch_u32_scan((const uint32_t*)flags.p, n, (uint64_t*)t32.p, (int32_t*)counts.p, stream_);
      C = (const int32_t*)counts.p;
    }
    if (comet_launch_window_pick(mode, nth, lo_kind, lo_off, hi_kind, hi_off, C, (const int32_t*)sp->p, (const int32_t*)sg->p, (const uint32_t*)first_part->p,
                                 (const uint32_t*)first_peer->p, n, (uint32_t*)idx->p, (uint8_t*)ok.p, stream_) != 0)
      throw CometError("window: launch failed");
    gather_rows(c, idx, ok, nullptr);
    HIP_CHECK(hipStreamSynchronize(stream_));   // scratch goes back to the pool
  };
  // a window function's frame as the kernels take it: WB_* kinds and offsets (RANGE value offsets: the address of the searched row positions)
  auto frame_of = [&](const Operator::WindowFn& fn, int& lo_kind, int& hi_kind, int64_t& lo_off, int64_t& hi_off) {
      auto bound_kind = [&](int k, bool upper) { return k == 0 ? 0 : k == 1 ? (fn.frame_rows ? 3 : 4) : (fn.frame_rows ? 1 : 2); (void)upper; };   // → WB_* of window_kernels.hip
      lo_kind = bound_kind(fn.frame_lower, false);
      hi_kind = bound_kind(fn.frame_upper, true);
      // RANGE frames with value offsets: the bounds are row positions searched over the ORDER BY key (window_range_bounds_kernel); the
      // frame kernels then read them from these arrays (their address travels in the offset slot of the frame)
      lo_off = fn.frame_lower_off;
      hi_off = fn.frame_upper_off;
      std::shared_ptr<DevBuf> range_lo, range_hi;
      if (lo_kind == 4 || hi_kind == 4) {
        const Operator::SortKey& ok = w.window_order[0];
        const int kc = ok.child->bound_index;
        const DeviceColumnView& kcol = in.cols[(size_t)kc];
        if (kcol.offset != 0) throw CometError("Window: ORDER BY column with a non-zero Arrow offset is not supported yet");
        range_lo = std::make_shared<DevBuf>();
        range_hi = std::make_shared<DevBuf>();
        range_lo->ensure((size_t)n * 4 + 16);
        range_hi->ensure((size_t)n * 4 + 16);
        // magnitudes: the literal when the plan carries one, else |offset| (planner.rs:3032-3035, :3091-3094)
        const int64_t dlo = fn.frame_lower_range ? fn.frame_lower_range->lit_i64 : (fn.frame_lower_off < 0 ? -fn.frame_lower_off : fn.frame_lower_off);
        const int64_t dhi = fn.frame_upper_range ? fn.frame_upper_range->lit_i64 : fn.frame_upper_off;
        if (comet_launch_window_range_bounds(fixed_width(in.types[(size_t)kc]), kcol.data, in.has_valid[(size_t)kc] ? kcol.valid : nullptr, (const int32_t*)sp->p,
                                             (const uint32_t*)first_part->p, n, ok.descending ? 1 : 0, ok.nulls_last ? 0 : 1, lo_kind == 4, dlo, hi_kind == 4, dhi,
                                             (int32_t*)range_lo->p, (int32_t*)range_hi->p, stream_) != 0)
          throw CometError("window: launch failed");
        if (lo_kind == 4) lo_off = (int64_t)(uintptr_t)range_lo->p;
        if (hi_kind == 4) hi_off = (int64_t)(uintptr_t)range_hi->p;
        out.owners.push_back(range_lo);
        out.owners.push_back(range_hi);
      }
  };
  struct Prefix { std::shared_ptr<DevBuf> S, SH, C; };   // 128-bit inclusive sums (low part), sums of the high 64 bits (wide decimals only), non-NULL prefix counts
  std::map<int, Prefix> prefix;                           // by argument column
  for (auto& fn : w.window_fns) {
    if (fn.is_agg) {
      const AggExpr& a = fn.agg;
      const ExprP& arg = a.children[0];
      int lo_kind, hi_kind;
      int64_t lo_off, hi_off;
      frame_of(fn, lo_kind, hi_kind, lo_off, hi_off);
      if (a.kind == AggKind::First || a.kind == AggKind::Last) {
        pick_and_gather(arg->bound_index, a.kind == AggKind::First ? 0 : 1, 0, fn.ignore_nulls || a.ignore_nulls, lo_kind, lo_off, hi_kind, hi_off);
        continue;
      }
      if (a.kind == AggKind::Min || a.kind == AggKind::Max) {
        // the frame's extreme: running extremes per partition from its start (P) and towards its end (Q) — two segmented scans — answer
        // every frame that touches a partition edge; a frame bounded on both sides is walked row by row (≤ 4097 rows)
        const int cc = arg->bound_index;
        const DeviceColumnView& sc = in.cols[(size_t)cc];
        if (sc.offset != 0) throw CometError("Window: aggregate over a column with a non-zero Arrow offset is not supported yet");
        const DType& at = in.types[(size_t)cc];
        const int width = at.id == TypeId::Decimal ? 16 : fixed_width(at);
        const int is_max = a.kind == AggKind::Max ? 1 : 0;
        DevBuf wide, okf, local, tl, P, Ph, Q, Qh;
        wide.ensure((size_t)n * 16 + 16);
        okf.ensure((size_t)n * 4 + 16);
        if (comet_launch_window_widen(width, sc.data, in.has_valid[(size_t)cc] ? sc.valid : nullptr, n, wide.p, nullptr, (uint32_t*)okf.p, stream_) != 0) throw CometError("window: launch failed");
        const bool need_p = lo_kind == 0, need_q = lo_kind != 0 && hi_kind == 0;
        if (need_p || need_q) {
          local.ensure((size_t)n * 32 + 64);
          tl.ensure((size_t)((n + 1023) / 1024 + 2) * 64 + 64);
          DevBuf& V = need_p ? P : Q;
          DevBuf& H = need_p ? Ph : Qh;
          V.ensure((size_t)n * 16 + 16);
          H.ensure((size_t)n + 16);
          if (comet_launch_window_running_extreme(wide.p, (const uint32_t*)okf.p, (const int32_t*)sp->p, n, need_p ? 0 : 1, is_max, local.p, tl.p, V.p, (uint8_t*)H.p, stream_) != 0)
            throw CometError("window: launch failed");
        }
        auto data = std::make_shared<DevBuf>(), okb = std::make_shared<DevBuf>(), bits = std::make_shared<DevBuf>();
        data->ensure((size_t)n * (size_t)width + 16);
        okb->ensure((size_t)n + 16);
        bits->ensure((size_t)((n + 7) / 8) + 16);
        if (comet_launch_window_minmax(is_max, lo_kind, lo_off, hi_kind, hi_off, wide.p, (const uint32_t*)okf.p, P.p, (const uint8_t*)Ph.p, Q.p, (const uint8_t*)Qh.p,
                                       (const int32_t*)sp->p, (const int32_t*)sg->p, (const uint32_t*)first_part->p, (const uint32_t*)first_peer->p, n, width, data->p, (uint8_t*)okb->p,
                                       stream_) != 0)
          throw CometError("window: launch failed");
        pq_launch_pack((const uint8_t*)okb->p, (uint8_t*)bits->p, n, stream_);
        HIP_CHECK(hipStreamSynchronize(stream_));   // scratch goes back to the pool
        add_col(at, data, bits);
        out.owners.push_back(okb);
        continue;
      }
      const int c = arg->kind == ExprKind::Bound ? arg->bound_index : -1 - (int)(arg->lit_null ? 1 : 0);   // literals: −1 non-NULL, −2 NULL
      auto it = prefix.find(c);
      if (it == prefix.end()) {
        auto S = std::make_shared<DevBuf>(), C = std::make_shared<DevBuf>();
        std::shared_ptr<DevBuf> SH;
        DevBuf wide, wide_hi, okf, t128, t32;
        wide.ensure((size_t)n * 16 + 16);
        okf.ensure((size_t)n * 4 + 16);
        S->ensure((size_t)n * 16 + 16);
        C->ensure((size_t)(n + 2) * 4);
        t128.ensure((size_t)((n + 2047) / 2048 + 2) * 16);
        t32.ensure((size_t)((n + 1023) / 1024 + 2) * 8);
        const void* src = nullptr;
        const uint8_t* vb = nullptr;
        int width = 8;
        if (c >= 0) {
          const DeviceColumnView& sc = in.cols[(size_t)c];
          if (sc.offset != 0) throw CometError("Window: aggregate over a column with a non-zero Arrow offset is not supported yet");
          src = sc.data;
          vb = in.has_valid[(size_t)c] ? sc.valid : nullptr;
          width = in.types[(size_t)c].id == TypeId::Decimal ? 16 : fixed_width(in.types[(size_t)c]);
        }
        DevBuf zero_bits;
        if (c == -2) {   // COUNT(NULL literal): no row counts — an all-zero validity bitmap
          zero_bits.ensure((size_t)((n + 7) / 8) + 16);
          HIP_CHECK(hipMemsetAsync(zero_bits.p, 0, (size_t)((n + 7) / 8), stream_));
          vb = (const uint8_t*)zero_bits.p;
        }
        const bool split = c >= 0 && in.types[(size_t)c].id == TypeId::Decimal && in.types[(size_t)c].precision > 18;
        if (split) {
          wide_hi.ensure((size_t)n * 16 + 16);
          SH = std::make_shared<DevBuf>();
          SH->ensure((size_t)n * 16 + 16);
        }
        if (comet_launch_window_widen(width, src, vb, n, wide.p, split ? wide_hi.p : nullptr, (uint32_t*)okf.p, stream_) != 0 ||
            comet_launch_scan128(wide.p, n, t128.p, S->p, stream_) != 0 || (split && comet_launch_scan128(wide_hi.p, n, t128.p, SH->p, stream_) != 0))
          throw CometError("window: launch failed");
        pq_launch_u32_scan((const uint32_t*)okf.p, n, (uint64_t*)t32.p, (int32_t*)C->p, stream_);
        HIP_CHECK(hipStreamSynchronize(stream_));   // scratch goes back to the pool
        it = prefix.emplace(c, Prefix{S, SH, C}).first;
      }
      const DType at = c >= 0 ? in.types[(size_t)c] : arg->dtype;
      int fnk = a.kind == AggKind::Count ? 2 : a.kind == AggKind::Avg ? 3 : (at.id == TypeId::Decimal ? 0 : 1);
      const DType rt = fnk == 2 || fnk == 1 ? DType::of(TypeId::Int64) : a.dtype;
      // precision bounds: SUM checks the result type; AVG checks the sum type, scales by 10^(result scale − sum scale) and checks the result type
      const DType sum_t = fnk == 3 ? a.sum_dtype : a.dtype;
      u128 bound = fnk == 0 || fnk == 3 ? pow10_u128_host(sum_t.precision) - 1 : 0, avg_bound = fnk == 3 ? pow10_u128_host(a.dtype.precision) - 1 : 0;
      i128 scaler = fnk == 3 ? (i128)pow10_u128_host(std::max(0, a.dtype.scale - sum_t.scale)) : 1;
      auto data = std::make_shared<DevBuf>(), okb = std::make_shared<DevBuf>(), bits = std::make_shared<DevBuf>();
      data->ensure((size_t)n * (fnk == 0 || fnk == 3 ? 16 : 8) + 16);
      okb->ensure((size_t)n + 16);
      bits->ensure((size_t)((n + 7) / 8) + 16);
      if (comet_launch_window_agg(fnk, lo_kind, lo_off, hi_kind, hi_off, it->second.S->p, it->second.SH ? it->second.SH->p : nullptr, (const int32_t*)it->second.C->p, (const int32_t*)sp->p, (const int32_t*)sg->p,
                                  (const uint32_t*)first_part->p, (const uint32_t*)first_peer->p, n, &bound, &scaler, &avg_bound, data->p, (uint8_t*)okb->p, stream_) != 0)
        throw CometError("window: launch failed");
      pq_launch_pack((const uint8_t*)okb->p, (uint8_t*)bits->p, n, stream_);
      add_col(rt, data, fnk == 2 ? nullptr : bits);
      if (fnk == 2) out.owners.push_back(bits);
      out.owners.push_back(okb);
      continue;
    }
    const std::string& f = fn.func;
    int kind = f == "row_number" ? 0 : f == "rank" ? 1 : f == "dense_rank" ? 2 : f == "percent_rank" ? 3 : f == "cume_dist" ? 4 : f == "ntile" ? 5 : -1;
    if (kind >= 0) {
      const bool dbl = kind == 3 || kind == 4;
      auto data = std::make_shared<DevBuf>();
      data->ensure((size_t)n * (dbl ? 8 : 4) + 16);
      if (comet_launch_window_rank(kind, kind == 5 ? fn.args[0]->lit_i64 : 0, (const int32_t*)sp->p, (const int32_t*)sg->p, (const uint32_t*)first_part->p,
                                   (const uint32_t*)first_peer->p, n, data->p, stream_) != 0)
        throw CometError("window: launch failed");
      add_col(DType::of(dbl ? TypeId::Double : TypeId::Int32), data, nullptr);
      continue;
    }
    if (f == "nth_value") {
      int lo_kind, hi_kind;
      int64_t lo_off, hi_off;
      frame_of(fn, lo_kind, hi_kind, lo_off, hi_off);
      pick_and_gather(fn.args[0]->bound_index, 2, fn.args[1]->lit_i64, fn.ignore_nulls, lo_kind, lo_off, hi_kind, hi_off);
      continue;
    }
    // lag / lead: a gather with NULL outside the partition
    const int c = fn.args[0]->bound_index;
    const int64_t k = fn.args.size() >= 2 ? fn.args[1]->lit_i64 : 1;
    const int64_t shift = f == "lag" ? -k : k;
    if (in.cols[(size_t)c].offset != 0) throw CometError(f + " over a column with a non-zero Arrow offset is not supported yet");
    DevBuf ok;
    auto idx = std::make_shared<DevBuf>();
    idx->ensure((size_t)n * 4 + 16);
    ok.ensure((size_t)n + 16);
    const Expr* dflt = fn.args.size() == 3 && !fn.args[2]->lit_null ? fn.args[2].get() : nullptr;
    if (fn.ignore_nulls && in.has_valid[(size_t)c] && k != 0) {
      // IGNORE NULLS: the k-th non-NULL row before / after the current one — the pick kernel's search over the non-NULL prefix counts,
      // with the whole partition as the frame
      DevBuf flags, t32, counts;
      flags.ensure((size_t)n * 4 + 16);
      counts.ensure((size_t)(n + 2) * 4);
      t32.ensure((size_t)((n + 1023) / 1024 + 2) * 8);
      if (comet_launch_window_valid_flags(in.cols[(size_t)c].valid, n, (uint32_t*)flags.p, stream_) != 0) throw CometError("window: launch failed");
      pq_launch_u32_scan((const uint32_t*)flags.p, n, (uint64_t*)t32.p, (int32_t*)counts.p, stream_);
      // a negative offset looks the other way: lag(x, -k) = lead(x, k) (k = 0 with IGNORE NULLS is refused at createPlan)
      const bool backwards = (f == "lag") == (k > 0);
      if (comet_launch_window_pick(backwards ? 3 : 4, k > 0 ? k : -k, 0, 0, 0, 0, (const int32_t*)counts.p, (const int32_t*)sp->p, (const int32_t*)sg->p, (const uint32_t*)first_part->p,
                                   (const uint32_t*)first_peer->p, n, (uint32_t*)idx->p, (uint8_t*)ok.p, stream_) != 0)
        throw CometError("window: launch failed");
      gather_rows(c, idx, ok, dflt);
      HIP_CHECK(hipStreamSynchronize(stream_));
      continue;
    }
    if (comet_launch_window_offset(shift, (const int32_t*)sp->p, (const uint32_t*)first_part->p, n, (uint32_t*)idx->p, (uint8_t*)ok.p, stream_) != 0) throw CometError("window: launch failed");
    gather_rows(c, idx, ok, dflt);
    HIP_CHECK(hipStreamSynchronize(stream_));   // `ok` goes back to the pool; idx / okv are released with this scope
  }
  timed_end();
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  return out;
}

DevTable ExecutionContext::expand(const Operator& ex, const DevTable& in) {
  const ExpandInfo& info = expand_info_.at(&ex);
  const size_t ncol = info.out_cols.size(), P = info.parts.size();
  const int64_t n = in.rows, total = n * (int64_t)P;
  if (total >= ((int64_t)1 << 32)) throw CometError("Expand: more than 2^32 output rows in one partition");
  Variant u;   // unified description of the output columns; k_pack comes from the first generated projection
  u.desc.out_cols = info.out_cols;
  std::vector<std::shared_ptr<DevBuf>> vals(ncol), vbytes(ncol);
  for (size_t c = 0; c < ncol; c++) {
    vals[c] = std::make_shared<DevBuf>();
    vbytes[c] = std::make_shared<DevBuf>();
    vals[c]->ensure((size_t)std::max<int64_t>(total, 1) * (size_t)out_width(info.out_cols[c]) + 16);
    vbytes[c]->ensure((size_t)std::max<int64_t>(total, 1) + 16);
    HIP_CHECK(hipMemsetAsync(vbytes[c]->p, 1, (size_t)std::max<int64_t>(total, 1), stream_));   // outputs the kernels treat as non-nullable stay valid
  }
  timed_begin();
  for (size_t p = 0; p < P; p++) {
    const ExpandPart& part = info.parts[p];
    const int64_t base = (int64_t)p * n;
    if (!part.proj->project_list.empty()) {
      auto pv = planned_variant(*part.proj, plan_hash_ ^ (0x9E3779B97F4A7C15ull * (uint64_t)(node_id_[part.proj.get()] + 1)), in.has_valid, true, &in.types);
      Variant v;
      v.desc = pv->desc;
      note_sites(v.desc);
      v.mod = jit_load(pv->code);
      if (!u.mod) u.mod = v.mod;
      CometKParams prm;
      memset(&prm, 0, sizeof prm);
      prm.n = n;
      for (size_t i = 0; i < in.cols.size(); i++) {
        prm.in[i].data = in.cols[i].data;
        prm.in[i].valid = in.has_valid[i] ? in.cols[i].valid : nullptr;
        prm.in[i].aux = in.cols[i].aux;
        prm.in[i].offset = in.cols[i].offset;
      }
      prm.out[kOutErr] = err_flags_.p;
      for (size_t k = 0; k < part.out_col.size(); k++) {
        const size_t c = (size_t)part.out_col[k];
        prm.out[kOutFirstCol + 2 * k] = (char*)vals[c]->p + (size_t)base * (size_t)out_width(info.out_cols[c]);
        prm.out[kOutFirstCol + 2 * k + 1] = (char*)vbytes[c]->p + (size_t)base;
      }
      if (n) launch(v, "k_emit", (int)std::min<int64_t>((n + 255) / 256, 256 * 8), prm);
      u.desc.kernels = v.desc.kernels;
      HIP_CHECK(hipStreamSynchronize(stream_));   // v (and its module reference) goes out of scope
    }
    for (int c : part.null_cols) {
      if (!n) continue;
      HIP_CHECK(hipMemsetAsync((char*)vbytes[(size_t)c]->p + (size_t)base, 0, (size_t)n, stream_));
      HIP_CHECK(hipMemsetAsync((char*)vals[(size_t)c]->p + (size_t)base * (size_t)out_width(info.out_cols[(size_t)c]), 0,
                               (size_t)n * (size_t)out_width(info.out_cols[(size_t)c]), stream_));
    }
  }
  timed_end();
  if (!u.mod) throw CometError("Expand: every projection consists of NULL literals only");
  DevTable out = outputs_to_table(u, vals, vbytes, total, [&](int c) { return std::make_pair(&in, c); });
  out.owners.push_back(u.mod);
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  return out;
}

}  // namespace comet
