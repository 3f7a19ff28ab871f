#include "exec.hpp"
#include "shuffle_format.hpp"

#include <cerrno>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>

#include "exec_internal.hpp"

namespace comet {

// `in` with every field of its struct columns as a column of its own behind the real ones (DType::virt_parent / virt_kid name them): what
// GetStructField folds into (codegen.cpp lower_struct_field).  A field's validity already says NULL where its struct is NULL (the Parquet
// scan derives both from one definition level), so the field column is the field's view as it is.
// … and the element column of every List of flat elements (virt_kid 0 of a List: what ListExtract / array_contains address, codegen.cpp list_ref).
static bool list_of_flat(const DType& t) { return t.id == TypeId::List && t.kids.size() == 1 && !t.kids[0].is_nested(); }
DevTable extend_struct_fields(const DevTable& in) {
  bool any = false;
  for (auto& t : in.types) any |= t.id == TypeId::Struct || list_of_flat(t);
  if (!any) return in;
  DevTable x = in;
  for (size_t i = 0; i < in.types.size(); i++) {
    if (list_of_flat(in.types[i]) && in.cols[i].kids.size() == 1) {
      DType kt = in.types[i].kids[0];
      kt.virt_parent = (int)i;
      kt.virt_kid = 0;
      x.types.push_back(kt);
      x.cols.push_back(in.cols[i].kids[0]);
      x.has_valid.push_back(!in.cols[i].kid_has_valid.empty() && in.cols[i].kid_has_valid[0]);
      continue;
    }
    if (in.types[i].id != TypeId::Struct) continue;
    for (size_t k = 0; k < in.types[i].kids.size() && k < in.cols[i].kids.size(); k++) {
      DType kt = in.types[i].kids[k];
      kt.virt_parent = (int)i;
      kt.virt_kid = (int)k;
      x.types.push_back(kt);
      x.cols.push_back(in.cols[i].kids[k]);
      x.has_valid.push_back(k < in.cols[i].kid_has_valid.size() && in.cols[i].kid_has_valid[k]);
    }
  }
  return x;
}

std::vector<DType> extend_struct_field_types(const std::vector<DType>& types) {
  std::vector<DType> x = types;
  for (size_t i = 0; i < types.size(); i++) {
    if (list_of_flat(types[i])) {
      DType kt = types[i].kids[0];
      kt.virt_parent = (int)i;
      kt.virt_kid = 0;
      x.push_back(kt);
      continue;
    }
    if (types[i].id != TypeId::Struct) continue;
    for (size_t k = 0; k < types[i].kids.size(); k++) {
      DType kt = types[i].kids[k];
      kt.virt_parent = (int)i;
      kt.virt_kid = (int)k;
      x.push_back(kt);
    }
  }
  return x;
}

namespace detail {

// Planned pipelines are shared by every task that runs the same plan bytes (a Spark stage = thousands of
// identical createPlan calls): (plan hash, validity pattern) → generated source + compiled code object.
std::mutex g_plan_mu;
std::map<std::string, std::shared_ptr<PlannedVariant>> g_plan_cache;

std::shared_ptr<PlannedVariant> planned_variant(const Operator& plan, uint64_t plan_hash, const std::vector<bool>& has_valid, bool compile,
                                                const std::vector<DType>* source_types, const std::vector<int>* str_fixed_len, const std::vector<int>* dict_id_col) {
  std::string key = std::to_string(plan_hash) + ":" + validity_key(has_valid);
  if (dict_id_col) {
    key += ":D";
    for (int c : *dict_id_col) key += std::to_string(c) + ",";
  }
  if (str_fixed_len) {
    key += ":L";
    for (int l : *str_fixed_len) key += std::to_string(l) + ",";
  }
  if (source_types) {
    key += ":";
    for (auto& t : *source_types) key += t.str() + ",";
  }
  std::shared_ptr<PlannedVariant> pv;
  {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plan_cache.find(key);
    if (it != g_plan_cache.end()) pv = it->second;
  }
  if (!pv) {
    pv = std::make_shared<PlannedVariant>();
    pv->desc = generate_pipeline(plan, has_valid, source_types, str_fixed_len, dict_id_col);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto res = g_plan_cache.emplace(key, pv);
    pv = res.first->second;
  }
  if (compile && !pv->code) {
    auto co = jit_compile(pv->desc.source);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (!pv->code) pv->code = co;
  }
  return pv;
}

}  // namespace detail

// ---------------------------------------------------------------------------------------------
ExecutionContext::ExecutionContext(OperatorP plan, uint64_t plan_hash, std::vector<std::pair<std::string, std::string>> config,
                                   std::vector<InputSource> inputs, int batch_size, int device_id)
    : plan_(std::move(plan)), plan_hash_(plan_hash), config_(std::move(config)), inputs_(std::move(inputs)), batch_size_(batch_size),
      device_id_(device_id) {
  chunk_rows_ = 4 << 20;
  mem_->owner = std::this_thread::get_id();
  for (auto& kv : config_) {
    if (kv.first == "spark.comet.gpu.chunkRows") chunk_rows_ = std::max<long long>(1024, atoll(kv.second.c_str()));
    if (kv.first == "spark.comet.gpu.memory.limit") mem_->dev_limit = std::max<long long>(0, atoll(kv.second.c_str()));
    if (kv.first == "spark.comet.gpu.join.fuseProbe") fuse_probe_ = kv.second != "false" && kv.second != "0";
    if (kv.first == "spark.comet.gpu.join.fuseBuild") fuse_build_ = (kv.second == "false" || kv.second == "0") ? 0 : kv.second == "always" ? 2 : 1;
  }
  if (const char* e = getenv("COMET_JOIN_FUSE_PROBE")) fuse_probe_ = atoi(e) != 0;
  if (const char* e = getenv("COMET_JOIN_FUSE_BUILD")) fuse_build_ = atoi(e);
  if (const char* e = getenv("COMET_GPU_CHUNK_ROWS")) chunk_rows_ = std::max<long long>(1024, atoll(e));
  // Semi-join reduction.  A LeftSemi / LeftAnti join only asks whether its build side HOLDS a key: how often is irrelevant.  An Inner join
  // below it whose output is projected onto columns of ONE of its sides therefore only has to say which rows of that side have a partner
  // — a LeftSemi join, whose output has at most one row per input row instead of one per matching pair (TPC-DS Q95: the ws_wh self-join
  // emits 68 M pairs where 25 M rows carry the same keys), and whose own build side is duplicate-insensitive in turn.  Same SET of rows,
  // so the consumer's answer is unchanged; the reference executes the plan as written (DataFusion has no such rule).
  // (the Scan leaves are bound to the input streams first, in the plan's own depth-first order: the rule may swap a join's children)
  {
    std::function<void(const Operator&)> bind_scans = [&](const Operator& op) {
      if (op.kind == OpKind::Scan) scan_input_.emplace(&op, scan_input_.size());
      for (auto& c : op.children) bind_scans(*c);
    };
    bind_scans(*plan_);
  }
  bool semi_reduction = true;
  for (auto& kv : config_)
    if (kv.first == "spark.comet.gpu.join.semiReduction") semi_reduction = kv.second != "false" && kv.second != "0";
  if (const char* e = getenv("COMET_JOIN_SEMI_REDUCTION")) semi_reduction = atoi(e) != 0;
  if (semi_reduction) {
    // The rewrite changes what the plan's nodes compile to (join type, build side, output width) while the plan BYTES — what plan_hash_ is
    // taken from, and with it every key of the process-wide kernel cache — stay the same: two contexts running the same bytes with and
    // without the rule would share each other's variants.  So every rewritten join is folded into plan_hash_ (which joins, swapped or not).
    uint64_t rewrite_sig = 0;
    int visit = 0;
    std::function<void(Operator&, bool)> reduce = [&](Operator& op, bool set_only) {   // set_only: the consumer looks at the SET of op's rows
      const int my_visit = visit++;
      if (op.kind == OpKind::Projection && set_only && op.children.size() == 1 && op.children[0]->kind == OpKind::HashJoin) {
        Operator& j = *op.children[0];
        if (j.join_type == JoinType::Inner && !j.null_aware_anti && j.children.size() == 2 && j.left_keys.size() == j.right_keys.size()) {
          // which join output columns does the projection read?  (left ++ right; the child schemas are not inferred yet: the left width
          // comes from the largest column index the left keys / the condition can tell apart — so take it from the children instead)
          std::function<int(const Operator&)> width = [&](const Operator& o) -> int {
            switch (o.kind) {
              case OpKind::Scan: return (int)o.scan_fields.size();
              case OpKind::NativeScan: return (int)(o.required_schema.size() + o.partition_schema.size());
              case OpKind::Projection: return (int)o.project_list.size();
              case OpKind::Filter: case OpKind::Sort: case OpKind::Limit: return o.children.size() == 1 ? width(*o.children[0]) : -1;
              case OpKind::HashJoin: {
                if (o.children.size() != 2) return -1;
                const int l = width(*o.children[0]), r = width(*o.children[1]);
                if (l < 0 || r < 0) return -1;
                return (o.join_type == JoinType::LeftSemi || o.join_type == JoinType::LeftAnti) ? l : l + r;
              }
              default: return -1;     // aggregates, windows, expands: not needed for this rule
            }
          };
          const int nl = width(*j.children[0]), nr = width(*j.children[1]);
          bool any_left = false, any_right = false, ok = nl >= 0 && nr >= 0;
          std::function<void(const ExprP&)> refs = [&](const ExprP& e) {
            if (e->kind == ExprKind::Bound) {
              if (e->bound_index < 0 || e->bound_index >= nl + nr) ok = false;
              else (e->bound_index < nl ? any_left : any_right) = true;
            }
            for (auto& c : e->children) refs(c);
          };
          for (auto& e : op.project_list) refs(e);
          if (ok && any_left != any_right) {
            if (any_right) {
              // keep the RIGHT side: swap the children so that it becomes the left of a LeftSemi join
              std::function<ExprP(const ExprP&, bool)> remap = [&](const ExprP& e, bool combined) -> ExprP {
                auto n = std::make_shared<Expr>(*e);
                if (e->kind == ExprKind::Bound) n->bound_index = combined ? (e->bound_index >= nl ? e->bound_index - nl : e->bound_index + nr) : e->bound_index - nl;
                for (auto& c : n->children) c = remap(c, combined);
                return n;
              };
              std::swap(j.children[0], j.children[1]);
              std::swap(j.left_keys, j.right_keys);
              j.build_side = j.build_side == BuildSide::Left ? BuildSide::Right : BuildSide::Left;
              if (j.join_condition) j.join_condition = remap(j.join_condition, true);
              for (auto& e : op.project_list) e = remap(e, false);
            }
            j.join_type = JoinType::LeftSemi;
            rewrite_sig = (rewrite_sig ^ (uint64_t)(2 * my_visit + (any_right ? 1 : 0) + 1)) * 1099511628211ull;
          }
        }
      }
      for (size_t i = 0; i < op.children.size(); i++) {
        bool child_set_only = false;
        switch (op.kind) {
          case OpKind::Projection: case OpKind::Filter: child_set_only = set_only; break;
          case OpKind::HashJoin:
            if ((op.join_type == JoinType::LeftSemi || op.join_type == JoinType::LeftAnti) && !op.null_aware_anti) child_set_only = i == 1 ? true : set_only;
            else child_set_only = set_only;      // duplicates on either side of a join only duplicate its output rows
            break;
          default: break;                        // aggregates, limits, windows, sorts, shuffle writers count their rows
        }
        reduce(*op.children[i], child_set_only);
      }
    };
    reduce(*plan_, false);
    if (rewrite_sig) plan_hash_ ^= rewrite_sig * 0x9E3779B97F4A7C15ull;
  }
  // Scan leaves in depth-first, left-before-right order map to the input streams (planner.rs:1726, :2391)
  std::function<void(const Operator&)> walk = [&](const Operator& op) {
    node_id_[&op] = (int)node_id_.size();
    if (op.kind == OpKind::Scan && !scan_input_.count(&op)) scan_input_.emplace(&op, scan_input_.size());   // (bound above, before the semi-join reduction)
    if (op.kind == OpKind::HashJoin || op.kind == OpKind::NativeScan || op.kind == OpKind::Sort || op.kind == OpKind::Limit || op.kind == OpKind::ShuffleWriter ||
        op.kind == OpKind::Expand || op.kind == OpKind::Window || op.kind == OpKind::Explode)
      has_join_ = true;
    if (op.kind == OpKind::Window) {
      for (int t = 0; t < 2; t++) {
        auto so = std::make_shared<Operator>();
        so->kind = OpKind::Sort;
        so->proto_tag = 103;
        (t == 0 ? window_psort_ : window_osort_)[&op] = so;
        node_id_[so.get()] = (int)node_id_.size();
      }
    }   // sources materialised in HBM
    if (op.kind == OpKind::ShuffleWriter) {
      if (&op != plan_.get()) throw CometError("ShuffleWriter must be the root of a native plan");
      bool computed = false;
      for (auto& e : op.shuffle_hash_exprs) computed |= e->kind != ExprKind::Bound;
      for (auto& k : op.shuffle_sort_orders) computed |= k.child->kind != ExprKind::Bound;
      if (op.shuffle_partitioning == Operator::Partitioning::Range) {
        // range partitioning compares order-preserving key bytes: one synthetic Sort describes the rows' keys, one the boundaries'
        for (int t = 0; t < 2; t++) {
          auto so = std::make_shared<Operator>();
          so->kind = OpKind::Sort;
          so->proto_tag = 103;
          (t == 0 ? range_sort_ : range_bsort_)[&op] = so;
          node_id_[so.get()] = (int)node_id_.size();
        }
      }
      if (computed) {
        // hash expressions that are not plain column references: evaluated by a Projection(child columns ++ expressions) fused
        // over the child (its project_list is filled in when the child's schema is known)
        auto pr = std::make_shared<Operator>();
        pr->kind = OpKind::Projection;
        pr->proto_tag = 101;
        pr->children = op.children;
        shuffle_projs_[&op] = pr;
        node_id_[pr.get()] = (int)node_id_.size();
      }
    }
    if (op.kind == OpKind::HashAgg && &op != plan_.get()) {
      // an aggregate below other operators: materialised too, unless it is the sink of the root chain (checked below)
      nested_aggs_.push_back(&op);
    }
    if (op.kind == OpKind::Unsupported)
      throw CometError(std::string("Operator ") + op_name(op.proto_tag) + " is not supported by the MI355X native engine");
    if (op.kind == OpKind::HashJoin && op.smj) {
      // synthetic Sort over the join output: left keys keep their column indices in left ++ right; a RightOuter join is ordered
      // by the right keys (shifted past the left columns, resolved when the schema is known)
      auto so = std::make_shared<Operator>();
      so->kind = OpKind::Sort;
      so->proto_tag = 103;
      smj_sorts_[&op] = so;
      node_id_[so.get()] = (int)node_id_.size();
    }
    for (auto& c : op.children) walk(*c);
  };
  walk(*plan_);
  // Which sort-merge joins must really deliver their output in key order?  Only those whose row order can reach something that
  // looks at it: the plan's output, a Limit, a shuffle file.  Aggregates, sorts and every join here (sort-merge joins run as hash
  // joins and do not need sorted inputs) forget the order of their inputs, so the sort below them would be wasted work
  // (TPC-DS Q95: five sorts of up to 68 M rows).
  std::function<void(const Operator&, bool)> mark = [&](const Operator& op, bool order_visible) {
    if (op.kind == OpKind::HashJoin && op.smj && order_visible) smj_needs_sort_.insert(&op);
    for (size_t i = 0; i < op.children.size(); i++) {
      bool v = order_visible;
      switch (op.kind) {
        case OpKind::HashAgg: case OpKind::Sort: v = false; break;
        case OpKind::HashJoin:
          // a plain hash join streams its probe side: the probe order shows in the output; a sort-merge join re-sorts (or not)
          v = !op.smj && order_visible && (op.build_side == BuildSide::Left ? i == 1 : i == 0);
          break;
        default: break;   // Projection / Filter / Limit / ShuffleWriter keep their input's order
      }
      mark(*op.children[i], v);
    }
  };
  mark(*plan_, true);
  if (scan_input_.empty() && !has_join_) throw CometError("Plan has no Scan leaf: only Scan-rooted pipelines are supported by the MI355X native engine");
  if (inputs_.size() != scan_input_.size())
    throw CometError("Plan has " + std::to_string(scan_input_.size()) + " Scan leaves but " + std::to_string(inputs_.size()) + " input streams were given");
  // the root chain ends at a Scan or at the first join below it
  root_source_ = plan_.get();
  while (!is_source(*root_source_, plan_.get())) {
    if (root_source_->children.size() != 1) throw CometError(std::string(op_name(root_source_->proto_tag)) + " expects exactly one child");
    root_source_ = root_source_->children[0].get();
  }
  if (root_source_->kind == OpKind::HashAgg) has_join_ = true;   // operators above an aggregate: the aggregate is materialised
  // a Scan with struct / list fields is a materialised source too: its chunk is resident as a whole (children and all) and the chain above
  // it runs over that table, where a struct's fields are columns of their own and nested rows are gathered by index
  if (root_source_->kind == OpKind::Scan)
    for (auto& t : root_source_->scan_fields) has_join_ = has_join_ || t.is_nested();
  // Validate the plan shape eagerly (generated, not compiled) so that unsupported operators fail at createPlan
  // like the reference's planner would on first execute.
  if (!has_join_) {
    in_types_ = root_source_->scan_fields;
    std::vector<bool> none(in_types_.size(), false);
    auto pv = planned_variant(*plan_, plan_hash_, none, false);
    explain_ = pv->desc.explain;
    sink_ = pv->desc.sink;
    if (sink_ == SinkKind::Output)
      for (auto& oc : pv->desc.out_cols) materialize_root_ |= oc.gather_src >= 0 || oc.packed_string || oc.view_src >= 0 || oc.fmt_kind || !oc.concat_cols.empty();   // Utf8 outputs are finished on the device (gather / unpack)
      materialize_root_ |= !pv->desc.derived.empty();      // (split: computed over the resident source)
    // a grouped aggregate keyed by Utf8 columns sees its whole input at once (like a join input): only then can strings longer
    // than the packed 15 bytes be swapped for representative row indices (prepare_dict_keys)
    if (sink_ == SinkKind::AggGrouped && !pv->desc.str_key_cols.empty()) has_join_ = true;
  } else {
    in_types_ = extend_struct_field_types(infer_schema(*root_source_));      // (a struct's fields are columns of their own to the chain above: GetStructField)
    if (plan_.get() != root_source_) {
      std::vector<bool> none(in_types_.size(), false);
      auto pv = planned_variant(*plan_, plan_hash_, none, false, &in_types_);
      explain_ = explain_ + pv->desc.explain;
      sink_ = pv->desc.sink;
    } else {
      sink_ = SinkKind::Output;
    }
  }
}

// Output schema of a sub-plan (no data needed): Scan fields, chain outputs, join = left ++ right (semi/anti: left)
// Where a fused chain stops: Scan leaves and everything whose result is materialised in HBM — joins, Parquet scans, sorts,
// limits, and an aggregate that is not the top of the chain being fused.
bool ExecutionContext::is_source(const Operator& op, const Operator* chain_top) {
  switch (op.kind) {
    case OpKind::Explode: case OpKind::Scan: case OpKind::HashJoin: case OpKind::NativeScan: case OpKind::Sort: case OpKind::Limit: case OpKind::ShuffleWriter: case OpKind::Expand: case OpKind::Window: return true;
    case OpKind::HashAgg: return &op != chain_top;
    default: return false;
  }
}

// Can this join read its probe chain's source directly?  The probe child must be a chain of Filters / Projections over a source
// (Scan, Parquet scan, another join, …), no join key pair may be two Utf8 columns (those may need the string dictionary of hash_join,
// which works on materialised columns), and the fused functor must generate — a computed Utf8 column in the chain's output does not.
// The BUILD child as a Filter / Projection chain over a plain Scan leaf (round 6): the build passes then read the Scan's table — no materialised copy of the
// filtered build side (TPC-DS Q95's two 70 M-row self-join builds: 1.3 ms of k_filter per run).  Same exclusions as on the probe side.
bool ExecutionContext::plan_fused_build(const Operator& join, FusedProbe& fb) {
  if (!fuse_build_ || join.children.size() != 2) return false;
  const bool build_left = join.build_side == BuildSide::Left;
  const Operator& child = *join.children[build_left ? 0 : 1];
  if (child.kind != OpKind::Filter && child.kind != OpKind::Projection) return false;
  const Operator* src = &child;
  while (src->kind != OpKind::Scan) {
    if ((src->kind != OpKind::Filter && src->kind != OpKind::Projection) || src->children.size() != 1) return false;
    src = src->children[0].get();
  }
  for (size_t k = 0; k < join.left_keys.size() && k < join.right_keys.size(); k++) {
    auto strk = [](const ExprP& e) { return e->kind == ExprKind::Bound && (!e->has_dtype || e->dtype.id == TypeId::String || e->dtype.id == TypeId::Bytes); };
    if (strk(join.left_keys[k]) && strk(join.right_keys[k])) return false;
  }
  try {
    fb.source = src;
    fb.fu.src_types = src->scan_fields;
    fb.fu.src_valid.assign(fb.fu.src_types.size(), false);
    fold_chain(child, *src, fb.fu.src_types, fb.fu.cols, fb.fu.preds);
  } catch (const CometError&) {
    return false;
  }
  // By default only chains whose Filters merely drop NULLs are fused: the build passes (run count, bitmap, histogram, scatter) each read the SOURCE, so a
  // selective filter makes them read several times what one k_filter would have left behind (SF100 Q3's customers of one market segment, 20 % of 15 M rows:
  // 6.96 → 7.08 ms fused), while isnotnull(…) keeps nearly every row and the materialised copy is pure overhead (TPC-DS Q95's two self-join builds: 13.9 → 13.35 ms)
  if (fuse_build_ < 2)
    for (auto& p : fb.fu.preds)
      if (p->kind != ExprKind::IsNotNull) return false;
  return true;
}

bool ExecutionContext::plan_fused_probe(const Operator& join, const std::vector<DType>& build_types, PipelineDesc& desc) {
  fused_probe_.erase(&join);
  fused_build_.erase(&join);
  if (!fuse_probe_ || join.children.size() != 2) return false;
  const bool build_left = join.build_side == BuildSide::Left;
  const Operator& child = *join.children[build_left ? 1 : 0];
  if (child.kind != OpKind::Filter && child.kind != OpKind::Projection) return false;
  const Operator* src = &child;
  while (!is_source(*src, &child)) {
    if ((src->kind != OpKind::Filter && src->kind != OpKind::Projection) || src->children.size() != 1) return false;
    src = src->children[0].get();
  }
  for (size_t k = 0; k < join.left_keys.size() && k < join.right_keys.size(); k++) {
    auto strk = [](const ExprP& e) { return e->kind == ExprKind::Bound && (!e->has_dtype || e->dtype.id == TypeId::String || e->dtype.id == TypeId::Bytes); };
    if (strk(join.left_keys[k]) && strk(join.right_keys[k])) return false;
  }
  const std::string explain_before = explain_;
  FusedProbe fp;
  fp.source = src;
  try {
    fp.fu.src_types = infer_schema(*src);
    fp.fu.src_valid.assign(fp.fu.src_types.size(), false);
    fold_chain(child, *src, fp.fu.src_types, fp.fu.cols, fp.fu.preds);
    std::vector<DType> none_t;
    std::vector<bool> bv(build_types.size(), false), none_v;
    FusedProbe fb;
    bool with_build = plan_fused_build(join, fb);
    if (with_build) {
      try {
        desc = generate_join(join, build_left ? build_types : none_t, build_left ? none_t : build_types, build_left ? bv : none_v, build_left ? none_v : bv, &fp.fu, &fb.fu);
      } catch (const CometError&) {
        with_build = false;
      }
    }
    if (!with_build) desc = generate_join(join, build_left ? build_types : none_t, build_left ? none_t : build_types, build_left ? bv : none_v, build_left ? none_v : bv, &fp.fu);
    else fused_build_[&join] = fb;
  } catch (const CometError&) {
    explain_ = explain_before;    // the unfused path reports its own errors
    return false;
  }
  fused_probe_[&join] = fp;
  return true;
}

std::vector<DType> ExecutionContext::infer_schema(const Operator& op) {
  if (op.kind == OpKind::Scan) return op.scan_fields;
  if (op.kind == OpKind::Sort || op.kind == OpKind::Limit) {
    if (op.children.size() != 1) throw CometError(std::string(op_name(op.proto_tag)) + " expects exactly one child");
    std::vector<DType> st = infer_schema(*op.children[0]);
    if (op.kind == OpKind::Sort) {
      std::vector<bool> none(st.size(), false);
      PipelineDesc d = generate_sort_keys(op, st, none);   // validates the sort expressions
      if (compile_in_infer_) jit_compile(d.source);
      explain_ += d.explain;
    } else {
      if (op.limit != -1 && op.offset > op.limit)
        throw CometError("Invalid limit/offset combination: [" + std::to_string(op.limit) + ". " + std::to_string(op.offset) + "]");
      explain_ += "  limit " + std::to_string(op.limit) + " offset " + std::to_string(op.offset) + "\n";
    }
    return st;
  }
  if (op.kind == OpKind::Window) {
    // WindowAggExec / BoundedWindowAggExec (planner.rs:2267-2379): child columns ++ one column per window expression
    if (op.children.size() != 1) throw CometError("Window expects exactly one child");
    std::vector<DType> st = infer_schema(*op.children[0]);
    Operator& ps = *window_psort_.at(&op);
    Operator& os = *window_osort_.at(&op);
    ps.sort_orders.clear();
    for (auto& e : op.window_partition) {
      Operator::SortKey k;
      k.child = e;
      ps.sort_orders.push_back(k);
    }
    os.sort_orders = op.window_order;
    std::vector<bool> none(st.size(), false);
    if (!ps.sort_orders.empty()) { PipelineDesc d = generate_sort_keys(ps, st, none); if (compile_in_infer_) jit_compile(d.source); }
    if (!os.sort_orders.empty()) { PipelineDesc d = generate_sort_keys(os, st, none); if (compile_in_infer_) jit_compile(d.source); }
    std::vector<DType> out = st;
    auto check_range_frame = [&](const Operator::WindowFn& fn) {
      if (!fn.frame_rows && (fn.frame_lower == 1 || fn.frame_upper == 1)) {
          if (op.window_order.size() != 1) throw CometError("Window: a RANGE frame with a value offset needs exactly one ORDER BY expression");
          const ExprP& ok = op.window_order[0].child;
          const DType kt = ok->kind == ExprKind::Bound && ok->bound_index >= 0 && (size_t)ok->bound_index < st.size() ? st[(size_t)ok->bound_index] : DType();
          if (!kt.is_integer())
            throw CometError("Window: RANGE frames with a value offset are supported over one integer ORDER BY column (got " + (kt.id == TypeId::Unknown ? std::string("an expression") : kt.str()) + ")");
          for (const ExprP& r : {fn.frame_lower == 1 ? fn.frame_lower_range : ExprP(), fn.frame_upper == 1 ? fn.frame_upper_range : ExprP()})
            if (r && (!r->dtype.is_integer() || r->lit_null || r->lit_i64 < 0))
              throw CometError("Window: a RANGE frame offset must be a non-negative integer literal of the ORDER BY column's type");
            else if (r) {
              // key ± offset is evaluated in the key's width (window_range_bounds_kernel wraps there): an offset the key type cannot hold
              // would wrap to a small one instead of covering the partition
              const int bits = kt.id == TypeId::Int8 ? 7 : kt.id == TypeId::Int16 ? 15 : kt.id == TypeId::Int32 ? 31 : 63;
              if (bits < 63 && (r->lit_i64 >> bits) != 0)
                throw CometError("Window: RANGE frame offset " + std::to_string(r->lit_i64) + " does not fit the ORDER BY column's type " + kt.str());
            }
        }
    };
    for (auto& fn : op.window_fns) {
      if (fn.is_agg) {
        // SUM / COUNT / AVG of exact types over a frame that starts at the partition start (whole partition, or up to the current row /
        // peer group) — the frames the reference runs with its own Spark-exact accumulators (planner.rs:2953-2972)
        const AggExpr& a = fn.agg;
        // frames: every combination of UNBOUNDED / CURRENT ROW bounds, ROWS frames with literal offsets (n PRECEDING / n FOLLOWING), and
        // RANGE frames with value offsets over ONE integer ORDER BY key (x PRECEDING below, y FOLLOWING above — all the JVM side sends,
        // CometWindowExec.scala:588-632; it keeps DATE / DECIMAL keys in Spark): two binary searches per row over the key
        check_range_frame(fn);
        const bool minmax = a.kind == AggKind::Min || a.kind == AggKind::Max;
        if (minmax && fn.frame_rows && fn.frame_lower == 1 && fn.frame_upper == 1 && fn.frame_upper_off - fn.frame_lower_off > 4096)
          throw CometError("Window: MIN / MAX over a sliding frame wider than 4096 rows is not supported yet");
        if (a.children.size() != 1) throw CometError("Window: aggregate window functions take one argument");
        const ExprP& arg = a.children[0];
        const bool lit = arg->kind == ExprKind::Literal;
        if (!lit && (arg->kind != ExprKind::Bound || arg->bound_index < 0 || (size_t)arg->bound_index >= st.size()))
          throw CometError("Window: the argument of an aggregate window function must be a column (or a literal for COUNT)");
        const DType at = lit ? arg->dtype : st[(size_t)arg->bound_index];
        if (a.kind == AggKind::First || a.kind == AggKind::Last) {
          // FIRST_VALUE / LAST_VALUE (planner.rs:3243-3251): the value of the frame's first / last row — or non-NULL row — of any flat type
          if (lit) throw CometError("Window: FIRST_VALUE / LAST_VALUE of a literal is not supported");
          out.push_back(at);
          continue;
        }
        if (a.kind == AggKind::Count) out.push_back(DType::of(TypeId::Int64));
        else if (lit) throw CometError("Window: SUM / AVG / MIN / MAX of a literal is not supported");
        else if (minmax && (at.is_integer() || at.id == TypeId::Decimal || at.id == TypeId::Date || at.id == TypeId::Timestamp || at.id == TypeId::TimestampNtz)) out.push_back(at);
        else if (a.kind == AggKind::Sum && at.id == TypeId::Decimal && a.dtype.id == TypeId::Decimal) out.push_back(a.dtype);
        else if (a.kind == AggKind::Sum && at.is_integer()) out.push_back(DType::of(TypeId::Int64));
        else if (a.kind == AggKind::Avg && at.id == TypeId::Decimal && a.dtype.id == TypeId::Decimal) out.push_back(a.dtype);
        else throw CometError("Window: aggregate (tag " + std::to_string(a.proto_tag) + ") over " + at.str() + " is not supported yet (SUM / AVG of decimals, SUM of integers, COUNT, MIN / MAX of integers, decimals, dates and timestamps are)");
        continue;
      }
      const std::string& f = fn.func;
      auto int_lit = [](const ExprP& x) { return x->kind == ExprKind::Literal && !x->lit_null && x->dtype.is_integer(); };
      if (f == "row_number" || f == "rank" || f == "dense_rank") out.push_back(DType::of(TypeId::Int32));
      else if (f == "percent_rank" || f == "cume_dist") out.push_back(DType::of(TypeId::Double));
      else if (f == "ntile") {
        if (fn.args.size() != 1 || !int_lit(fn.args[0]) || fn.args[0]->lit_i64 <= 0) throw CometError("ntile expects a positive literal bucket count");
        out.push_back(DType::of(TypeId::Int32));
      } else if (f == "lag" || f == "lead") {
        if (fn.args.size() < 1 || fn.args.size() > 3 || fn.args[0]->kind != ExprKind::Bound || fn.args[0]->bound_index < 0 || (size_t)fn.args[0]->bound_index >= st.size())
          throw CometError(f + " is supported for a column argument");
        if (fn.args.size() >= 2 && !int_lit(fn.args[1])) throw CometError(f + " expects a literal offset");
        if (fn.ignore_nulls && fn.args.size() >= 2 && fn.args[1]->lit_i64 == 0) throw CometError(f + " IGNORE NULLS with offset 0 is not supported");
        if (fn.args.size() == 3 && fn.args[2]->kind != ExprKind::Literal) throw CometError(f + " default value must be a literal");
        if (fn.args.size() == 3 && !fn.args[2]->lit_null) {
          const DType& at = st[(size_t)fn.args[0]->bound_index];
          if (at.id == TypeId::String || at.id == TypeId::Bytes || at.id == TypeId::Bool) throw CometError(f + " with a non-NULL default value over " + at.str() + " is not supported yet");
        }
        out.push_back(st[(size_t)fn.args[0]->bound_index]);
      } else if (f == "nth_value") {
        // nth_value(column, n) over the spec's frame (CometWindowExec.scala:293-306): the frame's n-th row (or n-th non-NULL row)
        if (fn.args.size() != 2 || fn.args[0]->kind != ExprKind::Bound || fn.args[0]->bound_index < 0 || (size_t)fn.args[0]->bound_index >= st.size())
          throw CometError("nth_value is supported for a column argument");
        if (!int_lit(fn.args[1]) || fn.args[1]->lit_i64 <= 0) throw CometError("nth_value expects a positive literal offset");
        check_range_frame(fn);
        out.push_back(st[(size_t)fn.args[0]->bound_index]);
      } else {
        throw CometError(f + " not supported for window function");
      }
    }
    explain_ += "  window: " + std::to_string(op.window_fns.size()) + " function(s), " + std::to_string(op.window_partition.size()) + " partition key(s), " +
                std::to_string(op.window_order.size()) + " order key(s)\n";
    return out;
  }
  if (op.kind == OpKind::Explode) {
    // Explode / posexplode [_outer] (planner.rs:1949-2110: a Projection of the carried columns ++ [pos] ++ the list, then UnnestExec): output =
    // project_list ++ [pos Int32] ++ [element].  The carried columns are a fused Projection over the resident child (evaluated once per INPUT
    // row, then gathered by the output rows' input row); the list must be a column of the child
    if (op.children.size() != 1) throw CometError("Explode expects exactly one child");
    std::vector<DType> st = infer_schema(*op.children[0]);
    const ExprP& lc = op.explode_child;
    if (!lc) throw CometError("Explode operator requires a child expression");
    if (lc->kind != ExprKind::Bound || lc->bound_index < 0 || (size_t)lc->bound_index >= st.size())
      throw CometError("Explode: the exploded expression must be a column of the child (computed arrays are not supported)");
    const DType& lt = st[(size_t)lc->bound_index];
    if (lt.id != TypeId::List || lt.kids.size() != 1) throw CometError("Explode: column " + std::to_string(lc->bound_index) + " is " + lt.str() + ", not a list (maps are not supported)");
    std::vector<DType> out;
    if (!op.project_list.empty()) {
      auto sc = std::make_shared<Operator>();
      sc->kind = OpKind::Scan;
      sc->proto_tag = 100;
      sc->scan_fields = st;
      auto pr = std::make_shared<Operator>();
      pr->kind = OpKind::Projection;
      pr->proto_tag = 101;
      pr->children.push_back(sc);
      pr->project_list = op.project_list;
      node_id_[pr.get()] = (int)node_id_.size();
      std::vector<DType> ext = extend_struct_field_types(st);
      std::vector<bool> none(ext.size(), false);
      PipelineDesc d = generate_pipeline(*pr, none, &ext);
      if (compile_in_infer_) jit_compile(d.source);
      explain_ += d.explain;
      for (auto& c : d.out_cols) { out.push_back(c.type); out.back().virt_parent = out.back().virt_kid = -1; }
      explode_proj_[&op] = pr;
    }
    if (op.explode_position) out.push_back(DType::of(TypeId::Int32));
    out.push_back(lt.kids[0]);
    explain_ += std::string("  explode") + (op.explode_outer ? "_outer" : "") + (op.explode_position ? " with positions" : "") + " of column " + std::to_string(lc->bound_index) + "\n";
    return out;
  }
  if (op.kind == OpKind::Expand) {
    // ExpandExec (operators/expand.rs; planner.rs:1913-1948): every input row yields one output row per projection (grouping sets /
    // rollup / cube, the count(DISTINCT …) rewrite over several columns).  Each projection is a fused Projection over the resident child;
    // all of them write into ONE set of output buffers at their row offset.  NULL literals of Utf8 (or unknown) type — the "not in this
    // grouping set" marker — are not generated code: their rows are simply marked invalid.
    if (op.children.size() != 1) throw CometError("Expand expects exactly one child");
    std::vector<DType> st = infer_schema(*op.children[0]);
    const size_t ncol = op.expand_projections.empty() ? 0 : op.expand_projections[0].size();
    ExpandInfo info;
    std::vector<std::vector<DType>> types(op.expand_projections.size(), std::vector<DType>(ncol));
    std::vector<std::vector<bool>> known(op.expand_projections.size(), std::vector<bool>(ncol, false));
    std::vector<std::vector<int>> gsrc(op.expand_projections.size(), std::vector<int>(ncol, -1));
    auto is_null_lit = [](const ExprP& e) { return e->kind == ExprKind::Literal && e->lit_null; };
    auto scan_of = [&]() {
      auto sc = std::make_shared<Operator>();
      sc->kind = OpKind::Scan;
      sc->proto_tag = 100;
      sc->scan_fields = st;
      return sc;
    };
    // pass 1: types of everything that is not a NULL literal
    for (size_t p = 0; p < op.expand_projections.size(); p++) {
      auto pr = std::make_shared<Operator>();
      pr->kind = OpKind::Projection;
      pr->proto_tag = 101;
      pr->children.push_back(scan_of());
      std::vector<size_t> at;
      for (size_t j = 0; j < ncol; j++)
        if (!is_null_lit(op.expand_projections[p][j])) { pr->project_list.push_back(op.expand_projections[p][j]); at.push_back(j); }
      if (pr->project_list.empty()) continue;
      std::vector<bool> none(st.size(), false);
      PipelineDesc d = generate_pipeline(*pr, none, &st);
      for (size_t k = 0; k < at.size(); k++) {
        types[p][at[k]] = d.out_cols[k].type;
        known[p][at[k]] = true;
        if (d.out_cols[k].view_src >= 0 || d.out_cols[k].fmt_kind || !d.out_cols[k].concat_cols.empty()) throw CometError("Expand: string functions with results of any length are not supported inside a grouping-set projection yet");
        gsrc[p][at[k]] = d.out_cols[k].packed_string ? -2 : d.out_cols[k].gather_src;   // −2: a computed (packed) string
      }
    }
    for (size_t j = 0; j < ncol; j++) {
      OutCol oc;
      bool have = false;
      for (size_t p = 0; p < op.expand_projections.size(); p++) {
        if (!known[p][j]) continue;
        if (!have) { oc.type = types[p][j]; oc.gather_src = gsrc[p][j] == -2 ? -1 : gsrc[p][j]; oc.packed_string = gsrc[p][j] == -2; have = true; }
        else if (types[p][j] != oc.type || gsrc[p][j] != (oc.packed_string ? -2 : oc.gather_src))
          throw CometError("Expand: column " + std::to_string(j) + " differs between the projections (" + oc.type.str() + " vs " + types[p][j].str() + ")");
      }
      if (!have) {
        const DType& lt = op.expand_projections[0][j]->dtype;
        if (lt.id == TypeId::Null || lt.id == TypeId::Unknown) throw CometError("Expand: column " + std::to_string(j) + " is NULL in every projection and carries no type");
        if (lt.id == TypeId::String || lt.id == TypeId::Bytes) throw CometError("Expand: a Utf8 column that is NULL in every projection is not supported yet");
        oc.type = lt;
      }
      oc.nullable = true;
      info.out_cols.push_back(oc);
    }
    // pass 2: the projections as executed
    for (size_t p = 0; p < op.expand_projections.size(); p++) {
      ExpandPart part;
      part.proj = std::make_shared<Operator>();
      part.proj->kind = OpKind::Projection;
      part.proj->proto_tag = 101;
      part.proj->children.push_back(scan_of());
      for (size_t j = 0; j < ncol; j++) {
        const ExprP& e = op.expand_projections[p][j];
        const DType& ut = info.out_cols[j].type;
        if (is_null_lit(e) && (ut.id == TypeId::String || ut.id == TypeId::Bytes)) { part.null_cols.push_back((int)j); continue; }
        if (is_null_lit(e)) {
          auto typed = std::make_shared<Expr>(*e);
          typed->dtype = ut;
          typed->has_dtype = true;
          part.proj->project_list.push_back(typed);
        } else {
          part.proj->project_list.push_back(e);
        }
        part.out_col.push_back((int)j);
      }
      node_id_[part.proj.get()] = (int)node_id_.size();
      if (!part.proj->project_list.empty()) {
        std::vector<bool> none(st.size(), false);
        PipelineDesc d = generate_pipeline(*part.proj, none, &st);
        if (compile_in_infer_) jit_compile(d.source);
      }
      info.parts.push_back(part);
    }
    explain_ += "  expand: " + std::to_string(info.parts.size()) + " projection(s) of " + std::to_string(ncol) + " column(s)\n";
    std::vector<DType> out;
    for (auto& oc : info.out_cols) out.push_back(oc.type);
    expand_info_[&op] = info;
    return out;
  }
  if (op.kind == OpKind::ShuffleWriter) {
    // ShuffleWriterExec (shuffle_writer.rs:60-110): consumes its child, writes the data + index files, yields no batches
    if (op.children.size() != 1) throw CometError("ShuffleWriter expects exactly one child");
    std::vector<DType> st = infer_schema(*op.children[0]);
    if (op.shuffle_num_partitions < 1) throw CometError("ShuffleWriter: num_partitions must be positive");
    if (op.shuffle_num_partitions > 4096) throw CometError("ShuffleWriter: more than 4096 output partitions are not supported yet");
    if (op.shuffle_codec < 0 || op.shuffle_codec > 3) throw CometError("Unsupported shuffle compression codec: " + std::to_string(op.shuffle_codec));
    if (op.shuffle_data_file.empty() || op.shuffle_index_file.empty()) throw CometError("ShuffleWriter: output_data_file / output_index_file missing");
    for (auto& e : op.shuffle_hash_exprs)
      if (e->kind == ExprKind::Bound && (e->bound_index < 0 || (size_t)e->bound_index >= st.size()))
        throw CometError("ShuffleWriter: hash expression references column " + std::to_string(e->bound_index) + " of " + std::to_string(st.size()));
    for (auto& t : st)
      if (expected_format(t) == "?") throw CometError("ShuffleWriter: column type " + t.str() + " is not supported");
    auto sp = shuffle_projs_.find(&op);
    if (sp != shuffle_projs_.end()) {
      Operator& pr = *sp->second;
      pr.project_list.clear();
      for (size_t i = 0; i < st.size(); i++) {
        auto b = std::make_shared<Expr>();
        b->kind = ExprKind::Bound;
        b->proto_tag = 3;
        b->bound_index = (int)i;
        b->dtype = st[i];
        b->has_dtype = true;
        pr.project_list.push_back(b);
      }
      for (auto& e : op.shuffle_hash_exprs)
        if (e->kind != ExprKind::Bound) pr.project_list.push_back(e);
      for (auto& k : op.shuffle_sort_orders)
        if (k.child->kind != ExprKind::Bound) pr.project_list.push_back(k.child);
      st = infer_schema(pr);   // validates (and, under compile_only, compiles) the fused chain; st now includes the computed key columns
    }
    if (op.shuffle_partitioning == Operator::Partitioning::Range) {
      // RangePartition (partitioning.proto:52-56; planner.rs:3298-3358): rows are compared with the boundary rows under sort_orders
      if (op.shuffle_sort_orders.empty()) throw CometError("ShuffleWriter: range partitioning without sort orders");
      for (auto& row : op.shuffle_bounds) {
        if (row.size() != op.shuffle_sort_orders.size()) throw CometError("ShuffleWriter: a range boundary row has " + std::to_string(row.size()) + " values for " +
                                                                           std::to_string(op.shuffle_sort_orders.size()) + " sort orders");
        for (auto& e : row)
          if (e->kind != ExprKind::Literal) throw CometError("ShuffleWriter: range boundaries must be literals");
      }
      if ((int)op.shuffle_bounds.size() + 1 > op.shuffle_num_partitions)
        throw CometError("ShuffleWriter: " + std::to_string(op.shuffle_bounds.size()) + " range boundaries need more than " + std::to_string(op.shuffle_num_partitions) + " partitions");
      Operator& so = *range_sort_.at(&op);
      Operator& sb = *range_bsort_.at(&op);
      so.sort_orders.clear();
      sb.sort_orders.clear();
      std::vector<DType> btypes;
      size_t next = st.size();
      for (auto& k : op.shuffle_sort_orders) next -= k.child->kind != ExprKind::Bound;
      size_t computed_at = next;
      for (size_t i = 0; i < op.shuffle_sort_orders.size(); i++) {
        const Operator::SortKey& k = op.shuffle_sort_orders[i];
        const int col = k.child->kind == ExprKind::Bound ? k.child->bound_index : (int)computed_at++;
        if (col < 0 || (size_t)col >= st.size()) throw CometError("ShuffleWriter: range sort order references column " + std::to_string(col));
        auto mk = [&](int idx, const DType& t) {
          auto b = std::make_shared<Expr>();
          b->kind = ExprKind::Bound;
          b->proto_tag = 3;
          b->bound_index = idx;
          b->dtype = t;
          b->has_dtype = true;
          return b;
        };
        Operator::SortKey a = k, b = k;
        a.child = mk(col, st[(size_t)col]);
        b.child = mk((int)i, st[(size_t)col]);
        so.sort_orders.push_back(a);
        sb.sort_orders.push_back(b);
        btypes.push_back(st[(size_t)col]);
      }
      std::vector<bool> none(st.size(), false), bnone(btypes.size(), true);
      PipelineDesc d1 = generate_sort_keys(so, st, none), d2 = generate_sort_keys(sb, btypes, bnone);
      if (d1.sort_key_bytes != d2.sort_key_bytes) throw CometError("internal: range boundary keys and row keys differ in width");
      if (compile_in_infer_) { jit_compile(d1.source); jit_compile(d2.source); }
    }
    explain_ += "  shuffle writer: " + std::to_string(op.shuffle_num_partitions) + " partition(s), codec " + std::to_string(op.shuffle_codec) + "\n";
    return {};
  }
  if (op.kind == OpKind::NativeScan) {
    std::vector<DType> out;
    for (auto& f : op.required_schema) out.push_back(f.dtype);
    for (auto& f : op.partition_schema) out.push_back(f.dtype);   // Hive partition columns follow the file columns
    explain_ += "  parquet scan: " + std::to_string(op.files.size()) + " file(s), " + std::to_string(out.size()) + " column(s)\n";
    return out;
  }
  if (op.kind == OpKind::HashJoin) {
    if (op.children.size() != 2) throw CometError("HashJoin expects two children");
    // the probe child: fused into the probe kernel when it is a Filter / Projection chain over a source (plan_fused_probe)
    const size_t pi = op.build_side == BuildSide::Left ? 1 : 0;
    std::vector<DType> l, r;
    PipelineDesc d;
    bool fused = false;
    if (pi == 1) {
      l = infer_schema(*op.children[0]);
      fused = plan_fused_probe(op, l, d);
      if (!fused) r = infer_schema(*op.children[1]);
    } else {
      // explain order stays left, right: decide on the left (probe) child once the right (build) schema is known
      const std::string before = explain_;
      explain_.clear();
      r = infer_schema(*op.children[1]);
      const std::string right_explain = explain_;
      explain_ = before;
      fused = plan_fused_probe(op, r, d);
      if (!fused) l = infer_schema(*op.children[0]);
      explain_ += right_explain;
    }
    if (!fused) {
      std::vector<bool> lv(l.size(), false), rv(r.size(), false);
      FusedProbe fb;
      bool with_build = plan_fused_build(op, fb);
      if (with_build) {
        try {
          d = generate_join(op, l, r, lv, rv, nullptr, &fb.fu);
          fused_build_[&op] = fb;
        } catch (const CometError&) {
          with_build = false;
        }
      }
      if (!with_build) d = generate_join(op, l, r, lv, rv);   // validates keys / join type
    }
    if (compile_in_infer_) jit_compile(d.source);
    explain_ += d.explain;
    if (fused) {
      // the logical schema of the fused child, for the sort-merge sort keys below: left ++ right as the join emits it
      const size_t nbuild = pi == 1 ? l.size() : r.size();
      std::vector<DType> probe_types;
      const size_t nprobe = fused_probe_.at(&op).fu.cols.size();
      if (d.out_cols.size() == nbuild + nprobe) {
        for (size_t c = 0; c < nprobe; c++) probe_types.push_back(d.out_cols[(pi == 1 ? nbuild : 0) + c].type);
      } else {
        probe_types.assign(nprobe, DType());     // semi / anti joins emit the left side only; the width is all that is needed
        if (pi == 0) for (size_t c = 0; c < nprobe && c < d.out_cols.size(); c++) probe_types[c] = d.out_cols[c].type;
      }
      (pi == 1 ? r : l) = probe_types;
    }
    if (op.smj) {
      Operator& so = *smj_sorts_.at(&op);
      so.sort_orders.clear();
      const bool by_right = op.join_type == JoinType::RightOuter;
      const std::vector<ExprP>& keys = by_right ? op.right_keys : op.left_keys;
      std::function<ExprP(const ExprP&)> shift = [&](const ExprP& e) -> ExprP {
        auto n = std::make_shared<Expr>(*e);
        if (e->kind == ExprKind::Bound) n->bound_index = e->bound_index + (int)l.size();
        for (auto& c : n->children) c = shift(c);
        return n;
      };
      for (size_t k = 0; k < keys.size(); k++) {
        Operator::SortKey sk;
        sk.child = by_right ? shift(keys[k]) : keys[k];
        if (k < op.smj_sort_options.size()) { sk.descending = op.smj_sort_options[k].first; sk.nulls_last = op.smj_sort_options[k].second; }
        so.sort_orders.push_back(sk);
      }
      explain_ += "  (sort-merge join: output ordered by the join keys)\n";
    }
    std::vector<DType> out;
    for (auto& c : d.out_cols) out.push_back(c.type);
    return out;
  }
  const Operator* src = &op;
  do {
    if (src->children.size() != 1) throw CometError(std::string(op_name(src->proto_tag)) + " expects exactly one child");
    src = src->children[0].get();
  } while (!is_source(*src, &op));
  std::vector<DType> st = extend_struct_field_types(infer_schema(*src));      // (as run_chain_to_device will see the source: struct fields as columns of their own)
  std::vector<bool> none(st.size(), false);
  PipelineDesc d = generate_pipeline(op, none, &st);
  if (compile_in_infer_) jit_compile(d.source);
  explain_ += d.explain;
  std::vector<DType> out;
  for (auto& c : d.out_cols) { out.push_back(c.type); out.back().virt_parent = out.back().virt_kid = -1; }
  return out;
}

ExecutionContext::~ExecutionContext() {
  mem_->owner = std::this_thread::get_id();   // the buffers die on this thread (after this body): their bytes go back to the manager from here
  // Dropping the context releases the input streams back to their producer (scan.rs:41-44).
  for (auto& in : inputs_) {
    if (in.host && in.host->release) in.host->release(in.host);
    if (in.dev && in.dev->release) in.dev->release(in.dev);
  }
  for (auto& st : staging_)
    if (st && st->busy) {
      (void)hipEventSynchronize(st->busy);
      pool_put_event(device_id_, st->busy);
      st->busy = nullptr;
    }
  if (stream_) {
    (void)hipStreamSynchronize(stream_);  // pooled buffers go back only once the stream is idle
    for (auto& pr : timed_) { pool_put_event(device_id_, pr.first); pool_put_event(device_id_, pr.second); }
    for (hipEvent_t& e : aux_ev_) if (e) { pool_put_event(device_id_, e); e = nullptr; }
    for (auto& p : kt_pending_) { pool_put_event(device_id_, p.a); pool_put_event(device_id_, p.b); }
    kt_pending_.clear();
    pool_put_stream(device_id_, stream_);
  }
}

const std::string& ExecutionContext::explain() { return explain_; }

// Would createPlan accept this plan?  Everything createPlan validates — decoding, operator and expression support, types — without
// compiling anything and without a GPU: planning a stage on the driver can ask before it commits to the native path.
std::string ExecutionContext::check_only(OperatorP plan, uint64_t plan_hash) {
  size_t nscan = 0;
  std::function<void(const Operator&)> cnt = [&](const Operator& op) {
    if (op.kind == OpKind::Scan) nscan++;
    for (auto& c : op.children) cnt(*c);
  };
  cnt(*plan);
  std::vector<InputSource> ins(nscan);
  ExecutionContext ctx(plan, plan_hash, {}, ins, 8192, 0);
  return ctx.explain_;
}

std::string ExecutionContext::compile_only(OperatorP plan, uint64_t plan_hash) {
  // count Scan leaves to fabricate the (never used) input list
  size_t nscan = 0;
  std::function<void(const Operator&)> cnt = [&](const Operator& op) {
    if (op.kind == OpKind::Scan) nscan++;
    for (auto& c : op.children) cnt(*c);
  };
  cnt(*plan);
  std::vector<InputSource> ins(nscan);
  ExecutionContext ctx(plan, plan_hash, {}, ins, 8192, 0);
  std::vector<bool> none(ctx.in_types_.size(), false);
  if (ctx.has_join_) {
    ctx.compile_in_infer_ = true;
    ctx.explain_.clear();
    ctx.infer_schema(*ctx.root_source_);
    if (ctx.plan_.get() != ctx.root_source_) {
      auto pv = planned_variant(*ctx.plan_, ctx.plan_hash_, none, true, &ctx.in_types_);
      ctx.explain_ += pv->desc.explain;
    }
  } else {
    planned_variant(*ctx.plan_, ctx.plan_hash_, none, true);
  }
  return ctx.explain_;
}

Variant& ExecutionContext::variant_for(const std::vector<bool>& has_valid, const std::vector<int>& str_fixed_len) {
  std::string key = validity_key(has_valid) + ":";
  bool any_fixed = false;
  for (int l : str_fixed_len) { key += std::to_string(l) + ","; any_fixed |= l >= 0; }
  auto it = variants_.find(key);
  if (it != variants_.end()) return it->second;
  auto pv = planned_variant(*plan_, plan_hash_, has_valid, true, has_join_ ? &in_types_ : nullptr, any_fixed ? &str_fixed_len : nullptr,
                            dict_id_col_.empty() ? nullptr : &dict_id_col_);
  Variant v;
  v.desc = pv->desc;
  note_sites(v.desc);
  v.mod = jit_load(pv->code);
  auto res = variants_.emplace(key, std::move(v));
  return res.first->second;
}

void ExecutionContext::launch(Variant& v, const char* kernel, int grid, CometKParams& prm, int block) {
  hipFunction_t fn = v.mod->fn(kernel);
  void* args[] = {&prm};
  // COMET_KERNEL_TIMES=1 / comet_set_kernel_times(1) (a measurement switch, off by default): an event pair around EVERY generated-kernel launch, summed per kernel
  // name (comet_plan_kernel_times) — what bench.py's Q3 / Q95 rooflines name their dominant kernel from
  if (!g_kernel_times.load(std::memory_order_relaxed)) {
    HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, 0, stream_, args, nullptr));
    return;
  }
  hipEvent_t a = pool_get_event(device_id_), b = pool_get_event(device_id_);
  HIP_CHECK(hipEventRecord(a, stream_));
  HIP_CHECK(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, 0, stream_, args, nullptr));
  HIP_CHECK(hipEventRecord(b, stream_));
  kt_pending_.push_back({kernel, a, b});
}

std::atomic<int> g_kernel_times{getenv("COMET_KERNEL_TIMES") != nullptr && atoi(getenv("COMET_KERNEL_TIMES")) != 0 ? 1 : 0};

// per-kernel-name totals of the launches timed so far, as JSON: {"k_jprobe": {"ms": 1.2, "calls": 3}, …}
std::string ExecutionContext::kernel_times_json() {
  if (!kt_pending_.empty()) {
    HIP_CHECK(hipStreamSynchronize(stream_));
    for (auto& p : kt_pending_) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { kernel_times_[p.name].first += ms; kernel_times_[p.name].second++; }
      pool_put_event(device_id_, p.a);
      pool_put_event(device_id_, p.b);
    }
    kt_pending_.clear();
  }
  std::string s = "{";
  for (auto& kv : kernel_times_) {
    char buf[160];
    snprintf(buf, sizeof buf, "%s\"%s\": {\"ms\": %.6f, \"calls\": %lld}", s.size() > 1 ? ", " : "", kv.first.c_str(), kv.second.first, (long long)kv.second.second);
    s += buf;
  }
  return s + "}";
}

// Utf8 group keys longer than the 15 bytes that fit the packed key words: replace them by representative row indices
// (strdict_kernels.hip).  `src` is the aggregate's complete, resident input; an Int64 index column is appended per long key column
// and the pipeline is generated with dict_id_col so that it groups on the index and emits it as the gather index of the string.
void ExecutionContext::prepare_dict_keys(DevTable& src) {
  if (src.rows == 0) return;
  std::vector<bool> none(in_types_.size(), false);
  auto pv = planned_variant(*plan_, plan_hash_, none, false, &in_types_);
  std::vector<int> key_cols = pv->desc.str_key_cols;
  std::sort(key_cols.begin(), key_cols.end());
  key_cols.erase(std::unique(key_cols.begin(), key_cols.end()), key_cols.end());
  if (key_cols.empty()) return;
  const int64_t n = src.rows;
  std::vector<int> id_col(src.cols.size(), -1);
  bool any = false;
  // Packed keys (≤ 15 bytes) are cheaper, but one long column — or a result that must stay on the device, where packed strings
  // cannot be expanded by the host — switches every Utf8 key column of the aggregate to row indices.
  bool need = device_result_;
  for (int c : key_cols) {
    const DeviceColumnView& sc = src.cols[(size_t)c];
    if (sc.offset != 0) return;     // sliced producer arrays keep the packed path (and its 15-byte limit)
    if (need) break;
    if (sc.fixed_len >= 0 && sc.fixed_len <= 15) continue;   // already known to hold values of one short length (TPC-H Q1's flag columns): nothing to measure
    uint32_t* mx = (uint32_t*)err_flags_.p + (kErrBytes / 4 - 1);   // last word of the error/aux block: scratch
    HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
    if (comet_launch_str_max_len((const int32_t*)sc.data, n, mx, stream_) != 0) throw CometError("string keys: launch failed");
    uint32_t longest = 0;
    read_small(&longest, mx, 4);
    HIP_CHECK(hipMemsetAsync(mx, 0, 4, stream_));
    need = longest > 15;
  }
  if (!need || src.cols.size() + key_cols.size() > COMET_MAX_IN) return;
  for (int c : key_cols) {
    const DeviceColumnView& sc = src.cols[(size_t)c];
    if (n >= ((int64_t)1 << 32) - 1) throw CometError("Utf8 group keys longer than 15 bytes over more than 2^32 rows are not supported");
    int64_t slots = 1024;
    while (slots < 2 * n) slots <<= 1;
    DevBuf table;   // only needed while the indices are computed
    table.ensure((size_t)slots * 4);
    HIP_CHECK(hipMemsetAsync(table.p, 0, (size_t)slots * 4, stream_));
    auto rep = std::make_shared<DevBuf>();
    rep->ensure((size_t)n * 8 + 16);
    if (comet_launch_str_dict_build((const int32_t*)sc.data, (const uint8_t*)sc.aux, src.has_valid[(size_t)c] ? sc.valid : nullptr, n, (uint32_t*)table.p, slots,
                                    (int64_t*)rep->p, stream_) != 0)
      throw CometError("string keys: launch failed");
    HIP_CHECK(hipStreamSynchronize(stream_));   // `table` goes back to the pool here
    DeviceColumnView idv;
    idv.data = rep->p;
    idv.valid = sc.valid;
    id_col[(size_t)c] = (int)src.cols.size();
    src.types.push_back(DType::of(TypeId::Int64));
    src.cols.push_back(idv);
    src.has_valid.push_back(src.has_valid[(size_t)c]);
    src.owners.push_back(rep);
    any = true;
  }
  if (!any) return;
  id_col.resize(src.cols.size(), -1);
  dict_id_col_ = id_col;
  in_types_ = src.types;
  dict_src_ = src;   // the emit step gathers the key strings from here
}

// dense outputs written by an emit kernel (values + validity BYTES) → Arrow-layout device table (validity bitmaps)
// out row k = source string idx[k] (Arrow Utf8: int32 offsets + bytes), any length: lengths → scan → copy
void ExecutionContext::take_utf8(const DeviceColumnView& src, const uint32_t* idx, const uint8_t* ok_bytes, const uint8_t* src_valid_bits, int64_t rows,
                                 DeviceColumnView& out, std::vector<std::shared_ptr<void>>& owners) {
  auto offsets = std::make_shared<DevBuf>();
  offsets->ensure((size_t)(rows + 1) * 4 + 16);
  auto data = std::make_shared<DevBuf>();
  if (rows == 0) {
    HIP_CHECK(hipMemsetAsync(offsets->p, 0, 4, stream_));
    data->ensure(16);
  } else {
    DevBuf lengths, tiles;
    lengths.ensure((size_t)rows * 4 + 16);
    tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
    const int32_t* offs = (const int32_t*)src.data + src.offset;
    if (comet_launch_take_utf8_lengths(offs, idx, ok_bytes, src_valid_bits, rows, (uint32_t*)lengths.p, stream_) != 0) throw CometError("take_utf8: launch failed");
    pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)offsets->p, stream_);
    int32_t total = 0;
    read_small(&total, (char*)offsets->p + (size_t)rows * 4, 4);
    if (total < 0) throw CometError("Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
    data->ensure((size_t)std::max(total, 1) + 16);
    if (comet_launch_take_utf8_copy(offs, (const uint8_t*)src.aux, idx, ok_bytes, src_valid_bits, rows, (const int32_t*)offsets->p, (uint8_t*)data->p, stream_) != 0)
      throw CometError("take_utf8: launch failed");
    HIP_CHECK(hipStreamSynchronize(stream_));   // lengths / tiles go back to the pool
  }
  out.data = offsets->p;
  out.aux = data->p;
  out.offset = 0;
  owners.push_back(offsets);
  owners.push_back(data);
}

DeviceColumnView ExecutionContext::take_column(const DeviceColumnView& src, const DType& t, bool has_valid, const uint32_t* idx, const uint8_t* ok_bytes, int64_t rows,
                                               bool& out_has_valid, std::vector<std::shared_ptr<void>>& owners) {
  DeviceColumnView out;
  out_has_valid = false;
  if (src.offset != 0 && t.id != TypeId::String && t.id != TypeId::Bytes) throw CometError("take: a column with a non-zero Arrow offset is not supported here");
  auto take_validity = [&]() {
    if (!has_valid || !src.valid) return;
    auto bm = std::make_shared<DevBuf>();
    bm->ensure((size_t)((rows + 7) / 8) + 16);
    if (rows && comet_launch_take(0, src.valid, idx, rows, bm->p, stream_) != 0) throw CometError("take: validity");
    out.valid = (const uint8_t*)bm->p;
    out_has_valid = true;
    owners.push_back(bm);
  };
  if (t.id == TypeId::Struct) {
    for (size_t k = 0; k < t.kids.size(); k++) {
      bool hv = false;
      out.kids.push_back(take_column(src.kids.at(k), t.kids[k], k < src.kid_has_valid.size() && src.kid_has_valid[k], idx, ok_bytes, rows, hv, owners));
      out.kid_has_valid.push_back(hv ? 1 : 0);
    }
    out.kid_rows = rows;
    take_validity();
    return out;
  }
  if (t.is_listlike()) {
    // lengths of the taken rows → new offsets → the source element index of every output element → the element column taken by those
    auto offsets = std::make_shared<DevBuf>();
    offsets->ensure((size_t)(rows + 1) * 4 + 16);
    int32_t total = 0;
    if (rows == 0) {
      HIP_CHECK(hipMemsetAsync(offsets->p, 0, 4, stream_));
    } else {
      DevBuf lengths, tiles;
      lengths.ensure((size_t)rows * 4 + 16);
      tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
      if (comet_launch_take_utf8_lengths((const int32_t*)src.data, idx, ok_bytes, has_valid ? src.valid : nullptr, rows, (uint32_t*)lengths.p, stream_) != 0)
        throw CometError("take (list): launch failed");
      pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)offsets->p, stream_);
      read_small(&total, (char*)offsets->p + (size_t)rows * 4, 4);
      if (total < 0) throw CometError("List column exceeds 2^31 elements");
      HIP_CHECK(hipStreamSynchronize(stream_));   // lengths / tiles go back to the pool
    }
    auto eidx = std::make_shared<DevBuf>();
    eidx->ensure((size_t)std::max(total, 1) * 4 + 16);
    if (rows && comet_launch_take_list_indices((const int32_t*)src.data, idx, rows, (const int32_t*)offsets->p, (uint32_t*)eidx->p, stream_) != 0)
      throw CometError("take (list): launch failed");
    bool hv = false;
    out.kids.push_back(take_column(src.kids.at(0), t.kids.at(0), !src.kid_has_valid.empty() && src.kid_has_valid[0], (const uint32_t*)eidx->p, nullptr, total, hv, owners));
    out.kid_has_valid.push_back(hv ? 1 : 0);
    out.kid_rows = total;
    out.data = offsets->p;
    owners.push_back(offsets);
    owners.push_back(eidx);
    take_validity();
    return out;
  }
  if (t.id == TypeId::String || t.id == TypeId::Bytes) {
    take_utf8(src, idx, ok_bytes, has_valid ? src.valid : nullptr, rows, out, owners);
    take_validity();
    return out;
  }
  const int w = t.id == TypeId::Bool ? 0 : fixed_width(t);
  auto vals = std::make_shared<DevBuf>();
  vals->ensure((w ? (size_t)std::max<int64_t>(rows, 1) * (size_t)w : (size_t)((rows + 7) / 8)) + 16);
  if (rows && comet_launch_take(w, src.data, idx, rows, vals->p, stream_) != 0) throw CometError("take: unsupported width");
  out.data = vals->p;
  owners.push_back(vals);
  take_validity();
  return out;
}

DevTable ExecutionContext::outputs_to_table(Variant& v, const std::vector<std::shared_ptr<DevBuf>>& vals,
                                            const std::vector<std::shared_ptr<DevBuf>>& valid_bytes, int64_t rows, const GatherSource& gather_source) {
  const PipelineDesc& d = v.desc;
  DevTable t;
  t.rows = rows;
  for (size_t j = 0; j < d.out_cols.size(); j++) {
    const OutCol& oc = d.out_cols[j];
    DeviceColumnView cv;
    cv.data = vals[j]->p;
    t.owners.push_back(vals[j]);
    bool hv = false;
    if (oc.packed_string) {
      // packed ≤ 15-byte strings (computed values, short group keys) → offsets + bytes: lengths, prefix sum, copy
      DevBuf lengths, tiles;
      auto offsets = std::make_shared<DevBuf>(), bytes = std::make_shared<DevBuf>();
      lengths.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
      tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
      offsets->ensure((size_t)(rows + 2) * 4);
      if (rows == 0) HIP_CHECK(hipMemsetAsync(offsets->p, 0, 8, stream_));
      if (comet_launch_str16_lengths(vals[j]->p, (oc.nullable && rows) ? (const uint8_t*)valid_bytes[j]->p : nullptr, rows, (uint32_t*)lengths.p, stream_) != 0)
        throw CometError("packed strings: launch failed");
      if (rows) pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)offsets->p, stream_);
      int32_t total = 0;
      if (rows) read_small(&total, (char*)offsets->p + (size_t)rows * 4, 4);
      bytes->ensure((size_t)total + 16);
      if (comet_launch_str16_copy(vals[j]->p, (const int32_t*)offsets->p, rows, (uint8_t*)bytes->p, stream_) != 0) throw CometError("packed strings: launch failed");
      HIP_CHECK(hipStreamSynchronize(stream_));   // lengths / tiles go back to the pool
      cv.data = offsets->p;
      cv.aux = bytes->p;
      t.owners.push_back(offsets);
      t.owners.push_back(bytes);
    }
    if (!oc.concat_cols.empty()) {
      // the emit kernel wrote source row indices: the parts' lengths per row, prefix sum, the parts' bytes one after the other
      if (!gather_source) throw CometError("internal: concat column without a source table");
      const uint8_t* okb = (oc.nullable && rows) ? (const uint8_t*)valid_bytes[j]->p : nullptr;
      CometConcatArgs ca;
      memset(&ca, 0, sizeof ca);
      ca.n = (int32_t)oc.concat_cols.size();
      std::string lits;
      std::vector<size_t> lit_at(oc.concat_cols.size(), 0);
      for (size_t k = 0; k < oc.concat_cols.size(); k++)
        if (oc.concat_cols[k] < 0) { lit_at[k] = lits.size(); lits += oc.concat_lits[k]; lits.append((16 - lits.size() % 16) % 16, '\0'); }
      auto lit_dev = std::make_shared<DevBuf>();
      lit_dev->ensure(lits.size() + 16);
      if (!lits.empty()) HIP_CHECK(hipMemcpyAsync(lit_dev->p, lits.data(), lits.size(), hipMemcpyHostToDevice, stream_));
      for (size_t k = 0; k < oc.concat_cols.size(); k++) {
        if (oc.concat_cols[k] < 0) {
          ca.lit_len[k] = (int32_t)oc.concat_lits[k].size();
          ca.bytes[k] = (const uint8_t*)lit_dev->p + lit_at[k];
          continue;
        }
        auto src = gather_source(oc.concat_cols[k]);
        const DeviceColumnView& sc = src.first->cols[(size_t)src.second];
        if (!sc.data) throw CometError("internal: concat over a column without offsets");
        ca.offs[k] = (const int32_t*)sc.data;
        ca.bytes[k] = (const uint8_t*)sc.aux;
        ca.first[k] = sc.offset;
      }
      DevBuf lengths, tiles;
      auto offsets = std::make_shared<DevBuf>(), bytes = std::make_shared<DevBuf>();
      lengths.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
      tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
      offsets->ensure((size_t)(rows + 2) * 4);
      if (rows == 0) HIP_CHECK(hipMemsetAsync(offsets->p, 0, 8, stream_));
      if (comet_launch_concat_lengths(&ca, (const uint32_t*)vals[j]->p, okb, rows, (uint32_t*)lengths.p, stream_) != 0) throw CometError("concat: launch failed");
      if (rows) pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)offsets->p, stream_);
      int32_t total = 0;
      if (rows) read_small(&total, (char*)offsets->p + (size_t)rows * 4, 4);
      if (total < 0) throw CometError("Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
      bytes->ensure((size_t)total + 16);
      if (comet_launch_concat_copy(&ca, (const uint32_t*)vals[j]->p, okb, rows, (const int32_t*)offsets->p, (uint8_t*)bytes->p, stream_) != 0) throw CometError("concat: launch failed");
      HIP_CHECK(hipStreamSynchronize(stream_));   // lengths / tiles / the literals go back to the pool
      cv.data = offsets->p;
      cv.aux = bytes->p;
      t.owners.push_back(offsets);
      t.owners.push_back(bytes);
    }
    if (oc.fmt_kind) {
      // the emit kernel wrote one value per row (i128): lengths of the written-out values, prefix sum, digits
      const uint8_t* okb = (oc.nullable && rows) ? (const uint8_t*)valid_bytes[j]->p : nullptr;
      DevBuf lengths, tiles;
      auto offsets = std::make_shared<DevBuf>(), bytes = std::make_shared<DevBuf>();
      lengths.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
      tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
      offsets->ensure((size_t)(rows + 2) * 4);
      if (rows == 0) HIP_CHECK(hipMemsetAsync(offsets->p, 0, 8, stream_));
      if (comet_launch_strfmt_lengths(oc.fmt_kind, oc.fmt_arg, vals[j]->p, okb, rows, (uint32_t*)lengths.p, stream_) != 0) throw CometError("formatted strings: launch failed");
      if (rows) pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)offsets->p, stream_);
      int32_t total = 0;
      if (rows) read_small(&total, (char*)offsets->p + (size_t)rows * 4, 4);
      if (total < 0) throw CometError("Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
      bytes->ensure((size_t)total + 16);
      if (comet_launch_strfmt_write(oc.fmt_kind, oc.fmt_arg, vals[j]->p, okb, rows, (const int32_t*)offsets->p, (uint8_t*)bytes->p, stream_) != 0) throw CometError("formatted strings: launch failed");
      HIP_CHECK(hipStreamSynchronize(stream_));   // lengths / tiles go back to the pool
      cv.data = offsets->p;
      cv.aux = bytes->p;
      t.owners.push_back(offsets);
      t.owners.push_back(bytes);
    }
    if (oc.view_src >= 0) {
      // the emit kernel wrote one strview per row (source row, byte slice, pad characters): lengths, prefix sum, copy
      if (!gather_source) throw CometError("internal: string-view column without a source table");
      auto src = gather_source(oc.view_src);
      const DeviceColumnView& sc = src.first->cols[(size_t)src.second];
      if (sc.fixed_len >= 0 && !sc.data) throw CometError("internal: string view over a column without offsets");
      const uint8_t* okb = (oc.nullable && rows) ? (const uint8_t*)valid_bytes[j]->p : nullptr;
      const uint8_t* pat = (const uint8_t*)oc.pad_pattern.data();
      const int32_t patn = (int32_t)oc.pad_pattern.size();
      DevBuf lengths, tiles;
      auto offsets = std::make_shared<DevBuf>(), bytes = std::make_shared<DevBuf>();
      lengths.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
      tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
      offsets->ensure((size_t)(rows + 2) * 4);
      if (rows == 0) HIP_CHECK(hipMemsetAsync(offsets->p, 0, 8, stream_));
      const int32_t* src_offs = (const int32_t*)sc.data + sc.offset;
      if (oc.case_mode ? comet_launch_strcase_lengths(vals[j]->p, okb, src_offs, (const uint8_t*)sc.aux, rows, oc.case_mode, (uint32_t*)lengths.p, stream_) != 0
                       : comet_launch_strview_lengths(vals[j]->p, okb, rows, pat, patn, (uint32_t*)lengths.p, stream_) != 0)
        throw CometError("string view: launch failed");
      if (rows) pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)offsets->p, stream_);
      int32_t total = 0;
      if (rows) read_small(&total, (char*)offsets->p + (size_t)rows * 4, 4);
      if (total < 0) throw CometError("Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
      bytes->ensure((size_t)total + 16);
      if (oc.case_mode ? comet_launch_strcase_write(vals[j]->p, okb, src_offs, (const uint8_t*)sc.aux, rows, oc.case_mode, (const int32_t*)offsets->p, (uint8_t*)bytes->p, stream_) != 0
                       : comet_launch_strview_copy(vals[j]->p, okb, src_offs, (const uint8_t*)sc.aux, rows, pat, patn, oc.pad_left ? 1 : 0, (const int32_t*)offsets->p, (uint8_t*)bytes->p,
                                                   stream_) != 0)
        throw CometError("string view: launch failed");
      HIP_CHECK(hipStreamSynchronize(stream_));   // lengths / tiles go back to the pool
      cv.data = offsets->p;
      cv.aux = bytes->p;
      t.owners.push_back(offsets);
      t.owners.push_back(bytes);
    }
    if (oc.gather_src >= 0) {
      // the emit kernel wrote source row indices: gather the strings now
      if (!gather_source) throw CometError("internal: gathered Utf8 column without a source table");
      auto src = gather_source(oc.gather_src);
      const DeviceColumnView& sc = src.first->cols[(size_t)src.second];
      if (oc.type.is_nested()) {
        // a nested column passed through: its rows — children and all — by the source row indices the emit kernel wrote.  (The output's own
        // validity is the kernel's ok byte, as for every column; the children keep theirs.)
        bool hv_unused = false;
        cv = take_column(sc, oc.type, false, (const uint32_t*)vals[j]->p, (oc.nullable && rows) ? (const uint8_t*)valid_bytes[j]->p : nullptr, rows, hv_unused, t.owners);
      } else {
        take_utf8(sc, (const uint32_t*)vals[j]->p, (oc.nullable && rows) ? (const uint8_t*)valid_bytes[j]->p : nullptr, nullptr, rows, cv, t.owners);
      }
    }
    if (oc.type.id == TypeId::Bool) {
      // kernels store booleans as bytes; Arrow wants bits
      auto bits = std::make_shared<DevBuf>();
      bits->ensure((size_t)((rows + 7) / 8) + 16);
      CometKParams pk;
      memset(&pk, 0, sizeof pk);
      pk.n = rows;
      pk.out[0] = vals[j]->p;
      pk.out[1] = bits->p;
      if (rows) launch(v, "k_pack", (int)std::min<int64_t>((rows + 255) / 256, 2048), pk);
      cv.data = bits->p;
      t.owners.push_back(bits);
    }
    if (oc.nullable && rows) {
      auto bm = std::make_shared<DevBuf>();
      bm->ensure((size_t)((rows + 7) / 8) + 16);
      CometKParams pk;
      memset(&pk, 0, sizeof pk);
      pk.n = rows;
      pk.out[0] = valid_bytes[j]->p;
      pk.out[1] = bm->p;
      launch(v, "k_pack", (int)std::min<int64_t>((rows + 255) / 256, 2048), pk);
      cv.valid = (const uint8_t*)bm->p;
      t.owners.push_back(bm);
      t.owners.push_back(valid_bytes[j]);
      hv = true;
    }
    t.types.push_back(oc.type);
    t.types.back().virt_parent = t.types.back().virt_kid = -1;      // (a struct field's column was the SOURCE's: the result is a column like any other)
    t.cols.push_back(cv);
    t.has_valid.push_back(hv);
  }
  return t;
}

// A derived Utf8 column (DerivedCol kind 3): a string function of a source column with new bytes (device/strfn.hpp) — lengths, prefix sum, bytes.
void ExecutionContext::extend_derived_strfn(DevTable& in, const DerivedCol& dc) {
  if (dc.src < 0 || (size_t)dc.src >= in.cols.size()) throw CometError("internal: unknown derived column");
  const DeviceColumnView sc = in.cols[(size_t)dc.src];
  const bool hv = in.has_valid[(size_t)dc.src];
  const int64_t rows = in.rows;
  if (!sc.data && rows) throw CometError("a string function over a Utf8 column without offsets is not supported");
  if (hv && sc.offset != 0) throw CometError("a string function over a sliced Utf8 column with NULLs is not supported yet");
  DevBuf args, lengths, tiles, flag;
  auto offsets = std::make_shared<DevBuf>(), bytes = std::make_shared<DevBuf>();
  const size_t na = dc.arg_a.size(), nb = dc.arg_b.size();
  args.ensure(na + nb + 16);
  if (na) HIP_CHECK(hipMemcpyAsync(args.p, dc.arg_a.data(), na, hipMemcpyHostToDevice, stream_));
  if (nb) HIP_CHECK(hipMemcpyAsync((char*)args.p + na, dc.arg_b.data(), nb, hipMemcpyHostToDevice, stream_));
  flag.ensure(16);
  HIP_CHECK(hipMemsetAsync(flag.p, 0, 4, stream_));
  lengths.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
  tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
  offsets->ensure((size_t)(rows + 2) * 4);
  HIP_CHECK(hipMemsetAsync(offsets->p, 0, 8, stream_));
  HIP_CHECK(hipStreamSynchronize(stream_));      // (the literals came from pageable memory)
  const int32_t* offs = (const int32_t*)sc.data + sc.offset;
  const uint8_t* vbits = hv ? sc.valid : nullptr;
  const uint8_t *a = (const uint8_t*)args.p, *b = (const uint8_t*)args.p + na;
  if (comet_launch_strfn_len(dc.op, offs, (const uint8_t*)sc.aux, vbits, sc.offset, rows, a, (int32_t)na, b, (int32_t)nb, dc.arg_k, (uint32_t*)lengths.p, (uint32_t*)flag.p, stream_) != 0)
    throw CometError("string function: launch failed");
  if (rows) pq_launch_u32_scan((const uint32_t*)lengths.p, rows, (uint64_t*)tiles.p, (int32_t*)offsets->p, stream_);
  int32_t total = 0;
  uint32_t too_long = 0;
  if (rows) read_small(&total, (char*)offsets->p + (size_t)rows * 4, 4);
  read_small(&too_long, flag.p, 4);
  if (total < 0 || too_long) throw CometError("Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
  bytes->ensure((size_t)total + 16);
  if (comet_launch_strfn_write(dc.op, offs, (const uint8_t*)sc.aux, vbits, sc.offset, rows, a, (int32_t)na, b, (int32_t)nb, dc.arg_k, (const int32_t*)offsets->p, (uint8_t*)bytes->p, stream_) != 0)
    throw CometError("string function: launch failed");
  HIP_CHECK(hipStreamSynchronize(stream_));      // the literals, lengths and tiles go back to the pool
  DeviceColumnView cv;
  cv.data = offsets->p;
  cv.aux = bytes->p;
  cv.valid = sc.valid;
  in.types.push_back(dc.type);
  in.cols.push_back(cv);
  in.has_valid.push_back(hv);
  in.owners.push_back(offsets);
  in.owners.push_back(bytes);
  strfn_rows_ += rows;
}

// The chain's derived columns (codegen.hpp DerivedCol) computed over its source table and appended to it.  split: two passes of the matcher per
// row (regex_kernels.hip) — the pieces counted, a prefix sum, every piece described as a view of its source value — and the element column
// assembled from the views like every string view's result.
void ExecutionContext::extend_derived(DevTable& in, const std::vector<DerivedCol>& derived) {
  for (const DerivedCol& dc : derived) {
    if (dc.kind == 3) { extend_derived_strfn(in, dc); continue; }
    if ((dc.kind != 1 && dc.kind != 2) || dc.src < 0 || (size_t)dc.src >= in.cols.size()) throw CometError("internal: unknown derived column");
    const DeviceColumnView sc = in.cols[(size_t)dc.src];
    const bool hv = in.has_valid[(size_t)dc.src];
    const int64_t rows = in.rows;
    if (!sc.data && rows) throw CometError("split over a Utf8 column without offsets is not supported");
    if (hv && sc.offset != 0) throw CometError("split over a sliced Utf8 column with NULLs is not supported yet");      // (the list shares the column's validity bitmap)
    DevBuf prog, counts, tiles;
    auto list_offs = std::make_shared<DevBuf>();
    prog.ensure(dc.prog.size() * 4 + 16);
    // (a program with \\b carries the \\w table: several KB — not a write_small)
    HIP_CHECK(hipMemcpyAsync(prog.p, dc.prog.data(), dc.prog.size() * 4, hipMemcpyHostToDevice, stream_));
    DevBuf prog_group;      // regexp_extract_all: the wanted group's program (group 0: the pattern's own)
    const uint32_t* prog2 = nullptr;
    if (dc.kind == 2) {
      if (dc.prog2.empty()) prog2 = (const uint32_t*)prog.p;
      else {
        prog_group.ensure(dc.prog2.size() * 4 + 16);
        HIP_CHECK(hipMemcpyAsync(prog_group.p, dc.prog2.data(), dc.prog2.size() * 4, hipMemcpyHostToDevice, stream_));
        prog2 = (const uint32_t*)prog_group.p;
      }
    }
    HIP_CHECK(hipStreamSynchronize(stream_));
    counts.ensure((size_t)std::max<int64_t>(rows, 1) * 4 + 16);
    tiles.ensure((size_t)((rows + 1023) / 1024 + 2) * 8);
    list_offs->ensure((size_t)(rows + 2) * 4);
    HIP_CHECK(hipMemsetAsync(list_offs->p, 0, 8, stream_));
    const int32_t* offs = (const int32_t*)sc.data + sc.offset;
    const uint8_t* vbits = hv ? sc.valid : nullptr;
    if (comet_launch_split_count(offs, (const uint8_t*)sc.aux, vbits, sc.offset, rows, (const uint32_t*)prog.p, prog2, dc.limit, (uint32_t*)counts.p, stream_) != 0) throw CometError("split: launch failed");
    if (rows) pq_launch_u32_scan((const uint32_t*)counts.p, rows, (uint64_t*)tiles.p, (int32_t*)list_offs->p, stream_);
    int32_t total = 0;
    if (rows) read_small(&total, (char*)list_offs->p + (size_t)rows * 4, 4);
    if (total < 0) throw CometError("split: more than 2^31 pieces in one table");
    DevBuf views, lengths, etiles;
    auto eoffs = std::make_shared<DevBuf>(), ebytes = std::make_shared<DevBuf>();
    views.ensure((size_t)std::max<int32_t>(total, 1) * 16 + 16);
    lengths.ensure((size_t)std::max<int32_t>(total, 1) * 4 + 16);
    etiles.ensure((size_t)((total + 1023) / 1024 + 2) * 8);
    eoffs->ensure((size_t)(total + 2) * 4);
    HIP_CHECK(hipMemsetAsync(eoffs->p, 0, 8, stream_));
    if (comet_launch_split_write(offs, (const uint8_t*)sc.aux, vbits, sc.offset, rows, (const uint32_t*)prog.p, prog2, dc.limit, (const int32_t*)list_offs->p, views.p, stream_) != 0)
      throw CometError("split: launch failed");
    if (comet_launch_strview_lengths(views.p, nullptr, total, nullptr, 0, (uint32_t*)lengths.p, stream_) != 0) throw CometError("split: launch failed");
    if (total) pq_launch_u32_scan((const uint32_t*)lengths.p, total, (uint64_t*)etiles.p, (int32_t*)eoffs->p, stream_);
    int32_t nbytes = 0;
    if (total) read_small(&nbytes, (char*)eoffs->p + (size_t)total * 4, 4);
    if (nbytes < 0) throw CometError("Utf8 column exceeds 2 GiB of string data (LargeUtf8 is not supported)");
    ebytes->ensure((size_t)nbytes + 16);
    if (comet_launch_strview_copy(views.p, nullptr, offs, (const uint8_t*)sc.aux, total, nullptr, 0, 0, (const int32_t*)eoffs->p, (uint8_t*)ebytes->p, stream_) != 0)
      throw CometError("split: launch failed");
    HIP_CHECK(hipStreamSynchronize(stream_));      // the program, counts, views and lengths go back to the pool
    DeviceColumnView elem;
    elem.data = eoffs->p;
    elem.aux = ebytes->p;
    DeviceColumnView lv;
    lv.data = list_offs->p;
    lv.valid = sc.valid;
    lv.offset = 0;
    lv.kids.push_back(elem);
    lv.kid_has_valid.push_back(false);
    lv.kid_rows = total;
    in.types.push_back(dc.type);
    in.cols.push_back(lv);
    in.has_valid.push_back(hv);
    DType et = dc.type.kids[0];      // the elements as a column of their own (codegen.cpp list_ref: ListExtract / array_contains over the derived list)
    et.virt_parent = (int)in.types.size() - 1;
    et.virt_kid = 0;
    in.types.push_back(et);
    in.cols.push_back(elem);
    in.has_valid.push_back(false);
    in.owners.push_back(list_offs);
    in.owners.push_back(eoffs);
    in.owners.push_back(ebytes);
    split_rows_ += rows;
  }
}

// ---- exact Float64 sums: window bookkeeping (device side: comet_device.hpp "Exact Float64 sums") ----

// Filter/Project chain `top` over the resident table `in` → resident table
DevTable ExecutionContext::run_chain_to_device(const Operator& top, const DevTable& in_plain) {
  DevTable in = extend_struct_fields(in_plain);
  if (in.cols.size() > COMET_MAX_IN) throw CometError("too many columns (struct fields included) for one GPU pipeline");
  auto pv = planned_variant(top, plan_hash_ ^ (0x9E3779B97F4A7C15ull * (uint64_t)(node_id_[&top] + 1)), in.has_valid, true, &in.types);
  if (pv->desc.sink != SinkKind::Output) throw CometError("internal: run_chain_to_device on an aggregate chain");
  extend_derived(in, pv->desc.derived);      // (split: list columns computed over the source, passed through by row index)
  if (in.cols.size() > COMET_MAX_IN) throw CometError("too many columns (derived columns included) for one GPU pipeline");
  Variant v;
  v.desc = pv->desc;
  note_sites(v.desc);
  v.mod = jit_load(pv->code);
  const PipelineDesc& d = v.desc;
  const int64_t n = in.rows;
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
  const size_t ncol = d.out_cols.size();
  std::vector<std::shared_ptr<DevBuf>> vals(ncol), vbytes(ncol);
  auto bind = [&](int64_t cap) {
    for (size_t j = 0; j < ncol; j++) {
      vals[j] = std::make_shared<DevBuf>();
      vals[j]->ensure((size_t)cap * out_width(d.out_cols[j]) + 16);
      prm.out[kOutFirstCol + 2 * j] = vals[j]->p;
      vbytes[j] = std::make_shared<DevBuf>();
      if (d.out_cols[j].nullable) {
        vbytes[j]->ensure((size_t)cap + 16);
        prm.out[kOutFirstCol + 2 * j + 1] = vbytes[j]->p;
      }
    }
  };
  int64_t out_rows = n;
  timed_begin();
  if (n == 0) {
    bind(1);
    out_rows = 0;
  } else if (d.has_filter) {
    bind(n);
    out_rows = launch_fused_filter(v, prm, n);
  } else {
    bind(n);
    launch(v, "k_emit", (int)std::min<int64_t>((n + 255) / 256, 256 * 8), prm);
  }
  timed_end();
  DevTable out = outputs_to_table(v, vals, vbytes, out_rows, [&](int c) { return std::make_pair(&in, c); });
  out.owners.push_back(v.mod);
  return out;
}

DevTable ExecutionContext::materialize(const Operator& op) {
  static const bool trace = getenv("COMET_TRACE_STAGES") != nullptr;
  Timer tm;
  struct Report {
    const Operator& op; Timer& tm; bool on;
    ~Report() { if (on) fprintf(stderr, "[comet] materialize %s: %.3f ms (incl. children)\n", op_name(op.proto_tag), tm.ns() / 1e6); }
  } report{op, tm, trace};
  if (op.kind == OpKind::Scan) {
    const size_t input = scan_input_.at(&op);
    DevTable t;
    t.types = op.scan_fields;
    if (inputs_[input].kind == 0) {
      // host stream: the whole input becomes one resident chunk (joins need their inputs complete)
      int64_t rows = 0;
      if (pull_host_table(input, op.scan_fields, INT64_MAX, t.cols, t.has_valid, rows)) t.rows = rows;
      if (t.cols.empty()) {
        t.cols.assign(op.scan_fields.size(), DeviceColumnView());
        t.has_valid.assign(op.scan_fields.size(), false);
      }
      HIP_CHECK(hipStreamSynchronize(stream_));
    } else {
      std::shared_ptr<void> keep;
      int64_t rows = 0;
      if (pull_device_table(input, op.scan_fields, t.cols, t.has_valid, rows, keep)) {
        t.rows = rows;
        t.owners.push_back(keep);
        std::vector<DeviceColumnView> c2;
        std::vector<bool> v2;
        std::shared_ptr<void> k2;
        int64_t r2 = 0;
        if (pull_device_table(input, op.scan_fields, c2, v2, r2, k2) && r2 > 0)
          throw CometError("a device input stream feeding a join must deliver a single batch");
      } else {
        t.cols.assign(op.scan_fields.size(), DeviceColumnView());
        t.has_valid.assign(op.scan_fields.size(), false);
      }
    }
    input_rows += t.rows;
    return t;
  }
  if (op.kind == OpKind::NativeScan) {
    DevTable t = scan_parquet(op);
    input_rows += t.rows;
    return t;
  }
  if (op.kind == OpKind::HashJoin) {
    auto fit = fused_probe_.find(&op);
    auto fbit = fused_build_.find(&op);
    DevTable j;
    if (fit != fused_probe_.end() || fbit != fused_build_.end()) {
      // the probe child's Filters / Projections run inside the probe kernel over the chain's source table; a fused build child's inside the build passes
      const bool build_left = op.build_side == BuildSide::Left;
      const bool fb_on = fbit != fused_build_.end(), fp_on = fit != fused_probe_.end();
      DevTable b = materialize(fb_on ? *fbit->second.source : *op.children[build_left ? 0 : 1]);
      DevTable s = materialize(fp_on ? *fit->second.source : *op.children[build_left ? 1 : 0]);
      JoinFusion fu, fub;
      if (fp_on) { fu = fit->second.fu; fu.src_valid = s.has_valid; }
      if (fb_on) { fub = fbit->second.fu; fub.src_valid = b.has_valid; join_fused_builds_++; }
      const char* sfx = fp_on && fb_on ? ":FB" : fp_on ? ":F" : ":B";
      j = build_left ? hash_join_impl(op, op, b, s, sfx, fp_on ? &fu : nullptr, fb_on ? &fub : nullptr) : hash_join_impl(op, op, s, b, sfx, fp_on ? &fu : nullptr, fb_on ? &fub : nullptr);
    } else {
      DevTable l = materialize(*op.children[0]);
      DevTable r = materialize(*op.children[1]);
      j = hash_join(op, l, r);
    }
    if (!op.smj || !smj_needs_sort_.count(&op)) return j;
    // SortMergeJoin: its output is ordered by the join keys (SortMergeJoinExec streams the sorted inputs, planner.rs:2126-2191)
    return sort_table(*smj_sorts_.at(&op), j);
  }
  if (op.kind == OpKind::Sort) {
    DevTable in = materialize(*op.children[0]);
    return sort_table(op, in);
  }
  if (op.kind == OpKind::Limit) {
    // LocalLimitExec / GlobalLimitExec (planner.rs:1436-1470): rows [offset, limit) of the child, in its order
    DevTable in = materialize(*op.children[0]);
    const int64_t off = std::min<int64_t>(std::max(0, op.offset), in.rows);
    const int64_t end = op.limit < 0 ? in.rows : std::min<int64_t>(in.rows, op.limit);
    return take_rows(in, nullptr, off, std::max<int64_t>(0, end - off), nullptr);
  }
  if (op.kind == OpKind::ShuffleWriter) return write_shuffle(op);
  if (op.kind == OpKind::Expand) {
    DevTable in = materialize(*op.children[0]);
    return expand(op, in);
  }
  if (op.kind == OpKind::Explode) {
    DevTable in = materialize(*op.children[0]);
    return explode(op, in);
  }
  if (op.kind == OpKind::Window) {
    DevTable in = materialize(*op.children[0]);
    return window(op, in);
  }
  if (op.kind == OpKind::HashAgg) return nested_aggregate(op);   // an aggregate below other operators
  // Filter / Projection chain: fused over its source
  const Operator* src = &op;
  do src = src->children[0].get(); while (!is_source(*src, &op));
  DevTable in = materialize(*src);
  DevTable out = run_chain_to_device(op, in);
  HIP_CHECK(hipStreamSynchronize(stream_));
  return out;
}



// An aggregate below other operators (Sort / Project / Filter / join over a HashAggregate): it runs as its own execution
// context over the input streams of its sub-tree, and its grouped result is handed over resident in HBM.
DevTable ExecutionContext::nested_aggregate(const Operator& agg) {
  // the Scan leaves of the sub-tree, in depth-first order, are a contiguous range of this context's inputs
  std::vector<size_t> idx;
  std::function<void(const Operator&)> walk = [&](const Operator& op) {
    if (op.kind == OpKind::Scan) idx.push_back(scan_input_.at(&op));
    for (auto& c : op.children) walk(*c);
  };
  walk(agg);
  std::vector<InputSource> sub_inputs;
  for (size_t i : idx) {
    sub_inputs.push_back(inputs_[i]);
    inputs_[i].host = nullptr;      // ownership moves to the sub-context (it releases the streams)
    inputs_[i].dev = nullptr;
    inputs_[i].exhausted = true;
  }
  // the sub-plan shares the Operator nodes: wrap the node in a non-owning shared_ptr
  OperatorP sub_plan(const_cast<Operator*>(&agg), [](Operator*) {});
  static const bool trace = getenv("COMET_TRACE_STAGES") != nullptr;
  Timer tm;
  auto mark = [&](const char* what) { if (trace) fprintf(stderr, "[comet] nested aggregate: %s at %.3f ms\n", what, tm.ns() / 1e6); };
  auto subp = std::make_unique<ExecutionContext>(sub_plan, plan_hash_ ^ (0x9E3779B97F4A7C15ull * (uint64_t)(node_id_[&agg] + 1)), config_, sub_inputs, 0, device_id_);
  ExecutionContext& sub = *subp;
  struct Gone { std::unique_ptr<ExecutionContext>& p; decltype(mark)& m; ~Gone() { p.reset(); m("context released"); } } gone{subp, mark};
  mark("context built");
  if (sub.sink_ != SinkKind::AggGrouped && sub.sink_ != SinkKind::AggNoGroup) throw CometError("internal: nested aggregate without an aggregate sink");
  sub.device_result_ = true;
  sub.start();
  mark("started");
  sub.run_to_completion();
  mark("input consumed");
  DevTable t;
  if (sub.sink_ == SinkKind::AggNoGroup) {
    // an ungrouped aggregate yields exactly one row (TPC-H Q14 / Q17 / Q19 compute on it): finish it the usual way, then put that row
    // back into HBM for the operators above
    sub.finish_aggregate();
    if (sub.ready_.empty()) throw CometError("internal: ungrouped aggregate produced no row");
    t = host_batch_to_table(sub.ready_.front());
    sub.ready_.clear();
  } else {
    t = sub.grouped_to_device();
  }
  mark("groups emitted");
  input_rows += sub.input_rows;
  part_merges_ += sub.part_merges_;
  sub.collect_timings();
  last_kernel_ms += sub.last_kernel_ms;
  last_kernel_launches += sub.last_kernel_launches;
  if (!sub.kt_pending_.empty() || !sub.kernel_times_.empty()) {
    sub.kernel_times_json();
    for (auto& kv : sub.kernel_times_) { kernel_times_[kv.first].first += kv.second.first; kernel_times_[kv.first].second += kv.second.second; }
  }
  return t;
}

// a (small) host batch → resident table
DevTable ExecutionContext::host_batch_to_table(const HostBatch& b) {
  DevTable t;
  t.rows = b.rows;
  auto up = [&](const std::vector<uint8_t>& v) -> const void* {
    auto d = std::make_shared<DevBuf>();
    d->ensure(v.size() + 16);
    if (!v.empty()) HIP_CHECK(hipMemcpy(d->p, v.data(), v.size(), hipMemcpyHostToDevice));
    t.owners.push_back(d);
    return d->p;
  };
  for (const HostColumn& c : b.cols) {
    DeviceColumnView v;
    v.data = up(c.values);
    const bool is_str = c.type.id == TypeId::String || c.type.id == TypeId::Bytes;
    if (is_str) v.aux = up(c.data);
    const bool hv = c.null_count > 0 && !c.validity.empty();
    if (hv) v.valid = (const uint8_t*)up(c.validity);
    t.types.push_back(c.type);
    t.cols.push_back(v);
    t.has_valid.push_back(hv);
  }
  return t;
}

// resident table → host batches of ≤ batch_size rows (root of a plan that ends in a join)
// ---- nested columns on their way out: the whole column comes back (one synchronous copy per buffer — nested results are not the hot path),
// batches are slices of it ----
HostColumn download_column(const DeviceColumnView& v, const DType& t, bool has_valid, int64_t rows, hipStream_t st) {
  HostColumn c;
  c.type = t;
  c.length = rows;
  auto fetch = [&](std::vector<uint8_t>& dst, const void* src, size_t bytes) {
    dst.resize(bytes);
    if (bytes) HIP_CHECK(hipMemcpyAsync(dst.data(), src, bytes, hipMemcpyDeviceToHost, st));
  };
  if (has_valid && v.valid && rows) fetch(c.validity, v.valid, (size_t)((rows + 7) / 8));
  if (t.id == TypeId::Struct) {
    for (size_t i = 0; i < t.kids.size(); i++)
      c.children.push_back(download_column(v.kids.at(i), t.kids[i], i < v.kid_has_valid.size() && v.kid_has_valid[i], rows, st));
  } else if (t.is_listlike()) {
    fetch(c.values, v.data, (size_t)(rows + 1) * 4);
    HIP_CHECK(hipStreamSynchronize(st));
    const int64_t nel = rows ? ((const int32_t*)c.values.data())[rows] : 0;
    if (nel < 0 || nel > v.kid_rows) throw CometError("internal: list offsets run past the element column");
    c.children.push_back(download_column(v.kids.at(0), t.kids.at(0), !v.kid_has_valid.empty() && v.kid_has_valid[0], nel, st));
  } else if (t.id == TypeId::String || t.id == TypeId::Bytes) {
    fetch(c.values, (const int32_t*)v.data + v.offset, (size_t)(rows + 1) * 4);
    HIP_CHECK(hipStreamSynchronize(st));
    const int32_t* o = (const int32_t*)c.values.data();
    const int64_t lo = rows ? o[0] : 0, hi = rows ? o[rows] : 0;
    if (hi > lo) fetch(c.data, (const uint8_t*)v.aux + lo, (size_t)(hi - lo));
    if (lo) { int32_t* w = (int32_t*)c.values.data(); for (int64_t i = 0; i <= rows; i++) w[i] -= (int32_t)lo; }
  } else if (t.id == TypeId::Bool) {
    fetch(c.values, v.data, (size_t)((rows + 7) / 8));
  } else {
    fetch(c.values, v.data, (size_t)rows * (size_t)fixed_width(t));
  }
  HIP_CHECK(hipStreamSynchronize(st));
  if (!c.validity.empty()) {
    int64_t nulls = 0;
    for (int64_t i = 0; i < rows; i++) nulls += !((c.validity[(size_t)(i >> 3)] >> (i & 7)) & 1);
    c.null_count = nulls;
    if (!nulls) c.validity.clear();
  }
  return c;
}
namespace {
std::vector<uint8_t> slice_bits(const std::vector<uint8_t>& b, int64_t off, int64_t len) {
  std::vector<uint8_t> o((size_t)((len + 7) / 8), 0);
  for (int64_t i = 0; i < len; i++)
    if ((b[(size_t)((off + i) >> 3)] >> ((off + i) & 7)) & 1) o[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
  return o;
}
HostColumn slice_column(const HostColumn& c, int64_t off, int64_t len) {
  if (off == 0 && len == c.length) return c;
  HostColumn o;
  o.type = c.type;
  o.length = len;
  if (!c.validity.empty()) {
    o.validity = slice_bits(c.validity, off, len);
    for (int64_t i = 0; i < len; i++) o.null_count += !((o.validity[(size_t)(i >> 3)] >> (i & 7)) & 1);
    if (!o.null_count) o.validity.clear();
  }
  const TypeId id = c.type.id;
  if (id == TypeId::Struct) {
    for (auto& k : c.children) o.children.push_back(slice_column(k, off, len));
  } else if (id == TypeId::List || id == TypeId::Map || id == TypeId::String || id == TypeId::Bytes) {
    const int32_t* src = (const int32_t*)c.values.data();
    o.values.resize((size_t)(len + 1) * 4);
    int32_t* w = (int32_t*)o.values.data();
    for (int64_t i = 0; i <= len; i++) w[i] = src[off + i] - src[off];
    if (id == TypeId::List || id == TypeId::Map) o.children.push_back(slice_column(c.children.at(0), src[off], src[off + len] - src[off]));
    else o.data.assign(c.data.begin() + src[off], c.data.begin() + src[off + len]);
  } else if (id == TypeId::Bool) {
    o.values = slice_bits(c.values, off, len);
  } else {
    const size_t w = (size_t)fixed_width(c.type);
    o.values.assign(c.values.begin() + (size_t)off * w, c.values.begin() + (size_t)(off + len) * w);
  }
  return o;
}
}  // namespace

void ExecutionContext::table_to_host_batches(const DevTable& t) {
  if (t.rows == 0) return;
  const size_t ncol = t.cols.size();
  std::vector<HostColumn> nested(ncol);      // nested columns, whole
  for (size_t j = 0; j < ncol; j++)
    if (t.types[j].is_nested()) nested[j] = download_column(t.cols[j], t.types[j], t.has_valid[j], t.rows, stream_);
  std::vector<std::vector<uint8_t>> hv(ncol), hb(ncol);
  std::vector<std::vector<uint8_t>> hd(ncol);
  // what has to come back: values (offsets for Utf8) and validity of every column; then the Utf8 bytes, whose size the offsets tell
  struct Piece { const void* src; std::vector<uint8_t>* dst; size_t bytes; };
  // A small result (the last batch of most queries: a few groups, a top-k) comes back through ONE kernel that writes every buffer into
  // pinned host memory and one synchronisation; a large one with a copy per buffer.
  auto fetch = [&](std::vector<Piece>& pieces) {
    size_t total = 0, longest = 0;
    for (auto& pc : pieces) { pc.dst->resize(pc.bytes); total += (pc.bytes + 15) & ~(size_t)15; longest = std::max(longest, pc.bytes); }
    if (pieces.empty()) return;
    static const bool batched = getenv("COMET_EXPORT_BATCHED") == nullptr || atoi(getenv("COMET_EXPORT_BATCHED")) != 0;
    if (batched && total > 0 && total <= ((size_t)1 << 20) && pieces.size() <= 256) {
      struct Desc { const uint8_t* src; uint8_t* dst; uint64_t len; };
      const size_t head = (pieces.size() * sizeof(Desc) + 63) & ~(size_t)63;
      export_host_.ensure(head + total + 64);
      Desc* d = (Desc*)export_host_.p;
      size_t at = head;
      for (size_t k = 0; k < pieces.size(); k++) {
        d[k].src = (const uint8_t*)pieces[k].src;
        d[k].dst = (uint8_t*)export_host_.p + at;
        d[k].len = pieces[k].bytes;
        at += (pieces[k].bytes + 15) & ~(size_t)15;
      }
      if (comet_launch_copy_small(export_host_.p, (int)pieces.size(), (uint64_t)longest, stream_) != 0) throw CometError("result export: launch failed");
      HIP_CHECK(hipStreamSynchronize(stream_));
      for (size_t k = 0; k < pieces.size(); k++)
        if (pieces[k].bytes) memcpy(pieces[k].dst->data(), d[k].dst, pieces[k].bytes);
      return;
    }
    for (auto& pc : pieces)
      if (pc.bytes) HIP_CHECK(hipMemcpyAsync(pc.dst->data(), pc.src, pc.bytes, hipMemcpyDeviceToHost, stream_));
    HIP_CHECK(hipStreamSynchronize(stream_));
  };
  std::vector<Piece> pieces;
  for (size_t j = 0; j < ncol; j++) {
    const DType& ty = t.types[j];
    if (ty.is_nested()) continue;
    const bool is_str = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    size_t bytes = is_str ? (size_t)(t.rows + 1) * 4 : ty.id == TypeId::Bool ? (size_t)((t.rows + 7) / 8) : (size_t)t.rows * fixed_width(ty);
    pieces.push_back({t.cols[j].data, &hv[j], bytes});
    if (t.has_valid[j]) pieces.push_back({t.cols[j].valid, &hb[j], (size_t)((t.rows + 7) / 8)});
  }
  fetch(pieces);
  pieces.clear();
  for (size_t j = 0; j < ncol; j++) {
    if (t.types[j].id == TypeId::String || t.types[j].id == TypeId::Bytes) {
      const int32_t* offs = (const int32_t*)hv[j].data();
      if (offs[t.rows] > 0) pieces.push_back({t.cols[j].aux, &hd[j], (size_t)offs[t.rows]});
    }
  }
  fetch(pieces);
  check_device_errors();
  const int64_t bs = batch_size_ > 0 ? batch_size_ : t.rows;
  auto getbit = [](const std::vector<uint8_t>& b, int64_t i) { return (b[(size_t)(i >> 3)] >> (i & 7)) & 1; };
  for (int64_t off = 0; off < t.rows; off += bs) {
    const int64_t len = std::min(bs, t.rows - off);
    HostBatch b;
    b.rows = len;
    for (size_t j = 0; j < ncol; j++) {
      if (t.types[j].is_nested()) { b.cols.push_back(slice_column(nested[j], off, len)); continue; }
      HostColumn c;
      c.type = t.types[j];
      c.length = len;
      if (c.type.id == TypeId::String || c.type.id == TypeId::Bytes) {
        const int32_t* offs = (const int32_t*)hv[j].data();
        c.values.resize((size_t)(len + 1) * 4);
        int32_t* o = (int32_t*)c.values.data();
        for (int64_t i = 0; i <= len; i++) o[i] = offs[off + i] - offs[off];
        c.data.assign(hd[j].begin() + offs[off], hd[j].begin() + offs[off + len]);
      } else if (c.type.id == TypeId::Bool) {
        c.values.assign((size_t)((len + 7) / 8), 0);
        for (int64_t i = 0; i < len; i++)
          if (getbit(hv[j], off + i)) c.values[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
      } else {
        int w = fixed_width(c.type);
        c.values.assign(hv[j].begin() + (size_t)off * w, hv[j].begin() + (size_t)(off + len) * w);
      }
      if (t.has_valid[j]) {
        int64_t nulls = 0;
        std::vector<uint8_t> bm((size_t)((len + 7) / 8), 0);
        for (int64_t i = 0; i < len; i++) {
          if (getbit(hb[j], off + i)) bm[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
          else nulls++;
        }
        c.null_count = nulls;
        if (nulls) c.validity = std::move(bm);
      }
      b.cols.push_back(std::move(c));
    }
    ready_.push_back(std::move(b));
  }
}

void ExecutionContext::run_to_completion() {
  if (has_join_) {
    DevTable src = extend_struct_fields(materialize(*root_source_));
    if (plan_.get() == root_source_) throw CometError("internal: bare join root");
    if (sink_ == SinkKind::AggGrouped) prepare_dict_keys(src);
    single_chunk_hint_ = true;      // (the join's output is the aggregate's whole input)
    process_chunk(src.cols, src.has_valid, src.rows);
    HIP_CHECK(hipStreamSynchronize(stream_));
    return;
  }
  // ungrouped / grouped aggregates are pipeline breakers: drain the input completely
  while (true) {
    bool more = inputs_[0].kind == 0 ? pull_host_chunk() : pull_device_batch();
    if (!more) break;
  }
}

void ExecutionContext::start() {
  if (!started_) {
    // Lazy start like the reference (jni_api.rs:795-872): nothing touches the input before the first executePlan.
    HIP_CHECK(hipSetDevice(device_id_));
    stream_ = pool_get_stream(device_id_);
    err_flags_.ensure(kErrBytes);
    HIP_CHECK(hipMemsetAsync(err_flags_.p, 0, kErrBytes, stream_));
    started_ = true;
  } else {
    HIP_CHECK(hipSetDevice(device_id_));
  }
}

namespace {
struct ExportedDeviceColumn {
  std::vector<std::shared_ptr<void>> owners;   // pooled buffers / producer arrays the pointers live in
  const void* buffers[3];
};
void release_device_array(ArrowArray* a) {
  delete (ExportedDeviceColumn*)a->private_data;
  a->release = nullptr;
}
void release_fmt_schema(ArrowSchema* s) {
  delete (std::string*)s->private_data;
  s->release = nullptr;
}
}  // namespace

// The stage boundary for multi-GPU plans: a Filter/Project/HashJoin plan's output stays resident so that the
// exchange (murmur3 → pmod → partition scatter → RCCL all-to-all) never touches the host.  One batch = the whole result.
void ExecutionContext::set_memory_manager(int64_t (*acquire)(void*, int64_t), void (*release)(void*, int64_t), void* ctx, long long task_id) {
  mem_->acquire = acquire;
  mem_->release = release;
  mem_->ctx = ctx;
  mem_->task_id = task_id;
}
void ExecutionContext::memory_stats(int64_t out[4]) {
  out[0] = mem_->host_used.load();
  out[1] = mem_->host_peak.load();
  out[2] = mem_->dev_used.load();
  out[3] = mem_->dev_peak.load();
}

int64_t ExecutionContext::execute_device(ArrowDeviceArray** out_arrays, ArrowSchema** out_schemas, int n_out) {
  mem_->owner = std::this_thread::get_id();      // Spark may move a task's calls between threads; the up-calls go with the caller
  AccountScope account(mem_);
  mem_->flush();
  Timer t;
  start();
  if (sink_ == SinkKind::AggNoGroup)
    throw CometError("comet_execute_plan_device: an ungrouped aggregate result is one row and is exported through comet_execute_plan");
  if (finished_) return -1;
  DevTable tab;
  if (sink_ == SinkKind::AggGrouped) {
    device_result_ = true;
    run_to_completion();
    tab = grouped_to_device();
  } else {
    tab = materialize(*plan_);
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  check_device_errors();
  collect_timings();
  finished_ = true;
  if ((size_t)n_out != tab.cols.size())
    throw CometError("Output column count mismatch: expected " + std::to_string(n_out) + ", got " + std::to_string(tab.cols.size()));
  for (size_t c = 0; c < tab.types.size(); c++)
    if (tab.types[c].is_nested()) throw CometError("executePlan (device arrays): nested column of type " + tab.types[c].str() + " — nested results are exported through host batches");
  for (int j = 0; j < n_out; j++) {
    const DType& ty = tab.types[(size_t)j];
    const bool is_str = ty.id == TypeId::String || ty.id == TypeId::Bytes;
    if (tab.cols[(size_t)j].offset != 0) throw CometError("comet_execute_plan_device: input column with a non-zero Arrow offset passed through");
    auto* ec = new ExportedDeviceColumn();
    ec->owners = tab.owners;
    ec->buffers[0] = tab.has_valid[(size_t)j] ? tab.cols[(size_t)j].valid : nullptr;
    ec->buffers[1] = tab.cols[(size_t)j].data;
    ec->buffers[2] = is_str ? tab.cols[(size_t)j].aux : nullptr;
    ArrowDeviceArray* d = out_arrays[j];
    memset(d, 0, sizeof *d);
    d->array.length = tab.rows;
    d->array.null_count = tab.has_valid[(size_t)j] ? -1 : 0;
    d->array.n_buffers = is_str ? 3 : 2;
    d->array.buffers = ec->buffers;
    d->array.private_data = ec;
    d->array.release = release_device_array;
    d->device_id = device_id_;
    d->device_type = ARROW_DEVICE_ROCM;
    d->sync_event = nullptr;   // the plan's stream was synchronised above
    ArrowSchema* s = out_schemas[j];
    memset(s, 0, sizeof *s);
    auto* fmt = new std::string(expected_format(ty));
    s->format = fmt->c_str();
    s->name = "";
    s->flags = ARROW_FLAG_NULLABLE;
    s->private_data = fmt;
    s->release = release_fmt_schema;
  }
  output_rows_ += tab.rows;
  elapsed_compute_ns_ += t.ns();
  return tab.rows;
}

void ExecutionContext::resolve_subqueries() {
  if (subqueries_resolved_) return;
  subqueries_resolved_ = true;
  uint64_t sig = 0xcbf29ce484222325ull;
  auto mix = [&](const void* p, size_t n) { for (size_t k = 0; k < n; k++) sig = (sig ^ ((const uint8_t*)p)[k]) * 0x100000001b3ull; };
  for (auto& e : plan_->subqueries) {
    if (e->kind != ExprKind::Subquery) continue;      // (one node, listed once)
    const int64_t id = e->lit_i64;
    bool is_null = true;
    std::string v;
    bool found = false;
    auto it = subquery_values_.find(id);
    if (it != subquery_values_.end()) { is_null = it->second.first; v = it->second.second; found = true; }
    else if (subquery_provider_) found = subquery_provider_(id, e->dtype, is_null, v);
    if (!found) throw CometError("Subquery " + std::to_string(id) + " is not registered with this plan (CometScalarSubquery.setSubquery / comet_plan_set_subquery)");
    auto need = [&](size_t n) { if (!is_null && v.size() < n) throw CometError("Subquery " + std::to_string(id) + ": the value has " + std::to_string(v.size()) + " bytes, its type needs " + std::to_string(n)); };
    e->kind = ExprKind::Literal;
    e->proto_tag = 2;
    e->lit_null = is_null;
    e->lit_i64 = 0;
    int64_t i = 0;
    double d = 0;
    switch (e->dtype.id) {
      case TypeId::Bool: need(1); e->lit_bool = !is_null && v[0] != 0; e->lit_case = 1; break;
      case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: case TypeId::Date: case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz:
        need(8);
        if (!is_null) memcpy(&i, v.data(), 8);
        e->lit_i64 = i;
        e->lit_case = e->dtype.id == TypeId::Int8 ? 2 : e->dtype.id == TypeId::Int16 ? 3 : (e->dtype.id == TypeId::Int32 || e->dtype.id == TypeId::Date) ? 4 : 5;
        break;
      case TypeId::Float: case TypeId::Double:
        need(8);
        if (!is_null) memcpy(&d, v.data(), 8);
        e->lit_f64 = e->dtype.id == TypeId::Float ? (double)(float)d : d;
        e->lit_case = e->dtype.id == TypeId::Float ? 6 : 7;
        break;
      case TypeId::Decimal: e->lit_dec = is_null ? 0 : decode_decimal_be(v); e->lit_case = 10; break;
      case TypeId::String: case TypeId::Bytes: e->lit_bytes = is_null ? std::string() : v; e->lit_case = e->dtype.id == TypeId::String ? 8 : 9; break;
      default: throw CometError("Unsupported scalar subquery data type " + e->dtype.str());
    }
    mix(&id, 8);
    mix(&is_null, 1);
    mix(v.data(), v.size());
  }
  // the plan's kernels carry the values: planned anew, under a hash that names them
  if (!plan_->subqueries.empty()) plan_hash_ ^= sig * 0x9E3779B97F4A7C15ull;
}

int64_t ExecutionContext::execute(ArrowArray** out_arrays, ArrowSchema** out_schemas, int n_out) {
  mem_->owner = std::this_thread::get_id();
  if (!subqueries_resolved_) resolve_subqueries();
  AccountScope account(mem_);
  mem_->flush();
  Timer t;
  start();
  const bool is_agg = sink_ != SinkKind::Output;
  if (!finished_) {
    if (is_agg) {
      run_to_completion();
      if (sink_ == SinkKind::AggGrouped) finish_grouped();
      else finish_aggregate();
      finished_ = true;
    } else {
      if (has_join_ || materialize_root_) {
        // a plan over materialised sources (joins, Parquet scans, sorts, limits, nested aggregates) or one that passes Utf8
        // columns through: the whole result is produced resident in HBM, then copied out in batches
        DevTable t = materialize(*plan_);
        HIP_CHECK(hipStreamSynchronize(stream_));
        table_to_host_batches(t);
        finished_ = true;
      }
      while (ready_.empty() && !finished_) {
        bool more = inputs_[0].kind == 0 ? pull_host_chunk() : pull_device_batch();
        if (!more) finished_ = true;
      }
    }
  }
  elapsed_compute_ns_ += t.ns();
  if (ready_.empty()) return -1;
  HostBatch b = std::move(ready_.front());
  ready_.pop_front();
  export_batch(b, out_arrays, out_schemas, n_out);
  output_rows_ += b.rows;
  return b.rows;
}

// ---------------------------------------------------------------------------------------------
// Arrow C Data export (prepare_output, jni_api.rs:674-742): one moved ArrowArray + ArrowSchema per
// output column, offset 0, buffers owned by the array until the consumer calls release.
// ---------------------------------------------------------------------------------------------
void ExecutionContext::export_batch(HostBatch& b, ArrowArray** out_arrays, ArrowSchema** out_schemas, int n_out) {
  export_host_batch(b, out_arrays, out_schemas, n_out);
}

std::string ExecutionContext::metrics_proto() {
  // tree mirrors the Operator tree (metrics/utils.rs:30-45); per-node attribution of a fused pipeline:
  // the root carries the measured values, fused children report zero time and their row counts unknown (0).
  std::function<MetricNode(const Operator&, bool)> build = [&](const Operator& op, bool root) {
    MetricNode n;
    n.metrics.emplace_back("output_rows", root ? output_rows_ : 0);
    n.metrics.emplace_back("elapsed_compute", root ? (int64_t)elapsed_compute_ns_ : 0);
    if (op.kind == OpKind::ShuffleWriter) {   // ShufflePartitionerMetrics (native/shuffle/src/metrics.rs:24-71)
      n.metrics.emplace_back("data_size", shuffle_data_size_);
      n.metrics.emplace_back("repart_time", (int64_t)shuffle_repart_ns_);
      n.metrics.emplace_back("write_time", (int64_t)shuffle_write_ns_);
      n.metrics.emplace_back("spill_count", 0);
      n.metrics.emplace_back("spilled_bytes", 0);
      n.metrics.emplace_back("staging_peak_bytes", shuffle_staged_peak_);   // largest slab of the partition-major table held in pinned host memory
    }
    if (op.kind == OpKind::NativeScan) {
      n.metrics.emplace_back("bytes_scanned", bytes_scanned_);
      n.metrics.emplace_back("row_groups_pruned_statistics", row_groups_pruned_ - row_groups_pruned_bloom_);
      n.metrics.emplace_back("row_groups_pruned_bloom_filter", row_groups_pruned_bloom_);      // (DataFusion's ParquetFileMetrics names)
      n.metrics.emplace_back("pages_decompressed_on_device", pages_inflated_on_device_);
      n.metrics.emplace_back("page_index_rows_pruned", rows_pruned_page_index_);
    }
    if (root && (op.kind == OpKind::HashAgg || part_merges_ > 0)) n.metrics.emplace_back("agg_partitioned_merges", part_merges_);      // merging aggregates run as partition → LDS merge → emit
    if (op.kind == OpKind::HashJoin && root) {      // (the plan's joins together: the counters are the context's)
      n.metrics.emplace_back("join_build_rows", join_build_rows_);
      n.metrics.emplace_back("join_probe_rows", join_probe_rows_);
      n.metrics.emplace_back("join_direct_maps", join_direct_maps_);      // joins probed through the direct map of a unique integer key
      n.metrics.emplace_back("join_bucket_tables", join_bucket_tables_);  // joins probed through the partitioned, LDS-built bucket table
      n.metrics.emplace_back("join_fused_builds", join_fused_builds_);    // joins whose build passes read their build chain's Scan table
      n.metrics.emplace_back("join_mono_tables", join_mono_tables_);      // … of them, with the order-preserving hash
      n.metrics.emplace_back("join_bitmap_only", join_bitmap_only_);      // semi / anti joins answered by the build side's key bitmap alone
    }
    for (auto& c : op.children) n.children.push_back(build(*c, false));
    return n;
  };
  return encode_metric_node(build(*plan_, true));
}

}  // namespace comet
