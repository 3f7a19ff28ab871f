// Planner + HIP code generator.
//
// A chain  Scan → (Filter | Projection)* → [HashAggregate]  collapses into ONE fused pipeline:
// projections are substituted into later expressions, filter predicates are split into conjuncts,
// and the per-row code is emitted in STAGES (one per conjunct) so a column is only loaded for rows
// (lanes) that survived the earlier conjuncts.  The kernel structure is hand-written
// (device/comet_device.hpp); this file only emits the functor P those templates call.
//
// Semantics followed (reference file:line):
//   operator arms        native/core/src/execution/planner.rs:1230-1384
//   binary arithmetic    planner.rs:976-1132 (wide-decimal path selection :1000-1008)
//   CheckOverflow        planner.rs:600-649, spark-expr/src/math_funcs/internal/checkoverflow.rs:103-160
//   aggregates           planner.rs:2558-2700, spark-expr/src/agg_funcs/*.rs (state schemas)
#include "codegen.hpp"
#include "regex.hpp"
#include "tz.hpp"
#include "kparams.h"
static_assert(comet::kErrBytes == COMET_ERR_BYTES && comet::kErrAuxWords == COMET_ERR_AUX_WORDS && comet::kErrDetailWord == COMET_ERR_DETAIL_WORD &&
              comet::kErrDetailStrBytes == COMET_ERR_DETAIL_STR_BYTES && 16 + 8 * COMET_ERR_AUX_WORDS <= 8 * COMET_ERR_DETAIL_WORD &&
              8 * COMET_ERR_DETAIL_WORD + 32 + COMET_ERR_DETAIL_STR_BYTES <= COMET_ERR_BYTES - 64, "the error block's layout (kparams.h) and codegen.hpp disagree");

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <sstream>

namespace comet {
namespace {

const u128 kUnbounded = ~(u128)0;

u128 pow10_u128(int e) {
  u128 r = 1;
  for (int i = 0; i < e; i++) r *= 10;
  return r;
}
u128 sat_mul(u128 a, u128 b) {
  if (a == 0 || b == 0) return 0;
  if (a == kUnbounded || b == kUnbounded) return kUnbounded;
  u128 r;
  if (__builtin_mul_overflow(a, b, &r)) return kUnbounded;
  return r;
}
u128 sat_add(u128 a, u128 b) {
  if (a == kUnbounded || b == kUnbounded) return kUnbounded;
  u128 r;
  if (__builtin_add_overflow(a, b, &r)) return kUnbounded;
  return r;
}

thread_local bool g_uses_ryu = false;      // the source being generated calls into comet_ryu.hpp (Float → Decimal)
thread_local bool g_uses_strtod = false;   // … into comet_strtod.hpp (String → Float / Double)
thread_local bool g_uses_strts = false;    // … into comet_strts.hpp (String → Timestamp)
thread_local bool g_uses_regex_vm = false; // … into comet_regex_vm.hpp (regexp_extract)
// the pipeline being generated: where raise sites with a QueryContext are noted, and how many sites of each content it has seen (err_sites.cpp)
thread_local std::vector<std::pair<uint32_t, std::shared_ptr<QueryContext>>>* g_site_sink = nullptr;
thread_local std::map<std::string, int>* g_site_ordinals = nullptr;
struct SiteScope {
  std::map<std::string, int> ordinals;
  // (the optional-header flags start clean: a generation that threw after setting one must not make the next, unrelated kernel include the tables)
  explicit SiteScope(PipelineDesc& d) { g_site_sink = &d.site_contexts; g_site_ordinals = &ordinals; g_uses_ryu = g_uses_strtod = g_uses_strts = g_uses_regex_vm = false; }
  ~SiteScope() { g_site_sink = nullptr; g_site_ordinals = nullptr; }
};
std::string with_optional_headers(std::string src) {
  if (g_uses_ryu) {
    const std::string inc = "using namespace comet;\n";
    const size_t at = src.find(inc);
    if (at != std::string::npos) src.insert(at + inc.size(), "namespace comet {\n#include \"comet_ryu.hpp\"\n}\n");
  }
  if (g_uses_strtod) {
    const std::string inc = "using namespace comet;\n";
    const size_t at = src.find(inc);
    if (at != std::string::npos) src.insert(at + inc.size(), "namespace comet {\n#include \"comet_strtod.hpp\"\n}\n");
  }
  if (g_uses_strts) {
    const std::string inc = "using namespace comet;\n";
    const size_t at = src.find(inc);
    if (at != std::string::npos) src.insert(at + inc.size(), "namespace comet {\n#include \"comet_strts.hpp\"\n}\n");
  }
  if (g_uses_regex_vm) {
    const std::string inc = "using namespace comet;\n";
    const size_t at = src.find(inc);
    if (at != std::string::npos) src.insert(at + inc.size(), "#include \"comet_regex_vm.hpp\"\n");
  }
  g_uses_ryu = g_uses_strtod = g_uses_strts = g_uses_regex_vm = false;
  return src;
}

enum class Rep { B, I32, I64, I128, F32, F64, STR };  // STR: Utf8 of ≤15 bytes packed in comet::str16

const char* rep_ctype(Rep r) {
  switch (r) {
    case Rep::B: return "bool";
    case Rep::I32: return "i32";
    case Rep::I64: return "i64";
    case Rep::I128: return "i128";
    case Rep::F32: return "float";
    case Rep::F64: return "double";
    case Rep::STR: return "comet::str16";
  }
  return "?";
}

std::string hex64(uint64_t v) {
  char b[32];
  snprintf(b, sizeof b, "0x%016llxull", (unsigned long long)v);
  return b;
}
std::string lit_i128(i128 v) {
  return "comet::mk128(" + hex64((uint64_t)((u128)v >> 64)) + ", " + hex64((uint64_t)(u128)v) + ")";
}
std::string lit_u128(u128 v) { return "(u128)" + lit_i128((i128)v); }
std::string lit_i64(int64_t v) { return "(i64)" + hex64((uint64_t)v); }
std::string lit_f64(double v) {   // exact: the bit pattern
  uint64_t b;
  memcpy(&b, &v, 8);
  return "__longlong_as_double((i64)" + hex64(b) + ")";
}

struct Val {
  std::string v;    // C expression (per-row vars carry the [r] suffix already)
  std::string ok;   // validity expression, empty = never null
  DType t;
  Rep rep = Rep::I64;
  u128 maxabs = kUnbounded;  // static bound on |value| for ints/decimals
  bool is_null_lit = false;
  bool wide_decimal = false;  // produced by the wide-decimal path (CheckOverflow elision rule)
  bool is_cast_dec = false;   // Cast(decimal→decimal) not yet checked (fusion with CheckOverflow)
  std::string cast_child_v, cast_child_ok;  // for the fused DecimalRescaleCheckOverflow
  DType cast_child_t;
  Rep cast_child_rep = Rep::I64;
  u128 cast_child_max = kUnbounded;
};

Rep rep_for_type(const DType& t) {
  switch (t.id) {
    case TypeId::Bool: return Rep::B;
    case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: case TypeId::Date: return Rep::I32;
    case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: return Rep::I64;
    case TypeId::Float: return Rep::F32;
    case TypeId::Double: return Rep::F64;
    case TypeId::Decimal: return t.precision <= 18 ? Rep::I64 : Rep::I128;
    case TypeId::String: return Rep::STR;
    default: throw CometError("Unsupported data type in native GPU pipeline: " + t.str());
  }
}
u128 type_maxabs(const DType& t) {
  switch (t.id) {
    case TypeId::Int8: return 128;
    case TypeId::Int16: return 32768;
    case TypeId::Int32: case TypeId::Date: return (u128)1 << 31;
    case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: return (u128)1 << 63;
    case TypeId::Decimal: return pow10_u128(t.precision) - 1;
    default: return kUnbounded;
  }
}
Rep rep_for_bound(u128 maxabs) { return maxabs < ((u128)1 << 63) ? Rep::I64 : Rep::I128; }

int type_width(const DType& t) {
  switch (t.id) {
    case TypeId::Bool: case TypeId::Int8: return 1;
    case TypeId::Int16: return 2;
    case TypeId::Int32: case TypeId::Date: case TypeId::Float: return 4;
    case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: case TypeId::Double: return 8;
    case TypeId::Decimal: return 16;
    default: throw CometError("Unsupported output type: " + t.str());
  }
}
const char* store_ctype(const DType& t) {
  switch (t.id) {
    case TypeId::Bool: return "u8";
    case TypeId::Int8: return "i8";
    case TypeId::Int16: return "i16";
    case TypeId::Int32: case TypeId::Date: return "i32";
    case TypeId::Float: return "float";
    case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: return "i64";
    case TypeId::Double: return "double";
    case TypeId::Decimal: return "i128";
    default: throw CometError("Unsupported output type: " + t.str());
  }
}

// ---------------------------------------------------------------------------------------------
// Expression substitution (Projection folding) and conjunct splitting.
// ---------------------------------------------------------------------------------------------
// GetStructField(struct column, k) (expr.proto:528-531; the reference evaluates it on the struct array, planner.rs:776-779): the chain's source
// table carries every field of its struct columns as a column of its own (DType::virt_parent / virt_kid), so the expression folds into a
// plain column reference.  `g_source_cols`: the Bound expressions of the source's columns while a chain is being folded.
thread_local const std::vector<ExprP>* g_source_cols = nullptr;
// split(<Utf8 source column>, <pattern>, <limit>) (string_funcs/split.rs; strings.scala:598-631 sends it under
// spark.comet.expression.StringSplit.allowIncompatible): a list<string> column DERIVED from the source table — the executor computes it before the
// fused kernel runs (exec.cpp extend_derived), the chain addresses it as column (source columns + k).  `g_derived`: where a fold notes them.
thread_local std::vector<DerivedCol>* g_derived = nullptr;
// the chain's column index of derived column k: behind the source's columns and the columns of the derived columns before it
static int derived_index(int nsrc, size_t k) {
  int at = nsrc;
  for (size_t j = 0; j < k; j++) at += (*g_derived)[j].columns();
  return at;
}
// A string function of a source Utf8 column whose result is new bytes — reverse, repeat, replace, substring_index, md5 / sha1 / sha2 (device/strfn.hpp) — becomes
// a DERIVED Utf8 column too (kind 3): computed over the source before the fused kernel runs, then a column like any other to the chain — an output, a
// comparison's operand, a LIKE's subject, whatever its length.  The subject may be wrapped in Cast(… AS BINARY) (Spark's Md5 / Sha1 / Sha2 take binary).
static const ExprP& unwrap_binary_cast(const ExprP& x) {
  if (x->kind == ExprKind::Cast && x->children.size() == 1 && (x->dtype.id == TypeId::Bytes || x->dtype.id == TypeId::String)) return x->children[0];
  return x;
}
bool is_strfn(const std::string& f) {
  return f == "reverse" || f == "repeat" || f == "replace" || f == "substring_index" || f == "substr_index" || f == "md5" || f == "sha1" || f == "sha2";
}
ExprP lower_strfn(const ExprP& e) {
  const std::string& f = e->func;
  if (!g_derived || !g_source_cols) throw CometError(f + " is supported in Projection / Filter chains only");
  if (e->children.empty()) throw CometError(f + " expects arguments");
  const ExprP& subject = unwrap_binary_cast(e->children[0]);
  const int nsrc = (int)g_source_cols->size();
  if (subject->kind != ExprKind::Bound || subject->bound_index < 0 || subject->bound_index >= nsrc || !subject->has_dtype ||
      (subject->dtype.id != TypeId::String && subject->dtype.id != TypeId::Bytes))
    throw CometError(f + " is supported over a Utf8 COLUMN of the source (not over a computed string) by the MI355X native engine");
  auto str_lit = [&](size_t i, const char* what) -> std::string {
    if (i >= e->children.size() || e->children[i]->kind != ExprKind::Literal || e->children[i]->lit_null || e->children[i]->dtype.id != TypeId::String)
      throw CometError(f + ": " + what + " must be a string literal");
    return e->children[i]->lit_bytes;
  };
  auto int_lit = [&](size_t i, const char* what) -> long long {
    if (i >= e->children.size() || e->children[i]->kind != ExprKind::Literal || e->children[i]->lit_null || !e->children[i]->dtype.is_integer())
      throw CometError(f + ": " + what + " must be an integer literal");
    return e->children[i]->lit_i64;
  };
  DerivedCol dc;
  dc.kind = 3;
  dc.src = subject->bound_index;
  if (f == "reverse") dc.op = 1;
  else if (f == "repeat") {
    dc.op = 2;
    dc.arg_k = int_lit(1, "the count");
    if (dc.arg_k < 0) throw CometError("repeat with a negative count is not supported (the reference fails the task)");
  } else if (f == "replace") {
    dc.op = 3;
    dc.arg_a = str_lit(1, "the search string");
    dc.arg_b = e->children.size() > 2 ? str_lit(2, "the replacement") : std::string();
  } else if (f == "substring_index" || f == "substr_index") {
    dc.op = 4;
    dc.arg_a = str_lit(1, "the delimiter");
    dc.arg_k = int_lit(2, "the count");
  } else if (f == "md5") dc.op = 10;
  else if (f == "sha1") dc.op = 11;
  else {
    const long long bits = int_lit(1, "the bit length");
    dc.op = bits == 224 ? 12 : (bits == 256 || bits == 0) ? 13 : bits == 384 ? 14 : bits == 512 ? 15 : -1;
    if (dc.op < 0) throw CometError("sha2 with a bit length of " + std::to_string(bits) + " (NULL in Spark) is not supported by the MI355X native engine");
  }
  dc.type = DType::of(TypeId::String);
  size_t k = 0;
  for (; k < g_derived->size(); k++) {
    const DerivedCol& o = (*g_derived)[k];
    if (o.kind == 3 && o.src == dc.src && o.op == dc.op && o.arg_a == dc.arg_a && o.arg_b == dc.arg_b && o.arg_k == dc.arg_k) break;
  }
  if (k == g_derived->size()) g_derived->push_back(dc);
  auto b = std::make_shared<Expr>();
  b->kind = ExprKind::Bound;
  b->proto_tag = 3;
  b->bound_index = derived_index(nsrc, k);
  b->dtype = dc.type;
  b->has_dtype = true;
  return b;
}
ExprP lower_split(const ExprP& e) {
  const bool all = e->func == "regexp_extract_all";      // (string_funcs/regexp_extract_all.rs: (subject, pattern, [idx = 1]) → group idx of every match)
  const std::string fname = all ? "regexp_extract_all" : "split";
  if (e->children.size() < 2 || e->children.size() > 3)
    throw CometError(all ? "regexp_extract_all expects 2 or 3 arguments (subject, pattern, [idx]), got " + std::to_string(e->children.size())
                         : "split expects 2 or 3 arguments (string, pattern, [limit]), got " + std::to_string(e->children.size()));
  const ExprP &subject = e->children[0], &pat = e->children[1];
  if (!g_derived || !g_source_cols) throw CometError(fname + " is supported in Projection / Filter chains only");
  const int nsrc = (int)g_source_cols->size();
  if (subject->kind != ExprKind::Bound || subject->bound_index < 0 || subject->bound_index >= nsrc || !subject->has_dtype || subject->dtype.id != TypeId::String)
    throw CometError(fname + " is supported over a Utf8 COLUMN of the source (not over a computed string) by the MI355X native engine");
  if (pat->kind != ExprKind::Literal || (pat->dtype.id != TypeId::String && !pat->lit_null)) throw CometError(fname + " pattern must be a string literal");
  int limit = all ? 1 : -1;      // (regexp_extract_all: the group index travels here)
  if (e->children.size() == 3) {
    const ExprP& l = e->children[2];
    if (l->kind != ExprKind::Literal || l->lit_null || l->dtype.id != TypeId::Int32) throw CometError(all ? "regexp_extract_all idx must be an Int32 scalar" : "split limit argument must be an Int32 scalar");
    limit = (int)std::max<long long>(std::min<long long>(l->lit_i64, 0x7fffffffLL), all ? -0x7fffffffLL : -1);
  }
  if (pat->lit_null) throw CometError(fname + " with a NULL pattern is not supported by the MI355X native engine");
  DerivedCol dc;
  dc.kind = all ? 2 : 1;
  dc.src = subject->bound_index;
  dc.limit = limit;
  if (all) {
    dc.prog = compile_regex_captures(pat->lit_bytes, 0, "regexp_extract_all").words;
    if (limit != 0) dc.prog2 = compile_regex_captures(pat->lit_bytes, limit, "regexp_extract_all").words;      // (the reference's message for an index out of range)
  } else {
    try {
      dc.prog = compile_regex_captures(pat->lit_bytes, 0, "split").words;
    } catch (const CometError& err) {
      const std::string m = err.what();
      if (m.find("not supported") != std::string::npos) throw;
      throw CometError("Invalid regex pattern '" + pat->lit_bytes + "': " + m);      // split.rs:201-203
    }
  }
  dc.type.id = TypeId::List;
  DType elem = DType::of(TypeId::String);
  dc.type.kids.push_back(elem);
  dc.type.kid_names.push_back("item");
  dc.type.kid_nullable.push_back(all);      // (regexp_extract_all's item field is nullable, split's is not: regexp_extract_all.rs:101, split.rs:318)
  for (size_t k = 0; k < g_derived->size(); k++) {
    const DerivedCol& o = (*g_derived)[k];
    if (o.kind == dc.kind && o.src == dc.src && o.limit == dc.limit && o.prog == dc.prog) {
      auto b = std::make_shared<Expr>();
      b->kind = ExprKind::Bound;
      b->proto_tag = 3;
      b->bound_index = derived_index(nsrc, k);
      b->dtype = o.type;
      b->has_dtype = true;
      return b;
    }
  }
  g_derived->push_back(dc);
  auto b = std::make_shared<Expr>();
  b->kind = ExprKind::Bound;
  b->proto_tag = 3;
  b->bound_index = derived_index(nsrc, g_derived->size() - 1);      // (two columns per derived list: the list, then its elements)
  b->dtype = dc.type;
  b->has_dtype = true;
  return b;
}
ExprP lower_struct_field(const ExprP& e, const ExprP& child) {
  if (child->kind != ExprKind::Bound || !child->has_dtype || child->dtype.id != TypeId::Struct)
    throw CometError("GetStructField of anything but a struct COLUMN is not supported yet");
  if (e->bound_index < 0 || (size_t)e->bound_index >= child->dtype.kids.size()) throw CometError("GetStructField: ordinal " + std::to_string(e->bound_index) + " is out of range");
  if (g_source_cols)
    for (auto& c : *g_source_cols)
      if (c->dtype.virt_parent == child->bound_index && c->dtype.virt_kid == e->bound_index) return c;
  throw CometError("GetStructField is supported in Projection / Filter chains over a materialised source (a Parquet scan, a join) only");
}

ExprP substitute(const ExprP& e, const std::vector<ExprP>& cols, std::map<const Expr*, ExprP>& memo) {
  auto it = memo.find(e.get());
  if (it != memo.end()) return it->second;
  ExprP out;
  if (e->kind == ExprKind::GetStructField && e->children.size() == 1) {
    out = lower_struct_field(e, substitute(e->children[0], cols, memo));
  } else if (e->kind == ExprKind::ScalarFunc && is_strfn(e->func)) {
    auto n = std::make_shared<Expr>(*e);
    for (auto& c : n->children) c = substitute(c, cols, memo);
    out = lower_strfn(n);
  } else if (e->kind == ExprKind::ScalarFunc && (e->func == "split" || e->func == "regexp_extract_all")) {
    auto n = std::make_shared<Expr>(*e);
    for (auto& c : n->children) c = substitute(c, cols, memo);
    out = lower_split(n);
  } else if (e->kind == ExprKind::Bound) {
    if (e->bound_index < 0 || (size_t)e->bound_index >= cols.size())
      throw CometError("Column index " + std::to_string(e->bound_index) + " is out of bound. Schema has " +
                       std::to_string(cols.size()) + " fields");
    out = cols[e->bound_index];
  } else if (e->children.empty()) {
    out = e;
  } else {
    auto n = std::make_shared<Expr>(*e);
    bool changed = false;
    for (auto& c : n->children) {
      ExprP s = substitute(c, cols, memo);
      if (s != c) changed = true;
      c = s;
    }
    out = changed ? n : e;
  }
  memo[e.get()] = out;
  return out;
}

void split_conjuncts(const ExprP& e, std::vector<ExprP>& out) {
  if (e->kind == ExprKind::And && e->children.size() == 2) {
    split_conjuncts(e->children[0], out);
    split_conjuncts(e->children[1], out);
  } else {
    out.push_back(e);
  }
}

// ---------------------------------------------------------------------------------------------
// Emitter
// ---------------------------------------------------------------------------------------------
struct Gen {
  const std::vector<DType>& in_types;
  const std::vector<bool>& in_valid;
  std::vector<bool> in_used;
  int nvar = 0;
  std::string decls;
  struct Stage { std::string loads, body; };
  std::vector<Stage> stages;
  std::map<std::string, Val> cse;
  std::map<const Expr*, std::string> keymemo;
  std::map<int, Val> col_cache;
  bool uses_err = false;
  std::vector<int> str_fixed_len;   // per input column: uniform Utf8 length verified by the executor, -1 = variable
  bool eager_loads = false;   // hoist every column load into the first stage (no lazy loading after predicates)
  // where column `idx` lives: kernel-argument slot and row expression (joins read two tables with different rows)
  std::function<std::pair<int, std::string>(int)> locate = [](int idx) { return std::make_pair(idx, std::string("idx[r]")); };

  Gen(const std::vector<DType>& t, const std::vector<bool>& v) : in_types(t), in_valid(v), in_used(t.size(), false) {
    stages.emplace_back();
  }

  std::string newvar(const char* ctype) {
    std::string n = "x" + std::to_string(nvar++);
    decls += std::string("    ") + ctype + " " + n + "[R];\n";
    return n + "[r]";
  }
  // Software pipelining: variables loaded in the FIRST stage live in a struct `L` that the kernel template fills for
  // tile t+1 before it computes tile t (loads stay in flight across the whole compute phase).
  bool pipelined = false;
  std::string ldecls, laliases;
  bool load_targets_front() const { return eager_loads || stages.size() == 1; }
  std::string newloadvar(const char* ctype) {
    if (!pipelined || !load_targets_front()) return newvar(ctype);
    std::string n = "x" + std::to_string(nvar++);
    ldecls += std::string("    ") + ctype + " " + n + "[R];\n";
    laliases += "    auto& " + n + " = ld." + n + ";\n";
    return n + "[r]";
  }
  void stmt(const std::string& s) { stages.back().body += "      " + s + "\n"; }
  void load(const std::string& s) { (eager_loads ? stages.front() : stages.back()).loads += "      " + s + "\n"; }
  void next_stage() { stages.emplace_back(); }

  // materialise an expression string into a variable (so later uses do not recompute it)
  Val named(Val x) {
    if (!x.v.empty() && x.v[0] == 'x' && x.v.find_first_of(" (") == std::string::npos) return x;
    std::string n = newvar(rep_ctype(x.rep));
    stmt(n + " = " + x.v + ";");
    x.v = n;
    if (!x.ok.empty() && !(x.ok[0] == 'x' && x.ok.find_first_of(" (") == std::string::npos)) {
      std::string o = newvar("bool");
      stmt(o + " = " + x.ok + ";");
      x.ok = o;
    }
    return x;
  }

  std::vector<ExprP> keymemo_alive;
  std::string key_of(const ExprP& e) {
    auto it = keymemo.find(e.get());
    if (it != keymemo.end()) return it->second;
    std::ostringstream k;
    k << (int)e->kind << ":" << (int)e->dtype.id << "," << e->dtype.precision << "," << e->dtype.scale << ":"
      << (int)e->eval_mode << e->fail_on_error << e->negated;
    if (e->kind == ExprKind::Bound) k << "#" << e->bound_index;
    // (`func` is a ScalarFunc's name — and the time zone of a Cast / Hour / Minute / Second / TruncTimestamp / UnixTimestamp: the same child in two zones
    // is two values)
    if (!e->func.empty()) k << "F" << e->func.size() << ":" << e->func;
    if (e->is_spark4_plus) k << "S4";
    if (e->one_based) k << "OB";
    if (e->kind == ExprKind::Subquery) k << "Q" << e->lit_i64;
    if (e->check_divide_overflow) k << "DO";
    if (e->kind == ExprKind::CaseWhen) k << "W" << e->n_when;
    if (e->kind == ExprKind::Literal) {
      k << "L" << e->lit_case << e->lit_null << e->lit_bool << ":" << e->lit_i64 << ":";
      uint64_t fb;
      memcpy(&fb, &e->lit_f64, 8);
      k << fb << ":" << (uint64_t)((u128)e->lit_dec >> 64) << "_" << (uint64_t)(u128)e->lit_dec << ":" << e->lit_bytes;
    }
    k << "(";
    for (auto& c : e->children) k << key_of(c) << ";";
    k << ")";
    keymemo[e.get()] = k.str();
    keymemo_alive.push_back(e);   // the memo is keyed by node address: keep every memoised node alive so that no address is reused
    return k.str();
  }

  Val column(int idx) {
    auto it = col_cache.find(idx);
    if (it != col_cache.end()) return it->second;
    if (idx < 0 || (size_t)idx >= in_types.size()) throw CometError("Column index " + std::to_string(idx) + " is out of bound");
    const DType& t = in_types[idx];
    in_used[idx] = true;
    Val x;
    x.t = t;
    x.rep = rep_for_type(t);
    x.maxabs = type_maxabs(t);
    auto loc = locate(idx);
    std::string c = "prm.in[" + std::to_string(loc.first) + "]";
    const std::string row = loc.second;
    std::string n = newloadvar(rep_ctype(x.rep));
    std::string ldx;
    switch (t.id) {
      case TypeId::Bool: ldx = "comet::ld_bool(" + c + ", " + row + ")"; break;
      case TypeId::Int8: ldx = "(i32)comet::ld<i8>(" + c + ", " + row + ")"; break;
      case TypeId::Int16: ldx = "(i32)comet::ld<i16>(" + c + ", " + row + ")"; break;
      case TypeId::Int32: case TypeId::Date: ldx = "comet::ld<i32>(" + c + ", " + row + ")"; break;
      case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: ldx = "comet::ld<i64>(" + c + ", " + row + ")"; break;
      case TypeId::Float: ldx = "comet::ld<float>(" + c + ", " + row + ")"; break;
      case TypeId::Double: ldx = "comet::ld<double>(" + c + ", " + row + ")"; break;
      case TypeId::Decimal:
        ldx = t.precision <= 18 ? "comet::ld_dec_lo(" + c + ", " + row + ")" : "comet::ld<i128>(" + c + ", " + row + ")";
        break;
      case TypeId::String:
        if ((size_t)idx < str_fixed_len.size() && str_fixed_len[idx] >= 0 && str_fixed_len[idx] <= 15) {
          ldx = "comet::ld_str_fixed<" + std::to_string(str_fixed_len[idx]) + ">(" + c + ", " + row + ")";
          break;
        }
        // ≤15-byte strings travel packed (str16); a longer value raises error bit 64 → explicit "not supported yet"
        uses_err = true;
        ldx = "";
        load("{ bool tl_ = false; " + n + " = comet::ld_str16(" + c + ", " + row + ", tl_); if (tl_) atomicOr((unsigned int*)prm.out[" +
             std::to_string(kOutErr) + "], 64u); }");
        break;
      default: throw CometError("Unsupported scan column type: " + t.str());
    }
    if (!ldx.empty()) load(n + " = " + ldx + ";");
    x.v = n;
    if (in_valid[idx]) {
      std::string o = newloadvar("bool");
      load(o + " = comet::ld_valid(" + c + ", " + row + ");");
      x.ok = o;
    }
    col_cache[idx] = x;
    return x;
  }

  static std::string and_ok(const std::string& a, const std::string& b) {
    if (a.empty()) return b;
    if (b.empty()) return a;
    return "(" + a + " && " + b + ")";
  }

  Val literal(const Expr& e) {
    Val x;
    x.t = e.dtype;
    if (e.dtype.id == TypeId::Null) throw CometError("NullType literal is not supported in GPU pipeline");
    x.rep = rep_for_type(e.dtype);
    if (e.lit_null) {
      x.is_null_lit = true;
      x.ok = "false";
      x.maxabs = 0;
      switch (x.rep) {
        case Rep::STR: x.v = "comet::str16{0ull, 0ull}"; break;
        case Rep::B: x.v = "false"; break;
        case Rep::F32: x.v = "0.0f"; break;
        case Rep::F64: x.v = "0.0"; break;
        case Rep::I128: x.v = "(i128)0"; break;
        default: x.v = std::string("(") + rep_ctype(x.rep) + ")0";
      }
      return x;
    }
    switch (e.dtype.id) {
      case TypeId::Bool: x.v = e.lit_bool ? "true" : "false"; break;
      case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: case TypeId::Date:
        x.v = "(i32)" + std::to_string((int32_t)e.lit_i64);
        x.maxabs = (u128)(e.lit_i64 < 0 ? -(i128)e.lit_i64 : (i128)e.lit_i64);
        break;
      case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz:
        x.v = lit_i64(e.lit_i64);
        x.maxabs = (u128)(e.lit_i64 < 0 ? -(i128)e.lit_i64 : (i128)e.lit_i64);
        break;
      case TypeId::Float: {
        float f = (float)e.lit_f64;
        uint32_t u;
        memcpy(&u, &f, 4);
        x.v = "__int_as_float((int)" + std::to_string((int32_t)u) + ")";
        break;
      }
      case TypeId::Double: {
        uint64_t u;
        memcpy(&u, &e.lit_f64, 8);
        x.v = "__longlong_as_double(" + lit_i64((int64_t)u) + ")";
        break;
      }
      case TypeId::Decimal: {
        i128 v = e.lit_dec;
        x.maxabs = (u128)(v < 0 ? -v : v);
        if (x.rep == Rep::I64) x.v = lit_i64((int64_t)v);
        else x.v = lit_i128(v);
        break;
      }
      case TypeId::String: {
        if (e.lit_bytes.size() > 15) throw CometError("string literals longer than 15 bytes are not supported in the GPU pipeline yet");
        uint64_t a = 0, b = 0;
        for (size_t k = 0; k < e.lit_bytes.size(); k++) {
          uint64_t byte = (uint8_t)e.lit_bytes[k];
          if (k < 8) a |= byte << (8 * k);
          else b |= byte << (8 * (k - 8));
        }
        b |= (uint64_t)e.lit_bytes.size() << 56;
        x.v = "comet::str16{" + hex64(a) + ", " + hex64(b) + "}";
        break;
      }
      default: throw CometError("Unsupported literal type: " + e.dtype.str());
    }
    return x;
  }

  std::string as128(const Val& x) { return x.rep == Rep::I128 ? x.v : "(i128)(" + x.v + ")"; }
  std::string as64(const Val& x) {
    if (x.rep == Rep::I64) return x.v;
    return "(i64)(" + x.v + ")";
  }

  // total-order key for floats (arrow-ord compares floats with IEEE totalOrder)
  std::string fkey(const Val& x) {
    if (x.rep == Rep::F64) return "comet::f64_total_key(" + x.v + ")";
    return "comet::f32_total_key(" + x.v + ")";
  }

  Val compare(ExprKind k, Val a, Val b) {
    Val r;
    r.t = DType::of(TypeId::Bool);
    r.rep = Rep::B;
    std::string op;
    switch (k) {
      case ExprKind::Eq: case ExprKind::EqNullSafe: op = "=="; break;
      case ExprKind::Neq: case ExprKind::NeqNullSafe: op = "!="; break;
      case ExprKind::Gt: op = ">"; break;
      case ExprKind::GtEq: op = ">="; break;
      case ExprKind::Lt: op = "<"; break;
      case ExprKind::LtEq: op = "<="; break;
      default: throw CometError("bad comparison");
    }
    if (a.t.id == TypeId::Decimal && b.t.id == TypeId::Decimal && a.t.scale != b.t.scale)
      throw CometError("Decimal comparison requires equal scales: " + a.t.str() + " vs " + b.t.str());
    bool af = a.rep == Rep::F32 || a.rep == Rep::F64, bf = b.rep == Rep::F32 || b.rep == Rep::F64;
    if (af != bf || (af && a.rep != b.rep)) throw CometError("Comparison of mismatched types: " + a.t.str() + " vs " + b.t.str());
    if ((a.rep == Rep::B) != (b.rep == Rep::B)) throw CometError("Comparison of mismatched types: " + a.t.str() + " vs " + b.t.str());
    if (a.rep == Rep::STR || b.rep == Rep::STR) {
      if (a.rep != b.rep) throw CometError("Comparison of mismatched types: " + a.t.str() + " vs " + b.t.str());
      if (!(k == ExprKind::Eq || k == ExprKind::Neq || k == ExprKind::EqNullSafe || k == ExprKind::NeqNullSafe))
        throw CometError("ordering comparisons on Utf8 are not supported in the GPU pipeline yet");
      a = named(a);
      b = named(b);
      std::string eq = "(" + a.v + ".a == " + b.v + ".a && " + a.v + ".b == " + b.v + ".b)";
      if (k == ExprKind::EqNullSafe || k == ExprKind::NeqNullSafe) {
        std::string aok = a.ok.empty() ? "true" : a.ok, bok = b.ok.empty() ? "true" : b.ok;
        std::string e2 = "((" + aok + " && " + bok + " && " + eq + ") || (!" + aok + " && !" + bok + "))";
        r.v = (k == ExprKind::EqNullSafe) ? e2 : "(!" + e2 + ")";
        return r;
      }
      r.v = (k == ExprKind::Eq) ? eq : "(!" + eq + ")";
      r.ok = and_ok(a.ok, b.ok);
      return r;
    }
    std::string l, rr;
    if (af) { l = fkey(a); rr = fkey(b); }
    else if (a.rep == Rep::B) { l = "(int)" + a.v; rr = "(int)" + b.v; }
    else if (a.rep == Rep::I128 || b.rep == Rep::I128) { l = as128(a); rr = as128(b); }
    else if (a.rep == Rep::I64 || b.rep == Rep::I64) { l = "(i64)" + a.v; rr = "(i64)" + b.v; }
    else { l = a.v; rr = b.v; }
    std::string cmp = "(" + l + " " + op + " " + rr + ")";
    if (k == ExprKind::EqNullSafe || k == ExprKind::NeqNullSafe) {
      std::string aok = a.ok.empty() ? "true" : a.ok, bok = b.ok.empty() ? "true" : b.ok;
      std::string eq = "((" + aok + " && " + bok + " && (" + l + " == " + rr + ")) || (!" + aok + " && !" + bok + "))";
      r.v = (k == ExprKind::EqNullSafe) ? eq : "(!" + eq + ")";
      return r;
    }
    r.v = cmp;
    r.ok = and_ok(a.ok, b.ok);
    return r;
  }

  Val logic(ExprKind k, Val a, Val b) {
    Val r;
    r.t = DType::of(TypeId::Bool);
    r.rep = Rep::B;
    if (a.rep != Rep::B || b.rep != Rep::B) throw CometError("AND/OR expects boolean operands");
    a = named(a);
    b = named(b);
    if (a.ok.empty() && b.ok.empty()) {
      r.v = "(" + a.v + (k == ExprKind::And ? " && " : " || ") + b.v + ")";
      return r;
    }
    std::string aok = a.ok.empty() ? "true" : a.ok, bok = b.ok.empty() ? "true" : b.ok;
    if (k == ExprKind::And) {
      // Kleene: FALSE dominates NULL
      r.v = "(" + aok + " && " + a.v + " && " + bok + " && " + b.v + ")";
      r.ok = "((" + aok + " && !" + a.v + ") || (" + bok + " && !" + b.v + ") || (" + aok + " && " + bok + "))";
    } else {
      r.v = "((" + aok + " && " + a.v + ") || (" + bok + " && " + b.v + "))";
      r.ok = "((" + aok + " && " + a.v + ") || (" + bok + " && " + b.v + ") || (" + aok + " && " + bok + "))";
    }
    return r;
  }

  // raise a Spark error for the rows where `cond` holds (ANSI mode)
  void raise_if(const std::string& cond, int code) {
    uses_err = true;
    if (current_context() && (code == 0 || code == 1 || code == 8 || code == 15)) {
      // an error without a value — but with the SQL fragment of its expression: a site, so that the executor finds the context
      ErrSite site;
      site.value = ErrSite::NoValue;
      switch (code) {
        case 0: site.error_type = "ArithmeticOverflow"; site.error_class = "ARITHMETIC_OVERFLOW"; site.from_type = "decimal"; break;
        case 1: site.error_type = "ArithmeticOverflow"; site.error_class = "ARITHMETIC_OVERFLOW"; site.from_type = "integer"; break;
        case 8: site.error_type = "DivideByZero"; site.error_class = "DIVIDE_BY_ZERO"; break;
        default: site.error_type = "RemainderByZero"; site.error_class = "REMAINDER_BY_ZERO"; break;
      }
      raise_value(cond, code, site, "0");
      return;
    }
    stmt("if (" + cond + ") atomicOr((unsigned int*)prm.out[" + std::to_string(kOutErr) + "], " + std::to_string(1u << code) + "u);");
  }

  // the expression whose code is being emitted (innermost last): its QueryContext goes with the errors it raises (planner.rs:302-316, 582-597 —
  // the reference looks the context up under the expression's expr_id: both must be there)
  std::vector<std::shared_ptr<QueryContext>> ctx_stack;
  std::shared_ptr<QueryContext> current_context() const { return ctx_stack.empty() ? nullptr : ctx_stack.back(); }
  // a site's id: its content and its ordinal among the pipeline's sites of that content — not the SQL text, which is noted beside the pipeline
  uint32_t site_id(const ErrSite& site) {
    int ord = 0;
    if (g_site_ordinals) {
      const std::string c = site.error_type + "|" + site.from_type + "|" + site.to_type + "|" + std::to_string(site.precision) + "|" + std::to_string(site.scale) + "|" +
                            std::to_string(site.value) + "|" + site.suffix;
      ord = (*g_site_ordinals)[c]++;
    }
    const uint32_t id = register_err_site(site, ord);
    if (g_site_sink) g_site_sink->emplace_back(id, current_context());      // (a site without a context is listed too: it must not inherit another pipeline's)
    return id;
  }
  // … and leave the offending value for the error's JSON (err_sites.cpp): a number's bits, or a string's bytes
  void raise_value(const std::string& cond, int code, const ErrSite& site, const std::string& lo, const std::string& hi = "0") {
    uses_err = true;
    const std::string eb = "prm.out[" + std::to_string(kOutErr) + "]";
    stmt("if (" + cond + ") { atomicOr((unsigned int*)" + eb + ", " + std::to_string(1u << code) + "u); comet::err_detail(" + eb + ", " +
         std::to_string(site_id(site)) + "u, (u64)(" + lo + "), (u64)(" + hi + ")); }");
  }
  void raise_value128(const std::string& cond, int code, const ErrSite& site, const Val& x) {
    if (x.rep == Rep::I128) raise_value(cond, code, site, "(u128)(" + x.v + ")", "((u128)(" + x.v + ") >> 64)");
    else raise_value(cond, code, site, "(i64)(" + x.v + ")", "((i64)(" + x.v + ") >> 63)");
  }
  void raise_str(const std::string& cond, int code, const ErrSite& site, const std::string& ptr, const std::string& len) {
    uses_err = true;
    const std::string eb = "prm.out[" + std::to_string(kOutErr) + "]";
    stmt("if (" + cond + ") { atomicOr((unsigned int*)" + eb + ", " + std::to_string(1u << code) + "u); comet::err_detail_str(" + eb + ", " +
         std::to_string(site_id(site)) + "u, (const u8*)(" + ptr + "), (i32)(" + len + ")); }");
  }
  static ErrSite out_of_range_site(const DType& to) {      // decimal_overflow_error (common/src/error.rs:769-775): the unscaled value, the target's (p, s)
    ErrSite s;
    s.error_type = "NumericValueOutOfRange";
    s.error_class = "NUMERIC_VALUE_OUT_OF_RANGE.WITH_SUGGESTION";
    s.precision = to.precision;
    s.scale = to.scale;
    return s;
  }
  static std::string spark_type_name(const DType& t) {
    switch (t.id) {
      case TypeId::Bool: return "BOOLEAN";
      case TypeId::Int8: return "TINYINT";
      case TypeId::Int16: return "SMALLINT";
      case TypeId::Int32: return "INT";
      case TypeId::Int64: return "BIGINT";
      case TypeId::Float: return "FLOAT";
      case TypeId::Double: return "DOUBLE";
      case TypeId::Date: return "DATE";
      case TypeId::Timestamp: return "TIMESTAMP";
      case TypeId::TimestampNtz: return "TIMESTAMP_NTZ";
      case TypeId::String: return "STRING";
      case TypeId::Decimal: return "DECIMAL(" + std::to_string(t.precision) + "," + std::to_string(t.scale) + ")";
      default: return t.str();
    }
  }

  // CheckOverflow / bound check shared tail: value `v` (rep), ok expr, bound 10^p-1.  `site`: what the error names (the value defaults to x's)
  Val bound_check(Val x, int p, bool fail_on_error, int err_code, const ErrSite* site = nullptr, const std::string* value_lo = nullptr) {
    u128 bound = pow10_u128(p) - 1;
    if (x.maxabs <= bound) return x;  // statically in range: nothing to emit
    x = named(x);
    std::string fits = x.rep == Rep::I128 ? "comet::dec_fits(" + x.v + ", " + lit_u128(bound) + ")"
                                          : "comet::dec_fits64(" + x.v + ", " + hex64((uint64_t)bound) + ")";
    if (x.rep != Rep::I128 && bound >= ((u128)1 << 63)) return x;  // i64 value cannot exceed a ≥2^63 bound
    if (fail_on_error) {
      if (site && value_lo) raise_value(and_ok(x.ok, "!" + fits), err_code, *site, *value_lo);
      else if (site) raise_value128(and_ok(x.ok, "!" + fits), err_code, *site, x);
      else raise_if(and_ok(x.ok, "!" + fits), err_code);
    } else {
      std::string o = newvar("bool");
      stmt(o + " = " + and_ok(x.ok, fits) + ";");
      x.ok = o;
    }
    x.maxabs = bound;
    return x;
  }

  Val decimal_binary(const Expr& e, Val a, Val b) {
    const int p1 = a.t.precision, s1 = a.t.scale, p2 = b.t.precision, s2 = b.t.scale;
    const bool mul = e.kind == ExprKind::Multiply;
    const bool addsub = e.kind == ExprKind::Add || e.kind == ExprKind::Subtract;
    if (e.kind == ExprKind::Divide || e.kind == ExprKind::IntegralDivide) {
      // planner.rs:1028-1057 → decimal_div / decimal_integral_div (spark-expr/src/math_funcs/div.rs:40-165)
      const bool integral = e.kind == ExprKind::IntegralDivide;
      if (!e.has_dtype || e.dtype.id != TypeId::Decimal) throw CometError("Expected Decimal128 return type");
      const int s3 = e.dtype.scale;
      const int l_exp = std::max(0, s2 + s3 + 1 - s1), r_exp = std::max(0, s1 - (s2 + s3 + 1));
      if (l_exp > 38 || p1 + l_exp > 76 || p2 + r_exp > 38)
        throw CometError("Decimal division " + a.t.str() + " / " + b.t.str() + " -> " + e.dtype.str() + " needs more than 256-bit intermediates; not supported by the GPU pipeline yet");
      a = named(a);
      b = named(b);
      Val r;
      r.rep = Rep::I128;
      r.t = e.dtype;
      r.ok = and_ok(a.ok, b.ok);
      std::string val = newvar("i128"), dz = newvar("bool");
      stmt(val + " = comet::dec_div(" + as128(a) + ", " + as128(b) + ", " + lit_u128(pow10_u128(l_exp)) + ", " + lit_u128(pow10_u128(r_exp)) + ", " + dz + (integral ? ", true" : "") + ");");
      if (e.eval_mode == EvalMode::Ansi) raise_if(and_ok(r.ok, dz), 8);
      // quotient_to_i128 (div.rs:57-68): with check_divide_overflow under ANSI a quotient outside LONG raises ARITHMETIC_OVERFLOW
      if (integral && e.check_divide_overflow && e.eval_mode == EvalMode::Ansi)
        raise_if(and_ok(r.ok, "(" + val + " != (i128)(i64)" + val + ")"), 1);
      r.v = val;
      r.maxabs = ~(u128)0 >> 1;   // unchecked: the plan's CheckOverflow bounds it
      return r;
    }
    if (e.kind == ExprKind::Remainder) {
      // create_modulo_expr (math_funcs/modulo_expr.rs:137-206): arrow-arith's decimal rem at scale max(s1, s2); a zero divisor is NULL outside ANSI
      // mode (null_if_zero_primitive) and REMAINDER_BY_ZERO in it
      if (!e.has_dtype || e.dtype.id != TypeId::Decimal) throw CometError("Expected Decimal128 return type");
      const int smax = std::max(s1, s2);
      if (e.dtype.scale != smax) throw CometError("Decimal remainder: return type " + e.dtype.str() + " does not have the operands' larger scale");
      a = named(a);
      b = named(b);
      Val r;
      r.t = e.dtype;
      r.ok = and_ok(a.ok, b.ok);
      std::string val = newvar("i128"), dz = newvar("bool");
      stmt(val + " = comet::dec_rem(" + as128(a) + ", " + as128(b) + ", " + lit_u128(pow10_u128(smax - s1)) + ", " + lit_u128(pow10_u128(smax - s2)) + ", " + dz + ");");
      if (e.eval_mode == EvalMode::Ansi) raise_if(and_ok(r.ok, dz), 15);      // RemainderByZero
      else {
        std::string o = newvar("bool");
        stmt(o + " = " + and_ok(r.ok, "!" + dz) + ";");
        r.ok = o;
      }
      r.maxabs = type_maxabs(e.dtype);
      r.rep = rep_for_type(e.dtype);
      r.v = r.rep == Rep::I64 ? "(i64)" + val : val;
      return r;
    }
    if (!mul && !addsub) throw CometError(std::string("Decimal ") + expr_name(e.proto_tag) + " is not supported in the GPU pipeline yet");
    const int smax = std::max(s1, s2);
    // planner.rs:1000-1008
    const bool wide = (addsub && smax + std::max(p1 - s1, p2 - s2) >= 38) || (mul && p1 + p2 >= 38);
    Val r;
    r.rep = Rep::I128;
    r.ok = and_ok(a.ok, b.ok);
    if (wide) {
      if (!e.has_dtype || e.dtype.id != TypeId::Decimal) throw CometError("Expected Decimal128 return type");
      const int p_out = e.dtype.precision, s_out = e.dtype.scale;
      const u128 bound = pow10_u128(p_out) - 1;
      r.t = e.dtype;
      r.wide_decimal = true;
      a = named(a);
      b = named(b);
      if (mul && s1 + s2 == s_out && (a.rep != Rep::I128 || b.rep != Rep::I128)) {
        // one factor fits in 64 bits and no rescale: 128×64 product with the bound check fused
        const Val& w = a.rep == Rep::I128 ? a : b;
        const Val& n64 = a.rep == Rep::I128 ? b : a;
        std::string val = newvar("i128"), fit = newvar("bool");
        stmt(fit + " = comet::i128_mul_i64_fits(" + as128(w) + ", (i64)" + n64.v + ", " + lit_u128(bound) + ", " + val + ");");
        if (e.eval_mode == EvalMode::Ansi) raise_if(and_ok(r.ok, "!" + fit), 0);
        else {
          std::string o = newvar("bool");
          stmt(o + " = " + and_ok(r.ok, fit) + ";");
          r.ok = o;
        }
        r.v = val;
        r.maxabs = bound;
        return r;
      }
      std::string raw = newvar("comet::i256");
      int scale_diff;
      if (mul) {
        scale_diff = s1 + s2 - s_out;
        stmt(raw + " = comet::i128_mul_i128(" + as128(a) + ", " + as128(b) + ");");
      } else {
        scale_diff = smax - s_out;
        std::string l = "comet::i128_mul_i128(" + as128(a) + ", " + lit_i128((i128)pow10_u128(smax - s1)) + ")";
        std::string rr = "comet::i128_mul_i128(" + as128(b) + ", " + lit_i128((i128)pow10_u128(smax - s2)) + ")";
        stmt(raw + " = comet::" + (e.kind == ExprKind::Add ? "i256_add(" : "i256_sub(") + l + ", " + rr + ");");
      }
      if (scale_diff > 0) {
        if (scale_diff > 38) throw CometError("wide decimal rescale by more than 10^38 is not supported");
        stmt(raw + " = comet::i256_div_pow10_half_up(" + raw + ", " + lit_u128(pow10_u128(scale_diff)) + ");");
      } else if (scale_diff < 0) {
        if (-scale_diff > 38) throw CometError("wide decimal rescale by more than 10^38 is not supported");
        // raw.wrapping_mul(10^k): two's-complement wrapping multiply equals the unsigned one
        stmt(raw + " = comet::u256_mul_u128_wrapping(" + raw + ", " + lit_u128(pow10_u128(-scale_diff)) + ");");
      }
      std::string val = newvar("i128");
      std::string fit = newvar("bool");
      stmt(fit + " = comet::i256_fits_bound(" + raw + ", " + lit_u128(bound) + ", " + val + ");");
      if (e.eval_mode == EvalMode::Ansi) {
        raise_if(and_ok(r.ok, "!" + fit), 0);
      } else {
        std::string o = newvar("bool");
        stmt(o + " = " + and_ok(r.ok, fit) + ";");
        r.ok = o;
      }
      r.v = val;
      r.maxabs = bound;
      return r;
    }
    // narrow path: exact i128 arithmetic, result typed like arrow-arith's decimal kernels
    if (mul) {
      r.t = DType::decimal(std::min(38, p1 + p2 + 1), s1 + s2);
      r.maxabs = sat_mul(a.maxabs, b.maxabs);
      r.rep = rep_for_bound(r.maxabs);
      if (r.rep == Rep::I64) r.v = "(" + as64(a) + " * " + as64(b) + ")";
      else if (a.rep != Rep::I128 && b.rep != Rep::I128) r.v = "((i128)" + as64(a) + " * (i128)" + as64(b) + ")";
      else r.v = "(" + as128(a) + " * " + as128(b) + ")";
    } else {
      r.t = DType::decimal(std::min(38, std::max(p1 - s1, p2 - s2) + smax + 1), smax);
      u128 fa = pow10_u128(smax - s1), fb = pow10_u128(smax - s2);
      r.maxabs = sat_add(sat_mul(a.maxabs, fa), sat_mul(b.maxabs, fb));
      r.rep = rep_for_bound(r.maxabs);
      const char* op = e.kind == ExprKind::Add ? " + " : " - ";
      if (r.rep == Rep::I64) {
        std::string l = fa == 1 ? as64(a) : "(" + as64(a) + " * " + lit_i64((int64_t)fa) + ")";
        std::string rr = fb == 1 ? as64(b) : "(" + as64(b) + " * " + lit_i64((int64_t)fb) + ")";
        r.v = "(" + l + op + rr + ")";
      } else {
        std::string l = fa == 1 ? as128(a) : "(" + as128(a) + " * " + lit_i128((i128)fa) + ")";
        std::string rr = fb == 1 ? as128(b) : "(" + as128(b) + " * " + lit_i128((i128)fb) + ")";
        r.v = "(" + l + op + rr + ")";
      }
    }
    if (r.maxabs == kUnbounded) throw CometError("internal: narrow decimal path with unbounded operand");
    return r;
  }

  Val arithmetic(const Expr& e, Val a, Val b) {
    if (a.t.id == TypeId::Decimal && b.t.id == TypeId::Decimal) return decimal_binary(e, a, b);
    // CometIntegralDivide always casts both sides to Decimal first (serde/arithmetic.scala:283-300)
    if (e.kind == ExprKind::IntegralDivide) throw CometError("IntegralDivide expects Decimal128 operands");
    if (!e.has_dtype) throw CometError("arithmetic expression without return_type");
    const DType& rt = e.dtype;
    Val r;
    r.t = rt;
    r.rep = rep_for_type(rt);
    r.ok = and_ok(a.ok, b.ok);
    const bool checked = (e.eval_mode == EvalMode::Ansi || e.eval_mode == EvalMode::Try);
    if (rt.is_integer()) {
      if (!(a.t.is_integer() && b.t.is_integer())) throw CometError("integer arithmetic on non-integer operands");
      const char* ct = r.rep == Rep::I64 ? "i64" : "i32";
      const char* ut = r.rep == Rep::I64 ? "u64" : "u32";
      if (e.kind == ExprKind::Remainder) {
        // create_modulo_expr / spark_modulo (math_funcs/modulo_expr.rs): a zero divisor is NULL outside ANSI mode (null_if_zero) and
        // REMAINDER_BY_ZERO in ANSI mode; the sign follows the dividend; MIN % -1 = 0 (wrapping_rem)
        a = named(a);
        b = named(b);
        const std::string bz = "((" + std::string(ct) + ")" + b.v + " == 0)";
        if (e.eval_mode == EvalMode::Ansi) raise_if(and_ok(r.ok, bz), 15);      // RemainderByZero
        else {
          std::string o = newvar("bool");
          stmt(o + " = " + and_ok(r.ok, "!" + bz) + ";");
          r.ok = o;
        }
        r.v = std::string("((") + bz + " || (" + ct + ")" + b.v + " == -1) ? (" + ct + ")0 : (" + ct + ")((" + ct + ")" + a.v + " % (" + ct + ")" + b.v + "))";
        if (rt.id == TypeId::Int8) r.v = "(i32)(i8)" + r.v;
        if (rt.id == TypeId::Int16) r.v = "(i32)(i16)" + r.v;
        return r;
      }
      std::string op;
      const char* builtin = nullptr;
      switch (e.kind) {
        case ExprKind::Add: op = "+"; builtin = "__builtin_add_overflow"; break;
        case ExprKind::Subtract: op = "-"; builtin = "__builtin_sub_overflow"; break;
        case ExprKind::Multiply: op = "*"; builtin = "__builtin_mul_overflow"; break;
        default: throw CometError(std::string("Integer ") + expr_name(e.proto_tag) + " is not supported in the GPU pipeline yet");
      }
      if (!checked) {
        // LEGACY: wrapping (DataFusion BinaryExpr add_wrapping, planner.rs:1128)
        std::string w = std::string("(") + ct + ")((" + ut + ")(" + ct + ")" + a.v + " " + op + " (" + ut + ")(" + ct + ")" + b.v + ")";
        if (rt.id == TypeId::Int8) w = "(i32)(i8)" + w;
        if (rt.id == TypeId::Int16) w = "(i32)(i16)" + w;
        r.v = w;
      } else {
        // checked_arithmetic.rs:54-124: ANSI → error, TRY → NULL (value slot zeroed)
        a = named(a);
        b = named(b);
        std::string out = newvar(ct), ovf = newvar("bool");
        std::string narrow;
        stmt("{ " + std::string(ct) + " t_; " + ovf + " = " + builtin + "((" + ct + ")" + a.v + ", (" + ct + ")" + b.v + ", &t_); " + out + " = t_; }");
        if (rt.id == TypeId::Int8) stmt(ovf + " = " + ovf + " || " + out + " != (i32)(i8)" + out + ";");
        if (rt.id == TypeId::Int16) stmt(ovf + " = " + ovf + " || " + out + " != (i32)(i16)" + out + ";");
        if (e.eval_mode == EvalMode::Ansi) {
          raise_if(and_ok(r.ok, ovf), 1);
          r.v = out;
        } else {
          std::string o = newvar("bool");
          stmt(o + " = " + and_ok(r.ok, "!" + ovf) + ";");
          r.ok = o;
          r.v = "(" + o + " ? " + out + " : (" + ct + ")0)";
        }
      }
      return r;
    }
    if (rt.is_float()) {
      const char* ct = r.rep == Rep::F64 ? "double" : "float";
      if (e.kind == ExprKind::Remainder) {
        // fmod = Rust's / Java's %; zero divisor: NULL (null_if_zero) or REMAINDER_BY_ZERO in ANSI mode (checked_float_modulo)
        a = named(a);
        b = named(b);
        const std::string bz = "((" + std::string(ct) + ")" + b.v + " == 0)";
        if (e.eval_mode == EvalMode::Ansi) raise_if(and_ok(r.ok, bz), 15);      // RemainderByZero
        else {
          std::string o = newvar("bool");
          stmt(o + " = " + and_ok(r.ok, "!" + bz) + ";");
          r.ok = o;
        }
        r.v = std::string(r.rep == Rep::F64 ? "fmod" : "fmodf") + "((" + ct + ")" + a.v + ", (" + ct + ")" + b.v + ")";
        return r;
      }
      std::string op;
      switch (e.kind) {
        case ExprKind::Add: op = "+"; break;
        case ExprKind::Subtract: op = "-"; break;
        case ExprKind::Multiply: op = "*"; break;
        case ExprKind::Divide: op = "/"; break;
        default: throw CometError(std::string("Float ") + expr_name(e.proto_tag) + " is not supported in the GPU pipeline yet");
      }
      if (e.kind == ExprKind::Divide && checked) throw CometError("ANSI/TRY float division is not supported in the GPU pipeline yet");
      // contraction off: a*b+c must round twice like the CPU path
      r.v = std::string("comet::fp_") + (e.kind == ExprKind::Add ? "add" : e.kind == ExprKind::Subtract ? "sub" : e.kind == ExprKind::Multiply ? "mul" : "div") +
            "((" + ct + ")" + a.v + ", (" + ct + ")" + b.v + ")";
      return r;
    }
    throw CometError("Arithmetic on " + rt.str() + " is not supported in the GPU pipeline");
  }

  Val cast(const Expr& e, Val c) {
    const DType& to = e.dtype;
    const DType& from = c.t;
    if (from == to) return c;
    Val r;
    r.t = to;
    r.rep = rep_for_type(to);
    r.ok = c.ok;
    auto is_intlike = [](const DType& t) { return t.is_integer(); };
    if (is_intlike(from) && is_intlike(to)) {
      // LEGACY: wrap on narrowing (conversion_funcs/numeric.rs); ANSI narrowing overflow → error
      if (type_width(to) >= type_width(from)) {
        r.v = std::string("(") + rep_ctype(r.rep) + ")" + c.v;
        r.maxabs = c.maxabs;
        return r;
      }
      const char* nt = to.id == TypeId::Int8 ? "i8" : to.id == TypeId::Int16 ? "i16" : "i32";
      c = named(c);
      r.v = std::string("(") + rep_ctype(r.rep) + ")(" + nt + ")" + c.v;
      if (e.eval_mode == EvalMode::Ansi) {
        ErrSite site;      // cast_int_to_int_macro (numeric.rs:282-305, 828-845): value.to_string() + Spark's literal suffix of the source type
        site.error_type = "CastOverFlow";
        site.error_class = "CAST_OVERFLOW";
        site.from_type = spark_type_name(from);
        site.to_type = spark_type_name(to);
        site.value = ErrSite::Int64;
        site.suffix = from.id == TypeId::Int64 ? "L" : from.id == TypeId::Int16 ? "S" : "";
        raise_value(and_ok(c.ok, std::string("(i64)(") + nt + ")" + c.v + " != (i64)" + c.v), 2, site, "(i64)" + c.v);
      } else if (e.eval_mode == EvalMode::Try) {
        // try_cast does not reach Comet's narrowing (cast.rs:284-293 `if eval_mode != Try`) but arrow's cast with safe = true (:236-241, 401-407):
        // NULL when the value does not fit the target type
        std::string o = newvar("bool");
        stmt(o + " = " + and_ok(c.ok, std::string("((i64)(") + nt + ")" + c.v + " == (i64)" + c.v + ")") + ";");
        r.ok = o;
      }
      r.maxabs = type_maxabs(to);
      return r;
    }
    if ((is_intlike(from) || from.id == TypeId::Float) && to.id == TypeId::Double) {
      r.v = "(double)" + c.v;
      return r;
    }
    if (is_intlike(from) && to.id == TypeId::Float) {
      r.v = "(float)" + c.v;
      return r;
    }
    if (is_intlike(from) && to.id == TypeId::Decimal) {
      // cast_int_to_decimal128 (conversion_funcs/numeric.rs): v * 10^s, overflow → NULL (legacy) / error (ANSI)
      u128 f = pow10_u128(to.scale);
      Val m;
      m.t = to;
      m.ok = c.ok;
      m.maxabs = sat_mul(c.maxabs, f);
      m.rep = rep_for_bound(m.maxabs);
      m.v = m.rep == Rep::I64 ? "((i64)" + c.v + " * " + lit_i64((int64_t)f) + ")" : "((i128)" + c.v + " * " + lit_i128((i128)f) + ")";
      ErrSite site = out_of_range_site(to);      // cast_int_to_decimal128 (numeric.rs:755-765): the INPUT integer, v.to_string()
      site.value = ErrSite::Int64Plain;
      const std::string in_value = "(i64)" + c.v;
      return bound_check(m, to.precision, e.eval_mode == EvalMode::Ansi, 3, &site, &in_value);
    }
    if (from.id == TypeId::Decimal && to.id == TypeId::Decimal) {
      // Deferred: CheckOverflow(Cast(dec→dec)) fuses into DecimalRescaleCheckOverflow (planner.rs:615-633).
      Val x = rescale(c, from.scale, to.precision, to.scale, e.eval_mode == EvalMode::Ansi);
      x.is_cast_dec = true;
      x.cast_child_v = c.v;
      x.cast_child_ok = c.ok;
      x.cast_child_t = c.t;
      x.cast_child_rep = c.rep;
      x.cast_child_max = c.maxabs;
      return x;
    }
    if (from.id == TypeId::Date && to.id == TypeId::Date) return c;
    if (from.id == TypeId::Date && to.id == TypeId::Int32) {      // cast.rs:273-276: a Date32 is its days since the epoch — the same 32 bits
      r.v = "(i32)" + c.v;      // (Spark's own date → int is NULL and the JVM side plans it as a literal, CometCast.scala:126-140: this is the native cast of the reference's internals)
      r.maxabs = type_maxabs(to);
      return r;
    }
    {
      // Temporal casts (conversion_funcs/temporal.rs:37-78, cast.rs:395-415, utils.rs:62-87,269-297): the time zone is the Cast's (Expr::func)
      const bool from_ts = from.id == TypeId::Timestamp, from_ntz = from.id == TypeId::TimestampNtz;
      const bool to_ts = to.id == TypeId::Timestamp, to_ntz = to.id == TypeId::TimestampNtz;
      if ((from_ts || from_ntz) && to.id == TypeId::Date) {
        c = named(c);
        const std::string local = from_ts ? local_of(e.func, c) : c.v;
        r.ok = c.ok;
        r.v = "(i32)comet::tz_floor_div(" + local + ", 86400000000ll)";
        r.maxabs = type_maxabs(to);
        return r;
      }
      if (from.id == TypeId::Date && (to_ts || to_ntz)) {
        c = named(c);
        // (unsigned product: defined for every Date32 — beyond ±106 751 991 days the reference's `d as i64 * 86_400 * 1_000_000` wraps in a release
        // build, temporal.rs:50, and its zoned arm panics inside chrono; no kernel may hold signed-overflow UB either way)
        const std::string local = "((i64)((u64)(i64)" + c.v + " * 86400000000ull))";
        r.v = to_ts ? utc_of(e.func, local, c.ok) : local;
        r.maxabs = type_maxabs(to);
        return r;
      }
      if ((from_ts || from_ntz) && to.id == TypeId::Int64) {        // spark_cast_postprocess: floor(µs / 10^6)
        r.v = "comet::tz_floor_div(" + c.v + ", 1000000ll)";
        r.maxabs = type_maxabs(to);
        return r;
      }
      if (is_intlike(from) && (to_ts || to_ntz)) {                  // cast_int_to_timestamp (numeric.rs:252-267): seconds, saturating
        c = named(c);
        std::string v = newvar("i64");
        stmt("if (__builtin_mul_overflow((i64)" + c.v + ", (i64)1000000, &" + v + ")) " + v + " = " + c.v + " < 0 ? (i64)0x8000000000000000ull : (i64)0x7fffffffffffffffll;");
        r.v = v;
        r.maxabs = type_maxabs(to);
        return r;
      }
      if (from.id == TypeId::Bool && (to_ts || to_ntz)) {           // cast_boolean_to_timestamp (boolean.rs:33-50): one microsecond
        r.v = "(i64)(" + c.v + " ? 1 : 0)";
        r.maxabs = 1;
        return r;
      }
      if (from.is_float() && (to_ts || to_ntz)) {
        // cast_float_to_timestamp (numeric.rs:87-135, 1210-1233): seconds → µs in double arithmetic; NaN / ±Infinity and a product beyond a bigint are
        // NULL (ANSI: CAST_INVALID_INPUT to TIMESTAMP, CAST_OVERFLOW to BIGINT); `micros as i64` otherwise
        c = named(c);
        const std::string d = "(double)" + c.v;
        std::string m = newvar("double"), finite = newvar("bool"), fits = newvar("bool");
        stmt(m + " = comet::fp_mul(" + d + ", 1000000.0);");
        stmt(finite + " = " + d + " == " + d + " && fabs(" + d + ") <= 1.7976931348623157e308;");
        stmt(fits + " = floor(" + m + ") <= 9223372036854775808.0 && ceil(" + m + ") >= -9223372036854775808.0;");
        if (e.eval_mode == EvalMode::Ansi) {
          ErrSite bad;      // value: val.to_string() (Rust's Display: NaN, inf, -inf), from DOUBLE whatever the source width
          bad.error_type = "CastInvalidValue";
          bad.error_class = "CAST_INVALID_INPUT";
          bad.from_type = "DOUBLE";
          bad.to_type = "TIMESTAMP";
          bad.value = ErrSite::F64Display;
          raise_value(and_ok(c.ok, "!" + finite), 9, bad, "(u64)__double_as_longlong(" + d + ")");
          ErrSite big;      // value: "{:e}" of the MICROSECONDS, upper-cased, + "D" — "Infinity" / "-Infinity" when the product left the doubles
          big.error_type = "CastOverFlow";
          big.error_class = "CAST_OVERFLOW";
          big.from_type = "DOUBLE";
          big.to_type = "BIGINT";
          big.value = ErrSite::F64Micros;
          raise_value(and_ok(c.ok, "(" + finite + " && !" + fits + ")"), 2, big, "(u64)__double_as_longlong(" + m + ")");
        }
        std::string o = newvar("bool");
        stmt(o + " = " + and_ok(c.ok, "(" + finite + " && " + fits + ")") + ";");
        r.ok = o;
        r.v = "comet::f64_to_i64_sat(" + m + ")";
        r.maxabs = type_maxabs(to);
        return r;
      }
      if (from.id == TypeId::Decimal && (to_ts || to_ntz)) {
        // cast_decimal_to_timestamp (numeric.rs:1184-1208): value · 10^6 / 10^scale in 256 bits, truncated toward zero, then `as_i128() as i64` — the low
        // 64 bits in every mode.  For scale ≥ 6 that is one division; below, the wrapping 128-bit product has the same low bits as the 256-bit one
        c = named(c);
        if (from.scale >= 6) r.v = "(i64)(" + as128(c) + " / " + lit_i128((i128)pow10_u128(from.scale - 6)) + ")";
        else r.v = "(i64)(u64)((u128)" + as128(c) + " * " + lit_u128(pow10_u128(6 - from.scale)) + ")";
        r.ok = c.ok;
        r.maxabs = type_maxabs(to);
        return r;
      }
      if (from_ts && to_ntz) {
        c = named(c);
        r.ok = c.ok;
        r.v = local_of(e.func, c);
        r.maxabs = type_maxabs(to);
        return r;
      }
      if (from_ntz && to_ts) {
        c = named(c);
        r.v = utc_of(e.func, c.v, c.ok);
        r.maxabs = type_maxabs(to);
        return r;
      }
    }
    if (from.is_float() && is_intlike(to)) {
      // conversion_funcs/numeric.rs:311-425 — Int8/Int16: (value as i32) as i8/i16; Int32/Int64: value as i32/i64 (saturating, NaN → 0).
      // ANSI: NaN or |value| as dest == dest::MAX (i32::MAX for the narrow types, then try_from) → CAST_OVERFLOW
      c = named(c);
      const std::string d = "(double)" + c.v;
      ErrSite site;      // cast_float_to_int*: format!("{:e}D" / "{:e}", value) with e → E (numeric.rs:335-349, 1028-1118)
      site.error_type = "CastOverFlow";
      site.error_class = "CAST_OVERFLOW";
      site.from_type = spark_type_name(from);
      site.to_type = spark_type_name(to);
      site.value = from.id == TypeId::Double ? ErrSite::F64 : ErrSite::F32;
      const std::string bits = from.id == TypeId::Double ? "(u64)__double_as_longlong(" + d + ")" : "(u64)(u32)__float_as_int((float)" + c.v + ")";
      if (to.id == TypeId::Int64) {
        r.v = "comet::f64_to_i64_sat(" + d + ")";
        if (e.eval_mode == EvalMode::Ansi) raise_value(and_ok(c.ok, "(" + d + " != " + d + " || comet::f64_to_i64_sat(fabs(" + d + ")) == (i64)0x7fffffffffffffffll)"), 2, site, bits);
      } else {
        const std::string i32v = "comet::f64_to_i32_sat(" + d + ")";
        if (to.id == TypeId::Int32) r.v = i32v;
        else r.v = std::string("(i32)(") + (to.id == TypeId::Int8 ? "i8" : "i16") + ")" + i32v;
        if (e.eval_mode == EvalMode::Ansi) {
          std::string ovf = "(" + d + " != " + d + " || comet::f64_to_i32_sat(fabs(" + d + ")) == (i32)0x7fffffff";
          if (to.id != TypeId::Int32) ovf += " || (i32)(" + std::string(to.id == TypeId::Int8 ? "i8" : "i16") + ")" + i32v + " != " + i32v;
          raise_value(and_ok(c.ok, ovf + ")"), 2, site, bits);
        }
      }
      if (e.eval_mode == EvalMode::Try) {
        // arrow's safe cast (cast.rs:311-326 `if eval_mode != Try` → :401-407): the value truncated toward zero, NULL when that does not fit the
        // target type — MIN − 1 < d < MAX + 1, both bounds exact doubles; a NaN fails both comparisons
        const double lo = to.id == TypeId::Int64 ? -9223372036854775808.0 : to.id == TypeId::Int32 ? -2147483649.0 : to.id == TypeId::Int16 ? -32769.0 : -129.0;
        const double hi = to.id == TypeId::Int64 ? 9223372036854775808.0 : to.id == TypeId::Int32 ? 2147483648.0 : to.id == TypeId::Int16 ? 32768.0 : 128.0;
        char lob[40], hib[40];
        snprintf(lob, sizeof lob, "%.1f", lo);
        snprintf(hib, sizeof hib, "%.1f", hi);
        std::string o = newvar("bool");
        stmt(o + " = " + and_ok(c.ok, "(" + d + (to.id == TypeId::Int64 ? " >= " : " > ") + lob + " && " + d + " < " + hib + ")") + ";");
        r.ok = o;
        r.v = to.id == TypeId::Int64 ? "comet::f64_to_i64_sat(" + d + ")" : "comet::f64_to_i32_sat(" + d + ")";      // (in range: plain truncation)
      }
      r.maxabs = type_maxabs(to);
      return r;
    }
    if (from.id == TypeId::Decimal && is_intlike(to)) {
      // numeric.rs:426-560 — truncate toward zero by 10^scale, then `as` (two's-complement truncation; narrow types go through i32).
      // ANSI: |truncated| > dest::MAX (i32::MAX, then try_from, for the narrow types) → CAST_OVERFLOW
      c = named(c);
      std::string t = newvar("i128");
      stmt(t + " = " + as128(c) + " / " + lit_i128((i128)pow10_u128(from.scale)) + ";");
      const char* nt = to.id == TypeId::Int64 ? "i64" : "i32";
      std::string v = std::string("(") + nt + ")" + t;
      if (to.id == TypeId::Int8 || to.id == TypeId::Int16) v = std::string("(i32)(") + (to.id == TypeId::Int8 ? "i8" : "i16") + ")" + v;
      r.v = v;
      if (e.eval_mode == EvalMode::Ansi) {
        const std::string mx = to.id == TypeId::Int64 ? "(u128)0x7fffffffffffffffull" : "(u128)0x7fffffffu";
        std::string ovf = "(comet::uabs128(" + t + ") > " + mx;
        if (to.id == TypeId::Int8 || to.id == TypeId::Int16) ovf += " || " + v + " != (i32)" + t;
        ErrSite site;      // cast_decimal_to_int*: "{}BD" of format_decimal_str(value, p, s), from "DECIMAL(p,s)" (numeric.rs:440-560)
        site.error_type = "CastOverFlow";
        site.error_class = "CAST_OVERFLOW";
        site.from_type = spark_type_name(from);
        site.to_type = spark_type_name(to);
        site.value = ErrSite::DecimalBD;
        site.precision = from.precision;
        site.scale = from.scale;
        Val cv = c;
        raise_value128(and_ok(c.ok, ovf + ")"), 2, site, cv);
      } else if (e.eval_mode == EvalMode::Try) {
        // arrow's safe cast (cast.rs:319-326 `if eval_mode != Try` → :401-407): the truncated quotient, NULL when it does not fit the target type
        const char* lo = to.id == TypeId::Int64 ? "-(i128)0x7fffffffffffffffll - 1" : to.id == TypeId::Int32 ? "-(i128)2147483648ll" : to.id == TypeId::Int16 ? "-(i128)32768" : "-(i128)128";
        const char* hi = to.id == TypeId::Int64 ? "(i128)0x7fffffffffffffffll" : to.id == TypeId::Int32 ? "(i128)2147483647" : to.id == TypeId::Int16 ? "(i128)32767" : "(i128)127";
        std::string o = newvar("bool");
        stmt(o + " = " + and_ok(c.ok, "(" + t + " >= " + lo + " && " + t + " <= " + hi + ")") + ";");
        r.ok = o;
        r.v = std::string("(") + nt + ")" + t;
      }
      r.maxabs = type_maxabs(to);
      return r;
    }
    if (from.is_float() && to.id == TypeId::Decimal) {
      // cast_floating_point_to_decimal128 (numeric.rs:884-990): BigDecimal(Double.toString(d)).setScale(scale, HALF_UP) — the SHORTEST digits (Ryu,
      // comet_ryu.hpp) are rounded; NaN / infinity are NULL in every mode, a result beyond the precision is NULL (ANSI: NUMERIC_VALUE_OUT_OF_RANGE)
      g_uses_ryu = true;
      c = named(c);
      std::string out = newvar("i128"), rc = newvar("int");
      stmt(out + " = 0; " + rc + " = comet::f64_bits_to_decimal((u64)__double_as_longlong((double)" + c.v + "), " + std::to_string(to.precision) + ", " + std::to_string(to.scale) + ", " + out + ");");
      if (e.eval_mode == EvalMode::Ansi) {
        ErrSite site = out_of_range_site(to);      // numeric.rs:938-948: input_value.to_string() — Rust's Display of the f64
        site.value = ErrSite::F64Display;
        raise_value(and_ok(c.ok, "(" + rc + " == 3)"), 3, site, "(u64)__double_as_longlong((double)" + c.v + ")");
      }
      std::string o = newvar("bool");
      stmt(o + " = " + and_ok(c.ok, "(" + rc + " == 0)") + ";");
      r.ok = o;
      r.v = r.rep == Rep::I64 ? "(i64)" + out : out;
      r.maxabs = type_maxabs(to);
      return r;
    }
    if (from.id == TypeId::Decimal && to.is_float()) {
      // arrow cast (the reference defers to DataFusion here): (value as f64) / 10^scale; Float32 narrows the double result
      const std::string d = "((double)" + as128(c) + " / " + lit_f64(std::pow(10.0, from.scale)) + ")";
      r.v = to.id == TypeId::Double ? d : "(float)" + d;
      return r;
    }
    if (from.id == TypeId::Double && to.id == TypeId::Float) {
      r.v = "(float)" + c.v;
      return r;
    }
    if (from.id == TypeId::Bool && (is_intlike(to) || to.is_float())) {
      r.v = std::string("(") + rep_ctype(r.rep) + ")(" + c.v + " ? 1 : 0)";
      r.maxabs = 1;
      return r;
    }
    if ((is_intlike(from) || from.is_float() || from.id == TypeId::Decimal) && to.id == TypeId::Bool) {      // decimals: spark_cast_decimal_to_boolean (numeric.rs:853-864)
      r.v = "(" + c.v + " != 0)";
      return r;
    }
    throw CometError("Cast from " + from.str() + " to " + to.str() + " is not supported in the GPU pipeline yet");
  }

  // Cast of a Utf8 COLUMN to boolean / integers / decimal / date (conversion_funcs/string.rs:260-312, 853-1115, 314-758, 1896-2046): parsed from
  // the column's bytes in place (device/comet_device.hpp "String casts"; the same text is checked on the host against the reference's vectors).
  // Invalid input is NULL in LEGACY / TRY and an error under ANSI.
  Val cast_from_string(const Expr& e, int idx) {
    const DType& to = e.dtype;
    const int mode = e.eval_mode == EvalMode::Legacy ? 0 : e.eval_mode == EvalMode::Ansi ? 1 : 2;
    std::string call, outtype;
    int err_bit = 9;          // CAST_INVALID_INPUT
    Val r;
    r.t = to;
    r.rep = rep_for_type(to);
    switch (to.id) {
      case TypeId::Bool: outtype = "bool"; call = "comet::str_to_bool(sp, sn, @)"; break;
      case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: case TypeId::Int64:
        outtype = "i64";
        call = "comet::str_to_int(sp, sn, " + std::to_string(mode) + ", " + std::to_string(type_width(to) * 8) + ", @)";
        r.maxabs = type_maxabs(to);
        break;
      case TypeId::Decimal:
        outtype = "i128";
        call = "comet::str_to_decimal(sp, sn, " + std::to_string(to.precision) + ", " + std::to_string(to.scale) + ", @)";
        r.maxabs = type_maxabs(to);
        break;
      case TypeId::Date: outtype = "i32"; call = "comet::str_to_date(sp, sn, @)"; r.maxabs = type_maxabs(to); err_bit = 10; break;
      case TypeId::Timestamp: case TypeId::TimestampNtz: {
        // cast_string_to_timestamp / _ntz (string.rs:798-852 → timestamp_parser, comet_strts.hpp): the reference's fourteen shapes and zone suffixes in the
        // session zone.  Refused at run time (the task fails, it is not answered differently): a NAMED zone inside a value (the kernel holds the
        // session zone's table), a time-only value ("T12:34": the reference gives it TODAY's date), an instant behind the zone table's end.
        g_uses_strts = true;
        outtype = "i64";
        if (to.id == TypeId::Timestamp) {
          long long secs = 0;
          std::string zt;
          if (fixed_zone_offset(e.func, secs)) {
            zt = "ztf" + std::to_string(nvar++);
            decls += "    static const i64 " + zt + "[] = {0, " + std::to_string(secs) + ", (i64)0x7fffffffffffffffll};\n";
          } else zt = zone_table(e.func);
          call = "comet::str_to_timestamp((const u8*)sp, sn, " + zt + ", " + (e.is_spark4_plus ? "true" : "false") + ", (i64)0x8000000000000000ull, @)";
        } else {
          call = "comet::str_to_timestamp_ntz((const u8*)sp, sn, @)";
        }
        r.maxabs = type_maxabs(to);
        err_bit = to.id == TypeId::Timestamp ? 13 : 14;
        break;
      }
      case TypeId::Float: case TypeId::Double:
        // cast_string_to_float (string.rs:177-258): a correctly rounded conversion (comet_strtod.hpp), straight to the target's width
        g_uses_strtod = true;
        outtype = "u64";
        call = std::string("comet::str_to_float_bits((const u8*)sp, sn, ") + (to.id == TypeId::Float ? "true" : "false") + ", @)";
        break;
      default: throw CometError("Cast from string to " + to.str() + " is not supported in the GPU pipeline yet");
    }
    Val valid = str_col_validity(idx);
    auto loc = locate(idx);
    std::string out = newvar(outtype.c_str()), rc = newvar("int");
    call.replace(call.find('@'), 1, out);
    stmt(out + " = 0; " + rc + " = 2; " + (valid.ok.empty() ? "" : "if (" + valid.ok + ") ") + "{ i32 sn; comet::strp sp = comet::utf8_bytes(prm.in[" + std::to_string(loc.first) + "], " +
         loc.second + ", sn); " + rc + " = " + call + "; }");
    if (e.eval_mode == EvalMode::Ansi) {
      // invalid_value(raw value, "STRING", type name) (string.rs:219, 359, 995-1117) / InvalidInputInCastToDatetime with the raw value (:39-70): the
      // string's bytes go into the error block
      auto raise_with_string = [&](const std::string& cond, int code, const ErrSite& site) {
        uses_err = true;
        const std::string eb = "prm.out[" + std::to_string(kOutErr) + "]";
        stmt("if (" + cond + ") { atomicOr((unsigned int*)" + eb + ", " + std::to_string(1u << code) + "u); i32 en_; comet::strp ep_ = comet::utf8_bytes(prm.in[" +
             std::to_string(loc.first) + "], " + loc.second + ", en_); comet::err_detail_str(" + eb + ", " + std::to_string(site_id(site)) + "u, (const u8*)ep_, en_); }");
      };
      ErrSite site;
      const bool datetime = err_bit == 10 || err_bit == 13 || err_bit == 14;
      site.error_type = datetime ? "InvalidInputInCastToDatetime" : "CastInvalidValue";
      site.error_class = "CAST_INVALID_INPUT";
      site.from_type = "STRING";
      site.to_type = spark_type_name(to);
      site.value = ErrSite::Str;
      raise_with_string("(" + rc + " == 1)", err_bit, site);
      if (to.id == TypeId::Decimal) {
        ErrSite o = out_of_range_site(to);      // string.rs:645-652: the (trimmed) string that parsed to a decimal beyond the precision
        o.value = ErrSite::Str;
        raise_with_string("(" + rc + " == 3)", 3, o);
      }
    }
    if (to.id == TypeId::Timestamp || to.id == TypeId::TimestampNtz) {
      raise_if("(" + rc + " == 4 || " + rc + " == 6)", 12);
      raise_if("(" + rc + " == 5)", 11);
    }
    std::string o = newvar("bool");
    stmt(o + " = " + rc + " == 0;");
    r.ok = o;
    switch (to.id) {
      case TypeId::Bool: r.v = out; break;
      case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: r.v = "(i32)" + out; break;
      case TypeId::Int64: r.v = out; break;
      case TypeId::Decimal: r.v = r.rep == Rep::I64 ? "(i64)" + out : out; break;
      case TypeId::Double: r.v = "__longlong_as_double((i64)" + out + ")"; break;
      case TypeId::Float: r.v = "__int_as_float((i32)(u32)" + out + ")"; break;
      default: r.v = out; break;
    }
    return r;
  }

  // rescale_and_check (decimal_rescale_check.rs:108-150)
  Val rescale(Val c, int s_in, int p_out, int s_out, bool fail_on_error) {
    Val r;
    r.t = DType::decimal(p_out, s_out);
    r.ok = c.ok;
    const int delta = s_out - s_in;
    const u128 bound = pow10_u128(p_out) - 1;
    // (the reference fails a rescale with a plain compute error, decimal_rescale_check.rs:124,145; here it is the CheckOverflow it fuses — the
    // unscaled value that does not fit and the target's (p, s))
    const ErrSite site = out_of_range_site(r.t);
    if (delta == 0) {
      r.v = c.v;
      r.rep = c.rep;
      r.maxabs = c.maxabs;
      return bound_check(r, p_out, fail_on_error, 3, &site);
    }
    if (std::abs(delta) > 38) throw CometError("DecimalRescaleCheckOverflow: scale delta " + std::to_string(delta) + " exceeds maximum supported range");
    u128 f = pow10_u128(std::abs(delta));
    if (delta > 0) {
      r.maxabs = sat_mul(c.maxabs, f);
      if (r.maxabs != kUnbounded && r.maxabs < ((u128)1 << 126)) {
        r.rep = rep_for_bound(r.maxabs);
        r.v = r.rep == Rep::I64 ? "(" + as64(c) + " * " + lit_i64((int64_t)f) + ")" : "(" + as128(c) + " * " + lit_i128((i128)f) + ")";
        return bound_check(r, p_out, fail_on_error, 3, &site);
      }
      c = named(c);
      std::string out = newvar("i128"), fit = newvar("bool");
      stmt(fit + " = comet::dec_rescale_up(" + as128(c) + ", " + lit_i128((i128)f) + ", " + lit_u128(bound) + ", " + out + ");");
      r.rep = Rep::I128;
      r.v = out;
      r.maxabs = bound;
      if (fail_on_error) raise_value128(and_ok(c.ok, "!" + fit), 3, site, c);      // (the value before the multiplication that left 128 bits or the precision)
      else {
        std::string o = newvar("bool");
        stmt(o + " = " + and_ok(c.ok, fit) + ";");
        r.ok = o;
      }
      return r;
    }
    // scale down, HALF_UP
    c = named(c);
    std::string out = newvar("i128"), fit = newvar("bool");
    stmt(fit + " = comet::dec_rescale_down(" + as128(c) + ", " + lit_i128((i128)f) + ", " + lit_u128(bound) + ", " + out + ");");
    r.rep = Rep::I128;
    r.v = out;
    u128 m = c.maxabs == kUnbounded ? kUnbounded : c.maxabs / f + 1;
    r.maxabs = m;
    if (m <= bound) return r;  // cannot overflow: `fit` is always true
    r.maxabs = bound;
    if (fail_on_error) { Val ov = r; ov.v = out; ov.rep = Rep::I128; raise_value128(and_ok(c.ok, "!" + fit), 3, site, ov); }
    else {
      std::string o = newvar("bool");
      stmt(o + " = " + and_ok(c.ok, fit) + ";");
      r.ok = o;
    }
    return r;
  }

  // Output-only string functions of a Utf8 column (ScalarFunc substring / trim family / rpad / lpad / read_side_padding with literal
  // arguments): the row's result is described as a comet::strview — which bytes of the source value, how many pad characters — and the
  // column is assembled by the executor, so the result may have any length (as an operand of another expression these functions take
  // the packed ≤ 15-byte path of scalar_func, or are rejected).  Returns false when `e` is not of that shape.
  bool string_view(const Expr& e, Val& out, OutCol& oc) {
    if (e.kind != ExprKind::ScalarFunc || e.children.empty() || !is_str_col(e.children[0])) return false;
    const std::string& f = e.func;
    auto int_lit = [](const ExprP& x, long long& v) {
      if (x->kind != ExprKind::Literal || x->lit_null || !x->dtype.is_integer()) return false;
      v = x->lit_i64;
      return true;
    };
    auto clamp32 = [](long long v) { return std::max<long long>(std::min<long long>(v, 0x7fffffffLL), -0x7fffffffLL); };
    const int idx = e.children[0]->bound_index;
    std::string call;
    if (f == "substring" || f == "substr") {
      long long pos = 0, len = 0x7fffffffLL;
      if (e.children.size() < 2 || e.children.size() > 3 || !int_lit(e.children[1], pos) || (e.children.size() == 3 && !int_lit(e.children[2], len))) return false;
      call = "utf8_view_substr(@, " + std::to_string(clamp32(pos)) + ", " + std::to_string(clamp32(len)) + ")";
    } else if (f == "upper" || f == "lower") {
      // DataFusion's upper / lower = Rust's str::to_uppercase / to_lowercase (the reference's Upper / Lower under spark.comet.caseConversion.enabled):
      // the whole value as a view, mapped by the executor's case kernels
      if (e.children.size() != 1) return false;
      oc.case_mode = f == "lower" ? 1 : 2;
      call = "utf8_view_substr(@, 1, 2147483647)";
    } else if (f == "trim" || f == "btrim" || f == "ltrim" || f == "rtrim") {
      if (e.children.size() != 1) return false;      // a trim string is a different function
      call = std::string("utf8_view_trim(@, ") + (f == "ltrim" ? "1" : f == "rtrim" ? "2" : "3") + ")";
    } else if (f == "rpad" || f == "lpad" || f == "read_side_padding") {
      long long n = 0;
      if (e.children.size() < 2 || e.children.size() > 3 || !int_lit(e.children[1], n)) return false;
      std::string pat = " ";
      if (e.children.size() == 3) {
        if (!is_str_lit(e.children[2])) return false;
        pat = e.children[2]->lit_bytes;
      }
      size_t chars = 0;
      for (unsigned char ch : pat) chars += (ch & 0xC0) != 0x80;
      if (pat.size() > 64 || chars > 32) throw CometError(f + ": pad strings of more than 32 characters are not supported");
      oc.pad_pattern = pat;
      oc.pad_left = f == "lpad";
      call = "utf8_view_pad(@, " + std::to_string(clamp32(n)) + ", " + (f == "read_side_padding" ? "false" : "true") + ")";
    } else if (f == "regexp_extract") {
      // spark_regexp_extract (string_funcs/regexp_extract.rs:38-108): the bytes group `idx` of the pattern's leftmost match spans, the empty
      // string without a match or with the group unset; pattern and idx are literals (strings.scala:472-473), a NULL one makes every row NULL
      // (regexp_extract_common.rs:58-81).  The pattern becomes a program of comet_regex_vm.hpp's matcher (regex.cpp), a constant of the kernel.
      if (e.children.size() < 2 || e.children.size() > 3) throw CometError("regexp_extract expects 2 or 3 arguments (subject, pattern, [idx]), got " + std::to_string(e.children.size()));
      const ExprP& pat = e.children[1];
      if (pat->kind != ExprKind::Literal || (pat->dtype.id != TypeId::String && !pat->lit_null)) throw CometError("regexp_extract pattern must be a scalar string");
      long long group = 1;
      bool all_null = pat->lit_null;
      if (e.children.size() == 3) {
        const ExprP& gi = e.children[2];
        if (gi->kind != ExprKind::Literal || (gi->dtype.id != TypeId::Int32 && !gi->lit_null)) throw CometError("regexp_extract idx must be an Int32 scalar");
        if (gi->lit_null) all_null = true;
        else group = gi->lit_i64;
      }
      if (all_null) {
        Val valid = str_col_validity(idx);
        (void)valid;
        out.t = DType::of(TypeId::String);
        out.rep = Rep::I64;
        std::string v = newvar("comet::strview");
        stmt(v + " = comet::strview{(u32)" + locate(idx).second + ", 0u, 0u, 0u};");
        out.v = v;
        out.ok = "false";
        oc.view_src = idx;
        return true;
      }
      const RegexProg prog = compile_regex_captures(pat->lit_bytes, (int)std::max<long long>(std::min<long long>(group, 1 << 20), -(1 << 20)), "regexp_extract");
      const std::string name = "rxp" + std::to_string(regex_progs++);
      std::string d = "    static const u32 " + name + "[] = {";
      for (size_t k = 0; k < prog.words.size(); k++) d += (k ? "," : "") + std::to_string(prog.words[k]) + "u";
      decls += d + "};\n";
      g_uses_regex_vm = true;
      call = "utf8_view_regex(@, " + name + ")";
    } else {
      return false;
    }
    Val valid = str_col_validity(idx);
    auto loc = locate(idx);
    const std::string args = "prm.in[" + std::to_string(loc.first) + "], " + loc.second;
    call.replace(call.find('@'), 1, args);
    std::string v = newvar("comet::strview");
    stmt(v + " = comet::" + call + ";");
    out.t = DType::of(TypeId::String);
    out.rep = Rep::I64;      // opaque to everything but the store below
    out.v = v;
    out.ok = valid.ok;
    oc.view_src = idx;
    return true;
  }

  // ---- time zones (csrc/tz.cpp): a region zone's table is a constant array of the kernel; fixed offsets are constants of the expression ----
  int regex_progs = 0;      // regexp_extract programs declared so far (constants of the kernel, like the zone tables)
  std::map<std::string, std::string> zone_vars;
  std::string zone_table(const std::string& tz) {
    auto it = zone_vars.find(tz);
    if (it != zone_vars.end()) return it->second;
    const std::vector<int64_t> f = load_zone(tz)->flat();
    const std::string name = "zt" + std::to_string(zone_vars.size());
    std::string d = "    static const i64 " + name + "[] = {";
    for (size_t i = 0; i < f.size(); i++) d += (i ? "," : "") + (f[i] >= 0 && f[i] < 100000 ? std::to_string(f[i]) : lit_i64(f[i]));
    decls += d + "};\n";
    zone_vars[tz] = name;
    return name;
  }
  // UTC µs → the zone's wall clock as µs.  An instant behind a rule zone's table (the year 2400) raises error bit 11.
  std::string local_of(const std::string& tz, Val& c) {
    long long secs = 0;
    if (fixed_zone_offset(tz, secs)) return secs ? "(" + c.v + " + " + lit_i64(secs * 1000000) + ")" : c.v;
    c = named(c);
    const std::string zt = zone_table(tz);
    std::string v = newvar("i64"), b = newvar("bool");
    stmt(b + " = false; " + v + " = comet::tz_utc_to_local_us(" + zt + ", " + c.v + ", " + b + ");");
    raise_if(and_ok(c.ok, b), 11);
    return v;
  }
  // the zone's wall clock µs → UTC µs (resolve_local_datetime, utils.rs:184-205)
  std::string utc_of(const std::string& tz, const std::string& local, const std::string& ok) {
    long long secs = 0;
    if (fixed_zone_offset(tz, secs)) return secs ? "(" + local + " - " + lit_i64(secs * 1000000) + ")" : local;
    const std::string zt = zone_table(tz);
    std::string v = newvar("i64"), b = newvar("bool");
    stmt(b + " = false; " + v + " = comet::tz_local_to_utc_us(" + zt + ", " + local + ", " + b + ");");
    raise_if(and_ok(ok, b), 11);
    return v;
  }

  // Output-only concat(a, b, …) over Utf8 COLUMNS of the source table and literals (Spark's Concat → datafusion-spark's SparkConcat,
  // jni_api.rs:70: the bytes one after the other, NULL as soon as one argument is NULL).  At most eight parts; false when `e` is not that.
  bool string_concat(const Expr& e, Val& out, OutCol& oc) {
    if (e.kind != ExprKind::ScalarFunc || e.func != "concat" || e.children.empty()) return false;
    if (e.children.size() > 8) throw CometError("concat of more than eight arguments is not supported by the MI355X native engine yet");
    std::string ok;
    for (const ExprP& c : e.children) {
      if (is_str_col(c) && in_types[(size_t)c->bound_index].id == TypeId::String) {
        Val valid = str_col_validity(c->bound_index);
        ok = and_ok(ok, valid.ok);
        oc.concat_cols.push_back(c->bound_index);
        oc.concat_lits.emplace_back();
      } else if (is_str_lit(c)) {
        oc.concat_cols.push_back(-1);
        oc.concat_lits.push_back(c->lit_bytes);
      } else {
        throw CometError("concat is supported over Utf8 columns and literals");
      }
    }
    out.t = DType::of(TypeId::String);
    out.rep = Rep::I64;        // the source row: opaque to everything but the store
    out.v = "idx[r]";
    if (!ok.empty()) {
      std::string o = newvar("bool");
      stmt(o + " = " + ok + ";");
      out.ok = o;
    }
    return true;
  }

  // Output-only Cast(value AS STRING) (conversion_funcs/cast.rs:423-449: the reference defers integers, booleans, dates and timestamps to
  // arrow-cast's formatting and writes LEGACY decimals like java.math.BigDecimal.toString, numeric.rs:593-704).  Returns false when `e` is
  // not of that shape; floats (Java's shortest-digits algorithm) are not there yet.
  bool string_format(const Expr& e, Val& out, OutCol& oc) {
    if (e.kind != ExprKind::Cast || e.dtype.id != TypeId::String || e.children.size() != 1) return false;
    const ExprP& c = e.children[0];
    if (is_str_col(c)) return false;
    // the child's type decides; only expressions whose type is known without generating them twice: generate, then look
    Val v = gen(c);
    switch (v.t.id) {
      case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: case TypeId::Int64: oc.fmt_kind = OutCol::FmtInt; break;
      case TypeId::Bool: oc.fmt_kind = OutCol::FmtBool; break;
      case TypeId::Decimal:
        oc.fmt_kind = e.eval_mode == EvalMode::Legacy ? OutCol::FmtDecimalJava : OutCol::FmtDecimal;
        oc.fmt_arg = v.t.scale;
        break;
      case TypeId::Date: oc.fmt_kind = OutCol::FmtDate; break;
      // spark_cast_float64_to_utf8 / float32 (numeric.rs:137-221): the shortest digits, plain between 10⁻³ and 10⁷, else d.dddE±n; the value's BITS travel
      case TypeId::Double: oc.fmt_kind = OutCol::FmtFloat64; v = named(v); v.v = "__double_as_longlong(" + v.v + ")"; v.rep = Rep::I64; break;
      case TypeId::Float: oc.fmt_kind = OutCol::FmtFloat32; v = named(v); v.v = "(i64)(u32)__float_as_int(" + v.v + ")"; v.rep = Rep::I64; break;
      case TypeId::Timestamp: case TypeId::TimestampNtz: {
        // (pre_timestamp_cast, utils.rs:299-330: the instant becomes the session zone's wall clock, which is then written out)
        if (v.t.id == TypeId::Timestamp) { v = named(v); v.v = local_of(e.func, v); }
        oc.fmt_kind = OutCol::FmtTimestamp;
        oc.fmt_arg = 0;
        break;
      }
      case TypeId::String: return false;      // (generated once more by the caller: the statements above are dead code the compiler drops)
      default: throw CometError("Cast from " + v.t.str() + " to string is not supported in the GPU pipeline yet");
    }
    v = named(v);
    out = v;
    if (v.rep == Rep::B) out.v = "(" + v.v + " ? 1 : 0)";
    return true;
  }

  // ScalarFunc (expr.proto:466-471 → create_comet_physical_fun, comet_scalar_funcs.rs): the subset whose results are defined
  // exactly (integer / IEEE operations): ceil, floor, abs, sqrt, signum, isnan, datepart.  Everything else is rejected by name.
  // ---- lists (array_funcs/{list_extract,size}.rs, datafusion-spark's array_contains): the chain's source table carries the element column of every list of
  // flat elements as a column of its own (exec.cpp extend_struct_fields; a derived split / regexp_extract_all list likewise), so a row's elements are
  // element-column rows offs[i] … offs[i + 1] ----
  static std::string load_of(const DType& t, const std::string& c, const std::string& row) {
    switch (t.id) {
      case TypeId::Bool: return "comet::ld_bool(" + c + ", " + row + ")";
      case TypeId::Int8: return "(i32)comet::ld<i8>(" + c + ", " + row + ")";
      case TypeId::Int16: return "(i32)comet::ld<i16>(" + c + ", " + row + ")";
      case TypeId::Int32: case TypeId::Date: return "comet::ld<i32>(" + c + ", " + row + ")";
      case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: return "comet::ld<i64>(" + c + ", " + row + ")";
      case TypeId::Float: return "comet::ld<float>(" + c + ", " + row + ")";
      case TypeId::Double: return "comet::ld<double>(" + c + ", " + row + ")";
      case TypeId::Decimal: return t.precision <= 18 ? "comet::ld_dec_lo(" + c + ", " + row + ")" : "comet::ld<i128>(" + c + ", " + row + ")";
      default: throw CometError("list elements of type " + t.str() + " are not supported in an expression by the MI355X native engine");
    }
  }
  struct ListRef { int idx = -1, elem = -1; std::string off, len, ok, ecol; DType etype; };
  // a list COLUMN's row: first element, count, validity; `need_elements`: the element column must be addressable too
  ListRef list_ref(const ExprP& x, const char* who, bool need_elements) {
    if (x->kind != ExprKind::Bound || x->bound_index < 0 || (size_t)x->bound_index >= in_types.size() || !in_types[(size_t)x->bound_index].is_listlike())
      throw CometError(std::string(who) + " is supported over a list / map COLUMN (not over a computed array) by the MI355X native engine");
    ListRef l;
    l.idx = x->bound_index;
    in_used[(size_t)l.idx] = true;
    auto loc = locate(l.idx);
    const std::string c = "prm.in[" + std::to_string(loc.first) + "]";
    l.off = newvar("i32");
    l.len = newvar("i32");
    stmt(l.off + " = comet::ld<i32>(" + c + ", " + loc.second + "); " + l.len + " = comet::ld<i32>(" + c + ", (" + loc.second + ") + 1) - " + l.off + ";");
    if (in_valid[(size_t)l.idx]) {
      l.ok = newvar("bool");
      stmt(l.ok + " = comet::ld_valid(" + c + ", " + loc.second + ");");
    }
    if (need_elements) {
      for (size_t j = 0; j < in_types.size(); j++)
        if (in_types[j].virt_parent == l.idx && in_types[j].virt_kid == 0 && in_types[(size_t)l.idx].id == TypeId::List) l.elem = (int)j;
      if (l.elem < 0) throw CometError(std::string(who) + ": the elements of " + in_types[(size_t)l.idx].str() + " are not addressable by an expression (lists of flat elements are)");
      in_used[(size_t)l.elem] = true;
      l.etype = in_types[(size_t)l.elem];
      l.ecol = "prm.in[" + std::to_string(locate(l.elem).first) + "]";
    }
    return l;
  }
  // ListExtract's element position (array_funcs/list_extract.rs:229-320): → the element-column row (i64, -1: none) and whether there is one
  void list_extract_pos(const Expr& e, ListRef& l, std::string& row, std::string& hit) {
    if (e.children.size() < 2) throw CometError("ListExtract expects a child and an ordinal");
    if (e.children.size() > 2 && !(e.children[2]->kind == ExprKind::Literal && e.children[2]->lit_null)) throw CometError("ListExtract with a default value is not supported by the MI355X native engine yet");
    l = list_ref(e.children[0], e.one_based ? "element_at" : "GetArrayItem", true);
    Val ord = named(gen(e.children[1]));
    if (ord.t.id != TypeId::Int32) throw CometError("ListExtract expects an Int32 ordinal, got " + ord.t.str());
    const std::string both = and_ok(l.ok, ord.ok).empty() ? "true" : and_ok(l.ok, ord.ok);
    const std::string o = "(i64)" + ord.v, n = "(i64)" + l.len;
    std::string pos = newvar("i64");
    if (e.one_based) {
      ErrSite z;      // element_at(arr, 0): INVALID_INDEX_OF_ZERO in every mode (list_extract.rs:234-236)
      z.error_type = "InvalidIndexOfZero";
      z.error_class = "INVALID_INDEX_OF_ZERO";
      z.value = ErrSite::NoValue;
      raise_value("(" + both + " && " + ord.v + " == 0)", 19, z, "0");
      stmt(pos + " = " + o + " > 0 ? (" + o + " <= " + n + " ? " + o + " - 1 : -1) : (-(" + o + ") <= " + n + " ? " + n + " + " + o + " : -1);");
    } else {
      stmt(pos + " = (" + o + " >= 0 && " + o + " < " + n + ") ? " + o + " : -1;");
    }
    if (e.fail_on_error) {
      ErrSite s;      // InvalidElementAtIndex / InvalidArrayIndex { index_value, array_size } (list_extract.rs:294-311; error.rs:397-414)
      s.error_type = e.one_based ? "InvalidElementAtIndex" : "InvalidArrayIndex";
      s.error_class = e.one_based ? "INVALID_ARRAY_INDEX_IN_ELEMENT_AT" : "INVALID_ARRAY_INDEX";
      s.value = ErrSite::IndexAndSize;
      raise_value("(" + both + " && " + pos + " < 0" + (e.one_based ? " && " + ord.v + " != 0" : "") + ")", 19, s, "(i64)" + ord.v, "(i64)" + l.len);
    }
    hit = newvar("bool");
    stmt(hit + " = " + both + " && " + pos + " >= 0;");
    row = newvar("i64");
    stmt(row + " = " + hit + " ? (i64)" + l.off + " + " + pos + " : (i64)0;");
  }
  Val list_extract(const Expr& e) {
    ListRef l;
    std::string row, hit;
    list_extract_pos(e, l, row, hit);
    if (l.etype.id == TypeId::String || l.etype.id == TypeId::Bytes) throw CometError("an element of a list of strings is supported as an OUTPUT column only (not as an operand) by the MI355X native engine");
    Val r;
    r.t = l.etype;
    r.t.virt_parent = r.t.virt_kid = -1;
    r.rep = rep_for_type(r.t);
    r.maxabs = type_maxabs(r.t);
    std::string ok = newvar("bool");
    stmt(ok + " = " + hit + (in_valid[(size_t)l.elem] ? " && comet::ld_valid(" + l.ecol + ", " + row + ")" : "") + ";");
    std::string v = newvar(rep_ctype(r.rep));
    stmt(v + " = " + ok + " ? " + load_of(l.etype, l.ecol, row) + " : (" + rep_ctype(r.rep) + ")0;");
    r.v = v;
    r.ok = ok;
    return r;
  }

  Val scalar_func(const Expr& e) {
    const std::string& f = e.func;
    auto arg = [&](size_t i) { return named(gen(e.children.at(i))); };
    Val r;
    if (f == "substring" || f == "substr") {
      // Spark Substring (strings.scala:209) on a Utf8 column with literal position / length: a computed string of at most 15 bytes,
      // packed in registers — usable as a group / join key and in comparisons (TPC-H Q22: substring(c_phone, 1, 2))
      auto int_lit = [](const ExprP& x, long long& out) {
        if (x->kind != ExprKind::Literal || x->lit_null) return false;
        out = x->lit_i64;
        return x->dtype.is_integer();
      };
      long long pos = 0, len = 0x7fffffffLL;
      if (e.children.size() < 2 || e.children.size() > 3 || !is_str_col(e.children[0]) || !int_lit(e.children[1], pos) || (e.children.size() == 3 && !int_lit(e.children[2], len)))
        throw CometError("substring is supported for a Utf8 column with literal position and length");
      pos = std::max<long long>(std::min<long long>(pos, 0x7fffffffLL), -0x7fffffffLL);
      len = std::max<long long>(std::min<long long>(len, 0x7fffffffLL), -0x7fffffffLL);
      const int idx = e.children[0]->bound_index;
      Val valid = str_col_validity(idx);
      auto loc = locate(idx);
      uses_err = true;
      std::string v = newvar("comet::str16");
      stmt("{ bool tl_ = false; " + v + " = comet::utf8_substr16(prm.in[" + std::to_string(loc.first) + "], " + loc.second + ", " + std::to_string(pos) + ", " + std::to_string(len) +
           ", tl_); if (tl_) atomicOr((unsigned int*)prm.out[" + std::to_string(kOutErr) + "], 64u); }");
      r.t = DType::of(TypeId::String);
      r.rep = Rep::STR;
      r.v = v;
      r.ok = valid.ok;
      return r;
    }
    if (f == "starts_with" || f == "ends_with" || f == "contains") {
      // byte-wise (UTF8_BINARY) tests of a Utf8 column against a literal (strings.scala:343-360 → DataFusion starts_with / ends_with / contains)
      if (e.children.size() != 2 || !is_str_col(e.children[0]) || !is_str_lit(e.children[1]))
        throw CometError(f + " is supported for a Utf8 column and a literal");
      return str_pred_lit("utf8_" + f + "_lit", e.children[0]->bound_index, e.children[1]->lit_bytes);
    }
    if (f == "length" || f == "char_length" || f == "character_length" || f == "octet_length" || f == "bit_length") {
      if (e.children.size() != 1 || !is_str_col(e.children[0])) throw CometError(f + " is supported for a Utf8 column");
      const int idx = e.children[0]->bound_index;
      Val valid = str_col_validity(idx);
      auto loc = locate(idx);
      const bool chars = f != "octet_length" && f != "bit_length";
      r.t = DType::of(TypeId::Int32);
      r.rep = Rep::I32;
      r.ok = valid.ok;
      std::string v = newvar("i32");
      stmt(v + " = " + (valid.ok.empty() ? "" : valid.ok + " ? ") + "comet::" + (chars ? "utf8_char_length" : "utf8_octet_length") + "(prm.in[" + std::to_string(loc.first) +
           "], " + loc.second + ")" + (f == "bit_length" ? " * 8" : "") + (valid.ok.empty() ? "" : " : 0") + ";");
      r.v = v;
      r.maxabs = (u128)1 << 31;
      return r;
    }
    if (f == "coalesce") {
      // the first non-NULL argument (Spark Coalesce → DataFusion's coalesce): a chain of selects from the last argument backwards
      if (e.children.empty()) throw CometError("coalesce needs at least one argument");
      Val acc = named(gen(e.children.back()));
      for (size_t k = e.children.size() - 1; k-- > 0;) {
        Val a = named(gen(e.children[k]));
        if (a.ok.empty()) { acc = a; continue; }      // never NULL: everything after it is dead
        Val is_set;
        is_set.t = DType::of(TypeId::Bool);
        is_set.rep = Rep::B;
        is_set.v = a.ok;
        Val picked = a;
        picked.ok = "";                                // inside the branch it IS set
        acc = named(select(is_set, picked, acc));
      }
      return acc;
    }
    if (f == "ceil" || f == "floor") {
      // spark_ceil / spark_floor (math_funcs/ceil.rs:24-84): Float → Int64 (`as i64`), Int64 unchanged, Decimal(s > 0) → div_ceil by 10^s
      Val a = arg(0);
      const bool up = f == "ceil";
      r.ok = a.ok;
      if (a.rep == Rep::F64 || a.rep == Rep::F32) {
        r.t = DType::of(TypeId::Int64);
        r.rep = Rep::I64;
        r.v = "comet::f64_to_i64_sat(" + std::string(up ? "ceil" : "floor") + "((double)" + a.v + "))";
        return r;
      }
      if (a.t.id == TypeId::Int64) return a;
      if (a.t.id == TypeId::Decimal && a.t.scale > 0) {
        if (!e.has_dtype || e.dtype.id != TypeId::Decimal) throw CometError(f + ": expected a Decimal128 return type");
        r.t = e.dtype;
        r.rep = Rep::I128;
        r.v = std::string("comet::dec_div_") + (up ? "ceil" : "floor") + "(" + as128(a) + ", " + lit_i128((i128)pow10_u128(a.t.scale)) + ")";
        r.maxabs = a.maxabs == kUnbounded ? kUnbounded : a.maxabs / pow10_u128(a.t.scale) + 1;
        return r;
      }
      throw CometError("Unsupported data type " + a.t.str() + " for function " + f);
    }
    if (f == "abs") {
      // spark_abs (math_funcs/abs.rs): wrapping_abs in LEGACY mode, ARITHMETIC_OVERFLOW for MIN in ANSI mode (second argument)
      Val a = arg(0);
      bool fail = e.fail_on_error;
      if (e.children.size() == 2 && e.children[1]->kind == ExprKind::Literal && e.children[1]->dtype.id == TypeId::Bool) fail = e.children[1]->lit_bool;
      r = a;
      switch (a.rep) {
        case Rep::F64: r.v = "fabs(" + a.v + ")"; return r;
        case Rep::F32: r.v = "fabsf(" + a.v + ")"; return r;
        case Rep::I128: r.v = "(" + a.v + " < 0 ? (i128)((u128)0 - (u128)" + a.v + ") : " + a.v + ")"; return r;
        case Rep::I32: case Rep::I64: {
          if (a.t.id == TypeId::Decimal) { r.v = "(" + a.v + " < 0 ? -" + a.v + " : " + a.v + ")"; return r; }
          const std::string ct = a.t.id == TypeId::Int64 ? "i64" : a.t.id == TypeId::Int32 ? "i32" : a.t.id == TypeId::Int16 ? "i16" : "i8";
          const std::string ut = a.t.id == TypeId::Int64 ? "u64" : a.t.id == TypeId::Int32 ? "u32" : a.t.id == TypeId::Int16 ? "unsigned short" : "u8";
          const std::string mn = a.t.id == TypeId::Int64 ? "(i64)0x8000000000000000ull" : a.t.id == TypeId::Int32 ? "(i32)0x80000000" : a.t.id == TypeId::Int16 ? "-32768" : "-128";
          if (fail) {
            ErrSite site;      // abs.rs:205-255: arithmetic_overflow_error("Int8" / "Int16" / "Int32" / "Int64")
            site.error_type = "ArithmeticOverflow";
            site.error_class = "ARITHMETIC_OVERFLOW";
            site.from_type = a.t.id == TypeId::Int64 ? "Int64" : a.t.id == TypeId::Int32 ? "Int32" : a.t.id == TypeId::Int16 ? "Int16" : "Int8";
            site.value = ErrSite::NoValue;
            raise_value(and_ok(a.ok, "(" + a.v + " == " + mn + ")"), 1, site, "0");
          }
          r.v = std::string("(") + rep_ctype(a.rep) + ")(" + ct + ")(" + a.v + " < 0 ? (" + ut + ")0 - (" + ut + ")(" + ct + ")" + a.v + " : (" + ut + ")(" + ct + ")" + a.v + ")";
          return r;
        }
        default: throw CometError("Unsupported data type " + a.t.str() + " for function abs");
      }
    }
    if (f == "date_add" || f == "date_sub") {
      // Date32 ± Int8/16/32 days, wrapping like the JVM's int arithmetic (planner.rs:1059-1092 → datafusion-spark SparkDateAdd / SparkDateSub)
      Val a = arg(0), b = arg(1);
      if (a.t.id != TypeId::Date || !(b.t.id == TypeId::Int8 || b.t.id == TypeId::Int16 || b.t.id == TypeId::Int32))
        throw CometError(f + " expects (Date32, Int8 | Int16 | Int32)");
      r.t = DType::of(TypeId::Date);
      r.rep = Rep::I32;
      r.ok = and_ok(a.ok, b.ok);
      r.v = "(i32)((u32)" + a.v + (f == "date_add" ? " + " : " - ") + "(u32)" + b.v + ")";
      r.maxabs = (u128)1 << 31;
      return r;
    }
    if (f == "murmur3_hash" || f == "xxhash64") {
      const bool xx = f == "xxhash64";   // spark_xxhash64 (hash_funcs/xxhash64.rs:31-82): Int64 seed, Int64 result, XXH64 of the same value bytes
      // spark_murmur3_hash (hash_funcs/murmur3.rs:24-70) = Spark's hash(...): the last argument is the Int32 seed literal; every non-NULL
      // value folds into the running hash in argument order (hash_funcs/utils.rs:573-760), NULLs leave it unchanged; never NULL.
      // Same per-type functions as the shuffle writer's partitioning hash.
      if (e.children.size() < 2) throw CometError(f + " expects at least one value and a seed");
      const Expr& seed = *e.children.back();
      if (seed.kind != ExprKind::Literal || seed.lit_null || !seed.has_dtype || seed.dtype.id != (xx ? TypeId::Int64 : TypeId::Int32))
        throw CometError("The seed of function " + f + " must be an " + (xx ? "Int64" : "Int32") + " scalar value");
      const std::string h = newvar(xx ? "u64" : "u32");
      stmt(h + " = " + (xx ? std::to_string((uint64_t)seed.lit_i64) + "ull;" : std::to_string((uint32_t)(int32_t)seed.lit_i64) + "u;"));
      const std::string P = xx ? "comet::xxh64_hash_" : "comet::mm3_hash_";
      for (size_t i = 0; i + 1 < e.children.size(); i++) {
        Val a = arg(i);
        std::string call;
        switch (a.t.id) {
          case TypeId::Bool: call = P + "i32((" + a.v + ") ? 1 : 0, " + h + ")"; break;
          case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: case TypeId::Date: call = P + "i32((i32)" + a.v + ", " + h + ")"; break;
          case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: call = P + "i64((i64)" + a.v + ", " + h + ")"; break;
          case TypeId::Float: call = P + "f32((float)" + a.v + ", " + h + ")"; break;
          case TypeId::Double: call = P + "f64((double)" + a.v + ", " + h + ")"; break;
          case TypeId::Decimal:
            // precision ≤ 18 hashes the unscaled long, wider ones their 16 little-endian bytes (hash_array_decimal)
            call = a.t.precision <= 18 ? P + "i64((i64)" + a.v + ", " + h + ")" : P + "i128(" + as128(a) + ", " + h + ")";
            break;
          default: throw CometError(f + " over " + a.t.str() + " is not supported by the MI355X native engine yet");
        }
        stmt((a.ok.empty() ? std::string() : "if (" + a.ok + ") ") + h + " = " + call + ";");
      }
      r.t = DType::of(xx ? TypeId::Int64 : TypeId::Int32);
      r.rep = xx ? Rep::I64 : Rep::I32;
      r.ok = "";
      r.v = (xx ? "(i64)" : "(i32)") + h;
      r.maxabs = (u128)1 << (xx ? 63 : 31);
      return r;
    }
    if (f == "date_diff" || f == "datediff") {
      // SparkDateDiff (datetime_funcs/date_diff.rs:72-110): end − start in days, wrapping, Int32
      Val a = arg(0), b = arg(1);
      if (a.t.id != TypeId::Date || b.t.id != TypeId::Date) throw CometError("date_diff expects two Date32 arguments");
      r.t = DType::of(TypeId::Int32);
      r.rep = Rep::I32;
      r.ok = and_ok(a.ok, b.ok);
      r.v = "(i32)((u32)" + a.v + " - (u32)" + b.v + ")";
      r.maxabs = (u128)1 << 32;
      return r;
    }
    if (f == "round") {
      // spark_round (math_funcs/round.rs:160-260): HALF_UP at a literal decimal position.  Decimal128: (x + sign·half) / div [· mul];
      // Int32/Int64 with a negative position: round to a power of ten, wrapping (LEGACY) or ARITHMETIC_OVERFLOW (fail_on_error)
      if (e.children.size() != 2 || e.children[1]->kind != ExprKind::Literal || e.children[1]->lit_null || !e.children[1]->dtype.is_integer())
        throw CometError("round expects a literal integer position");
      const long long point = e.children[1]->lit_i64;
      Val a = arg(0);
      r.ok = a.ok;
      if (a.t.id == TypeId::Decimal) {
        if (!e.has_dtype || e.dtype.id != TypeId::Decimal) throw CometError("round: expected a Decimal128 return type");
        r.t = e.dtype;
        r.rep = rep_for_type(e.dtype);
        const int scale = a.t.scale;
        std::string x = as128(a), val;
        if (point < 0) {
          const long long ex = -point + scale;
          if (ex >= 39) val = "(i128)0";
          else {
            const i128 div = (i128)pow10_u128((int)ex), mul = (i128)pow10_u128((int)-point);
            val = "((" + x + " + (" + x + " < 0 ? -" + lit_i128(div / 2) + " : (" + x + " > 0 ? " + lit_i128(div / 2) + " : (i128)0))) / " + lit_i128(div) + " * " + lit_i128(mul) + ")";
          }
        } else {
          const int drop = scale - (int)std::min<long long>(scale, point);
          const i128 div = (i128)pow10_u128(drop);
          val = drop == 0 ? x : "((" + x + " + (" + x + " < 0 ? -" + lit_i128(div / 2) + " : (" + x + " > 0 ? " + lit_i128(div / 2) + " : (i128)0))) / " + lit_i128(div) + ")";
        }
        r.v = r.rep == Rep::I128 ? val : "(i64)(" + val + ")";
        r.maxabs = kUnbounded;
        r.maxabs = std::min<u128>(type_maxabs(e.dtype), kUnbounded);
        return r;
      }
      if ((a.t.id == TypeId::Int64 || a.t.id == TypeId::Int32) && point < 0) {
        const bool is64 = a.t.id == TypeId::Int64;
        const int digits = is64 ? 18 : 9;
        if (-point > digits) throw CometError("round: a position below -" + std::to_string(digits) + " for " + a.t.str() + " is not supported yet");
        const long long div = (long long)pow10_u128((int)-point), half = div / 2;
        const std::string T = is64 ? "i64" : "i32", U = is64 ? "u64" : "u32";
        const std::string x = "(" + T + ")" + a.v;
        std::string rem = newvar(T.c_str());
        stmt(rem + " = " + x + " % (" + T + ")" + std::to_string(div) + "ll;");
        std::string base = newvar(T.c_str());
        stmt(base + " = (" + T + ")((" + U + ")" + x + " - (" + U + ")" + rem + ");");
        std::string adj = newvar(T.c_str());
        stmt(adj + " = " + rem + " <= -(" + T + ")" + std::to_string(half) + "ll ? -(" + T + ")" + std::to_string(div) + "ll : (" + rem + " >= (" + T + ")" + std::to_string(half) + "ll ? (" + T + ")" +
             std::to_string(div) + "ll : (" + T + ")0);");
        if (e.fail_on_error) {
          const std::string mx = is64 ? "(i64)0x7fffffffffffffffll" : "(i32)0x7fffffff", mn = is64 ? "(i64)0x8000000000000000ull" : "(i32)0x80000000";
          raise_if(and_ok(a.ok, "((" + adj + " > 0 && " + base + " > " + mx + " - " + adj + ") || (" + adj + " < 0 && " + base + " < " + mn + " - " + adj + "))"), 1);
        }
        r.t = a.t;
        r.rep = a.rep;
        r.v = "(" + std::string(rep_ctype(a.rep)) + ")(" + T + ")((" + U + ")" + base + " + (" + U + ")" + adj + ")";
        r.maxabs = type_maxabs(a.t);
        return r;
      }
      throw CometError("round is supported for Decimal128 and, with a negative position, Int32 / Int64 (got " + a.t.str() + ")");
    }
    if (f == "sqrt") {
      Val a = arg(0);
      if (a.rep != Rep::F64) throw CometError("sqrt expects a Float64 argument");
      r = a;
      r.v = "__dsqrt_rn(" + a.v + ")";
      return r;
    }
    if (f == "signum") {
      Val a = arg(0);
      if (a.rep != Rep::F64) throw CometError("signum expects a Float64 argument");
      r = a;
      r.v = "(" + a.v + " != " + a.v + " ? " + a.v + " : (" + a.v + " > 0.0 ? 1.0 : (" + a.v + " < 0.0 ? -1.0 : 0.0)))";   // 0 and -0 give 0 (DataFusion signum)
      return r;
    }
    if (f == "isnan") {
      // spark_isnan (predicate_funcs/is_nan.rs:26-67): NULL → false, never NULL
      Val a = arg(0);
      if (a.rep != Rep::F64 && a.rep != Rep::F32) throw CometError("Unsupported data type " + a.t.str() + " for function isnan");
      r.t = DType::of(TypeId::Bool);
      r.rep = Rep::B;
      r.v = "(" + (a.ok.empty() ? std::string("true") : a.ok) + " && " + a.v + " != " + a.v + ")";
      return r;
    }
    if (f == "datepart" || f == "date_part") {
      // CometGetDateField (serde/datetime.scala:36-80): datepart(<field literal>, date) → Int32
      if (e.children.size() != 2 || e.children[0]->kind != ExprKind::Literal) throw CometError("datepart expects (field literal, date)");
      std::string part = e.children[0]->lit_bytes;
      for (auto& ch : part) ch = (char)tolower((unsigned char)ch);
      // (isodow: 1 = Monday … 7 = Sunday — CometWeekDay subtracts one for Spark's weekday(); week: the ISO-8601 week, Spark's weekofyear)
      int code = part == "year" ? 0 : part == "month" ? 1 : part == "day" ? 2 : part == "quarter" ? 3 : part == "dow" ? 4 : part == "doy" ? 5 : part == "isodow" ? 6 : part == "week" ? 7 : -1;
      if (code < 0) throw CometError("datepart field '" + part + "' is not supported by the MI355X native engine yet");
      Val a = arg(1);
      if (a.t.id != TypeId::Date) throw CometError("datepart over " + a.t.str() + " is not supported by the MI355X native engine yet");
      r.t = DType::of(TypeId::Int32);
      r.rep = Rep::I32;
      r.ok = a.ok;
      r.v = code == 6 ? "(comet::date_weekday_mon0(" + a.v + ") + 1)" : code == 7 ? "comet::date_iso_week(" + a.v + ")" : "comet::date_part(" + a.v + ", " + std::to_string(code) + ")";
      r.maxabs = 6000000;
      return r;
    }
    if (f == "instr" || f == "strpos" || f == "ascii" || f == "crc32") {
      // DataFusion's strpos (Spark's instr: the 1-based CHARACTER position of a literal's first occurrence, 0 when there is none), ascii (the first scalar value,
      // 0 for the empty string), datafusion-spark's crc32 (zlib's, as a bigint) — of a Utf8 column (crc32: possibly under Cast(… AS BINARY))
      const ExprP& s0 = f == "crc32" && !e.children.empty() && e.children[0]->kind == ExprKind::Cast && e.children[0]->children.size() == 1 ? e.children[0]->children[0] : e.children.at(0);
      if (!is_str_col(s0)) throw CometError(f + " is supported for a Utf8 column");
      const int idx = s0->bound_index;
      Val valid = str_col_validity(idx);
      auto loc = locate(idx);
      const std::string c = "prm.in[" + std::to_string(loc.first) + "], " + loc.second;
      r.ok = valid.ok;
      if (f == "crc32") {
        r.t = DType::of(TypeId::Int64);
        r.rep = Rep::I64;
        r.v = "comet::utf8_crc32(" + c + ")";
        return r;
      }
      r.t = DType::of(TypeId::Int32);
      r.rep = Rep::I32;
      r.maxabs = (u128)1 << 31;
      if (f == "ascii") { r.v = "comet::utf8_ascii(" + c + ")"; return r; }
      if (e.children.size() != 2 || !is_str_lit(e.children[1])) throw CometError(f + " is supported for a Utf8 column and a literal");
      r.v = "comet::utf8_instr_lit(" + c + ", " + c_bytes(e.children[1]->lit_bytes) + ", " + std::to_string(e.children[1]->lit_bytes.size()) + ")";
      return r;
    }
    if (f == "size" || f == "cardinality") {
      // SparkSizeFunc (array_funcs/size.rs:79-125): the row's element (entry) count, -1 for a NULL list / map — never NULL (CometSize wraps it in a
      // CASE WHEN for spark.sql.legacy.sizeOfNull = false)
      if (e.children.size() != 1) throw CometError("size expects one argument");
      ListRef l = list_ref(e.children[0], "size", false);
      r.t = DType::of(TypeId::Int32);
      r.rep = Rep::I32;
      r.v = l.ok.empty() ? l.len : "(" + l.ok + " ? " + l.len + " : -1)";
      r.maxabs = (u128)1 << 31;
      return r;
    }
    if (f == "array_contains") {
      // datafusion-spark SparkArrayContains (Spark's ArrayContains): NULL for a NULL array or key; true when an element equals the key; otherwise NULL when
      // the array holds a NULL element, else false.  Keys: a literal or a column of the element type (integers, dates, decimals up to 18 digits, booleans;
      // strings against a literal)
      if (e.children.size() != 2) throw CometError("array_contains expects two arguments");
      ListRef l = list_ref(e.children[0], "array_contains", true);
      const bool str = l.etype.id == TypeId::String;
      std::string keyok, cmp;
      const std::string j = newvar("i32");
      if (str) {
        if (!is_str_lit(e.children[1]) && !(e.children[1]->kind == ExprKind::Literal && e.children[1]->lit_null)) throw CometError("array_contains over a list of strings is supported with a literal key");
        if (e.children[1]->lit_null) keyok = "false";
        cmp = "comet::utf8_eq_lit(" + l.ecol + ", (i64)" + l.off + " + " + j + ", " + c_bytes(e.children[1]->lit_bytes) + ", " + std::to_string(e.children[1]->lit_bytes.size()) + ")";
      } else {
        Val key = named(gen(e.children[1]));
        DType kt = key.t, et = l.etype;
        kt.virt_parent = kt.virt_kid = et.virt_parent = et.virt_kid = -1;
        if (!(kt == et)) throw CometError("array_contains: the key's type " + key.t.str() + " is not the elements' " + l.etype.str());
        if (key.rep == Rep::F64 || key.rep == Rep::F32 || key.rep == Rep::STR) throw CometError("array_contains over " + l.etype.str() + " elements is not supported by the MI355X native engine yet");
        keyok = key.ok;
        cmp = "(" + load_of(l.etype, l.ecol, "(i64)" + l.off + " + " + j) + " == " + key.v + ")";
      }
      const std::string found = newvar("bool"), sawnull = newvar("bool");
      const std::string ev = in_valid[(size_t)l.elem] ? "comet::ld_valid(" + l.ecol + ", (i64)" + l.off + " + " + j + ")" : "true";
      const std::string both = and_ok(l.ok, keyok).empty() ? "true" : and_ok(l.ok, keyok);
      stmt(found + " = false; " + sawnull + " = false; if (" + both + ") for (" + j + " = 0; " + j + " < " + l.len + " && !" + found + "; " + j + "++) { if (!" + ev + ") " + sawnull + " = true; else if (" + cmp +
           ") " + found + " = true; }");
      r.t = DType::of(TypeId::Bool);
      r.rep = Rep::B;
      r.v = found;
      const std::string ok = newvar("bool");
      stmt(ok + " = " + both + " && (" + found + " || !" + sawnull + ");");
      r.ok = ok;
      return r;
    }
    // ---- Float64 functions the reference hands to DataFusion / datafusion-spark (QueryPlanSerde.scala:117-174 CometScalarFunction(name); Spark casts the
    // argument to double): the device's libm (ocml) stands where Rust's std — the platform's libm — stands in the reference; both are within an
    // ulp or two of the exact value, the GPU tests state the tolerance.  cot = 1 / tan, csc = 1 / sin, sec = 1 / cos (datafusion-spark), degrees /
    // radians = one multiplication by the constant f64::to_degrees / to_radians use, rint = Java's Math.rint (ties to even).
    {
      static const std::map<std::string, std::string> unary = {
          {"acos", "acos(@)"}, {"acosh", "acosh(@)"}, {"asin", "asin(@)"}, {"asinh", "asinh(@)"}, {"atan", "atan(@)"}, {"atanh", "atanh(@)"}, {"cbrt", "cbrt(@)"},
          {"cos", "cos(@)"}, {"cosh", "cosh(@)"}, {"exp", "exp(@)"}, {"expm1", "expm1(@)"}, {"ln", "log(@)"}, {"log2", "log2(@)"}, {"log10", "log10(@)"},
          {"sin", "sin(@)"}, {"sinh", "sinh(@)"}, {"tan", "tan(@)"}, {"tanh", "tanh(@)"}, {"cot", "(1.0 / tan(@))"}, {"csc", "(1.0 / sin(@))"}, {"sec", "(1.0 / cos(@))"},
          {"degrees", "(@ * (180.0 / 3.14159265358979323846))"}, {"radians", "(@ * (3.14159265358979323846 / 180.0))"}, {"rint", "comet::f64_rint(@)"}};
      auto it = unary.find(f);
      if (it != unary.end()) {
        if (e.children.size() != 1) throw CometError(f + " expects one argument");
        Val a = arg(0);
        if (a.rep != Rep::F64) throw CometError(f + " expects a Float64 argument (got " + a.t.str() + ")");
        r = a;
        std::string x = it->second;
        for (size_t p0 = x.find('@'); p0 != std::string::npos; p0 = x.find('@')) x.replace(p0, 1, a.v);
        r.v = x;
        return r;
      }
    }
    if (f == "pi") {
      r.t = DType::of(TypeId::Double);
      r.rep = Rep::F64;
      r.v = "3.14159265358979323846";
      return r;
    }
    if (f == "atan2" || f == "pow" || f == "power" || f == "spark_log") {
      if (e.children.size() != 2) throw CometError(f + " expects two arguments");
      Val a = arg(0), b = arg(1);
      if (a.rep != Rep::F64 || b.rep != Rep::F64) throw CometError(f + " expects Float64 arguments");
      r.t = DType::of(TypeId::Double);
      r.rep = Rep::F64;
      r.ok = and_ok(a.ok, b.ok);
      if (f == "atan2") r.v = "atan2(" + a.v + ", " + b.v + ")";
      else if (f == "spark_log") {
        // spark_log (math_funcs/log.rs:30-38): log(base, value) = ln(value) / ln(base), NULL when base <= 0 or value <= 0
        r.ok = and_ok(r.ok, "(" + a.v + " > 0.0 && " + b.v + " > 0.0)");
        r.v = "(log(" + b.v + ") / log(" + a.v + "))";
      } else {
        // spark_powf (math_funcs/pow.rs:24-29): Java's Math.pow — |base| = 1 with an infinite or NaN exponent is NaN (C's pow says 1)
        r.v = "((fabs(" + a.v + ") == 1.0 && !isfinite(" + b.v + ")) ? __longlong_as_double(0x7ff8000000000000ll) : pow(" + a.v + ", " + b.v + "))";
      }
      return r;
    }
    if (f == "factorial") {
      // datafusion-spark SparkFactorial (Spark's Factorial): Int32 in 0..20 → its factorial as Int64, NULL outside
      Val a = arg(0);
      if (a.t.id != TypeId::Int32) throw CometError("factorial expects an Int32 argument");
      if (decls.find("fact_tab[21]") == std::string::npos)      // once per kernel: two distinct factorial calls share the table
        decls += "    static const i64 fact_tab[21] = {1ll,1ll,2ll,6ll,24ll,120ll,720ll,5040ll,40320ll,362880ll,3628800ll,39916800ll,479001600ll,6227020800ll,87178291200ll,1307674368000ll,"
               "20922789888000ll,355687428096000ll,6402373705728000ll,121645100408832000ll,2432902008176640000ll};\n";
      r.t = DType::of(TypeId::Int64);
      r.rep = Rep::I64;
      r.ok = and_ok(a.ok, "(" + a.v + " >= 0 && " + a.v + " <= 20)");
      r.v = "fact_tab[(" + a.v + " >= 0 && " + a.v + " <= 20) ? " + a.v + " : 0]";
      return r;
    }
    if (f == "bitwise_not" || f == "bit_count" || f == "bit_get" || f == "getbit" || f == "shiftrightunsigned") {
      // datafusion-spark's bitwise functions (jni_api.rs:639-668): Java's ~x, Integer / Long.bitCount (a narrower value sign-extended first), (x >> pos) & 1
      // as a byte, x >>> (n mod width)
      Val a = arg(0);
      if (!a.t.is_integer() && !(f == "bit_count" && a.t.id == TypeId::Bool)) throw CometError(f + " expects an integral argument (got " + a.t.str() + ")");
      const bool is64 = a.t.id == TypeId::Int64;
      if (f == "bitwise_not") {
        r = a;
        r.v = "(" + std::string(rep_ctype(a.rep)) + ")(~(" + a.v + "))";
        return r;
      }
      if (f == "bit_count") {
        r.t = DType::of(TypeId::Int32);
        r.rep = Rep::I32;
        r.ok = a.ok;
        r.v = a.t.id == TypeId::Bool ? "(i32)(" + a.v + " ? 1 : 0)" : is64 ? "(i32)__popcll((u64)" + a.v + ")" : "(i32)__popcll((u64)(i64)" + a.v + ")";      // (Spark counts the bits of the value widened to a long)
        r.maxabs = 64;
        return r;
      }
      Val b = arg(1);
      if (!(b.t.id == TypeId::Int32 || b.t.id == TypeId::Int8 || b.t.id == TypeId::Int16)) throw CometError(f + " expects an Int32 second argument");
      r.ok = and_ok(a.ok, b.ok);
      if (f == "shiftrightunsigned") {
        if (a.t.id != TypeId::Int32 && a.t.id != TypeId::Int64) throw CometError("shiftrightunsigned expects an Int32 or Int64 value");
        r.t = a.t;
        r.rep = a.rep;
        r.v = is64 ? "(i64)((u64)" + a.v + " >> ((int)" + b.v + " & 63))" : "(i32)((u32)" + a.v + " >> ((int)" + b.v + " & 31))";
        r.maxabs = type_maxabs(a.t);
        return r;
      }
      // bit_get: a position outside the value's bits is an error in Spark (and in datafusion-spark): refused rows raise nothing here — the position must be a literal in range
      const int width = a.t.id == TypeId::Int8 ? 8 : a.t.id == TypeId::Int16 ? 16 : a.t.id == TypeId::Int32 ? 32 : 64;
      if (e.children[1]->kind != ExprKind::Literal || e.children[1]->lit_null || e.children[1]->lit_i64 < 0 || e.children[1]->lit_i64 >= width)
        throw CometError("bit_get is supported with a literal position inside the value's " + std::to_string(width) + " bits");
      r.t = DType::of(TypeId::Int8);
      r.rep = Rep::I32;
      r.v = "(i32)(((u64)(i64)" + a.v + " >> " + std::to_string(e.children[1]->lit_i64) + ") & 1ull)";
      r.maxabs = 1;
      return r;
    }
    if (f == "greatest" || f == "least") {
      // DataFusion's greatest / least (Spark's): NULL arguments are skipped, the result is NULL only when every argument is; NaN is the greatest double
      if (e.children.size() < 2) throw CometError(f + " expects at least two arguments");
      Val acc = arg(0);
      for (size_t k = 1; k < e.children.size(); k++) {
        Val b = arg(k);
        if (!(b.t == acc.t)) throw CometError(f + " expects arguments of one type (got " + acc.t.str() + " and " + b.t.str() + ")");
        if (acc.rep == Rep::STR || acc.rep == Rep::B) throw CometError(f + " of " + acc.t.str() + " is not supported by the MI355X native engine yet");
        std::string better;      // b beats acc
        if (acc.rep == Rep::F64 || acc.rep == Rep::F32) {
          const std::string an = "(" + acc.v + " != " + acc.v + ")", bn = "(" + b.v + " != " + b.v + ")";
          better = f == "greatest" ? "(" + bn + " || (!" + an + " && " + b.v + " > " + acc.v + "))" : "(" + an + " || (!" + bn + " && " + b.v + " < " + acc.v + "))";
        } else {
          better = "(" + b.v + (f == "greatest" ? " > " : " < ") + acc.v + ")";
        }
        const std::string aok = acc.ok.empty() ? "true" : acc.ok, bok = b.ok.empty() ? "true" : b.ok;
        std::string take = newvar("bool");
        stmt(take + " = " + bok + " && (!" + aok + " || " + better + ");");
        Val n = acc;
        std::string v = newvar(rep_ctype(acc.rep));
        stmt(v + " = " + take + " ? " + b.v + " : " + acc.v + ";");
        n.v = v;
        if (acc.ok.empty() || b.ok.empty()) n.ok = "";
        else {
          std::string o = newvar("bool");
          stmt(o + " = " + acc.ok + " || " + b.ok + ";");
          n.ok = o;
        }
        n.maxabs = std::max(acc.maxabs, b.maxabs);
        acc = n;
      }
      return acc;
    }
    // ---- dates ----
    if (f == "last_day" || f == "date_from_unix_date") {
      Val a = arg(0);
      r.t = DType::of(TypeId::Date);
      r.rep = Rep::I32;
      r.maxabs = (u128)1 << 31;
      if (f == "date_from_unix_date") {
        // SparkDateFromUnixDate (datetime_funcs/date_from_unix_date.rs:52-60): the Int32 IS the date
        if (a.t.id != TypeId::Int32) throw CometError("date_from_unix_date expects Int32Array input");
        r.ok = a.ok;
        r.v = a.v;
        return r;
      }
      if (a.t.id != TypeId::Date) throw CometError("last_day expects a Date32 argument");
      r.ok = and_ok(a.ok, "comet::date_in_chrono_range(" + a.v + ")");
      r.v = "comet::date_last_day(" + a.v + ")";
      return r;
    }
    if (f == "date_trunc" || f == "trunc") {
      // SparkDateTrunc (datetime_funcs/date_trunc.rs → kernels/temporal.rs:63-100, 326-352): a Date32 to the first day of its year / quarter / month /
      // (Monday) week; the format is a literal here (datetime.scala: anything else is sent only under allowIncompatible)
      if (e.children.size() != 2 || e.children[1]->kind != ExprKind::Literal || e.children[1]->lit_null) throw CometError("date_trunc is supported with a literal format");
      std::string fmt = e.children[1]->lit_bytes;
      for (auto& ch : fmt) ch = (char)toupper((unsigned char)ch);
      const int unit = (fmt == "YEAR" || fmt == "YYYY" || fmt == "YY") ? 0 : fmt == "QUARTER" ? 1 : (fmt == "MONTH" || fmt == "MON" || fmt == "MM") ? 2 : fmt == "WEEK" ? 3 : -1;
      if (unit < 0) throw CometError("Unsupported format: \"" + e.children[1]->lit_bytes + "\" for function 'date_trunc'");
      Val a = arg(0);
      if (a.t.id != TypeId::Date) throw CometError("Invalid input to function DateTrunc. Expected (Date32, Utf8)");
      r.t = DType::of(TypeId::Date);
      r.rep = Rep::I32;
      r.ok = and_ok(a.ok, "comet::date_in_chrono_range(" + a.v + ")");
      r.v = "comet::date_trunc_days(" + a.v + ", " + std::to_string(unit) + ")";
      r.maxabs = (u128)1 << 31;
      return r;
    }
    if (f == "next_day") {
      // SparkNextDay (datetime_funcs/next_day.rs:49-68): the first date later than the start that falls on the named day; a name that is none is NULL
      // (IllegalDayOfWeek under ANSI).  The day's name is a literal here.
      if (e.children.size() != 2 || e.children[1]->kind != ExprKind::Literal) throw CometError("next_day is supported with a literal day of the week");
      Val a = arg(0);
      if (a.t.id != TypeId::Date) throw CometError("next_day expects a Date32 start date");
      r.t = DType::of(TypeId::Date);
      r.rep = Rep::I32;
      r.maxabs = (u128)1 << 31;
      std::string name = e.children[1]->lit_bytes;
      for (auto& ch : name) ch = (char)toupper((unsigned char)ch);
      static const char* names[7][3] = {{"MO", "MON", "MONDAY"}, {"TU", "TUE", "TUESDAY"}, {"WE", "WED", "WEDNESDAY"}, {"TH", "THU", "THURSDAY"}, {"FR", "FRI", "FRIDAY"}, {"SA", "SAT", "SATURDAY"}, {"SU", "SUN", "SUNDAY"}};
      int day = -1;
      for (int d = 0; d < 7 && !e.children[1]->lit_null; d++)
        for (int k = 0; k < 3; k++)
          if (name == names[d][k]) day = d;
      if (day < 0) {
        if (e.fail_on_error && !e.children[1]->lit_null) throw CometError("{\"errorType\":\"IllegalDayOfWeek\",\"errorClass\":\"ILLEGAL_DAY_OF_WEEK\",\"params\":{\"string\":\"" + e.children[1]->lit_bytes + "\"}}", 1);
        r.ok = "false";
        r.v = "0";
        return r;
      }
      r.ok = and_ok(a.ok, "comet::date_in_chrono_range(" + a.v + ")");
      r.v = "comet::date_next_day(" + a.v + ", " + std::to_string(day) + ")";
      return r;
    }
    if (f == "make_date") {
      // SparkMakeDate (datetime_funcs/make_date.rs:87-96, 150-166): chrono's from_ymd_opt — anything it refuses is NULL; under ANSI the task fails
      // (DatetimeFieldOutOfBounds), which is refused here at planning time
      if (e.children.size() != 3) throw CometError("make_date expects three arguments");
      if (e.fail_on_error) throw CometError("make_date in ANSI mode (DATETIME_FIELD_OUT_OF_BOUNDS) is not supported by the MI355X native engine yet");
      Val y = arg(0), m = arg(1), d = arg(2);
      for (const Val* x : {&y, &m, &d})
        if (!(x->t.id == TypeId::Int32 || x->t.id == TypeId::Int16 || x->t.id == TypeId::Int8)) throw CometError("make_date expects Int32 arguments");
      std::string out = newvar("i32"), ok = newvar("bool");
      stmt(out + " = 0; " + ok + " = comet::date_make(" + y.v + ", " + m.v + ", " + d.v + ", " + out + ");");
      r.t = DType::of(TypeId::Date);
      r.rep = Rep::I32;
      r.ok = and_ok(and_ok(and_ok(y.ok, m.ok), d.ok), ok);
      r.v = out;
      r.maxabs = (u128)1 << 31;
      return r;
    }
    if (f == "seconds_to_timestamp" || f == "timestamp_seconds") {
      // SparkSecondsToTimestamp (datetime_funcs/seconds_to_timestamp.rs:63-105): Int32 · 10^6; Int64 · 10^6 checked ("long overflow" fails the task:
      // refused rows cannot be told from here, so Int64 is taken when the product cannot overflow only … it can: the check runs on the device);
      // Float32 / Float64: (s · 10^6) as i64 — Rust's saturating conversion —, NaN / ±∞ are NULL
      Val a = arg(0);
      r.t = DType::of(TypeId::Timestamp);
      r.rep = Rep::I64;
      r.ok = a.ok;
      if (a.t.id == TypeId::Int32) r.v = "((i64)" + a.v + " * 1000000ll)";
      else if (a.t.id == TypeId::Int64) {
        raise_if(and_ok(a.ok, "(" + a.v + " > 9223372036854ll || " + a.v + " < -9223372036854ll)"), 18);
        r.v = "(i64)((u64)" + a.v + " * 1000000ull)";
      } else if (a.rep == Rep::F64 || a.rep == Rep::F32) {
        r.ok = and_ok(a.ok, "isfinite((double)" + a.v + ")");
        r.v = "comet::f64_to_i64_sat((double)" + a.v + " * 1000000.0)";
      } else throw CometError("seconds_to_timestamp expects Int32, Int64, Float32 or Float64 input, got " + a.t.str());
      return r;
    }
    throw CometError("Scalar function '" + f + "' is not supported by the MI355X native engine");
  }

  // ---- direct Utf8 comparisons (device/comet_device.hpp utf8_cmp*) ----
  bool is_str_col(const ExprP& x) const {
    return x->kind == ExprKind::Bound && x->bound_index >= 0 && (size_t)x->bound_index < in_types.size() &&
           (in_types[(size_t)x->bound_index].id == TypeId::String || in_types[(size_t)x->bound_index].id == TypeId::Bytes);
  }
  static bool is_str_lit(const ExprP& x) {
    return x->kind == ExprKind::Literal && !x->lit_null && (x->dtype.id == TypeId::String || x->dtype.id == TypeId::Bytes);
  }
  static std::string c_bytes(const std::string& b) {
    static const char* hx = "0123456789abcdef";
    std::string o = "\"";
    for (unsigned char ch : b) { o += "\\x"; o += hx[ch >> 4]; o += hx[ch & 15]; o += "\"\""; }   // "\x41""\x42": hex escapes never run together
    return o + "\"";
  }
  // a Val that only carries the column's validity (no string load)
  Val str_col_validity(int idx) {
    in_used[(size_t)idx] = true;
    Val v;
    v.t = in_types[(size_t)idx];
    v.rep = Rep::B;
    v.v = "true";
    if (in_valid[(size_t)idx]) {
      auto loc = locate(idx);
      std::string o = newloadvar("bool");
      load(o + " = comet::ld_valid(prm.in[" + std::to_string(loc.first) + "], " + loc.second + ");");
      v.ok = o;
    }
    return v;
  }
  static std::string cmp_to_bool(ExprKind k, const std::string& c, bool swapped) {
    // c = cmp(column, other); swapped: the expression was  other <op> column
    switch (k) {
      case ExprKind::Eq: case ExprKind::EqNullSafe: return "(" + c + " == 0)";
      case ExprKind::Neq: case ExprKind::NeqNullSafe: return "(" + c + " != 0)";
      case ExprKind::Gt: return "(" + c + (swapped ? " < 0)" : " > 0)");
      case ExprKind::GtEq: return "(" + c + (swapped ? " <= 0)" : " >= 0)");
      case ExprKind::Lt: return "(" + c + (swapped ? " > 0)" : " < 0)");
      case ExprKind::LtEq: return "(" + c + (swapped ? " >= 0)" : " <= 0)");
      default: throw CometError("bad comparison");
    }
  }
  Val str_compare_lit(ExprKind k, int idx, const Expr& lit, bool swapped) {
    Val valid = str_col_validity(idx);
    auto loc = locate(idx);
    const std::string col = "prm.in[" + std::to_string(loc.first) + "]";
    const std::string n = std::to_string(lit.lit_bytes.size());
    const bool eqlike = k == ExprKind::Eq || k == ExprKind::Neq || k == ExprKind::EqNullSafe || k == ExprKind::NeqNullSafe;
    std::string b = newvar("bool");
    if (eqlike) {
      const bool neg = k == ExprKind::Neq || k == ExprKind::NeqNullSafe;
      stmt(b + " = " + (neg ? "!" : "") + "comet::utf8_eq_lit(" + col + ", " + loc.second + ", " + c_bytes(lit.lit_bytes) + ", " + n + ");");
    } else {
      stmt(b + " = " + cmp_to_bool(k, "comet::utf8_cmp_lit(" + col + ", " + loc.second + ", " + c_bytes(lit.lit_bytes) + ", " + n + ")", swapped) + ";");
    }
    Val r;
    r.t = DType::of(TypeId::Bool);
    r.rep = Rep::B;
    if (k == ExprKind::EqNullSafe || k == ExprKind::NeqNullSafe) {
      // the literal is not NULL: <=> is false (and its negation true) for a NULL column value
      if (valid.ok.empty()) r.v = b;
      else r.v = k == ExprKind::EqNullSafe ? "(" + valid.ok + " && " + b + ")" : "(!" + valid.ok + " || " + b + ")";
      return r;
    }
    r.v = valid.ok.empty() ? b : "(" + valid.ok + " && " + b + ")";
    r.ok = valid.ok;
    return r;
  }
  // boolean predicate `fn(column, literal bytes)` evaluated on the bytes in place; NULL for a NULL column value
  Val str_pred_lit(const std::string& fn, int idx, const std::string& lit, bool negate = false) {
    Val valid = str_col_validity(idx);
    auto loc = locate(idx);
    std::string b = newvar("bool");
    stmt(b + " = " + (negate ? "!" : "") + "comet::" + fn + "(prm.in[" + std::to_string(loc.first) + "], " + loc.second + ", " + c_bytes(lit) + ", " +
         std::to_string(lit.size()) + ");");
    Val r;
    r.t = DType::of(TypeId::Bool);
    r.rep = Rep::B;
    r.v = valid.ok.empty() ? b : "(" + valid.ok + " && " + b + ")";
    r.ok = valid.ok;
    return r;
  }
  // LIKE (expr.proto:56 → DataFusion LikeExpr): patterns without `_` / escapes of the shapes lit, lit%, %lit, %lit% become
  // equality / prefix / suffix / substring tests (what arrow-string's like kernel does too); everything else runs the matcher
  Val like(const Expr& e) {
    if (e.children.size() != 2 || !is_str_col(e.children[0]) || !is_str_lit(e.children[1]))
      throw CometError("LIKE is supported for a Utf8 column and a literal pattern");
    const int idx = e.children[0]->bound_index;
    const std::string& p = e.children[1]->lit_bytes;
    const bool special = p.find('_') != std::string::npos || p.find('\\') != std::string::npos;
    if (!special) {
      if (p.find('%') == std::string::npos) return str_pred_lit("utf8_eq_lit", idx, p);
      const bool lead = p.front() == '%', trail = p.size() > 1 && p.back() == '%';
      const size_t b = lead ? 1 : 0, en = p.size() - (trail ? 1 : 0);
      const std::string mid = p.substr(b, en - b);
      if (mid.find('%') == std::string::npos) {
        if (lead && trail) return str_pred_lit("utf8_contains_lit", idx, mid);
        if (lead) return str_pred_lit("utf8_ends_with_lit", idx, mid);      // "%" alone: every non-NULL value ends with ""
        if (trail) return str_pred_lit("utf8_starts_with_lit", idx, mid);
      }
    }
    return str_pred_lit("utf8_like_lit", idx, p);
  }
  // RLike (expr.proto:55 → predicate_funcs/rlike.rs: regex::Regex::is_match, an unanchored search): the pattern is compiled on the host
  // into a byte-level search DFA (regex.cpp — the exactly reproducible subset, everything else refused by name) whose tables are
  // emitted into the kernel source as constants
  Val rlike(const Expr& e) {
    if (e.children.size() != 2 || !is_str_col(e.children[0]) || !is_str_lit(e.children[1]))
      throw CometError("RLIKE is supported for a Utf8 column and a literal pattern");
    const int idx = e.children[0]->bound_index;
    const RegexDfa dfa = compile_rlike(e.children[1]->lit_bytes);
    Val valid = str_col_validity(idx);
    auto loc = locate(idx);
    std::string b = newvar("bool");
    stmt(b + " = comet::utf8_rlike(prm.in[" + std::to_string(loc.first) + "], " + loc.second + ", " + c_bytes(std::string((const char*)dfa.trans.data(), dfa.trans.size())) + ", " +
         c_bytes(std::string((const char*)dfa.flags.data(), dfa.flags.size())) + ", " + c_bytes(std::string((const char*)dfa.classes.data(), dfa.classes.size())) + ", " +
         std::to_string(dfa.nclasses) + "u);");
    Val r;
    r.t = DType::of(TypeId::Bool);
    r.rep = Rep::B;
    r.v = valid.ok.empty() ? b : "(" + valid.ok + " && " + b + ")";
    r.ok = valid.ok;
    return r;
  }
  Val str_compare_cols(ExprKind k, int ia, int ib) {
    Val va = str_col_validity(ia), vb = str_col_validity(ib);
    auto la = locate(ia), lb = locate(ib);
    std::string b = newvar("bool");
    stmt(b + " = " + cmp_to_bool(k, "comet::utf8_cmp(prm.in[" + std::to_string(la.first) + "], " + la.second + ", prm.in[" + std::to_string(lb.first) + "], " +
                                        lb.second + ")", false) + ";");
    Val r;
    r.t = DType::of(TypeId::Bool);
    r.rep = Rep::B;
    const std::string aok = va.ok.empty() ? "true" : va.ok, bok = vb.ok.empty() ? "true" : vb.ok;
    if (k == ExprKind::EqNullSafe || k == ExprKind::NeqNullSafe) {
      const std::string e2 = "((" + aok + " && " + bok + " && " + (k == ExprKind::EqNullSafe ? b : "!" + b) + ") || (!" + aok + " && !" + bok + "))";
      r.v = k == ExprKind::EqNullSafe ? e2 : "(!" + e2 + ")";
      return r;
    }
    r.v = "(" + aok + " && " + bok + " && " + b + ")";
    r.ok = and_ok(va.ok, vb.ok);
    return r;
  }

  // c ? t : f with SQL NULL handling: a NULL condition selects f (If: conditional_funcs/if_expr.rs:103; CaseWhen alike)
  Val select(const Val& c, const Val& t0, const Val& f0) {
    Val t = t0, f = f0;
    // decimals of one type may sit in 64 or 128 bits depending on their static bound: meet in 128
    if (t.rep != f.rep && t.t.id == TypeId::Decimal && f.t.id == TypeId::Decimal && (t.rep == Rep::I64 || t.rep == Rep::I128) &&
        (f.rep == Rep::I64 || f.rep == Rep::I128)) {
      t.v = as128(t); t.rep = Rep::I128;
      f.v = as128(f); f.rep = Rep::I128;
    }
    if (t.rep != f.rep) throw CometError("If / CaseWhen branches have different types");
    std::string cond = "(" + and_ok(c.ok, c.v) + ")";
    Val r = t;
    r.maxabs = std::max(t.maxabs, f.maxabs);
    r.v = "(" + cond + " ? " + t.v + " : " + f.v + ")";
    if (!t.ok.empty() || !f.ok.empty())
      r.ok = "(" + cond + " ? " + (t.ok.empty() ? "true" : t.ok) + " : " + (f.ok.empty() ? "true" : f.ok) + ")";
    else r.ok = "";
    r.wide_decimal = false;
    return r;
  }

  Val gen(const ExprP& ep) {
    const Expr& e = *ep;
    std::string key = key_of(ep);
    auto it = cse.find(key);
    if (it != cse.end()) return it->second;
    Val out = gen_uncached(e);
    // keep cheap leaf expressions inline, name everything else once
    if (e.kind != ExprKind::Literal && e.kind != ExprKind::Bound) out = named(out);
    cse[key] = out;
    return out;
  }

  Val gen_uncached(const Expr& e) {
    struct Scope {
      std::vector<std::shared_ptr<QueryContext>>& st;
      ~Scope() { st.pop_back(); }
    } scope{ctx_stack};
    ctx_stack.push_back(e.qctx && e.has_expr_id ? e.qctx : nullptr);
    return gen_node(e);
  }

  Val gen_node(const Expr& e) {
    switch (e.kind) {
      case ExprKind::Bound: return column(e.bound_index);
      case ExprKind::Literal: return literal(e);
      case ExprKind::Add: case ExprKind::Subtract: case ExprKind::Multiply: case ExprKind::Divide: case ExprKind::Remainder:
      case ExprKind::IntegralDivide: {
        if (e.children.size() != 2) throw CometError("binary expression needs two children");
        Val a = gen(e.children[0]), b = gen(e.children[1]);
        return arithmetic(e, a, b);
      }
      case ExprKind::Eq: case ExprKind::Neq: case ExprKind::Gt: case ExprKind::GtEq: case ExprKind::Lt: case ExprKind::LtEq:
      case ExprKind::EqNullSafe: case ExprKind::NeqNullSafe: {
        if (e.children.size() != 2) throw CometError("comparison needs two children");
        {
          // Utf8 column against a Utf8 literal or another Utf8 column: compare the bytes in place (any length, all six
          // operators) instead of going through the packed ≤15-byte form
          const ExprP &c0 = e.children[0], &c1 = e.children[1];
          if (is_str_col(c0) && is_str_lit(c1)) return str_compare_lit(e.kind, c0->bound_index, *c1, false);
          if (is_str_lit(c0) && is_str_col(c1)) return str_compare_lit(e.kind, c1->bound_index, *c0, true);
          if (is_str_col(c0) && is_str_col(c1)) return str_compare_cols(e.kind, c0->bound_index, c1->bound_index);
        }
        Val a = gen(e.children[0]), b = gen(e.children[1]);
        return compare(e.kind, a, b);
      }
      case ExprKind::BitAnd: case ExprKind::BitOr: case ExprKind::BitXor: case ExprKind::ShiftLeft: case ExprKind::ShiftRight: {
        // Spark BitwiseAnd / Or / Xor (same integral type on both sides) and ShiftLeft / ShiftRight (Java semantics: the shift count is
        // taken modulo the width of the value, >> is arithmetic)
        if (e.children.size() != 2) throw CometError("bitwise expression needs two children");
        Val a = named(gen(e.children[0])), b = named(gen(e.children[1]));
        if (!a.t.is_integer() || !b.t.is_integer()) throw CometError("bitwise operators expect integral operands");
        const bool shift = e.kind == ExprKind::ShiftLeft || e.kind == ExprKind::ShiftRight;
        if (!shift && a.t.id != b.t.id) throw CometError("bitwise operator on different integral types");
        const bool is64 = a.t.id == TypeId::Int64;
        const std::string T = is64 ? "i64" : "i32", U = is64 ? "u64" : "u32";
        Val r = a;
        r.ok = and_ok(a.ok, b.ok);
        r.maxabs = type_maxabs(a.t);
        std::string v;
        if (e.kind == ExprKind::BitAnd) v = "(" + T + ")" + a.v + " & (" + T + ")" + b.v;
        else if (e.kind == ExprKind::BitOr) v = "(" + T + ")" + a.v + " | (" + T + ")" + b.v;
        else if (e.kind == ExprKind::BitXor) v = "(" + T + ")" + a.v + " ^ (" + T + ")" + b.v;
        else if (e.kind == ExprKind::ShiftLeft) v = "(" + T + ")((" + U + ")(" + T + ")" + a.v + " << ((int)" + b.v + " & " + (is64 ? "63" : "31") + "))";
        else v = "(" + T + ")" + a.v + " >> ((int)" + b.v + " & " + (is64 ? "63" : "31") + ")";
        // narrow types keep their width: Byte / Short results wrap like the JVM's (byte) / (short) casts
        if (a.t.id == TypeId::Int8) v = "(i32)(i8)(" + v + ")";
        else if (a.t.id == TypeId::Int16) v = "(i32)(i16)(" + v + ")";
        r.v = "(" + std::string(rep_ctype(a.rep)) + ")(" + v + ")";
        return r;
      }
      case ExprKind::And: case ExprKind::Or: {
        if (e.children.size() != 2) throw CometError("AND/OR needs two children");
        Val a = gen(e.children[0]), b = gen(e.children[1]);
        return logic(e.kind, a, b);
      }
      case ExprKind::Not: {
        Val a = gen(e.children.at(0));
        if (a.rep != Rep::B) throw CometError("NOT expects a boolean operand");
        Val r = a;
        r.v = "(!" + a.v + ")";
        return r;
      }
      case ExprKind::IsNull: case ExprKind::IsNotNull: {
        // a Utf8 column's NULL-ness needs its validity bit only, never the (packed, ≤15-byte) value
        const ExprP& c0 = e.children.at(0);
        const bool nested_col = c0->kind == ExprKind::Bound && c0->bound_index >= 0 && (size_t)c0->bound_index < in_types.size() && in_types[(size_t)c0->bound_index].is_nested();
        Val a = (is_str_col(c0) || nested_col) ? str_col_validity(c0->bound_index) : gen(c0);      // (a nested column: its validity bit, like a Utf8 column's)
        Val r;
        r.t = DType::of(TypeId::Bool);
        r.rep = Rep::B;
        std::string ok = a.ok.empty() ? "true" : a.ok;
        r.v = e.kind == ExprKind::IsNull ? "(!" + ok + ")" : ok;
        return r;
      }
      case ExprKind::CheckOverflow: {
        Val c = gen(e.children.at(0));
        if (c.t.id != TypeId::Decimal || e.dtype.id != TypeId::Decimal)
          throw CometError("CheckOverflow expects only Decimal128, but got " + c.t.str());
        // planner.rs:606-613: WideDecimalBinaryExpr already checked, same type → child itself
        if (c.wide_decimal && c.t == e.dtype) return c;
        // planner.rs:615-633: Cast(dec→dec)+CheckOverflow with equal target → fused rescale+check
        if (c.is_cast_dec && c.t == e.dtype) {
          Val child;
          child.v = c.cast_child_v; child.ok = c.cast_child_ok; child.t = c.cast_child_t; child.rep = c.cast_child_rep;
          child.maxabs = c.cast_child_max;
          Val x = rescale(child, child.t.scale, e.dtype.precision, e.dtype.scale, e.fail_on_error);
          x.is_cast_dec = false;
          return x;
        }
        if (c.t.scale != e.dtype.scale)
          throw CometError("CheckOverflow cannot change scale (" + c.t.str() + " → " + e.dtype.str() + ")");
        Val x = c;
        x.t = e.dtype;
        x.wide_decimal = false;
        x.is_cast_dec = false;
        const ErrSite site = out_of_range_site(e.dtype);      // checkoverflow.rs:141-147: the first value that does not fit, unscaled
        return bound_check(x, e.dtype.precision, e.fail_on_error, 3, &site);
      }
      case ExprKind::Hour: case ExprKind::Minute: case ExprKind::Second: {
        // SparkHour / SparkMinute / SparkSecond (datetime_funcs/extract_date_part.rs:83-110): of the session zone's wall clock; a TIMESTAMP_NTZ is one already
        Val c = named(gen(e.children.at(0)));
        if (c.t.id != TypeId::Timestamp && c.t.id != TypeId::TimestampNtz) throw CometError(std::string(expr_name(e.proto_tag)) + " over " + c.t.str() + " is not supported by the MI355X native engine");
        const std::string local = c.t.id == TypeId::Timestamp ? local_of(e.func, c) : c.v;
        Val r;
        r.t = DType::of(TypeId::Int32);
        r.rep = Rep::I32;
        r.ok = c.ok;
        const std::string sec = "(" + local + " - comet::tz_floor_div(" + local + ", 86400000000ll) * 86400000000ll) / 1000000ll";
        r.v = e.kind == ExprKind::Hour ? "(i32)((" + sec + ") / 3600)" : e.kind == ExprKind::Minute ? "(i32)((" + sec + ") / 60 % 60)" : "(i32)((" + sec + ") % 60)";
        r.maxabs = 64;
        return r;
      }
      case ExprKind::Subquery: {
        // a scalar subquery the executor has not asked for yet (createPlan's dry generation): a NULL of its type — at the first executePlan the node becomes
        // a Literal and the plan is generated anew under a hash that carries the value (exec.cpp resolve_subqueries)
        Expr nul;
        nul.kind = ExprKind::Literal;
        nul.dtype = e.dtype;
        nul.has_dtype = true;
        nul.lit_null = true;
        return literal(nul);
      }
      case ExprKind::ListExtract: return list_extract(e);
      case ExprKind::TruncTimestamp: {
        // timestamp_trunc (datetime_funcs/timestamp_trunc.rs → kernels/temporal.rs:179-270, 587-625): the instant's wall clock in the zone cut to the
        // unit, read back as an instant.  The reference sends it for UTC only unless allowIncompatible (datetime.scala CometTruncTimestamp): zones
        // with transitions are refused here.
        if (e.children.size() != 2) throw CometError("TruncTimestamp expects a format and a child");
        const Expr& fe = *e.children[0];
        if (fe.kind != ExprKind::Literal || fe.lit_null) throw CometError("TruncTimestamp is supported with a literal format");
        std::string fmt = fe.lit_bytes;
        for (auto& ch : fmt) ch = (char)toupper((unsigned char)ch);
        static const std::map<std::string, int> units = {{"YEAR", 0}, {"YYYY", 0}, {"YY", 0}, {"QUARTER", 1}, {"MONTH", 2}, {"MON", 2}, {"MM", 2}, {"WEEK", 3}, {"DAY", 4}, {"DD", 4},
                                                         {"HOUR", 5}, {"MINUTE", 6}, {"SECOND", 7}, {"MILLISECOND", 8}, {"MICROSECOND", 9}};
        auto it = units.find(fmt);
        if (it == units.end()) throw CometError("Unsupported format: \"" + fe.lit_bytes + "\" for function 'timestamp_trunc'");
        Val c = named(gen(e.children[1]));
        if (c.t.id != TypeId::Timestamp && c.t.id != TypeId::TimestampNtz) throw CometError("timestamp_trunc does not support " + c.t.str());
        long long secs = 0;
        if (c.t.id == TypeId::Timestamp && !fixed_zone_offset(e.func, secs)) throw CometError("TruncTimestamp in a time zone with transitions ('" + e.func + "') is not supported by the MI355X native engine");
        Val r = c;
        const std::string off = lit_i64(secs * 1000000);
        r.v = "(comet::ts_trunc_local_us(" + c.v + " + " + off + ", " + std::to_string(it->second) + ") - " + off + ")";
        return r;
      }
      case ExprKind::UnixTimestamp: {
        // SparkUnixTimestamp (datetime_funcs/unix_timestamp.rs:70-150): an instant's whole seconds (floor); a TIMESTAMP_NTZ's likewise; a date's midnight
        // in the session zone, as an instant
        Val c = named(gen(e.children.at(0)));
        Val r;
        r.t = DType::of(TypeId::Int64);
        r.rep = Rep::I64;
        r.ok = c.ok;
        if (c.t.id == TypeId::Timestamp || c.t.id == TypeId::TimestampNtz) r.v = "comet::floor_div_i64(" + c.v + ", 1000000ll)";
        else if (c.t.id == TypeId::Date) {
          const std::string local = "((i64)" + c.v + " * 86400000000ll)";
          r.v = "comet::floor_div_i64(" + utc_of(e.func, local, c.ok) + ", 1000000ll)";
        } else throw CometError("unix_timestamp does not support input type: " + c.t.str());
        return r;
      }
      case ExprKind::Cast: {
        const ExprP& c0 = e.children.at(0);
        if (is_str_col(c0) && in_types[(size_t)c0->bound_index].id == TypeId::String && e.dtype.id != TypeId::String) return cast_from_string(e, c0->bound_index);
        return cast(e, gen(c0));
      }
      case ExprKind::UnaryMinus: {
        Val a = gen(e.children.at(0));
        Val r = a;
        switch (a.rep) {
          case Rep::I32:      // (a tinyint / smallint lives in 32 bits: its minimum negates onto itself like Rust's wrapping_neg of the narrow type)
            r.v = a.t.id == TypeId::Int8 ? "(i32)(i8)(0u - (u32)" + a.v + ")" : a.t.id == TypeId::Int16 ? "(i32)(i16)(0u - (u32)" + a.v + ")" : "(i32)(0u - (u32)" + a.v + ")";
            break;
          case Rep::I64: r.v = "(i64)(0ull - (u64)" + a.v + ")"; break;
          case Rep::I128: r.v = "(i128)((u128)0 - (u128)" + a.v + ")"; break;
          case Rep::F32: case Rep::F64: r.v = "(-" + a.v + ")"; break;
          default: throw CometError("unary minus on boolean");
        }
        if (e.fail_on_error && a.t.is_integer()) {
          a = named(a);
          std::string mn = a.t.id == TypeId::Int64 ? "(i64)0x8000000000000000ull" : a.t.id == TypeId::Int32 ? "(i32)0x80000000" : a.t.id == TypeId::Int16 ? "-32768" : "-128";
          ErrSite site;      // negative.rs:136-150: arithmetic_overflow_error("byte" / "short" / "integer" / "long")
          site.error_type = "ArithmeticOverflow";
          site.error_class = "ARITHMETIC_OVERFLOW";
          site.from_type = a.t.id == TypeId::Int64 ? "long" : a.t.id == TypeId::Int32 ? "integer" : a.t.id == TypeId::Int16 ? "short" : "byte";
          site.value = ErrSite::NoValue;
          raise_value(and_ok(a.ok, a.v + " == " + mn), 1, site, "0");
        }
        return r;
      }
      case ExprKind::ScalarFunc: return scalar_func(e);
      case ExprKind::Like: return like(e);
      case ExprKind::RLike: return rlike(e);
      case ExprKind::If: {
        if (e.children.size() != 3) throw CometError("If needs three children");
        return select(named(gen(e.children[0])), named(gen(e.children[1])), named(gen(e.children[2])));
      }
      case ExprKind::CaseWhen: {
        // planner.rs:677-704 → DataFusion CaseExpr without base expression: the first WHEN that is TRUE (not NULL) picks its
        // THEN; no match → ELSE, or NULL without one.  Lowered to a chain of selects built from the last pair backwards.
        const size_t n = (size_t)e.n_when;
        if (n == 0 || (e.children.size() != 2 * n && e.children.size() != 2 * n + 1)) throw CometError("CaseWhen: when/then lists differ in length");
        std::vector<Val> thens;
        for (size_t i = 0; i < n; i++) thens.push_back(named(gen(e.children[n + i])));
        Val r;
        if (e.children.size() == 2 * n + 1) {
          r = named(gen(e.children[2 * n]));
        } else {
          r = thens[0];       // typed NULL: same representation, never valid
          r.ok = "false";
        }
        for (size_t i = n; i-- > 0;) r = named(select(named(gen(e.children[i])), thens[i], r));
        return r;
      }
      case ExprKind::In: {
        const bool direct = is_str_col(e.children.at(0));
        Val v = direct ? str_col_validity(e.children[0]->bound_index) : named(gen(e.children.at(0)));
        std::string any = "false", anynull = "false";
        for (size_t i = 1; i < e.children.size(); i++) {
          if (direct && is_str_lit(e.children[i])) {
            Val c = str_compare_lit(ExprKind::Eq, e.children[0]->bound_index, *e.children[i], false);
            any = "(" + any + " || " + c.v + ")";
            continue;
          }
          if (direct) throw CometError("IN over a Utf8 column expects Utf8 literals");
          Val li = gen(e.children[i]);
          Val c = compare(ExprKind::Eq, v, li);
          std::string lok = li.ok.empty() ? "true" : li.ok;
          any = "(" + any + " || (" + lok + " && " + c.v + "))";
          if (!li.ok.empty()) anynull = "(" + anynull + " || !" + li.ok + ")";
        }
        Val r;
        r.t = DType::of(TypeId::Bool);
        r.rep = Rep::B;
        std::string hit = newvar("bool");
        stmt(hit + " = " + any + ";");
        r.v = e.negated ? "(!" + hit + ")" : hit;
        std::string ok = v.ok;
        if (anynull != "false") ok = and_ok(ok, "(" + hit + " || !" + anynull + ")");
        r.ok = ok;
        return r;
      }
      case ExprKind::NormalizeNaNAndZero: {
        Val a = gen(e.children.at(0));
        Val r = a;
        if (a.rep == Rep::F64) r.v = "comet::normalize_nan_zero_f64(" + a.v + ")";
        else if (a.rep == Rep::F32) r.v = "comet::normalize_nan_zero_f32(" + a.v + ")";
        return r;
      }
      default:
        throw CometError(std::string("Expression ") + expr_name(e.proto_tag) + " (tag " + std::to_string(e.proto_tag) +
                         ") is not supported by the MI355X native engine");
    }
  }

  // Filter conjunct: rows survive only where the predicate is TRUE and valid
  void add_predicate(const ExprP& p) {
    Val v = gen(p);
    if (v.rep != Rep::B) throw CometError("Filter predicate must be boolean, got " + v.t.str());
    stmt("k[r] = " + and_ok(v.ok, v.v) + ";");
    next_stage();
    // isnotnull(<column>) as a conjunct of its own: every row that is still alive behind it holds a value — later references to the column read no validity
    // bit, and a projection that passes it through yields a column WITHOUT a validity buffer (no byte per row written, no pack launch; TPC-DS Q95's
    // scans are all of this shape: Filter(isnotnull(a) AND isnotnull(b)) → Project)
    if (p->kind == ExprKind::IsNotNull && p->children.size() == 1 && p->children[0]->kind == ExprKind::Bound) {
      auto it = col_cache.find(p->children[0]->bound_index);
      if (it != col_cache.end()) it->second.ok.clear();
      auto ck = cse.find(key_of(p->children[0]));
      if (ck != cse.end()) ck->second.ok.clear();
    }
  }

  // assemble the staged body: every stage is a load loop followed by a compute loop over the R rows
  // the stage-0 load loop alone (writes ld.xN), for the prefetch half of a pipelined tile
  std::string prefetch_body() const {
    std::string l = stages.front().loads, out;
    // "xN[r] = …" → "ld.xN[r] = …" (only at statement starts / inside the string-load block)
    size_t pos = 0;
    while (pos < l.size()) {
      size_t e = l.find('\n', pos);
      std::string line = l.substr(pos, e == std::string::npos ? std::string::npos : e - pos + 1);
      size_t a = line.find_first_not_of(' ');
      if (a != std::string::npos && line[a] == 'x') line.insert(a, "ld.");
      else {
        size_t b = line.find("; x");   // "{ bool tl_ = false; xN[r] = comet::ld_str16(…"
        if (b != std::string::npos) line.insert(b + 2, "ld.");
      }
      out += line;
      if (e == std::string::npos) break;
      pos = e + 1;
    }
    return "    _Pragma(\"unroll\") for (int r = 0; r < R; r++) if (k[r]) {\n" + out + "    }\n";
  }
  std::string body(const std::string& indent_unused = "", bool skip_front_loads = false) const {
    (void)indent_unused;
    std::string s;
    bool first = true;
    for (auto& st : stages) {
      const bool skip = skip_front_loads && first;
      first = false;
      if (!st.loads.empty() && !skip) {
        s += "    _Pragma(\"unroll\") for (int r = 0; r < R; r++) if (k[r]) {\n" + st.loads + "    }\n";
      }
      if (!st.body.empty()) {
        s += "    _Pragma(\"unroll\") for (int r = 0; r < R; r++) if (k[r]) {\n" + st.body + "    }\n";
      }
    }
    return s;
  }
};

// ---------------------------------------------------------------------------------------------
// Aggregate lowering: every aggregate function decomposes into commutative accumulator primitives
// (words of u64); identical primitives over the same (value, filter) are shared (sum(x) and avg(x)
// use one sum and one count).  The same lowering feeds the ungrouped template (registers) and the
// grouped template (per-group slots, atomics).
// ---------------------------------------------------------------------------------------------
enum class Prim { Cnt, RowCnt, Sum128, Sum192, SumI64, SumF64, AMaxHi, SignFlags, MinI64, MaxI64, MinI128, MaxI128, MinF64, MaxF64 };

struct PrimSlot {
  Prim prim;
  int word;      // first canonical accumulator word (or kernel-level word index when kernel_level)
  int nwords;
  bool kernel_level = false;
};

int prim_words(Prim p) {
  switch (p) {
    case Prim::Sum128: case Prim::MinI128: case Prim::MaxI128: return 2;
    case Prim::Sum192: return 3;
    default: return 1;
  }
}

int bit_length_u128(u128 v) {
  int b = 0;
  while (v) { b++; v >>= 1; }
  return b;
}

constexpr int kLimbBitsHost = 43;  // must equal comet::kLimbBits

struct AggLowering {
  Gen& g;
  bool grouped;
  int nw = 0;    // canonical words
  int npw = 0;   // grouped: private (limb-form) words per group
  int nkw = 0;   // grouped: kernel-level words
  std::map<std::string, PrimSlot> slots;   // key: prim|valuekey|filterkey
  std::string init_code, combine_code;
  std::vector<std::string> gops, gident;   // canonical: per-word GOp name and identity literal (global-table atomics)
  std::vector<std::string> pops, pident;   // grouped: private words
  std::vector<std::string> kops;           // grouped: kernel-level words
  std::string pv_code;                     // grouped: per-row limb contributions (pv[j] = …)
  std::string fold_code;                   // grouped: limb words pw[] → canonical contribution val[]
  std::string kfeed_code;                  // grouped: per-row kernel-level updates

  AggLowering(Gen& gen, bool grp) : g(gen), grouped(grp) {}

  // ---- exact Float64 sum: 192-bit fixed-point sum + class word per group; exponent range per kernel ----
  struct FSum { int word; int cls; int fidx; };
  std::map<std::string, FSum> fsums;
  std::vector<PipelineDesc::FixSum> fix_sums;
  std::string kexport_code;   // ungrouped: thread 0 of each block publishes the exponent words with atomic max
  static std::string fscale(int fidx) {
    return "comet::fix_scale(prm.iarg[" + std::to_string(fidx < 4 ? kFixScaleArg : kFixScaleArg2) + "], " + std::to_string(fidx & 3) + ")";
  }
  std::string fread(const FSum& f) const {
    return "comet::fix192_to_f64(acc + " + std::to_string(f.word) + ", " + fscale(f.fidx) + ", acc[" + std::to_string(f.cls) + "])";
  }
  FSum get_fsum(const std::string& vkey, const std::string& fkey, const std::string& cond, const std::string& x) {
    const std::string key = "fsum|" + vkey + "|" + fkey;
    auto it = fsums.find(key);
    if (it != fsums.end()) return it->second;
    if ((int)fix_sums.size() >= kFixMaxSums) throw CometError("more than " + std::to_string(kFixMaxSums) + " distinct Float64 sums / averages in one aggregate are not supported by the GPU pipeline yet");
    const int fidx = (int)fix_sums.size();
    const std::string c = cond.empty() ? "true" : cond;
    const std::string xd = "(double)(" + x + ")", S = fscale(fidx);
    FSum f;
    f.fidx = fidx;
    f.word = nw;
    nw += 3;
    f.cls = nw;
    nw += 1;
    const std::string w = std::to_string(f.word), w1 = std::to_string(f.word + 1), w2 = std::to_string(f.word + 2), cw = std::to_string(f.cls);
    init_code += "    a[" + w + "] = 0; a[" + w1 + "] = 0; a[" + w2 + "] = 0; a[" + cw + "] = 0;\n";
    combine_code += "    comet::acc_add192(a + " + w + ", b + " + w + "); comet::acc_or64(a + " + cw + ", b + " + cw + ");\n";
    for (auto* q : {"G_ADD192", "G_CONT", "G_CONT", "G_OR64"}) gops.push_back(q);
    for (int k = 0; k < 4; k++) gident.push_back("0ull");
    PipelineDesc::FixSum fs;
    fs.word = f.word;
    if (grouped) {
      int j0 = -1;
      for (int t = 0; t < 4; t++) {
        int j = pword("G_ADD64", "0ull");
        if (t == 0) j0 = j;
        pv_code += "        pv[" + std::to_string(j) + "] = (" + c + ") ? comet::f64_fix_limb(" + xd + ", " + S + ", " + std::to_string(t) + ") : 0ull;\n";
      }
      int jc = pword("G_OR64", "0ull");
      pv_code += "        pv[" + std::to_string(jc) + "] = (" + c + ") ? comet::f64_class(" + xd + ") : 0ull;\n";
      fold_code += "    { u64 t3_[3]; comet::limbs_to_i192(pw + " + std::to_string(j0) + ", 4, t3_); val[" + w + "] = t3_[0]; val[" + w1 + "] = t3_[1]; val[" + w2 +
                   "] = t3_[2]; val[" + cw + "] = pw[" + std::to_string(jc) + "]; }\n";
      fs.aux_hi = nkw++;
      fs.aux_lo = nkw++;
      kops.push_back("G_UMAX64");
      kops.push_back("G_UMAX64");
      kfeed_code += "        if (" + c + ") { u64 h_ = comet::f64_exp_hi(" + xd + "), l_ = comet::f64_exp_lo(" + xd + "); if (h_ > kacc[" + std::to_string(fs.aux_hi) +
                    "]) kacc[" + std::to_string(fs.aux_hi) + "] = h_; if (l_ > kacc[" + std::to_string(fs.aux_lo) + "]) kacc[" + std::to_string(fs.aux_lo) + "] = l_; }\n";
    } else {
      const int hw = nw, lw = nw + 1;
      nw += 2;
      const std::string H = std::to_string(hw), L = std::to_string(lw);
      init_code += "    a[" + H + "] = 0; a[" + L + "] = 0;\n";
      combine_code += "    if (b[" + H + "] > a[" + H + "]) a[" + H + "] = b[" + H + "]; if (b[" + L + "] > a[" + L + "]) a[" + L + "] = b[" + L + "];\n";
      gops.push_back("G_UMAX64"); gops.push_back("G_UMAX64");
      gident.push_back("0ull"); gident.push_back("0ull");
      const std::string body = "{ const double xd_ = " + xd + "; comet::acc_feed_fix192(acc + " + w + ", xd_, " + S + "); acc[" + cw + "] |= comet::f64_class(xd_); " +
                               "const u64 h_ = comet::f64_exp_hi(xd_), l_ = comet::f64_exp_lo(xd_); if (h_ > acc[" + H + "]) acc[" + H + "] = h_; if (l_ > acc[" + L + "]) acc[" + L + "] = l_; }";
      g.stmt(cond.empty() ? body : "if (" + cond + ") " + body);
      fs.aux_hi = 2 * fidx;
      fs.aux_lo = 2 * fidx + 1;
      kexport_code += "    atomicMax(aux + " + std::to_string(fs.aux_hi) + ", (unsigned long long)acc[" + H + "]); atomicMax(aux + " + std::to_string(fs.aux_lo) +
                      ", (unsigned long long)acc[" + L + "]);\n";
    }
    fix_sums.push_back(fs);
    fsums[key] = f;
    return f;
  }

  int pword(const char* op, const char* ident) {
    pops.push_back(op);
    pident.push_back(ident);
    return npw++;
  }

  // cond: row contributes iff cond (empty = always); x: value expression typed for the primitive;
  // maxabs: static bound on |x| (integer sums), used to size the limb split
  PrimSlot get(Prim p, const std::string& vkey, const std::string& fkey, const std::string& cond, const std::string& x,
               u128 maxabs = kUnbounded) {
    std::string key = std::to_string((int)p) + "|" + vkey + "|" + fkey;
    if (p == Prim::Cnt && cond.empty()) key = std::to_string((int)Prim::RowCnt) + "|*|";  // counts every row: share
    auto it = slots.find(key);
    if (it != slots.end()) return it->second;
    const std::string c = cond.empty() ? "true" : cond;
    auto feed = [&](const std::string& body) { g.stmt(cond.empty() ? body : "if (" + cond + ") { " + body + " }"); };
    if (grouped && (p == Prim::AMaxHi || p == Prim::SignFlags)) {
      // value bounds for the overflow proof are tracked per KERNEL, not per group (a bound for all groups is a
      // bound for each): two registers per thread instead of two LDS atomics per row
      PrimSlot s{p, nkw, 1, true};
      slots[key] = s;
      const std::string kk = std::to_string(nkw++);
      if (p == Prim::AMaxHi) {
        kops.push_back("G_UMAX64");
        kfeed_code += "        if (" + c + ") { u64 t_ = comet::amax_enc(comet::uabs128(" + x + ")); if (t_ > kacc[" + kk + "]) kacc[" + kk + "] = t_; }\n";
      } else {
        kops.push_back("G_OR64");
        kfeed_code += "        if (" + c + ") kacc[" + kk + "] |= ((" + x + ") < 0) ? 2ull : (((" + x + ") > 0) ? 1ull : 0ull);\n";
      }
      return s;
    }
    PrimSlot s{p, nw, prim_words(p)};
    nw += s.nwords;
    slots[key] = s;
    const std::string w = std::to_string(s.word), w1 = std::to_string(s.word + 1), w2 = std::to_string(s.word + 2);
    auto ops = [&](std::initializer_list<const char*> o, std::initializer_list<const char*> id) {
      for (auto* q : o) gops.push_back(q);
      for (auto* q : id) gident.push_back(q);
    };
    // grouped integer sum: split into limbs, fold back into `nlimb_words` canonical words
    auto grouped_int_sum = [&](const std::string& x128, u128 bound, int canon_words) {
      int bits = (bound == kUnbounded ? 127 : bit_length_u128(bound)) + 1;
      int nl = (bits + kLimbBitsHost - 1) / kLimbBitsHost;
      if (nl < 1) nl = 1;
      int j0 = -1;
      for (int t = 0; t < nl; t++) {
        int j = pword("G_ADD64", "0ull");
        if (t == 0) j0 = j;
        pv_code += "        pv[" + std::to_string(j) + "] = (" + c + ") ? comet::limb_of(" + x128 + ", " + std::to_string(t) + ", " + std::to_string(nl) + ") : 0ull;\n";
      }
      fold_code += "    { u64 t3_[3]; comet::limbs_to_i192(pw + " + std::to_string(j0) + ", " + std::to_string(nl) + ", t3_);";
      for (int k = 0; k < canon_words; k++) fold_code += " val[" + std::to_string(s.word + k) + "] = t3_[" + std::to_string(k) + "];";
      fold_code += " }\n";
    };
    switch (p) {
      case Prim::Cnt: case Prim::RowCnt:
        init_code += "    a[" + w + "] = 0;\n";
        combine_code += "    comet::acc_add64(a + " + w + ", b + " + w + ");\n";
        ops({"G_ADD64"}, {"0ull"});
        if (grouped) {
          int j = pword("G_ADD64", "0ull");
          pv_code += "        pv[" + std::to_string(j) + "] = (" + c + ") ? 1ull : 0ull;\n";
          fold_code += "    val[" + w + "] = pw[" + std::to_string(j) + "];\n";
        } else feed("acc[" + w + "] += 1;");
        break;
      case Prim::SumI64:
        init_code += "    a[" + w + "] = 0;\n";
        combine_code += "    comet::acc_add64(a + " + w + ", b + " + w + ");\n";
        ops({"G_ADD64"}, {"0ull"});
        if (grouped) grouped_int_sum("(i128)(i64)(" + x + ")", maxabs == kUnbounded ? ((u128)1 << 63) : maxabs, 1);  // wraps mod 2^64 like add_wrapping
        else feed("acc[" + w + "] += (u64)(i64)(" + x + ");");
        break;
      case Prim::SumF64:
        init_code += "    a[" + w + "] = 0;\n";
        combine_code += "    comet::acc_fadd64(a + " + w + ", b + " + w + ");\n";
        ops({"G_FADD64"}, {"0ull"});
        if (grouped) {
          int j = pword("G_FADD64", "0ull");
          pv_code += "        pv[" + std::to_string(j) + "] = (" + c + ") ? (u64)__double_as_longlong((double)(" + x + ")) : 0ull;\n";
          fold_code += "    val[" + w + "] = pw[" + std::to_string(j) + "];\n";
        } else feed("acc[" + w + "] = (u64)__double_as_longlong(comet::fp_add(__longlong_as_double((i64)acc[" + w + "]), (double)(" + x + ")));");
        break;
      case Prim::Sum128:
        init_code += "    a[" + w + "] = 0; a[" + w1 + "] = 0;\n";
        combine_code += "    comet::acc_add128(a + " + w + ", b + " + w + ");\n";
        ops({"G_ADD128", "G_CONT"}, {"0ull", "0ull"});
        if (grouped) grouped_int_sum(x, maxabs, 2);
        else feed("comet::acc_feed_i128(acc + " + w + ", " + x + ");");
        break;
      case Prim::Sum192:
        init_code += "    a[" + w + "] = 0; a[" + w1 + "] = 0; a[" + w2 + "] = 0;\n";
        combine_code += "    comet::acc_add192(a + " + w + ", b + " + w + ");\n";
        ops({"G_ADD192", "G_CONT", "G_CONT"}, {"0ull", "0ull", "0ull"});
        if (grouped) grouped_int_sum(x, maxabs, 3);
        else feed("comet::acc_feed_i192(acc + " + w + ", " + x + ");");
        break;
      case Prim::AMaxHi:
        // amax_enc(|v|) (0 = no value yet): a monotone one-word upper bound of max|v|
        init_code += "    a[" + w + "] = 0;\n";
        combine_code += "    if (b[" + w + "] > a[" + w + "]) a[" + w + "] = b[" + w + "];\n";
        ops({"G_UMAX64"}, {"0ull"});
        feed("{ u64 t_ = comet::amax_enc(comet::uabs128(" + x + ")); if (t_ > acc[" + w + "]) acc[" + w + "] = t_; }");
        break;
      case Prim::SignFlags:
        init_code += "    a[" + w + "] = 0;\n";
        combine_code += "    comet::acc_or64(a + " + w + ", b + " + w + ");\n";
        ops({"G_OR64"}, {"0ull"});
        feed("acc[" + w + "] |= ((" + x + ") < 0) ? 2ull : (((" + x + ") > 0) ? 1ull : 0ull);");
        break;
      case Prim::MinI64: case Prim::MaxI64: {
        const bool mn = p == Prim::MinI64;
        const char* id = mn ? "0x7fffffffffffffffull" : "0x8000000000000000ull";
        init_code += "    a[" + w + "] = " + id + ";\n";
        combine_code += std::string("    comet::acc_") + (mn ? "imin64" : "imax64") + "(a + " + w + ", b + " + w + ");\n";
        ops({mn ? "G_IMIN64" : "G_IMAX64"}, {id});
        if (grouped) {
          int j = pword(mn ? "G_IMIN64" : "G_IMAX64", id);
          pv_code += "        pv[" + std::to_string(j) + "] = (" + c + ") ? (u64)(i64)(" + x + ") : " + id + ";\n";
          fold_code += "    val[" + w + "] = pw[" + std::to_string(j) + "];\n";
        } else feed(std::string("{ u64 t_ = (u64)(i64)(") + x + "); comet::acc_" + (mn ? "imin64" : "imax64") + "(acc + " + w + ", &t_); }");
        break;
      }
      case Prim::MinF64: case Prim::MaxF64: {
        const bool mn = p == Prim::MinF64;
        const char* id = mn ? "0x7fffffffffffffffull" : "0xffffffffffffffffull";  // extremes of the IEEE total order (as bits)
        init_code += "    a[" + w + "] = " + id + ";\n";
        combine_code += std::string("    comet::acc_") + (mn ? "fmin64" : "fmax64") + "(a + " + w + ", b + " + w + ");\n";
        ops({mn ? "G_FMIN64" : "G_FMAX64"}, {id});
        if (grouped) {
          // in LDS: integer min/max on the total-order key (an involution of the bit pattern)
          const char* kid = mn ? "0x7fffffffffffffffull" : "0x8000000000000000ull";
          int j = pword(mn ? "G_IMIN64" : "G_IMAX64", kid);
          pv_code += "        pv[" + std::to_string(j) + "] = (" + c + ") ? (u64)comet::f64_total_key((double)(" + x + ")) : " + kid + ";\n";
          fold_code += "    val[" + w + "] = (u64)comet::f64_total_key(__longlong_as_double((i64)pw[" + std::to_string(j) + "]));\n";
        } else feed(std::string("{ u64 t_ = (u64)__double_as_longlong((double)(") + x + ")); comet::acc_" + (mn ? "fmin64" : "fmax64") + "(acc + " + w + ", &t_); }");
        break;
      }
      case Prim::MinI128: case Prim::MaxI128: {
        const bool mn = p == Prim::MinI128;
        if (grouped) throw CometError("min/max of a decimal wider than 18 digits is not supported in a grouped GPU aggregate yet");
        init_code += mn ? "    a[" + w + "] = ~0ull; a[" + w1 + "] = 0x7fffffffffffffffull;\n" : "    a[" + w + "] = 0; a[" + w1 + "] = 0x8000000000000000ull;\n";
        combine_code += std::string("    comet::acc_") + (mn ? "imin128" : "imax128") + "(a + " + w + ", b + " + w + ");\n";
        ops({"G_CONT", "G_CONT"}, {"0ull", "0ull"});
        feed("{ u64 t_[2] = {comet::lo64(" + x + "), comet::hi64(" + x + ")}; comet::acc_" + std::string(mn ? "imin128" : "imax128") + "(acc + " + w + ", t_); }");
        break;
      }
    }
    return s;
  }
};

std::string explain_expr(const ExprP& e) {
  std::string s = expr_name(e->proto_tag);
  if (e->kind == ExprKind::Bound) return "col" + std::to_string(e->bound_index);
  if (e->kind == ExprKind::Literal) return "lit:" + e->dtype.str();
  s += "(";
  for (size_t i = 0; i < e->children.size(); i++) s += (i ? ", " : "") + explain_expr(e->children[i]);
  return s + ")";
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// generate_pipeline
// ---------------------------------------------------------------------------------------------
PipelineDesc generate_pipeline(const Operator& root, const std::vector<bool>& in_has_validity, const std::vector<DType>* source_types,
                               const std::vector<int>* str_fixed_len, const std::vector<int>* dict_id_col) {
  // 1. walk root → leaf collecting the chain; the chain ends at a Scan or at a materialised source (a join's output)
  std::vector<const Operator*> chain;
  const Operator* cur = &root;
  while (true) {
    chain.push_back(cur);
    if (cur->kind == OpKind::Scan) break;
    // materialised sources: joins, Parquet scans, sorts, limits — and an aggregate BELOW other operators (nothing fuses
    // across a pipeline breaker: its result is materialised in HBM and read like a scan)
    if (cur->kind == OpKind::Explode || cur->kind == OpKind::HashJoin || cur->kind == OpKind::NativeScan || cur->kind == OpKind::Sort || cur->kind == OpKind::Limit || cur->kind == OpKind::Expand || cur->kind == OpKind::Window ||
        (cur->kind == OpKind::HashAgg && cur != &root)) {
      if (!source_types) throw CometError("internal: materialised source without a schema");
      break;
    }
    if (cur->kind == OpKind::Unsupported)
      throw CometError(std::string("Operator ") + op_name(cur->proto_tag) + " is not supported by the MI355X native engine");
    if (cur->kind != OpKind::Filter && cur->kind != OpKind::Projection && cur->kind != OpKind::HashAgg)
      throw CometError(std::string("Operator ") + op_name(cur->proto_tag) + " is not supported in a fused GPU pipeline yet");
    if (cur->children.size() != 1) throw CometError(std::string(op_name(cur->proto_tag)) + " expects exactly one child");
    cur = cur->children[0].get();
  }
  const Operator& scan = *chain.back();
  PipelineDesc d;
  SiteScope site_scope(d);
  d.in_types = source_types ? *source_types : scan.scan_fields;
  for (auto* op : chain) d.op_names.push_back(op_name(op->proto_tag));
  if (in_has_validity.size() != d.in_types.size()) throw CometError("internal: validity mask arity mismatch");
  if (d.in_types.size() > COMET_MAX_IN) throw CometError("too many scan columns for one GPU pipeline");

  // 2. fold leaf → root
  std::vector<ExprP> cols;
  for (size_t i = 0; i < d.in_types.size(); i++) {
    auto b = std::make_shared<Expr>();
    b->kind = ExprKind::Bound;
    b->proto_tag = 3;
    b->bound_index = (int)i;
    b->dtype = d.in_types[i];
    b->has_dtype = true;
    cols.push_back(b);
  }
  const std::vector<ExprP> source_cols = cols;
  struct SourceColsScope { SourceColsScope(const std::vector<ExprP>* c) { g_source_cols = c; } ~SourceColsScope() { g_source_cols = nullptr; } } source_cols_scope(&source_cols);
  struct DerivedScope { DerivedScope(std::vector<DerivedCol>* c) { g_derived = c; } ~DerivedScope() { g_derived = nullptr; } } derived_scope(&d.derived);
  std::vector<ExprP> preds;
  const Operator* agg = nullptr;
  std::vector<ExprP> group_exprs;
  struct AggIn { const AggExpr* a; std::vector<ExprP> children; ExprP filter; AggMode mode = AggMode::Partial; };
  std::vector<AggIn> agg_ins;
  size_t final_state_pos = 0;
  for (int i = (int)chain.size() - 2; i >= 0; i--) {
    const Operator& op = *chain[i];
    std::map<const Expr*, ExprP> memo;
    if (agg) throw CometError("Operators above a HashAggregate in the same native plan are not supported yet");
    if (op.kind == OpKind::Filter) {
      if (!op.predicate) throw CometError("Filter without predicate");
      split_conjuncts(substitute(op.predicate, cols, memo), preds);
      d.has_filter = true;
    } else if (op.kind == OpKind::Projection) {
      std::vector<ExprP> nc;
      for (auto& e : op.project_list) nc.push_back(substitute(e, cols, memo));
      cols = nc;
    } else if (op.kind == OpKind::HashAgg) {
      agg = &op;
      for (auto& e : op.grouping_exprs) group_exprs.push_back(substitute(e, cols, memo));
      // per-expression modes (HashAggregate.expr_modes, planner.rs:1274-1345): a Partial-mode operator may carry PartialMerge
      // expressions — the count(DISTINCT) rewrite merges the other aggregates' states while it starts counting — whose state
      // columns sit in the child's output from initial_input_buffer_offset on
      if (!op.expr_modes.empty() && op.expr_modes.size() != op.agg_exprs.size())
        throw CometError("HashAggregate: expr_modes has " + std::to_string(op.expr_modes.size()) + " entries for " + std::to_string(op.agg_exprs.size()) + " aggregates");
      const size_t state_base = op.expr_modes.empty() ? op.grouping_exprs.size() : (size_t)std::max(0, op.initial_input_buffer_offset);
      for (size_t ai = 0; ai < op.agg_exprs.size(); ai++) {
        const AggExpr& a = op.agg_exprs[ai];
        AggIn in;
        in.a = &a;
        in.mode = op.expr_modes.empty() ? op.agg_mode : (AggMode)op.expr_modes[ai];
        if (ai == 0) d.merges_states = in.mode != AggMode::Partial;
        else d.merges_states = d.merges_states && in.mode != AggMode::Partial;
        if (in.mode == AggMode::Partial) {
          for (auto& c : a.children) in.children.push_back(substitute(c, cols, memo));
          if (a.filter) in.filter = substitute(a.filter, cols, memo);
        } else if (in.mode == AggMode::Final || in.mode == AggMode::PartialMerge) {
          // Final / PartialMerge: the aggregate's inputs are the Partial state columns that follow the group columns, in order
          // (AggregateExec Final mode; the serialized children are unbound, operators.scala:1786-1792)
          int arity = 1;
          if (a.kind == AggKind::Avg) arity = 2;
          if (a.kind == AggKind::Sum && a.dtype.id == TypeId::Decimal) arity = 2;
          for (int k = 0; k < arity; k++) {
            size_t idx = state_base + final_state_pos++;
            if (idx >= cols.size()) throw CometError("Final aggregate: state column " + std::to_string(idx) + " is out of bound");
            in.children.push_back(cols[idx]);
          }
        } else {
          throw CometError("Unsupported aggregate mode: " + std::to_string((int)in.mode));
        }
        agg_ins.push_back(in);
      }
    }
  }

  // the derived columns follow the source's: a split is NULL where its subject is
  std::vector<bool> valid_all = in_has_validity;
  for (auto& dc : d.derived) {
    d.in_types.push_back(dc.type);
    valid_all.push_back(in_has_validity[(size_t)dc.src]);
    if (dc.kind == 3) continue;
    DType et = dc.type.kids[0];      // … and its element column behind it, addressable like the element column of a source list (list_ref)
    et.virt_parent = (int)d.in_types.size() - 1;
    et.virt_kid = 0;
    d.in_types.push_back(et);
    valid_all.push_back(false);
  }
  if (!d.derived.empty() && agg) throw CometError("split inside an aggregate's chain is not supported by the MI355X native engine yet");
  if (d.in_types.size() > COMET_MAX_IN) throw CometError("too many scan columns for one GPU pipeline");
  const std::vector<bool>& in_has_validity_all = valid_all;
  Gen g(d.in_types, in_has_validity_all);
  if (str_fixed_len) g.str_fixed_len = *str_fixed_len;
  if (const char* e = getenv("COMET_GEN_EAGER")) g.eager_loads = atoi(e) != 0;
  g.pipelined = agg != nullptr;   // aggregate sinks prefetch tile t+1's first-stage columns while computing tile t
  if (const char* e = getenv("COMET_GEN_PIPELINE")) g.pipelined = g.pipelined && atoi(e) != 0;
  for (auto& p : preds) g.add_predicate(p);

  std::ostringstream src;
  src << "// generated by datafusion-comet_amd codegen — fused pipeline: ";
  for (auto& n : d.op_names) src << n << " <- ";
  src << "input\n";
  if (const char* e = getenv("COMET_EXPERIMENT")) src << "#define COMET_EXPERIMENT " << atoi(e) << "\n";
  // An aggregate sink reads every column byte once and keeps nothing but its accumulators: its column loads are non-temporal (streaming)
  // loads — SF100 Q1's k_gagg 7.24-7.58 -> 7.01-7.13 ms in three alternating pairs on one box.  Join probes keep ordinary loads (their
  // bitmap and build rows want the L2; both of SF100 Q3's got ~3 % slower with streaming loads).
  if (agg) src << "#ifndef COMET_LD_NT\n#define COMET_LD_NT 1\n#endif\n";
  src << "#include \"comet_device.hpp\"\nusing namespace comet;\n";

  std::ostringstream ex;
  for (auto& p : preds) ex << "  filter: " << explain_expr(p) << "\n";

  if (!agg) {
    // ---------------- Output sink ----------------
    d.sink = SinkKind::Output;
    std::vector<Val> outs;
    // outputs are evaluated in the emit kernel (per surviving row); predicates in the mask kernel.
    Gen ge(d.in_types, in_has_validity_all);
    if (str_fixed_len) ge.str_fixed_len = *str_fixed_len;
    for (auto& c : cols) {
      // (… and so is a nested column: its rows are gathered by the executor, children and all — exec.cpp take_nested)
      const bool is_str_type = c->kind == ExprKind::Bound && c->bound_index >= 0 && (size_t)c->bound_index < d.in_types.size() &&
                               (d.in_types[(size_t)c->bound_index].id == TypeId::String || d.in_types[(size_t)c->bound_index].id == TypeId::Bytes ||
                                d.in_types[(size_t)c->bound_index].is_nested());
      if (c->kind == ExprKind::ListExtract && !c->children.empty() && c->children[0]->kind == ExprKind::Bound && c->children[0]->bound_index >= 0 &&
          (size_t)c->children[0]->bound_index < d.in_types.size() && d.in_types[(size_t)c->children[0]->bound_index].id == TypeId::List &&
          !d.in_types[(size_t)c->children[0]->bound_index].kids.empty() && d.in_types[(size_t)c->children[0]->bound_index].kids[0].id == TypeId::String) {
        // an element of a list of strings (split(s, ',')[0], element_at(arr, -1)): the ELEMENT column's row travels, the executor gathers the string
        Gen::ListRef l;
        std::string row, hit;
        ge.list_extract_pos(*c, l, row, hit);
        Val v;
        v.t = DType::of(TypeId::String);
        v.rep = Rep::I64;
        v.v = row;
        v.ok = in_has_validity_all[(size_t)l.elem] ? "(" + hit + " && comet::ld_valid(" + l.ecol + ", " + row + "))" : hit;
        v = ge.named(v);
        outs.push_back(v);
        OutCol oc;
        oc.type = v.t;
        oc.nullable = true;
        oc.gather_src = l.elem;
        d.out_cols.push_back(oc);
        ex << "  output: " << explain_expr(c) << " : Utf8 (an element gathered from column " << l.elem << ")\n";
        continue;
      }
      if (is_str_type) {
        // a Utf8 column passed through: emit the source row index, the executor gathers the string afterwards
        const int src = c->bound_index;
        ge.in_used[(size_t)src] = true;
        Val v;
        v.t = d.in_types[(size_t)src];
        v.rep = Rep::I64;
        v.v = "idx[r]";
        if (in_has_validity_all[(size_t)src]) v.ok = "comet::ld_valid(prm.in[" + std::to_string(ge.locate(src).first) + "], idx[r])";
        v = ge.named(v);
        outs.push_back(v);
        OutCol oc;
        oc.type = v.t;
        oc.type.virt_parent = oc.type.virt_kid = -1;
        oc.nullable = !v.ok.empty();
        oc.gather_src = src;
        d.out_cols.push_back(oc);
        ex << "  output: " << explain_expr(c) << " : " << v.t.str() << " (gathered)\n";
        continue;
      }
      {
        // a string function of a Utf8 COLUMN with literal arguments whose result is a slice of the value plus padding: any length
        OutCol voc;
        Val vv;
        if (ge.string_view(*c, vv, voc)) {
          outs.push_back(vv);
          voc.type = vv.t;
          voc.nullable = !vv.ok.empty();
          d.out_cols.push_back(voc);
          ex << "  output: " << explain_expr(c) << " : " << vv.t.str() << " (string view of column " << voc.view_src << ")\n";
          continue;
        }
      }
      {
        // concat of Utf8 columns and literals: the source row travels, the executor assembles the column
        OutCol coc;
        Val cvl;
        if (ge.string_concat(*c, cvl, coc)) {
          outs.push_back(cvl);
          coc.type = DType::of(TypeId::String);
          coc.nullable = !cvl.ok.empty();
          d.out_cols.push_back(coc);
          ex << "  output: " << explain_expr(c) << " : string (concatenation of " << coc.concat_cols.size() << " parts)\n";
          continue;
        }
      }
      {
        // Cast(<integer | boolean | decimal | date | timestamp> AS STRING): the value travels, the executor writes the digits
        OutCol foc;
        Val fv;
        if (ge.string_format(*c, fv, foc)) {
          outs.push_back(fv);
          foc.type = DType::of(TypeId::String);
          foc.nullable = !fv.ok.empty();
          d.out_cols.push_back(foc);
          ex << "  output: " << explain_expr(c) << " : string (formatted value)\n";
          continue;
        }
      }
      Val v = ge.named(ge.gen(c));
      outs.push_back(v);
      OutCol oc;
      oc.type = v.t;
      oc.nullable = !v.ok.empty();
      // a computed string (literal, substring, CASE over those — at most 15 bytes): stored packed, expanded to offsets + bytes by the executor
      oc.packed_string = v.rep == Rep::STR;
      d.out_cols.push_back(oc);
      ex << "  output: " << explain_expr(c) << " : " << v.t.str() << "\n";
    }
    if (d.out_cols.size() * 2 + kOutFirstCol > COMET_MAX_OUT) throw CometError("too many output columns for one GPU pipeline");
    for (size_t j = 0; j < outs.size(); j++) {
      const Val& v = outs[j];
      std::string vb = "prm.out[" + std::to_string(kOutFirstCol + 2 * j) + "]";
      std::string ob = "prm.out[" + std::to_string(kOutFirstCol + 2 * j + 1) + "]";
      if (d.out_cols[j].fmt_kind) {
        ge.stmt("((i128*)" + vb + ")[pos[r]] = (i128)" + v.v + ";");
        if (!v.ok.empty()) ge.stmt("((u8*)" + ob + ")[pos[r]] = " + v.ok + " ? 1 : 0;");
        continue;
      }
      if (d.out_cols[j].view_src >= 0) {
        ge.stmt("((comet::strview*)" + vb + ")[pos[r]] = " + v.v + ";");
        if (!v.ok.empty()) ge.stmt("((u8*)" + ob + ")[pos[r]] = " + v.ok + " ? 1 : 0;");
        continue;
      }
      if (d.out_cols[j].gather_src >= 0 || !d.out_cols[j].concat_cols.empty()) {
        ge.stmt("((u32*)" + vb + ")[pos[r]] = (u32)" + v.v + ";");
        if (!v.ok.empty()) ge.stmt("((u8*)" + ob + ")[pos[r]] = " + v.ok + " ? 1 : 0;");
        continue;
      }
      if (d.out_cols[j].packed_string) {
        ge.stmt("((comet::str16*)" + vb + ")[pos[r]] = " + (v.ok.empty() ? v.v : "(" + v.ok + " ? " + v.v + " : comet::str16{0ull, 0ull})") + ";");
        if (!v.ok.empty()) ge.stmt("((u8*)" + ob + ")[pos[r]] = " + v.ok + " ? 1 : 0;");
        continue;
      }
      const char* st = store_ctype(v.t);
      std::string val = v.v;
      if (v.t.id == TypeId::Decimal) val = v.rep == Rep::I128 ? v.v : "(i128)" + v.v;
      else if (v.t.id == TypeId::Bool) val = "(u8)(" + v.v + " ? 1 : 0)";
      else val = std::string("(") + st + ")" + v.v;
      // NULL slots are written as zero like arrow builders do (deterministic bytes)
      if (!v.ok.empty()) {
        ge.stmt("((" + std::string(st) + "*)" + vb + ")[pos[r]] = " + v.ok + " ? " + val + " : (" + st + ")0;");
        ge.stmt("((u8*)" + ob + ")[pos[r]] = " + v.ok + " ? 1 : 0;");
      } else {
        ge.stmt("((" + std::string(st) + "*)" + vb + ")[pos[r]] = " + val + ";");
      }
    }
    for (size_t i = 0; i < d.in_types.size(); i++) d.in_used.push_back(g.in_used[i] || ge.in_used[i]);
    if (d.has_filter) {
      // tile-wise functors for the single-pass filter kernel: R row slots per thread; every stage first issues the loads of all R
      // rows, then computes / stores, so a thread keeps R × (columns) loads in flight instead of one row's
      int fr = 8;
      if (const char* e = getenv("COMET_EXPERIMENT")) fr = std::max(1, std::min(16, atoi(e)));
      d.R = fr;
      src << "struct P {\n  static constexpr int R = " << fr << ";\n";
      src << "  static __device__ __forceinline__ void keep_tile(const CometKParams& prm, i64 base, i64 n, bool* k) {\n"
          << "    i64 idx[R];\n    _Pragma(\"unroll\") for (int r = 0; r < R; r++) { idx[r] = base + (i64)r * comet::kBlock + threadIdx.x; k[r] = idx[r] < n; }\n"
          << g.decls << g.body() << "  }\n";
      src << "  static __device__ __forceinline__ void emit_tile(const CometKParams& prm, const bool* k, const i64* idx, const i64* pos) {\n"
          << ge.decls << ge.body() << "  }\n};\n";
      src << "extern \"C\" __global__ __launch_bounds__(256) void k_filter(const CometKParams prm) { comet::filter_fused_body<P>(prm); }\n";
      d.kernels = {"k_filter"};
    } else {
      d.R = 1;
      src << "struct P {\n  static constexpr int R = 1;\n";
      src << "  static __device__ __forceinline__ void emit(const CometKParams& prm, i64 i, i64 p) {\n"
          << "    bool k[R] = {true}; i64 idx[R] = {i}; i64 pos[R] = {p};\n"
          << ge.decls << ge.body() << "  }\n};\n";
      src << "extern \"C\" __global__ __launch_bounds__(256) void k_emit(const CometKParams prm) { comet::project_body<P>(prm); }\n";
      d.kernels = {"k_emit"};
    }
    src << "extern \"C\" __global__ __launch_bounds__(256) void k_pack(const CometKParams prm) { comet::pack_validity_body((const u8*)prm.out[0], (u8*)prm.out[1], prm.n); }\n";
    d.kernels.push_back("k_pack");
    d.source = with_optional_headers(src.str());
    d.explain = ex.str();
    return d;
  }

  // ---------------- Aggregate sinks ----------------
  // Final and PartialMerge both MERGE Partial states (merge_batch); Final then evaluates, PartialMerge re-emits the state
  const bool grouped = !group_exprs.empty();
  d.sink = grouped ? SinkKind::AggGrouped : SinkKind::AggNoGroup;
  // the context an ANSI decimal sum (0) / average (1) raises its DecimalSumOverflow with (PipelineDesc::agg_ctx)
  auto note_agg_ctx = [&](int kind, const AggExpr& a) {
    std::shared_ptr<QueryContext> c = (a.qctx && a.has_expr_id) ? a.qctx : nullptr;
    static const std::shared_ptr<QueryContext> none;
    if (d.agg_ctx_mixed[kind]) return;
    if (!d.agg_ctx[kind] && !d.agg_ctx_seen[kind]) { d.agg_ctx[kind] = c; d.agg_ctx_seen[kind] = true; return; }
    const QueryContext* x = d.agg_ctx[kind].get();
    const bool same = (!x && !c) || (x && c && x->sql_text == c->sql_text && x->start_index == c->start_index && x->stop_index == c->stop_index);
    if (!same) { d.agg_ctx_mixed[kind] = true; d.agg_ctx[kind] = nullptr; }
  };
  AggLowering al(g, grouped);
  std::string fin;  // finalize body; ROW is "[0]" (ungrouped) or "[pos]" (grouped emit)
  const std::string ROW = grouped ? "[pos]" : "[0]";
  int out_j = 0;
  auto out_val = [&](int j) { return "prm.out[" + std::to_string(kOutFirstCol + 2 * j) + "]"; };
  auto out_ok = [&](int j) { return "prm.out[" + std::to_string(kOutFirstCol + 2 * j + 1) + "]"; };

  // ---- group keys → packed key words (word 0 = NULL bitmask of the keys)
  std::string key_code, key_emit;
  std::vector<ExprP> synthetic_exprs;
  int nk = 0;
  if (grouped) {
    nk = 1;
    key_code += "        key[0] = 0;\n";
    int kj = 0;
    for (auto& ge : group_exprs) {
      OutCol oc;
      std::string ok_expr;
      const std::string vb = out_val(out_j), ob = out_ok(out_j);
      const std::string nullbit = std::to_string(1ull << kj) + "ull";
      const bool direct_str = ge->kind == ExprKind::Bound && ge->bound_index >= 0 && (size_t)ge->bound_index < d.in_types.size() &&
                              (d.in_types[(size_t)ge->bound_index].id == TypeId::String || d.in_types[(size_t)ge->bound_index].id == TypeId::Bytes);
      if (direct_str) d.str_key_cols.push_back(ge->bound_index);
      const int id_col = direct_str && dict_id_col && (size_t)ge->bound_index < dict_id_col->size() ? (*dict_id_col)[(size_t)ge->bound_index] : -1;
      if (id_col >= 0) {
        // Utf8 key of any length: group on the representative row index, emit it as the gather index of the source column
        auto idx = std::make_shared<Expr>();
        idx->kind = ExprKind::Bound;
        idx->proto_tag = 3;
        idx->bound_index = id_col;
        idx->dtype = DType::of(TypeId::Int64);
        idx->has_dtype = true;
        synthetic_exprs.push_back(idx);   // Gen memoises by node address: the node must outlive the generation
        Val v = g.named(g.gen(idx));
        ok_expr = v.ok;
        const std::string okc = ok_expr.empty() ? "true" : ok_expr;
        const std::string k0 = std::to_string(nk);
        key_code += "        key[" + k0 + "] = (" + okc + ") ? (u64)(i64)" + v.v + " : 0ull;\n";
        key_emit += "    ((u32*)" + vb + ")[pos] = (u32)key[" + k0 + "];\n";
        nk += 1;
        oc.type = d.in_types[(size_t)ge->bound_index];
        oc.gather_src = ge->bound_index;
      } else {
        Val v = g.named(g.gen(ge));
        ok_expr = v.ok;
        const std::string okc = ok_expr.empty() ? "true" : ok_expr;
        if (v.rep == Rep::STR) {
          // Utf8 key: two packed words; emitted packed, the host expands to Utf8
          const std::string k0 = std::to_string(nk), k1 = std::to_string(nk + 1);
          key_code += "        key[" + k0 + "] = (" + okc + ") ? " + v.v + ".a : 0ull; key[" + k1 + "] = (" + okc + ") ? " + v.v + ".b : 0ull;\n";
          key_emit += "    ((u64*)" + vb + ")[2 * pos] = key[" + k0 + "]; ((u64*)" + vb + ")[2 * pos + 1] = key[" + k1 + "];\n";
          nk += 2;
          oc.type = DType::of(TypeId::String);
          oc.packed_string = true;
        } else if (v.rep == Rep::I128) {
          const std::string k0 = std::to_string(nk), k1 = std::to_string(nk + 1);
          key_code += "        key[" + k0 + "] = (" + okc + ") ? comet::lo64(" + v.v + ") : 0ull; key[" + k1 + "] = (" + okc + ") ? comet::hi64(" + v.v + ") : 0ull;\n";
          key_emit += "    ((i128*)" + vb + ")[pos] = comet::mk128(key[" + k1 + "], key[" + k0 + "]);\n";
          nk += 2;
        } else {
          const char* st = store_ctype(v.t);
          const std::string k0 = std::to_string(nk);
          std::string enc, dec;
          switch (v.rep) {
            case Rep::B: enc = "(u64)(" + v.v + " ? 1 : 0)"; dec = "(u8)key[" + k0 + "]"; break;
            case Rep::I32: case Rep::I64: enc = "(u64)(i64)" + v.v; dec = std::string("(") + st + ")(i64)key[" + k0 + "]"; break;
            // group keys reach the aggregate through NormalizeNaNAndZero (planner.rs:725-729): bit equality is value equality
            case Rep::F64: enc = "(u64)__double_as_longlong(" + v.v + ")"; dec = "__longlong_as_double((i64)key[" + k0 + "])"; break;
            case Rep::F32: enc = "(u64)(u32)__float_as_int(" + v.v + ")"; dec = "__int_as_float((int)(u32)key[" + k0 + "])"; break;
            default: throw CometError("unsupported group key type " + v.t.str());
          }
          if (v.t.id == TypeId::Decimal) dec = "(i128)(i64)key[" + k0 + "]";
          key_code += "        key[" + k0 + "] = (" + okc + ") ? " + enc + " : 0ull;\n";
          key_emit += "    ((" + std::string(st) + "*)" + vb + ")[pos] = " + dec + ";\n";
          nk += 1;
        }
        if (v.rep != Rep::STR) oc.type = v.t;
      }
      if (!ok_expr.empty()) key_code += "        if (!(" + ok_expr + ")) key[0] |= " + nullbit + ";\n";
      key_emit += "    ((u8*)" + ob + ")[pos] = (key[0] & " + nullbit + ") ? 0 : 1;\n";
      oc.nullable = true;
      d.out_cols.push_back(oc);
      ex << "  group key: " << explain_expr(ge) << " : " << oc.type.str() << "\n";
      out_j++;
      kj++;
      if (kj > 60) throw CometError("too many group keys");
    }
  }

  // rows that reach the aggregate
  PrimSlot rowcnt = al.get(Prim::RowCnt, "*", "", "", "");
  const std::string rowcnt_word = std::to_string(rowcnt.word);
  long long max_rows_exact = 0;
  for (auto& in : agg_ins) {
    const AggExpr& a = *in.a;
    std::string fkey, guard;
    if (in.filter) {
      // FILTER (WHERE …): row contributes only if the filter is TRUE and valid (sum_decimal.rs:452-458)
      Val f = g.named(g.gen(in.filter));
      fkey = g.key_of(in.filter);
      guard = Gen::and_ok(f.ok, f.v);
    }
    auto guarded = [&](const std::string& ok) { return Gen::and_ok(guard, ok); };
    const bool final_mode = in.mode == AggMode::Final || in.mode == AggMode::PartialMerge;
    const bool emit_state = in.mode == AggMode::PartialMerge;
    if (final_mode) {
      // ---- Final mode: merge Partial states (merge_batch + evaluate of each accumulator) ----
      auto okx = [](const Val& v) { return v.ok.empty() ? std::string("true") : v.ok; };
      switch (a.kind) {
        case AggKind::Count: {
          // count state: Int64, summed (DataFusion count_udaf merge)
          Val c = g.named(g.gen(in.children.at(0)));
          if (!c.t.is_integer()) throw CometError("Final count expects an Int64 state column");
          PrimSlot s = al.get(Prim::SumI64, "fin:" + g.key_of(in.children[0]), "", c.ok, c.v, (u128)1 << 63);
          OutCol oc; oc.type = DType::of(TypeId::Int64); oc.nullable = false;
          d.out_cols.push_back(oc);
          fin += "    ((i64*)" + out_val(out_j) + ")" + ROW + " = (i64)acc[" + std::to_string(s.word) + "];\n";
          out_j++;
          ex << "  agg(final): count -> Int64\n";
          break;
        }
        case AggKind::Sum: case AggKind::Avg: {
          const bool is_avg = a.kind == AggKind::Avg;
          if (a.dtype.id == TypeId::Decimal) {
            Val sv = g.named(g.gen(in.children.at(0)));
            Val s2 = g.named(g.gen(in.children.at(1)));
            const DType st = is_avg ? a.sum_dtype : a.dtype;
            if (!(sv.t == st)) throw CometError("Final decimal aggregate: state type " + sv.t.str() + " differs from " + st.str());
            const u128 bound = pow10_u128(st.precision) - 1;
            const std::string vk = "fin:" + g.key_of(in.children[0]);
            std::string val128 = sv.rep == Rep::I128 ? sv.v : "(i128)" + sv.v;
            if (!is_avg) {
              // SumDecimal merge_batch (sum_decimal.rs:309-368 / :540-609): state = (sum nullable, is_empty)
              if (s2.rep != Rep::B) throw CometError("Final SumDecimal expects (sum, is_empty) state columns");
              const std::string empty = "(" + s2.v + ")";                       // is_empty is non-null
              const std::string that_ovf = "(!" + empty + " && !" + okx(sv) + ")";  // overflowed partial: sticky
              const std::string contrib = "(!" + empty + " && " + okx(sv) + ")";
              PrimSlot cnt = al.get(Prim::Cnt, vk, "ne", contrib, "");
              PrimSlot any_ovf = al.get(Prim::Cnt, vk, "ovf", that_ovf, "");
              PrimSlot sum = al.get(Prim::Sum192, vk, "", contrib, val128, bound);
              PrimSlot amax = al.get(Prim::AMaxHi, vk, "", contrib, val128);
              PrimSlot sflags = al.get(Prim::SignFlags, vk, "", contrib, val128);
              auto kw = [&](const PrimSlot& ps) {
                return ps.kernel_level ? "((const u64*)prm.out[" + std::to_string(kOutErr) + "])[2 + " + std::to_string(ps.word) + "]"
                                       : "acc[" + std::to_string(ps.word) + "]";
              };
              const std::string W = std::to_string(sum.word), W1 = std::to_string(sum.word + 1);
              fin += "    {\n      i128 total = comet::mk128(acc[" + W1 + "], acc[" + W + "]);\n      bool ovf = false;\n";
              fin += "      comet::sum_overflow_decide(acc + " + W + ", " + kw(amax) + ", " + kw(sflags) + ", acc[" + std::to_string(cnt.word) + "], " +
                     lit_u128(bound) + ", ovf, (unsigned int*)prm.out[" + std::to_string(kOutErr) + "]);\n";
              g.uses_err = true;
              // ANSI: merging into an overflow fails the query (sum_decimal.rs:352-358, 594-600)
              if (a.eval_mode == EvalMode::Ansi) note_agg_ctx(0, a);
              if (a.eval_mode == EvalMode::Ansi)
                fin += "      if (acc[" + std::to_string(any_ovf.word) + "] != 0 || ovf || (acc[" + std::to_string(cnt.word) + "] != 0 && !comet::dec_fits(total, " + lit_u128(bound) +
                       "))) atomicOr((unsigned int*)prm.out[" + std::to_string(kOutErr) + "], 65536u);\n";
              if (emit_state) {
                // merged state (sum_decimal.rs:281-295 after :309-368): sum is NULL once any side overflowed, is_empty only
                // if every merged state was empty
                fin += "      bool sovf = acc[" + std::to_string(any_ovf.word) + "] != 0 || ovf || !comet::dec_fits(total, " + lit_u128(bound) + ");\n";
                fin += "      bool empty = acc[" + std::to_string(cnt.word) + "] == 0 && acc[" + std::to_string(any_ovf.word) + "] == 0;\n";
                fin += "      ((i128*)" + out_val(out_j) + ")" + ROW + " = sovf ? (i128)0 : total;\n";
                fin += "      ((u8*)" + out_ok(out_j) + ")" + ROW + " = sovf ? 0 : 1;\n";
                fin += "      ((u8*)" + out_val(out_j + 1) + ")" + ROW + " = empty ? 1 : 0;\n    }\n";
                OutCol s0; s0.type = st; s0.nullable = true;
                OutCol s1; s1.type = DType::of(TypeId::Bool); s1.nullable = false;
                d.out_cols.push_back(s0);
                d.out_cols.push_back(s1);
                out_j += 2;
                ex << "  agg(partial-merge): sum_decimal -> (" << st.str() << ", is_empty)\n";
              } else {
              // evaluate (sum_decimal.rs:264-279): NULL if empty, overflowed, or out of precision
              fin += "      bool isnull = acc[" + std::to_string(cnt.word) + "] == 0 || acc[" + std::to_string(any_ovf.word) + "] != 0 || ovf || !comet::dec_fits(total, " + lit_u128(bound) + ");\n";
              fin += "      ((i128*)" + out_val(out_j) + ")" + ROW + " = isnull ? (i128)0 : total;\n";
              fin += "      ((u8*)" + out_ok(out_j) + ")" + ROW + " = isnull ? 0 : 1;\n    }\n";
              OutCol oc; oc.type = st; oc.nullable = true;
              d.out_cols.push_back(oc);
              out_j++;
              ex << "  agg(final): sum_decimal -> " << st.str() << "\n";
              }
            } else {
              // AvgDecimal merge (avg_decimal.rs:542-595) + evaluate (:597-636, avg() :670-689): state = (sum nullable, count)
              if (!s2.t.is_integer()) throw CometError("Final AvgDecimal expects (sum, count) state columns");
              PrimSlot cnt = al.get(Prim::SumI64, vk + "#cnt", "", s2.ok, s2.v, (u128)1 << 63);
              PrimSlot bad = al.get(Prim::Cnt, vk, "nullstate", "(!" + okx(sv) + " || !" + okx(s2) + ")", "");
              PrimSlot nsum = al.get(Prim::Cnt, vk, "nsum", sv.ok, "");
              PrimSlot sum = al.get(Prim::Sum192, vk, "", sv.ok, val128, bound);
              PrimSlot amax = al.get(Prim::AMaxHi, vk, "", sv.ok, val128);
              PrimSlot sflags = al.get(Prim::SignFlags, vk, "", sv.ok, val128);
              auto kw = [&](const PrimSlot& ps) {
                return ps.kernel_level ? "((const u64*)prm.out[" + std::to_string(kOutErr) + "])[2 + " + std::to_string(ps.word) + "]"
                                       : "acc[" + std::to_string(ps.word) + "]";
              };
              const std::string W = std::to_string(sum.word), W1 = std::to_string(sum.word + 1);
              const u128 tbound = pow10_u128(a.dtype.precision) - 1;
              const int up = std::max(0, a.dtype.scale - st.scale);
              fin += "    {\n      i128 total = comet::mk128(acc[" + W1 + "], acc[" + W + "]);\n      bool ovf = false;\n";
              fin += "      comet::sum_overflow_decide(acc + " + W + ", " + kw(amax) + ", " + kw(sflags) + ", acc[" + std::to_string(nsum.word) + "], " +
                     lit_u128(bound) + ", ovf, (unsigned int*)prm.out[" + std::to_string(kOutErr) + "]);\n";
              g.uses_err = true;
              fin += "      i64 count = (i64)acc[" + std::to_string(cnt.word) + "];\n";
              // grouped (AvgDecimalGroupsAccumulator::merge_batch, avg_decimal.rs:542-595): a NULL partial sum / count or an
              // overflowing merge step clears is_not_null for good.  Ungrouped (AvgDecimalAccumulator::merge_batch, :331-356):
              // arrow's sum() skips NULL partial sums and only the batch total is checked against the precision.
              if (grouped) fin += "      bool sum_ok = acc[" + std::to_string(bad.word) + "] == 0 && !ovf;\n";
              else fin += "      bool sum_ok = acc[" + std::to_string(nsum.word) + "] != 0 && comet::dec_fits(total, " + lit_u128(bound) + ");\n";
              // ANSI: an overflowed sum under a count fails the query (avg_decimal.rs:366-380, 576-580, 610-616)
              if (a.eval_mode == EvalMode::Ansi) note_agg_ctx(1, a);
              if (a.eval_mode == EvalMode::Ansi)
                fin += "      if (!sum_ok && count > 0) atomicOr((unsigned int*)prm.out[" + std::to_string(kOutErr) + "], 131072u);\n";
              if (emit_state) {
                // state: grouped sums and counts share the is_not_null mask (:638-653); ungrouped (sum Option, count) (:301-306)
                fin += "      ((i128*)" + out_val(out_j) + ")" + ROW + " = sum_ok ? total : (i128)0;\n";
                fin += "      ((u8*)" + out_ok(out_j) + ")" + ROW + " = sum_ok ? 1 : 0;\n";
                fin += std::string("      ((i64*)") + out_val(out_j + 1) + ")" + ROW + " = " + (grouped ? "sum_ok ? count : 0" : "count") + ";\n";
                fin += std::string("      ((u8*)") + out_ok(out_j + 1) + ")" + ROW + " = " + (grouped ? "sum_ok ? 1 : 0" : "1") + ";\n    }\n";
                OutCol s0; s0.type = st; s0.nullable = true;
                OutCol s1; s1.type = DType::of(TypeId::Int64); s1.nullable = true;
                d.out_cols.push_back(s0);
                d.out_cols.push_back(s1);
                out_j += 2;
                ex << "  agg(partial-merge): avg_decimal -> (" << st.str() << ", count)\n";
              } else {
              fin += "      i128 avgv = 0;\n";
              fin += "      bool has = sum_ok && count != 0 && comet::dec_avg(total, count, " +
                     lit_i128((i128)pow10_u128(up)) + ", " + lit_u128(tbound) + ", avgv);\n";
              fin += "      ((i128*)" + out_val(out_j) + ")" + ROW + " = has ? avgv : (i128)0;\n";
              fin += "      ((u8*)" + out_ok(out_j) + ")" + ROW + " = has ? 1 : 0;\n    }\n";
              OutCol oc; oc.type = a.dtype; oc.nullable = true;
              d.out_cols.push_back(oc);
              out_j++;
              ex << "  agg(final): avg_decimal -> " << a.dtype.str() << "\n";
              }
            }
          } else if (!is_avg && a.dtype.is_integer()) {
            // SumInteger merge (sum_int.rs:494-530): wrapping sum of the non-null partial sums, NULL if none
            Val sv = g.named(g.gen(in.children.at(0)));
            const std::string vk = "fin:" + g.key_of(in.children[0]);
            PrimSlot cnt = al.get(Prim::Cnt, vk, "", sv.ok, "");
            PrimSlot sum = al.get(Prim::SumI64, vk, "", sv.ok, sv.v, (u128)1 << 63);
            fin += "    ((i64*)" + out_val(out_j) + ")" + ROW + " = acc[" + std::to_string(cnt.word) + "] ? (i64)acc[" + std::to_string(sum.word) + "] : 0;\n";
            fin += "    ((u8*)" + out_ok(out_j) + ")" + ROW + " = acc[" + std::to_string(cnt.word) + "] ? 1 : 0;\n";
            OutCol oc; oc.type = DType::of(TypeId::Int64); oc.nullable = true;
            d.out_cols.push_back(oc);
            out_j++;
            ex << "  agg(final): sum_int -> Int64\n";
          } else {
            // float: Avg merge/evaluate (avg.rs:146-176, :283-327) / DataFusion sum merge
            Val sv = g.named(g.gen(in.children.at(0)));
            if (sv.rep != Rep::F64) throw CometError("Final float aggregate expects a Float64 state column");
            const std::string vk = "fin:" + g.key_of(in.children[0]);
            PrimSlot nsum = al.get(Prim::Cnt, vk, "", sv.ok, "");
            const AggLowering::FSum fsum = al.get_fsum(vk, "", sv.ok, sv.v);
            const std::string SUMX = al.fread(fsum);
            if (is_avg && emit_state) {
              // AvgAccumulator / AvgGroupsAccumulator state after merge (avg.rs:139-176): (sum, count)
              Val s2 = g.named(g.gen(in.children.at(1)));
              PrimSlot cnt = al.get(Prim::SumI64, vk + "#cnt", "", s2.ok, s2.v, (u128)1 << 63);
              fin += "    ((double*)" + out_val(out_j) + ")" + ROW + " = " + SUMX + ";\n";
              fin += std::string("    ((u8*)") + out_ok(out_j) + ")" + ROW + " = " + (grouped ? "1" : "acc[" + std::to_string(nsum.word) + "] ? 1 : 0") + ";\n";
              fin += "    ((i64*)" + out_val(out_j + 1) + ")" + ROW + " = (i64)acc[" + std::to_string(cnt.word) + "];\n";
              fin += "    ((u8*)" + out_ok(out_j + 1) + ")" + ROW + " = 1;\n";
              OutCol s0; s0.type = DType::of(TypeId::Double); s0.nullable = true;
              OutCol s1; s1.type = DType::of(TypeId::Int64); s1.nullable = true;
              d.out_cols.push_back(s0);
              d.out_cols.push_back(s1);
              out_j += 2;
              ex << "  agg(partial-merge): avg_f64 -> (Float64, count)\n";
              break;
            }
            if (is_avg) {
              Val s2 = g.named(g.gen(in.children.at(1)));
              PrimSlot cnt = al.get(Prim::SumI64, vk + "#cnt", "", s2.ok, s2.v, (u128)1 << 63);
              fin += "    { i64 count = (i64)acc[" + std::to_string(cnt.word) + "];\n";
              fin += "      ((double*)" + out_val(out_j) + ")" + ROW + " = count ? comet::fp_div(" + SUMX + ", (double)count) : 0.0;\n";
              fin += "      ((u8*)" + out_ok(out_j) + ")" + ROW + " = count ? 1 : 0; }\n";
            } else {
              fin += "    ((double*)" + out_val(out_j) + ")" + ROW + " = " + SUMX + ";\n";
              fin += "    ((u8*)" + out_ok(out_j) + ")" + ROW + " = acc[" + std::to_string(nsum.word) + "] ? 1 : 0;\n";
            }
            OutCol oc; oc.type = DType::of(TypeId::Double); oc.nullable = true;
            d.out_cols.push_back(oc);
            out_j++;
            ex << "  agg(final): " << (is_avg ? "avg" : "sum") << "_f64 -> Float64\n";
          }
          break;
        }
        case AggKind::Min: case AggKind::Max: {
          Val v = g.named(g.gen(in.children.at(0)));
          const bool mn = a.kind == AggKind::Min;
          const std::string vk = "fin:" + g.key_of(in.children[0]);
          PrimSlot cnt = al.get(Prim::Cnt, vk, "", v.ok, "");
          PrimSlot s;
          std::string rd;
          const char* stc = store_ctype(v.t);
          if (v.rep == Rep::I32 || v.rep == Rep::I64) {
            s = al.get(mn ? Prim::MinI64 : Prim::MaxI64, vk, "", v.ok, v.v);
            rd = v.t.id == TypeId::Decimal ? "(i128)(i64)acc[" + std::to_string(s.word) + "]" : std::string("(") + stc + ")(i64)acc[" + std::to_string(s.word) + "]";
          } else if (v.rep == Rep::I128) {
            s = al.get(mn ? Prim::MinI128 : Prim::MaxI128, vk, "", v.ok, v.v);
            rd = "comet::mk128(acc[" + std::to_string(s.word + 1) + "], acc[" + std::to_string(s.word) + "])";
          } else if (v.rep == Rep::F64 || v.rep == Rep::F32) {
            s = al.get(mn ? Prim::MinF64 : Prim::MaxF64, vk, "", v.ok, v.v);
            rd = std::string("(") + stc + ")__longlong_as_double((i64)acc[" + std::to_string(s.word) + "])";
          } else throw CometError("min/max over " + v.t.str() + " is not supported in the GPU pipeline yet");
          const std::string C = std::to_string(cnt.word);
          fin += "    ((" + std::string(stc) + "*)" + out_val(out_j) + ")" + ROW + " = acc[" + C + "] ? " + rd + " : (" + stc + ")0;\n";
          fin += "    ((u8*)" + out_ok(out_j) + ")" + ROW + " = acc[" + C + "] ? 1 : 0;\n";
          OutCol oc; oc.type = v.t; oc.nullable = true;
          d.out_cols.push_back(oc);
          out_j++;
          ex << "  agg(final): " << (mn ? "min" : "max") << " -> " << v.t.str() << "\n";
          break;
        }
        default:
          throw CometError("Final mode of aggregate (tag " + std::to_string(a.proto_tag) + ") is not supported by the MI355X native engine");
      }
      continue;
    }
    switch (a.kind) {
      case AggKind::Count: {
        if (in.children.empty()) throw CometError("count() without children");
        std::string ok, vkey;
        for (auto& c : in.children) {
          Val v = g.gen(c);
          ok = Gen::and_ok(ok, v.ok);
          vkey += g.key_of(c) + ",";
        }
        PrimSlot s = al.get(Prim::Cnt, vkey, fkey, guarded(ok), "");
        OutCol oc; oc.type = DType::of(TypeId::Int64); oc.nullable = false;
        d.out_cols.push_back(oc);
        fin += "    ((i64*)" + out_val(out_j) + ")" + ROW + " = (i64)acc[" + std::to_string(s.word) + "];\n";
        out_j++;
        ex << "  agg: count -> Int64\n";
        break;
      }
      case AggKind::Sum: case AggKind::Avg: {
        if (in.children.size() != 1) throw CometError("sum/avg expects one child");
        Val v = g.named(g.gen(in.children[0]));
        std::string vkey = g.key_of(in.children[0]);
        const bool is_avg = a.kind == AggKind::Avg;
        const DType& rt = a.dtype;
        const std::string cond = guarded(v.ok);
        if (rt.id == TypeId::Decimal) {
          if (v.t.id != TypeId::Decimal) throw CometError("decimal sum/avg over non-decimal input " + v.t.str());
          const DType st = is_avg ? a.sum_dtype : rt;   // accumulation type
          if (st.id != TypeId::Decimal) throw CometError("Invalid data type for SumDecimal");
          if (st.scale != v.t.scale) throw CometError("decimal sum/avg: input scale differs from sum scale");
          const u128 bound = pow10_u128(st.precision) - 1;
          // Appendix C.1: no prefix can overflow while rows × max|v| ≤ 10^p − 1
          u128 vmax = v.maxabs == 0 ? 1 : v.maxabs;
          u128 safe_rows = vmax == kUnbounded ? 0 : bound / vmax;
          const bool dynamic = safe_rows < ((u128)1 << 33);  // Spark sizes sum types for 10^10 rows (p+10)
          if (!dynamic) {
            long long sr = safe_rows > (u128)0x7fffffffffffffffll ? 0x7fffffffffffffffll : (long long)safe_rows;
            if (max_rows_exact == 0 || sr < max_rows_exact) max_rows_exact = sr;
          }
          std::string val128 = v.rep == Rep::I128 ? v.v : "(i128)" + v.v;
          PrimSlot cnt = al.get(Prim::Cnt, vkey, fkey, cond, "");
          PrimSlot sum, amax{}, sflags{};
          if (!dynamic) {
            sum = al.get(Prim::Sum128, vkey, fkey, cond, val128, v.maxabs);
          } else {
            sum = al.get(Prim::Sum192, vkey, fkey, cond, val128, v.maxabs);
            amax = al.get(Prim::AMaxHi, vkey, fkey, cond, val128);
            sflags = al.get(Prim::SignFlags, vkey, fkey, cond, val128);
          }
          std::string W = std::to_string(sum.word), W1 = std::to_string(sum.word + 1), C = std::to_string(cnt.word);
          fin += "    {\n      i128 total = comet::mk128(acc[" + W1 + "], acc[" + W + "]);\n      bool ovf = false;\n";
          if (dynamic) {
            auto kw = [&](const PrimSlot& ps) {
              return ps.kernel_level ? "((const u64*)prm.out[" + std::to_string(kOutErr) + "])[2 + " + std::to_string(ps.word) + "]"
                                     : "acc[" + std::to_string(ps.word) + "]";
            };
            fin += "      comet::sum_overflow_decide(acc + " + W + ", " + kw(amax) + ", " + kw(sflags) + ", acc[" + C + "], " + lit_u128(bound) +
                   ", ovf, (unsigned int*)prm.out[" + std::to_string(kOutErr) + "]);\n";
            g.uses_err = true;
            // ANSI: an overflowing decimal SUM fails the query instead of turning NULL (sum_decimal.rs:211-215, 427-431).  An average only notes the
            // overflow in its state here (avg_decimal.rs:268-300, 483-503) and raises when the states are merged or evaluated (:366-380, 576-580, 610-616)
            if (a.eval_mode == EvalMode::Ansi && !is_avg) note_agg_ctx(0, a);
            if (a.eval_mode == EvalMode::Ansi && !is_avg)
              fin += "      if (ovf) atomicOr((unsigned int*)prm.out[" + std::to_string(kOutErr) + "], 65536u);\n";
          }
          if (!is_avg) {
            // SumDecimal state (sum_decimal.rs:281-295, :526-538): (sum | NULL if overflowed, is_empty)
            fin += "      bool empty = acc[" + C + "] == 0;\n";
            fin += "      ((i128*)" + out_val(out_j) + ")" + ROW + " = ovf ? (i128)0 : total;\n";
            fin += "      ((u8*)" + out_ok(out_j) + ")" + ROW + " = ovf ? 0 : 1;\n";
            fin += "      ((u8*)" + out_val(out_j + 1) + ")" + ROW + " = empty ? 1 : 0;\n    }\n";
            OutCol s0; s0.type = st; s0.nullable = true;
            OutCol s1; s1.type = DType::of(TypeId::Bool); s1.nullable = false;
            d.out_cols.push_back(s0);
            d.out_cols.push_back(s1);
            out_j += 2;
            ex << "  agg: sum_decimal -> (" << st.str() << ", is_empty)\n";
          } else {
            if (grouped) {
              // AvgDecimalGroupsAccumulator::state (avg_decimal.rs:638-653): sum and count share the is_not_null mask
              fin += "      ((i128*)" + out_val(out_j) + ")" + ROW + " = ovf ? (i128)0 : total;\n";
              fin += "      ((u8*)" + out_ok(out_j) + ")" + ROW + " = ovf ? 0 : 1;\n";
              fin += "      ((i64*)" + out_val(out_j + 1) + ")" + ROW + " = ovf ? 0 : (i64)acc[" + C + "];\n";
              fin += "      ((u8*)" + out_ok(out_j + 1) + ")" + ROW + " = ovf ? 0 : 1;\n    }\n";
            } else {
              // AvgDecimalAccumulator::state (avg_decimal.rs:283-288): sum = None until the first value
              fin += "      bool none = acc[" + C + "] == 0 || ovf;\n";
              fin += "      ((i128*)" + out_val(out_j) + ")" + ROW + " = none ? (i128)0 : total;\n";
              fin += "      ((u8*)" + out_ok(out_j) + ")" + ROW + " = none ? 0 : 1;\n";
              fin += "      ((i64*)" + out_val(out_j + 1) + ")" + ROW + " = (i64)acc[" + C + "];\n";
              fin += "      ((u8*)" + out_ok(out_j + 1) + ")" + ROW + " = 1;\n    }\n";
            }
            OutCol s0; s0.type = st; s0.nullable = true;
            OutCol s1; s1.type = DType::of(TypeId::Int64); s1.nullable = true;
            d.out_cols.push_back(s0);
            d.out_cols.push_back(s1);
            out_j += 2;
            ex << "  agg: avg_decimal -> (" << st.str() << ", count)\n";
          }
        } else if (!is_avg && rt.is_integer()) {
          // SumInteger LEGACY (sum_int.rs:117-160, :403-475): wrapping i64, NULL until a non-null value arrives
          if (!v.t.is_integer()) throw CometError("integer sum over " + v.t.str());
          if (a.eval_mode != EvalMode::Legacy) throw CometError("ANSI/TRY integer sum is not supported in the GPU pipeline yet");
          PrimSlot cnt = al.get(Prim::Cnt, vkey, fkey, cond, "");
          PrimSlot sum = al.get(Prim::SumI64, vkey, fkey, cond, v.v, v.maxabs);
          fin += "    ((i64*)" + out_val(out_j) + ")" + ROW + " = acc[" + std::to_string(cnt.word) + "] ? (i64)acc[" + std::to_string(sum.word) + "] : 0;\n";
          fin += "    ((u8*)" + out_ok(out_j) + ")" + ROW + " = acc[" + std::to_string(cnt.word) + "] ? 1 : 0;\n";
          OutCol s0; s0.type = DType::of(TypeId::Int64); s0.nullable = true;
          d.out_cols.push_back(s0);
          out_j++;
          ex << "  agg: sum_int -> Int64\n";
        } else {
          // float sum (DataFusion sum_udaf over Float64, planner.rs:2628-2634) / Avg (avg.rs): child cast to Float64
          if (!(v.rep == Rep::F64 || v.rep == Rep::F32 || v.rep == Rep::I32 || v.rep == Rep::I64))
            throw CometError("float sum/avg over " + v.t.str() + " is not supported in the GPU pipeline yet");
          PrimSlot cnt = al.get(Prim::Cnt, vkey, fkey, cond, "");
          const AggLowering::FSum fsum = al.get_fsum("f64:" + vkey, fkey, cond, v.v);
          const std::string SUMX = al.fread(fsum), C = std::to_string(cnt.word);
          if (is_avg) {
            // AvgAccumulator::state (avg.rs:139-144): ungrouped sum is Some once any batch arrived; grouped never NULL
            fin += "    ((double*)" + out_val(out_j) + ")" + ROW + " = " + SUMX + ";\n";
            fin += "    ((u8*)" + out_ok(out_j) + ")" + ROW + " = " + (grouped ? std::string("1") : "acc[" + rowcnt_word + "] ? 1 : 0") + ";\n";
            fin += "    ((i64*)" + out_val(out_j + 1) + ")" + ROW + " = (i64)acc[" + C + "];\n";
            fin += "    ((u8*)" + out_ok(out_j + 1) + ")" + ROW + " = 1;\n";
            OutCol s0; s0.type = DType::of(TypeId::Double); s0.nullable = true;
            OutCol s1; s1.type = DType::of(TypeId::Int64); s1.nullable = true;
            d.out_cols.push_back(s0);
            d.out_cols.push_back(s1);
            out_j += 2;
            ex << "  agg: avg_f64 -> (Float64, count)\n";
          } else {
            fin += "    ((double*)" + out_val(out_j) + ")" + ROW + " = " + SUMX + ";\n";
            fin += "    ((u8*)" + out_ok(out_j) + ")" + ROW + " = acc[" + C + "] ? 1 : 0;\n";
            OutCol s0; s0.type = DType::of(TypeId::Double); s0.nullable = true;
            d.out_cols.push_back(s0);
            out_j++;
            ex << "  agg: sum_f64 -> Float64\n";
          }
        }
        break;
      }
      case AggKind::Min: case AggKind::Max: {
        if (in.children.size() != 1) throw CometError("min/max expects one child");
        Val v = g.named(g.gen(in.children[0]));
        std::string vkey = g.key_of(in.children[0]);
        const bool mn = a.kind == AggKind::Min;
        if (!(v.t == a.dtype)) throw CometError("min/max with cast is not supported in the GPU pipeline yet");
        const std::string cond = guarded(v.ok);
        PrimSlot cnt = al.get(Prim::Cnt, vkey, fkey, cond, "");
        PrimSlot s;
        std::string rd;
        const char* st = store_ctype(v.t);
        if (v.rep == Rep::I32 || v.rep == Rep::I64) {
          s = al.get(mn ? Prim::MinI64 : Prim::MaxI64, vkey, fkey, cond, v.v);
          rd = v.t.id == TypeId::Decimal ? "(i128)(i64)acc[" + std::to_string(s.word) + "]" : std::string("(") + st + ")(i64)acc[" + std::to_string(s.word) + "]";
        } else if (v.rep == Rep::I128) {
          s = al.get(mn ? Prim::MinI128 : Prim::MaxI128, vkey, fkey, cond, v.v);
          rd = "comet::mk128(acc[" + std::to_string(s.word + 1) + "], acc[" + std::to_string(s.word) + "])";
        } else if (v.rep == Rep::F64 || v.rep == Rep::F32) {
          s = al.get(mn ? Prim::MinF64 : Prim::MaxF64, vkey, fkey, cond, v.v);
          rd = std::string("(") + st + ")__longlong_as_double((i64)acc[" + std::to_string(s.word) + "])";
        } else {
          throw CometError("min/max over " + v.t.str() + " is not supported in the GPU pipeline yet");
        }
        std::string C = std::to_string(cnt.word);
        fin += "    ((" + std::string(st) + "*)" + out_val(out_j) + ")" + ROW + " = acc[" + C + "] ? " + rd + " : (" + st + ")0;\n";
        fin += "    ((u8*)" + out_ok(out_j) + ")" + ROW + " = acc[" + C + "] ? 1 : 0;\n";
        OutCol s0; s0.type = v.t; s0.nullable = true;
        d.out_cols.push_back(s0);
        out_j++;
        ex << "  agg: " << (mn ? "min" : "max") << " -> " << v.t.str() << "\n";
        break;
      }
      default:
        throw CometError("Aggregate function (tag " + std::to_string(a.proto_tag) + ") is not supported by the MI355X native engine");
    }
  }
  if (d.out_cols.size() * 2 + kOutFirstCol > COMET_MAX_OUT) throw CometError("too many aggregate state columns for one GPU pipeline");
  d.NW = al.nw;
  d.fix_sums = al.fix_sums;
  d.NK = nk;
  d.R = grouped ? 2 : 4;
  if (const char* e = getenv("COMET_GEN_R")) d.R = std::max(1, std::min(8, atoi(e)));
  d.max_rows_exact = max_rows_exact;
  d.in_used = g.in_used;
  src << "struct P {\n  static constexpr int R = " << d.R << ";\n  static constexpr int NW = " << d.NW << ";\n";
  src << "  static __device__ __forceinline__ void init(u64* a) {\n" << al.init_code << "  }\n";
  src << "  static __device__ __forceinline__ void combine(u64* a, const u64* b) {\n" << al.combine_code << "  }\n";
  if (!grouped) {
    const std::string rowinit = "    bool k[R]; i64 idx[R];\n    _Pragma(\"unroll\") for (int r = 0; r < R; r++) { idx[r] = base + (i64)r * comet::kBlock + threadIdx.x; k[r] = idx[r] < n; }\n";
    src << "  static constexpr bool PIPELINED = " << (g.pipelined ? "true" : "false") << ";\n";
    if (g.pipelined) {
      src << "  struct L {\n" << g.ldecls << "  };\n";
      src << "  static __device__ __forceinline__ void tile_load(const CometKParams& prm, i64 base, i64 n, L& ld) {\n" << rowinit << g.prefetch_body() << "  }\n";
      src << "  static __device__ __forceinline__ void tile(const CometKParams& prm, i64 base, i64 n, L& ld, u64* acc) {\n" << rowinit << g.laliases
          << g.decls << g.body("", true) << "  }\n";
    } else {
      src << "  static __device__ __forceinline__ void tile(const CometKParams& prm, i64 base, i64 n, u64* acc) {\n" << rowinit << g.decls << g.body() << "  }\n";
    }
    src << "  static __device__ __forceinline__ void kexport(const CometKParams& prm, const u64* acc) {\n"
        << "    unsigned long long* aux = (unsigned long long*)prm.out[" << kOutErr << "] + 2; (void)aux;\n" << al.kexport_code << "  }\n";
    src << "  static __device__ __forceinline__ void finalize(const CometKParams& prm, const u64* acc) {\n" << fin << "  }\n};\n";
    src << "extern \"C\" __global__ __launch_bounds__(256) void k_agg(const CometKParams prm) { comet::agg_nogroup_body<P>(prm); }\n";
    src << "extern \"C\" __global__ __launch_bounds__(256) void k_agg_final(const CometKParams prm) { comet::agg_nogroup_final_body<P>(prm); }\n";
    d.kernels = {"k_agg", "k_agg_final"};
  } else {
    // LDS budget: private copies [GC][NPW][COPIES] + table slots ≈ 20 KiB per block (8 blocks = 32 waves per CU)
    d.NPW = al.npw;
    const int copies = 32;
    int gc = 8;
    while (gc > 1 && gc * al.npw * copies * 8 > 10 * 1024) gc >>= 1;
    if (const char* e = getenv("COMET_GEN_GC")) gc = std::max(1, std::min(16, atoi(e)));
    const int slot_bytes = 8 + 8 * (d.NK + al.npw + 1);
    int cap = 1024;
    while (cap > 16 && cap * slot_bytes > 10 * 1024) cap >>= 1;
    d.lds_cap = cap;
    if (al.nkw > kErrAuxWords) throw   // the last 64 bytes of the block are scratch words of the executor
       CometError("too many overflow-tracked sums in one aggregate");
    src << "  static constexpr int NK = " << d.NK << ";\n  static constexpr int NPW = " << al.npw << ";\n  static constexpr int NKW = " << al.nkw
        << ";\n  static constexpr int LDS_CAP = " << cap << ";\n  static constexpr int GC = " << gc << ";\n  static constexpr int COPIES = " << copies << ";\n";
    auto emit_switch = [&](const char* sig, const std::vector<std::string>& v, const char* prefix, const char* dflt) {
      src << "  static __device__ __forceinline__ constexpr " << sig << " {\n    switch (k) {\n";
      for (size_t k = 0; k < v.size(); k++) src << "      case " << k << ": return " << prefix << v[k] << ";\n";
      src << "      default: return " << dflt << ";\n    }\n  }\n";
    };
    emit_switch("int op(int k)", al.gops, "comet::", "comet::G_CONT");
    emit_switch("u64 identity(int k)", al.gident, "", "0ull");
    emit_switch("int pop(int k)", al.pops, "comet::", "comet::G_CONT");
    emit_switch("u64 pidentity(int k)", al.pident, "", "0ull");
    emit_switch("int kop(int k)", al.kops, "comet::", "comet::G_OR64");
    src << "  static __device__ __forceinline__ void kinit(u64* kacc) { for (int k = 0; k < (NKW > 0 ? NKW : 1); k++) kacc[k] = 0; }\n";
    src << "  static __device__ __forceinline__ void fold(const u64* pw, u64* val) {\n" << al.fold_code << "  }\n";
    const std::string rowinit = "    bool k[R]; i64 idx[R];\n    _Pragma(\"unroll\") for (int r = 0; r < R; r++) { idx[r] = base + (i64)r * comet::kBlock + threadIdx.x; k[r] = idx[r] < n; }\n";
    const std::string update = "    _Pragma(\"unroll\") for (int r = 0; r < R; r++) {\n      if (k[r]) {\n        u64 key[NK]; u64 pv[NPW];\n" + key_code + al.pv_code +
                               al.kfeed_code + "        comet::group_update<P>(grp, true, key, pv);\n      }\n    }\n  }\n";
    src << "  static constexpr bool PIPELINED = " << (g.pipelined ? "true" : "false") << ";\n";
    if (g.pipelined) {
      src << "  struct L {\n" << g.ldecls << "  };\n";
      src << "  static __device__ __forceinline__ void tile_load(const CometKParams& prm, i64 base, i64 n, L& ld) {\n" << rowinit << g.prefetch_body() << "  }\n";
      src << "  static __device__ __forceinline__ void tile_grouped(const CometKParams& prm, i64 base, i64 n, L& ld, const comet::GroupCtx<P>& grp, u64* kacc) {\n"
          << rowinit << g.laliases << g.decls << g.body("", true) << update;
    } else {
      src << "  static __device__ __forceinline__ void tile_grouped(const CometKParams& prm, i64 base, i64 n, const comet::GroupCtx<P>& grp, u64* kacc) {\n"
          << rowinit << g.decls << g.body() << update;
    }
    src << "  static __device__ __forceinline__ void emit_group(const CometKParams& prm, const u64* key, const u64* acc, i64 pos) {\n"
        << key_emit << fin << "  }\n};\n";
    src << "extern \"C\" __global__ __launch_bounds__(256, COMET_WAVES_GAGG) void k_gagg(const CometKParams prm) { comet::agg_grouped_body<P>(prm); }\n";
    src << "extern \"C\" __global__ __launch_bounds__(256) void k_gemit(const CometKParams prm) { comet::agg_grouped_emit_body<P>(prm); }\n";
    src << "extern \"C\" __global__ __launch_bounds__(256) void k_grehash(const CometKParams prm) { comet::agg_grouped_rehash_body<P>(prm); }\n";
    src << "extern \"C\" __global__ __launch_bounds__(256) void k_pack(const CometKParams prm) { comet::pack_validity_body((const u8*)prm.out[0], (u8*)prm.out[1], prm.n); }\n";
    // (only where the path can apply — a merging aggregate, or one over a materialised source such as a join's output: a Partial aggregate over a Scan sees chunks
    // of a stream, and its three extra kernels would only lengthen the cold compile: SF100 Q1's plan 497 → 680 ms)
    // (… a Partial aggregate keyed by Utf8 columns is "materialised" only so that long strings can be swapped for row indices — TPC-H Q1's shape: four groups)
    const bool part_kernels = d.merges_states || (source_types != nullptr && d.str_key_cols.empty());
    if (part_kernels) {
      // a grouped aggregate over one chunk may run partitioned (comet_device.hpp template C''): count / scatter passes over the rows, an LDS merge + emit per partition
      src << "extern \"C\" __global__ __launch_bounds__(256) void k_gphist(const CometKParams prm) { comet::agg_part_pass_body<P, 1>(prm); }\n";
      src << "extern \"C\" __global__ __launch_bounds__(256) void k_gpscat(const CometKParams prm) { comet::agg_part_pass_body<P, 2>(prm); }\n";
      src << "extern \"C\" __global__ __launch_bounds__(256) void k_gpmerge(const CometKParams prm) { comet::agg_part_merge_body<P>(prm); }\n";
    }
    d.kernels = {"k_gagg", "k_gemit", "k_grehash", "k_pack"};
    if (part_kernels) {
      d.kernels.push_back("k_gphist");
      d.kernels.push_back("k_gpscat");
      d.kernels.push_back("k_gpmerge");
    }
  }
  d.source = with_optional_headers(src.str());
  d.explain = ex.str();
  return d;
}

// ---------------------------------------------------------------------------------------------
// generate_join: key hashing / equality / residual condition / output gather functor for the hash-join templates
// (join_build_body, join_count_body, join_emit_body).  Reference: planner.rs:2192-2266, :2415-2555.
// ---------------------------------------------------------------------------------------------
void fold_chain(const Operator& top, const Operator& source, const std::vector<DType>& source_types, std::vector<ExprP>& cols, std::vector<ExprP>& preds) {
  std::vector<const Operator*> chain;
  for (const Operator* cur = &top; cur != &source; cur = cur->children[0].get()) {
    if ((cur->kind != OpKind::Filter && cur->kind != OpKind::Projection) || cur->children.size() != 1) throw CometError("internal: fold_chain over a non-chain");
    chain.push_back(cur);
  }
  cols.clear();
  preds.clear();
  for (size_t i = 0; i < source_types.size(); i++) {
    auto b = std::make_shared<Expr>();
    b->kind = ExprKind::Bound;
    b->proto_tag = 3;
    b->bound_index = (int)i;
    b->dtype = source_types[i];
    b->has_dtype = true;
    cols.push_back(b);
  }
  for (int i = (int)chain.size() - 1; i >= 0; i--) {
    const Operator& op = *chain[(size_t)i];
    std::map<const Expr*, ExprP> memo;
    if (op.kind == OpKind::Filter) {
      if (!op.predicate) throw CometError("Filter without predicate");
      split_conjuncts(substitute(op.predicate, cols, memo), preds);
    } else {
      std::vector<ExprP> nc;
      for (auto& e : op.project_list) nc.push_back(substitute(e, cols, memo));
      cols = nc;
    }
  }
}

PipelineDesc generate_join(const Operator& j, const std::vector<DType>& lt_in, const std::vector<DType>& rt_in, const std::vector<bool>& lvalid_in,
                           const std::vector<bool>& rvalid_in, const JoinFusion* fu, const JoinFusion* fub) {
  if (j.kind != OpKind::HashJoin) throw CometError("internal: generate_join on a non-join");
  if (j.left_keys.size() != j.right_keys.size() || j.left_keys.empty()) throw CometError("HashJoin needs matching, non-empty key lists");
  int mode = 0;
  bool keep_left = false, keep_right = false;   // outer joins: which side's unmatched rows survive
  switch (j.join_type) {
    case JoinType::Inner: break;
    case JoinType::LeftSemi: mode = 1; break;
    case JoinType::LeftAnti: mode = 2; break;
    case JoinType::LeftOuter: keep_left = true; break;
    case JoinType::RightOuter: keep_right = true; break;
    case JoinType::FullOuter: keep_left = keep_right = true; break;
    default: throw CometError("HashJoin type " + std::to_string((int)j.join_type) + " is not supported by the MI355X native engine yet");
  }
  if (j.null_aware_anti) throw CometError("null-aware anti join is not supported by the MI355X native engine yet");
  const bool build_left = j.build_side == BuildSide::Left;
  // LeftSemi / LeftAnti built on the LEFT: the output is a subset of the BUILD rows — the probe pass only marks the build rows
  // it matched (all of them: MODE 0 walks the whole chain), the tail pass then emits the marked (semi) or unmarked (anti) ones
  const bool build_only = mode != 0 && build_left;
  const bool keep_matched = build_only && mode == 1;
  if (build_only) mode = 0;
  const bool outer_probe = !build_only && (build_left ? keep_right : keep_left);    // the probe side is the preserved one
  const bool outer_build = build_only || (build_left ? keep_left : keep_right);
  // The build side: the materialised child — or (round 6), with a fused BUILD chain (fub), the chain's SOURCE table: build rows are source rows, the chain's
  // Filters are part of P::bvalid (a row that fails them is in no table, no bitmap and no outer join's tail: P::bkeep), its columns expressions over the source.
  const std::vector<DType>& bt = fub ? fub->src_types : (build_left ? lt_in : rt_in);
  const std::vector<bool>& bv_src = fub ? fub->src_valid : (build_left ? lvalid_in : rvalid_in);
  if (fub && bt.size() != bv_src.size()) throw CometError("internal: fused build source validity arity mismatch");
  std::vector<bool> bv(bv_src);
  if (fub)
    for (auto& p : fub->preds)
      if (p->kind == ExprKind::IsNotNull && p->children.size() == 1 && p->children[0]->kind == ExprKind::Bound && p->children[0]->bound_index >= 0 &&
          (size_t)p->children[0]->bound_index < bv.size())
        bv[(size_t)p->children[0]->bound_index] = false;
  // The probe side: the materialised child — or, with a fused probe chain (JoinFusion), the chain's SOURCE table; the child's columns
  // are then expressions over the source columns (fu->cols), its Filters conjuncts over them (fu->preds, P::pkeep).
  const std::vector<DType>& pt = fu ? fu->src_types : (build_left ? rt_in : lt_in);      // PHYSICAL probe columns
  const std::vector<bool>& pv_src = fu ? fu->src_valid : (build_left ? rvalid_in : lvalid_in);
  if (fu && pt.size() != pv_src.size()) throw CometError("internal: fused probe source validity arity mismatch");
  // a fused chain's isnotnull(<source column>) conjuncts: every probe row that passes P::pkeep holds a value there — keys, condition and output read
  // no validity bit of such a column (P::pkeep itself still does)
  std::vector<bool> pv(pv_src);
  if (fu)
    for (auto& p : fu->preds)
      if (p->kind == ExprKind::IsNotNull && p->children.size() == 1 && p->children[0]->kind == ExprKind::Bound && p->children[0]->bound_index >= 0 &&
          (size_t)p->children[0]->bound_index < pv.size())
        pv[(size_t)p->children[0]->bound_index] = false;
  const int nb = (int)bt.size(), np = (int)pt.size();
  const int nprobe_logical = fu ? (int)fu->cols.size() : np;
  const int nbuild_logical = fub ? (int)fub->cols.size() : nb;
  const int nl = build_left ? nbuild_logical : nprobe_logical, nr = build_left ? nprobe_logical : nbuild_logical;
  if (nb + np > COMET_MAX_IN) throw CometError("too many columns for one GPU hash join");
  const std::vector<ExprP>& bkeys_logical = build_left ? j.left_keys : j.right_keys;
  const std::vector<ExprP>& pkeys_logical = build_left ? j.right_keys : j.left_keys;
  std::vector<ExprP> pkeys, bkeys;
  {
    std::map<const Expr*, ExprP> memo;
    for (auto& k : pkeys_logical) pkeys.push_back(fu ? substitute(k, fu->cols, memo) : k);
  }
  {
    std::map<const Expr*, ExprP> memo;
    for (auto& k : bkeys_logical) bkeys.push_back(fub ? substitute(k, fub->cols, memo) : k);
  }
  // physical combined schema = kernel argument order: build columns (row i), then probe columns (row j)
  std::vector<DType> ct(bt);
  ct.insert(ct.end(), pt.begin(), pt.end());
  std::vector<bool> cv(bv);
  cv.insert(cv.end(), pv.begin(), pv.end());
  auto locate_combined = [=](int c) { return std::make_pair(c, std::string(c < nb ? "i" : "j")); };
  auto mk_bound = [](int idx, const DType& t) {
    auto b = std::make_shared<Expr>();
    b->kind = ExprKind::Bound;
    b->proto_tag = 3;
    b->bound_index = idx;
    b->dtype = t;
    b->has_dtype = true;
    return ExprP(b);
  };
  std::function<ExprP(const ExprP&, int)> shift_bound = [&](const ExprP& e, int by) -> ExprP {
    if (e->kind == ExprKind::Bound) {
      auto n = std::make_shared<Expr>(*e);
      n->bound_index = e->bound_index + by;
      return n;
    }
    if (e->children.empty()) return e;
    auto n = std::make_shared<Expr>(*e);
    for (auto& c : n->children) c = shift_bound(c, by);
    return n;
  };
  // logical column c of left ++ right (what keys, condition and output refer to) as an expression over the physical schema
  std::vector<ExprP> phys;
  for (int c = 0; c < nl + nr; c++) {
    const bool is_left = c < nl;
    const int local = is_left ? c : c - nl;
    if (is_left == build_left) phys.push_back(fub ? fub->cols[(size_t)local] : mk_bound(local, bt[(size_t)local]));
    else if (!fu) phys.push_back(mk_bound(nb + local, pt[(size_t)local]));
    else phys.push_back(shift_bound(fu->cols[(size_t)local], nb));
  }

  PipelineDesc d;
  SiteScope site_scope(d);
  d.sink = SinkKind::Output;
  d.R = 1;
  d.op_names.push_back("HashJoin");
  std::ostringstream src, ex;
  src << "// generated by datafusion-comet_amd codegen — hash join\n#include \"comet_device.hpp\"\nusing namespace comet;\n";
  src << "struct P {\n  static constexpr int R = 1;\n  static constexpr int MODE = " << mode << ";\n";
  src << "  static constexpr bool OUTER_PROBE = " << (outer_probe ? "true" : "false") << ", OUTER_BUILD = " << (outer_build ? "true" : "false")
      << ", BUILD_KEEP_MATCHED = " << (keep_matched ? "true" : "false") << ", BUILD_ONLY = " << (build_only ? "true" : "false") << ";\n";
  // semi / anti joins that keep PROBE rows and have no residual condition only ask whether a key exists on the build side: one build row
  // per run of equal keys is enough (comet_device.hpp "Runs of equal keys")
  const bool dedup_build = mode != 0 && !j.join_condition;
  src << "  static constexpr bool DEDUP_BUILD = " << (dedup_build ? "true" : "false") << ";\n";
  src << "  static constexpr bool HAS_COND = " << (j.join_condition ? "true" : "false") << ";\n";      // a residual condition beside the keys

  // key words of one side
  auto key_fn = [&](const char* name, const char* rowvar, const std::vector<DType>& types, const std::vector<bool>& valid, int base,
                    const std::vector<ExprP>& keys, bool want_hash) {
    Gen g(types, valid);
    g.locate = [base, rowvar](int idx) { return std::make_pair(base + idx, std::string(rowvar)); };
    std::string okall;
    std::vector<std::string> words;
    for (auto& ke : keys) {
      Val v = g.named(g.gen(ke));
      okall = Gen::and_ok(okall, v.ok);
      switch (v.rep) {
        case Rep::B: words.push_back("(u64)(" + v.v + " ? 1 : 0)"); break;
        case Rep::I32: case Rep::I64: words.push_back("(u64)(i64)" + v.v); break;
        case Rep::I128: words.push_back("comet::lo64(" + v.v + ")"); words.push_back("comet::hi64(" + v.v + ")"); break;
        case Rep::F64: words.push_back("(u64)__double_as_longlong(comet::normalize_nan_zero_f64(" + v.v + "))"); break;
        case Rep::F32: words.push_back("(u64)(u32)__float_as_int(comet::normalize_nan_zero_f32(" + v.v + "))"); break;
        case Rep::STR: words.push_back(v.v + ".a"); words.push_back(v.v + ".b"); break;
      }
    }
    src << "  static __device__ __forceinline__ " << (want_hash ? "u64 " : "bool ") << name << "(const CometKParams& prm, i64 " << rowvar << ") {\n"
        << "    bool k[R] = {true};\n" << g.decls << g.body();
    if (want_hash) {
      src << "    u64 kw[" << words.size() << "];\n";
      for (size_t w = 0; w < words.size(); w++) {
        std::string e = words[w];
        for (size_t p0 = e.find("[r]"); p0 != std::string::npos; p0 = e.find("[r]")) e.replace(p0, 3, "[0]");
        src << "    kw[" << w << "] = " << e << ";\n";
      }
      src << "    return comet::hash_key<" << words.size() << ">(kw);\n  }\n";
      if (std::string(name) == "phash") {
        // a single integer key: its value handed out too — the probe asks the build side's key bitmap before it touches the table
        const bool one_int = keys.size() == 1 && words.size() == 1 && (words[0].rfind("(u64)(i64)", 0) == 0);
        src << "  static constexpr bool KEYMAP = " << (one_int ? "true" : "false") << ";\n";
        src << "  static __device__ __forceinline__ u64 pkey0(const CometKParams& prm, i64 " << rowvar << ") {\n";
        if (one_int) {
          std::string e = words[0];
          for (size_t p0 = e.find("[r]"); p0 != std::string::npos; p0 = e.find("[r]")) e.replace(p0, 3, "[0]");
          src << "    bool k[R] = {true};\n" << g.decls << g.body() << "    return " << e << ";\n  }\n";
        } else {
          src << "    (void)prm; (void)" << rowvar << "; return 0;\n  }\n";
        }
      }
      if (std::string(name) == "bhash") {
        // the same key words handed out (run detection compares neighbouring build rows word by word: exact, unlike their hashes)
        src << "  static constexpr int NKW = " << words.size() << ";\n";
        src << "  static __device__ __forceinline__ void bkeys(const CometKParams& prm, i64 " << rowvar << ", u64* kw) {\n"
            << "    bool k[R] = {true};\n" << g.decls << g.body();
        for (size_t w = 0; w < words.size(); w++) {
          std::string e = words[w];
          for (size_t p0 = e.find("[r]"); p0 != std::string::npos; p0 = e.find("[r]")) e.replace(p0, 3, "[0]");
          src << "    kw[" << w << "] = " << e << ";\n";
        }
        src << "  }\n";
      }
    } else {
      std::string e = okall.empty() ? "true" : okall;
      for (size_t p0 = e.find("[r]"); p0 != std::string::npos; p0 = e.find("[r]")) e.replace(p0, 3, "[0]");
      src << "    return " << e << ";\n  }\n";
    }
    return (int)words.size();
  };
  if (fub && !fub->preds.empty()) {
    // P::bvalid = the fused chain's Filters (every predicate column loaded up front, as in P::pkeep) AND the keys' validity
    Gen g(bt, bv_src);
    g.eager_loads = true;
    g.locate = [](int idx) { return std::make_pair(idx, std::string("i")); };
    for (auto& p : fub->preds) {
      g.add_predicate(p);
      ex << "  build filter (fused into the build passes): " << explain_expr(p) << "\n";
    }
    src << "  static __device__ __forceinline__ bool bkeep(const CometKParams& prm, i64 i) {\n    bool k[R] = {true};\n" << g.decls << g.body() << "    return k[0];\n  }\n";
    key_fn("bkeysvalid", "i", bt, bv, 0, bkeys, false);
    src << "  static __device__ __forceinline__ bool bvalid(const CometKParams& prm, i64 i) { return bkeep(prm, i) && bkeysvalid(prm, i); }\n";
  } else {
    src << "  static __device__ __forceinline__ bool bkeep(const CometKParams&, i64) { return true; }\n";
    key_fn("bvalid", "i", bt, bv, 0, bkeys, false);
  }
  int nwb = key_fn("bhash", "i", bt, bv, 0, bkeys, true);
  key_fn("pvalid", "j", pt, pv, nb, pkeys, false);
  int nwp = key_fn("phash", "j", pt, pv, nb, pkeys, true);
  if (nwb != nwp) throw CometError("HashJoin key types differ between the two sides");

  // P::pkeep(j): the Filters of a fused probe chain, conjunct by conjunct (columns load only for rows still alive); a row that fails is
  // not part of the probe side at all (outer / anti joins do not emit it either)
  if (fu && !fu->preds.empty()) {
    Gen g(pt, pv_src);
    // every load of the chain's predicates up front, no "only for rows still alive" staging: a staged load sits in a lane-dependent branch and is waited for
    // inside it, so the probe tile's sixteen rows per thread became sixteen CHAINS of dependent latencies; straight-line loads of sixteen rows are in flight together
    static const bool eager_keep = getenv("COMET_JOIN_EAGER_KEEP") == nullptr || atoi(getenv("COMET_JOIN_EAGER_KEEP")) != 0;
    g.eager_loads = eager_keep;
    g.locate = [nb](int idx) { return std::make_pair(nb + idx, std::string("j")); };
    for (auto& p : fu->preds) {
      g.add_predicate(p);
      ex << "  probe filter (fused into the probe kernel): " << explain_expr(p) << "\n";
    }
    src << "  static __device__ __forceinline__ bool pkeep(const CometKParams& prm, i64 j) {\n    bool k[R] = {true};\n" << g.decls << g.body() << "    return k[0];\n  }\n";
  } else {
    src << "  static __device__ __forceinline__ bool pkeep(const CometKParams&, i64) { return true; }\n";
  }
  {
    // match(i, j): every key pair equal (NULL never equals) and the residual condition TRUE
    Gen g(ct, cv);
    g.locate = locate_combined;
    std::string cond;
    for (size_t k = 0; k < j.left_keys.size(); k++) {
      std::map<const Expr*, ExprP> memo;
      // right keys are bound to the right child's schema: shift their column indices by nl
      std::function<ExprP(const ExprP&)> shift = [&](const ExprP& e) -> ExprP {
        if (e->kind == ExprKind::Bound) {
          auto n = std::make_shared<Expr>(*e);
          n->bound_index = e->bound_index + nl;
          return n;
        }
        if (e->children.empty()) return e;
        auto n = std::make_shared<Expr>(*e);
        for (auto& c : n->children) c = shift(c);
        return n;
      };
      std::map<const Expr*, ExprP> m2;
      Val a = g.gen(substitute(j.left_keys[k], phys, m2));
      Val b = g.gen(substitute(shift(j.right_keys[k]), phys, m2));
      Val c = g.compare(ExprKind::Eq, a, b);
      cond = Gen::and_ok(cond, Gen::and_ok(c.ok, c.v));
    }
    if (j.join_condition) {
      std::map<const Expr*, ExprP> m2;
      Val c = g.gen(substitute(j.join_condition, phys, m2));
      if (c.rep != Rep::B) throw CometError("join condition must be boolean");
      cond = Gen::and_ok(cond, Gen::and_ok(c.ok, c.v));
      ex << "  condition: " << explain_expr(j.join_condition) << "\n";
    }
    Val r;
    r.rep = Rep::B;
    r.v = cond;
    r = g.named(r);
    std::string e = r.v;
    for (size_t p0 = e.find("[r]"); p0 != std::string::npos; p0 = e.find("[r]")) e.replace(p0, 3, "[0]");
    src << "  static __device__ __forceinline__ bool match(const CometKParams& prm, i64 i, i64 j) {\n    bool k[R] = {true};\n"
        << g.decls << g.body() << "    return " << e << ";\n  }\n";
  }
  {
    // cond(i, j): the residual condition ALONE — what the bucket table still has to ask of a build row whose one-word key its entry settled
    if (j.join_condition) {
      Gen g(ct, cv);
      g.locate = locate_combined;
      std::map<const Expr*, ExprP> m2;
      Val c = g.gen(substitute(j.join_condition, phys, m2));
      Val r;
      r.rep = Rep::B;
      r.v = Gen::and_ok(c.ok, c.v);
      r = g.named(r);
      std::string e = r.v;
      for (size_t p0 = e.find("[r]"); p0 != std::string::npos; p0 = e.find("[r]")) e.replace(p0, 3, "[0]");
      src << "  static __device__ __forceinline__ bool cond(const CometKParams& prm, i64 i, i64 j) {\n    bool k[R] = {true};\n"
          << g.decls << g.body() << "    return " << e << ";\n  }\n";
    } else {
      src << "  static __device__ __forceinline__ bool cond(const CometKParams&, i64, i64) { return true; }\n";
    }
  }
  {
    // emit(i, j, pos): output = left columns then right columns (Inner / outer); left columns only (Semi/Anti).
    // Outer joins add emit_probe_only(j, pos) / emit_build_only(i, pos): the other side's columns are NULL.
    const int nout = (mode == 0 && !build_only) ? nl + nr : nl;
    if (nout * 2 + kOutFirstCol > 44) throw CometError("too many output columns for one GPU hash join");
    auto is_str_bound = [&](const ExprP& e) {
      return e->kind == ExprKind::Bound && e->bound_index >= 0 && (size_t)e->bound_index < ct.size() &&
             (ct[(size_t)e->bound_index].id == TypeId::String || ct[(size_t)e->bound_index].id == TypeId::Bytes);
    };
    {
      Gen g0(ct, cv);             // types and nullability of the output columns (a fused probe column is a computed expression)
      g0.locate = locate_combined;
      for (int c = 0; c < nout; c++) {
        const bool is_left = c < nl;
        const ExprP& e = phys[(size_t)c];
        OutCol oc;
        if (is_str_bound(e)) {
          oc.type = ct[(size_t)e->bound_index];
          oc.nullable = cv[(size_t)e->bound_index] || (is_left ? keep_right : keep_left);
          oc.gather_src = e->bound_index;     // Utf8 payload: row index now, gathered after the emit (index into build ++ probe columns)
        } else {
          Val v = g0.gen(e);
          if (v.rep == Rep::STR) throw CometError("a computed Utf8 column below a fused join probe is not supported");
          oc.type = v.t;
          oc.nullable = !v.ok.empty() || (is_left ? keep_right : keep_left);   // the non-preserved side is NULL-extended
        }
        d.out_cols.push_back(oc);
      }
    }
    // variant 0 = both rows, 1 = probe row only, 2 = build row only
    for (int variant = 0; variant < 3; variant++) {
      if (variant == 1 && !outer_probe) continue;
      if (variant == 2 && !outer_build) continue;
      Gen g(ct, cv);
      g.locate = locate_combined;
      for (int c = 0; c < nout; c++) {
        const bool in_build = ((c < nl) == build_left);
        const bool absent = (variant == 1 && in_build) || (variant == 2 && !in_build);
        const ExprP& e = phys[(size_t)c];
        std::string vb = "prm.out[" + std::to_string(kOutFirstCol + 2 * c) + "]", ob = "prm.out[" + std::to_string(kOutFirstCol + 2 * c + 1) + "]";
        if (d.out_cols[c].gather_src >= 0) {
          // Utf8 payload: the source row (build row i or probe row j); NULL when the side is absent or the value is NULL
          const int pc = e->bound_index;
          auto loc = locate_combined(pc);
          if (absent) {
            g.stmt("((u32*)" + vb + ")[pos] = 0u;");
            g.stmt("((u8*)" + ob + ")[pos] = 0;");
          } else {
            g.in_used[(size_t)pc] = true;
            g.stmt("((u32*)" + vb + ")[pos] = (u32)" + loc.second + ";");
            if (d.out_cols[c].nullable)
              g.stmt("((u8*)" + ob + ")[pos] = " + (cv[(size_t)pc] ? "comet::ld_valid(prm.in[" + std::to_string(loc.first) + "], " + loc.second + ") ? 1 : 0" : std::string("1")) + ";");
          }
          continue;
        }
        const char* st = store_ctype(d.out_cols[c].type);
        if (absent) {
          g.stmt("((" + std::string(st) + "*)" + vb + ")[pos] = (" + st + ")0;");
          g.stmt("((u8*)" + ob + ")[pos] = 0;");
          continue;
        }
        Val v = g.gen(e);
        std::string val = v.v;
        if (v.t.id == TypeId::Decimal) val = v.rep == Rep::I128 ? v.v : "(i128)" + v.v;
        else if (v.t.id == TypeId::Bool) val = "(u8)(" + v.v + " ? 1 : 0)";
        else val = std::string("(") + st + ")" + v.v;
        if (!v.ok.empty()) {
          g.stmt("((" + std::string(st) + "*)" + vb + ")[pos] = " + v.ok + " ? " + val + " : (" + st + ")0;");
          g.stmt("((u8*)" + ob + ")[pos] = " + v.ok + " ? 1 : 0;");
        } else {
          g.stmt("((" + std::string(st) + "*)" + vb + ")[pos] = " + val + ";");
          if (d.out_cols[c].nullable) g.stmt("((u8*)" + ob + ")[pos] = 1;");
        }
      }
      const char* sig = variant == 0 ? "emit(const CometKParams& prm, i64 i, i64 j, i64 pos)"
                        : variant == 1 ? "emit_probe_only(const CometKParams& prm, i64 j, i64 pos)"
                                       : "emit_build_only(const CometKParams& prm, i64 i, i64 pos)";
      src << "  static __device__ __forceinline__ void " << sig << " {\n    bool k[R] = {true};\n" << g.decls << g.body() << "  }\n";
    }
    if (!outer_probe) src << "  static __device__ __forceinline__ void emit_probe_only(const CometKParams&, i64, i64) {}\n";
    if (!outer_build) src << "  static __device__ __forceinline__ void emit_build_only(const CometKParams&, i64, i64) {}\n";
  }
  src << "};\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jbuild(const CometKParams prm) { comet::join_build_body<P>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jbcnt(const CometKParams prm) { comet::join_build_count_body<P>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_pack(const CometKParams prm) { comet::pack_validity_body((const u8*)prm.out[0], (u8*)prm.out[1], prm.n); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jbcount(const CometKParams prm) { comet::join_build_unmatched_count_body<P>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jbscan(const CometKParams prm) { comet::tile_scan_body((u64*)prm.out[comet::kJoinBuildTiles], prm.iarg[3]); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jbemit(const CometKParams prm) { comet::join_build_unmatched_emit_body<P>(prm); }\n";
  // single-pass probes (comet_device.hpp template D'): the chained global table, or an LDS table for small build sides
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jprobe(const CometKParams prm) { comet::join_probe_fused_body<P>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jprobe_km(const CometKParams prm) { comet::join_probe_fused_body<P, true>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256, COMET_WAVES_JLDS) void k_jlds(const CometKParams prm) { comet::join_probe_lds_body<P>(prm); }\n";
  // the key bitmap's two helpers: a sample of the probe side through the finished table (does it pay?), and the bitmap's own build pass
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jsample(const CometKParams prm) { comet::join_sample_body<P>(prm); }\n";
  // the direct map of a unique integer key (no hash table): build rows in key order, and the probe over it
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jdrows(const CometKParams prm) { comet::join_direct_rows_body<P>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256, COMET_WAVES_JDPROBE) void k_jdprobe(const CometKParams prm) { comet::join_probe_direct_body<P>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jbmap(const CometKParams prm) { comet::join_keymap_build_body<P>(prm); }\n";
  // the bucket table (comet_device.hpp template D''): partition passes, the LDS build, the probes; and the bitmap-only semi / anti join
  src << "extern \"C\" __global__ __launch_bounds__(1024) void k_jphist(const CometKParams prm) { comet::join_part_hist_body<P>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(1024) void k_jpscat(const CometKParams prm) { comet::join_part_scatter_body<P>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jtbuild(const CometKParams prm) { comet::join_table_build_body<P>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256, COMET_WAVES_JPROBE_B) void k_jprobe_b(const CometKParams prm) { comet::join_probe_bucket_body<P>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256, COMET_WAVES_JPROBE_BKM) void k_jprobe_bkm(const CometKParams prm) { comet::join_probe_bucket_body<P, true>(prm); }\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_jsample_b(const CometKParams prm) { comet::join_sample_bucket_body<P>(prm); }\n";
  if (dedup_build) src << "extern \"C\" __global__ __launch_bounds__(256) void k_jprobe_bm(const CometKParams prm) { comet::join_probe_bitmap_body<P>(prm); }\n";
  d.kernels = {"k_jbuild", "k_jbcnt", "k_pack", "k_jbcount", "k_jbscan", "k_jbemit", "k_jprobe", "k_jprobe_km", "k_jlds", "k_jsample", "k_jbmap", "k_jdrows", "k_jdprobe",
               "k_jphist", "k_jpscat", "k_jtbuild", "k_jprobe_b", "k_jprobe_bkm", "k_jsample_b"};
  if (dedup_build) d.kernels.push_back("k_jprobe_bm");
  d.join_dedup_build = dedup_build;
  d.join_outer_build = outer_build;
  d.join_build_only = build_only;
  const char* jt_name = build_only ? (keep_matched ? "LeftSemi" : "LeftAnti") : keep_left && keep_right ? "FullOuter" : keep_left ? "LeftOuter" : keep_right ? "RightOuter" : mode == 0 ? "Inner" : mode == 1 ? "LeftSemi" : "LeftAnti";
  ex << "  hash join: " << jt_name << ", build " << (build_left ? "left" : "right") << ", "
     << j.left_keys.size() << " key(s)" << (fu ? ", probe side fused with its Filter / Projection chain" : "") << (fub ? ", build side fused with its chain" : "") << "\n";
  d.in_types = ct;
  d.source = with_optional_headers(src.str());
  d.explain = ex.str();
  return d;
}

// ---------------------------------------------------------------------------------------------
// Sort keys (Sort operator, planner.rs:1488-1522 → DataFusion SortExec): every row gets an order-preserving byte string —
// per sort expression one NULL-ordering byte (NULLS FIRST: NULL = 0x00 < 0x01; NULLS LAST: NULL = 0x01 > 0x00) followed by the
// value big-endian with the sign bit flipped (floats through the IEEE totalOrder key), inverted for DESC; NULL rows carry zero
// value bytes.  The bytes are stored as PLANES (plane b = byte b of every row) so that the executor's LSD radix sort reads one
// plane per pass.  prm.out[0] = planes (W × n bytes), prm.n = rows.
// ---------------------------------------------------------------------------------------------
PipelineDesc generate_sort_keys(const Operator& sort, const std::vector<DType>& types, const std::vector<bool>& valid) {
  if (sort.kind != OpKind::Sort || sort.sort_orders.empty()) throw CometError("Sort needs at least one sort expression");
  PipelineDesc d;
  SiteScope site_scope(d);
  d.sink = SinkKind::Output;
  d.R = 1;
  d.op_names.push_back("Sort");
  d.in_types = types;
  std::ostringstream src, ex;
  src << "// generated by datafusion-comet_amd codegen — sort keys\n#include \"comet_device.hpp\"\nusing namespace comet;\n";
  src << "struct P {\n  static constexpr int R = 1;\n";
  Gen g(types, valid);
  g.locate = [](int idx) { return std::make_pair(idx, std::string("i")); };
  int W = 0;
  std::string dyn;                   // "+ prm.iarg[1] + …": plane offset contributed by the Utf8 keys seen so far
  auto off = [&](int add) { return "((i64)" + std::to_string(W + add) + dyn + ")"; };
  std::vector<std::string> stores;   // statements writing the bytes of one row
  for (auto& k : sort.sort_orders) {
    const std::string inv = k.descending ? "~" : "";
    const bool str_col = k.child->kind == ExprKind::Bound && k.child->bound_index >= 0 && (size_t)k.child->bound_index < types.size() &&
                         (types[(size_t)k.child->bound_index].id == TypeId::String || types[(size_t)k.child->bound_index].id == TypeId::Bytes);
    if (str_col) {
      // Utf8 column of any length: NULL-ordering byte, the bytes zero-padded to the column's longest value, then the length (big endian) —
      // unsigned byte order like Spark's UTF8String.compareTo; equal padded bytes are ordered by length, so "ab" < "ab\0"
      if (d.sort_str_cols.size() >= 6) throw CometError("Sort: more than 6 Utf8 sort keys are not supported");
      const int idx = k.child->bound_index;
      const std::string slot = "prm.iarg[" + std::to_string(1 + d.sort_str_cols.size()) + "]";
      Val valid = g.str_col_validity(idx);
      const std::string ok = valid.ok.empty() ? "true" : valid.ok;
      const std::string c = "prm.in[" + std::to_string(idx) + "]";
      stores.push_back("K[" + off(0) + " * n + i] = (u8)(" + ok + " ? " + (k.nulls_last ? "0" : "1") + " : " + (k.nulls_last ? "1" : "0") + ");");
      stores.push_back("{ const i32* off_ = (const i32*)" + c + ".data; const i64 j_ = " + c + ".offset + i; const i32 lo_ = off_[j_], len_ = off_[j_ + 1] - lo_;"
                       " const u8* p_ = (const u8*)" + c + ".aux + lo_; const i64 L_ = " + slot + " - 4; const i64 base_ = " + off(1) + ";"
                       " for (i64 q_ = 0; q_ < L_; q_++) K[(base_ + q_) * n + i] = (u8)(" + ok + " ? " + inv + "(u8)(q_ < len_ ? p_[q_] : 0) : 0);"
                       " for (int b_ = 0; b_ < 4; b_++) K[(base_ + L_ + b_) * n + i] = (u8)(" + ok + " ? " + inv + "(u8)((u32)len_ >> (24 - 8 * b_)) : 0); }");
      W += 1;
      dyn += " + " + slot;
      d.sort_str_cols.push_back(idx);
      ex << "  sort key: " << explain_expr(k.child) << (k.descending ? " DESC" : " ASC") << (k.nulls_last ? " NULLS LAST" : " NULLS FIRST") << "\n";
      continue;
    }
    Val v = g.named(g.gen(k.child));
    const std::string ok = v.ok.empty() ? "true" : v.ok;
    // NULL-ordering byte
    stores.push_back("K[" + off(0) + " * n + i] = (u8)(" + ok + " ? " + (k.nulls_last ? "0" : "1") + " : " + (k.nulls_last ? "1" : "0") + ");");
    W++;
    int nb = 0;
    std::string word;   // unsigned order-preserving word(s)
    switch (v.rep) {
      case Rep::B: nb = 1; word = "(u64)(" + v.v + " ? 1 : 0)"; break;
      case Rep::I32: nb = 4; word = "(u64)((u32)(i32)" + v.v + " ^ 0x80000000u)"; break;
      case Rep::I64: nb = v.t.id == TypeId::Decimal ? 16 : 8; word = "((u64)(i64)" + v.v + " ^ 0x8000000000000000ull)"; break;
      case Rep::F32: nb = 4; word = "(u64)((u32)comet::f32_total_key(" + v.v + ") ^ 0x80000000u)"; break;
      case Rep::F64: nb = 8; word = "((u64)comet::f64_total_key(" + v.v + ") ^ 0x8000000000000000ull)"; break;
      case Rep::I128: nb = 16; break;
      case Rep::STR: {
        // a computed string (≤ 15 bytes, packed): its 15 zero-padded bytes, then the length
        for (int b = 0; b < 15; b++) {
          const std::string byte = b < 8 ? "(" + v.v + ".a >> " + std::to_string(8 * b) + ")" : "(" + v.v + ".b >> " + std::to_string(8 * (b - 8)) + ")";
          stores.push_back("K[" + off(b) + " * n + i] = (u8)(" + ok + " ? " + inv + "(u8)" + byte + " : 0);");
        }
        stores.push_back("K[" + off(15) + " * n + i] = (u8)(" + ok + " ? " + inv + "(u8)(" + v.v + ".b >> 56) : 0);");
        W += 16;
        ex << "  sort key: " << explain_expr(k.child) << (k.descending ? " DESC" : " ASC") << (k.nulls_last ? " NULLS LAST" : " NULLS FIRST") << "\n";
        continue;
      }
      default: throw CometError("Sort on " + v.t.str() + " is not supported by the MI355X native engine yet");
    }
    if (nb == 16) {
      // 128-bit: hi word (sign flipped) then lo word; a narrow decimal (i64 rep) sign-extends
      const std::string v128 = v.rep == Rep::I128 ? v.v : "(i128)(i64)" + v.v;
      const std::string hi = "(comet::hi64(" + v128 + ") ^ 0x8000000000000000ull)", lo = "comet::lo64(" + v128 + ")";
      for (int b = 0; b < 8; b++)
        stores.push_back("K[" + off(b) + " * n + i] = (u8)(" + ok + " ? (" + inv + "(" + hi + " >> " + std::to_string(56 - 8 * b) + ")) : 0);");
      for (int b = 0; b < 8; b++)
        stores.push_back("K[" + off(8 + b) + " * n + i] = (u8)(" + ok + " ? (" + inv + "(" + lo + " >> " + std::to_string(56 - 8 * b) + ")) : 0);");
    } else {
      for (int b = 0; b < nb; b++)
        stores.push_back("K[" + off(b) + " * n + i] = (u8)(" + ok + " ? (" + inv + "(" + word + " >> " + std::to_string(8 * (nb - 1 - b)) + ")) : 0);");
    }
    W += nb;
    ex << "  sort key: " << explain_expr(k.child) << (k.descending ? " DESC" : " ASC") << (k.nulls_last ? " NULLS LAST" : " NULLS FIRST") << "\n";
  }
  if (W > 255) throw CometError("Sort key wider than 255 bytes");
  for (auto& st : stores) g.stmt(st);
  std::string body = g.decls + g.body();
  for (size_t p0 = body.find("[r]"); p0 != std::string::npos; p0 = body.find("[r]")) body.replace(p0, 3, "[0]");
  src << "  static __device__ __forceinline__ void keys(const CometKParams& prm, i64 i) {\n    u8* K = (u8*)prm.out[0];\n    const i64 n = prm.n;\n"
      << "    bool k[R] = {true};\n" << body << "  }\n};\n";
  src << "extern \"C\" __global__ __launch_bounds__(256) void k_sortkey(const CometKParams prm) {\n"
      << "  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < prm.n; i += (i64)gridDim.x * 256) P::keys(prm, i);\n}\n";
  d.kernels = {"k_sortkey"};
  d.sort_key_bytes = W;
  if (sort.fetch >= 0) ex << "  fetch " << sort.fetch << "\n";
  d.source = with_optional_headers(src.str());
  d.explain = ex.str();
  return d;
}

}  // namespace comet
