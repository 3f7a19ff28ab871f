// Hand-written proto3 wire decoder for Comet's plan messages (no protoc / libprotobuf in this image).
// Field numbers follow native/proto/src/proto/{operator,expr,types,literal,config,metric}.proto; the
// reference decodes the same bytes with prost in native/core/src/execution/serde.rs:45-58.
#include <cstdlib>
#include <cstring>

#include "plan.hpp"

namespace comet {

namespace {

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  int depth = 0;     // message nesting: prost, which decodes these bytes in the reference, stops at 100 levels ("recursion limit reached")
  Reader(const uint8_t* d, size_t n) : p(d), end(d + n) {}
  bool done() const { return p >= end; }
  uint64_t varint() {
    uint64_t v = 0;
    int shift = 0;
    while (true) {
      if (p >= end) throw CometError("protobuf: truncated varint");
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) break;
      shift += 7;
      if (shift > 63) throw CometError("protobuf: varint too long");
    }
    return v;
  }
  // returns field number, sets wire type
  int tag(int& wt) {
    uint64_t t = varint();
    wt = (int)(t & 7);
    return (int)(t >> 3);
  }
  Reader sub() {
    uint64_t n = varint();
    if ((uint64_t)(end - p) < n) throw CometError("protobuf: truncated length-delimited field");
    Reader r(p, (size_t)n);
    r.depth = depth + 1;
    if (r.depth > 100) throw CometError("failed to decode Protobuf message: recursion limit reached");
    p += n;
    return r;
  }
  std::string bytes() {
    Reader r = sub();
    return std::string((const char*)r.p, (size_t)(r.end - r.p));
  }
  uint32_t fixed32() {
    if (end - p < 4) throw CometError("protobuf: truncated fixed32");
    uint32_t v;
    memcpy(&v, p, 4);
    p += 4;
    return v;
  }
  uint64_t fixed64() {
    if (end - p < 8) throw CometError("protobuf: truncated fixed64");
    uint64_t v;
    memcpy(&v, p, 8);
    p += 8;
    return v;
  }
  void skip(int wt) {
    switch (wt) {
      case 0: varint(); break;
      case 1: fixed64(); break;
      case 2: sub(); break;
      case 5: fixed32(); break;
      default: throw CometError("protobuf: unsupported wire type " + std::to_string(wt));
    }
  }
  // repeated scalar: packed (wt 2) or unpacked (wt 0)
  template <class F>
  void repeated_varint(int wt, F f) {
    if (wt == 2) {
      Reader r = sub();
      while (!r.done()) f(r.varint());
    } else {
      f(varint());
    }
  }
};

DType decode_datatype(Reader r) {
  DType d;
  d.id = TypeId::Bool;  // proto3 default enum value 0
  while (!r.done()) {
    int wt, f = r.tag(wt);
    if (f == 1 && wt == 0) {
      int id = (int)r.varint();
      d.id = (id >= 0 && id <= 17) ? (TypeId)id : TypeId::Unknown;
    } else if (f == 2 && wt == 2) {  // DataTypeInfo
      Reader info = r.sub();
      while (!info.done()) {
        int wt2, f2 = info.tag(wt2);
        if (f2 == 2 && wt2 == 2) {  // DecimalInfo
          Reader dec = info.sub();
          while (!dec.done()) {
            int wt3, f3 = dec.tag(wt3);
            if (f3 == 1 && wt3 == 0) d.precision = (int)(int32_t)dec.varint();
            else if (f3 == 2 && wt3 == 0) d.scale = (int)(int32_t)dec.varint();
            else dec.skip(wt3);
          }
        } else if (f2 == 3 && wt2 == 2) {  // ListInfo: element_type = 1, contains_null = 2
          Reader li = info.sub();
          DType el;
          el.id = TypeId::Bool;
          bool contains_null = false;
          while (!li.done()) {
            int wt3, f3 = li.tag(wt3);
            if (f3 == 1 && wt3 == 2) el = decode_datatype(li.sub());
            else if (f3 == 2 && wt3 == 0) contains_null = li.varint() != 0;
            else li.skip(wt3);
          }
          d.kids.assign(1, el);
          d.kid_names.assign(1, "element");
          d.kid_nullable.assign(1, contains_null ? 1 : 0);
        } else if (f2 == 4 && wt2 == 2) {  // MapInfo: key_type = 1, value_type = 2, value_contains_null = 3
          Reader mi = info.sub();
          DType kt, vt;
          kt.id = vt.id = TypeId::Bool;
          bool vnull = false;
          while (!mi.done()) {
            int wt3, f3 = mi.tag(wt3);
            if (f3 == 1 && wt3 == 2) kt = decode_datatype(mi.sub());
            else if (f3 == 2 && wt3 == 2) vt = decode_datatype(mi.sub());
            else if (f3 == 3 && wt3 == 0) vnull = mi.varint() != 0;
            else mi.skip(wt3);
          }
          DType entries = DType::of(TypeId::Struct);
          entries.kids = {kt, vt};
          entries.kid_names = {"key", "value"};
          entries.kid_nullable = {0, (char)(vnull ? 1 : 0)};
          d.kids.assign(1, entries);
          d.kid_names.assign(1, "entries");
          d.kid_nullable.assign(1, 0);
        } else if (f2 == 5 && wt2 == 2) {  // StructInfo: field_names = 1, field_datatypes = 2, field_nullable = 3 (packed or not)
          Reader si = info.sub();
          while (!si.done()) {
            int wt3, f3 = si.tag(wt3);
            if (f3 == 1 && wt3 == 2) d.kid_names.push_back(si.bytes());
            else if (f3 == 2 && wt3 == 2) d.kids.push_back(decode_datatype(si.sub()));
            else if (f3 == 3 && wt3 == 0) d.kid_nullable.push_back(si.varint() != 0 ? 1 : 0);
            else if (f3 == 3 && wt3 == 2) { Reader pk = si.sub(); while (!pk.done()) d.kid_nullable.push_back(pk.varint() != 0 ? 1 : 0); }
            else si.skip(wt3);
          }
          while (d.kid_nullable.size() < d.kids.size()) d.kid_nullable.push_back(1);
          while (d.kid_names.size() < d.kids.size()) d.kid_names.push_back("");
        } else {
          info.skip(wt2);
        }
      }
    } else {
      r.skip(wt);
    }
  }
  return d;
}

ExprP decode_expr(Reader r);

// BigInteger.toByteArray: big-endian two's complement (spark/.../serde/literals.scala:92-95;
// decoded by the reference at planner.rs:544-562).
i128 decode_be_twos_complement(const std::string& b) {
  if (b.empty()) return 0;
  if (b.size() > 16) {
    // allowed only if the extra leading bytes are pure sign extension
    uint8_t ext = (b[b.size() - 16] & 0x80) ? 0xff : 0x00;
    for (size_t i = 0; i + 16 < b.size(); i++)
      if ((uint8_t)b[i] != ext) throw CometError("Cannot parse decimal literal as i128");
  }
  u128 v = ((uint8_t)b[0] & 0x80) ? ~(u128)0 : 0;
  for (size_t i = 0; i < b.size(); i++) v = (v << 8) | (uint8_t)b[i];
  return (i128)v;
}

void decode_literal(Reader r, Expr& e) {
  while (!r.done()) {
    int wt, f = r.tag(wt);
    switch (f) {
      case 1: e.lit_bool = r.varint() != 0; e.lit_case = 1; break;
      case 2: case 3: case 4: e.lit_i64 = (int64_t)(int32_t)r.varint(); e.lit_case = f; break;
      case 5: e.lit_i64 = (int64_t)r.varint(); e.lit_case = 5; break;
      case 6: { uint32_t u = r.fixed32(); float fl; memcpy(&fl, &u, 4); e.lit_f64 = fl; e.lit_case = 6; break; }
      case 7: { uint64_t u = r.fixed64(); memcpy(&e.lit_f64, &u, 8); e.lit_case = 7; break; }
      case 8: case 9: e.lit_bytes = r.bytes(); e.lit_case = f; break;
      case 10: e.lit_dec = decode_be_twos_complement(r.bytes()); e.lit_case = 10; break;
      case 12: e.dtype = decode_datatype(r.sub()); e.has_dtype = true; break;
      case 13: e.lit_null = r.varint() != 0; break;
      default: r.skip(wt);
    }
  }
}

// MathExpr / BinaryExpr / UnaryExpr / Cast / CheckOverflow / BoundReference / In / If / CaseWhen share
// the shape "child exprs at low field numbers + a few scalars"; one generic walker with a per-kind
// field map keeps the decoder small.
void decode_expr_body(Reader r, Expr& e) {
  const ExprKind k = e.kind;
  while (!r.done()) {
    int wt, f = r.tag(wt);
    bool handled = false;
    switch (k) {
      case ExprKind::Unbound:
        if (f == 2 && wt == 2) { e.dtype = decode_datatype(r.sub()); e.has_dtype = true; handled = true; }
        break;
      case ExprKind::Bound:
        if (f == 1 && wt == 0) { e.bound_index = (int)(int32_t)r.varint(); handled = true; }
        else if (f == 2 && wt == 2) { e.dtype = decode_datatype(r.sub()); e.has_dtype = true; handled = true; }
        break;
      case ExprKind::Add: case ExprKind::Subtract: case ExprKind::Multiply: case ExprKind::Divide:
      case ExprKind::Remainder: case ExprKind::IntegralDivide:
        if ((f == 1 || f == 2) && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 4 && wt == 2) { e.dtype = decode_datatype(r.sub()); e.has_dtype = true; handled = true; }
        else if (f == 5 && wt == 0) { e.eval_mode = (EvalMode)r.varint(); handled = true; }
        else if (f == 6 && wt == 0) { e.check_divide_overflow = r.varint() != 0; handled = true; }
        break;
      case ExprKind::Cast:
        if (f == 1 && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 2 && wt == 2) { e.dtype = decode_datatype(r.sub()); e.has_dtype = true; handled = true; }
        else if (f == 3 && wt == 2) { e.func = r.bytes(); handled = true; }      // Cast.timezone (expr.proto:346) travels in `func`
        else if (f == 6 && wt == 0) { e.is_spark4_plus = r.varint() != 0; handled = true; }
        else if (f == 4 && wt == 0) { e.eval_mode = (EvalMode)r.varint(); handled = true; }
        break;
      case ExprKind::Hour: case ExprKind::Minute: case ExprKind::Second: case ExprKind::UnixTimestamp:
        if (f == 1 && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 2 && wt == 2) { e.func = r.bytes(); handled = true; }
        break;
      case ExprKind::TruncTimestamp:      // format = 1, child = 2 (in that order on the wire: children = [format, child]), timezone = 3
        if ((f == 1 || f == 2) && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 3 && wt == 2) { e.func = r.bytes(); handled = true; }
        break;
      case ExprKind::CheckOverflow:
        if (f == 1 && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 2 && wt == 2) { e.dtype = decode_datatype(r.sub()); e.has_dtype = true; handled = true; }
        else if (f == 3 && wt == 0) { e.fail_on_error = r.varint() != 0; handled = true; }
        break;
      case ExprKind::NormalizeNaNAndZero:
        if (f == 1 && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 2 && wt == 2) { e.dtype = decode_datatype(r.sub()); e.has_dtype = true; handled = true; }
        break;
      case ExprKind::UnaryMinus:
        if (f == 1 && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 2 && wt == 0) { e.fail_on_error = r.varint() != 0; handled = true; }
        break;
      case ExprKind::In:
        // children[0] = in_value, children[1..] = lists
        if ((f == 1 || f == 2) && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 3 && wt == 0) { e.negated = r.varint() != 0; handled = true; }
        break;
      case ExprKind::If:
        if (f >= 1 && f <= 3 && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        break;
      case ExprKind::GetStructField:
        if (f == 1 && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 2 && wt == 0) { e.bound_index = (int)(int32_t)r.varint(); handled = true; }
        break;
      case ExprKind::ListExtract:
        if (f >= 1 && f <= 3 && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 4 && wt == 0) { e.one_based = r.varint() != 0; handled = true; }
        else if (f == 5 && wt == 0) { e.fail_on_error = r.varint() != 0; handled = true; }
        break;
      case ExprKind::ScalarFunc:
        // ScalarFunc{func=1, args=2, return_type=3, fail_on_error=4} (expr.proto:466-471)
        if (f == 1 && wt == 2) { e.func = r.bytes(); handled = true; }
        else if (f == 2 && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        else if (f == 3 && wt == 2) { e.dtype = decode_datatype(r.sub()); e.has_dtype = true; handled = true; }
        else if (f == 4 && wt == 0) { e.fail_on_error = r.varint() != 0; handled = true; }
        break;
      case ExprKind::CaseWhen:
        // expr = 1 is never set by Spark (expr.proto:474-479); when = 2 and then = 3 are parallel lists, else_expr = 4.
        // proto3 writes fields in number order, so the children arrive as when* then* [else]
        if (f == 1 && wt == 2) throw CometError("CaseWhen with a base expression is not produced by Spark and is not supported");
        if (f == 2 && wt == 2) { e.children.push_back(decode_expr(r.sub())); e.n_when++; handled = true; }
        else if ((f == 3 || f == 4) && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        break;
      default:
        // BinaryExpr{left=1,right=2} and UnaryExpr{child=1}
        if ((f == 1 || f == 2) && wt == 2) { e.children.push_back(decode_expr(r.sub())); handled = true; }
        break;
    }
    if (!handled) r.skip(wt);
  }
}

// the contexts of the plan being decoded: their sql_text_idx is resolved against the ROOT operator's pool once the root is complete
thread_local std::vector<std::shared_ptr<QueryContext>>* g_decoded_contexts = nullptr;
thread_local std::vector<ExprP>* g_decoded_subqueries = nullptr;

std::shared_ptr<QueryContext> decode_query_context(Reader r) {
  auto c = std::make_shared<QueryContext>();
  while (!r.done()) {
    int wt, f = r.tag(wt);
    if (f == 1 && wt == 2) c->sql_text = r.bytes();
    else if (f == 2 && wt == 0) c->start_index = (int32_t)r.varint();
    else if (f == 3 && wt == 0) c->stop_index = (int32_t)r.varint();
    else if (f == 4 && wt == 2) { c->object_type = r.bytes(); c->has_object_type = true; }
    else if (f == 5 && wt == 2) { c->object_name = r.bytes(); c->has_object_name = true; }
    else if (f == 6 && wt == 0) c->line = (int32_t)r.varint();
    else if (f == 7 && wt == 0) c->start_position = (int32_t)r.varint();
    else if (f == 8 && wt == 0) c->sql_text_idx = (int32_t)r.varint();
    else r.skip(wt);
  }
  if (g_decoded_contexts) g_decoded_contexts->push_back(c);
  return c;
}

ExprP decode_expr(Reader r) {
  auto e = std::make_shared<Expr>();
  while (!r.done()) {
    int wt, f = r.tag(wt);
    if (f == 91 && wt == 0) { e->expr_id = r.varint(); e->has_expr_id = true; continue; }
    if (f == 90 && wt == 2) { e->qctx = decode_query_context(r.sub()); continue; }
    if (f == 90) { r.skip(wt); continue; }
    if (wt != 2) { r.skip(wt); continue; }
    e->proto_tag = f;
    switch (f) {
      case 2: e->kind = ExprKind::Literal; decode_literal(r.sub(), *e); break;
      case 50: {      // Subquery{id = 1, datatype = 2} (expr.proto:513-516)
        e->kind = ExprKind::Subquery;
        Reader b = r.sub();
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 0) e->lit_i64 = (int64_t)b.varint();
          else if (f2 == 2 && wt2 == 2) { e->dtype = decode_datatype(b.sub()); e->has_dtype = true; }
          else b.skip(wt2);
        }
        if (g_decoded_subqueries) g_decoded_subqueries->push_back(e);
        break;
      }
      case 3: case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 11: case 12: case 13: case 14:
      case 15: case 16: case 17: case 18: case 22: case 23: case 24: case 47: case 65: case 56: case 25: case 26: case 30: case 31: case 32: case 33: case 34: case 35: case 36: case 37: case 42: case 43: case 59: case 38: case 39: case 40: case 41:
      case 44: case 45: case 51: case 54:
        e->kind = (ExprKind)f;
        if (e->kind == ExprKind::Bound || e->kind == ExprKind::GetStructField) e->bound_index = 0;  // proto3 omits zero-valued scalars
        decode_expr_body(r.sub(), *e);
        break;
      default:
        e->kind = ExprKind::Unsupported;
        r.skip(wt);
    }
  }
  return e;
}

AggExpr decode_agg_expr(Reader r) {
  AggExpr a;
  while (!r.done()) {
    int wt, f = r.tag(wt);
    if (f == 89 && wt == 2) { a.filter = decode_expr(r.sub()); continue; }
    if (f == 91 && wt == 0) { a.expr_id = r.varint(); a.has_expr_id = true; continue; }
    if (f == 90 && wt == 2) { a.qctx = decode_query_context(r.sub()); continue; }      // AggExpr.query_context (expr.proto:171-175)
    if (f == 90 || wt != 2) { r.skip(wt); continue; }
    a.proto_tag = f;
    Reader b = r.sub();
    if (f >= 2 && f <= 8) a.kind = (AggKind)f;
    else { a.kind = AggKind::Unsupported; continue; }
    while (!b.done()) {
      int wt2, f2 = b.tag(wt2);
      if (f2 == 1 && wt2 == 2) a.children.push_back(decode_expr(b.sub()));
      else if (f2 == 2 && wt2 == 2) a.dtype = decode_datatype(b.sub());
      else if (f2 == 3 && wt2 == 2 && a.kind == AggKind::Avg) a.sum_dtype = decode_datatype(b.sub());
      else if (f2 == 3 && wt2 == 0 && a.kind == AggKind::Sum) a.eval_mode = (EvalMode)b.varint();
      else if (f2 == 4 && wt2 == 0 && a.kind == AggKind::Avg) a.eval_mode = (EvalMode)b.varint();
      else if (f2 == 3 && wt2 == 0 && (a.kind == AggKind::First || a.kind == AggKind::Last)) a.ignore_nulls = b.varint() != 0;   // expr.proto:210-220
      else b.skip(wt2);
    }
  }
  return a;
}

StructField decode_struct_field(Reader r) {
  StructField s;
  s.nullable = false;
  while (!r.done()) {
    int wt, f = r.tag(wt);
    if (f == 1 && wt == 2) s.name = r.bytes();
    else if (f == 2 && wt == 2) s.dtype = decode_datatype(r.sub());
    else if (f == 3 && wt == 0) s.nullable = r.varint() != 0;
    else if (f == 4 && wt == 2) {   // map<string,string> metadata: only "PARQUET:field_id" is consumed (schema_adapter.rs:67-73)
      Reader e = r.sub();
      std::string k, v;
      while (!e.done()) {
        int wt2, f2 = e.tag(wt2);
        if (f2 == 1 && wt2 == 2) k = e.bytes();
        else if (f2 == 2 && wt2 == 2) v = e.bytes();
        else e.skip(wt2);
      }
      if (k == "PARQUET:field_id") {
        char* endp = nullptr;
        long id = strtol(v.c_str(), &endp, 10);
        if (endp && *endp == 0 && !v.empty()) s.field_id = (int)id;   // parse::<i32>().ok(): unparsable ids are ignored
      }
    }
    else r.skip(wt);
  }
  return s;
}

void decode_native_scan(Reader r, Operator& op) {
  while (!r.done()) {
    int wt, f = r.tag(wt);
    if (f == 1 && wt == 2) {  // NativeScanCommon
      Reader c = r.sub();
      while (!c.done()) {
        int wt2, f2 = c.tag(wt2);
        if (f2 == 1 && wt2 == 2) op.required_schema.push_back(decode_struct_field(c.sub()));
        else if (f2 == 2 && wt2 == 2) op.data_schema.push_back(decode_struct_field(c.sub()));
        else if (f2 == 3 && wt2 == 2) op.partition_schema.push_back(decode_struct_field(c.sub()));
        else if (f2 == 4 && wt2 == 2) op.data_filters.push_back(decode_expr(c.sub()));
        else if (f2 == 5) c.repeated_varint(wt2, [&](uint64_t v) { op.projection_vector.push_back((int64_t)v); });
        else if (f2 == 6 && wt2 == 2) op.session_timezone = c.bytes();
        else if (f2 == 7 && wt2 == 2) op.default_values.push_back(decode_expr(c.sub()));
        else if (f2 == 8) c.repeated_varint(wt2, [&](uint64_t v) { op.default_values_indexes.push_back((int64_t)v); });
        else if (f2 == 11 && wt2 == 0) op.encryption_enabled = c.varint() != 0;
        else if (f2 == 15 && wt2 == 0) op.use_field_id = c.varint() != 0;
        else if (f2 == 16 && wt2 == 0) op.ignore_missing_field_id = c.varint() != 0;
        else if (f2 == 17 && wt2 == 0) op.allow_type_promotion = c.varint() != 0;
        else if (f2 == 18 && wt2 == 0) op.allow_timestamp_ltz_to_ntz = c.varint() != 0;
        else if (f2 == 9 && wt2 == 0) op.case_sensitive = c.varint() != 0;
        else if (f2 == 12 && wt2 == 2) op.scan_source = c.bytes();
        else if (f2 == 13 && wt2 == 2) op.scan_fields.push_back(decode_datatype(c.sub()));
        else c.skip(wt2);
      }
    } else if (f == 2 && wt == 2) {  // SparkFilePartition
      Reader fp = r.sub();
      while (!fp.done()) {
        int wt2, f2 = fp.tag(wt2);
        if (f2 == 1 && wt2 == 2) {
          Reader pf = fp.sub();
          PartitionedFile file;
          while (!pf.done()) {
            int wt3, f3 = pf.tag(wt3);
            if (f3 == 1 && wt3 == 2) file.file_path = pf.bytes();
            else if (f3 == 2 && wt3 == 0) file.start = (int64_t)pf.varint();
            else if (f3 == 3 && wt3 == 0) file.length = (int64_t)pf.varint();
            else if (f3 == 4 && wt3 == 0) file.file_size = (int64_t)pf.varint();
            else if (f3 == 5 && wt3 == 2) file.partition_values.push_back(decode_expr(pf.sub()));
            else pf.skip(wt3);
          }
          op.files.push_back(std::move(file));
        } else {
          fp.skip(wt2);
        }
      }
    } else {
      r.skip(wt);
    }
  }
}

// Expr { sort_order = 19 : SortOrder { child = 1, direction = 2, null_ordering = 3 } } (expr.proto:385-389)
Operator::SortKey decode_sort_order_expr(Reader e, const char* what) {
  Operator::SortKey k;
  while (!e.done()) {
    int wt3, f3 = e.tag(wt3);
    if (f3 == 19 && wt3 == 2) {
      Reader so = e.sub();
      while (!so.done()) {
        int wt4, f4 = so.tag(wt4);
        if (f4 == 1 && wt4 == 2) k.child = decode_expr(so.sub());
        else if (f4 == 2 && wt4 == 0) k.descending = so.varint() == 1;
        else if (f4 == 3 && wt4 == 0) k.nulls_last = so.varint() == 1;
        else so.skip(wt4);
      }
    } else {
      e.skip(wt3);
    }
  }
  if (!k.child) throw CometError(std::string(what) + ": sort_orders entry is not a SortOrder expression");
  return k;
}

OperatorP decode_operator_r(Reader r) {
  auto op = std::make_shared<Operator>();
  while (!r.done()) {
    int wt, f = r.tag(wt);
    if (f == 1 && wt == 2) { op->children.push_back(decode_operator_r(r.sub())); continue; }
    if (f == 2 && wt == 0) { op->plan_id = (uint32_t)r.varint(); continue; }
    if (f == 3 && wt == 2) { op->sql_text_pool.push_back(r.bytes()); continue; }      // operator.proto:39-47
    if (f < 100 || wt != 2) { r.skip(wt); continue; }
    op->proto_tag = f;
    Reader b = r.sub();
    switch (f) {
      case 100:
        op->kind = OpKind::Scan;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) op->scan_fields.push_back(decode_datatype(b.sub()));
          else if (f2 == 2 && wt2 == 2) op->scan_source = b.bytes();
          else b.skip(wt2);
        }
        break;
      case 116:   // ShuffleScan{fields = 1, source = 2}
        op->kind = OpKind::Scan;
        op->shuffle_scan = true;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) op->scan_fields.push_back(decode_datatype(b.sub()));
          else if (f2 == 2 && wt2 == 2) op->scan_source = b.bytes();
          else b.skip(wt2);
        }
        break;
      case 117: {
        // BroadcastNestedLoopJoin{join_type=1, build_side=2, condition=3} (operator.proto:773-777; planner.rs:1386-1430 → NestedLoopJoinExec):
        // executed as a hash join on a CONSTANT key — every build row sits in one chain that each probe row walks with the condition
        op->kind = OpKind::HashJoin;
        op->bnlj = true;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 0) op->join_type = (JoinType)b.varint();
          else if (f2 == 2 && wt2 == 0) op->build_side = (BuildSide)b.varint();
          else if (f2 == 3 && wt2 == 2) op->join_condition = decode_expr(b.sub());
          else b.skip(wt2);
        }
        for (int side = 0; side < 2; side++) {
          auto k = std::make_shared<Expr>();
          k->kind = ExprKind::Literal;
          k->proto_tag = 2;
          k->dtype = DType::of(TypeId::Int32);
          k->has_dtype = true;
          k->lit_case = 3;
          k->lit_i64 = 0;
          (side == 0 ? op->left_keys : op->right_keys).push_back(k);
        }
        break;
      }
      case 110: {
        // Window{window_expr=1 (WindowExpr{built_in_window_function=1, agg_func=2, spec=3, ignore_nulls=4, result_type=5}), order_by_list=2,
        //        partition_by_list=3, child=4}
        op->kind = OpKind::Window;
        OperatorP own_child;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) {
            Reader w = b.sub();
            Operator::WindowFn fn;
            while (!w.done()) {
              int wt3, f3 = w.tag(wt3);
              if (f3 == 1 && wt3 == 2) {
                ExprP e = decode_expr(w.sub());
                if (e->kind != ExprKind::ScalarFunc) throw CometError(std::string(expr_name(e->proto_tag)) + " not supported for window function");
                fn.func = e->func;
                fn.args = e->children;
              } else if (f3 == 2 && wt3 == 2) { fn.is_agg = true; fn.agg = decode_agg_expr(w.sub()); }
              else if (f3 == 3 && wt3 == 2) {
                // WindowSpecDefinition{partitionSpec=1, orderSpec=2, frameSpecification=3 WindowFrame{frame_type=1, lower_bound=2, upper_bound=3}}
                Reader sp = w.sub();
                while (!sp.done()) {
                  int wt4, f4 = sp.tag(wt4);
                  if (f4 != 3 || wt4 != 2) { sp.skip(wt4); continue; }
                  Reader fr = sp.sub();
                  while (!fr.done()) {
                    int wt5, f5 = fr.tag(wt5);
                    if (f5 == 1 && wt5 == 0) fn.frame_rows = fr.varint() == 0;
                    else if ((f5 == 2 || f5 == 3) && wt5 == 2) {
                      Reader bd = fr.sub();
                      int kind = f5 == 2 ? 0 : 2;   // proto3 default when the oneof is empty
                      int64_t off = 0;
                      while (!bd.done()) {
                        int wt6, f6 = bd.tag(wt6);
                        kind = f6 == 1 ? 0 : f6 == 2 ? 1 : 2;
                        if (f6 == 2 && wt6 == 2) {
                          // Preceding / Following { int64 offset = 1; Literal range_offset = 2 } (operator.proto:831-845)
                          Reader pf = bd.sub();
                          while (!pf.done()) {
                            int wt7, f7 = pf.tag(wt7);
                            if (f7 == 1 && wt7 == 0) off = (int64_t)pf.varint();
                            else if (f7 == 2 && wt7 == 2) {
                              fn.frame_range_literal = true;
                              auto lit = std::make_shared<Expr>();
                              lit->kind = ExprKind::Literal;
                              decode_literal(pf.sub(), *lit);
                              (f5 == 2 ? fn.frame_lower_range : fn.frame_upper_range) = lit;
                            } else pf.skip(wt7);
                          }
                        } else bd.skip(wt6);
                      }
                      (f5 == 2 ? fn.frame_lower : fn.frame_upper) = kind;
                      (f5 == 2 ? fn.frame_lower_off : fn.frame_upper_off) = off;
                    } else fr.skip(wt5);
                  }
                }
              }
              else if (f3 == 4 && wt3 == 0) fn.ignore_nulls = w.varint() != 0;
              else if (f3 == 5 && wt3 == 2) { fn.result_type = decode_datatype(w.sub()); fn.has_result_type = true; }
              else w.skip(wt3);
            }
            op->window_fns.push_back(fn);
          } else if (f2 == 2 && wt2 == 2) op->window_order.push_back(decode_sort_order_expr(b.sub(), "Window"));
          else if (f2 == 3 && wt2 == 2) op->window_partition.push_back(decode_expr(b.sub()));
          else if (f2 == 4 && wt2 == 2) own_child = decode_operator_r(b.sub());
          else b.skip(wt2);
        }
        if (own_child) op->window_child = own_child;
        break;
      }
      case 114:
        op->kind = OpKind::Explode;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) op->explode_child = decode_expr(b.sub());
          else if (f2 == 2 && wt2 == 0) op->explode_outer = b.varint() != 0;
          else if (f2 == 3 && wt2 == 2) op->project_list.push_back(decode_expr(b.sub()));
          else if (f2 == 4 && wt2 == 0) op->explode_position = b.varint() != 0;
          else b.skip(wt2);
        }
        break;
      case 107: {
        op->kind = OpKind::Expand;
        std::vector<ExprP> all;
        int per = 0;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) all.push_back(decode_expr(b.sub()));
          else if (f2 == 3 && wt2 == 0) per = (int)(int32_t)b.varint();
          else b.skip(wt2);
        }
        if (per <= 0 || all.empty() || all.size() % (size_t)per != 0) throw CometError("Expand: project_list does not split into projections of num_expr_per_project expressions");
        for (size_t i = 0; i < all.size(); i += (size_t)per) op->expand_projections.emplace_back(all.begin() + (long)i, all.begin() + (long)(i + (size_t)per));
        break;
      }
      case 106:
        op->kind = OpKind::ShuffleWriter;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) {
            Reader pr = b.sub();
            while (!pr.done()) {
              int wt3, f3 = pr.tag(wt3);
              if (wt3 != 2) { pr.skip(wt3); continue; }
              Reader q = pr.sub();
              op->shuffle_partitioning = (Operator::Partitioning)f3;
              if (f3 == 2) op->shuffle_num_partitions = 1;
              while (!q.done()) {
                int wt4, f4 = q.tag(wt4);
                if (f3 == 1 && f4 == 1 && wt4 == 2) op->shuffle_hash_exprs.push_back(decode_expr(q.sub()));
                else if (f3 == 1 && f4 == 2 && wt4 == 0) op->shuffle_num_partitions = (int)(int32_t)q.varint();
                else if (f3 == 3 && f4 == 2 && wt4 == 0) op->shuffle_num_partitions = (int)(int32_t)q.varint();
                else if (f3 == 3 && f4 == 1 && wt4 == 2) op->shuffle_sort_orders.push_back(decode_sort_order_expr(q.sub(), "RangePartition"));
                else if (f3 == 3 && f4 == 4 && wt4 == 2) {
                  Reader br = q.sub();
                  std::vector<ExprP> row;
                  while (!br.done()) {
                    int wt5, f5 = br.tag(wt5);
                    if (f5 == 1 && wt5 == 2) row.push_back(decode_expr(br.sub()));
                    else br.skip(wt5);
                  }
                  op->shuffle_bounds.push_back(row);
                }
                else if (f3 == 4 && f4 == 1 && wt4 == 0) op->shuffle_num_partitions = (int)(int32_t)q.varint();
                else if (f3 == 4 && f4 == 2 && wt4 == 0) op->shuffle_max_hash_columns = (int)(int32_t)q.varint();
                else q.skip(wt4);
              }
            }
          } else if (f2 == 3 && wt2 == 2) op->shuffle_data_file = b.bytes();
          else if (f2 == 4 && wt2 == 2) op->shuffle_index_file = b.bytes();
          else if (f2 == 5 && wt2 == 0) op->shuffle_codec = (int)b.varint();
          else if (f2 == 6 && wt2 == 0) op->shuffle_compression_level = (int)(int32_t)b.varint();
          else b.skip(wt2);
        }
        break;
      case 101:
        op->kind = OpKind::Projection;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) op->project_list.push_back(decode_expr(b.sub()));
          else b.skip(wt2);
        }
        break;
      case 102:
        op->kind = OpKind::Filter;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) op->predicate = decode_expr(b.sub());
          else b.skip(wt2);
        }
        break;
      case 104:
        op->kind = OpKind::HashAgg;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) op->grouping_exprs.push_back(decode_expr(b.sub()));
          else if (f2 == 2 && wt2 == 2) op->agg_exprs.push_back(decode_agg_expr(b.sub()));
          else if (f2 == 5 && wt2 == 0) op->agg_mode = (AggMode)b.varint();
          else if (f2 == 6) b.repeated_varint(wt2, [&](uint64_t v) { op->expr_modes.push_back((int)v); });
          else if (f2 == 7 && wt2 == 0) op->initial_input_buffer_offset = (int)(int32_t)b.varint();
          else b.skip(wt2);
        }
        break;
      case 103:
        op->kind = OpKind::Sort;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) {
            // Expr { sort_order = 19 : SortOrder { child = 1, direction = 2, null_ordering = 3 } }
            Reader e = b.sub();
            Operator::SortKey k;
            while (!e.done()) {
              int wt3, f3 = e.tag(wt3);
              if (f3 == 19 && wt3 == 2) {
                Reader so = e.sub();
                while (!so.done()) {
                  int wt4, f4 = so.tag(wt4);
                  if (f4 == 1 && wt4 == 2) k.child = decode_expr(so.sub());
                  else if (f4 == 2 && wt4 == 0) k.descending = so.varint() == 1;
                  else if (f4 == 3 && wt4 == 0) k.nulls_last = so.varint() == 1;
                  else so.skip(wt4);
                }
              } else {
                e.skip(wt3);
              }
            }
            if (!k.child) throw CometError("Sort: sort_orders entry is not a SortOrder expression");
            op->sort_orders.push_back(k);
          } else if (f2 == 3 && wt2 == 0) op->fetch = (int)(int32_t)b.varint();
          else if (f2 == 4 && wt2 == 0) op->skip = (int)(int32_t)b.varint();
          else b.skip(wt2);
        }
        break;
      case 105:
        op->kind = OpKind::Limit;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 0) op->limit = (int)(int32_t)b.varint();
          else if (f2 == 2 && wt2 == 0) op->offset = (int)(int32_t)b.varint();
          else b.skip(wt2);
        }
        break;
      case 108:
        // SortMergeJoin{left_join_keys=1, right_join_keys=2, join_type=3, sort_options=4 (SortOrder exprs), condition=5}
        op->kind = OpKind::HashJoin;
        op->smj = true;
        op->build_side = BuildSide::Right;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) op->left_keys.push_back(decode_expr(b.sub()));
          else if (f2 == 2 && wt2 == 2) op->right_keys.push_back(decode_expr(b.sub()));
          else if (f2 == 3 && wt2 == 0) op->join_type = (JoinType)b.varint();
          else if (f2 == 4 && wt2 == 2) {
            Reader e = b.sub();
            std::pair<bool, bool> opt(false, false);
            while (!e.done()) {
              int wt3, f3 = e.tag(wt3);
              if (f3 == 19 && wt3 == 2) {
                Reader so = e.sub();
                while (!so.done()) {
                  int wt4, f4 = so.tag(wt4);
                  if (f4 == 2 && wt4 == 0) opt.first = so.varint() == 1;
                  else if (f4 == 3 && wt4 == 0) opt.second = so.varint() == 1;
                  else so.skip(wt4);
                }
              } else {
                e.skip(wt3);
              }
            }
            op->smj_sort_options.push_back(opt);
          } else if (f2 == 5 && wt2 == 2) op->join_condition = decode_expr(b.sub());
          else b.skip(wt2);
        }
        if (op->join_type == JoinType::RightOuter) op->build_side = BuildSide::Left;   // probe (= preserved) side streams
        break;
      case 109:
        op->kind = OpKind::HashJoin;
        while (!b.done()) {
          int wt2, f2 = b.tag(wt2);
          if (f2 == 1 && wt2 == 2) op->left_keys.push_back(decode_expr(b.sub()));
          else if (f2 == 2 && wt2 == 2) op->right_keys.push_back(decode_expr(b.sub()));
          else if (f2 == 3 && wt2 == 0) op->join_type = (JoinType)b.varint();
          else if (f2 == 4 && wt2 == 2) op->join_condition = decode_expr(b.sub());
          else if (f2 == 5 && wt2 == 0) op->build_side = (BuildSide)b.varint();
          else if (f2 == 6 && wt2 == 0) op->null_aware_anti = b.varint() != 0;
          else b.skip(wt2);
        }
        break;
      case 111:
        op->kind = OpKind::NativeScan;
        decode_native_scan(b, *op);
        break;
      default:
        op->kind = OpKind::Unsupported;
    }
  }
  if (op->kind == OpKind::Window && op->children.empty() && op->window_child) op->children.push_back(op->window_child);
  return op;
}

void put_varint(std::string& s, uint64_t v) {
  while (v >= 0x80) { s.push_back((char)(v | 0x80)); v >>= 7; }
  s.push_back((char)v);
}

}  // namespace

OperatorP decode_operator(const uint8_t* data, size_t len) {
  std::vector<std::shared_ptr<QueryContext>> contexts;
  struct Scope {
    std::vector<std::shared_ptr<QueryContext>>*& slot;
    std::vector<std::shared_ptr<QueryContext>>* prev;
    ~Scope() { slot = prev; }
  } scope{g_decoded_contexts, g_decoded_contexts};
  g_decoded_contexts = &contexts;
  std::vector<ExprP> subqueries;
  struct SubScope { std::vector<ExprP>* prev; ~SubScope() { g_decoded_subqueries = prev; } } sub_scope{g_decoded_subqueries};
  g_decoded_subqueries = &subqueries;
  OperatorP root = decode_operator_r(Reader(data, len));
  root->subqueries = std::move(subqueries);
  // QueryContext.sql_text_idx → the root's pool (a query text shared by many expressions travels once per plan, expr.proto:137-141).  An index
  // outside the pool keeps the context's own sql_text, as the reference does ("warn rather than fail: a degraded error message is better
  // than a failed query", planner.rs:329-343).
  for (auto& c : contexts)
    if (c->sql_text_idx >= 0 && (size_t)c->sql_text_idx < root->sql_text_pool.size()) c->sql_text = root->sql_text_pool[(size_t)c->sql_text_idx];
  return root;
}
ExprP decode_expr_bytes(const uint8_t* data, size_t len) { return decode_expr(Reader(data, len)); }
DType decode_datatype_bytes(const uint8_t* data, size_t len) { return decode_datatype(Reader(data, len)); }
i128 decode_decimal_be(const std::string& bytes) { return decode_be_twos_complement(bytes); }

std::vector<std::pair<std::string, std::string>> decode_config_map(const uint8_t* data, size_t len) {
  std::vector<std::pair<std::string, std::string>> out;
  Reader r(data, len);
  while (!r.done()) {
    int wt, f = r.tag(wt);
    if (f == 1 && wt == 2) {  // map entry {key=1, value=2}
      Reader e = r.sub();
      std::string k, v;
      while (!e.done()) {
        int wt2, f2 = e.tag(wt2);
        if (f2 == 1 && wt2 == 2) k = e.bytes();
        else if (f2 == 2 && wt2 == 2) v = e.bytes();
        else e.skip(wt2);
      }
      out.emplace_back(std::move(k), std::move(v));
    } else {
      r.skip(wt);
    }
  }
  return out;
}

std::string encode_metric_node(const MetricNode& n) {
  std::string s;
  for (auto& kv : n.metrics) {
    std::string e;
    e.push_back((char)((1 << 3) | 2));
    put_varint(e, kv.first.size());
    e += kv.first;
    e.push_back((char)((2 << 3) | 0));
    put_varint(e, (uint64_t)kv.second);
    s.push_back((char)((1 << 3) | 2));
    put_varint(s, e.size());
    s += e;
  }
  for (auto& c : n.children) {
    std::string e = encode_metric_node(c);
    s.push_back((char)((2 << 3) | 2));
    put_varint(s, e.size());
    s += e;
  }
  return s;
}

std::string DType::str() const {
  switch (id) {
    case TypeId::Bool: return "Boolean";
    case TypeId::Int8: return "Int8";
    case TypeId::Int16: return "Int16";
    case TypeId::Int32: return "Int32";
    case TypeId::Int64: return "Int64";
    case TypeId::Float: return "Float32";
    case TypeId::Double: return "Float64";
    case TypeId::String: return "Utf8";
    case TypeId::Bytes: return "Binary";
    case TypeId::Timestamp: return "Timestamp(us, UTC)";
    case TypeId::TimestampNtz: return "Timestamp(us)";
    case TypeId::Date: return "Date32";
    case TypeId::Null: return "Null";
    case TypeId::Decimal: return "Decimal128(" + std::to_string(precision) + ", " + std::to_string(scale) + ")";
    case TypeId::Struct: {
      std::string o = "Struct(";
      for (size_t i = 0; i < kids.size(); i++) o += (i ? ", " : "") + (i < kid_names.size() ? kid_names[i] : std::string("?")) + ": " + kids[i].str();
      return o + ")";
    }
    case TypeId::List: return "List(" + (kids.empty() ? std::string("?") : kids[0].str()) + ")";
    case TypeId::Map: return "Map(" + (kids.empty() || kids[0].kids.size() != 2 ? std::string("?") : kids[0].kids[0].str() + ", " + kids[0].kids[1].str()) + ")";
    default: return "Unsupported(" + std::to_string((int)id) + ")";
  }
}

const char* op_name(int t) {
  switch (t) {
    case 100: return "Scan"; case 101: return "Projection"; case 102: return "Filter"; case 103: return "Sort";
    case 104: return "HashAggregate"; case 105: return "Limit"; case 106: return "ShuffleWriter";
    case 107: return "Expand"; case 108: return "SortMergeJoin"; case 109: return "HashJoin";
    case 110: return "Window"; case 111: return "NativeScan"; case 112: return "IcebergScan";
    case 113: return "ParquetWriter"; case 114: return "Explode"; case 115: return "CsvScan";
    case 116: return "ShuffleScan"; case 117: return "BroadcastNestedLoopJoin"; case 118: return "Sample";
    case 200: return "ContribScan"; default: return "Unknown";
  }
}

const char* expr_name(int t) {
  switch (t) {
    case 2: return "Literal"; case 3: return "BoundReference"; case 4: return "Add"; case 5: return "Subtract";
    case 6: return "Multiply"; case 7: return "Divide"; case 8: return "Cast"; case 9: return "Eq"; case 22: return "Hour"; case 23: return "Minute"; case 24: return "Second";
    case 10: return "Neq"; case 11: return "Gt"; case 12: return "GtEq"; case 13: return "Lt"; case 14: return "LtEq";
    case 15: return "IsNull"; case 16: return "IsNotNull"; case 17: return "And"; case 18: return "Or";
    case 19: return "SortOrder"; case 25: return "CheckOverflow"; case 26: return "Like"; case 30: return "RLike"; case 31: return "ScalarFunc";
    case 32: return "EqNullSafe"; case 33: return "NeqNullSafe"; case 37: return "Remainder"; case 38: return "CaseWhen";
    case 39: return "In"; case 40: return "Not"; case 41: return "UnaryMinus"; case 44: return "If"; case 54: return "GetStructField"; case 56: return "ListExtract";
    case 45: return "NormalizeNaNAndZero"; case 47: return "TruncTimestamp"; case 50: return "Subquery"; case 65: return "UnixTimestamp"; default: return "Expr";
  }
}

}  // namespace comet
