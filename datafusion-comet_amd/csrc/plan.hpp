// Plan IR decoded from Comet's protobuf plan (native/proto/src/proto/{operator,expr,types,literal}.proto).
// The IR mirrors the proto messages this engine executes; anything else decodes to Kind::Unsupported
// and is rejected at planning time with the operator/expression name, like the reference's planner
// does for unknown arms (native/core/src/execution/planner.rs:446, :1211).
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace comet {

struct CometError : std::runtime_error {
  // kind: 0 = CometNativeException, 1 = CometQueryExecutionException (json payload)
  int kind;
  explicit CometError(const std::string& m, int k = 0) : std::runtime_error(m), kind(k) {}
};

typedef __int128 i128;
typedef unsigned __int128 u128;

// spark.spark_expression.DataType.DataTypeId (types.proto:43-66)
enum class TypeId : int {
  Bool = 0, Int8 = 1, Int16 = 2, Int32 = 3, Int64 = 4, Float = 5, Double = 6, String = 7, Bytes = 8,
  Timestamp = 9, Decimal = 10, TimestampNtz = 11, Date = 12, Null = 13, List = 14, Map = 15, Struct = 16,
  Time = 17, Unknown = 99
};

struct DType {
  TypeId id = TypeId::Unknown;
  int precision = 0, scale = 0;  // Decimal only
  // Nested types (types.proto StructInfo / ListInfo): a Struct's fields, a List's one element type ("element").  Nested columns are read by
  // the Parquet scan, passed through Filter / Projection (gathered by row index like Utf8 columns), taken apart by GetStructField and exported;
  // no expression computes on them.
  std::vector<DType> kids;
  std::vector<std::string> kid_names;
  std::vector<char> kid_nullable;
  // a struct's field made addressable as a column of its own (exec.cpp extend_struct_fields: the chain's input table gets one such column
  // per field behind its real columns; GetStructField(Bound(parent), kid) is then a Bound reference to it).  Not part of the type's identity
  int virt_parent = -1, virt_kid = -1;
  bool is_nested() const { return id == TypeId::Struct || id == TypeId::List || id == TypeId::Map; }
  // a List, or a Map — which is laid out as a list of (key, value) entry structs (Arrow's Map layout): kids[0] = Struct(key, value)
  bool is_listlike() const { return id == TypeId::List || id == TypeId::Map; }
  bool operator==(const DType& o) const {
    if (id != o.id || precision != o.precision || scale != o.scale || kids.size() != o.kids.size()) return false;
    for (size_t i = 0; i < kids.size(); i++)
      if (kids[i] != o.kids[i] || (id == TypeId::Struct && i < kid_names.size() && i < o.kid_names.size() && kid_names[i] != o.kid_names[i])) return false;
    return true;
  }
  bool operator!=(const DType& o) const { return !(*this == o); }
  bool is_decimal() const { return id == TypeId::Decimal; }
  bool is_integer() const { return id == TypeId::Int8 || id == TypeId::Int16 || id == TypeId::Int32 || id == TypeId::Int64; }
  bool is_float() const { return id == TypeId::Float || id == TypeId::Double; }
  std::string str() const;
  static DType of(TypeId t) { DType d; d.id = t; return d; }
  static DType decimal(int p, int s) { DType d; d.id = TypeId::Decimal; d.precision = p; d.scale = s; return d; }
};

enum class EvalMode : int { Legacy = 0, Try = 1, Ansi = 2 };

// Expr.expr_struct oneof tags (expr.proto:30-107) for the arms on the hot path.
enum class ExprKind : int {
  Literal = 2, Bound = 3, Add = 4, Subtract = 5, Multiply = 6, Divide = 7, Cast = 8,
  Hour = 22, Minute = 23, Second = 24,      // expr.proto:436-453: child = 1, timezone = 2 (kept in Expr::func)
  TruncTimestamp = 47, UnixTimestamp = 65,  // expr.proto:507-511 (children = [format, child]), 455-458; the zone in Expr::func
  Eq = 9, Neq = 10, Gt = 11, GtEq = 12, Lt = 13, LtEq = 14, IsNull = 15, IsNotNull = 16, And = 17, Or = 18,
  CheckOverflow = 25, Like = 26, RLike = 30, ScalarFunc = 31, EqNullSafe = 32, NeqNullSafe = 33, BitAnd = 34, BitOr = 35, BitXor = 36, Remainder = 37, CaseWhen = 38, In = 39, Not = 40,
  UnaryMinus = 41, ShiftRight = 42, ShiftLeft = 43, If = 44, IntegralDivide = 59, NormalizeNaNAndZero = 45, Unbound = 51,
  GetStructField = 54,              // expr.proto:528-531: child = 1, ordinal = 2 (kept in Expr::bound_index)
  Subquery = 50,                    // expr.proto:513-516: a scalar subquery's result — id = 1 (kept in Expr::lit_i64), datatype = 2; a Literal once the executor has asked for it
  ListExtract = 56,                 // expr.proto:533-539: children = [child, ordinal, (default value)], one_based, fail_on_error
  Unsupported = -1
};

struct Expr;
typedef std::shared_ptr<Expr> ExprP;

// QueryContext (expr.proto:109-141; Spark's SQLQueryContext): where in the SQL text an expression stands — attached to the errors it raises
struct QueryContext {
  std::string sql_text;
  int32_t start_index = 0, stop_index = 0, line = 0, start_position = 0;
  bool has_object_type = false, has_object_name = false;
  std::string object_type, object_name;
  int32_t sql_text_idx = -1;      // index into the root Operator.sql_text_pool (resolved into sql_text by decode_operator)
};

struct Expr {
  ExprKind kind = ExprKind::Unsupported;
  int proto_tag = 0;              // raw oneof tag (for error messages)
  std::vector<ExprP> children;    // operands in proto field order
  DType dtype;                    // declared type (Literal.datatype, Bound.datatype, Cast.datatype,
                                  // CheckOverflow.datatype, MathExpr.return_type)
  bool has_dtype = false;
  EvalMode eval_mode = EvalMode::Legacy;
  bool fail_on_error = false;     // CheckOverflow / UnaryMinus
  bool check_divide_overflow = false;   // IntegralDivide (MathExpr field 6)
  bool one_based = false;         // ListExtract: element_at counts from 1 (negative: from the end), GetArrayItem from 0
  bool is_spark4_plus = false;    // Cast (expr.proto:349-351): Spark 4's reading of leading whitespace before T-prefixed time-only strings
  bool negated = false;           // In
  std::string func;               // ScalarFunc.func
  int n_when = 0;                 // CaseWhen: children = when[0..n) ++ then[0..n) ++ [else]
  int bound_index = -1;           // Bound
  // Literal payload
  bool lit_null = false;
  bool lit_bool = false;
  int64_t lit_i64 = 0;            // byte/short/int/long/date/timestamp
  double lit_f64 = 0;             // float/double
  i128 lit_dec = 0;               // decimal unscaled value (literal.proto:38, BE two's complement)
  std::string lit_bytes;          // string/bytes
  int lit_case = 0;               // which Literal.value arm was present (1..11), 0 = none
  uint64_t expr_id = 0;
  bool has_expr_id = false;
  std::shared_ptr<QueryContext> qctx;      // Expr.query_context = 90 (the reference registers it under expr_id, planner.rs:302-316: both must be there)
};

// AggExpr.expr_struct oneof tags (expr.proto:143-176)
enum class AggKind : int { Count = 2, Sum = 3, Min = 4, Max = 5, Avg = 6, First = 7, Last = 8, Unsupported = -1 };

struct AggExpr {
  AggKind kind = AggKind::Unsupported;
  int proto_tag = 0;
  std::vector<ExprP> children;    // Count.children or the single child
  DType dtype;                    // Sum/Min/Max/Avg.datatype (result type)
  DType sum_dtype;                // Avg.sum_datatype
  EvalMode eval_mode = EvalMode::Legacy;
  ExprP filter;                   // AggExpr.filter = 89
  bool ignore_nulls = false;      // First / Last
  uint64_t expr_id = 0;
  bool has_expr_id = false;
  std::shared_ptr<QueryContext> qctx;      // AggExpr.query_context = 90: goes with the aggregate's DecimalSumOverflow (sum_decimal.rs wrap_error_with_context)
};

// Operator.op_struct oneof tags (operator.proto:32-79)
enum class OpKind : int {
  Scan = 100, Projection = 101, Filter = 102, Sort = 103, HashAgg = 104, Limit = 105, ShuffleWriter = 106, Expand = 107, HashJoin = 109, Window = 110,   // SortMergeJoin (108) decodes to HashJoin + smj
  NativeScan = 111, Explode = 114, Unsupported = -1
};

enum class AggMode : int { Partial = 0, Final = 1, PartialMerge = 2 };
enum class JoinType : int { Inner = 0, LeftOuter = 1, RightOuter = 2, FullOuter = 3, LeftSemi = 4, LeftAnti = 5 };
enum class BuildSide : int { Left = 0, Right = 1 };

struct StructField {  // SparkStructField (operator.proto:117-124)
  std::string name;
  DType dtype;
  bool nullable = true;
  int field_id = -1;   // metadata["PARQUET:field_id"] (CometParquetUtils.PARQUET_FIELD_ID_META_KEY), -1 = none
  // the Parquet scan's own use: a LEAF of a nested column it reads — nest 1: field `name` of the struct column `parent`; 2: the element of
  // the list column `parent`; 3: field `name` of the struct elements of the list column `parent` (0: a top-level column)
  int nest = 0;
  std::string parent;
  int parent_field_id = -1;
};

struct PartitionedFile {  // SparkPartitionedFile (operator.proto:103-109)
  std::string file_path;
  int64_t start = 0, length = 0, file_size = 0;
  std::vector<ExprP> partition_values;
};

struct Operator;
typedef std::shared_ptr<Operator> OperatorP;

struct Operator {
  OpKind kind = OpKind::Unsupported;
  int proto_tag = 0;
  uint32_t plan_id = 0;
  std::vector<std::string> sql_text_pool;      // (root only) the SQL texts QueryContext.sql_text_idx points into
  std::vector<ExprP> subqueries;               // (root only) every Subquery expression of the plan (resolved into literals at the first executePlan)
  bool reader_api = false;                      // NativeScan built by the parquet.Native record-batch reader, not decoded from a plan
  std::vector<OperatorP> children;
  // Scan
  std::vector<DType> scan_fields;
  std::string scan_source;
  // Projection
  std::vector<ExprP> project_list;
  // Filter
  ExprP predicate;
  // HashAggregate
  std::vector<ExprP> grouping_exprs;
  std::vector<AggExpr> agg_exprs;
  AggMode agg_mode = AggMode::Partial;
  std::vector<int> expr_modes;
  int initial_input_buffer_offset = 0;
  // HashJoin
  std::vector<ExprP> left_keys, right_keys;
  JoinType join_type = JoinType::Inner;
  ExprP join_condition;
  BuildSide build_side = BuildSide::Left;
  bool null_aware_anti = false;
  bool bnlj = false;                           // BroadcastNestedLoopJoin (117): no equi-keys, only the condition
  // SortMergeJoin (operator.proto:765-771): executed as a hash join whose output is then sorted by the join keys
  bool smj = false;
  std::vector<std::pair<bool, bool>> smj_sort_options;   // per key: (descending, nulls_last)
  // Limit
  int limit = -1, offset = 0;
  // Sort (operator.proto:641-645; SortOrder expr.proto:385-389)
  struct SortKey { ExprP child; bool descending = false; bool nulls_last = false; };
  std::vector<SortKey> sort_orders;
  int fetch = -1, skip = 0;
  // NativeScan
  std::vector<StructField> required_schema, data_schema, partition_schema;
  std::vector<ExprP> data_filters;
  std::vector<int64_t> projection_vector;
  std::vector<PartitionedFile> files;
  std::string session_timezone;
  bool case_sensitive = false;                 // NativeScanCommon.case_sensitive (proto3 default)
  std::vector<ExprP> default_values;            // NativeScanCommon.default_values (literals), parallel to default_values_indexes
  std::vector<int64_t> default_values_indexes;  // required_schema positions that carry a default value
  bool encryption_enabled = false;
  bool use_field_id = false, ignore_missing_field_id = false;
  bool allow_type_promotion = false, allow_timestamp_ltz_to_ntz = false;   // proto3 defaults (Spark 3.x behaviour)
  // Window (operator.proto:793-862): child columns ++ one column per window expression; input sorted by (partition, order) keys
  struct WindowFn {
    std::string func;               // built_in_window_function: ScalarFunc name (row_number, rank, dense_rank, percent_rank, cume_dist, ntile, lag, lead)
    std::vector<ExprP> args;
    bool is_agg = false;            // agg_func present (aggregate over a frame)
    AggExpr agg;                    // the aggregate (children, result type)
    bool frame_rows = true;         // WindowFrame.frame_type: ROWS (proto3 default) or RANGE
    int frame_lower = 0, frame_upper = 2;   // 0 = UNBOUNDED, 1 = offset (PRECEDING / FOLLOWING), 2 = CURRENT ROW
    // offset bounds: rows relative to the current row, negative = PRECEDING, positive = FOLLOWING (both bounds; planner.rs:3016-3030)
    int64_t frame_lower_off = 0, frame_upper_off = 0;
    bool frame_range_literal = false;       // a RANGE frame with a value offset (Preceding / Following.range_offset)
    ExprP frame_lower_range, frame_upper_range;   // that offset: a literal of the ORDER BY key's type, magnitude only (CometWindowExec.scala:588-632)
    DType result_type;
    bool has_result_type = false;
    bool ignore_nulls = false;
  };
  std::vector<WindowFn> window_fns;
  std::shared_ptr<Operator> window_child;   // Window.child (used when Operator.children is empty)
  std::vector<SortKey> window_order;
  std::vector<ExprP> window_partition;
  // Expand (operator.proto:738-741): project_list holds num_expr_per_project expressions per projection, back to back
  std::vector<std::vector<ExprP>> expand_projections;
  // Explode (operator.proto:743-752): the list to explode, explode_outer, posexplode; project_list = the columns carried alongside
  ExprP explode_child;
  bool explode_outer = false, explode_position = false;
  // ShuffleScan (operator.proto:134-138) decodes to Scan with this flag: its input is a stream of shuffle blocks
  bool shuffle_scan = false;
  // ShuffleWriter (operator.proto:688-707; Partitioning partitioning.proto:29-66)
  enum class Partitioning : int { Hash = 1, Single = 2, Range = 3, RoundRobin = 4 };
  Partitioning shuffle_partitioning = Partitioning::Single;
  std::vector<ExprP> shuffle_hash_exprs;
  std::vector<SortKey> shuffle_sort_orders;             // RangePartition.sort_orders
  std::vector<std::vector<ExprP>> shuffle_bounds;       // RangePartition.boundary_rows: one literal per sort order, ascending in that order
  int shuffle_num_partitions = 1;
  int shuffle_max_hash_columns = 0;
  std::string shuffle_data_file, shuffle_index_file;
  int shuffle_codec = 0;              // CompressionCodec: 0 None, 1 Zstd, 2 Lz4, 3 Snappy
  int shuffle_compression_level = 1;
};

// proto.cpp
OperatorP decode_operator(const uint8_t* data, size_t len);
ExprP decode_expr_bytes(const uint8_t* data, size_t len);   // one serialized spark_expression.Expr
DType decode_datatype_bytes(const uint8_t* data, size_t len);
i128 decode_decimal_be(const std::string& bytes);   // BigInteger.toByteArray → the unscaled value (decimal literals, scalar subqueries)   // one serialized spark_expression.DataType (types.proto:27-41)
std::vector<std::pair<std::string, std::string>> decode_config_map(const uint8_t* data, size_t len);
// NativeMetricNode encoder (metric.proto:26-29)
struct MetricNode {
  std::vector<std::pair<std::string, int64_t>> metrics;
  std::vector<MetricNode> children;
};
std::string encode_metric_node(const MetricNode& n);

const char* op_name(int proto_tag);
const char* expr_name(int proto_tag);

}  // namespace comet
