// Planner + code generator: proto plan IR → fused pipeline description + HIP source whose only
// plan-specific part is the per-row expression functor; the kernels themselves are the hand-written
// templates of device/comet_device.hpp.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "plan.hpp"

namespace comet {

enum class SinkKind { Output, AggNoGroup, AggGrouped };

struct OutCol {
  DType type;
  bool nullable = true;
  // AggGrouped string keys come back packed (≤7 bytes + length) in a u64; host expands to Utf8.
  bool packed_string = false;
  // Utf8/Binary column passed through unchanged: the emit kernel writes the SOURCE ROW INDEX (u32) of every output row and
  // the executor gathers offsets + bytes afterwards (any string length).  gather_src = source column; for joins it indexes
  // left ++ right.  -1 = not a gathered column.
  int gather_src = -1;
  // Utf8 RESULT of any length that is a slice of source column view_src plus padding (substring, trim, rpad / lpad, read-side padding):
  // the kernel writes a comet::strview (16 bytes) per output row, the executor sizes and writes offsets + bytes afterwards
  int view_src = -1;
  // Utf8 RESULT that is a VALUE written out (Cast … AS STRING of an integer, boolean, decimal, date or timestamp): the kernel stores the value
  // as an i128 per output row, the executor sizes and writes the column (strfmt kernels).  0 = not such a column
  enum FmtKind { FmtNone = 0, FmtInt = 1, FmtBool = 2, FmtDecimal = 3, FmtDecimalJava = 4, FmtDate = 5, FmtTimestamp = 6, FmtFloat64 = 7, FmtFloat32 = 8 };
  int fmt_kind = FmtNone;
  long long fmt_arg = 0;            // FmtDecimal*: the scale; FmtTimestamp: the zone's offset from UTC in seconds
  // Utf8 RESULT that is the concatenation of Utf8 source columns and literals (concat): the kernel writes the SOURCE ROW INDEX (u32), the
  // executor sizes and writes the column.  concat_cols[k] = source column of part k, or −1: the literal concat_lits[k].  Empty = not such a column
  std::vector<int> concat_cols;
  std::vector<std::string> concat_lits;
  int case_mode = 0;                // a view whose bytes are case-mapped on the way out: 1 lower, 2 upper (device/case_map.hpp)
  std::string pad_pattern;          // the pad string (≤ 64 bytes, ≤ 32 characters)
  bool pad_left = false;
};

// A column the executor computes over the chain's SOURCE table before the fused kernel runs, addressed by the chain like a source column
// (in_types holds it behind the source's own columns): today split(<Utf8 column>, <pattern literal>, <limit literal>) → list<string>,
// which the chain passes through by row index like any nested column.
struct DerivedCol {
  int kind = 0;                     // 1: split, 2: regexp_extract_all (two columns each: the list, then its elements), 3: a string function → Utf8 (one column)
  int op = 0;                       // kind 3: device/strfn.hpp's operation (SF_REVERSE …)
  std::string arg_a, arg_b;         // kind 3: the literal arguments' bytes
  long long arg_k = 0;              // kind 3: the integer argument
  int columns() const { return kind == 3 ? 1 : 2; }
  int src = -1;                     // the source column
  std::vector<uint32_t> prog;       // the pattern as a group-0 program of device/regex_vm.hpp
  std::vector<uint32_t> prog2;      // regexp_extract_all: the wanted group's program (empty: group 0)
  int limit = -1;
  DType type;
};

// How the host must treat each group-key word / accumulator word of a grouped aggregate.
struct PipelineDesc {
  SinkKind sink = SinkKind::Output;
  bool has_filter = false;
  int R = 4;                       // rows per thread per tile
  int NW = 0;                      // accumulator words (aggregates)
  int NK = 0;                      // key words (grouped)
  int NPW = 0;                     // private limb-form words per group (grouped)
  int lds_cap = 0;
  std::vector<DType> in_types;     // Scan fields (behind them: the derived columns, in order)
  std::vector<DerivedCol> derived; // columns the executor computes over the source before the launch (split)
  std::vector<bool> in_used;       // columns the kernels actually read
  std::vector<OutCol> out_cols;    // output schema in order
  // raise sites of this pipeline's kernels whose expression carries a QueryContext: the executor attaches it to the error (the site ids in the
  // generated text do not depend on the SQL text, so plans that differ only in it share their code objects)
  std::vector<std::pair<uint32_t, std::shared_ptr<QueryContext>>> site_contexts;
  // the QueryContext of the pipeline's ANSI decimal sum ([0]) / average ([1]) — the DecimalSumOverflow flags do not say WHICH aggregate
  // overflowed, so a context is attached only when all of a kind that carry one carry the same; agg_ctx_mixed[k]: they differ
  std::shared_ptr<QueryContext> agg_ctx[2];
  bool agg_ctx_mixed[2] = {false, false}, agg_ctx_seen[2] = {false, false};
  std::string source;              // full HIP translation unit
  std::vector<std::string> kernels;  // extern "C" kernel names present in `source`
  // static row bound under which decimal sums cannot overflow (Appendix C.1 rule); 0 = no limit
  long long max_rows_exact = 0;
  int sort_key_bytes = 0;           // generate_sort_keys: width of one row's key (the fixed part when Utf8 columns are sort keys)
  // generate_sort_keys: Utf8 COLUMNS used as sort keys, in key order.  Their key bytes are the value zero-padded to the longest value of
  // the column plus a 4-byte length; the executor measures that length L_s and passes L_s + 4 in prm.iarg[1 + s] (at most 6 such keys)
  std::vector<int> sort_str_cols;
  bool join_build_only = false;     // LeftSemi/LeftAnti built on the left: only the tail pass produces rows
  bool join_outer_build = false;    // hash join that must also emit the build rows no probe row matched
  bool join_dedup_build = false;    // semi / anti join that keeps probe rows and has no residual condition: only a key's existence matters
  std::string explain;             // human-readable fused plan
  std::vector<std::string> op_names;  // operator names root→leaf (metrics tree / tracing label)
  // grouped aggregates: source columns that serve DIRECTLY as Utf8 group keys (the executor measures their longest value: above 15
  // bytes the key travels as a representative row index instead of packed bytes, see dict_id_col)
  std::vector<int> str_key_cols;
  // exact Float64 sums (comet_device.hpp "Exact Float64 sums"): per sum f its first canonical accumulator word (3 words, 192-bit fixed
  // point) and the aux words (u64 index after the 16-byte error header) that collect the exponent range seen; the scales travel packed
  // 16 bits each in prm.iarg[kFixScaleArg]
  struct FixSum { int word = 0; int aux_hi = 0, aux_lo = 0; };
  std::vector<FixSum> fix_sums;
  // grouped aggregate whose input rows are Partial states (Final / PartialMerge): about one input row per group and partition, so the
  // executor sizes the group table from the row count instead of growing it from the small default
  bool merges_states = false;
};
constexpr int kFixScaleArg = 6;      // scales of sums 0-3, 16 bits each
constexpr int kFixScaleArg2 = 4;     // … of sums 4-7 (round 6: eight exact Float64 sums / averages per aggregate)
constexpr int kFixMaxSums = 8;
constexpr int kFixW = 158;            // must equal comet::kFixW
constexpr int kFixDefaultScale = -94; // window [2^-94, 2^64): doubles from 2^-42 to 2^64 with every mantissa bit, without a re-run

// `in_has_validity[i]` tells whether input column i arrives with a validity bitmap in this batch
// chunk; kernels are specialised on it.  Throws CometError for unsupported plans.
// dict_id_col (optional, one entry per source column): index of an appended Int64 input column that holds, for every row, the index
// of the representative row of that row's string (strdict_kernels.hip), or -1.  A Utf8 group key that is such a column is grouped
// on the index and emitted as a gathered column (OutCol::gather_src) — strings of any length.
PipelineDesc generate_pipeline(const Operator& root, const std::vector<bool>& in_has_validity,
                               const std::vector<DType>* source_types = nullptr, const std::vector<int>* str_fixed_len = nullptr,
                               const std::vector<int>* dict_id_col = nullptr);

// Hash join of two materialised tables (left/right = the join's children in plan order).
// Sort: the kernel that writes every row's order-preserving key bytes (byte planes)
PipelineDesc generate_sort_keys(const Operator& sort, const std::vector<DType>& types, const std::vector<bool>& has_validity);
// Probe-side fusion (DataFusion's HashJoinExec STREAMS its probe side, planner.rs:2192-2266; materialising the probe child first costs a
// write and a read of every surviving row): when the probe child is a chain of Filters / Projections over a source, the join kernel
// reads the SOURCE table — the child's columns are expressions over the source columns, its Filters run inside the probe kernel.
struct JoinFusion {
  std::vector<DType> src_types;   // the chain's source table = the physical probe side
  std::vector<bool> src_valid;
  std::vector<ExprP> cols;        // probe child column c as an expression over the source columns
  std::vector<ExprP> preds;       // the chain's Filter conjuncts over the source columns
};
// left / right types and validity describe the join's children; with `probe_fusion` the probe side's entries are ignored
PipelineDesc generate_join(const Operator& join, const std::vector<DType>& left_types, const std::vector<DType>& right_types,
                           const std::vector<bool>& left_has_validity, const std::vector<bool>& right_has_validity,
                           const JoinFusion* probe_fusion = nullptr, const JoinFusion* build_fusion = nullptr);
// the (cols, preds) of a Filter / Projection chain `top … down to (excluding) source`, over the source columns
void fold_chain(const Operator& top, const Operator& source, const std::vector<DType>& source_types, std::vector<ExprP>& cols, std::vector<ExprP>& preds);

// index of out[] slots used by the generated kernels (must match exec.cpp)
constexpr int kOutPartials = 0;      // AggNoGroup: partials;  Output: mask words
constexpr int kOutCounts = 1;        // Output: tile counts / offsets
constexpr int kOutErr = 2;           // u32[4] error flags (ANSI overflow etc.)
constexpr int kErrBytes = 512;       // error/aux block (kparams.h COMET_ERR_BYTES): flags, group counter, aux words, error detail, scratch
constexpr int kErrAuxWords = 22;     // (COMET_ERR_AUX_WORDS)
constexpr int kErrDetailWord = 24;   // (COMET_ERR_DETAIL_WORD)
constexpr int kErrDetailStrBytes = 224;

// A raise site whose Spark error names the offending value (common/src/error.rs:318-380 params_as_json): what is static about it.  Sites are
// registered process-wide under an id derived from their content, so the generated text — and with it the code-object cache key — does not
// depend on the order plans arrive in.
struct ErrSite {
  enum Value : int { Unscaled128 = 0, Int64 = 1, F64 = 2, F32 = 3, Str = 4, DecimalBD = 5, F64Display = 6, Int64Plain = 7, NoValue = 8, F64Micros = 9, FunctionName = 10, IndexAndSize = 11 };
  std::string error_type;    // "NumericValueOutOfRange", "CastOverFlow", "CastInvalidValue", "InvalidInputInCastToDatetime"
  std::string error_class;
  std::string from_type, to_type;      // Spark SQL type names ("BIGINT", "DECIMAL(10,2)", "STRING" …)
  int precision = 0, scale = 0;        // NumericValueOutOfRange; DecimalBD: the SOURCE decimal's
  int value = Unscaled128;
  std::string suffix;                  // Int64: Spark's literal suffix ("L", "S", "")
};
uint32_t register_err_site(const ErrSite& s, int ordinal = 0);      // ordinal: the n-th site of this very content in one pipeline
bool lookup_err_site(uint32_t id, ErrSite& out);
// the error JSON of a site and the detail the device left (lo / hi: the value's bits or a string's length; str: its first bytes)
std::string err_site_json(const ErrSite& s, uint64_t lo, uint64_t hi, const uint8_t* str, size_t str_avail, const QueryContext* ctx = nullptr);
// DecimalSumOverflow of an ANSI decimal sum (kind 0) / average (1), with the aggregate's context when the plan carried one
std::string decimal_sum_overflow_json(int kind, const QueryContext* ctx);
constexpr int kOutFirstCol = 4;      // out[4+2j] = values of col j, out[5+2j] = validity bytes of col j

}  // namespace comet
