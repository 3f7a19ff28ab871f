/* comet_amd.h — C ABI of libcomet.so, the MI355X-native drop-in for Comet's native library on the
 * hot path  Scan → Filter/Project → HashAggregate (→ HashJoin probe).
 *
 * Each entry point names the reference interface it replaces.  The JNI symbols that Spark's
 * `Native.scala` binds (Java_org_apache_comet_Native_*) are thin shims over these functions
 * (datafusion-comet_amd/csrc/jni_shim.cpp); tests and bench.py drive the same functions through
 * ctypes + the Arrow C Data / C Stream / C Device interfaces, the JVM-free path that mirrors the
 * reference's TEST_EXEC_CONTEXT_ID planner tests (native/core/src/execution/planner.rs:245,4637-4936).
 *
 * Conventions: plain pointers and sizes only; no C++ exception ever crosses this boundary.  A failed
 * call returns the documented error value and leaves a message retrievable with comet_last_error().
 */
#ifndef COMET_AMD_H
#define COMET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libcomet.so is built with -fvisibility=hidden and an export list generated from this header (csrc/gen_exports.py): what is declared between this push and
 * the pop at the end, plus the Java_org_apache_comet_* names, is the library's whole dynamic symbol table (tests/test_boundary_cpu.py diffs `nm -D` against it). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

struct ArrowArray;
struct ArrowSchema;
struct ArrowDeviceArray;

/* how input i is handed over */
#define COMET_INPUT_HOST_STREAM 0   /* struct ArrowArrayStream*       — JVM path (CometNativeArrowSource.scala:67) */
#define COMET_INPUT_DEVICE_STREAM 1 /* struct ArrowDeviceArrayStream* — ARROW_DEVICE_ROCM buffers already in HBM */
#define COMET_INPUT_SHUFFLE_BLOCKS 2 /* struct CometShuffleBlockStream*  — input of a ShuffleScan leaf (operator.proto:134-138) */

/* The C face of org.apache.comet.CometShuffleBlockIterator (what the reference's ShuffleScanExec pulls through JNI,
 * native/core/src/execution/operators/shuffle_scan.rs:139-171): next_block plays hasNext() + getBuffer() — it returns the
 * length of the next block and points *data at its bytes, which start at the 4-byte codec tag (the 8-byte length and
 * 8-byte field-count words of the on-disk block are already consumed) and stay valid until the next call; -1 at the end,
 * -2 on error.  The library takes ownership and calls release when the plan is released. */
struct CometShuffleBlockStream {
  int64_t (*next_block)(struct CometShuffleBlockStream* self, const uint8_t** data);
  const char* (*get_last_error)(struct CometShuffleBlockStream* self);
  void (*release)(struct CometShuffleBlockStream* self);
  void* private_data;
};

/* error kinds → Java exception class (native/jni-bridge/src/errors.rs:473-560) */
#define COMET_ERR_NATIVE 0          /* org/apache/comet/CometNativeException(msg) */
#define COMET_ERR_QUERY_EXECUTION 1 /* org/apache/comet/exceptions/CometQueryExecutionException(json) */

/* Replaces Java_org_apache_comet_Native_createPlan (native/core/src/execution/jni_api.rs:371-562;
 * Scala declaration spark/src/main/scala/org/apache/comet/Native.scala:60-79).
 *   plan/plan_len       protobuf bytes of spark.spark_operator.Operator (native/proto/src/proto/operator.proto:32)
 *   config/config_len   protobuf bytes of spark.spark_config.ConfigMap (may be NULL/0)
 *   inputs/input_kinds  one entry per Scan leaf in depth-first order (planner.rs:1726); the library takes
 *                       ownership of each stream and releases it when the plan is released (scan.rs:41-44)
 *   batch_size          spark.comet.batchSize: upper bound on rows per output batch (0 = unbounded)
 *   device_id           HIP device ordinal this plan runs on (one plan = one Spark partition = one GPU)
 * Nothing is pulled from the inputs before the first comet_execute_plan (jni_api.rs:795-797).
 * Returns a handle > 0, or 0 on error (message via comet_last_error(0)). */
int64_t comet_create_plan(const uint8_t* plan, size_t plan_len, const uint8_t* config, size_t config_len,
                          void** inputs, const int32_t* input_kinds, int32_t n_inputs, int32_t partition_count,
                          int32_t batch_size, int32_t device_id);

/* Replaces Java_org_apache_comet_Native_executePlan (jni_api.rs:767-957; Native.scala:98-103).
 * Writes one MOVED ArrowArray + ArrowSchema per output column into the caller-allocated structs
 * (prepare_output, jni_api.rs:674-742; array offset is 0; buffers live until the consumer calls release).
 * Returns the number of rows, -1 at end of stream, -2 on error. */
int64_t comet_execute_plan(int64_t handle, struct ArrowArray** out_arrays, struct ArrowSchema** out_schemas,
                           int32_t n_out);

/* Device-resident variant of comet_execute_plan for plans that end in Filter / Projection / HashJoin (not aggregates):
 * the whole result is ONE batch that stays in HBM; out_arrays[i] is filled as an ArrowDeviceArray with
 * device_type = ARROW_DEVICE_ROCM, device_id = the plan's device, sync_event = NULL (the plan's stream has been
 * synchronised) and null_count = -1 where a validity bitmap is present.  This is the stage boundary of a multi-GPU
 * plan: what the reference hands to its shuffle writer (ShuffleWriterExec input, native/shuffle/src/shuffle_writer.rs)
 * stays on the GPU for the RCCL exchange.  Returns rows, -1 at end of stream (second call), -2 on error. */
int64_t comet_execute_plan_device(int64_t handle, struct ArrowDeviceArray** out_arrays, struct ArrowSchema** out_schemas,
                                  int32_t n_out);

/* Replaces Java_org_apache_comet_Native_releasePlan (jni_api.rs:961-990). Safe mid-stream. */
void comet_release_plan(int64_t handle);

/* Error text / kind of the last failed call on this handle (handle 0: last failed comet_create_plan on
 * the calling thread).  The JNI shim turns it into the Java exception (errors.rs:832-850). */
const char* comet_last_error(int64_t handle);
int32_t comet_last_error_kind(int64_t handle);

/* Serialized spark.spark_metric.NativeMetricNode for the plan (what the reference pushes through
 * CometMetricNode.set_all_from_bytes, native/core/src/execution/metrics/utils.rs:30-45).
 * Returns the byte length; copies at most cap bytes into buf. */
int64_t comet_plan_metrics(int64_t handle, uint8_t* buf, size_t cap);

/* Fused-pipeline description (the counterpart of spark.comet.explain.native.enabled, jni_api.rs:816-820). */
const char* comet_explain(int64_t handle);

/* Kernel timing of the work done so far by this plan, measured with HIP events on the plan's own
 * stream: total milliseconds, number of timed launches, input rows.  Used by bench.py's roofline. */
void comet_plan_kernel_stats(int64_t handle, double* kernel_ms, int64_t* launches, int64_t* input_rows);

/* With COMET_KERNEL_TIMES=1 in the environment or after comet_set_kernel_times(1) (a measurement switch; off by default) every generated-kernel launch of the plan is bracketed
 * by its own HIP event pair; this returns the totals per kernel name as JSON ({"k_jprobe": {"ms": 1.2, "calls": 3}, …}; "{}" when the switch
 * is off): the byte length, at most cap - 1 bytes + NUL copied into buf.  bench.py's Q3 / Q95 rooflines name their dominant kernel from it. */
int64_t comet_plan_kernel_times(int64_t handle, char* buf, size_t cap);
/* … the same switch at run time (process-wide; plans created afterwards time their launches). */
void comet_set_kernel_times(int32_t on);

/* … and of the input-verification launches that run ahead of them (utf8_uniform_kernel over the Utf8 offsets of an HBM-resident input,
 * the check behind addressing fixed-length strings directly): total milliseconds and launches, timed apart from the main kernels so that
 * bench.py's roofline charges every column's bytes to the kernel that reads them. */
void comet_plan_aux_kernel_stats(int64_t handle, double* aux_ms, int64_t* aux_launches);

/* Decode + plan + generate + compile (hiprtc, gfx950) without touching a GPU.  Returns 0 on success and
 * writes a description into out (NUL terminated, truncated to cap), -2 on error (comet_last_error(0)). */
int32_t comet_compile_plan(const uint8_t* plan, size_t plan_len, char* out, size_t cap);

/* "Will createPlan accept this plan?" — decode + plan + generate, nothing compiled, no GPU touched (milliseconds): for the JVM side's
 * planning-time decision between the native stage and Spark's own (Comet decides per operator / expression while it serializes the
 * plan, spark/src/main/scala/org/apache/comet/serde/QueryPlanSerde.scala:743,910; a native library that refuses at createPlan would
 * fail the task instead).  Returns 0 and a description in out, or -2 and the refusal — the operator / expression by name — in out. */
int32_t comet_check_plan(const uint8_t* plan, size_t plan_len, char* out, size_t cap);

/* Spark-compatible murmur3 (seed chaining) + pmod partition ids for the exchange step — replaces
 * create_murmur3_hashes (native/spark-expr/src/hash_funcs/murmur3.rs:185-198) and pmod
 * (native/shuffle/src/comet_partitioning.rs:51-57).  All pointers are DEVICE pointers.
 *   type_id: spark DataTypeId (types.proto:43-66) of the column; precision for DECIMAL
 *   hashes:  in/out uint32 per row (initialise to 42 for the first column)
 * Returns 0, or -2 on error. */
int32_t comet_murmur3_column(int32_t type_id, int32_t precision, const void* values, const uint8_t* validity,
                             const void* aux_bytes, int64_t n, uint32_t* hashes, void* hip_stream);
int32_t comet_pmod_partition(const uint32_t* hashes, int64_t n, int32_t num_partitions, int32_t* partition_ids,
                             void* hip_stream);

/* Exchange bookkeeping on device — replaces ScratchSpace::map_partition_ids_to_starts_and_indices
 * (native/shuffle/src/partitioners/multi_partition.rs:54-103): from one partition id per row compute
 *   partition_starts[num_partitions + 1]  (int64, device)  — slice k = [starts[k], starts[k+1])
 *   partition_row_indices[n]              (uint32, device) — row numbers grouped by partition, ascending inside each
 * e.g. ids [3,1,1,1,2,2,0] → indices [6,1,2,3,4,5,0], starts [0,1,4,6,7] (the reference's own example, :78-84).
 * Synchronises hip_stream before returning.  n < 2^32, num_partitions ≤ 4096.  Returns 0, or -2 on error. */
int32_t comet_partition_indices(const int32_t* partition_ids, int64_t n, int32_t num_partitions, int64_t* partition_starts,
                                uint32_t* partition_row_indices, void* hip_stream);

/* dst[k] = src[row_indices[k]] for one buffer of a column — the per-partition `take` that builds the outgoing
 * batches (multi_partition.rs:457-520, partitioned_batch_iterator.rs).  width_bytes ∈ {1,2,4,8,16}, or 0 for
 * bit-packed buffers (validity bitmaps, Boolean values).  Asynchronous on hip_stream.  Returns 0, or -2 on error. */
int32_t comet_take_column(int32_t width_bytes, const void* src, const uint32_t* row_indices, int64_t n, void* dst,
                          void* hip_stream);

/* The same take for a Utf8 / Binary column (int32 offsets + bytes), in two steps because the byte total is only known after
 * the first: comet_take_utf8_offsets writes the n + 1 new offsets and RETURNS the total number of bytes (it synchronises
 * hip_stream; -2 on error); comet_take_utf8_bytes then copies the string bytes into out_bytes (asynchronous).
 * validity_bits (may be NULL) is the SOURCE column's validity bitmap: NULL rows contribute no bytes. */
int64_t comet_take_utf8_offsets(const int32_t* offsets, const uint8_t* validity_bits, const uint32_t* row_indices, int64_t n,
                                int32_t* out_offsets, void* hip_stream);
int32_t comet_take_utf8_bytes(const int32_t* offsets, const uint8_t* bytes, const uint8_t* validity_bits, const uint32_t* row_indices,
                              int64_t n, const int32_t* out_offsets, uint8_t* out_bytes, void* hip_stream);

/* ---- org.apache.comet.parquet.Native — the record-batch reader of the iceberg-compat scan path ----------------------------------
 * Replaces Java_org_apache_comet_parquet_Native_{initRecordBatchReader, readNextRecordBatch, currentColumnBatch, closeRecordBatchReader}
 * (native/core/src/parquet/mod.rs:135-330).  required / data schema: the bytes of an Arrow IPC stream's Schema message (what the JVM
 * sends; parquet/util/jni.rs:21-27); filter: one serialized spark_expression.Expr bound to data_schema, or NULL; starts / lengths: byte
 * ranges selecting row groups by their midpoint.  next returns the rows of the batch now current (0 = end of file); column MOVES one
 * column of the current batch into caller-allocated Arrow C structs.  Errors: 0 / -2 and comet_last_error(0). */
int64_t comet_parquet_reader_init(const char* file_path, int64_t file_size, const int64_t* starts, const int64_t* lengths, int32_t n_ranges,
                                  const uint8_t* filter, size_t filter_len, const uint8_t* required_schema_ipc, size_t required_len,
                                  const uint8_t* data_schema_ipc, size_t data_len, const char* session_timezone, int32_t batch_size,
                                  int32_t case_sensitive, int32_t device_id);
int32_t comet_parquet_reader_next(int64_t handle);
int32_t comet_parquet_reader_column(int64_t handle, int32_t column, struct ArrowArray* out_array, struct ArrowSchema* out_schema);
void comet_parquet_reader_close(int64_t handle);

/* ---- memory accounting (csrc/exec.hpp MemAccount) --------------------------------------------------------------------------------------
 * Replaces CometUnifiedMemoryPool (native/core/src/execution/memory_pools/unified_pool.rs:64-150): call right after comet_create_plan.
 * Every growth of the plan's PINNED HOST staging calls acquire(ctx, bytes) on the thread that is inside comet_execute_plan — the JNI shim
 * forwards to CometTaskMemoryManager.acquireMemory(J)J — and the plan fails with the reference's "Task N failed to acquire B bytes, only
 * got G. Reserved: R" when less is granted (the partial grant is released first); release(ctx, bytes) hands bytes back (releases that
 * happen on other threads are queued until the task thread's next call; comet_release_plan returns whatever is left).  HBM is held against
 * the plan's own budget, config key spark.comet.gpu.memory.limit (bytes).  stats: {host bytes in use, host peak, HBM in use, HBM peak}. */
int32_t comet_plan_set_memory_manager(int64_t plan, int64_t (*acquire)(void* ctx, int64_t bytes), void (*release)(void* ctx, int64_t bytes), void* ctx,
                                      int64_t task_id);
void comet_plan_memory_stats(int64_t plan, int64_t* out4);

/* ---- Parquet scan planning — diagnostic entry ------------------------------------------------------------------------------------------
 * What the scan of a serialized plan's NativeScan would read: the row groups its byte ranges select, those the pushed-down data_filters
 * rule out by min / max statistics or — `column = literal`, `column IN (literals)` — by the column chunks' Bloom filters, and (page_index & 1) the row
 * ranges of the survivors the page index leaves (page_index & 2: WITHOUT the Bloom filters, datafusion.execution.parquet.bloom_filter_on_read = false) — as JSON
 * {"rows", "row_groups_pruned", "row_groups_pruned_bloom_filter", "page_index_rows_pruned", "row_groups": [{"row_group", "num_rows", "keep": [[begin, end), …]}]}
 * (row_groups_pruned counts those the Bloom filters ruled out too).  Reads footers, Bloom filters and page indexes only; needs no GPU.  Returns the JSON's length (truncated to cap − 1 in `out`), or -2. */
int64_t comet_parquet_prune_report(const uint8_t* plan, size_t plan_len, int32_t page_index, char* out, size_t cap);
/* The two functions a Parquet Bloom filter is probed with (parquet-format BloomFilter.md): XXH64 of a value's PLAIN encoding, and the split-block test of a
 * hash against a filter's bitset (a multiple of 32 bytes).  Host only; what tests/test_parquet_bloom_cpu.py pins against the xxhash package and pyarrow's filters. */
uint64_t comet_xxh64(const uint8_t* data, size_t len, uint64_t seed);
int32_t comet_sbbf_might_contain(const uint8_t* bitset, size_t nbytes, uint64_t hash);
/* Diagnostic, host only: the PLAIN value bytes the scan stages for column `column` of the plan's NativeScan (selected row groups, page after
 * page, NULLs left out, BYTE_ARRAY values as 4-byte length + bytes) — DELTA_BINARY_PACKED / DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY / BYTE_STREAM_SPLIT pages are
 * rewritten as PLAIN on the host (the reference reads them through arrow-rs, parquet/parquet_exec.rs:60-211).  Returns the byte count (the first
 * `cap` bytes are copied), -2 on error (comet_last_error(0)). */
int64_t comet_parquet_host_plain_values(const uint8_t* plan, size_t plan_len, int32_t column, uint8_t* out, size_t cap);

/* ---- RLIKE pattern compiler (csrc/regex.cpp) — diagnostic entry -------------------------------------------------------------------------
 * Compiles `pattern` the way the planner does for an RLike expression (the exactly reproducible subset of the Rust regex syntax the
 * reference evaluates with, predicate_funcs/rlike.rs) and walks the resulting tables over `value` on the host: 1 match, 0 no match,
 * -2 and comet_last_error(0) for a pattern outside the subset.  Needs no GPU; the device walks the same tables. */
int32_t comet_rlike_match(const char* pattern, const uint8_t* value, size_t value_len);

/* ---- regexp_extract's matcher (csrc/regex.cpp compile_regex_captures + csrc/device/regex_vm.hpp) — diagnostic entry -----------------------
 * Compiles `pattern` for capture group `group` the way the planner does for ScalarFunc regexp_extract (string_funcs/regexp_extract.rs:
 * the crate's leftmost match, one group's span) and runs the device's matcher over `value` on the host: 1 a match, 0 none (the result is
 * the empty string either way when the group is unset), [*start, *start + *len) = the group's bytes inside `value`; -2 and
 * comet_last_error(0) for a pattern outside the subset or a group index out of range (the reference's message).  Needs no GPU. */
int32_t comet_regexp_extract_host(const char* pattern, int32_t group, const uint8_t* value, size_t value_len, int32_t* start, int32_t* len);

#ifdef COMET_TEST_ABI
/* ==== TEST ABI: exported for the repo's own tests and tools, NOT part of the product boundary — an integrator does not bind these (define COMET_TEST_ABI to see them) ====
 * ---- what the generator writes — diagnostic entries (tests/emu: the generated per-row code compiled and run on the HOST against the oracle) --------------------
 * comet_plan_codegen: the HIP source generated for a Filter / Projection / HashAggregate chain over one Scan leaf (has_valid[k]: column k arrives with a validity
 * bitmap) and what the executor needs to read its outputs, as JSON {"sink", "has_filter", "R", "kernels": [...], "out": [{"type", "precision", "scale", "nullable",
 * "gather_src", "view_src", "fmt_kind", "packed_string", "concat", "case_mode", "pad", "pad_left"}], "source"}: the length, the text written when it fits `cap`.
 * comet_embedded_header: the text of a header hiprtc compiles that source against ("comet_device.hpp", "kparams.h", "comet_ryu.hpp", "comet_strtod.hpp",
 * "comet_strts.hpp", "comet_regex_vm.hpp").  -2 and comet_last_error(0) on failure.  Neither needs a GPU. */
int64_t comet_plan_codegen(const uint8_t* plan, size_t plan_len, const uint8_t* has_valid, int32_t n_valid, char* out, int64_t cap);
int64_t comet_embedded_header(const char* name, char* out, int64_t cap);
/* the Spark error JSON of a raise site of a pipeline generated in this process (the site id a kernel leaves in the error block's detail words, kparams.h) and the
 * detail it left: what check_device_errors throws, without the SQL context */
int64_t comet_error_site_json(uint32_t site_id, uint64_t lo, uint64_t hi, const uint8_t* str, int64_t str_avail, char* out, int64_t cap);
/* a streaming read of `bytes` bytes with `width`-byte loads (4 / 8 / 16) into a sink word: the known byte count tools/pmc_calibrate.py calibrates FETCH_SIZE on */
int comet_calib_read(const void* buf, int64_t bytes, int32_t width, uint64_t* sink, void* stream);
/* the headline's Utf8 pass on its own (kernels_static.hip utf8_uniform_kernel): *flag becomes non-zero unless every value of the n offsets has length L */
int comet_launch_utf8_uniform(const int32_t* offsets, int64_t n, int32_t L, uint32_t* flag, void* stream);
#endif /* COMET_TEST_ABI */

/* ---- scalar subqueries (expr.proto:513-516 Subquery{id, datatype}; native/core/src/execution/expressions/subquery.rs:72-180) ------------------------------
 * The reference asks the JVM for a subquery's value when the expression is first evaluated: CometScalarSubquery.isNull / getBoolean / getByte / getShort / getInt /
 * getLong / getFloat / getDouble / getDecimal / getString / getBinary (planId, id) (jni-bridge/src/comet_exec.rs:54-126).  Here every Subquery of a plan becomes a
 * literal at the first comet_execute_plan.  The values come from a table filled with comet_plan_set_subquery (JVM-free callers) or from a provider callback (the
 * JNI shim registers one that calls those static methods).  Value bytes: integers / dates / timestamps 8-byte little-endian, booleans 1 byte, floats and doubles an
 * 8-byte double, decimals BigInteger.toByteArray (big-endian two's complement of the unscaled value), strings / binary as they are.
 * provider: → 1 and *is_null / out[0, *len) (when *len > cap it is called once more with room), 0: no such subquery, < 0: failed.  type_id: types.proto's. */
typedef int32_t (*comet_subquery_provider)(void* ctx, int64_t id, int32_t type_id, int32_t* is_null, uint8_t* out, int64_t cap, int64_t* len);
int32_t comet_plan_set_subquery(int64_t handle, int64_t id, int32_t is_null, const uint8_t* value, size_t value_len);
int32_t comet_plan_set_subquery_provider(int64_t handle, comet_subquery_provider provider, void* provider_ctx);

/* The string functions and digests of csrc/device/strfn.hpp on the host — diagnostic entry: op 1 reverse, 2 repeat(k), 3 replace(a, b), 4 substring_index(a, k),
 * 10 md5, 11 sha1, 12-15 sha224 / 256 / 384 / 512 (hexadecimal digits): the result's length, its bytes written when they fit `cap`; op 20 crc32, 21 instr(a),
 * 22 ascii: the value.  Needs no GPU. */
int64_t comet_strfn_host(int32_t op, const uint8_t* value, int32_t n, const uint8_t* a, int32_t na, const uint8_t* b, int32_t nb, int64_t k, uint8_t* out, int64_t cap);

/* regexp_extract_all(value, pattern, group) (string_funcs/regexp_extract_all.rs) by the device's two passes (rx_find_all) on the host: the number of matches,
 * the group's (start, length) per match written while they fit `cap`; -2 and comet_last_error(0).  Needs no GPU. */
int32_t comet_extract_all_host(const char* pattern, int32_t group, const uint8_t* value, size_t value_len, int32_t* starts, int32_t* lens, int32_t cap);

/* split(value, pattern, limit) (string_funcs/split.rs:434-472) by the device's two passes (csrc/device/regex_vm.hpp rx_split) on the host: the
 * number of pieces, their (start, length) inside `value` written while they fit `cap`; -2 and comet_last_error(0) for a pattern outside the
 * subset.  Needs no GPU. */
int32_t comet_split_host(const char* pattern, int32_t limit, const uint8_t* value, size_t value_len, int32_t* starts, int32_t* lens, int32_t cap);

/* The calendar functions of the generated kernels (csrc/device/dates.hpp) on the host — diagnostic entry: fn 0 date_part(days, part), 1 isodow, 2 ISO week,
 * 3 date_trunc(days, unit), 4 last_day, 5 next_day(days, weekday Monday = 0), 6 make_date(y, m, d), 7 timestamp_trunc(wall-clock µs, unit).  → 1 and *out,
 * 0 = NULL, -2 and comet_last_error(0).  Needs no GPU. */
int32_t comet_date_fn_host(int32_t fn, int64_t a, int64_t b, int64_t c, int64_t* out);

/* ---- time zones (csrc/tz.cpp) -------------------------------------------------------------------------------------------------------------
 * The table a Cast / date-part expression with `zone` as its time zone is planned with: { n, offset before the first transition, first instant
 * the table does not answer, n transition instants (UTC seconds), n offsets (seconds east of UTC) } — read from the system's time-zone
 * database ($TZDIR, /usr/share/zoneinfo; the reference resolves the same names with chrono-tz, conversion_funcs/temporal.rs:58), fixed
 * offsets ("UTC", "+05:30") without it.  Returns the number of words (written when cap suffices), or -2 and comet_last_error(0).  Needs no GPU. */
int64_t comet_zone_table(const char* zone, int64_t* out, int64_t cap);

/* ---- Spark error JSON (csrc/err_sites.cpp) ---------------------------------------------------------------------------------------------
 * What a failing task throws through CometQueryExecutionException when the error names the offending value: {"errorType", "errorClass",
 * "params": {"value", "precision", "scale"} | {"value", "fromType", "toType"}} exactly as native/common/src/error.rs:318-380 serialises it and
 * spark/…/ShimSparkErrorConverter.scala reads it back.  This entry formats ONE such error from a raise site's static part and the value the
 * kernel left in its error block (value_kind: 0 unscaled decimal in (hi, lo), 1 integer + literal suffix, 2 double as "{:e}D", 3 float as
 * "{:e}", 4 the first str_avail bytes of a string of lo bytes, 5 decimal(precision, scale) + "BD", 6 double as Rust prints it, 7 integer) —
 * the executor calls the same routine; exported so that the formats are testable without a GPU.  Returns the JSON's length (written when cap
 * suffices). */
int64_t comet_error_json(const char* error_type, const char* error_class, const char* from_type, const char* to_type, int32_t precision, int32_t scale,
                         int32_t value_kind, const char* suffix, uint64_t lo, uint64_t hi, const uint8_t* str, int64_t str_avail, char* out, int64_t cap);

/* The same for the site_index-th raise site — in the order they are generated — of a Projection / Filter plan whose expressions carry a
 * QueryContext (expr.proto:103-141, planner.rs:302-316): the JSON then holds "context" and "summary" like SparkErrorWithContext::to_json
 * (error.rs:806-831).  site_index -1 / -2: the DecimalSumOverflow of the pipeline's ANSI decimal sum / average, with the aggregate's context.
 * Generates the plan's kernel text, compiles and runs nothing.  Returns the length, or -2 and comet_last_error(0). */
int64_t comet_plan_error_json(const uint8_t* plan, size_t plan_len, int32_t site_index, uint64_t lo, uint64_t hi, const uint8_t* str, int64_t str_avail,
                              char* out, int64_t cap);

/* The same by the site's ID — what a kernel leaves in the error block's detail words (kparams.h) — with the QueryContext the plan's pipeline gives that site, if
 * any: exactly what check_device_errors throws for it (the host emulation of generated code in tests/emu rebuilds the executor's error with this). */
int64_t comet_plan_site_error_json(const uint8_t* plan, size_t plan_len, uint32_t site_id, uint64_t lo, uint64_t hi, const uint8_t* str, int64_t str_avail,
                                   char* out, int64_t cap);

/* ---- the host page codecs (csrc/parquet_meta.cpp) ---------------------------------------------------------------------------------------
 * What the scan's host threads run on the pages the device does not decompress itself: Parquet CompressionCodec 0 UNCOMPRESSED, 1 SNAPPY,
 * 2 GZIP, 5 LZ4 (Hadoop-framed, or one raw block), 6 ZSTD, 7 LZ4_RAW; dst_len is the page header's uncompressed_page_size and must match exactly.  Needs no GPU.  0, or -2 and
 * comet_last_error(0). */
int32_t comet_page_decompress(int32_t codec, const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len);
/* Diagnostic entry for the scan's sparse reads of snappy pages (parquet_meta.hpp SnappyView: the run headers of a dictionary-encoded page
 * are read through the compressed stream, the page itself is inflated on the device): the bytes at `offsets` of the stream's OUTPUT.
 * Returns the uncompressed length, -1 when the stream is malformed or has more than max_elems elements (the scan then inflates the
 * page on the host), -2 on a reference outside the page (comet_last_error(0)). */
int64_t comet_snappy_view_read(const uint8_t* src, size_t src_len, int32_t max_elems, const int64_t* offsets, int32_t n, uint8_t* out);

/* ---- device-side page decompression (csrc/snappy_kernels.hip) — diagnostic entry ------------------------------------------------------
 * The Parquet scan ships snappy-compressed PLAIN data pages across PCIe as they are and one GPU workgroup per page decompresses them
 * (the reference decompresses on the task's CPU core inside the parquet crate's page reader, driven from native/core/src/parquet/mod.rs).
 * This call runs that kernel over `npages` raw snappy streams in host memory — for parity tests and for timing the kernel alone; the
 * scan itself never goes through host memory.  Returns 0; (page << 8 | code) for the first corrupt page (code 1 preamble / length,
 * 2 truncated, 3 bad copy offset, 4 output overrun, 5 short output); -1 for a HIP error.  *kernel_ms: the launch's duration. */
int64_t comet_snappy_inflate_pages(const uint8_t* streams, const int64_t* stream_off, const int32_t* stream_len, const int32_t* page_len,
                                   int32_t npages, uint8_t* out, const int64_t* out_off, int32_t device_id, double* kernel_ms);
/* The same through the MULTI-KERNEL pipeline the scan uses since round 3 (csrc/device/snappy2.hpp: window transfer functions → chunk
 * functions → one hop per chunk → element list → pointer jumping per 64 KiB fragment; pages that are not fragment-shaped fall back to the
 * one-wave kernel inside the same call).  status_out (optional, npages words): what the pipeline made of each page before the fallback —
 * 0 decoded, 1 handed to the one-wave kernel, >= 16 corrupt.  *kernel_ms: all launches of one decompression (scratch already sized). */
int64_t comet_snappy2_inflate_pages(const uint8_t* streams, const int64_t* stream_off, const int32_t* stream_len, const int32_t* page_len,
                                    int32_t npages, uint8_t* out, const int64_t* out_off, int32_t device_id, double* kernel_ms, uint32_t* status_out);
/* zstd pages (csrc/device/zstd2.hpp): the host walks each frame's block headers — sizes, modes, where the table descriptions and
 * bitstreams sit — and the device does the rest: Huffman literals a lane per stream, FSE sequences a lane per block, then the sequences are
 * executed by pointer jumping per 64 KiB fragment (the reference inflates zstd pages on the task's CPU core through the zstd crate, like
 * every other codec of the parquet crate's page reader).  `npages` single-frame streams of page_len[i] bytes each.  status_out (optional):
 * 0 decoded, 1 the host walk keeps this page on the host (dictionary id, several frames, a content size that differs from page_len — it is
 * left out of the launch and its output untouched), >= 16 corrupt.  Returns 0, (page << 8 | code) of the first corrupt page, -1 HIP error. */
int64_t comet_zstd2_inflate_pages(const uint8_t* streams, const int64_t* stream_off, const int32_t* stream_len, const int32_t* page_len,
                                  int32_t npages, uint8_t* out, const int64_t* out_off, int32_t device_id, double* kernel_ms, uint32_t* status_out);

/* ---- in-library hash exchange between GPUs (csrc/exchange.cpp) ----------------------------------------------------
 * The step Spark's exchange performs between two native stages, done GPU to GPU: rows are hash-partitioned exactly as the reference's
 * shuffle writer does (murmur3 seed 42 chained over the key columns → pmod → partition_starts / partition_row_indices,
 * native/shuffle/src/partitioners/multi_partition.rs:54-103, comet_partitioning.rs:51-57) and every rank receives partition `rank`:
 * sender after sender in rank order, each sender's rows in input order.
 *   comet_comm_unique_id / comet_comm_init_rank   one process per GPU: RCCL (dlopen'ed); the 128-byte id travels out of band
 *   comet_comm_init_local                         N task threads of one process (one GPU each, or shared): in-process rendezvous +
 *                                                 peer copies; every rank of `group_id` must call it with the same world size
 *   comet_comm_init_tcp                           one process per rank over TCP sockets (host memory on the wire)
 * All calls return 0 / a positive handle on success; on failure -2 / 0 and comet_exchange_last_error() (thread local) tells why. */
typedef struct CometExchangeColumn {
  int32_t type_id;          /* spark_expression.DataType.DataTypeId: fixed-width types, BOOL (bit-packed values), STRING / BYTES */
  int32_t precision;        /* decimals: selects the 8- or 16-byte hash form (hash_funcs/utils.rs:573-760) */
  const void* values;       /* device pointer; STRING / BYTES: the rows + 1 int32 offsets */
  const uint8_t* validity;  /* device Arrow bitmap or NULL */
  const uint8_t* aux;       /* STRING / BYTES: the value bytes; NULL otherwise */
} CometExchangeColumn;
int32_t comet_comm_unique_id(uint8_t* out128);
int64_t comet_comm_init_rank(const uint8_t* id128, int32_t world, int32_t rank, int32_t device_id);
int64_t comet_comm_init_local(int64_t group_id, int32_t world, int32_t rank, int32_t device_id);
/* TCP transport: one process per rank, host memory on the wire (HBM buffers are staged through pinned memory) — for ranks without an
 * RCCL-capable fabric between them, and what the CPU-only multi-process tests drive the exchange over.  `peers` = "host:port,host:port,…",
 * one entry per rank (rank r listens on its own port); blocks until every pair of ranks is connected or `timeout_ms` (0 = 60 s) passed.
 * A peer that dies or stays silent for `timeout_ms` during an exchange fails that exchange with an error naming it — nothing hangs. */
int64_t comet_comm_init_tcp(const char* peers, int32_t world, int32_t rank, int32_t device_id, int32_t timeout_ms);
/* "rccl" | "tcp" | "in-process" | "none (1 rank)": the wire this communicator moves its slices over */
const char* comet_comm_transport(int64_t comm);
/* out4[0] = ranks of the communicator as the wire itself reports them (RCCL: ncclCommCount, -1 if the library lacks the entry), out4[1] = this
 * rank there (ncclCommUserRank), out4[2] / out4[3] = bytes sent to / received from other ranks over RCCL so far.  0, or -1 (comet_last_error).
 * No reference counterpart: the reference's exchange is Spark's shuffle (shuffle_writer.rs:166-300); this is bench / test evidence. */
int32_t comet_comm_stats(int64_t comm, int64_t* out4);
void comet_comm_destroy(int64_t comm);
/* Collective: every rank of the communicator calls it with its shard (rows may be 0).  Returns a result handle. */
int64_t comet_exchange(int64_t comm, int32_t n_cols, const CometExchangeColumn* cols, int64_t rows, const int32_t* key_cols, int32_t n_keys);
int64_t comet_exchange_result_rows(int64_t result);
int32_t comet_exchange_result_column(int64_t result, int32_t col, void** values, void** validity);   /* device pointers owned by the result */
/* STRING / BYTES columns: the received value bytes (values = the rows + 1 rebuilt int32 offsets); NULL / 0 for other columns */
int32_t comet_exchange_result_aux(int64_t result, int32_t col, void** bytes, int64_t* n_bytes);
void comet_exchange_result_release(int64_t result);
const char* comet_exchange_last_error(void);

/* Replaces Java_org_apache_comet_Native_decodeShuffleBlock (native/core/src/execution/jni_api.rs:1163-1181 →
 * read_ipc_compressed, native/shuffle/src/ipc.rs:23-52): decodes ONE shuffle block — 4-byte codec tag "NONE" / "ZSTD" / "LZ4_" /
 * "SNAP" followed by the (compressed) Arrow IPC stream — and moves its single record batch into the caller-allocated Arrow C Data
 * structs, one per column, dictionary-encoded columns unpacked.  Host-side only (the block is host bytes and so is the result).
 * Returns the number of rows, or -2 on error (comet_last_error(0)). */
int64_t comet_decode_shuffle_block(const uint8_t* block, int64_t len, struct ArrowArray** out_arrays, struct ArrowSchema** out_schemas,
                                   int32_t n_out);

#ifdef COMET_TEST_ABI
/* (TEST ABI) What a Scan / ShuffleScan leaf does with the batches of one chunk of a NESTED input column before it uploads them (csrc/exec_util.cpp
 * append_nested_rows; the reference's ScanExec takes the batches as they come, operators/scan.rs:134-164, and DataFusion's kernels honour
 * a struct's validity themselves): `n` Arrow arrays of one struct / list column (any depth; offsets and slices as the producer left them) are
 * concatenated into ONE host column — list offsets rebased, children appended, every field's validity masked by its struct's — which is moved
 * into *out / *out_schema.  Host-side only; the test entry of that step.  Returns the rows, or -2 on error. */
int64_t comet_concat_nested_column(struct ArrowArray** arrays, struct ArrowSchema* schema, int32_t n, struct ArrowArray* out, struct ArrowSchema* out_schema);
#endif /* COMET_TEST_ABI */

/* The framing step of the shuffle writer on its own (ShuffleBlockWriter::write_batch, native/shuffle/src/writers/
 * shuffle_block_writer.rs:179-238): encodes the host-resident columns (Arrow C Data, one array + schema per column, any offset) as ONE
 * complete block — u64le length, u64le field count, codec tag, Arrow IPC stream under `codec` (0 none, 1 zstd, 2 lz4 frame,
 * 3 snappy framing; operator.proto:679-686).  The ShuffleWriter operator (106) runs the same code after the GPU has partitioned the
 * rows.  *out is malloc'ed (free with comet_free_buffer); zero rows produce no block (*out_len = 0).  Returns 0, or -2 on error. */
int32_t comet_encode_shuffle_block(struct ArrowArray** arrays, struct ArrowSchema** schemas, int32_t n_cols, int32_t codec,
                                   int32_t compression_level, uint8_t** out, int64_t* out_len);
void comet_free_buffer(uint8_t* p);

/* Replace Java_org_apache_comet_Native_columnarToRow{Init,Convert,Close} (native/core/src/execution/jni_api.rs:1253-1377 →
 * ColumnarToRowContext, columnar_to_row.rs:866-1345): Arrow columns → Spark UnsafeRow bytes.  init returns a handle (0 on error).
 * convert takes OWNERSHIP of the n_cols Arrow C Data structs (host memory, one array + schema per column, offset 0; released before it
 * returns), converts the first num_rows rows on the GPU of the handle and points *out_buffer at the row bytes, *out_offsets / *out_lengths
 * at num_rows int32 entries — all in host memory owned by the handle and valid until its next convert or close.  Row layout as the
 * reference writes it: null bitset, one 8-byte slot per field (integers sign-extended, floats as their bits, Decimal128(p ≤ 18) as the
 * unscaled long, variable-length fields as (offset << 32) | length), then the variable-length data padded to 8 bytes; wide decimals as
 * their minimal big-endian two's-complement bytes.  Returns 0, or -2 on error (comet_columnar_to_row_error). */
int64_t comet_columnar_to_row_init(int32_t batch_size, int32_t device_id);
int32_t comet_columnar_to_row_convert(int64_t handle, struct ArrowArray** arrays, struct ArrowSchema** schemas, int32_t n_cols, int64_t num_rows,
                                      const uint8_t** out_buffer, const int32_t** out_offsets, const int32_t** out_lengths);
void comet_columnar_to_row_close(int64_t handle);
const char* comet_columnar_to_row_error(int64_t handle);

/* Replace Java_org_apache_comet_Native_sortRowPartitionsNative (native/core/src/execution/jni_api.rs:1130-1160): sorts the n packed
 * i64 records (partition id in the high bits, row pointer below) of Spark's shuffle sorter in place, ascending as signed longs. */
void comet_sort_row_partitions(int64_t* records, int64_t n);

/* Replace Java_org_apache_comet_Native_writeSortedFileNative (jni_api.rs:1043-1127 → process_sorted_row_partition,
 * native/shuffle/src/spark_unsafe/row.rs:1342-1438): reads row_num Spark UnsafeRows (address + size each) of n_cols fields whose types
 * are the serialized spark_expression.DataType messages, and APPENDS them to file_path as shuffle blocks of at most batch_size rows
 * (same block format as comet_encode_shuffle_block).  compression_codec is "zstd" | "lz4" | "snappy"; anything else means lz4, as in
 * the reference.  checksum_algo: 0 CRC32, 1 Adler32, 2 CRC32C, continued from current_checksum unless that is INT64_MIN.
 * out_result = {bytes written, checksum or INT64_MIN when disabled, nanoseconds spent encoding}.  Host-memory work (the rows are JVM
 * off-heap pages).  Returns 0, or -2 on error (comet_last_error(0)). */
int32_t comet_write_sorted_rows(const int64_t* row_addresses, const int32_t* row_sizes, int64_t row_num, const uint8_t* const* serialized_datatypes,
                                const int32_t* datatype_lens, int32_t n_cols, const char* file_path, int32_t batch_size, int32_t checksum_enabled,
                                int32_t checksum_algo, int64_t current_checksum, const char* compression_codec, int32_t compression_level,
                                int64_t out_result[3]);

/* Host-only description of a Parquet footer as parsed by the library's own Thrift reader (rows, row groups, schema
 * elements, per-chunk codec/offsets) — the metadata the NativeScan path (native/core/src/parquet/parquet_exec.rs:60-211)
 * plans from.  Returns 0, or -2 on error. */
int32_t comet_parquet_describe(const char* path, char* out, size_t cap);

/* Library identity (NativeBase.java:82-106 loads "comet"). */
const char* comet_version(void);

/* The compiler behind createPlan's JIT in THIS process: "hiprtc <major.minor>; comgr <major.minor> <path of the libamd_comgr that compiles>".  Part of
 * the code-object cache key.  A JVM executor gets the installed ROCm's (libcomet.so's RUNPATH); a process that loaded another ROCm first (a Python wheel
 * with a bundled one) gets that one's unless the installed libamd_comgr was loaded into the global scope before it — kernels compiled by ROCm 7.0's
 * clang 20 ran 10-15 % slower than ROCm 7.2's clang 22 on the join kernels (profiles/r6_jit_compiler.md).  No reference counterpart (DataFusion is
 * compiled ahead of time, native/core/Cargo.toml). */
const char* comet_jit_toolchain(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
