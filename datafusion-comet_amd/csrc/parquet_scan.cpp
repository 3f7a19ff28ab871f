// NativeScan: Parquet files → Arrow columns resident in HBM.
// Host: footer + page headers (Thrift), page decompression, hybrid-run headers → PqPage / PqRun tables.
// Device: every level and value is decoded by parquet_kernels.hip.
// Reference path: NativeScan arm planner.rs:1523-1668 → init_datasource_exec parquet/parquet_exec.rs:60-211
// (column model SURVEY Appendix C.12: output = required_schema fields; row groups chosen by byte-range midpoint).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cerrno>
#include <atomic>
#include <cstdarg>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <exception>
#include <mutex>
#include <thread>

#include <sched.h>

#include "exec.hpp"
#include "parquet_dev.h"
namespace comet { void pool_miss_counters(int64_t out[4]); }
namespace comet { namespace detail {      // per-process stream / event pools (exec_memory.cpp)
hipStream_t pool_get_stream(int dev);
void pool_put_stream(int dev, hipStream_t s);
hipEvent_t pool_get_event(int dev);
void pool_put_event(int dev, hipEvent_t e);
int plans_executing();
hipStream_t shared_copy_stream(int dev);
} }
#include "device/snappy2.hpp"
#include "snappy2.hpp"
#include "zstd2.hpp"
#include "device/zstd2.hpp"
#include "parquet_meta.hpp"

extern "C" int comet_launch_fill(int width, void* dst, int64_t n, const void* value, void* stream);
extern "C" int comet_launch_take(int width, const void* src, const uint32_t* idx, int64_t n, void* dst, void* stream);
extern "C" int comet_launch_fill_utf8(int32_t* offsets, uint8_t* bytes, int64_t n, int32_t base, int32_t len, const uint8_t* dev_value, void* stream);
extern "C" {
void pq_launch_validity(const PqDecodeArgs* a, void* st);
void pq_launch_vidx(const uint8_t* valid, int64_t n, uint64_t* tiles, uint32_t* vidx, void* st);
void pq_launch_levels(const PqDecodeArgs* a, int which, uint8_t* out, void* st);
void pq_launch_level_ge(const uint8_t* lv, int64_t n, int thr, uint8_t* out, void* st);
void pq_launch_list_elem_entries(const uint8_t* def, int64_t n, int def_slot, const int32_t* elem_idx, uint32_t* entries, void* st);
void pq_launch_list_flags(const uint8_t* def, const uint8_t* rep, int64_t n, int def_slot, uint32_t* starts, uint32_t* elems, void* st);
void pq_launch_list_assemble(const uint8_t* def, const uint8_t* rep, int64_t n, int64_t rows, int def_list, int def_slot, int max_def, const int32_t* start_idx, const int32_t* elem_idx,
                             const uint8_t* values, int width, int32_t* offsets, uint8_t* list_valid, uint8_t* elem_valid, uint8_t* elem_values, uint32_t* err, void* st);
void pq_launch_decode_fixed(const PqDecodeArgs* a, void* st);
void pq_launch_decode_runs(const PqDecodeArgs* a, void* st);
void pq_launch_store_u32(const uint32_t* src, uint32_t* dst, void* st);
void pq_launch_count_runs(const PqPendingRuns* pend, int n, const uint8_t* bytes, uint32_t* counts, uint32_t* err, void* st);
void pq_launch_write_runs(const PqPendingRuns* pend, int n, const uint8_t* bytes, const int32_t* offsets, int32_t run_base, PqRun* runs, PqPage* pages, void* st);
void pq_launch_expand_nulls(const PqDecodeArgs* a, void* st);
void pq_launch_string_lengths(const PqDecodeArgs* a, void* st);
void pq_launch_string_copy(const PqDecodeArgs* a, void* st);
void pq_launch_u32_scan(const uint32_t* in, int64_t n, uint64_t* tiles, int32_t* out, void* st);
void pq_launch_pack(const uint8_t* bytes, uint8_t* bitmap, int64_t n, void* st);
void pq_launch_snappy(const PqInflate* jobs, int njobs, uint8_t* bytes, uint32_t* err, void* st);
void pq_launch_upload(const PqCopyDesc* descs, int n, void* st);
}

namespace comet {

namespace {

// Process-wide pool of scan threads (the counterpart of the reference's tokio worker threads, jni_api.rs:133-170): shared by every
// plan of the process, started on first use, never torn down.  Persistent threads keep their scratch buffers across scans.
class ScanPool {
 public:
  static ScanPool& get() {
    static ScanPool* p = new ScanPool();   // intentionally leaked
    return *p;
  }
  int size() const { return (int)nthreads_; }
  void submit(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      q_.push_back(std::move(f));
    }
    cv_.notify_one();
  }

 private:
  ScanPool() {
    // CPUs this process may actually burn: the cgroup quota (containers: a 256-thread host often grants 16 CPUs; more busy
    // threads than that only get throttled — measured 10.5 GB/s of LZ4 with 64 threads against 18.6 GB/s with 32 under a
    // 16-CPU quota), the affinity mask, and half the hardware threads (SMT siblings add little to byte-crunching loops)
    size_t n = std::min<size_t>(64, std::max(1u, std::thread::hardware_concurrency() / 2));
    {
      cpu_set_t set;
      if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<size_t>(n, (size_t)std::max(1, CPU_COUNT(&set)));
      long long quota = -1, period = 100000;
      if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "max 100000" or "<quota> <period>"
        char q[64] = {0};
        if (fscanf(f, "%63s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
      } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
        fclose(f1);
        if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
          if (fscanf(f2, "%lld", &period) != 1) period = 100000;
          fclose(f2);
        }
      }
      if (quota > 0 && period > 0) n = std::min<size_t>(n, (size_t)std::max<long long>(1, (quota + period - 1) / period) * 2);
    }
    if (const char* e = getenv("COMET_SCAN_THREADS")) n = (size_t)std::max(1, atoi(e));
    nthreads_ = n;
    for (size_t i = 0; i < n; i++) std::thread([this]() { run(); }).detach();
  }
  void run() {
    while (true) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !q_.empty(); });
        f = std::move(q_.front());
        q_.pop_front();
      }
      f();
    }
  }
  size_t nthreads_ = 1;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> q_;
};

}  // namespace

// fn(0) … fn(n-1) on the scan threads; returns when all are done (used for the staging copies of host input batches too)
void scan_pool_submit(std::function<void()> fn) { ScanPool::get().submit(std::move(fn)); }

void scan_pool_parallel(size_t n, const std::function<void(size_t)>& fn) {
  if (n == 0) return;
  if (n == 1) { fn(0); return; }
  struct State { std::mutex mu; std::condition_variable cv; size_t left; std::exception_ptr err; };
  auto st = std::make_shared<State>();
  st->left = n;
  for (size_t i = 0; i < n; i++) {
    ScanPool::get().submit([st, i, &fn]() {
      try {
        fn(i);
      } catch (...) {
        std::lock_guard<std::mutex> lk(st->mu);
        if (!st->err) st->err = std::current_exception();
      }
      {
        std::lock_guard<std::mutex> lk(st->mu);
        st->left--;
      }
      st->cv.notify_all();
    });
  }
  std::unique_lock<std::mutex> lk(st->mu);
  st->cv.wait(lk, [&] { return st->left == 0; });
  if (st->err) std::rethrow_exception(st->err);
}

namespace {

// A Parquet file opened for positional reads.  Column chunks are pread() by the scan threads into their own scratch
// (no mmap: tearing down a mapping of several hundred MB costs milliseconds of page-table work after every scan).
// the file is not there (errno ENOENT): Spark tells that apart from a file it cannot read (jni-bridge/src/errors.rs:633-655)
struct FileMissing : CometError {
  explicit FileMissing(const std::string& m) : CometError(m) {}
};
struct OpenFile {
  int fd = -1;
  size_t size = 0;
  std::vector<uint8_t> footer;   // "PAR1" + FileMetaData + length + "PAR1": what parse_footer needs
  explicit OpenFile(const std::string& path) {
    fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) {
      if (errno == ENOENT) throw FileMissing("Object at location " + path + " not found");      // (object_store's NotFound, word for word: the JVM side cuts the path out of it)
      throw CometError("cannot open Parquet file " + path + ": " + strerror(errno));
    }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); fd = -1; throw CometError("cannot stat " + path); }
    size = (size_t)st.st_size;
    try {
      uint8_t head[4], tail[8];
      if (size < 12) throw CometError("not a Parquet file (too short): " + path);
      read_at(head, 4, 0);
      read_at(tail, 8, (int64_t)size - 8);
      if (memcmp(tail + 4, "PAR1", 4) != 0 || memcmp(head, "PAR1", 4) != 0)      // (the magic before the length, as parquet-rs reads a footer)
        throw CometError("not a Parquet file (missing PAR1 magic; encrypted footers are not supported): " + path);
      uint32_t mlen;
      memcpy(&mlen, tail, 4);
      if ((size_t)mlen + 12 > size) throw CometError("parquet: bad footer length");
      footer.resize(4 + (size_t)mlen + 8);
      memcpy(footer.data(), head, 4);
      read_at(footer.data() + 4, (size_t)mlen + 8, (int64_t)size - 8 - (int64_t)mlen);
    } catch (...) {
      close(fd);
      fd = -1;
      throw;
    }
  }
  void read_at(uint8_t* dst, size_t n, int64_t off) const {
    size_t got = 0;
    while (got < n) {
      ssize_t r = pread(fd, dst + got, n - got, (off_t)(off + (int64_t)got));
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) throw CometError("parquet: short read");
      got += (size_t)r;
    }
  }
  ~OpenFile() {
    if (fd >= 0) close(fd);
  }
  OpenFile(const OpenFile&) = delete;
  OpenFile& operator=(const OpenFile&) = delete;
};

std::string path_from_uri(const std::string& uri) {
  // file_path is a URL-encoded URI (planner.rs:410-415)
  std::string s = uri;
  if (s.rfind("file://", 0) == 0) s = s.substr(7);
  else if (s.rfind("file:", 0) == 0) s = s.substr(5);
  std::string out;
  for (size_t i = 0; i < s.size(); i++) {
    if (s[i] == '%' && i + 2 < s.size() && isxdigit((unsigned char)s[i + 1]) && isxdigit((unsigned char)s[i + 2])) {
      out.push_back((char)strtol(s.substr(i + 1, 2).c_str(), nullptr, 16));
      i += 2;
    } else {
      out.push_back(s[i]);
    }
  }
  return out;
}

bool iequals(const std::string& a, const std::string& b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); i++)
    if (tolower((unsigned char)a[i]) != tolower((unsigned char)b[i])) return false;
  return true;
}

// parse the run headers of an RLE/bit-packed hybrid section living at staged[pos, end)
// `get(pos)` = byte at staged position pos: a pointer into decompressed bytes, or a sparse view of a still-compressed page (SnappyView)
template <class Get>
void parse_hybrid_runs_from(const Get& get, size_t pos, size_t end, int bw, int32_t max_values, std::vector<PqRun>& runs) {
  int32_t vstart = 0;
  const int vbytes = (bw + 7) / 8;
  while (pos < end && (max_values < 0 || vstart < max_values)) {
    uint64_t h = 0;
    int sh = 0;
    while (true) {
      if (pos >= end) throw CometError("parquet: truncated hybrid run header");
      uint8_t b = get(pos++);
      h |= (uint64_t)(b & 0x7f) << sh;
      if (!(b & 0x80)) break;
      sh += 7;
      if (sh > 56) throw CometError("parquet: hybrid run header longer than eight bytes");
    }
    PqRun r;
    memset(&r, 0, sizeof r);
    r.value_start = vstart;
    if (h & 1) {
      int64_t groups = (int64_t)(h >> 1);
      if (groups > (int64_t)(INT32_MAX - vstart) / 8) throw CometError("parquet: hybrid run count out of range");
      r.is_rle = 0;
      r.count = (int32_t)(groups * 8);
      r.byte_off = (int64_t)pos;
      if (pos + (size_t)(groups * bw) > end) {      // (device/pq_runs.hpp makes the same decision)
        const int64_t wanted = max_values >= 0 ? (int64_t)(max_values - vstart) : (int64_t)r.count;
        if (max_values < 0 || pos + (size_t)((wanted * bw + 7) / 8) > end) throw CometError("parquet: truncated bit-packed run");
      }
      pos += (size_t)(groups * bw);
    } else {
      r.is_rle = 1;
      if ((h >> 1) > (uint64_t)(INT32_MAX - vstart)) throw CometError("parquet: hybrid run count out of range");
      r.count = (int32_t)(h >> 1);
      uint32_t v = 0;
      for (int k = 0; k < vbytes; k++) {
        if (pos >= end) throw CometError("parquet: truncated RLE run");
        v |= (uint32_t)get(pos++) << (8 * k);
      }
      r.rle_value = v;
    }
    if (r.count == 0) continue;
    vstart += r.count;
    runs.push_back(r);
  }
}
void parse_hybrid_runs(const uint8_t* staged, size_t pos, size_t end, int bw, int32_t max_values, std::vector<PqRun>& runs) {
  parse_hybrid_runs_from([staged](size_t p) { return staged[p]; }, pos, end, bw, max_values, runs);
}

// ---- row-group pruning from column-chunk statistics (the reference pushes data_filters into DataFusion's ParquetSource, which
// prunes row groups by min/max; parquet_exec.rs:60-211).  Only an optimisation: the plan's Filter still runs on what is read.
// A row group is skipped when some conjunct `column <op> literal` cannot be TRUE for any row given [min, max] (and for
// IS NOT NULL when every value is NULL).  Integers, dates, timestamps, INT32/INT64-backed decimals and strings (unsigned bytewise) are handled.
bool stat_i64(const pq::ColumnMeta& cm, const pq::SchemaElement& el, int64_t& mn, int64_t& mx) {
  if (!cm.has_min_max) return false;
  auto rd = [&](const std::string& b, int64_t& v) {
    if (el.type == pq::INT32 && b.size() == 4) { int32_t x; memcpy(&x, b.data(), 4); v = x; return true; }
    if (el.type == pq::INT64 && b.size() == 8) { memcpy(&v, b.data(), 8); return true; }
    return false;
  };
  return rd(cm.min_value, mn) && rd(cm.max_value, mx);
}
bool lit_i64(const Expr& e, int64_t& v) {
  if (e.kind != ExprKind::Literal || e.lit_null) return false;
  switch (e.dtype.id) {
    case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: case TypeId::Int64: case TypeId::Date: case TypeId::Timestamp: case TypeId::TimestampNtz:
      v = e.lit_i64;
      return true;
    case TypeId::Decimal:
      if (e.lit_dec > (i128)INT64_MAX || e.lit_dec < (i128)INT64_MIN) return false;
      v = (int64_t)e.lit_dec;
      return true;
    default: return false;
  }
}
// can `pred` be proven FALSE (or NULL) for every row of the row group?
struct BloomCache;
bool prunes(const Expr& pred, const std::vector<StructField>& schema, const pq::FileMeta& fm, const pq::RowGroup& rg, bool case_sensitive, BloomCache* bloom = nullptr,
            bool* by_bloom = nullptr);

struct ColumnPlan {
  int leaf = -1;          // index into row_group.columns
  pq::SchemaElement el;
  int kind = -1;          // PQ_* conversion, -1 for strings
  int src_width = 0;
  int out_width = 0;      // bytes per output value (0 strings)
  bool is_string = false;
  int dec_scale_up = 0;   // decimal scale widening: unscaled value × 10^dec_scale_up
  bool missing = false;   // the file has no such column: every value is NULL (schema evolution)
  // Dremel levels of the leaf (parquet-format LogicalTypes.md "Nested Types"): a top-level column has max_def 0 / 1 and no repetition.  A
  // struct's field: def_parent = the level from which the struct itself is defined (0: a required struct).  A list's element: def_parent = the
  // level from which the list is defined (not NULL), def_slot = the level from which an entry holds an element slot (below: an empty or NULL list)
  int max_def = 0, max_rep = 0, def_parent = 0, def_slot = 0;
  int def_elem = 0;       // a list of structs: the level from which the element STRUCT is defined (= def_slot when it cannot be NULL)
  bool nested_leaf() const { return max_def > 1 || max_rep > 0; }
};

// The file's schema as a tree (FileMeta.schema is its depth-first flattening): per element its parent, its levels and — for a primitive —
// its ordinal among the leaves, which is the index of its chunk in every row group.
struct SchemaNode { int parent = -1, def = 0, rep = 0, leaf = -1; std::vector<int> kids; };
std::vector<SchemaNode> schema_tree(const pq::FileMeta& fm) {
  std::vector<SchemaNode> t(fm.schema.size());
  int leaf = 0;
  size_t i = 1;
  std::function<void(int)> walk = [&](int parent) {
    const int n = fm.schema[(size_t)parent].num_children;
    for (int k = 0; k < n && i < fm.schema.size(); k++) {
      const int me = (int)i++;
      const pq::SchemaElement& e = fm.schema[(size_t)me];
      t[(size_t)me].parent = parent;
      t[(size_t)me].def = t[(size_t)parent].def + (e.repetition != 0 ? 1 : 0);
      t[(size_t)me].rep = t[(size_t)parent].rep + (e.repetition == 2 ? 1 : 0);
      t[(size_t)parent].kids.push_back(me);
      if (e.num_children > 0) walk(me);
      else t[(size_t)me].leaf = leaf++;
    }
  };
  if (!fm.schema.empty()) walk(0);
  return t;
}

int out_width_of(const DType& t) {
  switch (t.id) {
    case TypeId::Bool: case TypeId::Int8: return 1;
    case TypeId::Int16: return 2;
    case TypeId::Int32: case TypeId::Date: case TypeId::Float: return 4;
    case TypeId::Decimal: return 16;
    case TypeId::String: case TypeId::Bytes: return 0;
    default: return 8;
  }
}

// per-scan options of the schema adapter (parquet/schema_adapter.rs, parquet_support.rs SparkParquetOptions)
struct ScanOptions {
  bool case_sensitive = false;
  bool match_by_id = false;            // use_field_id && the requested schema carries at least one field id
  bool ignore_missing_field_id = false;
  bool allow_type_promotion = false;
  bool allow_timestamp_ltz_to_ntz = false;
  // ship snappy PLAIN pages compressed and decompress them on the GPU (snappy_kernels.hip): 1 yes, 0 no, -1 decide per scan from the bytes
  // such pages hold and the host threads there are to decompress them (scan_parquet)
  int device_snappy_mode = -1;
  bool device_snappy = false;
  bool device_zstd_dict = false;       // COMET_DEVICE_ZSTD_DICT=1: dictionary-encoded zstd pages are inflated by the device too, their index sections come back and the host reads the run
                                       // headers there.  Off: measured on SF10 Q6 (profiles/r3_parquet_q6_zstd_dict.txt) the device inflates these Huffman-only pages at 4–7 GB/s
                                       // (one serial LDS-lookup chain per literal stream) where a host core does 6 GB/s — 87 vs 38 ms with 16 scan threads, 111 vs 97 ms with one
  bool device_runs = true;             // … and the run headers of those index sections are walked ON THE DEVICE (device/pq_runs.hpp: count, prefix sum, write — four bytes come back instead of
                                       // the sections); COMET_DEVICE_RUNS=0: the round-3 path, sections read back and parsed by the scan threads
  bool device_runs_snappy = false;     // snappy: a dictionary-encoded page's run headers are walked on the device too instead of being read THROUGH the compressed stream by the scan thread
                                       // (SnappyView: ≈ 6 µs of a scan thread per page — 9.6 of the 17 ms a one-core task spent preparing its 1500 pages, profiles/r5_executor_shape.md); decided per scan from
                                       // its scan threads (scan_parquet), COMET_DEVICE_RUNS_SNAPPY=0/1 forces it.  Pruned scans keep the host walk (their pages' runs are clipped on the host)
  bool device_zstd = true;             // zstd PLAIN pages of fixed-width columns take the device pipeline too (COMET_DEVICE_ZSTD=0: host threads inflate them)
  bool read_in_place = true;           // chunks whose pages the device inflates are pread() straight into their pinned slot; page bodies are uploaded from where they land (COMET_PARQUET_READ_IN_PLACE=0: read into scratch, copy bodies)
  bool device_dict_pages = true;       // dictionary-encoded snappy pages cross PCIe compressed too (COMET_DEVICE_DICT_PAGES=0: host-inflated as before)
  static ScanOptions of(const Operator& op) {
    ScanOptions o;
    o.case_sensitive = op.case_sensitive;
    bool ids = false;
    for (auto& f : op.required_schema) ids |= f.field_id >= 0;
    o.match_by_id = op.use_field_id && ids;
    o.ignore_missing_field_id = op.ignore_missing_field_id;
    o.allow_type_promotion = op.allow_type_promotion;
    o.allow_timestamp_ltz_to_ntz = op.allow_timestamp_ltz_to_ntz;
    return o;
  }
};

std::string json_escape(const std::string& v) {
  std::string o;
  for (char ch : v) {
    if (ch == '"' || ch == '\\') { o.push_back('\\'); o.push_back(ch); }
    else if ((unsigned char)ch < 0x20) o += ' ';
    else o.push_back(ch);
  }
  return o;
}
// the reference reports these through SparkError → CometQueryExecutionException JSON (native/common/src/error.rs:191-222, 528-560)
CometError spark_error(const std::string& type, const std::string& params_json) {
  return CometError("{\"errorType\":\"" + type + "\",\"params\":{" + params_json + "}}", 1);
}
const char* parquet_primitive_name(int pt) {
  switch (pt) {
    case pq::BOOLEAN: return "BOOLEAN"; case pq::INT32: return "INT32"; case pq::INT64: return "INT64"; case pq::INT96: return "INT96";
    case pq::FLOAT: return "FLOAT"; case pq::DOUBLE: return "DOUBLE"; case pq::BYTE_ARRAY: return "BINARY"; default: return "FIXED_LEN_BYTE_ARRAY";
  }
}
std::string spark_catalog_name(const DType& t) {   // schema_adapter.rs spark_catalog_name
  switch (t.id) {
    case TypeId::Bool: return "boolean"; case TypeId::Int8: return "tinyint"; case TypeId::Int16: return "smallint"; case TypeId::Int32: return "int";
    case TypeId::Int64: return "bigint"; case TypeId::Float: return "float"; case TypeId::Double: return "double"; case TypeId::String: return "string";
    case TypeId::Bytes: return "binary"; case TypeId::Date: return "date"; case TypeId::Timestamp: return "timestamp"; case TypeId::TimestampNtz: return "timestamp_ntz";
    case TypeId::Decimal: return "decimal(" + std::to_string(t.precision) + "," + std::to_string(t.scale) + ")";
    default: return "unknown";
  }
}
CometError schema_convert_error(const std::string& column, int pt, const DType& t) {
  return spark_error("ParquetSchemaConvert", "\"filePath\":\"\",\"column\":\"[" + json_escape(column) + "]\",\"physicalType\":\"" + parquet_primitive_name(pt) +
                                                 "\",\"sparkType\":\"" + spark_catalog_name(t) + "\"");
}

// Which leaf of the file holds the requested column (schema_adapter.rs:76-250 remap_physical_schema, Spark's clipParquetGroupFields):
// a requested field that carries a field id is looked up by id ONLY (no fall-back to its name) when id matching is on; the others by
// name, exact or ASCII-case-insensitive; more than one candidate is an error; none means the column is missing from this file.
ColumnPlan plan_column(const StructField& want, const pq::FileMeta& fm, const ScanOptions& so) {
  ColumnPlan cp;
  bool file_has_ids = false;
  for (size_t i = 1; i < fm.schema.size(); i++) file_has_ids |= fm.schema[i].field_id >= 0;
  if (so.match_by_id && !so.ignore_missing_field_id && !file_has_ids) throw spark_error("ParquetMissingFieldIds", "");
  const std::vector<SchemaNode> tree = schema_tree(fm);
  // the one child of group `g` that answers to (name, id): -1 none; more than one is the reference's duplicate-field error
  auto child_of = [&](int g, const std::string& name, int id) -> int {
    const bool by_id = so.match_by_id && id >= 0;
    std::vector<int> hits;
    for (int k : tree[(size_t)g].kids) {
      const pq::SchemaElement& e = fm.schema[(size_t)k];
      const bool hit = by_id ? e.field_id == id : (so.case_sensitive ? e.name == name : iequals(e.name, name));
      if (hit) hits.push_back(k);
    }
    if (hits.size() > 1) {
      std::string names;
      for (int h : hits) names += (names.empty() ? "" : ", ") + fm.schema[(size_t)h].name;
      if (by_id) throw spark_error("DuplicateFieldByFieldId", "\"requiredId\":" + std::to_string(id) + ",\"matchedFields\":\"" + json_escape(names) + "\"");
      throw spark_error("DuplicateFieldCaseInsensitive", "\"requiredFieldName\":\"" + json_escape(name) + "\",\"matchedOrcFields\":\"[" + json_escape(names) + "]\"");
    }
    return hits.empty() ? -1 : hits[0];
  };
  int at = -1;
  if (want.nest == 0) {
    at = child_of(0, want.name, want.field_id);
    if (at >= 0 && fm.schema[(size_t)at].num_children > 0)
      throw CometError("Parquet column '" + want.name + "' is a group (struct / list / map) in the file but is read as " + want.dtype.str());
  } else {
    const int g = child_of(0, want.parent, want.parent_field_id);
    if (g < 0) throw CometError("Parquet column '" + want.parent + "': a nested column the file does not have is not supported by the GPU scan yet");
    const pq::SchemaElement& ge = fm.schema[(size_t)g];
    if (ge.num_children == 0) throw CometError("Parquet column '" + want.parent + "' is a primitive in the file but is read as a nested column");
    if (ge.repetition == 2) throw CometError("Parquet column '" + want.parent + "': repeated groups outside a LIST annotation are not supported by the GPU scan yet");
    if (want.nest == 3) {
      // a list of structs: <rep> group <name> (LIST) { repeated group list { <rep> group element { fields } } } — or the legacy shape whose
      // repeated group IS the struct (several fields, or one field that is not named like an element wrapper)
      if (tree[(size_t)g].kids.size() != 1) throw CometError("Parquet column '" + want.parent + "': not a LIST group of one repeated field");
      const int rp = tree[(size_t)g].kids[0];
      const pq::SchemaElement& re = fm.schema[(size_t)rp];
      if (re.repetition != 2 || re.num_children == 0) throw CometError("Parquet column '" + want.parent + "': not a list of structs in this file");
      int es = rp;
      if (tree[(size_t)rp].kids.size() == 1) {
        const int only = tree[(size_t)rp].kids[0];
        const pq::SchemaElement& oe = fm.schema[(size_t)only];
        if (oe.num_children > 0 && oe.repetition != 2 && oe.converted_type != 3 && oe.converted_type != 1) es = only;      // the element wrapper
      }
      at = child_of(es, want.name, want.field_id);
      if (at < 0) throw CometError("Parquet column '" + want.parent + "': field '" + want.name + "' of its struct elements is not in the file (not supported by the GPU scan yet)");
      if (fm.schema[(size_t)at].num_children > 0 || fm.schema[(size_t)at].repetition == 2)
        throw CometError("Parquet column '" + want.parent + "." + want.name + "': nesting deeper than a list of flat structs is not supported by the GPU scan yet");
      cp.def_parent = tree[(size_t)g].def;
      cp.def_slot = tree[(size_t)rp].def;
      cp.def_elem = tree[(size_t)es].def;
    } else if (want.nest == 1) {
      if (ge.converted_type == 3 || ge.converted_type == 1 || ge.converted_type == 2)      // LIST, MAP, MAP_KEY_VALUE
        throw CometError("Parquet column '" + want.parent + "' is a list / map in the file but is read as a struct");
      at = child_of(g, want.name, want.field_id);
      if (at < 0) throw CometError("Parquet column '" + want.parent + "': struct field '" + want.name + "' the file does not have is not supported by the GPU scan yet");
      if (fm.schema[(size_t)at].num_children > 0 || fm.schema[(size_t)at].repetition == 2)
        throw CometError("Parquet column '" + want.parent + "." + want.name + "': nesting deeper than one level is not supported by the GPU scan yet");
      cp.def_parent = tree[(size_t)g].def;
    } else {
      // the standard three-level list: <list-repetition> group <name> (LIST) { repeated group list { <element-repetition> <type> element; } }
      // (parquet-format LogicalTypes.md "Lists"; the repeated group and the element may carry any name)
      if (tree[(size_t)g].kids.size() != 1) throw CometError("Parquet column '" + want.parent + "': not a LIST group of one repeated field");
      const int rp = tree[(size_t)g].kids[0];
      const pq::SchemaElement& re = fm.schema[(size_t)rp];
      if (re.repetition != 2) throw CometError("Parquet column '" + want.parent + "': not a LIST group of one repeated field");
      if (re.num_children == 0) at = rp;                                     // legacy two-level list: the repeated field IS the element (required)
      else if (tree[(size_t)rp].kids.size() == 1 && fm.schema[(size_t)tree[(size_t)rp].kids[0]].num_children == 0 && fm.schema[(size_t)tree[(size_t)rp].kids[0]].repetition != 2)
        at = tree[(size_t)rp].kids[0];
      else throw CometError("Parquet column '" + want.parent + "': lists of groups / of lists are not supported by the GPU scan yet");
      cp.def_parent = tree[(size_t)g].def;
      cp.def_slot = tree[(size_t)rp].def;
    }
  }

  if (at >= 0) {
    cp.leaf = tree[(size_t)at].leaf;
    cp.el = fm.schema[(size_t)at];
    cp.max_def = tree[(size_t)at].def;
    cp.max_rep = tree[(size_t)at].rep;
  }
  if (cp.leaf < 0) {
    // a column the file does not have reads as NULL, or as its default value (schema evolution; schema_adapter.rs replace_missing_with_defaults)
    cp.missing = true;
    cp.is_string = want.dtype.id == TypeId::String || want.dtype.id == TypeId::Bytes;
    cp.out_width = out_width_of(want.dtype);
    cp.kind = -2;
    return cp;
  }
  if (cp.el.repetition == 2 && want.nest != 2) throw CometError("repeated Parquet columns are not supported by the GPU scan yet");
  const DType& t = want.dtype;
  const int pt = cp.el.type;
  auto bad = [&]() { return CometError("Parquet column '" + want.name + "': physical type " + std::to_string(pt) + " cannot be read as " + t.str() + " by the GPU scan yet"); };
  // Logical annotations decide what the physical integers mean (the reference reads through the arrow parquet reader, which applies
  // them, then the schema adapter casts to the Spark type): TIMESTAMP(MILLIS) is scaled to microseconds, unsigned integers are
  // zero-extended into the wider Spark type (Spark maps UINT_8 → short, UINT_16 → int, UINT_32 → long, UINT_64 → decimal(20,0));
  // what this scan cannot convert is an error, never a silent reinterpretation.
  const pq::SchemaElement& el = cp.el;
  if (el.is_time) throw CometError("Parquet column '" + want.name + "': TIME columns are not supported by the GPU scan yet");
  if (el.ts_unit == 3) throw CometError("Parquet column '" + want.name + "': TIMESTAMP(NANOS) is not supported by the GPU scan yet");
  const bool is_unsigned = el.int_bits > 0 && !el.int_signed;
  if (is_unsigned) {
    const bool ok = (el.int_bits == 8 && (t.id == TypeId::Int16 || t.id == TypeId::Int32 || t.id == TypeId::Int64)) ||
                    (el.int_bits == 16 && (t.id == TypeId::Int32 || t.id == TypeId::Int64)) ||
                    (el.int_bits == 32 && t.id == TypeId::Int64) ||
                    (el.int_bits == 64 && t.id == TypeId::Decimal && t.precision - t.scale >= 20);
    if (!ok) throw CometError("Parquet column '" + want.name + "': UINT_" + std::to_string(el.int_bits) + " cannot be read as " + t.str() + " by the GPU scan");
  }
  // a TimestampLTZ column (isAdjustedToUTC, or INT96) read as TimestampNTZ: Spark 3.x rejects it (SPARK-36182), 4.0+ allows it
  if (t.id == TypeId::TimestampNtz && !so.allow_timestamp_ltz_to_ntz && ((el.ts_unit != 0 && el.ts_utc) || pt == pq::INT96))
    throw schema_convert_error(want.name, pt, t);
  if (el.ts_unit != 0 && !(t.id == TypeId::Timestamp || t.id == TypeId::TimestampNtz || t.id == TypeId::Int64))
    throw CometError("Parquet column '" + want.name + "': a TIMESTAMP column cannot be read as " + t.str());
  switch (t.id) {
    case TypeId::Int32: case TypeId::Date: if (pt != pq::INT32) throw bad(); cp.kind = PQ_COPY4; cp.src_width = 4; cp.out_width = 4; break;
    case TypeId::Int16: if (pt != pq::INT32) throw bad(); cp.kind = PQ_I32_TO_I16; cp.src_width = 4; cp.out_width = 2; break;
    case TypeId::Int8: if (pt != pq::INT32) throw bad(); cp.kind = PQ_I32_TO_I8; cp.src_width = 4; cp.out_width = 1; break;
    case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz:
      if (pt == pq::INT64) { cp.kind = (el.ts_unit == 1 && t.id != TypeId::Int64) ? PQ_I64_MILLIS_TO_MICROS : PQ_COPY8; cp.src_width = 8; }
      else if (pt == pq::INT32 && t.id == TypeId::Int64) {   // type promotion (schema_adapter.rs:749-771): rejected on Spark 3.x
        if (!so.allow_type_promotion && !is_unsigned) throw schema_convert_error(want.name, pt, t);
        cp.kind = (is_unsigned && el.int_bits == 32) ? PQ_U32_TO_I64 : PQ_I32_TO_I64; cp.src_width = 4;
      }
      else if (pt == pq::INT96 && t.id != TypeId::Int64) { cp.kind = PQ_INT96_TO_TS_MICROS; cp.src_width = 12; }   // legacy Spark/Impala timestamps
      else throw bad();
      cp.out_width = 8;
      break;
    case TypeId::Float: if (pt != pq::FLOAT) throw bad(); cp.kind = PQ_COPY4; cp.src_width = 4; cp.out_width = 4; break;
    case TypeId::Double:
      if (pt == pq::DOUBLE) { cp.kind = PQ_COPY8; cp.src_width = 8; }
      else if (pt == pq::FLOAT || pt == pq::INT32) {                                // FLOAT → DOUBLE, INT32 → DOUBLE promotions
        if (!so.allow_type_promotion) throw schema_convert_error(want.name, pt, t);
        cp.kind = pt == pq::FLOAT ? PQ_F32_TO_F64 : PQ_I32_TO_F64; cp.src_width = 4;
      }
      else throw bad();
      cp.out_width = 8;
      break;
    case TypeId::Bool: if (pt != pq::BOOLEAN) throw bad(); cp.kind = PQ_BOOL; cp.src_width = 0; cp.out_width = 1; break;
    case TypeId::Decimal:
      if (pt == pq::INT32) { cp.kind = PQ_I32_TO_DEC; cp.src_width = 4; }
      else if (pt == pq::INT64 && is_unsigned) { cp.kind = PQ_U64_TO_DEC; cp.src_width = 8; cp.dec_scale_up = t.scale; cp.out_width = 16; return cp; }   // UINT_64 → decimal(20,0)
      else if (pt == pq::INT64) { cp.kind = PQ_I64_TO_DEC; cp.src_width = 8; }
      else if (pt == pq::FLBA && cp.el.type_length >= 1 && cp.el.type_length <= 16) { cp.kind = PQ_FLBA_TO_DEC; cp.src_width = cp.el.type_length; }
      else throw bad();
      // decimal widening (parquet_support.rs): a larger scale multiplies the unscaled value, provided the integer digits still fit
      if (cp.el.scale > t.scale || (t.precision - t.scale) < (cp.el.precision - cp.el.scale))
        throw CometError("Parquet column '" + want.name + "': decimal(" + std::to_string(cp.el.precision) + "," + std::to_string(cp.el.scale) + ") cannot be read as " + t.str() + " without losing digits");
      cp.dec_scale_up = t.scale - cp.el.scale;
      cp.out_width = 16;
      break;
    case TypeId::String: case TypeId::Bytes: if (pt != pq::BYTE_ARRAY) throw bad(); cp.is_string = true; break;
    default: throw bad();
  }
  return cp;
}

// Statistics of a column chunk or of one of its pages, seen the same way (PLAIN-encoded min / max)
struct StatView {
  bool has_min_max = false;
  const std::string* mn = nullptr;
  const std::string* mx = nullptr;
  int64_t null_count = -1, num_values = 0;
  bool all_null = false;
  bool typed_order = true;      // min / max follow the column's order (a ColumnIndex always; a chunk's Statistics when they are the min_value / max_value pair)
};
bool stat_i64(const StatView& sv, const pq::SchemaElement& el, int64_t& mn, int64_t& mx) {
  if (!sv.has_min_max || !sv.mn || !sv.mx) return false;
  auto rd = [&](const std::string& b, int64_t& v) {
    if (el.type == pq::INT32 && b.size() == 4) { int32_t x; memcpy(&x, b.data(), 4); v = x; return true; }
    if (el.type == pq::INT64 && b.size() == 8) { memcpy(&v, b.data(), 8); return true; }
    return false;
  };
  return rd(*sv.mn, mn) && rd(*sv.mx, mx);
}
// the file column a bound reference of a pushed-down filter names: its leaf index, schema element and the type the scan reads it as
struct LeafCol { int leaf = -1; const pq::SchemaElement* el = nullptr; DType want; };
bool leaf_column(const Expr& b, const std::vector<StructField>& schema, const pq::FileMeta& fm, size_t ncols_in_rg, bool case_sensitive, LeafCol& out) {
  if (b.kind != ExprKind::Bound || b.bound_index < 0 || (size_t)b.bound_index >= schema.size()) return false;
  const StructField& f = schema[(size_t)b.bound_index];
  int leaf = 0;
  for (size_t i = 1; i < fm.schema.size(); i++) {
    const pq::SchemaElement& e = fm.schema[i];
    if (e.num_children > 0) return false;
    if ((case_sensitive && e.name == f.name) || (!case_sensitive && iequals(e.name, f.name))) {
      if ((size_t)leaf >= ncols_in_rg) return false;
      out.leaf = leaf;
      out.el = &e;
      out.want = f.dtype;
      return true;
    }
    leaf++;
  }
  return false;
}
// A leaf predicate (IsNotNull(col), col <cmp> literal) split into its parts; false = not a shape statistics can decide
struct LeafPred { ExprKind kind; const Expr* col = nullptr; const Expr* lit = nullptr; };
bool leaf_pred(const Expr& pred, LeafPred& lp) {
  if (pred.kind == ExprKind::IsNotNull && pred.children.size() == 1) {
    lp.kind = ExprKind::IsNotNull;
    lp.col = pred.children[0].get();
    return true;
  }
  const bool cmp = pred.kind == ExprKind::Eq || pred.kind == ExprKind::Lt || pred.kind == ExprKind::LtEq || pred.kind == ExprKind::Gt || pred.kind == ExprKind::GtEq;
  if (!cmp || pred.children.size() != 2) return false;
  const Expr *l = pred.children[0].get(), *r = pred.children[1].get();
  ExprKind k = pred.kind;
  if (l->kind == ExprKind::Literal) {   // literal <op> column  →  column <flipped op> literal
    std::swap(l, r);
    k = k == ExprKind::Lt ? ExprKind::Gt : k == ExprKind::LtEq ? ExprKind::GtEq : k == ExprKind::Gt ? ExprKind::Lt : k == ExprKind::GtEq ? ExprKind::LtEq : k;
  }
  lp.kind = k;
  lp.col = l;
  lp.lit = r;
  return true;
}
// do these statistics prove the leaf predicate FALSE (or NULL) for every row they cover?
bool stats_prove_false(const LeafPred& lp, const LeafCol& c, const StatView& sv) {
  if (lp.kind == ExprKind::IsNotNull) return sv.all_null || (sv.null_count >= 0 && sv.null_count == sv.num_values && sv.num_values > 0);
  if (sv.all_null) return true;       // a comparison with NULL is never true
  int64_t v, mn, mx;
  const pq::SchemaElement* el = c.el;
  const DType& t = c.want;
  if (el->type == pq::BYTE_ARRAY) {
    // strings: unsigned bytewise order — Spark's (UTF8String.compareTo) and the column order of BYTE_ARRAY statistics.  A writer may shorten them
    // (a prefix for min, an incremented prefix for max): still a lower and an upper bound, which is all these tests use.
    if (!sv.typed_order || !sv.has_min_max || !sv.mn || !sv.mx || t.id != TypeId::String || lp.lit->kind != ExprKind::Literal || lp.lit->lit_null ||
        lp.lit->dtype.id != TypeId::String)
      return false;
    const std::string& s = lp.lit->lit_bytes;
    const int lo = s.compare(*sv.mn) < 0 ? -1 : (s == *sv.mn ? 0 : 1);       // (std::string compares as unsigned char)
    const int hi = s.compare(*sv.mx) < 0 ? -1 : (s == *sv.mx ? 0 : 1);
    switch (lp.kind) {
      case ExprKind::Eq: return lo < 0 || hi > 0;
      case ExprKind::Lt: return lo <= 0;      // min >= literal
      case ExprKind::LtEq: return lo < 0;
      case ExprKind::Gt: return hi >= 0;      // max <= literal
      case ExprKind::GtEq: return hi > 0;
      default: return false;
    }
  }
  if (!lit_i64(*lp.lit, v) || !stat_i64(sv, *el, mn, mx)) return false;
  // statistics are in the file's unit / signedness: do not compare them with a microsecond or signed literal
  if ((el->ts_unit != 0 && el->ts_unit != 2) || (el->int_bits > 0 && !el->int_signed) || el->type == pq::INT96) return false;
  if (t.id == TypeId::Decimal && (el->scale != t.scale || !(lp.lit->dtype.id == TypeId::Decimal && lp.lit->dtype.scale == t.scale))) return false;   // same scale only
  if (t.id != TypeId::Decimal && lp.lit->dtype.id == TypeId::Decimal) return false;
  switch (lp.kind) {
    case ExprKind::Eq: return v < mn || v > mx;
    case ExprKind::Lt: return mn >= v;
    case ExprKind::LtEq: return mn > v;
    case ExprKind::Gt: return mx <= v;
    case ExprKind::GtEq: return mx < v;
    default: return false;
  }
}
StatView chunk_stats(const pq::ColumnMeta& cm) {
  StatView sv;
  sv.has_min_max = cm.has_min_max;
  sv.mn = &cm.min_value;
  sv.mx = &cm.max_value;
  sv.null_count = cm.null_count;
  sv.num_values = cm.num_values;
  sv.typed_order = cm.stats_typed_order;
  return sv;
}

// ---- row-group pruning from Bloom filters (DataFusion's ParquetSource probes a chunk's filter for `column = literal` and `column IN (literals)` when
// datafusion.execution.parquet.bloom_filter_on_read is on — the default, carried through by parquet_exec.rs:251-252).  A filter answers "certainly absent" or
// "maybe there"; like the statistics, only an optimisation.  The literal is hashed in the chunk's PLAIN encoding, so only pairs of file type and literal whose
// encoding is beyond doubt are probed: signed INT32 / INT64 integers, dates, microsecond timestamps and decimals of the file's scale, BYTE_ARRAY strings,
// FIXED_LEN_BYTE_ARRAY decimals of the file's scale.  Floats (NaN, -0.0), unsigned and INT96 columns, rebased dates (before 1582-10-15) are left alone.
struct BloomCache {
  const OpenFile* file;
  const pq::RowGroup* rg;
  std::map<int, std::shared_ptr<std::vector<uint8_t>>> by_leaf;      // nullptr = the chunk has no (usable) filter
  const std::vector<uint8_t>* get(int leaf) {
    auto it = by_leaf.find(leaf);
    if (it != by_leaf.end()) return it->second.get();
    std::shared_ptr<std::vector<uint8_t>> bits;
    const pq::ColumnMeta& cm = rg->columns[(size_t)leaf];
    if (cm.bloom_filter_offset > 0 && (size_t)cm.bloom_filter_offset + 40 <= file->size) {
      try {
        uint8_t head[64];
        const size_t avail = std::min<size_t>(sizeof head, file->size - (size_t)cm.bloom_filter_offset);
        file->read_at(head, avail, cm.bloom_filter_offset);
        int32_t nbytes = 0;
        const size_t hlen = pq::parse_bloom_header(head, avail, nbytes);
        // (128 MB: the format's upper bound of a filter)
        if (nbytes <= (128 << 20) && (size_t)cm.bloom_filter_offset + hlen + (size_t)nbytes <= file->size &&
            (cm.bloom_filter_length <= 0 || (size_t)cm.bloom_filter_length == hlen + (size_t)nbytes)) {
          bits = std::make_shared<std::vector<uint8_t>>((size_t)nbytes);
          file->read_at(bits->data(), bits->size(), cm.bloom_filter_offset + (int64_t)hlen);
        }
      } catch (const CometError&) {
        bits.reset();      // a filter this reader cannot make sense of prunes nothing
      }
    }
    by_leaf.emplace(leaf, bits);
    return bits.get();
  }
};
// the PLAIN encoding of `lit` as a value of the file column (false: not a pair this reader hashes)
bool plain_bytes_of(const Expr& lit, const LeafCol& c, std::string& out) {
  if (lit.kind != ExprKind::Literal || lit.lit_null) return false;
  const pq::SchemaElement& el = *c.el;
  const DType& t = c.want;
  if ((el.int_bits > 0 && !el.int_signed) || el.type == pq::INT96) return false;
  if (el.type == pq::BYTE_ARRAY) {
    if (!((t.id == TypeId::String && lit.dtype.id == TypeId::String) || (t.id == TypeId::Bytes && lit.dtype.id == TypeId::Bytes))) return false;
    out = lit.lit_bytes;
    return true;
  }
  const bool dec = t.id == TypeId::Decimal;
  if (dec != (lit.dtype.id == TypeId::Decimal)) return false;
  if (dec && (el.scale != t.scale || lit.dtype.scale != t.scale)) return false;
  if (el.type == pq::FLBA) {
    if (!dec || el.type_length < 1 || el.type_length > 16) return false;
    const int bits = 8 * el.type_length;
    if (bits < 128 && (lit.lit_dec >= ((i128)1 << (bits - 1)) || lit.lit_dec < -((i128)1 << (bits - 1)))) return false;
    out.assign((size_t)el.type_length, '\0');
    for (int b = 0; b < el.type_length; b++) out[(size_t)(el.type_length - 1 - b)] = (char)(uint8_t)((u128)lit.lit_dec >> (8 * b));
    return true;
  }
  int64_t v;
  if (!lit_i64(lit, v)) return false;
  switch (t.id) {
    case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: case TypeId::Int64: case TypeId::Decimal: break;
    case TypeId::Date: if (lit.dtype.id != TypeId::Date || v < -141427) return false; break;
    case TypeId::Timestamp: case TypeId::TimestampNtz:
      if ((lit.dtype.id != TypeId::Timestamp && lit.dtype.id != TypeId::TimestampNtz) || el.ts_unit != 2 || v < -141427ll * 86400000000ll) return false;
      break;
    default: return false;
  }
  if (t.id != TypeId::Timestamp && t.id != TypeId::TimestampNtz && el.ts_unit != 0) return false;
  if ((t.id == TypeId::Date) != (lit.dtype.id == TypeId::Date)) return false;
  if (el.type == pq::INT32) {
    if (v < INT32_MIN || v > INT32_MAX) return false;
    const int32_t x = (int32_t)v;
    out.assign((const char*)&x, 4);
    return true;
  }
  if (el.type == pq::INT64) { out.assign((const char*)&v, 8); return true; }
  return false;
}
// does the chunk's Bloom filter prove that NONE of the literals is a value of the column?
bool bloom_proves_absent(const LeafCol& c, const Expr* const* lits, size_t n, BloomCache& bc) {
  if (n == 0) return false;
  const std::vector<uint8_t>* bits = nullptr;
  std::string enc;
  for (size_t i = 0; i < n; i++) {
    if (lits[i]->kind == ExprKind::Literal && lits[i]->lit_null) continue;      // a NULL in the list matches nothing
    if (!plain_bytes_of(*lits[i], c, enc)) return false;
    if (!bits && !(bits = bc.get(c.leaf))) return false;
    if (pq::sbbf_might_contain(bits->data(), bits->size(), pq::xxh64(enc.data(), enc.size(), 0))) return false;
  }
  return bits != nullptr;
}

bool prunes(const Expr& pred, const std::vector<StructField>& schema, const pq::FileMeta& fm, const pq::RowGroup& rg, bool case_sensitive, BloomCache* bloom, bool* by_bloom) {
  if (pred.kind == ExprKind::And) {
    for (auto& c : pred.children)
      if (prunes(*c, schema, fm, rg, case_sensitive, bloom, by_bloom)) return true;
    return false;
  }
  if (pred.kind == ExprKind::Or) {
    bool any_bloom = false;
    for (auto& c : pred.children) {
      bool b = false;
      if (!prunes(*c, schema, fm, rg, case_sensitive, bloom, &b)) return false;
      any_bloom = any_bloom || b;
    }
    if (by_bloom && any_bloom) *by_bloom = true;
    return !pred.children.empty();
  }
  LeafCol col;
  if (pred.kind == ExprKind::In && !pred.negated && pred.children.size() >= 2) {
    // column IN (literals): every literal must be ruled out — by the chunk's min / max, and what they leave by its Bloom filter
    if (!leaf_column(*pred.children[0], schema, fm, rg.columns.size(), case_sensitive, col)) return false;
    const StatView sv = chunk_stats(rg.columns[(size_t)col.leaf]);
    std::vector<const Expr*> left;
    for (size_t i = 1; i < pred.children.size(); i++) {
      const Expr* l = pred.children[i].get();
      if (l->kind != ExprKind::Literal) return false;
      if (l->lit_null) continue;      // a NULL in the list matches nothing
      const LeafPred eq{ExprKind::Eq, pred.children[0].get(), l};
      if (!stats_prove_false(eq, col, sv)) left.push_back(l);
    }
    if (left.empty()) return true;
    if (!bloom || !bloom_proves_absent(col, left.data(), left.size(), *bloom)) return false;
    if (by_bloom) *by_bloom = true;
    return true;
  }
  LeafPred lp;
  if (!leaf_pred(pred, lp) || !leaf_column(*lp.col, schema, fm, rg.columns.size(), case_sensitive, col)) return false;
  if (stats_prove_false(lp, col, chunk_stats(rg.columns[(size_t)col.leaf]))) return true;
  if (bloom && lp.kind == ExprKind::Eq && bloom_proves_absent(col, &lp.lit, 1, *bloom)) {
    if (by_bloom) *by_bloom = true;
    return true;
  }
  return false;
}

// ---- page-index pruning: which ROWS of a row group can a pushed-down filter still be true for? ---------------------------------------
// Sorted, disjoint [begin, end) row ranges (row-group relative).
typedef std::vector<std::pair<int64_t, int64_t>> Ranges;
Ranges ranges_and(const Ranges& a, const Ranges& b) {
  Ranges o;
  size_t i = 0, j = 0;
  while (i < a.size() && j < b.size()) {
    const int64_t lo = std::max(a[i].first, b[j].first), hi = std::min(a[i].second, b[j].second);
    if (lo < hi) o.emplace_back(lo, hi);
    if (a[i].second < b[j].second) i++;
    else j++;
  }
  return o;
}
Ranges ranges_or(const Ranges& a, const Ranges& b) {
  Ranges all(a);
  all.insert(all.end(), b.begin(), b.end());
  std::sort(all.begin(), all.end());
  Ranges o;
  for (auto& r : all) {
    if (!o.empty() && r.first <= o.back().second) o.back().second = std::max(o.back().second, r.second);
    else o.push_back(r);
  }
  return o;
}
int64_t ranges_rows(const Ranges& r) {
  int64_t n = 0;
  for (auto& x : r) n += x.second - x.first;
  return n;
}
// loads (once per column chunk) and caches the page index of the chunks a filter refers to
struct PageIndexCache {
  const OpenFile* file;
  const pq::RowGroup* rg;
  std::map<int, std::shared_ptr<pq::PageIndex>> by_leaf;   // nullptr = the chunk has no (usable) page index
  const pq::PageIndex* get(int leaf) {
    auto it = by_leaf.find(leaf);
    if (it != by_leaf.end()) return it->second.get();
    std::shared_ptr<pq::PageIndex> pi;
    const pq::ColumnMeta& cm = rg->columns[(size_t)leaf];
    if (cm.column_index_offset > 0 && cm.column_index_length > 0 && cm.offset_index_offset > 0 && cm.offset_index_length > 0 &&
        (size_t)(cm.column_index_offset + cm.column_index_length) <= file->size && (size_t)(cm.offset_index_offset + cm.offset_index_length) <= file->size) {
      try {
        std::vector<uint8_t> ci((size_t)cm.column_index_length), oi((size_t)cm.offset_index_length);
        file->read_at(ci.data(), ci.size(), cm.column_index_offset);
        file->read_at(oi.data(), oi.size(), cm.offset_index_offset);
        pi = std::make_shared<pq::PageIndex>(pq::parse_page_index(ci.data(), ci.size(), oi.data(), oi.size()));
        if (pi->first_row.empty() || pi->first_row[0] != 0) pi.reset();
      } catch (const CometError&) {
        pi.reset();      // an index this reader cannot make sense of prunes nothing
      }
    }
    by_leaf[leaf] = pi;
    return pi.get();
  }
};
// rows of the row group for which `pred` may still be true, by the page statistics (ColumnIndex) of the columns it refers to
Ranges may_match(const Expr& pred, const std::vector<StructField>& schema, const pq::FileMeta& fm, const pq::RowGroup& rg, bool case_sensitive, PageIndexCache& pic) {
  const Ranges all{{0, rg.num_rows}};
  if (pred.kind == ExprKind::And) {
    Ranges r = all;
    for (auto& c : pred.children) r = ranges_and(r, may_match(*c, schema, fm, rg, case_sensitive, pic));
    return r;
  }
  if (pred.kind == ExprKind::Or) {
    if (pred.children.empty()) return all;
    Ranges r;
    for (auto& c : pred.children) r = ranges_or(r, may_match(*c, schema, fm, rg, case_sensitive, pic));
    return r;
  }
  LeafPred lp;
  LeafCol col;
  if (!leaf_pred(pred, lp) || !leaf_column(*lp.col, schema, fm, rg.columns.size(), case_sensitive, col)) return all;
  const pq::PageIndex* pi = pic.get(col.leaf);
  if (!pi) return all;
  Ranges r;
  const size_t np = pi->first_row.size();
  for (size_t k = 0; k < np; k++) {
    const int64_t lo = pi->first_row[k], hi = k + 1 < np ? pi->first_row[k + 1] : rg.num_rows;
    if (hi <= lo) continue;
    StatView sv;
    sv.all_null = pi->null_page[k] != 0;
    sv.has_min_max = !sv.all_null;
    sv.mn = &pi->min_value[k];
    sv.mx = &pi->max_value[k];
    sv.num_values = hi - lo;
    sv.null_count = pi->null_count.empty() ? -1 : pi->null_count[k];
    if (stats_prove_false(lp, col, sv)) continue;
    if (!r.empty() && r.back().second == lo) r.back().second = hi;
    else r.emplace_back(lo, hi);
  }
  return r;
}

// COMET_TRACE_STAGES: where the scan threads' time goes (summed over threads, printed by read_columns)
static std::atomic<int64_t> g_ns_read{0}, g_ns_walk{0}, g_ns_inflate{0}, g_ns_chunk{0};
static const bool g_host_timers = getenv("COMET_TRACE_STAGES") != nullptr;
struct HostTimer {
  std::atomic<int64_t>* to;
  std::chrono::steady_clock::time_point t0;
  explicit HostTimer(std::atomic<int64_t>& a) : to(g_host_timers ? &a : nullptr) { if (to) t0 = std::chrono::steady_clock::now(); }
  ~HostTimer() { if (to) to->fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count()); }
};

struct HostChunk {
  ColumnPlan cp;
  int max_def = 0;
  bool no_nulls = true;            // every definition level of the chunk equals max_def (or the column is required)
  int64_t n_rows = 0;
  int64_t compressed = 0;
  size_t spos = 0;                 // staged (uploaded) bytes actually used
  struct Pending { size_t page; size_t begin, end; int32_t values; };   // a dictionary-encoded page the device inflates: its run headers (bytes [begin, end) of the
                                                                      // device-decompressed region, slot relative) are read back and parsed once the device has inflated it
  std::vector<Pending> pending;
  int64_t dev_dict = -1;           // the chunk's dictionary page is inflated by the device: its offset in the device-decompressed region (else −1: dict_bytes holds it)
  size_t raw_lo = 0, raw_hi = 0;   // slot-relative extent of the page bodies that were read in place (behind the staged bytes) and cross PCIe from there
  size_t ipos = 0;                 // bytes of the device-decompressed region used
  int64_t pages_skipped = 0;       // data pages the page index ruled out
  std::vector<PqInflate> inflate;  // page bodies the device decompresses (offsets relative to the chunk's slot in either region)
  std::vector<PqInflate> zinflate; // … zstd ones: `preamble` = the page's first block in zblocks, `pad` = its block count
  std::vector<comet_zstd2::ZBlock> zblocks;   // what the host walk over those pages' frames found (device/zstd2.hpp)
  std::vector<PqPage> pages;
  std::vector<PqRun> def_runs, idx_runs, rep_runs;
  std::vector<uint8_t> dict_bytes;
  std::vector<int32_t> dict_offs;
  std::vector<int64_t> str_offs;
  std::exception_ptr err;
};

struct ChunkSource {
  const OpenFile* file;
  const pq::FileMeta* meta;
  int rg;
  const Ranges* keep;      // rows of the row group the page index could not rule out (nullptr: all of them)
};

// bytes the staged (decompressed) pages of a column chunk may take
// (device-decompressed pages start on 16-byte boundaries: at most one per 4 KiB of page data, hence the 1/256)
// A chunk with DELTA_* pages is rewritten as PLAIN on the host and outgrows its declared size: at most 8 bytes per value more.
// where a chunk starts in its file (its dictionary page when it has one)
int64_t chunk_file_offset(const pq::ColumnMeta& cm) {
  return (cm.dictionary_page_offset > 0 && cm.dictionary_page_offset < cm.data_page_offset) ? cm.dictionary_page_offset : cm.data_page_offset;
}
size_t staged_capacity(const pq::ColumnMeta& cm) {
  const size_t delta = cm.delta_encoded ? (size_t)std::max<int64_t>(cm.num_values, 0) * 8 + 64 : 0;
  return ((size_t)cm.total_uncompressed + (size_t)cm.total_uncompressed / 256 + delta + 128 + 15) & ~(size_t)15;
}
// offsets into the device-decompressed region carry this bit until the column's tables are assembled
constexpr int64_t kInflatedBit = (int64_t)1 << 62;
constexpr int32_t kMinDevicePage = 4096;
constexpr double kDeviceZstdBytesPerMs = 75.0e6;    // measured: 480 pages of 1 MiB (decimal-as-INT64, level 1) through the zstd pipeline in 6.5 ms (profiles/r4_zstd_kernel_stats.txt)
// … and what a scan thread does when it inflates such pages itself — SF10 Q6 from zstd Parquet, round 4 (profiles/r4_parquet_q6.txt): 334 ms of thread time for the
// 495 MB of PLAIN pages on the GPU box's 32 scan threads (1.5 GB/s each; a lone thread reaches 2.2) and the pages then cross PCIe inflated: 27.9 ms for the scan
// against 19.9–21.7 ms with the pages inflated on the device, whose pipeline (two chains of the sequence kernel side by side, ≈ 8 ms) runs under the uploads.
// The device path wins up to about forty scan threads.
constexpr double kHostZstdBytesPerMs = 1.5e6, kDeviceZstdSetupMs = 2.0;

// DELTA_BYTE_ARRAY pages are prefix-compressed: what they decode to is only known from their length blocks.  One extra pass over such a
// chunk (read, decompress, decode the two length blocks of every page) sizes its staging slot; nothing else pays for it.
size_t prefix_encoded_plain_bytes(const ChunkSource& src, const pq::ColumnMeta& cm, int max_def) {
  int64_t off = chunk_file_offset(cm);
  const int64_t chunk_end = off + cm.total_compressed;
  if (off < 0 || (size_t)chunk_end > src.file->size) throw CometError("parquet: column chunk outside the file");
  std::vector<uint8_t> raw((size_t)cm.total_compressed + 16), page;
  src.file->read_at(raw.data(), (size_t)cm.total_compressed, off);
  const uint8_t* chunk_data = raw.data() - off;
  size_t total = 0;
  int64_t values_seen = 0;
  while (values_seen < cm.num_values && off < chunk_end) {
    pq::PageHeader h = pq::parse_page_header(chunk_data + off, (size_t)(chunk_end - off));
    const uint8_t* body = chunk_data + off + h.header_len;
    if (h.compressed_size < 0 || h.uncompressed_size < 0 || (int64_t)h.header_len + h.compressed_size > chunk_end - off)
      throw CometError("parquet: page of " + std::to_string(h.compressed_size) + " bytes runs past its column chunk");
    off += (int64_t)h.header_len + h.compressed_size;
    if (h.type != pq::DATA_PAGE && h.type != pq::DATA_PAGE_V2) continue;
    values_seen += h.num_values;
    if (h.encoding != pq::DELTA_BYTE_ARRAY) continue;
    if (h.uncompressed_size < 0 || h.compressed_size < 0 || h.uncompressed_size > (1 << 30)) throw CometError("parquet: implausible page size");
    page.resize((size_t)h.uncompressed_size + 16);
    size_t vbegin = 0, vend = 0;
    if (h.type == pq::DATA_PAGE) {
      pq::decompress(cm.codec, body, (size_t)h.compressed_size, page.data(), (size_t)h.uncompressed_size);
      vend = (size_t)h.uncompressed_size;
      if (max_def > 0) {
        if (vend < 4) throw CometError("parquet: truncated data page");
        uint32_t dl;
        memcpy(&dl, page.data(), 4);
        vbegin = 4 + (size_t)dl;
      }
    } else {
      if (h.def_bytes < 0 || h.rep_bytes < 0 || h.def_bytes + h.rep_bytes > h.compressed_size) throw CometError("parquet: v2 page levels longer than the page");
      const size_t lv = (size_t)h.def_bytes + (size_t)h.rep_bytes;
      pq::decompress(h.v2_compressed ? cm.codec : pq::UNCOMPRESSED, body + lv, (size_t)h.compressed_size - lv, page.data(), (size_t)h.uncompressed_size - lv);
      vend = (size_t)h.uncompressed_size - lv;
    }
    if (vbegin > vend) throw CometError("parquet: definition levels longer than their page");
    total += pq::delta_byte_array_plain_size(page.data() + vbegin, vend - vbegin, h.num_values) + 16;
  }
  return total;
}
// staging slot of one column chunk: its declared size, what DELTA pages add, and for prefix-compressed strings what they measure to
// A chunk that is read in place gets the file's bytes BEHIND what the host may stage: [0, front) as before, [front, front + raw) the chunk as
// it sits in the file.  Page bodies the device inflates cross PCIe from there — the second pass over them (scratch → staging) was half of
// the host's work per scan once the device did the decompression.
size_t in_place_extra(const pq::ColumnMeta& cm) { return ((size_t)std::max<int64_t>(cm.total_compressed, 0) + 64 + 15) & ~(size_t)15; }
bool in_place_shape(const pq::ColumnMeta& cm, bool is_string, const ScanOptions& so) {
  return so.read_in_place && (cm.codec == pq::SNAPPY || (cm.codec == pq::ZSTD && so.device_zstd)) && !is_string && !cm.prefix_encoded;
}
size_t chunk_staging_capacity(const ChunkSource& src, const pq::ColumnMeta& cm, int max_def, bool is_string, const ScanOptions& so) {
  (void)is_string; (void)so;
  return staged_capacity(cm) + (cm.prefix_encoded ? ((prefix_encoded_plain_bytes(src, cm, max_def) + 15) & ~(size_t)15) : 0);
}

// Host half of one column chunk (one column of one row group): page walk, decompression straight into the column's pinned
// staging block at the chunk's slot, hybrid-run tables.  Independent of every other chunk, so chunks are prepared by a
// pool of host threads (scan_parquet); the calling thread then concatenates a column's tables and decodes the WHOLE column
// (all row groups) with one upload and one launch per kernel.
// `raw_area` (optional, above `staged` in the same pinned block, `raw_cap` bytes): where a chunk that is read in place keeps the file's bytes —
// the raw areas of a column's chunks lie NEXT TO EACH OTHER behind the column's staging slots, so what crosses PCIe compressed is contiguous
// and a column's ready chunks cross in a handful of copies.  `raw_read`: the scan threads have already read the chunk into it (in pieces).
void decode_chunk_host(const ChunkSource& src, const StructField& want, const ScanOptions& so, HostChunk& hc, uint8_t* staged, size_t slot_cap,
                       uint8_t* raw_area = nullptr, size_t raw_cap = 0, bool raw_read = false) {
  const pq::RowGroup& rg = src.meta->row_groups[(size_t)src.rg];
  ColumnPlan cp = plan_column(want, *src.meta, so);
  if (cp.missing) throw CometError("internal: chunk task for a missing column");
  if ((size_t)cp.leaf >= rg.columns.size()) throw CometError("parquet: column index out of range");
  const pq::ColumnMeta& cm = rg.columns[(size_t)cp.leaf];
  hc.cp = cp;
  const int max_def = cp.max_def;
  // (a nested leaf: its levels take more than one bit, a list's element has repetition levels in front of them and more entries than the row
  // group has rows; its pages are inflated on the host and its levels parsed there — the device paths below read one-bit levels)
  const bool nested_leaf = cp.nested_leaf();
  const int def_bw = max_def <= 1 ? 1 : 32 - __builtin_clz((unsigned)max_def);
  const int rep_bw = cp.max_rep <= 1 ? 1 : 32 - __builtin_clz((unsigned)cp.max_rep);
  if (nested_leaf && src.keep) throw CometError("internal: page-index pruning over a nested column");
  hc.max_def = max_def;
  hc.n_rows = src.keep ? ranges_rows(*src.keep) : cp.max_rep > 0 ? cm.num_values : rg.num_rows;
  hc.compressed = cm.total_compressed;
  size_t spos = 0;
  std::vector<PqPage>& pages = hc.pages;
  std::vector<PqRun>&def_runs = hc.def_runs, &idx_runs = hc.idx_runs;
  std::vector<uint8_t>& dict_bytes = hc.dict_bytes;
  std::vector<int32_t>& dict_offs = hc.dict_offs;
  std::vector<int64_t>& str_offs = hc.str_offs;

  int64_t off = chunk_file_offset(cm);
  const int64_t chunk_end = off + cm.total_compressed;
  if (off < 0 || (size_t)chunk_end > src.file->size) throw CometError("parquet: column chunk outside the file");
  // the whole (compressed) chunk into the back of its slot when the device will inflate its pages (their bodies are uploaded from where they
  // land), else into this thread's scratch; then parse from memory
  static thread_local std::vector<uint8_t> raw;
  const size_t staged_cap = slot_cap;
  const bool in_place = so.device_snappy && !nested_leaf && in_place_shape(cm, cp.is_string, so) && raw_area != nullptr && raw_area > staged && raw_cap >= in_place_extra(cm);
  uint8_t* rawp;
  if (in_place) {
    rawp = raw_area;
  } else {
    if (raw_read) throw CometError("internal: chunk read in pieces but not decoded in place");
    if (raw.size() < (size_t)cm.total_compressed + 16) raw.resize((size_t)cm.total_compressed + 16);
    rawp = raw.data();
  }
  if (!raw_read) { HostTimer tm(g_ns_read); src.file->read_at(rawp, (size_t)cm.total_compressed, off); }
  const uint8_t* chunk_data = rawp - off;         // so that chunk_data + file_offset addresses the byte
  int64_t values_seen = 0;      // rows of the row group the pages walked so far cover
  int64_t out_pos = 0;          // kept rows emitted so far (the chunk's output rows)
  size_t keep_i = 0;            // first kept range that may still overlap the next page
  std::vector<uint8_t> tmp;
  // a PLAIN page's values cut into chunks of 4096: the units the run-at-a-time decode kernel hands to its waves
  static const int32_t kPlainChunk = getenv("COMET_PQ_PLAIN_CHUNK") ? std::max(512, atoi(getenv("COMET_PQ_PLAIN_CHUNK")) & ~511) : 2048;
  auto plain_chunks = [&](PqPage& pg, int64_t values_off_flagged) {
    if (cp.is_string) return;
    pg.idx_run_first = (int32_t)idx_runs.size();
    const int64_t unit_bits = cp.kind == PQ_BOOL ? 1 : (int64_t)cp.src_width * 8;
    for (int32_t v = 0; v < pg.value_count; v += kPlainChunk) {
      PqRun r;
      memset(&r, 0, sizeof r);
      r.is_rle = 2;
      r.value_start = v;
      r.count = std::min(kPlainChunk, pg.value_count - v);
      r.byte_off = values_off_flagged + ((int64_t)v * unit_bits) / 8;
      idx_runs.push_back(r);
    }
    pg.idx_run_count = (int32_t)idx_runs.size() - pg.idx_run_first;
  };
  // the kept pieces of the page [r0, r1): one PqPage entry each, passing over the page's first levels / non-NULL values
  // (`levels`: the bytes the page's definition-level runs refer to, addressed like the runs' byte_off without its region flag)
  size_t cur_values_end = 0;    // end of the current page's staged value bytes (0: a device-inflated page, checked by the kernel's error word)
  auto emit = [&](const PqPage& pg_in, int64_t r0, int64_t r1, const uint8_t* levels) {
    PqPage pg = pg_in;
    auto non_null_before = [&](int64_t upto) -> int64_t {      // values among the page's first `upto` levels
      if (max_def == 0 || pg.def_run_count == 0) return upto;
      int64_t cnt = 0;
      for (int32_t k = 0; k < pg.def_run_count; k++) {
        const PqRun& r = def_runs[(size_t)(pg.def_run_first + k)];
        if (r.value_start >= upto) break;
        const int64_t m = std::min<int64_t>(r.count, upto - r.value_start);
        if (r.is_rle) {
          if (r.rle_value == (uint32_t)max_def) cnt += m;
        } else if (def_bw == 1) {
          const uint8_t* b = levels + (r.byte_off & ~kInflatedBit);
          for (int64_t i = 0; i < m; i++) cnt += (b[i >> 3] >> (i & 7)) & 1;     // max_def == 1: one bit per level
        } else {
          const uint8_t* b = levels + (r.byte_off & ~kInflatedBit);
          for (int64_t i = 0; i < m; i++) {
            const int64_t bit = i * def_bw;
            const uint32_t w = (uint32_t)b[bit >> 3] | ((uint32_t)b[(bit >> 3) + 1] << 8);      // (def_bw ≤ 8: a level spans at most two bytes; staged pages are followed by 16 zero bytes)
            cnt += ((w >> (bit & 7)) & ((1u << def_bw) - 1)) == (uint32_t)max_def;
          }
        }
      }
      return cnt;
    };
    // what the page's runs / PLAIN bytes hold: its non-NULL values; a PLAIN page's values are cut into the run-at-a-time kernel's chunks
    pg.value_count = (int32_t)non_null_before(pg.num_values);
    if (pg.encoding == 0 && !cp.is_string && cur_values_end && cp.kind != PQ_BOOL && !(pg.values_off & kInflatedBit) &&
        (uint64_t)pg.values_off + (uint64_t)pg.value_count * (uint64_t)pg.width > cur_values_end)
      throw CometError("parquet: a PLAIN page holds fewer bytes than its " + std::to_string(pg.value_count) + " values need");
    if (pg.encoding == 0) plain_chunks(pg, pg.values_off);
    if (!src.keep) {
      PqPage q = pg;
      q.row_start = out_pos;
      pages.push_back(q);
      out_pos += r1 - r0;
      return;
    }
    const Ranges& keep = *src.keep;
    while (keep_i < keep.size() && keep[keep_i].second <= r0) keep_i++;
    // Fixed-width columns: every kept piece gets its OWN units for the run-at-a-time kernel — the page's runs (or PLAIN chunks) clipped to
    // the piece's values, a bit-packed run possibly starting inside a byte (PqRun.pad = the bit) — so pruned scans decode at the same rate
    // as full ones (they took the row-at-a-time kernel, a quarter of the HBM roofline).  The page's own runs leave the table: nothing
    // refers to them any more.  (Strings keep the shared runs and the row-at-a-time path; COMET_PQ_DECODE_ROWS keeps it for everything.)
    static const bool clip_units = getenv("COMET_PQ_DECODE_ROWS") == nullptr;
    const bool clip = clip_units && !cp.is_string;
    std::vector<PqRun> page_runs;
    if (clip) {
      page_runs.assign(idx_runs.begin() + pg.idx_run_first, idx_runs.begin() + pg.idx_run_first + pg.idx_run_count);
      idx_runs.resize((size_t)pg.idx_run_first);
    }
    for (size_t k = keep_i; k < keep.size() && keep[k].first < r1; k++) {
      const int64_t a = std::max(keep[k].first, r0), b = std::min(keep[k].second, r1);
      if (a >= b) continue;
      PqPage q = pg;
      q.row_start = out_pos;
      q.num_values = (int32_t)(b - a);
      q.lvl_skip = (int32_t)(a - r0);
      q.val_skip = (int32_t)non_null_before(a - r0);
      if (clip) {
        const int64_t v0 = q.val_skip, v1 = std::min<int64_t>(non_null_before(b - r0), pg.value_count);
        q.idx_run_first = (int32_t)idx_runs.size();
        for (const PqRun& r : page_runs) {
          const int64_t lo = std::max<int64_t>(r.value_start, v0), hi = std::min<int64_t>((int64_t)r.value_start + r.count, v1);
          if (lo >= hi) continue;
          PqRun c = r;
          const int64_t skip = lo - r.value_start;
          c.value_start = (int32_t)(lo - v0);
          c.count = (int32_t)(hi - lo);
          if (r.is_rle != 1) {
            const int64_t unit_bits = r.is_rle == 0 ? (int64_t)pg.bit_width : (cp.kind == PQ_BOOL ? 1 : (int64_t)cp.src_width * 8);
            c.byte_off += (skip * unit_bits) / 8;
            c.pad = (int32_t)((skip * unit_bits) % 8);
          }
          idx_runs.push_back(c);
        }
        q.idx_run_count = (int32_t)idx_runs.size() - q.idx_run_first;
        q.value_count = (int32_t)std::max<int64_t>(v1 - v0, 0);
      }
      pages.push_back(q);
      out_pos += b - a;
    }
  };
  auto page_is_kept = [&](int64_t r0, int64_t r1) {
    if (!src.keep) return true;
    const Ranges& keep = *src.keep;
    while (keep_i < keep.size() && keep[keep_i].second <= r0) keep_i++;
    return keep_i < keep.size() && keep[keep_i].first < r1;
  };
  while (values_seen < cm.num_values && off < chunk_end) {
    pq::PageHeader h = pq::parse_page_header(chunk_data + off, (size_t)(chunk_end - off));
    const uint8_t* body = chunk_data + off + h.header_len;
    if (h.compressed_size < 0 || h.uncompressed_size < 0 || (int64_t)h.header_len + h.compressed_size > chunk_end - off)
      throw CometError("parquet: page of " + std::to_string(h.compressed_size) + " bytes runs past its column chunk");
    off += (int64_t)h.header_len + h.compressed_size;
    if (h.type == pq::DICTIONARY_PAGE) {
      // A fixed-width dictionary (PLAIN values) of a chunk that is read in place: the device inflates it like a data page and the decode
      // kernels index it where it lands — the host neither decompresses it nor sends it again with the column's tables (TPC-H
      // l_extendedprice: a 1 MiB dictionary per row group before the writer falls back to PLAIN pages).
      if (in_place && !cp.is_string && hc.dev_dict < 0 && (h.encoding == pq::PLAIN || h.encoding == pq::PLAIN_DICTIONARY) && h.uncompressed_size >= kMinDevicePage &&
          (int64_t)h.compressed_size <= (int64_t)h.uncompressed_size + h.uncompressed_size / 6 + 64) {
        const size_t ipage = (hc.ipos + 15) & ~(size_t)15, un_len = (size_t)h.uncompressed_size, comp_len = (size_t)h.compressed_size;
        bool ok = ipage + un_len + 32 <= staged_cap;
        PqInflate job;
        job.src_off = (int64_t)(body - staged);
        job.dst_off = (int64_t)ipage;
        job.src_len = (int32_t)comp_len;
        job.dst_len = (int32_t)un_len;
        job.pad = 0;
        comet_zstd2::PageWalk zw;
        if (ok && cm.codec == pq::SNAPPY) {
          job.preamble = comet_snappy2::preamble_length(body, (int32_t)comp_len);
          ok = job.preamble > 0;
          if (ok) hc.inflate.push_back(job);
        } else if (ok && cm.codec == pq::ZSTD && so.device_zstd) {
          ok = comet_zstd2::scan_page(body, (uint32_t)comp_len, (uint32_t)un_len, zw);
          if (ok) {
            job.preamble = (int32_t)hc.zblocks.size();
            job.pad = (int32_t)zw.blocks.size();
            hc.zblocks.insert(hc.zblocks.end(), zw.blocks.begin(), zw.blocks.end());
            hc.zinflate.push_back(job);
          }
        } else {
          ok = false;
        }
        if (ok) {
          if (hc.raw_hi == 0) hc.raw_lo = (size_t)job.src_off;
          hc.raw_hi = (size_t)job.src_off + comp_len;
          hc.ipos = ipage + un_len;
          hc.dev_dict = (int64_t)ipage;
          continue;
        }
      }
      tmp.resize((size_t)h.uncompressed_size + 8);
      pq::decompress(cm.codec, body, (size_t)h.compressed_size, tmp.data(), (size_t)h.uncompressed_size);
      if (h.encoding != pq::PLAIN && h.encoding != pq::PLAIN_DICTIONARY) throw CometError("parquet: unsupported dictionary page encoding");
      if (cp.is_string) {
        dict_offs.assign(1, 0);
        size_t p = 0;
        for (int i = 0; i < h.num_values; i++) {
          if (p + 4 > (size_t)h.uncompressed_size) throw CometError("parquet: truncated dictionary page");
          uint32_t len;
          memcpy(&len, tmp.data() + p, 4);
          p += 4;
          if (p + len > (size_t)h.uncompressed_size) throw CometError("parquet: truncated dictionary page");
          dict_bytes.insert(dict_bytes.end(), tmp.begin() + (long)p, tmp.begin() + (long)(p + len));
          p += len;
          dict_offs.push_back((int32_t)dict_bytes.size());
        }
      } else {
        dict_bytes.assign(tmp.begin(), tmp.begin() + h.uncompressed_size);
      }
      continue;
    }
    if (h.type != pq::DATA_PAGE && h.type != pq::DATA_PAGE_V2) continue;   // index pages etc.
    if (!page_is_kept(values_seen, values_seen + h.num_values)) {            // ruled out by the page index: not even decompressed
      values_seen += h.num_values;
      hc.pages_skipped++;
      continue;
    }
    if (spos + (size_t)h.uncompressed_size + 16 > staged_cap) {
      // total_uncompressed_size excludes nothing we stage, but stay safe against odd writers
      throw CometError("parquet: column chunk larger than its declared uncompressed size");
    }
    PqPage pg;
    memset(&pg, 0, sizeof pg);
    pg.row_start = values_seen;
    pg.num_values = h.num_values;
    pg.kind = cp.kind;
    pg.width = cp.src_width;
    pg.dec_scale_up = cp.dec_scale_up;
    size_t page_begin = spos, vals_begin, page_end;
    // zstd: PLAIN pages of fixed-width columns cross PCIe compressed.  The host walks the frame's block headers (sizes, modes, where the
    // tables and bitstreams sit: comet_zstd2::scan_page) and, for a v1 page of an optional column, decodes the page's first bytes — its
    // definition levels — from the first block's first literals and sequences; the device does the rest (device/zstd2.hpp).
    // Dictionary-encoded pages go the same way when the chunk is read whole: their run headers cannot be read through an entropy-coded stream,
    // so the page is registered as PENDING — the device inflates it, the index section comes back over PCIe and the host parses the headers
    // then (read_columns, "deferred").  The bit width — the index section's first byte — is read here with the levels.
    const bool z_dict = (h.encoding == pq::RLE_DICTIONARY || h.encoding == pq::PLAIN_DICTIONARY) && so.device_zstd_dict && src.keep == nullptr;
    if (so.device_snappy && so.device_zstd && !nested_leaf && cm.codec == pq::ZSTD && !cp.is_string && (h.encoding == pq::PLAIN || z_dict) && h.uncompressed_size >= kMinDevicePage &&
        (h.type == pq::DATA_PAGE || (h.v2_compressed && !h.rep_bytes && h.def_bytes >= 0 && h.compressed_size > h.def_bytes && h.uncompressed_size > h.def_bytes)) &&
        (in_place || h.compressed_size <= h.uncompressed_size)) {
      const size_t comp_off = h.type == pq::DATA_PAGE ? 0 : (size_t)h.def_bytes;
      const size_t comp_len = (size_t)h.compressed_size - comp_off, un_len = (size_t)h.uncompressed_size - comp_off;
      comet_zstd2::PageWalk zw;
      HostTimer tm_walk(g_ns_walk);
      bool ok = comet_zstd2::scan_page(body + comp_off, (uint32_t)comp_len, (uint32_t)un_len, zw);
      size_t lvl = 0;
      if (ok && h.type == pq::DATA_PAGE && max_def > 0) {
        if (h.def_encoding != pq::RLE) throw CometError("parquet: only RLE definition levels are supported");
        const size_t first = std::min<size_t>(un_len, 64);
        tmp.resize(first + 8);
        ok = first >= 4 && comet_zstd2::host_prefix(body, (uint32_t)comp_len, zw, tmp.data(), first) == first;
        if (ok) {
          uint32_t dl;
          memcpy(&dl, tmp.data(), 4);
          lvl = 4 + (size_t)dl;
          ok = lvl <= un_len && lvl <= ((size_t)1 << 20);
          const size_t need = lvl + (z_dict ? 1 : 0);
          if (ok && need > first) {
            tmp.resize(need + 8);
            ok = need <= un_len && comet_zstd2::host_prefix(body, (uint32_t)comp_len, zw, tmp.data(), need) == need;
          }
        }
      } else if (ok && z_dict) {
        tmp.resize(16);
        ok = un_len >= 1 && comet_zstd2::host_prefix(body + comp_off, (uint32_t)comp_len, zw, tmp.data(), 1) == 1;
      }
      int zbw = 0;
      if (ok && z_dict) {
        ok = lvl + 1 <= un_len;
        if (ok) zbw = tmp[lvl];
        if (ok && zbw > 32) throw CometError("parquet: dictionary index bit width > 32");
      }
      const size_t ipage = (hc.ipos + 15) & ~(size_t)15;
      if (ok && ipage + un_len + 32 <= staged_cap) {
        if (h.type == pq::DATA_PAGE) {
          if (max_def > 0) {
            const size_t first = def_runs.size();
            pg.def_run_first = (int32_t)first;
            parse_hybrid_runs(tmp.data() - ipage, ipage + 4, ipage + lvl, 1, h.num_values, def_runs);      // positions in the decompressed region, where the device puts these bytes
            pg.def_run_count = (int32_t)(def_runs.size() - first);
            for (size_t r = first; r < def_runs.size(); r++) def_runs[r].byte_off |= kInflatedBit;
          }
        } else {
          memcpy(staged + spos, body, (size_t)h.def_bytes);
          if (max_def > 0 && h.def_bytes) {
            pg.def_run_first = (int32_t)def_runs.size();
            parse_hybrid_runs(staged, spos, spos + (size_t)h.def_bytes, 1, h.num_values, def_runs);
            pg.def_run_count = (int32_t)def_runs.size() - pg.def_run_first;
          }
          spos += (size_t)h.def_bytes;
        }
        size_t cpos;
        if (in_place) {
          cpos = (size_t)(body + comp_off - staged);
          if (hc.raw_hi == 0) hc.raw_lo = cpos;
          hc.raw_hi = cpos + comp_len;
        } else {
          cpos = (spos + 15) & ~(size_t)15;
          if (cpos + comp_len + 32 > staged_cap) throw CometError("parquet: column chunk larger than its declared uncompressed size");
          memcpy(staged + cpos, body + comp_off, comp_len);
          memset(staged + cpos + comp_len, 0, 16);
          spos = cpos + comp_len;
        }
        PqInflate job;
        job.src_off = (int64_t)cpos;
        job.dst_off = (int64_t)ipage;
        job.src_len = (int32_t)comp_len;
        job.dst_len = (int32_t)un_len;
        job.preamble = (int32_t)hc.zblocks.size();
        job.pad = (int32_t)zw.blocks.size();
        hc.zblocks.insert(hc.zblocks.end(), zw.blocks.begin(), zw.blocks.end());
        hc.zinflate.push_back(job);
        hc.ipos = ipage + un_len;
        if (z_dict) {
          pg.encoding = 1;
          pg.bit_width = zbw;
          pg.values_off = (int64_t)(ipage + lvl + 1) | kInflatedBit;
          pg.idx_run_first = (int32_t)idx_runs.size();
          if (zbw == 0 || lvl + 1 == un_len) {             // every index is 0 (or a page of NULLs only): one run, nothing to read back
            PqRun r;
            memset(&r, 0, sizeof r);
            r.is_rle = 1;
            r.count = h.num_values;
            idx_runs.push_back(r);
          } else {
            hc.pending.push_back({pages.size(), ipage + lvl + 1, ipage + un_len, -1});
          }
          pg.idx_run_count = (int32_t)idx_runs.size() - pg.idx_run_first;
        } else {
          pg.encoding = 0;
          pg.values_off = (int64_t)(ipage + lvl) | kInflatedBit;
        }
        emit(pg, values_seen, values_seen + h.num_values, h.type == pq::DATA_PAGE ? tmp.data() - ipage : staged);
        values_seen += h.num_values;
        continue;
      }
    }
    // Device decompression: PLAIN fixed-width values under snappy need nothing from the host but the definition levels (the first bytes
    // of a v1 page's stream; outside the stream in a v2 page), so the body crosses PCIe compressed and a GPU workgroup inflates it.
    // (a page that did not compress — bit-packed dictionary indices of random values, doubles — is a little LARGER than its content: read
    // in place it crosses as it is and the one-wave kernel copies it at HBM speed; staged by copy it must fit the slot's uncompressed size)
    const int64_t dev_max_compressed = in_place ? (int64_t)h.uncompressed_size + h.uncompressed_size / 6 + 64 : (int64_t)h.uncompressed_size;
    const bool dev_shape = so.device_snappy && !nested_leaf && cm.codec == pq::SNAPPY && !cp.is_string && h.uncompressed_size >= kMinDevicePage &&
                           h.compressed_size <= dev_max_compressed && (h.type == pq::DATA_PAGE || (h.v2_compressed && !h.rep_bytes && h.compressed_size > h.def_bytes));
    bool dev_page = dev_shape && h.encoding == pq::PLAIN;
    // … and so do dictionary-encoded pages: the run headers of the index section are read THROUGH the compressed stream (SnappyView:
    // bit-packed indices do not compress, the stream is a handful of long literals), so the host does not inflate 1 MiB to look at a few
    // hundred header bytes — decompressing these pages on host threads was what the scan waited for once the PLAIN pages had moved to
    // the device (SF10 Q6: 8 of 14 ms).  A page that compresses into many elements is inflated on the host as before (it is small).
    pq::SnappyView view;
    const bool dict_enc = dev_shape && !dev_page && (h.encoding == pq::RLE_DICTIONARY || h.encoding == pq::PLAIN_DICTIONARY) && so.device_dict_pages;
    // (the device walks the run headers once it has inflated the page — nothing of the stream is looked at here but its first bytes)
    const bool walk_on_device = dict_enc && so.device_runs && so.device_runs_snappy && src.keep == nullptr;
    const bool dev_dict = walk_on_device ||
                          (dict_enc && view.build(body + (h.type == pq::DATA_PAGE ? 0 : h.def_bytes), (size_t)h.compressed_size - (h.type == pq::DATA_PAGE ? 0 : (size_t)h.def_bytes), 2048) &&
                           view.out_len == (size_t)h.uncompressed_size - (h.type == pq::DATA_PAGE ? 0 : (size_t)h.def_bytes));
    dev_page = dev_page || dev_dict;
    if (dev_page) {
      const size_t ipage = (hc.ipos + 15) & ~(size_t)15;
      size_t comp_off = 0, comp_len = (size_t)h.compressed_size, un_len = (size_t)h.uncompressed_size, lvl = 0;
      if (h.type == pq::DATA_PAGE) {
        if (max_def > 0) {
          if (h.def_encoding != pq::RLE) throw CometError("parquet: only RLE definition levels are supported");
          uint8_t pre[4];
          if (pq::snappy_prefix(body, comp_len, pre, 4) != 4) throw CometError("parquet: data page shorter than its level header");
          uint32_t dl;
          memcpy(&dl, pre, 4);
          lvl = 4 + (size_t)dl;
          if (lvl > un_len) throw CometError("parquet: definition levels longer than their page");
          const size_t want = lvl + (walk_on_device && lvl < un_len ? 1 : 0);      // … and the index section's first byte, the bit width
          tmp.resize(want + 8);
          if (pq::snappy_prefix(body, comp_len, tmp.data(), want) != want) throw CometError("parquet: data page shorter than its definition levels");
          const size_t first = def_runs.size();
          pg.def_run_first = (int32_t)first;
          // positions in the coordinates of the decompressed region, where the device will put these same bytes
          parse_hybrid_runs(tmp.data() - ipage, ipage + 4, ipage + lvl, 1, h.num_values, def_runs);
          pg.def_run_count = (int32_t)(def_runs.size() - first);
          for (size_t r = first; r < def_runs.size(); r++) def_runs[r].byte_off |= kInflatedBit;
        }
      } else {
        memcpy(staged + spos, body, (size_t)h.def_bytes);
        if (max_def > 0 && h.def_bytes) {
          pg.def_run_first = (int32_t)def_runs.size();
          parse_hybrid_runs(staged, spos, spos + (size_t)h.def_bytes, 1, h.num_values, def_runs);
          pg.def_run_count = (int32_t)def_runs.size() - pg.def_run_first;
        }
        spos += (size_t)h.def_bytes;
        comp_off = (size_t)h.def_bytes;
        comp_len -= (size_t)h.def_bytes;
        un_len -= (size_t)h.def_bytes;
      }
      size_t cpos;
      if (in_place) {
        // the body stays where pread() put it (any alignment; the bytes behind it are the next page's header or the slot's slack)
        cpos = (size_t)(body + comp_off - staged);
        if (ipage + un_len + 32 > staged_cap) throw CometError("parquet: column chunk larger than its declared uncompressed size");
        if (hc.raw_hi == 0) hc.raw_lo = cpos;
        hc.raw_hi = cpos + comp_len;
      } else {
        cpos = (spos + 15) & ~(size_t)15;
        if (cpos + comp_len + 32 > staged_cap || ipage + un_len + 32 > staged_cap) throw CometError("parquet: column chunk larger than its declared uncompressed size");
        memcpy(staged + cpos, body + comp_off, comp_len);
        memset(staged + cpos + comp_len, 0, 16);
      }
      PqInflate job;
      job.src_off = (int64_t)cpos;
      job.dst_off = (int64_t)ipage;
      job.src_len = (int32_t)comp_len;
      job.dst_len = (int32_t)un_len;
      job.preamble = comet_snappy2::preamble_length(body + comp_off, (int32_t)comp_len);     // where the stream's first element starts
      if (job.preamble <= 0) throw CometError("parquet: malformed snappy page (length preamble)");      // (rejected here like the dictionary page's, not left to the device kernel)
      job.pad = 0;
      hc.inflate.push_back(job);
      if (!in_place) spos = cpos + comp_len;
      hc.ipos = ipage + un_len;
      if (dev_dict) {
        // the index section as the device will see it: [lvl] = bit width, then the hybrid runs; positions in the decompressed region
        if (lvl + 1 > un_len) throw CometError("parquet: dictionary-encoded page without a bit width");
        auto get = [&](size_t p) { return view.at(p - ipage); };
        pg.encoding = 1;
        if (!walk_on_device) {
          pg.bit_width = get(ipage + lvl);
        } else if (h.type == pq::DATA_PAGE && max_def > 0) {
          pg.bit_width = tmp[lvl];
        } else {
          uint8_t b0 = 0;
          if (pq::snappy_prefix(body + comp_off, comp_len, &b0, 1) != 1) throw CometError("parquet: dictionary-encoded page without a bit width");
          pg.bit_width = b0;
        }
        if (pg.bit_width > 32) throw CometError("parquet: dictionary index bit width > 32");
        pg.values_off = (int64_t)(ipage + lvl + 1) | kInflatedBit;
        const size_t first = idx_runs.size();
        pg.idx_run_first = (int32_t)first;
        const bool is_pending = walk_on_device && pg.bit_width != 0 && lvl + 1 < un_len;
        if (is_pending) {
          // PENDING, like a zstd page of this kind: the column's deferred step counts, places and writes the runs on the device (read_columns)
          hc.pending.push_back({pages.size(), ipage + lvl + 1, ipage + un_len, -1});
        } else if (pg.bit_width == 0 || walk_on_device) {      // every index is 0 / a page of NULLs only: one run, nothing to walk
          PqRun r;
          memset(&r, 0, sizeof r);
          r.is_rle = 1;
          r.count = h.num_values;
          idx_runs.push_back(r);
        } else {
          parse_hybrid_runs_from(get, ipage + lvl + 1, ipage + un_len, pg.bit_width, -1, idx_runs);
          for (size_t r = first; r < idx_runs.size(); r++) idx_runs[r].byte_off |= kInflatedBit;
        }
        pg.idx_run_count = (int32_t)(idx_runs.size() - first);
        if (pg.idx_run_count == 0 && !is_pending) {   // page of NULLs only
          PqRun r;
          memset(&r, 0, sizeof r);
          r.is_rle = 1;
          r.count = h.num_values;
          idx_runs.push_back(r);
          pg.idx_run_count = 1;
        }
      } else {
        pg.encoding = 0;
        pg.values_off = (int64_t)(ipage + lvl) | kInflatedBit;
      }
      // the level bytes the runs refer to: a v1 page's were decoded into `tmp` (addressed in the coordinates of the decompressed region), a v2
      // page's were copied into the staged region
      emit(pg, values_seen, values_seen + h.num_values, h.type == pq::DATA_PAGE ? tmp.data() - ipage : staged);
      values_seen += h.num_values;
      continue;
    }
    if (h.type == pq::DATA_PAGE) {
      { HostTimer tm(g_ns_inflate); pq::decompress(cm.codec, body, (size_t)h.compressed_size, staged + spos, (size_t)h.uncompressed_size); }
      page_end = spos + (size_t)h.uncompressed_size;
      size_t p = page_begin;
      if (cp.max_rep > 0) {      // a list's element: [length][repetition levels] come first
        if (h.rep_encoding != pq::RLE) throw CometError("parquet: only RLE repetition levels are supported");
        uint32_t rl;
        if (p + 4 > page_end) throw CometError("parquet: truncated data page");
        memcpy(&rl, staged + p, 4);
        p += 4;
        if ((size_t)rl > page_end - p) throw CometError("parquet: repetition levels longer than their page");
        pg.rep_run_first = (int32_t)hc.rep_runs.size();
        parse_hybrid_runs(staged, p, p + rl, rep_bw, h.num_values, hc.rep_runs);
        pg.rep_run_count = (int32_t)hc.rep_runs.size() - pg.rep_run_first;
        p += rl;
      }
      if (max_def > 0) {
        if (h.def_encoding != pq::RLE) throw CometError("parquet: only RLE definition levels are supported");
        uint32_t dl;
        if (p + 4 > page_end) throw CometError("parquet: truncated data page");
        memcpy(&dl, staged + p, 4);
        p += 4;
        if ((size_t)dl > page_end - p) throw CometError("parquet: definition levels longer than their page");
        pg.def_run_first = (int32_t)def_runs.size();
        parse_hybrid_runs(staged, p, p + dl, def_bw, h.num_values, def_runs);
        pg.def_run_count = (int32_t)def_runs.size() - pg.def_run_first;
        p += dl;
      }
      vals_begin = p;
    } else {
      // v2: levels are never compressed and precede the (optionally compressed) values
      if (h.rep_bytes && cp.max_rep == 0) throw CometError("parquet: repetition levels in a column that is not repeated");
      if (h.rep_bytes < 0 || h.def_bytes < 0 || (int64_t)h.rep_bytes + h.def_bytes > h.compressed_size || (int64_t)h.rep_bytes + h.def_bytes > h.uncompressed_size)
        throw CometError("parquet: v2 page levels longer than the page");
      const size_t lv = (size_t)h.rep_bytes + (size_t)h.def_bytes;
      memcpy(staged + spos, body, lv);
      if (cp.max_rep > 0 && h.rep_bytes) {
        pg.rep_run_first = (int32_t)hc.rep_runs.size();
        parse_hybrid_runs(staged, spos, spos + (size_t)h.rep_bytes, rep_bw, h.num_values, hc.rep_runs);
        pg.rep_run_count = (int32_t)hc.rep_runs.size() - pg.rep_run_first;
      }
      if (max_def > 0 && h.def_bytes) {
        pg.def_run_first = (int32_t)def_runs.size();
        parse_hybrid_runs(staged, spos + (size_t)h.rep_bytes, spos + lv, def_bw, h.num_values, def_runs);
        pg.def_run_count = (int32_t)def_runs.size() - pg.def_run_first;
      }
      vals_begin = spos + lv;
      const size_t vcomp = (size_t)h.compressed_size - lv, vun = (size_t)h.uncompressed_size - lv;
      { HostTimer tm(g_ns_inflate); pq::decompress(h.v2_compressed ? cm.codec : pq::UNCOMPRESSED, body + lv, vcomp, staged + vals_begin, vun); }
      page_end = vals_begin + vun;
    }
    int value_encoding = h.encoding;
    if (value_encoding == pq::DELTA_BINARY_PACKED || value_encoding == pq::DELTA_LENGTH_BYTE_ARRAY || value_encoding == pq::DELTA_BYTE_ARRAY ||
        value_encoding == pq::BYTE_STREAM_SPLIT) {
      // encodings the device kernels do not read: the page's values are rewritten as PLAIN in place (parquet_meta.cpp); delta pages grow,
      // which staged_capacity allowed for
      std::vector<uint8_t> plain;
      const uint8_t* vsrc = staged + vals_begin;
      const size_t vlen = page_end - vals_begin;
      if (value_encoding == pq::DELTA_BINARY_PACKED) {
        if (cp.is_string || (cp.src_width != 4 && cp.src_width != 8) || cm.type == pq::FLOAT || cm.type == pq::DOUBLE)
          throw CometError("parquet: DELTA_BINARY_PACKED values of a column that is not INT32 / INT64");
        pq::delta_binary_to_plain(vsrc, vlen, cp.src_width, h.num_values, plain);
      } else if (value_encoding == pq::DELTA_LENGTH_BYTE_ARRAY) {
        if (!cp.is_string) throw CometError("parquet: DELTA_LENGTH_BYTE_ARRAY values of a column that is not BYTE_ARRAY");
        pq::delta_length_byte_array_to_plain(vsrc, vlen, h.num_values, plain);
      } else if (value_encoding == pq::DELTA_BYTE_ARRAY) {
        if (!cp.is_string) throw CometError("parquet: DELTA_BYTE_ARRAY values of a column that is not BYTE_ARRAY");
        pq::delta_byte_array_to_plain(vsrc, vlen, h.num_values, plain);
      } else {
        if (cp.is_string || cp.src_width <= 0) throw CometError("parquet: BYTE_STREAM_SPLIT values of a variable-length column");
        pq::byte_stream_split_to_plain(vsrc, vlen, cp.src_width, plain);
      }
      if (vals_begin + plain.size() + 16 > staged_cap) throw CometError("parquet: decoded " + std::string(value_encoding == pq::BYTE_STREAM_SPLIT ? "BYTE_STREAM_SPLIT" : "DELTA") + " page does not fit the column chunk's staging slot");
      if (!plain.empty()) memcpy(staged + vals_begin, plain.data(), plain.size());
      page_end = vals_begin + plain.size();
      value_encoding = pq::PLAIN;
    }
    if (value_encoding == pq::PLAIN) {
      pg.encoding = 0;
      pg.values_off = (int64_t)vals_begin;
      if (cp.is_string) {
        pg.str_first = (int64_t)str_offs.size();
        size_t p = vals_begin;
        while (p + 4 <= page_end) {
          uint32_t len;
          memcpy(&len, staged + p, 4);
          p += 4;
          str_offs.push_back((int64_t)p);
          p += len;
        }
      }
    } else if (h.encoding == pq::RLE_DICTIONARY || h.encoding == pq::PLAIN_DICTIONARY) {
      pg.encoding = 1;
      pg.bit_width = staged[vals_begin];
      if (pg.bit_width > 32) throw CometError("parquet: dictionary index bit width > 32");
      pg.values_off = (int64_t)vals_begin + 1;
      pg.idx_run_first = (int32_t)idx_runs.size();
      if (pg.bit_width == 0) {
        PqRun r;
        memset(&r, 0, sizeof r);
        r.is_rle = 1;
        r.count = h.num_values;
        idx_runs.push_back(r);
      } else {
        parse_hybrid_runs(staged, vals_begin + 1, page_end, pg.bit_width, -1, idx_runs);
      }
      pg.idx_run_count = (int32_t)idx_runs.size() - pg.idx_run_first;
      if (pg.idx_run_count == 0) {   // page of NULLs only
        PqRun r;
        memset(&r, 0, sizeof r);
        r.is_rle = 1;
        r.count = h.num_values;
        idx_runs.push_back(r);
        pg.idx_run_count = 1;
      }
    } else if (h.encoding == pq::RLE && cp.kind == PQ_BOOL) {
      // RLE booleans (what data-page-v2 writers emit): a 4-byte length, then hybrid runs of 1-bit values — decoded as indices into {0, 1}
      if (vals_begin + 4 > page_end) throw CometError("parquet: truncated RLE boolean page");
      uint32_t rl;
      memcpy(&rl, staged + vals_begin, 4);
      if (vals_begin + 4 + (size_t)rl > page_end) throw CometError("parquet: RLE boolean data longer than its page");
      if (dict_bytes.empty()) { dict_bytes.assign(16, 0); dict_bytes[1] = 1; }
      pg.encoding = 1;
      pg.bit_width = 1;
      pg.kind = PQ_COPY1;
      pg.width = 1;
      pg.values_off = (int64_t)vals_begin + 4;
      pg.idx_run_first = (int32_t)idx_runs.size();
      parse_hybrid_runs(staged, vals_begin + 4, vals_begin + 4 + (size_t)rl, 1, -1, idx_runs);
      pg.idx_run_count = (int32_t)idx_runs.size() - pg.idx_run_first;
      if (pg.idx_run_count == 0) {   // page of NULLs only
        PqRun r;
        memset(&r, 0, sizeof r);
        r.is_rle = 1;
        r.count = h.num_values;
        idx_runs.push_back(r);
        pg.idx_run_count = 1;
      }
    } else {
      throw CometError("parquet: value encoding " + std::to_string(h.encoding) +
                       " is not supported (PLAIN, RLE_DICTIONARY, RLE booleans, the three DELTA encodings and BYTE_STREAM_SPLIT are)");
    }
    spos = page_end;
    cur_values_end = dev_page ? 0 : page_end;
    emit(pg, values_seen, values_seen + h.num_values, staged);
    values_seen += h.num_values;
  }
  if (values_seen != (cp.max_rep > 0 ? cm.num_values : rg.num_rows) || out_pos != hc.n_rows) throw CometError("parquet: column chunk values do not add up to the row group's rows");
  if (pages.empty()) throw CometError("parquet: column chunk without data pages");
  memset(staged + spos, 0, 16);
  hc.spos = spos;
  for (const PqRun& r : def_runs)
    if (!r.is_rle || r.rle_value != (uint32_t)max_def) { hc.no_nulls = false; break; }
  str_offs.push_back(0);   // sentinel
}

// A column this file does not have (schema evolution): one synthetic page covering the row group — all NULL, or the column's default
// value (NativeScanCommon.default_values; schema_adapter.rs replace_missing_with_defaults) as a one-entry dictionary every row points at.
size_t synth_capacity(const DType& t, int64_t rows) { return t.id == TypeId::Bool ? (size_t)((rows + 7) / 8) + 32 : 32; }
void synth_chunk(const StructField& want, const Expr* dflt, int64_t rows, HostChunk& hc, uint8_t* staged, size_t cap) {
  const DType& t = want.dtype;
  const bool is_null = !dflt || dflt->lit_null;
  if (dflt && dflt->kind != ExprKind::Literal) throw CometError("Parquet column '" + want.name + "': default value is not a literal");
  hc.cp.missing = true;
  hc.cp.is_string = t.id == TypeId::String || t.id == TypeId::Bytes;
  hc.cp.out_width = out_width_of(t);
  hc.n_rows = rows;
  hc.compressed = 0;
  hc.max_def = is_null ? 1 : 0;
  hc.no_nulls = !is_null;
  PqPage pg;
  memset(&pg, 0, sizeof pg);
  pg.num_values = (int32_t)rows;
  pg.value_count = is_null ? 0 : (int32_t)rows;
  if (is_null) {
    PqRun r;
    memset(&r, 0, sizeof r);
    r.is_rle = 1;
    r.count = (int32_t)rows;
    r.rle_value = 0;
    hc.def_runs.push_back(r);
    pg.def_run_first = 0;
    pg.def_run_count = 1;
  }
  pg.encoding = 1;
  pg.bit_width = 0;
  PqRun ir;
  memset(&ir, 0, sizeof ir);
  ir.is_rle = 1;
  ir.count = (int32_t)rows;
  hc.idx_runs.push_back(ir);
  pg.idx_run_first = 0;
  pg.idx_run_count = 1;
  size_t spos = 0;
  unsigned char raw[16] = {0};
  auto dict = [&](int kind, int width) {
    pg.kind = kind;
    pg.width = width;
    hc.dict_bytes.assign(raw, raw + 16);
  };
  switch (t.id) {
    case TypeId::Bool: {
      pg.kind = PQ_BOOL;
      pg.encoding = 0;
      pg.values_off = 0;
      // PLAIN bits: the page's runs are chunks of its values (the units of the run-at-a-time kernel), not an index run
      hc.idx_runs.clear();
      for (int64_t v = 0; v < rows; v += 4096) {
        PqRun r;
        memset(&r, 0, sizeof r);
        r.is_rle = 2;
        r.value_start = (int32_t)v;
        r.count = (int32_t)std::min<int64_t>(4096, rows - v);
        r.byte_off = v / 8;
        hc.idx_runs.push_back(r);
      }
      pg.idx_run_count = (int32_t)hc.idx_runs.size();
      spos = (size_t)((rows + 7) / 8);
      if (spos + 16 > cap) throw CometError("internal: synthetic boolean page does not fit its slot");
      memset(staged, (!is_null && dflt->lit_bool) ? 0xff : 0x00, spos);
      break;
    }
    case TypeId::Int8: case TypeId::Int16: { int32_t x = is_null ? 0 : (int32_t)dflt->lit_i64; memcpy(raw, &x, 4); dict(t.id == TypeId::Int8 ? PQ_I32_TO_I8 : PQ_I32_TO_I16, 4); break; }
    case TypeId::Int32: case TypeId::Date: { int32_t x = is_null ? 0 : (int32_t)dflt->lit_i64; memcpy(raw, &x, 4); dict(PQ_COPY4, 4); break; }
    case TypeId::Float: { float x = is_null ? 0.f : (float)dflt->lit_f64; memcpy(raw, &x, 4); dict(PQ_COPY4, 4); break; }
    case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: { int64_t x = is_null ? 0 : dflt->lit_i64; memcpy(raw, &x, 8); dict(PQ_COPY8, 8); break; }
    case TypeId::Double: { double x = is_null ? 0.0 : dflt->lit_f64; memcpy(raw, &x, 8); dict(PQ_COPY8, 8); break; }
    case TypeId::Decimal: {
      u128 v = is_null ? 0 : (u128)dflt->lit_dec;
      for (int k = 0; k < 16; k++) raw[k] = (unsigned char)(v >> (8 * (15 - k)));   // FIXED_LEN_BYTE_ARRAY decimals are big-endian
      dict(PQ_FLBA_TO_DEC, 16);
      break;
    }
    case TypeId::String: case TypeId::Bytes: {
      const std::string& b = is_null ? std::string() : dflt->lit_bytes;
      hc.dict_bytes.assign(b.begin(), b.end());
      hc.dict_offs = {0, (int32_t)b.size()};
      break;
    }
    default: throw CometError("Parquet column '" + want.name + "' is missing from a file and its type " + t.str() + " cannot be synthesised by the GPU scan yet");
  }
  memset(staged + spos, 0, 16);
  hc.spos = spos;
  hc.pages.push_back(pg);
  hc.str_offs.push_back(0);   // sentinel
}

// One selected row group of a scan: its file, its place in the output, and the rows the page index could not rule out (null: all of them)
struct Sel {
  std::shared_ptr<OpenFile> file;
  std::shared_ptr<pq::FileMeta> meta;
  int rg;
  int64_t row_off;
  const PartitionedFile* pf;
  int64_t rows;
  std::shared_ptr<Ranges> keep;
};
// Which row groups does this partition's scan read (the byte-range midpoint rule of the Parquet readers), which of them survive the
// pushed-down filters' min / max statistics, and — with a page index (ColumnIndex / OffsetIndex; parquet_exec.rs turns on DataFusion's
// page-index pruning) — which ROWS of the survivors can the filters still be true for.  Pages outside those rows are neither decompressed
// nor uploaded and the scan emits only the kept rows; the Filter above re-checks every row it gets, as it does after row-group pruning.
void select_row_groups(const Operator& op, bool page_index, std::vector<Sel>& sels, int64_t& total_rows, int64_t& row_groups_pruned, int64_t& rows_pruned_page_index,
                       bool bloom_filters = true, int64_t* row_groups_pruned_bloom = nullptr) {
  for (auto& pf : op.files) {
    // A file that is missing, or whose footer cannot be read, fails the task the way the reference classifies it
    // (jni-bridge/src/errors.rs:600-735 try_classify_file_read_error): FileNotFound { message } — Spark's readCurrentFileNotFoundError — or
    // CannotReadFile { filePath, message } — its FAILED_READ_FILE; a bad magic carries Spark's own "is not a Parquet file" (:720-735)
    std::shared_ptr<OpenFile> mf;
    std::shared_ptr<pq::FileMeta> fm;
    try {
      mf = std::make_shared<OpenFile>(path_from_uri(pf.file_path));
      fm = std::make_shared<pq::FileMeta>(pq::parse_footer(mf->footer.data(), mf->footer.size()));
    } catch (const FileMissing& e) {
      if (op.reader_api) throw;
      throw spark_error("FileNotFound", "\"message\":\"" + json_escape(e.what()) + "\"");
    } catch (const CometError& e) {
      if (e.kind != 0 || op.reader_api) throw;
      std::string msg = e.what();
      if (msg.find("not a Parquet file") != std::string::npos) msg = "Invalid Parquet file. Corrupt footer (file is not a Parquet file): " + msg;
      throw spark_error("CannotReadFile", "\"filePath\":\"" + json_escape(pf.file_path) + "\",\"message\":\"" + json_escape(msg) + "\"");
    }
    for (size_t g = 0; g < fm->row_groups.size(); g++) {
      const pq::RowGroup& rg = fm->row_groups[g];
      if (rg.columns.empty()) continue;
      const pq::ColumnMeta& c0 = rg.columns[0];
      int64_t start = (c0.dictionary_page_offset > 0 && c0.dictionary_page_offset < c0.data_page_offset) ? c0.dictionary_page_offset : c0.data_page_offset;
      int64_t comp = rg.total_compressed;
      if (comp <= 0) { comp = 0; for (auto& c : rg.columns) comp += c.total_compressed; }
      int64_t mid = start + comp / 2;
      const bool whole = pf.length <= 0;
      if (!whole && !(mid >= pf.start && mid < pf.start + pf.length)) continue;
      bool skip = false, by_bloom = false;
      BloomCache blooms{mf.get(), &rg, {}};
      for (auto& df : op.data_filters)
        if (prunes(*df, op.required_schema, *fm, rg, op.case_sensitive, bloom_filters ? &blooms : nullptr, &by_bloom)) { skip = true; break; }
      if (skip) {
        row_groups_pruned++;
        if (by_bloom && row_groups_pruned_bloom) (*row_groups_pruned_bloom)++;      // (counted in row_groups_pruned too: the row groups the scan did not read)
        continue;
      }
      std::shared_ptr<Ranges> keep;
      int64_t rows = rg.num_rows;
      if (page_index && !op.data_filters.empty()) {
        PageIndexCache pic{mf.get(), &rg, {}};
        Ranges r{{0, rg.num_rows}};
        for (auto& df : op.data_filters) r = ranges_and(r, may_match(*df, op.required_schema, *fm, rg, op.case_sensitive, pic));
        const int64_t kept = ranges_rows(r);
        if (kept == 0) { row_groups_pruned++; rows_pruned_page_index += rg.num_rows; continue; }
        if (kept < rg.num_rows) {
          keep = std::make_shared<Ranges>(std::move(r));
          rows = kept;
          rows_pruned_page_index += rg.num_rows - kept;
        }
      }
      sels.push_back({mf, fm, (int)g, total_rows, &pf, rows, keep});
      total_rows += rows;
    }
  }
}

}  // namespace

// What select_row_groups decides for a NativeScan, as JSON — host only (footers and page indexes are read, no page is): the CPU-side
// check of row-group and page-index pruning (include/comet_amd.h comet_parquet_prune_report).
std::string parquet_prune_report(const Operator& op, bool page_index, bool bloom_filters) {
  std::vector<Sel> sels;
  int64_t total = 0, rg_pruned = 0, rows_pruned = 0, rg_bloom = 0;
  select_row_groups(op, page_index, sels, total, rg_pruned, rows_pruned, bloom_filters, &rg_bloom);
  std::string j = "{\"rows\": " + std::to_string(total) + ", \"row_groups_pruned\": " + std::to_string(rg_pruned) + ", \"row_groups_pruned_bloom_filter\": " + std::to_string(rg_bloom) +
                  ", \"page_index_rows_pruned\": " + std::to_string(rows_pruned) +
                  ", \"row_groups\": [";
  for (size_t i = 0; i < sels.size(); i++) {
    const Sel& sl = sels[i];
    const int64_t n = sl.meta->row_groups[(size_t)sl.rg].num_rows;
    j += std::string(i ? ", " : "") + "{\"row_group\": " + std::to_string(sl.rg) + ", \"num_rows\": " + std::to_string(n) + ", \"keep\": [";
    if (sl.keep) {
      for (size_t k = 0; k < sl.keep->size(); k++)
        j += std::string(k ? ", " : "") + "[" + std::to_string((*sl.keep)[k].first) + ", " + std::to_string((*sl.keep)[k].second) + "]";
    } else {
      j += "[0, " + std::to_string(n) + "]";
    }
    j += "]}";
  }
  return j + "]}";
}

// Diagnostic (host only, no device): the PLAIN value bytes the host stages for one column of a NativeScan — every selected row group,
// page after page, NULLs left out; BYTE_ARRAY values as 4-byte length + bytes.  Pages whose values reach the device as dictionary indices
// are refused.  What the CPU tests read to check the host-side rewriting of DELTA_* / BYTE_STREAM_SPLIT pages against pyarrow.
std::vector<uint8_t> parquet_host_plain_values(const Operator& op, size_t col) {
  if (col >= op.required_schema.size()) throw CometError("parquet_host_plain_values: no such column");
  ScanOptions so = ScanOptions::of(op);
  so.device_snappy = false;
  std::vector<Sel> sels;
  int64_t total = 0, rg_pruned = 0, rows_pruned = 0;
  select_row_groups(op, false, sels, total, rg_pruned, rows_pruned);
  std::vector<uint8_t> out;
  for (const Sel& sl : sels) {
    ColumnPlan cp = plan_column(op.required_schema[col], *sl.meta, so);
    if (cp.missing) continue;
    const pq::ColumnMeta& cm = sl.meta->row_groups[(size_t)sl.rg].columns[(size_t)cp.leaf];
    ChunkSource src{sl.file.get(), sl.meta.get(), sl.rg, nullptr};
    std::vector<uint8_t> staged(chunk_staging_capacity(src, cm, cp.el.repetition == 1 ? 1 : 0, cp.is_string, so) + 64);
    HostChunk hc;
    decode_chunk_host(src, op.required_schema[col], so, hc, staged.data(), staged.size());
    for (const PqPage& pg : hc.pages) {
      if (pg.encoding != 0) throw CometError("parquet_host_plain_values: the column has dictionary-encoded pages");
      if (hc.cp.is_string) {
        for (int32_t k = 0; k < pg.value_count; k++) {
          const int64_t off = hc.str_offs[(size_t)(pg.str_first + k)];
          uint32_t len;
          memcpy(&len, staged.data() + off - 4, 4);
          out.insert(out.end(), staged.data() + off - 4, staged.data() + off + len);
        }
      } else {
        const uint8_t* v = staged.data() + pg.values_off;
        out.insert(out.end(), v, v + (size_t)pg.value_count * (size_t)pg.width);
      }
    }
  }
  return out;
}

// COMET_TRACE_STAGES lines carry the scan's number: eight concurrent tasks write into one stderr
static std::atomic<int> g_scan_seq{0};
static void trace_line(int scan, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  fprintf(stderr, "[comet] parquet#%d: %s", scan, buf);
}

DevTable ExecutionContext::scan_parquet(const Operator& op) {
  static const bool trace = getenv("COMET_TRACE_STAGES") != nullptr;
  const int scan_id = g_scan_seq.fetch_add(1);
  int64_t miss0[4] = {0, 0, 0, 0};
  if (trace) pool_miss_counters(miss0);
  const auto t_begin = std::chrono::steady_clock::now();
  auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
  // The scan works on LEAVES: a top-level column of a flat type is its own leaf; a struct column is one leaf per field, a list column the one
  // leaf of its elements (one level of nesting: parquet_support.rs:249-383 converts structs, lists and maps of any depth).  Every leaf runs
  // through the same machinery — its entries (rows; a list leaf: its level entries) are decoded into values + validity — and a nested
  // column is assembled from its leaves' values and LEVELS at the end (pq_levels_kernel and the assembly kernels in parquet_kernels.hip).
  const size_t ntop = op.required_schema.size();
  std::vector<StructField> fields;               // the leaves
  struct TopCol { int kind = 0; size_t first = 0, count = 0; };      // kind 0 flat, 1 struct, 2 list
  std::vector<TopCol> tops(ntop);
  std::vector<int> top_of;                       // leaf → its top-level column
  for (size_t t = 0; t < ntop; t++) {
    const StructField& f = op.required_schema[t];
    tops[t].first = fields.size();
    if (f.dtype.id == TypeId::Struct) {
      tops[t].kind = 1;
      if (f.dtype.kids.empty()) throw CometError("Parquet column '" + f.name + "': a struct without fields");
      for (size_t k = 0; k < f.dtype.kids.size(); k++) {
        if (f.dtype.kids[k].is_nested()) throw CometError("Parquet column '" + f.name + "': nesting deeper than one level is not supported by the GPU scan yet");
        StructField lf;
        lf.name = k < f.dtype.kid_names.size() ? f.dtype.kid_names[k] : std::string();
        lf.dtype = f.dtype.kids[k];
        lf.nullable = k < f.dtype.kid_nullable.size() ? f.dtype.kid_nullable[k] != 0 : true;
        lf.nest = 1;
        lf.parent = f.name;
        lf.parent_field_id = f.field_id;
        fields.push_back(lf);
        top_of.push_back((int)t);
      }
    } else if (f.dtype.id == TypeId::List || f.dtype.id == TypeId::Map) {
      // (a map is a list of (key, value) entry structs — in the file: <rep> group m (MAP) { repeated group key_value { required key; <rep> value } } —
      // and in HBM: the same offsets + entries layout, Arrow's)
      tops[t].kind = 2;
      if (f.dtype.kids.size() != 1) throw CometError("Parquet column '" + f.name + "': a list without an element type");
      const DType& el = f.dtype.kids[0];
      if (f.dtype.id == TypeId::Map && (el.id != TypeId::Struct || el.kids.size() != 2)) throw CometError("Parquet column '" + f.name + "': a map type without (key, value) entries");
      if (el.id == TypeId::Struct && !el.kids.empty()) {
        // a list of structs: one leaf per field of the element struct, all under the same repeated group (the same repetition levels)
        tops[t].kind = 3;
        for (size_t k = 0; k < el.kids.size(); k++) {
          if (el.kids[k].is_nested()) throw CometError("Parquet column '" + f.name + "': nesting deeper than a list of flat structs is not supported by the GPU scan yet");
          StructField lf;
          lf.name = k < el.kid_names.size() ? el.kid_names[k] : std::string();
          lf.dtype = el.kids[k];
          lf.nest = 3;
          lf.parent = f.name;
          lf.parent_field_id = f.field_id;
          fields.push_back(lf);
          top_of.push_back((int)t);
        }
        tops[t].count = fields.size() - tops[t].first;
        continue;
      }
      if (el.is_nested()) throw CometError("Parquet column '" + f.name + "': lists of " + el.str() + " are not supported by the GPU scan yet (lists of flat types and of flat structs are)");
      StructField lf;
      lf.name = "element";
      lf.dtype = el;
      lf.nest = 2;
      lf.parent = f.name;
      lf.parent_field_id = f.field_id;
      fields.push_back(lf);
      top_of.push_back((int)t);
    } else if (f.dtype.is_nested()) {
      throw CometError("Parquet column '" + f.name + "': " + f.dtype.str() + " columns are not supported by the GPU scan yet");
    } else {
      fields.push_back(f);
      top_of.push_back((int)t);
    }
    tops[t].count = fields.size() - tops[t].first;
  }
  bool any_nested = false;
  for (auto& tc : tops) any_nested |= tc.kind != 0;
  const size_t ncol = fields.size();
  const size_t npart = op.partition_schema.size();
  DevTable out, lf_out;                          // lf_out: the leaves' columns (types / cols / has_valid), out: what the scan returns
  for (auto& f : op.required_schema) out.types.push_back(f.dtype);
  for (auto& f : op.partition_schema) out.types.push_back(f.dtype);
  out.cols.assign(ntop + npart, DeviceColumnView());
  out.has_valid.assign(ntop + npart, false);
  for (auto& f : fields) lf_out.types.push_back(f.dtype);
  lf_out.cols.assign(ncol, DeviceColumnView());
  lf_out.has_valid.assign(ncol, false);
  if (op.files.empty()) return out;   // EmptyExec (planner.rs:1548-1556)
  if (op.encryption_enabled) throw CometError("Parquet modular encryption is not supported by the GPU scan");
  ScanOptions so = ScanOptions::of(op);
  if (const char* e = getenv("COMET_DEVICE_DECOMPRESS")) so.device_snappy_mode = !strcmp(e, "auto") ? -1 : atoi(e) != 0;
  if (const char* e = getenv("COMET_DEVICE_DICT_PAGES")) so.device_dict_pages = atoi(e) != 0;
  if (const char* e = getenv("COMET_PARQUET_READ_IN_PLACE")) so.read_in_place = atoi(e) != 0;
  if (const char* e = getenv("COMET_DEVICE_ZSTD")) so.device_zstd = atoi(e) != 0;
  if (const char* e = getenv("COMET_DEVICE_ZSTD_DICT")) so.device_zstd_dict = atoi(e) != 0;
  if (const char* e = getenv("COMET_DEVICE_RUNS")) so.device_runs = atoi(e) != 0;
  for (auto& kv : config_)
    if (kv.first == "spark.comet.gpu.scan.deviceDecompress") so.device_snappy_mode = kv.second == "auto" ? -1 : (kv.second != "false" && kv.second != "0");
  if (op.default_values.size() != op.default_values_indexes.size()) throw CometError("NativeScan: default_values and default_values_indexes differ in length");
  auto default_of = [&](size_t leaf) -> const Expr* {
    if (fields[leaf].nest != 0) return nullptr;      // (default values belong to top-level columns)
    const size_t c = (size_t)top_of[leaf];
    for (size_t k = 0; k < op.default_values_indexes.size(); k++)
      if ((size_t)op.default_values_indexes[k] == c) return op.default_values[k].get();
    return nullptr;
  };

  bool page_index = true;
  if (const char* e = getenv("COMET_PARQUET_PAGE_INDEX")) page_index = atoi(e) != 0;
  for (auto& kv : config_)
    if (kv.first == "spark.comet.gpu.scan.pageIndex" || kv.first == "spark.sql.parquet.columnindex.access.enabled") page_index = kv.second != "false" && kv.second != "0";
  // pass 1: open files, pick row groups (midpoint rule), prune by statistics and page index, total rows
  std::vector<Sel> sels;
  int64_t total_rows = 0, rg_pruned = 0, rows_pruned = 0;
  if (any_nested) page_index = false;            // (a kept row range does not say which ENTRIES of a list leaf it covers)
  // datafusion.execution.parquet.bloom_filter_on_read (default true) reaches the reference's scan through spark.comet.datafusion.* (jni_api.rs:611-620, parquet_exec.rs:251-252)
  bool bloom_filters = true;
  int64_t rg_bloom = 0;
  if (const char* e = getenv("COMET_PARQUET_BLOOM_FILTER")) bloom_filters = atoi(e) != 0;
  for (auto& kv : config_)
    if (kv.first == "spark.comet.datafusion.execution.parquet.bloom_filter_on_read") bloom_filters = kv.second != "false" && kv.second != "0";
  select_row_groups(op, page_index, sels, total_rows, rg_pruned, rows_pruned, bloom_filters, &rg_bloom);
  row_groups_pruned_ += rg_pruned;
  row_groups_pruned_bloom_ += rg_bloom;
  rows_pruned_page_index_ += rows_pruned;
  out.rows = total_rows;
  bytes_scanned_ = 0;
  if (total_rows == 0) return out;
  if (total_rows >= ((int64_t)1 << 31)) throw CometError("GPU Parquet scan: more than 2^31 rows in one partition");

  if (trace) trace_line(scan_id, "scan began at %.2f ms of the process clock; footers + row-group selection done at %.2f ms\n", process_clock_ms() - ms_since(), ms_since());
  // Host threads prepare the column chunks (decompression dominates: ~1 GB/s per core for zstd) straight into one pinned
  // block per column; this thread concatenates a finished column's tables, uploads and decodes the whole column at once.
  // spark.comet.gpu.scanThreads / COMET_SCAN_THREADS bound the pool.
  const size_t nsel = sels.size();
  const size_t ntasks = ncol * nsel;
  std::vector<HostChunk> chunks(ntasks);
  std::vector<ColumnPlan> plans(ncol);            // what the column looks like in the files that have it (string-ness, output width)
  std::vector<char> chunk_missing(ntasks, 0);     // per (column, row group): this file lacks the column
  std::vector<char> all_missing(ncol, 0);         // no selected file has the column and it has no default: all-NULL fast path
  std::vector<std::vector<size_t>> slot_off(ncol, std::vector<size_t>(nsel + 1, 0));
  // raw areas (chunks read in place): raw_off[c][si] from raw_base[c], which lies behind the column's last staging slot
  std::vector<std::vector<size_t>> raw_off(ncol, std::vector<size_t>(nsel + 1, 0));
  std::vector<size_t> raw_base(ncol, 0);
  std::vector<std::unique_ptr<PinnedBuf>> col_staged(ncol);
  // a leaf's ENTRIES: its rows — or, the element leaf of a list, its level entries (ColumnMetaData.num_values); per row group where they start
  std::vector<std::vector<int64_t>> ent_off(ncol, std::vector<int64_t>(nsel + 1, 0));
  std::vector<int64_t> ent_total(ncol, 0);
  for (size_t c = 0; c < ncol; c++) {
    bool have = false;
    const Expr* dflt = default_of(c);
    for (size_t si = 0; si < nsel; si++) {
      ColumnPlan cp = plan_column(fields[c], *sels[si].meta, so);
      const pq::RowGroup& rg = sels[si].meta->row_groups[(size_t)sels[si].rg];
      if (!cp.missing && (size_t)cp.leaf >= rg.columns.size()) throw CometError("parquet: column index out of range");
      ent_off[c][si + 1] = ent_off[c][si] + ((!cp.missing && cp.max_rep > 0) ? rg.columns[(size_t)cp.leaf].num_values : sels[si].rows);
      chunk_missing[c * nsel + si] = cp.missing;
      if (!cp.missing && !have) { plans[c] = cp; have = true; }
      size_t cap_bytes, raw_bytes = 0;
      if (cp.missing) {
        cap_bytes = synth_capacity(fields[c].dtype, rg.num_rows);
      } else {
        ChunkSource csrc{sels[si].file.get(), sels[si].meta.get(), sels[si].rg, nullptr};
        cap_bytes = chunk_staging_capacity(csrc, rg.columns[(size_t)cp.leaf], cp.el.repetition == 1 ? 1 : 0, cp.is_string, so);   // reads the chunk only if it holds DELTA_BYTE_ARRAY pages
        if (in_place_shape(rg.columns[(size_t)cp.leaf], cp.is_string || cp.nested_leaf(), so)) raw_bytes = in_place_extra(rg.columns[(size_t)cp.leaf]);
      }
      slot_off[c][si + 1] = slot_off[c][si] + cap_bytes;
      raw_off[c][si + 1] = raw_off[c][si] + raw_bytes;
    }
    raw_base[c] = (slot_off[c][nsel] + 64 + 63) & ~(size_t)63;
    if (!have) {
      plans[c] = plan_column(fields[c], *sels[0].meta, so);
      all_missing[c] = dflt == nullptr || dflt->lit_null;
    }
    plans[c].out_width = out_width_of(fields[c].dtype);
    ent_total[c] = ent_off[c][nsel];
    if (ent_total[c] >= ((int64_t)1 << 31)) throw CometError("GPU Parquet scan: more than 2^31 list elements in one partition");
    col_staged[c].reset(new PinnedBuf());
    col_staged[c]->ensure(raw_base[c] + raw_off[c][nsel] + 64);
  }
  // shared state outlives this frame only through the shared_ptr the tasks hold
  struct Progress {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> done;
    size_t finished = 0;
    std::atomic<bool> cancelled{false};
    std::atomic<size_t> next{0};
  };
  auto prog = std::make_shared<Progress>();
  prog->done.assign(ntasks, 0);
  int max_inflight = ScanPool::get().size();
  for (auto& kv : config_)
    if (kv.first == "spark.comet.gpu.scanThreads") max_inflight = std::max(1, atoi(kv.second.c_str()));
  const int host_threads = std::max(1, std::min(max_inflight, ScanPool::get().size()));
  // A LONE task with a few scan threads (a Spark task owns one core; nothing else executes in the process) leaves to the device whatever the
  // device can do: the index sections of dictionary-encoded pages are inflated there whatever the codec, and their run headers are walked there
  // (SF10 Q6 with one scan thread: zstd 84 → 39 ms, the host no longer inflates 165 MB of index sections; snappy 25 ms, the host no longer
  // looks through every compressed page for its run headers — ≈ 6 µs a page).  Not so a scan with a dozen threads — they are idle anyway, and a
  // column whose run table the device sizes is decoded one host round trip later (zstd 16.9 against 15.0 ms) — and not so a task that is one
  // of many: eight tasks at once have sixteen cores between them and ONE device, whose decompression kernels are then the bound (zstd: a kernel
  // running 0.81 of the wave); measured over the SF10 Q6 file, eight / sixteen tasks: 19.1 / 25.5 ms (snappy) and 25.5 / 28.7 ms (zstd) with the
  // host doing its part, 22.7 / 32.4 and 31.2 / 32.5 ms with the device taking everything (profiles/r5_executor_shape.md).
  static const int kFewThreads = getenv("COMET_PQ_FEW_THREADS") ? atoi(getenv("COMET_PQ_FEW_THREADS")) : 4;
  static const int kLonePlans = getenv("COMET_PQ_LONE_PLANS") ? atoi(getenv("COMET_PQ_LONE_PLANS")) : 2;
  const bool few_threads = host_threads <= kFewThreads && detail::plans_executing() <= kLonePlans;
  if (getenv("COMET_DEVICE_ZSTD_DICT") == nullptr) so.device_zstd_dict = few_threads;
  so.device_runs_snappy = few_threads;
  if (const char* e = getenv("COMET_DEVICE_RUNS_SNAPPY")) so.device_runs_snappy = atoi(e) != 0;
  // Columns are taken largest first: the big PLAIN columns are the ones whose pages the device decompresses, and that kernel then runs
  // while the host threads are still preparing the small (dictionary-encoded) columns.
  std::vector<size_t> order(ncol);
  std::vector<int64_t> col_bytes(ncol, 0);
  int64_t plain_snappy_bytes = 0;     // uncompressed bytes of chunks that are snappy, fixed-width and (by bytes per value) mostly PLAIN
  int64_t plain_zstd_bytes = 0;       // … zstd, fixed-width, mostly PLAIN
  for (size_t c = 0; c < ncol; c++) {
    order[c] = c;
    for (size_t si = 0; si < nsel; si++) {
      if (chunk_missing[c * nsel + si]) continue;
      ColumnPlan cp = plan_column(fields[c], *sels[si].meta, so);
      const pq::ColumnMeta& cm = sels[si].meta->row_groups[(size_t)sels[si].rg].columns[(size_t)cp.leaf];
      col_bytes[c] += cm.total_compressed;
      // snappy chunks of fixed-width columns are what the device inflates: PLAIN pages, and dictionary-encoded pages whose run headers the
      // host reads through the compressed stream
      if (cm.codec == pq::SNAPPY && !cp.is_string && !cp.nested_leaf() && cp.src_width > 0 && cm.num_values > 0 &&
          (so.device_dict_pages || (double)cm.total_uncompressed >= 0.75 * (double)cp.src_width * (double)cm.num_values))
        plain_snappy_bytes += cm.total_uncompressed;
      // … and zstd chunks: PLAIN pages, and — when the chunk is read whole — dictionary-encoded pages (index sections come back for their run headers)
      if (cm.codec == pq::ZSTD && so.device_zstd && !cp.is_string && !cp.nested_leaf() && cp.src_width > 0 && cm.num_values > 0 &&
          ((so.device_zstd_dict && sels[si].keep == nullptr) || (double)cm.total_uncompressed >= 0.75 * (double)cp.src_width * (double)cm.num_values))
        plain_zstd_bytes += cm.total_uncompressed;
    }
  }
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return col_bytes[a] > col_bytes[b]; });
  // COMET_PQ_ORDER=small_first: the small columns' chunks are prepared and sent first (their slices cross in latency-bound copies that then
  // overlap the big column's host work), the big column last; =big_first (default) starts the device pipeline of the big column as early as possible
  // A lone few-thread task takes its small columns first: a column whose run table the device sizes is finished one host round trip after its
  // pages are inflated, and with the big column last those round trips happen WHILE the big column is read and crosses.
  static const char* order_env = getenv("COMET_PQ_ORDER");
  const bool small_first = order_env ? !strcmp(order_env, "small_first") : few_threads;
  if (small_first) std::reverse(order.begin(), order.end());
  // Measured on MI355X (profiles/r3_snappy_pipeline.json): the multi-kernel pipeline inflates PLAIN pages at 70–80 GB/s of output whatever
  // their number (it parallelises inside the pages), plus about half a millisecond of launches; a host core decompresses the same bytes at
  // ~1 GB/s.  Only a small scan on a host with many idle threads is better off on the host.
  // zstd (profiles/r3_zstd_pipeline.json): the sequence decoder is one scalar lane per 128 KiB block — what bounds it is that lane's
  // instruction count, not the number of blocks — so the device inflates at a rate that a host with a dozen idle cores matches; a Spark
  // task, which owns one core, is better off on the device by a wide margin.
  if (so.device_snappy_mode >= 0) {
    so.device_snappy = so.device_snappy_mode != 0;
  } else {
    // the two codecs decide separately: snappy pages by the pipeline's rate against ~1 GB/s per host thread, zstd pages by the model above
    const double T = (double)host_threads, sb = (double)plain_snappy_bytes, zb = (double)plain_zstd_bytes;
    const bool snappy_on_device = plain_snappy_bytes > 0 && 0.5 + sb / 60e6 < sb / 1e6 / T;
    const bool zstd_on_device = plain_zstd_bytes > 0 && so.device_zstd && kDeviceZstdSetupMs + zb / kDeviceZstdBytesPerMs < zb / kHostZstdBytesPerMs / T;
    so.device_snappy = snappy_on_device || zstd_on_device;
    if (!zstd_on_device && getenv("COMET_DEVICE_ZSTD") == nullptr) so.device_zstd = false;
  }
  if (trace) trace_line(scan_id, "%.1f MB of snappy / %.1f MB of zstd pages of fixed-width columns, decompressed on the %s%s\n", (double)plain_snappy_bytes / 1e6,
                     (double)plain_zstd_bytes / 1e6, so.device_snappy ? "device" : "host", so.device_snappy && plain_zstd_bytes && !so.device_zstd ? " (zstd: host)" : "");
  auto run_task = [&](size_t t, bool raw_read) {
    const size_t c = t / nsel, si = t % nsel;
    ChunkSource src{sels[si].file.get(), sels[si].meta.get(), sels[si].rg, sels[si].keep.get()};
    if (all_missing[c]) return;
    uint8_t* slot = (uint8_t*)col_staged[c]->p + slot_off[c][si];
    const size_t cap = slot_off[c][si + 1] - slot_off[c][si];
    const size_t rcap = raw_off[c][si + 1] - raw_off[c][si];
    uint8_t* rarea = rcap ? (uint8_t*)col_staged[c]->p + raw_base[c] + raw_off[c][si] : nullptr;
    if (chunk_missing[t]) synth_chunk(fields[c], default_of(c), sels[si].rows, chunks[t], slot, cap);
    else { HostTimer tm(g_ns_chunk); decode_chunk_host(src, fields[c], so, chunks[t], slot, cap, rarea, rcap, raw_read); }
  };
  // The unit of host work is a PIECE, in the order the device consumes the bytes (columns largest first, their chunks in row order).  A chunk
  // that is read in place is read in pieces of a couple of MiB by whichever threads are free, and the thread that lands its last piece walks
  // its pages — so the chunks of a column become ready ONE AFTER THE OTHER at the rate all readers reach together (32 whole-chunk reads
  // side by side finished together: the first copy of SF10 Q6 left at 2.5 ms, with a third of the file read); every other chunk is one
  // piece (read, decompress, walk — as before).
  struct Piece { size_t t; size_t lo, len; bool whole; };
  std::vector<Piece> pieces;
  std::unique_ptr<std::atomic<int>[]> pieces_left(new std::atomic<int>[ntasks]);
  static const size_t kReadPiece = getenv("COMET_PQ_READ_PIECE") ? (size_t)std::max(64 << 10, atoi(getenv("COMET_PQ_READ_PIECE"))) : ((size_t)2 << 20);
  for (size_t oi = 0; oi < ncol; oi++)
    for (size_t si = 0; si < nsel; si++) {
      const size_t c = order[oi], t = c * nsel + si;
      pieces_left[t].store(1);
      bool split = false;
      if (!all_missing[c] && !chunk_missing[t] && raw_off[c][si + 1] > raw_off[c][si] && so.device_snappy) {
        ColumnPlan cp = plan_column(fields[c], *sels[si].meta, so);
        const pq::ColumnMeta& cm = sels[si].meta->row_groups[(size_t)sels[si].rg].columns[(size_t)cp.leaf];
        const int64_t off = chunk_file_offset(cm);
        // (a chunk that lies outside its file stays whole: decode_chunk_host says so)
        if (in_place_shape(cm, cp.is_string || cp.nested_leaf(), so) && off >= 0 && cm.total_compressed > 0 && (size_t)(off + cm.total_compressed) <= sels[si].file->size &&
            raw_off[c][si + 1] - raw_off[c][si] >= in_place_extra(cm)) {
          const size_t total = (size_t)cm.total_compressed;
          size_t np = 0;
          for (size_t lo = 0; lo < total;) {
            size_t len = std::min(kReadPiece, total - lo);
            if (total - lo - len < kReadPiece / 4) len = total - lo;      // no crumbs
            pieces.push_back(Piece{t, lo, len, false});
            lo += len;
            np++;
          }
          pieces_left[t].store((int)np);
          split = true;
        }
      }
      if (!split) pieces.push_back(Piece{t, 0, 0, true});
    }
  const size_t npieces = pieces.size();
  auto read_piece = [&](const Piece& pc) {
    const size_t c = pc.t / nsel, si = pc.t % nsel;
    ColumnPlan cp = plan_column(fields[c], *sels[si].meta, so);
    const pq::ColumnMeta& cm = sels[si].meta->row_groups[(size_t)sels[si].rg].columns[(size_t)cp.leaf];
    HostTimer tc(g_ns_chunk);
    HostTimer tm(g_ns_read);
    sels[si].file->read_at((uint8_t*)col_staged[c]->p + raw_base[c] + raw_off[c][si] + pc.lo, pc.len, chunk_file_offset(cm) + (int64_t)pc.lo);
  };
  // `max_inflight` workers take the pieces in order (spark.comet.gpu.scanThreads: a task's share of the executor's pool — one, for a
  // Spark task that owns one core), each from the shared cursor until none is left
  const size_t nworkers = std::min<size_t>(npieces, (size_t)std::max(1, std::min(max_inflight, ScanPool::get().size())));
  std::mutex piece_err_mu;
  auto process_piece = [&, prog](size_t pi) {
    const Piece& pc = pieces[pi];
    const size_t t = pc.t;
    if (!pc.whole && !prog->cancelled.load()) {
      try {
        read_piece(pc);
      } catch (...) {
        std::lock_guard<std::mutex> lk(piece_err_mu);
        if (!chunks[t].err) chunks[t].err = std::current_exception();
      }
    }
    if (pieces_left[t].fetch_sub(1) != 1) return;      // another thread lands the chunk's last piece
    if (!prog->cancelled.load() && !chunks[t].err) {
      try {
        run_task(t, !pc.whole);
      } catch (...) {
        chunks[t].err = std::current_exception();
      }
    }
    {
      std::lock_guard<std::mutex> lk(prog->mu);
      prog->done[t] = 1;
      prog->finished++;
    }
    prog->cv.notify_all();
  };
  for (size_t wk = 0; wk < nworkers; wk++) {
    ScanPool::get().submit([prog, npieces, &process_piece]() {
      for (;;) {
        const size_t pi = prog->next.fetch_add(1);
        if (pi >= npieces) break;
        process_piece(pi);
      }
    });
  }
  // whatever happens below, no task may still reference this frame when it unwinds
  struct Drain {
    std::shared_ptr<Progress> p; size_t n;
    ~Drain() {
      p->cancelled.store(true);
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv.wait(lk, [&] { return p->finished == n; });
    }
  } drain{prog, ntasks};
  // While the chunk it needs is not ready, the task's own thread READS pieces too instead of sleeping (a task with one scan thread got its
  // 62 MB of SF10 Q6 at the 8 GB/s one pread() loop reaches next to seven others: 7.5 of its 17 ms).  Only pieces of chunks that are read in
  // place — 2 MiB of pread(), at most the page-header walk of the chunk whose last piece this is; a chunk that is decompressed on the host
  // would keep this thread from the copies and launches that are waiting for it.
  static const bool help_reading = getenv("COMET_PQ_TASK_READS") == nullptr || atoi(getenv("COMET_PQ_TASK_READS")) != 0;
  // (while_waiting: what else this thread has to do between pieces — columns whose run counts have come back from the device are finished
  // there; → true while it wants to be called again soon)
  std::function<bool()> while_waiting;
  auto wait_for = [&](size_t t) {
    for (;;) {
      {
        std::lock_guard<std::mutex> lk(prog->mu);
        if (prog->done[t]) break;
      }
      const bool again = while_waiting ? while_waiting() : false;
      size_t pi = prog->next.load();
      bool took = false;
      while (help_reading && pi < npieces && !pieces[pi].whole) {
        if (prog->next.compare_exchange_weak(pi, pi + 1)) { took = true; break; }
      }
      if (took) { process_piece(pi); continue; }
      std::unique_lock<std::mutex> lk(prog->mu);
      if (again) { prog->cv.wait_for(lk, std::chrono::microseconds(150), [&] { return prog->done[t] != 0; }); continue; }
      prog->cv.wait(lk, [&] { return prog->done[t] != 0; });
      break;
    }
    if (chunks[t].err) std::rethrow_exception(chunks[t].err);
  };

  struct ColumnDevice { DevBuf bytes, tables; PinnedBuf h_tables; std::vector<std::unique_ptr<Snappy2Scratch>> snappy2; std::vector<std::unique_ptr<Zstd2Scratch>> zstd2; };
  std::vector<std::shared_ptr<ColumnDevice>> keep;
  // The chunks' slices cross PCIe as soon as they are ready, one hipMemcpyAsync each (≈ 40 µs of submission and completion latency per copy
  // whatever its size).  Two alternatives sit behind switches because they were measured and lost (SF10 Q6 from snappy Parquet, 16.6 ms
  // with one copy per slice): COMET_PQ_UPLOAD=kernel batches the ready slices into ONE launch of a kernel that reads the pinned staging
  // memory across PCIe itself (22.5 ms: the SDMA engines move a slice at 50+ GB/s, a kernel's reads of host memory reach half of that);
  // COMET_PQ_COPY_STREAMS=n spreads the copies over n streams (17–18 ms, noisier).
  // (streams and events come from the process-wide pools: creating a stream costs 10 ms and destroying one 2 ms when eight tasks do it at
  // once — hipStreamCreateWithFlags was a quarter of the wall time of eight concurrent scans, profiles/r4_executor_hip_api.txt)
  // (COMET_PQ_SHARED_COPY_STREAM=1: ONE copy stream for all scans of the process, detail::shared_copy_stream — measured and lost: a submission
  // that stalls holds every task's copies behind it, 22.7 / 32.5 ms against 19.0 / 23.2 for eight / sixteen snappy tasks)
  static const bool shared_copies = getenv("COMET_PQ_SHARED_COPY_STREAM") != nullptr && atoi(getenv("COMET_PQ_SHARED_COPY_STREAM")) != 0;
  hipStream_t copy_stream = shared_copies ? detail::shared_copy_stream(device_id_) : detail::pool_get_stream(device_id_);
  std::vector<hipStream_t> extra_streams;
  std::vector<hipEvent_t> events;
  PinnedBuf upload_descs;
  struct StreamGuard {
    int dev; hipStream_t& s; std::vector<hipStream_t>& extra; std::vector<hipEvent_t>& ev; bool shared;
    ~StreamGuard() {
      for (hipStream_t x : extra) { (void)hipStreamSynchronize(x); detail::pool_put_stream(dev, x); }
      if (s && shared) {
        // this scan's copies (read from pinned staging that is about to go back to its pool) are behind an event of its own; other scans' later
        // copies are none of its business
        hipEvent_t e = detail::pool_get_event(dev);
        if (hipEventRecord(e, s) == hipSuccess) (void)hipEventSynchronize(e);
        detail::pool_put_event(dev, e);
      } else if (s) { (void)hipStreamSynchronize(s); detail::pool_put_stream(dev, s); }
      for (hipEvent_t e : ev) detail::pool_put_event(dev, e);
    }
  } stream_guard{device_id_, copy_stream, extra_streams, events, shared_copies};
  auto get_event = [&]() {
    hipEvent_t e = detail::pool_get_event(device_id_);
    events.push_back(e);
    return e;
  };
  static const bool upload_by_kernel = getenv("COMET_PQ_UPLOAD") && !strcmp(getenv("COMET_PQ_UPLOAD"), "kernel");
  static const int n_copy_streams = getenv("COMET_PQ_COPY_STREAMS") ? std::max(1, std::min(8, atoi(getenv("COMET_PQ_COPY_STREAMS")))) : 1;
  for (int k = 1; k < n_copy_streams && !upload_by_kernel; k++) {
    extra_streams.push_back(detail::pool_get_stream(device_id_));
  }
  const size_t max_descs = ntasks * 2 + ncol + 16;
  upload_descs.ensure(max_descs * sizeof(PqCopyDesc) + 64);
  size_t descs_used = 0, batch_first = 0, rr = 0;
  std::vector<char> stream_dirty(extra_streams.size() + 1, 0);
  auto upload = [&](void* dst, const void* src, size_t len) {
    if (!len) return;
    if (upload_by_kernel) {
      if (descs_used >= max_descs) throw CometError("internal: upload descriptor array too small");
      PqCopyDesc& d = ((PqCopyDesc*)upload_descs.p)[descs_used++];
      d.src = (const uint8_t*)src;
      d.dst = (uint8_t*)dst;
      d.len = (uint64_t)len;
    } else {
      const size_t k = rr++ % (extra_streams.size() + 1);
      const double t_copy = trace ? ms_since() : 0;
      HIP_CHECK(hipMemcpyAsync(dst, src, len, hipMemcpyHostToDevice, k ? extra_streams[k - 1] : copy_stream));
      if (trace && ms_since() - t_copy > 0.3) trace_line(scan_id, "hipMemcpyAsync of %.2f MB held this thread %.2f ms (from %.2f ms)\n", (double)len / 1e6, ms_since() - t_copy, t_copy);
      stream_dirty[k] = 1;
    }
  };
  // everything queued so far is on its way (kernel path: one launch for the batch)
  auto upload_flush = [&]() {
    if (upload_by_kernel && descs_used > batch_first) {
      pq_launch_upload((const PqCopyDesc*)upload_descs.p + batch_first, (int)(descs_used - batch_first), copy_stream);
      batch_first = descs_used;
    }
  };
  // `waiter` runs behind every upload queued so far
  auto upload_fence = [&](hipStream_t waiter) {
    upload_flush();
    hipEvent_t e = get_event();
    HIP_CHECK(hipEventRecord(e, copy_stream));
    HIP_CHECK(hipStreamWaitEvent(waiter, e, 0));
    for (size_t k = 0; k < extra_streams.size(); k++)
      if (stream_dirty[k + 1]) {
        hipEvent_t x = get_event();
        HIP_CHECK(hipEventRecord(x, extra_streams[k]));
        HIP_CHECK(hipStreamWaitEvent(waiter, x, 0));
        stream_dirty[k + 1] = 0;
      }
  };
  // The decompression pipelines of a column's chunk groups run on streams of their own, taken in turn: a group's kernels are a chain
  // (zstd: the sequence kernel lasts as long as ONE block's serial chain whatever the number of blocks — two groups side by side take no
  // longer than one; snappy: the one-lane-per-page hop kernel and the tails of the others leave most of the GPU idle), so the next group's
  // first kernels run under the previous group's last ones.  The column's decode kernels wait for every group (join_groups).
  // (… while few plans execute.  With several at once the other tasks' kernels fill those gaps, and every extra stream is a hardware queue
  // the tasks compete for — see detail::shared_copy_stream: the groups then run on the plan's own stream)
  static const int group_streams_env = getenv("COMET_PQ_GROUP_STREAMS") ? std::max(0, std::min(8, atoi(getenv("COMET_PQ_GROUP_STREAMS")))) : -1;
  const int n_group_streams = group_streams_env >= 0 ? group_streams_env : detail::plans_executing() >= 3 ? 0 : 2;
  std::vector<hipStream_t> group_streams;
  std::vector<hipEvent_t> group_events;
  std::vector<char> group_dirty;
  struct GroupStreamGuard {
    int dev; std::vector<hipStream_t>& gs;
    ~GroupStreamGuard() { for (hipStream_t x : gs) { (void)hipStreamSynchronize(x); detail::pool_put_stream(dev, x); } }
  } group_stream_guard{device_id_, group_streams};
  size_t group_rr = 0;
  hipEvent_t groups_may_start = nullptr;      // recorded on stream_ behind the set-up the groups depend on (the error words' memset)
  auto next_group_stream = [&]() -> hipStream_t {
    if (n_group_streams == 0) return stream_;
    const size_t k = group_rr++ % (size_t)n_group_streams;
    if (k >= group_streams.size()) {
      hipStream_t x = detail::pool_get_stream(device_id_);
      group_streams.push_back(x);
      group_dirty.push_back(0);
      HIP_CHECK(hipStreamWaitEvent(x, groups_may_start, 0));
    }
    group_dirty[k] = 1;
    return group_streams[k];
  };
  auto join_groups = [&]() {                  // stream_ runs behind every group launched so far
    for (size_t k = 0; k < group_streams.size(); k++)
      if (group_dirty[k]) {
        hipEvent_t e = get_event();
        HIP_CHECK(hipEventRecord(e, group_streams[k]));
        HIP_CHECK(hipStreamWaitEvent(stream_, e, 0));
        group_dirty[k] = 0;
      }
  };
  auto tiles = std::make_shared<DevBuf>();
  int64_t max_entries = total_rows;
  for (size_t c = 0; c < ncol; c++) max_entries = std::max(max_entries, ent_total[c]);
  tiles->ensure((size_t)((max_entries + 1023) / 1024 + 2) * 8);
  // one word per column: first failing page of the device decompression (job << 8 | code), 0 = fine
  auto inflate_err = std::make_shared<DevBuf>();
  inflate_err->ensure(ncol * 4 + 16);
  HIP_CHECK(hipMemsetAsync(inflate_err->p, 0, ncol * 4 + 16, stream_));
  groups_may_start = get_event();
  HIP_CHECK(hipEventRecord(groups_may_start, stream_));
  auto vidx = std::make_shared<DevBuf>();

  // everything behind a column's uploads: its tables (pages, runs, dictionaries) assembled and sent, the decode kernels queued.  A column
  // with dictionary-encoded pages the DEVICE inflates comes here late — their run headers are read back first (below).
  static const bool one_wave_snappy = getenv("COMET_SNAPPY_ONE_WAVE") != nullptr && atoi(getenv("COMET_SNAPPY_ONE_WAVE")) != 0;
  // (`dr`: the column has pages whose run headers the device walks — their descriptors, the prefix sum of their run counts and the total are there)
  struct DeviceRuns { std::shared_ptr<DevBuf> pend, offsets; int npend = 0; int64_t total = 0; };
  // the LEVELS of nested leaves, one byte per entry (finish_column fills them; the assembly at the end reads them)
  std::vector<std::shared_ptr<DevBuf>> leaf_def(ncol), leaf_rep(ncol);
  std::vector<std::shared_ptr<DevBuf>> leaf_raw_values(ncol);      // a list leaf: its values over ENTRIES (the elements are compacted out of them)
  auto finish_column = [&](const size_t c, const std::shared_ptr<ColumnDevice>& cd, const size_t S, const bool may_inflate, const DeviceRuns* dr) {
    const ColumnPlan& cp = plans[c];
    const bool is_string = cp.is_string;
    const int64_t nrows = ent_total[c];           // the leaf's entries (a list's element leaf: more than the scan has rows)
    auto values = std::make_shared<DevBuf>();
    auto valid_bytes = std::make_shared<DevBuf>();
    auto lengths = std::make_shared<DevBuf>();
    bool any_optional = false;
    size_t n_pages = 0, n_def = 0, n_idx = 0, n_dict = 0, n_doffs = 0, n_soffs = 0, n_jobs = 0, n_zjobs = 0, n_rep = 0;
    for (size_t si = 0; si < nsel; si++) {
      HostChunk& hc = chunks[c * nsel + si];
      any_optional |= hc.max_def > 0 && (!hc.no_nulls || cp.nested_leaf());      // (a nested leaf always keeps its levels: the assembly reads them)
      n_pages += hc.pages.size();
      n_def += (hc.no_nulls && !cp.nested_leaf()) ? 0 : hc.def_runs.size();
      n_rep += hc.rep_runs.size();
      n_idx += hc.idx_runs.size();
      n_dict += (hc.dict_bytes.size() + 15) & ~(size_t)15;
      n_doffs += hc.dict_offs.size();
      n_soffs += hc.str_offs.size();
      n_jobs += hc.inflate.size();
      n_zjobs += hc.zinflate.size();
    }
    if ((n_jobs || n_zjobs) && !may_inflate) throw CometError("internal: device pages in a column without a decompression region");
    if (trace) trace_line(scan_id, "column %zu host chunks ready at %.2f ms\n", c, ms_since());
    // concatenate the chunks' tables: offsets become column-global
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t o = 0;
    const size_t off_pages = o; o = al(o + n_pages * sizeof(PqPage));
    const size_t off_def = o; o = al(o + n_def * sizeof(PqRun) + 16);
    const size_t n_idx_dev = dr ? (size_t)dr->total : 0;      // runs the device writes behind the host's
    const size_t off_idx = o; o = al(o + (n_idx + n_idx_dev) * sizeof(PqRun) + 16);
    const size_t off_dict = o; o = al(o + n_dict + 16);
    const size_t off_doffs = o; o = al(o + n_doffs * 4 + 16);
    const size_t off_soffs = o; o = al(o + n_soffs * 8 + 16);
    const size_t off_jobs = o; o = al(o + n_jobs * sizeof(PqInflate) + 16);
    const size_t off_rep = o; o = al(o + n_rep * sizeof(PqRun) + 16);
    cd->h_tables.ensure(o + 16);
    cd->tables.ensure(o + 16);
    char* tb_h = (char*)cd->h_tables.p;
    bool runs_kernel_ok = true;
    {
      PqPage* P = (PqPage*)(tb_h + off_pages);
      PqRun* D = (PqRun*)(tb_h + off_def);
      PqRun* I = (PqRun*)(tb_h + off_idx);
      uint8_t* DB = (uint8_t*)(tb_h + off_dict);
      int32_t* DO = (int32_t*)(tb_h + off_doffs);
      int64_t* SO = (int64_t*)(tb_h + off_soffs);
      PqInflate* J = (PqInflate*)(tb_h + off_jobs);
      PqRun* RP = (PqRun*)(tb_h + off_rep);
      size_t ip = 0, id = 0, ii = 0, idb = 0, ido = 0, iso = 0, ij = 0, irp = 0;
      runs_kernel_ok = true;
      for (size_t si = 0; si < nsel; si++) {
        HostChunk& hc = chunks[c * nsel + si];
        const int64_t base = (int64_t)slot_off[c][si];
        // an offset into the chunk's slot of the staged region, or (flagged) of the device-decompressed region behind it
        auto global_off = [&](int64_t v) { return (v & kInflatedBit) ? (v & ~kInflatedBit) + base + (int64_t)S : v + base; };
        const bool nulls = hc.max_def > 0 && (!hc.no_nulls || cp.nested_leaf());
        for (const PqInflate& src : hc.inflate) {
          PqInflate job = src;
          job.src_off += base;
          job.dst_off += base + (int64_t)S;
          J[ij++] = job;
        }
        for (const PqPage& src : hc.pages) {
          PqPage pg = src;
          pg.row_start += ent_off[c][si];
          pg.rep_run_first += (int32_t)irp;
          pg.values_off = global_off(pg.values_off);
          pg.str_first += (int64_t)iso;
          if (nulls) pg.def_run_first += (int32_t)id;
          else pg.def_run_first = pg.def_run_count = 0;
          pg.idx_run_first += (int32_t)ii;
          // (a dictionary the device inflated sits in the column's byte buffer: addressed from the dictionary table's base like the others)
          pg.dict_off = hc.dev_dict >= 0 ? (int64_t)((char*)cd->bytes.p + base + (int64_t)S + hc.dev_dict - ((char*)cd->tables.p + off_dict)) : (int64_t)idb;
          pg.dict_offs_first = (int32_t)ido;
          P[ip++] = pg;
        }
        if (nulls)
          for (const PqRun& r : hc.def_runs) { D[id] = r; D[id].byte_off = global_off(r.byte_off); id++; }
        for (const PqRun& r : hc.idx_runs) { I[ii] = r; I[ii].byte_off = global_off(r.byte_off); ii++; }
        for (const PqRun& r : hc.rep_runs) { RP[irp] = r; RP[irp].byte_off = global_off(r.byte_off); irp++; }
        // every index run (and PLAIN chunk) learns its page: the run-at-a-time kernel starts from the run
        for (size_t gp = ip - hc.pages.size(); gp < ip; gp++)
          for (int32_t r = P[gp].idx_run_first; r < P[gp].idx_run_first + P[gp].idx_run_count; r++) I[r].page = (int32_t)gp;
        // (pieces of a pruned page have their own clipped units unless COMET_PQ_DECODE_ROWS keeps the shared runs and the row-at-a-time kernel)
        runs_kernel_ok &= sels[si].keep == nullptr || getenv("COMET_PQ_DECODE_ROWS") == nullptr;
        if (!hc.dict_bytes.empty()) memcpy(DB + idb, hc.dict_bytes.data(), hc.dict_bytes.size());
        idb += (hc.dict_bytes.size() + 15) & ~(size_t)15;
        if (!hc.dict_offs.empty()) memcpy(DO + ido, hc.dict_offs.data(), hc.dict_offs.size() * 4);
        ido += hc.dict_offs.size();
        for (int64_t v : hc.str_offs) SO[iso++] = v ? v + base : 0;
      }
      if (n_pages >= ((size_t)1 << 31) || n_idx + n_idx_dev >= ((size_t)1 << 31)) throw CometError("parquet: too many pages / runs in one column");
    }
    cd->tables.ensure(o + 16);
    upload(cd->tables.p, cd->h_tables.p, (o + 15) & ~(size_t)15);
    upload_fence(stream_);
    const char* tb = (const char*)cd->tables.p;
    if (dr && dr->npend)      // the device-walked pages' runs go behind the host's; each such page's (first, count) is filled in on the device
      pq_launch_write_runs((const PqPendingRuns*)dr->pend->p, dr->npend, (const uint8_t*)cd->bytes.p, (const int32_t*)dr->offsets->p, (int32_t)n_idx,
                           (PqRun*)((char*)cd->tables.p + off_idx), (PqPage*)((char*)cd->tables.p + off_pages), stream_);
    if (n_jobs) {
      if (n_jobs >= ((size_t)1 << 23)) throw CometError("parquet: too many pages in one column");
      // (the multi-kernel pipeline was launched group by group while the slices crossed PCIe, above)
      if (one_wave_snappy) {
        pq_launch_snappy((const PqInflate*)(tb + off_jobs), (int)n_jobs, (uint8_t*)cd->bytes.p, (uint32_t*)inflate_err->p + c, stream_);
        pages_inflated_on_device_ += (int64_t)n_jobs;
      }
    }

    PqDecodeArgs a;
    memset(&a, 0, sizeof a);
    a.pages = (const PqPage*)(tb + off_pages);
    a.npages = (int32_t)n_pages;
    a.max_def = any_optional ? std::max(cp.max_def, 1) : 0;      // (the validity kernel compares with it; every other kernel asks "> 0")
    a.rep_runs = (const PqRun*)(tb + off_rep);
    a.max_rep = cp.max_rep;
    a.def_runs = (const PqRun*)(tb + off_def);
    a.idx_runs = (const PqRun*)(tb + off_idx);
    a.bytes = (const uint8_t*)cd->bytes.p;
    a.dict = (const uint8_t*)(tb + off_dict);
    a.dict_offs = (const int32_t*)(tb + off_doffs);
    a.plain_str_offs = (const int64_t*)(tb + off_soffs);
    a.n_rows = nrows;
    a.out_width = cp.out_width;
    a.n_idx_runs = (int32_t)(n_idx + n_idx_dev);
    if (any_optional) {
      valid_bytes->ensure((size_t)nrows + 16);
      if (!vidx->p) vidx->ensure((size_t)nrows * 4 + 16);
      a.valid_out = (uint8_t*)valid_bytes->p;
      a.vidx = (uint32_t*)vidx->p;
      pq_launch_validity(&a, stream_);
      pq_launch_vidx(a.valid_out, nrows, (uint64_t*)tiles->p, a.vidx, stream_);
    }
    if (fields[c].nest != 0) {
      leaf_def[c] = std::make_shared<DevBuf>();
      leaf_def[c]->ensure((size_t)nrows + 16);
      if (any_optional) pq_launch_levels(&a, 0, (uint8_t*)leaf_def[c]->p, stream_);
      else HIP_CHECK(hipMemsetAsync(leaf_def[c]->p, cp.max_def, (size_t)nrows + 16, stream_));      // a chunk without level runs: everything defined
      out.owners.push_back(leaf_def[c]);
      if (fields[c].nest >= 2) {
        leaf_rep[c] = std::make_shared<DevBuf>();
        leaf_rep[c]->ensure((size_t)nrows + 16);
        PqDecodeArgs ar = a;
        if (!any_optional) ar.max_def = std::max(cp.max_def, 1);
        pq_launch_levels(&ar, 1, (uint8_t*)leaf_rep[c]->p, stream_);
        out.owners.push_back(leaf_rep[c]);
      }
    }
    if (!is_string) {
      values->ensure((size_t)nrows * cp.out_width + 16);
      a.values_out = values->p;
      // a column without NULLs (and without pruned pages) is decoded a RUN at a time: a wave takes one bit-packed run / RLE run / chunk of
      // a PLAIN page and every lane decodes 8 of its values with all loads in flight at once; otherwise row by row
      static const bool force_rows = getenv("COMET_PQ_DECODE_ROWS") != nullptr;
      if (runs_kernel_ok && !force_rows) {
        if (any_optional) {
          // with NULLs a page's runs hold fewer values than it has rows: the values are decoded densely by their ordinal among the
          // column's non-NULL values (vidx of the page's first row + index in the page), then spread to their rows
          auto dense = std::make_shared<DevBuf>();
          dense->ensure((size_t)nrows * cp.out_width + 16);
          a.dense_out = dense->p;
          pq_launch_decode_runs(&a, stream_);
          pq_launch_expand_nulls(&a, stream_);
          a.dense_out = nullptr;
          out.owners.push_back(dense);
        } else {
          pq_launch_decode_runs(&a, stream_);
        }
      } else {
        pq_launch_decode_fixed(&a, stream_);
      }
    } else {
      lengths->ensure((size_t)nrows * 4 + 16);
      a.lengths_out = (uint32_t*)lengths->p;
      pq_launch_string_lengths(&a, stream_);
    }
    DeviceColumnView cv;
    if (is_string) {
      auto offsets = std::make_shared<DevBuf>();
      offsets->ensure((size_t)(nrows + 1) * 4 + 16);
      pq_launch_u32_scan((const uint32_t*)lengths->p, nrows, (uint64_t*)tiles->p, (int32_t*)offsets->p, stream_);
      int32_t total_bytes = 0;
      read_small(&total_bytes, (char*)offsets->p + (size_t)nrows * 4, 4);
      auto data = std::make_shared<DevBuf>();
      data->ensure((size_t)std::max(total_bytes, 1) + 16);
      a.str_offsets = (const int32_t*)offsets->p;
      a.str_bytes_out = (uint8_t*)data->p;
      pq_launch_string_copy(&a, stream_);
      cv.data = offsets->p;
      cv.aux = data->p;
      out.owners.push_back(offsets);
      out.owners.push_back(data);
      out.owners.push_back(lengths);
    } else if (lf_out.types[c].id == TypeId::Bool) {
      auto bits = std::make_shared<DevBuf>();
      bits->ensure((size_t)((nrows + 7) / 8) + 16);
      pq_launch_pack((const uint8_t*)values->p, (uint8_t*)bits->p, nrows, stream_);
      cv.data = bits->p;
      out.owners.push_back(bits);
      out.owners.push_back(values);
    } else {
      cv.data = values->p;
      out.owners.push_back(values);
      leaf_raw_values[c] = values;
    }
    if (any_optional) {
      auto bm = std::make_shared<DevBuf>();
      bm->ensure((size_t)((nrows + 7) / 8) + 16);
      pq_launch_pack((const uint8_t*)valid_bytes->p, (uint8_t*)bm->p, nrows, stream_);
      cv.valid = (const uint8_t*)bm->p;
      lf_out.has_valid[c] = true;
      out.owners.push_back(bm);
    }
    out.owners.push_back(valid_bytes);
    lf_out.cols[c] = cv;
  };
  struct Deferred { size_t c; std::shared_ptr<ColumnDevice> cd; size_t S; bool may_inflate; std::shared_ptr<PinnedBuf> readback; hipEvent_t done; std::vector<const uint8_t*> at;
                    DeviceRuns dr; std::shared_ptr<DevBuf> counts; bool finished = false; };
  std::deque<Deferred> deferred;      // (a deque: entries are finished in place while later columns append)
  // a column whose run counts the device has delivered is finished as soon as this thread notices — between the chunks of the columns behind it
  static const bool eager_finish = getenv("COMET_PQ_EAGER_FINISH") == nullptr || atoi(getenv("COMET_PQ_EAGER_FINISH")) != 0;
  auto finish_counted = [&](Deferred& d) {
    int32_t total = 0;
    memcpy(&total, (char*)d.readback->p + (((size_t)d.dr.npend * sizeof(PqPendingRuns) + 15) & ~(size_t)15) + 16, 4);
    if (total < 0) throw CometError("parquet: too many runs in one column");
    d.dr.total = total;
    if (trace) trace_line(scan_id, "column %zu: %d runs counted on the device at %.2f ms\n", d.c, total, ms_since());
    finish_column(d.c, d.cd, d.S, d.may_inflate, &d.dr);
    out.owners.push_back(d.dr.pend);
    out.owners.push_back(d.dr.offsets);
    out.owners.push_back(d.counts);
    d.finished = true;
  };
  while_waiting = [&]() -> bool {
    if (!eager_finish) return false;
    bool open = false;
    for (Deferred& d : deferred) {
      if (d.finished || !d.dr.npend) continue;
      if (hipEventQuery(d.done) == hipSuccess) finish_counted(d);
      else open = true;
    }
    return open;
  };
  // (an error thrown while read-backs are in flight: they land in pinned staging memory that goes back to its pool — wait for them first)
  struct ReadbackGuard {
    hipStream_t s;
    const std::deque<Deferred>& d;
    bool done = false;
    ~ReadbackGuard() { if (!done && !d.empty()) (void)hipStreamSynchronize(s); }
  } readback_guard{stream_, deferred};
  for (size_t oi = 0; oi < ncol; oi++) {
    const size_t c = order[oi];
    const ColumnPlan& cp = plans[c];
    if (all_missing[c]) {
      // all-NULL column: zeroed values, zeroed validity bitmap
      DeviceColumnView mv;
      auto zeros = std::make_shared<DevBuf>();
      const size_t vb = cp.is_string ? (size_t)(total_rows + 1) * 4 : lf_out.types[c].id == TypeId::Bool ? (size_t)((total_rows + 7) / 8) : (size_t)total_rows * cp.out_width;
      zeros->ensure(vb + 16);
      HIP_CHECK(hipMemsetAsync(zeros->p, 0, vb + 16, stream_));
      auto bm = std::make_shared<DevBuf>();
      bm->ensure((size_t)((total_rows + 7) / 8) + 16);
      HIP_CHECK(hipMemsetAsync(bm->p, 0, (size_t)((total_rows + 7) / 8) + 16, stream_));
      mv.data = zeros->p;
      mv.valid = (const uint8_t*)bm->p;
      if (cp.is_string) mv.aux = zeros->p;   // no bytes are ever addressed (all offsets 0)
      lf_out.has_valid[c] = true;
      lf_out.cols[c] = mv;
      out.owners.push_back(zeros);
      out.owners.push_back(bm);
      for (size_t si = 0; si < nsel; si++) wait_for(c * nsel + si);
      continue;
    }
    auto cd = std::make_shared<ColumnDevice>();
    keep.push_back(cd);
    // the column's page bytes cross PCIe in slices as soon as their chunks are ready, on the copy stream
    // [0, S): what the host staged (decompressed pages, or compressed bodies for the device); [S, 2S): pages the device decompresses
    // (the staged region: the chunks' slots, then their raw areas — what crosses compressed, contiguous; the decompressed region mirrors the slots only)
    const size_t S = (raw_base[c] + raw_off[c][nsel] + 64 + 15) & ~(size_t)15;
    const bool may_inflate = so.device_snappy && !cp.is_string && !cp.missing;
    cd->bytes.ensure(may_inflate ? S + ((slot_off[c][nsel] + 64 + 15) & ~(size_t)15) + 64 : S);
    size_t n_jobs = 0, n_zjobs = 0;
    std::vector<PqInflate> group_jobs, zgroup_jobs;
    std::vector<comet_zstd2::ZBlock> zgroup_blocks;
    size_t group_bytes = 0, zstd_chunks_in_group = 0;
    // A copy costs ≈ 40–60 µs of submission and completion latency whatever its size (a 1 MB slice crosses in 20 µs), and one stream
    // carries them one after the other: a column of 60 small chunks spent more time between its copies than in them.  So pieces that are
    // READY and lie close together in the column's staging block cross as ONE copy — the bytes between them (a slot's unused tail) ride
    // along; a piece waits for company only while no thread would have to wait for it.
    // Two runs of pieces are open at a time: what the host staged (in the slots) and the chunks' raw areas (behind them, contiguous).
    constexpr size_t kMergeGap = (size_t)512 << 10;
    size_t pend_lo[2] = {0, 0}, pend_hi[2] = {0, 0};
    auto flush_run = [&](int w) {
      if (pend_hi[w] > pend_lo[w]) upload((char*)cd->bytes.p + pend_lo[w], (char*)col_staged[c]->p + pend_lo[w], pend_hi[w] - pend_lo[w]);
      pend_lo[w] = pend_hi[w] = 0;
    };
    auto flush_pieces = [&]() { flush_run(0); flush_run(1); };
    auto pending_piece_bytes = [&]() { return (pend_hi[0] - pend_lo[0]) + (pend_hi[1] - pend_lo[1]); };
    auto push_piece = [&](size_t lo, size_t hi) {
      if (hi <= lo) return;
      const int w = lo >= raw_base[c] ? 1 : 0;
      if (pend_hi[w] > pend_lo[w] && lo >= pend_lo[w] && lo <= pend_hi[w] + kMergeGap) { pend_hi[w] = std::max(pend_hi[w], hi); return; }
      flush_run(w);
      pend_lo[w] = lo;
      pend_hi[w] = hi;
    };
    for (size_t si = 0; si < nsel; si++) {
      {
        bool ready;
        { std::lock_guard<std::mutex> lk(prog->mu); ready = prog->done[c * nsel + si] != 0; }
        // what is ready crosses while this thread waits — once it is worth a copy: a task with one scan thread gets its chunks one by one, and a
        // hipMemcpyAsync per 0.7 MB chunk cost each of eight concurrent tasks 120 µs a call (profiles/r4_executor_hip_api.txt)
        if (!ready && pending_piece_bytes() >= ((size_t)2 << 20)) { flush_pieces(); upload_flush(); }
      }
      const double t_wait = trace ? ms_since() : 0;
      if (while_waiting) (void)while_waiting();
      wait_for(c * nsel + si);
      if (trace && ms_since() - t_wait > 0.3) trace_line(scan_id, "column %zu waited %.2f ms for chunk %zu (until %.2f ms)\n", c, ms_since() - t_wait, si, ms_since());
      HostChunk& hc = chunks[c * nsel + si];
      // is the chunk behind this one ready too?  Then its slices join this batch (one launch for all of them)
      bool next_ready = false;
      if (si + 1 < nsel) {
        std::lock_guard<std::mutex> lk(prog->mu);
        next_ready = prog->done[c * nsel + si + 1] != 0;
      }
      bytes_scanned_ += hc.compressed;
      n_jobs += hc.inflate.size();
      n_zjobs += hc.zinflate.size();
      // only the bytes the chunk actually staged cross PCIe
      if (hc.spos) push_piece(slot_off[c][si], slot_off[c][si] + std::min((hc.spos + 16 + 15) & ~(size_t)15, slot_off[c][si + 1] - slot_off[c][si]));
      if (hc.raw_hi > hc.raw_lo) {      // page bodies read in place: from where pread() put them (+ the few bytes behind the last one the kernels' vector loads touch)
        const size_t raw_end = raw_base[c] + raw_off[c][si + 1] - slot_off[c][si];      // slot-relative, like raw_lo / raw_hi
        const size_t lo = hc.raw_lo & ~(size_t)15, hi = std::min((hc.raw_hi + 32 + 15) & ~(size_t)15, raw_end);
        push_piece(slot_off[c][si] + lo, slot_off[c][si] + hi);
      }
      // Pages the device decompresses: the pipeline is launched for a GROUP of chunks as soon as their slices are across, so it runs
      // while the column's later chunks are still being read and uploaded (launched once per column it started only after the last slice:
      // 7 ms of decompression behind 10 ms of upload, SF10 Q6).  COMET_SNAPPY_ONE_WAVE=1 keeps the one-wave-per-page kernel, per column.
      if (!one_wave_snappy)
        for (const PqInflate& src : hc.inflate) {
          PqInflate job = src;
          job.src_off += (int64_t)slot_off[c][si];
          job.dst_off += (int64_t)slot_off[c][si] + (int64_t)S;
          group_jobs.push_back(job);
          group_bytes += (size_t)job.src_len;
        }
      {
        const int32_t first_block = (int32_t)zgroup_blocks.size();
        zgroup_blocks.insert(zgroup_blocks.end(), hc.zblocks.begin(), hc.zblocks.end());
        for (const PqInflate& src : hc.zinflate) {
          PqInflate job = src;
          job.src_off += (int64_t)slot_off[c][si];
          job.dst_off += (int64_t)slot_off[c][si] + (int64_t)S;
          job.preamble += first_block;
          zgroup_jobs.push_back(job);
          group_bytes += (size_t)job.src_len;
        }
      }
      // (zstd: the sequence kernel lasts as long as ONE block's serial chain — ≈ 3.4 ms for 16 K sequences — whatever the number of blocks,
      // up to the 4096 the GPU holds at once, and a stream runs its groups one behind the other: five groups of 1300 blocks on two streams
      // took three chains in a row (SF10 Q6: device idle at 20.7 ms with every launch issued by 5.3 ms).  Groups of ≈ 2800 blocks: two of
      // them fill the GPU side by side)
      // (a launch is a chain's latency however small it is: what is left of the column joins this group when it fits the GPU with it —
      // judged by the blocks per chunk seen so far)
      zstd_chunks_in_group += hc.zinflate.empty() ? 0 : 1;
      const size_t zleft_est = zstd_chunks_in_group ? (nsel - 1 - si) * zgroup_blocks.size() / zstd_chunks_in_group : 0;
      static const size_t kZGroupBlocks = getenv("COMET_PQ_ZGROUP_BLOCKS") ? (size_t)std::max(64, atoi(getenv("COMET_PQ_ZGROUP_BLOCKS"))) : 2800;
      const bool zfull = zgroup_blocks.size() >= kZGroupBlocks && zgroup_blocks.size() + zleft_est > (kZGroupBlocks == 2800 ? (size_t)4096 : kZGroupBlocks * 3 / 2);
      // (snappy: a lone few-thread task gets its big column at the rate one or two pread() loops reach — groups of 12 MB start inflating
      // while the rest is still being read; a scan with many threads keeps groups of 48 MB, fewer launches)
      static const size_t kSGroupEnv = getenv("COMET_PQ_SGROUP_MB") ? (size_t)std::max(1, atoi(getenv("COMET_PQ_SGROUP_MB"))) << 20 : 0;
      const size_t sgroup_bytes = kSGroupEnv ? kSGroupEnv : few_threads ? ((size_t)12 << 20) : ((size_t)48 << 20);
      const bool group_full = (!group_jobs.empty() || !zgroup_jobs.empty()) && ((zgroup_jobs.empty() ? group_bytes >= sgroup_bytes : zfull) || si + 1 == nsel);
      if (group_full || si + 1 == nsel) flush_pieces();
      if (!next_ready || group_full) upload_flush();
      if (group_full) {
        if (group_jobs.size() >= ((size_t)1 << 23) || zgroup_jobs.size() >= ((size_t)1 << 23)) throw CometError("parquet: too many pages in one column");
        const double t_launch = trace ? ms_since() : 0;
        // The pipelines' tables cross on the COPY stream, behind the page bytes; the group's stream is then fenced behind both and gets kernels
        // only.  (A host → device copy queued on a stream that waits for another stream's event holds the calling thread until that event has
        // happened: with the tables sent on the group's stream every launch here cost its task 7–10 ms — the time its page bytes needed to
        // cross — and eight concurrent tasks issued nothing else meanwhile: the first session of round 5, stage traces.)
        Snappy2Scratch* sn = nullptr;
        Zstd2Scratch* zs = nullptr;
        if (!group_jobs.empty()) {
          cd->snappy2.emplace_back(new Snappy2Scratch());
          sn = cd->snappy2.back().get();
          sn->stage(group_jobs.data(), (int)group_jobs.size(), copy_stream);
        }
        if (!zgroup_jobs.empty()) {
          cd->zstd2.emplace_back(new Zstd2Scratch());
          zs = cd->zstd2.back().get();
          zs->stage(zgroup_jobs.data(), (int)zgroup_jobs.size(), zgroup_blocks.data(), copy_stream);
        }
        stream_dirty[0] = 1;
        hipStream_t gs = next_group_stream();
        const double t_stream = trace ? ms_since() : 0;
        upload_fence(gs);
        const double t_fence = trace ? ms_since() : 0;
        if (sn) sn->launch((uint8_t*)cd->bytes.p, (uint32_t*)inflate_err->p + c, gs);
        if (zs) zs->launch((uint8_t*)cd->bytes.p, (uint32_t*)inflate_err->p + c, gs);
        if (trace) trace_line(scan_id, "column %zu group of %zu snappy / %zu zstd pages (%zu blocks, %.1f MB) up to chunk %zu launched at %.2f ms\n", c, group_jobs.size(),
                           zgroup_jobs.size(), zgroup_blocks.size(), (double)group_bytes / 1e6, si, ms_since());
        if (trace && ms_since() - t_launch > 0.3)
          trace_line(scan_id, "… that launch took %.2f ms of this thread (tables + stream %.2f, fence %.2f, kernels %.2f)\n", ms_since() - t_launch, t_stream - t_launch, t_fence - t_stream,
                     ms_since() - t_fence);
        pages_inflated_on_device_ += (int64_t)(group_jobs.size() + zgroup_jobs.size());
        group_jobs.clear();
        zgroup_jobs.clear();
        zgroup_blocks.clear();
        group_bytes = 0;
        zstd_chunks_in_group = 0;
      }
    }
    // Dictionary-encoded pages the device inflates (zstd: their literals are entropy-coded, the host cannot look through the compressed stream
    // as it does with snappy): the index sections come BACK once the device has inflated them — a few MB per column at PCIe speed — and the
    // host reads the run headers out of them (the decoded values never come back).  Such a column is finished behind all uploads.
    flush_pieces();
    join_groups();
    size_t pending_bytes = 0;
    for (size_t si = 0; si < nsel; si++)
      for (const HostChunk::Pending& pe : chunks[c * nsel + si].pending) pending_bytes += ((pe.end - pe.begin) + 15) & ~(size_t)15;
    if (pending_bytes && so.device_runs) {
      // The device walks the run headers where the sections lie (device/pq_runs.hpp): one descriptor per page in the coordinates of the column's
      // byte buffer, a count pass, a prefix sum — and FOUR BYTES come back (the total, which sizes the run table) instead of the sections.
      Deferred d{c, cd, S, may_inflate, std::make_shared<PinnedBuf>(), get_event(), {}, {}, nullptr};
      size_t npend = 0, page_base = 0;
      for (size_t si = 0; si < nsel; si++) npend += chunks[c * nsel + si].pending.size();
      if (npend >= ((size_t)1 << 30)) throw CometError("parquet: too many pages in one column");
      d.readback->ensure(npend * sizeof(PqPendingRuns) + 128);
      PqPendingRuns* pd = (PqPendingRuns*)d.readback->p;
      size_t k = 0;
      for (size_t si = 0; si < nsel; si++) {
        const HostChunk& hc = chunks[c * nsel + si];
        for (const HostChunk::Pending& pe : hc.pending) {
          PqPendingRuns& x = pd[k++];
          x.begin = (int64_t)(slot_off[c][si] + S + pe.begin);
          x.end = (int64_t)(slot_off[c][si] + S + pe.end);
          x.bit_width = hc.pages[pe.page].bit_width;
          x.max_values = pe.values;
          x.page = (int32_t)(page_base + pe.page);      // column-global: the chunks' pages are concatenated in row-group order (finish_column)
          x.pad = 0;
        }
        page_base += hc.pages.size();
      }
      d.dr.pend = std::make_shared<DevBuf>();
      d.dr.offsets = std::make_shared<DevBuf>();
      d.counts = std::make_shared<DevBuf>();
      d.dr.pend->ensure(npend * sizeof(PqPendingRuns) + 16);
      d.counts->ensure(npend * 4 + 16);
      d.dr.offsets->ensure((npend + 1) * 4 + 16);
      d.dr.npend = (int)npend;
      // (descriptors on the copy stream, stream_ fenced behind it: no host → device copy on a stream that waits — see the group launch above)
      upload(d.dr.pend->p, pd, (npend * sizeof(PqPendingRuns) + 15) & ~(size_t)15);
      upload_fence(stream_);
      pq_launch_count_runs((const PqPendingRuns*)d.dr.pend->p, (int)npend, (const uint8_t*)cd->bytes.p, (uint32_t*)d.counts->p, (uint32_t*)inflate_err->p + c, stream_);
      auto rtiles = std::make_shared<DevBuf>();
      rtiles->ensure((size_t)((npend + 1023) / 1024 + 2) * 8);
      pq_launch_u32_scan((const uint32_t*)d.counts->p, (int64_t)npend, (uint64_t*)rtiles->p, (int32_t*)d.dr.offsets->p, stream_);
      out.owners.push_back(rtiles);
      // the total lands behind the descriptors in the same pinned block, STORED there by a one-lane kernel (pinned host memory is device
      // addressable): a device → host copy command on this stream would hold the calling thread until the decompression before it is done
      pq_launch_store_u32((const uint32_t*)((char*)d.dr.offsets->p + npend * 4), (uint32_t*)((char*)d.readback->p + ((npend * sizeof(PqPendingRuns) + 15) & ~(size_t)15) + 16), stream_);
      HIP_CHECK(hipEventRecord(d.done, stream_));
      deferred.push_back(std::move(d));
      if (trace) trace_line(scan_id, "column %zu: the device walks the run headers of %zu pages (%.1f MB of index sections stay where they are)\n", c, npend, (double)pending_bytes / 1e6);
      continue;
    }
    if (pending_bytes) {
      // where they land: the chunk's own pinned slot, between its staged bytes and the page bodies read in place — a device-inflated chunk
      // leaves that part (sized for host-inflated pages) unused; a chunk without the room gets a buffer of its own
      Deferred d{c, cd, S, may_inflate, std::make_shared<PinnedBuf>(), get_event(), {}, {}, nullptr};
      size_t spill = 0;
      for (size_t si = 0; si < nsel; si++) {
        const HostChunk& hc = chunks[c * nsel + si];
        size_t need = 0;
        for (const HostChunk::Pending& pe : hc.pending) need += ((pe.end - pe.begin) + 15) & ~(size_t)15;
        const size_t lo = (hc.spos + 63) & ~(size_t)63, hi = slot_off[c][si + 1] - slot_off[c][si];
        if (lo + need > hi) spill += need;
      }
      if (spill) d.readback->ensure(spill + 64);
      size_t at = 0;
      for (size_t si = 0; si < nsel; si++) {
        const HostChunk& hc = chunks[c * nsel + si];
        size_t need = 0;
        for (const HostChunk::Pending& pe : hc.pending) need += ((pe.end - pe.begin) + 15) & ~(size_t)15;
        const size_t lo = (hc.spos + 63) & ~(size_t)63, hi = slot_off[c][si + 1] - slot_off[c][si];
        const bool in_slot = lo + need <= hi;
        uint8_t* to = in_slot ? (uint8_t*)col_staged[c]->p + slot_off[c][si] + lo : (uint8_t*)d.readback->p + at;
        if (!in_slot) at += need;
        for (const HostChunk::Pending& pe : hc.pending) {
          HIP_CHECK(hipMemcpyAsync(to, (char*)cd->bytes.p + slot_off[c][si] + S + pe.begin, pe.end - pe.begin, hipMemcpyDeviceToHost, stream_));
          d.at.push_back(to);
          to += ((pe.end - pe.begin) + 15) & ~(size_t)15;
        }
      }
      HIP_CHECK(hipEventRecord(d.done, stream_));
      deferred.push_back(std::move(d));
      if (trace) trace_line(scan_id, "column %zu waits for %.1f MB of index sections from the device\n", c, (double)pending_bytes / 1e6);
      continue;
    }
    finish_column(c, cd, S, may_inflate, nullptr);
  }
  if (trace && !deferred.empty()) trace_line(scan_id, "uploads of all columns issued at %.2f ms\n", ms_since());
  while_waiting = nullptr;
  for (Deferred& d : deferred) {
    if (d.finished) continue;
    HIP_CHECK(hipEventSynchronize(d.done));
    if (d.dr.npend) {
      finish_counted(d);
      continue;
    }
    if (trace) trace_line(scan_id, "column %zu index sections back at %.2f ms\n", d.c, ms_since());
    // the run headers of every pending page, parsed on the scan threads (a chunk per task), positions in the coordinates of the
    // device-decompressed region — where the decode kernels will read the indices
    std::vector<std::exception_ptr> errs(nsel);
    std::atomic<size_t> left{0};
    std::mutex mu;
    std::condition_variable cv;
    size_t k = 0;
    for (size_t si = 0; si < nsel; si++) {
      HostChunk& hc = chunks[d.c * nsel + si];
      if (hc.pending.empty()) continue;
      const size_t first = k;
      k += hc.pending.size();
      left.fetch_add(1);
      ScanPool::get().submit([&, si, first]() {
        try {
          HostChunk& h = chunks[d.c * nsel + si];
          for (size_t j = 0; j < h.pending.size(); j++) {
            const HostChunk::Pending& pe = h.pending[j];
            PqPage& pg = h.pages[pe.page];
            const uint8_t* base = d.at[first + j] - pe.begin;      // so that base + position addresses the byte
            const size_t r0 = h.idx_runs.size();
            parse_hybrid_runs(base, pe.begin, pe.end, pg.bit_width, pe.values, h.idx_runs);
            for (size_t r = r0; r < h.idx_runs.size(); r++) h.idx_runs[r].byte_off |= kInflatedBit;
            if (h.idx_runs.size() == r0) {   // page of NULLs only
              PqRun r;
              memset(&r, 0, sizeof r);
              r.is_rle = 1;
              r.count = pg.num_values;
              h.idx_runs.push_back(r);
            }
            pg.idx_run_first = (int32_t)r0;
            pg.idx_run_count = (int32_t)(h.idx_runs.size() - r0);
          }
        } catch (...) {
          errs[si] = std::current_exception();
        }
        {
          std::lock_guard<std::mutex> lk(mu);     // (notified under the lock: the waiter owns mu / cv and leaves their scope as soon as it sees zero)
          left.fetch_sub(1);
          cv.notify_all();
        }
      });
    }
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return left.load() == 0; });
    }
    for (auto& e : errs)
      if (e) std::rethrow_exception(e);
    if (trace) trace_line(scan_id, "column %zu run headers parsed at %.2f ms\n", d.c, ms_since());
    finish_column(d.c, d.cd, d.S, d.may_inflate, nullptr);
  }
  readback_guard.done = true;
  // ---- the scan's columns from their leaves: a flat column is its leaf; a struct is its fields' columns under a validity of its own; a list
  // is assembled from its element leaf's levels (rows start where the repetition level is 0, an entry holds an element slot from the
  // repeated group's definition level on) ----
  for (size_t t = 0; t < ntop; t++) {
    const TopCol& tc = tops[t];
    const size_t l0 = tc.first;
    if (tc.kind == 0) {
      out.cols[t] = lf_out.cols[l0];
      out.has_valid[t] = lf_out.has_valid[l0];
      continue;
    }
    DeviceColumnView nv;
    if (tc.kind == 3) {
      // A list of structs: its fields' leaves sit under ONE repeated group — the same repetition levels, the same element slots.  The first
      // leaf's levels give the rows' offsets and validity and the entry of every slot; every field is then TAKEN out of its leaf's column over
      // entries by those (exec.cpp take_column: any flat type), the element struct's validity is one more level comparison taken the same way.
      const ColumnPlan& cp = plans[l0];
      const int64_t n = ent_total[l0];
      for (size_t l = l0; l < l0 + tc.count; l++)
        if (ent_total[l] != n || !leaf_def[l] || !leaf_rep[l]) throw CometError("Parquet column '" + fields[l0].parent + "': the fields of its struct elements differ in their level entries");
      auto starts = std::make_shared<DevBuf>(), elems = std::make_shared<DevBuf>(), start_idx = std::make_shared<DevBuf>(), elem_idx = std::make_shared<DevBuf>();
      auto offsets = std::make_shared<DevBuf>(), lvb = std::make_shared<DevBuf>(), lbm = std::make_shared<DevBuf>(), evb = std::make_shared<DevBuf>(), entries = std::make_shared<DevBuf>();
      starts->ensure((size_t)n * 4 + 16);
      elems->ensure((size_t)n * 4 + 16);
      start_idx->ensure((size_t)(n + 1) * 4 + 16);
      elem_idx->ensure((size_t)(n + 1) * 4 + 16);
      offsets->ensure((size_t)(total_rows + 1) * 4 + 16);
      lvb->ensure((size_t)total_rows + 16);
      lbm->ensure((size_t)((total_rows + 7) / 8) + 16);
      evb->ensure((size_t)n + 16);
      entries->ensure((size_t)n * 4 + 16);
      const uint8_t* defp = (const uint8_t*)leaf_def[l0]->p;
      const uint8_t* repp = (const uint8_t*)leaf_rep[l0]->p;
      pq_launch_list_flags(defp, repp, n, cp.def_slot, (uint32_t*)starts->p, (uint32_t*)elems->p, stream_);
      pq_launch_u32_scan((const uint32_t*)starts->p, n, (uint64_t*)tiles->p, (int32_t*)start_idx->p, stream_);
      pq_launch_u32_scan((const uint32_t*)elems->p, n, (uint64_t*)tiles->p, (int32_t*)elem_idx->p, stream_);
      pq_launch_list_assemble(defp, repp, n, total_rows, cp.def_parent, cp.def_slot, cp.max_def, (const int32_t*)start_idx->p, (const int32_t*)elem_idx->p, nullptr, 0,
                              (int32_t*)offsets->p, (uint8_t*)lvb->p, (uint8_t*)evb->p, nullptr, (uint32_t*)inflate_err->p + l0, stream_);
      pq_launch_pack((const uint8_t*)lvb->p, (uint8_t*)lbm->p, total_rows, stream_);
      pq_launch_list_elem_entries(defp, n, cp.def_slot, (const int32_t*)elem_idx->p, (uint32_t*)entries->p, stream_);
      int32_t nel = 0;
      read_small(&nel, (char*)elem_idx->p + (size_t)n * 4, 4);
      if (nel < 0 || nel > n) throw CometError("internal: list element count out of range");
      DeviceColumnView sv;      // the element struct
      for (size_t l = l0; l < l0 + tc.count; l++) {
        bool khv = false;
        sv.kids.push_back(take_column(lf_out.cols[l], lf_out.types[l], lf_out.has_valid[l], (const uint32_t*)entries->p, nullptr, nel, khv, out.owners));
        sv.kid_has_valid.push_back(khv ? 1 : 0);
      }
      sv.kid_rows = nel;
      const bool elem_nullable = cp.def_elem > cp.def_slot;
      if (elem_nullable) {      // an optional element struct: NULL where the definition level stops short of it
        auto sb = std::make_shared<DevBuf>(), st = std::make_shared<DevBuf>(), sbm = std::make_shared<DevBuf>();
        sb->ensure((size_t)n + 16);
        st->ensure((size_t)std::max(nel, 1) + 16);
        sbm->ensure((size_t)((nel + 7) / 8) + 16);
        pq_launch_level_ge(defp, n, cp.def_elem, (uint8_t*)sb->p, stream_);
        if (nel > 0 && comet_launch_take(1, sb->p, (const uint32_t*)entries->p, nel, st->p, stream_) != 0) throw CometError("parquet: launch failed");
        if (nel > 0) pq_launch_pack((const uint8_t*)st->p, (uint8_t*)sbm->p, nel, stream_);
        sv.valid = (const uint8_t*)sbm->p;
        for (auto& b : {sb, st, sbm}) out.owners.push_back(b);
      }
      HIP_CHECK(hipStreamSynchronize(stream_));      // (`entries` and the scratch arrays may go back to their pools)
      nv.data = offsets->p;
      nv.kids.push_back(sv);
      nv.kid_has_valid.push_back(elem_nullable ? 1 : 0);
      nv.kid_rows = nel;
      if (cp.def_parent > 0) { nv.valid = (const uint8_t*)lbm->p; out.has_valid[t] = true; }
      for (auto& b : {offsets, lbm}) out.owners.push_back(b);
      out.cols[t] = nv;
      continue;
    }
    if (tc.kind == 1) {
      for (size_t l = l0; l < l0 + tc.count; l++) {
        nv.kids.push_back(lf_out.cols[l]);
        nv.kid_has_valid.push_back(lf_out.has_valid[l] ? 1 : 0);
      }
      nv.kid_rows = total_rows;
      const int def_parent = plans[l0].def_parent;
      if (def_parent > 0) {        // an optional struct: NULL where its fields' definition level stops short of it
        auto vb = std::make_shared<DevBuf>(), bm = std::make_shared<DevBuf>();
        vb->ensure((size_t)total_rows + 16);
        bm->ensure((size_t)((total_rows + 7) / 8) + 16);
        pq_launch_level_ge((const uint8_t*)leaf_def[l0]->p, total_rows, def_parent, (uint8_t*)vb->p, stream_);
        pq_launch_pack((const uint8_t*)vb->p, (uint8_t*)bm->p, total_rows, stream_);
        nv.valid = (const uint8_t*)bm->p;
        out.has_valid[t] = true;
        out.owners.push_back(vb);
        out.owners.push_back(bm);
      }
    } else {
      const ColumnPlan& cp = plans[l0];
      const int64_t n = ent_total[l0];
      const TypeId eid = lf_out.types[l0].id;
      const bool by_take = eid == TypeId::String || eid == TypeId::Bytes || eid == TypeId::Bool;      // elements that are not one fixed-width value each
      const int w = by_take ? 0 : cp.out_width;
      if ((!by_take && !leaf_raw_values[l0]) || !leaf_def[l0] || !leaf_rep[l0]) throw CometError("internal: list column without its leaf's levels");
      auto starts = std::make_shared<DevBuf>(), elems = std::make_shared<DevBuf>(), start_idx = std::make_shared<DevBuf>(), elem_idx = std::make_shared<DevBuf>();
      auto offsets = std::make_shared<DevBuf>(), lvb = std::make_shared<DevBuf>(), lbm = std::make_shared<DevBuf>(), evb = std::make_shared<DevBuf>(), ebm = std::make_shared<DevBuf>(),
           evals = std::make_shared<DevBuf>();
      starts->ensure((size_t)n * 4 + 16);
      elems->ensure((size_t)n * 4 + 16);
      start_idx->ensure((size_t)(n + 1) * 4 + 16);
      elem_idx->ensure((size_t)(n + 1) * 4 + 16);
      offsets->ensure((size_t)(total_rows + 1) * 4 + 16);
      lvb->ensure((size_t)total_rows + 16);
      lbm->ensure((size_t)((total_rows + 7) / 8) + 16);
      evb->ensure((size_t)n + 16);
      ebm->ensure((size_t)((n + 7) / 8) + 16);
      evals->ensure((size_t)n * (size_t)w + 16);
      HIP_CHECK(hipMemsetAsync(evb->p, 0, (size_t)n + 16, stream_));
      const uint8_t* defp = (const uint8_t*)leaf_def[l0]->p;
      const uint8_t* repp = (const uint8_t*)leaf_rep[l0]->p;
      pq_launch_list_flags(defp, repp, n, cp.def_slot, (uint32_t*)starts->p, (uint32_t*)elems->p, stream_);
      pq_launch_u32_scan((const uint32_t*)starts->p, n, (uint64_t*)tiles->p, (int32_t*)start_idx->p, stream_);
      pq_launch_u32_scan((const uint32_t*)elems->p, n, (uint64_t*)tiles->p, (int32_t*)elem_idx->p, stream_);
      pq_launch_list_assemble(defp, repp, n, total_rows, cp.def_parent, cp.def_slot, cp.max_def, (const int32_t*)start_idx->p, (const int32_t*)elem_idx->p,
                              by_take ? nullptr : (const uint8_t*)leaf_raw_values[l0]->p, w, (int32_t*)offsets->p, (uint8_t*)lvb->p, (uint8_t*)evb->p, (uint8_t*)evals->p,
                              (uint32_t*)inflate_err->p + l0, stream_);
      pq_launch_pack((const uint8_t*)lvb->p, (uint8_t*)lbm->p, total_rows, stream_);
      DeviceColumnView ev;
      nv.data = offsets->p;
      if (by_take) {
        // strings / booleans: the leaf's column over ENTRIES is a column like any other (offsets + bytes, or bits, + validity); the elements
        // are its entries that hold a slot, taken in order (one small read tells how many there are)
        auto entries = std::make_shared<DevBuf>();
        entries->ensure((size_t)n * 4 + 16);
        pq_launch_list_elem_entries(defp, n, cp.def_slot, (const int32_t*)elem_idx->p, (uint32_t*)entries->p, stream_);
        int32_t nel = 0;
        read_small(&nel, (char*)elem_idx->p + (size_t)n * 4, 4);
        if (nel < 0 || nel > n) throw CometError("internal: list element count out of range");
        bool ehv = false;
        ev = take_column(lf_out.cols[l0], lf_out.types[l0], lf_out.has_valid[l0], (const uint32_t*)entries->p, nullptr, nel, ehv, out.owners);
        HIP_CHECK(hipStreamSynchronize(stream_));      // (`entries` may go back to its pool)
        nv.kids.push_back(ev);
        nv.kid_has_valid.push_back(ehv ? 1 : 0);
        nv.kid_rows = nel;
      } else {
        pq_launch_pack((const uint8_t*)evb->p, (uint8_t*)ebm->p, n, stream_);
        ev.data = evals->p;
        ev.valid = (const uint8_t*)ebm->p;
        nv.kids.push_back(ev);
        nv.kid_has_valid.push_back(cp.max_def > cp.def_slot ? 1 : 0);     // (elements can be NULL only if the element field is optional)
        nv.kid_rows = n;                                                  // room for; offsets[rows] says how many there are
      }
      if (cp.def_parent > 0) { nv.valid = (const uint8_t*)lbm->p; out.has_valid[t] = true; }
      for (auto& b : {starts, elems, start_idx, elem_idx, offsets, lvb, lbm, evb, ebm, evals}) out.owners.push_back(b);
    }
    out.cols[t] = nv;
  }
  // Hive partition columns: one constant per file (SparkPartitionedFile.partition_values, operator.proto:103-109), appended after
  // the file columns (planner.rs:1558-1575); a NULL partition value clears the validity of its rows
  for (size_t p = 0; p < npart; p++) {
    const DType& t = op.partition_schema[p].dtype;
    const bool is_str = t.id == TypeId::String || t.id == TypeId::Bytes;
    const int w = is_str ? 0 : (t.id == TypeId::Bool ? 1 : out_width_of(t));
    auto vals = std::make_shared<DevBuf>();
    auto bytes = std::make_shared<DevBuf>();
    auto vbytes = std::make_shared<DevBuf>();
    vbytes->ensure((size_t)total_rows + 16);
    bool any_null = false;
    int64_t str_total = 0;
    if (is_str) {
      for (auto& sel : sels) {
        if (p >= sel.pf->partition_values.size()) throw CometError("parquet: file without a value for partition column '" + op.partition_schema[p].name + "'");
        const Expr& lit = *sel.pf->partition_values[p];
        if (!lit.lit_null) str_total += (int64_t)lit.lit_bytes.size() * sel.rows;
      }
      if (str_total >= ((int64_t)1 << 31)) throw CometError("parquet: partition string column exceeds 2 GiB");
      vals->ensure((size_t)(total_rows + 1) * 4 + 16);
      bytes->ensure((size_t)std::max<int64_t>(str_total, 1) + 16);
    } else {
      vals->ensure((size_t)total_rows * (size_t)w + 16);
    }
    int64_t str_pos = 0;
    std::vector<std::shared_ptr<DevBuf>> lit_keep;
    for (auto& sel : sels) {
      if (p >= sel.pf->partition_values.size()) throw CometError("parquet: file without a value for partition column '" + op.partition_schema[p].name + "'");
      const Expr& lit = *sel.pf->partition_values[p];
      if (lit.kind != ExprKind::Literal) throw CometError("parquet: partition value is not a literal");
      const uint8_t ok = lit.lit_null ? 0 : 1;
      any_null |= lit.lit_null;
      if (comet_launch_fill(1, (uint8_t*)vbytes->p + sel.row_off, sel.rows, &ok, stream_) != 0) throw CometError("partition column: launch failed");
      if (is_str) {
        const int32_t len = lit.lit_null ? 0 : (int32_t)lit.lit_bytes.size();
        auto dv = std::make_shared<DevBuf>();
        dv->ensure((size_t)std::max(len, 1) + 16);
        if (len) {
          small_host_.ensure(4096);
          if (len > 2048) throw CometError("parquet: partition string value longer than 2048 bytes");
          write_small(dv->p, lit.lit_bytes.data(), (size_t)len);
        }
        lit_keep.push_back(dv);
        if (comet_launch_fill_utf8((int32_t*)vals->p + sel.row_off, (uint8_t*)bytes->p, sel.rows, (int32_t)str_pos, len, (const uint8_t*)dv->p, stream_) != 0)
          throw CometError("partition column: launch failed");
        str_pos += (int64_t)len * sel.rows;
      } else {
        unsigned char raw[16] = {0};
        switch (t.id) {
          case TypeId::Bool: raw[0] = lit.lit_bool ? 1 : 0; break;
          case TypeId::Int8: { int8_t x = (int8_t)lit.lit_i64; memcpy(raw, &x, 1); break; }
          case TypeId::Int16: { int16_t x = (int16_t)lit.lit_i64; memcpy(raw, &x, 2); break; }
          case TypeId::Int32: case TypeId::Date: { int32_t x = (int32_t)lit.lit_i64; memcpy(raw, &x, 4); break; }
          case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz: memcpy(raw, &lit.lit_i64, 8); break;
          case TypeId::Float: { float x = (float)lit.lit_f64; memcpy(raw, &x, 4); break; }
          case TypeId::Double: memcpy(raw, &lit.lit_f64, 8); break;
          case TypeId::Decimal: memcpy(raw, &lit.lit_dec, 16); break;
          default: throw CometError("parquet: partition column of type " + t.str() + " is not supported by the GPU scan yet");
        }
        if (comet_launch_fill(w, (char*)vals->p + (size_t)sel.row_off * (size_t)w, sel.rows, raw, stream_) != 0) throw CometError("partition column: launch failed");
      }
    }
    HIP_CHECK(hipStreamSynchronize(stream_));   // literal staging buffers
    DeviceColumnView cv;
    cv.data = vals->p;
    out.owners.push_back(vals);
    if (is_str) {
      cv.aux = bytes->p;
      out.owners.push_back(bytes);
    } else if (t.id == TypeId::Bool) {
      auto bits = std::make_shared<DevBuf>();
      bits->ensure((size_t)((total_rows + 7) / 8) + 16);
      pq_launch_pack((const uint8_t*)vals->p, (uint8_t*)bits->p, total_rows, stream_);
      cv.data = bits->p;
      out.owners.push_back(bits);
    }
    if (any_null) {
      auto bm = std::make_shared<DevBuf>();
      bm->ensure((size_t)((total_rows + 7) / 8) + 16);
      pq_launch_pack((const uint8_t*)vbytes->p, (uint8_t*)bm->p, total_rows, stream_);
      cv.valid = (const uint8_t*)bm->p;
      out.has_valid[ntop + p] = true;
      out.owners.push_back(bm);
    }
    out.owners.push_back(vbytes);
    out.cols[ntop + p] = cv;
  }
  if (trace) trace_line(scan_id, "all launches issued at %.2f ms\n", ms_since());
  if (trace) trace_line(scan_id, "scan threads spent %.2f ms on chunks: %.2f reading, %.2f walking zstd frames, %.2f inflating pages\n", (double)g_ns_chunk.exchange(0) / 1e6,
                     (double)g_ns_read.exchange(0) / 1e6, (double)g_ns_walk.exchange(0) / 1e6, (double)g_ns_inflate.exchange(0) / 1e6);
  HIP_CHECK(hipStreamSynchronize(stream_));
  if (trace) trace_line(scan_id, "device idle at %.2f ms\n", ms_since());
  if (trace) {
    int64_t miss1[4];
    pool_miss_counters(miss1);
    trace_line(scan_id, "pool misses while this scan ran (process-wide): %lld hipMalloc (%.2f ms), %lld hipHostMalloc (%.2f ms)\n", (long long)(miss1[0] - miss0[0]),
               (double)(miss1[1] - miss0[1]) / 1e6, (long long)(miss1[2] - miss0[2]), (double)(miss1[3] - miss0[3]) / 1e6);
  }
  {
    std::vector<uint32_t> ierr(ncol, 0);
    for (size_t c0 = 0; c0 < ncol; c0 += 512) {   // read_small carries up to 4 KiB
      const size_t n = std::min<size_t>(512, ncol - c0);
      read_small(ierr.data() + c0, (char*)inflate_err->p + c0 * 4, n * 4);
    }
    for (size_t c = 0; c < ncol; c++) {
      if (ierr[c] == 0xD1u) throw CometError("Parquet column '" + fields[c].parent + "': the list's repetition levels do not add up to the row group's rows");
      if (ierr[c]) throw CometError("Parquet column '" + fields[c].name + "': corrupt compressed data page (device decompression, page job " +
                                    std::to_string(ierr[c] >> 8) + ", code " + std::to_string(ierr[c] & 0xff) + ")");
    }
  }
  out.owners.push_back(tiles);
  out.owners.push_back(vidx);
  // staging buffers can go back to the pools now that the stream is idle
  keep.clear();
  if (trace) trace_line(scan_id, "device buffers released at %.2f ms\n", ms_since());
  // the run tables are many large vectors (one munmap each when freed): give them to a throw-away thread instead of paying
  // ~5 ms of page-table teardown on the query's critical path
  {
    auto* garbage = new std::vector<HostChunk>(std::move(chunks));
    std::thread([garbage]() { delete garbage; }).detach();
  }
  if (trace) trace_line(scan_id, "host tables handed off at %.2f ms\n", ms_since());
  col_staged.clear();
  if (trace) trace_line(scan_id, "staging released at %.2f ms\n", ms_since());
  sels.clear();
  if (trace) trace_line(scan_id, "files closed at %.2f ms\n", ms_since());
  return out;
}

}  // namespace comet
