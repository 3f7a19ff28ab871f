// NativeScan: Parquet files → Arrow columns resident in HBM.
// Host: footer + page headers (Thrift), page decompression, hybrid-run headers → PqPage / PqRun tables.
// Device: every level and value is decoded by parquet_kernels.hip.
// Reference path: NativeScan arm planner.rs:1523-1668 → init_datasource_exec parquet/parquet_exec.rs:60-211
// (column model SURVEY Appendix C.12: output = required_schema fields; row groups chosen by byte-range midpoint).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>

#include "exec.hpp"
#include "parquet_dev.h"
#include "parquet_meta.hpp"

extern "C" {
void pq_launch_validity(const PqDecodeArgs* a, void* st);
void pq_launch_vidx(const uint8_t* valid, int64_t n, uint64_t* tiles, uint32_t* vidx, void* st);
void pq_launch_decode_fixed(const PqDecodeArgs* a, void* st);
void pq_launch_string_lengths(const PqDecodeArgs* a, void* st);
void pq_launch_string_copy(const PqDecodeArgs* a, void* st);
void pq_launch_u32_scan(const uint32_t* in, int64_t n, uint64_t* tiles, int32_t* out, void* st);
void pq_launch_pack(const uint8_t* bytes, uint8_t* bitmap, int64_t n, void* st);
}

namespace comet {

namespace {

struct MappedFile {
  const uint8_t* data = nullptr;
  size_t size = 0;
  int fd = -1;
  explicit MappedFile(const std::string& path) {
    fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw CometError("cannot open Parquet file " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw CometError("cannot stat " + path); }
    size = (size_t)st.st_size;
    if (size) {
      void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (p == MAP_FAILED) { close(fd); throw CometError("cannot mmap " + path); }
      data = (const uint8_t*)p;
    }
  }
  ~MappedFile() {
    if (data) munmap((void*)data, size);
    if (fd >= 0) close(fd);
  }
  MappedFile(const MappedFile&) = delete;
  MappedFile& operator=(const MappedFile&) = delete;
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
void parse_hybrid_runs(const uint8_t* staged, size_t pos, size_t end, int bw, int32_t max_values, std::vector<PqRun>& runs) {
  int32_t vstart = 0;
  const int vbytes = (bw + 7) / 8;
  while (pos < end && (max_values < 0 || vstart < max_values)) {
    uint64_t h = 0;
    int sh = 0;
    while (true) {
      if (pos >= end) throw CometError("parquet: truncated hybrid run header");
      uint8_t b = staged[pos++];
      h |= (uint64_t)(b & 0x7f) << sh;
      if (!(b & 0x80)) break;
      sh += 7;
    }
    PqRun r;
    memset(&r, 0, sizeof r);
    r.value_start = vstart;
    if (h & 1) {
      int64_t groups = (int64_t)(h >> 1);
      r.is_rle = 0;
      r.count = (int32_t)(groups * 8);
      r.byte_off = (int64_t)pos;
      pos += (size_t)(groups * bw);
    } else {
      r.is_rle = 1;
      r.count = (int32_t)(h >> 1);
      uint32_t v = 0;
      for (int k = 0; k < vbytes; k++) {
        if (pos >= end) throw CometError("parquet: truncated RLE run");
        v |= (uint32_t)staged[pos++] << (8 * k);
      }
      r.rle_value = v;
    }
    if (r.count == 0) continue;
    vstart += r.count;
    runs.push_back(r);
  }
}

struct ColumnPlan {
  int leaf = -1;          // index into row_group.columns
  pq::SchemaElement el;
  int kind = -1;          // PQ_* conversion, -1 for strings
  int src_width = 0;
  int out_width = 0;      // bytes per output value (0 strings)
  bool is_string = false;
};

ColumnPlan plan_column(const StructField& want, const pq::FileMeta& fm, bool case_sensitive) {
  ColumnPlan cp;
  int leaf = 0;
  for (size_t i = 1; i < fm.schema.size(); i++) {
    const pq::SchemaElement& e = fm.schema[i];
    if (e.num_children > 0) throw CometError("nested Parquet schemas are not supported by the GPU scan yet");
    if ((case_sensitive && e.name == want.name) || (!case_sensitive && iequals(e.name, want.name))) {
      cp.leaf = leaf;
      cp.el = e;
      break;
    }
    leaf++;
  }
  if (cp.leaf < 0) throw CometError("Parquet column '" + want.name + "' not found (missing-column defaults are not supported by the GPU scan yet)");
  if (cp.el.repetition == 2) throw CometError("repeated Parquet columns are not supported by the GPU scan yet");
  const DType& t = want.dtype;
  const int pt = cp.el.type;
  auto bad = [&]() { return CometError("Parquet column '" + want.name + "': physical type " + std::to_string(pt) + " cannot be read as " + t.str() + " by the GPU scan yet"); };
  switch (t.id) {
    case TypeId::Int32: case TypeId::Date: if (pt != pq::INT32) throw bad(); cp.kind = PQ_COPY4; cp.src_width = 4; cp.out_width = 4; break;
    case TypeId::Int16: if (pt != pq::INT32) throw bad(); cp.kind = PQ_I32_TO_I16; cp.src_width = 4; cp.out_width = 2; break;
    case TypeId::Int8: if (pt != pq::INT32) throw bad(); cp.kind = PQ_I32_TO_I8; cp.src_width = 4; cp.out_width = 1; break;
    case TypeId::Int64: case TypeId::Timestamp: case TypeId::TimestampNtz:
      if (pt == pq::INT64) { cp.kind = PQ_COPY8; cp.src_width = 8; }
      else if (pt == pq::INT32 && t.id == TypeId::Int64) { cp.kind = PQ_I32_TO_I64; cp.src_width = 4; }   // type promotion (schema_adapter.rs)
      else throw bad();
      cp.out_width = 8;
      break;
    case TypeId::Float: if (pt != pq::FLOAT) throw bad(); cp.kind = PQ_COPY4; cp.src_width = 4; cp.out_width = 4; break;
    case TypeId::Double: if (pt != pq::DOUBLE) throw bad(); cp.kind = PQ_COPY8; cp.src_width = 8; cp.out_width = 8; break;
    case TypeId::Bool: if (pt != pq::BOOLEAN) throw bad(); cp.kind = PQ_BOOL; cp.src_width = 0; cp.out_width = 1; break;
    case TypeId::Decimal:
      if (pt == pq::INT32) { cp.kind = PQ_I32_TO_DEC; cp.src_width = 4; }
      else if (pt == pq::INT64) { cp.kind = PQ_I64_TO_DEC; cp.src_width = 8; }
      else if (pt == pq::FLBA && cp.el.type_length >= 1 && cp.el.type_length <= 16) { cp.kind = PQ_FLBA_TO_DEC; cp.src_width = cp.el.type_length; }
      else throw bad();
      if (cp.el.scale != t.scale) throw CometError("Parquet decimal scale differs from the requested type (decimal widening is not supported by the GPU scan yet)");
      cp.out_width = 16;
      break;
    case TypeId::String: case TypeId::Bytes: if (pt != pq::BYTE_ARRAY) throw bad(); cp.is_string = true; break;
    default: throw bad();
  }
  return cp;
}

struct ChunkBuffers {   // device + host staging for one column chunk; kept alive until the stream is idle
  DevBuf bytes, pages, def_runs, idx_runs, dict, dict_offs, str_offs;
  PinnedBuf h_bytes, h_tables;
};

}  // namespace

DevTable ExecutionContext::scan_parquet(const Operator& op) {
  if (!op.partition_schema.empty()) throw CometError("Hive-partition columns are not supported by the GPU Parquet scan yet");
  const size_t ncol = op.required_schema.size();
  DevTable out;
  for (auto& f : op.required_schema) out.types.push_back(f.dtype);
  out.cols.assign(ncol, DeviceColumnView());
  out.has_valid.assign(ncol, false);
  if (op.files.empty()) return out;   // EmptyExec (planner.rs:1548-1556)

  // pass 1: open files, pick row groups (midpoint rule), total rows
  struct Sel { std::shared_ptr<MappedFile> file; std::shared_ptr<pq::FileMeta> meta; int rg; int64_t row_off; };
  std::vector<Sel> sels;
  int64_t total_rows = 0;
  for (auto& pf : op.files) {
    auto mf = std::make_shared<MappedFile>(path_from_uri(pf.file_path));
    auto fm = std::make_shared<pq::FileMeta>(pq::parse_footer(mf->data, mf->size));
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
      sels.push_back({mf, fm, (int)g, total_rows});
      total_rows += rg.num_rows;
    }
  }
  out.rows = total_rows;
  bytes_scanned_ = 0;
  if (total_rows == 0) return out;
  if (total_rows >= ((int64_t)1 << 31)) throw CometError("GPU Parquet scan: more than 2^31 rows in one partition");

  std::vector<std::shared_ptr<ChunkBuffers>> keep;
  auto tiles = std::make_shared<DevBuf>();
  tiles->ensure((size_t)((total_rows + 1023) / 1024 + 2) * 8);
  auto vidx = std::make_shared<DevBuf>();
  vidx->ensure((size_t)total_rows * 4 + 16);

  for (size_t c = 0; c < ncol; c++) {
    const StructField& want = op.required_schema[c];
    auto values = std::make_shared<DevBuf>();
    auto valid_bytes = std::make_shared<DevBuf>();
    auto lengths = std::make_shared<DevBuf>();
    valid_bytes->ensure((size_t)total_rows + 16);
    bool any_optional = false;
    std::vector<PqDecodeArgs> string_args;   // replayed for the copy phase
    bool is_string = false;
    int out_width = 0;
    for (auto& sel : sels) {
      const pq::RowGroup& rg = sel.meta->row_groups[sel.rg];
      ColumnPlan cp = plan_column(want, *sel.meta, true);
      if ((size_t)cp.leaf >= rg.columns.size()) throw CometError("parquet: column index out of range");
      const pq::ColumnMeta& cm = rg.columns[cp.leaf];
      is_string = cp.is_string;
      out_width = cp.out_width;
      if (!values->p && !cp.is_string) values->ensure((size_t)total_rows * cp.out_width + 16);
      if (cp.is_string && !lengths->p) lengths->ensure((size_t)total_rows * 4 + 16);
      const int max_def = cp.el.repetition == 1 ? 1 : 0;
      any_optional |= max_def > 0;
      const int64_t n_rows = rg.num_rows;

      auto cb = std::make_shared<ChunkBuffers>();
      keep.push_back(cb);
      cb->h_bytes.ensure((size_t)cm.total_uncompressed + 64);
      uint8_t* staged = (uint8_t*)cb->h_bytes.p;
      size_t spos = 0;
      std::vector<PqPage> pages;
      std::vector<PqRun> def_runs, idx_runs;
      std::vector<uint8_t> dict_bytes;
      std::vector<int32_t> dict_offs;
      std::vector<int64_t> str_offs;

      int64_t off = (cm.dictionary_page_offset > 0 && cm.dictionary_page_offset < cm.data_page_offset) ? cm.dictionary_page_offset : cm.data_page_offset;
      const int64_t chunk_end = off + cm.total_compressed;
      if (off < 0 || (size_t)chunk_end > sel.file->size) throw CometError("parquet: column chunk outside the file");
      bytes_scanned_ += cm.total_compressed;
      int64_t values_seen = 0;
      std::vector<uint8_t> tmp;
      while (values_seen < cm.num_values && off < chunk_end) {
        pq::PageHeader h = pq::parse_page_header(sel.file->data + off, (size_t)(chunk_end - off));
        const uint8_t* body = sel.file->data + off + h.header_len;
        off += (int64_t)h.header_len + h.compressed_size;
        if (h.type == pq::DICTIONARY_PAGE) {
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
        if (spos + (size_t)h.uncompressed_size + 16 > cb->h_bytes.cap) {
          // total_uncompressed_size excludes nothing we stage, but stay safe against odd writers
          throw CometError("parquet: column chunk larger than its declared uncompressed size");
        }
        PqPage pg;
        memset(&pg, 0, sizeof pg);
        pg.row_start = values_seen;
        pg.num_values = h.num_values;
        size_t page_begin = spos, vals_begin, page_end;
        if (h.type == pq::DATA_PAGE) {
          pq::decompress(cm.codec, body, (size_t)h.compressed_size, staged + spos, (size_t)h.uncompressed_size);
          page_end = spos + (size_t)h.uncompressed_size;
          size_t p = page_begin;
          if (max_def > 0) {
            if (h.def_encoding != pq::RLE) throw CometError("parquet: only RLE definition levels are supported");
            uint32_t dl;
            memcpy(&dl, staged + p, 4);
            p += 4;
            pg.def_run_first = (int32_t)def_runs.size();
            parse_hybrid_runs(staged, p, p + dl, 1, h.num_values, def_runs);
            pg.def_run_count = (int32_t)def_runs.size() - pg.def_run_first;
            p += dl;
          }
          vals_begin = p;
        } else {
          // v2: levels are never compressed and precede the (optionally compressed) values
          if (h.rep_bytes) throw CometError("parquet: repetition levels are not supported");
          memcpy(staged + spos, body, (size_t)h.def_bytes);
          if (max_def > 0 && h.def_bytes) {
            pg.def_run_first = (int32_t)def_runs.size();
            parse_hybrid_runs(staged, spos, spos + (size_t)h.def_bytes, 1, h.num_values, def_runs);
            pg.def_run_count = (int32_t)def_runs.size() - pg.def_run_first;
          }
          vals_begin = spos + (size_t)h.def_bytes;
          const size_t vcomp = (size_t)h.compressed_size - (size_t)h.def_bytes, vun = (size_t)h.uncompressed_size - (size_t)h.def_bytes;
          pq::decompress(h.v2_compressed ? cm.codec : pq::UNCOMPRESSED, body + h.def_bytes, vcomp, staged + vals_begin, vun);
          page_end = vals_begin + vun;
        }
        if (h.encoding == pq::PLAIN) {
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
          throw CometError("parquet: RLE-encoded booleans are not supported yet");
        } else {
          throw CometError("parquet: value encoding " + std::to_string(h.encoding) + " is not supported yet (PLAIN and RLE_DICTIONARY are)");
        }
        spos = page_end;
        values_seen += h.num_values;
        pages.push_back(pg);
      }
      if (values_seen != n_rows) throw CometError("parquet: column chunk values do not add up to the row group's rows (nested data?)");
      if (pages.empty()) throw CometError("parquet: column chunk without data pages");
      memset(staged + spos, 0, 16);

      // tables → one pinned block → device
      const size_t sz_pages = pages.size() * sizeof(PqPage), sz_def = def_runs.size() * sizeof(PqRun), sz_idx = idx_runs.size() * sizeof(PqRun);
      const size_t sz_do = dict_offs.size() * 4, sz_so = (str_offs.size() + 1) * 8;
      cb->bytes.ensure(spos + 16);
      HIP_CHECK(hipMemcpyAsync(cb->bytes.p, staged, spos + 16, hipMemcpyHostToDevice, stream_));
      auto up = [&](DevBuf& d, const void* src, size_t n) {
        d.ensure(n + 16);
        if (n) HIP_CHECK(hipMemcpy(d.p, src, n, hipMemcpyHostToDevice));   // small tables: synchronous copy from pageable memory
      };
      up(cb->pages, pages.data(), sz_pages);
      up(cb->def_runs, def_runs.data(), sz_def);
      up(cb->idx_runs, idx_runs.data(), sz_idx);
      up(cb->dict, dict_bytes.data(), dict_bytes.size());
      up(cb->dict_offs, dict_offs.data(), sz_do);
      str_offs.push_back(0);
      up(cb->str_offs, str_offs.data(), sz_so);

      PqDecodeArgs a;
      memset(&a, 0, sizeof a);
      a.pages = (const PqPage*)cb->pages.p;
      a.npages = (int32_t)pages.size();
      a.max_def = max_def;
      a.def_runs = (const PqRun*)cb->def_runs.p;
      a.idx_runs = (const PqRun*)cb->idx_runs.p;
      a.bytes = (const uint8_t*)cb->bytes.p;
      a.dict = (const uint8_t*)cb->dict.p;
      a.dict_offs = (const int32_t*)cb->dict_offs.p;
      a.plain_str_offs = (const int64_t*)cb->str_offs.p;
      a.n_rows = n_rows;
      a.kind = cp.kind;
      a.width = cp.src_width;
      a.valid_out = (uint8_t*)valid_bytes->p + sel.row_off;
      a.vidx = (uint32_t*)vidx->p + sel.row_off;
      pq_launch_validity(&a, stream_);
      if (max_def > 0) pq_launch_vidx(a.valid_out, n_rows, (uint64_t*)tiles->p, a.vidx, stream_);
      if (!cp.is_string) {
        a.values_out = (char*)values->p + (size_t)sel.row_off * cp.out_width;
        pq_launch_decode_fixed(&a, stream_);
      } else {
        a.lengths_out = (uint32_t*)lengths->p + sel.row_off;
        pq_launch_string_lengths(&a, stream_);
        string_args.push_back(a);
      }
      // vidx is reused by the next chunk of this column: keep launches ordered on the single stream (they are)
      if (cp.is_string) {
        // the copy phase needs vidx again: give string chunks their own index buffer
        auto own = std::make_shared<DevBuf>();
        own->ensure((size_t)n_rows * 4 + 16);
        if (max_def > 0) HIP_CHECK(hipMemcpyAsync(own->p, a.vidx, (size_t)n_rows * 4, hipMemcpyDeviceToDevice, stream_));
        string_args.back().vidx = (uint32_t*)own->p;
        out.owners.push_back(own);
      }
    }
    DeviceColumnView cv;
    if (is_string) {
      auto offsets = std::make_shared<DevBuf>();
      offsets->ensure((size_t)(total_rows + 1) * 4 + 16);
      pq_launch_u32_scan((const uint32_t*)lengths->p, total_rows, (uint64_t*)tiles->p, (int32_t*)offsets->p, stream_);
      int32_t total_bytes = 0;
      read_small(&total_bytes, (char*)offsets->p + (size_t)total_rows * 4, 4);
      auto data = std::make_shared<DevBuf>();
      data->ensure((size_t)std::max(total_bytes, 1) + 16);
      size_t k = 0;
      for (auto& sel : sels) {
        PqDecodeArgs a = string_args[k++];
        a.str_offsets = (const int32_t*)offsets->p + sel.row_off;
        a.str_bytes_out = (uint8_t*)data->p;
        pq_launch_string_copy(&a, stream_);
      }
      cv.data = offsets->p;
      cv.aux = data->p;
      out.owners.push_back(offsets);
      out.owners.push_back(data);
      out.owners.push_back(lengths);
    } else if (out.types[c].id == TypeId::Bool) {
      auto bits = std::make_shared<DevBuf>();
      bits->ensure((size_t)((total_rows + 7) / 8) + 16);
      pq_launch_pack((const uint8_t*)values->p, (uint8_t*)bits->p, total_rows, stream_);
      cv.data = bits->p;
      out.owners.push_back(bits);
      out.owners.push_back(values);
    } else {
      (void)out_width;
      cv.data = values->p;
      out.owners.push_back(values);
    }
    if (any_optional) {
      auto bm = std::make_shared<DevBuf>();
      bm->ensure((size_t)((total_rows + 7) / 8) + 16);
      pq_launch_pack((const uint8_t*)valid_bytes->p, (uint8_t*)bm->p, total_rows, stream_);
      cv.valid = (const uint8_t*)bm->p;
      out.has_valid[c] = true;
      out.owners.push_back(bm);
    }
    out.owners.push_back(valid_bytes);
    out.cols[c] = cv;
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  out.owners.push_back(tiles);
  out.owners.push_back(vidx);
  // staging buffers can go back to the pools now that the stream is idle
  keep.clear();
  return out;
}

}  // namespace comet
