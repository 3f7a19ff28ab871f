// Parquet footer / page-header parsing (Thrift compact protocol, hand-written: no thrift or parquet library in
// this image) and host-side page preparation for the device decoder.
// The reference delegates this to the `parquet` 58.4.0 crate through DataFusion's ParquetSource
// (native/core/src/parquet/parquet_exec.rs:60-211); the format itself is the public Apache Parquet spec.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace comet {
namespace pq {

enum PhysType : int { BOOLEAN = 0, INT32 = 1, INT64 = 2, INT96 = 3, FLOAT = 4, DOUBLE = 5, BYTE_ARRAY = 6, FLBA = 7 };
enum Encoding : int { PLAIN = 0, PLAIN_DICTIONARY = 2, RLE = 3, BIT_PACKED = 4, DELTA_BINARY_PACKED = 5, DELTA_LENGTH_BYTE_ARRAY = 6, DELTA_BYTE_ARRAY = 7,
                      RLE_DICTIONARY = 8, BYTE_STREAM_SPLIT = 9 };
enum Codec : int { UNCOMPRESSED = 0, SNAPPY = 1, GZIP = 2, LZO = 3, BROTLI = 4, LZ4 = 5, ZSTD = 6, LZ4_RAW = 7 };
enum PageType : int { DATA_PAGE = 0, INDEX_PAGE = 1, DICTIONARY_PAGE = 2, DATA_PAGE_V2 = 3 };

struct SchemaElement {
  int type = -1;          // PhysType, -1 for groups
  int type_length = 0;
  int repetition = 0;     // 0 required, 1 optional, 2 repeated
  std::string name;
  int num_children = 0;
  int converted_type = -1;
  int scale = 0, precision = 0;
  int field_id = -1;      // SchemaElement.field_id (9), -1 = absent
  // from LogicalType (10) or, for older writers, converted_type (6)
  int ts_unit = 0;        // timestamps: 0 = not a timestamp, 1 millis, 2 micros, 3 nanos
  bool ts_utc = true;     // isAdjustedToUTC
  int int_bits = 0;       // INTEGER annotation: 8 / 16 / 32 / 64, 0 = none
  bool int_signed = true;
  bool is_time = false;   // TIME_MILLIS / TIME_MICROS / TIME(…)
};

struct ColumnMeta {
  int type = 0;
  std::vector<std::string> path;
  int codec = 0;
  int64_t num_values = 0;
  int64_t total_uncompressed = 0, total_compressed = 0;
  int64_t data_page_offset = 0, dictionary_page_offset = 0;
  bool delta_encoded = false;   // ColumnMetaData.encodings names a DELTA_* encoding: decoded pages outgrow total_uncompressed_size
  bool prefix_encoded = false;  // … DELTA_BYTE_ARRAY among them: the decoded size is only known after reading the pages' length blocks
  // Statistics (parquet.thrift Statistics: 3 null_count, 5 max_value, 6 min_value; 1/2 = deprecated max/min, signed order only)
  bool has_min_max = false;
  std::string min_value, max_value;   // PLAIN-encoded
  bool stats_typed_order = false;     // they are Statistics.min_value / max_value (5 / 6: the column's own order, unsigned bytewise for BYTE_ARRAY), not the deprecated pair
  int64_t null_count = -1;            // -1 = not recorded
  // page index (ColumnChunk fields 4-7): where the OffsetIndex / ColumnIndex of this chunk sit in the file (0 = the writer wrote none)
  int64_t offset_index_offset = 0, column_index_offset = 0;
  int32_t offset_index_length = 0, column_index_length = 0;
  // ColumnMetaData 14 / 15: where the chunk's Bloom filter (header + bitset) sits (0 = none; a length is optional in the format)
  int64_t bloom_filter_offset = 0;
  int32_t bloom_filter_length = 0;
};

// The split-block Bloom filter of a column chunk (parquet-format BloomFilter.md): blocks of 256 bits = eight 32-bit words; a value's XXH64 (seed 0, over its
// PLAIN encoding — a BYTE_ARRAY without its length prefix) picks the block with its high 32 bits and one bit per word with its low 32 bits times eight salts.
// Only BLOCK / XXHASH / UNCOMPRESSED exist in the format; anything else is "no usable filter".
size_t parse_bloom_header(const uint8_t* data, size_t len, int32_t& num_bytes);     // → the header's length in bytes; throws CometError on what it cannot use
bool sbbf_might_contain(const uint8_t* bits, size_t nbytes, uint64_t hash);
uint64_t xxh64(const void* data, size_t len, uint64_t seed);

// The page index of one column chunk (parquet.thrift ColumnIndex / OffsetIndex): one entry per DATA page, in file order.
struct PageIndex {
  std::vector<int64_t> first_row;          // OffsetIndex.page_locations[i].first_row_index (row-group relative)
  std::vector<char> null_page;             // ColumnIndex.null_pages: the page holds only NULLs (min / max are then meaningless)
  std::vector<std::string> min_value, max_value;   // PLAIN-encoded
  std::vector<int64_t> null_count;         // empty if the writer recorded none
};
PageIndex parse_page_index(const uint8_t* column_index, size_t ci_len, const uint8_t* offset_index, size_t oi_len);

struct RowGroup {
  std::vector<ColumnMeta> columns;
  int64_t total_byte_size = 0, num_rows = 0, total_compressed = 0;
};

struct FileMeta {
  std::vector<SchemaElement> schema;   // depth-first, element 0 = root
  int64_t num_rows = 0;
  std::vector<RowGroup> row_groups;
};

struct PageHeader {
  int type = -1;
  int32_t uncompressed_size = 0, compressed_size = 0;
  int32_t num_values = 0;
  int encoding = 0;
  int def_encoding = 0, rep_encoding = 0;
  // v2
  int32_t num_nulls = 0, num_rows = 0, def_bytes = 0, rep_bytes = 0;
  bool v2_compressed = true;
  size_t header_len = 0;   // bytes consumed by the thrift header
};

FileMeta parse_footer(const uint8_t* file, size_t size);
PageHeader parse_page_header(const uint8_t* p, size_t avail);

// decompress one page body; throws CometError for unsupported codecs
void decompress(int codec, const uint8_t* src, size_t src_len, uint8_t* dst, size_t dst_len);
// first `want` bytes of a raw snappy stream (the levels in front of a v1 page's values); returns the bytes produced
size_t snappy_prefix(const uint8_t* src, size_t n, uint8_t* dst, size_t want);

// Random access to the OUTPUT of a raw snappy stream without producing it: the element list (a few entries per 64 KiB for data that
// does not compress — bit-packed dictionary indices are such data) maps an output position to a literal byte of the compressed stream
// or, through copies, to an earlier output position.  The scan reads the run headers of a dictionary-encoded page this way and ships
// the page compressed for the device to inflate, instead of decompressing it on a host core just to look at a few hundred header bytes.
struct SnappyView {
  struct El { uint32_t out_pos, len, src; uint8_t copy; };
  const uint8_t* stream = nullptr;
  size_t stream_len = 0, out_len = 0;
  std::vector<El> els;
  mutable size_t last = 0;
  // false: malformed, or more than max_elems elements (a page that compresses well: decompressing it is cheap, do that instead)
  bool build(const uint8_t* src, size_t n, size_t max_elems);
  uint8_t at(size_t o) const;      // byte o of the output; throws CometError on a malformed reference
};
// Value encodings the device kernels do not read are rewritten as PLAIN on the host (parquet-format Encodings.md): DELTA_BINARY_PACKED
// (INT32 / INT64, `width` = 4 / 8), DELTA_LENGTH_BYTE_ARRAY (→ 4-byte length + bytes per value), BYTE_STREAM_SPLIT (`width` bytes per value).
// `src` holds the page's value bytes; the PLAIN bytes are appended to `out`.  Throws on truncated or inconsistent input.
void delta_binary_to_plain(const uint8_t* src, size_t len, int width, int64_t max_values, std::vector<uint8_t>& out);
void delta_length_byte_array_to_plain(const uint8_t* src, size_t len, int64_t max_values, std::vector<uint8_t>& out);
void byte_stream_split_to_plain(const uint8_t* src, size_t len, int width, std::vector<uint8_t>& out);
// DELTA_BYTE_ARRAY (incremental encoding: a DELTA_BINARY_PACKED block of prefix lengths, then the suffixes as DELTA_LENGTH_BYTE_ARRAY) →
// PLAIN; and the PLAIN size alone (Σ 4 + prefix + suffix), from the two length blocks
void delta_byte_array_to_plain(const uint8_t* src, size_t len, int64_t max_values, std::vector<uint8_t>& out);
size_t delta_byte_array_plain_size(const uint8_t* src, size_t len, int64_t max_values);
std::vector<int64_t> delta_binary_unpack(const uint8_t* src, size_t len, int64_t max_values, size_t* consumed);

}  // namespace pq
}  // namespace comet
