// Comet shuffle block format (host side): what sits between two native stages when Spark's exchange carries the data.
//
//   block  := u64le length-of-the-rest | u64le field_count | 4-byte codec tag | payload
//   tag    := "NONE" | "ZSTD" | "LZ4_" | "SNAP"            (native/shuffle/src/writers/shuffle_block_writer.rs:86-137)
//   payload:= Arrow IPC stream (schema message, one record batch message, end-of-stream) — raw, one zstd frame, one LZ4
//             frame, or Snappy framing format                (shuffle_block_writer.rs:179-238, native/shuffle/src/ipc.rs:23-52)
//
// The readers hand the bytes after the two u64 words (tag + payload) to the native side
// (operators/shuffle_scan.rs:139-171, jni_api.rs:1163-1181); `decode_shuffle_block` takes exactly that.
// The Arrow IPC messages (flatbuffers) and the three codecs are written out by hand — there is no Arrow C++ / flatbuffers /
// lz4 / snappy library in this image; zstd comes from libzstd.so.1 through dlopen like the Parquet reader's.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "exec.hpp"

namespace comet {

enum class ShuffleCodec : int { None = 0, Zstd = 1, Lz4 = 2, Snappy = 3 };   // operator.proto:679-686

// rows [first, first+rows) of one host-resident column in Arrow layout
struct ColumnSlice {
  DType type;
  const uint8_t* validity = nullptr;   // bitmap addressed from row 0 of the buffers, or nullptr (no nulls)
  const void* values = nullptr;        // fixed-width values / boolean bits / int32 offsets, addressed from row 0
  const uint8_t* data = nullptr;       // Utf8 / Binary bytes: data[0] is byte `data_origin` of the column's value bytes
  int64_t data_origin = 0;
  int64_t first = 0;
  // nested columns: a Struct's fields (each addressed like their parent: its rows from `first` on), a List's one element column — addressed
  // from ELEMENT 0 of the whole column (the list's `values` are its int32 offsets; the rows' elements are offsets[first] … offsets[first + rows))
  std::vector<ColumnSlice> kids;
};

// appends one complete block (length word included) for `rows` rows to `out`; nothing is written for rows == 0
// (shuffle_block_writer.rs:185-187).  Returns the bytes appended.
size_t encode_shuffle_block(const std::vector<ColumnSlice>& cols, int64_t rows, ShuffleCodec codec, int level, std::vector<uint8_t>& out);

// `block` starts at the codec tag.  Dictionary-encoded columns are unpacked (shuffle_scan.rs:175-183).
HostBatch decode_shuffle_block(const uint8_t* block, size_t len);

// codec primitives (also used by tests through the C ABI)
void snappy_compress_raw(const uint8_t* src, size_t n, std::vector<uint8_t>& out);
void lz4_compress_block(const uint8_t* src, size_t n, std::vector<uint8_t>& out);
size_t lz4_decompress_block(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap, size_t dst_pos);
uint32_t xxh32(const uint8_t* p, size_t n, uint32_t seed);
uint32_t crc32c(const uint8_t* p, size_t n, uint32_t init = 0);   // init = the running value (crc32c_append)

// move a HostBatch into caller-allocated Arrow C Data structs, one per column (prepare_output, jni_api.rs:674-742)
void export_host_batch(HostBatch& b, ArrowArray** out_arrays, ArrowSchema** out_schemas, int n_out);
std::string expected_format(const DType& t);

// A stream of shuffle blocks handed over by the caller (the C-ABI face of CometShuffleBlockIterator.java: hasNext() returns the
// next block's length or -1, getBuffer() its bytes starting at the codec tag).
struct CometShuffleBlockStreamC {
  int64_t (*next_block)(CometShuffleBlockStreamC* self, const uint8_t** data);   // length, -1 at end, -2 on error
  const char* (*get_last_error)(CometShuffleBlockStreamC* self);
  void (*release)(CometShuffleBlockStreamC* self);
  void* private_data;
};
// wraps it as an ArrowArrayStream of struct arrays (one per block), so a ShuffleScan leaf runs through the same host-input
// path as a Scan leaf; takes ownership of `blocks`
// schema message of an Arrow IPC stream → fields (names, types, nullability, PARQUET:field_id)
std::vector<StructField> decode_ipc_schema(const uint8_t* p, size_t n);

ArrowArrayStream* shuffle_blocks_as_arrow_stream(CometShuffleBlockStreamC* blocks, std::vector<DType> types);

}  // namespace comet
