// The row-based ("JVM columnar") shuffle entry points: Spark's own sorter hands over UnsafeRow addresses and the native side turns them
// into shuffle blocks.  Host-memory work by definition — the rows live in JVM off-heap pages — so this is host code:
//   Native.sortRowPartitionsNative  (native/core/src/execution/jni_api.rs:1130-1160): in-place ascending sort of packed i64 records
//   Native.writeSortedFileNative    (jni_api.rs:1043-1127 → process_sorted_row_partition, native/shuffle/src/spark_unsafe/row.rs:1342-1438)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "shuffle_format.hpp"

namespace comet {

// ascending signed order, in place (the reference calls rdxsort on the i64 slice)
void sort_row_partitions(int64_t* a, size_t n);

struct SortedFileResult {
  int64_t written = 0;          // bytes appended to the file
  bool has_checksum = false;
  uint32_t checksum = 0;
  int64_t encode_nanos = 0;
};
// checksum_algo: 0 CRC32, 1 Adler32, 2 CRC32C (writers/checksum.rs:39-73).  Rows are read as UnsafeRows of schema.size() fields;
// every `batch_size` rows become one shuffle block appended to `path`.
SortedFileResult write_sorted_rows(const int64_t* row_addresses, const int32_t* row_sizes, size_t row_num, const std::vector<DType>& schema,
                                   const std::string& path, size_t batch_size, bool checksum_enabled, int checksum_algo, bool has_initial,
                                   uint32_t initial_checksum, ShuffleCodec codec, int level);

uint32_t crc32_ieee(const uint8_t* p, size_t n, uint32_t init = 0);
uint32_t adler32(const uint8_t* p, size_t n, uint32_t init = 1);

}  // namespace comet
