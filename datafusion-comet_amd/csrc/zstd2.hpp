// Scratch buffers + launcher of the zstd pipeline (zstd2.cpp; device/zstd2.hpp holds the algorithm).  One object per group of pages being
// decompressed: its buffers must outlive the work queued on the stream (the scan keeps it with the column's other device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "exec.hpp"
#include "parquet_dev.h"

namespace comet_zstd2 { struct ZBlock; }

namespace comet {

struct Zstd2Scratch {
  DevBuf tables, recs, lits, status;
  PinnedBuf h_tables;
  int64_t blocks_ = 0, records_ = 0;
  // jobs: the pages (offsets into bytes_dev); job i's blocks — what the host walk over its frame found (comet_zstd2::scan_page) — are
  // blocks_host[jobs[i].preamble … + jobs[i].pad).  err_dev: one word, first failing page as (page << 8 | code).  Nothing is read back.
  void run(const PqInflate* jobs_host, int njobs, const comet_zstd2::ZBlock* blocks_host, uint8_t* bytes_dev, uint32_t* err_dev, hipStream_t st);
};

}  // namespace comet
