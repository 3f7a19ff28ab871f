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
  DevBuf tables, recs, lits;
  PinnedBuf h_tables;
  uint32_t* status = nullptr;      // per page, inside `tables` (uploaded as zeros)
  int64_t blocks_ = 0, records_ = 0;
  // stage(): tables built and sent on `copy_st` (a stream that waits for nothing); launch(): kernels only, on a stream fenced behind copy_st
  // (snappy2.hpp says why: a host → device copy queued on a waiting stream holds the calling thread)
  void stage(const PqInflate* jobs_host, int njobs, const comet_zstd2::ZBlock* blocks_host, hipStream_t copy_st);
  void launch(uint8_t* bytes_dev, uint32_t* err_dev, hipStream_t st);
  int njobs_ = 0;
  int64_t nblocks_ = 0, nrecs_ = 0;
  size_t o_pages_ = 0, o_blocks_ = 0, o_bp_ = 0, o_ord_ = 0;
  // jobs: the pages (offsets into bytes_dev); job i's blocks — what the host walk over its frame found (comet_zstd2::scan_page) — are
  // blocks_host[jobs[i].preamble … + jobs[i].pad).  err_dev: one word, first failing page as (page << 8 | code).  Nothing is read back.
  void run(const PqInflate* jobs_host, int njobs, const comet_zstd2::ZBlock* blocks_host, uint8_t* bytes_dev, uint32_t* err_dev, hipStream_t st);
};

}  // namespace comet
