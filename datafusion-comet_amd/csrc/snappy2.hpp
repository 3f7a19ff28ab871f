// Scratch buffers + launcher of the multi-kernel snappy pipeline (snappy2.cpp).  One object per column being scanned: its buffers must
// outlive the work queued on the stream (the scan keeps it with the column's other device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include "exec.hpp"
#include "parquet_dev.h"

namespace comet {

struct Snappy2Scratch {
  DevBuf tables, fns, ins, elems, status, frag_chunk;
  PinnedBuf h_tables;
  int64_t chunks_ = 0, frags_ = 0;
  // jobs: the pages to decompress (offsets into bytes_dev; `preamble` filled in by the host, which has seen the compressed bytes), in host
  // memory — uploaded with the pipeline's own tables; err_dev: one word, first failing page as (page << 8 | code)
  void run(const PqInflate* jobs_host, int njobs, uint8_t* bytes_dev, uint32_t* err_dev, hipStream_t st);
};

}  // namespace comet
