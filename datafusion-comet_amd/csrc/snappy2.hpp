// Scratch buffers + launcher of the multi-kernel snappy pipeline (snappy2.cpp).  One object per column being scanned: its buffers must
// outlive the work queued on the stream (the scan keeps it with the column's other device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include "exec.hpp"
#include "parquet_dev.h"

namespace comet {

struct Snappy2Scratch {
  DevBuf tables, fns, ins, elems, frag_chunk;
  PinnedBuf h_tables;
  uint32_t* status = nullptr;      // per page, inside `tables` (uploaded with its initial values: no copy or memset on the compute stream)
  int64_t chunks_ = 0, frags_ = 0;
  // Two steps, because a host → device copy must never be queued on a stream that is waiting for another stream's event: the runtime then
  // holds the CALLING THREAD until that event has happened (measured: 7–10 ms per decompression group with eight concurrent scans, the time
  // the group's own page bytes needed to cross PCIe — profiles/r5_executor_trace.txt).  stage(): tables built and sent on `copy_st`, a stream
  // that waits for nothing; launch(): kernels only, on a stream the caller has fenced behind copy_st.
  void stage(const PqInflate* jobs_host, int njobs, hipStream_t copy_st);
  void launch(uint8_t* bytes_dev, uint32_t* err_dev, hipStream_t st);
  int njobs_ = 0;
  int64_t nchunks_ = 0, nfrags_ = 0;
  size_t o_pages_ = 0, o_cp_ = 0, o_fp_ = 0, o_st_ = 0, o_jobs_ = 0;
  // jobs: the pages to decompress (offsets into bytes_dev; `preamble` filled in by the host, which has seen the compressed bytes), in host
  // memory — uploaded with the pipeline's own tables; err_dev: one word, first failing page as (page << 8 | code)
  void run(const PqInflate* jobs_host, int njobs, uint8_t* bytes_dev, uint32_t* err_dev, hipStream_t st);
};

}  // namespace comet
