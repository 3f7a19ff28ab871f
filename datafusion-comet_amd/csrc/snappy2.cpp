// Host side of the multi-kernel snappy pipeline (device/snappy2.hpp, snappy2_kernels.hip): builds the page / chunk / fragment tables,
// sizes the scratch buffers, queues kernels A–D on the caller's stream — nothing is read back, so the scan stays asynchronous — and lets
// the one-wave kernel (snappy_kernels.hip) pick up the pages the pipeline flagged as not fragment-shaped.
#include <hip/hip_runtime.h>

#include <cstring>

#include "device/snappy2.hpp"
#include "exec.hpp"
#include "parquet_dev.h"
#include "snappy2.hpp"
#include "../../include/comet_amd.h"

extern "C" {
void sn2_launch_window(const void* pages, const int32_t* chunk_page, const uint8_t* bytes, void* fns, const uint32_t* status, int64_t nchunks, void* st);
void sn2_launch_chain(void* pages, int npages, const uint8_t* bytes, const void* fns, void* ins, int32_t* frag_chunk, uint32_t* status, void* st);
void sn2_launch_emit(const void* pages, const int32_t* chunk_page, const uint8_t* bytes, const void* ins, void* elems, const uint32_t* status, int64_t nchunks, void* st);
void sn2_launch_exec(const void* pages, const int32_t* frag_page, uint8_t* bytes, const void* elems, const void* ins, const int32_t* frag_chunk, uint32_t* status, int64_t nfrags, void* st);
void pq_launch_snappy_fallback(const PqInflate* jobs, int njobs, uint8_t* bytes, const uint32_t* status, uint32_t* err, void* st);
}

namespace comet {

using namespace comet_snappy2;

void Snappy2Scratch::run(const PqInflate* jobs_host, int njobs, uint8_t* bytes_dev, uint32_t* err_dev, hipStream_t st) {
  stage(jobs_host, njobs, st);
  launch(bytes_dev, err_dev, st);
}

void Snappy2Scratch::stage(const PqInflate* jobs_host, int njobs, hipStream_t copy_st) {
  njobs_ = njobs;
  if (njobs <= 0) return;
  std::vector<i64> so((size_t)njobs), dof((size_t)njobs);
  std::vector<i32> sl((size_t)njobs), dl((size_t)njobs), body((size_t)njobs);
  for (int i = 0; i < njobs; i++) {
    so[(size_t)i] = jobs_host[i].src_off;
    dof[(size_t)i] = jobs_host[i].dst_off;
    sl[(size_t)i] = jobs_host[i].src_len;
    dl[(size_t)i] = jobs_host[i].dst_len;
    body[(size_t)i] = jobs_host[i].preamble;
  }
  Plan pl = make_plan(so.data(), dof.data(), sl.data(), dl.data(), body.data(), njobs);
  // elements: at most one per two compressed bytes (the shortest elements — a one-byte literal, a two-byte copy — take two)
  i64 nelems = 0;
  for (int i = 0; i < njobs; i++) {
    pl.pages[(size_t)i].elem_first = nelems;
    nelems += (i64)sl[(size_t)i] / 2 + 2;
  }
  const size_t b_pages = sizeof(Page) * (size_t)njobs, b_cp = 4 * (size_t)pl.nchunks + 16, b_fp = 4 * (size_t)pl.nfrags + 16, b_st = 4 * (size_t)njobs + 16;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t b_jobs = sizeof(PqInflate) * (size_t)njobs;
  const size_t o_pages = 0, o_cp = al(b_pages), o_fp = o_cp + al(b_cp), o_st = o_fp + al(b_fp), o_jobs = o_st + al(b_st), total = o_jobs + al(b_jobs);
  h_tables.ensure(total + 16);
  memcpy((char*)h_tables.p + o_jobs, jobs_host, b_jobs);
  // a page that did not compress (doubles, random keys) is one long literal per 64 KiB block: the one-wave kernel copies those at
  // hundreds of GB/s and there is nothing for the pipeline to parallelise — routed there up front
  uint32_t* st0 = (uint32_t*)((char*)h_tables.p + o_st);
  for (int i = 0; i < njobs; i++) st0[i] = ((i64)sl[(size_t)i] * 100 >= (i64)dl[(size_t)i] * 97 && dl[(size_t)i] >= 4096) ? (uint32_t)ST_FALLBACK : (uint32_t)ST_OK;
  memcpy((char*)h_tables.p + o_pages, pl.pages.data(), b_pages);
  if (pl.nchunks) memcpy((char*)h_tables.p + o_cp, pl.chunk_page.data(), 4 * (size_t)pl.nchunks);
  if (pl.nfrags) memcpy((char*)h_tables.p + o_fp, pl.frag_page.data(), 4 * (size_t)pl.nfrags);
  tables.ensure(total + 16);
  HIP_CHECK(hipMemcpyAsync(tables.p, h_tables.p, total, hipMemcpyHostToDevice, copy_st));
  fns.ensure(sizeof(ChunkFn) * (size_t)pl.nchunks * kWin + 16);
  ins.ensure(sizeof(ChunkIn) * (size_t)pl.nchunks + 16);
  elems.ensure(sizeof(Elem) * (size_t)nelems + 16);
  frag_chunk.ensure(4 * (size_t)pl.nfrags + 16);
  status = (uint32_t*)((char*)tables.p + o_st);      // the per-page status words live where their initial values were uploaded
  nchunks_ = pl.nchunks;
  nfrags_ = pl.nfrags;
  o_pages_ = o_pages; o_cp_ = o_cp; o_fp_ = o_fp; o_st_ = o_st; o_jobs_ = o_jobs;
}

void Snappy2Scratch::launch(uint8_t* bytes_dev, uint32_t* err_dev, hipStream_t st) {
  if (njobs_ <= 0) return;
  char* tb = (char*)tables.p;
  sn2_launch_window(tb + o_pages_, (const int32_t*)(tb + o_cp_), bytes_dev, fns.p, (const uint32_t*)status, nchunks_, st);
  sn2_launch_chain(tb + o_pages_, njobs_, bytes_dev, fns.p, ins.p, (int32_t*)frag_chunk.p, status, st);
  sn2_launch_emit(tb + o_pages_, (const int32_t*)(tb + o_cp_), bytes_dev, ins.p, elems.p, (const uint32_t*)status, nchunks_, st);
  sn2_launch_exec(tb + o_pages_, (const int32_t*)(tb + o_fp_), bytes_dev, elems.p, ins.p, (const int32_t*)frag_chunk.p, status, nfrags_, st);
  // what the pipeline would not decode — legal streams that are not fragment-shaped — goes to the one-wave kernel; errors to `err`
  pq_launch_snappy_fallback((const PqInflate*)(tb + o_jobs_), njobs_, bytes_dev, (const uint32_t*)status, err_dev, st);
  chunks_ += nchunks_;
  frags_ += nfrags_;
}

}  // namespace comet

// Diagnostic / test entry (include/comet_amd.h), the counterpart of comet_snappy_inflate_pages for the multi-kernel pipeline: `npages` raw
// snappy streams from host memory → pages back in host memory.  status_out[i] (optional): what the PIPELINE made of page i before the
// fallback (0 decoded, 1 handed to the one-wave kernel, ≥ 16 corrupt).  Returns 0, (page << 8 | code) of the first failing page, or -1.
extern "C" int64_t comet_snappy2_inflate_pages(const uint8_t* streams, const int64_t* stream_off, const int32_t* stream_len, const int32_t* page_len,
                                               int32_t npages, uint8_t* out, const int64_t* out_off, int32_t device_id, double* kernel_ms, uint32_t* status_out) {
  using namespace comet;
  if (npages <= 0) return 0;
  try {
    HIP_CHECK(hipSetDevice(device_id));
    auto up16 = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
    std::vector<PqInflate> jobs((size_t)npages);
    int64_t in_total = 0, out_total = 0;
    for (int i = 0; i < npages; i++) {
      jobs[(size_t)i].src_off = in_total;
      jobs[(size_t)i].src_len = stream_len[i];
      jobs[(size_t)i].preamble = comet_snappy2::preamble_length(streams + stream_off[i], stream_len[i]);
      jobs[(size_t)i].pad = 0;
      in_total = up16(in_total + stream_len[i]) + 16;
    }
    for (int i = 0; i < npages; i++) {
      jobs[(size_t)i].dst_off = in_total + out_total;
      jobs[(size_t)i].dst_len = page_len[i];
      out_total = up16(out_total + page_len[i]) + 16;
    }
    DevBuf bytes, djobs, derr;
    bytes.ensure((size_t)(in_total + out_total) + 1024);
    djobs.ensure(sizeof(PqInflate) * (size_t)npages + 16);
    derr.ensure(64);
    hipStream_t st = nullptr;
    HIP_CHECK(hipMemset(bytes.p, 0, (size_t)in_total));
    HIP_CHECK(hipMemset(derr.p, 0, 4));
    for (int i = 0; i < npages; i++)
      if (stream_len[i]) HIP_CHECK(hipMemcpy((char*)bytes.p + jobs[(size_t)i].src_off, streams + stream_off[i], (size_t)stream_len[i], hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(djobs.p, jobs.data(), sizeof(PqInflate) * (size_t)npages, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    Snappy2Scratch sc;
    // the first call sizes the scratch buffers (allocation is not decompression): run once untimed when a time is asked for
    if (kernel_ms) {
      sc.run(jobs.data(), npages, (uint8_t*)bytes.p, (uint32_t*)derr.p, st);
      HIP_CHECK(hipStreamSynchronize(st));
      HIP_CHECK(hipMemset(derr.p, 0, 4));
    }
    HIP_CHECK(hipEventRecord(e0, st));
    sc.run(jobs.data(), npages, (uint8_t*)bytes.p, (uint32_t*)derr.p, st);
    HIP_CHECK(hipEventRecord(e1, st));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (kernel_ms) *kernel_ms = (double)ms;
    if (status_out) HIP_CHECK(hipMemcpy(status_out, sc.status, 4 * (size_t)npages, hipMemcpyDeviceToHost));
    uint32_t h_err = 0;
    HIP_CHECK(hipMemcpy(&h_err, derr.p, 4, hipMemcpyDeviceToHost));
    if (h_err) return (int64_t)h_err;
    for (int i = 0; i < npages; i++)
      if (page_len[i]) HIP_CHECK(hipMemcpy(out + out_off[i], (char*)bytes.p + jobs[(size_t)i].dst_off, (size_t)page_len[i], hipMemcpyDeviceToHost));
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}
