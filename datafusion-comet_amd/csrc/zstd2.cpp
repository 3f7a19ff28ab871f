// Host side of the zstd pipeline (device/zstd2.hpp, zstd2_kernels.hip): lays out the page / block tables the host walk produced, sizes the
// record and literal scratch, queues kernels A–D on the caller's stream — nothing is read back, so the scan stays asynchronous.
#include "zstd2.hpp"
#include "../../include/comet_amd.h"
#include <algorithm>

#include <hip/hip_runtime.h>

#include <cstring>

#include "device/zstd2.hpp"

extern "C" {
void zs2_launch_entropy(const void* pages, void* blocks, const int32_t* block_page, const int32_t* order, const uint8_t* bytes, uint8_t* lits, void* recs, uint32_t* status, int64_t nblocks, void* st);
void zs2_launch_blocks(const void* pages, int npages, void* blocks, uint32_t* status, void* st);
void zs2_launch_scan(const void* pages, const void* blocks, const int32_t* block_page, void* recs, uint32_t* status, int64_t nblocks, void* st);
void zs2_launch_exec(const void* pages, int npages, uint8_t* bytes, const uint8_t* lits, const void* recs, uint32_t* status, void* st);
void zs2_launch_report(const uint32_t* status, int npages, uint32_t* err, void* st);
}

namespace comet {

using namespace comet_zstd2;

void Zstd2Scratch::run(const PqInflate* jobs_host, int njobs, const ZBlock* blocks_host, uint8_t* bytes_dev, uint32_t* err_dev, hipStream_t st) {
  stage(jobs_host, njobs, blocks_host, st);
  launch(bytes_dev, err_dev, st);
}

void Zstd2Scratch::stage(const PqInflate* jobs_host, int njobs, const ZBlock* blocks_host, hipStream_t copy_st) {
  njobs_ = njobs;
  if (njobs <= 0) return;
  int64_t nblocks = 0, nrecs = 0, nlits = 0;
  for (int i = 0; i < njobs; i++) nblocks += jobs_host[i].pad;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t b_pages = sizeof(ZPage) * (size_t)njobs, b_blocks = sizeof(ZBlock) * (size_t)nblocks + 16, b_bp = 4 * (size_t)nblocks + 16;
  const size_t o_pages = 0, o_blocks = al(b_pages), o_bp = o_blocks + al(b_blocks), o_ord = o_bp + al(b_bp), o_st = o_ord + al(b_bp), total = o_st + al(4 * (size_t)njobs + 16);
  h_tables.ensure(total + 16);
  ZPage* P = (ZPage*)((char*)h_tables.p + o_pages);
  ZBlock* B = (ZBlock*)((char*)h_tables.p + o_blocks);
  int32_t* BP = (int32_t*)((char*)h_tables.p + o_bp);
  int32_t* ORD = (int32_t*)((char*)h_tables.p + o_ord);
  int64_t bi = 0;
  for (int i = 0; i < njobs; i++) {
    ZPage& pg = P[i];
    pg.src_off = jobs_host[i].src_off;
    pg.dst_off = jobs_host[i].dst_off;
    pg.src_len = jobs_host[i].src_len;
    pg.dst_len = jobs_host[i].dst_len;
    pg.block_first = (int32_t)bi;
    pg.nblocks = jobs_host[i].pad;
    pg.rec_first = nrecs;
    pg.lit_first = nlits;
    pg.pad = 0;
    uint32_t pr = 0, plit = 0;
    for (int k = 0; k < jobs_host[i].pad; k++) {
      const ZBlock& b = blocks_host[jobs_host[i].preamble + k];
      B[bi] = b;
      BP[bi] = i;
      bi++;
      pr += b.nseq + 1;
      plit += b.lit_regen;
    }
    pg.nrecs = pr;
    nrecs += pr;
    nlits += plit;
  }
  // The sequence kernel lasts as long as its longest chain, and a launch beyond the 4096 blocks a GPU holds at once runs in passes: longest
  // blocks first, so that the short ones fill the slots the long ones leave (counting sort by sequence count, 1 K sequences per bucket).
  {
    std::vector<int32_t> count(130, 0);
    auto bucket = [](const ZBlock& b) { return 128 - (int)std::min<uint32_t>(128, (b.type == 2 ? b.nseq : 0) >> 10); };
    for (int64_t k = 0; k < nblocks; k++) count[(size_t)bucket(B[k]) + 1]++;
    for (size_t k = 1; k < count.size(); k++) count[k] += count[k - 1];
    for (int64_t k = 0; k < nblocks; k++) ORD[count[(size_t)bucket(B[k])]++] = (int32_t)k;
  }
  memset((char*)h_tables.p + o_st, 0, 4 * (size_t)njobs + 16);      // the per-page status words start clear
  tables.ensure(total + 16);
  HIP_CHECK(hipMemcpyAsync(tables.p, h_tables.p, total, hipMemcpyHostToDevice, copy_st));
  recs.ensure(sizeof(ZRec) * (size_t)nrecs + 64);
  lits.ensure((size_t)nlits + 64);
  status = (uint32_t*)((char*)tables.p + o_st);
  nblocks_ = nblocks;
  nrecs_ = nrecs;
  o_pages_ = o_pages; o_blocks_ = o_blocks; o_bp_ = o_bp; o_ord_ = o_ord;
}

void Zstd2Scratch::launch(uint8_t* bytes_dev, uint32_t* err_dev, hipStream_t st) {
  if (njobs_ <= 0) return;
  char* tb = (char*)tables.p;
  zs2_launch_entropy(tb + o_pages_, tb + o_blocks_, (const int32_t*)(tb + o_bp_), (const int32_t*)(tb + o_ord_), bytes_dev, (uint8_t*)lits.p, recs.p, status, nblocks_, st);
  zs2_launch_blocks(tb + o_pages_, njobs_, tb + o_blocks_, status, st);
  zs2_launch_scan(tb + o_pages_, tb + o_blocks_, (const int32_t*)(tb + o_bp_), recs.p, status, nblocks_, st);
  zs2_launch_exec(tb + o_pages_, njobs_, bytes_dev, (const uint8_t*)lits.p, recs.p, status, st);
  zs2_launch_report((const uint32_t*)status, njobs_, err_dev, st);
  blocks_ += nblocks_;
  records_ += nrecs_;
}

}  // namespace comet

// Diagnostic / test entry (include/comet_amd.h): `npages` zstd frames from host memory → pages back in host memory, through the host walk
// and kernels A–D.  status_out[i] (optional): 0 decoded, 1 the host walk would not send this page to the device (it is left out of the
// launch and its output untouched), ≥ 16 corrupt.  Returns 0, (page << 8 | code) of the first failing page, or -1.
extern "C" int64_t comet_zstd2_inflate_pages(const uint8_t* streams, const int64_t* stream_off, const int32_t* stream_len, const int32_t* page_len,
                                             int32_t npages, uint8_t* out, const int64_t* out_off, int32_t device_id, double* kernel_ms, uint32_t* status_out) {
  using namespace comet;
  if (npages <= 0) return 0;
  try {
    HIP_CHECK(hipSetDevice(device_id));
    auto up16 = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
    std::vector<PqInflate> jobs;
    std::vector<int> job_page;
    std::vector<ZBlock> blocks;
    std::vector<int64_t> src_off((size_t)npages), dst_off((size_t)npages);
    int64_t in_total = 0, out_total = 0;
    for (int i = 0; i < npages; i++) { src_off[(size_t)i] = in_total; in_total = up16(in_total + stream_len[i]) + 16; }
    for (int i = 0; i < npages; i++) { dst_off[(size_t)i] = in_total + out_total; out_total = up16(out_total + page_len[i]) + 16; }
    for (int i = 0; i < npages; i++) {
      PageWalk w;
      if (status_out) status_out[i] = 1;
      if (!scan_page(streams + stream_off[i], (uint32_t)stream_len[i], (uint32_t)page_len[i], w)) continue;
      PqInflate j;
      j.src_off = src_off[(size_t)i];
      j.dst_off = dst_off[(size_t)i];
      j.src_len = stream_len[i];
      j.dst_len = page_len[i];
      j.preamble = (int32_t)blocks.size();
      j.pad = (int32_t)w.blocks.size();
      blocks.insert(blocks.end(), w.blocks.begin(), w.blocks.end());
      jobs.push_back(j);
      job_page.push_back(i);
    }
    if (jobs.empty()) return 0;
    DevBuf bytes, derr;
    bytes.ensure((size_t)(in_total + out_total) + 1024);
    derr.ensure(64);
    hipStream_t st = nullptr;
    HIP_CHECK(hipMemset(bytes.p, 0, (size_t)in_total));
    HIP_CHECK(hipMemset(derr.p, 0, 4));
    for (int i = 0; i < npages; i++)
      if (stream_len[i]) HIP_CHECK(hipMemcpy((char*)bytes.p + src_off[(size_t)i], streams + stream_off[i], (size_t)stream_len[i], hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    Zstd2Scratch sc;
    if (kernel_ms) {       // the first call sizes the scratch buffers (allocation is not decompression): run once untimed
      sc.run(jobs.data(), (int)jobs.size(), blocks.data(), (uint8_t*)bytes.p, (uint32_t*)derr.p, st);
      HIP_CHECK(hipStreamSynchronize(st));
      HIP_CHECK(hipMemset(derr.p, 0, 4));
    }
    HIP_CHECK(hipEventRecord(e0, st));
    sc.run(jobs.data(), (int)jobs.size(), blocks.data(), (uint8_t*)bytes.p, (uint32_t*)derr.p, st);
    HIP_CHECK(hipEventRecord(e1, st));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (kernel_ms) *kernel_ms = (double)ms;
    std::vector<uint32_t> st_host(jobs.size());
    HIP_CHECK(hipMemcpy(st_host.data(), sc.status, 4 * jobs.size(), hipMemcpyDeviceToHost));
    int64_t rc = 0;
    for (size_t k = 0; k < jobs.size(); k++) {
      const int i = job_page[k];
      if (status_out) status_out[i] = st_host[k];
      if (st_host[k] != 0) { if (!rc) rc = ((int64_t)i << 8) | st_host[k]; continue; }
      if (page_len[i]) HIP_CHECK(hipMemcpy(out + out_off[i], (char*)bytes.p + dst_off[(size_t)i], (size_t)page_len[i], hipMemcpyDeviceToHost));
    }
    return rc;
  } catch (const std::exception&) {
    return -1;
  }
}
