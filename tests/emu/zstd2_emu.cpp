// TEST INFRASTRUCTURE: the zstd pipeline (csrc/device/zstd2.hpp — the SAME source the gfx950 kernels compile) run on the host: the threads
// of a workgroup run one after the other inside each phase, the workgroups one after the other, workgroup memory is an ordinary struct.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "device/zstd2.hpp"

using namespace comet_zstd2;

// → 0; status_out[i]: 0 decoded, 1 the host walk would not send the page to the device, ≥ 16 corrupt
extern "C" int64_t zs2_emu_inflate_pages(const uint8_t* streams, const int64_t* stream_off, const int32_t* stream_len, const int32_t* page_len, int32_t npages,
                                         uint8_t* out, const int64_t* out_off, uint32_t* status_out, int32_t* info_out /* [0] max jump rounds, [1] blocks, [2] records, [3] literal bytes, [4 … 19] PageWalk::seen summed */) {
  auto up16 = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
  std::vector<ZPage> pages((size_t)npages);
  std::vector<ZBlock> blocks;
  std::vector<i32> block_page;
  std::vector<u32> status((size_t)npages, 0);
  i64 in_total = 0, out_total = 0, nrecs = 0, nlits = 0;
  i64 seen[16] = {0};
  for (int i = 0; i < npages; i++) { pages[(size_t)i].src_off = in_total; in_total = up16(in_total + stream_len[i]) + 16; }
  for (int i = 0; i < npages; i++) { pages[(size_t)i].dst_off = in_total + out_total; out_total = up16(out_total + page_len[i]) + 16; }
  std::vector<u8> bytes((size_t)(in_total + out_total) + 1024, 0);
  for (int i = 0; i < npages; i++) {
    ZPage& pg = pages[(size_t)i];
    memcpy(bytes.data() + pg.src_off, streams + stream_off[i], (size_t)stream_len[i]);
    pg.src_len = stream_len[i];
    pg.dst_len = page_len[i];
    PageWalk w;
    pg.block_first = (i32)blocks.size();
    pg.nblocks = 0;
    pg.nrecs = 0;
    pg.rec_first = nrecs;
    pg.lit_first = nlits;
    if (!scan_page(streams + stream_off[i], (u32)stream_len[i], (u32)page_len[i], w)) { status[(size_t)i] = 1; continue; }
    pg.nblocks = (i32)w.blocks.size();
    for (int k = 0; k < 16; k++) seen[k] += w.seen[k];
    for (const ZBlock& b : w.blocks) { blocks.push_back(b); block_page.push_back(i); }
    pg.nrecs = w.nrecs;
    nrecs += w.nrecs;
    nlits += w.nlits;
  }
  std::vector<ZRec> recs((size_t)nrecs + 1);
  std::vector<u8> lits((size_t)nlits + 64, 0xcd);
  // kernel A1: literals
  auto LL = std::make_unique<LitLds>();
  for (size_t bi = 0; bi < blocks.size(); bi++) {
    const int pi = block_page[bi];
    const ZPage& pg = pages[(size_t)pi];
    const ZBlock& b = blocks[bi];
    const u8* src = bytes.data() + pg.src_off;
    u8* pl = lits.data() + pg.lit_first;
    memset(LL.get(), 0xee, sizeof(LitLds));
    LL->status = 0;
    LL->huf_log = 0;
    LL->nstreams = 0;
    const u32 rounds = lit_rounds(b);
    if (!rounds) {
      for (int t = 0; t < 64; t++) lit_plain(src, b, pl, t);
      continue;
    }
    for (int t = 0; t < 64; t++) lit_stage(LL.get(), src, b, t);
    lit_table(LL.get(), b);
    for (int t = 0; t < 64; t++) lit_fill(LL.get(), src, b, pg.src_len, t);
    LitState st[4];
    const int ns = (int)LL->nstreams;
    for (int k = 0; k < ns; k++) { lit_fill_done(LL.get(), k); lit_begin(LL.get(), st[k], k); }
    for (u32 r = 0; r < rounds && !LL->status; r++) {
      for (int k = 0; k < ns; k++) lit_round(LL.get(), st[k], k);
      if (LL->status) break;
      for (int t = 0; t < 64; t++) lit_flush(LL.get(), b, pl, r, t);
      for (int t = 0; t < 64; t++) lit_fill(LL.get(), src, b, pg.src_len, t);
      for (int k = 0; k < ns; k++) lit_fill_done(LL.get(), k);
    }
    if (LL->status && LL->status > status[(size_t)pi]) status[(size_t)pi] = LL->status;
  }
  // kernel A2: sequences, four blocks per workgroup (16 threads each; a group's thread 0 decodes)
  auto SL = std::make_unique<SeqLds>();
  for (size_t b0 = 0; b0 < blocks.size(); b0 += kSeqLanes) {
    memset(SL.get(), 0xee, sizeof(SeqLds));
    ZBlock none;
    memset(&none, 0, sizeof none);
    none.type = 3;
    ZPage nopage;
    memset(&nopage, 0, sizeof nopage);
    auto blk = [&](int k) -> ZBlock& { return b0 + (size_t)k < blocks.size() ? blocks[b0 + (size_t)k] : none; };
    auto page = [&](int k) -> const ZPage& { return b0 + (size_t)k < blocks.size() ? pages[(size_t)block_page[b0 + (size_t)k]] : nopage; };
    auto srcp = [&](int k) { return bytes.data() + page(k).src_off; };
    auto recp = [&](int k) { return recs.data() + page(k).rec_first + blk(k).rec_first; };
    for (int t = 0; t < 64; t++) seq_stage(SL.get(), t / kSeqGroup, srcp(t / kSeqGroup), blk(t / kSeqGroup), (u32)page(t / kSeqGroup).src_len, t % kSeqGroup);
    for (int k = 0; k < kSeqLanes; k++) seq_tables(SL.get(), k, blk(k));
    for (int t = 0; t < 64; t++) seq_fill(SL.get(), t / kSeqGroup, srcp(t / kSeqGroup), blk(t / kSeqGroup), page(t / kSeqGroup).src_len, t % kSeqGroup);
    SeqState st[kSeqLanes];
    u32 rounds = 0;
    for (int k = 0; k < kSeqLanes; k++) {
      if (seq_block_has_stream(blk(k))) { seq_fill_done(SL.get(), k); seq_start(SL.get(), k, st[k], blk(k)); }
      rounds = seq_rounds_of(SL.get(), k) > rounds ? seq_rounds_of(SL.get(), k) : rounds;
    }
    for (u32 r = 0; r < rounds; r++) {
      for (int k = 0; k < kSeqLanes; k++) seq_chain_round(SL.get(), k, st[k], blk(k));
      for (int t = 0; t < 64; t++) seq_values(SL.get(), t / kSeqGroup, t % kSeqGroup);
      for (int t = 0; t < 64; t++) seq_history_local(SL.get(), t / kSeqGroup, t % kSeqGroup);
      for (int p = 0; p < kSeqScanSteps; p++)
        for (int t = 0; t < 64; t++) seq_history_step(SL.get(), t / kSeqGroup, p, t % kSeqGroup);
      for (int t = 0; t < 64; t++) seq_history_apply(SL.get(), t / kSeqGroup, r, recp(t / kSeqGroup), t % kSeqGroup);
      for (int k = 0; k < kSeqLanes; k++) seq_round_check(SL.get(), k, r);
      for (int t = 0; t < 64; t++) seq_fill(SL.get(), t / kSeqGroup, srcp(t / kSeqGroup), blk(t / kSeqGroup), page(t / kSeqGroup).src_len, t % kSeqGroup);
      for (int k = 0; k < kSeqLanes; k++) seq_fill_done(SL.get(), k);
    }
    for (int k = 0; k < kSeqLanes && b0 + (size_t)k < blocks.size(); k++) {
      seq_finish(SL.get(), k, st[k], &blk(k), recp(k), rounds);
      const int pi = block_page[b0 + (size_t)k];
      if (seq_status(SL.get(), k) && seq_status(SL.get(), k) > status[(size_t)pi]) status[(size_t)pi] = seq_status(SL.get(), k);
    }
  }
  // kernel B
  for (int i = 0; i < npages; i++)
    if (status[(size_t)i] == ST_OK) page_blocks(pages[(size_t)i], blocks.data(), status.data(), i);
  // kernel C
  for (size_t bi = 0; bi < blocks.size(); bi++) {
    const int pi = block_page[bi];
    if (status[(size_t)pi] != ST_OK) continue;
    const ZPage& pg = pages[(size_t)pi];
    const ZBlock& b = blocks[bi];
    u32 st = 0;
    for (int t = 0; t < kScanThreads; t++) { const u32 x = fix_records(recs.data() + pg.rec_first + b.rec_first, b.nseq + 1, b, t, kScanThreads); st = x > st ? x : st; }
    if (st && st > status[(size_t)pi]) status[(size_t)pi] = st;
  }
  // kernel D
  auto X = std::make_unique<ZExecLds>();
  int max_rounds = 0;
  for (int pi = 0; pi < npages; pi++) {
    if (status[(size_t)pi] != ST_OK) continue;
    const ZPage& pg = pages[(size_t)pi];
    const ZRec* pr = recs.data() + pg.rec_first;
    const u32 nrec = pg.nrecs;
    u8* dst = bytes.data() + pg.dst_off;
    const u8* pl = lits.data() + pg.lit_first;
    u32 lo = 0;
    bool bad = false;
    for (u32 f0 = 0; f0 < (u32)pg.dst_len && !bad; f0 += kFrag) {
      const u32 f1 = f0 + kFrag < (u32)pg.dst_len ? f0 + kFrag : (u32)pg.dst_len, frag_len = f1 - f0;
      memset(X->e.src, 0xee, sizeof X->e.src);
      X->e.covered = X->e.changed = X->e.flags = 0;
      X->nq = 0;
      X->next_lo = nrec;
      for (int t = 0; t < kExecThreads; t++) zfrag_scatter(X.get(), pr, nrec, lo, f0, f1, pl, dst, t, kExecThreads);
      for (int t = 0; t < kExecThreads; t++) zfrag_long_parts(X.get(), f0, pl, dst, t, kExecThreads);
      if ((X->e.flags & 3u) || X->e.covered != frag_len) {
        status[(size_t)pi] = (X->e.flags & 2u) ? (u32)ST_ERR_OFFSET : (u32)ST_ERR_LENGTH;
        bad = true;
        break;
      }
      lo = X->next_lo;
      if (!(X->e.flags & 4u)) continue;
      int round = 0;
      for (; round < 20; round++) {
        bool any = false;
        for (int t = 0; t < kExecThreads; t++) any |= comet_snappy2::frag_jump(&X->e, frag_len, t, kExecThreads);
        if (!any) break;
      }
      if (round > max_rounds) max_rounds = round;
      for (int t = 0; t < kExecThreads; t++) comet_snappy2::frag_resolve(&X->e, f0, frag_len, dst, t, kExecThreads);
    }
  }
  for (int i = 0; i < npages; i++) {
    status_out[i] = status[(size_t)i];
    if (status[(size_t)i] == ST_OK && page_len[i]) memcpy(out + out_off[i], bytes.data() + pages[(size_t)i].dst_off, (size_t)page_len[i]);
  }
  if (info_out) { info_out[0] = max_rounds; info_out[1] = (int32_t)blocks.size(); info_out[2] = (int32_t)nrecs; info_out[3] = (int32_t)nlits; for (int k = 0; k < 16; k++) info_out[4 + k] = (int32_t)seen[k]; }
  return 0;
}

// the host's look at a page's first bytes (what the scan does for a v1 page's definition levels); → bytes produced, -1: the walk refuses the page
extern "C" int64_t zs2_emu_host_prefix(const uint8_t* stream, int32_t len, int32_t page_len, uint8_t* out, int64_t n) {
  PageWalk w;
  if (!scan_page(stream, (u32)len, (u32)page_len, w)) return -1;
  return (int64_t)host_prefix(stream, (u32)len, w, out, (size_t)n);
}
