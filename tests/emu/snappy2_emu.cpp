// TEST INFRASTRUCTURE: the multi-kernel snappy pipeline (csrc/device/snappy2.hpp — the SAME source the gfx950 kernels compile) run on the
// host.  Every kernel there is a sequence of phases, a phase a plain function of the thread index; here the threads of a workgroup run one
// after the other inside each phase and the workgroups one after the other, with the workgroup memory as an ordinary struct.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "device/snappy2.hpp"

using namespace comet_snappy2;

extern "C" int64_t sn2_emu_inflate_pages(const uint8_t* streams, const int64_t* stream_off, const int32_t* stream_len, const int32_t* page_len, int32_t npages,
                                         uint8_t* out, const int64_t* out_off, uint32_t* status_out, int32_t* rounds_out) {
  auto up16 = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
  std::vector<i64> so((size_t)npages), dof((size_t)npages);
  std::vector<i32> body((size_t)npages);
  i64 in_total = 0, out_total = 0;
  for (int i = 0; i < npages; i++) { so[(size_t)i] = in_total; in_total = up16(in_total + stream_len[i]) + 16; }
  for (int i = 0; i < npages; i++) { dof[(size_t)i] = in_total + out_total; out_total = up16(out_total + page_len[i]) + 16; }
  std::vector<u8> bytes((size_t)(in_total + out_total) + 1024, 0);
  for (int i = 0; i < npages; i++) {
    memcpy(bytes.data() + so[(size_t)i], streams + stream_off[i], (size_t)stream_len[i]);
    body[(size_t)i] = preamble_length(streams + stream_off[i], stream_len[i]);
  }
  Plan pl = make_plan(so.data(), dof.data(), stream_len, page_len, body.data(), npages);
  std::vector<u32> status((size_t)npages, 0);
  std::vector<ChunkFn> fns((size_t)pl.nchunks * kWin);
  std::vector<ChunkIn> ins((size_t)pl.nchunks);
  std::vector<i32> frag_chunk((size_t)pl.nfrags + 1, 0);
  auto L = std::make_unique<ChunkLds>();
  // kernel A
  for (i64 c = 0; c < pl.nchunks; c++) {
    const Page& pg = pl.pages[(size_t)pl.chunk_page[(size_t)c]];
    const i64 chunk_pos = (i64)pg.body + (c - pg.chunk_first) * (i64)kChunk;
    for (int t = 0; t < kWins; t++) chunk_stage(L.get(), bytes.data() + pg.src_off, pg.src_len, chunk_pos, t);
    for (int t = 0; t < kWins; t++) chunk_tables(L.get(), pg.src_len, chunk_pos, t);
    for (int t = 0; t < kWins; t++) fns[(size_t)(c * kWin + t)] = chunk_compose(L.get(), t, (i64)pg.src_len - chunk_pos);
  }
  // kernel B
  for (int i = 0; i < npages; i++) {
    if (body[(size_t)i] == 0) { status[(size_t)i] = ST_ERR_PREAMBLE; continue; }
    page_chain(&pl.pages[(size_t)i], bytes.data(), fns.data(), ins.data(), frag_chunk.data(), status.data(), i);
  }
  i64 nelems = 0;
  for (int i = 0; i < npages; i++) {
    pl.pages[(size_t)i].elem_first = nelems;
    if (status[(size_t)i] == ST_OK) nelems += pl.pages[(size_t)i].nelems;
  }
  std::vector<Elem> elems((size_t)nelems + 1);
  // kernel C
  for (i64 c = 0; c < pl.nchunks; c++) {
    const int pi = pl.chunk_page[(size_t)c];
    if (status[(size_t)pi] != ST_OK) continue;
    const ChunkIn in = ins[(size_t)c];
    if (in.entry == kNoEntry) continue;
    const Page& pg = pl.pages[(size_t)pi];
    const i64 chunk_pos = (i64)pg.body + (c - pg.chunk_first) * (i64)kChunk;
    for (int t = 0; t < kWins; t++) chunk_stage(L.get(), bytes.data() + pg.src_off, pg.src_len, chunk_pos, t);
    for (int t = 0; t < kWins; t++) chunk_tables(L.get(), pg.src_len, chunk_pos, t);
    chunk_window_entries(L.get(), in, (i64)pg.src_len - chunk_pos);
    for (int t = 0; t < kWins; t++) chunk_emit(L.get(), chunk_pos, pg.src_len, elems.data() + pg.elem_first, t);
  }
  // kernel D
  auto X = std::make_unique<ExecLds>();
  int max_rounds = 0;
  for (i64 f = 0; f < pl.nfrags; f++) {
    const int pi = pl.frag_page[(size_t)f];
    if (status[(size_t)pi] != (u32)ST_OK) continue;
    const Page& pg = pl.pages[(size_t)pi];
    const u32 frag_out = (u32)(f - pg.frag_first) * (u32)kFrag;
    const u32 frag_end = frag_out + (u32)kFrag < (u32)pg.dst_len ? frag_out + (u32)kFrag : (u32)pg.dst_len;
    const u32 frag_len = frag_end - frag_out;
    const Elem* pe = elems.data() + pg.elem_first;
    memset(X->src, 0xee, sizeof X->src);          // (uninitialised on the device: make a coverage bug visible)
    X->nbig = X->covered = X->changed = X->flags = 0;
    X->lo = X->hi = 0xffffffffu;
    const i32 fl = (i32)(f - pg.frag_first);
    for (int t = 0; t < kExecThreads; t++)
      frag_range_search(X.get(), pe, pg.nelems, ins.data() + pg.chunk_first, pg.nchunks, frag_chunk[(size_t)f], fl + 1 < pg.nfrags ? frag_chunk[(size_t)f + 1] : -1, frag_out, frag_end, t, kExecThreads);
    const u32 lo = X->lo, hi = X->hi;
    const u8* src = bytes.data() + pg.src_off;
    u8* dst = bytes.data() + pg.dst_off;
    for (int t = 0; t < kExecThreads; t++) frag_scatter(X.get(), pe, lo, hi, frag_out, frag_end, src, dst, t, kExecThreads);
    for (int t = 0; t < kExecThreads; t++) frag_big_literals(X.get(), frag_out, src, dst, t, kExecThreads);
    if ((X->flags & 3u) || X->covered != frag_len) {
      const u32 code = (X->flags & 2u) ? (u32)ST_ERR_BAD_COPY : (X->flags & 1u) ? (u32)ST_FALLBACK : (u32)ST_ERR_LENGTH;
      if (code > status[(size_t)pi]) status[(size_t)pi] = code;
      continue;
    }
    if (!(X->flags & 4u)) continue;
    int round = 0;
    for (; round < 18; round++) {
      bool any = false;
      for (int t = 0; t < kExecThreads; t++) any |= frag_jump(X.get(), frag_len, t, kExecThreads);
      if (!any) break;
    }
    if (round > max_rounds) max_rounds = round;
    for (int t = 0; t < kExecThreads; t++) frag_resolve(X.get(), frag_out, frag_len, dst, t, kExecThreads);
  }
  for (int i = 0; i < npages; i++) {
    status_out[i] = status[(size_t)i];
    if (status[(size_t)i] == ST_OK && page_len[i]) memcpy(out + out_off[i], bytes.data() + dof[(size_t)i], (size_t)page_len[i]);
  }
  if (rounds_out) *rounds_out = max_rounds;
  return 0;
}
