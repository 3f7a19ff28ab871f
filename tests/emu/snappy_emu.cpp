// Host emulation of one 64-lane wavefront running csrc/device/snappy_inflate.hpp — TEST INFRASTRUCTURE for the CPU-only suite.
// Each lane is a coroutine (ucontext); every cross-lane primitive (ballot, readlane, scan, lds_sync) is a rendezvous of all 64 lanes, so
// a lane sees another lane's LDS / slot writes only across a primitive — stricter than the hardware's lockstep, never laxer: code that is
// correct here does not depend on instruction-level lockstep.  Built by tests/test_snappy_emu_cpu.py with g++.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <stdio.h>
#define SNAPPY_TRACE(lane, p, q, o, total, starts, is_start, kind, len, off, opos) \
  do { if (getenv("SNAPPY_EMU_TRACE") && (o) + (int)(total) > atoi(getenv("SNAPPY_EMU_TRACE")) && (o) <= atoi(getenv("SNAPPY_EMU_TRACE")) && (is_start)) \
    fprintf(stderr, "win p=%d q=%d o=%d total=%u lane=%d kind=%u len=%u off=%u opos=%u\n", p, q, o, total, lane, kind, len, off, opos); } while (0)
#include "device/snappy_inflate.hpp"

namespace {
using namespace comet_snappy;

struct Sched {
  ucontext_t main_ctx;
  ucontext_t lane_ctx[64];
  char* stacks[64];
  bool finished[64];
  int current = 0;
  uint32_t slot[2][64];
  int result[64];
  // job
  Lds* lds;
  const u8* src;
  int src_len;
  u8* dst;
  int dst_len;
};
Sched* g = nullptr;

struct EmuWave {
  int l;
  unsigned gen = 0;
  int lane() const { return l; }
  void sync() { swapcontext(&g->lane_ctx[l], &g->main_ctx); }   // the scheduler resumes this lane once every lane has arrived
  u64 ballot(bool b) {
    uint32_t* s = g->slot[gen++ & 1];
    s[l] = b ? 1u : 0u;
    sync();
    u64 m = 0;
    for (int i = 0; i < 64; i++) m |= (u64)(s[i] != 0) << i;
    return m;
  }
  u32 readlane(u32 v, int srclane) {
    uint32_t* s = g->slot[gen++ & 1];
    s[l] = v;
    sync();
    return s[srclane & 63];
  }
  u32 incl_scan_add(u32 v) {
    uint32_t* s = g->slot[gen++ & 1];
    s[l] = v;
    sync();
    u32 acc = 0;
    for (int i = 0; i <= l; i++) acc += s[i];
    return acc;
  }
  u32 gather(u32 v, u32 srclane) {
    uint32_t* s = g->slot[gen++ & 1];
    s[l] = v;
    sync();
    return s[srclane & 63];
  }
  u32 wave_min(u32 v) {
    uint32_t* s = g->slot[gen++ & 1];
    s[l] = v;
    sync();
    u32 m = s[0];
    for (int i = 1; i < 64; i++) m = s[i] < m ? s[i] : m;
    return m;
  }
  void lds_sync() { sync(); }
  void release_stores() { sync(); }
  int ctz64(u64 m) const { return __builtin_ctzll(m); }
  u8 load_coherent_byte(const u8* p) const { return *p; }
};

void lane_main(int l) {
  EmuWave w;
  w.l = l;
  g->result[l] = inflate_page(w, g->lds, g->src, g->src_len, g->dst, g->dst_len);
  g->finished[l] = true;
  swapcontext(&g->lane_ctx[l], &g->main_ctx);
}
}  // namespace

// returns the wave's (uniform) error code; -1 if the lanes disagree (a bug in the kernel's uniformity)
extern "C" int emu_snappy_inflate(const uint8_t* src, int src_len, uint8_t* dst, int dst_len) {
  Sched s;
  g = &s;
  s.lds = (Lds*)aligned_alloc(16, sizeof(Lds));
  memset(s.lds, 0xCD, sizeof(Lds));
  // the kernel reads the source up to the next multiple of 16 and wants both buffers 16-byte aligned
  const int src16 = (src_len + 15) & ~15;
  u8* srcbuf = (u8*)aligned_alloc(16, (size_t)src16 + 16);
  memset(srcbuf, 0, (size_t)src16 + 16);
  memcpy(srcbuf, src, (size_t)src_len);
  u8* dstbuf = (u8*)aligned_alloc(16, (((size_t)dst_len + 15) & ~(size_t)15) + 1024 + 16);
  s.src = srcbuf;
  s.src_len = src_len;
  s.dst = dstbuf;
  s.dst_len = dst_len;
  const size_t kStack = 256 * 1024;
  for (int l = 0; l < 64; l++) {
    s.finished[l] = false;
    s.stacks[l] = (char*)malloc(kStack);
    getcontext(&s.lane_ctx[l]);
    s.lane_ctx[l].uc_stack.ss_sp = s.stacks[l];
    s.lane_ctx[l].uc_stack.ss_size = kStack;
    s.lane_ctx[l].uc_link = &s.main_ctx;
    makecontext(&s.lane_ctx[l], (void (*)())lane_main, 1, l);
  }
  for (;;) {
    int alive = 0;
    for (int l = 0; l < 64; l++) {
      if (s.finished[l]) continue;
      alive++;
      swapcontext(&s.main_ctx, &s.lane_ctx[l]);   // runs lane l up to its next rendezvous
    }
    if (!alive) break;
  }
  int rc = s.result[0];
  for (int l = 1; l < 64; l++)
    if (s.result[l] != rc) rc = -1;
  if (rc == 0) memcpy(dst, dstbuf, (size_t)dst_len);
  for (int l = 0; l < 64; l++) free(s.stacks[l]);
  free(s.lds);
  free(srcbuf);
  free(dstbuf);
  g = nullptr;
  return rc;
}
