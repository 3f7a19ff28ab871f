// TEST INFRASTRUCTURE: damaged zstd frames through the host walk, the host prefix decoder and the emulated device pipeline, built with
// -fsanitize=address,undefined by tests/test_zstd2_emu_cpu.py: whatever the bytes say, nothing may read or write outside its buffers.
// Input file: repeated (u32 stream length, u32 page length, stream bytes).  argv: file, mutations per frame, seed.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int64_t zs2_emu_inflate_pages(const uint8_t* streams, const int64_t* stream_off, const int32_t* stream_len, const int32_t* page_len, int32_t npages,
                                         uint8_t* out, const int64_t* out_off, uint32_t* status_out, int32_t* info_out);
extern "C" int64_t zs2_emu_host_prefix(const uint8_t* stream, int32_t len, int32_t page_len, uint8_t* out, int64_t n);

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  const int per = atoi(argv[2]);
  uint64_t x = (uint64_t)atoll(argv[3]) * 0x9E3779B97F4A7C15ull + 1;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  long ran = 0, accepted = 0, corrupt = 0, refused = 0;
  for (;;) {
    uint32_t hdr[2];
    if (fread(hdr, 4, 2, f) != 2) break;
    std::vector<uint8_t> orig(hdr[0]);
    if (hdr[0] && fread(orig.data(), 1, hdr[0], f) != hdr[0]) return 2;
    for (int m = 0; m < per; m++) {
      std::vector<uint8_t> s = orig;       // exact size: ASan sees any read past the stream (the emulation copies it into a padded buffer like the scan does)
      const int kind = (int)(rnd() % 5);
      if (kind == 0 && !s.empty()) s[rnd() % s.size()] ^= (uint8_t)(1u << (rnd() % 8));
      else if (kind == 1 && !s.empty()) for (int k = 0; k < 4; k++) s[rnd() % s.size()] = (uint8_t)rnd();
      else if (kind == 2 && s.size() > 8) s.resize(8 + rnd() % (s.size() - 8));
      else if (kind == 3 && s.size() > 16) { const size_t a = rnd() % s.size(), b = rnd() % s.size(), n = rnd() % 32; for (size_t k = 0; k < n && a + k < s.size() && b + k < s.size(); k++) s[a + k] = s[b + k]; }
      else if (!s.empty()) { const size_t a = rnd() % s.size(); for (size_t k = a; k < s.size() && k < a + 3; k++) s[k] = (uint8_t)(rnd() % 2 ? 0 : 0xff); }
      int32_t plen = (int32_t)hdr[1];
      if (rnd() % 16 == 0) plen = (int32_t)(rnd() % (2 * (uint64_t)hdr[1] + 2));
      const int64_t so = 0, oo = 0;
      const int32_t sl = (int32_t)s.size();
      std::vector<uint8_t> out((size_t)plen + 1);
      uint32_t st = 0;
      int32_t info[24];
      zs2_emu_inflate_pages(s.data(), &so, &sl, &plen, 1, out.data(), &oo, &st, info);
      ran++;
      if (st == 0) accepted++; else if (st == 1) refused++; else corrupt++;
      const int64_t n = plen ? (int64_t)(rnd() % (uint64_t)plen) + 1 : 0;
      std::vector<uint8_t> pre((size_t)n + 1);
      (void)zs2_emu_host_prefix(s.data(), sl, plen, pre.data(), n);
    }
  }
  printf("%ld %ld %ld %ld\n", ran, accepted, refused, corrupt);
  return 0;
}
