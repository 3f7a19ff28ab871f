// Comet shuffle blocks: Arrow IPC stream encode / decode and the block codecs.  See shuffle_format.hpp for the layout and
// the reference citations.  Everything here is host-side framing of buffers the GPU has already partitioned
// (exec.cpp write_shuffle) or is about to consume (ShuffleScan inputs); the reference does the same work on the CPU with the
// arrow-ipc, zstd, lz4_flex and snap crates.
#include "shuffle_format.hpp"

#include "parquet_meta.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <map>

namespace comet {
namespace {

template <class T>
T rd(const uint8_t* p) {
  T v;
  memcpy(&v, p, sizeof v);
  return v;
}
template <class T>
void app(std::vector<uint8_t>& b, T v) {
  const uint8_t* p = (const uint8_t*)&v;
  b.insert(b.end(), p, p + sizeof v);
}

// ---------------------------------------------------------------------------------------------------------------
// xxHash32 (LZ4 frame header checksum) and CRC-32C (Snappy framing), both bit-exact restatements of the published algorithms
// ---------------------------------------------------------------------------------------------------------------
inline uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
const uint32_t XP1 = 2654435761u, XP2 = 2246822519u, XP3 = 3266489917u, XP4 = 668265263u, XP5 = 374761393u;

struct Crc32cTable {
  uint32_t t[256];
  Crc32cTable() {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      t[i] = c;
    }
  }
};

}  // namespace

uint32_t xxh32(const uint8_t* p, size_t n, uint32_t seed) {
  const uint8_t* end = p + n;
  uint32_t h;
  if (n >= 16) {
    uint32_t v1 = seed + XP1 + XP2, v2 = seed + XP2, v3 = seed, v4 = seed - XP1;
    const uint8_t* lim = end - 16;
    do {
      v1 = rotl(v1 + rd<uint32_t>(p) * XP2, 13) * XP1;
      v2 = rotl(v2 + rd<uint32_t>(p + 4) * XP2, 13) * XP1;
      v3 = rotl(v3 + rd<uint32_t>(p + 8) * XP2, 13) * XP1;
      v4 = rotl(v4 + rd<uint32_t>(p + 12) * XP2, 13) * XP1;
      p += 16;
    } while (p <= lim);
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
  } else {
    h = seed + XP5;
  }
  h += (uint32_t)n;
  while (p + 4 <= end) {
    h = rotl(h + rd<uint32_t>(p) * XP3, 17) * XP4;
    p += 4;
  }
  while (p < end) {
    h = rotl(h + (*p) * XP5, 11) * XP1;
    p++;
  }
  h ^= h >> 15;
  h *= XP2;
  h ^= h >> 13;
  h *= XP3;
  h ^= h >> 16;
  return h;
}

__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(const uint8_t* p, size_t n, uint32_t init) {   // the x86 CRC32 instruction IS CRC-32C
  uint64_t c = init ^ 0xFFFFFFFFu;
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    c = __builtin_ia32_crc32di(c, v);
    p += 8;
    n -= 8;
  }
  uint32_t c32 = (uint32_t)c;
  while (n--) c32 = __builtin_ia32_crc32qi(c32, *p++);
  return c32 ^ 0xFFFFFFFFu;
}

uint32_t crc32c(const uint8_t* p, size_t n, uint32_t init) {
  static const bool hw = __builtin_cpu_supports("sse4.2");
  if (hw) return crc32c_hw(p, n, init);
  static const Crc32cTable tab;
  uint32_t c = init ^ 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = tab.t[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------------------------
// Snappy: raw-format compressor (greedy, 4-byte hash matches; format_description.txt) + the framing format
// (framing_format.txt: stream identifier, then chunks of ≤ 65536 uncompressed bytes each carrying a masked CRC-32C)
// ---------------------------------------------------------------------------------------------------------------
namespace {
void snappy_emit_literal(const uint8_t* p, size_t len, std::vector<uint8_t>& out) {
  if (!len) return;
  size_t n = len - 1;
  if (n < 60) out.push_back((uint8_t)(n << 2));
  else if (n < 256) { out.push_back(60 << 2); out.push_back((uint8_t)n); }
  else if (n < 65536) { out.push_back(61 << 2); out.push_back((uint8_t)n); out.push_back((uint8_t)(n >> 8)); }
  else if (n < (1u << 24)) { out.push_back(62 << 2); out.push_back((uint8_t)n); out.push_back((uint8_t)(n >> 8)); out.push_back((uint8_t)(n >> 16)); }
  else { out.push_back(63 << 2); app<uint32_t>(out, (uint32_t)n); }
  out.insert(out.end(), p, p + len);
}
void snappy_emit_copy(size_t offset, size_t len, std::vector<uint8_t>& out) {
  while (len > 0) {
    size_t l = len > 64 ? (len - 64 < 4 ? 60 : 64) : len;   // never leave a tail shorter than 4
    if (l >= 4 && l <= 11 && offset < 2048) {
      out.push_back((uint8_t)(1 | ((l - 4) << 2) | ((offset >> 8) << 5)));
      out.push_back((uint8_t)offset);
    } else if (offset < 65536) {
      out.push_back((uint8_t)(2 | ((l - 1) << 2)));
      out.push_back((uint8_t)offset);
      out.push_back((uint8_t)(offset >> 8));
    } else {
      out.push_back((uint8_t)(3 | ((l - 1) << 2)));
      app<uint32_t>(out, (uint32_t)offset);
    }
    len -= l;
  }
}
}  // namespace

void snappy_compress_raw(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
  size_t v = n;
  while (v >= 0x80) { out.push_back((uint8_t)(v | 0x80)); v >>= 7; }
  out.push_back((uint8_t)v);
  const int kBits = 14;
  static thread_local std::vector<uint32_t> table;
  table.resize((size_t)1 << kBits);
  // 64 KiB windows, each with a fresh table (offsets then always fit 16 bits)
  for (size_t base = 0; base < n; base += 65536) {
    const size_t end = std::min(n, base + 65536);
    std::fill(table.begin(), table.end(), 0xFFFFFFFFu);
    size_t lit = base, i = base;
    uint32_t skip = 32;   // like the reference encoder: probe less often the longer nothing matches
    while (i + 4 <= end) {
      const uint32_t w = rd<uint32_t>(src + i);
      const uint32_t h = (w * 0x1e35a7bdu) >> (32 - kBits);
      const uint32_t cand = table[h];
      table[h] = (uint32_t)(i - base);
      if (cand != 0xFFFFFFFFu && rd<uint32_t>(src + base + cand) == w) {
        const size_t c = base + cand;
        size_t len = 4;
        while (i + len < end && src[c + len] == src[i + len]) len++;
        snappy_emit_literal(src + lit, i - lit, out);
        snappy_emit_copy(i - c, len, out);
        i += len;
        lit = i;
        skip = 32;
      } else {
        i += skip++ >> 5;
      }
    }
    snappy_emit_literal(src + lit, end - lit, out);
  }
}

namespace {
uint32_t snappy_mask(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }
const uint8_t kSnappyStreamId[10] = {0xff, 0x06, 0x00, 0x00, 0x73, 0x4e, 0x61, 0x50, 0x70, 0x59};

void snappy_frame_encode(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
  out.insert(out.end(), kSnappyStreamId, kSnappyStreamId + 10);
  static thread_local std::vector<uint8_t> tmp;
  for (size_t off = 0; off < n; off += 65536) {
    const size_t len = std::min<size_t>(65536, n - off);
    tmp.clear();
    snappy_compress_raw(src + off, len, tmp);
    const bool stored = tmp.size() >= len - len / 8;   // what the snap crate does for incompressible chunks
    const size_t body = 4 + (stored ? len : tmp.size());
    out.push_back(stored ? 0x01 : 0x00);
    out.push_back((uint8_t)body);
    out.push_back((uint8_t)(body >> 8));
    out.push_back((uint8_t)(body >> 16));
    app<uint32_t>(out, snappy_mask(crc32c(src + off, len)));
    if (stored) out.insert(out.end(), src + off, src + off + len);
    else out.insert(out.end(), tmp.begin(), tmp.end());
  }
}

std::vector<uint8_t> snappy_frame_decode(const uint8_t* p, size_t n) {
  std::vector<uint8_t> out;
  size_t i = 0;
  while (i < n) {
    if (i + 4 > n) throw CometError("shuffle block: truncated Snappy chunk header");
    const uint8_t type = p[i];
    const size_t len = (size_t)p[i + 1] | ((size_t)p[i + 2] << 8) | ((size_t)p[i + 3] << 16);
    i += 4;
    if (i + len > n) throw CometError("shuffle block: truncated Snappy chunk");
    if (type == 0xff) {
      if (len != 6 || memcmp(p + i, kSnappyStreamId + 4, 6) != 0) throw CometError("shuffle block: bad Snappy stream identifier");
    } else if (type == 0x00 || type == 0x01) {
      if (len < 4) throw CometError("shuffle block: Snappy chunk without checksum");
      const uint32_t want = rd<uint32_t>(p + i);
      const size_t at = out.size();
      if (type == 0x01) {
        out.insert(out.end(), p + i + 4, p + i + len);
      } else {
        size_t ulen = 0, k = i + 4;
        int shift = 0;
        while (true) {
          if (k >= i + len || shift > 28) throw CometError("shuffle block: bad Snappy preamble");
          const uint8_t b = p[k++];
          ulen |= (size_t)(b & 0x7f) << shift;
          if (!(b & 0x80)) break;
          shift += 7;
        }
        if (ulen > 65536) throw CometError("shuffle block: Snappy chunk larger than 65536 bytes");
        out.resize(at + ulen);
        pq::decompress(pq::SNAPPY, p + i + 4, len - 4, out.data() + at, ulen);
      }
      if (snappy_mask(crc32c(out.data() + at, out.size() - at)) != want) throw CometError("shuffle block: Snappy chunk checksum mismatch");
    } else if (type >= 0x02 && type <= 0x7f) {
      throw CometError("shuffle block: reserved unskippable Snappy chunk type " + std::to_string(type));
    }   // 0x80..0xfe: skippable / padding
    i += len;
  }
  return out;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// LZ4: block format (lz4_Block_format.md) compressor / decompressor and the frame format (lz4_Frame_format.md)
// ---------------------------------------------------------------------------------------------------------------
void lz4_compress_block(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
  int kBits = 12;   // table sized to the input: a 64 Ki-entry table costs more to clear than a small block to compress
  while (kBits < 16 && ((size_t)1 << kBits) < n / 4) kBits++;
  static thread_local std::vector<uint32_t> table;
  table.assign((size_t)1 << kBits, 0xFFFFFFFFu);
  auto emit = [&](const uint8_t* lit, size_t lit_len, size_t offset, size_t match_len) {   // match_len 0: last literals
    const size_t ml = match_len ? match_len - 4 : 0;
    out.push_back((uint8_t)((std::min<size_t>(lit_len, 15) << 4) | std::min<size_t>(ml, 15)));
    if (lit_len >= 15) {
      size_t r = lit_len - 15;
      while (r >= 255) { out.push_back(255); r -= 255; }
      out.push_back((uint8_t)r);
    }
    out.insert(out.end(), lit, lit + lit_len);
    if (match_len) {
      out.push_back((uint8_t)offset);
      out.push_back((uint8_t)(offset >> 8));
      if (ml >= 15) {
        size_t r = ml - 15;
        while (r >= 255) { out.push_back(255); r -= 255; }
        out.push_back((uint8_t)r);
      }
    }
  };
  size_t lit = 0, i = 0;
  if (n >= 13) {
    const size_t match_limit = n - 12;   // a match may not start within the last 12 bytes …
    const size_t end_limit = n - 5;      // … nor cover the last 5
    uint32_t skip = 64;   // LZ4's acceleration: step grows by one every 64 misses
    while (i < match_limit) {
      const uint32_t w = rd<uint32_t>(src + i);
      const uint32_t h = (w * 2654435761u) >> (32 - kBits);
      const uint32_t cand = table[h];
      table[h] = (uint32_t)i;
      if (cand != 0xFFFFFFFFu && i - cand <= 65535 && rd<uint32_t>(src + cand) == w) {
        size_t len = 4;
        while (i + len + 8 <= end_limit && rd<uint64_t>(src + cand + len) == rd<uint64_t>(src + i + len)) len += 8;
        while (i + len < end_limit && src[cand + len] == src[i + len]) len++;
        emit(src + lit, i - lit, i - cand, len);
        i += len;
        lit = i;
        skip = 64;
      } else {
        i += skip++ >> 6;
      }
    }
  }
  emit(src + lit, n - lit, 0, 0);
}

// decodes one block into dst[dst_pos..); matches may reach back into dst[0..dst_pos) (linked blocks).  Returns the new position.
size_t lz4_decompress_block(const uint8_t* src, size_t n, uint8_t* dst, size_t dst_cap, size_t dst_pos) {
  size_t i = 0, o = dst_pos;
  while (i < n) {
    const uint8_t token = src[i++];
    size_t lit = token >> 4;
    if (lit == 15) {
      uint8_t b;
      do {
        if (i >= n) throw CometError("lz4: truncated literal length");
        b = src[i++];
        lit += b;
      } while (b == 255);
    }
    if (i + lit > n || o + lit > dst_cap) throw CometError("lz4: literal overruns buffer");
    memcpy(dst + o, src + i, lit);
    i += lit;
    o += lit;
    if (i >= n) break;   // last sequence: literals only
    if (i + 2 > n) throw CometError("lz4: truncated offset");
    const size_t off = (size_t)src[i] | ((size_t)src[i + 1] << 8);
    i += 2;
    size_t ml = token & 15;
    if (ml == 15) {
      uint8_t b;
      do {
        if (i >= n) throw CometError("lz4: truncated match length");
        b = src[i++];
        ml += b;
      } while (b == 255);
    }
    ml += 4;
    if (off == 0 || off > o || o + ml > dst_cap) throw CometError("lz4: bad match");
    if (off >= ml) memcpy(dst + o, dst + o - off, ml);
    else for (size_t k = 0; k < ml; k++) dst[o + k] = dst[o + k - off];   // overlapping run
    o += ml;
  }
  return o;
}

namespace {
const size_t kLz4BlockMax = 4u << 20;

void lz4_frame_encode(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
  app<uint32_t>(out, 0x184D2204u);
  const uint8_t desc[2] = {0x60 /* version 01, independent blocks */, 0x70 /* 4 MiB blocks */};
  out.push_back(desc[0]);
  out.push_back(desc[1]);
  out.push_back((uint8_t)(xxh32(desc, 2, 0) >> 8));
  static thread_local std::vector<uint8_t> tmp;
  for (size_t off = 0; off < n; off += kLz4BlockMax) {
    const size_t len = std::min(kLz4BlockMax, n - off);
    tmp.clear();
    lz4_compress_block(src + off, len, tmp);
    if (tmp.size() >= len) {
      app<uint32_t>(out, (uint32_t)len | 0x80000000u);
      out.insert(out.end(), src + off, src + off + len);
    } else {
      app<uint32_t>(out, (uint32_t)tmp.size());
      out.insert(out.end(), tmp.begin(), tmp.end());
    }
  }
  app<uint32_t>(out, 0);   // end mark
}

std::vector<uint8_t> lz4_frame_decode(const uint8_t* p, size_t n) {
  if (n < 7 || rd<uint32_t>(p) != 0x184D2204u) throw CometError("shuffle block: not an LZ4 frame");
  const uint8_t flg = p[4], bd = p[5];
  if ((flg >> 6) != 1) throw CometError("shuffle block: unsupported LZ4 frame version");
  const bool block_checksum = flg & 0x10, has_size = flg & 0x08, content_checksum = flg & 0x04, has_dict = flg & 0x01;
  size_t i = 6 + (has_size ? 8 : 0) + (has_dict ? 4 : 0);
  if (i + 1 > n) throw CometError("shuffle block: truncated LZ4 frame header");
  if ((uint8_t)(xxh32(p + 4, i - 4, 0) >> 8) != p[i]) throw CometError("shuffle block: LZ4 frame header checksum mismatch");
  i++;
  const int bs = (bd >> 4) & 7;
  if (bs < 4) throw CometError("shuffle block: bad LZ4 block size code");
  const size_t block_max = (size_t)1 << (8 + 2 * bs);
  std::vector<uint8_t> out;
  if (has_size) out.reserve((size_t)rd<uint64_t>(p + 6));
  size_t pos = 0;
  while (true) {
    if (i + 4 > n) throw CometError("shuffle block: truncated LZ4 frame");
    const uint32_t w = rd<uint32_t>(p + i);
    i += 4;
    if (w == 0) break;
    const size_t len = w & 0x7FFFFFFFu;
    if (i + len > n) throw CometError("shuffle block: truncated LZ4 block");
    if (w & 0x80000000u) {
      out.resize(pos + len);
      memcpy(out.data() + pos, p + i, len);
      pos += len;
    } else {
      out.resize(pos + block_max);
      pos = lz4_decompress_block(p + i, len, out.data(), out.size(), pos);
      out.resize(pos);
    }
    i += len + (block_checksum ? 4 : 0);
  }
  if (content_checksum) {
    if (i + 4 > n) throw CometError("shuffle block: truncated LZ4 content checksum");
    if (rd<uint32_t>(p + i) != xxh32(out.data(), out.size(), 0)) throw CometError("shuffle block: LZ4 content checksum mismatch");
  }
  return out;
}

// ---------------------------------------------------------------------------------------------------------------
// zstd through libzstd.so.1 (no headers in the image: the few prototypes used are restated here)
// ---------------------------------------------------------------------------------------------------------------
struct ZInBuf { const void* src; size_t size; size_t pos; };
struct ZOutBuf { void* dst; size_t size; size_t pos; };
struct Zstd {
  size_t (*compress)(void*, size_t, const void*, size_t, int) = nullptr;
  size_t (*bound)(size_t) = nullptr;
  unsigned (*is_error)(size_t) = nullptr;
  void* (*create_dstream)() = nullptr;
  size_t (*free_dstream)(void*) = nullptr;
  size_t (*decompress_stream)(void*, ZOutBuf*, ZInBuf*) = nullptr;
  Zstd() {
    void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    compress = (decltype(compress))dlsym(h, "ZSTD_compress");
    bound = (decltype(bound))dlsym(h, "ZSTD_compressBound");
    is_error = (decltype(is_error))dlsym(h, "ZSTD_isError");
    create_dstream = (decltype(create_dstream))dlsym(h, "ZSTD_createDStream");
    free_dstream = (decltype(free_dstream))dlsym(h, "ZSTD_freeDStream");
    decompress_stream = (decltype(decompress_stream))dlsym(h, "ZSTD_decompressStream");
  }
  bool ok() const { return compress && bound && is_error && create_dstream && free_dstream && decompress_stream; }
};
const Zstd& zstd() {
  static const Zstd z;
  if (!z.ok()) throw CometError("shuffle block: the ZSTD codec needs libzstd.so.1, which could not be loaded");
  return z;
}

void zstd_encode(const uint8_t* src, size_t n, int level, std::vector<uint8_t>& out) {
  const Zstd& z = zstd();
  const size_t at = out.size(), cap = z.bound(n);
  out.resize(at + cap);
  const size_t rc = z.compress(out.data() + at, cap, src, n, level);
  if (z.is_error(rc)) throw CometError("shuffle block: zstd compression failed");
  out.resize(at + rc);
}

std::vector<uint8_t> zstd_decode(const uint8_t* p, size_t n) {   // streaming: the reference's encoder does not record the content size
  const Zstd& z = zstd();
  void* ds = z.create_dstream();
  if (!ds) throw CometError("shuffle block: ZSTD_createDStream failed");
  std::vector<uint8_t> out(std::max<size_t>(n * 4, 1 << 16));
  ZInBuf in{p, n, 0};
  ZOutBuf ob{out.data(), out.size(), 0};
  while (in.pos < in.size) {
    if (ob.pos == ob.size) {
      out.resize(out.size() * 2);
      ob.dst = out.data();
      ob.size = out.size();
    }
    const size_t rc = z.decompress_stream(ds, &ob, &in);
    if (z.is_error(rc)) {
      z.free_dstream(ds);
      throw CometError("shuffle block: zstd decompression failed");
    }
    if (rc == 0 && in.pos == in.size) break;
  }
  // drain what the decoder still buffers
  while (true) {
    if (ob.pos == ob.size) {
      out.resize(out.size() * 2);
      ob.dst = out.data();
      ob.size = out.size();
    }
    const size_t before = ob.pos;
    const size_t rc = z.decompress_stream(ds, &ob, &in);
    if (z.is_error(rc)) {
      z.free_dstream(ds);
      throw CometError("shuffle block: zstd decompression failed");
    }
    if (ob.pos == before) break;
  }
  z.free_dstream(ds);
  out.resize(ob.pos);
  return out;
}

// ---------------------------------------------------------------------------------------------------------------
// flatbuffers, written front to back: parents first with placeholder offsets, children after them (uoffsets point forward)
// ---------------------------------------------------------------------------------------------------------------
struct FbField {
  int slot;
  int size;        // 1, 2, 4, 8
  uint64_t value;  // scalar bits (ignored for refs)
  bool ref;
};

struct FbWriter {
  std::vector<uint8_t> b;
  template <class T> void put(T v) { app<T>(b, v); }
  template <class T> void set(size_t at, T v) { memcpy(&b[at], &v, sizeof v); }
  void pad_to(size_t a) { while (b.size() % a) b.push_back(0); }
  void patch(size_t at, size_t target) { set<uint32_t>(at, (uint32_t)(target - at)); }

  // vtable + table.  ref_pos[slot] = position of that slot's offset placeholder.
  size_t table(int nslots, std::vector<FbField> fs, std::vector<size_t>& ref_pos) {
    std::stable_sort(fs.begin(), fs.end(), [](const FbField& a, const FbField& c) { return a.size > c.size; });
    bool has8 = false;
    for (auto& f : fs) has8 |= f.size == 8;
    const size_t vts = 4 + 2 * (size_t)nslots;
    size_t start = b.size() + vts;
    while (start % 4 || (has8 && (start + 4) % 8)) start++;
    b.resize(start - vts, 0);
    std::vector<uint16_t> offs((size_t)nslots, 0);
    size_t off = 4;
    for (auto& f : fs) {
      while ((start + off) % (size_t)f.size) off++;
      offs[(size_t)f.slot] = (uint16_t)off;
      off += (size_t)f.size;
    }
    put<uint16_t>((uint16_t)vts);
    put<uint16_t>((uint16_t)off);
    for (uint16_t o : offs) put<uint16_t>(o);
    put<int32_t>((int32_t)vts);   // soffset: vtable = table − soffset
    ref_pos.assign((size_t)nslots, 0);
    for (auto& f : fs) {
      b.resize(start + offs[(size_t)f.slot], 0);
      if (f.ref) {
        ref_pos[(size_t)f.slot] = b.size();
        put<uint32_t>(0);
      } else if (f.size == 8) put<uint64_t>(f.value);
      else if (f.size == 4) put<uint32_t>((uint32_t)f.value);
      else if (f.size == 2) put<uint16_t>((uint16_t)f.value);
      else put<uint8_t>((uint8_t)f.value);
    }
    b.resize(start + off, 0);
    return start;
  }
  size_t string(const std::string& s) {
    pad_to(4);
    const size_t p = b.size();
    put<uint32_t>((uint32_t)s.size());
    b.insert(b.end(), s.begin(), s.end());
    b.push_back(0);
    return p;
  }
  size_t offset_vector(size_t n, std::vector<size_t>& elem_pos) {
    pad_to(4);
    const size_t p = b.size();
    put<uint32_t>((uint32_t)n);
    elem_pos.clear();
    for (size_t i = 0; i < n; i++) {
      elem_pos.push_back(b.size());
      put<uint32_t>(0);
    }
    return p;
  }
  size_t struct16_vector(const std::vector<std::pair<int64_t, int64_t>>& v) {   // FieldNode / Buffer: two longs, 8-aligned
    while ((b.size() + 4) % 8) b.push_back(0);
    const size_t p = b.size();
    put<uint32_t>((uint32_t)v.size());
    for (auto& e : v) {
      put<int64_t>(e.first);
      put<int64_t>(e.second);
    }
    return p;
  }
};

FbField scalar(int slot, int size, uint64_t v) { return FbField{slot, size, v, false}; }
FbField ref(int slot) { return FbField{slot, 4, 0, true}; }

// flatbuffer Type union ids (format/Schema.fbs)
enum { FB_Int = 2, FB_FloatingPoint = 3, FB_Binary = 4, FB_Utf8 = 5, FB_Bool = 6, FB_Decimal = 7, FB_Date = 8, FB_Timestamp = 10, FB_List = 12, FB_Struct = 13, FB_Map = 17 };
enum { MSG_Schema = 1, MSG_DictionaryBatch = 2, MSG_RecordBatch = 3 };

uint8_t fb_type_id(const DType& t) {
  switch (t.id) {
    case TypeId::Bool: return FB_Bool;
    case TypeId::Float: case TypeId::Double: return FB_FloatingPoint;
    case TypeId::String: return FB_Utf8;
    case TypeId::Bytes: return FB_Binary;
    case TypeId::Decimal: return FB_Decimal;
    case TypeId::Date: return FB_Date;
    case TypeId::Timestamp: case TypeId::TimestampNtz: return FB_Timestamp;
    case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: case TypeId::Int64: return FB_Int;
    case TypeId::List: return FB_List;
    case TypeId::Struct: return FB_Struct;
    case TypeId::Map: return FB_Map;
    default: throw CometError("shuffle writer: column type " + t.str() + " is not supported");
  }
}

void write_type(FbWriter& w, const DType& t, size_t type_ref) {
  std::vector<size_t> rp;
  size_t tab;
  switch (t.id) {
    case TypeId::Bool: tab = w.table(0, {}, rp); break;
    case TypeId::Int8: case TypeId::Int16: case TypeId::Int32: case TypeId::Int64: {
      const int bits = t.id == TypeId::Int8 ? 8 : t.id == TypeId::Int16 ? 16 : t.id == TypeId::Int32 ? 32 : 64;
      tab = w.table(2, {scalar(0, 4, (uint64_t)bits), scalar(1, 1, 1)}, rp);
      break;
    }
    case TypeId::Float: tab = w.table(1, {scalar(0, 2, 1)}, rp); break;
    case TypeId::Double: tab = w.table(1, {scalar(0, 2, 2)}, rp); break;
    case TypeId::String: case TypeId::Bytes: case TypeId::List: case TypeId::Struct: tab = w.table(0, {}, rp); break;
    case TypeId::Map: tab = w.table(1, {scalar(0, 1, 0 /* keysSorted = false */)}, rp); break;
    case TypeId::Decimal: tab = w.table(3, {scalar(0, 4, (uint64_t)t.precision), scalar(1, 4, (uint64_t)t.scale), scalar(2, 4, 128)}, rp); break;
    case TypeId::Date: tab = w.table(1, {scalar(0, 2, 0 /* DAY */)}, rp); break;
    case TypeId::Timestamp: {
      tab = w.table(2, {scalar(0, 2, 2 /* MICROSECOND */), ref(1)}, rp);
      const size_t tz = w.string("UTC");
      w.patch(rp[1], tz);
      break;
    }
    case TypeId::TimestampNtz: tab = w.table(2, {scalar(0, 2, 2)}, rp); break;
    default: throw CometError("shuffle writer: column type " + t.str() + " is not supported");
  }
  w.patch(type_ref, tab);
}

// one encapsulated IPC message: continuation marker, metadata length (8-padded), flatbuffer
void append_message(std::vector<uint8_t>& out, FbWriter& w) {
  while (w.b.size() % 8) w.b.push_back(0);
  app<uint32_t>(out, 0xFFFFFFFFu);
  app<int32_t>(out, (int32_t)w.b.size());
  out.insert(out.end(), w.b.begin(), w.b.end());
}

void write_schema_message(const std::vector<ColumnSlice>& cols, std::vector<uint8_t>& out) {
  FbWriter w;
  w.put<uint32_t>(0);
  std::vector<size_t> mrp, srp, elem, frp;
  const size_t msg = w.table(5, {scalar(0, 2, 4 /* MetadataVersion V5 */), scalar(1, 1, MSG_Schema), ref(2), scalar(3, 8, 0)}, mrp);
  w.patch(0, msg);
  const size_t schema = w.table(4, {scalar(0, 2, 0 /* little endian */), ref(1)}, srp);
  w.patch(mrp[2], schema);
  const size_t fields = w.offset_vector(cols.size(), elem);
  w.patch(srp[1], fields);
  // one Field table (Schema.fbs: name, nullable, type_type, type, dictionary, children), its children behind it, depth first
  std::function<void(const DType&, const std::string&, bool, size_t)> write_field = [&](const DType& t, const std::string& fname, bool nullable, size_t slot) {
    std::vector<size_t> rp;
    const size_t f = w.table(7, {ref(0), scalar(1, 1, nullable ? 1 : 0), scalar(2, 1, fb_type_id(t)), ref(3), ref(5)}, rp);
    w.patch(slot, f);
    const size_t name = w.string(fname);
    w.patch(rp[0], name);
    write_type(w, t, rp[3]);
    const size_t nk = t.is_nested() ? t.kids.size() : 0;
    std::vector<size_t> kid_slots;
    const size_t children = w.offset_vector(nk, kid_slots);
    w.patch(rp[5], children);
    for (size_t k = 0; k < nk; k++)
      write_field(t.kids[k], t.id == TypeId::List ? std::string("item") : t.id == TypeId::Map ? std::string("entries") : (k < t.kid_names.size() ? t.kid_names[k] : std::string()),
                  t.id == TypeId::Map ? false : (k < t.kid_nullable.size() ? t.kid_nullable[k] != 0 : true), kid_slots[k]);
  };
  for (size_t c = 0; c < cols.size(); c++) write_field(cols[c].type, "c" + std::to_string(c), true, elem[c]);
  append_message(out, w);
}

// bits [first, first+n) of src → a fresh bitmap; returns the number of SET bits
int64_t slice_bits(const uint8_t* src, int64_t first, int64_t n, std::vector<uint8_t>& dst) {
  dst.assign((size_t)((n + 7) / 8), 0);
  if (n == 0) return 0;
  const int sh = (int)(first & 7);
  const uint8_t* s = src + (first >> 3);
  const size_t nb = dst.size();
  if (sh == 0) {
    memcpy(dst.data(), s, nb);
  } else {
    const size_t src_bytes = (size_t)((sh + n + 7) / 8);
    for (size_t i = 0; i < nb; i++) {
      const uint32_t lo = s[i], hi = i + 1 < src_bytes ? s[i + 1] : 0;
      dst[i] = (uint8_t)((lo >> sh) | (hi << (8 - sh)));
    }
  }
  if (n & 7) dst[nb - 1] &= (uint8_t)((1u << (n & 7)) - 1);
  int64_t set = 0;
  size_t i = 0;
  for (; i + 8 <= nb; i += 8) set += __builtin_popcountll(rd<uint64_t>(dst.data() + i));
  for (; i < nb; i++) set += __builtin_popcount(dst[i]);
  return set;
}

int ipc_fixed_width(const DType& t) {
  switch (t.id) {
    case TypeId::Int8: return 1;
    case TypeId::Int16: return 2;
    case TypeId::Int32: case TypeId::Float: case TypeId::Date: return 4;
    case TypeId::Int64: case TypeId::Double: case TypeId::Timestamp: case TypeId::TimestampNtz: return 8;
    case TypeId::Decimal: return 16;
    default: return 0;
  }
}

void write_batch_message(const std::vector<ColumnSlice>& cols, int64_t rows, std::vector<uint8_t>& out) {
  static thread_local std::vector<uint8_t> body;   // reused: a fresh few-hundred-KB vector per block is an mmap + page faults
  body.clear();
  std::vector<std::pair<int64_t, int64_t>> nodes, buffers;
  std::vector<uint8_t> bits;
  auto add_buffer = [&](const void* p, size_t n) {
    buffers.emplace_back((int64_t)body.size(), (int64_t)n);
    if (n) body.insert(body.end(), (const uint8_t*)p, (const uint8_t*)p + n);
    while (body.size() % 8) body.push_back(0);
  };
  // field nodes and buffers in depth-first pre-order (Message.fbs RecordBatch): a struct is its validity, then its fields over the same rows;
  // a list its validity and its offsets (rebased to 0), then its elements offsets[first] … offsets[first + rows)
  std::function<void(const ColumnSlice&, int64_t, int64_t)> emit = [&](const ColumnSlice& c, int64_t first, int64_t rows) {
    int64_t nulls = 0;
    if (c.validity) {
      nulls = rows - slice_bits(c.validity, first, rows, bits);
      if (nulls) add_buffer(bits.data(), bits.size());
    }
    if (!nulls) add_buffer(nullptr, 0);
    nodes.emplace_back(rows, nulls);
    if (c.type.id == TypeId::Struct) {
      if (c.kids.size() != c.type.kids.size()) throw CometError("shuffle writer: struct column without its fields");
      for (const ColumnSlice& k : c.kids) emit(k, first, rows);
      return;
    }
    if (c.type.is_listlike()) {
      if (c.kids.size() != 1) throw CometError("shuffle writer: list column without its elements");
      const int32_t* offs = (const int32_t*)c.values + first;
      const int32_t base = offs[0];
      buffers.emplace_back((int64_t)body.size(), (int64_t)(rows + 1) * 4);
      const size_t at = body.size();
      body.resize(at + (size_t)(rows + 1) * 4);
      int32_t* o = (int32_t*)(body.data() + at);
      for (int64_t i = 0; i <= rows; i++) o[i] = offs[i] - base;
      while (body.size() % 8) body.push_back(0);
      emit(c.kids[0], (int64_t)base, (int64_t)(offs[rows] - base));
      return;
    }
    if (c.type.id == TypeId::String || c.type.id == TypeId::Bytes) {
      const int32_t* offs = (const int32_t*)c.values + first;
      const int32_t base = offs[0];
      buffers.emplace_back((int64_t)body.size(), (int64_t)(rows + 1) * 4);
      const size_t at = body.size();
      body.resize(at + (size_t)(rows + 1) * 4);
      int32_t* o = (int32_t*)(body.data() + at);
      for (int64_t i = 0; i <= rows; i++) o[i] = offs[i] - base;
      while (body.size() % 8) body.push_back(0);
      add_buffer(c.data + ((int64_t)base - c.data_origin), (size_t)(offs[rows] - base));
    } else if (c.type.id == TypeId::Bool) {
      slice_bits((const uint8_t*)c.values, first, rows, bits);
      add_buffer(bits.data(), bits.size());
    } else {
      const int w = ipc_fixed_width(c.type);
      if (!w) throw CometError("shuffle writer: column type " + c.type.str() + " is not supported");
      add_buffer((const uint8_t*)c.values + (size_t)first * (size_t)w, (size_t)rows * (size_t)w);
    }
  };
  for (const ColumnSlice& c : cols) emit(c, c.first, rows);
  FbWriter w;
  w.put<uint32_t>(0);
  std::vector<size_t> mrp, rrp;
  const size_t msg = w.table(5, {scalar(0, 2, 4), scalar(1, 1, MSG_RecordBatch), ref(2), scalar(3, 8, (uint64_t)body.size())}, mrp);
  w.patch(0, msg);
  const size_t rb = w.table(5, {scalar(0, 8, (uint64_t)rows), ref(1), ref(2)}, rrp);
  w.patch(mrp[2], rb);
  const size_t nv = w.struct16_vector(nodes);
  w.patch(rrp[1], nv);
  const size_t bv = w.struct16_vector(buffers);
  w.patch(rrp[2], bv);
  append_message(out, w);
  out.insert(out.end(), body.begin(), body.end());
}

// ---------------------------------------------------------------------------------------------------------------
// flatbuffer reader (bounds-checked accessors over untrusted bytes)
// ---------------------------------------------------------------------------------------------------------------
struct FbTable {
  const uint8_t* buf = nullptr;
  size_t len = 0, pos = 0;
  bool valid() const { return buf != nullptr; }
  const uint8_t* field(int slot, size_t size) const {
    if (pos + 4 > len) throw CometError("shuffle block: corrupt IPC metadata");
    const int64_t vt = (int64_t)pos - rd<int32_t>(buf + pos);
    if (vt < 0 || (size_t)vt + 4 > len) throw CometError("shuffle block: corrupt IPC metadata");
    const uint16_t vts = rd<uint16_t>(buf + vt);
    if ((size_t)vt + vts > len) throw CometError("shuffle block: corrupt IPC metadata");
    if (4 + 2 * (size_t)slot + 2 > vts) return nullptr;
    const uint16_t off = rd<uint16_t>(buf + vt + 4 + 2 * slot);
    if (!off) return nullptr;
    if (pos + off + size > len) throw CometError("shuffle block: corrupt IPC metadata");
    return buf + pos + off;
  }
  template <class T> T get(int slot, T def) const {
    const uint8_t* p = field(slot, sizeof(T));
    return p ? rd<T>(p) : def;
  }
  size_t target(int slot) const {   // position a uoffset field points at, 0 if absent
    const uint8_t* p = field(slot, 4);
    if (!p) return 0;
    const size_t t = (size_t)(p - buf) + rd<uint32_t>(p);
    if (t + 4 > len) throw CometError("shuffle block: corrupt IPC metadata");
    return t;
  }
  FbTable child(int slot) const {
    const size_t t = target(slot);
    FbTable c;
    if (t) { c.buf = buf; c.len = len; c.pos = t; }
    return c;
  }
  std::string str(int slot) const {
    const size_t t = target(slot);
    if (!t) return "";
    const uint32_t n = rd<uint32_t>(buf + t);
    if (t + 4 + n > len) throw CometError("shuffle block: corrupt IPC metadata");
    return std::string((const char*)buf + t + 4, n);
  }
  // vector: returns element count and the position of element 0
  size_t vec(int slot, size_t elem_size, size_t& first) const {
    const size_t t = target(slot);
    first = 0;
    if (!t) return 0;
    const uint32_t n = rd<uint32_t>(buf + t);
    if (t + 4 + (size_t)n * elem_size > len) throw CometError("shuffle block: corrupt IPC metadata");
    first = t + 4;
    return n;
  }
  FbTable elem_table(size_t first, size_t i) const {
    const size_t at = first + 4 * i;
    FbTable c;
    c.buf = buf;
    c.len = len;
    c.pos = at + rd<uint32_t>(buf + at);
    if (c.pos + 4 > len) throw CometError("shuffle block: corrupt IPC metadata");
    return c;
  }
};

struct IpcField {
  DType type;
  bool dict = false;
  int64_t dict_id = 0;
  int index_width = 4;
};

DType type_from_fb(int type_id, const FbTable& t) {
  switch (type_id) {
    case FB_Bool: return DType::of(TypeId::Bool);
    case FB_Int: {
      const int bits = t.valid() ? t.get<int32_t>(0, 0) : 0;
      const bool sign = t.valid() ? t.get<uint8_t>(1, 0) != 0 : false;
      if (!sign) throw CometError("shuffle block: unsigned integer columns are not supported");
      switch (bits) {
        case 8: return DType::of(TypeId::Int8);
        case 16: return DType::of(TypeId::Int16);
        case 32: return DType::of(TypeId::Int32);
        case 64: return DType::of(TypeId::Int64);
      }
      throw CometError("shuffle block: bad integer width");
    }
    case FB_FloatingPoint: {
      const int p = t.valid() ? t.get<int16_t>(0, 0) : 0;
      if (p == 1) return DType::of(TypeId::Float);
      if (p == 2) return DType::of(TypeId::Double);
      throw CometError("shuffle block: half floats are not supported");
    }
    case FB_Utf8: return DType::of(TypeId::String);
    case FB_Binary: return DType::of(TypeId::Bytes);
    case FB_Decimal: {
      if (t.get<int32_t>(2, 128) != 128) throw CometError("shuffle block: only 128-bit decimals are supported");
      return DType::decimal(t.get<int32_t>(0, 0), t.get<int32_t>(1, 0));
    }
    case FB_Date:
      if ((t.valid() ? t.get<int16_t>(0, 1) : 1) != 0) throw CometError("shuffle block: only Date32 is supported");
      return DType::of(TypeId::Date);
    case FB_Timestamp: {
      if ((t.valid() ? t.get<int16_t>(0, 0) : 0) != 2) throw CometError("shuffle block: only microsecond timestamps are supported");
      return DType::of(t.str(1).empty() ? TypeId::TimestampNtz : TypeId::Timestamp);
    }
  }
  throw CometError("shuffle block: Arrow type id " + std::to_string(type_id) + " is not supported");
}
// a Field table → its type, children included (List: one child; Struct: its fields by name)
DType type_of_field(const FbTable& f, int depth = 0) {
  const int tid = f.get<uint8_t>(2, 0);
  if (tid != FB_List && tid != FB_Struct && tid != FB_Map) return type_from_fb(tid, f.child(3));
  if (depth > 8) throw CometError("shuffle block: types nested deeper than 8 levels");
  DType t = DType::of(tid == FB_List ? TypeId::List : tid == FB_Map ? TypeId::Map : TypeId::Struct);
  size_t first;
  const size_t nk = f.vec(5, 4, first);
  if (tid != FB_Struct && nk != 1) throw CometError("shuffle block: a list / map field with " + std::to_string(nk) + " children");
  for (size_t k = 0; k < nk; k++) {
    FbTable kf = f.elem_table(first, k);
    if (kf.child(4).valid()) throw CometError("shuffle block: dictionary-encoded nested fields are not supported");
    t.kids.push_back(type_of_field(kf, depth + 1));
    t.kid_names.push_back(tid == FB_List ? std::string("element") : tid == FB_Map ? std::string("entries") : kf.str(0));
    t.kid_nullable.push_back(kf.get<uint8_t>(1, 0) != 0 ? 1 : 0);
  }
  if (tid == FB_Map) {
    if (t.kids[0].id != TypeId::Struct || t.kids[0].kids.size() != 2) throw CometError("shuffle block: a map whose entries are not (key, value) structs");
    t.kids[0].kid_names = {"key", "value"};
  }
  return t;
}

struct BodyCursor {
  const uint8_t* body;
  size_t body_len;
  const FbTable* rb;
  size_t nodes_first = 0, n_nodes = 0, bufs_first = 0, n_bufs = 0, node = 0, buf = 0;
  std::pair<int64_t, int64_t> next_node() {
    if (node >= n_nodes) throw CometError("shuffle block: record batch has too few field nodes");
    const uint8_t* p = rb->buf + nodes_first + 16 * node++;
    return {rd<int64_t>(p), rd<int64_t>(p + 8)};
  }
  std::pair<const uint8_t*, size_t> next_buffer() {
    if (buf >= n_bufs) throw CometError("shuffle block: record batch has too few buffers");
    const uint8_t* p = rb->buf + bufs_first + 16 * buf++;
    const int64_t off = rd<int64_t>(p), n = rd<int64_t>(p + 8);
    if (off < 0 || n < 0 || (size_t)off + (size_t)n > body_len) throw CometError("shuffle block: buffer outside the message body");
    return {body + off, (size_t)n};
  }
};

HostColumn read_plain_column(const DType& type, BodyCursor& cur, int64_t rows_expected) {
  HostColumn c;
  c.type = type;
  auto node = cur.next_node();
  c.length = node.first;
  c.null_count = node.second;
  if (rows_expected >= 0 && c.length != rows_expected) throw CometError("shuffle block: column length differs from the batch length");
  // what arrow's own reader refuses as well: a block from disk is not trusted to describe itself consistently
  if (c.length < 0 || c.length > 0x7fffffff) throw CometError("shuffle block: column length " + std::to_string(c.length) + " out of range");
  if (c.null_count < 0 || c.null_count > c.length) throw CometError("shuffle block: null count " + std::to_string(c.null_count) + " of a column of " + std::to_string(c.length) + " rows");
  if (type.id == TypeId::Decimal && (type.precision < 1 || type.precision > 38 || type.scale < 0 || type.scale > type.precision))
    throw CometError("shuffle block: decimal(" + std::to_string(type.precision) + "," + std::to_string(type.scale) + ") is not a Spark decimal type");
  auto vb = cur.next_buffer();
  const size_t bm = (size_t)((c.length + 7) / 8);
  if (c.null_count > 0) {
    if (vb.second < bm) throw CometError("shuffle block: validity buffer too short");
    c.validity.assign(vb.first, vb.first + bm);
  }
  if (type.id == TypeId::Struct) {
    for (const DType& kt : type.kids) c.children.push_back(read_plain_column(kt, cur, c.length));
    return c;
  }
  if (type.is_listlike()) {
    auto ob = cur.next_buffer();
    c.values.assign((size_t)(c.length + 1) * 4, 0);
    int32_t base = 0, last = 0;
    if (c.length > 0) {
      if (ob.second < (size_t)(c.length + 1) * 4) throw CometError("shuffle block: offsets buffer too short");
      base = rd<int32_t>(ob.first);
      int32_t* o = (int32_t*)c.values.data();
      int32_t prev = 0;
      for (int64_t i = 0; i <= c.length; i++) {
        o[i] = rd<int32_t>(ob.first + 4 * i) - base;
        if (o[i] < prev) throw CometError("shuffle block: list offsets are not monotonic");
        prev = o[i];
      }
      last = o[c.length];
    }
    if (type.kids.size() != 1) throw CometError("shuffle block: list type without an element type");
    HostColumn el = read_plain_column(type.kids[0], cur, -1);
    if (base < 0 || (int64_t)base + last > el.length) throw CometError("shuffle block: list offsets run past the element column");
    if (base != 0 || el.length != last) {      // a sliced writer: keep exactly the elements the offsets address
      HostColumn cut;
      cut.type = el.type;
      cut.length = last;
      const int w = ipc_fixed_width(el.type);
      if (!w || !el.children.empty()) throw CometError("shuffle block: a list whose offsets do not start at 0 is supported for fixed-width elements only");
      cut.values.assign(el.values.begin() + (size_t)base * (size_t)w, el.values.begin() + (size_t)(base + last) * (size_t)w);
      if (!el.validity.empty()) {
        cut.validity.assign((size_t)((last + 7) / 8), 0);
        for (int32_t i = 0; i < last; i++) {
          const int64_t sidx = (int64_t)base + i;
          if ((el.validity[(size_t)(sidx >> 3)] >> (sidx & 7)) & 1) cut.validity[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
          else cut.null_count++;
        }
        if (!cut.null_count) cut.validity.clear();
      }
      el = std::move(cut);
    }
    c.children.push_back(std::move(el));
    return c;
  }
  if (type.id == TypeId::String || type.id == TypeId::Bytes) {
    auto ob = cur.next_buffer();
    auto db = cur.next_buffer();
    c.values.assign((size_t)(c.length + 1) * 4, 0);
    if (c.length > 0) {
      if (ob.second < (size_t)(c.length + 1) * 4) throw CometError("shuffle block: offsets buffer too short");
      const int32_t base = rd<int32_t>(ob.first);
      int32_t* o = (int32_t*)c.values.data();
      int32_t prev = 0;
      for (int64_t i = 0; i <= c.length; i++) {
        o[i] = rd<int32_t>(ob.first + 4 * i) - base;
        if (o[i] < prev) throw CometError("shuffle block: string offsets are not monotonic");
        prev = o[i];
      }
      if ((size_t)base + (size_t)o[c.length] > db.second) throw CometError("shuffle block: string data buffer too short");
      c.data.assign(db.first + base, db.first + base + o[c.length]);
    }
  } else {
    auto vals = cur.next_buffer();
    const size_t need = type.id == TypeId::Bool ? bm : (size_t)c.length * (size_t)ipc_fixed_width(type);
    if (vals.second < need) throw CometError("shuffle block: values buffer too short");
    c.values.assign(vals.first, vals.first + need);
  }
  return c;
}

// indices (any signed width) + dictionary values → plain column (copy.rs:69-93 unpack semantics: null index → null)
HostColumn unpack_dictionary(const HostColumn& idx, int index_width, const HostColumn& dict) {
  HostColumn out;
  out.type = dict.type;
  out.length = idx.length;
  const int64_t n = idx.length;
  auto bit = [](const std::vector<uint8_t>& b, int64_t i) { return b.empty() || ((b[(size_t)(i >> 3)] >> (i & 7)) & 1); };
  std::vector<int64_t> ix((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    const uint8_t* p = idx.values.data() + (size_t)i * (size_t)index_width;
    ix[(size_t)i] = index_width == 1 ? rd<int8_t>(p) : index_width == 2 ? rd<int16_t>(p) : index_width == 4 ? rd<int32_t>(p) : rd<int64_t>(p);
  }
  std::vector<uint8_t> valid((size_t)((n + 7) / 8), 0);
  int64_t nulls = 0;
  for (int64_t i = 0; i < n; i++) {
    bool ok = bit(idx.validity, i);
    if (ok) {
      if (ix[(size_t)i] < 0 || ix[(size_t)i] >= dict.length) throw CometError("shuffle block: dictionary index out of range");
      ok = bit(dict.validity, ix[(size_t)i]);
    }
    if (ok) valid[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
    else nulls++;
  }
  out.null_count = nulls;
  if (nulls) out.validity = valid;
  auto is_ok = [&](int64_t i) { return (valid[(size_t)(i >> 3)] >> (i & 7)) & 1; };
  if (dict.type.id == TypeId::String || dict.type.id == TypeId::Bytes) {
    const int32_t* doff = (const int32_t*)dict.values.data();
    out.values.assign((size_t)(n + 1) * 4, 0);
    int32_t* o = (int32_t*)out.values.data();
    int64_t total = 0;
    for (int64_t i = 0; i < n; i++) {
      o[i] = (int32_t)total;
      if (is_ok(i)) total += doff[ix[(size_t)i] + 1] - doff[ix[(size_t)i]];
      if (total > INT32_MAX) throw CometError("shuffle block: unpacked string column exceeds 2 GiB");
    }
    o[n] = (int32_t)total;
    out.data.resize((size_t)total);
    for (int64_t i = 0; i < n; i++)
      if (is_ok(i)) memcpy(out.data.data() + o[i], dict.data.data() + doff[ix[(size_t)i]], (size_t)(o[i + 1] - o[i]));
  } else if (dict.type.id == TypeId::Bool) {
    out.values.assign((size_t)((n + 7) / 8), 0);
    for (int64_t i = 0; i < n; i++)
      if (is_ok(i) && ((dict.values[(size_t)(ix[(size_t)i] >> 3)] >> (ix[(size_t)i] & 7)) & 1)) out.values[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
  } else {
    const size_t w = (size_t)ipc_fixed_width(dict.type);
    out.values.assign((size_t)n * w, 0);
    for (int64_t i = 0; i < n; i++)
      if (is_ok(i)) memcpy(out.values.data() + (size_t)i * w, dict.values.data() + (size_t)ix[(size_t)i] * w, w);
  }
  return out;
}

// The Schema message at the head of an Arrow IPC stream → field names, types and nullability (the JVM hands schemas over this way:
// parquet/util/jni.rs deserialize_schema = StreamReader::try_new(bytes).schema()).  Field custom_metadata "PARQUET:field_id" is kept.
std::vector<StructField> decode_ipc_schema_impl(const uint8_t* p, size_t n) {
  size_t pos = 0;
  if (n < 8) throw CometError("Arrow IPC schema: truncated");
  int32_t meta_len;
  if (rd<uint32_t>(p) == 0xFFFFFFFFu) { meta_len = rd<int32_t>(p + 4); pos = 8; }
  else { meta_len = rd<int32_t>(p); pos = 4; }
  if (meta_len < 8 || pos + (size_t)meta_len > n) throw CometError("Arrow IPC schema: truncated metadata");
  FbTable msg;
  msg.buf = p + pos;
  msg.len = (size_t)meta_len;
  msg.pos = rd<uint32_t>(msg.buf);
  if (msg.pos + 4 > msg.len || msg.get<uint8_t>(1, 0) != MSG_Schema) throw CometError("Arrow IPC schema: the stream does not start with a Schema message");
  FbTable header = msg.child(2);
  if (!header.valid()) throw CometError("Arrow IPC schema: message without header");
  std::vector<StructField> out;
  size_t first;
  const size_t nf = header.vec(1, 4, first);
  for (size_t i = 0; i < nf; i++) {
    FbTable f = header.elem_table(first, i);
    StructField sf;
    sf.name = f.str(0);
    sf.nullable = f.get<uint8_t>(1, 0) != 0;
    sf.dtype = type_of_field(f);
    size_t kv0;
    const size_t nkv = f.vec(6, 4, kv0);        // Field.custom_metadata: [KeyValue{key, value}]
    for (size_t k = 0; k < nkv; k++) {
      FbTable kv = f.elem_table(kv0, k);
      if (kv.str(0) == "PARQUET:field_id") sf.field_id = atoi(kv.str(1).c_str());
    }
    out.push_back(sf);
  }
  return out;
}

HostBatch decode_ipc_stream(const uint8_t* p, size_t n) {
  std::vector<IpcField> fields;
  std::map<int64_t, HostColumn> dictionaries;
  bool have_schema = false;
  size_t pos = 0;
  while (pos + 4 <= n) {
    int32_t meta_len;
    if (rd<uint32_t>(p + pos) == 0xFFFFFFFFu) {
      if (pos + 8 > n) throw CometError("shuffle block: truncated IPC message header");
      meta_len = rd<int32_t>(p + pos + 4);
      pos += 8;
    } else {
      meta_len = rd<int32_t>(p + pos);   // pre-0.15 framing without the continuation marker
      pos += 4;
    }
    if (meta_len == 0) break;   // end of stream
    if (meta_len < 8 || pos + (size_t)meta_len > n) throw CometError("shuffle block: truncated IPC metadata");
    FbTable msg;
    msg.buf = p + pos;
    msg.len = (size_t)meta_len;
    msg.pos = rd<uint32_t>(msg.buf);
    if (msg.pos + 4 > msg.len) throw CometError("shuffle block: corrupt IPC metadata");
    pos += (size_t)meta_len;
    const int64_t body_len = msg.get<int64_t>(3, 0);
    if (body_len < 0 || pos + (size_t)body_len > n) throw CometError("shuffle block: truncated IPC message body");
    const uint8_t* body = p + pos;
    pos += (size_t)body_len;
    const int kind = msg.get<uint8_t>(1, 0);
    FbTable header = msg.child(2);
    if (!header.valid()) throw CometError("shuffle block: IPC message without header");
    if (kind == MSG_Schema) {
      if (header.get<int16_t>(0, 0) != 0) throw CometError("shuffle block: big-endian IPC streams are not supported");
      size_t first;
      const size_t nf = header.vec(1, 4, first);
      for (size_t i = 0; i < nf; i++) {
        FbTable f = header.elem_table(first, i);
        IpcField fld;
        fld.type = type_of_field(f);
        FbTable de = f.child(4);
        if (de.valid()) {
          fld.dict = true;
          fld.dict_id = de.get<int64_t>(0, 0);
          FbTable it = de.child(1);
          const int bits = it.valid() ? it.get<int32_t>(0, 32) : 32;
          if (it.valid() && it.get<uint8_t>(1, 0) == 0 && bits == 64) throw CometError("shuffle block: unsigned 64-bit dictionary indices are not supported");
          fld.index_width = bits / 8;
        }
        fields.push_back(fld);
      }
      have_schema = true;
      continue;
    }
    if (!have_schema) throw CometError("shuffle block: IPC stream does not start with a schema message");
    FbTable rb = kind == MSG_DictionaryBatch ? header.child(1) : header;
    if (!rb.valid()) throw CometError("shuffle block: dictionary batch without data");
    if (rb.child(3).valid()) throw CometError("shuffle block: IPC body compression is not supported (the block codec compresses the whole stream)");
    BodyCursor cur{body, (size_t)body_len, &rb};
    cur.n_nodes = rb.vec(1, 16, cur.nodes_first);
    cur.n_bufs = rb.vec(2, 16, cur.bufs_first);
    const int64_t rows = rb.get<int64_t>(0, 0);
    if (kind == MSG_DictionaryBatch) {
      if (header.get<uint8_t>(2, 0)) throw CometError("shuffle block: delta dictionaries are not supported");
      const int64_t id = header.get<int64_t>(0, 0);
      const IpcField* owner = nullptr;
      for (auto& f : fields)
        if (f.dict && f.dict_id == id) owner = &f;
      if (!owner) throw CometError("shuffle block: dictionary batch for an unknown dictionary id");
      dictionaries[id] = read_plain_column(owner->type, cur, rows);
      continue;
    }
    if (kind != MSG_RecordBatch) throw CometError("shuffle block: unexpected IPC message type " + std::to_string(kind));
    HostBatch b;
    b.rows = rows;
    for (auto& f : fields) {
      if (!f.dict) {
        b.cols.push_back(read_plain_column(f.type, cur, rows));
        continue;
      }
      DType it = DType::of(f.index_width == 1 ? TypeId::Int8 : f.index_width == 2 ? TypeId::Int16 : f.index_width == 4 ? TypeId::Int32 : TypeId::Int64);
      HostColumn idx = read_plain_column(it, cur, rows);
      auto d = dictionaries.find(f.dict_id);
      if (d == dictionaries.end()) throw CometError("shuffle block: record batch references a dictionary that was not sent");
      b.cols.push_back(unpack_dictionary(idx, f.index_width, d->second));
    }
    return b;   // one batch per block (ipc.rs: reader.next())
  }
  throw CometError("shuffle block: IPC stream holds no record batch");
}

const uint8_t kIpcEos[8] = {0xff, 0xff, 0xff, 0xff, 0, 0, 0, 0};

}  // namespace

std::vector<StructField> decode_ipc_schema(const uint8_t* p, size_t n) { return decode_ipc_schema_impl(p, n); }

size_t encode_shuffle_block(const std::vector<ColumnSlice>& cols, int64_t rows, ShuffleCodec codec, int level, std::vector<uint8_t>& out) {
  if (rows == 0) return 0;
  const size_t start = out.size();
  app<uint64_t>(out, 0);                       // length of the rest, filled in below
  app<uint64_t>(out, (uint64_t)cols.size());   // field count
  const char* tag = codec == ShuffleCodec::None ? "NONE" : codec == ShuffleCodec::Zstd ? "ZSTD" : codec == ShuffleCodec::Lz4 ? "LZ4_" : "SNAP";
  out.insert(out.end(), tag, tag + 4);
  if (codec == ShuffleCodec::None) {
    write_schema_message(cols, out);
    write_batch_message(cols, rows, out);
    out.insert(out.end(), kIpcEos, kIpcEos + 8);
  } else {
    static thread_local std::vector<uint8_t> ipc;
    ipc.clear();
    write_schema_message(cols, ipc);
    write_batch_message(cols, rows, ipc);
    ipc.insert(ipc.end(), kIpcEos, kIpcEos + 8);
    if (codec == ShuffleCodec::Zstd) zstd_encode(ipc.data(), ipc.size(), level, out);
    else if (codec == ShuffleCodec::Lz4) lz4_frame_encode(ipc.data(), ipc.size(), out);
    else snappy_frame_encode(ipc.data(), ipc.size(), out);
  }
  const uint64_t rest = out.size() - start - 8;
  if (rest > (uint64_t)INT32_MAX)
    throw CometError("Shuffle block size " + std::to_string(rest) + " exceeds maximum size of " + std::to_string(INT32_MAX) +
                     ". Try reducing batch size or increasing compression level");
  memcpy(out.data() + start, &rest, 8);
  return out.size() - start;
}

HostBatch decode_shuffle_block(const uint8_t* block, size_t len) {
  if (len < 4) throw CometError("Failed to decode batch: block shorter than its codec tag");
  const uint8_t* p = block + 4;
  const size_t n = len - 4;
  if (memcmp(block, "NONE", 4) == 0) return decode_ipc_stream(p, n);
  std::vector<uint8_t> raw;
  if (memcmp(block, "ZSTD", 4) == 0) raw = zstd_decode(p, n);
  else if (memcmp(block, "LZ4_", 4) == 0) raw = lz4_frame_decode(p, n);
  else if (memcmp(block, "SNAP", 4) == 0) raw = snappy_frame_decode(p, n);
  else throw CometError("Failed to decode batch: invalid compression codec: " + std::string((const char*)block, 4));
  return decode_ipc_stream(raw.data(), raw.size());
}

// ---------------------------------------------------------------------------------------------------------------
// Arrow C Data export of a HostBatch, and the block stream → ArrowArrayStream adapter
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct ExportedColumn {
  HostColumn col;
  const void* buffers[3];
  std::vector<ArrowArray> children;        // nested columns: the children's arrays live here, released with their parent
  std::vector<ArrowArray*> child_ptrs;
};
void release_array(ArrowArray* a) {
  auto* ec = (ExportedColumn*)a->private_data;
  for (auto& c : ec->children)
    if (c.release) c.release(&c);
  delete ec;
  a->release = nullptr;
}
struct ExportedSchema {
  std::string format, name;
  std::vector<ArrowSchema> children;
  std::vector<ArrowSchema*> child_ptrs;
};
void release_schema(ArrowSchema* s) {
  auto* es = (ExportedSchema*)s->private_data;
  for (auto& c : es->children)
    if (c.release) c.release(&c);
  delete es;
  s->release = nullptr;
}
void export_column(HostColumn&& col, ArrowArray* a) {
  auto* ec = new ExportedColumn();
  ec->col = std::move(col);
  memset(a, 0, sizeof *a);
  a->length = ec->col.length;
  a->null_count = ec->col.null_count;
  const TypeId id = ec->col.type.id;
  const bool is_str = id == TypeId::String || id == TypeId::Bytes;
  static const uint8_t kEmpty[8] = {0};
  ec->buffers[0] = ec->col.null_count ? ec->col.validity.data() : nullptr;
  if (id == TypeId::Struct) {
    a->n_buffers = 1;                       // Arrow struct layout: validity only
  } else {
    a->n_buffers = is_str ? 3 : 2;          // (a List: validity + int32 offsets)
    ec->buffers[1] = ec->col.values.empty() ? (const void*)kEmpty : (const void*)ec->col.values.data();
    ec->buffers[2] = is_str ? (ec->col.data.empty() ? (const void*)kEmpty : (const void*)ec->col.data.data()) : nullptr;
  }
  if (id == TypeId::Struct || id == TypeId::List || id == TypeId::Map) {
    ec->children.resize(ec->col.children.size());
    for (size_t i = 0; i < ec->col.children.size(); i++) {
      export_column(std::move(ec->col.children[i]), &ec->children[i]);
      ec->child_ptrs.push_back(&ec->children[i]);
    }
    ec->col.children.clear();
    a->n_children = (int64_t)ec->children.size();
    a->children = ec->child_ptrs.data();
  }
  a->buffers = ec->buffers;
  a->private_data = ec;
  a->release = release_array;
}
void export_schema_named(const DType& t, const std::string& name, bool nullable, ArrowSchema* s) {
  memset(s, 0, sizeof *s);
  auto* es = new ExportedSchema();
  es->format = expected_format(t);
  es->name = name;
  s->format = es->format.c_str();
  s->name = es->name.c_str();
  s->flags = nullable ? ARROW_FLAG_NULLABLE : 0;
  if (t.id == TypeId::Struct || t.id == TypeId::List || t.id == TypeId::Map) {
    es->children.resize(t.kids.size());
    for (size_t i = 0; i < t.kids.size(); i++) {
      const std::string kn = t.id == TypeId::List ? "element" : t.id == TypeId::Map ? "entries" : (i < t.kid_names.size() ? t.kid_names[i] : std::string());
      export_schema_named(t.kids[i], kn, t.id == TypeId::Map ? false : (i < t.kid_nullable.size() ? t.kid_nullable[i] != 0 : true), &es->children[i]);
      es->child_ptrs.push_back(&es->children[i]);
    }
    s->n_children = (int64_t)es->children.size();
    s->children = es->child_ptrs.data();
  }
  s->private_data = es;
  s->release = release_schema;
}
void export_schema(const DType& t, ArrowSchema* s) { export_schema_named(t, "", true, s); }
}  // namespace

std::string expected_format(const DType& t) {
  switch (t.id) {
    case TypeId::Bool: return "b";
    case TypeId::Int8: return "c";
    case TypeId::Int16: return "s";
    case TypeId::Int32: return "i";
    case TypeId::Int64: return "l";
    case TypeId::Float: return "f";
    case TypeId::Double: return "g";
    case TypeId::Date: return "tdD";
    case TypeId::Timestamp: return "tsu:UTC";
    case TypeId::TimestampNtz: return "tsu:";
    case TypeId::String: return "u";
    case TypeId::Bytes: return "z";
    case TypeId::Decimal: return "d:" + std::to_string(t.precision) + "," + std::to_string(t.scale);
    case TypeId::Struct: return "+s";
    case TypeId::List: return "+l";
    case TypeId::Map: return "+m";
    default: return "?";
  }
}

void export_host_batch(HostBatch& b, ArrowArray** out_arrays, ArrowSchema** out_schemas, int n_out) {
  if ((size_t)n_out != b.cols.size())
    throw CometError("Output column count mismatch: expected " + std::to_string(n_out) + ", got " + std::to_string(b.cols.size()));
  for (int j = 0; j < n_out; j++) {
    const DType t = b.cols[(size_t)j].type;
    export_column(std::move(b.cols[(size_t)j]), out_arrays[j]);
    export_schema(t, out_schemas[j]);
  }
}

namespace {
struct BlockStream {
  CometShuffleBlockStreamC* blocks;
  std::vector<DType> types;
  std::string error;
};
struct StructHolder {
  std::vector<ArrowArray> children;
  std::vector<ArrowArray*> child_ptrs;
  const void* buffers[1] = {nullptr};
};
struct SchemaHolder {
  std::vector<ArrowSchema> children;
  std::vector<ArrowSchema*> child_ptrs;
};
void release_struct(ArrowArray* a) {
  auto* h = (StructHolder*)a->private_data;
  for (auto& c : h->children)
    if (c.release) c.release(&c);
  delete h;
  a->release = nullptr;
}
void release_struct_schema(ArrowSchema* s) {
  auto* h = (SchemaHolder*)s->private_data;
  for (auto& c : h->children)
    if (c.release) c.release(&c);
  delete h;
  s->release = nullptr;
}
int bs_get_schema(ArrowArrayStream* st, ArrowSchema* out) {
  auto* s = (BlockStream*)st->private_data;
  auto* h = new SchemaHolder();
  h->children.resize(s->types.size());
  for (size_t i = 0; i < s->types.size(); i++) {
    export_schema(s->types[i], &h->children[i]);
    h->child_ptrs.push_back(&h->children[i]);
  }
  memset(out, 0, sizeof *out);
  out->format = "+s";
  out->name = "";
  out->n_children = (int64_t)s->types.size();
  out->children = h->child_ptrs.data();
  out->private_data = h;
  out->release = release_struct_schema;
  return 0;
}
int bs_get_next(ArrowArrayStream* st, ArrowArray* out) {
  auto* s = (BlockStream*)st->private_data;
  try {
    const uint8_t* data = nullptr;
    const int64_t len = s->blocks->next_block(s->blocks, &data);
    if (len == -1) {
      memset(out, 0, sizeof *out);   // released array = end of stream
      return 0;
    }
    if (len < 0) {
      const char* e = s->blocks->get_last_error ? s->blocks->get_last_error(s->blocks) : nullptr;
      throw CometError(std::string("shuffle block iterator failed") + (e ? std::string(": ") + e : std::string()));
    }
    HostBatch b = decode_shuffle_block(data, (size_t)len);
    if (b.cols.size() != s->types.size())
      throw CometError("Shuffle block column count mismatch: got " + std::to_string(b.cols.size()) + " but expected " + std::to_string(s->types.size()));
    for (size_t i = 0; i < b.cols.size(); i++)
      if (b.cols[i].type != s->types[i])
        throw CometError("Shuffle block column " + std::to_string(i) + " has type " + b.cols[i].type.str() + " but the plan declares " + s->types[i].str());
    auto* h = new StructHolder();
    h->children.resize(b.cols.size());
    for (size_t i = 0; i < b.cols.size(); i++) {
      export_column(std::move(b.cols[i]), &h->children[i]);
      h->child_ptrs.push_back(&h->children[i]);
    }
    memset(out, 0, sizeof *out);
    out->length = b.rows;
    out->n_buffers = 1;
    out->buffers = h->buffers;
    out->n_children = (int64_t)h->children.size();
    out->children = h->child_ptrs.data();
    out->private_data = h;
    out->release = release_struct;
    return 0;
  } catch (const std::exception& e) {
    s->error = e.what();
    return 5;   // EIO
  }
}
const char* bs_last_error(ArrowArrayStream* st) { return ((BlockStream*)st->private_data)->error.c_str(); }
void bs_release(ArrowArrayStream* st) {
  auto* s = (BlockStream*)st->private_data;
  if (s->blocks && s->blocks->release) s->blocks->release(s->blocks);
  delete s;
  delete st;   // the adapter struct itself is ours (allocated in shuffle_blocks_as_arrow_stream)
}
}  // namespace

ArrowArrayStream* shuffle_blocks_as_arrow_stream(CometShuffleBlockStreamC* blocks, std::vector<DType> types) {
  auto* s = new BlockStream{blocks, std::move(types), ""};
  auto* st = new ArrowArrayStream();
  st->get_schema = bs_get_schema;
  st->get_next = bs_get_next;
  st->get_last_error = bs_last_error;
  st->release = bs_release;
  st->private_data = s;
  return st;
}

}  // namespace comet
