// String functions whose result is NEW bytes (not a slice of the source value), and the byte-oriented hashes: what a chain's DERIVED Utf8 columns are computed
// with (codegen.hpp DerivedCol kind 3; strfn_kernels.hip: one pass for the lengths, a prefix sum, one pass that writes) and what the fused kernels call for
// instr / ascii / crc32.  The reference hands these to DataFusion / datafusion-spark (QueryPlanSerde.scala:208-249; jni_api.rs:635-670):
//   reverse            the value's scalar values in reverse order (str.chars().rev())
//   repeat(n)          the value n times (n ≥ 0)
//   replace(from, to)  Rust's str::replace — every non-overlapping occurrence, left to right; an EMPTY `from` matches before every character and at the end
//   substring_index(delim, count)   DataFusion's substr_index: what lies before the count-th occurrence of delim (count > 0) or behind the |count|-th from the
//                      right (count < 0), occurrences counted without overlap; the whole value when there are fewer; "" for count = 0 or an empty delim
//   md5 / sha1 / sha2(224 | 256 | 384 | 512)   the digest's lower-case hexadecimal digits (RFC 1321, FIPS 180-4)
//   crc32              zlib's CRC-32 of the bytes;  instr: the 1-based CHARACTER position of the first occurrence (0: none);  ascii: the first scalar value (0: empty)
// Plain C++ over a byte pointer: comet_device.hpp and strfn_kernels.hip include it for the device, capi.cpp runs the same source on the host for the CPU
// tests (comet_strfn_host) against Python's str methods, hashlib and zlib.
#pragma once
typedef unsigned char sf_u8;
typedef unsigned int sf_u32;
typedef int sf_i32;
typedef unsigned long long sf_u64;
typedef long long sf_i64;
#ifndef STRFN
#define STRFN inline
#endif

enum { SF_REVERSE = 1, SF_REPEAT = 2, SF_REPLACE = 3, SF_SUBSTRING_INDEX = 4, SF_MD5 = 10, SF_SHA1 = 11, SF_SHA224 = 12, SF_SHA256 = 13, SF_SHA384 = 14, SF_SHA512 = 15 };

// first occurrence of pat[0, m) in text[from, n): its byte offset, or -1 (m = 0: `from`)
template <class P, class Q>
STRFN sf_i32 sf_find(P text, sf_i32 n, sf_i32 from, Q pat, sf_i32 m) {
  for (sf_i32 i = from; i + m <= n; i++) {
    sf_i32 k = 0;
    while (k < m && text[i + k] == pat[k]) k++;
    if (k == m) return i;
  }
  return -1;
}
template <class P, class Q>
STRFN sf_i32 sf_rfind(P text, sf_i32 until, Q pat, sf_i32 m) {      // the last occurrence that ENDS at or before `until`
  for (sf_i32 i = until - m; i >= 0; i--) {
    sf_i32 k = 0;
    while (k < m && text[i + k] == pat[k]) k++;
    if (k == m) return i;
  }
  return -1;
}
template <class P>
STRFN sf_i32 sf_char_len(P p, sf_i32 i, sf_i32 n) {      // bytes of the character that starts at i (valid UTF-8; anything else: one byte)
  const sf_u32 c = p[i];
  sf_i32 l = c < 0x80u ? 1 : c < 0xE0u ? 2 : c < 0xF0u ? 3 : 4;
  if (c >= 0x80u && c < 0xC0u) l = 1;
  return i + l <= n ? l : n - i;
}

// ---- substring_index: the slice [*a, *b) of the value ----
template <class P>
STRFN void sf_substring_index(P p, sf_i32 n, const sf_u8* d, sf_i32 m, sf_i64 count, sf_i32& a, sf_i32& b) {
  a = 0;
  b = 0;
  if (count == 0 || m == 0) return;
  if (count > 0) {
    sf_i32 at = 0;
    for (sf_i64 k = 0; k < count; k++) {
      const sf_i32 f = sf_find(p, n, at, d, m);
      if (f < 0) { b = n; return; }      // fewer occurrences: the whole value
      if (k == count - 1) { b = f; return; }
      at = f + m;
    }
    b = n;
    return;
  }
  sf_i32 until = n;
  for (sf_i64 k = 0; k < -count; k++) {
    const sf_i32 f = sf_rfind(p, until, d, m);
    if (f < 0) { a = 0; b = n; return; }
    if (k == -count - 1) { a = f + m; b = n; return; }
    until = f;
  }
  b = n;
}

// ---- digests ----
STRFN sf_u32 sf_rotl(sf_u32 x, int c) { return (x << c) | (x >> (32 - c)); }
STRFN sf_u32 sf_rotr(sf_u32 x, int c) { return (x >> c) | (x << (32 - c)); }
STRFN sf_u64 sf_rotr64(sf_u64 x, int c) { return (x >> c) | (x << (64 - c)); }

// the message as 64- or 128-byte blocks: block k's byte j, with the 0x80 byte and the bit length behind the message (little-endian length: md5)
template <class P>
STRFN sf_u8 sf_padded(P p, sf_i64 n, sf_i64 total, sf_i64 at, bool le_len, int len_bytes) {
  if (at < n) return p[at];
  if (at == n) return 0x80u;
  const sf_i64 tail = total - at;      // 1 … len_bytes: inside the length field
  if (tail > len_bytes) return 0;
  const sf_u64 bits = (sf_u64)n * 8ull;
  const int k = (int)(len_bytes - tail);      // index from the field's first byte
  if (le_len) return k < 8 ? (sf_u8)(bits >> (8 * k)) : 0;
  const int from_end = len_bytes - 1 - k;
  return from_end < 8 ? (sf_u8)(bits >> (8 * from_end)) : 0;
}

template <class P>
STRFN void sf_md5(P p, sf_i64 n, sf_u8* out16) {
  const sf_u32 K[64] = {0xd76aa478u, 0xe8c7b756u, 0x242070dbu, 0xc1bdceeeu, 0xf57c0fafu, 0x4787c62au, 0xa8304613u, 0xfd469501u, 0x698098d8u, 0x8b44f7afu, 0xffff5bb1u, 0x895cd7beu, 0x6b901122u,
                        0xfd987193u, 0xa679438eu, 0x49b40821u, 0xf61e2562u, 0xc040b340u, 0x265e5a51u, 0xe9b6c7aau, 0xd62f105du, 0x02441453u, 0xd8a1e681u, 0xe7d3fbc8u, 0x21e1cde6u, 0xc33707d6u,
                        0xf4d50d87u, 0x455a14edu, 0xa9e3e905u, 0xfcefa3f8u, 0x676f02d9u, 0x8d2a4c8au, 0xfffa3942u, 0x8771f681u, 0x6d9d6122u, 0xfde5380cu, 0xa4beea44u, 0x4bdecfa9u, 0xf6bb4b60u,
                        0xbebfbc70u, 0x289b7ec6u, 0xeaa127fau, 0xd4ef3085u, 0x04881d05u, 0xd9d4d039u, 0xe6db99e5u, 0x1fa27cf8u, 0xc4ac5665u, 0xf4292244u, 0x432aff97u, 0xab9423a7u, 0xfc93a039u,
                        0x655b59c3u, 0x8f0ccc92u, 0xffeff47du, 0x85845dd1u, 0x6fa87e4fu, 0xfe2ce6e0u, 0xa3014314u, 0x4e0811a1u, 0xf7537e82u, 0xbd3af235u, 0x2ad7d2bbu, 0xeb86d391u};
  const int S[16] = {7, 12, 17, 22, 5, 9, 14, 20, 4, 11, 16, 23, 6, 10, 15, 21};
  sf_u32 h0 = 0x67452301u, h1 = 0xefcdab89u, h2 = 0x98badcfeu, h3 = 0x10325476u;
  const sf_i64 total = ((n + 8) / 64 + 1) * 64;
  for (sf_i64 base = 0; base < total; base += 64) {
    sf_u32 w[16];
    for (int j = 0; j < 16; j++) {
      sf_u32 v = 0;
      for (int b = 0; b < 4; b++) v |= (sf_u32)sf_padded(p, n, total, base + 4 * j + b, true, 8) << (8 * b);
      w[j] = v;
    }
    sf_u32 a = h0, b = h1, c = h2, d = h3;
    for (int i = 0; i < 64; i++) {
      sf_u32 f;
      int g;
      if (i < 16) { f = (b & c) | (~b & d); g = i; }
      else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
      else { f = c ^ (b | ~d); g = (7 * i) & 15; }
      const sf_u32 t = d;
      d = c;
      c = b;
      b = b + sf_rotl(a + f + K[i] + w[g], S[(i >> 4) * 4 + (i & 3)]);
      a = t;
    }
    h0 += a; h1 += b; h2 += c; h3 += d;
  }
  const sf_u32 h[4] = {h0, h1, h2, h3};
  for (int j = 0; j < 16; j++) out16[j] = (sf_u8)(h[j >> 2] >> (8 * (j & 3)));
}

template <class P>
STRFN void sf_sha1(P p, sf_i64 n, sf_u8* out20) {
  sf_u32 h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
  const sf_i64 total = ((n + 8) / 64 + 1) * 64;
  for (sf_i64 base = 0; base < total; base += 64) {
    sf_u32 w[80];
    for (int j = 0; j < 16; j++) {
      sf_u32 v = 0;
      for (int b = 0; b < 4; b++) v = (v << 8) | sf_padded(p, n, total, base + 4 * j + b, false, 8);
      w[j] = v;
    }
    for (int j = 16; j < 80; j++) w[j] = sf_rotl(w[j - 3] ^ w[j - 8] ^ w[j - 14] ^ w[j - 16], 1);
    sf_u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
    for (int i = 0; i < 80; i++) {
      sf_u32 f, k;
      if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
      else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
      else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
      else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
      const sf_u32 t = sf_rotl(a, 5) + f + e + k + w[i];
      e = d; d = c; c = sf_rotl(b, 30); b = a; a = t;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
  }
  for (int j = 0; j < 20; j++) out20[j] = (sf_u8)(h[j >> 2] >> (24 - 8 * (j & 3)));
}

template <class P>
STRFN void sf_sha256(P p, sf_i64 n, bool is224, sf_u8* out32) {
  const sf_u32 K[64] = {0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u,
                        0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du,
                        0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu,
                        0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u,
                        0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
  sf_u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
  if (is224) {
    const sf_u32 i224[8] = {0xc1059ed8u, 0x367cd507u, 0x3070dd17u, 0xf70e5939u, 0xffc00b31u, 0x68581511u, 0x64f98fa7u, 0xbefa4fa4u};
    for (int j = 0; j < 8; j++) h[j] = i224[j];
  }
  const sf_i64 total = ((n + 8) / 64 + 1) * 64;
  for (sf_i64 base = 0; base < total; base += 64) {
    sf_u32 w[64];
    for (int j = 0; j < 16; j++) {
      sf_u32 v = 0;
      for (int b = 0; b < 4; b++) v = (v << 8) | sf_padded(p, n, total, base + 4 * j + b, false, 8);
      w[j] = v;
    }
    for (int j = 16; j < 64; j++) {
      const sf_u32 s0 = sf_rotr(w[j - 15], 7) ^ sf_rotr(w[j - 15], 18) ^ (w[j - 15] >> 3), s1 = sf_rotr(w[j - 2], 17) ^ sf_rotr(w[j - 2], 19) ^ (w[j - 2] >> 10);
      w[j] = w[j - 16] + s0 + w[j - 7] + s1;
    }
    sf_u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      const sf_u32 S1 = sf_rotr(e, 6) ^ sf_rotr(e, 11) ^ sf_rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
      const sf_u32 S0 = sf_rotr(a, 2) ^ sf_rotr(a, 13) ^ sf_rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  for (int j = 0; j < 32; j++) out32[j] = (sf_u8)(h[j >> 2] >> (24 - 8 * (j & 3)));
}

template <class P>
STRFN void sf_sha512(P p, sf_i64 n, bool is384, sf_u8* out64) {
  const sf_u64 K[80] = {
      0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull, 0x3956c25bf348b538ull, 0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull,
      0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull, 0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull,
      0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull, 0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
      0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull, 0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull, 0x06ca6351e003826full, 0x142929670a0e6e70ull,
      0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull, 0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
      0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull, 0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
      0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull, 0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull,
      0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull, 0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull,
      0xca273eceea26619cull, 0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull, 0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
      0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull, 0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};
  sf_u64 h[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
  if (is384) {
    const sf_u64 i384[8] = {0xcbbb9d5dc1059ed8ull, 0x629a292a367cd507ull, 0x9159015a3070dd17ull, 0x152fecd8f70e5939ull, 0x67332667ffc00b31ull, 0x8eb44a8768581511ull, 0xdb0c2e0d64f98fa7ull, 0x47b5481dbefa4fa4ull};
    for (int j = 0; j < 8; j++) h[j] = i384[j];
  }
  const sf_i64 total = ((n + 16) / 128 + 1) * 128;
  for (sf_i64 base = 0; base < total; base += 128) {
    sf_u64 w[80];
    for (int j = 0; j < 16; j++) {
      sf_u64 v = 0;
      for (int b = 0; b < 8; b++) v = (v << 8) | sf_padded(p, n, total, base + 8 * j + b, false, 16);
      w[j] = v;
    }
    for (int j = 16; j < 80; j++) {
      const sf_u64 s0 = sf_rotr64(w[j - 15], 1) ^ sf_rotr64(w[j - 15], 8) ^ (w[j - 15] >> 7), s1 = sf_rotr64(w[j - 2], 19) ^ sf_rotr64(w[j - 2], 61) ^ (w[j - 2] >> 6);
      w[j] = w[j - 16] + s0 + w[j - 7] + s1;
    }
    sf_u64 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 80; i++) {
      const sf_u64 S1 = sf_rotr64(e, 14) ^ sf_rotr64(e, 18) ^ sf_rotr64(e, 41), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
      const sf_u64 S0 = sf_rotr64(a, 28) ^ sf_rotr64(a, 34) ^ sf_rotr64(a, 39), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  for (int j = 0; j < 64; j++) out64[j] = (sf_u8)(h[j >> 3] >> (56 - 8 * (j & 7)));
}

template <class P>
STRFN sf_u32 sf_crc32(P p, sf_i64 n) {      // zlib's: polynomial 0xEDB88320, reflected, initial and final complement
  sf_u32 c = 0xffffffffu;
  for (sf_i64 i = 0; i < n; i++) {
    c ^= p[i];
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
  }
  return ~c;
}
template <class P, class Q>
STRFN sf_i32 sf_instr(P p, sf_i32 n, Q sub, sf_i32 m) {      // 1-based character position of the first occurrence; 0: none
  const sf_i32 f = sf_find(p, n, 0, sub, m);
  if (f < 0) return 0;
  sf_i32 chars = 0;
  for (sf_i32 i = 0; i < f; i++) chars += (p[i] & 0xC0u) != 0x80u;
  return chars + 1;
}
template <class P>
STRFN sf_i32 sf_ascii(P p, sf_i32 n) {      // the first scalar value; 0 for the empty string
  if (n <= 0) return 0;
  const sf_u32 c = p[0];
  if (c < 0x80u || n < 2) return (sf_i32)c;
  if (c < 0xE0u) return (sf_i32)(((c & 0x1Fu) << 6) | (p[1] & 0x3Fu));
  if (c < 0xF0u && n >= 3) return (sf_i32)(((c & 0x0Fu) << 12) | ((sf_u32)(p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu));
  if (n >= 4) return (sf_i32)(((c & 0x07u) << 18) | ((sf_u32)(p[1] & 0x3Fu) << 12) | ((sf_u32)(p[2] & 0x3Fu) << 6) | (p[3] & 0x3Fu));
  return (sf_i32)c;
}

// ---- the two passes: the result's byte count, then its bytes.  a / b: the literal arguments, k: the integer argument ----
template <class P>
STRFN sf_i64 sf_len(int op, P p, sf_i32 n, const sf_u8* a, sf_i32 na, const sf_u8* b, sf_i32 nb, sf_i64 k) {
  switch (op) {
    case SF_REVERSE: return n;
    case SF_REPEAT: return k <= 0 ? 0 : (sf_i64)n * k;
    case SF_REPLACE: {
      if (na == 0) {      // before every character and at the end
        sf_i64 chars = 0;
        for (sf_i32 i = 0; i < n; i++) chars += (p[i] & 0xC0u) != 0x80u;
        return (sf_i64)n + (chars + 1) * nb;
      }
      sf_i64 out = 0;
      sf_i32 at = 0;
      while (true) {
        const sf_i32 f = sf_find(p, n, at, a, na);
        if (f < 0) break;
        out += (f - at) + nb;
        at = f + na;
      }
      return out + (n - at);
    }
    case SF_SUBSTRING_INDEX: {
      sf_i32 x, y;
      sf_substring_index(p, n, a, na, k, x, y);
      return y - x;
    }
    case SF_MD5: return 32;
    case SF_SHA1: return 40;
    case SF_SHA224: return 56;
    case SF_SHA256: return 64;
    case SF_SHA384: return 96;
    case SF_SHA512: return 128;
    default: return 0;
  }
}
STRFN void sf_hex(const sf_u8* d, int nd, sf_u8* out) {
  for (int j = 0; j < nd; j++) {
    const sf_u32 hi = d[j] >> 4, lo = d[j] & 15u;
    out[2 * j] = (sf_u8)(hi < 10 ? '0' + hi : 'a' + hi - 10);
    out[2 * j + 1] = (sf_u8)(lo < 10 ? '0' + lo : 'a' + lo - 10);
  }
}
template <class P>
STRFN void sf_write(int op, P p, sf_i32 n, const sf_u8* a, sf_i32 na, const sf_u8* b, sf_i32 nb, sf_i64 k, sf_u8* out) {
  switch (op) {
    case SF_REVERSE: {
      sf_i32 i = 0, o = n;
      while (i < n) {
        const sf_i32 l = sf_char_len(p, i, n);
        o -= l;
        for (sf_i32 j = 0; j < l; j++) out[o + j] = p[i + j];
        i += l;
      }
      break;
    }
    case SF_REPEAT:
      for (sf_i64 r = 0; r < k; r++)
        for (sf_i32 j = 0; j < n; j++) out[r * n + j] = p[j];
      break;
    case SF_REPLACE: {
      sf_i64 o = 0;
      if (na == 0) {
        sf_i32 i = 0;
        while (true) {
          for (sf_i32 j = 0; j < nb; j++) out[o++] = b[j];
          if (i >= n) break;
          const sf_i32 l = sf_char_len(p, i, n);
          for (sf_i32 j = 0; j < l; j++) out[o++] = p[i + j];
          i += l;
        }
        break;
      }
      sf_i32 at = 0;
      while (true) {
        const sf_i32 f = sf_find(p, n, at, a, na);
        if (f < 0) break;
        for (sf_i32 j = at; j < f; j++) out[o++] = p[j];
        for (sf_i32 j = 0; j < nb; j++) out[o++] = b[j];
        at = f + na;
      }
      for (sf_i32 j = at; j < n; j++) out[o++] = p[j];
      break;
    }
    case SF_SUBSTRING_INDEX: {
      sf_i32 x, y;
      sf_substring_index(p, n, a, na, k, x, y);
      for (sf_i32 j = x; j < y; j++) out[j - x] = p[j];
      break;
    }
    case SF_MD5: { sf_u8 d[16]; sf_md5(p, n, d); sf_hex(d, 16, out); break; }
    case SF_SHA1: { sf_u8 d[20]; sf_sha1(p, n, d); sf_hex(d, 20, out); break; }
    case SF_SHA224: { sf_u8 d[32]; sf_sha256(p, n, true, d); sf_hex(d, 28, out); break; }
    case SF_SHA256: { sf_u8 d[32]; sf_sha256(p, n, false, d); sf_hex(d, 32, out); break; }
    case SF_SHA384: { sf_u8 d[64]; sf_sha512(p, n, true, d); sf_hex(d, 48, out); break; }
    case SF_SHA512: { sf_u8 d[64]; sf_sha512(p, n, false, d); sf_hex(d, 64, out); break; }
    default: break;
  }
}
