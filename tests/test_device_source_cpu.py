"""The XXH64 / murmur3 and decimal device functions of csrc/device/comet_device.hpp are plain integer arithmetic: their text is compiled for the host
here (gcc, CDEV = static inline) and checked against the independent `xxhash` package and the reference's own vectors
(spark-expr/src/hash_funcs/xxhash64.rs:155-240, murmur3.rs:209-265) — a CPU-side check of the exact source the GPU pipelines JIT."""
import ctypes
import os
import re
import subprocess

import numpy as np
import xxhash

_HERE = os.path.dirname(os.path.abspath(__file__))
_HDR = os.path.join(_HERE, "..", "datafusion-comet_amd", "csrc", "device", "comet_device.hpp")


def _build(tmp_path):
    src = open(_HDR).read()
    a = src.index("CDEV u32 rotl32(u32 x, int r)")
    b = src.index("CDEV i32 pmod(u32 hash, i32 n)")
    body = src[a:b]
    shim = """
#include <stdint.h>
#include <string.h>
typedef long long i64; typedef unsigned long long u64; typedef int i32; typedef unsigned int u32; typedef short i16; typedef signed char i8;
typedef unsigned char u8; typedef __int128 i128; typedef unsigned __int128 u128;
#define CDEV static inline
static inline i64 __double_as_longlong(double d) { i64 x; memcpy(&x, &d, 8); return x; }
static inline i32 __float_as_int(float f) { i32 x; memcpy(&x, &f, 4); return x; }
""" + body + """
extern "C" {
u64 t_xx_i32(i32 v, u64 s) { return xxh64_hash_i32(v, s); }
u64 t_xx_i64(i64 v, u64 s) { return xxh64_hash_i64(v, s); }
u64 t_xx_i128(u64 lo, u64 hi, u64 s) { return xxh64_hash_i128((i128)(((u128)hi << 64) | lo), s); }
u64 t_xx_f64(double v, u64 s) { return xxh64_hash_f64(v, s); }
u64 t_xx_f32(float v, u64 s) { return xxh64_hash_f32(v, s); }
u32 t_mm_i32(i32 v, u32 s) { return mm3_hash_i32(v, s); }
u32 t_mm_i64(i64 v, u32 s) { return mm3_hash_i64(v, s); }
}
"""
    c = tmp_path / "dev_hash.cpp"
    c.write_text(shim)
    so = tmp_path / "libdevhash.so"
    subprocess.check_call(["g++", "-O1", "-fPIC", "-shared", "-Wno-unused-function", "-o", str(so), str(c)])
    m = ctypes.CDLL(str(so))
    u64, u32 = ctypes.c_uint64, ctypes.c_uint32
    m.t_xx_i32.restype, m.t_xx_i32.argtypes = u64, [ctypes.c_int32, u64]
    m.t_xx_i64.restype, m.t_xx_i64.argtypes = u64, [ctypes.c_int64, u64]
    m.t_xx_i128.restype, m.t_xx_i128.argtypes = u64, [u64, u64, u64]
    m.t_xx_f64.restype, m.t_xx_f64.argtypes = u64, [ctypes.c_double, u64]
    m.t_xx_f32.restype, m.t_xx_f32.argtypes = u64, [ctypes.c_float, u64]
    m.t_mm_i32.restype, m.t_mm_i32.argtypes = u32, [ctypes.c_int32, u32]
    m.t_mm_i64.restype, m.t_mm_i64.argtypes = u32, [ctypes.c_int64, u32]
    return m


def test_device_hash_source_on_host(tmp_path):
    m = _build(tmp_path)
    # the reference's vectors, seed 42
    assert [m.t_xx_i32(v, 42) for v in (1, 0, -1, 2**31 - 1, -2**31)] == [0xa309b38455455929, 0x3229fbc4681e48f3, 0x1bfdda8861c06e45, 0x14f0ac009c21721c,
                                                                         0x1cc7cb8d034769cd]
    assert [m.t_xx_i64(v, 42) for v in (1, 0, -1, 2**63 - 1, -2**63)] == [0x9ed50fd59358d232, 0xb71b47ebda15746c, 0x358ae035bfb46fd2, 0xd2f1c616ae7eb306,
                                                                         0x88608019c494c1f4]
    assert [m.t_mm_i32(v, 42) for v in (1, 0, -1, 2**31 - 1, -2**31)] == [0xdea578e3, 0x379fae8f, 0xa0590e3d, 0x07fb67e7, 0x2b1f0fc6]
    assert [m.t_mm_i64(v, 42) for v in (1, 0, -1, 2**63 - 1, -2**63)] == [0x99f0149d, 0x9c67b85d, 0xc8008529, 0xa05b5d7b, 0xcd1e64fb]
    assert m.t_xx_f64(0.0, 42) == m.t_xx_f64(-0.0, 42) == 0xb71b47ebda15746c
    # random values and seeds against the independent implementation
    rng = np.random.default_rng(4)
    for _ in range(2000):
        seed = int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2))
        v32, v64 = int(rng.integers(-2**31, 2**31)), int(rng.integers(-2**63, 2**63 - 1))
        lo, hi = int(rng.integers(0, 2**63)) * 2 + 1, int(rng.integers(0, 2**63)) * 2
        assert m.t_xx_i32(v32, seed) == xxhash.xxh64_intdigest((v32 & 0xFFFFFFFF).to_bytes(4, "little"), seed)
        assert m.t_xx_i64(v64, seed) == xxhash.xxh64_intdigest((v64 & (2**64 - 1)).to_bytes(8, "little"), seed)
        assert m.t_xx_i128(lo, hi, seed) == xxhash.xxh64_intdigest(lo.to_bytes(8, "little") + hi.to_bytes(8, "little"), seed)
        d = float(rng.standard_normal())
        assert m.t_xx_f64(d, seed) == xxhash.xxh64_intdigest(np.float64(d).tobytes(), seed)
        f = np.float32(rng.standard_normal())
        assert m.t_xx_f32(float(f), seed) == xxhash.xxh64_intdigest(f.tobytes(), seed)


def _build_decimal(tmp_path):
    src = open(_HDR).read()
    a = src.index("CDEV constexpr i128 mk128(")
    a_end = src.index("\n", src.index("CDEV u128 uabs128(")) + 1
    b = src.index("struct i256 {")
    b_end = src.index("CDEV u32 rotl32(u32 x, int r)")
    shim = """
#include <stdint.h>
#include <string.h>
typedef long long i64; typedef unsigned long long u64; typedef int i32; typedef unsigned int u32; typedef short i16; typedef signed char i8;
typedef unsigned char u8; typedef __int128 i128; typedef unsigned __int128 u128;
#define CDEV static inline
""" + src[a:a_end] + src[b:b_end] + """
static i128 mk(u64 lo, u64 hi) { return (i128)(((u128)hi << 64) | lo); }
extern "C" {
// decimal_div / decimal_integral_div: returns 1 on a zero divisor; the quotient comes back as two words
int t_dec_div(u64 llo, u64 lhi, u64 rlo, u64 rhi, u64 lmlo, u64 lmhi, u64 rmlo, u64 rmhi, int integral, u64* out) {
  bool dz = false;
  i128 q = dec_div(mk(llo, lhi), mk(rlo, rhi), (u128)mk(lmlo, lmhi), (u128)mk(rmlo, rmhi), dz, integral != 0);
  out[0] = (u64)(u128)q; out[1] = (u64)((u128)q >> 64);
  return dz ? 1 : 0;
}
// the wide multiply path: a*b, HALF_UP division by 10^k (k may be 0 → divisor 1 skipped by the caller), bound check; returns fits
int t_wide_mul(u64 alo, u64 ahi, u64 blo, u64 bhi, u64 dlo, u64 dhi, u64 bndlo, u64 bndhi, u64* out) {
  i256 raw = i128_mul_i128(mk(alo, ahi), mk(blo, bhi));
  const u128 d = (u128)mk(dlo, dhi);
  if (d > 1) raw = i256_div_pow10_half_up(raw, d);
  i128 v;
  const bool fits = i256_fits_bound(raw, (u128)mk(bndlo, bndhi), v);
  out[0] = (u64)(u128)v; out[1] = (u64)((u128)v >> 64);
  return fits ? 1 : 0;
}
}
"""
    c = tmp_path / "dev_dec.cpp"
    c.write_text(shim)
    so = tmp_path / "libdevdec.so"
    subprocess.check_call(["g++", "-O1", "-fPIC", "-shared", "-Wno-unused-function", "-o", str(so), str(c)])
    m = ctypes.CDLL(str(so))
    u64 = ctypes.c_uint64
    m.t_dec_div.restype, m.t_dec_div.argtypes = ctypes.c_int, [u64] * 8 + [ctypes.c_int, ctypes.POINTER(u64)]
    m.t_wide_mul.restype, m.t_wide_mul.argtypes = ctypes.c_int, [u64] * 8 + [ctypes.POINTER(u64)]
    return m


def _w(v):   # two's-complement 128-bit value as (lo, hi) words
    v &= (1 << 128) - 1
    return v & (2**64 - 1), v >> 64


def _from_words(out):
    v = out[0] | (out[1] << 64)
    return v - (1 << 128) if v >> 127 else v


def test_device_decimal_source_on_host(tmp_path):
    """dec_div (decimal_div / decimal_integral_div, div.rs:40-165) and the 256-bit multiply → HALF_UP rescale → bound check chain
    (wide_decimal_binary_expr.rs:179-350) from the device header, compiled for the host, against exact Python integers."""
    import random
    m = _build_decimal(tmp_path)
    rnd = random.Random(8)
    out = (ctypes.c_uint64 * 2)()
    trunc = lambda a, b: abs(a) // abs(b) * (-1 if (a < 0) != (b < 0) else 1)
    for _ in range(4000):
        p1, p2 = rnd.randint(1, 38), rnd.randint(1, 38)
        l = rnd.randint(-(10**p1 - 1), 10**p1 - 1)
        r = rnd.choice([0, rnd.randint(-(10**p2 - 1), 10**p2 - 1)]) if rnd.random() < 0.05 else rnd.randint(-(10**p2 - 1), 10**p2 - 1)
        l_exp = rnd.randint(0, min(38, 76 - p1))
        r_exp = rnd.randint(0, 38 - p2)
        integral = rnd.random() < 0.5
        dz = m.t_dec_div(*_w(l), *_w(r), *_w(10**l_exp), *_w(10**r_exp), 1 if integral else 0, out)
        if r == 0:
            assert dz == 1
            continue
        assert dz == 0
        div = trunc(l * 10**l_exp, r * 10**r_exp)
        q = div if integral else (div - 5 if div < 0 else div + 5)
        q = trunc(q, 10)
        want = q if -2**127 <= q < 2**127 else 2**127 - 1            # to_i128().unwrap_or(i128::MAX)
        assert _from_words(out) == want, (l, r, l_exp, r_exp, integral)
    for _ in range(4000):
        a = rnd.randint(-(10**38 - 1), 10**38 - 1) if rnd.random() < 0.7 else rnd.randint(-10**9, 10**9)
        b = rnd.randint(-(10**38 - 1), 10**38 - 1) if rnd.random() < 0.7 else rnd.randint(-10**9, 10**9)
        k = rnd.randint(0, 38)
        p_out = rnd.randint(1, 38)
        prod = a * b
        if k:
            d = 10**k
            half = d // 2
            scaled = trunc(prod + (half if prod >= 0 else -half), d)      # div_round_half_up, :300-330
        else:
            scaled = prod
        bound = 10**p_out - 1
        fits = m.t_wide_mul(*_w(a), *_w(b), *_w(10**k), *_w(bound), out)
        assert bool(fits) == (abs(scaled) <= bound), (a, b, k, p_out)
        if fits:
            assert _from_words(out) == scaled


# --------------------------------------------------------------------------- exact Float64 sums

def _build_fix(tmp_path):
    src = open(_HDR).read()
    a = src.index("constexpr int kFixW = 158;")
    b = src.index("// SumDecimal overflow is prefix-order dependent in the reference")
    add = src[src.index("CDEV void acc_add192(u64* a, const u64* b) {"):src.index("CDEV void acc_umax128(")]
    l0 = src.index("// Σ_j sext(w[j]) · 2^(43·j) as a 192-bit two's-complement number")
    limbs = src[l0:src.index("\n}\n", l0) + 3]
    shim = """
#include <stdint.h>
#include <string.h>
#include <math.h>
typedef long long i64; typedef unsigned long long u64; typedef int i32; typedef unsigned int u32; typedef short i16; typedef unsigned short u16;
typedef signed char i8; typedef unsigned char u8; typedef __int128 i128; typedef unsigned __int128 u128;
#define CDEV static inline
constexpr int kLimbBits = 43;
static inline i64 __double_as_longlong(double d) { i64 x; memcpy(&x, &d, 8); return x; }
static inline double __longlong_as_double(i64 v) { double x; memcpy(&x, &v, 8); return x; }
""" + add + src[a:b] + limbs + """
extern "C" {
// sum n doubles at scale s: through the 192-bit feeder (mode 0) or through the four 43-bit limbs of the grouped path (mode 1)
double t_fix_sum(const double* x, i64 n, int s, int mode, u64* exp_out) {
  u64 acc[3] = {0, 0, 0}, limb[4] = {0, 0, 0, 0}, cls = 0, hi = 0, lo = 0;
  for (i64 i = 0; i < n; i++) {
    if (mode == 0) acc_feed_fix192(acc, x[i], s);
    else for (int j = 0; j < 4; j++) limb[j] += f64_fix_limb(x[i], s, j);
    cls |= f64_class(x[i]);
    u64 h = f64_exp_hi(x[i]), l = f64_exp_lo(x[i]);
    if (h > hi) hi = h;
    if (l > lo) lo = l;
  }
  if (mode == 1) limbs_to_i192(limb, 4, acc);
  exp_out[0] = hi; exp_out[1] = lo;
  return fix192_to_f64(acc, s, cls);
}
int t_fix_scale(i64 packed, int f) { return fix_scale(packed, f); }
}
"""
    c = tmp_path / "dev_fix.cpp"
    c.write_text(shim)
    so = tmp_path / "libdevfix.so"
    subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-function", "-o", str(so), str(c)])
    m = ctypes.CDLL(str(so))
    m.t_fix_sum.restype = ctypes.c_double
    m.t_fix_sum.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    m.t_fix_scale.restype = ctypes.c_int
    m.t_fix_scale.argtypes = [ctypes.c_int64, ctypes.c_int]
    return m


def _window(xs):
    """(low, top): 2^low = lowest set bit of any addend, every |x| < 2^top — what the executor derives from the tracked exponent words."""
    import math
    lo, top = None, None
    for x in xs:
        if x == 0 or math.isinf(x) or math.isnan(x):
            continue
        m, e = math.frexp(abs(x))                 # |x| = m · 2^e, 0.5 ≤ m < 1
        mant = int(m * (1 << 53))                 # exact
        low = e - 53 + ((mant & -mant).bit_length() - 1)
        lo = low if lo is None else min(lo, low)
        top = e if top is None else max(top, e)
    return lo, top


def test_exact_float_sum_device_source_on_host(tmp_path):
    """The fixed-point Float64 sum the GPU pipelines use, run on the host from the same source text: with every addend inside the window the
    result is the correctly rounded exact sum — Python's math.fsum — bit for bit, through both feeders, whatever the magnitudes and signs."""
    import math
    m = _build_fix(tmp_path)
    rng = np.random.default_rng(17)
    exp_out = (ctypes.c_uint64 * 2)()

    def run(xs, s, mode):
        a = np.ascontiguousarray(np.asarray(xs, dtype=np.float64))
        return m.t_fix_sum(a.ctypes.data, len(a), s, mode, exp_out)

    cases = [
        rng.standard_normal(5000) * 1e6,
        rng.random(3000),
        np.concatenate([rng.standard_normal(2000) * 1e15, -rng.standard_normal(2000) * 1e-10]),      # 83 binary orders of magnitude
        np.array([1e16, 1.0, -1e16, 1.0, 3.0, 1e-3]),                                                 # sequential f64 addition loses the 1.0s
        np.array([0.1] * 10),
        np.array([5e-324, 5e-324, -5e-324, 2.2250738585072014e-308]),                                 # subnormals
        np.array([1.7976931348623157e308, -1.7976931348623157e308, 1e292]),
        np.array([-0.0, 0.0]),
        rng.integers(-2**40, 2**40, 4000).astype(np.float64) / 1024.0,
        np.array([2.0**-60, 2.0**40, -2.0**40]),
    ]
    for xs in cases:
        want = math.fsum(xs.tolist())
        lo, top = _window(xs.tolist())
        s = -94 if lo is None else lo
        assert lo is None or top - lo <= 158
        for mode in (0, 1):
            got = run(xs, s, mode)
            assert got == want and math.copysign(1, got) == math.copysign(1, want + 0.0), (xs[:4], mode, got, want)
        if lo is not None:
            assert (int(exp_out[0]) - 1200, 1200 - int(exp_out[1])) == (top, lo)
    # any scale at or below the lowest set bit gives the same bits; results never depend on the order of the addends
    xs = rng.standard_normal(1000) * 1e3
    lo, top = _window(xs.tolist())
    base = run(xs, lo, 0)
    assert base == math.fsum(xs.tolist()) == run(xs, lo - 20, 1) == run(rng.permutation(xs), lo - 7, 0)
    # a scale above the lowest bits truncates toward zero, by less than rows · 2^s
    s_hi = lo + 30
    assert abs(run(xs, s_hi, 0) - base) <= len(xs) * 2.0 ** s_hi
    # IEEE outcome of non-finite addends
    inf, nan = float("inf"), float("nan")
    assert run([1.0, inf, 2.0], -94, 0) == inf and run([1.0, -inf], -94, 1) == -inf
    assert math.isnan(run([inf, -inf, 1.0], -94, 0)) and math.isnan(run([1.0, nan], -94, 1))
    # packed scales round-trip through the kernel argument word
    packed = sum(((v & 0xFFFF) << (16 * i)) for i, v in enumerate([-94, 100, -1300, 0]))
    if packed >= 1 << 63:
        packed -= 1 << 64
    assert [m.t_fix_scale(packed, i) for i in range(4)] == [-94, 100, -1300, 0]
