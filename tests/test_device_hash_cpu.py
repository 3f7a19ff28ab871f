"""The XXH64 / murmur3 device functions of csrc/device/comet_device.hpp are plain integer arithmetic: their text is compiled for the host
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
