// TEST INFRASTRUCTURE: what a generated pipeline's source (codegen.cpp) and csrc/device/comet_device.hpp need in order to compile for the HOST, so that the
// per-row code the generator wrote — P::keep_tile / P::emit_tile / P::emit and every CDEV helper they call — can be run on the CPU against the oracle
// (tests/test_codegen_emu_cpu.py).  The kernel BODIES (wave ballots, LDS tables, ordered compaction) are not emulated: they only have to compile here; the
// driver below walks the rows itself.  Nothing under datafusion-comet_amd/ includes this file.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)
#define __shared__ static
#define __HIPCC_RTC__ 1

struct EmuDim3 { unsigned x = 0, y = 0, z = 0; };
static thread_local EmuDim3 threadIdx, blockIdx;
static EmuDim3 gridDim = {1, 1, 1}, blockDim = {256, 1, 1};

static inline void __syncthreads() {}
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }      // (a wave of ONE lane: bodies compile, the driver does not run them)
template <class T> static inline T __shfl(T v, int, int = 64) { return v; }
template <class T> static inline T __shfl_up(T v, unsigned, int = 64) { return v; }
template <class T> static inline T __shfl_down(T v, unsigned, int = 64) { return v; }
template <class T> static inline T __shfl_xor(T v, int, int = 64) { return v; }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * (unsigned __int128)b) >> 64); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline double __dsqrt_rn(double x) { return std::sqrt(x); }
template <class T> static inline T emu_nt_load(const T* p) { return *p; }
#define __builtin_nontemporal_load(p) emu_nt_load(p)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
static inline void __threadfence() {}
static inline void __threadfence_block() {}
template <class T, class U> static inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> static inline T atomicAnd(T* p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U> static inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U, class V> static inline T atomicCAS(T* p, U c, V v) { T o = *p; if (o == (T)c) *p = (T)v; return o; }
using std::isfinite;
using std::isnan;
using std::isinf;
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __ddiv_rn(double a, double b) { return a / b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline int __syncthreads_or(int p) { return p; }
static inline int __syncthreads_and(int p) { return p; }
static inline int __syncthreads_count(int p) { return p ? 1 : 0; }
enum { __HIP_MEMORY_SCOPE_SINGLETHREAD = 1, __HIP_MEMORY_SCOPE_WAVEFRONT = 2, __HIP_MEMORY_SCOPE_WORKGROUP = 3, __HIP_MEMORY_SCOPE_AGENT = 4, __HIP_MEMORY_SCOPE_SYSTEM = 5 };
template <class T, class U> static inline void __hip_atomic_store(T* p, U v, int, int) { *p = (T)v; }
template <class T> static inline T __hip_atomic_load(const T* p, int, int) { return *p; }
template <class T, class U> static inline T __hip_atomic_fetch_add(T* p, U v, int, int) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T __hip_atomic_fetch_or(T* p, U v, int, int) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> static inline T __hip_atomic_fetch_and(T* p, U v, int, int) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U> static inline T __hip_atomic_fetch_max(T* p, U v, int, int) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T __hip_atomic_fetch_min(T* p, U v, int, int) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T __hip_atomic_exchange(T* p, U v, int, int) { T o = *p; *p = (T)v; return o; }
template <class T, class U> static inline bool __hip_atomic_compare_exchange_strong(T* p, T* expected, U v, int, int, int) {
  if (*p == *expected) { *p = (T)v; return true; }
  *expected = *p;
  return false;
}
