// tests/hostsim: a small HIP execution-model emulator (TEST INFRASTRUCTURE ONLY).
//
// There is no GPU in the build container, so the kernels in hifihr_amd/csrc/*.hip would otherwise be
// written blind.  This header stands in for <hip/hip_runtime.h> when those SAME source files are compiled
// with g++ (see tests/hostsim/Makefile): every workgroup runs as a set of cooperatively scheduled fibers
// (one per work-item) on one OS thread, __syncthreads() and the 64-wide wave primitives are real
// rendezvous points, __shared__ is per-worker storage and atomics are real atomics.  It exists to catch
// indexing, barrier and gradient-formula bugs before GPU time is spent; it is slow, it is never loaded by
// the hifihr_amd package, it is not an oracle and it is not a fallback.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define HIFIHR_HOSTSIM 1
#define __global__
#define __device__
#define __host__
#define __constant__ const
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(::hostsim::dyn_smem());

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(4) uchar4 { unsigned char x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{x, y, z, w}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorNotSupported = 801 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace hostsim {
struct Fiber;
struct Ctx {           // what the kernel sees as threadIdx/blockIdx/...
  dim3 tid, bid, bdim, gdim;
};
Ctx& cur();
char* dyn_smem();
void syncthreads();
uint32_t wave_exchange(uint32_t v, int src_lane);                 // value of src_lane (or own if out of range)
unsigned long long wave_ballot(bool p);
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
int lane_id();
void wave_gather2(float a, float b, float* A64, float* B64);      // every lane's (a, b), lane-indexed
void yield_now();                                                 // spin loops on another workgroup's flag: let the OS run it
}  // namespace hostsim

#define threadIdx (::hostsim::cur().tid)
#define blockIdx (::hostsim::cur().bid)
#define blockDim (::hostsim::cur().bdim)
#define gridDim (::hostsim::cur().gdim)
#define warpSize 64

static inline void __syncthreads() { ::hostsim::syncthreads(); }

template <typename T>
static inline T hs_shfl_(T v, int src) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- and 8-byte shuffles only");
  if constexpr (sizeof(T) == 8) {                 // a 64-bit shuffle is two 32-bit ones (as on the device)
    uint32_t u[2];
    std::memcpy(u, &v, 8);
    u[0] = ::hostsim::wave_exchange(u[0], src);
    u[1] = ::hostsim::wave_exchange(u[1], src);
    T r;
    std::memcpy(&r, u, 8);
    return r;
  } else {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    u = ::hostsim::wave_exchange(u, src);
    T r;
    std::memcpy(&r, &u, 4);
    return r;
  }
}
template <typename T> static inline T __shfl(T v, int src, int w = 64) { return hs_shfl_(v, (::hostsim::lane_id() & ~(w - 1)) + (src & (w - 1))); }
template <typename T> static inline T __shfl_down(T v, unsigned d, int w = 64) {
  const int l = ::hostsim::lane_id();
  const int s = ((l & (w - 1)) + (int)d < w) ? l + (int)d : l;
  return hs_shfl_(v, s);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int w = 64) {
  const int l = ::hostsim::lane_id();
  const int s = ((l & (w - 1)) >= (int)d) ? l - (int)d : l;
  return hs_shfl_(v, s);
}
template <typename T> static inline T __shfl_xor(T v, int m, int w = 64) { return hs_shfl_(v, ::hostsim::lane_id() ^ m); }
static inline unsigned long long __ballot(int p) { return ::hostsim::wave_ballot(p != 0); }
static inline int __any(int p) { return __ballot(p) != 0ull; }
static inline int __all(int p) { return ::hostsim::wave_ballot(p == 0) == 0ull; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }

static inline float atomicAdd(float* p, float v) {
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
  float f;
  do {
    std::memcpy(&f, &old, 4);
    f += v;
    std::memcpy(&nw, &f, 4);
  } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  std::memcpy(&f, &old, 4);
  return f;
}
static inline double atomicAdd(double* p, double v) {
  uint64_t* ip = reinterpret_cast<uint64_t*>(p);
  uint64_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), nw;
  double f;
  do {
    std::memcpy(&f, &old, 8);
    f += v;
    std::memcpy(&nw, &f, 8);
  } while (!__atomic_compare_exchange_n(ip, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  std::memcpy(&f, &old, 8);
  return f;
}
static inline double unsafeAtomicAdd(double* p, double v) { return atomicAdd(p, v); }
static inline float atomicExch(float* p, float v) {
  uint32_t nw, old;
  std::memcpy(&nw, &v, 4);
  old = __atomic_exchange_n(reinterpret_cast<uint32_t*>(p), nw, __ATOMIC_RELAXED);
  float f;
  std::memcpy(&f, &old, 4);
  return f;
}
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {
  unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline int atomicCAS(int* p, int expected, int desired) {
  __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return expected;                                   // the value found (== the old `expected` on success)
}
static inline int atomicMax(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
static inline int atomicMin(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}

static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
#define __expf expf
#define __powf powf
static inline float __saturatef(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }

// ---- runtime API subset (host memory plays device memory) ----
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hostsim error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
// 4 "CUs": the balanced convolution schedule (conv.hip) then runs 16 persistent workgroups, so small test shapes split tiles
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 4; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }

// ---- MFMA emulation (exact: the hardware result is a k-ordered fmaf chain, cdna_hip_programming.md section 3) ----
typedef float hs_floatx16 __attribute__((vector_size(64)));
typedef float hs_floatx4 __attribute__((vector_size(16)));
static inline hs_floatx16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hs_floatx16 c, int, int, int) {
  float A[64], B[64];
  ::hostsim::wave_gather2(a, b, A, B);
  const int l = ::hostsim::lane_id();
  const int j = l & 31;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float d = c[r];
    d = fmaf(A[i], B[j], d);                 // k = 0: A[i][0] from lane i, B[0][j] from lane j
    d = fmaf(A[i + 32], B[j + 32], d);       // k = 1
    c[r] = d;
  }
  return c;
}

// v_mfma_f32_16x16x4_f32: lane l supplies A[row l & 15][k l >> 4] and B[k l >> 4][col l & 15]; D register e of lane l is
// D[row 4 (l >> 4) + e][col l & 15]
static inline hs_floatx4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hs_floatx4 c, int, int, int) {
  float A[64], B[64];
  ::hostsim::wave_gather2(a, b, A, B);
  const int l = ::hostsim::lane_id();
  const int col = l & 15;
  for (int e = 0; e < 4; ++e) {
    const int row = 4 * (l >> 4) + e;
    float d = c[e];
    for (int k = 0; k < 4; ++k) d = fmaf(A[16 * k + row], B[16 * k + col], d);
    c[e] = d;
  }
  return c;
}

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
  ::hostsim::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
