// Texture-PCA decode: tex[b][n] = mean[n] + sum_k coef[b][k] basis[k][n]   (K ~ 10 components, n = texels x 3), and its backward.
//
// Replaces the texture half of the NIMBLE layer's forward as the reference consumes it (reference models_res_nimble.py:57,133-142:
// `texture_params` [B,10] -> the hand's texture; SURVEY.md section 8 A9 / N4: "texture-PCA decode (10 x tex^2 x 3 GEMV, HBM-bound)").
// The NIMBLE assets are not available, so the basis is whatever the caller provides (models.py: a seeded stand-in basis over the 778
// vertex colours; a 1024^2 x 3 map is the same call with n = 3 145 728).
// HBM-bound: the basis (K n floats) is streamed ONCE per batch tile of 16 images with 16-byte loads, each thread keeps 16 float4
// accumulators, the coefficients sit in LDS.  Algorithmic bytes: 4 n (K + 1 + B).  Backward: dcoef[b][k] = sum_n g[b][n] basis[k][n]
// as per-thread partial dot products, folded per workgroup in LDS, one float atomic per (workgroup, b, k).
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"

namespace hifihr {

constexpr int kTexBT = 16;      // batch tile
constexpr int kTexMaxK = 32;

__global__ __launch_bounds__(256) void texpca_fwd_kernel(const float* __restrict__ coef, const float* __restrict__ basis,
                                                        const float* __restrict__ mean, int B, int K, long n, float* __restrict__ out) {
  __shared__ float cs[kTexBT * kTexMaxK];
  const long n4 = n / 4;
  for (int b0 = 0; b0 < B; b0 += kTexBT) {
    const int nb = min(kTexBT, B - b0);
    __syncthreads();
    for (int e = threadIdx.x; e < kTexBT * K; e += 256) cs[e] = (e / K < nb) ? coef[(size_t)(b0 + e / K) * K + e % K] : 0.f;
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      const float4 m = mean ? *reinterpret_cast<const float4*>(mean + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 acc[kTexBT];
#pragma unroll
      for (int b = 0; b < kTexBT; ++b) acc[b] = m;
      for (int k = 0; k < K; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(basis + (size_t)k * n + 4 * i);
#pragma unroll
        for (int b = 0; b < kTexBT; ++b) {
          const float c = cs[b * K + k];
          acc[b].x = fmaf(c, w.x, acc[b].x); acc[b].y = fmaf(c, w.y, acc[b].y); acc[b].z = fmaf(c, w.z, acc[b].z); acc[b].w = fmaf(c, w.w, acc[b].w);
        }
      }
#pragma unroll
      for (int b = 0; b < kTexBT; ++b)
        if (b < nb) *reinterpret_cast<float4*>(out + (size_t)(b0 + b) * n + 4 * i) = acc[b];
    }
  }
}

// dcoef[B][K] (zero on entry) += sum_n g[b][n] basis[k][n].  grid = (pieces of n, batch tiles): a workgroup takes one piece of n for the
// images of its batch tile in turn, so a small texture (64 x 64 x 3: 12 pieces) still fills the chip -- round 3 walked the whole batch
// serially inside 12 workgroups, 374 us at B = 48 -- and a large one (1024^2 x 3) streams its basis from HBM once per batch tile.
__global__ __launch_bounds__(256) void texpca_bwd_kernel(const float* __restrict__ g, const float* __restrict__ basis, int B, int K, long n,
                                                        float* __restrict__ dcoef) {
  __shared__ float red[4][kTexMaxK];
  const long n4 = n / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bt = (B + (int)gridDim.y - 1) / (int)gridDim.y;
  const int b_lo = (int)blockIdx.y * bt, b_hi = min(B, b_lo + bt);
  for (int b = b_lo; b < b_hi; ++b) {
    float part[kTexMaxK];
#pragma unroll
    for (int k = 0; k < kTexMaxK; ++k) part[k] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      const float4 gv = *reinterpret_cast<const float4*>(g + (size_t)b * n + 4 * i);
#pragma unroll
      for (int k = 0; k < kTexMaxK; ++k) {
        if (k < K) {
          const float4 w = *reinterpret_cast<const float4*>(basis + (size_t)k * n + 4 * i);
          part[k] += gv.x * w.x + gv.y * w.y + gv.z * w.z + gv.w * w.w;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kTexMaxK; ++k) {
      if (k < K) {
        float v = part[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wave][k] = v;
      }
    }
    __syncthreads();
    if (threadIdx.x < K) atomicAdd(dcoef + (size_t)b * K + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
}

hipError_t launch_texpca_fwd(const float* coef, const float* basis, const float* mean, int B, int K, long n, float* out, hipStream_t st) {
  if (K < 1 || K > kTexMaxK || n < 4 || n % 4 != 0 || B < 1) return hipErrorInvalidValue;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(texpca_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, coef, basis, mean, B, K, n, out);
  return hipGetLastError();
}

hipError_t launch_texpca_bwd(const float* g, const float* basis, int B, int K, long n, float* dcoef_zeroed, hipStream_t st) {
  if (K < 1 || K > kTexMaxK || n < 4 || n % 4 != 0 || B < 1) return hipErrorInvalidValue;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 512) blocks = 512;           // (workgroups x B x K) float atomics at the end: keep them few
  // batch tiles: one image per workgroup while pieces x B stays a few waves of the chip (the basis of such a texture sits in L2 / the
  // Infinity Cache); larger textures take tiles of images so that the basis crosses the fabric B / tile times
  long tiles = B;
  while (tiles > 1 && blocks * tiles > 4096) tiles = (tiles + 1) / 2;
  hipLaunchKernelGGL(texpca_bwd_kernel, dim3((unsigned)blocks, (unsigned)tiles), dim3(256), 0, st, g, basis, B, K, n, dcoef_zeroed);
  return hipGetLastError();
}

}  // namespace hifihr
