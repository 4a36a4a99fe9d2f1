// Squeeze-and-excitation of the EfficientNet MBConv block on NHWC fp32 activations:
//   y = x * sigmoid( W2 swish( W1 mean_hw(x) + b1 ) + b2 )
// Replaces reference network/efficientnet_pt/model.py:82-86 (adaptive_avg_pool2d, _se_reduce, swish, _se_expand, sigmoid,
// broadcast multiply) and its autograd: ~12 ATen / MIOpen launches forward and ~20 backward per block (26 blocks) become
//   forward : se_pool (partial means, atomics) -> two small-batch linear launches (mlp.hip, swish / sigmoid epilogues)
//             -> se_scale
//   backward: se_bwd_gate (dgate[b][c] = sum_hw dy * x) -> the two linear backward pairs -> se_bwd_dx
//             (dx = dy * gate + dmean / HW: the gradient of the pooling branch is folded into the same pass)
// i.e. the activation tensor is read once per forward kernel and three times + written once in the whole backward.
// The reductions split HW over blockIdx.z so that >= ~512 workgroups run even for the early 112x112 layers with few channels;
// partial sums are added with fp32 atomics into a zeroed [B][C] buffer.
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"

namespace hifihr {

// workgroup = (64 channels, image b, HW slice z); thread = (row lane 0..15, float4 channel lane 0..15)
// MODE 0: out[b][c] += scale * sum x;  MODE 1: out[b][c] += sum a * x   (a = dy)
template <int MODE>
__global__ __launch_bounds__(256) void se_reduce_kernel(const float* __restrict__ x, const float* __restrict__ a, int HW, int C, float scale,
                                                       float* __restrict__ out) {
  __shared__ float4 red[16][16];
  const int b = blockIdx.y, cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cl * 4;
  const bool cok = c < C;
  const int per = (HW + gridDim.z - 1) / gridDim.z;
  const int r0 = blockIdx.z * per, r1 = min(HW, r0 + per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cok) {
    const size_t base = (size_t)b * HW * C + c;
    for (int r = r0 + rl; r < r1; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(x + base + (size_t)r * C);
      if (MODE == 0) {
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      } else {
        const float4 d = *reinterpret_cast<const float4*>(a + base + (size_t)r * C);
        s.x = fmaf(d.x, v.x, s.x); s.y = fmaf(d.y, v.y, s.y); s.z = fmaf(d.z, v.z, s.z); s.w = fmaf(d.w, v.w, s.w);
      }
    }
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && cok) {
    for (int r = 1; r < 16; ++r) { const float4 o = red[r][cl]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    float* o = out + (size_t)b * C + c;
    atomicAdd(o, s.x * scale); atomicAdd(o + 1, s.y * scale); atomicAdd(o + 2, s.z * scale); atomicAdd(o + 3, s.w * scale);
  }
}

// y = x * gate[b][c]                       (add == nullptr)
// y = x * gate[b][c] + add[b][c] * ascale  (backward: x = dy, add = dmean, ascale = 1 / HW)
__global__ __launch_bounds__(256) void se_scale_kernel(const float* __restrict__ x, const float* __restrict__ gate, const float* __restrict__ add,
                                                      float ascale, int B, int HW, int C, float* __restrict__ y) {
  const int C4 = C / 4;
  const size_t total = (size_t)B * HW * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % C4);
    const size_t b = i / ((size_t)HW * C4);
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    const float4 g = *reinterpret_cast<const float4*>(gate + b * C + cg * 4);
    float4 r = make_float4(v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w);
    if (add != nullptr) {
      const float4 ad = *reinterpret_cast<const float4*>(add + b * C + cg * 4);
      r.x = fmaf(ad.x, ascale, r.x); r.y = fmaf(ad.y, ascale, r.y); r.z = fmaf(ad.z, ascale, r.z); r.w = fmaf(ad.w, ascale, r.w);
    }
    *reinterpret_cast<float4*>(y + i * 4) = r;
  }
}

static dim3 se_reduce_grid(int B, int HW, int C) {
  const int cb = (C + 63) / 64;
  int z = (512 + B * cb - 1) / (B * cb);
  if (z > HW / 64) z = HW / 64;
  if (z < 1) z = 1;
  return dim3(cb, B, z);
}

hipError_t launch_se_pool(const float* x, int B, int HW, int C, float* mean_zeroed, hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(se_reduce_kernel<0>, se_reduce_grid(B, HW, C), dim3(256), 0, st, x, nullptr, HW, C, 1.0f / (float)HW, mean_zeroed);
  return hipGetLastError();
}

hipError_t launch_se_bwd_gate(const float* dy, const float* x, int B, int HW, int C, float* dgate_zeroed, hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(se_reduce_kernel<1>, se_reduce_grid(B, HW, C), dim3(256), 0, st, x, dy, HW, C, 1.0f, dgate_zeroed);
  return hipGetLastError();
}

hipError_t launch_se_scale(const float* x, const float* gate, const float* add, float ascale, int B, int HW, int C, float* y, hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  size_t blocks = ((size_t)B * HW * (C / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(se_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, gate, add, ascale, B, HW, C, y);
  return hipGetLastError();
}

}  // namespace hifihr
