// Depthwise convolution (groups == channels) on NHWC activations, fp32: forward, backward-data, backward-weight.
//
// Replaces the depthwise `Conv2dStaticSamePadding(groups=oup)` of the reference's EfficientNet MBConv blocks (reference
// network/efficientnet_pt/model.py:49-55,80; utils.py:122-145) -- k = 3 or 5, stride 1 or 2, TensorFlow-style
// asymmetric zero padding (top/left given explicitly; bottom/right implied by the output size).  0.2 % of the network's
// FLOPs but, unfused in MIOpen's fp32 NHWC path, >90 % of its time on this GPU (naive kernels, ~13 ms per launch).
// Bandwidth-bound: every lane owns 4 consecutive channels (float4) of one output pixel; the k*k taps of neighbouring
// pixels overlap, so re-reads come from L1/L2.  Weight layout [C][k][k] (= torch's [C,1,k,k]).
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"

namespace hifihr {

__global__ __launch_bounds__(256) void dwconv_fwd_kernel(DwGeom g, const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y) {
  const int C4 = g.C / 4;
  const long total = (long)g.N * g.OH * g.OW * C4;
  const int KK = g.K * g.K;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = (int)(idx % C4);
    long m = idx / C4;
    const int ow = (int)(m % g.OW); m /= g.OW;
    const int oh = (int)(m % g.OH);
    const int n = (int)(m / g.OH);
    const float* wc = w + (size_t)cg * 4 * KK;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < g.K; ++r) {
      const int ih = oh * g.stride - g.pt + r;
      if (ih < 0 || ih >= g.H) continue;
      for (int s = 0; s < g.K; ++s) {
        const int iw = ow * g.stride - g.pl + s;
        if (iw < 0 || iw >= g.W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * g.H + ih) * g.W + iw) * g.C + cg * 4);
        const int t = r * g.K + s;
        acc.x += v.x * wc[t]; acc.y += v.y * wc[KK + t]; acc.z += v.z * wc[2 * KK + t]; acc.w += v.w * wc[3 * KK + t];
      }
    }
    *reinterpret_cast<float4*>(y + idx * 4) = acc;
  }
}

__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(DwGeom g, const float* __restrict__ dy, const float* __restrict__ w,
                                                             float* __restrict__ dx) {
  const int C4 = g.C / 4;
  const long total = (long)g.N * g.H * g.W * C4;
  const int KK = g.K * g.K;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = (int)(idx % C4);
    long m = idx / C4;
    const int iw = (int)(m % g.W); m /= g.W;
    const int ih = (int)(m % g.H);
    const int n = (int)(m / g.H);
    const float* wc = w + (size_t)cg * 4 * KK;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < g.K; ++r) {
      const int th = ih + g.pt - r;
      if (th < 0 || th % g.stride != 0) continue;
      const int oh = th / g.stride;
      if (oh >= g.OH) continue;
      for (int s = 0; s < g.K; ++s) {
        const int tw = iw + g.pl - s;
        if (tw < 0 || tw % g.stride != 0) continue;
        const int ow = tw / g.stride;
        if (ow >= g.OW) continue;
        const float4 v = *reinterpret_cast<const float4*>(dy + (((size_t)n * g.OH + oh) * g.OW + ow) * g.C + cg * 4);
        const int t = r * g.K + s;
        acc.x += v.x * wc[t]; acc.y += v.y * wc[KK + t]; acc.z += v.z * wc[2 * KK + t]; acc.w += v.w * wc[3 * KK + t];
      }
    }
    *reinterpret_cast<float4*>(dx + idx * 4) = acc;
  }
}

// dw[c][r][s] += sum over output pixels of dy * x.  Threads own a channel group and a row lane (as the batch-norm
// kernels do); the k*k partial sums live in registers, row lanes are folded through LDS, one atomic per value per block.
template <int K>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(DwGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dw) {
  constexpr int KK = K * K;
  __shared__ float lds[256 * 4];
  const int C4 = g.C / 4;
  // channel groups are tiled over blockIdx.y in chunks of <= 256 groups
  const int cgs = min(C4 - (int)blockIdx.y * 256, 256);
  const int CT = cgs, RL = 256 / CT;
  const int cl = threadIdx.x % CT, rl = threadIdx.x / CT;
  const bool active = rl < RL;
  const int cg = blockIdx.y * 256 + cl;
  const long M = (long)g.N * g.OH * g.OW;
  float4 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    for (long m = (long)blockIdx.x * RL + rl; m < M; m += (long)gridDim.x * RL) {
      long q = m;
      const int ow = (int)(q % g.OW); q /= g.OW;
      const int oh = (int)(q % g.OH);
      const int n = (int)(q / g.OH);
      const float4 d = *reinterpret_cast<const float4*>(dy + m * g.C + cg * 4);
#pragma unroll
      for (int r = 0; r < K; ++r) {
        const int ih = oh * g.stride - g.pt + r;
#pragma unroll
        for (int s = 0; s < K; ++s) {
          const int iw = ow * g.stride - g.pl + s;
          if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) {
            const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * g.H + ih) * g.W + iw) * g.C + cg * 4);
            float4& a = acc[r * K + s];
            a.x += d.x * v.x; a.y += d.y * v.y; a.z += d.z * v.z; a.w += d.w * v.w;
          }
        }
      }
    }
  }
  // fold the row lanes tap by tap, then one atomic per (channel, tap) per block
  for (int t = 0; t < KK; ++t) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < KK; ++u)
      if (u == t) a = acc[u];
    __syncthreads();
    lds[threadIdx.x * 4 + 0] = a.x; lds[threadIdx.x * 4 + 1] = a.y; lds[threadIdx.x * 4 + 2] = a.z; lds[threadIdx.x * 4 + 3] = a.w;
    __syncthreads();
    if (active && rl == 0) {
      for (int r = 1; r < RL; ++r) {
        const float* p = lds + (r * CT + cl) * 4;
        a.x += p[0]; a.y += p[1]; a.z += p[2]; a.w += p[3];
      }
      float* o = dw + (size_t)cg * 4 * KK + t;
      atomicAdd(o, a.x); atomicAdd(o + KK, a.y); atomicAdd(o + 2 * KK, a.z); atomicAdd(o + 3 * KK, a.w);
    }
  }
}

static unsigned ew_grid(long total) {
  long b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

hipError_t launch_dwconv_fwd(const DwGeom& g, const float* x, const float* w, float* y, hipStream_t st) {
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(ew_grid((long)g.N * g.OH * g.OW * (g.C / 4))), dim3(256), 0, st, g, x, w, y);
  return hipGetLastError();
}
hipError_t launch_dwconv_bwd_data(const DwGeom& g, const float* dy, const float* w, float* dx, hipStream_t st) {
  hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3(ew_grid((long)g.N * g.H * g.W * (g.C / 4))), dim3(256), 0, st, g, dy, w, dx);
  return hipGetLastError();
}
hipError_t launch_dwconv_bwd_weight(const DwGeom& g, const float* x, const float* dy, float* dw, hipStream_t st) {
  const int C4 = g.C / 4;
  const dim3 grid(256, (C4 + 255) / 256);      // <= 256 row slabs: at most 256 atomics land on one address
  if (g.K == 3) hipLaunchKernelGGL(dwconv_bwd_weight_kernel<3>, grid, dim3(256), 0, st, g, x, dy, dw);
  else if (g.K == 5) hipLaunchKernelGGL(dwconv_bwd_weight_kernel<5>, grid, dim3(256), 0, st, g, x, dy, dw);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace hifihr
