// Pooling layers of the encoder on NHWC fp32 activations (HBM-bound, float4 = 4 channels per lane, no atomics).
//
//   mmpool_fwd/bwd     MMPool((1,1)) of the reference (network/res_encoder.py:247-265, called at :49):
//                      y = max_hw(x) * w + mean_hw(x) * (1 - w), w = sigmoid(p), p a learned scalar.  One kernel per
//                      direction instead of adaptive_max_pool2d + adaptive_avg_pool2d + sigmoid + 4 elementwise ops
//                      (the ATen adaptive max pool alone took 143 us per step in round 1's profile).
//   maxpool_fwd/bwd    nn.MaxPool2d(3, 2, 1) after the ResNet stem (torchvision resnet18.maxpool as the reference
//                      builds it, network/res_encoder.py:345-373) and the MaxPool2d(3, 1, 1) / MaxPool2d(2, 2) of the
//                      LightEstimator (network/res_encoder.py:150-210).  The forward stores the winning tap (first
//                      maximum in scan order like ATen) as one byte; the backward GATHERS (every input pixel looks at
//                      the windows that contain it), so dx is written exactly once and needs no zero fill.
//   bias_relu_bwd      backward of the LightEstimator's conv + bias + ReLU epilogue: masked gradient + bias gradient.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "hifihr_internal.h"

namespace hifihr {

// ------------------------------------------------------------------------------------------------
// MMPool: workgroup = (image b, 64 channels); thread = (row lane 0..15, float4 channel lane 0..15)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void take_max(float& m, int& mi, float v, int i) {
  if (v > m || (v == m && i < mi)) { m = v; mi = i; }
}

__global__ __launch_bounds__(256) void mmpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ p, int HW, int C,
                                                        float* __restrict__ y, int* __restrict__ argmax, float* __restrict__ xmax,
                                                        float* __restrict__ xavg) {
  __shared__ float4 s_max[16][16], s_sum[16][16];
  __shared__ int4 s_idx[16][16];
  const int b = blockIdx.y, cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cl * 4;
  const bool cok = c < C;
  float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX), s = make_float4(0.f, 0.f, 0.f, 0.f);
  int4 mi = make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);
  if (cok) {
    const float* xb = x + (size_t)b * HW * C + c;
    for (int r = rl; r < HW; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)r * C);
      take_max(m.x, mi.x, v.x, r); take_max(m.y, mi.y, v.y, r); take_max(m.z, mi.z, v.z, r); take_max(m.w, mi.w, v.w, r);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  s_max[rl][cl] = m; s_sum[rl][cl] = s; s_idx[rl][cl] = mi;
  __syncthreads();
  if (rl == 0 && cok) {
    for (int r = 1; r < 16; ++r) {
      const float4 om = s_max[r][cl], os = s_sum[r][cl];
      const int4 oi = s_idx[r][cl];
      take_max(m.x, mi.x, om.x, oi.x); take_max(m.y, mi.y, om.y, oi.y); take_max(m.z, mi.z, om.z, oi.z); take_max(m.w, mi.w, om.w, oi.w);
      s.x += os.x; s.y += os.y; s.z += os.z; s.w += os.w;
    }
    const float w = 1.0f / (1.0f + expf(-p[0])), inv = 1.0f / (float)HW;
    const float4 avg = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    const size_t o = (size_t)b * C + c;
    *reinterpret_cast<float4*>(y + o) = make_float4(m.x * w + avg.x * (1.f - w), m.y * w + avg.y * (1.f - w),
                                                     m.z * w + avg.z * (1.f - w), m.w * w + avg.w * (1.f - w));
    *reinterpret_cast<int4*>(argmax + o) = mi;
    *reinterpret_cast<float4*>(xmax + o) = m;
    *reinterpret_cast<float4*>(xavg + o) = avg;
  }
}

// dx = gy * (1 - w) / HW + [row == argmax] * gy * w;   dp_acc += sum gy * (max - avg) * w * (1 - w)   (block 0 alone)
__global__ __launch_bounds__(256) void mmpool_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ p,
                                                        const int* __restrict__ argmax, const float* __restrict__ xmax,
                                                        const float* __restrict__ xavg, int B, int HW, int C,
                                                        float* __restrict__ dx, float* __restrict__ dp_acc) {
  const int b = blockIdx.y, cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cl * 4;
  const float w = 1.0f / (1.0f + expf(-p[0]));
  if (c < C) {
    const size_t o = (size_t)b * C + c;
    const float4 g = *reinterpret_cast<const float4*>(gy + o);
    const int4 mi = *reinterpret_cast<const int4*>(argmax + o);
    const float ka = (1.f - w) / (float)HW;
    float* xb = dx + (size_t)b * HW * C + c;
    for (int r = rl; r < HW; r += 16) {
      float4 v = make_float4(g.x * ka, g.y * ka, g.z * ka, g.w * ka);
      if (r == mi.x) v.x += g.x * w;
      if (r == mi.y) v.y += g.y * w;
      if (r == mi.z) v.z += g.z * w;
      if (r == mi.w) v.w += g.w * w;
      *reinterpret_cast<float4*>(xb + (size_t)r * C) = v;
    }
  }
  // d/dp = w (1 - w) sum_{b,c} gy (xmax - xavg): the first workgroup of every image folds that image's C values (one float4 per
  // lane, shuffle tree + 4 partials through LDS) and adds one float: B atomics.  (A single workgroup walking all B*C values was
  // a 64-step latency chain = most of this kernel's time while the other workgroups had long finished.)
  if (dp_acc != nullptr && blockIdx.x == 0) {
    __shared__ float red[4];
    float a = 0.f;
    for (int c4 = threadIdx.x; c4 * 4 < C; c4 += 256) {
      const size_t o = (size_t)b * C + c4 * 4;
      const float4 g = *reinterpret_cast<const float4*>(gy + o);
      const float4 xm = *reinterpret_cast<const float4*>(xmax + o), xa = *reinterpret_cast<const float4*>(xavg + o);
      a += g.x * (xm.x - xa.x) + g.y * (xm.y - xa.y) + g.z * (xm.z - xa.z) + g.w * (xm.w - xa.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dp_acc, (red[0] + red[1] + red[2] + red[3]) * w * (1.f - w));
  }
}

hipError_t launch_mmpool_fwd(const float* x, const float* p, int B, int HW, int C, float* y, int* argmax, float* xmax, float* xavg,
                             hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mmpool_fwd_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, x, p, HW, C, y, argmax, xmax, xavg);
  return hipGetLastError();
}

hipError_t launch_mmpool_bwd(const float* gy, const float* p, const int* argmax, const float* xmax, const float* xavg, int B, int HW,
                             int C, float* dx, float* dp_acc, hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mmpool_bwd_kernel, dim3((C + 63) / 64, B), dim3(256), 0, st, gy, p, argmax, xmax, xavg, B, HW, C, dx, dp_acc);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(K, S, P): thread = (output pixel, 4 channels)
// ------------------------------------------------------------------------------------------------
// FLAT (round 5): y is written in NCHW order, i.e. as the [N][C * OH * OW] matrix that `x.view(B, -1)` of the reference makes of the pooled
// tensor (network/res_encoder.py:199-201: the light estimator's last pool feeds a Linear) -- a reshape of the channels-last tensor is a copy
// kernel forward and another one for the gradient; the tap indices stay in the kernels' own [N][OH][OW][C] order.
template <int K, int S, int P, bool FLAT = false>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C, int OH, int OW,
                                                         float* __restrict__ y, unsigned char* __restrict__ tap) {
  const int C4 = C / 4;
  const size_t total = (size_t)N * OH * OW * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % C4);
    size_t rest = i / C4;
    const int ow = (int)(rest % OW); rest /= OW;
    const int oh = (int)(rest % OH);
    const int n = (int)(rest / OH);
    float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    int4 mt = make_int4(0, 0, 0, 0);
    bool first = true;
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const int ih = oh * S - P + r;
#pragma unroll
      for (int s = 0; s < K; ++s) {
        const int iw = ow * S - P + s;
        if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * H + ih) * W + iw) * C + cg * 4);
        const int t = r * K + s;
        // ATen: (val > maxval) || isnan(val), starting from the first in-range tap
        if (first || v.x > m.x || v.x != v.x) { m.x = v.x; mt.x = t; }
        if (first || v.y > m.y || v.y != v.y) { m.y = v.y; mt.y = t; }
        if (first || v.z > m.z || v.z != v.z) { m.z = v.z; mt.z = t; }
        if (first || v.w > m.w || v.w != v.w) { m.w = v.w; mt.w = t; }
        first = false;
      }
    }
    if (FLAT) {
      float* o = y + (((size_t)n * C + cg * 4) * OH + oh) * OW + ow;
      const size_t pl = (size_t)OH * OW;
      o[0] = m.x; o[pl] = m.y; o[2 * pl] = m.z; o[3 * pl] = m.w;
    } else {
      *reinterpret_cast<float4*>(y + i * 4) = m;
    }
    *reinterpret_cast<uchar4*>(tap + i * 4) = make_uchar4((unsigned char)mt.x, (unsigned char)mt.y, (unsigned char)mt.z, (unsigned char)mt.w);
  }
}

// thread = (input pixel, 4 channels): sum gy over the windows whose winning tap is this pixel
// ymask (round 4): the pool's own OUTPUT when its input was a ReLU's output -- a window's gradient then passes only where the pooled value
// is positive (the winning tap holds the window's maximum, and relu'(z) = [relu(z) > 0] there; the other taps get nothing anyway), i.e.
// the ReLU's backward without a pass over the four-times larger pre-pool tensors
template <int K, int S, int P, bool FLAT = false>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ tap,
                                                         const float* __restrict__ ymask, int N, int H, int W, int C, int OH, int OW,
                                                         float* __restrict__ dx) {
  const int C4 = C / 4;
  const size_t total = (size_t)N * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % C4);
    size_t rest = i / C4;
    const int iw = (int)(rest % W); rest /= W;
    const int ih = (int)(rest % H);
    const int n = (int)(rest / H);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    // windows oh with oh * S - P + r == ih: r = (ih + P) % S + j * S, j = 0 .. ceil(K / S) - 1.  All candidate loads are issued
    // unconditionally (clamped addresses) before any is used: loads behind a branch serialise into one HBM latency each.
    constexpr int NR = (K + S - 1) / S;
    int ohc[NR], owc[NR], rc[NR], sc[NR];
    bool vh[NR], vw[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      rc[j] = (ih + P) % S + j * S;
      const int th = ih + P - rc[j];
      vh[j] = rc[j] < K && th >= 0 && th / S < OH;
      ohc[j] = vh[j] ? th / S : 0;
      sc[j] = (iw + P) % S + j * S;
      const int tw = iw + P - sc[j];
      vw[j] = sc[j] < K && tw >= 0 && tw / S < OW;
      owc[j] = vw[j] ? tw / S : 0;
    }
    uchar4 t[NR][NR];
    float4 g[NR][NR];
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const size_t o = ((((size_t)n * OH + ohc[j]) * OW + owc[k]) * C4 + cg) * 4;
        t[j][k] = *reinterpret_cast<const uchar4*>(tap + o);
        if (FLAT) {                                         // gy in NCHW order (the gradient of the flattened matrix)
          const float* q = gy + (((size_t)n * C + cg * 4) * OH + ohc[j]) * OW + owc[k];
          const size_t pl = (size_t)OH * OW;
          g[j][k] = make_float4(q[0], q[pl], q[2 * pl], q[3 * pl]);
        } else {
          g[j][k] = *reinterpret_cast<const float4*>(gy + o);
        }
        if (ymask != nullptr) {                             // (uniform)
          const float4 m = *reinterpret_cast<const float4*>(ymask + o);
          g[j][k].x = m.x > 0.f ? g[j][k].x : 0.f; g[j][k].y = m.y > 0.f ? g[j][k].y : 0.f;
          g[j][k].z = m.z > 0.f ? g[j][k].z : 0.f; g[j][k].w = m.w > 0.f ? g[j][k].w : 0.f;
        }
      }
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const bool ok = vh[j] && vw[k];
        const unsigned char me = (unsigned char)(rc[j] * K + sc[k]);
        if (ok && t[j][k].x == me) a.x += g[j][k].x;
        if (ok && t[j][k].y == me) a.y += g[j][k].y;
        if (ok && t[j][k].z == me) a.z += g[j][k].z;
        if (ok && t[j][k].w == me) a.w += g[j][k].w;
      }
    *reinterpret_cast<float4*>(dx + i * 4) = a;
  }
}

static unsigned pool_grid(size_t total) {
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

hipError_t launch_maxpool_flat(const float* x_or_gy, unsigned char* tap, int N, int H, int W, int C, int k, int s, int p, float* y_or_dx, int bwd,
                               hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  const int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
  if (!bwd) {
    const dim3 grid(pool_grid((size_t)N * OH * OW * (C / 4)));
    if (k == 3 && s == 2 && p == 1) hipLaunchKernelGGL((maxpool_fwd_kernel<3, 2, 1, true>), grid, dim3(256), 0, st, x_or_gy, N, H, W, C, OH, OW, y_or_dx, tap);
    else if (k == 3 && s == 1 && p == 1) hipLaunchKernelGGL((maxpool_fwd_kernel<3, 1, 1, true>), grid, dim3(256), 0, st, x_or_gy, N, H, W, C, OH, OW, y_or_dx, tap);
    else if (k == 2 && s == 2 && p == 0) hipLaunchKernelGGL((maxpool_fwd_kernel<2, 2, 0, true>), grid, dim3(256), 0, st, x_or_gy, N, H, W, C, OH, OW, y_or_dx, tap);
    else return hipErrorInvalidValue;
  } else {
    const dim3 grid(pool_grid((size_t)N * H * W * (C / 4)));
    const float* none = nullptr;
    if (k == 3 && s == 2 && p == 1) hipLaunchKernelGGL((maxpool_bwd_kernel<3, 2, 1, true>), grid, dim3(256), 0, st, x_or_gy, tap, none, N, H, W, C, OH, OW, y_or_dx);
    else if (k == 3 && s == 1 && p == 1) hipLaunchKernelGGL((maxpool_bwd_kernel<3, 1, 1, true>), grid, dim3(256), 0, st, x_or_gy, tap, none, N, H, W, C, OH, OW, y_or_dx);
    else if (k == 2 && s == 2 && p == 0) hipLaunchKernelGGL((maxpool_bwd_kernel<2, 2, 0, true>), grid, dim3(256), 0, st, x_or_gy, tap, none, N, H, W, C, OH, OW, y_or_dx);
    else return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_maxpool_fwd(const float* x, int N, int H, int W, int C, int k, int s, int p, float* y, unsigned char* tap,
                              hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  const int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
  const dim3 grid(pool_grid((size_t)N * OH * OW * (C / 4)));
  if (k == 3 && s == 2 && p == 1) hipLaunchKernelGGL((maxpool_fwd_kernel<3, 2, 1>), grid, dim3(256), 0, st, x, N, H, W, C, OH, OW, y, tap);
  else if (k == 3 && s == 1 && p == 1) hipLaunchKernelGGL((maxpool_fwd_kernel<3, 1, 1>), grid, dim3(256), 0, st, x, N, H, W, C, OH, OW, y, tap);
  else if (k == 2 && s == 2 && p == 0) hipLaunchKernelGGL((maxpool_fwd_kernel<2, 2, 0>), grid, dim3(256), 0, st, x, N, H, W, C, OH, OW, y, tap);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_maxpool_bwd(const float* gy, const unsigned char* tap, const float* ymask, int N, int H, int W, int C, int k, int s, int p, float* dx,
                              hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  const int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
  const dim3 grid(pool_grid((size_t)N * H * W * (C / 4)));
  if (k == 3 && s == 2 && p == 1) hipLaunchKernelGGL((maxpool_bwd_kernel<3, 2, 1>), grid, dim3(256), 0, st, gy, tap, ymask, N, H, W, C, OH, OW, dx);
  else if (k == 3 && s == 1 && p == 1) hipLaunchKernelGGL((maxpool_bwd_kernel<3, 1, 1>), grid, dim3(256), 0, st, gy, tap, ymask, N, H, W, C, OH, OW, dx);
  else if (k == 2 && s == 2 && p == 0) hipLaunchKernelGGL((maxpool_bwd_kernel<2, 2, 0>), grid, dim3(256), 0, st, gy, tap, ymask, N, H, W, C, OH, OW, dx);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// conv + bias + ReLU backward: g = dy * (y > 0), db_acc[c] += sum_rows g      (tiny tensors: LightEstimator)
// workgroup = 64 rows x all channels; thread = (row lane, float4 channel group); per-channel sums through LDS + atomics
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, long M, int C,
                                                           float* __restrict__ g, float* __restrict__ db_acc) {
  __shared__ float sums[256];
  const int C4 = C / 4;                              // C4 <= 64 (checked by the launcher): RL = 256 / C4 rows per pass
  const int RL = 256 / C4;
  const int cg = threadIdx.x % C4, rl = threadIdx.x / C4;
  for (int c = threadIdx.x; c < C; c += 256) sums[c] = 0.f;
  __syncthreads();
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < RL) {
    for (long m = (long)blockIdx.x * RL + rl; m < M; m += (long)gridDim.x * RL) {
      const size_t o = (size_t)m * C + cg * 4;
      float4 v = *reinterpret_cast<const float4*>(dy + o);
      const float4 yy = *reinterpret_cast<const float4*>(y + o);
      v.x = yy.x > 0.f ? v.x : 0.f; v.y = yy.y > 0.f ? v.y : 0.f; v.z = yy.z > 0.f ? v.z : 0.f; v.w = yy.w > 0.f ? v.w : 0.f;
      *reinterpret_cast<float4*>(g + o) = v;
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    atomicAdd(&sums[cg * 4], a.x); atomicAdd(&sums[cg * 4 + 1], a.y); atomicAdd(&sums[cg * 4 + 2], a.z); atomicAdd(&sums[cg * 4 + 3], a.w);
  }
  __syncthreads();
  if (db_acc != nullptr)
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(db_acc + c, sums[c]);
}

hipError_t launch_bias_relu_bwd(const float* dy, const float* y, long M, int C, float* g, float* db_acc, hipStream_t st) {
  if (C % 4 != 0 || C > 256) return hipErrorInvalidValue;
  const int RL = 256 / (C / 4);
  long blocks = (M + RL - 1) / RL;
  // with a bias gradient: few workgroups so the per-channel atomics stay uncontended (LightEstimator, tiny tensors);
  // without one (frozen VGG19 of the perceptual loss, hundreds of MB per layer): a plain streaming pass over the whole chip
  const long cap = db_acc != nullptr ? 64 : 8192;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(bias_relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, y, M, C, g, db_acc);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// LightEstimator's output split (reference network/res_encoder.py:205-210): lights[B][6] -> colors = hardtanh(lights[:, :3]) and
// directions = lights[:, 3:] as two contiguous [B][3] tensors, one launch each way (as separate ATen calls: clamp + copy forward,
// hardtanh_backward + cat backward).  nn.Hardtanh: clamp to [-1, 1], gradient 1 strictly inside and 0 at / beyond the ends.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void light_split_fwd_kernel(const float* __restrict__ lights, int B, float* __restrict__ colors,
                                                             float* __restrict__ dirs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * 3) return;
  const int b = i / 3, c = i - 3 * b;
  const float x = lights[b * 6 + c];
  colors[i] = x != x ? x : fminf(fmaxf(x, -1.f), 1.f);      // (a NaN colour stays NaN, as torch's hardtanh leaves it)
  dirs[i] = lights[b * 6 + 3 + c];
}
__global__ __launch_bounds__(256) void light_split_bwd_kernel(const float* __restrict__ lights, const float* __restrict__ gcolors,
                                                             const float* __restrict__ gdirs, int B, float* __restrict__ glights) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * 3) return;
  const int b = i / 3, c = i - 3 * b;
  const float x = lights[b * 6 + c];
  glights[b * 6 + c] = (gcolors != nullptr && !(x <= -1.f || x >= 1.f)) ? gcolors[i] : 0.f;   // hardtanh_backward's own predicate (NaN passes)
  glights[b * 6 + 3 + c] = gdirs != nullptr ? gdirs[i] : 0.f;
}
hipError_t launch_light_split_fwd(const float* lights, int B, float* colors, float* dirs, hipStream_t st) {
  hipLaunchKernelGGL(light_split_fwd_kernel, dim3((B * 3 + 255) / 256), dim3(256), 0, st, lights, B, colors, dirs);
  return hipGetLastError();
}
hipError_t launch_light_split_bwd(const float* lights, const float* gcolors, const float* gdirs, int B, float* glights, hipStream_t st) {
  hipLaunchKernelGGL(light_split_bwd_kernel, dim3((B * 3 + 255) / 256), dim3(256), 0, st, lights, gcolors, gdirs, B, glights);
  return hipGetLastError();
}

}  // namespace hifihr
