// Train-mode batch normalisation fused with the residual add and ReLU, NHWC (x[M][C], M = N*H*W), fp32.
//
// Replaces, per BasicBlock of the reference's ResNet trunk (reference network/res_encoder.py:364-373, the vendored
// torchvision BasicBlock): nn.BatchNorm2d in training mode (batch statistics, running-stat update with momentum
// 0.1, unbiased running variance) + `out += identity` + ReLU and their autograd -- in ATen/MIOpen 3 + 1 + 1 forward
// and 3 + 1 + 1 backward launches with a full HBM round trip each -- with
//   forward : per-channel sum / sum-of-squares come out of the convolution epilogue (conv.hip), then ONE apply kernel
//   backward: ONE reduction kernel (sum g, sum g*xhat with the ReLU mask applied on the fly) + ONE apply kernel that
//             writes dx (and the masked gradient for the identity branch) and accumulates dgamma / dbeta straight
//             into the flat gradient buffer.
// All kernels are HBM-bound: every lane moves float4 (4 consecutive channels), a thread keeps the same channel group
// for its whole grid-stride loop so mean / invstd / gamma / beta live in registers.
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"

namespace hifihr {

// standalone statistics (used when the producer is not one of our convolutions): stats[kStatSlots][2][C] += sums
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long M, int C, float* __restrict__ stats) {
  __shared__ float4 red[2][256];
  const int C4 = C / 4;
  const int cg = threadIdx.x % C4;            // channel group of this thread
  const int rl = threadIdx.x / C4, RL = 256 / C4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long m = (long)blockIdx.x * RL + rl; m < M; m += (long)gridDim.x * RL) {
    const float4 v = *reinterpret_cast<const float4*>(x + m * C + cg * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
  }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
  __syncthreads();
  if (rl == 0) {
    for (int r = 1; r < RL; ++r) {
      const float4 a = red[0][r * C4 + cg], b = red[1][r * C4 + cg];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
    }
    float* ps = stats + (size_t)(blockIdx.x & (kStatSlots - 1)) * 2 * C + cg * 4;
    float* pq = ps + C;
    atomicAdd(ps, s.x); atomicAdd(ps + 1, s.y); atomicAdd(ps + 2, s.z); atomicAdd(ps + 3, s.w);
    atomicAdd(pq, q.x); atomicAdd(pq + 1, q.y); atomicAdd(pq + 2, q.z); atomicAdd(pq + 3, q.w);
  }
}

// sum of the kStatSlots partial copies of entry `idx` of a [kStatSlots][2][C] buffer
__device__ __forceinline__ float slot_sum(const float* __restrict__ buf, int C, int idx) {
  float a = 0.f;
#pragma unroll 8
  for (int sl = 0; sl < kStatSlots; ++sl) a += buf[(size_t)sl * 2 * C + idx];
  return a;
}

// one thread per channel: fold the slot partials into mean / invstd, update the running statistics
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float* __restrict__ stats, long M, int C, float eps, float momentum,
                                                             float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float invM = 1.0f / (float)M;
  const float mu = slot_sum(stats, C, c) * invM;
  float var = slot_sum(stats, C, C + c) * invM - mu * mu;   // biased batch variance
  var = fmaxf(var, 0.f);
  save_mean[c] = mu;
  save_invstd[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    const float unbiased = (M > 1) ? var * ((float)M / (float)(M - 1)) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ save_mean,
                                                        const float* __restrict__ save_invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ residual,
                                                        int relu, long M, int C, float* __restrict__ y) {
  const int C4 = C / 4;
  const int cg = threadIdx.x % C4;
  const int rl = threadIdx.x / C4, RL = 256 / C4;
  float scale[4], shift[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = cg * 4 + k;
    scale[k] = save_invstd[c] * gamma[c];
    shift[k] = beta[c] - save_mean[c] * scale[k];
  }
  for (long m = (long)blockIdx.x * RL + rl; m < M; m += (long)gridDim.x * RL) {
    const size_t o = (size_t)m * C + cg * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    float4 r = make_float4(v.x * scale[0] + shift[0], v.y * scale[1] + shift[1], v.z * scale[2] + shift[2], v.w * scale[3] + shift[3]);
    if (residual) {
      const float4 a = *reinterpret_cast<const float4*>(residual + o);
      r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
    }
    if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
    *reinterpret_cast<float4*>(y + o) = r;
  }
}

// red[kStatSlots][2][C] += (sum g, sum g * xhat), g = dy * (y > 0 if relu)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ x, const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, int relu, long M, int C,
                                                           float* __restrict__ red) {
  __shared__ float4 lds[2][256];
  const int C4 = C / 4;
  const int cg = threadIdx.x % C4;
  const int rl = threadIdx.x / C4, RL = 256 / C4;
  const float4 mu = *reinterpret_cast<const float4*>(save_mean + cg * 4);
  const float4 is = *reinterpret_cast<const float4*>(save_invstd + cg * 4);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long m = (long)blockIdx.x * RL + rl; m < M; m += (long)gridDim.x * RL) {
    const size_t o = (size_t)m * C + cg * 4;
    float4 g = *reinterpret_cast<const float4*>(dy + o);
    if (relu) {
      const float4 yy = *reinterpret_cast<const float4*>(y + o);
      g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f; g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
    }
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
    q.x += g.x * ((v.x - mu.x) * is.x); q.y += g.y * ((v.y - mu.y) * is.y);
    q.z += g.z * ((v.z - mu.z) * is.z); q.w += g.w * ((v.w - mu.w) * is.w);
  }
  lds[0][threadIdx.x] = s; lds[1][threadIdx.x] = q;
  __syncthreads();
  if (rl == 0) {
    for (int r = 1; r < RL; ++r) {
      const float4 a = lds[0][r * C4 + cg], b = lds[1][r * C4 + cg];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
    }
    float* ps = red + (size_t)(blockIdx.x & (kStatSlots - 1)) * 2 * C + cg * 4;
    float* pq = ps + C;
    atomicAdd(ps, s.x); atomicAdd(ps + 1, s.y); atomicAdd(ps + 2, s.z); atomicAdd(ps + 3, s.w);
    atomicAdd(pq, q.x); atomicAdd(pq + 1, q.y); atomicAdd(pq + 2, q.z); atomicAdd(pq + 3, q.w);
  }
}

// one thread per channel: fold the slot partials of the backward reduction into tot[2][C]; dgamma / dbeta accumulate
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float* __restrict__ red, int C, float* __restrict__ tot,
                                                             float* __restrict__ dgamma_acc, float* __restrict__ dbeta_acc) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float sg = slot_sum(red, C, c), sgx = slot_sum(red, C, C + c);
  tot[c] = sg;
  tot[C + c] = sgx;
  if (dgamma_acc) dgamma_acc[c] += sgx;
  if (dbeta_acc) dbeta_acc[c] += sg;
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat));  dres = g (when the block has an identity branch)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          const float* __restrict__ x, const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ tot, int relu, long M, int C,
                                                          float* __restrict__ dx, float* __restrict__ dres) {
  const int C4 = C / 4;
  const int cg = threadIdx.x % C4;
  const int rl = threadIdx.x / C4, RL = 256 / C4;
  const float invM = 1.0f / (float)M;
  float mu[4], is[4], k1[4], mg[4], mgx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = cg * 4 + k;
    mu[k] = save_mean[c]; is[k] = save_invstd[c];
    k1[k] = gamma[c] * is[k];
    mg[k] = tot[c] * invM; mgx[k] = tot[C + c] * invM;
  }
  for (long m = (long)blockIdx.x * RL + rl; m < M; m += (long)gridDim.x * RL) {
    const size_t o = (size_t)m * C + cg * 4;
    float4 g = *reinterpret_cast<const float4*>(dy + o);
    if (relu) {
      const float4 yy = *reinterpret_cast<const float4*>(y + o);
      g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f; g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
    }
    const float4 v = *reinterpret_cast<const float4*>(x + o);
    float4 r;
    r.x = k1[0] * (g.x - mg[0] - (v.x - mu[0]) * is[0] * mgx[0]);
    r.y = k1[1] * (g.y - mg[1] - (v.y - mu[1]) * is[1] * mgx[1]);
    r.z = k1[2] * (g.z - mg[2] - (v.z - mu[2]) * is[2] * mgx[2]);
    r.w = k1[3] * (g.w - mg[3] - (v.w - mu[3]) * is[3] * mgx[3]);
    *reinterpret_cast<float4*>(dx + o) = r;
    if (dres) *reinterpret_cast<float4*>(dres + o) = g;
  }
}

static unsigned bn_grid(long M, int C) {
  const int RL = 256 / (C / 4);
  long blocks = (M + RL - 1) / RL;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}
static bool bn_c_ok(int C) { return C >= 4 && C % 4 == 0 && (256 % (C / 4)) == 0; }

hipError_t launch_bn_stats(const float* x, long M, int C, float* stats, hipStream_t st) {
  if (!bn_c_ok(C)) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(stats, 0, (size_t)kStatSlots * 2 * C * sizeof(float), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(bn_grid(M, C)), dim3(256), 0, st, x, M, C, stats);
  return hipGetLastError();
}

hipError_t launch_bn_act_fwd(const float* x, const float* stats, const float* gamma, const float* beta, const float* residual,
                             int relu, long M, int C, float eps, float momentum, float* y, float* save_mean, float* save_invstd,
                             float* running_mean, float* running_var, hipStream_t st) {
  if (!bn_c_ok(C)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, st, stats, M, C, eps, momentum, save_mean,
                     save_invstd, running_mean, running_var);
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(bn_grid(M, C)), dim3(256), 0, st, x, save_mean, save_invstd, gamma, beta, residual,
                     relu, M, C, y);
  return hipGetLastError();
}

hipError_t launch_bn_act_bwd(const float* dy, const float* y, const float* x, const float* save_mean, const float* save_invstd,
                             const float* gamma, int relu, long M, int C, float* red, float* dx, float* dres, float* dgamma_acc,
                             float* dbeta_acc, hipStream_t st) {
  if (!bn_c_ok(C)) return hipErrorInvalidValue;
  // red: kStatSlots slot partials followed by the [2][C] totals
  hipError_t e = hipMemsetAsync(red, 0, (size_t)kStatSlots * 2 * C * sizeof(float), st);
  if (e != hipSuccess) return e;
  float* tot = red + (size_t)kStatSlots * 2 * C;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(bn_grid(M, C)), dim3(256), 0, st, dy, y, x, save_mean, save_invstd, relu, M, C, red);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, st, red, C, tot, dgamma_acc, dbeta_acc);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_grid(M, C)), dim3(256), 0, st, dy, y, x, save_mean, save_invstd, gamma, tot, relu,
                     M, C, dx, dres);
  return hipGetLastError();
}

}  // namespace hifihr
