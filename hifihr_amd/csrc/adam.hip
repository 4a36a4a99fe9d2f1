// Fused Adam over one flat fp32 parameter buffer (HBM-bound: 16 B read + 12 B written per parameter).
// Replaces torch.optim.Adam.step() as the reference configures it (reference train_hrnet.py:546-551:
// betas (0.9, 0.999), eps 1e-8, weight_decay 0 -- or 0.01 coupled L2 when optimizer == "AdamW", which in
// the reference is still optim.Adam) plus, for data parallel runs, the 1/world scaling of the all-reduced
// gradient, which is folded into the gradient read.
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"

namespace hifihr {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, size_t n, float grad_scale, float beta1, float beta2,
                                                  float eps, float weight_decay, float step_size, float inv_sqrt_bc2,
                                                  const float* __restrict__ dyn) {
  // dyn (optional, device float[2] = {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)}): lets a captured hipGraph replay
  // the step with per-step scalars that the host refreshes outside the graph
  if (dyn) { step_size = dyn[0]; inv_sqrt_bc2 = dyn[1]; }
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    float* pa = reinterpret_cast<float*>(&pp);
    float* ga = reinterpret_cast<float*>(&gg);
    float* ma = reinterpret_cast<float*>(&mm);
    float* va = reinterpret_cast<float*>(&vv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = ga[k] * grad_scale + weight_decay * pa[k];
      ma[k] = beta1 * ma[k] + (1.f - beta1) * gr;
      va[k] = beta2 * va[k] + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(va[k]) * inv_sqrt_bc2 + eps;
      pa[k] = pa[k] - step_size * (ma[k] / denom);
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  // tail (n not a multiple of 4)
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gr = g[i] * grad_scale + weight_decay * p[i];
    const float mi = beta1 * m[i] + (1.f - beta1) * gr;
    const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
}

// The same update with the step counter and the learning rate in DEVICE memory (round 5): the kernel derives the bias corrections itself
// (one thread per workgroup, double precision: the expressions launch_adam evaluates on the host) and the last workgroup to finish advances
// the counter -- every workgroup read it when it started, none starts after the last one has finished.  A captured step replays with
// nothing to refresh from the host: the per-step upload of {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} (hifihr_adam_step_dyn) was a blit
// kernel of its own in front of every replay.
struct AdamState {          // 48 bytes (include/hifihr.h: hifihr_adam_step_counted)
  double lr, beta1, beta2;
  double pow1, pow2;        // beta1^step, beta2^step (running products: a double pow() per workgroup start costs microseconds)
  int step;                 // completed steps
  int done;                 // workgroups of the running launch that have finished (zero between launches)
};
__global__ __launch_bounds__(256) void adam_kernel_counted(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, size_t n, float grad_scale, float eps, float weight_decay,
                                                          AdamState* __restrict__ st) {
  __shared__ float sc[4];
  int t = 0;
  double q1 = 0.0, q2 = 0.0;
  if (threadIdx.x == 0) {
    t = st->step + 1;
    const double b1 = st->beta1, b2 = st->beta2;
    q1 = st->pow1 * b1; q2 = st->pow2 * b2;                  // beta^t
    sc[0] = (float)(st->lr / (1.0 - q1));
    sc[1] = (float)(1.0 / sqrt(1.0 - q2));
    sc[2] = (float)b1; sc[3] = (float)b2;
  }
  __syncthreads();
  const float step_size = sc[0], inv_sqrt_bc2 = sc[1], beta1 = sc[2], beta2 = sc[3];
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float4* p4 = reinterpret_cast<float4*>(p);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    float* pa = reinterpret_cast<float*>(&pp);
    float* ga = reinterpret_cast<float*>(&gg);
    float* ma = reinterpret_cast<float*>(&mm);
    float* va = reinterpret_cast<float*>(&vv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = ga[k] * grad_scale + weight_decay * pa[k];
      ma[k] = beta1 * ma[k] + (1.f - beta1) * gr;
      va[k] = beta2 * va[k] + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(va[k]) * inv_sqrt_bc2 + eps;
      pa[k] = pa[k] - step_size * (ma[k] / denom);
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gr = g[i] * grad_scale + weight_decay * p[i];
    const float mi = beta1 * m[i] + (1.f - beta1) * gr;
    const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
  __syncthreads();
  // No fence: a workgroup's reads of the state are consumed (data dependence) long before its arrival below, and the last arrival's stores
  // only have to be visible to the NEXT launch.  (__threadfence() here writes the L2 back once per workgroup -- with 140 MB of freshly
  // written moments in it: measured +110 us per launch.)
  if (threadIdx.x == 0) {
    if (atomicAdd(&st->done, 1) == (int)gridDim.x - 1) {       // (relaxed, device scope) the last workgroup
      st->step = t;
      st->pow1 = q1; st->pow2 = q2;
      st->done = 0;
    }
  }
}

size_t adam_state_bytes() { return sizeof(AdamState); }

hipError_t launch_adam_counted(float* p, const float* g, float* m, float* v, size_t n, float grad_scale, float eps, float weight_decay,
                               void* state, hipStream_t st) {
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel_counted, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n, grad_scale, eps, weight_decay,
                     static_cast<AdamState*>(state));
  return hipGetLastError();
}

hipError_t launch_adam(float* p, const float* g, float* m, float* v, size_t n, float grad_scale, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int step, const float* dyn, hipStream_t st) {
  float step_size = 0.f, inv_sqrt_bc2 = 0.f;
  if (!dyn) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    step_size = (float)((double)lr / bc1);
    inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  }
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;        // grid-stride: ~8 workgroups per CU
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n, grad_scale, beta1, beta2, eps,
                     weight_decay, step_size, inv_sqrt_bc2, dyn);
  return hipGetLastError();
}

}  // namespace hifihr
