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

// ------------------------------------------------------------------------------------------------
// The two little layers between the pooling and the scaling, fused (round 4).  Round 3 ran them on the head kernels of csrc/mlp.hip:
// 2 launches forward and 4 backward per block -- 168 launches of 14-24 us, 3.05 ms of the EfficientNet-b3 step at batch 48 -- each
// bounded by its latency (a [48][<= 2304] x [<= 96][<= 2304] product is ~10 MFLOP).  Here
//   se_mlp_fwd_kernel    one workgroup per SAMPLE: the pooled means in LDS, layer 1 as wave dot products over the channels (W1[SQ][C]
//                        rows read coalesced), swish, layer 2 one channel per thread over the transposed W2T[SQ][C] (coalesced too: the
//                        per-step weight re-layout launch provides it), sigmoid; it also hands the pooled-sum accumulator back zeroed
//                        (no fill launch per block) and keeps mean / z1 / h1 for the backward;
//   se_mlp_bwd_x_kernel  one workgroup per sample: dz2 = dgate gate (1 - gate), dh1 = dz2 W2 (wave dot products over W2T rows), dz1 =
//                        dh1 swish'(z1);
//   se_mlp_bwd_w_kernel  one workgroup per 8 channels, over all samples: dmean = dz1 W1 (its W1 columns in LDS, read once for the
//                        whole batch), dW2 += dz2^T h1, db2, dW1 += dz1^T mean, db1 -- every
//                        output element is owned by one thread and summed over the batch in a fixed order (bit-reproducible, like the
//                        kernels it replaces), accumulated straight into the caller's (flat) gradient buffers.
// Replaces reference network/efficientnet_pt/model.py:83-85 (_se_reduce, swish, _se_expand; the sigmoid of :86) and their autograd.
// ------------------------------------------------------------------------------------------------
constexpr int kSeMaxC = 4096, kSeMaxSQ = 256;
constexpr int kSeThreads = 1024;   // per-sample kernels: 16 waves (48 samples are only 48 workgroups: the parallelism has to come from inside)

__device__ __forceinline__ float se_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// dot(row[0 .. C), vec[0 .. C)) by one wave, vec in LDS; eight 16-byte loads of the row in flight per lane (these kernels are chains of
// L2 latencies: the first version issued one load per trip and ran SLOWER than the two head-kernel launches it replaced)
__device__ __forceinline__ float se_wave_dot(const float* __restrict__ row, const float* vec, int C4, int lane) {
  const float4* wr = reinterpret_cast<const float4*>(row);
  float acc = 0.f;
  for (int base = 0; base < C4; base += 8 * 64) {
    float4 w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c4 = base + 64 * u + lane;
      w[u] = c4 < C4 ? wr[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c4 = base + 64 * u + lane;
      if (c4 < C4) {
        const float4 m = *reinterpret_cast<const float4*>(&vec[4 * c4]);
        acc = fmaf(w[u].x, m.x, acc); acc = fmaf(w[u].y, m.y, acc); acc = fmaf(w[u].z, m.z, acc); acc = fmaf(w[u].w, m.w, acc);
      }
    }
  }
  return se_wave_sum(acc);
}

// sum_j vec[j] M[j][c] for one channel c per thread (M[SQ][C], vec in LDS): sixteen loads in flight, two accumulation chains
__device__ __forceinline__ float se_col_dot(const float* __restrict__ M, const float* vec, int C, int SQ, int c) {
  float acc0 = 0.f, acc1 = 0.f;
  for (int j0 = 0; j0 < SQ; j0 += 16) {
    float w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = (j0 + u < SQ) ? M[(size_t)(j0 + u) * C + c] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
      if (j0 + u < SQ) acc0 = fmaf(vec[j0 + u], w[u], acc0);
      if (j0 + u + 1 < SQ) acc1 = fmaf(vec[j0 + u + 1], w[u + 1], acc1);
    }
  }
  return acc0 + acc1;
}

__global__ __launch_bounds__(kSeThreads) void se_mlp_fwd_kernel(float* __restrict__ mean_acc, const float* __restrict__ W1, const float* __restrict__ b1,
                                                               const float* __restrict__ W2T, const float* __restrict__ b2, int C, int SQ,
                                                               float* __restrict__ mean_out, float* __restrict__ z1, float* __restrict__ h1,
                                                               float* __restrict__ gate) {
  __shared__ __attribute__((aligned(16))) float s_mean[kSeMaxC];
  __shared__ float s_h1[kSeMaxSQ];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < C; c += kSeThreads) {
    const float v = mean_acc[(size_t)b * C + c];
    s_mean[c] = v;
    mean_out[(size_t)b * C + c] = v;
    mean_acc[(size_t)b * C + c] = 0.f;                      // the accumulator goes back to the zero pool
  }
  __syncthreads();
  for (int j = wave; j < SQ; j += kSeThreads / 64) {
    const float acc = se_wave_dot(W1 + (size_t)j * C, s_mean, C / 4, lane);
    if (lane == 0) {
      const float z = acc + b1[j];
      const float h = z * fast_sigmoid(z);                  // swish (csrc/mlp.hip act 2: same expression)
      z1[(size_t)b * SQ + j] = z;
      h1[(size_t)b * SQ + j] = h;
      s_h1[j] = h;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += kSeThreads) {
    const float v = se_col_dot(W2T, s_h1, C, SQ, c) + b2[c];
    gate[(size_t)b * C + c] = fast_sigmoid(v);              // sigmoid (act 3)
  }
}

__global__ __launch_bounds__(kSeThreads) void se_mlp_bwd_x_kernel(float* __restrict__ dgate_acc, const float* __restrict__ gate,
                                                                 const float* __restrict__ z1,
                                                                 const float* __restrict__ W2T, int C, int SQ, float* __restrict__ dz2,
                                                                 float* __restrict__ dz1) {
  __shared__ __attribute__((aligned(16))) float s_dz2[kSeMaxC];
  __shared__ float s_dz1[kSeMaxSQ];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < C; c += kSeThreads) {
    const float d = dgate_acc[(size_t)b * C + c], g = gate[(size_t)b * C + c];
    const float v = d * (g * (1.f - g));                    // (csrc/mlp.hip act 3: v *= y (1 - y))
    s_dz2[c] = v;
    dz2[(size_t)b * C + c] = v;
    dgate_acc[(size_t)b * C + c] = 0.f;
  }
  __syncthreads();
  for (int j = wave; j < SQ; j += kSeThreads / 64) {
    const float acc = se_wave_dot(W2T + (size_t)j * C, s_dz2, C / 4, lane);
    if (lane == 0) {
      const float z = z1[(size_t)b * SQ + j], sg = fast_sigmoid(z);
      const float v = acc * (sg * (1.f + z * (1.f - sg)));  // swish' (act 2)
      s_dz1[j] = v;
      dz1[(size_t)b * SQ + j] = v;
    }
  }
}

constexpr int kSeWB = 48;          // samples per pass of the weight-gradient kernel (LDS: 143 KB at SQ = 256)
constexpr int kSeWC = 8;           // channels per workgroup (32: 12 us of LDS reads per workgroup on the 2304-channel blocks and 72 workgroups; 8: 288)
// workgroup = channels [kSeWC blockIdx.x, + kSeWC); dynamic LDS: h1 / dz1 [kSeWB][SQ] each, dz2 / mean [kSeWB][kSeWC] each.  Every
// output element belongs to ONE thread, which sums it over the batch in order and adds it to the gradient buffer with a no-return
// atomic: a single adder per address, so the result is still bit-reproducible, and no read-modify-write latency per element.
__global__ __launch_bounds__(256) void se_mlp_bwd_w_kernel(const float* __restrict__ dz2, const float* __restrict__ dz1, const float* __restrict__ h1,
                                                          const float* __restrict__ mean, const float* __restrict__ W1, int B, int C, int SQ,
                                                          float* __restrict__ dmean, float* __restrict__ dW1_acc, float* __restrict__ db1_acc,
                                                          float* __restrict__ dW2_acc, float* __restrict__ db2_acc) {
  HIP_DYNAMIC_SHARED(float, se_smem)
  float* s_h1 = se_smem;
  float* s_dz1 = s_h1 + kSeWB * SQ;
  float* s_dz2 = s_dz1 + kSeWB * SQ;
  float* s_mean = s_dz2 + kSeWB * kSeWC;
  float* s_w1 = s_mean + kSeWB * kSeWC;                      // W1[j][c0 .. + kSeWC): read ONCE for all samples (a sample per workgroup re-read it 48 times)
  const int tid = threadIdx.x, c0 = blockIdx.x * kSeWC, nc = min(kSeWC, C - c0);
  for (int e = tid; e < SQ * kSeWC; e += 256) {
    const int j = e / kSeWC, c = e - j * kSeWC;
    s_w1[e] = c < nc ? W1[(size_t)j * C + c0 + c] : 0.f;
  }
  for (int b0 = 0; b0 < B; b0 += kSeWB) {
    const int nb = min(kSeWB, B - b0);
    __syncthreads();
    for (int e = tid; e < nb * SQ; e += 256) { s_h1[e] = h1[(size_t)b0 * SQ + e]; s_dz1[e] = dz1[(size_t)b0 * SQ + e]; }
    for (int e = tid; e < nb * kSeWC; e += 256) {
      const int bb = e / kSeWC, c = e - bb * kSeWC;
      const bool ok = c < nc;
      s_dz2[e] = ok ? dz2[(size_t)(b0 + bb) * C + c0 + c] : 0.f;
      s_mean[e] = ok ? mean[(size_t)(b0 + bb) * C + c0 + c] : 0.f;
    }
    __syncthreads();
    // dmean[b][c0 + c] = sum_j dz1[b][j] W1[j][c0 + c]: the gradient that reaches the pooled means (se_scale adds it / HW to dx)
    for (int o = tid; o < nb * kSeWC; o += 256) {
      const int bb = o / kSeWC, c = o - bb * kSeWC;
      if (c < nc) {
        float acc = 0.f;
        for (int j = 0; j < SQ; ++j) acc = fmaf(s_dz1[bb * SQ + j], s_w1[j * kSeWC + c], acc);
        dmean[(size_t)(b0 + bb) * C + c0 + c] = acc;
      }
    }
    // dW2[c0 + c][j]: contiguous over o = c SQ + j
    for (int o = tid; o < nc * SQ; o += 256) {
      const int c = o / SQ, j = o - c * SQ;
      float acc = 0.f;
      for (int bb = 0; bb < nb; ++bb) acc = fmaf(s_dz2[bb * kSeWC + c], s_h1[bb * SQ + j], acc);
      atomicAdd(&dW2_acc[(size_t)c0 * SQ + o], acc);
    }
    // dW1[j][c0 + c]: contiguous over c
    for (int o = tid; o < SQ * kSeWC; o += 256) {
      const int j = o / kSeWC, c = o - j * kSeWC;
      if (c < nc) {
        float acc = 0.f;
        for (int bb = 0; bb < nb; ++bb) acc = fmaf(s_dz1[bb * SQ + j], s_mean[bb * kSeWC + c], acc);
        atomicAdd(&dW1_acc[(size_t)j * C + c0 + c], acc);
      }
    }
    if (tid < nc) {
      float acc = 0.f;
      for (int bb = 0; bb < nb; ++bb) acc += s_dz2[bb * kSeWC + tid];
      atomicAdd(&db2_acc[c0 + tid], acc);
    }
    if (blockIdx.x == 0)
      for (int j = tid; j < SQ; j += 256) {
        float acc = 0.f;
        for (int bb = 0; bb < nb; ++bb) acc += s_dz1[bb * SQ + j];
        atomicAdd(&db1_acc[j], acc);
      }
  }
}

bool se_mlp_supported(int C, int SQ) { return C >= 4 && C % 4 == 0 && C <= kSeMaxC && SQ >= 1 && SQ <= kSeMaxSQ; }

hipError_t launch_se_mlp_fwd(float* mean_acc, const float* W1, const float* b1, const float* W2T, const float* b2, int B, int C, int SQ,
                             float* mean_out, float* z1, float* h1, float* gate, hipStream_t st) {
  if (!se_mlp_supported(C, SQ)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(B), dim3(kSeThreads), 0, st, mean_acc, W1, b1, W2T, b2, C, SQ, mean_out, z1, h1, gate);
  return hipGetLastError();
}

hipError_t launch_se_mlp_bwd(float* dgate_acc, const float* gate, const float* z1, const float* h1, const float* mean, const float* W1,
                             const float* W2T, int B, int C, int SQ, float* dz2, float* dz1, float* dmean, float* dW1_acc, float* db1_acc,
                             float* dW2_acc, float* db2_acc, hipStream_t st) {
  if (!se_mlp_supported(C, SQ)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(se_mlp_bwd_x_kernel, dim3(B), dim3(kSeThreads), 0, st, dgate_acc, gate, z1, W2T, C, SQ, dz2, dz1);
  const size_t lds = (size_t)(2 * kSeWB * SQ + 2 * kSeWB * kSeWC + SQ * kSeWC) * sizeof(float);
  // the attribute is per DEVICE: a process-wide "already set" flag left the second GPU of a process without it; a launch that fits the
  // default 64 KB needs no attribute at all (and nothing is then issued inside a stream capture)
  if (lds > 64 * 1024) {
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidValue;
    if (!attr_set[dev]) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(se_mlp_bwd_w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)((2 * kSeWB * kSeMaxSQ + 2 * kSeWB * kSeWC + kSeMaxSQ * kSeWC) * sizeof(float)));
      if (e != hipSuccess) return e;
      attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL(se_mlp_bwd_w_kernel, dim3((C + kSeWC - 1) / kSeWC), dim3(256), lds, st, dz2, dz1, h1, mean, W1, B, C, SQ, dmean, dW1_acc, db1_acc,
                     dW2_acc, db2_acc);
  return hipGetLastError();
}

// drop-connect + skip of an MBConv block (reference network/efficientnet_pt/utils.py:82-91, model.py:91-94) in ONE pass:
//   out = x / keep * floor(keep + u[b]) (+ skip),  u[b] the block's per-sample uniform draw;  the backward is the same kernel on dy
// without a skip.  ATen ran six launches forward (add, floor, div, mul on broadcast shapes, add) and four backward per skip block.
__global__ __launch_bounds__(256) void drop_connect_add_kernel(const float* __restrict__ x, const float* __restrict__ skip, const float* __restrict__ u,
                                                              float keep, int B, size_t per4, float* __restrict__ out) {
  const float inv_keep = 1.0f / keep;
  const size_t total = (size_t)B * per4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i / per4);
    const float m = floorf(keep + u[b]);
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    float4 r = make_float4(v.x * inv_keep * m, v.y * inv_keep * m, v.z * inv_keep * m, v.w * inv_keep * m);
    if (skip != nullptr) {
      const float4 a = *reinterpret_cast<const float4*>(skip + i * 4);
      r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
    }
    *reinterpret_cast<float4*>(out + i * 4) = r;
  }
}

hipError_t launch_drop_connect_add(const float* x, const float* skip, const float* u, float keep, int B, size_t per_sample, float* out,
                                   hipStream_t st) {
  if (per_sample % 4 != 0 || !(keep > 0.f)) return hipErrorInvalidValue;
  size_t blocks = ((size_t)B * (per_sample / 4) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(drop_connect_add_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, skip, u, keep, B, per_sample / 4, out);
  return hipGetLastError();
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
