// Train-mode batch normalisation fused with the residual add and the activation (ReLU / swish), NHWC (x[M][C],
// M = N*H*W), fp32, any C % 4 == 0 (C <= 4096).
//
// Replaces, per block of the reference's encoders (ResNet BasicBlock: reference network/res_encoder.py:364-373 +
// vendored utils/Freihand_GNN_mano/network/resnet.py; EfficientNet MBConv: network/efficientnet_pt/model.py:67-94):
// nn.BatchNorm2d in training mode (batch statistics, running-stat update, unbiased running variance) + `out += identity`
// + ReLU, or + MemoryEfficientSwish (utils.py:36-52), and their autograd, with
//   forward : per-channel sum / sum-of-squares come out of the convolution epilogue (conv.hip, dwconv.hip, wino.hip) or
//             bn_stats_kernel; ONE apply kernel whose workgroups each fold the slot partials into scale / shift in LDS
//             (workgroup 0 also writes mean / invstd and the running statistics)
//   backward: ONE reduction kernel (sum g, sum g*xhat, activation derivative applied on the fly), ONE apply kernel that
//             folds the slots the same way (workgroup 0 accumulates dgamma / dbeta straight into the flat gradient buffer)
//             and writes dx (and the masked gradient of the identity branch).
//   Up to 512 channels the separate one-thread-per-channel finalize launches of the first version (2 per layer, ~6.8 us
//   each in a replayed graph) are gone: the LAST workgroup of an apply kernel to finish -- elected through 32 + 1 arrival
//   counters behind the slots -- zeroes the slots for the next producer.  Wider layers keep the finalize launch (kFuseMaxC).
// All kernels are HBM-bound: every lane moves float4 (4 consecutive channels) and keeps the same channel group(s) for
// its whole grid-stride loop, so scale / shift live in registers.  Partial sums use float atomics spread over
// kStatSlots copies (thousands of atomics on ONE address serialise at ~100 ns each).
// The slot buffers are SELF-CLEANING: producers add into a buffer that must be all zero on entry, and the finalize
// kernel that folds the slots writes the zeros back -- no memset launch per batch-norm (40 per ResNet-18 step).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdlib>

#include "hifihr_internal.h"
#include "bn_fold.h"

namespace hifihr {

constexpr int kBnUnroll = 4;    // rows per trip of the streaming loops (independent 16-byte loads in flight per lane)
constexpr int kMaxNG = 4;        // channel groups (of 4 channels) per thread: C <= 4 * 256 * kMaxNG = 4096

// Thread -> (row lane, channel groups).  C4 <= 256: CT = C4 threads cover a row, RL = 256 / CT rows per pass, one
// group per thread.  C4 > 256: one row per pass, thread t owns groups t, t + 256, ...
struct BnMap {
  int CT, RL, NG, cg0, rl;
  bool active;
};
__device__ __forceinline__ BnMap bn_map(int C) {
  BnMap m;
  const int C4 = C / 4;
  if (C4 <= 256) {
    m.CT = C4; m.RL = 256 / C4; m.NG = 1;
    m.cg0 = threadIdx.x % C4; m.rl = threadIdx.x / C4;
    m.active = m.rl < m.RL;
  } else {
    m.CT = 256; m.RL = 1; m.NG = (C4 + 255) / 256;
    m.cg0 = threadIdx.x; m.rl = 0; m.active = true;
  }
  return m;
}

__device__ __forceinline__ void atomic_add4(float* p, const float4& v) {
  atomicAdd(p, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}
__device__ __forceinline__ void acc4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// d act(z) / dz
__device__ __forceinline__ float act_grad(int act, float z, float y) {
  if (act == 1) return y > 0.f ? 1.f : 0.f;
  if (act == 2) {
    const float s = fast_sigmoid(z);
    return s * (1.f + z * (1.f - s));
  }
  return 1.f;
}

// block-level reduction of the per-thread (sum, sumsq)-like pairs over the row lanes + slot-spread atomics
template <int MNG>
__device__ __forceinline__ void reduce_and_add(const BnMap& mp, int C, float4 (&s)[MNG], float4 (&q)[MNG],
                                               float4 (*lds)[256], float* __restrict__ out) {
  const int slot = blockIdx.x & (stat_slots_used(C) - 1);
  float* base = out + (size_t)slot * 2 * C;
  if (mp.NG == 1 && mp.RL > 1) {
    lds[0][threadIdx.x] = s[0]; lds[1][threadIdx.x] = q[0];
    __syncthreads();
    if (mp.active && mp.rl == 0) {
      for (int r = 1; r < mp.RL; ++r) { acc4(s[0], lds[0][r * mp.CT + mp.cg0]); acc4(q[0], lds[1][r * mp.CT + mp.cg0]); }
      atomic_add4(base + mp.cg0 * 4, s[0]);
      atomic_add4(base + C + mp.cg0 * 4, q[0]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < MNG; ++j) {
      const int cg = mp.cg0 + 256 * j;
      if (j < mp.NG && cg * 4 < C && mp.active) {
        atomic_add4(base + cg * 4, s[j]);
        atomic_add4(base + C + cg * 4, q[j]);
      }
    }
  }
}

// Forward statistics of a tensor whose producer is not one of our convolutions: shifted sums per thread (hifihr_internal.h, "FORWARD
// statistics"), unshifted in fp64, folded over the row lanes through LDS and added to the double slots.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long M, int C, float* __restrict__ stats) {
  __shared__ double lds[2][256][4];
  const BnMap mp = bn_map(C);
  double* const slots = reinterpret_cast<double*>(stats) + (size_t)(blockIdx.x & (stat_slots_used(C) - 1)) * 2 * C;
#pragma unroll
  for (int j = 0; j < kMaxNG; ++j) {
    if (j >= mp.NG) break;                                 // (uniform)
    const int cg = mp.cg0 + 256 * j;
    const bool mine = mp.active && cg * 4 < C;
    float4 k = make_float4(0.f, 0.f, 0.f, 0.f), s = k, q = k;
    int n = 0;
    if (mine) {
      for (long m = (long)blockIdx.x * mp.RL + mp.rl; m < M; m += (long)gridDim.x * mp.RL) {
        const float4 v = *reinterpret_cast<const float4*>(x + m * C + cg * 4);
        if (n == 0) k = v;
        const float4 d = make_float4(v.x - k.x, v.y - k.y, v.z - k.z, v.w - k.w);
        acc4(s, d);
        q.x += d.x * d.x; q.y += d.y * d.y; q.z += d.z * d.z; q.w += d.w * d.w;
        ++n;
      }
    }
    double S1[4], S2[4];
    stat_unshift(n, k.x, s.x, q.x, S1[0], S2[0]); stat_unshift(n, k.y, s.y, q.y, S1[1], S2[1]);
    stat_unshift(n, k.z, s.z, q.z, S1[2], S2[2]); stat_unshift(n, k.w, s.w, q.w, S1[3], S2[3]);
    if (mp.NG == 1 && mp.RL > 1) {                        // (uniform) fold the row lanes of a channel group
#pragma unroll
      for (int e = 0; e < 4; ++e) { lds[0][threadIdx.x][e] = S1[e]; lds[1][threadIdx.x][e] = S2[e]; }
      __syncthreads();
      if (mp.active && mp.rl == 0) {
        for (int r = 1; r < mp.RL; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) { S1[e] += lds[0][r * mp.CT + mp.cg0][e]; S2[e] += lds[1][r * mp.CT + mp.cg0][e]; }
      }
    }
    if (mine && (mp.RL == 1 || mp.NG > 1 || mp.rl == 0)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { stat_atomic_add(slots + cg * 4 + e, S1[e]); stat_atomic_add(slots + C + cg * 4 + e, S2[e]); }
    }
  }
}

constexpr int kMaxC = 4 * 256 * kMaxNG;     // 4096 channels: scale / shift tables of the apply kernels (2 x 16 KB of LDS)

// Wide layers (C > kFuseMaxC): folding 256 C bytes of partials in EVERY workgroup costs more than it saves (EfficientNet's
// 1392-channel layers: +1.2 ms per step measured), so they keep a one-thread-per-channel finalize launch that also cleans the slots.
constexpr int kFuseMaxC = 512;     // 256 -> 512 in round 2: ResNet-18's layer 4 loses its 10 finalize launches per step (6.19 -> 6.175 ms)

__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(float* __restrict__ stats, long M, int C, float eps, float momentum,
                                                             float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double* buf = reinterpret_cast<double*>(stats);
  double s = 0.0, q = 0.0;
  const int ns = stat_slots_used(C);
#pragma unroll 8
  for (int sl = 0; sl < ns; ++sl) {
    double* p = buf + (size_t)sl * 2 * C;
    s += p[c]; q += p[C + c];
    p[c] = 0.0; p[C + c] = 0.0;
  }
  const double md = s / (double)M, vd = q / (double)M - md * md;
  const float mu = (float)md;
  const float var = vd > 0.0 ? (float)vd : 0.f;
  save_mean[c] = mu;
  save_invstd[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    const float unbiased = (M > 1) ? var * ((float)M / (float)(M - 1)) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// tot[2][C] = slot sums of the backward reduction (behind the slots), slots cleaned, dgamma / dbeta accumulated
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(float* __restrict__ red, int C, float* __restrict__ dgamma_acc,
                                                             float* __restrict__ dbeta_acc) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float sg = 0.f, sgx = 0.f;
  const int ns = stat_slots_used(C);
#pragma unroll 8
  for (int sl = 0; sl < ns; ++sl) {
    float* p = red + (size_t)sl * 2 * C;
    sg += p[c]; sgx += p[C + c];
    p[c] = 0.f; p[C + c] = 0.f;
  }
  float* tot = stat_bwd_totals(red, C);
  tot[c] = sg;
  tot[C + c] = sgx;
  if (dgamma_acc) dgamma_acc[c] += sgx;
  if (dbeta_acc) dbeta_acc[c] += sg;
}

// PRE = true: mean / invstd (forward) or the totals behind the slots (backward) were produced by a finalize launch
// MNG: channel groups per thread the registers are sized for (1: C <= 1024 -- every ResNet layer -- 80-100 registers less than the
// general form, i.e. 4-5 waves per SIMD instead of 2-3; kMaxNG: up to 4096 channels)
template <bool PRE, int MNG>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ residual, int act, long M,
                                                        int C, float eps, float momentum, float* __restrict__ y, float* __restrict__ save_mean,
                                                        float* __restrict__ save_invstd, float* __restrict__ running_mean,
                                                        float* __restrict__ running_var) {
  __shared__ float s_sc[kMaxC], s_sh[kMaxC];
  const BnMap mp = bn_map(C);
  for (int c = threadIdx.x; c < C; c += 256) {            // every workgroup folds the slot partials itself (L2-resident)
    float mu, is, var = 0.f;
    if (PRE) {
      mu = save_mean[c]; is = save_invstd[c];
      if (stats == nullptr) is = 1.0f / sqrtf(is + eps);       // evaluation mode: (running_mean, running_var) were passed in
    } else {
      slot_mean_var(stats, C, c, M, mu, var);                               // biased batch variance (fp64 fold of the double slots)
      is = 1.0f / sqrtf(var + eps);
    }
    const float sc = is * gamma[c];
    s_sc[c] = sc;
    s_sh[c] = beta[c] - mu * sc;
    if (!PRE && blockIdx.x == 0) {
      save_mean[c] = mu;
      save_invstd[c] = is;
      if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        const float unbiased = (M > 1) ? var * ((float)M / (float)(M - 1)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
  }
  __syncthreads();
  if (mp.active) {
    float4 scale[MNG], shift[MNG];
#pragma unroll
    for (int j = 0; j < MNG; ++j) {
      const int cg = mp.cg0 + 256 * j;
      if (j < mp.NG && cg * 4 < C) {
        scale[j] = *reinterpret_cast<const float4*>(&s_sc[cg * 4]);
        shift[j] = *reinterpret_cast<const float4*>(&s_sh[cg * 4]);
      }
    }
    auto finish = [&](float4 r, const float4& res, size_t o) {
      if (residual) acc4(r, res);
      if (act == 1) {
        r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);
      } else if (act == 2) {
        r.x = r.x * fast_sigmoid(r.x); r.y = r.y * fast_sigmoid(r.y); r.z = r.z * fast_sigmoid(r.z); r.w = r.w * fast_sigmoid(r.w);
      }
      *reinterpret_cast<float4*>(y + o) = r;
    };
    const long stride = (long)gridDim.x * mp.RL;
    long m = (long)blockIdx.x * mp.RL + mp.rl;
    if (mp.NG == 1) {
      // four rows per trip, their loads issued before the first use: one float4 in flight per lane left the small grids of the
      // deep layers latency-bound (kBnUnroll independent 16-byte loads per lane instead)
      const float4 sc = scale[0], sh = shift[0];
      const int cb = mp.cg0 * 4;
      for (; m + (kBnUnroll - 1) * stride < M; m += kBnUnroll * stride) {
        float4 v[kBnUnroll], rs[kBnUnroll];
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) {
          const size_t o = (size_t)(m + u * stride) * C + cb;
          v[u] = *reinterpret_cast<const float4*>(x + o);
          rs[u] = residual ? *reinterpret_cast<const float4*>(residual + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u)
          finish(make_float4(v[u].x * sc.x + sh.x, v[u].y * sc.y + sh.y, v[u].z * sc.z + sh.z, v[u].w * sc.w + sh.w), rs[u],
                 (size_t)(m + u * stride) * C + cb);
      }
    }
    for (; m < M; m += stride) {
#pragma unroll
      for (int j = 0; j < MNG; ++j) {
        const int cg = mp.cg0 + 256 * j;
        if (j < mp.NG && cg * 4 < C) {
          const size_t o = (size_t)m * C + cg * 4;
          const float4 v = *reinterpret_cast<const float4*>(x + o);
          finish(make_float4(v.x * scale[j].x + shift[j].x, v.y * scale[j].y + shift[j].y, v.z * scale[j].z + shift[j].z,
                             v.w * scale[j].w + shift[j].w),
                 residual ? *reinterpret_cast<const float4*>(residual + o) : make_float4(0.f, 0.f, 0.f, 0.f), o);
        }
      }
    }
  }
  if (!PRE) {
    unsigned* cnt = stat_fwd_counters(stats, C);
    if (last_workgroup(cnt)) clear_slots_fwd(stats, C, cnt);
  }
}

// g = dy * act'(z).  ReLU: the mask is y > 0 read from the forward's output when the layer had a residual input; without one
// y = max(z, 0) with z = x * sc + sh recomputed from x by the forward's own expression (same bits), so y is not read at all
// (y == nullptr: a third less traffic in both backward passes).  Swish recomputes z the same way.
__device__ __forceinline__ float4 masked_grad(int act, const float4& dy, bool have_y, const float4& yv, const float4& v,
                                              const float4& sc, const float4& sh) {
  float4 g = dy;
  if (act == 1) {
    float4 yy;
    if (have_y) yy = yv;
    else yy = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
    g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f; g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
  } else if (act == 2) {
    g.x *= act_grad(2, v.x * sc.x + sh.x, 0.f); g.y *= act_grad(2, v.y * sc.y + sh.y, 0.f);
    g.z *= act_grad(2, v.z * sc.z + sh.z, 0.f); g.w *= act_grad(2, v.w * sc.w + sh.w, 0.f);
  }
  return g;
}

// red[kStatSlots][2][C] += (sum g, sum g * xhat)
template <int MNG>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ x, const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int act, long M, int C,
                                                           float* __restrict__ red) {
  __shared__ float4 lds[2][256];
  const BnMap mp = bn_map(C);
  float4 s[MNG], q[MNG], mu[MNG], is[MNG], sc[MNG], sh[MNG];
#pragma unroll
  for (int j = 0; j < MNG; ++j) {
    s[j] = make_float4(0.f, 0.f, 0.f, 0.f); q[j] = s[j]; mu[j] = s[j]; is[j] = s[j]; sc[j] = s[j]; sh[j] = s[j];
    const int cg = mp.cg0 + 256 * j;
    if (j < mp.NG && cg * 4 < C && mp.active) {
      mu[j] = *reinterpret_cast<const float4*>(save_mean + cg * 4);
      is[j] = *reinterpret_cast<const float4*>(save_invstd + cg * 4);
      if (act == 2 || (act == 1 && y == nullptr)) {
        const float4 ga = *reinterpret_cast<const float4*>(gamma + cg * 4), be = *reinterpret_cast<const float4*>(beta + cg * 4);
        sc[j] = make_float4(is[j].x * ga.x, is[j].y * ga.y, is[j].z * ga.z, is[j].w * ga.w);
        sh[j] = make_float4(be.x - mu[j].x * sc[j].x, be.y - mu[j].y * sc[j].y, be.z - mu[j].z * sc[j].z, be.w - mu[j].w * sc[j].w);
      }
    }
  }
  if (mp.active) {
    const bool have_y = act == 1 && y != nullptr;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto accum = [&](int j, const float4& g, const float4& v) {
      acc4(s[j], g);
      q[j].x += g.x * ((v.x - mu[j].x) * is[j].x); q[j].y += g.y * ((v.y - mu[j].y) * is[j].y);
      q[j].z += g.z * ((v.z - mu[j].z) * is[j].z); q[j].w += g.w * ((v.w - mu[j].w) * is[j].w);
    };
    const long stride = (long)gridDim.x * mp.RL;
    long m = (long)blockIdx.x * mp.RL + mp.rl;
    if (mp.NG == 1) {
      const int cb = mp.cg0 * 4;
      for (; m + (kBnUnroll - 1) * stride < M; m += kBnUnroll * stride) {
        float4 v[kBnUnroll], d[kBnUnroll], yv[kBnUnroll];
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) {
          const size_t o = (size_t)(m + u * stride) * C + cb;
          v[u] = *reinterpret_cast<const float4*>(x + o);
          d[u] = *reinterpret_cast<const float4*>(dy + o);
          yv[u] = have_y ? *reinterpret_cast<const float4*>(y + o) : zero4;
        }
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) accum(0, masked_grad(act, d[u], have_y, yv[u], v[u], sc[0], sh[0]), v[u]);
      }
    }
    for (; m < M; m += stride) {
#pragma unroll
      for (int j = 0; j < MNG; ++j) {
        const int cg = mp.cg0 + 256 * j;
        if (j < mp.NG && cg * 4 < C) {
          const size_t o = (size_t)m * C + cg * 4;
          const float4 v = *reinterpret_cast<const float4*>(x + o);
          accum(j, masked_grad(act, *reinterpret_cast<const float4*>(dy + o), have_y,
                               have_y ? *reinterpret_cast<const float4*>(y + o) : zero4, v, sc[j], sh[j]), v);
        }
      }
    }
  }
  reduce_and_add(mp, C, s, q, lds, red);
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat));  dres = g (when the block has an identity branch).
// Every workgroup folds the slot partials of the reduction into mean(g), mean(g * xhat) in LDS; workgroup 0 accumulates
// dgamma / dbeta; the last workgroup to finish zeroes the slots.
template <bool PRE, int MNG>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                          const float* __restrict__ x, const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ red, int act,
                                                          long M, int C, float* __restrict__ dx, float* __restrict__ dres,
                                                          float* __restrict__ dgamma_acc, float* __restrict__ dbeta_acc) {
  __shared__ float s_mg[kMaxC], s_mgx[kMaxC];
  const BnMap mp = bn_map(C);
  const float invM = 1.0f / (float)M;
  for (int c = threadIdx.x; c < C; c += 256) {
    float sg, sgx;
    if (PRE) {
      const float* tot = stat_bwd_totals(red, C);
      sg = tot[c]; sgx = tot[C + c];
    } else {
      slot_sum2(red, C, c, sg, sgx);
    }
    s_mg[c] = sg * invM;
    s_mgx[c] = sgx * invM;
    if (!PRE && blockIdx.x == 0) {
      if (dgamma_acc) dgamma_acc[c] += sgx;
      if (dbeta_acc) dbeta_acc[c] += sg;
    }
  }
  __syncthreads();
  if (mp.active) {
    float4 mu[MNG], is[MNG], k1[MNG], mg[MNG], mgx[MNG], sh[MNG];
#pragma unroll
    for (int j = 0; j < MNG; ++j) {
      const int cg = mp.cg0 + 256 * j;
      if (j < mp.NG && cg * 4 < C) {
        mu[j] = *reinterpret_cast<const float4*>(save_mean + cg * 4);
        is[j] = *reinterpret_cast<const float4*>(save_invstd + cg * 4);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + cg * 4);
        k1[j] = make_float4(is[j].x * ga.x, is[j].y * ga.y, is[j].z * ga.z, is[j].w * ga.w);
        mg[j] = *reinterpret_cast<const float4*>(&s_mg[cg * 4]);
        mgx[j] = *reinterpret_cast<const float4*>(&s_mgx[cg * 4]);
        sh[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act == 2 || (act == 1 && y == nullptr)) {
          const float4 be = *reinterpret_cast<const float4*>(beta + cg * 4);
          sh[j] = make_float4(be.x - mu[j].x * k1[j].x, be.y - mu[j].y * k1[j].y, be.z - mu[j].z * k1[j].z, be.w - mu[j].w * k1[j].w);
        }
      }
    }
    const bool have_y = act == 1 && y != nullptr;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto emit = [&](int j, const float4& g, const float4& v, size_t o) {
      float4 r;
      r.x = k1[j].x * (g.x - mg[j].x - (v.x - mu[j].x) * is[j].x * mgx[j].x);
      r.y = k1[j].y * (g.y - mg[j].y - (v.y - mu[j].y) * is[j].y * mgx[j].y);
      r.z = k1[j].z * (g.z - mg[j].z - (v.z - mu[j].z) * is[j].z * mgx[j].z);
      r.w = k1[j].w * (g.w - mg[j].w - (v.w - mu[j].w) * is[j].w * mgx[j].w);
      *reinterpret_cast<float4*>(dx + o) = r;
      if (dres) *reinterpret_cast<float4*>(dres + o) = g;
    };
    const long stride = (long)gridDim.x * mp.RL;
    long m = (long)blockIdx.x * mp.RL + mp.rl;
    if (mp.NG == 1) {
      const int cb = mp.cg0 * 4;
      for (; m + (kBnUnroll - 1) * stride < M; m += kBnUnroll * stride) {
        float4 v[kBnUnroll], d[kBnUnroll], yv[kBnUnroll];
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u) {
          const size_t o = (size_t)(m + u * stride) * C + cb;
          v[u] = *reinterpret_cast<const float4*>(x + o);
          d[u] = *reinterpret_cast<const float4*>(dy + o);
          yv[u] = have_y ? *reinterpret_cast<const float4*>(y + o) : zero4;
        }
#pragma unroll
        for (int u = 0; u < kBnUnroll; ++u)
          emit(0, masked_grad(act, d[u], have_y, yv[u], v[u], k1[0], sh[0]), v[u], (size_t)(m + u * stride) * C + cb);
      }
    }
    for (; m < M; m += stride) {
#pragma unroll
      for (int j = 0; j < MNG; ++j) {
        const int cg = mp.cg0 + 256 * j;
        if (j < mp.NG && cg * 4 < C) {
          const size_t o = (size_t)m * C + cg * 4;
          const float4 v = *reinterpret_cast<const float4*>(x + o);
          emit(j, masked_grad(act, *reinterpret_cast<const float4*>(dy + o), have_y, have_y ? *reinterpret_cast<const float4*>(y + o) : zero4,
                              v, k1[j], sh[j]), v, o);
        }
      }
    }
  }
  if (!PRE) {
    unsigned* cnt = stat_bwd_counters(red, C);
    if (last_workgroup(cnt)) clear_slots(red, C, cnt);
  }
}

// ------------------------------------------------------------------------------------------------
// The ResNet stem: MaxPool2d(3, 2, 1)(ReLU(BN(x))) (reference: vendored resnet.py conv1 -> bn1 -> relu -> maxpool) without the
// full-resolution activation or its gradient ever reaching HBM.  At batch 32 that tensor is 103 MB: the separate kernels wrote it,
// read it back for the pool, wrote its gradient in the pool's backward and read that twice in the batch-norm backward.
//   forward : thread = (pooled pixel, 4 channels): nine taps of x through scale / shift / ReLU, maximum + winning tap (ATen's
//             tie rule: the first in-range tap in scan order wins, a later one only if strictly greater)
//   backward: thread = (input pixel, 4 channels): dy of the pool's backward gathered on the fly from the (at most four) windows
//             whose winning tap is this pixel, then the ordinary batch-norm reduction / apply on it.
// ------------------------------------------------------------------------------------------------
struct StemPool {
  int N, H, W, OH, OW;
};

__device__ __forceinline__ float4 bn_relu4(const float4& v, const float4& sc, const float4& sh) {
  return make_float4(fmaxf(v.x * sc.x + sh.x, 0.f), fmaxf(v.y * sc.y + sh.y, 0.f), fmaxf(v.z * sc.z + sh.z, 0.f),
                     fmaxf(v.w * sc.w + sh.w, 0.f));
}

__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, StemPool g,
                                                              int C, float eps, float momentum, float* __restrict__ y,
                                                              unsigned char* __restrict__ tap, float* __restrict__ save_mean,
                                                              float* __restrict__ save_invstd, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var) {
  __shared__ float s_sc[kFuseMaxC], s_sh[kFuseMaxC];
  const long M = (long)g.N * g.H * g.W;
  for (int c = threadIdx.x; c < C; c += 256) {
    float mu, var;
    slot_mean_var(stats, C, c, M, mu, var);
    const float is = 1.0f / sqrtf(var + eps);
    const float sc = is * gamma[c];
    s_sc[c] = sc;
    s_sh[c] = beta[c] - mu * sc;
    if (blockIdx.x == 0) {
      save_mean[c] = mu;
      save_invstd[c] = is;
      if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
        const float unbiased = (M > 1) ? var * ((float)M / (float)(M - 1)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
  }
  __syncthreads();
  const int C4 = C / 4, RL = 256 / C4;
  const int cg = threadIdx.x % C4, rl = threadIdx.x / C4;
  if (rl < RL) {
    const float4 sc = *reinterpret_cast<const float4*>(&s_sc[cg * 4]), sh = *reinterpret_cast<const float4*>(&s_sh[cg * 4]);
    const long Mo = (long)g.N * g.OH * g.OW;
    for (long o = (long)blockIdx.x * RL + rl; o < Mo; o += (long)gridDim.x * RL) {
      const unsigned ou = (unsigned)o;                       // (the launcher checks N * H * W < 2^31: 32-bit divisions, a 64-bit one costs ~100 instructions)
      const unsigned r2 = ou / (unsigned)g.OW;
      const int ow = (int)(ou - r2 * (unsigned)g.OW);
      const int n = (int)(r2 / (unsigned)g.OH), oh = (int)(r2 - (unsigned)n * (unsigned)g.OH);
      // all nine loads issued before the first use (clamped addresses, validity kept aside)
      float4 v[3][3];
      bool ok[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int ih = oh * 2 - 1 + r;
        const int ihc = ih < 0 ? 0 : (ih >= g.H ? g.H - 1 : ih);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int iw = ow * 2 - 1 + s;
          const int iwc = iw < 0 ? 0 : (iw >= g.W ? g.W - 1 : iw);
          ok[r][s] = ih == ihc && iw == iwc;
          v[r][s] = *reinterpret_cast<const float4*>(x + (((size_t)n * g.H + ihc) * g.W + iwc) * C + cg * 4);
        }
      }
      float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
      int4 mt = make_int4(0, 0, 0, 0);
      bool first = true;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          if (!ok[r][s]) continue;
          const float4 a = bn_relu4(v[r][s], sc, sh);
          const int t = r * 3 + s;
          if (first || a.x > m.x) { m.x = a.x; mt.x = t; }
          if (first || a.y > m.y) { m.y = a.y; mt.y = t; }
          if (first || a.z > m.z) { m.z = a.z; mt.z = t; }
          if (first || a.w > m.w) { m.w = a.w; mt.w = t; }
          first = false;
        }
      *reinterpret_cast<float4*>(y + (size_t)o * C + cg * 4) = m;
      *reinterpret_cast<uchar4*>(tap + (size_t)o * C + cg * 4) =
          make_uchar4((unsigned char)mt.x, (unsigned char)mt.y, (unsigned char)mt.z, (unsigned char)mt.w);
    }
  }
  unsigned* cnt = stat_fwd_counters(stats, C);
  if (last_workgroup(cnt)) clear_slots_fwd(stats, C, cnt);
}

// Gradient of MaxPool2d(3, 2, 1) at input pixel m = (n, ih, iw), channels cg*4..+3.  Windows oh with oh * 2 - 1 + r == ih:
// r = (ih + 1) % 2 + 2 j, j = 0, 1 -- two candidate rows x two candidate columns, all eight loads issued unconditionally.
struct PoolTaps {
  uchar4 t[2][2];
  float4 g[2][2];
  bool ok[2][2];
  unsigned char me[2][2];
};
__device__ __forceinline__ void pool_taps_load(const float* __restrict__ gy, const unsigned char* __restrict__ tap, const StemPool& g, int C,
                                               long m, int cg, PoolTaps& p) {
  const unsigned mu = (unsigned)m;                           // (N * H * W < 2^31, checked by the launcher)
  const unsigned r2 = mu / (unsigned)g.W;
  const int iw = (int)(mu - r2 * (unsigned)g.W);
  const int n = (int)(r2 / (unsigned)g.H), ih = (int)(r2 - (unsigned)n * (unsigned)g.H);
  int ohc[2], owc[2], rc[2], sc[2];
  bool vh[2], vw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    rc[j] = (ih + 1) % 2 + j * 2;
    const int th = ih + 1 - rc[j];
    vh[j] = rc[j] < 3 && th >= 0 && th / 2 < g.OH;
    ohc[j] = vh[j] ? th / 2 : 0;
    sc[j] = (iw + 1) % 2 + j * 2;
    const int tw = iw + 1 - sc[j];
    vw[j] = sc[j] < 3 && tw >= 0 && tw / 2 < g.OW;
    owc[j] = vw[j] ? tw / 2 : 0;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const size_t o = (((size_t)n * g.OH + ohc[j]) * g.OW + owc[k]) * C + cg * 4;
      p.t[j][k] = *reinterpret_cast<const uchar4*>(tap + o);
      p.g[j][k] = *reinterpret_cast<const float4*>(gy + o);
      p.ok[j][k] = vh[j] && vw[k];
      p.me[j][k] = (unsigned char)(rc[j] * 3 + sc[k]);
    }
}
__device__ __forceinline__ float4 pool_taps_sum(const PoolTaps& p) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (p.ok[j][k] && p.t[j][k].x == p.me[j][k]) a.x += p.g[j][k].x;
      if (p.ok[j][k] && p.t[j][k].y == p.me[j][k]) a.y += p.g[j][k].y;
      if (p.ok[j][k] && p.t[j][k].z == p.me[j][k]) a.z += p.g[j][k].z;
      if (p.ok[j][k] && p.t[j][k].w == p.me[j][k]) a.w += p.g[j][k].w;
    }
  return a;
}

constexpr int kPoolUnroll = 2;   // input pixels per trip: 2 x (8 gather loads + 1 x load) in flight per lane

__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ tap,
                                                                const float* __restrict__ x, const float* __restrict__ save_mean,
                                                                const float* __restrict__ save_invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, StemPool g, int C,
                                                                float* __restrict__ red) {
  __shared__ float4 lds[2][256];
  const BnMap mp = bn_map(C);
  float4 s[kMaxNG], q[kMaxNG];
#pragma unroll
  for (int j = 0; j < kMaxNG; ++j) { s[j] = make_float4(0.f, 0.f, 0.f, 0.f); q[j] = s[j]; }
  if (mp.active) {
    const int cb = mp.cg0 * 4;
    const float4 mu = *reinterpret_cast<const float4*>(save_mean + cb), is = *reinterpret_cast<const float4*>(save_invstd + cb);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + cb), be = *reinterpret_cast<const float4*>(beta + cb);
    const float4 sc = make_float4(is.x * ga.x, is.y * ga.y, is.z * ga.z, is.w * ga.w);
    const float4 sh = make_float4(be.x - mu.x * sc.x, be.y - mu.y * sc.y, be.z - mu.z * sc.z, be.w - mu.w * sc.w);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto accum = [&](const float4& gr, const float4& v) {
      acc4(s[0], gr);
      q[0].x += gr.x * ((v.x - mu.x) * is.x); q[0].y += gr.y * ((v.y - mu.y) * is.y);
      q[0].z += gr.z * ((v.z - mu.z) * is.z); q[0].w += gr.w * ((v.w - mu.w) * is.w);
    };
    const long M = (long)g.N * g.H * g.W;
    const long stride = (long)gridDim.x * mp.RL;
    long m = (long)blockIdx.x * mp.RL + mp.rl;
    for (; m + (kPoolUnroll - 1) * stride < M; m += kPoolUnroll * stride) {
      PoolTaps p[kPoolUnroll];
      float4 v[kPoolUnroll];
#pragma unroll
      for (int u = 0; u < kPoolUnroll; ++u) {
        v[u] = *reinterpret_cast<const float4*>(x + (size_t)(m + u * stride) * C + cb);
        pool_taps_load(gy, tap, g, C, m + u * stride, mp.cg0, p[u]);
      }
#pragma unroll
      for (int u = 0; u < kPoolUnroll; ++u) accum(masked_grad(1, pool_taps_sum(p[u]), false, zero4, v[u], sc, sh), v[u]);
    }
    for (; m < M; m += stride) {
      PoolTaps p;
      const float4 v = *reinterpret_cast<const float4*>(x + (size_t)m * C + cb);
      pool_taps_load(gy, tap, g, C, m, mp.cg0, p);
      accum(masked_grad(1, pool_taps_sum(p), false, zero4, v, sc, sh), v);
    }
  }
  reduce_and_add(mp, C, s, q, lds, red);
}

// The same reduction walked over the POOLED grid (round 6): dy of the pool's backward is non-zero only at a window's winning tap, and only
// where the ReLU passed -- i.e. where the pooled output z = max relu(x sc + sh) is > 0 -- so
//   sum_m g_m = sum_o gy_o [z_o > 0],   sum_m g_m xhat_m = sum_o gy_o [z_o > 0] xhat(winner of o)
// and the winner's xhat follows from z itself: z = gamma xhat + beta  =>  xhat = (z - beta) / gamma.  The pass then reads gy and the pooled
// output (2 x 25.7 MB at B = 32) instead of x, gy and the taps of every input pixel (215 MB by the counters: 43.7 us of the step).  The
// recovered xhat carries the rounding of z (|beta| eps / |gamma|): channels whose |gamma| is below kGammaGather fetch the winner's x
// through the saved tap instead and use (x - mean) invstd like the pass above (exact; rare: batch-norm scales sit near 1).
constexpr float kGammaGather = 1e-3f;
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_y_kernel(const float* __restrict__ gy, const float* __restrict__ ypool,
                                                                  const unsigned char* __restrict__ tap, const float* __restrict__ x,
                                                                  const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, StemPool g,
                                                                  int C, float* __restrict__ red) {
  __shared__ float4 lds[2][256];
  const BnMap mp = bn_map(C);
  float4 s[kMaxNG], q[kMaxNG];
#pragma unroll
  for (int j = 0; j < kMaxNG; ++j) { s[j] = make_float4(0.f, 0.f, 0.f, 0.f); q[j] = s[j]; }
  if (mp.active) {
    const int cb = mp.cg0 * 4;
    const float4 mu = *reinterpret_cast<const float4*>(save_mean + cb), is = *reinterpret_cast<const float4*>(save_invstd + cb);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + cb), be = *reinterpret_cast<const float4*>(beta + cb);
    const bool far[4] = {fabsf(ga.x) >= kGammaGather, fabsf(ga.y) >= kGammaGather, fabsf(ga.z) >= kGammaGather, fabsf(ga.w) >= kGammaGather};
    const float rg[4] = {far[0] ? 1.0f / ga.x : 0.f, far[1] ? 1.0f / ga.y : 0.f, far[2] ? 1.0f / ga.z : 0.f, far[3] ? 1.0f / ga.w : 0.f};
    const bool all_far = far[0] && far[1] && far[2] && far[3];
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w}, bev[4] = {be.x, be.y, be.z, be.w};
    auto accum = [&](const float4& gr4, const float4& z4, long o) {
      const float gr[4] = {gr4.x, gr4.y, gr4.z, gr4.w}, z[4] = {z4.x, z4.y, z4.z, z4.w};
      float xh[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) xh[k] = (z[k] - bev[k]) * rg[k];
      if (!all_far) {                                         // (per thread: its four channels; uniform over the rows it walks)
        const unsigned ou = (unsigned)o;
        const unsigned r2 = ou / (unsigned)g.OW;
        const int ow = (int)(ou - r2 * (unsigned)g.OW);
        const int n = (int)(r2 / (unsigned)g.OH), oh = (int)(r2 - (unsigned)n * (unsigned)g.OH);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (!far[k]) {
            const int t = tap[(size_t)o * C + cb + k];
            const int ih = oh * 2 - 1 + t / 3, iw = ow * 2 - 1 + t % 3;
            xh[k] = (x[(((size_t)n * g.H + ih) * g.W + iw) * C + cb + k] - muv[k]) * isv[k];
          }
      }
      float sv[4] = {0.f, 0.f, 0.f, 0.f}, qv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (z[k] > 0.f) { sv[k] = gr[k]; qv[k] = gr[k] * xh[k]; }
      s[0].x += sv[0]; s[0].y += sv[1]; s[0].z += sv[2]; s[0].w += sv[3];
      q[0].x += qv[0]; q[0].y += qv[1]; q[0].z += qv[2]; q[0].w += qv[3];
    };
    const long Mo = (long)g.N * g.OH * g.OW;
    const long stride = (long)gridDim.x * mp.RL;
    long o = (long)blockIdx.x * mp.RL + mp.rl;
    for (; o + (kBnUnroll - 1) * stride < Mo; o += kBnUnroll * stride) {
      float4 gr[kBnUnroll], z[kBnUnroll];
#pragma unroll
      for (int u = 0; u < kBnUnroll; ++u) {
        gr[u] = *reinterpret_cast<const float4*>(gy + (size_t)(o + u * stride) * C + cb);
        z[u] = *reinterpret_cast<const float4*>(ypool + (size_t)(o + u * stride) * C + cb);
      }
#pragma unroll
      for (int u = 0; u < kBnUnroll; ++u) accum(gr[u], z[u], o + u * stride);
    }
    for (; o < Mo; o += stride)
      accum(*reinterpret_cast<const float4*>(gy + (size_t)o * C + cb), *reinterpret_cast<const float4*>(ypool + (size_t)o * C + cb), o);
  }
  reduce_and_add(mp, C, s, q, lds, red);
}

__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const float* __restrict__ gy, const unsigned char* __restrict__ tap,
                                                               const float* __restrict__ x, const float* __restrict__ save_mean,
                                                               const float* __restrict__ save_invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ red, StemPool g, int C,
                                                               float* __restrict__ dx, float* __restrict__ dgamma_acc,
                                                               float* __restrict__ dbeta_acc) {
  __shared__ float s_mg[kFuseMaxC], s_mgx[kFuseMaxC];
  const BnMap mp = bn_map(C);
  const long M = (long)g.N * g.H * g.W;
  const float invM = 1.0f / (float)M;
  for (int c = threadIdx.x; c < C; c += 256) {
    float sg, sgx;
    slot_sum2(red, C, c, sg, sgx);
    s_mg[c] = sg * invM;
    s_mgx[c] = sgx * invM;
    if (blockIdx.x == 0) {
      if (dgamma_acc) dgamma_acc[c] += sgx;
      if (dbeta_acc) dbeta_acc[c] += sg;
    }
  }
  __syncthreads();
  if (mp.active) {
    const int cb = mp.cg0 * 4;
    const float4 mu = *reinterpret_cast<const float4*>(save_mean + cb), is = *reinterpret_cast<const float4*>(save_invstd + cb);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + cb), be = *reinterpret_cast<const float4*>(beta + cb);
    const float4 k1 = make_float4(is.x * ga.x, is.y * ga.y, is.z * ga.z, is.w * ga.w);
    const float4 sh = make_float4(be.x - mu.x * k1.x, be.y - mu.y * k1.y, be.z - mu.z * k1.z, be.w - mu.w * k1.w);
    const float4 mg = *reinterpret_cast<const float4*>(&s_mg[cb]), mgx = *reinterpret_cast<const float4*>(&s_mgx[cb]);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto emit = [&](const float4& gr, const float4& v, size_t o) {
      float4 r;
      r.x = k1.x * (gr.x - mg.x - (v.x - mu.x) * is.x * mgx.x);
      r.y = k1.y * (gr.y - mg.y - (v.y - mu.y) * is.y * mgx.y);
      r.z = k1.z * (gr.z - mg.z - (v.z - mu.z) * is.z * mgx.z);
      r.w = k1.w * (gr.w - mg.w - (v.w - mu.w) * is.w * mgx.w);
      *reinterpret_cast<float4*>(dx + o) = r;
    };
    const long stride = (long)gridDim.x * mp.RL;
    long m = (long)blockIdx.x * mp.RL + mp.rl;
    for (; m + (kPoolUnroll - 1) * stride < M; m += kPoolUnroll * stride) {
      PoolTaps p[kPoolUnroll];
      float4 v[kPoolUnroll];
#pragma unroll
      for (int u = 0; u < kPoolUnroll; ++u) {
        v[u] = *reinterpret_cast<const float4*>(x + (size_t)(m + u * stride) * C + cb);
        pool_taps_load(gy, tap, g, C, m + u * stride, mp.cg0, p[u]);
      }
#pragma unroll
      for (int u = 0; u < kPoolUnroll; ++u)
        emit(masked_grad(1, pool_taps_sum(p[u]), false, zero4, v[u], k1, sh), v[u], (size_t)(m + u * stride) * C + cb);
    }
    for (; m < M; m += stride) {
      PoolTaps p;
      const float4 v = *reinterpret_cast<const float4*>(x + (size_t)m * C + cb);
      pool_taps_load(gy, tap, g, C, m, mp.cg0, p);
      emit(masked_grad(1, pool_taps_sum(p), false, zero4, v, k1, sh), v, (size_t)m * C + cb);
    }
  }
  unsigned* cnt = stat_bwd_counters(red, C);
  if (last_workgroup(cnt)) clear_slots(red, C, cnt);
}

// Apply kernels: every workgroup re-reads the 32 x 2 x C slot partials (256 C bytes, L2-resident), so the grid is also bounded by
// the tensor size: at most one workgroup per 128 rows (and at least 256 so that the GPU stays busy on small layers).
static unsigned bn_grid(long M, int C, bool fused) {
  const int C4 = C / 4;
  const int RL = C4 <= 256 ? 256 / C4 : 1;
  long blocks = (M + RL - 1) / RL;
  long cap = fused ? M / 128 : 2048;
  if (cap < 256) cap = 256;
  if (cap > 2048) cap = 2048;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}
// Reductions end with 2 * C float atomics per workgroup: keep (workgroups x C) bounded, i.e. give every workgroup enough
// rows to amortise them (EfficientNet's 1392-channel layers at 14x14 ran 2048 workgroups x 2784 atomics = 23 MB of atomics
// for a 35 MB tensor in round 1: 40 us for a reduction that moves 15 us worth of bytes).
static unsigned bn_reduce_grid(long M, int C) {
  const int C4 = C / 4;
  const int RL = C4 <= 256 ? 256 / C4 : 1;
  long blocks = (M + RL - 1) / RL;
  long cap = (1L << 17) / C;                       // <= 128 K atomically added floats per statistic per launch (swept 2^16..2^20)
  if (const char* e = getenv("HIFIHR_BN_CAP_LOG2")) cap = (1L << atoi(e)) / C;
  if (cap > 2048) cap = 2048;
  if (cap < 128) cap = 128;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}
static bool bn_c_ok(int C) { return C >= 4 && C % 4 == 0 && C <= 4 * 256 * kMaxNG; }

hipError_t launch_bn_stats(const float* x, long M, int C, float* stats, hipStream_t st) {
  if (!bn_c_ok(C)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(bn_reduce_grid(M, C)), dim3(256), 0, st, x, M, C, stats);
  return hipGetLastError();
}

hipError_t launch_bn_act_fwd(const float* x, float* stats, const float* gamma, const float* beta, const float* residual,
                             int act, long M, int C, float eps, float momentum, float* y, float* save_mean, float* save_invstd,
                             float* running_mean, float* running_var, hipStream_t st) {
  if (!bn_c_ok(C)) return hipErrorInvalidValue;
  if (C <= kFuseMaxC) {
    if (C <= 1024) hipLaunchKernelGGL((bn_act_fwd_kernel<false, 1>), dim3(bn_grid(M, C, true)), dim3(256), 0, st, x, stats, gamma, beta, residual, act, M, C, eps,
                       momentum, y, save_mean, save_invstd, running_mean, running_var);
    else hipLaunchKernelGGL((bn_act_fwd_kernel<false, kMaxNG>), dim3(bn_grid(M, C, true)), dim3(256), 0, st, x, stats, gamma, beta, residual, act, M, C, eps,
                       momentum, y, save_mean, save_invstd, running_mean, running_var);
  } else {
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, st, stats, M, C, eps, momentum, save_mean,
                       save_invstd, running_mean, running_var);
    if (C <= 1024) hipLaunchKernelGGL((bn_act_fwd_kernel<true, 1>), dim3(bn_grid(M, C, false)), dim3(256), 0, st, x, stats, gamma, beta, residual, act, M, C, eps,
                       momentum, y, save_mean, save_invstd, running_mean, running_var);
    else hipLaunchKernelGGL((bn_act_fwd_kernel<true, kMaxNG>), dim3(bn_grid(M, C, false)), dim3(256), 0, st, x, stats, gamma, beta, residual, act, M, C, eps,
                       momentum, y, save_mean, save_invstd, running_mean, running_var);
  }
  return hipGetLastError();
}

hipError_t launch_bn_finalize_fwd(float* stats, long M, int C, float eps, float momentum, float* save_mean, float* save_invstd,
                                  float* running_mean, float* running_var, hipStream_t st) {
  if (!bn_c_ok(C) || M <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, st, stats, M, C, eps, momentum, save_mean, save_invstd,
                     running_mean, running_var);
  return hipGetLastError();
}

// evaluation mode (module.eval(): running statistics, nothing updated): y = act((x - running_mean) / sqrt(running_var + eps) * gamma + beta + residual)
hipError_t launch_bn_act_eval(const float* x, const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                              const float* residual, int act, long M, int C, float eps, float* y, hipStream_t st) {
  if (!bn_c_ok(C)) return hipErrorInvalidValue;
  if (C <= 1024) hipLaunchKernelGGL((bn_act_fwd_kernel<true, 1>), dim3(bn_grid(M, C, false)), dim3(256), 0, st, x, (float*)nullptr, gamma, beta, residual, act, M, C,
                     eps, 0.f, y, const_cast<float*>(running_mean), const_cast<float*>(running_var), (float*)nullptr, (float*)nullptr);
  else hipLaunchKernelGGL((bn_act_fwd_kernel<true, kMaxNG>), dim3(bn_grid(M, C, false)), dim3(256), 0, st, x, (float*)nullptr, gamma, beta, residual, act, M, C,
                     eps, 0.f, y, const_cast<float*>(running_mean), const_cast<float*>(running_var), (float*)nullptr, (float*)nullptr);
  return hipGetLastError();
}

hipError_t launch_bn_act_bwd(const float* dy, const float* y, const float* x, const float* save_mean, const float* save_invstd,
                             const float* gamma, const float* beta, int act, long M, int C, float* red, float* dx, float* dres,
                             float* dgamma_acc, float* dbeta_acc, hipStream_t st) {
  if (!bn_c_ok(C)) return hipErrorInvalidValue;
  // red: kStatSlots slot partials + arrival counters (all zero on entry, all zero again on return)
  if (C <= 1024) hipLaunchKernelGGL((bn_bwd_reduce_kernel<1>), dim3(bn_reduce_grid(M, C)), dim3(256), 0, st, dy, y, x, save_mean, save_invstd, gamma, beta, act,
                     M, C, red);
  else hipLaunchKernelGGL((bn_bwd_reduce_kernel<kMaxNG>), dim3(bn_reduce_grid(M, C)), dim3(256), 0, st, dy, y, x, save_mean, save_invstd, gamma, beta, act,
                     M, C, red);
  if (C <= kFuseMaxC) {
    if (C <= 1024) hipLaunchKernelGGL((bn_bwd_apply_kernel<false, 1>), dim3(bn_grid(M, C, true)), dim3(256), 0, st, dy, y, x, save_mean, save_invstd, gamma, beta,
                       red, act, M, C, dx, dres, dgamma_acc, dbeta_acc);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<false, kMaxNG>), dim3(bn_grid(M, C, true)), dim3(256), 0, st, dy, y, x, save_mean, save_invstd, gamma, beta,
                       red, act, M, C, dx, dres, dgamma_acc, dbeta_acc);
  } else {
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, st, red, C, dgamma_acc, dbeta_acc);
    if (C <= 1024) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, 1>), dim3(bn_grid(M, C, false)), dim3(256), 0, st, dy, y, x, save_mean, save_invstd, gamma, beta,
                       red, act, M, C, dx, dres, dgamma_acc, dbeta_acc);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<true, kMaxNG>), dim3(bn_grid(M, C, false)), dim3(256), 0, st, dy, y, x, save_mean, save_invstd, gamma, beta,
                       red, act, M, C, dx, dres, dgamma_acc, dbeta_acc);
  }
  return hipGetLastError();
}

// The apply half alone: g is ALREADY the masked gradient and red ALREADY holds its reduction sums (written by the fused output
// transform of csrc/wino4_bn.hip); dx = gamma invstd (g - mean g - xhat mean(g xhat)), slots handed back zeroed.  C <= kFuseMaxC.
hipError_t launch_bn_bwd_apply(const float* g, const float* x, const float* save_mean, const float* save_invstd, const float* gamma, long M,
                               int C, float* red, float* dx, float* dgamma_acc, float* dbeta_acc, hipStream_t st) {
  if (!bn_c_ok(C) || C > kFuseMaxC) return hipErrorInvalidValue;
  if (C <= 1024) hipLaunchKernelGGL((bn_bwd_apply_kernel<false, 1>), dim3(bn_grid(M, C, true)), dim3(256), 0, st, g, (const float*)nullptr, x, save_mean, save_invstd,
                     gamma, (const float*)nullptr, red, 0, M, C, dx, (float*)nullptr, dgamma_acc, dbeta_acc);
  else hipLaunchKernelGGL((bn_bwd_apply_kernel<false, kMaxNG>), dim3(bn_grid(M, C, true)), dim3(256), 0, st, g, (const float*)nullptr, x, save_mean, save_invstd,
                     gamma, (const float*)nullptr, red, 0, M, C, dx, (float*)nullptr, dgamma_acc, dbeta_acc);
  return hipGetLastError();
}

static bool bn_pool_ok(int N, int H, int W, int C) {
  return N > 0 && H >= 2 && W >= 2 && C >= 4 && C % 4 == 0 && C <= kFuseMaxC && (long)N * H * W < (1L << 31);
}

bool bn_relu_maxpool_supported(int N, int H, int W, int C) { return bn_pool_ok(N, H, W, C); }

hipError_t launch_bn_relu_maxpool_fwd(const float* x, float* stats, const float* gamma, const float* beta, int N, int H, int W, int C,
                                      float eps, float momentum, float* y, unsigned char* tap, float* save_mean, float* save_invstd,
                                      float* running_mean, float* running_var, hipStream_t st) {
  if (!bn_pool_ok(N, H, W, C)) return hipErrorInvalidValue;
  const StemPool g{N, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
  hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(bn_grid((long)N * g.OH * g.OW, C, true)), dim3(256), 0, st, x, stats, gamma, beta, g, C, eps,
                     momentum, y, tap, save_mean, save_invstd, running_mean, running_var);
  return hipGetLastError();
}

hipError_t launch_bn_relu_maxpool_bwd(const float* gy, const unsigned char* tap, const float* x, const float* save_mean,
                                      const float* save_invstd, const float* gamma, const float* beta, int N, int H, int W, int C, float* red,
                                      float* dx, float* dgamma_acc, float* dbeta_acc, hipStream_t st, const float* pooled) {
  if (!bn_pool_ok(N, H, W, C)) return hipErrorInvalidValue;
  const StemPool g{N, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
  const long M = (long)N * H * W;
  // pooled != null (C <= 1024: one channel group per thread): the reduction over the pooled grid (bn_pool_bwd_reduce_y_kernel)
  if (pooled != nullptr && C <= 1024)
    hipLaunchKernelGGL(bn_pool_bwd_reduce_y_kernel, dim3(bn_reduce_grid((long)N * g.OH * g.OW, C)), dim3(256), 0, st, gy, pooled, tap, x, save_mean,
                       save_invstd, gamma, beta, g, C, red);
  else
    hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(bn_reduce_grid(M, C)), dim3(256), 0, st, gy, tap, x, save_mean, save_invstd, gamma, beta, g,
                       C, red);
  hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(bn_grid(M, C, true)), dim3(256), 0, st, gy, tap, x, save_mean, save_invstd, gamma, beta, red,
                     g, C, dx, dgamma_acc, dbeta_acc);
  return hipGetLastError();
}

}  // namespace hifihr
