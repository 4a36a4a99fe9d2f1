// Batch-norm fused into the Winograd F(4x4, 3x3) transforms that sit next to it in a ResNet BasicBlock (round 3).
//
// Replaces, for every BatchNorm2d whose consumer is a stride-1 3x3 convolution on the Winograd path (reference
// network/res_encoder.py:364-373 + the vendored BasicBlock of utils/Freihand_GNN_mano/network/resnet.py: conv1 -> bn1 -> relu -> conv2,
// and bn2 -> (+ identity) -> relu -> the next block's conv1), the separate batch-norm launches of csrc/bn.hip:
//
//   forward   wino4_bn_input_transform_kernel: reads the RAW output of the previous convolution (and the identity branch), folds the
//             batch-norm slot partials itself, applies scale / shift (+ residual) + ReLU on the fly and writes V = B^T a B directly --
//             the activation a = relu(bn(x) (+ res)) never makes its own round trip through HBM (with a residual it is also written
//             once, by the tile that owns the pixel, because the next block's identity branch reads it).
//             Round 2 ran bn_act_fwd (read x, write a) and then wino4_input_transform (read a with a 2.25x halo, write V).
//   backward  wino4_output_transform_bnred_kernel: the output transform of the backward-data product yields d loss / d a; the
//             ReLU mask (recomputed from the raw x, or read from the block output when a residual was added), the addition of the
//             identity branch's gradient and the batch-norm backward REDUCTION (sum g, sum g xhat into the float slots) happen in its
//             epilogue -- bn_bwd_reduce_kernel's pass over (g, x) is gone; it writes g = masked gradient for the apply pass.
//             wino4_bn_bwd_dual_transform_kernel: the batch-norm backward APPLY dx = gamma invstd (g - mean g - xhat mean(g xhat))
//             evaluated on the fly inside the dual input transform of the PRODUCER convolution's backward (V' = B^T dx B for its
//             backward-data, Y' = A dx A^T for its backward-weight): dx never reaches HBM either.
// Arithmetic: scale / shift / residual / ReLU in bn_act_fwd_kernel's own order (same bits as the unfused path), transforms from
// wino4_math.h.
#include <hip/hip_runtime.h>

#include "bn_fold.h"
#include "hifihr_internal.h"
#include "wino4_math.h"

namespace hifihr {

using namespace w4;

constexpr int kWbnMaxC = 512;        // scale / shift tables in LDS (every workgroup folds all C channels: 256 C bytes from L2)

__device__ __forceinline__ V4 bn_res_relu(const V4& v, const V4& sc, const V4& sh, const V4& r, bool has_res) {
  V4 z = V4{v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w};
  if (has_res) z = z + r;
  return V4{fmaxf(z.x, 0.f), fmaxf(z.y, 0.f), fmaxf(z.z, 0.f), fmaxf(z.w, 0.f)};
}

// One (tile, 4 channels) item: the 6 x 6 patch of raw values (and of the identity branch) in registers.
template <bool RES>
struct WbnPatch {
  V4 v[6][6];
  V4 r[RES ? 6 : 1][RES ? 6 : 1];
  unsigned long long ok;                                    // bit 6 * row + column: the pixel is inside the image
  int n, th, tw, cg;
  size_t t;
};

template <bool RES>
__device__ __forceinline__ void wbn_load(WbnPatch<RES>& p, size_t i, const float* __restrict__ x, const float* __restrict__ res, int H, int W,
                                         int C, int TH, int TW) {
  const int C4 = C / 4;
  p.cg = (int)(i % C4);
  p.t = i / C4;
  p.tw = (int)(p.t % TW); p.th = (int)((p.t / TW) % TH); p.n = (int)(p.t / ((size_t)TW * TH));
  p.ok = 0ull;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const int iw = 4 * p.tw - 1 + c;
    const bool cok = iw >= 0 && iw < W;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int ih = 4 * p.th - 1 + r;
      const bool ok = cok && ih >= 0 && ih < H;
      const size_t o = (((size_t)p.n * H + (ok ? ih : 0)) * W + (ok ? iw : 0)) * C + p.cg * 4;
      p.v[r][c] = ld4(x + o);
      if constexpr (RES) p.r[r][c] = ld4(res + o);
      p.ok |= ok ? (1ull << (6 * r + c)) : 0ull;               // (r, c are compile-time: folds into per-position predicates)
    }
  }
}

template <bool RES>
__device__ __forceinline__ void wbn_emit(const WbnPatch<RES>& p, const float* s_sc, const float* s_sh, float* __restrict__ out,
                                         float* __restrict__ V, int H, int W, int C, size_t T) {
  const V4 sc = ld4(&s_sc[p.cg * 4]), sh = ld4(&s_sh[p.cg * 4]);
  V4 tt[6][6];                                              // tt = B^T a, built column by column
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    V4 col[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const bool ok = (p.ok >> (6 * r + c)) & 1ull;
      const V4 a = bn_res_relu(p.v[r][c], sc, sh, RES ? p.r[RES ? r : 0][RES ? c : 0] : zero4(), RES);
      col[r] = ok ? a : zero4();                            // the convolution pads the ACTIVATION with zeros
      if (RES && r >= 1 && r <= 4 && c >= 1 && c <= 4 && ok)                // this tile's own 4 x 4 pixels: the block output
        st4(out + (((size_t)p.n * H + (4 * p.th - 1 + r)) * W + (4 * p.tw - 1 + c)) * C + p.cg * 4, a);
    }
    V4 o[6];
    bt6(col, o);
#pragma unroll
    for (int r = 0; r < 6; ++r) tt[r][c] = o[r];
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    V4 o[6];
    bt6(tt[r], o);
#pragma unroll
    for (int c = 0; c < 6; ++c) st4(V + ((size_t)(r * 6 + c) * T + p.t) * C + p.cg * 4, o[c]);
  }
}

// thread = (tile, 4 channels).  x: raw convolution output [N][H][W][C]; res / out (RES): identity branch and block output.
// These launches are small (one item per thread, 130-250 workgroups on layers 2-4) and latency-bound: 16-25 us against 8-14 for the plain
// transform plus 9-12 for the bn_act_fwd launch it absorbs.  (Requesting the thread's whole patch BEFORE the slot fold, so that the
// fold's L2 round trip runs under the patch's latency, measured slower: 19.4 / 23.2 us against 17.4 / 22.4 -- 228 registers.)
template <bool RES>
__global__ __launch_bounds__(256) void wino4_bn_input_transform_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                         const float* __restrict__ res, float* __restrict__ out,
                                                                         float* __restrict__ V, int N, int H, int W, int C, int TH, int TW,
                                                                         float eps, float momentum, float* __restrict__ save_mean,
                                                                         float* __restrict__ save_invstd, float* __restrict__ running_mean,
                                                                         float* __restrict__ running_var) {
  __shared__ float s_sc[kWbnMaxC], s_sh[kWbnMaxC];
  const long M = (long)N * H * W;
  const size_t T = (size_t)N * TH * TW, total = T * (C / 4);
  for (int c = threadIdx.x; c < C; c += 256) {            // every workgroup folds the slot partials itself (L2-resident)
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
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    WbnPatch<RES> p;
    wbn_load<RES>(p, i, x, res, H, W, C, TH, TW);
    wbn_emit<RES>(p, s_sc, s_sh, out, V, H, W, C, T);
  }
  unsigned* cnt = stat_fwd_counters(stats, C);
  if (last_workgroup(cnt)) clear_slots_fwd(stats, C, cnt);
}

static unsigned wbn_grid(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

bool wino4_bn_supported(int C) { return C >= 4 && C % 4 == 0 && C <= kWbnMaxC; }

hipError_t launch_wino4_bn_input_transform(const float* x, float* stats, const float* gamma, const float* beta, const float* res, float* out,
                                           float* V, int N, int H, int W, int C, float eps, float momentum, float* save_mean,
                                           float* save_invstd, float* running_mean, float* running_var, hipStream_t st) {
  if (!wino4_bn_supported(C) || ((res == nullptr) != (out == nullptr))) return hipErrorInvalidValue;
  const int TH = (H + 3) / 4, TW = (W + 3) / 4;
  const size_t total = (size_t)N * TH * TW * (C / 4);
  if (res != nullptr)
    hipLaunchKernelGGL((wino4_bn_input_transform_kernel<true>), dim3(wbn_grid(total)), dim3(256), 0, st, x, stats, gamma, beta, res, out, V, N, H,
                       W, C, TH, TW, eps, momentum, save_mean, save_invstd, running_mean, running_var);
  else
    hipLaunchKernelGGL((wino4_bn_input_transform_kernel<false>), dim3(wbn_grid(total)), dim3(256), 0, st, x, stats, gamma, beta, res, out, V, N, H,
                       W, C, TH, TW, eps, momentum, save_mean, save_invstd, running_mean, running_var);
  return hipGetLastError();
}

}  // namespace hifihr
