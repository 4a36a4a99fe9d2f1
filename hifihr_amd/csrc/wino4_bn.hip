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

#include <cstdlib>

#include "bn_fold.h"
#include "hifihr_internal.h"
#include "wino4_math.h"

namespace hifihr {

using namespace w4;

TileGeo wino4_geo(int N, int H, int W);                    // csrc/wino4.hip: plain tiles, or 16 images per mosaic

constexpr int kWbnMaxC = 512;        // scale / shift tables in LDS (every workgroup folds all C channels: 256 C bytes from L2)

__device__ __forceinline__ V4 relu0(const V4& z) { return V4{fmaxf(z.x, 0.f), fmaxf(z.y, 0.f), fmaxf(z.z, 0.f), fmaxf(z.w, 0.f)}; }
__device__ __forceinline__ float relu0(float z) { return fmaxf(z, 0.f); }

template <typename T>
__device__ __forceinline__ T bn_res_relu(const T& v, const T& sc, const T& sh, const T& r, bool has_res) {
  T z = v * sc + sh;
  if (has_res) z = z + r;
  return relu0(z);
}

// One (tile, VW channels) item: the 6 x 6 patch of raw values (and of the identity branch) in registers.  T = V4: four consecutive
// channels per thread; T = float: one -- four times the threads with a quarter of the serial work each (round 4: these launches put
// 784 waves on the chip's 1 024 SIMDs and every wave spends half its life waiting for its 36-72 loads: profiles/r04_pmc_wino_bn.txt).
template <bool RES, typename T>
struct WbnPatch {
  T v[6][6];
  T r[RES ? 6 : 1][RES ? 6 : 1];
  unsigned long long ok;                                    // bit 6 * row + column: the pixel is inside an image
  int cg;
  size_t t;
  size_t own[4][4];                                         // RES: pixel index of the tile's own 4 x 4 pixels (the block output is written there)
};

template <bool RES, typename T>
__device__ __forceinline__ void wbn_load(WbnPatch<RES, T>& p, size_t i, const float* __restrict__ x, const float* __restrict__ res,
                                         const TileGeo& geo, int C) {
  constexpr int VW = VecWidth<T>::n;
  const int CV = C / VW;
  p.cg = (int)(i % CV);
  p.t = i / CV;
  const TileAt at = tile_at(geo, p.t);
    const AxisBase by_ = axis_base(geo, at.y0, geo.H), bx_ = axis_base(geo, at.x0, geo.W);
  AxisPx ry[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) ry[r] = row_px(geo, at, by_, r - 1);
  p.ok = 0ull;
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    const AxisPx cx = col_px(geo, bx_, c - 1);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      bool ok;
      const size_t px = tile_pixel(ry[r], cx, ok);
      const size_t o = px * C + p.cg * VW;
      p.v[r][c] = ldT<T>(x + o);
      if constexpr (RES) {
        p.r[r][c] = ldT<T>(res + o);
        if (r >= 1 && r <= 4 && c >= 1 && c <= 4) p.own[r - 1][c - 1] = px;
      }
      p.ok |= ok ? (1ull << (6 * r + c)) : 0ull;               // (r, c are compile-time: folds into per-position predicates)
    }
  }
}

template <bool RES, typename T>
__device__ __forceinline__ void wbn_emit(const WbnPatch<RES, T>& p, const float* s_sc, const float* s_sh, float* __restrict__ out,
                                         float* __restrict__ V, int C, size_t Tn) {
  constexpr int VW = VecWidth<T>::n;
  const T sc = ldT<T>(&s_sc[p.cg * VW]), sh = ldT<T>(&s_sh[p.cg * VW]);
  T tt[6][6];                                               // tt = B^T a, built column by column
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    T col[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const bool ok = (p.ok >> (6 * r + c)) & 1ull;
      const T a = bn_res_relu<T>(p.v[r][c], sc, sh, RES ? p.r[RES ? r : 0][RES ? c : 0] : zeroT<T>(), RES);
      col[r] = ok ? a : zeroT<T>();                         // the convolution pads the ACTIVATION with zeros
      if (RES && r >= 1 && r <= 4 && c >= 1 && c <= 4 && ok)                // this tile's own 4 x 4 pixels: the block output
        stT(out + p.own[RES ? r - 1 : 0][RES ? c - 1 : 0] * C + p.cg * VW, a);
    }
    T o[6];
    bt6(col, o);
#pragma unroll
    for (int r = 0; r < 6; ++r) tt[r][c] = o[r];
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    T o[6];
    bt6(tt[r], o);
#pragma unroll
    for (int c = 0; c < 6; ++c) stT(V + ((size_t)(r * 6 + c) * Tn + p.t) * C + p.cg * VW, o[c]);
  }
}

// thread = (tile, VW channels).  x: raw convolution output [N][H][W][C]; res / out (RES): identity branch and block output.
// These launches are small (one item per thread, 130-250 workgroups on layers 2-4 at four channels per thread) and latency-bound: 16-25 us
// against 8-14 for the plain transform plus 9-12 for the bn_act_fwd launch it absorbs.  (Requesting the thread's whole patch BEFORE the
// slot fold, so that the fold's L2 round trip runs under the patch's latency, measured slower: 19.4 / 23.2 us against 17.4 / 22.4 -- 228
// registers.)
template <bool RES, typename T, bool MOS>
__global__ __launch_bounds__(256) void wino4_bn_input_transform_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                         const float* __restrict__ res, float* __restrict__ out,
                                                                         float* __restrict__ V, TileGeo geo, int C,
                                                                         float eps, float momentum, float* __restrict__ save_mean,
                                                                         float* __restrict__ save_invstd, float* __restrict__ running_mean,
                                                                         float* __restrict__ running_var) {
  if (!MOS) geo.G = 0;                                     // (compile-time in the plain form: csrc/wino4.hip wino4_input_transform_kernel)
  __shared__ float s_sc[kWbnMaxC], s_sh[kWbnMaxC];
  const long M = (long)geo.N * geo.H * geo.W;
  const size_t Tn = tile_count(geo), total = Tn * (C / VecWidth<T>::n);
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
    WbnPatch<RES, T> p;
    wbn_load<RES, T>(p, i, x, res, geo, C);
    wbn_emit<RES, T>(p, s_sc, s_sh, out, V, C, Tn);
  }
  unsigned* cnt = stat_fwd_counters(stats, C);
  if (last_workgroup(cnt)) clear_slots_fwd(stats, C, cnt);
}

// ------------------------------------------------------------------------------------------------
// backward, part 1: output transform of the backward-data product + ReLU mask (+ the identity branch's gradient) + the batch-norm
// backward reduction.  workgroup = 16 tile lanes x 16 float4 channel lanes (64 channels, blockIdx.y); Mm[36][T][C] -> g[N][H][W][C],
// red[kStatSlots][2][C] += (sum g, sum g xhat).  RES: the mask is (block output > 0); else z = x sc + sh > 0 recomputed by the forward's
// expression (same bits: csrc/bn.hip masked_grad).  ADD: gadd = gradient that reached the block output through the identity branch.
// ------------------------------------------------------------------------------------------------
template <bool RES, bool ADD, bool MOS>
__global__ __launch_bounds__(256) void wino4_output_transform_bnred_kernel(const float* __restrict__ Mm, const float* __restrict__ x,
                                                                          const float* __restrict__ outp, const float* __restrict__ gadd,
                                                                          const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                          float* __restrict__ red, float* __restrict__ g, TileGeo geo, int C) {
  if (!MOS) geo.G = 0;
  __shared__ float4 redl[2][16][16];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int k = blockIdx.y * 64 + cl * 4;
  const bool kok = k < C;
  const size_t T = tile_count(geo);
  V4 sg = zero4(), sq = zero4();
  if (kok) {
    const V4 mu = ld4(save_mean + k), is = ld4(save_invstd + k), ga = ld4(gamma + k), be = ld4(beta + k);
    const V4 sc = V4{is.x * ga.x, is.y * ga.y, is.z * ga.z, is.w * ga.w};
    const V4 sh = V4{be.x - mu.x * sc.x, be.y - mu.y * sc.y, be.z - mu.z * sc.z, be.w - mu.w * sc.w};
    for (size_t t = (size_t)blockIdx.x * 16 + tl; t < T; t += (size_t)gridDim.x * 16) {
      const TileAt at = tile_at(geo, t);
    const AxisBase by_ = axis_base(geo, at.y0, geo.H), bx_ = axis_base(geo, at.x0, geo.W);
      AxisPx cx[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) cx[b] = col_px(geo, bx_, b);
      V4 s[4][6];                                           // s = A^T m, built column by column
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        V4 col[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) col[r] = ld4(Mm + ((size_t)(r * 6 + c) * T + t) * C + k);
        V4 o[4];
        at6(col, o);
#pragma unroll
        for (int a = 0; a < 4; ++a) s[a][c] = o[a];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const AxisPx ry = row_px(geo, at, by_, a);
        if (ry.ok) {                                        // (the row's operands are requested before its transform)
          V4 xv[4], ov[4], av[4];
          bool ok[4];
          size_t ob[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            ob[b] = tile_pixel(ry, cx[b], ok[b]) * C + k;
            const size_t o = ob[b];
            xv[b] = ld4(x + o);
            ov[b] = RES ? ld4(outp + o) : zero4();
            av[b] = ADD ? ld4(gadd + o) : zero4();
          }
          V4 o[4];
          at6(s[a], o);
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            if (ok[b]) {
              V4 d = o[b];
              if (ADD) d = d + av[b];
              V4 z;
              if (RES) z = ov[b];
              else z = V4{xv[b].x * sc.x + sh.x, xv[b].y * sc.y + sh.y, xv[b].z * sc.z + sh.z, xv[b].w * sc.w + sh.w};
              const V4 gm = V4{z.x > 0.f ? d.x : 0.f, z.y > 0.f ? d.y : 0.f, z.z > 0.f ? d.z : 0.f, z.w > 0.f ? d.w : 0.f};
              st4(g + ob[b], gm);
              sg = sg + gm;
              sq.x += gm.x * ((xv[b].x - mu.x) * is.x); sq.y += gm.y * ((xv[b].y - mu.y) * is.y);
              sq.z += gm.z * ((xv[b].z - mu.z) * is.z); sq.w += gm.w * ((xv[b].w - mu.w) * is.w);
            }
          }
        }
      }
    }
  }
  redl[0][tl][cl] = make_float4(sg.x, sg.y, sg.z, sg.w); redl[1][tl][cl] = make_float4(sq.x, sq.y, sq.z, sq.w);
  __syncthreads();
  if (tl == 0 && kok) {
    float4 a1 = redl[0][0][cl], a2 = redl[1][0][cl];
    for (int r = 1; r < 16; ++r) {
      const float4 b1 = redl[0][r][cl], b2 = redl[1][r][cl];
      a1 = make_float4(a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w);
      a2 = make_float4(a2.x + b2.x, a2.y + b2.y, a2.z + b2.z, a2.w + b2.w);
    }
    float* sp = red + (size_t)(blockIdx.x & (stat_slots_used(C) - 1)) * 2 * C;
    atomicAdd(sp + k, a1.x); atomicAdd(sp + k + 1, a1.y); atomicAdd(sp + k + 2, a1.z); atomicAdd(sp + k + 3, a1.w);
    atomicAdd(sp + C + k, a2.x); atomicAdd(sp + C + k + 1, a2.y); atomicAdd(sp + C + k + 2, a2.z); atomicAdd(sp + C + k + 3, a2.w);
  }
}

// ------------------------------------------------------------------------------------------------
// backward, part 2: the batch-norm backward APPLY inside the dual input transform of the producer convolution's backward.
// thread = (tile, 4 channels).  g = masked gradient [N][H][W][K], y = the convolution's raw output (the batch-norm's input), red = the
// reduction sums part 1 left in the float slots (folded here; workgroup 0 accumulates dgamma / dbeta; the last workgroup to finish hands
// the slots back zeroed).  dy = gamma invstd (g - mean g - xhat mean(g xhat)) -- bn_bwd_apply_kernel's expression -- goes straight into
// V' = B^T dy B (backward-data) and Y' = A dy A^T (backward-weight).
// ------------------------------------------------------------------------------------------------
template <typename T, bool MOS>
__global__ __launch_bounds__(256) void wino4_bn_bwd_dual_transform_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                                         const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                                         const float* __restrict__ gamma, float* __restrict__ red,
                                                                         float* __restrict__ V, float* __restrict__ Y, TileGeo geo, int K,
                                                                         float* __restrict__ dgamma_acc, float* __restrict__ dbeta_acc) {
  // T = V4: thread = (tile, 4 channels); T = float: thread = (tile, channel) -- four times the threads with a quarter of the serial
  // work each (these launches are latency-bound: 130-250 workgroups of 36 loads + 72 stores per lane at T = V4)
  constexpr int VW = VecWidth<T>::n;
  if (!MOS) geo.G = 0;
  __shared__ float s_mg[kWbnMaxC], s_mgx[kWbnMaxC];
  const float invM = 1.0f / (float)((long)geo.N * geo.H * geo.W);
  for (int c = threadIdx.x; c < K; c += 256) {
    float sgv, sgx;
    slot_sum2(red, K, c, sgv, sgx);                        // (fold ~2 us + the election below ~1.5 us of a 16-28 us launch: HIFIHR_DBG_NOFOLD experiment, round 3)
    s_mg[c] = sgv * invM;
    s_mgx[c] = sgx * invM;
    if (blockIdx.x == 0) {
      if (dgamma_acc) dgamma_acc[c] += sgx;
      if (dbeta_acc) dbeta_acc[c] += sgv;
    }
  }
  __syncthreads();
  const int KV = K / VW;
  const size_t Tn = tile_count(geo), total = Tn * KV;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % KV);
    const size_t t = i / KV;
    const TileAt at = tile_at(geo, t);
    const AxisBase by_ = axis_base(geo, at.y0, geo.H), bx_ = axis_base(geo, at.x0, geo.W);
    AxisPx ry[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) ry[r] = row_px(geo, at, by_, r - 1);
    const T mu = ldT<T>(save_mean + cg * VW), is = ldT<T>(save_invstd + cg * VW), ga = ldT<T>(gamma + cg * VW);
    const T k1 = is * ga;
    const T mg = ldT<T>(&s_mg[cg * VW]), mgx = ldT<T>(&s_mgx[cg * VW]);
    T tt[6][6];                                             // tt = B^T d, built column by column
    T ty[6][4];                                             // A dy (6 x 4) of the central block
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const AxisPx cx = col_px(geo, bx_, c - 1);
      T col[6], yv[6];
      bool okr[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const size_t o = tile_pixel(ry[r], cx, okr[r]) * K + cg * VW;
        col[r] = ldT<T>(g + o);
        yv[r] = ldT<T>(y + o);
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const T d = k1 * (col[r] - mg - (yv[r] - mu) * is * mgx);          // bn_bwd_apply_kernel's expression, operation for operation
        col[r] = okr[r] ? d : zeroT<T>();
      }
      T o[6];
      bt6(col, o);
#pragma unroll
      for (int r = 0; r < 6; ++r) tt[r][c] = o[r];
      if (c >= 1 && c <= 4) {
        T o2[6];
        a4(col + 1, o2);
#pragma unroll
        for (int r = 0; r < 6; ++r) ty[r][c - 1] = o2[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      T o[6];
      bt6(tt[r], o);
#pragma unroll
      for (int c = 0; c < 6; ++c) stT(V + ((size_t)(r * 6 + c) * Tn + t) * K + cg * VW, o[c]);
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      T o[6];
      a4(ty[r], o);
#pragma unroll
      for (int c = 0; c < 6; ++c) stT(Y + ((size_t)(r * 6 + c) * Tn + t) * K + cg * VW, o[c]);
    }
  }
  unsigned* cnt = stat_bwd_counters(red, K);
  if (last_workgroup(cnt)) clear_slots(red, K, cnt);
}

static unsigned wbn_grid(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// one channel per thread instead of four while the four-channel form would not even give every SIMD two waves (HIFIHR_WINO_VEC=4 / 1
// force a form: A/B)
static bool wbn_scalar(size_t items_v4) {
  static const int force = [] { const char* e = getenv("HIFIHR_WINO_VEC"); return e ? atoi(e) : 0; }();
  if (force == 1) return true;
  if (force == 4) return false;
  return items_v4 < (size_t)256 * 2048;
}

bool wino4_bn_supported(int C) { return C >= 4 && C % 4 == 0 && C <= kWbnMaxC; }

hipError_t launch_wino4_bn_input_transform(const float* x, float* stats, const float* gamma, const float* beta, const float* res, float* out,
                                           float* V, int N, int H, int W, int C, float eps, float momentum, float* save_mean,
                                           float* save_invstd, float* running_mean, float* running_var, hipStream_t st) {
  if (!wino4_bn_supported(C) || ((res == nullptr) != (out == nullptr))) return hipErrorInvalidValue;
  const TileGeo geo = wino4_geo(N, H, W);
  const size_t total = tile_count(geo) * (C / 4);
  static const int in_scalar = [] { const char* e = getenv("HIFIHR_WINO_IN_VEC"); return e ? atoi(e) : 0; }();   // 1: one channel per thread; 4: four (A/B)
  const bool scalar = in_scalar == 1;      // (measured: no gain -- 18.3 / 13.6 / 19.6 us against 16.2 / 13.1 / 17.2 at four channels per thread)
#define HIFIHR_WBN_IN2(R_, T_, M_, TOT_)                                                                                                       \
  hipLaunchKernelGGL((wino4_bn_input_transform_kernel<R_, T_, M_>), dim3(wbn_grid(TOT_)), dim3(256), 0, st, x, stats, gamma, beta, res, out, V, \
                     geo, C, eps, momentum, save_mean, save_invstd, running_mean, running_var)
#define HIFIHR_WBN_IN(R_, T_, TOT_) { if (geo.G) HIFIHR_WBN_IN2(R_, T_, true, TOT_); else HIFIHR_WBN_IN2(R_, T_, false, TOT_); }
  if (res != nullptr) { if (scalar) HIFIHR_WBN_IN(true, float, total * 4) else HIFIHR_WBN_IN(true, V4, total) }
  else { if (scalar) HIFIHR_WBN_IN(false, float, total * 4) else HIFIHR_WBN_IN(false, V4, total) }
#undef HIFIHR_WBN_IN
#undef HIFIHR_WBN_IN2
  return hipGetLastError();
}

hipError_t launch_wino4_output_transform_bnred(const float* Mm, const float* x, const float* outp, const float* gadd, const float* save_mean,
                                               const float* save_invstd, const float* gamma, const float* beta, float* red, float* g, int N,
                                               int H, int W, int C, hipStream_t st) {
  if (!wino4_bn_supported(C)) return hipErrorInvalidValue;
  const TileGeo geo = wino4_geo(N, H, W);
  const size_t T = tile_count(geo);
  size_t bx = (T + 15) / 16;
  if (bx > 1024) bx = 1024;
  const dim3 grid((unsigned)bx, (C + 63) / 64);
#define HIFIHR_BNRED2(R_, A_, M_)                                                                                                          \
  hipLaunchKernelGGL((wino4_output_transform_bnred_kernel<R_, A_, M_>), grid, dim3(256), 0, st, Mm, x, outp, gadd, save_mean, save_invstd, gamma, \
                     beta, red, g, geo, C)
#define HIFIHR_BNRED(R_, A_) { if (geo.G) HIFIHR_BNRED2(R_, A_, true); else HIFIHR_BNRED2(R_, A_, false); }
  if (outp != nullptr) {
    if (gadd != nullptr) HIFIHR_BNRED(true, true) else HIFIHR_BNRED(true, false)
  } else {
    if (gadd != nullptr) HIFIHR_BNRED(false, true) else HIFIHR_BNRED(false, false)
  }
#undef HIFIHR_BNRED
#undef HIFIHR_BNRED2
  return hipGetLastError();
}

hipError_t launch_wino4_bn_bwd_dual_transform(const float* g, const float* y, const float* save_mean, const float* save_invstd, const float* gamma,
                                              float* red, float* V, float* Y, int N, int H, int W, int K, float* dgamma_acc, float* dbeta_acc,
                                              hipStream_t st) {
  if (!wino4_bn_supported(K)) return hipErrorInvalidValue;
  const TileGeo geo = wino4_geo(N, H, W);
  const size_t total = tile_count(geo) * (K / 4);
#define HIFIHR_WBN_DUAL(T_, M_, TOT_)                                                                                                       \
  hipLaunchKernelGGL((wino4_bn_bwd_dual_transform_kernel<T_, M_>), dim3(wbn_grid(TOT_)), dim3(256), 0, st, g, y, save_mean, save_invstd, gamma, \
                     red, V, Y, geo, K, dgamma_acc, dbeta_acc)
  if (wbn_scalar(total)) { if (geo.G) HIFIHR_WBN_DUAL(float, true, total * 4); else HIFIHR_WBN_DUAL(float, false, total * 4); }
  else { if (geo.G) HIFIHR_WBN_DUAL(V4, true, total); else HIFIHR_WBN_DUAL(V4, false, total); }
#undef HIFIHR_WBN_DUAL
  return hipGetLastError();
}

}  // namespace hifihr

