// Depthwise convolution (groups == channels) on NHWC activations, fp32: forward, backward-data, backward-weight.
//
// Replaces the depthwise `Conv2dStaticSamePadding(groups=oup)` of the reference's EfficientNet MBConv blocks (reference
// network/efficientnet_pt/model.py:49-55,80; utils.py:122-145) -- k = 3 or 5, stride 1 or 2, TensorFlow-style
// asymmetric zero padding (top/left given explicitly; bottom/right implied by the output size).  0.2 % of the network's
// FLOPs but, unfused in MIOpen's fp32 NHWC path, >90 % of its time on this GPU.  HBM/L1-bound streaming kernels:
//
//   workgroup = 64 channels (16 float4 lanes) x 16 pixel lanes; blockIdx.y = channel block, whose k*k x 64 weights sit in
//   LDS transposed to [tap][channel] so every tap is one float4 read.  A thread produces FOUR horizontally adjacent
//   outputs per step: the 3*stride + k input columns of a row are loaded once (unconditionally, clamped, masked) and
//   feed all four -- 10 loads per output at k = 5 instead of 25, none of them inside a branch (round 1's first
//   version ran 3-7x off the bandwidth bound on exactly those two points).
//   forward   optionally adds the per-channel sum / sum of squares of y to the batch-norm slot buffer (bn.hip) from the
//             registers, which removes a full read of y by bn_stats_kernel per MBConv block.
//   bwd-data  the same walk over input pixels; stride 2 contributes only taps of matching parity (resolved at compile
//             time per parity of the first column).
//   bwd-weight register accumulators [k*k] x float4 per thread over a grid-stride loop, folded over the pixel lanes
//             through LDS, then one atomic per (channel, tap) per workgroup (<= 128 workgroups per channel block).
// Weight layout [C][k][k] (= torch's [C,1,k,k]).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "hifihr_internal.h"

namespace hifihr {

constexpr int kPW = 4;        // outputs per thread per step (horizontally adjacent)

// kRowAtATime (k = 5, forward and backward-data): the loop over the window's rows is NOT unrolled.  Unrolled, the scheduler hoists the
// loads of all k rows in front of the first multiply-add (40 float4 at k = 5): 290-366 registers, ONE 256-thread workgroup per CU, and
// the ~900 workgroups of a 7 x 7 or 14 x 14 layer run as 3-4 rounds that each pay the whole load latency (tools/time_dwconv.py: 0.17-0.30
// of a 5 TB/s stream).  One row of loads in flight per thread at 104-122 registers: 816 channels at 14 x 14 67 -> 39 us forward, 44 -> 25
// backward-data; per EfficientNet-b3 step at batch 48 forward 1 186 -> ~960 us, backward-data 975 -> ~760.  k = 3 keeps the unrolled
// form (102-183 registers unrolled; row at a time it lost on the 112 x 112 layers: 73 -> 95 us), and so does backward-weight (its 25
// accumulators need static indices; the row-at-a-time form with a uniform branch per candidate row measured 1 253 -> 1 345 us per step).
// (An `asm volatile("" ::: "memory")` between the rows does not stop the hoisting: the loads are from const __restrict__ pointers.)

__device__ __forceinline__ float4 fma4(const float4& a, const float4& b, const float4& c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// weights of this workgroup's 64 channels -> LDS [k*k][64]
template <int KK>
__device__ __forceinline__ void stage_weights(const float* __restrict__ w, int C, int c0, float (*wl)[64]) {
  for (int e = threadIdx.x; e < KK * 64; e += 256) {
    const int c = e / KK, t = e - c * KK;
    wl[t][c] = (c0 + c < C) ? w[(size_t)(c0 + c) * KK + t] : 0.f;
  }
}

// one input row segment: NC float4 columns starting at column iw0, zero outside [0, W) or when the row is invalid
template <int NC>
__device__ __forceinline__ void load_row(const float* __restrict__ base, bool rowok, int iw0, int W, int C, float4 (&v)[NC]) {
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int iw = iw0 + j;
    const bool ok = rowok && iw >= 0 && iw < W;
    const float4 t = *reinterpret_cast<const float4*>(base + (size_t)(ok ? iw : 0) * C);     // unconditional, clamped
    v[j] = ok ? t : zero4();
  }
}

// PRE (round 5): the kernel's input is the RAW output e of the block's expand convolution and the batch-norm + swish that precede the
// depthwise convolution in an MBConv block (reference network/efficientnet_pt/model.py:73-80: x = swish(bn0(expand_conv(x))); x =
// depthwise_conv(x)) are applied as the rows are loaded: a = swish(e * sc + sh), zero where the window leaves the image (the padding
// is of a, not of e).  The activated tensor -- six times the block's input, the largest tensor of the network -- is then neither written
// by a batch-norm launch nor read back here.  sc / sh: this thread's four channels, from the layer's (mean, invstd, gamma, beta).
struct DwPre {
  const float *mean, *invstd, *gamma, *beta;
};
__device__ __forceinline__ void dw_pre_affine(const DwPre& p, int c, bool cok, float4& sc, float4& sh) {
  sc = zero4(); sh = zero4();
  if (!cok) return;
  const float4 mu = *reinterpret_cast<const float4*>(p.mean + c), is = *reinterpret_cast<const float4*>(p.invstd + c);
  const float4 ga = *reinterpret_cast<const float4*>(p.gamma + c), be = *reinterpret_cast<const float4*>(p.beta + c);
  sc = make_float4(is.x * ga.x, is.y * ga.y, is.z * ga.z, is.w * ga.w);          // (the expressions of bn_act_fwd_kernel: same bits)
  sh = make_float4(be.x - mu.x * sc.x, be.y - mu.y * sc.y, be.z - mu.z * sc.z, be.w - mu.w * sc.w);
}
__device__ __forceinline__ float4 dw_swish_affine(const float4& t, const float4& sc, const float4& sh) {
  float4 r = make_float4(t.x * sc.x + sh.x, t.y * sc.y + sh.y, t.z * sc.z + sh.z, t.w * sc.w + sh.w);
  r.x = r.x * fast_sigmoid(r.x); r.y = r.y * fast_sigmoid(r.y); r.z = r.z * fast_sigmoid(r.z); r.w = r.w * fast_sigmoid(r.w);
  return r;
}
template <int NC>
__device__ __forceinline__ void load_row_pre(const float* __restrict__ base, bool rowok, int iw0, int W, int C, const float4& sc,
                                             const float4& sh, float4 (&v)[NC]) {
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int iw = iw0 + j;
    const bool ok = rowok && iw >= 0 && iw < W;
    const float4 t = *reinterpret_cast<const float4*>(base + (size_t)(ok ? iw : 0) * C);
    v[j] = ok ? dw_swish_affine(t, sc, sh) : zero4();
  }
}

template <int K, int S, bool PRE>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(DwGeom g, const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y, float* __restrict__ stats, DwPre pre) {
  constexpr int KK = K * K, NC = (kPW - 1) * S + K;
  __shared__ float wl[KK][64];
  __shared__ double red[2][16][16][4];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c0 = blockIdx.y * 64, c = c0 + cl * 4;
  const bool cok = c < g.C;
  stage_weights<KK>(w, g.C, c0, wl);
  float4 psc = zero4(), psh = zero4();
  if constexpr (PRE) dw_pre_affine(pre, c, cok, psc, psh);
  __syncthreads();
  const int OWB = (g.OW + kPW - 1) / kPW;
  const long nb = (long)g.N * g.OH * OWB;
  Stat4 st;                                         // batch-norm statistics of y (hifihr_internal.h "FORWARD statistics")
  if (cok) {
    for (long pb = (long)blockIdx.x * 16 + pl; pb < nb; pb += (long)gridDim.x * 16) {
      int owb, oh, n;
      if (g.vorder) { oh = (int)(pb % g.OH); const long q = pb / g.OH; owb = (int)(q % OWB); n = (int)(q / OWB); }
      else { owb = (int)(pb % OWB); const long q = pb / OWB; oh = (int)(q % g.OH); n = (int)(q / g.OH); }
      const int ow0 = owb * kPW;
      float4 acc[kPW];
#pragma unroll
      for (int p = 0; p < kPW; ++p) acc[p] = zero4();
      auto row = [&](int r) {
        const int ih = oh * S - g.pt + r;
        const bool rowok = ih >= 0 && ih < g.H;
        float4 v[NC];
        if constexpr (PRE) load_row_pre<NC>(x + ((size_t)n * g.H + (rowok ? ih : 0)) * g.W * g.C + c, rowok, ow0 * S - g.pl, g.W, g.C, psc, psh, v);
        else load_row<NC>(x + ((size_t)n * g.H + (rowok ? ih : 0)) * g.W * g.C + c, rowok, ow0 * S - g.pl, g.W, g.C, v);
#pragma unroll
        for (int s = 0; s < K; ++s) {
          const float4 wt = *reinterpret_cast<const float4*>(&wl[r * K + s][cl * 4]);
#pragma unroll
          for (int p = 0; p < kPW; ++p) acc[p] = fma4(v[p * S + s], wt, acc[p]);
        }
      };
      if constexpr (K >= 5) {                         // (see kRowAtATime)
#pragma unroll 1
        for (int r = 0; r < K; ++r) row(r);
      } else {
#pragma unroll
        for (int r = 0; r < K; ++r) row(r);
      }
#pragma unroll
      for (int p = 0; p < kPW; ++p) {
        if (ow0 + p < g.OW) {
          *reinterpret_cast<float4*>(y + (((size_t)n * g.OH + oh) * g.OW + ow0 + p) * g.C + c) = acc[p];
          if (p == 0) st.seed(acc[p]);
          st.add(acc[p]);
        }
      }
    }
  }
  if (stats != nullptr)                             // uniform: fold the 16 pixel lanes, one atomic per channel per workgroup
    st.fold16(red, pl, cl, cok, reinterpret_cast<double*>(stats) + (size_t)(blockIdx.x & (stat_slots_used(g.C) - 1)) * 2 * g.C + c, g.C);
}

// dx[n][ih][iw] = sum_{r,s} dy[n][(ih + pt - r)/S][(iw + pl - s)/S] * w[r][s] over the taps where the divisions are exact.
// Four adjacent iw per thread, first one a multiple of 4.  With tw = iw + pl - s = base + u, base = iw0 + pl - (K-1),
// u = p + K-1 - s: for S = 2 a tap contributes iff (u & 1) == PAR (PAR = parity of base) and reads column (u + PAR) / 2
// of the dy row segment that starts at floor(base / 2).
template <int K, int S, int PAR>
__device__ __forceinline__ void dgrad_block(const DwGeom& g, const float* __restrict__ dy, const float (*wl)[64], int cl, int n, int ih,
                                            int iw0, int c, float* __restrict__ dx) {
  constexpr int NC = (S == 1) ? (K + kPW - 1) : ((K + kPW - 2 + PAR) / 2 + 1);
  const int base = iw0 + g.pl - (K - 1);
  const int owb0 = (S == 1) ? base : ((base - PAR) / 2);      // floor(base / 2): base - PAR is even
  float4 acc[kPW];
#pragma unroll
  for (int p = 0; p < kPW; ++p) acc[p] = zero4();
  auto row = [&](int r) {
    const int th = ih + g.pt - r;
    const bool rowok = th >= 0 && (S == 1 || (th & 1) == 0) && (th / S) < g.OH;
    const int oh = rowok ? th / S : 0;
    float4 d[NC];
    load_row<NC>(dy + ((size_t)n * g.OH + oh) * g.OW * g.C + c, rowok, owb0, g.OW, g.C, d);
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const float4 wt = *reinterpret_cast<const float4*>(&wl[r * K + s][cl * 4]);
#pragma unroll
      for (int p = 0; p < kPW; ++p) {
        const int u = p + K - 1 - s;
        if constexpr (S == 1) {
          acc[p] = fma4(d[u], wt, acc[p]);
        } else {
          if ((u & 1) == PAR) acc[p] = fma4(d[((u + PAR) / 2) < NC ? (u + PAR) / 2 : 0], wt, acc[p]);
        }
      }
    }
  };
  if constexpr (K >= 5) {                             // (see kRowAtATime)
#pragma unroll 1
    for (int r = 0; r < K; ++r) row(r);
  } else {
#pragma unroll
    for (int r = 0; r < K; ++r) row(r);
  }
#pragma unroll
  for (int p = 0; p < kPW; ++p)
    if (iw0 + p < g.W) *reinterpret_cast<float4*>(dx + (((size_t)n * g.H + ih) * g.W + iw0 + p) * g.C + c) = acc[p];
}

template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(DwGeom g, const float* __restrict__ dy, const float* __restrict__ w,
                                                             float* __restrict__ dx) {
  constexpr int KK = K * K;
  __shared__ float wl[KK][64];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c0 = blockIdx.y * 64, c = c0 + cl * 4;
  stage_weights<KK>(w, g.C, c0, wl);
  __syncthreads();
  if (c >= g.C) return;
  const int WB = (g.W + kPW - 1) / kPW;
  const long nb = (long)g.N * g.H * WB;
  const bool odd = ((g.pl - (K - 1)) & 1) != 0;      // parity of base (iw0 is a multiple of 4); uniform
  for (long pb = (long)blockIdx.x * 16 + pl; pb < nb; pb += (long)gridDim.x * 16) {
    int iwb, ih, n;
    if (g.vorder) { ih = (int)(pb % g.H); const long q = pb / g.H; iwb = (int)(q % WB); n = (int)(q / WB); }
    else { iwb = (int)(pb % WB); const long q = pb / WB; ih = (int)(q % g.H); n = (int)(q / g.H); }
    if (S == 1 || !odd) dgrad_block<K, S, 0>(g, dy, wl, cl, n, ih, iwb * kPW, c, dx);
    else dgrad_block<K, S, 1>(g, dy, wl, cl, n, ih, iwb * kPW, c, dx);
  }
}

// dw[c][r][s] += sum over output pixels of dy * x.  (Round 4 measured a workgroup per filter ROW -- blockIdx.z = r, K accumulators, 64-92
// registers instead of 139-354: 1 247 -> 1 291 us per EfficientNet-b3 step, dy and x are then re-read K times from L2; not kept.)
template <int K, int S, bool PRE>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(DwGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ dw, DwPre pre) {
  constexpr int KK = K * K, NC = (kPW - 1) * S + K;
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + cl * 4;
  const bool cok = c < g.C;
  float4 psc = zero4(), psh = zero4();
  if constexpr (PRE) dw_pre_affine(pre, c, cok, psc, psh);
  const int OWB = (g.OW + kPW - 1) / kPW;
  const long nb = (long)g.N * g.OH * OWB;
  float4 acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = zero4();
  if (cok) {
    for (long pb = (long)blockIdx.x * 16 + pl; pb < nb; pb += (long)gridDim.x * 16) {
      int owb, oh, n;
      if (g.vorder) { oh = (int)(pb % g.OH); const long q = pb / g.OH; owb = (int)(q % OWB); n = (int)(q / OWB); }
      else { owb = (int)(pb % OWB); const long q = pb / OWB; oh = (int)(q % g.OH); n = (int)(q / g.OH); }
      const int ow0 = owb * kPW;
      float4 d[kPW];
      {
        const float* dp = dy + (((size_t)n * g.OH + oh) * g.OW) * g.C + c;
#pragma unroll
        for (int p = 0; p < kPW; ++p) {
          const bool ok = ow0 + p < g.OW;
          const float4 t = *reinterpret_cast<const float4*>(dp + (size_t)(ok ? ow0 + p : 0) * g.C);
          d[p] = ok ? t : zero4();
        }
      }
#pragma unroll
      for (int r = 0; r < K; ++r) {
        const int ih = oh * S - g.pt + r;
        const bool rowok = ih >= 0 && ih < g.H;
        float4 v[NC];
        if constexpr (PRE) load_row_pre<NC>(x + ((size_t)n * g.H + (rowok ? ih : 0)) * g.W * g.C + c, rowok, ow0 * S - g.pl, g.W, g.C, psc, psh, v);
        else load_row<NC>(x + ((size_t)n * g.H + (rowok ? ih : 0)) * g.W * g.C + c, rowok, ow0 * S - g.pl, g.W, g.C, v);
#pragma unroll
        for (int s = 0; s < K; ++s)
#pragma unroll
          for (int p = 0; p < kPW; ++p) acc[r * K + s] = fma4(d[p], v[p * S + s], acc[r * K + s]);
      }
    }
  }
  // Fold the 16 pixel lanes: the 4 lanes of a wave with shuffles, the 4 waves through LDS (one barrier), then the k*k x 16
  // (tap, channel group) sums are spread over the threads for the atomics (the first version walked the taps one by one:
  // 2 barriers per tap and 100 atomics in a row from 16 threads).
  __shared__ float4 red[4][KK][16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    float4 a = acc[t];
    a.x += __shfl_down(a.x, 32, 64); a.y += __shfl_down(a.y, 32, 64); a.z += __shfl_down(a.z, 32, 64); a.w += __shfl_down(a.w, 32, 64);
    a.x += __shfl_down(a.x, 16, 64); a.y += __shfl_down(a.y, 16, 64); a.z += __shfl_down(a.z, 16, 64); a.w += __shfl_down(a.w, 16, 64);
    if (lane < 16) red[wave][t][lane] = a;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < KK * 16; e += 256) {
    const int t = e >> 4, l = e & 15;
    const int cc = blockIdx.y * 64 + l * 4;
    if (cc < g.C) {
      const float4 a0 = red[0][t][l], a1 = red[1][t][l], a2 = red[2][t][l], a3 = red[3][t][l];
      float* o = dw + (size_t)cc * KK + t;
      atomicAdd(o, (a0.x + a1.x) + (a2.x + a3.x)); atomicAdd(o + KK, (a0.y + a1.y) + (a2.y + a3.y));
      atomicAdd(o + 2 * KK, (a0.z + a1.z) + (a2.z + a3.z)); atomicAdd(o + 3 * KK, (a0.w + a1.w) + (a2.w + a3.w));
    }
  }
}

static int dw_vorder() {
  static const int v = [] { const char* e = getenv("HIFIHR_DW_VORDER"); return e ? atoi(e) : 0; }();
  return v;
}
static unsigned dw_grid_x(long nb, long cap) {
  long b = (nb + 15) / 16;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

#define HIFIHR_DW_DISPATCH(KERNEL, GRID, ...)                                                                  \
  do {                                                                                                          \
    if (g.K == 3 && g.stride == 1) hipLaunchKernelGGL((KERNEL<3, 1>), GRID, dim3(256), 0, st, __VA_ARGS__);      \
    else if (g.K == 3 && g.stride == 2) hipLaunchKernelGGL((KERNEL<3, 2>), GRID, dim3(256), 0, st, __VA_ARGS__); \
    else if (g.K == 5 && g.stride == 1) hipLaunchKernelGGL((KERNEL<5, 1>), GRID, dim3(256), 0, st, __VA_ARGS__); \
    else if (g.K == 5 && g.stride == 2) hipLaunchKernelGGL((KERNEL<5, 2>), GRID, dim3(256), 0, st, __VA_ARGS__); \
    else return hipErrorInvalidValue;                                                                           \
  } while (0)
#define HIFIHR_DW_DISPATCH_PRE(KERNEL, PRE_, GRID, ...)                                                              \
  do {                                                                                                                \
    if (g.K == 3 && g.stride == 1) hipLaunchKernelGGL((KERNEL<3, 1, PRE_>), GRID, dim3(256), 0, st, __VA_ARGS__);      \
    else if (g.K == 3 && g.stride == 2) hipLaunchKernelGGL((KERNEL<3, 2, PRE_>), GRID, dim3(256), 0, st, __VA_ARGS__); \
    else if (g.K == 5 && g.stride == 1) hipLaunchKernelGGL((KERNEL<5, 1, PRE_>), GRID, dim3(256), 0, st, __VA_ARGS__); \
    else if (g.K == 5 && g.stride == 2) hipLaunchKernelGGL((KERNEL<5, 2, PRE_>), GRID, dim3(256), 0, st, __VA_ARGS__); \
    else return hipErrorInvalidValue;                                                                                 \
  } while (0)

hipError_t launch_dwconv_fwd(const DwGeom& g_in, const float* x, const float* w, float* y, float* stats, hipStream_t st, const float* pre_mean,
                             const float* pre_invstd, const float* pre_gamma, const float* pre_beta) {
  DwGeom g = g_in;
  g.vorder = dw_vorder();
  const long nb = (long)g.N * g.OH * ((g.OW + kPW - 1) / kPW);
  // with statistics every workgroup ends with 128 float atomics: bound (workgroups x channels) like bn.hip does
  static const long cap_stats = [] { const char* e = getenv("HIFIHR_DW_FWD_CAP"); return e && atol(e) > 0 ? atol(e) : 256L; }();
  static const long cap_plain = [] { const char* e = getenv("HIFIHR_DW_CAP"); return e && atol(e) > 0 ? atol(e) : 2048L; }();
  const dim3 grid(dw_grid_x(nb, stats != nullptr ? cap_stats : cap_plain), (g.C + 63) / 64);
  const DwPre pre{pre_mean, pre_invstd, pre_gamma, pre_beta};
  if (pre_mean != nullptr) { HIFIHR_DW_DISPATCH_PRE(dwconv_fwd_kernel, true, grid, g, x, w, y, stats, pre); }
  else { HIFIHR_DW_DISPATCH_PRE(dwconv_fwd_kernel, false, grid, g, x, w, y, stats, pre); }
  return hipGetLastError();
}
hipError_t launch_dwconv_bwd_data(const DwGeom& g_in, const float* dy, const float* w, float* dx, hipStream_t st) {
  DwGeom g = g_in;
  g.vorder = dw_vorder();
  const long nb = (long)g.N * g.H * ((g.W + kPW - 1) / kPW);
  static const long cap_plain = [] { const char* e = getenv("HIFIHR_DW_CAP"); return e && atol(e) > 0 ? atol(e) : 2048L; }();
  const dim3 grid(dw_grid_x(nb, cap_plain), (g.C + 63) / 64);
  HIFIHR_DW_DISPATCH(dwconv_bwd_data_kernel, grid, g, dy, w, dx);
  return hipGetLastError();
}
hipError_t launch_dwconv_bwd_weight(const DwGeom& g_in, const float* x, const float* dy, float* dw, hipStream_t st, const float* pre_mean,
                                    const float* pre_invstd, const float* pre_gamma, const float* pre_beta) {
  DwGeom g = g_in;
  g.vorder = dw_vorder();
  const long nb = (long)g.N * g.OH * ((g.OW + kPW - 1) / kPW);
  // every workgroup ends with k*k x 64 atomics: give each pixel lane ~8 steps before that, between 8 and 128 workgroups per block
  long gx = nb / (16 * 8);
  if (const char* e = getenv("HIFIHR_DW_WGRAD_STEPS")) gx = nb / (16 * (atoi(e) > 0 ? atoi(e) : 8));
  if (gx < 8) gx = 8;
  if (gx > 128) gx = 128;
  const dim3 grid((unsigned)gx, (g.C + 63) / 64);
  const DwPre pre{pre_mean, pre_invstd, pre_gamma, pre_beta};
  if (pre_mean != nullptr) { HIFIHR_DW_DISPATCH_PRE(dwconv_bwd_weight_kernel, true, grid, g, x, dy, dw, pre); }
  else { HIFIHR_DW_DISPATCH_PRE(dwconv_bwd_weight_kernel, false, grid, g, x, dy, dw, pre); }
  return hipGetLastError();
}

}  // namespace hifihr
