// NHWC fp32 convolution for gfx950 as implicit GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces what the reference's encoder dispatches to cuDNN/MIOpen (reference network/res_encoder.py:364-373:
// conv2d forward, backward-data, backward-weight of the ResNet trunk).  fp32 in, fp32 accumulate: the MFMA result
// is bit-for-bit a k-ordered fmaf chain, so numerics are those of a plain fp32 convolution.
//
//   conv_igemm_kernel<BM,BN>   Y[m][k] = sum_q A[m][q] W[k][q],  m = (n,oh,ow), q = (r,s,c).
//        The A tile is gathered on the fly (im2col never materialised): every lane loads 16 B (4 channels of one
//        tap) per row, tiles go through LDS k-interleaved so each lane reads its 8 k-steps as two ds_read_b128
//        (row stride 20 floats: conflict-free), double-buffered, one barrier per 16-deep K chunk, 32 MFMAs
//        per wave between barriers.  The same kernel computes backward-data (dgrad = 1): rows are input pixels
//        and the gather walks dy with the stride divisibility test, against the [C][R][S][K] transposed weight.
//   conv_wgrad_kernel<BM,BN>   dW[k][q] += sum_m dy[m][k] A[m][q]; the pixel range is split over blockIdx.z and
//        partial tiles are added with fp32 atomics (256 contiguous bytes per wave instruction).
//   weight_transpose_kernel    [K][R][S][C] -> [C][R][S][K] for dgrad.
//   image_to_nhwc4_kernel      normalize_batch_3C (reference network/res_encoder.py:212-216) fused with the
//                              NCHW(3) -> NHWC(4, zero padded) repack the first convolution consumes.
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"

namespace hifihr {

#if defined(HIFIHR_HOSTSIM)
typedef hs_floatx16 floatx16;
#else
typedef float floatx16 __attribute__((ext_vector_type(16)));
#endif

constexpr int kBK = 16;     // K-chunk depth (8 MFMA k-steps of 2)
constexpr int kLD = 20;     // LDS row stride in floats (80 B: ds_read_b128 conflict-free, 16-B aligned)

template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvGeom g, const float* __restrict__ src,
                                                        const float* __restrict__ wgt, const float* __restrict__ bias,
                                                        float* __restrict__ dst) {
  constexpr int TM = BM / 64, TN = BN / 64;      // 32x32 MFMA tiles per wave (waves are arranged 2 x 2)
  __shared__ __attribute__((aligned(16))) float As[2][BM * kLD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * kLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, r31 = lane & 31;
  const int M = g.N * g.OH * g.OW;
  const int Q = g.R * g.S * g.IC;
  const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
  const int lrow = tid >> 2, seg = (tid & 3) * 4;

  // per-thread row bookkeeping for the gather (constant over the K loop)
  int a_n[TM], a_h[TM], a_w[TM];
  bool a_ok[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = bm0 + lrow + 64 * i;
    a_ok[i] = m < M;
    const int mm = a_ok[i] ? m : 0;
    const int n = mm / (g.OH * g.OW);
    const int rem = mm - n * (g.OH * g.OW);
    const int oh = rem / g.OW, ow = rem - oh * g.OW;
    a_n[i] = n;
    a_h[i] = g.dgrad ? oh + g.pad : oh * g.stride - g.pad;
    a_w[i] = g.dgrad ? ow + g.pad : ow * g.stride - g.pad;
  }

  float4 ra[TM], rb[TN];
  auto load_global = [&](int q0) {
    const int q = q0 + seg;
    int r = 0, s = 0, c = 0;
    const bool qok = q < Q;
    if (qok) {
      const int t = q / g.IC;
      c = q - t * g.IC;
      r = t / g.S;
      s = t - r * g.S;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qok && a_ok[i]) {
        int ih, iw;
        bool ok;
        if (g.dgrad) {
          const int th = a_h[i] - r, tw = a_w[i] - s;
          ok = (th >= 0) && (tw >= 0) && (th % g.stride == 0) && (tw % g.stride == 0);
          ih = th / g.stride; iw = tw / g.stride;
          ok = ok && (ih < g.IH) && (iw < g.IW);
        } else {
          ih = a_h[i] + r; iw = a_w[i] + s;
          ok = (ih >= 0) && (ih < g.IH) && (iw >= 0) && (iw < g.IW);
        }
        if (ok) v = *reinterpret_cast<const float4*>(src + (((size_t)a_n[i] * g.IH + ih) * g.IW + iw) * g.IC + c);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int k = bn0 + lrow + 64 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qok && k < g.OC) v = *reinterpret_cast<const float4*>(wgt + (size_t)k * Q + q);
      rb[j] = v;
    }
  };
  auto store_lds = [&](int buf) {
    // k-interleave: even columns of the chunk -> floats [0,8), odd columns -> [8,16) of the row
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float* p = &As[buf][(lrow + 64 * i) * kLD];
      *reinterpret_cast<float2*>(p + seg / 2) = make_float2(ra[i].x, ra[i].z);
      *reinterpret_cast<float2*>(p + 8 + seg / 2) = make_float2(ra[i].y, ra[i].w);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float* p = &Bs[buf][(lrow + 64 * j) * kLD];
      *reinterpret_cast<float2*>(p + seg / 2) = make_float2(rb[j].x, rb[j].z);
      *reinterpret_cast<float2*>(p + 8 + seg / 2) = make_float2(rb[j].y, rb[j].w);
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nch = (Q + kBK - 1) / kBK;
  load_global(0);
  store_lds(0);
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) load_global((ch + 1) * kBK);
    float a[TM][8], b[TN][8];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float* p = &As[buf][(wm * (BM / 2) + i * 32 + r31) * kLD + half * 8];
      const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
      a[i][0] = lo.x; a[i][1] = lo.y; a[i][2] = lo.z; a[i][3] = lo.w; a[i][4] = hi.x; a[i][5] = hi.y; a[i][6] = hi.z; a[i][7] = hi.w;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float* p = &Bs[buf][(wn * (BN / 2) + j * 32 + r31) * kLD + half * 8];
      const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
      b[j][0] = lo.x; b[j][1] = lo.y; b[j][2] = lo.z; b[j][3] = lo.w; b[j][4] = hi.x; b[j][5] = hi.y; b[j][6] = hi.z; b[j][7] = hi.w;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    if (ch + 1 < nch) store_lds(buf ^ 1);
    __syncthreads();
  }

  // epilogue: D[i][j], i = pixel row, j = output channel; lanes 0..31 hold 32 consecutive channels of one row
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int k = bn0 + wn * (BN / 2) + j * 32 + r31;
      const float bv = (bias && k < g.OC) ? bias[k] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = bm0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (m < M && k < g.OC) dst[(size_t)m * g.OC + k] = acc[i][j][e] + bv;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// backward weight
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ dw, int chunks_per_split) {
  // GEMM: dW[k][q] (BM x BN tile) = sum over pixels m of dy[m][k] * patch[m][q];  g describes the FORWARD conv
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int LA = BM + 4, LB = BN + 4;
  __shared__ __attribute__((aligned(16))) float As[2][kBK * LA];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBK * LB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, r31 = lane & 31;
  const int M = g.N * g.OH * g.OW;
  const int Q = g.R * g.S * g.IC;
  const int K = g.OC;
  const int bq0 = blockIdx.x * BN, bk0 = blockIdx.y * BM;
  const int nch_total = (M + kBK - 1) / kBK;
  const int ch_lo = blockIdx.z * chunks_per_split;
  const int ch_hi = min(ch_lo + chunks_per_split, nch_total);
  if (ch_lo >= ch_hi) return;

  // load mapping: a row of the tile is one pixel m; BM/4 (BN/4) float4 per row
  constexpr int APR = BM / 4, BPR = BN / 4;            // float4 per row
  constexpr int AROWS = 256 / APR, BROWS = 256 / BPR;  // rows covered per pass
  constexpr int AL = kBK / AROWS, BL = kBK / BROWS;    // passes
  const int a_row = tid / APR, a_col = (tid % APR) * 4;
  const int b_row = tid / BPR, b_col = (tid % BPR) * 4;
  // the q coordinates of this thread's patch column are constant over the loop
  const int q = bq0 + b_col;
  const bool qok = q < Q;
  int qr = 0, qs = 0, qc = 0;
  if (qok) {
    const int t = q / g.IC;
    qc = q - t * g.IC;
    qr = t / g.S;
    qs = t - qr * g.S;
  }
  float4 ra[AL], rb[BL];
  auto load_global = [&](int ch) {
    const int m0 = ch * kBK;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int m = m0 + a_row + AROWS * i;
      const int k = bk0 + a_col;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M && k < K) v = *reinterpret_cast<const float4*>(dy + (size_t)m * K + k);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      const int m = m0 + b_row + BROWS * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M && qok) {
        const int n = m / (g.OH * g.OW);
        const int rem = m - n * (g.OH * g.OW);
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
        const int ih = oh * g.stride - g.pad + qr, iw = ow * g.stride - g.pad + qs;
        if (ih >= 0 && ih < g.IH && iw >= 0 && iw < g.IW)
          v = *reinterpret_cast<const float4*>(x + (((size_t)n * g.IH + ih) * g.IW + iw) * g.IC + qc);
      }
      rb[i] = v;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i) *reinterpret_cast<float4*>(&As[buf][(a_row + AROWS * i) * LA + a_col]) = ra[i];
#pragma unroll
    for (int i = 0; i < BL; ++i) *reinterpret_cast<float4*>(&Bs[buf][(b_row + BROWS * i) * LB + b_col]) = rb[i];
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_global(ch_lo);
  store_lds(0);
  __syncthreads();
  for (int ch = ch_lo; ch < ch_hi; ++ch) {
    const int buf = (ch - ch_lo) & 1;
    if (ch + 1 < ch_hi) load_global(ch + 1);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][(2 * t + half) * LA + wm * (BM / 2) + i * 32 + r31];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[buf][(2 * t + half) * LB + wn * (BN / 2) + j * 32 + r31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (ch + 1 < ch_hi) store_lds(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int qq = bq0 + wn * (BN / 2) + j * 32 + r31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = bk0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (k < K && qq < Q) atomicAdd(dw + (size_t)k * Q + qq, acc[i][j][e]);
      }
    }
}

// [K][RS][C] -> [C][RS][K]
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int RS, int C) {
  const size_t n = (size_t)K * RS * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % K);
    const size_t rest = i / K;
    const int rs = (int)(rest % RS);
    const int c = (int)(rest / RS);
    wt[i] = w[((size_t)k * RS + rs) * C + c];
  }
}

// images NCHW [B][3][H][W] in [0,1] -> NHWC4 [B][H][W][4] = ((x - mean) / std, 0)
__global__ __launch_bounds__(256) void image_to_nhwc4_kernel(const float* __restrict__ img, float4* __restrict__ out, int B, int HW) {
  const size_t n = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t b = i / HW, p = i - b * HW;
    const float* s = img + b * 3 * HW + p;
    out[i] = make_float4((s[0] - 0.485f) / 0.229f, (s[HW] - 0.456f) / 0.224f, (s[2 * (size_t)HW] - 0.406f) / 0.225f, 0.f);
  }
}
// gradient of the above is never needed (images carry no gradient)

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int pick_tile(long M, int OC) {
  // 0: 128x128, 1: 128x64, 2: 64x64 -- the largest tile that still gives ~2 workgroups per CU
  if (OC % 128 == 0 || OC > 128) {
    if ((M / 128) * ((OC + 127) / 128) >= 512) return 0;
  }
  if ((M / 128) * ((OC + 63) / 64) >= 400) return 1;
  return 2;
}

hipError_t launch_conv_igemm(const ConvGeom& g, const float* src, const float* wgt, const float* bias, float* dst, hipStream_t st) {
  const long M = (long)g.N * g.OH * g.OW;
  if (g.IC % 4 != 0) return hipErrorInvalidValue;
  switch (pick_tile(M, g.OC)) {
    case 0:
      hipLaunchKernelGGL((conv_igemm_kernel<128, 128>), dim3((unsigned)((M + 127) / 128), (g.OC + 127) / 128), dim3(256), 0, st, g, src, wgt, bias, dst);
      break;
    case 1:
      hipLaunchKernelGGL((conv_igemm_kernel<128, 64>), dim3((unsigned)((M + 127) / 128), (g.OC + 63) / 64), dim3(256), 0, st, g, src, wgt, bias, dst);
      break;
    default:
      hipLaunchKernelGGL((conv_igemm_kernel<64, 64>), dim3((unsigned)((M + 63) / 64), (g.OC + 63) / 64), dim3(256), 0, st, g, src, wgt, bias, dst);
  }
  return hipGetLastError();
}

hipError_t launch_conv_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, hipStream_t st) {
  const long M = (long)g.N * g.OH * g.OW;
  const int Q = g.R * g.S * g.IC;
  if (g.IC % 4 != 0 || g.OC % 4 != 0) return hipErrorInvalidValue;
  const int nch = (int)((M + kBK - 1) / kBK);
  const bool big = (g.OC % 128 == 0);
  const int tiles = big ? ((g.OC + 127) / 128) * ((Q + 127) / 128) : ((g.OC + 63) / 64) * ((Q + 127) / 128);
  int splits = (1024 + tiles - 1) / tiles;
  if (splits > nch / 4) splits = nch / 4;
  if (splits < 1) splits = 1;
  const int cps = (nch + splits - 1) / splits;
  splits = (nch + cps - 1) / cps;
  if (big)
    hipLaunchKernelGGL((conv_wgrad_kernel<128, 128>), dim3((Q + 127) / 128, (g.OC + 127) / 128, splits), dim3(256), 0, st, g, x, dy, dw, cps);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<64, 128>), dim3((Q + 127) / 128, (g.OC + 63) / 64, splits), dim3(256), 0, st, g, x, dy, dw, cps);
  return hipGetLastError();
}

hipError_t launch_weight_transpose(const float* w, float* wt, int K, int RS, int C, hipStream_t st) {
  const size_t n = (size_t)K * RS * C;
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(weight_transpose_kernel, dim3(blocks), dim3(256), 0, st, w, wt, K, RS, C);
  return hipGetLastError();
}

hipError_t launch_image_to_nhwc4(const float* img, float* out, int B, int HW, hipStream_t st) {
  const size_t n = (size_t)B * HW;
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(image_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, st, img, reinterpret_cast<float4*>(out), B, HW);
  return hipGetLastError();
}

}  // namespace hifihr
