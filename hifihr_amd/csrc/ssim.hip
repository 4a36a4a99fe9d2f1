// Fused SSIM (11x11 gaussian window, zero padding) forward and backward for gfx950.  The window sums are explicit fused multiply-adds
// (round 3: the kernels are VALU-bound -- 11 x 5 products per output in each pass -- and the library is built with -ffp-contract=off).
//
// Replaces pytorch_ssim.ssim (reference utils/pytorch_ssim/__init__.py:17-37,65-73): five depthwise 11x11
// F.conv2d calls, ~10 elementwise launches and a mean in the forward, and their autograd in the backward
// (8 MIOpen convolutions of ~1 ms each on [32,3,224,224] in the un-fused step) with one HBM-bound launch per
// direction:
//   ssim_fwd_kernel  one workgroup per 32x32 tile of one (batch, channel) plane (four outputs per thread): the 42x42 halo of both images is
//                    staged in LDS, the five windowed moments are formed separably (row pass into LDS, column pass
//                    in registers), the SSIM value is reduced per workgroup (deterministic partial sums) and the
//                    three derivative maps dS/dmu1, dS/dE[x^2], dS/dE[xy] are saved for the backward pass.
//   ssim_bwd_kernel  d mean(SSIM) / d img1 = G * A + 2 img1 (G * B) + img2 (G * C)   (G symmetric), same tiling.
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"

namespace hifihr {

constexpr int kST = 32;             // tile edge (round 3: 16 -> 32, four outputs per thread: the 26 x 26 halo of a 16 x 16 tile cost 2.6 loads per
                                    // output and 18 816 three-barrier workgroups per launch; 42 x 42 per 32 x 32 is 1.7)
constexpr int kSO = kST * kST / 256; // outputs per thread
constexpr int kSR = 5;              // window radius (11 taps)
constexpr int kSH = kST + 2 * kSR;  // 42

constexpr int kSF = 48;             // width of the staged frame: image columns [ox - 8, ox + 40), the halo [ox - 5, ox + 37) is frame columns 3 .. 44

// One halo plane into LDS.  VEC (W % 4 == 0, 16-byte aligned planes): the frame starts 8 columns left of the tile, so every group of four
// columns is one aligned float4 that lies wholly inside or wholly outside the image -- 504 vector loads per plane for the workgroup (two per
// thread) and one 16-byte LDS store each; the scalar form is 1 764 loads with ~13 VALU instructions of index arithmetic apiece, which was a
// quarter of the forward kernel's instruction count (round 5: the kernels are VALU-bound by instruction COUNT: ~1 500 per thread per tile).
template <bool VEC>
__device__ __forceinline__ void ssim_stage(float (*__restrict__ dst)[kSF], const float* __restrict__ p, int H, int W, int ox, int oy, int tid) {
  if (VEC) {
    for (int e = tid; e < kSH * (kSF / 4); e += 256) {
      const int r = e / (kSF / 4), sg = e - r * (kSF / 4);
      const int y = oy + r - kSR, x0 = ox - 8 + 4 * sg;
      const bool in = (y >= 0) && (y < H) && (x0 >= 0) && (x0 + 3 < W);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (in) v = *reinterpret_cast<const float4*>(p + (size_t)y * W + x0);
      *reinterpret_cast<float4*>(&dst[r][4 * sg]) = v;
    }
  } else {
    for (int e = tid; e < kSH * kSH; e += 256) {
      const int r = e / kSH, c = e - r * kSH;
      const int y = oy + r - kSR, x = ox + c - kSR;
      const bool in = (y >= 0) && (y < H) && (x >= 0) && (x < W);
      dst[r][c + 3] = in ? p[(size_t)y * W + x] : 0.f;
    }
  }
}
// the 14 inputs of four adjacent outputs of halo row r (frame columns c0 + 3 .. c0 + 16) as five 16-byte LDS reads
__device__ __forceinline__ void ssim_row14(const float (*__restrict__ src)[kSF], int r, int c0, float* __restrict__ a) {
  float f[20];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(&src[r][c0 + 4 * q]);
    f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
  }
#pragma unroll
  for (int k = 0; k < 14; ++k) a[k] = f[3 + k];
}

#if defined(HIFIHR_HOSTSIM)
__device__ __forceinline__ float ssim_rcp(float x) { return 1.0f / x; }
#else
__device__ __forceinline__ float ssim_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#endif

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <bool VEC>
__global__ __launch_bounds__(256) void ssim_fwd_kernel(SsimWindow win, const float* __restrict__ img1, const float* __restrict__ img2,
                                                      int H, int W, float* __restrict__ partial, float* __restrict__ dA,
                                                      float* __restrict__ dB, float* __restrict__ dC) {
  __shared__ __attribute__((aligned(16))) float xs[kSH][kSF], ys[kSH][kSF];
  __shared__ float hq[5][kSH][kST + 1];
  __shared__ float red[4];
  const int plane = blockIdx.z;
  const int ox = blockIdx.x * kST, oy = blockIdx.y * kST;
  const int tid = threadIdx.x;
  const float* p1 = img1 + (size_t)plane * H * W;
  const float* p2 = img2 + (size_t)plane * H * W;
  ssim_stage<VEC>(xs, p1, H, W, ox, oy, tid);
  ssim_stage<VEC>(ys, p2, H, W, ox, oy, tid);
  __syncthreads();
  // row pass: for every halo row, 32 output columns, five moments.  A work item is four adjacent outputs of a row: their 14 inputs are read
  // once into registers (28 LDS reads for 4 outputs where one output per item read 22 each); same summation order per output as before
  for (int e = tid; e < kSH * (kST / 4); e += 256) {
    const int r = e / (kST / 4), c0 = 4 * (e - r * (kST / 4));
    float a[14], b[14], aa[14], bb[14], ab[14];
    ssim_row14(xs, r, c0, a);
    ssim_row14(ys, r, c0, b);
#pragma unroll
    for (int k = 0; k < 14; ++k) { aa[k] = a[k] * a[k]; bb[k] = b[k] * b[k]; ab[k] = a[k] * b[k]; }        // once per input, not once per (output, tap)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float w = win.g[k];
        m1 = fmaf(w, a[o + k], m1); m2 = fmaf(w, b[o + k], m2); e11 = fmaf(w, aa[o + k], e11); e22 = fmaf(w, bb[o + k], e22);
        e12 = fmaf(w, ab[o + k], e12);
      }
      hq[0][r][c0 + o] = m1; hq[1][r][c0 + o] = m2; hq[2][r][c0 + o] = e11; hq[3][r][c0 + o] = e22; hq[4][r][c0 + o] = e12;
    }
  }
  __syncthreads();
  // column pass: a thread takes kSO vertically adjacent outputs of one column -- their kSO + 10 moment rows are read once
  float val = 0.f;
  {
    const int tx = tid % kST, ty0 = (tid / kST) * kSO;
    const int x = ox + tx;
    float h[5][kSO + 10];
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int k = 0; k < kSO + 10; ++k) h[q][k] = hq[q][ty0 + k][tx];
#pragma unroll
    for (int o4 = 0; o4 < kSO; ++o4) {
      const int y = oy + ty0 + o4;
      if (x < W && y < H) {
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
          const float w = win.g[k];
          mu1 = fmaf(w, h[0][o4 + k], mu1); mu2 = fmaf(w, h[1][o4 + k], mu2); e11 = fmaf(w, h[2][o4 + k], e11);
          e22 = fmaf(w, h[3][o4 + k], e22); e12 = fmaf(w, h[4][o4 + k], e12);
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
        const float a1 = 2.f * mu12 + C1, a2 = 2.f * s12 + C2, b1 = mu1_sq + mu2_sq + C1, b2 = s1 + s2 + C2;
        // two hardware reciprocals (1 ulp) instead of four IEEE divisions (~9 instructions each in a VALU-bound kernel); b1, b2 >= C1, C2 > 0
        const float ib1 = ssim_rcp(b1), ib2 = ssim_rcp(b2), inv = ib1 * ib2;
        const float sv = (a1 * a2) * inv;
        val += sv;
        if (dA) {
          const size_t o = (size_t)plane * H * W + (size_t)y * W + x;
          dA[o] = 2.f * mu2 * (a2 - a1) * inv - 2.f * mu1 * sv * ib1 + 2.f * mu1 * sv * ib2;   // d s / d mu1 (total)
          dB[o] = -sv * ib2;                                                                  // d s / d E[x^2]
          dC[o] = 2.f * a1 * inv;                                                             // d s / d E[xy]
        }
      }
    }
  }
  const float tot = block_sum_256(val, red);
  if (tid == 0) partial[((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = tot;
}

template <bool VEC>
__global__ __launch_bounds__(256) void ssim_bwd_kernel(SsimWindow win, const float* __restrict__ img1, const float* __restrict__ img2,
                                                      const float* __restrict__ dA, const float* __restrict__ dB,
                                                      const float* __restrict__ dC, const float* __restrict__ gscale, float inv_n,
                                                      int H, int W, float* __restrict__ gimg1) {
  __shared__ __attribute__((aligned(16))) float ta[kSH][kSF], tb[kSH][kSF], tc[kSH][kSF];
  __shared__ float hq[3][kSH][kST + 1];
  const int plane = blockIdx.z;
  const int ox = blockIdx.x * kST, oy = blockIdx.y * kST;
  const int tid = threadIdx.x;
  const size_t po = (size_t)plane * H * W;
  ssim_stage<VEC>(ta, dA + po, H, W, ox, oy, tid);
  ssim_stage<VEC>(tb, dB + po, H, W, ox, oy, tid);
  ssim_stage<VEC>(tc, dC + po, H, W, ox, oy, tid);
  __syncthreads();
  for (int e = tid; e < kSH * (kST / 4); e += 256) {        // (four adjacent outputs per item, as in the forward)
    const int r = e / (kST / 4), c0 = 4 * (e - r * (kST / 4));
    float va[14], vb[14], vc[14];
    ssim_row14(ta, r, c0, va);
    ssim_row14(tb, r, c0, vb);
    ssim_row14(tc, r, c0, vc);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float w = win.g[k];
        a = fmaf(w, va[o + k], a); b = fmaf(w, vb[o + k], b); cc = fmaf(w, vc[o + k], cc);
      }
      hq[0][r][c0 + o] = a; hq[1][r][c0 + o] = b; hq[2][r][c0 + o] = cc;
    }
  }
  __syncthreads();
  const float sc = gscale[0] * inv_n;
  const int tx = tid % kST, ty0 = (tid / kST) * kSO;
  const int x = ox + tx;
  float h[3][kSO + 10];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int k = 0; k < kSO + 10; ++k) h[q][k] = hq[q][ty0 + k][tx];
#pragma unroll
  for (int o4 = 0; o4 < kSO; ++o4) {
    const int y = oy + ty0 + o4;
    if (x < W && y < H) {
      float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float w = win.g[k];
        a = fmaf(w, h[0][o4 + k], a); b = fmaf(w, h[1][o4 + k], b); cc = fmaf(w, h[2][o4 + k], cc);
      }
      const size_t o = po + (size_t)y * W + x;
      gimg1[o] = sc * (a + 2.f * img1[o] * b + img2[o] * cc);
    }
  }
}

// out[0] = offset + scale * sum(partial[0 .. count)): the scalar behind the forward (SSIM = sum / n, or the ssim_tex loss term
// lambda - lambda * sum / n) in one launch instead of a reduction, a fill and an add; fixed summation order (deterministic)
__global__ __launch_bounds__(256) void ssim_finish_kernel(const float* __restrict__ partial, int count, float scale, float offset,
                                                         float* __restrict__ out) {
  __shared__ float red[4];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = threadIdx.x;
  for (; i + 768 < count; i += 1024) { s0 += partial[i]; s1 += partial[i + 256]; s2 += partial[i + 512]; s3 += partial[i + 768]; }
  for (; i < count; i += 256) s0 += partial[i];
  const float s = block_sum_256((s0 + s1) + (s2 + s3), red);
  if (threadIdx.x == 0) out[0] = offset + scale * s;
}

int ssim_tile_edge() { return kST; }

// vector staging: whole float4 groups inside or outside the image, planes 16-byte aligned (plane stride H * W floats with W % 4 == 0)
static bool ssim_vec_ok(int W, const float* a, const float* b, const float* c) {
  static const int on = [] { const char* e = getenv("HIFIHR_SSIM_VEC"); return e ? atoi(e) : 1; }();
  return on && W % 4 == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

hipError_t launch_ssim_finish(const float* partial, int count, float scale, float offset, float* out, hipStream_t st) {
  hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(256), 0, st, partial, count, scale, offset, out);
  return hipGetLastError();
}

hipError_t launch_ssim_fwd(const SsimWindow& win, const float* img1, const float* img2, int planes, int H, int W, float* partial,
                           float* dA, float* dB, float* dC, hipStream_t st) {
  const dim3 grid((W + kST - 1) / kST, (H + kST - 1) / kST, planes);
  if (ssim_vec_ok(W, img1, img2, nullptr))
    hipLaunchKernelGGL(ssim_fwd_kernel<true>, grid, dim3(256), 0, st, win, img1, img2, H, W, partial, dA, dB, dC);
  else
    hipLaunchKernelGGL(ssim_fwd_kernel<false>, grid, dim3(256), 0, st, win, img1, img2, H, W, partial, dA, dB, dC);
  return hipGetLastError();
}

hipError_t launch_ssim_bwd(const SsimWindow& win, const float* img1, const float* img2, const float* dA, const float* dB,
                           const float* dC, const float* gscale, int planes, int H, int W, float* gimg1, hipStream_t st, float out_scale) {
  const dim3 grid((W + kST - 1) / kST, (H + kST - 1) / kST, planes);
  const float inv_n = out_scale / ((float)planes * (float)H * (float)W);
  if (ssim_vec_ok(W, dA, dB, dC))
    hipLaunchKernelGGL(ssim_bwd_kernel<true>, grid, dim3(256), 0, st, win, img1, img2, dA, dB, dC, gscale, inv_n, H, W, gimg1);
  else
    hipLaunchKernelGGL(ssim_bwd_kernel<false>, grid, dim3(256), 0, st, win, img1, img2, dA, dB, dC, gscale, inv_n, H, W, gimg1);
  return hipGetLastError();
}

}  // namespace hifihr
