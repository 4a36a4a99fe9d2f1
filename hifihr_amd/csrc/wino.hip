// Winograd F(2x2, 3x3) transforms for the stride-1 3x3 convolutions with many channels (ResNet-18 layers 3 and 4 at 14x14:
// 60 % of the network's FLOPs).  out = A^T [ (G g G^T) . (B^T d B) ] A turns a 3x3 convolution over 2x2 output tiles into 16
// independent [tiles x C] x [C x K] GEMMs with 2.25x fewer multiplications; the GEMMs run on the MFMA implicit-GEMM kernel
// (conv.hip, batch = 16, balanced schedule), these kernels are the HBM-bound glue:
//   wino_weight_transform   U[16][K][C]  = G g G^T     (flip = 1: g rotated by 180 degrees, for backward-data on the
//                                                        [C][3][3][K] transposed weights)
//   wino_input_transform    V[16][T][C]  = B^T d B     d = 4x4 input patch of tile t (zero outside the image), T = N*ceil(H/2)*ceil(W/2)
//   wino_output_transform   y[N][H][W][K] = A^T m A    m = M[.][t][.]; optionally adds the batch-norm sum / sum of squares of y
//                                                        to the slot buffer (bn.hip), like the direct kernel's epilogue does,
//                                                        or applies bias (+ ReLU) like the direct kernel's act epilogue (VGG19)
//   wino_dy_transform       Y'[16][T][K] = A dy A^T    backward-weight: dU[pos][k][c] = sum_t Y'[pos][t][k] V[pos][t][c] (16 batched
//   wino_dw_transform       dw[K][3][3][C] += G^T dU G  reductions on conv_wgrad_kernel), then back to the 3x3 filter
// Replaces (together with the batched GEMM) the same cuDNN/MIOpen dispatches as conv.hip (reference network/res_encoder.py:364-373).
// fp32 throughout: F(2,3) has transform constants {0, +-1, +-1/2}, its rounding error stays at a few ulp of the direct result.
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"
#include "wino4_math.h"

namespace hifihr {

__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// u = G g G^T for one (k, 4 channels) column of 3x3 taps, stored to U[pos][k][c..c+3] (row stride `ld` floats between k)
__device__ __forceinline__ void wino_weight_tile(const float4 (&g)[3][3], float* __restrict__ U, size_t plane, size_t off) {
  // t = G g (4x3), u = t G^T (4x4);  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
  float4 t[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const float4 sum02 = add4(g[0][s], g[2][s]);
    t[0][s] = g[0][s];
    t[1][s] = make_float4(0.5f * (sum02.x + g[1][s].x), 0.5f * (sum02.y + g[1][s].y), 0.5f * (sum02.z + g[1][s].z), 0.5f * (sum02.w + g[1][s].w));
    t[2][s] = make_float4(0.5f * (sum02.x - g[1][s].x), 0.5f * (sum02.y - g[1][s].y), 0.5f * (sum02.z - g[1][s].z), 0.5f * (sum02.w - g[1][s].w));
    t[3][s] = g[2][s];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float4 sum02 = add4(t[r][0], t[r][2]);
    float4 u[4];
    u[0] = t[r][0];
    u[1] = make_float4(0.5f * (sum02.x + t[r][1].x), 0.5f * (sum02.y + t[r][1].y), 0.5f * (sum02.z + t[r][1].z), 0.5f * (sum02.w + t[r][1].w));
    u[2] = make_float4(0.5f * (sum02.x - t[r][1].x), 0.5f * (sum02.y - t[r][1].y), 0.5f * (sum02.z - t[r][1].z), 0.5f * (sum02.w - t[r][1].w));
    u[3] = t[r][2];
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<float4*>(U + (size_t)(r * 4 + c) * plane + off) = u[c];
  }
}

// thread = (output channel k, 4 input channels); w[K][3][3][C] -> U[16][K][C]   (bid / nblk: this job's block index / count)
__device__ __forceinline__ void wino_weight_transform_body(const float* __restrict__ w, float* __restrict__ U, int K, int C, int flip,
                                                           unsigned bid, unsigned nblk) {
  const int C4 = C / 4;
  const size_t total = (size_t)K * C4;
  for (size_t i = (size_t)bid * 256 + threadIdx.x; i < total; i += (size_t)nblk * 256) {
    const int cg = (int)(i % C4), k = (int)(i / C4);
    float4 g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int rr = flip ? 2 - r : r, ss = flip ? 2 - s : s;
        g[r][s] = *reinterpret_cast<const float4*>(w + (((size_t)k * 3 + rr) * 3 + ss) * C + cg * 4);
      }
    wino_weight_tile(g, U, (size_t)K * C, (size_t)k * C + cg * 4);
  }
}

// backward-data weights straight from w[K][3][3][C]: U'[16][C][K] = G g' G^T with g'[c][r][s][k] = w[k][2-r][2-s][c] (the transposed,
// 180-degree rotated filter) -- the same values wino_weight_transform(flip = 1) computes from the [C][3][3][K] transpose, without
// materialising it.  A workgroup stages a (64 k) x (16 c) block of w through LDS (coalesced 64-byte runs along c), then thread
// (4 consecutive k, c) reads its taps from LDS and writes float4 along k: 256 contiguous bytes per 16 lanes.
constexpr int kWtLd = 17;                                  // padded c stride of the staged block (bank spread)
__device__ __forceinline__ void wino_weight_transform_t_body(const float* __restrict__ w, float* __restrict__ U, int K, int C, unsigned bid,
                                                             unsigned nblk, float* __restrict__ lds /* [64 * 9 * kWtLd] */) {
  const int kt = (K + 63) / 64, ct = (C + 15) / 16;
  for (int tile = (int)bid; tile < kt * ct; tile += (int)nblk) {
    const int k0 = (tile % kt) * 64, c0 = (tile / kt) * 16;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 9 * 4; i += 256) {           // row = (k, rs), 4 float4 per row
      const int q = i & 3, row = i >> 2;
      const int k = k0 + row / 9, c = c0 + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < K && c < C) v = *reinterpret_cast<const float4*>(w + ((size_t)k * 9 + row % 9) * C + c);
      float* d = lds + row * kWtLd + q * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int kg = threadIdx.x & 15, cl = threadIdx.x >> 4;
    const int k = k0 + kg * 4, c = c0 + cl;
    if (k < K && c < C) {
      float4 g[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
          const float* p = lds + ((kg * 4) * 9 + (2 - r) * 3 + (2 - s2)) * kWtLd + cl;
          g[r][s2] = make_float4(p[0], p[9 * kWtLd], p[18 * kWtLd], p[27 * kWtLd]);
        }
      wino_weight_tile(g, U, (size_t)C * K, (size_t)c * K + k);
    }
  }
}

// [K][RS][C] -> [C][RS][K]
__device__ __forceinline__ void weight_transpose_body(const float* __restrict__ w, float* __restrict__ wt, int K, int RS, int C, unsigned bid,
                                                      unsigned nblk) {
  const size_t n = (size_t)K * RS * C;
  for (size_t i = (size_t)bid * 256 + threadIdx.x; i < n; i += (size_t)nblk * 256) {
    const int k = (int)(i % K);
    const size_t rest = i / K;
    const int rs = (int)(rest % RS);
    const int c = (int)(rest / RS);
    wt[i] = w[((size_t)k * RS + rs) * C + c];
  }
}

// kind 5: w[K][RS][C] -> wp[K][RS][C4], C4 = C rounded up to a multiple of 4, zero fill (the 3-channel stem filter for NHWC4 images)
__device__ __forceinline__ void weight_pad_c4_body(const float* __restrict__ w, float* __restrict__ wp, int K, int RS, int C, unsigned bid,
                                                   unsigned nblk) {
  const int C4 = (C + 3) & ~3;
  const size_t n = (size_t)K * RS * C4;
  for (size_t i = (size_t)bid * 256 + threadIdx.x; i < n; i += (size_t)nblk * 256) {
    const int c = (int)(i % C4);
    wp[i] = c < C ? w[(i / C4) * C + c] : 0.f;
  }
}

__global__ __launch_bounds__(256) void wino_weight_transform_kernel(const float* __restrict__ w, float* __restrict__ U, int K, int C, int flip) {
  wino_weight_transform_body(w, U, K, C, flip, blockIdx.x, gridDim.x);
}

// Every per-step weight re-layout of a model in ONE launch (the weights change once per step, the ~40 separate transposes /
// transforms of a ResNet-18 step cost ~5 us of launch floor each): blockIdx.y = job, blockIdx.x strides over the job's elements.
__global__ __launch_bounds__(256) void weight_prep_kernel(const PrepJob* __restrict__ jobs) {
  __shared__ float lds[64 * 9 * kWtLd];
  const PrepJob j = jobs[blockIdx.y];
  if (j.kind == 0) weight_transpose_body(j.src, j.dst, j.K, j.RS, j.C, blockIdx.x, gridDim.x);
  else if (j.kind == 1) wino_weight_transform_body(j.src, j.dst, j.K, j.C, 0, blockIdx.x, gridDim.x);
  else if (j.kind == 2) wino_weight_transform_t_body(j.src, j.dst, j.K, j.C, blockIdx.x, gridDim.x, lds);
  else if (j.kind == 3) w4::wino4_weight_transform_body(j.src, j.dst, j.K, j.C, blockIdx.x, gridDim.x);        // F(4x4, 3x3): U[36][K][C]
  else if (j.kind == 4) w4::wino4_weight_transform_t_body(j.src, j.dst, j.K, j.C, blockIdx.x, gridDim.x, lds);  // F(4x4, 3x3): U'[36][C][K]
  else weight_pad_c4_body(j.src, j.dst, j.K, j.RS, j.C, blockIdx.x, gridDim.x);
}

// thread = (tile, 4 channels); x[N][H][W][C] -> V[16][T][C]
// DUAL: x is a gradient dy that backward-data (V = B^T d B of the padded 4x4 patch) AND backward-weight (Y' = A dy A^T of the patch's
// central 2x2 block = this tile's outputs) both consume: one read of dy produces both (saves the wino_dy_transform launch + pass)
template <bool DUAL>
__global__ __launch_bounds__(256) void wino_input_transform_kernel(const float* __restrict__ x, float* __restrict__ V, float* __restrict__ Y, int N,
                                                                  int H, int W, int C, int TH, int TW) {
  const int C4 = C / 4;
  const size_t T = (size_t)N * TH * TW, total = T * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % C4);
    const size_t t = i / C4;
    const int tw = (int)(t % TW), th = (int)((t / TW) % TH), n = (int)(t / ((size_t)TW * TH));
    float4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ih = 2 * th - 1 + r;
      const bool rok = ih >= 0 && ih < H;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int iw = 2 * tw - 1 + c;
        const bool ok = rok && iw >= 0 && iw < W;
        const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)n * H + (rok ? ih : 0)) * W + (ok ? iw : 0)) * C + cg * 4);
        d[r][c] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (DUAL) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 ty[4][2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        ty[0][b] = d[1][1 + b];
        ty[1][b] = add4(d[1][1 + b], d[2][1 + b]);
        ty[2][b] = sub4(d[1][1 + b], d[2][1 + b]);
        ty[3][b] = sub4(z, d[2][1 + b]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* o = Y + ((size_t)(r * 4) * T + t) * C + cg * 4;
        *reinterpret_cast<float4*>(o) = ty[r][0];
        *reinterpret_cast<float4*>(o + T * C) = add4(ty[r][0], ty[r][1]);
        *reinterpret_cast<float4*>(o + 2 * T * C) = sub4(ty[r][0], ty[r][1]);
        *reinterpret_cast<float4*>(o + 3 * T * C) = sub4(z, ty[r][1]);
      }
    }
    // t = B^T d,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1];  v = t B
    float4 tt[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      tt[0][c] = sub4(d[0][c], d[2][c]);
      tt[1][c] = add4(d[1][c], d[2][c]);
      tt[2][c] = sub4(d[2][c], d[1][c]);
      tt[3][c] = sub4(d[1][c], d[3][c]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float4 v0 = sub4(tt[r][0], tt[r][2]), v1 = add4(tt[r][1], tt[r][2]), v2 = sub4(tt[r][2], tt[r][1]), v3 = sub4(tt[r][1], tt[r][3]);
      float* o = V + ((size_t)(r * 4) * T + t) * C + cg * 4;
      *reinterpret_cast<float4*>(o) = v0;
      *reinterpret_cast<float4*>(o + T * C) = v1;
      *reinterpret_cast<float4*>(o + 2 * T * C) = v2;
      *reinterpret_cast<float4*>(o + 3 * T * C) = v3;
    }
  }
}

// workgroup = 16 tile lanes x 16 float4 channel lanes (64 channels, blockIdx.y); M[16][T][K] -> y[N][H][W][K] (+ stats)
__global__ __launch_bounds__(256) void wino_output_transform_kernel(const float* __restrict__ Mm, float* __restrict__ y, float* __restrict__ stats,
                                                                   const float* __restrict__ bias, int relu, int N, int H, int W, int K,
                                                                   int TH, int TW) {
  __shared__ double red[2][16][16][4];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int k = blockIdx.y * 64 + cl * 4;
  const bool kok = k < K;
  const size_t T = (size_t)N * TH * TW;
  Stat4 st;                                         // batch-norm statistics of y (hifihr_internal.h "FORWARD statistics")
  if (kok) {
    const float4 bv = bias != nullptr ? *reinterpret_cast<const float4*>(bias + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float lo = relu ? 0.f : -3.402823466e38f;
    for (size_t t = (size_t)blockIdx.x * 16 + tl; t < T; t += (size_t)gridDim.x * 16) {
      const int tw = (int)(t % TW), th = (int)((t / TW) % TH), n = (int)(t / ((size_t)TW * TH));
      float4 m[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) m[r][c] = *reinterpret_cast<const float4*>(Mm + ((size_t)(r * 4 + c) * T + t) * K + k);
      // s = A^T m,  A^T = [1 1 1 0; 0 1 -1 -1];  out = s A
      float4 s[2][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s[0][c] = add4(add4(m[0][c], m[1][c]), m[2][c]);
        s[1][c] = sub4(sub4(m[1][c], m[2][c]), m[3][c]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int oh = 2 * th + a;
        float4 o0 = add4(add4(add4(s[a][0], s[a][1]), s[a][2]), bv), o1 = add4(sub4(sub4(s[a][1], s[a][2]), s[a][3]), bv);
        o0 = make_float4(fmaxf(o0.x, lo), fmaxf(o0.y, lo), fmaxf(o0.z, lo), fmaxf(o0.w, lo));
        o1 = make_float4(fmaxf(o1.x, lo), fmaxf(o1.y, lo), fmaxf(o1.z, lo), fmaxf(o1.w, lo));
        if (oh < H) {
          float* p = y + (((size_t)n * H + oh) * W + 2 * tw) * K + k;
          *reinterpret_cast<float4*>(p) = o0;
          if (a == 0) st.seed(o0);
          st.add(o0);
          if (2 * tw + 1 < W) {
            *reinterpret_cast<float4*>(p + K) = o1;
            st.add(o1);
          }
        }
      }
    }
  }
  if (stats != nullptr)                             // uniform
    st.fold16(red, tl, cl, kok, reinterpret_cast<double*>(stats) + (size_t)(blockIdx.x & (stat_slots_used(K) - 1)) * 2 * K + k, K);
}

// backward-weight glue.  thread = (tile, 4 channels): Y'[16][T][K] = A dy A^T with A = [1 0; 1 1; 1 -1; 0 -1] (dy outside the image = 0)
__global__ __launch_bounds__(256) void wino_dy_transform_kernel(const float* __restrict__ dy, float* __restrict__ Y, int N, int H, int W, int K, int TH,
                                                               int TW) {
  const int K4 = K / 4;
  const size_t T = (size_t)N * TH * TW, total = T * K4;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int kg = (int)(i % K4);
    const size_t t = i / K4;
    const int tw = (int)(t % TW), th = (int)((t / TW) % TH), n = (int)(t / ((size_t)TW * TH));
    float4 d[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oh = 2 * th + a, ow = 2 * tw + b;
        const bool ok = oh < H && ow < W;
        const float4 v = *reinterpret_cast<const float4*>(dy + (((size_t)n * H + (ok ? oh : 0)) * W + (ok ? ow : 0)) * K + kg * 4);
        d[a][b] = ok ? v : z;
      }
    float4 tt[4][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      tt[0][b] = d[0][b];
      tt[1][b] = add4(d[0][b], d[1][b]);
      tt[2][b] = sub4(d[0][b], d[1][b]);
      tt[3][b] = sub4(z, d[1][b]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* o = Y + ((size_t)(r * 4) * T + t) * K + kg * 4;
      *reinterpret_cast<float4*>(o) = tt[r][0];
      *reinterpret_cast<float4*>(o + T * K) = add4(tt[r][0], tt[r][1]);
      *reinterpret_cast<float4*>(o + 2 * T * K) = sub4(tt[r][0], tt[r][1]);
      *reinterpret_cast<float4*>(o + 3 * T * K) = sub4(z, tt[r][1]);
    }
  }
}

// thread = (k, 4 channels): dw[K][3][3][C] += G^T dU G,  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
__global__ __launch_bounds__(256) void wino_dw_transform_kernel(float* __restrict__ dU, float* __restrict__ dw, int K, int C, int clear) {
  const int C4 = C / 4;
  const size_t total = (size_t)K * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % C4), k = (int)(i / C4);
    float4 u[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float4* p = reinterpret_cast<float4*>(dU + ((size_t)(r * 4 + c) * K + k) * C + cg * 4);
        u[r][c] = *p;
        if (clear) *p = make_float4(0.f, 0.f, 0.f, 0.f);      // self-cleaning accumulator: zero again for the next reduction
      }
    float4 t[3][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 h12 = add4(u[1][c], u[2][c]), d12 = sub4(u[1][c], u[2][c]);
      t[0][c] = make_float4(u[0][c].x + 0.5f * h12.x, u[0][c].y + 0.5f * h12.y, u[0][c].z + 0.5f * h12.z, u[0][c].w + 0.5f * h12.w);
      t[1][c] = make_float4(0.5f * d12.x, 0.5f * d12.y, 0.5f * d12.z, 0.5f * d12.w);
      t[2][c] = make_float4(u[3][c].x + 0.5f * h12.x, u[3][c].y + 0.5f * h12.y, u[3][c].z + 0.5f * h12.z, u[3][c].w + 0.5f * h12.w);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float4 h12 = add4(t[r][1], t[r][2]), d12 = sub4(t[r][1], t[r][2]);
      const float4 g0 = make_float4(t[r][0].x + 0.5f * h12.x, t[r][0].y + 0.5f * h12.y, t[r][0].z + 0.5f * h12.z, t[r][0].w + 0.5f * h12.w);
      const float4 g1 = make_float4(0.5f * d12.x, 0.5f * d12.y, 0.5f * d12.z, 0.5f * d12.w);
      const float4 g2 = make_float4(t[r][3].x + 0.5f * h12.x, t[r][3].y + 0.5f * h12.y, t[r][3].z + 0.5f * h12.z, t[r][3].w + 0.5f * h12.w);
      float* o = dw + (((size_t)k * 3 + r) * 3) * C + cg * 4;
      float4 a = *reinterpret_cast<float4*>(o);          *reinterpret_cast<float4*>(o) = add4(a, g0);
      a = *reinterpret_cast<float4*>(o + C);             *reinterpret_cast<float4*>(o + C) = add4(a, g1);
      a = *reinterpret_cast<float4*>(o + 2 * C);         *reinterpret_cast<float4*>(o + 2 * C) = add4(a, g2);
    }
  }
}

// dw[K][3][3][C] += G^T (sum over slabs of dU) G: the backward-weight reduction of gemm.hip arrives as `parts` slabs
// dU_parts[part][16][K][C] (one per split of the tile range), summed here while they are read -- in a fixed order, so the weight
// gradient is bit-reproducible and nothing has to be zero-initialised or cleaned.
// One thread per (item = (k, channel group), Winograd position): the 16 positions of an item are summed over the slabs by 16 threads
// (parts loads in flight each) and exchanged through LDS; 9 of them then apply G^T . G with the arithmetic order of
// wino_dw_transform_kernel.  (One thread per item summing 16 x parts strided float4 ran 16 - 256 workgroups: 18 us per layer at
// 1.1 TB/s, tools/kernel_traffic.sh.)
__global__ __launch_bounds__(256) void wino_dw_transform_parts_kernel(const float* __restrict__ dU, int parts, float* __restrict__ dw, int K, int C) {
  __shared__ float4 su[16][16];
  const int C4 = C / 4;
  const size_t total = (size_t)K * C4, plane = (size_t)K * C, slab = 16 * plane;
  const int il = threadIdx.x & 15, pos = threadIdx.x >> 4;
  for (size_t base = (size_t)blockIdx.x * 16; base < total; base += (size_t)gridDim.x * 16) {      // (uniform)
    const size_t i = base + il;
    const bool ok = i < total;
    const int cg = ok ? (int)(i % C4) : 0, k = ok ? (int)(i / C4) : 0;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      const float* p = dU + (size_t)pos * plane + (size_t)k * C + cg * 4;
      a = *reinterpret_cast<const float4*>(p);
#pragma unroll 4
      for (int z = 1; z < parts; ++z) a = add4(a, *reinterpret_cast<const float4*>(p + z * slab));
    }
    su[pos][il] = a;
    __syncthreads();
    if (pos < 9 && ok) {
      const int r = pos / 3, c2 = pos - 3 * r;
      float4 t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 u0 = su[c][il], u1 = su[4 + c][il], u2 = su[8 + c][il], u3 = su[12 + c][il];
        const float4 h12 = add4(u1, u2), d12 = sub4(u1, u2);
        if (r == 0) t[c] = make_float4(u0.x + 0.5f * h12.x, u0.y + 0.5f * h12.y, u0.z + 0.5f * h12.z, u0.w + 0.5f * h12.w);
        else if (r == 1) t[c] = make_float4(0.5f * d12.x, 0.5f * d12.y, 0.5f * d12.z, 0.5f * d12.w);
        else t[c] = make_float4(u3.x + 0.5f * h12.x, u3.y + 0.5f * h12.y, u3.z + 0.5f * h12.z, u3.w + 0.5f * h12.w);
      }
      const float4 h12 = add4(t[1], t[2]), d12 = sub4(t[1], t[2]);
      float4 gq;
      if (c2 == 0) gq = make_float4(t[0].x + 0.5f * h12.x, t[0].y + 0.5f * h12.y, t[0].z + 0.5f * h12.z, t[0].w + 0.5f * h12.w);
      else if (c2 == 1) gq = make_float4(0.5f * d12.x, 0.5f * d12.y, 0.5f * d12.z, 0.5f * d12.w);
      else gq = make_float4(t[3].x + 0.5f * h12.x, t[3].y + 0.5f * h12.y, t[3].z + 0.5f * h12.z, t[3].w + 0.5f * h12.w);
      float* o = dw + (((size_t)k * 3 + r) * 3 + c2) * C + cg * 4;
      *reinterpret_cast<float4*>(o) = add4(*reinterpret_cast<float4*>(o), gq);
    }
    __syncthreads();
  }
}

static unsigned wino_grid(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

hipError_t launch_wino_weight_transform(const float* w, float* U, int K, int C, int flip, hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(wino_weight_transform_kernel, dim3(wino_grid((size_t)K * (C / 4))), dim3(256), 0, st, w, U, K, C, flip);
  return hipGetLastError();
}

hipError_t launch_weight_prep(const PrepJob* jobs, int njobs, int blocks_per_job, hipStream_t st) {
  if (njobs <= 0 || njobs > 65535 || blocks_per_job <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(weight_prep_kernel, dim3((unsigned)blocks_per_job, (unsigned)njobs), dim3(256), 0, st, jobs);
  return hipGetLastError();
}

hipError_t launch_wino_input_transform(const float* x, float* V, float* Y, int N, int H, int W, int C, hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  const int TH = (H + 1) / 2, TW = (W + 1) / 2;
  const dim3 grid(wino_grid((size_t)N * TH * TW * (C / 4)));
  if (Y != nullptr) hipLaunchKernelGGL(wino_input_transform_kernel<true>, grid, dim3(256), 0, st, x, V, Y, N, H, W, C, TH, TW);
  else hipLaunchKernelGGL(wino_input_transform_kernel<false>, grid, dim3(256), 0, st, x, V, Y, N, H, W, C, TH, TW);
  return hipGetLastError();
}

hipError_t launch_wino_output_transform(const float* Mm, float* y, float* stats, const float* bias, int relu, int N, int H, int W, int K,
                                        hipStream_t st) {
  if (K % 4 != 0) return hipErrorInvalidValue;
  const int TH = (H + 1) / 2, TW = (W + 1) / 2;
  const size_t T = (size_t)N * TH * TW;
  size_t bx = (T + 15) / 16;
  const size_t cap = stats != nullptr ? 256 : 2048;       // with statistics: bound (workgroups x channels) atomics (bn.hip)
  if (bx > cap) bx = cap;
  hipLaunchKernelGGL(wino_output_transform_kernel, dim3((unsigned)bx, (K + 63) / 64), dim3(256), 0, st, Mm, y, stats, bias, relu, N, H, W, K, TH, TW);
  return hipGetLastError();
}

hipError_t launch_wino_dy_transform(const float* dy, float* Y, int N, int H, int W, int K, hipStream_t st) {
  if (K % 4 != 0) return hipErrorInvalidValue;
  const int TH = (H + 1) / 2, TW = (W + 1) / 2;
  hipLaunchKernelGGL(wino_dy_transform_kernel, dim3(wino_grid((size_t)N * TH * TW * (K / 4))), dim3(256), 0, st, dy, Y, N, H, W, K, TH, TW);
  return hipGetLastError();
}

hipError_t launch_wino_dw_transform(float* dU, float* dw, int K, int C, int clear, hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(wino_dw_transform_kernel, dim3(wino_grid((size_t)K * (C / 4))), dim3(256), 0, st, dU, dw, K, C, clear);
  return hipGetLastError();
}

hipError_t launch_wino_dw_transform_parts(const float* dU_parts, int parts, float* dw, int K, int C, hipStream_t st) {
  if (C % 4 != 0 || parts < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(wino_dw_transform_parts_kernel, dim3(wino_grid((size_t)K * (C / 4) * 16)), dim3(256), 0, st, dU_parts, parts, dw, K, C);
  return hipGetLastError();
}

}  // namespace hifihr
