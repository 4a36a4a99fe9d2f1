// Winograd F(4x4, 3x3) transforms: the same pipeline as csrc/wino.hip (weight / input transform, batched GEMMs on csrc/gemm.hip, output
// transform; backward-data on the rotated transposed filter; backward-weight as dw = G^T [ sum_t (A dy A^T) . (B^T d B) ] G) with 4x4
// output tiles: 36 products per 16 outputs instead of 16 per 4, i.e. 4x fewer multiplications than the direct convolution where
// F(2x2, 3x3) has 2.25x fewer -- the 36 batched GEMMs of a 28x28 layer do 0.56 of the work of its 16 F(2x2) ones, those of a 14x14
// layer (4x4 tiles of a map padded to 16x16) 0.73, and the transformed tensors V / M / Y' shrink by the same factors.
// Replaces (with the batched GEMM) the same cuDNN / MIOpen dispatches as conv.hip (reference network/res_encoder.py:364-373).
// fp32 throughout.  The transform constants are Lavin & Gray's for the points {0, +-1, +-2, inf}:
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Measured against an fp64 direct convolution on layer-sized problems (tools/wino43_error.py): 8-9e-6 of max |y| forward, 2-3e-6 of
// max |dw| for the weight gradient -- inside the 3e-5 / 1e-4 the convolution tests allow (F(2x2, 3x3): ~1e-6).
// Position index p = 6 * row + column; U[36][K][C], V[36][T][C], M[36][T][K], T = N * ceil(H / 4) * ceil(W / 4).
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"
#include "wino4_math.h"

namespace hifihr {

using namespace w4;

__global__ __launch_bounds__(256) void wino4_weight_transform_kernel(const float* __restrict__ w, float* __restrict__ U, int K, int C, int flip) {
  __shared__ float lds[64 * 9 * kWt4Ld];
  if (flip) {
    // flip = 1 here means: w is ALREADY the [K'][3][3][C'] transpose of the forward filter (what hifihr_weight_transpose produces); only
    // the 180-degree rotation is left to do
    const int C4 = C / 4;
    const size_t total = (size_t)K * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
      const int cg = (int)(i % C4), k = (int)(i / C4);
      V4 g[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) g[r][s] = ld4(w + (((size_t)k * 3 + (2 - r)) * 3 + (2 - s)) * C + cg * 4);
      wino4_weight_tile(g, U, (size_t)K * C, (size_t)k * C + cg * 4);
    }
    (void)lds;
  } else {
    wino4_weight_transform_body(w, U, K, C, blockIdx.x, gridDim.x);
  }
}

// thread = (tile, 4 channels); x[N][H][W][C] -> V[36][T][C].  DUAL: x is a gradient dy that backward-data (V = B^T d B of the padded 6x6
// patch) AND backward-weight (Y' = A dy A^T of the patch's central 4x4 block = this tile's outputs) both consume: one read of dy.
// MOS: the mosaic tile geometry (a template flag: with G known to be 0 the plain form keeps round 3's index arithmetic -- as a run-time
// field the extra selects cost the 28 x 28 layers' launches 1-1.5 us each)
template <bool DUAL, bool MOS>
__global__ __launch_bounds__(256) void wino4_input_transform_kernel(const float* __restrict__ x, float* __restrict__ V, float* __restrict__ Y,
                                                                   TileGeo geo, int C) {
  if (!MOS) geo.G = 0;
  const int C4 = C / 4;
  const size_t T = tile_count(geo), total = T * C4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cg = (int)(i % C4);
    const size_t t = i / C4;
    const TileAt at = tile_at(geo, t);
    const AxisBase by_ = axis_base(geo, at.y0, geo.H), bx_ = axis_base(geo, at.x0, geo.W);
    AxisPx ry[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) ry[r] = row_px(geo, at, by_, r - 1);
    V4 tt[6][6];                                            // tt = B^T d, built column by column
    V4 ty[6][4];                                            // DUAL: A dy (6 x 4) of the central block
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const AxisPx cx = col_px(geo, bx_, c - 1);
      V4 col[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        bool ok;
        const size_t px = tile_pixel(ry[r], cx, ok);
        const V4 v = ld4(x + px * C + cg * 4);
        col[r] = ok ? v : zero4();
      }
      V4 o[6];
      bt6(col, o);
#pragma unroll
      for (int r = 0; r < 6; ++r) tt[r][c] = o[r];
      if (DUAL && c >= 1 && c <= 4) {
        V4 o2[6];
        a4(col + 1, o2);
#pragma unroll
        for (int r = 0; r < 6; ++r) ty[r][c - 1] = o2[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      V4 o[6];
      bt6(tt[r], o);
#pragma unroll
      for (int c = 0; c < 6; ++c) st4(V + ((size_t)(r * 6 + c) * T + t) * C + cg * 4, o[c]);
    }
    if (DUAL) {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        V4 o[6];
        a4(ty[r], o);
#pragma unroll
        for (int c = 0; c < 6; ++c) st4(Y + ((size_t)(r * 6 + c) * T + t) * C + cg * 4, o[c]);
      }
    }
  }
}

// workgroup = 16 tile lanes x 16 float4 channel lanes (64 channels, blockIdx.y); M[36][T][K] -> y[N][H][W][K] (+ stats | bias, ReLU)
// mask (backward-data of a layer whose INPUT is a ReLU's output, round 4): y = mask > 0 ? v : 0 -- the ReLU's backward applied where its
// gradient is produced, instead of a bias_relu_bwd pass over (dy, y) in front of the previous layer's backward
template <bool MOS>
__global__ __launch_bounds__(256) void wino4_output_transform_kernel(const float* __restrict__ Mm, float* __restrict__ y, float* __restrict__ stats,
                                                                    const float* __restrict__ bias, int relu, const float* __restrict__ mask,
                                                                    TileGeo geo, int K) {
  if (!MOS) geo.G = 0;
  __shared__ double red[2][16][16][4];
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int k = blockIdx.y * 64 + cl * 4;
  const bool kok = k < K;
  const size_t T = tile_count(geo);
  Stat4 st;                                                 // batch-norm statistics of y (hifihr_internal.h "FORWARD statistics")
  if (kok) {
    const V4 bv = bias != nullptr ? ld4(bias + k) : zero4();
    const float lo = relu ? 0.f : -3.402823466e38f;
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
        for (int r = 0; r < 6; ++r) col[r] = ld4(Mm + ((size_t)(r * 6 + c) * T + t) * K + k);
        V4 o[4];
        at6(col, o);
#pragma unroll
        for (int a = 0; a < 4; ++a) s[a][c] = o[a];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const AxisPx ry = row_px(geo, at, by_, a);
        V4 o[4];
        at6(s[a], o);
        if (ry.ok) {
          size_t po[4];
          bool okb[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) po[b] = tile_pixel(ry, cx[b], okb[b]) * K + k;
          V4 mk[4];
          if (mask != nullptr) {                            // (uniform) the row's mask values requested together
#pragma unroll
            for (int b = 0; b < 4; ++b) mk[b] = ld4(mask + po[b]);
          }
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            if (okb[b]) {
              V4 v = o[b] + bv;
              v = V4{fmaxf(v.x, lo), fmaxf(v.y, lo), fmaxf(v.z, lo), fmaxf(v.w, lo)};
              if (mask != nullptr) v = V4{mk[b].x > 0.f ? v.x : 0.f, mk[b].y > 0.f ? v.y : 0.f, mk[b].z > 0.f ? v.z : 0.f, mk[b].w > 0.f ? v.w : 0.f};
              st4(y + po[b], v);
              if (a <= 1 && b <= 1) st.seed(make_float4(v.x, v.y, v.z, v.w));      // (a tile's first pixel INSIDE an image is one of these four:
              st.add(make_float4(v.x, v.y, v.z, v.w));                              //  a mosaic tile may start on a line of zeros)
            }
          }
        }
      }
    }
  }
  if (stats != nullptr)                             // uniform
    st.fold16(red, tl, cl, kok, reinterpret_cast<double*>(stats) + (size_t)(blockIdx.x & (stat_slots_used(K) - 1)) * 2 * K + k, K);
}

// backward-weight glue.  thread = (tile, 4 channels): Y'[36][T][K] = A dy A^T (dy outside the image = 0)
template <bool MOS>
__global__ __launch_bounds__(256) void wino4_dy_transform_kernel(const float* __restrict__ dy, float* __restrict__ Y, TileGeo geo, int K) {
  if (!MOS) geo.G = 0;
  const int K4 = K / 4;
  const size_t T = tile_count(geo), total = T * K4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int kg = (int)(i % K4);
    const size_t t = i / K4;
    const TileAt at = tile_at(geo, t);
    const AxisBase by_ = axis_base(geo, at.y0, geo.H), bx_ = axis_base(geo, at.x0, geo.W);
    AxisPx ry[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) ry[a] = row_px(geo, at, by_, a);
    V4 ty[6][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const AxisPx cx = col_px(geo, bx_, b);
      V4 col[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        bool ok;
        const size_t px = tile_pixel(ry[a], cx, ok);
        const V4 v = ld4(dy + px * K + kg * 4);
        col[a] = ok ? v : zero4();
      }
      V4 o[6];
      a4(col, o);
#pragma unroll
      for (int r = 0; r < 6; ++r) ty[r][b] = o[r];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      V4 o[6];
      a4(ty[r], o);
#pragma unroll
      for (int c = 0; c < 6; ++c) st4(Y + ((size_t)(r * 6 + c) * T + t) * K + kg * 4, o[c]);
    }
  }
}

// dw[K][3][3][C] += G^T (sum over slabs of dU) G; dU_parts[part][36][K][C].  thread = item (k, 4 channels): consecutive lanes read
// consecutive 16 bytes of every position plane (one 1 KiB run per wave and position), the six positions of a column are in flight
// together, G^T is applied column by column (18 intermediate values per item stay in registers), then row by row.  No LDS, no barrier:
// the first form exchanged the position sums through LDS between 16 x 16 threads and ran at 2.2 TB/s (26 us for 512 x 512 channels).
// Slabs are added in slab order: bit-reproducible, nothing to zero.
// (body + thin kernel: wino4_dw_transform_multi_kernel below runs it on a RANGE of a launch's workgroups -- item i0 first, `stride` apart)
__device__ __forceinline__ void dw_parts_items(const float* __restrict__ dU, int parts, float* __restrict__ dw, int K, int C, size_t i0, size_t stride) {
  const int C4 = C / 4;
  const size_t total = (size_t)K * C4, plane = (size_t)K * C, slab = 36 * plane;
  for (size_t i = i0; i < total; i += stride) {
    const int cg = (int)(i % C4), k = (int)(i / C4);
    const float* p = dU + (size_t)k * C + cg * 4;
    float* const out = dw + (size_t)k * 9 * C + cg * 4;
    V4 old[9];                                                // the read half of dw += ...: issued with the slab loads
#pragma unroll
    for (int q = 0; q < 9; ++q) old[q] = ld4(out + (size_t)q * C);
    V4 t[3][6];                                               // t = G^T u
#pragma unroll
    for (int jc = 0; jc < 6; ++jc) {
      V4 col[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] = ld4(p + (size_t)(r * 6 + jc) * plane);
      for (int z = 1; z < parts; ++z) {
#pragma unroll
        for (int r = 0; r < 6; ++r) col[r] = col[r] + ld4(p + (size_t)z * slab + (size_t)(r * 6 + jc) * plane);
      }
      V4 o[3];
      gt6(col, o);
      t[0][jc] = o[0]; t[1][jc] = o[1]; t[2][jc] = o[2];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      V4 o[3];
      gt6(t[r], o);
#pragma unroll
      for (int c = 0; c < 3; ++c) st4(out + (size_t)(r * 3 + c) * C, old[r * 3 + c] + o[c]);
    }
  }
}

__global__ __launch_bounds__(256) void wino4_dw_transform_parts_kernel(const float* __restrict__ dU, int parts, float* __restrict__ dw, int K, int C) {
  dw_parts_items(dU, parts, dw, K, C, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

// The same transform for SMALL layers (128 x 128 channels: 4 096 items = 16 workgroups of the kernel above, every thread walking 36
// positions x `parts` slabs as 6 x parts dependent load batches: 18 us of latency for 17 MB, three times per step).  Wide form: a
// workgroup is 32 items x 6 position columns; thread (item, jc) sums column jc over the slabs with its loads in flight together, applies
// G^T down the column and leaves the three values in LDS; threads jc < 3 then finish row jc of the item.  Slabs are still added in slab
// order (bit-reproducible).
// (body: workgroup `blk` of the layer, threads 0 .. 191 of a workgroup that may be larger; every thread of the workgroup reaches the barrier)
__device__ __forceinline__ void dw_parts_wide_block(const float* __restrict__ dU, int parts, float* __restrict__ dw, int K, int C, int blk,
                                                    V4 (*s_t)[3][6]) {
  const int C4 = C / 4;
  const size_t total = (size_t)K * C4, plane = (size_t)K * C, slab = 36 * plane;
  const int il = threadIdx.x & 31, jc = threadIdx.x >> 5;
  const size_t i = (size_t)blk * 32 + il;
  const bool live = i < total && jc < 6;
  const int cg = live ? (int)(i % C4) : 0, k = live ? (int)(i / C4) : 0;
  const float* p = dU + (size_t)k * C + cg * 4;
  float* const out = dw + (size_t)k * 9 * C + cg * 4;
  V4 old[3];
  if (live && jc < 3) {
#pragma unroll
    for (int c = 0; c < 3; ++c) old[c] = ld4(out + (size_t)(jc * 3 + c) * C);
  }
  if (live) {
    V4 col[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) col[r] = ld4(p + (size_t)(r * 6 + jc) * plane);
    int z = 1;
    for (; z + 1 < parts; z += 2) {                           // two slabs per trip: twelve loads in flight
      V4 a[6], b[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        a[r] = ld4(p + (size_t)z * slab + (size_t)(r * 6 + jc) * plane);
        b[r] = ld4(p + (size_t)(z + 1) * slab + (size_t)(r * 6 + jc) * plane);
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] = (col[r] + a[r]) + b[r];
    }
    for (; z < parts; ++z) {
#pragma unroll
      for (int r = 0; r < 6; ++r) col[r] = col[r] + ld4(p + (size_t)z * slab + (size_t)(r * 6 + jc) * plane);
    }
    V4 o[3];
    gt6(col, o);
    s_t[il][0][jc] = o[0]; s_t[il][1][jc] = o[1]; s_t[il][2][jc] = o[2];
  }
  __syncthreads();
  if (live && jc < 3) {
    V4 t[6], o[3];
#pragma unroll
    for (int c = 0; c < 6; ++c) t[c] = s_t[il][jc][c];
    gt6(t, o);
#pragma unroll
    for (int c = 0; c < 3; ++c) st4(out + (size_t)(jc * 3 + c) * C, old[c] + o[c]);
  }
}
__global__ __launch_bounds__(192) void wino4_dw_transform_parts_wide_kernel(const float* __restrict__ dU, int parts, float* __restrict__ dw, int K, int C) {
  __shared__ V4 s_t[32][3][6];
  dw_parts_wide_block(dU, parts, dw, K, C, (int)blockIdx.x, s_t);
}

// The weight-gradient transforms of SEVERAL layers in ONE launch (round 6).  dw is needed by the optimizer only, so a step may collect the
// (slabs, target) pairs of its F(4x4) layers during backward and run them all here, right before Adam: ten 5-11 us launches that sat
// between the backward products at 3.9 TB/s (each a prologue and two or three trips of loads: hifihr_wino_dw_transform_parts_m) become
// one stream over the same 298 MB.  The jobs travel BY VALUE in the kernel arguments (no device table: nothing to upload, nothing a
// captured graph could find stale); job j owns workgroups [wg0_j, wg0_{j+1}); layers of <= 8 192 items keep the wide form above.
struct DwJob {
  const float* dU;
  float* dw;
  int parts, K, C, wg0;
};
constexpr int kDwMaxJobs = 24;
struct DwJobs {
  DwJob j[kDwMaxJobs];
  int n;
};
__host__ __device__ inline bool dw_job_wide(int K, int C) { return (size_t)K * (C / 4) <= 8192; }
__global__ __launch_bounds__(256) void wino4_dw_transform_multi_kernel(DwJobs js) {
  __shared__ V4 s_t[32][3][6];
  int jb = 0;
  for (int q = 1; q < js.n; ++q)
    if ((int)blockIdx.x >= js.j[q].wg0) jb = q;             // (uniform; wg0 ascending)
  const DwJob job = js.j[jb];
  const int wb = (int)blockIdx.x - job.wg0;
  if (dw_job_wide(job.K, job.C)) dw_parts_wide_block(job.dU, job.parts, job.dw, job.K, job.C, wb, s_t);
  else dw_parts_items(job.dU, job.parts, job.dw, job.K, job.C, (size_t)wb * 256 + threadIdx.x, ~(size_t)0 >> 1);   // (one item per thread)
}

static unsigned wino4_grid(size_t total) {
  size_t b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

hipError_t launch_wino4_weight_transform(const float* w, float* U, int K, int C, int flip, hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(wino4_weight_transform_kernel, dim3(wino4_grid((size_t)K * (C / 4))), dim3(256), 0, st, w, U, K, C, flip);
  return hipGetLastError();
}

// the tile geometry of a layer: every launcher of the F(4x4, 3x3) pipeline and the products' row count come through here
TileGeo wino4_geo(int N, int H, int W) {
  static const int on = [] { const char* e = getenv("HIFIHR_WINO_MOSAIC"); return e ? atoi(e) : 1; }();
  return make_tile_geo(N, H, W, on != 0);
}
long wino4_tiles(int N, int H, int W) { return (long)tile_count(wino4_geo(N, H, W)); }
long wino4_tiles_real(int N, int H, int W) {
  const TileGeo g = wino4_geo(N, H, W);
  return g.G ? (long)(g.N / (g.G * g.G)) * g.TH * g.TW : (long)tile_count(g);
}

hipError_t launch_wino4_input_transform(const float* x, float* V, float* Y, int N, int H, int W, int C, hipStream_t st) {
  if (C % 4 != 0) return hipErrorInvalidValue;
  const TileGeo geo = wino4_geo(N, H, W);
  const size_t total = tile_count(geo) * (C / 4);
  if (geo.G) {
    if (Y != nullptr) hipLaunchKernelGGL((wino4_input_transform_kernel<true, true>), dim3(wino4_grid(total)), dim3(256), 0, st, x, V, Y, geo, C);
    else hipLaunchKernelGGL((wino4_input_transform_kernel<false, true>), dim3(wino4_grid(total)), dim3(256), 0, st, x, V, Y, geo, C);
  } else {
    if (Y != nullptr) hipLaunchKernelGGL((wino4_input_transform_kernel<true, false>), dim3(wino4_grid(total)), dim3(256), 0, st, x, V, Y, geo, C);
    else hipLaunchKernelGGL((wino4_input_transform_kernel<false, false>), dim3(wino4_grid(total)), dim3(256), 0, st, x, V, Y, geo, C);
  }
  return hipGetLastError();
}

hipError_t launch_wino4_output_transform(const float* Mm, float* y, float* stats, const float* bias, int relu, const float* mask, int N, int H, int W, int K,
                                         hipStream_t st) {
  if (K % 4 != 0) return hipErrorInvalidValue;
  const TileGeo geo = wino4_geo(N, H, W);
  const size_t T = tile_count(geo);
  size_t bx = (T + 15) / 16;
  if (bx > 1024) bx = 1024;
  if (geo.G) hipLaunchKernelGGL(wino4_output_transform_kernel<true>, dim3((unsigned)bx, (K + 63) / 64), dim3(256), 0, st, Mm, y, stats, bias, relu, mask, geo, K);
  else hipLaunchKernelGGL(wino4_output_transform_kernel<false>, dim3((unsigned)bx, (K + 63) / 64), dim3(256), 0, st, Mm, y, stats, bias, relu, mask, geo, K);
  return hipGetLastError();
}

hipError_t launch_wino4_dy_transform(const float* dy, float* Y, int N, int H, int W, int K, hipStream_t st) {
  if (K % 4 != 0) return hipErrorInvalidValue;
  const TileGeo geo = wino4_geo(N, H, W);
  if (geo.G) hipLaunchKernelGGL(wino4_dy_transform_kernel<true>, dim3(wino4_grid(tile_count(geo) * (K / 4))), dim3(256), 0, st, dy, Y, geo, K);
  else hipLaunchKernelGGL(wino4_dy_transform_kernel<false>, dim3(wino4_grid(tile_count(geo) * (K / 4))), dim3(256), 0, st, dy, Y, geo, K);
  return hipGetLastError();
}

hipError_t launch_wino4_dw_transform_multi(const WinoDwJob* jobs, int njobs, hipStream_t st) {
  for (int base = 0; base < njobs; base += kDwMaxJobs) {
    DwJobs js;
    js.n = njobs - base < kDwMaxJobs ? njobs - base : kDwMaxJobs;
    long wg = 0;
    for (int q = 0; q < js.n; ++q) {
      const WinoDwJob& in = jobs[base + q];
      if (in.dU == nullptr || in.dw == nullptr || in.parts < 1 || in.K <= 0 || in.C < 4 || in.C % 4 != 0) return hipErrorInvalidValue;
      const size_t total = (size_t)in.K * (in.C / 4);
      js.j[q] = DwJob{in.dU, in.dw, in.parts, in.K, in.C, (int)wg};
      wg += (long)(dw_job_wide(in.K, in.C) ? (total + 31) / 32 : (total + 255) / 256);
      if (wg >= (1L << 30)) return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(wino4_dw_transform_multi_kernel, dim3((unsigned)wg), dim3(256), 0, st, js);
  }
  return hipGetLastError();
}

hipError_t launch_wino4_dw_transform_parts(const float* dU_parts, int parts, float* dw, int K, int C, hipStream_t st) {
  if (C % 4 != 0 || parts < 1) return hipErrorInvalidValue;
  const size_t total = (size_t)K * (C / 4);
  static const int wide_on = [] { const char* e = getenv("HIFIHR_WINO_DW_WIDE"); return e ? atoi(e) : 1; }();
  if (wide_on && total <= 8192)
    hipLaunchKernelGGL(wino4_dw_transform_parts_wide_kernel, dim3((unsigned)((total + 31) / 32)), dim3(192), 0, st, dU_parts, parts, dw, K, C);
  else
    hipLaunchKernelGGL(wino4_dw_transform_parts_kernel, dim3(wino4_grid(total)), dim3(256), 0, st, dU_parts, parts, dw, K, C);
  return hipGetLastError();
}

}  // namespace hifihr
