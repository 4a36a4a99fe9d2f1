// Winograd F(4x4, 3x3) transform arithmetic shared by csrc/wino4.hip (the transform kernels) and csrc/wino.hip (weight_prep_kernel, which
// re-lays every convolution weight of a step out in one launch).  Not part of the ABI.  Constants: see the header of wino4.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace hifihr {
namespace w4 {

struct V4 {
  float x, y, z, w;
};

// ------------------------------------------------------------------------------------------------
// Tile geometry of the F(4x4, 3x3) pipeline.  Plain: image n is cut into ceil(H / 4) x ceil(W / 4) tiles of its own -- a 14 x 14 map
// (ResNet layers 3-4: 60 % of the trunk's multiplications) pays for a 16 x 16 one, 23 % of the batched products and of V / M / Y' are
// padding.  MOSAIC (round 4; G = 4): G x G images are laid out as ONE map with a single line of zeros between neighbours (stride H + 1:
// the line is the bottom padding of one image and the top padding of the next, so every pixel still sees its own zero border) and the
// 4 x 4 tiles are cut from that map: (4 (H + 1) / 4)^2 = 225 tiles per 16 images at 14 x 14 instead of 256 (-12 %), 196 instead of 256 at
// 13 x 13.  Only the tile -> pixel mapping of the transform kernels changes (tile_at / axis_px below); the products see fewer rows.
// Used when it saves tiles: H == W, H % 4 in {1, 2}, N a multiple of 16 (HIFIHR_WINO_MOSAIC=0 switches it off).
// ------------------------------------------------------------------------------------------------
struct TileGeo {
  int N, H, W, TH, TW, G;                                   // G = 0: plain; else images per mosaic side.  TH x TW tiles per image / per mosaic
};
__host__ __device__ inline bool tile_mosaic_shape(int N, int H, int W) { return H == W && (H % 4 == 1 || H % 4 == 2) && N % 16 == 0 && N > 0; }
__host__ __device__ inline TileGeo make_tile_geo(int N, int H, int W, bool mosaic) {
  TileGeo g{N, H, W, (H + 3) / 4, (W + 3) / 4, 0};
  if (mosaic && tile_mosaic_shape(N, H, W)) { g.G = 4; g.TH = H + 1; g.TW = W + 1; }          // 4 (H + 1) / 4 tiles per side
  return g;
}
// rows of V / M / Y' per position.  Mosaic: rounded up to a multiple of 32 (the backward-weight products walk the tiles 32 at a time);
// the tiles past the last mosaic belong to no image: the input transforms write zeros for them, the output transforms nothing.
__host__ __device__ inline size_t tile_count(const TileGeo& g) {
  if (!g.G) return (size_t)g.N * g.TH * g.TW;
  const size_t t = (size_t)(g.N / (g.G * g.G)) * g.TH * g.TW;
  return (t + 31) / 32 * 32;
}
struct TileAt { int n, y0, x0; };                           // first image (mosaic: of the group), output coordinate of the tile's first pixel
__device__ __forceinline__ TileAt tile_at(const TileGeo& g, size_t t) {
  const int tw = (int)(t % g.TW), th = (int)((t / g.TW) % g.TH), q = (int)(t / ((size_t)g.TW * g.TH));
  return TileAt{g.G ? q * g.G * g.G : q, 4 * th, 4 * tw};
}
// A pixel of the patch = (row part) + (column part): ok flags and pixel-index offsets prepared ONCE per tile row / column (one integer
// division per axis and tile in the mosaic form, none in the plain form), so that the 36 pixels of a patch cost one addition and one AND
// each.  (The first version resolved every pixel through the mosaic arithmetic: 1-3 us per transform launch, more than the mosaic saved.)
struct AxisPx { bool ok; int off; };                        // rows: ((n0 + img G) H + y) W;  columns: img H W + x   (pixel index = row.off + col.off)
struct AxisBase { int i0, p0; };                            // image index and in-image coordinate of the tile's first OUTPUT row / column
__device__ __forceinline__ AxisBase axis_base(const TileGeo& g, int m0, int L) {
  if (!g.G) return AxisBase{0, m0};
  const int i = m0 / (L + 1);
  return AxisBase{i, m0 - i * (L + 1)};
}
// d places after the tile's first row (d = -1 .. 4: the 6-wide input patch; 0 .. 3: the tile's outputs)
__device__ __forceinline__ AxisPx row_px(const TileGeo& g, const TileAt& a, const AxisBase& b, int d) {
  int p = b.p0 + d, i = b.i0;
  if (g.G && p > g.H) { p -= g.H + 1; ++i; }                // past this image's zero line: the next image (d <= 4 < H + 1: at most once)
  const bool ok = p >= 0 && p < g.H && i < (g.G ? g.G : 1) && a.n < g.N;     // (a.n >= N: the padding tiles behind the last mosaic)
  return AxisPx{ok, ok ? ((a.n + i * g.G) * g.H + p) * g.W : 0};
}
__device__ __forceinline__ AxisPx col_px(const TileGeo& g, const AxisBase& b, int d) {
  int p = b.p0 + d, i = b.i0;
  if (g.G && p > g.W) { p -= g.W + 1; ++i; }
  const bool ok = p >= 0 && p < g.W && i < (g.G ? g.G : 1);
  return AxisPx{ok, ok ? i * g.H * g.W + p : 0};
}
__device__ __forceinline__ size_t tile_pixel(const AxisPx& r, const AxisPx& c, bool& ok) {
  ok = r.ok && c.ok;
  return ok ? (size_t)(r.off + c.off) : (size_t)0;
}
__device__ __forceinline__ V4 operator+(const V4& a, const V4& b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ V4 operator-(const V4& a, const V4& b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
__device__ __forceinline__ V4 operator*(float s, const V4& a) { return V4{s * a.x, s * a.y, s * a.z, s * a.w}; }
__device__ __forceinline__ V4 ld4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return V4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4(float* p, const V4& v) { *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ V4 zero4() { return V4{0.f, 0.f, 0.f, 0.f}; }
// element-type generic access: T = V4 (four consecutive channels per thread) or float (one channel per thread: four times the threads,
// a quarter of the serial work each -- the small layers' transform launches are latency-bound, not bandwidth-bound)
template <typename T> __device__ __forceinline__ T ldT(const float* p);
template <> __device__ __forceinline__ V4 ldT<V4>(const float* p) { return ld4(p); }
template <> __device__ __forceinline__ float ldT<float>(const float* p) { return *p; }
__device__ __forceinline__ void stT(float* p, const V4& v) { st4(p, v); }
__device__ __forceinline__ void stT(float* p, float v) { *p = v; }
template <typename T> __device__ __forceinline__ T zeroT();
template <> __device__ __forceinline__ V4 zeroT<V4>() { return zero4(); }
template <> __device__ __forceinline__ float zeroT<float>() { return 0.f; }
template <typename T> struct VecWidth;
template <> struct VecWidth<V4> { static constexpr int n = 4; };
template <> struct VecWidth<float> { static constexpr int n = 1; };
__device__ __forceinline__ V4 operator*(const V4& a, const V4& b) { return V4{a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }

// o[6] = B^T x[6]
template <typename T>
__device__ __forceinline__ void bt6(const T* x, T* o) {
  o[0] = 4.f * x[0] - 5.f * x[2] + x[4];
  o[1] = x[3] + x[4] - 4.f * (x[1] + x[2]);
  o[2] = 4.f * (x[1] - x[2]) - x[3] + x[4];
  o[3] = 2.f * (x[3] - x[1]) - x[2] + x[4];
  o[4] = 2.f * (x[1] - x[3]) - x[2] + x[4];
  o[5] = 4.f * x[1] - 5.f * x[3] + x[5];
}
// o[4] = A^T x[6]
template <typename T>
__device__ __forceinline__ void at6(const T* x, T* o) {
  const T s12 = x[1] + x[2], d12 = x[1] - x[2], s34 = x[3] + x[4], d34 = x[3] - x[4];
  o[0] = x[0] + s12 + s34;
  o[1] = d12 + 2.f * d34;
  o[2] = s12 + 4.f * s34;
  o[3] = d12 + 8.f * d34 + x[5];
}
// o[6] = G x[3]
template <typename T>
__device__ __forceinline__ void g3(const T* x, T* o) {
  const T s02 = x[0] + x[2];
  o[0] = 0.25f * x[0];
  o[1] = (-1.f / 6.f) * (s02 + x[1]);
  o[2] = (-1.f / 6.f) * (s02 - x[1]);
  const T a = (1.f / 24.f) * x[0] + (1.f / 6.f) * x[2], b = (1.f / 12.f) * x[1];
  o[3] = a + b;
  o[4] = a - b;
  o[5] = x[2];
}
// o[6] = A x[4]   (A = (A^T)^T)
template <typename T>
__device__ __forceinline__ void a4(const T* x, T* o) {
  const T s02 = x[0] + x[2], s13 = x[1] + x[3];
  const T e = x[0] + 4.f * x[2], f = 2.f * x[1] + 8.f * x[3];
  o[0] = x[0];
  o[1] = s02 + s13;
  o[2] = s02 - s13;
  o[3] = e + f;
  o[4] = e - f;
  o[5] = x[3];
}
// o[3] = G^T x[6]
template <typename T>
__device__ __forceinline__ void gt6(const T* x, T* o) {
  const T s12 = x[1] + x[2], d12 = x[2] - x[1], s34 = x[3] + x[4], d34 = x[3] - x[4];
  o[0] = 0.25f * x[0] + (-1.f / 6.f) * s12 + (1.f / 24.f) * s34;
  o[1] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
  o[2] = (-1.f / 6.f) * s12 + (1.f / 6.f) * s34 + x[5];
}

// u = G g G^T for one 3x3 filter (4 values per tap), stored to U[p][...]: U + p * plane + off
__device__ __forceinline__ void wino4_weight_tile(const V4 (&g)[3][3], float* __restrict__ U, size_t plane, size_t off) {
  V4 t[6][3];                                               // t = G g: column by column
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const V4 col[3] = {g[0][s], g[1][s], g[2][s]};
    V4 o[6];
    g3(col, o);
#pragma unroll
    for (int r = 0; r < 6; ++r) t[r][s] = o[r];
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    V4 o[6];
    g3(t[r], o);
#pragma unroll
    for (int c = 0; c < 6; ++c) st4(U + (size_t)(r * 6 + c) * plane + off, o[c]);
  }
}

// thread = (output channel k, 4 input channels); w[K][3][3][C] -> U[36][K][C]
__device__ __forceinline__ void wino4_weight_transform_body(const float* __restrict__ w, float* __restrict__ U, int K, int C, unsigned bid, unsigned nblk) {
  const int C4 = C / 4;
  const size_t total = (size_t)K * C4;
  for (size_t i = (size_t)bid * 256 + threadIdx.x; i < total; i += (size_t)nblk * 256) {
    const int cg = (int)(i % C4), k = (int)(i / C4);
    V4 g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) g[r][s] = ld4(w + (((size_t)k * 3 + r) * 3 + s) * C + cg * 4);
    wino4_weight_tile(g, U, (size_t)K * C, (size_t)k * C + cg * 4);
  }
}

// backward-data weights straight from w[K][3][3][C]: U'[36][C][K] = G g' G^T, g'[c][r][s][k] = w[k][2-r][2-s][c]; a (64 k) x (16 c) block
// of w is staged through LDS (see wino_weight_transform_t_body in wino.hip)
constexpr int kWt4Ld = 17;
__device__ __forceinline__ void wino4_weight_transform_t_body(const float* __restrict__ w, float* __restrict__ U, int K, int C, unsigned bid, unsigned nblk,
                                              float* __restrict__ lds /* [64 * 9 * kWt4Ld] */) {
  const int kt = (K + 63) / 64, ct = (C + 15) / 16;
  for (int tile = (int)bid; tile < kt * ct; tile += (int)nblk) {
    const int k0 = (tile % kt) * 64, c0 = (tile / kt) * 16;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 9 * 4; i += 256) {
      const int q = i & 3, row = i >> 2;
      const int k = k0 + row / 9, c = c0 + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < K && c < C) v = *reinterpret_cast<const float4*>(w + ((size_t)k * 9 + row % 9) * C + c);
      float* d = lds + row * kWt4Ld + q * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int kg = threadIdx.x & 15, cl = threadIdx.x >> 4;
    const int k = k0 + kg * 4, c = c0 + cl;
    if (k < K && c < C) {
      V4 g[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
          const float* p = lds + ((kg * 4) * 9 + (2 - r) * 3 + (2 - s2)) * kWt4Ld + cl;
          g[r][s2] = V4{p[0], p[9 * kWt4Ld], p[18 * kWt4Ld], p[27 * kWt4Ld]};
        }
      wino4_weight_tile(g, U, (size_t)C * K, (size_t)c * K + k);
    }
  }
}


}  // namespace w4
}  // namespace hifihr
