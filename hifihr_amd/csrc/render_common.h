// Shared between csrc/render.hip (vertex stage, binning, forward tile kernel) and csrc/render_bwd.hip (backward kernels): constants,
// workspace layout, TexturesUV sampling.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "hifihr_internal.h"
#include "render_math.h"

namespace hifihr {

constexpr int kTile = 16;            // output pixels per tile edge of the backward kernel (the forward uses 8: render_tile())
constexpr int kFaceRec = 12;         // float4 per packed face record: NDC, position, unit normal, colour of the three corners

__device__ __forceinline__ float wsum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}

// value of lane + D inside the 16-lane row (0 past the end of the row): one VALU instruction with a DPP operand on the device
template <int D>
__device__ __forceinline__ float row_shl_f32(float v) {
#if defined(HIFIHR_HOSTSIM)
  const float t = __shfl_down(v, D, 64);
  return ((::hostsim::lane_id() & 15) + D < 16) ? t : 0.f;
#else
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + D, 0xf, 0xf, true));
#endif
}
// value of lane - 1 inside the 16-lane row (`fill` at the start of a row)
__device__ __forceinline__ int row_shr1_i32(int v, int fill) {
#if defined(HIFIHR_HOSTSIM)
  const int t = __shfl_up(v, 1, 64);
  return (::hostsim::lane_id() & 15) >= 1 ? t : fill;
#else
  return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false);
#endif
}

#if defined(HIFIHR_HOSTSIM)
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float fast_rsq(float x) { return 1.0f / sqrtf(x); }
#else
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
#endif

// ------------------------------------------------------------------------------------------------
// TexturesUV (PyTorch3D renderer/mesh/textures.py TexturesUV.sample_textures [recalled]; reference models_res_nimble.py:203-208 hands the
// NIMBLE texture image to the renderer this way): per sample, uv = sum_k bary_k uv[faces_uvs[f][k]] with the rasteriser's
// perspective-corrected barycentrics, then F.grid_sample(flip(maps, vertical), 2 uv - 1, bilinear, align_corners=True, padding border).
// Fused into the tile kernels' per-sample shading (template flag UV: the vertex-colour instantiations are unchanged): the forward
// interpolates the face's three uvs with the barycentrics it already has and fetches four texels; the backward scatters d loss / d texel
// into the texture (float atomics) and adds d texel / d uv . uv_k to the barycentric gradient, so the path to the vertices runs through the
// same bary_bwd and per-vertex accumulators as every other attribute.
// ------------------------------------------------------------------------------------------------
struct UvSample { int x0, x1, y0, y1; float wx, wy; bool in_x, in_y; };      // rows are those of the UNFLIPPED map
__device__ __forceinline__ UvSample uv_sample(float u, float v, int TH, int TW) {
  UvSample q;
  float ix = ((2.f * u - 1.f) + 1.f) * 0.5f * (float)(TW - 1);                 // grid_sample, align_corners = True
  float iy = ((2.f * v - 1.f) + 1.f) * 0.5f * (float)(TH - 1);                 // row of the flipped map
  q.in_x = ix >= 0.f && ix <= (float)(TW - 1);                                   // border padding: coordinates clipped (zero gradient outside)
  q.in_y = iy >= 0.f && iy <= (float)(TH - 1);
  ix = fminf(fmaxf(ix, 0.f), (float)(TW - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(TH - 1));
  const float fx = floorf(ix), fy = floorf(iy);
  q.wx = ix - fx; q.wy = iy - fy;
  q.x0 = (int)fx; q.x1 = min(q.x0 + 1, TW - 1);
  const int r0 = (int)fy, r1 = min(r0 + 1, TH - 1);
  q.y0 = TH - 1 - r0; q.y1 = TH - 1 - r1;                                        // un-flip
  return q;
}

struct TexUvDev {
  const int* faces_uvs;        // [F][3]
  const float* verts_uvs;      // [Vt][2]
  const float* maps;           // [B][TH][TW][3]
  float* gmaps;                // [B][TH][TW][3] (backward; accumulated into) or null
  int TH, TW;
};

// bilinear texel at (u, v) and, if asked, its derivatives with respect to the (clamped) pixel coordinates ix, iy
__device__ __forceinline__ void uv_fetch(const TexUvDev& t, int b, const UvSample& q, float (&T)[3], float* dix, float* diy) {
  const float* m = t.maps + (size_t)b * t.TH * t.TW * 3;
  const float* p00 = m + ((size_t)q.y0 * t.TW + q.x0) * 3;
  const float* p01 = m + ((size_t)q.y0 * t.TW + q.x1) * 3;
  const float* p10 = m + ((size_t)q.y1 * t.TW + q.x0) * 3;
  const float* p11 = m + ((size_t)q.y1 * t.TW + q.x1) * 3;
  const float w00 = (1.f - q.wx) * (1.f - q.wy), w01 = q.wx * (1.f - q.wy), w10 = (1.f - q.wx) * q.wy, w11 = q.wx * q.wy;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v00 = p00[c], v01 = p01[c], v10 = p10[c], v11 = p11[c];
    T[c] = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
    if (dix != nullptr) {
      dix[c] = (v01 - v00) * (1.f - q.wy) + (v11 - v10) * q.wy;
      diy[c] = (v10 - v00) * (1.f - q.wx) + (v11 - v01) * q.wx;
    }
  }
}


// workspace: four float4[B][V] vertex arrays, the float[B][V][12] gradient records of the backward, the forward's per-tile face lists
// int cnt[B][tiles^2] and int list[B][tiles^2][F] (worst case: the whole mesh inside one tile; sized for the 8-pixel grid), then the
// packed face records float4[B][F][kFaceRec] (written by render_bin_kernel, read by both tile kernels)
// (+ 8 floats per image behind the gradient records: the backward's light-colour / light-direction accumulators [B][3] + [B][3])
static inline size_t vertex_part_bytes(const RenderDev& r, int B) {
  return (size_t)B * r.V * (4 * sizeof(float4) + 12 * sizeof(float)) + (size_t)B * 8 * sizeof(float);
}
static inline float* light_records(const RenderDev& r, int B, void* ws) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (size_t)B * r.V * (4 * sizeof(float4) + 12 * sizeof(float)));
}
// The backward ACCUMULATES into the gradient records and the light accumulators (atomics): they must start at zero.  The forward's vertex
// kernel zeroes them (every backward follows a forward on the same workspace: the face records it reads are the forward's), and says so
// here; a backward that finds its workspace not marked -- a second backward after one forward -- clears them itself (two memsets).
void render_ws_mark_clean(void* ws, bool clean);
bool render_ws_take_clean(void* ws);

static inline int render_tile() {                // tile edge of the forward (8: see render_fwd2_kernel; HIFIHR_RENDER_TILE=16 for the A/B)
  static const int v = [] { const char* e = getenv("HIFIHR_RENDER_TILE"); return (e && atoi(e) == 16) ? 16 : 8; }();
  return v;
}
static inline size_t list_part_bytes(const RenderDev& r, int B) {
  const size_t tiles = (size_t)((r.H + 7) / 8) * ((r.H + 7) / 8);                 // the finest grid either form uses
  return ((size_t)B * tiles * sizeof(int) * (1 + (size_t)r.F) + 255) / 256 * 256;
}
// ... then the packed face records float4[B][F][kFaceRec] (written by render_bin_kernel, read by both tile kernels)
static inline float4* face_records(const RenderDev& r, int B, void* ws) {
  return reinterpret_cast<float4*>(reinterpret_cast<char*>(ws) + vertex_part_bytes(r, B) + list_part_bytes(r, B));
}
static inline void carve(const RenderDev& r, int B, void* ws, float4** vndc, float4** vpos, float4** vnrm, float4** vcol, float** gvrec,
                  int** tile_cnt = nullptr, int** tile_list = nullptr, int tile_edge = kTile) {
  float4* p = reinterpret_cast<float4*>(ws);
  const size_t n = (size_t)B * r.V;
  *vndc = p; *vpos = p + n; *vnrm = p + 2 * n; *vcol = p + 3 * n;
  *gvrec = reinterpret_cast<float*>(p + 4 * n);
  if (tile_cnt != nullptr) {
    const size_t tiles = (size_t)((r.H + tile_edge - 1) / tile_edge) * ((r.H + tile_edge - 1) / tile_edge);
    *tile_cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + vertex_part_bytes(r, B));
    *tile_list = *tile_cnt + (size_t)B * tiles;
  }
}


}  // namespace hifihr
