// Hard rasteriser + Phong shader + aa x aa resolve for gfx950, forward and backward.
//
// Replaces the reference's renderer_p3d(...) call and the avg_pool2d after it (reference
// models_res_nimble.py:208-211), i.e. PyTorch3D's rasterize_meshes (coarse+fine CUDA kernels),
// interpolate_face_attributes x3, ~25 elementwise lighting launches over [B,672,672,*] tensors,
// hard_rgb_blend and the pooling, with:
//   render_vertex_kernel      per vertex: NDC projection, area-weighted vertex normal (CSR gather), packing
//   render_bin_kernel         per (face, image): the face's screen bounding box against the tile grid; the face id is appended
//                             to the list of every 16x16-pixel tile it overlaps (same predicate the tiles used to evaluate for
//                             ALL faces: 196 tiles x 1538 faces x 3 gathers per image, most of the forward's time)
//   render_fwd2_kernel        one 256-thread workgroup per 8x8 output-pixel tile (24x24 samples at aa=3): the tile's face list is
//                             staged in LDS, (face, pixel) candidates are enumerated densely over the lanes, conservative rejects
//                             (stage A) feed ONE survivor queue per workgroup, the exact sample tests (stage B) run on dense waves of
//                             it (nearest depth, ties keep the lower face index -- a 64-bit LDS atomicMin on (depth, face id), so the
//                             list order does not matter), then (busy pixel, sample row) items are shaded and resolved; the
//                             per-sample face id side buffer (4 B/sample) is the only per-sample HBM traffic.  A tile with an empty
//                             list writes background and leaves.
//   render_bwd_kernel         (csrc/render_bwd.hip) 16x16 tiles, no rasterisation: reads the face ids, recomputes barycentrics and
//                             shading, back-propagates to per-vertex records with float atomics
//   render_vertex_bwd_kernel  per vertex: folds NDC / position / normal gradients into d(verts)
// Rounding-sensitive maths lives in render_math.h and matches oracle/raster_oracle.c operation for operation.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "render_common.h"

namespace hifihr {

constexpr int kRecW = 11;            // floats per face record in LDS: 3 x (x, y), 3 z, face id, candidate rectangle (odd pitch: no bank conflicts
                                     // between lanes on different faces)

// ------------------------------------------------------------------------------------------------
// vertex stage
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void render_vertex_kernel(RenderDev r, const float* __restrict__ verts,
                                                           const float* __restrict__ vcolors, long vcol_bstride,
                                                           const float* __restrict__ cam, float4* __restrict__ vndc,
                                                           float4* __restrict__ vpos, float4* __restrict__ vnrm,
                                                           float4* __restrict__ vcol, int* __restrict__ tile_cnt, int ntiles) {
  const int b = blockIdx.y;
  const int v = blockIdx.x * 256 + threadIdx.x;
  // the per-tile face counters of this image start at zero for render_bin_kernel (the workspace arrives uninitialised)
  for (int t = v; t < ntiles; t += gridDim.x * 256) tile_cnt[(size_t)b * ntiles + t] = 0;
  if (v >= r.V) return;
  const float* vb = verts + (size_t)b * r.V * 3;
  const float X = vb[3 * v], Y = vb[3 * v + 1], Z = vb[3 * v + 2];
  const float fx = cam[4 * b], fy = cam[4 * b + 1], px = cam[4 * b + 2], py = cam[4 * b + 3];
  const size_t o = (size_t)b * r.V + v;
  vndc[o] = make_float4((X * fx + Z * px) / Z, (Y * fy + Z * py) / Z, Z, 0.f);
  vpos[o] = make_float4(X, Y, Z, 0.f);
  float s[3] = {0.f, 0.f, 0.f};
  for (int e = r.vf_off[v]; e < r.vf_off[v + 1]; ++e) {
    const int f = r.vf_idx[e] >> 2;
    const int i0 = r.faces[3 * f], i1 = r.faces[3 * f + 1], i2 = r.faces[3 * f + 2];
    const float a[3] = {vb[3 * i2] - vb[3 * i1], vb[3 * i2 + 1] - vb[3 * i1 + 1], vb[3 * i2 + 2] - vb[3 * i1 + 2]};
    const float c[3] = {vb[3 * i0] - vb[3 * i1], vb[3 * i0 + 1] - vb[3 * i1 + 1], vb[3 * i0 + 2] - vb[3 * i1 + 2]};
    s[0] += a[1] * c[2] - a[2] * c[1];
    s[1] += a[2] * c[0] - a[0] * c[2];
    s[2] += a[0] * c[1] - a[1] * c[0];
  }
  float n[3], inv;
  normalize3(s, n, &inv);
  vnrm[o] = make_float4(n[0], n[1], n[2], inv);
  const float* cb = vcolors + (size_t)b * vcol_bstride + 3 * v;
  vcol[o] = make_float4(cb[0], cb[1], cb[2], 0.f);
}

// ------------------------------------------------------------------------------------------------
// per-image face binning: tile_cnt[b][ty][tx] faces in tile_list[b][ty][tx][0 .. F)
// ------------------------------------------------------------------------------------------------
// Also packs the face's 12 vertex records (NDC, position, unit normal, colour of its three corners: kFaceRec float4 = 192 bytes) into
// frec[b][f]: the tile kernels then fetch a winning face with ONE level of indirection and twelve independent 16-byte loads instead
// of face -> three vertex indices -> twelve gathers (two dependent global latencies per distinct face of a pixel, ~2 us each).
// Round 3, second form: a workgroup is 256 faces x 4 lanes.  The four lanes of a face pack three of its twelve vertex records each (192
// contiguous bytes per face); lane 0 bins: the face's tile range by two binary searches per axis over the SAME predicates the linear scan
// evaluated (28 tiles per axis at 224^2: 10 evaluations instead of 56), the appends counted in LDS first -- one global atomic per
// (workgroup, touched tile) reserves the workgroup's run of list slots, where every (face, tile) pair used to pay a returning atomic on
// the ~60 counters the hand covers (3 000 per image on ~10 cache lines).  44 -> see profiles/r03_time_render.txt.
constexpr int kBinFaces = 256;
template <int AA, int TILE>
__global__ __launch_bounds__(4 * kBinFaces) void render_bin_kernel(RenderDev r, const float4* __restrict__ vndc, const float4* __restrict__ vpos,
                                                                  const float4* __restrict__ vnrm, const float4* __restrict__ vcol,
                                                                  float4* __restrict__ frec, int* __restrict__ tile_cnt,
                                                                  int* __restrict__ tile_list) {
  HIP_DYNAMIC_SHARED(int, s_bin)                       // [tiles^2] counts of this workgroup, then [tiles^2] the reserved list offsets
  const int b = blockIdx.y;
  const int part = threadIdx.x & 3;
  const int f = blockIdx.x * kBinFaces + (threadIdx.x >> 2);
  const int H = r.H, S = H * AA, tiles = (H + TILE - 1) / TILE, nt = tiles * tiles;
  int* const s_cnt = s_bin;
  int* const s_base = s_bin + nt;
  for (int t = threadIdx.x; t < nt; t += 4 * kBinFaces) s_cnt[t] = 0;
  const bool live = f < r.F;
  int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
  if (live) {
    const float4* vb = vndc + (size_t)b * r.V;
    const int i0 = r.faces[3 * f], i1 = r.faces[3 * f + 1], i2 = r.faces[3 * f + 2];
    if (frec != nullptr) {
      const size_t vo = (size_t)b * r.V;
      const float4* src = part == 0 ? vndc : (part == 1 ? vpos : (part == 2 ? vnrm : vcol));
      float4* q = frec + ((size_t)b * r.F + f) * kFaceRec + 3 * part;
      q[0] = src[vo + i0]; q[1] = src[vo + i1]; q[2] = src[vo + i2];
    }
    if (part == 0) {
      const float4 a = vb[i0], c = vb[i1], d = vb[i2];
      FaceXYZ fc;
      fc.x0 = a.x; fc.y0 = a.y; fc.z0 = a.z; fc.x1 = c.x; fc.y1 = c.y; fc.z1 = c.z; fc.x2 = d.x; fc.y2 = d.y; fc.z2 = d.z;
      if (!face_is_rejected(fc)) {
        const float xmin = fminf(fc.x0, fminf(fc.x1, fc.x2)), xmax = fmaxf(fc.x0, fmaxf(fc.x1, fc.x2));
        const float ymin = fminf(fc.y0, fminf(fc.y1, fc.y2)), ymax = fmaxf(fc.y0, fmaxf(fc.y1, fc.y2));
        // tile t spans samples [t * TILE * AA, t * TILE * AA + n * AA - 1]; its NDC bounds are the ones render_fwd2_kernel tests against
        // (index 0 holds the largest coordinate, so both bounds fall with t).  A tile is listed when !(vmin > hi_t) && !(vmax < lo_t): the
        // first holds for t <= T1, the second for t >= T0 (negated comparisons: a NaN coordinate lists the face in every tile, as before).
        const auto hi_of = [&](int t) { return pix_to_ndc(S - 1 - min(t * TILE * AA, S - 1), S); };
        const auto lo_of = [&](int t) { const int o = t * TILE, n = min(TILE, H - o); return pix_to_ndc(S - 1 - min(o * AA + n * AA - 1, S - 1), S); };
        const auto last_le_hi = [&](float vmin) {        // largest t with !(vmin > hi_t), -1 if none
          int lo = -1, hi = tiles - 1;
          while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (!(vmin > hi_of(m))) lo = m; else hi = m - 1; }
          return lo;
        };
        const auto first_ge_lo = [&](float vmax) {       // smallest t with !(vmax < lo_t), tiles if none
          int lo = 0, hi = tiles;
          while (lo < hi) { const int m = (lo + hi) >> 1; if (!(vmax < lo_of(m))) hi = m; else lo = m + 1; }
          return lo;
        };
        tx0 = first_ge_lo(xmax); tx1 = last_le_hi(xmin);
        ty0 = first_ge_lo(ymax); ty1 = last_le_hi(ymin);
      }
    }
  }
  __syncthreads();
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&s_cnt[ty * tiles + tx], 1);
  __syncthreads();
  for (int t = threadIdx.x; t < nt; t += 4 * kBinFaces) {
    const int c = s_cnt[t];
    if (c > 0) s_base[t] = atomicAdd(tile_cnt + (size_t)b * nt + t, c);
    s_cnt[t] = 0;
  }
  __syncthreads();
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) {
      const int t = ty * tiles + tx;
      const int slot = s_base[t] + atomicAdd(&s_cnt[t], 1);
      tile_list[((size_t)b * nt + t) * r.F + slot] = f;
    }
}

// ------------------------------------------------------------------------------------------------
// forward, second form (round 3): 256-thread workgroups, work compacted at every stage.
//   * candidates that survive stage A go to ONE queue per workgroup (wave-aggregated append); stage B -- the exact sample tests -- then
//     runs on dense waves of the queue (the first form kept a queue per wave: 8 partially filled stage-B passes per tile);
//   * shading runs on (busy pixel, sample row) items compacted over the tile: a tile of the hand's outline has ~25 % busy pixels, and
//     the first form shaded all 256 lanes x 9 samples with the other 4 waves of its 512-thread workgroup idle (27 of the launch's
//     36 M vector instructions); row sums go through LDS and are added per pixel in a fixed order (deterministic);
//   * shading arithmetic with fused multiply-adds and approximate reciprocals / square roots (pixels are compared at 1e-4; coverage
//     and depth -- stage B -- keep the oracle's exact sequence); face data from the packed records (render_bin_kernel).
// 36 KB of LDS: four workgroups per CU.
// ------------------------------------------------------------------------------------------------
#if defined(HIFIHR_RENDER_STAMP2)
__device__ unsigned long long g_r2_stamp[16];      // per phase: summed cycles of thread 0 over the busy tiles; [15] = busy tiles
__device__ unsigned g_r2_hist[32];
#define R2_T0 unsigned long long r2_t = __builtin_amdgcn_s_memtime(); const unsigned long long r2_begin = r2_t; (void)r2_begin;
__shared__ unsigned long long s_r2_acc[8];          // thread 0's phase sums of THIS tile: flushed once, at the end (atomics issued mid-kernel
                                                    // stay in vmcnt order in front of the very gathers that are being timed)
#define R2_STAMP(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) s_r2_acc[i] += n_ - r2_t; r2_t = n_; }
#else
#define R2_T0
#define R2_STAMP(i)
#endif
#if defined(HIFIHR_R2_NOSTORE)
#define R2_STORE(...) if (r.H < 0) { __VA_ARGS__; }
#else
#define R2_STORE(...) __VA_ARGS__
#endif
constexpr int kF2Threads = 256;
constexpr int kF2Cap = 256;          // faces per pass
constexpr int kQCap = 1024;          // survivor queue entries: a round appends at most 256


// shade_fwd (render_math.h) with contraction and approximate reciprocal square roots; integer shininess by repeated squaring
__device__ __forceinline__ void shade_fwd_fast(const ShadeConsts& c, const LightDir& L, const float* P, const float* N, const float* T, float* rgb) {
#pragma clang fp contract(fast)
  const float n2 = N[0] * N[0] + N[1] * N[1] + N[2] * N[2];
  const float inv_n = n2 > kNormEps * kNormEps ? fast_rsq(n2) : 1.0f / kNormEps;
  const float nh[3] = {N[0] * inv_n, N[1] * inv_n, N[2] * inv_n};
  float lh[3] = {L.l[0], L.l[1], L.l[2]};
  if (c.point_light) {
    const float d[3] = {L.l[0] - P[0], L.l[1] - P[1], L.l[2] - P[2]};
    const float d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const float inv_l = d2 > kNormEps * kNormEps ? fast_rsq(d2) : 1.0f / kNormEps;
    lh[0] = d[0] * inv_l; lh[1] = d[1] * inv_l; lh[2] = d[2] * inv_l;
  }
  const float cosang = nh[0] * lh[0] + nh[1] * lh[1] + nh[2] * lh[2];
  const float angle = fmaxf(cosang, 0.f);
  const float p2 = P[0] * P[0] + P[1] * P[1] + P[2] * P[2];
  const float inv_v = p2 > kNormEps * kNormEps ? fast_rsq(p2) : 1.0f / kNormEps;
  const float vh[3] = {-P[0] * inv_v, -P[1] * inv_v, -P[2] * inv_v};
  const float r[3] = {-lh[0] + 2.f * (cosang * nh[0]), -lh[1] + 2.f * (cosang * nh[1]), -lh[2] + 2.f * (cosang * nh[2])};
  const float d = vh[0] * r[0] + vh[1] * r[1] + vh[2] * r[2];
  const float alpha = (cosang > 0.f) ? fmaxf(d, 0.f) : 0.f;
  float pw;
  const int si = (int)c.shininess;
  if ((float)si == c.shininess && si >= 0 && si < 1024) {       // (uniform) alpha^si by squaring
    pw = 1.f;
    float base = alpha;
    for (int e = si; e > 0; e >>= 1) {
      if (e & 1) pw *= base;
      base *= base;
    }
  } else {
    pw = powf(alpha, c.shininess);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) rgb[k] = (c.amb[k] + c.mdiff[k] * (L.lc[k] * angle)) * T[k] + c.spec[k] * pw;
}

template <int AA, int TILE>
struct Fwd2Lds {
  union {
    struct {
      float rec[kF2Cap * kRecW];                         // listed faces of this pass
      int coff[kF2Cap + 1];                              // exclusive prefix sum of candidate pixels per listed face
      int q[kQCap];                                      // surviving (face, pixel) candidates of the workgroup
    } r;
    float part[TILE * TILE * AA * 4];                  // shading: (r, g, b, hits) of a (busy pixel, sample row)
  } u;
  unsigned long long zbuf[TILE * AA * TILE * AA];      // per sample: (depth bits << 32) | face id, min-reduced
  float sxs[TILE * AA], sys[TILE * AA];
  int wave_tot[4];
  int qn, nbusy;
  unsigned char busy[TILE * TILE];
};

// exact sample tests of one surviving (face, pixel) candidate: the oracle's arithmetic, 64-bit atomicMin on (depth, face id)
template <int AA, int TILE>
__device__ __forceinline__ void stage_b2(Fwd2Lds<AA, TILE>& L, int entry) {
  constexpr int SW = TILE * AA;
  const int k = entry >> 8, cy = (entry >> 4) & 15, cx = entry & 15;
  const float* q = L.u.r.rec + k * kRecW;
  FaceXYZ f;
  f.x0 = q[0]; f.y0 = q[1]; f.x1 = q[2]; f.y1 = q[3]; f.x2 = q[4]; f.y2 = q[5]; f.z0 = q[6]; f.z1 = q[7]; f.z2 = q[8];
  const unsigned fidu = (unsigned)__float_as_int(q[9]);
  const float xmin = fminf(f.x0, fminf(f.x1, f.x2)), xmax = fmaxf(f.x0, fmaxf(f.x1, f.x2));
  const float ymin = fminf(f.y0, fminf(f.y1, f.y2)), ymax = fmaxf(f.y0, fmaxf(f.y1, f.y2));
#pragma unroll
  for (int i = 0; i < AA; ++i) {
#pragma unroll
    for (int j = 0; j < AA; ++j) {
      float bary[3], pz;
      if (sample_face(f, xmin, xmax, ymin, ymax, L.sxs[cx * AA + j], L.sys[cy * AA + i], bary, &pz)) {
        const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(pz) << 32) | fidu;   // pz >= 0
        atomicMin(&L.zbuf[(cy * AA + i) * SW + cx * AA + j], key);
      }
    }
  }
}

// the n faces listed in LDS: candidate rectangles, scan, stage A over all (face, pixel) candidates, stage B over the survivors
template <int AA, int TILE>
__device__ __forceinline__ void raster_pass2(Fwd2Lds<AA, TILE>& L, int n, int cols, int rows) {
  constexpr int SW = TILE * AA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  R2_T0
  int cnt = 0;
  if (tid < n) {
    const float* q = L.u.r.rec + tid * kRecW;
    const float xmin = fminf(q[0], fminf(q[2], q[4])), xmax = fmaxf(q[0], fmaxf(q[2], q[4]));
    const float ymin = fminf(q[1], fminf(q[3], q[5])), ymax = fmaxf(q[1], fmaxf(q[3], q[5]));
    const auto prefix = [](const float* sv, int off, int nn, float v, bool strict) {          // see raster_candidates
      int len = 0;
#pragma unroll
      for (int step = TILE; step > 0; step >>= 1) {
        const int t = len + step;
        if (t <= nn) {
          const float a = sv[(t - 1) * AA + off];
          if (strict ? (a > v) : (a >= v)) len = t;
        }
      }
      return len;
    };
    const int x0 = prefix(L.sxs, AA - 1, cols, xmax, true), x1 = prefix(L.sxs, 0, cols, xmin, false) - 1;
    const int y0 = prefix(L.sys, AA - 1, rows, ymax, true), y1 = prefix(L.sys, 0, rows, ymin, false) - 1;
    const int w = x1 - x0 + 1, h = y1 - y0 + 1;
    if (w > 0 && h > 0) {
      cnt = w * h;
      L.u.r.rec[tid * kRecW + 10] = __int_as_float(x0 | (y0 << 4) | (w << 8));
    }
  }
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) L.wave_tot[wave] = incl;
  if (tid == 0) L.qn = 0;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += L.wave_tot[w];
  if (tid < n) L.u.r.coff[tid] = base + incl - cnt;
  const int total = L.wave_tot[0] + L.wave_tot[1] + L.wave_tot[2] + L.wave_tot[3];
  if (tid == 0) L.u.r.coff[n] = total;
  __syncthreads();
  R2_STAMP(5)
#if defined(HIFIHR_RENDER_STAMP2)
  (void)0;
#endif
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int cb = 0; cb < total; cb += kF2Threads) {
    if (cb > 0) {
      __syncthreads();                                       // the queue count of the previous round is complete
      // (uniform) no room for another round: drain.  (Draining after EVERY round in tiles with > 1024 candidates, so that stage A's depth
      // reject sees the front layers early, measured slower: 163 vs 121 us -- two more barriers and a partially filled stage B per round.)
      if (L.qn > kQCap - kF2Threads) {
        const int qn = L.qn;
        for (int e = tid; e < qn; e += kF2Threads) stage_b2<AA, TILE>(L, L.u.r.q[e]);
        __syncthreads();
        if (tid == 0) L.qn = 0;
        __syncthreads();
      }
    }
    const int c = cb + tid;
    int lo = 0, hi = n - 1;
#pragma unroll
    for (int it = 0; it < 8; ++it) {                         // largest k with coff[k] <= c (n <= 256; idempotent once lo == hi)
      const int mid = (lo + hi + 1) >> 1;
      const bool le = L.u.r.coff[mid] <= min(c, total - 1);
      lo = le ? mid : lo;
      hi = le ? hi : mid - 1;
    }
    bool survive = false;
    int packed = 0;
    if (c < total) {
      const int k = lo;
      const float* q = L.u.r.rec + k * kRecW;
      const int info = __float_as_int(q[10]);
      const int x0 = info & 15, y0 = (info >> 4) & 15, w = info >> 8;
      const int local = c - L.u.r.coff[k];
      const int dy = local / w, dx = local - dy * w;
      const int cx = x0 + dx, cy = y0 + dy;
      FaceXYZ f;
      f.x0 = q[0]; f.y0 = q[1]; f.x1 = q[2]; f.y1 = q[3]; f.x2 = q[4]; f.y2 = q[5]; f.z0 = q[6]; f.z1 = q[7]; f.z2 = q[8];
      // both conservative rejects argue from a face wholly in front of the camera (inside <=> inside the 2-D triangle; depth = a convex
      // combination of the vertex depths); a face with a vertex at / behind the camera plane skips them (render_math.h sample_face)
      const float zmin_f = fminf(f.z0, fminf(f.z1, f.z2));
      survive = zmin_f <= 0.f || !square_misses_face(f, L.sxs[cx * AA + AA - 1], L.sxs[cx * AA], L.sys[cy * AA + AA - 1], L.sys[cy * AA]);
      if (survive && zmin_f > 0.f) {
        const float znear = zmin_f * (1.0f - 1e-5f);
        bool behind = true;
#pragma unroll
        for (int i = 0; i < AA; ++i)
#pragma unroll
          for (int j = 0; j < AA; ++j) {
            const unsigned long long key = L.zbuf[(cy * AA + i) * SW + cx * AA + j];
            behind = behind && (key != ~0ull) && (__int_as_float((int)(unsigned)(key >> 32)) < znear);
          }
        survive = !behind;
      }
      packed = (k << 8) | (cy << 4) | cx;
    }
    const unsigned long long m = __ballot(survive);
    int qb = 0;
    if (lane == 0 && m != 0ull) qb = atomicAdd(&L.qn, __popcll(m));
    qb = __shfl(qb, 0, 64);
    if (survive) L.u.r.q[qb + __popcll(m & lt)] = packed;
  }
  __syncthreads();
  R2_STAMP(6)
  {
    const int qn = L.qn;
#if defined(HIFIHR_RENDER_STAMP2)
    (void)0;
#endif
    for (int e = tid; e < qn; e += kF2Threads) stage_b2<AA, TILE>(L, L.u.r.q[e]);
  }
  __syncthreads();
  R2_STAMP(7)
}

template <int AA, int TILE, bool UV>
__global__ __launch_bounds__(kF2Threads) void render_fwd2_kernel(RenderDev r, const float4* __restrict__ frec,
                                                                 const float* __restrict__ light_color, const float* __restrict__ light_dir,
                                                                 float* __restrict__ rgba, int* __restrict__ face_id,
                                                                 const int* __restrict__ tile_cnt, const int* __restrict__ tile_list,
                                                                 TexUvDev tuv, int nB, int xcd_map) {
  HIP_DYNAMIC_SHARED(float4, smem_raw)
  Fwd2Lds<AA, TILE>& L = *reinterpret_cast<Fwd2Lds<AA, TILE>*>(smem_raw);
  constexpr int SW = TILE * AA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = r.H, S = H * AA;
  const int tiles = (H + TILE - 1) / TILE;
  // 1-D grid, XCD-aware: workgroups are dealt round-robin over the 8 XCDs (id % 8), so image b = 8 g + (id % 8): all tiles of an image
  // run on ONE XCD and its face records / tile lists (0.3 MB + 0.1 MB per image) stay in that XCD's 4 MB L2 -- with blockIdx.z = image
  // every XCD touched every image (9.4 MB of records at B = 32) and the gathers were served from the Infinity Cache under load
  // (staging 12.5 us, shading 21 us per tile: tools/render_stamp2.py)
  int b, tix, tiy;
  if (xcd_map) {
    const int bid = blockIdx.x, xcd = bid & 7, jj = bid >> 3;
    b = (jj / (tiles * tiles)) * 8 + xcd;
    if (b >= nB) return;
    tix = (jj % (tiles * tiles)) % tiles; tiy = (jj % (tiles * tiles)) / tiles;
  } else {
    const int bid = blockIdx.x;
    b = bid / (tiles * tiles);
    tix = (bid % (tiles * tiles)) % tiles; tiy = (bid % (tiles * tiles)) / tiles;
  }
  const int ox = tix * TILE, oy = tiy * TILE;
  const int cols = min(TILE, H - ox), rows = min(TILE, H - oy);
  const size_t tile = ((size_t)b * tiles + tiy) * tiles + tix;
  const int nlist = tile_cnt[tile];
  const int* flist = tile_list + tile * r.F;
  const size_t plane = (size_t)H * H;
  const float inv = (float)(AA * AA);
  {
    // each wave owns a compact 8x8-pixel quadrant of the tile (8 x 8 tiles: wave 0 alone)
    const int tx = (lane & 7) + (TILE > 8 ? 8 * (wave & 1) : 0), ty = (lane >> 3) + (TILE > 8 ? 8 * (wave >> 1) : 0);
    const int px = ox + tx, py = oy + ty;
    const bool live = (tid < TILE * TILE) && (px < H) && (py < H);
    if (nlist == 0) {
      // background (four of five tiles): one pixel per lane.  (All 256 lanes storing 16-byte pieces of the sample rows measured slower:
      // 45.6 vs 34 us for an all-background launch.)
      if (!live) return;
      float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < AA; ++i)
#pragma unroll
        for (int j = 0; j < AA; ++j) {
          R2_STORE(face_id[((size_t)b * S + (py * AA + i)) * S + (px * AA + j)] = -1);
          acc[0] += r.bg[0]; acc[1] += r.bg[1]; acc[2] += r.bg[2];
        }
      float* o = rgba + (size_t)b * 4 * plane + (size_t)py * H + px;
      R2_STORE(o[0] = acc[0] / inv; o[plane] = acc[1] / inv; o[2 * plane] = acc[2] / inv; o[3 * plane] = 0.f / inv);
      return;
    }
  }
#if defined(HIFIHR_RENDER_STAMP2)
  if (tid < 8) s_r2_acc[tid] = 0ull;
#endif
  R2_T0
  for (int e = tid; e < 2 * SW; e += kF2Threads) {
    const int idx = e < SW ? e : e - SW;
    const int g = min((e < SW ? ox : oy) * AA + idx, S - 1);
    const float v = pix_to_ndc(S - 1 - g, S);
    if (e < SW) L.sxs[idx] = v; else L.sys[idx] = v;
  }
  for (int e = tid; e < SW * SW; e += kF2Threads) L.zbuf[e] = ~0ull;
  const float4* fr = frec + (size_t)b * r.F * kFaceRec;
  for (int base = 0; base < nlist; base += kF2Cap) {
    const int n = min(kF2Cap, nlist - base);
    __syncthreads();                                         // (previous pass done with rec; first pass: sxs / zbuf written)
    R2_STAMP(0)
    if (tid < n) {
      const int f = flist[base + tid];
      const float4 a = fr[(size_t)f * kFaceRec], c = fr[(size_t)f * kFaceRec + 1], d = fr[(size_t)f * kFaceRec + 2];
      float* q = L.u.r.rec + tid * kRecW;
      q[0] = a.x; q[1] = a.y; q[2] = c.x; q[3] = c.y; q[4] = d.x; q[5] = d.y; q[6] = a.z; q[7] = c.z;
      q[8] = d.z; q[9] = __int_as_float(f); q[10] = 0.f;
    }
    __syncthreads();
    R2_STAMP(1)
    raster_pass2<AA, TILE>(L, n, cols, rows);
    R2_STAMP(2)
  }
  // ---- resolve: face ids out, busy pixels compacted ----
  const int tx = (lane & 7) + (TILE > 8 ? 8 * (wave & 1) : 0), ty = (lane >> 3) + (TILE > 8 ? 8 * (wave >> 1) : 0);
  const int px = ox + tx, py = oy + ty;
  const bool live = (tid < TILE * TILE) && (px < H) && (py < H);
  bool busy = false;
  if (live) {
#pragma unroll
    for (int i = 0; i < AA; ++i)
#pragma unroll
      for (int j = 0; j < AA; ++j) {
        const unsigned long long key = L.zbuf[(ty * AA + i) * SW + tx * AA + j];
        const int f = (key == ~0ull) ? -1 : (int)(unsigned)(key & 0xffffffffull);
        R2_STORE(face_id[((size_t)b * S + (py * AA + i)) * S + (px * AA + j)] = f);
        busy = busy || (f >= 0);
      }
    if (!busy) {
      float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < AA * AA; ++s2) { acc[0] += r.bg[0]; acc[1] += r.bg[1]; acc[2] += r.bg[2]; }
      float* o = rgba + (size_t)b * 4 * plane + (size_t)py * H + px;
      R2_STORE(o[0] = acc[0] / inv; o[plane] = acc[1] / inv; o[2 * plane] = acc[2] / inv; o[3 * plane] = 0.f / inv);
    }
  }
  const unsigned long long bm = __ballot(busy);
  if (lane == 0) L.wave_tot[wave] = __popcll(bm);
  __syncthreads();                                           // (also: every wave is done with rec / q before `part` overlays them)
  int bbase = 0;
  for (int w = 0; w < wave; ++w) bbase += L.wave_tot[w];
  if (busy) L.busy[bbase + __popcll(bm & ((1ull << lane) - 1ull))] = (unsigned char)((ty << 4) | tx);
  const int nbusy = L.wave_tot[0] + L.wave_tot[1] + L.wave_tot[2] + L.wave_tot[3];
  __syncthreads();
  R2_STAMP(3)
  if (nbusy == 0) return;
  // ---- shade (busy pixel, sample row) items ----
  LightDir Ld;
  {
    const float raw[3] = {light_dir[3 * b], light_dir[3 * b + 1], light_dir[3 * b + 2]};
    normalize3(raw, Ld.l, &Ld.inv_norm);
    if (r.sc.point_light) { Ld.l[0] = raw[0]; Ld.l[1] = raw[1]; Ld.l[2] = raw[2]; }
    Ld.lc[0] = light_color[3 * b]; Ld.lc[1] = light_color[3 * b + 1]; Ld.lc[2] = light_color[3 * b + 2];
  }
  for (int e = tid; e < nbusy * AA; e += kF2Threads) {
    const int pi = e / AA, i = e - pi * AA;
    const int code = L.busy[pi], bx = code & 15, by = code >> 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int cur_f = -1;
    FaceXYZ fc;
    fc.x0 = fc.y0 = fc.z0 = fc.x1 = fc.y1 = fc.z1 = fc.x2 = fc.y2 = fc.z2 = 0.f;
    float4 p0, p1, p2, n0, n1, n2, c0, c1, c2;
    p0 = p1 = p2 = n0 = n1 = n2 = c0 = c1 = c2 = make_float4(0.f, 0.f, 0.f, 0.f);
    float inv_area = 0.f;
    float fu[3] = {0.f, 0.f, 0.f}, fv[3] = {0.f, 0.f, 0.f};
    const float syi = L.sys[by * AA + i];
#pragma unroll
    for (int j = 0; j < AA; ++j) {
      const unsigned long long key = L.zbuf[(by * AA + i) * SW + bx * AA + j];
      if (key == ~0ull) {
        acc[0] += r.bg[0]; acc[1] += r.bg[1]; acc[2] += r.bg[2];
        continue;
      }
      const int f = (int)(unsigned)(key & 0xffffffffull);
      if (f != cur_f) {
        cur_f = f;
        const float4* q = fr + (size_t)f * kFaceRec;
        const float4 a = q[0], c = q[1], d = q[2];
        p0 = q[3]; p1 = q[4]; p2 = q[5]; n0 = q[6]; n1 = q[7]; n2 = q[8]; c0 = q[9]; c1 = q[10]; c2 = q[11];
        fc.x0 = a.x; fc.y0 = a.y; fc.z0 = a.z; fc.x1 = c.x; fc.y1 = c.y; fc.z1 = c.z; fc.x2 = d.x; fc.y2 = d.y; fc.z2 = d.z;
        inv_area = fast_rcp(edge_fn(fc.x2, fc.y2, fc.x0, fc.y0, fc.x1, fc.y1) + kRasterEps);
        if constexpr (UV) {
#pragma unroll
          for (int k = 0; k < 3; ++k) { const int iu = tuv.faces_uvs[3 * f + k]; fu[k] = tuv.verts_uvs[2 * iu]; fv[k] = tuv.verts_uvs[2 * iu + 1]; }
        }
      }
      const float sxj = L.sxs[bx * AA + j];
      float bary[3];
      {
        const float w0 = edge_fn(sxj, syi, fc.x1, fc.y1, fc.x2, fc.y2) * inv_area;
        const float w1 = edge_fn(sxj, syi, fc.x2, fc.y2, fc.x0, fc.y0) * inv_area;
        const float w2 = edge_fn(sxj, syi, fc.x0, fc.y0, fc.x1, fc.y1) * inv_area;
        const float t0 = w0 * fc.z1 * fc.z2, t1 = fc.z0 * w1 * fc.z2, t2 = fc.z0 * fc.z1 * w2;
        const float id = fast_rcp(fmaxf(t0 + t1 + t2, kRasterEps));
        bary[0] = t0 * id; bary[1] = t1 * id; bary[2] = t2 * id;
      }
      const float P[3] = {bary[0] * p0.x + bary[1] * p1.x + bary[2] * p2.x, bary[0] * p0.y + bary[1] * p1.y + bary[2] * p2.y,
                          bary[0] * p0.z + bary[1] * p1.z + bary[2] * p2.z};
      const float N[3] = {bary[0] * n0.x + bary[1] * n1.x + bary[2] * n2.x, bary[0] * n0.y + bary[1] * n1.y + bary[2] * n2.y,
                          bary[0] * n0.z + bary[1] * n1.z + bary[2] * n2.z};
      float T[3] = {bary[0] * c0.x + bary[1] * c1.x + bary[2] * c2.x, bary[0] * c0.y + bary[1] * c1.y + bary[2] * c2.y,
                    bary[0] * c0.z + bary[1] * c1.z + bary[2] * c2.z};
      if constexpr (UV) {
        const float u = bary[0] * fu[0] + bary[1] * fu[1] + bary[2] * fu[2], v = bary[0] * fv[0] + bary[1] * fv[1] + bary[2] * fv[2];
        uv_fetch(tuv, b, uv_sample(u, v, tuv.TH, tuv.TW), T, nullptr, nullptr);
      }
      float rgb[3];
      shade_fwd_fast(r.sc, Ld, P, N, T, rgb);
      acc[0] += rgb[0]; acc[1] += rgb[1]; acc[2] += rgb[2]; acc[3] += 1.f;
    }
    float* pp = L.u.part + (size_t)e * 4;
    pp[0] = acc[0]; pp[1] = acc[1]; pp[2] = acc[2]; pp[3] = acc[3];
  }
  __syncthreads();
  R2_STAMP(4)
#if defined(HIFIHR_RENDER_STAMP2)
  if (tid < 8) atomicAdd(&g_r2_stamp[tid], s_r2_acc[tid]);
  if (tid == 0) {
    const unsigned long long dur = __builtin_amdgcn_s_memtime() - r2_begin;
    atomicMax(&g_r2_stamp[10], dur);
    atomicAdd(&g_r2_hist[min(31, (int)(dur >> 14))], 1u);          // buckets of 16384 cycles (~7.8 us)
  }
  if (tid == 0) { atomicAdd(&g_r2_stamp[15], 1ull); atomicAdd(&g_r2_stamp[14], (unsigned long long)nbusy); atomicAdd(&g_r2_stamp[13], (unsigned long long)nlist); }
#endif
  for (int pi = tid; pi < nbusy; pi += kF2Threads) {
    const int code = L.busy[pi], bx = code & 15, by = code >> 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < AA; ++i) {
      const float* pp = L.u.part + (size_t)(pi * AA + i) * 4;
      acc[0] += pp[0]; acc[1] += pp[1]; acc[2] += pp[2]; acc[3] += pp[3];
    }
    float* o = rgba + (size_t)b * 4 * plane + (size_t)(oy + by) * H + (ox + bx);
    o[0] = acc[0] / inv; o[plane] = acc[1] / inv; o[2 * plane] = acc[2] / inv; o[3 * plane] = acc[3] / inv;
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// workspace: four float4[B][V] vertex arrays, the float[B][V][12] gradient records of the backward, then the forward's per-tile face
// lists: int cnt[B][tiles^2] and int list[B][tiles^2][F] (worst case: the whole mesh inside one tile)
size_t render_workspace_bytes(const RenderDev& r, int B) {
  return vertex_part_bytes(r, B) + list_part_bytes(r, B) + (size_t)B * r.F * kFaceRec * sizeof(float4);
}
#if defined(HIFIHR_RENDER_STAMP2)
extern "C" int hifihr_debug_render_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_r2_stamp), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_r2_stamp), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
extern "C" int hifihr_debug_render_hist(unsigned* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_r2_hist), sizeof(unsigned) * 32) != hipSuccess) return -1;
  if (reset) { unsigned z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_r2_hist), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

hipError_t launch_render_fwd(const RenderDev& r, const float* verts, const float* vcolors, long vcol_bstride, const float* cam,
                             const float* light_color, const float* light_dir, int B, float* rgba, int* face_id, void* ws,
                             hipStream_t st, const TexUvPass* uv) {
  float4 *vndc, *vpos, *vnrm, *vcol;
  float* gvrec;
  int *tile_cnt, *tile_list;
  const int te = render_tile();
  carve(r, B, ws, &vndc, &vpos, &vnrm, &vcol, &gvrec, &tile_cnt, &tile_list, te);
  const int tiles = (r.H + te - 1) / te;
  hipLaunchKernelGGL(render_vertex_kernel, dim3((r.V + 255) / 256, B), dim3(256), 0, st, r, verts, vcolors, vcol_bstride, cam,
                     vndc, vpos, vnrm, vcol, tile_cnt, tiles * tiles);
  const dim3 bgrid((r.F + kBinFaces - 1) / kBinFaces, B);
  static const int xm = [] { const char* e = getenv("HIFIHR_RENDER_XCD"); return e ? atoi(e) : 0; }();      // A/B: images pinned to XCDs
  const dim3 grid1((unsigned)((xm ? 8 * ((B + 7) / 8) : B) * tiles * tiles));
  float4* frec = face_records(r, B, ws);
  const TexUvDev td = uv != nullptr ? TexUvDev{uv->faces_uvs, uv->verts_uvs, uv->maps, nullptr, uv->TH, uv->TW} : TexUvDev{};
#define HIFIHR_RENDER_FWD2(AA_, T_)                                                                                                      \
  {                                                                                                                                     \
    hipLaunchKernelGGL((render_bin_kernel<AA_, T_>), bgrid, dim3(4 * kBinFaces), (size_t)2 * tiles * tiles * sizeof(int), st, r, vndc,     \
                       vpos, vnrm, vcol, frec, tile_cnt, tile_list);                                                                    \
    if (uv != nullptr)                                                                                                                  \
      hipLaunchKernelGGL((render_fwd2_kernel<AA_, T_, true>), grid1, dim3(kF2Threads), sizeof(Fwd2Lds<AA_, T_>), st, r, frec, light_color, \
                         light_dir, rgba, face_id, tile_cnt, tile_list, td, B, xm);                                                     \
    else                                                                                                                                \
      hipLaunchKernelGGL((render_fwd2_kernel<AA_, T_, false>), grid1, dim3(kF2Threads), sizeof(Fwd2Lds<AA_, T_>), st, r, frec,           \
                         light_color, light_dir, rgba, face_id, tile_cnt, tile_list, td, B, xm);                                        \
  }
#define HIFIHR_RENDER_FWD(AA_) if (te == 8) HIFIHR_RENDER_FWD2(AA_, 8) else HIFIHR_RENDER_FWD2(AA_, 16)
  switch (r.aa) {
    case 1: HIFIHR_RENDER_FWD(1) break;
    case 2: HIFIHR_RENDER_FWD(2) break;
    case 3: HIFIHR_RENDER_FWD(3) break;
    default: return hipErrorInvalidValue;
  }
#undef HIFIHR_RENDER_FWD
#undef HIFIHR_RENDER_FWD2
  return hipGetLastError();
}

}  // namespace hifihr
