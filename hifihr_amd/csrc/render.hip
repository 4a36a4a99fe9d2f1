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
#include <mutex>

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
                                                           float4* __restrict__ vcol, int* __restrict__ tile_cnt, int ntiles,
                                                           int* __restrict__ qctl, float* __restrict__ gvrec, float* __restrict__ glrec) {
  const int b = blockIdx.y;
  const int v = blockIdx.x * 256 + threadIdx.x;
  // the backward's accumulators start at zero (render_common.h: render_ws_mark_clean)
  if (blockIdx.x == 0 && threadIdx.x < 3) { glrec[3 * b + threadIdx.x] = 0.f; glrec[3 * (gridDim.y + b) + threadIdx.x] = 0.f; }
  if (v < r.V) {
    float4* g4 = reinterpret_cast<float4*>(gvrec + ((size_t)b * r.V + v) * 12);
    g4[0] = g4[1] = g4[2] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (qctl != nullptr && blockIdx.x == 0) {                  // render_fwd3_kernel's class counts; this image's arrival counter
    if (b == 0 && threadIdx.x < 64) qctl[threadIdx.x] = 0;
    if (threadIdx.x == 64) qctl[64 + b] = 0;
  }
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
// work queue of the forward's third form (render_fwd3_kernel, below)
// ------------------------------------------------------------------------------------------------
constexpr int kF3Cap = 128;           // faces per pass and the target length of a part
constexpr int kF3Parts = 8;           // a tile is cut into at most this many parts
constexpr int kF3SplitSlots = 128;    // split tiles per image that get a merge buffer (further long tiles are walked by one workgroup)
constexpr int kF3Classes = 4;
constexpr int kF3CtlInts = 64;        // [2..5] items per class; then [kF3CtlInts + b] = bin workgroups of image b that have finished (all zeroed by render_vertex_kernel)
struct F3Ws {
  int part_faces;                    // target faces per part of a split tile (kF3Cap .. : see f3_part_faces)
  int* ctl;                          // [kF3CtlInts]
  int2* queue;                       // [kF3Classes][cap]: (b << 18 | tile << 6 | part << 3 | P - 1, split slot or -1)
  int* arrive;                       // [B][kF3SplitSlots]
  unsigned long long* gz;            // [B][kF3SplitSlots][SW * SW]
  int cap;
};
static inline size_t f3_queue_cap(const RenderDev& r, int B) {
  const size_t nt = (size_t)((r.H + 7) / 8) * ((r.H + 7) / 8);
  return (size_t)B * (nt + (size_t)(kF3Parts - 1) * kF3SplitSlots);
}
static inline size_t f3_part_bytes(const RenderDev& r, int B) {
  const size_t sw2 = (size_t)(8 * r.aa) * (8 * r.aa);
  return (size_t)(kF3CtlInts + B) * 4 + (size_t)kF3Classes * f3_queue_cap(r, B) * sizeof(int2) + (size_t)B * kF3SplitSlots * 4 +
         (size_t)B * kF3SplitSlots * sw2 * 8 + 256;
}
static inline bool f3_supported(const RenderDev& r, int B) {
  return r.H <= 512 && B < (1 << 14) && render_tile() == 8;         // 64 x 64 tiles and the batch index fit the item code
}
static inline F3Ws f3_carve(const RenderDev& r, int B, void* ws) {
  char* p = reinterpret_cast<char*>(face_records(r, B, ws)) + (size_t)B * r.F * kFaceRec * sizeof(float4);
  p = reinterpret_cast<char*>(((uintptr_t)p + 255) / 256 * 256);
  F3Ws w;
  // Faces per part: a part starts with an EMPTY depth buffer, so the conservative depth reject of stage A sees nothing of what the other
  // parts hold -- cutting a list finer than it must be multiplies the exact sample tests.  On the MANO mesh (41 faces per covered tile on
  // average, a few tiles of 300-600) parts of 128 shorten the launch's tail; on a dense skin (11 976 faces: hundreds per tile everywhere)
  // they made the launch 18 % longer than one workgroup per tile (838 vs 710 us at B = 48): the target grows with the mesh.
  static const int forced = [] { const char* e = getenv("HIFIHR_RENDER_PART"); return e ? atoi(e) : 0; }();
  w.part_faces = forced > 0 ? forced : (r.F <= 2048 ? kF3Cap : (r.F <= 8192 ? 2 * kF3Cap : 4 * kF3Cap));
  w.cap = (int)f3_queue_cap(r, B);
  const size_t sw2 = (size_t)(8 * r.aa) * (8 * r.aa);
  w.gz = reinterpret_cast<unsigned long long*>(p); p += (size_t)B * kF3SplitSlots * sw2 * 8;
  w.queue = reinterpret_cast<int2*>(p); p += (size_t)kF3Classes * w.cap * sizeof(int2);
  w.arrive = reinterpret_cast<int*>(p); p += (size_t)B * kF3SplitSlots * 4;
  w.ctl = reinterpret_cast<int*>(p);
  return w;
}

__device__ __forceinline__ int f3_class(int faces_per_part) {
  return faces_per_part >= 96 ? 0 : (faces_per_part >= 48 ? 1 : (faces_per_part >= 24 ? 2 : 3));
}

// The work items of image b (one workgroup of THREADS threads: the LAST workgroup of render_bin_kernel to finish the image -- a separate
// launch of one workgroup per image cost 5.5 us): items of its covered tiles into the class queues, merge buffers of its split tiles
// initialised.  The counts were written by other workgroups' device-scope atomics: read with an atomic too.
template <int THREADS>
__device__ __forceinline__ void f3_sched_image(const RenderDev& r, int* __restrict__ tile_cnt, const F3Ws& w, int sw2, int b) {
  __shared__ int s_cnt[kF3Classes], s_base[kF3Classes], s_nslot;
  const int tid = threadIdx.x;
  const int tiles = (r.H + 7) / 8, nt = tiles * tiles;
  if (tid < kF3Classes) s_cnt[tid] = 0;
  if (tid == 0) s_nslot = 0;
  __syncthreads();
  // pass 1: parts and slots per tile (kept in registers: <= ceil(4096 / THREADS) tiles per thread), class counts
  constexpr int kPer = (4096 + THREADS - 1) / THREADS;
  int code[kPer], slot_of[kPer], local[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int t = tid + THREADS * k;
    code[k] = -1; slot_of[k] = -1; local[k] = 0;
    if (t < nt) {
      const int n = atomicAdd(&tile_cnt[(size_t)b * nt + t], 0);
      if (n > 0) {
        int P = min(kF3Parts, (n + w.part_faces - 1) / w.part_faces);
        if (P > 1) {
          const int sl = atomicAdd(&s_nslot, 1);
          if (sl < kF3SplitSlots) slot_of[k] = sl; else P = 1;
        }
        const int cls = f3_class((n + P - 1) / P);
        local[k] = atomicAdd(&s_cnt[cls], P);
        code[k] = (cls << 4) | (P - 1);
      }
    }
  }
  __syncthreads();
  if (tid < kF3Classes) s_base[tid] = s_cnt[tid] > 0 ? atomicAdd(&w.ctl[2 + tid], s_cnt[tid]) : 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    if (code[k] >= 0) {
      const int t = tid + THREADS * k, cls = code[k] >> 4, P = (code[k] & 15) + 1;
      const int slot = slot_of[k] >= 0 ? b * kF3SplitSlots + slot_of[k] : -1;
      int2* q = w.queue + (size_t)cls * w.cap + s_base[cls] + local[k];
      for (int p = 0; p < P; ++p) q[p] = int2{(b << 18) | (t << 6) | (p << 3) | (P - 1), slot};
    }
  }
  const int nslot = min(s_nslot, kF3SplitSlots);
  for (int e = tid; e < nslot; e += THREADS) w.arrive[b * kF3SplitSlots + e] = 0;
  unsigned long long* gz = w.gz + (size_t)b * kF3SplitSlots * sw2;
  for (int e = tid; e < nslot * sw2; e += THREADS) gz[e] = ~0ull;
}

#if defined(HIFIHR_HOSTSIM)
#define HIFIHR_R_WAIT_VMEM() ((void)0)
// emulator builds only: how many split-tile merges / resolves by the last arriver / background strips ran (tests assert that the
// small test images reach every path of render_fwd3_kernel)
static int g_f3_dbg[4];
extern "C" void hifihr_hostsim_render_fwd3_counts(int* out4, int reset) {
  for (int i = 0; i < 4; ++i) { out4[i] = g_f3_dbg[i]; if (reset) g_f3_dbg[i] = 0; }
}
#define F3_DBG(i) if (threadIdx.x == 0) __atomic_fetch_add(&g_f3_dbg[i], 1, __ATOMIC_RELAXED);
#else
#define F3_DBG(i)
#define HIFIHR_R_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

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
                                                                  int* __restrict__ tile_list, F3Ws w3, int sched) {
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
  if (sched) {
    // the image's last workgroup to get here turns the finished counts into work items (f3_sched_image)
    __shared__ int s_last;
    HIFIHR_R_WAIT_VMEM();                                    // this wave's count atomics are acknowledged ...
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&w3.ctl[kF3CtlInts + b], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (s_last) f3_sched_image<4 * kBinFaces>(r, tile_cnt, w3, (8 * AA) * (8 * AA), b);
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
// render_fwd3_kernel, on the 100 MHz clock every CU shares: [0] earliest workgroup start, [1 + c] latest end of an item of class c,
// [5 + c] latest START of an item of class c, [9] latest end of a workgroup (fill included), [10] items, [11] split parts
__device__ unsigned long long g_r3_time[16];
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

template <int AA, int TILE, int CAP = kF2Cap>
struct Fwd2Lds {
  union {
    struct {
      float rec[CAP * kRecW];                            // listed faces of this pass
      int coff[CAP + 1];                                 // exclusive prefix sum of candidate pixels per listed face
      unsigned short q[kQCap];                           // surviving (face, pixel) candidates of the workgroup: k << 8 | cy << 4 | cx
    } r;
    float part[TILE * TILE * AA * 4];                  // shading: (r, g, b, hits) of a (busy pixel, sample row)
  } u;
  unsigned long long zbuf[TILE * AA * TILE * AA];      // per sample: (depth bits << 32) | face id, min-reduced
  float sxs[TILE * AA], sys[TILE * AA];
  int wave_tot[4];
  int qn, nbusy;
#if defined(HIFIHR_RENDER_STAMP2)
  unsigned long long stamp_begin; int stamp_nlist;
#endif
  int item, last;                                      // render_fwd3_kernel: the work item in hand; "this workgroup merged a split tile last"
  unsigned char busy[TILE * TILE];
};

// exact sample tests of one surviving (face, pixel) candidate: the oracle's arithmetic, 64-bit atomicMin on (depth, face id).
// (Round 4 measured the alternative that compacts the COVERED (survivor, sample) pairs of a wave first -- sign tests only, then the
// divisions on dense lanes of a per-wave LDS list: a quarter fewer vector instructions, and the tile kernel 83 -> 102 us.  The kernel is
// bound by its chain of dependent LDS / shuffle steps per tile, not by instruction issue; the extra scan, list write and read cost more
// than the idle lanes they remove.)
template <int AA, int TILE, int CAP>
__device__ __forceinline__ void stage_b2(Fwd2Lds<AA, TILE, CAP>& L, int entry) {
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
template <int AA, int TILE, int CAP>
__device__ __forceinline__ void raster_pass2(Fwd2Lds<AA, TILE, CAP>& L, int n, int cols, int rows) {
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
  if (tid == 0) atomicAdd(&g_r2_stamp[12], (unsigned long long)total);        // (face, pixel) candidates of this pass
#endif
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int cb = 0; cb < total; cb += kF2Threads) {
    if (cb > 0) {
      __syncthreads();                                       // the queue count of the previous round is complete
      // (uniform) no room for another round: drain.  (Draining after EVERY round in tiles with > 1024 candidates, so that stage A's depth
      // reject sees the front layers early, measured slower: 163 vs 121 us -- two more barriers and a partially filled stage B per round.)
      if (L.qn > kQCap - kF2Threads) {
        const int qn = L.qn;
        for (int e = tid; e < qn; e += kF2Threads) stage_b2<AA, TILE, CAP>(L, (int)L.u.r.q[e]);
        __syncthreads();
        if (tid == 0) L.qn = 0;
        __syncthreads();
      }
    }
    const int c = cb + tid;
    int lo = 0, hi = n - 1;
#pragma unroll
    for (int it = 0; it < 8; ++it) {                         // largest k with coff[k] <= c (n <= CAP <= 256; idempotent once lo == hi)
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
      // (the depth reject assumes pz >= zmin_f (1 - 1e-5): the corrected barycentrics sum to ~1 only while the denominator z_i z_j
      // stays well above sample_face's 1e-8 clamp -- faces nearer than 1e-3 skip it and go through the exact tests)
      if (survive && zmin_f > 1e-3f) {
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
    if (survive) L.u.r.q[qb + __popcll(m & lt)] = (unsigned short)packed;
  }
  __syncthreads();
  R2_STAMP(6)
  {
    const int qn = L.qn;
#if defined(HIFIHR_RENDER_STAMP2)
    if (tid == 0) atomicAdd(&g_r2_stamp[11], (unsigned long long)qn);         // survivors of stage A that reach the final drain
#endif
    for (int e = tid; e < qn; e += kF2Threads) stage_b2<AA, TILE, CAP>(L, (int)L.u.r.q[e]);
  }
  __syncthreads();
  R2_STAMP(7)
}

// resolve + shade of one tile whose samples' nearest (depth, face id) keys stand in L.zbuf: face ids out, busy pixels compacted,
// (busy pixel, sample row) items shaded, pixels summed in a fixed order.  Shared by render_fwd2_kernel (one workgroup per tile) and
// render_fwd3_kernel (persistent workgroups); every path through it is workgroup-uniform.
template <int AA, int TILE, bool UV, int CAP>
__device__ __forceinline__ void resolve_shade2(Fwd2Lds<AA, TILE, CAP>& L, const RenderDev& r, const float4* __restrict__ fr,
                                               const float* __restrict__ light_color, const float* __restrict__ light_dir,
                                               float* __restrict__ rgba, int* __restrict__ face_id, const TexUvDev& tuv, int b, int ox, int oy) {
  constexpr int SW = TILE * AA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = r.H, S = H * AA;
  const size_t plane = (size_t)H * H;
  const float inv = (float)(AA * AA);
#if defined(HIFIHR_RENDER_STAMP2)
  unsigned long long r2_t = __builtin_amdgcn_s_memtime();
#endif
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
  if (tid < 8) { atomicAdd(&g_r2_stamp[tid], s_r2_acc[tid]); s_r2_acc[tid] = 0ull; }
  if (tid == 0) {
    const unsigned long long dur = __builtin_amdgcn_s_memtime() - L.stamp_begin;
    atomicMax(&g_r2_stamp[10], dur);
    atomicAdd(&g_r2_stamp[9], dur);                                 // sum over the resolving items: the phases above must add up to it
    atomicAdd(&g_r2_hist[min(31, (int)(dur >> 14))], 1u);          // buckets of 16384 cycles (~7.8 us)
  }
  if (tid == 0) { atomicAdd(&g_r2_stamp[15], 1ull); atomicAdd(&g_r2_stamp[14], (unsigned long long)nbusy); atomicAdd(&g_r2_stamp[13], (unsigned long long)L.stamp_nlist); }
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
#if defined(HIFIHR_RENDER_STAMP2)
  if (tid == 0) { L.stamp_begin = r2_begin; L.stamp_nlist = nlist; }
#endif
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
    raster_pass2<AA, TILE, kF2Cap>(L, n, cols, rows);
    R2_STAMP(2)
  }
  resolve_shade2<AA, TILE, UV, kF2Cap>(L, r, fr, light_color, light_dir, rgba, face_id, tuv, b, ox, oy);
}

// ------------------------------------------------------------------------------------------------
// forward, third form (round 4): persistent workgroups on a work queue, longest lists first, long lists split.
//
// What the phase stamps of the second form said (profiles/r03_render_fwd_phase_stamps.txt): one workgroup per 8 x 8-pixel tile makes
// the launch as long as its slowest tile -- 125 us where the average covered tile takes 29 and the 2 870 covered tiles of a batch of 32
// would fill the chip for ~46 us -- and the 22 000 background tiles are 22 000 workgroup launches that store 2.3 KB each.  Here
//   f3_sched_image        (the last workgroup of render_bin_kernel to finish an image) turns the per-tile face counts into WORK ITEMS: a covered tile whose list is longer
//                         than kF3Cap faces is cut into P <= kF3Parts parts (slices of its list) that different workgroups rasterise;
//                         items are queued in four classes by faces per part, and the classes are served in that order (longest first);
//   render_fwd3_kernel    4 096 workgroups take the items round-robin in queue order (no queue head: see the kernel).  A part
//                         rasterises its slice into the workgroup's LDS depth buffer exactly as the second form did; parts of a split
//                         tile then merge into a tile-shared buffer in global memory with 64-bit atomicMin on (depth bits, face id) --
//                         order-independent, so the face ids stay bit-exact -- and the part that arrives LAST (a counter) reads the
//                         merged buffer back (atomics on both sides: nothing to fence) and resolves / shades the tile.  Nobody waits
//                         for anybody.  When the items are gone the workgroups share out the BACKGROUND: strips of tiles with empty
//                         lists are filled with 16-byte stores while the last covered tiles finish on other CUs.
// ------------------------------------------------------------------------------------------------
// background of the empty tiles of strip `it` (half a row of tiles of one image): face ids -1, pixels = background, alpha 0
template <int AA>
__device__ __forceinline__ void f3_fill_strip(const RenderDev& r, const int* __restrict__ tile_cnt, float* __restrict__ rgba,
                                              int* __restrict__ face_id, int it) {
  const int tid = threadIdx.x;
  const int H = r.H, S = H * AA, tiles = (H + 7) / 8, nt = tiles * tiles;
  const int half = it & 1, row = (it >> 1) % tiles, b = (it >> 1) / tiles;
  const int c0 = half ? tiles / 2 : 0, c1 = half ? tiles : tiles / 2;
  const size_t plane = (size_t)H * H;
  const float inv = (float)(AA * AA);
  float bgp[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {                            // the sum of AA^2 background samples over AA^2, as the covered tiles form it
    float a = 0.f;
    for (int s2 = 0; s2 < AA * AA; ++s2) a += r.bg[c];
    bgp[c] = a / inv;
  }
  const float bga = 0.f / inv;
  const int oy = row * 8, rows = min(8, H - oy);
  const bool vec = (S % 4) == 0;
  // which tiles of the strip are empty: one count per lane, ONE round trip (a load per loop trip serialised 14 latencies per strip)
  unsigned long long empty = 0ull;
  for (int t0 = c0; t0 < c1; t0 += 64) {                    // (<= 32 tiles per half row at H <= 512: one trip)
    const int tx = t0 + (tid & 63);
    const bool e = tx < c1 && tile_cnt[(size_t)b * nt + row * tiles + tx] == 0;
    if (t0 == c0) empty = __ballot(e);
  }
  for (int tx = c0; tx < c1; ++tx) {
    if (!((empty >> (tx - c0)) & 1ull)) continue;           // (uniform)
    const int ox = tx * 8, cols = min(8, H - ox);
    const int wS = cols * AA, hS = rows * AA;                                // samples of this tile
    int* fbase = face_id + ((size_t)b * S + (size_t)oy * AA) * S + (size_t)ox * AA;
    if (vec && (wS % 4) == 0) {
      const int w4 = wS / 4;
      for (int e = tid; e < hS * w4; e += kF2Threads) {
        const int y = e / w4, x = e - y * w4;
        *reinterpret_cast<int4*>(fbase + (size_t)y * S + 4 * x) = make_int4(-1, -1, -1, -1);
      }
    } else {
      for (int e = tid; e < hS * wS; e += kF2Threads) {
        const int y = e / wS, x = e - y * wS;
        fbase[(size_t)y * S + x] = -1;
      }
    }
    if (cols == 8 && (H % 4) == 0) {
      for (int e = tid; e < 4 * rows * 2; e += kF2Threads) {
        const int c = e / (rows * 2), q = e - c * rows * 2, y = q >> 1, x = (q & 1) * 4;
        const float v = c < 3 ? bgp[c] : bga;
        *reinterpret_cast<float4*>(rgba + (size_t)b * 4 * plane + (size_t)c * plane + (size_t)(oy + y) * H + (ox + x)) = make_float4(v, v, v, v);
      }
    } else {
      for (int e = tid; e < 4 * rows * cols; e += kF2Threads) {
        const int c = e / (rows * cols), q = e - c * rows * cols, y = q / cols, x = q - y * cols;
        rgba[(size_t)b * 4 * plane + (size_t)c * plane + (size_t)(oy + y) * H + (ox + x)] = c < 3 ? bgp[c] : bga;
      }
    }
  }
}

// CAP: faces per rasterisation pass.  128 for hand-sized meshes (41 faces per covered tile on the MANO mesh: one pass nearly everywhere, 36 KB of
// LDS, four workgroups per CU); 256 for dense skins (hundreds of faces per tile everywhere: the per-pass chain of barriers, scans and queue
// rounds is what a tile costs there, and half the passes beat a fourth workgroup per CU -- HIFIHR_RENDER_CAP forces either).
template <int AA, bool UV, int CAP>
__global__ __launch_bounds__(kF2Threads) void render_fwd3_kernel(RenderDev r, const float4* __restrict__ frec,
                                                                 const float* __restrict__ light_color, const float* __restrict__ light_dir,
                                                                 float* __restrict__ rgba, int* __restrict__ face_id,
                                                                 const int* __restrict__ tile_cnt, const int* __restrict__ tile_list,
                                                                 TexUvDev tuv, int nB, F3Ws w) {
  constexpr int TILE = 8, SW = TILE * AA;
  HIP_DYNAMIC_SHARED(float4, smem_raw)
  Fwd2Lds<AA, TILE, CAP>& L = *reinterpret_cast<Fwd2Lds<AA, TILE, CAP>*>(smem_raw);
  const int tid = threadIdx.x;
  const int H = r.H, S = H * AA;
  const int tiles = (H + TILE - 1) / TILE, nt = tiles * tiles;
  int ncls[kF3Classes], total = 0;
#pragma unroll
  for (int c = 0; c < kF3Classes; ++c) { ncls[c] = w.ctl[2 + c]; total += ncls[c]; }
  // Items are dealt round-robin in queue order (longest lists first): workgroup g takes items g, g + G, ...  The grid (4 096) is larger
  // than the chip holds at once, so the hardware dispatcher hands out the later, shorter items as earlier workgroups retire -- and there
  // is no queue head to contend on (one word serves ~88 atomics per microsecond: a head per item measured 40 us slower).
#if defined(HIFIHR_RENDER_STAMP2)
  if (tid < 8) s_r2_acc[tid] = 0ull;
  if (tid == 0) atomicMin(&g_r3_time[0], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
  for (int it = blockIdx.x; it < total; it += gridDim.x) {
#if defined(HIFIHR_RENDER_STAMP2)
    const unsigned long long r3_start = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { L.stamp_begin = __builtin_amdgcn_s_memtime(); }
#endif
    int cls = 0, off = it;
#pragma unroll
    for (int c = 0; c < kF3Classes - 1; ++c)
      if (cls == c && off >= ncls[c]) { off -= ncls[c]; cls = c + 1; }
    const int2 item = w.queue[(size_t)cls * w.cap + off];
    const int b = (int)((unsigned)item.x >> 18), t = (item.x >> 6) & 4095, part = (item.x >> 3) & 7, P = (item.x & 7) + 1;
    const int tix = t % tiles, tiy = t / tiles;
    const int ox = tix * TILE, oy = tiy * TILE;
    const int cols = min(TILE, H - ox), rows = min(TILE, H - oy);
    const size_t tile = (size_t)b * nt + t;
    const int nlist = tile_cnt[tile];
    const int* flist = tile_list + tile * r.F;
    const int lo = (int)((long)nlist * part / P), hi = (int)((long)nlist * (part + 1) / P);
    R2_T0
    for (int e = tid; e < 2 * SW; e += kF2Threads) {
      const int idx = e < SW ? e : e - SW;
      const int g = min((e < SW ? ox : oy) * AA + idx, S - 1);
      const float v = pix_to_ndc(S - 1 - g, S);
      if (e < SW) L.sxs[idx] = v; else L.sys[idx] = v;
    }
    for (int e = tid; e < SW * SW; e += kF2Threads) L.zbuf[e] = ~0ull;
    const float4* fr = frec + (size_t)b * r.F * kFaceRec;
    for (int base = lo; base < hi; base += CAP) {
      const int n = min(CAP, hi - base);
      __syncthreads();                                       // (previous pass done with rec; first pass: sxs / zbuf written)
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
      raster_pass2<AA, TILE, CAP>(L, n, cols, rows);
      R2_STAMP(2)
    }
    bool resolve = true;
    if (P > 1) {
      // merge this part's nearest keys into the tile's shared buffer; the part whose arrival completes the count resolves the tile
      unsigned long long* gz = w.gz + (size_t)item.y * (SW * SW);
      F3_DBG(0)
      for (int e = tid; e < SW * SW; e += kF2Threads) {
        const unsigned long long key = L.zbuf[e];
        if (key != ~0ull) atomicMin(&gz[e], key);
      }
      HIFIHR_R_WAIT_VMEM();                                  // every wave's atomics acknowledged (they execute at the memory side) ...
      __syncthreads();                                       // ... before the one arrival is counted
      if (tid == 0) L.last = (atomicAdd(&w.arrive[item.y], 1) == P - 1) ? 1 : 0;
      __syncthreads();
      resolve = L.last != 0;
      if (resolve) {
        F3_DBG(1)
        for (int e = tid; e < SW * SW; e += kF2Threads) L.zbuf[e] = atomicMin(&gz[e], ~0ull);     // (a read: atomics on both sides)
        __syncthreads();
      }
    }
#if defined(HIFIHR_RENDER_STAMP2)
    if (tid == 0) L.stamp_nlist = hi - lo;
#endif
    if (resolve) resolve_shade2<AA, TILE, UV, CAP>(L, r, fr, light_color, light_dir, rgba, face_id, tuv, b, ox, oy);
    __syncthreads();                                         // everyone is done with this item's LDS
#if defined(HIFIHR_RENDER_STAMP2)
    if (tid == 0) {
      atomicMax(&g_r3_time[1 + cls], (unsigned long long)__builtin_amdgcn_s_memrealtime());
      atomicMax(&g_r3_time[5 + cls], r3_start);
      atomicAdd(&g_r3_time[10], 1ull);
      if (P > 1) atomicAdd(&g_r3_time[11], 1ull);
      if (!resolve) {                                        // (a resolving item is counted inside resolve_shade2)
        const unsigned long long dur = __builtin_amdgcn_s_memtime() - L.stamp_begin;
        atomicAdd(&g_r2_hist[min(31, (int)(dur >> 14))], 1u);
      }
    }
#endif
  }
  // ---- the background: half rows of tiles with empty lists, dealt from the far end of the grid (the workgroups with the least to do) ----
  const int nfill = nB * tiles * 2;
  for (int it = (int)gridDim.x - 1 - (int)blockIdx.x; it < nfill; it += gridDim.x) {
    F3_DBG(2)
    f3_fill_strip<AA>(r, tile_cnt, rgba, face_id, it);
  }
#if defined(HIFIHR_RENDER_STAMP2)
  __syncthreads();
  if (tid < 8) atomicAdd(&g_r2_stamp[tid], s_r2_acc[tid]);
  if (tid == 0) atomicMax(&g_r3_time[9], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// workspace: four float4[B][V] vertex arrays, the float[B][V][12] gradient records of the backward, then the forward's per-tile face
// lists: int cnt[B][tiles^2] and int list[B][tiles^2][F] (worst case: the whole mesh inside one tile)
namespace {
struct WsClean { void* ws; bool clean; };
WsClean g_ws_clean[32] = {};
int g_ws_clean_next = 0;
// forward runs on the caller's thread, backward on autograd's device worker thread (one per device): the table is shared by all of them
std::mutex g_ws_clean_mu;
}  // namespace
void render_ws_mark_clean(void* ws, bool clean) {
  std::lock_guard<std::mutex> lock(g_ws_clean_mu);
  for (auto& e : g_ws_clean)
    if (e.ws == ws) { e.clean = clean; return; }
  g_ws_clean[g_ws_clean_next] = WsClean{ws, clean};           // (round robin: an evicted workspace just gets its memsets back)
  g_ws_clean_next = (g_ws_clean_next + 1) % 32;
}
bool render_ws_take_clean(void* ws) {
  std::lock_guard<std::mutex> lock(g_ws_clean_mu);
  for (auto& e : g_ws_clean)
    if (e.ws == ws) { const bool c = e.clean; e.clean = false; return c; }
  return false;
}

size_t render_workspace_bytes(const RenderDev& r, int B) {
  return vertex_part_bytes(r, B) + list_part_bytes(r, B) + (size_t)B * r.F * kFaceRec * sizeof(float4) + f3_part_bytes(r, B);
}
#if defined(HIFIHR_RENDER_STAMP2)
extern "C" int hifihr_debug_render_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_r2_stamp), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_r2_stamp), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
extern "C" int hifihr_debug_render_times(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_r3_time), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    z[0] = ~0ull;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_r3_time), z, sizeof(z)) != hipSuccess) return -1;
  }
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
  static const int use3 = [] { const char* e = getenv("HIFIHR_RENDER_FWD3"); return e ? atoi(e) : 1; }();     // 0: the second form (A/B)
  const bool f3 = use3 != 0 && f3_supported(r, B);
  const F3Ws w3 = f3_carve(r, B, ws);
  hipLaunchKernelGGL(render_vertex_kernel, dim3((r.V + 255) / 256, B), dim3(256), 0, st, r, verts, vcolors, vcol_bstride, cam,
                     vndc, vpos, vnrm, vcol, tile_cnt, tiles * tiles, f3 ? w3.ctl : nullptr, gvrec, light_records(r, B, ws));
  render_ws_mark_clean(ws, true);
  const dim3 bgrid((r.F + kBinFaces - 1) / kBinFaces, B);
  static const int xm = [] { const char* e = getenv("HIFIHR_RENDER_XCD"); return e ? atoi(e) : 0; }();      // A/B: images pinned to XCDs
  const dim3 grid1((unsigned)((xm ? 8 * ((B + 7) / 8) : B) * tiles * tiles));
  float4* frec = face_records(r, B, ws);
  const TexUvDev td = uv != nullptr ? TexUvDev{uv->faces_uvs, uv->verts_uvs, uv->maps, nullptr, uv->TH, uv->TW} : TexUvDev{};
  if (f3) {
    static const int cus = [] {
      int dev = 0;
      hipDeviceProp_t prop;
      return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                 ? prop.multiProcessorCount : 256;
    }();
    static const int per_cu = [] { const char* e = getenv("HIFIHR_RENDER_WGS"); const int v = e ? atoi(e) : 16; return v > 0 ? v : 16; }();
    const dim3 grid3((unsigned)(cus * per_cu));
    static const int cap_forced = [] { const char* e = getenv("HIFIHR_RENDER_CAP"); return e ? atoi(e) : 0; }();
    const int cap3 = cap_forced == 128 || cap_forced == 256 ? cap_forced : (r.F > 2048 ? 256 : kF3Cap);
#define HIFIHR_RENDER_FWD3(AA_)                                                                                                          \
    {                                                                                                                                   \
      hipLaunchKernelGGL((render_bin_kernel<AA_, 8>), bgrid, dim3(4 * kBinFaces), (size_t)2 * tiles * tiles * sizeof(int), st, r, vndc,   \
                         vpos, vnrm, vcol, frec, tile_cnt, tile_list, w3, 1);                                                           \
      if (uv != nullptr && cap3 == 256)                                                                                                 \
        hipLaunchKernelGGL((render_fwd3_kernel<AA_, true, 256>), grid3, dim3(kF2Threads), sizeof(Fwd2Lds<AA_, 8, 256>), st, r, frec,      \
                           light_color, light_dir, rgba, face_id, tile_cnt, tile_list, td, B, w3);                                      \
      else if (uv != nullptr)                                                                                                           \
        hipLaunchKernelGGL((render_fwd3_kernel<AA_, true, kF3Cap>), grid3, dim3(kF2Threads), sizeof(Fwd2Lds<AA_, 8, kF3Cap>), st, r, frec, \
                           light_color, light_dir, rgba, face_id, tile_cnt, tile_list, td, B, w3);                                      \
      else if (cap3 == 256)                                                                                                             \
        hipLaunchKernelGGL((render_fwd3_kernel<AA_, false, 256>), grid3, dim3(kF2Threads), sizeof(Fwd2Lds<AA_, 8, 256>), st, r, frec,     \
                           light_color, light_dir, rgba, face_id, tile_cnt, tile_list, td, B, w3);                                      \
      else                                                                                                                              \
        hipLaunchKernelGGL((render_fwd3_kernel<AA_, false, kF3Cap>), grid3, dim3(kF2Threads), sizeof(Fwd2Lds<AA_, 8, kF3Cap>), st, r, frec,\
                           light_color, light_dir, rgba, face_id, tile_cnt, tile_list, td, B, w3);                                      \
    }
    switch (r.aa) {
      case 1: HIFIHR_RENDER_FWD3(1) break;
      case 2: HIFIHR_RENDER_FWD3(2) break;
      case 3: HIFIHR_RENDER_FWD3(3) break;
      default: return hipErrorInvalidValue;
    }
#undef HIFIHR_RENDER_FWD3
    return hipGetLastError();
  }
#define HIFIHR_RENDER_FWD2(AA_, T_)                                                                                                      \
  {                                                                                                                                     \
    hipLaunchKernelGGL((render_bin_kernel<AA_, T_>), bgrid, dim3(4 * kBinFaces), (size_t)2 * tiles * tiles * sizeof(int), st, r, vndc,     \
                       vpos, vnrm, vcol, frec, tile_cnt, tile_list, w3, 0);                                                             \
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
