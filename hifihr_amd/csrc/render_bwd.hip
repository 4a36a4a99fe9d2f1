// Backward of the hard rasteriser + Phong shader (see csrc/render.hip for the pipeline): render_bwd_kernel reads the forward's face ids, recomputes barycentrics and shading, back-propagates to per-vertex
// records; render_vertex_bwd_kernel folds them into d(verts).  Replaces autograd through PyTorch3D's rasterize_meshes /
// interpolate_face_attributes / phong_shading / hard_rgb_blend + avg_pool2d (reference models_res_nimble.py:208-211).
#include "render_common.h"

namespace hifihr {

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// Per-tile gradient accumulators.  A tile touches a few dozen vertices of the mesh: their 12-float gradient records live in a small
// open-addressing hash table in LDS (key = vertex index, claimed with atomicCAS on first touch), added to with ds_add_f32 and flushed once
// per tile with contiguous global atomics.  Round 2 kept a record for EVERY vertex of the mesh in LDS (V x 12 floats = 37 KB for MANO:
// four workgroups per CU, 37 KB to zero and to scan per tile, and no LDS path at all for a 5 990-vertex skin, whose gradients went
// out as one global float atomic per lane per value -- 64 rows per wave instruction, ~0.08 TB/s).  A vertex that finds the table
// full (a tile with more distinct vertices than slots) falls back to global atomics for that record.
template <int HT>
struct BwdAcc {
  int keys[HT];
  float rec[HT][12];
};
template <int HT>
__device__ __forceinline__ int acc_slot(BwdAcc<HT>& A, int v) {
  unsigned s = ((unsigned)v * 2654435761u) >> 16;
#pragma unroll 1
  for (int probe = 0; probe < 16; ++probe) {
    s &= (unsigned)(HT - 1);
    const int old = atomicCAS(&A.keys[s], -1, v);
    if (old == -1 || old == v) return (int)s;
    ++s;
  }
  return -1;
}
template <int HT>
__device__ __forceinline__ void flush_face(BwdAcc<HT>& A, float* __restrict__ gglobal, const int* idx, const float* acc) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int s = acc_slot(A, idx[k]);
    if (s >= 0) {                                  // (two bodies: an address that may be LDS or global compiles to FLAT atomics)
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        const float v = acc[k * 12 + c];
        if (v != 0.f) atomicAdd(&A.rec[s][c], v);  // (all of these LDS atomics together: ~50 of the launch's ~140 us -- measured by compiling them out)
      }
    } else {
      float* dst = gglobal + (size_t)idx[k] * 12;
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        const float v = acc[k * 12 + c];
        if (v != 0.f) atomicAdd(dst + c, v);
      }
    }
  }
}

// TexturesUV: d loss / d texel.  Round 3 sent every sample's 4 texels x 3 channels to the texture gradient as global float atomics --
// 12 per sample, 78 M per launch at B = 48 on a 64 x 64 map where ~30 samples of an image meet in every texel: 1.95 ms, 0.008 of the HBM
// roof.  Now (a) a lane keeps the texel QUAD of its current sample in 12 registers and adds to it while consecutive samples fall into the
// same quad (a coarse map: most of a pixel's AA x AA samples), (b) a quad that is left goes to a per-tile open-addressing table in LDS
// keyed by the texel index (ds_add_f32; the table full: global atomics for that texel), (c) the table is flushed once per tile.
template <int TT>
struct TexAcc {
  int keys[TT];
  float rec[TT][3];
};
template <int TT>
__device__ __forceinline__ int tex_slot(TexAcc<TT>& A, int t) {
  unsigned s = ((unsigned)t * 2654435761u) >> 14;
#pragma unroll 1
  for (int probe = 0; probe < 16; ++probe) {
    s &= (unsigned)(TT - 1);
    const int old = atomicCAS(&A.keys[s], -1, t);
    if (old == -1 || old == t) return (int)s;
    ++s;
  }
  return -1;
}
template <int TT>
__device__ __forceinline__ void flush_texel(TexAcc<TT>& A, float* __restrict__ gm, int texel, const float* v) {
  if (v[0] == 0.f && v[1] == 0.f && v[2] == 0.f) return;
  const int s = tex_slot(A, texel);
  if (s >= 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) atomicAdd(&A.rec[s][c], v[c]);
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) atomicAdd(gm + (size_t)texel * 3 + c, v[c]);
  }
}
// the quad (x0, x1, y0, y1) a lane has been adding to: its four texels to the table
template <int TT>
__device__ __forceinline__ void flush_quad(TexAcc<TT>& A, float* __restrict__ gm, int TW, const int* qd, float* tacc) {
  flush_texel(A, gm, qd[2] * TW + qd[0], tacc);
  flush_texel(A, gm, qd[2] * TW + qd[1], tacc + 3);
  flush_texel(A, gm, qd[3] * TW + qd[0], tacc + 6);
  flush_texel(A, gm, qd[3] * TW + qd[1], tacc + 9);
#pragma unroll
  for (int k = 0; k < 12; ++k) tacc[k] = 0.f;
}

// 16 x 16-pixel tiles, one pixel per lane (its AA x AA samples in turn).  Measured alternative (round 3): 8 x 8-pixel tiles with the three
// sample rows of a pixel on separate lanes (a third of the serial work per lane, four times the tiles in flight, the rows' per-face
// records combined by shuffles before the LDS atomics): 199 us against 173 for this form at B = 32 -- the kernel is bound by its
// instruction and LDS-atomic throughput, not by its slowest tile: every row lane gathers and sets up its face again.
// Second measured alternative (round 3, late): FACE-parallel -- a workgroup per tile of the forward's binning, lane k takes listed face k,
// loads its records once, walks the samples of its bounding box in the tile (ids staged in LDS), accumulates the whole face-in-tile
// gradient in 36 registers without atomics, lanes of a wave converging on the gradient body each round: parity green, 1 290 us (8 x 8
// tiles; 2 500 at 16 x 16; 1 830 with scattered global atomics instead of the LDS records).  A face holds ~45 samples where a pixel holds
// 9: five times the serial depth per lane on no more lanes (118 k (face, tile) pairs against 132 k covered pixels).
template <int AA, bool UV>
__global__ __launch_bounds__(256) void render_bwd_kernel(RenderDev r, const float4* __restrict__ frec,
                                                        const float* __restrict__ light_color,
                                                        const float* __restrict__ light_dir, const int* __restrict__ face_id,
                                                        const float* __restrict__ grad_rgba, float* __restrict__ gvrec,
                                                        float* __restrict__ glight_color, float* __restrict__ glight_dir,
                                                        TexUvDev tuv) {
  constexpr int HT = 512;                                  // hash slots: > 2x the distinct vertices of a typical tile (BwdAcc)
  constexpr int TT = UV ? 2048 : 1;                       // texel slots (TexAcc; vertex-colour instantiations carry a dummy)
  __shared__ BwdAcc<HT> A;
  __shared__ TexAcc<TT> TA;
  __shared__ float red[4 * 6];
  __shared__ int any_hit[4];
  const int b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = r.H, S = H * AA;
  const int px = blockIdx.x * kTile + (lane & 7) + 8 * (wave & 1), py = blockIdx.y * kTile + (lane >> 3) + 8 * (wave >> 1);
  const bool live = (px < H) && (py < H);
  // face ids of this lane's samples; tiles without any covered sample leave at once.  (The ids are read again inside the sample loop --
  // L1 / L2 hits -- instead of being kept: the loop is not unrolled, see kRolled below.)
  const int* const fid_px = face_id + ((size_t)b * S + (size_t)py * AA) * S + (size_t)px * AA;       // sample (0, 0) of this lane's pixel
  int fid[AA * AA];                                          // (kept by the unrolled, vertex-colour form only)
  bool hit = false;
#pragma unroll
  for (int i = 0; i < AA; ++i)
#pragma unroll
    for (int j = 0; j < AA; ++j) {
      const int f = live ? fid_px[(size_t)i * S + j] : -1;
      fid[i * AA + j] = f;
      hit = hit || (f >= 0);
    }
  const unsigned long long hm = __ballot(hit);
  if (lane == 0) any_hit[wave] = (hm != 0ull);
  __syncthreads();
  if (!(any_hit[0] | any_hit[1] | any_hit[2] | any_hit[3])) return;
  for (int e = tid; e < HT; e += 256) A.keys[e] = -1;
  for (int e = tid; e < HT * 12; e += 256) (&A.rec[0][0])[e] = 0.f;
  if constexpr (UV) {
    for (int e = tid; e < TT; e += 256) TA.keys[e] = -1;
    for (int e = tid; e < TT * 3; e += 256) (&TA.rec[0][0])[e] = 0.f;
  }
  __syncthreads();
  float glc[3] = {0.f, 0.f, 0.f}, gl[3] = {0.f, 0.f, 0.f};
  LightDir Ld;
  const float raw[3] = {light_dir[3 * b], light_dir[3 * b + 1], light_dir[3 * b + 2]};
  normalize3(raw, Ld.l, &Ld.inv_norm);
  if (r.sc.point_light) { Ld.l[0] = raw[0]; Ld.l[1] = raw[1]; Ld.l[2] = raw[2]; }
  Ld.lc[0] = light_color[3 * b]; Ld.lc[1] = light_color[3 * b + 1]; Ld.lc[2] = light_color[3 * b + 2];
  const size_t vo = (size_t)b * r.V;
  float* gv = gvrec + vo * 12;                               // this image's global records (table overflow, final flush)
  float acc[36];
  int cur = -1, cidx[3] = {0, 0, 0};
  float tacc[12];                                           // (UV) gradient of the texel quad qd = (x0, x1, y0, y1) held by this lane
  int qd[4] = {-1, -1, -1, -1};
#pragma unroll
  for (int k = 0; k < 12; ++k) tacc[k] = 0.f;
  float* const gm = (UV && tuv.gmaps != nullptr) ? tuv.gmaps + (size_t)b * tuv.TH * tuv.TW * 3 : nullptr;
  if (live) {
    const size_t plane = (size_t)H * H;
    const float* g = grad_rgba + (size_t)b * 4 * plane + (size_t)py * H + px;
    const float inv = (float)(AA * AA);
    const float g_rgb[3] = {g[0] / inv, g[plane] / inv, g[2 * plane] / inv};
    FaceXYZ fc;
    fc.x0 = fc.y0 = fc.z0 = fc.x1 = fc.y1 = fc.z1 = fc.x2 = fc.y2 = fc.z2 = 0.f;
    float pos[3][3] = {}, nrm[3][3] = {}, col[3][3] = {};
    float fu[3] = {0.f, 0.f, 0.f}, fv[3] = {0.f, 0.f, 0.f};  // (UV) the face's three texture coordinates: loaded with the face
    // kRolled (TexturesUV only, below): the AA x AA sample loop is NOT unrolled.  Unrolled (round 3) the nine copies of the body cost 198 registers
    // without and 296 with TexturesUV -- two, resp. ONE 256-thread workgroup per CU for a kernel that waits on LDS atomics and gathers.
    auto sample = [&](int i, int j, int f) {
      {
        if (f < 0) return;
        const float syi = pix_to_ndc(S - 1 - (py * AA + i), S);
        const float sxj = pix_to_ndc(S - 1 - (px * AA + j), S);
        if (f != cur) {
          if (cur >= 0) flush_face(A, gv, cidx, acc);
          cur = f;
          cidx[0] = r.faces[3 * f]; cidx[1] = r.faces[3 * f + 1]; cidx[2] = r.faces[3 * f + 2];
#pragma unroll
          for (int k = 0; k < 36; ++k) acc[k] = 0.f;
          // the face's twelve vertex records, once per run of samples on this face: packed by the forward's render_bin_kernel (one level
          // of indirection, independent 16-byte loads); round 2 re-gathered them through the vertex indices for every sample
          const float4* q = frec + ((size_t)b * r.F + f) * kFaceRec;
          const float4 a = q[0], c = q[1], d = q[2];
          fc.x0 = a.x; fc.y0 = a.y; fc.z0 = a.z; fc.x1 = c.x; fc.y1 = c.y; fc.z1 = c.z; fc.x2 = d.x; fc.y2 = d.y; fc.z2 = d.z;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float4 p = q[3 + k], n = q[6 + k], t = q[9 + k];
            pos[k][0] = p.x; pos[k][1] = p.y; pos[k][2] = p.z;
            nrm[k][0] = n.x; nrm[k][1] = n.y; nrm[k][2] = n.z;
            col[k][0] = t.x; col[k][1] = t.y; col[k][2] = t.z;
            if constexpr (UV) { col[k][0] = 0.f; col[k][1] = 0.f; col[k][2] = 0.f; }          // TexturesUV: the colour is no function of these
          }
          if constexpr (UV) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { const int iu = tuv.faces_uvs[3 * f + k]; fu[k] = tuv.verts_uvs[2 * iu]; fv[k] = tuv.verts_uvs[2 * iu + 1]; }
          }
        }
        float bary[3];
        bary_of(fc, sxj, syi, bary);
        float P[3], N[3], T[3];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
          P[c3] = bary[0] * pos[0][c3] + bary[1] * pos[1][c3] + bary[2] * pos[2][c3];
          N[c3] = bary[0] * nrm[0][c3] + bary[1] * nrm[1][c3] + bary[2] * nrm[2][c3];
          T[c3] = bary[0] * col[0][c3] + bary[1] * col[1][c3] + bary[2] * col[2][c3];
        }
        float dix[3], diy[3];
        UvSample q{};
        if constexpr (UV) {
          const float u = bary[0] * fu[0] + bary[1] * fu[1] + bary[2] * fu[2], v = bary[0] * fv[0] + bary[1] * fv[1] + bary[2] * fv[2];
          q = uv_sample(u, v, tuv.TH, tuv.TW);
          uv_fetch(tuv, b, q, T, dix, diy);
        }
        float gP[3], gN[3], gT[3];
        shade_bwd(r.sc, Ld, P, N, T, g_rgb, gP, gN, gT, glc, gl);
        float guv[2] = {0.f, 0.f};                               // d loss / d (u, v) of this sample
        if constexpr (UV) {
          float gix = 0.f, giy = 0.f;
          const float w00 = (1.f - q.wx) * (1.f - q.wy), w01 = q.wx * (1.f - q.wy), w10 = (1.f - q.wx) * q.wy, w11 = q.wx * q.wy;
          if (gm != nullptr) {
            if (q.x0 != qd[0] || q.y0 != qd[2] || q.x1 != qd[1] || q.y1 != qd[3]) {
              if (qd[0] >= 0) flush_quad(TA, gm, tuv.TW, qd, tacc);
              qd[0] = q.x0; qd[1] = q.x1; qd[2] = q.y0; qd[3] = q.y1;
            }
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) {
              tacc[c3] += gT[c3] * w00; tacc[3 + c3] += gT[c3] * w01; tacc[6 + c3] += gT[c3] * w10; tacc[9 + c3] += gT[c3] * w11;
            }
          }
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) {
            gix += gT[c3] * dix[c3]; giy += gT[c3] * diy[c3];
          }
          guv[0] = q.in_x ? gix * (float)(tuv.TW - 1) : 0.f;     // d ix / d u = TW - 1; zero where grid_sample clipped the coordinate
          guv[1] = q.in_y ? giy * (float)(tuv.TH - 1) : 0.f;
        }
        float gb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          gb[k] = gP[0] * pos[k][0] + gP[1] * pos[k][1] + gP[2] * pos[k][2] + gN[0] * nrm[k][0] + gN[1] * nrm[k][1] +
                  gN[2] * nrm[k][2] + gT[0] * col[k][0] + gT[1] * col[k][1] + gT[2] * col[k][2];
          if constexpr (UV) gb[k] += guv[0] * fu[k] + guv[1] * fv[k];                            // the texel's dependence on the barycentrics
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) {
            acc[k * 12 + 3 + c3] += bary[k] * gP[c3];
            acc[k * 12 + 6 + c3] += bary[k] * gN[c3];
            acc[k * 12 + 9 + c3] += bary[k] * gT[c3];
          }
        }
        float gn[9];
        bary_bwd(fc, sxj, syi, gb, gn);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          acc[k * 12 + 0] += gn[3 * k]; acc[k * 12 + 1] += gn[3 * k + 1]; acc[k * 12 + 2] += gn[3 * k + 2];
        }
      }
    };
    if constexpr (UV) {                                       // (rolled: 296 -> 234 registers, two workgroups per CU instead of one:
#pragma unroll 1                                              //  NIMBLE-shaped mesh, B = 48: 465 -> 400 us)
      for (int i = 0; i < AA; ++i)
#pragma unroll 1
        for (int j = 0; j < AA; ++j) sample(i, j, fid_px[(size_t)i * S + j]);
    } else {                                                  // (vertex colours: 198 registers unrolled, 188 rolled -- two workgroups per CU either
                                                              //  way, and rolled measured 165 -> 176 us at B = 32)
#if !defined(HIFIHR_RBWD_SCAN_ORDER)
      // Round 6: the pixel's samples are walked in FACE order (keys (face, sample index) through an odd-even transposition network in
      // registers): a pixel that an edge crosses reads A A B / A B B / B B B in scan order -- four runs, i.e. three in-loop flushes into the
      // tile's vertex table and three reloads of twelve face records, each executed under divergence -- and two runs when sorted.  Entry point
      // at B = 32: 163 -> 145 us (timing-only builds without the in-loop flushes: 114).  -DHIFIHR_RBWD_SCAN_ORDER: the scan order (A/B builds).
      constexpr int NS = AA * AA;
      unsigned key[NS];
#pragma unroll
      for (int t = 0; t < NS; ++t) key[t] = ((fid[t] >= 0 ? (unsigned)fid[t] : 0x0FFFFFFFu) << 4) | (unsigned)t;
#pragma unroll
      for (int rnd = 0; rnd < NS; ++rnd)
#pragma unroll
        for (int t = rnd & 1; t + 1 < NS; t += 2) {
          const unsigned lo = min(key[t], key[t + 1]), hi = max(key[t], key[t + 1]);
          key[t] = lo; key[t + 1] = hi;
        }
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        const unsigned fk = key[t] >> 4;
        const int idx = (int)(key[t] & 15u);
        sample(idx / AA, idx - (idx / AA) * AA, fk == 0x0FFFFFFFu ? -1 : (int)fk);
      }
#else
#pragma unroll
      for (int i = 0; i < AA; ++i)
#pragma unroll
        for (int j = 0; j < AA; ++j) sample(i, j, fid[i * AA + j]);
#endif
    }
  }
  // The last run of every lane is flushed HERE, at a wave-uniform point -- and that is where the LDS atomics met 64 ways: the lanes of a
  // wave are an 8 x 8 block of pixels, a face covers ~5 x 5 of them, and all of its lanes added to the same 36 addresses at once (the
  // flushes inside the sample loop run with a few lanes each).  Lanes of one pixel row that hold the same face are neighbours: a
  // segmented sum over the row (three DPP steps per value, no LDS) leaves one lane per (row, face) to add.
  {
    const int face = cur;
    const int prev = row_shr1_i32(face, -2);
    const bool head = ((lane & 7) == 0) || (prev != face);
    const unsigned long long heads = __ballot(head);
    const unsigned after = (unsigned)((heads >> 1) >> lane);          // bit k: lane + 1 + k starts a segment
    const int room = 7 - (lane & 7);                                   // lanes to the right inside the 8-pixel row
    const bool ok1 = room >= 1 && (after & 1u) == 0u, ok2 = room >= 2 && (after & 3u) == 0u, ok4 = room >= 4 && (after & 15u) == 0u;
#pragma unroll
    for (int k = 0; k < 36; ++k) {
      float v = (cur >= 0) ? acc[k] : 0.f;
      const float t1 = row_shl_f32<1>(v); v += ok1 ? t1 : 0.f;
      const float t2 = row_shl_f32<2>(v); v += ok2 ? t2 : 0.f;
      const float t4 = row_shl_f32<4>(v); v += ok4 ? t4 : 0.f;
      acc[k] = v;
    }
    if (head && cur >= 0) flush_face(A, gv, cidx, acc);
  }
  if constexpr (UV) {
    if (gm != nullptr && qd[0] >= 0) flush_quad(TA, gm, tuv.TW, qd, tacc);
  }
  __syncthreads();
  if constexpr (UV) {
    if (gm != nullptr) {
      for (int e = tid; e < TT * 3; e += 256) {
        const int key = TA.keys[e / 3];
        if (key >= 0) {
          const float v = (&TA.rec[0][0])[e];
          if (v != 0.f) atomicAdd(gm + (size_t)key * 3 + (e % 3), v);
        }
      }
    }
  }
  for (int e = tid; e < HT * 12; e += 256) {
    const int key = A.keys[e / 12];
    if (key >= 0) {
      const float v = (&A.rec[0][0])[e];
      if (v != 0.f) atomicAdd(gv + (size_t)key * 12 + (e % 12), v);
    }
  }
  // ---- light gradients: workgroup reduction, one atomic set per tile ----
  float v6[6] = {glc[0], glc[1], glc[2], gl[0], gl[1], gl[2]};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float s = wsum(v6[k]);
    if (lane == 0) red[wave * 6 + k] = s;
  }
  __syncthreads();
  if (tid == 0) {
    float t[6];
    for (int k = 0; k < 6; ++k) t[k] = red[k] + red[6 + k] + red[12 + k] + red[18 + k];
    if (t[0] != 0.f || t[1] != 0.f || t[2] != 0.f || t[3] != 0.f || t[4] != 0.f || t[5] != 0.f) {
      for (int k = 0; k < 3; ++k) atomicAdd(glight_color + 3 * b + k, t[k]);
      float gd[3];
      normalize3_bwd(raw, Ld.l, Ld.inv_norm, t + 3, gd);       // through F.normalize(direction)
      for (int k = 0; k < 3; ++k) atomicAdd(glight_dir + 3 * b + k, gd[k]);
    }
  }
}

// per vertex: fold the per-vertex gradient records into d(verts) (and d(vertex colours))
__global__ __launch_bounds__(256) void render_vertex_bwd_kernel(RenderDev r, const float* __restrict__ verts,
                                                               const float* __restrict__ cam, const float4* __restrict__ vndc,
                                                               const float4* __restrict__ vnrm, const float* __restrict__ gvrec,
                                                               float* __restrict__ gverts, float* __restrict__ gvcolors,
                                                               const float* __restrict__ glrec, float* __restrict__ glight_color,
                                                               float* __restrict__ glight_dir) {
  const int b = blockIdx.y;
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 3) {                  // the light gradients leave the workspace's accumulators
    glight_color[3 * b + threadIdx.x] = glrec[3 * b + threadIdx.x];
    glight_dir[3 * b + threadIdx.x] = glrec[3 * (gridDim.y + b) + threadIdx.x];
  }
  if (v >= r.V) return;
  const size_t vo = (size_t)b * r.V;
  const float* vb = verts + vo * 3;
  const float* g = gvrec + (vo + v) * 12;
  const float Z = vb[3 * v + 2];
  const float fx = cam[4 * b], fy = cam[4 * b + 1], px = cam[4 * b + 2], py = cam[4 * b + 3];
  const float4 nd = vndc[vo + v];
  // x = (X fx + Z px) / Z ; y likewise ; z = Z
  float gx = g[3] + g[0] * fx / Z;
  float gy = g[4] + g[1] * fy / Z;
  float gz = g[5] + g[2] + g[0] * (px - nd.x) / Z + g[1] * (py - nd.y) / Z;
  // vertex normals: n = normalize(sum_f cross(v2 - v1, v0 - v1))
  for (int e = r.vf_off[v]; e < r.vf_off[v + 1]; ++e) {
    const int f = r.vf_idx[e] >> 2, role = r.vf_idx[e] & 3;
    const int id[3] = {r.faces[3 * f], r.faces[3 * f + 1], r.faces[3 * f + 2]};
    float gfn[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 n = vnrm[vo + id[k]];
      const float* gk = gvrec + (vo + id[k]) * 12 + 6;
      const float gn[3] = {gk[0], gk[1], gk[2]};
      if (n.w < 1.0f / kNormEps) {
        const float d = n.x * gn[0] + n.y * gn[1] + n.z * gn[2];
        gfn[0] += (gn[0] - n.x * d) * n.w; gfn[1] += (gn[1] - n.y * d) * n.w; gfn[2] += (gn[2] - n.z * d) * n.w;
      } else {
        gfn[0] += gn[0] * n.w; gfn[1] += gn[1] * n.w; gfn[2] += gn[2] * n.w;
      }
    }
    const float* q0 = vb + 3 * id[0];
    const float* q1 = vb + 3 * id[1];
    const float* q2 = vb + 3 * id[2];
    const float A[3] = {q2[0] - q1[0], q2[1] - q1[1], q2[2] - q1[2]};
    const float Bv[3] = {q0[0] - q1[0], q0[1] - q1[1], q0[2] - q1[2]};
    // fn = A x Bv :  gA = Bv x gfn ,  gB = gfn x A
    const float gA[3] = {Bv[1] * gfn[2] - Bv[2] * gfn[1], Bv[2] * gfn[0] - Bv[0] * gfn[2], Bv[0] * gfn[1] - Bv[1] * gfn[0]};
    const float gB[3] = {gfn[1] * A[2] - gfn[2] * A[1], gfn[2] * A[0] - gfn[0] * A[2], gfn[0] * A[1] - gfn[1] * A[0]};
    if (role == 0) { gx += gB[0]; gy += gB[1]; gz += gB[2]; }
    else if (role == 2) { gx += gA[0]; gy += gA[1]; gz += gA[2]; }
    else { gx -= gA[0] + gB[0]; gy -= gA[1] + gB[1]; gz -= gA[2] + gB[2]; }
  }
  float* o = gverts + (vo + v) * 3;
  o[0] = gx; o[1] = gy; o[2] = gz;
  if (gvcolors) {
    float* oc = gvcolors + (vo + v) * 3;
    oc[0] = g[9]; oc[1] = g[10]; oc[2] = g[11];
  }
}

hipError_t launch_render_bwd(const RenderDev& r, const float* verts, const float* cam, const float* light_color,
                             const float* light_dir, const int* face_id, const float* grad_rgba, int B, float* gverts,
                             float* gvcolors, float* glight_color, float* glight_dir, void* ws, hipStream_t st, const TexUvPass* uv) {
  float4 *vndc, *vpos, *vnrm, *vcol;
  float* gvrec;
  carve(r, B, ws, &vndc, &vpos, &vnrm, &vcol, &gvrec);
  (void)vpos; (void)vcol;
  // the tile kernel accumulates (atomics) into the gradient records and into light accumulators [B][3] + [B][3] in the workspace; the forward's
  // vertex kernel left both zeroed unless this is a second backward after one forward (render_common.h: render_ws_mark_clean)
  float* const glrec = light_records(r, B, ws);
  if (!render_ws_take_clean(ws)) {
    hipError_t e = hipMemsetAsync(gvrec, 0, (size_t)B * r.V * 12 * sizeof(float), st);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(glrec, 0, (size_t)B * 6 * sizeof(float), st)) != hipSuccess) return e;
  }
  const int tiles = (r.H + kTile - 1) / kTile;
  const dim3 grid(tiles, tiles, B);
  const float4* frec = face_records(r, B, ws);                // written by the forward of the same (handle, workspace, batch)
  const TexUvDev td = uv != nullptr ? TexUvDev{uv->faces_uvs, uv->verts_uvs, uv->maps, uv->gmaps, uv->TH, uv->TW} : TexUvDev{};
#define HIFIHR_RENDER_BWD(AA_, UV_)                                                                                                      \
  hipLaunchKernelGGL((render_bwd_kernel<AA_, UV_>), grid, dim3(256), 0, st, r, frec, light_color, light_dir, face_id, grad_rgba, gvrec,   \
                     glrec, glrec + (size_t)3 * B, td)
  switch (r.aa) {
    case 1: if (uv != nullptr) { HIFIHR_RENDER_BWD(1, true); } else { HIFIHR_RENDER_BWD(1, false); } break;
    case 2: if (uv != nullptr) { HIFIHR_RENDER_BWD(2, true); } else { HIFIHR_RENDER_BWD(2, false); } break;
    case 3: if (uv != nullptr) { HIFIHR_RENDER_BWD(3, true); } else { HIFIHR_RENDER_BWD(3, false); } break;
    default: return hipErrorInvalidValue;
  }
#undef HIFIHR_RENDER_BWD
  hipLaunchKernelGGL(render_vertex_bwd_kernel, dim3((r.V + 255) / 256, B), dim3(256), 0, st, r, verts, cam, vndc, vnrm, gvrec, gverts, gvcolors,
                     glrec, glight_color, glight_dir);
  return hipGetLastError();
}

}  // namespace hifihr
