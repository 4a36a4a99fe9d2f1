// MANO linear-blend skinning for gfx950: forward, backward, joint regression.
//
// Replaces ManoLayer.forward (reference utils/my_mano.py:315-483), which issues ~60 small ATen launches
// and one host sync per call, with ONE launch per direction:
//   mano_fwd_kernel   grid (kTiles, B) x 256 threads.  Every workgroup redoes the tiny per-hand
//                     prologue (PCA 45x45, 16 Rodrigues, kinematic chain) in LDS, then each thread blends and
//                     skins one vertex.  Tables are stored structure-of-arrays over the vertex index
//                     ([row][xyz][vertex]) so a wave reads 256 contiguous bytes per table row; the 1.4 MB of
//                     tables stay L2-resident across the batch.  HBM-bound on verts/v_posed writes.
//   mano_bwd_kernel   grid B x 512 threads, deterministic (no float atomics): per-joint reductions are done
//                     as an in-LDS (16x778)x(778x12) product, the blend-shape transposes as wave dot products.
//   mano_joints_*     xyz_from_vertice + root-relative step (models_res_nimble.py:153,160-166).
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"
#include "mano_math.h"

namespace hifihr {

__device__ __constant__ int c_reorder21[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};
// xyz_from_vertice: FreiHAND slot -> regressed MANO joint id (>=0) or -(vertex id) - 1 for the five tips
__device__ __constant__ int c_xyz_src[21] = {0, 13, 14, 15, -744 - 1, 1, 2, 3, -320 - 1, 4, 5, 6, -443 - 1,
                                             10, 11, 12, -555 - 1, 7, 8, 9, -672 - 1};

struct ManoSmall {          // per-hand state kept in LDS
  float pose[48];
  float beta[kNB];
  float fp[48];             // full axis-angle pose
  float Rl[kNJ * 9];
  float Rg[kNJ * 9];
  float tg[kNJ * 3];
  float J[kNJ * 3];
  float Ap[kNJ * 12];
  float pm[kNP];            // pose_map = R - I for joints 1..15
};

// Prologue shared by forward and backward.  Needs >= 128 threads; ends with a barrier.
__device__ __forceinline__ void mano_prologue(const ManoDev& t, const float* __restrict__ pose,
                                              const float* __restrict__ beta, int b, ManoSmall& s) {
  const int tid = threadIdx.x;
  if (tid < 48) s.pose[tid] = pose[b * 48 + tid];
  if (tid >= 64 && tid < 64 + kNB) s.beta[tid - 64] = beta[b * kNB + tid - 64];
  __syncthreads();
  if (tid < kNPCA) {                       // my_mano.py:340-348: coeffs.mm(selected_comps) + hands_mean
    float acc = 0.f;
    for (int k = 0; k < kNPCA; ++k) acc += s.pose[3 + k] * t.comps[k * kNPCA + tid];
    s.fp[3 + tid] = t.mean[tid] + acc;
  } else if (tid < 48) {
    s.fp[tid - kNPCA] = s.pose[tid - kNPCA];
  }
  if (tid >= 64 && tid < 64 + 48) {        // J = J_regressor (v_template + shapedirs beta), my_mano.py:386-389
    const int e = tid - 64;
    float acc = t.jt[e];
    for (int k = 0; k < kNB; ++k) acc += t.jsd[e * kNB + k] * s.beta[k];
    s.J[e] = acc;
  }
  __syncthreads();
  if (tid < kNJ) {
    rodrigues_fwd(s.fp + 3 * tid, s.Rl + 9 * tid, nullptr);
    if (tid >= 1) {
      for (int k = 0; k < 9; ++k) s.pm[(tid - 1) * 9 + k] = s.Rl[9 * tid + k] - ((k % 4 == 0) ? 1.f : 0.f);
    }
  }
  __syncthreads();
  if (tid == 0) chain_fwd_root(s.Rl, s.J, s.Rg, s.tg);
  __syncthreads();
  if (tid < 5) chain_fwd_finger(tid, s.Rl, s.J, s.Rg, s.tg);
  __syncthreads();
  if (tid < kNJ) chain_make_ap(tid, s.J, s.Rg, s.tg, s.Ap);
  __syncthreads();
}

// Round 4: 7 tiles of 112 vertices on 128-thread workgroups (round 3: 4 tiles of 195 on 256 threads).  The kernel is a chain of load
// latencies per workgroup -- the per-hand prologue, then 145 table rows per vertex -- and a batch of 32 hands gave the 256-CU chip 128
// workgroups; 224 smaller ones with all 27 rows of a group of pose blend shapes in flight per lane (5 groups instead of 27 of 5) shorten
// the chain.  Every workgroup redoes the prologue (16 Rodrigues + the kinematic chain: ~2 us).
constexpr int kFwdTiles = 7;
constexpr int kFwdTileV = 112;   // 7 x 112 = 784 >= 778
constexpr int kFwdThreads = 128;

__global__ __launch_bounds__(kFwdThreads) void mano_fwd_kernel(ManoDev t, const float* __restrict__ pose,
                                                      const float* __restrict__ beta, float* __restrict__ verts,
                                                      float* __restrict__ jtr, float* __restrict__ saved_vposed) {
  __shared__ ManoSmall s;
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  mano_prologue(t, pose, beta, b, s);

  const float cx = s.tg[12], cy = s.tg[13], cz = s.tg[14];   // centre = joint 9 of the 21 = MANO joint 4
  if (blockIdx.x == 0 && tid < 21 && jtr) {
    const int src = c_reorder21[tid];
    if (src < kNJ) {
      float* o = jtr + ((size_t)b * 21 + tid) * 3;
      o[0] = s.tg[3 * src] - cx; o[1] = s.tg[3 * src + 1] - cy; o[2] = s.tg[3 * src + 2] - cz;
    }
  }
  const int v = blockIdx.x * kFwdTileV + tid;
  if (tid < kFwdTileV && v < kNV) {

  // v_posed = v_template + shapedirs.beta + posedirs.pose_map      (my_mano.py:386-393)
  float vp[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) vp[c] = t.tmpl[c * kNVP + v];
  for (int k = 0; k < kNB; ++k) {
    const float bk = s.beta[k];
    const float* row = t.sd + (size_t)k * 3 * kNVP + v;
    vp[0] += row[0] * bk; vp[1] += row[kNVP] * bk; vp[2] += row[2 * kNVP] * bk;
  }
  static_assert(kNP % 27 == 0, "pose blend shapes in groups of 27");
  for (int p0 = 0; p0 < kNP; p0 += 27) {
    float r[27][3];
#pragma unroll
    for (int u = 0; u < 27; ++u) {                        // 81 independent loads, then the multiply-adds in the table's order
      const float* row = t.pd + (size_t)(p0 + u) * 3 * kNVP + v;
      r[u][0] = row[0]; r[u][1] = row[kNVP]; r[u][2] = row[2 * kNVP];
    }
#pragma unroll
    for (int u = 0; u < 27; ++u) {
      const float pk = s.pm[p0 + u];
      vp[0] += r[u][0] * pk; vp[1] += r[u][1] * pk; vp[2] += r[u][2] * pk;
    }
  }
  // T = sum_i w_i A'_i ; vert = T [v_posed;1]                        (my_mano.py:441-451)
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = 0.f;
  for (int i = 0; i < kNJ; ++i) {
    const float w = t.w[i * kNVP + v];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] += w * s.Ap[12 * i + k];
  }
  float o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) o[r] = T[4 * r] * vp[0] + T[4 * r + 1] * vp[1] + T[4 * r + 2] * vp[2] + T[4 * r + 3];
  o[0] -= cx; o[1] -= cy; o[2] -= cz;
  float* ov = verts + ((size_t)b * kNV + v) * 3;
  ov[0] = o[0]; ov[1] = o[1]; ov[2] = o[2];
  if (saved_vposed) {
    float* sv = saved_vposed + ((size_t)b * kNV + v) * 3;
    sv[0] = vp[0]; sv[1] = vp[1]; sv[2] = vp[2];
  }
  if (jtr) {                                                        // finger tips, my_mano.py:457
    int tip = -1;
    if (v == 745) tip = 0; else if (v == 317) tip = 1; else if (v == 444) tip = 2; else if (v == 556) tip = 3; else if (v == 673) tip = 4;
    if (tip >= 0) {
      float* oj = jtr + ((size_t)b * 21 + 4 + 4 * tip) * 3;         // slot with REORDER21[slot] == 16 + tip
      oj[0] = o[0]; oj[1] = o[1]; oj[2] = o[2];
    }
  }
  }   // (vertex lanes)
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
constexpr int kBwdThreads = 1024;   // 16 waves: the kernel is a chain of load latencies on ONE workgroup per hand; more waves and all loads of a
                                    // table row in flight at once (phase 3) are what shorten it

struct ManoBwdLds {
  ManoSmall s;
  alignas(16) float gv[3 * kNVP];      // incoming vertex gradient (incl. tips and centring), SoA
  alignas(16) float vp[3 * kNVP];      // saved v_posed, SoA
  alignas(16) float gvp[3 * kNVP];     // gradient wrt v_posed, SoA (same indexing as a table row)
  float gAp[kNJ * 12];
  float gRg[kNJ * 9];
  float gtg[kNJ * 3];
  float gJ[kNJ * 3];
  float gRl[kNJ * 9];
  float gpm[kNP];
  float gbeta_blend[kNB];
  float gfp[48];
  float root_acc[5 * 15];  // per finger: gRg0[9], gtg0[3], gJ0[3]
  float gcenter[3];
  float red[kBwdThreads / 64 * 3];
  float gj21[63];          // fused head (ManoBwdHead): gradient wrt the 21 root-relative joints (+ the root's share)
  float gj16[kNJ * 3];     //   ... folded onto the 16 regressed joints
  float gr[3];             //   ... wrt the root joint
};

// gradient inputs of the fused form (hifihr_mano_full_bwd): the joint regression's backward as the head of this kernel; on == 0: the plain layer (gverts / gjtr)
struct ManoBwdHead {
  int on, root_id;
  const float* gjoints_rel;    // [B][21][3] or null
  const float* gverts_rel;     // [B][778][3] or null
  const float* gverts_cam;     // [B][778][3] or null
  const float* groot;          // [B][3] or null
  const float* gpose_add;      // [B][48] or null: added to gpose (the gradient that reaches `pose` through its OTHER consumer,
  const float* gbeta_add;      // [B][10] or null   e.g. the mpose / mshape regularisers: autograd's accumulation launches folded in)
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}

__global__ __launch_bounds__(kBwdThreads) void mano_bwd_kernel(ManoDev t, const float* __restrict__ pose,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ saved_vposed,
                                                             const float* __restrict__ gverts,
                                                             const float* __restrict__ gjtr,
                                                             float* __restrict__ gpose, float* __restrict__ gbeta, ManoBwdHead head) {
  HIP_DYNAMIC_SHARED(float4, smem_raw)   // float4: 16-byte aligned base (ds_read_b128 in phase 3)
  ManoBwdLds& L = *reinterpret_cast<ManoBwdLds*>(smem_raw);
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  mano_prologue(t, pose, beta, b, L.s);

  // ---- phase 0 (head.on): mano_joints_bwd_kernel folded in: from (gjoints_rel, gverts_rel,
  //      gverts_cam, groot) to the gradient wrt the layer's vertices, formed per vertex in registers and handed to phase 1 ----
  float g0[3] = {0.f, 0.f, 0.f};
  if (head.on) {                                                    // (uniform)
    static_assert(kBwdThreads >= kNVP, "a vertex per thread");
    const int v = tid;
    if (v < kNV) {
      if (head.gverts_rel) { const float* p = head.gverts_rel + ((size_t)b * kNV + v) * 3; g0[0] = p[0]; g0[1] = p[1]; g0[2] = p[2]; }
      if (head.gverts_cam) { const float* p = head.gverts_cam + ((size_t)b * kNV + v) * 3; g0[0] += p[0]; g0[1] += p[1]; g0[2] += p[2]; }
    }
    if (tid < 63) L.gj21[tid] = head.gjoints_rel ? head.gjoints_rel[(size_t)b * 63 + tid] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float r = wave_sum(g0[c]);
      if (lane == 0) L.red[wave * 3 + c] = r;
    }
    __syncthreads();
    if (tid < 3) {                                                  // gradient wrt the root joint: groot - sum gjoints_rel - sum (gverts_rel + gverts_cam)
      float a = 0.f;
      if (head.root_id >= 0) {
        a = head.groot ? head.groot[b * 3 + tid] : 0.f;
        for (int w = 0; w < kBwdThreads / 64; ++w) a -= L.red[w * 3 + tid];
        for (int j = 0; j < 21; ++j) a -= L.gj21[j * 3 + tid];
      }
      L.gr[tid] = a;
    }
    __syncthreads();
    if (tid < 3 && head.root_id >= 0) L.gj21[head.root_id * 3 + tid] += L.gr[tid];
    __syncthreads();
    if (tid < kNJ * 3) {
      const int j = tid / 3, c = tid % 3;
      float a = 0.f;
      for (int sl = 0; sl < 21; ++sl)
        if (c_xyz_src[sl] == j) a += L.gj21[sl * 3 + c];
      L.gj16[tid] = a;
    }
    __syncthreads();
    if (v < kNV) {
      float wj[kNJ];
#pragma unroll
      for (int j = 0; j < kNJ; ++j) wj[j] = t.jreg[j * kNVP + v];
#pragma unroll
      for (int j = 0; j < kNJ; ++j) { g0[0] += wj[j] * L.gj16[j * 3]; g0[1] += wj[j] * L.gj16[j * 3 + 1]; g0[2] += wj[j] * L.gj16[j * 3 + 2]; }
      for (int sl = 4; sl < 21; sl += 4)
        if (-c_xyz_src[sl] - 1 == v) { g0[0] += L.gj21[sl * 3]; g0[1] += L.gj21[sl * 3 + 1]; g0[2] += L.gj21[sl * 3 + 2]; }
    }
    __syncthreads();                                                // (L.red is reused by phase 1)
  }

  // ---- phase 1: stage gv (+ tip gradients) and v_posed; centre gradient = -(sum gverts + sum gjtr) ----
  float csum[3] = {0.f, 0.f, 0.f};
  for (int v = tid; v < kNVP; v += kBwdThreads) {
    float g[3] = {0.f, 0.f, 0.f}, p[3] = {0.f, 0.f, 0.f};
    if (v < kNV) {
      if (head.on) {
        g[0] = g0[0]; g[1] = g0[1]; g[2] = g0[2];                   // (one trip of this loop: v == tid)
      } else if (gverts) {
        const float* gp = gverts + ((size_t)b * kNV + v) * 3;
        g[0] = gp[0]; g[1] = gp[1]; g[2] = gp[2];
      }
      const float* sp = saved_vposed + ((size_t)b * kNV + v) * 3;
      p[0] = sp[0]; p[1] = sp[1]; p[2] = sp[2];
      csum[0] += g[0]; csum[1] += g[1]; csum[2] += g[2];           // centring acts on verts before tips are read
      if (gjtr) {
        int tip = -1;
        if (v == 745) tip = 0; else if (v == 317) tip = 1; else if (v == 444) tip = 2; else if (v == 556) tip = 3; else if (v == 673) tip = 4;
        if (tip >= 0) {
          const float* gj = gjtr + ((size_t)b * 21 + 4 + 4 * tip) * 3;
          g[0] += gj[0]; g[1] += gj[1]; g[2] += gj[2];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { L.gv[c * kNVP + v] = g[c]; L.vp[c * kNVP + v] = p[c]; }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r = wave_sum(csum[c]);
    if (lane == 0) L.red[wave * 3 + c] = r;
  }
  __syncthreads();
  if (tid < 3) {
    float acc = 0.f;
    for (int w = 0; w < kBwdThreads / 64; ++w) acc += L.red[w * 3 + tid];
    if (gjtr)
      for (int j = 0; j < 21; ++j) acc += gjtr[((size_t)b * 21 + j) * 3 + tid];
    L.gcenter[tid] = -acc;        // d/d(centre): every output had the centre subtracted
  }
  // ---- phase 2a: gAp[i][4r+c] = sum_v w[v,i] gv[r][v] [vp;1][c][v].  Round 4: one WAVE per joint -- its weight row is read once (13
  //      coalesced loads per lane, all in flight) and feeds the joint's 12 sums, each folded by a wave reduction.  (Round 3: 192 sums x 4
  //      lanes, each lane walking a quarter of the vertices with its own loads of the weight row: ~25 dependent L2 round trips.) ----
  {
    constexpr int kIt = (kNV + 63) / 64;
    static_assert(kBwdThreads / 64 >= kNJ, "a wave per joint");
    if (wave < kNJ) {
      const float* wrow = t.w + wave * kNVP;
      float wv[kIt];
#pragma unroll
      for (int i = 0; i < kIt; ++i) { const int v = lane + 64 * i; wv[i] = v < kNV ? wrow[v] : 0.f; }
      float acc[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k] = 0.f;
#pragma unroll
      for (int i = 0; i < kIt; ++i) {
        const int v = lane + 64 * i;
        if (v < kNV) {
          const float w = wv[i];
          const float p[4] = {L.vp[v], L.vp[kNVP + v], L.vp[2 * kNVP + v], 1.f};
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const float wg = w * L.gv[r * kNVP + v];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[4 * r + c] += wg * p[c];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const float r = wave_sum(acc[k]);
        if (lane == 0) L.gAp[wave * 12 + k] = r;
      }
    }
  }
  // ---- phase 2b: gvp[v] = sum_i w[v,i] Rg_i^T gv[v]  (a vertex per thread, its 16 weights in flight together) ----
  for (int v = tid; v < kNVP; v += kBwdThreads) {
    float o[3] = {0.f, 0.f, 0.f};
    if (v < kNV) {
      const float g[3] = {L.gv[v], L.gv[kNVP + v], L.gv[2 * kNVP + v]};
      float wv[kNJ];
#pragma unroll
      for (int i = 0; i < kNJ; ++i) wv[i] = t.w[i * kNVP + v];
#pragma unroll
      for (int i = 0; i < kNJ; ++i) {
        const float w = wv[i];
        const float* R = L.s.Rg + 9 * i;
        o[0] += w * (R[0] * g[0] + R[3] * g[1] + R[6] * g[2]);
        o[1] += w * (R[1] * g[0] + R[4] * g[1] + R[7] * g[2]);
        o[2] += w * (R[2] * g[0] + R[5] * g[1] + R[8] * g[2]);
      }
    }
    L.gvp[v] = o[0]; L.gvp[kNVP + v] = o[1]; L.gvp[2 * kNVP + v] = o[2];
  }
  __syncthreads();

  // ---- phase 3: blend-shape transposes: gpm[p] = <posedirs row p, gvp>, gbeta[k] = <shapedirs row k, gvp>.  A wave walks its rows two at
  //      a time (20 float4 per lane in flight): the rows come from L2 / HBM (1.4 MB of tables per hand-workgroup) and every trip is one
  //      memory latency ----
  {
    constexpr int kRow4 = 3 * kNVP / 4;     // float4 per table row
    const float4* g4 = reinterpret_cast<const float4*>(L.gvp);
    constexpr int kIt = (kRow4 + 63) / 64;       // 10 float4 per lane per row: all issued before the first is used
    constexpr int kW = kBwdThreads / 64;
    for (int row0 = wave; row0 < kNP + kNB; row0 += 2 * kW) {
      const int row1 = row0 + kW;
      const bool two = row1 < kNP + kNB;
      const float* base0 = (row0 < kNP) ? t.pd + (size_t)row0 * 3 * kNVP : t.sd + (size_t)(row0 - kNP) * 3 * kNVP;
      const float* base1 = !two ? base0 : ((row1 < kNP) ? t.pd + (size_t)row1 * 3 * kNVP : t.sd + (size_t)(row1 - kNP) * 3 * kNVP);
      const float4* r40 = reinterpret_cast<const float4*>(base0);
      const float4* r41 = reinterpret_cast<const float4*>(base1);
      float4 a0[kIt], a1[kIt];
#pragma unroll
      for (int i = 0; i < kIt; ++i) {
        const int e = lane + 64 * i;
        a0[i] = r40[e < kRow4 ? e : kRow4 - 1];
        a1[i] = r41[e < kRow4 ? e : kRow4 - 1];
      }
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int i = 0; i < kIt; ++i) {
        const int e = lane + 64 * i;
        if (e < kRow4) {
          const float4 g = g4[e];
          acc0 += a0[i].x * g.x + a0[i].y * g.y + a0[i].z * g.z + a0[i].w * g.w;
          acc1 += a1[i].x * g.x + a1[i].y * g.y + a1[i].z * g.z + a1[i].w * g.w;
        }
      }
      acc0 = wave_sum(acc0);
      acc1 = wave_sum(acc1);
      if (lane == 0) {
        if (row0 < kNP) L.gpm[row0] = acc0; else L.gbeta_blend[row0 - kNP] = acc0;
        if (two) { if (row1 < kNP) L.gpm[row1] = acc1; else L.gbeta_blend[row1 - kNP] = acc1; }
      }
    }
  }
  // ---- phase 4: fold gAp into the chain, walk the chain backwards ----
  if (tid < kNJ * 9) L.gRg[tid] = 0.f;
  if (tid < kNJ * 3) { L.gtg[tid] = 0.f; L.gJ[tid] = 0.f; }
  if (tid < 5 * 15) L.root_acc[tid] = 0.f;
  __syncthreads();
  if (tid < kNJ) {
    chain_make_ap_bwd(tid, L.s.J, L.s.Rg, L.gAp, L.gRg, L.gtg, L.gJ);
    // joints of the 21-vector that are chain translations: slot s has source REORDER21[s] < 16
    if (gjtr) {
      for (int sl = 0; sl < 21; ++sl) {
        if (c_reorder21[sl] == tid) {
          const float* gj = gjtr + ((size_t)b * 21 + sl) * 3;
          L.gtg[3 * tid] += gj[0]; L.gtg[3 * tid + 1] += gj[1]; L.gtg[3 * tid + 2] += gj[2];
        }
      }
    }
    if (tid == 4) { L.gtg[12] += L.gcenter[0]; L.gtg[13] += L.gcenter[1]; L.gtg[14] += L.gcenter[2]; }
  }
  __syncthreads();
  if (tid < 5) {
    float* acc = L.root_acc + 15 * tid;
    chain_bwd_finger(tid, L.s.Rl, L.s.J, L.s.Rg, L.gRg, L.gtg, L.gRl, L.gJ, acc, acc + 9, acc + 12);
  }
  __syncthreads();
  if (tid < 15) {               // root: Rg_0 = Rl_0, tg_0 = J_0
    float a = 0.f;
    for (int f = 0; f < 5; ++f) a += L.root_acc[15 * f + tid];
    if (tid < 9) L.gRl[tid] = L.gRg[tid] + a;
    else if (tid < 12) L.root_acc[tid] = L.gtg[tid - 9] + a;      // reuse slot: total g(tg_0)
    else L.root_acc[tid] = L.gJ[tid - 12] + a;                    // gJ_0 (chain part)
  }
  __syncthreads();
  if (tid < 3) L.gJ[tid] = L.root_acc[12 + tid] + L.root_acc[9 + tid];
  // pose_map gradient goes to the local rotations of joints 1..15 (pose_map = Rl - I)
  if (tid >= 64 && tid < 64 + kNP) L.gRl[9 + (tid - 64)] += L.gpm[tid - 64];
  __syncthreads();
  if (tid < kNJ) rodrigues_bwd(L.s.fp + 3 * tid, L.gRl + 9 * tid, L.gfp + 3 * tid);
  __syncthreads();
  // ---- phase 5: PCA transpose and shape gradient ----
  if (tid < 3) {
    gpose[b * 48 + tid] = L.gfp[tid] + (head.gpose_add ? head.gpose_add[b * 48 + tid] : 0.f);
  } else if (tid < 48) {
    const int k = tid - 3;
    float acc = 0.f;
    for (int j = 0; j < kNPCA; ++j) acc += t.comps[k * kNPCA + j] * L.gfp[3 + j];
    gpose[b * 48 + tid] = acc + (head.gpose_add ? head.gpose_add[b * 48 + tid] : 0.f);
  } else if (tid >= 64 && tid < 64 + kNB) {
    const int k = tid - 64;
    float acc = L.gbeta_blend[k];
    for (int e = 0; e < kNJ * 3; ++e) acc += t.jsd[e * kNB + k] * L.gJ[e];
    gbeta[b * kNB + k] = acc + (head.gbeta_add ? head.gbeta_add[b * kNB + k] : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------
// xyz_from_vertice + root-relative
// ------------------------------------------------------------------------------------------------
constexpr int kJointsFwdThreads = 1024;   // 16 waves: the 48 regressor dot products (one wave each) take 3 rounds instead of 12; one
                                          // workgroup per hand is all the parallelism a batch of 32 offers, so the kernel is a latency chain
__global__ __launch_bounds__(kJointsFwdThreads) void mano_joints_fwd_kernel(ManoDev t, const float* __restrict__ verts, int root_id,
                                                             float* __restrict__ joints_rel, float* __restrict__ verts_rel,
                                                             float* __restrict__ root_out, const float* __restrict__ root_xyz,
                                                             float* __restrict__ verts_cam) {
  __shared__ float sv[3 * kNVP];
  __shared__ float j16[kNJ * 3];
  __shared__ float j21[21 * 3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int v = tid; v < kNV; v += kJointsFwdThreads) {
    const float* p = verts + ((size_t)b * kNV + v) * 3;
    sv[v] = p[0]; sv[kNVP + v] = p[1]; sv[2 * kNVP + v] = p[2];
  }
  __syncthreads();
  // one wave per regressed joint: its regressor row is read once (13 coalesced loads per lane, all in flight) and feeds the three
  // coordinate sums (round 3: 48 wave dot products in three rounds, each with its own walk over the row)
  static_assert(kJointsFwdThreads / 64 >= kNJ, "a wave per joint");
  if (wave < kNJ) {
    constexpr int kIt = (kNV + 63) / 64;
    const float* jr = t.jreg + wave * kNVP;
    float rv[kIt];
#pragma unroll
    for (int i = 0; i < kIt; ++i) { const int v = lane + 64 * i; rv[i] = v < kNV ? jr[v] : 0.f; }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < kIt; ++i) {
      const int v = lane + 64 * i;
      if (v < kNV) { a0 += rv[i] * sv[v]; a1 += rv[i] * sv[kNVP + v]; a2 += rv[i] * sv[2 * kNVP + v]; }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { j16[wave * 3] = a0; j16[wave * 3 + 1] = a1; j16[wave * 3 + 2] = a2; }
  }
  __syncthreads();
  if (tid < 63) {
    const int sl = tid / 3, c = tid % 3;
    const int src = c_xyz_src[sl];
    j21[tid] = (src >= 0) ? j16[src * 3 + c] : sv[c * kNVP + (-src - 1)];
  }
  __syncthreads();
  float r[3] = {0.f, 0.f, 0.f};
  if (root_id >= 0) { r[0] = j21[root_id * 3]; r[1] = j21[root_id * 3 + 1]; r[2] = j21[root_id * 3 + 2]; }
  if (tid < 63) joints_rel[(size_t)b * 63 + tid] = j21[tid] - r[tid % 3];
  if (tid < 3 && root_out) root_out[b * 3 + tid] = r[tid];
  // verts_cam = verts_rel + root_xyz: the `skin_meshes.offset_verts_(-pred_root); .offset_verts_(root_xyz)` in front of the renderer
  // (models_res_nimble.py:203-205), here instead of an elementwise launch of its own
  float off[3] = {0.f, 0.f, 0.f};
  if (root_xyz) { off[0] = root_xyz[b * 3]; off[1] = root_xyz[b * 3 + 1]; off[2] = root_xyz[b * 3 + 2]; }
  if (verts_rel || verts_cam) {
    for (int e = tid; e < kNV * 3; e += kJointsFwdThreads) {
      const int v = e / 3, c = e % 3;
      const float rel = sv[c * kNVP + v] - r[c];
      if (verts_rel) verts_rel[(size_t)b * kNV * 3 + e] = rel;
      if (verts_cam) verts_cam[(size_t)b * kNV * 3 + e] = rel + off[c];
    }
  }
}

__global__ __launch_bounds__(256) void mano_joints_bwd_kernel(ManoDev t, const float* __restrict__ gjoints_rel,
                                                             const float* __restrict__ gverts_rel,
                                                             const float* __restrict__ groot, int root_id,
                                                             float* __restrict__ gverts) {
  __shared__ float gj21[63];
  __shared__ float gj16[kNJ * 3];
  __shared__ float red[4 * 3];
  __shared__ float gr[3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // gradient wrt the root joint: groot - sum(gjoints_rel) - sum(gverts_rel)
  float cs[3] = {0.f, 0.f, 0.f};
  if (gverts_rel && root_id >= 0) {
    for (int v = tid; v < kNV; v += 256) {
      const float* p = gverts_rel + ((size_t)b * kNV + v) * 3;
      cs[0] += p[0]; cs[1] += p[1]; cs[2] += p[2];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float s = wave_sum(cs[c]);
    if (lane == 0) red[wave * 3 + c] = s;
  }
  if (tid < 63) gj21[tid] = gjoints_rel ? gjoints_rel[(size_t)b * 63 + tid] : 0.f;
  __syncthreads();
  if (tid < 3) {
    float a = groot ? groot[b * 3 + tid] : 0.f;
    if (root_id >= 0) {
      for (int w = 0; w < 4; ++w) a -= red[w * 3 + tid];
      for (int j = 0; j < 21; ++j) a -= gj21[j * 3 + tid];
    } else {
      a = 0.f;
    }
    gr[tid] = a;
  }
  __syncthreads();
  if (tid < 3 && root_id >= 0) gj21[root_id * 3 + tid] += gr[tid];
  __syncthreads();
  if (tid < kNJ * 3) {
    const int j = tid / 3, c = tid % 3;
    float a = 0.f;
    for (int sl = 0; sl < 21; ++sl)
      if (c_xyz_src[sl] == j) a += gj21[sl * 3 + c];
    gj16[tid] = a;
  }
  __syncthreads();
  for (int v = tid; v < kNV; v += 256) {
    float g[3] = {0.f, 0.f, 0.f};
    if (gverts_rel) {
      const float* p = gverts_rel + ((size_t)b * kNV + v) * 3;
      g[0] = p[0]; g[1] = p[1]; g[2] = p[2];
    }
    for (int j = 0; j < kNJ; ++j) {
      const float w = t.jreg[j * kNVP + v];
      g[0] += w * gj16[j * 3]; g[1] += w * gj16[j * 3 + 1]; g[2] += w * gj16[j * 3 + 2];
    }
    for (int sl = 4; sl < 21; sl += 4) {
      if (-c_xyz_src[sl] - 1 == v) { g[0] += gj21[sl * 3]; g[1] += gj21[sl * 3 + 1]; g[2] += gj21[sl * 3 + 2]; }
    }
    float* o = gverts + ((size_t)b * kNV + v) * 3;
    o[0] = g[0]; o[1] = g[1]; o[2] = g[2];
  }
}

// ------------------------------------------------------------------------------------------------
// launchers (called by the C ABI in hifihr_api.hip)
// ------------------------------------------------------------------------------------------------
hipError_t launch_mano_fwd(const ManoDev& t, const float* pose, const float* beta, int B, float* verts, float* jtr,
                           float* saved, hipStream_t st) {
  hipLaunchKernelGGL(mano_fwd_kernel, dim3(kFwdTiles, B), dim3(kFwdThreads), 0, st, t, pose, beta, verts, jtr, saved);
  return hipGetLastError();
}

// Forward of the fused form: TWO launches (layer, then joint regression + root-relative step + camera-space offset).  MEASURED dead end
// (round 5): the joint regression as the tail of mano_fwd_kernel -- the last of a hand's seven tile workgroups to arrive (device-scope
// arrival counter, release / acquire fences) reads the hand's vertices back and regresses the joints on its 128 threads -- ran 33.6 us
// against 17.3 + 5.4 us for the two launches (38.8 before its loads were taken off relaxed atomics): the fences and a 2-wave
// regression cost more than the launch they save.  The BACKWARD is one launch (mano_bwd_kernel's ManoBwdHead).
hipError_t launch_mano_full_fwd(const ManoDev& t, const float* pose, const float* beta, int B, int root_id, const float* root_xyz,
                                float* verts, float* joints_rel, float* verts_rel, float* verts_cam, float* root_out, float* saved,
                                hipStream_t st) {
  hipLaunchKernelGGL(mano_fwd_kernel, dim3(kFwdTiles, B), dim3(kFwdThreads), 0, st, t, pose, beta, verts, (float*)nullptr, saved);
  hipLaunchKernelGGL(mano_joints_fwd_kernel, dim3(B), dim3(kJointsFwdThreads), 0, st, t, (const float*)verts, root_id, joints_rel, verts_rel,
                     root_out, root_xyz, verts_cam);
  return hipGetLastError();
}

static hipError_t mano_bwd_attr() {
  if (sizeof(ManoBwdLds) <= 64 * 1024) return hipSuccess;
  static bool attr_set[16] = {};                                   // per DEVICE (the attribute is)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidValue;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mano_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ManoBwdLds));
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  return hipSuccess;
}

hipError_t launch_mano_bwd(const ManoDev& t, const float* pose, const float* beta, const float* saved,
                           const float* gverts, const float* gjtr, int B, float* gpose, float* gbeta, hipStream_t st) {
  if (hipError_t e = mano_bwd_attr()) return e;
  hipLaunchKernelGGL(mano_bwd_kernel, dim3(B), dim3(kBwdThreads), sizeof(ManoBwdLds), st, t, pose, beta, saved,
                     gverts, gjtr, gpose, gbeta, ManoBwdHead{});
  return hipGetLastError();
}

hipError_t launch_mano_full_bwd(const ManoDev& t, const float* pose, const float* beta, const float* saved, const float* gjoints_rel,
                                const float* gverts_rel, const float* gverts_cam, const float* groot, const float* gpose_add,
                                const float* gbeta_add, int B, int root_id, float* gpose, float* gbeta, hipStream_t st) {
  if (hipError_t e = mano_bwd_attr()) return e;
  ManoBwdHead head{1, root_id, gjoints_rel, gverts_rel, gverts_cam, groot, gpose_add, gbeta_add};
  hipLaunchKernelGGL(mano_bwd_kernel, dim3(B), dim3(kBwdThreads), sizeof(ManoBwdLds), st, t, pose, beta, saved,
                     nullptr, nullptr, gpose, gbeta, head);
  return hipGetLastError();
}

hipError_t launch_mano_joints_fwd(const ManoDev& t, const float* verts, int B, int root_id, float* joints_rel,
                                  float* verts_rel, float* root, hipStream_t st) {
  hipLaunchKernelGGL(mano_joints_fwd_kernel, dim3(B), dim3(kJointsFwdThreads), 0, st, t, verts, root_id, joints_rel, verts_rel, root,
                     (const float*)nullptr, (float*)nullptr);
  return hipGetLastError();
}

hipError_t launch_mano_joints_bwd(const ManoDev& t, const float* gjoints_rel, const float* gverts_rel,
                                  const float* groot, int B, int root_id, float* gverts, hipStream_t st) {
  hipLaunchKernelGGL(mano_joints_bwd_kernel, dim3(B), dim3(256), 0, st, t, gjoints_rel, gverts_rel, groot, root_id, gverts);
  return hipGetLastError();
}

}  // namespace hifihr
