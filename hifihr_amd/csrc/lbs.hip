// Generic linear-blend skinning for gfx950 (any vertex / joint / shape-component count, any kinematic tree, sparse skin weights):
// the NIMBLE-shaped hand layer.  The reference's `hand_model: "nimble"` configs call MyNIMBLELayer (un-vendored submodule, SURVEY.md
// section 8 A9: source and assets absent), whose skin mesh has 5 990 vertices driven by 25 joints from 20 shape / 30 pose components
// (reference models_res_nimble.py:56-57,133-142).  This file provides the kernels for a layer of that SHAPE on caller-supplied tables
// (hifihr_amd/nimble_tables.py builds seeded synthetic ones): the MANO formulation (reference utils/my_mano.py:386-451) without
// pose-corrective blend shapes, generalised --
//   v_shaped = template + shapedirs . beta;   J = jt + jsd . beta   (joint regressor folded into the tables)
//   R_j = Rodrigues(theta_j);  G_j = G_parent(j) [R_j | J_j - J_parent];  A_j = [Rg_j | tg_j - Rg_j J_j]
//   v = sum_k w_k A_{idx_k} [v_shaped; 1]   (K <= 8 non-zero weights per vertex);   posed joints = tg_j.
//   lbs_fwd_kernel      grid (ceil(V / 256), B): per-hand prologue in LDS (every workgroup redoes it: J Rodrigues + a J-step chain),
//                       one vertex per lane; tables are structure-of-arrays over the vertex index (coalesced 256-byte rows).
//   lbs_bwd_vert_kernel same grid: d(v_shaped) -> dbeta (wave dot products + one atomic per (workgroup, component)); d(A_j) = a
//                       (joints x vertices) . (vertices x 12) product summed on the matrix cores from a dense LDS image of the
//                       workgroup's skin weights, then one global atomic per (workgroup, joint, entry).
//   lbs_bwd_chain_kernel  one wave per hand: d(A), d(posed joints) -> reverse kinematic chain -> d(theta) (Rodrigues reverse mode),
//                       d(J) -> dbeta.
// HBM-bound / latency-bound like the MANO kernels: per hand 12 V bytes out + the tables once per launch (L2-resident).
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"
#include "mano_math.h"

namespace hifihr {

#if defined(HIFIHR_HOSTSIM)
typedef hs_floatx4 hs_or_native_floatx4;
#else
typedef float hs_or_native_floatx4 __attribute__((ext_vector_type(4)));
#endif

struct LbsSmall {
  float beta[kLbsMaxS];
  float theta[kLbsMaxJ * 3];
  float J[kLbsMaxJ * 3];
  float Rl[kLbsMaxJ * 9];
  float Rg[kLbsMaxJ * 9];
  float tg[kLbsMaxJ * 3];
  float Ap[kLbsMaxJ * 12];
  int parent[kLbsMaxJ];
  int depth[kLbsMaxJ];       // tree depth of a joint (root: 0)
  int maxdepth;
};

// needs >= 128 threads; ends with a barrier
__device__ __forceinline__ void lbs_prologue(const LbsDev& t, const float* __restrict__ theta, const float* __restrict__ beta, int b, LbsSmall& s) {
  const int tid = threadIdx.x;
  if (tid < t.S) s.beta[tid] = beta[(size_t)b * t.S + tid];
  if (tid >= 32 && tid < 32 + t.J) s.parent[tid - 32] = t.parent[tid - 32];
  for (int e = tid; e < t.J * 3; e += blockDim.x) s.theta[e] = theta[(size_t)b * t.J * 3 + e];
  __syncthreads();
  for (int e = tid; e < t.J * 3; e += blockDim.x) {
    float acc = t.jt[e];
    for (int k = 0; k < t.S; ++k) acc += t.jsd[e * t.S + k] * s.beta[k];
    s.J[e] = acc;
  }
  if (tid < t.J) rodrigues_fwd(s.theta + 3 * tid, s.Rl + 9 * tid, nullptr);
  if (tid == 0) s.maxdepth = 0;
  int dep = 0;
  if (tid >= 64 && tid < 64 + t.J) {                 // depth of joint tid - 64: a walk up the (LDS-resident) parent array
    for (int p = s.parent[tid - 64]; p >= 0; p = s.parent[p]) ++dep;
    s.depth[tid - 64] = dep;
  }
  __syncthreads();
  if (tid >= 64 && tid < 64 + t.J) atomicMax(&s.maxdepth, dep);
  __syncthreads();
  // the kinematic chain level by level: the joints of one tree depth in parallel (a hand is 5-6 levels deep; one lane walking all
  // joints paid ~30 dependent LDS round trips per joint)
  const int nlev = s.maxdepth;
  for (int l = 0; l <= nlev; ++l) {
    if (tid < t.J && s.depth[tid] == l) {
      const int i = tid;
      if (l == 0) {
        for (int k = 0; k < 9; ++k) s.Rg[9 * i + k] = s.Rl[9 * i + k];
        for (int k = 0; k < 3; ++k) s.tg[3 * i + k] = s.J[3 * i + k];
      } else {
        const int p = s.parent[i];
        mat3_mul(s.Rg + 9 * p, s.Rl + 9 * i, s.Rg + 9 * i);
        const float d[3] = {s.J[3 * i] - s.J[3 * p], s.J[3 * i + 1] - s.J[3 * p + 1], s.J[3 * i + 2] - s.J[3 * p + 2]};
        float r[3];
        mat3_vec(s.Rg + 9 * p, d, r);
        for (int k = 0; k < 3; ++k) s.tg[3 * i + k] = r[k] + s.tg[3 * p + k];
      }
    }
    __syncthreads();
  }
  if (tid < t.J) chain_make_ap(tid, s.J, s.Rg, s.tg, s.Ap);
  __syncthreads();
}

__global__ __launch_bounds__(256) void lbs_fwd_kernel(LbsDev t, const float* __restrict__ theta, const float* __restrict__ beta,
                                                     float* __restrict__ verts, float* __restrict__ joints) {
  __shared__ LbsSmall s;
  const int b = blockIdx.y, tid = threadIdx.x;
  lbs_prologue(t, theta, beta, b, s);
  if (blockIdx.x == 0 && joints != nullptr)
    for (int e = tid; e < t.J * 3; e += 256) joints[(size_t)b * t.J * 3 + e] = s.tg[e];
  const int v = blockIdx.x * 256 + tid;
  if (v >= t.V) return;
  float vs[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) vs[c] = t.tmpl[c * t.Vp + v];
  for (int k = 0; k < t.S; ++k) {
    const float bk = s.beta[k];
    const float* row = t.sd + (size_t)k * 3 * t.Vp + v;
    vs[0] += row[0] * bk; vs[1] += row[t.Vp] * bk; vs[2] += row[2 * t.Vp] * bk;
  }
  float o[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < t.K; ++k) {
    const float w = t.wval[k * t.Vp + v];
    const float* A = s.Ap + 12 * t.widx[k * t.Vp + v];
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] += w * (A[4 * r] * vs[0] + A[4 * r + 1] * vs[1] + A[4 * r + 2] * vs[2] + A[4 * r + 3]);
  }
  float* out = verts + ((size_t)b * t.V + v) * 3;
  out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
}

// gA[B][J][12], gbeta[B][S]: ZERO on entry, accumulated with float atomics
// d(A_j)[r][c] = sum over the vertices of w_vj g_v[r] [vs_v; 1][c] is a (joints x vertices) . (vertices x 12) product with a sparse left
// factor.  Summed with LDS float atomics it serialised on the joints a patch of skin shares (131 us at V = 5 990, B = 32); here the
// workgroup spreads its 256 vertices' weights into a dense [256][32] LDS image (a vertex's row is private to its lane: no atomics),
// writes the 12 outer-product entries per vertex next to it, and the matrix cores do the sum: 2 joint tiles x 16 k-steps of
// v_mfma_f32_16x16x4_f32 per wave, the four waves' tiles added through LDS in a fixed order -- deterministic inside the workgroup.
__global__ __launch_bounds__(256) void lbs_bwd_vert_kernel(LbsDev t, const float* __restrict__ theta, const float* __restrict__ beta,
                                                          const float* __restrict__ gverts, float* __restrict__ gA, float* __restrict__ gbeta) {
  __shared__ LbsSmall s;
  __shared__ __attribute__((aligned(16))) float Wd[256][kLbsMaxJ];      // dense skin weights of this workgroup's vertices (32 KB); later the waves' partial tiles
  __shared__ __attribute__((aligned(16))) float X[256][16];             // g (x) [vs; 1], 12 entries + 4 zeros per vertex (16 KB)
  __shared__ float accB[4][kLbsMaxS];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  lbs_prologue(t, theta, beta, b, s);
  const int v = blockIdx.x * 256 + tid;
  const bool ok = v < t.V;
  float vs[3] = {0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f}, gvs[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < kLbsMaxJ / 4; ++q) *reinterpret_cast<float4*>(&Wd[tid][4 * q]) = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
#pragma unroll
    for (int c = 0; c < 3; ++c) vs[c] = t.tmpl[c * t.Vp + v];
    for (int k = 0; k < t.S; ++k) {
      const float bk = s.beta[k];
      const float* row = t.sd + (size_t)k * 3 * t.Vp + v;
      vs[0] += row[0] * bk; vs[1] += row[t.Vp] * bk; vs[2] += row[2 * t.Vp] * bk;
    }
    const float* gp = gverts + ((size_t)b * t.V + v) * 3;
    g[0] = gp[0]; g[1] = gp[1]; g[2] = gp[2];
    for (int k = 0; k < t.K; ++k) {
      const float w = t.wval[k * t.Vp + v];
      if (w == 0.f) continue;
      const int j = t.widx[k * t.Vp + v];
      Wd[tid][j] = w;                                        // (a joint appears once per vertex: the tables are built from a dense row)
      const float* A = s.Ap + 12 * j;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float wg = w * g[r];
        gvs[0] += wg * A[4 * r]; gvs[1] += wg * A[4 * r + 1]; gvs[2] += wg * A[4 * r + 2];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) *reinterpret_cast<float4*>(&X[tid][4 * r]) = make_float4(g[r] * vs[0], g[r] * vs[1], g[r] * vs[2], g[r]);
  *reinterpret_cast<float4*>(&X[tid][12]) = make_float4(0.f, 0.f, 0.f, 0.f);
  // dbeta[k] = sum_v gvs . shapedirs[k][:, v]
  for (int k = 0; k < t.S; ++k) {
    float p = 0.f;
    if (ok) {
      const float* row = t.sd + (size_t)k * 3 * t.Vp + v;
      p = gvs[0] * row[0] + gvs[1] * row[t.Vp] + gvs[2] * row[2 * t.Vp];
    }
    for (int o = 32; o > 0; o >>= 1) p += __shfl_down(p, o, 64);
    if (lane == 0) accB[wave][k] = p;
  }
  __syncthreads();
  // out[j][e] = sum_v Wd[v][j] X[v][e]: wave w takes vertices 64 w .. 64 w + 63 (16 k-steps of 4), both joint tiles
  {
    const int r = lane & 15, gq = lane >> 4;
    hs_or_native_floatx4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int st = 0; st < 16; ++st) {
      const int vv = 64 * wave + 4 * st + gq;
      const float xb = X[vv][r];
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Wd[vv][r], xb, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Wd[vv][16 + r], xb, d1, 0, 0, 0);
    }
    __syncthreads();                                         // every wave has read its rows of Wd: reuse it for the partial tiles
    float* red = &Wd[0][0] + wave * 512;                     // [2 tiles][16 joints][16 entries]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[(4 * gq + e) * 16 + r] = d0[e];
      red[256 + (4 * gq + e) * 16 + r] = d1[e];
    }
  }
  __syncthreads();
  for (int i = tid; i < t.J * 12; i += 256) {
    const int j = i / 12, e = i - 12 * j;
    const float* red = &Wd[0][0] + j * 16 + e;               // tile j / 16 starts at 256 (j / 16): j * 16 covers both
    const float x = red[0] + red[512] + red[1024] + red[1536];
    if (x != 0.f) atomicAdd(gA + (size_t)b * t.J * 12 + i, x);
  }
  if (tid < t.S) atomicAdd(gbeta + (size_t)b * t.S + tid, accB[0][tid] + accB[1][tid] + accB[2][tid] + accB[3][tid]);
}

// one wave per hand: gA (grad of [Rg | t']), gjoints (grad of the posed joints tg; may be null) -> gtheta[B][J][3] (overwritten), gbeta += (J part)
__global__ __launch_bounds__(128) void lbs_bwd_chain_kernel(LbsDev t, const float* __restrict__ theta, const float* __restrict__ beta,
                                                           const float* __restrict__ gA, const float* __restrict__ gjoints,
                                                           float* __restrict__ gtheta, float* __restrict__ gbeta) {
  __shared__ LbsSmall s;
  __shared__ float gRg[kLbsMaxJ * 9], gtg[kLbsMaxJ * 3], gJ[kLbsMaxJ * 3], gRl[kLbsMaxJ * 9], gAp[kLbsMaxJ * 12];
  const int b = blockIdx.x, tid = threadIdx.x;
  lbs_prologue(t, theta, beta, b, s);
  for (int e = tid; e < t.J * 9; e += 128) gRg[e] = 0.f;
  for (int e = tid; e < t.J * 3; e += 128) { gtg[e] = gjoints ? gjoints[(size_t)b * t.J * 3 + e] : 0.f; gJ[e] = 0.f; }
  for (int e = tid; e < t.J * 12; e += 128) gAp[e] = gA[(size_t)b * t.J * 12 + e];
  __syncthreads();
  if (tid < t.J) chain_make_ap_bwd(tid, s.J, s.Rg, gAp, gRg, gtg, gJ);     // joint-local: no two lanes touch the same entries
  __syncthreads();
  // reverse chain, deepest level first: the PARENTS of level l gather from their children (a child's gradients are complete once the
  // level below has run; gathering instead of scattering needs no atomics and keeps the sums in a fixed order)
  for (int l = s.maxdepth - 1; l >= 0; --l) {
    if (tid < t.J && s.depth[tid] == l) {
      const int p = tid;
      for (int i = p + 1; i < t.J; ++i) {
        if (s.parent[i] != p) continue;
        float tmp[9];
        mat3_mul_nt(gRg + 9 * i, s.Rl + 9 * i, tmp);             // Rg_i = Rg_p Rl_i
        for (int k = 0; k < 9; ++k) gRg[9 * p + k] += tmp[k];
        mat3_mul_tn(s.Rg + 9 * p, gRg + 9 * i, gRl + 9 * i);
        const float d[3] = {s.J[3 * i] - s.J[3 * p], s.J[3 * i + 1] - s.J[3 * p + 1], s.J[3 * i + 2] - s.J[3 * p + 2]};
        const float* gt = gtg + 3 * i;                            // tg_i = Rg_p (J_i - J_p) + tg_p
        for (int r = 0; r < 3; ++r)
          for (int k = 0; k < 3; ++k) gRg[9 * p + 3 * r + k] += gt[r] * d[k];
        float rt[3];
        mat3t_vec(s.Rg + 9 * p, gt, rt);
        for (int k = 0; k < 3; ++k) { gJ[3 * i + k] += rt[k]; gJ[3 * p + k] -= rt[k]; gtg[3 * p + k] += gt[k]; }
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int k = 0; k < 9; ++k) gRl[k] = gRg[k];                // root: Rg_0 = Rl_0, tg_0 = J_0
    for (int k = 0; k < 3; ++k) gJ[k] += gtg[k];
  }
  __syncthreads();
  if (tid < t.J) {
    float gw[3];
    rodrigues_bwd(s.theta + 3 * tid, gRl + 9 * tid, gw);
    for (int k = 0; k < 3; ++k) gtheta[((size_t)b * t.J + tid) * 3 + k] = gw[k];
  }
  if (tid >= 64 && tid < 64 + t.S) {                            // J = jt + jsd beta
    const int k = tid - 64;
    float acc = 0.f;
    for (int e = 0; e < t.J * 3; ++e) acc += gJ[e] * t.jsd[e * t.S + k];
    atomicAdd(gbeta + (size_t)b * t.S + k, acc);
  }
}

static bool lbs_ok(const LbsDev& t) {
  return t.V > 0 && t.Vp >= t.V && t.J >= 1 && t.J <= kLbsMaxJ && t.S >= 0 && t.S <= kLbsMaxS && t.K >= 1 && t.K <= 8;
}

hipError_t launch_lbs_fwd(const LbsDev& t, const float* theta, const float* beta, int B, float* verts, float* joints, hipStream_t st) {
  if (!lbs_ok(t) || B < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(lbs_fwd_kernel, dim3((t.V + 255) / 256, B), dim3(256), 0, st, t, theta, beta, verts, joints);
  return hipGetLastError();
}

hipError_t launch_lbs_bwd(const LbsDev& t, const float* theta, const float* beta, const float* gverts, const float* gjoints, int B,
                          float* gA_zeroed, float* gtheta, float* gbeta_zeroed, hipStream_t st) {
  if (!lbs_ok(t) || B < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(lbs_bwd_vert_kernel, dim3((t.V + 255) / 256, B), dim3(256), 0, st, t, theta, beta, gverts, gA_zeroed, gbeta_zeroed);
  hipLaunchKernelGGL(lbs_bwd_chain_kernel, dim3(B), dim3(128), 0, st, t, theta, beta, gA_zeroed, gjoints, gtheta, gbeta_zeroed);
  return hipGetLastError();
}

}  // namespace hifihr
