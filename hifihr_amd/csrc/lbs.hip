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
//   lbs_bwd_vert_kernel same grid: d(v_shaped) -> dbeta (wave dot products + one atomic per (workgroup, component)); d(A_j) summed in
//                       LDS per wave, then one global atomic per (workgroup, joint, entry).
//   lbs_bwd_chain_kernel  one wave per hand: d(A), d(posed joints) -> reverse kinematic chain -> d(theta) (Rodrigues reverse mode),
//                       d(J) -> dbeta.
// HBM-bound / latency-bound like the MANO kernels: per hand 12 V bytes out + the tables once per launch (L2-resident).
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"
#include "mano_math.h"

namespace hifihr {

struct LbsSmall {
  float beta[kLbsMaxS];
  float theta[kLbsMaxJ * 3];
  float J[kLbsMaxJ * 3];
  float Rl[kLbsMaxJ * 9];
  float Rg[kLbsMaxJ * 9];
  float tg[kLbsMaxJ * 3];
  float Ap[kLbsMaxJ * 12];
  int parent[kLbsMaxJ];
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
  __syncthreads();
  if (tid == 0) {                                  // the chain is a dependent walk over <= 32 joints: one lane
    for (int k = 0; k < 9; ++k) s.Rg[k] = s.Rl[k];
    for (int k = 0; k < 3; ++k) s.tg[k] = s.J[k];
    for (int i = 1; i < t.J; ++i) {
      const int p = s.parent[i];
      mat3_mul(s.Rg + 9 * p, s.Rl + 9 * i, s.Rg + 9 * i);
      const float d[3] = {s.J[3 * i] - s.J[3 * p], s.J[3 * i + 1] - s.J[3 * p + 1], s.J[3 * i + 2] - s.J[3 * p + 2]};
      float r[3];
      mat3_vec(s.Rg + 9 * p, d, r);
      for (int k = 0; k < 3; ++k) s.tg[3 * i + k] = r[k] + s.tg[3 * p + k];
    }
  }
  __syncthreads();
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
__global__ __launch_bounds__(256) void lbs_bwd_vert_kernel(LbsDev t, const float* __restrict__ theta, const float* __restrict__ beta,
                                                          const float* __restrict__ gverts, float* __restrict__ gA, float* __restrict__ gbeta) {
  __shared__ LbsSmall s;
  __shared__ float accA[4][kLbsMaxJ * 12];
  __shared__ float accB[4][kLbsMaxS];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  lbs_prologue(t, theta, beta, b, s);
  for (int e = tid; e < 4 * kLbsMaxJ * 12; e += 256) (&accA[0][0])[e] = 0.f;
  __syncthreads();
  const int v = blockIdx.x * 256 + tid;
  const bool ok = v < t.V;
  float vs[3] = {0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f}, gvs[3] = {0.f, 0.f, 0.f};
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
      const float* A = s.Ap + 12 * j;
      float* a = accA[wave] + 12 * j;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float wg = w * g[r];
        atomicAdd(a + 4 * r, wg * vs[0]); atomicAdd(a + 4 * r + 1, wg * vs[1]); atomicAdd(a + 4 * r + 2, wg * vs[2]); atomicAdd(a + 4 * r + 3, wg);
        gvs[0] += wg * A[4 * r]; gvs[1] += wg * A[4 * r + 1]; gvs[2] += wg * A[4 * r + 2];
      }
    }
  }
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
  for (int e = tid; e < t.J * 12; e += 256) {
    const float x = accA[0][e] + accA[1][e] + accA[2][e] + accA[3][e];
    if (x != 0.f) atomicAdd(gA + (size_t)b * t.J * 12 + e, x);
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
  if (tid == 0) {
    for (int i = t.J - 1; i >= 1; --i) {
      const int p = s.parent[i];
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
