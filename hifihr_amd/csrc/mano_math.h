// Scalar math of the MANO layer shared by the HIP kernels (device) and by tests/hostsim (host, to check
// the hand-derived gradients without a GPU).  Not an oracle and not a CPU fallback: the package never
// compiles this for the host.
//
// Follows reference utils/manopth/rodrigues_layer.py:15-54 (batch_rodrigues -> quat2mat) and the
// kinematic chain of reference utils/my_mano.py:396-439.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define HIFIHR_HD __host__ __device__ __forceinline__
#else
#define HIFIHR_HD inline
#endif

namespace hifihr {

constexpr int kNV = 778;
constexpr int kNJ = 16;
constexpr int kNB = 10;
constexpr int kNP = 135;
constexpr int kNPCA = 45;

// parent of joint i in MANO order (kintree_table row 0)
HIFIHR_HD int mano_parent(int i) { return (i == 0) ? -1 : (((i - 1) % 3 == 0) ? 0 : i - 1); }

// ---- 3x3 helpers (row-major) -------------------------------------------------------------------
HIFIHR_HD void mat3_mul(const float* a, const float* b, float* c) {  // c = a b
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) c[r * 3 + k] = a[r * 3 + 0] * b[0 * 3 + k] + a[r * 3 + 1] * b[1 * 3 + k] + a[r * 3 + 2] * b[2 * 3 + k];
}
HIFIHR_HD void mat3_mul_tn(const float* a, const float* b, float* c) {  // c = a^T b
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) c[r * 3 + k] = a[0 * 3 + r] * b[0 * 3 + k] + a[1 * 3 + r] * b[1 * 3 + k] + a[2 * 3 + r] * b[2 * 3 + k];
}
HIFIHR_HD void mat3_mul_nt(const float* a, const float* b, float* c) {  // c = a b^T
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) c[r * 3 + k] = a[r * 3 + 0] * b[k * 3 + 0] + a[r * 3 + 1] * b[k * 3 + 1] + a[r * 3 + 2] * b[k * 3 + 2];
}
HIFIHR_HD void mat3_vec(const float* a, const float* v, float* o) {  // o = a v
  for (int r = 0; r < 3; ++r) o[r] = a[r * 3 + 0] * v[0] + a[r * 3 + 1] * v[1] + a[r * 3 + 2] * v[2];
}
HIFIHR_HD void mat3t_vec(const float* a, const float* v, float* o) {  // o = a^T v
  for (int r = 0; r < 3; ++r) o[r] = a[0 * 3 + r] * v[0] + a[1 * 3 + r] * v[1] + a[2 * 3 + r] * v[2];
}

// ---- batch_rodrigues (rodrigues_layer.py:43-54) + quat2mat (:15-40), one rotation ----------------
struct RodriguesTmp {
  float n, u[3], s, c, m, p[4];
};

HIFIHR_HD void rodrigues_fwd(const float* w, float* R, RodriguesTmp* t) {
  const float a0 = w[0] + 1e-8f, a1 = w[1] + 1e-8f, a2 = w[2] + 1e-8f;   // norm(axisang + 1e-8)
  const float n = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
  const float u0 = w[0] / n, u1 = w[1] / n, u2 = w[2] / n;
  const float h = n * 0.5f;
  const float c = cosf(h), s = sinf(h);
  const float q0 = c, q1 = s * u0, q2 = s * u1, q3 = s * u2;
  const float m = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  const float W = q0 / m, X = q1 / m, Y = q2 / m, Z = q3 / m;
  const float w2 = W * W, x2 = X * X, y2 = Y * Y, z2 = Z * Z;
  const float wx = W * X, wy = W * Y, wz = W * Z, xy = X * Y, xz = X * Z, yz = Y * Z;
  R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
  R[3] = 2 * wz + 2 * xy;   R[4] = w2 - x2 + y2 - z2;   R[5] = 2 * yz - 2 * wx;
  R[6] = 2 * xz - 2 * wy;   R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
  if (t) {
    t->n = n; t->u[0] = u0; t->u[1] = u1; t->u[2] = u2; t->s = s; t->c = c; t->m = m;
    t->p[0] = W; t->p[1] = X; t->p[2] = Y; t->p[3] = Z;
  }
}

// reverse mode of rodrigues_fwd: gR[9] -> gw[3] (overwritten)
HIFIHR_HD void rodrigues_bwd(const float* w, const float* g, float* gw) {
  float R[9];
  RodriguesTmp t;
  rodrigues_fwd(w, R, &t);
  const float W = t.p[0], X = t.p[1], Y = t.p[2], Z = t.p[3];
  float gp[4];
  gp[0] = 2 * W * (g[0] + g[4] + g[8]) + 2 * (-Z * g[1] + Y * g[2] + Z * g[3] - X * g[5] - Y * g[6] + X * g[7]);
  gp[1] = 2 * X * (g[0] - g[4] - g[8]) + 2 * (Y * g[1] + Z * g[2] + Y * g[3] - W * g[5] + Z * g[6] + W * g[7]);
  gp[2] = 2 * Y * (-g[0] + g[4] - g[8]) + 2 * (X * g[1] + W * g[2] + X * g[3] + Z * g[5] - W * g[6] + Z * g[7]);
  gp[3] = 2 * Z * (-g[0] - g[4] + g[8]) + 2 * (-W * g[1] + X * g[2] + W * g[3] + Y * g[5] + X * g[6] + Y * g[7]);
  // p = q / m
  const float pd = t.p[0] * gp[0] + t.p[1] * gp[1] + t.p[2] * gp[2] + t.p[3] * gp[3];
  float gq[4];
  for (int k = 0; k < 4; ++k) gq[k] = (gp[k] - t.p[k] * pd) / t.m;
  const float gc = gq[0];
  const float gs = gq[1] * t.u[0] + gq[2] * t.u[1] + gq[3] * t.u[2];
  const float gu[3] = {t.s * gq[1], t.s * gq[2], t.s * gq[3]};
  const float gh = -t.s * gc + t.c * gs;
  float gn = 0.5f * gh;
  const float inv_n = 1.0f / t.n;
  gn -= (gu[0] * w[0] + gu[1] * w[1] + gu[2] * w[2]) * inv_n * inv_n;
  for (int k = 0; k < 3; ++k) gw[k] = gu[k] * inv_n + gn * (w[k] + 1e-8f) * inv_n;
}

// ---- kinematic chain (my_mano.py:396-439), joint order = MANO order 0..15 ------------------------
// Rl[16][9] local rotations, J[16][3] rest joints  ->  Rg[16][9], tg[16][3] global transforms and
// Ap[16][12] = [Rg | tg - Rg J]  (the "results2" of my_mano.py:437-439), row-major 3x4.
// Forward for one finger f (0..4): joints 1+3f, 2+3f, 3+3f.  Root (joint 0) must already be in Rg/tg.
HIFIHR_HD void chain_fwd_finger(int f, const float* Rl, const float* J, float* Rg, float* tg) {
  for (int l = 0; l < 3; ++l) {
    const int i = 1 + 3 * f + l;
    const int p = (l == 0) ? 0 : i - 1;
    mat3_mul(Rg + 9 * p, Rl + 9 * i, Rg + 9 * i);
    const float d[3] = {J[3 * i] - J[3 * p], J[3 * i + 1] - J[3 * p + 1], J[3 * i + 2] - J[3 * p + 2]};
    float r[3];
    mat3_vec(Rg + 9 * p, d, r);
    for (int k = 0; k < 3; ++k) tg[3 * i + k] = r[k] + tg[3 * p + k];
  }
}
HIFIHR_HD void chain_fwd_root(const float* Rl, const float* J, float* Rg, float* tg) {
  for (int k = 0; k < 9; ++k) Rg[k] = Rl[k];
  for (int k = 0; k < 3; ++k) tg[k] = J[k];
}
HIFIHR_HD void chain_make_ap(int i, const float* J, const float* Rg, const float* tg, float* Ap) {
  float rj[3];
  mat3_vec(Rg + 9 * i, J + 3 * i, rj);
  for (int r = 0; r < 3; ++r) {
    Ap[12 * i + 4 * r + 0] = Rg[9 * i + 3 * r + 0];
    Ap[12 * i + 4 * r + 1] = Rg[9 * i + 3 * r + 1];
    Ap[12 * i + 4 * r + 2] = Rg[9 * i + 3 * r + 2];
    Ap[12 * i + 4 * r + 3] = tg[3 * i + r] - rj[r];
  }
}

// Reverse of chain_make_ap for joint i: gAp[12] (grad of [Rg | t']) is folded into gRg[9], gtg[3], gJ[3]
// (all accumulated).
HIFIHR_HD void chain_make_ap_bwd(int i, const float* J, const float* Rg, const float* gAp, float* gRg, float* gtg, float* gJ) {
  float gt[3];
  for (int r = 0; r < 3; ++r) {
    gt[r] = gAp[12 * i + 4 * r + 3];
    gRg[9 * i + 3 * r + 0] += gAp[12 * i + 4 * r + 0] - gt[r] * J[3 * i + 0];
    gRg[9 * i + 3 * r + 1] += gAp[12 * i + 4 * r + 1] - gt[r] * J[3 * i + 1];
    gRg[9 * i + 3 * r + 2] += gAp[12 * i + 4 * r + 2] - gt[r] * J[3 * i + 2];
    gtg[3 * i + r] += gt[r];
  }
  float rt[3];
  mat3t_vec(Rg + 9 * i, gt, rt);
  for (int k = 0; k < 3; ++k) gJ[3 * i + k] -= rt[k];
}

// Reverse of chain_fwd_finger: consumes gRg/gtg of the finger's joints (tip to base), produces gRl for
// them, accumulates gJ, and accumulates the root's share into gRg0_acc[9], gtg0_acc[3], gJ0_acc[3]
// (per-finger accumulators so that five fingers can run on five threads; the caller sums them).
HIFIHR_HD void chain_bwd_finger(int f, const float* Rl, const float* J, const float* Rg, float* gRg, float* gtg,
                                float* gRl, float* gJ, float* gRg0_acc, float* gtg0_acc, float* gJ0_acc) {
  for (int l = 2; l >= 0; --l) {
    const int i = 1 + 3 * f + l;
    const int p = (l == 0) ? 0 : i - 1;
    float* gRp = (l == 0) ? gRg0_acc : gRg + 9 * p;
    float* gtp = (l == 0) ? gtg0_acc : gtg + 3 * p;
    float* gJp = (l == 0) ? gJ0_acc : gJ + 3 * p;
    // Rg_i = Rg_p Rl_i
    float tmp[9];
    mat3_mul_nt(gRg + 9 * i, Rl + 9 * i, tmp);          // gRg_p += gRg_i Rl_i^T
    for (int k = 0; k < 9; ++k) gRp[k] += tmp[k];
    mat3_mul_tn(Rg + 9 * p, gRg + 9 * i, gRl + 9 * i);  // gRl_i = Rg_p^T gRg_i
    // tg_i = Rg_p (J_i - J_p) + tg_p
    const float d[3] = {J[3 * i] - J[3 * p], J[3 * i + 1] - J[3 * p + 1], J[3 * i + 2] - J[3 * p + 2]};
    const float* gt = gtg + 3 * i;
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) gRp[3 * r + k] += gt[r] * d[k];
    float rt[3];
    mat3t_vec(Rg + 9 * p, gt, rt);
    for (int k = 0; k < 3; ++k) {
      gJ[3 * i + k] += rt[k];
      gJp[k] -= rt[k];
      gtp[k] += gt[k];
    }
  }
}

}  // namespace hifihr
