// Scalar math of the hard rasteriser + Phong shader, shared by the HIP kernels (device) and tests/hostsim.
// Not an oracle, not a CPU fallback (see mano_math.h).
//
// Semantics restated from PyTorch3D as the reference configures it (reference models_res_nimble.py:70-96,
// 184-190, 208): MeshRasterizer(image_size=224*3, blur_radius=0, faces_per_pixel=1, perspective-correct
// barycentrics, no culling / clipping) followed by HardPhongShader(DirectionalLights, Materials) and
// hard_rgb_blend with a white background.  PyTorch3D is not vendored by the reference, so these semantics
// are [recalled] and parity at that boundary is unpinned; the rounding-
// sensitive part (coverage, depth, tie rule) is written operation-for-operation like oracle/raster_oracle.c
// and both are compiled with -ffp-contract=off so that face indices are bit-identical.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define HIFIHR_HD __host__ __device__ __forceinline__
#else
#ifndef HIFIHR_HD
#define HIFIHR_HD inline
#endif
#endif

namespace hifihr {

constexpr float kRasterEps = 1e-8f;   // PyTorch3D kEpsilon
constexpr float kNormEps = 1e-6f;     // F.normalize eps used by lighting and vertex normals

// NDC coordinate of sample index i of an S-wide square grid, PyTorch3D PixToNonSquareNdc (square case).
// Index 0 is NDC +1 after the flip done by the caller (i = S - 1 - pixel).
HIFIHR_HD float pix_to_ndc(int i, int S) {
  const float range = 2.0f;
  const float offset = range / 2.0f;
  return -offset + (range * (float)i + offset) / (float)S;
}

// EdgeFunctionForward(p, v0, v1)
HIFIHR_HD float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

struct FaceXYZ {
  float x0, y0, x1, y1, x2, y2, z0, z1, z2;
};

// Face-level rejection (independent of the sample): degenerate area, or the WHOLE face at / behind the image plane
// (faces are kept when zmax >= 1e-8; a face that merely straddles the plane is NOT dropped here -- its
// samples go through the per-sample test, which rejects them by pz < 0 when one vertex is behind the camera).  Round 1 dropped
// faces with zmin < 1e-8; the two rules differ only for geometry with vertices behind the camera (never a hand at 0.3-0.8 m),
// tests/golden/raster_known.json case "two_vertices_behind_camera" pins the chosen one.
HIFIHR_HD bool face_is_rejected(const FaceXYZ& f) {
  const float face_area = edge_fn(f.x0, f.y0, f.x1, f.y1, f.x2, f.y2);
  const bool zero_area = (face_area <= kRasterEps) && (face_area >= -kRasterEps);
  const float zmax = fmaxf(f.z0, fmaxf(f.z1, f.z2));
  return zero_area || (zmax < kRasterEps);
}

// One sample against one face.  Returns true when the sample is covered; then bary[3] are the
// perspective-corrected barycentrics and *pz the interpolated depth.  Exactly the sequence of
// oracle/raster_oracle.c (bbox test, divided and perspective-corrected barycentrics, pz >= 0, strict b > 0).
HIFIHR_HD bool sample_face(const FaceXYZ& f, float xmin, float xmax, float ymin, float ymax, float px, float py,
                           float* bary, float* pz) {
  if (px > xmax || px < xmin || py > ymax || py < ymin) return false;
  const float area = edge_fn(f.x2, f.y2, f.x0, f.y0, f.x1, f.y1) + kRasterEps;
  const float e0 = edge_fn(px, py, f.x1, f.y1, f.x2, f.y2);
  const float e1 = edge_fn(px, py, f.x2, f.y2, f.x0, f.y0);
  const float e2 = edge_fn(px, py, f.x0, f.y0, f.x1, f.y1);
  // The inside test is on the perspective-CORRECTED barycentrics b_i = w_i z_j z_k / denom (PyTorch3D CheckPixelInsideFace [recalled];
  // round 3).  For a face wholly in front of the camera (z0, z1, z2 > 0, denom > 0) b_i > 0 needs w_i = e_i / area > 0, which is
  // impossible when e_i == 0 or sign(e_i) != sign(area): a cheap exact early-out before the six divisions.  With a vertex at or behind
  // the camera plane the signs decouple (a sample OUTSIDE the 2-D triangle can have all b_i > 0) and every sample of the bounding box
  // goes through the full arithmetic.
  const bool all_front = fminf(f.z0, fminf(f.z1, f.z2)) > 0.f;
  if (all_front) {
    const bool pos = area > 0.f;
    if (pos ? (e0 <= 0.f || e1 <= 0.f || e2 <= 0.f) : (e0 >= 0.f || e1 >= 0.f || e2 >= 0.f)) return false;
  }
  const float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
  const float t0 = w0 * f.z1 * f.z2;
  const float t1 = f.z0 * w1 * f.z2;
  const float t2 = f.z0 * f.z1 * w2;
  const float denom = fmaxf(t0 + t1 + t2, kRasterEps);
  const float b0 = t0 / denom, b1 = t1 / denom, b2 = t2 / denom;
  const float z = b0 * f.z0 + b1 * f.z1 + b2 * f.z2;
  if (z < 0.f) return false;
  if (!(b0 > 0.f && b1 > 0.f && b2 > 0.f)) return false;
  bary[0] = b0; bary[1] = b1; bary[2] = b2;
  *pz = z;
  return true;
}

// Conservative pixel reject: true only when NO point of the axis-aligned square [xlo, xhi] x [ylo, yhi] (a pixel's sample span) can
// pass sample_face's inside test.  Each edge function is affine, so its extreme over the square sits at a corner; if that extreme
// is on the outside by more than the rounding error of the evaluation (16 ulp of the two products: generous), every sample of the
// square fails `e > 0` (or `e < 0` for the other orientation) in sample_face as well: skipping them leaves the result unchanged.
HIFIHR_HD bool edge_outside_square(float ax, float ay, float bx, float by, float xlo, float xhi, float ylo, float yhi, bool pos) {
  const float dx = bx - ax, dy = by - ay;
  // e(p) = (px - ax) dy - (py - ay) dx grows with px iff dy > 0 and with py iff dx < 0; take the max corner (pos) or the min corner
  const float px = ((dy > 0.f) == pos) ? xhi : xlo;
  const float py = ((dx < 0.f) == pos) ? yhi : ylo;
  const float u = px - ax, v = py - ay;
  const float e = u * dy - v * dx;                       // same expression as edge_fn
  const float tol = 1e-6f * (fabsf(u * dy) + fabsf(v * dx));
  return pos ? (e < -tol) : (e > tol);
}
HIFIHR_HD bool square_misses_face(const FaceXYZ& f, float xlo, float xhi, float ylo, float yhi) {
  const float area = edge_fn(f.x2, f.y2, f.x0, f.y0, f.x1, f.y1) + kRasterEps;
  const bool pos = area > 0.f;
  return edge_outside_square(f.x1, f.y1, f.x2, f.y2, xlo, xhi, ylo, yhi, pos) ||
         edge_outside_square(f.x2, f.y2, f.x0, f.y0, xlo, xhi, ylo, yhi, pos) ||
         edge_outside_square(f.x0, f.y0, f.x1, f.y1, xlo, xhi, ylo, yhi, pos);
}

// Barycentrics of a sample known to be covered by face f (backward pass recomputation).
HIFIHR_HD void bary_of(const FaceXYZ& f, float px, float py, float* bary) {
  const float area = edge_fn(f.x2, f.y2, f.x0, f.y0, f.x1, f.y1) + kRasterEps;
  const float w0 = edge_fn(px, py, f.x1, f.y1, f.x2, f.y2) / area;
  const float w1 = edge_fn(px, py, f.x2, f.y2, f.x0, f.y0) / area;
  const float w2 = edge_fn(px, py, f.x0, f.y0, f.x1, f.y1) / area;
  const float t0 = w0 * f.z1 * f.z2, t1 = f.z0 * w1 * f.z2, t2 = f.z0 * f.z1 * w2;
  const float denom = fmaxf(t0 + t1 + t2, kRasterEps);
  bary[0] = t0 / denom; bary[1] = t1 / denom; bary[2] = t2 / denom;
}

// Reverse mode of bary_of: gb[3] -> gradient wrt the face's NDC vertices, g[9] = (gx0,gy0,gz0, gx1,gy1,gz1,
// gx2,gy2,gz2), overwritten.  Mirrors PyTorch3D's BarycentricPerspectiveCorrectionBackward +
// BarycentricCoordsBackward (the max(.,eps) clamps pass gradient straight through).
HIFIHR_HD void bary_bwd(const FaceXYZ& f, float px, float py, const float* gb, float* g) {
  const float area = edge_fn(f.x2, f.y2, f.x0, f.y0, f.x1, f.y1) + kRasterEps;
  const float e0 = edge_fn(px, py, f.x1, f.y1, f.x2, f.y2);
  const float e1 = edge_fn(px, py, f.x2, f.y2, f.x0, f.y0);
  const float e2 = edge_fn(px, py, f.x0, f.y0, f.x1, f.y1);
  const float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
  const float t0 = w0 * f.z1 * f.z2, t1 = f.z0 * w1 * f.z2, t2 = f.z0 * f.z1 * w2;
  const float denom = fmaxf(t0 + t1 + t2, kRasterEps);
  const float gden = -(t0 * gb[0] + t1 * gb[1] + t2 * gb[2]) / (denom * denom);
  const float gt0 = gden + gb[0] / denom, gt1 = gden + gb[1] / denom, gt2 = gden + gb[2] / denom;
  // t0 = w0 z1 z2 ; t1 = z0 w1 z2 ; t2 = z0 z1 w2
  const float gw0 = gt0 * f.z1 * f.z2, gw1 = gt1 * f.z0 * f.z2, gw2 = gt2 * f.z0 * f.z1;
  const float gz0 = gt1 * w1 * f.z2 + gt2 * f.z1 * w2;
  const float gz1 = gt0 * w0 * f.z2 + gt2 * f.z0 * w2;
  const float gz2 = gt0 * w0 * f.z1 + gt1 * f.z0 * w1;
  // w_i = e_i / area
  const float ge0 = gw0 / area, ge1 = gw1 / area, ge2 = gw2 / area;
  const float garea = -(gw0 * w0 + gw1 * w1 + gw2 * w2) / area;
  float gx0 = 0.f, gy0 = 0.f, gx1 = 0.f, gy1 = 0.f, gx2 = 0.f, gy2 = 0.f;
  // edge_fn(p, a, b): d/da = (py - by, bx - px), d/db = (-(py - ay), px - ax), d/dp = (by - ay, -(bx - ax))
  // e0 = edge_fn(p, v1, v2)
  gx1 += ge0 * (py - f.y2); gy1 += ge0 * (f.x2 - px); gx2 += ge0 * -(py - f.y1); gy2 += ge0 * (px - f.x1);
  // e1 = edge_fn(p, v2, v0)
  gx2 += ge1 * (py - f.y0); gy2 += ge1 * (f.x0 - px); gx0 += ge1 * -(py - f.y2); gy0 += ge1 * (px - f.x2);
  // e2 = edge_fn(p, v0, v1)
  gx0 += ge2 * (py - f.y1); gy0 += ge2 * (f.x1 - px); gx1 += ge2 * -(py - f.y0); gy1 += ge2 * (px - f.x0);
  // area = edge_fn(v2, v0, v1) (+eps): p = v2, a = v0, b = v1
  gx0 += garea * (f.y2 - f.y1); gy0 += garea * (f.x1 - f.x2);
  gx1 += garea * -(f.y2 - f.y0); gy1 += garea * (f.x2 - f.x0);
  gx2 += garea * (f.y1 - f.y0); gy2 += garea * -(f.x1 - f.x0);
  g[0] = gx0; g[1] = gy0; g[2] = gz0; g[3] = gx1; g[4] = gy1; g[5] = gz1; g[6] = gx2; g[7] = gy2; g[8] = gz2;
}

// ---- Phong shading of one covered sample (PyTorch3D phong_shading / _apply_lighting / DirectionalLights) ----
struct ShadeConsts {       // per renderer
  float amb[3];            // materials.ambient_color * lights.ambient_color
  float mdiff[3];          // materials.diffuse_color
  float spec[3];           // materials.specular_color * lights.specular_color
  float shininess;
  int point_light;         // 0: DirectionalLights (models_res_nimble.py:188-190); 1: PointLights (:191-198, light_estimation = false):
                           //    the "direction" input is the light's LOCATION and the direction of a sample is location - P
};
struct LightDir {          // per image
  float lc[3];             // lights.diffuse_color
  float l[3];              // normalised direction (directional), or the raw location (point)
  float inv_norm;          // 1 / max(|direction|, eps)
};

HIFIHR_HD void normalize3(const float* v, float* o, float* inv_len) {
  const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float inv = 1.0f / fmaxf(n, kNormEps);
  o[0] = v[0] * inv; o[1] = v[1] * inv; o[2] = v[2] * inv;
  *inv_len = inv;
}
// reverse of normalize3: go -> gv (overwritten).  For |v| <= eps the denominator is the constant eps.
HIFIHR_HD void normalize3_bwd(const float* v, const float* o, float inv, const float* go, float* gv) {
  const float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (n2 > kNormEps * kNormEps) {
    const float d = o[0] * go[0] + o[1] * go[1] + o[2] * go[2];
    gv[0] = (go[0] - o[0] * d) * inv; gv[1] = (go[1] - o[1] * d) * inv; gv[2] = (go[2] - o[2] * d) * inv;
  } else {
    gv[0] = go[0] * inv; gv[1] = go[1] * inv; gv[2] = go[2] * inv;
  }
}

struct ShadeTmp {
  float nh[3], inv_n, vh[3], inv_v, cosang, r[3], d, alpha, pw, tex[3], N[3], Vd[3], lh[3], Lraw[3], inv_l;
};

// P, N, T: interpolated position, normal, texel.  rgb out.
HIFIHR_HD void shade_fwd(const ShadeConsts& c, const LightDir& L, const float* P, const float* N, const float* T,
                         float* rgb, ShadeTmp* t) {
  float nh[3], inv_n, vh[3], inv_v;
  normalize3(N, nh, &inv_n);
  float lh[3] = {L.l[0], L.l[1], L.l[2]}, inv_l = 1.f, Lraw[3] = {0.f, 0.f, 0.f};
  if (c.point_light) {                                // PointLights: direction = location - point, normalised per sample
    Lraw[0] = L.l[0] - P[0]; Lraw[1] = L.l[1] - P[1]; Lraw[2] = L.l[2] - P[2];
    normalize3(Lraw, lh, &inv_l);
  }
  const float cosang = nh[0] * lh[0] + nh[1] * lh[1] + nh[2] * lh[2];
  const float angle = fmaxf(cosang, 0.f);
  const float Vd[3] = {-P[0], -P[1], -P[2]};          // camera centre (0,0,0) - point
  normalize3(Vd, vh, &inv_v);
  const float r[3] = {-lh[0] + 2.f * (cosang * nh[0]), -lh[1] + 2.f * (cosang * nh[1]), -lh[2] + 2.f * (cosang * nh[2])};
  const float d = vh[0] * r[0] + vh[1] * r[1] + vh[2] * r[2];
  const float alpha = (cosang > 0.f) ? fmaxf(d, 0.f) : 0.f;
  const float pw = powf(alpha, c.shininess);
#pragma unroll
  for (int k = 0; k < 3; ++k) rgb[k] = (c.amb[k] + c.mdiff[k] * (L.lc[k] * angle)) * T[k] + c.spec[k] * pw;
  if (t) {
    for (int k = 0; k < 3; ++k) { t->nh[k] = nh[k]; t->vh[k] = vh[k]; t->r[k] = r[k]; t->tex[k] = T[k]; t->N[k] = N[k]; t->Vd[k] = Vd[k]; }
    t->inv_n = inv_n; t->inv_v = inv_v; t->cosang = cosang; t->d = d; t->alpha = alpha; t->pw = pw;
    for (int k = 0; k < 3; ++k) { t->lh[k] = lh[k]; t->Lraw[k] = Lraw[k]; }
    t->inv_l = inv_l;
  }
}

// Reverse of shade_fwd.  g_rgb[3] -> gP[3], gN[3], gT[3] (overwritten), glc[3] and gl[3] (gradient wrt the
// light colour and wrt the NORMALISED light direction; both ACCUMULATED).
HIFIHR_HD void shade_bwd(const ShadeConsts& c, const LightDir& L, const float* P, const float* N, const float* T,
                         const float* g_rgb, float* gP, float* gN, float* gT, float* glc, float* gl) {
  float rgb[3];
  ShadeTmp t;
  shade_fwd(c, L, P, N, T, rgb, &t);
  const float angle = fmaxf(t.cosang, 0.f);
  float g_cos = 0.f, g_pw = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    gT[k] = g_rgb[k] * (c.amb[k] + c.mdiff[k] * (L.lc[k] * angle));
    const float g_diff = g_rgb[k] * c.mdiff[k] * T[k];
    glc[k] += g_diff * angle;
    if (t.cosang > 0.f) g_cos += g_diff * L.lc[k];
    g_pw += g_rgb[k] * c.spec[k];
  }
  // pw = alpha^s ; d(alpha^s)/d(alpha) = s alpha^(s-1) (0 at alpha = 0 for s > 1)
  const float g_alpha = (t.alpha > 0.f) ? g_pw * c.shininess * powf(t.alpha, c.shininess - 1.f) : 0.f;
  const float g_d = (t.cosang > 0.f && t.d > 0.f) ? g_alpha : 0.f;
  float g_vh[3], g_r[3], g_nh[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { g_vh[k] = g_d * t.r[k]; g_r[k] = g_d * t.vh[k]; }
  // r = -l + 2 cos nh
  g_cos += 2.f * (g_r[0] * t.nh[0] + g_r[1] * t.nh[1] + g_r[2] * t.nh[2]);
  float g_lh[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    g_nh[k] = 2.f * t.cosang * g_r[k] + g_cos * t.lh[k];
    g_lh[k] = -g_r[k] + g_cos * t.nh[k];
  }
  normalize3_bwd(N, t.nh, t.inv_n, g_nh, gN);
  float gVd[3];
  normalize3_bwd(t.Vd, t.vh, t.inv_v, g_vh, gVd);
  gP[0] = -gVd[0]; gP[1] = -gVd[1]; gP[2] = -gVd[2];
  if (c.point_light) {                                // the direction depends on the point: d(location - P)/dP = -I; the location is constant
    float gL[3];
    normalize3_bwd(t.Lraw, t.lh, t.inv_l, g_lh, gL);
    gP[0] -= gL[0]; gP[1] -= gL[1]; gP[2] -= gL[2];
  } else {
    gl[0] += g_lh[0]; gl[1] += g_lh[1]; gl[2] += g_lh[2];
  }
}

}  // namespace hifihr
