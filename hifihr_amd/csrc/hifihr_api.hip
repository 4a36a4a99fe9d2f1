// C ABI of libhifihr.so (see include/hifihr.h).  Thin: argument checks, table upload at create time,
// kernel launches on the caller's stream.  No allocation and no synchronisation in compute calls.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/hifihr.h"
#include "hifihr_internal.h"
#include "mano_math.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess) return fail(HIFIHR_EHIP, "%s: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

int upload(DevBuf& b, const std::vector<float>& h) {
  HIP_TRY(hipMalloc(&b.p, h.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(b.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return HIFIHR_OK;
}

}  // namespace

struct hifihr_mano {
  DevBuf tmpl, sd, pd, w, jreg, comps, mean, jt, jsd;
  hifihr::ManoDev dev;
};

struct hifihr_lbs {
  DevBuf tmpl, sd, widx, wval, jt, jsd, parent;
  hifihr::LbsDev dev;
};

struct hifihr_renderer {
  DevBuf faces, vf_off, vf_idx;
  DevBuf faces_uvs, verts_uvs;       // TexturesUV tables (hifihr_renderer_set_uv), empty otherwise
  int n_uv = 0;
  hifihr::RenderDev dev;
};

namespace {
int upload_i(DevBuf& b, const std::vector<int>& h) {
  HIP_TRY(hipMalloc(&b.p, (h.size() ? h.size() : 1) * sizeof(int)));
  HIP_TRY(hipMemcpy(b.p, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
  return HIFIHR_OK;
}
}  // namespace

extern "C" {

int hifihr_version(void) { return 1; }
const char* hifihr_last_error(void) { return g_err; }

int hifihr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int hifihr_mano_create(hifihr_mano_t** out, const float* v_template, const float* shapedirs, const float* posedirs,
                       const float* j_regressor, const float* weights, const float* comps, const float* mean) {
  using namespace hifihr;
  if (!out || !v_template || !shapedirs || !posedirs || !j_regressor || !weights || !comps || !mean)
    return fail(HIFIHR_EINVAL, "hifihr_mano_create: null argument");
  hifihr_mano* h = new (std::nothrow) hifihr_mano();
  if (!h) return fail(HIFIHR_ENOMEM, "hifihr_mano_create: out of host memory");
  // host re-layout: [v][c][k] -> [k][c][vpad]
  std::vector<float> tm(3 * kNVP, 0.f), sd((size_t)kNB * 3 * kNVP, 0.f), pd((size_t)kNP * 3 * kNVP, 0.f),
      w((size_t)kNJ * kNVP, 0.f), jr((size_t)kNJ * kNVP, 0.f), jt(kNJ * 3), jsd((size_t)kNJ * 3 * kNB);
  for (int v = 0; v < kNV; ++v)
    for (int c = 0; c < 3; ++c) {
      tm[c * kNVP + v] = v_template[v * 3 + c];
      for (int k = 0; k < kNB; ++k) sd[((size_t)k * 3 + c) * kNVP + v] = shapedirs[((size_t)v * 3 + c) * kNB + k];
      for (int k = 0; k < kNP; ++k) pd[((size_t)k * 3 + c) * kNVP + v] = posedirs[((size_t)v * 3 + c) * kNP + k];
    }
  for (int v = 0; v < kNV; ++v)
    for (int j = 0; j < kNJ; ++j) {
      w[(size_t)j * kNVP + v] = weights[v * kNJ + j];
      jr[(size_t)j * kNVP + v] = j_regressor[(size_t)j * kNV + v];
    }
  // J = J_regressor (v_template + shapedirs beta) is linear in beta: fold the regressor into 16x3 (+16x3x10)
  // tables once, accumulated in double (reference my_mano.py:386-389 does the 778-term sums per call).
  for (int j = 0; j < kNJ; ++j)
    for (int c = 0; c < 3; ++c) {
      double a = 0.0;
      for (int v = 0; v < kNV; ++v) a += (double)j_regressor[(size_t)j * kNV + v] * (double)v_template[v * 3 + c];
      jt[j * 3 + c] = (float)a;
      for (int k = 0; k < kNB; ++k) {
        double s = 0.0;
        for (int v = 0; v < kNV; ++v) s += (double)j_regressor[(size_t)j * kNV + v] * (double)shapedirs[((size_t)v * 3 + c) * kNB + k];
        jsd[((size_t)j * 3 + c) * kNB + k] = (float)s;
      }
    }
  std::vector<float> cm(comps, comps + kNPCA * kNPCA), mn(mean, mean + kNPCA);
  int rc;
  if ((rc = upload(h->tmpl, tm)) || (rc = upload(h->sd, sd)) || (rc = upload(h->pd, pd)) || (rc = upload(h->w, w)) ||
      (rc = upload(h->jreg, jr)) || (rc = upload(h->comps, cm)) || (rc = upload(h->mean, mn)) ||
      (rc = upload(h->jt, jt)) || (rc = upload(h->jsd, jsd))) {
    delete h;
    return rc;
  }
  h->dev = ManoDev{(const float*)h->tmpl.p, (const float*)h->sd.p,   (const float*)h->pd.p,
                   (const float*)h->w.p,    (const float*)h->jreg.p, (const float*)h->comps.p,
                   (const float*)h->mean.p, (const float*)h->jt.p,   (const float*)h->jsd.p};
  *out = h;
  return HIFIHR_OK;
}

int hifihr_mano_destroy(hifihr_mano_t* h) {
  delete h;
  return HIFIHR_OK;
}

int hifihr_lbs_create(hifihr_lbs_t** out, int V, int J, int S, const float* v_template, const float* shapedirs, const float* j_regressor,
                      const float* weights, const int* parents) {
  using namespace hifihr;
  if (!out || !v_template || (S > 0 && !shapedirs) || !j_regressor || !weights || !parents)
    return fail(HIFIHR_EINVAL, "hifihr_lbs_create: null argument");
  if (V < 1 || J < 1 || J > kLbsMaxJ || S < 0 || S > kLbsMaxS) return fail(HIFIHR_EINVAL, "hifihr_lbs_create: V >= 1, 1 <= J <= 32, 0 <= S <= 32");
  if (parents[0] != -1) return fail(HIFIHR_EINVAL, "hifihr_lbs_create: parents[0] must be -1");
  for (int j = 1; j < J; ++j)
    if (parents[j] < 0 || parents[j] >= j) return fail(HIFIHR_EINVAL, "hifihr_lbs_create: parents must be topologically ordered (0 <= parents[j] < j)");
  int K = 1;
  for (int v = 0; v < V; ++v) {
    int nz = 0;
    for (int j = 0; j < J; ++j) nz += weights[(size_t)v * J + j] != 0.f;
    if (nz > K) K = nz;
  }
  if (K > kLbsMaxK) return fail(HIFIHR_EINVAL, "hifihr_lbs_create: more than 8 non-zero skin weights on a vertex");
  hifihr_lbs* h = new (std::nothrow) hifihr_lbs();
  if (!h) return fail(HIFIHR_ENOMEM, "hifihr_lbs_create: out of host memory");
  const int Vp = (V + 63) / 64 * 64;
  std::vector<float> tm((size_t)3 * Vp, 0.f), sd((size_t)(S ? S : 1) * 3 * Vp, 0.f), wv((size_t)K * Vp, 0.f), jt((size_t)J * 3), jsd((size_t)J * 3 * (S ? S : 1), 0.f);
  std::vector<int> wi((size_t)K * Vp, 0), par(parents, parents + J);
  for (int v = 0; v < V; ++v) {
    for (int c = 0; c < 3; ++c) {
      tm[(size_t)c * Vp + v] = v_template[v * 3 + c];
      for (int k = 0; k < S; ++k) sd[((size_t)k * 3 + c) * Vp + v] = shapedirs[((size_t)v * 3 + c) * S + k];
    }
    int n = 0;
    for (int j = 0; j < J; ++j)
      if (weights[(size_t)v * J + j] != 0.f) { wi[(size_t)n * Vp + v] = j; wv[(size_t)n * Vp + v] = weights[(size_t)v * J + j]; ++n; }
  }
  for (int j = 0; j < J; ++j)            // J = J_regressor (v_template + shapedirs beta), folded once in double
    for (int c = 0; c < 3; ++c) {
      double a = 0.0;
      for (int v = 0; v < V; ++v) a += (double)j_regressor[(size_t)j * V + v] * (double)v_template[v * 3 + c];
      jt[j * 3 + c] = (float)a;
      for (int k = 0; k < S; ++k) {
        double s = 0.0;
        for (int v = 0; v < V; ++v) s += (double)j_regressor[(size_t)j * V + v] * (double)shapedirs[((size_t)v * 3 + c) * S + k];
        jsd[((size_t)j * 3 + c) * S + k] = (float)s;
      }
    }
  int rc;
  if ((rc = upload(h->tmpl, tm)) || (rc = upload(h->sd, sd)) || (rc = upload(h->wval, wv)) || (rc = upload(h->jt, jt)) ||
      (rc = upload(h->jsd, jsd)) || (rc = upload_i(h->widx, wi)) || (rc = upload_i(h->parent, par))) {
    delete h;
    return rc;
  }
  h->dev = LbsDev{V, Vp, J, S, K, (const float*)h->tmpl.p, (const float*)h->sd.p, (const int*)h->widx.p, (const float*)h->wval.p,
                  (const float*)h->jt.p, (const float*)h->jsd.p, (const int*)h->parent.p};
  *out = h;
  return HIFIHR_OK;
}

int hifihr_lbs_destroy(hifihr_lbs_t* h) {
  delete h;
  return HIFIHR_OK;
}

int hifihr_lbs_fwd(const hifihr_lbs_t* h, const float* theta, const float* beta, int B, float* verts, float* joints, void* stream) {
  if (!h || !theta || (h->dev.S > 0 && !beta) || !verts || B < 0) return fail(HIFIHR_EINVAL, "hifihr_lbs_fwd: bad argument");
  if (B == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_lbs_fwd(h->dev, theta, beta, B, verts, joints, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_lbs_bwd(const hifihr_lbs_t* h, const float* theta, const float* beta, const float* gverts, const float* gjoints, int B,
                   float* scratch_zeroed, float* gtheta, float* gbeta_zeroed, void* stream) {
  if (!h || !theta || !gverts || !scratch_zeroed || !gtheta || (h->dev.S > 0 && (!beta || !gbeta_zeroed)) || B < 0)
    return fail(HIFIHR_EINVAL, "hifihr_lbs_bwd: bad argument");
  if (B == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_lbs_bwd(h->dev, theta, beta, gverts, gjoints, B, scratch_zeroed, gtheta, gbeta_zeroed, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_mano_lbs_fwd(const hifihr_mano_t* h, const float* pose, const float* beta, int B, float* verts, float* jtr,
                        float* saved, void* stream) {
  if (!h || !pose || !beta || !verts || B < 0) return fail(HIFIHR_EINVAL, "hifihr_mano_lbs_fwd: bad argument");
  if (B == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_mano_fwd(h->dev, pose, beta, B, verts, jtr, saved, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_mano_lbs_bwd(const hifihr_mano_t* h, const float* pose, const float* beta, const float* saved,
                        const float* gverts, const float* gjtr, int B, float* gpose, float* gbeta, void* stream) {
  if (!h || !pose || !beta || !saved || !gpose || !gbeta || B < 0)
    return fail(HIFIHR_EINVAL, "hifihr_mano_lbs_bwd: bad argument");
  if (B == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_mano_bwd(h->dev, pose, beta, saved, gverts, gjtr, B, gpose, gbeta, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_mano_full_fwd(const hifihr_mano_t* h, const float* pose, const float* beta, int B, int root_id, const float* root_xyz,
                         float* verts, float* joints_rel, float* verts_rel, float* verts_cam, float* root, float* saved, void* stream) {
  if (!h || !pose || !beta || !verts || !joints_rel || !verts_rel || B < 0 || root_id >= 21)
    return fail(HIFIHR_EINVAL, "hifihr_mano_full_fwd: bad argument");
  if (B == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_mano_full_fwd(h->dev, pose, beta, B, root_id, root_xyz, verts, joints_rel, verts_rel, verts_cam, root, saved,
                                       (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_mano_full_bwd(const hifihr_mano_t* h, const float* pose, const float* beta, const float* saved, const float* gjoints_rel,
                         const float* gverts_rel, const float* gverts_cam, const float* groot, const float* gpose_add,
                         const float* gbeta_add, int B, int root_id, float* gpose, float* gbeta, void* stream) {
  if (!h || !pose || !beta || !saved || !gpose || !gbeta || B < 0 || root_id >= 21)
    return fail(HIFIHR_EINVAL, "hifihr_mano_full_bwd: bad argument");
  if (B == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_mano_full_bwd(h->dev, pose, beta, saved, gjoints_rel, gverts_rel, gverts_cam, groot, gpose_add, gbeta_add, B, root_id,
                                       gpose, gbeta, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_mano_joints_fwd(const hifihr_mano_t* h, const float* verts, int B, int root_id, float* joints_rel,
                           float* verts_rel, float* root, void* stream) {
  if (!h || !verts || !joints_rel || B < 0 || root_id >= 21) return fail(HIFIHR_EINVAL, "hifihr_mano_joints_fwd: bad argument");
  if (B == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_mano_joints_fwd(h->dev, verts, B, root_id, joints_rel, verts_rel, root, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_mano_joints_bwd(const hifihr_mano_t* h, const float* gjoints_rel, const float* gverts_rel, const float* groot,
                           int B, int root_id, float* gverts, void* stream) {
  if (!h || !gverts || B < 0 || root_id >= 21) return fail(HIFIHR_EINVAL, "hifihr_mano_joints_bwd: bad argument");
  if (B == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_mano_joints_bwd(h->dev, gjoints_rel, gverts_rel, groot, B, root_id, gverts, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_renderer_create(hifihr_renderer_t** out, const int32_t* faces, int V, int F, int image_size, int aa,
                           const float* ambient3, const float* mat_diffuse3, const float* specular3, float shininess,
                           const float* background3) {
  if (!out || !faces || V <= 0 || F <= 0 || image_size <= 0 || aa < 1 || aa > 3 || !ambient3 || !mat_diffuse3 || !specular3 ||
      !background3)
    return fail(HIFIHR_EINVAL, "hifihr_renderer_create: bad argument");
  std::vector<int> fv(faces, faces + (size_t)F * 3), off(V + 1, 0), idx((size_t)F * 3);
  for (int i = 0; i < F * 3; ++i) {
    if (fv[i] < 0 || fv[i] >= V) return fail(HIFIHR_EINVAL, "hifihr_renderer_create: face index out of range");
    off[fv[i] + 1]++;
  }
  for (int v = 0; v < V; ++v) off[v + 1] += off[v];
  std::vector<int> cur(off.begin(), off.end() - 1);
  for (int f = 0; f < F; ++f)            // incident faces in ascending face order => deterministic normal sums
    for (int k = 0; k < 3; ++k) idx[cur[fv[3 * f + k]]++] = f * 4 + k;
  hifihr_renderer* h = new (std::nothrow) hifihr_renderer();
  if (!h) return fail(HIFIHR_ENOMEM, "hifihr_renderer_create: out of host memory");
  int rc;
  if ((rc = upload_i(h->faces, fv)) || (rc = upload_i(h->vf_off, off)) || (rc = upload_i(h->vf_idx, idx))) {
    delete h;
    return rc;
  }
  hifihr::RenderDev& d = h->dev;
  d.V = V; d.F = F; d.H = image_size; d.aa = aa;
  d.faces = (const int*)h->faces.p; d.vf_off = (const int*)h->vf_off.p; d.vf_idx = (const int*)h->vf_idx.p;
  for (int k = 0; k < 3; ++k) {
    d.sc.amb[k] = ambient3[k]; d.sc.mdiff[k] = mat_diffuse3[k]; d.sc.spec[k] = specular3[k]; d.bg[k] = background3[k];
  }
  d.sc.shininess = shininess;
  d.sc.point_light = 0;
  *out = h;
  return HIFIHR_OK;
}

int hifihr_texture_pca_fwd(const float* coef, const float* basis, const float* mean, int B, int K, long n, float* out, void* stream) {
  if (!coef || !basis || !out || B <= 0 || K <= 0 || K > 32 || n < 4 || n % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_texture_pca_fwd: bad argument (1 <= K <= 32, n % 4 == 0)");
  HIP_TRY(hifihr::launch_texpca_fwd(coef, basis, mean, B, K, n, out, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_texture_pca_bwd(const float* gtex, const float* basis, int B, int K, long n, float* dcoef_zeroed, void* stream) {
  if (!gtex || !basis || !dcoef_zeroed || B <= 0 || K <= 0 || K > 32 || n < 4 || n % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_texture_pca_bwd: bad argument (1 <= K <= 32, n % 4 == 0)");
  HIP_TRY(hifihr::launch_texpca_bwd(gtex, basis, B, K, n, dcoef_zeroed, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_renderer_set_light_mode(hifihr_renderer_t* h, int point_lights) {
  if (!h || point_lights < 0 || point_lights > 1) return fail(HIFIHR_EINVAL, "hifihr_renderer_set_light_mode: bad argument");
  h->dev.sc.point_light = point_lights;
  return HIFIHR_OK;
}

int hifihr_renderer_set_uv(hifihr_renderer_t* h, const int32_t* faces_uvs, const float* verts_uvs, int n_uv) {
  if (!h || !faces_uvs || !verts_uvs || n_uv <= 0) return fail(HIFIHR_EINVAL, "hifihr_renderer_set_uv: bad argument");
  for (int i = 0; i < 3 * h->dev.F; ++i)
    if (faces_uvs[i] < 0 || faces_uvs[i] >= n_uv) return fail(HIFIHR_EINVAL, "hifihr_renderer_set_uv: uv index out of range");
  DevBuf fu, vu;
  HIP_TRY(hipMalloc(&fu.p, (size_t)3 * h->dev.F * sizeof(int)));
  HIP_TRY(hipMemcpy(fu.p, faces_uvs, (size_t)3 * h->dev.F * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&vu.p, (size_t)2 * n_uv * sizeof(float)));
  HIP_TRY(hipMemcpy(vu.p, verts_uvs, (size_t)2 * n_uv * sizeof(float), hipMemcpyHostToDevice));
  std::swap(h->faces_uvs.p, fu.p);
  std::swap(h->verts_uvs.p, vu.p);
  h->n_uv = n_uv;
  return HIFIHR_OK;
}

size_t hifihr_render_uv_scratch_bytes(const hifihr_renderer_t* h, int B) {
  (void)h; (void)B;
  return 0;                       // the texture is sampled inside the tile kernels: no per-sample scratch
}

int hifihr_render_fwd_uv(const hifihr_renderer_t* h, const float* verts, const float* maps, int TH, int TW, const float* cam,
                         const float* light_color, const float* light_dir, int B, float* rgba, int32_t* face_id, float* texels_scratch,
                         void* ws, void* stream) {
  if (!h || !verts || !maps || TH < 1 || TW < 1 || !cam || !light_color || !light_dir || !rgba || !face_id || !ws || B < 0)
    return fail(HIFIHR_EINVAL, "hifihr_render_fwd_uv: bad argument");
  (void)texels_scratch;
  if (h->n_uv == 0) return fail(HIFIHR_EINVAL, "hifihr_render_fwd_uv: no UV tables (hifihr_renderer_set_uv)");
  if (B == 0) return HIFIHR_OK;
  const hifihr::TexUvPass uv{static_cast<const int*>(h->faces_uvs.p), static_cast<const float*>(h->verts_uvs.p), maps, nullptr, TH, TW};
  // (the vertex-colour input of the tile kernel is not used in this mode: any [V][3] buffer serves, the vertices themselves here)
  HIP_TRY(hifihr::launch_render_fwd(h->dev, verts, verts, (long)h->dev.V * 3, cam, light_color, light_dir, B, rgba, face_id, ws,
                                    (hipStream_t)stream, &uv));
  return HIFIHR_OK;
}

int hifihr_render_bwd_uv(const hifihr_renderer_t* h, const float* verts, const float* maps, int TH, int TW, const float* cam,
                         const float* light_color, const float* light_dir, const int32_t* face_id, const float* grad_rgba, int B,
                         const float* texels_scratch, float* gtexels_scratch, float* gverts, float* gmaps_acc, float* glight_color,
                         float* glight_dir, void* ws, void* stream) {
  if (!h || !verts || !maps || TH < 1 || TW < 1 || !cam || !light_color || !light_dir || !face_id || !grad_rgba || !gverts || !glight_color ||
      !glight_dir || !ws || B < 0)
    return fail(HIFIHR_EINVAL, "hifihr_render_bwd_uv: bad argument");
  (void)texels_scratch; (void)gtexels_scratch;
  if (h->n_uv == 0) return fail(HIFIHR_EINVAL, "hifihr_render_bwd_uv: no UV tables (hifihr_renderer_set_uv)");
  if (B == 0) return HIFIHR_OK;
  const hifihr::TexUvPass uv{static_cast<const int*>(h->faces_uvs.p), static_cast<const float*>(h->verts_uvs.p), maps, gmaps_acc, TH, TW};
  HIP_TRY(hifihr::launch_render_bwd(h->dev, verts, cam, light_color, light_dir, face_id, grad_rgba, B, gverts, nullptr, glight_color,
                                    glight_dir, ws, (hipStream_t)stream, &uv));
  return HIFIHR_OK;
}

int hifihr_renderer_destroy(hifihr_renderer_t* h) {
  delete h;
  return HIFIHR_OK;
}

size_t hifihr_render_workspace_bytes(const hifihr_renderer_t* h, int B) {
  if (!h || B < 0) return 0;
  return hifihr::render_workspace_bytes(h->dev, B);
}

int hifihr_render_fwd(const hifihr_renderer_t* h, const float* verts, const float* vcolors, int vcolors_batched, const float* cam,
                      const float* light_color, const float* light_dir, int B, float* rgba, int32_t* face_id, void* ws,
                      void* stream) {
  if (!h || !verts || !vcolors || !cam || !light_color || !light_dir || !rgba || !face_id || !ws || B < 0)
    return fail(HIFIHR_EINVAL, "hifihr_render_fwd: bad argument");
  if (B == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_render_fwd(h->dev, verts, vcolors, vcolors_batched ? (long)h->dev.V * 3 : 0L, cam, light_color,
                                    light_dir, B, rgba, face_id, ws, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_render_bwd(const hifihr_renderer_t* h, const float* verts, const float* cam, const float* light_color,
                      const float* light_dir, const int32_t* face_id, const float* grad_rgba, int B, float* gverts,
                      float* gvcolors, float* glight_color, float* glight_dir, void* ws, void* stream) {
  if (!h || !verts || !cam || !light_color || !light_dir || !face_id || !grad_rgba || !gverts || !glight_color || !glight_dir ||
      !ws || B < 0)
    return fail(HIFIHR_EINVAL, "hifihr_render_bwd: bad argument");
  if (B == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_render_bwd(h->dev, verts, cam, light_color, light_dir, face_id, grad_rgba, B, gverts, gvcolors,
                                    glight_color, glight_dir, ws, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float grad_scale,
                     float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || step < 1) return fail(HIFIHR_EINVAL, "hifihr_adam_step: bad argument");
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
    return fail(HIFIHR_EINVAL, "hifihr_adam_step: buffers must be 16-byte aligned");
  if (n == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_adam(params, grads, exp_avg, exp_avg_sq, n, grad_scale, lr, beta1, beta2, eps, weight_decay, step,
                              nullptr, (hipStream_t)stream));
  return HIFIHR_OK;
}

size_t hifihr_adam_state_bytes(void) { return hifihr::adam_state_bytes(); }

int hifihr_adam_step_counted(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float grad_scale, float eps,
                             float weight_decay, void* state_d, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !state_d) return fail(HIFIHR_EINVAL, "hifihr_adam_step_counted: bad argument");
  if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) || ((uintptr_t)state_d & 7))
    return fail(HIFIHR_EINVAL, "hifihr_adam_step_counted: buffers must be 16-byte aligned (the state: 8)");
  HIP_TRY(hifihr::launch_adam_counted(params, grads, exp_avg, exp_avg_sq, n, grad_scale, eps, weight_decay, state_d, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_adam_step_dyn(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float grad_scale,
                         float beta1, float beta2, float eps, float weight_decay, const float* dyn_d, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !dyn_d) return fail(HIFIHR_EINVAL, "hifihr_adam_step_dyn: bad argument");
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
    return fail(HIFIHR_EINVAL, "hifihr_adam_step_dyn: buffers must be 16-byte aligned");
  if (n == 0) return HIFIHR_OK;
  HIP_TRY(hifihr::launch_adam(params, grads, exp_avg, exp_avg_sq, n, grad_scale, 0.f, beta1, beta2, eps, weight_decay, 1, dyn_d,
                              (hipStream_t)stream));
  return HIFIHR_OK;
}

static int conv_dims_ok(int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
  return N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && stride > 0 && pad >= 0 && (H + 2 * pad - R) >= 0 &&
         (W + 2 * pad - S) >= 0;
}

size_t hifihr_conv2d_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int bwd_data) {
  if (!conv_dims_ok(N, H, W, C, K, R, S, stride, pad)) return 0;
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
  const hifihr::ConvGeom f{N, H, W, C, OH, OW, K, R, S, stride, pad, 0};
  const hifihr::ConvGeom b{N, OH, OW, K, H, W, C, R, S, stride, pad, 1};
  return hifihr::conv_sk_workspace_bytes(bwd_data ? b : f);
}

int hifihr_conv2d_fwd(const float* x, const float* w, const float* bias, int act, float* y, int N, int H, int W, int C, int K, int R,
                      int S, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !w || !y || !conv_dims_ok(N, H, W, C, K, R, S, stride, pad) || C % 4 || act < 0 || act > 1)
    return fail(HIFIHR_EINVAL, "hifihr_conv2d_fwd: bad argument (C must be a multiple of 4; act 0/1)");
  hifihr::ConvGeom g{N, H, W, C, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K, R, S, stride, pad, 0, act};
  HIP_TRY(hifihr::launch_conv_igemm(g, x, w, bias, y, nullptr, ws, ws_bytes, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv2d_describe(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dgrad, char* out, int cap) {
  if (!out || cap < 24 || !conv_dims_ok(N, H, W, C, K, R, S, stride, pad)) return fail(HIFIHR_EINVAL, "hifihr_conv2d_describe: bad argument");
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
  const hifihr::ConvGeom g = dgrad == 1 ? hifihr::ConvGeom{N, OH, OW, K, H, W, C, R, S, stride, pad, 1} : hifihr::ConvGeom{N, H, W, C, OH, OW, K, R, S, stride, pad, 0};
  if (hifihr::conv_is_gemm(g)) {          // 1x1 / stride 1: the GEMM kernels (the weight gradient needs the caller's workspace for that)
    const long M = (long)N * OH * OW;
    const bool wg = hifihr::conv_wgrad_workspace_bytes(g) > 0;      // (a 1x1 shape has no other slab kernel)
    if (dgrad == 2 && wg) hifihr::bgemm_describe_batch(1, K, C, (int)M, 1, out, cap);
    else if (dgrad != 2) hifihr::bgemm_describe_batch(0, (int)M, g.OC, g.IC, 1, out, cap);
    if (dgrad != 2 || wg) return HIFIHR_OK;
  }
  if (dgrad == 2) snprintf(out, cap, "%s", hifihr::conv_halo_wgrad_supported(g) ? "conv_halo_wgrad_kernel" : hifihr::conv_stem_wgrad_supported(g) ? "conv_stem_wgrad_kernel" : "conv_wgrad_kernel");
  else snprintf(out, cap, "%s", hifihr::conv_halo_supported(g, nullptr) ? "conv_halo_kernel" : hifihr::conv_stem_supported(g, nullptr) ? "conv_stem_kernel" :
                (dgrad == 0 && hifihr::conv_rows_supported(g, nullptr)) ? "bgemm_nt_rows_kernel<2>" : "conv_igemm_kernel");      // (strided forward: the gathering row-share GEMM)
  return HIFIHR_OK;
}

static bool pair_geoms(int N, int H, int W, int C, int stride, int K1, int R1, int pad1, int K2, int R2, int pad2, hifihr::ConvGeom* g1, hifihr::ConvGeom* g2) {
  if (!conv_dims_ok(N, H, W, C, K1, R1, R1, stride, pad1) || !conv_dims_ok(N, H, W, C, K2, R2, R2, stride, pad2) || C % 4) return false;
  *g1 = hifihr::ConvGeom{N, H, W, C, (H + 2 * pad1 - R1) / stride + 1, (W + 2 * pad1 - R1) / stride + 1, K1, R1, R1, stride, pad1, 0};
  *g2 = hifihr::ConvGeom{N, H, W, C, (H + 2 * pad2 - R2) / stride + 1, (W + 2 * pad2 - R2) / stride + 1, K2, R2, R2, stride, pad2, 0};
  return true;
}
int hifihr_conv2d_fwd_bnstats_pair_supported(int N, int H, int W, int C, int stride, int K1, int R1, int pad1, int K2, int R2, int pad2) {
  hifihr::ConvGeom g1, g2;
  return (pair_geoms(N, H, W, C, stride, K1, R1, pad1, K2, R2, pad2, &g1, &g2) && hifihr::conv_rows_pair_supported(g1, g2) &&
          hifihr::bgemm_nt_stats_supported(K1) && hifihr::bgemm_nt_stats_supported(K2)) ? 1 : 0;
}
int hifihr_conv2d_fwd_bnstats_pair(const float* x, const float* w1, float* y1, float* stats1, int K1, int R1, int pad1, const float* w2, float* y2,
                                   float* stats2, int K2, int R2, int pad2, int N, int H, int W, int C, int stride, void* stream) {
  hifihr::ConvGeom g1, g2;
  if (!x || !w1 || !y1 || !stats1 || !w2 || !y2 || !stats2 || !pair_geoms(N, H, W, C, stride, K1, R1, pad1, K2, R2, pad2, &g1, &g2) ||
      !hifihr_conv2d_fwd_bnstats_pair_supported(N, H, W, C, stride, K1, R1, pad1, K2, R2, pad2))
    return fail(HIFIHR_EINVAL, "hifihr_conv2d_fwd_bnstats_pair: bad argument or unsupported pair (ask hifihr_conv2d_fwd_bnstats_pair_supported)");
  const float* zeros = hifihr::conv_halo_zero_page((hipStream_t)stream);
  if (zeros == nullptr) return fail(HIFIHR_EINVAL, "hifihr_conv2d_fwd_bnstats_pair: the zero page is not available inside a capture (call once outside)");
  HIP_TRY(hifihr::launch_conv_rows_pair(g1, x, w1, y1, stats1, g2, w2, y2, stats2, zeros, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv2d_fwd_bnstats(const float* x, const float* w, float* y, float* stats, int N, int H, int W, int C, int K, int R,
                              int S, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !w || !y || !stats || !conv_dims_ok(N, H, W, C, K, R, S, stride, pad) || C % 4)
    return fail(HIFIHR_EINVAL, "hifihr_conv2d_fwd_bnstats: bad argument (C must be a multiple of 4)");
  hifihr::ConvGeom g{N, H, W, C, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K, R, S, stride, pad, 0};
  HIP_TRY(hifihr::launch_conv_igemm(g, x, w, nullptr, y, stats, ws, ws_bytes, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv2d_bwd_data(const float* dy, const float* w, float* dx, float* wt_scratch, int N, int H, int W, int C, int K,
                           int R, int S, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !w || !dx || !wt_scratch || !conv_dims_ok(N, H, W, C, K, R, S, stride, pad) || K % 4 || (K % 16 && stride != 1))
    return fail(HIFIHR_EINVAL, "hifihr_conv2d_bwd_data: bad argument (K % 4 == 0; K % 16 == 0 when stride > 1)");
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
  HIP_TRY(hifihr::launch_weight_transpose(w, wt_scratch, K, R * S, C, (hipStream_t)stream));
  hifihr::ConvGeom g{N, OH, OW, K, H, W, C, R, S, stride, pad, 1};
  HIP_TRY(hifihr::launch_conv_igemm(g, dy, wt_scratch, nullptr, dx, nullptr, ws, ws_bytes, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv2d_bwd_data_pre(const float* dy, const float* wt, float* dx, int N, int H, int W, int C, int K, int R, int S, int stride,
                               int pad, void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !wt || !dx || !conv_dims_ok(N, H, W, C, K, R, S, stride, pad) || K % 4 || (K % 16 && stride != 1))
    return fail(HIFIHR_EINVAL, "hifihr_conv2d_bwd_data_pre: bad argument (K % 4 == 0; K % 16 == 0 when stride > 1)");
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
  hifihr::ConvGeom g{N, OH, OW, K, H, W, C, R, S, stride, pad, 1};
  HIP_TRY(hifihr::launch_conv_igemm(g, dy, wt, nullptr, dx, nullptr, ws, ws_bytes, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv2d_bwd_data_pre_res(const float* dy, const float* wt, const float* res, float* dx, int N, int H, int W, int C, int K, int R, int S,
                                   int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !wt || !dx || !res || !conv_dims_ok(N, H, W, C, K, R, S, stride, pad) || K % 4 || (K % 16 && stride != 1))
    return fail(HIFIHR_EINVAL, "hifihr_conv2d_bwd_data_pre_res: bad argument (K % 4 == 0; K % 16 == 0 when stride > 1)");
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
  hifihr::ConvGeom g{N, OH, OW, K, H, W, C, R, S, stride, pad, 1};
  g.residual = res;
  HIP_TRY(hifihr::launch_conv_igemm(g, dy, wt, nullptr, dx, nullptr, ws, ws_bytes, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv2d_bwd_data_pre_plus1x1_supported(int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
  // the fast (non-generic) gather of conv_igemm_kernel on parity classes: K % 16 == 0; a stride that makes classes; pad < the filter; the
  // 1x1 / same stride / pad 0 convolution's outputs are the 3x3's (same OH x OW)
  if (!conv_dims_ok(N, H, W, C, K, R, S, stride, pad) || stride < 2 || K % 16 || C % 4 || R * S > 62) return 0;
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
  if (OH != (H - 1) / stride + 1 || OW != (W - 1) / stride + 1) return 0;
  if ((long)N * OH * OW * K >= (1L << 30) || (long)C * R * S * K >= (1L << 30)) return 0;
  static const int on = [] { const char* e = getenv("HIFIHR_DGRAD_PLUS1X1"); return e ? atoi(e) : 1; }();
  return on;
}

int hifihr_conv2d_bwd_data_pre_plus1x1(const float* dy, const float* wt, const float* dy2, const float* wt2, float* dx, int N, int H, int W, int C,
                                       int K, int R, int S, int stride, int pad, void* stream) {
  if (!dy || !wt || !dy2 || !wt2 || !dx || !hifihr_conv2d_bwd_data_pre_plus1x1_supported(N, H, W, C, K, R, S, stride, pad))
    return fail(HIFIHR_EINVAL, "hifihr_conv2d_bwd_data_pre_plus1x1: bad argument (see hifihr_conv2d_bwd_data_pre_plus1x1_supported)");
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - S) / stride + 1;
  hifihr::ConvGeom g{N, OH, OW, K, H, W, C, R, S, stride, pad, 1};
  g.src2 = dy2; g.wgt2 = wt2;
  HIP_TRY(hifihr::launch_conv_igemm(g, dy, wt, nullptr, dx, nullptr, nullptr, 0, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv2d_bwd_weight_plus1x1_supported(int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
  if (!conv_dims_ok(N, H, W, C, K, R, S, stride, pad) || C % 4 || K % 4) return 0;
  hifihr::ConvGeom g{N, H, W, C, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K, R, S, stride, pad, 0};
  return hifihr::conv_wgrad_plus1x1_supported(g) ? 1 : 0;
}

int hifihr_conv2d_bwd_weight_plus1x1(const float* x, const float* dy, float* dw, const float* dy2, float* dw2, int N, int H, int W, int C, int K,
                                     int R, int S, int stride, int pad, void* stream) {
  if (!x || !dy || !dw || !dy2 || !dw2 || !hifihr_conv2d_bwd_weight_plus1x1_supported(N, H, W, C, K, R, S, stride, pad))
    return fail(HIFIHR_EINVAL, "hifihr_conv2d_bwd_weight_plus1x1: bad argument (see hifihr_conv2d_bwd_weight_plus1x1_supported)");
  hifihr::ConvGeom g{N, H, W, C, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K, R, S, stride, pad, 0};
  HIP_TRY(hifihr::launch_conv_wgrad_plus1x1(g, x, dy, dw, dy2, dw2, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_weight_prep(const hifihr_prep_job* jobs, int njobs, int blocks_per_job, void* stream) {
  static_assert(sizeof(hifihr_prep_job) == sizeof(hifihr::PrepJob), "hifihr_prep_job layout");
  if (!jobs || njobs <= 0 || blocks_per_job <= 0) return fail(HIFIHR_EINVAL, "hifihr_weight_prep: bad argument");
  HIP_TRY(hifihr::launch_weight_prep(reinterpret_cast<const hifihr::PrepJob*>(jobs), njobs, blocks_per_job, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv2d_bwd_weight(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                             int stride, int pad, void* stream) {
  return hifihr_conv2d_bwd_weight_ws(x, dy, dw, N, H, W, C, K, R, S, stride, pad, nullptr, 0, stream);
}

size_t hifihr_conv2d_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad) {
  if (!conv_dims_ok(N, H, W, C, K, R, S, stride, pad) || C % 4 || K % 4) return 0;
  hifihr::ConvGeom g{N, H, W, C, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K, R, S, stride, pad, 0};
  return hifihr::conv_wgrad_workspace_bytes(g);
}

int hifihr_conv2d_bwd_weight_ws(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                                int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !dy || !dw || !conv_dims_ok(N, H, W, C, K, R, S, stride, pad) || C % 4 || K % 4)
    return fail(HIFIHR_EINVAL, "hifihr_conv2d_bwd_weight: bad argument (C and K must be multiples of 4)");
  hifihr::ConvGeom g{N, H, W, C, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K, R, S, stride, pad, 0};
  HIP_TRY(hifihr::launch_conv_wgrad(g, x, dy, dw, ws, ws ? ws_bytes : 0, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv2d_bwd_weight_c3_supported(int N, int H, int W, int K, int R, int S, int stride, int pad) {
  if (!conv_dims_ok(N, H, W, 4, K, R, S, stride, pad) || K % 4) return 0;
  hifihr::ConvGeom g{N, H, W, 4, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K, R, S, stride, pad, 0};
  return hifihr::conv_wgrad_c3_supported(g) ? 1 : 0;
}

int hifihr_conv2d_bwd_weight_c3(const float* x4, const float* dy, float* dw3, int N, int H, int W, int K, int R, int S, int stride, int pad,
                                void* ws, size_t ws_bytes, void* stream) {
  if (!x4 || !dy || !dw3 || !ws || !hifihr_conv2d_bwd_weight_c3_supported(N, H, W, K, R, S, stride, pad))
    return fail(HIFIHR_EINVAL, "hifihr_conv2d_bwd_weight_c3: only the 7x7 / stride 2 / 64-filter stem on NHWC4 images, with the workspace of "
                               "hifihr_conv2d_wgrad_workspace_bytes (ask hifihr_conv2d_bwd_weight_c3_supported)");
  hifihr::ConvGeom g{N, H, W, 4, (H + 2 * pad - R) / stride + 1, (W + 2 * pad - S) / stride + 1, K, R, S, stride, pad, 0};
  HIP_TRY(hifihr::launch_conv_wgrad_c3(g, x4, dy, dw3, ws, ws_bytes, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_image_to_nhwc4(const float* images, float* out, int B, int H, int W, void* stream) {
  if (!images || !out || B <= 0 || H <= 0 || W <= 0) return fail(HIFIHR_EINVAL, "hifihr_image_to_nhwc4: bad argument");
  HIP_TRY(hifihr::launch_image_to_nhwc4(images, out, B, H, W, H, W, 0, 0, 1, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_image_to_nhwc4_padded(const float* images, float* out, int B, int H, int W, int pad_top, int pad_left, int pad_bottom,
                                 int pad_right, int normalize, void* stream) {
  if (!images || !out || B <= 0 || H <= 0 || W <= 0 || pad_top < 0 || pad_left < 0 || pad_bottom < 0 || pad_right < 0)
    return fail(HIFIHR_EINVAL, "hifihr_image_to_nhwc4_padded: bad argument");
  HIP_TRY(hifihr::launch_image_to_nhwc4(images, out, B, H, W, H + pad_top + pad_bottom, W + pad_left + pad_right, pad_top, pad_left,
                                        normalize ? 1 : 0, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_ssim_partial_count(int planes, int H, int W) {
  if (planes <= 0 || H <= 0 || W <= 0) return 0;
  const int t = hifihr::ssim_tile_edge();
  return planes * ((H + t - 1) / t) * ((W + t - 1) / t);
}

int hifihr_ssim_fwd(const float* window11, const float* img1, const float* img2, int planes, int H, int W, float* partial,
                    float* dA, float* dB, float* dC, void* stream) {
  if (!window11 || !img1 || !img2 || !partial || planes <= 0 || H <= 0 || W <= 0 || ((dA != nullptr) != (dB != nullptr)) ||
      ((dA != nullptr) != (dC != nullptr)))
    return fail(HIFIHR_EINVAL, "hifihr_ssim_fwd: bad argument");
  hifihr::SsimWindow win;
  for (int k = 0; k < 11; ++k) win.g[k] = window11[k];
  HIP_TRY(hifihr::launch_ssim_fwd(win, img1, img2, planes, H, W, partial, dA, dB, dC, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_ssim_bwd(const float* window11, const float* img1, const float* img2, const float* dA, const float* dB, const float* dC,
                    const float* grad_out, int planes, int H, int W, float* gimg1, void* stream) {
  if (!window11 || !img1 || !img2 || !dA || !dB || !dC || !grad_out || !gimg1 || planes <= 0 || H <= 0 || W <= 0)
    return fail(HIFIHR_EINVAL, "hifihr_ssim_bwd: bad argument");
  hifihr::SsimWindow win;
  for (int k = 0; k < 11; ++k) win.g[k] = window11[k];
  HIP_TRY(hifihr::launch_ssim_bwd(win, img1, img2, dA, dB, dC, grad_out, planes, H, W, gimg1, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_ssim_bwd_scaled(const float* window11, const float* img1, const float* img2, const float* dA, const float* dB, const float* dC,
                           const float* grad_out, float out_scale, int planes, int H, int W, float* gimg1, void* stream) {
  if (!window11 || !img1 || !img2 || !dA || !dB || !dC || !grad_out || !gimg1 || planes <= 0 || H <= 0 || W <= 0)
    return fail(HIFIHR_EINVAL, "hifihr_ssim_bwd_scaled: bad argument");
  hifihr::SsimWindow win;
  for (int k = 0; k < 11; ++k) win.g[k] = window11[k];
  HIP_TRY(hifihr::launch_ssim_bwd(win, img1, img2, dA, dB, dC, grad_out, planes, H, W, gimg1, (hipStream_t)stream, out_scale));
  return HIFIHR_OK;
}

int hifihr_ssim_finish(const float* partial, int count, float scale, float offset, float* out, void* stream) {
  if (!partial || !out || count <= 0) return fail(HIFIHR_EINVAL, "hifihr_ssim_finish: bad argument");
  HIP_TRY(hifihr::launch_ssim_finish(partial, count, scale, offset, out, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_bn_stats_floats(int C) { return C > 0 ? hifihr::stat_buffer_floats(C) : 0; }   // double slots + 64 counter words (hifihr_internal.h)

static int bn_dims_ok(long M, int C) { return M > 0 && C >= 4 && C % 4 == 0 && C <= 4096; }

int hifihr_bn_stats(const float* x, long M, int C, float* stats, void* stream) {
  if (!x || !stats || !bn_dims_ok(M, C)) return fail(HIFIHR_EINVAL, "hifihr_bn_stats: bad argument (C % 4 == 0, C <= 4096)");
  HIP_TRY(hifihr::launch_bn_stats(x, M, C, stats, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_bn_act_fwd(const float* x, float* stats, const float* gamma, const float* beta, const float* residual, int act,
                      long M, int C, float eps, float momentum, float* y, float* save_mean, float* save_invstd,
                      float* running_mean, float* running_var, void* stream) {
  if (!x || !stats || !gamma || !beta || !y || !save_mean || !save_invstd || !bn_dims_ok(M, C) ||
      ((running_mean != nullptr) != (running_var != nullptr)))
    return fail(HIFIHR_EINVAL, "hifihr_bn_act_fwd: bad argument (C % 4 == 0, C <= 4096)");
  if (act < 0 || act > 2 || (act == 2 && residual)) return fail(HIFIHR_EINVAL, "hifihr_bn_act_fwd: act must be 0/1/2 (swish takes no residual)");
  HIP_TRY(hifihr::launch_bn_act_fwd(x, stats, gamma, beta, residual, act, M, C, eps, momentum, y, save_mean, save_invstd,
                                    running_mean, running_var, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_bn_act_eval(const float* x, const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                       const float* residual, int act, long M, int C, float eps, float* y, void* stream) {
  if (!x || !running_mean || !running_var || !gamma || !beta || !y || !bn_dims_ok(M, C) || act < 0 || act > 2 || (act == 2 && residual))
    return fail(HIFIHR_EINVAL, "hifihr_bn_act_eval: bad argument (C % 4 == 0, C <= 4096; swish takes no residual)");
  HIP_TRY(hifihr::launch_bn_act_eval(x, running_mean, running_var, gamma, beta, residual, act, M, C, eps, y, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_bn_act_bwd(const float* dy, const float* y, const float* x, const float* save_mean, const float* save_invstd,
                      const float* gamma, const float* beta, int act, long M, int C, float* red_scratch, float* dx, float* dres,
                      float* dgamma_acc, float* dbeta_acc, void* stream) {
  if (!dy || !x || !save_mean || !save_invstd || !gamma || !red_scratch || !dx || !bn_dims_ok(M, C) || (act == 1 && !y && !beta) ||
      (act == 2 && !beta) || act < 0 || act > 2)
    return fail(HIFIHR_EINVAL, "hifihr_bn_act_bwd: bad argument (C % 4 == 0, C <= 4096; act 1 needs y or beta, act 2 needs beta)");
  HIP_TRY(hifihr::launch_bn_act_bwd(dy, y, x, save_mean, save_invstd, gamma, beta, act, M, C, red_scratch, dx, dres, dgamma_acc,
                                    dbeta_acc, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_bn_relu_maxpool_supported(int N, int H, int W, int C) { return hifihr::bn_relu_maxpool_supported(N, H, W, C) ? 1 : 0; }

int hifihr_bn_relu_maxpool_fwd(const float* x, float* stats, const float* gamma, const float* beta, int N, int H, int W, int C, float eps,
                               float momentum, float* y, unsigned char* tap, float* save_mean, float* save_invstd, float* running_mean,
                               float* running_var, void* stream) {
  if (!x || !stats || !gamma || !beta || !y || !tap || !save_mean || !save_invstd || !hifihr::bn_relu_maxpool_supported(N, H, W, C))
    return fail(HIFIHR_EINVAL, "hifihr_bn_relu_maxpool_fwd: bad argument (C % 4 == 0, C <= 512, H, W >= 2)");
  HIP_TRY(hifihr::launch_bn_relu_maxpool_fwd(x, stats, gamma, beta, N, H, W, C, eps, momentum, y, tap, save_mean, save_invstd, running_mean,
                                             running_var, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_bn_relu_maxpool_bwd(const float* gy, const unsigned char* tap, const float* x, const float* save_mean, const float* save_invstd,
                               const float* gamma, const float* beta, int N, int H, int W, int C, float* red_scratch, float* dx,
                               float* dgamma_acc, float* dbeta_acc, void* stream) {
  if (!gy || !tap || !x || !save_mean || !save_invstd || !gamma || !beta || !red_scratch || !dx ||
      !hifihr::bn_relu_maxpool_supported(N, H, W, C))
    return fail(HIFIHR_EINVAL, "hifihr_bn_relu_maxpool_bwd: bad argument (C % 4 == 0, C <= 512, H, W >= 2)");
  HIP_TRY(hifihr::launch_bn_relu_maxpool_bwd(gy, tap, x, save_mean, save_invstd, gamma, beta, N, H, W, C, red_scratch, dx, dgamma_acc,
                                             dbeta_acc, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_bn_relu_maxpool_bwd_y(const float* gy, const float* pooled, const unsigned char* tap, const float* x, const float* save_mean,
                                 const float* save_invstd, const float* gamma, const float* beta, int N, int H, int W, int C, float* red_scratch,
                                 float* dx, float* dgamma_acc, float* dbeta_acc, void* stream) {
  if (!gy || !pooled || !tap || !x || !save_mean || !save_invstd || !gamma || !beta || !red_scratch || !dx ||
      !hifihr::bn_relu_maxpool_supported(N, H, W, C))
    return fail(HIFIHR_EINVAL, "hifihr_bn_relu_maxpool_bwd_y: bad argument (C % 4 == 0, C <= 512, H, W >= 2)");
  HIP_TRY(hifihr::launch_bn_relu_maxpool_bwd(gy, tap, x, save_mean, save_invstd, gamma, beta, N, H, W, C, red_scratch, dx, dgamma_acc,
                                             dbeta_acc, (hipStream_t)stream, pooled));
  return HIFIHR_OK;
}

static int dw_ok(int N, int H, int W, int C, int OH, int OW, int K, int stride, int pt, int pl) {
  return N > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0 && OH > 0 && OW > 0 && (K == 3 || K == 5) && (stride == 1 || stride == 2) && pt >= 0 && pl >= 0;
}

int hifihr_dwconv2d_fwd(const float* x, const float* w, float* y, float* stats, int N, int H, int W, int C, int OH, int OW, int K,
                        int stride, int pad_top, int pad_left, void* stream) {
  if (!x || !w || !y || !dw_ok(N, H, W, C, OH, OW, K, stride, pad_top, pad_left)) return fail(HIFIHR_EINVAL, "hifihr_dwconv2d_fwd: bad argument");
  hifihr::DwGeom g{N, H, W, C, OH, OW, K, stride, pad_top, pad_left};
  HIP_TRY(hifihr::launch_dwconv_fwd(g, x, w, y, stats, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_dwconv2d_fwd_bnswish(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                const float* w, float* y, float* stats, int N, int H, int W, int C, int OH, int OW, int K, int stride,
                                int pad_top, int pad_left, void* stream) {
  if (!x || !mean || !invstd || !gamma || !beta || !w || !y || !dw_ok(N, H, W, C, OH, OW, K, stride, pad_top, pad_left))
    return fail(HIFIHR_EINVAL, "hifihr_dwconv2d_fwd_bnswish: bad argument");
  hifihr::DwGeom g{N, H, W, C, OH, OW, K, stride, pad_top, pad_left};
  HIP_TRY(hifihr::launch_dwconv_fwd(g, x, w, y, stats, (hipStream_t)stream, mean, invstd, gamma, beta));
  return HIFIHR_OK;
}

int hifihr_dwconv2d_bwd_weight_bnswish(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                       const float* dy, float* dw, int N, int H, int W, int C, int OH, int OW, int K, int stride,
                                       int pad_top, int pad_left, void* stream) {
  if (!x || !mean || !invstd || !gamma || !beta || !dy || !dw || !dw_ok(N, H, W, C, OH, OW, K, stride, pad_top, pad_left))
    return fail(HIFIHR_EINVAL, "hifihr_dwconv2d_bwd_weight_bnswish: bad argument");
  hifihr::DwGeom g{N, H, W, C, OH, OW, K, stride, pad_top, pad_left};
  HIP_TRY(hifihr::launch_dwconv_bwd_weight(g, x, dy, dw, (hipStream_t)stream, mean, invstd, gamma, beta));
  return HIFIHR_OK;
}

int hifihr_bn_finalize_fwd(float* stats, long M, int C, float eps, float momentum, float* save_mean, float* save_invstd,
                           float* running_mean, float* running_var, void* stream) {
  if (!stats || !save_mean || !save_invstd || M <= 0 || C < 4 || C % 4 != 0 || (running_mean == nullptr) != (running_var == nullptr))
    return fail(HIFIHR_EINVAL, "hifihr_bn_finalize_fwd: bad argument");
  HIP_TRY(hifihr::launch_bn_finalize_fwd(stats, M, C, eps, momentum, save_mean, save_invstd, running_mean, running_var, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_dwconv2d_bwd_data(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int OH, int OW, int K, int stride,
                             int pad_top, int pad_left, void* stream) {
  if (!dy || !w || !dx || !dw_ok(N, H, W, C, OH, OW, K, stride, pad_top, pad_left)) return fail(HIFIHR_EINVAL, "hifihr_dwconv2d_bwd_data: bad argument");
  hifihr::DwGeom g{N, H, W, C, OH, OW, K, stride, pad_top, pad_left};
  HIP_TRY(hifihr::launch_dwconv_bwd_data(g, dy, w, dx, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_dwconv2d_bwd_weight(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int OH, int OW, int K,
                               int stride, int pad_top, int pad_left, void* stream) {
  if (!x || !dy || !dw || !dw_ok(N, H, W, C, OH, OW, K, stride, pad_top, pad_left)) return fail(HIFIHR_EINVAL, "hifihr_dwconv2d_bwd_weight: bad argument");
  hifihr::DwGeom g{N, H, W, C, OH, OW, K, stride, pad_top, pad_left};
  HIP_TRY(hifihr::launch_dwconv_bwd_weight(g, x, dy, dw, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_mmpool_fwd(const float* x, const float* p, int B, int HW, int C, float* y, int* argmax, float* xmax, float* xavg,
                      void* stream) {
  if (!x || !p || !y || !argmax || !xmax || !xavg || B <= 0 || HW <= 0 || C < 4 || C % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_mmpool_fwd: bad argument (C % 4 == 0)");
  HIP_TRY(hifihr::launch_mmpool_fwd(x, p, B, HW, C, y, argmax, xmax, xavg, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_mmpool_bwd(const float* gy, const float* p, const int* argmax, const float* xmax, const float* xavg, int B, int HW, int C,
                      float* dx, float* dp_acc, void* stream) {
  if (!gy || !p || !argmax || !xmax || !xavg || !dx || B <= 0 || HW <= 0 || C < 4 || C % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_mmpool_bwd: bad argument (C % 4 == 0)");
  HIP_TRY(hifihr::launch_mmpool_bwd(gy, p, argmax, xmax, xavg, B, HW, C, dx, dp_acc, (hipStream_t)stream));
  return HIFIHR_OK;
}

static int pool_ok(int k, int s, int p) { return (k == 3 && s == 2 && p == 1) || (k == 3 && s == 1 && p == 1) || (k == 2 && s == 2 && p == 0); }

int hifihr_maxpool2d_fwd(const float* x, int N, int H, int W, int C, int k, int s, int p, float* y, unsigned char* tap, void* stream) {
  if (!x || !y || !tap || N <= 0 || H <= 0 || W <= 0 || C < 4 || C % 4 != 0 || !pool_ok(k, s, p) || H + 2 * p < k || W + 2 * p < k)
    return fail(HIFIHR_EINVAL, "hifihr_maxpool2d_fwd: bad argument (C % 4 == 0; (k,s,p) in {(3,2,1), (3,1,1), (2,2,0)})");
  HIP_TRY(hifihr::launch_maxpool_fwd(x, N, H, W, C, k, s, p, y, tap, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_maxpool2d_bwd(const float* gy, const unsigned char* tap, int N, int H, int W, int C, int k, int s, int p, float* dx,
                         void* stream) {
  if (!gy || !tap || !dx || N <= 0 || H <= 0 || W <= 0 || C < 4 || C % 4 != 0 || !pool_ok(k, s, p) || H + 2 * p < k || W + 2 * p < k)
    return fail(HIFIHR_EINVAL, "hifihr_maxpool2d_bwd: bad argument (C % 4 == 0; (k,s,p) in {(3,2,1), (3,1,1), (2,2,0)})");
  HIP_TRY(hifihr::launch_maxpool_bwd(gy, tap, nullptr, N, H, W, C, k, s, p, dx, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_maxpool2d_fwd_flat(const float* x, int N, int H, int W, int C, int k, int s, int p, float* y_flat, unsigned char* tap, void* stream) {
  if (!x || !y_flat || !tap || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_maxpool2d_fwd_flat: bad argument (C % 4 == 0; (k,s,p) in {(3,2,1), (3,1,1), (2,2,0)})");
  HIP_TRY(hifihr::launch_maxpool_flat(x, tap, N, H, W, C, k, s, p, y_flat, 0, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_maxpool2d_bwd_flat(const float* gy_flat, const unsigned char* tap, int N, int H, int W, int C, int k, int s, int p, float* dx, void* stream) {
  if (!gy_flat || !dx || !tap || N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_maxpool2d_bwd_flat: bad argument (C % 4 == 0; (k,s,p) in {(3,2,1), (3,1,1), (2,2,0)})");
  HIP_TRY(hifihr::launch_maxpool_flat(gy_flat, const_cast<unsigned char*>(tap), N, H, W, C, k, s, p, dx, 1, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_maxpool2d_bwd_relu(const float* gy, const unsigned char* tap, const float* y, int N, int H, int W, int C, int k, int s, int p,
                              float* dx, void* stream) {
  if (!gy || !tap || !y || !dx || N <= 0 || H <= 0 || W <= 0 || C < 4 || C % 4 != 0 || !pool_ok(k, s, p) || H + 2 * p < k || W + 2 * p < k)
    return fail(HIFIHR_EINVAL, "hifihr_maxpool2d_bwd_relu: bad argument (C % 4 == 0; (k,s,p) in {(3,2,1), (3,1,1), (2,2,0)})");
  HIP_TRY(hifihr::launch_maxpool_bwd(gy, tap, y, N, H, W, C, k, s, p, dx, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_bias_relu_bwd(const float* dy, const float* y, long M, int C, float* g, float* db_acc, void* stream) {
  if (!dy || !y || !g || M <= 0 || C < 4 || C % 4 != 0 || C > 256)
    return fail(HIFIHR_EINVAL, "hifihr_bias_relu_bwd: bad argument (C % 4 == 0, C <= 256)");
  HIP_TRY(hifihr::launch_bias_relu_bwd(dy, y, M, C, g, db_acc, (hipStream_t)stream));
  return HIFIHR_OK;
}

static int geom_args(hifihr::GeomLossArgs& a, const float* joints, const float* joints_gt, const float* verts, const float* verts_gt,
                     const float* shape, const float* pose, const int32_t* faces, int B, int J, int V, int F, int NS, int NP, int mse,
                     const float* lambda5) {
  if (!joints || !joints_gt || !verts || !verts_gt || !lambda5 || B <= 0 || J <= 0 || V <= 0 || F < 0 || NS < 0 || NP < 0 ||
      (F > 0 && !faces) || (NS > 0 && !shape) || (NP > 0 && !pose))
    return 0;
  a.joints = joints; a.joints_gt = joints_gt; a.verts = verts; a.verts_gt = verts_gt; a.shape = shape; a.pose = pose;
  a.faces = F > 0 ? faces : nullptr; a.vf_off = nullptr; a.vf_idx = nullptr;
  a.B = B; a.J = J; a.V = V; a.F = F; a.NS = NS; a.NP = NP; a.mse = mse ? 1 : 0;
  for (int k = 0; k < 5; ++k) a.lambda[k] = lambda5[k];
  return 1;
}

int hifihr_geom_loss_fwd(const float* joints, const float* joints_gt, const float* verts, const float* verts_gt, const float* shape,
                         const float* pose, const int32_t* faces, int B, int J, int V, int F, int NS, int NP, int mse,
                         const float* lambda5, float* partial, float* out, void* stream) {
  hifihr::GeomLossArgs a;
  if (!partial || !out || !geom_args(a, joints, joints_gt, verts, verts_gt, shape, pose, faces, B, J, V, F, NS, NP, mse, lambda5))
    return fail(HIFIHR_EINVAL, "hifihr_geom_loss_fwd: bad argument");
  HIP_TRY(hifihr::launch_geom_loss_fwd(a, partial, out, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_joint_terms_fwd(const float* j2d, const float* j2d_gt, const float* joints, const float* joints_gt, int B, int J, int mse,
                           const float* lam3_host, float* out3, void* stream) {
  if ((!j2d && !joints) || (j2d && !j2d_gt) || (joints && !joints_gt) || !lam3_host || !out3 || B <= 0 || J != 21)
    return fail(HIFIHR_EINVAL, "hifihr_joint_terms_fwd: bad argument (J must be 21)");
  HIP_TRY(hifihr::launch_joint_terms_fwd(j2d, j2d_gt, joints, joints_gt, B, J, mse, lam3_host, out3, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_joint_terms_bwd(const float* j2d, const float* j2d_gt, const float* joints, const float* joints_gt, int B, int J, int mse,
                           const float* lam3_host, const float* gout3, float* g_j2d, float* g_joints, void* stream) {
  if ((!j2d && !joints) || (j2d && !j2d_gt) || (joints && !joints_gt) || !lam3_host || !gout3 || B <= 0 || J != 21 || (g_j2d && !j2d) ||
      (g_joints && !joints))
    return fail(HIFIHR_EINVAL, "hifihr_joint_terms_bwd: bad argument (J must be 21)");
  HIP_TRY(hifihr::launch_joint_terms_bwd(j2d, j2d_gt, joints, joints_gt, B, J, mse, lam3_host, gout3, g_j2d, g_joints, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_geom_loss_bwd(const float* joints, const float* joints_gt, const float* verts, const float* verts_gt, const float* shape,
                         const float* pose, const int32_t* faces, const int32_t* vf_off, const int32_t* vf_idx, int B, int J, int V,
                         int F, int NS, int NP, int mse, const float* lambda5, const float* gout, float* gj, float* gv, float* gshape,
                         float* gpose, void* stream) {
  hifihr::GeomLossArgs a;
  if (!gout || (F > 0 && (!vf_off || !vf_idx)) ||
      !geom_args(a, joints, joints_gt, verts, verts_gt, shape, pose, faces, B, J, V, F, NS, NP, mse, lambda5))
    return fail(HIFIHR_EINVAL, "hifihr_geom_loss_bwd: bad argument");
  a.vf_off = vf_off; a.vf_idx = vf_idx;
  HIP_TRY(hifihr::launch_geom_loss_bwd(a, gout, gj, gv, gshape, gpose, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_photo_loss_partial_floats(void) { return hifihr::photo_loss_partial_floats(); }

int hifihr_photo_loss_fwd(const float* rgba, const float* imgs, const int64_t* seg, int B, int H, int W, float l_tex, float l_mrgb,
                          float l_sil, float* re_img_m, float* mask_rgbs, float* partial, float* out, void* stream) {
  if (!rgba || !imgs || !seg || !re_img_m || !mask_rgbs || !partial || !out || B <= 0 || H <= 0 || W <= 0 || (H * W) % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_photo_loss_fwd: bad argument (H*W % 4 == 0)");
  HIP_TRY(hifihr::launch_photo_loss_fwd(rgba, imgs, (const long long*)seg, B, H * W, l_tex, l_mrgb, l_sil, re_img_m, mask_rgbs, partial,
                                        out, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_photo_loss_bwd(const float* rgba, const float* re_img_m, const float* mask_rgbs, const float* g_re_img, const float* gout,
                          const float* fwd_out, int B, int H, int W, float l_tex, float l_mrgb, float* grad_rgba, void* stream) {
  if (!rgba || !re_img_m || !mask_rgbs || !fwd_out || !grad_rgba || B <= 0 || H <= 0 || W <= 0 || (H * W) % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_photo_loss_bwd: bad argument (H*W % 4 == 0)");
  HIP_TRY(hifihr::launch_photo_loss_bwd(rgba, re_img_m, mask_rgbs, g_re_img, gout, fwd_out, B, H * W, l_tex, l_mrgb, grad_rgba,
                                        (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_sil_post(const float* rgba, const float* imgs, int B, int H, int W, float* re_sil, float* mask_rgbs, void* stream) {
  if (!rgba || !re_sil || (mask_rgbs && !imgs) || B <= 0 || H <= 0 || W <= 0 || (H * W) % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_sil_post: bad argument (H*W % 4 == 0)");
  HIP_TRY(hifihr::launch_sil_post(rgba, imgs, B, H * W, re_sil, mask_rgbs, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_light_split_fwd(const float* lights, int B, float* colors, float* directions, void* stream) {
  if (!lights || !colors || !directions || B <= 0) return fail(HIFIHR_EINVAL, "hifihr_light_split_fwd: bad argument");
  HIP_TRY(hifihr::launch_light_split_fwd(lights, B, colors, directions, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_light_split_bwd(const float* lights, const float* gcolors, const float* gdirections, int B, float* glights, void* stream) {
  if (!lights || !glights || B <= 0) return fail(HIFIHR_EINVAL, "hifihr_light_split_bwd: bad argument");
  HIP_TRY(hifihr::launch_light_split_bwd(lights, gcolors, gdirections, B, glights, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_loss_total_fwd(const float* const* parts, const int* counts, int nparts, float* total, void* stream) {
  if (!parts || !counts || !total || nparts < 1 || nparts > hifihr::kLossTotalParts) return fail(HIFIHR_EINVAL, "hifihr_loss_total_fwd: 1 .. 4 parts");
  hifihr::LossTotalParts p{};
  for (int i = 0; i < nparts; ++i) {
    if (!parts[i] || counts[i] < 0 || counts[i] > 64) return fail(HIFIHR_EINVAL, "hifihr_loss_total_fwd: bad part");
    p.v[i] = parts[i]; p.n[i] = counts[i];
  }
  HIP_TRY(hifihr::launch_loss_total_fwd(p, total, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_loss_total_bwd(const float* gtotal, float* const* grads, const int* counts, const int* lengths, int nparts, void* stream) {
  if (!gtotal || !grads || !counts || !lengths || nparts < 1 || nparts > hifihr::kLossTotalParts)
    return fail(HIFIHR_EINVAL, "hifihr_loss_total_bwd: 1 .. 4 parts");
  hifihr::LossTotalGrads q{};
  for (int i = 0; i < nparts; ++i) {
    if (!grads[i] || counts[i] < 0 || lengths[i] < counts[i] || lengths[i] > 64) return fail(HIFIHR_EINVAL, "hifihr_loss_total_bwd: bad part");
    q.g[i] = grads[i]; q.n[i] = counts[i]; q.len[i] = lengths[i];
  }
  HIP_TRY(hifihr::launch_loss_total_bwd(gtotal, q, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_linear_fwd(const float* x, const float* w, const float* b, int B, int I, int O, int act, const float* gamma, const float* beta,
                      float eps, float momentum, float* running_mean, float* running_var, float* y, float* z, float* save_mean,
                      float* save_invstd, void* stream) {
  if (!x || !w || !y || B <= 0 || I <= 0 || O <= 0 || act < 0 || act > 3 || (act >= 2 && gamma) || (act == 2 && !z))
    return fail(HIFIHR_EINVAL, "hifihr_linear_fwd: bad argument (act 0..3; swish / sigmoid take no batch-norm; swish needs z)");
  if (gamma && (!beta || !z || !save_mean || !save_invstd || B > 64 || ((running_mean != nullptr) != (running_var != nullptr))))
    return fail(HIFIHR_EINVAL, "hifihr_linear_fwd: batch-norm needs beta, z, save_mean, save_invstd and B <= 64");
  hifihr::LinearArgs a{x, w, b, y, z, gamma, beta, save_mean, save_invstd, running_mean, running_var, eps, momentum, B, I, O, act};
  HIP_TRY(hifihr::launch_linear_fwd(a, (hipStream_t)stream));
  return HIFIHR_OK;
}

static int group_member_ok(const hifihr_linear_desc& d) {
  return d.x && d.w && d.y && d.B > 0 && d.I > 0 && d.O > 0 && (d.act == 0 || d.act == 1);
}

int hifihr_linear_fwd_group(const hifihr_linear_desc* descs, int n, void* stream) {
  if (!descs || n <= 0 || n > hifihr::kMaxLinearGroup) return fail(HIFIHR_EINVAL, "hifihr_linear_fwd_group: 1..6 members");
  hifihr::LinearGroup grp{};
  grp.n = n;
  for (int i = 0; i < n; ++i) {
    const hifihr_linear_desc& d = descs[i];
    if (!group_member_ok(d)) return fail(HIFIHR_EINVAL, "hifihr_linear_fwd_group: bad member (x, w, y, sizes; act 0 / 1)");
    grp.a[i] = hifihr::LinearArgs{d.x, d.w, d.b, d.y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, d.B, d.I, d.O, d.act};
  }
  HIP_TRY(hifihr::launch_linear_fwd_group(grp, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_linear_bwd_group(const hifihr_linear_desc* descs, int n, void* stream) {
  if (!descs || n <= 0 || n > hifihr::kMaxLinearGroup) return fail(HIFIHR_EINVAL, "hifihr_linear_bwd_group: 1..6 members");
  hifihr::LinearGroup grp{};
  hifihr::LinearGradsGroup gg{};
  grp.n = n;
  for (int i = 0; i < n; ++i) {
    const hifihr_linear_desc& d = descs[i];
    if (!group_member_ok(d) || !d.dy || !d.dz_scratch) return fail(HIFIHR_EINVAL, "hifihr_linear_bwd_group: bad member (dy, dz_scratch, y for act 1)");
    grp.a[i] = hifihr::LinearArgs{d.x, d.w, nullptr, d.y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, d.B, d.I, d.O, d.act};
    gg.g[i] = hifihr::LinearGrads{d.dy, d.dz_scratch, d.dW_acc, d.db_acc, nullptr, nullptr, d.dx};
  }
  HIP_TRY(hifihr::launch_linear_bwd_group(grp, gg, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_linear_bwd(const float* dy, const float* y, const float* x, const float* w, int B, int I, int O, int act, const float* gamma,
                      const float* z, const float* save_mean, const float* save_invstd, float* dz_scratch, float* dW_acc, float* db_acc,
                      float* dgamma_acc, float* dbeta_acc, float* dx, void* stream) {
  if (!dy || !x || !w || B <= 0 || I <= 0 || O <= 0 || act < 0 || act > 3 || ((act == 1 || act == 3) && !y) ||
      (act == 2 && !z) || (dx && !dz_scratch))
    return fail(HIFIHR_EINVAL, "hifihr_linear_bwd: bad argument (act 1 / 3 need y, act 2 needs z; dx needs dz_scratch)");
  if (gamma && (!z || !save_mean || !save_invstd || B > 64))
    return fail(HIFIHR_EINVAL, "hifihr_linear_bwd: batch-norm needs z, save_mean, save_invstd and B <= 64");
  hifihr::LinearArgs a{x, w, nullptr, const_cast<float*>(y), const_cast<float*>(z), gamma, nullptr, const_cast<float*>(save_mean),
                       const_cast<float*>(save_invstd), nullptr, nullptr, 0.f, 0.f, B, I, O, act};
  hifihr::LinearGrads g{dy, dz_scratch, dW_acc, db_acc, dgamma_acc, dbeta_acc, dx};
  HIP_TRY(hifihr::launch_linear_bwd(a, g, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_se_pool(const float* x, int B, int HW, int C, float* mean_zeroed, void* stream) {
  if (!x || !mean_zeroed || B <= 0 || HW <= 0 || C < 4 || C % 4 != 0) return fail(HIFIHR_EINVAL, "hifihr_se_pool: bad argument (C % 4 == 0)");
  HIP_TRY(hifihr::launch_se_pool(x, B, HW, C, mean_zeroed, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_se_scale(const float* x, const float* gate, const float* add, float add_scale, int B, int HW, int C, float* y, void* stream) {
  if (!x || !gate || !y || B <= 0 || HW <= 0 || C < 4 || C % 4 != 0) return fail(HIFIHR_EINVAL, "hifihr_se_scale: bad argument (C % 4 == 0)");
  HIP_TRY(hifihr::launch_se_scale(x, gate, add, add_scale, B, HW, C, y, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_se_bwd_gate(const float* dy, const float* x, int B, int HW, int C, float* dgate_zeroed, void* stream) {
  if (!dy || !x || !dgate_zeroed || B <= 0 || HW <= 0 || C < 4 || C % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_se_bwd_gate: bad argument (C % 4 == 0)");
  HIP_TRY(hifihr::launch_se_bwd_gate(dy, x, B, HW, C, dgate_zeroed, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_se_mlp_supported(int C, int SQ) { return hifihr::se_mlp_supported(C, SQ) ? 1 : 0; }

int hifihr_drop_connect_add(const float* x, const float* skip, const float* u, float keep, int B, size_t per_sample, float* out, void* stream) {
  if (!x || !u || !out || B <= 0 || per_sample == 0 || per_sample % 4 != 0 || !(keep > 0.f))
    return fail(HIFIHR_EINVAL, "hifihr_drop_connect_add: bad argument (per_sample % 4 == 0, keep > 0)");
  HIP_TRY(hifihr::launch_drop_connect_add(x, skip, u, keep, B, per_sample, out, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_se_mlp_fwd(float* mean_acc, const float* w1, const float* b1, const float* w2t, const float* b2, int B, int C, int SQ, float* mean,
                      float* z1, float* h1, float* gate, void* stream) {
  if (!mean_acc || !w1 || !b1 || !w2t || !b2 || !mean || !z1 || !h1 || !gate || B <= 0 || !hifihr::se_mlp_supported(C, SQ))
    return fail(HIFIHR_EINVAL, "hifihr_se_mlp_fwd: bad argument (C % 4 == 0, C <= 4096, SQ <= 256)");
  HIP_TRY(hifihr::launch_se_mlp_fwd(mean_acc, w1, b1, w2t, b2, B, C, SQ, mean, z1, h1, gate, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_se_mlp_bwd(float* dgate_acc, const float* gate, const float* z1, const float* h1, const float* mean, const float* w1, const float* w2t,
                      int B, int C, int SQ, float* dz2, float* dz1, float* dmean, float* dw1_acc, float* db1_acc, float* dw2_acc, float* db2_acc,
                      void* stream) {
  if (!dgate_acc || !gate || !z1 || !h1 || !mean || !w1 || !w2t || !dz2 || !dz1 || !dmean || !dw1_acc || !db1_acc || !dw2_acc || !db2_acc ||
      B <= 0 || !hifihr::se_mlp_supported(C, SQ))
    return fail(HIFIHR_EINVAL, "hifihr_se_mlp_bwd: bad argument (C % 4 == 0, C <= 4096, SQ <= 256)");
  HIP_TRY(hifihr::launch_se_mlp_bwd(dgate_acc, gate, z1, h1, mean, w1, w2t, B, C, SQ, dz2, dz1, dmean, dw1_acc, db1_acc, dw2_acc, db2_acc,
                                    (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_weight_transpose(const float* w, float* wt, int K, int RS, int C, void* stream) {
  if (!w || !wt || K <= 0 || RS <= 0 || C <= 0) return fail(HIFIHR_EINVAL, "hifihr_weight_transpose: bad argument");
  HIP_TRY(hifihr::launch_weight_transpose(w, wt, K, RS, C, (hipStream_t)stream));
  return HIFIHR_OK;
}

/* ---- Winograd F(2x2, 3x3) ---- */
static hifihr::ConvGeom wino_gemm_geom(long T, int C, int K) {
  hifihr::ConvGeom g{1, (int)T, 1, C, (int)T, 1, K, 1, 1, 1, 0, 0, 0, 16, T * (long)C, (long)K * C, T * (long)K};
  return g;
}

// HIFIHR_BGEMM=0 keeps the Winograd products on round 1's gather kernels (conv.hip); tuning / A-B runs only
static bool use_bgemm() {
  const char* e = getenv("HIFIHR_BGEMM");
  return e == nullptr || atoi(e) != 0;
}

// Output-tile edge of the Winograd algorithm a layer of this geometry runs: 4 = F(4x4, 3x3) (csrc/wino4.hip: 36 positions, 4x fewer
// multiplications than direct) where the batched GEMMs of csrc/gemm.hip take the shape in both directions, else 2 = F(2x2, 3x3).
// HIFIHR_WINO_M=2 keeps every layer on F(2x2, 3x3).
static int wino_m(int N, int H, int W, int C, int K) {
  static const int pref = [] { const char* e = getenv("HIFIHR_WINO_M"); return e ? atoi(e) : 4; }();
  if (pref != 4 || !use_bgemm() || H < 4 || W < 4) return 2;
  const long T4 = hifihr::wino4_tiles(N, H, W);
  if (T4 >= (1L << 30)) return 2;
  // forward (V U^T: rows T4, N = K, reduction C), backward-data (roles of C and K swapped), backward-weight (slabs of Y'^T V)
  if (!hifihr::bgemm_nt_supported((int)T4, K, C) || !hifihr::bgemm_nt_supported((int)T4, C, K) || !hifihr::bgemm_tn_supported(K, C, (int)T4)) return 2;
  return 4;
}
static long wino_T(int m, int N, int H, int W) { return m == 4 ? hifihr::wino4_tiles(N, H, W) : (long)N * ((H + m - 1) / m) * ((W + m - 1) / m); }

long hifihr_wino_tiles(int N, int H, int W, int m) { return (N > 0 && H > 0 && W > 0 && (m == 2 || m == 4)) ? wino_T(m, N, H, W) : 0; }
long hifihr_wino_tiles_computed(int N, int H, int W, int m) {
  if (!(N > 0 && H > 0 && W > 0 && (m == 2 || m == 4))) return 0;
  return m == 4 ? hifihr::wino4_tiles_real(N, H, W) : wino_T(m, N, H, W);
}

int hifihr_wino_tile(int N, int H, int W, int C, int K) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0) return 2;
  return wino_m(N, H, W, C, K);
}

size_t hifihr_wino_gemm_workspace_bytes_m(int N, int H, int W, int C, int K, int m) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0) return 0;
  if (m == 4) return hifihr::bgemm_nt_workspace_bytes((int)wino_T(4, N, H, W), K, C, 36);
  return hifihr_wino_gemm_workspace_bytes(N, H, W, C, K);
}

size_t hifihr_wino_gemm_workspace_bytes(int N, int H, int W, int C, int K) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0) return 0;
  const long T = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
  if (use_bgemm() && T < (1L << 30) && hifihr::bgemm_nt_supported((int)T, K, C))
    return hifihr::bgemm_nt_workspace_bytes((int)T, K, C, 16);       // the persistent (balanced) kernel's slabs + flags, or 0
  return hifihr::conv_sk_workspace_bytes(wino_gemm_geom(T, C, K));
}

size_t hifihr_bgemm_nt_workspace_bytes(int M, int N, int K, int batch) { return hifihr::bgemm_nt_workspace_bytes(M, N, K, batch); }

int hifihr_bgemm_nt(const float* A, const float* B, float* C, int M, int N, int K, int batch, void* ws, size_t ws_bytes, void* stream) {
  if (!A || !B || !C || batch <= 0 || !hifihr::bgemm_nt_supported(M, N, K))
    return fail(HIFIHR_EINVAL, "hifihr_bgemm_nt: bad argument (K % 32 == 0, N % 64 == 0)");
  HIP_TRY(hifihr::launch_bgemm_nt(A, B, C, M, N, K, batch, ws, ws_bytes, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_bgemm_describe(int tn, int M, int N, int K, char* out, int cap) {
  if (!out || cap < 8) return fail(HIFIHR_EINVAL, "hifihr_bgemm_describe: bad argument");
  if (tn ? !hifihr::bgemm_tn_supported(M, N, K) : !hifihr::bgemm_nt_supported(M, N, K)) { out[0] = 0; return HIFIHR_OK; }
  hifihr::bgemm_describe(tn, M, N, K, out, cap);
  return HIFIHR_OK;
}

int hifihr_bgemm_describe_batch(int tn, int M, int N, int K, int batch, char* out, int cap) {
  if (!out || cap < 8 || batch <= 0) return fail(HIFIHR_EINVAL, "hifihr_bgemm_describe_batch: bad argument");
  if (tn ? !hifihr::bgemm_tn_supported(M, N, K) : !hifihr::bgemm_nt_supported(M, N, K)) { out[0] = 0; return HIFIHR_OK; }
  hifihr::bgemm_describe_batch(tn, M, N, K, batch, out, cap);
  return HIFIHR_OK;
}

int hifihr_bgemm_tn_parts(int M, int N, int T, int batch) {
  if (batch <= 0 || !hifihr::bgemm_tn_supported(M, N, T)) return 0;
  return hifihr::bgemm_tn_parts(M, N, T, batch);
}

int hifihr_bgemm_tn(const float* A, const float* B, float* C_parts, int M, int N, int T, int batch, int parts, void* stream) {
  if (!A || !B || !C_parts || batch <= 0 || parts <= 0 || !hifihr::bgemm_tn_supported(M, N, T))
    return fail(HIFIHR_EINVAL, "hifihr_bgemm_tn: bad argument (M % 64 == 0, N % 64 == 0)");
  HIP_TRY(hifihr::launch_bgemm_tn(A, B, C_parts, M, N, T, batch, parts, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_wgrad_parts_m(int N, int H, int W, int C, int K, int m) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || !use_bgemm()) return 0;
  if (m == 4) {
    const long T4 = wino_T(4, N, H, W);
    return (T4 < (1L << 30) && hifihr::bgemm_tn_supported(K, C, (int)T4)) ? hifihr::bgemm_tn_parts(K, C, (int)T4, 36) : 0;
  }
  return hifihr_wino_wgrad_parts(N, H, W, C, K);
}

int hifihr_wino_wgrad_parts(int N, int H, int W, int C, int K) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || !use_bgemm()) return 0;
  const long T = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
  if (T >= (1L << 30) || !hifihr::bgemm_tn_supported(K, C, (int)T)) return 0;
  return hifihr::bgemm_tn_parts(K, C, (int)T, 16);
}

int hifihr_wino_wgrad_gemm_parts_m(const float* V, const float* Y, float* dU_parts, int N, int H, int W, int C, int K, int parts, int m, void* stream) {
  if (m != 4) return hifihr_wino_wgrad_gemm_parts(V, Y, dU_parts, N, H, W, C, K, parts, stream);
  if (!V || !Y || !dU_parts || N <= 0 || H <= 0 || W <= 0 || parts <= 0)
    return fail(HIFIHR_EINVAL, "hifihr_wino_wgrad_gemm_parts: bad argument");
  const long T4 = wino_T(4, N, H, W);
  if (T4 >= (1L << 30) || !hifihr::bgemm_tn_supported(K, C, (int)T4) || parts != hifihr::bgemm_tn_parts(K, C, (int)T4, 36))
    return fail(HIFIHR_EINVAL, "hifihr_wino_wgrad_gemm_parts: parts must be hifihr_wino_wgrad_parts_m(N, H, W, C, K, 4) > 0");
  // (the rows behind the last tile mosaic are zero in V and Y: the row-share kernel skips their k-steps)
  HIP_TRY(hifihr::launch_bgemm_tn(Y, V, dU_parts, K, C, (int)T4, 36, parts, (hipStream_t)stream, (int)hifihr::wino4_tiles_real(N, H, W)));
  return HIFIHR_OK;
}

int hifihr_wino_wgrad_gemm_parts(const float* V, const float* Y, float* dU_parts, int N, int H, int W, int C, int K, int parts, void* stream) {
  if (!V || !Y || !dU_parts || N <= 0 || H <= 0 || W <= 0 || parts <= 0)
    return fail(HIFIHR_EINVAL, "hifihr_wino_wgrad_gemm_parts: bad argument");
  const long T = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
  if (T >= (1L << 30) || !hifihr::bgemm_tn_supported(K, C, (int)T) || parts != hifihr::bgemm_tn_parts(K, C, (int)T, 16))
    return fail(HIFIHR_EINVAL, "hifihr_wino_wgrad_gemm_parts: parts must be hifihr_wino_wgrad_parts(N, H, W, C, K) > 0");
  HIP_TRY(hifihr::launch_bgemm_tn(Y, V, dU_parts, K, C, (int)T, 16, parts, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_dw_transform_parts_m(const float* dU_parts, int parts, float* dw_acc, int K, int C, int m, void* stream) {
  if (!dU_parts || !dw_acc || parts <= 0 || K <= 0 || C < 4 || C % 4 != 0 || (m != 2 && m != 4))
    return fail(HIFIHR_EINVAL, "hifihr_wino_dw_transform_parts: bad argument");
  if (m == 4) HIP_TRY(hifihr::launch_wino4_dw_transform_parts(dU_parts, parts, dw_acc, K, C, (hipStream_t)stream));
  else HIP_TRY(hifihr::launch_wino_dw_transform_parts(dU_parts, parts, dw_acc, K, C, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino4_dw_transform_multi(const hifihr_wino_dw_job* jobs, int njobs, void* stream) {
  static_assert(sizeof(hifihr_wino_dw_job) == sizeof(hifihr::WinoDwJob), "job layouts");
  if (!jobs || njobs <= 0) return fail(HIFIHR_EINVAL, "hifihr_wino4_dw_transform_multi: bad argument");
  for (int i = 0; i < njobs; ++i)
    if (!jobs[i].du_parts_d || !jobs[i].dw_acc_d || jobs[i].parts < 1 || jobs[i].K <= 0 || jobs[i].C < 4 || jobs[i].C % 4 != 0)
      return fail(HIFIHR_EINVAL, "hifihr_wino4_dw_transform_multi: bad job (pointers, parts >= 1, K > 0, C % 4 == 0)");
  HIP_TRY(hifihr::launch_wino4_dw_transform_multi(reinterpret_cast<const hifihr::WinoDwJob*>(jobs), njobs, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_dw_transform_parts(const float* dU_parts, int parts, float* dw_acc, int K, int C, void* stream) {
  return hifihr_wino_dw_transform_parts_m(dU_parts, parts, dw_acc, K, C, 2, stream);
}

int hifihr_wino_weight_transform_m(const float* w, float* U, int K, int C, int flip, int m, void* stream) {
  if (!w || !U || K <= 0 || C < 4 || C % 4 != 0 || (m != 2 && m != 4)) return fail(HIFIHR_EINVAL, "hifihr_wino_weight_transform: bad argument (C % 4 == 0)");
  if (m == 4) HIP_TRY(hifihr::launch_wino4_weight_transform(w, U, K, C, flip ? 1 : 0, (hipStream_t)stream));
  else HIP_TRY(hifihr::launch_wino_weight_transform(w, U, K, C, flip ? 1 : 0, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_weight_transform(const float* w, float* U, int K, int C, int flip, void* stream) {
  return hifihr_wino_weight_transform_m(w, U, K, C, flip, 2, stream);
}

int hifihr_wino_input_transform_m(const float* x, float* V, int N, int H, int W, int C, int m, void* stream) {
  if (m != 4) return hifihr_wino_input_transform(x, V, N, H, W, C, stream);
  if (!x || !V || N <= 0 || H <= 0 || W <= 0 || C < 4 || C % 4 != 0) return fail(HIFIHR_EINVAL, "hifihr_wino_input_transform: bad argument");
  HIP_TRY(hifihr::launch_wino4_input_transform(x, V, nullptr, N, H, W, C, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_gemm_m(const float* V, const float* U, float* M, int N, int H, int W, int C, int K, int m, void* ws, size_t ws_bytes, void* stream) {
  if (m != 4) return hifihr_wino_gemm(V, U, M, N, H, W, C, K, ws, ws_bytes, stream);
  const long T4 = (N > 0 && H > 0 && W > 0) ? wino_T(4, N, H, W) : 0;
  if (!V || !U || !M || T4 <= 0 || T4 >= (1L << 30) || !hifihr::bgemm_nt_supported((int)T4, K, C))
    return fail(HIFIHR_EINVAL, "hifihr_wino_gemm: bad argument (F(4x4, 3x3) needs C % 32 == 0, K % 64 == 0)");
  const long Tr = hifihr::wino4_tiles_real(N, H, W);       // mosaic tiles: the rows behind the last mosaic are padding (zeros in V, unread in M)
  if (Tr < T4) HIP_TRY(hifihr::launch_bgemm_nt(V, U, M, (int)Tr, K, C, 36, ws, ws_bytes, (hipStream_t)stream, nullptr, (int)T4));
  else HIP_TRY(hifihr::launch_bgemm_nt(V, U, M, (int)T4, K, C, 36, ws, ws_bytes, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino4_bwd_gemm_pair_supported(int N, int H, int W, int C, int K) {
  const long T4 = (N > 0 && H > 0 && W > 0) ? wino_T(4, N, H, W) : 0;
  if (T4 <= 0 || T4 >= (1L << 30) || C <= 0 || K <= 0 || !hifihr::bgemm_nt_supported((int)T4, C, K) || !hifihr::bgemm_tn_supported(K, C, (int)T4)) return 0;
  const long Tr = hifihr::wino4_tiles_real(N, H, W);
  return hifihr::bgemm_nt_tn_pair_supported((int)(Tr < T4 ? Tr : T4), (int)T4, C, K, 36, K, C, (int)T4, 36, hifihr::bgemm_tn_parts(K, C, (int)T4, 36)) ? 1 : 0;
}

int hifihr_wino4_bwd_gemm_pair(const float* V2, const float* U2, float* M2, const float* Vx, const float* Yt, float* dU_parts, int N, int H,
                               int W, int C, int K, int parts, void* stream) {
  const long T4 = (N > 0 && H > 0 && W > 0) ? wino_T(4, N, H, W) : 0;
  if (!V2 || !U2 || !M2 || !Vx || !Yt || !dU_parts || T4 <= 0 || T4 >= (1L << 30) || parts <= 0 || !hifihr::bgemm_nt_supported((int)T4, C, K) ||
      !hifihr::bgemm_tn_supported(K, C, (int)T4) || parts != hifihr::bgemm_tn_parts(K, C, (int)T4, 36))
    return fail(HIFIHR_EINVAL, "hifihr_wino4_bwd_gemm_pair: bad argument (C, K % 64 == 0; parts = hifihr_wino_wgrad_parts_m(N, H, W, C, K, 4))");
  const long Tr = hifihr::wino4_tiles_real(N, H, W);
  const hipError_t e = hifihr::launch_bgemm_nt_tn_pair(V2, U2, M2, (int)(Tr < T4 ? Tr : T4), (int)T4, C, K, 36, Yt, Vx, dU_parts, K, C, (int)T4, 36,
                                                       parts, (hipStream_t)stream, (int)(Tr < T4 ? Tr : T4));
  if (e == hipSuccess) return HIFIHR_OK;
  if (e != hipErrorNotSupported) HIP_TRY(e);
  // not a pair of row-share products: the two launches of hifihr_wino_gemm_m (with C and K exchanged) / hifihr_wino_wgrad_gemm_parts_m
  if (Tr < T4) HIP_TRY(hifihr::launch_bgemm_nt(V2, U2, M2, (int)Tr, C, K, 36, nullptr, 0, (hipStream_t)stream, nullptr, (int)T4));
  else HIP_TRY(hifihr::launch_bgemm_nt(V2, U2, M2, (int)T4, C, K, 36, nullptr, 0, (hipStream_t)stream));
  HIP_TRY(hifihr::launch_bgemm_tn(Yt, Vx, dU_parts, K, C, (int)T4, 36, parts, (hipStream_t)stream, (int)(Tr < T4 ? Tr : T4)));
  return HIFIHR_OK;
}

int hifihr_wino_dy_transform_m(const float* dy, float* Y, int N, int H, int W, int K, int m, void* stream) {
  if (m != 4) return hifihr_wino_dy_transform(dy, Y, N, H, W, K, stream);
  if (!dy || !Y || N <= 0 || H <= 0 || W <= 0 || K < 4 || K % 4 != 0) return fail(HIFIHR_EINVAL, "hifihr_wino_dy_transform: bad argument");
  HIP_TRY(hifihr::launch_wino4_dy_transform(dy, Y, N, H, W, K, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_output_transform_m(const float* M, float* y, float* stats, int N, int H, int W, int K, int m, void* stream) {
  if (m != 4) return hifihr_wino_output_transform(M, y, stats, N, H, W, K, stream);
  if (!M || !y || N <= 0 || H <= 0 || W <= 0 || K < 4 || K % 4 != 0) return fail(HIFIHR_EINVAL, "hifihr_wino_output_transform: bad argument");
  HIP_TRY(hifihr::launch_wino4_output_transform(M, y, stats, nullptr, 0, nullptr, N, H, W, K, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv3x3_c64_wino_supported(int N, int H, int W, int C, int K) { return hifihr::conv_wino2_supported(N, H, W, C, K) ? 1 : 0; }

int hifihr_zero_page_ready(void* stream) { return hifihr::conv_halo_zero_page((hipStream_t)stream) != nullptr ? 1 : 0; }

int hifihr_conv3x3_c64_wino(const float* x, const float* u, const float* bias, int relu, float* y, float* stats, int N, int H, int W, void* stream) {
  if (!x || !u || !y) return fail(HIFIHR_EINVAL, "hifihr_conv3x3_c64_wino: null pointer");
  if (!hifihr::conv_wino2_supported(N, H, W, 64, 64))
    return fail(HIFIHR_EINVAL, "hifihr_conv3x3_c64_wino: needs even H and even W >= 14 (or HIFIHR_CONV_WINO2=0 is set)");
  HIP_TRY(hifihr::launch_conv_wino2(x, u, bias, relu, y, stats, N, H, W, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv3x3_c64_wino_res(const float* x, const float* u, const float* res, float* y, int N, int H, int W, void* stream) {
  if (!x || !u || !y || !res) return fail(HIFIHR_EINVAL, "hifihr_conv3x3_c64_wino_res: null pointer");
  if (!hifihr::conv_wino2_supported(N, H, W, 64, 64))
    return fail(HIFIHR_EINVAL, "hifihr_conv3x3_c64_wino_res: needs even H and even W >= 14 (or HIFIHR_CONV_WINO2=0 is set)");
  HIP_TRY(hifihr::launch_conv_wino2(x, u, nullptr, 0, y, nullptr, N, H, W, (hipStream_t)stream, res));
  return HIFIHR_OK;
}

int hifihr_conv3x3_c64_bwd_pair_supported(int N, int H, int W) { return hifihr::conv_c64_bwd_pair_supported(N, H, W) ? 1 : 0; }

int hifihr_conv3x3_c64_bwd_pair(const float* dy, const float* u_bwd, const float* res, float* dx, const float* x, float* dw, void* ws,
                                size_t ws_bytes, int N, int H, int W, void* stream) {
  if (!dy || !u_bwd || !dx || !x || !dw) return fail(HIFIHR_EINVAL, "hifihr_conv3x3_c64_bwd_pair: null pointer");
  if (!hifihr::conv_c64_bwd_pair_supported(N, H, W))
    return fail(HIFIHR_EINVAL, "hifihr_conv3x3_c64_bwd_pair: unsupported shape (ask hifihr_conv3x3_c64_bwd_pair_supported)");
  HIP_TRY(hifihr::launch_conv_c64_bwd_pair(dy, u_bwd, res, dx, x, dw,
                                           (ws && ws_bytes >= hifihr::conv_halo_wgrad_slab_bytes()) ? static_cast<float*>(ws) : nullptr, N, H, W,
                                           (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_conv3x3_c64_bwd_pair_slabs(const float* dy, const float* u_bwd, const float* res, float* dx, const float* x, void* slabs, size_t slab_bytes,
                                      int N, int H, int W, int* nslab, void* stream) {
  if (!dy || !u_bwd || !dx || !x || !slabs || !nslab) return fail(HIFIHR_EINVAL, "hifihr_conv3x3_c64_bwd_pair_slabs: null pointer");
  if (!hifihr::conv_c64_bwd_pair_supported(N, H, W) || slab_bytes < hifihr::conv_halo_wgrad_slab_bytes())
    return fail(HIFIHR_EINVAL, "hifihr_conv3x3_c64_bwd_pair_slabs: unsupported shape, or fewer than hifihr_conv2d_wgrad_workspace_bytes slab bytes");
  HIP_TRY(hifihr::launch_conv_c64_bwd_pair(dy, u_bwd, res, dx, x, nullptr, static_cast<float*>(slabs), N, H, W, (hipStream_t)stream, nslab));
  return HIFIHR_OK;
}

int hifihr_conv_halo_wgrad_reduce_multi(const hifihr_halo_reduce_job* jobs, int njobs, void* stream) {
  static_assert(sizeof(hifihr_halo_reduce_job) == sizeof(hifihr::HaloReduceJob), "job layouts");
  if (!jobs || njobs <= 0) return fail(HIFIHR_EINVAL, "hifihr_conv_halo_wgrad_reduce_multi: bad argument");
  for (int i = 0; i < njobs; ++i)
    if (!jobs[i].slabs_d || !jobs[i].dw_acc_d || jobs[i].nslab <= 0) return fail(HIFIHR_EINVAL, "hifihr_conv_halo_wgrad_reduce_multi: bad job");
  HIP_TRY(hifihr::launch_conv_halo_wgrad_reduce_multi(reinterpret_cast<const hifihr::HaloReduceJob*>(jobs), njobs, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_bn_input_supported(int C, int m) { return (m == 4 && hifihr::wino4_bn_supported(C)) ? 1 : 0; }

int hifihr_wino_bn_input_transform(const float* x, float* stats, const float* gamma, const float* beta, const float* residual, float* out,
                                   float* V, int N, int H, int W, int C, int m, float eps, float momentum, float* save_mean,
                                   float* save_invstd, float* running_mean, float* running_var, void* stream) {
  if (!x || !stats || !gamma || !beta || !V || !save_mean || !save_invstd || N <= 0 || H <= 0 || W <= 0 || ((residual == nullptr) != (out == nullptr)))
    return fail(HIFIHR_EINVAL, "hifihr_wino_bn_input_transform: bad argument");
  if (!hifihr_wino_bn_input_supported(C, m)) return fail(HIFIHR_EINVAL, "hifihr_wino_bn_input_transform: needs m = 4 and C % 4 == 0, C <= 512");
  HIP_TRY(hifihr::launch_wino4_bn_input_transform(x, stats, gamma, beta, residual, out, V, N, H, W, C, eps, momentum, save_mean, save_invstd,
                                                  running_mean, running_var, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_output_transform_bnred(const float* Mm, const float* x, const float* out, const float* gadd, const float* save_mean,
                                       const float* save_invstd, const float* gamma, const float* beta, float* red, float* g, int N, int H, int W,
                                       int C, int m, void* stream) {
  if (!Mm || !x || !save_mean || !save_invstd || !gamma || !beta || !red || !g || N <= 0 || H <= 0 || W <= 0)
    return fail(HIFIHR_EINVAL, "hifihr_wino_output_transform_bnred: bad argument");
  if (!hifihr_wino_bn_input_supported(C, m)) return fail(HIFIHR_EINVAL, "hifihr_wino_output_transform_bnred: needs m = 4 and C % 4 == 0, C <= 512");
  HIP_TRY(hifihr::launch_wino4_output_transform_bnred(Mm, x, out, gadd, save_mean, save_invstd, gamma, beta, red, g, N, H, W, C, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_bn_bwd_dual_transform(const float* g, const float* y, const float* save_mean, const float* save_invstd, const float* gamma,
                                      float* red, float* V, float* Yt, int N, int H, int W, int K, int m, float* dgamma_acc, float* dbeta_acc,
                                      void* stream) {
  if (!g || !y || !save_mean || !save_invstd || !gamma || !red || !V || !Yt || N <= 0 || H <= 0 || W <= 0)
    return fail(HIFIHR_EINVAL, "hifihr_wino_bn_bwd_dual_transform: bad argument");
  if (!hifihr_wino_bn_input_supported(K, m)) return fail(HIFIHR_EINVAL, "hifihr_wino_bn_bwd_dual_transform: needs m = 4 and K % 4 == 0, K <= 512");
  HIP_TRY(hifihr::launch_wino4_bn_bwd_dual_transform(g, y, save_mean, save_invstd, gamma, red, V, Yt, N, H, W, K, dgamma_acc, dbeta_acc,
                                                     (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_bn_bwd_apply(const float* g, const float* x, const float* save_mean, const float* save_invstd, const float* gamma, long M, int C,
                        float* red, float* dx, float* dgamma_acc, float* dbeta_acc, void* stream) {
  if (!g || !x || !save_mean || !save_invstd || !gamma || !red || !dx || M <= 0) return fail(HIFIHR_EINVAL, "hifihr_bn_bwd_apply: bad argument");
  HIP_TRY(hifihr::launch_bn_bwd_apply(g, x, save_mean, save_invstd, gamma, M, C, red, dx, dgamma_acc, dbeta_acc, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_input_dy_transform_m(const float* dy, float* V, float* Yt, int N, int H, int W, int K, int m, void* stream) {
  if (m != 4) return hifihr_wino_input_dy_transform(dy, V, Yt, N, H, W, K, stream);
  if (!dy || !V || !Yt || N <= 0 || H <= 0 || W <= 0 || K < 4 || K % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_wino_input_dy_transform: bad argument");
  HIP_TRY(hifihr::launch_wino4_input_transform(dy, V, Yt, N, H, W, K, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_output_transform_act_m(const float* M, float* y, const float* bias, int act, int N, int H, int W, int K, int m, void* stream) {
  if (m != 4) return hifihr_wino_output_transform_act(M, y, bias, act, N, H, W, K, stream);
  if (!M || !y || N <= 0 || H <= 0 || W <= 0 || K < 4 || K % 4 != 0 || act < 0 || act > 1)
    return fail(HIFIHR_EINVAL, "hifihr_wino_output_transform_act: bad argument");
  HIP_TRY(hifihr::launch_wino4_output_transform(M, y, nullptr, bias, act, nullptr, N, H, W, K, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_output_transform_mask_m(const float* M, float* y, const float* mask, int N, int H, int W, int K, int m, void* stream) {
  if (m != 4) return fail(HIFIHR_EINVAL, "hifihr_wino_output_transform_mask_m: the F(4x4, 3x3) pipeline only (m == 4)");
  if (!M || !y || !mask || N <= 0 || H <= 0 || W <= 0 || K < 4 || K % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_wino_output_transform_mask_m: bad argument");
  HIP_TRY(hifihr::launch_wino4_output_transform(M, y, nullptr, nullptr, 0, mask, N, H, W, K, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_input_transform(const float* x, float* V, int N, int H, int W, int C, void* stream) {
  if (!x || !V || N <= 0 || H <= 0 || W <= 0 || C < 4 || C % 4 != 0) return fail(HIFIHR_EINVAL, "hifihr_wino_input_transform: bad argument");
  HIP_TRY(hifihr::launch_wino_input_transform(x, V, nullptr, N, H, W, C, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_gemm(const float* V, const float* U, float* M, int N, int H, int W, int C, int K, void* ws, size_t ws_bytes, void* stream) {
  if (!V || !U || !M || N <= 0 || H <= 0 || W <= 0 || C < 32 || C % 32 != 0 || K < 4 || K % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_wino_gemm: bad argument (C % 32 == 0, K % 4 == 0)");
  const long T = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
  if (use_bgemm() && T < (1L << 30) && hifihr::bgemm_nt_supported((int)T, K, C)) {
    HIP_TRY(hifihr::launch_bgemm_nt(V, U, M, (int)T, K, C, 16, ws, ws_bytes, (hipStream_t)stream));
    return HIFIHR_OK;
  }
  HIP_TRY(hifihr::launch_conv_igemm(wino_gemm_geom(T, C, K), V, U, nullptr, M, nullptr, ws, ws_bytes, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_dy_transform(const float* dy, float* Y, int N, int H, int W, int K, void* stream) {
  if (!dy || !Y || N <= 0 || H <= 0 || W <= 0 || K < 4 || K % 4 != 0) return fail(HIFIHR_EINVAL, "hifihr_wino_dy_transform: bad argument");
  HIP_TRY(hifihr::launch_wino_dy_transform(dy, Y, N, H, W, K, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_wgrad_gemm(const float* V, const float* Y, float* dU_zeroed, int N, int H, int W, int C, int K, void* stream) {
  if (!V || !Y || !dU_zeroed || N <= 0 || H <= 0 || W <= 0 || C < 4 || C % 4 != 0 || K < 4 || K % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_wino_wgrad_gemm: bad argument (C % 4 == 0, K % 4 == 0)");
  const long T = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
  HIP_TRY(hifihr::launch_conv_wgrad(wino_gemm_geom(T, C, K), V, Y, dU_zeroed, nullptr, 0, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_dw_transform(float* dU, float* dw_acc, int K, int C, int clear_du, void* stream) {
  if (!dU || !dw_acc || K <= 0 || C < 4 || C % 4 != 0) return fail(HIFIHR_EINVAL, "hifihr_wino_dw_transform: bad argument");
  HIP_TRY(hifihr::launch_wino_dw_transform(dU, dw_acc, K, C, clear_du ? 1 : 0, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_output_transform(const float* M, float* y, float* stats, int N, int H, int W, int K, void* stream) {
  if (!M || !y || N <= 0 || H <= 0 || W <= 0 || K < 4 || K % 4 != 0) return fail(HIFIHR_EINVAL, "hifihr_wino_output_transform: bad argument");
  HIP_TRY(hifihr::launch_wino_output_transform(M, y, stats, nullptr, 0, N, H, W, K, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_freihand_augment(const uint32_t* img_rgbx, const uint8_t* mask, const int* idx, const int* coef_fix, int B, int H, int W,
                            float* out_img, float* out_mask, void* stream) {
  if (!idx || !coef_fix || (!out_img && !out_mask) || (out_img && !img_rgbx) || (out_mask && !mask) || B <= 0 || H <= 0 || W <= 0)
    return fail(HIFIHR_EINVAL, "hifihr_freihand_augment: bad argument");
  HIP_TRY(hifihr::launch_freihand_augment(img_rgbx, mask, idx, coef_fix, B, H, W, out_img, out_mask, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_freihand_batch(const uint32_t* img_rgbx, const uint8_t* mask, const float* Ks, const float* joints, const float* verts,
                          const float* scales, int J, int V, const int* packed, int B, int H, int W, float* out_img, float* out_mask,
                          long long* out_segm, float* out_Ks, float* out_Ps, float* out_joints, float* out_verts, float* out_j2d,
                          float* out_scales, long long* out_idxs, void* stream) {
  if (!img_rgbx || !mask || !Ks || !scales || !packed || B <= 0 || H <= 0 || W <= 0 || (long)H * W >= (1L << 24) || J < 0 || V < 0 ||
      (J > 0 && !joints) || (V > 0 && !verts) || (out_j2d && !out_joints && J > 0 && !joints))
    return fail(HIFIHR_EINVAL, "hifihr_freihand_batch: bad argument");
  HIP_TRY(hifihr::launch_freihand_batch(img_rgbx, mask, Ks, joints, verts, scales, J, V, packed, B, H, W, out_img, out_mask, out_segm, out_Ks,
                                        out_Ps, out_joints, out_verts, out_j2d, out_scales, out_idxs, hifihr::BatchStepOut{-1, 1.f, nullptr, nullptr, nullptr, nullptr},
                                        (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_freihand_batch_step(const uint32_t* img_rgbx, const uint8_t* mask, const float* Ks, const float* joints, const float* verts,
                               const float* scales, int J, int V, const int* packed, int B, int H, int W, float* out_img, float* out_mask,
                               long long* out_segm, float* out_Ks, float* out_Ps, float* out_joints, float* out_verts, float* out_j2d,
                               float* out_scales, long long* out_idxs, int root_id, float image_size, float* out_root,
                               float* out_joints_rel, float* out_verts_rel, float* out_cam_ndc, void* stream) {
  if (!img_rgbx || !mask || !Ks || !scales || !packed || B <= 0 || H <= 0 || W <= 0 || (long)H * W >= (1L << 24) || J < 0 || V < 0 ||
      (J > 0 && !joints) || (V > 0 && !verts) || root_id >= J || !(image_size > 0.f))
    return fail(HIFIHR_EINVAL, "hifihr_freihand_batch_step: bad argument");
  HIP_TRY(hifihr::launch_freihand_batch(img_rgbx, mask, Ks, joints, verts, scales, J, V, packed, B, H, W, out_img, out_mask, out_segm, out_Ks,
                                        out_Ps, out_joints, out_verts, out_j2d, out_scales, out_idxs,
                                        hifihr::BatchStepOut{root_id, image_size, out_root, out_joints_rel, out_verts_rel, out_cam_ndc},
                                        (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_input_dy_transform(const float* dy, float* V, float* Yt, int N, int H, int W, int K, void* stream) {
  if (!dy || !V || !Yt || N <= 0 || H <= 0 || W <= 0 || K < 4 || K % 4 != 0)
    return fail(HIFIHR_EINVAL, "hifihr_wino_input_dy_transform: bad argument");
  HIP_TRY(hifihr::launch_wino_input_transform(dy, V, Yt, N, H, W, K, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_procrustes_error(const float* pred, const float* gt, int B, int N, float* aligned, float* err_sum, void* stream) {
  if (!pred || !gt || !err_sum || B <= 0 || N <= 0) return fail(HIFIHR_EINVAL, "hifihr_procrustes_error: bad argument");
  HIP_TRY(hifihr::launch_procrustes(pred, gt, B, N, aligned, err_sum, (hipStream_t)stream));
  return HIFIHR_OK;
}

size_t hifihr_ho3d_workspace_bytes(int B, int out_size) {
  return (B > 0 && out_size > 0 && out_size <= 256) ? hifihr::ho3d_workspace_bytes(B, out_size) : 0;
}

int hifihr_ho3d_batch(const uint32_t* img_rgbx, const uint8_t* hand_mask, const float* Ks, const float* uv21, const float* xyz21, int FH,
                      int FW, const int* packed, int B, int out_size, void* ws, size_t ws_bytes, float* out_img, float* out_mask,
                      float* out_K, float* out_uv21, float* out_xyz21, void* stream) {
  if (!packed || !ws || B <= 0 || FH <= 0 || FW <= 0 || out_size <= 0 || out_size > 256 || ws_bytes < hifihr::ho3d_workspace_bytes(B, out_size) ||
      (out_img && !img_rgbx) || (out_mask && !hand_mask) || (out_K && !Ks) || (out_uv21 && !uv21) || (out_xyz21 && !xyz21))
    return fail(HIFIHR_EINVAL, "hifihr_ho3d_batch: bad argument (out_size <= 256, workspace of hifihr_ho3d_workspace_bytes)");
  HIP_TRY(hifihr::launch_ho3d_batch(img_rgbx, hand_mask, Ks, uv21, xyz21, FH, FW, packed, B, out_size, ws, out_img, out_mask, out_K, out_uv21,
                                    out_xyz21, (hipStream_t)stream));
  return HIFIHR_OK;
}

int hifihr_wino_output_transform_act(const float* M, float* y, const float* bias, int act, int N, int H, int W, int K, void* stream) {
  if (!M || !y || N <= 0 || H <= 0 || W <= 0 || K < 4 || K % 4 != 0 || act < 0 || act > 1)
    return fail(HIFIHR_EINVAL, "hifihr_wino_output_transform_act: bad argument");
  HIP_TRY(hifihr::launch_wino_output_transform(M, y, nullptr, bias, act, N, H, W, K, (hipStream_t)stream));
  return HIFIHR_OK;
}

}  // extern "C"
