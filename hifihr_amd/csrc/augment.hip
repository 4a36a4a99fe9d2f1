// FreiHAND training augmentation on the device: the in-plane rotation warp of image and mask that
// the reference does per sample on CPU workers with PIL (reference data/dataset.py:223-270 -> utils/handutils.py:48-60
// transform_img = Image.transform(size, AFFINE, coefficients), nearest-neighbour, zero fill), followed by to_tensor
// (u8 / 255, CHW) and, for masks, torch.round.
//
// The decoded dataset lives in HBM as uint8 (32 560 x 224 x 224 RGBX = 6.5 GB of the 288 GB); a batch is gathered and warped by
// one launch, no host pixels involved.  PIL's nearest-neighbour affine runs in 16.16 fixed point: with the six coefficients
// (a b c; d e f) it visits output pixel (x, y) at
//     xin = (FIX(c + a/2 + b/2) + x FIX(a) + y FIX(b)) >> 16,   yin = (FIX(f + d/2 + e/2) + x FIX(d) + y FIX(e)) >> 16,
// FIX(v) = floor(v 65536 + 0.5), and copies the input pixel when it lies inside the image.  The host computes the six FIX values
// per sample (hifihr_amd/data.py, same float arithmetic as numpy / PIL); the kernel is integer arithmetic and a gather: bit-exact.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "hifihr_internal.h"

namespace hifihr {

// thread = one output pixel; grid = (ceil(H*W / 256), B)
__global__ __launch_bounds__(256) void freihand_augment_kernel(const uint32_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                              const int* __restrict__ idx, const int* __restrict__ coef, int H, int W,
                                                              float* __restrict__ out_img, float* __restrict__ out_mask,
                                                              long long* __restrict__ out_segm) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  const int* c = coef + b * 6;                   // {FIX(a), FIX(b), FIX(c + a/2 + b/2), FIX(d), FIX(e), FIX(f + d/2 + e/2)}
  const int xin = (c[2] + x * c[0] + y * c[1]) >> 16;
  const int yin = (c[5] + x * c[3] + y * c[4]) >> 16;
  const bool in = xin >= 0 && xin < W && yin >= 0 && yin < H;
  const size_t src = ((size_t)idx[b] * H + (in ? yin : 0)) * W + (in ? xin : 0);
  const size_t plane = (size_t)H * W;
  if (out_img != nullptr) {
    const uint32_t v = in ? img[src] : 0u;       // little-endian R, G, B, X
    float* o = out_img + (size_t)b * 3 * plane + p;
    o[0] = (float)(v & 0xffu) / 255.0f;
    o[plane] = (float)((v >> 8) & 0xffu) / 255.0f;
    o[2 * plane] = (float)((v >> 16) & 0xffu) / 255.0f;
  }
  if (out_mask != nullptr) {
    const float m = (in ? mask[src] : (uint8_t)0) >= 128 ? 1.0f : 0.0f;      // round(u8 / 255)
    float* o = out_mask + (size_t)b * 3 * plane + p;
    o[0] = m; o[plane] = m; o[2 * plane] = m;
  }
  if (out_segm != nullptr)                       // masks[:, 0].long() of the reference's data_dic (traineval_util.py:104)
    out_segm[(size_t)b * plane + p] = ((in ? mask[src] : (uint8_t)0) >= 128) ? 1ll : 0ll;
}

// The same, four consecutive pixels of a row per thread (W % 4 == 0, 16-byte aligned planes): the outputs are seven planes of 4 (8) bytes
// per pixel -- one float4 (two longlong2) store per plane instead of four scalar ones (round 5; DESIGN.md section 0 has the times;
// the arithmetic per pixel is the scalar kernel's).
__global__ __launch_bounds__(256) void freihand_augment4_kernel(const uint32_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                               const int* __restrict__ idx, const int* __restrict__ coef, int H, int W,
                                                               float* __restrict__ out_img, float* __restrict__ out_mask,
                                                               long long* __restrict__ out_segm) {
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;            // group of four pixels
  const int W4 = W >> 2;
  if (q >= H * W4) return;
  const int y = q / W4, x0 = (q - y * W4) << 2;
  const int* c = coef + b * 6;
  const int c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
  const size_t base = (size_t)idx[b] * H;
  const size_t plane = (size_t)H * W;
  const int p = y * W + x0;
  uint32_t v[4];
  uint8_t mk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int x = x0 + i;
    const int xin = (c2 + x * c0 + y * c1) >> 16;
    const int yin = (c5 + x * c3 + y * c4) >> 16;
    const bool in = xin >= 0 && xin < W && yin >= 0 && yin < H;
    const size_t src = (base + (in ? yin : 0)) * W + (in ? xin : 0);
    v[i] = (out_img != nullptr && in) ? img[src] : 0u;
    mk[i] = ((out_mask != nullptr || out_segm != nullptr) && in) ? mask[src] : (uint8_t)0;
  }
  if (out_img != nullptr) {
    float* o = out_img + (size_t)b * 3 * plane + p;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      *reinterpret_cast<float4*>(o + ch * plane) = make_float4((float)((v[0] >> (8 * ch)) & 0xffu) / 255.0f, (float)((v[1] >> (8 * ch)) & 0xffu) / 255.0f,
                                                               (float)((v[2] >> (8 * ch)) & 0xffu) / 255.0f, (float)((v[3] >> (8 * ch)) & 0xffu) / 255.0f);
  }
  if (out_mask != nullptr) {
    const float4 m = make_float4(mk[0] >= 128 ? 1.0f : 0.0f, mk[1] >= 128 ? 1.0f : 0.0f, mk[2] >= 128 ? 1.0f : 0.0f, mk[3] >= 128 ? 1.0f : 0.0f);
    float* o = out_mask + (size_t)b * 3 * plane + p;
    *reinterpret_cast<float4*>(o) = m;
    *reinterpret_cast<float4*>(o + plane) = m;
    *reinterpret_cast<float4*>(o + 2 * plane) = m;
  }
  if (out_segm != nullptr) {
    struct alignas(16) Pair { long long a, b; };               // one 16-byte store
    Pair* o = reinterpret_cast<Pair*>(out_segm + (size_t)b * plane + p);
    o[0] = Pair{mk[0] >= 128 ? 1ll : 0ll, mk[1] >= 128 ? 1ll : 0ll};
    o[1] = Pair{mk[2] >= 128 ? 1ll : 0ll, mk[3] >= 128 ? 1ll : 0ll};
  }
}

static void launch_augment_planes(const uint32_t* img, const uint8_t* mask, const int* idx, const int* coef, int B, int H, int W, float* out_img,
                                  float* out_mask, long long* out_segm, hipStream_t st) {
  static const int vec = [] { const char* e = getenv("HIFIHR_AUGMENT_VEC4"); return e ? atoi(e) : 1; }();
  const bool aligned = ((reinterpret_cast<uintptr_t>(out_img) | reinterpret_cast<uintptr_t>(out_mask) | reinterpret_cast<uintptr_t>(out_segm)) & 15) == 0;
  if (vec && W % 4 == 0 && aligned)
    hipLaunchKernelGGL(freihand_augment4_kernel, dim3((H * (W / 4) + 255) / 256, B), dim3(256), 0, st, img, mask, idx, coef, H, W, out_img, out_mask,
                       out_segm);
  else
    hipLaunchKernelGGL(freihand_augment_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, st, img, mask, idx, coef, H, W, out_img, out_mask,
                       out_segm);
}

// Everything else a FreiHAND training batch holds (reference data/dataset.py:256-275 per sample + utils/traineval_util.py:21-111
// data_dic per batch), one workgroup per sample:
//   Ks     = post_rot_trans . K[idx]                          (:258-260)
//   joints = (R joints[idx]^T)^T, verts likewise              (:271-275)
//   Ps     = [Ks | 0],  j2d_gt = proj_func(joints, Ks) = (Ks j)_xy / (Ks j)_z   (fh_utils.py:30-39),  scales, idxs (int64)
// Sums of three products in the order k = 0, 1, 2, fp32, IEEE division: the same expressions as the torch broadcast form in
// hifihr_amd/data.py:batch (which the tests compare it with).
struct BatchMeta {
  const float *Ks, *joints, *verts, *scales;     // cache: [n][3][3], [n][J][3], [n][V][3], [n]
  const int* packed;                             // [25 B]: idx[B], coef[B][6], post[B][3][3] (f32 bits), rot[B][3][3] (f32 bits)
  int B, J, V;
  float *oKs, *oPs, *ojoints, *overts, *oj2d, *oscales;
  long long* oidx;
  BatchStepOut x;                                  // what the training step derives from the batch on every iteration (all optional)
};
__global__ __launch_bounds__(256) void freihand_batch_meta_kernel(BatchMeta m) {
  __shared__ float sK[9], sR[9], sRoot[3];
  const int b = blockIdx.x, t = threadIdx.x;
  const int id = m.packed[b];
  const float* post = reinterpret_cast<const float*>(m.packed + 7 * m.B) + b * 9;
  const float* rot = reinterpret_cast<const float*>(m.packed + 16 * m.B) + b * 9;
  if (t < 9) {
    const int i = t / 3, j = t - 3 * i;
    const float* K = m.Ks + (size_t)id * 9;
    float a = post[i * 3 + 0] * K[0 * 3 + j];
    a += post[i * 3 + 1] * K[1 * 3 + j];
    a += post[i * 3 + 2] * K[2 * 3 + j];
    sK[t] = a;
    sR[t] = rot[t];
    if (m.oKs) m.oKs[b * 9 + t] = a;
    if (m.oPs) m.oPs[b * 12 + i * 4 + j] = a;
  }
  if (t < 3 && m.oPs) m.oPs[b * 12 + t * 4 + 3] = 0.f;
  if (t == 0) {
    if (m.oscales) m.oscales[b] = m.scales[id];
    if (m.oidx) m.oidx[b] = id;
  }
  __syncthreads();
  // the step's own terms (train_hrnet.py:62-68; models_res_nimble.py:228-235): root = joints[:, root_id] of the ROTATED joints, the
  // root-relative ground truth, the NDC camera (-2 fx / s, -2 fy / s, 1 - 2 cx / s, 1 - 2 cy / s) of the rotated intrinsics
  if (t < 3) {
    float a = 0.f;
    if (m.x.root_id >= 0 && m.x.root_id < m.J) {
      const float* src = m.joints + ((size_t)id * m.J + m.x.root_id) * 3;
      a = src[0] * sR[t * 3 + 0];
      a += src[1] * sR[t * 3 + 1];
      a += src[2] * sR[t * 3 + 2];
    }
    sRoot[t] = a;
    if (m.x.oroot) m.x.oroot[b * 3 + t] = a;
  }
  if (t >= 64 && t < 68 && m.x.ocam) {
    const int e = t - 64;
    const float sc = -2.0f / m.x.image_size;
    const float k = e == 0 ? sK[0] : e == 1 ? sK[4] : e == 2 ? sK[2] : sK[5];
    m.x.ocam[b * 4 + e] = (e < 2 ? 0.f : 1.f) + k * sc;
  }
  __syncthreads();
  for (int p = t; p < m.J + m.V; p += 256) {
    const bool isj = p < m.J;
    const float* src = isj ? m.joints + ((size_t)id * m.J + p) * 3 : m.verts + ((size_t)id * m.V + (p - m.J)) * 3;
    const float x = src[0], y = src[1], z = src[2];
    float r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float a = x * sR[i * 3 + 0];
      a += y * sR[i * 3 + 1];
      a += z * sR[i * 3 + 2];
      r[i] = a;
    }
    float* dst = isj ? (m.ojoints ? m.ojoints + ((size_t)b * m.J + p) * 3 : nullptr)
                     : (m.overts ? m.overts + ((size_t)b * m.V + (p - m.J)) * 3 : nullptr);
    if (dst) { dst[0] = r[0]; dst[1] = r[1]; dst[2] = r[2]; }
    float* rel = isj ? (m.x.ojoints_rel ? m.x.ojoints_rel + ((size_t)b * m.J + p) * 3 : nullptr)
                     : (m.x.overts_rel ? m.x.overts_rel + ((size_t)b * m.V + (p - m.J)) * 3 : nullptr);
    if (rel) { rel[0] = r[0] - sRoot[0]; rel[1] = r[1] - sRoot[1]; rel[2] = r[2] - sRoot[2]; }
    if (isj && m.oj2d) {
      float uv[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float a = r[0] * sK[i * 3 + 0];
        a += r[1] * sK[i * 3 + 1];
        a += r[2] * sK[i * 3 + 2];
        uv[i] = a;
      }
      m.oj2d[((size_t)b * m.J + p) * 2 + 0] = uv[0] / uv[2];
      m.oj2d[((size_t)b * m.J + p) * 2 + 1] = uv[1] / uv[2];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// HO-3D sample assembly (the HO-3D half): the hand crop of reference data/dataset.py:1105-1215 --
// `func_transforms.resized_crop(image, y1, x1, size, size, [224, 224])` for the frame (bilinear) and for the hand mask (bicubic), i.e.
// Pillow's Image.crop (box rounded half-to-even, zero fill outside the frame) followed by Image.resize for 8-bit images
// (libImaging/Resample.c): a horizontal then a vertical pass, each a normalised filter of support (filter support x max(scale, 1))
// evaluated in DOUBLE, quantised to 22-bit fixed point, accumulated in 32-bit integers with rounding and clipped to uint8 BETWEEN the passes.
// The kernels follow that arithmetic step by step (IEEE double on the device, integer accumulation): bit-exact with Pillow
// (tests/golden/ho3d_path.npz holds Pillow's own outputs).
//   ho3d_coeff_kernel     (4 tables, B) x 256: table t = filter (t >> 1: bilinear, bicubic) x axis (t & 1: x, y): per output row / column its
//                         first source index, tap count and <= kHoTaps fixed-point coefficients (Resample.c precompute_coeffs +
//                         normalize_coeffs_8bpc)
//   ho3d_resample_kernel  thread = output pixel: for each of its <= kHoTaps source rows the horizontal sum (rounded, clipped to 8 bits),
//                         then the vertical sum of those; frame as u8 / 255 in three planes, mask as round(u8 / 255)
//   ho3d_meta_kernel      uv21_crop = (uv21 - centre) * scale + 112, K_crop = T . S . K (:1186-1210), xyz21 gathered
// ------------------------------------------------------------------------------------------------
constexpr int kHoTaps = 16;          // taps per output element: 2 * ceil(support) + 1 <= 13 for a 640-pixel window with the bicubic filter
constexpr int kHoBits = 32 - 8 - 2;  // PRECISION_BITS of Resample.c

struct Ho3dTables {                  // per sample and table: bounds[out][2] = (first source index, taps), kk[out][kHoTaps]
  int* bounds;
  int* kk;
};

__device__ __forceinline__ double ho_filter(int bicubic, double x) {
  x = x < 0.0 ? -x : x;
  if (!bicubic) return x < 1.0 ? 1.0 - x : 0.0;
  const double a = -0.5;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

__global__ __launch_bounds__(256) void ho3d_coeff_kernel(const int* __restrict__ boxes, int out_size, Ho3dTables tb) {
  const int b = blockIdx.y, t = blockIdx.x;                  // table t: filter t >> 1, axis t & 1
  const int xx = threadIdx.x;
  if (xx >= out_size) return;
  const int* box = boxes + b * 4;                            // x0, y0, x1, y1 of the rounded crop box
  const int in_size = (t & 1) ? box[3] - box[1] : box[2] - box[0];
  const int bicubic = t >> 1;
  int* bo = tb.bounds + ((size_t)(b * 4 + t) * out_size + xx) * 2;
  int* ko = tb.kk + ((size_t)(b * 4 + t) * out_size + xx) * kHoTaps;
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = (bicubic ? 2.0 : 1.0) * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = 0.0 + (xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  if (xmax > kHoTaps) xmax = kHoTaps;                        // (cannot happen for windows <= 640 pixels; keeps the table in bounds)
  if (xmax < 0) xmax = 0;
  double k[kHoTaps];
  double ww = 0.0;
  for (int x = 0; x < kHoTaps; ++x) {
    double w = 0.0;
    if (x < xmax) { w = ho_filter(bicubic, (x + xmin - center + 0.5) * ss); ww += w; }
    k[x] = w;
  }
  for (int x = 0; x < kHoTaps; ++x) {
    double v = k[x];
    if (x < xmax && ww != 0.0) v /= ww;
    ko[x] = v < 0 ? (int)(-0.5 + v * (double)(1 << kHoBits)) : (int)(0.5 + v * (double)(1 << kHoBits));
  }
  bo[0] = xmin; bo[1] = xmax;
}

__device__ __forceinline__ int ho_clip8(int v) {
  v >>= kHoBits;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// grid = (ceil(out^2 / 256), B, 2): z = 0 the frame (RGBX words, bilinear tables 0 / 1), z = 1 the hand mask (bytes, bicubic tables 2 / 3)
__global__ __launch_bounds__(256) void ho3d_resample_kernel(const uint32_t* __restrict__ img, const uint8_t* __restrict__ mask,
                                                           const int* __restrict__ idx, const int* __restrict__ boxes, int FH, int FW,
                                                           int out_size, Ho3dTables tb, float* __restrict__ out_img,
                                                           float* __restrict__ out_mask) {
  const int b = blockIdx.y, which = blockIdx.z;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= out_size * out_size) return;
  if ((which == 0 && out_img == nullptr) || (which == 1 && out_mask == nullptr)) return;
  const int yy = p / out_size, xx = p - yy * out_size;
  const int* box = boxes + b * 4;
  const int bx0 = box[0], by0 = box[1];
  const size_t tx = (size_t)(b * 4 + 2 * which) * out_size + xx, ty = (size_t)(b * 4 + 2 * which + 1) * out_size + yy;
  const int x0 = tb.bounds[tx * 2], nx = tb.bounds[tx * 2 + 1], y0 = tb.bounds[ty * 2], ny = tb.bounds[ty * 2 + 1];
  const int* kx = tb.kk + tx * kHoTaps;
  const int* ky = tb.kk + ty * kHoTaps;
  const size_t frame = (size_t)idx[b] * FH * FW;
  const int half = 1 << (kHoBits - 1);
  int v0 = half, v1 = half, v2 = half;
  for (int j = 0; j < ny; ++j) {
    const int fy = by0 + y0 + j;                             // frame row of cropped row y0 + j
    const bool rok = fy >= 0 && fy < FH;
    int h0 = half, h1 = half, h2 = half;
    for (int i = 0; i < nx; ++i) {
      const int fx = bx0 + x0 + i;
      const bool ok = rok && fx >= 0 && fx < FW;             // Image.crop fills what lies outside the frame with zeros
      const int w = kx[i];
      if (which == 0) {
        const uint32_t px = ok ? img[frame + (size_t)fy * FW + fx] : 0u;
        h0 += (int)(px & 0xffu) * w; h1 += (int)((px >> 8) & 0xffu) * w; h2 += (int)((px >> 16) & 0xffu) * w;
      } else {
        h0 += (int)(ok ? mask[frame + (size_t)fy * FW + fx] : (uint8_t)0) * w;
      }
    }
    const int w = ky[j];
    v0 += ho_clip8(h0) * w;
    if (which == 0) { v1 += ho_clip8(h1) * w; v2 += ho_clip8(h2) * w; }
  }
  const size_t plane = (size_t)out_size * out_size;
  if (which == 0) {
    float* o = out_img + (size_t)b * 3 * plane + p;
    o[0] = (float)ho_clip8(v0) / 255.0f; o[plane] = (float)ho_clip8(v1) / 255.0f; o[2 * plane] = (float)ho_clip8(v2) / 255.0f;
  } else {
    out_mask[(size_t)b * plane + p] = ho_clip8(v0) >= 128 ? 1.0f : 0.0f;                 // to_tensor().round()
  }
}

struct Ho3dMeta {
  const float *Ks, *uv21, *xyz21;                // cache: [n][3][3], [n][21][2], [n][21][3]
  const int* idx;
  const float* win;                              // [B][3]: crop centre (u, v), scale
  float *oK, *ouv, *oxyz;
  float half_res;                                // inp_res // 2
};
__global__ __launch_bounds__(64) void ho3d_meta_kernel(Ho3dMeta m) {
  const int b = blockIdx.x, t = threadIdx.x, id = m.idx[b];
  const float cu = m.win[b * 3], cv = m.win[b * 3 + 1], s = m.win[b * 3 + 2];
  if (t < 21) {
    if (m.ouv) {
      m.ouv[(b * 21 + t) * 2] = (m.uv21[((size_t)id * 21 + t) * 2] - cu) * s + m.half_res;
      m.ouv[(b * 21 + t) * 2 + 1] = (m.uv21[((size_t)id * 21 + t) * 2 + 1] - cv) * s + m.half_res;
    }
    if (m.oxyz)
      for (int c = 0; c < 3; ++c) m.oxyz[(b * 21 + t) * 3 + c] = m.xyz21[((size_t)id * 21 + t) * 3 + c];
  }
  if (t < 9 && m.oK) {
    // K_crop = T . (S . K): S = diag(s, s, 1), T = [1 0 -t1; 0 1 -t2; 0 0 1], t = centre * s - inp_res // 2
    const int i = t / 3, j = t - 3 * i;
    const float* K = m.Ks + (size_t)id * 9;
    const float sk = (i < 2 ? s : 1.0f) * K[i * 3 + j];
    const float tr = i == 0 ? -(cu * s - m.half_res) : i == 1 ? -(cv * s - m.half_res) : 0.0f;
    m.oK[b * 9 + t] = i < 2 ? sk + tr * (1.0f * K[6 + j]) : sk;
  }
}

hipError_t launch_ho3d_batch(const uint32_t* img, const uint8_t* mask, const float* Ks, const float* uv21, const float* xyz21, int FH, int FW,
                             const int* packed, int B, int out_size, void* ws, float* out_img, float* out_mask, float* out_K,
                             float* out_uv, float* out_xyz, hipStream_t st) {
  if (B <= 0 || FH <= 0 || FW <= 0 || out_size <= 0 || out_size > 256 || ws == nullptr) return hipErrorInvalidValue;
  // packed: idx[B], boxes[B][4], window[B][3] (float bits)
  const int* idx = packed;
  const int* boxes = packed + B;
  const float* win = reinterpret_cast<const float*>(packed + 5 * B);
  Ho3dTables tb;
  tb.bounds = static_cast<int*>(ws);
  tb.kk = tb.bounds + (size_t)B * 4 * out_size * 2;
  hipLaunchKernelGGL(ho3d_coeff_kernel, dim3(4, B), dim3(256), 0, st, boxes, out_size, tb);
  hipLaunchKernelGGL(ho3d_resample_kernel, dim3((out_size * out_size + 255) / 256, B, 2), dim3(256), 0, st, img, mask, idx, boxes, FH, FW,
                     out_size, tb, out_img, out_mask);
  const Ho3dMeta m{Ks, uv21, xyz21, idx, win, out_K, out_uv, out_xyz, (float)(out_size / 2)};
  hipLaunchKernelGGL(ho3d_meta_kernel, dim3(B), dim3(64), 0, st, m);
  return hipGetLastError();
}

size_t ho3d_workspace_bytes(int B, int out_size) { return (size_t)B * 4 * out_size * (2 + kHoTaps) * sizeof(int); }

hipError_t launch_freihand_augment(const uint32_t* img, const uint8_t* mask, const int* idx, const int* coef, int B, int H, int W,
                                   float* out_img, float* out_mask, hipStream_t st) {
  if (B <= 0 || H <= 0 || W <= 0 || (long)H * W >= (1L << 24)) return hipErrorInvalidValue;
  launch_augment_planes(img, mask, idx, coef, B, H, W, out_img, out_mask, nullptr, st);
  return hipGetLastError();
}

hipError_t launch_freihand_batch(const uint32_t* img, const uint8_t* mask, const float* Ks, const float* joints, const float* verts,
                                 const float* scales, int J, int V, const int* packed, int B, int H, int W, float* out_img, float* out_mask,
                                 long long* out_segm, float* oKs, float* oPs, float* ojoints, float* overts, float* oj2d, float* oscales,
                                 long long* oidx, const BatchStepOut& step, hipStream_t st) {
  if (B <= 0 || H <= 0 || W <= 0 || (long)H * W >= (1L << 24) || J < 0 || V < 0) return hipErrorInvalidValue;
  launch_augment_planes(img, mask, packed, packed + B, B, H, W, out_img, out_mask, out_segm, st);
  const BatchMeta m{Ks, joints, verts, scales, packed, B, J, V, oKs, oPs, ojoints, overts, oj2d, oscales, oidx, step};
  hipLaunchKernelGGL(freihand_batch_meta_kernel, dim3(B), dim3(256), 0, st, m);
  return hipGetLastError();
}

}  // namespace hifihr
