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
};
__global__ __launch_bounds__(256) void freihand_batch_meta_kernel(BatchMeta m) {
  __shared__ float sK[9], sR[9];
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

hipError_t launch_freihand_augment(const uint32_t* img, const uint8_t* mask, const int* idx, const int* coef, int B, int H, int W,
                                   float* out_img, float* out_mask, hipStream_t st) {
  if (B <= 0 || H <= 0 || W <= 0 || (long)H * W >= (1L << 24)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(freihand_augment_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, st, img, mask, idx, coef, H, W, out_img, out_mask,
                     (long long*)nullptr);
  return hipGetLastError();
}

hipError_t launch_freihand_batch(const uint32_t* img, const uint8_t* mask, const float* Ks, const float* joints, const float* verts,
                                 const float* scales, int J, int V, const int* packed, int B, int H, int W, float* out_img, float* out_mask,
                                 long long* out_segm, float* oKs, float* oPs, float* ojoints, float* overts, float* oj2d, float* oscales,
                                 long long* oidx, hipStream_t st) {
  if (B <= 0 || H <= 0 || W <= 0 || (long)H * W >= (1L << 24) || J < 0 || V < 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(freihand_augment_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, st, img, mask, packed, packed + B, H, W, out_img,
                     out_mask, out_segm);
  const BatchMeta m{Ks, joints, verts, scales, packed, B, J, V, oKs, oPs, ojoints, overts, oj2d, oscales, oidx};
  hipLaunchKernelGGL(freihand_batch_meta_kernel, dim3(B), dim3(256), 0, st, m);
  return hipGetLastError();
}

}  // namespace hifihr
