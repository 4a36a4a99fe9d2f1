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
                                                              float* __restrict__ out_img, float* __restrict__ out_mask) {
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
}

hipError_t launch_freihand_augment(const uint32_t* img, const uint8_t* mask, const int* idx, const int* coef, int B, int H, int W,
                                   float* out_img, float* out_mask, hipStream_t st) {
  if (B <= 0 || H <= 0 || W <= 0 || (long)H * W >= (1L << 24)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(freihand_augment_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, st, img, mask, idx, coef, H, W, out_img, out_mask);
  return hipGetLastError();
}

}  // namespace hifihr
