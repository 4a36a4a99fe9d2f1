// What the f32 matrix pipe of THIS chip sustains (the ceiling every GEMM-shaped kernel here is priced against):
// register-only MFMA loops, random operands, 1 or 2 waves per SIMD, 16x16x4 vs 32x32x2.  Build: tools/build_mfma_peak.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(const float* __restrict__ in, float* __restrict__ out, int iters) {
  floatx4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0, 0, 0, 0};
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 256 * i]; b[i] = in[1024 + threadIdx.x + 256 * i]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + k) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(const float* __restrict__ in, float* __restrict__ out, int iters) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + 256 * i]; b[i] = in[1024 + threadIdx.x + 256 * i]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + k) & 3], b[i & 3], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static double run(F launch, double flop) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return flop * 10 / (ms * 1e-3) / 1e12;
}

int main() {
  float *in, *out;
  std::vector<float> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMalloc(&in, 2048 * 4); hipMalloc(&out, 4096 * 256 * 4);
  hipMemcpy(in, h.data(), 2048 * 4, hipMemcpyHostToDevice);
  const int iters = 4000;
  for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
    const int grid = 256 * wgs_per_cu;
    const double f16 = 2.0 * 16 * 16 * 4 * 4 * 16 * iters * 4.0 * grid;    // per MFMA 2*16*16*4 flop; 4 k x 16 acc per iter; 4 waves
    printf("waves/SIMD %d: 16x16x4 x16acc %.1f TF", wgs_per_cu, run([&] { hipLaunchKernelGGL(k16<16>, dim3(grid), dim3(256), 0, 0, in, out, iters); }, f16));
    const double f4 = 2.0 * 16 * 16 * 4 * 4 * 4 * iters * 4.0 * grid;
    printf(" | x4acc %.1f TF", run([&] { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, in, out, iters); }, f4));
    const double f32 = 2.0 * 32 * 32 * 2 * 4 * 4 * iters * 4.0 * grid;
    printf(" | 32x32x2 x4acc %.1f TF", run([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, in, out, iters); }, f32));
    const double f321 = 2.0 * 32 * 32 * 2 * 4 * 1 * iters * 4.0 * grid;
    printf(" | x1acc %.1f TF\n", run([&] { hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, in, out, iters); }, f321));
  }
  return 0;
}
