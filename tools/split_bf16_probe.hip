// PROBE (measurement only, not part of libhifihr.so): one Winograd product of the 512-channel layers,
//   36 x [450 x 512] . [512 x 512]^T   (C[b][m][n] = sum_k A[b][m][k] B[b][n][k], f32 in, f32 out)
// on the bf16 matrix pipe with SPLIT operands: x = hi + lo (two bf16 pieces, "bf16x3": hi.hi + hi.lo + lo.hi, ~2^-17 per product) and
// x = hi + mid + lo (three pieces, "bf16x6": hh + hm + mh + hl + lh + mm, ~2^-24: f32-level), f32 accumulation either way -- against the
// f32-MFMA row-share kernel of libhifihr.so on the same operands and against a float64 reference.  gfx950 has no xf32; its f32 MFMA runs at
// 1/16 of the bf16 rate (MI355X_MICROARCH.md "Matrix cores"), so three / six bf16 products cost 3/16 / 6/16 of the f32 matrix time.
// The headline path stays f32: this file only says what such a `conv_precision` would buy and what it would cost in error.
// Build: tools/build_split_bf16_probe.sh; run on the GPU box: tools/_probe/split_bf16_probe [path to libhifihr.so]
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)

// x -> NP bf16 pieces (round-to-nearest-even each; piece p holds what pieces 0 .. p-1 left over); out[p][n]
template <int NP>
__global__ void split_kernel(const float* __restrict__ x, __bf16* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = x[i];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const __bf16 h = (__bf16)r;
    out[(size_t)p * n + i] = h;
    r -= (float)h;
  }
}

// 128 x 128 tile per 256-thread workgroup (2 x 2 waves of 64 x 64 = 4 x 4 MFMA tiles), 32-deep k-steps, every piece of both operands
// staged through registers into a double-buffered, XOR-swizzled LDS image (rows of 64 bytes), one barrier per k-step.
// NP pieces; products (pa, pb) with pa + pb < NP... i.e. NP = 2: 3 products, NP = 3: 6 products.
template <int NP>
__global__ __launch_bounds__(256) void gemm_split_kernel(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bp, float* __restrict__ C,
                                                         int M, int N, int K, size_t a_piece, size_t b_piece) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kTile = 128 * 64;                            // bytes of one piece of one operand per k-step
  constexpr int kBuf = 2 * NP * kTile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const size_t ab = (size_t)blockIdx.z * M * K, bb = (size_t)blockIdx.z * N * K;
  // staging: thread -> two 16-byte chunks per (operand, piece): chunk c = tid + 256 u: row c >> 2, segment c & 3
  uint4 st[2 * NP][2];
  auto gload = [&](int ks) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int c = tid + 256 * u, row = c >> 2, seg = c & 3;
          const __bf16* src = o == 0 ? Ap + (size_t)p * a_piece + ab + (size_t)min(m0 + row, M - 1) * K + ks * 32 + seg * 8
                                     : Bp + (size_t)p * b_piece + bb + (size_t)(n0 + row) * K + ks * 32 + seg * 8;
          st[o * NP + p][u] = *reinterpret_cast<const uint4*>(src);
        }
  };
  auto lwrite = [&](int buf) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int c = tid + 256 * u, row = c >> 2, seg = c & 3;
          *reinterpret_cast<uint4*>(smem + buf * kBuf + (o * NP + p) * kTile + row * 64 + ((seg ^ ((row >> 2) & 3)) * 16)) = st[o * NP + p][u];
        }
  };
  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / 32;
  gload(0);
  lwrite(0);
  __syncthreads();
  const int swz = ((g ^ ((r >> 2) & 3)) * 16);
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + 1 < nk) gload(ks + 1);
    const char* base = smem + (ks & 1) * kBuf;
    bf16x8 fa[NP][4], fb[NP][4];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[p][i] = *reinterpret_cast<const bf16x8*>(base + (0 * NP + p) * kTile + (wm * 64 + 16 * i + r) * 64 + swz);
        fb[p][i] = *reinterpret_cast<const bf16x8*>(base + (1 * NP + p) * kTile + (wn * 64 + 16 * i + r) * 64 + swz);
      }
    // smallest terms first; the n-indexed operand goes in as MFMA "A": a lane then owns 4 consecutive n of row m = .. + (lane & 15)
#pragma unroll
    for (int s = NP - 1; s >= 0; --s)
#pragma unroll
      for (int pa = 0; pa <= s; ++pa) {
        const int pb = s - pa;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[pb][j], fa[pa][i], acc[i][j], 0, 0, 0);
      }
    if (ks + 1 < nk) lwrite((ks + 1) & 1);
    __syncthreads();
  }
  float* Cb = C + (size_t)blockIdx.z * M * N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + 16 * i + r;
    if (m < M) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4*>(Cb + (size_t)m * N + n0 + wn * 64 + 16 * j + 4 * g) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

__global__ void ref64_kernel(const float* __restrict__ A, const float* __restrict__ B, double* __restrict__ C, int M, int N, int K) {
  const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
  C[(size_t)m * N + n] = s;
}

typedef int (*bgemm_nt_fn)(const float*, const float*, float*, int, int, int, int, void*, size_t, void*);

template <class F>
static float time_us(F f, int n = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < n; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.f / n;
}

static void errors(const char* tag, const float* c_d, const double* ref_d, size_t n) {
  std::vector<float> c(n);
  std::vector<double> ref(n);
  CK(hipMemcpy(c.data(), c_d, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ref.data(), ref_d, n * 8, hipMemcpyDeviceToHost));
  double se = 0, sr = 0, mx = 0;
  for (size_t i = 0; i < n; ++i) { const double d = c[i] - ref[i]; se += d * d; sr += ref[i] * ref[i]; mx = fmax(mx, fabs(d)); }
  const double rms = sqrt(sr / n);
  printf("  %-28s max |err| / rms(C) = %.3e   rms err / rms(C) = %.3e\n", tag, mx / rms, sqrt(se / n) / rms);
}

int main(int argc, char** argv) {
  const int K = 512, N = 512;
  const char* libpath = argc > 1 ? argv[1] : "hifihr_amd/libhifihr.so";
  void* h = dlopen(libpath, RTLD_NOW);
  bgemm_nt_fn f32gemm = h ? (bgemm_nt_fn)dlsym(h, "hifihr_bgemm_nt") : nullptr;
  if (!f32gemm) fprintf(stderr, "(no %s: the f32 kernel is not timed)\n", libpath);
  struct Cfg { int batch, M; const char* what; };
  const Cfg cfgs[] = {{36, 450, "the layer's shape: 576 tiles on 512 slots (1.125 rounds)"}, {32, 512, "512 tiles on 512 slots (one round): the kernel's own rate"}};
  for (const Cfg& cf : cfgs) {
    const int batch = cf.batch, M = cf.M;
    const size_t na = (size_t)batch * M * K, nb = (size_t)batch * N * K, nc = (size_t)batch * M * N;
    std::vector<float> ha(na), hb(nb);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    auto gauss = [&]() { float t = 0; for (int i = 0; i < 6; ++i) t += rnd(); return t * 1.41421f; };      // ~N(0, 1)
    for (auto& v : ha) v = gauss();
    for (auto& v : hb) v = 0.05f * gauss();
    float *A, *B, *C, *C2;
    double* R;
    __bf16 *A2, *B2, *A3, *B3;
    CK(hipMalloc(&A, na * 4)); CK(hipMalloc(&B, nb * 4)); CK(hipMalloc(&C, nc * 4)); CK(hipMalloc(&C2, nc * 4));
    CK(hipMalloc(&R, (size_t)M * N * 8));
    CK(hipMalloc(&A2, na * 2 * 2)); CK(hipMalloc(&B2, nb * 2 * 2)); CK(hipMalloc(&A3, na * 2 * 3)); CK(hipMalloc(&B3, nb * 2 * 3));
    CK(hipMemcpy(A, ha.data(), na * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), nb * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref64_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, B, R, M, N, K);
    const float t_s2 = time_us([&] { hipLaunchKernelGGL(split_kernel<2>, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, 0, A, A2, na); });
    hipLaunchKernelGGL(split_kernel<2>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, 0, B, B2, nb);
    hipLaunchKernelGGL(split_kernel<3>, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, 0, A, A3, na);
    hipLaunchKernelGGL(split_kernel<3>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, 0, B, B3, nb);
    const dim3 grid(N / 128, (M + 127) / 128, batch);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 2 * 128 * 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 3 * 128 * 64));
    const double flop = 2.0 * batch * M * N * K;
    printf("%d x [%d x %d] . [%d x %d]^T  (%s)\n", batch, M, K, N, K, cf.what);
    const float t3 = time_us([&] { hipLaunchKernelGGL(gemm_split_kernel<2>, grid, dim3(256), 2 * 2 * 2 * 128 * 64, 0, A2, B2, C, M, N, K, na, nb); });
    CK(hipDeviceSynchronize());
    printf("  bf16x3 (hi, lo)              %7.1f us   %6.1f TFLOP/s f32-equivalent   (operand split of A: %.1f us, in production the input transform's store)\n",
           t3, flop / t3 / 1e6, t_s2);
    errors("bf16x3 vs float64", C, R, (size_t)M * N);
    const float t6 = time_us([&] { hipLaunchKernelGGL(gemm_split_kernel<3>, grid, dim3(256), 2 * 2 * 3 * 128 * 64, 0, A3, B3, C2, M, N, K, na, nb); });
    CK(hipDeviceSynchronize());
    printf("  bf16x6 (hi, mid, lo)         %7.1f us   %6.1f TFLOP/s f32-equivalent\n", t6, flop / t6 / 1e6);
    errors("bf16x6 vs float64", C2, R, (size_t)M * N);
    if (f32gemm) {
      const float tf = time_us([&] { if (f32gemm(A, B, C, M, N, K, batch, nullptr, 0, nullptr) != 0) { fprintf(stderr, "hifihr_bgemm_nt failed\n"); exit(1); } });
      CK(hipDeviceSynchronize());
      printf("  f32 MFMA (libhifihr.so)      %7.1f us   %6.1f TFLOP/s\n", tf, flop / tf / 1e6);
      errors("f32 MFMA vs float64", C, R, (size_t)M * N);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(C2)); CK(hipFree(R)); CK(hipFree(A2)); CK(hipFree(B2)); CK(hipFree(A3)); CK(hipFree(B3));
  }
  return 0;
}
