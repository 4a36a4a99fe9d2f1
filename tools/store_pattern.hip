// Probe: what does a wave pay per global_store_dwordx4, by address pattern?  (round 6; the row-share GEMM's tile epilogue -- 16 stores per lane,
// 64 KB per workgroup -- takes ~3 500 cycles: ~220 per instruction and wave.)  `waves` waves per workgroup, one workgroup per CU on `cus` CUs;
// every wave issues NST stores of 16 bytes per lane back to back and stamps s_memtime around them (issue time, then until all have retired).
// An instruction writes R rows of 1024 / R contiguous bytes each at a row pitch of P bytes (R = 16, P = 512: the GEMM epilogue at N = 128;
// R = 1: 1 KB contiguous).  Successive instructions move on by R rows (or, `half`: the GEMM's order -- the two 64-byte halves of 16 rows' lines
// in two successive instructions).
// build: hipcc -O3 --offload-arch=gfx950 tools/store_pattern.hip -o tools/_probe/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int NST = 16;
template <int NT>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamps, int rounds, int R, int P, int half, int stride_kb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int lpr = 64 / R;                                            // lanes per row
  char* base = reinterpret_cast<char*>(out) + ((size_t)blockIdx.x * nw + wave) * ((size_t)stride_kb << 10);   // 1 MB apart: every wave on the same channels; 66 KB: spread
  const size_t off = (size_t)(lane / lpr) * P + (size_t)(lane % lpr) * 16;
  float4 v = make_float4(lane, wave, blockIdx.x, 1.f);
  unsigned long long issue = 0, retire = 0;
  for (int it = 0; it < rounds; ++it) {
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int s = 0; s < NST; ++s) {
      const size_t o = half ? off + (size_t)(s >> 1) * R * P + (s & 1) * (1024 / R) : off + (size_t)s * R * P;
      typedef float fx4 __attribute__((ext_vector_type(4)));
      if (NT) __builtin_nontemporal_store(fx4{v.x, v.y, v.z, v.w}, reinterpret_cast<fx4*>(base + o)); else *reinterpret_cast<float4*>(base + o) = v;
      v.x += 1.f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    issue += t1 - t0; retire += t2 - t0;
  }
  if (lane == 0) { atomicAdd(&stamps[0], issue); atomicAdd(&stamps[1], retire); atomicAdd(&stamps[2], 1ull); }
}
int main(int argc, char** argv) {
  const int rounds = 20;
  float* out; unsigned long long* st;
  if (hipMalloc(&out, (size_t)256 * 8 * (1 << 20)) != hipSuccess || hipMalloc(&st, 64) != hipSuccess) return 1;
  struct Cfg { int R, P, half; } cfgs[] = {{16, 512, 1}, {16, 1024, 1}, {16, 2048, 1}, {8, 512, 0}, {8, 2048, 0}, {4, 512, 0}, {4, 1024, 0}, {2, 512, 0}, {1, 1024, 0}};
  for (int nt : {0, 1})
  for (int stride_kb : {1024, 66})
  for (int cus : {256, 32})
    for (int waves : {4})
      for (const Cfg& c : cfgs) {
        if (hipMemset(st, 0, 64) != hipSuccess) return 1;
        if (nt) hipLaunchKernelGGL(k<1>, dim3(cus), dim3(64 * waves), 0, 0, out, st, rounds, c.R, c.P, c.half, stride_kb);
        else hipLaunchKernelGGL(k<0>, dim3(cus), dim3(64 * waves), 0, 0, out, st, rounds, c.R, c.P, c.half, stride_kb);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        unsigned long long h[3];
        if (hipMemcpy(h, st, 24, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        const double n = (double)h[2] * rounds * NST;
        printf("%s region stride %4d KB, %3d CUs x %d waves, %2d rows x %4d B at pitch %4d%s: %4.0f cycles per store to issue, %4.0f until retired -> %.1f B/clk/CU\n", nt ? "nontemporal" : "plain      ", stride_kb, cus, waves, c.R, 1024 / c.R,
               c.P, c.half ? " (halves)" : "         ", h[0] / n, h[1] / n, waves * 1024.0 / (h[1] / n));
      }
  return 0;
}
