// Does the A-operand access pattern of bgemm_nt_rows_kernel cost HBM efficiency?  A workgroup streams tiles of 128 rows x K floats as K/32
// chunks of [128 rows][32 floats] (what the loader waves fetch): row-major A[M][K] makes a chunk 128 separate 128-byte segments 4 K bytes
// apart; a chunk-tiled layout makes it 16 KB contiguous.  Plain 16-byte loads, 8 in flight per lane, nothing else in the kernel.
// build: hipcc -O3 --offload-arch=gfx950 tools/stream_pattern.hip -o tools/_probe/stream_pattern ; run: tools/_probe/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int K, bool TILED, bool STORE>
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ A, float4* __restrict__ C, long rows_total, float* __restrict__ sink) {
  constexpr int NCH = K / 32;
  const long tiles = rows_total / 128;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    for (int lc = 0; lc < NCH; ++lc) {
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int piece = threadIdx.x + 256 * i;            // 1024 pieces of 16 bytes = 128 rows x 8 segments
        const int row = piece >> 3, seg = piece & 7;
        size_t idx;                                          // in float4 units
        if (TILED) idx = ((size_t)t * NCH + lc) * 1024 + piece;
        else idx = ((size_t)t * 128 + row) * (K / 4) + lc * 8 + seg;
        v[i] = A[idx];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
    if (STORE) {                                             // the tile's 128 x 128 output: 4096 float4, 16 per lane
#pragma unroll
      for (int i = 0; i < 16; ++i) C[(size_t)t * 4096 + threadIdx.x + 256 * i] = acc;
    }
  }
  if (acc.x == 12345.678f) sink[0] = acc.y + acc.z + acc.w;
}

template <int K, bool TILED, bool STORE>
static void run(const char* name, const float4* A, float4* C, long rows, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<K, TILED, STORE>), dim3(256 * 2), dim3(256), 0, 0, A, C, rows, sink);
  hipEventRecord(e0);
  const int n = 20;
  for (int w = 0; w < n; ++w) hipLaunchKernelGGL((probe<K, TILED, STORE>), dim3(256 * 2), dim3(256), 0, 0, A, C, rows, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / n, bytes = (double)rows * K * 4 + (STORE ? (double)rows * 128 * 4 : 0.0);
  printf("%-44s %8.1f us  %6.2f TB/s\n", name, us, bytes / us / 1e6);
}

int main() {
  const long rows = 36L * 1568 * 4 / 128 * 128;             // ~4 launches' worth of the 128-channel layers' V rows
  float4 *A, *C; float* sink;
  hipMalloc(&A, (size_t)rows * 512 * 4); hipMalloc(&C, (size_t)rows * 128 * 4); hipMalloc(&sink, 4);
  hipMemset(A, 0, (size_t)rows * 512 * 4);
  run<128, false, false>("K=128 row-major, loads only", A, C, rows, sink);
  run<128, true, false>("K=128 chunk-tiled, loads only", A, C, rows, sink);
  run<128, false, true>("K=128 row-major, loads + tile stores", A, C, rows, sink);
  run<128, true, true>("K=128 chunk-tiled, loads + tile stores", A, C, rows, sink);
  run<256, false, false>("K=256 row-major, loads only", A, C, rows, sink);
  run<256, true, false>("K=256 chunk-tiled, loads only", A, C, rows, sink);
  run<512, false, false>("K=512 row-major, loads only", A, C, rows, sink);
  run<512, true, false>("K=512 chunk-tiled, loads only", A, C, rows, sink);
  return 0;
}
