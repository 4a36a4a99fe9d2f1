// Batched fp32 GEMM for gfx950 on the f32 matrix cores (v_mfma_f32_16x16x4_f32): the 16 independent products a Winograd
// F(2x2, 3x3) layer consists of (wino.hip).  Replaces the GEMM half of what the reference's encoder dispatches to one vendor
// library call per convolution (reference network/res_encoder.py:364-373: conv2d forward / backward-data / backward-weight of the
// stride-1 3x3 layers with >= 128 channels); round 1 ran these products on the implicit-GEMM gather kernel (conv.hip) or on
// rocBLAS.  Plain GEMMs need no tap bookkeeping, so this kernel spends its instructions on the matrix pipe:
//
//   bgemm_nt_kernel   C[p][m][n] = sum_k A[p][m][k] B[p][n][k]      forward / backward-data  (A = V[16][T][C], B = U[16][K][C])
//   bgemm_tn_kernel   C[z][p][m][n] = sum_{t in split z} A[p][t][m] B[p][t][n]   backward-weight  (A = Y'[16][T][K], B = V[16][T][C]);
//                     the reduction over tiles t is split over blockIdx (z) into slabs that wino_dw_transform sums while it reads
//                     them -- no atomics, no zero-initialised accumulator, bit-reproducible.
//
// Structure (both): 128x128 (or 64-wide) macro-tile per 256-thread workgroup, 2 x 2 waves of 64x64, 4 x 4 MFMA 16x16 tiles with
// independent accumulators per wave (the 16x16x4 form issues every 32 cycles with a 40-cycle dependent latency: 16 accumulators
// keep the pipe full); 32-deep K chunks = 128 MFMAs = 4096 matrix-pipe cycles per wave between two barriers; operands go
// global -> LDS with direct-to-LDS loads (global_load_lds_dwordx4: no VGPR staging, no ds_write pass), two LDS stages of 32 KB,
// so the loads of chunk c + 1 have a whole chunk of MFMAs to land; two workgroups share a CU and cover each other's barrier.
// LDS images: NT rows are 128 B (32 k), XOR-swizzled at load time on the SOURCE address (an LDS-DMA destination is lane-linear) so
// that the fragment ds_read_b128 (lane (r, g): row r, k-segment g) are bank-conflict-free; the lane's float4 holds k = 4g .. 4g+3,
// i.e. the operands of 4 consecutive MFMAs whose hardware k index g stands for k = 4g + s (both operands permuted alike: any
// bijection of k is a valid reduction order).  TN rows are the 128 output columns of one t (512 B), read with one ds_read_b128
// per k-step for the lane-indexed operand (lane r supplies rows 4r .. 4r+3 to MFMA tiles 0 .. 3: a row permutation the epilogue
// undoes) and 4 ds_read_b32 for the register-indexed one.
// MFMA orientation: the D tile has its rows in registers (4 consecutive per lane) and its column on the lane, so the operand that
// indexes the CONTIGUOUS output dimension n is fed as MFMA "A": every lane then owns 4 consecutive n and stores them as one float4.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "hifihr_internal.h"
#include "lds_dma.h"

namespace hifihr {

// ------------------------------------------------------------------------------------------------
// NT: C[m][n] = sum_k A[m][k] B[n][k]   (both operands K-contiguous; K % 32 == 0, N % BN == 0, any M)
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256) void bgemm_nt_kernel(BgemmArgs a) {
  constexpr int WM = BM / 2, WN = BN / 2;        // per-wave sub-tile (waves 2 x 2)
  constexpr int TI = WN / 16, TJ = WM / 16;      // MFMA tiles: i over n (D rows, registers), j over m (D columns, lanes)
  constexpr int STAGE = (BM + BN) * 32;          // floats per LDS stage
  constexpr int PA = BM / 32, PB = BN / 32;      // 1 KiB pieces (8 rows x 128 B) per wave per chunk
  __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;

  int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tiles_pb = a.tiles_m * a.tiles_n;
  const int p = wg / tiles_pb;
  wg -= p * tiles_pb;
  const int tn = wg / a.tiles_m, tm = wg - tn * a.tiles_m;      // m-tiles fastest: neighbours share the B panel
  const int m0 = tm * BM, n0 = tn * BN;
  const float* __restrict__ A = a.A + (size_t)p * a.sa;
  const float* __restrict__ B = a.B + (size_t)p * a.sb;
  float* __restrict__ C = a.C + (size_t)p * a.sc;

  // loader: piece i of this wave covers rows 8 * (wave + 4 i) .. + 7 of the A (i < PA) or B tile; lane -> (row lane >> 3, physical
  // 16-byte segment lane & 7), which holds logical segment (lane & 7) ^ ((row >> 1) & 7).  Rows past M read row M - 1 (discarded).
  unsigned goff[PA + PB];
#pragma unroll
  for (int i = 0; i < PA + PB; ++i) {
    const bool isA = i < PA;
    const int row = 8 * (wave + 4 * (isA ? i : i - PA)) + (lane >> 3);
    const int seg = (lane & 7) ^ ((row >> 1) & 7);
    int grow = (isA ? m0 : n0) + row;
    const int lim = isA ? a.M : a.N;
    grow = grow < lim ? grow : lim - 1;
    goff[i] = (unsigned)grow * (unsigned)(isA ? a.lda : a.ldb) + seg * 4;
  }
  auto issue = [&](int chunk, int stage) {
    float* base = lds + stage * STAGE;
#pragma unroll
    for (int i = 0; i < PA + PB; ++i) {
      const bool isA = i < PA;
      const float* src = (isA ? A : B) + goff[i] + chunk * 32;
      float* dst = base + (isA ? 0 : BM * 32) + 256 * (wave + 4 * (isA ? i : i - PA));
      HIFIHR_GLDS16(src, dst, lane);
    }
  };

  floatx4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int nch = a.K / 32;
  const int sw = (r >> 1) & 7;
  const int offA = (wm * WM + r) * 32, offB = BM * 32 + (wn * WN + r) * 32;
  issue(0, 0);
  HIFIHR_WAIT_LOADS();
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const float* s = lds + (c & 1) * STAGE;
    float4 fa[TJ][2], fb[TI][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ps = ((g + 4 * h) ^ sw) * 4;
#pragma unroll
      for (int i = 0; i < TI; ++i) fb[i][h] = *reinterpret_cast<const float4*>(s + offB + i * 512 + ps);
#pragma unroll
      for (int j = 0; j < TJ; ++j) fa[j][h] = *reinterpret_cast<const float4*>(s + offA + j * 512 + ps);
    }
    if (c + 1 < nch) issue(c + 1, (c + 1) & 1);          // lands during the 128 MFMAs below
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            const float bv = k4 == 0 ? fb[i][h].x : k4 == 1 ? fb[i][h].y : k4 == 2 ? fb[i][h].z : fb[i][h].w;
            const float av = k4 == 0 ? fa[j][h].x : k4 == 1 ? fa[j][h].y : k4 == 2 ? fa[j][h].z : fa[j][h].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, acc[i][j], 0, 0, 0);
          }
    }
    HIFIHR_PIN();
    HIFIHR_WAIT_LOADS();
    __syncthreads();
  }

  // D[i][j]: register e of lane (r, g) = C[m = .. 16 j + r][n = .. 16 i + 4 g + e]
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int m = m0 + wm * WM + 16 * j + r;
    if (m < a.M) {
      float* row = C + (size_t)m * a.ldc + n0 + wn * WN + 4 * g;
#pragma unroll
      for (int i = 0; i < TI; ++i)
        *reinterpret_cast<float4*>(row + 16 * i) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TN: C[z][m][n] = sum_{t in split z} A[t][m] B[t][n]   (both operands stored t-major; M % BM == 0, N % BN == 0, any T)
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256) void bgemm_tn_kernel(BgemmArgs a) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TI = WN / 16, TJ = WM / 16;      // i over n (registers), j over m (lanes, row-permuted: m = TJ * r + j)
  constexpr int STAGE = (BM + BN) * 32;
  constexpr int PA = BM / 32, PB = BN / 32;      // pieces per wave per chunk
  constexpr int RA = 256 / BM, RB = 256 / BN;    // t rows per 1 KiB piece
  __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;

  int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tiles_pb = a.tiles_m * a.tiles_n;
  const int per_split = tiles_pb * a.batch;
  const int z = wg / per_split;
  wg -= z * per_split;
  const int p = wg / tiles_pb;
  wg -= p * tiles_pb;
  const int tn = wg / a.tiles_m, tm = wg - tn * a.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const float* __restrict__ A = a.A + (size_t)p * a.sa + m0;
  const float* __restrict__ B = a.B + (size_t)p * a.sb + n0;
  float* __restrict__ C = a.C + (size_t)z * a.sc_split + (size_t)p * a.sc;

  const int nch_total = (a.K + 31) / 32;
  const int c_lo = z * a.cps, c_hi = min(c_lo + a.cps, nch_total);

  // piece i of this wave: t rows R * (wave + 4 i') .. of the chunk, 16 bytes per lane along the tile's columns
  const int tA = lane / (BM / 4), cA = (lane % (BM / 4)) * 4;
  const int tB = lane / (BN / 4), cB = (lane % (BN / 4)) * 4;
  auto issue = [&](int chunk, int stage) {
    float* base = lds + stage * STAGE;
    const int t0 = chunk * 32;
    if (t0 + 32 <= a.K) {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int q = wave + 4 * i;
        HIFIHR_GLDS16(A + (size_t)(t0 + RA * q + tA) * a.lda + cA, base + 256 * q, lane);
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int q = wave + 4 * i;
        HIFIHR_GLDS16(B + (size_t)(t0 + RB * q + tB) * a.ldb + cB, base + BM * 32 + 256 * q, lane);
      }
    } else {
      // ragged last chunk (T % 32 != 0): rows past T must contribute zeros, which an LDS-DMA cannot produce
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int q = wave + 4 * i, t = t0 + RA * q + tA;
        const float4 v = t < a.K ? *reinterpret_cast<const float4*>(A + (size_t)t * a.lda + cA) : zero;
        *reinterpret_cast<float4*>(base + 256 * q + 4 * lane) = v;
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int q = wave + 4 * i, t = t0 + RB * q + tB;
        const float4 v = t < a.K ? *reinterpret_cast<const float4*>(B + (size_t)t * a.ldb + cB) : zero;
        *reinterpret_cast<float4*>(base + BM * 32 + 256 * q + 4 * lane) = v;
      }
    }
  };

  floatx4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  if (c_lo < c_hi) {
    issue(c_lo, 0);
    HIFIHR_PIN();
    HIFIHR_WAIT_LOADS();
    __syncthreads();
  }
  for (int c = c_lo; c < c_hi; ++c) {
    const float* s = lds + ((c - c_lo) & 1) * STAGE;
    float fa[8][TJ], fb[8][TI];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float* rowA = s + (4 * k + g) * BM + wm * WM + TJ * r;
      const float* rowB = s + BM * 32 + (4 * k + g) * BN + wn * WN + r;
      if constexpr (TJ == 4) {
        const float4 v = *reinterpret_cast<const float4*>(rowA);
        fa[k][0] = v.x; fa[k][1] = v.y; fa[k][2] = v.z; fa[k][3] = v.w;
      } else {
        const float2 v = *reinterpret_cast<const float2*>(rowA);
        fa[k][0] = v.x; fa[k][1] = v.y;
      }
#pragma unroll
      for (int i = 0; i < TI; ++i) fb[k][i] = rowB[16 * i];
    }
    if (c + 1 < c_hi) issue(c + 1, (c + 1 - c_lo) & 1);
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[k][i], fa[k][j], acc[i][j], 0, 0, 0);
    HIFIHR_PIN();
    HIFIHR_WAIT_LOADS();
    __syncthreads();
  }

  // D[i][j]: register e of lane (r, g) = C[m = .. TJ r + j][n = .. 16 i + 4 g + e]
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    float* row = C + (size_t)(m0 + wm * WM + TJ * r + j) * a.ldc + n0 + wn * WN + 4 * g;
#pragma unroll
    for (int i = 0; i < TI; ++i)
      *reinterpret_cast<float4*>(row + 16 * i) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  }
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised form (both layouts): 4 MFMA waves + NLOAD loader waves per workgroup, one workgroup per CU.
// Measured on MI355X (tools/time_gemm.py, round 2): in the kernels above every MFMA wave also issues its share of the LDS-DMA
// loads, and an LDS-DMA instruction costs its wave 60-185 cycles of issue time (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"):
// 8 per chunk in front of 128 MFMAs (4096 matrix-pipe cycles) leave the pipe idle ~35 % of the time, and two co-resident workgroups
// run in lockstep and stall together.  Here the loads are issued by waves that do nothing else, so the MFMA waves' instruction
// stream is ds_read + MFMA only:
//   LDS: 4 stages of (BM + BN) x 32 floats (128 KB at 128x128).
//   loader, iteration c : issue chunk c + 3 into stage (c + 3) & 3 (last read in iteration c - 1, released by barrier c - 1);
//                         wait until chunk c + 2 has landed (counted vmcnt: only chunk c + 3's pieces may stay in flight);
//                         barrier c.                                       -> a load has two iterations (~8000 cycles) to land
//   MFMA wave, iteration c : MFMAs of half 0 of chunk c while the fragments of half 1 load from LDS, MFMAs of half 1 while the
//                         fragments of half 0 of chunk c + 1 load (landed: confirmed at barrier c - 1); barrier c.
//   The barrier is a raw s_barrier: __syncthreads() would drain the loader's in-flight DMA (its fence waits vmcnt(0)).
// ------------------------------------------------------------------------------------------------
// HIFIHR_GEMM_STAMP (diagnostic build, tools/build_gemm_probe.sh): wave 0 of every workgroup adds to g_gemm_stamp
// [0] shader cycles in the main loop, [1] 100 MHz real-time ticks of the same span (-> clock held under load), [2] chunks,
// [3] cycles spent at the per-chunk barrier, [4] waves counted, [5] cycles from kernel entry to the end of the epilogue
#if defined(HIFIHR_GEMM_STAMP)
__device__ unsigned long long g_gemm_stamp[8];
#endif

template <int BM, int BN, bool TNL, int NLOAD>
__global__ __launch_bounds__(256 + 64 * NLOAD) void bgemm_ws_kernel(BgemmArgs a) {
#if defined(HIFIHR_GEMM_STAMP)
  const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
#endif
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TI = WN / 16, TJ = WM / 16;
  constexpr int STAGE = (BM + BN) * 32;
  constexpr int NPA = BM / 8, NP = (BM + BN) / 8;          // 1 KiB pieces per chunk: A, total
  constexpr int PL = NP / NLOAD;                            // pieces per loader wave per chunk
  static_assert(NP % NLOAD == 0 && (PL == 8 || PL == 16 || PL == 32), "loader split");
  constexpr int RA = 256 / BM, RB = 256 / BN;               // TN: t rows per piece
  __shared__ __attribute__((aligned(1024))) float lds[4 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tiles_pb = a.tiles_m * a.tiles_n;
  const int per_split = tiles_pb * a.batch;
  const int z = wg / per_split;
  wg -= z * per_split;
  const int p = wg / tiles_pb;
  wg -= p * tiles_pb;
  const int tn = wg / a.tiles_m, tm = wg - tn * a.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const float* __restrict__ A = a.A + (size_t)p * a.sa;
  const float* __restrict__ B = a.B + (size_t)p * a.sb;
  const int nch_total = TNL ? (a.K + 31) / 32 : a.K / 32;
  const int c_lo = z * a.cps, c_hi = min(c_lo + a.cps, nch_total);
  const int nch = c_hi - c_lo;
  if (nch <= 0) return;                                      // (uniform over the workgroup)

  if (wave >= 4) {
    HIFIHR_SET_LOADER_PRIO();
    // ---------------- loader ----------------
    const int l = wave - 4;
    unsigned goff[PL];
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int q = l + NLOAD * i;
      const bool isA = q < NPA;
      const int qq = isA ? q : q - NPA;
      if (TNL) {
        const int bw = isA ? BM : BN, rp = isA ? RA : RB;
        const int t = rp * qq + lane / (bw / 4), col = (lane % (bw / 4)) * 4;
        goff[i] = (unsigned)t * (unsigned)(isA ? a.lda : a.ldb) + (isA ? m0 : n0) + col;
      } else {
        const int row = 8 * qq + (lane >> 3);
        const int seg = (lane & 7) ^ ((row >> 1) & 7);
        int grow = (isA ? m0 : n0) + row;
        const int lim = isA ? a.M : a.N;
        grow = grow < lim ? grow : lim - 1;
        goff[i] = (unsigned)grow * (unsigned)(isA ? a.lda : a.ldb) + seg * 4;
      }
    }
    auto issue = [&](int c) {                                // chunk c of this split -> stage c & 3
      float* base = lds + (c & 3) * STAGE;
      const int cg = c_lo + c;
      if (!TNL || cg * 32 + 32 <= a.K) {
        const size_t adv_a = TNL ? (size_t)cg * 32 * a.lda : (size_t)cg * 32;
        const size_t adv_b = TNL ? (size_t)cg * 32 * a.ldb : (size_t)cg * 32;
#pragma unroll
        for (int i = 0; i < PL; ++i) {
          const int q = l + NLOAD * i;
          const bool isA = q < NPA;
          HIFIHR_GLDS16((isA ? A + adv_a : B + adv_b) + goff[i], base + (isA ? 256 * q : BM * 32 + 256 * (q - NPA)), lane);
        }
      } else {
        // ragged last chunk of the t range (TN): rows past T contribute zeros, which an LDS-DMA cannot produce
#pragma unroll
        for (int i = 0; i < PL; ++i) {
          const int q = l + NLOAD * i;
          const bool isA = q < NPA;
          const int qq = isA ? q : q - NPA;
          const int bw = isA ? BM : BN, rp = isA ? RA : RB;
          const int t = cg * 32 + rp * qq + lane / (bw / 4);
          const float* src = (isA ? A : B) + (size_t)cg * 32 * (isA ? a.lda : a.ldb) + goff[i];
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (t < a.K) v = *reinterpret_cast<const float4*>(src);
          *reinterpret_cast<float4*>(base + (isA ? 256 * q : BM * 32 + 256 * qq) + 4 * lane) = v;
        }
        HIFIHR_WAIT_VM(0);
        HIFIHR_WAIT_LGKM0();
      }
    };
    issue(0);
    if (nch > 1) issue(1);
    if (nch > 2) issue(2);
    // chunks 0 and 1 landed: only chunk 2's pieces may stay in flight
    if (nch > 2) { if (PL == 16) HIFIHR_WAIT_VM(16); else if (PL == 8) HIFIHR_WAIT_VM(8); else HIFIHR_WAIT_VM(32); }
    else HIFIHR_WAIT_VM(0);
    HIFIHR_RAW_BARRIER();                                    // barrier -1
    for (int c = 0; c < nch; ++c) {
      if (c + 3 < nch) {
        issue(c + 3);
        if (PL == 16) HIFIHR_WAIT_VM(16); else if (PL == 8) HIFIHR_WAIT_VM(8); else HIFIHR_WAIT_VM(32);
      } else {
        HIFIHR_WAIT_VM(0);
      }
      HIFIHR_RAW_BARRIER();                                  // barrier c
    }
    return;
  }

  // ---------------- MFMA waves ----------------
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;
  floatx4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  // fragments of one half chunk (16 k): fm[j][s] = operand of D column tile j (lane-indexed, m) at k-step s, fn[i][s] likewise for n
  float fm[2][TJ][4], fn[2][TI][4];
  auto read_half = [&](int c, int h, int slot) {
    const float* s = lds + (c & 3) * STAGE;
    if (TNL) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* rowA = s + (16 * h + 4 * k + g) * BM + wm * WM + TJ * r;
        const float* rowB = s + BM * 32 + (16 * h + 4 * k + g) * BN + wn * WN + r;
        if constexpr (TJ == 4) {
          const float4 v = *reinterpret_cast<const float4*>(rowA);
          fm[slot][0][k] = v.x; fm[slot][1][k] = v.y; fm[slot][2][k] = v.z; fm[slot][3][k] = v.w;
        } else {
          const float2 v = *reinterpret_cast<const float2*>(rowA);
          fm[slot][0][k] = v.x; fm[slot][1][k] = v.y;
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) fn[slot][i][k] = rowB[16 * i];
      }
    } else {
      const int ps = ((g + 4 * h) ^ ((r >> 1) & 7)) * 4;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(s + BM * 32 + (wn * WN + 16 * i + r) * 32 + ps);
        fn[slot][i][0] = v.x; fn[slot][i][1] = v.y; fn[slot][i][2] = v.z; fn[slot][i][3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(s + (wm * WM + 16 * j + r) * 32 + ps);
        fm[slot][j][0] = v.x; fm[slot][j][1] = v.y; fm[slot][j][2] = v.z; fm[slot][j][3] = v.w;
      }
    }
  };
  auto mfma_half = [&](int slot) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fn[slot][i][k], fm[slot][j][k], acc[i][j], 0, 0, 0);
  };

  HIFIHR_RAW_BARRIER();                                      // barrier -1: chunks 0 and 1 are in LDS
  read_half(0, 0, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {                              // (see the end of the loop body)
#pragma unroll
    for (int i = 0; i < TI; ++i) HIFIHR_TOUCH(fn[0][i][k]);
#pragma unroll
    for (int j = 0; j < TJ; ++j) HIFIHR_TOUCH(fm[0][j][k]);
  }
#if defined(HIFIHR_GEMM_STAMP)
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_bar = 0;
#endif
  for (int c = 0; c < nch; ++c) {
    read_half(c, 1, 1);
    HIFIHR_PIN();                                            // reads FIRST: the scheduler otherwise sinks them below the MFMA block
    mfma_half(0);
    HIFIHR_PIN();
    read_half(c + 1, 0, 0);                                  // unconditional (past the last chunk: a stale stage, never used): a branch
    HIFIHR_PIN();                                            // here makes the waits in front of the next MFMA block conservative
    mfma_half(1);
    HIFIHR_PIN();
    // "use" the prefetched fragments here, where their data has long arrived: hipcc's waitcnt pass is imprecise across the loop
    // back-edge (it emitted lgkmcnt(0) in front of the next iteration's first MFMA, i.e. waited for the reads issued just before
    // it); with nothing pending at the loop head the waits inside the iteration are exact counted ones
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int i = 0; i < TI; ++i) HIFIHR_TOUCH(fn[0][i][k]);
#pragma unroll
      for (int j = 0; j < TJ; ++j) HIFIHR_TOUCH(fm[0][j][k]);
    }
#if defined(HIFIHR_GEMM_STAMP)
    HIFIHR_TOUCH(acc[0][0][0]);
    const unsigned long long st_b0 = __builtin_amdgcn_s_memtime();
#endif
    HIFIHR_RAW_BARRIER();                                    // barrier c
#if defined(HIFIHR_GEMM_STAMP)
    st_bar += __builtin_amdgcn_s_memtime() - st_b0;
#endif
  }
#if defined(HIFIHR_GEMM_STAMP)
  const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
#endif

  float* __restrict__ C = a.C + (size_t)z * a.sc_split + (size_t)p * a.sc;
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int m = m0 + wm * WM + (TNL ? TJ * r + j : 16 * j + r);
    if (m < a.M) {
      float* row = C + (size_t)m * a.ldc + n0 + wn * WN + 4 * g;
#pragma unroll
      for (int i = 0; i < TI; ++i)
        *reinterpret_cast<float4*>(row + 16 * i) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
#if defined(HIFIHR_GEMM_STAMP)
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicAdd(&g_gemm_stamp[0], st_t1 - st_t0); atomicAdd(&g_gemm_stamp[1], st_r1 - st_r0); atomicAdd(&g_gemm_stamp[2], (unsigned long long)nch);
    atomicAdd(&g_gemm_stamp[3], st_bar); atomicAdd(&g_gemm_stamp[4], 1ull); atomicAdd(&g_gemm_stamp[5], __builtin_amdgcn_s_memtime() - st_entry);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Persistent, balanced NT form (forward / backward-data products: 4-16 chunks per tile).
// Measured on MI355X (tools/gemm_stamp.py, round 2) on the 512-channel product: a tile's main loop is 76 600 cycles, its first
// loads and its store tail 13 300 more (15 %), and 832 tiles on 256 CUs leave the last quarter-round mostly idle (3.25 rounds):
// bgemm_ws_kernel reaches 0.58 of the matrix peak there while the same inner loop runs at 0.75 on the long reductions of the
// weight gradient.  Here ONE workgroup per CU (grid = CUs) walks an equal share of the (tile, chunk) sequence:
//   * the loader waves' chunk stream runs across tile boundaries, so the first chunks of the next tile land while the MFMA waves
//     still compute / store the current one (no per-tile load latency);
//   * every workgroup gets ceil(tiles x chunks / CUs) chunks (stream-K): nobody idles in a last partial round.  A share starts and
//     ends mid-tile, so a tile is split between at most TWO neighbouring workgroups (shares are at least one tile long): the one
//     that owns the tile's TAIL (its share starts there, so it is done with it early) parks its partial sums in a slab of its own
//     and raises a flag; the one that owns the HEAD reaches it at the END of its share, adds the slab and stores the tile.
//     Hand-off per MFMA wave (each wave owns a 64 x 64 quarter of the tile on both sides, no workgroup barrier involved, which
//     keeps the loader waves' barrier count untouched): plain 16-byte slab stores, s_waitcnt vmcnt(0), agent-scope release, flag
//     store -- relaxed poll of the flag, agent-scope acquire, plain loads (cdna_hip_programming.md Guideline 16).  The consumer
//     zeroes the flag again: the workspace is zero-initialised once and self-cleaning, like conv.hip's.
//     No deadlock: workgroup w only ever waits for w + 1, the last one waits for nobody, and a producer writes its slab before
//     anything else, so whatever order the dispatcher picks, the highest unfinished workgroup can always complete.
// ------------------------------------------------------------------------------------------------
template <int NLOAD>
__global__ __launch_bounds__(256 + 64 * NLOAD) void bgemm_nt_sk_kernel(BgemmArgs a, float* __restrict__ slabs, unsigned* __restrict__ flags) {
  constexpr int BM = 128, BN = 128, WM = 64, WN = 64, TI = 4, TJ = 4;
  constexpr int STAGE = (BM + BN) * 32;
  constexpr int NPA = BM / 8, NP = (BM + BN) / 8;
  constexpr int PL = NP / NLOAD;
  static_assert(PL == 8 || PL == 16 || PL == 32, "loader split");
  __shared__ __attribute__((aligned(1024))) float lds[4 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch = a.K / 32;
  const int tiles_pb = a.tiles_m * a.tiles_n;
  const long total = (long)tiles_pb * a.batch * nch;
  const int G = (int)gridDim.x;
  const int per = (int)((total + G - 1) / G);
  const int wg = xcd_remap((int)blockIdx.x, G);               // neighbouring shares (the two halves of a split tile) on one XCD
  const long it0 = (long)wg * per;
  const long it1 = it0 + per < total ? it0 + per : total;
  const int n_it = (int)(it1 - it0);
  if (n_it <= 0) return;
  const int tile0 = (int)(it0 / nch), c0 = (int)(it0 - (long)tile0 * nch);

  if (wave >= 4) {
    HIFIHR_SET_LOADER_PRIO();
    // ---------------- loader ----------------
    const int l = wave - 4;
    unsigned goff[PL];
    const float* __restrict__ Ap = a.A;
    const float* __restrict__ Bp = a.B;
    auto setup = [&](int tile) {
      const int p = tile / tiles_pb, rem = tile - p * tiles_pb;
      const int tn = rem / a.tiles_m, tm = rem - tn * a.tiles_m;
      Ap = a.A + (size_t)p * a.sa; Bp = a.B + (size_t)p * a.sb;
#pragma unroll
      for (int i = 0; i < PL; ++i) {
        const int q = l + NLOAD * i;
        const bool isA = q < NPA;
        const int row = 8 * (isA ? q : q - NPA) + (lane >> 3);
        const int seg = (lane & 7) ^ ((row >> 1) & 7);
        int grow = (isA ? tm * BM : tn * BN) + row;
        const int lim = isA ? a.M : a.N;
        grow = grow < lim ? grow : lim - 1;
        goff[i] = (unsigned)grow * (unsigned)(isA ? a.lda : a.ldb) + seg * 4;
      }
    };
    int tile = tile0, c = c0;
    setup(tile);
    auto issue = [&](int j) {                                  // local iteration j -> stage j & 3; advances (tile, c)
      float* base = lds + (j & 3) * STAGE;
#pragma unroll
      for (int i = 0; i < PL; ++i) {
        const int q = l + NLOAD * i;
        const bool isA = q < NPA;
        HIFIHR_GLDS16((isA ? Ap : Bp) + goff[i] + c * 32, base + (isA ? 256 * q : BM * 32 + 256 * (q - NPA)), lane);
      }
      if (++c == nch) { c = 0; ++tile; if (j + 1 < n_it) setup(tile); }
    };
    issue(0);
    if (n_it > 1) issue(1);
    if (n_it > 2) issue(2);
    if (n_it > 2) { if (PL == 16) HIFIHR_WAIT_VM(16); else if (PL == 8) HIFIHR_WAIT_VM(8); else HIFIHR_WAIT_VM(32); }
    else HIFIHR_WAIT_VM(0);
    HIFIHR_RAW_BARRIER();
    for (int j = 0; j < n_it; ++j) {
      if (j + 3 < n_it) {
        issue(j + 3);
        if (PL == 16) HIFIHR_WAIT_VM(16); else if (PL == 8) HIFIHR_WAIT_VM(8); else HIFIHR_WAIT_VM(32);
      } else {
        HIFIHR_WAIT_VM(0);
      }
      HIFIHR_RAW_BARRIER();
    }
    return;
  }

  // ---------------- MFMA waves ----------------
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, g = lane >> 4;
  floatx4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  float fm[2][TJ][4], fn[2][TI][4];
  auto read_half = [&](int j, int h, int slot) {
    const float* s = lds + (j & 3) * STAGE;
    const int ps = ((g + 4 * h) ^ ((r >> 1) & 7)) * 4;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(s + BM * 32 + (wn * WN + 16 * i + r) * 32 + ps);
      fn[slot][i][0] = v.x; fn[slot][i][1] = v.y; fn[slot][i][2] = v.z; fn[slot][i][3] = v.w;
    }
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) {
      const float4 v = *reinterpret_cast<const float4*>(s + (wm * WM + 16 * jj + r) * 32 + ps);
      fm[slot][jj][0] = v.x; fm[slot][jj][1] = v.y; fm[slot][jj][2] = v.z; fm[slot][jj][3] = v.w;
    }
  };
  auto mfma_half = [&](int slot) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fn[slot][i][k], fm[slot][jj][k], acc[i][jj], 0, 0, 0);
  };
  // slab layout: [workgroup][wave][reg 0..63][lane] float4-granular: quarter-tile of this wave in its own register order
  auto slab_of = [&](int w) { return slabs + ((size_t)w * 4 + wave) * (64 * 64); };

  HIFIHR_RAW_BARRIER();
  read_half(0, 0, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int i = 0; i < TI; ++i) HIFIHR_TOUCH(fn[0][i][k]);
#pragma unroll
    for (int jj = 0; jj < TJ; ++jj) HIFIHR_TOUCH(fm[0][jj][k]);
  }
  int tile = tile0, c = c0, seg_c0 = c0;
  for (int j = 0; j < n_it; ++j) {
    read_half(j, 1, 1);
    HIFIHR_PIN();
    mfma_half(0);
    HIFIHR_PIN();
    read_half(j + 1, 0, 0);
    HIFIHR_PIN();
    mfma_half(1);
    HIFIHR_PIN();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int i = 0; i < TI; ++i) HIFIHR_TOUCH(fn[0][i][k]);
#pragma unroll
      for (int jj = 0; jj < TJ; ++jj) HIFIHR_TOUCH(fm[0][jj][k]);
    }
    HIFIHR_RAW_BARRIER();                                      // stage j & 3 goes back to the loader before the tile's epilogue
    const bool tile_end = c == nch - 1;
    if (tile_end || j == n_it - 1) {
      if (seg_c0 > 0) {
        // this share started inside the tile: park the partial sums for the owner of the tile's head
        float* sl = slab_of(wg);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int jj = 0; jj < TJ; ++jj)
            *reinterpret_cast<float4*>(sl + ((i * TJ + jj) * 64 + lane) * 4) = make_float4(acc[i][jj][0], acc[i][jj][1], acc[i][jj][2], acc[i][jj][3]);
#if !defined(HIFIHR_HOSTSIM)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(flags + wg * 4 + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        HIFIHR_WAVE_SYNC();
        if (lane == 0) __atomic_store_n(flags + wg * 4 + wave, 1u, __ATOMIC_SEQ_CST);
#endif
      } else {
        if (!tile_end) {
          // this share ends inside the tile: the next workgroup owns the rest and has parked it long ago
          unsigned* fl = flags + (wg + 1) * 4 + wave;
#if !defined(HIFIHR_HOSTSIM)
          while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(2);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
          while (__atomic_load_n(fl, __ATOMIC_SEQ_CST) == 0u) ::hostsim::yield_now();
          HIFIHR_WAVE_SYNC();
#endif
          float* sl = slab_of(wg + 1);
#pragma unroll
          for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int jj = 0; jj < TJ; ++jj) {
              float4* q = reinterpret_cast<float4*>(sl + ((i * TJ + jj) * 64 + lane) * 4);
              const float4 v = *q;
              acc[i][jj][0] += v.x; acc[i][jj][1] += v.y; acc[i][jj][2] += v.z; acc[i][jj][3] += v.w;
              *q = make_float4(0.f, 0.f, 0.f, 0.f);          // the workspace goes back all zero (it is shared with conv.hip's schedule)
            }
#if !defined(HIFIHR_HOSTSIM)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_store(fl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // clean for the next launch
#else
          HIFIHR_WAVE_SYNC();
          if (lane == 0) __atomic_store_n(fl, 0u, __ATOMIC_SEQ_CST);
#endif
        }
        const int p = tile / tiles_pb, rem = tile - p * tiles_pb;
        const int tn = rem / a.tiles_m, tm = rem - tn * a.tiles_m;
        float* __restrict__ C = a.C + (size_t)p * a.sc;
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) {
          const int m = tm * BM + wm * WM + 16 * jj + r;
          if (m < a.M) {
            float* row = C + (size_t)m * a.ldc + tn * BN + wn * WN + 4 * g;
#pragma unroll
            for (int i = 0; i < TI; ++i)
              *reinterpret_cast<float4*>(row + 16 * i) = make_float4(acc[i][jj][0], acc[i][jj][1], acc[i][jj][2], acc[i][jj][3]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int jj = 0; jj < TJ; ++jj) acc[i][jj] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    if (tile_end) { c = 0; ++tile; seg_c0 = 0; } else { ++c; }
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Persistent row-share NT form: C[p][m][n] = sum_k A[p][m][k] B[p][n][k], N a multiple of 128.
// What conv_halo_kernel (csrc/conv_halo.hip) showed for the layer-1 convolution carries over to the Winograd products: one workgroup per
// CU that walks an EQUAL share of the work as one continuous chunk stream -- its loader waves run three chunks ahead ACROSS tile
// boundaries, so no tile pays a pipeline fill or drain -- beats both the per-tile grid (3.06 rounds of 784 tiles, a fill and a drain
// per 4-16 chunk tile) and the stream-K split above (tiles shared through slabs and flags).  The share is a range of the flattened
// (problem p, 128-column tile, row) space, cut into tiles of up to 128 rows (the last one of a share may be shorter: the MFMA waves
// split the COLUMNS of a tile, 32 each, and take all of its 16-row blocks, so a short tile simply has fewer blocks); whole tiles
// only: nothing is exchanged between workgroups and nothing needs a workspace.  LDS: 4 stages x (A 128x32 + B 128x32) = 128 KB.
// ------------------------------------------------------------------------------------------------
static int gemm_cus();
struct RowsTile { int p, nt, m0, rows; };
// q = n / d, r = n - q d for 0 <= n, 0 < d < 2^31: one 32-bit unsigned division when n fits (every shape of the step does; ~30 instructions),
// the 64-bit sequence (several hundred cycles of dependent VALU work) otherwise
__device__ __forceinline__ void divmod_pos(long n, int d, long& q, int& r) {
  if (n < (1L << 32)) {
    const unsigned nn = (unsigned)n, qq = nn / (unsigned)d;
    q = (long)qq; r = (int)(nn - qq * (unsigned)d);
  } else {
    q = n / d; r = (int)(n - q * d);
  }
}
// The tiles of a share [cur, end) of the flattened (problem p, 128-column tile nt, row) space, in order.  The position is kept as (p, nt, m0) and
// moved on by additions: the two 64-bit divisions per tile the first form paid (cur / M, col / tiles_n -- in the counting loop, in the
// loader's bind and between the MFMA waves' tiles: ~1 000 cycles each, 3 000 before a workgroup's first load was issued) are now one 32-bit
// pair per wave at entry (round 6; tools/gemm_stamp4.py: entry -> first barrier 6 900 cycles, 1 900 between two tiles at the 128-channel shape).
struct RowsWalk {
  int p, nt, m0;
  long cur, end;
  __device__ __forceinline__ RowsWalk(const BgemmArgs& a, long lo, long hi) : cur(lo), end(hi) {
    long col; int pp;
    divmod_pos(lo, a.M, col, m0);
    long pl;
    divmod_pos(col, a.tiles_n, pl, nt);
    pp = (int)pl; p = pp;
  }
  __device__ __forceinline__ bool done() const { return cur >= end; }
  __device__ __forceinline__ RowsTile next(const BgemmArgs& a) {
    const long lim = min((long)a.M - m0, end - cur);
    const int rows = (int)min(128L, lim);
    const RowsTile t{p, nt, m0, rows};
    cur += rows; m0 += rows;
    if (m0 == a.M) { m0 = 0; if (++nt == a.tiles_n) { nt = 0; ++p; } }
    return t;
  }
};

// RAGGED (round 3): N need not be a multiple of 128 nor K of 32 (both of 4) -- EfficientNet's 1x1 convolutions (24, 40, 48, 96, 136, 144,
// 232, 288, 816, 1392 channels).  The loader reads the 16-byte operand segments that fall past row N of B or past column K from a page of
// zeros, the epilogue stores and counts only columns < N; the MFMA waves are the same.  A template flag: the square Winograd products keep the
// plain loader.
// MODE 2 (round 4): the A operand is GATHERED -- a forward convolution as this GEMM (the strided 3x3 and the stride-2 1x1 layers of the
// ResNet trunks sat on conv_igemm_kernel at 0.2-0.45 of the peak: 64 x 64 tiles, the gather through registers into LDS by the MFMA waves
// themselves).  Row m of A is the patch of output pixel m = (n, oh, ow) in (r, s, c) order; a chunk of 32 k-values is 32 channels of ONE
// tap, so a loader lane's 16-byte piece is a contiguous 16 bytes of the image -- the per-lane source address LDS-DMA takes anyway -- or
// the zero page when the tap falls outside the image.  The loader keeps the pixel's base pointer and top-left coordinate per piece (bound
// once per tile: two integer divisions per lane and piece) and walks (r, s, channel block) incrementally; the MFMA waves, the chunk stream
// across tile boundaries, the statistics epilogue are the plain kernel's.
// (body + thin kernel: bgemm_nt_tn_pair_kernel below runs this body and the TN one side by side in ONE launch)
template <int MODE>
__device__ __forceinline__ void nt_rows_body(const BgemmArgs& a, long per, float* __restrict__ lds, int bid, int nblk) {
  constexpr bool RAGGED = MODE == 1;
  constexpr bool CONVG = MODE == 2;
#if defined(HIFIHR_GEMM_STAMP)       // [0] cycles in the chunk loops, [1] 100 MHz ticks of them, [2] chunks, [3] at barriers, [4] workgroups,
  const unsigned long long st_entry = __builtin_amdgcn_s_memtime();      // [5] entry -> exit, [6] epilogues, [7] entry -> barrier -1
  unsigned long long st_loop = 0, st_real = 0, st_bar = 0, st_epi = 0, st_first = 0;
#endif
  constexpr int STAGE = 256 * 32;                            // floats per stage: A rows 0..127, B rows 128..255
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = xcd_remap(bid, nblk);
  const long total = (long)a.batch * a.tiles_n * a.M;
  const long s_lo = (long)wg * per, s_hi = min(s_lo + per, total);
  if (s_lo >= s_hi) return;                                  // (uniform)
  const int nch = RAGGED ? (a.K + 31) / 32 : a.K / 32;
  int ntiles = 0;
  const RowsWalk walk0(a, s_lo, s_hi);
  for (RowsWalk c = walk0; !c.done(); c.next(a)) ++ntiles;
  const int nchunks = ntiles * nch;
  const bool head1 = MODE == 0 && nch >= 2;                  // (uniform) chunk 1 belongs to the first tile and is issued by the MFMA waves

  if (wave >= 4) {
    // ---------------- loader: piece q = l + 4 i (i < 8) of a chunk: q < 16 rows 8 q .. + 7 of A, else rows 8 (q - 16) .. of B ----------------
    const int l = wave - 4;
    HIFIHR_SET_LOADER_PRIO();
    RowsWalk lw = walk0;
    RowsTile t = lw.next(a);
    const float* src[8];
    int kseg[8];                                             // RAGGED: first k of this lane's segment within a chunk, or 1 << 30 for a B row >= N
    int ih0[4] = {0, 0, 0, 0}, iw0[4] = {0, 0, 0, 0};        // CONVG: image coordinate of tap (0, 0) of the pixel of A piece i (i < 4: l + 4 i < 16)
    auto bind = [&](const RowsTile& tt) {                    // per-lane source row of every piece for this tile (k offset added per chunk)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = l + 4 * i;
        const bool isA = q < 16;
        const int row = 8 * (q & 15) + (lane >> 3);
        const int seg = (lane & 7) ^ ((row >> 1) & 7);
        if (isA && CONVG) {
          const int m = tt.m0 + min(row, tt.rows - 1);       // output pixel (n, oh, ow); rows past the tile: discarded
          const int n = m / (a.cOH * a.cOW), rem = m - n * (a.cOH * a.cOW), oh = rem / a.cOW, ow = rem - oh * a.cOW;
          ih0[i & 3] = oh * a.cStride - a.cPad; iw0[i & 3] = ow * a.cStride - a.cPad;
          // (formed with signed arithmetic: a pointer in front of the image is never dereferenced, the zero page is read instead)
          src[i] = a.A + (((long)n * a.cIH + ih0[i & 3]) * a.cIW + iw0[i & 3]) * a.cC + seg * 4;
        } else if (isA) src[i] = a.A + (size_t)tt.p * a.sa + (size_t)(tt.m0 + min(row, tt.rows - 1)) * a.lda + seg * 4;   // rows past the tile: discarded
        else src[i] = a.B + (size_t)tt.p * a.sb + (size_t)(tt.nt * 128 + row) * a.ldb + seg * 4;
        if (RAGGED) kseg[i] = (!isA && tt.nt * 128 + row >= a.N) ? (1 << 30) : seg * 4;
      }
    };
    bind(t);
    int li = 0, lc = 0;                                      // tile / chunk-in-tile the NEXT issue belongs to
    int tr = 0, ts = 0, tcb = 0;                             // CONVG: tap (r, s) and 32-channel block of chunk lc
    const int cpb = CONVG ? a.cC / 32 : 1;
    auto issue_next = [&](int gc, bool issue = true) {     // issue == false: chunk gc is somebody else's (chunk 1, below) -- only move on
      float* base = lds + (gc & 3) * STAGE;
      const long aoff = CONVG ? ((long)tr * a.cIW + ts) * a.cC + tcb * 32 : 0;
      if (issue) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float* s = src[i] + lc * 32;
          if (RAGGED) s = (kseg[i] + lc * 32 < a.K) ? s : a.zeros;       // (a row >= N: 1 << 30; the last chunk's segments past column K)
          if (CONVG && i < 4) {
            const int ih = ih0[i] + tr, iw = iw0[i] + ts;
            s = ((unsigned)ih < (unsigned)a.cIH && (unsigned)iw < (unsigned)a.cIW) ? src[i] + aoff : a.zeros;
          }
          HIFIHR_GLDS16(s, base + 256 * (l + 4 * i), lane);
        }
      }
      if (CONVG) {
        if (++tcb == cpb) { tcb = 0; if (++ts == a.cS) { ts = 0; ++tr; } }
      }
      if (++lc == nch) {
        lc = 0; tr = 0; ts = 0; tcb = 0;
        if (++li < ntiles) { t = lw.next(a); bind(t); }
      }
    };
    // The first chunks.  A wave gets an LDS-DMA instruction out every ~200 cycles, so three chunks in a row from the loader waves put 3 200
    // cycles + the memory latency in front of the first MFMA (tools/gemm_stamp4.py: entry -> first barrier 6 900 cycles, 14 % of a
    // workgroup's life at the 128-channel shape).  `head1`: chunk 1 -- the second chunk of the share's first tile -- is issued by the MFMA
    // waves, which have nothing else to do yet (wave w = loader w's pieces, below), beside chunk 0 here.
    issue_next(0);
    if (nchunks > 1) issue_next(1, !head1);
    if (nchunks > 2) issue_next(2);
    if (nchunks > 2) HIFIHR_WAIT_VM(8); else HIFIHR_WAIT_VM(0);      // chunks 0 and 1 landed (head1: chunk 0; chunk 1 is waited for by its issuers)
    HIFIHR_RAW_BARRIER();                                    // barrier -1
    for (int gc = 0; gc < nchunks; ++gc) {
      if (gc + 3 < nchunks) { issue_next(gc + 3); HIFIHR_WAIT_VM(8); }   // chunk gc + 2 landed; only chunk gc + 3's pieces in flight
      else HIFIHR_WAIT_VM(0);
      HIFIHR_RAW_BARRIER();                                  // barrier gc
    }
    return;
  }

  // ---------------- MFMA waves: wave w = columns 32 w .. 32 w + 31 of the tile, every 16-row block ----------------
  const int r = lane & 15, g = lane >> 4;
  const char* const lds_b = reinterpret_cast<const char*>(lds);
  int aoff[2], boff[2];                                      // lane part of the fragment addresses (h = 0, 1); + 2048 per row block
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int sw = ((g + 4 * h) ^ ((r >> 1) & 7)) * 16;
    aoff[h] = r * 128 + sw;
    boff[h] = (128 + 32 * wave + r) * 128 + sw;
  }
  if (head1) {
    // chunk 1 of the first tile: MFMA wave w issues the pieces of loader wave w (the loader's bind() for a plain tile, k offset 32)
    RowsWalk hw = walk0;
    const RowsTile tt = hw.next(a);
    float* base = lds + STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = wave + 4 * i;
      const int row = 8 * (q & 15) + (lane >> 3);
      const int seg = (lane & 7) ^ ((row >> 1) & 7);
      const float* s = q < 16 ? a.A + (size_t)tt.p * a.sa + (size_t)(tt.m0 + min(row, tt.rows - 1)) * a.lda + seg * 4
                              : a.B + (size_t)tt.p * a.sb + (size_t)(tt.nt * 128 + row) * a.ldb + seg * 4;
      HIFIHR_GLDS16(s + 32, base + 256 * q, lane);
    }
    HIFIHR_WAIT_VM(0);
  }
  HIFIHR_RAW_BARRIER();                                      // barrier -1
#if defined(HIFIHR_GEMM_STAMP)
  st_first = __builtin_amdgcn_s_memtime() - st_entry;
#endif
  RowsWalk mw = walk0;
  int gc = 0;
  // batch-norm statistics of the output (a.stats != null): per-column sums of this wave's 32 columns, kept in registers over the tiles of
  // one column tile (a share walks the rows of a column tile before it moves on), then folded over the 16 row lanes and added to the
  // slot buffer -- the epilogue conv_igemm_kernel / conv_halo_kernel have, so a 1x1 convolution on this kernel needs no statistics pass
  // (shifted sums: d = y - sk with sk the lane's first row of the column since the last flush; unshifted in fp64 at the flush --
  //  hifihr_internal.h "FORWARD statistics")
  float ssum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ssq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float sk[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  int stat_nt = -1, sn = 0;
  auto flush_stats = [&]() {
    if (stat_nt < 0) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double sa, sb;
        stat_unshift(sn, sk[i][e], ssum[i][e], ssq[i][e], sa, sb);
        for (int o = 1; o < 16; o <<= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
        if (r == 0 && (!RAGGED || stat_nt * 128 + 32 * wave + 16 * i + 4 * g + e < a.N)) {      // (columns >= N hold zeros: no slot)
          double* sp = reinterpret_cast<double*>(a.stats) + (size_t)(wg & (stat_slots_used(a.N) - 1)) * 2 * a.N + (size_t)stat_nt * 128 + 32 * wave +
                       16 * i + 4 * g + e;
          stat_atomic_add(sp, sa); stat_atomic_add(sp + a.N, sb);
        }
        ssum[i][e] = 0.f; ssq[i][e] = 0.f;
      }
    sn = 0;
  };
  auto run_tile = [&](auto nbc, const RowsTile& t) {
    constexpr int NB = decltype(nbc)::value;
#if defined(HIFIHR_PROBE_SPLIT_BF16)
    if constexpr (MODE == 3) {
      // PROBE (tools/split_bf16_probe.py; not in libhifihr.so): the operands are SPLIT bf16 -- the 128 bytes of a row's 32-deep chunk hold the
      // 32 leading pieces hi = bf16(x) (segments 0 .. 3) and the 32 remainders lo = bf16(x - hi) (segments 4 .. 7): the same bytes and the
      // same LDS image as f32, so the loader waves above are untouched, and the 16 bytes lane (r, g) reads at segment g (+ 4 for lo) ARE
      // its operand of v_mfma_f32_16x16x32_bf16 (k = 8 g .. 8 g + 7).  Per chunk: hi.hi, hi.lo, lo.hi = 3 x 2 NB MFMAs of 16 cycles (768 for
      // NB = 8) where the f32 form spends 4 096.  hi of the NEXT chunk is prefetched under the cross terms (two register sets for hi).
      typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
      bf8 ahi[2][NB], alo[NB], bhi[2][2], blo[2];
      auto rd = [&](int gcc, int h, bf8 (&a)[NB], bf8 (&b)[2]) {
        const char* st = lds_b + (gcc & 3) * (STAGE * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const bf8*>(st + boff[h] + i * 2048);
#pragma unroll
        for (int j = 0; j < NB; ++j) a[j] = *reinterpret_cast<const bf8*>(st + aoff[h] + j * 2048);
      };
      floatx4 acc[2][NB];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
      auto mm = [&](bf8 (&b)[2], bf8 (&a)[NB]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[i], a[j], acc[i][j], 0, 0, 0);
      };
      rd(gc, 0, ahi[0], bhi[0]);
      for (int c = 0; c < nch; c += 2) {                       // (nch even: the probe's K % 64 == 0)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          rd(gc, 1, alo, blo);
          mm(bhi[u], ahi[u]);                                  // hi . hi
          rd(gc + 1, 0, ahi[u ^ 1], bhi[u ^ 1]);               // (landed: barrier gc - 1; past the tile's last chunk: re-read at the next tile's start)
          mm(bhi[u], alo);                                     // hi . lo  (n-side hi, m-side lo)
          mm(blo, ahi[u]);                                     // lo . hi
          HIFIHR_RAW_BARRIER();                                // barrier gc
          ++gc;
        }
      }
      float* C = a.C + (size_t)t.p * a.sc + (size_t)t.nt * 128 + 32 * wave + 4 * g;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int m = 16 * j + r;
        if (m < t.rows) {
          float* row = C + (size_t)(t.m0 + m) * a.ldc;
          *reinterpret_cast<float4*>(row) = make_float4(acc[0][j][0], acc[0][j][1], acc[0][j][2], acc[0][j][3]);
          *reinterpret_cast<float4*>(row + 16) = make_float4(acc[1][j][0], acc[1][j][1], acc[1][j][2], acc[1][j][3]);
        }
      }
      return;
    }
#endif
    float fm[2][NB][4], fn[2][2][4];
    auto read_half = [&](int gcc, int h, int slot) {
      const char* st = lds_b + (gcc & 3) * (STAGE * 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(st + boff[h] + i * 2048);
        fn[slot][i][0] = v.x; fn[slot][i][1] = v.y; fn[slot][i][2] = v.z; fn[slot][i][3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(st + aoff[h] + j * 2048);
        fm[slot][j][0] = v.x; fm[slot][j][1] = v.y; fm[slot][j][2] = v.z; fm[slot][j][3] = v.w;
      }
    };
    floatx4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    auto mfma_half = [&](int slot) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fn[slot][i][k], fm[slot][j][k], acc[i][j], 0, 0, 0);
    };
    auto touch = [&]() {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        HIFIHR_TOUCH(fn[0][0][k]); HIFIHR_TOUCH(fn[0][1][k]);
#pragma unroll
        for (int j = 0; j < NB; ++j) HIFIHR_TOUCH(fm[0][j][k]);
      }
    };
    auto interleave = [&]() {                                // NB + 2 LDS reads of the next half between the 8 NB MFMAs of this one
#pragma unroll
      for (int i = 0; i < NB + 2; ++i) {
        HIFIHR_SCHED_GROUP(0x008, NB >= 4 ? 4 : 2);
        HIFIHR_SCHED_GROUP(0x100, 1);
        HIFIHR_SCHED_GROUP(0x002, 1);
      }
      HIFIHR_SCHED_GROUP(0x008, 8 * NB - (NB >= 4 ? 4 : 2) * (NB + 2) > 0 ? 8 * NB - (NB >= 4 ? 4 : 2) * (NB + 2) : 0);
    };
    read_half(gc, 0, 0);
    touch();
#if defined(HIFIHR_GEMM_STAMP)
    const unsigned long long l0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // register e of lane (r, g) of block (i, j) = C[m0 + 16 j + r][128 nt + 32 wave + 16 i + 4 g + e]
    float* C = a.C + (size_t)t.p * a.sc + (size_t)t.nt * 128 + 32 * wave + 4 * g;
    const int col0 = t.nt * 128 + 32 * wave + 4 * g;         // (RAGGED: the lane's column blocks col0 .. + 3 and col0 + 16 .. + 19 against N)
    auto store_block = [&](int j) {
      const int m = 16 * j + r;
      if (m < t.rows) {
        float* row = C + (size_t)(t.m0 + m) * a.ldc;
        if (!RAGGED || col0 < a.N) *reinterpret_cast<float4*>(row) = make_float4(acc[0][j][0], acc[0][j][1], acc[0][j][2], acc[0][j][3]);
        if (!RAGGED || col0 + 16 < a.N) *reinterpret_cast<float4*>(row + 16) = make_float4(acc[1][j][0], acc[1][j][1], acc[1][j][2], acc[1][j][3]);
      }
    };
    for (int c = 0; c < nch - 1; ++c, ++gc) {
      read_half(gc, 1, 1);
      mfma_half(0);
      interleave();
      HIFIHR_PIN();
      read_half(gc + 1, 0, 0);                               // (landed: barrier gc - 1)
      mfma_half(1);
      interleave();
      HIFIHR_PIN();
      touch();
#if defined(HIFIHR_GEMM_STAMP)
      HIFIHR_TOUCH(acc[0][0][0]);
      const unsigned long long b0 = __builtin_amdgcn_s_memtime();
#endif
      HIFIHR_RAW_BARRIER();                                  // barrier gc
#if defined(HIFIHR_GEMM_STAMP)
      st_bar += __builtin_amdgcn_s_memtime() - b0;
#endif
    }
    {
      // The tile's LAST chunk runs row block by row block, both halves of the chunk per block (every accumulator still sums its k-steps in
      // the plain loop's order: the same bits -- tests/test_hostsim_gemm.py::test_row_share_kernels_keep_the_summation_order), so that
      // block j is final sixteen MFMAs after block j - 1 and the stores of block j - 1 -- two per lane -- go out between the MFMAs of
      // the blocks behind it: the 16 stores of a tile back to back took 3 500 cycles (a wave that waits to issue a store issues no MFMA;
      // tools/gemm_stamp4.py: 15 % of a workgroup's life at the 128-channel shape), spread over the chunk's 4 096 cycles of matrix work
      // most of them find the address path free.  (No prefetch of the next tile's first half here: slot 0 is in use until the last
      // block; every tile starts with that read anyway.)
      read_half(gc, 1, 1);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fn[h][i][k], fm[h][j][k], acc[i][j], 0, 0, 0);
        HIFIHR_PIN();
        if (j > 0) {
          store_block(j - 1);
          HIFIHR_PIN();
        }
      }
      store_block(NB - 1);
#if defined(HIFIHR_GEMM_STAMP)
      HIFIHR_TOUCH(acc[0][0][0]);
      const unsigned long long b0 = __builtin_amdgcn_s_memtime();
#endif
      HIFIHR_RAW_BARRIER();                                  // barrier gc
#if defined(HIFIHR_GEMM_STAMP)
      st_bar += __builtin_amdgcn_s_memtime() - b0;
#endif
      ++gc;
    }
#if defined(HIFIHR_GEMM_STAMP)
    const unsigned long long l1 = __builtin_amdgcn_s_memtime();
    st_loop += l1 - l0; st_real += __builtin_amdgcn_s_memrealtime() - r0;
#endif
    if (a.stats != nullptr) {                                // (uniform) the statistics of the tile, from the accumulators the stores left in place
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (16 * j + r < t.rows) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (sn == 0) sk[i][e] = acc[i][j][e];
              const float d = acc[i][j][e] - sk[i][e];
              ssum[i][e] += d; ssq[i][e] += d * d;
            }
          ++sn;
        }
      }
    }
#if defined(HIFIHR_GEMM_STAMP)
    st_epi += __builtin_amdgcn_s_memtime() - l1;
#endif
  };
  for (int ti = 0; ti < ntiles; ++ti) {
    const RowsTile t = mw.next(a);
    if (a.stats != nullptr && t.nt != stat_nt) { flush_stats(); stat_nt = t.nt; }       // (uniform; batch == 1: nt identifies the columns)
    switch ((t.rows + 15) >> 4) {
      case 8: run_tile(std::integral_constant<int, 8>{}, t); break;
      case 7: run_tile(std::integral_constant<int, 7>{}, t); break;
      case 6: run_tile(std::integral_constant<int, 6>{}, t); break;
      case 5: run_tile(std::integral_constant<int, 5>{}, t); break;
      case 4: run_tile(std::integral_constant<int, 4>{}, t); break;
      case 3: run_tile(std::integral_constant<int, 3>{}, t); break;
      case 2: run_tile(std::integral_constant<int, 2>{}, t); break;
      default: run_tile(std::integral_constant<int, 1>{}, t); break;
    }
  }
  if (a.stats != nullptr) flush_stats();
#if defined(HIFIHR_GEMM_STAMP)
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicAdd(&g_gemm_stamp[0], st_loop); atomicAdd(&g_gemm_stamp[1], st_real); atomicAdd(&g_gemm_stamp[2], (unsigned long long)nchunks);
    atomicAdd(&g_gemm_stamp[3], st_bar); atomicAdd(&g_gemm_stamp[4], 1ull); atomicAdd(&g_gemm_stamp[5], __builtin_amdgcn_s_memtime() - st_entry);
    atomicAdd(&g_gemm_stamp[6], st_epi); atomicAdd(&g_gemm_stamp[7], st_first);
  }
#endif
}

constexpr int kRowsStage = 256 * 32;                         // floats per LDS stage of the row-share kernels (4 stages = 128 KB)
template <int MODE>
__global__ __launch_bounds__(512) void bgemm_nt_rows_kernel(BgemmArgs a, long per) {
  __shared__ __attribute__((aligned(1024))) float lds[4 * kRowsStage];
  nt_rows_body<MODE>(a, per, lds, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// Persistent row-share TN form: C[p][m][n] = sum_t A[p][t][m] B[p][t][n]  (backward-weight of the Winograd layers: A = Y'[P][T][K],
// B = V[P][T][C]), N a multiple of 128, M a multiple of 16, T a multiple of 32.  The same schedule as bgemm_nt_rows_kernel -- one
// workgroup per CU walks an equal share of the flattened (problem, 128-column tile, 16-row block) space as ONE continuous chunk stream,
// four loader waves three chunks ahead across tile boundaries, four MFMA waves splitting a tile's columns -- with the operands in their
// t-major layout: a chunk is 32 t rows x 128 floats of A (the tile's rows are A's columns) and of B.  Every tile covers the whole
// reduction, so the result is complete (one slab: nothing for the consumer to sum) and bit-reproducible.  The plain 64x64 kernel ran
// these 36-problem products at 0.52-0.57 of the peak (a pipeline fill and drain per 16-chunk tile, every wave issuing its own LDS-DMA).
// LDS image of a chunk: A [32][128] at pitch 512 B; B [32][128] with the 16-byte segments of row t XOR-ed by 4 (t & 3) (on the source
// address): lane (r, g) of a k-step reads B[t = 4 k + g][n = .. + r] as one ds_read_b32, and the four t rows of a wave would otherwise
// hit the same 16 banks.  A fragments are ds_read_b128: lane r owns rows 4 r .. 4 r + 3 (and 64 + 4 r .. for 128-row tiles) of the
// tile -- a row permutation the epilogue undoes.  Tiles are 128, 64, 32 or 16 rows (the tail of a share is cut into powers of two).
// ------------------------------------------------------------------------------------------------
struct TnTile { int p, nt, m0, nb; };

// The tiles of one workgroup, in order: `nr` ranges [lo0 + i step, + len) of the flattened (problem, 128-column tile, 16-row block) space -- its
// slices of the rounds of the XCD-coherent schedule (BgemmArgs::co_r) -- then one range [tail_lo, tail_hi): the contiguous share (all of
// the work when nr == 0).  Every range is cut into tiles of 8, 4, 2 or 1 row blocks (TnWalk::next).
// (position (p, nt, mb0) of `cur` kept by additions inside a range, divisions -- 32-bit, divmod_pos -- only where a new range starts: see RowsWalk)
struct TnWalk {
  int nr, ri;
  long lo0, step, len, tail_lo, tail_hi, cur, end;
  int p, nt, mb0;
  __device__ __forceinline__ bool next(const BgemmArgs& a, TnTile& t) {
    const int MB = a.M / 16;
    if (cur >= end) {
      do {
        if (ri < nr) { cur = lo0 + (long)ri * step; end = cur + len; }
        else if (ri == nr && tail_lo < tail_hi) { cur = tail_lo; end = tail_hi; }
        else return false;
        ++ri;
      } while (cur >= end);
      long col, pl;
      divmod_pos(cur, MB, col, mb0);
      divmod_pos(col, a.tiles_n, pl, nt);
      p = (int)pl;
    }
    const long lim = min((long)MB - mb0, end - cur);
    const int nb = lim >= 8 ? 8 : lim >= 4 ? 4 : lim >= 2 ? 2 : 1;
    t = TnTile{p, nt, mb0 * 16, nb};
    cur += nb; mb0 += nb;
    if (mb0 == MB) { mb0 = 0; if (++nt == a.tiles_n) { nt = 0; ++p; } }
    return true;
  }
};
__device__ __forceinline__ TnWalk tn_walk(const BgemmArgs& a, long per, int wg, int nblk) {
  TnWalk w;
  w.ri = 0; w.cur = 0; w.end = 0; w.p = 0; w.nt = 0; w.mb0 = 0;
  const long total = (long)a.batch * a.tiles_n * (a.M / 16);
  if (a.co_rounds > 0) {
    const int W = nblk >> 3, x = wg / W, j = wg - x * W;      // (nblk % 8 == 0: xcd_remap hands an XCD W consecutive workgroup ids)
    const long bpp = (long)a.tiles_n * (a.M / 16), bw = a.co_r * bpp / W;
    w.nr = a.co_rounds; w.step = 8L * a.co_r * bpp; w.lo0 = (long)x * a.co_r * bpp + (long)j * bw; w.len = bw;
    const long t0 = (long)a.co_rounds * w.step, rest = total - t0, pt = (rest + nblk - 1) / nblk;
    w.tail_lo = min(t0 + (long)wg * pt, total); w.tail_hi = min(w.tail_lo + pt, total);
  } else {
    w.nr = 0; w.lo0 = 0; w.step = 0; w.len = 0;
    w.tail_lo = min((long)wg * per, total); w.tail_hi = min(w.tail_lo + per, total);
  }
  return w;
}

__device__ __forceinline__ void tn_rows_body(const BgemmArgs& a, long per, float* __restrict__ lds, int bid, int nblk) {
  constexpr int STAGE = 256 * 32;                            // floats per stage: A rows t 0..31 (x 128), then B rows t 0..31 (x 128)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = xcd_remap(bid, nblk);
  const TnWalk walk = tn_walk(a, per, wg, nblk);
  const int nch = a.K / 32;
  int ntiles = 0;
  {
    TnWalk c = walk;
    TnTile tt;
    while (c.next(a, tt)) ++ntiles;
  }
  if (ntiles == 0) return;                                   // (uniform)
  const int nchunks = ntiles * nch;
  const bool head1 = nch >= 2;                               // (uniform) chunk 1 = the first tile's second chunk: issued by the MFMA waves

  if (wave >= 4) {
    HIFIHR_SET_LOADER_PRIO();
    // ---------------- loader: piece q = l + 4 i (i < 8) of a chunk: q < 16 rows t = 2 q, 2 q + 1 of A, else rows 2 (q - 16) .. of B;
    //                  lane -> (row lane >> 5, physical 16-byte segment lane & 31) ----------------
    const int l = wave - 4;
    TnWalk lw = walk;
    TnTile t;
    lw.next(a, t);
    const float* src[8];
    size_t step[8];
    auto bind = [&](const TnTile& tt) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = l + 4 * i;
        const bool isA = q < 16;
        const int trow = 2 * (q & 15) + (lane >> 5);
        const int ps = lane & 31;
        if (isA) {
          int col = tt.m0 + 4 * ps;
          col = col <= a.M - 4 ? col : a.M - 4;              // columns past the matrix: discarded rows of the tile
          src[i] = a.A + (size_t)tt.p * a.sa + (size_t)trow * a.lda + col;
          step[i] = (size_t)32 * a.lda;
        } else {
          const int seg = ps ^ (4 * (trow & 3));
          src[i] = a.B + (size_t)tt.p * a.sb + (size_t)trow * a.ldb + tt.nt * 128 + 4 * seg;
          step[i] = (size_t)32 * a.ldb;
        }
      }
    };
    bind(t);
    int li = 0, lc = 0;
    auto issue_next = [&](int gc, bool issue = true) {       // issue == false: chunk 1, issued by the MFMA waves (see nt_rows_body)
      float* base = lds + (gc & 3) * STAGE;
      if (issue) {
#pragma unroll
        for (int i = 0; i < 8; ++i) HIFIHR_GLDS16(src[i] + lc * step[i], base + 256 * (l + 4 * i), lane);
      }
      if (++lc == nch) {
        lc = 0;
        if (++li < ntiles) { lw.next(a, t); bind(t); }
      }
    };
    issue_next(0);
    if (nchunks > 1) issue_next(1, !head1);
    if (nchunks > 2) issue_next(2);
    if (nchunks > 2) HIFIHR_WAIT_VM(8); else HIFIHR_WAIT_VM(0);
    HIFIHR_RAW_BARRIER();                                    // barrier -1
    for (int gc = 0; gc < nchunks; ++gc) {
      if (gc + 3 < nchunks) { issue_next(gc + 3); HIFIHR_WAIT_VM(8); }
      else HIFIHR_WAIT_VM(0);
      HIFIHR_RAW_BARRIER();                                  // barrier gc
    }
    return;
  }

  // ---------------- MFMA waves: wave w = columns 32 w .. 32 w + 31 of the tile, every row block ----------------
  const int r = lane & 15, g = lane >> 4;
  int boff[2];                                               // float offset inside a B row of this lane's two columns (swizzle of t & 3 = g applied)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int colb = 32 * wave + 16 * i + r;
    boff[i] = 4 * ((colb >> 2) ^ (4 * g)) + (colb & 3);
  }
  if (head1) {
    // chunk 1 of the first tile beside the loaders' chunk 0 (nt_rows_body): MFMA wave w issues loader wave w's pieces
    TnWalk hw = walk;
    TnTile tt;
    hw.next(a, tt);
    float* base = lds + STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = wave + 4 * i;
      const int trow = 2 * (q & 15) + (lane >> 5);
      const int ps = lane & 31;
      const float* sp;
      if (q < 16) {
        int col = tt.m0 + 4 * ps;
        col = col <= a.M - 4 ? col : a.M - 4;
        sp = a.A + (size_t)tt.p * a.sa + (size_t)(trow + 32) * a.lda + col;
      } else {
        const int seg = ps ^ (4 * (trow & 3));
        sp = a.B + (size_t)tt.p * a.sb + (size_t)(trow + 32) * a.ldb + tt.nt * 128 + 4 * seg;
      }
      HIFIHR_GLDS16(sp, base + 256 * q, lane);
    }
    HIFIHR_WAIT_VM(0);
  }
  HIFIHR_RAW_BARRIER();                                      // barrier -1
  TnWalk mw = walk;
  int gc = 0;
  auto run_tile = [&](auto nbc, const TnTile& t) {
    constexpr int NB = decltype(nbc)::value;
    float fm[2][4][NB], fn[2][4][2];
    auto read_half = [&](int gcc, int h, int slot) {        // k-steps 4 h .. 4 h + 3 of chunk gcc: t = 16 h + 4 k + g
      const float* st = lds + (gcc & 3) * STAGE;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int trow = 16 * h + 4 * k + g;
        const float* ar = st + trow * 128;
        const float* br = st + 32 * 128 + trow * 128;
        fn[slot][k][0] = br[boff[0]]; fn[slot][k][1] = br[boff[1]];
        if constexpr (NB == 8) {
          const float4 v0 = *reinterpret_cast<const float4*>(ar + 4 * r), v1 = *reinterpret_cast<const float4*>(ar + 64 + 4 * r);
          fm[slot][k][0] = v0.x; fm[slot][k][1] = v0.y; fm[slot][k][2] = v0.z; fm[slot][k][3] = v0.w;
          fm[slot][k][4] = v1.x; fm[slot][k][5] = v1.y; fm[slot][k][6] = v1.z; fm[slot][k][7] = v1.w;
        } else if constexpr (NB == 4) {
          const float4 v0 = *reinterpret_cast<const float4*>(ar + 4 * r);
          fm[slot][k][0] = v0.x; fm[slot][k][1] = v0.y; fm[slot][k][2] = v0.z; fm[slot][k][3] = v0.w;
        } else if constexpr (NB == 2) {
          const float2 v0 = *reinterpret_cast<const float2*>(ar + 2 * r);
          fm[slot][k][0] = v0.x; fm[slot][k][1] = v0.y;
        } else {
          fm[slot][k][0] = ar[r];
        }
      }
    };
    floatx4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    auto mfma_half = [&](int slot) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fn[slot][k][i], fm[slot][k][j], acc[i][j], 0, 0, 0);
    };
    // rows t >= k_valid of the problem are zero in both operands (BgemmArgs::k_valid): k-steps (4 rows each) wholly behind them add nothing
    int kv_steps = nch * 8;                                   // k-steps of this tile's reduction that carry data
    if (a.k_valid > 0) {
      const int part = a.splits > 1 ? t.p % a.splits : 0;
      const int valid = min(max(a.k_valid - part * a.K, 0), a.K);
      kv_steps = (valid + 3) >> 2;
    }
    auto touch = [&]() {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        HIFIHR_TOUCH(fn[0][k][0]); HIFIHR_TOUCH(fn[0][k][1]);
#pragma unroll
        for (int j = 0; j < NB; ++j) HIFIHR_TOUCH(fm[0][k][j]);
      }
    };
    auto interleave = [&]() {                                // the LDS reads of the next half spread between the 8 NB MFMAs of this one
      constexpr int RH = 4 * (2 + (NB >= 4 ? NB / 4 : 1));   // reads per half
      constexpr int MG = (8 * NB) / RH > 0 ? (8 * NB) / RH : 1;
#pragma unroll
      for (int i = 0; i < RH; ++i) {
        HIFIHR_SCHED_GROUP(0x008, MG);
        HIFIHR_SCHED_GROUP(0x100, 1);
        HIFIHR_SCHED_GROUP(0x002, 1);
      }
      HIFIHR_SCHED_GROUP(0x008, 8 * NB - MG * RH > 0 ? 8 * NB - MG * RH : 0);
    };
    // register e of lane (r, g) of block (i, j) = C[m0 + row(j, r)][128 nt + 32 wave + 16 i + 4 g + e]
    // (T-split, a.splits > 1: "problem" t.p is part t.p % splits of problem t.p / splits; its result is that problem's tile of slab `part`)
    const int preal = a.splits > 1 ? t.p / a.splits : t.p, part = t.p - preal * (a.splits > 1 ? a.splits : 1);
    float* C = a.C + (size_t)part * a.sc_split + (size_t)preal * a.sc + (size_t)t.nt * 128 + 32 * wave + 4 * g;
    auto store_block = [&](int j) {
      const int m = NB == 8 ? 64 * (j >> 2) + 4 * r + (j & 3) : NB == 4 ? 4 * r + j : NB == 2 ? 2 * r + j : r;
      float* row = C + (size_t)(t.m0 + m) * a.ldc;
      *reinterpret_cast<float4*>(row) = make_float4(acc[0][j][0], acc[0][j][1], acc[0][j][2], acc[0][j][3]);
      *reinterpret_cast<float4*>(row + 16) = make_float4(acc[1][j][0], acc[1][j][1], acc[1][j][2], acc[1][j][3]);
    };
    read_half(gc, 0, 0);
    touch();
    const int nfull = min(nch, kv_steps >> 3);               // chunks whose eight k-steps all carry data
    const int rem0 = nfull < nch ? kv_steps - 8 * nfull : 0; // live k-steps of the chunk behind them (450 real rows = 14 chunks + 2 rows: 1)
    // `fused`: the tile's last full chunk runs row block by row block (nt_rows_body: the stores of block j - 1 between the MFMAs of the blocks
    // behind it), together with the ONE live k-step of the chunk behind it, taken straight from LDS (landed: barrier gc - 1) -- every
    // accumulator still sums its k-steps in the plain loop's order (t ascending).  More than one live k-step behind the full chunks, or no full chunk: the plain form.
    const bool fused = nfull >= 1 && rem0 <= 1;              // (uniform)
    const int nplain = fused ? nfull - 1 : nfull;
    for (int c = 0; c < nplain; ++c, ++gc) {
      read_half(gc, 1, 1);
      mfma_half(0);
      interleave();
      HIFIHR_PIN();
      read_half(gc + 1, 0, 0);                               // (landed: barrier gc - 1)
      mfma_half(1);
      interleave();
      HIFIHR_PIN();
      touch();
      HIFIHR_RAW_BARRIER();                                  // barrier gc
    }
    if (fused) {
      read_half(gc, 1, 1);
      float tb0 = 0.f, tb1 = 0.f, tam[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) tam[j] = 0.f;
      if (rem0 == 1) {                                       // k-step 0 of chunk gc + 1 (rows t = g)
        const float* st = lds + ((gc + 1) & 3) * STAGE;
        const float* ar = st + g * 128;
        const float* br = st + 32 * 128 + g * 128;
        tb0 = br[boff[0]]; tb1 = br[boff[1]];
        if constexpr (NB == 8) {
          const float4 v0 = *reinterpret_cast<const float4*>(ar + 4 * r), v1 = *reinterpret_cast<const float4*>(ar + 64 + 4 * r);
          tam[0] = v0.x; tam[1] = v0.y; tam[2] = v0.z; tam[3] = v0.w; tam[4] = v1.x; tam[5] = v1.y; tam[6] = v1.z; tam[7] = v1.w;
        } else if constexpr (NB == 4) {
          const float4 v0 = *reinterpret_cast<const float4*>(ar + 4 * r);
          tam[0] = v0.x; tam[1] = v0.y; tam[2] = v0.z; tam[3] = v0.w;
        } else if constexpr (NB == 2) {
          const float2 v0 = *reinterpret_cast<const float2*>(ar + 2 * r);
          tam[0] = v0.x; tam[1] = v0.y;
        } else {
          tam[0] = ar[r];
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fn[h][k][i], fm[h][k][j], acc[i][j], 0, 0, 0);
        if (rem0 == 1) {                                     // (uniform)
          acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tb0, tam[j], acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(tb1, tam[j], acc[1][j], 0, 0, 0);
        }
        HIFIHR_PIN();
        if (j > 0) {
          store_block(j - 1);
          HIFIHR_PIN();
        }
      }
      store_block(NB - 1);
      HIFIHR_RAW_BARRIER();                                  // barrier gc
      ++gc;
      for (int c = nfull; c < nch; ++c, ++gc) HIFIHR_RAW_BARRIER();      // the chunks behind: their live k-step is done, the rest is zeros
      return;
    }
    // The tail: a chunk that runs into the zero rows (its first `rem` k-steps carry data: 450 real rows = 14 chunks + 2 rows) and chunks of
    // zeros only.  The loader waves stream them like any other chunk (the chunk stream and its barriers are theirs to keep); here a ROLLED
    // loop takes the live k-steps straight from LDS -- no second register set, no schedule: a few hundred cycles per tile -- and the rest
    // only meets the barriers.
    for (int c = nfull; c < nch; ++c, ++gc) {
      const int rem = c == nfull ? rem0 : 0;
      const float* st = lds + (gc & 3) * STAGE;
#pragma unroll 1
      for (int kk = 0; kk < rem; ++kk) {
        const int trow = 4 * kk + g;
        const float* ar = st + trow * 128;
        const float* br = st + 32 * 128 + trow * 128;
        const float b0 = br[boff[0]], b1 = br[boff[1]];
        float am[NB];
        if constexpr (NB == 8) {
          const float4 v0 = *reinterpret_cast<const float4*>(ar + 4 * r), v1 = *reinterpret_cast<const float4*>(ar + 64 + 4 * r);
          am[0] = v0.x; am[1] = v0.y; am[2] = v0.z; am[3] = v0.w; am[4] = v1.x; am[5] = v1.y; am[6] = v1.z; am[7] = v1.w;
        } else if constexpr (NB == 4) {
          const float4 v0 = *reinterpret_cast<const float4*>(ar + 4 * r);
          am[0] = v0.x; am[1] = v0.y; am[2] = v0.z; am[3] = v0.w;
        } else if constexpr (NB == 2) {
          const float2 v0 = *reinterpret_cast<const float2*>(ar + 2 * r);
          am[0] = v0.x; am[1] = v0.y;
        } else {
          am[0] = ar[r];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, am[j], acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, am[j], acc[1][j], 0, 0, 0);
        }
      }
      HIFIHR_RAW_BARRIER();                                  // barrier gc
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) store_block(j);
  };
  for (int ti = 0; ti < ntiles; ++ti) {
    TnTile t;
    mw.next(a, t);
    switch (t.nb) {
      case 8: run_tile(std::integral_constant<int, 8>{}, t); break;
      case 4: run_tile(std::integral_constant<int, 4>{}, t); break;
      case 2: run_tile(std::integral_constant<int, 2>{}, t); break;
      default: run_tile(std::integral_constant<int, 1>{}, t); break;
    }
  }
}

__global__ __launch_bounds__(512) void bgemm_tn_rows_kernel(BgemmArgs a, long per) {
  __shared__ __attribute__((aligned(1024))) float lds[4 * kRowsStage];
  tn_rows_body(a, per, lds, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// An NT product and a TN product that do not depend on each other -- the backward-data product V2 . U'^T and the backward-weight product
// Y'^T . V of ONE Winograd layer -- in ONE launch (round 5): workgroups [0, ga) walk the NT shares, [ga, gridDim.x) the TN shares.  A
// launch of these kernels on the narrow layers is shaped by its ends (256 workgroups ask for their first three
// chunks at once, ~4 us before the first MFMA; the last tiles' stores drain with nothing to overlap them; ~4 us of launch floor): side by
// side the two products share one start and one end, and every workgroup owns twice the tiles.  Same bodies, same per-tile arithmetic:
// results identical to the separate launches.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void bgemm_nt_tn_pair_kernel(BgemmArgs a, long per_a, int ga, BgemmArgs b, long per_b) {
  __shared__ __attribute__((aligned(1024))) float lds[4 * kRowsStage];
  if ((int)blockIdx.x < ga) nt_rows_body<0>(a, per_a, lds, (int)blockIdx.x, ga);
  else tn_rows_body(b, per_b, lds, (int)blockIdx.x - ga, (int)gridDim.x - ga);
}

static int gemm_cus() {
  // HIFIHR_GEMM_CUS (tests: the emulator reports 4 compute units, and the XCD-coherent TN schedule needs a multiple of 8 workgroups)
  if (const char* e = getenv("HIFIHR_GEMM_CUS")) { const int v = atoi(e); if (v > 0) return v; }
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// loader waves of the wave-specialised kernels (0: the 4-wave kernels); HIFIHR_GEMM_WS overrides (tuning)
static int gemm_ws_loaders() {
  if (const char* e = getenv("HIFIHR_GEMM_WS")) return atoi(e);
  return 4;
}

// Tile choice, measured on MI355X at B = 32 (tools/time_gemm.py, profiles/r02_time_gemm.txt):
//   NT  reduction length >= 512: 128x128 wave-specialised (4 loader waves); shorter reductions (4-8 chunks per tile): the 4-wave
//       64x64 kernel -- a tile's fixed cost (first loads, store tail) then weighs more than the macro-tile's operand reuse, and
//       four small workgroups per CU overlap it.
//   TN  >= 512 x 256 outputs: 128x128 wave-specialised, one workgroup per CU (slabs = 256 / tiles); smaller outputs: 64x64 tiles
//       with the t range split until ~512 workgroups exist.
static void nt_tile(int M, int N, int K, int* bm, int* bn) {
  (void)M;
  if (K >= 512 && N % 128 == 0) { *bm = 128; *bn = 128; }
  else { *bm = 64; *bn = 64; }
}
static int gemm_cus();
static void tn_tile(int M, int N, int batch, int* bm, int* bn) {
  *bm = 64; *bn = 64;
  if ((long)M * N >= 512L * 256 && M % 128 == 0 && N % 128 == 0) {
    // 128x128 tiles run one workgroup per CU: more tiles than CUs only pays when the rounds come out nearly whole.  The 36 problems of a
    // Winograd F(4x4, 3x3) layer give 288 / 576 tiles (1.125 / 2.25 rounds): the 64x64 kernel (several workgroups per CU) is faster there
    // (tools/time_gemm_tn_f4.py: 85 -> 69 us at 256 x 512 channels, 136 -> 131 at 512 x 512)
    const long t128 = (long)(M / 128) * (N / 128) * batch;
    const int cus = gemm_cus();
    const long rounds = (t128 + cus - 1) / cus;
    if (t128 <= cus || (double)rounds * cus <= 1.15 * (double)t128) { *bm = 128; *bn = 128; }
  }
}

size_t bgemm_nt_workspace_bytes(int M, int N, int K, int batch);

// the persistent row-share kernel serves every NT product whose N is a multiple of 128 (HIFIHR_GEMM_ROWS=0: the older kernels)
static bool nt_rows(int N) {
  static const int on = [] { const char* e = getenv("HIFIHR_GEMM_ROWS"); return e ? atoi(e) : 1; }();
  return on && N % 128 == 0 && getenv("HIFIHR_GEMM_NT_TILE") == nullptr;
}

// the persistent row-share TN kernel: complete products (one slab) when every CU gets at least eight 16-row blocks (HIFIHR_GEMM_TN_ROWS=0:
// the per-tile kernels with T-split slabs)
static bool tn_rows(int M, int N, int T, int batch) {
  static const int on = [] { const char* e = getenv("HIFIHR_GEMM_TN_ROWS"); return e ? atoi(e) : 1; }();
  if (!on || N % 128 != 0 || M % 16 != 0 || T % 32 != 0 || T < 64) return false;
  if (getenv("HIFIHR_GEMM_TN_TILE") != nullptr || getenv("HIFIHR_GEMM_TN_PARTS") != nullptr) return false;
  return (long)batch * (N / 128) * (M / 16) >= 8L * gemm_cus();      // (256 x 256 channels, 4.5 blocks per CU: 38 us here, 35 on the 64x64 kernel)
}

// T-split on the same kernel (round 5): a product with too few 16-row blocks for the row-share schedule -- the 1x1 backward-weight
// [6272 x 512]^T . [6272 x 256] is 64 blocks, the 36 products of a 128-channel F(4x4) layer 288 -- is cut along T into P parts that
// the kernel walks as P problems of their own (part s of problem p starts at A + (p P + s) T' M: the plain problem stride), each
// landing in slab s; the consumers sum the slabs in slab order, as they do behind the per-tile kernels these shapes ran on
// (bgemm_ws_kernel<128,128,true>: 30 TF on the 1x1 product; bgemm_tn_kernel<64,64>: 0.43 of the peak).  P = the smallest divisor of
// the chunk count that gives every CU ~7 blocks with at least 4 chunks per part.  HIFIHR_GEMM_TN_SPLIT=0: the per-tile kernels.
// MEASURED (tools/time_gemm_tn_split.py, gpurun_out/tn_split_*.txt): the 36 products of a 128-channel F(4x4) layer 26.7 -> 25.4 us, of a
// 256-channel one 35.8 -> 31.6 us; SINGLE products lose (1x1 backward-weight 256 -> 512 at 14 x 14: 22.9 -> 25.4 us, 512 -> 512: 36.8 ->
// 42.3 us: 28 / 14 slabs of a whole filter each) -- batched products only.
// Long reductions (VGG19's layers at 112 x 112 / 56 x 56: T = 9 408 ... 37 632) stay on the per-tile kernels too: config 3 measured
// 34.41 ms/step without the split against 34.49 / 34.53 with it (gpurun_out/c3_*.json).
static int tn_rows_split(int M, int N, int T, int batch) {
  if (batch < 2 || T > 4096) return 0;
  static const int on = [] { const char* e = getenv("HIFIHR_GEMM_TN_SPLIT"); return e ? atoi(e) : 1; }();
  static const int on_rows = [] { const char* e = getenv("HIFIHR_GEMM_TN_ROWS"); return e ? atoi(e) : 1; }();
  if (!on || !on_rows || N % 128 != 0 || M % 16 != 0 || T % 32 != 0 || T < 64) return 0;
  if (getenv("HIFIHR_GEMM_TN_TILE") != nullptr || getenv("HIFIHR_GEMM_TN_PARTS") != nullptr) return 0;
  const int nch = T / 32;
  const long blocks = (long)batch * (N / 128) * (M / 16), need = 7L * gemm_cus();
  for (int P = 2; P <= nch / 4; ++P)
    if (nch % P == 0 && blocks * P >= need) return P;
  return 0;
}

// which kernel instantiation a shape runs on, as rocprof names it (bench.py groups its roofline lines by this)
void bgemm_describe(int tn, int M, int N, int K, char* out, int cap) { bgemm_describe_batch(tn, M, N, K, 16, out, cap); }

void bgemm_describe_batch(int tn, int M, int N, int K, int batch, char* out, int cap) {
  int bm, bn;
  const char* e;
  if (tn) {
    tn_tile(M, N, batch, &bm, &bn);
    if ((e = getenv("HIFIHR_GEMM_TN_TILE")) != nullptr) { const int v = atoi(e); bm = v / 1000; bn = v % 1000; if (M % bm) bm = 64; if (N % bn) bn = 64; }
  } else {
    nt_tile(M, N, K, &bm, &bn);
    if ((e = getenv("HIFIHR_GEMM_NT_TILE")) != nullptr) { const int v = atoi(e); bm = v / 1000; bn = v % 1000; if (N % bn) bn = 64; }
  }
  const int nload = gemm_ws_loaders();
  // (the instantiation rocprof lists: <0> plain, <1> ragged N / K; <2> = the gathering form, named by hifihr_conv2d_describe)
  if (!tn && (nt_rows(N) || bgemm_nt_ragged_supported(M, N, K))) { snprintf(out, cap, "bgemm_nt_rows_kernel<%d>", bgemm_nt_ragged_supported(M, N, K) ? 1 : 0); return; }
  if (tn && (tn_rows(M, N, K, batch) || tn_rows_split(M, N, K, batch))) { snprintf(out, cap, "bgemm_tn_rows_kernel"); return; }
  if (!tn && bm == 128 && bn == 128 && nload > 0 && bgemm_nt_workspace_bytes(M, N, K, 16) > 0) snprintf(out, cap, "bgemm_nt_sk_kernel<%d>", nload == 2 ? 2 : 4);
  else if (bm == 128 && bn == 128 && nload > 0) snprintf(out, cap, "bgemm_ws_kernel<128, 128, %s, %d>", tn ? "true" : "false", nload == 1 ? 1 : nload == 4 ? 4 : 2);
  else snprintf(out, cap, "%s<%d, %d>", tn ? "bgemm_tn_kernel" : "bgemm_nt_kernel", bm, bn);
}

bool bgemm_nt_supported(int M, int N, int K) { return M > 0 && K >= 32 && K % 32 == 0 && N >= 64 && N % 64 == 0; }
bool bgemm_tn_supported(int M, int N, int T) { return T > 0 && M >= 64 && M % 64 == 0 && N >= 64 && N % 64 == 0; }

static size_t sk_flag_bytes(int G) { return (size_t)((G + 1) * 4 * sizeof(unsigned) + 255) / 256 * 256; }

// bytes of zero-initialised, self-cleaning workspace the persistent NT kernel wants for this shape (0: the shape runs on a
// kernel that needs none)
size_t bgemm_nt_workspace_bytes(int M, int N, int K, int batch) {
  if (!bgemm_nt_supported(M, N, K) || batch <= 0 || gemm_ws_loaders() <= 0) return 0;
  if (nt_rows(N)) return 0;                    // whole tiles only: nothing is exchanged between workgroups
  if (const char* e = getenv("HIFIHR_GEMM_SK")) { if (atoi(e) == 0) return 0; }
  int bm, bn;
  nt_tile(M, N, K, &bm, &bn);
  if (const char* e = getenv("HIFIHR_GEMM_NT_TILE")) { const int v = atoi(e); bm = v / 1000; bn = v % 1000; if (N % bn) bn = 64; }
  if (bm != 128 || bn != 128) return 0;
  const int G = gemm_cus(), nch = K / 32;
  const long total = (long)((M + 127) / 128) * (N / 128) * batch * nch;
  const long per = (total + G - 1) / G;
  if (per < nch) return 0;                    // shares shorter than a tile would split it three ways: the per-tile kernel instead
  // measured (tools/time_gemm.py, profiles/r02_time_gemm.txt): 832 tiles x 16 chunks 141.5 -> 132.7 us, but 416 x 16 71.7 -> 76.3 and the
  // 4- / 8-chunk tiles much worse (the tile epilogue stalls the whole workgroup, nothing else is resident on the CU to cover it):
  // only where the per-tile kernel would run >= 3 rounds
  if (total / nch < 3L * G) return 0;
  return sk_flag_bytes(G) + (size_t)G * 128 * 128 * sizeof(float);
}

// ragged N / K on the row-share kernel (EfficientNet's 1x1 convolutions).  Measured at batch 48 against conv_igemm_kernel
// (tools/time_conv1x1.py with EFFNET=1, HIFIHR_GEMM_RAGGED=0 for the other side): a wash on most shapes -- these products are 20-50 us
// launches on 2 352-9 408 rows, bounded by their size, and the implicit GEMM's 64-column tiles waste less of a 136- or 232-wide output --
// a win where the 128-column tiles are >= 90 % full and the reduction is long (1392 -> 384: 45 -> 35 us, 232 -> 1392: 28.5 -> 25.7), a loss
// below (32 -> 192: 45 -> 54, 576 -> 136: 30 -> 35).  Hence: tiles >= 90 % full and K >= 128 (HIFIHR_GEMM_RAGGED=2: every shape the kernel
// can take, for the A/B).
bool bgemm_nt_ragged_supported(int M, int N, int K) {
  static const int on = [] { const char* e = getenv("HIFIHR_GEMM_RAGGED"); return e ? atoi(e) : 1; }();
  if (!on || !nt_rows(128) || M <= 0 || N < 96 || N % 4 != 0 || K < 16 || K % 4 != 0 || (N % 128 == 0 && K % 32 == 0)) return false;
  if (on >= 2) return true;
  return K >= 128 && 10L * N >= 9L * ((N + 127) / 128 * 128);
}
bool bgemm_nt_stats_supported(int N) { return nt_rows(N) || (nt_rows(128) && N >= 96 && N % 4 == 0); }      // the statistics epilogue exists in the row-share kernel only

// ---- forward convolutions on the row-share kernel with the gather in its loader waves (MODE 2) ----
bool conv_rows_supported(const ConvGeom& g, const float* bias) {
  static const int on = [] { const char* e = getenv("HIFIHR_CONV_ROWS"); return e ? atoi(e) : 1; }();
  if (!on || !nt_rows(128) || g.dgrad || bias != nullptr || g.relu || g.batch > 1) return false;
  if (g.IC % 32 != 0 || g.OC % 128 != 0 || g.R != g.S || (g.R != 1 && g.R != 3)) return false;
  const long M = (long)g.N * g.OH * g.OW;
  if (M >= (1L << 31) || (long)g.N * g.IH * g.IW * g.IC >= (1L << 31)) return false;
  if (on >= 2) return true;                                  // (every shape the kernel takes: the A/B)
  return g.stride == 2;                                      // stride 1: 1x1 is a plain GEMM already, 3x3 runs as Winograd / on the halo kernel
}

static BgemmArgs conv_rows_args(const ConvGeom& g, const float* src, const float* wgt, float* dst, float* stats, const float* zeros) {
  BgemmArgs a{};
  const long M = (long)g.N * g.OH * g.OW;
  a.A = src; a.B = wgt; a.C = dst; a.M = (int)M; a.N = g.OC; a.K = g.R * g.S * g.IC;
  a.lda = 0; a.ldb = a.K; a.ldc = g.OC; a.sa = 0; a.sb = 0; a.sc = 0; a.batch = 1;
  a.stats = stats; a.zeros = zeros;
  a.cIH = g.IH; a.cIW = g.IW; a.cC = g.IC; a.cOH = g.OH; a.cOW = g.OW; a.cR = g.R; a.cS = g.S; a.cStride = g.stride; a.cPad = g.pad;
  a.tiles_n = g.OC / 128; a.tiles_m = (int)((M + 127) / 128); a.splits = 1; a.cps = a.K / 32; a.sc_split = 0;
  return a;
}
// shares of `cus` workgroups over the flattened (column tile, row) space of a gathered product, whole 16-row blocks each
static int conv_rows_shares(const BgemmArgs& a, int cus, long* per_out) {
  const long total = (long)a.tiles_n * a.M;
  long per = (total + cus - 1) / cus;
  per = (per + 15) / 16 * 16;
  if (per < 16) per = 16;
  *per_out = per;
  return (int)((total + per - 1) / per);
}

hipError_t launch_conv_rows(const ConvGeom& g, const float* src, const float* wgt, float* dst, float* stats, const float* zeros, hipStream_t st) {
  if (!conv_rows_supported(g, nullptr) || zeros == nullptr) return hipErrorInvalidValue;
  const BgemmArgs a = conv_rows_args(g, src, wgt, dst, stats, zeros);
  long per;
  const int G = conv_rows_shares(a, gemm_cus(), &per);
  hipLaunchKernelGGL(bgemm_nt_rows_kernel<2>, dim3(G), dim3(512), 0, st, a, per);
  return hipGetLastError();
}

// TWO gathered products on the same image in ONE launch (round 6): the strided 3x3 convolution of a residual stage's first block and the
// stride-2 1x1 convolution of its downsample branch read the same x and do not depend on each other; the 1x1 one is a 10-15 us launch that is
// all ends (0.4 GFLOP on the whole chip).  Workgroups [0, ga) walk the first product's shares, the rest the second's, ga in proportion
// to the flops: same per-tile arithmetic, results identical to the separate launches.  HIFIHR_CONV_ROWS_PAIR=0: two launches.
__global__ __launch_bounds__(512) void bgemm_nt_rows_pair2_kernel(BgemmArgs a, long per_a, int ga, BgemmArgs b, long per_b) {
  __shared__ __attribute__((aligned(1024))) float lds[4 * kRowsStage];
  if ((int)blockIdx.x < ga) nt_rows_body<2>(a, per_a, lds, (int)blockIdx.x, ga);
  else nt_rows_body<2>(b, per_b, lds, (int)blockIdx.x - ga, (int)gridDim.x - ga);
}
bool conv_rows_pair_supported(const ConvGeom& g1, const ConvGeom& g2) {
  static const int on = [] { const char* e = getenv("HIFIHR_CONV_ROWS_PAIR"); return e ? atoi(e) : 1; }();
  return on && conv_rows_supported(g1, nullptr) && conv_rows_supported(g2, nullptr) && g1.N == g2.N && g1.IH == g2.IH && g1.IW == g2.IW &&
         g1.IC == g2.IC && gemm_cus() >= 16;
}
hipError_t launch_conv_rows_pair(const ConvGeom& g1, const float* src, const float* w1, float* y1, float* stats1, const ConvGeom& g2, const float* w2,
                                 float* y2, float* stats2, const float* zeros, hipStream_t st) {
  if (!conv_rows_pair_supported(g1, g2) || zeros == nullptr) return hipErrorInvalidValue;
  const BgemmArgs a = conv_rows_args(g1, src, w1, y1, stats1, zeros), b = conv_rows_args(g2, src, w2, y2, stats2, zeros);
  const double fa = (double)a.M * a.N * a.K, fb = (double)b.M * b.N * b.K;
  const int cus = gemm_cus();
  int ga = (int)(cus * fa / (fa + fb) + 0.5);
  if (ga < 8) ga = 8;
  if (ga > cus - 8) ga = cus - 8;
  long per_a, per_b;
  ga = conv_rows_shares(a, ga, &per_a);
  const int gb = conv_rows_shares(b, cus - ga, &per_b);
  hipLaunchKernelGGL(bgemm_nt_rows_pair2_kernel, dim3(ga + gb), dim3(512), 0, st, a, per_a, ga, b, per_b);
  return hipGetLastError();
}

hipError_t launch_bgemm_nt(const float* A, const float* B, float* C, int M, int N, int K, int batch, void* ws, size_t ws_bytes, hipStream_t st,
                           float* stats, int M_alloc) {
  // M_alloc > M: the problems of the batch are M_alloc rows apart in A and C, only the first M of each are computed (the mosaic tile
  // count of the F(4x4, 3x3) pipeline is rounded up for the backward-weight products; the row-share kernel walks the real rows)
  if (M_alloc > M && nt_rows(N) && !bgemm_nt_ragged_supported(M, N, K) && bgemm_nt_supported(M_alloc, N, K) && batch > 0 && stats == nullptr &&
      (long)M_alloc * K < (1L << 31) && (long)N * K < (1L << 31)) {
    BgemmArgs a{};
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
    a.sa = (long)M_alloc * K; a.sb = (long)N * K; a.sc = (long)M_alloc * N; a.batch = batch;
    a.tiles_n = N / 128; a.tiles_m = (M + 127) / 128; a.splits = 1; a.cps = K / 32; a.sc_split = 0;
    const long total = (long)batch * a.tiles_n * M;
    const int cus = gemm_cus();
    long per = (total + cus - 1) / cus;
    if (per < 16) per = 16;
    const int G = (int)((total + per - 1) / per);
    hipLaunchKernelGGL(bgemm_nt_rows_kernel<0>, dim3(G), dim3(512), 0, st, a, per);
    return hipGetLastError();
  }
  if (M_alloc > M) M = M_alloc;
  const bool ragged = bgemm_nt_ragged_supported(M, N, K);
  if ((!bgemm_nt_supported(M, N, K) && !ragged) || batch <= 0) return hipErrorInvalidValue;
  if (stats != nullptr && (batch != 1 || !(nt_rows(N) || ragged))) return hipErrorInvalidValue;
  if ((long)M * K >= (1L << 31) || (long)N * K >= (1L << 31)) return hipErrorInvalidValue;      // 32-bit element offsets
  BgemmArgs a{};
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
  a.sa = (long)M * K; a.sb = (long)N * K; a.sc = (long)M * N; a.batch = batch;
  a.stats = stats;
  if (ragged) {
    a.zeros = conv_halo_zero_page(st);                       // (allocated on first use outside a capture: csrc/conv_halo.hip)
    if (a.zeros == nullptr) return hipErrorNotReady;
    a.tiles_n = (N + 127) / 128; a.tiles_m = (M + 127) / 128; a.splits = 1; a.cps = (K + 31) / 32; a.sc_split = 0;
    const long total = (long)batch * a.tiles_n * M;
    const int cus = gemm_cus();
    long per = (total + cus - 1) / cus;
    if (per < 16) per = 16;
    const int G = (int)((total + per - 1) / per);
    hipLaunchKernelGGL(bgemm_nt_rows_kernel<1>, dim3(G), dim3(512), 0, st, a, per);
    return hipGetLastError();
  }
  if (nt_rows(N)) {
    a.tiles_n = N / 128; a.tiles_m = (M + 127) / 128; a.splits = 1; a.cps = K / 32; a.sc_split = 0;
    const long total = (long)batch * a.tiles_n * M;
    const int cus = gemm_cus();
    long per = (total + cus - 1) / cus;
    if (per < 16) per = 16;
    const int G = (int)((total + per - 1) / per);
    hipLaunchKernelGGL(bgemm_nt_rows_kernel<0>, dim3(G), dim3(512), 0, st, a, per);
    return hipGetLastError();
  }
  int bm, bn;
  nt_tile(M, N, K, &bm, &bn);
  if (const char* e = getenv("HIFIHR_GEMM_NT_TILE")) { const int v = atoi(e); bm = v / 1000; bn = v % 1000; if (N % bn) bn = 64; }
  a.tiles_m = (M + bm - 1) / bm; a.tiles_n = N / bn; a.splits = 1; a.cps = K / 32; a.sc_split = 0;
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n * batch));
  const int nload = gemm_ws_loaders();
  if (nload > 0 && bm == 128 && bn == 128 && ws != nullptr && ws_bytes >= bgemm_nt_workspace_bytes(M, N, K, batch) &&
      bgemm_nt_workspace_bytes(M, N, K, batch) > 0) {
    const int G = gemm_cus();
    unsigned* flags = static_cast<unsigned*>(ws);
    float* slabs = reinterpret_cast<float*>(static_cast<char*>(ws) + sk_flag_bytes(G));
    if (nload == 2) hipLaunchKernelGGL((bgemm_nt_sk_kernel<2>), dim3(G), dim3(384), 0, st, a, slabs, flags);
    else hipLaunchKernelGGL((bgemm_nt_sk_kernel<4>), dim3(G), dim3(512), 0, st, a, slabs, flags);
    return hipGetLastError();
  }
  if (nload > 0 && bm == 128 && bn == 128) {
    if (nload == 1) hipLaunchKernelGGL((bgemm_ws_kernel<128, 128, false, 1>), grid, dim3(320), 0, st, a);
    else if (nload == 4) hipLaunchKernelGGL((bgemm_ws_kernel<128, 128, false, 4>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((bgemm_ws_kernel<128, 128, false, 2>), grid, dim3(384), 0, st, a);
    return hipGetLastError();
  }
  if (bm == 128 && bn == 128) hipLaunchKernelGGL((bgemm_nt_kernel<128, 128>), grid, dim3(256), 0, st, a);
  else if (bm == 128 && bn == 64) hipLaunchKernelGGL((bgemm_nt_kernel<128, 64>), grid, dim3(256), 0, st, a);
  else if (bm == 64 && bn == 128) hipLaunchKernelGGL((bgemm_nt_kernel<64, 128>), grid, dim3(256), 0, st, a);
  else if (bm == 64 && bn == 64) hipLaunchKernelGGL((bgemm_nt_kernel<64, 64>), grid, dim3(256), 0, st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// number of K-split slabs launch_bgemm_tn writes for this shape (the caller provides parts * batch * M * N floats)
int bgemm_tn_parts(int M, int N, int T, int batch) {
  if (tn_rows(M, N, T, batch)) return 1;
  if (const int P = tn_rows_split(M, N, T, batch)) return P;
  if (const char* e = getenv("HIFIHR_GEMM_TN_PARTS")) { const int v = atoi(e); if (v > 0) return v; }
  int bm, bn;
  tn_tile(M, N, batch, &bm, &bn);
  if (const char* e = getenv("HIFIHR_GEMM_TN_TILE")) { const int v = atoi(e); bm = v / 1000; bn = v % 1000; if (M % bm) bm = 64; if (N % bn) bn = 64; }
  const int tiles = (M / bm) * (N / bn) * batch, nch = (T + 31) / 32;
  // 128x128 (one workgroup per CU): fill the CUs once; 64x64: ~2 workgroups per CU; at least 7 chunks per slab so that the slab
  // round trip (written here, summed by wino_dw_transform_parts) stays small next to the reduction.  With the 36 problems of an
  // F(4x4, 3x3) layer the 64x64 kernel does best at ~4.5 workgroups per CU (tools/time_gemm_tn_f4.py).
  int splits = ((bm == 128 && bn == 128) ? gemm_cus() : batch > 16 ? (9 * gemm_cus() / 2 + tiles / 2) : 2 * gemm_cus()) / tiles;
  if (splits > nch / 7) splits = nch / 7;      // (7, not 8: the 49 chunks of a 28 x 28 F(4x4) layer split 7 x 7 -- 30.6 -> 25.9 us)
  if (splits < 1) splits = 1;
  const int cps = (nch + splits - 1) / splits;
  return (nch + cps - 1) / cps;
}

// XCD-coherent schedule of the TN row-share kernel for G workgroups (BgemmArgs::co_r): the smallest number r of problems per XCD and round
// that gives every workgroup of the XCD at least one whole 128-row tile per round, while the operands of those r problems fit the XCD's L2
// and at least one full round exists.  HIFIHR_GEMM_TN_COHERENT=0: contiguous shares.
static void tn_coherent(BgemmArgs& a, int G) {
  static const int on = [] { const char* e = getenv("HIFIHR_GEMM_TN_COHERENT"); return e ? atoi(e) : 1; }();
  a.co_r = 0; a.co_rounds = 0;
  if (!on || G < 8 || G % 8 != 0 || a.M % 16 != 0) return;
  const int W = G / 8;
  const long bpp = (long)a.tiles_n * (a.M / 16);
  if (bpp <= 8) return;                                     // one tile per problem: no two workgroups share a panel
  const double mb_per_problem = 4.0 * a.K * ((double)a.M + 128.0 * a.tiles_n) / (1 << 20);
  for (int r = 1; r <= 8; ++r) {
    if ((r * bpp) % W != 0 || r * bpp / W < 8) continue;
    if (r * mb_per_problem > 3.5 || a.batch / (8 * r) < 1) return;
    a.co_r = r; a.co_rounds = a.batch / (8 * r);
    return;
  }
}

hipError_t launch_bgemm_tn(const float* A, const float* B, float* Cparts, int M, int N, int T, int batch, int parts, hipStream_t st, int T_valid) {
  if (!bgemm_tn_supported(M, N, T) || batch <= 0 || parts <= 0 || T_valid < 0 || T_valid > T) return hipErrorInvalidValue;
  BgemmArgs a{};
  static const int skip_on = [] { const char* e = getenv("HIFIHR_GEMM_TN_SKIP"); return e ? atoi(e) : 1; }();      // (0: walk the zero rows too, A/B timing)
  a.k_valid = (skip_on && T_valid < T) ? T_valid : 0;        // (row-share kernel only: the per-tile kernels walk every row)
  a.A = A; a.B = B; a.C = Cparts; a.M = M; a.N = N; a.K = T; a.lda = M; a.ldb = N; a.ldc = N;
  a.sa = (long)T * M; a.sb = (long)T * N; a.sc = (long)M * N; a.batch = batch;
  const int P = tn_rows(M, N, T, batch) ? 1 : tn_rows_split(M, N, T, batch);
  if (P > 0) {
    if (parts != P) return hipErrorInvalidValue;
    const int Tp = T / P;                                    // (P == 1: the whole reduction per tile, one slab)
    a.K = Tp; a.sa = (long)Tp * M; a.sb = (long)Tp * N; a.batch = batch * P;
    a.tiles_n = N / 128; a.tiles_m = (M + 127) / 128; a.splits = P; a.cps = Tp / 32; a.sc_split = P > 1 ? (long)batch * M * N : 0;
    const long total = (long)a.batch * a.tiles_n * (M / 16);
    const int cus = gemm_cus();
    long per = (total + cus - 1) / cus;
    if (per < 4) per = 4;
    const int G = (int)((total + per - 1) / per);
    tn_coherent(a, G);
    hipLaunchKernelGGL(bgemm_tn_rows_kernel, dim3(G), dim3(512), 0, st, a, per);
    return hipGetLastError();
  }
  int bm, bn;
  tn_tile(M, N, batch, &bm, &bn);
  if (const char* e = getenv("HIFIHR_GEMM_TN_TILE")) { const int v = atoi(e); bm = v / 1000; bn = v % 1000; if (M % bm) bm = 64; if (N % bn) bn = 64; }
  a.tiles_m = M / bm; a.tiles_n = N / bn;
  const int nch = (T + 31) / 32;
  a.cps = (nch + parts - 1) / parts;
  if ((nch + a.cps - 1) / a.cps != parts) return hipErrorInvalidValue;       // parts must come from bgemm_tn_parts
  a.splits = parts; a.sc_split = (long)batch * M * N;
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n * batch * parts));
  const int nload = gemm_ws_loaders();
  if (nload > 0 && bm == 128 && bn == 128) {
    if (nload == 1) hipLaunchKernelGGL((bgemm_ws_kernel<128, 128, true, 1>), grid, dim3(320), 0, st, a);
    else if (nload == 4) hipLaunchKernelGGL((bgemm_ws_kernel<128, 128, true, 4>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((bgemm_ws_kernel<128, 128, true, 2>), grid, dim3(384), 0, st, a);
    return hipGetLastError();
  }
  if (bm == 128 && bn == 128) hipLaunchKernelGGL((bgemm_tn_kernel<128, 128>), grid, dim3(256), 0, st, a);
  else if (bm == 128 && bn == 64) hipLaunchKernelGGL((bgemm_tn_kernel<128, 64>), grid, dim3(256), 0, st, a);
  else if (bm == 64 && bn == 128) hipLaunchKernelGGL((bgemm_tn_kernel<64, 128>), grid, dim3(256), 0, st, a);
  else if (bm == 64 && bn == 64) hipLaunchKernelGGL((bgemm_tn_kernel<64, 64>), grid, dim3(256), 0, st, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// The pair launch (bgemm_nt_tn_pair_kernel): C[p][m][n] = sum_k A[p][m][k] B[p][n][k] (plain row-share form: N % 128 == 0, K % 32 == 0) AND
// C2parts = A2^T . B2 (row-share TN form, one slab or the T-split) in ONE launch.  hipErrorNotSupported: one of the two is not on its
// row-share kernel (the caller then launches them separately).  The CUs are divided in proportion to the products' flops, the NT side
// weighted by HIFIHR_GEMM_PAIR_NT_WEIGHT / 100 (default 100; measured on the ResNet-18 step: 70 -> 5.33 ms, 85 -> 5.21, 100 -> 5.17, 110 -> 5.22, 130 -> 5.50; separate launches 5.27).
bool bgemm_nt_tn_pair_supported(int M, int M_alloc, int N, int K, int batch, int M2, int N2, int T2, int batch2, int parts2) {
  static const int on = [] { const char* e = getenv("HIFIHR_GEMM_PAIR"); return e ? atoi(e) : 1; }();
  if (M_alloc < M) M_alloc = M;
  if (!on || !nt_rows(N) || bgemm_nt_ragged_supported(M, N, K) || !bgemm_nt_supported(M_alloc, N, K) || batch <= 0 ||
      (long)M_alloc * K >= (1L << 31) || (long)N * K >= (1L << 31))
    return false;
  if (!bgemm_tn_supported(M2, N2, T2) || batch2 <= 0 || parts2 <= 0) return false;
  const int P = tn_rows(M2, N2, T2, batch2) ? 1 : tn_rows_split(M2, N2, T2, batch2);
  if (P <= 0 || P != parts2) return false;
  // Long products gain nothing from sharing a launch (their ends are a small part of them): VGG19's layers at 112 x 112 / 56 x 56 (22-44
  // GFLOP each) measured 33.71 ms/step apart against 33.75 paired (config 3); the ResNet layers (1.9-8.5 GFLOP) 5.25 -> 5.16 ms/step.
  static const double max_gf = [] { const char* e = getenv("HIFIHR_GEMM_PAIR_MAX_GFLOP"); const double v = e ? atof(e) : 12.0; return v > 0 ? v : 12.0; }();
  return 2.0 * batch * (double)M * N * K <= max_gf * 1e9;
}

hipError_t launch_bgemm_nt_tn_pair(const float* A, const float* B, float* C, int M, int M_alloc, int N, int K, int batch, const float* A2,
                                   const float* B2, float* C2parts, int M2, int N2, int T2, int batch2, int parts2, hipStream_t st, int T2_valid) {
  if (!bgemm_nt_tn_pair_supported(M, M_alloc, N, K, batch, M2, N2, T2, batch2, parts2)) return hipErrorNotSupported;
  if (T2_valid < 0 || T2_valid > T2) return hipErrorInvalidValue;
  if (M_alloc < M) M_alloc = M;                               // (M_alloc > M: problems M_alloc rows apart, M of them computed -- see launch_bgemm_nt)
  const int P = parts2;
  BgemmArgs a{};
  a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
  a.sa = (long)M_alloc * K; a.sb = (long)N * K; a.sc = (long)M_alloc * N; a.batch = batch;
  a.tiles_n = N / 128; a.tiles_m = (M + 127) / 128; a.splits = 1; a.cps = K / 32; a.sc_split = 0;
  BgemmArgs b{};
  const int Tp = T2 / P;
  b.A = A2; b.B = B2; b.C = C2parts; b.M = M2; b.N = N2; b.K = Tp; b.lda = M2; b.ldb = N2; b.ldc = N2;
  b.sa = (long)Tp * M2; b.sb = (long)Tp * N2; b.sc = (long)M2 * N2; b.batch = batch2 * P;
  b.tiles_n = N2 / 128; b.tiles_m = (M2 + 127) / 128; b.splits = P; b.cps = Tp / 32; b.sc_split = P > 1 ? (long)batch2 * M2 * N2 : 0;
  static const int skip_on = [] { const char* e = getenv("HIFIHR_GEMM_TN_SKIP"); return e ? atoi(e) : 1; }();
  b.k_valid = (skip_on && T2_valid > 0 && T2_valid < T2) ? T2_valid : 0;
  const int T2w = b.k_valid > 0 ? (b.k_valid + 3) / 4 * 4 : T2;          // rows whose k-steps run (the split of the CUs follows the work)
  static const int wnt = [] { const char* e = getenv("HIFIHR_GEMM_PAIR_NT_WEIGHT"); const int v = e ? atoi(e) : 100; return v > 0 ? v : 100; }();
  const double fa = 2.0 * batch * (double)M * N * K * (wnt / 100.0), fb = 2.0 * batch2 * (double)M2 * N2 * T2w;
  const int cus = gemm_cus();
  int ga = (int)(cus * fa / (fa + fb) + 0.5);
  if (ga < 8) ga = 8;
  if (ga > cus - 8) ga = cus - 8;
  int gb = cus - ga;
  const long total_a = (long)batch * a.tiles_n * M;
  long per_a = (total_a + ga - 1) / ga;
  if (per_a < 16) per_a = 16;
  ga = (int)((total_a + per_a - 1) / per_a);
  const long total_b = (long)b.batch * b.tiles_n * (M2 / 16);
  long per_b = (total_b + gb - 1) / gb;
  if (per_b < 4) per_b = 4;
  gb = (int)((total_b + per_b - 1) / per_b);
  tn_coherent(b, gb);
  hipLaunchKernelGGL(bgemm_nt_tn_pair_kernel, dim3(ga + gb), dim3(512), 0, st, a, per_a, ga, b, per_b);
  return hipGetLastError();
}


#if defined(HIFIHR_PROBE_SPLIT_BF16)
// PROBE build only (tools/build_split_bf16_probe.sh): x[rows][K] f32 -> the split image of nt_rows_body<3> (per row and 32-deep chunk: 32 bf16
// leading pieces, then the 32 bf16 remainders; round-to-nearest-even both), and the product on it.
__global__ __launch_bounds__(256) void probe_split_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;                  // element (row, k): chunk k / 32, position k % 32
  if (i >= n) return;
  const float v = x[i];
  const __bf16 hi = (__bf16)v;
  const __bf16 lo = (__bf16)(v - (float)hi);
  const size_t chunk = i >> 5, pos = i & 31;
  out[chunk * 64 + pos] = hi;
  out[chunk * 64 + 32 + pos] = lo;
}
__global__ __launch_bounds__(512) void bgemm_nt_rows_bf16x3_kernel(BgemmArgs a, long per) {
  __shared__ __attribute__((aligned(1024))) float lds[4 * kRowsStage];
  nt_rows_body<3>(a, per, lds, (int)blockIdx.x, (int)gridDim.x);
}
#endif
}  // namespace hifihr

#if defined(HIFIHR_GEMM_STAMP)
extern "C" int hifihr_gemm_stamp_read(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(hifihr::g_gemm_stamp), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hifihr::g_gemm_stamp), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

#if defined(HIFIHR_PROBE_SPLIT_BF16)
extern "C" int hifihr_probe_split_bf16(const float* x, void* out, long n, void* stream) {
  hipLaunchKernelGGL(hifihr::probe_split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<__bf16*>(out), (size_t)n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// c[b][M][N] = a[b][M][K] . b[b][N][K]^T on the split images (same byte sizes and strides as the f32 operands); N % 128 == 0, K % 64 == 0
extern "C" int hifihr_probe_bgemm_nt_bf16x3(const void* A, const void* B, float* C, int M, int N, int K, int batch, void* stream) {
  if (N % 128 != 0 || K % 64 != 0 || M <= 0 || batch <= 0) return -1;
  hifihr::BgemmArgs a{};
  a.A = reinterpret_cast<const float*>(A); a.B = reinterpret_cast<const float*>(B); a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
  a.sa = (long)M * K; a.sb = (long)N * K; a.sc = (long)M * N; a.batch = batch;
  a.tiles_n = N / 128; a.tiles_m = (M + 127) / 128; a.splits = 1; a.cps = K / 32; a.sc_split = 0;
  const long total = (long)batch * a.tiles_n * M;
  const int cus = hifihr::gemm_cus();
  long per = (total + cus - 1) / cus;
  if (per < 16) per = 16;
  const int G = (int)((total + per - 1) / per);
  hipLaunchKernelGGL(hifihr::bgemm_nt_rows_bf16x3_kernel, dim3(G), dim3(512), 0, (hipStream_t)stream, a, per);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif
