// NHWC fp32 convolution for gfx950 as implicit GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces what the reference's encoder dispatches to cuDNN/MIOpen (reference network/res_encoder.py:364-373:
// conv2d forward, backward-data, backward-weight of the ResNet trunk).  fp32 in, fp32 accumulate: the MFMA result
// is bit-for-bit a k-ordered fmaf chain, so numerics are those of a plain fp32 convolution.
//
//   conv_igemm_kernel<BM,BN>   Y[m][k] = sum_q A[m][q] W[k][q],  m = (n,oh,ow), q = (r,s,c).
//        The A tile is gathered on the fly (im2col never materialised): every lane loads 16 B (4 channels of one
//        tap) per row, tiles go through LDS k-interleaved so each lane reads its 8 k-steps as two ds_read_b128
//        (row stride 20 floats: conflict-free), double-buffered, one barrier per 16-deep K chunk, 32 MFMAs
//        per wave between barriers.  The same kernel computes backward-data (dgrad = 1): rows are input pixels
//        and the gather walks dy with the stride divisibility test, against the [C][R][S][K] transposed weight.
//   conv_wgrad_kernel<BM,BN>   dW[k][q] += sum_m dy[m][k] A[m][q]; the pixel range is split over blockIdx.z and
//        partial tiles are added with fp32 atomics (256 contiguous bytes per wave instruction).
//   weight_transpose_kernel    [K][R][S][C] -> [C][R][S][K] for dgrad.
//   image_to_nhwc4_kernel      normalize_batch_3C (reference network/res_encoder.py:212-216) fused with the
//                              NCHW(3) -> NHWC(4, zero padded) repack the first convolution consumes.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "hifihr_internal.h"

namespace hifihr {

#if defined(HIFIHR_HOSTSIM)
typedef hs_floatx16 floatx16;
#else
typedef float floatx16 __attribute__((ext_vector_type(16)));
#endif

#if defined(HIFIHR_HOSTSIM)
#define HIFIHR_KEEP(x) ((void)0)
#define HIFIHR_SCHED_FENCE() ((void)0)
#else
// nothing may be scheduled across this point (keeps the masked LDS stores of the prefetched chunk, and therefore
// their vmcnt wait, BEHIND the MFMA block that hides the load latency)
#define HIFIHR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define HIFIHR_KEEP(x) asm volatile("" : "+v"(x))
#endif

// (the K-chunk depth is a template parameter of both kernels: BK for the implicit GEMM, BKW for the weight gradient)

// Balanced ("stream-K") schedule.  The dispatcher spreads a grid evenly over the 256 CUs and 4 of these workgroups are
// resident per CU, so a launch of T tiles costs ceil(T / 1024) full rounds: 784 tiles (layer 4 at batch 32) take as long as
// 1024 (tools/conv_quant_probe.py).  With SK = true the grid is exactly CUs x 4 persistent workgroups and each one walks an
// equal share of the T x nch (tile, K-chunk) iterations.  A tile whose chunks are shared by several workgroups is summed in
// a zero-initialised, SELF-CLEANING fp32 workspace with atomics (accumulator layout = the adders' own register layout, so
// every wave instruction adds 256 contiguous bytes); a per-tile arrival counter elects the LAST workgroup to arrive, which
// swaps the sums back out (atomicExch leaves zeros behind) and runs the normal epilogue -- no workgroup ever waits for
// another one, and all cross-workgroup traffic is device-scope read-modify-write at the memory side (no stale-cache window).
// XCD locality: workgroup b runs on XCD b % 8 (each XCD has its own 4 MB L2), so the persistent workgroups are renumbered to
// give every XCD ONE contiguous eighth of the iteration space, and tiles are ordered column-tile fastest: an XCD then works
// on a band of pixel rows (its slice of the activations stays in its L2 across the 9 taps and all column tiles) while the
// column tiles that run side by side stream the same weight slab.
// In-phase variant (mode 1, used when tiles % 8 == 0 and tiles <= workgroups): per XCD, workgroup `slot` < Tx owns chunks
// [0, S) of tile `slot` of the XCD's band -- all of them walk the K loop IN STEP, so the column tiles of a pixel band read
// the same activation chunk and the row tiles the same weight chunk within the L2 residency window -- and the remaining
// Gx - Tx workgroups share the tails [S, nch) of the band's tiles.  The plain stream-K split (mode 0) staggers every
// workgroup's phase; PMC showed an 11 % L2 hit rate for it (TCC_HIT / TCC_REQ, layer 4) against 76 % in phase, and 4.5x the
// fabric reads -- which bought only 1.5 % of time: the kernel is not bound by L2 misses.
struct SkArgs {
  int tiles_x, tiles_y, nch, per, total;   // row / column tiles, K chunks per tile, iterations per workgroup, tiles * nch
  int mode, Tx, S, tper;                   // mode 1: tiles per XCD, chunk split point, tail iterations per tail workgroup
  int tiles_pb;                            // tiles per batch entry (tiles_x * tiles_y); tile / tiles_pb = batch index
  float* ws;                      // [tiles][BM * BN]
  unsigned* cnt;                  // [tiles] arrival counters (zero between launches)
};

// Ablation builds for tools/conv_ablate.py only (results are wrong for any value but 0): 1 = no global loads in the main
// loop, 2 = also no LDS stores, 3 = also no barrier, 4 = also no LDS reads (MFMA + epilogue only).
#ifndef HIFIHR_CONV_PROBE
#define HIFIHR_CONV_PROBE 0
#endif
// HIFIHR_CONV_STAMP (diagnostic build, tools/build_conv_probes.sh): per-wave s_memtime stamps around the four phases of a
// K chunk, summed into g_conv_stamp[0..3] = cycles in (load issue, LDS reads + MFMA block, vmcnt wait + LDS stores, barrier),
// [4] = chunks.  The stamps sit where no LDS operation is outstanding, so their lgkmcnt wait perturbs little.
#if defined(HIFIHR_CONV_STAMP)
__device__ unsigned long long g_conv_stamp[8];
#define HIFIHR_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define HIFIHR_STAMP(var) ((void)0)
#endif

// How one launch (or one parity class of a strided dgrad launch) walks rows and taps.
//   forward : rows = output pixels, tap j reads src row  oy*stride - pad + j
//   dgrad   : rows = the input pixels of parity class (ph, pw) = (oy*st + ph, ox*st + pw); only taps
//             r = r0 + st*j with r0 = (ph + pad) % st can contribute, and they read dy row  oy + aof - j  with
//             no divisibility test left in the loop (a stride-2 3x3 dgrad does 1/4 of the naive MFMA work).
struct GatherPlan {
  int OHs, OWs;            // row sub-grid
  int dmul, dofh, dofw;    // dst pixel = (oy*dmul + dofh, ox*dmul + dofw)
  int amul, aofh, aofw;    // src base  = oy*amul + aofh
  int nr, ns, r0, s0, rstep, sign;
};

__device__ __forceinline__ GatherPlan make_plan(const ConvGeom& g, int cls) {
  GatherPlan p;
  if (!g.dgrad) {
    p.OHs = g.OH; p.OWs = g.OW; p.dmul = 1; p.dofh = 0; p.dofw = 0; p.amul = g.stride; p.aofh = -g.pad; p.aofw = -g.pad;
    p.nr = g.R; p.ns = g.S; p.r0 = 0; p.s0 = 0; p.rstep = 1; p.sign = 1;
  } else {
    const int st = g.stride, ph = cls / st, pw = cls % st;
    p.OHs = (g.OH - ph + st - 1) / st; p.OWs = (g.OW - pw + st - 1) / st;
    if (p.OHs < 0) p.OHs = 0;
    if (p.OWs < 0) p.OWs = 0;
    p.dmul = st; p.dofh = ph; p.dofw = pw; p.amul = 1;
    p.r0 = (ph + g.pad) % st; p.s0 = (pw + g.pad) % st;
    p.nr = (p.r0 < g.R) ? (g.R - p.r0 + st - 1) / st : 0;
    p.ns = (p.s0 < g.S) ? (g.S - p.s0 + st - 1) / st : 0;
    p.aofh = (ph + g.pad - p.r0) / st; p.aofw = (pw + g.pad - p.s0) / st;
    p.rstep = st; p.sign = -1;
  }
  return p;
}

// GENERIC = false requires IC % BK == 0 (a BK-deep K chunk never straddles a tap: tap bookkeeping is scalar and
// division-free).  BK = 16 or 32: depth of one K chunk = BK/2 MFMA k-steps between two barriers.  GENERIC = true (forward only) handles any IC % 4 == 0 with per-chunk divisions (the 4-channel stem).
// waves_per_eu: the 64x64 kernels must stay at <= 128 VGPRs so that 4 workgroups share a CU (the balanced schedule launches
// exactly CUs x 4 persistent workgroups; at 136 registers only 3 fit and the fourth quarter of the grid runs as a second round)
#if defined(HIFIHR_HOSTSIM)
#define HIFIHR_WAVES_PER_EU(n)
#define HIFIHR_WAIT_VMEM() ((void)0)
#else
#if defined(HIFIHR_CONV_NO_CAP)       /* tuning builds (tools/build_conv_probes.sh) */
#define HIFIHR_WAVES_PER_EU(n)
#else
#define HIFIHR_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif
// Wait until this wave's outstanding vector-memory operations (here: its device-scope atomic adds, which execute at the
// memory side) have been acknowledged.  NOT __threadfence(): an agent-scope fence also writes back and invalidates this
// XCD's L2 (buffer_wbl2 / buffer_inv sc1), which nothing here needs -- the hand-off below consists of atomics only -- and
// which cost ~200 us per launch when every pass of every workgroup did it.
#define HIFIHR_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
template <int BM, int BN, bool GENERIC, int BK, bool SK>
__global__ __launch_bounds__(256) HIFIHR_WAVES_PER_EU((SK && BM == 64) ? 4 : 1) void conv_igemm_kernel(ConvGeom g, const float* __restrict__ src0,
                                                        const float* __restrict__ wgt0, const float* __restrict__ bias,
                                                        float* __restrict__ dst0, float* __restrict__ stats, SkArgs sk) {
  // stats (optional, forward only): [kStatSlots][2][OC] per-channel sum and sum of squares of the output, accumulated with
  // atomics from the accumulator registers -- the batch-norm that follows needs no separate pass over y
  constexpr int TM = BM / 64, TN = BN / 64;      // 32x32 MFMA tiles per wave (waves are arranged 2 x 2)
  constexpr int LD = BK + 4;                     // LDS row stride: 80 B / 144 B keep ds_read_b128 conflict-free
  constexpr int SEGS = BK / 4;                   // float4 segments per row
  constexpr int RP = 256 / SEGS;                 // rows covered by one load pass of the workgroup
  constexpr int AL = BM / RP, BL = BN / RP;      // load passes per chunk
  constexpr int HK = BK / 2;                     // k-steps per chunk = floats per lane-half per row
  __shared__ __attribute__((aligned(16))) float As[2][BM * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, r31 = lane & 31;
  const int cls = (SK || g.batch > 1) ? 0 : (int)blockIdx.z;
  const GatherPlan P = make_plan(g, cls);
  // the second convolution's tap (ConvGeom::src2): class (0, 0) of a strided backward-data launch only (never SK, never GENERIC, batch 1)
  const bool extra = !SK && !GENERIC && g.src2 != nullptr && g.dgrad && cls == 0;
  const int M = g.N * P.OHs * P.OWs;
  const int Qw = g.R * g.S * g.IC;                // row length of the weight matrix
  const int lrow = tid / SEGS, seg = (tid % SEGS) * 4;
  const int nch_tile = GENERIC ? (Qw + BK - 1) / BK : P.nr * P.ns * (g.IC / BK) + (extra ? g.IC / BK : 0);
  __shared__ int sk_last;
  int wg = blockIdx.x;
  int it = 0, it_end = 0, tile_base = 0, L = 0;
  bool tail_wg = false;
  if (SK) {
    if (sk.mode == 1) {
      const int xcd = wg & 7, slot = wg >> 3;
      tile_base = xcd * sk.Tx;
      L = nch_tile - sk.S;
      tail_wg = slot >= sk.Tx;
      if (!tail_wg) { it = slot * nch_tile; it_end = it + sk.S; }                    // chunks [0, S) of tile tile_base + slot
      else { it = (slot - sk.Tx) * sk.tper; it_end = min(it + sk.tper, sk.Tx * L); }  // range of the band's tail space
    } else {
      if ((gridDim.x & 7) == 0) wg = (wg & 7) * (gridDim.x >> 3) + (wg >> 3);       // XCD b % 8 -> contiguous eighth
      it = wg * sk.per;
      it_end = min(it + sk.per, sk.total);
    }
  }
  do {                                            // SK: one pass per (tile, chunk range) of this workgroup's share
  int tile = 0, c_begin = 0, c_end = nch_tile, tx = blockIdx.x, ty = blockIdx.y, bz = (!SK && g.batch > 1) ? (int)blockIdx.z : 0;
  if (SK) {
    if (it >= it_end) break;
    if (sk.mode == 1 && tail_wg) {
      const int tl = it / L;
      tile = tile_base + tl;
      c_begin = sk.S + (it - tl * L);
      c_end = min(nch_tile, c_begin + (it_end - it));
    } else {
      tile = it / nch_tile;
      c_begin = it - tile * nch_tile;
      c_end = min(sk.mode == 1 ? sk.S : nch_tile, c_begin + (it_end - it));
      if (sk.mode == 1) tile += tile_base;
    }
    bz = tile / sk.tiles_pb;
    const int tin = tile - bz * sk.tiles_pb;
    tx = tin / sk.tiles_y; ty = tin - tx * sk.tiles_y;
    it += c_end - c_begin;
  }
  // batched use: this pass's problem
  const float* __restrict__ src = src0 + (size_t)bz * g.src_bs;
  const float* __restrict__ wgt = wgt0 + (size_t)bz * g.wgt_bs;
  float* __restrict__ dst = dst0 + (size_t)bz * g.dst_bs;
  const int bm0 = tx * BM, bn0 = ty * BN;
  if (!SK && bm0 >= M) return;                    // parity classes can be smaller than the launch grid

  // per-thread row bookkeeping for the gather (constant over the K loop)
  size_t a_base[AL];
  int a_h[AL], a_w[AL];
  bool a_ok[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int m = bm0 + lrow + RP * i;
    a_ok[i] = m < M;
    const int mm = a_ok[i] ? m : 0;
    const int n = mm / (P.OHs * P.OWs);
    const int rem = mm - n * (P.OHs * P.OWs);
    const int oy = rem / P.OWs, ox = rem - oy * P.OWs;
    a_base[i] = (size_t)n * g.IH * g.IW * g.IC;
    a_h[i] = oy * P.amul + P.aofh;
    a_w[i] = ox * P.amul + P.aofw;
  }

  // Non-GENERIC fast path: everything that depends on the row is folded into a 32-bit element offset of the (tap 0, 0)
  // position and a bit mask of the taps that fall inside the image, so a load costs one add, one bit test and one select
  // in the K loop (the loop was VALU-bound on address arithmetic: 75 vector instructions per 16 MFMAs in round 1).
  // Rows past M and output channels past OC read valid memory and produce values nobody stores; only padding taps must
  // contribute exact zeros, which the mask takes care of at LDS-store time.
  int a_off[AL], w_off[BL], w_off2[BL];
  unsigned long long a_mask[AL];
  if (!GENERIC) {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      a_off[i] = (int)((long)a_base[i] + ((long)a_h[i] * g.IW + a_w[i]) * g.IC) + seg;
      unsigned rbits = 0;
      unsigned long long cbits = 0, m = 0;
      for (int r = 0; r < P.nr; ++r) { const int ih = a_h[i] + P.sign * r; if (ih >= 0 && ih < g.IH) rbits |= 1u << r; }
      for (int c = 0; c < P.ns; ++c) { const int iw = a_w[i] + P.sign * c; if (iw >= 0 && iw < g.IW) cbits |= 1ull << c; }
      for (int r = 0; r < P.nr; ++r) if ((rbits >> r) & 1u) m |= cbits << (r * P.ns);
      a_mask[i] = a_ok[i] ? ((m & 0x3fffffffffffffffull) | (1ull << 62)) : 0ull;   // bit 63 stays clear: the "no tap" bit of prefetches past the
    }                                                                             // end; bit 62 = the row exists: the second convolution's tap
#pragma unroll
    for (int j = 0; j < BL; ++j) {
      const int k = bn0 + lrow + RP * j;
      w_off[j] = (k < g.OC ? k : 0) * Qw + seg;
      w_off2[j] = (k < g.OC ? k : 0) * g.IC + seg;          // (the second convolution's filter: rows of IC floats)
    }
  }

  const int nch = c_end - c_begin;                // chunks of this pass
  int jr = 0, js = 0, c0 = 0, qgen = 0, issued = 0;   // tap state of the NEXT chunk to load
  if (SK && c_begin > 0) {                        // (SK is never GENERIC) chunk -> (tap row, tap column, channel block)
    const int cb = g.IC / BK, t2 = c_begin / cb;
    c0 = (c_begin - t2 * cb) * BK;
    jr = t2 / P.ns; js = t2 - jr * P.ns;
  }

  // kPF register stages: the chunk consumed now was loaded kPF - 1 chunk-computations ago, so ~2 MFMA blocks (plus the
  // other resident workgroups) cover the L2/HBM latency even when only 2-3 workgroups fit the grid per CU.
  constexpr int kPF = 3;
  float4 ra[kPF][AL], rb[kPF][BL];
  bool va[kPF][AL], vb[kPF][BL];
  auto load_global = [&](const int st) {
    // Loads are UNCONDITIONAL (out-of-range taps read a clamped, valid address) and masked when they are written to
    // LDS after the MFMA block: a load inside a branch makes the compiler wait for it right there, which exposes
    // the full memory latency in every K chunk.
    if (GENERIC) {
      const int q = qgen + seg;
      const bool qok = q < Qw;
      const int t = qok ? q / g.IC : 0;
      const int c = qok ? q - t * g.IC : 0;
      const int dr = t / g.S, ds = t - dr * g.S;
      qgen += BK;
#pragma unroll
      for (int i = 0; i < AL; ++i) {
        const int ih = a_h[i] + P.sign * dr, iw = a_w[i] + P.sign * ds;
        const bool ok = qok && a_ok[i] && ih >= 0 && ih < g.IH && iw >= 0 && iw < g.IW;
        const size_t off = ok ? a_base[i] + ((size_t)ih * g.IW + iw) * g.IC + c : 0;
        ra[st][i] = *reinterpret_cast<const float4*>(src + off);
        va[st][i] = ok;
      }
#pragma unroll
      for (int j = 0; j < BL; ++j) {
        const int k = bn0 + lrow + RP * j;
        const bool ok = qok && k < g.OC;
        const size_t off = ok ? (size_t)k * Qw + q : 0;
        rb[st][j] = *reinterpret_cast<const float4*>(wgt + off);
        vb[st][j] = ok;
      }
    } else {
      const bool qok = issued < nch;         // false for the prefetches issued past the last chunk (their data is never used;
      ++issued;                              // a parity class can also have nr > 0 but ns == 0: no chunk at all)
      // (extra: behind the class's own taps, jr == P.nr, come the IC / BK chunks of the second convolution: its dy at the class pixel itself,
      //  its filter row of IC floats)
      const bool ex = extra && qok && jr >= P.nr;                                                // wave-uniform
      const int tbit = !qok ? 63 : ex ? 62 : jr * P.ns + js;                                     // wave-uniform (SALU)
      const int toff = !qok ? 0 : ex ? c0 - (P.aofh * g.IW + P.aofw) * g.IC : P.sign * (jr * g.IW + js) * g.IC + c0;
      const int wq = !qok ? 0 : ex ? c0 : ((P.r0 + P.rstep * jr) * g.S + (P.s0 + P.rstep * js)) * g.IC + c0;
      const float* __restrict__ sp = ex ? g.src2 : src;
      const float* __restrict__ wp = ex ? g.wgt2 : wgt;
      c0 += BK;
      if (c0 >= g.IC) { c0 = 0; if (++js == P.ns || ex) { js = 0; ++jr; } }
#pragma unroll
      for (int i = 0; i < AL; ++i) {
        const bool ok = ((a_mask[i] >> tbit) & 1ull) != 0;
        const unsigned off = ok ? (unsigned)(a_off[i] + toff) : 0u;
        ra[st][i] = *reinterpret_cast<const float4*>(sp + off);
        va[st][i] = ok;
      }
#pragma unroll
      for (int j = 0; j < BL; ++j) {
        rb[st][j] = *reinterpret_cast<const float4*>(wp + (unsigned)((ex ? w_off2[j] : w_off[j]) + wq));
        vb[st][j] = true;
      }
    }
  };
  auto store_lds = [&](const int st, int buf) {
    // k-interleave: even columns of the chunk -> floats [0,HK), odd columns -> [HK,BK) of the row
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      float* p = &As[buf][(lrow + RP * i) * LD];
      const float4 v = va[st][i] ? ra[st][i] : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float2*>(p + seg / 2) = make_float2(v.x, v.z);
      *reinterpret_cast<float2*>(p + HK + seg / 2) = make_float2(v.y, v.w);
    }
#pragma unroll
    for (int j = 0; j < BL; ++j) {
      float* p = &Bs[buf][(lrow + RP * j) * LD];
      const float4 v = (!GENERIC || vb[st][j]) ? rb[st][j] : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float2*>(p + seg / 2) = make_float2(v.x, v.z);
      *reinterpret_cast<float2*>(p + HK + seg / 2) = make_float2(v.y, v.w);
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#if defined(HIFIHR_CONV_STAMP)
  unsigned long long stamp_acc[5] = {0, 0, 0, 0, 0};
#endif
  // prologue: chunks 0, 1, 2 in flight (masked past the end), chunk 0 to LDS
  load_global(0);
  load_global(1);
  load_global(2);
  store_lds(0, 0);
  __syncthreads();
  for (int ch0 = 0; ch0 < nch; ch0 += kPF) {
#pragma unroll
    for (int u = 0; u < kPF; ++u) {
      const int ch = ch0 + u;
      if (ch >= nch) break;
      const int buf = ch & 1;
      // stage u held chunk ch (already in LDS): refill it with chunk ch + 3.  Issued unconditionally (masked, clamped
      // loads past the last tap): a conditional load turns the registers into phis whose copies wait at once.
      HIFIHR_STAMP(t0);
      if (HIFIHR_CONV_PROBE < 1) load_global(u);
      HIFIHR_SCHED_FENCE();               // loads first, then the MFMA block
      HIFIHR_STAMP(t1);
      float a[TM][HK], b[TN][HK];
#if HIFIHR_CONV_PROBE >= 4
#pragma unroll
      for (int t = 0; t < HK; ++t) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i][t] = ra[0][0].x + (float)(t + ch);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j][t] = rb[0][0].x;
      }
#else
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float* p = &As[buf][(wm * (BM / 2) + i * 32 + r31) * LD + half * HK];
#pragma unroll
        for (int v4 = 0; v4 < HK / 4; ++v4) {
          const float4 q = *reinterpret_cast<const float4*>(p + 4 * v4);
          a[i][4 * v4] = q.x; a[i][4 * v4 + 1] = q.y; a[i][4 * v4 + 2] = q.z; a[i][4 * v4 + 3] = q.w;
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float* p = &Bs[buf][(wn * (BN / 2) + j * 32 + r31) * LD + half * HK];
#pragma unroll
        for (int v4 = 0; v4 < HK / 4; ++v4) {
          const float4 q = *reinterpret_cast<const float4*>(p + 4 * v4);
          b[j][4 * v4] = q.x; b[j][4 * v4 + 1] = q.y; b[j][4 * v4 + 2] = q.z; b[j][4 * v4 + 3] = q.w;
        }
      }
#endif
#pragma unroll
      for (int t = 0; t < HK; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
      HIFIHR_SCHED_FENCE();
#if defined(HIFIHR_CONV_STAMP)
      HIFIHR_KEEP(acc[0][0][0]);          // the stamp must not be taken before the last MFMA has delivered
#endif
      HIFIHR_STAMP(t2);
      if (HIFIHR_CONV_PROBE < 2) store_lds((u + 1) % kPF, buf ^ 1);  // chunk ch + 1 (loaded two chunk-computations ago) -> the other LDS buffer
#if defined(HIFIHR_CONV_STAMP)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      HIFIHR_STAMP(t3);
      if (HIFIHR_CONV_PROBE < 3 && HIFIHR_CONV_PROBE != -1) __syncthreads();     // -1: everything but the barrier
#if defined(HIFIHR_CONV_STAMP)
      {
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        stamp_acc[0] += t1 - t0; stamp_acc[1] += t2 - t1; stamp_acc[2] += t3 - t2; stamp_acc[3] += t4 - t3; stamp_acc[4] += 1ull;
      }
#endif
    }
  }

#if defined(HIFIHR_CONV_STAMP)
  if (lane == 0)
    for (int k = 0; k < 5; ++k) atomicAdd(&g_conv_stamp[k], stamp_acc[k]);
#endif
  if (SK && nch != nch_tile) {
    // This pass covered part of the tile's K range: add the partial sums to the tile's workspace, then count arrivals.
    int nseg;                                     // workgroups that share this tile
    if (sk.mode == 1) {
      const int tl = tile - tile_base;
      nseg = 1 + ((tl + 1) * L - 1) / sk.tper - (tl * L) / sk.tper + 1;
    } else {
      nseg = ((tile + 1) * nch_tile - 1) / sk.per - (tile * nch_tile) / sk.per + 1;
    }
    float* wt = sk.ws + (size_t)tile * (BM * BN) + wave * 64 + lane;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) atomicAdd(wt + ((i * TN + j) * 16 + e) * 256, acc[i][j][e]);
    HIFIHR_WAIT_VMEM();                           // every add of this workgroup is performed before it is counted
    __syncthreads();
    if (tid == 0) sk_last = (atomicAdd(sk.cnt + tile, 1u) == (unsigned)(nseg - 1)) ? 1 : 0;
    __syncthreads();
    const bool last = sk_last != 0;
    __syncthreads();                              // sk_last may be rewritten by the next pass
    if (!last) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = atomicExch(wt + ((i * TN + j) * 16 + e) * 256, 0.f);   // read and clean
    if (tid == 0) sk.cnt[tile] = 0u;              // nobody else touches this counter in this launch any more
  }

  // epilogue: D[i][j], i = pixel row, j = output channel; lanes 0..31 hold 32 consecutive channels of one row, so
  // every store instruction writes two 128-byte row segments.  The bias is fetched ONCE up front: a load inside the
  // store loop makes the compiler wait vmcnt(0) per element, which also drains the previous store (16 serialised
  // round trips per tile).
  float bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int k = bn0 + wn * (BN / 2) + j * 32 + r31;
    bv[j] = (bias != nullptr && k < g.OC) ? bias[k] : 0.f;
  }
  if (P.dmul == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int k = bn0 + wn * (BN / 2) + j * 32 + r31;
        float* col = dst + k;
        if (g.residual != nullptr) {        // uniform (backward-data + the other consumer's gradient): all 16 loads first, one wait
          float rv[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int m = bm0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            rv[e] = (m < M && k < g.OC) ? g.residual[(size_t)bz * g.dst_bs + (size_t)m * g.OC + k] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += rv[e];
        }
        if (bias != nullptr) {              // uniform; the bias-free path below contains no load at all
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            acc[i][j][e] += bv[j];
            HIFIHR_KEEP(acc[i][j][e]);      // keep the add (and its one vmcnt wait) out of the per-element branches
          }
        }
        if (g.relu) {                       // uniform (LightEstimator's conv + bias + ReLU in one launch)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaxf(acc[i][j][e], 0.f);
        }
        // batch-norm statistics: sums of d = y - ks, ks = the lane's first row of this column (hifihr_internal.h "FORWARD statistics")
        const float ks = acc[i][j][0];
        float ssum = 0.f, ssq = 0.f;
        int sn = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = bm0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
          if (m < M && k < g.OC) {
            col[(size_t)m * g.OC] = acc[i][j][e];
            const float d = acc[i][j][e] - ks;
            ssum += d;
            ssq += d * d;
            ++sn;
          }
        }
        if (stats != nullptr) {                  // uniform
          double S1, S2;
          stat_unshift(sn, ks, ssum, ssq, S1, S2);
          S1 += __shfl_down(S1, 32, 64);         // lanes l and l + 32 hold the same channel, different rows
          S2 += __shfl_down(S2, 32, 64);
          if (half == 0 && k < g.OC) {           // 32 consecutive channels
            double* sp = reinterpret_cast<double*>(stats) + (size_t)((tx * 2 + wm) & (stat_slots_used(g.OC) - 1)) * 2 * g.OC;   // slot by row tile
            stat_atomic_add(sp + k, S1);
            stat_atomic_add(sp + g.OC + k, S2);
          }
        }
      }
  } else {
    // parity class of a strided dgrad: scatter rows to the strided destination pixels
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      size_t rowoff[16];
      bool rok[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = bm0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        rok[e] = m < M;
        const int mm = rok[e] ? m : 0;
        const int n = mm / (P.OHs * P.OWs);
        const int rem = mm - n * (P.OHs * P.OWs);
        const int oy = rem / P.OWs, ox = rem - oy * P.OWs;
        rowoff[e] = (((size_t)n * g.OH + (oy * P.dmul + P.dofh)) * g.OW + (ox * P.dmul + P.dofw)) * g.OC;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int k = bn0 + wn * (BN / 2) + j * 32 + r31;
        if (g.residual != nullptr) {        // uniform: the parity classes write disjoint pixels, each adds its own pixels' residual
          float rv[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) rv[e] = (rok[e] && k < g.OC) ? g.residual[(size_t)bz * g.dst_bs + rowoff[e] + k] : 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += rv[e];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (rok[e] && k < g.OC) dst[rowoff[e] + k] = acc[i][j][e];
      }
    }
  }
  } while (SK);
}

// ------------------------------------------------------------------------------------------------
// backward weight
// ------------------------------------------------------------------------------------------------
// dy2 / dw2 / qtiles1 (round 6; null / 0: off): the weight gradient of a SECOND convolution of the same input -- 1x1, the same stride, pad 0, the
// same output channels (a residual stage's downsample branch) -- in the same launch: its patch column is tap (pad, pad) of this one, so the
// column tiles blockIdx.x >= qtiles1 run dW2[k][c] = sum_m dy2[m][k] x[m, tap (pad, pad)][c] with everything else unchanged (a 17 us launch
// of 0.4 GFLOP otherwise)
template <int BM, int BN, int BKW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(ConvGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ dw, int chunks_per_split, const float* __restrict__ dy2,
                                                        float* __restrict__ dw2, int qtiles1) {
  // GEMM: dW[k][q] (BM x BN tile) = sum over pixels m of dy[m][k] * patch[m][q];  g describes the FORWARD conv
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int LA = BM + 4, LB = BN + 4;
  __shared__ __attribute__((aligned(16))) float As[2][BKW * LA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BKW * LB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, r31 = lane & 31;
  const int M = g.N * g.OH * g.OW;
  const bool second = dy2 != nullptr && (int)blockIdx.x >= qtiles1;      // (uniform) a column tile of the second convolution
  if (second) { dy = dy2; dw = dw2; }
  const int Q = second ? g.IC : g.R * g.S * g.IC;                         // row length of this tile's dW
  const int K = g.OC;
  const int bq0 = (second ? (int)blockIdx.x - qtiles1 : (int)blockIdx.x) * BN, bk0 = blockIdx.y * BM;
  const int nch_total = (M + BKW - 1) / BKW;
  int zsplit = blockIdx.z;
  if (g.batch > 1) {                               // batched use (Winograd weight gradient): blockIdx.z = batch * splits + split
    const int nsplit = gridDim.z / g.batch, bz = blockIdx.z / nsplit;
    zsplit = blockIdx.z - bz * nsplit;
    x += (size_t)bz * g.src_bs; dy += (size_t)bz * g.dst_bs; dw += (size_t)bz * g.wgt_bs;
  }
  const int ch_lo = zsplit * chunks_per_split;
  const int ch_hi = min(ch_lo + chunks_per_split, nch_total);
  if (ch_lo >= ch_hi) return;

  // load mapping: a row of the tile is one pixel m; BM/4 (BN/4) float4 per row
  constexpr int APR = BM / 4, BPR = BN / 4;            // float4 per row
  constexpr int AROWS = 256 / APR, BROWS = 256 / BPR;  // rows covered per pass
  constexpr int AL = BKW / AROWS, BL = BKW / BROWS;    // passes
  const int a_row = tid / APR, a_col = (tid % APR) * 4;
  const int b_row = tid / BPR, b_col = (tid % BPR) * 4;
  // the q coordinates of this thread's patch column are constant over the loop
  const int q = bq0 + b_col;
  const bool qok = q < Q;
  int qr = 0, qs = 0, qc = 0;
  if (qok && second) {
    qc = q; qr = g.pad; qs = g.pad;                  // x[oh stride, ow stride]: this convolution's tap (pad, pad)
  } else if (qok) {
    const int t = q / g.IC;
    qc = q - t * g.IC;
    qr = t / g.S;
    qs = t - qr * g.S;
  }
  float4 ra[AL], rb[BL];
  bool va[AL], vb[BL];
  // unconditional, clamped loads + masking at LDS-store time (see conv_igemm_kernel)
  auto load_global = [&](int ch) {
    const int m0 = ch * BKW;
    const bool chok = ch < ch_hi;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int m = m0 + a_row + AROWS * i;
      const int k = bk0 + a_col;
      const bool ok = chok && m < M && k < K;
      ra[i] = *reinterpret_cast<const float4*>(dy + (ok ? (size_t)m * K + k : 0));
      va[i] = ok;
    }
#pragma unroll
    for (int i = 0; i < BL; ++i) {
      const int m = m0 + b_row + BROWS * i;
      const int mm = (m < M) ? m : 0;
      const int n = mm / (g.OH * g.OW);
      const int rem = mm - n * (g.OH * g.OW);
      const int oh = rem / g.OW, ow = rem - oh * g.OW;
      const int ih = oh * g.stride - g.pad + qr, iw = ow * g.stride - g.pad + qs;
      const bool ok = chok && m < M && qok && ih >= 0 && ih < g.IH && iw >= 0 && iw < g.IW;
      rb[i] = *reinterpret_cast<const float4*>(x + (ok ? (((size_t)n * g.IH + ih) * g.IW + iw) * g.IC + qc : 0));
      vb[i] = ok;
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i)
      *reinterpret_cast<float4*>(&As[buf][(a_row + AROWS * i) * LA + a_col]) = va[i] ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < BL; ++i)
      *reinterpret_cast<float4*>(&Bs[buf][(b_row + BROWS * i) * LB + b_col]) = vb[i] ? rb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_global(ch_lo);
  store_lds(0);
  __syncthreads();
  for (int ch = ch_lo; ch < ch_hi; ++ch) {
    const int buf = (ch - ch_lo) & 1;
    load_global(ch + 1);                  // unconditional prefetch (masked past the end of this split)
    HIFIHR_SCHED_FENCE();
#pragma unroll
    for (int t = 0; t < BKW / 2; ++t) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[buf][(2 * t + half) * LA + wm * (BM / 2) + i * 32 + r31];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[buf][(2 * t + half) * LB + wn * (BN / 2) + j * 32 + r31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    HIFIHR_SCHED_FENCE();
    store_lds(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int qq = bq0 + wn * (BN / 2) + j * 32 + r31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = bk0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (k < K && qq < Q) atomicAdd(dw + (size_t)k * Q + qq, acc[i][j][e]);
      }
    }
}

// [K][RS][C] -> [C][RS][K]
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int RS, int C) {
  const size_t n = (size_t)K * RS * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % K);
    const size_t rest = i / K;
    const int rs = (int)(rest % RS);
    const int c = (int)(rest / RS);
    wt[i] = w[((size_t)k * RS + rs) * C + c];
  }
}

// images NCHW [B][3][H][W] in [0,1] -> NHWC4 [B][OH][OW][4]; output pixel (oy, ox) = input pixel (oy - pt, ox - pl) or zero
// (explicit zero border: the EfficientNet stem's asymmetric "same" padding); normalize: ((x - mean) / std, 0)
__global__ __launch_bounds__(256) void image_to_nhwc4_kernel(const float* __restrict__ img, float4* __restrict__ out, int B, int H, int W,
                                                            int OH, int OW, int pt, int pl, int normalize) {
  const size_t n = (size_t)B * OH * OW, HW = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t b = i / ((size_t)OH * OW), p = i - b * OH * OW;
    const int oy = (int)(p / OW), ox = (int)(p - (size_t)oy * OW);
    const int iy = oy - pt, ix = ox - pl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      const float* s = img + b * 3 * HW + (size_t)iy * W + ix;
      v = normalize ? make_float4((s[0] - 0.485f) / 0.229f, (s[HW] - 0.456f) / 0.224f, (s[2 * HW] - 0.406f) / 0.225f, 0.f)
                    : make_float4(s[0], s[HW], s[2 * HW], 0.f);
    }
    out[i] = v;
  }
}
// gradient of the above is never needed (images carry no gradient)

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int pick_tile(long M, int OC, bool generic) {
  // 0: 128x128, 1: 128x64, 2: 64x64.  Measured on MI355X at B = 32 (tools/time_conv.py, round 1): the 64x64 tile wins
  // on every ResNet-18 layer (7 resident workgroups per CU hide the per-chunk barrier), except the 4-channel stem.
  if (const char* e = getenv("HIFIHR_CONV_TILE")) return atoi(e);     // tuning/diagnostic override
  (void)M; (void)OC;
  return generic ? 1 : 2;
}

template <int BM, int BN>
static void launch_igemm_tile(const ConvGeom& g, long Mmax, int classes, bool generic, int bk, const float* src, const float* wgt,
                              const float* bias, float* dst, float* stats, hipStream_t st) {
  const dim3 grid((unsigned)((Mmax + BM - 1) / BM), (g.OC + BN - 1) / BN, g.batch > 1 ? g.batch : classes);
  const SkArgs none{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, nullptr, nullptr};
  if (generic)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, true, 16, false>), grid, dim3(256), 0, st, g, src, wgt, bias, dst, stats, none);
  else if (bk == 32)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, false, 32, false>), grid, dim3(256), 0, st, g, src, wgt, bias, dst, stats, none);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, false, 16, false>), grid, dim3(256), 0, st, g, src, wgt, bias, dst, stats, none);
}

// ---- balanced schedule (one gather class, source channels % 32 == 0) ----
constexpr int kSkMinChunks = 16;         // K chunks per tile below which splitting is not worth a workspace round trip

static int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// variant of the persistent kernel: tile, K-chunk depth and the number of workgroups that are resident per CU
struct SkVariant {
  int bm, bn, bk, occ;
};
static SkVariant sk_variant() {
  // 0: 64x64 BK32 (36.9 KB LDS, <= 128 VGPRs: 4 per CU).  Others are tuning experiments (HIFIHR_CONV_SK_VARIANT).
  static const SkVariant v[] = {{64, 64, 32, 4}, {128, 64, 16, 3}, {128, 64, 32, 2}, {128, 128, 16, 2}};
  int i = 0;
  if (const char* e = getenv("HIFIHR_CONV_SK_VARIANT")) i = atoi(e);
  if (i < 0 || i > 3) i = 0;
  SkVariant r = v[i];
  if (const char* e = getenv("HIFIHR_CONV_SK_OCC")) r.occ = atoi(e) > 0 ? atoi(e) : r.occ;
  return r;
}

struct SkPlan {
  bool use;
  int tiles_x, tiles, nch, wgs, per;
  SkVariant v;
};

static SkPlan sk_plan(const ConvGeom& g) {
  SkPlan p{false, 0, 0, 0, 0, 0, sk_variant()};
  if (const char* e = getenv("HIFIHR_CONV_SK")) { if (atoi(e) == 0) return p; }
  const bool one_class = !g.dgrad || g.stride == 1;
  if (!one_class || g.IC % 32 != 0) return p;
  const long M = (long)g.N * g.OH * g.OW;
  p.tiles_x = (int)((M + p.v.bm - 1) / p.v.bm);
  p.tiles = p.tiles_x * ((g.OC + p.v.bn - 1) / p.v.bn) * (g.batch > 1 ? g.batch : 1);
  p.nch = g.R * g.S * (g.IC / p.v.bk);
  const int slots = device_cus() * p.v.occ;
  int minch = kSkMinChunks;
  if (const char* e = getenv("HIFIHR_CONV_SK_MINCH")) minch = atoi(e);
  // (fewer tiles than a quarter of the slots stay data-parallel: the balanced schedule for EfficientNet's 7 x 7 / 14 x 14 pointwise layers --
  //  37-150 tiles -- measured 34.04 vs 33.82 ms per config-3 step, HIFIHR_CONV_SK_TILEDIV=32: the workspace round trip costs more than the
  //  idle CUs; those products are bound by each workgroup streaming the whole weight panel for 16-64 rows)
  static const int min_tile_div = [] { const char* e = getenv("HIFIHR_CONV_SK_TILEDIV"); return e && atoi(e) > 0 ? atoi(e) : 4; }();
  if (p.nch * p.v.bk < minch * 32 || p.tiles < slots / min_tile_div) return p;
  // rounds the data-parallel grid costs vs the balanced share: only switch when > 5 % is on the table
  const double dp = (double)((p.tiles + slots - 1) / slots), sk = (double)p.tiles / slots;
  if (dp < 1.05 * sk && p.v.bm == 64) return p;
  const long total = (long)p.tiles * p.nch;
  p.per = (int)((total + slots - 1) / slots);
  p.wgs = (int)((total + p.per - 1) / p.per);
  p.use = true;
  return p;
}

size_t conv_sk_workspace_bytes(const ConvGeom& g) {
  const SkPlan p = sk_plan(g);
  if (!p.use) return 0;
  return (size_t)((p.tiles * sizeof(unsigned) + 255) / 256 * 256) + (size_t)p.tiles * p.v.bm * p.v.bn * sizeof(float);
}

template <int BM, int BN, int BK>
static void launch_sk(const SkPlan& p, const ConvGeom& g, const float* src, const float* wgt, float* dst, float* stats, void* sk_ws,
                      hipStream_t st) {
  const size_t cnt_bytes = (p.tiles * sizeof(unsigned) + 255) / 256 * 256;
  const int nb = g.batch > 1 ? g.batch : 1;
  SkArgs a{p.tiles_x, p.tiles / nb / p.tiles_x, p.nch, p.per, p.tiles * p.nch, 0, 0, 0, 0, p.tiles / nb,
           reinterpret_cast<float*>(static_cast<char*>(sk_ws) + cnt_bytes), static_cast<unsigned*>(sk_ws)};
  int wgs = p.wgs;
  const int slots = device_cus() * p.v.occ;
  // measured (tools/time_conv_sk.py): +1.5 % on the 144-chunk layer-4 shapes, -9 % on 36-chunk layer 2 (short tails): long K only
  bool inphase = p.tiles % 8 == 0 && slots % 8 == 0 && p.tiles <= slots && (p.nch >= 100 || device_cus() < 16);
  if (const char* e = getenv("HIFIHR_CONV_SK_INPHASE")) inphase = inphase && atoi(e) != 0;
  if (inphase) {
    const int Tx = p.tiles / 8, Gx = slots / 8;
    int S = (int)(((long)Tx * p.nch + Gx - 1) / Gx);
    if (S > p.nch) S = p.nch;
    const int L = p.nch - S;
    if (Tx == Gx || L == 0) { S = p.nch; }
    a.mode = 1; a.Tx = Tx; a.S = S;
    const int ntail = Gx - Tx;
    a.tper = (S < p.nch && ntail > 0) ? (int)(((long)Tx * (p.nch - S) + ntail - 1) / ntail) : 1;
    wgs = slots;
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, false, BK, true>), dim3(wgs), dim3(256), 0, st, g, src, wgt, nullptr, dst, stats, a);
}

// 3x3 / stride 1 / pad 1 onto FOUR output channels: the backward-data of a network's first layer on an NHWC4 image (VGG19 conv1_1 of the
// perceptual loss, 64 -> 3 (+1) channels on 224 x 224 or 512 x 512: the gradient that flows back into the renderer).  As an implicit GEMM
// its 4 output channels fill 1/16 of a 64-wide tile (1.7 ms at B = 48, 2.9 ms at 16 x 512 x 512).  Here: thread = output pixel, the
// 4 x 9 x IC filter comes through the scalar cache (wave-uniform addresses), 16 VALU multiply-adds per 16-byte load of the source.
// Lanes: four per output pixel (lane & 3 owns a quarter of the input channels, a contiguous 64 / 16 * IC bytes of every tap -- a wave reads
// 16 pixels x IC channels as one contiguous run; one lane per pixel was bound by the cache-line rate of its 256-byte stride, 0.9 ms);
// the filter lives in LDS as [quarter][tap][output channel][IC / 4]; the four partial sums meet through two DPP exchanges.
template <int IC>
__global__ __launch_bounds__(256) void conv3x3_oc4_kernel(const float* __restrict__ src, const float* __restrict__ wgt, float* __restrict__ dst,
                                                         int N, int H, int W, int sign) {
  constexpr int QC = IC / 4;                       // channels per lane
  constexpr int QS = 36 * QC + 4;                  // floats per quarter in LDS (+4: the four quarters of one read land in different banks)
  constexpr int U = 4;                             // pixels per lane (p, p + 64, ...): every filter value read from LDS feeds U multiply-adds
  __shared__ __attribute__((aligned(16))) float wl[4 * QS];
  for (int i = threadIdx.x; i < 4 * 36 * QC; i += 256) {
    const int j = i % QC, c = (i / QC) & 3, tap = (i / (4 * QC)) % 9, q = i / (36 * QC);
    wl[q * QS + (tap * 4 + c) * QC + j] = wgt[((size_t)c * 9 + tap) * IC + 16 * (j >> 2) + 4 * q + (j & 3)];      // lane q: channels 16 jj + 4 q + e
  }
  __syncthreads();
  const int q = threadIdx.x & 3;
  const float* wq = wl + q * QS;
  // workgroup = 16 x 16 output pixels (the rows above / below are re-read from L1, not from L2: one contiguous run of pixels per workgroup
  // made every vertical tap an L2 read, 0.76 ms); wave w, team t = (lane >> 2): pixels (x0 + t, y0 + 4 w + u), u = 0 .. 3
  const int tx = W + 15 >> 4, ty = H + 15 >> 4;
  const long tiles = (long)N * ty * tx;
  const int t = (threadIdx.x & 63) >> 2, wv_ = threadIdx.x >> 6;
  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int txi = (int)(tile % tx);
    const long r2 = tile / tx;
    const int tyi = (int)(r2 % ty), n = (int)(r2 / ty);
    const int x = txi * 16 + t, yb = tyi * 16 + 4 * wv_;
    const bool xin = x < W;
    const int xc = xin ? x : W - 1;
    float acc[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.f;
#pragma unroll 1                                           // (unrolled, the compiler hoists all 9 x U x QC / 4 loads: 891 spilled registers)
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s = tap - 3 * r;
      const int ix = sign > 0 ? xc + s - 1 : xc + 1 - s;
      const bool okx = ix >= 0 && ix < W;
      float4 v[U][QC / 4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int y = yb + u;
        const int iy = sign > 0 ? y + r - 1 : y + 1 - r;
        const bool ok = okx && iy >= 0 && iy < H;
        // the four lanes of a team read 64 contiguous bytes per load (half a cache line), four such runs per tap
        const float4* sp = reinterpret_cast<const float4*>(src + (((size_t)n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * IC) + q;
#pragma unroll
        for (int j = 0; j < QC / 4; ++j) {
          const float4 tv = sp[4 * j];
          v[u][j] = ok ? tv : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4* wp = reinterpret_cast<const float4*>(wq + (tap * 4 + c) * QC);
#pragma unroll
        for (int j = 0; j < QC / 4; ++j) {
          const float4 wv = wp[j];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            // fused multiply-adds (the file is built with -ffp-contract=off: a separate multiply and add would double the VALU work)
            acc[u][c] = __builtin_fmaf(v[u][j].x, wv.x, acc[u][c]); acc[u][c] = __builtin_fmaf(v[u][j].y, wv.y, acc[u][c]);
            acc[u][c] = __builtin_fmaf(v[u][j].z, wv.z, acc[u][c]); acc[u][c] = __builtin_fmaf(v[u][j].w, wv.w, acc[u][c]);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[u][c] += __shfl_xor(acc[u][c], 1, 64);
        acc[u][c] += __shfl_xor(acc[u][c], 2, 64);
      }
      const int y = yb + u;
      if (xin && y < H) dst[(((size_t)n * H + y) * W + x) * 4 + q] = q == 0 ? acc[u][0] : q == 1 ? acc[u][1] : q == 2 ? acc[u][2] : acc[u][3];
    }
  }
}

// The same convolution with the input HALO of an output tile staged in LDS (round 4).  The kernel above re-reads every input pixel for each
// of its nine taps, and a 16 x 16-pixel tile of 64 channels (65 KB) does not fit the 32 KB L1: the nine reads go to L2 -- 9 x 616 MB at
// batch 48, 682 us where the tensors' single pass is 131 us at 5 TB/s.  Here a workgroup stages the 18 x 10-pixel halo of a 16 x 8-pixel
// output tile once (57.6 KB at a pitch of 80 floats per pixel: the two teams of an 8-lane LDS access land in different banks; zeros
// outside the image), two workgroups per CU so that one stages while the other multiplies, and the taps read LDS.  Same lane roles (four
// lanes per output pixel, a quarter of the channels each), same filter image, same summation order per lane as the kernel above.
template <int IC>
__global__ __launch_bounds__(256) void conv3x3_oc4_tile_kernel(const float* __restrict__ src, const float* __restrict__ wgt, float* __restrict__ dst,
                                                              int N, int H, int W, int sign) {
  constexpr int QC = IC / 4, QS = 36 * QC + 4;
  constexpr int TW = 16, TH = 8, HP = TW + 2, HR = TH + 2, PS = IC + 16;      // tile, halo pitch (pixels), halo rows, floats per halo pixel
  constexpr int U = TH / 4;                                                  // output rows per lane (wave w: rows U w .. U w + U - 1)
  constexpr int SEG = IC / 4;                                                // float4 per pixel
  constexpr int NL = (HR * HP * SEG + 255) / 256;                            // staging loads per thread
  __shared__ __attribute__((aligned(16))) float wl[4 * QS];
  __shared__ __attribute__((aligned(16))) float hl[HR * HP * PS];
  for (int i = threadIdx.x; i < 4 * 36 * QC; i += 256) {
    const int j = i % QC, c = (i / QC) & 3, tap = (i / (4 * QC)) % 9, q = i / (36 * QC);
    wl[q * QS + (tap * 4 + c) * QC + j] = wgt[((size_t)c * 9 + tap) * IC + 16 * (j >> 2) + 4 * q + (j & 3)];
  }
  const int q = threadIdx.x & 3;
  const float* wq = wl + q * QS;
  const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
  const long tiles = (long)N * ty * tx;
  const int t = (threadIdx.x & 63) >> 2, wv_ = threadIdx.x >> 6;
  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int txi = (int)(tile % tx);
    const long r2 = tile / tx;
    const int tyi = (int)(r2 % ty), n = (int)(r2 / ty);
    const int x0 = txi * TW, y0 = tyi * TH;
    // ---- stage the halo: every load of the tile in flight, then the LDS stores ----
    float4 hv[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int e = threadIdx.x + 256 * k;
      const int px = e / SEG, seg = e - px * SEG;
      const int hy = px / HP, hx = px - hy * HP;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = e < HR * HP * SEG && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const float4 tv = *reinterpret_cast<const float4*>(src + (((size_t)n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * IC + seg * 4);
      hv[k] = ok ? tv : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();                                        // (the previous tile's taps are read; first trip: the filter image is written)
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int e = threadIdx.x + 256 * k;
      if (e < HR * HP * SEG) {
        const int px = e / SEG, seg = e - px * SEG;
        *reinterpret_cast<float4*>(&hl[px * PS + seg * 4]) = hv[k];
      }
    }
    __syncthreads();
    float acc[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.f;
#pragma unroll 3
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s = tap - 3 * r;
      const int hx = t + 1 + (sign > 0 ? s - 1 : 1 - s);
      float4 v[U][QC / 4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int hy = U * wv_ + u + 1 + (sign > 0 ? r - 1 : 1 - r);
        const float* hp = hl + (hy * HP + hx) * PS + 4 * q;
#pragma unroll
        for (int j = 0; j < QC / 4; ++j) v[u][j] = *reinterpret_cast<const float4*>(hp + 16 * j);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4* wp = reinterpret_cast<const float4*>(wq + (tap * 4 + c) * QC);
#pragma unroll
        for (int j = 0; j < QC / 4; ++j) {
          const float4 wv = wp[j];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            acc[u][c] = __builtin_fmaf(v[u][j].x, wv.x, acc[u][c]); acc[u][c] = __builtin_fmaf(v[u][j].y, wv.y, acc[u][c]);
            acc[u][c] = __builtin_fmaf(v[u][j].z, wv.z, acc[u][c]); acc[u][c] = __builtin_fmaf(v[u][j].w, wv.w, acc[u][c]);
          }
        }
      }
    }
    const int x = x0 + t;
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[u][c] += __shfl_xor(acc[u][c], 1, 64);
        acc[u][c] += __shfl_xor(acc[u][c], 2, 64);
      }
      const int y = y0 + U * wv_ + u;
      if (x < W && y < H) dst[(((size_t)n * H + y) * W + x) * 4 + q] = q == 0 ? acc[u][0] : q == 1 ? acc[u][1] : q == 2 ? acc[u][2] : acc[u][3];
    }
  }
}

static bool conv_oc4_supported(const ConvGeom& g, const float* bias, const float* stats) {
  static const int on = [] { const char* e = getenv("HIFIHR_CONV_OC4"); return e ? atoi(e) : 1; }();
  return on && g.OC == 4 && g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && (g.IC == 64 || g.IC == 32) && g.batch <= 1 && !g.relu &&
         bias == nullptr && stats == nullptr && g.IH == g.OH && g.IW == g.OW;
}

// A 1x1 convolution with stride 1 and no padding is a plain GEMM over the pixel rows (ResNet's stride-1 projection shortcut and every
// conv1 / conv3 of a bottleneck block): it runs on the kernels of csrc/gemm.hip, 2-3x faster there than as an implicit GEMM with one tap
// (tools/time_conv1x1.py).  HIFIHR_CONV1X1_GEMM=0 keeps the implicit-GEMM kernels (A/B timing).
bool conv_is_gemm(const ConvGeom& g) {
  static const int on = [] { const char* e = getenv("HIFIHR_CONV1X1_GEMM"); return e ? atoi(e) : 1; }();
  const long M = (long)g.N * g.OH * g.OW;
  return on && g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0 && g.batch <= 1 && g.OH == g.IH && g.OW == g.IW &&
         M * (g.IC > g.OC ? g.IC : g.OC) < (1L << 31) &&
         ((bgemm_nt_supported((int)M, g.OC, g.IC) && g.OC % 128 == 0) || bgemm_nt_ragged_supported((int)M, g.OC, g.IC));
}

// dw[i] += sum over the slabs, in a fixed order (bit-reproducible).  64 float4 per workgroup; wave w sums slabs 16 q + 4 w .. + 3 with the
// four loads of a trip in flight; the waves' partial sums meet in LDS (one thread per float4 walking every slab: 10 us for 28 slabs of 512 KB).
__global__ __launch_bounds__(256) void slab_sum_acc_kernel(const float* __restrict__ slabs, int nslab, long n4, float* __restrict__ dw) {
  __shared__ float4 part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  const bool live = i < n4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    for (int z0 = 4 * w; z0 < nslab; z0 += 16) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (z0 + u < nslab) ? reinterpret_cast<const float4*>(slabs)[(size_t)(z0 + u) * n4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && live) {
    float4 d = reinterpret_cast<float4*>(dw)[i];
    const float4 p0 = part[0][lane], p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
    d.x += (p0.x + p1.x) + (p2.x + p3.x); d.y += (p0.y + p1.y) + (p2.y + p3.y);
    d.z += (p0.z + p1.z) + (p2.z + p3.z); d.w += (p0.w + p1.w) + (p2.w + p3.w);
    reinterpret_cast<float4*>(dw)[i] = d;
  }
}

static bool conv_wgrad_is_gemm(const ConvGeom& g) {
  // a 64 x 256 filter is only four 64x64 tiles: 128 row slabs of it (71 us at 32 x 56 x 56) lose to conv_wgrad_kernel's atomics (55 us)
  return conv_is_gemm(g) && (long)g.OC * g.IC >= 32768 && bgemm_tn_supported(g.OC, g.IC, (int)((long)g.N * g.OH * g.OW));
}
static size_t conv_wgrad_gemm_bytes(const ConvGeom& g) {
  return (size_t)bgemm_tn_parts(g.OC, g.IC, (int)((long)g.N * g.OH * g.OW), 1) * g.OC * g.IC * sizeof(float);
}

hipError_t launch_conv_igemm(const ConvGeom& g, const float* src, const float* wgt, const float* bias, float* dst, float* stats,
                             void* sk_ws, size_t sk_ws_bytes, hipStream_t st) {
  if (stats != nullptr && g.dgrad) return hipErrorInvalidValue;   // stats: all zero on entry (self-cleaning, see bn.hip)
  if (g.IC % 4 != 0) return hipErrorInvalidValue;
  // the fast gather uses 32-bit element offsets (scaled by 4 in the address) and a 63-bit tap mask
  if ((long)g.N * g.IH * g.IW * g.IC >= (1L << 30) || (long)g.OC * g.R * g.S * g.IC >= (1L << 30)) return hipErrorInvalidValue;
  const bool res = g.residual != nullptr || g.src2 != nullptr;   // (only conv_igemm_kernel adds a residual / runs a second convolution's tap: the specialised kernels are skipped)
  if (g.src2 != nullptr && (!g.dgrad || g.stride < 2 || g.wgt2 == nullptr || g.batch > 1 || g.IC % 16 != 0 || g.R * g.S > 62)) return hipErrorInvalidValue;
  if (res && !g.dgrad) return hipErrorInvalidValue;
  if (!res && conv_oc4_supported(g, bias, stats)) {
    const long M = (long)g.N * g.OH * g.OW;
    long blocks = (long)g.N * ((g.OH + 15) / 16) * ((g.OW + 15) / 16);     // 16 x 16-pixel tiles
    const long cap = (long)device_cus() * 16;              // grid-stride: the filter goes to LDS once per workgroup
    if (blocks > cap) blocks = cap;
    static const int tiled = [] { const char* e = getenv("HIFIHR_OC4_TILE"); return e ? atoi(e) : 1; }();
    if (g.IC == 64 && tiled) {
      long tb = (long)g.N * ((g.OH + 7) / 8) * ((g.OW + 15) / 16);             // 16 x 8-pixel tiles, two workgroups per CU
      if (tb > (long)device_cus() * 2) tb = (long)device_cus() * 2;
      hipLaunchKernelGGL(conv3x3_oc4_tile_kernel<64>, dim3((unsigned)tb), dim3(256), 0, st, src, wgt, dst, g.N, g.OH, g.OW, g.dgrad ? -1 : 1);
    } else if (g.IC == 64) hipLaunchKernelGGL(conv3x3_oc4_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, st, src, wgt, dst, g.N, g.OH, g.OW, g.dgrad ? -1 : 1);
    else hipLaunchKernelGGL(conv3x3_oc4_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, st, src, wgt, dst, g.N, g.OH, g.OW, g.dgrad ? -1 : 1);
    return hipGetLastError();
  }
  // (the GEMM kernels have no activation epilogue: a fused ReLU keeps the implicit-GEMM kernel)
  if (!res && conv_is_gemm(g) && bias == nullptr && !g.relu && (stats == nullptr || bgemm_nt_stats_supported(g.OC))) {
    // 1x1 / stride 1: y[M][OC] = x[M][IC] . w[OC][IC]^T, and backward-data the same product on (dy, w^T): the GEMM kernels of
    // csrc/gemm.hip (bgemm_nt_rows_kernel: N % 128 == 0; the statistics of a batch-norm consumer come out of its epilogue)
    const long M = (long)g.N * g.OH * g.OW;
    const hipError_t e = launch_bgemm_nt(src, wgt, dst, (int)M, g.OC, g.IC, 1, nullptr, 0, st, stats);      // (statistics: the row-share kernel's epilogue)
    // hipErrorNotReady: the ragged form wants the zero page and this is its first use inside a stream capture -- the implicit-GEMM
    // kernel below takes the launch, as for the halo and stem kernels
    if (e != hipErrorNotReady) return e;
  }
  if (!res && conv_halo_supported(g, bias)) {
    if (const float* zeros = conv_halo_zero_page(st)) return launch_conv_halo(g, src, wgt, bias, dst, stats, zeros, st);
  }
  if (conv_rows_supported(g, bias) && (stats == nullptr || bgemm_nt_stats_supported(g.OC))) {
    // strided 3x3 / 1x1 forward: the row-share GEMM with the patch gather in its loader waves (csrc/gemm.hip); without the zero page
    // (first use inside a stream capture) the implicit-GEMM kernel below takes the launch
    if (const float* zeros = conv_halo_zero_page(st)) return launch_conv_rows(g, src, wgt, dst, stats, zeros, st);
  }
  if (conv_stem_supported(g, bias)) {
    if (const float* zeros = conv_halo_zero_page(st)) return launch_conv_stem(g, src, wgt, dst, stats, zeros, st);
  }
  const bool generic = (g.IC % 16) != 0 || g.R * g.S > 62;   // (62 tap bits + the row bit + the "no tap" bit of the gather's mask)
  if (generic && g.dgrad && g.stride != 1) return hipErrorInvalidValue;   // strided dgrad needs source channels % 16 == 0
  int bk = (g.IC % 32 == 0) ? 32 : 16;
  if (const char* e = getenv("HIFIHR_CONV_BK")) bk = (atoi(e) == 32 && g.IC % 32 == 0) ? 32 : 16;   // tuning override
  const int classes = g.dgrad ? g.stride * g.stride : 1;
  if (g.batch > 1 && (classes != 1 || bias != nullptr || stats != nullptr)) return hipErrorInvalidValue;
  const int st_ = g.dgrad ? g.stride : 1;
  const long Mmax = (long)g.N * ((g.OH + st_ - 1) / st_) * ((g.OW + st_ - 1) / st_);   // rows of the largest class
  const int tile = pick_tile(Mmax * classes, g.OC, generic);
  if (sk_ws != nullptr && tile == 2 && bk == 32 && bias == nullptr && g.src2 == nullptr) {
    const SkPlan p = sk_plan(g);
    if (p.use && sk_ws_bytes >= conv_sk_workspace_bytes(g)) {
      if (p.v.bm == 64) launch_sk<64, 64, 32>(p, g, src, wgt, dst, stats, sk_ws, st);
      else if (p.v.bn == 128) launch_sk<128, 128, 16>(p, g, src, wgt, dst, stats, sk_ws, st);
      else if (p.v.bk == 16) launch_sk<128, 64, 16>(p, g, src, wgt, dst, stats, sk_ws, st);
      else launch_sk<128, 64, 32>(p, g, src, wgt, dst, stats, sk_ws, st);
      return hipGetLastError();
    }
  }
  switch (tile) {
    case 0: launch_igemm_tile<128, 128>(g, Mmax, classes, generic, bk, src, wgt, bias, dst, stats, st); break;
    case 1: launch_igemm_tile<128, 64>(g, Mmax, classes, generic, bk, src, wgt, bias, dst, stats, st); break;
    default: launch_igemm_tile<64, 64>(g, Mmax, classes, generic, bk, src, wgt, bias, dst, stats, st);
  }
  return hipGetLastError();
}

template <int BM, int BN, int BKW>
static void launch_wgrad_tile(const ConvGeom& g, int Q, long M, int slots, const float* x, const float* dy, float* dw, hipStream_t st,
                              const float* dy2 = nullptr, float* dw2 = nullptr) {
  const int nch = (int)((M + BKW - 1) / BKW);
  const int nbatch = g.batch > 1 ? g.batch : 1;
  const int qtiles1 = (Q + BN - 1) / BN, qtiles2 = dy2 != nullptr ? (g.IC + BN - 1) / BN : 0;
  const int tiles = ((g.OC + BM - 1) / BM) * (qtiles1 + qtiles2) * nbatch;
  // tiles * splits workgroups: stay at or just below the resident slots so that every CU gets the same number (the
  // dispatcher spreads a grid evenly; 1152 workgroups on 1024 slots cost five rounds on some CUs: tools/conv_quant_probe.py)
  int splits = slots / tiles;
  if (const char* e = getenv("HIFIHR_WGRAD_SPLITS")) splits = atoi(e);
  if (const char* e = getenv("HIFIHR_WGRAD_MAXSPLIT")) { if (splits > atoi(e)) splits = atoi(e); }
  if (splits > nch / 4) splits = nch / 4;
  if (splits < 1) splits = 1;
  const int cps = (nch + splits - 1) / splits;
  splits = (nch + cps - 1) / cps;
  hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, BKW>), dim3(qtiles1 + qtiles2, (g.OC + BM - 1) / BM, splits * nbatch), dim3(256), 0, st, g, x, dy,
                     dw, cps, dy2, dw2, qtiles1);
}

// dw += the weight gradient of the strided convolution g, dw2 += the weight gradient of the 1x1 / same stride / pad 0 convolution of the same
// input with the same output channels, in ONE launch of conv_wgrad_kernel (see its header)
bool conv_wgrad_plus1x1_supported(const ConvGeom& g) {
  static const int on = [] { const char* e = getenv("HIFIHR_WGRAD_PLUS1X1"); return e ? atoi(e) : 1; }();
  return on && !g.dgrad && g.stride >= 2 && g.pad < g.R && g.pad < g.S && g.batch <= 1 && g.IC % 4 == 0 && g.OC % 4 == 0 &&
         g.OH == (g.IH - 1) / g.stride + 1 && g.OW == (g.IW - 1) / g.stride + 1 && !conv_halo_wgrad_supported(g) && !conv_stem_wgrad_supported(g) &&
         !conv_wgrad_is_gemm(g);
}
hipError_t launch_conv_wgrad_plus1x1(const ConvGeom& g, const float* x, const float* dy, float* dw, const float* dy2, float* dw2, hipStream_t st) {
  if (!conv_wgrad_plus1x1_supported(g) || dy2 == nullptr || dw2 == nullptr) return hipErrorInvalidValue;
  const long M = (long)g.N * g.OH * g.OW;
  launch_wgrad_tile<64, 64, 16>(g, g.R * g.S * g.IC, M, device_cus() * 8, x, dy, dw, st, dy2, dw2);
  return hipGetLastError();
}

size_t conv_wgrad_workspace_bytes(const ConvGeom& g) {
  if (conv_wgrad_is_gemm(g)) return conv_wgrad_gemm_bytes(g);
  if (conv_halo_wgrad_supported(g)) return conv_halo_wgrad_slab_bytes();
  if (conv_stem_wgrad_supported(g)) return conv_stem_wgrad_slab_bytes();
  return 0;
}

// The stem with a 3-channel parameter (reference conv1 = nn.Conv2d(3, 64, 7, 2, 3)): x is NHWC4 with a zero fourth plane, dw3 is
// [64][7][7][3].  Only the slab kernel writes that layout; the caller provides its scratch (conv_wgrad_workspace_bytes).
bool conv_wgrad_c3_supported(const ConvGeom& g) { return g.IC == 4 && conv_stem_wgrad_supported(g); }
hipError_t launch_conv_wgrad_c3(const ConvGeom& g, const float* x, const float* dy, float* dw3, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!conv_wgrad_c3_supported(g) || ws == nullptr || ws_bytes < conv_stem_wgrad_slab_bytes()) return hipErrorInvalidValue;
  return launch_conv_stem_wgrad(g, x, dy, dw3, static_cast<float*>(ws), st, 3);
}

hipError_t launch_conv_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st) {
  const long M = (long)g.N * g.OH * g.OW;
  const int Q = g.R * g.S * g.IC;
  if (g.IC % 4 != 0 || g.OC % 4 != 0) return hipErrorInvalidValue;
  if (conv_wgrad_is_gemm(g) && ws != nullptr && ws_bytes >= conv_wgrad_gemm_bytes(g)) {
    // 1x1 / stride 1: dw[OC][IC] += dy[M][OC]^T . x[M][IC] -- the TN product of csrc/gemm.hip over row slabs, summed in slab order
    const int parts = bgemm_tn_parts(g.OC, g.IC, (int)M, 1);
    const hipError_t e = launch_bgemm_tn(dy, x, static_cast<float*>(ws), g.OC, g.IC, (int)M, 1, parts, st);
    if (e != hipSuccess) return e;
    const long n4 = (long)g.OC * g.IC / 4;
    hipLaunchKernelGGL(slab_sum_acc_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, static_cast<const float*>(ws), parts, n4, dw);
    return hipGetLastError();
  }
  // slab kernels: the caller's scratch (any contents) when it is large enough, else library-owned scratch (csrc/conv_halo.hip)
  if (conv_halo_wgrad_supported(g)) {
    const hipError_t e = launch_conv_halo_wgrad(g, x, dy, dw, ws_bytes >= conv_halo_wgrad_slab_bytes() ? static_cast<float*>(ws) : nullptr, st);
    if (e != hipErrorNotReady) return e;
  }
  if (conv_stem_wgrad_supported(g)) {
    const hipError_t e = launch_conv_stem_wgrad(g, x, dy, dw, ws_bytes >= conv_stem_wgrad_slab_bytes() ? static_cast<float*>(ws) : nullptr, st);
    if (e != hipErrorNotReady) return e;
  }
  // 0: 128x128 (4 resident per CU), 1: 64x128, n >= 2: 64x64 with n workgroups per CU.  Measured at B = 32 (tools/time_wgrad.py):
  // the atomic epilogue moves (workgroups x tile bytes), so below 512 output channels the 64x64 tile (a quarter of the atomic
  // volume, 8 per CU) wins by 5-15 %; the 29.6 GFLOP layer-4 shapes keep the 128x128 tile (100 vs 81 TF).
  int tile = (g.OC % 128 == 0 && g.OC >= 512) ? 0 : 8;
  if (const char* e = getenv("HIFIHR_WGRAD_TILE")) tile = atoi(e);
  const int cus = device_cus();
  int bkw = 16;
  if (const char* e = getenv("HIFIHR_WGRAD_BK")) bkw = atoi(e);
  if (tile == 0) {
    if (bkw == 32) launch_wgrad_tile<128, 128, 32>(g, Q, M, cus * 2, x, dy, dw, st);
    else launch_wgrad_tile<128, 128, 16>(g, Q, M, cus * 4, x, dy, dw, st);
  } else if (tile == 1) {
    launch_wgrad_tile<64, 128, 16>(g, Q, M, cus * 4, x, dy, dw, st);
  } else {
    if (bkw == 32) launch_wgrad_tile<64, 64, 32>(g, Q, M, cus * (tile == 2 ? 4 : tile), x, dy, dw, st);
    else launch_wgrad_tile<64, 64, 16>(g, Q, M, cus * (tile == 2 ? 4 : tile), x, dy, dw, st);
  }
  return hipGetLastError();
}

#if defined(HIFIHR_CONV_STAMP)
}  // namespace hifihr
extern "C" int hifihr_conv_stamp_read(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(hifihr::g_conv_stamp), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hifihr::g_conv_stamp), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
namespace hifihr {
#endif

hipError_t launch_weight_transpose(const float* w, float* wt, int K, int RS, int C, hipStream_t st) {
  const size_t n = (size_t)K * RS * C;
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(weight_transpose_kernel, dim3(blocks), dim3(256), 0, st, w, wt, K, RS, C);
  return hipGetLastError();
}

hipError_t launch_image_to_nhwc4(const float* img, float* out, int B, int H, int W, int OH, int OW, int pt, int pl, int normalize,
                                 hipStream_t st) {
  const size_t n = (size_t)B * OH * OW;
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(image_to_nhwc4_kernel, dim3(blocks), dim3(256), 0, st, img, reinterpret_cast<float4*>(out), B, H, W, OH, OW, pt, pl,
                     normalize);
  return hipGetLastError();
}

}  // namespace hifihr
