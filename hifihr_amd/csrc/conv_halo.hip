// 3x3 / stride 1 / pad 1 NHWC fp32 convolution with 64 input and 64 output channels (ResNet layer 1: reference
// network/res_encoder.py:364-373 dispatches these to cuDNN / MIOpen): forward and backward-data on the f32 matrix cores
// (v_mfma_f32_16x16x4_f32), input staged ONCE per output tile.
//
// Why a second kernel beside conv_igemm_kernel (csrc/conv.hip): as an implicit GEMM this layer is M = 100 352 pixels x N = 64 x K = 576
// at batch 32 -- the narrow N makes every MFMA pay for 1.5x the operand bytes of a square tile, the gather re-reads every input pixel
// once per tap (9x), and round 1's ablation put 28 % of the main loop into register-staged loads + LDS stores.  Here
//   * an output tile is 8 rows x 14 columns (112 pixels = 7 MFMA row blocks) x all 64 output channels; its 10 x 16-pixel input halo
//     (40 KB) goes to LDS once, by LDS-DMA, and the nine taps read it at shifted addresses;
//   * the weights stream through a 4-stage ring of one tap each ([2 channel halves][64 n][32 c], 16 KB, 9 per tile), laid out like the
//     B operand of bgemm_ws_kernel (csrc/gemm.hip): 4 dedicated loader waves, counted vmcnt waits, one raw s_barrier per tap;
//   * 4 MFMA waves (one 16-channel column block each, 7 accumulators) see only ds_read_b128 + MFMA;
//   * workgroups are persistent (one per CU) and walk equal shares of the (image, column tile, row) strips, cut into tiles of 8 rows
//     (or 4 at the end of a share: 7 168 strips / 256 CUs = 28 = 8 + 8 + 8 + 4 at batch 32), so no quantisation round is lost;
//     the next tile's halo is loaded while the current one is multiplied (two halo buffers).
// LDS: 2 x 44 KB halo + 4 x 16 KB weight stages = 152 KB.  Per tile 187 KB come from L2 (implicit GEMM, 128 x 64 tile: 442 KB).
// Halo image: pixel-major rows of 64 floats + 4 floats of padding (272 B): an MFMA operand read touches 16 (nearly always)
// consecutive halo pixels at one 16-byte segment each -> 16-byte bank units (pixel + segment) mod 16, at most one 2-way conflict per
// ds_read_b128 lane group -- and, unlike an XOR swizzle, a tap or channel offset is a plain addition to a lane's address.  The
// LDS-DMA writes the image in lane-linear 1 KiB pieces (43 per halo); lanes that fall on padding, on pixels outside the image or
// past the halo's end read a zero buffer.
// Backward-data = the same kernel on dy with the [C][R][S][K] transposed filter and mirrored tap offsets (sign = -1).
// Forward epilogue: per-channel sum / sum of squares of the outputs for the following batch norm, accumulated in registers over
// the workgroup's tiles and added once to stats slot (workgroup & 31) (same buffers as conv_igemm_kernel's epilogue, csrc/bn.hip).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "hifihr_internal.h"
#include "lds_dma.h"

namespace hifihr {

namespace {

constexpr int kTW = 14, kTH = 8;                 // output tile: columns, rows
constexpr int kHP = 16;                          // halo pitch (pixels): 14 + 2
constexpr int kHaloPix = (kTH + 2) * kHP;        // 160
constexpr int kPixF = 68;                        // floats per halo pixel row: 64 channels + 4 of padding (272 B)
constexpr int kHaloPieces = (kHaloPix * kPixF * 4 + 1023) / 1024;    // 1 KiB LDS-DMA pieces that cover a halo buffer: 43
constexpr int kHalo = (kHaloPieces + 1) / 2 * 2 * 256;               // floats per halo buffer (44 pieces = 44 KB)
constexpr int kWStage = 64 * 64;                 // floats per weight stage: one tap, [2 channel halves][64 n][32 c] (16 KB)
constexpr int kStages = 4;                       // weight stages (a power of two)
constexpr int kTaps = 9;
constexpr int kNL = 4;                           // loader waves
constexpr int kHaloPer = (kHaloPieces + kNL - 1) / kNL;             // halo pieces per loader wave: 11 (the last wave: 10)
static_assert(2 * (kTaps - 3) >= kHaloPer, "the next tile's halo is issued two pieces per tap and must land two taps before the tile ends");

struct HaloArgs {
  const float* src;     // [N][H][W][64]
  const float* wgt;     // [64][9][64]: row n, q = tap * 64 + c
  float* dst;           // [N][H][W][64]
  float* stats;         // [kStatSlots][2][64] or null
  const float* zeros;   // >= 16 bytes of zeros
  const float* bias;    // [64] or null: added in the epilogue (VGG19's conv1_2 of the perceptual loss), then ReLU if `relu`
  int relu;
  int N, H, W, sign;    // sign = +1: tap (r, s) reads (y + r - 1, x + s - 1); -1: (y + 1 - r, x + 1 - s)
  int ctiles, total, per;   // W / 14; N * ctiles * H strips ordered (n, column tile, y); strips per workgroup
};

struct Tile { int n, x0, y0, rows; };

// rows of the tile that starts at strip `cur` of a share ending at `end`; the tile itself (strips are ordered (n, column tile, y))
// `lead`: the row limit of the share's FIRST tile (strip `first`), kTH elsewhere -- odd workgroups of conv_wino2_kernel start with a 4-row
// tile so that the tile ends (output bursts) of neighbouring workgroups do not coincide
__device__ __forceinline__ int tile_rows(int H, int cur, int end, int first = -1, int lead = kTH) {
  return min(min(cur == first ? lead : kTH, H - cur % H), end - cur);
}
__device__ __forceinline__ Tile tile_of(int H, int ctiles, int cur, int end, int first = -1, int lead = kTH) {
  const int col = cur / H, y = cur - col * H;
  const int n = col / ctiles, ct = col - n * ctiles;
  return Tile{n, ct * kTW, y, min(min(cur == first ? lead : kTH, H - y), end - cur)};
}

// the tile that starts at strip `cur` of a share ending at `end`
__device__ __forceinline__ Tile tile_at(const HaloArgs& a, int cur, int end) {
  const int col = cur / a.H, y = cur - col * a.H;
  const int n = col / a.ctiles, ct = col - n * a.ctiles;
  int rows = min(min(kTH, a.H - y), end - cur);
  return Tile{n, ct * kTW, y, rows};
}

}  // namespace

// HIFIHR_HALO_STAMP (diagnostic build, tools/build_halo_probe.sh + tools/halo_stamp.py): MFMA wave 0 of every workgroup adds to
// g_halo_stamp [0] cycles inside the chunk loops, [1] 100 MHz ticks of the same spans, [2] chunks, [3] cycles at the per-chunk
// barrier, [4] workgroups, [5] cycles kernel entry -> exit, [6] cycles in the epilogues, [7] loader wave 0: cycles waiting on vmcnt
// HIFIHR_HALO_ABLATE (timing experiments only, results are wrong): 1 = no loader waves and no barriers (the MFMA waves' own
// instruction stream alone), 2 = also no LDS reads (MFMA only), 3 = loaders and barriers kept, LDS reads removed
#ifndef HIFIHR_HALO_ABLATE
#define HIFIHR_HALO_ABLATE 0
#endif
#if defined(HIFIHR_HALO_STAMP)
__device__ unsigned long long g_halo_stamp[8];
#define HALO_T() __builtin_amdgcn_s_memtime()
#endif
#if HIFIHR_HALO_ABLATE == 1 || HIFIHR_HALO_ABLATE == 2
#define HALO_BARRIER() ((void)0)
#else
#define HALO_BARRIER() HIFIHR_RAW_BARRIER()
#endif

// EPI: the epilogue adds a bias, applies ReLU and masks a ragged last column tile (the perceptual loss's VGG19 conv1_2; any W % 14 != 0).
// The encoder's layer 1 runs the plain form: the extra epilogue work cost it 2.5 us per launch (rocprofv3, 69.9 -> 72.4 us).
template <bool EPI>
__global__ __launch_bounds__(256 + 64 * kNL) void conv_halo_kernel(HaloArgs a) {
#if defined(HIFIHR_HALO_STAMP)
  const unsigned long long st_entry = HALO_T();
  unsigned long long st_loop = 0, st_real = 0, st_bar = 0, st_epi = 0, st_vm = 0;
#endif
  __shared__ __attribute__((aligned(1024))) float lds[2 * kHalo + kStages * kWStage];
  float* const halo = lds;
  float* const wst = lds + 2 * kHalo;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int s_lo = wg * a.per, s_hi = min(s_lo + a.per, a.total);
  if (s_lo >= s_hi) return;                                  // (uniform)
  int ntiles = 0;
  for (int cur = s_lo; cur < s_hi; cur += tile_at(a, cur, s_hi).rows) ++ntiles;
  const int ntaps = ntiles * kTaps;                          // iterations of this workgroup: one per (tile, tap)

  if (wave >= 4) {
    if (HIFIHR_HALO_ABLATE == 1 || HIFIHR_HALO_ABLATE == 2) return;
    // ---------------- loader ----------------
    const int l = wave - 4;
    HIFIHR_SET_LOADER_PRIO();
    // weights of one tap: 16 pieces; piece q = l + 4 i: channel half q >> 3, rows 8 (q & 7) .. + 7 of that [64][32] half;
    // lane -> (row, physical 16-byte segment), which holds logical segment (lane & 7) ^ ((row >> 1) & 7)
    unsigned woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = l + kNL * i;
      const int row = 8 * (q & 7) + (lane >> 3);
      woff[i] = (unsigned)row * 576u + (unsigned)((q >> 3) * 32) + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 4);
    }
    auto issue_w = [&](int gt) {                             // global tap gt -> stage gt & 3
      float* base = wst + (gt & (kStages - 1)) * kWStage;
      const float* src = a.wgt + (gt % kTaps) * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) HIFIHR_GLDS16(src + woff[i], base + 256 * (l + kNL * i), lane);
    };
    // halo: piece hq = l + 4 i fills LDS bytes [1024 hq, 1024 hq + 1024) of the buffer; lane -> 16 bytes at 1024 hq + 16 lane =
    // segment (that offset % 272) / 16 of halo pixel (that offset / 272); the padding and the bytes past pixel 159 read zeros.
    // The (pixel, segment) of a lane's share of piece i does not depend on the tile: packed once, (dy << 12 | dx << 8 | seg) or -1.
    int hpk[kHaloPer];
#pragma unroll
    for (int i = 0; i < kHaloPer; ++i) {
      const int o = (l + kNL * i) * 1024 + lane * 16;
      const int hp = o / (kPixF * 4), within = o - hp * (kPixF * 4);
      hpk[i] = (hp < kHaloPix && within < 256) ? ((hp >> 4) << 12) | ((hp & 15) << 8) | (within >> 4) : -1;
    }
    auto issue_h1 = [&](const Tile& t, int buf, int i) {     // (i: compile-time after unrolling)
      const int hq = l + kNL * i;
      const int pk = hpk[i];
      const int iy = t.y0 - 1 + (pk >> 12), ix = t.x0 - 1 + ((pk >> 8) & 15);
      const bool ok = pk >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float* src = ok ? a.src + (((size_t)t.n * a.H + iy) * a.W + ix) * 64 + (pk & 15) * 4 : a.zeros;
      HIFIHR_GLDS16(src, halo + buf * kHalo + 256 * hq, lane);
    };
    const int nh = (kHaloPieces - l + kNL - 1) / kNL;        // halo pieces of this wave: 11, 11, 11, 10
    int cur = s_lo;
    Tile t = tile_at(a, cur, s_hi);
#pragma unroll
    for (int i = 0; i < kHaloPer; ++i)
      if (i < nh) issue_h1(t, 0, i);
    issue_w(0);
    issue_w(1);
    issue_w(2);
    HIFIHR_WAIT_VM(4);                                       // halo 0 and taps 0, 1 landed; tap 2 may be in flight
    HIFIHR_RAW_BARRIER();                                    // barrier -1
    int gt = 0;
    for (int ti = 0; ti < ntiles; ++ti) {
      cur += t.rows;
      const bool more = ti + 1 < ntiles;
      Tile nt = t;
      if (more) nt = tile_at(a, cur, s_hi);
      const int nbuf = (ti + 1) & 1;
#pragma unroll
      for (int tap = 0; tap < kTaps; ++tap, ++gt) {
        // tap gt + 3 -> the stage tap gt - 1 was read from (released by barrier gt - 1); two pieces of the next tile's halo
        const bool w = gt + 3 < ntaps;
        if (w) issue_w(gt + 3);
        int hcnt = 0;
        if (more && 2 * tap < kHaloPer) {
          issue_h1(nt, nbuf, 2 * tap);
          hcnt = 1;
          if (2 * tap + 1 < nh) { issue_h1(nt, nbuf, 2 * tap + 1); hcnt = 2; }
        }
        // tap gt + 2 (issued one iteration ago) must have landed before barrier gt: the MFMA waves prefetch from it during iteration
        // gt + 1.  Loads land in order, so exactly this iteration's pieces may stay in flight.
#if defined(HIFIHR_HALO_STAMP)
        const unsigned long long v0 = HALO_T();
#endif
        const int out = (w ? 4 : 0) + hcnt;
        if (out == 6) HIFIHR_WAIT_VM(6);
        else if (out == 5) HIFIHR_WAIT_VM(5);
        else if (out == 4) HIFIHR_WAIT_VM(4);
        else if (out == 2) HIFIHR_WAIT_VM(2);
        else if (out == 1) HIFIHR_WAIT_VM(1);
        else HIFIHR_WAIT_VM(0);
#if defined(HIFIHR_HALO_STAMP)
        st_vm += HALO_T() - v0;
#endif
        HIFIHR_RAW_BARRIER();                                // barrier gt
      }
      t = nt;
    }
#if defined(HIFIHR_HALO_STAMP)
    if (tid == 256) atomicAdd(&g_halo_stamp[7], st_vm);
#endif
    return;
  }

  // ---------------- MFMA waves ----------------
  const int r = lane & 15, g = lane >> 4;
  // LDS byte addresses are (lane part) + (wave-uniform part): the halo rows are padded, not XOR-swizzled, so that a tap / channel
  // offset is a plain addition.  hoff[j]: halo pixel of tile pixel 16 j + r at tap offset (0, 0), segment g; woff[h]: this lane's
  // weight row (swizzled like the GEMM tiles), segment g + 4 h.
  int hoff[7], woff[2];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int p = 16 * j + r, ty = p / kTW;
    hoff[j] = (ty * kHP + (p - ty * kTW)) * (kPixF * 4) + g * 16;
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) woff[h] = (16 * wave + r) * 128 + (((g + 4 * h) ^ ((r >> 1) & 7)) * 16);
  const char* const halo_b = reinterpret_cast<const char*>(halo);
  const char* const wst_b = reinterpret_cast<const char*>(wst);
  // batch-norm statistics: shifted sums (d = y - sk, sk = the lane's first output of each channel; hifihr_internal.h "FORWARD statistics")
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f}, sk[4] = {0.f, 0.f, 0.f, 0.f};
  int sn = 0;

  HALO_BARRIER();                                            // barrier -1
  int cur = s_lo, gt = 0;
  // one tile with NB MFMA row blocks (7: up to 8 rows, 4: up to 4 rows = 56 pixels)
  auto run_tile = [&](auto nbc, const Tile& t, int hbuf) {
    constexpr int NB = decltype(nbc)::value;
    float fm[2][NB][4], fn[2][4];
    // fragments of quarter qd = 2 * (channel half) + h (16 input channels) of tap `tap` of this tile; weights of global tap gtt
    auto read_q = [&](int gtt, int tap, int qd, int slot) {
      if (HIFIHR_HALO_ABLATE >= 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          HIFIHR_TOUCH(fn[slot][k]);
#pragma unroll
          for (int j = 0; j < NB; ++j) HIFIHR_TOUCH(fm[slot][j][k]);
        }
        return;
      }
      const int tr = tap / 3, ts = tap - 3 * tr;
      const int toff = (1 + a.sign * (tr - 1)) * kHP + (1 + a.sign * (ts - 1));
      const int hu = hbuf * (kHalo * 4) + toff * (kPixF * 4) + qd * 64;                               // (uniform)
      {
        const float4 v = *reinterpret_cast<const float4*>(wst_b + (gtt & (kStages - 1)) * (kWStage * 4) + (qd >> 1) * 8192 + woff[qd & 1]);
        fn[slot][0] = v.x; fn[slot][1] = v.y; fn[slot][2] = v.z; fn[slot][3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(halo_b + hu + hoff[j]);
        fm[slot][j][0] = v.x; fm[slot][j][1] = v.y; fm[slot][j][2] = v.z; fm[slot][j][3] = v.w;
      }
    };
    floatx4 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    auto mfma_q = [&](int slot) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fn[slot][k], fm[slot][j][k], acc[j], 0, 0, 0);
    };
    auto touch = [&]() {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        HIFIHR_TOUCH(fn[0][k]);
#pragma unroll
        for (int j = 0; j < NB; ++j) HIFIHR_TOUCH(fm[0][j][k]);
      }
    };
    // the (NB + 1) LDS reads of the NEXT quarter are spread between the 4 NB MFMAs of the current one
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < NB + 1; ++i) {
        HIFIHR_SCHED_GROUP(0x008, 3);                        // MFMA
        HIFIHR_SCHED_GROUP(0x100, 1);                        // DS read
        HIFIHR_SCHED_GROUP(0x002, 2);                        // VALU
      }
      HIFIHR_SCHED_GROUP(0x008, 4 * NB - 3 * (NB + 1));
    };
    read_q(gt, 0, 0, 0);
    touch();
#if defined(HIFIHR_HALO_STAMP)
    const unsigned long long l0 = HALO_T(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int tap = 0; tap < kTaps; ++tap, ++gt) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        // quarter qd + 1 of this tap, or quarter 0 of the next one (landed: confirmed at barrier gt - 1; past the tile's last tap the
        // addresses stay inside the LDS array and the values are never used -- no branch: see bgemm_ws_kernel)
        if (qd < 3) read_q(gt, tap, qd + 1, (qd + 1) & 1);
        else read_q(gt + 1, tap + 1, 0, 0);
        mfma_q(qd & 1);
        interleave();
        HIFIHR_PIN();
      }
      touch();
#if defined(HIFIHR_HALO_STAMP)
      HIFIHR_TOUCH(acc[0][0]);
      const unsigned long long b0 = HALO_T();
#endif
      HALO_BARRIER();                                        // barrier gt
#if defined(HIFIHR_HALO_STAMP)
      st_bar += HALO_T() - b0;
#endif
    }
#if defined(HIFIHR_HALO_STAMP)
    const unsigned long long l1 = HALO_T();
    st_loop += l1 - l0; st_real += __builtin_amdgcn_s_memrealtime() - r0;
#endif
    // epilogue: register e of lane (r, g) of block j = out[pixel 16 j + r][channel 16 wave + 4 g + e]
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EPI) {
      if (a.bias != nullptr) b4 = *reinterpret_cast<const float4*>(a.bias + 16 * wave + 4 * g);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int p = 16 * j + r, ty = p / kTW, tx = p - ty * kTW;
      if (ty < t.rows && (!EPI || t.x0 + tx < a.W)) {        // (EPI: a ragged last column tile, W % 14 != 0 -- the loader read zeros there)
        float* o = a.dst + (((size_t)t.n * a.H + t.y0 + ty) * a.W + t.x0 + tx) * 64 + 16 * wave + 4 * g;
        float4 v = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
        if constexpr (EPI) {
          v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        *reinterpret_cast<float4*>(o) = v;
        if (sn == 0) { sk[0] = v.x; sk[1] = v.y; sk[2] = v.z; sk[3] = v.w; }
        const float d0 = v.x - sk[0], d1 = v.y - sk[1], d2 = v.z - sk[2], d3 = v.w - sk[3];
        ssum[0] += d0; ssq[0] += d0 * d0; ssum[1] += d1; ssq[1] += d1 * d1;
        ssum[2] += d2; ssq[2] += d2 * d2; ssum[3] += d3; ssq[3] += d3 * d3;
        ++sn;
      }
    }
#if defined(HIFIHR_HALO_STAMP)
    st_epi += HALO_T() - l1;
#endif
  };
  for (int ti = 0; ti < ntiles; ++ti) {
    const Tile t = tile_at(a, cur, s_hi);
    cur += t.rows;
    if (t.rows > 4) run_tile(std::integral_constant<int, 7>{}, t, ti & 1);
    else run_tile(std::integral_constant<int, 4>{}, t, ti & 1);
  }
#if defined(HIFIHR_HALO_STAMP)
  if (tid == 0) {
    atomicAdd(&g_halo_stamp[0], st_loop); atomicAdd(&g_halo_stamp[1], st_real); atomicAdd(&g_halo_stamp[2], (unsigned long long)ntaps);
    atomicAdd(&g_halo_stamp[3], st_bar); atomicAdd(&g_halo_stamp[4], 1ull); atomicAdd(&g_halo_stamp[5], HALO_T() - st_entry);
    atomicAdd(&g_halo_stamp[6], st_epi);
  }
#endif
  if (a.stats != nullptr) {                                  // (uniform)
    double S1[4], S2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      stat_unshift(sn, sk[e], ssum[e], ssq[e], S1[e], S2[e]);
      for (int o = 1; o < 16; o <<= 1) { S1[e] += __shfl_xor(S1[e], o, 64); S2[e] += __shfl_xor(S2[e], o, 64); }
    }
    if (r == 0) {
      double* sp = reinterpret_cast<double*>(a.stats) + (size_t)(wg & (stat_slots_used(64) - 1)) * 2 * 64 + 16 * wave + 4 * g;
#pragma unroll
      for (int e = 0; e < 4; ++e) { stat_atomic_add(sp + e, S1[e]); stat_atomic_add(sp + 64 + e, S2[e]); }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same layer as Winograd F(2x2, 3x3) WITHOUT transform-domain tensors in HBM (round 3).
// csrc/wino.hip / wino4.hip write V = B^T d B and read M back through HBM (2.25x / 4x the activation each way), which is why layer 1
// (64 channels at 56 x 56: the largest activations of the trunk) stayed on the direct kernel above.  Here the transforms live in the
// registers of the MFMA waves and the skeleton is the direct kernel's: halo staged once per 8 x 14 tile (two buffers), 4 loader waves,
// weights streamed through LDS, raw barriers.
//   * A STAGE is one row i of the 4 x 4 transform positions xi = 4 i + j and one half h of the input channels: U[4 i + j][64 n][32 h ..
//     + 31] for j = 0..3 (4 x 8 KB; U[16][64 n][64 c] is what hifihr_weight_prep kinds 1 / 2 already produce), a ring of two stages;
//     8 stages and 8 barriers per tile (the first version had one position per stage: 16 barriers, 1 450 cycles per 1 024 of MFMA).
//   * input transform: B^T has two non-zeros per row, so the four positions of row i share R[c'] = d[ra][c'] +- d[rb][c'] (c' = 0..3:
//     8 patch reads of 16 channels, 4 packed-pair adds each) and V_j = R[ca] +- R[cb]: 2 VALU instructions per operand register
//     where one position at a time needs 3 -- it matters: a wave's VALU instructions do NOT hide behind its own MFMAs on this part
//     (measured: removing the 12 transform instructions per 8 MFMAs took 17 % off the launch);
//   * output transform: M_xi (accumulated over both channel halves) is added with its sign to the <= 4 outputs of the 2 x 2 tile it
//     feeds (A^T = [1 1 1 0; 0 1 -1 -1]) when row i is complete: 4 running outputs per lane, nothing leaves the registers.
// A tile is 4 x 7 = 28 Winograd tiles = 2 MFMA row blocks: MFMA wave w owns row block w >> 1 and the two 16-channel column blocks of
// half w & 1 (NCB = 2); a tile of <= 4 rows (14 tiles, 1 row block) gives wave w the single column block 2 (w & 1) + (w >> 1) (NCB = 1).
// Per 8 x 14 tile and wave: 16 x 16 x 2 = 512 MFMAs against the direct kernel's 9 x 16 x 7 = 1008; per stage 64 MFMAs, 32 LDS reads.
// Rows and shares are even (launcher), W even (a ragged last column tile is masked in the epilogue).  Backward-data = the same kernel on dy with U' (kind 2: transposed, rotated filter).
// Measured at layer 1 (B = 32, 56 x 56; the direct kernel: 77 us): one position per stage, transform in front of its MFMAs 59 us; three-deep
// software pipeline of the same 58; this form 54 (in-kernel stamps: 2 400 cycles of an MFMA wave per 2 048-cycle stage + 120-200 at the
// barrier, 4 000-5 000 per tile in the epilogue, where every workgroup stores at the same moment).  Dead ends, all measured: (1) handing
// the tile's output to the loader waves through the dead halo buffer (stores spread over the next tile): the in-order vmcnt makes the
// weight stages wait for the stores to RETIRE (~2 us each): 700-1 300 cycles at every barrier, 57 us; (2) two loader + two output waves:
// a wave issues one LDS-DMA instruction per ~200 cycles, so 16 pieces per loader wave and stage take 3 500 cycles: 65 us (108 when the
// unrolled loader spilled: a scratch reload in a loader costs microseconds); (3) four accumulator chains instead of two, scheduler hints
// removed: +-0; (4) odd workgroups starting with a 4-row tile so that tile ends do not coincide (HIFIHR_W2_STAGGER): +-0 at B = 32, 18 -> 22 us at
// B = 8 -- the epilogue's cost is the issue of its 8 stores per lane, not a burst on the fabric.  profiles/r03_time_conv_wino2.txt, r03_wino2_stamps.txt.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int kW2Stages = 8;                     // per tile: (row i of positions, channel half h)
constexpr int kW2Stage = 4 * 64 * 32;            // floats per weight stage: [4 j][64 n][32 c] (32 KB)
static_assert(2 * kW2Stage == kStages * kWStage, "the two weight stages take the direct kernel's ring");
static_assert(2 * 6 >= kHaloPer, "the next tile's halo is issued two pieces per stage in stages 0..5 and waited for in stage 6");
struct Wino2Args {
  const float* src;     // [N][H][W][64]
  const float* U;       // [16][64 n][64 c]
  float* dst;           // [N][H][W][64]
  float* stats;         // forward statistics slots or null
  const float* zeros;
  const float* bias;    // [64] or null (EPI)
  const float* res;     // [N][H][W][64] or null (EPI): added to the output (backward-data + the gradient of the input's other consumer)
  int relu;
  int N, H, W;
  int ctiles, total, per;
};
// B^T row i = +d[kBa[i]] + (kBneg[i] ? -1 : +1) d[kBb[i]]
__device__ constexpr int kBa[4] = {0, 1, 2, 1}, kBb[4] = {2, 2, 1, 3};
__device__ constexpr int kBneg[4] = {1, 0, 1, 1};
// A^T[p][i]
__device__ constexpr int kAt[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
#if defined(HIFIHR_HOSTSIM)
typedef float floatx2 __attribute__((vector_size(8)));
#else
typedef float floatx2 __attribute__((ext_vector_type(2)));
#endif
struct F4 { floatx2 lo, hi; };                   // four channels as two packed pairs (v_pk_add_f32; scalar adds under -fno-slp-vectorize, which the
                                                 // guide's constants table suggests beside MFMAs, measured 55.7 vs 54.8 us here: kept packed)
__device__ __forceinline__ F4 ld_f4(const char* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return F4{floatx2{v.x, v.y}, floatx2{v.z, v.w}};
}
__device__ __forceinline__ F4 addsub(const F4& a, const F4& b, bool neg) { return neg ? F4{a.lo - b.lo, a.hi - b.hi} : F4{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ float comp(const F4& a, int k) { return k < 2 ? a.lo[k] : a.hi[k - 2]; }
}  // namespace

// HIFIHR_W2_ABLATE (timing experiments only, results are wrong; tools/build_wino2_probe.sh): 4 = the input-transform arithmetic removed
#ifndef HIFIHR_W2_ABLATE
#define HIFIHR_W2_ABLATE 0
#endif
#ifndef HIFIHR_W2_STAGGER
#define HIFIHR_W2_STAGGER 0
#endif
// STATS: the batch-norm statistics epilogue of the forward (24 registers of shifted sums) is compiled in -- the kernel sits at the 256-register
// limit of its 512-thread workgroup, and backward-data launches do without them
// (the body, so that conv_c64_bwd_pair_kernel can run it on a SHARE of a launch's workgroups: bid of nblk)
template <bool EPI, bool STATS>
__device__ __forceinline__ void wino2_body(const Wino2Args& a, float* __restrict__ lds, int bid, int nblk) {
#if defined(HIFIHR_HALO_STAMP)
  const unsigned long long st_entry = HALO_T();
  unsigned long long st_loop = 0, st_real = 0, st_bar = 0, st_epi = 0, st_vm = 0;
#endif
  float* const halo = lds;
  float* const wst = lds + 2 * kHalo;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = xcd_remap(bid, nblk);
  const int s_lo = wg * a.per, s_hi = min(s_lo + a.per, a.total);
  if (s_lo >= s_hi) return;                                  // (uniform)
  int ntiles = 0;
  const int lead = HIFIHR_W2_STAGGER && (wg & 1) ? 4 : kTH;
  for (int cur = s_lo; cur < s_hi; cur += tile_rows(a.H, cur, s_hi, s_lo, lead)) ++ntiles;
  const int nst = ntiles * kW2Stages;                        // iterations of this workgroup: one per (tile, stage)

  if (wave >= 4) {
    // ---------------- loader ----------------
    const int l = wave - 4;
    HIFIHR_SET_LOADER_PRIO();
    // weights of a stage: 32 pieces of 1 KiB; piece pq = l + 4 m: position j = pq >> 3, rows 8 (pq & 7) .. + 7 of that [64 n][32 c] block;
    // lane -> (row, physical 16-byte segment), which holds logical segment (lane & 7) ^ ((row >> 1) & 7)
    unsigned woff[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int pq = l + kNL * m;
      const int row = 8 * (pq & 7) + (lane >> 3);
      woff[m] = (unsigned)(pq >> 3) * 4096u + (unsigned)row * 64u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 4);
    }
    auto issue_w = [&](int gs) {                             // global stage gs -> ring slot gs & 1
      float* base = wst + (gs & 1) * kW2Stage;
      const int s = gs & (kW2Stages - 1);
      const float* src = a.U + (s >> 1) * (4 * 4096) + (s & 1) * 32;
#pragma unroll
      for (int m = 0; m < 8; ++m) HIFIHR_GLDS16(src + woff[m], base + 256 * (l + kNL * m), lane);
    };
    int hpk[kHaloPer];
#pragma unroll
    for (int i = 0; i < kHaloPer; ++i) {
      const int o = (l + kNL * i) * 1024 + lane * 16;
      const int hp = o / (kPixF * 4), within = o - hp * (kPixF * 4);
      hpk[i] = (hp < kHaloPix && within < 256) ? ((hp >> 4) << 12) | ((hp & 15) << 8) | (within >> 4) : -1;
    }
    auto issue_h1 = [&](const Tile& t, int buf, int i) {
      const int hq = l + kNL * i;
      const int pk = hpk[i];
      const int iy = t.y0 - 1 + (pk >> 12), ix = t.x0 - 1 + ((pk >> 8) & 15);
      const bool ok = pk >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float* src = ok ? a.src + (((size_t)t.n * a.H + iy) * a.W + ix) * 64 + (pk & 15) * 4 : a.zeros;
      HIFIHR_GLDS16(src, halo + buf * kHalo + 256 * hq, lane);
    };
    const int nh = (kHaloPieces - l + kNL - 1) / kNL;
    int cur = s_lo;
    Tile t = tile_of(a.H, a.ctiles, cur, s_hi, s_lo, lead);
#pragma unroll
    for (int i = 0; i < kHaloPer; ++i)
      if (i < nh) issue_h1(t, 0, i);
    issue_w(0);
    HIFIHR_WAIT_VM(0);
    HIFIHR_RAW_BARRIER();                                    // barrier -1
    int gs = 0;
    for (int ti = 0; ti < ntiles; ++ti) {
      cur += t.rows;
      const bool more = ti + 1 < ntiles;
      Tile nt = t;
      if (more) nt = tile_of(a.H, a.ctiles, cur, s_hi, s_lo, lead);
      const int nbuf = (ti + 1) & 1;
#pragma unroll
      for (int s = 0; s < kW2Stages; ++s, ++gs) {
        // stage gs + 1 into the slot stage gs - 1 was read from (released by barrier gs - 1); it must have landed before barrier gs,
        // after which the MFMA waves read it.  Loads land in order: only this iteration's halo pieces (issued after it) may stay in flight.
        if (gs + 1 < nst) issue_w(gs + 1);
        int hcnt = 0;
        if (more && 2 * s < kHaloPer) {
          issue_h1(nt, nbuf, 2 * s);
          hcnt = 1;
          if (2 * s + 1 < nh) { issue_h1(nt, nbuf, 2 * s + 1); hcnt = 2; }
        }
#if defined(HIFIHR_HALO_STAMP)
        const unsigned long long v0 = HALO_T();
#endif
        if (hcnt == 2) HIFIHR_WAIT_VM(2);
        else if (hcnt == 1) HIFIHR_WAIT_VM(1);
        else HIFIHR_WAIT_VM(0);
#if defined(HIFIHR_HALO_STAMP)
        st_vm += HALO_T() - v0;
#endif
        HIFIHR_RAW_BARRIER();                                // barrier gs
      }
      t = nt;
    }
#if defined(HIFIHR_HALO_STAMP)
    if (tid == 256) atomicAdd(&g_halo_stamp[7], st_vm);
#endif
    return;
  }

  // ---------------- MFMA waves ----------------
  const int r = lane & 15, g = lane >> 4;
  const int rbw = wave >> 1, chw = wave & 1;
  auto patch_off = [&](int tt) {                             // halo byte offset of tile tt's 4 x 4 patch origin, this lane's 16-byte segment
    const int ty2 = tt / 7, tx2 = tt - 7 * ty2;
    return (2 * ty2 * kHP + 2 * tx2) * (kPixF * 4) + g * 16;
  };
  const int hoff_full = patch_off(min(16 * rbw + r, 27)), hoff_half = patch_off(min(r, 13));
  const int wswz[2] = {((g) ^ ((r >> 1) & 7)) * 16, ((g + 4) ^ ((r >> 1) & 7)) * 16};
  const char* const halo_b = reinterpret_cast<const char*>(halo);
  const char* const wst_b = reinterpret_cast<const char*>(wst);
  // batch-norm statistics (shifted sums): set s = the lane's channels 32 chw + 16 s + 4 g .. + 3
  float ssum[2][4] = {}, ssq[2][4] = {}, sk[2][4] = {};
  int sn[2] = {0, 0};

  HIFIHR_RAW_BARRIER();                                      // barrier -1
  int cur = s_lo, gs = 0;
  auto run_tile = [&](auto ncbc, const Tile& t, int hbuf) {
    constexpr int NCB = decltype(ncbc)::value;
    const int tt = NCB == 2 ? 16 * rbw + r : r;
    const int hoff = hbuf * (kHalo * 4) + (NCB == 2 ? hoff_full : hoff_half);
    const int cb0 = NCB == 2 ? 2 * chw : 2 * chw + rbw;
    const int wrow = (16 * cb0 + r) * 128;
    F4 raw[8];                                                // patch rows (ra, rb) x columns 0..3 of one 16-channel quarter
    F4 V[4];                                                  // the four positions' operands of a quarter
    float4 wv[2][NCB];                                        // weights of one position of a quarter, two buffers (j & 1)
    floatx4 Y[4][NCB], M[4][NCB];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < NCB; ++c) Y[p][c] = floatx4{0.f, 0.f, 0.f, 0.f};
    // quarter Q = 2 s + q of the tile: stage s = (i = s >> 1, h = s & 1), channels 32 h + 16 q + 4 g .. + 3
    auto read_raw = [&](int Q) {
      const int i = (Q >> 2) & 3, qd = Q & 3;                 // qd = 2 h + q: the quarter's channel offset
      const char* hb = halo_b + hoff + qd * 64;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        raw[c] = ld_f4(hb + (kBa[i] * kHP + c) * (kPixF * 4));
        raw[4 + c] = ld_f4(hb + (kBb[i] * kHP + c) * (kPixF * 4));
      }
    };
    auto read_w = [&](int gst, int q, int j) {                // weights of position j, quarter q of global stage gst
      const char* wb = wst_b + (gst & 1) * (kW2Stage * 4) + j * 8192 + wrow + wswz[q];
#pragma unroll
      for (int c = 0; c < NCB; ++c) wv[j & 1][c] = *reinterpret_cast<const float4*>(wb + c * (16 * 128));
    };
    auto transform = [&](int Q) {                             // raw -> V
      const int i = (Q >> 2) & 3;
      F4 R[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) R[c] = addsub(raw[c], raw[4 + c], kBneg[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        V[j] = addsub(R[kBa[j]], R[kBb[j]], kBneg[j]);
        if (HIFIHR_W2_ABLATE == 4) V[j] = raw[j];
      }
    };
    auto fold = [&](int i) {                                  // row i's accumulators into the outputs they feed: A^T[p >> 1][i] A^T[p & 1][j]
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int co = kAt[p >> 1][i] * kAt[p & 1][j];
          if (co == 1) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) Y[p][c] += M[j][c];
          } else if (co == -1) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) Y[p][c] -= M[j][c];
          }
        }
    };
    read_raw(0);
#if defined(HIFIHR_HALO_STAMP)
    const unsigned long long l0 = HALO_T(), r0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int s = 0; s < kW2Stages; ++s, ++gs) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int Q = 2 * s + q;
        if (q == 0) {
          read_w(gs, 0, 0);                                   // (stage gs landed: confirmed at barrier gs - 1)
          if ((s & 1) == 0 && s > 0) fold((s >> 1) - 1);      // the previous row: its MFMAs retired before the barrier
        }
        transform(Q);
        if (Q + 1 < 2 * kW2Stages) read_raw(Q + 1);           // lands behind this quarter's MFMAs (the next tile's first quarter is read
                                                              // at its start: another halo buffer)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < 3) read_w(gs, q, j + 1);
          else if (q == 0) read_w(gs, 1, 0);
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
              // the row's first quarter starts from zero
              const floatx4 acc = ((Q & 3) == 0 && k == 0) ? floatx4{0.f, 0.f, 0.f, 0.f} : M[j][c];
              M[j][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(reinterpret_cast<const float*>(&wv[j & 1][c])[k], comp(V[j], k), acc, 0, 0, 0);
            }
          // the prefetch FIRST: left alone the scheduler sinks it below this position's MFMAs (the two weight buffers then share
          // registers) and the next position starts by waiting for the LDS -- 8 exposed latencies per stage
          HIFIHR_SCHED_GROUP(0x100, NCB);
          HIFIHR_SCHED_GROUP(0x008, 4 * NCB);
          HIFIHR_PIN();
        }
      }
#if defined(HIFIHR_HALO_STAMP)
      HIFIHR_TOUCH(M[0][0][0]);
      const unsigned long long b0 = HALO_T();
#endif
      HIFIHR_RAW_BARRIER();                                  // barrier gs
#if defined(HIFIHR_HALO_STAMP)
      st_bar += HALO_T() - b0;
#endif
    }
#if defined(HIFIHR_HALO_STAMP)
    const unsigned long long l1 = HALO_T();
    st_loop += l1 - l0; st_real += __builtin_amdgcn_s_memrealtime() - r0;
#endif
    fold(3);
    // epilogue: register e of Y[2 py + px][c] = out[y0 + 2 ty2 + py][x0 + 2 tx2 + px][16 (cb0 + c) + 4 g + e]
    if (tt < (t.rows >> 1) * 7 && t.x0 + 2 * (tt % 7) < a.W) {      // (a ragged last column tile: the loader read zeros there)
      const int ty2 = tt / 7, tx2 = tt - 7 * ty2;
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        const int ch = 16 * (cb0 + c) + 4 * g;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 r4[4] = {b4, b4, b4, b4};
        if constexpr (EPI) {
          if (a.bias != nullptr) b4 = *reinterpret_cast<const float4*>(a.bias + ch);
          if (a.res != nullptr) {                             // (uniform) the four loads of this column block in flight together
#pragma unroll
            for (int p = 0; p < 4; ++p)
              r4[p] = *reinterpret_cast<const float4*>(a.res + (((size_t)t.n * a.H + t.y0 + 2 * ty2 + (p >> 1)) * a.W + t.x0 + 2 * tx2 + (p & 1)) * 64 + ch);
          }
        }
        auto put = [&](float (&k4)[4], float (&s4)[4], float (&q4)[4], int& n) {
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            float* o = a.dst + (((size_t)t.n * a.H + t.y0 + 2 * ty2 + (p >> 1)) * a.W + t.x0 + 2 * tx2 + (p & 1)) * 64 + ch;
            float4 v = make_float4(Y[p][c][0], Y[p][c][1], Y[p][c][2], Y[p][c][3]);
            if constexpr (EPI) {
              v.x += b4.x + r4[p].x; v.y += b4.y + r4[p].y; v.z += b4.z + r4[p].z; v.w += b4.w + r4[p].w;
              if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            *reinterpret_cast<float4*>(o) = v;
            if constexpr (STATS) {
#if HIFIHR_W2_ABLATE != 7
            if (n == 0) { k4[0] = v.x; k4[1] = v.y; k4[2] = v.z; k4[3] = v.w; }
            const float d0 = v.x - k4[0], d1 = v.y - k4[1], d2 = v.z - k4[2], d3 = v.w - k4[3];
            s4[0] += d0; q4[0] += d0 * d0; s4[1] += d1; q4[1] += d1 * d1;
            s4[2] += d2; q4[2] += d2 * d2; s4[3] += d3; q4[3] += d3 * d3;
            ++n;
#endif
            }
          }
        };
        if (NCB == 2 ? c == 0 : rbw == 0) put(sk[0], ssum[0], ssq[0], sn[0]);
        else put(sk[1], ssum[1], ssq[1], sn[1]);
      }
    }
#if defined(HIFIHR_HALO_STAMP)
    st_epi += HALO_T() - l1;
#endif
  };
  for (int ti = 0; ti < ntiles; ++ti) {
    const Tile t = tile_of(a.H, a.ctiles, cur, s_hi, s_lo, lead);
    cur += t.rows;
    if (t.rows > 4) run_tile(std::integral_constant<int, 2>{}, t, ti & 1);
    else run_tile(std::integral_constant<int, 1>{}, t, ti & 1);
  }
#if defined(HIFIHR_HALO_STAMP)
  if (tid == 0) {
    atomicAdd(&g_halo_stamp[0], st_loop); atomicAdd(&g_halo_stamp[1], st_real); atomicAdd(&g_halo_stamp[2], (unsigned long long)nst);
    atomicAdd(&g_halo_stamp[3], st_bar); atomicAdd(&g_halo_stamp[4], 1ull); atomicAdd(&g_halo_stamp[5], HALO_T() - st_entry);
    atomicAdd(&g_halo_stamp[6], st_epi);
  }
#endif
  if (STATS && a.stats != nullptr) {                         // (uniform)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      double S1[4], S2[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        stat_unshift(sn[s], sk[s][e], ssum[s][e], ssq[s][e], S1[e], S2[e]);
        for (int o = 1; o < 16; o <<= 1) { S1[e] += __shfl_xor(S1[e], o, 64); S2[e] += __shfl_xor(S2[e], o, 64); }
      }
      if (r == 0) {
        double* sp = reinterpret_cast<double*>(a.stats) + (size_t)(wg & (stat_slots_used(64) - 1)) * 2 * 64 + 32 * chw + 16 * s + 4 * g;
#pragma unroll
        for (int e = 0; e < 4; ++e) { stat_atomic_add(sp + e, S1[e]); stat_atomic_add(sp + 64 + e, S2[e]); }
      }
    }
  }
}

template <bool EPI, bool STATS>
__global__ __launch_bounds__(256 + 64 * kNL) void conv_wino2_kernel(Wino2Args a) {
  __shared__ __attribute__((aligned(1024))) float lds[2 * kHalo + 2 * kW2Stage];
  wino2_body<EPI, STATS>(a, lds, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// Backward-weight of the same layer: dW[k][tap][c] += sum over pixels dy[p][k] x[p + tap][c].
// The reduction runs over PIXELS, so a workgroup keeps the whole 9 x 64 x 64 gradient in its accumulators (MFMA wave w: the 16 output
// channels 4 i + w, all 64 input channels, 9 taps = 36 accumulator tiles, 144 registers) while it walks its share of the image in the
// same 8 x 14 tiles as the forward: per tile the input halo (44 KB) and the dy tile (28 KB) go to LDS once (two buffers: the loader
// waves fill the next tile while this one is multiplied; ONE barrier per tile), and every MFMA k-step is 4 pixels:
//   A[i][kk] = dy[pixel kk][4 i + w]            one ds_read_b32 per 4-pixel group
//   B[kk][j] = x[pixel kk + tap][4 j + m]       one ds_read_b128 per (group, tap): its 4 components feed the 4 input-channel tiles m
// (the channel <-> tile-row permutation 4 i + w / 4 j + m is undone by the slab store).  36 MFMAs per 10 LDS reads.
// Each workgroup writes its partial gradient to its own slab [9][64][64]; conv_halo_wgrad_reduce_kernel adds the slabs to dW in a
// fixed order (no atomics: the weight gradient is bit-reproducible, unlike conv_wgrad_kernel's atomic epilogue).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int kDyT = kTH * kTW * 64;             // floats of a dy tile: 112 pixels x 64 channels (28 KB)
constexpr int kWgBuf = kHalo + kDyT;             // floats per buffer
constexpr int kWgPieces = kHaloPieces + kTH * kTW / 4;      // LDS-DMA pieces per tile: 43 + 28
constexpr int kWgPer = (kWgPieces + kNL - 1) / kNL;        // per loader wave: 18
struct HaloWgradArgs {
  const float* x;       // [N][H][W][64]
  const float* dy;      // [N][H][W][64]
  float* slabs;         // [workgroups][9][64][64]
  const float* zeros;
  int N, H, W;
  int ctiles, total, per;
};
}  // namespace

__device__ __forceinline__ void halo_wgrad_body(const HaloWgradArgs& a, float* __restrict__ lds, int bid, int nblk) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = xcd_remap(bid, nblk);
  HaloArgs sh;                                               // (tile_at only reads H, ctiles)
  sh.H = a.H; sh.ctiles = a.ctiles;
  const int s_lo = wg * a.per, s_hi = min(s_lo + a.per, a.total);
  int ntiles = 0;
  for (int cur = s_lo; cur < s_hi; cur += tile_at(sh, cur, s_hi).rows) ++ntiles;

  if (wave >= 4) {
    // ---------------- loader ----------------
    const int l = wave - 4;
    HIFIHR_SET_LOADER_PRIO();
    // what a lane moves in piece i does not depend on the tile: packed once (halo: dy << 12 | dx << 8 | segment or -1; dy tile:
    // row << 16 | column << 8 | segment), so a tile costs a few adds per piece instead of two divisions
    int pk[kWgPer];
#pragma unroll
    for (int i = 0; i < kWgPer; ++i) {
      const int q = l + kNL * i;
      if (q < kHaloPieces) {
        const int o = q * 1024 + lane * 16;
        const int hp = o / (kPixF * 4), within = o - hp * (kPixF * 4);
        pk[i] = (hp < kHaloPix && within < 256) ? ((hp >> 4) << 12) | ((hp & 15) << 8) | (within >> 4) : -1;
      } else {
        const int p = 4 * (q - kHaloPieces) + (lane >> 4), ty = p / kTW;
        pk[i] = (ty << 16) | ((p - ty * kTW) << 8) | (lane & 15);
      }
    }
    auto issue_tile = [&](const Tile& t, int buf) {
      float* base = lds + buf * kWgBuf;
      const float* ximg = a.x + (size_t)t.n * a.H * a.W * 64;
      const float* dyt = a.dy + (((size_t)t.n * a.H + t.y0) * a.W + t.x0) * 64;
#pragma unroll
      for (int i = 0; i < kWgPer; ++i) {
        const int q = l + kNL * i;
        if (q < kHaloPieces) {                               // (uniform) halo piece: see conv_halo_kernel
          const int iy = t.y0 - 1 + (pk[i] >> 12), ix = t.x0 - 1 + ((pk[i] >> 8) & 15);
          const bool ok = pk[i] >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          HIFIHR_GLDS16(ok ? ximg + (iy * a.W + ix) * 64 + (pk[i] & 15) * 4 : a.zeros, base + 256 * q, lane);
        } else if (q < kWgPieces) {                          // dy piece: tile pixels 4 (q - 43) .. + 3, rows past the tile read zeros
          const int ty = pk[i] >> 16, tx = (pk[i] >> 8) & 255;
          HIFIHR_GLDS16(ty < t.rows ? dyt + (ty * a.W + tx) * 64 + (pk[i] & 255) * 4 : a.zeros, base + kHalo + 256 * (q - kHaloPieces), lane);
        }
      }
    };
    int cur = s_lo;
    Tile t = tile_at(sh, cur, s_hi);
    if (ntiles > 0) issue_tile(t, 0);
    HIFIHR_WAIT_VM(0);
    HIFIHR_RAW_BARRIER();                                    // barrier -1
    for (int ti = 0; ti < ntiles; ++ti) {
      cur += t.rows;
      if (ti + 1 < ntiles) {
        t = tile_at(sh, cur, s_hi);
        issue_tile(t, (ti + 1) & 1);                         // (its buffer was last read in iteration ti - 1: released by barrier ti - 1)
      }
      HIFIHR_WAIT_VM(0);
      HIFIHR_RAW_BARRIER();                                  // barrier ti: tile ti + 1 is in LDS, buffer ti & 1 is free
    }
    return;
  }

  // ---------------- MFMA waves ----------------
  const int r = lane & 15, g = lane >> 4;
  floatx4 acc[kTaps][4];
#pragma unroll
  for (int tp = 0; tp < kTaps; ++tp)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[tp][m] = floatx4{0.f, 0.f, 0.f, 0.f};
  const char* const lds_b = reinterpret_cast<const char*>(lds);
  const int dy_lane = kHalo * 4 + (4 * r + wave) * 4;        // byte offset of this lane's dy channel inside a buffer (+ 256 * pixel)
  const int hx_lane = r * 16;                                // ... of its 16-byte input-channel segment (+ 272 * halo pixel)

  HIFIHR_RAW_BARRIER();                                      // barrier -1
  int cur = s_lo;
  for (int ti = 0; ti < ntiles; ++ti) {
    const Tile t = tile_at(sh, cur, s_hi);
    cur += t.rows;
    const char* const buf = lds_b + (ti & 1) * (kWgBuf * 4);
    const int ngroups = (t.rows * kTW + 3) >> 2;             // 4-pixel groups that hold pixels of this tile (dy is zero past its rows)
    float fy[2];
    float4 fx[2][kTaps];
    auto read_group = [&](int P, int slot) {
      P = min(P, kTH * kTW / 4 - 1);                         // (the software pipeline reads one or two groups past the last: stay inside the tile)
      const int tpix = 4 * P + g, ty = tpix / kTW;
      const int hpix = ty * kHP + (tpix - ty * kTW);
      fy[slot] = *reinterpret_cast<const float*>(buf + dy_lane + tpix * 256);
      const char* hb = buf + hx_lane + hpix * (kPixF * 4);
#pragma unroll
      for (int tp = 0; tp < kTaps; ++tp) fx[slot][tp] = *reinterpret_cast<const float4*>(hb + ((tp / 3) * kHP + (tp % 3)) * (kPixF * 4));
    };
    auto mfma_group = [&](int slot) {
#pragma unroll
      for (int tp = 0; tp < kTaps; ++tp) {
        acc[tp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[slot], fx[slot][tp].x, acc[tp][0], 0, 0, 0);
        acc[tp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[slot], fx[slot][tp].y, acc[tp][1], 0, 0, 0);
        acc[tp][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[slot], fx[slot][tp].z, acc[tp][2], 0, 0, 0);
        acc[tp][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[slot], fx[slot][tp].w, acc[tp][3], 0, 0, 0);
      }
    };
    auto interleave = [&]() {                                // 10 LDS reads of the next group between the 36 MFMAs of this one
#pragma unroll
      for (int i = 0; i < kTaps + 1; ++i) {
        HIFIHR_SCHED_GROUP(0x008, 3);
        HIFIHR_SCHED_GROUP(0x100, 1);
        HIFIHR_SCHED_GROUP(0x002, 2);
      }
      HIFIHR_SCHED_GROUP(0x008, 4 * kTaps - 3 * (kTaps + 1));
    };
    read_group(0, 0);
    for (int P = 0; P < ngroups; P += 2) {                   // (ngroups is even: rows * 14 / 4 with the tile heights used, else padded)
      read_group(P + 1, 1);                                  // (past the last group: inside the buffer, zero dy or never used)
      mfma_group(0);
      interleave();
      HIFIHR_PIN();
      read_group(P + 2, 0);
      if (P + 1 < ngroups) mfma_group(1);
      interleave();
      HIFIHR_PIN();
    }
    HIFIHR_RAW_BARRIER();                                    // barrier ti
  }
  // slab[tap][k][c]: register e of lane (r, g) of tile (tap, m) = dW[k = 4 (4 g + e) + wave][c = 4 r + m]
  float* slab = a.slabs + (size_t)wg * (kTaps * 64 * 64);
#pragma unroll
  for (int tp = 0; tp < kTaps; ++tp)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * (4 * g + e) + wave;
      *reinterpret_cast<float4*>(slab + ((size_t)tp * 64 + k) * 64 + 4 * r) = make_float4(acc[tp][0][e], acc[tp][1][e], acc[tp][2][e], acc[tp][3][e]);
    }
}

__global__ __launch_bounds__(256 + 64 * kNL) void conv_halo_wgrad_kernel(HaloWgradArgs a) {
  __shared__ __attribute__((aligned(1024))) float lds[2 * kWgBuf];
  halo_wgrad_body(a, lds, (int)blockIdx.x, (int)gridDim.x);
}

// Layer 1's backward has two independent halves per convolution -- the data gradient (conv_wino2_kernel with the transposed filters)
// and the weight gradient (conv_halo_wgrad_kernel) -- both persistent, both ~30..45 us at the config batch, i.e. short enough that the
// launch's two ends (ramp-up, and the tail where the last workgroups run alone) are a visible share of each.  ONE launch runs both: the
// first `ga` workgroups take the data gradient, the others the weight gradient, each with its own share arithmetic (bit-identical to
// the separate launches of the same share sizes).  Same idea as bgemm_nt_tn_pair_kernel (gemm.hip).
template <bool EPI>
__global__ __launch_bounds__(256 + 64 * kNL) void conv_c64_bwd_pair_kernel(Wino2Args d, int ga, HaloWgradArgs w) {
  constexpr int kPairLds = 2 * kHalo + 2 * kW2Stage > 2 * kWgBuf ? 2 * kHalo + 2 * kW2Stage : 2 * kWgBuf;
  __shared__ __attribute__((aligned(1024))) float lds[kPairLds];
  const int b = (int)blockIdx.x;
  if (b < ga) wino2_body<EPI, false>(d, lds, b, ga);         // (uniform per workgroup)
  else halo_wgrad_body(w, lds, b - ga, (int)gridDim.x - ga);
}

// dw[k][tap][c] += sum over the slabs, in a fixed order (bit-reproducible).  A workgroup takes 64 gradient elements; wave w sums slabs
// 64 q + 16 w .. + 15 (q = 0, 1, ..) with its 16 loads of a trip in flight, the four waves' partial sums meet in LDS.  (Round 3: one
// thread per element walking every slab, 32 loads in flight -- 144 workgroups, 8 dependent trips for 256 slabs: 10 us for 38 MB.)
__device__ __forceinline__ void halo_wgrad_reduce_body(const float* __restrict__ slabs, int nslab, float* __restrict__ dw, int blk) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blk * 64 + lane;                             // index into a slab: (tap * 64 + k) * 64 + c
  constexpr size_t kSlab = (size_t)kTaps * 64 * 64;
  float s = 0.f;
  for (int z0 = 16 * w; z0 < nslab; z0 += 64) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = (z0 + u < nslab) ? slabs[(size_t)(z0 + u) * kSlab + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0) {
    const int c = i & 63, k = (i >> 6) & 63, tp = i >> 12;
    dw[((size_t)k * kTaps + tp) * 64 + c] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  }
}
__global__ __launch_bounds__(256) void conv_halo_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ dw) {
  halo_wgrad_reduce_body(slabs, nslab, dw, (int)blockIdx.x);
}
// the slab sums of SEVERAL 64 -> 64 layers in one launch (round 6; blockIdx.y = layer): a step whose weight gradients are read by the
// optimizer only runs them all in front of it (hifihr_conv_halo_wgrad_reduce_multi; jobs in the kernel arguments)
constexpr int kHaloReduceMaxJobs = 16;
struct HaloReduceJobs {
  const float* slabs[kHaloReduceMaxJobs];
  float* dw[kHaloReduceMaxJobs];
  int nslab[kHaloReduceMaxJobs];
};
__global__ __launch_bounds__(256) void conv_halo_wgrad_reduce_multi_kernel(HaloReduceJobs js) {
  const int j = blockIdx.y;
  halo_wgrad_reduce_body(js.slabs[j], js.nslab[j], js.dw[j], (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// The stem: 7x7 / stride 2 / pad 3 convolution of the NHWC4 image (3 channels + a zero one) to 64 channels (reference
// network/res_encoder.py:364-373, conv1 of the vendored ResNet), forward (+ BN statistics).
// As an implicit GEMM it is M = 401 408 x N = 64 x K = 196 with 16-byte taps: conv_igemm_kernel's generic gather runs it at 52 TFLOP/s
// (134 us).  Here the whole filter lives in LDS for the life of a persistent workgroup -- [64 n][7 rows][8 tap slots][4 channels], the
// 8th slot and the 4th channel zero, row pitch padded to 228 floats (58 KB) -- an output tile is the forward kernel's 8 x 14 pixels,
// its 21 x 33-pixel input halo (12 KB at a 36-pixel pitch) is staged once by LDS-DMA (two buffers, one barrier per tile), and the
// reduction is ordered (tap row, 4 taps, channel): one ds_read_b128 hands a lane the 4 channels of ONE tap, lane group g of an MFMA
// k-step takes tap 4 KK + g, and the four MFMAs a b128 feeds are the four channels -- of which the padding channel is skipped, so a
// tile costs 7 x 2 x 3 MFMAs per row block (294 per wave) against 343 for the dense 7 x 7 x 4 ordering.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int kSWP = 228;                        // floats per filter row n in LDS: 7 * 8 * 4 = 224 + 4 (the 16 rows of an operand read spread over all banks)
constexpr int kSHP = 36;                         // halo pitch in pixels (33 used)
constexpr int kSHRows = 2 * kTH + 5;             // 21
constexpr int kSHalo = kSHRows * kSHP * 4;       // floats per halo buffer: 3024 (12 KB)
constexpr int kSPieces = (kSHalo * 4 + 1023) / 1024;        // 12 LDS-DMA pieces
constexpr int kSHaloAl = kSPieces * 256;         // floats reserved per buffer (whole pieces)
struct StemArgs {
  const float* src;     // [N][IH][IW][4]
  const float* wgt;     // [64][7][7][4]
  float* dst;           // [N][OH][OW][64]
  float* stats;
  const float* zeros;
  int N, IH, IW, OH, OW;
  int ctiles, total, per;
};
}  // namespace

__global__ __launch_bounds__(384) void conv_stem_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(1024))) float lds[64 * kSWP + 2 * kSHaloAl];
  float* const wl = lds;
  float* const halo = lds + 64 * kSWP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  HaloArgs sh;                                               // (tile_at reads H, ctiles only)
  sh.H = a.OH; sh.ctiles = a.ctiles;
  const int s_lo = wg * a.per, s_hi = min(s_lo + a.per, a.total);
  int ntiles = 0;
  for (int cur = s_lo; cur < s_hi; cur += tile_at(sh, cur, s_hi).rows) ++ntiles;
  // the filter: [n][tr][slot][c] with slot 7 zero; every wave helps.  The 4th channel is the NHWC4 padding in the encoder (its filter
  // taps are zero and its MFMAs are skipped); a caller with four real channels is detected here and gets all four.
  __shared__ int c3_any;
  if (tid == 0) c3_any = 0;
  __syncthreads();
  bool c3_mine = false;
  for (int e = tid; e < 64 * 7 * 8; e += 384) {
    const int n = e / 56, rem = e - n * 56, tr = rem >> 3, sl = rem & 7;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sl < 7) v = *reinterpret_cast<const float4*>(a.wgt + ((size_t)(n * 7 + tr) * 7 + sl) * 4);
    c3_mine = c3_mine || v.w != 0.f;
    *reinterpret_cast<float4*>(wl + n * kSWP + (tr * 8 + sl) * 4) = v;
  }
  if (c3_mine) c3_any = 1;

  if (wave >= 4) {
    HIFIHR_SET_LOADER_PRIO();
    // ---------------- loader (2 waves): the halo of the next tile ----------------
    const int l = wave - 4;
    int pk[kSPieces / 2];                                    // (row << 8 | column) of this lane's pixel in piece i, -1: padding of the LDS image
#pragma unroll
    for (int i = 0; i < kSPieces / 2; ++i) {
      const int hp = 64 * (l + 2 * i) + lane, hy = hp / kSHP, hx = hp - hy * kSHP;
      pk[i] = (hy < kSHRows && hx < 2 * kTW + 5) ? (hy << 8) | hx : -1;
    }
    auto issue_tile = [&](const Tile& t, int buf) {
      const float* ximg = a.src + (size_t)t.n * a.IH * a.IW * 4;
#pragma unroll
      for (int i = 0; i < kSPieces / 2; ++i) {
        const int iy = 2 * t.y0 - 3 + (pk[i] >> 8), ix = 2 * t.x0 - 3 + (pk[i] & 255);
        const bool ok = pk[i] >= 0 && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
        HIFIHR_GLDS16(ok ? ximg + (iy * a.IW + ix) * 4 : a.zeros, halo + buf * kSHaloAl + 256 * (l + 2 * i), lane);
      }
    };
    int cur = s_lo;
    Tile t = tile_at(sh, cur, s_hi);
    if (ntiles > 0) issue_tile(t, 0);
    HIFIHR_WAIT_VM(0);
    __syncthreads();                                         // barrier -1 (also orders the filter writes above)
    for (int ti = 0; ti < ntiles; ++ti) {
      cur += t.rows;
      if (ti + 1 < ntiles) {
        t = tile_at(sh, cur, s_hi);
        issue_tile(t, (ti + 1) & 1);
      }
      HIFIHR_WAIT_VM(0);
      HIFIHR_RAW_BARRIER();                                  // barrier ti
    }
    return;
  }

  // ---------------- MFMA waves: wave w = output channels 16 w .. 16 w + 15 ----------------
  const int r = lane & 15, g = lane >> 4;
  int hoff[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int p = 16 * j + r, ty = p / kTW;
    hoff[j] = ((2 * ty) * kSHP + 2 * (p - ty * kTW) + g) * 16;
  }
  const int woff = (16 * wave + r) * (kSWP * 4) + g * 16;
  const char* const wl_b = reinterpret_cast<const char*>(wl);
  const char* const halo_b = reinterpret_cast<const char*>(halo);
  // batch-norm statistics: shifted sums (d = y - sk, sk = the lane's first output of each channel; hifihr_internal.h "FORWARD statistics")
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f}, sk[4] = {0.f, 0.f, 0.f, 0.f};
  int sn = 0;
  __syncthreads();                                           // barrier -1
  const bool c3 = c3_any != 0;                               // (uniform)
  int cur = s_lo;
  for (int ti = 0; ti < ntiles; ++ti) {
    const Tile t = tile_at(sh, cur, s_hi);
    cur += t.rows;
    const char* const hb = halo_b + (ti & 1) * (kSHaloAl * 4);
    floatx4 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    float4 fw[2], fx[2][7];
    auto read_step = [&](int st, int slot) {                 // step st = 2 tr + KK
      const int tr = st >> 1, kk = st & 1;
      fw[slot] = *reinterpret_cast<const float4*>(wl_b + woff + (tr * 8 + 4 * kk) * 16);
      const char* hs = hb + (tr * kSHP + 4 * kk) * 16;
#pragma unroll
      for (int j = 0; j < 7; ++j) fx[slot][j] = *reinterpret_cast<const float4*>(hs + hoff[j]);
    };
    auto mfma_step = [&](int slot) {
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[slot].x, fx[slot][j].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[slot].y, fx[slot][j].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[slot].z, fx[slot][j].z, acc[j], 0, 0, 0);
      if (c3) {
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[slot].w, fx[slot][j].w, acc[j], 0, 0, 0);
      }
    };
    auto interleave = [&]() {                                // 8 LDS reads of the next step between the 21 MFMAs of this one
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        HIFIHR_SCHED_GROUP(0x008, 2);
        HIFIHR_SCHED_GROUP(0x100, 1);
        HIFIHR_SCHED_GROUP(0x002, 1);
      }
      HIFIHR_SCHED_GROUP(0x008, 5);
    };
    read_step(0, 0);
#pragma unroll
    for (int st = 0; st < 14; st += 2) {
      read_step(st + 1, 1);
      mfma_step(0);
      interleave();
      HIFIHR_PIN();
      read_step(st + 2 < 14 ? st + 2 : 13, 0);
      mfma_step(1);
      interleave();
      HIFIHR_PIN();
    }
    HIFIHR_RAW_BARRIER();                                    // barrier ti
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int p = 16 * j + r, ty = p / kTW, tx = p - ty * kTW;
      if (ty < t.rows) {
        float* o = a.dst + (((size_t)t.n * a.OH + t.y0 + ty) * a.OW + t.x0 + tx) * 64 + 16 * wave + 4 * g;
        *reinterpret_cast<float4*>(o) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (sn == 0) sk[e] = acc[j][e];
          const float d = acc[j][e] - sk[e];
          ssum[e] += d; ssq[e] += d * d;
        }
        ++sn;
      }
    }
  }
  if (a.stats != nullptr) {                                  // (uniform)
    double S1[4], S2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      stat_unshift(sn, sk[e], ssum[e], ssq[e], S1[e], S2[e]);
      for (int o = 1; o < 16; o <<= 1) { S1[e] += __shfl_xor(S1[e], o, 64); S2[e] += __shfl_xor(S2[e], o, 64); }
    }
    if (r == 0) {
      double* sp = reinterpret_cast<double*>(a.stats) + (size_t)(wg & (stat_slots_used(64) - 1)) * 2 * 64 + 16 * wave + 4 * g;
#pragma unroll
      for (int e = 0; e < 4; ++e) { stat_atomic_add(sp + e, S1[e]); stat_atomic_add(sp + 64 + e, S2[e]); }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The stem's weight gradient: dW[k][tr][s][c] += sum over output pixels dy[p][k] x[2 p + (tr, s) - 3][c].
// Same persistent tiling as conv_stem_kernel, the x halo (22 rows here) and the dy tile staged once per tile.  An MFMA k-step is 4
// output pixels; operand "A" = dy (lane r: channel 16 w + r of the wave's 16), operand "B" = one ds_read_b128 of x per (pixel, PAIR of
// tap rows): lane r of a pair reads the 4 channels of tap (2 pair + r / 8, r % 8), and the b128's components feed the MFMAs of
// channels 0, 1, 2 -- the padding channel is skipped, tap slot 7 and tap row 7 are never stored.  12 MFMAs per 5 LDS reads; the
// wave's 16 x (4 pairs x 3 channels x 16 slots) result stays in 12 accumulator tiles for its whole share and goes to a per-workgroup
// slab at the end; conv_stem_wgrad_reduce_kernel adds the slabs in order (bit-reproducible; conv_wgrad_kernel used float atomics).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int kSWRows = 2 * kTH + 6;             // halo rows incl. the one tap row 7 of the last pair would touch: 22
constexpr int kSWHalo = kSWRows * kSHP * 4;      // 3168 floats
constexpr int kSWPieces = (kSWHalo * 4 + 1023) / 1024;      // 13
constexpr int kSWHaloAl = kSWPieces * 256;       // 3328 floats
constexpr int kSWBuf = kSWHaloAl + kDyT;         // floats per buffer: halo + dy tile
constexpr int kSWAll = kSWPieces + kTH * kTW / 4;            // LDS-DMA pieces per tile: 13 + 28
struct StemWgradArgs {
  const float* x;       // [N][IH][IW][4]
  const float* dy;      // [N][OH][OW][64]
  float* slabs;         // [workgroups][64][7][7][4]
  const float* zeros;
  int N, IH, IW, OH, OW;
  int ctiles, total, per;
};
}  // namespace

__global__ __launch_bounds__(512) void conv_stem_wgrad_kernel(StemWgradArgs a) {
  __shared__ __attribute__((aligned(1024))) float lds[2 * kSWBuf];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  HaloArgs sh;
  sh.H = a.OH; sh.ctiles = a.ctiles;
  const int s_lo = wg * a.per, s_hi = min(s_lo + a.per, a.total);
  int ntiles = 0;
  for (int cur = s_lo; cur < s_hi; cur += tile_at(sh, cur, s_hi).rows) ++ntiles;

  if (wave >= 4) {
    HIFIHR_SET_LOADER_PRIO();
    // ---------------- loader (4 waves: 41 pieces per tile, each with per-lane address arithmetic -- two waves were the bottleneck) ----------------
    const int l = wave - 4;
    // what a lane moves in piece i does not depend on the tile: packed once -- halo pieces (row << 8 | column, -1: padding of the
    // image in LDS), dy pieces (tile row << 16 | tile column << 8 | 16-byte segment) -- so a tile costs a few adds per piece
    constexpr int NPW = (kSWAll + 3) / 4;
    int pk[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int q = l + 4 * i;
      if (q < kSWPieces) {
        const int hp = 64 * q + lane, hy = hp / kSHP, hx = hp - hy * kSHP;
        pk[i] = (hy < kSWRows && hx < 2 * kTW + 5) ? (hy << 8) | hx : -1;
      } else {
        const int p = 4 * (q - kSWPieces) + (lane >> 4), ty = p / kTW;
        pk[i] = (ty << 16) | ((p - ty * kTW) << 8) | (lane & 15);
      }
    }
    auto issue_tile = [&](const Tile& t, int buf) {
      float* base = lds + buf * kSWBuf;
      const float* ximg = a.x + (size_t)t.n * a.IH * a.IW * 4;
      const float* dyt = a.dy + (((size_t)t.n * a.OH + t.y0) * a.OW + t.x0) * 64;
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        const int q = l + 4 * i;
        if (q < kSWPieces) {                                 // (uniform) halo piece: 64 pixels of 16 bytes
          const int iy = 2 * t.y0 - 3 + (pk[i] >> 8), ix = 2 * t.x0 - 3 + (pk[i] & 255);
          const bool ok = pk[i] >= 0 && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
          HIFIHR_GLDS16(ok ? ximg + (iy * a.IW + ix) * 4 : a.zeros, base + 256 * q, lane);
        } else if (q < kSWAll) {                             // dy piece: 4 tile pixels; rows past the tile read zeros
          const int ty = pk[i] >> 16, tx = (pk[i] >> 8) & 255;
          HIFIHR_GLDS16(ty < t.rows ? dyt + (ty * a.OW + tx) * 64 + (pk[i] & 255) * 4 : a.zeros, base + kSWHaloAl + 256 * (q - kSWPieces), lane);
        }
      }
    };
    int cur = s_lo;
    Tile t = tile_at(sh, cur, s_hi);
    if (ntiles > 0) issue_tile(t, 0);
    HIFIHR_WAIT_VM(0);
    HIFIHR_RAW_BARRIER();                                    // barrier -1
    for (int ti = 0; ti < ntiles; ++ti) {
      cur += t.rows;
      if (ti + 1 < ntiles) {
        t = tile_at(sh, cur, s_hi);
        issue_tile(t, (ti + 1) & 1);
      }
      HIFIHR_WAIT_VM(0);
      HIFIHR_RAW_BARRIER();                                  // barrier ti
    }
    return;
  }

  // ---------------- MFMA waves: wave w = dy channels 16 w .. 16 w + 15 ----------------
  const int r = lane & 15, g = lane >> 4;
  floatx4 acc[4][4];                                         // [pair][channel]; channel 3 only moves when the input's 4th channel is not padding
#pragma unroll
  for (int pr = 0; pr < 4; ++pr)
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[pr][m] = floatx4{0.f, 0.f, 0.f, 0.f};
  const char* const lds_b = reinterpret_cast<const char*>(lds);
  const int dy_lane = kSWHaloAl * 4 + (16 * wave + r) * 4;
  const int hx_lane = ((r >> 3) * kSHP + (r & 7)) * 16;      // this lane's tap of a pair: (row r / 8, slot r % 8)
  // 4-pixel groups repeat with period 7 (two output rows = 28 pixels): per-lane byte offsets of group q's pixel inside a 2-row block,
  // so the inner loop has no division and is fully unrolled (84 MFMAs per block)
  int gdy[7], ghx[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const int tp = 4 * q + g, ty = tp / kTW;
    gdy[q] = dy_lane + tp * 256;
    ghx[q] = hx_lane + ((2 * ty) * kSHP + 2 * (tp - ty * kTW)) * 16;
  }
  HIFIHR_RAW_BARRIER();                                      // barrier -1
  int cur = s_lo;
  for (int ti = 0; ti < ntiles; ++ti) {
    const Tile t = tile_at(sh, cur, s_hi);
    cur += t.rows;
    const char* const buf = lds_b + (ti & 1) * (kSWBuf * 4);
    const int nblk = (t.rows + 1) >> 1;                      // 2-row blocks (dy is zero past the tile's rows)
    float fy[2];
    float4 fx[2][4];
    auto read_group = [&](int blk, int q, int slot) {        // (q: compile-time)
      const char* base = buf + blk * (2 * kTW * 256);        // dy: 28 pixels per block
      fy[slot] = *reinterpret_cast<const float*>(base + gdy[q]);
      const char* hb = buf + blk * (4 * kSHP * 16) + ghx[q];  // halo: 4 input rows per block
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) fx[slot][pr] = *reinterpret_cast<const float4*>(hb + (2 * pr) * kSHP * 16);
    };
    // the 4th input channel is the NHWC4 padding in the encoder (exact zeros: its products are skipped); a caller with four real
    // channels gets them: every wave scans the tile's halo once (13 LDS reads per lane) and takes the 4-channel loop if any is set
    bool nz = false;
    for (int e = lane; e < kSWRows * kSHP; e += 64) nz = nz || *reinterpret_cast<const float*>(buf + e * 16 + 12) != 0.f;
    const bool c3 = __ballot(nz) != 0ull;                    // (wave-uniform)
    auto run_blocks = [&](auto c3c) {
      constexpr bool C3 = decltype(c3c)::value;
      auto mfma_group = [&](int slot) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          acc[pr][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[slot], fx[slot][pr].x, acc[pr][0], 0, 0, 0);
          acc[pr][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[slot], fx[slot][pr].y, acc[pr][1], 0, 0, 0);
          acc[pr][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[slot], fx[slot][pr].z, acc[pr][2], 0, 0, 0);
          if (C3) acc[pr][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fy[slot], fx[slot][pr].w, acc[pr][3], 0, 0, 0);
        }
      };
      auto step = [&](int blk, int q, int rd_slot) {          // read group (blk, q) into rd_slot while the other slot is multiplied
        read_group(blk, q, rd_slot);
        mfma_group(rd_slot ^ 1);
#pragma unroll
        for (int i = 0; i < 5; ++i) {                        // 5 LDS reads between the 12 (16) MFMAs
          HIFIHR_SCHED_GROUP(0x008, 2);
          HIFIHR_SCHED_GROUP(0x100, 1);
        }
        HIFIHR_SCHED_GROUP(0x008, C3 ? 6 : 2);
        HIFIHR_PIN();
      };
      read_group(0, 0, 0);
      for (int blk = 0; blk < nblk; ++blk) {
        const int nxt = min(blk + 1, kTH / 2 - 1);           // (past the last block: a valid address, never used)
        step(blk, 1, 1); step(blk, 2, 0); step(blk, 3, 1); step(blk, 4, 0); step(blk, 5, 1); step(blk, 6, 0); step(nxt, 0, 1);
        if (++blk >= nblk) break;                            // (7 groups per block: the slot parity flips from block to block)
        const int nx2 = min(blk + 1, kTH / 2 - 1);
        step(blk, 1, 0); step(blk, 2, 1); step(blk, 3, 0); step(blk, 4, 1); step(blk, 5, 0); step(blk, 6, 1); step(nx2, 0, 0);
      }
    };
    if (c3) run_blocks(std::true_type{}); else run_blocks(std::false_type{});
    HIFIHR_RAW_BARRIER();                                    // barrier ti
  }
  // register e of lane (r, g) of tile (pair, m) = dW[k = 16 wave + 4 g + e][tr = 2 pair + r / 8][s = r % 8][c = m]
  float* slab = a.slabs + (size_t)wg * (64 * 196);
  const int trl = r >> 3, sl = r & 7;
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    const int tr = 2 * pr + trl;
    if (tr < 7 && sl < 7) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* o = slab + (((size_t)(16 * wave + 4 * g + e) * 7 + tr) * 7 + sl) * 4;
        o[0] = acc[pr][0][e]; o[1] = acc[pr][1][e]; o[2] = acc[pr][2][e]; o[3] = acc[pr][3][e];
      }
    }
  }
}

// dw[64][7][7][4] += sum over the slabs, in slab order
// dwC = channels of the destination [64][7][7][dwC]: 4, or 3 = the reference's parameter (nn.Conv2d(3, 64, 7, 2, 3)) whose image
// arrives here as NHWC4 with a zero fourth plane -- the gradient lands in the parameter's own layout, no padded temporary
__global__ __launch_bounds__(256) void conv_stem_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslab, float* __restrict__ dw,
                                                                    int dwC) {
  // (the slabs split over the four waves of a 64-element workgroup, as conv_halo_wgrad_reduce_kernel: 49 workgroups walked 256 slabs
  //  in 8 dependent trips, 9 us for 12.8 MB)
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float s = 0.f;
  for (int z0 = 16 * w; z0 < nslab; z0 += 64) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = (z0 + u < nslab) ? slabs[(size_t)(z0 + u) * (64 * 196) + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  part[w][lane] = s;
  __syncthreads();
  if (w == 0 && (i & 3) < dwC) dw[(i >> 2) * dwC + (i & 3)] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

#if defined(HIFIHR_HALO_STAMP)
}  // namespace hifihr
extern "C" int hifihr_halo_stamp_read(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(hifihr::g_halo_stamp), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hifihr::g_halo_stamp), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
namespace hifihr {
#endif

static int halo_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            ? p.multiProcessorCount : 256;
  }
  return n;
}

// 256 zero bytes the halo's out-of-image pixels are read from.  Allocated on first use (never inside a stream capture: the caller
// then falls back to conv_igemm_kernel for that launch).
const float* conv_halo_zero_page(hipStream_t st) {
  static float* page[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (page[dev] == nullptr) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 256) != hipSuccess) { (void)hipFree(p); return nullptr; }
    page[dev] = static_cast<float*>(p);
  }
  return page[dev];
}

bool conv_halo_supported(const ConvGeom& g, const float* bias) {
  static const int on = [] { const char* e = getenv("HIFIHR_CONV_HALO"); return e ? atoi(e) : 1; }();
  return on && g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.IC == 64 && g.OC == 64 && g.batch <= 1 &&
         !((g.relu || bias != nullptr) && g.dgrad) && g.IH == g.OH && g.IW == g.OW && g.OW >= kTW &&
         (long)g.N * g.OH * g.OW * 64 < (1L << 31);
}

// slabs of conv_halo_wgrad_kernel: library-owned scratch.  Weight gradients may run on a side stream beside the main one and the
// captured step runs on yet another, so a device owns a small POOL of slab buffers, all allocated at the first (eager) use, and a stream
// is bound to a free one the first time it shows up -- which needs no allocation and is therefore legal inside a stream capture.
static float* halo_wgrad_scratch(hipStream_t st, size_t bytes) {
  constexpr int kPool = 3;
  struct Dev { bool ready; size_t bytes; float* buf[kPool]; hipStream_t owner[kPool]; int used; };
  static Dev devs[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  Dev& d = devs[dev];
  if (!d.ready) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    for (int i = 0; i < kPool; ++i) {
      void* p = nullptr;
      if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
      d.buf[i] = static_cast<float*>(p);
    }
    d.bytes = bytes; d.used = 0; d.ready = true;
  }
  if (bytes > d.bytes) return nullptr;
  for (int i = 0; i < d.used; ++i)
    if (d.owner[i] == st) return d.buf[i];
  if (d.used >= kPool) return nullptr;
  d.owner[d.used] = st;
  return d.buf[d.used++];
}

bool conv_stem_wgrad_supported(const ConvGeom& g) {
  static const int on = [] { const char* e = getenv("HIFIHR_CONV_STEM_WGRAD"); return e ? atoi(e) : 1; }();
  return on && conv_stem_supported(g, nullptr);
}

size_t conv_stem_wgrad_slab_bytes() { return (size_t)halo_cus() * 64 * 196 * sizeof(float); }

hipError_t launch_conv_stem_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, float* slabs, hipStream_t st, int dw_channels) {
  if (!conv_stem_wgrad_supported(g) || (dw_channels != 3 && dw_channels != 4)) return hipErrorInvalidValue;
  const float* zeros = conv_halo_zero_page(st);
  if (zeros == nullptr) return hipErrorNotReady;
  StemWgradArgs a;
  a.x = x; a.dy = dy; a.zeros = zeros; a.N = g.N; a.IH = g.IH; a.IW = g.IW; a.OH = g.OH; a.OW = g.OW;
  a.ctiles = g.OW / kTW;
  a.total = g.N * a.ctiles * g.OH;
  int G = halo_cus();
  a.per = (a.total + G - 1) / G;
  if (a.per < 4) a.per = 4;
  G = (a.total + a.per - 1) / a.per;
  a.slabs = slabs != nullptr ? slabs : halo_wgrad_scratch(st, (size_t)halo_cus() * kTaps * 64 * 64 * sizeof(float));      // (the pool's buffers: 9 x 64 x 64 >= 64 x 196 floats each)
  if (a.slabs == nullptr) return hipErrorNotReady;
  hipLaunchKernelGGL(conv_stem_wgrad_kernel, dim3(G), dim3(512), 0, st, a);
  hipLaunchKernelGGL(conv_stem_wgrad_reduce_kernel, dim3(64 * 196 / 64), dim3(256), 0, st, a.slabs, G, dw, dw_channels);
  return hipGetLastError();
}

bool conv_halo_wgrad_supported(const ConvGeom& g) {
  static const int on = [] { const char* e = getenv("HIFIHR_CONV_HALO_WGRAD"); return e ? atoi(e) : 1; }();
  return on && conv_halo_supported(g, nullptr) && !g.dgrad && g.OW % kTW == 0;      // (the dy tiles of the weight gradient are not masked)
}

size_t conv_halo_wgrad_slab_bytes() { return (size_t)halo_cus() * kTaps * 64 * 64 * sizeof(float); }

hipError_t launch_conv_halo_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, float* slabs, hipStream_t st) {
  if (!conv_halo_wgrad_supported(g)) return hipErrorInvalidValue;
  const float* zeros = conv_halo_zero_page(st);
  if (zeros == nullptr) return hipErrorNotReady;
  HaloWgradArgs a;
  a.x = x; a.dy = dy; a.zeros = zeros; a.N = g.N; a.H = g.OH; a.W = g.OW;
  a.ctiles = g.OW / kTW;
  a.total = g.N * a.ctiles * g.OH;
  int G = halo_cus();
  a.per = (a.total + G - 1) / G;
  if (a.per < 4) a.per = 4;
  G = (a.total + a.per - 1) / a.per;
  a.slabs = slabs != nullptr ? slabs : halo_wgrad_scratch(st, (size_t)halo_cus() * kTaps * 64 * 64 * sizeof(float));
  if (a.slabs == nullptr) return hipErrorNotReady;
  hipLaunchKernelGGL(conv_halo_wgrad_kernel, dim3(G), dim3(256 + 64 * kNL), 0, st, a);
  hipLaunchKernelGGL(conv_halo_wgrad_reduce_kernel, dim3(kTaps * 64 * 64 / 64), dim3(256), 0, st, a.slabs, G, dw);
  return hipGetLastError();
}

bool conv_stem_supported(const ConvGeom& g, const float* bias) {
  static const int on = [] { const char* e = getenv("HIFIHR_CONV_STEM"); return e ? atoi(e) : 1; }();
  return on && !g.dgrad && g.R == 7 && g.S == 7 && g.stride == 2 && g.pad == 3 && g.IC == 4 && g.OC == 64 && g.batch <= 1 && !g.relu &&
         bias == nullptr && g.OW % kTW == 0 && (long)g.N * g.OH * g.OW * 64 < (1L << 31);
}

hipError_t launch_conv_stem(const ConvGeom& g, const float* src, const float* wgt, float* dst, float* stats, const float* zeros, hipStream_t st) {
  if (!conv_stem_supported(g, nullptr) || zeros == nullptr) return hipErrorInvalidValue;
  StemArgs a;
  a.src = src; a.wgt = wgt; a.dst = dst; a.stats = stats; a.zeros = zeros;
  a.N = g.N; a.IH = g.IH; a.IW = g.IW; a.OH = g.OH; a.OW = g.OW;
  a.ctiles = g.OW / kTW;
  a.total = g.N * a.ctiles * g.OH;
  int G = halo_cus();
  a.per = (a.total + G - 1) / G;
  if (a.per < 4) a.per = 4;
  G = (a.total + a.per - 1) / a.per;
  hipLaunchKernelGGL(conv_stem_kernel, dim3(G), dim3(384), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_conv_halo(const ConvGeom& g, const float* src, const float* wgt, const float* bias, float* dst, float* stats,
                            const float* zeros, hipStream_t st) {
  if (!conv_halo_supported(g, bias) || zeros == nullptr || (stats != nullptr && g.dgrad)) return hipErrorInvalidValue;
  HaloArgs a;
  a.src = src; a.wgt = wgt; a.dst = dst; a.stats = stats; a.zeros = zeros; a.bias = bias; a.relu = g.relu;
  a.N = g.N; a.H = g.OH; a.W = g.OW; a.sign = g.dgrad ? -1 : 1;
  a.ctiles = (g.OW + kTW - 1) / kTW;                         // the last column tile may be ragged (512 = 36 x 14 + 8)
  a.total = g.N * a.ctiles * g.OH;
  int G = halo_cus();
  a.per = (a.total + G - 1) / G;
  if (a.per < 4) a.per = 4;                                  // tiny problems: fewer workgroups, tiles of >= 4 rows
  G = (a.total + a.per - 1) / a.per;
  if (bias != nullptr || g.relu || g.OW % kTW != 0) hipLaunchKernelGGL(conv_halo_kernel<true>, dim3(G), dim3(256 + 64 * kNL), 0, st, a);
  else hipLaunchKernelGGL(conv_halo_kernel<false>, dim3(G), dim3(256 + 64 * kNL), 0, st, a);
  return hipGetLastError();
}

// conv_wino2_kernel: the 64 -> 64 stride-1 3x3 layers with even H and W (ResNet layer 1 at 56 x 56, VGG19 conv1_2 at 224 x 224 and, with a ragged
// last column tile, at 512 x 512 = 36 x 14 + 8)
bool conv_wino2_supported(int N, int H, int W, int C, int K) {
  static const int on = [] { const char* e = getenv("HIFIHR_CONV_WINO2"); return e ? atoi(e) : 1; }();
  return on && C == 64 && K == 64 && N > 0 && H >= 2 && H % 2 == 0 && W >= kTW && W % 2 == 0 && (long)N * H * W * 64 < (1L << 31);
}

hipError_t launch_conv_wino2(const float* src, const float* U, const float* bias, int relu, float* dst, float* stats, int N, int H, int W,
                             hipStream_t st, const float* res) {
  if (!conv_wino2_supported(N, H, W, 64, 64)) return hipErrorInvalidValue;
  const float* zeros = conv_halo_zero_page(st);
  if (zeros == nullptr) return hipErrorNotReady;
  Wino2Args a;
  a.src = src; a.U = U; a.dst = dst; a.stats = stats; a.zeros = zeros; a.bias = bias; a.relu = relu; a.res = res;
  a.N = N; a.H = H; a.W = W;
  a.ctiles = (W + kTW - 1) / kTW;                           // the last column tile may be ragged (W even: whole 2 x 2 tiles)
  a.total = N * a.ctiles * H;
  int G = halo_cus();
  a.per = (a.total + G - 1) / G;
  if (a.per < 4) a.per = 4;
  a.per = (a.per + 1) & ~1;                                  // even shares: every tile is whole 2 x 2 Winograd tiles
  G = (a.total + a.per - 1) / a.per;
  if (bias != nullptr || relu || res != nullptr) {
    if (stats != nullptr) hipLaunchKernelGGL((conv_wino2_kernel<true, true>), dim3(G), dim3(256 + 64 * kNL), 0, st, a);
    else hipLaunchKernelGGL((conv_wino2_kernel<true, false>), dim3(G), dim3(256 + 64 * kNL), 0, st, a);
  } else {
    if (stats != nullptr) hipLaunchKernelGGL((conv_wino2_kernel<false, true>), dim3(G), dim3(256 + 64 * kNL), 0, st, a);
    else hipLaunchKernelGGL((conv_wino2_kernel<false, false>), dim3(G), dim3(256 + 64 * kNL), 0, st, a);
  }
  return hipGetLastError();
}

// Data gradient + weight gradient of one 64 -> 64 3x3 stride-1 layer in ONE launch (conv_c64_bwd_pair_kernel), then the slab reduction.
//   dx = conv(dy, U_bwd) [+ res]      (launch_conv_wino2's arguments: U_bwd = the Winograd-domain filters of the transposed convolution)
//   dw += sum_p dy[p] x[p + tap]      (launch_conv_halo_wgrad's)
// The workgroups are split in proportion to the two kernels' measured times at the config batch (HIFIHR_C64_PAIR_DGRAD_PCT, default 44).
bool conv_c64_bwd_pair_supported(int N, int H, int W) {
  static const int on = [] { const char* e = getenv("HIFIHR_C64_PAIR"); return e ? atoi(e) : 1; }();
  ConvGeom g{};
  g.N = N; g.IH = H; g.IW = W; g.IC = 64; g.OC = 64; g.R = 3; g.S = 3; g.stride = 1; g.pad = 1; g.OH = H; g.OW = W; g.dgrad = 0; g.batch = 1;
  return on && conv_wino2_supported(N, H, W, 64, 64) && conv_halo_wgrad_supported(g) && halo_cus() >= 2;
}

hipError_t launch_conv_halo_wgrad_reduce_multi(const HaloReduceJob* jobs, int njobs, hipStream_t st) {
  for (int base = 0; base < njobs; base += kHaloReduceMaxJobs) {
    HaloReduceJobs js;
    const int n = njobs - base < kHaloReduceMaxJobs ? njobs - base : kHaloReduceMaxJobs;
    for (int q = 0; q < n; ++q) {
      if (jobs[base + q].slabs == nullptr || jobs[base + q].dw == nullptr || jobs[base + q].nslab <= 0) return hipErrorInvalidValue;
      js.slabs[q] = jobs[base + q].slabs; js.dw[q] = jobs[base + q].dw; js.nslab[q] = jobs[base + q].nslab;
    }
    hipLaunchKernelGGL(conv_halo_wgrad_reduce_multi_kernel, dim3(kTaps * 64 * 64 / 64, n), dim3(256), 0, st, js);
  }
  return hipGetLastError();
}

// nslab_out != null: the slab sum is LEFT to the caller (launch_conv_halo_wgrad_reduce_multi); *nslab_out = slabs written into `slabs`
hipError_t launch_conv_c64_bwd_pair(const float* dy, const float* U_bwd, const float* res, float* dx, const float* x, float* dw, float* slabs,
                                    int N, int H, int W, hipStream_t st, int* nslab_out) {
  if (!conv_c64_bwd_pair_supported(N, H, W)) return hipErrorInvalidValue;
  const float* zeros = conv_halo_zero_page(st);
  if (zeros == nullptr) return hipErrorNotReady;
  static const int pct = [] { const char* e = getenv("HIFIHR_C64_PAIR_DGRAD_PCT"); int v = e ? atoi(e) : 44; return v < 5 ? 5 : (v > 95 ? 95 : v); }();
  const int cus = halo_cus();
  int ga = (cus * pct + 50) / 100;
  if (ga < 1) ga = 1;
  if (ga > cus - 1) ga = cus - 1;
  int gb = cus - ga;
  Wino2Args d;
  d.src = dy; d.U = U_bwd; d.dst = dx; d.stats = nullptr; d.zeros = zeros; d.bias = nullptr; d.relu = 0; d.res = res;
  d.N = N; d.H = H; d.W = W;
  d.ctiles = (W + kTW - 1) / kTW;
  d.total = N * d.ctiles * H;
  d.per = (d.total + ga - 1) / ga;
  if (d.per < 4) d.per = 4;
  d.per = (d.per + 1) & ~1;
  ga = (d.total + d.per - 1) / d.per;
  HaloWgradArgs w;
  w.x = x; w.dy = dy; w.zeros = zeros; w.N = N; w.H = H; w.W = W;
  w.ctiles = W / kTW;
  w.total = N * w.ctiles * H;
  w.per = (w.total + gb - 1) / gb;
  if (w.per < 4) w.per = 4;
  gb = (w.total + w.per - 1) / w.per;
  w.slabs = slabs != nullptr ? slabs : halo_wgrad_scratch(st, (size_t)halo_cus() * kTaps * 64 * 64 * sizeof(float));
  if (w.slabs == nullptr) return hipErrorNotReady;
  if (res != nullptr) hipLaunchKernelGGL((conv_c64_bwd_pair_kernel<true>), dim3(ga + gb), dim3(256 + 64 * kNL), 0, st, d, ga, w);
  else hipLaunchKernelGGL((conv_c64_bwd_pair_kernel<false>), dim3(ga + gb), dim3(256 + 64 * kNL), 0, st, d, ga, w);
  if (nslab_out != nullptr) *nslab_out = gb;
  else hipLaunchKernelGGL(conv_halo_wgrad_reduce_kernel, dim3(kTaps * 64 * 64 / 64), dim3(256), 0, st, w.slabs, gb, dw);
  return hipGetLastError();
}

}  // namespace hifihr
