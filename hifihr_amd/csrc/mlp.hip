// Small-batch fully connected layers of the regression heads: y = act(BN1d?(x W^T + b)), fp32.
// Replaces nn.Linear (+ nn.BatchNorm1d in training mode) (+ nn.ReLU) and their autograd in the reference's HandEncoder /
// LightEstimator heads (reference network/res_encoder.py:52-145 base_layers, *_reg, :150-210 light_reg).  At B = 32 these
// layers are <= 34 MFLOP each: rocBLAS + ATen spent 8-9 launches of ~5 us per layer per direction on them (GEMM, bias,
// clamp, four batch-norm kernels, two AccumulateGrad adds); here a layer is ONE launch forward and TWO backward:
//   linear_fwd_kernel     workgroup = 16 output features x all rows; x / W staged through LDS in chunks of up to 512 input
//                         features with every global load of a chunk in flight at once (the layers are latency-bound:
//                         round 1's first version exposed one HBM latency per 64-feature chunk and ran 15-30 us); bias,
//                         train-mode BatchNorm1d (batch statistics are local: the workgroup owns its features for every
//                         row) and ReLU in the epilogue.
//   linear_bwd_w_kernel   same ownership: activation mask + BatchNorm backward per feature, db / dgamma / dbeta, then
//                         dW[16][I] += dz^T x (dz held in registers, x in LDS) accumulated STRAIGHT into the (flat)
//                         gradient buffer; also zero-fills dx.
//   linear_bwd_x_kernel   dx = dz W, split over the output features (grid.y) with fp32 atomics into the zeroed dx; W is
//                         fetched in batches of 16 independent loads.
// The FLOP count is irrelevant here (fp32 MFMA would buy 2x on microseconds of math); launch count, loads in flight and
// >= 64 workgroups per launch are what these kernels are shaped for.
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"
#include "lds_dma.h"

namespace hifihr {

constexpr int kFT = 16;       // output features per workgroup

// 4 consecutive floats of a row of length `len` starting at column c4 (zero past the end); rows are only 16-byte aligned
// when the row length is a multiple of 4 (the squeeze-excite layers have 6, 10, 34, 58 ... input features)
__device__ __forceinline__ float4 load4_guarded(const float* __restrict__ row, int c4, int len, bool aligned) {
  if (aligned) return *reinterpret_cast<const float4*>(row + c4);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < len) v.x = row[c4];
  if (c4 + 1 < len) v.y = row[c4 + 1];
  if (c4 + 2 < len) v.z = row[c4 + 2];
  if (c4 + 3 < len) v.w = row[c4 + 3];
  return v;
}

template <int RB>
struct MlpCfg {
  static constexpr int IC = (RB == 32) ? 512 : 256;   // input features per LDS chunk
  static constexpr int LD = IC + 4;                   // LDS row stride: float4 reads of 16 different rows hit 64 banks
  static constexpr int XL = RB * IC / 1024;           // float4 loads of x per thread per chunk
  static constexpr int WL = kFT * IC / 1024;          // float4 loads of W per thread per chunk
  static constexpr size_t lds_bytes = (size_t)(RB + kFT) * LD * sizeof(float);
};

template <int RB>
__device__ __forceinline__ void linear_fwd_body(const LinearArgs& a, const int bx, const int by) {
  using C = MlpCfg<RB>;
  constexpr int KR = RB / 16;                     // rows per thread
  HIP_DYNAMIC_SHARED(float, smem);
  float* xs = smem;                               // [RB][LD]
  float* ws = smem + RB * C::LD;                  // [kFT][LD]
  __shared__ float zs[RB][kFT + 1];
  __shared__ float sc[kFT], sh[kFT];
  const int tid = threadIdx.x, ol = tid & 15, rg = tid >> 4;
  const int o0 = bx * kFT, row0 = by * RB;
  const int nrow = min(RB, a.B - row0);
  float acc[KR];
  // Round 4: the products on the matrix cores.  The FMA loop read 12 KB of LDS per 4 096 flops (one float4 of W and KR of x per 4 KR
  // fused multiply-adds per thread): a workgroup was bound by its LDS bandwidth -- 20 us for the 1024 -> 512 layer on 32 workgroups.
  // v_mfma_f32_16x16x4_f32 takes the same operands at 1 / 12 of the LDS traffic: wave w multiplies the chunk's k-groups w, w + 4, ..
  // (16 k-values each: one ds_read_b128 of W and RB / 16 of x feed 4 x RB / 16 MFMAs); the four waves' partial sums meet in LDS in a
  // fixed order, and the thread -> (feature, rows) ownership of the epilogue below is unchanged.
  constexpr int NRB = RB / 16;                    // 16-row blocks
  __shared__ float part[4][RB][kFT + 1];
  const int lane = tid & 63, mw = tid >> 6, mr = lane & 15, mg = lane >> 4;
  floatx4 macc[NRB];
#pragma unroll
  for (int j = 0; j < NRB; ++j) macc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
  constexpr int CPR = C::IC / 4;                  // float4 per LDS row
  const bool al4 = (a.I & 3) == 0;
  for (int i0 = 0; i0 < a.I; i0 += C::IC) {
    const int ilen = min(C::IC, a.I - i0);
    float4 xr[C::XL], wr[C::WL];
#pragma unroll
    for (int p = 0; p < C::XL; ++p) {             // all loads of the chunk first ...
      const int e = tid + 256 * p, r = e / CPR, c4 = (e % CPR) * 4;
      xr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < nrow && c4 < ilen) xr[p] = load4_guarded(a.x + (size_t)(row0 + r) * a.I + i0, c4, ilen, al4);
    }
#pragma unroll
    for (int p = 0; p < C::WL; ++p) {
      const int e = tid + 256 * p, r = e / CPR, c4 = (e % CPR) * 4;
      wr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (o0 + r < a.O && c4 < ilen) wr[p] = load4_guarded(a.W + (size_t)(o0 + r) * a.I + i0, c4, ilen, al4);
    }
#pragma unroll
    for (int p = 0; p < C::XL; ++p) {             // ... then the LDS stores
      const int e = tid + 256 * p, r = e / CPR, c4 = (e % CPR) * 4;
      *reinterpret_cast<float4*>(&xs[r * C::LD + c4]) = xr[p];
    }
#pragma unroll
    for (int p = 0; p < C::WL; ++p) {
      const int e = tid + 256 * p, r = e / CPR, c4 = (e % CPR) * 4;
      *reinterpret_cast<float4*>(&ws[r * C::LD + c4]) = wr[p];
    }
    __syncthreads();
    // (rows >= nrow, features >= O and columns >= ilen of the chunk hold zeros in LDS: whole 16-wide k-groups, no masks)
    const int nkq = (ilen + 15) >> 4;
#pragma unroll 2
    for (int kq = mw; kq < nkq; kq += 4) {
      const float4 w4 = *reinterpret_cast<const float4*>(&ws[mr * C::LD + 16 * kq + 4 * mg]);
      float4 x4[NRB];
#pragma unroll
      for (int j = 0; j < NRB; ++j) x4[j] = *reinterpret_cast<const float4*>(&xs[(16 * j + mr) * C::LD + 16 * kq + 4 * mg]);
      const float wc[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
          const float xc = c == 0 ? x4[j].x : (c == 1 ? x4[j].y : (c == 2 ? x4[j].z : x4[j].w));
          macc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[c], xc, macc[j], 0, 0, 0);      // D[feature 4 mg + e][row 16 j + mr]
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NRB; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) part[mw][16 * j + mr][4 * mg + e] = macc[j][e];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KR; ++k) acc[k] = (part[0][rg + 16 * k][ol] + part[1][rg + 16 * k][ol]) + (part[2][rg + 16 * k][ol] + part[3][rg + 16 * k][ol]);
  const int o = o0 + ol;
  const bool ook = o < a.O;
  const float bias = (a.b != nullptr && ook) ? a.b[o] : 0.f;
#pragma unroll
  for (int k = 0; k < KR; ++k) acc[k] += bias;
  if (a.gamma != nullptr) {                       // train-mode BatchNorm1d over the B rows (all of them in this workgroup)
#pragma unroll
    for (int k = 0; k < KR; ++k) zs[rg + 16 * k][ol] = acc[k];
    __syncthreads();
    if (tid < kFT && o0 + tid < a.O) {
      float s = 0.f;
      for (int r = 0; r < nrow; ++r) s += zs[r][tid];
      const float mu = s / (float)nrow;
      float q = 0.f;
      for (int r = 0; r < nrow; ++r) { const float d = zs[r][tid] - mu; q += d * d; }
      const float var = q / (float)nrow;
      const float is = 1.0f / sqrtf(var + a.eps);
      const int oo = o0 + tid;
      sc[tid] = is * a.gamma[oo];
      sh[tid] = a.beta[oo] - mu * sc[tid];
      a.save_mean[oo] = mu;
      a.save_invstd[oo] = is;
      if (a.running_mean != nullptr) {
        a.running_mean[oo] = (1.f - a.momentum) * a.running_mean[oo] + a.momentum * mu;
        const float unb = nrow > 1 ? var * ((float)nrow / (float)(nrow - 1)) : var;
        a.running_var[oo] = (1.f - a.momentum) * a.running_var[oo] + a.momentum * unb;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int r = rg + 16 * k;
      if (r < nrow && ook) a.z[(size_t)(row0 + r) * a.O + o] = acc[k];
      acc[k] = acc[k] * sc[ol] + sh[ol];
    }
  }
#pragma unroll
  for (int k = 0; k < KR; ++k) {
    const int r = rg + 16 * k;
    if (r < nrow && ook) {
      const size_t off = (size_t)(row0 + r) * a.O + o;
      float v = acc[k];
      if (a.act == 1) v = fmaxf(v, 0.f);
      else if (a.act == 2) { if (a.gamma == nullptr) a.z[off] = v; v = v * fast_sigmoid(v); }      // swish keeps z for the backward
      else if (a.act == 3) v = fast_sigmoid(v);
      a.y[off] = v;
    }
  }
}

// per 16 output features: dz = BN'(dy * act'(y)); db, dgamma, dbeta, dW += dz^T x; writes dz[B][O]; zero-fills dx
constexpr int kWC = 128;      // input features per slice (grid.y) of the dW phase: 32 rows x kWC of x in LDS.  (512 until round 4: 64 workgroups on the
                              // 1024 x 512 layer; the matrix-core form has nothing to amortise over a wide slice, and 256 workgroups cover the chip)
constexpr int kWLD = kWC + 4;
constexpr int kRowsMax = 64;  // rows per pass (batch-norm layers: one pass)

__device__ __forceinline__ void linear_bwd_w_body(const LinearArgs& a, const LinearGrads& g, const int bx, const int by, const int ngx,
                                                  const int ngy) {
  HIP_DYNAMIC_SHARED(float, xs);                  // [32][kWLD]
  __shared__ float dzs[kRowsMax][kFT + 1];
  __shared__ float red[2][kFT];
  const int tid = threadIdx.x, ol = tid & 15, rg = tid >> 4;
  const int o0 = bx * kFT;
  const int o = o0 + ol;
  const bool ook = o < a.O;
  // blockIdx.y = 512-wide slice of the input features: the squeeze-excite "reduce" layers have 6-96 outputs and up to 2304
  // inputs, i.e. 1-6 feature blocks only; every slice recomputes the (tiny) dz part and slice 0 alone accumulates db / dgamma /
  // dbeta and writes dz.
  const bool first_slice = by == 0;
  // the dx zero fill rides along (linear_bwd_x_kernel adds into it): each workgroup clears an equal slice
  const bool al4 = (a.I & 3) == 0;
  if (g.dx != nullptr) {
    const unsigned nwg = ngx * ngy, wgi = by * ngx + bx;
    if (al4) {
      const size_t n4 = (size_t)a.B * a.I / 4, per = (n4 + nwg - 1) / nwg;
      const size_t lo = per * wgi, hi = lo + per < n4 ? lo + per : n4;
      for (size_t i = lo + tid; i < hi; i += 256) reinterpret_cast<float4*>(g.dx)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const size_t n = (size_t)a.B * a.I, per = (n + nwg - 1) / nwg;
      const size_t lo = per * wgi, hi = lo + per < n ? lo + per : n;
      for (size_t i = lo + tid; i < hi; i += 256) g.dx[i] = 0.f;
    }
  }
  const bool bn = a.gamma != nullptr;
  for (int row0 = 0; row0 < a.B; row0 += kRowsMax) {     // batch-norm layers have B <= 64 (checked by the launcher): one pass
    const int nrow = min(kRowsMax, a.B - row0);
    float gv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = rg + 16 * k;
      float v = 0.f;
      if (r < nrow && ook) {
        const size_t off = (size_t)(row0 + r) * a.O + o;
        v = g.dy[off];
        if (a.act == 1) { if (!(a.y[off] > 0.f)) v = 0.f; }
        else if (a.act == 2) { const float z = a.z[off], sg = fast_sigmoid(z); v *= sg * (1.f + z * (1.f - sg)); }
        else if (a.act == 3) { const float yy = a.y[off]; v *= yy * (1.f - yy); }
      }
      gv[k] = v;
    }
    if (bn) {
      float xh[4];                                   // xhat of this thread's elements (loads issued before the reduction)
      if (ook) {
        const float mu = a.save_mean[o], is = a.save_invstd[o];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = rg + 16 * k;
          xh[k] = (r < nrow) ? (a.z[(size_t)(row0 + r) * a.O + o] - mu) * is : 0.f;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) xh[k] = 0.f;
      }
      // per-feature sums over the rows: 16 row groups x 4 rows each -> LDS -> 16 threads
      float sg = 0.f, sgx = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { sg += gv[k]; sgx += gv[k] * xh[k]; }
      dzs[rg][ol] = sg; dzs[16 + rg][ol] = sgx;
      __syncthreads();
      if (tid < kFT) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0 += dzs[r][tid]; s1 += dzs[16 + r][tid]; }
        red[0][tid] = s0 / (float)nrow;
        red[1][tid] = s1 / (float)nrow;
        if (o0 + tid < a.O && first_slice) {
          if (g.dgamma_acc) g.dgamma_acc[o0 + tid] += s1;
          if (g.dbeta_acc) g.dbeta_acc[o0 + tid] += s0;
        }
      }
      __syncthreads();
      if (ook) {
        const float k1 = a.gamma[o] * a.save_invstd[o];
#pragma unroll
        for (int k = 0; k < 4; ++k) gv[k] = (rg + 16 * k < nrow) ? k1 * (gv[k] - red[0][ol] - xh[k] * red[1][ol]) : 0.f;
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = rg + 16 * k;
      dzs[r][ol] = gv[k];
      if (r < nrow && ook && g.dz != nullptr && first_slice) g.dz[(size_t)(row0 + r) * a.O + o] = gv[k];
    }
    __syncthreads();
    if (tid < kFT && o0 + tid < a.O && g.db_acc != nullptr && first_slice) {
      float s = 0.f;
      for (int r = 0; r < nrow; ++r) s += dzs[r][tid];
      g.db_acc[o0 + tid] += s;
    }
    if (g.dW_acc == nullptr) { __syncthreads(); continue; }
    // dW[o0 + f][i] += sum_r dz[r][f] * x[r][i] on the matrix cores (round 4; the FMA form read one float of x from LDS per four
    // multiply-adds: LDS-bound, 23 us for a 1024 x 512 layer).  Wave w owns the inputs 16 NT w .. of the slice as NT 16-input
    // column blocks; an MFMA k-step is four batch rows: A = dz[row 4 kk + mg][feature mr] (8 registers per 32-row group), B = x[row 4 kk
    // + mg][input 16 t + mr] from LDS; D register e of lane (mr, mg) = dW[feature 4 mg + e][input 16 t + mr].
    const int lane = tid & 63, mw = tid >> 6, mr = lane & 15, mg = lane >> 4;
    {
      const int i0 = by * kWC;
      const int ilen = min(kWC, a.I - i0);
      constexpr int NT = kWC / 64;                   // column blocks per wave
      floatx4 wacc[NT];
#pragma unroll
      for (int c = 0; c < NT; ++c) wacc[c] = floatx4{0.f, 0.f, 0.f, 0.f};
      for (int rh = 0; rh < nrow; rh += 32) {        // 32 rows of x at a time through LDS, their dz fragments in registers
        const int nr = min(32, nrow - rh);
        constexpr int kWP = kWC / 32;                // float4 of the 32 x kWC image per thread
        float4 xr[kWP];
#pragma unroll
        for (int p = 0; p < kWP; ++p) {
          const int e = tid + 256 * p, r = e / (kWC / 4), c4 = (e % (kWC / 4)) * 4;
          xr[p] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (r < nr && c4 < ilen) xr[p] = load4_guarded(a.x + (size_t)(row0 + rh + r) * a.I + i0, c4, ilen, al4);
        }
#pragma unroll
        for (int p = 0; p < kWP; ++p) {
          const int e = tid + 256 * p, r = e / (kWC / 4), c4 = (e % (kWC / 4)) * 4;
          *reinterpret_cast<float4*>(&xs[r * kWLD + c4]) = xr[p];
        }
        float dzf[8];                                // (rows >= nrow of dzs hold zeros; rh + 31 < kRowsMax)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) dzf[kk] = dzs[rh + 4 * kk + mg][mr];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NT; ++c) {
          const int t = NT * mw + c;
          if (16 * t < ilen) {                       // (uniform per wave)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
              wacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(dzf[kk], xs[(4 * kk + mg) * kWLD + 16 * t + mr], wacc[c], 0, 0, 0);
          }
        }
        __syncthreads();
      }
#pragma unroll
      for (int c = 0; c < NT; ++c) {
        const int i = i0 + 16 * (NT * mw + c) + mr;
        if (i < a.I) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int oo = o0 + 4 * mg + e;
            if (oo < a.O) g.dW_acc[(size_t)oo * a.I + i] += wacc[c][e];
          }
        }
      }
    }
  }
}

constexpr int kOS = 64;       // output features per split of the dx kernel

// dx[b][i] += sum_{o in split} dz[b][o] * W[o][i];  workgroup = 64 input features x one split of 64 output features.
// Round 4: on the matrix cores (the FMA form read one float of dz from LDS per multiply-add: LDS-bound, 15 us per base layer).  Wave w owns
// the 16 inputs 16 w .. of the workgroup and all (<= 4) 16-row blocks.  k = output feature, in groups of 16: lane (mr, mg) holds
// dz[row 16 j + mr][16 q + 4 mg + c] (one ds_read_b128: c = 0..3 are four MFMAs) and W[os0 + 16 q + 4 mg + c][i] (four coalesced dword loads,
// all 16 of a split in flight); D register e of lane (mr, mg) = dx[row 16 j + 4 mg + e][input 16 w + mr].
__device__ __forceinline__ void linear_bwd_x_body(const LinearArgs& a, const LinearGrads& g, const int bx, const int by) {
  constexpr int kDL = kOS + 4;                    // row stride of the dz image: float4 reads of 16 rows spread over the banks
  __shared__ __attribute__((aligned(16))) float dzs[kRowsMax][kDL];
  const int tid = threadIdx.x, lane = tid & 63, mw = tid >> 6, mr = lane & 15, mg = lane >> 4;
  const int i = bx * 64 + 16 * mw + mr;
  const int ic = i < a.I ? i : a.I - 1;           // clamped: the loads below stay unconditional
  const int os0 = by * kOS, on = min(kOS, a.O - os0);
  float wv[kOS / 16][4];                          // W[os0 + 16 q + 4 mg + c][i]; rows past `on` re-read the last row (dz is 0 there)
#pragma unroll
  for (int q = 0; q < kOS / 16; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int o = os0 + 16 * q + 4 * mg + c;
      wv[q][c] = a.W[(size_t)(o < a.O ? o : a.O - 1) * a.I + ic];
    }
  for (int row0 = 0; row0 < a.B; row0 += kRowsMax) {
    const int nrow = min(kRowsMax, a.B - row0);
    for (int e = tid; e < kRowsMax * kOS; e += 256) {
      const int r = e / kOS, c = e - r * kOS;
      dzs[r][c] = (r < nrow && c < on) ? g.dz[(size_t)(row0 + r) * a.O + os0 + c] : 0.f;
    }
    __syncthreads();
    constexpr int NRB = kRowsMax / 16;
    floatx4 acc[NRB];
#pragma unroll
    for (int j = 0; j < NRB; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < kOS / 16; ++q) {
      if (16 * q < on) {                          // (uniform)
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
          if (16 * j < nrow) {                    // (uniform)
            const float4 d4 = *reinterpret_cast<const float4*>(&dzs[16 * j + mr][16 * q + 4 * mg]);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4.x, wv[q][0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4.y, wv[q][1], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4.z, wv[q][2], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(d4.w, wv[q][3], acc[j], 0, 0, 0);
          }
        }
      }
    }
    if (i < a.I) {
#pragma unroll
      for (int j = 0; j < NRB; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 16 * j + 4 * mg + e;
          if (r < nrow) atomicAdd(g.dx + (size_t)(row0 + r) * a.I + i, acc[j][e]);
        }
    }
    __syncthreads();
  }
}

template <int RB>
__global__ __launch_bounds__(256) void linear_fwd_kernel(LinearArgs a) { linear_fwd_body<RB>(a, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(LinearArgs a, LinearGrads g) {
  linear_bwd_w_body(a, g, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(LinearArgs a, LinearGrads g) { linear_bwd_x_body(a, g, blockIdx.x, blockIdx.y); }

// Grouped launches: the regression heads are five to six independent little MLPs of the same depth (reference
// network/res_encoder.py:112-131); each layer alone is a latency-bound ~8 us launch on 8-32 workgroups.  blockIdx.z picks the member,
// the grid covers the largest member and workgroups outside a member's own extent leave at once.
template <int RB>
__global__ __launch_bounds__(256) void linear_fwd_group_kernel(LinearGroup grp) {
  const LinearArgs& a = grp.a[blockIdx.z];
  if ((int)blockIdx.x * kFT >= a.O || (int)blockIdx.y * RB >= a.B) return;
  linear_fwd_body<RB>(a, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void linear_bwd_w_group_kernel(LinearGroup grp, LinearGradsGroup gg) {
  const LinearArgs& a = grp.a[blockIdx.z];
  const int ngx = (a.O + kFT - 1) / kFT, ngy = (a.I + kWC - 1) / kWC;
  if ((int)blockIdx.x >= ngx || (int)blockIdx.y >= ngy) return;
  linear_bwd_w_body(a, gg.g[blockIdx.z], blockIdx.x, blockIdx.y, ngx, ngy);
}
__global__ __launch_bounds__(256) void linear_bwd_x_group_kernel(LinearGroup grp, LinearGradsGroup gg) {
  const LinearArgs& a = grp.a[blockIdx.z];
  const LinearGrads& g = gg.g[blockIdx.z];
  if (g.dx == nullptr || (int)blockIdx.x * 64 >= a.I || (int)blockIdx.y * kOS >= a.O) return;
  linear_bwd_x_body(a, g, blockIdx.x, blockIdx.y);
}

template <int RB>
static hipError_t launch_fwd_rb(const LinearArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_fwd_kernel<RB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)MlpCfg<RB>::lds_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((linear_fwd_kernel<RB>), dim3((a.O + kFT - 1) / kFT, (a.B + RB - 1) / RB), dim3(256), MlpCfg<RB>::lds_bytes, st, a);
  return hipGetLastError();
}

hipError_t launch_linear_fwd(const LinearArgs& a, hipStream_t st) {
  if (a.gamma != nullptr && a.B > kRowsMax) return hipErrorInvalidValue;
  return a.B <= 32 ? launch_fwd_rb<32>(a, st) : launch_fwd_rb<64>(a, st);
}

hipError_t launch_linear_bwd(const LinearArgs& a, const LinearGrads& g, hipStream_t st) {
  if ((a.gamma != nullptr && a.B > kRowsMax) || (g.dx != nullptr && g.dz == nullptr)) return hipErrorInvalidValue;
  constexpr size_t lds = (size_t)32 * kWLD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bwd_w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((a.O + kFT - 1) / kFT, (a.I + kWC - 1) / kWC), dim3(256), lds, st, a, g);
  if (g.dx != nullptr)
    hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((a.I + 63) / 64, (a.O + kOS - 1) / kOS), dim3(256), 0, st, a, g);
  return hipGetLastError();
}

template <int RB>
static hipError_t launch_fwd_group_rb(const LinearGroup& grp, dim3 grid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_fwd_group_kernel<RB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)MlpCfg<RB>::lds_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((linear_fwd_group_kernel<RB>), grid, dim3(256), MlpCfg<RB>::lds_bytes, st, grp);
  return hipGetLastError();
}

hipError_t launch_linear_fwd_group(const LinearGroup& grp, hipStream_t st) {
  if (grp.n <= 0 || grp.n > kMaxLinearGroup) return hipErrorInvalidValue;
  int gx = 0, maxB = 0;
  for (int i = 0; i < grp.n; ++i) {
    if (grp.a[i].gamma != nullptr) return hipErrorInvalidValue;          // batch-norm members go through launch_linear_fwd
    gx = max(gx, (grp.a[i].O + kFT - 1) / kFT);
    maxB = max(maxB, grp.a[i].B);
  }
  if (maxB <= 32) return launch_fwd_group_rb<32>(grp, dim3(gx, (maxB + 31) / 32, grp.n), st);
  return launch_fwd_group_rb<64>(grp, dim3(gx, (maxB + 63) / 64, grp.n), st);
}

hipError_t launch_linear_bwd_group(const LinearGroup& grp, const LinearGradsGroup& gg, hipStream_t st) {
  if (grp.n <= 0 || grp.n > kMaxLinearGroup) return hipErrorInvalidValue;
  constexpr size_t lds = (size_t)32 * kWLD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bwd_w_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  int wx = 0, wy = 0, xx = 0, xy = 0;
  bool any_dx = false;
  for (int i = 0; i < grp.n; ++i) {
    const LinearArgs& a = grp.a[i];
    if (a.gamma != nullptr || (gg.g[i].dx != nullptr && gg.g[i].dz == nullptr)) return hipErrorInvalidValue;
    wx = max(wx, (a.O + kFT - 1) / kFT); wy = max(wy, (a.I + kWC - 1) / kWC);
    if (gg.g[i].dx != nullptr) { any_dx = true; xx = max(xx, (a.I + 63) / 64); xy = max(xy, (a.O + kOS - 1) / kOS); }
  }
  hipLaunchKernelGGL(linear_bwd_w_group_kernel, dim3(wx, wy, grp.n), dim3(256), lds, st, grp, gg);
  if (any_dx) hipLaunchKernelGGL(linear_bwd_x_group_kernel, dim3(xx, xy, grp.n), dim3(256), 0, st, grp, gg);
  return hipGetLastError();
}

}  // namespace hifihr
