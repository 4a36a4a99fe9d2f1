// Folding the batch-norm slot partials inside a consumer kernel (csrc/bn.hip's apply kernels, csrc/wino4_bn.hip's fused input transform):
// mean / variance from the fp64 forward slots, the float backward sums, the arrival-counter election of the workgroup that hands the
// slots back zeroed.  Device code shared between translation units; not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "hifihr_internal.h"

namespace hifihr {

// True in every thread of exactly ONE workgroup of the launch: the last one to get here.  Called after the workgroup's last
// read of the slot buffer; the elected workgroup may then overwrite it.  32 first-level counters keep the same-address
// atomic traffic at <= grid / 32 per counter.
__device__ __forceinline__ bool last_workgroup(unsigned* __restrict__ cnt) {
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned slot = blockIdx.x & 31u, nslots = gridDim.x < 32u ? gridDim.x : 32u;
    const unsigned in_slot = (gridDim.x + 31u - slot) / 32u;
    int last = 0;
    if (atomicAdd(cnt + slot, 1u) == in_slot - 1u) last = (atomicAdd(cnt + 32, 1u) == nslots - 1u) ? 1 : 0;
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

__device__ __forceinline__ void clear_slots(float* __restrict__ buf, int C, unsigned* __restrict__ cnt) {
  float4* p = reinterpret_cast<float4*>(buf);
  const int n4 = stat_slots_used(C) * 2 * C / 4;         // (the other slots were never written)
  for (int i = threadIdx.x; i < n4; i += 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (threadIdx.x < 33) cnt[threadIdx.x] = 0u;
}

// both statistics of channel c with all loads in flight at once: this fold is the prologue of every workgroup of an apply kernel,
// and in groups of 8 dependent-latency batches it cost ~5 us of the ~9 us a small layer's launch takes (tools/time_bn.py).  Only the
// slots the layer's producers use are read (stat_slots_used: 8 / 16 / 32 by channel count).
template <int NS>
__device__ __forceinline__ void slot_sum2_n(const float* __restrict__ buf, int C, int c, float& s0, float& s1) {
  float a[NS], b[NS];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) { a[sl] = buf[(size_t)sl * 2 * C + c]; b[sl] = buf[(size_t)sl * 2 * C + C + c]; }
  s0 = 0.f; s1 = 0.f;
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) { s0 += a[sl]; s1 += b[sl]; }
}
// One case per slot count, each behind a compiler barrier: without it the optimiser hoists the loads of the slots all cases share (0 .. 7)
// in front of the branch and the 32-slot case -- the 64- / 128-channel layers -- waits for them before it issues the other 48: two to
// three memory latencies instead of one, +2.3 us per apply launch (seen in the ISA: 16 + 16 + 48 loads per wait instead of 64).
#if defined(HIFIHR_HOSTSIM)
#define HIFIHR_NO_HOIST() ((void)0)
#else
#define HIFIHR_NO_HOIST() asm volatile("" ::: "memory")
#endif
__device__ __forceinline__ void slot_sum2(const float* __restrict__ buf, int C, int c, float& s0, float& s1) {
  const int ns = stat_slots_used(C);                      // (uniform)
  if (ns == kStatSlots) { HIFIHR_NO_HOIST(); slot_sum2_n<kStatSlots>(buf, C, c, s0, s1); }
  else if (ns == 16) { HIFIHR_NO_HOIST(); slot_sum2_n<16>(buf, C, c, s0, s1); }
  else { HIFIHR_NO_HOIST(); slot_sum2_n<8>(buf, C, c, s0, s1); }
}

// FORWARD statistics (double slots): mean and biased variance of channel c from the slot partials, all loads in flight
template <int NS>
__device__ __forceinline__ void slot_sums_fwd_n(const double* __restrict__ buf, int C, int c, double& s0, double& s1) {
  double a[NS], b[NS];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) { a[sl] = buf[(size_t)sl * 2 * C + c]; b[sl] = buf[(size_t)sl * 2 * C + C + c]; }
  s0 = 0.0; s1 = 0.0;
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) { s0 += a[sl]; s1 += b[sl]; }
}
__device__ __forceinline__ void slot_mean_var(const float* __restrict__ stats, int C, int c, long M, float& mu, float& var) {
  const double* buf = reinterpret_cast<const double*>(stats);
  double s0, s1;
  const int ns = stat_slots_used(C);                      // (uniform)
  if (ns == kStatSlots) { HIFIHR_NO_HOIST(); slot_sums_fwd_n<kStatSlots>(buf, C, c, s0, s1); }
  else if (ns == 16) { HIFIHR_NO_HOIST(); slot_sums_fwd_n<16>(buf, C, c, s0, s1); }
  else { HIFIHR_NO_HOIST(); slot_sums_fwd_n<8>(buf, C, c, s0, s1); }
  const double m = s0 / (double)M;
  const double v = s1 / (double)M - m * m;           // fp64: the cancellation costs 2^-53 mean^2 / var
  mu = (float)m;
  var = v > 0.0 ? (float)v : 0.f;
}
__device__ __forceinline__ void clear_slots_fwd(float* __restrict__ stats, int C, unsigned* __restrict__ cnt) {
  float4* p = reinterpret_cast<float4*>(stats);
  const int n4 = stat_slots_used(C) * 2 * C / 2;       // doubles: 8 bytes each (the other slots were never written)
  for (int i = threadIdx.x; i < n4; i += 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (threadIdx.x < 33) cnt[threadIdx.x] = 0u;
}

}  // namespace hifihr
