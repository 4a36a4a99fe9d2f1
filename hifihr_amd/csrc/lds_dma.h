// LDS-DMA / wave-specialisation idioms shared by the hand-written GEMM and convolution kernels (csrc/gemm.hip, csrc/conv_halo.hip).
// Not part of the ABI.  Under HIFIHR_HOSTSIM (tests/hostsim) every macro has an emulator form.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>

namespace hifihr {

#if defined(HIFIHR_HOSTSIM)
typedef hs_floatx4 floatx4;
// emulation of an LDS-DMA: lane `lane` of the wave copies 16 bytes to (wave-uniform base) + 16 * lane
#define HIFIHR_GLDS16(gptr, lds_wave_base, lane) std::memcpy(reinterpret_cast<char*>(lds_wave_base) + 16 * (lane), (gptr), 16)
#define HIFIHR_WAIT_LOADS() ((void)0)
#define HIFIHR_PIN() ((void)0)
#define HIFIHR_SCHED_GROUP(mask, n) ((void)0)
#else
// ask the scheduler for `n` instructions of class `mask` (0x008 MFMA, 0x100 LDS read, 0x002 VALU) next, inside the region that
// the surrounding HIFIHR_PIN()s delimit
#define HIFIHR_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define HIFIHR_GLDS16(gptr, lds_wave_base, lane)                                                     \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),           \
                                   (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0)
#define HIFIHR_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// MFMAs touch registers only, so the scheduler would otherwise sink the whole block below the wait + barrier that follow it
// (seen in the ISA: 1 MFMA, vmcnt(0), s_barrier, 127 MFMAs), exposing the load latency the block is there to hide
#define HIFIHR_PIN() __builtin_amdgcn_sched_barrier(0)
#endif

// XCD-aware workgroup renumbering: blocks b and b + 8 share an XCD (its own 4 MB L2), so give every XCD one contiguous eighth
// of the tile list (consecutive tiles share an operand panel).  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// Wave priority for the instruction arbiter of a SIMD that two waves share (s_setprio).  The loader waves of the wave-specialised kernels
// are the second-dispatched half of their 512-thread workgroup and lose the arbitration against the MFMA wave of their SIMD at equal
// priority (MI355X_MICROARCH.md, "Two waves per SIMD", item 4): their few LDS-DMA issues then start late and the MFMA waves wait at the
// stage barrier.  One static raise at the top of the loader branch: 5.79 -> 5.755 ms/step (priority 1 and 3 alike; -DHIFIHR_LOADER_PRIO=0
// for the A/B).
#ifndef HIFIHR_LOADER_PRIO
#define HIFIHR_LOADER_PRIO 1
#endif
#if defined(HIFIHR_HOSTSIM) || HIFIHR_LOADER_PRIO == 0
#define HIFIHR_SET_LOADER_PRIO() ((void)0)
#else
#define HIFIHR_SET_LOADER_PRIO() __builtin_amdgcn_s_setprio(HIFIHR_LOADER_PRIO)
#endif
#if defined(HIFIHR_HOSTSIM)
#define HIFIHR_RAW_BARRIER() __syncthreads()
#define HIFIHR_WAIT_VM(n) ((void)0)
#define HIFIHR_WAIT_LGKM0() ((void)0)
#define HIFIHR_TOUCH(x) ((void)0)
// the emulator runs the lanes of a wave one after the other between rendezvous points: a wave-level rendezvous where the hardware's
// lockstep execution is relied on (all lanes have stored before lane 0 raises a flag; all lanes have polled before it is lowered)
#define HIFIHR_WAVE_SYNC() ((void)__ballot(1))
#else
#define HIFIHR_WAVE_SYNC() ((void)0)
#define HIFIHR_TOUCH(x) asm volatile("" : "+v"(x))
#define HIFIHR_RAW_BARRIER()                     \
  do {                                           \
    asm volatile("" ::: "memory");               \
    __builtin_amdgcn_s_barrier();                \
    asm volatile("" ::: "memory");               \
  } while (0)
#define HIFIHR_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define HIFIHR_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

}  // namespace hifihr
