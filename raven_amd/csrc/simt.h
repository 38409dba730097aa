// simt.h — the cross-lane vocabulary of the kernels that are written once and run twice: on gfx950 (every function here
// is one or two CDNA4 instructions) and on the host under a 64-fibre wavefront emulator (simt_emu.cpp), which is how
// the CPU test-suite steps through a whole kernel without a GPU (tests/test_poa3_emulation.py).
//
// Rules for code written against sv:: (they are what makes the emulation faithful):
//   * every sv:: call that moves data between lanes (ballot, bperm, rl, rfl, row_shr, any, wave_sum, sync) is reached by
//     ALL 64 lanes of the wave, the same number of times, in the same order — no cross-lane call under a
//     lane-dependent branch.  The emulator checks the call sites agree and aborts with both source lines otherwise.
//   * a value another lane wrote to LDS or to global memory is read only after an sv::sync() (on the GPU: wave barrier +
//     memory fence of the workgroup; in the emulator: every fibre reaches the sync before any continues).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rvn {
namespace simt_emu {  // host side, simt_emu.cpp
int lane();
int exchange(int v, int src_lane, int site);  // all 64 fibres push v, each gets the value of src_lane (mod 64)
unsigned long long ballot(bool p, int site);
void sync(int site);
// runs fn(arg) once per lane of ONE wave; returns when all 64 fibres returned
void run_wave(void (*fn)(void*), void* arg);
}  // namespace simt_emu

namespace sv {

#if defined(__HIP_DEVICE_COMPILE__)
#define RVN_SV_SITE
#define RVN_SV_SITE_ARG
__device__ __forceinline__ int lane() { return static_cast<int>(threadIdx.x & 63); }
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int bperm(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ int rl(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// v with lane `dst_lane` (wave-uniform) replaced by `value` (wave-uniform): v_writelane_b32
// (this hipcc has no __builtin_amdgcn_writelane: the LLVM intrinsic by its name; the backend puts the lane select into M0)
__device__ int rvn_llvm_writelane(int value, int dst_lane, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ int wl(int v, int value, int dst_lane) { return rvn_llvm_writelane(value, dst_lane, v); }
__device__ __forceinline__ void sync() {
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
}
// value of the lane N below in the same row of 16 lanes; the first N lanes of a row get `fill`
template <int N>
__device__ __forceinline__ int row_shr(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, 0x110 + N, 0xf, 0xf, false);
}
// between two phases of ONE kernel that hand data to each other through global memory: the wave's stores and atomics have
// been issued to the CU's L1 / the XCD's L2 in order, which is all a reader on the SAME CU needs (workgroup scope: the
// vector L1 is write-through and shared by the CU).  NOT __threadfence(): an agent-scope release on gfx950 writes the
// XCD's whole L2 back (the L2s of the eight XCDs are not coherent with each other) — measured: two of those per window
// and layer round doubled the window-consensus stage (profiles/r05: 918 -> 1771 ms per C4 round).
__device__ __forceinline__ void phase_fence() {
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ unsigned atomic_add(unsigned* p, unsigned v) { return atomicAdd(p, v); }
__device__ __forceinline__ unsigned long long atomic_add(unsigned long long* p, unsigned long long v) { return atomicAdd(p, v); }
__device__ __forceinline__ unsigned long long clock() { return __builtin_readcyclecounter(); }
#else
inline int lane() { return simt_emu::lane(); }
inline unsigned long long ballot(bool p, int site = __builtin_LINE()) { return simt_emu::ballot(p, site); }
inline int bperm(int v, int src_lane, int site = __builtin_LINE()) { return simt_emu::exchange(v, src_lane & 63, site); }
inline int rl(int v, int src_lane, int site = __builtin_LINE()) { return simt_emu::exchange(v, src_lane & 63, site); }
inline int rfl(int v, int site = __builtin_LINE()) { return simt_emu::exchange(v, 0, site); }
inline int wl(int v, int value, int dst_lane) { return simt_emu::lane() == (dst_lane & 63) ? value : v; }
inline void sync(int site = __builtin_LINE()) { simt_emu::sync(site); }
inline void phase_fence(int site = __builtin_LINE()) { simt_emu::sync(site); }
template <int N>
inline int row_shr(int v, int fill, int site = __builtin_LINE()) {
  const int l = simt_emu::lane();
  const int got = simt_emu::exchange(v, (l - N) & 63, site);
  return (l & 15) >= N ? got : fill;
}
inline unsigned atomic_add(unsigned* p, unsigned v) {
  const unsigned o = *p;
  *p = o + v;
  return o;
}
inline unsigned long long atomic_add(unsigned long long* p, unsigned long long v) {
  const unsigned long long o = *p;
  *p = o + v;
  return o;
}
inline unsigned long long clock() { return 0; }
#endif

__host__ __device__ inline bool any(bool p) { return ballot(p) != 0; }
__host__ __device__ inline unsigned wave_sum(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += static_cast<unsigned>(bperm(static_cast<int>(v), lane() ^ off));
  return v;
}
__host__ __device__ inline int wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = bperm(v, lane() ^ off);
    v = o > v ? o : v;
  }
  return v;
}

}  // namespace sv
}  // namespace rvn
