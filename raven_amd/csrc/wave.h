// wave.h — 64-lane wavefront helpers for gfx950 (CDNA4). Wave width is hard-coded 64.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rvn {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x & 63); }

__device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ULL << lane_id()) - 1ULL;
}

// Wave-wide inclusive scan (sum) via DPP-free shuffles.
template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T v) {
  const int lane = lane_id();
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    T o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  return v;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    T o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// Inclusive max-scan.
template <typename T>
__device__ __forceinline__ T wave_inclusive_max(T v) {
  const int lane = lane_id();
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    T o = __shfl_up(v, off, 64);
    if (lane >= off) v = o > v ? o : v;
  }
  return v;
}

// ---- DPP cross-lane primitives (single VALU instructions on gfx9-family, vs ~100-cycle ds_bpermute) ----
// value of lane-1 (whole-wave shift right by one); lane 0 receives `fill`
__device__ __forceinline__ int dpp_wave_shr1(int v, int fill) {
  return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// inclusive prefix max over the 64 lanes (the classic row_shr 1/2/3/4/8 + row_bcast 15/31 ladder)
__device__ __forceinline__ int wave_inclusive_max_dpp(int v, int identity) {
  int a = v;
  int t = __builtin_amdgcn_update_dpp(identity, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
  a = t > a ? t : a;
  t = __builtin_amdgcn_update_dpp(identity, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
  a = t > a ? t : a;
  t = __builtin_amdgcn_update_dpp(identity, v, 0x113 /* row_shr:3 */, 0xf, 0xf, false);
  a = t > a ? t : a;
  t = __builtin_amdgcn_update_dpp(identity, a, 0x114 /* row_shr:4 */, 0xf, 0xe, false);
  a = t > a ? t : a;
  t = __builtin_amdgcn_update_dpp(identity, a, 0x118 /* row_shr:8 */, 0xf, 0xc, false);
  a = t > a ? t : a;
  t = __builtin_amdgcn_update_dpp(identity, a, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
  a = t > a ? t : a;
  t = __builtin_amdgcn_update_dpp(identity, a, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
  a = t > a ? t : a;
  return a;
}

// The same prefix max with the DPP operand fused into v_max (one instruction per step instead of v_mov_dpp + v_max):
// a lane without a source (bound_ctrl off) is simply not written and keeps its value, which is the identity of max.
// All 64 lanes must be active.  s_nop 1 = the two wait states a DPP read needs after a VALU write of the register.
__device__ __forceinline__ int wave_inclusive_max_fused(int x) {
  asm volatile(
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
      : "+v"(x));
  return x;
}

// Mask of lanes (among `valid` lanes) whose 8-bit digit equals this lane's digit.
__device__ __forceinline__ unsigned long long match_digit8(unsigned d, bool valid) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// Block-wide exclusive scan of one value per thread (blockDim.x = 256, 4 waves).
// `smem` needs 4 entries. Returns exclusive prefix; *total gets the block sum.
template <typename T>
__device__ __forceinline__ T block_exclusive_sum_256(T v, T* smem, T* total) {
  const int lane = lane_id();
  const int w = threadIdx.x >> 6;
  T inc = wave_inclusive_sum(v);
  if (lane == 63) smem[w] = inc;
  __syncthreads();
  T base = 0;
  T tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    T s = smem[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

}  // namespace rvn
