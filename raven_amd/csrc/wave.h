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
