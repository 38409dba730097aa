// kmer.h — canonical k-mer extraction + invertible hash; __host__ __device__ so the same code is
// unit-tested on the CPU (tests/test_hostdev.py via rvn_test_* hooks) and used by the kernels.
//
// Packed layout (biosoup::NucleicAcid): base i at bits (2i mod 64) of word i/32.
// For the k-mer starting at base p let x = sum_j c[p+j] << 2j (bits read LSB-first from the stream).
// ram's rolling registers (minimizer_engine.cpp, Minimize(sequence)) are then
//   forward  = sum_j c[p+j] << 2(k-1-j)  = reverse the 2-bit groups of x
//   reverse  = sum_j (3-c[p+j]) << 2j    = ~x & mask
// so no rolling state is needed: every position is independent.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rvn {

__host__ __device__ __forceinline__ std::uint64_t bitrev64(std::uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brevll(x);
#else
  x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
  x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
  return (x >> 32) | (x << 32);
#endif
}

// Reverse the order of the k 2-bit groups held in the low 2k bits of x.
__host__ __device__ __forceinline__ std::uint64_t reverse_groups(std::uint64_t x, unsigned k) {
  std::uint64_t y = bitrev64(x);
  y = ((y & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((y & 0x5555555555555555ULL) << 1);
  return y >> (64 - 2 * k);
}

// ram's hash (Thomas Wang 64-bit mix masked to 2k bits).
__host__ __device__ __forceinline__ std::uint64_t hash64(std::uint64_t key, std::uint64_t mask) {
  key = ((~key) + (key << 21)) & mask;
  key = key ^ (key >> 24);
  key = ((key + (key << 3)) + (key << 8)) & mask;
  key = key ^ (key >> 14);
  key = ((key + (key << 2)) + (key << 4)) & mask;
  key = key ^ (key >> 28);
  key = (key + (key << 31)) & mask;
  return key;
}

// Same function in 32-bit arithmetic; exact whenever mask < 2^32 (2k <= 31 is what we use it for).
__host__ __device__ __forceinline__ std::uint32_t hash32(std::uint32_t key, std::uint32_t mask) {
  key = ((~key) + (key << 21)) & mask;
  key = key ^ (key >> 24);
  key = ((key + (key << 3)) + (key << 8)) & mask;
  key = key ^ (key >> 14);
  key = ((key + (key << 2)) + (key << 4)) & mask;
  key = key ^ (key >> 28);
  key = (key + (key << 31)) & mask;
  return key;
}

// 2k bits starting at bit offset `bit` of the little-endian word stream (w0 = word containing the
// first bit, w1 = the following word).
__host__ __device__ __forceinline__ std::uint64_t extract_bits(std::uint64_t w0, std::uint64_t w1, unsigned off,
                                                               std::uint64_t mask) {
  std::uint64_t x = w0 >> off;
  if (off) x |= w1 << (64 - off);
  return x & mask;
}

// Canonical hashed k-mer. Returns false for palindromes (forward == reverse, skipped by ram).
// *strand = 1 when the reverse complement is the smaller one.
template <typename V>
__host__ __device__ __forceinline__ bool canonical_hash(std::uint64_t x, unsigned k, std::uint64_t mask, V* value,
                                                        unsigned* strand) {
  const std::uint64_t fwd = reverse_groups(x, k);
  const std::uint64_t rev = (~x) & mask;
  if (fwd == rev) return false;
  const std::uint64_t m = fwd < rev ? fwd : rev;
  *strand = fwd < rev ? 0u : 1u;
  if (sizeof(V) == 4) {
    *value = static_cast<V>(hash32(static_cast<std::uint32_t>(m), static_cast<std::uint32_t>(mask)));
  } else {
    *value = static_cast<V>(hash64(m, mask));
  }
  return true;
}

}  // namespace rvn
