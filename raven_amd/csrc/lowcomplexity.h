// lowcomplexity.h — the low-complexity k-mer filter of raven::Pile::AddKmers (RavenLib/src/pile.cc:73-117),
// __host__ __device__ so the same code is unit-tested on the CPU and used by the kernel.
//
// pile.cc works on std::string / std::vector<std::string>: (1) collapse runs of equal bases (std::unique on
// 1-char strings); (2) split the result into consecutive pairs starting at index 0 (a trailing single stays
// alone), drop consecutive equal pairs; (3) the same with pairs starting at index 1 ([c0], [c1 c2], [c3 c4],
// ...).  After every stage the k-mer is rejected when fewer than kmer_len / 2 + 1 characters are left.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rvn {

// collapse consecutive equal groups; groups are given by their start offsets into `in` (group g covers
// [start(g), start(g+1)) ).  Helper for stages 2 and 3: pair groups beginning at `first_pair` (0 or 1).
__host__ __device__ inline std::uint32_t lc_collapse_pairs(const std::uint8_t* in, std::uint32_t n,
                                                           std::uint32_t first_pair, std::uint8_t* out) {
  // group boundaries: [0, first_pair) as a single leading group of length first_pair (0 or 1), then pairs
  std::uint32_t out_n = 0;
  std::uint32_t prev_b = 0, prev_len = 0;
  bool have_prev = false;
  std::uint32_t i = 0;
  while (i < n) {
    std::uint32_t len;
    if (i < first_pair) len = first_pair - i;  // leading single (stage 3)
    else len = (n - i >= 2) ? 2 : 1;
    bool equal = have_prev && prev_len == len;
    if (equal)
      for (std::uint32_t t = 0; t < len; ++t) equal = equal && in[prev_b + t] == in[i + t];
    if (!equal) {
      for (std::uint32_t t = 0; t < len; ++t) out[out_n++] = in[i + t];
      prev_b = i;  // std::unique compares with the last KEPT element; equal groups are identical, so either works
      prev_len = len;
      have_prev = true;
    }
    i += len;
  }
  return out_n;
}

// true when the k-mer (codes[0..k)) survives the three stages, i.e. AddKmers marks its pile cell
__host__ __device__ inline bool lc_kmer_passes(const std::uint8_t* codes, std::uint32_t k) {
  const std::uint32_t need = k / 2 + 1;
  std::uint8_t a[32], b[32];
  std::uint32_t n = 0;
  for (std::uint32_t i = 0; i < k; ++i)
    if (i == 0 || codes[i] != codes[i - 1]) a[n++] = codes[i];
  if (n < need) return false;
  n = lc_collapse_pairs(a, n, 0, b);
  if (n < need) return false;
  n = lc_collapse_pairs(b, n, 1, a);
  if (n < need) return false;
  return true;
}

}  // namespace rvn
