// TEST DOUBLE (tests/cpp only): the handful of biosoup::NucleicAcid members the facade touches, so the
// facade can be compiled and exercised here without the real biosoup (absent from this image).  A Raven
// build uses the real header; this file is never installed and is not part of the product.
#pragma once
#include <atomic>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace biosoup {
class NucleicAcid {
 public:
  NucleicAcid(const std::string& name_, const std::string& data) : id(num_objects++), name(name_), inflated_len(data.size()) {
    deflated_data.assign((data.size() + 31) / 32, 0);
    for (std::size_t i = 0; i < data.size(); ++i) {
      std::uint64_t c;
      switch (data[i]) {
        case 'A': case 'a': c = 0; break;
        case 'C': case 'c': c = 1; break;
        case 'G': case 'g': c = 2; break;
        case 'T': case 't': c = 3; break;
        default: throw std::invalid_argument("[test double] bad base");
      }
      deflated_data[i >> 5] |= c << ((i << 1) & 63);
    }
  }
  std::string InflateData(std::uint32_t i = 0, std::uint32_t len = 0xFFFFFFFFu) const {
    std::string out;
    if (i >= inflated_len) return out;
    len = len < inflated_len - i ? len : inflated_len - i;
    out.reserve(len);
    for (std::uint32_t p = i; p < i + len; ++p) out += "ACGT"[(deflated_data[p >> 5] >> ((p << 1) & 63)) & 3];
    return out;
  }
  static std::atomic<std::uint32_t> num_objects;
  std::uint32_t id;
  std::string name;
  std::vector<std::uint64_t> deflated_data;
  std::vector<std::uint8_t> block_quality;
  std::uint32_t inflated_len;
  bool is_reverse_complement = false;
};
}  // namespace biosoup
