// poa_oracle.h — shared declarations of the POA oracle (TEST INFRASTRUCTURE ONLY, see raven_oracle.cpp header)
#pragma once
#include <cstdint>
#include <vector>

namespace poa {
struct Layer {
  const std::uint8_t* codes;
  const std::uint8_t* qual;  // Phred+33 characters or nullptr
  std::uint32_t len, begin, end;
};
bool WindowConsensus(const std::vector<Layer>& layers, std::int8_t m, std::int8_t n, std::int8_t g, bool trim,
                     std::vector<std::uint8_t>* consensus, std::vector<std::uint32_t>* coverages_out,
                     bool device_order = false);
}  // namespace poa
