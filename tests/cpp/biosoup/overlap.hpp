// TEST DOUBLE (tests/cpp only) of biosoup::Overlap — see nucleic_acid.hpp in this directory.
#pragma once
#include <cstdint>
#include <string>

namespace biosoup {
struct Overlap {
  Overlap() = default;  // construct.cc resizes vectors of overlaps
  Overlap(std::uint32_t lhs_id_, std::uint32_t lhs_begin_, std::uint32_t lhs_end_, std::uint32_t rhs_id_,
          std::uint32_t rhs_begin_, std::uint32_t rhs_end_, std::uint32_t score_, bool strand_ = true)
      : lhs_id(lhs_id_), lhs_begin(lhs_begin_), lhs_end(lhs_end_), rhs_id(rhs_id_), rhs_begin(rhs_begin_),
        rhs_end(rhs_end_), score(score_), strand(strand_) {}
  std::uint32_t lhs_id = 0, lhs_begin = 0, lhs_end = 0, rhs_id = 0, rhs_begin = 0, rhs_end = 0, score = 0;
  bool strand = true;
  std::string alignment;
};
}  // namespace biosoup
