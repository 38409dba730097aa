// TEST DOUBLES (tests/cpp only) of the parts of RAVEN ITSELF that stay Raven's when the engine is swapped in: the Pile
// members construct.cc touches between the two mapping passes and the overlap rules of overlap_utils.cc, re-stated for
// the stage program (construct_stage_test.cpp).  A Raven build uses its own pile.cc / overlap_utils.cc; these exist so
// that the reference's stage order can be exercised end to end through the facades without the reference tree.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

#include "biosoup/overlap.hpp"

namespace raven_double {

struct Pile {  // raven::Pile as far as construct.cc:123-315 uses it (cells = bases >> 4, kPSS = 4)
  Pile(std::uint32_t id_, std::uint32_t len) : id(id_), data(len >> 4, 0), begin_(0), end_(len >> 4) {}
  // hooks of include/raven_hip/find_overlaps.hpp
  void AdoptCoverage(const std::uint16_t* d, std::size_t n) { data.assign(d, d + n); }
  void AdoptAnnotation(std::uint32_t b, std::uint32_t e, std::uint16_t m, bool inv) {
    begin_ = b;
    end_ = e;
    median_ = m;
    if (inv) invalid = true;
  }
  void AdoptChimericRegions(const std::uint32_t* pairs, std::size_t n) {
    chimeric_regions.clear();
    for (std::size_t i = 0; i < n; ++i) chimeric_regions.emplace_back(pairs[2 * i], pairs[2 * i + 1]);
  }
  void AdoptKmers(const std::uint8_t* c, std::size_t n) { kmers.assign(c, c + n); }
  // pile.h accessors
  std::uint32_t begin() const { return begin_ << 4; }
  std::uint32_t end() const { return end_ << 4; }
  std::uint16_t median() const { return median_; }
  bool is_invalid() const { return invalid; }
  bool is_contained() const { return contained; }
  bool is_maybe_chimeric() const { return !chimeric_regions.empty(); }
  void set_is_invalid() { invalid = true; }
  void set_is_contained() { contained = true; }
  // pile.cc:12-17
  static std::uint16_t Clamp(double v) { return v < 65535.0 ? static_cast<std::uint16_t>(v) : 65535; }
  // Pile::UpdateValidRegion (pile.cc): shrink to [b, e), zero the coverage outside, invalid when shorter than 1260 bases
  void UpdateValidRegion(std::uint32_t b, std::uint32_t e) {
    if (b >= e || e - b < (1260u >> 4)) {
      invalid = true;
      return;
    }
    for (std::uint32_t i = begin_; i < b; ++i) data[i] = 0;
    for (std::uint32_t i = e; i < end_; ++i) data[i] = 0;
    begin_ = b;
    end_ = e;
  }
  // Pile::ClearChimericRegions: the longest stretch between regions whose coverage dips to the median survives
  void ClearChimericRegions(std::uint16_t med) {
    std::uint32_t best_b = 0, best_e = 0, last = begin_;
    std::vector<std::pair<std::uint32_t, std::uint32_t>> unresolved;
    for (const auto& r : chimeric_regions) {
      if (begin_ > r.first || end_ < r.second) continue;
      bool dips = false;
      for (std::uint32_t i = r.first; i <= r.second && !dips; ++i) dips = Clamp(data[i] * 1.82) <= med;
      if (dips) {
        if (r.first - last > best_e - best_b) {
          best_b = last;
          best_e = r.first;
        }
        last = r.second;
      } else {
        unresolved.push_back(r);
      }
    }
    if (end_ - last > best_e - best_b) {
      best_b = last;
      best_e = end_;
    }
    if (best_b != begin_ || best_e != end_) chimeric = true;
    chimeric_regions.swap(unresolved);
    UpdateValidRegion(best_b, best_e);
  }

  std::uint32_t id;
  std::vector<std::uint16_t> data;
  std::vector<std::uint8_t> kmers;
  std::vector<std::pair<std::uint32_t, std::uint32_t>> chimeric_regions;
  std::uint32_t begin_, end_;
  std::uint16_t median_ = 0;
  bool invalid = false, contained = false, chimeric = false;
};

using Piles = std::vector<std::unique_ptr<Pile>>;

// OverlapUpdate (overlap_utils.cc): clip to both valid regions; what one side loses outside its region the other side
// loses at the matching end (which end: by strand); dropped when a pile is invalid, nothing is left, or < 84 bases.
inline bool OverlapUpdate(biosoup::Overlap& o, const Piles& piles) {
  const Pile& L = *piles[o.lhs_id];
  const Pile& R = *piles[o.rhs_id];
  if (L.is_invalid() || R.is_invalid()) return false;
  if (o.lhs_begin >= L.end() || o.lhs_end <= L.begin() || o.rhs_begin >= R.end() || o.rhs_end <= R.begin()) return false;
  const std::uint32_t lh = o.lhs_begin < L.begin() ? L.begin() - o.lhs_begin : 0, lt = o.lhs_end > L.end() ? o.lhs_end - L.end() : 0;
  const std::uint32_t rh = o.rhs_begin < R.begin() ? R.begin() - o.rhs_begin : 0, rt = o.rhs_end > R.end() ? o.rhs_end - R.end() : 0;
  std::uint32_t lb = o.lhs_begin + (o.strand ? rh : rt), le = o.lhs_end - (o.strand ? rt : rh);
  std::uint32_t rb = o.rhs_begin + (o.strand ? lh : lt), re = o.rhs_end - (o.strand ? lt : lh);
  if (lb >= L.end() || le <= L.begin() || rb >= R.end() || re <= R.begin()) return false;
  lb = std::max(lb, L.begin());
  le = std::min(le, L.end());
  rb = std::max(rb, R.begin());
  re = std::min(re, R.end());
  if (lb >= le || le - lb < 84 || rb >= re || re - rb < 84) return false;
  o.lhs_begin = lb;
  o.lhs_end = le;
  o.rhs_begin = rb;
  o.rhs_end = re;
  return true;
}

// GetOverlapType (overlap_utils.cc): 0 internal, 1 lhs contained, 2 rhs contained, 3 lhs -> rhs, 4 rhs -> lhs
inline std::uint32_t GetOverlapType(const biosoup::Overlap& o, const Piles& piles) {
  const Pile& L = *piles[o.lhs_id];
  const Pile& R = *piles[o.rhs_id];
  const std::uint32_t l_len = L.end() - L.begin(), r_len = R.end() - R.begin();
  const std::uint32_t lb = o.lhs_begin - L.begin(), le = o.lhs_end - L.begin();
  const std::uint32_t rb = o.strand ? o.rhs_begin - R.begin() : r_len - (o.rhs_end - R.begin());
  const std::uint32_t re = o.strand ? o.rhs_end - R.begin() : r_len - (o.rhs_begin - R.begin());
  const std::uint32_t overhang = std::min(lb, rb) + std::min(l_len - le, r_len - re);
  if (le - lb < (le - lb + overhang) * 0.875 || re - rb < (re - rb + overhang) * 0.875) return 0;
  if (lb <= rb && l_len - le <= r_len - re) return 1;
  if (rb <= lb && r_len - re <= l_len - le) return 2;
  return lb > rb ? 3 : 4;
}

}  // namespace raven_double
