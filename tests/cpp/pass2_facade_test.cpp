// The second-pass facade of include/raven_hip/find_overlaps.hpp the way RavenLib/src/construct.cc:733-738 calls it:
// first pass, trimming, then raven::FindOverlapsAndRepetetiveRegions with the reference's signature on a Pile stand-in.
// Prints a deterministic dump that tests/test_gpu_facade.py compares with the ctypes path; tests/test_abi.py only
// compiles and links it (no GPU).
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>

#include "raven_hip/find_overlaps.hpp"

std::atomic<std::uint32_t> biosoup::NucleicAcid::num_objects{0};

namespace {
struct TestPile {  // raven::Pile's members the two templates use (+ the two Adopt hooks of INTEGRATION.md)
  TestPile(std::uint32_t id_, std::uint32_t len) : id(id_), data(len >> 4, 0), begin_(0), end_(len >> 4) {}
  void AdoptCoverage(const std::uint16_t* d, std::size_t n) { data.assign(d, d + n); }
  void AdoptKmers(const std::uint8_t* c, std::size_t n) { kmers.assign(c, c + n); }
  std::uint32_t begin() const { return begin_ << 4; }  // raven::Pile::begin() (pile.h): cells -> bases
  std::uint32_t end() const { return end_ << 4; }
  bool is_invalid() const { return invalid; }
  void set_is_invalid() { invalid = true; }
  void set_is_contained() { contained = true; }
  std::uint32_t id;
  std::vector<std::uint16_t> data;
  std::vector<std::uint8_t> kmers;
  std::uint32_t begin_, end_;
  bool invalid = false, contained = false;
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  std::vector<std::unique_ptr<biosoup::NucleicAcid>> sequences;
  std::string line;
  while (std::getline(in, line)) sequences.emplace_back(new biosoup::NucleicAcid("r" + std::to_string(sequences.size()), line));
  const double identity = argc > 2 ? std::atof(argv[2]) : 0.0;
  try {
    ram::MinimizerEngine minimizer_engine{nullptr, 15, 5};
    std::vector<std::unique_ptr<TestPile>> piles;
    std::vector<std::vector<biosoup::Overlap>> overlaps(sequences.size());
    raven::FindOverlapsAndCreatePiles<TestPile>(nullptr, minimizer_engine, sequences, 0.001, piles, overlaps, 32, false);
    // TrimAndAnnotatePiles would set begin_/end_/invalid here (construct.cc:123-152); the stand-in keeps whole piles and
    // invalidates the last one: with NO invalid pile the reference maps nothing at all (its `s`, construct.cc:343-349)
    if (!piles.empty()) piles.back()->invalid = true;
    raven::FindOverlapsAndRepetetiveRegions<TestPile>(nullptr, minimizer_engine, 0.001, 28, identity, piles, overlaps, sequences);
    std::printf("lists %zu\n", overlaps.size());
    for (const auto& o : overlaps.back())
      std::printf("O %u %u %u %u %u %u %u %d\n", o.lhs_id, o.lhs_begin, o.lhs_end, o.rhs_id, o.rhs_begin, o.rhs_end, o.score,
                  o.strand ? 1 : 0);
    for (std::size_t i = 0; i < piles.size(); ++i) {
      std::uint64_t h = 0;
      for (auto v : piles[i]->kmers) h = h * 1000003ULL + v;
      std::printf("P %zu %d %d %zu %llu\n", i, piles[i]->contained ? 1 : 0, piles[i]->invalid ? 1 : 0, piles[i]->kmers.size(),
                  static_cast<unsigned long long>(h));
    }
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "error: %s\n", ex.what());
    return 1;
  }
  return 0;
}
