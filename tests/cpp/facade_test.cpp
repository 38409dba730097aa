// Exercises the C++ facade (include/ram/minimizer_engine.hpp, include/raven_hip/find_overlaps.hpp) the way
// RavenLib/src/construct.cc does: ConstructGraph's engine + FindOverlapsAndCreatePiles call, then per-read
// Map() vs MapBatch().  Reads a FASTA-like text file (one sequence per line) and prints a deterministic dump
// that tests/test_gpu_facade.py compares with the ctypes path.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>

#include "raven_hip/find_overlaps.hpp"

std::atomic<std::uint32_t> biosoup::NucleicAcid::num_objects{0};

namespace {
struct TestPile {  // stands in for raven::Pile (+ the AdoptCoverage hook of INTEGRATION.md)
  TestPile(std::uint32_t id_, std::uint32_t len) : id(id_), data(len >> 4, 0) {}
  void AdoptCoverage(const std::uint16_t* d, std::size_t n) { data.assign(d, d + n); }
  std::uint32_t id;
  std::vector<std::uint16_t> data;
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  std::vector<std::unique_ptr<biosoup::NucleicAcid>> sequences;
  std::string line;
  while (std::getline(in, line)) sequences.emplace_back(new biosoup::NucleicAcid("r" + std::to_string(sequences.size()), line));

  try {
    ram::MinimizerEngine minimizer_engine{nullptr, 15, 5};  // construct.cc:661-662
    bool threw = false;
    try {
      minimizer_engine.Filter(2.0);
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    std::printf("filter_throws %d\n", threw ? 1 : 0);

    std::vector<std::unique_ptr<TestPile>> piles;
    std::vector<std::vector<biosoup::Overlap>> overlaps(sequences.size());
    raven::FindOverlapsAndCreatePiles<TestPile>(nullptr, minimizer_engine, sequences, 0.001, piles, overlaps, 32, false);
    std::uint64_t cov = 0;
    for (const auto& p : piles)
      for (auto v : p->data) cov = cov * 1000003ULL + v;
    std::printf("piles %zu cov_hash %llu\n", piles.size(), static_cast<unsigned long long>(cov));
    for (std::size_t i = 0; i < overlaps.size(); ++i)
      for (const auto& o : overlaps[i])
        std::printf("O %zu %u %u %u %u %u %u %u %d\n", i, o.lhs_id, o.lhs_begin, o.lhs_end, o.rhs_id, o.rhs_begin,
                    o.rhs_end, o.score, o.strand ? 1 : 0);

    // the engine still holds the index of the last Minimize: per-read Map == MapBatch
    minimizer_engine.Minimize(sequences.begin(), sequences.end(), false);
    minimizer_engine.Filter(0.001);
    auto batch = minimizer_engine.MapBatch(sequences.begin(), sequences.end(), true, true, true);
    std::size_t mism = 0, total = 0;
    for (std::size_t i = 0; i < sequences.size() && i < 16; ++i) {
      std::vector<std::uint32_t> filtered;
      auto one = minimizer_engine.Map(sequences[i], true, true, true, &filtered);
      total += one.size();
      if (one.size() != batch[i].size()) ++mism;
      else
        for (std::size_t j = 0; j < one.size(); ++j)
          if (one[j].lhs_begin != batch[i][j].lhs_begin || one[j].rhs_id != batch[i][j].rhs_id || one[j].score != batch[i][j].score) ++mism;
    }
    std::printf("map_single_vs_batch mismatches %zu total %zu\n", mism, total);

    // Map() from concurrent threads, the way construct.cc:60-64 / :373-381 submit it to the pool: indexed sequences
    // (served from one batched pass) and a copy that is NOT part of the indexed vector (mapped on its own)
    {
      std::vector<std::vector<biosoup::Overlap>> got(sequences.size());
      std::vector<std::vector<std::uint32_t>> got_f(sequences.size());
      auto batch_f = std::vector<std::vector<std::uint32_t>>();
      auto batch2 = minimizer_engine.MapBatch(sequences.begin(), sequences.end(), true, true, false, &batch_f);
      const unsigned n_thr = 8;
      std::vector<std::thread> pool;
      std::vector<std::unique_ptr<biosoup::NucleicAcid>> outsiders;
      for (std::size_t i = 0; i < sequences.size() && i < 6; ++i) {
        outsiders.emplace_back(new biosoup::NucleicAcid(sequences[i]->name, sequences[i]->InflateData()));
        outsiders.back()->id = sequences[i]->id;
      }
      std::vector<std::vector<biosoup::Overlap>> got_out(outsiders.size());
      for (unsigned t = 0; t < n_thr; ++t)
        pool.emplace_back([&, t]() {
          for (std::size_t i = t; i < sequences.size(); i += n_thr)
            got[i] = minimizer_engine.Map(sequences[i], true, true, false, &got_f[i]);
          for (std::size_t i = t; i < outsiders.size(); i += n_thr)
            got_out[i] = minimizer_engine.Map(outsiders[i], true, true, false);
        });
      for (auto& th : pool) th.join();
      auto same = [](const std::vector<biosoup::Overlap>& a, const std::vector<biosoup::Overlap>& b) {
        if (a.size() != b.size()) return false;
        for (std::size_t j = 0; j < a.size(); ++j)
          if (a[j].lhs_id != b[j].lhs_id || a[j].lhs_begin != b[j].lhs_begin || a[j].lhs_end != b[j].lhs_end ||
              a[j].rhs_id != b[j].rhs_id || a[j].rhs_begin != b[j].rhs_begin || a[j].rhs_end != b[j].rhs_end ||
              a[j].score != b[j].score || a[j].strand != b[j].strand)
            return false;
        return true;
      };
      std::size_t bad = 0, n_ovl = 0;
      for (std::size_t i = 0; i < sequences.size(); ++i) {
        n_ovl += got[i].size();
        if (!same(got[i], batch2[i]) || got_f[i] != batch_f[i]) ++bad;
      }
      for (std::size_t i = 0; i < outsiders.size(); ++i)
        if (!same(got_out[i], batch2[i])) ++bad;
      std::printf("map_concurrent mismatches %zu total %zu\n", bad, n_ovl);
    }
  } catch (const std::exception& ex) {
    std::printf("EXCEPTION %s\n", ex.what());
    return 1;
  }
  return 0;
}
