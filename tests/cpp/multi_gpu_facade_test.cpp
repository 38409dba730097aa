// raven::FindOverlapsAndCreatePiles<Pile> and one polishing round over a raven::DeviceGroup
// (include/raven_hip/multi_gpu.hpp) against the single-device templates on the same input: what a Raven build on a
// multi-GPU node calls in place of construct.cc:661-669 / polish.cc:51.  Virtual ranks on the one GPU of the test box.
#include <cstdio>
#include <fstream>
#include <string>

#include "racon/polisher.hpp"
#include "raven_doubles.hpp"
#include "raven_hip/find_overlaps.hpp"
#include "raven_hip/multi_gpu.hpp"

std::atomic<std::uint32_t> biosoup::NucleicAcid::num_objects{0};

using Sequences = std::vector<std::unique_ptr<biosoup::NucleicAcid>>;

static Sequences Load(const char* path, const std::string& prefix) {
  biosoup::NucleicAcid::num_objects = 0;
  Sequences v;
  std::ifstream in(path);
  std::string line;
  while (std::getline(in, line))
    if (!line.empty()) v.emplace_back(new biosoup::NucleicAcid(prefix + std::to_string(v.size()), line));
  return v;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  try {
    auto sequences = Load(argv[1], "r");
    const int n_ranks = std::atoi(argv[3]);
    raven_double::Piles p1, p2;
    std::vector<std::vector<biosoup::Overlap>> o1(sequences.size()), o2(sequences.size());
    {
      ram::MinimizerEngine engine{nullptr, 15, 5};
      raven::FindOverlapsAndCreatePiles<raven_double::Pile>(nullptr, engine, sequences, 0.001, p1, o1, 32, false);
    }
    raven::DeviceGroup group(std::vector<int>(n_ranks, 0));
    raven::FindOverlapsAndCreatePiles<raven_double::Pile>(nullptr, group, sequences, 0.001, p2, o2, 32, false);
    std::size_t n_ovl = 0, bad = 0;
    for (std::size_t i = 0; i < sequences.size(); ++i) {
      n_ovl += o1[i].size();
      bool same = p1[i]->data == p2[i]->data && o1[i].size() == o2[i].size();
      for (std::size_t j = 0; same && j < o1[i].size(); ++j) {
        const auto &a = o1[i][j], &b = o2[i][j];
        same = a.lhs_id == b.lhs_id && a.lhs_begin == b.lhs_begin && a.lhs_end == b.lhs_end && a.rhs_id == b.rhs_id &&
               a.rhs_begin == b.rhs_begin && a.rhs_end == b.rhs_end && a.score == b.score && a.strand == b.strand;
      }
      bad += same ? 0 : 1;
    }
    std::printf("ranks %u overlaps %zu differing_piles %zu\n", group.size(), n_ovl, bad);
    auto targets = Load(argv[2], "Utg");
    auto reads = Load(argv[1], "read");
    auto polisher = racon::Polisher::Create(nullptr, 0.0, 0.3, 500, true, 3, -5, -4);
    auto single = polisher->Polish(targets, reads, false);
    auto multi = raven::PolishRound(group, targets, reads, false);
    std::size_t differ = single.size() == multi.size() ? 0 : 1;
    for (std::size_t i = 0; i < single.size() && i < multi.size(); ++i)
      differ += (single[i]->deflated_data == multi[i]->deflated_data && single[i]->inflated_len == multi[i]->inflated_len) ? 0 : 1;
    std::printf("polished %zu differing_targets %zu\n", single.size(), differ);
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "error: %s\n", ex.what());
    return 1;
  }
  return 0;
}
