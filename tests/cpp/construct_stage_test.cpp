// The reference's overlap stages in the reference's order, through the facades: raven::ConstructGraph
// (RavenLib/src/construct.cc:650-707) from stage -5 up to the hand-over to the layout —
//   FindOverlapsAndCreatePiles -> TrimAndAnnotatePiles -> ResolveContainedReads (identity filter through edlibAlign, as
//   construct.cc:162-217 does) -> ResolveChimericSequences -> FindOverlapsAndRepetetiveRegions
// — then, in further modes, the unitig hand-over to the polisher (GetUnitigs' name rule, common.cc:227-252, parsed back
// as polish.cc:50-74 does over cfg.num_rounds rounds) and SalvagePlasmids' use of the engine (assemble.cc:732-795).
// What stays Raven's own code in a real build (Pile's host members, overlap_utils.cc, the loops of construct.cc) is a
// test double here (raven_doubles.hpp); the mapping, annotation, identity and polishing work goes through
// include/raven_hip/find_overlaps.hpp, include/ram/minimizer_engine.hpp, include/edlib.h, include/racon/polisher.hpp.
// Prints deterministic dumps that tests/test_gpu_stages.py compares with the oracle's statement of the same sequences.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "edlib.h"
#include "racon/polisher.hpp"
#include "raven_doubles.hpp"
#include "raven_hip/find_overlaps.hpp"

std::atomic<std::uint32_t> biosoup::NucleicAcid::num_objects{0};

namespace {

using raven_double::GetOverlapType;
using raven_double::OverlapUpdate;
using raven_double::Pile;
using Sequences = std::vector<std::unique_ptr<biosoup::NucleicAcid>>;
using Overlaps = std::vector<std::vector<biosoup::Overlap>>;

Sequences Load(const char* path, const std::string& prefix) {
  Sequences v;
  std::ifstream in(path);
  std::string line;
  while (std::getline(in, line))
    if (!line.empty()) v.emplace_back(new biosoup::NucleicAcid(prefix + std::to_string(v.size()), line));
  return v;
}

std::string ReverseComplement(const std::string& s) {
  std::string r(s.rbegin(), s.rend());
  for (char& c : r) c = c == 'A' ? 'T' : (c == 'C' ? 'G' : (c == 'G' ? 'C' : 'A'));
  return r;
}

// construct.cc:154-248
void ResolveContainedReads(const raven_double::Piles& piles, Overlaps& overlaps, const Sequences& sequences, double identity) {
  if (identity != 0) {
    for (std::uint32_t i = 0; i < overlaps.size(); ++i) {
      std::uint32_t k = 0;
      for (std::uint32_t j = 0; j < overlaps[i].size(); ++j) {
        if (!OverlapUpdate(overlaps[i][j], piles)) continue;
        const auto& it = overlaps[i][j];
        auto lhs = sequences[it.lhs_id]->InflateData(it.lhs_begin, it.lhs_end - it.lhs_begin);
        auto rhs = sequences[it.rhs_id]->InflateData(it.rhs_begin, it.rhs_end - it.rhs_begin);
        if (!it.strand) rhs = ReverseComplement(rhs);
        auto result = edlibAlign(lhs.c_str(), lhs.size(), rhs.c_str(), rhs.size(), edlibDefaultAlignConfig());
        const double score = result.status == EDLIB_STATUS_OK
                                 ? 1. - static_cast<double>(result.editDistance) / std::max(lhs.size(), rhs.size())
                                 : 0.;
        edlibFreeAlignResult(result);
        if (score < identity) continue;
        overlaps[i][k++] = overlaps[i][j];
      }
      overlaps[i].resize(k);
    }
  }
  for (std::uint32_t i = 0; i < overlaps.size(); ++i) {
    std::uint32_t k = 0;
    for (std::uint32_t j = 0; j < overlaps[i].size(); ++j) {
      if (!OverlapUpdate(overlaps[i][j], piles)) continue;
      const std::uint32_t type = GetOverlapType(overlaps[i][j], piles);
      if (type == 1 && !piles[overlaps[i][j].rhs_id]->is_maybe_chimeric()) {
        piles[i]->set_is_contained();
      } else if (type == 2 && !piles[i]->is_maybe_chimeric()) {
        piles[overlaps[i][j].rhs_id]->set_is_contained();
      } else {
        overlaps[i][k++] = overlaps[i][j];
      }
    }
    overlaps[i].resize(k);
  }
  for (std::uint32_t i = 0; i < piles.size(); ++i) {
    if (piles[i]->is_contained()) {
      piles[i]->set_is_invalid();
      std::vector<biosoup::Overlap>().swap(overlaps[i]);
    }
  }
}

// construct.cc:250-313
void ResolveChimericSequences(const raven_double::Piles& piles, Overlaps& overlaps) {
  std::vector<std::uint16_t> medians;
  for (const auto& it : piles)
    if (it->median() != 0) medians.emplace_back(it->median());
  if (medians.empty()) return;  // (the reference would index an empty vector here)
  std::nth_element(medians.begin(), medians.begin() + medians.size() / 2, medians.end());
  const std::uint16_t median = medians[medians.size() / 2];
  for (const auto& it : piles) {
    if (it->is_invalid()) continue;
    it->ClearChimericRegions(median);
    if (it->is_invalid()) std::vector<biosoup::Overlap>().swap(overlaps[it->id]);
  }
  for (std::uint32_t i = 0; i < overlaps.size(); ++i) {
    std::uint32_t k = 0;
    for (std::uint32_t j = 0; j < overlaps[i].size(); ++j)
      if (OverlapUpdate(overlaps[i][j], piles)) overlaps[i][k++] = overlaps[i][j];
    overlaps[i].resize(k);
  }
  for (const auto& it : overlaps) {
    for (const auto& jt : it) {
      const std::uint32_t type = GetOverlapType(jt, piles);
      if (type == 1) {
        piles[jt.lhs_id]->set_is_contained();
        piles[jt.lhs_id]->set_is_invalid();
      } else if (type == 2) {
        piles[jt.rhs_id]->set_is_contained();
        piles[jt.rhs_id]->set_is_invalid();
      }
    }
  }
  overlaps.clear();
}

void DumpPiles(const char* tag, const raven_double::Piles& piles) {
  for (const auto& p : piles) {
    std::uint64_t h = 0;
    for (auto v : p->data) h = h * 1000003ULL + v;
    std::printf("%s %u %u %u %u %d %d %d %zu %llu\n", tag, p->id, p->begin_, p->end_, p->median_, p->invalid ? 1 : 0,
                p->contained ? 1 : 0, p->chimeric ? 1 : 0, p->chimeric_regions.size(), static_cast<unsigned long long>(h));
  }
}

int Stages(const char* reads_path, double identity) {
  auto sequences = Load(reads_path, "r");
  int stage = -5;
  // construct.cc:661-666
  ram::MinimizerEngine minimizer_engine{nullptr, 15, 5};
  Overlaps overlaps;
  overlaps.resize(sequences.size());
  raven_double::Piles piles;
  if (stage == -5) {
    raven::Pass1Handle pass;
    raven::FindOverlapsAndCreatePiles<Pile>(nullptr, minimizer_engine, sequences, 0.001, piles, overlaps, 32, false,
                                            1ULL << 32, 1ULL << 30, &pass);
    std::size_t n_ovl = 0;
    for (const auto& it : overlaps) n_ovl += it.size();
    std::printf("pass1 overlaps %zu\n", n_ovl);
    raven::TrimAndAnnotatePiles<Pile>(nullptr, piles, overlaps, pass);
    DumpPiles("A", piles);
    ResolveContainedReads(piles, overlaps, sequences, identity);
    n_ovl = 0;
    for (const auto& it : overlaps) n_ovl += it.size();
    std::printf("resolved overlaps %zu\n", n_ovl);
    ResolveChimericSequences(piles, overlaps);
    DumpPiles("B", piles);
    ++stage;
  }
  if (stage == -4) {
    raven::FindOverlapsAndRepetetiveRegions<Pile>(nullptr, minimizer_engine, 0.001, 15, identity, piles, overlaps, sequences);
    std::printf("lists %zu\n", overlaps.size());
    for (const auto& o : overlaps.back())
      std::printf("O %u %u %u %u %u %u %u %d\n", o.lhs_id, o.lhs_begin, o.lhs_end, o.rhs_id, o.rhs_begin, o.rhs_end, o.score,
                  o.strand ? 1 : 0);
    for (const auto& p : piles) {
      std::uint64_t h = 0;
      for (auto v : p->kmers) h = h * 1000003ULL + v;
      std::printf("K %u %d %d %zu %llu\n", p->id, p->contained ? 1 : 0, p->invalid ? 1 : 0, p->kmers.size(),
                  static_cast<unsigned long long>(h));
    }
    // ResolveRepeatInducedOverlaps + ConstructAssemblyGraph: the layout, on the host, untouched (out of this repo's scope)
    ++stage;
  }
  std::printf("stage %d\n", stage);
  return 0;
}

// common.cc:227-252 (GetUnitigs' names) -> polish.cc:43-74 (rounds, tag parsing, the 0.42 rotation of circular unitigs)
int Polish(const char* unitigs_path, const char* reads_path, int rounds) {
  auto drafts = Load(unitigs_path, "d");
  biosoup::NucleicAcid::num_objects = 0;  // common.cc:231
  Sequences unitigs;
  struct Node {
    std::string data;
    bool circular, polished;
  };
  std::vector<Node> nodes;
  for (std::size_t i = 0; i < drafts.size(); ++i) {
    const std::uint32_t node_id = 100 + 2 * static_cast<std::uint32_t>(i);  // ids of non-rc nodes: not the position in the list
    nodes.resize(node_id + 1);
    nodes[node_id] = Node{drafts[i]->InflateData(), (i & 1) != 0, false};
    const std::string name = "Utg" + std::to_string(node_id) + " LN:i:" + std::to_string(drafts[i]->inflated_len) +
                             " RC:i:" + std::to_string(7 + i) + " XO:i:" + std::to_string((i & 1) ? 1 : 0);
    unitigs.emplace_back(new biosoup::NucleicAcid(name, drafts[i]->InflateData()));
  }
  biosoup::NucleicAcid::num_objects = 0;
  auto sequences = Load(reads_path, "read");
  auto polisher = racon::Polisher::Create(nullptr, 0.0, 0.3, 500, true, 3, -5, -4, 0, false, 0);
  for (int stage = 0; stage < rounds; ++stage) {
    auto polished = polisher->Polish(unitigs, sequences, false);
    unitigs.swap(polished);
    for (const auto& it : unitigs) {
      Node& node = nodes[std::atoi(&it->name[3])];
      std::size_t tag;
      if ((tag = it->name.rfind(':')) != std::string::npos) {
        if (std::atof(&it->name[tag + 1]) > 0) {
          if (node.circular) {  // rotate
            auto s = it->InflateData();
            const std::size_t b = 0.42 * s.size();
            s = s.substr(b) + s.substr(0, b);
            it->deflated_data = biosoup::NucleicAcid{"", s}.deflated_data;
          }
          node.polished = true;
          node.data = it->InflateData();
        }
      }
    }
    for (const auto& it : unitigs) std::printf("R %d %s\n", stage, it->name.c_str());
  }
  for (std::size_t id = 0; id < nodes.size(); ++id)
    if (!nodes[id].data.empty()) std::printf("N %zu %d %s\n", id, nodes[id].polished ? 1 : 0, nodes[id].data.c_str());
  return 0;
}

// assemble.cc:732-795: duplicates among circular non-unitig nodes, then against the unitigs
int Plasmids(const char* plasmids_path, const char* unitigs_path) {
  auto plasmids = Load(plasmids_path, "Ctg");
  std::sort(plasmids.begin(), plasmids.end(),
            [](const std::unique_ptr<biosoup::NucleicAcid>& lhs, const std::unique_ptr<biosoup::NucleicAcid>& rhs) -> bool {
              return lhs->inflated_len < rhs->inflated_len;
            });
  for (std::uint32_t i = 0; i < plasmids.size(); ++i) plasmids[i]->id = i;
  ram::MinimizerEngine minimizer_engine{nullptr};
  minimizer_engine.Minimize(plasmids.begin(), plasmids.end());
  minimizer_engine.Filter(0.001);
  for (auto& it : plasmids) {
    if (!minimizer_engine.Map(it, true, true).empty()) {
      std::printf("dup_within %s\n", it->name.c_str());
      it.reset();
    }
  }
  plasmids.erase(std::remove(plasmids.begin(), plasmids.end(), nullptr), plasmids.end());
  if (plasmids.empty()) {
    std::printf("salvaged 0\n");
    return 0;
  }
  auto unitigs = Load(unitigs_path, "Utg");
  minimizer_engine.Minimize(unitigs.begin(), unitigs.end(), true);
  minimizer_engine.Filter(0.001);
  for (auto& it : plasmids) {
    if (!minimizer_engine.Map(it, false, false).empty()) {
      std::printf("dup_unitig %s\n", it->name.c_str());
      it.reset();
    }
  }
  plasmids.erase(std::remove(plasmids.begin(), plasmids.end(), nullptr), plasmids.end());
  for (const auto& it : plasmids) std::printf("kept %s %d\n", it->name.c_str(), std::atoi(&it->name[3]));
  std::printf("salvaged %zu\n", plasmids.size());
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string mode = argv[1];
  try {
    if (mode == "stages") return Stages(argv[2], argc > 3 ? std::atof(argv[3]) : 0.0);
    if (mode == "polish" && argc >= 4) return Polish(argv[2], argv[3], argc > 4 ? std::atoi(argv[4]) : 2);
    if (mode == "plasmids" && argc >= 4) return Plasmids(argv[2], argv[3]);
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "error: %s\n", ex.what());
    return 1;
  }
  return 2;
}
