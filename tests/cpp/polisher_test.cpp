// Exercises include/racon/polisher.hpp the way RavenLib/src/polish.cc:43-60 does: Create(...) with Raven's
// arguments, Polish(targets, sequences, false), then the tag parsing Raven applies to the result names.
// Input: two text files, one sequence per line (targets, reads); a third optional argument gives a constant
// block quality for every read.  Prints a deterministic dump that tests/test_gpu_facade.py compares with ctypes.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "racon/polisher.hpp"

std::atomic<std::uint32_t> biosoup::NucleicAcid::num_objects{0};

static std::vector<std::unique_ptr<biosoup::NucleicAcid>> Load(const char* path, const char* prefix, int q) {
  std::vector<std::unique_ptr<biosoup::NucleicAcid>> v;
  std::ifstream in(path);
  std::string line;
  while (std::getline(in, line)) {
    v.emplace_back(new biosoup::NucleicAcid(prefix + std::to_string(v.size()), line));
    if (q >= 0) v.back()->block_quality.assign((line.size() + 63) / 64, static_cast<std::uint8_t>(q));
  }
  return v;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int q = argc > 3 ? std::atoi(argv[3]) : -1;
  try {
    auto targets = Load(argv[1], "Utg", -1);
    biosoup::NucleicAcid::num_objects = 0;
    auto reads = Load(argv[2], "read", q);
    bool threw = false;
    try {
      racon::Polisher::Create(nullptr, 10, 0.3, 0, true, 3, -5, -4);
    } catch (const std::invalid_argument&) {
      threw = true;
    }
    std::printf("zero_window_throws %d\n", threw ? 1 : 0);
    auto polisher = racon::Polisher::Create(nullptr, q >= 0 ? 10.0 : 0.0, 0.3, 500, true, 3, -5, -4, 0, false, 0);  // polish.cc:43-48
    auto polished = polisher->Polish(targets, reads, false);                                                   // polish.cc:51
    std::printf("polished %zu\n", polished.size());
    for (const auto& it : polished) {
      // Raven: node id after "Utg", polished ratio after the last ':' (polish.cc:55-59)
      const std::size_t tag = it->name.rfind(':');
      const double ratio = std::atof(&it->name[tag + 1]);
      const long id = std::atol(&it->name[3]);
      std::string data(it->inflated_len, 'A');
      for (std::uint32_t i = 0; i < it->inflated_len; ++i) data[i] = "ACGT"[(it->deflated_data[i >> 5] >> ((i << 1) & 63)) & 3];
      std::printf("P %ld %.6f %s\nS %s\n", id, ratio, it->name.c_str(), data.c_str());
    }
    // a second round the way polish.cc:50-52 runs it: the first round's result as targets (the facade recognises it and
    // takes the consensus the engine still holds in HBM); then a third one whose first target was rotated in place like a
    // circular unitig (polish.cc:60-65): not the engine's copy any more, the facade uploads it
    auto second = polisher->Polish(polished, reads, false);
    for (const auto& it : second) {
      std::string data(it->inflated_len, 'A');
      for (std::uint32_t i = 0; i < it->inflated_len; ++i) data[i] = "ACGT"[(it->deflated_data[i >> 5] >> ((i << 1) & 63)) & 3];
      std::printf("P2 %s\nS2 %s\n", it->name.c_str(), data.c_str());
    }
    {
      auto s = second[0]->InflateData();
      const std::size_t b = static_cast<std::size_t>(0.42 * s.size());
      s = s.substr(b) + s.substr(0, b);
      second[0]->deflated_data = biosoup::NucleicAcid{"", s}.deflated_data;
      auto third = polisher->Polish(second, reads, false);
      for (const auto& it : third) {
        std::string data(it->inflated_len, 'A');
        for (std::uint32_t i = 0; i < it->inflated_len; ++i) data[i] = "ACGT"[(it->deflated_data[i >> 5] >> ((i << 1) & 63)) & 3];
        std::printf("S3 %s\n", data.c_str());
      }
    }
    std::printf("resident_rounds %zu\n", polisher->resident_rounds());
    auto kept = polisher->Polish(targets, std::vector<std::unique_ptr<biosoup::NucleicAcid>>{}, true);
    std::printf("dropped_without_reads %zu\n", targets.size() - kept.size());
  } catch (const std::exception& ex) {
    std::printf("EXCEPTION %s\n", ex.what());
    return 1;
  }
  return 0;
}
