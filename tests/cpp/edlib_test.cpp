// The identity-filter call pattern of RavenLib/src/construct.cc:176-203 against include/edlib.h (the drop-in served
// by libraven_hip.so): inflate two spans, reverse-complement the rhs for opposite-strand overlaps, edlibAlign with
// the default config, score = 1 - ed / max(len), edlibFreeAlignResult — from several threads at once, as the
// reference's pool does.  Every distance is checked against a plain DP computed here.
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "edlib.h"

static int dp_distance(const std::string& a, const std::string& b) {
  std::vector<int> prev(b.size() + 1), cur(b.size() + 1);
  for (size_t j = 0; j <= b.size(); ++j) prev[j] = static_cast<int>(j);
  for (size_t i = 1; i <= a.size(); ++i) {
    cur[0] = static_cast<int>(i);
    for (size_t j = 1; j <= b.size(); ++j)
      cur[j] = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), prev[j - 1] + (a[i - 1] != b[j - 1]));
    prev.swap(cur);
  }
  return prev[b.size()];
}

static std::string revcomp(const std::string& s) {
  std::string r(s.rbegin(), s.rend());
  for (auto& c : r) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A';
  return r;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  std::vector<std::string> seqs;
  std::string line;
  while (std::getline(in, line)) seqs.push_back(line);
  if (seqs.size() < 2) return 2;

  // config sanity (no device needed)
  EdlibAlignConfig cfg = edlibDefaultAlignConfig();
  std::printf("default_config %d %d %d\n", cfg.k, static_cast<int>(cfg.mode), static_cast<int>(cfg.task));
  {
    EdlibAlignResult r = edlibAlign("ACGT", 4, "ACGT", 4, edlibNewAlignConfig(-1, EDLIB_MODE_HW, EDLIB_TASK_DISTANCE, nullptr, 0));
    std::printf("hw_mode_status %d\n", r.status);
    edlibFreeAlignResult(r);
    r = edlibAlign("ACGTN", 5, "ACGTN", 5, edlibDefaultAlignConfig());
    std::printf("five_symbols_status %d\n", r.status);
    edlibFreeAlignResult(r);
  }
  {
    EdlibAlignResult r = edlibAlign(seqs[0].c_str(), static_cast<int>(seqs[0].size()), seqs[0].c_str(),
                                    static_cast<int>(seqs[0].size()), edlibDefaultAlignConfig());
    if (r.status != EDLIB_STATUS_OK) {
      std::printf("NO_DEVICE status %d\n", r.status);
      return 1;
    }
    std::printf("self %d locations %d end %d alphabet %d\n", r.editDistance, r.numLocations,
                r.numLocations ? r.endLocations[0] : -1, r.alphabetLength);
    edlibFreeAlignResult(r);
  }
  // pairs (i, i+1), alternating strands; 8 threads
  const size_t n_pairs = seqs.size() - 1;
  std::vector<int> got(n_pairs, -2), want(n_pairs, -3);
  std::vector<double> score(n_pairs, 0);
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < 8; ++t)
    pool.emplace_back([&, t]() {
      for (size_t i = t; i < n_pairs; i += 8) {
        std::string lhs = seqs[i];
        std::string rhs = (i & 1) ? revcomp(seqs[i + 1]) : seqs[i + 1];
        auto result = edlibAlign(lhs.c_str(), lhs.size(), rhs.c_str(), rhs.size(), edlibDefaultAlignConfig());
        score[i] = result.status == EDLIB_STATUS_OK
                       ? 1. - static_cast<double>(result.editDistance) / std::max(lhs.size(), rhs.size())
                       : 0.;
        got[i] = result.status == EDLIB_STATUS_OK ? result.editDistance : -1;
        edlibFreeAlignResult(result);
        want[i] = dp_distance(lhs, rhs);
      }
    });
  for (auto& th : pool) th.join();
  size_t bad = 0;
  for (size_t i = 0; i < n_pairs; ++i) bad += got[i] != want[i];
  std::printf("pairs %zu mismatches %zu\n", n_pairs, bad);
  // k threshold: distance above k -> -1
  {
    EdlibAlignResult r = edlibAlign(seqs[0].c_str(), seqs[0].size(), seqs[1].c_str(), seqs[1].size(),
                                    edlibNewAlignConfig(std::max(0, want[0] - 1), EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, nullptr, 0));
    std::printf("k_below %d\n", want[0] > 0 ? r.editDistance : -1);
    edlibFreeAlignResult(r);
    r = edlibAlign(seqs[0].c_str(), seqs[0].size(), seqs[1].c_str(), seqs[1].size(),
                   edlibNewAlignConfig(want[0], EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, nullptr, 0));
    std::printf("k_equal %d\n", r.editDistance == want[0] ? 1 : 0);
    edlibFreeAlignResult(r);
  }
  return 0;
}
