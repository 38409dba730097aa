// N engines behind one host process (include/raven_hip.h: rvn_group_*): the sharded FindOverlapsAndCreatePiles pass and
// the sharded polishing round driven from C++ — what raven::ConstructGraph / raven::Polish would hold instead of one
// engine (RavenLib/src/construct.cc:661-669, polish.cc:43-51) on a multi-GPU node.  The test boxes have one GPU: the
// same device is listed `n_ranks` times (virtual ranks; the exchanges are then device-to-device copies on one GPU).
// Every rank's slice of pile coverage / overlap lists and the polished consensus must equal the single-engine calls
// bit for bit; the program checks that itself and prints one line per check for tests/test_gpu_group.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "raven_hip.h"

namespace {

struct Packed {
  std::vector<uint64_t> words, woff;
  std::vector<uint32_t> len;
};

Packed Load(const char* path) {
  Packed p;
  p.woff.push_back(0);
  std::ifstream in(path);
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    const size_t w0 = p.words.size();
    p.words.resize(w0 + (line.size() + 31) / 32, 0);
    for (size_t i = 0; i < line.size(); ++i) {
      const uint64_t c = line[i] == 'A' ? 0 : (line[i] == 'C' ? 1 : (line[i] == 'G' ? 2 : 3));
      p.words[w0 + (i >> 5)] |= c << ((i << 1) & 63);
    }
    p.woff.push_back(p.words.size());
    p.len.push_back(static_cast<uint32_t>(line.size()));
  }
  p.words.push_back(0);  // pad word, as every caller of rvn_reads_upload provides
  return p;
}

void Check(int rc, const char* what) {
  if (rc != RVN_OK) {
    std::fprintf(stderr, "%s: %s\n", what, rvn_last_error());
    std::exit(1);
  }
}

struct PassResult {
  std::vector<uint16_t> data;
  std::vector<uint64_t> poff;
  std::vector<rvn_overlap> ovl;
  std::vector<uint32_t> ooff;
};

PassResult Fetch(rvn_pass1* p, uint32_t n) {
  PassResult r;
  r.data.resize(rvn_pass1_pile_words(p));
  r.poff.resize(n + 1);
  Check(rvn_pass1_fetch_piles(p, r.data.data(), r.poff.data()), "fetch piles");
  r.ovl.resize(rvn_pass1_num_overlaps(p));
  r.ooff.resize(n + 1);
  Check(rvn_pass1_fetch_overlaps(p, r.ovl.data(), r.ooff.data()), "fetch overlaps");
  return r;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const Packed reads = Load(argv[1]);
  const Packed drafts = Load(argv[2]);
  const uint32_t n_ranks = static_cast<uint32_t>(std::atoi(argv[3]));
  const uint64_t flush = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : (1ULL << 30);
  // (index batch: 2^32 bases in the reference; a smaller value forces several batches on a test-sized read set)
  const uint64_t index_batch = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : (1ULL << 32);
  const uint32_t n = static_cast<uint32_t>(reads.len.size());

  // ---- single engine ----
  rvn_engine* e = nullptr;
  Check(rvn_engine_create(&e, 15, 5, 500, 4, 100, 10000, 0), "engine");
  rvn_reads* rd = nullptr;
  Check(rvn_reads_upload(e, reads.words.data(), reads.woff[n], reads.woff.data(), reads.len.data(), nullptr, n, &rd), "upload");
  rvn_pass1* p1 = nullptr;
  Check(rvn_find_overlaps_and_create_piles(e, rd, 0.001, 32, 0, index_batch, flush, &p1), "pass");
  const PassResult single = Fetch(p1, n);
  rvn_pass1_destroy(p1);
  std::printf("single overlaps %zu\n", single.ovl.size());

  // ---- the group ----
  std::vector<int> devices(n_ranks, 0);
  rvn_group* g = nullptr;
  Check(rvn_group_create(&g, 15, 5, 500, 4, 100, 10000, devices.data(), n_ranks), "group");
  std::vector<uint32_t> bounds(n_ranks + 1);
  std::vector<rvn_pass1*> passes(n_ranks, nullptr);
  Check(rvn_group_find_overlaps_and_create_piles_batched(g, reads.words.data(), reads.woff.data(), reads.len.data(), n, 0.001, 32,
                                                         0, index_batch, flush, bounds.data(), passes.data()),
        "group pass");
  std::printf("bounds");
  for (uint32_t b : bounds) std::printf(" %u", b);
  std::printf("\n");
  for (uint32_t r = 0; r < n_ranks; ++r) {
    const PassResult got = Fetch(passes[r], n);
    const uint32_t lo = bounds[r], hi = bounds[r + 1];
    bool same = true;
    for (uint32_t i = lo; i < hi && same; ++i) {
      const uint64_t cells = single.poff[i + 1] - single.poff[i];
      same = got.poff[i + 1] - got.poff[i] == cells &&
             std::memcmp(&got.data[got.poff[i]], &single.data[single.poff[i]], cells * 2) == 0;
      const uint32_t cnt = single.ooff[i + 1] - single.ooff[i];
      same = same && got.ooff[i + 1] - got.ooff[i] == cnt &&
             std::memcmp(&got.ovl[got.ooff[i]], &single.ovl[single.ooff[i]], static_cast<size_t>(cnt) * sizeof(rvn_overlap)) == 0;
    }
    std::printf("rank %u reads %u identical %d\n", r, hi - lo, same ? 1 : 0);
    rvn_pass1_destroy(passes[r]);
  }

  // ---- one polishing round: single engine vs group ----
  const uint32_t nt = static_cast<uint32_t>(drafts.len.size());
  rvn_reads* td = nullptr;
  Check(rvn_reads_upload(e, drafts.words.data(), drafts.woff[nt], drafts.woff.data(), drafts.len.data(), nullptr, nt, &td), "targets");
  std::vector<uint64_t> ooff(nt + 1, 0);
  for (uint32_t t = 0; t < nt; ++t) ooff[t + 1] = ooff[t] + 2ULL * drafts.len[t] + 1024;
  std::vector<uint8_t> c1(ooff[nt]), c2(ooff[nt]);
  std::vector<uint32_t> l1(nt), l2(nt);
  std::vector<double> r1(nt), r2(nt);
  Check(rvn_polish_round(e, td, rd, nullptr, nullptr, 0.0, 0.3, 500, 1, 3, -5, -4, c1.data(), ooff.data(), l1.data(), r1.data(), nullptr),
        "round");
  Check(rvn_group_polish_round(g, drafts.words.data(), drafts.woff.data(), drafts.len.data(), nt, reads.words.data(),
                               reads.woff.data(), reads.len.data(), n, 0.0, 0.3, 500, 1, 3, -5, -4, c2.data(), ooff.data(), l2.data(),
                               r2.data()),
        "group round");
  for (uint32_t t = 0; t < nt; ++t) {
    const bool same = l1[t] == l2[t] && std::memcmp(&c1[ooff[t]], &c2[ooff[t]], l1[t]) == 0 && r1[t] == r2[t];
    std::printf("target %u len %u ratio %.6f identical %d\n", t, l1[t], r1[t], same ? 1 : 0);
  }
  // ---- the same round with block qualities (FASTQ variant: mean-quality filter at q = 10, quality-weighted edges):
  // seeded block qualities, a fifth of the reads below the threshold ----
  {
    std::vector<uint8_t> quals;
    std::vector<uint64_t> qoff(1, 0);
    uint64_t state = 0x9E3779B97F4A7C15ULL;
    for (uint32_t i = 0; i < n; ++i) {
      state = state * 6364136223846793005ULL + 1442695040888963407ULL;
      const bool low = (state >> 33) % 5 == 0;
      const uint32_t blocks = (reads.len[i] + 63) / 64;
      for (uint32_t b = 0; b < blocks; ++b) {
        state = state * 6364136223846793005ULL + 1442695040888963407ULL;
        quals.push_back(static_cast<uint8_t>(33 + (low ? 4 : 12) + (state >> 40) % 8));
      }
      qoff.push_back(quals.size());
    }
    Check(rvn_reads_attach_quality(e, rd, quals.data(), qoff.data(), 6), "attach");
    rvn_polish_stats st1{};
    Check(rvn_polish_round(e, td, rd, nullptr, nullptr, 10.0, 0.3, 500, 1, 3, -5, -4, c1.data(), ooff.data(), l1.data(), r1.data(), &st1),
          "round with qualities");
    Check(rvn_group_polish_round_q(g, drafts.words.data(), drafts.woff.data(), drafts.len.data(), nt, reads.words.data(),
                                   reads.woff.data(), reads.len.data(), n, quals.data(), qoff.data(), 6, 10.0, 0.3, 500, 1, 3, -5, -4,
                                   c2.data(), ooff.data(), l2.data(), r2.data()),
          "group round with qualities");
    for (uint32_t t = 0; t < nt; ++t) {
      const bool same = l1[t] == l2[t] && std::memcmp(&c1[ooff[t]], &c2[ooff[t]], l1[t]) == 0 && r1[t] == r2[t];
      std::printf("quality target %u len %u ratio %.6f identical %d dropped_layers %llu\n", t, l1[t], r1[t], same ? 1 : 0,
                  static_cast<unsigned long long>(st1.n_dropped_layers));
    }
  }
  {
    std::vector<uint8_t> direct(static_cast<size_t>(n_ranks) * n_ranks, 0);
    const int all = rvn_group_peer_access(g, direct.data());
    std::printf("peer access all %d\n", all);
  }
  rvn_reads_destroy(td);
  rvn_reads_destroy(rd);
  rvn_group_destroy(g);
  rvn_engine_destroy(e);
  return 0;
}
