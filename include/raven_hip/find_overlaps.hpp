// raven_hip/find_overlaps.hpp — raven::FindOverlapsAndCreatePiles (RavenLib/src/construct.cc:14-121, decl
// RavenLib/include/raven/graph/construct.h:22-29) with the reference's signature and semantics, running the
// whole pass on the MI355X through one C-ABI call.  Header-only; see INTEGRATION.md.
//
// PileT must provide PileT(std::uint32_t id, std::uint32_t len) and
//   void AdoptCoverage(const std::uint16_t* data, std::size_t n)   // replaces Pile::data_ (n == len >> 4)
// (a two-line addition to raven::Pile; AddLayers itself is no longer called on this path).
#ifndef RAVEN_HIP_FIND_OVERLAPS_HPP_
#define RAVEN_HIP_FIND_OVERLAPS_HPP_

#include <cstdint>
#include <iostream>
#include <memory>
#include <vector>

#include "ram/minimizer_engine.hpp"

namespace raven {

template <typename PileT>
void FindOverlapsAndCreatePiles(const std::shared_ptr<thread_pool::ThreadPool>& /*thread_pool*/,
                                ram::MinimizerEngine& minimizer_engine,
                                const std::vector<std::unique_ptr<biosoup::NucleicAcid>>& sequences, double freq,
                                std::vector<std::unique_ptr<PileT>>& piles,
                                std::vector<std::vector<biosoup::Overlap>>& overlaps,
                                std::size_t kMaxNumOverlaps = 32, bool useMinhash = false,
                                std::uint64_t index_batch_bases = 1ULL << 32, std::uint64_t flush_bases = 1ULL << 30) {
  piles.reserve(sequences.size());
  for (const auto& it : sequences) piles.emplace_back(new PileT(it->id, it->inflated_len));
  if (sequences.empty()) return;
  if (overlaps.size() < sequences.size()) overlaps.resize(sequences.size());

  ram::detail::ReadsHandle reads;
  reads.Upload(minimizer_engine.handle(), sequences.begin(), sequences.end());
  rvn_pass1* p = nullptr;
  ram::detail::Check(rvn_find_overlaps_and_create_piles(minimizer_engine.handle(), reads.h, freq,
                                                        static_cast<std::uint32_t>(kMaxNumOverlaps), useMinhash,
                                                        index_batch_bases, flush_bases, &p));
  struct Guard {
    rvn_pass1* p;
    ~Guard() { rvn_pass1_destroy(p); }
  } guard{p};

  const std::size_t n = sequences.size();
  std::vector<std::uint16_t> data(rvn_pass1_pile_words(p));
  std::vector<std::uint64_t> poff(n + 1);
  ram::detail::Check(rvn_pass1_fetch_piles(p, data.data(), poff.data()));
  for (std::size_t i = 0; i < n; ++i) piles[i]->AdoptCoverage(data.data() + poff[i], poff[i + 1] - poff[i]);

  std::vector<rvn_overlap> flat(rvn_pass1_num_overlaps(p));
  std::vector<std::uint32_t> ooff(n + 1);
  ram::detail::Check(rvn_pass1_fetch_overlaps(p, flat.data(), ooff.data()));
  for (std::size_t i = 0; i < n; ++i) {
    overlaps[i].clear();
    overlaps[i].reserve(ooff[i + 1] - ooff[i]);
    for (std::uint32_t j = ooff[i]; j < ooff[i + 1]; ++j) overlaps[i].emplace_back(ram::detail::ToOverlap(flat[j]));
  }
}

// raven::FindOverlapsAndRepetetiveRegions (RavenLib/src/construct.cc:316-491, decl construct.h:49-54) with the
// reference's signature: the second all-vs-all pass on the valid reads, one C-ABI call.  The reference re-sorts
// `sequences` valid-first for the duration of the call and restores the id order before returning (construct.cc:324-332,
// :486-490); the device pass selects the valid reads itself, so `sequences` is left as it is (ids must equal positions,
// the invariant of construct.cc:25).  Effects, as in the reference: overlaps gets the extra slot overlaps.back(),
// contained piles are marked and set invalid, Pile::kmers_ of the valid piles is filled.
// PileT must provide begin(), end(), is_invalid(), set_is_contained(), set_is_invalid() (all raven::Pile members) and
//   void AdoptKmers(const std::uint8_t* cells, std::size_t n)      // replaces Pile::kmers_ (n == (len >> 4) + 1)
template <typename PileT>
void FindOverlapsAndRepetetiveRegions(const std::shared_ptr<thread_pool::ThreadPool>& /*thread_pool*/,
                                      ram::MinimizerEngine& minimizer_engine, double freq, std::uint8_t kmer_len,
                                      double identity, const std::vector<std::unique_ptr<PileT>>& piles,
                                      std::vector<std::vector<biosoup::Overlap>>& overlaps,
                                      std::vector<std::unique_ptr<biosoup::NucleicAcid>>& sequences,
                                      std::uint64_t batch_bases = 1ULL << 30) {
  const std::size_t n = sequences.size();
  overlaps.resize(n + 1);  // construct.cc:352
  if (n == 0) return;
  ram::detail::ReadsHandle reads;
  reads.Upload(minimizer_engine.handle(), sequences.begin(), sequences.end());
  std::vector<std::uint32_t> begin(n), end(n);
  std::vector<std::uint8_t> invalid(n);
  for (std::size_t i = 0; i < n; ++i) {
    begin[i] = piles[i]->begin();
    end[i] = piles[i]->end();
    invalid[i] = piles[i]->is_invalid() ? 1 : 0;
  }
  rvn_pass2* p = nullptr;
  ram::detail::Check(rvn_find_overlaps_and_repetitive_regions(minimizer_engine.handle(), reads.h, begin.data(), end.data(),
                                                              invalid.data(), freq, kmer_len, identity, batch_bases, &p));
  struct Guard {
    rvn_pass2* p;
    ~Guard() { rvn_pass2_destroy(p); }
  } guard{p};
  std::vector<rvn_overlap> flat(rvn_pass2_num_overlaps(p));
  std::vector<std::uint8_t> contained(n), kmers(rvn_pass2_kmer_cells(p));
  std::vector<std::uint64_t> koff(n + 1);
  ram::detail::Check(rvn_pass2_fetch(p, flat.data(), contained.data(), kmers.data(), koff.data()));
  for (std::size_t i = 0; i < n; ++i) {
    if (koff[i + 1] > koff[i]) piles[i]->AdoptKmers(kmers.data() + koff[i], koff[i + 1] - koff[i]);
    if (contained[i]) {  // construct.cc:438-441, :466-470
      piles[i]->set_is_contained();
      piles[i]->set_is_invalid();
    }
  }
  auto& back = overlaps.back();
  back.reserve(back.size() + flat.size());
  for (const auto& o : flat) back.emplace_back(ram::detail::ToOverlap(o));
}

}  // namespace raven

#endif  // RAVEN_HIP_FIND_OVERLAPS_HPP_
