// raven_hip/find_overlaps.hpp — raven::FindOverlapsAndCreatePiles (RavenLib/src/construct.cc:14-121, decl
// RavenLib/include/raven/graph/construct.h:22-29) with the reference's signature and semantics, running the
// whole pass on the MI355X through one C-ABI call.  Header-only; see INTEGRATION.md.
//
// PileT must provide PileT(std::uint32_t id, std::uint32_t len) and
//   void AdoptCoverage(const std::uint16_t* data, std::size_t n)   // replaces Pile::data_ (n == len >> 4)
// (a two-line addition to raven::Pile; AddLayers itself is no longer called on this path).
//
// Optional: pass a raven::Pass1Handle to FindOverlapsAndCreatePiles and the coverage stays in HBM for
// raven::TrimAndAnnotatePiles(thread_pool, piles, overlaps, handle) below — construct.cc:123-152 (FindValidRegion(4),
// FindMedian, FindChimericRegions of every pile) on the device instead of Pile's host loops; PileT then also provides
//   void AdoptAnnotation(std::uint32_t begin, std::uint32_t end, std::uint16_t median, bool invalid)   // cells, as Pile::begin_ / end_
//   void AdoptChimericRegions(const std::uint32_t* pairs, std::size_t n)                                // Pile::chimeric_regions_
#ifndef RAVEN_HIP_FIND_OVERLAPS_HPP_
#define RAVEN_HIP_FIND_OVERLAPS_HPP_

#include <cstdint>
#include <iostream>
#include <memory>
#include <vector>

#include "ram/minimizer_engine.hpp"

namespace raven {

// the first pass's result in HBM, kept alive between FindOverlapsAndCreatePiles and TrimAndAnnotatePiles
struct Pass1Handle {
  rvn_pass1* p = nullptr;
  Pass1Handle() = default;
  Pass1Handle(const Pass1Handle&) = delete;
  Pass1Handle& operator=(const Pass1Handle&) = delete;
  ~Pass1Handle() { rvn_pass1_destroy(p); }
};

template <typename PileT>
void FindOverlapsAndCreatePiles(const std::shared_ptr<thread_pool::ThreadPool>& /*thread_pool*/,
                                ram::MinimizerEngine& minimizer_engine,
                                const std::vector<std::unique_ptr<biosoup::NucleicAcid>>& sequences, double freq,
                                std::vector<std::unique_ptr<PileT>>& piles,
                                std::vector<std::vector<biosoup::Overlap>>& overlaps,
                                std::size_t kMaxNumOverlaps = 32, bool useMinhash = false,
                                std::uint64_t index_batch_bases = 1ULL << 32, std::uint64_t flush_bases = 1ULL << 30,
                                Pass1Handle* keep = nullptr) {
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
    Pass1Handle* keep;
    ~Guard() {
      if (keep) {
        rvn_pass1_destroy(keep->p);
        keep->p = p;
      } else {
        rvn_pass1_destroy(p);
      }
    }
  } guard{p, keep};

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

// raven::TrimAndAnnotatePiles (RavenLib/src/construct.cc:123-152) on the coverage arrays the first pass left in HBM:
// Pile::FindValidRegion(4) (+ UpdateValidRegion), FindMedian and FindChimericRegions of every pile in two device calls;
// overlaps[i] of an invalid pile is released as the reference does (:134-135).  The piles get their trimmed coverage,
// valid region, median, validity and chimeric regions through the Adopt* hooks (see the top of this file).
template <typename PileT>
void TrimAndAnnotatePiles(const std::shared_ptr<thread_pool::ThreadPool>& /*thread_pool*/,
                          const std::vector<std::unique_ptr<PileT>>& piles,
                          std::vector<std::vector<biosoup::Overlap>>& overlaps, Pass1Handle& pass) {
  const std::size_t n = piles.size();
  if (n == 0 || pass.p == nullptr) return;
  std::vector<std::uint32_t> begin(n), end(n), roff(n + 1);
  std::vector<std::uint16_t> median(n);
  std::vector<std::uint8_t> invalid(n);
  ram::detail::Check(rvn_pass1_trim_and_annotate(pass.p, 4, begin.data(), end.data(), median.data(), invalid.data()));
  std::uint32_t* regions = nullptr;
  ram::detail::Check(rvn_pass1_find_chimeric_regions(pass.p, invalid.data(), roff.data(), &regions));
  struct Free {
    void* p;
    ~Free() { rvn_free(p); }
  } free_regions{regions};
  std::vector<std::uint16_t> data(rvn_pass1_pile_words(pass.p));
  std::vector<std::uint64_t> poff(n + 1);
  ram::detail::Check(rvn_pass1_fetch_piles(pass.p, data.data(), poff.data()));
  for (std::size_t i = 0; i < n; ++i) {
    piles[i]->AdoptCoverage(data.data() + poff[i], poff[i + 1] - poff[i]);
    piles[i]->AdoptAnnotation(begin[i], end[i], median[i], invalid[i] != 0);
    piles[i]->AdoptChimericRegions(regions + 2 * static_cast<std::size_t>(roff[i]), roff[i + 1] - roff[i]);
    if (invalid[i]) std::vector<biosoup::Overlap>().swap(overlaps[i]);
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
