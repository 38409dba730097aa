// raven_hip/multi_gpu.hpp — the two hot calls of Raven on ALL the GPUs of a node from one host process:
// raven::FindOverlapsAndCreatePiles (RavenLib/src/construct.cc:14-121) and one racon round of raven::Polish
// (RavenLib/src/polish.cc:50-51) over a raven::DeviceGroup, i.e. rvn_group_* of raven_hip.h (one engine + one worker
// thread per device; reads sharded by pile, minimizers by hash class, three in-process exchanges per flush window;
// polishing: reads mapped by slice, windows by range).  Same signatures as the single-device templates of
// raven_hip/find_overlaps.hpp with the group in the engine's place; results are bit-identical to them.
// Header-only; link with -lraven_hip.  A torch.distributed job (one process per GPU over RCCL) uses raven_amd/sharded.py
// on the same stage entry points instead.
#ifndef RAVEN_HIP_MULTI_GPU_HPP_
#define RAVEN_HIP_MULTI_GPU_HPP_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ram/minimizer_engine.hpp"

namespace raven {

// The engines of a node.  devices empty: RVN_DEVICES ("0,1,2,3") if set, else every visible device.  An ordinal may
// repeat (virtual ranks on one GPU).
class DeviceGroup {
 public:
  explicit DeviceGroup(std::vector<int> devices = {}, std::uint32_t k = 15, std::uint32_t w = 5,
                       std::uint32_t bandwidth = 500, std::uint32_t chain = 4, std::uint32_t matches = 100,
                       std::uint32_t gap = 10000) {
    if (devices.empty()) {
      if (const char* env = std::getenv("RVN_DEVICES")) {
        for (const char* p = env; *p;) {
          devices.push_back(std::atoi(p));
          while (*p && *p != ',') ++p;
          if (*p == ',') ++p;
        }
      } else {
        for (int d = 0; d < rvn_device_count(); ++d) devices.push_back(d);
      }
    }
    if (devices.empty()) throw std::runtime_error("[raven_hip] no HIP device available (this library has no CPU path)");
    ram::detail::Check(rvn_group_create(&group_, k, w, bandwidth, chain, matches, gap, devices.data(),
                                        static_cast<std::uint32_t>(devices.size())));
  }
  DeviceGroup(const DeviceGroup&) = delete;
  DeviceGroup& operator=(const DeviceGroup&) = delete;
  ~DeviceGroup() { rvn_group_destroy(group_); }
  rvn_group* handle() const { return group_; }
  std::uint32_t size() const { return rvn_group_size(group_); }

 private:
  rvn_group* group_ = nullptr;
};

// raven::FindOverlapsAndCreatePiles over the group.  PileT as in find_overlaps.hpp (PileT(id, len), AdoptCoverage).
// One index batch: the read set must hold fewer than 2^32 bases (construct.cc:35's batch size).
template <typename PileT>
void FindOverlapsAndCreatePiles(const std::shared_ptr<thread_pool::ThreadPool>& /*thread_pool*/, DeviceGroup& group,
                                const std::vector<std::unique_ptr<biosoup::NucleicAcid>>& sequences, double freq,
                                std::vector<std::unique_ptr<PileT>>& piles,
                                std::vector<std::vector<biosoup::Overlap>>& overlaps, std::size_t kMaxNumOverlaps = 32,
                                bool useMinhash = false, std::uint64_t flush_bases = 1ULL << 30) {
  piles.reserve(sequences.size());
  for (const auto& it : sequences) piles.emplace_back(new PileT(it->id, it->inflated_len));
  if (sequences.empty()) return;
  if (overlaps.size() < sequences.size()) overlaps.resize(sequences.size());
  const std::size_t n = sequences.size();
  ram::detail::PackedReads<decltype(sequences.begin())> p(sequences.begin(), sequences.end());
  const std::uint32_t world = group.size();
  std::vector<std::uint32_t> bounds(world + 1);
  std::vector<rvn_pass1*> passes(world, nullptr);
  ram::detail::Check(rvn_group_find_overlaps_and_create_piles(group.handle(), p.packed.data(), p.word_offsets.data(),
                                                              p.lengths.data(), static_cast<std::uint32_t>(n), freq,
                                                              static_cast<std::uint32_t>(kMaxNumOverlaps), useMinhash ? 1 : 0,
                                                              flush_bases, bounds.data(), passes.data()));
  struct Guard {
    std::vector<rvn_pass1*>& v;
    ~Guard() {
      for (rvn_pass1* x : v) rvn_pass1_destroy(x);
    }
  } guard{passes};
  std::vector<std::uint16_t> data;
  std::vector<std::uint64_t> poff(n + 1);
  std::vector<rvn_overlap> flat;
  std::vector<std::uint32_t> ooff(n + 1);
  for (std::uint32_t r = 0; r < world; ++r) {  // every rank's handle is complete for its own read range
    data.resize(rvn_pass1_pile_words(passes[r]));
    ram::detail::Check(rvn_pass1_fetch_piles(passes[r], data.data(), poff.data()));
    flat.resize(rvn_pass1_num_overlaps(passes[r]));
    ram::detail::Check(rvn_pass1_fetch_overlaps(passes[r], flat.data(), ooff.data()));
    for (std::size_t i = bounds[r]; i < bounds[r + 1]; ++i) {
      piles[i]->AdoptCoverage(data.data() + poff[i], poff[i + 1] - poff[i]);
      overlaps[i].clear();
      overlaps[i].reserve(ooff[i + 1] - ooff[i]);
      for (std::uint32_t j = ooff[i]; j < ooff[i + 1]; ++j) overlaps[i].emplace_back(ram::detail::ToOverlap(flat[j]));
    }
  }
}

// One racon round (racon::Polisher::Polish, polish.cc:51) over the group: same result names and tags as
// racon/polisher.hpp.  Qualities as there: when EVERY sequence carries biosoup's block_quality the round is the FASTQ
// variant (racon's mean-quality filter against quality_threshold — the avg_q of polish.cc:26-41 — and quality-weighted
// edges); otherwise unit weights and no filter.
inline std::vector<std::unique_ptr<biosoup::NucleicAcid>> PolishRound(
    DeviceGroup& group, const std::vector<std::unique_ptr<biosoup::NucleicAcid>>& targets,
    const std::vector<std::unique_ptr<biosoup::NucleicAcid>>& sequences, bool drop_unpolished, double error_threshold = 0.3,
    std::uint32_t window_len = 500, bool trim = true, std::int8_t match = 3, std::int8_t mismatch = -5, std::int8_t gap = -4,
    double quality_threshold = 10.0) {
  std::vector<std::unique_ptr<biosoup::NucleicAcid>> dst;
  if (targets.empty()) return dst;
  ram::detail::PackedReads<decltype(targets.begin())> t(targets.begin(), targets.end());
  ram::detail::PackedReads<decltype(sequences.begin())> r(sequences.begin(), sequences.end());
  const std::size_t n = targets.size();
  std::vector<std::uint64_t> ooff(n + 1, 0);
  for (std::size_t i = 0; i < n; ++i) ooff[i + 1] = ooff[i] + 2ULL * targets[i]->inflated_len + 1024;
  std::vector<std::uint8_t> codes(ooff[n] + 1);
  std::vector<std::uint32_t> len(n);
  std::vector<double> ratio(n);
  // biosoup keeps one mean Phred per 64 bases (block_quality) and racon only ever sees that mean: block bytes + 33 go over
  // the boundary as they are (block shift 6), as in racon/polisher.hpp
  bool has_q = !sequences.empty();
  for (const auto& s : sequences) has_q = has_q && !s->block_quality.empty();
  std::vector<std::uint8_t> quals;
  std::vector<std::uint64_t> qoff(1, 0);
  if (has_q) {
    for (const auto& s : sequences) {
      const std::size_t blocks = (static_cast<std::size_t>(s->inflated_len) + 63) / 64;
      for (std::size_t i = 0; i < blocks; ++i)
        quals.push_back(static_cast<std::uint8_t>((i < s->block_quality.size() ? s->block_quality[i] : 0) + 33));
      qoff.push_back(quals.size());
    }
  }
  ram::detail::Check(rvn_group_polish_round_q(group.handle(), t.packed.data(), t.word_offsets.data(), t.lengths.data(),
                                              static_cast<std::uint32_t>(n), r.packed.data(), r.word_offsets.data(),
                                              r.lengths.data(), static_cast<std::uint32_t>(sequences.size()),
                                              has_q ? quals.data() : nullptr, has_q ? qoff.data() : nullptr, 6,
                                              has_q ? quality_threshold : 0.0, error_threshold, window_len, trim ? 1 : 0, match,
                                              mismatch, gap, codes.data(), ooff.data(), len.data(), ratio.data()));
  // reads used per target (racon's RC:i: tag): every rank holds the complete best-overlap table, so rank 0's counts are
  // the round's
  std::vector<std::uint32_t> used(n, 0);
  ram::detail::Check(rvn_polish_target_reads(rvn_group_engine(group.handle(), 0), used.data(), static_cast<std::uint32_t>(n)));
  for (std::size_t i = 0; i < n; ++i) {
    if (drop_unpolished && ratio[i] == 0.0) continue;
    std::string data(len[i], 'A');
    for (std::uint32_t j = 0; j < len[i]; ++j) data[j] = "ACGT"[codes[ooff[i] + j] & 3];
    char tags[96];
    std::snprintf(tags, sizeof(tags), " LN:i:%u RC:i:%u XC:f:%.6f", len[i], used[i], ratio[i]);
    const std::string& name = targets[i]->name;
    dst.emplace_back(new biosoup::NucleicAcid(name.substr(0, name.find(' ')) + tags, data));
  }
  return dst;
}

}  // namespace raven

#endif  // RAVEN_HIP_MULTI_GPU_HPP_
