// racon/polisher.hpp — drop-in facade with the interface of racon::Polisher as Raven uses it
// (RavenLib/src/polish.cc:43-51: Create(thread_pool, q, e, w, trim, m, n, g, cuda_poa_batches, cuda_banded,
// cuda_aln_batches) and Polish(targets, sequences, drop_unpolished)), backed by rvn_polish_round of raven_hip.h:
// read -> target mapping, window layers, POA consensus and stitching all run on the MI355X.  Header-only; link with
// -lraven_hip.  Needs the caller's biosoup::NucleicAcid (name / deflated_data / block_quality / inflated_len and
// the (name, data) constructor), like racon's own header.
//
// What a maintainer should know:
//  * one Polish() call = one racon round; raven::Polish's loop over rounds (polish.cc:49-85) stays as it is;
//  * the three cuda_* arguments and the thread pool are accepted and ignored;
//  * result names carry racon's tags " LN:i:<len> RC:i:<reads used> XC:f:<polished window ratio>" — Raven parses the
//    float after the last ':' (polish.cc:57-59) and the node id after "Utg" (polish.cc:55);
//  * qualities: biosoup keeps one mean Phred per 64-base block (block_quality); they are expanded to per-base
//    Phred+33 for the C ABI, as racon's InflateQuality() does.  Sets without qualities use unit weights;
//  * errors of the C ABI are rethrown (std::invalid_argument for RVN_EINVAL, std::runtime_error otherwise); with
//    no GPU, Create() throws — there is no CPU fallback.
#ifndef RACON_POLISHER_HPP_  // same guard as racon's header: include one or the other
#define RACON_POLISHER_HPP_

#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ram/minimizer_engine.hpp"

namespace racon {

class Polisher {
 public:
  using Sequences = std::vector<std::unique_ptr<biosoup::NucleicAcid>>;

  static std::unique_ptr<Polisher> Create(std::shared_ptr<thread_pool::ThreadPool> /*thread_pool*/ = nullptr,
                                          double quality_threshold = 10.0, double error_threshold = 0.3,
                                          std::uint32_t window_len = 500, bool trim_consensus = true,
                                          std::int8_t match = 3, std::int8_t mismatch = -5, std::int8_t gap = -4,
                                          std::uint32_t /*cuda_poa_batches*/ = 0, bool /*cuda_banded*/ = false,
                                          std::uint32_t /*cuda_aln_batches*/ = 0, int device = 0) {
    if (window_len == 0) throw std::invalid_argument("[racon::Polisher::Create] error: invalid window length");
    if (gap > 0) throw std::invalid_argument("[racon::Polisher::Create] error: gap penalty must be non-positive");
    return std::unique_ptr<Polisher>(
        new Polisher(quality_threshold, error_threshold, window_len, trim_consensus, match, mismatch, gap, device));
  }

  Polisher(const Polisher&) = delete;
  Polisher& operator=(const Polisher&) = delete;
  ~Polisher() {
    rvn_reads_destroy(reads_.h);  // before the engine it belongs to
    reads_.h = nullptr;
    rvn_engine_destroy(engine_);
  }

  // racon: one polishing round of `targets` with `sequences`; targets without a polished window are dropped when
  // drop_unpolished (polish.cc:51 passes false)
  Sequences Polish(const Sequences& targets, const Sequences& sequences, bool drop_unpolished) {
    Sequences dst;
    if (targets.empty()) return dst;
    // raven::Polish hands round r's result to round r + 1 as targets (polish.cc:50-52: `unitigs.swap(polished)`), rotated
    // in place where a unitig is circular (polish.cc:60-65).  The engine still holds the last round's consensus in HBM: when
    // the targets ARE that result — same number, lengths and 2-bit words of every sequence — the target set is made from
    // there (rvn_polish_output_as_reads) instead of packing and uploading it again; anything else takes the upload.
    ram::detail::ReadsHandle t;
    bool resident = last_out_.size() == targets.size() && !last_out_.empty();
    for (std::size_t i = 0; resident && i < targets.size(); ++i)
      resident = last_out_[i] == Fingerprint(*targets[i]);
    if (!resident || rvn_polish_output_as_reads(engine_, &t.h) != 0) {
      rvn_reads_destroy(t.h);  // (nothing resident any more — the engine gave its scratch back —: the host's copy)
      t.h = nullptr;
      t.Upload(engine_, targets.begin(), targets.end());
    } else {
      ++resident_rounds_;
    }
    last_out_.clear();
    // raven::Polish calls Polish() once per round with the SAME read set (polish.cc:50-51): upload it (and expand
    // its qualities) only when the container changes
    // (same storage, same size AND same content fingerprint: a caller may refill the vector in place)
    const std::uint64_t fp = Fingerprint(sequences);
    if (reads_.h == nullptr || reads_key_ != sequences.data() || reads_n_ != sequences.size() || reads_fp_ != fp) {
      reads_.Upload(engine_, sequences.begin(), sequences.end());
      reads_key_ = sequences.data();
      reads_n_ = sequences.size();
      reads_fp_ = fp;
      // biosoup keeps one mean Phred per 64 bases (block_quality) and racon only ever sees that mean: the block
      // bytes go to HBM once, beside the bases, and stay there for every round — only if every read has them
      has_q_ = !sequences.empty();
      for (const auto& s : sequences) has_q_ = has_q_ && !s->block_quality.empty();
      if (has_q_) {
        std::vector<std::uint8_t> quals;
        std::vector<std::uint64_t> qoff(1, 0);
        for (const auto& s : sequences) {
          const std::size_t nb = (static_cast<std::size_t>(s->inflated_len) + 63) / 64;
          for (std::size_t i = 0; i < nb; ++i)
            quals.push_back(static_cast<std::uint8_t>((i < s->block_quality.size() ? s->block_quality[i] : 0) + 33));
          qoff.push_back(quals.size());
        }
        ram::detail::Check(rvn_reads_attach_quality(engine_, reads_.h, quals.data(), qoff.data(), 6));
      }
    }
    ram::detail::ReadsHandle& r = reads_;

    const std::size_t n = targets.size();
    std::vector<std::uint64_t> ooff(n + 1, 0);
    for (std::size_t i = 0; i < n; ++i) ooff[i + 1] = ooff[i] + 2ULL * targets[i]->inflated_len + 1024;
    std::vector<std::uint8_t> codes(ooff[n] + 1);
    std::vector<std::uint32_t> len(n), used(n);
    std::vector<double> ratio(n);
    rvn_polish_stats st{};
    // qualities: the ones attached to the read set (none attached = unit weights, no quality filter)
    ram::detail::Check(rvn_polish_round(engine_, t.h, r.h, nullptr, nullptr, q_, e_, w_, trim_ ? 1 : 0, m_, n_, g_,
                                        codes.data(), ooff.data(), len.data(), ratio.data(), &st));
    ram::detail::Check(rvn_polish_target_reads(engine_, used.data(), static_cast<std::uint32_t>(n)));
    for (std::size_t i = 0; i < n; ++i) {
      if (drop_unpolished && ratio[i] == 0.0) continue;
      std::string data(len[i], 'A');
      for (std::uint32_t j = 0; j < len[i]; ++j) data[j] = "ACGT"[codes[ooff[i] + j] & 3];
      char tags[96];
      std::snprintf(tags, sizeof(tags), " LN:i:%u RC:i:%u XC:f:%.6f", len[i], used[i], ratio[i]);
      const std::string& name = targets[i]->name;
      dst.emplace_back(new biosoup::NucleicAcid(name.substr(0, name.find(' ')) + tags, data));
    }
    if (dst.size() == n)  // (every target came back: what the engine holds is this result, sequence for sequence)
      for (const auto& it : dst) last_out_.push_back(Fingerprint(*it));
    return dst;
  }

  rvn_engine* handle() const { return engine_; }
  // rounds whose targets were taken from the consensus the engine held in HBM (diagnostics, tests)
  std::size_t resident_rounds() const { return resident_rounds_; }
  // forget the device copy of the read set (the next Polish() uploads again)
  void Invalidate() {
    rvn_reads_destroy(reads_.h);
    reads_.h = nullptr;
  }

 private:
  // length and every 2-bit word of one sequence (FNV-1a)
  static std::pair<std::uint32_t, std::uint64_t> Fingerprint(const biosoup::NucleicAcid& s) {
    std::uint64_t h = 1469598103934665603ULL;
    for (std::uint64_t w : s.deflated_data) {
      h ^= w;
      h *= 1099511628211ULL;
    }
    return {s.inflated_len, h};
  }
  // cheap content fingerprint of a read set: ids, lengths and one data word of every read (FNV-1a)
  static std::uint64_t Fingerprint(const Sequences& sequences) {
    std::uint64_t h = 1469598103934665603ULL;
    auto mix = [&h](std::uint64_t v) {
      h ^= v;
      h *= 1099511628211ULL;
    };
    for (const auto& s : sequences) {
      mix(s->id);
      mix(s->inflated_len);
      if (!s->deflated_data.empty()) mix(s->deflated_data[s->deflated_data.size() / 2]);
    }
    return h;
  }

  Polisher(double q, double e, std::uint32_t w, bool trim, std::int8_t m, std::int8_t n, std::int8_t g, int device)
      : q_(q), e_(e), w_(w), trim_(trim), m_(m), n_(n), g_(g) {
    // racon maps with ram's (k = 15, w = 5, bandwidth 500, chain 4, matches 100, gap 10000)
    ram::detail::Check(rvn_engine_create(&engine_, 15, 5, 500, 4, 100, 10000, device));
  }

  double q_, e_;
  std::uint32_t w_;
  bool trim_;
  int m_, n_, g_;
  rvn_engine* engine_ = nullptr;
  // read set of the previous Polish() call, kept on the device
  ram::detail::ReadsHandle reads_;
  const void* reads_key_ = nullptr;
  std::size_t reads_n_ = 0;
  std::uint64_t reads_fp_ = 0;
  bool has_q_ = false;
  // (length, content hash) of every sequence the last Polish() returned — empty when it dropped any
  std::vector<std::pair<std::uint32_t, std::uint64_t>> last_out_;
  std::size_t resident_rounds_ = 0;
};

}  // namespace racon

#endif  // RACON_POLISHER_HPP_
