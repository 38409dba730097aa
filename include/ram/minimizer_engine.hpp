// ram/minimizer_engine.hpp — drop-in facade with the interface of ram::MinimizerEngine as Raven uses it
// (RavenLib/src/construct.cc:42-44,62,363,372,377-381,661-662; RavenLib/src/assemble.cc:753-757), backed by
// the MI355X engine behind the C ABI of raven_hip.h.  Header-only; link with -lraven_hip.
//
// It needs the caller's biosoup headers (biosoup::NucleicAcid with id / deflated_data / inflated_len,
// biosoup::Overlap with the 8-argument constructor) exactly like ram's own header does.
//
// Differences a maintainer should know about:
//  * the thread pool argument is accepted and ignored (parallelism is on the device);
//  * Map() is const and thread-safe like ram's.  For a sequence of the indexed range (both of Raven's passes) the
//    first call maps that whole range in one device pass and every later call reads the cached result; any other
//    sequence is uploaded and mapped on its own.  MapBatch() / raven_hip/find_overlaps.hpp remain the fast path;
//  * errors of the C ABI are rethrown as the exception types ram/biosoup use (std::invalid_argument for
//    RVN_EINVAL, std::runtime_error otherwise).
#ifndef RAM_MINIMIZER_ENGINE_HPP_  // same guard as ram's header: include one or the other
#define RAM_MINIMIZER_ENGINE_HPP_

#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "biosoup/nucleic_acid.hpp"
#include "biosoup/overlap.hpp"
#include "raven_hip.h"

namespace thread_pool {
class ThreadPool;
}

namespace ram {

namespace detail {

inline void Check(int rc) {
  if (rc == RVN_OK) return;
  std::string msg = rvn_last_error();
  if (rc == RVN_EINVAL) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

// Concatenates deflated_data of [first, last) in the layout rvn_reads_upload expects.
template <typename It>
struct PackedReads {
  std::vector<std::uint64_t> packed, word_offsets;
  std::vector<std::uint32_t> lengths, ids;
  PackedReads(It first, It last) {
    word_offsets.push_back(0);
    for (auto it = first; it != last; ++it) {
      const auto& s = *it;
      packed.insert(packed.end(), s->deflated_data.begin(), s->deflated_data.end());
      word_offsets.push_back(packed.size());
      lengths.push_back(s->inflated_len);
      ids.push_back(s->id);
    }
    packed.push_back(0);  // pad word
  }
};

struct ReadsHandle {
  rvn_reads* h = nullptr;
  ReadsHandle() = default;
  ReadsHandle(const ReadsHandle&) = delete;
  ReadsHandle& operator=(const ReadsHandle&) = delete;
  ~ReadsHandle() { rvn_reads_destroy(h); }
  template <typename It>
  void Upload(rvn_engine* e, It first, It last) {
    rvn_reads_destroy(h);
    h = nullptr;
    PackedReads<It> p(first, last);
    Check(rvn_reads_upload(e, p.packed.data(), p.packed.size() - 1, p.word_offsets.data(), p.lengths.data(),
                           p.ids.data(), static_cast<std::uint32_t>(p.lengths.size()), &h));
  }
};

inline biosoup::Overlap ToOverlap(const rvn_overlap& o) {
  return biosoup::Overlap{o.lhs_id, o.lhs_begin, o.lhs_end, o.rhs_id, o.rhs_begin, o.rhs_end, o.score, o.strand != 0};
}

}  // namespace detail

class MinimizerEngine {
 public:
  using Sequences = std::vector<std::unique_ptr<biosoup::NucleicAcid>>;

  MinimizerEngine(std::shared_ptr<thread_pool::ThreadPool> /*thread_pool*/ = nullptr, std::uint32_t k = 15,
                  std::uint32_t w = 5, std::uint32_t bandwidth = 500, std::uint32_t chain = 4,
                  std::uint32_t matches = 100, std::uint32_t gap = 10000, int device = 0) {
    detail::Check(rvn_engine_create(&engine_, k, w, bandwidth, chain, matches, gap, device));
  }
  MinimizerEngine(const MinimizerEngine&) = delete;
  MinimizerEngine& operator=(const MinimizerEngine&) = delete;
  ~MinimizerEngine() { rvn_engine_destroy(engine_); }

  // ram: transform set of sequences to minimizer index (construct.cc:42-43, :363)
  void Minimize(Sequences::const_iterator first, Sequences::const_iterator last, bool minhash = false) {
    std::lock_guard<std::mutex> lk(mu_);
    cache_.valid = false;
    index_reads_.Upload(engine_, first, last);
    index_first_ = first == last ? nullptr : &*first;
    index_count_ = static_cast<std::size_t>(last - first);
    // what sat in every slot when the index was built: Raven re-sorts `sequences` in place (construct.cc:324, :486), and a
    // cached answer must not be served for whatever read occupies the slot afterwards
    index_print_.resize(index_count_);
    for (std::size_t i = 0; i < index_count_; ++i) index_print_[i] = Fingerprint(*(first + i));
    detail::Check(rvn_engine_minimize(engine_, index_reads_.h, 0, static_cast<std::uint32_t>(last - first), minhash));
  }

  // ram: set occurrence frequency threshold (construct.cc:44, :372); throws std::invalid_argument outside [0,1]
  void Filter(double frequency) {
    std::lock_guard<std::mutex> lk(mu_);
    cache_.valid = false;
    detail::Check(rvn_engine_filter(engine_, frequency));
  }

  // ram: find overlaps in the index (construct.cc:62, :377-381).  const and safe under concurrent callers, as
  // ram's is (Raven calls it from its pool workers).  When `sequence` is one of the sequences the index was built
  // from — which is how both of Raven's passes call it — the first call maps the WHOLE indexed range in one device
  // pass and later calls (any thread) are served from that result; other sequences are mapped one at a time.
  std::vector<biosoup::Overlap> Map(const std::unique_ptr<biosoup::NucleicAcid>& sequence, bool avoid_equal,
                                    bool avoid_symmetric, bool minhash = false,
                                    std::vector<std::uint32_t>* filtered = nullptr) const {
    const std::unique_ptr<biosoup::NucleicAcid>* first = &sequence;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (index_first_ && first >= index_first_ && first < index_first_ + index_count_ &&
          Fingerprint(sequence) == index_print_[static_cast<std::size_t>(first - index_first_)]) {
        const std::size_t i = static_cast<std::size_t>(first - index_first_);
        const int flags = (avoid_equal ? 1 : 0) | (avoid_symmetric ? 2 : 0) | (minhash ? 4 : 0);
        if (!cache_.valid || cache_.flags != flags || (filtered && !cache_.has_filtered)) {
          auto res = MapUploaded(index_reads_.h, 0, static_cast<std::uint32_t>(index_count_), avoid_equal,
                                 avoid_symmetric, minhash, filtered ? 1 : 0);
          cache_.overlaps = std::move(res.first);
          cache_.filtered = std::move(res.second);
          cache_.flags = flags;
          cache_.has_filtered = filtered != nullptr;
          cache_.served.assign(index_count_, false);
          cache_.valid = true;
        }
        // every entry is served once per pass (Raven maps each read once): hand it over instead of copying it, so the
        // host copy of a ~1 Gbase batch's overlaps shrinks as it is consumed; a second request re-maps that read singly
        if (!cache_.served[i]) {
          cache_.served[i] = true;
          if (filtered) *filtered = std::move(cache_.filtered[i]);
          return std::move(cache_.overlaps[i]);
        }
      }
    }
    detail::ReadsHandle q;
    q.Upload(engine_, first, first + 1);
    auto res = MapUploaded(q.h, 0, 1, avoid_equal, avoid_symmetric, minhash, filtered ? 1 : 0);
    if (filtered) *filtered = std::move(res.second[0]);
    return std::move(res.first[0]);
  }

  // Batched form: Map() of every sequence in [first, last) in one device pass; result[i] == Map(*(first+i)).
  std::vector<std::vector<biosoup::Overlap>> MapBatch(Sequences::const_iterator first, Sequences::const_iterator last,
                                                      bool avoid_equal, bool avoid_symmetric, bool minhash = false,
                                                      std::vector<std::vector<std::uint32_t>>* filtered = nullptr) const {
    detail::ReadsHandle q;
    q.Upload(engine_, first, last);
    auto res = MapUploaded(q.h, 0, static_cast<std::uint32_t>(last - first), avoid_equal, avoid_symmetric, minhash,
                           filtered ? 1 : 0);
    if (filtered) *filtered = std::move(res.second);
    return std::move(res.first);
  }

  rvn_engine* handle() const { return engine_; }

 private:
  // one critical section inside the library (rvn_engine_map_collect): map + fetch of THIS call's result
  std::pair<std::vector<std::vector<biosoup::Overlap>>, std::vector<std::vector<std::uint32_t>>> MapUploaded(
      rvn_reads* reads, std::uint32_t first, std::uint32_t last, bool avoid_equal, bool avoid_symmetric, bool minhash,
      int want_filtered) const {
    rvn_overlap* flat = nullptr;
    std::uint32_t *off = nullptr, *pos = nullptr, *foff = nullptr;
    detail::Check(rvn_engine_map_collect(engine_, reads, first, last, avoid_equal, avoid_symmetric, minhash,
                                         want_filtered, &flat, &off, &pos, &foff));
    struct Free {
      void* p;
      ~Free() { rvn_free(p); }
    } f0{flat}, f1{off}, f2{pos}, f3{foff};
    std::vector<std::vector<biosoup::Overlap>> out(last - first);
    for (std::uint32_t i = 0; i < last - first; ++i) {
      out[i].reserve(off[i + 1] - off[i]);
      for (std::uint32_t j = off[i]; j < off[i + 1]; ++j) out[i].emplace_back(detail::ToOverlap(flat[j]));
    }
    std::vector<std::vector<std::uint32_t>> filt;
    if (want_filtered) {
      filt.resize(last - first);
      for (std::uint32_t i = 0; i < last - first; ++i) filt[i].assign(pos + foff[i], pos + foff[i + 1]);
    }
    return {std::move(out), std::move(filt)};
  }

  struct Cache {
    bool valid = false, has_filtered = false;
    int flags = 0;
    std::vector<std::vector<biosoup::Overlap>> overlaps;
    std::vector<std::vector<std::uint32_t>> filtered;
    std::vector<bool> served;
  };
  struct Print {  // identity of the read in an indexed slot
    std::uint32_t id = 0, len = 0;
    std::uint64_t word = 0;
    bool operator==(const Print& o) const { return id == o.id && len == o.len && word == o.word; }
  };
  static Print Fingerprint(const std::unique_ptr<biosoup::NucleicAcid>& s) {
    Print p;
    if (s) {
      p.id = static_cast<std::uint32_t>(s->id);
      p.len = s->inflated_len;
      p.word = s->deflated_data.empty() ? 0 : s->deflated_data[s->deflated_data.size() / 2];
    }
    return p;
  }
  mutable std::mutex mu_;  // guards the cache and the index read set
  mutable Cache cache_;
  const std::unique_ptr<biosoup::NucleicAcid>* index_first_ = nullptr;  // the caller's sequences the index covers
  std::vector<Print> index_print_;
  std::size_t index_count_ = 0;
  rvn_engine* engine_ = nullptr;
  detail::ReadsHandle index_reads_;
};

}  // namespace ram

#endif  // RAM_MINIMIZER_ENGINE_HPP_
