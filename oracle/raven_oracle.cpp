// raven_oracle.cpp — CPU restatement ("oracle") of Raven's overlap hot path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing under raven_amd/ (the product) may
// include, link, dlopen or call this file.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg use it — as the checker / the timed CPU
// baseline, never as the thing shipped.
//
// PARITY STATUS: "parity unpinned" for the parts that live in un-vendored
// third-party code (ram, biosoup, edlib): those libraries are absent from
// /root/reference (fetched by CMake FetchContent at configure time,
// Raven.deps.cmake:39-54; ram arrives transitively, RavenLib/RavenLib.cmake:39)
// and there is no network, so the restatement below follows their *published*
// algorithms (lbcb-sci/ram `MinimizerEngine`, unpinned version reached through
// racon@library; rvaser/biosoup `NucleicAcid`; Martinsos/edlib global NW
// distance) anchored on the reference's own call sites:
//   RavenLib/src/construct.cc:14-121  FindOverlapsAndCreatePiles (restated in full)
//   RavenLib/src/construct.cc:42-44   Minimize(first,last,minhash) / Filter(freq)
//   RavenLib/src/construct.cc:62      Map(seq, true, true, true)
//   RavenLib/src/pile.cc:12-62        clamp / Pile::Pile / Pile::AddLayers (in repo: pinned by source)
//   RavenLib/src/overlap_utils.cc:5-12 OverlapReverse / GetOverlapLength (in repo)
// The in-repo parts (pile.cc, overlap_utils.cc, construct.cc) are restated from
// source that IS present; they cannot be compiled from /root/reference because
// they include biosoup/cereal headers the image lacks (writing stand-in headers
// is not allowed), so they too are checked only against definitional
// properties (tests/test_oracle_*.py).
//
// Written in C++ (g++) rather than C for one reason: construct.cc:98-107 calls
// the *unstable* std::sort; which equal-length overlaps survive the top-kMax
// truncation is defined only by libstdc++'s introsort, so the oracle calls the
// same library routine the reference would be linked with.

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "poa_oracle.h"

namespace orc {

// ---------------------------------------------------------------- biosoup ----
// biosoup::NucleicAcid packing: 32 bases per uint64, base i at
// bits (2i mod 64) LSB-first, A=0 C=1 G=2 T=3.  Code(i) for the non-RC case.
struct Read {
  const std::uint64_t* words;
  std::uint32_t len;
  std::uint32_t id;
  inline std::uint64_t Code(std::uint32_t i) const {
    return (words[i >> 5] >> ((i << 1) & 63)) & 3;
  }
};

// biosoup::Overlap without the alignment string: 8 x u32.
struct Overlap {
  std::uint32_t lhs_id, lhs_begin, lhs_end;
  std::uint32_t rhs_id, rhs_begin, rhs_end;
  std::uint32_t score;
  std::uint32_t strand;
};

// overlap_utils.cc:5-8
static inline Overlap OverlapReverse(const Overlap& o) {
  return {o.rhs_id, o.rhs_begin, o.rhs_end, o.lhs_id,
          o.lhs_begin, o.lhs_end, o.score, o.strand};
}
// overlap_utils.cc:10-12
static inline std::uint32_t GetOverlapLength(const Overlap& o) {
  return std::max(o.rhs_end - o.rhs_begin, o.lhs_end - o.lhs_begin);
}

// -------------------------------------------------------------------- ram ----
struct Kmer {
  std::uint64_t value;
  std::uint64_t origin;  // id << 32 | position << 1 | strand
  std::uint32_t id() const { return static_cast<std::uint32_t>(origin >> 32); }
  std::uint32_t position() const { return static_cast<std::uint32_t>(origin) >> 1; }
  bool strand() const { return origin & 1; }
};
struct Match {
  std::uint64_t group;      // (rhs_id << 1 | same_strand) << 32 | diagonal
  std::uint64_t positions;  // lhs_pos << 32 | rhs_pos
  std::uint32_t rhs_id() const { return static_cast<std::uint32_t>(group >> 33); }
  bool strand() const { return (group >> 32) & 1; }
  std::uint32_t lhs_position() const { return static_cast<std::uint32_t>(positions >> 32); }
  std::uint32_t rhs_position() const { return static_cast<std::uint32_t>(positions); }
};

// ram's RadixSort: stable LSD byte-wise counting sort on the low `max_bits`
// (rounded up to whole bytes) of key(x).
template <typename T, typename KeyFn>
static void RadixSort(T* first, T* last, std::uint8_t max_bits, KeyFn key) {
  if (first >= last) return;
  std::size_t n = last - first;
  std::vector<T> tmp(n);
  T* src = first;
  T* dst = tmp.data();
  for (std::uint32_t shift = 0; shift < max_bits; shift += 8) {
    std::size_t counts[256] = {};
    for (std::size_t i = 0; i < n; ++i) ++counts[(key(src[i]) >> shift) & 0xFF];
    std::size_t sum = 0;
    for (auto& c : counts) { std::size_t t = c; c = sum; sum += t; }
    for (std::size_t i = 0; i < n; ++i) dst[counts[(key(src[i]) >> shift) & 0xFF]++] = src[i];
    std::swap(src, dst);
  }
  if (src != first) std::memcpy(first, src, n * sizeof(T));
}

struct Counters {
  std::uint64_t index_bases = 0;      // N of the Minimize calls
  std::uint64_t index_minimizers = 0; // M_i
  std::uint64_t index_keys = 0;       // U
  std::uint64_t query_bases = 0;
  std::uint64_t query_minimizers = 0; // M_q
  std::uint64_t matches = 0;          // H
  std::uint64_t overlaps = 0;         // O (Map outputs)
};

class MinimizerEngine {
 public:
  MinimizerEngine(std::uint32_t k, std::uint32_t w, std::uint32_t bandwidth,
                  std::uint32_t chain, std::uint32_t matches, std::uint32_t gap)
      : k_(std::min(std::max(k, 1U), 31U)),
        w_(w),
        bandwidth_(bandwidth),
        chain_(chain),
        matches_(matches),
        gap_(gap),
        occurrence_(-1),
        index_(1U << std::min(14U, 2 * k_)) {}

  static std::uint64_t Hash(std::uint64_t key, std::uint64_t mask) {
    key = ((~key) + (key << 21)) & mask;
    key = key ^ (key >> 24);
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ (key >> 14);
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ (key >> 28);
    key = (key + (key << 31)) & mask;
    return key;
  }

  // ram MinimizerEngine::Minimize(sequence, minhash) — rolling canonical k-mer
  // hash + monotone deque, robust winnowing (all ties of the window minimum are
  // emitted once).  DEVIATION (degenerate inputs only): ram resizes the minhash
  // sketch to len/k even when fewer minimizers exist, which would append
  // zero-valued dummies; we clamp to the sketch size instead.
  std::vector<Kmer> Minimize(const Read& sequence, bool minhash) const {
    std::vector<Kmer> dst;
    if (sequence.len < k_) return dst;

    std::uint64_t mask = (1ULL << (k_ * 2)) - 1;
    std::deque<Kmer> window;
    auto window_add = [&](std::uint64_t value, std::uint64_t location) {
      while (!window.empty() && window.back().value > value) window.pop_back();
      window.push_back(Kmer{value, location});
    };
    auto window_update = [&](std::uint32_t position) {
      while (!window.empty() && window.front().position() < position) window.pop_front();
    };

    std::uint64_t shift = (k_ - 1) * 2;
    std::uint64_t minimizer = 0, reverse_minimizer = 0;
    std::uint64_t id = static_cast<std::uint64_t>(sequence.id) << 32;
    const std::uint64_t is_stored = 1ULL << 63;

    for (std::uint32_t i = 0; i < sequence.len; ++i) {
      std::uint64_t c = sequence.Code(i);
      minimizer = ((minimizer << 2) | c) & mask;
      reverse_minimizer = (reverse_minimizer >> 2) | ((c ^ 3) << shift);
      if (i >= k_ - 1U) {
        if (minimizer < reverse_minimizer) {
          window_add(Hash(minimizer, mask), (i - (k_ - 1U)) << 1 | 0);
        } else if (minimizer > reverse_minimizer) {
          window_add(Hash(reverse_minimizer, mask), (i - (k_ - 1U)) << 1 | 1);
        }
      }
      if (i >= (k_ - 1U) + (w_ - 1U)) {
        for (auto it = window.begin(); it != window.end(); ++it) {
          if (it->value != window.front().value) break;
          if (it->origin & is_stored) continue;
          dst.push_back(Kmer{it->value, id | it->origin});
          it->origin |= is_stored;
        }
        window_update(i - (k_ - 1U) - (w_ - 1U) + 1);
      }
    }

    if (minhash) {
      RadixSort(dst.data(), dst.data() + dst.size(), k_ * 2,
                [](const Kmer& x) { return x.value; });
      dst.resize(std::min<std::size_t>(dst.size(), sequence.len / k_));
      RadixSort(dst.data(), dst.data() + dst.size(), 64,
                [](const Kmer& x) { return x.origin; });
    }
    return dst;
  }

  // ram MinimizerEngine::Minimize(first, last, minhash): bucket by low bits,
  // stable sort each bucket by value, build origins + locator.
  void Minimize(const Read* first, const Read* last, bool minhash, unsigned n_threads) {
    for (auto& it : index_) {
      it.origins.clear();
      it.locator.clear();
    }
    if (first >= last) return;

    std::vector<std::vector<Kmer>> minimizers(index_.size());
    std::uint64_t mask = index_.size() - 1;
    std::size_t n = last - first;
    {
      std::vector<std::vector<Kmer>> sketches(n);
      ParallelFor(n, n_threads, [&](std::size_t i) { sketches[i] = Minimize(first[i], minhash); });
      for (std::size_t i = 0; i < n; ++i) {
        counters.index_bases += first[i].len;
        counters.index_minimizers += sketches[i].size();
        for (const auto& jt : sketches[i]) minimizers[jt.value & mask].push_back(jt);
        std::vector<Kmer>().swap(sketches[i]);
      }
    }
    ParallelFor(minimizers.size(), n_threads, [&](std::size_t i) {
      auto& m = minimizers[i];
      if (m.empty()) return;
      RadixSort(m.data(), m.data() + m.size(), k_ * 2, [](const Kmer& x) { return x.value; });
      m.push_back(Kmer{~0ULL, ~0ULL});  // stop dummy
      auto& idx = index_[i];
      for (std::uint64_t j = 1, c = 1; j < m.size(); ++j, ++c) {
        if (m[j - 1].value != m[j].value) {
          if (c == 1) {
            idx.locator.emplace(m[j - 1].value << 1 | 1, m[j - 1].origin);
          } else {
            idx.locator.emplace(m[j - 1].value << 1, idx.origins.size() << 32 | c);
            for (std::uint64_t k = j - c; k < j; ++k) idx.origins.push_back(m[k].origin);
          }
          c = 0;
        }
      }
      std::vector<Kmer>().swap(m);
    });
    for (const auto& it : index_) counters.index_keys += it.locator.size();
  }

  // ram MinimizerEngine::Filter: returns false where ram throws invalid_argument.
  bool Filter(double frequency) {
    if (!(0 <= frequency && frequency <= 1)) return false;
    if (frequency == 0) {
      occurrence_ = -1;
      return true;
    }
    std::vector<std::uint32_t> occurrences;
    for (const auto& it : index_) {
      for (const auto& jt : it.locator) {
        if (jt.first & 1) occurrences.push_back(1);
        else occurrences.push_back(static_cast<std::uint32_t>(jt.second));
      }
    }
    if (occurrences.empty()) {
      occurrence_ = -1;
      return true;
    }
    std::size_t nth = static_cast<std::size_t>((1 - frequency) * occurrences.size());
    if (nth >= occurrences.size()) nth = occurrences.size() - 1;  // ram: UB when f rounds to 0
    std::nth_element(occurrences.begin(), occurrences.begin() + nth, occurrences.end());
    occurrence_ = occurrences[nth] + 1;
    return true;
  }

  std::uint32_t Find(std::uint64_t key, const std::uint64_t** dst) const {
    const auto& idx = index_[key & (index_.size() - 1)];
    auto it = idx.locator.find(key << 1);
    if (it == idx.locator.end()) {
      it = idx.locator.find(key << 1 | 1);
      if (it == idx.locator.end()) return 0;
    }
    if (it->first & 1) {
      *dst = &(it->second);
      return 1;
    }
    *dst = &(idx.origins[it->second >> 32]);
    return static_cast<std::uint32_t>(it->second);
  }

  // ram MinimizerEngine::Map(sequence, avoid_equal, avoid_symmetric, minhash, filtered)
  std::vector<Overlap> Map(const Read& sequence, bool avoid_equal, bool avoid_symmetric,
                           bool minhash, std::vector<std::uint32_t>* filtered,
                           Counters* ctr = nullptr, std::vector<Match>* matches_out = nullptr) const {
    auto sketch = Minimize(sequence, minhash);
    if (ctr) {
      ctr->query_bases += sequence.len;
      ctr->query_minimizers += sketch.size();
    }
    if (sketch.empty()) return {};

    std::vector<Match> matches;
    for (const auto& it : sketch) {
      const std::uint64_t* origins = nullptr;
      std::uint32_t n = Find(it.value, &origins);
      if (n > occurrence_) {
        if (filtered) filtered->push_back(it.position());
        continue;
      }
      for (std::uint32_t j = 0; j < n; ++j) {
        Kmer jt{it.value, origins[j]};
        if (avoid_equal && sequence.id == jt.id()) continue;
        if (avoid_symmetric && sequence.id > jt.id()) continue;
        std::uint64_t strand = (it.strand() & 1) == (jt.strand() & 1);
        std::uint64_t lhs_pos = it.position();
        std::uint64_t rhs_pos = jt.position();
        std::uint64_t diagonal = !strand ? rhs_pos + lhs_pos : rhs_pos - lhs_pos + (3ULL << 30);
        matches.push_back(Match{(((static_cast<std::uint64_t>(jt.id()) << 1) | strand) << 32) | diagonal,
                                (lhs_pos << 32) | rhs_pos});
      }
    }
    if (ctr) ctr->matches += matches.size();
    if (matches_out) *matches_out = matches;
    auto dst = Chain(sequence.id, std::move(matches));
    if (ctr) ctr->overlaps += dst.size();
    return dst;
  }

  // ram MinimizerEngine::Chain
  std::vector<Overlap> Chain(std::uint64_t lhs_id, std::vector<Match>&& matches) const {
    RadixSort(matches.data(), matches.data() + matches.size(), 64,
              [](const Match& m) { return m.group; });
    matches.push_back(Match{~0ULL, ~0ULL});  // stop dummy

    std::vector<std::pair<std::uint64_t, std::uint64_t>> intervals;
    for (std::uint64_t i = 1, j = 0; i < matches.size(); ++i) {
      if (matches[i].group - matches[j].group > bandwidth_) {
        if (i - j >= 4) {
          if (!intervals.empty() && intervals.back().second > j) {  // extend
            intervals.back().second = i;
          } else {  // new
            intervals.emplace_back(j, i);
          }
        }
        ++j;
        while (j < i && matches[i].group - matches[j].group > bandwidth_) ++j;
      }
    }

    std::vector<Overlap> dst;
    for (const auto& it : intervals) {
      std::uint64_t j = it.first;
      std::uint64_t i = it.second;
      if (i - j < chain_) continue;

      RadixSort(matches.data() + j, matches.data() + i, 64,
                [](const Match& m) { return m.positions; });

      std::uint64_t strand = matches[j].strand();
      std::vector<std::uint64_t> indices;
      if (strand) {
        indices = LongestSubsequence(matches.data() + j, matches.data() + i, std::less<std::uint64_t>());
      } else {
        indices = LongestSubsequence(matches.data() + j, matches.data() + i, std::greater<std::uint64_t>());
      }
      if (indices.size() < chain_) continue;

      indices.push_back(matches.size() - 1 - j);  // stop dummy from above
      for (std::uint64_t k = 1, l = 0; k < indices.size(); ++k) {
        if (matches[j + indices[k]].lhs_position() - matches[j + indices[k - 1]].lhs_position() > gap_) {
          if (k - l < chain_) {
            l = k;
            continue;
          }
          std::uint32_t lhs_matches = 0, lhs_begin = 0, lhs_end = 0;
          std::uint32_t rhs_matches = 0, rhs_begin = 0, rhs_end = 0;
          for (std::uint64_t m = l; m < k; ++m) {
            std::uint32_t lhs_pos = matches[j + indices[m]].lhs_position();
            if (lhs_pos > lhs_end) {
              lhs_matches += lhs_end - lhs_begin;
              lhs_begin = lhs_pos;
            }
            lhs_end = lhs_pos + k_;

            std::uint32_t rhs_pos = matches[j + indices[m]].rhs_position();
            rhs_pos = strand ? rhs_pos : (1U << 31) - (rhs_pos + k_ - 1);
            if (rhs_pos > rhs_end) {
              rhs_matches += rhs_end - rhs_begin;
              rhs_begin = rhs_pos;
            }
            rhs_end = rhs_pos + k_;
          }
          lhs_matches += lhs_end - lhs_begin;
          rhs_matches += rhs_end - rhs_begin;
          if (std::min(lhs_matches, rhs_matches) < matches_) {
            l = k;
            continue;
          }
          dst.push_back(Overlap{
              static_cast<std::uint32_t>(lhs_id),
              matches[j + indices[l]].lhs_position(),
              k_ + matches[j + indices[k - 1]].lhs_position(),
              matches[j].rhs_id(),
              strand ? matches[j + indices[l]].rhs_position() : matches[j + indices[k - 1]].rhs_position(),
              k_ + (strand ? matches[j + indices[k - 1]].rhs_position() : matches[j + indices[l]].rhs_position()),
              std::min(lhs_matches, rhs_matches),
              static_cast<std::uint32_t>(strand)});
          l = k;
        }
      }
    }
    return dst;
  }

  // ram MinimizerEngine::LongestSubsequence (patience sort with ram's exact
  // binary search; the predicate is not monotone in general, so the probe
  // sequence itself is part of the specification).
  template <typename Cmp>
  static std::vector<std::uint64_t> LongestSubsequence(const Match* first, const Match* last, const Cmp& compare) {
    if (first >= last) return {};
    std::vector<std::uint64_t> minimal(last - first + 1, 0);
    std::vector<std::uint64_t> predecessor(last - first, 0);
    std::uint64_t longest = 0;
    for (auto it = first; it != last; ++it) {
      std::uint64_t lo = 1, hi = longest;
      while (lo <= hi) {
        std::uint64_t mid = lo + (hi - lo) / 2;
        if ((first + minimal[mid])->lhs_position() < it->lhs_position() &&
            compare((first + minimal[mid])->rhs_position(), it->rhs_position())) {
          lo = mid + 1;
        } else {
          hi = mid - 1;
        }
      }
      predecessor[it - first] = minimal[lo - 1];
      minimal[lo] = it - first;
      longest = std::max(longest, lo);
    }
    std::vector<std::uint64_t> dst;
    for (std::uint64_t i = 0, j = minimal[longest]; i < longest; ++i) {
      dst.push_back(j);
      j = predecessor[j];
    }
    std::reverse(dst.begin(), dst.end());
    return dst;
  }

  template <typename F>
  static void ParallelFor(std::size_t n, unsigned n_threads, F f) {
    if (n_threads <= 1 || n < 2) {
      for (std::size_t i = 0; i < n; ++i) f(i);
      return;
    }
    std::vector<std::thread> ts;
    std::size_t chunk = (n + n_threads - 1) / n_threads;
    for (unsigned t = 0; t < n_threads; ++t) {
      std::size_t b = t * chunk, e = std::min(n, b + chunk);
      if (b >= e) break;
      ts.emplace_back([=, &f] { for (std::size_t i = b; i < e; ++i) f(i); });
    }
    for (auto& t : ts) t.join();
  }

  struct Index {
    std::vector<std::uint64_t> origins;
    std::unordered_map<std::uint64_t, std::uint64_t> locator;
  };

  std::uint32_t k_, w_, bandwidth_, chain_, matches_, gap_;
  std::uint32_t occurrence_;
  std::vector<Index> index_;
  Counters counters;
};

// ------------------------------------------------------------------- pile ----
constexpr std::uint32_t kPSS = 4;  // pile.h:21

template <typename T>
static inline T clamp16(T v) {  // pile.cc:12-17
  return (v < 65535) ? v : 65535;
}

// pile.cc:33-62 restated on a bare uint16 array (`data` = Pile::data_, `id` = Pile::id_).
static void PileAddLayers(std::uint16_t* data, std::uint32_t id, const Overlap* begin, const Overlap* end) {
  if (begin >= end) return;
  std::vector<std::uint32_t> boundaries;
  for (auto it = begin; it != end; ++it) {
    if (it->lhs_id == id) {
      boundaries.push_back(((it->lhs_begin >> kPSS) + 1) << 1);
      boundaries.push_back(((it->lhs_end >> kPSS) - 1) << 1 | 1);
    } else if (it->rhs_id == id) {
      boundaries.push_back(((it->rhs_begin >> kPSS) + 1) << 1);
      boundaries.push_back(((it->rhs_end >> kPSS) - 1) << 1 | 1);
    }
  }
  std::sort(boundaries.begin(), boundaries.end());
  std::uint32_t coverage = 0;
  std::uint32_t last_boundary = 0;
  for (const auto& it : boundaries) {
    if (coverage > 0) {
      for (std::uint32_t i = last_boundary; i < (it >> 1); ++i) {
        data[i] = clamp16(data[i] + coverage);
      }
    }
    last_boundary = it >> 1;
    coverage += it & 1 ? -1 : 1;
  }
}

// --------------------------------------------- FindOverlapsAndCreatePiles ----
struct Pass1Result {
  std::vector<std::uint64_t> pile_offsets;  // n+1, in uint16 units
  std::vector<std::uint16_t> pile_data;
  std::vector<std::vector<Overlap>> overlaps;
  std::uint32_t last_occurrence = 0;
  double t_minimize = 0, t_map = 0;
};

// construct.cc:14-121 restated. `index_batch_bases` (reference 1<<32) and
// `flush_bases` (reference 1<<30) are parameters so tests can exercise the
// multi-batch paths on small inputs.
static void FindOverlapsAndCreatePiles(MinimizerEngine& engine, const std::vector<Read>& sequences,
                                       double freq, std::size_t kMaxNumOverlaps, bool useMinhash,
                                       std::uint64_t index_batch_bases, std::uint64_t flush_bases,
                                       unsigned n_threads, Pass1Result& res) {
  std::size_t n = sequences.size();
  res.pile_offsets.assign(n + 1, 0);
  for (std::size_t i = 0; i < n; ++i) res.pile_offsets[i + 1] = res.pile_offsets[i] + (sequences[i].len >> kPSS);
  res.pile_data.assign(res.pile_offsets[n], 0);
  res.overlaps.assign(n, {});
  auto& overlaps = res.overlaps;

  std::uint64_t bytes = 0;
  for (std::uint32_t i = 0, j = 0; i < n; ++i) {
    bytes += sequences[i].len;
    if (i != n - 1ULL && bytes < index_batch_bases) continue;
    bytes = 0;

    auto t0 = std::chrono::steady_clock::now();
    engine.Minimize(sequences.data() + j, sequences.data() + i + 1, useMinhash, n_threads);
    engine.Filter(freq);
    res.last_occurrence = engine.occurrence_;
    auto t1 = std::chrono::steady_clock::now();
    res.t_minimize += std::chrono::duration<double>(t1 - t0).count();

    std::vector<std::uint32_t> num_overlaps(n);
    for (std::uint32_t k = 0; k < n; ++k) num_overlaps[k] = overlaps[k].size();

    std::uint32_t flush_first = 0;
    for (std::uint32_t k = 0; k < i + 1; ++k) {
      bytes += sequences[k].len;
      if (k != i && bytes < flush_bases) continue;
      bytes = 0;

      std::size_t cnt = k + 1 - flush_first;
      std::vector<std::vector<Overlap>> results(cnt);
      std::vector<Counters> ctrs(n_threads > 1 ? n_threads : 1);
      {
        // thread_pool->Submit per read; results drained in submission order
        unsigned nt = n_threads > 1 ? n_threads : 1;
        std::vector<std::thread> ts;
        std::size_t chunk = (cnt + nt - 1) / nt;
        auto work = [&](unsigned t) {
          std::size_t b = t * chunk, e = std::min(cnt, b + chunk);
          for (std::size_t q = b; q < e; ++q)
            results[q] = engine.Map(sequences[flush_first + q], true, true, true, nullptr, &ctrs[t]);
        };
        if (nt == 1) work(0);
        else {
          for (unsigned t = 0; t < nt; ++t) ts.emplace_back(work, t);
          for (auto& t : ts) t.join();
        }
      }
      for (const auto& c : ctrs) {
        engine.counters.query_bases += c.query_bases;
        engine.counters.query_minimizers += c.query_minimizers;
        engine.counters.matches += c.matches;
        engine.counters.overlaps += c.overlaps;
      }
      for (auto& it : results) {
        for (const auto& jt : it) {
          overlaps[jt.lhs_id].push_back(jt);
          overlaps[jt.rhs_id].push_back(OverlapReverse(jt));
        }
      }
      flush_first = k + 1;

      MinimizerEngine::ParallelFor(n, n_threads, [&](std::size_t p) {
        if (overlaps[p].empty() || overlaps[p].size() == num_overlaps[p]) return;
        PileAddLayers(res.pile_data.data() + res.pile_offsets[p], sequences[p].id,
                      overlaps[p].data() + num_overlaps[p], overlaps[p].data() + overlaps[p].size());
        num_overlaps[p] = std::min(overlaps[p].size(), kMaxNumOverlaps);
        if (overlaps[p].size() < kMaxNumOverlaps) return;
        std::sort(overlaps[p].begin(), overlaps[p].end(),
                  [&](const Overlap& lhs, const Overlap& rhs) -> bool {
                    return GetOverlapLength(lhs) > GetOverlapLength(rhs);
                  });
        std::vector<Overlap> tmp;
        tmp.insert(tmp.end(), overlaps[p].begin(), overlaps[p].begin() + kMaxNumOverlaps);
        tmp.swap(overlaps[p]);
      });
    }
    auto t2 = std::chrono::steady_clock::now();
    res.t_map += std::chrono::duration<double>(t2 - t1).count();
    j = i + 1;
  }
}

// ---------------------------------------------------------- edit distance ----
// edlibAlign(default config) = global unit-cost edit distance (construct.cc:190-197).
// Any exact algorithm is equivalent; this is the textbook two-row DP.
static std::uint32_t EditDistance(const char* a, std::uint32_t n, const char* b, std::uint32_t m) {
  if (n == 0) return m;
  if (m == 0) return n;
  std::vector<std::uint32_t> prev(m + 1), cur(m + 1);
  for (std::uint32_t j = 0; j <= m; ++j) prev[j] = j;
  for (std::uint32_t i = 1; i <= n; ++i) {
    cur[0] = i;
    const char ai = a[i - 1];
    for (std::uint32_t j = 1; j <= m; ++j) {
      std::uint32_t s = prev[j - 1] + (ai != b[j - 1]);
      std::uint32_t d = prev[j] + 1, r = cur[j - 1] + 1;
      cur[j] = std::min(s, std::min(d, r));
    }
    prev.swap(cur);
  }
  return prev[m];
}

}  // namespace orc

// ------------------------------------------------------------------ C API ----
extern "C" {

struct orc_engine {
  orc::MinimizerEngine e;
  orc_engine(std::uint32_t k, std::uint32_t w, std::uint32_t b, std::uint32_t c, std::uint32_t m, std::uint32_t g)
      : e(k, w, b, c, m, g) {}
};

static std::vector<orc::Read> MakeReads(const std::uint64_t* packed, const std::uint64_t* word_offsets,
                                        const std::uint32_t* lengths, const std::uint32_t* ids, std::uint32_t n) {
  std::vector<orc::Read> r(n);
  for (std::uint32_t i = 0; i < n; ++i) r[i] = orc::Read{packed + word_offsets[i], lengths[i], ids ? ids[i] : i};
  return r;
}

orc_engine* orc_engine_create(std::uint32_t k, std::uint32_t w, std::uint32_t bandwidth, std::uint32_t chain,
                              std::uint32_t matches, std::uint32_t gap) {
  return new orc_engine(k, w, bandwidth, chain, matches, gap);
}
void orc_engine_destroy(orc_engine* e) { delete e; }

// single-read sketch; returns count (writes at most cap)
std::uint64_t orc_sketch(orc_engine* e, const std::uint64_t* words, std::uint32_t len, std::uint32_t id, int minhash,
                         std::uint64_t* values, std::uint64_t* origins, std::uint64_t cap) {
  auto s = e->e.Minimize(orc::Read{words, len, id}, minhash != 0);
  for (std::size_t i = 0; i < s.size() && i < cap; ++i) {
    values[i] = s[i].value;
    origins[i] = s[i].origin;
  }
  return s.size();
}

void orc_engine_minimize(orc_engine* e, const std::uint64_t* packed, const std::uint64_t* word_offsets,
                         const std::uint32_t* lengths, const std::uint32_t* ids, std::uint32_t first,
                         std::uint32_t last, int minhash, unsigned n_threads) {
  std::uint32_t n = last;
  auto reads = MakeReads(packed, word_offsets, lengths, ids, n);
  e->e.Minimize(reads.data() + first, reads.data() + last, minhash != 0, n_threads);
}

int orc_engine_filter(orc_engine* e, double f) { return e->e.Filter(f) ? 0 : -1; }
std::uint32_t orc_engine_occurrence(orc_engine* e) { return e->e.occurrence_; }
// tools/filter_sensitivity.py: what the uncertain +1 of Filter is worth on a data set
void orc_engine_set_occurrence(orc_engine* e, std::uint32_t occurrence) { e->e.occurrence_ = occurrence; }

// index lookup: number of origins for `value`, copies up to cap
std::uint32_t orc_engine_find(orc_engine* e, std::uint64_t value, std::uint64_t* origins, std::uint32_t cap) {
  const std::uint64_t* p = nullptr;
  std::uint32_t n = e->e.Find(value, &p);
  for (std::uint32_t i = 0; i < n && i < cap; ++i) origins[i] = p[i];
  return n;
}

// Map one read. Returns overlap count; matches (pre-chain, emission order) optionally returned.
std::uint64_t orc_engine_map(orc_engine* e, const std::uint64_t* words, std::uint32_t len, std::uint32_t id,
                             int avoid_equal, int avoid_symmetric, int minhash, orc::Overlap* out,
                             std::uint64_t cap, std::uint32_t* filtered, std::uint64_t filtered_cap,
                             std::uint64_t* n_filtered, std::uint64_t* match_groups, std::uint64_t* match_positions,
                             std::uint64_t match_cap, std::uint64_t* n_matches) {
  std::vector<std::uint32_t> filt;
  std::vector<orc::Match> matches;
  auto o = e->e.Map(orc::Read{words, len, id}, avoid_equal != 0, avoid_symmetric != 0, minhash != 0,
                    filtered || n_filtered ? &filt : nullptr, nullptr, &matches);
  for (std::size_t i = 0; i < o.size() && i < cap; ++i) out[i] = o[i];
  if (filtered)
    for (std::size_t i = 0; i < filt.size() && i < filtered_cap; ++i) filtered[i] = filt[i];
  if (n_filtered) *n_filtered = filt.size();
  if (match_groups && match_positions)
    for (std::size_t i = 0; i < matches.size() && i < match_cap; ++i) {
      match_groups[i] = matches[i].group;
      match_positions[i] = matches[i].positions;
    }
  if (n_matches) *n_matches = matches.size();
  return o.size();
}

void orc_engine_counters(orc_engine* e, std::uint64_t* out7) {
  const auto& c = e->e.counters;
  out7[0] = c.index_bases; out7[1] = c.index_minimizers; out7[2] = c.index_keys;
  out7[3] = c.query_bases; out7[4] = c.query_minimizers; out7[5] = c.matches; out7[6] = c.overlaps;
}

// Chain a caller-supplied match list (unit test hook for Chain/LIS).
std::uint64_t orc_chain(orc_engine* e, std::uint32_t lhs_id, const std::uint64_t* groups,
                        const std::uint64_t* positions, std::uint64_t n, orc::Overlap* out, std::uint64_t cap) {
  std::vector<orc::Match> m(n);
  for (std::uint64_t i = 0; i < n; ++i) m[i] = orc::Match{groups[i], positions[i]};
  auto o = e->e.Chain(lhs_id, std::move(m));
  for (std::size_t i = 0; i < o.size() && i < cap; ++i) out[i] = o[i];
  return o.size();
}

// raven::Pile::FindValidRegion(coverage) + UpdateValidRegion + FindMedian (RavenLib/src/pile.cc:122-174, called from
// TrimAndAnnotatePiles, construct.cc:131-139) on one pile given as cells [0, n) (begin_ = 0, end_ = n):
// valid region = the first longest maximal run of cells >= coverage that is TERMINATED by a lower cell (a run that
// reaches the end of the pile is not considered — the reference's loop only records a run when it finds its end);
// shorter than 1260 >> 4 cells (or none) -> invalid, data untouched; else cells outside the region are zeroed and the
// median is the element at sorted position size / 2 of the region.
void orc_pile_trim_and_median(std::uint16_t* data, std::uint32_t n, std::uint16_t coverage, std::uint32_t* begin,
                              std::uint32_t* end, std::uint16_t* median, std::uint8_t* invalid) {
  std::uint32_t best_b = 0, best_e = 0;
  std::uint32_t i = 0;
  while (i < n) {
    if (data[i] < coverage) {
      ++i;
      continue;
    }
    std::uint32_t j = i + 1;
    while (j < n && data[j] >= coverage) ++j;
    if (j == n) break;  // no terminating cell: every later start fails the same way
    if (best_e - best_b < j - i) {
      best_b = i;
      best_e = j;
    }
    i = j + 1;
  }
  *median = 0;
  if (best_b >= best_e || best_e - best_b < (1260u >> orc::kPSS)) {
    *invalid = 1;
    *begin = 0;
    *end = n;
    return;
  }
  *invalid = 0;
  for (std::uint32_t x = 0; x < best_b; ++x) data[x] = 0;
  for (std::uint32_t x = best_e; x < n; ++x) data[x] = 0;
  *begin = best_b;
  *end = best_e;
  std::vector<std::uint16_t> tmp(data + best_b, data + best_e);
  std::nth_element(tmp.begin(), tmp.begin() + tmp.size() / 2, tmp.end());
  *median = tmp[tmp.size() / 2];
}

void orc_pile_add_layers(std::uint16_t* data, std::uint32_t id, const orc::Overlap* ovl, std::uint64_t n) {
  orc::PileAddLayers(data, id, ovl, ovl + n);
}

// std::sort-based top-kMax truncation of one pile's overlap list (construct.cc:92-107).
std::uint64_t orc_truncate(orc::Overlap* ovl, std::uint64_t n, std::uint64_t kmax) {
  if (n < kmax) return n;
  std::sort(ovl, ovl + n, [](const orc::Overlap& l, const orc::Overlap& r) {
    return orc::GetOverlapLength(l) > orc::GetOverlapLength(r);
  });
  return kmax;
}

struct orc_pass1 {
  orc::Pass1Result r;
  std::vector<std::uint64_t> ovl_offsets;
  std::vector<orc::Overlap> ovl_flat;
};

orc_pass1* orc_find_overlaps_and_create_piles(orc_engine* e, const std::uint64_t* packed,
                                              const std::uint64_t* word_offsets, const std::uint32_t* lengths,
                                              const std::uint32_t* ids, std::uint32_t n, double freq,
                                              std::uint64_t kmax, int use_minhash, std::uint64_t index_batch_bases,
                                              std::uint64_t flush_bases, unsigned n_threads) {
  auto reads = MakeReads(packed, word_offsets, lengths, ids, n);
  auto* p = new orc_pass1();
  orc::FindOverlapsAndCreatePiles(e->e, reads, freq, kmax, use_minhash != 0, index_batch_bases, flush_bases,
                                  n_threads, p->r);
  p->ovl_offsets.assign(n + 1, 0);
  for (std::uint32_t i = 0; i < n; ++i) p->ovl_offsets[i + 1] = p->ovl_offsets[i] + p->r.overlaps[i].size();
  p->ovl_flat.reserve(p->ovl_offsets[n]);
  for (auto& v : p->r.overlaps) p->ovl_flat.insert(p->ovl_flat.end(), v.begin(), v.end());
  return p;
}
void orc_pass1_destroy(orc_pass1* p) { delete p; }
std::uint64_t orc_pass1_pile_words(orc_pass1* p) { return p->r.pile_data.size(); }
const std::uint16_t* orc_pass1_pile_data(orc_pass1* p) { return p->r.pile_data.data(); }
const std::uint64_t* orc_pass1_pile_offsets(orc_pass1* p) { return p->r.pile_offsets.data(); }
std::uint64_t orc_pass1_num_overlaps(orc_pass1* p) { return p->ovl_flat.size(); }
const orc::Overlap* orc_pass1_overlaps(orc_pass1* p) { return p->ovl_flat.data(); }
const std::uint64_t* orc_pass1_overlap_offsets(orc_pass1* p) { return p->ovl_offsets.data(); }
std::uint32_t orc_pass1_occurrence(orc_pass1* p) { return p->r.last_occurrence; }
double orc_pass1_t_minimize(orc_pass1* p) { return p->r.t_minimize; }
double orc_pass1_t_map(orc_pass1* p) { return p->r.t_map; }

// raven::Pile::AddKmers (RavenLib/src/pile.cc:64-120), literal: `kmers_cells` has cells + 1 entries
// (pile.cc:71: kmers_.resize(data_.size() + 1)); positions are the `filtered` output of Map.
void orc_pile_add_kmers(const std::uint64_t* words, std::uint32_t len, const std::uint32_t* positions,
                        std::uint64_t n, std::uint32_t kmer_len, std::uint8_t* kmers_cells) {
  orc::Read r{words, len, 0};
  for (std::uint64_t q = 0; q < n; ++q) {
    const std::uint32_t it = positions[q];
    std::string kmer;
    for (std::uint32_t i = 0; i < kmer_len; ++i) kmer += "ACGT"[r.Code(it + i)];  // InflateData(it, kmer_len)
    std::vector<std::string> polymers;
    for (const auto& c : kmer) polymers.emplace_back(1, c);
    polymers.erase(std::unique(polymers.begin(), polymers.end()), polymers.end());
    kmer.clear();
    for (const auto& p : polymers) kmer += p;
    if (kmer.size() < kmer_len / 2 + 1) continue;
    polymers.clear();
    for (auto jt = kmer.begin(); jt != kmer.end(); ++jt) {
      if ((jt - kmer.begin()) % 2 == 1) polymers.back() += *jt;
      else polymers.emplace_back(1, *jt);
    }
    polymers.erase(std::unique(polymers.begin(), polymers.end()), polymers.end());
    kmer.clear();
    for (const auto& p : polymers) kmer += p;
    if (kmer.size() < kmer_len / 2 + 1) continue;
    polymers.clear();
    for (auto jt = kmer.begin(); jt != kmer.end(); ++jt) {
      if (!polymers.empty() && (jt - kmer.begin()) % 2 == 0) polymers.back() += *jt;
      else polymers.emplace_back(1, *jt);
    }
    polymers.erase(std::unique(polymers.begin(), polymers.end()), polymers.end());
    kmer.clear();
    for (const auto& p : polymers) kmer += p;
    if (kmer.size() < kmer_len / 2 + 1) continue;
    kmers_cells[it >> orc::kPSS] = 1;
  }
}

}  // extern "C" (reopened below)

// ---- Pile::FindSlopes / FindChimericRegions / MergeRegions (RavenLib/src/pile.cc:176-187, :373-400, :403-600; the file
// is in the reference tree) restated with the reference's own data structures (monotone deques, std::sort, vectors) ----
namespace orc {

using Region = std::pair<std::uint32_t, std::uint32_t>;

static std::vector<Region> MergeRegions(const std::vector<Region>& src) {  // pile.cc:373-400
  std::vector<Region> dst;
  std::vector<bool> is_merged(src.size(), 0);
  for (std::uint32_t i = 0; i < src.size(); ++i) {
    if (is_merged[i]) continue;
    Region r = src[i];
    while (true) {
      is_merged[i] = false;
      for (std::uint32_t j = i + 1; j < src.size(); ++j) {
        if (is_merged[j]) continue;
        if (r.first < src[j].second && r.second > src[j].first) {
          is_merged[i] = true;
          is_merged[j] = true;
          r.first = std::min(r.first, src[j].first);
          r.second = std::max(r.second, src[j].second);
        }
      }
      if (!is_merged[i]) break;
    }
    dst.emplace_back(r);
  }
  return dst;
}

static std::vector<Region> FindSlopes(const std::vector<std::uint16_t>& data_, double q) {  // pile.cc:403-600
  using Subpile = std::deque<std::pair<std::int32_t, std::uint16_t>>;
  auto subpile_add = [](Subpile& s, std::uint16_t value, std::int32_t position) -> void {
    while (!s.empty() && s.back().second <= value) s.pop_back();
    s.emplace_back(position, value);
  };
  auto subpile_update = [](Subpile& s, std::int32_t position) {
    while (!s.empty() && s.front().first <= position) s.pop_front();
  };
  std::vector<Region> dst;
  std::int32_t w = 847 >> kPSS;
  std::int32_t data_size = data_.size();
  Subpile left_subpile;
  std::uint32_t first_down = 0, last_down = 0;
  bool found_down = false;
  Subpile right_subpile;
  std::uint32_t first_up = 0, last_up = 0;
  bool found_up = false;
  for (std::int32_t i = 0; i < w && i < data_size; ++i) subpile_add(right_subpile, data_[i], i);  // (reference: i < w only)
  for (std::int32_t i = 0; i < data_size; ++i) {
    if (i > 0) subpile_add(left_subpile, data_[i - 1], i - 1);
    subpile_update(left_subpile, i - 1 - w);
    if (i < data_size - w) subpile_add(right_subpile, data_[i + w], i + w);
    subpile_update(right_subpile, i);
    std::uint16_t d = clamp16(data_[i] * q);
    if (i != 0 && left_subpile.front().second > d) {
      if (found_down) {
        if (i - last_down > 1) {
          dst.emplace_back(first_down << 1 | 0, last_down);
          first_down = i;
        }
      } else {
        found_down = true;
        first_down = i;
      }
      last_down = i;
    }
    if (i != (data_size - 1) && right_subpile.front().second > d) {
      if (found_up) {
        if (i - last_up > 1) {
          dst.emplace_back(first_up << 1 | 1, last_up);
          first_up = i;
        }
      } else {
        found_up = true;
        first_up = i;
      }
      last_up = i;
    }
  }
  if (found_down) dst.emplace_back(first_down << 1 | 0, last_down);
  if (found_up) dst.emplace_back(first_up << 1 | 1, last_up);
  if (dst.empty()) return dst;
  while (true) {  // separate overlapping slopes
    std::sort(dst.begin(), dst.end());
    bool is_changed = false;
    for (std::uint32_t i = 0; i < dst.size() - 1; ++i) {
      if (dst[i].second < (dst[i + 1].first >> 1)) continue;
      if (dst[i].first & 1) {
        right_subpile.clear();
        found_up = false;
        std::uint32_t subpile_begin = dst[i].first >> 1;
        std::uint32_t subpile_end = std::min(dst[i].second, dst[i + 1].second);
        for (std::uint32_t j = subpile_begin; j < subpile_end + 1; ++j) subpile_add(right_subpile, data_[j], j);
        for (std::uint32_t j = subpile_begin; j < subpile_end; ++j) {
          subpile_update(right_subpile, j);
          if (clamp16(data_[j] * q) < right_subpile.front().second) {
            if (found_up) {
              if (j - last_up > 1) {
                dst.emplace_back(first_up << 1 | 1, last_up);
                first_up = j;
              }
            } else {
              found_up = true;
              first_up = j;
            }
            last_up = j;
          }
        }
        if (found_up) dst.emplace_back(first_up << 1 | 1, last_up);
        dst[i].first = subpile_end << 1 | 1;
      } else {
        if (dst[i].second == (dst[i + 1].first >> 1)) continue;
        left_subpile.clear();
        found_down = false;
        std::uint32_t subpile_begin = std::max(dst[i].first >> 1, dst[i + 1].first >> 1);
        std::uint32_t subpile_end = dst[i].second;
        for (std::uint32_t j = subpile_begin; j < subpile_end + 1; ++j) {
          if (left_subpile.empty() == false && clamp16(data_[j] * q) < left_subpile.front().second) {
            if (found_down) {
              if (j - last_down > 1) {
                dst.emplace_back(first_down << 1, last_down);
                first_down = j;
              }
            } else {
              found_down = true;
              first_down = j;
            }
            last_down = j;
          }
          subpile_add(left_subpile, data_[j], j);
        }
        if (found_down) dst.emplace_back(first_down << 1, last_down);
        dst[i].second = subpile_begin;
      }
      is_changed = true;
      break;
    }
    if (!is_changed) break;
  }
  for (std::uint32_t i = 0; i < dst.size() - 1; ++i) {  // narrow slopes
    if ((dst[i].first & 1) && !(dst[i + 1].first & 1)) {
      std::uint32_t subpile_begin = dst[i].second;
      std::uint32_t subpile_end = dst[i + 1].first >> 1;
      if (subpile_end - subpile_begin > static_cast<std::uint32_t>(w)) continue;
      std::uint16_t max_coverage = 0;
      for (std::uint32_t j = subpile_begin + 1; j < subpile_end; ++j) max_coverage = std::max(max_coverage, data_[j]);
      std::uint32_t valid_point = dst[i].first >> 1;
      for (std::uint32_t j = dst[i].first >> 1; j <= subpile_begin; ++j)
        if (max_coverage > clamp16(data_[j] * q)) valid_point = j;
      dst[i].second = valid_point;
      valid_point = dst[i + 1].second;
      for (std::uint32_t j = subpile_end; j <= dst[i + 1].second; ++j) {
        if (max_coverage > clamp16(data_[j] * q)) {
          valid_point = j;
          break;
        }
      }
      dst[i + 1].first = valid_point << 1 | 0;
    }
  }
  return dst;
}

}  // namespace orc

extern "C" {

// Pile::FindChimericRegions on one coverage array: out = (begin, end) cell pairs; returns their number (<= cap or -1)
std::int64_t orc_find_chimeric_regions(const std::uint16_t* data, std::uint32_t size, std::uint32_t* out, std::uint64_t cap) {
  std::vector<std::uint16_t> d(data, data + size);
  auto slopes = orc::FindSlopes(d, 1.82);
  std::vector<orc::Region> regions;
  if (!slopes.empty()) {
    for (std::uint32_t i = 0; i < slopes.size() - 1; ++i)
      if (!(slopes[i].first & 1) && (slopes[i + 1].first & 1)) regions.emplace_back(slopes[i].first >> 1, slopes[i + 1].second);
    regions = orc::MergeRegions(regions);
  }
  if (regions.size() > cap) return -1;
  for (std::size_t i = 0; i < regions.size(); ++i) {
    out[2 * i] = regions[i].first;
    out[2 * i + 1] = regions[i].second;
  }
  return static_cast<std::int64_t>(regions.size());
}

}  // extern "C"

// ---- overlap bookkeeping against the piles' valid regions (RavenLib/src/overlap_utils.cc, in the reference tree) ----
namespace orc {

struct PileView {  // what OverlapUpdate / GetOverlapType read from a raven::Pile (pile.h:37-47)
  std::uint32_t begin, end;  // Pile::begin() / end(): bases (begin_ << kPSS)
  bool invalid;
};

// overlap_utils.cc:14-80
static bool OverlapUpdate(Overlap& o, const std::vector<PileView>& piles) {
  if (piles[o.lhs_id].invalid || piles[o.rhs_id].invalid) return false;
  if (o.lhs_begin >= piles[o.lhs_id].end || o.lhs_end <= piles[o.lhs_id].begin ||
      o.rhs_begin >= piles[o.rhs_id].end || o.rhs_end <= piles[o.rhs_id].begin) return false;
  std::uint32_t lhs_begin = o.lhs_begin + (o.strand
      ? (o.rhs_begin < piles[o.rhs_id].begin ? piles[o.rhs_id].begin - o.rhs_begin : 0)
      : (o.rhs_end > piles[o.rhs_id].end ? o.rhs_end - piles[o.rhs_id].end : 0));
  std::uint32_t lhs_end = o.lhs_end - (o.strand
      ? (o.rhs_end > piles[o.rhs_id].end ? o.rhs_end - piles[o.rhs_id].end : 0)
      : (o.rhs_begin < piles[o.rhs_id].begin ? piles[o.rhs_id].begin - o.rhs_begin : 0));
  std::uint32_t rhs_begin = o.rhs_begin + (o.strand
      ? (o.lhs_begin < piles[o.lhs_id].begin ? piles[o.lhs_id].begin - o.lhs_begin : 0)
      : (o.lhs_end > piles[o.lhs_id].end ? o.lhs_end - piles[o.lhs_id].end : 0));
  std::uint32_t rhs_end = o.rhs_end - (o.strand
      ? (o.lhs_end > piles[o.lhs_id].end ? o.lhs_end - piles[o.lhs_id].end : 0)
      : (o.lhs_begin < piles[o.lhs_id].begin ? piles[o.lhs_id].begin - o.lhs_begin : 0));
  if (lhs_begin >= piles[o.lhs_id].end || lhs_end <= piles[o.lhs_id].begin ||
      rhs_begin >= piles[o.rhs_id].end || rhs_end <= piles[o.rhs_id].begin) return false;
  lhs_begin = std::max(lhs_begin, piles[o.lhs_id].begin);
  lhs_end = std::min(lhs_end, piles[o.lhs_id].end);
  rhs_begin = std::max(rhs_begin, piles[o.rhs_id].begin);
  rhs_end = std::min(rhs_end, piles[o.rhs_id].end);
  if (lhs_begin >= lhs_end || lhs_end - lhs_begin < 84 || rhs_begin >= rhs_end || rhs_end - rhs_begin < 84) return false;
  o.lhs_begin = lhs_begin;
  o.lhs_end = lhs_end;
  o.rhs_begin = rhs_begin;
  o.rhs_end = rhs_end;
  return true;
}

// overlap_utils.cc:82-113
static std::uint32_t GetOverlapType(const Overlap& o, const std::vector<PileView>& piles) {
  std::uint32_t lhs_length = piles[o.lhs_id].end - piles[o.lhs_id].begin;
  std::uint32_t lhs_begin = o.lhs_begin - piles[o.lhs_id].begin;
  std::uint32_t lhs_end = o.lhs_end - piles[o.lhs_id].begin;
  std::uint32_t rhs_length = piles[o.rhs_id].end - piles[o.rhs_id].begin;
  std::uint32_t rhs_begin = o.strand ? o.rhs_begin - piles[o.rhs_id].begin
                                     : rhs_length - (o.rhs_end - piles[o.rhs_id].begin);
  std::uint32_t rhs_end = o.strand ? o.rhs_end - piles[o.rhs_id].begin
                                   : rhs_length - (o.rhs_begin - piles[o.rhs_id].begin);
  std::uint32_t overhang = std::min(lhs_begin, rhs_begin) + std::min(lhs_length - lhs_end, rhs_length - rhs_end);
  if (lhs_end - lhs_begin < (lhs_end - lhs_begin + overhang) * 0.875 ||
      rhs_end - rhs_begin < (rhs_end - rhs_begin + overhang) * 0.875) return 0;  // internal
  if (lhs_begin <= rhs_begin && lhs_length - lhs_end <= rhs_length - rhs_end) return 1;  // lhs contained
  if (rhs_begin <= lhs_begin && rhs_length - rhs_end <= lhs_length - lhs_end) return 2;  // rhs contained
  if (lhs_begin > rhs_begin) return 3;  // lhs -> rhs
  return 4;                             // rhs -> lhs
}

// the identity score of construct.cc:176-203 / :393-420: spans inflated, rhs reverse-complemented on the opposite
// strand, edlibAlign(default) -> 1 - distance / max(length)
static double IdentityScore(const Overlap& it, const Read& lhs_seq, const Read& rhs_seq) {
  std::string lhs, rhs;
  for (std::uint32_t i = it.lhs_begin; i < it.lhs_end; ++i) lhs += "ACGT"[lhs_seq.Code(i)];
  for (std::uint32_t i = it.rhs_begin; i < it.rhs_end; ++i) rhs += "ACGT"[rhs_seq.Code(i)];
  if (!it.strand) {
    std::string rc(rhs.rbegin(), rhs.rend());
    for (auto& c : rc) c = c == 'A' ? 'T' : (c == 'C' ? 'G' : (c == 'G' ? 'C' : 'A'));
    rhs = rc;
  }
  const std::uint32_t d = EditDistance(lhs.c_str(), lhs.size(), rhs.c_str(), rhs.size());
  return 1. - static_cast<double>(d) / std::max(lhs.size(), rhs.size());
}

}  // namespace orc

extern "C" {

// OverlapUpdate + GetOverlapType on a list (tests of the device classification): ok[i] = OverlapUpdate result, the
// overlap is updated in place when ok, type[i] = GetOverlapType of the updated overlap (undefined when !ok)
void orc_overlap_update_and_type(orc::Overlap* ovl, std::uint64_t n, const std::uint32_t* begin, const std::uint32_t* end,
                                 const std::uint8_t* invalid, std::uint32_t n_piles, std::uint8_t* ok, std::uint32_t* type) {
  std::vector<orc::PileView> piles(n_piles);
  for (std::uint32_t i = 0; i < n_piles; ++i) piles[i] = orc::PileView{begin[i], end[i], invalid[i] != 0};
  for (std::uint64_t i = 0; i < n; ++i) {
    ok[i] = orc::OverlapUpdate(ovl[i], piles) ? 1 : 0;
    type[i] = ok[i] ? orc::GetOverlapType(ovl[i], piles) : 0xFFFFFFFFu;
  }
}

// The identity filter loop of ResolveContainedReads (construct.cc:162-217) on per-pile overlap lists (CSR, in place):
// returns the new total; offsets are rewritten.
std::uint64_t orc_identity_filter(const std::uint64_t* packed, const std::uint64_t* word_offsets, const std::uint32_t* lengths,
                                  std::uint32_t n_reads, orc::Overlap* ovl, std::uint32_t* offsets,
                                  const std::uint32_t* begin, const std::uint32_t* end, const std::uint8_t* invalid,
                                  double identity) {
  auto reads = MakeReads(packed, word_offsets, lengths, nullptr, n_reads);
  std::vector<orc::PileView> piles(n_reads);
  for (std::uint32_t i = 0; i < n_reads; ++i) piles[i] = orc::PileView{begin[i], end[i], invalid[i] != 0};
  std::uint64_t out = 0;
  std::uint32_t prev_end = offsets[0];
  for (std::uint32_t i = 0; i < n_reads; ++i) {
    const std::uint32_t b = prev_end, e = offsets[i + 1];
    prev_end = e;
    offsets[i] = static_cast<std::uint32_t>(out);
    for (std::uint32_t j = b; j < e; ++j) {
      orc::Overlap o = ovl[j];
      if (!orc::OverlapUpdate(o, piles)) continue;
      if (orc::IdentityScore(o, reads[o.lhs_id], reads[o.rhs_id]) < identity) continue;
      ovl[out++] = o;
    }
  }
  offsets[n_reads] = static_cast<std::uint32_t>(out);
  return out;
}

// raven::FindOverlapsAndRepetetiveRegions (construct.cc:316-491) restated: valid reads first (by id), index batches of
// `batch_bases` (reference 1 << 30) of valid reads without minhash, Map(true, true, false, &filtered) of every valid
// read up to the batch end, Pile::AddKmers of the filtered positions, optional identity filter, then the serial
// merge: OverlapUpdate, GetOverlapType, containment flags, consecutive same-pair overlaps keep the longer; finally
// contained piles become invalid and the list is re-checked with OverlapUpdate.
// Outputs: overlaps (the extra slot overlaps.back()), contained[n] (set_is_contained by this pass), kmers: per read
// (len >> 4) + 1 cells at kmers_off[i] (all reads; untouched = 0).  Returns the number of overlaps (<= cap or -1).
std::int64_t orc_second_pass(orc_engine* eng, const std::uint64_t* packed, const std::uint64_t* word_offsets,
                             const std::uint32_t* lengths, std::uint32_t n_reads, const std::uint32_t* begin,
                             const std::uint32_t* end, const std::uint8_t* invalid, double freq, std::uint32_t kmer_len,
                             double identity, std::uint64_t batch_bases, orc::Overlap* out, std::uint64_t cap,
                             std::uint8_t* contained, std::uint8_t* kmers, const std::uint64_t* kmers_off) {
  auto all = MakeReads(packed, word_offsets, lengths, nullptr, n_reads);
  std::vector<orc::PileView> piles(n_reads);
  for (std::uint32_t i = 0; i < n_reads; ++i) piles[i] = orc::PileView{begin[i], end[i], invalid[i] != 0};
  // construct.cc:324-332: valid first, each group by id
  std::vector<orc::Read> sequences;
  for (std::uint32_t i = 0; i < n_reads; ++i) if (!piles[i].invalid) sequences.push_back(all[i]);
  // construct.cc:343-349: s = position of the first invalid pile, 0 when there is none (the reference's loop leaves it
  // at its initial value: with every pile valid nothing is mapped)
  const std::uint32_t s = sequences.size() == n_reads ? 0 : sequences.size();
  std::vector<orc::Overlap> back;
  std::uint64_t bytes = 0;
  for (std::uint32_t i = 0, j = 0; i < s; ++i) {
    bytes += sequences[i].len;
    if (i != s - 1 && bytes < batch_bases) continue;
    bytes = 0;
    eng->e.Minimize(sequences.data() + j, sequences.data() + i + 1, false, 1);
    eng->e.Filter(freq);
    for (std::uint32_t k = 0; k < i + 1; ++k) {
      std::vector<std::uint32_t> filtered;
      auto dst = eng->e.Map(sequences[k], true, true, false, &filtered);
      if (!filtered.empty())
        orc_pile_add_kmers(sequences[k].words, sequences[k].len, filtered.data(), filtered.size(), kmer_len,
                           kmers + kmers_off[sequences[k].id]);
      if (identity != 0) {
        std::uint32_t kk = 0;
        for (std::uint32_t x = 0; x < dst.size(); ++x) {
          if (!orc::OverlapUpdate(dst[x], piles)) continue;
          if (orc::IdentityScore(dst[x], all[dst[x].lhs_id], all[dst[x].rhs_id]) < identity) continue;
          dst[kk++] = dst[x];
        }
        dst.resize(kk);
      }
      for (auto& jt : dst) {  // construct.cc:430-455
        if (!orc::OverlapUpdate(jt, piles)) continue;
        const std::uint32_t type = orc::GetOverlapType(jt, piles);
        if (type == 0) continue;
        if (type == 1) contained[jt.lhs_id] = 1;
        else if (type == 2) contained[jt.rhs_id] = 1;
        else if (!back.empty() && back.back().lhs_id == jt.lhs_id && back.back().rhs_id == jt.rhs_id) {
          if (orc::GetOverlapLength(back.back()) < orc::GetOverlapLength(jt)) back.back() = jt;
        } else {
          back.push_back(jt);
        }
      }
    }
    j = i + 1;
  }
  for (std::uint32_t i = 0; i < n_reads; ++i) if (contained[i]) piles[i].invalid = true;  // construct.cc:466-470
  std::uint64_t k = 0;
  for (std::uint64_t i = 0; i < back.size(); ++i)
    if (orc::OverlapUpdate(back[i], piles)) back[k++] = back[i];
  back.resize(k);
  if (k > cap) return -1;
  for (std::uint64_t i = 0; i < k; ++i) out[i] = back[i];
  return static_cast<std::int64_t>(k);
}

}  // extern "C"

// ---- racon::Polisher::Polish, one round (recollection of racon@library polisher.cpp /
// overlap.cpp / window.cpp; raven call site RavenLib/src/polish.cc:43-51) --------------------------------
// (1) ram(15,5) index of the targets, Filter(0.001), Map(read, false, false); keep the longest overlap per read,
// drop it if 1 - min(span)/max(span) > e; (2) reverse-complement reads on the opposite strand; (3) global NW
// path (edlib EDLIB_TASK_PATH: any optimal unit-cost path; here a plain DP with diagonal-first traceback)
// -> standard CIGAR -> racon's find_breaking_points_from_cigar; (4) windows of w target bases with a dummy '!'
// backbone quality, layers >= 0.02 w and mean quality >= q; (5) Window::GenerateConsensus; (6) stitch + ratio.
namespace orc {

static void NwPathFull(const std::vector<std::uint8_t>& q, const std::vector<std::uint8_t>& t, std::vector<char>* ops) {
  // ops from the start: 'M' (match/mismatch), 'I' (query base only), 'D' (target base only)
  const std::size_t n = q.size(), m = t.size();
  std::vector<std::uint32_t> prev(m + 1), cur(m + 1);
  std::vector<std::uint8_t> dir((n + 1) * (m + 1));
  for (std::size_t j = 0; j <= m; ++j) { prev[j] = j; dir[j] = 2; }
  for (std::size_t i = 1; i <= n; ++i) {
    cur[0] = i;
    dir[i * (m + 1)] = 1;
    for (std::size_t j = 1; j <= m; ++j) {
      std::uint32_t d = prev[j - 1] + (q[i - 1] != t[j - 1]), u = prev[j] + 1, l = cur[j - 1] + 1;
      std::uint32_t best = std::min(d, std::min(u, l));
      cur[j] = best;
      dir[i * (m + 1) + j] = best == d ? 0 : (best == u ? 1 : 2);
    }
    prev.swap(cur);
  }
  ops->clear();
  std::size_t i = n, j = m;
  while (i > 0 || j > 0) {
    std::uint8_t d = dir[i * (m + 1) + j];
    if (i > 0 && j > 0 && d == 0) { ops->push_back('M'); --i; --j; }
    else if (i > 0 && (d == 1 || j == 0)) { ops->push_back('I'); --i; }
    else { ops->push_back('D'); --j; }
  }
  std::reverse(ops->begin(), ops->end());
}

// The same DP restricted to the diagonals an alignment of cost <= k can visit (Ukkonen): j - i in
// [min(0, m - n) - k, max(0, m - n) + k]; k doubles until the distance found is <= k.  The result is the full matrix's,
// traceback included: every cell the traceback visits lies on an optimal path, an optimal path of cost d <= k never
// leaves the band, so its cells (and every predecessor that TIES for the minimum, which is on an optimal path too) carry
// exact values; cells whose best prefix would leave the band are only too large and never win.  orc_nw_full(1) switches
// back to the full matrix (tests/test_oracle.py compares the two).
static bool g_nw_full = false;
static bool NwPathBand(const std::vector<std::uint8_t>& q, const std::vector<std::uint8_t>& t, std::uint64_t k, std::vector<char>* ops) {
  const std::int64_t n = q.size(), m = t.size();
  const std::int64_t lo = std::min<std::int64_t>(0, m - n) - static_cast<std::int64_t>(k);
  const std::int64_t hi = std::max<std::int64_t>(0, m - n) + static_cast<std::int64_t>(k);
  const std::int64_t W = hi - lo + 1;
  const std::uint32_t kInf = 0x3FFFFFFFu;
  // row i holds columns j = i + lo .. i + hi at index j - i - lo; so (i-1, j-1) is the same index of the row above,
  // (i-1, j) index + 1 of the row above, (i, j-1) index - 1 of this row
  std::vector<std::uint32_t> prev(W + 2, kInf), cur(W + 2, kInf);
  std::vector<std::uint8_t> dir(static_cast<std::size_t>(n + 1) * W);
  for (std::int64_t x = 0; x < W; ++x) {
    const std::int64_t j = x + lo;
    if (j >= 0 && j <= m) { prev[x + 1] = static_cast<std::uint32_t>(j); dir[x] = 2; }
  }
  for (std::int64_t i = 1; i <= n; ++i) {
    std::fill(cur.begin(), cur.end(), kInf);
    for (std::int64_t x = 0; x < W; ++x) {
      const std::int64_t j = i + lo + x;
      if (j < 0 || j > m) continue;
      if (j == 0) { cur[x + 1] = static_cast<std::uint32_t>(i); dir[i * W + x] = 1; continue; }
      const std::uint32_t d = prev[x + 1] + (q[i - 1] != t[j - 1]), u = prev[x + 2] + 1, l = cur[x] + 1;
      const std::uint32_t best = std::min(d, std::min(u, l));
      cur[x + 1] = best;
      dir[i * W + x] = best == d ? 0 : (best == u ? 1 : 2);
    }
    prev.swap(cur);
  }
  const std::int64_t xe = m - n - lo;
  if (prev[xe + 1] > k) return false;
  ops->clear();
  std::int64_t i = n, j = m;
  while (i > 0 || j > 0) {
    const std::uint8_t d = dir[i * W + (j - i - lo)];
    if (i > 0 && j > 0 && d == 0) { ops->push_back('M'); --i; --j; }
    else if (i > 0 && (d == 1 || j == 0)) { ops->push_back('I'); --i; }
    else { ops->push_back('D'); --j; }
  }
  std::reverse(ops->begin(), ops->end());
  return true;
}
static void NwPath(const std::vector<std::uint8_t>& q, const std::vector<std::uint8_t>& t, std::vector<char>* ops) {
  if (g_nw_full || q.empty() || t.empty()) return NwPathFull(q, t, ops);
  const std::uint64_t longest = std::max(q.size(), t.size());
  for (std::uint64_t k = std::max<std::uint64_t>(32, longest / 8);; k *= 2) {
    if (k >= longest) return NwPathFull(q, t, ops);
    if (NwPathBand(q, t, k, ops)) return;
  }
}

// racon Overlap::find_breaking_points_from_cigar: for every window of w target bases the first aligned pair
// (t, q) and one past the last aligned pair, in window order; windows without an aligned pair contribute nothing
static void BreakingPoints(const std::vector<char>& ops, std::uint32_t q_begin, std::uint32_t t_begin,
                           std::uint32_t t_end, std::uint32_t w,
                           std::vector<std::pair<std::uint32_t, std::uint32_t>>* out) {
  std::vector<std::int64_t> window_ends;
  for (std::uint32_t i = 0; i < t_end; i += w)
    if (i > t_begin) window_ends.push_back(static_cast<std::int64_t>(i) - 1);
  window_ends.push_back(static_cast<std::int64_t>(t_end) - 1);
  auto& bp = *out;
  bp.clear();
  std::size_t wi = 0;
  bool found_first = false;
  std::pair<std::uint32_t, std::uint32_t> first_match{0, 0}, last_match{0, 0};
  std::int64_t q_ptr = static_cast<std::int64_t>(q_begin) - 1, t_ptr = static_cast<std::int64_t>(t_begin) - 1;
  for (char op : ops) {
    if (op == 'M') {
      ++q_ptr; ++t_ptr;
      if (!found_first) { found_first = true; first_match = {static_cast<std::uint32_t>(t_ptr), static_cast<std::uint32_t>(q_ptr)}; }
      last_match = {static_cast<std::uint32_t>(t_ptr + 1), static_cast<std::uint32_t>(q_ptr + 1)};
      if (wi < window_ends.size() && t_ptr == window_ends[wi]) {
        if (found_first) { bp.push_back(first_match); bp.push_back(last_match); }
        found_first = false;
        ++wi;
      }
    } else if (op == 'I') {
      ++q_ptr;
    } else {
      ++t_ptr;
      if (wi < window_ends.size() && t_ptr == window_ends[wi]) {
        if (found_first) { bp.push_back(first_match); bp.push_back(last_match); }
        found_first = false;
        ++wi;
      }
    }
  }
}

}  // namespace orc

// query / target: one-byte codes of the two spans; out: (t, q) pairs, two per window that has an aligned pair
extern "C" void orc_nw_full(int on) { orc::g_nw_full = on != 0; }

extern "C" std::uint64_t orc_nw_breakpoints(const std::uint8_t* query, std::uint32_t n, const std::uint8_t* target,
                                            std::uint32_t m, std::uint32_t q_begin, std::uint32_t t_begin,
                                            std::uint32_t w, std::uint32_t* out, std::uint64_t cap,
                                            std::uint32_t* distance) {
  std::vector<std::uint8_t> q(query, query + n), t(target, target + m);
  std::vector<char> ops;
  orc::NwPath(q, t, &ops);
  if (distance) {
    std::uint32_t d = 0;
    std::size_t qi = 0, ti = 0;
    for (char op : ops) {
      if (op == 'M') { d += q[qi] != t[ti]; ++qi; ++ti; }
      else if (op == 'I') { ++d; ++qi; }
      else { ++d; ++ti; }
    }
    *distance = d;
  }
  std::vector<std::pair<std::uint32_t, std::uint32_t>> bp;
  orc::BreakingPoints(ops, q_begin, t_begin, t_begin + m, w, &bp);
  for (std::size_t i = 0; i < bp.size() && 2 * i + 1 < cap; ++i) {
    out[2 * i] = bp[i].first;
    out[2 * i + 1] = bp[i].second;
  }
  return bp.size();
}

// one layer of a window as racon would add it: {window, read index, first base in the oriented read, bases, begin,
// end, reverse-complemented}
using LayerDump = std::vector<std::array<std::uint32_t, 7>>;

static int PolishRound(const std::uint64_t* t_packed, const std::uint64_t* t_word_off, const std::uint32_t* t_len,
                     const std::uint32_t* t_ids, std::uint32_t n_targets, const std::uint64_t* r_packed,
                     const std::uint64_t* r_word_off, const std::uint32_t* r_len, const std::uint32_t* r_ids,
                     std::uint32_t n_reads, const std::uint8_t* quals, const std::uint64_t* qual_off, double q_thr,
                     double err_thr, std::uint32_t w, int trim, int m, int n, int g, std::uint8_t* out,
                     const std::uint64_t* out_off, std::uint32_t* out_len, double* ratio, LayerDump* dump) {
  auto targets = MakeReads(t_packed, t_word_off, t_len, t_ids, n_targets);
  auto reads = MakeReads(r_packed, r_word_off, r_len, r_ids, n_reads);
  orc::MinimizerEngine engine(15, 5, 500, 4, 100, 10000);
  engine.Minimize(targets.data(), targets.data() + n_targets, false, 1);
  engine.Filter(0.001);
  std::vector<std::uint32_t> id_to_t;
  for (std::uint32_t t = 0; t < n_targets; ++t) {
    if (targets[t].id >= id_to_t.size()) id_to_t.resize(targets[t].id + 1, 0xFFFFFFFFu);
    id_to_t[targets[t].id] = t;
  }
  std::vector<std::uint64_t> first_window(n_targets + 1, 0);
  for (std::uint32_t t = 0; t < n_targets; ++t) first_window[t + 1] = first_window[t] + (targets[t].len + w - 1) / w;
  struct LayerData { std::vector<std::uint8_t> codes, qual; std::uint32_t begin, end; };
  std::vector<std::vector<LayerData>> win_layers(first_window[n_targets]);

  for (std::uint32_t r = 0; r < n_reads; ++r) {
    auto ovl = engine.Map(reads[r], false, false, false, nullptr);
    if (ovl.empty()) continue;
    auto length = [](const orc::Overlap& o) { return std::max(o.lhs_end - o.lhs_begin, o.rhs_end - o.rhs_begin); };
    orc::Overlap best = ovl.front();
    for (const auto& o : ovl) if (length(best) < length(o)) best = o;
    double a = best.lhs_end - best.lhs_begin, b = best.rhs_end - best.rhs_begin;
    if (1.0 - std::min(a, b) / std::max(a, b) > err_thr) continue;
    if (best.rhs_id >= id_to_t.size() || id_to_t[best.rhs_id] == 0xFFFFFFFFu) continue;
    const std::uint32_t t = id_to_t[best.rhs_id];
    const std::uint32_t qlen = reads[r].len;
    // read in the target's orientation, with qualities
    std::vector<std::uint8_t> rq(qlen), rqual;
    const bool rc = !best.strand;
    for (std::uint32_t i = 0; i < qlen; ++i) rq[i] = rc ? 3 - reads[r].Code(qlen - 1 - i) : reads[r].Code(i);
    if (quals) {
      rqual.resize(qlen);
      for (std::uint32_t i = 0; i < qlen; ++i) rqual[i] = quals[qual_off[r] + (rc ? qlen - 1 - i : i)];
    }
    const std::uint32_t q_begin = rc ? qlen - best.lhs_end : best.lhs_begin;
    const std::uint32_t q_end = rc ? qlen - best.lhs_begin : best.lhs_end;
    std::vector<std::uint8_t> qs(rq.begin() + q_begin, rq.begin() + q_end), ts(best.rhs_end - best.rhs_begin);
    for (std::uint32_t i = 0; i < ts.size(); ++i) ts[i] = targets[t].Code(best.rhs_begin + i);
    std::vector<char> ops;
    orc::NwPath(qs, ts, &ops);
    std::vector<std::pair<std::uint32_t, std::uint32_t>> bp;
    orc::BreakingPoints(ops, q_begin, best.rhs_begin, best.rhs_end, w, &bp);
    for (std::size_t j = 0; j + 1 < bp.size(); j += 2) {
      if (bp[j + 1].second - bp[j].second < 0.02 * w) continue;
      if (quals) {
        double sum = 0;
        for (std::uint32_t x = bp[j].second; x < bp[j + 1].second; ++x) sum += static_cast<double>(rqual[x]) - 33.0;
        if (sum / (bp[j + 1].second - bp[j].second) < q_thr) continue;
      }
      const std::uint64_t window_id = first_window[t] + bp[j].first / w;
      const std::uint32_t window_start = (bp[j].first / w) * w;
      LayerData L;
      L.codes.assign(rq.begin() + bp[j].second, rq.begin() + bp[j + 1].second);
      if (quals) L.qual.assign(rqual.begin() + bp[j].second, rqual.begin() + bp[j + 1].second);
      L.begin = bp[j].first - window_start;
      L.end = bp[j + 1].first - window_start - 1;
      if (L.begin >= L.end) continue;  // racon's AddLayer would reject it
      if (dump) {
        const std::uint32_t bl = std::min<std::uint32_t>(w, targets[t].len - window_start);
        dump->push_back({static_cast<std::uint32_t>(window_id), r, bp[j].second, bp[j + 1].second - bp[j].second, L.begin,
                         std::min(L.end, bl - 1), rc ? 1u : 0u});
      }
      win_layers[window_id].push_back(std::move(L));
    }
  }
  if (dump) {  // racon: layers of a window in the stable order of their begin position
    std::stable_sort(dump->begin(), dump->end(), [](const std::array<std::uint32_t, 7>& a, const std::array<std::uint32_t, 7>& b) {
      return a[0] < b[0] || (a[0] == b[0] && a[4] < b[4]);
    });
    return 0;
  }
  for (std::uint32_t t = 0; t < n_targets; ++t) {
    std::vector<std::uint8_t> polished;
    std::uint64_t nw = first_window[t + 1] - first_window[t], n_pol = 0;
    for (std::uint64_t k = 0; k < nw; ++k) {
      const std::uint32_t ws = static_cast<std::uint32_t>(k) * w;
      const std::uint32_t bl = std::min<std::uint32_t>(w, targets[t].len - ws);
      std::vector<std::uint8_t> bb(bl), bbq(bl, '!');
      for (std::uint32_t x = 0; x < bl; ++x) bb[x] = targets[t].Code(ws + x);
      std::vector<poa::Layer> layers;
      layers.push_back(poa::Layer{bb.data(), bbq.data(), bl, 0, bl ? bl - 1 : 0});
      for (const auto& L : win_layers[first_window[t] + k])
        layers.push_back(poa::Layer{L.codes.data(), L.qual.empty() ? nullptr : L.qual.data(),
                                    static_cast<std::uint32_t>(L.codes.size()), L.begin, std::min(L.end, bl - 1)});
      std::vector<std::uint8_t> cons;
      n_pol += poa::WindowConsensus(layers, m, n, g, trim != 0, &cons, nullptr) ? 1 : 0;
      polished.insert(polished.end(), cons.begin(), cons.end());
    }
    if (polished.size() > out_off[t + 1] - out_off[t]) return -1;
    std::memcpy(out + out_off[t], polished.data(), polished.size());
    out_len[t] = polished.size();
    ratio[t] = nw ? static_cast<double>(n_pol) / nw : 0.0;
  }
  return 0;
}

extern "C" int orc_polish_round(const std::uint64_t* t_packed, const std::uint64_t* t_word_off, const std::uint32_t* t_len,
                     const std::uint32_t* t_ids, std::uint32_t n_targets, const std::uint64_t* r_packed,
                     const std::uint64_t* r_word_off, const std::uint32_t* r_len, const std::uint32_t* r_ids,
                     std::uint32_t n_reads, const std::uint8_t* quals, const std::uint64_t* qual_off, double q_thr,
                     double err_thr, std::uint32_t w, int trim, int m, int n, int g, std::uint8_t* out,
                     const std::uint64_t* out_off, std::uint32_t* out_len, double* ratio) {
  return PolishRound(t_packed, t_word_off, t_len, t_ids, n_targets, r_packed, r_word_off, r_len, r_ids, n_reads, quals,
                     qual_off, q_thr, err_thr, w, trim, m, n, g, out, out_off, out_len, ratio, nullptr);
}

// The layers racon would hand to its windows (steps 1-4 of the round, no consensus): 7 uint32 per layer, windows in
// order, layers of a window in the stable order of their begin position.  Returns the number of layers.
extern "C" std::uint64_t orc_polish_layers(const std::uint64_t* t_packed, const std::uint64_t* t_word_off,
                                           const std::uint32_t* t_len, const std::uint32_t* t_ids, std::uint32_t n_targets,
                                           const std::uint64_t* r_packed, const std::uint64_t* r_word_off,
                                           const std::uint32_t* r_len, const std::uint32_t* r_ids, std::uint32_t n_reads,
                                           const std::uint8_t* quals, const std::uint64_t* qual_off, double q_thr,
                                           double err_thr, std::uint32_t w, std::uint32_t* out, std::uint64_t cap) {
  LayerDump dump;
  PolishRound(t_packed, t_word_off, t_len, t_ids, n_targets, r_packed, r_word_off, r_len, r_ids, n_reads, quals, qual_off,
              q_thr, err_thr, w, 0, 3, -5, -4, nullptr, nullptr, nullptr, nullptr, &dump);
  for (std::uint64_t i = 0; i < dump.size() && i < cap; ++i)
    for (int x = 0; x < 7; ++x) out[7 * i + x] = dump[i][x];
  return dump.size();
}

extern "C" {

// McIlroy's "killer adversary for quicksort" run against std::sort itself: produces values on which
// libstdc++'s introsort exhausts its depth limit and falls back to heapsort (used to test that the device
// restatement of std::sort follows the same path).
void orc_antiqsort(std::uint32_t* out_vals, std::uint32_t n) {
  const std::uint32_t gas = n;
  std::vector<std::uint32_t> val(n, gas), ptr(n);
  for (std::uint32_t i = 0; i < n; ++i) ptr[i] = i;
  std::uint32_t nsolid = 0, candidate = 0;
  std::sort(ptr.begin(), ptr.end(), [&](std::uint32_t x, std::uint32_t y) {
    if (val[x] == gas && val[y] == gas) {
      if (x == candidate) val[x] = nsolid++;
      else val[y] = nsolid++;
    }
    if (val[x] == gas) candidate = x;
    else if (val[y] == gas) candidate = y;
    return val[x] < val[y];
  });
  for (std::uint32_t i = 0; i < n; ++i) out_vals[i] = val[i];
}

std::uint32_t orc_edit_distance(const char* a, std::uint32_t n, const char* b, std::uint32_t m) {
  return orc::EditDistance(a, n, b, m);
}

}  // extern "C"
