// pass2.hip — the second mapping pass and the identity filters of the overlap phase on the device:
//   raven::FindOverlapsAndRepetetiveRegions   RavenLib/src/construct.cc:316-491  (second_pass)
//   identity filter of ResolveContainedReads   RavenLib/src/construct.cc:162-217  (identity_filter_lists)
// with the per-overlap rules of RavenLib/src/overlap_utils.cc (overlap_rules.h) and Pile::AddKmers (pile.cc:64-120).
//
// Second pass, per index batch of 2^30 bases of VALID reads (valid first, by id — construct.cc:324-359):
//   Minimize(batch) without minhash, Filter(freq), Map(read, true, true, false, &filtered) of every valid read up to the
//   batch end (one device pass), AddKmers of the filtered positions into the k-mer cells kept in HBM, [identity != 0:
//   OverlapUpdate -> batched exact edit distance of the two spans -> drop below the threshold], then the merge of
//   construct.cc:430-455 in Map-output order: OverlapUpdate, GetOverlapType, containment flags, survivors (type 3 / 4)
//   appended to the result list.  After the last batch: consecutive overlaps of the same read pair keep the longer one
//   (the first of equal length), contained piles become invalid, the list is re-checked with OverlapUpdate.
// Every step is a flat kernel over all overlaps of a batch; order-dependent steps (the de-duplication of consecutive
// pairs) work on the compacted list, whose order is the reference's serial order.
#include <algorithm>
#include <cstdio>
#include <vector>

#include "engine.h"
#include "kmer.h"
#include "lowcomplexity.h"
#include "overlap_rules.h"
#include "wave.h"

namespace rvn {

namespace {

constexpr u32 kPSS2 = 4;

// one thread per output word: words of read i of the subset come from read src[i] of the full set
__global__ void gather_reads_kernel(const u64* __restrict__ in_packed, const u64* __restrict__ in_word_off,
                                    const u32* __restrict__ src, const u64* __restrict__ out_word_off, u32 n, u64 n_words,
                                    u64* __restrict__ out_packed) {
  const u64 wi = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (wi >= n_words) return;
  u32 lo = 0, hi = n;
  while (hi - lo > 1) {
    const u32 mid = lo + (hi - lo) / 2;
    if (out_word_off[mid] <= wi) lo = mid;
    else hi = mid;
  }
  out_packed[wi] = in_packed[in_word_off[src[lo]] + (wi - out_word_off[lo])];
}

// Pile::AddKmers on the `filtered` flags of a Map batch: one thread per query minimizer
__global__ void add_kmers_flags_kernel(const u64* __restrict__ packed, const u64* __restrict__ word_off,
                                       const u8* __restrict__ filtered, const u64* __restrict__ org,
                                       const u32* __restrict__ read_off, u32 n_reads, u64 n_query, u32 k,
                                       const u64* __restrict__ kmers_off, u8* __restrict__ kmers) {
  const u64 q = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= n_query || !filtered[q]) return;
  u32 lo = 0, hi = n_reads;  // read of this minimizer: last r with read_off[r] <= q
  while (hi - lo > 1) {
    const u32 mid = lo + (hi - lo) / 2;
    if (read_off[mid] <= q) lo = mid;
    else hi = mid;
  }
  const u32 p = static_cast<u32>(org[q]) >> 1;
  const u64* w = packed + word_off[lo];
  const u64 mask = k >= 32 ? ~0ULL : ((1ULL << (2 * k)) - 1);
  const u32 bit = 2 * p;
  const u64 x = extract_bits(w[bit >> 6], w[(bit >> 6) + 1], bit & 63, mask);
  u8 codes[32];
  for (u32 i = 0; i < k; ++i) codes[i] = static_cast<u8>((x >> (2 * i)) & 3);
  if (lc_kmer_passes(codes, k)) kmers[kmers_off[lo] + (p >> kPSS2)] = 1;
}

// OverlapUpdate of every overlap, in place: updated coordinates + ok flag
__global__ void update_kernel(Overlap* __restrict__ ovl, u64 n, const PileRegion* __restrict__ regions,
                              u8* __restrict__ ok) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Overlap o = ovl[i];
  const bool good = overlap_update(o, regions[o.lhs_id], regions[o.rhs_id]);
  ovl[i] = o;
  ok[i] = good ? 1 : 0;
}

struct EdPairRec {
  u32 a_idx, a_begin, a_len, b_idx, b_begin, b_len, strand, pad;
};

// edlibAlign(lhs span, rhs span [reverse-complemented on the opposite strand]) pairs of the overlaps that survived the update
// the reference's keep rule on a distance x: !(1. - x / max(length) < identity), in double
__device__ __forceinline__ bool identity_keeps(u32 x, u32 maxlen, double identity) {
  return !(1. - static_cast<double>(x) / static_cast<double>(maxlen) < identity);
}

// Also the largest distance that still passes the filter (kmax): the edit-distance stage then needs ONE sweep at that
// threshold per pair — "exact distance if <= kmax, else above" decides the overlap exactly as the unbounded distance would.
__global__ void ed_pairs_kernel(const Overlap* __restrict__ ovl, const u8* __restrict__ ok, const u32* __restrict__ slot,
                                u64 n, const u32* __restrict__ index_of, double identity, EdPairRec* __restrict__ pairs,
                                u32* __restrict__ kmax) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n || !ok[i]) return;
  const Overlap o = ovl[i];
  const u32 a = o.lhs_end - o.lhs_begin, b = o.rhs_end - o.rhs_begin;
  pairs[slot[i]] = EdPairRec{index_of[o.lhs_id], o.lhs_begin, a, index_of[o.rhs_id], o.rhs_begin, b, o.strand ? 1u : 0u, 0u};
  const u32 maxlen = a > b ? a : b;
  double guess = (1. - identity) * static_cast<double>(maxlen);
  guess = guess < 0 ? 0 : (guess > static_cast<double>(maxlen) ? static_cast<double>(maxlen) : guess);
  u32 t = static_cast<u32>(guess);
  while (t < maxlen && identity_keeps(t + 1, maxlen, identity)) ++t;  // the rule is monotone in the distance
  while (t > 0 && !identity_keeps(t, maxlen, identity)) --t;
  kmax[slot[i]] = t;
}

// score = 1 - distance / max(length) in double; overlaps below the identity threshold lose their ok flag
__global__ void identity_keep_kernel(const Overlap* __restrict__ ovl, u8* __restrict__ ok, const u32* __restrict__ slot,
                                     u64 n, const u32* __restrict__ dist, double identity) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n || !ok[i]) return;
  const Overlap o = ovl[i];
  const u32 a = o.lhs_end - o.lhs_begin, b = o.rhs_end - o.rhs_begin;
  // dist is exact up to the pair's kmax and 0xFFFFFFFE above it: the rule gives the same answer either way
  const double score = 1. - static_cast<double>(dist[slot[i]]) / static_cast<double>(a > b ? a : b);
  if (score < identity) ok[i] = 0;
}

// the merge of construct.cc:430-455 without its order-dependent part: OverlapUpdate (a no-op on an already updated
// overlap), type, containment flags; keep[i] = goes to the result list (type 3 / 4)
__global__ void classify_kernel(Overlap* __restrict__ ovl, const u8* __restrict__ ok, u64 n,
                                const PileRegion* __restrict__ regions, u8* __restrict__ contained,
                                u8* __restrict__ keep) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u8 k = 0;
  if (ok[i]) {
    Overlap o = ovl[i];
    const PileRegion L = regions[o.lhs_id], R = regions[o.rhs_id];
    if (overlap_update(o, L, R)) {
      const u32 type = overlap_type(o, L, R);
      if (type == 1) contained[o.lhs_id] = 1;
      else if (type == 2) contained[o.rhs_id] = 1;
      else if (type >= 3) {
        ovl[i] = o;
        k = 1;
      }
    }
  }
  keep[i] = k;
}

__global__ void compact_kernel(const Overlap* __restrict__ in, const u8* __restrict__ keep, const u32* __restrict__ slot,
                               u64 n, Overlap* __restrict__ out) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n && keep[i]) out[slot[i]] = in[i];
}

// consecutive overlaps of the same (lhs, rhs) pair: the survivor is the first one of maximal length
// (construct.cc:444-453: replaced only when strictly longer).  One thread per run head.
__global__ void dedup_kernel(const Overlap* __restrict__ ovl, u64 n, u8* __restrict__ keep) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Overlap o = ovl[i];
  if (i > 0 && ovl[i - 1].lhs_id == o.lhs_id && ovl[i - 1].rhs_id == o.rhs_id) return;  // not a run head
  u64 best = i;
  u32 best_len = overlap_length(o);
  keep[i] = 0;
  for (u64 j = i + 1; j < n && ovl[j].lhs_id == o.lhs_id && ovl[j].rhs_id == o.rhs_id; ++j) {
    keep[j] = 0;
    const u32 l = overlap_length(ovl[j]);
    if (best_len < l) {
      best_len = l;
      best = j;
    }
  }
  keep[best] = 1;
}

__global__ void merge_invalid_kernel(PileRegion* __restrict__ regions, const u8* __restrict__ contained, u32 n) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && contained[i]) regions[i].invalid = 1;
}

// keep flags -> compacted list (in place through a temporary); returns the new count
u64 compact(Engine& e, DevBuf& list, u64 n, const u8* d_keep, DevBuf& tmp_slot, DevBuf& tmp_out) {
  if (n == 0) return 0;
  hipStream_t s = e.stream;
  u32* d_slot = tmp_slot.get<u32>(n + 2);
  exclusive_scan_u8_u32(d_keep, d_slot, n, e.scan_tmp, s);
  const u64 m = read_back(e, d_slot + n, 4);
  Overlap* d_out = tmp_out.get<Overlap>(m + 1);
  compact_kernel<<<div_up(n, 256), 256, 0, s>>>(list.as<Overlap>(), d_keep, d_slot, n, d_out);
  RVN_LAUNCH_CHECK();
  RVN_HIP(rvn_stream_sync(s));
  std::swap(list.ptr, tmp_out.ptr);
  std::swap(list.cap, tmp_out.cap);
  return m;
}

// OverlapUpdate + [identity] on a device list; ok flags out.  regions indexed by read id, index_of: id -> index in r.
void update_and_identity(Engine& e, const ReadsDev& r, Overlap* d_ovl, u64 n, const PileRegion* d_regions,
                         const u32* d_index_of, double identity, u8* d_ok) {
  if (n == 0) return;
  hipStream_t s = e.stream;
  update_kernel<<<div_up(n, 256), 256, 0, s>>>(d_ovl, n, d_regions, d_ok);
  RVN_LAUNCH_CHECK();
  if (identity == 0) return;
  u32* d_slot = e.p2_slot.get<u32>(n + 2);
  exclusive_scan_u8_u32(d_ok, d_slot, n, e.scan_tmp, s);
  const u64 np = read_back(e, d_slot + n, 4);
  if (np == 0) return;
  if (np >= 0xFFFFFFFFULL) throw std::invalid_argument("[raven_hip] identity filter: too many pairs in one batch");
  EdPairRec* d_pairs = e.p2_pairs.get<EdPairRec>(np + 1);
  u32* d_dist = e.p2_dist.get<u32>(2 * (np + 1));
  u32* d_kmax = d_dist + np + 1;
  ed_pairs_kernel<<<div_up(n, 256), 256, 0, s>>>(d_ovl, d_ok, d_slot, n, d_index_of, identity, d_pairs, d_kmax);
  RVN_LAUNCH_CHECK();
  edit_distance_dev(e, r, reinterpret_cast<const u32*>(d_pairs), static_cast<u32>(np), d_dist, d_kmax);
  identity_keep_kernel<<<div_up(n, 256), 256, 0, s>>>(d_ovl, d_ok, d_slot, n, d_dist, identity);
  RVN_LAUNCH_CHECK();
}

PileRegion* upload_regions(Engine& e, const u32* h_begin, const u32* h_end, const u8* h_invalid, u32 n) {
  std::vector<PileRegion> reg(n);
  for (u32 i = 0; i < n; ++i) reg[i] = PileRegion{h_begin[i], h_end[i], h_invalid[i] ? 1u : 0u};
  PileRegion* d = e.p2_regions.get<PileRegion>(static_cast<size_t>(n) + 1);
  RVN_HIP(hipMemcpyAsync(d, reg.data(), static_cast<size_t>(n) * sizeof(PileRegion), hipMemcpyHostToDevice, e.stream));
  RVN_HIP(rvn_stream_sync(e.stream));
  return d;
}

}  // namespace

// Device copy of the reads `src` (indices into R, increasing) as a read set of its own: ids = the original indices
void reads_subset(Engine& e, const ReadsDev& R, const std::vector<u32>& src, ReadsDev& V) {
  hipStream_t s = e.stream;
  const u32 n = static_cast<u32>(src.size());
  V.n = n;
  V.h_word_off.assign(static_cast<size_t>(n) + 1, 0);
  V.h_len.resize(n);
  V.h_id.resize(n);
  V.total_bases = 0;
  for (u32 i = 0; i < n; ++i) {
    V.h_len[i] = R.h_len[src[i]];
    V.h_id[i] = R.h_id[src[i]];
    V.total_bases += V.h_len[i];
    V.h_word_off[i + 1] = V.h_word_off[i] + (R.h_word_off[src[i] + 1] - R.h_word_off[src[i]]);
  }
  V.ids_are_indices = false;
  V.n_words = V.h_word_off[n];
  u64* d_packed = V.packed.get<u64>(V.n_words + 2);
  u64* d_wo = V.word_off.get<u64>(static_cast<size_t>(n) + 1);
  u32* d_len = V.len.get<u32>(static_cast<size_t>(n) + 1);
  u32* d_id = V.id.get<u32>(static_cast<size_t>(n) + 1);
  u32* d_src = e.p2_slot.get<u32>(static_cast<size_t>(n) + 2);
  RVN_HIP(hipMemcpyAsync(d_wo, V.h_word_off.data(), (static_cast<size_t>(n) + 1) * 8, hipMemcpyHostToDevice, s));
  if (n) {
    RVN_HIP(hipMemcpyAsync(d_len, V.h_len.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, s));
    RVN_HIP(hipMemcpyAsync(d_id, V.h_id.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, s));
    RVN_HIP(hipMemcpyAsync(d_src, src.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, s));
    if (V.n_words) {
      gather_reads_kernel<<<div_up(V.n_words, 256), 256, 0, s>>>(R.packed.as<u64>(), R.word_off.as<u64>(), d_src, d_wo, n,
                                                                 V.n_words, d_packed);
      RVN_LAUNCH_CHECK();
    }
  }
  RVN_HIP(hipMemsetAsync(d_packed + V.n_words, 0, 16, s));
  RVN_HIP(rvn_stream_sync(s));
  reads_build_tiles(e, V);
}

// raven::FindOverlapsAndRepetetiveRegions (construct.cc:316-491); R = all reads with ids[i] == i
void second_pass(Engine& e, const ReadsDev& R, const u32* h_begin, const u32* h_end, const u8* h_invalid, double freq,
                 u32 kmer_len, double identity, u64 batch_bases, Pass2State& out) {
  hipStream_t s = e.stream;
  const u32 n = R.n;
  out.n = n;
  out.n_overlaps = 0;
  PileRegion* d_regions = upload_regions(e, h_begin, h_end, h_invalid, n);
  u8* d_contained = out.contained.get<u8>(static_cast<size_t>(n) + 16);
  RVN_HIP(hipMemsetAsync(d_contained, 0, static_cast<size_t>(n) + 16, s));
  // valid reads first, by id (construct.cc:324-349); index_of: id -> position among the valid reads
  std::vector<u32> valid;
  std::vector<u32> index_of(n, 0xFFFFFFFFu);
  out.h_kmers_off.assign(static_cast<size_t>(n) + 1, 0);
  for (u32 i = 0; i < n; ++i) {
    if (!h_invalid[i]) {
      index_of[i] = static_cast<u32>(valid.size());
      valid.push_back(i);
      out.h_kmers_off[i + 1] = out.h_kmers_off[i] + (static_cast<u64>(R.h_len[i]) >> kPSS2) + 1;
    } else {
      out.h_kmers_off[i + 1] = out.h_kmers_off[i];
    }
  }
  out.kmers_total = out.h_kmers_off[n];
  u8* d_kmers = out.kmers.get<u8>(out.kmers_total + 16);
  RVN_HIP(hipMemsetAsync(d_kmers, 0, out.kmers_total + 16, s));
  // construct.cc:343-349: `s` is the position of the FIRST INVALID pile after the valid-first sort and stays 0 when no
  // pile is invalid at all — the reference then maps nothing and overlaps.back() stays empty.  Reproduced on purpose
  // (results identical to the reference's on the same input); real read sets always have contained, hence invalid, piles.
  const u32 sv = valid.size() == n ? 0u : static_cast<u32>(valid.size());
  if (sv == 0) {
    if (valid.size() == n && n > 0)  // (say so: an empty second pass otherwise looks like "no overlaps found")
      std::fprintf(stderr, "[raven_hip] second pass: every pile is valid, so nothing is mapped — the reference's behaviour "
                           "(construct.cc:343-349), reproduced on purpose\n");
    RVN_HIP(rvn_stream_sync(s));
    return;
  }
  ReadsDev V;
  reads_subset(e, R, valid, V);
  u32* d_index_of = e.p2_index_of.get<u32>(static_cast<size_t>(n) + 1);
  RVN_HIP(hipMemcpyAsync(d_index_of, index_of.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, s));
  std::vector<u64> v_kmers_off(static_cast<size_t>(sv) + 1);
  for (u32 i = 0; i < sv; ++i) v_kmers_off[i] = out.h_kmers_off[valid[i]];
  v_kmers_off[sv] = out.kmers_total;
  u64* d_v_kmers_off = e.p2_kmers_off.get<u64>(static_cast<size_t>(sv) + 1);
  RVN_HIP(hipMemcpyAsync(d_v_kmers_off, v_kmers_off.data(), v_kmers_off.size() * 8, hipMemcpyHostToDevice, s));
  RVN_HIP(rvn_stream_sync(s));

  u64 acc_n = 0;  // survivors of all batches so far, in the reference's merge order (out.ovl)
  u64 bytes = 0;
  for (u32 i = 0, j = 0; i < sv; ++i) {
    bytes += V.h_len[i];
    if (i != sv - 1 && bytes < batch_bases) continue;
    bytes = 0;
    engine_minimize(e, V, j, i + 1, false);
    index_filter(e, freq);
    MapOut& mo = e.map_out;
    map_batch(e, V, 0, i + 1, true, true, false, true, mo);
    e.c_intervals += mo.n_intervals;
    // Pile::AddKmers(filtered, kmer_len, sequence) of every mapped read (construct.cc:382)
    if (mo.n_query) {
      RVN_KLAUNCH(kKAddKmers, add_kmers_flags_kernel<<<div_up(mo.n_query, 256), 256, 0, s>>>(
                                  V.packed.as<u64>(), V.word_off.as<u64>(), mo.filtered.as<u8>(), e.query_sketch.org.as<u64>(),
                                  e.query_sketch.read_off.as<u32>(), i + 1, mo.n_query, kmer_len, d_v_kmers_off, d_kmers));
    }
    const u64 O = mo.n_overlaps;
    if (O) {
      Overlap* d_ovl = mo.ovl.as<Overlap>();
      u8* d_ok = e.p2_ok.get<u8>(O + 16);
      u8* d_keep = e.p2_keep.get<u8>(O + 16);
      update_and_identity(e, V, d_ovl, O, d_regions, d_index_of, identity, d_ok);
      classify_kernel<<<div_up(O, 256), 256, 0, s>>>(d_ovl, d_ok, O, d_regions, d_contained, d_keep);
      RVN_LAUNCH_CHECK();
      u32* d_slot = e.p2_slot.get<u32>(O + 2);
      exclusive_scan_u8_u32(d_keep, d_slot, O, e.scan_tmp, s);
      const u64 m = read_back(e, d_slot + O, 4);
      if (m) {
        // append to the result list (grow-preserving)
        if ((acc_n + m + 1) * sizeof(Overlap) > out.ovl.cap) {
          DevBuf bigger;
          bigger.reserve((acc_n + m + 1) * sizeof(Overlap) * 2);
          if (acc_n) RVN_HIP(hipMemcpyAsync(bigger.ptr, out.ovl.ptr, acc_n * sizeof(Overlap), hipMemcpyDeviceToDevice, s));
          RVN_HIP(rvn_stream_sync(s));
          std::swap(out.ovl.ptr, bigger.ptr);
          std::swap(out.ovl.cap, bigger.cap);
        }
        compact_kernel<<<div_up(O, 256), 256, 0, s>>>(d_ovl, d_keep, d_slot, O, out.ovl.as<Overlap>() + acc_n);
        RVN_LAUNCH_CHECK();
        acc_n += m;
      }
    }
    j = i + 1;
  }
  // consecutive overlaps of the same pair keep the longer one (construct.cc:444-453)
  if (acc_n) {
    u8* d_keep = e.p2_keep.get<u8>(acc_n + 16);
    RVN_HIP(hipMemsetAsync(d_keep, 1, acc_n, s));
    dedup_kernel<<<div_up(acc_n, 256), 256, 0, s>>>(out.ovl.as<Overlap>(), acc_n, d_keep);
    RVN_LAUNCH_CHECK();
    acc_n = compact(e, out.ovl, acc_n, d_keep, e.p2_slot, e.p2_tmp_ovl);
  }
  // contained piles become invalid (construct.cc:466-470); the list is re-checked (construct.cc:472-480)
  merge_invalid_kernel<<<div_up(n, 256), 256, 0, s>>>(d_regions, d_contained, n);
  RVN_LAUNCH_CHECK();
  if (acc_n) {
    u8* d_ok = e.p2_ok.get<u8>(acc_n + 16);
    update_kernel<<<div_up(acc_n, 256), 256, 0, s>>>(out.ovl.as<Overlap>(), acc_n, d_regions, d_ok);
    RVN_LAUNCH_CHECK();
    acc_n = compact(e, out.ovl, acc_n, d_ok, e.p2_slot, e.p2_tmp_ovl);
  }
  RVN_HIP(rvn_stream_sync(s));
  out.n_overlaps = acc_n;
}

// The identity filter loop of ResolveContainedReads (construct.cc:162-217) on per-pile overlap lists (CSR, host, in
// place): OverlapUpdate, edlib score of the two spans, survivors keep their updated coordinates and their order.
void identity_filter_lists(Engine& e, const ReadsDev& R, Overlap* h_ovl, u32* h_off, const u32* h_begin, const u32* h_end,
                           const u8* h_invalid, double identity) {
  hipStream_t s = e.stream;
  const u32 n = R.n;
  const u64 O = h_off[n];
  if (O == 0) return;
  PileRegion* d_regions = upload_regions(e, h_begin, h_end, h_invalid, n);
  std::vector<u32> index_of(n);
  for (u32 i = 0; i < n; ++i) index_of[i] = i;
  u32* d_index_of = e.p2_index_of.get<u32>(static_cast<size_t>(n) + 1);
  RVN_HIP(hipMemcpyAsync(d_index_of, index_of.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice, s));
  Overlap* d_ovl = e.p2_tmp_ovl.get<Overlap>(O + 1);
  u8* d_ok = e.p2_ok.get<u8>(O + 16);
  RVN_HIP(hipMemcpyAsync(d_ovl, h_ovl, O * sizeof(Overlap), hipMemcpyHostToDevice, s));
  update_and_identity(e, R, d_ovl, O, d_regions, d_index_of, identity, d_ok);
  std::vector<u8> ok(O);
  std::vector<Overlap> upd(O);
  RVN_HIP(hipMemcpyAsync(ok.data(), d_ok, O, hipMemcpyDeviceToHost, s));
  RVN_HIP(hipMemcpyAsync(upd.data(), d_ovl, O * sizeof(Overlap), hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  u64 k = 0;
  u32 prev_end = h_off[0];
  for (u32 i = 0; i < n; ++i) {  // per-pile compaction of the survivors (construct.cc:208-210)
    const u32 b = prev_end, en = h_off[i + 1];
    prev_end = en;
    h_off[i] = static_cast<u32>(k);
    for (u32 x = b; x < en; ++x)
      if (ok[x]) h_ovl[k++] = upd[x];
  }
  h_off[n] = static_cast<u32>(k);
}

}  // namespace rvn
