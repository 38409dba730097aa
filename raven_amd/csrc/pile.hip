// pile.hip — the serial merge, Pile::AddLayers and the top-kMax truncation of
// FindOverlapsAndCreatePiles (RavenLib/src/construct.cc:72-113, RavenLib/src/pile.cc:12-62),
// for one flush of Map results, entirely on the device.
//
// Reference order of pile p's list after the merge (construct.cc:72-77, serial, submission order):
//   kept overlaps (previous flushes, <= kMax, sorted) ++ reverse(o) for every new o with rhs_id == p in
//   global (query, emission) order ++ new overlaps with lhs_id == p in emission order
// (avoid_symmetric => lhs_id < rhs_id, so the reversed ones always come from earlier queries).
// AddLayers then adds coverage of the NEW overlaps only (construct.cc:89-90), and lists that reached
// kMax are std::sort'ed by length (unstable -> introsort.h) and cut to kMax (construct.cc:92-107).
//
// Pile ids must equal read indices (the reference's own invariant, construct.cc:25,74-75).
#include <algorithm>

#include <cstdlib>
#include "engine.h"
#include "introsort.h"
#include "slopes.h"
#include "wave.h"

namespace rvn {

namespace {

constexpr u32 kPSS = 4;  // pile.h:21

__global__ void rhs_keys_kernel(const Overlap* __restrict__ ovl, u32 n, u32* __restrict__ keys,
                                u32* __restrict__ idx, u32* __restrict__ in_cnt) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 r = ovl[i].rhs_id;
  keys[i] = r;
  idx[i] = i;
  atomicAdd(&in_cnt[r], 1u);
}

// tot[p] = kept + new when the pile has new overlaps, else 0 (list untouched)
__global__ void pile_counts_kernel(const u32* __restrict__ in_cnt, const u32* __restrict__ ovl_read_off, u32 first,
                                   u32 last, const u32* __restrict__ kept_off, u32 n, u32 kmax,
                                   u32* __restrict__ tot, u32* __restrict__ new_kept_cnt) {
  u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  u32 out = 0;
  if (p >= first && p < last) out = ovl_read_off[p - first + 1] - ovl_read_off[p - first];
  const u32 nn = in_cnt[p] + out;
  const u32 kept = kept_off[p + 1] - kept_off[p];
  const u32 t = nn ? kept + nn : 0;
  tot[p] = t;
  new_kept_cnt[p] = nn ? (t < kmax ? t : kmax) : kept;
}

__device__ __forceinline__ Overlap reverse_overlap(const Overlap& o) {  // overlap_utils.cc:5-8
  return Overlap{o.rhs_id, o.rhs_begin, o.rhs_end, o.lhs_id, o.lhs_begin, o.lhs_end, o.score, o.strand};
}
__device__ __forceinline__ u32 overlap_length(const Overlap& o) {  // overlap_utils.cc:10-12
  const u32 a = o.rhs_end - o.rhs_begin, b = o.lhs_end - o.lhs_begin;
  return a > b ? a : b;
}

// one wave per pile: build the merged list
__global__ __launch_bounds__(256) void pile_build_kernel(const Overlap* __restrict__ ovl,
                                                        const u32* __restrict__ ovl_read_off, u32 first, u32 last,
                                                        const u32* __restrict__ in_idx_sorted,
                                                        const u32* __restrict__ in_off,
                                                        const Overlap* __restrict__ kept,
                                                        const u32* __restrict__ kept_off,
                                                        const u32* __restrict__ list_off, u32 n,
                                                        Overlap* __restrict__ list) {
  const u32 p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const u32 lb = list_off[p], le = list_off[p + 1];
  if (le == lb) return;
  const int lane = lane_id();
  const u32 kb = kept_off[p], kn = kept_off[p + 1] - kb;
  const u32 ib = in_off[p], in = in_off[p + 1] - ib;
  for (u32 i = lane; i < kn; i += 64) list[lb + i] = kept[kb + i];
  for (u32 i = lane; i < in; i += 64) list[lb + kn + i] = reverse_overlap(ovl[in_idx_sorted[ib + i]]);
  if (p >= first && p < last) {
    const u32 ob = ovl_read_off[p - first], on = ovl_read_off[p - first + 1] - ob;
    for (u32 i = lane; i < on; i += 64) list[lb + kn + in + i] = ovl[ob + i];
  }
}

// Pile::AddLayers (pile.cc:33-62) as an order-free per-cell sum: coverage(cell) = #{begin events <= cell}
// - #{end events <= cell} in uint32 wrap-around arithmetic (the reference's `coverage` is a uint32_t that is allowed
// to wrap, pile.cc:60), data = clamp(data + coverage) ONCE per cell per call, as the reference's sweep does.
// One workgroup per pile: the events are scattered into a difference array in LDS (+1 at the begin cell, -1 at the
// end cell, wrapping adds), a block-wide prefix sum turns it into the coverage of every cell — O(events + cells),
// whatever the depth of the pile.  Piles longer than the LDS tile are processed tile by tile with a carried prefix.
constexpr u32 kCellTile = 8192;
__device__ __forceinline__ u32 diff_slot(u32 c) { return c + (c >> 5); }  // one pad word per 32: segment scans hit distinct banks
__global__ __launch_bounds__(256) void add_layers_kernel(const Overlap* __restrict__ list,
                                                        const u32* __restrict__ list_off,
                                                        const u32* __restrict__ kept_off,
                                                        const u64* __restrict__ pile_off,
                                                        const u32* __restrict__ ids, u16* __restrict__ data) {
  __shared__ u32 diff[kCellTile + kCellTile / 32 + 1];
  __shared__ u32 red[4];
  const u32 p = blockIdx.x;
  const u32 lb = list_off[p], le = list_off[p + 1];
  if (le == lb) return;
  const u32 kn = kept_off[p + 1] - kept_off[p];
  const u32 nb = lb + kn;  // first new overlap
  const u32 nn = le - nb;
  if (nn == 0) return;
  const u32 id = ids[p];
  const u64 d0 = pile_off[p];
  const u32 cells = static_cast<u32>(pile_off[p + 1] - d0);
  for (u32 tile_lo = 0; tile_lo < cells; tile_lo += kCellTile) {
    const u32 tile_n = min(kCellTile, cells - tile_lo);
    __syncthreads();
    for (u32 c = threadIdx.x; c < diff_slot(tile_n) + 1; c += 256) diff[c] = 0;
    __syncthreads();
    u32 carry_part = 0;  // events before this tile
    for (u32 i = threadIdx.x; i < nn; i += 256) {
      const Overlap o = list[nb + i];
      u32 b, e;
      if (o.lhs_id == id) {
        b = (o.lhs_begin >> kPSS) + 1;
        e = (o.lhs_end >> kPSS) - 1;
      } else if (o.rhs_id == id) {
        b = (o.rhs_begin >> kPSS) + 1;
        e = (o.rhs_end >> kPSS) - 1;
      } else {
        continue;
      }
      // the reference stores events as (x << 1 | flag) in 32 bits and sweeps over x = event >> 1
      b = (b << 1) >> 1;
      e = ((e << 1) | 1u) >> 1;
      if (b < tile_lo) carry_part += 1u;
      else if (b - tile_lo < tile_n) atomicAdd(&diff[diff_slot(b - tile_lo)], 1u);
      if (e < tile_lo) carry_part -= 1u;
      else if (e - tile_lo < tile_n) atomicAdd(&diff[diff_slot(e - tile_lo)], 0xFFFFFFFFu);
    }
    __syncthreads();
    const u32 seg = (tile_n + 255) / 256;
    const u32 lo = min(threadIdx.x * seg, tile_n), hi = min(lo + seg, tile_n);
    u32 seg_sum = 0;
    for (u32 c = lo; c < hi; ++c) seg_sum += diff[diff_slot(c)];
    // coverage just before this thread's first cell = all events before the tile + the segments of the threads before
    u32 carry_total = 0, seg_total = 0;
    (void)block_exclusive_sum_256(carry_part, red, &carry_total);
    const u32 excl = block_exclusive_sum_256(seg_sum, red, &seg_total);
    u32 run = carry_total + excl;
    for (u32 c = lo; c < hi; ++c) {
      run += diff[diff_slot(c)];
      if (run) {
        const u32 v = static_cast<u32>(data[d0 + tile_lo + c]) + run;
        data[d0 + tile_lo + c] = static_cast<u16>(v < 65535u ? v : 65535u);
      }
    }
  }
}

// one lane per pile: exact std::sort on (length << 32 | local index), top-kMax
__global__ void truncate_sort_kernel(const Overlap* __restrict__ list, const u32* __restrict__ list_off, u32 n,
                                     u32 kmax, u64* __restrict__ keys) {
  u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const u32 lb = list_off[p], t = list_off[p + 1] - lb;
  if (t == 0) return;
  u64* kk = keys + lb;
  for (u32 i = 0; i < t; ++i) kk[i] = (static_cast<u64>(overlap_length(list[lb + i])) << 32) | i;
  if (t >= kmax) std_sort(kk, kk + t, LenDesc());
}

// one wave per pile: write the new kept list
__global__ __launch_bounds__(256) void kept_write_kernel(const Overlap* __restrict__ list,
                                                        const u32* __restrict__ list_off,
                                                        const u64* __restrict__ keys,
                                                        const Overlap* __restrict__ kept_old,
                                                        const u32* __restrict__ kept_off_old,
                                                        const u32* __restrict__ kept_off_new, u32 n,
                                                        Overlap* __restrict__ kept_new) {
  const u32 p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const int lane = lane_id();
  const u32 nb = kept_off_new[p], nn = kept_off_new[p + 1] - nb;
  const u32 lb = list_off[p], t = list_off[p + 1] - lb;
  if (t == 0) {
    const u32 ob = kept_off_old[p];
    for (u32 i = lane; i < nn; i += 64) kept_new[nb + i] = kept_old[ob + i];
  } else {
    for (u32 i = lane; i < nn; i += 64) kept_new[nb + i] = list[lb + static_cast<u32>(keys[lb + i])];
  }
}

}  // namespace

void piles_init(Engine& e, const ReadsDev& r, PileState& ps) {
  ps.n = r.n;
  std::vector<u64> off(static_cast<size_t>(r.n) + 1, 0);
  for (u32 i = 0; i < r.n; ++i) off[i + 1] = off[i] + (r.h_len[i] >> kPSS);
  ps.pile_words = off[r.n];
  u64* d_off = ps.pile_off.get<u64>(off.size());
  RVN_HIP(hipMemcpyAsync(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice, e.stream));
  u16* d = ps.pile_data.get<u16>(ps.pile_words + 1);
  RVN_HIP(hipMemsetAsync(d, 0, (ps.pile_words + 1) * 2, e.stream));
  u32* ko = ps.kept_off.get<u32>(static_cast<size_t>(r.n) + 1);
  RVN_HIP(hipMemsetAsync(ko, 0, (static_cast<size_t>(r.n) + 1) * 4, e.stream));
  ps.kept.reserve(64);
  ps.kept_total = 0;
  RVN_HIP(rvn_stream_sync(e.stream));  // `off` is a stack-owned host buffer
}

}  // namespace rvn
#include "kmer.h"
#include "lowcomplexity.h"
namespace rvn {
namespace {
// Pile::AddKmers for a batch of reads: one lane per filtered position; marks out[pile_off[read] + (pos >> 4)].
__global__ void add_kmers_kernel(const u64* __restrict__ packed, const u64* __restrict__ word_off,
                                 const u32* __restrict__ pos, const u32* __restrict__ pos_read, u64 n, u32 k,
                                 const u64* __restrict__ out_off, u8* __restrict__ out) {
  const u64 q = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const u32 r = pos_read[q], p = pos[q];
  const u64* w = packed + word_off[r];
  const u64 mask = (1ULL << (2 * k)) - 1;
  const u32 bit = 2 * p;
  const u64 x = extract_bits(w[bit >> 6], w[(bit >> 6) + 1], bit & 63, mask);
  u8 codes[32];
  for (u32 i = 0; i < k; ++i) codes[i] = static_cast<u8>((x >> (2 * i)) & 3);
  if (lc_kmer_passes(codes, k)) out[out_off[r] + (p >> kPSS)] = 1;
}
}  // namespace

// h_pos: filtered positions of all reads concatenated, h_pos_off[n_reads+1]; h_out: per read (len>>4)+1 bytes,
// concatenated at h_out_off (OR-ed into the existing content, like repeated AddKmers calls)
void pile_add_kmers_batch(Engine& e, const ReadsDev& r, const u32* h_pos, const u64* h_pos_off, u32 n_reads,
                          u32 first_read, u8* h_out, const u64* h_out_off) {
  const u64 n = h_pos_off[n_reads];
  if (n == 0) return;
  hipStream_t s = e.stream;
  std::vector<u32> pos_read(n);
  for (u32 i = 0; i < n_reads; ++i)
    for (u64 q = h_pos_off[i]; q < h_pos_off[i + 1]; ++q) pos_read[q] = first_read + i;
  // out_off indexed by absolute read index
  std::vector<u64> off(static_cast<size_t>(r.n) + 1, 0);
  for (u32 i = 0; i < n_reads; ++i) off[first_read + i] = h_out_off[i];
  const u64 out_total = h_out_off[n_reads];
  u32* d_pos = e.tmp_a.get<u32>(n + 1);
  u32* d_pr = e.tmp_b.get<u32>(n + 1);
  u64* d_off = e.tmp_c.get<u64>(off.size() + 1);
  u8* d_out = e.tmp_d.get<u8>(out_total + 16);
  RVN_HIP(hipMemcpyAsync(d_pos, h_pos, n * 4, hipMemcpyHostToDevice, s));
  RVN_HIP(hipMemcpyAsync(d_pr, pos_read.data(), n * 4, hipMemcpyHostToDevice, s));
  RVN_HIP(hipMemcpyAsync(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice, s));
  RVN_HIP(hipMemcpyAsync(d_out, h_out, out_total, hipMemcpyHostToDevice, s));
  RVN_KLAUNCH(kKAddKmers, add_kmers_kernel<<<div_up(n, 256), 256, 0, s>>>(r.packed.as<u64>(), r.word_off.as<u64>(), d_pos,
                                                                          d_pr, n, e.k, d_off, d_out));
  RVN_HIP(hipMemcpyAsync(h_out, d_out, out_total, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
}

void pile_add_layers_single(Engine& e, PileState& ps, const u32* d_ids, const Overlap* h_ovl, u32 n) {
  hipStream_t s = e.stream;
  Overlap* list = ps.new_list.get<Overlap>(static_cast<size_t>(n) + 1);
  RVN_HIP(hipMemcpyAsync(list, h_ovl, static_cast<size_t>(n) * sizeof(Overlap), hipMemcpyHostToDevice, s));
  u32* offs = ps.tmp3.get<u32>(4);
  const u32 h_offs[4] = {0, n, 0, 0};  // list_off = {0, n}; kept_off = {0, 0}
  RVN_HIP(hipMemcpyAsync(offs, h_offs, sizeof(h_offs), hipMemcpyHostToDevice, s));
  RVN_KLAUNCH(kKAddLayers, add_layers_kernel<<<1, 256, 0, s>>>(list, offs, offs + 2, ps.pile_off.as<u64>(), d_ids, ps.pile_data.as<u16>()));
  RVN_HIP(rvn_stream_sync(s));
}

void piles_merge(Engine& e, const ReadsDev& r, const MapOut& mo, u32 kmax, PileState& ps) {
  const u32 O = static_cast<u32>(mo.n_overlaps);
  if (O == 0) return;
  hipStream_t s = e.stream;
  const u32 n = ps.n;
  const Overlap* ovl = mo.ovl.as<Overlap>();
  const u32* ovl_read_off = mo.ovl_read_off.as<u32>();

  StageTimer tm(e, StageTimes::kMerge);
  u32* keys0 = ps.tmp1.get<u32>(static_cast<size_t>(O) * 2 + 2);
  u32* keys1 = keys0 + O + 1;
  u32* idx0 = ps.tmp2.get<u32>(static_cast<size_t>(O) * 2 + 2);
  u32* idx1 = idx0 + O + 1;
  // tmp3: in_cnt[n+1], in_off[n+1], tot[n+1], list_off[n+1], new_kept_cnt[n+1], new_kept_off[n+1]
  const size_t stride = static_cast<size_t>(n) + 2;
  u32* base = ps.tmp3.get<u32>(stride * 6);
  u32 *in_cnt = base, *in_off = base + stride, *tot = base + 2 * stride, *list_off = base + 3 * stride,
      *new_kept_cnt = base + 4 * stride, *new_kept_off = base + 5 * stride;
  RVN_HIP(hipMemsetAsync(in_cnt, 0, stride * 4, s));
  RVN_KLAUNCH(kKPileKeys, rhs_keys_kernel<<<div_up(O, 256), 256, 0, s>>>(ovl, O, keys0, idx0, in_cnt));
  const int cur = radix_sort_pairs_u32_u32(keys0, keys1, idx0, idx1, O, 32, e.sort_tmp, e.scan_tmp, s, kKPileSortUp,
                                           kKPileSortDown);
  const u32* in_idx_sorted = cur ? idx1 : idx0;
  exclusive_scan_u32_u32(in_cnt, in_off, n, e.scan_tmp, s);
  RVN_KLAUNCH(kKPileCounts, pile_counts_kernel<<<div_up(n, 256), 256, 0, s>>>(in_cnt, ovl_read_off, mo.first, mo.last, ps.kept_off.as<u32>(), n,
                                                    kmax, tot, new_kept_cnt));
  exclusive_scan_u32_u32(tot, list_off, n, e.scan_tmp, s);
  exclusive_scan_u32_u32(new_kept_cnt, new_kept_off, n, e.scan_tmp, s);
  RVN_HIP(hipMemcpyAsync(e.h_pin, list_off + n, 4, hipMemcpyDeviceToHost, s));
  RVN_HIP(hipMemcpyAsync(e.h_pin + 1, new_kept_off + n, 4, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  const u32 L = static_cast<u32>(e.h_pin[0]), K = static_cast<u32>(e.h_pin[1]);
  Overlap* list = ps.new_list.get<Overlap>(static_cast<size_t>(L) + 1);
  RVN_KLAUNCH(kKPileBuild, pile_build_kernel<<<div_up(n, 4), 256, 0, s>>>(ovl, ovl_read_off, mo.first, mo.last, in_idx_sorted, in_off,
                                                 ps.kept.as<Overlap>(), ps.kept_off.as<u32>(), list_off, n, list));
  tm.stop();
  {
    StageTimer t(e, StageTimes::kPile);
    RVN_KLAUNCH(kKAddLayers, add_layers_kernel<<<n, 256, 0, s>>>(list, list_off, ps.kept_off.as<u32>(), ps.pile_off.as<u64>(),
                                        r.id.as<u32>(), ps.pile_data.as<u16>()));
    t.stop();
  }
  {
    StageTimer t(e, StageTimes::kTruncate);
    u64* skeys = ps.tmp4.get<u64>(static_cast<size_t>(L) + 1);
    RVN_KLAUNCH(kKTruncateSort, truncate_sort_kernel<<<div_up(n, 64), 64, 0, s>>>(list, list_off, n, kmax, skeys));
    Overlap* kept_new = ps.tmp5.get<Overlap>(static_cast<size_t>(K) + 1);
    RVN_KLAUNCH(kKKeptWrite, kept_write_kernel<<<div_up(n, 4), 256, 0, s>>>(list, list_off, skeys, ps.kept.as<Overlap>(),
                                                   ps.kept_off.as<u32>(), new_kept_off, n, kept_new));
    // adopt: kept <- kept_new, kept_off <- new_kept_off
    std::swap(ps.kept.ptr, ps.tmp5.ptr);
    std::swap(ps.kept.cap, ps.tmp5.cap);
    u32* ko = ps.kept_off.get<u32>(static_cast<size_t>(n) + 1);
    RVN_HIP(hipMemcpyAsync(ko, new_kept_off, (static_cast<size_t>(n) + 1) * 4, hipMemcpyDeviceToDevice, s));
    ps.kept_total = K;
    t.stop();
  }
}


// ---- raven::Pile::FindValidRegion(coverage) + UpdateValidRegion + FindMedian on every pile of a pass ----------------
// (RavenLib/src/pile.cc:122-174 as called by TrimAndAnnotatePiles, construct.cc:131-139; SURVEY 8(f) rank 1: consumes
// the coverage arrays where they are, in HBM.)  One wave per pile.  Valid region = the first longest maximal run of
// cells >= coverage that is terminated by a lower cell (the reference never records a run that reaches the end of the
// pile); median = radix select (two 8-bit passes over an LDS histogram) of the element at sorted position size / 2.
namespace {
__global__ __launch_bounds__(256) void pile_trim_kernel(u16* __restrict__ data, const u64* __restrict__ pile_off, u32 n,
                                                       u32 coverage, u32 min_cells, u32* __restrict__ out_begin,
                                                       u32* __restrict__ out_end, u16* __restrict__ out_median,
                                                       u8* __restrict__ out_invalid) {
  __shared__ u32 s_hist[4][256];
  const u32 wv = threadIdx.x >> 6;
  const u32 pile = blockIdx.x * 4 + wv;
  if (pile >= n) return;
  const int lane = lane_id();
  const u64 off = pile_off[pile];
  const u32 len = static_cast<u32>(pile_off[pile + 1] - off);
  u16* d = data + off;
  u32 best_b = 0, best_e = 0, run_start = 0;
  bool in_run = false;
  for (u32 base = 0; base < len; base += 64) {
    const u32 cnt = len - base < 64 ? len - base : 64;
    const bool ge = static_cast<u32>(lane) < cnt && d[base + lane] >= coverage;
    const unsigned long long m = __ballot(ge);
    const unsigned long long valid = cnt == 64 ? ~0ULL : ((1ULL << cnt) - 1ULL);
    u32 pos = 0;
    while (pos < cnt) {
      if (in_run) {
        const unsigned long long z = (~m & valid) >> pos;  // next cell below the threshold
        if (!z) break;
        const u32 p = pos + static_cast<u32>(__builtin_ctzll(z));
        const u32 rl_ = base + p - run_start;
        if (best_e - best_b < rl_) {
          best_b = run_start;
          best_e = base + p;
        }
        in_run = false;
        pos = p + 1;
      } else {
        const unsigned long long o = m >> pos;
        if (!o) break;
        const u32 p = pos + static_cast<u32>(__builtin_ctzll(o));
        run_start = base + p;
        in_run = true;
        pos = p + 1;
      }
    }
  }
  const bool invalid = best_b >= best_e || best_e - best_b < min_cells;
  if (invalid) {
    if (lane == 0) {
      out_begin[pile] = 0;
      out_end[pile] = len;
      out_median[pile] = 0;
      out_invalid[pile] = 1;
    }
    return;
  }
  for (u32 i = lane; i < len; i += 64)
    if (i < best_b || i >= best_e) d[i] = 0;
  // median of d[best_b, best_e)
  const u32 msize = best_e - best_b;
  u32 rank = msize / 2;
  u32* hist = s_hist[wv];
  u32 prefix_val = 0;  // high byte once known
  for (int pass = 0; pass < 2; ++pass) {
    for (u32 i = lane; i < 256; i += 64) hist[i] = 0;
    __builtin_amdgcn_wave_barrier();
    for (u32 i = best_b + lane; i < best_e; i += 64) {
      const u32 v = d[i];
      if (pass == 0) atomicAdd(&hist[v >> 8], 1u);
      else if ((v >> 8) == prefix_val) atomicAdd(&hist[v & 255u], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    const u32 h0 = hist[lane * 4], h1 = hist[lane * 4 + 1], h2 = hist[lane * 4 + 2], h3 = hist[lane * 4 + 3];
    const u32 sum4 = h0 + h1 + h2 + h3;
    const u32 incl = wave_inclusive_sum(sum4);
    const u32 excl = incl - sum4;
    const unsigned long long hit = __ballot(rank >= excl && rank < incl);
    const int src = __builtin_ctzll(hit);
    // the lane that owns the bin resolves it
    u32 bin = 0, before = 0;
    if (lane == src) {
      u32 c = excl;
      bin = lane * 4;
      before = c;
      if (rank >= c + h0) {
        c += h0;
        bin = lane * 4 + 1;
        before = c;
        if (rank >= c + h1) {
          c += h1;
          bin = lane * 4 + 2;
          before = c;
          if (rank >= c + h2) {
            c += h2;
            bin = lane * 4 + 3;
            before = c;
          }
        }
      }
    }
    bin = static_cast<u32>(__shfl(static_cast<int>(bin), src, 64));
    before = static_cast<u32>(__shfl(static_cast<int>(before), src, 64));
    rank -= before;
    if (pass == 0) prefix_val = bin;
    else prefix_val = (prefix_val << 8) | bin;
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    out_begin[pile] = best_b;
    out_end[pile] = best_e;
    out_median[pile] = static_cast<u16>(prefix_val);
    out_invalid[pile] = 0;
  }
}
}  // namespace

void piles_trim_and_median(Engine& e, PileState& ps, u32 coverage, u32* h_begin, u32* h_end, u16* h_median, u8* h_invalid) {
  const u32 n = ps.n;
  if (n == 0) return;
  hipStream_t s = e.stream;
  u32* d_begin = e.tmp_a.get<u32>(static_cast<size_t>(n) + 1);
  u32* d_end = e.tmp_b.get<u32>(static_cast<size_t>(n) + 1);
  u16* d_med = e.tmp_c.get<u16>(static_cast<size_t>(n) + 1);
  u8* d_inv = e.tmp_d.get<u8>(static_cast<size_t>(n) + 1);
  RVN_KLAUNCH(kKPileTrim, pile_trim_kernel<<<div_up(n, 4), 256, 0, s>>>(ps.pile_data.as<u16>(), ps.pile_off.as<u64>(), n,
                                                                       coverage, 1260u >> kPSS, d_begin, d_end, d_med, d_inv));
  if (h_begin) RVN_HIP(hipMemcpyAsync(h_begin, d_begin, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost, s));
  if (h_end) RVN_HIP(hipMemcpyAsync(h_end, d_end, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost, s));
  if (h_median) RVN_HIP(hipMemcpyAsync(h_median, d_med, static_cast<size_t>(n) * 2, hipMemcpyDeviceToHost, s));
  if (h_invalid) RVN_HIP(hipMemcpyAsync(h_invalid, d_inv, static_cast<size_t>(n), hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
}


// ---- Pile::FindChimericRegions for every valid pile (construct.cc:139 inside TrimAndAnnotatePiles) ---------------------
namespace {
// one THREAD per pile: FindSlopes(1.82) + pit pairing + MergeRegions (slopes.h) on the coverage array where it is (piles
// longer than the wave kernel's LDS, and RVN_CHIMERIC_PER_THREAD)
__global__ __launch_bounds__(64) void pile_chimeric_kernel(const u16* __restrict__ data, const u64* __restrict__ pile_off,
                                                          const u8* __restrict__ invalid, u32 n,
                                                          SlopeRegion* __restrict__ slopes, u16* __restrict__ tmp,
                                                          u32* __restrict__ out_tmp, u32* __restrict__ count,
                                                          u32* __restrict__ n_overflow) {
  const u32 p = blockIdx.x * 64 + threadIdx.x;
  if (p >= n) return;
  u32 c = 0;
  if (!invalid[p]) {
    const u64 off = pile_off[p];
    const u32 len = static_cast<u32>(pile_off[p + 1] - off);
    bool overflow = false;
    // Scratch bound (slopes.h): slope regions of one kind are pairwise disjoint, non-empty runs of cells at every moment
    // of the 'separate overlapping slopes' loop — a re-evaluated region is cut back to the part its new runs do not
    // touch — so there are never more than len of each kind, 2 * len in all; the pits paired from them are at most half
    // of the final list (<= len), which is also what the merged-flag scratch (len cells) and the output hold.
    c = find_chimeric_regions(data + off, static_cast<int>(len), slopes + 2 * off, 2 * len, tmp + off, out_tmp + 2 * off, len,
                              &overflow);
    if (overflow) {
      atomicAdd(n_overflow, 1u);
      c = 0;
    }
  }
  count[p] = c;
}
// One WAVE per pile.  The first sweep of FindSlopes — for every cell, is the highest coverage within 52 cells on its left
// (right) above coverage * q — is the wave's: the coverage sits in LDS, windowed maxima come from five doubling passes
// (M32[i] = max of cells i .. i + 31; a 52-cell window is two overlapping M32), the runs of flagged cells from ballots.
// What follows (separating overlapping slopes, narrowing, pit pairing, MergeRegions) works on a handful of regions and
// is lane 0's, on the same code as the one-thread-per-pile kernel and the CPU hook (slopes.h), reading the LDS copy.
constexpr int kChimCells = 4096;  // cells (16 bases each) a wave keeps in LDS; longer piles take the one-thread path
struct alignas(16) ChimLds {
  u16 data[kChimCells + 192];  // cell i at [64 + i]; zeros outside the pile
  u16 a[kChimCells + 192];
  u16 b[kChimCells + 192];
};
__global__ __launch_bounds__(64) void pile_chimeric_wave_kernel(const u16* __restrict__ data, const u64* __restrict__ pile_off,
                                                               const u8* __restrict__ invalid, u32 n,
                                                               SlopeRegion* __restrict__ slopes, u16* __restrict__ tmp,
                                                               u32* __restrict__ out_tmp, u32* __restrict__ count,
                                                               u32* __restrict__ n_overflow) {
  __shared__ ChimLds S;
  const u32 p = blockIdx.x;
  const int lane = static_cast<int>(threadIdx.x);
  if (p >= n) return;
  if (invalid[p]) {
    if (lane == 0) count[p] = 0;
    return;
  }
  const u64 off = pile_off[p];
  const u32 len = static_cast<u32>(pile_off[p + 1] - off);
  SlopeRegion* dst = slopes + 2 * off;
  const u32 cap = 2 * len;
  bool overflow = false;
  u32 c = 0;
  if (len > static_cast<u32>(kChimCells) || len == 0) {
    if (lane == 0 && len) c = find_chimeric_regions(data + off, static_cast<int>(len), dst, cap, tmp + off, out_tmp + 2 * off, len, &overflow);
  } else {
    const int total = static_cast<int>(len) + 192;
    for (int i = lane; i < total; i += 64) {
      const int cell = i - 64;
      S.data[i] = (cell >= 0 && cell < static_cast<int>(len)) ? data[off + cell] : static_cast<u16>(0);
    }
    __syncthreads();
    // M2 -> a, M4 -> b, M8 -> a, M16 -> b, M32 -> a (reads beyond the array's end are never needed: 128 cells of zeros)
    auto pass = [&](const u16* src, u16* out, int half) {
      for (int i = lane; i < total - half; i += 64) {
        const u16 x = src[i], y = src[i + half];
        out[i] = x > y ? x : y;
      }
      for (int i = total - half + lane; i < total; i += 64) out[i] = 0;
      __syncthreads();
    };
    pass(S.data, S.a, 1);
    pass(S.a, S.b, 2);
    pass(S.b, S.a, 4);
    pass(S.a, S.b, 8);
    pass(S.b, S.a, 16);
    const u16* M = S.a + 64;  // M[i] = max(data[i .. i + 31])
    const u16* D = S.data + 64;
    const int w = 847 >> 4;   // 52
    auto flags_of = [&](int i, bool& down, bool& up) {
      const u16 d = static_cast<u16>(slope_clamp(static_cast<double>(D[i]) * 1.82));
      const u16 l0 = M[i - w], l1 = M[i - 32];
      const u16 r0 = M[i + 1], r1 = M[i + w - 31];
      const u16 lmax = l0 > l1 ? l0 : l1, rmax = r0 > r1 ? r0 : r1;
      down = lmax > d;
      up = rmax > d;
    };
    SlopeRegion* ups = reinterpret_cast<SlopeRegion*>(out_tmp + 2 * off);  // room for len regions; ups are at most len / 2
    u32 nd = 0, nde = 0, nu = 0, nue = 0;          // runs started / ended so far, per kind
    bool carry_d = false, carry_u = false;        // flag of the cell before the chunk
    for (u32 c0 = 0; c0 < len; c0 += 64) {
      const int i = static_cast<int>(c0) + lane;
      bool fd = false, fu = false;
      if (i < static_cast<int>(len)) flags_of(i, fd, fu);
      bool nd_next = false, nu_next = false;      // flag of the first cell of the next chunk (wave-uniform)
      if (c0 + 64 < len) flags_of(static_cast<int>(c0) + 64, nd_next, nu_next);
      const unsigned long long bd = __ballot(fd), bu = __ballot(fu);
      const unsigned long long sd = bd & ~((bd << 1) | (carry_d ? 1ULL : 0ULL)), su = bu & ~((bu << 1) | (carry_u ? 1ULL : 0ULL));
      const unsigned long long ed = bd & ~((bd >> 1) | (nd_next ? 1ULL << 63 : 0ULL)), eu = bu & ~((bu >> 1) | (nu_next ? 1ULL << 63 : 0ULL));
      const unsigned long long below = (1ULL << lane) - 1ULL;
      // downs go straight to dst, ups to the (still unused) output scratch and behind the downs afterwards
      if ((sd >> lane) & 1ULL) {
        const u32 k = nd + static_cast<u32>(__popcll(sd & below));
        if (k < cap) dst[k].first = static_cast<u32>(i) << 1;
        else overflow = true;
      }
      if ((ed >> lane) & 1ULL) {
        const u32 k = nde + static_cast<u32>(__popcll(ed & below));
        if (k < cap) dst[k].second = static_cast<u32>(i);
      }
      if ((su >> lane) & 1ULL) {
        const u32 k = nu + static_cast<u32>(__popcll(su & below));
        if (k < len) ups[k].first = static_cast<u32>(i) << 1 | 1u;
        else overflow = true;
      }
      if ((eu >> lane) & 1ULL) {
        const u32 k = nue + static_cast<u32>(__popcll(eu & below));
        if (k < len) ups[k].second = static_cast<u32>(i);
      }
      nd += static_cast<u32>(__popcll(sd));
      nde += static_cast<u32>(__popcll(ed));
      nu += static_cast<u32>(__popcll(su));
      nue += static_cast<u32>(__popcll(eu));
      carry_d = (bd >> 63) & 1ULL;
      carry_u = (bu >> 63) & 1ULL;
    }
    overflow = __ballot(overflow) != 0 || nd + nu > cap;
    __threadfence_block();
    __syncthreads();
    if (!overflow)
      for (u32 k = lane; k < nu; k += 64) dst[nd + k] = ups[k];  // the ups behind the downs
    __threadfence_block();
    __syncthreads();
    if (lane == 0 && !overflow) {
      const u32 ns = find_slopes_rest(D, 1.82, dst, cap, nd + nu, tmp + off, &overflow);
      c = pair_and_merge_slopes(dst, ns, tmp + off, out_tmp + 2 * off, len, &overflow);
    }
  }
  if (lane == 0) {
    if (overflow) {
      atomicAdd(n_overflow, 1u);
      c = 0;
    }
    count[p] = c;
  }
}
__global__ void chimeric_gather_kernel(const u32* __restrict__ out_tmp, const u64* __restrict__ pile_off,
                                       const u32* __restrict__ count, const u32* __restrict__ roff, u32 n,
                                       u32* __restrict__ regions) {
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const u32 c = count[p];
  const u32* src = out_tmp + 2 * pile_off[p];
  u32* dst = regions + 2ULL * roff[p];
  for (u32 i = 0; i < 2 * c; ++i) dst[i] = src[i];
}
}  // namespace

// h_invalid: the piles' is_invalid flags (from piles_trim_and_median); h_off[n + 1] / regions: CSR of (begin, end) cell pairs
void piles_find_chimeric_regions(Engine& e, PileState& ps, const u8* h_invalid, std::vector<u32>& h_off,
                                 std::vector<u32>& h_regions) {
  hipStream_t s = e.stream;
  const u32 n = ps.n;
  h_off.assign(static_cast<size_t>(n) + 1, 0);
  h_regions.clear();
  if (n == 0 || ps.pile_words == 0) return;
  u8* d_inv = e.tmp_a.get<u8>(static_cast<size_t>(n) + 16);
  RVN_HIP(hipMemcpyAsync(d_inv, h_invalid, n, hipMemcpyHostToDevice, s));
  SlopeRegion* d_slopes = e.tmp_b.get<SlopeRegion>(2 * ps.pile_words + 2);
  u16* d_tmp = e.tmp_c.get<u16>(ps.pile_words + 1);
  u32* d_out = e.tmp_d.get<u32>(2 * ps.pile_words + 4);
  u32* d_cnt = e.tmp_e.get<u32>(2 * static_cast<size_t>(n) + 8);
  u32* d_roff = d_cnt + n + 1;
  u32* d_ovf = e.tmp_f.get<u32>(4);
  RVN_HIP(hipMemsetAsync(d_ovf, 0, 4, s));
  if (knob("RVN_CHIMERIC_PER_THREAD"))  // (the earlier layout: one thread per pile; kept for comparisons)
    RVN_KLAUNCH(kKPileTrim, pile_chimeric_kernel<<<div_up(n, 64), 64, 0, s>>>(ps.pile_data.as<u16>(), ps.pile_off.as<u64>(), d_inv, n,
                                                                             d_slopes, d_tmp, d_out, d_cnt, d_ovf));
  else
    RVN_KLAUNCH(kKPileTrim, pile_chimeric_wave_kernel<<<n, 64, 0, s>>>(ps.pile_data.as<u16>(), ps.pile_off.as<u64>(), d_inv, n,
                                                                      d_slopes, d_tmp, d_out, d_cnt, d_ovf));
  exclusive_scan_u32_u32(d_cnt, d_roff, n, e.scan_tmp, s);
  RVN_HIP(hipMemcpyAsync(h_off.data(), d_roff, (static_cast<size_t>(n) + 1) * 4, hipMemcpyDeviceToHost, s));
  if (read_back(e, d_ovf, 4) != 0)
    throw HipError("[raven_hip] FindChimericRegions: a pile produced more than two slope regions per cell (internal error: the bound in pile.hip is a proof)");
  RVN_HIP(rvn_stream_sync(s));
  const u32 total = h_off[n];
  h_regions.assign(2ULL * total, 0);
  if (total) {
    u32* d_regions = e.sort_tmp.get<u32>(2ULL * total + 2);
    chimeric_gather_kernel<<<div_up(n, 256), 256, 0, s>>>(d_out, ps.pile_off.as<u64>(), d_cnt, d_roff, n, d_regions);
    RVN_LAUNCH_CHECK();
    RVN_HIP(hipMemcpyAsync(h_regions.data(), d_regions, 2ULL * total * 4, hipMemcpyDeviceToHost, s));
    RVN_HIP(rvn_stream_sync(s));
  }
}

}  // namespace rvn
