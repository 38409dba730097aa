// sketch.hip — minimizer sketch of 2-bit packed reads (replaces ram::MinimizerEngine::Minimize(seq, minhash),
// call sites RavenLib/src/construct.cc:42-43 (index side) and :62 (query side, always minhash)).
//
// Closed form of ram's deque winnowing (proved in DESIGN.md §3.1): with P = len-k+1 k-mer positions,
// h(p) the hashed canonical k-mer (palindromes excluded) and windows s in [0, P-w] covering positions
// [s, s+w-1], position p is a minimizer iff h(p) == min_{q in window s} h(q) for some window s containing p.
// The reference emits them in position order.  Every position is therefore independent:
//   * one workgroup per tile of 1024 positions (+ w-1 halo each side) of one read,
//   * the tile's packed words are staged once in LDS (coalesced 8-byte loads, <= 52 words),
//   * every lane extracts its k-mers straight from the LDS words (no rolling state),
//   * window minima M[s] and the per-position test run out of LDS,
//   * compaction in position order by wave ballots.
// Two passes (count, scan, write) recompute the hashes instead of spilling a worst-case buffer:
// HBM traffic is N/4 bytes read per pass + 12|16 B per emitted minimizer.
#include <stdexcept>
#include <string>

#include "engine.h"
#include "kmer.h"
#include "wave.h"

namespace rvn {

namespace {

constexpr int kThreads = 256;
constexpr int kHaloMax = kMaxWindow - 1;
constexpr int kHN = kSketchTile + 2 * kHaloMax;          // hashed positions incl. halos
constexpr int kWordsMax = (kHN + 31 + 31) / 32 + 3;       // packed words covering them

template <typename V>
struct Inval {
  static constexpr V value = static_cast<V>(~static_cast<V>(0));
};

template <typename V, bool WRITE>
__global__ __launch_bounds__(kThreads) void sketch_kernel(const u64* __restrict__ packed,
                                                         const u64* __restrict__ word_off,
                                                         const u32* __restrict__ lens, const u32* __restrict__ ids,
                                                         const u32* __restrict__ tile_read,
                                                         const u32* __restrict__ tile_start, u32 tile_first, u32 k,
                                                         u32 w, u32* __restrict__ tile_cnt,
                                                         const u32* __restrict__ tile_off, V* __restrict__ out_val,
                                                         u64* __restrict__ out_org) {
  __shared__ u64 s_words[kWordsMax];
  __shared__ V s_h[kHN];
  __shared__ V s_m[kSketchTile + kHaloMax];
  __shared__ u8 s_strand[kHN];
  __shared__ u32 s_wtot[4][4];

  const u32 t = tile_first + blockIdx.x;
  const u32 r = tile_read[t];
  const u32 s0 = tile_start[t];
  const u32 len = lens[r];
  const u32 P = len - k + 1;  // tiles exist only when len >= k + w - 1
  const u32 ntile = min(static_cast<u32>(kSketchTile), P - s0);
  const u32 halo = w - 1;
  const u32 HN = ntile + 2 * halo;
  const u64 mask = (k == 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);

  // positions covered: [plo, phi] clipped to [0, P-1]
  const long long pfirst = static_cast<long long>(s0) - halo;  // position of s_h[0]
  const u32 plo = pfirst < 0 ? 0u : static_cast<u32>(pfirst);
  const u32 phi = min(P - 1, s0 + ntile - 1 + halo);
  const u32 wlo = plo >> 5;
  const u32 whi = ((phi + k - 1) >> 5) + 1;  // +1: second word of a straddling extract (pad word exists)
  const u64* rw = packed + word_off[r];
  for (u32 i = threadIdx.x; i <= whi - wlo; i += kThreads) s_words[i] = rw[wlo + i];
  __syncthreads();

  for (u32 j = threadIdx.x; j < HN; j += kThreads) {
    const long long p = pfirst + j;
    V hv = Inval<V>::value;
    unsigned strand = 0;
    if (p >= 0 && p < static_cast<long long>(P)) {
      const u32 bit = 2u * static_cast<u32>(p) - 64u * wlo;
      const u32 wi = bit >> 6;
      const u64 x = extract_bits(s_words[wi], s_words[wi + 1], bit & 63u, mask);
      V val;
      if (canonical_hash<V>(x, k, mask, &val, &strand)) hv = val;
    }
    s_h[j] = hv;
    s_strand[j] = static_cast<u8>(strand);
  }
  __syncthreads();

  // window minima: window j starts at position pfirst + j and covers s_h[j .. j+w-1]
  const u32 nwin = ntile + halo;
  for (u32 j = threadIdx.x; j < nwin; j += kThreads) {
    const long long ws = pfirst + j;
    V m = Inval<V>::value;
    if (ws >= 0 && ws + w <= static_cast<long long>(P)) {
      for (u32 q = 0; q < w; ++q) {
        V x = s_h[j + q];
        m = x < m ? x : m;
      }
    }
    s_m[j] = m;
  }
  __syncthreads();

  // selection + ordered compaction, 4 rows of 256 positions
  const int lane = lane_id();
  const int wv = threadIdx.x >> 6;
  bool sel[4];
  u32 pre[4];
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const u32 q = row * kThreads + threadIdx.x;  // tile coordinate
    bool f = false;
    if (q < ntile) {
      const V v = s_h[q + halo];
      if (v != Inval<V>::value) {
        for (u32 j = 0; j < w; ++j) f = f || (s_m[q + j] == v);  // windows starting at p-w+1 .. p
      }
    }
    sel[row] = f;
    const unsigned long long b = __ballot(f);
    pre[row] = __popcll(b & lanemask_lt());
    if (lane == 0) s_wtot[row][wv] = __popcll(b);
  }
  __syncthreads();
  u32 total = 0;
  u32 base[4];
#pragma unroll
  for (int row = 0; row < 4; ++row) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      if (x == wv) base[row] = total;
      total += s_wtot[row][x];
    }
  }
  if (!WRITE) {
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
    return;
  }
  const u64 obase = tile_off[blockIdx.x];
  const u64 idhi = static_cast<u64>(ids[r]) << 32;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    if (sel[row]) {
      const u32 q = row * kThreads + threadIdx.x;
      const u64 o = obase + base[row] + pre[row];
      out_val[o] = s_h[q + halo];
      out_org[o] = idhi | (static_cast<u64>(s0 + q) << 1) | s_strand[q + halo];
    }
  }
}

__global__ void gather_u32_kernel(const u32* __restrict__ src, const u32* __restrict__ idx, u32 idx_sub,
                                  u32* __restrict__ dst, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i] - idx_sub];
}

__device__ __forceinline__ u64 shr_safe(u64 v, int s) { return s >= 64 ? 0 : (v >> s); }

// minhash: per read keep the len/k smallest values (ties by position), ram Minimize(seq, true).
// One workgroup per read: MSD radix-select of the rank-R value, then an ordered flag pass.
template <typename V>
__global__ __launch_bounds__(kThreads) void minhash_select_kernel(const V* __restrict__ val,
                                                                 const u32* __restrict__ read_off,
                                                                 const u32* __restrict__ lens, u32 first, u32 k,
                                                                 int top_shift, u8* __restrict__ flags,
                                                                 u64* __restrict__ org_flag,
                                                                 unsigned long long* __restrict__ kept_total) {
  __shared__ u32 hist[256];
  __shared__ u64 s_prefix;
  __shared__ u32 s_remaining;
  __shared__ u32 s_wt[4];
  __shared__ u32 s_run;
  const u32 r = blockIdx.x;
  const u32 b = read_off[r], e = read_off[r + 1];
  const u32 n = e - b;
  if (n == 0) return;
  const u32 R = min(n, lens[first + r] / k);
  const V* v = val + b;
  u8* fl = flags + b;
  u64* og = org_flag ? org_flag + b : nullptr;
  if (kept_total && threadIdx.x == 0) atomicAdd(kept_total, static_cast<unsigned long long>(R));
  if (R == n || R == 0) {
    for (u32 i = threadIdx.x; i < n; i += kThreads) {
      fl[i] = R ? 1 : 0;
      if (og && R) og[i] |= kQueryFlag;
    }
    return;
  }
  if (threadIdx.x == 0) {
    s_prefix = 0;
    s_remaining = R;
  }
  for (int shift = top_shift; shift >= 0; shift -= 8) {
    hist[threadIdx.x] = 0;
    __syncthreads();
    const u64 prefix = s_prefix;
    for (u32 i = threadIdx.x; i < n; i += kThreads) {
      const u64 x = v[i];
      if (shr_safe(x, shift + 8) == prefix) atomicAdd(&hist[(x >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int l = threadIdx.x;
      const u32 h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
      const u32 s = h0 + h1 + h2 + h3;
      const u32 inc = wave_inclusive_sum(s);
      const u32 rem = s_remaining;
      const unsigned long long crossed = __ballot(inc >= rem);
      const int first_lane = __ffsll(static_cast<long long>(crossed)) - 1;
      if (l == first_lane) {
        u32 cum = inc - s;
        u32 d;
        if (cum + h0 >= rem) d = 0;
        else if ((cum += h0, cum + h1 >= rem)) d = 1;
        else if ((cum += h1, cum + h2 >= rem)) d = 2;
        else { cum += h2; d = 3; }
        s_remaining = rem - cum;
        s_prefix = (prefix << 8) | (4 * l + d);
      }
    }
    __syncthreads();
  }
  const V T = static_cast<V>(s_prefix);
  const u32 rem = s_remaining;  // how many values == T to keep (first by position)
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  const int lane = lane_id();
  const int wv = threadIdx.x >> 6;
  for (u32 start = 0; start < n; start += kThreads) {
    const u32 i = start + threadIdx.x;
    const bool valid = i < n;
    const V x = valid ? v[i] : static_cast<V>(0);
    const bool eq = valid && x == T;
    const unsigned long long bm = __ballot(eq);
    if (lane == 0) s_wt[wv] = __popcll(bm);
    __syncthreads();
    u32 basec = s_run;
    for (int q = 0; q < wv; ++q) basec += s_wt[q];
    const u32 rank = basec + __popcll(bm & lanemask_lt());
    if (valid) {
      const bool sel = x < T || (eq && rank < rem);
      fl[i] = sel ? 1 : 0;
      if (og && sel) og[i] |= kQueryFlag;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_run += s_wt[0] + s_wt[1] + s_wt[2] + s_wt[3];
    __syncthreads();
  }
}

template <typename V>
__global__ void compact_sketch_kernel(const V* __restrict__ val, const u64* __restrict__ org,
                                      const u8* __restrict__ flags, const u32* __restrict__ scan, u64 n,
                                      V* __restrict__ oval, u64* __restrict__ oorg) {
  u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) {
    u32 o = scan[i];
    oval[o] = val[i];
    oorg[o] = org[i];
  }
}

__global__ void sum_u32_kernel(const u32* __restrict__ v, u32 n, unsigned long long* __restrict__ out) {
  unsigned long long acc = 0;
  for (u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<u64>(gridDim.x) * blockDim.x) acc += v[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

template <typename V>
void sketch_raw_impl(Engine& e, const ReadsDev& r, u32 first, u32 last, Sketch& out) {
  hipStream_t s = e.stream;
  out.first = first;
  out.last = last;
  out.count = 0;
  const u32 nr = last - first;
  u32* read_off = out.read_off.get<u32>(static_cast<size_t>(nr) + 1);
  const u32 tf = r.h_read_tile_off[first], tl = r.h_read_tile_off[last];
  const u32 nt = tl - tf;
  if (nt == 0) {
    RVN_HIP(hipMemsetAsync(read_off, 0, (static_cast<size_t>(nr) + 1) * 4, s));
    out.val.reserve(16);
    out.org.reserve(16);
    return;
  }
  u32* tile_cnt = e.tmp_a.get<u32>(nt);
  u32* tile_off = e.tmp_b.get<u32>(static_cast<size_t>(nt) + 1);
  RVN_KLAUNCH(kKSketchCount, sketch_kernel<V, false><<<nt, kThreads, 0, s>>>(
                                 r.packed.as<u64>(), r.word_off.as<u64>(), r.len.as<u32>(), r.id.as<u32>(),
                                 r.tile_read.as<u32>(), r.tile_start.as<u32>(), tf, e.k, e.w, tile_cnt, nullptr,
                                 nullptr, nullptr));
  // The scan of the tile counts is 32-bit (offsets into one sketch are u32 everywhere downstream).  A tile holds at most
  // kSketchTile minimizers, so fewer than 2^32 / kSketchTile tiles cannot wrap; a larger range (>= 4.3 G k-mer positions)
  // gets its total summed in 64 bits first and is refused when it does not fit (the foreign prefix of a late
  // index batch reaches that regime before any other sketch does — rvn_shard_sketch_range cuts it into bounded pieces)
  u64 limit = 1ULL << 32;
#if defined(RVN_DEBUG_KNOBS)
  if (const char* lim = knob("RVN_SKETCH_LIMIT")) limit = std::strtoull(lim, nullptr, 10);  // (tests: the refusal without 13 Gbases)
#endif
  if (static_cast<u64>(nt) * kSketchTile >= limit) {
    unsigned long long* d_sum = e.sketch_sum.get<unsigned long long>(1);
    RVN_HIP(hipMemsetAsync(d_sum, 0, 8, s));
    sum_u32_kernel<<<std::min<u32>(div_up(nt, 1024), 4096), 256, 0, s>>>(tile_cnt, nt, d_sum);
    RVN_LAUNCH_CHECK();
    const u64 sum = read_back(e, d_sum, 8);
    if (sum >= limit)
      throw std::invalid_argument("[raven_hip] a sketch of " + std::to_string(sum) + " minimizers in one call (reads " + std::to_string(first) +
                                  " .. " + std::to_string(last) + "): 2^32 or more are not supported — sketch the range in pieces");
  }
  exclusive_scan_u32_u32(tile_cnt, tile_off, nt, e.scan_tmp, s);
  const u32 total = static_cast<u32>(read_back(e, tile_off + nt, 4));
  V* val = out.val.get<V>(static_cast<size_t>(total) + 1);
  u64* org = out.org.get<u64>(static_cast<size_t>(total) + 1);
  RVN_KLAUNCH(kKSketchWrite, sketch_kernel<V, true><<<nt, kThreads, 0, s>>>(
                                 r.packed.as<u64>(), r.word_off.as<u64>(), r.len.as<u32>(), r.id.as<u32>(),
                                 r.tile_read.as<u32>(), r.tile_start.as<u32>(), tf, e.k, e.w, nullptr, tile_off, val,
                                 org));
  RVN_KLAUNCH(kKGather, gather_u32_kernel<<<div_up(nr + 1, 256), 256, 0, s>>>(
                            tile_off, r.read_tile_off.as<u32>() + first, tf, read_off, nr + 1));
  out.count = total;
}

template <typename V>
void sketch_minhash_impl(Engine& e, const ReadsDev& r, const Sketch& raw, Sketch& out) {
  hipStream_t s = e.stream;
  const u32 first = raw.first, last = raw.last;
  const u32 nr = last - first;
  const u64 total = raw.count;
  out.first = first;
  out.last = last;
  out.count = 0;
  u32* read_off_final = out.read_off.get<u32>(static_cast<size_t>(nr) + 1);
  if (total == 0) {
    RVN_HIP(hipMemsetAsync(read_off_final, 0, (static_cast<size_t>(nr) + 1) * 4, s));
    out.val.reserve(16);
    out.org.reserve(16);
    return;
  }
  const V* val = raw.val.as<V>();
  const u64* org = raw.org.as<u64>();
  const u32* raw_read_off = raw.read_off.as<u32>();
  u8* flags = e.tmp_c.get<u8>(static_cast<size_t>(total) + 1);
  u32* fscan = e.tmp_d.get<u32>(static_cast<size_t>(total) + 1);
  const int nbytes = (2 * e.k + 7) / 8;
  RVN_KLAUNCH(kKMinhashSelect, minhash_select_kernel<V><<<nr, kThreads, 0, s>>>(val, raw_read_off, r.len.as<u32>(),
                                                                                first, e.k, 8 * (nbytes - 1), flags, nullptr,
                                                                                nullptr));
  exclusive_scan_u8_u32(flags, fscan, total, e.scan_tmp, s);
  const u32 kept = static_cast<u32>(read_back(e, fscan + total, 4));
  V* oval = out.val.get<V>(static_cast<size_t>(kept) + 1);
  u64* oorg = out.org.get<u64>(static_cast<size_t>(kept) + 1);
  RVN_KLAUNCH(kKCompactSketch,
              compact_sketch_kernel<V><<<div_up(total, 256), 256, 0, s>>>(val, org, flags, fscan, total, oval, oorg));
  // read_off_final[i] = fscan[raw_read_off[i]]
  RVN_KLAUNCH(kKGather,
              gather_u32_kernel<<<div_up(nr + 1, 256), 256, 0, s>>>(fscan, raw_read_off, 0, read_off_final, nr + 1));
  out.count = kept;
}

}  // namespace

void reads_build_tiles(Engine& e, ReadsDev& r) {
  const u32 need = e.k + e.w - 1;
  r.h_read_tile_off.assign(static_cast<size_t>(r.n) + 1, 0);
  std::vector<u32> tr, ts;
  for (u32 i = 0; i < r.n; ++i) {
    r.h_read_tile_off[i] = static_cast<u32>(tr.size());
    const u32 len = r.h_len[i];
    if (len < need) continue;
    const u32 P = len - e.k + 1;
    for (u32 s0 = 0; s0 < P; s0 += kSketchTile) {
      tr.push_back(i);
      ts.push_back(s0);
    }
  }
  r.h_read_tile_off[r.n] = static_cast<u32>(tr.size());
  r.n_tiles = static_cast<u32>(tr.size());
  u32* d_tr = r.tile_read.get<u32>(tr.size() + 1);
  u32* d_ts = r.tile_start.get<u32>(ts.size() + 1);
  u32* d_rto = r.read_tile_off.get<u32>(static_cast<size_t>(r.n) + 1);
  if (!tr.empty()) {
    RVN_HIP(hipMemcpy(d_tr, tr.data(), tr.size() * 4, hipMemcpyHostToDevice));
    RVN_HIP(hipMemcpy(d_ts, ts.data(), ts.size() * 4, hipMemcpyHostToDevice));
  }
  RVN_HIP(hipMemcpy(d_rto, r.h_read_tile_off.data(), (static_cast<size_t>(r.n) + 1) * 4, hipMemcpyHostToDevice));
}

void sketch_raw(Engine& e, const ReadsDev& r, u32 first, u32 last, Sketch& out) {
  if (e.val64) sketch_raw_impl<u64>(e, r, first, last, out);
  else sketch_raw_impl<u32>(e, r, first, last, out);
}

void sketch_minhash(Engine& e, const ReadsDev& r, const Sketch& raw, Sketch& out) {
  if (e.val64) sketch_minhash_impl<u64>(e, r, raw, out);
  else sketch_minhash_impl<u32>(e, r, raw, out);
}

u64 sketch_flag_queries(Engine& e, const ReadsDev& r, Sketch& raw) {
  hipStream_t s = e.stream;
  const u64 total = raw.count;
  if (total == 0) return 0;
  const u32 nr = raw.last - raw.first;
  u8* flags = e.tmp_c.get<u8>(static_cast<size_t>(total) + 1);
  unsigned long long* kept = e.tmp_e.get<unsigned long long>(2);
  RVN_HIP(hipMemsetAsync(kept, 0, 8, s));
  const int nbytes = (2 * e.k + 7) / 8;
  if (e.val64) {
    RVN_KLAUNCH(kKMinhashSelect, minhash_select_kernel<u64><<<nr, kThreads, 0, s>>>(
                                     raw.val.as<u64>(), raw.read_off.as<u32>(), r.len.as<u32>(), raw.first, e.k,
                                     8 * (nbytes - 1), flags, raw.org.as<u64>(), kept));
  } else {
    RVN_KLAUNCH(kKMinhashSelect, minhash_select_kernel<u32><<<nr, kThreads, 0, s>>>(
                                     raw.val.as<u32>(), raw.read_off.as<u32>(), r.len.as<u32>(), raw.first, e.k,
                                     8 * (nbytes - 1), flags, raw.org.as<u64>(), kept));
  }
  return read_back(e, kept, 8);
}

void sketch_range(Engine& e, const ReadsDev& r, u32 first, u32 last, bool minhash, Sketch& out) {
  if (!minhash) {
    sketch_raw(e, r, first, last, out);
    return;
  }
  sketch_raw(e, r, first, last, e.raw_sketch);
  sketch_minhash(e, r, e.raw_sketch, out);
}

// ---- packing of one-byte codes (rvn_reads_upload_codes: consensus of a polishing round -> targets of the next) ------
namespace {
__global__ void pack_codes_kernel(const u8* __restrict__ codes, const u64* __restrict__ base_off,
                                  const u64* __restrict__ word_off, u32 n_reads, u64 n_words, u64* __restrict__ packed) {
  const u64 wi = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (wi >= n_words) return;
  // read of this word: last i with word_off[i] <= wi
  u32 lo = 0, hi = n_reads;
  while (hi - lo > 1) {
    const u32 mid = lo + (hi - lo) / 2;
    if (word_off[mid] <= wi) lo = mid;
    else hi = mid;
  }
  const u64 first = base_off[lo] + (wi - word_off[lo]) * 32;
  const u64 end = base_off[lo + 1];
  u64 w = 0;
  for (u32 x = 0; x < 32 && first + x < end; ++x) w |= static_cast<u64>(codes[first + x] & 3u) << (2 * x);
  packed[wi] = w;
}
}  // namespace

void pack_codes_on_device(Engine& e, const u8* d_codes, const u64* d_base_off, const u64* d_word_off, u32 n_reads,
                          u64 n_words, u64* d_packed) {
  if (n_words == 0 || n_reads == 0) return;
  pack_codes_kernel<<<div_up(n_words, 256), 256, 0, e.stream>>>(d_codes, d_base_off, d_word_off, n_reads, n_words, d_packed);
  RVN_LAUNCH_CHECK();
}

}  // namespace rvn
