// edit_distance.hip — batched exact global unit-cost edit distance between substrings of the packed reads
// (replaces edlibAlign(lhs, rhs, edlibDefaultAlignConfig()).editDistance, call sites
// RavenLib/src/construct.cc:190-197 (ResolveContainedReads identity filter) and :407-416).
//
// edlib = Myers' bit-vector algorithm with Ukkonen banding; any exact algorithm gives the same number, so
// parity is pinned by the textbook DP (oracle).  MI355X formulation:
//   * the pattern (lhs span) is cut into 64-row blocks; a block's column state is Myers' (Pv, Mv) pair of
//     64-bit vertical-delta vectors, advanced by one text symbol with ~25 integer ops (64 DP cells);
//   * ONE WAVE PER PAIR, lanes = blocks (R blocks per lane), processed systolically: block b works on column
//     j at step t = j + b/R, its horizontal carry (hout in {-1,0,+1}) and bottom score travel to the next
//     lane with one shuffle — no LDS, no global traffic besides the packed symbols;
//   * banding: a path of cost <= k never leaves |i - j| <= k, so only blocks intersecting that band are
//     computed; blocks enter the band with the all-(+1) upper bound (edlib's rule) and retire above it, lanes
//     are reused as a ring (lane = (b/R) mod 64), capacity k <= 32 R (64-1);  the result is exact iff it is
//     <= k, otherwise k is doubled and the pair is redone (in-kernel);
//   * pairs whose distance exceeds the ring capacity fall back to an unbanded striped sweep (same block
//     update, stripe boundaries through a small global array).
// Integer VALU bound (no MFMA, negligible HBM): report cell updates/s.
#include <algorithm>
#include <cstring>

#include "engine.h"
#include "myers.h"
#include "wave.h"

namespace rvn {

namespace {

struct EdPair {
  u32 a_idx, a_begin, a_len;  // pattern (rows): lhs span
  u32 b_idx, b_begin, b_len;  // text (columns): rhs span
  u32 strand;                 // 1: same strand; 0: rhs is reverse-complemented (construct.cc:184-188)
  u32 pad;
};

constexpr u32 kEdOverflow = 0xFFFFFFFFu;

// Banded ring sweep for one pair with threshold k (k >= 64). Returns the banded result (exact iff <= k).
template <int R>
__device__ u32 ed_banded(const u64* __restrict__ a_words, u64 a_base, u32 n, const u64* __restrict__ b_words,
                         u64 b_base, u32 m, bool rc, long long k) {
  const int lane = lane_id();
  const long long nb = (static_cast<long long>(n) + 63) >> 6;
  const long long n_super = (nb + R - 1) / R;
  u64 Pv[R], Mv[R], peq[R][4];
  int score[R];
  long long s = lane;  // current super-block of this lane
  bool fresh = true;   // state for super-block s not initialised yet
  TextCursor tc;
  int hout_last = 1;
  int score_last = 0;
  u32 result = 0;
  const long long t_end = static_cast<long long>(m) + n_super;  // last needed step: m + (n_super - 1)
  for (long long t = 0; t < t_end; ++t) {
    // carry from the previous lane (ring), produced at step t-1 for the same column
    const int src = (lane + 63) & 63;
    const int hin_prev = __shfl(hout_last, src, 64);
    const int score_prev = __shfl(score_last, src, 64);
    // retire finished super-blocks (ring advance)
    while (s < n_super) {
      const long long last_b = s * R + R - 1 < nb ? s * R + R - 1 : nb - 1;
      const long long jout = 64 * last_b + 64 + k;
      if (t - s > (jout < m ? jout : m)) {
        s += 64;
        fresh = true;
      } else {
        break;
      }
    }
    if (s >= n_super) continue;
    const long long j = t - s;
    const long long b0 = s * R;
    const long long jin0 = 64 * b0 - k + 1 < 1 ? 1 : 64 * b0 - k + 1;
    if (j < jin0 || j > m) continue;
    if (fresh) {
#pragma unroll
      for (int r = 0; r < R; ++r) load_peq(a_words, a_base, n, static_cast<u32>(b0 + r), peq[r]);
      tc.init(b_words, b_base, m, rc, j);
      fresh = false;
    }
    const unsigned c = tc.get(j);
    // producer block b0-1 (previous lane): active at column j iff j <= 64 b0 + k
    const bool prod_active = b0 > 0 && j <= 64 * b0 + k;
    int hin = prod_active ? hin_prev : 1;
    int above_prev_col = prod_active ? score_prev - hin_prev : score_prev;  // score of block b-1 at column j-1
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long b = b0 + r;
      if (b >= nb) break;
      const long long jin = 64 * b - k + 1 < 1 ? 1 : 64 * b - k + 1;
      const long long jout = 64 * b + 64 + k;
      if (j < jin) break;  // this and all lower blocks are still below the band
      if (j > jout) {      // retired above the band: the block below sees the +1 boundary
        hin = 1;
        continue;
      }
      if (j == jin) {
        Pv[r] = ~0ULL;
        Mv[r] = 0;
        score[r] = jin == 1 ? static_cast<int>(64 * (b + 1)) : above_prev_col + 64;
      }
      const int old = score[r];
      const u64 eq = c == 0 ? peq[r][0] : (c == 1 ? peq[r][1] : (c == 2 ? peq[r][2] : peq[r][3]));
      const int hout = myers_block(Pv[r], Mv[r], eq, hin);
      score[r] = old + hout;
      above_prev_col = old;
      hin = hout;
      if (b == nb - 1 && j == m) {
        // D[n][m] = bottom score of the last block minus the vertical deltas of the padded rows
        const u32 used = n - static_cast<u32>(64 * b);
        const u64 padmask = used >= 64 ? 0ULL : ~((1ULL << used) - 1ULL);
        result = static_cast<u32>(score[r] - RVN_POPC64(Pv[r] & padmask) + RVN_POPC64(Mv[r] & padmask)) + 1u;
      }
    }
    hout_last = hin;
    score_last = score[R - 1];
  }
  // exactly one lane holds result+1
  result = wave_max(result);
  return result - 1u;
}

constexpr u32 kEdAbove = 0xFFFFFFFEu;  // bounded mode: the distance exceeds the pair's threshold (its exact value is not needed)
// internal to edit_distance_dev: "beyond the WIDEST lane window" (the wave kernel's business; a pair that overflowed a narrower
// window gets a second lane pass with the widest one first).  Never leaves this file.
constexpr u32 kEdOverflowWide = 0xFFFFFFFDu;
constexpr int kEdLaneWidest = 7;  // slots of the widest lane window (threshold 385 for spans of equal length, as the 8 blocks of rounds 1-5)

// One wave per pair.  todo: indices of the pairs to process (null = all).  kmax (nullable): per-pair threshold — a
// caller that only needs to know whether the distance is <= kmax (the identity filters: score >= identity) gets the
// exact distance when it is, kEdAbove otherwise, from ONE sweep at that threshold instead of a doubling sequence.
template <int R>
__global__ __launch_bounds__(256) void ed_banded_kernel(const u64* __restrict__ packed,
                                                       const u64* __restrict__ word_off,
                                                       const EdPair* __restrict__ pairs, const u32* __restrict__ todo,
                                                       u32 n_todo, const u32* __restrict__ kmax,
                                                       u32* __restrict__ out) {
  const u32 q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n_todo) return;
  const u32 p = todo ? todo[q] : q;
  const EdPair pr = pairs[p];
  const u32 n = pr.a_len, m = pr.b_len;
  const long long km = kmax ? static_cast<long long>(kmax[p]) : (1LL << 40);
  u32 res;
  if (n == 0 || m == 0) {
    res = n + m;
    if (static_cast<long long>(res) > km) res = kEdAbove;
  } else {
    const u64* aw = packed + word_off[pr.a_idx];
    const u64* bw = packed + word_off[pr.b_idx];
    const long long cap = 32LL * R * 63;  // ring capacity
    const long long d = n > m ? n - m : m - n;
    long long k = d < 64 ? 64 : d;
    res = kEdOverflow;
    if (d > km) {
      res = kEdAbove;  // the distance is at least the length difference
    } else {
      if (kmax && km <= cap && km > k) k = km;  // bounded: one sweep at the threshold decides
      while (k <= cap) {
        const u32 r = ed_banded<R>(aw, pr.a_begin, n, bw, pr.b_begin, m, pr.strand == 0, k);
        if (static_cast<long long>(r) <= k) {
          res = static_cast<long long>(r) <= km ? r : kEdAbove;
          break;
        }
        if (k >= km) {  // distance > k >= threshold
          res = kEdAbove;
          break;
        }
        if (k == cap) break;
        k = 2 * k < cap ? 2 * k : cap;
      }
    }
  }
  if (lane_id() == 0) out[p] = res;
}

// The text of a lane, 32 columns at a time, fetched at the SAME columns by every lane of the wave (column j with
// (j - 1) % 32 == 0), one group ahead.  TextCursor refills a lane when ITS position crosses a word: with 64 lanes at 64
// phases some lane crosses in nearly every column, the refill branch — and the wait for its load that the compiler has
// to put into it — was executed by the whole wave nearly every column: one memory latency per column (round 6: the
// lane kernel ran 15x below its instruction count).  Here a group is an unaligned 32-base fetch (two words) clipped to
// the pair's span; the reverse strand is turned round on the way in (bit reversal + complement).
struct TextGroups {
  const u64* words;
  long long first;  // base index of column 1 (forward) / of column 1 in the rc direction
  long long lo;     // first base of the span
  u32 m;
  bool rc;
  u64 cur, nxt;     // bases of columns 32 g + 1 .. 32 g + 32 (2 bits each, column order) of the current / next group
  __device__ __forceinline__ u64 fetch(long long j0) const {  // columns j0 .. j0 + 31
    if (j0 > static_cast<long long>(m)) return 0;
    const long long span_last32 = m >= 32 ? lo + m - 32 : lo;  // last start whose 32 bases stay inside the span
    if (!rc) {
      const long long start = first + (j0 - 1);
      const long long s = start < span_last32 ? start : span_last32;
      return load_bases32(words, static_cast<u64>(s)) >> (2 * (start - s));
    }
    const long long top = first - (j0 - 1);  // base of column j0; the columns go DOWN from it
    long long s = top - 31;
    if (s < lo) s = lo;
    u64 y = __brevll(load_bases32(words, static_cast<u64>(s)));  // pair at offset o -> pair 31 - o, its two bits swapped
    y = ((y & 0x5555555555555555ULL) << 1) | ((y >> 1) & 0x5555555555555555ULL);
    return ~(y >> (2 * (31 - (top - s))));  // column j0 + t = offset (top - s) - t -> pair t; complement = 3 - code
  }
  __device__ __forceinline__ void init(const u64* w, u64 b_base, u32 m_, bool rc_) {
    words = w;
    rc = rc_;
    m = m_;
    lo = static_cast<long long>(b_base);
    first = rc_ ? lo + m_ - 1 : lo;
    cur = fetch(1);
    nxt = fetch(33);
  }
  // call once per column, in column order
  __device__ __forceinline__ unsigned get(int j) {
    const int x = (j - 1) & 31;
    if (x == 0 && j > 1) {  // (wave-uniform)
      cur = nxt;
      nxt = fetch(j + 32);
    }
    return static_cast<unsigned>(cur >> (2 * x)) & 3u;
  }
};

// raw words of pattern block b (what load_peq reads), loaded at one column and turned into masks at a later one
struct PeqRaw {
  u64 w[4];
  u32 row0;
};
__device__ __forceinline__ void peq_raw_load(const u64* __restrict__ words, u64 a_base, u32 n, u32 b, PeqRaw& r) {
  r.row0 = b * 64;
  r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
  if (r.row0 < n) {  // exactly the words load_peq touches
    const u64 bit = (a_base + r.row0) * 2;
    const bool off = (bit & 63) != 0, second = r.row0 + 32 < n;
    r.w[0] = words[bit >> 6];
    if (off || second) r.w[1] = words[(bit >> 6) + 1];
    if (off && second) r.w[2] = words[(bit >> 6) + 2];
  }
}
__device__ __forceinline__ void peq_from_raw(const PeqRaw& r, u64 a_base, u32 n, u64 (&peq)[4]) {
  const unsigned off = static_cast<unsigned>(((a_base + r.row0) * 2) & 63);
  u64 lo = r.w[0] >> off, hi = r.w[1] >> off;
  if (off) {
    lo |= r.w[1] << (64 - off);
    hi |= r.w[2] << (64 - off);
  }
  if (!(r.row0 < n)) lo = 0;
  if (!(r.row0 + 32 < n)) hi = 0;
  const u32 valid = n > r.row0 ? (n - r.row0 >= 64 ? 64u : n - r.row0) : 0u;
  const u64 vmask = valid >= 64 ? ~0ULL : ((1ULL << valid) - 1ULL);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const u64 rep = 0x5555555555555555ULL * static_cast<u64>(c);
    const u64 ml = lo ^ rep, mh = hi ^ rep;
    const u64 el = compress_even(~(ml | (ml >> 1)));
    const u64 eh = compress_even(~(mh | (mh >> 1)));
    peq[c] = (el | (eh << 32)) & vmask;
  }
}

// One LANE per pair: banded Myers over a window of W consecutive 64-row blocks that moves down the diagonal, all state
// in registers.  For pairs whose band is a few blocks wide (HiFi-like spans: distance ~1 % of the length) the
// wave-per-pair systolic kernel above keeps 1 of 64 lanes busy; here 64 pairs share a wave.  Result exact if <= the
// threshold, else kEdAbove (kmax reached) or an overflow code (kEdOverflow: a wider window may do; kEdOverflowWide from
// the widest window: the pair needs the wave kernel).  `order`: pairs sorted by text length, so that the lanes of a wave
// run loops of similar length.
// Round 6: NOTHING in the column loop happens at a lane's own phase any more.  Rounds 1-5 let every lane slide its
// window, take in a block and refill its text word when ITS band got there — with 64 lanes at 64 phases each of these
// happened in nearly every column for somebody, so the whole wave ran the slide (13 register moves per slot), the entry
// (13 selects per slot), and waited for a load issued one column earlier, nearly every column.  Now the window of every
// lane moves by one block at the columns 64 c + 1 (it is one block taller than a band needs at any single column, so
// that it holds the band of a whole 64-column cycle: a band of lo rows above and hi rows below the diagonal needs
// W >= ceil(lo / 64) + ceil(hi / 64) + 1 slots — FEWER than the per-lane sliding of rounds 1-5, whose accounting kept two
// spare blocks), the masks of the block that comes in are loaded two cycles ahead and converted one cycle ahead (PeqRaw),
// the text comes in groups of 32 columns (TextGroups); a block's entry into and exit from the BAND (edlib's rule: the
// all-(+1) bound below the block above; +1 boundary under a block that left) stay per lane and per column, as
// predicates of the slot loop: the band, and with it every result and every overflow decision, is the one of rounds 1-5
// for the same threshold.
template <int W>
__device__ __forceinline__ u32 ed_lane_threshold(u32 n, u32 m, u32 km, bool* fits) {
  const u32 d = n > m ? n - m : m - n;
  // lo = s + (m > n ? d : 0), hi = s + (n > m ? d : 0); the largest s with ceil(lo / 64) + ceil(hi / 64) <= W - 1:
  // a units for the side without d, the rest for the side with it
  int s_max = -1;
#pragma unroll
  for (int a = 0; a <= W - 1; ++a) {
    const int with_d = 64 * (W - 1 - a) - static_cast<int>(d);
    const int s = 64 * a < with_d ? 64 * a : with_d;
    s_max = s > s_max ? s : s_max;
  }
  *fits = s_max >= 0;
  const u32 k = d + 2u * static_cast<u32>(s_max > 0 ? s_max : 0) + 1u;
  return k > km ? km : k;
}

template <int W>
__global__ __launch_bounds__(64) void ed_lane_kernel(const u64* __restrict__ packed, const u64* __restrict__ word_off,
                                                    const EdPair* __restrict__ pairs, const u32* __restrict__ order,
                                                    u32 n_order, const u32* __restrict__ kmax, u32* __restrict__ out) {
  const u32 q = blockIdx.x * 64 + threadIdx.x;
  if (q >= n_order) return;
  const u32 p = order[q];
  const EdPair pr = pairs[p];
  const u32 n = pr.a_len, m = pr.b_len;
  const u32 km = kmax ? kmax[p] : 0xFFFFFFF0u;
  if (n == 0 || m == 0) {
    out[p] = (n + m) > km ? kEdAbove : n + m;
    return;
  }
  const u32 d = n > m ? n - m : m - n;
  if (d > km) {
    out[p] = kEdAbove;
    return;
  }
  constexpr u32 kOverflow = W >= kEdLaneWidest ? kEdOverflowWide : kEdOverflow;
  bool fits;
  const u32 k = ed_lane_threshold<W>(n, m, km, &fits);
  if (!fits) {
    out[p] = kOverflow;
    return;
  }
  const int lo = static_cast<int>((k - d) / 2 + (m > n ? d : 0u));  // band rows above the diagonal: block t is in the band
  const int hi = static_cast<int>((k - d) / 2 + (n > m ? d : 0u));  // of column j while 64 t - hi + 1 <= j <= 64 t + 64 + lo
  const int nb = static_cast<int>((n + 63) >> 6);
  const int sL = (lo + 63) >> 6;  // the window's first block in cycle c (columns 64 c + 1 .. 64 c + 64): max(0, c - sL)
  const u64* aw = packed + word_off[pr.a_idx];
  const u64* bw = packed + word_off[pr.b_idx];
  u64 Pv[W], Mv[W], peq[W][4];
  int sc[W];
  int wb = 0;  // block held by slot 0
#pragma unroll
  for (int i = 0; i < W; ++i) {
    Pv[i] = ~0ULL;
    Mv[i] = 0;
    sc[i] = 0;
    load_peq(aw, pr.a_begin, n, static_cast<u32>(i), peq[i]);  // (blocks beyond the pattern: all-zero masks, never in the band)
  }
  u64 staged[4];  // match masks of block wb + W (the next to come in)
  load_peq(aw, pr.a_begin, n, static_cast<u32>(W), staged);
  PeqRaw raw;     // words of block wb + W + 1 (the one after)
  peq_raw_load(aw, pr.a_begin, n, static_cast<u32>(W + 1), raw);
  TextGroups tg;
  tg.init(bw, pr.b_begin, m, pr.strand == 0);
  u32 result = 0xFFFFFFFFu;
  for (int j = 1; j <= static_cast<int>(m); ++j) {
    if (((j - 1) & 63) == 0) {  // wave-uniform: the window moves on
      const int c = (j - 1) >> 6;
      if (c - sL > wb) {
#pragma unroll
        for (int i = 0; i + 1 < W; ++i) {
          Pv[i] = Pv[i + 1];
          Mv[i] = Mv[i + 1];
          sc[i] = sc[i + 1];
#pragma unroll
          for (int x = 0; x < 4; ++x) peq[i][x] = peq[i + 1][x];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) peq[W - 1][x] = staged[x];
        ++wb;
        peq_from_raw(raw, pr.a_begin, n, staged);
        peq_raw_load(aw, pr.a_begin, n, static_cast<u32>(wb + W + 1), raw);
      }
    }
    const unsigned c = tg.get(j);
    const bool c0 = c == 0, c1 = c == 1, c2 = c == 2;
    const int u = j - 64 * wb + hi - 1;   // slot i has entered the band iff u >= 64 i, enters now iff u == 64 i (or j == 1)
    const int v = j - 64 * wb - lo - 64;  // slot i has left the band iff v > 64 i
    int hin = 1;        // above the first band block: the matrix border or a block that left the band (+1 boundary)
    int above_old = 0;  // score of the block above at column j - 1
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const bool in_band = wb + i < nb && u >= 64 * i && v <= 64 * i;
      if (in_band) {
        if (u == 64 * i || j == 1) {  // first band column of the block
          Pv[i] = ~0ULL;
          Mv[i] = 0;
          // band starts at column 1: column 0 holds D(i, 0) = i; later: edlib's all-(+1) upper bound below the block above
          sc[i] = (64 * (wb + i) + 1 - hi <= 1) ? 64 * (wb + i + 1) : above_old + 64;
        }
        const int old = sc[i];
        const u64 eq = c0 ? peq[i][0] : (c1 ? peq[i][1] : (c2 ? peq[i][2] : peq[i][3]));
        const int hout = myers_block(Pv[i], Mv[i], eq, hin);
        sc[i] = old + hout;
        hin = hout;
        above_old = old;
      } else {
        hin = 1;  // (a block that left: the +1 boundary for the one below; a block not yet in: nothing below it is)
      }
    }
    if (j == static_cast<int>(m)) {
      const int slot = nb - 1 - wb;
#pragma unroll
      for (int i = 0; i < W; ++i) {
        if (i == slot) {
          const u32 used = n - static_cast<u32>(64 * (nb - 1));
          const u64 padmask = used >= 64 ? 0ULL : ~((1ULL << used) - 1ULL);
          result = static_cast<u32>(sc[i] - RVN_POPC64(Pv[i] & padmask) + RVN_POPC64(Mv[i] & padmask));
        }
      }
    }
  }
  u32 res;
  if (result <= k) res = result;
  else if (k >= km) res = kEdAbove;
  else res = kOverflow;
  out[p] = res;
}

__global__ void ed_keys_kernel(const EdPair* __restrict__ pairs, u32 n, u32* __restrict__ keys, u32* __restrict__ vals) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = 0xFFFFFFFFu - pairs[i].b_len;  // longest text first
  vals[i] = i;
}
// wide: also the pairs beyond the widest lane window (what the wave kernel takes); otherwise only those a narrower window lost
__global__ void ed_collect_overflow_kernel(const u32* __restrict__ out, u32 n, bool wide, u32* __restrict__ todo, u32* __restrict__ cnt) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (out[i] == kEdOverflow || (wide && out[i] == kEdOverflowWide))) todo[atomicAdd(cnt, 1u)] = i;
}
// The sample's verdict: cnt[1] = pairs the widest lane window decided, cnt[16 + b] = histogram of distance / length in steps
// of 1 / 1024 over the pairs whose distance is known (b = 63: 6.2 % and more).
constexpr u32 kEdHistBins = 64, kEdHistAt = 16, kEdBoundsAt = 2;
__global__ void ed_sample_kernel(const EdPair* __restrict__ pairs, const u32* __restrict__ out, const u32* __restrict__ sample, u32 n,
                                 u32* __restrict__ cnt) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 p = sample[i];
  const u32 o = out[p];
  if (o != kEdOverflowWide) atomicAdd(cnt + 1, 1u);
  if (o < kEdOverflowWide) {
    const u32 len = max(max(pairs[p].a_len, pairs[p].b_len), 1u);
    atomicAdd(cnt + kEdHistAt + min<u32>(kEdHistBins - 1, static_cast<u32>((static_cast<u64>(o) << 10) / len)), 1u);
  }
}
// Window classes of the lane kernel.  The pairs are sorted by text length (longest first), the distance a pair is expected
// to have is rate x length with the rate of the sample's 90th percentile + 10 % + 16: a class is a contiguous piece of the
// order.  cnt[kEdBoundsAt + c] = first position whose pair fits the window of class c (windows kEdClassB, narrowest first);
// a wrong guess costs a second pass with the widest window, never a result.
constexpr int kEdClasses = 2;                        // narrower windows than the widest one
__constant__ int kEdClassB[kEdClasses] = {3, 5};     // (an even number of slots adds nothing for spans of equal length)
__global__ void ed_classes_kernel(const u32* __restrict__ sorted_keys, u32 n_main, u32* __restrict__ cnt) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u32 total = 0;
  for (u32 b = 0; b < kEdHistBins; ++b) total += cnt[kEdHistAt + b];
  u32 bin = kEdHistBins - 1;
  if (total) {
    u32 run = 0;
    for (u32 b = 0; b < kEdHistBins; ++b) {
      run += cnt[kEdHistAt + b];
      if (10ULL * run >= 9ULL * total) {
        bin = b;
        break;
      }
    }
  }
  for (int c = 0; c < kEdClasses; ++c) {
    const u32 room = 64u * static_cast<u32>(kEdClassB[c] - 1);  // (threshold of the window for spans of equal length, - 1)
    // rate = (bin + 1) / 1024, need = 1.1 rate len + 16 <= room  <=>  len <= (room - 16) * 1024 / (1.1 (bin + 1))
    const u32 len_max = total ? static_cast<u32>((static_cast<u64>(room - 16) * 10240ULL) / (11ULL * (bin + 1))) : 0u;
    const u32 key_min = 0xFFFFFFFFu - len_max;  // keys = 0xFFFFFFFF - text length, ascending
    u32 lo = 0, hi = n_main;
    while (lo < hi) {
      const u32 mid = lo + (hi - lo) / 2;
      if (sorted_keys[mid] < key_min) lo = mid + 1;
      else hi = mid;
    }
    cnt[kEdBoundsAt + c] = lo;
  }
}

// Unbanded striped sweep (fallback; any distance): stripes of 64 blocks, lane = block, boundary hout per column
// through `hb` (two int8 rows of m+1 entries per pair).
__global__ __launch_bounds__(64) void ed_full_kernel(const u64* __restrict__ packed,
                                                    const u64* __restrict__ word_off,
                                                    const EdPair* __restrict__ pairs,
                                                    const u32* __restrict__ todo, u32 n_todo,
                                                    const u64* __restrict__ hb_off, signed char* __restrict__ hb,
                                                    u32* __restrict__ out) {
  const u32 q = blockIdx.x;
  if (q >= n_todo) return;
  const u32 p = todo[q];
  const EdPair pr = pairs[p];
  const u32 n = pr.a_len, m = pr.b_len;
  const u64* aw = packed + word_off[pr.a_idx];
  const u64* bw = packed + word_off[pr.b_idx];
  const int lane = lane_id();
  const long long nb = (static_cast<long long>(n) + 63) >> 6;
  signed char* hb0 = hb + hb_off[q];
  signed char* hb1 = hb0 + (m + 1);
  u32 result = 0;
  for (long long stripe = 0; stripe * 64 < nb; ++stripe) {
    const long long b = stripe * 64 + lane;
    const signed char* hin_row = (stripe & 1) ? hb1 : hb0;
    signed char* hout_row = (stripe & 1) ? hb0 : hb1;
    const long long last_lane = (nb - stripe * 64 < 64 ? nb - stripe * 64 : 64) - 1;
    u64 Pv = ~0ULL, Mv = 0, peq[4];
    int score = static_cast<int>(64 * (b + 1));
    if (b < nb) load_peq(aw, pr.a_begin, n, static_cast<u32>(b), peq);
    TextCursor tc;
    bool started = false;
    int hout_last = 1;
    for (long long t = 0; t < static_cast<long long>(m) + 64; ++t) {
      const int hin_prev = __shfl_up(hout_last, 1, 64);
      const long long j = t - lane + 1;  // lane l handles column t - l + 1
      if (b >= nb || j < 1 || j > m) continue;
      if (!started) {
        tc.init(bw, pr.b_begin, m, pr.strand == 0, j);
        started = true;
      }
      const unsigned c = tc.get(j);
      int hin = lane == 0 ? (stripe == 0 ? 1 : static_cast<int>(hin_row[j])) : hin_prev;
      const u64 eq = c == 0 ? peq[0] : (c == 1 ? peq[1] : (c == 2 ? peq[2] : peq[3]));
      const int hout = myers_block(Pv, Mv, eq, hin);
      score += hout;
      hout_last = hout;
      if (lane == last_lane) hout_row[j] = static_cast<signed char>(hout);
      if (b == nb - 1 && j == m) {
        const u32 used = n - static_cast<u32>(64 * b);
        const u64 padmask = used >= 64 ? 0ULL : ~((1ULL << used) - 1ULL);
        result = static_cast<u32>(score - RVN_POPC64(Pv & padmask) + RVN_POPC64(Mv & padmask)) + 1u;
      }
    }
    __threadfence_block();  // the stripe's boundary row must be visible to lane 0 of the next stripe
  }
  result = wave_max(result);
  if (lane == 0) out[p] = result - 1u;
}

}  // namespace

// pairs and results resident in HBM: d_pairs = n_pairs x {a_idx,a_begin,a_len,b_idx,b_begin,b_len,strand,0}, d_out u32[n].
// d_kmax (nullable): per-pair thresholds — distances above them come back as 0xFFFFFFFE (see ed_banded_kernel).
void edit_distance_dev(Engine& e, const ReadsDev& r, const u32* d_pairs_raw, u32 n_pairs, u32* d_out, const u32* d_kmax) {
  if (n_pairs == 0) return;
  hipStream_t s = e.stream;
  const EdPair* d_pairs = reinterpret_cast<const EdPair*>(d_pairs_raw);
  u32* d_cnt = e.ed_cnt.get<u32>(kEdHistAt + kEdHistBins);
  RVN_HIP(hipMemsetAsync(d_cnt, 0, (kEdHistAt + kEdHistBins) * 4, s));
  RVN_HIP(hipMemsetAsync(d_out, 0xFF, static_cast<size_t>(n_pairs) * 4, s));  // everything starts as "needs the wave kernel"
  // ---- stage 1: one lane per pair for narrow bands.  Tried on a sample first: it pays only when most pairs are
  // closer than ~384 edits (HiFi-like); for ONT-like spans nearly every pair would come back as overflow.
  u32* d_sk = e.ed_sort.get<u32>(4 * static_cast<size_t>(n_pairs) + 8);
  u32* d_sk1 = d_sk + n_pairs + 1;
  u32* d_sv = d_sk1 + n_pairs + 1;
  u32* d_sv1 = d_sv + n_pairs + 1;
  ed_keys_kernel<<<div_up(n_pairs, 256), 256, 0, s>>>(d_pairs, n_pairs, d_sk, d_sv);
  RVN_LAUNCH_CHECK();
  const int which = radix_sort_pairs_u32_u32(d_sk, d_sk1, d_sv, d_sv1, n_pairs, 32, e.sort_tmp, e.scan_tmp, s, kKPileSortUp,
                                             kKPileSortDown);
  const u32* d_order = which ? d_sv1 : d_sv;
  const u32* d_sorted_keys = which ? d_sk1 : d_sk;
  const u32 n_sample = std::min<u32>(n_pairs, 2048);
  const u32 n_main = n_pairs - n_sample;
  // the sample: every (n_pairs / n_sample)-th pair of the sorted order would need a gather; the shortest pairs (the
  // tail of the order) are the cheapest probe and representative of the error level
  const u32* d_sample = d_order + n_main;
  RVN_KLAUNCH(kKEditLane, ed_lane_kernel<kEdLaneWidest><<<div_up(n_sample, 64), 64, 0, s>>>(
                              r.packed.as<u64>(), r.word_off.as<u64>(), d_pairs, d_sample, n_sample, d_kmax, d_out));
  ed_sample_kernel<<<div_up(n_sample, 256), 256, 0, s>>>(d_pairs, d_out, d_sample, n_sample, d_cnt);
  RVN_LAUNCH_CHECK();
  ed_classes_kernel<<<1, 64, 0, s>>>(d_sorted_keys, n_main, d_cnt);
  RVN_LAUNCH_CHECK();
  read_back(e, d_cnt, (kEdBoundsAt + kEdClasses) * 4);
  u32 h_cnt[kEdBoundsAt + kEdClasses];
  std::memcpy(h_cnt, e.h_pin, sizeof(h_cnt));
  const u32 done = h_cnt[1];
  bool narrow_used = false;
  if (2 * done >= n_sample && n_main) {
    // a window as wide as the pair is expected to need (the work of a lane is proportional to the slots of its window: with
    // the widest one for everybody the HiFi identity filter computed 8 blocks per column where 3 to 5 hold the band)
    u32 from = 0;  // positions [from, to) of the order take the window of B blocks; widest (longest pairs) first
    auto piece = [&](u32 to, int B) {
      to = std::min(std::max(to, from), n_main);
      const u32 cn = to - from;
      const u32* ord = d_order + from;
      from = to;
      if (!cn) return;
#define RVN_ED_LANE(B_)                                                                                        \
  case B_:                                                                                                      \
    RVN_KLAUNCH(kKEditLane, ed_lane_kernel<B_><<<div_up(cn, 64), 64, 0, s>>>(r.packed.as<u64>(), r.word_off.as<u64>(), \
                                                                           d_pairs, ord, cn, d_kmax, d_out));   \
    break
      switch (B) {
        RVN_ED_LANE(3);
        RVN_ED_LANE(5);
        default:
          RVN_KLAUNCH(kKEditLane, ed_lane_kernel<kEdLaneWidest><<<div_up(cn, 64), 64, 0, s>>>(
                                      r.packed.as<u64>(), r.word_off.as<u64>(), d_pairs, ord, cn, d_kmax, d_out));
      }
#undef RVN_ED_LANE
      if (B < kEdLaneWidest) narrow_used = true;
    };
    static const int class_b[kEdClasses] = {3, 5};  // (= kEdClassB on the device)
    piece(h_cnt[kEdBoundsAt + kEdClasses - 1], kEdLaneWidest);
    for (int c = kEdClasses - 1; c >= 1; --c) piece(h_cnt[kEdBoundsAt + c - 1], class_b[c]);
    piece(n_main, class_b[0]);
  }
  u32* d_todo = e.ed_todo.get<u32>(static_cast<size_t>(n_pairs) + 1);
  if (narrow_used) {  // second chance with the widest window for the pairs a narrower one lost
    RVN_HIP(hipMemsetAsync(d_cnt, 0, 4, s));
    ed_collect_overflow_kernel<<<div_up(n_pairs, 256), 256, 0, s>>>(d_out, n_pairs, false, d_todo, d_cnt);
    RVN_LAUNCH_CHECK();
    const u32 n_again = static_cast<u32>(read_back(e, d_cnt, 4));
    if (n_again)
      RVN_KLAUNCH(kKEditLane, ed_lane_kernel<kEdLaneWidest><<<div_up(n_again, 64), 64, 0, s>>>(
                                  r.packed.as<u64>(), r.word_off.as<u64>(), d_pairs, d_todo, n_again, d_kmax, d_out));
  }
  // ---- stage 2: the wave-per-pair kernel for what is left ----
  RVN_HIP(hipMemsetAsync(d_cnt, 0, 4, s));
  ed_collect_overflow_kernel<<<div_up(n_pairs, 256), 256, 0, s>>>(d_out, n_pairs, true, d_todo, d_cnt);
  RVN_LAUNCH_CHECK();
  const u32 n_todo = static_cast<u32>(read_back(e, d_cnt, 4));
  if (n_todo == 0) return;
  RVN_KLAUNCH(kKEditBanded, ed_banded_kernel<4><<<div_up(n_todo, 4), 256, 0, s>>>(
                                r.packed.as<u64>(), r.word_off.as<u64>(), d_pairs, d_todo, n_todo, d_kmax, d_out));
  // ---- stage 3: pairs beyond the ring capacity (rare): unbanded striped sweep, exact ----
  RVN_HIP(hipMemsetAsync(d_cnt, 0, 4, s));
  ed_collect_overflow_kernel<<<div_up(n_pairs, 256), 256, 0, s>>>(d_out, n_pairs, true, d_todo, d_cnt);
  RVN_LAUNCH_CHECK();
  const u32 n_full = static_cast<u32>(read_back(e, d_cnt, 4));
  if (n_full == 0) return;
  std::vector<u32> todo(n_full);
  std::vector<EdPair> hp(n_pairs);
  RVN_HIP(hipMemcpyAsync(todo.data(), d_todo, static_cast<size_t>(n_full) * 4, hipMemcpyDeviceToHost, s));
  RVN_HIP(hipMemcpyAsync(hp.data(), d_pairs, static_cast<size_t>(n_pairs) * sizeof(EdPair), hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  std::sort(todo.begin(), todo.end());
  std::vector<u64> hb_off;
  u64 hb_total = 0;
  for (u32 i : todo) {
    hb_off.push_back(hb_total);
    hb_total += 2ULL * (static_cast<u64>(hp[i].b_len) + 1);
  }
  u64* d_off = e.tmp_d.get<u64>(hb_off.size() + 1);
  signed char* d_hb = e.tmp_f.get<signed char>(hb_total + 16);
  RVN_HIP(hipMemcpyAsync(d_todo, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, s));
  RVN_HIP(hipMemcpyAsync(d_off, hb_off.data(), hb_off.size() * 8, hipMemcpyHostToDevice, s));
  RVN_KLAUNCH(kKEditFull, ed_full_kernel<<<n_full, 64, 0, s>>>(r.packed.as<u64>(), r.word_off.as<u64>(), d_pairs, d_todo, n_full,
                                                               d_off, d_hb, d_out));
  RVN_HIP(rvn_stream_sync(s));  // the host lists above are locals
}

// pairs: host array of n_pairs x 8 u32 {a_idx,a_begin,a_len,b_idx,b_begin,b_len,strand,0}; out: host u32[n_pairs]
void edit_distance_batch(Engine& e, const ReadsDev& r, const u32* h_pairs, u32 n_pairs, u32* h_out,
                         double* kernel_ms, u64* cells) {
  if (n_pairs == 0) return;
  hipStream_t s = e.stream;
  EdPair* d_pairs = e.tmp_a.get<EdPair>(static_cast<size_t>(n_pairs) + 1);
  u32* d_out = e.tmp_b.get<u32>(static_cast<size_t>(n_pairs) + 1);
  RVN_HIP(hipMemcpyAsync(d_pairs, h_pairs, static_cast<size_t>(n_pairs) * sizeof(EdPair), hipMemcpyHostToDevice, s));
  RVN_HIP(hipEventRecord(e.ev0, s));
  edit_distance_dev(e, r, reinterpret_cast<const u32*>(d_pairs), n_pairs, d_out, nullptr);
  RVN_HIP(hipEventRecord(e.ev1, s));
  RVN_HIP(hipMemcpyAsync(h_out, d_out, static_cast<size_t>(n_pairs) * 4, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  RVN_HIP(hipEventSynchronize(e.ev1));
  if (kernel_ms) {
    float ms = 0;
    RVN_HIP(hipEventElapsedTime(&ms, e.ev0, e.ev1));
    *kernel_ms = ms;
  }
  if (cells) {
    const EdPair* hp = reinterpret_cast<const EdPair*>(h_pairs);
    u64 c = 0;
    for (u32 i = 0; i < n_pairs; ++i) c += static_cast<u64>(hp[i].a_len) * hp[i].b_len;
    *cells = c;
  }
}

}  // namespace rvn
