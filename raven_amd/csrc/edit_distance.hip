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

template <int R>
__global__ __launch_bounds__(256) void ed_banded_kernel(const u64* __restrict__ packed,
                                                       const u64* __restrict__ word_off,
                                                       const EdPair* __restrict__ pairs, u32 n_pairs,
                                                       u32* __restrict__ out) {
  const u32 p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_pairs) return;
  const EdPair pr = pairs[p];
  const u32 n = pr.a_len, m = pr.b_len;
  u32 res;
  if (n == 0 || m == 0) {
    res = n + m;
  } else {
    const u64* aw = packed + word_off[pr.a_idx];
    const u64* bw = packed + word_off[pr.b_idx];
    const long long cap = 32LL * R * 63;  // ring capacity
    long long k = n > m ? n - m : m - n;
    if (k < 64) k = 64;
    res = kEdOverflow;
    while (k <= cap) {
      const u32 r = ed_banded<R>(aw, pr.a_begin, n, bw, pr.b_begin, m, pr.strand == 0, k);
      if (static_cast<long long>(r) <= k) {
        res = r;
        break;
      }
      if (k == cap) break;
      k = 2 * k < cap ? 2 * k : cap;
    }
  }
  if (lane_id() == 0) out[p] = res;
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

__global__ void ed_count_overflow_kernel(const u32* __restrict__ out, u32 n, u32* __restrict__ cnt) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && out[i] == kEdOverflow) atomicAdd(cnt, 1u);
}

}  // namespace

// pairs and results resident in HBM: d_pairs = n_pairs x {a_idx,a_begin,a_len,b_idx,b_begin,b_len,strand,0}, d_out u32[n]
void edit_distance_dev(Engine& e, const ReadsDev& r, const u32* d_pairs_raw, u32 n_pairs, u32* d_out) {
  if (n_pairs == 0) return;
  hipStream_t s = e.stream;
  const EdPair* d_pairs = reinterpret_cast<const EdPair*>(d_pairs_raw);
  RVN_KLAUNCH(kKEditBanded, ed_banded_kernel<4><<<div_up(n_pairs, 4), 256, 0, s>>>(
                                r.packed.as<u64>(), r.word_off.as<u64>(), d_pairs, n_pairs, d_out));
  // pairs beyond the ring capacity (rare): unbanded striped sweep
  u32* d_cnt = e.ed_cnt.get<u32>(4);
  RVN_HIP(hipMemsetAsync(d_cnt, 0, 4, s));
  ed_count_overflow_kernel<<<div_up(n_pairs, 256), 256, 0, s>>>(d_out, n_pairs, d_cnt);
  RVN_LAUNCH_CHECK();
  if (read_back(e, d_cnt, 4) == 0) return;
  std::vector<u32> h_out(n_pairs);
  std::vector<EdPair> hp(n_pairs);
  RVN_HIP(hipMemcpyAsync(h_out.data(), d_out, static_cast<size_t>(n_pairs) * 4, hipMemcpyDeviceToHost, s));
  RVN_HIP(hipMemcpyAsync(hp.data(), d_pairs, static_cast<size_t>(n_pairs) * sizeof(EdPair), hipMemcpyDeviceToHost, s));
  RVN_HIP(hipStreamSynchronize(s));
  std::vector<u32> todo;
  std::vector<u64> hb_off;
  u64 hb_total = 0;
  for (u32 i = 0; i < n_pairs; ++i) {
    if (h_out[i] == kEdOverflow) {
      todo.push_back(i);
      hb_off.push_back(hb_total);
      hb_total += 2ULL * (static_cast<u64>(hp[i].b_len) + 1);
    }
  }
  u32* d_todo = e.tmp_c.get<u32>(todo.size() + 1);
  u64* d_off = e.tmp_d.get<u64>(hb_off.size() + 1);
  signed char* d_hb = e.tmp_f.get<signed char>(hb_total + 16);
  RVN_HIP(hipMemcpyAsync(d_todo, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, s));
  RVN_HIP(hipMemcpyAsync(d_off, hb_off.data(), hb_off.size() * 8, hipMemcpyHostToDevice, s));
  RVN_KLAUNCH(kKEditFull, ed_full_kernel<<<static_cast<u32>(todo.size()), 64, 0, s>>>(
                              r.packed.as<u64>(), r.word_off.as<u64>(), d_pairs, d_todo,
                              static_cast<u32>(todo.size()), d_off, d_hb, d_out));
  RVN_HIP(hipStreamSynchronize(s));  // the host lists above are locals
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
  edit_distance_dev(e, r, reinterpret_cast<const u32*>(d_pairs), n_pairs, d_out);
  RVN_HIP(hipEventRecord(e.ev1, s));
  RVN_HIP(hipMemcpyAsync(h_out, d_out, static_cast<size_t>(n_pairs) * 4, hipMemcpyDeviceToHost, s));
  RVN_HIP(hipStreamSynchronize(s));
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
