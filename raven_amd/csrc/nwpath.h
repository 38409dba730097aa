// nwpath.h — the global alignment PATH of a read against its target span and the window breakpoints racon derives
// from it (racon Overlap::find_breaking_points: edlibAlign(query, target, EDLIB_MODE_NW, EDLIB_TASK_PATH) -> CIGAR ->
// for every window of w target bases the first and the last aligned ('M') pair; call site of the whole round:
// RavenLib/src/polish.cc:43-51).  The polishing front end (polish.hip) runs this for the best overlap of every read.
//
// Formulation for the device (kernels in nwpath.hip; everything here is __host__ __device__ so that the very same
// code can be stepped lane by lane on the CPU, rvn_test_nw_breakpoints):
//   forward   Myers' bit-vector blocks over a DIAGONAL band that contains every alignment of cost <= k
//             (offsets row - column in [-lo, hi], lo + hi = k - |n - m| + |n - m|): one wave per alignment, lanes =
//             64-row blocks (R per lane) reused as a ring, systolic over the columns exactly like the distance
//             kernel (edit_distance.hip) — but every block update also stores its (Pv, Mv) vertical-delta words and
//             its bottom score, time-major ([step][lane][r]) so that the 64 lanes of a step write one contiguous run.
//             The result is exact iff it is <= k (edlib's criterion); otherwise the job is redone with 2k.
//   traceback one thread per alignment walks from (n, m) to (0, 0).  Equal bases always take the diagonal
//             (D(i,j) = D(i-1,j-1) when the bases match, so no score is needed); at a mismatch the three neighbour
//             scores come from the stored words (block bottom score -/+ popcounts of the deltas above the row).
//             Preference on ties: diagonal, then query-base-only ('I'), then target-base-only ('D') — the rule of the
//             CPU restatement (oracle NwPath); edlib documents no tie rule for paths found by its Hirschberg split,
//             any optimal path is "the" edlib path.  Banded values equal the full-matrix values on every cell the
//             walk can take (each lies on an optimal alignment, which the band contains, and a too-large banded value
//             of a cell that is NOT a valid predecessor stays invalid), so the path is the full-matrix path.
//             The walk emits no CIGAR: it folds find_breaking_points in and writes, per window, the first / last
//             aligned pair and the read offsets at eight fixed target positions (the POA band guide).
#pragma once

#include "myers.h"

namespace rvn {

// rows = target span (pattern, forward strand), columns = read span in the target's orientation (text)
struct NwJob {
  u64 t_word;   // first word of the target in the packed target set
  u64 r_word;   // first word of the read in the packed read set
  u64 store;    // first slot of this job in the (Pv, Mv) / score arrays
  u64 bp_off;   // first window record of this job
  u32 t_begin, n;  // target span [t_begin, t_begin + n)
  u32 q_begin, m;  // read span [q_begin, q_begin + m) in the orientation of the target
  u32 r_len;       // length of the read
  u32 rc;          // 1: the read is reverse-complemented (overlap on the opposite strand)
  u32 k;           // cost threshold of this attempt (>= |n - m|)
  u32 lo, hi;      // band: -lo <= row - column <= hi
  u32 L;           // ring lanes in use (<= 64)
  u32 R;           // blocks per lane
  u32 read, target;  // indices in their sets
  u32 n_windows;     // windows touched by the target span
  u32 pad_;
};
static_assert(sizeof(NwJob) == 96, "NwJob layout");

struct NwWindowRec {  // per (job, window): racon's breakpoint pair + band guide
  u32 first_t, first_q;  // first aligned pair of the window (target / oriented read position); first_t == ~0: none
  u32 last_t, last_q;    // one past the last aligned pair
  u16 grid[8];           // read offset (relative to first_q, clamped) where the path crosses target position
                         // window_start + (x * w) / 8; 0xFFFF = the alignment does not cover it
};
static_assert(sizeof(NwWindowRec) == 32, "NwWindowRec layout");

constexpr u32 kNwInf = 0x3FFFFFFFu;

__host__ __device__ inline u32 nw_band_lo(u32 n, u32 m, u32 k) {  // most negative offset: (k - |d|) / 2 (+ |d| if m > n)
  const u32 d = n > m ? n - m : m - n;
  return (k - d) / 2 + (m > n ? d : 0u);
}
__host__ __device__ inline u32 nw_band_hi(u32 n, u32 m, u32 k) {
  const u32 d = n > m ? n - m : m - n;
  return (k - d) / 2 + (n > m ? d : 0u);
}
// ring lanes needed so that a lane's next super-block never has to start before its current one retired
__host__ __device__ inline u32 nw_ring_lanes(u32 lo, u32 hi, u32 R) {
  const u64 num = 64ULL * R + lo + hi;
  const u64 den = 64ULL * R + 1;
  return static_cast<u32>((num + den - 1) / den);
}
__host__ __device__ inline u64 nw_store_slots(u32 n, u32 m, u32 L, u32 R) {
  const u64 nb = (static_cast<u64>(n) + 63) >> 6;
  const u64 n_super = (nb + R - 1) / R;
  return (static_cast<u64>(m) + n_super + 1) * L * R;
}
// first / last column at which block b is inside the band
__host__ __device__ inline long long nw_jin(long long b, long long hi) {
  const long long j = 64 * b + 1 - hi;
  return j < 1 ? 1 : j;
}
__host__ __device__ inline long long nw_jout(long long b, long long lo) { return 64 * b + 64 + lo; }
__host__ __device__ inline u64 nw_slot(long long b, long long j, u32 L, u32 R) {
  const long long s = b / R;
  return (static_cast<u64>(j + s) * L + static_cast<u64>(s % L)) * R + static_cast<u64>(b % R);
}

struct NwPm {
  u64 pv, mv;
};

// One lane of the forward sweep.  The kernel (and the CPU stepper) calls step(t, ...) for t = 0 .. m + n_super with
// the producer lane's (hout_last, score_last) of the previous step.
template <int R>
struct NwLane {
  const u64* a_words;
  const u64* b_words;
  u64 a_base, b_base;
  u32 n, m;
  bool rc;
  long long lo, hi, nb, n_super;
  int L, lane;
  NwPm* st_pm;
  int* st_sc;
  u64 Pv[R], Mv[R], peq[R][4];
  int score[R];
  long long s;
  bool fresh;
  TextCursor tc;
  int hout_last, score_last;
  u32 result;  // D(n, m) + 1 on the one lane that computes it

  __host__ __device__ void init(const NwJob& J, const u64* t_words, const u64* r_words, NwPm* pm, int* sc, int lane_) {
    a_words = t_words + J.t_word;
    a_base = J.t_begin;
    n = J.n;
    b_words = r_words + J.r_word;
    rc = J.rc != 0;
    m = J.m;
    b_base = rc ? static_cast<u64>(J.r_len) - J.q_begin - J.m : J.q_begin;
    lo = J.lo;
    hi = J.hi;
    nb = (static_cast<long long>(n) + 63) >> 6;
    n_super = (nb + R - 1) / R;
    L = static_cast<int>(J.L);
    lane = lane_;
    st_pm = pm + J.store;
    st_sc = sc + J.store;
    s = lane < L ? lane : n_super;  // lanes beyond the ring never work
    fresh = true;
    hout_last = 1;
    score_last = 0;
    result = 0;
  }

  __host__ __device__ void step(long long t, int hin_prev, int score_prev) {
    while (s < n_super) {  // retire finished super-blocks (ring advance)
      const long long last_b = s * R + R - 1 < nb ? s * R + R - 1 : nb - 1;
      const long long jout = nw_jout(last_b, lo);
      if (t - s > (jout < m ? jout : m)) {
        s += L;
        fresh = true;
      } else {
        break;
      }
    }
    if (s >= n_super) return;
    const long long j = t - s;
    const long long b0 = s * R;
    if (j < nw_jin(b0, hi) || j > m) return;
    if (fresh) {
#pragma unroll
      for (int r = 0; r < R; ++r) load_peq(a_words, a_base, n, static_cast<u32>(b0 + r), peq[r]);
      tc.init(b_words, b_base, m, rc, j);
      fresh = false;
    }
    const unsigned c = tc.get(j);
    // producer block b0-1 (previous lane of the ring): inside the band at column j iff j <= jout(b0 - 1)
    const bool prod_active = b0 > 0 && j <= nw_jout(b0 - 1, lo);
    int hin = prod_active ? hin_prev : 1;
    int above_prev_col = prod_active ? score_prev - hin_prev : score_prev;  // score of block b-1 at column j-1
    const u64 slot0 = (static_cast<u64>(t) * L + static_cast<u64>(lane)) * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const long long b = b0 + r;
      if (b >= nb) break;
      const long long jin = nw_jin(b, hi);
      if (j < jin) break;  // this and all lower blocks are still below the band
      if (j > nw_jout(b, lo)) {  // retired above the band: the block below sees the +1 boundary
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
      st_pm[slot0 + r] = NwPm{Pv[r], Mv[r]};
      st_sc[slot0 + r] = score[r];
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
};

// D(x, y) of the banded matrix (x rows of the target span, y columns of the read span); kNwInf outside the band
__host__ __device__ inline u32 nw_cell(const NwJob& J, const NwPm* __restrict__ pm, const int* __restrict__ sc, long long x,
                                       long long y) {
  if (x == 0) return static_cast<u32>(y);
  if (y == 0) return static_cast<u32>(x);
  const long long b = (x - 1) >> 6;
  if (y < nw_jin(b, J.hi) || y > nw_jout(b, J.lo)) return kNwInf;
  const u64 slot = J.store + nw_slot(b, y, J.L, J.R);
  const NwPm v = pm[slot];
  const unsigned bit = static_cast<unsigned>((x - 1) & 63);
  const u64 above = bit == 63 ? 0ULL : (~0ULL << (bit + 1));  // rows of the block below row x
  return static_cast<u32>(sc[slot] - static_cast<int>(RVN_POPC64(v.pv & above)) + static_cast<int>(RVN_POPC64(v.mv & above)));
}

// Walks the optimal path of job J backwards and writes one NwWindowRec per window of the target span.
// `distance` = D(n, m) from the forward sweep (exact).  Returns 0, or 1 when the walk did not end with cost 0 (which
// would mean the stored band is inconsistent — reported, never ignored).
__host__ __device__ inline int nw_traceback(const NwJob& J, const u64* __restrict__ t_words_all,
                                            const u64* __restrict__ r_words_all, const NwPm* __restrict__ pm,
                                            const int* __restrict__ sc, u32 distance, u32 w,
                                            NwWindowRec* __restrict__ recs_all) {
  const u64* tw = t_words_all + J.t_word;
  const u64* rw = r_words_all + J.r_word;
  NwWindowRec* recs = recs_all + J.bp_off;
  const u32 win0 = J.t_begin / w;
  for (u32 x = 0; x < J.n_windows; ++x) {
    NwWindowRec e;
    e.first_t = e.first_q = e.last_t = e.last_q = 0xFFFFFFFFu;
    for (int g = 0; g < 8; ++g) e.grid[g] = 0xFFFFu;
    recs[x] = e;
  }
  long long i = J.n, j = J.m;
  u32 cur = distance;
  // state of the window being walked through (windows are visited from the last to the first)
  u32 cw = 0xFFFFFFFFu;
  bool have = false;
  u32 first_t = 0, first_q = 0, last_t = 0, last_q = 0;
  u32 gq[8];
  int gx = -1;
  u32 gt = 0;
  auto flush = [&]() {
    if (cw == 0xFFFFFFFFu) return;
    NwWindowRec e;
    e.first_t = have ? first_t : 0xFFFFFFFFu;
    e.first_q = first_q;
    e.last_t = last_t;
    e.last_q = last_q;
    for (int g = 0; g < 8; ++g) {
      u32 off = 0xFFFFu;
      if (have && gq[g] != 0xFFFFFFFFu) {
        off = gq[g] > first_q ? gq[g] - first_q : 0u;
        const u32 len = last_q - first_q;
        off = off < len ? off : len;
        off = off < 0xFFFEu ? off : 0xFFFEu;
      }
      e.grid[g] = static_cast<u16>(off);
    }
    recs[cw - win0] = e;
  };
  auto enter = [&](u32 wi) {
    flush();
    cw = wi;
    have = false;
    for (int g = 0; g < 8; ++g) gq[g] = 0xFFFFFFFFu;
    gx = 7;
    gt = wi * w + static_cast<u32>((7ULL * w) / 8);
  };
  // the path consumes target base t with the read standing at oriented position q
  auto on_target_base = [&](u32 t, u32 q) {
    const u32 wi = t / w;
    if (wi != cw) enter(wi);
    while (gx >= 0 && t < gt) {
      --gx;
      if (gx >= 0) gt = wi * w + static_cast<u32>((static_cast<u64>(gx) * w) / 8);
    }
    if (gx >= 0 && t == gt) gq[gx] = q;
  };
  const bool rc = J.rc != 0;
  auto tcode = [&](long long row) -> u32 { return packed_code(tw, static_cast<u64>(J.t_begin) + row - 1); };
  auto qcode = [&](long long col) -> u32 {
    const u64 x = static_cast<u64>(J.q_begin) + col - 1;  // position in the oriented read
    return rc ? 3u - packed_code(rw, static_cast<u64>(J.r_len) - 1 - x) : packed_code(rw, x);
  };
  while (i > 0 || j > 0) {
    bool diag = false;
    if (i > 0 && j > 0) {
      if (tcode(i) == qcode(j)) {
        diag = true;
      } else if (nw_cell(J, pm, sc, i - 1, j - 1) + 1 == cur) {
        diag = true;
        --cur;
      }
    }
    if (diag) {  // CIGAR 'M'
      const u32 t = J.t_begin + static_cast<u32>(i - 1), q = J.q_begin + static_cast<u32>(j - 1);
      on_target_base(t, q);
      if (!have) {
        have = true;
        last_t = t + 1;
        last_q = q + 1;
      }
      first_t = t;
      first_q = q;
      --i;
      --j;
    } else if (j > 0 && (i == 0 || nw_cell(J, pm, sc, i, j - 1) + 1 == cur)) {  // 'I': read base only
      --j;
      --cur;
    } else {  // 'D': target base only
      on_target_base(J.t_begin + static_cast<u32>(i - 1), J.q_begin + static_cast<u32>(j));
      --i;
      --cur;
    }
  }
  flush();
  return cur == 0 ? 0 : 1;
}

}  // namespace rvn
