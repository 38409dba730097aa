// nwpath.h — the global alignment PATH of a read against its target span and the window breakpoints racon derives
// from it (racon Overlap::find_breaking_points: edlibAlign(query, target, EDLIB_MODE_NW, EDLIB_TASK_PATH) -> CIGAR ->
// for every window of w target bases the first and the last aligned ('M') pair; call site of the whole round:
// RavenLib/src/polish.cc:43-51).  The polishing front end (polish.hip) runs this for the best overlap of every read.
//
// Formulation for the device (kernel in nwpath.hip; everything here is __host__ __device__ so that the very same
// code can be stepped lane by lane on the CPU, rvn_test_nw_breakpoints).  One wave owns one alignment from start to end:
//   sweep     Myers' bit-vector blocks over a DIAGONAL band that contains every alignment of cost <= k (offsets
//             row - column in [-lo, hi]): lanes = 64-row blocks (R per lane) reused as a ring, systolic over the
//             columns exactly like the distance kernel (edit_distance.hip).  The result is exact iff it is <= k
//             (edlib's criterion); otherwise k is doubled and the sweep repeated, in the kernel.
//   pass 1    one sweep over all columns that keeps only a CHECKPOINT of the band every kNwSeg columns
//             ((Pv, Mv, bottom score) of every block inside the band at that column): O(m / kNwSeg x band) memory
//             per alignment instead of O(m x band) — whole-matrix stores made the stage quadratic in the read length.
//   walk      the path is walked backwards segment by segment: the wave re-sweeps the kNwSeg columns of a segment from
//             its checkpoint, this time storing every block update into a small per-wave scratch, then walks through
//             them.  Equal bases always take the diagonal (D(i,j) = D(i-1,j-1) when the bases match, so no score is
//             needed); at a mismatch the neighbour scores come from the stored words (block bottom score -/+ popcounts
//             of the deltas below the row).  Preference on ties: diagonal, then query-base-only ('I'), then
//             target-base-only ('D') — the rule of the CPU restatement (oracle NwPath); edlib documents no tie rule for
//             paths found by its Hirschberg split, any optimal path is "the" edlib path.  Banded values equal the
//             full-matrix values on every cell the walk can take (each lies on an optimal alignment, which the band
//             contains, and a too-large banded value of a cell that is NOT a valid predecessor stays invalid), so the
//             path is the full-matrix path.  The walk emits no CIGAR: it folds find_breaking_points in and writes, per
//             window, the first / last aligned pair and the read offsets at eight fixed target positions (POA band guide).
#pragma once

#include "myers.h"

namespace rvn {

constexpr int kNwSeg = 256;  // columns per segment (checkpoint spacing)

// rows = target span (pattern, forward strand), columns = read span in the target's orientation (text)
struct NwJob {
  u64 t_word;   // first word of the target in the packed target set
  u64 r_word;   // first word of the read in the packed read set
  u64 ckpt;     // first checkpoint slot of this job
  u64 bp_off;   // first window record of this job
  u32 t_begin, n;  // target span [t_begin, t_begin + n)
  u32 q_begin, m;  // read span [q_begin, q_begin + m) in the orientation of the target
  u32 r_len;       // length of the read
  u32 rc;          // 1: the read is reverse-complemented (overlap on the opposite strand)
  u32 k;           // first cost threshold (>= |n - m|); doubled in the kernel up to kcap
  u32 kcap;        // largest threshold this launch may use (ring of <= 64 lanes with R blocks each, checkpoint rows)
  u32 ckpt_nb;     // checkpoint row stride: blocks inside the band at kcap
  u32 R;           // blocks per lane
  u32 read, target;  // indices in their sets
  u32 n_windows;     // windows touched by the target span
  u32 bin;           // 0: wave-per-alignment kernel (R blocks per lane); 1..4: lane-per-alignment kernel, ring of 8 * bin blocks
};
static_assert(sizeof(NwJob) == 88, "NwJob layout");

struct NwWindowRec {  // per (job, window): racon's breakpoint pair + band guide
  u32 first_t, first_q;  // first aligned pair of the window (target / oriented read position); first_t == ~0: none
  u32 last_t, last_q;    // one past the last aligned pair
  u16 grid[8];           // read offset (relative to first_q, clamped) where the path crosses target position
                         // window_start + (x * w) / 8; 0xFFFF = the alignment does not cover it
};
static_assert(sizeof(NwWindowRec) == 32, "NwWindowRec layout");

constexpr u32 kNwInf = 0x3FFFFFFFu;

// All row / column / block / step indices fit 32 bits (spans are shorter than 2^31 bases): plain ints keep the
// kernels' register count down; only final addresses are computed in 64 bits.
struct NwBand {  // diagonal band of threshold k: -lo <= row - column <= hi; L ring lanes
  int lo, hi, nb, n_super;
  int L;
};
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
  const u64 l = (num + den - 1) / den;
  return static_cast<u32>(l < 1 ? 1 : l);
}
__host__ __device__ inline NwBand nw_band(u32 n, u32 m, u32 k, u32 R) {
  NwBand B;
  B.lo = static_cast<int>(nw_band_lo(n, m, k));
  B.hi = static_cast<int>(nw_band_hi(n, m, k));
  B.nb = static_cast<int>((static_cast<u64>(n) + 63) >> 6);
  B.n_super = (B.nb + static_cast<int>(R) - 1) / static_cast<int>(R);
  B.L = static_cast<int>(nw_ring_lanes(static_cast<u32>(B.lo), static_cast<u32>(B.hi), R));
  return B;
}
// first / last column at which block b is inside the band
__host__ __device__ inline int nw_jin(int b, int hi) {
  const int j = 64 * b + 1 - hi;
  return j < 1 ? 1 : j;
}
__host__ __device__ inline int nw_jout(int b, int lo) { return 64 * b + 64 + lo; }
// first / last block inside the band at column j
__host__ __device__ inline int nw_bfirst(int j, int lo) {
  const int x = j - 64 - lo;  // smallest b with 64 b + 64 + lo >= j
  return x <= 0 ? 0 : (x + 63) >> 6;
}
__host__ __device__ inline int nw_blast(int j, int hi, int nb) {
  const int b = (j + hi - 1) >> 6;  // largest b with 64 b + 1 - hi <= j  (j >= 1, hi >= 0)
  return b < nb - 1 ? b : nb - 1;
}
// blocks per checkpoint row at threshold k
__host__ __device__ inline u32 nw_ckpt_blocks(u32 n, u32 m, u32 k) {
  return (nw_band_lo(n, m, k) + nw_band_hi(n, m, k)) / 64 + 3;
}
__host__ __device__ inline u64 nw_ckpt_slots(u32 m, u32 ckpt_nb) {
  return (static_cast<u64>(m) / kNwSeg + 1) * ckpt_nb;
}
// rows (systolic steps) of the per-wave segment scratch; one row = 64 lanes x R entries
__host__ __device__ constexpr u32 nw_seg_rows() { return kNwSeg + 64 + kNwSeg / 64 + 6; }

struct NwPm {
  u64 pv, mv;
};

// Where a sweep keeps block states.
struct NwStore {
  NwPm* ck_pm;   // checkpoints of the job: [column / kNwSeg][block - bfirst(column)], stride ckpt_nb
  int* ck_sc;
  u32 ckpt_nb;
  NwPm* seg_pm;  // scratch of the segment being walked: [step - t0][lane][r]
  int* seg_sc;
};

// One lane of a sweep over columns (j0, j_end] of the band.  The kernel (and the CPU stepper) calls step(t, ...) for
// t = t0 .. t1 with the producer lane's (hout_last, score_last) of the previous step.
// mode 0: pass 1 — every kNwSeg-th column is checkpointed;  mode 1: segment — every block update goes to the scratch.
template <int R>
struct NwLane {
  const u64* a_words;
  const u64* b_words;
  u64 a_base, b_base;
  u32 n, m;
  bool rc;
  NwBand B;
  int lane, mode;
  int j0, j_end, t0;
  NwStore st;
  u64 Pv[R], Mv[R];
  BlockPlanes pl[R];  // match masks of the lane's blocks as bit planes (myers.h)
  int score[R];
  int s;
  bool fresh;
  TextCursor tc;
  bool tc_valid;  // tc stands at the lane's current column (only the block at the top of the band reads the text itself)
  // what the consumer (next lane of the ring) reads one step later: (hout + 1) | text symbol << 2 of the column just done.
  // The symbol of column j travels down the band with the horizontal deltas, so only the topmost block of the band at
  // column j loads it — a load on every lane's path would make every step wait on the vector-memory counter, which
  // the kernel's stores share.
  int xfer_last, score_last;
  u32 result;  // D(n, m) + 1 on the one lane that computes it
  // R == 1: column range (c_ja .. c_jb) of the block the lane holds and the last column (c_prod) at which the block above
  // is still inside the band, valid while c_s == s — what the plain block update (fast_step) needs instead of re-deriving
  // the band geometry at every step
  int c_s, c_any, c_ja, c_jb, c_end, c_prod, c_rn;
  bool c_final;

  __host__ __device__ void init(const NwJob& J, const u64* t_words, const u64* r_words, const NwBand& band,
                                const NwStore& store, int lane_) {
    a_words = t_words + J.t_word;
    a_base = J.t_begin;
    n = J.n;
    b_words = r_words + J.r_word;
    rc = J.rc != 0;
    m = J.m;
    b_base = rc ? static_cast<u64>(J.r_len) - J.q_begin - J.m : J.q_begin;
    B = band;
    st = store;
    lane = lane_;
  }

  // first / last step of a sweep over columns (j0, j_end]
  __host__ __device__ static int sweep_t0(const NwBand& B, int j0) { return j0 + 1 + nw_bfirst(j0 + 1, B.lo) / R; }
  __host__ __device__ static int sweep_t1(const NwBand& B, int j_end) { return j_end + nw_blast(j_end, B.hi, B.nb) / R; }

  __host__ __device__ void begin_sweep(int j0_, int j_end_, int mode_) {
    j0 = j0_;
    j_end = j_end_;
    mode = mode_;
    t0 = sweep_t0(B, j0);
    s = lane < B.L ? lane : B.n_super;  // lanes beyond the ring never work
    while (s < B.n_super) {  // super-blocks that left the band before the sweep begins
      const int last_b = s * R + R - 1 < B.nb ? s * R + R - 1 : B.nb - 1;
      if (nw_jout(last_b, B.lo) < j0 + 1) s += B.L;
      else break;
    }
    fresh = true;
    tc_valid = false;
    xfer_last = 2;  // hout = +1, symbol 0
    score_last = 0;
    result = 0;
    c_s = -1;
  }

  __host__ __device__ void refresh_cache() {
    c_s = s;
    if (s < B.n_super) {
      const int b0 = s * R;
      const int b_last = b0 + R - 1 < B.nb ? b0 + R - 1 : B.nb - 1;
      const int jin0 = nw_jin(b0, B.hi), jin_last = nw_jin(b_last, B.hi);
      const int jout0 = nw_jout(b0, B.lo), jout_last = nw_jout(b_last, B.lo);
      c_any = jin0 > j0 + 1 ? jin0 : j0 + 1;          // first column at which a block of the lane is active
      c_ja = jin_last > j0 + 1 ? jin_last : j0 + 1;   // first column of the last block to enter
      c_jb = jout0 < j_end ? jout0 : j_end;           // last column with every block inside
      c_end = jout_last < j_end ? jout_last : j_end;  // last column of the lane's blocks (then: ring advance)
      c_prod = b0 > 0 ? nw_jout(b0 - 1, B.lo) : -1;
      c_rn = b_last - b0 + 1;
      c_final = b_last == B.nb - 1;
    }
  }

  // What step(t, ..) would do on this lane: 0 = nothing, 1 = the plain update of ALL its blocks at column t - s
  // (fast_step does exactly that), 2 = anything else (ring advance, a block entering or leaving the band, first
  // columns, the final cell, ...).  The kernels take the short path when no lane of the wave says 2.
  __host__ __device__ int classify(int t) const {  // branch-free: it runs on every lane at every step
    const int j = t - s;
    const bool live = s < B.n_super;
    const bool active = j >= c_any && j <= c_end;  // some block of the lane is inside the band and the sweep
    const bool plain = j > c_ja && j <= c_jb;      // all of them are, and none is at its first column
    const bool event = (c_s != s) || j > c_end     // stale cache / ring advance
                       || (active && (!plain || fresh || (j > c_prod && !tc_valid)  // partial / starts reading the text
                                      || (c_final && j == static_cast<int>(m))));    // the final cell
    return live ? (event ? 2 : (active ? 1 : 0)) : 0;
  }

  __host__ __device__ void fast_step(int t, int x_prev) {
    const int j = t - s;
    const bool fed = j <= c_prod;
    const unsigned c = fed ? static_cast<unsigned>(x_prev >> 2) : tc.get(j);
    int hin = fed ? (x_prev & 3) - 1 : 1;
    const u32 slot0 = (static_cast<u32>(t - t0) * static_cast<u32>(B.L) + static_cast<u32>(lane)) * R;
    const bool ckpt = mode != 1 && j % kNwSeg == 0;
    const u64 cs0 = ckpt ? static_cast<u64>(j / kNwSeg) * st.ckpt_nb + static_cast<u64>(s * R - nw_bfirst(j, B.lo)) : 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < c_rn) {
        const int hout = myers_block(Pv[r], Mv[r], planes_eq(pl[r], c), hin);
        score[r] += hout;
        hin = hout;
        if (mode == 1) {
          st.seg_pm[slot0 + r] = NwPm{Pv[r], Mv[r]};
          st.seg_sc[slot0 + r] = score[r];
        } else if (ckpt) {
          st.ck_pm[cs0 + r] = NwPm{Pv[r], Mv[r]};
          st.ck_sc[cs0 + r] = score[r];
        }
      }
    }
    xfer_last = (hin + 1) | static_cast<int>(c << 2);
    score_last = score[R - 1];
  }

  __host__ __device__ void step(int t, int x_prev, int score_prev) {
    const int hin_prev = (x_prev & 3) - 1;
    while (s < B.n_super) {  // retire finished super-blocks (ring advance)
      const int last_b = s * R + R - 1 < B.nb ? s * R + R - 1 : B.nb - 1;
      const int jout = nw_jout(last_b, B.lo);
      if (t - s > (jout < j_end ? jout : j_end)) {
        s += B.L;
        fresh = true;
      } else {
        break;
      }
    }
    if (s >= B.n_super) return;
    const int j = t - s;
    const int b0 = s * R;
    if (j <= j0 || j < nw_jin(b0, B.hi) || j > j_end) return;
    if (fresh) {
#pragma unroll
      for (int r = 0; r < R; ++r) pl[r] = load_planes(a_words, a_base, n, static_cast<u32>(b0 + r));
      tc_valid = false;
      fresh = false;
    }
    // producer block b0-1 (previous lane of the ring): inside the band at column j iff j <= jout(b0 - 1)
    const bool prod_active = b0 > 0 && j <= nw_jout(b0 - 1, B.lo);
    unsigned c;
    if (prod_active) {  // the symbol of column j arrives with the producer's delta
      c = static_cast<unsigned>(x_prev >> 2);
      tc_valid = false;
    } else {
      if (!tc_valid) {
        tc.init(b_words, b_base, m, rc, j);
        tc_valid = true;
      }
      c = tc.get(j);
    }
    int hin = prod_active ? hin_prev : 1;
    int above_prev_col = prod_active ? score_prev - hin_prev : score_prev;  // score of block b-1 at column j-1
    const u64 seg_slot0 = (static_cast<u64>(t - t0) * B.L + static_cast<u64>(lane)) * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int b = b0 + r;
      if (b >= B.nb) break;
      const int jin = nw_jin(b, B.hi);
      if (j < jin) break;  // this and all lower blocks are still below the band
      if (j > nw_jout(b, B.lo)) {  // retired above the band: the block below sees the +1 boundary
        hin = 1;
        continue;
      }
      if (j == (jin > j0 + 1 ? jin : j0 + 1)) {  // first column of this block in this sweep
        if (jin <= j0) {  // inside the band before the sweep began: resume from the checkpoint of column j0
          const u64 cs = static_cast<u64>(j0 / kNwSeg) * st.ckpt_nb + static_cast<u64>(b - nw_bfirst(j0, B.lo));
          const NwPm v = st.ck_pm[cs];
          Pv[r] = v.pv;
          Mv[r] = v.mv;
          score[r] = st.ck_sc[cs];
        } else {  // enters the band here: edlib's all-(+1) upper bound
          Pv[r] = ~0ULL;
          Mv[r] = 0;
          score[r] = jin == 1 ? static_cast<int>(64 * (b + 1)) : above_prev_col + 64;
        }
      }
      const int old = score[r];
      const int hout = myers_block(Pv[r], Mv[r], planes_eq(pl[r], c), hin);
      score[r] = old + hout;
      above_prev_col = old;
      hin = hout;
      if (mode == 1) {
        st.seg_pm[seg_slot0 + r] = NwPm{Pv[r], Mv[r]};
        st.seg_sc[seg_slot0 + r] = score[r];
      } else if (j % kNwSeg == 0) {
        const u64 cs = static_cast<u64>(j / kNwSeg) * st.ckpt_nb + static_cast<u64>(b - nw_bfirst(j, B.lo));
        st.ck_pm[cs] = NwPm{Pv[r], Mv[r]};
        st.ck_sc[cs] = score[r];
      }
      if (b == B.nb - 1 && j == m) {
        // D[n][m] = bottom score of the last block minus the vertical deltas of the padded rows
        const u32 used = n - static_cast<u32>(64 * b);
        const u64 padmask = used >= 64 ? 0ULL : ~((1ULL << used) - 1ULL);
        result = static_cast<u32>(score[r] - RVN_POPC64(Pv[r] & padmask) + RVN_POPC64(Mv[r] & padmask)) + 1u;
      }
    }
    xfer_last = (hin + 1) | static_cast<int>(c << 2);
    score_last = score[R - 1];
  }
};

// Cell values of the segment held in the wave kernel's scratch (time-major [step][lane][r]) + the checkpoint column
struct NwSegCells {
  NwBand B;
  NwStore st;
  int seg_j0, seg_t0;
  u32 R;
  // D(x, y) of the banded matrix for y in [seg_j0, seg_j0 + kNwSeg]; kNwInf outside the band
  __host__ __device__ u32 get(int x, int y) const {
    if (x == 0) return static_cast<u32>(y);
    if (y == 0) return static_cast<u32>(x);
    const int b = (x - 1) >> 6;
    if (y < nw_jin(b, B.hi) || y > nw_jout(b, B.lo)) return kNwInf;
    NwPm v;
    int sc;
    if (y == seg_j0) {  // the checkpointed column
      const u64 cs = static_cast<u64>(y / kNwSeg) * st.ckpt_nb + static_cast<u64>(b - nw_bfirst(y, B.lo));
      v = st.ck_pm[cs];
      sc = st.ck_sc[cs];
    } else {
      const int s = b / static_cast<int>(R);
      const u64 slot = (static_cast<u64>(y + s - seg_t0) * B.L + static_cast<u64>(s % B.L)) * R + static_cast<u64>(b % static_cast<int>(R));
      v = st.seg_pm[slot];
      sc = st.seg_sc[slot];
    }
    const unsigned bit = static_cast<unsigned>((x - 1) & 63);
    const u64 below = bit == 63 ? 0ULL : (~0ULL << (bit + 1));  // rows of the block below row x
    return static_cast<u32>(sc - static_cast<int>(RVN_POPC64(v.pv & below)) + static_cast<int>(RVN_POPC64(v.mv & below)));
  }
};

// The backward walk, resumable segment by segment.  Cells::get(x, y) = D(x, y) for the columns of the current segment.
template <class Cells>
struct NwWalkerT {
  Cells cells;
  // job
  const u64* tw;
  const u64* rw;
  u32 t_begin, q_begin, r_len, w, win0;
  bool rc;
  NwWindowRec* recs;
  int seg_j0;
  // position
  int i, j;
  u32 cur;
  // window being walked through (windows are visited from the last to the first); cw_lo = its first target base
  u32 cw, cw_lo;
  bool have;
  u32 first_t, first_q, last_t, last_q;
  u32 gq[8];
  int gx;
  u32 gt;
  // the next (up to 32) bases of the target and of the oriented read, the current one in the top two bits; the walk
  // compares 32 bases per step with them and takes a whole run of matches at once
  u64 t_tail, q_tail;
  int t_have, q_have;

  __host__ __device__ void init(const NwJob& J, const u64* t_words_all, const u64* r_words_all, u32 distance, u32 w_,
                                NwWindowRec* recs_all) {
    tw = t_words_all + J.t_word;
    rw = r_words_all + J.r_word;
    t_begin = J.t_begin;
    q_begin = J.q_begin;
    r_len = J.r_len;
    w = w_;
    win0 = J.t_begin / w_;
    rc = J.rc != 0;
    recs = recs_all + J.bp_off;
    seg_j0 = 0;
    i = static_cast<int>(J.n);
    j = static_cast<int>(J.m);
    cur = distance;
    cw = 0xFFFFFFFFu;
    cw_lo = 0;
    have = false;
    first_t = first_q = last_t = last_q = 0;
    for (int g = 0; g < 8; ++g) gq[g] = 0xFFFFFFFFu;
    gx = -1;
    gt = 0;
    t_tail = q_tail = 0;
    t_have = q_have = 0;
  }
  __host__ __device__ u32 cell(int x, int y) const { return cells.get(x, y); }

  // order of the 32 two-bit groups reversed
  __host__ __device__ static u64 rev2(u64 v) {
    v = ((v & 0x3333333333333333ULL) << 2) | ((v >> 2) & 0x3333333333333333ULL);
    v = ((v & 0x0F0F0F0F0F0F0F0FULL) << 4) | ((v >> 4) & 0x0F0F0F0F0F0F0F0FULL);
    v = ((v & 0x00FF00FF00FF00FFULL) << 8) | ((v >> 8) & 0x00FF00FF00FF00FFULL);
    v = ((v & 0x0000FFFF0000FFFFULL) << 16) | ((v >> 16) & 0x0000FFFF0000FFFFULL);
    return (v << 32) | (v >> 32);
  }
  // bases [first, first + cnt) of a packed sequence (1 <= cnt <= 32), base `first` in the low bits; never touches a
  // word that holds none of them
  __host__ __device__ static u64 load_span(const u64* words, u64 first, int cnt) {
    const u64 wi = first >> 5;
    const unsigned off = static_cast<unsigned>(first & 31) * 2;
    u64 x = words[wi] >> off;
    if (off && ((first + static_cast<u64>(cnt) - 1) >> 5) != wi) x |= words[wi + 1] << (64 - off);
    return x;
  }
  __host__ __device__ void refill_t() {  // rows i, i - 1, ...
    const u64 pos = static_cast<u64>(t_begin) + static_cast<u64>(i) - 1;
    const int cnt = i < 32 ? i : 32;
    t_tail = load_span(tw, pos - static_cast<u64>(cnt - 1), cnt) << (2 * (32 - cnt));
    t_have = cnt;
  }
  __host__ __device__ void refill_q() {  // columns j, j - 1, ... of the oriented read
    const u64 x = static_cast<u64>(q_begin) + static_cast<u64>(j) - 1;
    const int cnt = j < 32 ? j : 32;
    if (!rc) {
      q_tail = load_span(rw, x - static_cast<u64>(cnt - 1), cnt) << (2 * (32 - cnt));
    } else {  // stored position r_len - 1 - x and upwards, complemented
      q_tail = ~rev2(load_span(rw, static_cast<u64>(r_len) - 1 - x, cnt));
    }
    q_have = cnt;
  }
  // number of matches going down the diagonal from (i, j), at most `lim` (>= 1) and at most what the tails hold
  __host__ __device__ int match_run(int lim) {
    if (t_have == 0) refill_t();
    if (q_have == 0) refill_q();
    int avail = t_have < q_have ? t_have : q_have;
    avail = avail < lim ? avail : lim;
    const u64 x = t_tail ^ q_tail;
    const u64 y = (x | (x >> 1)) & 0x5555555555555555ULL;  // one bit per differing base, the current base at bit 62
#if defined(__HIP_DEVICE_COMPILE__)
    const int run = y ? (__clzll(static_cast<long long>(y)) >> 1) : 32;
#else
    const int run = y ? (__builtin_clzll(y) >> 1) : 32;
#endif
    return run < avail ? run : avail;
  }
  __host__ __device__ void consume(int dt, int dq) {  // the walk moved dt rows and dq columns
    t_tail = dt >= 32 ? 0 : t_tail << (2 * dt);
    t_have -= dt;
    q_tail = dq >= 32 ? 0 : q_tail << (2 * dq);
    q_have -= dq;
  }

  __host__ __device__ void flush(bool write) {
    if (cw == 0xFFFFFFFFu || !write) return;
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
  }
  // the path consumes target base t with the read standing at oriented position q
  __host__ __device__ void on_target_base(u32 t, u32 q, bool write) {
    if (cw == 0xFFFFFFFFu || t < cw_lo || t - cw_lo >= w) {  // another window (the division only here)
      const u32 wi = t / w;
      flush(write);
      cw = wi;
      cw_lo = wi * w;
      have = false;
      for (int g = 0; g < 8; ++g) gq[g] = 0xFFFFFFFFu;
      gx = 7;
      gt = cw_lo + static_cast<u32>((7ULL * w) / 8);
    }
    while (gx >= 0 && t < gt) {
      --gx;
      if (gx >= 0) gt = cw_lo + static_cast<u32>((static_cast<u64>(gx) * w) / 8);
    }
    if (gx >= 0 && t == gt) {
#pragma unroll
      for (int g = 0; g < 8; ++g)
        if (g == gx) gq[g] = q;  // unrolled select: a runtime index would push the array to scratch memory
    }
  }
  __host__ __device__ void take_diag(bool write) {  // CIGAR 'M'
    const u32 t = t_begin + static_cast<u32>(i - 1), q = q_begin + static_cast<u32>(j - 1);
    on_target_base(t, q, write);
    if (!have) {
      have = true;
      last_t = t + 1;
      last_q = q + 1;
    }
    first_t = t;
    first_q = q;
    --i;
    --j;
    consume(1, 1);
  }
  // r >= 1 consecutive 'M' steps whose target bases lie in one window: what r calls of take_diag leave behind
  __host__ __device__ void take_diag_run(int r, bool write) {
    const u32 t_hi = t_begin + static_cast<u32>(i - 1), q_hi = q_begin + static_cast<u32>(j - 1);
    const u32 t_lo = t_hi - static_cast<u32>(r - 1);
    on_target_base(t_hi, q_hi, write);
    if (!have) {
      have = true;
      last_t = t_hi + 1;
      last_q = q_hi + 1;
    }
    // the bases t_hi - 1 .. t_lo: only the grid points among them leave a trace
    u32 t_prev = t_hi;
    while (t_prev > t_lo) {
      while (gx >= 0 && gt >= t_prev) {  // on_target_base(t_prev - 1, ..) would step past these
        --gx;
        if (gx >= 0) gt = cw_lo + static_cast<u32>((static_cast<u64>(gx) * w) / 8);
      }
      if (gx < 0 || gt < t_lo) break;
      const u32 qv = q_hi - (t_hi - gt);
#pragma unroll
      for (int g = 0; g < 8; ++g)
        if (g == gx) gq[g] = qv;
      t_prev = gt;
    }
    first_t = t_lo;
    first_q = q_hi - static_cast<u32>(r - 1);
    i -= r;
    j -= r;
    consume(r, r);
  }

  // walks while the current column lies inside the segment in the scratch (j > seg_j0) and rows remain
  __host__ __device__ void walk(bool write) {
    while (i > 0 && j > seg_j0) {
      const int room = i < j - seg_j0 ? i : j - seg_j0;
      int r = match_run(room);
      if (r > 0) {  // a match is always taken diagonally: the whole run at once, window by window
        const u32 t_hi = t_begin + static_cast<u32>(i - 1);
        // bases from t_hi down to the start of its window
        const bool same = cw != 0xFFFFFFFFu && t_hi >= cw_lo && t_hi - cw_lo < w;
        const u32 in_window = same ? t_hi - cw_lo + 1 : t_hi - (t_hi / w) * w + 1;
        r = static_cast<u32>(r) < in_window ? r : static_cast<int>(in_window);
        take_diag_run(r, write);
      } else if (cell(i - 1, j - 1) + 1 == cur) {
        --cur;
        take_diag(write);
      } else if (cell(i, j - 1) + 1 == cur) {  // 'I': read base only
        --j;
        --cur;
        consume(0, 1);
      } else {  // 'D': target base only
        on_target_base(t_begin + static_cast<u32>(i - 1), q_begin + static_cast<u32>(j), write);
        --i;
        --cur;
        consume(1, 0);
      }
    }
  }
  // the rest needs no scores: only read bases (i == 0) or only target bases (j == 0) are left
  __host__ __device__ int finish(bool write) {
    while (j > 0 && i == 0) {
      --j;
      --cur;
    }
    while (i > 0 && j == 0) {
      on_target_base(t_begin + static_cast<u32>(i - 1), q_begin, write);
      --i;
      --cur;
    }
    flush(write);
    return (cur == 0 && i == 0 && j == 0) ? 0 : 1;
  }
};

// the wave kernel's walker
struct NwWalker : NwWalkerT<NwSegCells> {
  __host__ __device__ void init(const NwJob& J, const u64* t_words_all, const u64* r_words_all, const NwBand& band,
                                const NwStore& store, u32 distance, u32 w_, NwWindowRec* recs_all) {
    NwWalkerT<NwSegCells>::init(J, t_words_all, r_words_all, distance, w_, recs_all);
    cells.B = band;
    cells.st = store;
    cells.R = J.R;
    cells.seg_j0 = 0;
    cells.seg_t0 = 0;
  }
  __host__ __device__ void set_segment(int j0, int t0) {
    seg_j0 = j0;
    cells.seg_j0 = j0;
    cells.seg_t0 = t0;
  }
};

}  // namespace rvn
