// nwpath.h — the global alignment PATH of a read against its target span and the window breakpoints racon derives
// from it (racon Overlap::find_breaking_points: edlibAlign(query, target, EDLIB_MODE_NW, EDLIB_TASK_PATH) -> CIGAR ->
// for every window of w target bases the first and the last aligned ('M') pair; call site of the whole round:
// RavenLib/src/polish.cc:43-51).  The polishing front end (polish.hip) runs this for the best overlap of every read.
//
// Formulation for the device (kernels in nwpath.hip; everything here, in nwsweep.h and in nwtrace.h is
// __host__ __device__ so that the very same code can be stepped lane by lane on the CPU, rvn_test_nw_breakpoints):
//   sweep (nwsweep.h)   ONE forward pass of Myers' bit-vector blocks over a DIAGONAL band that contains every alignment
//             of cost <= k (offsets row - column in [-lo, hi]).  Lanes = super-blocks of R 64-row blocks, reused as a
//             ring of L lanes, systolic over the columns: at step t the lane that holds super-block s works on column
//             t - s and hands its horizontal delta to the next lane.  The result is exact iff it is <= k (edlib's
//             criterion); a too small k is doubled by the host.  All block state stays in registers; what the sweep
//             leaves in HBM is small and written coalesced, indexed by STEP (so all lanes write at the same step):
//               hs   the horizontal delta at the bottom of every block at every step, 2 bits (one u32 per lane and block
//                    every 16 steps) — with it ANY single block can be recomputed later without the blocks above it;
//               ck   (Pv, Mv) of every block every 32 steps.
//   trace (nwtrace.h)   the path is walked backwards by ONE lane per alignment.  The walker only ever needs the block
//             that holds its row: that block is recomputed from its last checkpoint (<= 32 columns, horizontal input
//             from hs) into LDS, then walked.  Equal bases always take the diagonal (D(i,j) = D(i-1,j-1) when the bases
//             match, so no score is needed); at a mismatch the three neighbours are compared through the stored vertical
//             deltas and the block's horizontal input — differences only, no absolute score is stored anywhere.
//             Preference on ties: diagonal, then query-base-only ('I'), then target-base-only ('D') — the rule of the
//             CPU restatement (oracle NwPath); edlib documents no tie rule for paths found by its Hirschberg split, any
//             optimal path is "the" edlib path.  Banded values equal the full-matrix values on every cell the walk can
//             take (each lies on an optimal alignment, which the band contains); values the band only bounds from above
//             (entering blocks, the +1 boundary of retired ones) can never pass a test the true value fails, so the path
//             is the full-matrix path.  The walk emits no CIGAR: it folds find_breaking_points in and writes, per
//             window, the first / last aligned pair and the read offsets at eight fixed target positions (POA band guide).
#pragma once

#include "myers.h"

namespace rvn {

constexpr int kNwHsSteps = 16;  // steps per hs word (2 bits each)
constexpr int kNwCkSteps = 32;  // steps between checkpoints = longest re-sweep of the walk

// rows = target span (pattern, forward strand), columns = read span in the target's orientation (text)
struct NwJob {
  u64 t_word;   // first word of the target in the packed target set
  u64 r_word;   // first word of the read in the packed read set
  u64 ckpt;     // first checkpoint entry (16 B each) of this job in the launch's ck buffer
  u64 hs;       // first word of this job in the launch's hs buffer
  u64 bp_off;   // first window record of this job
  u32 t_begin, n;  // target span [t_begin, t_begin + n)
  u32 q_begin, m;  // read span [q_begin, q_begin + m) in the orientation of the target
  u32 r_len;       // length of the read
  u32 rc;          // 1: the read is reverse-complemented (overlap on the opposite strand)
  u32 k;           // cost threshold of this launch (>= |n - m|); doubled by the host when the result exceeds it
  u32 kcap;        // largest threshold the job's current kernel variant (R, G) can hold
  u32 R;           // blocks per lane
  u32 G;           // lanes per alignment of the sweep kernel (ring of L <= G lanes)
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

// All row / column / block / step indices fit 32 bits (spans are shorter than 2^30 bases): plain ints keep the
// kernels' register count down; only final addresses are computed in 64 bits.
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

// Geometry of one sweep: the band of threshold k cut into super-blocks of R blocks.  A super-block is inside the band as
// a whole (from the first column of its first block to the last column of its last block): the computed region is a
// superset of Ukkonen's band, which only makes more cells exact.
struct NwGeo {
  int n, m;      // rows, columns
  int lo, hi;    // -lo <= row - column <= hi
  int R, L;      // blocks per lane, ring lanes
  int nb;        // 64-row blocks
  int n_super;   // super-blocks
  int n_steps;   // systolic steps t = 1 .. n_steps (super-block s works on column t - s); the last one only retires
  __host__ __device__ int ja(int s) const {  // first column of super-block s
    const int j = 64 * s * R + 1 - hi;
    return j < 1 ? 1 : j;
  }
  __host__ __device__ int je(int s) const {  // last column of super-block s
    const int j = 64 * (s * R + R) + lo;
    return j < m ? j : m;
  }
  __host__ __device__ int jfed(int s) const {  // last column at which the super-block above is still inside the band
    return s > 0 ? 64 * s * R + lo : 0;
  }
  __host__ __device__ u64 hs_words() const {
    return static_cast<u64>((n_steps + kNwHsSteps - 1) / kNwHsSteps) * static_cast<u64>(L) * static_cast<u64>(R);
  }
  __host__ __device__ u64 ck_entries() const {
    return static_cast<u64>(n_steps / kNwCkSteps) * static_cast<u64>(L) * static_cast<u64>(R);
  }
};
__host__ __device__ inline NwGeo nw_geo(u32 n, u32 m, u32 k, u32 R) {
  NwGeo g;
  g.n = static_cast<int>(n);
  g.m = static_cast<int>(m);
  g.lo = static_cast<int>(nw_band_lo(n, m, k));
  g.hi = static_cast<int>(nw_band_hi(n, m, k));
  g.R = static_cast<int>(R);
  g.L = static_cast<int>(nw_ring_lanes(static_cast<u32>(g.lo), static_cast<u32>(g.hi), R));
  g.nb = static_cast<int>((static_cast<u64>(n) + 63) >> 6);
  g.n_super = (g.nb + g.R - 1) / g.R;
  g.n_steps = g.m + g.n_super;
  return g;
}

struct NwPm {
  u64 pv, mv;
};

// Horizontal deltas travel as two bits: bit 0 = +1, bit 1 = -1 (0 = no change).
// edlib calculateBlock with that encoding: advance one 64-row block by one column.
__host__ __device__ __forceinline__ int myers_block2(u64& Pv, u64& Mv, u64 Eq, int hin) {
  const u64 pos = static_cast<u64>(hin & 1), neg = static_cast<u64>((hin >> 1) & 1);
  const u64 Xv = Eq | Mv;
  Eq |= neg;
  const u64 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
  u64 Ph = Mv | ~(Xh | Pv);
  u64 Mh = Pv & Xh;
  const int hout = static_cast<int>(Ph >> 63) | (static_cast<int>(Mh >> 63) << 1);
  Ph = (Ph << 1) | pos;
  Mh = (Mh << 1) | neg;
  Pv = Mh | ~(Xv | Ph);
  Mv = Ph & Xv;
  return hout;
}
__host__ __device__ __forceinline__ int nw_delta(int bits) { return (bits & 1) - ((bits >> 1) & 1); }

// 16 text symbols (2 bits each, the symbol of column col0 in bits 0-1) of columns col0 .. col0 + 15 of the read span in
// the target's orientation; columns outside 1 .. m hold arbitrary symbols (nobody uses them).
// first = index (in the read's packed words) of the base of column 1: b_base forward, b_base + m - 1 reverse-complemented.
// Split in two so that a kernel can issue the loads a whole group of steps before it needs the symbols: both words are
// loaded unconditionally (every packed set has slack behind its last word), nothing waits in between.
struct NwRaw {
  u64 w0, w1;
};
__host__ __device__ __forceinline__ long long nw_text16_base(long long first, bool rc, int col0) {
  // reverse-complemented: stored positions first - (col - 1), descending — the 16 bases ENDING at the one of col0
  return rc ? first - col0 - 14 : first + col0 - 1;
}
__host__ __device__ __forceinline__ NwRaw nw_text16_load(const u64* __restrict__ words, long long base) {
  if (base < 0) base = 0;  // before the first word of the read: the finish shifts garbage columns in instead
  const u64 wi = static_cast<u64>(base) >> 5;
  return NwRaw{words[wi], words[wi + 1]};
}
__host__ __device__ __forceinline__ u32 nw_text16_finish(const NwRaw& r, long long base, bool rc) {
  int sh = 0;
  if (base < 0) {
    sh = base < -16 ? 32 : static_cast<int>(-base) * 2;
    base = 0;
  }
  const unsigned off = static_cast<unsigned>(static_cast<u64>(base) * 2) & 63u;
  u64 x = r.w0 >> off;
  if (off) x |= r.w1 << (64 - off);
  u32 v = static_cast<u32>(x);
  v = sh >= 32 ? 0u : v << sh;
  if (rc) {  // reverse the 16 bases, complement
#if defined(__HIP_DEVICE_COMPILE__)
    v = __brev(v);
#else
    v = ((v & 0x0000FFFFu) << 16) | (v >> 16);
    v = ((v & 0x00FF00FFu) << 8) | ((v >> 8) & 0x00FF00FFu);
    v = ((v & 0x0F0F0F0Fu) << 4) | ((v >> 4) & 0x0F0F0F0Fu);
    v = ((v & 0x33333333u) << 2) | ((v >> 2) & 0x33333333u);
    v = ((v & 0x55555555u) << 1) | ((v >> 1) & 0x55555555u);
#endif
    v = ~(((v & 0x55555555u) << 1) | ((v >> 1) & 0x55555555u));  // the two bits of a base back in order
  }
  return v;
}
__host__ __device__ __forceinline__ u32 nw_text16(const u64* __restrict__ words, long long first, bool rc, int col0) {
  const long long base = nw_text16_base(first, rc, col0);
  return nw_text16_finish(nw_text16_load(words, base), base, rc);
}

// Bit planes of pattern block b (myers.h load_planes) with three unconditional loads: nothing waits on a loaded value to
// decide about the next load
__host__ __device__ __forceinline__ BlockPlanes nw_load_planes(const u64* __restrict__ words, u64 a_base, u32 n, u32 b) {
  const u64 row0 = static_cast<u64>(b) * 64;
  BlockPlanes p;
  if (row0 >= n) {
    p.lo = p.hi = p.valid = 0;
    return p;
  }
  const u64 bit = (a_base + row0) * 2;
  const u64 wi = bit >> 6;
  const unsigned off = static_cast<unsigned>(bit & 63);
  const u64 x0 = words[wi], x1 = words[wi + 1], x2 = words[wi + 2];
  const u64 w0 = off ? (x0 >> off) | (x1 << (64 - off)) : x0;
  const u64 w1 = off ? (x1 >> off) | (x2 << (64 - off)) : x1;
  const u32 valid = n - row0 >= 64 ? 64u : static_cast<u32>(n - row0);
  p.lo = compress_even(w0) | (compress_even(w1) << 32);
  p.hi = compress_even(w0 >> 1) | (compress_even(w1 >> 1) << 32);
  p.valid = valid >= 64 ? ~0ULL : ((1ULL << valid) - 1ULL);
  p.lo &= p.valid;
  p.hi &= p.valid;
  return p;
}

// The backward walk, resumable strip by strip.  Cells answers, for a mismatching cell (i, j) of the strip it holds (rows
// above row_lo, columns above seg_j0), whether the diagonal / the left neighbour is one below D(i, j).
template <class Cells>
struct NwWalkerT {
  Cells cells;
  // job
  u32 t_begin, q_begin, w, win0;
  NwWindowRec* recs;
  int seg_j0;   // the strip in `cells`: columns seg_j0 + 1 .. and rows row_lo + 1 ..
  int row_lo;
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

  __host__ __device__ void init(const NwJob& J, u32 distance, u32 w_, NwWindowRec* recs_all) {
    t_begin = J.t_begin;
    q_begin = J.q_begin;
    w = w_;
    win0 = J.t_begin / w_;
    recs = recs_all + J.bp_off;
    seg_j0 = 0;
    row_lo = 0;
    i = static_cast<int>(J.n);
    j = static_cast<int>(J.m);
    cur = distance;
    cw = 0xFFFFFFFFu;
    cw_lo = 0;
    have = false;
    first_t = first_q = last_t = last_q = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) gq[g] = 0xFFFFFFFFu;
    gx = -1;
    gt = 0;
  }

  __host__ __device__ void flush(bool write) {
    if (cw == 0xFFFFFFFFu || !write) return;
    NwWindowRec e;
    e.first_t = have ? first_t : 0xFFFFFFFFu;
    e.first_q = first_q;
    e.last_t = last_t;
    e.last_q = last_q;
#pragma unroll
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
    // the common case first: same window, and the next grid position lies further down
    if (cw != 0xFFFFFFFFu && t >= cw_lo && t - cw_lo < w && (gx < 0 || gt < t)) return;
    if (cw == 0xFFFFFFFFu || t < cw_lo || t - cw_lo >= w) {  // another window (the division only here)
      const u32 wi = t / w;
      flush(write);
      cw = wi;
      cw_lo = wi * w;
      have = false;
#pragma unroll
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
  }

  // walks while the current cell lies inside the strip (j > seg_j0, i > row_lo)
  __host__ __device__ void walk(bool write) {
    while (i > row_lo && j > seg_j0) {
      const int room = i - row_lo < j - seg_j0 ? i - row_lo : j - seg_j0;
      int r = cells.match_run(i, j, room);  // bases come from the strip: no memory access in the walk
      if (r > 0) {  // a match is always taken diagonally: the whole run at once, window by window
        const u32 t_hi = t_begin + static_cast<u32>(i - 1);
        const u32 t_lo = t_hi - static_cast<u32>(r - 1);
        if (have && cw != 0xFFFFFFFFu && t_lo >= cw_lo && t_hi - cw_lo < w && (gx < 0 || gt < t_lo)) {
          // the whole run inside the current window and above the next grid position: only the first pair moves
          first_t = t_lo;
          first_q = q_begin + static_cast<u32>(j - r);
          i -= r;
          j -= r;
          continue;
        }
        // bases from t_hi down to the start of its window
        const bool same = cw != 0xFFFFFFFFu && t_hi >= cw_lo && t_hi - cw_lo < w;
        const u32 in_window = same ? t_hi - cw_lo + 1 : t_hi - (t_hi / w) * w + 1;
        r = static_cast<u32>(r) < in_window ? r : static_cast<int>(in_window);
        take_diag_run(r, write);
      } else if (const int mv = cells.decide(i, j); mv == 0) {  // substitution: the diagonal
        --cur;
        take_diag(write);
      } else if (mv == 1) {  // 'I': read base only
        --j;
        --cur;
      } else {  // 'D': target base only
        on_target_base(t_begin + static_cast<u32>(i - 1), q_begin + static_cast<u32>(j), write);
        --i;
        --cur;
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

}  // namespace rvn
