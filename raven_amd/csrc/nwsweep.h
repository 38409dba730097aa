// nwsweep.h — one lane of the forward sweep of the alignment-path stage (see nwpath.h for the whole picture; the
// alignment itself is racon's edlibAlign(NW, TASK_PATH) per overlap, reached from RavenLib/src/polish.cc:51).
//
// The lane holds one super-block (R consecutive 64-row blocks) of the band at a time, entirely in registers, and steps
// through its columns; at step t it works on column t - s (s = index of its super-block), takes the horizontal delta of
// the super-block above from the previous lane of the ring (that lane was at the same column one step earlier) and
// passes its own on.  The loop around it (kernel: nwpath.hip; CPU stepper: the same file) does, per step,
//     x  = xf of the ring's previous lane            (one ds_bpermute)
//     if any lane has an event at t: event(...)      (super-block retires / next one enters: rare, ~once per 65 steps)
//     step(t, x)                                     (the Myers update of the R blocks: the hot path)
// and every 16 steps stores the lanes' hs words, every 32 steps their (Pv, Mv) — all lanes at the same step, so the
// stores are coalesced.  Lanes that hold no super-block at a step (before their first one enters, between two, after the
// last) run the same instructions on dead state: nobody reads what they produce, so the hot path carries no activity
// predicate at all — only `fed` (does the lane above still deliver?) is tested.
//
// Match masks: the four Peq words of each of the lane's blocks sit in LDS ([block][symbol][lane], written when the
// super-block enters); the symbol of the NEXT column is known one step ahead (a 16-column window of the read is held in
// a register and refilled every 16 steps straight from the packed read), so the LDS read of the next step's masks is
// issued a whole step early and never waits.
#pragma once

#include "nwpath.h"

namespace rvn {

constexpr int kNwNever = 0x7FFFFFFF;

template <int R, int LANES>
struct NwSweepLane {
  // job
  const u64* a_words;  // target words of the job
  u64 a_base;          // first row's base index
  const u64* b_words;  // read words of the job
  long long b_first;   // base index of column 1 (forward) / of column 1 in the reverse-complemented direction
  bool rc;
  NwGeo g;
  int lig;        // position in the ring (lanes >= g.L never work)
  u64* peq;       // LDS: element [(r * 4 + c) * LANES + lane]
  int lane;       // lane index inside `peq`
  // state
  int s;          // super-block held (or waited for); >= n_super: none left
  bool active;
  int t_evt;      // step of the next event: entry of s (waiting) or the step after its last column (active)
  int t_fed;      // last step at which the lane above delivers the horizontal delta
  u64 Pv[R], Mv[R];
  int sc;         // score at the bottom row of the lane's last block
  u64 eq[R];      // match masks of the column of the NEXT step (prefetched)
  u32 acc[R];     // horizontal deltas out of each block, the 16 steps of the current hs word
  u32 w_cur, w_nxt;  // text symbols of the current / next group of 16 steps
  NwRaw raw;         // words of the group after that, in flight
  int xf;         // what the next lane of the ring reads one step later (2 bits)
  u32 result;     // D(n, m) + 1 on the lane that retired the last super-block

  __host__ __device__ void init(const NwJob& J, const u64* t_words, const u64* r_words, const NwGeo& geo, int lig_,
                                u64* peq_, int lane_) {
    a_words = t_words + J.t_word;
    a_base = J.t_begin;
    b_words = r_words + J.r_word;
    rc = J.rc != 0;
    const long long b_base = rc ? static_cast<long long>(J.r_len) - J.q_begin - J.m : static_cast<long long>(J.q_begin);
    b_first = rc ? b_base + static_cast<long long>(J.m) - 1 : b_base;
    g = geo;
    lig = lig_;
    peq = peq_;
    lane = lane_;
    s = lig_ < geo.L ? lig_ : geo.n_super;
    active = false;
    t_evt = (s < g.n_super && g.ja(s) <= g.m) ? g.ja(s) + s : kNwNever;
    t_fed = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      Pv[r] = ~0ULL;
      Mv[r] = 0;
      eq[r] = 0;
      acc[r] = 0;
    }
    sc = 0;
    w_cur = w_nxt = 0;
    raw = NwRaw{0, 0};
    xf = 1;
    result = 0;
  }
  // a lane without a job: runs along, never has an event
  __host__ __device__ void init_idle(u64* peq_, int lane_) {
    a_words = b_words = nullptr;
    a_base = 0;
    b_first = 0;
    rc = false;
    g = NwGeo{0, 0, 0, 0, R, 0, 0, 0, 0};
    lig = 0;
    peq = peq_;
    lane = lane_;
    s = 0;
    active = false;
    t_evt = kNwNever;
    t_fed = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      Pv[r] = ~0ULL;
      Mv[r] = 0;
      eq[r] = 0;
      acc[r] = 0;
    }
    sc = 0;
    w_cur = w_nxt = 0;
    raw = NwRaw{0, 0};
    xf = 1;
    result = 0;
  }

  // text symbols of the 16 steps of group gi (steps t = 16 gi + 1 .. 16 gi + 16): columns 16 gi + 1 - s ..
  __host__ __device__ u32 window(int gi) const {
    const int col0 = kNwHsSteps * gi + 1 - s;
    if (col0 > g.m || b_words == nullptr) return 0;  // nothing of the read there (and nothing beyond it to load)
    return nw_text16(b_words, b_first, rc, col0);
  }
  // the same in two halves: loads now, symbols one group of steps later
  __host__ __device__ void window_load(int gi) {
    const int col0 = kNwHsSteps * gi + 1 - s;
    if (col0 > g.m || b_words == nullptr) return;
    raw = nw_text16_load(b_words, nw_text16_base(b_first, rc, col0));
  }
  __host__ __device__ u32 window_finish(int gi) const {
    const int col0 = kNwHsSteps * gi + 1 - s;
    if (col0 > g.m || b_words == nullptr) return 0;
    return nw_text16_finish(raw, nw_text16_base(b_first, rc, col0), rc);
  }
  __host__ __device__ void fetch_eq(unsigned c) {
#pragma unroll
    for (int r = 0; r < R; ++r) eq[r] = peq[(r * 4 + static_cast<int>(c)) * LANES + lane];
  }

  __host__ __device__ bool has_event(int t) const { return t == t_evt; }

  // x_prev / sc_prev: xf and sc of the ring's previous lane after step t - 1
  __host__ __device__ void event(int t, int x_prev, int sc_prev) {
    if (active && t == t_evt) {  // the super-block's last column was step t - 1
      if (s == g.n_super - 1) {
        // D(n, m) = bottom score minus the vertical deltas of the padded rows below row n
        int v = sc;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int row0 = 64 * (s * R + r);
          const int used = g.n > row0 ? (g.n - row0 >= 64 ? 64 : g.n - row0) : 0;
          const u64 padmask = used >= 64 ? 0ULL : ~((1ULL << used) - 1ULL);
          v -= static_cast<int>(RVN_POPC64(Pv[r] & padmask)) - static_cast<int>(RVN_POPC64(Mv[r] & padmask));
        }
        result = static_cast<u32>(v) + 1u;
      }
      s += g.L;
      active = false;
      t_evt = (s < g.n_super && g.ja(s) <= g.m) ? g.ja(s) + s : kNwNever;
    }
    if (!active && t == t_evt) {  // super-block s enters the band at column t - s
      const int ja = t - s;
      // edlib's rule for a block entering the band: +1 down every row from the block above (an upper bound)
      sc = ja == 1 ? 64 * (s * R + R) : sc_prev - nw_delta(x_prev) + 64 * R;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        Pv[r] = ~0ULL;
        Mv[r] = 0;
        const BlockPlanes pl = nw_load_planes(a_words, a_base, static_cast<u32>(g.n), static_cast<u32>(s * R + r));
#pragma unroll
        for (int c = 0; c < 4; ++c) peq[(r * 4 + c) * LANES + lane] = planes_eq(pl, static_cast<unsigned>(c));
      }
      t_fed = g.jfed(s) + (s > 0 ? s : 0);
      t_evt = g.je(s) + s + 1;
      active = true;
      const int u = t - 1;
      w_cur = window(u >> 4);
      w_nxt = window((u >> 4) + 1);
      window_load((u >> 4) + 2);
      fetch_eq((w_cur >> (2 * (u & 15))) & 3u);
    }
  }

  // the Myers update of the lane's blocks at step t (column t - s); eq[] holds this column's masks
  __host__ __device__ void step(int t, int x_prev) {
    const int u = t - 1, pos = u & 15;
    int h = t <= t_fed ? x_prev : 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      h = myers_block2(Pv[r], Mv[r], eq[r], h);
      acc[r] |= static_cast<u32>(h) << (2 * pos);
    }
    sc += nw_delta(h);
    xf = h;
    // masks of the next step's column
    const unsigned c = (pos == 15 ? w_nxt : (w_cur >> (2 * (pos + 1)))) & 3u;
    fetch_eq(c);
  }
  // after the last step of a group of 16 (the caller has stored acc[]): next text window
  __host__ __device__ void next_group(int t) {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0;
    w_cur = w_nxt;
    w_nxt = window_finish((t >> 4) + 1);  // t = 16 (gi + 1): the group after the one that starts now; loaded 16 steps ago
    window_load((t >> 4) + 2);
  }
};

}  // namespace rvn
