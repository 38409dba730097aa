// nwtrace.h — the backward walk of the alignment-path stage, ONE lane per alignment (see nwpath.h; the path is
// racon's edlib NW path, RavenLib/src/polish.cc:51 -> Overlap::find_breaking_points).
//
// The sweep (nwsweep.h) left, per alignment, the horizontal delta out of every block at every step (hs, 2 bits) and
// (Pv, Mv) of every block every 32 steps (ck).  The walker stands in one 64-row block at a time; that block alone is
// recomputed from its last checkpoint at or before the walker's column (at most 32 columns; its horizontal input is the
// hs stream of the block above, its own vertical state the checkpoint), every column's (Pv, Mv) kept in a small strip
// (LDS on the device: [column][lane]), and walked until the walker leaves the block upwards or the strip to the left.
// Walk decisions at a mismatching cell (i, j), bit p of the block, need differences only:
//     dh(i-1, j) = hin(j) + sum_{q < p} dv(q, j) - sum_{q < p} dv(q, j-1)        (hin = the block's horizontal input)
//     D(i-1, j-1) + 1 == D(i, j)   <=>   dv(p, j) + dh(i-1, j) == 1               (substitution: the diagonal)
//     D(i,   j-1) + 1 == D(i, j)   <=>   dh(i-1, j) + dv(p, j) - dv(p, j-1) == 1   (read base only)
// An entering block's first column takes the all-(+1) column edlib assumes before it as column j - 1, a block whose
// upper neighbour has left the band takes hin = +1: both are upper bounds, and an upper bound passes a test only if the
// true value does (nwpath.h).
#pragma once

#include "nwpath.h"

namespace rvn {

constexpr int kNwStripCols = kNwCkSteps + 1;  // checkpoint column + up to 32 recomputed ones

// (Pv, Mv) of the strip's columns.  LANES = 64 on the device (LDS, element [column * 64 + lane]); 1 on the host.
template <int LANES>
struct NwStripMem {
  u64* pv;
  u64* mv;
  int lane;
  __host__ __device__ int at(int col) const { return col * LANES + lane; }
};

template <int LANES>
struct NwStripCells {
  NwStripMem<LANES> mem;
  int j0;     // column of strip entry 0
  u64 hinw;   // horizontal input of the block at columns j0 + 1 .. j0 + 32, 2 bits each
  u64 tlo, thi;  // bit planes of the block's 64 target bases (bit p = row p of the block)
  u32 rlo, rhi;  // bit planes of the read bases of columns j0 + 1 .. j0 + 32 (bit x = column j0 + 1 + x)
  // number of matches going up the diagonal from (i, j), at most lim (<= rows left in the block, <= columns left in
  // the strip): both sequences are in registers, 32 bases are compared at once
  __host__ __device__ int match_run(int i, int j, int lim) const {
    const int p = (i - 1) & 63, x0 = j - j0 - 1;
    const int sh = p - x0;
    const u32 a_lo = sh >= 0 ? static_cast<u32>(tlo >> sh) : static_cast<u32>(tlo << -sh);
    const u32 a_hi = sh >= 0 ? static_cast<u32>(thi >> sh) : static_cast<u32>(thi << -sh);
    const u32 below = x0 >= 31 ? 0xFFFFFFFFu : ((2u << x0) - 1u);  // columns j0 + 1 .. j
    const u32 mism = ((a_lo ^ rlo) | (a_hi ^ rhi)) & below;
#if defined(__HIP_DEVICE_COMPILE__)
    const int run = mism ? x0 - (31 - __clz(static_cast<int>(mism))) : x0 + 1;
#else
    const int run = mism ? x0 - (31 - __builtin_clz(mism)) : x0 + 1;
#endif
    return run < lim ? run : lim;
  }
  // dh(i-1, j) and the two vertical deltas at (i, j) / (i, j-1)
  __host__ __device__ void deltas(int i, int j, int* a, int* dvj, int* dv1) const {
    const int c = j - j0;
    const unsigned p = static_cast<unsigned>((i - 1) & 63);
    const u64 pvj = mem.pv[mem.at(c)], mvj = mem.mv[mem.at(c)];
    const u64 pv1 = mem.pv[mem.at(c - 1)], mv1 = mem.mv[mem.at(c - 1)];
    const u64 lm = (1ULL << p) - 1ULL;  // rows of the block above row i
    const int hin = nw_delta(static_cast<int>(hinw >> (2 * (c - 1))) & 3);
    *a = hin + static_cast<int>(RVN_POPC64(pvj & lm)) - static_cast<int>(RVN_POPC64(mvj & lm)) -
         static_cast<int>(RVN_POPC64(pv1 & lm)) + static_cast<int>(RVN_POPC64(mv1 & lm));
    *dvj = static_cast<int>((pvj >> p) & 1ULL) - static_cast<int>((mvj >> p) & 1ULL);
    *dv1 = static_cast<int>((pv1 >> p) & 1ULL) - static_cast<int>((mv1 >> p) & 1ULL);
  }
  // which neighbour the path takes from a mismatching cell: 0 the diagonal, 1 the left one (read base only), 2 the
  // upper one (target base only) — racon's order
  __host__ __device__ int decide(int i, int j) const {
    int a, dvj, dv1;
    deltas(i, j, &a, &dvj, &dv1);
    return dvj + a == 1 ? 0 : (a + dvj - dv1 == 1 ? 1 : 2);
  }
};

// 2 x 32 bits of the hs stream of (ring lane `lane`, block r of the lane) for the steps u_first .. u_first + 31 (0-based),
// the bits of step u_first in bits 0-1
__host__ __device__ inline u64 nw_hs_bits(const u32* __restrict__ hs, const NwGeo& g, int lane, int r, int u_first) {
  const u64 stride = static_cast<u64>(g.L) * static_cast<u64>(g.R);
  const u64 g0 = static_cast<u64>(u_first) >> 4;
  const u32* p = hs + g0 * stride + static_cast<u64>(lane) * static_cast<u64>(g.R) + static_cast<u64>(r);
  const unsigned sh = 2u * (static_cast<unsigned>(u_first) & 15u);
  const u32 x0 = p[0], x1 = p[stride], x2 = p[2 * stride];  // all three at once (the buffer has slack behind the last job)
  const u64 lo = static_cast<u64>(x0) | (static_cast<u64>(x1) << 32);
  u64 v = lo >> sh;
  if (sh) v |= static_cast<u64>(x2) << (64 - sh);
  return v;
}

// The whole walk of one alignment.  hs / ck: the job's regions.  Returns 0 (records written) or 1 (inconsistent).
// SC = columns of a strip that are KEPT (32: the whole checkpoint interval; 16: only the last sixteen columns up to the walker's
// — a walker that leaves them to the left comes back for the rest of the interval, which is then at most sixteen columns —:
// half the strip memory, i.e. twice the resident walkers per CU where the strips live in LDS, for 1.5x the Myers steps, which
// are not what a wave of 64 diverged walkers is waiting for).
template <int LANES, int SC = kNwCkSteps>
__host__ __device__ inline int nw_trace_job(const NwJob& J, const NwGeo& g, const u64* __restrict__ t_words_all,
                                            const u64* __restrict__ r_words_all, const u32* __restrict__ hs,
                                            const NwPm* __restrict__ ck, const NwStripMem<LANES>& mem, u32 distance, u32 w,
                                            NwWindowRec* __restrict__ recs_all) {
  const u64* tw = t_words_all + J.t_word;
  const u64* rw = r_words_all + J.r_word;
  const bool rc = J.rc != 0;
  const long long b_base = rc ? static_cast<long long>(J.r_len) - J.q_begin - J.m : static_cast<long long>(J.q_begin);
  const long long b_first = rc ? b_base + static_cast<long long>(J.m) - 1 : b_base;
  NwWalkerT<NwStripCells<LANES>> wk;
  wk.cells.mem = mem;
  wk.init(J, distance, w, recs_all);
  const int R = g.R, L = g.L;
  while (wk.i > 0 && wk.j > 0) {
    const int b = (wk.i - 1) >> 6;
    const int s = b / R, r = b - s * R, p = s % L;
    const int j = wk.j;
    const int ja = g.ja(s), je = g.je(s);
    if (s >= g.n_super || j < ja || j > je) return 1;  // the walk left the band: cannot happen for a result <= k
    // the block's state at the last checkpoint at or before column j - 1 (or the column before it entered the band)
    const int u = j - 1 + s;            // 0-based step of column j
    const int q = (u >> 5) - 1;         // checkpoint q = state after step 32 q + 31 = column 32 (q + 1) - s
    const int jc = 32 * (q + 1) - s;
    u64 pv, mv;
    int j0;
    if (q >= 0 && jc >= ja) {
      j0 = jc;
      const NwPm v = ck[(static_cast<u64>(q) * static_cast<u64>(L) + static_cast<u64>(p)) * static_cast<u64>(R) + static_cast<u64>(r)];
      pv = v.pv;
      mv = v.mv;
    } else {
      j0 = ja - 1;
      pv = ~0ULL;
      mv = 0;
    }
    const int len = j - j0;  // 1 .. 32 columns to recompute
    // horizontal input at columns j0 + 1 ..: the block above — same lane (r > 0) or the ring's previous lane one step earlier
    u64 hinw;
    if (r > 0) {
      hinw = nw_hs_bits(hs, g, p, r - 1, j0 + s);
    } else if (s > 0 && j0 + 1 <= g.jfed(s)) {
      hinw = nw_hs_bits(hs, g, p == 0 ? L - 1 : p - 1, R - 1, j0 + s - 1);
      const int nfed = g.jfed(s) - j0;  // columns that are still fed
      if (nfed < 32) hinw = (hinw & ((1ULL << (2 * nfed)) - 1ULL)) | (0x5555555555555555ULL << (2 * nfed));
    } else {
      hinw = 0x5555555555555555ULL;  // +1 everywhere: the matrix border or a retired block above
    }
    const BlockPlanes pl = nw_load_planes(tw, J.t_begin, static_cast<u32>(g.n), static_cast<u32>(b));
    u64 text = static_cast<u64>(nw_text16(rw, b_first, rc, j0 + 1));
    if (len > 16) text |= static_cast<u64>(nw_text16(rw, b_first, rc, j0 + 17)) << 32;
    const int skip = len > SC ? len - SC : 0;  // columns recomputed on the way, not kept
    int c = 1;
    for (; c <= skip; ++c) {
      const unsigned sym = static_cast<unsigned>(text >> (2 * (c - 1))) & 3u;
      (void)myers_block2(pv, mv, planes_eq(pl, sym), static_cast<int>(hinw >> (2 * (c - 1))) & 3);
    }
    mem.pv[mem.at(0)] = pv;
    mem.mv[mem.at(0)] = mv;
    for (; c <= len; ++c) {
      const unsigned sym = static_cast<unsigned>(text >> (2 * (c - 1))) & 3u;
      (void)myers_block2(pv, mv, planes_eq(pl, sym), static_cast<int>(hinw >> (2 * (c - 1))) & 3);
      mem.pv[mem.at(c - skip)] = pv;
      mem.mv[mem.at(c - skip)] = mv;
    }
    const u64 text_s = text >> (2 * skip);
    wk.cells.j0 = j0 + skip;
    wk.cells.hinw = hinw >> (2 * skip);
    wk.cells.tlo = pl.lo;
    wk.cells.thi = pl.hi;
    wk.cells.rlo = static_cast<u32>(compress_even(text_s));
    wk.cells.rhi = static_cast<u32>(compress_even(text_s >> 1));
    wk.seg_j0 = j0 + skip;
    wk.row_lo = 64 * b;
    wk.walk(true);
  }
  return wk.finish(true);
}

// ---- the same walk by a GROUP of lanes per alignment (round 6) -----------------------------------------------------------
// A lane per alignment pays ~1 us per column: of a walker's turn through one strip three quarters are the strip's
// recomputation (up to 32 dependent Myers steps), the rest the walk, and 64 walkers of a wave are in 64 different branches
// of it.  The longest alignments of a round (a few thousand, tens of kilobases each) therefore take tens of milliseconds
// whatever the machine's width, and that is the floor of a small batch or of one rank's share of a sharded round.
// Here GL lanes own ONE alignment.  Which strips the walker is going to need is predictable — the path runs along the
// straight line from its present cell to the origin, give or take a few rows —, and a strip (block b, checkpoint q) is
// computable from the sweep's checkpoints and hs stream ALONE.  So the group works in batches:
//   plan   every lane follows the predicted line through the (block, checkpoint) grid and keeps the t-th strip it meets
//          (both candidates where the line passes within a row of a strip's upper left corner);
//          entry 0 is the strip of the walker's present cell by construction;
//   fill   lane t recomputes strip t, the WHOLE of it (all columns of the checkpoint interval), into its column of the
//          wave's strip memory, and leaves a 48-byte head (key, first column, horizontal input, bases) beside it;
//   walk   every lane of the group steps the same walker through the same strips (same values in every lane: no
//          divergence inside a group, LDS reads are broadcasts; lane 0 writes the records).  A strip is left for good —
//          upwards or to the left —, so the batch is searched forwards only; a cell whose strip is not among the planned
//          ones (the path strayed further than the margin) ends the batch and the next plan starts from that cell.
// A wrong prediction costs a batch that serves fewer strips, never a result: every strip the walker uses is the one its
// present cell asks for, computed as nw_trace_job computes it.
constexpr int kNwGroupMax = 64;

struct NwStripHead {
  int b, q;    // key: 64-row block, checkpoint index of the strip's first column (-1 .. ; b < 0: not a strip of the band)
  int j0;      // column of strip entry 0
  int len;     // columns recomputed (1 .. 32)
  u64 hinw, tlo, thi;
  u32 rlo, rhi;
};
static_assert(sizeof(NwStripHead) == 48, "NwStripHead layout");

struct NwStripKey {
  int b, q;
  __host__ __device__ bool operator==(const NwStripKey& o) const { return b == o.b && q == o.q; }
};
__host__ __device__ __forceinline__ NwStripKey nw_strip_key(const NwGeo& g, int i, int j) {
  const int b = (i - 1) >> 6;
  const int s = b / g.R;
  return NwStripKey{b, ((j - 1 + s) >> 5) - 1};
}
// first column - 1 of strip (b, q): its checkpoint column, or the column before the super-block enters the band
__host__ __device__ __forceinline__ int nw_strip_j0(const NwGeo& g, int s, int q) {
  const int jc = 32 * (q + 1) - s, ja = g.ja(s);
  return (q >= 0 && jc >= ja) ? jc : ja - 1;
}

// The key of the t-th strip along the predicted path from cell (i, j) (t = 0: the strip of that cell); returns the number
// of strips planned (<= GL), *mine = the t-th of them if t is below that number.  Every lane runs the same loop.
template <int GL>
__host__ __device__ inline int nw_plan_strips(const NwGeo& g, int i, int j, int t, NwStripKey* mine) {
  const float slope = static_cast<float>(i) / static_cast<float>(j);  // rows per column
  const float inv = static_cast<float>(j) / static_cast<float>(i);
  NwStripKey k1{-1, 0}, k2{-1, 0}, k3{-1, 0};  // the last three keys planned (a corner names a strip twice)
  int n = 0;
  int ci = i, cj = j;
  auto emit = [&](const NwStripKey& k) {
    if (k == k1 || k == k2 || k == k3 || n >= GL) return;
    if (n == t) *mine = k;
    ++n;
    k3 = k2;
    k2 = k1;
    k1 = k;
  };
  while (n < GL && ci > 0 && cj > 0) {
    const NwStripKey k = nw_strip_key(g, ci, cj);
    emit(k);
    const int s = k.b / g.R;
    const int j0 = nw_strip_j0(g, s, k.q), row_lo = 64 * k.b;
    const int dcl = cj - j0, dru = ci - row_lo;  // columns to the strip's left edge, rows to the block's upper edge (>= 1)
    const bool can_left = j0 >= g.ja(s) && j0 >= 1, can_up = row_lo >= 1;
    if (!can_left && !can_up) break;
    const int margin = 1;  // (rows; measured on the host stepper: 0 .. 12 and growing along the batch — 1 serves the most columns per batch)
    const int rr = static_cast<int>(static_cast<float>(dcl) * slope + 0.5f);  // rows the line climbs until the left edge
    if (can_left && (!can_up || rr + margin < dru)) {  // out through the left edge
      ci -= rr < dru ? rr : dru - 1;
      cj = j0;
    } else if (can_up && (!can_left || rr > dru + margin)) {  // out through the upper edge
      int cc = static_cast<int>(static_cast<float>(dru) * inv + 0.5f);
      cc = cc < dcl ? cc : dcl - 1;
      ci = row_lo;
      cj -= cc;
    } else {  // past the corner, on either side: the strip left of this one, the one above it, then on from the corner
      emit(nw_strip_key(g, row_lo + 1, j0));
      emit(nw_strip_key(g, row_lo, j0 + 1));
      ci = row_lo;
      cj = j0;
    }
  }
  return n;
}

// Strip `key` of the job, all its columns, into column mem.lane of the strip memory + its head.  head.b = -1: the key names
// no strip of this band (the prediction left it).
template <int LANES>
__host__ __device__ inline void nw_fill_strip(const NwJob& J, const NwGeo& g, const u64* __restrict__ tw,
                                              const u64* __restrict__ rw, long long b_first, bool rc,
                                              const u32* __restrict__ hs, const NwPm* __restrict__ ck, const NwStripKey& key,
                                              const NwStripMem<LANES>& mem, NwStripHead* head) {
  const int R = g.R, L = g.L;
  const int b = key.b, q = key.q;
  NwStripHead h;
  h.b = -1;
  h.q = q;
  h.j0 = 0;
  h.len = 0;
  h.hinw = h.tlo = h.thi = 0;
  h.rlo = h.rhi = 0;
  const int s = b >= 0 ? b / R : 0, r = b - s * R, p = s % L;
  const int ja = g.ja(s), je = g.je(s);
  const int jc = 32 * (q + 1) - s;
  const bool from_ck = q >= 0 && jc >= ja;  // (as nw_trace_job: the checkpoint, or the column before the block enters the band)
  const int j0 = from_ck ? jc : ja - 1;
  int jr = jc + 32;
  jr = jr < je ? jr : je;
  if (b < 0 || b >= g.nb || s >= g.n_super || q < -1 || jr <= j0) {
    *head = h;
    return;
  }
  u64 pv, mv;
  if (from_ck) {
    const NwPm v = ck[(static_cast<u64>(q) * static_cast<u64>(L) + static_cast<u64>(p)) * static_cast<u64>(R) + static_cast<u64>(r)];
    pv = v.pv;
    mv = v.mv;
  } else {
    pv = ~0ULL;
    mv = 0;
  }
  const int len = jr - j0;  // 1 .. 32
  u64 hinw;
  if (r > 0) {
    hinw = nw_hs_bits(hs, g, p, r - 1, j0 + s);
  } else if (s > 0 && j0 + 1 <= g.jfed(s)) {
    hinw = nw_hs_bits(hs, g, p == 0 ? L - 1 : p - 1, R - 1, j0 + s - 1);
    const int nfed = g.jfed(s) - j0;
    if (nfed < 32) hinw = (hinw & ((1ULL << (2 * nfed)) - 1ULL)) | (0x5555555555555555ULL << (2 * nfed));
  } else {
    hinw = 0x5555555555555555ULL;
  }
  const BlockPlanes pl = nw_load_planes(tw, J.t_begin, static_cast<u32>(g.n), static_cast<u32>(b));
  u64 text = static_cast<u64>(nw_text16(rw, b_first, rc, j0 + 1));
  if (len > 16) text |= static_cast<u64>(nw_text16(rw, b_first, rc, j0 + 17)) << 32;
  mem.pv[mem.at(0)] = pv;
  mem.mv[mem.at(0)] = mv;
  for (int c = 1; c <= len; ++c) {
    const unsigned sym = static_cast<unsigned>(text >> (2 * (c - 1))) & 3u;
    (void)myers_block2(pv, mv, planes_eq(pl, sym), static_cast<int>(hinw >> (2 * (c - 1))) & 3);
    mem.pv[mem.at(c)] = pv;
    mem.mv[mem.at(c)] = mv;
  }
  h.b = b;
  h.j0 = j0;
  h.len = len;
  h.hinw = hinw;
  h.tlo = pl.lo;
  h.thi = pl.hi;
  h.rlo = static_cast<u32>(compress_even(text));
  h.rhi = static_cast<u32>(compress_even(text >> 1));
  *head = h;
}

// The group's walk.  mem: the strip memory with mem.lane = the group's FIRST column (lane t of the group fills column
// mem.lane + t); heads: the group's GL heads; t: this lane's index in the group; Sync: makes the strips and heads written by
// the group's lanes visible to all of them (one wave: a fence; the host stepper calls the phases lane by lane instead, see
// nw_trace_group_host).  Returns 0 (records written by lane 0) or 1 (inconsistent), the same in every lane of the group.
template <int LANES, int GL>
struct NwGroupWalk {
  NwWalkerT<NwStripCells<LANES>> wk;
  NwStripMem<LANES> mem;
  NwStripHead* heads;
  const u64 *tw, *rw;
  long long b_first;
  bool rc;
  int n_plan, cur;

  __host__ __device__ void init(const NwJob& J, const u64* __restrict__ t_words_all, const u64* __restrict__ r_words_all,
                                const NwStripMem<LANES>& mem_, NwStripHead* heads_, u32 distance, u32 w, NwWindowRec* recs_all) {
    tw = t_words_all + J.t_word;
    rw = r_words_all + J.r_word;
    rc = J.rc != 0;
    const long long b_base = rc ? static_cast<long long>(J.r_len) - J.q_begin - J.m : static_cast<long long>(J.q_begin);
    b_first = rc ? b_base + static_cast<long long>(J.m) - 1 : b_base;
    mem = mem_;
    heads = heads_;
    wk.cells.mem = mem_;
    wk.init(J, distance, w, recs_all);
    n_plan = 0;
    cur = 0;
  }
  __host__ __device__ bool done() const { return !(wk.i > 0 && wk.j > 0); }
  // plan + fill of lane t (call for every lane of the group, then make the writes visible, then walk_batch)
  __host__ __device__ void fill(const NwJob& J, const NwGeo& g, const u32* __restrict__ hs, const NwPm* __restrict__ ck, int t) {
    NwStripKey key{-1, 0};
    n_plan = nw_plan_strips<GL>(g, wk.i, wk.j, t, &key);
    cur = 0;
    NwStripMem<LANES> m = mem;
    m.lane = mem.lane + t;
    if (t < n_plan) nw_fill_strip<LANES>(J, g, tw, rw, b_first, rc, hs, ck, key, m, heads + t);
  }
  // walks through the planned strips as far as they serve; 1: the walk left the band (cannot happen for a result <= k)
  __host__ __device__ int walk_batch(const NwGeo& g, bool write) {
    while (wk.i > 0 && wk.j > 0) {
      const NwStripKey key = nw_strip_key(g, wk.i, wk.j);
      const int s = key.b / g.R;
      if (s >= g.n_super || wk.j < g.ja(s) || wk.j > g.je(s)) return 1;
      int t = cur;
      while (t < n_plan && !(heads[t].b == key.b && heads[t].q == key.q)) ++t;
      if (t >= n_plan) return 0;  // not planned: the next batch starts here
      const NwStripHead h = heads[t];
      if (wk.j > h.j0 + h.len) return 1;  // (a strip holds every column of its checkpoint interval inside the band)
      wk.cells.mem.lane = mem.lane + t;
      wk.cells.j0 = h.j0;
      wk.cells.hinw = h.hinw;
      wk.cells.tlo = h.tlo;
      wk.cells.thi = h.thi;
      wk.cells.rlo = h.rlo;
      wk.cells.rhi = h.rhi;
      wk.seg_j0 = h.j0;
      wk.row_lo = 64 * key.b;
      wk.walk(write);
      cur = t + 1;
    }
    return 0;
  }
};

// host stepper of the group walk (LANES = GL columns of strip memory): the phases lane by lane
template <int GL>
inline int nw_trace_group_host(const NwJob& J, const NwGeo& g, const u64* t_words_all, const u64* r_words_all, const u32* hs,
                               const NwPm* ck, u32 distance, u32 w, NwWindowRec* recs_all, u64* n_batches) {
  u64 pv[kNwStripCols * GL], mv[kNwStripCols * GL];
  NwStripHead heads[GL];
  NwGroupWalk<GL, GL> G;
  G.init(J, t_words_all, r_words_all, NwStripMem<GL>{pv, mv, 0}, heads, distance, w, recs_all);
  while (!G.done()) {
    for (int t = 0; t < GL; ++t) G.fill(J, g, hs, ck, t);
    if (n_batches) ++*n_batches;
    if (G.walk_batch(g, true)) return 1;
  }
  return G.wk.finish(true);
}

}  // namespace rvn
