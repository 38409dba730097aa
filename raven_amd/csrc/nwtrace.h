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
template <int LANES>
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
    mem.pv[mem.at(0)] = pv;
    mem.mv[mem.at(0)] = mv;
    for (int c = 1; c <= len; ++c) {
      const unsigned sym = static_cast<unsigned>(text >> (2 * (c - 1))) & 3u;
      (void)myers_block2(pv, mv, planes_eq(pl, sym), static_cast<int>(hinw >> (2 * (c - 1))) & 3);
      mem.pv[mem.at(c)] = pv;
      mem.mv[mem.at(c)] = mv;
    }
    wk.cells.j0 = j0;
    wk.cells.hinw = hinw;
    wk.cells.tlo = pl.lo;
    wk.cells.thi = pl.hi;
    wk.cells.rlo = static_cast<u32>(compress_even(text));
    wk.cells.rhi = static_cast<u32>(compress_even(text >> 1));
    wk.seg_j0 = j0;
    wk.row_lo = 64 * b;
    wk.walk(true);
  }
  return wk.finish(true);
}

}  // namespace rvn
