// nwlane.h — the alignment-path stage (nwpath.h) for NARROW bands, one LANE per alignment.
//
// The wave-per-alignment formulation of nwpath.h maps the band's 64-row blocks to lanes: a 10 kb ONT read against its
// target has a band of ~20 blocks, so two thirds of the wave idle, and a HiFi read (3 blocks) uses 1 lane in 20.  Here a
// lane owns a whole alignment and walks its band block by block, 64 alignments per wave: the band state — (Pv, Mv, bottom
// score) and the two bit planes of the pattern for every block inside the band — lives in LDS as [block slot][lane]
// (lane-consecutive 8-byte words: conflict-free), a ring of NB slots indexed by block % NB.  Same algorithm, same
// checkpoints, same walk, same results as nwpath.h; the wave-per-alignment kernel keeps the wide bands (> 32 blocks).
//   pass 1     sweep over all columns, checkpoint of the band every kNwSeg columns, threshold doubled until exact
//   segments   re-sweep of a segment from its checkpoint into a global scratch laid out [column][block][lane] (the 64
//              lanes of a wave run in lockstep over (column, block), so a store instruction writes 64 consecutive
//              entries), then the lane walks its own path through it
// Everything is __host__ __device__: the CPU stepper (rvn_test_nw_breakpoints with force_r < 0) runs nw_lane_job on
// plain arrays.
#pragma once

#include "nwpath.h"

namespace rvn {

// Band state of one alignment.  LANES = 64 on the device (arrays in LDS, element [slot * 64 + lane]); 1 on the host.
template <int NB, int LANES>
struct NwLaneMem {
  u64* pv;
  u64* mv;
  u64* plo;
  u64* phi;
  int* sc;
  int lane;
  __host__ __device__ int at(int slot) const { return slot * LANES + lane; }
};

struct NwLaneStore {
  NwPm* ck_pm;  // checkpoints of the job: [column / kNwSeg][block - bfirst(column)], stride ckpt_nb
  int* ck_sc;
  u32 ckpt_nb;
  NwPm* seg_pm;  // segment scratch of the wave: [(column - j0 - 1) * NB + (block - bfirst(column))][lane]
  int* seg_sc;
};

// One sweep over columns (j0, j_end] of the band; returns D(n, m) + 1 if column m was reached, else 0.
// mode 0: pass 1 (checkpoints), mode 1: segment (every block update to the scratch).
template <int NB, int LANES>
__host__ __device__ inline u32 nw_lane_sweep(const NwJob& J, const u64* __restrict__ tw, const u64* __restrict__ rw,
                                             const NwBand& B, const NwLaneMem<NB, LANES>& M, const NwLaneStore& st, int j0,
                                             int j_end, int mode) {
  const u32 n = J.n, m = J.m;
  const bool rc = J.rc != 0;
  const u64 b_base = rc ? static_cast<u64>(J.r_len) - J.q_begin - J.m : J.q_begin;
  int top;  // last block whose state is in the ring
  if (j0 == 0) {
    top = -1;
  } else {  // resume: every block inside the band at column j0 comes from the checkpoint
    const int bf = nw_bfirst(j0, B.lo), bl = nw_blast(j0, B.hi, B.nb);
    for (int b = bf; b <= bl; ++b) {
      const u64 cs = static_cast<u64>(j0 / kNwSeg) * st.ckpt_nb + static_cast<u64>(b - bf);
      const NwPm v = st.ck_pm[cs];
      const int s = M.at(b % NB);
      M.pv[s] = v.pv;
      M.mv[s] = v.mv;
      M.sc[s] = st.ck_sc[cs];
      const BlockPlanes p = load_planes(tw, J.t_begin, n, static_cast<u32>(b));
      M.plo[s] = p.lo;
      M.phi[s] = p.hi;
    }
    top = bl;
  }
  TextCursor tc;
  tc.init(rw, b_base, m, rc, j0 + 1);
  u32 result = 0;
  for (int j = j0 + 1; j <= j_end; ++j) {
    const unsigned c = tc.get(j);
    const int bf = nw_bfirst(j, B.lo), bl = nw_blast(j, B.hi, B.nb);
    while (top < bl) {  // blocks entering the band at this column: edlib's all-(+1) upper bound below the block above
      ++top;
      const int s = M.at(top % NB);
      M.pv[s] = ~0ULL;
      M.mv[s] = 0;
      M.sc[s] = nw_jin(top, B.hi) == 1 ? 64 * (top + 1) : M.sc[M.at((top - 1) % NB)] + 64;
      const BlockPlanes p = load_planes(tw, J.t_begin, n, static_cast<u32>(top));
      M.plo[s] = p.lo;
      M.phi[s] = p.hi;
    }
    int hin = 1;  // above the first band block: the matrix border or a block that left the band (+1 boundary)
    // The block loop is software-pipelined: the state of block b + 1 is fetched from LDS before block b is computed, so
    // that the loads' latency overlaps the ~60 dependent ALU instructions of a block update (a lane-per-alignment wave
    // has little else to hide it behind: the band rings of 64 alignments fill most of a CU's LDS).
    int sl = bf % NB;
    int s = M.at(sl);
    u64 pv = M.pv[s], mv = M.mv[s], lo = M.plo[s], hi = M.phi[s];
    int sc = M.sc[s];
    const u64 seg0 = (static_cast<u64>(j - j0 - 1) * NB) * LANES + M.lane;
    const bool ck = mode == 0 && (j % kNwSeg == 0);
    const u64 ck0 = static_cast<u64>(j / kNwSeg) * st.ckpt_nb;
    for (int b = bf; b <= bl; ++b) {
      const int sl_n = sl + 1 == NB ? 0 : sl + 1;
      const int s_n = M.at(sl_n);
      u64 pv_n = 0, mv_n = 0, lo_n = 0, hi_n = 0;
      int sc_n = 0;
      if (b < bl) {
        pv_n = M.pv[s_n];
        mv_n = M.mv[s_n];
        lo_n = M.plo[s_n];
        hi_n = M.phi[s_n];
        sc_n = M.sc[s_n];
      }
      u64 eq = ((c & 1u) ? lo : ~lo) & ((c & 2u) ? hi : ~hi);
      if (b == B.nb - 1) {  // rows beyond n never match
        const u32 used = n - static_cast<u32>(64 * b);
        if (used < 64) eq &= (1ULL << used) - 1ULL;
      }
      const int hout = myers_block(pv, mv, eq, hin);
      sc += hout;
      M.pv[s] = pv;
      M.mv[s] = mv;
      M.sc[s] = sc;
      hin = hout;
      if (mode == 1) {
        const u64 slot = seg0 + static_cast<u64>(b - bf) * LANES;
        st.seg_pm[slot] = NwPm{pv, mv};
        st.seg_sc[slot] = sc;
      } else if (ck) {
        const u64 cs = ck0 + static_cast<u64>(b - bf);
        st.ck_pm[cs] = NwPm{pv, mv};
        st.ck_sc[cs] = sc;
      }
      if (b == B.nb - 1 && j == static_cast<int>(m)) {
        const u32 used = n - static_cast<u32>(64 * b);
        const u64 padmask = used >= 64 ? 0ULL : ~((1ULL << used) - 1ULL);
        result = static_cast<u32>(sc - RVN_POPC64(pv & padmask) + RVN_POPC64(mv & padmask)) + 1u;
      }
      sl = sl_n;
      s = s_n;
      pv = pv_n;
      mv = mv_n;
      lo = lo_n;
      hi = hi_n;
      sc = sc_n;
    }
  }
  return result;
}

// Cell values of the segment in the lane kernel's scratch layout (the walk's view of the band)
template <int NB, int LANES>
struct NwLaneCells {
  NwBand B;
  NwLaneStore st;
  int seg_j0, lane;
  __host__ __device__ u32 get(int x, int y) const {
    if (x == 0) return static_cast<u32>(y);
    if (y == 0) return static_cast<u32>(x);
    const int b = (x - 1) >> 6;
    if (y < nw_jin(b, B.hi) || y > nw_jout(b, B.lo)) return kNwInf;
    NwPm v;
    int sc;
    const int bf = nw_bfirst(y, B.lo);
    if (y == seg_j0) {
      const u64 cs = static_cast<u64>(y / kNwSeg) * st.ckpt_nb + static_cast<u64>(b - bf);
      v = st.ck_pm[cs];
      sc = st.ck_sc[cs];
    } else {
      const u64 slot = (static_cast<u64>(y - seg_j0 - 1) * NB + static_cast<u64>(b - bf)) * LANES + lane;
      v = st.seg_pm[slot];
      sc = st.seg_sc[slot];
    }
    const unsigned bit = static_cast<unsigned>((x - 1) & 63);
    const u64 below = bit == 63 ? 0ULL : (~0ULL << (bit + 1));
    return static_cast<u32>(sc - static_cast<int>(RVN_POPC64(v.pv & below)) + static_cast<int>(RVN_POPC64(v.mv & below)));
  }
};

// The whole job on one lane: pass 1 with threshold doubling, then the walk segment by segment.
// Returns 0 (records written), 1 (walk inconsistent), 2 (distance above kcap: the caller retries with a wider kernel);
// *distance / *k_used as in the wave kernel.  `active` = false makes the lane run through without touching memory
// (a wave's tail lanes without a job).
template <int NB, int LANES>
__host__ __device__ inline int nw_lane_job(const NwJob& J, const u64* __restrict__ t_words_all,
                                           const u64* __restrict__ r_words_all, const NwLaneMem<NB, LANES>& M,
                                           NwLaneStore st, u32 w, NwWindowRec* __restrict__ recs_all, u32* distance,
                                           u32* k_used) {
  const u64* tw = t_words_all + J.t_word;
  const u64* rw = r_words_all + J.r_word;
  u32 k = J.k;
  NwBand B;
  u32 res = 0;
  for (;;) {
    B = nw_band(J.n, J.m, k, 1);
    res = nw_lane_sweep<NB, LANES>(J, tw, rw, B, M, st, 0, static_cast<int>(J.m), 0) - 1u;
    if (res <= k) break;
    if (k >= J.kcap) {
      *distance = res;
      *k_used = k;
      return 2;
    }
    k = 2 * k < J.kcap ? 2 * k : J.kcap;
  }
  *distance = res;
  *k_used = k;
  NwWalkerT<NwLaneCells<NB, LANES>> wk;
  wk.cells.B = B;
  wk.cells.st = st;
  wk.cells.lane = M.lane;
  wk.init(J, t_words_all, r_words_all, res, w, recs_all);
  for (int sg = (static_cast<int>(J.m) - 1) / kNwSeg; sg >= 0 && wk.i > 0; --sg) {
    const int j0 = sg * kNwSeg;
    const int j_end = j0 + kNwSeg < static_cast<int>(J.m) ? j0 + kNwSeg : static_cast<int>(J.m);
    nw_lane_sweep<NB, LANES>(J, tw, rw, B, M, st, j0, j_end, 1);
    wk.cells.seg_j0 = j0;
    wk.seg_j0 = j0;
    wk.walk(true);
  }
  return wk.finish(true);
}

}  // namespace rvn
