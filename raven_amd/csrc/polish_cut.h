// polish_cut.h — where a read is cut at a window boundary of its target (host code, no HIP): the step that replaces
// racon's whole-overlap edlib path + find_breaking_points (polish.hip step 3).  Kept in a header of its own so that
// it can be unit-tested without a GPU (rvn_test_window_cut, tests/test_polish_cut.py).
#pragma once

#include <algorithm>
#include <utility>
#include <vector>

#include "common.h"

namespace rvn {

constexpr u32 kCutMax = 4096;  // longest residual of an anchor gap aligned for a window cut

struct WindowCut {
  u32 ql, tl, qr, tr;  // the piece left of the boundary ends at (ql, tl) exclusive; the one right of it begins at (qr, tr)
};

struct CutScratch {
  std::vector<u16> dp;  // banded NW of a residual
  std::vector<i32> cen;
  std::vector<u8> qrem;
  u64 n_nw = 0, nw_cells = 0;
};

// an: chain anchors as (target position, read position in the target's orientation), increasing in both, k bases each;
// B: boundary in target coordinates, an.front().first <= B < an.back().first + k; tbase(x) / qbase(x): 2-bit codes.
// Returns the (read, target) position pairs at which the piece left of B ends (target <= B) and the piece right of it
// begins (target >= B).  Inside an exact-match region both are (q(B), B).  Otherwise a greedy walk from the two anchors
// around B takes matches and isolated errors, a cluster of errors leaves a residual for a small banded unit-cost NW,
// and — like racon's breakpoints — pieces end / begin on aligned pairs (CIGAR 'M'); unaligned bases at the cut belong
// to neither piece.  Only a residual longer than kCutMax is left uncut (pieces end / begin at its borders).
template <class TB, class QB>
inline WindowCut window_cut(const std::vector<std::pair<u32, u32>>& an, u32 k, u32 B, u32 tlen, u32 qlen, TB&& tbase,
                            QB&& qbase, CutScratch& sc) {
    size_t lo = 0, hi = an.size();  // last anchor with t <= B
    while (hi - lo > 1) {
      const size_t mid = (lo + hi) / 2;
      if (an[mid].first <= B) lo = mid;
      else hi = mid;
    }
    const u32 ta = an[lo].first, qa = an[lo].second;
    if (B < ta + k || lo + 1 >= an.size()) return {qa + (B - ta), B, qa + (B - ta), B};
    const u32 tc = an[lo + 1].first, qc = an[lo + 1].second;
    u32 t0 = ta + k, q0 = qa + k;  // gap [t0, tc) <-> [q0, qc)
    if (tc > t0 && qc > q0) {
      // Greedy walk from both anchors towards B: a match, or an isolated error (substitution followed by 3
      // matches; one extra / one missing read base followed by 4 matches) is taken as the aligned path, which is
      // what any unit-cost aligner does there; only a cluster of errors stops the walk and leaves a residual
      // for the NW below.  Pieces end / begin on aligned pairs ('M'), bases in indels at the cut go to neither.
      const u32 tl_len = tlen;
      auto fwd = [&](u32 tt, u32 qq, u32 cnt) {
        for (u32 x = 0; x < cnt; ++x)
          if (tt + x >= tl_len || qq + x >= qlen || tbase(tt + x) != qbase(qq + x)) return false;
        return true;
      };
      auto bwd = [&](u32 tt, u32 qq, u32 cnt) {  // bases tt, tt-1, .. and qq, qq-1, ..
        for (u32 x = 0; x < cnt; ++x)
          if (tt < x || qq < x || tbase(tt - x) != qbase(qq - x)) return false;
        return true;
      };
      u32 lqd = q0, ltd = t0;  // end (exclusive) of the last aligned pair seen walking forward
      bool left_set = false;
      u32 lq = 0, lt = 0;
      while (t0 < tc && q0 < qc) {
        if (tbase(t0) == qbase(q0) || fwd(t0 + 1, q0 + 1, 3)) {
          if (t0 >= B) return left_set ? WindowCut{lq, lt, q0, t0} : WindowCut{lqd, ltd, q0, t0};  // left: after the last pair
          ++t0;
          ++q0;
          lqd = q0;
          ltd = t0;
        } else if (fwd(t0, q0 + 1, 4)) {  // extra base in the read
          if (t0 == B && !left_set) {
            left_set = true;
            lq = lqd;
            lt = ltd;
          }
          ++q0;
        } else if (fwd(t0 + 1, q0, 4)) {  // base missing in the read
          if (t0 == B && !left_set) {
            left_set = true;
            lq = lqd;
            lt = ltd;
          }
          ++t0;
        } else {
          break;
        }
      }
      if (left_set) {  // the cut sits in an indel and the walk stopped before the next aligned pair
        lqd = lq;
        ltd = lt;
      }
      u32 t1 = tc, q1 = qc;
      u32 rq = qc, rt = tc;  // first aligned pair at/after B seen walking backward (anchor C starts with one)
      while (t1 > t0 && q1 > q0) {
        if (tbase(t1 - 1) == qbase(q1 - 1) || (t1 >= 2 && q1 >= 2 && bwd(t1 - 2, q1 - 2, 3))) {
          if (t1 - 1 < B) return WindowCut{q1, t1, rq, rt};  // first pair left of the cut: the left piece ends after it
          --t1;
          --q1;
          rq = q1;
          rt = t1;
            } else if (q1 >= 2 && bwd(t1 - 1, q1 - 2, 4)) {  // extra base in the read
          --q1;
        } else if (t1 >= 2 && bwd(t1 - 2, q1 - 1, 4)) {  // base missing in the read
          --t1;
        } else {
          break;
        }
      }
      if (t1 <= t0 || q1 <= q0) {
      // nothing left to align between the two walks: the cut falls between them
      return WindowCut{lqd, ltd, rq, rt};
    }
    // B lies in the unmatched remainder [t0, t1) <-> [q0, q1) around the error(s): a unit-cost NW of the two
      // short segments decides where B maps, as racon's base-level path would
      const u32 nt = t1 - t0, nq = q1 - q0;
      if (nt <= kCutMax && nq <= kCutMax) {
        // dp(i, j): unit-cost NW of target residual [0, i) vs read residual [0, j), banded around the straight
        // line between the two walks (half-width 16 + the length difference).  Rows are stored with -inf padding
        // so the inner loop needs no bounds checks: row i holds columns sc.cen[i] - W .. sc.cen[i] + W at [pad, pad + bw).
        const u32 W = 16 + (nt > nq ? nt - nq : nq - nt);
        const u32 bw = 2 * W + 1;
        const u32 kInf = 0x3FFFu;
        sc.cen.resize(nt + 1);
        u32 max_shift = 0;
        for (u32 i = 0; i <= nt; ++i) {
          sc.cen[i] = static_cast<i32>(static_cast<u64>(i) * nq / std::max(nt, 1u));
          if (i) max_shift = std::max<u32>(max_shift, static_cast<u32>(sc.cen[i] - sc.cen[i - 1]));
        }
        const u32 pad = max_shift + 2;
        const u32 stride = bw + 2 * pad;
        sc.dp.assign(static_cast<size_t>(nt + 1) * stride, static_cast<u16>(kInf));
        ++sc.n_nw;
        sc.nw_cells += static_cast<u64>(nt + 1) * bw;
        u16* dp = sc.dp.data();
        auto at = [&](u32 i, i32 j) -> u32 {  // dp value or inf outside the band / matrix (traceback only)
          if (j < 0 || j > static_cast<i32>(nq)) return kInf;
          const i32 o = j - sc.cen[i] + static_cast<i32>(W);
          if (o < 0 || o >= static_cast<i32>(bw)) return kInf;
          return dp[static_cast<size_t>(i) * stride + pad + o];
        };
        sc.qrem.resize(nq + 1);
        for (u32 j = 0; j < nq; ++j) sc.qrem[j + 1] = static_cast<u8>(qbase(q0 + j));
        {  // row 0
          u16* r0 = dp + pad;
          for (i32 o = 0; o < static_cast<i32>(bw); ++o) {
            const i32 j = o + sc.cen[0] - static_cast<i32>(W);
            if (j >= 0 && j <= static_cast<i32>(nq)) r0[o] = static_cast<u16>(j);
          }
        }
        for (u32 i = 1; i <= nt; ++i) {
          const i32 c = sc.cen[i];
          const i32 sh = c - sc.cen[i - 1];
          const u32 tb_ = tbase(t0 + i - 1);
          const u16* prev = dp + static_cast<size_t>(i - 1) * stride + pad + sh;  // prev[o] = dp(i-1, j)
          u16* cur = dp + static_cast<size_t>(i) * stride + pad;
          const i32 jlo = std::max<i32>(0, c - static_cast<i32>(W)), jhi = std::min<i32>(nq, c + static_cast<i32>(W));
          const i32 base_o = -c + static_cast<i32>(W);
          if (jlo == 0) cur[base_o] = static_cast<u16>(i);
          for (i32 j = std::max(jlo, 1); j <= jhi; ++j) {
            const i32 o = j + base_o;
            const u32 d = prev[o - 1] + (tb_ != sc.qrem[j] ? 1u : 0u);
            const u32 u = prev[o] + 1u, l = cur[o - 1] + 1u;
            const u32 v = std::min(d, std::min(u, l));
            cur[o] = static_cast<u16>(v < kInf ? v : kInf);
          }
        }
        // like racon's breakpoints, a piece ends / begins on an aligned pair (CIGAR 'M'): the left piece ends
        // after the last pair with target < B, the right piece begins at the first pair with target >= B;
        // unaligned bases in between belong to neither
        const u32 ib = B >= t0 ? std::min(B - t0, nt) : 0;  // residual rows left of the cut (all of them when the backward walk passed B)
        u32 i = nt;
        i32 j = static_cast<i32>(nq);
        u32 li = 0, lj = 0;    // end (exclusive) of the last pair left of the cut; (0, 0) = exact region before
        u32 ri = nt, rj = nq;  // first pair at/after the cut; (nt, nq) = exact region after
        bool have_left = false;
        while ((i > 0 || j > 0) && !have_left) {
          const u32 here = at(i, j);
          if (i > 0 && j > 0 && here == at(i - 1, j - 1) + (tbase(t0 + i - 1) != sc.qrem[j] ? 1u : 0u)) {
            --i;
            --j;  // pair (target t0 + i, read q0 + j)
            if (i >= ib) {
              ri = i;
              rj = static_cast<u32>(j);
            } else {
              li = i + 1;
              lj = static_cast<u32>(j) + 1;
              have_left = true;
            }
          } else if (i > 0 && here == at(i - 1, j) + 1u) {
            --i;
          } else if (j > 0) {
            --j;
          } else {
            --i;
          }
        }
        const bool no_left = li == 0 && lj == 0 && !have_left;
        const bool no_right = ri == nt && rj == nq;
        return WindowCut{no_left ? lqd : q0 + lj, no_left ? ltd : t0 + li, no_right ? rq : q0 + rj, no_right ? rt : t0 + ri};
      }
      return WindowCut{lqd, ltd, rq, rt};
    }
    // overlapping / out-of-order anchors: cut at the anchor ends
    return {q0, t0, qc, tc};
}

}  // namespace rvn
