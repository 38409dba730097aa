// poa2_window.h — the 64 / 128 / 256-column POA window function of poa2.hip (one wave = one window), as a header: poa2.hip builds
// its kernels around it, and poa4.hip's persistent kernel calls the 64-column instance for the handful of windows its
// 32-column first attempt hands on (round 6: a launch of its own behind the persistent kernel cost one window's latency,
// 30-38 ms per C4 round, for three windows).  Device code only; internal linkage (every including file gets its own copy).
// The algorithm is described at the top of poa2.hip.
#pragma once

#include "poa.h"

namespace rvn {
namespace {
namespace p2 {

constexpr int kRing = 32;  // score rows kept in LDS
// The band is NCH chunks of 64 columns (one column per lane and chunk).  NCH = 1 (+-32 around the expected
// column) is the first attempt; windows whose traceback touches the band edge are repeated with NCH = 2 and, if
// that is not enough either with NCH = 4 (256 columns, two waves per workgroup) before the full-matrix kernel.
constexpr u32 kNone = 0xFFFFu;
constexpr i32 kNegBig = -0x3FFFFFFF;

template <int NCH>
struct alignas(16) Poa2Lds {  // per wave
  static constexpr int kBand = 64 * NCH;
  static constexpr int kRingStride = kBand + 2;
  union {
    // DP: score rows, slot = (row - 1) % kRing.  Every row is [pad][kBand cells][pad] with the pads (and one guard
    // cell in front of row 0) holding -inf, so a clamped index replaces the "is this column in the predecessor's
    // band" branches: cell c of the band is ring[2 + slot * kRingStride + c].
    i16 ring[2 + kRing * kRingStride];
    u8 stage[64 * kBand];  // traceback: backpointer rows of one 64-row block
    struct {
      u16 tgt[kPoa2MaxSeq];  // AddAlignment: graph node of every sequence position
    } add;
  } u;
  u8 seq_pad[kPoa2MaxSeq + 8];  // the layer's codes at seq_pad + 4; seq_pad[3] = 0xFF (position -1 matches nothing)
};  // (the layer's weights and the traceback's position -> node table live in HBM: every KB here is occupancy)

// k-th in-edge (among those inside the subgraph) of v, as a row index (rank + 1); slow path for in-degree > 4
__device__ __noinline__ u32 poa2_nth_pred(const Poa2Slot& g, u32 v, u32 k, bool full) {
  const u32 c = g.in_cnt[v];
  u32 seen = 0;
  for (u32 i = 0; i < c; ++i) {
    const u32 t = g.in_tail[v * kPoaMaxIn + i];
    if (full || g.mark[t]) {
      if (seen == k) return static_cast<u32>(g.rank_of[t]) + 1;
      ++seen;
    }
  }
  return 0;
}

// Ring miss: predecessor row `pr` from the int16 copy in HBM, one chunk of 64 columns; returns up | dg << 32.  Out of
// line AND returning in registers on purpose: on gfx9 the vector memory counter is shared by loads and stores, so a
// load (or a result handed back through scratch memory) on the common path makes every DP row wait for the previous
// row's stores to be acknowledged — the wait for this load stays inside the function.
__device__ __noinline__ unsigned long long poa2_fetch_miss(const i16* __restrict__ Hs, const uint4* __restrict__ tb, u32 pr,
                                                           i32 j_first, i32 kBand) {
  const i32 pb = static_cast<i32>(tb[pr].x & 0xFFFFu);
  const i16* R = Hs + static_cast<size_t>(pr) * kBand;
  const i32 i0 = j_first - pb;
  const i32 a0 = i0 < 0 ? 0 : (i0 > kBand - 1 ? kBand - 1 : i0);
  const i32 a0m = i0 < 1 ? 0 : (i0 > kBand ? kBand - 1 : i0 - 1);
  const i32 u0 = R[a0], d0 = R[a0m];
  const i32 up = (i0 >= 0 && i0 < kBand) ? u0 : kNegInf16;
  const i32 dg = (i0 >= 1 && i0 <= kBand) ? d0 : kNegInf16;
  return static_cast<unsigned long long>(static_cast<u32>(up)) | (static_cast<unsigned long long>(static_cast<u32>(dg)) << 32);
}

// (diagonal, vertical) candidate pair out of the virtual start row H[0][j] = j * g; out of line so that its four
// instructions are not hoisted into every row (only rows without an in-edge inside the subgraph use it)
__device__ __noinline__ u32 poa2_virtual_pair(i32 jv, i32 jgv, i32 gp) {
  const i32 dgv = jv >= 1 ? jgv - gp : kNegInf16;
  return (static_cast<u32>(dgv) & 0xFFFFu) | (static_cast<u32>(jgv) << 16);
}

typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 as_pk16(u32 x) { return __builtin_bit_cast(pk16, x); }

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Consensus of a finished window.  The heaviest-path pass (spoa TraverseHeaviestBundle) walks the nodes in topological
// order and looks up the scores of their in-edges' tails: with the scores in the wave's LDS (the DP buffers are dead by
// now; one int32 per node) and the first four in-edges of 64 nodes at a time in registers, a node costs one LDS round
// trip instead of three dependent global loads.  Same scalar rule as poa_consensus_scores_lane0, same results.
template <int NCH>
__device__ void poa2_consensus(Poa2Slot& g, u32 n_nodes, u32 nmax, const PoaWindow& win, int trim, Poa2Lds<NCH>& S,
                               u8* __restrict__ out, u32* out_len) {
  const int lane = lane_id();
  constexpr u32 kCap = sizeof(Poa2Lds<NCH>) / 4;
  i32* lsc = reinterpret_cast<i32*>(&S);
  i32 maxn = -1;
  if (n_nodes > kCap) {
    if (lane == 0) maxn = poa_consensus_scores_lane0(g, n_nodes);
    maxn = rfl(maxn);
  } else {
    i32 max_sc = 0;
    const u32 nn = static_cast<u32>(rfl(static_cast<int>(n_nodes)));
    for (u32 r0 = 0; r0 < nn; r0 += 64) {
      const u32 rows = nn - r0 < 64 ? nn - r0 : 64;
      int m_it = 0, m_c = 0, m_t01 = 0, m_t23 = 0, m_w0 = 0, m_w1 = 0, m_w2 = 0, m_w3 = 0;
      if (static_cast<u32>(lane) < rows) {
        m_it = g.order[r0 + lane];
        m_c = g.in_cnt[m_it];
        const u16* tp = g.in_tail + static_cast<size_t>(m_it) * kPoaMaxIn;
        const i32* wp = g.in_w + static_cast<size_t>(m_it) * kPoaMaxIn;
        m_t01 = static_cast<int>(static_cast<u32>(tp[0]) | (static_cast<u32>(tp[1]) << 16));
        m_t23 = static_cast<int>(static_cast<u32>(tp[2]) | (static_cast<u32>(tp[3]) << 16));
        m_w0 = wp[0];
        m_w1 = wp[1];
        m_w2 = wp[2];
        m_w3 = wp[3];
      }
      for (u32 l = 0; l < rows; ++l) {
        const u32 it = static_cast<u32>(rl(m_it, static_cast<int>(l)));
        const u32 c = static_cast<u32>(rl(m_c, static_cast<int>(l)));
        const u32 t01 = static_cast<u32>(rl(m_t01, static_cast<int>(l))), t23 = static_cast<u32>(rl(m_t23, static_cast<int>(l)));
        i32 sc = -1, pd = -1, pd_sc = 0;
        for (u32 k = 0; k < c; ++k) {
          i32 wgt, t;
          if (k < 4) {
            t = static_cast<i32>(((k < 2 ? t01 : t23) >> (16 * (k & 1))) & 0xFFFFu);
            wgt = k == 0 ? rl(m_w0, static_cast<int>(l)) : (k == 1 ? rl(m_w1, static_cast<int>(l)) : (k == 2 ? rl(m_w2, static_cast<int>(l)) : rl(m_w3, static_cast<int>(l))));
          } else {
            wgt = rfl(g.in_w[static_cast<size_t>(it) * kPoaMaxIn + k]);
            t = rfl(static_cast<int>(g.in_tail[static_cast<size_t>(it) * kPoaMaxIn + k]));
          }
          const i32 st = rfl(lsc[t]);
          if (sc < wgt || (sc == wgt && pd_sc <= st)) {
            sc = wgt;
            pd = t;
            pd_sc = st;
          }
        }
        if (pd != -1) sc += pd_sc;
        if (lane == 0) {
          lsc[it] = sc;
          g.scores[it] = sc;
          g.preds[it] = pd;
        }
        if (maxn == -1 || max_sc < sc) {
          maxn = static_cast<i32>(it);
          max_sc = sc;
        }
      }
    }
  }
  wsync();  // scores / predecessors in HBM visible to lane 0's branch completion and traceback
  u32 cl = 0;
  i32 begin = 0, end = -1;
  if (lane == 0) poa_consensus_trace_lane0(g, n_nodes, nmax, win, trim, maxn, &cl, &begin, &end);
  cl = static_cast<u32>(rfl(static_cast<int>(cl)));
  begin = rfl(begin);
  end = rfl(end);
  wsync();  // g.stack
  i32 n_out = end - begin + 1;
  if (n_out < 0) n_out = 0;
  if (static_cast<u32>(n_out) > win.out_cap) n_out = static_cast<i32>(win.out_cap);
  for (i32 p = lane; p < n_out; p += 64) out[p] = g.code[g.stack[cl - 1 - static_cast<u32>(begin + p)]];
  if (lane == 0) *out_len = static_cast<u32>(n_out);
}

template <int NCH>
__device__ __forceinline__ u32 poa2_window(const PoaWindow& win, const PoaLayer* __restrict__ layers,
                                           const PoaSrc& src, Poa2Slot& g,
                                           u32 nmax, u32 lmax, int m, int n_, int gp, int trim, Poa2Lds<NCH>& S,
                                           u8* __restrict__ out, u32* out_len,
                                           unsigned long long* __restrict__ phase_cycles, u32 probe) {
  constexpr int kBand = 64 * NCH;
  constexpr int kRingStride = kBand + 2;
  const int lane = lane_id();
  unsigned long long t_sub = 0, t_dp = 0, t_tb = 0, t_add = 0, t_ord = 0, t_cons = 0, t0 = 0;
  unsigned long long c_full = 0, c_band = 0;  // DP cells: full-matrix equivalent / inside the computed band
  auto tick = [&]() { t0 = __builtin_readcyclecounter(); };
  auto tock = [&](unsigned long long& acc) { acc += __builtin_readcyclecounter() - t0; };
  const PoaLayer bb = layers[win.layer_first];
  const u32 blen = bb.len;
  auto copy_backbone = [&]() {
    const u32 n = blen < win.out_cap ? blen : win.out_cap;
    for (u32 i = lane; i < n; i += 64) out[i] = static_cast<u8>(poa_layer_code(src, bb, i));
    if (lane == 0) *out_len = n;
  };
  // layers dropped by racon's mean-quality filter do not count as sequences of the window
  u32 n_eff = win.n_layers;
  if (src.layer_ok) {
    u32 cnt = 0;
    for (u32 i = 1 + lane; i < win.n_layers; i += 64) cnt += src.layer_ok[win.layer_first + i] ? 1u : 0u;
    n_eff = 1 + wave_sum(cnt);
  }
  if (n_eff < 3) {
    copy_backbone();
    return 0;
  }
  if (blen == 0 || blen > nmax || blen > lmax) {
    copy_backbone();
    return 4;
  }
  // ---- backbone graph (spoa AddAlignment with an empty alignment) ----
  u32 n_nodes = blen;
  for (u32 i = lane; i < blen; i += 64) {
    g.code[i] = static_cast<u8>(poa_layer_code(src, bb, i));
    g.al_cnt[i] = 0;
    g.visits[i] = blen >= 2 ? 1 : 0;
    g.rank_of[i] = static_cast<u16>(i);
    g.order[i] = static_cast<u16>(i);
    g.bpos[i] = static_cast<u16>(i);
    const i32 wi = poa_layer_weight(src, bb, i);
    if (i > 0) {
      const i32 wp = poa_layer_weight(src, bb, i - 1);
      g.in_cnt[i] = 1;
      g.in_tail[i * kPoaMaxIn] = static_cast<u16>(i - 1);
      g.in_w[i * kPoaMaxIn] = wp + wi;
    } else {
      g.in_cnt[i] = 0;
    }
    g.out_cnt[i] = i + 1 < blen ? 1 : 0;
  }
  wsync();
  const u32 offset = static_cast<u32>(0.01 * blen);
  u32 failed = 0;
  u32 dev_max = 0;  // probe: largest distance of a traceback cell from the centre of its row's band
  // (match | gap << 16), (mismatch | gap << 16): what v_pk_add_i16 adds to a (diagonal, vertical) candidate pair
  const u32 pk_match = (static_cast<u32>(m) & 0xFFFFu) | (static_cast<u32>(gp) << 16);
  const u32 pk_mismatch = (static_cast<u32>(n_) & 0xFFFFu) | (static_cast<u32>(gp) << 16);

  for (u32 li = 1; li < win.n_layers && !failed; ++li) {
    const PoaLayer L = layers[win.layer_first + li];
    const u32 len = L.len;
    if (len == 0 || (src.layer_ok && !src.layer_ok[win.layer_first + li])) continue;
    if (len > lmax || len > kPoa2MaxSeq) {
      failed = 4;
      break;
    }
    for (u32 i = lane; i < len; i += 64) {
      S.seq_pad[4 + i] = static_cast<u8>(poa_layer_code(src, L, i));
      g.pos_node[i] = static_cast<u16>(kNone);
    }
    if (lane == 0) S.seq_pad[3] = 0xFF;
    const bool full = L.begin < offset && L.end > blen - offset;
    tick();
    // ---- 1. subgraph marks ----
    if (!full) poa_subgraph_marks(g, n_nodes, nmax, L.begin, L.end);
    tock(t_sub);
    tick();
    // ---- 2. banded NW ----
    const u32 w = len + 1;
    const i32 lb = static_cast<i32>(L.begin);
    const i32 span = static_cast<i32>(L.end) - static_cast<i32>(L.begin) + 1;
    i32 best_score = -0x7FFFFFFF;
    u32 best_row = 0, best_node = 0;  // end node: best score of the last column, equal scores -> smallest node id
    int ring_tag = 0, ring_b = 0;  // lane s describes ring slot s: row stored there (0 = none), its band start
    u32 last_row = 0xFFFFFFFFu;  // NCH == 1: the row computed last, its band start and its cells (one per lane)
    i32 last_b = 0, last_h = 0;
    u32 ring_miss = 0;  // != 0: a predecessor row was no longer in the ring
    int m_b_last = 0;
    u32 marked_before = 0;
    {  // -inf pads (the union is reused by the traceback / AddAlignment of the previous layer)
      const u32 sl = static_cast<u32>(lane) >> 1;
      S.u.ring[1 + sl * kRingStride + ((lane & 1) ? kBand + 1 : 0)] = static_cast<i16>(kNegInf16);
      if (lane == 0) S.u.ring[0] = static_cast<i16>(kNegInf16);
    }
    bool dirty = false;  // HBM score rows stored since the last fence
    const i32 lane_gp = lane * gp;
    for (u32 r0 = 0; r0 < n_nodes; r0 += 64) {
      // metadata of 64 rows at once, one row per lane; it is also the traceback's row table
      int m_v = 0, m_np = 0, m_p01 = 0, m_p23 = 0, m_code = 0, m_outc = 1, m_marked = 0, m_b = 0;
      int m_meta = 0;  // NCH == 1: marked | #in-edges << 1 | code << 6 | end node << 8, one readlane per row
      const int m_b_prev = m_b_last;  // band start | ring index << 16 of the previous block of 64 rows
      int m_bi = 0;                   // NCH == 1: band start | (number of marked rows before this one) << 16
      if (r0 + lane < n_nodes) {
        m_v = g.order[r0 + lane];
        m_marked = (full || g.mark[m_v]) ? 1 : 0;
        if (m_marked) {
          m_code = g.code[m_v];
          m_outc = full ? g.out_cnt[m_v] : g.sub_out[m_v];
          const u32 c = g.in_cnt[m_v];
          for (u32 k = 0; k < c; ++k) {
            const u32 t = g.in_tail[m_v * kPoaMaxIn + k];
            if (full || g.mark[t]) {
              const int pr = static_cast<int>(g.rank_of[t]) + 1;
              if (m_np < 2) m_p01 |= pr << (16 * m_np);
              else if (m_np < 4) m_p23 |= pr << (16 * (m_np - 2));
              ++m_np;
            }
          }
          i32 b = poa_layer_center(L, static_cast<i32>(g.bpos[m_v]) - lb, span) - kBand / 2;
          const i32 bmax = static_cast<i32>(w) - kBand;
          b = b > bmax ? bmax : b;
          b = b < 0 ? 0 : b;
          m_b = b;
        }
        uint4 t;
        t.x = static_cast<u32>(m_b) | (static_cast<u32>(m_v) << 16);
        t.y = static_cast<u32>(m_np);
        t.z = static_cast<u32>(m_p01);
        t.w = static_cast<u32>(m_p23);
        g.tb[r0 + lane + 1] = t;
        m_meta = m_marked | (m_np << 1) | (m_code << 6) | ((m_outc == 0 ? 1 : 0) << 8);
      }
      // every load above must have returned BEFORE the row loop: a wait for them inside the loop would also wait
      // for the rows' own stores (loads and stores share the vector memory counter)
      asm volatile("" ::"v"(m_v), "v"(m_np), "v"(m_p01), "v"(m_p23), "v"(m_code), "v"(m_outc), "v"(m_marked), "v"(m_b), "v"(m_meta));
      {  // ring slots are handed out per COMPUTED row, so rows outside the layer's subgraph do not age the ring
        const unsigned long long mk = __ballot(m_marked != 0);
        const unsigned long long below = (1ULL << lane) - 1ULL;
        m_bi = m_b | static_cast<int>(((marked_before + static_cast<u32>(__popcll(mk & below))) & 0xFFFFu) << 16);
        marked_before += static_cast<u32>(__popcll(mk));
      }
      m_b_last = m_bi;
      const u32 rows_here = static_cast<u32>(rfl(static_cast<int>(n_nodes - r0 < 64 ? n_nodes - r0 : 64)));  // uniform loop
      {  // work counters: rows of this layer's (sub)graph x layer length = the cells spoa's full NW computes
        const u32 marked_rows = static_cast<u32>(__popcll(__ballot(m_marked != 0)));
        c_full += static_cast<unsigned long long>(marked_rows) * len;
        c_band += static_cast<unsigned long long>(marked_rows) * (w < static_cast<u32>(kBand) ? w : static_cast<u32>(kBand));
      }
      if constexpr (NCH == 1) {
        // One chunk of 64 columns.  The two candidates a predecessor row contributes to a cell — diagonal from its
        // column j - 1, vertical from its column j — are adjacent int16 cells of the ring row, i.e. ONE 32-bit LDS
        // read, and stay packed: v_pk_add_i16 adds (match/mismatch, gap) to both, v_pk_max_i16 folds the in-edges.
        // When the predecessor is the row computed just before and its band starts at the same or the previous
        // column (the chain case), the pair comes out of that row's registers through one DPP shift.  (Reading the
        // other rows' pairs one row ahead was tried: the loop is issue-bound, the extra instructions cost more than
        // the hidden LDS latency.)  Columns beyond the layer (only
        // when the layer is shorter than the band) and column 0's diagonal need no masks: no valid cell ever reads
        // them (column 0 reads the -inf pad or the explicit -inf of the virtual row).  A predecessor that has left
        // the ring is not looked up in HBM: the window is repeated by the 128-column kernel, which keeps a score copy
        // in HBM for that case (status 7); it does
        // not happen on racon-like windows.
        unsigned long long todo = __ballot((m_meta & 1) != 0);
        // Ring slot of a row = its index among the computed (marked) rows mod kRing, so a predecessor is still in its slot
        // iff at most kRing rows were computed since; index and band start of a predecessor come from the block metadata
        // (this block's or the previous one's): the ring needs no tags.
        auto ring_pair = [&](u32 pr, i32 jv, u32 cur_idx) -> u32 {
          const u32 pbi = static_cast<u32>((pr - 1 >= r0) ? rl(m_bi, static_cast<int>((pr - 1) & 63))
                                                           : rl(m_b_prev, static_cast<int>((pr - 1) & 63)));
          const u32 pidx = pbi >> 16;
          ring_miss |= (((cur_idx - pidx) & 0xFFFFu) > static_cast<u32>(kRing) || pr + 63 < r0) ? 1u : 0u;
          const u32 slot = pidx & (kRing - 1);
          i32 cc = jv - static_cast<i32>(pbi & 0xFFFFu);
          cc = cc < -1 ? -1 : (cc > kBand ? kBand : cc);
          u32 pair;  // low half: predecessor's column j - 1 (diagonal), high half: its column j (vertical)
          __builtin_memcpy(&pair, &S.u.ring[2 + static_cast<i32>(slot) * kRingStride + cc - 1], 4);
          return pair;
        };
        u32 pkm_v = pk_match, pkx_v = pk_mismatch;
        while (todo) {
          asm volatile("" : "+v"(pkm_v), "+v"(pkx_v));  // keep both in VGPRs: one v_cndmask per row instead of rebuilding them
          const int ri = __builtin_ctzll(todo);
          todo &= todo - 1;
          const u32 row = r0 + static_cast<u32>(ri) + 1;
          const int meta = rl(m_meta, ri);
          const u32 bi = static_cast<u32>(rl(m_bi, ri));
          const i32 b = static_cast<i32>(bi & 0xFFFFu);
          const u32 cur_idx = bi >> 16;
          const u32 p01 = static_cast<u32>(rl(m_p01, ri));
          u32 np = (static_cast<u32>(meta) >> 1) & 31u;
          if (np == 0) np = 1;  // no in-edge inside the subgraph: the virtual start row (p01 == 0)
          const u32 vc = (static_cast<u32>(meta) >> 6) & 3u;
          const i32 jv = b + lane;
          const i32 jgv = __mul24(jv, gp);
          const u32 ch = S.seq_pad[3 + jv];
          const pk16 addc = as_pk16(ch == vc ? pkm_v : pkx_v);
          pk16 acc;
          u32 kd = 0, kv = 0;
          for (u32 k = 0; k < np; ++k) {
            u32 pr;
            if (k < 2) pr = (p01 >> (16 * k)) & 0xFFFFu;
            else if (k < 4) pr = (static_cast<u32>(rl(m_p23, ri)) >> (16 * (k - 2))) & 0xFFFFu;
            else pr = static_cast<u32>(rfl(static_cast<int>(poa2_nth_pred(g, static_cast<u32>(rl(m_v, ri)), k, full))));
            u32 pair;
            const u32 shift = static_cast<u32>(b - last_b);
            if (pr == last_row && shift <= 1u) {
              // last_h: lane l holds column last_b + l of row last_row
              const i32 nb = shift ? __builtin_amdgcn_update_dpp(kNegInf16, last_h, 0x130 /* wave_shl:1 */, 0xf, 0xf, false)
                                   : __builtin_amdgcn_update_dpp(kNegInf16, last_h, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
              const i32 upv = shift ? nb : last_h, dgv = shift ? last_h : nb;
              pair = (static_cast<u32>(dgv) & 0xFFFFu) | (static_cast<u32>(upv) << 16);
            } else if (pr == 0) {  // H[0][j] = j * g
              pair = poa2_virtual_pair(jv, jgv, gp);
            } else {
              pair = ring_pair(pr, jv, cur_idx);
            }
            const pk16 cand = as_pk16(pair) + addc;
            if (k == 0) {
              acc = cand;
            } else {  // spoa: the FIRST in-edge reaching the maximum wins -> strict comparisons
              kd = cand.x > acc.x ? k : kd;
              kv = cand.y > acc.y ? k : kv;
              acc = __builtin_elementwise_max(acc, cand);
            }
          }
          const i32 bdv = acc.x, bvv = acc.y;
          const i32 best = bdv >= bvv ? bdv : bvv;
          u32 code = bdv >= bvv ? kd : 16u + kv;
          // spoa's traceback priority: diagonal (first in-edge reaching the max), vertical, horizontal
          const i32 xs = wave_inclusive_max_fused(best - jgv);
          i32 hh = xs + jgv;
          if (hh > best) code = 32u;
          hh = hh < kNegInf16 ? kNegInf16 : hh;
          const u32 rslot = cur_idx & (kRing - 1);
          S.u.ring[2 + rslot * kRingStride + lane] = static_cast<i16>(hh);
          g.BP[static_cast<size_t>(row) * kBand + lane] = static_cast<u8>(code);
          last_row = static_cast<u32>(rfl(static_cast<int>(row)));
          last_b = rfl(b);
          last_h = hh;
          if (meta & 256) {  // an end node: score of the last column if the band has it
            const i32 idx = static_cast<i32>(w) - 1 - b;
            const i32 sce = (idx >= 0 && idx < 64) ? rl(hh, idx) : -0x7FFFFFFF;
            const u32 vnode = static_cast<u32>(rl(m_v, ri));
            if (sce > best_score || (sce == best_score && best_row != 0 && vnode < best_node)) {
              best_score = sce;
              best_row = row;
              best_node = vnode;
            }
          }
        }
      } else
      for (u32 ri = 0; ri < rows_here; ++ri) {
        if (!rl(m_marked, static_cast<int>(ri))) continue;
        const u32 row = static_cast<u32>(rfl(static_cast<int>(r0 + ri + 1)));  // uniform: keeps the row addressing scalar
        const u32 v = static_cast<u32>(rl(m_v, static_cast<int>(ri)));
        u32 np = static_cast<u32>(rl(m_np, static_cast<int>(ri)));
        const u32 p01 = static_cast<u32>(rl(m_p01, static_cast<int>(ri)));
        const u32 p23 = static_cast<u32>(rl(m_p23, static_cast<int>(ri)));
        const i32 b = rl(m_b, static_cast<int>(ri));
        const u32 vc = static_cast<u32>(rl(m_code, static_cast<int>(ri)));
        if (np == 0) np = 1;  // no in-edge inside the subgraph: the virtual start row (p01 == 0)
        i32 j[NCH], jg[NCH], sc[NCH], bd[NCH], bv[NCH];
        u32 kd[NCH], kv[NCH];
        bool val[NCH], dok[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          j[c] = b + 64 * c + lane;
          jg[c] = (b + 64 * c) * gp + lane_gp;  // j * g without a vector multiply
          val[c] = j[c] < static_cast<i32>(w);
          dok[c] = val[c] && j[c] >= 1;
          // match/mismatch per column (clamped read; masked by dok)
          const i32 qi = j[c] >= 1 ? (j[c] - 1 < kPoa2MaxSeq ? j[c] - 1 : kPoa2MaxSeq - 1) : 0;
          sc[c] = vc == S.seq_pad[4 + qi] ? m : n_;
          bd[c] = kNegBig;
          bv[c] = kNegBig;
          kd[c] = 0;
          kv[c] = 0;
        }
        for (u32 k = 0; k < np; ++k) {
          u32 pr;
          if (k < 2) pr = (p01 >> (16 * k)) & 0xFFFFu;
          else if (k < 4) pr = (p23 >> (16 * (k - 2))) & 0xFFFFu;
          else pr = poa2_nth_pred(g, v, k, full);
          i32 up[NCH], dg[NCH];
          if (pr == 0) {  // H[0][j] = j * g
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              up[c] = jg[c];
              dg[c] = jg[c] - gp;
            }
          } else {
            const u32 slot = (pr - 1) & (kRing - 1);
            if (static_cast<u32>(rl(ring_tag, static_cast<int>(slot))) == pr) {
              // column j of the predecessor = cell (j - pb); out-of-band cells land on a -inf pad
              const i32 pb = rl(ring_b, static_cast<int>(slot));
              const i32 base = 2 + static_cast<i32>(slot) * kRingStride;
#pragma unroll
              for (int c = 0; c < NCH; ++c) {
                i32 cc = j[c] - pb;
                cc = cc < -1 ? -1 : (cc > kBand ? kBand : cc);
                up[c] = S.u.ring[base + cc];
                dg[c] = S.u.ring[base + cc - 1];
              }
            } else {  // ring miss (rare): the int16 copy in HBM
              if (dirty) {  // earlier stores must have landed
                wsync();
                dirty = false;
              }
#pragma unroll
              for (int c = 0; c < NCH; ++c) {
                const unsigned long long ud = poa2_fetch_miss(g.Hs, g.tb, pr, j[c], kBand);
                up[c] = static_cast<i32>(static_cast<u32>(ud));
                dg[c] = static_cast<i32>(static_cast<u32>(ud >> 32));
              }
            }
          }
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const i32 d = dok[c] ? dg[c] + sc[c] : kNegBig;
            kd[c] = d > bd[c] ? k : kd[c];
            bd[c] = d > bd[c] ? d : bd[c];
            const i32 x = val[c] ? up[c] + gp : kNegBig;
            kv[c] = x > bv[c] ? k : kv[c];
            bv[c] = x > bv[c] ? x : bv[c];
          }
        }
        // spoa's traceback priority: diagonal (first in-edge reaching the max), vertical, horizontal
        const u32 rslot = (row - 1) & (kRing - 1);
        const u32 rbase = 2 + rslot * kRingStride;
        i32 h[NCH];
        i32 carry = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (c == 0 || b + 64 * c < static_cast<i32>(w)) {  // the chunk has columns of the sequence
            const i32 best = bd[c] >= bv[c] ? bd[c] : bv[c];
            u32 code = bd[c] >= bv[c] ? kd[c] : 16u + kv[c];
            i32 x = val[c] ? best - jg[c] : kNegBig;
            x = wave_inclusive_max_dpp(x, kNegBig);
            i32 hh = x + jg[c];
            if (c > 0) {  // the gap chain entering from the previous chunk
              const i32 viac = carry + lane_gp + gp;
              hh = viac > hh ? viac : hh;
            }
            if (hh > best) code = 32u;
            hh = hh < kNegInf16 ? kNegInf16 : hh;
            h[c] = hh;
            carry = rl(hh, 63);
            S.u.ring[rbase + 64 * c + lane] = static_cast<i16>(hh);
            g.Hs[static_cast<size_t>(row) * kBand + 64 * c + lane] = static_cast<i16>(hh);
            g.BP[static_cast<size_t>(row) * kBand + 64 * c + lane] = static_cast<u8>(code);
          } else {
            h[c] = kNegInf16;
          }
        }
        if (lane == static_cast<int>(rslot)) {
          ring_tag = static_cast<int>(row);
          ring_b = b;
        }
        dirty = true;
        if (rl(m_outc, static_cast<int>(ri)) == 0) {  // an end node: score of the last column if the band has it
          const i32 idx = static_cast<i32>(w) - 1 - b;
          i32 sce = -0x7FFFFFFF;
#pragma unroll
          for (int c = 0; c < NCH; ++c)
            if (idx >= 64 * c && idx < 64 * (c + 1)) sce = rl(h[c], idx - 64 * c);  // that chunk is active: idx < w - b
          const u32 vnode = static_cast<u32>(rl(m_v, static_cast<int>(ri)));
          if (sce > best_score || (sce == best_score && best_row != 0 && vnode < best_node)) {
            best_score = sce;
            best_row = row;
            best_node = vnode;
          }
        }
      }
    }
    wsync();  // backpointers visible to the traceback
    tock(t_dp);
    tick();
    if (ring_miss) {
      failed = 7u | (li << 8);
      break;
    }
    if (best_row == 0) {  // the last column is in no end node's band
      failed = kPoaBandHit | (li << 8);
      break;
    }
    // ---- 3. traceback over backpointer bytes, block-staged in LDS ----
    // A scalar walk: the row table of the staged 64-row block lives in registers (lane l = row l of the block, read with
    // v_readlane), the backpointer of the current cell is one uniform LDS read, everything else is SALU — about one LDS
    // latency per step.  (Committing runs of diagonal moves through rank-adjacent rows with 64 speculating lanes was
    // the earlier scheme; on a graph that already holds 20 layers consecutive path nodes are rarely rank-adjacent and
    // the speculation cost more per step than it saved.)
    u32 bad = 0, band_hit = 0;
    {
      // every walk variable is wave-uniform; readfirstlane tells the compiler so (scalar registers, scalar branches)
      u32 i = static_cast<u32>(rfl(static_cast<int>(best_row)));
      i32 j = rfl(static_cast<i32>(w) - 1);
      const u32 n_rows_total = static_cast<u32>(rfl(static_cast<int>(n_nodes))) + 1;
      const u32 max_steps = static_cast<u32>(rfl(static_cast<int>(nmax + lmax + 2)));
      u32 cur_blk = 0xFFFFFFFFu;
      u32 steps = 0;
      int tbx = 0, tby = 0, tbz = 0, tbw = 0;
      while (i != 0) {  // once on the virtual row only insertions remain: pos_node already says kNone
        if (++steps > max_steps) {
          bad = 6;
          break;
        }
        const u32 blk = (i - 1) >> 6;
        if (blk != cur_blk) {
          wsync();
          const u32 row0 = blk * 64 + 1;
          const u32 nrows = n_rows_total - row0 < 64 ? n_rows_total - row0 : 64;
          const uint4* src = reinterpret_cast<const uint4*>(g.BP + static_cast<size_t>(row0) * kBand);
          uint4* dst = reinterpret_cast<uint4*>(S.u.stage);
          // 64 rows x kBand bytes, 16 B per lane and step, global -> LDS without passing through registers (one memory
          // round trip per block).  Indices beyond the last row clamp to the block's first bytes: those LDS rows are
          // never read.
          const u32 q_end = nrows * (kBand / 16);
#pragma unroll
          for (u32 it = 0; it < 4 * NCH; ++it) {
            const u32 q = it * 64 + lane;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + (q < q_end ? q : 0)),
                (__attribute__((address_space(3))) void*)(dst + it * 64), 16, 0, 0);
          }
          const uint4 t = g.tb[row0 + (static_cast<u32>(lane) < nrows ? lane : 0)];
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          tbx = static_cast<int>(t.x);
          tby = static_cast<int>(t.y);
          tbz = static_cast<int>(t.z);
          tbw = static_cast<int>(t.w);
          wsync();
          cur_blk = blk;
        }
        const int l = static_cast<int>((i - 1) & 63);
        const u32 x = static_cast<u32>(rl(tbx, l));
        const i32 bt = static_cast<i32>(x & 0xFFFFu);
        const u32 node = x >> 16;
        const i32 idx = j - bt;
        if (idx < 0 || idx >= kBand) {  // the path left the stored band: the alignment does not fit this band width
          band_hit = 1;
          break;
        }
        if ((idx < 2 && bt > 0) || (idx > kBand - 3 && bt + kBand < static_cast<i32>(w))) band_hit = 1;
        if (probe && (idx >= kBand / 2 ? bt + kBand < static_cast<i32>(w) : bt > 0)) {  // (not where the band is clamped to the layer's ends)
          const u32 dv = static_cast<u32>(idx >= kBand / 2 ? idx - kBand / 2 : kBand / 2 - 1 - idx);
          dev_max = dv > dev_max ? dv : dev_max;
        }
        const u32 code = static_cast<u32>(rfl(static_cast<int>(S.u.stage[l * kBand + idx])));
        if (code == 32u) {
          if (j == 0) {
            bad = 6;
            break;
          }
          --j;  // insertion: pos_node[j] stays kNone
          continue;
        }
        const u32 k = code & 15u;
        u32 pr;
        if (rl(tby, l) == 0) pr = 0;
        else if (k < 2) pr = (static_cast<u32>(rl(tbz, l)) >> (16 * k)) & 0xFFFFu;
        else if (k < 4) pr = (static_cast<u32>(rl(tbw, l)) >> (16 * (k - 2))) & 0xFFFFu;
        else pr = static_cast<u32>(rfl(static_cast<int>(poa2_nth_pred(g, node, k, full))));
        if (code < 16u) {
          if (j == 0) {
            bad = 6;
            break;
          }
          --j;
          g.pos_node[j] = static_cast<u16>(node);  // every lane stores the same value
        }
        i = pr;
      }
    }
    wsync();
    tock(t_tb);
    tick();
    if (bad) {
      failed = bad | (li << 8);
      break;
    }
    if (band_hit) {
      failed = kPoaBandHit | (li << 8);
      break;
    }
    // ---- 4. spoa AddAlignment, one sequence position per lane ----
    const u32 n_old = n_nodes;
    u32 first_p = 0xFFFFFFFFu;
    for (u32 p0 = 0; p0 < len && first_p == 0xFFFFFFFFu; p0 += 64) {
      const u32 p = p0 + lane;
      const unsigned long long bal = __ballot(p < len && g.pos_node[p] != kNone);
      if (bal) first_p = p0 + static_cast<u32>(__builtin_ctzll(bal));
    }
    // New nodes anchored after a column (aligned group) go after ALL its members; the unaligned prefix goes
    // before all members of the first column.
    u32 carry_slot = n_old, carry_b = static_cast<u32>(lb);
    if (first_p != 0xFFFFFFFFu) {
      const u32 an = g.pos_node[first_p];
      u32 r = g.rank_of[an];
      const u32 ac = g.al_cnt[an];
      for (u32 k = 0; k < ac; ++k) {
        const u32 rk = g.rank_of[g.al[an * 4 + k]];
        r = rk < r ? rk : r;
      }
      carry_slot = r;
      carry_b = g.bpos[an];
    }
    u32 total_new = 0;
    u32 ok = 1, why = 3;
    for (u32 p0 = 0; p0 < len; p0 += 64) {
      const u32 p = p0 + lane;
      const bool valid = p < len;
      const u32 an = valid ? g.pos_node[p] : kNone;
      const u32 letter = valid ? S.seq_pad[4 + p] : 0u;
      const bool has = valid && an != kNone;
      u32 tgt = kNone, gslot = 0, gb = 0, ac = 0;
      if (has) {
        u32 rmax = g.rank_of[an];
        ac = g.al_cnt[an];
        if (g.code[an] == letter) tgt = an;
        for (u32 k = 0; k < ac; ++k) {
          const u32 kt = g.al[an * 4 + k];
          const u32 rk = g.rank_of[kt];
          rmax = rk > rmax ? rk : rmax;
          if (tgt == kNone && g.code[kt] == letter) tgt = kt;
        }
        gslot = rmax + 1;
        gb = g.bpos[an];
      }
      // order slot / backbone coordinate of the last aligned position at or before p
      const unsigned long long bal = __ballot(has);
      const unsigned long long below = bal & (lane == 63 ? ~0ULL : ((2ULL << lane) - 1ULL));
      const int src = below ? 63 - __builtin_clzll(below) : 0;
      const u32 s_sh = static_cast<u32>(__shfl(static_cast<int>(gslot), src, 64));
      const u32 b_sh = static_cast<u32>(__shfl(static_cast<int>(gb), src, 64));
      const u32 fslot = below ? s_sh : carry_slot;
      const u32 fb = below ? b_sh : carry_b;
      if (bal) {
        const int top = 63 - __builtin_clzll(bal);
        carry_slot = static_cast<u32>(rl(static_cast<int>(gslot), top));
        carry_b = static_cast<u32>(rl(static_cast<int>(gb), top));
      }
      const bool is_new = valid && tgt == kNone;
      const unsigned long long nb = __ballot(is_new);
      const u32 cnt = static_cast<u32>(__builtin_popcountll(nb));
      if (n_old + total_new + cnt > nmax || total_new + cnt > lmax) {
        ok = 0;
        why = 2;
        break;
      }
      if (is_new) {
        const u32 t = total_new + static_cast<u32>(__builtin_popcountll(nb & lanemask_lt()));
        const u32 id = n_old + t;
        tgt = id;
        g.code[id] = static_cast<u8>(letter);
        g.in_cnt[id] = 0;
        g.out_cnt[id] = 0;
        g.visits[id] = 0;
        g.new_slot[t] = static_cast<u16>(fslot);
        g.bpos[id] = static_cast<u16>(fb);
        u32 c2 = 0;
        if (has) {  // joins an's aligned group
          for (u32 k = 0; k < ac; ++k) {
            const u32 kt = g.al[an * 4 + k];
            const u32 ck = g.al_cnt[kt];
            if (ck < 4) {
              g.al[kt * 4 + ck] = static_cast<u16>(id);
              g.al_cnt[kt] = static_cast<u8>(ck + 1);
            }
            if (c2 < 4) g.al[id * 4 + c2++] = static_cast<u16>(kt);
          }
          if (ac < 4) {
            g.al[an * 4 + ac] = static_cast<u16>(id);
            g.al_cnt[an] = static_cast<u8>(ac + 1);
          }
          if (c2 < 4) g.al[id * 4 + c2++] = static_cast<u16>(an);
        }
        g.al_cnt[id] = static_cast<u8>(c2);
      }
      total_new += cnt;
      if (valid) {
        S.u.add.tgt[p] = static_cast<u16>(tgt);
        if (len >= 2) g.visits[tgt] += 1;
      }
    }
    wsync();
    if (ok) {
      for (u32 p0 = 0; p0 < len; p0 += 64) {
        const u32 p = p0 + lane;
        bool okl = true;
        if (p >= 1 && p < len)
          okl = poa_add_edge(g, S.u.add.tgt[p - 1], S.u.add.tgt[p],
                             static_cast<i32>(static_cast<u8>(poa_layer_weight(src, L, p - 1))) +
                                 static_cast<i32>(static_cast<u8>(poa_layer_weight(src, L, p))));
        if (__ballot(!okl)) {
          ok = 0;
          why = 3;
        }
      }
    }
    wsync();
    if (!ok) {
      failed = why;
      break;
    }
    const u32 n_new = total_new;
    n_nodes = n_old + n_new;
    tock(t_add);
    tick();
    // ---- 5. order rebuild: old rank r -> r + #(new slots <= r); t-th new node -> slot_t + t ----
    if (n_new) {
      for (u32 r = lane; r < n_old; r += 64) {
        u32 lo = 0, hi = n_new;  // upper_bound(new_slot, r)
        while (lo < hi) {
          const u32 mid = (lo + hi) >> 1;
          if (g.new_slot[mid] <= r) lo = mid + 1;
          else hi = mid;
        }
        g.order2[r + lo] = g.order[r];
      }
      for (u32 t = lane; t < n_new; t += 64) g.order2[static_cast<u32>(g.new_slot[t]) + t] = static_cast<u16>(n_old + t);
      wsync();
      for (u32 r = lane; r < n_nodes; r += 64) {
        const u32 v = g.order2[r];
        g.order[r] = static_cast<u16>(v);
        g.rank_of[v] = static_cast<u16>(r);
      }
      wsync();
    }
    tock(t_ord);
  }
  if (failed) {
    copy_backbone();
    return failed;
  }
  tick();
  PoaWindow weff = win;
  weff.n_layers = n_eff;
  poa2_consensus<NCH>(g, n_nodes, nmax, weff, trim, S, out, out_len);
  wsync();
  tock(t_cons);
  if (phase_cycles && lane == 0) {
    atomicAdd(&phase_cycles[0], t_sub);
    atomicAdd(&phase_cycles[1], t_dp);
    atomicAdd(&phase_cycles[2], t_tb);
    atomicAdd(&phase_cycles[3], t_add);
    atomicAdd(&phase_cycles[4], t_ord);
    atomicAdd(&phase_cycles[5], t_cons);
    atomicAdd(&phase_cycles[6], c_full);
    atomicAdd(&phase_cycles[7], c_band);
  }
  return 1u | (probe ? (dev_max > 255u ? 255u : dev_max) << 16 : 0u);
}

}  // namespace p2
}  // namespace
}  // namespace rvn
