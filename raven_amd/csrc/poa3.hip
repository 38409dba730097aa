// poa3.hip — banded POA window kernel, FOUR windows per wave: the first attempt of rvn_poa_consensus_batch / of a
// polishing round's consensus stage (racon Window::GenerateConsensus over spoa, as driven by raven::Polish,
// RavenLib/src/polish.cc:43-51).  Same algorithm, graph layout, band rule and tie rules as poa2.hip's 64-column kernel
// (results are identical window for window); what changes is how the work sits on the wave.
//
// poa2 gives a window a whole wave: one DP row of 64 cells per loop iteration, every per-row decision (which
// predecessor rows, where their band starts, which ring slot) decoded by the scalar unit — 58 vector + 86 scalar
// instructions per row, and the scalar issue slots are what the kernel runs out of (profiles/r03_sq_poa_*.csv).
// Here a window owns a ROW OF 16 LANES (the unit DPP row operations work on), each lane four adjacent band columns, so
// one instruction stream advances four windows at once and everything that was scalar per row becomes a
// vector value that is uniform inside a 16-lane group:
//   * NW rows: the row's descriptor (band start, ring slot, node code, in-degree) and the descriptors of its first four
//     predecessor rows (ring slot + band start, resolved once per 16-row block by the lane that owns the row) reach
//     the group through ds_bpermute; the two candidates a predecessor contributes to a cell (diagonal from its column
//     j - 1, vertical from column j) come from one 12-byte LDS read per lane (five adjacent int16 cells of the
//     predecessor's ring row, -inf pads instead of range checks) and are folded with v_max3 on
//     (score << 6 | diagonal << 5 | 15 - in-edge): the maximum carries spoa's tie rule (diagonal before vertical, first
//     in-edge first) and the backpointer falls out of its low bits.  The horizontal gap chain is a prefix maximum of
//     H - j*g: three in-lane steps + four DPP row shifts.
//   * traceback: four walks in lockstep, one per group; 32-row blocks of backpointers and their row table staged in
//     the group's LDS.
//   * the graph update (spoa AddAlignment), the order rebuild, the subgraph marks and the consensus stay wave-wide per
//     window (they are 14 % of poa2's cycles) and run for the wave's windows one after the other.
// Per wave: 13.1 KB of LDS (16 ring rows + the layer's codes per window) -> 12 waves = 48 windows per CU (poa2: 24).
//
// Written against sv:: (simt.h): the same source runs under the host wavefront emulator, which is how the CPU suite
// checks this kernel against the oracle (tests/test_poa3_emulation.py -> rvn_poa_banded_emulate).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "poa.h"

namespace rvn {

namespace {

// A wave holds G windows; a window owns GS = 64 / G lanes, each lane C adjacent band columns (band = GS * C).
//   <4, 4>: four windows, 64-column band (the layout this file was first written for)
//   <4, 2>: four windows, 32 columns      <2, 1>: two windows, 32 columns      <2, 2>: two windows, 64 columns
// A 32-column band is enough for every window of a C4 polishing round (no alignment path strays more
// than 13 columns from the guided band centre); what does touch its edge goes on to the 64-column kernels.
template <int G_, int C_>
struct P3Cfg {
  static constexpr int G = G_, C = C_;
  static constexpr int GS = 64 / G_;
  static constexpr int kBand = GS * C_;
  static constexpr int kRing = 16;                // score rows a window keeps in LDS
  static constexpr int kPad = (C_ + 2) & ~1;      // -inf cells either side of a ring row (>= C + 1, even)
  static constexpr int kStride = kBand + 2 * kPad;  // int16 cells per ring row
  static constexpr int kNR = (C_ + 3) / 2;        // dwords read for the C + 1 cells a predecessor row contributes
  static constexpr int kNW = (C_ + 2) / 2;        // packed words they end up in
  static constexpr int kTbRows = 32;              // rows per traceback block
  static constexpr int kBpChunks = kTbRows * kBand / 16 / GS;  // 16-byte chunks of a block's backpointers per lane
  static constexpr int kTbPerLane = kTbRows / GS;              // row-table entries of a block per lane
  static_assert(G_ == 2 || G_ == 4, "a window owns one or two DPP rows");
  static_assert(C_ == 1 || C_ == 2 || C_ == 4, "columns per lane");
  static_assert(kBpChunks >= 1 && kBpChunks <= 8 && kTbPerLane >= 1 && kTbPerLane <= 2, "traceback staging registers");
};
constexpr u32 kNone3 = 0xFFFFu;
constexpr u32 kDescVirtual = 1u << 30, kDescMiss = 1u << 31;
constexpr i32 kVeryNeg = -0x40000000;
unsigned long long g_emu_lost_rows = 0;  // host emulation only: predecessor rows fetched from the HBM copy

template <class K>
struct alignas(16) Poa3Group {
  union {
    u32 ring32[K::kRing * K::kStride / 2 + 4];  // DP: [kRing][kPad | band cells | kPad] int16, slot = computed-row index % kRing
    struct {
      u8 bp[K::kTbRows * K::kBand];  // traceback: backpointer rows of one block
      u32 tb[K::kTbRows * 3];        // and their row table: band start | node << 16, in-edge rows 0,1, in-edge rows 2,3
    } tr;
  } u;
  u8 seq_pad[kPoa2MaxSeq + 16];  // the layer's codes at seq_pad + 4; seq_pad[3] = 0xFF (position -1 matches nothing)
};
template <class K>
struct alignas(16) Poa3Lds {
  Poa3Group<K> g[K::G];
};
static_assert(sizeof(Poa3Lds<P3Cfg<4, 4>>) <= 13648, "twelve waves per CU need <= 13.3 KB of LDS each");
static_assert(sizeof(Poa3Lds<P3Cfg<4, 2>>) <= 10240, "sixteen waves per CU need <= 10 KB of LDS each");

struct Poa3Args {
  const PoaWindow* windows;
  u32 n_windows;
  const PoaLayer* layers;
  PoaSrc src;
  unsigned char* scratch;
  size_t slot_bytes;
  u32 nmax, lmax;
  int m, n_, gp, trim;
  u8* out;
  u32* out_len;
  u32* status;
  unsigned long long* phase_cycles;
  const u32* sched;
  u32* next;
};

__host__ __device__ __forceinline__ u32 funnel_shr(u32 hi, u32 lo, u32 sh) {  // v_alignbit_b32; sh in [0, 31]
  return static_cast<u32>(((static_cast<unsigned long long>(hi) << 32) | lo) >> sh);
}
__host__ __device__ __forceinline__ i32 sext16(u32 x) { return static_cast<i32>(static_cast<i16>(x & 0xFFFFu)); }
__host__ __device__ __forceinline__ i32 max3(i32 a, i32 b, i32 c) {
  const i32 ab = a > b ? a : b;
  return ab > c ? ab : c;
}
// LDS traffic of one wave is executed in order on the GPU; the emulator's fibres need a rendezvous between a lane's
// LDS write and another lane's read of it
__host__ __device__ __forceinline__ void lds_order() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_wave_barrier();
#else
  sv::sync();
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
#define P3_MARK(x) asm volatile("; P3MARK " x)
#define P3_NO_IF_CONVERSION() asm volatile("")  // keeps a rare, wave-uniform branch a branch
// a group-parallel phase is its own function = its own register allocation (the kernel is built for 168 VGPRs); the
// address spaces the inliner would have seen are handed over as assumptions, or every access becomes a FLAT one
#define P3_PHASE  // (noinline was measured: no fewer spills inside the DP, and FLAT accesses where an assumption is not enough)
#define P3_ASSUME_GLOBAL(p) \
  __builtin_assume(!__builtin_amdgcn_is_shared((const void*)(p))); \
  __builtin_assume(!__builtin_amdgcn_is_private((const void*)(p)))
#define P3_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
#else
#define P3_MARK(x)
#define P3_NO_IF_CONVERSION()
#define P3_PHASE
#define P3_ASSUME_GLOBAL(p)
#define P3_ASSUME_LDS(p)
#endif

__host__ __device__ __forceinline__ i32 mul24(i32 a, i32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __mul24(a, b);
#else
  return a * b;
#endif
}

// The lane's slot pointer, made opaque to the optimiser: field addresses derived from it are computed where they are
// used instead of being hoisted out of the row loop and kept (or spilled) there — the DP is built for 168 VGPRs.
__host__ __device__ __forceinline__ unsigned char* opaque(unsigned char* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(p));
#endif
  return p;
}

// int16 half of `w` (HI: the upper one) * 64 + add: v_mad_i32_i16 reads the half through op_sel, so the score cells
// stay packed as the LDS read delivered them
template <bool HI>
__host__ __device__ __forceinline__ i32 cell_x64_plus(u32 w, i32 add) {
#if defined(__HIP_DEVICE_COMPILE__)
  i32 d;
  if constexpr (HI) asm("v_mad_i32_i16 %0, %1, 64, %2 op_sel:[1,0,0,0]" : "=v"(d) : "v"(w), "v"(add));
  else asm("v_mad_i32_i16 %0, %1, 64, %2" : "=v"(d) : "v"(w), "v"(add));
  return d;
#else
  return static_cast<i32>(static_cast<i16>((HI ? w >> 16 : w) & 0xFFFFu)) * 64 + add;
#endif
}
// (lo & 0xFFFF) | hi << 16
__host__ __device__ __forceinline__ u32 pack16(i32 lo, i32 hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(static_cast<u32>(hi), static_cast<u32>(lo), 0x05040100u);
#else
  return (static_cast<u32>(lo) & 0xFFFFu) | (static_cast<u32>(hi) << 16);
#endif
}
// both int16 halves clamped from below to kNegInf16
__host__ __device__ __forceinline__ u32 clamp_pair(u32 w) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef short pk16 __attribute__((ext_vector_type(2)));
  const pk16 lim = {static_cast<short>(kNegInf16), static_cast<short>(kNegInf16)};
  return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(pk16, w), lim));
#else
  const i32 a = sext16(w), b = sext16(w >> 16);
  return pack16(a < kNegInf16 ? kNegInf16 : a, b < kNegInf16 ? kNegInf16 : b);
#endif
}

// inclusive prefix maximum over the GS lanes of a window (one or two DPP rows)
template <int GS>
__host__ __device__ __forceinline__ i32 group_prefix_max(i32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  // the DPP operand fused into v_max: a lane without a source (bound_ctrl off) is not written and keeps its value;
  // s_nop 1 = the two wait states a DPP read needs after a VALU write of the register
  asm volatile(
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf"
      : "+v"(x));
  if constexpr (GS == 32)  // the upper row of each pair takes the lower row's total
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x));
  return x;
#else
  i32 o;
  o = sv::row_shr<1>(x, kVeryNeg);
  x = o > x ? o : x;
  o = sv::row_shr<2>(x, kVeryNeg);
  x = o > x ? o : x;
  o = sv::row_shr<4>(x, kVeryNeg);
  x = o > x ? o : x;
  o = sv::row_shr<8>(x, kVeryNeg);
  x = o > x ? o : x;
  if constexpr (GS == 32) {
    const int l = sv::lane();
    o = sv::bperm(x, (l & ~15) - 1);
    if (l & 16) x = o > x ? o : x;
  }
  return x;
#endif
}
// value of the lane below in the same window; the window's first lane gets `fill`
template <int GS>
__host__ __device__ __forceinline__ i32 group_shift1(i32 v, i32 fill) {
  if constexpr (GS == 16) {
    return sv::row_shr<1>(v, fill);
  } else {
#if defined(__HIP_DEVICE_COMPILE__)
    const i32 t = __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
#else
    const i32 t = sv::bperm(v, sv::lane() - 1);
#endif
    return (sv::lane() & (GS - 1)) == 0 ? fill : t;
  }
}

// k-th in-edge (among those inside the subgraph) of v, as a row index (rank + 1)
__host__ __device__ inline u32 poa3_nth_pred(const Poa2Slot& g, u32 v, u32 k, bool full) {
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

#if defined(__HIP_DEVICE_COMPILE__)
#define P3_RARE __attribute__((noinline))
#else
#define P3_RARE
#endif
// The ncell cells (columns j0 - 1 .., packed like the LDS read delivers them) of the k-th in-edge's row of `row`, for a
// predecessor that has left the LDS ring (one row in ~10^5: a long bubble ahead of it): from the int16 copy every row
// leaves in HBM.  Out of line on purpose — its registers and loads stay out of the row loop.
__host__ __device__ P3_RARE uint4 poa3_lost_cells(unsigned char* slot_mem, u32 nmax, u32 lmax, u32 band, u32 row, u32 k,
                                                  bool full, i32 j0, int ncell) {
  const Poa2Slot g = poa2_carve(slot_mem, nmax, lmax, band, true);
  const uint4 t = g.tb[row];
  u32 prow;
  if (k < 2) prow = (t.z >> (16 * k)) & 0xFFFFu;
  else if (k < 4) prow = (t.w >> (16 * (k - 2))) & 0xFFFFu;
  else prow = poa3_nth_pred(g, t.x >> 16, k, full);
  const i32 pbx = static_cast<i32>(g.tb[prow].x & 0xFFFFu);
  const i16* hrow = g.Hs + static_cast<size_t>(prow) * band;
  u32 wds[3] = {0, 0, 0};
  for (int u = 0; u < ncell; ++u) {
    const i32 c = j0 - pbx - 1 + u;
    const i32 v = (c >= 0 && c < static_cast<i32>(band)) ? static_cast<i32>(hrow[c]) : kNegInf16;
    wds[u >> 1] |= (static_cast<u32>(v) & 0xFFFFu) << (16 * (u & 1));
  }
  uint4 r;
  r.x = wds[0];
  r.y = wds[1];
  r.z = wds[2];
  r.w = 0;
  return r;
}

// Descriptor of predecessor row `pr` for the row whose computed-row index is cur_idx: int16(-band start - 1) | the ring
// row's first cell (slot * stride + pad) << 16, so that the predecessor's cell under column j is at
// (desc >> 16) + clamp(j + int16(desc)); kDescMiss if the row is no longer in the ring.  bi_cur / bi_prev: (band start |
// computed index << 16) of row pr as the current / the previous block of GS rows holds it.
template <class K>
__host__ __device__ __forceinline__ u32 poa3_desc(u32 pr, u32 r0, u32 cur_idx, u32 bi_cur, u32 bi_prev) {
  const bool in_cur = pr >= r0 + 1;
  const bool in_prev = !in_cur && pr + K::GS >= r0 + 1;
  const u32 pbi = in_cur ? bi_cur : bi_prev;
  const bool miss = !(in_cur || in_prev) || ((cur_idx - (pbi >> 16)) & 0xFFFFu) > static_cast<u32>(K::kRing);
  const u32 rel = (0u - (pbi & 0xFFFFu) - 1u) & 0xFFFFu;
  const u32 first = ((pbi >> 16) & (K::kRing - 1)) * K::kStride + K::kPad;  // <= 1146: 11 bits
  return rel | (first << 16) | (miss ? kDescMiss : 0u);
}

// ---- banded NW of one layer per group (the wave's windows in lockstep) -----------------------------------------------
// Group-uniform inputs: act (this group aligns a layer now), nn (graph nodes), full, the layer.  Outputs (group-uniform):
// best_row (0: the last column is in no end node's band).
template <class K>
__host__ __device__ P3_PHASE inline void poa3_dp(const Poa3Args A, Poa3Lds<K>& S, unsigned char* slot_mem, bool act, u32 nn,
                                                 bool full, const PoaLayer* Lp, u32 len, i32 lb, i32 span, u32& best_row) {
  constexpr int C = K::C, GS = K::GS, kBand = K::kBand, kPad = K::kPad, kStride = K::kStride;
  P3_ASSUME_GLOBAL(slot_mem);
  P3_ASSUME_GLOBAL(Lp);
  P3_ASSUME_GLOBAL(A.phase_cycles);
  P3_ASSUME_LDS(&S);
  const int lane = sv::lane();
  const int gl = lane & (GS - 1), gbase = lane & ~(GS - 1);
  Poa3Group<K>& Sg = S.g[lane / GS];
  u8* const bp_rows = poa2_carve(slot_mem, A.nmax, A.lmax, kBand, true).BP + C * gl;   // the two stores of every row
  i16* const hs_rows = poa2_carve(slot_mem, A.nmax, A.lmax, kBand, true).Hs + C * gl;
  i16* ring16 = reinterpret_cast<i16*>(Sg.u.ring32);
  const u32 w = len + 1;
  const i32 gp = A.gp;
  // -inf pads of the ring rows (the union is reused by the traceback of the previous layer)
  for (int idx = gl; idx < K::kRing * 2 * kPad; idx += GS) {
    const int r = idx / (2 * kPad), c = idx % (2 * kPad);
    ring16[r * kStride + (c < kPad ? c : kBand + c)] = static_cast<i16>(kNegInf16);
  }
  lds_order();
  // what a candidate adds to (predecessor cell << 6): (match | mismatch | gap) << 6, + 32 for a diagonal, + 15 - in-edge
  const i32 mD15 = A.m * 64 + 32 + 15, nD15 = A.n_ * 64 + 32 + 15, gV15 = gp * 64 + 15;
  i32 best_score = -0x7FFFFFFF;
  best_row = 0;
  u32 marked_before = 0;
  int m_bi_prev = 0;
  const u32 max_nn = static_cast<u32>(sv::wave_max(act ? static_cast<int>(nn) : 0));
  // the layer's band guide in registers (poa_layer_center without its loads)
  const u32* wayp = reinterpret_cast<const u32*>(Lp->way);
  const u32 way0 = wayp[0], way1 = wayp[1], way2 = wayp[2], way3 = wayp[3];
  auto way_at = [&](i32 idx) -> i32 {
    const u32 ws = idx < 4 ? (idx < 2 ? way0 : way1) : (idx < 6 ? way2 : way3);
    return static_cast<i32>((ws >> (16 * (idx & 1))) & 0xFFFFu);
  };
  auto center = [&](i32 x) -> i32 {
    x = x < 0 ? 0 : (x > span ? span : x);
    const i32 seg = (x * 8) / (span > 0 ? span : 1);
    const i32 sg = seg > 7 ? 7 : seg;
    const i32 x0 = (sg * span) / 8, x1 = ((sg + 1) * span) / 8;
    const i32 wa = sg == 0 ? 0 : way_at(sg - 1);
    const i32 wb = sg == 7 ? static_cast<i32>(len) : way_at(sg);
    return wa + (x - x0) * (wb - wa) / (x1 > x0 ? x1 - x0 : 1);
  };
  // Metadata of a block's GS rows, one row per lane of the group, is a chain of dependent gathers (order -> node
  // fields and in-edge tails -> ranks of the tails).  The chain of block b + 1 is issued in three stages spread over
  // the row loop of block b, so its latency hides behind a third of the block's DP rows per stage.
  bool nx_ok = false;
  int nx_v = 0;
  u32 nx_marked = 0, nx_code = 0, nx_outc = 1, nx_c = 0, nx_bpos = 0, nx_t01 = 0, nx_t23 = 0;
  u32 nx_rk0 = 0, nx_rk1 = 0, nx_rk2 = 0, nx_rk3 = 0, nx_m0 = 1, nx_m1 = 1, nx_m2 = 1, nx_m3 = 1;
  auto stage1 = [&](u32 rbase) {
    const Poa2Slot g = poa2_carve(opaque(slot_mem), A.nmax, A.lmax, kBand, true);
    nx_ok = act && rbase + static_cast<u32>(gl) < nn;
    nx_v = nx_ok ? static_cast<int>(g.order[rbase + gl]) : 0;
  };
  auto stage2 = [&]() {
    const Poa2Slot g = poa2_carve(opaque(slot_mem), A.nmax, A.lmax, kBand, true);
    nx_marked = 0;
    if (nx_ok) {
      nx_marked = full ? 1u : g.mark[nx_v];
      nx_code = g.code[nx_v];
      nx_outc = full ? g.out_cnt[nx_v] : g.sub_out[nx_v];
      nx_c = g.in_cnt[nx_v];
      nx_bpos = g.bpos[nx_v];
      const u32* tp = reinterpret_cast<const u32*>(g.in_tail + static_cast<size_t>(nx_v) * kPoaMaxIn);
      nx_t01 = tp[0];
      nx_t23 = tp[1];
    }
  };
  auto stage3 = [&]() {
    const Poa2Slot g = poa2_carve(opaque(slot_mem), A.nmax, A.lmax, kBand, true);
    if (nx_ok && nx_marked) {  // tails beyond the in-degree are stale memory: clamp them to node 0
      const u32 t0 = nx_c > 0 ? nx_t01 & 0xFFFFu : 0u, t1 = nx_c > 1 ? nx_t01 >> 16 : 0u;
      const u32 t2 = nx_c > 2 ? nx_t23 & 0xFFFFu : 0u, t3 = nx_c > 3 ? nx_t23 >> 16 : 0u;
      nx_rk0 = g.rank_of[t0];
      nx_rk1 = g.rank_of[t1];
      nx_rk2 = g.rank_of[t2];
      nx_rk3 = g.rank_of[t3];
      if (!full) {  // (the values are only looked at when the block starts: no wait for them here)
        nx_m0 = g.mark[t0];
        nx_m1 = g.mark[t1];
        nx_m2 = g.mark[t2];
        nx_m3 = g.mark[t3];
      }
    }
  };
  stage1(0);
  stage2();
  stage3();
  for (u32 r0 = 0; r0 < max_nn; r0 += GS) {
    // ---- the block's GS rows, one per lane of the group: metadata, traceback row table, predecessor descriptors ----
    const int m_v = nx_v;
    int m_np = 0, m_code = 0, m_outc = 1, m_marked = 0, m_b = 0;
    u32 p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    if (nx_ok) {
      const Poa2Slot g = poa2_carve(opaque(slot_mem), A.nmax, A.lmax, kBand, true);
      m_marked = nx_marked ? 1 : 0;
      if (m_marked) {
        m_code = static_cast<int>(nx_code);
        m_outc = static_cast<int>(nx_outc);
        auto take = [&](u32 pr) {
          if (m_np == 0) p0 = pr;
          else if (m_np == 1) p1 = pr;
          else if (m_np == 2) p2 = pr;
          else if (m_np == 3) p3 = pr;
          ++m_np;
        };
        if (nx_c > 0 && (full || nx_m0)) take(nx_rk0 + 1);
        if (nx_c > 1 && (full || nx_m1)) take(nx_rk1 + 1);
        if (nx_c > 2 && (full || nx_m2)) take(nx_rk2 + 1);
        if (nx_c > 3 && (full || nx_m3)) take(nx_rk3 + 1);
        for (u32 k = 4; k < nx_c; ++k) {  // rare
          const u32 t = g.in_tail[m_v * kPoaMaxIn + k];
          if (full || g.mark[t]) take(static_cast<u32>(g.rank_of[t]) + 1);
        }
        i32 b = center(static_cast<i32>(nx_bpos) - lb) - kBand / 2;
        const i32 bmax = static_cast<i32>(w) - kBand;
        b = b > bmax ? bmax : b;
        b = b < 0 ? 0 : b;
        m_b = b;
      }
      uint4 t;
      t.x = static_cast<u32>(m_b) | (static_cast<u32>(m_v) << 16);
      t.y = static_cast<u32>(m_np);
      t.z = p0 | (p1 << 16);
      t.w = p2 | (p3 << 16);
      g.tb[r0 + gl + 1] = t;
    }
    // ring slots are handed out per COMPUTED row, so rows outside the layer's subgraph do not age the ring
    const u32 grp = static_cast<u32>(sv::ballot(m_marked != 0) >> gbase) & (GS == 32 ? 0xFFFFFFFFu : 0xFFFFu);
    const u32 idx_in = (marked_before + static_cast<u32>(__builtin_popcount(grp & ((1u << gl) - 1u)))) & 0xFFFFu;
    marked_before += static_cast<u32>(__builtin_popcount(grp));
    const int m_bi = m_b | static_cast<int>(idx_in << 16);
    u32 desc0 = 0, desc1 = 0, desc2 = 0, desc3 = 0;
    {
      auto owner = [&](u32 pr) { return gbase | static_cast<int>((pr - 1) & static_cast<u32>(GS - 1)); };
      const u32 bc0 = static_cast<u32>(sv::bperm(m_bi, owner(p0)));
      const u32 bp0 = static_cast<u32>(sv::bperm(m_bi_prev, owner(p0)));
      desc0 = m_np == 0 ? kDescVirtual : poa3_desc<K>(p0, r0, idx_in, bc0, bp0);
      if (sv::any(m_np > 1)) {
        const u32 bc1 = static_cast<u32>(sv::bperm(m_bi, owner(p1)));
        const u32 bp1 = static_cast<u32>(sv::bperm(m_bi_prev, owner(p1)));
        desc1 = poa3_desc<K>(p1, r0, idx_in, bc1, bp1);
      }
      if (sv::any(m_np > 2)) {
        const u32 bc2 = static_cast<u32>(sv::bperm(m_bi, owner(p2)));
        const u32 bp2 = static_cast<u32>(sv::bperm(m_bi_prev, owner(p2)));
        desc2 = poa3_desc<K>(p2, r0, idx_in, bc2, bp2);
        const u32 bc3 = static_cast<u32>(sv::bperm(m_bi, owner(p3)));
        const u32 bp3 = static_cast<u32>(sv::bperm(m_bi_prev, owner(p3)));
        desc3 = poa3_desc<K>(p3, r0, idx_in, bc3, bp3);
      }
    }
    // marked | #in-edges << 1 | code << 6 | end node << 8 | band start << 9 | ring slot << 19
    const int m_w0 = m_marked | ((m_np > 31 ? 31 : m_np) << 1) | (m_code << 6) | ((m_outc == 0 ? 1 : 0) << 8) | (m_b << 9) |
                     static_cast<int>((idx_in & (K::kRing - 1)) << 19);
    u32 W0n = static_cast<u32>(sv::bperm(m_w0, gbase));  // row 0's words; row ri + 1's are fetched during row ri
    u32 d0n = static_cast<u32>(sv::bperm(static_cast<int>(desc0), gbase));
    // ---- the rows, in order; row ri of every group in the same iteration ----
    auto do_row = [&](int ri) {
      P3_MARK("row_begin");
      const int src = gbase | ri;
      const u32 W0 = W0n;
      u32 d = d0n;
      W0n = static_cast<u32>(sv::bperm(m_w0, gbase | ((ri + 1) & (GS - 1))));
      d0n = static_cast<u32>(sv::bperm(static_cast<int>(desc0), gbase | ((ri + 1) & (GS - 1))));
      const bool actv = (W0 & 1u) != 0;
      if (!sv::any(actv)) return;
      const u32 np = (W0 >> 1) & 31u;
      const u32 npe = np ? np : 1u;  // no in-edge inside the subgraph: the virtual start row
      const u32 vc = (W0 >> 6) & 3u;
      const bool endn = ((W0 >> 8) & 1u) != 0;
      const i32 b = static_cast<i32>((W0 >> 9) & 1023u);
      const u32 slot = (W0 >> 19) & 15u;
      const u32 row = r0 + static_cast<u32>(ri) + 1;
      const i32 j0 = b + C * gl;
      // the layer's codes under the lane's columns (column j compares with position j - 1)
      u32 chars;
      if constexpr (C == 1) {
        chars = Sg.seq_pad[3 + j0];
      } else {
        const u32 sa = static_cast<u32>(3 + j0);
        const u32* sq = reinterpret_cast<const u32*>(Sg.seq_pad);
        chars = funnel_shr(sq[(sa >> 2) + 1], sq[sa >> 2], 8u * (sa & 3u));
      }
      i32 subD[C], best[C];
#pragma unroll
      for (int t = 0; t < C; ++t) subD[t] = ((chars >> (8 * t)) & 0xFFu) == vc ? mD15 : nD15;
      // the C + 1 adjacent cells of predecessor row `dd` (the row's k-th in-edge) starting under this lane's column
      // j0 - 1, packed two to a word as the LDS read delivers them
      auto pred_cells = [&](u32 dd, bool valid, u32 k, u32(&cw)[K::kNW]) {
        i32 start = j0 + sext16(dd);  // the predecessor's cell under this lane's column j0 - 1
        start = start < -(C + 1) ? -(C + 1) : (start > kBand ? kBand : start);
        const u32 base = ((dd >> 16) & 0x7FFu) + static_cast<u32>(start);
        const u32 dw = base >> 1, par16 = (base & 1u) * 16u;
        u32 x[K::kNR];
#pragma unroll
        for (int i = 0; i < K::kNR; ++i) x[i] = Sg.u.ring32[dw + i];
#pragma unroll
        for (int i = 0; i < K::kNW; ++i) cw[i] = i + 1 < K::kNR ? funnel_shr(x[i + 1 < K::kNR ? i + 1 : i], x[i], par16) : x[i] >> par16;
        const bool lost = valid && (dd & kDescMiss) != 0;
        if (sv::any(lost)) {  // the row has left the ring
          P3_NO_IF_CONVERSION();
          sv::sync();  // the rows' stores have landed
#if !defined(__HIP_DEVICE_COMPILE__)
          if (lost && gl == 0) ++g_emu_lost_rows;
#endif
          if (lost) {
            const uint4 r = poa3_lost_cells(slot_mem, A.nmax, A.lmax, kBand, row, k, full, j0, C + 1);
            cw[0] = r.x;
            if constexpr (K::kNW > 1) cw[1] = r.y;
            if constexpr (K::kNW > 2) cw[2] = r.z;
          }
        }
      };
      // cell u of the packed words, * 64 + add
      auto cell_plus = [&](const u32(&cw)[K::kNW], int u, i32 add) -> i32 {
        return (u & 1) ? cell_x64_plus<true>(cw[u >> 1], add) : cell_x64_plus<false>(cw[u >> 1], add);
      };
      P3_MARK("edges_begin");
      {  // in-edge 0 (or the virtual start row): every computed row has it
        u32 cw[K::kNW];
        pred_cells(d, actv, 0u, cw);
        const bool virt = (d & kDescVirtual) != 0;
        if (sv::any(actv && virt)) {  // H[0][j] = j * g
          P3_NO_IF_CONVERSION();
          i32 vcl[2 * K::kNW];
#pragma unroll
          for (int u = 0; u < 2 * K::kNW; ++u) {
            const i32 jc = j0 - 1 + u;
            vcl[u] = jc >= 0 ? mul24(jc, gp) : kNegInf16;
          }
#pragma unroll
          for (int i = 0; i < K::kNW; ++i) cw[i] = virt ? pack16(vcl[2 * i], vcl[2 * i + 1]) : cw[i];
        }
#pragma unroll
        for (int t = 0; t < C; ++t) {
          const i32 dd = cell_plus(cw, t, subD[t]), vv = cell_plus(cw, t + 1, gV15);
          best[t] = dd > vv ? dd : vv;
        }
      }
      if (sv::any(actv && npe > 1u)) {
        P3_NO_IF_CONVERSION();
        for (u32 k = 1;; ++k) {
          const bool vk = actv && k < npe;
          if (!sv::any(vk)) break;
          if (k == 1) d = static_cast<u32>(sv::bperm(static_cast<int>(desc1), src));
          else if (k == 2) d = static_cast<u32>(sv::bperm(static_cast<int>(desc2), src));
          else if (k == 3) d = static_cast<u32>(sv::bperm(static_cast<int>(desc3), src));
          else {  // rare: the row's owner looks the in-edge up, the group resolves its ring slot
            u32 prk = 0;
            if (vk && gl == ri)
              prk = poa3_nth_pred(poa2_carve(opaque(slot_mem), A.nmax, A.lmax, kBand, true), static_cast<u32>(m_v), k, full);
            prk = static_cast<u32>(sv::bperm(static_cast<int>(prk), src));
            const int sl = gbase | static_cast<int>((prk - 1) & static_cast<u32>(GS - 1));
            const u32 bc = static_cast<u32>(sv::bperm(m_bi, sl));
            const u32 bp = static_cast<u32>(sv::bperm(m_bi_prev, sl));
            const u32 cur_idx = static_cast<u32>(sv::bperm(m_bi, src)) >> 16;
            d = poa3_desc<K>(prk, r0, cur_idx, bc, bp);
          }
          u32 cw[K::kNW];
          pred_cells(d, vk, k, cw);
          // a group without a k-th in-edge subtracts 2^29 instead of k: its candidates never win
          const i32 off = vk ? static_cast<i32>(k) : 0x20000000;
          const i32 gk = gV15 - off;
#pragma unroll
          for (int t = 0; t < C; ++t) {
            const i32 dd = cell_plus(cw, t, subD[t] - off), vv = cell_plus(cw, t + 1, gk);
            best[t] = max3(best[t], dd, vv);
          }
        }
      }
      P3_MARK("edges_end");
      // spoa's traceback priority: diagonal (first in-edge reaching the max), vertical, horizontal
      const i32 jg0 = mul24(j0, gp);
      i32 sc[C], y[C];
#pragma unroll
      for (int t = 0; t < C; ++t) {
        sc[t] = best[t] >> 6;
        y[t] = sc[t] - (jg0 + t * gp);
      }
#pragma unroll
      for (int t = 1; t < C; ++t) y[t] = y[t] > y[t - 1] ? y[t] : y[t - 1];
      const i32 s = group_prefix_max<GS>(y[C - 1]);
      i32 ex = s;
      if constexpr (C > 1) ex = group_shift1<GS>(s, kVeryNeg);  // lanes to the left of this one
      // backpointer byte: 64 = horizontal, else the winner's key bits (diagonal << 5 | 15 - in-edge)
      i32 hh[C];
      u32 codes = 0;
#pragma unroll
      for (int t = 0; t < C; ++t) {
        const i32 yt = y[t] > ex ? y[t] : ex;
        const i32 h = yt + (jg0 + t * gp);
        const u32 code = h > sc[t] ? 64u : (static_cast<u32>(best[t]) & 63u);
        hh[t] = h;
        codes |= code << (8 * t);
      }
      const u32 first = slot * kStride + kPad + static_cast<u32>(C * gl);  // this lane's cells in the row's ring slot
      u32 h01 = 0, h23 = 0;
      if constexpr (C == 1) {
        h01 = static_cast<u32>(hh[0] < kNegInf16 ? kNegInf16 : hh[0]) & 0xFFFFu;
        if (actv) {
          ring16[first] = static_cast<i16>(h01);
          hs_rows[static_cast<size_t>(row) * kBand] = static_cast<i16>(h01);
          bp_rows[static_cast<size_t>(row) * kBand] = static_cast<u8>(codes);
        }
      } else if constexpr (C == 2) {
        h01 = clamp_pair(pack16(hh[0], hh[1]));
        if (actv) {
          Sg.u.ring32[first >> 1] = h01;
          *reinterpret_cast<u32*>(hs_rows + static_cast<size_t>(row) * kBand) = h01;
          *reinterpret_cast<u16*>(bp_rows + static_cast<size_t>(row) * kBand) = static_cast<u16>(codes);
        }
      } else {
        h01 = clamp_pair(pack16(hh[0], hh[1]));
        h23 = clamp_pair(pack16(hh[2], hh[3]));
        if (actv) {
          Sg.u.ring32[first >> 1] = h01;
          Sg.u.ring32[(first >> 1) + 1] = h23;
          *reinterpret_cast<uint2*>(hs_rows + static_cast<size_t>(row) * kBand) = uint2{h01, h23};
          *reinterpret_cast<u32*>(bp_rows + static_cast<size_t>(row) * kBand) = codes;
        }
      }
      if (sv::any(actv && endn)) {  // an end node: score of the last column if the band has it
        const i32 idx = static_cast<i32>(w) - 1 - b;
        const int tt = idx & (C - 1);
        const i32 mine = sext16(((tt & 2) ? h23 : h01) >> (16 * (tt & 1)));
        const i32 sce = sv::bperm(mine, gbase | ((idx / C) & (GS - 1)));
        if (actv && endn && idx >= 0 && idx < kBand && sce > best_score) {
          best_score = sce;
          best_row = row;
        }
      }
      lds_order();
      P3_MARK("row_end");
    };
    // the next block's gather chain, one stage per third of the block (beyond the last block every lane is switched off)
    stage1(r0 + GS);
    for (int ri = 0; ri < GS / 3; ++ri) do_row(ri);
    stage2();
    for (int ri = GS / 3; ri < 2 * GS / 3; ++ri) do_row(ri);
    stage3();
    for (int ri = 2 * GS / 3; ri < GS; ++ri) do_row(ri);
    m_bi_prev = m_bi;
  }
  // work counters: rows of this layer's (sub)graph x layer length = the cells spoa's full NW computes | cells in the band
  if (gl == 0 && act && A.phase_cycles) {
    sv::atomic_add(&A.phase_cycles[6], static_cast<unsigned long long>(marked_before) * len);
    sv::atomic_add(&A.phase_cycles[7], static_cast<unsigned long long>(marked_before) * (w < static_cast<u32>(kBand) ? w : static_cast<u32>(kBand)));
  }
}

// ---- traceback of the wave's windows in lockstep -------------------------------------------------------------------
template <class K>
__host__ __device__ P3_PHASE inline void poa3_traceback(const Poa3Args A, Poa3Lds<K>& S, unsigned char* slot, bool act, u32 nn,
                                                        bool full, u32 len, u32 best_row, u32& bad, u32& band_hit) {
  constexpr int GS = K::GS, kBand = K::kBand, kTbRows = K::kTbRows;
  P3_ASSUME_GLOBAL(slot);
  P3_ASSUME_LDS(&S);
  const int lane = sv::lane();
  const int gl = lane & (GS - 1);
  Poa3Group<K>& Sg = S.g[lane / GS];
  const Poa2Slot g = poa2_carve(slot, A.nmax, A.lmax, kBand, true);
  const u32 w = len + 1;
  bad = 0;
  band_hit = 0;
  u32 i = act ? best_row : 0;
  i32 j = static_cast<i32>(w) - 1;
  bool done = !act || i == 0;
  u32 cur_blk = 0xFFFFFFFFu;
  u32 steps = 0;
  const u32 max_steps = A.nmax + A.lmax + 2;
  const u32 n_rows_total = nn + 1;
  // A block (32 rows of backpointers + their row table) is fetched into registers one block AHEAD of the walk — the
  // walk moves up through the rows, so while it is inside block b the loads of block b - 1 are in flight — and only
  // copied into the group's LDS when the walk gets there: no global-memory round trip on the walk's critical path.
  // (named registers, not arrays: an array that lives across the walk loop ends up in scratch memory)
  uint4 pb0{}, pb1{}, pb2{}, pb3{}, pb4{}, pb5{}, pb6{}, pb7{}, pt0{}, pt1{};
  u32 pf_blk = 0xFFFFFFFFu;
  auto fetch = [&](u32 b) {
    const u32 row0 = b * kTbRows + 1;
    const u32 nrows = n_rows_total - row0 < static_cast<u32>(kTbRows) ? n_rows_total - row0 : static_cast<u32>(kTbRows);
    const uint4* bsrc = reinterpret_cast<const uint4*>(g.BP + static_cast<size_t>(row0) * kBand);
    const u32 ugl = static_cast<u32>(gl);
    auto chunk = [&](u32 it) -> uint4 {
      const u32 c = it * GS + ugl;  // 16-byte chunk of the block: row c * 16 / band
      return bsrc[(c * 16) / kBand < nrows ? c : 0];
    };
    pb0 = chunk(0);
    if constexpr (K::kBpChunks > 1) pb1 = chunk(1);
    if constexpr (K::kBpChunks > 2) pb2 = chunk(2);
    if constexpr (K::kBpChunks > 3) pb3 = chunk(3);
    if constexpr (K::kBpChunks > 4) pb4 = chunk(4);
    if constexpr (K::kBpChunks > 5) pb5 = chunk(5);
    if constexpr (K::kBpChunks > 6) pb6 = chunk(6);
    if constexpr (K::kBpChunks > 7) pb7 = chunk(7);
    pt0 = g.tb[row0 + (ugl < nrows ? ugl : 0)];
    if constexpr (K::kTbPerLane > 1) pt1 = g.tb[row0 + (GS + ugl < nrows ? GS + ugl : 0)];
    pf_blk = b;
  };
  while (sv::any(!done)) {
    P3_MARK("tb_top");
    const u32 blk = done ? cur_blk : (i - 1) >> 5;
    const bool need = !done && blk != cur_blk;
    if (sv::any(need)) {
      lds_order();
      if (need) {
        if (pf_blk != blk) fetch(blk);
        uint4* bdst = reinterpret_cast<uint4*>(Sg.u.tr.bp) + gl;
        bdst[0] = pb0;
        if constexpr (K::kBpChunks > 1) bdst[GS] = pb1;
        if constexpr (K::kBpChunks > 2) bdst[2 * GS] = pb2;
        if constexpr (K::kBpChunks > 3) bdst[3 * GS] = pb3;
        if constexpr (K::kBpChunks > 4) bdst[4 * GS] = pb4;
        if constexpr (K::kBpChunks > 5) bdst[5 * GS] = pb5;
        if constexpr (K::kBpChunks > 6) bdst[6 * GS] = pb6;
        if constexpr (K::kBpChunks > 7) bdst[7 * GS] = pb7;
        u32* tdst = Sg.u.tr.tb + 3 * gl;
        tdst[0] = pt0.x;
        tdst[1] = pt0.z;
        tdst[2] = pt0.w;
        if constexpr (K::kTbPerLane > 1) {
          tdst[3 * GS] = pt1.x;
          tdst[3 * GS + 1] = pt1.z;
          tdst[3 * GS + 2] = pt1.w;
        }
        cur_blk = blk;
        if (blk > 0) fetch(blk - 1);
      }
      lds_order();
    }
    P3_MARK("tb_step");
    if (!done) {
      if (++steps > max_steps) {
        bad = 6;
        done = true;
      } else {
        const u32 l = (i - 1) & (kTbRows - 1);
        const u32 x = Sg.u.tr.tb[l * 3], z01 = Sg.u.tr.tb[l * 3 + 1];  // one ds_read2
        const i32 bt = static_cast<i32>(x & 0xFFFFu);
        const u32 node = x >> 16;
        const i32 idx = j - bt;
        if (idx < 0 || idx >= kBand) {  // the path left the stored band: the alignment does not fit this band width
          band_hit = 1;
          done = true;
        } else {
          if ((idx < 2 && bt > 0) || (idx > kBand - 3 && bt + kBand < static_cast<i32>(w))) band_hit = 1;
          const u32 code = Sg.u.tr.bp[l * kBand + static_cast<u32>(idx)];
          if (code == 64u) {
            if (j == 0) {
              bad = 6;
              done = true;
            } else {
              --j;  // insertion: pos_node[j] stays kNone
            }
          } else {
            const u32 k = 15u - (code & 15u);
            u32 pr;
            if (k < 2) pr = (z01 >> (16 * k)) & 0xFFFFu;
            else if (k < 4) pr = (Sg.u.tr.tb[l * 3 + 2] >> (16 * (k - 2))) & 0xFFFFu;
            else pr = poa3_nth_pred(g, node, k, full);
            if (code & 32u) {  // diagonal
              if (j == 0) {
                bad = 6;
                done = true;
              } else {
                --j;
                if (gl == 0) g.pos_node[j] = static_cast<u16>(node);
              }
            }
            if (!done) {
              i = pr;
              if (i == 0) done = true;  // on the virtual row only insertions remain: pos_node already says kNone
            }
          }
        }
      }
    }
  }
}

// ---- wave-wide per-window steps (as in poa2.hip) -------------------------------------------------------------------
__host__ __device__ inline void poa3_copy_backbone(const Poa3Args& A, const PoaWindow& win, const PoaLayer& bb, u8* out,
                                                   u32* out_len) {
  const int lane = sv::lane();
  const u32 n = bb.len < win.out_cap ? bb.len : win.out_cap;
  for (u32 i = lane; i < n; i += 64) out[i] = static_cast<u8>(poa_layer_code(A.src, bb, i));
  if (lane == 0) *out_len = n;
}

// window set-up: 0 = backbone returned (< 3 sequences), 4 = beyond a length limit (backbone returned), 1 = graph built
__host__ __device__ inline u32 poa3_init_window(const Poa3Args& A, const PoaWindow& win, Poa2Slot& g, u32 wi, u32& n_nodes,
                                                u32& n_eff) {
  const int lane = sv::lane();
  const PoaLayer bb = A.layers[win.layer_first];
  const u32 blen = bb.len;
  n_eff = win.n_layers;
  if (A.src.layer_ok) {  // layers dropped by racon's mean-quality filter do not count as sequences of the window
    u32 cnt = 0;
    for (u32 i = 1 + lane; i < win.n_layers; i += 64) cnt += A.src.layer_ok[win.layer_first + i] ? 1u : 0u;
    n_eff = 1 + sv::wave_sum(cnt);
  }
  n_nodes = 0;
  if (n_eff < 3) {
    poa3_copy_backbone(A, win, bb, A.out + win.out_off, A.out_len + wi);
    return 0;
  }
  if (blen == 0 || blen > A.nmax || blen > A.lmax) {
    poa3_copy_backbone(A, win, bb, A.out + win.out_off, A.out_len + wi);
    return 4;
  }
  // backbone graph (spoa AddAlignment with an empty alignment)
  n_nodes = blen;
  for (u32 i = lane; i < blen; i += 64) {
    g.code[i] = static_cast<u8>(poa_layer_code(A.src, bb, i));
    g.al_cnt[i] = 0;
    g.visits[i] = blen >= 2 ? 1 : 0;
    g.rank_of[i] = static_cast<u16>(i);
    g.order[i] = static_cast<u16>(i);
    g.bpos[i] = static_cast<u16>(i);
    const i32 wgt = poa_layer_weight(A.src, bb, i);
    if (i > 0) {
      const i32 wp = poa_layer_weight(A.src, bb, i - 1);
      g.in_cnt[i] = 1;
      g.in_tail[i * kPoaMaxIn] = static_cast<u16>(i - 1);
      g.in_w[i * kPoaMaxIn] = wp + wgt;
    } else {
      g.in_cnt[i] = 0;
    }
    g.out_cnt[i] = i + 1 < blen ? 1 : 0;
  }
  sv::sync();
  return 1;
}

// spoa AddAlignment, one sequence position per lane, + the incremental order rebuild.  Returns 0 or the failure code.
template <class K>
__host__ __device__ inline u32 poa3_add_alignment(const Poa3Args& A, Poa2Slot& g, Poa3Group<K>& Sg, const PoaLayer& L,
                                                  u32& n_nodes, unsigned long long& t_add, unsigned long long& t_ord) {
  const int lane = sv::lane();
  const u32 len = L.len;
  const u32 nmax = A.nmax, lmax = A.lmax;
  const u32 lb = L.begin;
  unsigned long long t0 = sv::clock();
  const u32 n_old = n_nodes;
  u32 first_p = 0xFFFFFFFFu;
  for (u32 p0 = 0; p0 < len && first_p == 0xFFFFFFFFu; p0 += 64) {
    const u32 p = p0 + lane;
    const unsigned long long bal = sv::ballot(p < len && g.pos_node[p] != kNone3);
    if (bal) first_p = p0 + static_cast<u32>(__builtin_ctzll(bal));
  }
  // New nodes anchored after a column (aligned group) go after ALL its members; the unaligned prefix goes
  // before all members of the first column.
  u32 carry_slot = n_old, carry_b = lb;
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
    const u32 an = valid ? g.pos_node[p] : kNone3;
    const u32 letter = valid ? Sg.seq_pad[4 + p] : 0u;
    const bool has = valid && an != kNone3;
    u32 tgt = kNone3, gslot = 0, gb = 0, ac = 0;
    if (has) {
      u32 rmax = g.rank_of[an];
      ac = g.al_cnt[an];
      if (g.code[an] == letter) tgt = an;
      for (u32 k = 0; k < ac; ++k) {
        const u32 kt = g.al[an * 4 + k];
        const u32 rk = g.rank_of[kt];
        rmax = rk > rmax ? rk : rmax;
        if (tgt == kNone3 && g.code[kt] == letter) tgt = kt;
      }
      gslot = rmax + 1;
      gb = g.bpos[an];
    }
    // order slot / backbone coordinate of the last aligned position at or before p
    const unsigned long long bal = sv::ballot(has);
    const unsigned long long below = bal & (lane == 63 ? ~0ULL : ((2ULL << lane) - 1ULL));
    const int src = below ? 63 - __builtin_clzll(below) : 0;
    const u32 s_sh = static_cast<u32>(sv::bperm(static_cast<int>(gslot), src));
    const u32 b_sh = static_cast<u32>(sv::bperm(static_cast<int>(gb), src));
    const u32 fslot = below ? s_sh : carry_slot;
    const u32 fb = below ? b_sh : carry_b;
    {
      const int top = bal ? 63 - __builtin_clzll(bal) : 0;
      const u32 cs = static_cast<u32>(sv::rl(static_cast<int>(gslot), top));
      const u32 cb = static_cast<u32>(sv::rl(static_cast<int>(gb), top));
      if (bal) {
        carry_slot = cs;
        carry_b = cb;
      }
    }
    const bool is_new = valid && tgt == kNone3;
    const unsigned long long nb = sv::ballot(is_new);
    const u32 cnt = static_cast<u32>(__builtin_popcountll(nb));
    if (n_old + total_new + cnt > nmax || total_new + cnt > lmax) {
      ok = 0;
      why = 2;
      break;
    }
    if (is_new) {
      const u32 t = total_new + static_cast<u32>(__builtin_popcountll(nb & ((1ULL << lane) - 1ULL)));
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
      g.tgt[p] = static_cast<u16>(tgt);
      if (len >= 2) g.visits[tgt] += 1;
    }
  }
  sv::sync();
  if (ok) {
    for (u32 p0 = 0; p0 < len; p0 += 64) {
      const u32 p = p0 + lane;
      bool okl = true;
      if (p >= 1 && p < len)
        okl = poa_add_edge(g, g.tgt[p - 1], g.tgt[p],
                           static_cast<i32>(static_cast<u8>(poa_layer_weight(A.src, L, p - 1))) +
                               static_cast<i32>(static_cast<u8>(poa_layer_weight(A.src, L, p))));
      if (sv::ballot(!okl)) {
        ok = 0;
        why = 3;
      }
    }
  }
  sv::sync();
  if (!ok) return why;
  const u32 n_new = total_new;
  n_nodes = n_old + n_new;
  t_add += sv::clock() - t0;
  t0 = sv::clock();
  // order rebuild: old rank r -> r + #(new slots <= r); t-th new node -> slot_t + t
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
    sv::sync();
    for (u32 r = lane; r < n_nodes; r += 64) {
      const u32 v = g.order2[r];
      g.order[r] = static_cast<u16>(v);
      g.rank_of[v] = static_cast<u16>(r);
    }
    sv::sync();
  }
  t_ord += sv::clock() - t0;
  return 0;
}

// Consensus of a finished window: spoa's heaviest bundle with the node scores in the WAVE's LDS (all four groups'
// buffers: the DP of every window of the wave is over by now), first four in-edges of 64 nodes at a time in registers
// (as poa2_consensus), then branch completion + racon's coverage trim on lane 0 and a parallel output copy.
template <class K>
__host__ __device__ inline void poa3_consensus(Poa2Slot& g, u32 n_nodes, u32 nmax, const PoaWindow& win, int trim, Poa3Lds<K>& S,
                                               u8* out, u32* out_len) {
  const int lane = sv::lane();
  constexpr u32 kCap = sizeof(Poa3Lds<K>) / 4;
  i32* lsc = reinterpret_cast<i32*>(&S);
  i32 maxn = -1;
  if (n_nodes > kCap) {
    if (lane == 0) maxn = poa_consensus_scores_lane0(g, n_nodes);
    maxn = sv::rfl(maxn);
  } else {
    i32 max_sc = 0;
    const u32 nn = n_nodes;
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
        const int li = static_cast<int>(l);
        const u32 it = static_cast<u32>(sv::rl(m_it, li));
        const u32 c = static_cast<u32>(sv::rl(m_c, li));
        const u32 t01 = static_cast<u32>(sv::rl(m_t01, li)), t23 = static_cast<u32>(sv::rl(m_t23, li));
        const i32 w0 = sv::rl(m_w0, li), w1 = sv::rl(m_w1, li), w2 = sv::rl(m_w2, li), w3 = sv::rl(m_w3, li);
        i32 sc = -1, pd = -1, pd_sc = 0;
        for (u32 k = 0; k < c; ++k) {
          i32 wgt, t;
          if (k < 4) {
            t = static_cast<i32>(((k < 2 ? t01 : t23) >> (16 * (k & 1))) & 0xFFFFu);
            wgt = k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : w3));
          } else {
            wgt = g.in_w[static_cast<size_t>(it) * kPoaMaxIn + k];
            t = static_cast<i32>(g.in_tail[static_cast<size_t>(it) * kPoaMaxIn + k]);
          }
          const i32 st = lsc[t];
          if (sc < wgt || (sc == wgt && pd_sc <= st)) {
            sc = wgt;
            pd = t;
            pd_sc = st;
          }
        }
        if (pd != -1) sc += pd_sc;
        lds_order();  // every lane has read the scores it needs before this node's is written
        if (lane == 0) {
          lsc[it] = sc;
          g.scores[it] = sc;
          g.preds[it] = pd;
        }
        lds_order();
        if (maxn == -1 || max_sc < sc) {
          maxn = static_cast<i32>(it);
          max_sc = sc;
        }
      }
    }
  }
  sv::sync();  // scores / predecessors in HBM visible to lane 0's branch completion and traceback
  u32 cl = 0;
  i32 begin = 0, end = -1;
  if (lane == 0) poa_consensus_trace_lane0(g, n_nodes, nmax, win, trim, maxn, &cl, &begin, &end);
  cl = static_cast<u32>(sv::rfl(static_cast<int>(cl)));
  begin = sv::rfl(begin);
  end = sv::rfl(end);
  sv::sync();  // g.stack
  i32 n_out = end - begin + 1;
  if (n_out < 0) n_out = 0;
  if (static_cast<u32>(n_out) > win.out_cap) n_out = static_cast<i32>(win.out_cap);
  for (i32 p = lane; p < n_out; p += 64) out[p] = g.code[g.stack[cl - 1 - static_cast<u32>(begin + p)]];
  if (lane == 0) *out_len = static_cast<u32>(n_out);
}

// ---- one persistent wave: takes four windows at a time ------------------------------------------------------------
enum : u32 { kIdle = 0, kRunning = 1, kLayersDone = 2, kFinal = 3, kFailed = 4 };

template <class K>
__host__ __device__ inline void poa3_wave(const Poa3Args& A, Poa3Lds<K>& S, u32 slot0) {
  constexpr int kG = K::G, GS = K::GS, kBand = K::kBand;
  const int lane = sv::lane();
  const int q = lane / GS;
  unsigned long long t_sub = 0, t_dp = 0, t_tb = 0, t_add = 0, t_ord = 0, t_cons = 0, t0 = 0;
  unsigned char* const my_slot = A.scratch + static_cast<size_t>(slot0 + q) * A.slot_bytes;  // per lane: its group's window
  for (;;) {
    u32 first = 0;
    if (lane == 0) first = sv::atomic_add(A.next, static_cast<u32>(kG));
    first = static_cast<u32>(sv::rfl(static_cast<int>(first)));
    if (first >= A.n_windows) break;
    // group-uniform state of the lane's window
    const u32 pos = first + static_cast<u32>(q);
    const bool have = pos < A.n_windows;
    const u32 wi = have ? (A.sched ? A.sched[pos] : pos) : 0u;
    const PoaWindow win = A.windows[wi];
    u32 phase = have ? kRunning : kIdle;
    u32 status = 0, nn = 0, n_eff = 0, li = 1;
    auto window_of = [&](int q2, u32& wi2) -> PoaWindow {  // group q2's window as wave-uniform values
      PoaWindow wq;
      wq.layer_first = static_cast<u32>(sv::rl(static_cast<int>(win.layer_first), q2 * GS));
      wq.n_layers = static_cast<u32>(sv::rl(static_cast<int>(win.n_layers), q2 * GS));
      wq.out_off = static_cast<u32>(sv::rl(static_cast<int>(win.out_off), q2 * GS));
      wq.out_cap = static_cast<u32>(sv::rl(static_cast<int>(win.out_cap), q2 * GS));
      wi2 = static_cast<u32>(sv::rl(static_cast<int>(wi), q2 * GS));
      return wq;
    };
    for (int q2 = 0; q2 < kG; ++q2) {
      if (sv::rl(static_cast<int>(phase), q2 * GS) != static_cast<int>(kRunning)) continue;
      u32 wi2;
      const PoaWindow wq = window_of(q2, wi2);
      Poa2Slot g = poa2_carve(A.scratch + static_cast<size_t>(slot0 + q2) * A.slot_bytes, A.nmax, A.lmax, kBand, true);
      u32 nn2 = 0, ne2 = 0;
      const u32 r = poa3_init_window(A, wq, g, wi2, nn2, ne2);
      if (q == q2) {
        nn = nn2;
        n_eff = ne2;
        if (r != 1) {
          phase = kFinal;
          status = r;
        }
      }
    }
    // ---- layers: every window of the wave aligns its next layer in the same round ----
    for (;;) {
      bool act = false, full = false;
      u32 len = 0;
      i32 lb = 0, span = 0;
      const PoaLayer* Lp = A.layers;
      for (int q2 = 0; q2 < kG; ++q2) {
        if (sv::rl(static_cast<int>(phase), q2 * GS) != static_cast<int>(kRunning)) continue;
        u32 wi2;
        const PoaWindow wq = window_of(q2, wi2);
        u32 liq = static_cast<u32>(sv::rl(static_cast<int>(li), q2 * GS));
        const u32 nnq = static_cast<u32>(sv::rl(static_cast<int>(nn), q2 * GS));
        while (liq < wq.n_layers && (A.layers[wq.layer_first + liq].len == 0 ||
                                     (A.src.layer_ok && !A.src.layer_ok[wq.layer_first + liq])))
          ++liq;
        if (liq >= wq.n_layers) {
          if (q == q2) phase = kLayersDone;
          continue;
        }
        const PoaLayer L = A.layers[wq.layer_first + liq];
        if (L.len > A.lmax || L.len > static_cast<u32>(kPoa2MaxSeq)) {
          if (q == q2) {
            phase = kFailed;
            status = 4;
          }
          continue;
        }
        Poa2Slot g = poa2_carve(A.scratch + static_cast<size_t>(slot0 + q2) * A.slot_bytes, A.nmax, A.lmax, kBand, true);
        Poa3Group<K>& Sg = S.g[q2];
        for (u32 i = lane; i < L.len; i += 64) {
          Sg.seq_pad[4 + i] = static_cast<u8>(poa_layer_code(A.src, L, i));
          g.pos_node[i] = static_cast<u16>(kNone3);
        }
        if (lane == 0) Sg.seq_pad[3] = 0xFF;
        const u32 blen = A.layers[wq.layer_first].len;
        const u32 offset = static_cast<u32>(0.01 * blen);
        const bool fullq = L.begin < offset && L.end > blen - offset;
        t0 = sv::clock();
        if (!fullq) poa_subgraph_marks(g, nnq, A.nmax, L.begin, L.end);
        t_sub += sv::clock() - t0;
        if (q == q2) {
          act = true;
          full = fullq;
          len = L.len;
          lb = static_cast<i32>(L.begin);
          span = static_cast<i32>(L.end) - static_cast<i32>(L.begin) + 1;
          Lp = A.layers + wq.layer_first + liq;
          li = liq;
        }
      }
      if (!sv::any(act)) break;
      sv::sync();
      t0 = sv::clock();
      u32 best_row = 0;
      poa3_dp<K>(A, S, my_slot, act, nn, full, Lp, len, lb, span, best_row);
      sv::sync();  // backpointers visible to the traceback
      t_dp += sv::clock() - t0;
      t0 = sv::clock();
      const bool had = act;
      if (act && best_row == 0) {  // the last column is in no end node's band
        phase = kFailed;
        status = kPoaBandHit | (li << 8);
        act = false;
      }
      u32 bad = 0, band_hit = 0;
      poa3_traceback<K>(A, S, my_slot, act, nn, full, len, best_row, bad, band_hit);
      if (act && bad) {
        phase = kFailed;
        status = bad | (li << 8);
        act = false;
      } else if (act && band_hit) {
        phase = kFailed;
        status = kPoaBandHit | (li << 8);
        act = false;
      }
      sv::sync();  // pos_node
      t_tb += sv::clock() - t0;
      for (int q2 = 0; q2 < kG; ++q2) {
        if (!sv::rl(act ? 1 : 0, q2 * GS)) continue;
        u32 wi2;
        const PoaWindow wq = window_of(q2, wi2);
        const u32 liq = static_cast<u32>(sv::rl(static_cast<int>(li), q2 * GS));
        u32 nnq = static_cast<u32>(sv::rl(static_cast<int>(nn), q2 * GS));
        const PoaLayer L = A.layers[wq.layer_first + liq];
        Poa2Slot g = poa2_carve(A.scratch + static_cast<size_t>(slot0 + q2) * A.slot_bytes, A.nmax, A.lmax, kBand, true);
        const u32 why = poa3_add_alignment<K>(A, g, S.g[q2], L, nnq, t_add, t_ord);
        if (q == q2) {
          if (why) {
            phase = kFailed;
            status = why;
          } else {
            nn = nnq;
          }
        }
      }
      if (had) ++li;
    }
    // ---- results ----
    t0 = sv::clock();
    for (int q2 = 0; q2 < kG; ++q2) {
      const u32 ph = static_cast<u32>(sv::rl(static_cast<int>(phase), q2 * GS));
      if (ph == kIdle) continue;
      u32 wi2;
      PoaWindow wq = window_of(q2, wi2);
      u32 st = static_cast<u32>(sv::rl(static_cast<int>(status), q2 * GS));
      if (ph == kFailed) {
        poa3_copy_backbone(A, wq, A.layers[wq.layer_first], A.out + wq.out_off, A.out_len + wi2);
      } else if (ph == kLayersDone) {
        Poa2Slot g = poa2_carve(A.scratch + static_cast<size_t>(slot0 + q2) * A.slot_bytes, A.nmax, A.lmax, kBand, true);
        const u32 nnq = static_cast<u32>(sv::rl(static_cast<int>(nn), q2 * GS));
        wq.n_layers = static_cast<u32>(sv::rl(static_cast<int>(n_eff), q2 * GS));
        sv::sync();
        poa3_consensus<K>(g, nnq, A.nmax, wq, A.trim, S, A.out + wq.out_off, A.out_len + wi2);
        sv::sync();
        st = 1;
      }
      if (lane == 0) A.status[wi2] = st;
    }
    t_cons += sv::clock() - t0;
    sv::sync();
  }
  if (A.phase_cycles) {
    if (lane == 0) {
      sv::atomic_add(&A.phase_cycles[0], t_sub);
      sv::atomic_add(&A.phase_cycles[1], t_dp);
      sv::atomic_add(&A.phase_cycles[2], t_tb);
      sv::atomic_add(&A.phase_cycles[3], t_add);
      sv::atomic_add(&A.phase_cycles[4], t_ord);
      sv::atomic_add(&A.phase_cycles[5], t_cons);
    }
  }
}

template <int G, int C, int OCC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void poa3_kernel(const Poa3Args A, u32 n_waves) {
  using K = P3Cfg<G, C>;
  __shared__ Poa3Lds<K> lds;
  if (blockIdx.x >= n_waves) return;
  poa3_wave<K>(A, lds, blockIdx.x * G);
}

template <class K>
struct EmuCall {
  const Poa3Args* A;
  Poa3Lds<K>* S;
};
template <class K>
void emu_entry(void* p) {
  EmuCall<K>* c = static_cast<EmuCall<K>*>(p);
  poa3_wave<K>(*c->A, *c->S, 0);
}

Poa3Args args_of(const PoaBatchDev& b, unsigned char* scratch, size_t slot_bytes) {
  Poa3Args A{};
  A.windows = b.wins;
  A.n_windows = b.n_windows;
  A.layers = b.layers;
  A.src = b.src;
  A.scratch = scratch;
  A.slot_bytes = slot_bytes;
  A.nmax = b.nmax;
  A.lmax = b.lmax;
  A.m = b.m;
  A.n_ = b.n;
  A.gp = b.g;
  A.trim = b.trim;
  A.out = b.out;
  A.out_len = b.out_len;
  A.status = b.status;
  A.phase_cycles = b.phase_cycles;
  A.sched = b.sched;
  A.next = b.next;
  return A;
}

template <int G, int C, int OCC>
void launch_variant(Engine& e, const PoaBatchDev& b) {
  using K = P3Cfg<G, C>;
  const size_t slot_bytes = poa2_slot_bytes(b.nmax, b.lmax, K::kBand, true);
  size_t free_b = 0, total_b = 0;
  RVN_HIP(hipMemGetInfo(&free_b, &total_b));
  // resident waves per CU: what the LDS footprint and the register budget the kernel is built for (OCC per SIMD) allow
  u32 per_cu = std::min<u32>(static_cast<u32>((160u * 1024u) / sizeof(Poa3Lds<K>)), 4u * OCC);
  if (const char* ev = std::getenv("RVN_POA_WAVES_PER_CU")) per_cu = static_cast<u32>(std::atoi(ev));  // occupancy experiments
  per_cu = per_cu < 1 ? 1 : per_cu;
  u32 n_waves = std::min<u32>((b.n_windows + G - 1) / G, 256 * per_cu);
  const size_t budget = e.poa2_scratch.cap + free_b / 2;
  if (static_cast<size_t>(n_waves) * G * slot_bytes > budget)
    n_waves = static_cast<u32>(std::max<size_t>(1, budget / (slot_bytes * G)));
  unsigned char* d_scratch = e.poa2_scratch.get<unsigned char>(static_cast<size_t>(n_waves) * G * slot_bytes + 256);
  RVN_HIP(hipMemsetAsync(b.next, 0, 4, e.stream));
  const Poa3Args A = args_of(b, d_scratch, slot_bytes);
  RVN_KLAUNCH(kKPoaBanded, (poa3_kernel<G, C, OCC><<<n_waves, 64, 0, e.stream>>>(A, n_waves)));
}

template <class K>
void emulate_variant(const std::vector<PoaWindow>& wins, const std::vector<PoaLayer>& lays, const PoaSrc& src, u32 max_bb,
                     u32 max_len, int m, int n, int g, int trim, u8* out, u32* out_len, u32* status) {
  PoaBatchDev b{};
  b.lmax = std::min<u32>(kPoaMaxSeq, std::max<u32>(64, ((max_len + 63) / 64) * 64));
  b.nmax = std::min<u32>(8192, std::max<u32>(512, max_bb * 6));
  const size_t slot_bytes = poa2_slot_bytes(b.nmax, b.lmax, K::kBand, true);
  std::vector<unsigned char> scratch(slot_bytes * K::G + 256, 0);
  unsigned long long phase[10] = {};
  u32 next = 0;
  b.wins = wins.data();
  b.n_windows = static_cast<u32>(wins.size());
  b.layers = lays.data();
  b.src = src;
  b.m = m;
  b.n = n;
  b.g = g;
  b.trim = trim;
  b.out = out;
  b.out_len = out_len;
  b.status = status;
  b.phase_cycles = phase;
  b.sched = nullptr;
  b.next = &next;
  const Poa3Args A = args_of(b, scratch.data(), slot_bytes);
  std::vector<Poa3Lds<K>> lds(1);
  std::memset(static_cast<void*>(lds.data()), 0, sizeof(Poa3Lds<K>));
  EmuCall<K> call{&A, lds.data()};
  g_emu_lost_rows = 0;
  simt_emu::run_wave(&emu_entry<K>, &call);
  if (std::getenv("RVN_POA3_DEBUG"))
    std::fprintf(stderr, "[raven_hip] poa3 emulation: %llu predecessor rows read from the HBM copy\n", g_emu_lost_rows);
}

}  // namespace

// variant: 0 = four windows per wave, 64-column band; 1 = four windows, 32 columns; 2 = two windows, 32 columns;
// 3 = two windows, 64 columns
int poa_v3_band(int variant) { return (variant == 1 || variant == 2) ? 32 : 64; }

void poa_v3_launch(Engine& e, const PoaBatchDev& b, int variant) {
  if (b.n_windows == 0) return;
  switch (variant) {
    case 1: launch_variant<4, 2, 4>(e, b); break;
    case 2: launch_variant<2, 1, 4>(e, b); break;
    case 3: launch_variant<2, 2, 4>(e, b); break;
    default: launch_variant<4, 4, 3>(e, b); break;
  }
}

// The same kernel source on the host, one emulated wave (simt_emu): windows / layers / sources are host arrays.  TEST
// INFRASTRUCTURE (rvn_poa_banded_emulate); first attempt only — a window that needs a wider band comes back flagged.
void poa_v3_emulate(const std::vector<PoaWindow>& wins, const std::vector<PoaLayer>& lays, const PoaSrc& src, u32 max_bb,
                    u32 max_len, int m, int n, int g, int trim, u8* out, u32* out_len, u32* status, int variant) {
  if (wins.empty()) return;
  switch (variant) {
    case 1: emulate_variant<P3Cfg<4, 2>>(wins, lays, src, max_bb, max_len, m, n, g, trim, out, out_len, status); break;
    case 2: emulate_variant<P3Cfg<2, 1>>(wins, lays, src, max_bb, max_len, m, n, g, trim, out, out_len, status); break;
    case 3: emulate_variant<P3Cfg<2, 2>>(wins, lays, src, max_bb, max_len, m, n, g, trim, out, out_len, status); break;
    default: emulate_variant<P3Cfg<4, 4>>(wins, lays, src, max_bb, max_len, m, n, g, trim, out, out_len, status); break;
  }
}

}  // namespace rvn
