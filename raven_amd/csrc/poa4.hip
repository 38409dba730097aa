// poa4.hip — banded POA window kernel with the graph ROWS ON THE LANES: the first attempt of rvn_poa_consensus_batch / of
// a polishing round's consensus stage (racon Window::GenerateConsensus over spoa, as driven by raven::Polish,
// RavenLib/src/polish.cc:43-51; scores RavenLib/include/raven/graph/polish.hpp:13-17).  Same algorithm, graph layout, tie
// rules and escalation as poa2.hip (a window whose alignment touches the 32-column band goes on to poa2's 64 / 128 /
// 256 columns and the full-matrix kernel); what changes is how the NW of a layer sits on the wave.
//
// poa2 advances ONE graph row per loop iteration (64 lanes = 64 columns): every iteration is a dependent chain
// (ring write -> ring read -> candidates -> 6-step prefix maximum -> ring write) behind ~90 scalar instructions of row
// decode, and the kernel waits more than it issues.  Here a window owns 16 lanes, a lane owns a whole graph row and
// walks it left to right, two columns per step, so
//   * the horizontal gap chain is two max instructions inside the lane (no cross-lane prefix),
//   * a predecessor row is read from the window's LDS ring where another lane left it >= 1 step earlier (one aligned
//     32-bit read per in-edge and step: the static schedule below guarantees the cells exist), and a row's descriptor
//     (band start, match mask against the layer, LDS addresses of up to 8 predecessor rows) is one 32-byte record built
//     by a per-layer pre-pass, so the step loop decodes nothing,
//   * in-edges are folded with v_max on (score << 8 | diagonal << 3 | 7 - in-edge) keys: spoa's tie rule (diagonal
//     before vertical, first in-edge first) is the maximum's and the backpointer falls out of its low bits: FOUR bits per
//     cell.  0 = horizontal AND "vertical through the eighth in-edge" (tag 7 - 7): for the rows that have eight in-edges
//     (2.5 % of C4-like windows have one) the NW leaves one bit per column beside the row saying which of the two a code 0
//     is (Poa4Slot::v7; round 5 sent the window to poa2 when a walk met the case: 836 windows and a second kernel per round),
//   * backpointers leave as ONE coalesced 8-byte store per lane and 8 steps into a time-major stream (step, lane); the
//     traceback maps (row, column) -> (step, lane) through the descriptor.
// Schedule (all band starts even and non-decreasing along the topological order): row rho of the layer's rank range
// runs on lane rho % 16 during steps S - 1 .. S + 15, S = rho + rho / 16 + (b_rho - b_0) / 2 + 1 (step S - 1 only reads:
// it loads the cell left of the first column), computing columns b + 2k, b + 2k + 1 at step S + k.  A predecessor pi
// < rho with band start b_pi <= b_rho has written column j by step S_pi + (j - b_pi) / 2 <= S_rho + (j - b_rho) / 2 - 1:
// one step before it is read.  Sixteen lanes are always busy except for the (b - b_0) / 2 drift (~20 % of the steps).
//
// How a batch runs (round 5): ONE launch of persistent waves.  A wave takes a group of four windows (scheduling order =
// by decreasing layer count) from an atomic counter and carries it from the backbone graphs to the consensus: per layer the
// graph side of each window in turn (all 64 lanes on one window: graph update with the previous alignment, the next
// layer's set-up, its row descriptors), then NW and traceback of the four side by side.  Round 4 ran every phase as a
// kernel of its own over a chunk of windows in lock step (five launches per layer round on four streams): the rounds'
// barriers left ~45 % of the wave slots idle and every phase met its window's data cold (poa4_persistent below).
//
// Integer VALU + LDS bound; no MFMA.  Written against sv:: (simt.h): the same source runs under the host wavefront
// emulator (tests/test_poa4_emulation.py -> rvn_poa_banded_emulate: variant 5 = the persistent kernel, variant 4 = the
// same phase functions stepped wave by wave in lock step).
#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "poa.h"
#include "poa2_window.h"

namespace rvn {

namespace {

struct P4 {
  static constexpr int G = 4, GS = 16, kBand = 32;
  static constexpr int kRing = 22;    // score rows a window keeps in LDS = longest in-edge (in ranks) + 1 (a longer one sends the
                                      // window to poa2: none in 9 220 layers of 300 C4-like windows, the longest was 17.  Rounds 4-5
                                      // kept 24; round 6 gave two rows' worth of LDS to the lanes' row descriptors (Poa4Lds::qa ..)
  static constexpr int kRowB = 64;    // bytes per ring row: the 32 cells of the row's band, addressed by the COLUMN (below)
  static constexpr int kMaxD = 8;     // largest band-start difference along an in-edge (a larger one sends the window to poa2)
  static constexpr int kEdges = 8;    // in-edges a row descriptor holds (4-bit backpointers: 8 diagonal + 7 vertical codes + 0 = horizontal
                                      // or vertical through in-edge 7, see above)
  static constexpr int kEdgesMax = 15;  // in-edges of a row this kernel follows: a row of 9..15 keeps in-edges 0..6 in its descriptor and
                                        // in-edges 7..14 in an overflow record (Poa4Slot::ovf) the step's rare path reads; a row with more
                                        // sends the window to poa2 (the graph itself keeps kPoaMaxIn = 16)
  static constexpr int kU = 8;        // steps between two service points (descriptor prefetch, backpointer store)
};
constexpr u32 kNone4 = 0xFFFFu;
constexpr i32 kNegU = -0x30000000;   // the horizontal chain's value left of a band (a key: score << 8)
constexpr u32 kInactiveS = 0x7FFFu;  // "no row": the 2 S field of a descriptor (odd — a real row's is even — and later than any step)

// LDS of the NW.  A ring row holds the 32 cells of its row's band as 16 words, and the word of column pair p = j / 2 is
// p mod 16 WHATEVER the row's band start (a band is 16 pairs wide: every pair of it has a word of its own).  The score rings
// of the two windows of a HALF-WAVE (lanes 0-31: q = 0, 1; lanes 32-63: q = 2, 3) are interleaved word by word — word w of
// window q's ring at pair base + (2 w + (q & 1)) * 4.  A ds_read_b32 serves a half-wave per LDS cycle over 32 banks; the bank
// of an access is 2 (p mod 16) + (q & 1), whichever ring slot it goes to: at one step the 16 lanes of a window are at 16
// consecutive column pairs (lane to lane: one row on, one pair back), so in-edge reads of any slot, the -inf words of the
// in-edges a row does not have (same addressing) and the own row's store fall into 16 different banks of the window's
// parity.  (Round 4 / early round 5 addressed a row from its band start, pads of -inf either side: the lane-to-lane bank
// step was 10 or 8 with the drift of the band along the rows, and 48 % of the kernel's LDS cycles were bank conflicts.)
// What the pads were for — a predecessor's band ending left of the column asked for, or beginning at it — now reads
// another cell of the predecessor's row instead of -inf.  That is harmless by construction: such a candidate can only
// RAISE a cell, every cell the traceback walks is checked to lie inside its row's band (a move through such a candidate
// lands outside and sends the window to the 64-column kernel, as any band miss), and a walk that never leaves the bands has
// met only true values — so its score is both an upper and a lower bound of the banded optimum and its moves are the true
// maxima's.  The ring starts every layer as -inf so that what a first column finds left of a predecessor's
// band is low, not stale.  slot = rho % kRing.
constexpr int kRowW4 = P4::kRowB / 4;
struct alignas(16) Poa4Lds {
  // The row descriptor a lane is working on, and the one it works on next (round 6): two slots per lane, [slot][lane] so that
  // the 64 lanes' reads fall into different banks whichever slot each lane is at.  The NW step reads its row's words from
  // here (one ds_read_b128 + one ds_read_b32, issued a step AHEAD) instead of keeping current / next / in-flight copies in
  // registers: finishing a row is a pointer flip.  Rounds 4-5 copied seven registers under a lane mask at every row switch,
  // and one of the wave's 64 lanes switches in 98 % of the steps — a third of the step's instructions.
  uint4 qa[2][64];  // {2 S | own row's ring bytes << 16, in-edges 0 | 1 << 16, in-edges 2 | 3 << 16, match mask}
  u32 qm[2][64];    // bits 3..6: (8 x (band start / 2 mod 16) - 8 S) mod 128 (the step's column pair as ring bytes = (this + 8 t) & 0x78);
                    // bit 8 (alone in its byte: one SDWA compare): an end node's row; bits 24..27: in-edges; bit 31: the row has no in-edge, or more than four (the step's rare path)
  uint2 qe[2][64];  // {in-edges 4 | 5 << 16, in-edges 6 | 7 << 16}
  u32 qc[2][64];    // node | band start << 16 | in-edges << 26 | marked << 30 | end node << 31 (rare paths only)
  u32 ring[2][2 * P4::kRing * kRowW4];
  u32 dump[2][32];      // where rows outside the layer's subgraph leave their cells (interleaved like the rings)
  u32 neg[40];          // -inf cells: what a descriptor's unused in-edges point at (window parity p: words p, p + 2, ..)
};
// byte offsets from the start of Poa4Lds (what a row descriptor holds)
__host__ __device__ __forceinline__ u32 poa4_ring_byte(int q, u32 w) {
  return static_cast<u32>(offsetof(Poa4Lds, ring)) + static_cast<u32>(q >> 1) * static_cast<u32>(sizeof(u32) * 2 * P4::kRing * kRowW4) +
         (2u * w + static_cast<u32>(q & 1)) * 4u;
}
__host__ __device__ __forceinline__ u32 poa4_dump_byte(int q) {
  return static_cast<u32>(offsetof(Poa4Lds, dump)) + static_cast<u32>(q >> 1) * 128u + static_cast<u32>(q & 1) * 4u;
}
__host__ __device__ __forceinline__ u32 poa4_neg_byte(int q) { return static_cast<u32>(offsetof(Poa4Lds, neg)) + static_cast<u32>(q & 1) * 4u; }
static_assert(offsetof(Poa4Lds, qa) == 0, "a lane's slot pointer is a byte offset into qa");
static_assert(offsetof(Poa4Lds, ring) % 128 == 0 && offsetof(Poa4Lds, dump) % 128 == 0 && (sizeof(u32) * 2 * P4::kRing * kRowW4) % 128 == 0,
              "a row's ring bytes leave bits 3..6 to the column pair");
// (the phases of the kernel share the wave's LDS as a union: the NW the score rings, the graph update room for 896 order
// slots, the set-up + descriptor pass the layer's bytes, the traceback its staged rows)
static_assert(sizeof(Poa4Lds) <= 10240, "sixteen waves per CU need <= 10 KB of LDS each");
static_assert(P4::kRowB == 2 * P4::kBand, "ring row = the 32 cells of a band");

struct Poa4Args {
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
  u32* esc;      // queue of the windows the 32-column attempt hands on, taken up by the waves of the SAME launch between two groups
                 // (poa4_esc_*): [0] entries pushed, [1] entries taken, [2] windows that went on to 128 columns, [4 + k] window
                 // index (0xFFFFFFFF until stored); null: none
  u32 esc_cap;   // entries the queue holds (a window beyond them keeps its status for the host's escalation, as without a queue)
  u32 esc_wide;  // != 0: the wave's four slots also hold the 128-column function's one
};

// Per-window scratch: poa2's graph arrays + the row descriptors of the current layer + its backpointer stream.
struct Poa4Slot {
  Poa2Slot g;
  uint4* desc;   // 2 per row: {2 S | own << 16, node | b << 16 | np << 26 | marked << 30 | end << 31, rank distances of in-edges 0..5
                 //            (5 bits each), e0 | e1 << 16}, {e2 | e3 << 16, e4 | e5 << 16, e6 | e7 << 16, match mask}: what the
                 //            traceback needs of a row is its first 16 bytes
  u32* rb;       // per node: rank | backbone coordinate << 16 (kept by the set-up and by poa4_update_graph)
  u32* rbl;      // per node, for the CURRENT layer: rank | band start << 16 (first pass of the descriptor phase)
  u32* v7;       // per row with eight in-edges: bit c = column band start + c took the VERTICAL move through the eighth in-edge —
                 // the one move the 4-bit codes cannot tell from "horizontal" (both code 0); written by the NW for such rows only.
                 // Per row with 9..15 in-edges: bit c = column band start + c took its move through one of the in-edges 7..14
                 // (the code's low three bits then count from in-edge 7: 14 - in-edge)
  uint4* ovf;    // per row with 9..15 in-edges: the ring bytes of in-edges 7..14 (8 x 16 bits; -inf cells for those it does not have)
  uint2* bps;    // backpointer stream: [step / 8][lane of the window] 8 bytes = 8 steps x 2 columns x 4 bits
  u32* seq2g;    // the current layer: [0, 60) 2 bits per base, [64, 96) its band guide as eight segments (set-up kernel ->
                 // descriptor / graph update kernels)
};
__host__ __device__ inline u32 poa4_desc_rows(u32 nmax) { return nmax + 64; }
__host__ __device__ inline u32 poa4_steps(u32 nmax, u32 lmax) { return nmax + nmax / 16 + lmax / 2 + 96; }
__host__ __device__ inline size_t poa4_v7_bytes(u32 nmax) { return (static_cast<size_t>(poa4_desc_rows(nmax)) * 4 + 255) & ~size_t(255); }
inline size_t poa4_slot_bytes(u32 nmax, u32 lmax) {
  size_t b = poa2_slot_bytes(nmax, lmax, 0, false);
  b += static_cast<size_t>(poa4_desc_rows(nmax)) * 32;
  b += 2 * ((static_cast<size_t>(nmax) * 4 + 255) & ~size_t(255));
  b += (static_cast<size_t>(poa4_steps(nmax, lmax)) / P4::kU + 2) * 16 * 8;
  b += poa4_v7_bytes(nmax);
  b += static_cast<size_t>(poa4_desc_rows(nmax)) * 16;
  b += 512;
  return (b + 255) & ~size_t(255);
}
__host__ __device__ inline Poa4Slot poa4_carve(unsigned char* base, u32 nmax, u32 lmax) {
  Poa4Slot s;
  s.g = poa2_carve(base, nmax, lmax, 0, false);
  size_t o = 0;
  poa2_fields(nmax, lmax, 0, [&](int, size_t x) { o += (x + 255) & ~size_t(255); }, false);
  s.desc = reinterpret_cast<uint4*>(base + o);
  o += static_cast<size_t>(poa4_desc_rows(nmax)) * 32;
  s.rb = reinterpret_cast<u32*>(base + o);
  o += (static_cast<size_t>(nmax) * 4 + 255) & ~size_t(255);
  s.rbl = reinterpret_cast<u32*>(base + o);
  o += (static_cast<size_t>(nmax) * 4 + 255) & ~size_t(255);
  s.bps = reinterpret_cast<uint2*>(base + o);
  o += (static_cast<size_t>(poa4_steps(nmax, lmax)) / P4::kU + 2) * 16 * 8;
  s.v7 = reinterpret_cast<u32*>(base + o);
  o += poa4_v7_bytes(nmax);
  s.ovf = reinterpret_cast<uint4*>(base + o);  // (right behind v7: the NW's rare path derives it from that pointer)
  o += static_cast<size_t>(poa4_desc_rows(nmax)) * 16;
  s.seq2g = reinterpret_cast<u32*>(base + o);
  return s;
}

// ---- small instruction-level helpers (one CDNA4 instruction each; plain C++ under the emulator) ----------------------
__host__ __device__ __forceinline__ u32 funnel_shr(u32 hi, u32 lo, u32 sh) {  // v_alignbit_b32; sh in [0, 31]
  return static_cast<u32>(((static_cast<unsigned long long>(hi) << 32) | lo) >> sh);
}
__host__ __device__ __forceinline__ i32 sext16(u32 x) { return static_cast<i32>(static_cast<i16>(x & 0xFFFFu)); }
__host__ __device__ __forceinline__ i32 imax(i32 a, i32 b) { return a > b ? a : b; }
__host__ __device__ __forceinline__ i32 imin(i32 a, i32 b) { return a < b ? a : b; }
// a + (p & 0xFFFF) / a + (p >> 16): v_add_u32 with an SDWA source select, so two 16-bit fields share a register
template <bool HI>
__host__ __device__ __forceinline__ u32 add_half(u32 a, u32 p) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 d;
  if constexpr (HI)
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(d) : "v"(a), "v"(p));
  else
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(d) : "v"(a), "v"(p));
  return d;
#else
  return a + (HI ? p >> 16 : p & 0xFFFFu);
#endif
}
// int16 half of w * 256 + tag: v_mad_i32_i16 reads the half through op_sel, the cells stay packed as the LDS read delivers
// them.  A key is score << 8 | tag (round 6; rounds 4-5: << 4): the score is bytes 1..2 of the key, so the two cells of a step
// go back into one packed word with ONE v_perm_b32 (keys_to_pair) instead of two shifts and a pack.
template <bool HI, int TAG>
__host__ __device__ __forceinline__ i32 cell_key(u32 w) {
#if defined(__HIP_DEVICE_COMPILE__)
  i32 d;
  if constexpr (HI) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(w), "s"(256), "n"(TAG));
  else asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(w), "s"(256), "n"(TAG));
  return d;
#else
  return static_cast<i32>(static_cast<i16>((HI ? w >> 16 : w) & 0xFFFFu)) * 256 + TAG;
#endif
}
// (k0 >> 8) & 0xFFFF | (k1 >> 8) << 16
__host__ __device__ __forceinline__ u32 keys_to_pair(i32 k0, i32 k1) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(static_cast<u32>(k1), static_cast<u32>(k0), 0x06050201u);
#else
  return ((static_cast<u32>(k0) >> 8) & 0xFFFFu) | ((static_cast<u32>(k1) >> 8) << 16);
#endif
}
__host__ __device__ __forceinline__ i32 imax3(i32 a, i32 b, i32 c) {
  const i32 m = a > b ? a : b;
  return m > c ? m : c;
}
// byte B of acc replaced by the low byte of v (v_perm_b32; the other bytes of v are ignored)
template <int B>
__host__ __device__ __forceinline__ u32 put_byte(u32 acc, u32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr u32 sel = B == 0 ? 0x0C0C0C04u : (B == 1 ? 0x0C0C0400u : (B == 2 ? 0x0C040100u : 0x04020100u));
  return __builtin_amdgcn_perm(v, acc, sel);
#else
  const u32 keep = B == 0 ? 0u : (B == 1 ? 0xFFu : (B == 2 ? 0xFFFFu : 0xFFFFFFu));
  return (acc & keep) | ((v & 0xFFu) << (8 * B));
#endif
}
__host__ __device__ __forceinline__ u32 pack16(i32 lo, i32 hi) {  // (lo & 0xFFFF) | hi << 16
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(static_cast<u32>(hi), static_cast<u32>(lo), 0x05040100u);
#else
  return (static_cast<u32>(lo) & 0xFFFFu) | (static_cast<u32>(hi) << 16);
#endif
}
__host__ __device__ __forceinline__ u32 clamp_pair(u32 w) {  // both int16 halves clamped from below to kNegInf16
#if defined(__HIP_DEVICE_COMPILE__)
  typedef short pk16 __attribute__((ext_vector_type(2)));
  const pk16 lim = {static_cast<short>(kNegInf16), static_cast<short>(kNegInf16)};
  return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(pk16, w), lim));
#else
  const i32 a = sext16(w), b = sext16(w >> 16);
  return pack16(a < kNegInf16 ? kNegInf16 : a, b < kNegInf16 ? kNegInf16 : b);
#endif
}
// LDS of the wave as bytes from the start of its Poa4Lds (on the GPU the struct is the kernel's only __shared__ object)
__host__ __device__ __forceinline__ u32 lds_ld32(const Poa4Lds& S, u32 off) {
  return *reinterpret_cast<const u32*>(reinterpret_cast<const unsigned char*>(&S) + off);
}
__host__ __device__ __forceinline__ void lds_st32(Poa4Lds& S, u32 off, u32 v) {
  *reinterpret_cast<u32*>(reinterpret_cast<unsigned char*>(&S) + off) = v;
}
__host__ __device__ __forceinline__ i32 lds_ld16(const Poa4Lds& S, u32 off) {
  return *reinterpret_cast<const i16*>(reinterpret_cast<const unsigned char*>(&S) + off);
}
// LDS traffic of one wave is executed in order on the GPU; the emulator's fibres need a rendezvous between a lane's LDS
// write and another lane's read of it
__host__ __device__ __forceinline__ void lds_order() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_wave_barrier();
#else
  sv::sync();
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
#define P4_MARK(x) asm volatile("; P4MARK " x)
#define P4_ASSUME_GLOBAL(p) \
  __builtin_assume(!__builtin_amdgcn_is_shared((const void*)(p))); \
  __builtin_assume(!__builtin_amdgcn_is_private((const void*)(p)))
#define P4_ASSUME_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
#else
#define P4_MARK(x)
#define P4_ASSUME_GLOBAL(p)
#define P4_ASSUME_LDS(p)
#endif

// The lane's slot pointer, made opaque to the optimiser: field addresses derived from it are computed where they are used
// instead of being hoisted out of a loop and kept (or spilled: a reload is a memory load the loop then waits for) there.
__host__ __device__ __forceinline__ unsigned char* opaque(unsigned char* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(p));
#endif
  return p;
}

__host__ __device__ inline u32 group_min_u(u32 v) {
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    const u32 o = static_cast<u32>(sv::bperm(static_cast<int>(v), sv::lane() ^ off));
    v = o < v ? o : v;
  }
  return v;
}
__host__ __device__ inline i32 group_max_i(i32 v) {
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) {
    const i32 o = sv::bperm(v, sv::lane() ^ off);
    v = o > v ? o : v;
  }
  return v;
}

// k-th in-edge (among those inside the subgraph) of v as a rank
__host__ __device__ inline u32 poa4_nth_pred_rank(const Poa4Slot& sl, u32 v, u32 k, bool full) {
  const Poa2Slot& g = sl.g;
  const u32 c = g.in_cnt[v];
  u32 seen = 0;
  for (u32 i = 0; i < c; ++i) {
    const u32 t = g.in_tail[v * kPoaMaxIn + i];
    if (full || g.mark[t]) {
      if (seen == k) return sl.rb[t] & 0xFFFFu;
      ++seen;
    }
  }
  return 0;
}

// ---- the row descriptors of a layer: flat over the nodes ---------------------------------------------------------------
// One wave = 256 nodes of one window (a launch gives a window ceil(nmax / 256) waves; those beyond its graph return at
// once).  Everything a node contributes is a coalesced load (rb[] holds rank and backbone coordinate of a node in one
// word), only its in-edges' tails are gathered; band starts come from the layer's guide through per-segment reciprocals
// kept in LDS.  The descriptor of every node whose rank lies in the rank range [r_lo, r_hi) of the layer's subgraph is
// written at its row rho = rank - r_lo; the steps of the layer's NW, the "beyond this kernel's limits" flag and the work
// counter are folded into the window's record with atomics.
__host__ __device__ __forceinline__ u32 mulhi_u32(u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return static_cast<u32>((static_cast<unsigned long long>(a) * b) >> 32);
#endif
}
// n / d for n * d < 2^32 through m = magic_of(d): floor(2^32 / d) + 1, 0 standing for d == 1
__host__ __device__ __forceinline__ u32 magic_of(u32 d) { return d <= 1 ? 0u : static_cast<u32>(0x100000000ULL / d) + 1u; }
__host__ __device__ __forceinline__ u32 div_magic(u32 n, u32 m) { return m ? mulhi_u32(n, m) : n; }

struct alignas(16) Poa4LdsDesc {
  u32 segtab[32];  // the layer's band guide: {x0, wa, wb - wa, magic(x1 - x0)} per segment
  u32 seq2[60];    // the layer, 2 bits per base
  u8 bytes[1024];  // (set-up kernel: one-byte codes before they are packed; then 8192 rank-indexed bits of the subgraph sweep)
};

// the layer's band guide as eight segments in LDS (lanes 0..7), and the band start of a backbone coordinate from it
__host__ __device__ inline void poa4_guide_to_lds(Poa4LdsDesc& S, const PoaLayer* Lp, u32 len, i32 span) {
  const int lane = sv::lane();
  if (lane < 8) {
    const i32 sg = lane;
    const i32 x0 = (sg * span) / 8, x1 = ((sg + 1) * span) / 8;
    const i32 wa = sg == 0 ? 0 : static_cast<i32>(Lp->way[sg - 1]);
    const i32 wb = sg == 7 ? static_cast<i32>(len) : static_cast<i32>(Lp->way[sg]);
    S.segtab[4 * sg] = static_cast<u32>(x0);
    S.segtab[4 * sg + 1] = static_cast<u32>(wa);
    S.segtab[4 * sg + 2] = static_cast<u32>(wb - wa);
    S.segtab[4 * sg + 3] = magic_of(static_cast<u32>(x1 > x0 ? x1 - x0 : 1));
  }
}
template <class K>
__host__ __device__ __forceinline__ i32 poa4_band_start(const Poa4LdsDesc& S, i32 bpos, i32 lb, i32 span, u32 span_magic, u32 len) {
  // even, in [0, max(0, w - 31)]; = poa_layer_center(bpos - lb) - 16 clamped
  i32 x = bpos - lb;
  x = x < 0 ? 0 : (x > span ? span : x);
  const u32 seg = div_magic(static_cast<u32>(x) * 8u, span_magic);
  const u32 sg = seg > 7u ? 7u : seg;
  const uint4 st = *reinterpret_cast<const uint4*>(&S.segtab[4 * sg]);
  const i32 dw = static_cast<i32>(st.z);
  const u32 num = static_cast<u32>(x - static_cast<i32>(st.x)) * static_cast<u32>(dw < 0 ? -dw : dw);
  const i32 qn = static_cast<i32>(div_magic(num, st.w));
  i32 b = static_cast<i32>(st.y) + (dw < 0 ? -qn : qn) - K::kBand / 2;
  const i32 bmax = static_cast<i32>(len) + 1 - K::kBand;
  b = b > bmax ? bmax : b;
  b = b < 0 ? 0 : b;
  // even; at the right limit rounded UP, so that the band still holds the layer's last column
  return (b == bmax && bmax > 0) ? (b + 1) & ~1 : b & ~1;
}

// ---- banded NW of one layer per window, rows on lanes ---------------------------------------------------------------
// Outputs (group-uniform): best_rho1 = 1 + row (rho) of the end node with the best score in the last column, 0: the last
// column is in no end node's band.
//
// The step (round 6).  Counters of round 5 (SQ_* count quad-cycles: 33 G VALU instructions against ~44-51 G SIMD quad-cycles of
// the launch) say the kernel keeps the vector ALUs busy 65-75 % of the time at four waves per SIMD: a wave's step does not
// wait for LDS, it waits for the other waves' instructions — so the step is priced in instructions, and round 5's ~88
// vector instructions for 2 x 64 cells are now ~50:
//   * the lane's row descriptor is read from LDS a step ahead (Poa4Lds::qa / qm); the end of a row flips a pointer (rounds 4-5:
//     ~28 instructions under a lane mask in 98 % of the steps: seven register copies, the end-node test, the schedule test);
//   * keys are score << 8 | tag: both cells return to a packed pair of int16 with one v_perm_b32;
//   * the horizontal chain stays in the key domain (tag 0 = "horizontal"): a cell is max3(diagonal, vertical, horizontal) and
//     its backpointer code the key's low nibble — no score / code selects;
//   * the step's column pair comes from the step counter and a per-row constant ((qm + 8 t) & 0x78), the row's own step number
//     (k) is only compared; the match bits of the step are two sign-extending bit-field extracts;
//   * the end-node score is looked at by the service point after the row's last pair has been stored, the schedule is
//     checked where descriptors are fetched (consecutive rows of a lane start >= 17 steps apart): neither in the step.
// The scores of the NW: racon's defaults as raven::Polish passes them (m = 3, n = -5, g = -4: RavenLib/include/raven/graph/
// polish.hpp:13-17) are literals of a specialised instance — the persistent kernel keeps ~100 scalar registers live across the
// NW, and round 5's step fetched its score constants out of spilled scalar registers (v_readlane) every time; any other
// scores take the instance that reads them from the batch.
struct Poa4ScoresAny {
  i32 m, n, g;
  __host__ __device__ explicit Poa4ScoresAny(const Poa4Args& A) : m(A.m), n(A.n_), g(A.gp) {}
};
struct Poa4ScoresRacon {
  static constexpr i32 m = 3, n = -5, g = -4;
  __host__ __device__ explicit Poa4ScoresRacon(const Poa4Args&) {}
};
template <class K, class SC>
__host__ __device__ inline void poa4_dp(const Poa4Args A, Poa4Lds& S, unsigned char* slot_mem, bool act, u32 t_end, u32 len,
                                        u32& best_rho1) {
  P4_ASSUME_GLOBAL(slot_mem);
  P4_ASSUME_LDS(&S);
  const int lane = sv::lane();
  const int gl = lane & 15, q = lane >> 4;
  const Poa4Slot sl = poa4_carve(slot_mem, A.nmax, A.lmax);
  // the rings start as -inf (the union is reused by the layer set-up and by the traceback), the wave's -inf cells
  {
    const u32 neg = pack16(kNegInf16, kNegInf16);
    for (int idx = gl; idx < K::kRing * kRowW4; idx += 16) lds_st32(S, poa4_ring_byte(q, static_cast<u32>(idx)), neg);
    if (lane < 40) S.neg[lane] = neg;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(RVN_DEBUG_KNOBS)
    // (host emulator only, tests/test_poa4_emulation.py: the rings start as scores of cells far off the diagonal (1), as huge ones (2), as scores around those of a
    // band's edge in a window's first rows (3) instead
    // of -inf — what a read beside a predecessor's band then finds is adversarial, and every window that still comes back
    // polished must equal the oracle: exactness rests on the traceback's band checks, not on what the ring holds)
    if (const char* f = knob("RVN_POA4_RING_FILL")) {
      const int mode = std::atoi(f);
      for (int idx = gl; idx < K::kRing * kRowW4; idx += 16) {
        u32 h = (static_cast<u32>(idx) * 2654435761u) ^ (static_cast<u32>(q) * 40503u) ^ (len * 97u);
        h ^= h >> 13;
        const i32 base = mode == 3 ? 10 : -100, span = mode == 3 ? 50 : 800;
        const i32 lo = mode == 2 ? 28000 : base - static_cast<i32>(h % span), hi = mode == 2 ? 28000 : base - static_cast<i32>((h >> 12) % span);
        if (mode >= 1 && mode <= 3) lds_st32(S, poa4_ring_byte(q, static_cast<u32>(idx)), pack16(lo, hi));
      }
    }
#endif
  }
  const SC sc(A);
  const i32 xD = sc.n * 256 + 8, dD = (sc.m - sc.n) * 256, gK = sc.g * 256;  // keys = score * 256 + tag; the diagonal's tag bit rides on the score term
  const i32 gp = sc.g;
  const u32 neg_off = poa4_neg_byte(q);
  const u32 neg2 = neg_off | (neg_off << 16);
  const u32 negpair = pack16(kNegInf16, kNegInf16);
  // a descriptor as the desc pass left it (2 x 16 bytes) into one of the lane's two slots; `on` = false: a row that never starts
  // (l16 = 16 x lane: the lane's place in a slot.  The loop derives it from its slot pointer where it parks: as three lane-derived
  // LDS addresses of their own (qa, qm / qc, qe) the compiler kept two in registers across the NW and one in scratch memory, reloaded
  // — behind a wait for everything outstanding — at every service point)
  auto park = [&](u32 l16, u32 par, uint4 da, uint4 db, bool on) __attribute__((always_inline)) {
    const u32 s2 = on ? (da.x & 0xFFFFu) : kInactiveS;
    const u32 c1 = on ? da.y : 0u;
    const u32 np = (c1 >> 26) & 15u;
    // bytes of the band's first column pair in a ring row, minus 8 S: the pair of step t is (this + 8 t) & 0x78
    const u32 r0 = (((c1 >> 14) & 0x78u) - 4u * s2) & 0x78u;
    const bool rare = on && s2 != kInactiveS && (np == 0u || np > 4u);
    unsigned char* const base = reinterpret_cast<unsigned char*>(&S);
    const u32 a16 = l16 + par * (64u * 16u);
    *reinterpret_cast<uint4*>(base + offsetof(Poa4Lds, qa) + a16) =
        uint4{on ? da.x : (kInactiveS | (neg_off << 16)), on ? da.w : neg2, on ? db.x : neg2, db.w};
    *reinterpret_cast<u32*>(base + offsetof(Poa4Lds, qm) + (a16 >> 2)) = r0 | ((c1 >> 31) << 8) | (np << 24) | (rare ? 0x80000000u : 0u);
    *reinterpret_cast<uint2*>(base + offsetof(Poa4Lds, qe) + (a16 >> 1)) = uint2{on ? db.y : neg2, on ? db.z : neg2};
    *reinterpret_cast<u32*>(base + offsetof(Poa4Lds, qc) + (a16 >> 2)) = c1;
  };
  bool sched_bad = false;
  u32 ld_rho = static_cast<u32>(gl) + 16u;  // the last row whose descriptor went to LDS
  u32 ld_s2;                                // its 2 S
  {
    const uint4 a = sl.desc[2 * static_cast<size_t>(gl)], b = sl.desc[2 * static_cast<size_t>(gl) + 1];
    const uint4 a2 = sl.desc[2 * static_cast<size_t>(gl + 16)], b2 = sl.desc[2 * static_cast<size_t>(gl + 16) + 1];
    park(static_cast<u32>(lane) * 16u, 0, a, b, act);  // (a group without a layer in this round has no descriptors: it idles on -inf cells)
    park(static_cast<u32>(lane) * 16u, 1, a2, b2, act);
    const u32 s0 = act ? (a.x & 0xFFFFu) : kInactiveS;
    ld_s2 = act ? (a2.x & 0xFFFFu) : kInactiveS;
    // consecutive rows of a lane start at least 17 steps apart (band starts do not decrease along the order)
    if (act && s0 != kInactiveS && ld_s2 < s0 + 34u) sched_bad = true;
  }
  lds_order();
  // the lane's slot as bytes into qa (qm = bytes / 4, qe = bytes / 2); the words of the step to come
  u32 ptr = static_cast<u32>(offsetof(Poa4Lds, qa)) + static_cast<u32>(lane) * 16u;
  constexpr u32 kFlip = 64u * 16u;
  uint4 cw = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(&S) + ptr);
  u32 cm = lds_ld32(S, static_cast<u32>(offsetof(Poa4Lds, qm)) + (ptr >> 2));
  uint4 lA = uint4{0, 0, 0, 0}, lB = uint4{0, 0, 0, 0};
  bool ld_pending = false;
  i32 Am1 = 0, UK = kNegU;
  u32 v7m = 0;
  i32 best_score = -0x7FFFFFFF;
  u32 best_row = 0;
  const u32 T = static_cast<u32>(sv::wave_max(act ? static_cast<int>(t_end) : 0));
  // An end node's score in the layer's last column, looked at by the first service point after the row's last pair was
  // stored: the row's descriptor is still in the lane's other slot (the next fetch parks 8 steps after it was issued, and it
  // is issued here at the earliest), and its cells are still in the ring — the row kRing rows on, which takes the ring slot
  // over, stores its first pair >= kRing + 1 steps after this row stored its first, i.e. >= 8 steps after this row's last:
  // not before the step this service point precedes.
  static_assert(K::kRing >= 16 + K::kU - 2, "an end node's cells outlive the row by a service interval");
  auto end_check = [&](u32 t0) __attribute__((always_inline)) {
    const u32 optr = ptr ^ kFlip;
    const u32 px = lds_ld32(S, optr), pm = lds_ld32(S, static_cast<u32>(offsetof(Poa4Lds, qm)) + (optr >> 2));
    const i32 fin2 = static_cast<i32>(2u * t0) - static_cast<i32>(px & 0xFFFFu) - 30;  // 2 x steps since the row's last pair
    const bool hit = (pm & 0x100u) != 0 && fin2 >= 2 && fin2 <= 2 * K::kU;
    if (sv::any(hit)) {
      const u32 c1 = lds_ld32(S, static_cast<u32>(offsetof(Poa4Lds, qc)) + (optr >> 2));
      const i32 idx = static_cast<i32>(len) - static_cast<i32>((c1 >> 16) & 0x3FFu);
      if (hit && idx >= 0 && idx < K::kBand) {
        const u32 cb3 = (c1 >> 14) & 0x78u;
        const i32 sce = lds_ld16(S, (px >> 16) + ((cb3 + 8u * (static_cast<u32>(idx) >> 1)) & 0x78u) + 2u * (static_cast<u32>(idx) & 1u));
        const u32 rho = ((cw.x & 0xFFFFu) == ld_s2 ? ld_rho : ld_rho - 16u) - 16u;  // (the lane is at the last row parked, or at the one before)
        const u32 cand = ((c1 & 0xFFFFu) << 16) | (rho + 1u);  // node id | 1 + row: equal scores -> smallest node id
        if (sce > best_score || (sce == best_score && cand < best_row)) {
          best_score = sce;
          best_row = cand;
        }
      }
    }
  };
  u32 t0 = 0, accp0 = 0, accp1 = 0;
  uint2* bp = sl.bps + gl;  // where the backpointers of the eight steps before go: [step / 8][lane of the window]
#if defined(RVN_DEBUG_KNOBS)
  unsigned long long t_sp = 0;
#endif
  for (; t0 < T; t0 += K::kU) {
#if defined(RVN_DEBUG_KNOBS)
    const unsigned long long sp0 = sv::clock();
#endif
    end_check(t0);
    // ---- service point: the descriptor fetched 8 steps ago goes into the slot of the row the lane finished last; a lane
    // that has moved on to the last row it holds fetches the one after (needed 17 steps after that switch at the earliest:
    // fetched <= 8 steps after it, parked <= 16 steps after it).  The backpointers of the PREVIOUS eight steps leave here,
    // after the two: the only wait for vector memory in the loop that finds anything outstanding is the one in front of the
    // parking, and what it finds — the fetch and the store of the service point before — was issued eight steps ago.  (Rounds 4-5
    // stored at the end of the eight steps: the wait at the top of the next eight then sat behind a store issued a moment
    // ago, every time, for the length of a write to HBM.) ----
#if defined(__HIP_DEVICE_COMPILE__)
    // (the loads of the service point before are waited for HERE, by every lane: left to the branch below, the compiler has
    // to wait again where the next loads overwrite their registers — behind the store issued a moment before)
    // (an empty asm statement is no use of a register for the compiler's wait counting; an instruction is)
    {
      u32 seen = (lA.x | lA.y | lA.z) | (lA.w | lB.x | lB.y) | (lB.z | lB.w);
      asm volatile("" : "+v"(seen));
    }
#endif
    if (ld_pending) {
      const u32 rho = ld_rho + 16u;
      u32 l16 = ptr & (kFlip - 1u);
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(l16));  // (from the slot pointer, here: not a value kept since before the loop)
#endif
      park(l16, (rho >> 4) & 1u, lA, lB, true);
      const u32 s2 = lA.x & 0xFFFFu;
      if (ld_s2 != kInactiveS && s2 < ld_s2 + 34u) sched_bad = true;  // the row's first step would be over: band starts decreased along the order
      ld_s2 = s2;
      ld_rho = rho;
    }
    {
      // (every lane loads at every service point — those with nothing to fetch the wave's first descriptor, one request for all of
      // them: a load under a lane mask makes its registers a merge of two definitions around the loop, and the compiler then
      // shuffles — and waits for — the loaded registers right behind the load)
      const bool want = !ld_pending && act && ld_s2 != kInactiveS && (cw.x & 0xFFFFu) == ld_s2;  // (a lane's rows have increasing S: the row it is at IS the last one parked)
      const uint4* src = sl.desc + (want ? 2 * static_cast<size_t>(ld_rho + 16) : 0);
#if defined(P4_EXP) && P4_EXP == 6  // (timing experiment: no descriptor fetch)
      (void)src;
      ld_pending = false;
#else
      lA = src[0];
      lB = src[1];
      ld_pending = want;
#endif
    }
    // (the store BEHIND the loads: the compiler re-waits for "everything" in front of the loads whatever was waited for above
    // — free when nothing is outstanding, the length of a write to HBM behind a store)
#if defined(P4_EXP) && (P4_EXP == 3 || P4_EXP == 6)  // (timing experiment: no backpointer store)
    if (act && t0 != 0 && accp0 == 0x12345678) *bp = uint2{accp0, accp1};
#else
    if (act && t0 != 0) *bp = uint2{accp0, accp1};
#endif
#if defined(RVN_DEBUG_KNOBS)
    t_sp += sv::clock() - sp0;
#endif
    u32 acc0 = 0, acc1 = 0;
#pragma unroll
    for (int u = 0; u < K::kU; ++u) {
      P4_MARK("step_begin");
      const u32 t = t0 + static_cast<u32>(u);
      const i32 k2 = static_cast<i32>(2u * t) - static_cast<i32>(cw.x & 0xFFFFu);  // 2 k: -2 = the pair left of the band, 0 .. 30 the band
      // the step's column pair as a byte offset into ANY ring row (interleaved rings: a pair is 8 bytes on)
      const u32 off4 = (cm + 8u * t) & 0x78u;
      // ---- in-edges: one aligned pair of predecessor cells each (columns j, j + 1 of this step).  A row's unused in-edges
      // point at -inf cells: in-edges 1..3 are read whether the row has them or not, so the four reads are in flight together
      // (issuing them under the exec mask of the lanes that have the in-edge was measured in round 5: slower) ----
#if defined(P4_EXP) && P4_EXP == 1  // (timing experiment: no ring reads)
      u32 wd = add_half<false>(off4, cw.y);
      const u32 w1 = add_half<true>(off4, cw.y), w2 = add_half<false>(off4, cw.z), w3 = add_half<true>(off4, cw.z);
#else
      u32 wd = lds_ld32(S, add_half<false>(off4, cw.y));
      const u32 w1 = lds_ld32(S, add_half<true>(off4, cw.y));
      const u32 w2 = lds_ld32(S, add_half<false>(off4, cw.z));
      const u32 w3 = lds_ld32(S, add_half<true>(off4, cw.z));
#endif
      // ---- the words of the next step: the same row's, or — after the row's last pair — the other slot's ----
      const u32 nptr = k2 == 30 ? ptr ^ kFlip : ptr;
#if defined(P4_EXP) && P4_EXP == 2  // (timing experiment: no descriptor reads)
      const uint4 nw = uint4{cw.x + (nptr & 0x10000u), cw.y, cw.z, cw.w};
      const u32 nm = cm;
#else
      const uint4 nw = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(&S) + nptr);
      const u32 nm = lds_ld32(S, static_cast<u32>(offsetof(Poa4Lds, qm)) + (nptr >> 2));
#endif
      i32 A0, A1;
      bool wave_has8 = false;  // (wave-uniform: some lane's row has eight in-edges)
      if (sv::any(static_cast<i32>(cm) < 0)) {  // rows without an in-edge inside the subgraph, rows with more than four (1 % of the rows)
        P4_MARK("rare_begin");
        const uint2 ce = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(&S) + offsetof(Poa4Lds, qe) + (ptr >> 1));
        const u32 np = (cm >> 24) & 15u;
        if (sv::any(np == 0)) {  // the virtual start row H[0][j] = j * g
          const u32 c1 = lds_ld32(S, static_cast<u32>(offsetof(Poa4Lds, qc)) + (ptr >> 2));
          const i32 j0 = static_cast<i32>((c1 >> 16) & 0x3FFu) + k2;
          u32 vp = pack16(j0 * gp, (j0 + 1) * gp);
          if (j0 < 0) vp = negpair;
          wd = np == 0 ? vp : wd;
        }
        A0 = cell_key<false, 7>(wd);
        A1 = cell_key<true, 7>(wd);
#define P4_EDGE(E, REG, HI)                                        \
  {                                                                \
    const u32 we = lds_ld32(S, add_half<HI>(off4, REG));           \
    A0 = imax(A0, cell_key<false, 7 - E>(we));                     \
    A1 = imax(A1, cell_key<true, 7 - E>(we));                      \
  }
        if (sv::any(np > 4)) {
          P4_EDGE(4, ce.x, false)
          if (sv::any(np > 5)) {
            P4_EDGE(5, ce.x, true)
            if (sv::any(np > 6)) {
              P4_EDGE(6, ce.y, false)
              P4_EDGE(7, ce.y, true)
              wave_has8 = sv::any(np > 7);
            }
          }
        }
#undef P4_EDGE
        P4_MARK("rare_end");
      } else {
        A0 = cell_key<false, 7>(wd);
        A1 = cell_key<true, 7>(wd);
      }
      A0 = imax3(A0, cell_key<false, 6>(w1), cell_key<false, 5>(w2));
      A1 = imax3(A1, cell_key<true, 6>(w1), cell_key<true, 5>(w2));
      A0 = imax(A0, cell_key<false, 4>(w3));
      A1 = imax(A1, cell_key<true, 4>(w3));
      // ---- the two cells: spoa's priority diagonal (first in-edge reaching the maximum), vertical, horizontal — the order of
      // the tags (8 | 7 - e, 7 - e, 0); equal keys are "vertical through the eighth in-edge" and "horizontal": code 0 for both ----
      const u32 cs = cw.w >> (static_cast<u32>(k2) & 31u);
      const i32 m0 = static_cast<i32>(cs << 31) >> 31, m1 = static_cast<i32>(cs << 30) >> 31;  // -1: the row's base matches the column's
      i32 M0 = imax3(Am1 + xD + (m0 & dD), A0 + gK, UK + gK);
      const i32 U0K = M0 & ~0xFF;
      i32 M1 = imax3(A0 + xD + (m1 & dD), A1 + gK, U0K + gK);
      const bool in_band = k2 >= 0;
#if defined(P4_EXP) && P4_EXP == 4  // (timing experiment: no ring store)
      if (in_band && M0 == 0x12345678) lds_st32(S, add_half<true>(off4, cw.x), clamp_pair(keys_to_pair(M0, M1)));
#else
      if (in_band) lds_st32(S, add_half<true>(off4, cw.x), clamp_pair(keys_to_pair(M0, M1)));
#endif
      UK = in_band ? (M1 & ~0xFF) : kNegU;
      if (wave_has8) {
        // A row of eight in-edges (a tenth of the 2.5 % of C4-like windows that have one): "vertical through the eighth" has
        // tag 0, the code of "horizontal".  Which of the two a code 0 of such a row means goes into a 32-bit mask beside the
        // row, one bit per column (round 5 handed the window to the 64-column kernel when a walk met the case: 836 windows and
        // 59 ms of a second kernel per C4 round).  The vertical candidate wins a tie with the horizontal one (max3's key
        // order), and a diagonal of the same score would have left a tag >= 8.
        P4_MARK("rare_begin");
        const u32 npc = (cm >> 24) & 15u;
        const bool has8 = npc > 7u;
        u32 b0 = (in_band && (M0 & 0xFF) == 0 && M0 == A0 + gK) ? 1u : 0u;
        u32 b1 = (in_band && (M1 & 0xFF) == 0 && M1 == A1 + gK) ? 1u : 0u;
        if (sv::any(npc > 8u)) {
          // A row of 9..15 in-edges (241 of the 244 windows a C4 round of round 6 still handed to the 64-column kernel had one:
          // a launch of 38 ms behind the persistent kernel for a thousandth of the windows).  In-edges 7..14 are a second group
          // with tags of their own (7 - (e - 7)): their cells of this step's pair and of the pair before come out of the ring here
          // (addresses: the row's overflow record, a global load — on this path only), the group's two cells are folded like the
          // first group's, and a cell takes the second group's key where that one is better by spoa's order — higher score, then
          // diagonal before vertical before horizontal; the first group has the earlier in-edges and keeps a tie inside a class.
          // The row's mask (v7: such a row keeps no eighth in-edge in its descriptor, code 0 is "horizontal") says per column which
          // group the code counts in.  The cells were stored above with the first group's values: stored again here, same lane,
          // before any other lane reads them (a step later at the earliest).
          const bool has9 = npc > 8u;
          const u32 rho_c = (cw.x & 0xFFFFu) == ld_s2 ? ld_rho : ld_rho - 16u;
          // (a loop that is NOT unrolled, one in-edge per turn, its ring bytes fetched as the 16 bits they are: this path runs
          // for one row in ~1 000 windows and must not cost the step's common path a register — unrolled over the record's four
          // words it pushed the descriptor pointer of the service point into scratch memory: two reloads and two waits for
          // "everything outstanding" per eight steps, +35 % on the whole NW)
          const u16* const ovp = reinterpret_cast<const u16*>(reinterpret_cast<const unsigned char*>(sl.v7) + poa4_v7_bytes(A.nmax)) +
                                 8u * static_cast<size_t>(has9 ? rho_c : 0u);
          const u32 offp = (off4 - 8u) & 0x78u;
          i32 B0 = kNegU, B1 = kNegU, Bm1 = kNegU;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
          for (i32 e = 0; e < 8; ++e) {
            const u32 rb = has9 ? static_cast<u32>(ovp[e]) : neg_off;
            const u32 wc = lds_ld32(S, rb + off4), wp = lds_ld32(S, rb + offp);
            const i32 tag = 7 - e;
            B0 = imax(B0, sext16(wc) * 256 + tag);
            B1 = imax(B1, sext16(wc >> 16) * 256 + tag);
            Bm1 = imax(Bm1, sext16(wp >> 16) * 256 + tag);
          }
          // class of a key: 2 diagonal, 1 vertical, 0 horizontal (first group: tag 0 — its vertical tags are 7..1 in such a row)
          auto better = [](i32 kb, i32 ka) -> bool {
            const i32 ca = (ka & 8) ? 2 : ((ka & 7) ? 1 : 0), cb = (kb & 8) ? 2 : 1;
            return ((kb & ~0xFF) | cb) > ((ka & ~0xFF) | ca);
          };
          const i32 N0 = imax(Bm1 + xD + (m0 & dD), B0 + gK);
          const bool t0b = has9 && better(N0, M0);
          const i32 M0n = t0b ? N0 : M0;
          const i32 M1a = imax3(A0 + xD + (m1 & dD), A1 + gK, (M0n & ~0xFF) + gK);
          const i32 N1 = imax(B0 + xD + (m1 & dD), B1 + gK);
          const bool t1b = has9 && better(N1, M1a);
          const i32 M1n = t1b ? N1 : M1a;
          if (has9) {
            M0 = M0n;
            M1 = M1n;
            if (in_band) lds_st32(S, add_half<true>(off4, cw.x), clamp_pair(keys_to_pair(M0, M1)));
            UK = in_band ? (M1 & ~0xFF) : kNegU;
            b0 = (in_band && t0b) ? 1u : 0u;
            b1 = (in_band && t1b) ? 1u : 0u;
          }
        }
        v7m = (k2 <= 0 ? 0u : v7m) | ((b0 | (b1 << 1)) << (static_cast<u32>(k2) & 31u));
        if (has8 && k2 == 30) sl.v7[(cw.x & 0xFFFFu) == ld_s2 ? ld_rho : ld_rho - 16u] = v7m;
        P4_MARK("rare_end");
      }
      Am1 = A1;
      u32 cp = (static_cast<u32>(M0) & 15u) | (static_cast<u32>(M1) << 4);  // (byte 0: code of the pair; the rest is dropped by put_byte)
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(cp));  // computed here: sunk to the store it would keep the step's two keys alive for eight steps
#endif
      if (u == 0) acc0 = put_byte<0>(acc0, cp);
      else if (u == 1) acc0 = put_byte<1>(acc0, cp);
      else if (u == 2) acc0 = put_byte<2>(acc0, cp);
      else if (u == 3) acc0 = put_byte<3>(acc0, cp);
      else if (u == 4) acc1 = put_byte<0>(acc1, cp);
      else if (u == 5) acc1 = put_byte<1>(acc1, cp);
      else if (u == 6) acc1 = put_byte<2>(acc1, cp);
      else acc1 = put_byte<3>(acc1, cp);
      lds_order();
      ptr = nptr;
      cw = nw;
      cm = nm;
      P4_MARK("step_end");
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // (the registers the store above took its data from stay untouched until here: the compiler guards a register that an
    // outstanding store reads — data or address — against being overwritten by waiting for the store: right behind it, had it been
    // free to reuse them)
    asm volatile("" : : "v"(accp0), "v"(accp1), "v"(bp));
#endif
    accp0 = acc0;
    accp1 = acc1;
    if (t0 != 0) bp += 16;
  }
  if (act && t0 != 0) *bp = uint2{accp0, accp1};
  end_check(t0);  // rows that finished in the loop's last steps
#if defined(RVN_DEBUG_KNOBS)
  if (A.phase_cycles && lane == 0) sv::atomic_add(&A.phase_cycles[9], t_sp);
#endif
  if (A.phase_cycles && lane == 0) sv::atomic_add(&A.phase_cycles[14], static_cast<unsigned long long>((T + K::kU - 1) / K::kU * K::kU));
  // among the end nodes with the best score the one with the SMALLEST NODE ID (a rule that does not depend on the order of
  // the rows: spoa takes the first in ITS rank order, which is not the device's), as poa2 / poa pick it
  {
    const i32 gs = group_max_i(best_score);
    const u32 cand = (best_score == gs && best_row != 0) ? best_row : 0xFFFFFFFFu;
    const u32 br = group_min_u(cand);
    best_rho1 = (act && br != 0xFFFFFFFFu) ? (br & 0xFFFFu) : 0u;
    if (sv::any(sched_bad)) {
      const u32 any_bad = static_cast<u32>(sv::ballot(sched_bad) >> (lane & ~15)) & 0xFFFFu;
      if (any_bad) best_rho1 = 0;  // (reported as a band miss: the 64-column kernel takes the window)
    }
  }
}

// ---- traceback of the wave's windows, round-synchronous -------------------------------------------------------------
// A round = every window walks through kTbG blocks of 16 rows.  Lane l of a window fetches rows 16 * block + l of the
// round — three descriptor words and the 24 bytes of the backpointer stream that hold the row's 32 four-bit codes (its 16
// steps lie in at most three 8-step blocks of the stream) — a round AHEAD into registers, descriptors two rounds ahead (the
// codes' addresses come out of them), and parks them in the window's LDS when the round begins.  The walk itself is the
// same for all 16 lanes of a window (every lane reads the same two LDS addresses per step: the current row's descriptor,
// then the code under column j): no cross-lane traffic, no select trees.  All four windows change rounds at the same
// point of the loop, so the wait there is for loads issued a round ago, not for another window's prefetch of a moment
// ago (the memory counter is the wave's, not the window's).
constexpr int kTbG = 2;
struct alignas(16) Poa4LdsTb {
  // per row of the round: the 16 code bytes of its 32 columns ALIGNED TO THE COLUMN (byte (j >> 1) & 15 holds the codes of
  // columns j, j + 1: where the code of (row, j) lies does not depend on the row's band start, so the walk reads it beside
  // the row's record, not behind it), then {band start | edge flags | in-edges, node, rank distances of in-edges 0..5, -}
  uint4 row[P4::G][16 * kTbG][2];
};
// the 16 code bytes of a row (its steps 0 .. 15 start at byte S % 8 of the 24 the lane fetched) rotated so that the byte of
// columns (j, j + 1) sits at index (j >> 1) & 15; bt = the row's (even) band start
__host__ __device__ __forceinline__ uint4 poa4_align_codes(u32 c0, u32 c1, u32 c2, u32 c3, u32 c4, u32 c5, u32 S, u32 bt) {
  const u32 sb = S & 7u;
  const bool hi = sb >= 4u;
  const u32 w0 = hi ? c1 : c0, w1 = hi ? c2 : c1, w2 = hi ? c3 : c2, w3 = hi ? c4 : c3, w4 = hi ? c5 : c4;
  const u32 sh = 8u * (sb & 3u);
  const u32 b0 = funnel_shr(w1, w0, sh), b1 = funnel_shr(w2, w1, sh), b2 = funnel_shr(w3, w2, sh), b3 = funnel_shr(w4, w3, sh);
  // rotate the 128-bit ring left by r bytes: byte k goes to (k + r) & 15
  const u32 r = (bt >> 1) & 15u, rw = r >> 2, rs = 8u * (r & 3u);
  const u32 o0 = rw == 0 ? b0 : (rw == 1 ? b3 : (rw == 2 ? b2 : b1));
  const u32 o1 = rw == 0 ? b1 : (rw == 1 ? b0 : (rw == 2 ? b3 : b2));
  const u32 o2 = rw == 0 ? b2 : (rw == 1 ? b1 : (rw == 2 ? b0 : b3));
  const u32 o3 = rw == 0 ? b3 : (rw == 1 ? b2 : (rw == 2 ? b1 : b0));
  if (rs == 0) return uint4{o0, o1, o2, o3};
  const u32 back = 32u - rs;
  return uint4{funnel_shr(o0, o3, back), funnel_shr(o1, o0, back), funnel_shr(o2, o1, back), funnel_shr(o3, o2, back)};
}
template <class K>
__host__ __device__ inline void poa4_traceback(const Poa4Args A, Poa4LdsTb& S, unsigned char* slot_mem, bool act, u32 r_lo,
                                               u32 n_rows, bool full, u32 len, u32 best_rho1, u32& bad, u32& band_hit) {
  P4_ASSUME_GLOBAL(slot_mem);
  P4_ASSUME_LDS(&S);
  (void)n_rows;
  const int lane = sv::lane();
  const int gl = lane & 15, q = lane >> 4;
  const Poa4Slot sl = poa4_carve(slot_mem, A.nmax, A.lmax);
  const uint4* const dsc = sl.desc;
  // (the backpointer stream's address is formed from the descriptors' where the codes are fetched, once per round of 32 rows: as a
  // pointer of its own it lived — with the descriptors' — in scratch memory across the walk, reloaded at every change of round)
  const size_t bps_off = static_cast<size_t>(poa4_desc_rows(A.nmax)) * 32 + 2 * ((static_cast<size_t>(A.nmax) * 4 + 255) & ~size_t(255));  // (poa4_carve)
  // (likewise the position -> node table: a scalar distance from the descriptors, the address formed where a block of sixteen leaves)
  const i32 pn_off = sv::rfl(static_cast<i32>(reinterpret_cast<const unsigned char*>(sl.g.pos_node) - reinterpret_cast<const unsigned char*>(sl.desc)));
  auto pos_node_at = [&](u32 idx) __attribute__((always_inline)) -> u16* {
    i32 po = pn_off;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(po));
#endif
    return reinterpret_cast<u16*>(const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(dsc)) + static_cast<ptrdiff_t>(po)) + idx;
  };
  const u32 w = len + 1;
  bad = 0;
  band_hit = 0;
  u32 i = act ? best_rho1 : 0;  // 1 + rho of the current row; 0 = the virtual start row
  i32 j = static_cast<i32>(w) - 1;
  bool done = !act || i == 0;
  u32 steps = 0, n_switch = 0;
  unsigned long long t_change = 0;
  u32 pbuf = kNone4, pblk = 0xFFFFFFFFu;  // the node of position 16 pblk + lane-of-the-window, for the block the walk is in
  // the next round (descriptors and codes in flight), the round after (descriptors in flight)
  u32 nd0[kTbG] = {}, nd1[kTbG] = {}, nd7[kTbG] = {}, fd0[kTbG] = {}, fd1[kTbG] = {}, fd7[kTbG] = {};
  // (native vectors, not HIP's uint4 class: arrays of the latter stay in scratch memory when passed by reference)
  typedef u32 v2u __attribute__((ext_vector_type(2)));
  v2u na[kTbG] = {}, nb[kTbG] = {}, nc[kTbG] = {};
  u32 c_rnd = 0xFFFFFFFFu, n_rnd = 0xFFFFFFFFu, f_rnd = 0xFFFFFFFFu;  // rounds LDS / the two register sets hold
  auto load_desc = [&](u32 rnd, u32 (&d0)[kTbG], u32 (&d1)[kTbG], u32 (&d7)[kTbG]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < kTbG; ++h) {
      const size_t rho = (static_cast<size_t>(rnd) * kTbG + h) * 16 + static_cast<size_t>(gl);
      const uint4 da = dsc[2 * rho];
      d0[h] = da.x;
      d1[h] = da.y;
      d7[h] = da.z;
    }
  };
  auto load_codes = [&](const u32 (&d0)[kTbG], v2u (&a)[kTbG], v2u (&b)[kTbG], v2u (&c)[kTbG]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < kTbG; ++h) {
      const u32 s2 = d0[h] & 0xFFFFu;
      const size_t tb = s2 == kInactiveS ? 0u : (s2 >> 1) / K::kU;
      size_t bo = bps_off;
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+s"(bo));
#endif
      const v2u* src = reinterpret_cast<const v2u*>(reinterpret_cast<const unsigned char*>(dsc) + bo) + tb * 16 + static_cast<size_t>(gl);
      a[h] = src[0];
      b[h] = src[16];
      c[h] = src[32];
    }
  };
  while (sv::any(!done)) {
    // ---- change of round, all windows at once ----
    const u32 rnd = done ? c_rnd : (i - 1) / (16 * kTbG);
    const bool change = !done && rnd != c_rnd;
    // (the step and cycle counters of this function stay in the product build: without them — nothing else changed — the
    // kernel measured 18 % SLOWER on the same box, 700 against 592 ms per C4 round, at every setting of the block alignment;
    // the generated code differs by the two s_memtime reads and a handful of register moves.  Not understood)
    const unsigned long long tc0 = sv::clock();
    if (sv::any(change)) {
      lds_order();  // the walks of the previous round have read their last row
      if (change) {
        ++n_switch;
        if (rnd != n_rnd) {  // the first round of the walk (or a jump the sets do not cover): both levels right here
          load_desc(rnd, nd0, nd1, nd7);
          load_codes(nd0, na, nb, nc);
          f_rnd = 0xFFFFFFFFu;
        }
#pragma unroll
        for (int h = 0; h < kTbG; ++h) {
          uint4* dst = S.row[q][16 * h + gl];
          dst[0] = poa4_align_codes(na[h].x, na[h].y, nb[h].x, nb[h].y, nc[h].x, nc[h].y, (nd0[h] & 0xFFFFu) >> 1, (nd1[h] >> 16) & 0x3FFu);
          const u32 bt = (nd1[h] >> 16) & 0x3FFu;
          dst[1] = uint4{bt | (bt != 0 ? 0x10000u : 0u) | (bt + K::kBand < w ? 0x20000u : 0u) | (((nd1[h] >> 26) & 15u) << 20),
                         nd1[h] & 0xFFFFu, nd7[h], 0u};
        }
        c_rnd = rnd;
        if (rnd >= 1) {
          if (f_rnd == rnd - 1) {
#pragma unroll
            for (int h = 0; h < kTbG; ++h) {
              nd0[h] = fd0[h];
              nd1[h] = fd1[h];
              nd7[h] = fd7[h];
            }
          } else {
            load_desc(rnd - 1, nd0, nd1, nd7);
          }
          load_codes(nd0, na, nb, nc);
          n_rnd = rnd - 1;
        } else {
          n_rnd = 0xFFFFFFFFu;
        }
        if (rnd >= 2) {
          load_desc(rnd - 2, fd0, fd1, fd7);
          f_rnd = rnd - 2;
        } else {
          f_rnd = 0xFFFFFFFFu;
        }
      }
      lds_order();
    }
    t_change += sv::clock() - tc0;
    // ---- the walk through the round's rows: every lane of the window does the same ----
    // One step = the row's record and the code under column j out of LDS (two dependent reads), then straight-line
    // selects: the four windows of a wave are in different cases at every step, so any branch here is taken by somebody
    // and all of them would be executed anyway.  Every step lowers i + j, so the walk ends without a step counter.
    bool in_round = !done;
    while (sv::any(in_round)) {
      P4_MARK("tb_step_begin");
      const u32 l = (i - 1) & (16 * kTbG - 1);
      // the row's record and the code under column j: ONE round trip to LDS (rounds 4-5 read the rank distances behind a
      // test of the in-degree, a third dependent read); the code of column j is byte (j >> 1) & 15 of the row whatever its
      // band start
      const uint4 d = S.row[q][l][1];
      u32 byte = reinterpret_cast<const u8*>(S.row[q][l])[(static_cast<u32>(j) >> 1) & 15u];
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(byte) : "v"(d.x), "v"(d.y), "v"(d.z));  // (all of it is wanted here, not where a branch first asks for it)
#endif
      const u32 code = (byte >> (4u * (static_cast<u32>(j) & 1u))) & 15u;  // 0: horizontal; else diagonal << 3 | 7 - in-edge
      // d.x: band start | (band start != 0) << 16 | (the matrix goes on right of the band) << 17 | in-edges << 20; d.y: node
      const u32 bt = d.x & 0xFFFFu, np = (d.x >> 20) & 15u, node = d.y;
      const u32 idx = static_cast<u32>(j) - bt;
      const bool oob = idx >= static_cast<u32>(K::kBand);  // the path left the stored band
      bool isH = code == 0u;
      const u32 k = (~code) & 7u;
      u32 ni = np ? i - ((d.z >> ((5u * k) & 31u)) & 31u) : 0u;  // (k >= 6: corrected below; the shift count as the GPU's shifter takes it)
      if (sv::any(in_round && np >= 7)) {  // rows of seven or eight in-edges (1 % of the windows have one)
        // code 0 in a row of eight in-edges is "horizontal" or "vertical through the eighth": the NW left the answer in the
        // row's v7 mask; in-edges 6 and 7: their ranks come from the graph
        // A row of 9..15 in-edges: the mask says which columns took their move through the second group (in-edges 7..14, counted
        // from 7 by the code's low bits; the descriptor of such a row has no eighth in-edge, code 0 outside the mask is "horizontal")
        u32 kk = k;
        if (in_round && !oob && np >= 8u && (isH || np >= 9u)) {
          const u32 m7 = poa4_carve(opaque(slot_mem), A.nmax, A.lmax).v7[i - 1];
          if ((m7 >> idx) & 1u) {
            isH = false;  // (np = 8: k = (~0) & 7 = 7, the eighth in-edge; code >> 3 = 0: no column is left)
            if (np >= 9u) kk = k + 7u;
          }
        }
        if (in_round && !oob && !isH && kk >= 6)
          ni = poa4_nth_pred_rank(poa4_carve(opaque(slot_mem), A.nmax, A.lmax), node, kk, full) - r_lo + 1;
      }
      // (a diagonal or horizontal code in column 0 — never on a consistent stream — leaves the band in the next step)
      const bool ok = in_round && !oob;
      // ---- the node a diagonal move aligns position j - 1 to: sixteen positions are collected in the window's lanes (lane =
      // position mod 16) and leave as one 32-byte store when the walk enters another block of sixteen; a position the walk
      // passes horizontally stays kNone, as the set-up left every position (rounds 4-5: one 2-byte store per diagonal step) ----
      const bool diag = ok && (code & 8u) != 0 && j > 0;
      const u32 p = static_cast<u32>(j) - 1u;
      if (sv::any(diag && (p >> 4) != pblk)) {
        const bool fl = diag && (p >> 4) != pblk;
        if (fl && pblk != 0xFFFFFFFFu) *pos_node_at(16u * pblk + static_cast<u32>(gl)) = static_cast<u16>(pbuf);
        if (fl) {
          pbuf = kNone4;
          pblk = p >> 4;
        }
      }
      pbuf = (diag && (p & 15u) == static_cast<u32>(gl)) ? node : pbuf;
      // ---- how near the path comes to an edge of the band beyond which the matrix goes on: within two cells = a band hit ----
      if (sv::any(ok && (idx - 2u) > static_cast<u32>(K::kBand - 5))) {
        if (ok && ((idx < 2u && (d.x & 0x10000u)) || (idx > static_cast<u32>(K::kBand - 3) && (d.x & 0x20000u)))) band_hit = 1;
      }
      if (in_round && oob) band_hit = 1;
      steps += in_round ? 1u : 0u;
      j -= ok ? static_cast<i32>(isH ? 1u : (code >> 3)) : 0;
      const bool row_move = ok && !isH;
      i = row_move ? ni : i;
      done = done || (in_round && (!ok || (row_move && ni == 0)));  // on the virtual row only insertions remain
      in_round = !done && (i - 1) / (16 * kTbG) == c_rnd;
      P4_MARK("tb_step_end");
    }
  }
  if (pblk != 0xFFFFFFFFu) *pos_node_at(16u * pblk + static_cast<u32>(gl)) = static_cast<u16>(pbuf);  // the positions of the last block
  if (A.phase_cycles && gl == 0 && act) {
    sv::atomic_add(&A.phase_cycles[11], static_cast<unsigned long long>(steps));
    sv::atomic_add(&A.phase_cycles[12], static_cast<unsigned long long>(n_switch));
  }
  if (A.phase_cycles && lane == 0) sv::atomic_add(&A.phase_cycles[13], t_change);
}

// ---- wave-wide per-window steps (as in poa2.hip) -------------------------------------------------------------------
__host__ __device__ inline void poa4_copy_backbone(const Poa4Args& A, const PoaWindow& win, const PoaLayer& bb, u8* out,
                                                   u32* out_len) {
  const int lane = sv::lane();
  const u32 n = bb.len < win.out_cap ? bb.len : win.out_cap;
  for (u32 i = lane; i < n; i += 64) out[i] = static_cast<u8>(poa_layer_code(A.src, bb, i));
  if (lane == 0) *out_len = n;
}

// window set-up: 0 = backbone returned (< 3 sequences), 4 = beyond a length limit (backbone returned), 1 = graph built
__host__ __device__ inline u32 poa4_init_window(const Poa4Args& A, const PoaWindow& win, Poa2Slot& g, u32* rb, u32 wi,
                                                u32& n_nodes, u32& n_eff) {
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
    poa4_copy_backbone(A, win, bb, A.out + win.out_off, A.out_len + wi);
    return 0;
  }
  if (blen == 0 || blen > A.nmax || blen > A.lmax) {
    poa4_copy_backbone(A, win, bb, A.out + win.out_off, A.out_len + wi);
    return 4;
  }
  // backbone graph (spoa AddAlignment with an empty alignment)
  n_nodes = blen;
  for (u32 i = lane; i < blen; i += 64) {
    g.code[i] = static_cast<u8>(poa_layer_code(A.src, bb, i));
    g.al_cnt[i] = 0;
    g.visits[i] = blen >= 2 ? 1 : 0;
    g.order[i] = static_cast<u16>(i);
    rb[i] = i | (i << 16);  // (rank and backbone coordinate of a node live in rb[] alone here: poa2's rank_of[] / bpos[] are not kept)
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
  // nodes to come: poa4_update_graph counts on zero in-degree / out-degree / visits of an id it hands out
  for (u32 i = blen + lane; i < A.nmax; i += 64) {
    g.in_cnt[i] = 0;
    g.out_cnt[i] = 0;
    g.visits[i] = 0;
  }
  sv::sync();
  return 1;
}

template <class GT>
__host__ __device__ __forceinline__ u32 poa4_letter(const GT& Sg, u32 p) {
  return (Sg.seq2[(p + 1) >> 4] >> (2 * ((p + 1) & 15u))) & 3u;
}

// ---- graph update of the wave's windows side by side: spoa AddAlignment + the incremental order rebuild ---------------
// Sixteen lanes per window, four sequence positions per lane and iteration, every level of the gather chain
// (position -> aligned node -> its aligned group -> the node the position lands on) issued for all four positions before
// any is consumed.  A path visits distinct nodes of distinct columns, so target lookup, node creation, group updates and
// the edge (p - 1 -> p) are conflict-free between positions; new ids and order slots come from ballots / prefix counts
// in path order, identical to the serial order of creation.  in_cnt / out_cnt / visits of every node id are zero from
// the window's set-up on, so creating a node writes only what is not zero, and an edge's weight and its tail's
// out-degree are added with atomics (no read-modify-write round trip).  The new nodes' order slots stay in the group's
// LDS (the DP ring is free); ranks move by a binary search there, and the order array is rebuilt into the window's
// second order buffer (`flip` says which of the two is current).  Returns 0 or the failure code (group-uniform).
__host__ __device__ __forceinline__ Poa2Slot poa4_graph(unsigned char* slot_mem, u32 nmax, u32 lmax, bool flip) {
  Poa2Slot g = poa4_carve(slot_mem, nmax, lmax).g;
  if (flip) {
    u16* t = g.order;
    g.order = g.order2;
    g.order2 = t;
  }
  return g;
}
__host__ __device__ __forceinline__ void atomic_add_i32(i32* p, i32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  *p += v;
#endif
}
__host__ __device__ __forceinline__ void atomic_inc_u16(u16* base, u32 idx) {  // base 4-byte aligned
#if defined(__HIP_DEVICE_COMPILE__)
  __hip_atomic_fetch_add(reinterpret_cast<u32*>(base) + (idx >> 1), (idx & 1u) ? 0x10000u : 1u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
#else
  base[idx] += 1;
#endif
}

// GW = lanes per window: 64 (one window per wave, 256 positions per iteration — the update kernel) or 16.
template <int GW, int UP>
__host__ __device__ inline u32 poa4_update_graph(const Poa4Args A, u16* nslot, const u32* seq2, unsigned char* slot_mem, bool act,
                                                 const PoaLayer* Lp, u32 len, u32 lb, u32& nn, bool& flip,
                                                 unsigned long long& t_add, unsigned long long& t_ord) {
  P4_ASSUME_GLOBAL(slot_mem);
  P4_ASSUME_GLOBAL(Lp);
  P4_ASSUME_LDS(nslot);
  P4_ASSUME_LDS(seq2);
  const int lane = sv::lane();
  const int gl = lane & (GW - 1), gbase = lane & ~(GW - 1);
  constexpr unsigned long long kGMask = GW == 64 ? ~0ULL : ((1ULL << (GW & 63)) - 1ULL);
  auto shift1 = [&](int v, int fill) -> int {  // value of the lane below in the window; its first lane gets `fill`
    if constexpr (GW == 16) {
      return sv::row_shr<1>(v, fill);
    } else {
      const int t = sv::bperm(v, lane - 1);
      return gl == 0 ? fill : t;
    }
  };
  auto gmax = [&](i32 v) -> i32 {
    if constexpr (GW == 16) return group_max_i(v);
    else return sv::wave_max(v);
  };
  const Poa2Slot g = poa4_graph(slot_mem, A.nmax, A.lmax, flip);
  // rank and backbone coordinate of a node are ONE word (rb[v] = rank | coordinate << 16): one gather where poa2's separate
  // rank_of[] / bpos[] arrays took two, one scattered store per node in the order rebuild instead of two
  const u32* const rb32 = poa4_carve(slot_mem, A.nmax, A.lmax).rb;
  u16* const rb16 = reinterpret_cast<u16*>(poa4_carve(slot_mem, A.nmax, A.lmax).rb);  // [2 v] rank, [2 v + 1] backbone coordinate
  const PoaLayer L = *Lp;
  const u32 nmax = A.nmax, lmax = A.lmax;
  unsigned long long t0 = sv::clock();
  const u32 n_old = nn;
  const u32 max_len = static_cast<u32>(sv::wave_max(act ? static_cast<int>(len) : 0));
  auto gballot = [&](bool p) -> unsigned long long { return (sv::ballot(p) >> gbase) & kGMask; };
  auto letter_at = [&](u32 p) -> u32 { return (seq2[(p + 1) >> 4] >> (2 * ((p + 1) & 15u))) & 3u; };
  // ---- the first aligned position: the unaligned prefix goes before all members of its column ----
  u32 first_p = 0xFFFFFFFFu;
  for (u32 p0 = 0; p0 < max_len; p0 += UP * GW) {
    if (!sv::any(act && first_p == 0xFFFFFFFFu && p0 < len)) break;
    u32 pn[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const u32 p = p0 + static_cast<u32>(GW) * static_cast<u32>(u) + static_cast<u32>(gl);
      pn[u] = (act && p < len) ? g.pos_node[p] : kNone4;
    }
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const unsigned long long bal = gballot(pn[u] != kNone4);
      if (first_p == 0xFFFFFFFFu && bal) first_p = p0 + static_cast<u32>(GW) * static_cast<u32>(u) + static_cast<u32>(__builtin_ctzll(bal));
    }
  }
  u32 carry_slot = n_old, carry_b = lb;
  if (act && first_p != 0xFFFFFFFFu) {
    const u32 an = g.pos_node[first_p];
    const u32 rb_an = rb32[an];
    u32 r = rb_an & 0xFFFFu;
    const u32 ac = g.al_cnt[an];
    const uint2 al2 = *reinterpret_cast<const uint2*>(g.al + static_cast<size_t>(an) * 4);
    const u32 a0 = al2.x & 0xFFFFu, a1 = al2.x >> 16, a2 = al2.y & 0xFFFFu;
    const u32 r0 = rb32[ac > 0 ? a0 : an] & 0xFFFFu, r1 = rb32[ac > 1 ? a1 : an] & 0xFFFFu, r2 = rb32[ac > 2 ? a2 : an] & 0xFFFFu;
    r = r0 < r ? r0 : r;
    r = r1 < r ? r1 : r;
    r = r2 < r ? r2 : r;
    carry_slot = r;
    carry_b = rb_an >> 16;
  }
  u32 total_new = 0;
  u32 why = 0;  // != 0: the window has failed; its lanes keep step with the wave without touching the graph
  u32 prev_tgt = kNone4;   // node of position p0 - 1 (the last position of the previous iteration)
  i32 prev_w = 0;          // its weight
  for (u32 p0 = 0; p0 < max_len; p0 += UP * GW) {
    const bool go = act && why == 0;
    u32 p[UP], an[UP], letter[UP];
    bool valid[UP], has[UP];
    // level 1: the nodes the positions are aligned to
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      p[u] = p0 + static_cast<u32>(GW) * static_cast<u32>(u) + static_cast<u32>(gl);
      valid[u] = go && p[u] < len;
      an[u] = valid[u] ? g.pos_node[p[u]] : kNone4;
      letter[u] = valid[u] ? letter_at(p[u]) : 0u;
    }
    // level 2: those nodes
    u32 c_an[UP], ac[UP], rk_an[UP], bp_an[UP];
    uint2 al2[UP];
    i32 wgt[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      has[u] = valid[u] && an[u] != kNone4;
      const u32 a = has[u] ? an[u] : 0u;
      c_an[u] = g.code[a];
      ac[u] = g.al_cnt[a];
      const u32 rba = rb32[a];
      rk_an[u] = rba & 0xFFFFu;
      bp_an[u] = rba >> 16;
      al2[u] = *reinterpret_cast<const uint2*>(g.al + static_cast<size_t>(a) * 4);
      wgt[u] = valid[u] ? static_cast<i32>(static_cast<u8>(poa_layer_weight(A.src, L, p[u]))) : 0;
    }
    // level 3: their aligned groups (at most three other letters)
    u32 kt[UP][3], c_kt[UP][3], rk_kt[UP][3];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      if (!has[u]) ac[u] = 0;
      kt[u][0] = al2[u].x & 0xFFFFu;
      kt[u][1] = al2[u].x >> 16;
      kt[u][2] = al2[u].y & 0xFFFFu;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const u32 t = static_cast<u32>(k) < ac[u] ? kt[u][k] : 0u;
        c_kt[u][k] = g.code[t];
        rk_kt[u][k] = rb32[t] & 0xFFFFu;
      }
    }
    // where every position lands; order slot / backbone coordinate of the last aligned position at or before it
    u32 tgt[UP], id_new[UP];
    bool is_new[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      u32 t = kNone4, rmax = rk_an[u];
      if (has[u]) {
        if (c_an[u] == letter[u]) t = an[u];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (static_cast<u32>(k) < ac[u]) {
            rmax = rk_kt[u][k] > rmax ? rk_kt[u][k] : rmax;
            if (t == kNone4 && c_kt[u][k] == letter[u]) t = kt[u][k];
          }
        }
      }
      const u32 gslot = rmax + 1, gb = bp_an[u];
      const unsigned long long bal = gballot(has[u]);
      const unsigned long long below = bal & (gl == 63 ? ~0ULL : ((2ULL << gl) - 1ULL));
      const int src = below ? 63 - __builtin_clzll(below) : 0;
      const u32 s_sh = static_cast<u32>(sv::bperm(static_cast<int>(gslot), gbase | src));
      const u32 b_sh = static_cast<u32>(sv::bperm(static_cast<int>(gb), gbase | src));
      const u32 fslot = below ? s_sh : carry_slot;
      const u32 fb = below ? b_sh : carry_b;
      {
        const int top = bal ? 63 - __builtin_clzll(bal) : 0;
        const u32 cs = static_cast<u32>(sv::bperm(static_cast<int>(gslot), gbase | top));
        const u32 cb = static_cast<u32>(sv::bperm(static_cast<int>(gb), gbase | top));
        if (bal) {
          carry_slot = cs;
          carry_b = cb;
        }
      }
      is_new[u] = valid[u] && t == kNone4;
      const unsigned long long nb = gballot(is_new[u]);
      const u32 cnt = static_cast<u32>(__builtin_popcountll(nb));
      if (go && why == 0 && (n_old + total_new + cnt > nmax || total_new + cnt > lmax)) why = 2;
      id_new[u] = 0;
      if (is_new[u] && why == 0) {
        const u32 tn = total_new + static_cast<u32>(__builtin_popcountll(nb & ((1ULL << gl) - 1ULL)));
        const u32 id = n_old + tn;
        id_new[u] = id;
        t = id;
        nslot[tn] = static_cast<u16>(fslot);
        g.code[id] = static_cast<u8>(letter[u]);
        rb16[2 * static_cast<size_t>(id) + 1] = static_cast<u16>(fb);
        u32 c2 = 0;
        uint2 mine = uint2{0, 0};
        if (has[u]) {  // joins an's aligned group: every member lists every other one
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            if (static_cast<u32>(k) < ac[u]) {
              g.al[static_cast<size_t>(kt[u][k]) * 4 + ac[u]] = static_cast<u16>(id);
              g.al_cnt[kt[u][k]] = static_cast<u8>(ac[u] + 1);
              if (c2 == 0) mine.x = kt[u][k];
              else if (c2 == 1) mine.x |= kt[u][k] << 16;
              else mine.y = kt[u][k];
              ++c2;
            }
          }
          if (ac[u] < 4) {
            g.al[static_cast<size_t>(an[u]) * 4 + ac[u]] = static_cast<u16>(id);
            g.al_cnt[an[u]] = static_cast<u8>(ac[u] + 1);
          }
          if (c2 == 0) mine.x = an[u];
          else if (c2 == 1) mine.x |= an[u] << 16;
          else if (c2 == 2) mine.y = an[u];
          else mine.y |= an[u] << 16;
          ++c2;
        }
        *reinterpret_cast<uint2*>(g.al + static_cast<size_t>(id) * 4) = mine;
        g.al_cnt[id] = static_cast<u8>(c2);
      }
      if (why == 0) total_new += cnt;
      tgt[u] = t;
    }
    // level 4: the in-edge lists of the nodes the positions land on (a new node's is empty)
    u32 icnt[UP];
    uint4 ta[UP], tb[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const bool old = valid[u] && why == 0 && !is_new[u];
      const u32 t = old ? tgt[u] : 0u;
      icnt[u] = old ? g.in_cnt[t] : 0u;
      ta[u] = old ? *reinterpret_cast<const uint4*>(g.in_tail + static_cast<size_t>(t) * kPoaMaxIn) : uint4{0, 0, 0, 0};
      tb[u] = old ? *reinterpret_cast<const uint4*>(g.in_tail + static_cast<size_t>(t) * kPoaMaxIn + 8) : uint4{0, 0, 0, 0};
    }
    // the edges (p - 1 -> p)
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      // node and weight of position p - 1: the lane below; lane 0 takes the last lane of the previous quarter
      const u32 last_t = static_cast<u32>(sv::bperm(static_cast<int>(u == 0 ? prev_tgt : tgt[u > 0 ? u - 1 : 0]), gbase | (GW - 1)));
      const i32 last_w = sv::bperm(u == 0 ? prev_w : wgt[u > 0 ? u - 1 : 0], gbase | (GW - 1));
      const u32 tail = static_cast<u32>(shift1(static_cast<int>(tgt[u]), static_cast<int>(last_t)));
      const i32 tw = shift1(wgt[u], last_w);
      if (valid[u] && why == 0) {
        const u32 head = tgt[u];
        if (len >= 2) {
          if (is_new[u]) g.visits[head] = 1;
          else atomic_inc_u16(g.visits, head);
        }
        if (p[u] >= 1) {
          const i32 weight = tw + wgt[u];
          const u32 c = icnt[u];
          u32 found = 0xFFu;
#pragma unroll
          for (int i2 = 0; i2 < 16; ++i2) {
            const uint4& src4 = i2 < 8 ? ta[u] : tb[u];
            const u32 wd = (i2 & 7) < 2 ? src4.x : ((i2 & 7) < 4 ? src4.y : ((i2 & 7) < 6 ? src4.z : src4.w));
            const u32 tt = (wd >> (16 * (i2 & 1))) & 0xFFFFu;
            if (static_cast<u32>(i2) < c && tt == tail && found == 0xFFu) found = static_cast<u32>(i2);
          }
          if (found != 0xFFu) {
            atomic_add_i32(g.in_w + static_cast<size_t>(head) * kPoaMaxIn + found, weight);
          } else if (c >= static_cast<u32>(kPoaMaxIn)) {
            why = 3;
          } else {
            g.in_tail[static_cast<size_t>(head) * kPoaMaxIn + c] = static_cast<u16>(tail);
            g.in_w[static_cast<size_t>(head) * kPoaMaxIn + c] = weight;
            g.in_cnt[head] = static_cast<u8>(c + 1);
            atomic_inc_u16(g.out_cnt, tail);
          }
        }
      }
    }
    // a failure anywhere in the window stops the whole window
    {
      const u32 wmax = static_cast<u32>(gmax(static_cast<i32>(why)));
      why = wmax;
    }
    prev_tgt = static_cast<u32>(sv::bperm(static_cast<int>(tgt[UP - 1]), gbase | (GW - 1)));
    prev_w = sv::bperm(wgt[UP - 1], gbase | (GW - 1));
  }
  t_add += sv::clock() - t0;
  t0 = sv::clock();
  const u32 n_new = total_new;
  lds_order();  // nslot
  // ---- order rebuild: old rank r -> r + #(new slots <= r); t-th new node -> slot_t + t ----
  const bool doit = act && why == 0 && n_new != 0;
  if (sv::any(doit)) {
    const u32 max_old = static_cast<u32>(sv::wave_max(doit ? static_cast<int>(n_old) : 0));
    for (u32 r0 = 0; r0 < max_old; r0 += UP * GW) {
      u32 rr[UP], vv[UP];
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        rr[u] = r0 + static_cast<u32>(GW) * static_cast<u32>(u) + static_cast<u32>(gl);
        vv[u] = (doit && rr[u] < n_old) ? g.order[rr[u]] : 0u;
      }
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        if (doit && rr[u] < n_old) {
          u32 lo = 0, hi = n_new;  // upper_bound(nslot, r)
          while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (nslot[mid] <= rr[u]) lo = mid + 1;
            else hi = mid;
          }
          g.order2[rr[u] + lo] = static_cast<u16>(vv[u]);
          rb16[2 * static_cast<size_t>(vv[u])] = static_cast<u16>(rr[u] + lo);
        }
      }
    }
    const u32 max_new = static_cast<u32>(sv::wave_max(doit ? static_cast<int>(n_new) : 0));
    for (u32 t = static_cast<u32>(gl); t < max_new; t += GW) {
      if (doit && t < n_new) {
        const u32 r = static_cast<u32>(nslot[t]) + t;
        g.order2[r] = static_cast<u16>(n_old + t);
        rb16[2 * static_cast<size_t>(n_old + t)] = static_cast<u16>(r);
      }
    }
    if (doit) flip = !flip;
  }
  if (act && why == 0) nn = n_old + n_new;
  sv::sync();
  t_ord += sv::clock() - t0;
  return why;
}

// Consensus of a finished window: spoa's heaviest bundle with the node scores in the WAVE's LDS (the DP of every window
// of the wave is over by now), first four in-edges of 64 nodes at a time in registers (as poa2_consensus), then branch
// completion + racon's coverage trim on lane 0 and a parallel output copy.
__host__ __device__ inline void poa4_consensus_general(Poa2Slot& g, u32 n_nodes, u32 nmax, const PoaWindow& win, int trim, Poa4Lds& S,
                                               u8* out, u32* out_len) {
  const int lane = sv::lane();
  // (the lane-0 code shared with poa2.hip wants a node's rank in rank_of[], which this kernel does not keep per layer)
  for (u32 r = lane; r < n_nodes; r += 64) g.rank_of[g.order[r]] = static_cast<u16>(r);
  constexpr u32 kCap = sizeof(Poa4Lds) / 4;
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

// The same for the common case, with nothing but registers and LDS between the graph and the consensus (round 5; the
// general form above took 10 % of the kernel's wave cycles: every node's score went through LDS and the walk back along
// the predecessors was ~600 dependent loads from global memory on lane 0).  Along the rank order a node's in-edges reach
// back a few ranks (at most 23 inside any layer's subgraph), so the scores of the last 64 ranks live in ONE vector register
// — lane = rank mod 64, read with v_readlane, written with v_writelane: the serial loop over the nodes is scalar code with
// no memory access at all —, the predecessors go to LDS (u16 per node) where the walk back follows them, and the stack of
// the consensus nodes stays in LDS for the output copy.  A window beyond what this form holds (more than 2 436 nodes, an
// in-edge reaching back more than 63 ranks) or whose heaviest path does not end in an end node (spoa's branch completion
// rewrites scores) takes the general form.  Same tie rules, same results.
__host__ __device__ inline void poa4_consensus(Poa2Slot& g, const u32* rb, u32 n_nodes, u32 nmax, const PoaWindow& win, int trim,
                                               Poa4Lds& S, u8* out, u32* out_len) {
  const int lane = sv::lane();
  constexpr u32 kCapNodes = sizeof(Poa4Lds) / 4;  // u16 predecessor per node + u16 stack entry per consensus node
  u16* lpred = reinterpret_cast<u16*>(&S);
  u16* lstack = lpred + kCapNodes;
  bool general = n_nodes > kCapNodes;
  i32 maxn = -1, max_sc = 0;
  if (!general) {
    int ring = 0;  // lane l: score of the last node computed at a rank = l (mod 64)
    u32 bad = 0;
    for (u32 r0 = 0; r0 < n_nodes; r0 += 64) {
      const u32 rows = n_nodes - r0 < 64 ? n_nodes - r0 : 64;
      int m_it = 0, m_c = 0, m_lb = 0, m_t01 = 0, m_t23 = 0, m_w0 = 0, m_w1 = 0, m_w2 = 0, m_w3 = 0;
      if (static_cast<u32>(lane) < rows) {
        const u32 r = r0 + static_cast<u32>(lane);
        m_it = g.order[r];
        m_c = g.in_cnt[m_it];
        const u16* tp = g.in_tail + static_cast<size_t>(m_it) * kPoaMaxIn;
        const i32* wp = g.in_w + static_cast<size_t>(m_it) * kPoaMaxIn;
        const uint2 t4 = *reinterpret_cast<const uint2*>(tp);
        const int4 w4 = *reinterpret_cast<const int4*>(wp);
        m_t01 = static_cast<int>(t4.x);
        m_t23 = static_cast<int>(t4.y);
        m_w0 = w4.x;
        m_w1 = w4.y;
        m_w2 = w4.z;
        m_w3 = w4.w;
        u32 lb = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // how many ranks the first four in-edges reach back
          const u32 t = ((k < 2 ? t4.x : t4.y) >> (16 * (k & 1))) & 0xFFFFu;
          const u32 d = static_cast<u32>(k) < static_cast<u32>(m_c) ? r - (rb[t] & 0xFFFFu) : 1u;
          if (d < 1 || d > 63) bad = 1;
          lb |= (d & 63u) << (8 * k);
        }
        m_lb = static_cast<int>(lb);
      }
      int predv = 0xFFFF;
      for (u32 l = 0; l < rows; ++l) {
        const int li = static_cast<int>(l);
        const u32 r = r0 + l;
        const u32 it = static_cast<u32>(sv::rl(m_it, li));
        const u32 c = static_cast<u32>(sv::rl(m_c, li));
        const u32 lb = static_cast<u32>(sv::rl(m_lb, li));
        const u32 t01 = static_cast<u32>(sv::rl(m_t01, li)), t23 = static_cast<u32>(sv::rl(m_t23, li));
        const i32 w0 = sv::rl(m_w0, li), w1 = sv::rl(m_w1, li), w2 = sv::rl(m_w2, li), w3 = sv::rl(m_w3, li);
        i32 sc = -1, pd = -1, pd_sc = 0;
        for (u32 k = 0; k < c; ++k) {
          i32 wgt, t;
          u32 d;
          if (k < 4) {
            t = static_cast<i32>(((k < 2 ? t01 : t23) >> (16 * (k & 1))) & 0xFFFFu);
            wgt = k == 0 ? w0 : (k == 1 ? w1 : (k == 2 ? w2 : w3));
            d = (lb >> (8 * k)) & 0xFFu;
          } else {  // (1 % of the nodes)
            wgt = g.in_w[static_cast<size_t>(it) * kPoaMaxIn + k];
            t = static_cast<i32>(g.in_tail[static_cast<size_t>(it) * kPoaMaxIn + k]);
            d = r - (rb[t] & 0xFFFFu);
            if (d < 1 || d > 63) {
              bad = 1;
              d = 1;
            }
          }
          const i32 st = sv::rl(ring, static_cast<int>((r - d) & 63u));
          if (sc < wgt || (sc == wgt && pd_sc <= st)) {
            sc = wgt;
            pd = t;
            pd_sc = st;
          }
        }
        if (pd != -1) sc += pd_sc;
        ring = sv::wl(ring, sc, li);
        predv = sv::wl(predv, pd == -1 ? 0xFFFF : pd, li);
        if (maxn == -1 || max_sc < sc) {
          maxn = static_cast<i32>(it);
          max_sc = sc;
        }
      }
      if (static_cast<u32>(lane) < rows) lpred[m_it] = static_cast<u16>(predv);
    }
    general = sv::any(bad != 0) || (maxn >= 0 && g.out_cnt[maxn] != 0);  // (the latter: spoa's branch completion)
  }
  if (general) {
    sv::sync();
    poa4_consensus_general(g, n_nodes, nmax, win, trim, S, out, out_len);
    return;
  }
  lds_order();  // lpred
  u32 cl = 0;
  i32 begin = 0, end = -1;
  if (lane == 0) {
    // the walk back along the predecessors (reverse order into the stack), then racon's coverage trim — as
    // poa_consensus_trace_lane0 (poa.h), everything it follows in LDS
    u32 cur = maxn < 0 ? 0xFFFFu : static_cast<u32>(maxn);
    while (cur != 0xFFFFu && cl < kCapNodes) {
      lstack[cl++] = static_cast<u16>(cur);
      cur = lpred[cur];
    }
    end = static_cast<i32>(cl) - 1;
    if (trim) {
      const u32 avg = (win.n_layers - 1) / 2;
      auto cov = [&](i32 pos) -> u32 {  // coverage of a consensus node = visits of the node + of its aligned nodes
        const u32 v = lstack[cl - 1 - static_cast<u32>(pos)];
        u32 c = g.visits[v];
        for (u32 k = 0; k < g.al_cnt[v]; ++k) c += g.visits[g.al[v * 4 + k]];
        return c;
      };
      for (; begin < static_cast<i32>(cl); ++begin)
        if (cov(begin) >= avg) break;
      for (; end >= 0; --end)
        if (cov(end) >= avg) break;
      if (begin >= end) {  // racon: warning only, consensus kept untrimmed
        begin = 0;
        end = static_cast<i32>(cl) - 1;
      }
    }
  }
  cl = static_cast<u32>(sv::rfl(static_cast<int>(cl)));
  begin = sv::rfl(begin);
  end = sv::rfl(end);
  lds_order();  // lstack
  i32 n_out = end - begin + 1;
  if (n_out < 0) n_out = 0;
  if (static_cast<u32>(n_out) > win.out_cap) n_out = static_cast<i32>(win.out_cap);
  for (i32 p = lane; p < n_out; p += 64) out[p] = g.code[lstack[cl - 1 - static_cast<u32>(begin + p)]];
  if (lane == 0) *out_len = static_cast<u32>(n_out);
}

// ---- the phases of a window's layer -------------------------------------------------------------------------------------
// What a window carries from phase to phase lives in an 80-byte record beside its graph (Poa4Win); a phase function reads
// it, does its part for one window (graph side) or for the wave's four (alignment side) and lane 0 stores the new state.
enum : u32 { kIdle = 0, kRunning = 1, kLayersDone = 2, kFinal = 3, kFailed = 4, kHandedOn = 5 };  // (kHandedOn: queued for the 64-column attempt of this launch — whoever runs it writes the window's result)

struct Poa4Win {  // per window in flight (four per resident wave)
  u32 wi;        // window index
  u32 phase, status, nn, n_eff, li, flip;
  u32 act, full, len, lb, span;  // the layer of this round (act = 0: none)
  u32 r_lo, n_rows, t_end, best_rho1;
  u32 b_first;   // band start of the layer's first row
  u32 dflag;     // != 0: the descriptor pass found the layer beyond this kernel's limits
  u32 cells_full, cells_band;  // work counters: rows of the layers' subgraphs x layer length / x band width
};
static_assert(sizeof(Poa4Win) == 80, "state record");
__host__ __device__ __forceinline__ void atomic_max_u32(u32* p, u32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  atomicMax(p, v);
#else
  if (v > *p) *p = v;
#endif
}
__host__ __device__ __forceinline__ void atomic_add_u32(u32* p, u32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  atomicAdd(p, v);
#else
  *p += v;
#endif
}

struct Poa4Ctx {  // what a phase function needs beside the batch description
  Poa4Win* st;     // state records, four per resident wave
  u32 first;       // position of the batch's first window in scheduling order
  u32 count;       // windows of the batch
  u32 quad_delta;  // (group of four windows the wave is working on) - (the wave's own index), modulo 2^32: the wave's records
                   // and scratch slots are its own, the windows are those of the group (0 when the emulator steps the
                   // phases wave by wave in lock step)
  u32 gw;          // windows per group, 1 .. 4 (0 = 4): a batch that does not fill the machine's wave slots with groups of
                   // four is dealt out in smaller groups (round 6) — a wave's layer then has fewer graph sides in front of
                   // its NW, and the latency of a group is what a small batch costs.  The wave's other slots stay empty.
};
__host__ __device__ __forceinline__ u32 poa4_group_windows(const Poa4Ctx& C) { return C.gw ? C.gw : static_cast<u32>(P4::G); }

__host__ __device__ __forceinline__ unsigned char* poa4_slot_of(const Poa4Args& A, const Poa4Ctx&, u32 wave, int q) {
  return A.scratch + (static_cast<size_t>(wave) * P4::G + static_cast<size_t>(q)) * A.slot_bytes;
}
// the lane's window of this wave: record index, or 0xFFFFFFFF beyond the batch
__host__ __device__ __forceinline__ u32 poa4_my_record(const Poa4Ctx& C, u32 wave, int q) {
  const u32 gw = poa4_group_windows(C);
  const u32 pos = (wave + C.quad_delta) * gw + static_cast<u32>(q);  // position within the batch
  return (static_cast<u32>(q) < gw && pos < C.count) ? wave * P4::G + static_cast<u32>(q) : 0xFFFFFFFFu;
}
__host__ __device__ __forceinline__ u32 poa4_position(const Poa4Ctx& C, u32 wave, int q) {
  return C.first + (wave + C.quad_delta) * poa4_group_windows(C) + static_cast<u32>(q);
}

// phase 0: graph of the backbone, state record
__host__ __device__ inline void poa4_phase_init(const Poa4Args& A, const Poa4Ctx& C, u32 wave) {
  const int lane = sv::lane();
  const unsigned long long t_init0 = sv::clock();
  for (int q2 = 0; q2 < P4::G; ++q2) {
    const u32 rec = poa4_my_record(C, wave, q2);
    if (rec == 0xFFFFFFFFu) continue;
    const u32 pos = poa4_position(C, wave, q2);
    const u32 wi = A.sched ? A.sched[pos] : pos;
    const PoaWindow wq = A.windows[wi];
    const Poa4Slot sl2 = poa4_carve(poa4_slot_of(A, C, wave, q2), A.nmax, A.lmax);
    Poa2Slot g = sl2.g;
    u32 nn2 = 0, ne2 = 0;
    const u32 r = poa4_init_window(A, wq, g, sl2.rb, wi, nn2, ne2);
    if (lane == 0) {
      Poa4Win w{};
      w.wi = wi;
      w.phase = r == 1 ? kRunning : kFinal;
      w.status = r == 1 ? 0u : r;
      w.nn = nn2;
      w.n_eff = ne2;
      w.li = 1;
      C.st[rec] = w;
    }
  }
  if (A.phase_cycles && lane == 0) sv::atomic_add(&A.phase_cycles[15], sv::clock() - t_init0);
}

// spoa Graph::Subgraph as marks, for one window by one wave: the ancestors (through in-edges and aligned nodes) of
// backbone node `end` among the nodes with id >= begin; sub_out = out-degree inside the subgraph; returns the rank range
// [r_lo, r_hi) that holds the marked nodes.  poa.h's version walks the graph with a stack on lane 0 — a chain of
// dependent global loads per node.  Here the nodes are swept in DECREASING rank, 64 ranks at a time: the lanes fetch
// their node's in-edges and aligned group and turn them into rank distances (everything a block needs is in flight
// together), then the wave walks the block's 64 nodes in order on 64-bit masks of marks kept in scalar registers (this
// block, the one above, the one below; a push further down goes through a bit array in LDS).  A node is marked iff its
// id is >= begin and it, or a member of its aligned group, was reached through an in-edge of a marked node: members of
// a column are reached only from later columns, which the sweep has left behind by then.
__host__ __device__ inline void poa4_subgraph_marks(Poa2Slot& g, const u32* rb, const u16* order, u32* pend, u32 n_nodes,
                                                    u32 begin, u32 end, u32& r_lo_out, u32& r_hi_out) {
  const int lane = sv::lane();
  for (u32 i = lane; i < n_nodes; i += 64) {
    g.mark[i] = 0;
    g.sub_out[i] = 0;
  }
  for (u32 i = lane; i < 256; i += 64) pend[i] = 0;  // rank-indexed bits for pushes beyond the block below
  lds_order();
  const u32 end_rank = rb[end] & 0xFFFFu;
  // (the sweep starts three ranks above `end`: the other letters of its column are reached from it and rank behind it)
  const u32 top_rank = end_rank + 3 < n_nodes ? end_rank + 3 : n_nodes - 1;
  unsigned long long w_cur = 0, w_above = 0, w_low = 0;
  u32 r_lo = 0xFFFFFFFFu, r_hi = 0;
  const i32 top_blk = static_cast<i32>(top_rank >> 6);
  if ((end_rank >> 6) == static_cast<u32>(top_blk)) w_cur = 1ULL << (end_rank & 63u);
  else w_low = 1ULL << (end_rank & 63u);
  for (i32 blk = top_blk; blk >= 0; --blk) {
    const u32 r0 = static_cast<u32>(blk) << 6;
    const u32 r = r0 + static_cast<u32>(lane);
    const bool in = r < n_nodes && r <= top_rank;
    const u32 v = in ? order[r] : 0u;
    u32 c = in ? g.in_cnt[v] : 0u;
    const u32 ac = in ? g.al_cnt[v] : 0u;
    const uint4 ta = *reinterpret_cast<const uint4*>(g.in_tail + static_cast<size_t>(v) * kPoaMaxIn);
    const uint2 al2 = *reinterpret_cast<const uint2*>(g.al + static_cast<size_t>(v) * 4);
    const bool ok = in && v >= begin;
    // rank distances of the first eight in-edges' tails (0: none, or a tail below `begin`) and of the aligned nodes
    u32 rec_a = 0, rec_b = 0, rec_f = ok ? 1u : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u32 wd = k < 2 ? ta.x : (k < 4 ? ta.y : (k < 6 ? ta.z : ta.w));
      const u32 t = static_cast<u32>(k) < c ? (wd >> (16 * (k & 1))) & 0xFFFFu : 0u;
      const u32 rt = rb[t] & 0xFFFFu;
      u32 lb = (static_cast<u32>(k) < c && t >= begin) ? r - rt : 0u;
      if (lb > 254u) {  // (a very long in-edge: through the slow path below)
        lb = 0;
        rec_f |= 2u;
      }
      if (k < 4) rec_a |= lb << (8 * k);
      else rec_b |= lb << (8 * (k - 4));
    }
    if (c > 8) rec_f |= 2u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const u32 m = k == 0 ? al2.x & 0xFFFFu : (k == 1 ? al2.x >> 16 : al2.y & 0xFFFFu);
      const bool has = static_cast<u32>(k) < ac;
      const u32 rm = rb[has ? m : 0u] & 0xFFFFu;
      const i32 d = has ? static_cast<i32>(rm) - static_cast<i32>(r) : 0;  // |d| <= 3: a column's nodes are rank-contiguous
      const u32 enc = (has && d >= -7 && d <= 7 && d != 0) ? static_cast<u32>(d + 8) : 0u;
      if (has && enc == 0) rec_f |= 2u;
      rec_f |= enc << (4 + 4 * k);
    }
    const int top_l = blk == top_blk ? static_cast<int>(top_rank & 63u) : 63;
    for (int l = top_l; l >= 0; --l) {
      const u32 f = static_cast<u32>(sv::rl(static_cast<int>(rec_f), l));
      bool reached = ((w_cur >> l) & 1ULL) != 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const u32 enc = (f >> (4 + 4 * k)) & 15u;
        if (enc) {
          const int pos = l + static_cast<int>(enc) - 8;
          const bool bit = pos >= 64 ? ((w_above >> (pos - 64)) & 1ULL) != 0
                                     : (pos >= 0 ? ((w_cur >> pos) & 1ULL) != 0 : ((w_low >> (pos + 64)) & 1ULL) != 0);
          reached = reached || bit;
        }
      }
      if (f & 2u) {  // slow path: a far tail, more than eight in-edges, or an aligned node out of place
        const u32 vs = order[r0 + static_cast<u32>(l)];
        const u32 as = g.al_cnt[vs];
        for (u32 k = 0; k < as; ++k) reached = reached || g.mark[g.al[vs * 4 + k]] != 0 ||
                                                 ((pend[(rb[g.al[vs * 4 + k]] & 0xFFFFu) >> 5] >> (rb[g.al[vs * 4 + k]] & 31u)) & 1u) != 0;
      }
      const bool marked = (f & 1u) != 0 && (reached || ((pend[(r0 + l) >> 5] >> ((r0 + l) & 31u)) & 1u) != 0);
      if (marked) {
        w_cur |= 1ULL << l;
        const u32 a = static_cast<u32>(sv::rl(static_cast<int>(rec_a), l)), b = static_cast<u32>(sv::rl(static_cast<int>(rec_b), l));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const u32 lb = ((k < 4 ? a : b) >> (8 * (k & 3))) & 0xFFu;
          if (lb) {
            const int pos = l - static_cast<int>(lb);
            if (pos >= 0) w_cur |= 1ULL << pos;
            else if (pos >= -64) w_low |= 1ULL << (pos + 64);
            else if (lane == 0) pend[(r0 + l - lb) >> 5] |= 1u << ((r0 + l - lb) & 31u);
          }
        }
        if (f & 2u) {  // slow path: every in-edge straight from the graph
          const u32 vs = order[r0 + static_cast<u32>(l)];
          const u32 cs = g.in_cnt[vs];
          for (u32 k = 0; k < cs; ++k) {
            const u32 t = g.in_tail[vs * kPoaMaxIn + k];
            if (t < begin) continue;
            const u32 rt = rb[t] & 0xFFFFu;
            if (rt >= r0) w_cur |= 1ULL << (rt - r0);
            else if (rt + 64 >= r0) w_low |= 1ULL << (rt + 64 - r0);
            else if (lane == 0) pend[rt >> 5] |= 1u << (rt & 31u);
          }
        }
        lds_order();
      } else {
        w_cur &= ~(1ULL << l);
      }
    }
    if (in && ((w_cur >> lane) & 1ULL)) {
      g.mark[v] = 1;
      r_lo = r < r_lo ? r : r_lo;
      r_hi = r + 1 > r_hi ? r + 1 : r_hi;
    }
    w_above = w_cur;
    w_cur = w_low;
    w_low = 0;
  }
  sv::sync();
  for (u32 v = lane; v < n_nodes; v += 64) {
    if (!g.mark[v]) continue;
    const u32 c = g.in_cnt[v];
    for (u32 k = 0; k < c; ++k) {
      const u32 t = g.in_tail[v * kPoaMaxIn + k];
      if (g.mark[t]) sv::atomic_add(reinterpret_cast<unsigned int*>(g.sub_out) + (t >> 1), (t & 1) ? 0x10000u : 1u);
    }
  }
  {
    const i32 lo_neg = sv::wave_max(r_lo == 0xFFFFFFFFu ? -0x7FFFFFFF : -static_cast<i32>(r_lo));
    const i32 hi = sv::wave_max(static_cast<i32>(r_hi));
    r_lo_out = lo_neg == -0x7FFFFFFF ? 0u : static_cast<u32>(-lo_neg);
    r_hi_out = lo_neg == -0x7FFFFFFF ? 0u : static_cast<u32>(hi);
  }
  sv::sync();
}

// phase A1 of a round, one window per wave: the window's next layer (codes packed 2 bits per base into the window's
// scratch, alignment results reset, subgraph marks and the rank range of the subgraph for a partial layer)
template <class K>
__host__ __device__ inline void poa4_phase_setup(const Poa4Args& A, const Poa4Ctx& C, Poa4LdsDesc& S, u32 rec) {
  const int lane = sv::lane();
  const u32 wave_dp = rec / P4::G;
  const int q = static_cast<int>(rec % P4::G);
  if (poa4_my_record(C, wave_dp, q) == 0xFFFFFFFFu) return;
  Poa4Win me = C.st[rec];
  sv::sync();  // (read by every lane before lane 0 stores the new state)
  if (me.phase != kRunning) return;
  const unsigned long long t0 = sv::clock();
  const PoaWindow wq = A.windows[me.wi];
  u32 liq = me.li;
  while (liq < wq.n_layers &&
         (A.layers[wq.layer_first + liq].len == 0 || (A.src.layer_ok && !A.src.layer_ok[wq.layer_first + liq])))
    ++liq;
  me.act = 0;
  me.li = liq;
  if (liq >= wq.n_layers) {
    me.phase = kLayersDone;
    if (lane == 0) C.st[rec] = me;
    return;
  }
  const PoaLayer L = A.layers[wq.layer_first + liq];
  if (L.len > A.lmax || L.len > static_cast<u32>(kPoa2MaxSeq)) {
    me.phase = kFailed;
    me.status = 4;
    if (lane == 0) C.st[rec] = me;
    return;
  }
  const Poa4Slot sl = poa4_carve(poa4_slot_of(A, C, wave_dp, q), A.nmax, A.lmax);
  Poa2Slot g = sl.g;
  for (u32 i = lane; i < L.len; i += 64) {
    S.bytes[i] = static_cast<u8>(poa_layer_code(A.src, L, i));
    g.pos_node[i] = static_cast<u16>(kNone4);
  }
  lds_order();
  if (lane < 60) {
    u32 x = 0;
    for (u32 c = 0; c < 16; ++c) {
      const i32 p = static_cast<i32>(static_cast<u32>(lane) * 16 + c) - 1;
      if (p >= 0 && p < static_cast<i32>(L.len)) x |= static_cast<u32>(S.bytes[p] & 3u) << (2 * c);
    }
    sl.seq2g[lane] = x;
  }
  const u32 blen = A.layers[wq.layer_first].len;
  const u32 offset = static_cast<u32>(0.01 * blen);
  const bool full = L.begin < offset && L.end > blen - offset;
  u32 r_lo = 0, r_hi = me.nn;
  if (!full) {
    lds_order();  // (the byte codes are packed: their LDS becomes the sweep's bit array)
    poa4_subgraph_marks(g, sl.rb, me.flip ? g.order2 : g.order, reinterpret_cast<u32*>(S.bytes), me.nn, L.begin, L.end, r_lo, r_hi);
  }
  const i32 lb = static_cast<i32>(L.begin), span = static_cast<i32>(L.end) - static_cast<i32>(L.begin) + 1;
  poa4_guide_to_lds(S, A.layers + wq.layer_first + liq, L.len, span);
  lds_order();
  if (lane < 32) sl.seq2g[64 + lane] = S.segtab[lane];
  u32 b_first = 0;
  if (r_hi > r_lo) {
    const u32 v0 = (me.flip ? g.order2 : g.order)[r_lo];
    b_first = static_cast<u32>(poa4_band_start<K>(S, static_cast<i32>(sl.rb[v0] >> 16), lb, span,
                                                   magic_of(static_cast<u32>(span > 0 ? span : 1)), L.len));
  }
  me.act = 1;
  me.full = full ? 1u : 0u;
  me.len = L.len;
  me.lb = L.begin;
  me.span = static_cast<u32>(span);
  me.r_lo = r_lo;
  me.n_rows = r_hi - r_lo;
  me.t_end = 0;
  me.best_rho1 = 0;
  me.b_first = b_first;
  me.dflag = 0;
  if (lane == 0) C.st[rec] = me;
  if (A.phase_cycles && lane == 0) sv::atomic_add(&A.phase_cycles[0], sv::clock() - t0);
}

// The row descriptors of a layer, by the window's wave in two passes over the nodes (64 x kDescPer nodes per turn).  Pass 1
// streams over the nodes and leaves rank | band start of every node for THIS layer in rbl[]; pass 2 builds the
// descriptors: everything that depends only on (window, node) is a coalesced load (rb[] holds rank and backbone coordinate
// of a node in one word), an in-edge tail's rank and band start come from rbl[] with one gather (round 4 re-evaluated the
// guide per edge: nine evaluations per node), and in-edges beyond the largest in-degree of the 256 nodes in flight are not
// looked at at all.  The descriptor of every node whose rank lies in the rank range [r_lo, r_hi) of the layer's subgraph
// is written at its row rho = rank - r_lo.
constexpr int kDescPer = 4;
template <class K>
__host__ __device__ inline void poa4_phase_desc_onewave(const Poa4Args& A, const Poa4Ctx& C, Poa4LdsDesc& S, u32 rec) {
  const int lane = sv::lane();
  const u32 wave_dp = rec / P4::G;
  const int q = static_cast<int>(rec % P4::G);
  if (poa4_my_record(C, wave_dp, q) == 0xFFFFFFFFu) return;
  const Poa4Slot sl = poa4_carve(poa4_slot_of(A, C, wave_dp, q), A.nmax, A.lmax);
  const Poa2Slot& g = sl.g;
  const Poa4Win me = C.st[rec];
  if (me.phase != kRunning || !me.act) return;
  const unsigned long long t_desc0 = sv::clock();
  const u32 nn = me.nn;
  const bool full = me.full != 0;
  const u32 len = me.len, r_lo = me.r_lo, n_rows = me.n_rows, r_hi = r_lo + n_rows;
  const i32 lb = static_cast<i32>(me.lb), span = static_cast<i32>(me.span), b_first = static_cast<i32>(me.b_first);
  const u32 span_magic = magic_of(static_cast<u32>(span > 0 ? span : 1));
  const u32 dump_off = poa4_dump_byte(q);
  const u32 neg_off = poa4_neg_byte(q);
  const u32 neg2 = neg_off | (neg_off << 16);
  u32 flag = 0, marked_rows = 0;
  i32 t_end = 0;
  {  // the layer's packed codes and guide into LDS
    const u32 gw = sl.seq2g[lane < 60 ? lane : 64 + (lane - 60)];
    const u32 gw2 = lane < 28 ? sl.seq2g[68 + lane] : 0u;
    if (lane < 60) S.seq2[lane] = gw;
    else S.segtab[lane - 60] = gw;
    if (lane < 28) S.segtab[4 + lane] = gw2;
  }
  lds_order();
  // ---- pass 1: rank | band start of every node ----
  for (u32 v0 = 0; v0 < nn; v0 += 64 * kDescPer) {
    u32 rbv[kDescPer];
#pragma unroll
    for (int u = 0; u < kDescPer; ++u) {
      const u32 v = v0 + static_cast<u32>(u) * 64 + static_cast<u32>(lane);
      rbv[u] = v < nn ? sl.rb[v] : 0u;
    }
#pragma unroll
    for (int u = 0; u < kDescPer; ++u) {
      const u32 v = v0 + static_cast<u32>(u) * 64 + static_cast<u32>(lane);
      if (v < nn) {
        const i32 b = poa4_band_start<K>(S, static_cast<i32>(rbv[u] >> 16), lb, span, span_magic, len);
        sl.rbl[v] = (rbv[u] & 0xFFFFu) | (static_cast<u32>(b) << 16);
      }
    }
  }
  sv::sync();  // rbl[]: written above by this wave, gathered below
  // ---- pass 2: the descriptors ----
  for (u32 v0 = 0; v0 < nn; v0 += 64 * kDescPer) {
    u32 vv[kDescPer], rbv[kDescPer], cc[kDescPer], code[kDescPer], outc_f[kDescPer], outc_s[kDescPer], mk[kDescPer];
    uint4 tl[kDescPer];
#pragma unroll
    for (int u = 0; u < kDescPer; ++u) {
      vv[u] = v0 + static_cast<u32>(u) * 64 + static_cast<u32>(lane);
      const u32 vq = vv[u] < nn ? vv[u] : 0u;
      rbv[u] = sl.rbl[vq];
      cc[u] = g.in_cnt[vq];
      code[u] = g.code[vq];
      outc_f[u] = g.out_cnt[vq];
      outc_s[u] = full ? 0u : g.sub_out[vq];
      mk[u] = full ? 1u : g.mark[vq];
      tl[u] = *reinterpret_cast<const uint4*>(g.in_tail + static_cast<size_t>(vq) * kPoaMaxIn);
    }
    bool ok[kDescPer], marked[kDescPer];
    u32 cmax = 0;
#pragma unroll
    for (int u = 0; u < kDescPer; ++u) {
      const u32 r = rbv[u] & 0xFFFFu;
      ok[u] = vv[u] < nn && r >= r_lo && r < r_hi;
      marked[u] = ok[u] && mk[u] != 0;
      if (!marked[u]) cc[u] = 0;
      cmax = cc[u] > cmax ? cc[u] : cmax;
    }
    cmax = static_cast<u32>(sv::wave_max(static_cast<i32>(cmax)));  // in-edges anybody of this turn has (uniform)
    u32 trb[kDescPer][8], tmk[kDescPer][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (static_cast<u32>(k) < cmax) {
#pragma unroll
        for (int u = 0; u < kDescPer; ++u) {
          const u32 wd = k < 2 ? tl[u].x : (k < 4 ? tl[u].y : (k < 6 ? tl[u].z : tl[u].w));
          const u32 t = static_cast<u32>(k) < cc[u] ? (wd >> (16 * (k & 1))) & 0xFFFFu : 0u;
          trb[u][k] = sl.rbl[t];
          tmk[u][k] = full ? 1u : g.mark[t];
        }
      } else {
#pragma unroll
        for (int u = 0; u < kDescPer; ++u) {
          trb[u][k] = 0;
          tmk[u][k] = 0;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kDescPer; ++u) {
      const u32 v = vv[u];
      const u32 r = rbv[u] & 0xFFFFu;
      const i32 b = static_cast<i32>(rbv[u] >> 16);
      const u32 rho = r - r_lo;
      u32 ep[4] = {neg2, neg2, neg2, neg2};
      u32 np = 0, lbw = 0, lbk7 = 0;
      auto edge = [&](u32 rbt, bool inside) {
        if (!inside) return;
        const u32 lbk = r - (rbt & 0xFFFFu);
        const i32 d = b - static_cast<i32>(rbt >> 16);
        if (lbk < 1 || lbk > static_cast<u32>(K::kRing - 1) || lbk > rho || d < 0 || d > K::kMaxD) {
          flag = 7;
        } else if (np < static_cast<u32>(K::kEdges)) {
          // (the row of the tail's ring slot: where a column lies in it does not depend on either band start)
          const u32 e = poa4_ring_byte(q, ((rho - lbk) % static_cast<u32>(K::kRing)) * static_cast<u32>(kRowW4));
          const u32 idx = np >> 1;
          const u32 keep = (np & 1) ? 0x0000FFFFu : 0xFFFF0000u;
          const u32 put = (np & 1) ? e << 16 : e;
#pragma unroll
          for (u32 i = 0; i < 4; ++i) ep[i] = i == idx ? ((ep[i] & keep) | put) : ep[i];
          if (np < 6) lbw |= lbk << (5 * np);
          if (np == 7) lbk7 = lbk;
        } else if (np < static_cast<u32>(K::kEdgesMax)) {
          // a ninth .. fifteenth in-edge (only the loop over in-edges 8.. below gets here: one row in ~1 000 windows).  The row's
          // overflow record takes in-edges 7..14 — the eighth moves out of the descriptor, so that code 0 of such a row is
          // "horizontal" and nothing else — and the NW's rare path reads their cells of the column pair BEFORE the step's as well
          // (it keeps no "cell left of this step's" for them), a step later than the descriptor's in-edges are read: one step less
          // of the margin before the ring slot's next owner stores, i.e. in-edges of at most kRing - 2 ranks.
          const u32 e = poa4_ring_byte(q, ((rho - lbk) % static_cast<u32>(K::kRing)) * static_cast<u32>(kRowW4));
          u16* const o16 = reinterpret_cast<u16*>(sl.ovf + rho);
          if (np == static_cast<u32>(K::kEdges)) {
            sl.ovf[rho] = uint4{(ep[3] >> 16) | (neg_off << 16), neg2, neg2, neg2};
            ep[3] = (ep[3] & 0xFFFFu) | (neg_off << 16);
            if (lbk7 > static_cast<u32>(K::kRing - 2)) flag = 7;
          }
          o16[np - 7] = static_cast<u16>(e);
          if (lbk > static_cast<u32>(K::kRing - 2)) flag = 7;
        }
        ++np;
      };
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (static_cast<u32>(k) < cmax) edge(trb[u][k], static_cast<u32>(k) < cc[u] && tmk[u][k] != 0);
      for (u32 k = 8; k < cc[u]; ++k) {  // rare
        const u32 t = g.in_tail[static_cast<size_t>(v) * kPoaMaxIn + k];
        edge(sl.rbl[t], full || g.mark[t] != 0);
      }
      if (np > static_cast<u32>(K::kEdgesMax)) flag = 3;
      if (ok[u]) {
        // match mask of the row's 32 columns against the layer
        u32 mm = 0;
        {
          const u32 wi = static_cast<u32>(b) >> 4, sh = 2u * (static_cast<u32>(b) & 15u);
          const u32 x0 = S.seq2[wi], x1 = S.seq2[wi + 1], x2 = S.seq2[wi + 2];
          const u32 pat = code[u] * 0x55555555u;
          const u32 elo = funnel_shr(x1, x0, sh) ^ pat, ehi = funnel_shr(x2, x1, sh) ^ pat;
          auto even_bits = [](u32 y) -> u32 {
            y = ~(y | (y >> 1)) & 0x55555555u;
            y = (y | (y >> 1)) & 0x33333333u;
            y = (y | (y >> 2)) & 0x0F0F0F0Fu;
            y = (y | (y >> 4)) & 0x00FF00FFu;
            y = (y | (y >> 8)) & 0x0000FFFFu;
            return y;
          };
          mm = even_bits(elo) | (even_bits(ehi) << 16);
        }
        i32 sdiff = b - b_first;
        if (sdiff < 0) {  // (never: a node's backbone coordinate does not decrease along the order)
          flag = 7;
          sdiff = 0;
        }
        const u32 Srow = rho + (rho >> 4) + (static_cast<u32>(sdiff) >> 1) + 1u;
        const u32 own = marked[u] ? poa4_ring_byte(q, (rho % static_cast<u32>(K::kRing)) * static_cast<u32>(kRowW4)) : dump_off;
        const bool endn = marked[u] && (full ? outc_f[u] : outc_s[u]) == 0;
        uint4 da, db;
        da.x = (2u * Srow) | (own << 16);
        da.y = v | (static_cast<u32>(b) << 16) | ((np > 15u ? 15u : np) << 26) | (marked[u] ? 1u << 30 : 0u) | (endn ? 1u << 31 : 0u);
        da.z = lbw;
        da.w = ep[0];
        db.x = ep[1];
        db.y = ep[2];
        db.z = ep[3];
        db.w = mm;
        sl.desc[2 * static_cast<size_t>(rho)] = da;
        sl.desc[2 * static_cast<size_t>(rho) + 1] = db;
        t_end = static_cast<i32>(Srow) + 17 > t_end ? static_cast<i32>(Srow) + 17 : t_end;
        if (marked[u]) ++marked_rows;
      }
    }
  }
  // rows beyond the last one: what the lanes' descriptor prefetch runs into
  if (lane < 32) {
    const size_t rho = static_cast<size_t>(n_rows) + static_cast<size_t>(lane);
    sl.desc[2 * rho] = uint4{kInactiveS | (dump_off << 16), 0u, 0u, neg2};
    sl.desc[2 * rho + 1] = uint4{neg2, neg2, neg2, 0u};
  }
  t_end = sv::wave_max(t_end);
  flag = static_cast<u32>(sv::wave_max(static_cast<i32>(flag)));
  marked_rows = sv::wave_sum(marked_rows);
  if (lane == 0) {  // (one wave per window: plain stores)
    Poa4Win& w = C.st[rec];
    w.t_end = static_cast<u32>(t_end);
    w.dflag = flag;
    w.cells_full += marked_rows * len;
    w.cells_band += marked_rows * (len + 1 < 32u ? len + 1 : 32u);
  }
  if (A.phase_cycles && lane == 0) sv::atomic_add(&A.phase_cycles[10], sv::clock() - t_desc0);
}

// phase B: the NW
template <class K, class SC>
__host__ __device__ inline void poa4_phase_dp(const Poa4Args& A, const Poa4Ctx& C, Poa4Lds& S, u32 wave) {
  const int lane = sv::lane();
  const int q = lane / K::GS;
  const unsigned long long t0 = sv::clock();
  const u32 my_rec = poa4_my_record(C, wave, q);
  u32 act = 0, t_end = 0, len = 0, li = 0;
  if (my_rec != 0xFFFFFFFFu) {
    Poa4Win& w = C.st[my_rec];
    act = (w.phase == kRunning && w.act) ? 1u : 0u;
    t_end = w.t_end;
    len = w.len;
    li = w.li;
    if (act && (w.dflag || t_end + 8u > poa4_steps(A.nmax, A.lmax))) {
      // beyond this kernel's limits (in-degree, in-edge length, band step): the 64-column kernel's job
      if ((lane & (K::GS - 1)) == 0) {
        w.phase = kFailed;
        // (bits 24-27: why, for the statistics of the debug build — 1..7 a limit of this kernel: 3 in-degree, 7 band step along
        // an in-edge, 1 steps; 9 the last column in no end node's band; 10 the walk came near a band's edge; 11 the walk met
        // a backpointer it cannot follow)
        w.status = kPoaBandHit | (li << 8) | ((w.dflag ? (w.dflag & 7u) : 1u) << 24);
        w.act = 0;
      }
      act = 0;
    }
  }
  if (!sv::any(act != 0)) return;
  u32 best_rho1 = 0;
  poa4_dp<K, SC>(A, S, poa4_slot_of(A, C, wave, q), act != 0, t_end, len, best_rho1);
  if (act && (lane & (K::GS - 1)) == 0) {
    Poa4Win& w = C.st[my_rec];
    w.best_rho1 = best_rho1;
    if (best_rho1 == 0) {  // the last column is in no end node's band
      w.phase = kFailed;
      w.status = kPoaBandHit | (li << 8) | (9u << 24);
      w.act = 0;
    }
  }
  if (A.phase_cycles && lane == 0) sv::atomic_add(&A.phase_cycles[1], sv::clock() - t0);
}

// phase C: the traceback
template <class K>
__host__ __device__ inline void poa4_phase_tb(const Poa4Args& A, const Poa4Ctx& C, Poa4LdsTb& S, u32 wave) {
  const int lane = sv::lane();
  const int q = lane / K::GS;
  const unsigned long long t0 = sv::clock();
  const u32 my_rec = poa4_my_record(C, wave, q);
  u32 act = 0, r_lo = 0, n_rows = 0, full = 0, len = 0, best = 0, li = 0;
  if (my_rec != 0xFFFFFFFFu) {
    const Poa4Win& w = C.st[my_rec];
    act = (w.phase == kRunning && w.act) ? 1u : 0u;
    r_lo = w.r_lo;
    n_rows = w.n_rows;
    full = w.full;
    len = w.len;
    best = w.best_rho1;
    li = w.li;
  }
  if (!sv::any(act != 0)) return;
  u32 bad = 0, band_hit = 0;
  poa4_traceback<K>(A, S, poa4_slot_of(A, C, wave, q), act != 0, r_lo, n_rows, full != 0, len, best, bad, band_hit);
  if (act && (bad || band_hit) && (lane & (K::GS - 1)) == 0) {
    Poa4Win& w = C.st[my_rec];
    w.phase = kFailed;
    w.status = (bad ? bad : kPoaBandHit) | (li << 8) | ((bad ? 11u : 10u) << 24);
    w.act = 0;
  }
  if (A.phase_cycles && lane == 0) sv::atomic_add(&A.phase_cycles[2], sv::clock() - t0);
}

// phase D: the graph update, one window per wave
constexpr int kUpdPer = 2;  // sequence positions per lane and iteration of the graph update (4: 117 registers instead of 80 in a
                            // kernel of its own; inside the persistent kernel's 128 registers the two-position form is faster)
struct alignas(16) Poa4LdsUpd {
  u16 nslot[kPoa2MaxSeq + 16];  // order slots of the layer's new nodes
  u32 seq2[64];                 // the layer, 2 bits per base
};
template <int UP>
__host__ __device__ inline void poa4_phase_update(const Poa4Args& A, const Poa4Ctx& C, Poa4LdsUpd& S, u32 rec) {
  const int lane = sv::lane();
  const u32 wave_dp = rec / P4::G;
  const int q = static_cast<int>(rec % P4::G);
  if (poa4_my_record(C, wave_dp, q) == 0xFFFFFFFFu) return;
  Poa4Win me = C.st[rec];
  if (me.phase != kRunning) return;
  const bool act = me.act != 0;
  unsigned char* const my_slot = poa4_slot_of(A, C, wave_dp, q);
  if (act && lane < 60) S.seq2[lane] = poa4_carve(my_slot, A.nmax, A.lmax).seq2g[lane];  // the layer's packed codes back into LDS
  lds_order();
  unsigned long long t_add = 0, t_ord = 0;
  u32 nn = me.nn;
  bool flip = me.flip != 0;
  const PoaLayer* Lp = A.layers;
  if (act) Lp = A.layers + A.windows[me.wi].layer_first + me.li;
  const u32 why = poa4_update_graph<64, UP>(A, S.nslot, S.seq2, my_slot, act, Lp, me.len, me.lb, nn, flip, t_add, t_ord);
  if (lane == 0) {
    if (act && why) {
      me.phase = kFailed;
      me.status = why;
    } else if (act) {
      me.nn = nn;
      me.flip = flip ? 1u : 0u;
    }
    me.li = me.li + 1;
    me.act = 0;
    C.st[rec] = me;
  }
  if (A.phase_cycles && lane == 0) {
    sv::atomic_add(&A.phase_cycles[3], t_add);
    sv::atomic_add(&A.phase_cycles[4], t_ord);
  }
}

// last phase: consensus (or the backbone of a window that failed), status
__host__ __device__ inline void poa4_phase_final(const Poa4Args& A, const Poa4Ctx& C, Poa4Lds& S, u32 wave) {
  const int lane = sv::lane();
  const unsigned long long t0 = sv::clock();
  for (int q2 = 0; q2 < P4::G; ++q2) {
    const u32 rec = poa4_my_record(C, wave, q2);
    if (rec == 0xFFFFFFFFu) continue;
    const Poa4Win w = C.st[rec];
    if (w.phase == kHandedOn) continue;  // (result and status: the wave that takes the window off the queue)
    PoaWindow wq = A.windows[w.wi];
    u32 st = w.status;
    if (w.phase == kFailed || w.phase == kRunning) {  // (still running: the host stopped the rounds early — never)
      poa4_copy_backbone(A, wq, A.layers[wq.layer_first], A.out + wq.out_off, A.out_len + w.wi);
      if (w.phase == kRunning) st = 5;
    } else if (w.phase == kLayersDone) {
      Poa2Slot g = poa4_graph(poa4_slot_of(A, C, wave, q2), A.nmax, A.lmax, w.flip != 0);
      wq.n_layers = w.n_eff;
      poa4_consensus(g, poa4_carve(poa4_slot_of(A, C, wave, q2), A.nmax, A.lmax).rb, w.nn, A.nmax, wq, A.trim, S, A.out + wq.out_off,
                     A.out_len + w.wi);
      sv::sync();
      st = 1;
    }
    if (lane == 0) {
      A.status[w.wi] = st;
      if (A.phase_cycles) {
        sv::atomic_add(&A.phase_cycles[6], static_cast<unsigned long long>(w.cells_full));
        sv::atomic_add(&A.phase_cycles[7], static_cast<unsigned long long>(w.cells_band));
      }
    }
  }
  if (A.phase_cycles && lane == 0) sv::atomic_add(&A.phase_cycles[5], sv::clock() - t0);
}

// ---- the two sides of a layer ----------------------------------------------------------------------------------------------
// Phases of the same SHAPE follow each other directly: the graph side (one window on the wave's 64 lanes: update with the
// previous alignment -> the next layer's set-up -> its row descriptors) and the alignment side (the wave's four windows
// side by side: NW -> traceback).  Between two phases the wave's stores are ordered at workgroup scope (sv::phase_fence):
// the next phase reads what this wave just wrote, through the CU's own L1 / the XCD's L2.
struct alignas(16) Poa4LdsGraph {
  union {
    Poa4LdsUpd upd;
    Poa4LdsDesc desc;
  } u;
};
template <class K, int UP>
__host__ __device__ inline void poa4_phase_graph(const Poa4Args& A, const Poa4Ctx& C, Poa4LdsGraph& S, u32 rec) {
  const u32 wave_dp = rec / P4::G;
  const int q = static_cast<int>(rec % P4::G);
  if (poa4_my_record(C, wave_dp, q) == 0xFFFFFFFFu) return;
  {
    const Poa4Win me = C.st[rec];
    sv::sync();  // (every lane has the record as the previous round left it before any phase below rewrites it)
    if (me.phase != kRunning) return;
    if (me.act) {  // the layer aligned in the previous round goes into the graph
      poa4_phase_update<UP>(A, C, S.u.upd, rec);
      sv::phase_fence();
    }
  }
  poa4_phase_setup<K>(A, C, S.u.desc, rec);
  sv::phase_fence();
  poa4_phase_desc_onewave<K>(A, C, S.u.desc, rec);
}
struct alignas(16) Poa4LdsNw {
  union {
    Poa4Lds dp;
    Poa4LdsTb tb;
  } u;
};
template <class K, class SC>
__host__ __device__ inline void poa4_phase_nw(const Poa4Args& A, const Poa4Ctx& C, Poa4LdsNw& S, u32 wave) {
  poa4_phase_dp<K, SC>(A, C, S.u.dp, wave);
  sv::phase_fence();  // backpointer stream and the windows' records: written above, read below
  poa4_phase_tb<K>(A, C, S.u.tb, wave);
}

// ---- one launch for the whole batch: persistent waves (round 5) ------------------------------------------------------------
// Per-round launches (round 4: five phase kernels; first half of round 5: two) keep every window of a chunk in lock step: a
// round ends when its slowest wave ends, and the busy wave-cycles of a C4-like batch added up to ~55 % of the slots x time
// the launches occupied (tools/bench_poa.py phase counters; 918 / 841 ms per C4 round against 739 here).  Here a wave takes a GROUP of four windows (scheduling order = by decreasing layer count, so the four have
// about the same number of layers) from an atomic counter and carries it from the backbone graph to the consensus without
// waiting for anybody else: per layer the graph side of each of its windows in turn (all 64 lanes on one window), then
// the alignment side of the four side by side.  Scratch and state records belong to the WAVE (blockIdx), not to the
// windows: a launch of n resident waves needs 4 n slots whatever the batch size (~10 GB instead of 30 GB per chunk), no
// chunks, no layer histogram, no host round trip.
struct alignas(16) Poa4LdsAll {
  union {
    Poa4LdsGraph g;
    Poa4LdsNw n;
    Poa4Lds f;
    p2::Poa2Lds<1> w64;  // the 64-column window function (poa2_window.h), for a window taken off the queue
    p2::Poa2Lds<2> w128; // and the 128-column one behind it
  } u;
};
static_assert(sizeof(Poa4LdsAll) <= 10240, "sixteen waves per CU");

// ---- the windows the first attempt hands on, inside the same launch (round 6) ------------------------------------------------
// 3 of 200 000 windows of a C4 round (244 before rows of 9..15 in-edges stayed here) need the 64-column band.  As a launch of
// their own behind this one they cost one window's latency whatever their number — 30-38 ms per round, and the reason a batch
// of a few thousand windows did not start here at all.  Now a wave whose window fails with a band hit puts its index into a queue
// in HBM (agent-scope atomics: the waves sit on eight XCDs whose L2s do not see each other's lines) and goes on with the rest of
// its group; every wave looks at the queue between two groups and runs what it finds — one window on the whole wave, poa2's window
// function with its 64-column band, the wave's four scratch slots as that function's one — and again before it leaves.  A window
// pushed after a wave looked is taken by any wave that comes by later, at the latest by the pusher itself: no wave waits for
// another.  What the 64 columns cannot do either comes back with bit 28 set in its status (the host goes on with 128 columns).
constexpr u32 kEscEmpty = 0xFFFFFFFFu;
__host__ __device__ __forceinline__ u32 poa4_esc_load(const u32* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *p;
#endif
}
// wave-uniform; every lane takes part (lane 0 adds one, the others nothing: see poa4_persistent on why not `if (lane == 0)`)
__host__ __device__ inline bool poa4_esc_push(const Poa4Args& A, u32 wi) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 k = __hip_atomic_fetch_add(A.esc, sv::lane() == 0 ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  k = static_cast<u32>(sv::rfl(static_cast<int>(k)));
  if (k >= A.esc_cap) return false;
  if (sv::lane() == 0) __hip_atomic_store(A.esc + 4 + k, wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
#else
  (void)A;
  (void)wi;
  return false;
#endif
}
// the next window of the queue (wave-uniform), kEscEmpty: none right now
__host__ __device__ inline u32 poa4_esc_pop(const Poa4Args& A) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (;;) {
    u32 p = static_cast<u32>(sv::rfl(static_cast<int>(poa4_esc_load(A.esc))));
    p = p < A.esc_cap ? p : A.esc_cap;
    const u32 t = static_cast<u32>(sv::rfl(static_cast<int>(poa4_esc_load(A.esc + 1))));
    if (t >= p) return kEscEmpty;
    // (all 64 lanes try the same exchange: one of them gets it, or a lane of another wave did)
    u32 expected = t;
    const bool won = __hip_atomic_compare_exchange_strong(A.esc + 1, &expected, t + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!sv::any(won)) continue;
    for (;;) {  // (the pusher stores the index right behind its ticket)
      const u32 e = static_cast<u32>(sv::rfl(static_cast<int>(poa4_esc_load(A.esc + 4 + t))));
      if (e != kEscEmpty) return e;
      __builtin_amdgcn_s_sleep(8);
    }
  }
#else
  (void)A;
  return kEscEmpty;
#endif
}
// NOT inlined, and not handed the batch description either: as part of the persistent kernel's one function the 64-column code
// changed the register allocation of the NW and of the traceback (the walk's descriptor pointer went to scratch memory again:
// tests/test_poa4_isa.py), and a reference to the kernel's arguments passed to a real call moves ALL of them to the stack.  The
// callee reads the arguments where the launch left them — the kernel-argument segment, scalar loads — and gets five registers.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __attribute__((noinline)) void poa4_esc_window_far(p2::Poa2Lds<1>* Sp, u32 ka_lo, u32 ka_hi, u32 pw_, u32 wi_) {
  P4_ASSUME_LDS(Sp);
  typedef const __attribute__((address_space(4))) Poa4Args* ArgsPtr;
  // (the address of the kernel-argument segment comes from the kernel: the intrinsic is null in a function that is not one)
  const unsigned long long ka = (static_cast<unsigned long long>(static_cast<u32>(sv::rfl(static_cast<int>(ka_hi)))) << 32) |
                                static_cast<u32>(sv::rfl(static_cast<int>(ka_lo)));
  const Poa4Args A = *(ArgsPtr)ka;  // (the kernel's first parameter)
  const u32 pw = static_cast<u32>(sv::rfl(static_cast<int>(pw_))), wi = static_cast<u32>(sv::rfl(static_cast<int>(wi_)));
  const PoaWindow win = A.windows[wi];
  Poa2Slot g = poa2_carve(A.scratch + static_cast<size_t>(pw) * P4::G * A.slot_bytes, A.nmax, A.lmax, 64);
  u32 st = p2::poa2_window<1>(win, A.layers, A.src, g, A.nmax, A.lmax, A.m, A.n_, A.gp, A.trim, *Sp, A.out + win.out_off, A.out_len + wi,
                              A.phase_cycles, 0u);
  if ((st & 0xFFu) != 1u) st |= kPoaTried64;
  if (A.esc_wide && ((st & 0xFFu) == kPoaBandHit || (st & 0xFFu) == 7u)) {
    // the 128-column band right behind it (one window per C4 step: as a launch of its own, 29 ms); its LDS is the same 10 KB
    Poa2Slot g2 = poa2_carve(A.scratch + static_cast<size_t>(pw) * P4::G * A.slot_bytes, A.nmax, A.lmax, 128);
    sv::sync();
    if (sv::lane() == 0) atomicAdd(A.esc + 2, 1u);
    st = p2::poa2_window<2>(win, A.layers, A.src, g2, A.nmax, A.lmax, A.m, A.n_, A.gp, A.trim, *reinterpret_cast<p2::Poa2Lds<2>*>(Sp), A.out + win.out_off,
                            A.out_len + wi, A.phase_cycles, 0u);
    if ((st & 0xFFu) != 1u) st |= kPoaTried64 | kPoaTried128;
  }
  if (sv::lane() == 0) A.status[wi] = st;
  sv::sync();
}
#endif
__host__ __device__ __forceinline__ void poa4_esc_window(Poa4LdsAll& S, u32 pw, u32 wi) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  poa4_esc_window_far(&S.u.w64, static_cast<u32>(ka), static_cast<u32>(ka >> 32), pw, wi);
#else
  (void)S;
  (void)pw;
  (void)wi;
#endif
}
template <class K, int UP, class SC>
__host__ __device__ inline void poa4_persistent(const Poa4Args& A, const Poa4Ctx& C0, Poa4LdsAll& S, u32 pw) {
  const int lane = sv::lane();
  const u32 n_quads = (C0.count + poa4_group_windows(C0) - 1) / poa4_group_windows(C0);
  for (;;) {
    if (A.esc) {  // windows handed on by anybody's first attempt since this wave last looked
      for (;;) {
        const u32 wi = poa4_esc_pop(A);
        if (wi == kEscEmpty) break;
        poa4_esc_window(S, pw, wi);
        sv::phase_fence();
      }
    }
    // (every lane takes part in the fetch — lane 0 adds one, the others nothing: see nwpath.hip on why not `if (lane == 0)`)
    u32 quad = sv::atomic_add(A.next, lane == 0 ? 1u : 0u);
    quad = static_cast<u32>(sv::rfl(static_cast<int>(quad)));
    if (quad >= n_quads) break;
    Poa4Ctx C = C0;
    C.quad_delta = quad - pw;  // (modulo 2^32: position = first + (pw + delta) * 4 + q)
    poa4_phase_init(A, C, pw);
    sv::phase_fence();
    for (;;) {
      for (int q = 0; q < P4::G; ++q) {
        poa4_phase_graph<K, UP>(A, C, S.u.g, pw * P4::G + static_cast<u32>(q));
        sv::phase_fence();
      }
      bool any_layer = false;
      for (int q = 0; q < P4::G; ++q) {
        const u32 rec = poa4_my_record(C, pw, q);
        if (rec != 0xFFFFFFFFu) {
          const Poa4Win& w = C.st[rec];
          any_layer = any_layer || (w.phase == kRunning && w.act != 0);
          // a window that just failed with a band hit (or beyond a limit of this kernel that the 64-column function does not have):
          // into the queue right away, the rest of the group goes on
          if (A.esc && w.phase == kFailed && (w.status & 0xFFu) == kPoaBandHit) {
            const bool queued = poa4_esc_push(A, w.wi);
            if (queued && lane == 0) C.st[rec].phase = kHandedOn;
          }
        }
      }
      sv::sync();  // (the records are read by every lane before the alignment side rewrites them)
      if (!any_layer) break;
      poa4_phase_nw<K, SC>(A, C, S.u.n, pw);
      sv::phase_fence();
    }
    poa4_phase_final(A, C, S.u.f, pw);
    sv::phase_fence();
  }
}

// ---- the kernel: one wave per workgroup, resident waves = the grid ---------------------------------------------------------
// (four waves per SIMD: 128 registers, 9.7 KB of LDS.  Measured at C4 on one box: three waves per SIMD with 168 registers
// 869 ms per round, four 766 ms, five — 96 registers, a 19-row score ring = 8 KB of LDS — 844 ms: profiles/r05_poa_occupancy.txt)
template <int UP, class SC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void poa4_persistent_kernel(const Poa4Args A, const Poa4Ctx C) {
  __shared__ Poa4LdsAll lds;
  poa4_persistent<P4, UP, SC>(A, C, lds, blockIdx.x);
}
__host__ __device__ inline bool poa4_racon_scores(const Poa4Args& A) {
  return A.m == Poa4ScoresRacon::m && A.n_ == Poa4ScoresRacon::n && A.gp == Poa4ScoresRacon::g;
}

Poa4Args args_of4(const PoaBatchDev& b, unsigned char* scratch, size_t slot_bytes) {
  Poa4Args A{};
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
  A.esc = nullptr;
  A.esc_cap = 0;
  A.esc_wide = 0;
  return A;
}

}  // namespace

// One launch, persistent waves (poa4_persistent).  Resident waves = 16 per CU (four per SIMD); fewer when the batch is
// small or the scratch does not fit.
void poa_v4_launch(Engine& e, const PoaBatchDev& b) {
  if (b.n_windows == 0) return;
  const size_t slot_bytes = poa4_slot_bytes(b.nmax, b.lmax);
  int dev = 0, cus = 256;
  RVN_HIP(hipGetDevice(&dev));
  RVN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // windows per group: four, or two for a batch that leaves more than half of the wave slots empty even then (measured on
  // tools/small_batch_survey.sh: 2 500 windows 28.3 -> 24.6 ms, 5 000 windows 32.0 -> 29.9 ms; one per group is no better
  // than two, three no better than four at 10 000)
  u32 gw = b.n_windows > 2u * static_cast<u32>(cus) * 16u ? static_cast<u32>(P4::G) : 2u;
  if (const char* ev = knob("RVN_POA_GW")) gw = std::min<u32>(P4::G, std::max(1, std::atoi(ev)));  // (debug builds: A/B)
  const u32 n_quads = (b.n_windows + gw - 1) / gw;
  size_t free_b = 0, total_b = 0;
  RVN_HIP(hipMemGetInfo(&free_b, &total_b));
  const size_t per_wave = P4::G * (slot_bytes + sizeof(Poa4Win));
  // (a buffer handed back to the block pool at a stage entry is still there for the taking)
  const size_t budget = std::max(e.poa2_scratch.cap, devpool::free_largest()) + free_b / 2;
  u32 n_waves = std::min<u32>(n_quads, static_cast<u32>(cus) * 16u);
  if (static_cast<size_t>(n_waves) * per_wave + 1024 > budget) n_waves = static_cast<u32>(std::max<size_t>(1, (budget - 1024) / per_wave));
  const size_t slots = static_cast<size_t>(n_waves) * P4::G;
  unsigned char* d_scratch = e.poa2_scratch.get<unsigned char>(slots * (slot_bytes + sizeof(Poa4Win)) + 512);
  Poa4Win* d_st = reinterpret_cast<Poa4Win*>(d_scratch + slots * slot_bytes + 256);
  Poa4Args A = args_of4(b, d_scratch, slot_bytes);
  if (b.esc && poa2_slot_bytes(b.nmax, b.lmax, 64) <= P4::G * slot_bytes) {  // (the wave's four slots as the 64-column function's one)
    A.esc = b.esc;
    A.esc_cap = b.esc_cap;
    A.esc_wide = poa2_slot_bytes(b.nmax, b.lmax, 128) <= P4::G * slot_bytes ? 1u : 0u;
  }
  hipStream_t s = e.stream;
  RVN_HIP(hipMemsetAsync(b.next, 0, 4, s));
  const Poa4Ctx C{d_st, 0, b.n_windows, 0, gw};
  // (two sequence positions per lane and turn of the graph update: the kernel's 128 registers hold it without the spills the
  // four-position variant brings — 739 against 829 ms per C4 round)
  if (poa4_racon_scores(A)) {
    RVN_KLAUNCH_ON(kKPoaRows, s, (poa4_persistent_kernel<kUpdPer, Poa4ScoresRacon><<<n_waves, 64, 0, s>>>(A, C)));
  } else {
    RVN_KLAUNCH_ON(kKPoaRows, s, (poa4_persistent_kernel<kUpdPer, Poa4ScoresAny><<<n_waves, 64, 0, s>>>(A, C)));
  }
}

#ifdef RVN_TEST_HOOKS
// The same phase functions on the host, wave by wave under the wavefront emulator (simt_emu): windows / layers /
// sources are host arrays.  TEST INFRASTRUCTURE (rvn_poa_banded_emulate); first attempt only — a window that needs a
// wider band comes back flagged.
namespace {
struct EmuCall4 {
  const Poa4Args* A;
  const Poa4Ctx* C;
  Poa4Lds* S;
  Poa4LdsGraph* SG;
  Poa4LdsNw* SN;
  Poa4LdsAll* SA;
  u32 wave;
  int phase;
};
void emu_entry4(void* p) {
  EmuCall4* c = static_cast<EmuCall4*>(p);
  switch (c->phase) {
    case 0: poa4_phase_init(*c->A, *c->C, c->wave); break;
    case 1: poa4_phase_graph<P4, kUpdPer>(*c->A, *c->C, *c->SG, c->wave); break;
    case 2:
      if (poa4_racon_scores(*c->A)) poa4_phase_nw<P4, Poa4ScoresRacon>(*c->A, *c->C, *c->SN, c->wave);
      else poa4_phase_nw<P4, Poa4ScoresAny>(*c->A, *c->C, *c->SN, c->wave);
      break;
    case 9:
      if (poa4_racon_scores(*c->A)) poa4_persistent<P4, kUpdPer, Poa4ScoresRacon>(*c->A, *c->C, *c->SA, c->wave);
      else poa4_persistent<P4, kUpdPer, Poa4ScoresAny>(*c->A, *c->C, *c->SA, c->wave);
      break;
    default: poa4_phase_final(*c->A, *c->C, *c->S, c->wave); break;
  }
}
}  // namespace

void poa_v4_emulate(const std::vector<PoaWindow>& wins, const std::vector<PoaLayer>& lays, const PoaSrc& src, u32 max_bb,
                    u32 max_len, int m, int n, int g, int trim, u8* out, u32* out_len, u32* status, bool persistent) {
  if (wins.empty()) return;
  PoaBatchDev b{};
  b.lmax = std::min<u32>(kPoaMaxSeq, std::max<u32>(64, ((max_len + 63) / 64) * 64));
  b.nmax = std::min<u32>(8192, std::max<u32>(512, max_bb * 6));
  const size_t slot_bytes = poa4_slot_bytes(b.nmax, b.lmax);
  const u32 count = static_cast<u32>(wins.size());
  const u32 n_waves = (count + P4::G - 1) / P4::G;
  std::vector<unsigned char> scratch(slot_bytes * std::max<u32>(n_waves, 2) * P4::G + 256, 0);
  std::vector<Poa4Win> st(static_cast<size_t>(std::max<u32>(n_waves, 2)) * P4::G);
  unsigned long long phase[16] = {};
  u32 next = 0;
  b.wins = wins.data();
  b.n_windows = count;
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
  const Poa4Args A = args_of4(b, scratch.data(), slot_bytes);
  const Poa4Ctx C{st.data(), 0, count, 0};
  std::vector<Poa4Lds> lds(1);
  std::vector<Poa4LdsGraph> ldsg(1);
  std::vector<Poa4LdsNw> ldsn(1);
  if (persistent) {
    // two "resident" waves; the emulator runs a wave to its end, so wave 1 (started first) takes every group of windows —
    // with a group index below AND above its own — and wave 0 finds the counter exhausted
    std::vector<Poa4LdsAll> ldsa(1);
    const Poa4Ctx CP{st.data(), 0, count, 0};
    for (u32 pw : {1u, 0u}) {
      std::memset(static_cast<void*>(ldsa.data()), 0, sizeof(Poa4LdsAll));
      EmuCall4 call{&A, &CP, lds.data(), ldsg.data(), ldsn.data(), ldsa.data(), pw, 9};
      simt_emu::run_wave(&emu_entry4, &call);
    }
    return;
  }
  u32 max_layers = 0;
  for (const PoaWindow& w : wins) max_layers = std::max(max_layers, w.n_layers);
  auto run = [&](int ph) {
    const u32 waves = ph == 1 ? n_waves * P4::G : n_waves;
    for (u32 wv = 0; wv < waves; ++wv) {
      if (ph == 1 && (wv >= count || st[wv].phase != kRunning)) continue;  // (waves that would return at once: not worth 64 fibres each)
      std::memset(static_cast<void*>(lds.data()), 0, sizeof(Poa4Lds));  // (a fresh workgroup's LDS holds anything: zeros here)
      std::memset(static_cast<void*>(ldsg.data()), 0, sizeof(Poa4LdsGraph));
      std::memset(static_cast<void*>(ldsn.data()), 0, sizeof(Poa4LdsNw));
      EmuCall4 call{&A, &C, lds.data(), ldsg.data(), ldsn.data(), nullptr, wv, ph};
      simt_emu::run_wave(&emu_entry4, &call);
    }
  };
  run(0);
  for (u32 round = 1; round <= max_layers; ++round) {  // as poa_v4_launch: graph side, then (while layers remain) the alignment side
    run(1);
    if (round < max_layers) run(2);
  }
  run(5);
}

#endif  // RVN_TEST_HOOKS

}  // namespace rvn
