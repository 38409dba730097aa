// poa.h — pieces shared by the two POA window kernels (poa.hip: full-matrix kernel, the exact fallback;
// poa2.hip: banded LDS kernel, the fast path).  Both keep the spoa graph as SoA arrays in a per-slot global
// scratch with the same field names, so the graph-only steps (Subgraph marks, heaviest-bundle consensus) are
// written once here as templates over the slot type.
#pragma once

#include <vector>

#include "engine.h"
#include "simt.h"
#include "wave.h"

namespace rvn {

constexpr int kPoaMaxIn = 16;     // in-edges per node kept (overflow -> window reported as failed)
constexpr int kPoaMaxSeq = 1024;  // longest layer (bases)
constexpr i32 kNegInf16 = -30000;
constexpr int kPoa2MaxSeq = 896;  // longest layer of the banded kernels (longer ones: full-matrix kernel); sizes the LDS buffers

struct PoaWindow {  // host-prepared, one per window
  u32 layer_first, n_layers;  // range in the (begin-sorted) layer table; layer_first = backbone
  u32 out_off, out_cap;
};
// A layer is either a run of one-byte codes (rvn_poa_consensus_batch: caller-built windows) or a slice of a 2-bit
// packed read set already resident in HBM (rvn_polish_round: no host-side expansion of the reads).
enum : u32 {
  kLayerQual = 1u,      // weights = Phred - 33 from the quality array (else unit weight)
  kLayerPacked = 2u,    // bases come from a packed read set: code_off = first word of the read
  kLayerRc = 4u,        // packed: reverse complement of the read
  kLayerTarget = 8u,    // packed: the read set is the target set (backbones)
  kLayerZeroW = 16u,    // weight 0 (racon's dummy '!' backbone quality)
};
struct PoaLayer {
  u64 code_off;  // bytes: offset into codes | packed: word offset of the read
  u64 qual_off;  // offset of the layer (bytes) / of the read (packed) in the quality array
  u32 len, begin, end, flags;
  u32 q_begin, q_len;  // packed: first base of the layer in the (oriented) read, length of the read
  // Band guide of the banded kernel: layer offsets that backbone positions begin + i * span / 8 (i = 1..7,
  // span = end - begin + 1) map to.  A straight line (i * len / 8) when nothing better is known; the polishing front
  // end fills it from the chain anchors, so the band follows the read through local indel bursts.
  u16 way[7];
  u16 pad_;
};
inline void poa_layer_linear_way(PoaLayer& L) {
  for (u32 i = 1; i < 8; ++i) L.way[i - 1] = static_cast<u16>(static_cast<u64>(L.len) * i / 8);
  L.pad_ = 0;
}
// expected layer offset of backbone position begin + x (x clamped to [0, span]), piecewise linear through `way`
__host__ __device__ __forceinline__ i32 poa_layer_center(const PoaLayer& L, i32 x, i32 span) {
  x = x < 0 ? 0 : (x > span ? span : x);
  const i32 seg = (x * 8) / (span > 0 ? span : 1);      // 0..8
  const i32 s = seg > 7 ? 7 : seg;
  const i32 x0 = (s * span) / 8, x1 = ((s + 1) * span) / 8;
  const i32 w0 = s == 0 ? 0 : static_cast<i32>(L.way[s - 1]);
  const i32 w1 = s == 7 ? static_cast<i32>(L.len) : static_cast<i32>(L.way[s]);
  return w0 + (x - x0) * (w1 - w0) / (x1 > x0 ? x1 - x0 : 1);
}
struct PoaSrc {
  const u8* codes;
  const u8* quals;
  const u64* packed_reads;
  const u64* packed_targets;
  const u8* read_quals;    // Phred+33 of the packed reads, one byte per 2^qual_shift bases (nullable)
  u32 qual_shift;          // 0: per base; 6: biosoup's block qualities (mean of 64 bases)
  const u8* layer_ok;      // per layer: 0 = dropped by the mean-quality filter (nullable = all kept)
};

__host__ __device__ __forceinline__ u32 poa_layer_src_pos(const PoaLayer& L, u32 i) {
  const u32 pos = L.q_begin + i;
  return (L.flags & kLayerRc) ? L.q_len - 1 - pos : pos;
}
__host__ __device__ __forceinline__ u32 poa_layer_code(const PoaSrc& src, const PoaLayer& L, u32 i) {
  if (L.flags & kLayerPacked) {
    const u64* P = (L.flags & kLayerTarget) ? src.packed_targets : src.packed_reads;
    const u32 sp = poa_layer_src_pos(L, i);
    const u32 c = static_cast<u32>(P[L.code_off + (sp >> 5)] >> ((sp & 31) << 1)) & 3u;
    return (L.flags & kLayerRc) ? 3u - c : c;
  }
  return src.codes[L.code_off + i];
}
__host__ __device__ __forceinline__ i32 poa_layer_weight(const PoaSrc& src, const PoaLayer& L, u32 i) {
  if (L.flags & kLayerZeroW) return 0;
  if (!(L.flags & kLayerQual)) return 1;
  if (L.flags & kLayerPacked)
    return static_cast<i32>(src.read_quals[L.qual_off + (poa_layer_src_pos(L, i) >> src.qual_shift)]) - 33;
  return static_cast<i32>(src.quals[L.qual_off + i]) - 33;
}

// window status: 0 backbone returned (< 3 sequences), 1 polished, 2 node limit, 3 in-degree limit, 4 length limit,
// 5..7 internal (| layer << 8), 8 alignment left the band (banded kernel only; such windows are re-run by the
// full-matrix kernel)
constexpr u32 kPoaBandHit = 8;

struct PoaBatchDev {  // device-side batch description shared by both launchers
  const PoaWindow* wins;
  u32 n_windows;
  const PoaLayer* layers;
  PoaSrc src;
  u32 nmax, lmax;
  int m, n, g, trim;
  u8* out;
  u32* out_len;
  u32* status;
  unsigned long long* phase_cycles;
  const u32* sched;  // window order for the persistent waves (heaviest first), or null
  u32* next;         // work counter (zeroed by the launcher)
  u32 probe;         // diagnostics (RVN_POA_BAND_PROBE): a polished window's status carries, << 16, how far its alignment
                     // paths strayed from the band's centre (0..31 columns) — what a narrower band would have to hold
  u32* esc;          // poa4.hip only: queue of the windows its 32-column attempt hands on to the 64-column window function
                     // INSIDE the same launch ([0] pushed, [1] taken, [2] went on to 128 columns, [4 + k] window index; set up by poa_run_dev), or null
  u32 esc_cap;       // entries of the queue
  u32 esc_wide;      // != 0: a window the 64 columns cannot hold goes through the 128-column function right there (set by poa_v4_launch)
};
// Smallest batch of windows that starts with poa4.hip's rows-on-lanes kernel (engine option poa_rows_min_windows; smaller ones
// start with poa2.hip's one-window-per-wave kernel).  Round 5: 20 000 — what the first attempt handed on cost a second launch of
// one window's latency, whatever the count.  Round 6: those windows are done inside the first launch (poa4_esc_*), and what
// is left of the threshold is the first attempt's own latency on a nearly empty machine: C2's rounds of 10 000 windows 63 ms
// against 67, 10 000 windows of tools/bench_poa.py (a tenth of them handed on: no band guides) 58.6 against 58.4, 5 000 of
// those 47 against 33.
constexpr u32 kPoaRowsMinWindowsDefault = 8192;
constexpr u32 kPoaTried64 = 0x10000000u;   // status bit of a window the 64-column function already had inside poa4.hip's launch
constexpr u32 kPoaTried128 = 0x20000000u;  // ... and the 128-column function behind it

// Per-window scratch of the banded kernels (poa2.hip; poa4.hip adds its own arrays behind it): the spoa graph as SoA arrays, the backpointer matrix of
// the current layer and the traceback's row table, carved out of one allocation per resident window.
struct Poa2Slot {
  i16* Hs;    // (nmax + 1) x band scores (ring misses only)
  u8* BP;     // (nmax + 1) x band backpointers: 0..15 diagonal via in-edge k, 16..31 vertical, 32 horizontal
  uint4* tb;  // per row: x = band start | node << 16, y = #in-edges, z = rows of in-edges 0,1, w = in-edges 2,3
  u8* code;
  u8* in_cnt;
  u16* in_tail;
  i32* in_w;
  u16* out_cnt;
  u8* al_cnt;
  u16* al;
  u16* visits;
  u16* rank_of;
  u16* order;
  u16* order2;
  u8* mark;
  u16* sub_out;
  u16* bpos;
  u16* new_slot;
  i32* scores;
  i32* preds;
  u16* stack;
  u16* pos_node;  // traceback result of the current layer: node aligned to position p, or kNone
  u16* tgt;       // AddAlignment: graph node of every sequence position (unused: poa2.hip keeps it in LDS)
};

template <class F>
__host__ __device__ inline void poa2_fields(u32 nmax, u32 lmax, u32 band, F&& f, bool hs = true) {
  f(0, hs ? static_cast<size_t>(nmax + 1) * band * 2 : 0);
  f(1, static_cast<size_t>(nmax + 1) * band);
  f(2, static_cast<size_t>(nmax + 1) * 16);
  f(3, nmax);
  f(4, nmax);
  f(5, static_cast<size_t>(nmax) * kPoaMaxIn * 2);
  f(6, static_cast<size_t>(nmax) * kPoaMaxIn * 4);
  f(7, static_cast<size_t>(nmax) * 2);
  f(8, nmax);
  f(9, static_cast<size_t>(nmax) * 4 * 2);
  f(10, static_cast<size_t>(nmax) * 2);
  f(11, static_cast<size_t>(nmax) * 2);
  f(12, static_cast<size_t>(nmax) * 2);
  f(13, static_cast<size_t>(nmax) * 2);
  f(14, nmax);
  f(15, static_cast<size_t>(nmax) * 2 + 4);
  f(16, static_cast<size_t>(nmax) * 2);
  f(17, static_cast<size_t>(lmax + 2) * 2);
  f(18, static_cast<size_t>(nmax) * 4);
  f(19, static_cast<size_t>(nmax) * 4);
  f(20, static_cast<size_t>(nmax) * 2);
  f(21, static_cast<size_t>(lmax + 8) * 2);
  f(22, static_cast<size_t>(lmax + 8) * 2);
}

inline size_t poa2_slot_bytes(u32 nmax, u32 lmax, u32 band, bool hs = true) {
  size_t b = 0;
  poa2_fields(nmax, lmax, band, [&](int, size_t x) { b += (x + 255) & ~size_t(255); }, hs);
  return b;
}

__host__ __device__ inline Poa2Slot poa2_carve(unsigned char* base, u32 nmax, u32 lmax, u32 band, bool hs = true) {
  unsigned char* p[23];
  size_t o = 0;
  poa2_fields(nmax, lmax, band, [&](int i, size_t x) {
    p[i] = base + o;
    o += (x + 255) & ~size_t(255);
  }, hs);
  Poa2Slot s;
  s.Hs = reinterpret_cast<i16*>(p[0]);
  s.BP = p[1];
  s.tb = reinterpret_cast<uint4*>(p[2]);
  s.code = p[3];
  s.in_cnt = p[4];
  s.in_tail = reinterpret_cast<u16*>(p[5]);
  s.in_w = reinterpret_cast<i32*>(p[6]);
  s.out_cnt = reinterpret_cast<u16*>(p[7]);
  s.al_cnt = p[8];
  s.al = reinterpret_cast<u16*>(p[9]);
  s.visits = reinterpret_cast<u16*>(p[10]);
  s.rank_of = reinterpret_cast<u16*>(p[11]);
  s.order = reinterpret_cast<u16*>(p[12]);
  s.order2 = reinterpret_cast<u16*>(p[13]);
  s.mark = p[14];
  s.sub_out = reinterpret_cast<u16*>(p[15]);
  s.bpos = reinterpret_cast<u16*>(p[16]);
  s.new_slot = reinterpret_cast<u16*>(p[17]);
  s.scores = reinterpret_cast<i32*>(p[18]);
  s.preds = reinterpret_cast<i32*>(p[19]);
  s.stack = reinterpret_cast<u16*>(p[20]);
  s.pos_node = reinterpret_cast<u16*>(p[21]);
  s.tgt = reinterpret_cast<u16*>(p[22]);
  return s;
}

void poa_v1_launch(Engine& e, const PoaBatchDev& b);  // poa.hip
// windows + begin-sorted layer descriptors on the host, all sources of `src` resident in HBM (poa.hip)
void poa_run(Engine& e, const std::vector<PoaWindow>& wins, const std::vector<PoaLayer>& lays, const PoaSrc& src,
             u32 max_bb, u32 max_len, int m, int n, int g, int trim, u8* h_out, u64 out_total, u32* h_out_len,
             u32* h_status, double* device_ms, bool allow_full = true);  // allow_full: escalate to the full-matrix kernel
// the same with windows, layers and outputs resident in HBM (polish.hip); h_status receives the per-window status
void poa_run_dev(Engine& e, const PoaWindow* d_wins, const PoaLayer* d_lays, u32 n_windows, const PoaSrc& src, u32 max_bb,
                 u32 max_len, int m, int n, int g, int trim, u8* d_out, u32* d_len, u32* d_status,
                 std::vector<u32>& h_status, double* device_ms, bool allow_full = true);
void poa_v2_launch(Engine& e, const PoaBatchDev& b, int nch);  // poa2.hip: band = 64 * nch columns
// poa4.hip: rows on lanes, four windows per wave, 32-column band (the first attempt of the default mode)
void poa_v4_launch(Engine& e, const PoaBatchDev& b);
void poa_v4_emulate(const std::vector<PoaWindow>& wins, const std::vector<PoaLayer>& lays, const PoaSrc& src, u32 max_bb,
                    u32 max_len, int m, int n, int g, int trim, u8* out, u32* out_len, u32* status, bool persistent = false);

// Persistent waves take windows from a shared counter (longest-processing-time-first order when `sched` is given).
__device__ __forceinline__ u32 poa_next_window(u32* next, const u32* sched, u32 n_windows) {
  u32 i = 0;
  if (lane_id() == 0) i = atomicAdd(next, 1u);
  i = static_cast<u32>(__builtin_amdgcn_readfirstlane(static_cast<int>(i)));
  if (i >= n_windows) return 0xFFFFFFFFu;
  return sched ? sched[i] : i;
}

__host__ __device__ __forceinline__ void wsync() { sv::sync(); }

// spoa Graph::AddEdge on the SoA graph. Returns false on in-degree overflow.  Touches only head's in-edge list
// and tail's out-degree, so lanes working on distinct (tail, head) pairs do not conflict.
template <class G>
__host__ __device__ inline bool poa_add_edge(G& g, u32 tail, u32 head, i32 weight) {
  const u32 c = g.in_cnt[head];
  for (u32 i = 0; i < c; ++i) {
    if (g.in_tail[head * kPoaMaxIn + i] == tail) {
      g.in_w[head * kPoaMaxIn + i] += weight;
      return true;
    }
  }
  if (c >= kPoaMaxIn) return false;
  g.in_tail[head * kPoaMaxIn + c] = static_cast<u16>(tail);
  g.in_w[head * kPoaMaxIn + c] = weight;
  g.in_cnt[head] = static_cast<u8>(c + 1);
  g.out_cnt[tail] += 1;
  return true;
}

// spoa Graph::Subgraph as marks: ancestors (through in-edges and aligned nodes) of backbone node `end` with
// id >= begin; sub_out = out-degree inside the subgraph.  Whole wave; lane 0 runs the DFS.
template <class G>
__host__ __device__ inline void poa_subgraph_marks(G& g, u32 n_nodes, u32 nmax, u32 begin, u32 end) {
  const int lane = sv::lane();
  for (u32 i = lane; i < n_nodes; i += 64) {
    g.mark[i] = 0;
    g.sub_out[i] = 0;
  }
  wsync();
  if (lane == 0) {
    u32 sp = 0;
    g.stack[sp++] = static_cast<u16>(end);
    while (sp) {
      const u32 curr = g.stack[--sp];
      if (!g.mark[curr] && curr >= begin) {
        const u32 c = g.in_cnt[curr];
        for (u32 k = 0; k < c && sp < nmax; ++k) g.stack[sp++] = g.in_tail[curr * kPoaMaxIn + k];
        const u32 a = g.al_cnt[curr];
        for (u32 k = 0; k < a && sp < nmax; ++k) g.stack[sp++] = g.al[curr * 4 + k];
        g.mark[curr] = 1;
      }
    }
  }
  wsync();
  for (u32 v = lane; v < n_nodes; v += 64) {
    if (!g.mark[v]) continue;
    const u32 c = g.in_cnt[v];
    for (u32 k = 0; k < c; ++k) {
      const u32 t = g.in_tail[v * kPoaMaxIn + k];
      if (g.mark[t]) sv::atomic_add(reinterpret_cast<unsigned int*>(g.sub_out) + (t >> 1), (t & 1) ? 0x10000u : 1u);
    }
  }
  wsync();
}

// Consensus: spoa TraverseHeaviestBundle + BranchCompletion, racon's coverage trim (lane 0).
// Part 1: heaviest-path scores and predecessors of every node in topological order; returns the best-scoring node.
template <class G>
__host__ __device__ inline i32 poa_consensus_scores_lane0(G& g, u32 n_nodes) {
  i32 maxn = -1;
  for (u32 r = 0; r < n_nodes; ++r) {
    const u32 it = g.order[r];
    // (the predecessor's score travels with it: with `g.scores[pd] <= g.scores[t]` in the condition, hipcc 7.2 compiled the
    // lane-0 scalarised loop of the branch completion below so that a TIE updated sc but not pd — found on one window in
    // 20 000 whose consensus ended two bases early on the device and not under the host emulator)
    i32 sc = -1, pd = -1, pd_sc = 0;
    const u32 c = g.in_cnt[it];
    for (u32 k = 0; k < c; ++k) {
      const i32 wgt = g.in_w[it * kPoaMaxIn + k];
      const i32 t = g.in_tail[it * kPoaMaxIn + k];
      const i32 st = g.scores[t];
      if (sc < wgt || (sc == wgt && pd_sc <= st)) {
        sc = wgt;
        pd = t;
        pd_sc = st;
      }
    }
    if (pd != -1) sc += pd_sc;
    g.scores[it] = sc;
    g.preds[it] = pd;
    if (maxn == -1 || g.scores[maxn] < sc) maxn = static_cast<i32>(it);
  }
  return maxn;
}

// Part 2: branch completion from `maxn`, traceback into g.stack (reverse order), racon's coverage trim.
// Consensus position p (begin <= p <= end) is node g.stack[cl - 1 - p].
template <class G>
__host__ __device__ inline void poa_consensus_trace_lane0(G& g, u32 n_nodes, u32 nmax, const PoaWindow& win, int trim, i32 maxn,
                                                 u32* cl_out, i32* begin_out, i32* end_out) {
  u32 guard = 0;
  while (g.out_cnt[maxn] != 0 && guard++ < nmax) {
    // BranchCompletion(rank of maxn)
    const u32 start = static_cast<u32>(maxn);
    const u32 rank = g.rank_of[start];
    for (u32 r = 0; r < n_nodes; ++r) {  // heads of start's out-edges: other tails lose their score
      const u32 hd = g.order[r];
      const u32 c = g.in_cnt[hd];
      bool from_start = false;
      for (u32 k = 0; k < c; ++k) from_start |= g.in_tail[hd * kPoaMaxIn + k] == start;
      if (!from_start) continue;
      for (u32 k = 0; k < c; ++k) {
        const u32 t = g.in_tail[hd * kPoaMaxIn + k];
        if (t != start) g.scores[t] = -1;
      }
    }
    i32 mx = -1;
    for (u32 r = rank + 1; r < n_nodes; ++r) {
      const u32 it = g.order[r];
      i32 sc = -1, pd = -1, pd_sc = 0;
      const u32 c = g.in_cnt[it];
      for (u32 k = 0; k < c; ++k) {
        const i32 t = g.in_tail[it * kPoaMaxIn + k];
        const i32 st = g.scores[t];
        if (st == -1) continue;
        const i32 wgt = g.in_w[it * kPoaMaxIn + k];
        if (sc < wgt || (sc == wgt && pd_sc <= st)) {
          sc = wgt;
          pd = t;
          pd_sc = st;
        }
      }
      if (pd != -1) sc += pd_sc;
      g.scores[it] = sc;
      g.preds[it] = pd;
      if (mx == -1 || g.scores[mx] < sc) mx = static_cast<i32>(it);
    }
    if (mx == -1) break;
    maxn = mx;
  }
  // traceback into stack (reverse), then forward with coverage + trim
  u32 cl = 0;
  i32 cur = maxn;
  while (cur != -1 && cl < nmax) {
    g.stack[cl++] = static_cast<u16>(cur);
    cur = g.preds[cur];
  }
  // coverage of consensus node = visits of the node + its aligned nodes (spoa Node::Coverage summed, racon)
  i32 begin = 0, end = static_cast<i32>(cl) - 1;
  if (trim) {
    const u32 avg = (win.n_layers - 1) / 2;
    auto cov = [&](i32 pos) -> u32 {  // pos in forward consensus coordinates
      const u32 v = g.stack[cl - 1 - pos];
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
  *cl_out = cl;
  *begin_out = begin;
  *end_out = end;
}

template <class G>
__host__ __device__ inline void poa_consensus_lane0(G& g, u32 n_nodes, u32 nmax, const PoaWindow& win, int trim,
                                           u8* __restrict__ out, u32* out_len) {
  const i32 maxn = poa_consensus_scores_lane0(g, n_nodes);
  u32 cl = 0;
  i32 begin = 0, end = -1;
  poa_consensus_trace_lane0(g, n_nodes, nmax, win, trim, maxn, &cl, &begin, &end);
  u32 cons_len = 0;
  for (i32 p = begin; p <= end && cons_len < win.out_cap; ++p) out[cons_len++] = g.code[g.stack[cl - 1 - p]];
  *out_len = cons_len;
}

}  // namespace rvn
