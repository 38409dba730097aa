// polish.hip — one racon polishing round on the device (racon::Polisher::Polish as driven by raven::Polish,
// RavenLib/src/polish.cc:43-51 with e = 0.3, w = 500, trim = true, AlignCfg m/n/g):
//   1. index the targets (unitigs), Filter(0.001), Map every read (avoid_equal = avoid_symmetric = false) — the
//      same device engine as the overlap phase, with the chain ANCHORS of every overlap kept;
//   2. keep each read's longest overlap, drop it when 1 - min(span)/max(span) > e;
//   3. cut the read into window layers.  racon takes the breakpoints from an edlib NW path (CIGAR); here they
//      come from the chain anchors (exact k-mer matches the alignment path passes through): the read position at
//      a window boundary is exact when the boundary falls inside an anchor's k-mer and interpolated linearly
//      between the bracketing anchors otherwise, so the layers tile the windows like racon's do.  Layers shorter
//      than 0.02 w, or (with qualities) below the mean-quality threshold q, are dropped exactly as in racon;
//   4. window consensus = the POA kernel (poa.hip); 5. stitch the windows, polished ratio per target.
// Step 3 is a deliberate, documented deviation from racon (no per-base path alignment on the device yet); the
// consensus is compared with the CPU restatement of racon's own pipeline within tolerance.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include "poa.h"

namespace rvn {

namespace {

// racon: a layer is used only if the mean Phred of its bases reaches q.  One wave per layer.
__global__ __launch_bounds__(256) void layer_quality_kernel(const PoaLayer* __restrict__ layers, u32 n_layers,
                                                           const u8* __restrict__ read_quals, double q_thr,
                                                           u8* __restrict__ ok) {
  const u32 li = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (li >= n_layers) return;
  const PoaLayer L = layers[li];
  const int lane = lane_id();
  if (!(L.flags & kLayerQual) || (L.flags & kLayerTarget) || L.len == 0) {
    if (lane == 0) ok[li] = 1;
    return;
  }
  u32 sum = 0;
  for (u32 i = lane; i < L.len; i += 64) sum += static_cast<u32>(read_quals[L.qual_off + poa_layer_src_pos(L, i)]) - 33u;
  sum = wave_sum(sum);
  if (lane == 0) ok[li] = static_cast<double>(sum) / static_cast<double>(L.len) >= q_thr ? 1 : 0;
}

struct BestOverlap {
  bool valid = false;
  Overlap o{};
  u64 aoff = 0;
  u32 acnt = 0;
};

}  // namespace

void polish_round(Engine& e, ReadsDev& T, ReadsDev& R, const u8* h_quals, const u64* h_qual_off, double q_thr,
                  double err_thr, u32 w, bool trim, int m, int n, int g, std::vector<std::vector<u8>>& polished,
                  std::vector<double>& ratio, PolishStats& stats) {
  hipStream_t s = e.stream;
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
  const auto t_all = clk::now();
  polished.assign(T.n, {});
  ratio.assign(T.n, 0.0);
  stats = PolishStats();
  if (T.n == 0) return;

  // ---- 1. map reads to targets --------------------------------------------------------------------
  {
    StageTimer t(e, StageTimes::kSketch);
    e.query_ready = false;
    e.index.has_query_flags = false;
    e.index.all_query = false;
    sketch_raw(e, T, 0, T.n, e.index_sketch);
    t.stop();
  }
  index_build(e, e.index_sketch, true);
  index_filter(e, 0.001);
  const bool keep = e.keep_anchors;
  e.keep_anchors = true;
  MapOut& mo = e.map_out;
  try {
    map_batch(e, R, 0, R.n, false, false, false, false, mo);
  } catch (...) {
    e.keep_anchors = keep;
    throw;
  }
  e.keep_anchors = keep;
  RVN_HIP(hipStreamSynchronize(s));
  const u64 O = mo.n_overlaps;
  std::vector<Overlap> ovl(O);
  std::vector<u32> roff(static_cast<size_t>(R.n) + 1, 0);
  std::vector<u64> aoff(O);
  std::vector<u32> acnt(O);
  std::vector<u64> anchors(mo.n_matches);
  if (O) {
    RVN_HIP(hipMemcpy(ovl.data(), mo.ovl.ptr, O * sizeof(Overlap), hipMemcpyDeviceToHost));
    RVN_HIP(hipMemcpy(aoff.data(), mo.anchor_off.ptr, O * 8, hipMemcpyDeviceToHost));
    RVN_HIP(hipMemcpy(acnt.data(), mo.anchor_cnt.ptr, O * 4, hipMemcpyDeviceToHost));
    RVN_HIP(hipMemcpy(anchors.data(), mo.anchors.ptr, mo.n_matches * 8, hipMemcpyDeviceToHost));
  }
  RVN_HIP(hipMemcpy(roff.data(), mo.ovl_read_off.ptr, roff.size() * 4, hipMemcpyDeviceToHost));
  stats.n_overlaps = O;
  stats.map_ms = ms_since(t_all);
  const auto t_host = clk::now();

  // ---- 2. best overlap per read ------------------------------------------------------------------------
  auto span_len = [](const Overlap& o) {
    return std::max(o.lhs_end - o.lhs_begin, o.rhs_end - o.rhs_begin);
  };
  std::vector<BestOverlap> best(R.n);
  for (u32 r = 0; r < R.n; ++r) {
    for (u32 i = roff[r]; i < roff[r + 1]; ++i) {
      if (!best[r].valid || span_len(best[r].o) < span_len(ovl[i])) {
        best[r].valid = true;
        best[r].o = ovl[i];
        best[r].aoff = aoff[i];
        best[r].acnt = acnt[i];
      }
    }
    if (best[r].valid) {
      const Overlap& o = best[r].o;
      const double a = o.lhs_end - o.lhs_begin, b = o.rhs_end - o.rhs_begin;
      const double err = 1.0 - std::min(a, b) / std::max(a, b);
      if (err > err_thr) best[r].valid = false;
    }
  }

  // ---- 3. windows and layers ----------------------------------------------------------------------------
  // target id -> index in T (ids are arbitrary); windows are numbered target by target
  std::vector<u32> id_to_t;
  {
    u32 max_id = 0;
    for (u32 t = 0; t < T.n; ++t) max_id = std::max(max_id, T.h_id[t]);
    id_to_t.assign(static_cast<size_t>(max_id) + 1, 0xFFFFFFFFu);
    for (u32 t = 0; t < T.n; ++t) id_to_t[T.h_id[t]] = t;
  }
  std::vector<u64> first_window(static_cast<size_t>(T.n) + 1, 0);
  for (u32 t = 0; t < T.n; ++t) first_window[t + 1] = first_window[t] + (static_cast<u64>(T.h_len[t]) + w - 1) / w;
  const u64 n_windows = first_window[T.n];
  struct LayerRef {
    u32 read, q_begin, q_len, t_begin, t_end, rc;
  };
  std::vector<std::vector<LayerRef>> win_layers(n_windows);
  const u32 k = e.k;
  e.polish_target_reads.assign(T.n, 0);
  for (u32 r = 0; r < R.n; ++r) {
    if (!best[r].valid) continue;
    const Overlap& o = best[r].o;
    if (o.rhs_id >= id_to_t.size() || id_to_t[o.rhs_id] == 0xFFFFFFFFu) continue;
    const u32 t = id_to_t[o.rhs_id];
    const u32 qlen = R.h_len[r];
    const bool rc = o.strand == 0;
    ++stats.n_reads_used;
    ++e.polish_target_reads[t];
    // anchors as (t, q') increasing in both; q' in the orientation that matches the target
    std::vector<std::pair<u32, u32>> an(best[r].acnt);
    for (u32 i = 0; i < best[r].acnt; ++i) {
      const u64 a = anchors[best[r].aoff + i];
      const u32 qp = static_cast<u32>(a >> 32), tp = static_cast<u32>(a);
      an[i] = rc ? std::make_pair(tp, qlen - qp - k) : std::make_pair(tp, qp);
    }
    if (rc) std::reverse(an.begin(), an.end());
    if (an.size() < 2) continue;
    // read position at target coordinate B (a window boundary inside the chain): exact inside an anchor's k-mer,
    // linear between the end of the anchor before and the start of the anchor after otherwise
    auto q_at = [&](u32 B) -> u32 {
      size_t lo = 0, hi = an.size();  // last anchor with t <= B
      while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (an[mid].first <= B) lo = mid;
        else hi = mid;
      }
      const u32 ta = an[lo].first, qa = an[lo].second;
      if (B < ta + k || lo + 1 >= an.size()) return qa + (B - ta);
      const u32 tc = an[lo + 1].first, qc = an[lo + 1].second;
      if (tc <= ta + k || qc <= qa + k) return qa + k;
      const double f = static_cast<double>(B - ta - k) / static_cast<double>(tc - ta - k);
      return qa + k + static_cast<u32>(f * static_cast<double>(qc - qa - k) + 0.5);
    };
    const u32 t_first = an.front().first, t_last_end = an.back().first + k;  // chain covers [t_first, t_last_end)
    const u32 q_first = an.front().second, q_last_end = an.back().second + k;
    for (u32 wi = t_first / w; static_cast<u64>(wi) * w < t_last_end; ++wi) {
      const u32 ws = wi * w;
      const u32 we = std::min<u32>(T.h_len[t], ws + w);  // exclusive
      const u32 t_b = std::max(ws, t_first), t_e = std::min(we, t_last_end);  // [t_b, t_e)
      if (t_e <= t_b + 1) continue;
      const u32 q_b = t_b == t_first ? q_first : q_at(t_b);
      u32 q_e = t_e == t_last_end ? q_last_end : q_at(t_e);
      if (q_e > qlen) q_e = qlen;
      if (q_e <= q_b || (q_e - q_b) < 0.02 * w) continue;
      {  // interpolated breakpoints can be off where the chain has a long anchor-free stretch across a window
         // boundary; a piece whose length disagrees with its target span by more than any plausible indel
         // imbalance is misplaced, and racon's exact breakpoints would never have produced it: drop it
        const double span = t_e - t_b, ql = q_e - q_b;
        if (std::abs(ql - span) > std::max(16.0, 0.08 * span)) {
          ++stats.n_dropped_layers;
          continue;
        }
      }
      win_layers[first_window[t] + wi].push_back(LayerRef{r, q_b, q_e - q_b, t_b - ws, t_e - 1 - ws, rc});
    }
  }

  // ---- 4. layer descriptors for the POA batch: bases and qualities stay in HBM (packed read sets) ---------------
  std::vector<PoaWindow> wins(n_windows);
  std::vector<PoaLayer> lays;
  std::vector<u64> out_off{0};
  lays.reserve(n_windows * 32);
  const bool any_q = h_quals != nullptr;
  u32 max_bb = 1, max_len = 1;
  for (u32 t = 0; t < T.n; ++t) {
    const u32 tlen = T.h_len[t];
    for (u64 wi = 0; wi < first_window[t + 1] - first_window[t]; ++wi) {
      const u64 gw = first_window[t] + wi;
      const u32 ws = static_cast<u32>(wi) * w;
      const u32 bl = std::min<u32>(w, tlen - ws);
      wins[gw].layer_first = static_cast<u32>(lays.size());
      PoaLayer B{};
      B.code_off = T.h_word_off[t];
      B.len = bl;
      B.begin = 0;
      B.end = bl ? bl - 1 : 0;
      B.flags = kLayerPacked | kLayerTarget | kLayerZeroW;  // weight 0 = racon's dummy '!' backbone quality
      B.q_begin = ws;
      B.q_len = tlen;
      lays.push_back(B);
      max_bb = std::max(max_bb, bl);
      auto& wl = win_layers[gw];
      // racon: layers in stable order of their begin position
      std::stable_sort(wl.begin(), wl.end(), [](const LayerRef& a, const LayerRef& b) { return a.t_begin < b.t_begin; });
      for (const auto& L : wl) {
        PoaLayer P{};
        P.code_off = R.h_word_off[L.read];
        P.qual_off = any_q ? h_qual_off[L.read] : 0;
        P.len = L.q_len;
        P.begin = L.t_begin;
        P.end = std::min(L.t_end, bl - 1);
        P.flags = kLayerPacked | (L.rc ? kLayerRc : 0u) | (any_q ? kLayerQual : 0u);
        P.q_begin = L.q_begin;
        P.q_len = R.h_len[L.read];
        lays.push_back(P);
        max_len = std::max(max_len, L.q_len);
        ++stats.n_layers;
      }
      wins[gw].n_layers = static_cast<u32>(lays.size()) - wins[gw].layer_first;
      wins[gw].out_off = static_cast<u32>(out_off.back());
      wins[gw].out_cap = 4 * bl + 256;
      out_off.push_back(out_off.back() + 4ULL * bl + 256);
    }
  }
  max_len = std::max(max_len, max_bb);
  PoaSrc src{};
  src.packed_reads = R.packed.as<u64>();
  src.packed_targets = T.packed.as<u64>();
  if (any_q) {
    // qualities to HBM once per call; racon's mean-quality filter as per-layer flags computed on the device
    const u64 qtotal = h_qual_off[R.n];
    u8* d_q = e.polish_quals.get<u8>(qtotal + 16);
    RVN_HIP(hipMemcpyAsync(d_q, h_quals, qtotal, hipMemcpyHostToDevice, s));
    PoaLayer* d_l = e.tmp_d.get<PoaLayer>(lays.size() + 1);
    RVN_HIP(hipMemcpyAsync(d_l, lays.data(), lays.size() * sizeof(PoaLayer), hipMemcpyHostToDevice, s));
    u8* d_ok = e.tmp_a.get<u8>(lays.size() + 16);
    const u32 nl = static_cast<u32>(lays.size());
    layer_quality_kernel<<<(nl + 3) / 4, 256, 0, s>>>(d_l, nl, d_q, q_thr, d_ok);
    RVN_HIP(hipGetLastError());
    src.read_quals = d_q;
    src.layer_ok = d_ok;
    if (q_thr > 0) {  // only for the statistics: how many layers survive
      std::vector<u8> okh(nl);
      RVN_HIP(hipMemcpyAsync(okh.data(), d_ok, nl, hipMemcpyDeviceToHost, s));
      RVN_HIP(hipStreamSynchronize(s));
      u64 kept = 0;
      for (u32 i = 0; i < nl; ++i) kept += (okh[i] && !(lays[i].flags & kLayerTarget)) ? 1 : 0;
      stats.n_layers = kept;
    }
  }
  stats.n_windows = n_windows;
  std::vector<u8> cons(out_off.back() + 16);
  std::vector<u32> cons_len(n_windows), status(n_windows);
  double ms = 0;
  stats.host_ms = ms_since(t_host);
  const auto t_poa = clk::now();
  poa_run(e, wins, lays, src, max_bb, max_len, m, n, g, trim ? 1 : 0, cons.data(), out_off.back(), cons_len.data(),
          status.data(), &ms);
  stats.poa_ms = ms;
  const double poa_wall = ms_since(t_poa);
  const auto t_st = clk::now();

  // ---- 5. stitch -------------------------------------------------------------------------------------------
  for (u32 t = 0; t < T.n; ++t) {
    u64 polished_windows = 0;
    const u64 nw = first_window[t + 1] - first_window[t];
    for (u64 wi = 0; wi < nw; ++wi) {
      const u64 gw = first_window[t] + wi;
      polished_windows += status[gw] == 1 ? 1 : 0;
      if (status[gw] >= 2) ++stats.n_failed_windows;
      polished[t].insert(polished[t].end(), cons.begin() + out_off[gw], cons.begin() + out_off[gw] + cons_len[gw]);
    }
    ratio[t] = nw ? static_cast<double>(polished_windows) / nw : 0.0;
    stats.n_polished_windows += polished_windows;
  }
  stats.host_ms += ms_since(t_st) + (poa_wall - ms);  // stitching + the batch's host-side preparation and copies
  stats.total_ms = ms_since(t_all);
}

}  // namespace rvn
