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
#include <vector>

#include "engine.h"

namespace rvn {

namespace {

inline u8 code_at(const std::vector<u64>& packed, u64 word_off, u32 i) {
  return static_cast<u8>((packed[word_off + (i >> 5)] >> ((i << 1) & 63)) & 3);
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
  polished.assign(T.n, {});
  ratio.assign(T.n, 0.0);
  stats = PolishStats();
  if (T.n == 0) return;
  if (T.h_packed.empty() || (R.n && R.h_packed.empty()))
    throw std::invalid_argument("[raven_hip] polish needs read sets uploaded with host copies (rvn_reads_upload)");

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
  for (u32 r = 0; r < R.n; ++r) {
    if (!best[r].valid) continue;
    const Overlap& o = best[r].o;
    if (o.rhs_id >= id_to_t.size() || id_to_t[o.rhs_id] == 0xFFFFFFFFu) continue;
    const u32 t = id_to_t[o.rhs_id];
    const u32 qlen = R.h_len[r];
    const bool rc = o.strand == 0;
    ++stats.n_reads_used;
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
      bool ok = true;
      if (h_quals) {  // racon: mean quality of the layer must reach q
        double sum = 0;
        const u64 qb = h_qual_off[r];
        for (u32 x = q_b; x < q_e; ++x) {
          const u32 src = rc ? qlen - 1 - x : x;
          sum += static_cast<double>(h_quals[qb + src]) - 33.0;
        }
        ok = sum / (q_e - q_b) >= q_thr;
      }
      if (ok) win_layers[first_window[t] + wi].push_back(LayerRef{r, q_b, q_e - q_b, t_b - ws, t_e - 1 - ws, rc});
    }
  }

  // ---- 4. flatten for the POA batch ------------------------------------------------------------------------
  std::vector<u8> codes, quals;
  std::vector<u64> layer_off{0}, out_off{0};
  std::vector<u32> begins, ends, hasq, win_off{0};
  const bool any_q = h_quals != nullptr;
  for (u32 t = 0; t < T.n; ++t) {
    const u32 tlen = T.h_len[t];
    for (u64 wi = 0; wi < first_window[t + 1] - first_window[t]; ++wi) {
      const u32 ws = static_cast<u32>(wi) * w;
      const u32 bl = std::min<u32>(w, tlen - ws);
      for (u32 x = 0; x < bl; ++x) codes.push_back(code_at(T.h_packed, T.h_word_off[t], ws + x));
      if (any_q) quals.insert(quals.end(), bl, static_cast<u8>('!'));  // racon's dummy backbone quality
      layer_off.push_back(codes.size());
      begins.push_back(0);
      ends.push_back(bl ? bl - 1 : 0);
      hasq.push_back(1);  // backbone weight 0 ('!'), as racon's dummy quality
      for (const auto& L : win_layers[first_window[t] + wi]) {
        const u32 qlen = R.h_len[L.read];
        for (u32 x = 0; x < L.q_len; ++x) {
          const u32 pos = L.q_begin + x;
          const u8 c = L.rc ? static_cast<u8>(3 - code_at(R.h_packed, R.h_word_off[L.read], qlen - 1 - pos))
                            : code_at(R.h_packed, R.h_word_off[L.read], pos);
          codes.push_back(c);
          if (any_q) quals.push_back(h_quals[h_qual_off[L.read] + (L.rc ? qlen - 1 - pos : pos)]);
        }
        layer_off.push_back(codes.size());
        begins.push_back(L.t_begin);
        ends.push_back(std::min(L.t_end, bl - 1));
        hasq.push_back(any_q ? 1 : 0);
        ++stats.n_layers;
      }
      win_off.push_back(static_cast<u32>(begins.size()));
      out_off.push_back(out_off.back() + 4ULL * bl + 256);
    }
  }
  // a backbone-only quality array is still needed when no read has qualities (backbone weight must be 0)
  std::vector<u8> bb_quals;
  const u8* q_ptr = nullptr;
  if (any_q) {
    q_ptr = quals.data();
  } else {
    bb_quals.assign(codes.size(), static_cast<u8>('!'));
    q_ptr = bb_quals.data();
    // only backbones have has_qual = 1 here
  }
  stats.n_windows = n_windows;
  std::vector<u8> cons(out_off.back() + 16);
  std::vector<u32> cons_len(n_windows), status(n_windows);
  double ms = 0;
  poa_consensus_batch(e, codes.data(), q_ptr, layer_off.data(), begins.data(), ends.data(), hasq.data(), win_off.data(),
                      static_cast<u32>(n_windows), m, n, g, trim ? 1 : 0, cons.data(), out_off.data(), cons_len.data(),
                      status.data(), &ms);
  stats.poa_ms = ms;

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
}

}  // namespace rvn
