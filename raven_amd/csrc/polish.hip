// polish.hip — one racon polishing round on the device (racon::Polisher::Polish as driven by raven::Polish,
// RavenLib/src/polish.cc:43-51 with e = 0.3, w = 500, trim = true, AlignCfg m/n/g):
//   1. index the targets (unitigs), Filter(0.001), Map every read (avoid_equal = avoid_symmetric = false) — the
//      same device engine as the overlap phase, with the chain ANCHORS of every overlap kept;
//   2. keep each read's longest overlap, drop it when 1 - min(span)/max(span) > e;
//   3. cut the read into window layers.  racon takes the breakpoints from an edlib NW path of the whole overlap
//      (CIGAR); here they come from the chain anchors: a window boundary inside an anchor's k-mer, or inside the
//      exact-match extension of the two anchors around it, is cut exactly; otherwise a small unit-cost NW of the
//      unmatched remainder of that one anchor gap (a few bases to a few hundred) decides, and pieces begin / end on
//      aligned pairs exactly as racon's find_breaking_points does.  No base-level alignment of whole reads is
//      needed.  Layers shorter than 0.02 w, or (with qualities) below the mean-quality threshold q, are dropped
//      exactly as in racon;
//   4. window consensus = the POA kernel (poa.hip); 5. stitch the windows, polished ratio per target.
// Step 3 differs from racon only in which optimal alignment decides a cut when several exist; the consensus is
// compared with the CPU restatement of racon's own pipeline (identical or within a few edits).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <thread>
#include <vector>

#include "poa.h"
#include "polish_cut.h"

namespace rvn {

namespace {

inline u32 code_at(const std::vector<u64>& packed, u64 word_off, u32 i) {
  return static_cast<u32>(packed[word_off + (i >> 5)] >> ((i << 1) & 63)) & 3u;
}

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
                  std::vector<double>& ratio, PolishStats& stats, u64 win_first, u64 win_last,
                  std::vector<u32>* win_count, std::vector<u32>* win_polished) {
  hipStream_t s = e.stream;
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
  const auto t_all = clk::now();
  const bool dbg = std::getenv("RVN_POLISH_DEBUG") != nullptr;
  auto lap = [&, last = clk::now()](const char* what) mutable {
    if (dbg) std::fprintf(stderr, "[raven_hip] polish: %-28s %8.1f ms\n", what, ms_since(last));
    last = clk::now();
  };
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
  // the anchors are the bulk of the read-back: pinned staging while that is cheap, a plain buffer beyond 256 MB
  const bool pin_anchors = (mo.n_matches + 1) * 8 <= (256ULL << 20);
  u64* anchors = pin_anchors ? e.pin_big.get<u64>(mo.n_matches + 1) : e.host_big.get<u64>(mo.n_matches + 1);
  if (O) {
    RVN_HIP(hipMemcpyAsync(anchors, mo.anchors.ptr, mo.n_matches * 8, hipMemcpyDeviceToHost, s));
    RVN_HIP(hipMemcpy(ovl.data(), mo.ovl.ptr, O * sizeof(Overlap), hipMemcpyDeviceToHost));
    RVN_HIP(hipMemcpy(aoff.data(), mo.anchor_off.ptr, O * 8, hipMemcpyDeviceToHost));
    RVN_HIP(hipMemcpy(acnt.data(), mo.anchor_cnt.ptr, O * 4, hipMemcpyDeviceToHost));
    RVN_HIP(hipStreamSynchronize(s));
  }
  RVN_HIP(hipMemcpy(roff.data(), mo.ovl_read_off.ptr, roff.size() * 4, hipMemcpyDeviceToHost));
  stats.n_overlaps = O;
  stats.map_ms = ms_since(t_all);
  lap("index + map + read-back");

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

  lap("best overlap per read");
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
  std::vector<u32> win_t_all(n_windows);
  for (u32 t = 0; t < T.n; ++t)
    for (u64 gw = first_window[t]; gw < first_window[t + 1]; ++gw) win_t_all[gw] = t;
  struct LayerRef {
    u32 read, q_begin, q_len, t_begin, t_end, rc;
    u16 way[7];  // PoaLayer::way: where the chain says the piece is at 1/8 .. 7/8 of its target span
  };
  const u32 k = e.k;
  e.polish_target_reads.assign(T.n, 0);
  // reads are independent: host threads each take a contiguous range of reads and emit (window, layer) pairs;
  // the ranges are appended in read order, so a window's layers keep racon's order (overlap order, then the stable
  // sort by begin position in step 4)
  struct Emit {
    u64 window;
    LayerRef layer;
  };
  struct Part {
    std::vector<Emit> emits;
    std::vector<u32> target_reads;
    u64 used = 0, dropped = 0;
    u64 n_cuts = 0, n_nw = 0, nw_cells = 0;
  };
  u32 thr_cap = 128;
  if (const char* ev = std::getenv("RVN_HOST_THREADS")) thr_cap = std::max(1, std::atoi(ev));
  const u32 n_thr = std::max(1u, std::min<u32>(thr_cap, std::min<u32>(std::thread::hardware_concurrency(), R.n / 128 + 1)));
  // The window range of this call is processed in chunks: while the GPU runs the POA of one chunk (background
  // thread), the host threads cut the reads of the next one.  A read belongs to every chunk its overlap touches.
  const u64 W0 = std::min<u64>(win_first, n_windows), W1 = std::min<u64>(win_last, n_windows);
  const u64 chunk_w = e.polish_chunk_windows ? e.polish_chunk_windows : std::max<u64>(1, W1 - W0);
  const u32 n_chunks = static_cast<u32>(W1 > W0 ? (W1 - W0 + chunk_w - 1) / chunk_w : 0);
  std::vector<std::vector<u32>> chunk_reads(n_chunks);
  for (u32 r = 0; r < R.n; ++r) {
    if (!best[r].valid) continue;
    const Overlap& o = best[r].o;
    if (o.rhs_id >= id_to_t.size() || id_to_t[o.rhs_id] == 0xFFFFFFFFu || best[r].acnt < 2) {
      best[r].valid = false;
      continue;
    }
    const u32 t = id_to_t[o.rhs_id];
    ++stats.n_reads_used;
    ++e.polish_target_reads[t];
    const u64 g_lo = first_window[t] + o.rhs_begin / w, g_hi = first_window[t] + (o.rhs_end ? (o.rhs_end - 1) / w : 0);
    if (g_hi < W0 || g_lo >= W1) continue;
    const u32 c_a = static_cast<u32>((std::max(g_lo, W0) - W0) / chunk_w);
    const u32 c_b = static_cast<u32>((std::min(g_hi, W1 - 1) - W0) / chunk_w);
    for (u32 c = c_a; c <= c_b; ++c) chunk_reads[c].push_back(r);
  }
  u64 c_lo = 0, c_hi = 0;  // window range of the chunk being cut
  const std::vector<u32>* cur_reads = nullptr;
  std::vector<Part> parts;
  auto work = [&](u32 ti) {
    Part& P = parts[ti];
    const std::vector<u32>& cr = *cur_reads;
    const size_t i_lo = cr.size() * ti / n_thr, i_hi = cr.size() * (ti + 1) / n_thr;
    std::vector<std::pair<u32, u32>> an;
    CutScratch sc;
    for (size_t ii = i_lo; ii < i_hi; ++ii) {
      const u32 r = cr[ii];
      const Overlap& o = best[r].o;
      const u32 t = id_to_t[o.rhs_id];
      const u32 qlen = R.h_len[r];
      const bool rc = o.strand == 0;
      // anchors as (t, q') increasing in both; q' in the orientation that matches the target
      an.resize(best[r].acnt);
      for (u32 i = 0; i < best[r].acnt; ++i) {
        const u64 a = anchors[best[r].aoff + i];
        const u32 qp = static_cast<u32>(a >> 32), tp = static_cast<u32>(a);
        an[i] = rc ? std::make_pair(tp, qlen - qp - k) : std::make_pair(tp, qp);
      }
      if (rc) std::reverse(an.begin(), an.end());
      if (an.size() < 2) continue;
      // read position at target coordinate B (a window boundary inside the chain): exact inside an anchor's
      // k-mer or inside the exact-match extension of the bracketing anchors (the bases are compared on the host
      // copies of the packed sets); only the unmatched remainder of the gap — a handful of bases around the
      // error that ended the matches — is split proportionally
      const u64 t_wo = T.h_word_off[t], r_wo = R.h_word_off[r];
      auto tbase = [&](u32 x) -> u32 { return code_at(T.h_packed, t_wo, x); };
      auto qbase = [&](u32 x) -> u32 {  // read in the orientation of the target
        return rc ? 3u - code_at(R.h_packed, r_wo, qlen - 1 - x) : code_at(R.h_packed, r_wo, x);
      };
      // cut(B): the (read, target) position pairs at which the piece left of boundary B ends (target <= B) and the
      // piece right of it begins (target >= B).  Inside an exact-match region both are (q(B), B); when B falls into
      // the unmatched remainder of an anchor gap — the few bases around the error(s) that ended the exact
      // extensions — a small unit-cost NW of the two remainders decides, so pieces tile the windows exactly as
      // racon's CIGAR breakpoints do.  Only a remainder longer than kCutMax is given to neither window (the piece
      // on the left ends where the exact region before it ends, the one on the right begins where the exact
      // region after it begins): a guessed cut would append true neighbour bases to a window.
      auto cut = [&](u32 B) -> WindowCut { return window_cut(an, k, B, T.h_len[t], qlen, tbase, qbase, sc); };
      const u32 t_first = an.front().first, t_last_end = an.back().first + k;  // chain covers [t_first, t_last_end)
      const u32 q_first = an.front().second, q_last_end = an.back().second + k;
      WindowCut carry{q_first, t_first, q_first, t_first};
      u32 carry_at = 0xFFFFFFFFu;  // boundary `carry` was computed for
      for (u32 wi = t_first / w; static_cast<u64>(wi) * w < t_last_end; ++wi) {
        if (first_window[t] + wi + 1 < c_lo || first_window[t] + wi > c_hi) continue;  // far from this chunk
        const u32 ws = wi * w;
        const u32 we = std::min<u32>(T.h_len[t], ws + w);  // exclusive
        u32 t_b = std::max(ws, t_first), t_e = std::min(we, t_last_end);  // [t_b, t_e)
        if (t_e <= t_b + 1) continue;
        u32 q_b = q_first, q_e = q_last_end;
        if (t_b != t_first) {  // the cut at this window's start is normally the previous window's end cut
          if (carry_at != t_b) {
            carry = cut(t_b);
            carry_at = t_b;
          }
          q_b = carry.qr;
          t_b = carry.tr;
        }
        if (t_e != t_last_end) {
          carry = cut(t_e);
          carry_at = t_e;
          ++P.n_cuts;
          q_e = carry.ql;
          t_e = carry.tl;
        }
        if (t_e <= t_b + 1 || t_e > we) continue;
        if (q_e > qlen) q_e = qlen;
        if (q_e <= q_b || (q_e - q_b) < 0.02 * w) continue;
        {  // cuts are exact, so a piece may legitimately be much shorter or longer than its target span (real
           // reads lose whole homopolymer runs); only an absurd ratio — a chain that jumped a repeat copy — is dropped
          const double span = t_e - t_b, ql = q_e - q_b;
          if (ql > 2.0 * span + 32 || span > 2.0 * ql + 32) {
            ++P.dropped;
            continue;
          }
        }
        LayerRef lr{r, q_b, q_e - q_b, t_b - ws, t_e - 1 - ws, rc, {}};
        {  // band guide: read offsets at eighths of the target span, linear between the bracketing anchors
          const u32 span = t_e - t_b;
          u32 prev = 0;
          for (u32 i = 1; i < 8; ++i) {
            const u32 tau = t_b + static_cast<u32>(static_cast<u64>(span) * i / 8);
            size_t lo = 0, hi = an.size();
            while (hi - lo > 1) {
              const size_t mid = (lo + hi) / 2;
              if (an[mid].first <= tau) lo = mid;
              else hi = mid;
            }
            u32 q = an[lo].second + (tau >= an[lo].first ? std::min(tau - an[lo].first, k) : 0u);
            if (tau > an[lo].first + k && lo + 1 < an.size() && an[lo + 1].first > an[lo].first + k &&
                an[lo + 1].second > an[lo].second + k) {
              const double f = static_cast<double>(tau - an[lo].first - k) / (an[lo + 1].first - an[lo].first - k);
              q = an[lo].second + k + static_cast<u32>(f * (an[lo + 1].second - an[lo].second - k));
            }
            u32 off = q > q_b ? q - q_b : 0;
            off = std::min(std::max(off, prev), q_e - q_b);
            lr.way[i - 1] = static_cast<u16>(std::min<u32>(off, 0xFFFFu));
            prev = off;
          }
        }
        if (first_window[t] + wi < c_lo || first_window[t] + wi >= c_hi) continue;  // another chunk's / rank's window
        P.emits.push_back(Emit{first_window[t] + wi, lr});
      }
    }
    P.n_nw = sc.n_nw;
    P.nw_cells = sc.nw_cells;
  };
  const bool any_q = h_quals != nullptr;
  PoaSrc src{};
  src.packed_reads = R.packed.as<u64>();
  src.packed_targets = T.packed.as<u64>();
  if (any_q) {  // qualities to HBM once per call
    const u64 qtotal = h_qual_off[R.n];
    u8* d_q = e.polish_quals.get<u8>(qtotal + 16);
    RVN_HIP(hipMemcpyAsync(d_q, h_quals, qtotal, hipMemcpyHostToDevice, s));
    RVN_HIP(hipStreamSynchronize(s));
    src.read_quals = d_q;
  }
  if (win_count) win_count->assign(T.n, 0);
  if (win_polished) win_polished->assign(T.n, 0);
  std::vector<u64> t_windows(T.n, 0), t_polished(T.n, 0);
  for (u32 t = 0; t < T.n; ++t) polished[t].reserve(static_cast<size_t>(T.h_len[t]) + T.h_len[t] / 16 + 1024);

  struct Chunk {
    u64 lo = 0, hi = 0;
    std::vector<PoaWindow> wins;
    std::vector<PoaLayer> lays;
    std::vector<u64> out_off;
    std::vector<u32> cons_len, status;
    u32 max_bb = 1, max_len = 1;
    double ms = 0;
    u64 kept_layers = 0;
  };
  Chunk slots[2];
  std::vector<std::vector<LayerRef>> win_layers;
  std::vector<u32> win_t;
  std::vector<u64> lay_first;
  std::thread bg;
  std::exception_ptr bg_err;
  double host_busy = 0;
  u64 dbg_cuts = 0, dbg_nw = 0, dbg_cells = 0;

  std::vector<std::vector<u8>> win_cons(W1 - W0);  // consensus of every window of the range, stitched at the end
  std::vector<u32> win_status(W1 - W0, 0);
  std::vector<PoaWindow> fb_wins;  // windows both bands could not do, with their layers
  std::vector<PoaLayer> fb_lays;
  std::vector<u64> fb_gw;

  // background: quality flags + POA of one chunk (the only HIP work while the main thread cuts the next chunk)
  auto run_chunk = [&](Chunk* C) {
    try {
      RVN_HIP(hipSetDevice(e.device));
      PoaSrc csrc = src;
      const u32 nl = static_cast<u32>(C->lays.size());
      if (any_q && nl) {  // racon's mean-quality filter as per-layer flags computed on the device
        PoaLayer* d_l = e.tmp_d.get<PoaLayer>(C->lays.size() + 1);
        RVN_HIP(hipMemcpyAsync(d_l, C->lays.data(), C->lays.size() * sizeof(PoaLayer), hipMemcpyHostToDevice, s));
        u8* d_ok = e.tmp_a.get<u8>(C->lays.size() + 16);
        layer_quality_kernel<<<(nl + 3) / 4, 256, 0, s>>>(d_l, nl, src.read_quals, q_thr, d_ok);
        RVN_HIP(hipGetLastError());
        csrc.layer_ok = d_ok;
        if (q_thr > 0) {  // only for the statistics: how many layers survive
          std::vector<u8> okh(nl);
          RVN_HIP(hipMemcpyAsync(okh.data(), d_ok, nl, hipMemcpyDeviceToHost, s));
          RVN_HIP(hipStreamSynchronize(s));
          u64 kept = 0;
          for (u32 i = 0; i < nl; ++i) kept += (okh[i] && !(C->lays[i].flags & kLayerTarget)) ? 1 : 0;
          C->kept_layers = kept;
        }
      }
      u8* cons = e.pin_out.get<u8>(C->out_off.back() + 16);
      // with several chunks the rare windows that need the full-matrix kernel are collected and run ONCE at the end
      // (that kernel's latency is ~0.25 s per launch, whatever the number of windows)
      poa_run(e, C->wins, C->lays, csrc, C->max_bb, C->max_len, m, n, g, trim ? 1 : 0, cons, C->out_off.back(),
              C->cons_len.data(), C->status.data(), &C->ms, n_chunks == 1);
    } catch (...) {
      bg_err = std::current_exception();
    }
  };
  auto finish_chunk = [&](Chunk* C) {  // join + stitch the windows of the chunk, in window order
    if (bg.joinable()) bg.join();
    if (bg_err) std::rethrow_exception(bg_err);
    const u8* cons = static_cast<const u8*>(e.pin_out.ptr);
    stats.poa_ms += C->ms;
    for (u64 gw = C->lo; gw < C->hi; ++gw) {
      const u64 i = gw - C->lo;
      win_status[gw - W0] = C->status[i];
      win_cons[gw - W0].assign(cons + C->out_off[i], cons + C->out_off[i] + C->cons_len[i]);
      if (n_chunks > 1 && (C->status[i] & 0xFF) >= 2) {  // keep its layers for the final full-matrix batch
        PoaWindow fw = C->wins[i];
        const u32 lf = fw.layer_first;
        fw.layer_first = static_cast<u32>(fb_lays.size());
        fb_lays.insert(fb_lays.end(), C->lays.begin() + lf, C->lays.begin() + lf + fw.n_layers);
        fb_wins.push_back(fw);
        fb_gw.push_back(gw);
      }
    }
    if (any_q && q_thr > 0) stats.n_layers += C->kept_layers;
    else stats.n_layers += C->lays.size() - C->wins.size();
  };

  Chunk* pending = nullptr;
  for (u32 c = 0; c < n_chunks; ++c) {
    const auto t_prep = clk::now();
    Chunk* C = &slots[c & 1];
    c_lo = W0 + static_cast<u64>(c) * chunk_w;
    c_hi = std::min(W1, c_lo + chunk_w);
    C->lo = c_lo;
    C->hi = c_hi;
    const u64 nwc = c_hi - c_lo;
    cur_reads = &chunk_reads[c];
    parts.assign(n_thr, Part());
    {
      std::vector<std::thread> pool;
      for (u32 ti = 1; ti < n_thr; ++ti) pool.emplace_back(work, ti);
      work(0);
      for (auto& th : pool) th.join();
    }
    win_layers.assign(nwc, {});
    for (const Part& P : parts) {  // thread order == read order
      stats.n_dropped_layers += P.dropped;
      dbg_cuts += P.n_cuts;
      dbg_nw += P.n_nw;
      dbg_cells += P.nw_cells;
      for (const Emit& em : P.emits) win_layers[em.window - c_lo].push_back(em.layer);
    }
    // ---- 4. layer descriptors: bases and qualities stay in HBM (packed read sets) ----
    C->wins.assign(nwc, PoaWindow{});
    C->out_off.assign(nwc + 1, 0);
    lay_first.assign(nwc + 1, 0);
    for (u64 i = 0; i < nwc; ++i) {
      const u64 gw = c_lo + i;
      const u32 t = win_t_all[gw];
      const u32 bl = std::min<u32>(w, T.h_len[t] - static_cast<u32>(gw - first_window[t]) * w);
      lay_first[i + 1] = lay_first[i] + 1 + win_layers[i].size();
      C->out_off[i + 1] = C->out_off[i] + 2ULL * bl + 128;
    }
    C->lays.resize(lay_first[nwc]);
    C->cons_len.assign(nwc, 0);
    C->status.assign(nwc, 0);
    C->max_bb = 1;
    C->max_len = 1;
    C->kept_layers = 0;
    C->ms = 0;
    {
      const u32 n_fill = static_cast<u32>(std::max<u64>(1, std::min<u64>(n_thr, nwc / 256 + 1)));
      std::vector<std::pair<u32, u32>> maxes(n_fill, {1u, 1u});
      auto fill = [&](u32 ti) {
        const u64 i_lo = nwc * ti / n_fill, i_hi = nwc * (ti + 1) / n_fill;
        u32 mb = 1, ml = 1;
        for (u64 i = i_lo; i < i_hi; ++i) {
          const u64 gw = c_lo + i;
          const u32 t = win_t_all[gw];
          const u32 tlen = T.h_len[t];
          const u32 ws = static_cast<u32>(gw - first_window[t]) * w;
          const u32 bl = std::min<u32>(w, tlen - ws);
          PoaLayer* out_l = C->lays.data() + lay_first[i];
          PoaLayer B{};
          B.code_off = T.h_word_off[t];
          B.len = bl;
          B.begin = 0;
          B.end = bl ? bl - 1 : 0;
          B.flags = kLayerPacked | kLayerTarget | kLayerZeroW;  // weight 0 = racon's dummy '!' backbone quality
          B.q_begin = ws;
          B.q_len = tlen;
          poa_layer_linear_way(B);
          *out_l++ = B;
          mb = std::max(mb, bl);
          auto& wl = win_layers[i];
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
            for (int x = 0; x < 7; ++x) P.way[x] = L.way[x];
            *out_l++ = P;
            ml = std::max(ml, L.q_len);
          }
          C->wins[i].layer_first = static_cast<u32>(lay_first[i]);
          C->wins[i].n_layers = static_cast<u32>(lay_first[i + 1] - lay_first[i]);
          C->wins[i].out_off = static_cast<u32>(C->out_off[i]);
          C->wins[i].out_cap = 2 * bl + 128;
        }
        maxes[ti] = {mb, ml};
      };
      std::vector<std::thread> pool;
      for (u32 ti = 1; ti < n_fill; ++ti) pool.emplace_back(fill, ti);
      fill(0);
      for (auto& th : pool) th.join();
      for (const auto& m2 : maxes) {
        C->max_bb = std::max(C->max_bb, m2.first);
        C->max_len = std::max(C->max_len, m2.second);
      }
      C->max_len = std::max(C->max_len, C->max_bb);
    }
    host_busy += ms_since(t_prep);
    if (pending) finish_chunk(pending);  // the previous chunk's POA ran while this one was being cut
    pending = C;
    bg = std::thread(run_chunk, C);
  }
  if (pending) finish_chunk(pending);
  lap("cuts + descriptors + POA (pipelined)");
  if (dbg)
    std::fprintf(stderr, "[raven_hip] polish: %u chunk(s), %llu cuts, %llu with a residual NW (%llu cells), %u host threads, "
                 "host busy %.1f ms\n", n_chunks, (unsigned long long)dbg_cuts, (unsigned long long)dbg_nw,
                 (unsigned long long)dbg_cells, n_thr, host_busy);

  if (!fb_wins.empty()) {  // one full-matrix batch for what neither band width could align
    u64 oo = 0;
    u32 mb = 1, ml = 1;
    for (auto& fw : fb_wins) {
      fw.out_off = static_cast<u32>(oo);
      oo += fw.out_cap;
      mb = std::max(mb, fb_lays[fw.layer_first].len);
      for (u32 x = 0; x < fw.n_layers; ++x) ml = std::max(ml, fb_lays[fw.layer_first + x].len);
    }
    PoaSrc fsrc = src;
    if (any_q) {
      const u32 nl = static_cast<u32>(fb_lays.size());
      PoaLayer* d_l = e.tmp_d.get<PoaLayer>(fb_lays.size() + 1);
      RVN_HIP(hipMemcpyAsync(d_l, fb_lays.data(), fb_lays.size() * sizeof(PoaLayer), hipMemcpyHostToDevice, s));
      u8* d_ok = e.tmp_a.get<u8>(fb_lays.size() + 16);
      layer_quality_kernel<<<(nl + 3) / 4, 256, 0, s>>>(d_l, nl, src.read_quals, q_thr, d_ok);
      RVN_HIP(hipGetLastError());
      fsrc.layer_ok = d_ok;
    }
    u8* cons = e.pin_out.get<u8>(oo + 16);
    std::vector<u32> fl(fb_wins.size()), fs(fb_wins.size());
    double fms = 0;
    const int mode = e.poa_mode;
    e.poa_mode = 1;
    try {
      poa_run(e, fb_wins, fb_lays, fsrc, mb, ml, m, n, g, trim ? 1 : 0, cons, oo, fl.data(), fs.data(), &fms);
    } catch (...) {
      e.poa_mode = mode;
      throw;
    }
    e.poa_mode = mode;
    stats.poa_ms += fms;
    for (size_t i = 0; i < fb_wins.size(); ++i) {
      win_status[fb_gw[i] - W0] = fs[i];
      win_cons[fb_gw[i] - W0].assign(cons + fb_wins[i].out_off, cons + fb_wins[i].out_off + fl[i]);
    }
    e.poa_fallback_windows = static_cast<u32>(fb_wins.size());
  }
  // ---- 5. stitch the windows in order; per-target results -----------------------------------------------
  for (u64 gw = W0; gw < W1; ++gw) {
    const u32 t = win_t_all[gw];
    const u32 st = win_status[gw - W0];
    ++t_windows[t];
    if (st == 1) ++t_polished[t];
    if (st >= 2) ++stats.n_failed_windows;
    polished[t].insert(polished[t].end(), win_cons[gw - W0].begin(), win_cons[gw - W0].end());
  }
  stats.n_windows = W1 - W0;
  for (u32 t = 0; t < T.n; ++t) {
    ratio[t] = t_windows[t] ? static_cast<double>(t_polished[t]) / t_windows[t] : 0.0;
    stats.n_polished_windows += t_polished[t];
    if (win_count) (*win_count)[t] = static_cast<u32>(t_windows[t]);
    if (win_polished) (*win_polished)[t] = static_cast<u32>(t_polished[t]);
  }
  stats.host_ms = host_busy;
  stats.total_ms = ms_since(t_all);
}

}  // namespace rvn
