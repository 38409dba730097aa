// polish.hip — one racon polishing round on the device (racon::Polisher::Polish as driven by raven::Polish,
// RavenLib/src/polish.cc:43-51 with e = 0.3, w = 500, trim = true, AlignCfg m/n/g), racon's own pipeline stage by stage:
//   1. index the targets (unitigs) with ram(15, 5), Filter(0.001), Map every read (avoid_equal = avoid_symmetric =
//      false) in windows of 2^30 read bases; per read keep the longest overlap, drop it when
//      1 - min(span)/max(span) > e                                                     [map.hip + best_overlap_kernel]
//   2. global alignment PATH of every kept read against its target span (edlib NW, EDLIB_TASK_PATH in racon) and the
//      breakpoints at every multiple of w on the target: first / last aligned pair per window
//      (racon find_breaking_points_from_cigar)                                                          [nwpath.hip]
//   3. window layers: a piece is used if it has >= 0.02 w bases, begin < end, and (with qualities) a mean Phred >= q;
//      layers of a window in the stable order of their begin position; the backbone carries racon's dummy '!' quality
//      (weight 0).  Layers are descriptors into the packed read sets already in HBM       [layer_* kernels below]
//   4. window consensus = the POA kernels (poa2.hip / poa.hip)
//   5. stitch the windows of each target, polished ratio per target                              [stitch_kernel]
// Everything between the mapping and the final consensus bytes stays in HBM; the host plans the alignment batches
// from one 36-byte record per read and reads back per-window status words.
#include <algorithm>
#include <cstring>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nwpath.h"
#include "poa.h"

namespace rvn {

namespace {

// number of non-zero bytes of flags[0, n) added to *out (n < 2^32)
__global__ __launch_bounds__(256) void count_flags_kernel(const u8* __restrict__ flags, u64 n, u32* __restrict__ out) {
  u32 c = 0;
  for (u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<u64>(gridDim.x) * blockDim.x)
    c += flags[i] ? 1u : 0u;
  c = wave_sum(c);
  if (lane_id() == 0 && c) atomicAdd(out, c);
}

// racon: a layer is used only if the mean Phred of its bases reaches q.  One wave per layer.
__global__ __launch_bounds__(256) void layer_quality_kernel(const PoaLayer* __restrict__ layers, u32 n_layers,
                                                           const u8* __restrict__ read_quals, u32 qual_shift,
                                                           double q_thr, u8* __restrict__ ok) {
  const u32 li = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (li >= n_layers) return;
  const PoaLayer L = layers[li];
  const int lane = lane_id();
  if (!(L.flags & kLayerQual) || (L.flags & kLayerTarget) || L.len == 0) {
    if (lane == 0) ok[li] = 1;
    return;
  }
  u32 sum = 0;
  for (u32 i = lane; i < L.len; i += 64)
    sum += static_cast<u32>(read_quals[L.qual_off + (poa_layer_src_pos(L, i) >> qual_shift)]) - 33u;
  sum = wave_sum(sum);
  if (lane == 0) ok[li] = static_cast<double>(sum) / static_cast<double>(L.len) >= q_thr ? 1 : 0;
}

__device__ __forceinline__ u32 span_len(const Overlap& o) {
  const u32 a = o.lhs_end - o.lhs_begin, b = o.rhs_end - o.rhs_begin;
  return a > b ? a : b;
}

// racon: the longest overlap of every read (the first one among equals), dropped when its two spans differ by more
// than the error threshold.  One thread per read of the mapped window [first, first + n).
__global__ void best_overlap_kernel(const Overlap* __restrict__ ovl, const u32* __restrict__ roff, u32 first, u32 n,
                                    double err_thr, const u32* __restrict__ id_to_t, u32 n_ids,
                                    Overlap* __restrict__ best, u32* __restrict__ best_t) {
  const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const u32 b = roff[r], e = roff[r + 1];
  u32 bt = 0xFFFFFFFFu;
  Overlap bo{};
  if (e > b) {
    bo = ovl[b];
    u32 bl = span_len(bo);
    for (u32 i = b + 1; i < e; ++i) {
      const Overlap o = ovl[i];
      const u32 l = span_len(o);
      if (bl < l) {
        bo = o;
        bl = l;
      }
    }
    const double a = bo.lhs_end - bo.lhs_begin, c = bo.rhs_end - bo.rhs_begin;
    const double err = 1.0 - (a < c ? a : c) / (a > c ? a : c);
    if (!(err > err_thr) && bo.rhs_id < n_ids) bt = id_to_t[bo.rhs_id];
  }
  best[first + r] = bo;
  best_t[first + r] = bt;
}

struct WinMeta {  // host-built, one per window of the processed range
  u64 t_word;     // first word of the window's target
  u32 target, start, len, t_len;  // target index, first base, bases (<= w), length of the target
  u32 out_off, pad_;
};

// racon's layer rules on one window record; true = the piece becomes a layer of window `wi` of its target
__device__ __forceinline__ bool layer_rules(const NwWindowRec& rec, u32 w, u32* begin, u32* end) {
  if (rec.first_t == 0xFFFFFFFFu) return false;
  const u32 qlen = rec.last_q - rec.first_q;
  if (static_cast<double>(qlen) < 0.02 * w) return false;  // racon: breaking_points[j+1].second - [j].second < 0.02 * w
  const u32 ws = (rec.first_t / w) * w;
  *begin = rec.first_t - ws;
  *end = rec.last_t - ws - 1;
  return *begin < *end;  // racon Window::AddLayer rejects begin >= end
}

// pass 1: layers per window.  One thread per alignment job.
__global__ void layer_count_kernel(const NwJob* __restrict__ jobs, u32 n_jobs, const NwWindowRec* __restrict__ recs,
                                   const u64* __restrict__ first_window, u32 w, u64 W0, u64 W1,
                                   u32* __restrict__ win_cnt, u8* __restrict__ keep) {
  const u32 ji = blockIdx.x * blockDim.x + threadIdx.x;
  if (ji >= n_jobs) return;
  const NwJob J = jobs[ji];
  for (u32 x = 0; x < J.n_windows; ++x) {
    const NwWindowRec rec = recs[J.bp_off + x];
    u32 b, e;
    u8 k = 0;
    if (layer_rules(rec, w, &b, &e)) {
      const u64 gw = first_window[J.target] + rec.first_t / w;
      if (gw >= W0 && gw < W1) {
        atomicAdd(&win_cnt[gw - W0], 1u);
        k = 1;
      }
    }
    keep[J.bp_off + x] = k;
  }
}

// pass 2: layer descriptors into their window's segment (any order; window_finish_kernel sorts them)
__global__ void layer_fill_kernel(const NwJob* __restrict__ jobs, u32 n_jobs, const NwWindowRec* __restrict__ recs,
                                  const u8* __restrict__ keep, const u64* __restrict__ first_window, u32 w, u64 W0,
                                  const u32* __restrict__ win_off, u32* __restrict__ win_fill,
                                  const WinMeta* __restrict__ meta, const u64* __restrict__ qual_off, u32 qual_flag,
                                  PoaLayer* __restrict__ lays, u64* __restrict__ keys, u32* __restrict__ max_len) {
  const u32 ji = blockIdx.x * blockDim.x + threadIdx.x;
  if (ji >= n_jobs) return;
  const NwJob J = jobs[ji];
  u32 local_max = 0;
  for (u32 x = 0; x < J.n_windows; ++x) {
    if (!keep[J.bp_off + x]) continue;
    const NwWindowRec rec = recs[J.bp_off + x];
    u32 b, e;
    layer_rules(rec, w, &b, &e);
    const u32 wi = static_cast<u32>(first_window[J.target] + rec.first_t / w - W0);
    const u32 slot = win_off[wi] + wi + 1 + atomicAdd(&win_fill[wi], 1u);
    const u32 bl = meta[wi].len;
    PoaLayer L{};
    L.code_off = J.r_word;
    L.qual_off = qual_flag ? qual_off[J.read] : 0;
    L.len = rec.last_q - rec.first_q;
    L.begin = b;
    L.end = e < bl - 1 ? e : bl - 1;
    L.flags = kLayerPacked | (J.rc ? kLayerRc : 0u) | (qual_flag ? kLayerQual : 0u);
    L.q_begin = rec.first_q;
    L.q_len = J.r_len;
    // band guide: layer offsets at eighths of the target span, interpolated between the path's samples
    {
      const u32 ws = (rec.first_t / w) * w;
      const u32 span = rec.last_t - rec.first_t;  // == end - begin + 1 before the clamp
      u32 pt = rec.first_t, po = 0;               // last known point at or before tau
      int g = 0;
      u32 prev = 0;
      for (u32 i = 1; i < 8; ++i) {
        const u32 tau = rec.first_t + static_cast<u32>((static_cast<u64>(span) * i) / 8);
        u32 nt = rec.last_t, no = L.len;  // next known point after tau
        for (; g < 8; ++g) {
          const u32 gt = ws + static_cast<u32>((static_cast<u64>(g) * w) / 8);
          if (rec.grid[g] == 0xFFFFu || gt <= rec.first_t || gt >= rec.last_t) continue;
          if (gt <= tau) {
            pt = gt;
            po = rec.grid[g];
            continue;
          }
          nt = gt;
          no = rec.grid[g];
          break;
        }
        u32 off = po;
        if (nt > pt && no > po) off = po + static_cast<u32>((static_cast<u64>(tau - pt) * (no - po)) / (nt - pt));
        off = off < prev ? prev : off;
        off = off > L.len ? L.len : off;
        L.way[i - 1] = static_cast<u16>(off < 0xFFFFu ? off : 0xFFFFu);
        prev = off;
      }
      L.pad_ = 0;
    }
    lays[slot] = L;
    keys[slot] = (static_cast<u64>(b) << 32) | ji;  // racon: stable sort by begin; jobs are in read (= overlap) order
    local_max = local_max > L.len ? local_max : L.len;
  }
  if (local_max) atomicMax(max_len, local_max);
}

// pass 3, one wave per window: backbone descriptor, layers into racon's order (begin, then read), PoaWindow
__global__ __launch_bounds__(256) void window_finish_kernel(const WinMeta* __restrict__ meta, u32 n_windows,
                                                           const u32* __restrict__ win_off,
                                                           const u32* __restrict__ win_cnt,
                                                           const PoaLayer* __restrict__ lays_in,
                                                           const u64* __restrict__ keys, PoaLayer* __restrict__ lays,
                                                           PoaWindow* __restrict__ wins) {
  const u32 wi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wi >= n_windows) return;
  const int lane = lane_id();
  const u32 base = win_off[wi] + wi;
  const u32 n = win_cnt[wi];
  const WinMeta M = meta[wi];
  if (lane == 0) {
    PoaLayer B{};
    B.code_off = M.t_word;
    B.len = M.len;
    B.begin = 0;
    B.end = M.len ? M.len - 1 : 0;
    B.flags = kLayerPacked | kLayerTarget | kLayerZeroW;  // weight 0 = racon's dummy '!' backbone quality
    B.q_begin = M.start;
    B.q_len = M.t_len;
    for (u32 i = 1; i < 8; ++i) B.way[i - 1] = static_cast<u16>(static_cast<u64>(B.len) * i / 8);
    B.pad_ = 0;
    lays[base] = B;
    PoaWindow W;
    W.layer_first = base;
    W.n_layers = n + 1;
    W.out_off = M.out_off;
    W.out_cap = 2 * M.len + 128;
    wins[wi] = W;
  }
  for (u32 a = lane; a < n; a += 64) {
    const u64 ka = keys[base + 1 + a];
    u32 rank = 0;
    for (u32 b = 0; b < n; ++b) rank += keys[base + 1 + b] < ka ? 1u : 0u;  // keys are distinct (job index)
    lays[base + 1 + rank] = lays_in[base + 1 + a];
  }
}

// consensus of window wi -> its place in the stitched output.  One workgroup per window.
__global__ __launch_bounds__(256) void stitch_kernel(const PoaWindow* __restrict__ wins, const u32* __restrict__ len,
                                                    const u64* __restrict__ cons_off, const u8* __restrict__ out,
                                                    u8* __restrict__ final_out) {
  const u32 wi = blockIdx.x;
  const u32 n = len[wi];
  const u8* src = out + wins[wi].out_off;
  u8* dst = final_out + cons_off[wi];
  for (u32 i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

}  // namespace

// Step 1 of a racon round for the reads [r_first, r_last): index the targets, map the reads (batches of about 1 GB as in
// racon), keep the best overlap of every read (longest, error filter `err_thr`; racon Polisher::Initialize).
// best[i] / best_t[i] describe read r_first + i; best_t == 0xFFFFFFFF: the read is not used.  Reads are independent of
// each other here, which is what lets the sharded round map a slice per rank and all-gather the (small) table.
namespace {
__global__ void flag_origins_kernel(const u64* __restrict__ org, u64 n, u64 flags, u64* __restrict__ out) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) out[i] = org[i] | flags;
}
}  // namespace

void polish_map_best(Engine& e, ReadsDev& T, ReadsDev& R, u32 r_first, u32 r_last, double err_thr,
                     std::vector<Overlap>& best, std::vector<u32>& best_t, u64* n_overlaps) {
  hipStream_t s = e.stream;
  r_last = std::min(r_last, R.n);
  r_first = std::min(r_first, r_last);
  const u32 nr_all = r_last - r_first;
  best.assign(nr_all, Overlap{});
  best_t.assign(nr_all, 0xFFFFFFFFu);
  if (n_overlaps) *n_overlaps = 0;
  if (T.n == 0 || nr_all == 0) return;
  {
    StageTimer t(e, StageTimes::kSketch);
    e.query_ready = false;
    e.index.has_query_flags = false;
    e.index.all_query = false;
    sketch_raw(e, T, 0, T.n, e.index_sketch);
    t.stop();
  }
  // Option polish_join (round 6; VERDICT r03-r05: the probes' over-fetch — one random binary search + origin gather per read
  // minimizer, 46-89 GB of cache lines per launch for ~7 GB of data): the reads' minimizers are SORTED WITH the targets' —
  // query-only entries ahead of the members of every run, as the sharded pass carries the reads of earlier index batches
  // (kForeignFlag) — and one streaming pass over the runs emits the matches (map_batch_query_only).  The targets' sketch is
  // kept aside and appended to every read batch's; Filter's cutoff is that of the targets' index alone, computed once, as
  // racon does.  Bit-identical to the probes and SLOWER at C4 (88 against 59 ms of kernels per round: sorting 1.1 G entries
  // costs what probing them costs) — hence an option, not the default.
  const bool joined = R.ids_are_indices && e.opt.polish_join != 0;
  const u64 t_count = e.index_sketch.count;
  const size_t vb = e.val64 ? 8 : 4;
  if (joined && t_count) {
    RVN_HIP(hipMemcpyAsync(e.pl_tval.get<unsigned char>((t_count + 1) * vb), e.index_sketch.val.ptr, t_count * vb, hipMemcpyDeviceToDevice, s));
    RVN_HIP(hipMemcpyAsync(e.pl_torg.get<u64>(t_count + 1), e.index_sketch.org.ptr, t_count * 8, hipMemcpyDeviceToDevice, s));
  }
  index_build(e, e.index_sketch, !joined);
  index_filter(e, 0.001);
  const u32 occurrence = e.index.occurrence;
  u32 max_id = 0;
  for (u32 t = 0; t < T.n; ++t) max_id = std::max(max_id, T.h_id[t]);
  {
    std::vector<u32> id_to_t(static_cast<size_t>(max_id) + 1, 0xFFFFFFFFu);
    for (u32 t = 0; t < T.n; ++t) id_to_t[T.h_id[t]] = t;
    u32* d = e.pl_idmap.get<u32>(id_to_t.size());
    RVN_HIP(hipMemcpyAsync(d, id_to_t.data(), id_to_t.size() * 4, hipMemcpyHostToDevice, s));
    RVN_HIP(rvn_stream_sync(s));
  }
  u64 cache_default = 0;
  {
    size_t free_b = 0, total_b = 0;
    RVN_HIP(hipMemGetInfo(&free_b, &total_b));
    cache_default = total_b / 8;
  }
  auto swap_sketch = [](Sketch& a, Sketch& b) {
    std::swap(a.first, b.first);
    std::swap(a.last, b.last);
    std::swap(a.count, b.count);
    for (auto pr : {std::make_pair(&a.val, &b.val), std::make_pair(&a.org, &b.org), std::make_pair(&a.read_off, &b.read_off)}) {
      std::swap(pr.first->ptr, pr.second->ptr);
      std::swap(pr.first->cap, pr.second->cap);
    }
  };
  Overlap* d_best = e.pl_best.get<Overlap>(static_cast<size_t>(R.n) + 1);
  u32* d_best_t = e.pl_best_t.get<u32>(static_cast<size_t>(R.n) + 1);
  const bool keep = e.keep_anchors;
  e.keep_anchors = false;
  try {
    // racon maps its reads in batches of about 1 GB; the match / overlap memory of a round is that of one batch
    u64 bases = 0;
    for (u32 r0 = r_first, r = r_first; r < r_last; ++r) {
      bases += R.h_len[r];
      if (r != r_last - 1 && bases < (1ULL << 30)) continue;
      bases = 0;
      MapOut& mo = e.map_out;
      if (joined && t_count) {
        {
          StageTimer t(e, StageTimes::kQuery);
          sketch_raw(e, R, r0, r + 1, e.raw_sketch);
          t.stop();
        }
        const u64 nq = e.raw_sketch.count, m = nq + t_count;
        if (m >= (1ULL << 32)) throw HipError("[raven_hip] polishing round: a read batch with >= 2^32 minimizers");
        unsigned char* cv = e.index_sketch.val.get<unsigned char>((m + 1) * vb);
        u64* co = e.index_sketch.org.get<u64>(m + 1);
        if (nq) {
          RVN_HIP(hipMemcpyAsync(cv, e.raw_sketch.val.ptr, nq * vb, hipMemcpyDeviceToDevice, s));
          RVN_KLAUNCH(kKGather, flag_origins_kernel<<<div_up(nq, 256), 256, 0, s>>>(e.raw_sketch.org.as<u64>(), nq,
                                                                                    kQueryFlag | kForeignFlag, co));
        }
        RVN_HIP(hipMemcpyAsync(cv + nq * vb, e.pl_tval.ptr, t_count * vb, hipMemcpyDeviceToDevice, s));
        RVN_HIP(hipMemcpyAsync(co + nq, e.pl_torg.ptr, t_count * 8, hipMemcpyDeviceToDevice, s));
        e.index_sketch.first = r0;
        e.index_sketch.last = r + 1;
        e.index_sketch.count = m;
        index_build(e, e.index_sketch, false);   // stable sort: the reads' entries stay ahead of the targets' in every run
        e.index.occurrence = occurrence;
        e.index.has_query_flags = true;
        e.index.all_query = false;
        map_batch_query_only(e, R, r0, r + 1, nq, mo);
      } else {
        // The reads' sketch does not change between rounds (only the targets do): kept in HBM per read batch and handed to
        // map_batch as its prepared query sketch (round 6: 32 ms of sketch kernels per C4 round).  One read set at a time,
        // within the option's budget; handed back with the scratch (engine_release_scratch).
        Engine::PolishSketch* slot = nullptr;
        const u64 budget = e.opt.polish_sketch_cache_mb < 0 ? cache_default : static_cast<u64>(e.opt.polish_sketch_cache_mb) << 20;
        if (budget) {
          if (e.polish_sketch_owner != R.serial) {
            e.polish_sketches.clear();
            e.polish_sketch_owner = R.serial;
          }
          for (auto& c : e.polish_sketches)
            if (c->first == r0 && c->last == r + 1) slot = c.get();
        }
        if (slot) {  // the kept sketch becomes the query sketch of this call ...
          swap_sketch(e.query_sketch, slot->sk);
          e.query_ready = true;
          e.query_ready_first = r0;
          e.query_ready_last = r + 1;
          e.query_ready_minhash = false;
        }
        map_batch(e, R, r0, r + 1, false, false, false, false, mo);
        if (!slot && budget) {
          u64 held = 0;
          for (auto& c : e.polish_sketches) held += c->sk.val.cap + c->sk.org.cap + c->sk.read_off.cap;
          if (held + e.query_sketch.val.cap + e.query_sketch.org.cap + e.query_sketch.read_off.cap <= budget) {
            e.polish_sketches.emplace_back(new Engine::PolishSketch());
            slot = e.polish_sketches.back().get();
            slot->first = r0;
            slot->last = r + 1;
          }
        }
        if (slot) swap_sketch(e.query_sketch, slot->sk);  // ... and goes back (or: the one just computed is kept)
      }
      if (n_overlaps) *n_overlaps += mo.n_overlaps;
      const u32 nr = r + 1 - r0;
      RVN_KLAUNCH(kKBestOverlap, best_overlap_kernel<<<div_up(nr, 256), 256, 0, s>>>(
                                     mo.ovl.as<Overlap>(), mo.ovl_read_off.as<u32>(), r0, nr, err_thr,
                                     e.pl_idmap.as<u32>(), max_id + 1, d_best, d_best_t));
      r0 = r + 1;
    }
  } catch (...) {
    e.keep_anchors = keep;
    e.polish_sketches.clear();  // (a kept sketch may be out on loan as the query sketch: nothing kept is trusted after a failure)
    e.polish_sketch_owner = 0;
    throw;
  }
  e.keep_anchors = keep;
  RVN_HIP(hipMemcpyAsync(best.data(), d_best + r_first, static_cast<size_t>(nr_all) * sizeof(Overlap), hipMemcpyDeviceToHost, s));
  RVN_HIP(hipMemcpyAsync(best_t.data(), d_best_t + r_first, static_cast<size_t>(nr_all) * 4, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
}

void polish_round(Engine& e, ReadsDev& T, ReadsDev& R, const u8* h_quals, const u64* h_qual_off, double q_thr,
                  double err_thr, u32 w, bool trim, int m, int n, int g, std::vector<std::vector<u8>>& polished,
                  std::vector<double>& ratio, PolishStats& stats, u64 win_first, u64 win_last,
                  std::vector<u32>* win_count, std::vector<u32>* win_polished, const PolishDirectOut* direct) {
  hipStream_t s = e.stream;
  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
  const auto t_all = clk::now();
  const bool dbg = knob("RVN_POLISH_DEBUG") != nullptr;
  auto lap = [&, last = clk::now()](const char* what) mutable {
    if (dbg) {
      (void)rvn_stream_sync(s);
      std::fprintf(stderr, "[raven_hip] polish: %-32s %8.1f ms\n", what, ms_since(last));
    }
    last = clk::now();
  };
  polished.assign(T.n, {});
  ratio.assign(T.n, 0.0);
  stats = PolishStats();
  e.polish_target_reads.assign(T.n, 0);
  if (win_count) win_count->assign(T.n, 0);
  if (win_polished) win_polished->assign(T.n, 0);
  if (T.n == 0) return;
  double host_ms = 0;

  // ---- 1. map the reads to the targets; best overlap per read (device), unless the caller supplies the table -----------
  std::vector<Overlap> best;
  std::vector<u32> best_t;
  if (e.polish_given_valid) {  // rvn_polish_set_best: the sharded round maps a slice of the reads per rank
    if (e.polish_given_best.size() != R.n)
      throw std::invalid_argument("[raven_hip] polishing round: the supplied best-overlap table does not match the read set");
    best.swap(e.polish_given_best);
    best_t.swap(e.polish_given_best_t);
    e.polish_given_valid = false;
    for (u32 r = 0; r < R.n; ++r)
      if (best_t[r] != 0xFFFFFFFFu && best_t[r] >= T.n)
        throw std::invalid_argument("[raven_hip] polishing round: best-overlap table names a target that does not exist");
  } else {
    u64 n_ovl = 0;
    polish_map_best(e, T, R, 0, R.n, err_thr, best, best_t, &n_ovl);
    stats.n_overlaps = n_ovl;
  }
  stats.map_ms = ms_since(t_all);
  lap("index + map + best overlap");

  // ---- host planning: window tables of the processed range, one alignment job per used read -------------------------
  auto t_host = clk::now();
  std::vector<u64> first_window(static_cast<size_t>(T.n) + 1, 0);
  for (u32 t = 0; t < T.n; ++t) first_window[t + 1] = first_window[t] + (static_cast<u64>(T.h_len[t]) + w - 1) / w;
  const u64 n_windows_all = first_window[T.n];
  const u64 W0 = std::min<u64>(win_first, n_windows_all), W1 = std::min<u64>(win_last, n_windows_all);
  const u32 nw = static_cast<u32>(W1 > W0 ? W1 - W0 : 0);
  std::vector<NwJob> jobs;
  u64 n_recs = 0;
  {
    // per read: does it give a job of this window range, and how many window records (in parallel); then the records' offsets
    // in read order and the jobs themselves (in parallel again) — 295 000 reads at C4, and the GPU waits for this planning
    std::vector<u32> wins_of(R.n, 0);  // 0: no job
    parallel_for(R.n, 16384, [&](size_t r0, size_t r1) {
      for (size_t r = r0; r < r1; ++r) {
        if (best_t[r] == 0xFFFFFFFFu) continue;
        const Overlap& o = best[r];
        if (o.rhs_end <= o.rhs_begin || o.lhs_end <= o.lhs_begin) continue;
        const u32 t = best_t[r];
        const u64 g_lo = first_window[t] + o.rhs_begin / w, g_hi = first_window[t] + (o.rhs_end - 1) / w;
        if (g_hi < W0 || g_lo >= W1) continue;  // another rank's windows
        wins_of[r] = (o.rhs_end - 1) / w - o.rhs_begin / w + 1;
      }
    });
    std::vector<u64> rec_off(static_cast<size_t>(R.n) + 1, 0);
    std::vector<u32> job_of(static_cast<size_t>(R.n) + 1, 0);
    for (u32 r = 0; r < R.n; ++r) {
      if (best_t[r] != 0xFFFFFFFFu) {
        ++stats.n_reads_used;
        ++e.polish_target_reads[best_t[r]];
      }
      rec_off[r + 1] = rec_off[r] + wins_of[r];
      job_of[r + 1] = job_of[r] + (wins_of[r] ? 1u : 0u);
    }
    n_recs = rec_off[R.n];
    jobs.resize(job_of[R.n]);
    parallel_for(R.n, 16384, [&](size_t r0, size_t r1) {
      for (size_t r = r0; r < r1; ++r) {
        if (!wins_of[r]) continue;
        const Overlap& o = best[r];
        const u32 t = best_t[r];
        const u32 qlen = R.h_len[r];
        const bool rc = o.strand == 0;
        NwJob J{};
        J.t_word = T.h_word_off[t];
        J.r_word = R.h_word_off[r];
        J.t_begin = o.rhs_begin;
        J.n = o.rhs_end - o.rhs_begin;
        J.q_begin = rc ? qlen - o.lhs_end : o.lhs_begin;  // racon reverse-complements the read (step 2 of its pipeline)
        J.m = o.lhs_end - o.lhs_begin;
        J.r_len = qlen;
        J.rc = rc ? 1 : 0;
        J.read = static_cast<u32>(r);
        J.target = t;
        J.n_windows = wins_of[r];
        J.bp_off = rec_off[r];
        jobs[job_of[r]] = J;
      }
    });
  }
  std::vector<WinMeta> meta(nw);
  {
    u32 t = 0;
    u64 out_total = 0;
    for (u32 i = 0; i < nw; ++i) {
      const u64 gw = W0 + i;
      while (first_window[t + 1] <= gw) ++t;
      WinMeta& M = meta[i];
      M.target = t;
      M.t_word = T.h_word_off[t];
      M.start = static_cast<u32>(gw - first_window[t]) * w;
      M.t_len = T.h_len[t];
      M.len = std::min<u32>(w, M.t_len - M.start);
      if (out_total + 2ULL * M.len + 128 > 0xFFFFFFFFULL)
        throw std::invalid_argument("[raven_hip] polishing round: window range too large for one batch (use rvn_polish_round_range)");
      M.out_off = static_cast<u32>(out_total);
      M.pad_ = 0;
      out_total += 2ULL * M.len + 128;
    }
    stats.n_windows = nw;
    host_ms += ms_since(t_host);
    if (nw == 0) {
      stats.host_ms = host_ms;
      stats.total_ms = ms_since(t_all);
      return;
    }
    WinMeta* d_meta = e.pl_win_meta.get<WinMeta>(nw + 1);
    u64* d_fw = e.pl_first_window.get<u64>(first_window.size());
    RVN_HIP(hipMemcpyAsync(d_meta, meta.data(), nw * sizeof(WinMeta), hipMemcpyHostToDevice, s));
    RVN_HIP(hipMemcpyAsync(d_fw, first_window.data(), first_window.size() * 8, hipMemcpyHostToDevice, s));
    (void)e.pl_out.get<u8>(out_total + 16);
  }
  const WinMeta* d_meta = e.pl_win_meta.as<WinMeta>();
  const u64* d_fw = e.pl_first_window.as<u64>();
  u8* d_out = e.pl_out.as<u8>();
  lap("planning");

  // ---- 2. alignment paths + breakpoints ------------------------------------------------------------------------------
  NwWindowRec* d_recs = e.pl_recs.get<NwWindowRec>(n_recs + 1);
  NwStats nst;
  nw_breakpoints(e, T, R, jobs, w, d_recs, n_recs, nst);
  stats.align_ms = nst.ms;
  stats.n_aligned = nst.n_aligned;
  stats.n_align_retries = nst.n_retries;
  stats.n_dropped_layers = nst.n_unaligned;
  stats.align_band_cells = nst.band_cells;
  stats.align_store_bytes = nst.store_bytes;
  lap("alignment paths + breakpoints");

  // ---- 3. window layers (descriptors; bases and qualities stay where they are) ----------------------------------------
  const u32 nj = static_cast<u32>(jobs.size());
  PoaSrc src{};
  src.packed_reads = R.packed.as<u64>();
  src.packed_targets = T.packed.as<u64>();
  src.qual_shift = 0;
  const u64* d_qual_off = nullptr;
  if (h_quals != nullptr) {  // caller-supplied per-base qualities: to HBM for this call
    const u64 qtotal = h_qual_off[R.n];
    u8* d_q = e.polish_quals.get<u8>(qtotal + 16);
    u64* d_qo = e.pl_qual_off.get<u64>(static_cast<size_t>(R.n) + 1);
    RVN_HIP(hipMemcpyAsync(d_q, h_quals, qtotal, hipMemcpyHostToDevice, s));
    RVN_HIP(hipMemcpyAsync(d_qo, h_qual_off, (static_cast<size_t>(R.n) + 1) * 8, hipMemcpyHostToDevice, s));
    src.read_quals = d_q;
    d_qual_off = d_qo;
  } else if (R.qual_shift >= 0) {  // qualities attached to the read set: already resident
    src.read_quals = R.quals.as<u8>();
    src.qual_shift = static_cast<u32>(R.qual_shift);
    d_qual_off = R.qual_off.as<u64>();
  }
  const bool any_q = src.read_quals != nullptr;
  NwJob* d_jobs = e.nw_jobs.get<NwJob>(nj + 1);
  if (nj) RVN_HIP(hipMemcpyAsync(d_jobs, jobs.data(), static_cast<size_t>(nj) * sizeof(NwJob), hipMemcpyHostToDevice, s));
  u32* d_win_cnt = e.pl_win_cnt.get<u32>(nw + 1);
  u32* d_win_fill = e.pl_win_fill.get<u32>(nw + 1);
  u32* d_win_off = e.pl_win_off.get<u32>(nw + 2);
  u8* d_keep = e.pl_keep.get<u8>(n_recs + 1);
  u32* d_misc = e.pl_misc.get<u32>(4);
  RVN_HIP(hipMemsetAsync(d_win_cnt, 0, static_cast<size_t>(nw) * 4, s));
  RVN_HIP(hipMemsetAsync(d_win_fill, 0, static_cast<size_t>(nw) * 4, s));
  RVN_HIP(hipMemsetAsync(d_misc, 0, 16, s));
  if (nj)
    RVN_KLAUNCH(kKLayerBuild, layer_count_kernel<<<div_up(nj, 128), 128, 0, s>>>(d_jobs, nj, d_recs, d_fw, w, W0, W1,
                                                                                 d_win_cnt, d_keep));
  exclusive_scan_u32_u32(d_win_cnt, d_win_off, nw, e.scan_tmp, s);
  const u64 n_read_layers = read_back(e, d_win_off + nw, 4);
  const u64 n_lay = n_read_layers + nw;
  if (n_lay >= 0xFFFFFFFFULL) throw std::invalid_argument("[raven_hip] polishing round: too many layers for one batch");
  PoaLayer* d_lays_tmp = e.pl_lays_tmp.get<PoaLayer>(n_lay + 1);
  PoaLayer* d_lays = e.pl_lays.get<PoaLayer>(n_lay + 1);
  u64* d_keys = e.pl_keys.get<u64>(n_lay + 1);
  PoaWindow* d_wins = e.pl_wins.get<PoaWindow>(nw + 1);
  if (nj)
    RVN_KLAUNCH(kKLayerBuild, layer_fill_kernel<<<div_up(nj, 128), 128, 0, s>>>(
                                  d_jobs, nj, d_recs, d_keep, d_fw, w, W0, d_win_off, d_win_fill, d_meta, d_qual_off,
                                  any_q ? 1u : 0u, d_lays_tmp, d_keys, d_misc));
  RVN_KLAUNCH(kKLayerBuild, window_finish_kernel<<<div_up(nw, 4), 256, 0, s>>>(d_meta, nw, d_win_off, d_win_cnt,
                                                                              d_lays_tmp, d_keys, d_lays, d_wins));
  const u32 max_len = static_cast<u32>(read_back(e, d_misc, 4));
  if (any_q) {  // racon's mean-quality filter as per-layer flags
    u8* d_ok = e.pl_ok.get<u8>(n_lay + 16);
    layer_quality_kernel<<<div_up(n_lay, 4), 256, 0, s>>>(d_lays, static_cast<u32>(n_lay), src.read_quals, src.qual_shift,
                                                           q_thr, d_ok);
    RVN_LAUNCH_CHECK();
    src.layer_ok = d_ok;
    if (q_thr > 0) {  // statistics only: layers that survive (counted on the device: the flags of 6 M layers went to the host
                      // and through a scalar loop with the GPU idle, every round)
      count_flags_kernel<<<1024, 256, 0, s>>>(d_ok, n_lay, d_misc + 1);
      RVN_LAUNCH_CHECK();
      const u64 kept = read_back(e, d_misc + 1, 4);
      stats.n_layers = kept - nw;  // backbones are always flagged ok
    } else {
      stats.n_layers = n_read_layers;
    }
  } else {
    stats.n_layers = n_read_layers;
  }
  e.polish_last_windows = nw;
  e.polish_last_layers = n_lay;
  e.polish_last_w0 = W0;
  e.polish_last_has_ok = any_q;
  e.polish_last_read_off = R.h_word_off;
  lap("window layers");

  // ---- 4. window consensus -----------------------------------------------------------------------------------------------
  u32* d_len = e.pl_len.get<u32>(nw + 1);
  u32* d_status = e.pl_status.get<u32>(nw + 1);
  RVN_HIP(hipMemsetAsync(d_len, 0, static_cast<size_t>(nw) * 4, s));
  std::vector<u32> h_status;
  double poa_ms = 0;
  if (knob("RVN_POLISH_SKIP_POA")) {  // profiling of the stages before the consensus only: empty windows
    std::fprintf(stderr, "[raven_hip] RVN_POLISH_SKIP_POA is set: the round returns UNPOLISHED windows (profiling switch)\n");
    h_status.assign(nw, 0);
  } else {
    poa_run_dev(e, d_wins, d_lays, nw, src, w, std::max<u32>(max_len, w), m, n, g, trim ? 1 : 0, d_out, d_len, d_status,
                h_status, &poa_ms);
  }
  stats.poa_ms = poa_ms;
  lap("POA");

  // ---- 5. stitch the windows in order; per-target results -----------------------------------------------------------
  u64* d_cons_off = e.pl_cons_off.get<u64>(nw + 2);
  exclusive_scan_u32_u64(d_len, d_cons_off, nw, e.scan_tmp, s);
  std::vector<u64> cons_off(nw + 1);
  RVN_HIP(hipMemcpyAsync(cons_off.data(), d_cons_off, (static_cast<size_t>(nw) + 1) * 8, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  const u64 total = cons_off[nw];
  u8* d_final = e.pl_final.get<u8>(total + 16);
  RVN_KLAUNCH(kKStitch, stitch_kernel<<<nw, 256, 0, s>>>(d_wins, d_len, d_cons_off, d_out, d_final));
  u8* h_final = e.pin_out.get<u8>(total + 16);
  RVN_HIP(hipMemcpyAsync(h_final, d_final, total, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  t_host = clk::now();
  std::vector<u64> t_windows(T.n, 0), t_polished(T.n, 0);
  for (u32 i = 0; i < nw; ++i) {
    const u32 t = meta[i].target;
    const u32 st = h_status[i];
    ++t_windows[t];
    if (st == 1) ++t_polished[t];
    if (st >= 2) ++stats.n_failed_windows;
  }
  {
    std::vector<std::pair<u32, u32>> span(T.n, {0u, 0u});  // windows [first, second) of the range that belong to target t
    {
      u32 i = 0;
      for (u32 t = 0; t < T.n; ++t) {
        const u32 i0 = i;
        while (i < nw && meta[i].target == t) ++i;
        span[t] = {i0, i};
      }
    }
    std::atomic<bool> too_small{false};
    parallel_for(T.n, 1, [&](size_t t0, size_t t1) {  // (first touch of 100 MB of host pages at C4: a few threads)
      for (size_t t = t0; t < t1; ++t) {
        if (direct) direct->len[t] = 0;
        if (span[t].second <= span[t].first) continue;
        const u8* from = h_final + cons_off[span[t].first];
        const u64 bytes = cons_off[span[t].second] - cons_off[span[t].first];
        if (!direct) {
          polished[t].assign(from, from + bytes);
        } else if (bytes > direct->off[t + 1] - direct->off[t]) {
          too_small = true;
        } else {
          std::memcpy(direct->out + direct->off[t], from, bytes);
          direct->len[t] = bytes;
        }
      }
    });
    if (too_small) throw std::invalid_argument("[raven_hip] polishing round: output buffer too small");
    // (a round over every window leaves the targets' consensus, in target order, in pl_final: rvn_polish_output_as_reads)
    e.pl_last_valid = false;
    if (W0 == 0 && W1 == n_windows_all && T.n > 0) {
      e.pl_last_off.assign(static_cast<size_t>(T.n) + 1, 0);
      bool all = true;
      for (u32 t = 0; t < T.n; ++t) {
        all = all && span[t].second > span[t].first;  // (a target without a window in the range has no consensus here)
        e.pl_last_off[t] = all ? cons_off[span[t].first] : 0;
      }
      e.pl_last_off[T.n] = total;
      e.pl_last_valid = all && T.n > 0;
    }
    for (u32 t = 0; t < T.n; ++t) {
      ratio[t] = t_windows[t] ? static_cast<double>(t_polished[t]) / t_windows[t] : 0.0;
      stats.n_polished_windows += t_polished[t];
      if (win_count) (*win_count)[t] = static_cast<u32>(t_windows[t]);
      if (win_polished) (*win_polished)[t] = static_cast<u32>(t_polished[t]);
    }
  }
  host_ms += ms_since(t_host);
  lap("stitch + read-back");
  stats.host_ms = host_ms;
  stats.total_ms = ms_since(t_all);
}

}  // namespace rvn
