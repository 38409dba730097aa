// group.hip — N GPUs behind ONE host process: the sharded FindOverlapsAndCreatePiles pass and the sharded polishing round
// of SURVEY §8(e) driven from C++ (one worker thread + one engine per device), for callers that are not a
// one-process-per-GPU torch.distributed job: raven::ConstructGraph / raven::Polish own ONE ram::MinimizerEngine /
// racon::Polisher (RavenLib/src/construct.cc:661-669, polish.cc:43-51), so "all visible GPUs" has to live behind that
// one object.  The stages are the rvn_shard_* entry points raven_amd/sharded.py drives over RCCL; here the three
// exchanges of a flush window are in-process:
//     every rank publishes (device pointer, per-destination counts) -> barrier -> every rank pulls its pieces with
//     hipMemcpyPeerAsync (xGMI between devices, a plain device copy between two engines of one device) -> barrier
// i.e. the same all-to-all by pairwise puts/gets that ncclSend/ncclRecv performs, without a communicator (one process:
// every device pointer is directly addressable).  The Filter's count histogram and the polishing round's tables are
// reduced / gathered through host memory of the process.  A device may be listed several times (virtual ranks on one
// GPU: how the single-GPU test boxes run this path).  Results are bit-identical to the single-engine calls
// (tests/cpp/group_test.cpp, tests/test_gpu_group.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raven_hip.h"
#include "engine.h"

namespace rvn {
void set_last_error(const std::string& msg);  // engine.hip: the calling thread's rvn_last_error()
}

using rvn::DevBuf;
using rvn::u32;
using rvn::u64;
using rvn::u8;

struct rvn_group {
  std::vector<rvn_engine*> eng;
  std::vector<int> dev;
  std::vector<hipStream_t> copy_stream;
  std::vector<uint8_t> peer_direct;  // [i * world + j]: rank i reaches rank j's memory directly (same device, or peer access on)
  // barrier + shared tables of the collectives
  std::mutex mu;
  std::condition_variable cv;
  u32 arrived = 0;
  u64 generation = 0;
  std::atomic<bool> failed{false};
  std::string error;
  int error_code = RVN_OK;
  struct Slot {
    const void* ptr = nullptr;
    std::vector<u64> counts;
    std::vector<u64> host;   // host-side payload of a reduce / gather
    std::vector<u8> bytes;
  };
  std::vector<Slot> slot;
  u32 world() const { return static_cast<u32>(eng.size()); }
};

namespace {

struct Abort {};  // another rank failed: leave quietly

struct Rank {
  rvn_group& g;
  u32 r;
  rvn_engine* e;
  hipStream_t cs;
  Rank(rvn_group& g_, u32 r_) : g(g_), r(r_), e(g_.eng[r_]), cs(g_.copy_stream[r_]) {}

  void fail(int code, const std::string& msg) {
    {
      std::lock_guard<std::mutex> lk(g.mu);
      if (!g.failed.load()) {
        g.error = msg;
        g.error_code = code;
      }
      g.failed.store(true);
    }
    g.cv.notify_all();
    throw Abort();
  }
  void check(int rc) {
    if (rc != RVN_OK) fail(rc, rvn_last_error());
  }
  void hip(hipError_t err, const char* what) {
    if (err != hipSuccess) fail(RVN_EHIP, std::string("[raven_hip] group: ") + what + ": " + hipGetErrorString(err));
  }
  void barrier() {
    std::unique_lock<std::mutex> lk(g.mu);
    if (g.failed.load()) throw Abort();
    const u64 gen = g.generation;
    if (++g.arrived == g.world()) {
      g.arrived = 0;
      ++g.generation;
      g.cv.notify_all();
    } else {
      g.cv.wait(lk, [&] { return g.generation != gen || g.failed.load(); });
      if (g.generation == gen) throw Abort();
    }
  }
  // all-to-all of 64-bit words: piece h of `d_send` (counts[h] words, pieces back to back) goes to rank h; returns the words
  // received from every source in rank order (concatenated in `recv`)
  std::vector<u64> all_to_all(const u64* d_send, const std::vector<u64>& counts, DevBuf& recv) {
    g.slot[r].ptr = d_send;
    g.slot[r].counts = counts;
    barrier();
    std::vector<u64> lens(g.world());
    u64 total = 0;
    for (u32 s = 0; s < g.world(); ++s) total += lens[s] = g.slot[s].counts[r];
    u64* dst = recv.get<u64>(total + 2);
    u64 at = 0;
    for (u32 s = 0; s < g.world(); ++s) {
      u64 src_off = 0;
      for (u32 h = 0; h < r; ++h) src_off += g.slot[s].counts[h];
      if (lens[s])
        hip(hipMemcpyPeerAsync(dst + at, g.dev[r], static_cast<const u64*>(g.slot[s].ptr) + src_off, g.dev[s], lens[s] * 8, cs),
            "peer copy");
      at += lens[s];
    }
    hip(hipStreamSynchronize(cs), "peer copy");
    barrier();  // every rank has pulled: send buffers may be reused
    return lens;
  }
  // element-wise sum of equally long host vectors over the ranks
  std::vector<u64> all_reduce_sum(const std::vector<u64>& v) {
    g.slot[r].host = v;
    barrier();
    std::vector<u64> out(v.size(), 0);
    for (u32 s = 0; s < g.world(); ++s)
      for (size_t i = 0; i < v.size(); ++i) out[i] += g.slot[s].host[i];
    barrier();
    return out;
  }
  // concatenation of the ranks' host vectors in rank order (+ the piece lengths)
  std::vector<u64> all_gather(const std::vector<u64>& v, std::vector<u64>* lens = nullptr) {
    g.slot[r].host = v;
    barrier();
    std::vector<u64> out;
    if (lens) lens->clear();
    for (u32 s = 0; s < g.world(); ++s) {
      out.insert(out.end(), g.slot[s].host.begin(), g.slot[s].host.end());
      if (lens) lens->push_back(g.slot[s].host.size());
    }
    barrier();
    return out;
  }
  std::vector<u8> all_gather_bytes(const std::vector<u8>& v, std::vector<u64>* lens) {
    g.slot[r].bytes = v;
    barrier();
    std::vector<u8> out;
    lens->clear();
    for (u32 s = 0; s < g.world(); ++s) {
      out.insert(out.end(), g.slot[s].bytes.begin(), g.slot[s].bytes.end());
      lens->push_back(g.slot[s].bytes.size());
    }
    barrier();
    return out;
  }
};

// contiguous read ranges balanced by bases (raven_amd/sharded.py::partition_reads)
std::vector<u32> partition_reads(const uint32_t* lengths, u32 n, u32 world) {
  std::vector<double> cum(static_cast<size_t>(n) + 1, 0.0);
  for (u32 i = 0; i < n; ++i) cum[i + 1] = cum[i] + lengths[i];
  std::vector<u32> b(world + 1, 0);
  b[world] = n;
  for (u32 h = 1; h < world; ++h) {
    const double target = cum[n] * h / world;
    b[h] = static_cast<u32>(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
  }
  for (u32 h = 1; h <= world; ++h) b[h] = std::max(b[h], b[h - 1]);
  return b;
}

// query windows of a pass as the reference flushes them (construct.cc:56-70)
std::vector<std::pair<u32, u32>> flush_windows(const uint32_t* lengths, u32 n, u64 flush_bases) {
  std::vector<std::pair<u32, u32>> out;
  u32 first = 0;
  u64 acc = 0;
  for (u32 k = 0; k < n; ++k) {
    acc += lengths[k];
    if (k != n - 1 && acc < flush_bases) continue;
    out.emplace_back(first, k + 1);
    first = k + 1;
    acc = 0;
  }
  return out;
}

// ram::MinimizerEngine::Filter on the count-of-counts summed over the ranks (sharded.py::occurrence_from_histogram)
u32 occurrence_of(Rank& R, const std::vector<u64>& hist, const std::vector<u64>& over, double freq) {
  if (freq == 0) return 0xFFFFFFFFu;
  std::vector<u64> all_over = R.all_gather(over);
  const std::vector<u64> h = R.all_reduce_sum(hist);
  u64 u = 0;
  for (u64 x : h) u += x;
  if (u == 0) return 0xFFFFFFFFu;
  const u64 nth = std::min<u64>(static_cast<u64>((1 - freq) * static_cast<double>(u)), u - 1);
  u64 cum = 0;
  u32 c = 0;
  for (; c < 65536; ++c) {
    cum += h[c];
    if (cum >= nth + 1) break;
  }
  if (c >= 65535) {
    u64 below = 0;
    for (u32 x = 0; x < 65535; ++x) below += h[x];
    std::sort(all_over.begin(), all_over.end());
    c = static_cast<u32>(all_over[nth - below]);
  }
  return c + 1;
}

struct PassArgs {
  const uint64_t* packed;
  const uint64_t* word_offsets;
  const uint32_t* lengths;
  u32 n;
  double freq;
  u32 kmax;
  int use_minhash;
  u64 flush_bases;
  u64 index_batch_bases;
  const std::vector<u32>* bounds;
  rvn_pass1** out;
};

void pass_rank(Rank& R, const PassArgs& A) {
  rvn_group& G = R.g;
  const u32 world = G.world(), g = R.r, n_total = A.n;
  R.hip(hipSetDevice(G.dev[g]), "hipSetDevice");
  const std::vector<u32>& bounds = *A.bounds;
  const u32 lo = bounds[g], hi = bounds[g + 1], n_own = hi - lo;
  // this rank's reads, ids = global indices
  std::vector<u64> woff(static_cast<size_t>(n_own) + 1);
  const u64 w0 = A.word_offsets[lo];
  for (u32 i = 0; i <= n_own; ++i) woff[i] = A.word_offsets[lo + i] - w0;
  std::vector<u32> ids(n_own);
  for (u32 i = 0; i < n_own; ++i) ids[i] = lo + i;
  rvn_reads* own = nullptr;
  R.check(rvn_reads_upload(R.e, A.packed + w0, woff[n_own], woff.data(), A.lengths + lo, ids.data(), n_own, &own));
  struct OwnGuard {
    rvn_reads* p;
    ~OwnGuard() { rvn_reads_destroy(p); }
  } own_guard{own};
  DevBuf val, org, val_f, org_f, val_p, org_p, vcat, ocat, grp, pos, seg, per_read, cnt_flat, grp_flat, pos_flat, seg_own, grp_own, pos_own, ovl,
      ovl_p, off_own, recv;

  std::vector<u64> r_split(world);
  for (u32 h = 0; h < world; ++h) r_split[h] = bounds[h + 1] - bounds[h];
  rvn_pass1* p = nullptr;
  R.check(rvn_shard_piles_create(R.e, A.lengths, n_total, &p));
  A.out[g] = p;
  // Index batches as the reference cuts them (construct.cc:32-37: a batch closes with the read that brings its bases to
  // 2^32); every read up to a batch's end is mapped against it (:59-64)
  for (const auto& batch : flush_windows(A.lengths, n_total, A.index_batch_bases)) {
  // 1. sketch: the members of the batch and, as query-only entries, the minhash-selected minimizers of this rank's EARLIER
  //    reads; minimizers to the owner of their hash class (stable partition keeps (read, position) order)
  const u32 f_hi = std::min(std::max(batch.first, lo), hi) - lo;
  const u32 m_lo = f_hi, m_hi = std::min(std::max(batch.second, lo), hi) - lo;
  uint64_t n_for = 0, n_mem = 0;
  if (f_hi > 0) {
    R.check(rvn_shard_sketch_range(R.e, own, 0, f_hi, A.use_minhash, 1, &n_for));
    u64* fv = val_f.get<u64>(n_for + 2);
    u64* fo = org_f.get<u64>(n_for + 2);
    if (n_for) R.check(rvn_shard_sketch_fetch_dev(R.e, fv, fo));
  }
  if (m_hi > m_lo) R.check(rvn_shard_sketch_range(R.e, own, m_lo, m_hi, A.use_minhash, 0, &n_mem));
  const uint64_t n_min = n_for + n_mem;
  u64* d_val = val.get<u64>(n_min + 2);
  u64* d_org = org.get<u64>(n_min + 2);
  if (n_for) {
    R.hip(hipMemcpyAsync(d_val, val_f.as<u64>(), n_for * 8, hipMemcpyDeviceToDevice, R.cs), "copy");
    R.hip(hipMemcpyAsync(d_org, org_f.as<u64>(), n_for * 8, hipMemcpyDeviceToDevice, R.cs), "copy");
    R.hip(hipStreamSynchronize(R.cs), "copy");
  }
  if (n_mem) R.check(rvn_shard_sketch_fetch_dev(R.e, d_val + n_for, d_org + n_for));
  std::vector<u64> cnt(world, 0);
  const u64 *send_v = d_val, *send_o = d_org;
  if (world == 1) {
    cnt[0] = n_min;
  } else {
    u64* pv = val_p.get<u64>(n_min + 2);
    u64* po = org_p.get<u64>(n_min + 2);
    R.check(rvn_shard_split_minimizers_dev(R.e, d_val, d_org, n_min, world, pv, po, cnt.data()));
    send_v = pv;
    send_o = po;
  }
  u64 n_cat = 0;
  for (u64 x : R.all_to_all(send_v, cnt, vcat)) n_cat += x;
  R.all_to_all(send_o, cnt, ocat);
  // 2. index shard; 3. exact global Filter
  uint64_t n_flagged = 0;
  R.check(rvn_shard_count_flagged_dev(R.e, ocat.as<u64>(), n_cat, &n_flagged));
  R.check(rvn_shard_index_build_dev(R.e, vcat.as<u64>(), ocat.as<u64>(), n_cat, A.use_minhash, n_flagged));
  std::vector<u64> hist(65536, 0);
  std::vector<u32> over32(1 << 20);
  uint32_t n_over = 0;
  R.check(rvn_shard_key_histogram(R.e, hist.data(), over32.data(), static_cast<u32>(over32.size()), &n_over));
  if (n_over > over32.size()) {
    over32.resize(n_over);
    R.check(rvn_shard_key_histogram(R.e, hist.data(), over32.data(), static_cast<u32>(over32.size()), &n_over));
  }
  std::vector<u64> over(over32.begin(), over32.begin() + n_over);
  R.check(rvn_engine_set_occurrence(R.e, occurrence_of(R, hist, over, A.freq)));

  // 4.-6. per flush window of query reads (merge + AddLayers + truncation per window, as the reference flushes)
  for (const auto& win : flush_windows(A.lengths, batch.second, A.flush_bases)) {
    uint64_t n_m = 0;
    R.check(rvn_shard_join_range(R.e, n_total, 1, 1, win.first, win.second, &n_m));
    u64* d_grp = grp.get<u64>(n_m + 2);
    u64* d_pos = pos.get<u64>(n_m + 2);
    u64* d_seg = seg.get<u64>(static_cast<size_t>(n_total) + 2);
    u64* d_per = per_read.get<u64>(static_cast<size_t>(n_total) + 2);
    R.check(rvn_shard_join_fetch_dev(R.e, d_grp, d_pos, d_seg));
    R.check(rvn_shard_adjacent_diff_dev(R.e, d_seg, n_total, d_per));
    // matches are in read order: the cut points of the read ranges
    std::vector<u64> cuts(world + 1);
    for (u32 h = 0; h <= world; ++h)
      R.hip(hipMemcpyAsync(&cuts[h], d_seg + bounds[h], 8, hipMemcpyDeviceToHost, R.cs), "read cut points");
    R.hip(hipStreamSynchronize(R.cs), "read cut points");
    std::vector<u64> m_split(world);
    for (u32 h = 0; h < world; ++h) m_split[h] = cuts[h + 1] - cuts[h];
    const std::vector<u64> cnt_lens = R.all_to_all(d_per, r_split, cnt_flat);
    const std::vector<u64> m_lens = R.all_to_all(d_grp, m_split, grp_flat);
    R.all_to_all(d_pos, m_split, pos_flat);
    u64 n_in = 0;
    for (u64 x : m_lens) n_in += x;
    u64* d_seg_own = seg_own.get<u64>(static_cast<size_t>(n_own) + 2);
    u64* d_grp_own = grp_own.get<u64>(n_in + 2);
    u64* d_pos_own = pos_own.get<u64>(n_in + 2);
    std::vector<const u64*> c_ptr(world), g_ptr(world), p_ptr(world);
    u64 at_c = 0, at_m = 0;
    for (u32 h = 0; h < world; ++h) {  // per-source views into the flat receive buffers
      c_ptr[h] = cnt_flat.as<u64>() + at_c;
      g_ptr[h] = grp_flat.as<u64>() + at_m;
      p_ptr[h] = pos_flat.as<u64>() + at_m;
      at_c += cnt_lens[h];
      at_m += m_lens[h];
    }
    R.check(rvn_shard_regroup_dev(R.e, world, c_ptr.data(), g_ptr.data(), p_ptr.data(), m_lens.data(), n_own, d_seg_own,
                                  d_grp_own, d_pos_own));
    // chain
    uint64_t n_o = 0;
    R.check(rvn_shard_chain_dev(R.e, own, d_grp_own, d_pos_own, d_seg_own, n_in, &n_o));
    rvn_overlap* d_ovl = ovl.get<rvn_overlap>(n_o + 2);
    rvn_overlap* d_ovl_p = ovl_p.get<rvn_overlap>(n_o + 2);
    u32* d_off_own = off_own.get<u32>(static_cast<size_t>(n_own) + 2);
    R.check(rvn_engine_map_fetch_dev(R.e, d_ovl, d_off_own));
    // overlaps also to the owner of their rhs read (stable partition; the own ones stay); merge + piles
    std::vector<u64> o_cnt(world + 1, 0);
    R.check(rvn_shard_split_overlaps_dev(R.e, d_ovl, n_o, bounds.data(), world, g, d_ovl_p, o_cnt.data()));
    std::vector<u64> send(world);
    for (u32 h = 0; h < world; ++h) send[h] = 4 * o_cnt[h];  // an overlap = four 64-bit words
    const std::vector<u64> recv_lens = R.all_to_all(reinterpret_cast<const u64*>(d_ovl_p), send, recv);
    std::vector<const rvn_overlap*> parts;
    std::vector<u64> part_n;
    u64 at = 0;
    for (u32 s = 0; s < world; ++s) {
      if (recv_lens[s]) {
        parts.push_back(reinterpret_cast<const rvn_overlap*>(recv.as<u64>() + at));
        part_n.push_back(recv_lens[s] / 4);
      }
      at += recv_lens[s];
    }
    parts.push_back(d_ovl);
    part_n.push_back(n_o);
    R.check(rvn_shard_piles_merge_parts_dev(p, static_cast<u32>(parts.size()), parts.data(), part_n.data(), A.kmax));
  }
  }  // index batches
}

struct PolishArgs {
  const uint64_t *t_packed, *t_woff;
  const uint32_t* t_len;
  u32 nt;
  const uint64_t *r_packed, *r_woff;
  const uint32_t* r_len;
  u32 nr;
  double q, err;
  u32 w;
  int trim, m, n, gp;
  const uint8_t* r_quals;       // Phred+33 per 2^shift bases of a read, or nullptr (unit weights, no quality filter)
  const uint64_t* r_qual_off;
  int qual_shift;
  uint8_t* out_codes;
  const uint64_t* out_offsets;
  uint32_t* out_len;
  double* ratio;
};

void polish_rank(Rank& R, const PolishArgs& A) {
  rvn_group& G = R.g;
  const u32 world = G.world(), g = R.r;
  R.hip(hipSetDevice(G.dev[g]), "hipSetDevice");
  rvn_reads *targets = nullptr, *reads = nullptr;
  R.check(rvn_reads_upload(R.e, A.t_packed, A.t_woff[A.nt], A.t_woff, A.t_len, nullptr, A.nt, &targets));
  struct Guard {
    rvn_reads*& p;
    ~Guard() { rvn_reads_destroy(p); }
  } tg{targets}, rg{reads};
  R.check(rvn_reads_upload(R.e, A.r_packed, A.r_woff[A.nr], A.r_woff, A.r_len, nullptr, A.nr, &reads));
  // qualities travel with the read set (biosoup's block qualities: racon's mean-quality filter and the quality-weighted
  // edges of the window graphs, polish.cc:26-41): every rank holds all reads, so every rank attaches all of them
  if (A.r_quals) R.check(rvn_reads_attach_quality(R.e, reads, A.r_quals, A.r_qual_off, A.qual_shift));
  u64 n_win = 0;
  for (u32 t = 0; t < A.nt; ++t) n_win += (static_cast<u64>(A.t_len[t]) + A.w - 1) / A.w;
  const u64 w_lo = n_win * g / world, w_hi = n_win * (g + 1) / world;
  if (world > 1) {  // reads are mapped independently of each other: a slice per rank, the table all-gathered
    const u32 r_lo = static_cast<u32>(static_cast<u64>(A.nr) * g / world), r_hi = static_cast<u32>(static_cast<u64>(A.nr) * (g + 1) / world);
    std::vector<rvn_overlap> best(r_hi - r_lo);
    std::vector<u32> bt(r_hi - r_lo);
    uint64_t n_ovl = 0;
    R.check(rvn_polish_map_best(R.e, targets, reads, r_lo, r_hi, A.err, best.data(), bt.data(), &n_ovl));
    std::vector<u64> mine(static_cast<size_t>(r_hi - r_lo) * 5);  // 8 + 1 (+ 1 pad) 32-bit words per read
    for (u32 i = 0; i < r_hi - r_lo; ++i) {
      u32 rec[10];
      std::memcpy(rec, &best[i], 32);
      rec[8] = bt[i];
      rec[9] = 0;
      std::memcpy(&mine[static_cast<size_t>(i) * 5], rec, 40);
    }
    const std::vector<u64> table = R.all_gather(mine);
    std::vector<rvn_overlap> all_best(A.nr);
    std::vector<u32> all_bt(A.nr);
    for (u32 i = 0; i < A.nr; ++i) {
      u32 rec[10];
      std::memcpy(rec, &table[static_cast<size_t>(i) * 5], 40);
      std::memcpy(&all_best[i], rec, 32);
      all_bt[i] = rec[8];
    }
    R.check(rvn_polish_set_best(R.e, all_best.data(), all_bt.data(), A.nr));
  }
  // this rank's window range: pieces of every target's consensus
  std::vector<u8> codes(A.out_offsets[A.nt] + 16);
  std::vector<u32> len(A.nt), nw(A.nt), npol(A.nt);
  std::vector<double> ratio(A.nt);
  R.check(rvn_polish_round_range(R.e, targets, reads, nullptr, nullptr, A.q, A.err, A.w, A.trim, A.m, A.n, A.gp, w_lo, w_hi,
                                 codes.data(), A.out_offsets, len.data(), ratio.data(), nw.data(), npol.data(), nullptr));
  std::vector<u8> mine;
  std::vector<u64> lens64(A.nt), counts(2 * static_cast<size_t>(A.nt));
  for (u32 t = 0; t < A.nt; ++t) {
    mine.insert(mine.end(), codes.begin() + A.out_offsets[t], codes.begin() + A.out_offsets[t] + len[t]);
    lens64[t] = len[t];
    counts[t] = nw[t];
    counts[A.nt + t] = npol[t];
  }
  std::vector<u64> piece_bytes;
  const std::vector<u8> all = R.all_gather_bytes(mine, &piece_bytes);
  const std::vector<u64> all_lens = R.all_gather(lens64);
  const std::vector<u64> tot = R.all_reduce_sum(counts);
  if (g == 0) {  // concatenate the pieces in rank order
    std::vector<u64> at(world, 0);
    u64 base = 0;
    for (u32 s = 0; s < world; ++s) {
      at[s] = base;
      base += piece_bytes[s];
    }
    for (u32 t = 0; t < A.nt; ++t) {
      u64 o = A.out_offsets[t];
      for (u32 s = 0; s < world; ++s) {
        const u64 l = all_lens[static_cast<size_t>(s) * A.nt + t];
        std::memcpy(A.out_codes + o, all.data() + at[s], l);
        at[s] += l;
        o += l;
      }
      A.out_len[t] = static_cast<u32>(o - A.out_offsets[t]);
      A.ratio[t] = tot[t] ? static_cast<double>(tot[A.nt + t]) / static_cast<double>(tot[t]) : 0.0;
    }
  }
}

template <class F>
int run_ranks(rvn_group* g, F fn) {
  g->failed.store(false);
  g->error.clear();
  g->error_code = RVN_OK;
  g->arrived = 0;
  std::vector<std::thread> th;
  for (u32 r = 0; r < g->world(); ++r) {
    th.emplace_back([g, r, &fn]() {
      Rank R(*g, r);
      try {
        fn(R);
      } catch (const Abort&) {
      } catch (const std::exception& ex) {
        try {
          R.fail(RVN_EHIP, ex.what());
        } catch (const Abort&) {
        }
      }
    });
  }
  for (auto& t : th) t.join();
  if (g->failed.load()) {
    rvn::set_last_error(g->error);
    return g->error_code ? g->error_code : RVN_EHIP;
  }
  return RVN_OK;
}

}  // namespace

extern "C" {

int rvn_group_create(rvn_group** out, uint32_t k, uint32_t w, uint32_t bandwidth, uint32_t chain, uint32_t matches,
                     uint32_t gap, const int* devices, uint32_t n_devices) {
  if (!out || !devices || n_devices == 0 || n_devices > 16) {  // (the partition kernels of shard.hip take at most 16 ranks)
    rvn::set_last_error("[raven_hip] rvn_group_create: invalid argument (1 .. 16 devices)");
    return RVN_EINVAL;
  }
  *out = nullptr;
  std::unique_ptr<rvn_group> g(new rvn_group());
  for (uint32_t i = 0; i < n_devices; ++i) {
    rvn_engine* e = nullptr;
    const int rc = rvn_engine_create(&e, k, w, bandwidth, chain, matches, gap, devices[i]);
    if (rc != RVN_OK) {
      for (rvn_engine* x : g->eng) rvn_engine_destroy(x);
      return rc;
    }
    g->eng.push_back(e);
    g->dev.push_back(devices[i]);
  }
  for (uint32_t i = 0; i < n_devices; ++i) {
    hipStream_t s = nullptr;
    if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
      for (rvn_engine* x : g->eng) rvn_engine_destroy(x);
      for (size_t c = 0; c < g->copy_stream.size(); ++c)  // the streams created so far
        if (hipSetDevice(devices[c]) == hipSuccess) (void)hipStreamDestroy(g->copy_stream[c]);
      rvn::set_last_error("[raven_hip] rvn_group_create: cannot create a copy stream");
      return RVN_EHIP;
    }
    g->copy_stream.push_back(s);
    // Direct peer access where the devices differ.  A pair without it still works — hipMemcpyPeerAsync then stages the
    // copy through host memory — but at PCIe speed instead of xGMI's: the group records which pairs are direct
    // (rvn_group_peer_access) so that a caller / a bench line can say which it measured.
    for (uint32_t j = 0; j < n_devices; ++j) {
      int direct = 1;
      if (devices[j] != devices[i]) {
        int can = 0;
        direct = (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) ? 1 : 0;
        if (direct) {
          const hipError_t pe = hipDeviceEnablePeerAccess(devices[j], 0);
          if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) direct = 0;
        }
        (void)hipGetLastError();
      }
      g->peer_direct.push_back(static_cast<uint8_t>(direct));
    }
  }
  (void)hipGetLastError();
  g->slot.resize(n_devices);
  *out = g.release();
  return RVN_OK;
}

void rvn_group_destroy(rvn_group* g) {
  if (!g) return;
  for (size_t i = 0; i < g->copy_stream.size(); ++i) {
    (void)hipSetDevice(g->dev[i]);
    (void)hipStreamDestroy(g->copy_stream[i]);
  }
  for (rvn_engine* e : g->eng) rvn_engine_destroy(e);
  delete g;
}

uint32_t rvn_group_size(const rvn_group* g) { return g ? g->world() : 0; }

int rvn_group_peer_access(const rvn_group* g, uint8_t* direct) {
  if (!g) return RVN_EINVAL;
  const size_t n = static_cast<size_t>(g->world()) * g->world();
  int all = 1;
  for (size_t i = 0; i < n && i < g->peer_direct.size(); ++i) {
    if (direct) direct[i] = g->peer_direct[i];
    all = all && g->peer_direct[i];
  }
  return all ? 1 : 0;
}

rvn_engine* rvn_group_engine(rvn_group* g, uint32_t rank) { return (g && rank < g->world()) ? g->eng[rank] : nullptr; }

int rvn_group_find_overlaps_and_create_piles_batched(rvn_group* g, const uint64_t* packed, const uint64_t* word_offsets,
                                                     const uint32_t* lengths, uint32_t n_reads, double freq, uint32_t kmax,
                                                     int use_minhash, uint64_t index_batch_bases, uint64_t flush_bases,
                                                     uint32_t* bounds, rvn_pass1** out) {
  if (!g || !packed || !word_offsets || !lengths || !bounds || !out || n_reads == 0) {
    rvn::set_last_error("[raven_hip] rvn_group_find_overlaps_and_create_piles: invalid argument");
    return RVN_EINVAL;
  }
  const std::vector<u32> b = partition_reads(lengths, n_reads, g->world());
  for (u32 h = 0; h <= g->world(); ++h) bounds[h] = b[h];
  for (u32 h = 0; h < g->world(); ++h) out[h] = nullptr;
  PassArgs A{packed, word_offsets, lengths, n_reads, freq, kmax, use_minhash, flush_bases ? flush_bases : (1ULL << 30),
             index_batch_bases ? index_batch_bases : (1ULL << 32), &b, out};
  const int rc = run_ranks(g, [&](Rank& R) { pass_rank(R, A); });
  if (rc != RVN_OK)
    for (u32 h = 0; h < g->world(); ++h) {
      rvn_pass1_destroy(out[h]);
      out[h] = nullptr;
    }
  return rc;
}

int rvn_group_find_overlaps_and_create_piles(rvn_group* g, const uint64_t* packed, const uint64_t* word_offsets,
                                             const uint32_t* lengths, uint32_t n_reads, double freq, uint32_t kmax,
                                             int use_minhash, uint64_t flush_bases, uint32_t* bounds, rvn_pass1** out) {
  return rvn_group_find_overlaps_and_create_piles_batched(g, packed, word_offsets, lengths, n_reads, freq, kmax, use_minhash,
                                                          1ULL << 32, flush_bases, bounds, out);
}

int rvn_group_polish_round_q(rvn_group* g, const uint64_t* t_packed, const uint64_t* t_word_offsets, const uint32_t* t_lengths,
                             uint32_t n_targets, const uint64_t* r_packed, const uint64_t* r_word_offsets,
                             const uint32_t* r_lengths, uint32_t n_reads, const uint8_t* r_quals, const uint64_t* r_qual_offsets,
                             int qual_block_shift, double q, double err, uint32_t w, int trim, int match, int mismatch, int gap,
                             uint8_t* out_codes, const uint64_t* out_offsets, uint32_t* out_len, double* ratio) {
  if (!g || !t_packed || !t_word_offsets || !t_lengths || !r_packed || !r_word_offsets || !r_lengths || !out_codes ||
      !out_offsets || !out_len || !ratio || n_targets == 0 || w == 0 || (r_quals && !r_qual_offsets)) {
    rvn::set_last_error("[raven_hip] rvn_group_polish_round: invalid argument");
    return RVN_EINVAL;
  }
  PolishArgs A{t_packed, t_word_offsets, t_lengths, n_targets, r_packed, r_word_offsets, r_lengths, n_reads, q,  err,
               w,        trim,           match,     mismatch,  gap,      r_quals,        r_qual_offsets, qual_block_shift,
               out_codes, out_offsets, out_len, ratio};
  return run_ranks(g, [&](Rank& R) { polish_rank(R, A); });
}

int rvn_group_polish_round(rvn_group* g, const uint64_t* t_packed, const uint64_t* t_word_offsets, const uint32_t* t_lengths,
                           uint32_t n_targets, const uint64_t* r_packed, const uint64_t* r_word_offsets,
                           const uint32_t* r_lengths, uint32_t n_reads, double q, double err, uint32_t w, int trim, int match,
                           int mismatch, int gap, uint8_t* out_codes, const uint64_t* out_offsets, uint32_t* out_len,
                           double* ratio) {
  return rvn_group_polish_round_q(g, t_packed, t_word_offsets, t_lengths, n_targets, r_packed, r_word_offsets, r_lengths, n_reads,
                                  nullptr, nullptr, 0, q, err, w, trim, match, mismatch, gap, out_codes, out_offsets, out_len, ratio);
}

}  // extern "C"
