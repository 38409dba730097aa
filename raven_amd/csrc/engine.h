// engine.h — internal structures of the MI355X overlap engine (not part of the public C ABI).
#pragma once

#include <string>
#include <atomic>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"

namespace rvn {

constexpr int kSketchTile = 1024;  // k-mer positions per sketch workgroup
constexpr int kMaxWindow = 256;    // largest supported winnowing window w

// Device-resident read set: concatenated 2-bit packed words (every read starts on a word
// boundary, one pad word at the end), per-read word offsets / lengths / ids.
inline u64 next_reads_serial() {
  static std::atomic<u64> counter{0};
  return ++counter;
}

struct ReadsDev {
  u32 n = 0;
  u64 total_bases = 0;
  u64 n_words = 0;
  DevBuf packed;    // u64[n_words + 1]
  DevBuf word_off;  // u64[n + 1]
  DevBuf len;       // u32[n]
  DevBuf id;        // u32[n]
  std::vector<u64> h_word_off;
  std::vector<u32> h_len, h_id;
  // qualities attached to the read set (rvn_reads_attach_quality): Phred+33, one byte per 2^qual_shift bases,
  // read i at qual_off[i]; used by the polishing rounds when the caller passes none
  DevBuf quals, qual_off;
  std::vector<u64> h_qual_off;
  int qual_shift = -1;  // -1: none attached
  u64 serial = next_reads_serial();  // unique per read set of the process (what a cache of derived data is keyed by)
  bool ids_are_indices = false;  // ids[i] == i and all < 2^31 (needed by the pass-1 merge and the self-join)
  // sketch tiles for the owning engine's (k, w)
  u32 n_tiles = 0;
  DevBuf tile_read;      // u32[n_tiles]  read index of the tile
  DevBuf tile_start;     // u32[n_tiles]  first k-mer position of the tile
  DevBuf read_tile_off;  // u32[n + 1]
  std::vector<u32> h_read_tile_off;
};

// Sketch of a read range: minimizers in (read, position) order.
struct Sketch {
  u32 first = 0, last = 0;
  u64 count = 0;
  DevBuf val;       // V[count]   (u32 when 2k < 32, else u64)
  DevBuf org;       // u64[count] id << 32 | pos << 1 | strand
  DevBuf read_off;  // u32[last - first + 1]
};

struct Index {
  u64 m = 0;  // minimizers in the index
  u64 u = 0;  // distinct keys
  int table_bits = 0;
  int shift = 0;
  DevBuf s_val[2];  // sorted values (ping-pong)
  DevBuf s_org[2];  // sorted origins
  int cur = 0;
  DevBuf u_val;    // V[u]
  DevBuf u_start;  // u32[u + 1]
  DevBuf table;    // u32[2^table_bits + 1]
  DevBuf direct;   // u64[4^k]: (count << 32) | first entry of value v's run, 0 = v is not in the index — a probe is ONE
                   // cache line instead of table -> search in u_val -> u_start (built for 2k <= 30 bits and large indexes only)
  bool direct_built = false;
  u32 occurrence = 0xFFFFFFFFu;
  bool table_built = false;      // u_val / table are built lazily (only the probe path needs them)
  bool has_query_flags = false;  // origins carry kQueryFlag
  bool all_query = false;        // every index entry is a query minimizer (index built with minhash)
  u32 first = 0, last = 0;       // read range the index was built from
};

struct MapOut {
  u32 first = 0, last = 0;
  u64 n_query = 0;    // M_q
  u64 n_matches = 0;  // H
  u64 n_intervals = 0;
  u64 n_overlaps = 0;  // O
  DevBuf ovl;          // Overlap[n_overlaps] in (query read, emission) order
  DevBuf ovl_read_off; // u32[last - first + 1]
  DevBuf filtered;     // u8[n_query] (1 = skipped by the occurrence filter); valid when requested
  // chain anchors of every overlap (lhs_pos << 32 | rhs_pos, ascending along the chain); when requested
  bool has_anchors = false;
  DevBuf anchors;      // u64[n_matches] (sparse: regions of the emitting intervals)
  DevBuf anchor_off;   // u64[n_overlaps] index of an overlap's first anchor in `anchors`
  DevBuf anchor_cnt;   // u32[n_overlaps]
};

struct StageTimes {
  // accumulated device milliseconds per stage (HIP events on the engine stream)
  enum { kSketch, kSort, kIndex, kFilter, kQuery, kMatch, kSegSort, kIntervals, kChain, kCompact, kMerge, kPile,
         kTruncate, kNum };
  double ms[kNum] = {};
  u64 launches[kNum] = {};
};

struct PileState;

// Tuning a deployment may set (rvn_engine_set_option; 0 = the built-in default everywhere).  None of them changes a result.
struct EngineOptions {
  long long nw_budget_mb = 0;        // alignment-path stage: HBM for the stored band words (default: a quarter of the free memory, <= 64 GB)
  long long nw_group_walk = 0;       // alignment-path stage: 1 = every walk one lane per alignment, 2 = every walk a group of lanes per
                                     // alignment (nwtrace.h), 3 = one lane per alignment with strips of sixteen kept columns, otherwise
                                     // by the number of alignments in the launch.  Same records either way
  long long index_direct_min_keys = 0;  // index: distinct values from which every possible value is addressed directly (index.hip; default 8 M,
                                     // 1 = every index with 2k <= 30 bits — the tests of that path on small inputs).  Same matches either way
  long long poa_rows_min_windows = -1;  // window-consensus stage: smallest batch that starts with the rows-on-lanes kernel (poa4.hip);
                                        // a smaller one starts with the 64-column kernel (poa2.hip).  < 0: the default, kPoaRowsMinWindowsDefault = 8 192
  long long io_threads = 0;          // rvn_reads_load: inflate threads (default min(32, cores - 2))
  long long io_slab_mb = 0;          // ... page-locked slab size (default 8)
  long long io_ring = 0;             // ... slabs in flight (default 8)
  long long io_zlib = 0;             // ... != 0: zlib instead of inflate_fast.h on a single gzip member
  long long arena_mb = 0;            // device arena (common.h: devpool): its size when it starts (default: free memory - margin)
  long long arena_margin_mb = 0;     // ... memory left to the driver (default max(12 GB, 1/16 of the device))
  long long no_arena = 0;            // ... != 0: never start one
  long long release_always = 0;      // != 0: every stage entry hands the scratch back (the tests of that path)
  long long polish_join = 0;         // != 0: a polishing round maps by sorting the reads' minimizers with the targets' and streaming
                                     // the runs instead of probing the targets' index (round 6: built, bit-identical, slower — DESIGN.md 3.7)
  long long polish_sketch_cache_mb = -1;  // HBM for the reads' sketch kept between polishing rounds (< 0: an eighth of the device; 0: none)
};
const char* engine_option_names();   // comma-separated, for the error message
long long* engine_option(EngineOptions& o, const char* name);

struct Engine {
  EngineOptions opt;
  // Every C-ABI entry point that touches the engine's state (scratch buffers, stream, last Map result) holds this
  // lock for its whole duration: ram::MinimizerEngine::Map is const and called concurrently from Raven's pool workers
  // (RavenLib/src/construct.cc:60-64, :373-381), so the boundary has to be safe under concurrent callers.
  std::recursive_mutex mu;
  u32 k, w, bandwidth, chain, matches, gap;
  int device = 0;
  bool val64 = false;  // true when minimizer values need 64 bits
  hipStream_t stream = nullptr;
  Index index;
  Sketch index_sketch, query_sketch, raw_sketch;
  // query sketch prepared ahead of map_batch (valid for exactly this range / minhash flag)
  bool query_ready = false;
  u32 query_ready_first = 0, query_ready_last = 0;
  bool query_ready_minhash = false;
  u64 join_query_count = 0;  // number of query minimizers flagged in the index (self-join path)
  bool shard_sketch_minhash = false;           // which sketch rvn_shard_sketch left its result in
  u32 shard_join_reads = 0;                    // rvn_shard_join: segments of the last join
  u64 shard_join_matches = 0;
  MapOut map_out;
  // scratch
  DevBuf tmp_a, tmp_b, tmp_c, tmp_d, tmp_e, tmp_f, scan_tmp, sort_tmp;
  DevBuf sh_hist, sh_off, sh_ptrs;  // shard.hip: tile histograms / offsets / pointer tables of the partition steps
  DevBuf q_start, q_cnt, m_off;
  // the reads' sketch of a polishing round's mapping, kept for the next round (the reads do not change between rounds; only
  // the targets do): per read batch, for ONE read set at a time (`owner` = ReadsDev::serial)
  struct PolishSketch {
    u32 first = 0, last = 0;
    Sketch sk;
  };
  u64 polish_sketch_owner = 0;
  std::vector<std::unique_ptr<PolishSketch>> polish_sketches;
  DevBuf pl_tval, pl_torg;          // the targets' minimizers of a polishing round, appended to every read batch's (polish.hip)
  DevBuf foreign_val, foreign_org;  // a query-only sketch appended from its pieces (rvn_shard_sketch_range)
  DevBuf sketch_sum;  // 64-bit total of a sketch whose 32-bit offsets could wrap (sketch.hip)
  DevBuf m_grp[2], m_pos[2];
  DevBuf seg_off, iv_slot_begin, iv_slot_end, iv_cnt, iv_off, iv_begin, iv_end;
  DevBuf lis_min, lis_pred, lis_tail, lis_mask, ovl_slots, ovl_flags, ovl_scan, chain_big;
  DevBuf poa_scratch, poa2_scratch, polish_quals;
  DevBuf ed_cnt, ed_sort, ed_todo;
  // second pass / identity filters (pass2.hip)
  DevBuf p2_slot, p2_pairs, p2_dist, p2_regions, p2_index_of, p2_kmers_off, p2_ok, p2_keep, p2_tmp_ovl;
  DevBuf io_text[2];  // input path: the file's text in HBM (io.hip)
  int stage_kind = 0;   // the stage entry point running (engine_release_scratch_if_tight)
  u32 oom_mask = 0;     // kinds of stages that ran out of device memory once: they start from released scratch
  std::vector<std::pair<std::unique_ptr<PinBuf>, bool>> io_pin;  // its page-locked slabs (buffer, handed out)
  DevBuf poa_sched, poa_redo_w, poa_redo_i;  // LPT order / escalation lists of a POA batch (poa_run_dev)
  // alignment-path stage of a polishing round (nwpath.hip): stored band words + scores, jobs, results
  DevBuf nw_hs, nw_ck, nw_hs2, nw_ck2, nw_hs3, nw_ck3, nw_hs4, nw_ck4, nw_strip, nw_jobs, nw_res;  // alignment paths: horizontal-delta streams, checkpoints, jobs, results
  double nw_rate = -1.0;  // running estimate of edit distance / length of the read-to-target alignments (< 0: unknown)
  // polishing front end (polish.hip): best overlaps, window records, layer tables, consensus
  DevBuf pl_best, pl_best_t, pl_idmap, pl_recs, pl_keep, pl_win_cnt, pl_win_off, pl_win_fill, pl_win_meta, pl_first_window,
      pl_keys, pl_lays_tmp, pl_lays, pl_wins, pl_out, pl_len, pl_status, pl_ok, pl_cons_off, pl_final, pl_qual_off, pl_misc;
  // the last COMPLETE polishing round's stitched consensus is still in pl_final: byte offsets of the targets' sequences there
  // (rvn_polish_output_as_reads: the next round's targets without the way over the host)
  std::vector<u64> pl_last_off;
  bool pl_last_valid = false;
  std::vector<u32> polish_target_reads;  // reads used per target in the last polishing round
  // best-overlap table for the NEXT polishing round (rvn_polish_set_best; consumed by that round)
  std::vector<Overlap> polish_given_best;
  std::vector<u32> polish_given_best_t;
  bool polish_given_valid = false;
  // layer table of the last polishing round (still in pl_wins / pl_lays / pl_ok), for rvn_polish_fetch_layers
  u32 polish_last_windows = 0;
  u64 polish_last_layers = 0, polish_last_w0 = 0;
  bool polish_last_has_ok = false;
  std::vector<u64> polish_last_read_off;
  int poa_mode = 0;  // 0 banded 32 (poa4.hip) -> 64 -> 128 -> 256 (poa2.hip) -> full matrix; 1 full matrix only; 2 / 3 / 4 band 64 / 128 / 256 only (tests); 9 poa4.hip only (band 32, rows on lanes)
  u32 poa_fallback_windows = 0;  // windows of the last batch that needed more than the 128-column band
  u32 poa_fullmatrix_windows = 0;  // ... of which re-run by the full-matrix kernel
  u32 poa_wide_windows = 0;      // windows of the last batch re-run with the 128-column band
  u32 poa_narrow_windows = 0;    // windows of the 32-column first attempt (poa4.hip) re-run with the 64-column band
  DevBuf anc_slot_off, anc_slot_cnt;
  bool keep_anchors = false;  // map_batch also returns the chain anchors of every overlap
  unsigned long long poa_phase_cycles[8] = {};  // subgraph, dp, traceback, add, order, consensus (last call); [6], [7]: DP cells
  // DP cells of the banded POA kernel since the last reset_stats: full-matrix equivalent (graph rows x layer length of
  // every layer alignment: what spoa computes) / inside the computed band; number of batches
  u64 poa_cells_full = 0, poa_cells_band = 0, poa_calls = 0;
  StageTimes times;
  KernelTimers ktimers;
  // counters for algorithmic bytes
  u64 c_index_bases = 0, c_index_min = 0, c_index_keys = 0, c_query_bases = 0, c_query_min = 0, c_matches = 0,
      c_overlaps = 0;
  u64 c_intervals = 0;
  bool timing = true;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t nw_streams[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // alignment-path stage: [0..3] walk streams (beside the sweeps), one per buffer set; [4] uploads of a pass planned while another one sweeps
  hipEvent_t nw_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // [0..3] the walk of a buffer set is done, [4] sweep -> walk
  hipStream_t nw_side[3] = {nullptr, nullptr, nullptr};  // ... sweep launches of few waves (the pilot's, the several-blocks-per-lane variants of the longest alignments) beside the main stream's
  hipEvent_t nw_side_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // [0..2] the side stream's sweeps are done, [3] main -> side
  u64* h_pin = nullptr;  // pinned host scratch for small device->host size read-backs
  PinBuf pin_big;        // pinned staging for bulk read-backs up to 256 MB (polishing: chain anchors)
  HostBuf host_big;      // ... and the unpinned one for larger ones
  PinBuf pin_out;        // pinned consensus buffer of the POA chunk in flight
  u64 polish_chunk_windows = 16384;  // windows per POA chunk of a polishing round (0 = everything in one batch)
  PileState* pile_pool = nullptr;  // buffers of the last destroyed pass, adopted by the next one (engine.hip)
  std::shared_ptr<int> life = std::make_shared<int>(0);  // lets handles notice that their engine is gone
};

// Hands every scratch / intermediate buffer of the engine back to the allocator (index, sketches, last Map result,
// stage scratch).  Buffers only ever grow, so a stage with a very different footprint (HiFi first pass -> polishing)
// can otherwise find the HBM full of the previous stage's scratch.  Only valid between stages: nothing of the
// released state may be needed afterwards (every stage entry point rebuilds what it uses).
void engine_release_scratch(Engine& e);
// ... when less than a third of the device memory is free (called at stage entry points)
void engine_release_scratch_if_tight(Engine& e, int stage_kind);  // kind: 0 pass 1, 1 pass 2, 2 polishing round, 3 polishing map

// Reads one 4- or 8-byte value from the device through pinned memory (stream-ordered, then synchronises).
inline u64 read_back(Engine& e, const void* dptr, size_t bytes) {
  e.h_pin[0] = 0;
  RVN_HIP(hipMemcpyAsync(e.h_pin, dptr, bytes, hipMemcpyDeviceToHost, e.stream));
  RVN_HIP(rvn_stream_sync(e.stream));
  return e.h_pin[0];
}

// Stage timing helper: records HIP events on the engine stream around a stage.
struct StageTimer {
  Engine& e;
  int stage;
  StageTimer(Engine& eng, int st) : e(eng), stage(st) {
    if (e.timing) RVN_HIP(hipEventRecord(e.ev0, e.stream));
  }
  void stop() {
    if (!e.timing) return;
    RVN_HIP(hipEventRecord(e.ev1, e.stream));
    RVN_HIP(hipEventSynchronize(e.ev1));
    float ms = 0;
    RVN_HIP(hipEventElapsedTime(&ms, e.ev0, e.ev1));
    e.times.ms[stage] += ms;
    e.times.launches[stage] += 1;
  }
};

// input path (io.hip)
struct LoadStats {
  u64 n_sequences = 0, n_bases = 0;
  int has_quality = 0;
  double parse_s = 0, device_s = 0, total_s = 0;  // record scanner | copies + packing on the device (caller thread) | whole call
  u32 inflate_threads = 0, members = 0;           // the inflate pool and what it found in the archive
  int streaming = 0, restarted = 0;               // one stream front to back | a wrong member cut made the load start over
};
void reads_load(Engine& e, const std::string& path, ReadsDev& R, std::vector<std::string>& names, LoadStats& st);

// ---- stages (one translation unit each) -------------------------------------
void reads_build_tiles(Engine& e, ReadsDev& r);
// 2-bit packing of one-byte codes already in HBM: read i = codes[base_off[i] ..), words at word_off[i] (sketch.hip)
void pack_codes_on_device(Engine& e, const u8* d_codes, const u64* d_base_off, const u64* d_word_off, u32 n_reads,
                          u64 n_words, u64* d_packed);
void sketch_raw(Engine& e, const ReadsDev& r, u32 first, u32 last, Sketch& out);
void sketch_minhash(Engine& e, const ReadsDev& r, const Sketch& raw, Sketch& out);
void sketch_range(Engine& e, const ReadsDev& r, u32 first, u32 last, bool minhash, Sketch& out);
void index_build(Engine& e, Sketch& sk, bool build_table = true);  // consumes sk.val/sk.org
void index_build_table(Engine& e);                               // lazy: distinct keys + direct-address table
// minhash-select on a raw sketch, marking the selected minimizers with kQueryFlag in raw.org; returns their count
u64 sketch_flag_queries(Engine& e, const ReadsDev& r, Sketch& raw);
void index_filter(Engine& e, double freq);               // sets e.index.occurrence
void index_key_histogram(Engine& e, std::vector<u64>& hist, std::vector<u32>& over);  // count-of-counts (65536 bins)
void map_batch(Engine& e, const ReadsDev& r, u32 first, u32 last, bool avoid_equal, bool avoid_symmetric,
               bool minhash, bool want_filtered, MapOut& out);

// self-join of the index for global query ids 0..n_reads-1 -> e.m_grp[0] / e.m_pos[0] / e.seg_off (map.hip)
void map_batch_query_only(Engine& e, const ReadsDev& r, u32 first, u32 last, u64 n_query, MapOut& out);
u64 join_index_matches(Engine& e, u32 n_reads, bool avoid_equal, bool avoid_symmetric, u32 q_lo = 0,
                       u32 q_hi = 0xFFFFFFFFu);  // only query reads with q_lo <= id < q_hi
// chain stage of Map on matches already in e.m_grp[0] / e.m_pos[0] / e.seg_off (map.hip)
void chain_matches(Engine& e, const ReadsDev& r, u32 first, u32 last, u64 H, MapOut& out);

// Batched exact edit distance (edit_distance.hip). h_pairs: n_pairs x {a_idx,a_begin,a_len,b_idx,b_begin,b_len,strand,0}
void edit_distance_batch(Engine& e, const ReadsDev& r, const u32* h_pairs, u32 n_pairs, u32* h_out, double* kernel_ms,
                         u64* cells);
// the same with pairs and distances resident in HBM (pass2.hip: identity filters)
void edit_distance_dev(Engine& e, const ReadsDev& r, const u32* d_pairs, u32 n_pairs, u32* d_out, const u32* d_kmax = nullptr);

// Batched POA window consensus (poa.hip); all arrays are host pointers, see rvn_poa_consensus_batch
void poa_consensus_batch(Engine& e, const u8* h_codes, const u8* h_quals, const u64* h_layer_off, const u32* h_begins,
                         const u32* h_ends, const u32* h_has_qual, const u32* h_win_off, u32 n_windows, int m, int n,
                         int g, int trim, u8* h_out, const u64* h_out_off, u32* h_out_len, u32* h_status,
                         double* device_ms);

// poa4.hip's phase functions stepped through on the host (wavefront emulator): see rvn_poa_banded_emulate
void poa_banded_emulate(const u8* h_codes, const u8* h_quals, const u64* h_layer_off, const u32* h_begins,
                        const u32* h_ends, const u32* h_has_qual, const u32* h_win_off, u32 n_windows, int m, int n, int g,
                        int trim, u8* h_out, const u64* h_out_off, u32* h_out_len, u32* h_status, int variant);

struct PolishStats {
  u64 n_overlaps = 0, n_reads_used = 0, n_layers = 0, n_windows = 0, n_polished_windows = 0, n_failed_windows = 0;
  u64 n_dropped_layers = 0;  // reads whose alignment is beyond the path kernel (band threshold > ~32 000): not used
  double poa_ms = 0;                            // device time of the POA batch
  double map_ms = 0, host_ms = 0, total_ms = 0;  // wall: index + map | host planning (jobs, window tables) | all
  double align_ms = 0;                           // device time of the alignment-path stage (forward + traceback)
  u64 n_aligned = 0, n_align_retries = 0, align_band_cells = 0, align_store_bytes = 0;
};

// alignment-path stage (nwpath.hip)
struct NwJob;
struct NwWindowRec;
struct NwStats {
  u64 n_aligned = 0, n_retries = 0, n_unaligned = 0, n_batches = 0;
  u64 band_cells = 0, sum_distance = 0, store_bytes = 0;
  double ms = 0;
};
void nw_breakpoints(Engine& e, const ReadsDev& T, const ReadsDev& R, std::vector<NwJob>& jobs, u32 w, NwWindowRec* d_recs,
                    u64 n_recs, NwStats& st);
// the same code stepped on the CPU (64 emulated lanes): test hook, see rvn_test_nw_breakpoints
int nw_breakpoints_host(const u64* t_words, u32 t_len, const u64* r_words, u32 r_len, u32 t_begin, u32 n, u32 q_begin, u32 m,
                        int rc, u32 w, u32 k, int force_R, NwWindowRec* recs, u32* distance, u32* band);
// One racon polishing round (polish.hip): targets T, reads R, optional per-base Phred+33 qualities of the reads
// shard.hip — partition / regroup steps of the sharded pass, all pointers device pointers unless noted
void shard_split_minimizers(Engine& e, const u64* d_val, const u64* d_org, u64 n, u32 world, u64* d_val_out, u64* d_org_out,
                            u64* counts /* host [world] */);
void shard_split_overlaps(Engine& e, const Overlap* d_ovl, u64 n, const u32* bounds /* host [world + 1] */, u32 world, u32 self,
                          Overlap* d_out, u64* counts /* host [world + 1], [world] = overlaps that stay */);
u64 shard_count_flagged(Engine& e, const u64* d_org, u64 n);
void shard_adjacent_diff(Engine& e, const u64* d_seg, u64 n, u64* d_cnt);
void shard_regroup(Engine& e, u32 world, const u64* const* d_cnt, const u64* const* d_grp, const u64* const* d_pos,
                   const u64* n_src /* host */, u32 n_reads, u64* d_seg, u64* d_grp_out, u64* d_pos_out);
void shard_lhs_offsets(Engine& e, const Overlap* d_ovl, u64 n, u32 n_reads, u32* d_off);
// polish_round's consensus straight into the caller's buffer (target t at out + off[t], at most off[t + 1] - off[t] bytes;
// len[t] = its length) instead of into `polished` (which then stays empty): one pass over the 100 MB of a C4 round less
struct PolishDirectOut {
  u8* out;
  const u64* off;
  u64* len;
};
void polish_map_best(Engine& e, ReadsDev& T, ReadsDev& R, u32 r_first, u32 r_last, double err_thr,
                     std::vector<Overlap>& best, std::vector<u32>& best_t, u64* n_overlaps);
void polish_round(Engine& e, ReadsDev& T, ReadsDev& R, const u8* h_quals, const u64* h_qual_off, double q_thr,
                  double err_thr, u32 w, bool trim, int m, int n, int g, std::vector<std::vector<u8>>& polished,
                  std::vector<double>& ratio, PolishStats& stats, u64 win_first = 0, u64 win_last = ~0ULL,
                  std::vector<u32>* win_count = nullptr, std::vector<u32>* win_polished = nullptr,
                  const PolishDirectOut* direct = nullptr);

// Result of the second mapping pass (pass2.hip), resident in HBM
struct Pass2State {
  u32 n = 0;
  u64 n_overlaps = 0;
  DevBuf ovl;        // Overlap[n_overlaps]: overlaps.back() of construct.cc:352,451
  DevBuf contained;  // u8[n]: piles this pass marked as contained
  DevBuf kmers;      // Pile::kmers_ cells: (len >> 4) + 1 bytes per VALID read at h_kmers_off[id] (0 bytes for invalid ones)
  std::vector<u64> h_kmers_off;
  u64 kmers_total = 0;
};
// ram::MinimizerEngine::Minimize(first, last, minhash) on a read set (engine.hip)
void engine_minimize(Engine& e, const ReadsDev& r, u32 first, u32 last, bool minhash);
void reads_subset(Engine& e, const ReadsDev& R, const std::vector<u32>& src, ReadsDev& V);
void second_pass(Engine& e, const ReadsDev& R, const u32* h_begin, const u32* h_end, const u8* h_invalid, double freq,
                 u32 kmer_len, double identity, u64 batch_bases, Pass2State& out);
void identity_filter_lists(Engine& e, const ReadsDev& R, Overlap* h_ovl, u32* h_off, const u32* h_begin, const u32* h_end,
                           const u8* h_invalid, double identity);

// Pass-1 state: per-pile kept overlaps + coverage (pile.hip)
struct PileState {
  u32 n = 0;
  DevBuf pile_off;   // u64[n + 1] in u16 units
  DevBuf pile_data;  // u16[total]
  u64 pile_words = 0;
  DevBuf kept_off;   // u32[n + 1]
  DevBuf kept;       // Overlap[kept_total]
  u64 kept_total = 0;
  DevBuf new_off, new_list, tmp1, tmp2, tmp3, tmp4, tmp5, tmp6;
};
void piles_init(Engine& e, const ReadsDev& r, PileState& ps);
void piles_merge(Engine& e, const ReadsDev& r, const MapOut& mo, u32 kmax, PileState& ps);
// Pile::FindValidRegion(coverage) + FindMedian on every pile, in place in HBM (pile.hip); host output arrays of n
void piles_trim_and_median(Engine& e, PileState& ps, u32 coverage, u32* h_begin, u32* h_end, u16* h_median, u8* h_invalid);
// Pile::FindChimericRegions of every valid pile on the coverage in HBM (pile.hip): CSR of (begin, end) cell pairs
void piles_find_chimeric_regions(Engine& e, PileState& ps, const u8* h_invalid, std::vector<u32>& h_off,
                                 std::vector<u32>& h_regions);
// Pile::AddKmers for reads [first_read, first_read + n_reads) (pile.hip)
void pile_add_kmers_batch(Engine& e, const ReadsDev& r, const u32* h_pos, const u64* h_pos_off, u32 n_reads,
                          u32 first_read, u8* h_out, const u64* h_out_off);
// Pile::AddLayers on a single pile (ps initialised for one read); h_ovl is a host array
void pile_add_layers_single(Engine& e, PileState& ps, const u32* d_ids, const Overlap* h_ovl, u32 n);

}  // namespace rvn
