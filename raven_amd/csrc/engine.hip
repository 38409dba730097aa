// engine.hip — host orchestration + the C ABI declared in include/raven_hip.h.
#include "engine.h"

#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <unordered_map>

#include "../../include/raven_hip.h"
#ifdef RVN_TEST_HOOKS
#include "../../include/raven_hip_test.h"
#endif
#include "introsort.h"
#include "nwpath.h"
#include "overlap_rules.h"
#include "slopes.h"
#include "poa.h"
#include "kmer.h"
#include "lowcomplexity.h"
#include "freelist.h"
#ifdef RVN_TEST_HOOKS
#include "io_text.h"
#include "inflate_fast.h"
#endif

using namespace rvn;

static_assert(sizeof(rvn_overlap) == sizeof(rvn::Overlap), "overlap layout");

struct rvn_engine {
  Engine e;
};
struct rvn_reads {
  ReadsDev r;
  std::vector<std::string> names;  // rvn_reads_load: the sequences' names
};
// The pile buffers (coverage, kept lists, merge scratch: a dozen allocations) are recycled through the engine:
// destroying a pass hands them back, the next pass adopts them, so steady-state passes do not touch the allocator.
struct rvn_pass1 {
  Engine* e = nullptr;
  std::unique_ptr<PileState> state;
  PileState& ps;
  std::weak_ptr<int> engine_life;  // a handle may outlive its engine (e.g. interpreter teardown order)
  std::unique_ptr<ReadsDev> meta;  // sharded pass: lengths / ids of ALL reads (piles need no bases)
  explicit rvn_pass1(Engine& eng)
      : e(&eng), state(eng.pile_pool ? eng.pile_pool : new PileState()), ps(*state), engine_life(eng.life) {
    eng.pile_pool = nullptr;
  }
  ~rvn_pass1() {
    if (!engine_life.expired() && !e->pile_pool) e->pile_pool = state.release();
  }
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

}  // namespace

namespace rvn {
void set_last_error(const std::string& msg) { g_err = msg; }  // group.hip: errors of its worker threads, to the caller
}  // namespace rvn

namespace {

template <typename F>
int guarded(F f) {
  try {
    return f();
  } catch (const HipError& ex) {
    return fail(RVN_EHIP, ex.what());
  } catch (const std::bad_alloc&) {
    return fail(RVN_ENOMEM, "[raven_hip] out of host memory");
  } catch (const std::invalid_argument& ex) {
    return fail(RVN_EINVAL, ex.what());
  } catch (const std::exception& ex) {
    return fail(RVN_EHIP, ex.what());
  }
}

// same, holding the engine's lock for the whole call (nullptr: the lambda reports the NULL handle itself)
// same, holding the engine's lock for the whole call (nullptr: the lambda reports the NULL handle itself).  A stage
// that runs out of DEVICE memory is run once more after every scratch buffer of the engine (the other phase's included)
// and every parked block went back to the driver: the entry points are functions of their arguments, a stage that
// failed half-way leaves nothing behind but scratch.
template <typename F>
int guarded(Engine* e, F f) {
  if (!e) return guarded(f);
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  try {
    return f();
  } catch (const DeviceOutOfMemory& ex) {
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
    if (knob("RVN_DEBUG_MEM")) std::fprintf(stderr, "[raven_hip] %s: all scratch back to the driver, stage repeated\n", ex.what());
    e->oom_mask |= 1u << (e->stage_kind & 31);  // next time this kind of stage starts from released scratch
    try {
      rvn::engine_release_scratch(*e);
    } catch (const std::exception& ex2) {
      return fail(RVN_EHIP, ex2.what());
    }
  } catch (const HipError& ex) {
    return fail(RVN_EHIP, ex.what());
  } catch (const std::bad_alloc&) {
    return fail(RVN_ENOMEM, "[raven_hip] out of host memory");
  } catch (const std::invalid_argument& ex) {
    return fail(RVN_EINVAL, ex.what());
  } catch (const std::exception& ex) {
    return fail(RVN_EHIP, ex.what());
  }
  return guarded(f);
}

const char* kStageNames[StageTimes::kNum] = {"sketch", "sort", "index", "filter", "query_sketch", "match",
                                             "seg_sort", "intervals", "chain", "compact", "merge", "pile",
                                             "truncate"};

struct UseTimers {
  explicit UseTimers(Engine& e) {
    e.ktimers.stream = e.stream;
    // Fold the event pairs of the previous entry point into the per-site sums now (the stream is idle between entry
    // points): the events are reused instead of growing the pool by two hipEventCreate per launch, which cost more
    // than the launches themselves on a slow host (0.5 s per pass at C4).
    if (e.ktimers.enabled && !e.ktimers.recs.empty()) {
      (void)rvn_stream_sync(e.stream);
      e.ktimers.resolve();
    }
    g_kernel_timers = &e.ktimers;
  }
  ~UseTimers() { g_kernel_timers = nullptr; }
};

int fetch_values(Engine& e, const DevBuf& val, u64 n, uint64_t* values);

void swap_bufs(DevBuf& a, DevBuf& b) {
  std::swap(a.ptr, b.ptr);
  std::swap(a.cap, b.cap);
}

// Sketch [first,last) and build the index.  With prefetch_query the minhash QUERY sketch of the same
// range (construct.cc:62 always maps with minhash=true) is derived from the same raw sketch before the
// index sort consumes it, so map_batch over that range does not sketch again.
void do_minimize(Engine& e, const ReadsDev& r, u32 first, u32 last, bool minhash, bool prefetch_query = false) {
  bool raw_handed_over = false;
  {
    StageTimer t(e, StageTimes::kSketch);
    e.query_ready = false;
    Index& ix = e.index;
    ix.has_query_flags = false;
    ix.all_query = false;
    sketch_raw(e, r, first, last, e.raw_sketch);
    const bool join = prefetch_query && r.ids_are_indices;  // map_batch will self-join instead of probing
    if (join && !minhash) {
      e.join_query_count = sketch_flag_queries(e, r, e.raw_sketch);
      ix.has_query_flags = true;
    } else if (prefetch_query && !join) {
      sketch_minhash(e, r, e.raw_sketch, e.query_sketch);
      e.query_ready = true;
      e.query_ready_first = first;
      e.query_ready_last = last;
      e.query_ready_minhash = true;
    }
    Sketch& is = e.index_sketch;
    if (join && minhash) {
      sketch_minhash(e, r, e.raw_sketch, is);
      ix.all_query = true;
    } else if (!minhash) {
      swap_bufs(is.val, e.raw_sketch.val);
      swap_bufs(is.org, e.raw_sketch.org);
      swap_bufs(is.read_off, e.raw_sketch.read_off);
      is.first = first;
      is.last = last;
      is.count = e.raw_sketch.count;
      e.raw_sketch.count = 0;
      raw_handed_over = true;
    } else if (prefetch_query) {
      const Sketch& qs = e.query_sketch;
      const size_t vb = e.val64 ? 8 : 4;
      is.first = first;
      is.last = last;
      is.count = qs.count;
      is.val.reserve((qs.count + 1) * vb);
      is.org.reserve((qs.count + 1) * 8);
      if (qs.count) {
        RVN_HIP(hipMemcpyAsync(is.val.ptr, qs.val.ptr, qs.count * vb, hipMemcpyDeviceToDevice, e.stream));
        RVN_HIP(hipMemcpyAsync(is.org.ptr, qs.org.ptr, qs.count * 8, hipMemcpyDeviceToDevice, e.stream));
      }
    } else {
      sketch_minhash(e, r, e.raw_sketch, is);
    }
    t.stop();
  }
  for (u32 i = first; i < last; ++i) e.c_index_bases += r.h_len[i];
  e.c_index_min += e.index_sketch.count;
  index_build(e, e.index_sketch, !(e.index.has_query_flags || e.index.all_query));
  e.c_index_keys += e.index.u;
  // index_build adopted the sketch's buffers and left the ones it displaced in index_sketch: they go back to the raw
  // sketch, so that TWO sets circulate (raw sketch <-> index side 0) and the second pass already finds its buffers —
  // left alone, three sets rotate through the three owners and every one of them is grown once (0.5 s at C4).
  if (raw_handed_over) {
    if (e.raw_sketch.val.cap < e.index_sketch.val.cap) swap_bufs(e.raw_sketch.val, e.index_sketch.val);
    if (e.raw_sketch.org.cap < e.index_sketch.org.cap) swap_bufs(e.raw_sketch.org, e.index_sketch.org);
  }
}

}  // namespace

namespace rvn {
namespace devpool {
namespace {
struct Arena {
  char* base = nullptr;
  FreeList list;  // freelist.h: offsets of the blocks in use and of the holes
};
constexpr int kMaxDevices = 16;
constexpr size_t kGrain = 64 << 10;
std::mutex g_mu;
Arena g_arena[kMaxDevices];
Arena* mine() {
  int d = 0;
  (void)hipGetDevice(&d);
  return (d >= 0 && d < kMaxDevices) ? &g_arena[d] : nullptr;
}
size_t offset_of(const Arena& a, const void* p) { return static_cast<size_t>(static_cast<const char*>(p) - a.base); }
bool inside(const Arena& a, const void* p) {
  return a.base && static_cast<const char*>(p) >= a.base && static_cast<const char*>(p) < a.base + a.list.size;
}
}  // namespace
bool active() {
  std::lock_guard<std::mutex> lk(g_mu);
  const Arena* a = mine();
  return a && a->base;
}
bool start(size_t bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  Arena* a = mine();
  if (!a || a->base) return a && a->base;
  bytes = bytes / kGrain * kGrain;
  if (bytes < (1ULL << 30)) return false;
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  a->base = static_cast<char*>(p);
  a->list.reset(bytes, kGrain);
  return true;
}
void* alloc(size_t bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  Arena* a = mine();
  if (!a || !a->base) return nullptr;
  size_t off = 0;
  return a->list.alloc(bytes, &off) ? a->base + off : nullptr;
}
// The arena a pointer lies in, whatever device is current (one virtual address space for all devices of the process): a
// buffer carved from device i's arena may be released while device j is current — a worker's error path, a handle
// destroyed from the main thread — and must go back to ITS arena, not be mistaken for a driver allocation.
namespace {
int owner_of(const void* p) {
  for (int d = 0; d < kMaxDevices; ++d)
    if (inside(g_arena[d], p)) return d;
  return -1;
}
}  // namespace
bool give_back(void* p) {
  int owner = -1;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    owner = owner_of(p);
    if (owner < 0 || !g_arena[owner].list.owns(offset_of(g_arena[owner], p))) return false;
  }
  // what hipFree does implicitly: nobody still reads the block when the next owner writes (the owning device's queues)
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != owner) (void)hipSetDevice(owner);
  (void)hipDeviceSynchronize();
  if (cur != owner) (void)hipSetDevice(cur);
  std::lock_guard<std::mutex> lk(g_mu);
  Arena& a = g_arena[owner];
  return inside(a, p) && a.list.release(offset_of(a, p));
}
size_t free_total() {
  std::lock_guard<std::mutex> lk(g_mu);
  const Arena* a = mine();
  return a && a->base ? a->list.free_total() : 0;
}
size_t free_largest() {
  std::lock_guard<std::mutex> lk(g_mu);
  const Arena* a = mine();
  return a && a->base ? a->list.free_largest() : 0;
}
size_t size() {
  std::lock_guard<std::mutex> lk(g_mu);
  const Arena* a = mine();
  return a && a->base ? a->list.size : 0;
}
void stop() {
  std::lock_guard<std::mutex> lk(g_mu);
  Arena* a = mine();
  if (!a || !a->base || !a->list.in_use.empty()) return;
  (void)hipFree(a->base);
  a->base = nullptr;
  a->list.reset(0, kGrain);
}
}  // namespace devpool

const char* engine_option_names() {
  return "nw_budget_mb, nw_group_walk, index_direct_min_keys, poa_rows_min_windows, io_threads, io_slab_mb, io_ring, io_zlib, arena_mb, arena_margin_mb, "
         "no_arena, release_always, polish_join, polish_sketch_cache_mb";
}
long long* engine_option(EngineOptions& o, const char* name) {
  const std::string n(name ? name : "");
  if (n == "nw_budget_mb") return &o.nw_budget_mb;
  if (n == "nw_group_walk") return &o.nw_group_walk;
  if (n == "index_direct_min_keys") return &o.index_direct_min_keys;
  if (n == "poa_rows_min_windows") return &o.poa_rows_min_windows;
  if (n == "io_threads") return &o.io_threads;
  if (n == "io_slab_mb") return &o.io_slab_mb;
  if (n == "io_ring") return &o.io_ring;
  if (n == "io_zlib") return &o.io_zlib;
  if (n == "arena_mb") return &o.arena_mb;
  if (n == "arena_margin_mb") return &o.arena_margin_mb;
  if (n == "no_arena") return &o.no_arena;
  if (n == "release_always") return &o.release_always;
  if (n == "polish_join") return &o.polish_join;
  if (n == "polish_sketch_cache_mb") return &o.polish_sketch_cache_mb;
  return nullptr;
}

void engine_release_scratch(Engine& e) {
  if (e.stream) (void)rvn_stream_sync(e.stream);
  e.query_ready = false;
  DevBuf* bufs[] = {
      &e.index.s_val[0], &e.index.s_val[1], &e.index.s_org[0], &e.index.s_org[1], &e.index.u_val, &e.index.u_start,
      &e.index.table, &e.index.direct, &e.index_sketch.val, &e.index_sketch.org, &e.index_sketch.read_off, &e.query_sketch.val,
      &e.query_sketch.org, &e.query_sketch.read_off, &e.raw_sketch.val, &e.raw_sketch.org, &e.raw_sketch.read_off,
      &e.map_out.ovl, &e.map_out.ovl_read_off, &e.map_out.filtered, &e.map_out.anchors, &e.map_out.anchor_off,
      &e.map_out.anchor_cnt, &e.tmp_a, &e.tmp_b, &e.tmp_c, &e.tmp_d, &e.tmp_e, &e.tmp_f, &e.scan_tmp, &e.sort_tmp,
      &e.q_start, &e.q_cnt, &e.m_off, &e.m_grp[0], &e.m_grp[1], &e.m_pos[0], &e.m_pos[1], &e.seg_off, &e.iv_slot_begin,
      &e.iv_slot_end, &e.iv_cnt, &e.iv_off, &e.iv_begin, &e.iv_end, &e.lis_min, &e.lis_pred, &e.lis_tail, &e.lis_mask,
      &e.ovl_slots, &e.ovl_flags, &e.ovl_scan, &e.chain_big, &e.sh_hist, &e.sh_off, &e.sh_ptrs, &e.poa_scratch, &e.poa2_scratch, &e.polish_quals, &e.ed_cnt, &e.ed_sort, &e.ed_todo, &e.p2_slot,
      &e.p2_pairs, &e.p2_dist, &e.p2_regions, &e.p2_index_of, &e.p2_kmers_off, &e.p2_ok, &e.p2_keep, &e.p2_tmp_ovl,
      &e.poa_sched, &e.poa_redo_w, &e.poa_redo_i, &e.nw_hs, &e.nw_ck, &e.nw_hs2, &e.nw_ck2, &e.nw_hs3, &e.nw_ck3, &e.nw_hs4, &e.nw_ck4, &e.nw_strip, &e.nw_jobs, &e.nw_res,
      &e.pl_best, &e.pl_best_t, &e.pl_idmap, &e.pl_recs, &e.pl_keep, &e.pl_win_cnt, &e.pl_win_off, &e.pl_win_fill,
      &e.pl_win_meta, &e.pl_first_window, &e.pl_keys, &e.pl_lays_tmp, &e.pl_lays, &e.pl_wins, &e.pl_out, &e.pl_len,
      &e.pl_status, &e.pl_ok, &e.pl_cons_off, &e.pl_final, &e.pl_qual_off, &e.pl_misc, &e.anc_slot_off, &e.anc_slot_cnt};
  for (DevBuf* b : bufs) b->release();
  e.pl_last_valid = false;
  e.polish_sketches.clear();  // (the reads' sketch kept between polishing rounds: derived data, recomputed when needed)
  e.polish_sketch_owner = 0;
  e.pl_tval.release();
  e.pl_torg.release();
  e.foreign_val.release();
  e.foreign_org.release();
  e.index.m = e.index.u = 0;
  e.index.table_built = false;
  e.index.direct_built = false;
  e.map_out.n_query = e.map_out.n_matches = e.map_out.n_intervals = e.map_out.n_overlaps = 0;
  e.map_out.first = e.map_out.last = 0;
  e.map_out.has_anchors = false;
  e.polish_last_windows = 0;
  e.polish_last_layers = 0;
  delete e.pile_pool;
  e.pile_pool = nullptr;
}
void engine_release_scratch_if_tight(Engine& e, int stage_kind) {
  e.stage_kind = stage_kind;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
  const bool trace = knob("RVN_DEBUG_MEM") != nullptr;
  if (devpool::active()) {
    // the arena is on (the workload did not fit once): a stage starts from an empty arena when less than a quarter of it
    // is free, or when this kind of stage has run out of memory before with the other phase's scratch alive
    const size_t afree = devpool::free_total(), asize = devpool::size();
    const bool tight = afree * 4 < asize || ((e.oom_mask >> stage_kind) & 1u) || e.opt.release_always != 0;
    if (trace)
      std::fprintf(stderr, "[raven_hip] stage entry (kind %d): arena %.1f GB free of %.1f GB, driver %.1f GB free%s\n", stage_kind,
                   afree / 1e9, asize / 1e9, free_b / 1e9, tight ? " -> scratch released" : "");
    if (tight) engine_release_scratch(e);
    return;
  }
  // (a quarter, not a third: a C4 step settles at ~220 GB of grow-only stage buffers on a 309 GB device — alignment
  // store, window-consensus chunk, sort scratch — and handing them back costs seconds of re-allocation in the next step)
  const bool tight = free_b * 4 < total_b || e.opt.release_always != 0;  // (the latter: tests of this path)
  if (trace)
    std::fprintf(stderr, "[raven_hip] stage entry (kind %d): %.1f GB free of %.1f GB%s\n", stage_kind, free_b / 1e9, total_b / 1e9,
                 tight ? " -> scratch released, arena started" : "");
  if (!tight) return;
  engine_release_scratch(e);
  // From here on the scratch lives in one arena (common.h: devpool): everything that is free now except a margin for
  // the driver's own needs, the buffers that stay outside (reads, pile handles in use) and other users of the device.
  if (e.opt.no_arena) return;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
  size_t margin = std::max<size_t>(12ULL << 30, total_b / 16);
  if (e.opt.arena_margin_mb > 0) margin = static_cast<size_t>(e.opt.arena_margin_mb) << 20;
  if (e.opt.arena_mb > 0) margin = free_b > (static_cast<size_t>(e.opt.arena_mb) << 20) ? free_b - (static_cast<size_t>(e.opt.arena_mb) << 20) : free_b;
  if (free_b > margin + (1ULL << 30)) {
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = devpool::start(free_b - margin);
    if (trace)
      std::fprintf(stderr, "[raven_hip] arena of %.1f GB %s (%.0f ms)\n", (free_b - margin) / 1e9, ok ? "started" : "refused",
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
}
void engine_minimize(Engine& e, const ReadsDev& r, u32 first, u32 last, bool minhash) {
  do_minimize(e, r, first, last, minhash);
}
}  // namespace rvn

struct rvn_pass2 {
  Engine* e = nullptr;
  Pass2State st;
  std::weak_ptr<int> engine_life;
};

extern "C" {

const char* rvn_last_error(void) { return g_err.c_str(); }

int rvn_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rvn_engine_create(rvn_engine** out, uint32_t k, uint32_t w, uint32_t bandwidth, uint32_t chain, uint32_t matches,
                      uint32_t gap, int device) {
  return guarded([&]() -> int {
    if (!out) return fail(RVN_EINVAL, "[raven_hip] rvn_engine_create: out == NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
      return fail(RVN_ENODEVICE, "[raven_hip] no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(RVN_EINVAL, "[raven_hip] invalid device ordinal");
    if (w == 0 || w > static_cast<uint32_t>(kMaxWindow))
      return fail(RVN_EINVAL, "[raven_hip] window length must be in [1, 256]");
    if (chain == 0) return fail(RVN_EINVAL, "[raven_hip] chain must be >= 1");
    RVN_HIP(hipSetDevice(device));
    std::unique_ptr<rvn_engine> h(new rvn_engine());
    Engine& e = h->e;
    e.k = std::min(std::max(k, 1u), 31u);  // as ram's constructor
    e.w = w;
    e.bandwidth = bandwidth;
    e.chain = chain;
    e.matches = matches;
    e.gap = gap;
    e.device = device;
    e.val64 = 2 * e.k >= 32;
    RVN_HIP(hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking));
    RVN_HIP(hipEventCreate(&e.ev0));
    RVN_HIP(hipEventCreate(&e.ev1));
    RVN_HIP(hipHostMalloc(reinterpret_cast<void**>(&e.h_pin), 4096, hipHostMallocDefault));
    *out = h.release();
    return RVN_OK;
  });
}

void rvn_engine_destroy(rvn_engine* h) {
  if (!h) return;
  (void)hipSetDevice(h->e.device);
  if (h->e.stream) (void)rvn_stream_sync(h->e.stream);
  if (h->e.ev0) (void)hipEventDestroy(h->e.ev0);
  if (h->e.ev1) (void)hipEventDestroy(h->e.ev1);
  for (hipEvent_t ev : h->e.nw_ev)
    if (ev) (void)hipEventDestroy(ev);
  for (hipStream_t st2 : h->e.nw_streams)
    if (st2) (void)hipStreamDestroy(st2);
  for (hipEvent_t ev : h->e.nw_side_ev)
    if (ev) (void)hipEventDestroy(ev);
  for (hipStream_t st2 : h->e.nw_side)
    if (st2) (void)hipStreamDestroy(st2);
  if (h->e.stream) (void)hipStreamDestroy(h->e.stream);
  if (h->e.h_pin) (void)hipHostFree(h->e.h_pin);
  delete h->e.pile_pool;
  h->e.pile_pool = nullptr;
  delete h;
  devpool::stop();
}

int rvn_reads_upload(rvn_engine* h, const uint64_t* packed, uint64_t n_words, const uint64_t* word_offsets,
                     const uint32_t* lengths, const uint32_t* ids, uint32_t n, rvn_reads** out) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !out || (n && (!packed || !word_offsets || !lengths)))
      return fail(RVN_EINVAL, "[raven_hip] rvn_reads_upload: NULL argument");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    std::unique_ptr<rvn_reads> rr(new rvn_reads());
    ReadsDev& r = rr->r;
    r.n = n;
    r.h_word_off.assign(word_offsets, word_offsets + n + 1);
    r.h_len.assign(lengths, lengths + n);
    r.h_id.resize(n);
    r.ids_are_indices = true;
    for (u32 i = 0; i < n; ++i) {
      r.h_id[i] = ids ? ids[i] : i;
      if (r.h_id[i] != i || i >= kMaxReadId) r.ids_are_indices = false;
      // (bits 63 / 62 of a minimizer's origin word are the query / query-only flags: 30 bits of read id are left)
      if (r.h_id[i] >= kMaxReadId) return fail(RVN_EINVAL, "[raven_hip] read ids must be below 2^30");
    }
    r.total_bases = 0;
    for (u32 i = 0; i < n; ++i) {
      r.total_bases += lengths[i];
      const u64 need = (static_cast<u64>(lengths[i]) + 31) / 32;
      if (word_offsets[i + 1] < word_offsets[i] || word_offsets[i + 1] - word_offsets[i] < need ||
          word_offsets[i + 1] > n_words)
        return fail(RVN_EINVAL, "[raven_hip] rvn_reads_upload: word_offsets inconsistent with lengths");
    }
    r.n_words = n_words;
    u64* d_packed = r.packed.get<u64>(n_words + 2);
    if (n_words) RVN_HIP(hipMemcpy(d_packed, packed, n_words * 8, hipMemcpyHostToDevice));
    RVN_HIP(hipMemset(d_packed + n_words, 0, 16));  // pad words: kernels may read one word past a read
    u64* d_wo = r.word_off.get<u64>(static_cast<size_t>(n) + 1);
    RVN_HIP(hipMemcpy(d_wo, r.h_word_off.data(), (static_cast<size_t>(n) + 1) * 8, hipMemcpyHostToDevice));
    u32* d_len = r.len.get<u32>(static_cast<size_t>(n) + 1);
    u32* d_id = r.id.get<u32>(static_cast<size_t>(n) + 1);
    if (n) {
      RVN_HIP(hipMemcpy(d_len, r.h_len.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice));
      RVN_HIP(hipMemcpy(d_id, r.h_id.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice));
    }
    reads_build_tiles(e, r);
    *out = rr.release();
    return RVN_OK;
  });
}

void rvn_reads_destroy(rvn_reads* r) { delete r; }

namespace {
// a read set from one-byte codes that are already in HBM (d_codes + boff[i] .. + boff[i + 1]: read i), packed there
int reads_from_device_codes(Engine& e, const u8* d_codes, const std::vector<u64>& boff, const uint32_t* ids, uint32_t n, rvn_reads** out) {
  std::vector<u64> woff(static_cast<size_t>(n) + 1, 0);
  std::vector<u32> lens(n);
  for (u32 i = 0; i < n; ++i) {
    if (boff[i + 1] < boff[i] || boff[i + 1] - boff[i] > 0xFFFFFFFFULL)
      return fail(RVN_EINVAL, "[raven_hip] rvn_reads_upload_codes: bad offsets");
    lens[i] = static_cast<u32>(boff[i + 1] - boff[i]);
    woff[i + 1] = woff[i] + (static_cast<u64>(lens[i]) + 31) / 32;
  }
  const u64 n_words = woff[n], n_codes = n ? boff[n] - boff[0] : 0;
  u64* d_boff = e.tmp_b.get<u64>(static_cast<size_t>(n) + 1);
  u64* d_woff = e.tmp_c.get<u64>(static_cast<size_t>(n) + 1);
  RVN_HIP(hipMemcpy(d_boff, boff.data(), boff.size() * 8, hipMemcpyHostToDevice));
  RVN_HIP(hipMemcpy(d_woff, woff.data(), woff.size() * 8, hipMemcpyHostToDevice));
  std::unique_ptr<rvn_reads> rr(new rvn_reads());
  ReadsDev& r = rr->r;
  u64* d_packed = r.packed.get<u64>(n_words + 2);
  pack_codes_on_device(e, d_codes, d_boff, d_woff, n, n_words, d_packed);
  RVN_HIP(hipMemsetAsync(d_packed + n_words, 0, 16, e.stream));
  RVN_HIP(rvn_stream_sync(e.stream));
  r.n = n;
  r.h_word_off = woff;
  r.h_len = lens;
  r.h_id.resize(n);
  r.ids_are_indices = true;
  r.total_bases = n_codes;
  for (u32 i = 0; i < n; ++i) {
    r.h_id[i] = ids ? ids[i] : i;
    if (r.h_id[i] != i || i >= kMaxReadId) r.ids_are_indices = false;
    if (r.h_id[i] >= kMaxReadId) return fail(RVN_EINVAL, "[raven_hip] read ids must be below 2^30");  // (bit 62 of an origin word is kForeignFlag: ADVICE r05)
  }
  r.n_words = n_words;
  u64* d_wo = r.word_off.get<u64>(static_cast<size_t>(n) + 1);
  RVN_HIP(hipMemcpy(d_wo, r.h_word_off.data(), (static_cast<size_t>(n) + 1) * 8, hipMemcpyHostToDevice));
  u32* d_len = r.len.get<u32>(static_cast<size_t>(n) + 1);
  u32* d_id = r.id.get<u32>(static_cast<size_t>(n) + 1);
  if (n) {
    RVN_HIP(hipMemcpy(d_len, r.h_len.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice));
    RVN_HIP(hipMemcpy(d_id, r.h_id.data(), static_cast<size_t>(n) * 4, hipMemcpyHostToDevice));
  }
  reads_build_tiles(e, r);
  *out = rr.release();
  return RVN_OK;
}
}  // namespace

int rvn_reads_upload_codes(rvn_engine* h, const uint8_t* codes, const uint64_t* offsets, const uint32_t* ids, uint32_t n,
                           rvn_reads** out) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !out || (n && (!codes || !offsets))) return fail(RVN_EINVAL, "[raven_hip] rvn_reads_upload_codes: NULL argument");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    for (u32 i = 0; i < n; ++i)
      if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] > 0xFFFFFFFFULL)
        return fail(RVN_EINVAL, "[raven_hip] rvn_reads_upload_codes: bad offsets");
    const u64 n_codes = n ? offsets[n] - offsets[0] : 0;
    // bases to HBM as bytes, packed there (one thread per word)
    u8* d_codes = e.tmp_a.get<u8>(n_codes + 16);
    if (n_codes) RVN_HIP(hipMemcpy(d_codes, codes + (n ? offsets[0] : 0), n_codes, hipMemcpyHostToDevice));
    std::vector<u64> boff(static_cast<size_t>(n) + 1, 0);
    for (u32 i = 0; i <= n && n; ++i) boff[i] = offsets[i] - offsets[0];
    return reads_from_device_codes(e, d_codes, boff, ids, n, out);
  });
}

// The consensus of the engine's last COMPLETE polishing round as a read set, straight from HBM: what raven::Polish hands the
// next round's racon::Polisher as targets (RavenLib/src/polish.cc:43-74: the polished sequences of round r are the targets of
// round r + 1).  The host has them too (rvn_polish_round returned them); this spares their way back — at C4 100 MB through
// the host's page cache and PCIe per round, ~40 ms with the GPU idle (profiles/r06_gaps.txt).  Bit-identical to
// rvn_reads_upload_codes of the sequences the round returned (tests/test_gpu_polish.py).
int rvn_polish_output_as_reads(rvn_engine* h, rvn_reads** out) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !out) return fail(RVN_EINVAL, "[raven_hip] rvn_polish_output_as_reads: NULL argument");
    Engine& e = h->e;
    if (!e.pl_last_valid) return fail(RVN_EINVAL, "[raven_hip] rvn_polish_output_as_reads: no complete polishing round's consensus is resident");
    RVN_HIP(hipSetDevice(e.device));
    const u32 n = static_cast<u32>(e.pl_last_off.size() - 1);
    return reads_from_device_codes(e, e.pl_final.ptr ? reinterpret_cast<const u8*>(e.pl_final.ptr) : nullptr, e.pl_last_off, nullptr, n, out);
  });
}

int rvn_reads_load(rvn_engine* h, const char* path, rvn_reads** out, rvn_load_stats* stats) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !path || !out) return fail(RVN_EINVAL, "[raven_hip] rvn_reads_load: NULL argument");
    *out = nullptr;
    RVN_HIP(hipSetDevice(h->e.device));
    std::unique_ptr<rvn_reads> rr(new rvn_reads());
    LoadStats st;
    reads_load(h->e, path, rr->r, rr->names, st);
    if (stats) {
      stats->n_sequences = st.n_sequences;
      stats->n_bases = st.n_bases;
      stats->has_quality = st.has_quality;
      stats->parse_s = st.parse_s;
      stats->device_s = st.device_s;
      stats->total_s = st.total_s;
      stats->inflate_threads = st.inflate_threads;
      stats->members = st.members;
      stats->streaming = st.streaming;
      stats->restarted = st.restarted;
    }
    *out = rr.release();
    return RVN_OK;
  });
}

const char* rvn_reads_name(const rvn_reads* r, uint32_t i) {
  return (r && i < r->names.size()) ? r->names[i].c_str() : "";
}

int rvn_reads_info(const rvn_reads* r, uint32_t* n_reads, uint64_t* n_words, uint64_t* n_bases, uint64_t* n_quality_bytes,
                   int* quality_shift) {
  if (!r) return fail(RVN_EINVAL, "[raven_hip] NULL read set");
  if (n_reads) *n_reads = r->r.n;
  if (n_words) *n_words = r->r.n_words;
  if (n_bases) *n_bases = r->r.total_bases;
  if (n_quality_bytes) *n_quality_bytes = r->r.qual_shift >= 0 && !r->r.h_qual_off.empty() ? r->r.h_qual_off.back() : 0;
  if (quality_shift) *quality_shift = r->r.qual_shift;
  return RVN_OK;
}

int rvn_reads_fetch(const rvn_reads* r, uint64_t* packed, uint64_t* word_offsets, uint32_t* lengths, uint8_t* quals,
                    uint64_t* quality_offsets) {
  return guarded([&]() -> int {
    if (!r) return fail(RVN_EINVAL, "[raven_hip] NULL read set");
    const ReadsDev& rd = r->r;
    if (packed && rd.n_words) RVN_HIP(hipMemcpy(packed, rd.packed.ptr, rd.n_words * 8, hipMemcpyDeviceToHost));
    if (word_offsets) std::memcpy(word_offsets, rd.h_word_off.data(), rd.h_word_off.size() * 8);
    if (lengths && rd.n) std::memcpy(lengths, rd.h_len.data(), static_cast<size_t>(rd.n) * 4);
    if (rd.qual_shift >= 0 && !rd.h_qual_off.empty()) {
      if (quals && rd.h_qual_off.back()) RVN_HIP(hipMemcpy(quals, rd.quals.ptr, rd.h_qual_off.back(), hipMemcpyDeviceToHost));
      if (quality_offsets) std::memcpy(quality_offsets, rd.h_qual_off.data(), rd.h_qual_off.size() * 8);
    }
    return RVN_OK;
  });
}

int rvn_reads_attach_quality(rvn_engine* h, rvn_reads* rr, const uint8_t* quals, const uint64_t* offsets, int block_shift) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !rr) return fail(RVN_EINVAL, "[raven_hip] rvn_reads_attach_quality: NULL argument");
    ReadsDev& r = rr->r;
    if (!quals) {  // detach
      r.qual_shift = -1;
      return RVN_OK;
    }
    if (!offsets || (block_shift != 0 && block_shift != 6))
      return fail(RVN_EINVAL, "[raven_hip] rvn_reads_attach_quality: offsets missing or block_shift not 0 / 6");
    for (u32 i = 0; i < r.n; ++i) {
      const u64 need = (static_cast<u64>(r.h_len[i]) + (1u << block_shift) - 1) >> block_shift;
      if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] < need)
        return fail(RVN_EINVAL, "[raven_hip] rvn_reads_attach_quality: offsets inconsistent with the read lengths");
    }
    RVN_HIP(hipSetDevice(h->e.device));
    const u64 total = offsets[r.n];
    u8* dq = r.quals.get<u8>(total + 16);
    u64* dqo = r.qual_off.get<u64>(static_cast<size_t>(r.n) + 1);
    if (total) RVN_HIP(hipMemcpy(dq, quals, total, hipMemcpyHostToDevice));
    RVN_HIP(hipMemcpy(dqo, offsets, (static_cast<size_t>(r.n) + 1) * 8, hipMemcpyHostToDevice));
    r.h_qual_off.assign(offsets, offsets + r.n + 1);
    r.qual_shift = block_shift;
    return RVN_OK;
  });
}

int rvn_engine_minimize(rvn_engine* h, const rvn_reads* r, uint32_t first, uint32_t last, int minhash) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !r || first > last || last > r->r.n) return fail(RVN_EINVAL, "[raven_hip] rvn_engine_minimize: bad range");
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    do_minimize(h->e, r->r, first, last, minhash != 0);
    RVN_HIP(rvn_stream_sync(h->e.stream));
    return RVN_OK;
  });
}

int rvn_engine_filter(rvn_engine* h, double f) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    if (!(0 <= f && f <= 1)) return fail(RVN_EINVAL, "[ram::MinimizerEngine::Filter] error: invalid frequency");
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    index_filter(h->e, f);
    return RVN_OK;
  });
}

uint32_t rvn_engine_occurrence(const rvn_engine* h) { return h ? h->e.index.occurrence : 0; }

int rvn_engine_map_batch(rvn_engine* h, const rvn_reads* r, uint32_t first, uint32_t last, int avoid_equal,
                         int avoid_symmetric, int minhash, int want_filtered, uint64_t* n_overlaps) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !r || first > last || last > r->r.n) return fail(RVN_EINVAL, "[raven_hip] rvn_engine_map_batch: bad range");
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    map_batch(h->e, r->r, first, last, avoid_equal != 0, avoid_symmetric != 0, minhash != 0, want_filtered != 0,
              h->e.map_out);
    h->e.c_intervals += h->e.map_out.n_intervals;
    RVN_HIP(rvn_stream_sync(h->e.stream));
    if (n_overlaps) *n_overlaps = h->e.map_out.n_overlaps;
    return RVN_OK;
  });
}

int rvn_engine_map_fetch(rvn_engine* h, rvn_overlap* overlaps, uint32_t* read_offsets) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    MapOut& m = h->e.map_out;
    RVN_HIP(hipSetDevice(h->e.device));
    if (overlaps && m.n_overlaps)
      RVN_HIP(hipMemcpy(overlaps, m.ovl.ptr, m.n_overlaps * sizeof(Overlap), hipMemcpyDeviceToHost));
    if (read_offsets)
      RVN_HIP(hipMemcpy(read_offsets, m.ovl_read_off.ptr, (static_cast<size_t>(m.last - m.first) + 1) * 4,
                        hipMemcpyDeviceToHost));
    return RVN_OK;
  });
}

int rvn_engine_map_fetch_filtered(rvn_engine* h, uint32_t* positions, uint32_t* read_offsets, uint64_t* total) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    Engine& e = h->e;
    MapOut& m = e.map_out;
    RVN_HIP(hipSetDevice(e.device));
    const u64 nq = m.n_query;
    const u32 nr = m.last - m.first;
    std::vector<u8> flags(nq);
    std::vector<u64> org(nq);
    std::vector<u32> roff(static_cast<size_t>(nr) + 1, 0);
    if (nq) {
      RVN_HIP(hipMemcpy(flags.data(), m.filtered.ptr, nq, hipMemcpyDeviceToHost));
      RVN_HIP(hipMemcpy(org.data(), e.query_sketch.org.ptr, nq * 8, hipMemcpyDeviceToHost));
    }
    RVN_HIP(hipMemcpy(roff.data(), e.query_sketch.read_off.ptr, roff.size() * 4, hipMemcpyDeviceToHost));
    u64 tot = 0;
    for (u32 i = 0; i < nr; ++i) {
      if (read_offsets) read_offsets[i] = static_cast<u32>(tot);
      for (u32 q = roff[i]; q < roff[i + 1]; ++q) {
        if (flags[q]) {
          if (positions) positions[tot] = static_cast<u32>(org[q]) >> 1;
          ++tot;
        }
      }
    }
    if (read_offsets) read_offsets[nr] = static_cast<u32>(tot);
    if (total) *total = tot;
    return RVN_OK;
  });
}

void rvn_free(void* p) { std::free(p); }

int rvn_engine_release_scratch(rvn_engine* h) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    RVN_HIP(hipSetDevice(h->e.device));
    engine_release_scratch(h->e);
    devpool::stop();  // the caller asked for the memory itself (the arena goes if nothing of it is in use)
    return RVN_OK;
  });
}

int rvn_engine_map_collect(rvn_engine* h, const rvn_reads* r, uint32_t first, uint32_t last, int avoid_equal,
                           int avoid_symmetric, int minhash, int want_filtered, rvn_overlap** overlaps,
                           uint32_t** read_offsets, uint32_t** filtered, uint32_t** filtered_offsets) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !r || !overlaps || !read_offsets || first > last || last > r->r.n)
      return fail(RVN_EINVAL, "[raven_hip] rvn_engine_map_collect: bad argument");
    if (want_filtered && (!filtered || !filtered_offsets))
      return fail(RVN_EINVAL, "[raven_hip] rvn_engine_map_collect: filtered requested without output pointers");
    *overlaps = nullptr;
    *read_offsets = nullptr;
    if (filtered) *filtered = nullptr;
    if (filtered_offsets) *filtered_offsets = nullptr;
    uint64_t n = 0;
    int rc = rvn_engine_map_batch(h, r, first, last, avoid_equal, avoid_symmetric, minhash, want_filtered, &n);
    if (rc != RVN_OK) return rc;
    const size_t nr = last - first;
    auto* ov = static_cast<rvn_overlap*>(std::malloc((n + 1) * sizeof(rvn_overlap)));
    auto* off = static_cast<uint32_t*>(std::malloc((nr + 1) * 4));
    uint32_t *fp = nullptr, *fo = nullptr;
    auto drop = [&]() {
      std::free(ov);
      std::free(off);
      std::free(fp);
      std::free(fo);
    };
    if (!ov || !off) {
      drop();
      return fail(RVN_ENOMEM, "[raven_hip] out of host memory");
    }
    rc = rvn_engine_map_fetch(h, ov, off);
    if (rc == RVN_OK && want_filtered) {
      uint64_t total = 0;
      rc = rvn_engine_map_fetch_filtered(h, nullptr, nullptr, &total);
      if (rc == RVN_OK) {
        fp = static_cast<uint32_t*>(std::malloc((total + 1) * 4));
        fo = static_cast<uint32_t*>(std::malloc((nr + 1) * 4));
        if (!fp || !fo) {
          drop();
          return fail(RVN_ENOMEM, "[raven_hip] out of host memory");
        }
        rc = rvn_engine_map_fetch_filtered(h, fp, fo, &total);
      }
    }
    if (rc != RVN_OK) {
      drop();
      return rc;
    }
    *overlaps = ov;
    *read_offsets = off;
    if (want_filtered) {
      *filtered = fp;
      *filtered_offsets = fo;
    }
    return RVN_OK;
  });
}

int rvn_find_overlaps_and_create_piles(rvn_engine* h, const rvn_reads* rr, double freq, uint32_t kmax,
                                       int use_minhash, uint64_t index_batch_bases, uint64_t flush_bases,
                                       rvn_pass1** out) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !rr || !out) return fail(RVN_EINVAL, "[raven_hip] NULL argument");
    if (!(0 <= freq && freq <= 1)) return fail(RVN_EINVAL, "[ram::MinimizerEngine::Filter] error: invalid frequency");
    Engine& e = h->e;
    const ReadsDev& r = rr->r;
    const bool dbg = knob("RVN_DEBUG_PASS1") != nullptr;  // host wall time per stage (synchronising)
    auto t_last = std::chrono::steady_clock::now();
    for (u32 i = 0; i < r.n; ++i)
      if (r.h_id[i] != i) return fail(RVN_EINVAL, "[raven_hip] FindOverlapsAndCreatePiles requires ids[i] == i");
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    if (dbg) std::fprintf(stderr, "[raven_hip] pass1: %-12s %8.1f ms\n", "timers",
                          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_last).count());
    engine_release_scratch_if_tight(e, 0);
    auto lap = [&](const char* what) {
      if (!dbg) return;
      (void)rvn_stream_sync(e.stream);
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[raven_hip] pass1: %-12s %8.1f ms\n", what,
                   std::chrono::duration<double, std::milli>(now - t_last).count());
      t_last = now;
    };
    std::unique_ptr<rvn_pass1> p(new rvn_pass1(e));
    lap("handle");
    piles_init(e, r, p->ps);
    lap("piles_init");
    const u32 n = r.n;
    // construct.cc:32-120
    u64 bytes = 0;
    for (u32 i = 0, j = 0; i < n; ++i) {
      bytes += r.h_len[i];
      if (i != n - 1 && bytes < index_batch_bases) continue;
      bytes = 0;
      // does the first query flush cover exactly the index range [j, i+1)?  (always true for one batch)
      bool prefetch = false;
      if (j == 0) {
        u64 fb = 0;
        u32 kk = 0;
        for (; kk < i + 1; ++kk) {
          fb += r.h_len[kk];
          if (kk != i && fb < flush_bases) continue;
          break;
        }
        prefetch = (kk == i);
      }
      do_minimize(e, r, j, i + 1, use_minhash != 0, prefetch);
      lap("minimize");
      index_filter(e, freq);
      lap("filter");
      u32 flush_first = 0;
      for (u32 k = 0; k < i + 1; ++k) {
        bytes += r.h_len[k];
        if (k != i && bytes < flush_bases) continue;
        bytes = 0;
        map_batch(e, r, flush_first, k + 1, true, true, true, false, e.map_out);
        lap("map_batch");
        e.c_intervals += e.map_out.n_intervals;
        piles_merge(e, r, e.map_out, kmax, p->ps);
        lap("piles_merge");
        flush_first = k + 1;
      }
      j = i + 1;
    }
    RVN_HIP(rvn_stream_sync(e.stream));
    *out = p.release();
    return RVN_OK;
  });
}

uint64_t rvn_pass1_pile_words(const rvn_pass1* p) { return p ? p->ps.pile_words : 0; }
uint64_t rvn_pass1_num_overlaps(const rvn_pass1* p) { return p ? p->ps.kept_total : 0; }

int rvn_pass1_fetch_piles(const rvn_pass1* p, uint16_t* data, uint64_t* offsets) {
  return guarded(p ? p->e : nullptr, [&]() -> int {
    if (!p) return fail(RVN_EINVAL, "[raven_hip] NULL pass1");
    RVN_HIP(hipSetDevice(p->e->device));
    if (data && p->ps.pile_words)
      RVN_HIP(hipMemcpy(data, p->ps.pile_data.ptr, p->ps.pile_words * 2, hipMemcpyDeviceToHost));
    if (offsets)
      RVN_HIP(hipMemcpy(offsets, p->ps.pile_off.ptr, (static_cast<size_t>(p->ps.n) + 1) * 8, hipMemcpyDeviceToHost));
    return RVN_OK;
  });
}

int rvn_pass1_trim_and_annotate(rvn_pass1* p, uint32_t coverage, uint32_t* begin, uint32_t* end, uint16_t* median,
                                uint8_t* invalid) {
  return guarded(p ? p->e : nullptr, [&]() -> int {
    if (!p) return fail(RVN_EINVAL, "[raven_hip] NULL pass1");
    if (coverage > 65535) return fail(RVN_EINVAL, "[raven_hip] coverage threshold above 65535");
    RVN_HIP(hipSetDevice(p->e->device));
    UseTimers ut(*p->e);
    piles_trim_and_median(*p->e, p->ps, coverage, begin, end, median, invalid);
    return RVN_OK;
  });
}

int rvn_pass1_find_chimeric_regions(rvn_pass1* p, const uint8_t* invalid, uint32_t* region_offsets, uint32_t** regions) {
  return guarded(p ? p->e : nullptr, [&]() -> int {
    if (!p || !invalid || !region_offsets || !regions) return fail(RVN_EINVAL, "[raven_hip] NULL argument");
    *regions = nullptr;
    RVN_HIP(hipSetDevice(p->e->device));
    UseTimers ut(*p->e);
    std::vector<u32> off, reg;
    piles_find_chimeric_regions(*p->e, p->ps, invalid, off, reg);
    std::memcpy(region_offsets, off.data(), off.size() * 4);
    auto* out = static_cast<uint32_t*>(std::malloc((reg.size() + 1) * 4));
    if (!out) return fail(RVN_ENOMEM, "[raven_hip] out of host memory");
    if (!reg.empty()) std::memcpy(out, reg.data(), reg.size() * 4);
    *regions = out;
    return RVN_OK;
  });
}

int rvn_pass1_fetch_overlaps(const rvn_pass1* p, rvn_overlap* overlaps, uint32_t* offsets) {
  return guarded(p ? p->e : nullptr, [&]() -> int {
    if (!p) return fail(RVN_EINVAL, "[raven_hip] NULL pass1");
    RVN_HIP(hipSetDevice(p->e->device));
    if (overlaps && p->ps.kept_total)
      RVN_HIP(hipMemcpy(overlaps, p->ps.kept.ptr, p->ps.kept_total * sizeof(Overlap), hipMemcpyDeviceToHost));
    if (offsets)
      RVN_HIP(hipMemcpy(offsets, p->ps.kept_off.ptr, (static_cast<size_t>(p->ps.n) + 1) * 4, hipMemcpyDeviceToHost));
    return RVN_OK;
  });
}

void rvn_pass1_destroy(rvn_pass1* p) { delete p; }

int rvn_find_overlaps_and_repetitive_regions(rvn_engine* h, const rvn_reads* rr, const uint32_t* pile_begin,
                                             const uint32_t* pile_end, const uint8_t* pile_invalid, double freq,
                                             uint32_t kmer_len, double identity, uint64_t batch_bases, rvn_pass2** out) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !rr || !out || (rr->r.n && (!pile_begin || !pile_end || !pile_invalid)))
      return fail(RVN_EINVAL, "[raven_hip] rvn_find_overlaps_and_repetitive_regions: NULL argument");
    if (!(0 <= freq && freq <= 1)) return fail(RVN_EINVAL, "[ram::MinimizerEngine::Filter] error: invalid frequency");
    if (kmer_len == 0 || kmer_len > 32) return fail(RVN_EINVAL, "[raven_hip] kmer_len must be in [1, 32]");
    if (batch_bases == 0) return fail(RVN_EINVAL, "[raven_hip] batch_bases must be positive");
    Engine& e = h->e;
    const ReadsDev& r = rr->r;
    for (u32 i = 0; i < r.n; ++i)
      if (r.h_id[i] != i) return fail(RVN_EINVAL, "[raven_hip] FindOverlapsAndRepetetiveRegions requires ids[i] == i");
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    engine_release_scratch_if_tight(e, 1);
    std::unique_ptr<rvn_pass2> p(new rvn_pass2());
    p->e = &e;
    p->engine_life = e.life;
    second_pass(e, r, pile_begin, pile_end, pile_invalid, freq, kmer_len, identity, batch_bases, p->st);
    *out = p.release();
    return RVN_OK;
  });
}

uint64_t rvn_pass2_num_overlaps(const rvn_pass2* p) { return p ? p->st.n_overlaps : 0; }
uint64_t rvn_pass2_kmer_cells(const rvn_pass2* p) { return p ? p->st.kmers_total : 0; }

int rvn_pass2_fetch(const rvn_pass2* p, rvn_overlap* overlaps, uint8_t* contained, uint8_t* kmers, uint64_t* kmers_offsets) {
  if (!p) return fail(RVN_EINVAL, "[raven_hip] NULL pass2");
  if (p->engine_life.expired()) return fail(RVN_EINVAL, "[raven_hip] the engine of this result is gone");
  return guarded(p->e, [&]() -> int {
    RVN_HIP(hipSetDevice(p->e->device));
    const Pass2State& st = p->st;
    if (overlaps && st.n_overlaps)
      RVN_HIP(hipMemcpy(overlaps, st.ovl.ptr, st.n_overlaps * sizeof(Overlap), hipMemcpyDeviceToHost));
    if (contained && st.n) RVN_HIP(hipMemcpy(contained, st.contained.ptr, st.n, hipMemcpyDeviceToHost));
    if (kmers && st.kmers_total) RVN_HIP(hipMemcpy(kmers, st.kmers.ptr, st.kmers_total, hipMemcpyDeviceToHost));
    if (kmers_offsets) std::memcpy(kmers_offsets, st.h_kmers_off.data(), st.h_kmers_off.size() * 8);
    return RVN_OK;
  });
}

void rvn_pass2_destroy(rvn_pass2* p) { delete p; }

int rvn_filter_overlaps_by_identity(rvn_engine* h, const rvn_reads* rr, rvn_overlap* overlaps, uint32_t* offsets,
                                    const uint32_t* pile_begin, const uint32_t* pile_end, const uint8_t* pile_invalid,
                                    double identity) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !rr || !offsets || !pile_begin || !pile_end || !pile_invalid || (offsets[rr->r.n] && !overlaps))
      return fail(RVN_EINVAL, "[raven_hip] rvn_filter_overlaps_by_identity: NULL argument");
    const ReadsDev& r = rr->r;
    for (u32 i = 0; i < r.n; ++i)
      if (r.h_id[i] != i) return fail(RVN_EINVAL, "[raven_hip] the identity filter requires ids[i] == i");
    for (u64 x = 0; x < offsets[r.n]; ++x)
      if (overlaps[x].lhs_id >= r.n || overlaps[x].rhs_id >= r.n)
        return fail(RVN_EINVAL, "[raven_hip] rvn_filter_overlaps_by_identity: overlap of an unknown read");
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    identity_filter_lists(h->e, r, reinterpret_cast<Overlap*>(overlaps), offsets, pile_begin, pile_end, pile_invalid, identity);
    return RVN_OK;
  });
}

#ifdef RVN_TEST_HOOKS
int64_t rvn_test_find_chimeric_regions(const uint16_t* data, uint32_t size, uint32_t* out, uint64_t cap_pairs) {
  if (!data || !out || size == 0) return RVN_EINVAL;
  std::vector<SlopeRegion> slopes(2 * static_cast<size_t>(size) + 2);  // same bounds as the device path (pile.hip)
  std::vector<u16> tmp(size + 1);
  bool overflow = false;
  const u32 n = find_chimeric_regions(data, static_cast<int>(size), slopes.data(), 2 * size, tmp.data(), out,
                                      static_cast<u32>(std::min<uint64_t>(cap_pairs, size)), &overflow);
  return overflow ? -5 : static_cast<int64_t>(n);
}

int rvn_test_overlap_update_and_type(rvn_overlap* overlaps, uint64_t n, const uint32_t* pile_begin, const uint32_t* pile_end,
                                     const uint8_t* pile_invalid, uint32_t n_piles, uint8_t* ok, uint32_t* type) {
  return rvn_overlap_update_and_type(overlaps, n, pile_begin, pile_end, pile_invalid, n_piles, ok, type);
}
#endif  // RVN_TEST_HOOKS

int rvn_overlap_update_and_type(rvn_overlap* overlaps, uint64_t n, const uint32_t* pile_begin, const uint32_t* pile_end,
                                const uint8_t* pile_invalid, uint32_t n_piles, uint8_t* ok, uint32_t* type) {
  if (!overlaps || !pile_begin || !pile_end || !pile_invalid || !ok || !type) return fail(RVN_EINVAL, "[raven_hip] NULL argument");
  for (uint64_t i = 0; i < n; ++i) {
    Overlap& o = reinterpret_cast<Overlap*>(overlaps)[i];
    if (o.lhs_id >= n_piles || o.rhs_id >= n_piles) return fail(RVN_EINVAL, "[raven_hip] overlap names a pile beyond n_piles");
    const PileRegion L{pile_begin[o.lhs_id], pile_end[o.lhs_id], pile_invalid[o.lhs_id] ? 1u : 0u};
    const PileRegion R{pile_begin[o.rhs_id], pile_end[o.rhs_id], pile_invalid[o.rhs_id] ? 1u : 0u};
    ok[i] = overlap_update(o, L, R) ? 1 : 0;
    type[i] = ok[i] ? overlap_type(o, L, R) : 0xFFFFFFFFu;
  }
  return RVN_OK;
}

int rvn_pile_add_layers(rvn_engine* h, uint16_t* data, uint32_t cells, uint32_t id, const rvn_overlap* overlaps,
                        uint64_t n) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || (cells && !data) || (n && !overlaps)) return fail(RVN_EINVAL, "[raven_hip] NULL argument");
    if (n == 0 || cells == 0) return RVN_OK;
    if (n >= (1ULL << 31)) return fail(RVN_EINVAL, "[raven_hip] too many overlaps");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    // a one-pile PileState whose "new list" is the caller's overlaps
    ReadsDev r;
    r.n = 1;
    r.h_len = {cells << 4};
    r.h_id = {id};
    u32* d_id = r.id.get<u32>(1);
    RVN_HIP(hipMemcpy(d_id, &id, 4, hipMemcpyHostToDevice));
    PileState ps;
    piles_init(e, r, ps);
    RVN_HIP(hipMemcpy(ps.pile_data.ptr, data, static_cast<size_t>(cells) * 2, hipMemcpyHostToDevice));
    pile_add_layers_single(e, ps, d_id, reinterpret_cast<const Overlap*>(overlaps), static_cast<u32>(n));
    RVN_HIP(hipMemcpy(data, ps.pile_data.ptr, static_cast<size_t>(cells) * 2, hipMemcpyDeviceToHost));
    return RVN_OK;
  });
}

int rvn_pile_add_kmers_batch(rvn_engine* h, const rvn_reads* r, uint32_t first_read, uint32_t n_reads,
                             const uint32_t* positions, const uint64_t* position_offsets, uint8_t* kmers,
                             const uint64_t* kmers_offsets) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !r || (n_reads && (!position_offsets || !kmers || !kmers_offsets)))
      return fail(RVN_EINVAL, "[raven_hip] NULL argument");
    const ReadsDev& rd = r->r;
    if (static_cast<u64>(first_read) + n_reads > rd.n) return fail(RVN_EINVAL, "[raven_hip] bad read range");
    for (uint32_t i = 0; i < n_reads; ++i) {
      const u32 len = rd.h_len[first_read + i];
      if (kmers_offsets[i + 1] - kmers_offsets[i] < (static_cast<u64>(len) >> 4) + 1)
        return fail(RVN_EINVAL, "[raven_hip] rvn_pile_add_kmers_batch: kmers buffer smaller than (len >> 4) + 1");
      for (uint64_t q = position_offsets[i]; q < position_offsets[i + 1]; ++q)
        if (static_cast<u64>(positions[q]) + h->e.k > len)
          return fail(RVN_EINVAL, "[raven_hip] rvn_pile_add_kmers_batch: k-mer position outside its read");
    }
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    pile_add_kmers_batch(h->e, rd, positions, position_offsets, n_reads, first_read, kmers, kmers_offsets);
    return RVN_OK;
  });
}

int rvn_polish_round_range(rvn_engine* h, rvn_reads* targets, rvn_reads* reads, const uint8_t* read_quals,
                           const uint64_t* qual_offsets, double q, double err, uint32_t w, int trim, int match,
                           int mismatch, int gap, uint64_t window_first, uint64_t window_last, uint8_t* out_codes,
                           const uint64_t* out_offsets, uint32_t* out_len, double* ratio, uint32_t* n_windows,
                           uint32_t* n_polished, rvn_polish_stats* stats) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !targets || !reads || !out_codes || !out_offsets || !out_len)
      return fail(RVN_EINVAL, "[raven_hip] NULL argument");
    if (w == 0) return fail(RVN_EINVAL, "[racon::Polisher::Create] error: invalid window length!");
    if (read_quals && !qual_offsets) return fail(RVN_EINVAL, "[raven_hip] qualities without offsets");
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    engine_release_scratch_if_tight(h->e, 2);
    std::vector<std::vector<u8>> polished;
    std::vector<double> rt;
    PolishStats st;
    std::vector<u32> wc, wp;
    // (the consensus goes from the page-locked read-back buffer straight into the caller's — usually never touched — pages, on a
    // few threads: 100 MB at C4; a buffer too small for a target fails the call with RVN_EINVAL)
    std::vector<u64> lens(targets->r.n, 0);
    const PolishDirectOut direct{out_codes, out_offsets, lens.data()};
    polish_round(h->e, targets->r, reads->r, read_quals, qual_offsets, q, err, w, trim != 0, match, mismatch, gap,
                 polished, rt, st, window_first, window_last, &wc, &wp, &direct);
    for (u32 t = 0; t < targets->r.n; ++t) {
      out_len[t] = static_cast<uint32_t>(lens[t]);
      if (ratio) ratio[t] = rt[t];
      if (n_windows) n_windows[t] = wc[t];
      if (n_polished) n_polished[t] = wp[t];
    }
    if (stats) {
      stats->n_overlaps = st.n_overlaps;
      stats->n_reads_used = st.n_reads_used;
      stats->n_layers = st.n_layers;
      stats->n_windows = st.n_windows;
      stats->n_polished_windows = st.n_polished_windows;
      stats->n_failed_windows = st.n_failed_windows;
      stats->poa_ms = st.poa_ms;
      stats->map_ms = st.map_ms;
      stats->host_ms = st.host_ms;
      stats->total_ms = st.total_ms;
      stats->n_dropped_layers = st.n_dropped_layers;
      stats->align_ms = st.align_ms;
      stats->n_aligned = st.n_aligned;
      stats->n_align_retries = st.n_align_retries;
      stats->align_band_cells = st.align_band_cells;
      stats->align_store_bytes = st.align_store_bytes;
    }
    return RVN_OK;
  });
}

int rvn_edit_distance_batch(rvn_engine* h, const rvn_reads* r, const rvn_ed_pair* pairs, uint32_t n_pairs,
                            uint32_t* distances, double* device_ms, uint64_t* cells) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !r || (n_pairs && (!pairs || !distances))) return fail(RVN_EINVAL, "[raven_hip] NULL argument");
    const ReadsDev& rd = r->r;
    for (uint32_t i = 0; i < n_pairs; ++i) {
      const rvn_ed_pair& p = pairs[i];
      if (p.lhs_read >= rd.n || p.rhs_read >= rd.n ||
          static_cast<u64>(p.lhs_begin) + p.lhs_len > rd.h_len[p.lhs_read] ||
          static_cast<u64>(p.rhs_begin) + p.rhs_len > rd.h_len[p.rhs_read])
        return fail(RVN_EINVAL, "[raven_hip] rvn_edit_distance_batch: span outside its read");
    }
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    static_assert(sizeof(rvn_ed_pair) == 32, "pair layout");
    edit_distance_batch(h->e, rd, reinterpret_cast<const u32*>(pairs), n_pairs, distances, device_ms, cells);
    return RVN_OK;
  });
}

int rvn_poa_consensus_batch(rvn_engine* h, const uint8_t* codes, const uint8_t* quals, const uint64_t* layer_offsets,
                            const uint32_t* begins, const uint32_t* ends, const uint32_t* has_qual,
                            const uint32_t* window_offsets, uint32_t n_windows, int match, int mismatch, int gap,
                            int trim, uint8_t* consensus, const uint64_t* consensus_offsets, uint32_t* consensus_len,
                            uint32_t* status, double* device_ms) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || (n_windows && (!codes || !layer_offsets || !begins || !ends || !window_offsets || !consensus ||
                             !consensus_offsets || !consensus_len || !status)))
      return fail(RVN_EINVAL, "[raven_hip] NULL argument");
    for (uint32_t w = 0; w < n_windows; ++w) {
      const uint32_t f = window_offsets[w], l = window_offsets[w + 1];
      if (l <= f) return fail(RVN_EINVAL, "[raven_hip] rvn_poa_consensus_batch: window without a backbone");
      const uint64_t blen = layer_offsets[f + 1] - layer_offsets[f];
      for (uint32_t i = f + 1; i < l; ++i)  // racon Window::AddLayer checks
        if (layer_offsets[i + 1] > layer_offsets[i] && (begins[i] >= ends[i] || ends[i] >= blen))
          return fail(RVN_EINVAL, "[racon::Window::AddLayer] error: layer begin and end positions are invalid!");
    }
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    poa_consensus_batch(h->e, codes, quals, layer_offsets, begins, ends, has_qual, window_offsets, n_windows, match,
                        mismatch, gap, trim, consensus, consensus_offsets, consensus_len, status, device_ms);
    return RVN_OK;
  });
}

#ifdef RVN_TEST_HOOKS
int rvn_poa_banded_emulate(const uint8_t* codes, const uint8_t* quals, const uint64_t* layer_offsets, const uint32_t* begins,
                           const uint32_t* ends, const uint32_t* has_qual, const uint32_t* window_offsets,
                           uint32_t n_windows, int match, int mismatch, int gap, int trim, uint8_t* consensus,
                           const uint64_t* consensus_offsets, uint32_t* consensus_len, uint32_t* status, int variant) {
  return guarded([&]() -> int {
    if (n_windows && (!codes || !layer_offsets || !begins || !ends || !window_offsets || !consensus || !consensus_offsets ||
                      !consensus_len || !status))
      return fail(RVN_EINVAL, "[raven_hip] NULL argument");
    for (uint32_t w = 0; w < n_windows; ++w)
      if (window_offsets[w + 1] <= window_offsets[w])
        return fail(RVN_EINVAL, "[raven_hip] rvn_poa_banded_emulate: window without a backbone");
    poa_banded_emulate(codes, quals, layer_offsets, begins, ends, has_qual, window_offsets, n_windows, match, mismatch, gap,
                       trim, consensus, consensus_offsets, consensus_len, status, variant);
    return RVN_OK;
  });
}
#endif  // RVN_TEST_HOOKS

int rvn_polish_map_best(rvn_engine* h, rvn_reads* targets, rvn_reads* reads, uint32_t read_first, uint32_t read_last,
                        double err, rvn_overlap* best, uint32_t* best_target, uint64_t* n_overlaps) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !targets || !reads || !best || !best_target) return fail(RVN_EINVAL, "[raven_hip] rvn_polish_map_best: NULL argument");
    if (read_first > read_last || read_last > reads->r.n) return fail(RVN_EINVAL, "[raven_hip] rvn_polish_map_best: bad read range");
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    engine_release_scratch_if_tight(h->e, 3);
    std::vector<Overlap> b;
    std::vector<u32> bt;
    u64 n = 0;
    polish_map_best(h->e, targets->r, reads->r, read_first, read_last, err, b, bt, &n);
    if (!b.empty()) std::memcpy(best, b.data(), b.size() * sizeof(Overlap));
    if (!bt.empty()) std::memcpy(best_target, bt.data(), bt.size() * 4);
    if (n_overlaps) *n_overlaps = n;
    return RVN_OK;
  });
}

int rvn_polish_set_best(rvn_engine* h, const rvn_overlap* best, const uint32_t* best_target, uint32_t n_reads) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || (n_reads && (!best || !best_target))) return fail(RVN_EINVAL, "[raven_hip] rvn_polish_set_best: NULL argument");
    Engine& e = h->e;
    e.polish_given_best.resize(n_reads);
    e.polish_given_best_t.assign(best_target, best_target + n_reads);
    if (n_reads) std::memcpy(e.polish_given_best.data(), best, static_cast<size_t>(n_reads) * sizeof(Overlap));
    e.polish_given_valid = true;
    return RVN_OK;
  });
}

int rvn_polish_round(rvn_engine* h, rvn_reads* targets, rvn_reads* reads, const uint8_t* read_quals,
                     const uint64_t* qual_offsets, double q, double err, uint32_t w, int trim, int match, int mismatch,
                     int gap, uint8_t* out_codes, const uint64_t* out_offsets, uint32_t* out_len, double* ratio,
                     rvn_polish_stats* stats) {
  return rvn_polish_round_range(h, targets, reads, read_quals, qual_offsets, q, err, w, trim, match, mismatch, gap, 0,
                                ~0ULL, out_codes, out_offsets, out_len, ratio, nullptr, nullptr, stats);
}

// ---- stage-level entry points of the sharded single-genome pass (host side raven_amd/sharded.py) ----
int rvn_shard_sketch(rvn_engine* h, const rvn_reads* rr, int index_minhash, uint64_t* count) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !rr || !count) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_sketch: NULL argument");
    Engine& e = h->e;
    const ReadsDev& r = rr->r;
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    StageTimer t(e, StageTimes::kSketch);
    e.query_ready = false;
    sketch_raw(e, r, 0, r.n, e.raw_sketch);
    if (index_minhash) {
      sketch_minhash(e, r, e.raw_sketch, e.index_sketch);
      e.shard_sketch_minhash = true;
      *count = e.index_sketch.count;
    } else {
      e.join_query_count = sketch_flag_queries(e, r, e.raw_sketch);  // minhash-selected entries get kQueryFlag
      e.shard_sketch_minhash = false;
      *count = e.raw_sketch.count;
    }
    for (u32 i = 0; i < r.n; ++i) e.c_index_bases += r.h_len[i];
    t.stop();
    RVN_HIP(rvn_stream_sync(e.stream));
    return RVN_OK;
  });
}

namespace {
__global__ void or_flags_kernel(u64* __restrict__ org, u64 n, u64 flags) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) org[i] |= flags;
}
}  // namespace

int rvn_shard_sketch_range(rvn_engine* h, const rvn_reads* rr, uint32_t first, uint32_t last, int index_minhash, int foreign,
                           uint64_t* count) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !rr || !count) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_sketch_range: NULL argument");
    Engine& e = h->e;
    const ReadsDev& r = rr->r;
    if (first > last || last > r.n) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_sketch_range: range beyond the read set");
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    StageTimer t(e, StageTimes::kSketch);
    e.query_ready = false;
    *count = 0;
    e.shard_sketch_minhash = (index_minhash != 0) || (foreign != 0);
    Sketch& res = e.shard_sketch_minhash ? e.index_sketch : e.raw_sketch;
    res.count = 0;
    if (first == last) {
      t.stop();
      return RVN_OK;
    }
    if (foreign) {
      // reads of an EARLIER index batch: only what Map() would look up for them — their minhash-selected minimizers — as
      // query-only entries.  The prefix [0, f_hi) of a late batch has no bound of its own (>= 12.9 Gbases at w = 5
      // are >= 2^32 raw minimizers, and the sketch's offsets are 32-bit), so it is sketched in pieces of <= kForeignPieceBases
      // and the minhash-selected entries (at most len / k per read) are appended: what comes out is bounded by its own
      // count only, which the index build checks against 2^32 (index.hip).
      u64 piece_bases = kForeignPieceBases;
#if defined(RVN_DEBUG_KNOBS)
      if (const char* pb = knob("RVN_FOREIGN_PIECE_BASES")) piece_bases = std::strtoull(pb, nullptr, 10);  // (tests: several pieces on a small set)
#endif
      std::vector<u32> cuts{first};
      u64 acc = 0, bound = 0;
      for (u32 i = first; i < last; ++i) {
        if (acc && acc + r.h_len[i] > piece_bases) {
          cuts.push_back(i);
          acc = 0;
        }
        acc += r.h_len[i];
        bound += r.h_len[i] / static_cast<u32>(e.k) + 1;
      }
      cuts.push_back(last);
      if (cuts.size() == 2) {
        sketch_raw(e, r, first, last, e.raw_sketch);
        sketch_minhash(e, r, e.raw_sketch, e.index_sketch);
      } else {
        const size_t vb = e.val64 ? 8 : 4;
        unsigned char* av = e.foreign_val.get<unsigned char>((bound + 1) * vb);
        u64* ao = e.foreign_org.get<u64>(bound + 1);
        u64 n_acc = 0;
        for (size_t c = 0; c + 1 < cuts.size(); ++c) {
          sketch_raw(e, r, cuts[c], cuts[c + 1], e.raw_sketch);
          sketch_minhash(e, r, e.raw_sketch, e.index_sketch);
          const u64 n = e.index_sketch.count;
          if (n_acc + n > bound) throw HipError("[raven_hip] rvn_shard_sketch_range: more selected minimizers than len / k per read");
          if (n) {
            RVN_HIP(hipMemcpyAsync(av + n_acc * vb, e.index_sketch.val.ptr, n * vb, hipMemcpyDeviceToDevice, e.stream));
            RVN_HIP(hipMemcpyAsync(ao + n_acc, e.index_sketch.org.ptr, n * 8, hipMemcpyDeviceToDevice, e.stream));
          }
          n_acc += n;
        }
        RVN_HIP(rvn_stream_sync(e.stream));
        std::swap(e.index_sketch.val.ptr, e.foreign_val.ptr);
        std::swap(e.index_sketch.val.cap, e.foreign_val.cap);
        std::swap(e.index_sketch.org.ptr, e.foreign_org.ptr);
        std::swap(e.index_sketch.org.cap, e.foreign_org.cap);
        e.index_sketch.first = first;
        e.index_sketch.last = last;
        e.index_sketch.count = n_acc;  // (read_off of the pieces is not kept: nothing downstream of a query-only sketch reads it)
      }
      const u64 n = e.index_sketch.count;
      if (n) {
        or_flags_kernel<<<static_cast<u32>((n + 255) / 256), 256, 0, e.stream>>>(e.index_sketch.org.as<u64>(), n, kQueryFlag | kForeignFlag);
        RVN_LAUNCH_CHECK();
      }
    } else if (index_minhash) {
      sketch_raw(e, r, first, last, e.raw_sketch);
      sketch_minhash(e, r, e.raw_sketch, e.index_sketch);
    } else {
      sketch_raw(e, r, first, last, e.raw_sketch);
      e.join_query_count = sketch_flag_queries(e, r, e.raw_sketch);  // minhash-selected entries get kQueryFlag
    }
    *count = res.count;
    if (!foreign)
      for (u32 i = first; i < last; ++i) e.c_index_bases += r.h_len[i];
    t.stop();
    RVN_HIP(rvn_stream_sync(e.stream));
    return RVN_OK;
  });
}

int rvn_shard_sketch_fetch(rvn_engine* h, uint64_t* values, uint64_t* origins) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    Engine& e = h->e;
    Sketch& s = e.shard_sketch_minhash ? e.index_sketch : e.raw_sketch;
    RVN_HIP(hipSetDevice(e.device));
    fetch_values(e, s.val, s.count, values);
    if (origins && s.count) RVN_HIP(hipMemcpy(origins, s.org.ptr, s.count * 8, hipMemcpyDeviceToHost));
    return RVN_OK;
  });
}

int rvn_shard_index_build(rvn_engine* h, const uint64_t* values, const uint64_t* origins, uint64_t n, int all_query) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || (n && (!values || !origins))) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_index_build: NULL argument");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    Sketch& sk = e.index_sketch;
    sk.first = 0;
    sk.last = 0;
    sk.count = n;
    u64 flagged = 0;
    if (e.val64) {
      u64* dv = sk.val.get<u64>(n + 1);
      if (n) RVN_HIP(hipMemcpy(dv, values, n * 8, hipMemcpyHostToDevice));
    } else {
      std::vector<u32> tmp(n);
      for (u64 i = 0; i < n; ++i) tmp[i] = static_cast<u32>(values[i]);
      u32* dv = sk.val.get<u32>(n + 1);
      if (n) RVN_HIP(hipMemcpy(dv, tmp.data(), n * 4, hipMemcpyHostToDevice));
    }
    u64* dorg = sk.org.get<u64>(n + 1);
    if (n) RVN_HIP(hipMemcpy(dorg, origins, n * 8, hipMemcpyHostToDevice));
    for (u64 i = 0; i < n; ++i) flagged += (origins[i] & kQueryFlag) ? 1 : 0;
    e.c_index_min += n;
    index_build(e, sk, false);
    e.c_index_keys += e.index.u;
    e.index.has_query_flags = !all_query;
    e.index.all_query = all_query != 0;
    e.join_query_count = all_query ? n : flagged;
    RVN_HIP(rvn_stream_sync(e.stream));
    return RVN_OK;
  });
}

int rvn_shard_key_counts(rvn_engine* h, uint32_t* counts) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    Engine& e = h->e;
    const u64 u = e.index.u;
    if (u == 0) return RVN_OK;
    if (!counts) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_key_counts: NULL argument");
    RVN_HIP(hipSetDevice(e.device));
    std::vector<u32> st(u + 1);
    RVN_HIP(hipMemcpy(st.data(), e.index.u_start.ptr, (u + 1) * 4, hipMemcpyDeviceToHost));
    // (members only: query-only entries of reads outside the index batch, kForeignFlag, are the front of their run; a run
    // without members reports 0 and the caller drops it)
    std::vector<u64> so(e.index.m);
    if (e.index.m) RVN_HIP(hipMemcpy(so.data(), e.index.s_org[e.index.cur].ptr, e.index.m * 8, hipMemcpyDeviceToHost));
    for (u64 i = 0; i < u; ++i) counts[i] = st[i + 1] - st[i] - run_foreign_prefix(so.data(), st[i], st[i + 1] - st[i]);
    return RVN_OK;
  });
}

int rvn_engine_set_occurrence(rvn_engine* h, uint32_t occurrence) {
  if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
  h->e.index.occurrence = occurrence;
  return RVN_OK;
}

int rvn_shard_join(rvn_engine* h, uint32_t n_reads_total, int avoid_equal, int avoid_symmetric, uint64_t* n_matches) {
  return rvn_shard_join_range(h, n_reads_total, avoid_equal, avoid_symmetric, 0, n_reads_total, n_matches);
}

int rvn_shard_join_range(rvn_engine* h, uint32_t n_reads_total, int avoid_equal, int avoid_symmetric, uint32_t query_first,
                         uint32_t query_last, uint64_t* n_matches) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !n_matches) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_join: NULL argument");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    e.shard_join_reads = n_reads_total;
    e.shard_join_matches = join_index_matches(e, n_reads_total, avoid_equal != 0, avoid_symmetric != 0, query_first, query_last);
    *n_matches = e.shard_join_matches;
    RVN_HIP(rvn_stream_sync(e.stream));
    return RVN_OK;
  });
}

int rvn_shard_join_fetch(rvn_engine* h, uint64_t* grp, uint64_t* pos, uint64_t* seg_off) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    const u64 H = e.shard_join_matches;
    if (H && grp) RVN_HIP(hipMemcpy(grp, e.m_grp[0].ptr, H * 8, hipMemcpyDeviceToHost));
    if (H && pos) RVN_HIP(hipMemcpy(pos, e.m_pos[0].ptr, H * 8, hipMemcpyDeviceToHost));
    if (seg_off)
      RVN_HIP(hipMemcpy(seg_off, e.seg_off.ptr, (static_cast<size_t>(e.shard_join_reads) + 1) * 8, hipMemcpyDeviceToHost));
    return RVN_OK;
  });
}

int rvn_shard_chain(rvn_engine* h, const rvn_reads* own, const uint64_t* grp, const uint64_t* pos,
                    const uint64_t* seg_off, uint64_t* n_overlaps) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !own || !seg_off || !n_overlaps) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_chain: NULL argument");
    Engine& e = h->e;
    const ReadsDev& r = own->r;
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    const u32 nr = r.n;
    const u64 H = seg_off[nr];
    if (H && (!grp || !pos)) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_chain: NULL matches");
    u64* d_seg = e.seg_off.get<u64>(static_cast<size_t>(nr) + 2);
    RVN_HIP(hipMemcpy(d_seg, seg_off, (static_cast<size_t>(nr) + 1) * 8, hipMemcpyHostToDevice));
    u64* g0 = e.m_grp[0].get<u64>(H + 1);
    u64* p0 = e.m_pos[0].get<u64>(H + 1);
    e.m_grp[1].reserve((H + 1) * 8);
    e.m_pos[1].reserve((H + 1) * 8);
    if (H) {
      RVN_HIP(hipMemcpy(g0, grp, H * 8, hipMemcpyHostToDevice));
      RVN_HIP(hipMemcpy(p0, pos, H * 8, hipMemcpyHostToDevice));
    }
    MapOut& out = e.map_out;
    out.first = 0;
    out.last = nr;
    out.n_query = 0;
    out.n_matches = H;
    out.n_intervals = out.n_overlaps = 0;
    for (u32 i = 0; i < nr; ++i) e.c_query_bases += r.h_len[i];
    chain_matches(e, r, 0, nr, H, out);
    e.c_intervals += out.n_intervals;
    RVN_HIP(rvn_stream_sync(e.stream));
    *n_overlaps = out.n_overlaps;
    return RVN_OK;
  });
}

int rvn_shard_piles_create(rvn_engine* h, const uint32_t* lengths, uint32_t n_reads_total, rvn_pass1** out) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !lengths || !out) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_piles_create: NULL argument");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    std::unique_ptr<rvn_pass1> p(new rvn_pass1(e));
    // metadata-only read set: piles need lengths and ids (== indices), not bases
    p->meta.reset(new ReadsDev());
    ReadsDev& meta = *p->meta;
    meta.n = n_reads_total;
    meta.h_len.assign(lengths, lengths + n_reads_total);
    meta.h_id.resize(n_reads_total);
    for (u32 i = 0; i < n_reads_total; ++i) meta.h_id[i] = i;
    meta.ids_are_indices = true;
    u32* d_len = meta.len.get<u32>(static_cast<size_t>(n_reads_total) + 1);
    u32* d_id = meta.id.get<u32>(static_cast<size_t>(n_reads_total) + 1);
    if (n_reads_total) {
      RVN_HIP(hipMemcpy(d_len, meta.h_len.data(), static_cast<size_t>(n_reads_total) * 4, hipMemcpyHostToDevice));
      RVN_HIP(hipMemcpy(d_id, meta.h_id.data(), static_cast<size_t>(n_reads_total) * 4, hipMemcpyHostToDevice));
    }
    piles_init(e, meta, p->ps);
    *out = p.release();
    return RVN_OK;
  });
}

// One flush (construct.cc:79-110) of a sharded pass: merge the Map outputs of the window into the piles, AddLayers, truncate.
int rvn_shard_piles_merge(rvn_pass1* p, const rvn_overlap* overlaps, uint64_t n, uint32_t kmax) {
  return guarded(p ? p->e : nullptr, [&]() -> int {
    if (!p || !p->meta || (n && !overlaps)) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_piles_merge: bad handle or NULL overlaps");
    Engine& e = *p->e;
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    const u32 n_reads_total = p->meta->n;
    // overlaps arrive in (query read k, emission) order; per-k offsets as Map would have produced them
    std::vector<u32> off(static_cast<size_t>(n_reads_total) + 1, 0);
    const Overlap* ov = reinterpret_cast<const Overlap*>(overlaps);
    u32 prev = 0;
    for (u64 i = 0; i < n; ++i) {
      if (ov[i].lhs_id >= n_reads_total || ov[i].rhs_id >= n_reads_total || ov[i].lhs_id < prev)
        return fail(RVN_EINVAL, "[raven_hip] rvn_shard_piles: overlaps must be ordered by lhs_id and ids < n_reads");
      prev = ov[i].lhs_id;
      ++off[ov[i].lhs_id + 1];
    }
    for (u32 i = 0; i < n_reads_total; ++i) off[i + 1] += off[i];
    MapOut mo;
    mo.first = 0;
    mo.last = n_reads_total;
    mo.n_overlaps = n;
    Overlap* d_ov = mo.ovl.get<Overlap>(n + 1);
    if (n) RVN_HIP(hipMemcpy(d_ov, ov, n * sizeof(Overlap), hipMemcpyHostToDevice));
    u32* d_off = mo.ovl_read_off.get<u32>(off.size());
    RVN_HIP(hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    piles_merge(e, *p->meta, mo, kmax, p->ps);
    RVN_HIP(rvn_stream_sync(e.stream));
    return RVN_OK;
  });
}

int rvn_shard_piles_merge_dev(rvn_pass1* p, const rvn_overlap* d_overlaps, const uint32_t* d_ovl_read_off, uint64_t n,
                              uint32_t kmax) {
  return guarded(p ? p->e : nullptr, [&]() -> int {
    if (!p || !p->meta || !d_ovl_read_off || (n && !d_overlaps))
      return fail(RVN_EINVAL, "[raven_hip] rvn_shard_piles_merge_dev: bad handle or NULL argument");
    Engine& e = *p->e;
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    const u32 n_reads_total = p->meta->n;
    MapOut mo;
    mo.first = 0;
    mo.last = n_reads_total;
    mo.n_overlaps = n;
    Overlap* d_ov = mo.ovl.get<Overlap>(n + 1);
    if (n) RVN_HIP(hipMemcpyAsync(d_ov, d_overlaps, n * sizeof(Overlap), hipMemcpyDeviceToDevice, e.stream));
    u32* d_off = mo.ovl_read_off.get<u32>(static_cast<size_t>(n_reads_total) + 1);
    RVN_HIP(hipMemcpyAsync(d_off, d_ovl_read_off, (static_cast<size_t>(n_reads_total) + 1) * 4, hipMemcpyDeviceToDevice,
                           e.stream));
    piles_merge(e, *p->meta, mo, kmax, p->ps);
    RVN_HIP(rvn_stream_sync(e.stream));
    return RVN_OK;
  });
}

int rvn_shard_piles(rvn_engine* h, const uint32_t* lengths, uint32_t n_reads_total, const rvn_overlap* overlaps,
                    uint64_t n, uint32_t kmax, rvn_pass1** out) {
  if (!out) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_piles: NULL argument");
  rvn_pass1* p = nullptr;
  int rc = rvn_shard_piles_create(h, lengths, n_reads_total, &p);
  if (rc != RVN_OK) return rc;
  rc = rvn_shard_piles_merge(p, overlaps, n, kmax);
  if (rc != RVN_OK) {
    rvn_pass1_destroy(p);
    return rc;
  }
  *out = p;
  return RVN_OK;
}

// ---- device-pointer variants: the exchange buffers of the sharded pass stay in HBM (torch CUDA tensors) ----
namespace {
__global__ void widen_u32_u64_kernel(const u32* __restrict__ src, u64* __restrict__ dst, u64 n) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
__global__ void narrow_u64_u32_kernel(const u64* __restrict__ src, u32* __restrict__ dst, u64 n) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = static_cast<u32>(src[i]);
}
}  // namespace

int rvn_shard_sketch_fetch_dev(rvn_engine* h, uint64_t* d_values, uint64_t* d_origins) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    Engine& e = h->e;
    Sketch& s = e.shard_sketch_minhash ? e.index_sketch : e.raw_sketch;
    RVN_HIP(hipSetDevice(e.device));
    if (s.count == 0) return RVN_OK;
    if (!d_values || !d_origins) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_sketch_fetch_dev: NULL argument");
    if (e.val64) RVN_HIP(hipMemcpyAsync(d_values, s.val.ptr, s.count * 8, hipMemcpyDeviceToDevice, e.stream));
    else widen_u32_u64_kernel<<<static_cast<u32>((s.count + 255) / 256), 256, 0, e.stream>>>(s.val.as<u32>(), d_values, s.count);
    RVN_HIP(hipMemcpyAsync(d_origins, s.org.ptr, s.count * 8, hipMemcpyDeviceToDevice, e.stream));
    RVN_HIP(rvn_stream_sync(e.stream));
    return RVN_OK;
  });
}

int rvn_shard_index_build_dev(rvn_engine* h, const uint64_t* d_values, const uint64_t* d_origins, uint64_t n,
                              int all_query, uint64_t n_flagged) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || (n && (!d_values || !d_origins))) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_index_build_dev: NULL argument");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    Sketch& sk = e.index_sketch;
    sk.first = 0;
    sk.last = 0;
    sk.count = n;
    if (e.val64) {
      u64* dv = sk.val.get<u64>(n + 1);
      if (n) RVN_HIP(hipMemcpyAsync(dv, d_values, n * 8, hipMemcpyDeviceToDevice, e.stream));
    } else {
      u32* dv = sk.val.get<u32>(n + 1);
      if (n) narrow_u64_u32_kernel<<<static_cast<u32>((n + 255) / 256), 256, 0, e.stream>>>(d_values, dv, n);
    }
    u64* dorg = sk.org.get<u64>(n + 1);
    if (n) RVN_HIP(hipMemcpyAsync(dorg, d_origins, n * 8, hipMemcpyDeviceToDevice, e.stream));
    e.c_index_min += n;
    index_build(e, sk, false);
    e.c_index_keys += e.index.u;
    e.index.has_query_flags = !all_query;
    e.index.all_query = all_query != 0;
    e.join_query_count = all_query ? n : n_flagged;
    RVN_HIP(rvn_stream_sync(e.stream));
    return RVN_OK;
  });
}

int rvn_shard_key_histogram(rvn_engine* h, uint64_t* hist, uint32_t* over, uint32_t over_cap, uint32_t* n_over) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !hist || !n_over) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_key_histogram: NULL argument");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    std::vector<u64> hv;
    std::vector<u32> ov;
    index_key_histogram(e, hv, ov);
    for (size_t i = 0; i < hv.size(); ++i) hist[i] = hv[i];
    *n_over = static_cast<u32>(ov.size());
    if (ov.size() > over_cap) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_key_histogram: overflow list does not fit");
    for (size_t i = 0; i < ov.size(); ++i) over[i] = ov[i];
    return RVN_OK;
  });
}

int rvn_shard_join_fetch_dev(rvn_engine* h, uint64_t* d_grp, uint64_t* d_pos, uint64_t* d_seg_off) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    const u64 H = e.shard_join_matches;
    if (H && d_grp) RVN_HIP(hipMemcpyAsync(d_grp, e.m_grp[0].ptr, H * 8, hipMemcpyDeviceToDevice, e.stream));
    if (H && d_pos) RVN_HIP(hipMemcpyAsync(d_pos, e.m_pos[0].ptr, H * 8, hipMemcpyDeviceToDevice, e.stream));
    if (d_seg_off)
      RVN_HIP(hipMemcpyAsync(d_seg_off, e.seg_off.ptr, (static_cast<size_t>(e.shard_join_reads) + 1) * 8,
                             hipMemcpyDeviceToDevice, e.stream));
    RVN_HIP(rvn_stream_sync(e.stream));
    return RVN_OK;
  });
}

int rvn_shard_chain_dev(rvn_engine* h, const rvn_reads* own, const uint64_t* d_grp, const uint64_t* d_pos,
                        const uint64_t* d_seg_off, uint64_t n_matches, uint64_t* n_overlaps) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !own || !d_seg_off || !n_overlaps) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_chain_dev: NULL argument");
    Engine& e = h->e;
    const ReadsDev& r = own->r;
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    const u32 nr = r.n;
    const u64 H = n_matches;
    if (H && (!d_grp || !d_pos)) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_chain_dev: NULL matches");
    u64* d_seg = e.seg_off.get<u64>(static_cast<size_t>(nr) + 2);
    RVN_HIP(hipMemcpyAsync(d_seg, d_seg_off, (static_cast<size_t>(nr) + 1) * 8, hipMemcpyDeviceToDevice, e.stream));
    u64* g0 = e.m_grp[0].get<u64>(H + 1);
    u64* p0 = e.m_pos[0].get<u64>(H + 1);
    e.m_grp[1].reserve((H + 1) * 8);
    e.m_pos[1].reserve((H + 1) * 8);
    if (H) {
      RVN_HIP(hipMemcpyAsync(g0, d_grp, H * 8, hipMemcpyDeviceToDevice, e.stream));
      RVN_HIP(hipMemcpyAsync(p0, d_pos, H * 8, hipMemcpyDeviceToDevice, e.stream));
    }
    MapOut& out = e.map_out;
    out.first = 0;
    out.last = nr;
    out.n_query = 0;
    out.n_matches = H;
    out.n_intervals = out.n_overlaps = 0;
    for (u32 i = 0; i < nr; ++i) e.c_query_bases += r.h_len[i];
    chain_matches(e, r, 0, nr, H, out);
    e.c_intervals += out.n_intervals;
    RVN_HIP(rvn_stream_sync(e.stream));
    *n_overlaps = out.n_overlaps;
    return RVN_OK;
  });
}

int rvn_engine_map_fetch_dev(rvn_engine* h, rvn_overlap* d_overlaps, uint32_t* d_read_offsets) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    MapOut& m = h->e.map_out;
    RVN_HIP(hipSetDevice(h->e.device));
    if (d_overlaps && m.n_overlaps)
      RVN_HIP(hipMemcpyAsync(d_overlaps, m.ovl.ptr, m.n_overlaps * sizeof(Overlap), hipMemcpyDeviceToDevice, h->e.stream));
    if (d_read_offsets)
      RVN_HIP(hipMemcpyAsync(d_read_offsets, m.ovl_read_off.ptr, (static_cast<size_t>(m.last - m.first) + 1) * 4,
                             hipMemcpyDeviceToDevice, h->e.stream));
    RVN_HIP(rvn_stream_sync(h->e.stream));
    return RVN_OK;
  });
}

int rvn_shard_piles_dev(rvn_engine* h, const uint32_t* lengths, uint32_t n_reads_total, const rvn_overlap* d_overlaps,
                        const uint32_t* d_ovl_read_off, uint64_t n, uint32_t kmax, rvn_pass1** out) {
  if (!out) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_piles_dev: NULL argument");
  rvn_pass1* p = nullptr;
  int rc = rvn_shard_piles_create(h, lengths, n_reads_total, &p);
  if (rc != RVN_OK) return rc;
  rc = rvn_shard_piles_merge_dev(p, d_overlaps, d_ovl_read_off, n, kmax);
  if (rc != RVN_OK) {
    rvn_pass1_destroy(p);
    return rc;
  }
  *out = p;
  return RVN_OK;
}

// ---- partition / regroup steps of the sharded pass on device pointers (shard.hip) ----
int rvn_shard_split_minimizers_dev(rvn_engine* h, const uint64_t* d_values, const uint64_t* d_origins, uint64_t n,
                                   uint32_t world, uint64_t* d_values_out, uint64_t* d_origins_out, uint64_t* counts) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !counts || world == 0 || world > 16 || (n && (!d_values || !d_origins || !d_values_out || !d_origins_out)))
      return fail(RVN_EINVAL, "[raven_hip] rvn_shard_split_minimizers_dev: bad argument");
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    shard_split_minimizers(h->e, d_values, d_origins, n, world, d_values_out, d_origins_out, counts);
    return RVN_OK;
  });
}

int rvn_shard_split_overlaps_dev(rvn_engine* h, const rvn_overlap* d_overlaps, uint64_t n, const uint32_t* bounds,
                                 uint32_t world, uint32_t self, rvn_overlap* d_out, uint64_t* counts) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !counts || !bounds || world == 0 || world > 16 || self >= world || (n && (!d_overlaps || !d_out)))
      return fail(RVN_EINVAL, "[raven_hip] rvn_shard_split_overlaps_dev: bad argument");
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    shard_split_overlaps(h->e, reinterpret_cast<const Overlap*>(d_overlaps), n, bounds, world, self,
                         reinterpret_cast<Overlap*>(d_out), counts);
    return RVN_OK;
  });
}

int rvn_shard_count_flagged_dev(rvn_engine* h, const uint64_t* d_origins, uint64_t n, uint64_t* count) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !count || (n && !d_origins)) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_count_flagged_dev: bad argument");
    RVN_HIP(hipSetDevice(h->e.device));
    *count = shard_count_flagged(h->e, d_origins, n);
    return RVN_OK;
  });
}

int rvn_shard_adjacent_diff_dev(rvn_engine* h, const uint64_t* d_seg_off, uint64_t n, uint64_t* d_counts) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || (n && (!d_seg_off || !d_counts))) return fail(RVN_EINVAL, "[raven_hip] rvn_shard_adjacent_diff_dev: bad argument");
    RVN_HIP(hipSetDevice(h->e.device));
    shard_adjacent_diff(h->e, d_seg_off, n, d_counts);
    RVN_HIP(rvn_stream_sync(h->e.stream));
    return RVN_OK;
  });
}

int rvn_shard_regroup_dev(rvn_engine* h, uint32_t world, const uint64_t* const* d_counts, const uint64_t* const* d_group,
                          const uint64_t* const* d_positions, const uint64_t* n_per_source, uint32_t n_reads,
                          uint64_t* d_seg_off, uint64_t* d_group_out, uint64_t* d_positions_out) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || world == 0 || world > 16 || !d_counts || !d_group || !d_positions || !n_per_source || !d_seg_off)
      return fail(RVN_EINVAL, "[raven_hip] rvn_shard_regroup_dev: bad argument");
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    shard_regroup(h->e, world, d_counts, d_group, d_positions, n_per_source, n_reads, d_seg_off, d_group_out, d_positions_out);
    RVN_HIP(rvn_stream_sync(h->e.stream));
    return RVN_OK;
  });
}

int rvn_shard_piles_merge_parts_dev(rvn_pass1* p, uint32_t n_parts, const rvn_overlap* const* d_parts,
                                    const uint64_t* n_per_part, uint32_t kmax) {
  return guarded(p ? p->e : nullptr, [&]() -> int {
    if (!p || !p->meta || (n_parts && (!d_parts || !n_per_part)))
      return fail(RVN_EINVAL, "[raven_hip] rvn_shard_piles_merge_parts_dev: bad handle or NULL argument");
    Engine& e = *p->e;
    RVN_HIP(hipSetDevice(e.device));
    UseTimers ut(e);
    const u32 n_reads_total = p->meta->n;
    u64 n = 0;
    for (u32 i = 0; i < n_parts; ++i) n += n_per_part[i];
    MapOut mo;
    mo.first = 0;
    mo.last = n_reads_total;
    mo.n_overlaps = n;
    Overlap* d_ov = mo.ovl.get<Overlap>(n + 1);
    u64 at = 0;
    for (u32 i = 0; i < n_parts; ++i) {  // parts in ascending lhs-owner order: the list stays grouped by lhs read
      if (n_per_part[i])
        RVN_HIP(hipMemcpyAsync(d_ov + at, d_parts[i], n_per_part[i] * sizeof(Overlap), hipMemcpyDeviceToDevice, e.stream));
      at += n_per_part[i];
    }
    u32* d_off = mo.ovl_read_off.get<u32>(static_cast<size_t>(n_reads_total) + 1);
    shard_lhs_offsets(e, d_ov, n, n_reads_total, d_off);
    piles_merge(e, *p->meta, mo, kmax, p->ps);
    RVN_HIP(rvn_stream_sync(e.stream));
    return RVN_OK;
  });
}

int rvn_engine_set_option(rvn_engine* h, const char* name, int64_t value, int64_t* previous) {
  if (!h) return fail(RVN_EINVAL, "[raven_hip] rvn_engine_set_option: engine == NULL");
  long long* slot = engine_option(h->e.opt, name);
  // -1 = "the built-in default" for every option (the one way to get poa_rows_min_windows' default back: its 0 means
  // "every batch"); any other negative value is refused
  if (!slot || value < -1)
    return fail(RVN_EINVAL, std::string("[raven_hip] rvn_engine_set_option: unknown option or value below -1 (options: ") +
                                engine_option_names() + ")");
  const bool is_rows = slot == &h->e.opt.poa_rows_min_windows;
  if (slot == &h->e.opt.io_ring && value == 1)
    return fail(RVN_EINVAL, "[raven_hip] rvn_engine_set_option: io_ring needs at least 2 slabs in flight (0 or -1: the default)");
  std::lock_guard<std::recursive_mutex> lk(h->e.mu);
  if (previous) *previous = *slot < 0 ? (is_rows ? static_cast<int>(kPoaRowsMinWindowsDefault) : -1) : *slot;  // (the value the default stands for where it has one)
  *slot = value == -1 ? ((is_rows || slot == &h->e.opt.polish_sketch_cache_mb) ? -1 : 0) : value;
  return RVN_OK;
}

uint64_t rvn_polish_set_chunk_windows(rvn_engine* h, uint64_t windows) {
  if (!h) return 0;
  const uint64_t prev = h->e.polish_chunk_windows;
  h->e.polish_chunk_windows = windows;
  return prev;
}

int rvn_polish_target_reads(const rvn_engine* h, uint32_t* counts, uint32_t n_targets) {
  if (!h || !counts || n_targets != h->e.polish_target_reads.size())
    return fail(RVN_EINVAL, "[raven_hip] rvn_polish_target_reads: no polishing round with that many targets");
  for (uint32_t i = 0; i < n_targets; ++i) counts[i] = h->e.polish_target_reads[i];
  return RVN_OK;
}

void rvn_poa_phase_cycles(const rvn_engine* h, uint64_t out[6]) {
  for (int i = 0; i < 6; ++i) out[i] = h ? h->e.poa_phase_cycles[i] : 0;
}

int rvn_poa_set_mode(rvn_engine* h, int mode) {
  if (!h) return -1;
  const int prev = h->e.poa_mode;
  if ((mode >= 0 && mode <= 4) || mode == 9) h->e.poa_mode = mode;
  return prev;
}

uint32_t rvn_poa_fallback_windows(const rvn_engine* h) { return h ? h->e.poa_fallback_windows : 0; }
uint32_t rvn_poa_wide_windows(const rvn_engine* h) { return h ? h->e.poa_wide_windows : 0; }
uint32_t rvn_poa_narrow_windows(const rvn_engine* h) { return h ? h->e.poa_narrow_windows : 0; }

int rvn_engine_sketch(rvn_engine* h, const rvn_reads* r, uint32_t first, uint32_t last, int minhash, uint64_t* count) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !r || first > last || last > r->r.n) return fail(RVN_EINVAL, "[raven_hip] rvn_engine_sketch: bad range");
    RVN_HIP(hipSetDevice(h->e.device));
    UseTimers ut(h->e);
    sketch_range(h->e, r->r, first, last, minhash != 0, h->e.query_sketch);
    RVN_HIP(rvn_stream_sync(h->e.stream));
    if (count) *count = h->e.query_sketch.count;
    return RVN_OK;
  });
}

namespace {
int fetch_values(Engine& e, const DevBuf& val, u64 n, uint64_t* values) {
  if (!values || n == 0) return RVN_OK;
  if (e.val64) {
    RVN_HIP(hipMemcpy(values, val.ptr, n * 8, hipMemcpyDeviceToHost));
  } else {
    std::vector<u32> tmp(n);
    RVN_HIP(hipMemcpy(tmp.data(), val.ptr, n * 4, hipMemcpyDeviceToHost));
    for (u64 i = 0; i < n; ++i) values[i] = tmp[i];
  }
  return RVN_OK;
}
}  // namespace

int rvn_engine_sketch_fetch(rvn_engine* h, uint64_t* values, uint64_t* origins, uint32_t* read_offsets) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    Engine& e = h->e;
    Sketch& s = e.query_sketch;
    RVN_HIP(hipSetDevice(e.device));
    fetch_values(e, s.val, s.count, values);
    if (origins && s.count) RVN_HIP(hipMemcpy(origins, s.org.ptr, s.count * 8, hipMemcpyDeviceToHost));
    if (read_offsets)
      RVN_HIP(hipMemcpy(read_offsets, s.read_off.ptr, (static_cast<size_t>(s.last - s.first) + 1) * 4,
                        hipMemcpyDeviceToHost));
    return RVN_OK;
  });
}

int rvn_engine_index_size(const rvn_engine* h, uint64_t* n_minimizers, uint64_t* n_keys) {
  if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
  if (n_minimizers) *n_minimizers = h->e.index.m;
  if (n_keys) *n_keys = h->e.index.u;
  return RVN_OK;
}

int rvn_engine_index_fetch(rvn_engine* h, uint64_t* values, uint64_t* origins) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    Engine& e = h->e;
    Index& ix = e.index;
    RVN_HIP(hipSetDevice(e.device));
    fetch_values(e, ix.s_val[ix.cur], ix.m, values);
    if (origins && ix.m) {
      RVN_HIP(hipMemcpy(origins, ix.s_org[ix.cur].ptr, ix.m * 8, hipMemcpyDeviceToHost));
      // (both flag bits: after a multi-batch shard build the index also holds query-only entries of earlier batches'
      // reads — they come back as plain id << 32 | pos << 1 | strand like the members)
      for (u64 i = 0; i < ix.m; ++i) origins[i] &= ~(kQueryFlag | kForeignFlag);
    }
    return RVN_OK;
  });
}

int rvn_engine_counters(const rvn_engine* h, uint64_t out[8]) {
  if (!h || !out) return fail(RVN_EINVAL, "[raven_hip] NULL argument");
  const Engine& e = h->e;
  out[0] = e.c_index_bases;
  out[1] = e.c_index_min;
  out[2] = e.c_index_keys;
  out[3] = e.c_query_bases;
  out[4] = e.c_query_min;
  out[5] = e.c_matches;
  out[6] = e.c_overlaps;
  out[7] = e.c_intervals;
  return RVN_OK;
}

int rvn_engine_num_stages(void) { return StageTimes::kNum; }
const char* rvn_engine_stage_name(int s) { return (s >= 0 && s < StageTimes::kNum) ? kStageNames[s] : ""; }

int rvn_engine_stage_ms(const rvn_engine* h, double* ms, uint64_t* launches, int n) {
  if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
  for (int i = 0; i < n && i < StageTimes::kNum; ++i) {
    if (ms) ms[i] = h->e.times.ms[i];
    if (launches) launches[i] = h->e.times.launches[i];
  }
  return RVN_OK;
}

void rvn_engine_reset_stats(rvn_engine* h) {
  if (!h) return;
  Engine& e = h->e;
  e.times = StageTimes();
  e.ktimers.reset();
  e.c_index_bases = e.c_index_min = e.c_index_keys = e.c_query_bases = e.c_query_min = e.c_matches = e.c_overlaps =
      e.c_intervals = 0;
  e.poa_cells_full = e.poa_cells_band = e.poa_calls = 0;
}

void rvn_poa_work(const rvn_engine* h, uint64_t out[3]) {
  if (!h || !out) return;
  out[0] = h->e.poa_cells_full;
  out[1] = h->e.poa_cells_band;
  out[2] = h->e.poa_calls;
}

void rvn_engine_set_timing(rvn_engine* h, int enabled) {
  if (h) h->e.timing = enabled != 0;
}

void rvn_engine_set_kernel_timing(rvn_engine* h, int enabled) {
  if (h) h->e.ktimers.enabled = enabled != 0;
}
int rvn_engine_num_kernel_sites(void) { return kKNumSites; }
const char* rvn_engine_kernel_site_name(int i) { return (i >= 0 && i < kKNumSites) ? kKernelSiteNames[i] : ""; }
int rvn_engine_kernel_ms(rvn_engine* h, double* ms, uint64_t* launches, int n) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h) return fail(RVN_EINVAL, "[raven_hip] NULL engine");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    RVN_HIP(rvn_stream_sync(e.stream));
    e.ktimers.resolve();
    for (int i = 0; i < n && i < kKNumSites; ++i) {
      if (ms) ms[i] = e.ktimers.ms[i];
      if (launches) launches[i] = e.ktimers.launches[i];
    }
    return RVN_OK;
  });
}

#ifdef RVN_TEST_HOOKS
// ---- host test hooks ---------------------------------------------------------------------------
uint64_t rvn_test_hash(uint64_t key, uint32_t k, int use32) {
  const u64 mask = (1ULL << (2 * k)) - 1;
  if (use32) return hash32(static_cast<u32>(key), static_cast<u32>(mask));
  return hash64(key, mask);
}

int rvn_test_canonical(const uint64_t* words, uint32_t pos, uint32_t k, int use32, uint64_t* value,
                       uint32_t* strand) {
  const u64 mask = (1ULL << (2 * k)) - 1;
  const u32 bit = 2 * pos;
  const u64 x = extract_bits(words[bit >> 6], words[(bit >> 6) + 1], bit & 63, mask);
  unsigned st = 0;
  bool ok;
  if (use32) {
    u32 v = 0;
    ok = canonical_hash<u32>(x, k, mask, &v, &st);
    *value = v;
  } else {
    u64 v = 0;
    ok = canonical_hash<u64>(x, k, mask, &v, &st);
    *value = v;
  }
  *strand = st;
  return ok ? 1 : 0;
}
#endif  // RVN_TEST_HOOKS

int rvn_polish_fetch_layers(rvn_engine* h, uint32_t* out, uint64_t cap, uint64_t* n_out) {
  return guarded(h ? &h->e : nullptr, [&]() -> int {
    if (!h || !n_out) return fail(RVN_EINVAL, "[raven_hip] rvn_polish_fetch_layers: NULL argument");
    Engine& e = h->e;
    RVN_HIP(hipSetDevice(e.device));
    const u32 nw = e.polish_last_windows;
    const u64 nl = e.polish_last_layers;
    std::vector<PoaWindow> wins(nw);
    std::vector<PoaLayer> lays(nl);
    std::vector<u8> ok(nl, 1);
    if (nw) RVN_HIP(hipMemcpy(wins.data(), e.pl_wins.ptr, nw * sizeof(PoaWindow), hipMemcpyDeviceToHost));
    if (nl) RVN_HIP(hipMemcpy(lays.data(), e.pl_lays.ptr, nl * sizeof(PoaLayer), hipMemcpyDeviceToHost));
    if (nl && e.polish_last_has_ok) RVN_HIP(hipMemcpy(ok.data(), e.pl_ok.ptr, nl, hipMemcpyDeviceToHost));
    const std::vector<u64>& ro = e.polish_last_read_off;
    u64 n = 0;
    for (u32 i = 0; i < nw; ++i) {
      for (u32 x = 1; x < wins[i].n_layers; ++x) {  // layer 0 = backbone
        const u64 li = static_cast<u64>(wins[i].layer_first) + x;
        if (!ok[li]) continue;
        const PoaLayer& L = lays[li];
        if (out && n < cap) {
          const u64 read = static_cast<u64>(std::upper_bound(ro.begin(), ro.end(), L.code_off) - ro.begin()) - 1;
          uint32_t* o = out + 7 * n;
          o[0] = static_cast<uint32_t>(e.polish_last_w0 + i);
          o[1] = static_cast<uint32_t>(read);
          o[2] = L.q_begin;
          o[3] = L.len;
          o[4] = L.begin;
          o[5] = L.end;
          o[6] = (L.flags & kLayerRc) ? 1u : 0u;
        }
        ++n;
      }
    }
    *n_out = n;
    return RVN_OK;
  });
}

#ifdef RVN_TEST_HOOKS
// The host half of rvn_reads_load (io_text.h: member cut + inflate pool + record scanner) without a device: the kept
// text is assembled in host memory exactly as the H2D copies would lay it out in HBM, then cut into the records' fields.
// Outputs are malloc'ed (rvn_free): bases and qualities back to back, lengths, names separated by '\n';
// info[8] = {gzip, streaming, members, threads, restarted, loop microseconds, scan microseconds, fast single-stream decoder}.
int rvn_test_parse_file(const char* path, int fastq, uint32_t threads, int force_streaming, uint64_t slab_bytes,
                        uint8_t** bases, uint8_t** quals, uint32_t** lengths, uint32_t* n_records, char** names,
                        uint32_t* info) {
  return guarded([&]() -> int {
    if (!path || !bases || !quals || !lengths || !n_records || !names) return fail(RVN_EINVAL, "[raven_hip] NULL argument");
    for (int attempt = 0; attempt < 2; ++attempt) {
      try {
        io::SourceOptions opt;
        opt.threads = threads;
        opt.force_streaming = force_streaming != 0 || attempt == 1;
        if (slab_bytes) opt.slab_bytes = slab_bytes;
        io::TextSource src(path, opt);
        io::RecordScanner sc(fastq != 0);
        std::vector<u8> text;
        std::vector<io::TextRecord> recs;
        std::vector<std::string> nm;
        u8* slab = nullptr;
        u64 n = 0;
        bool first = true;
        const bool timing_only = knob("RVN_TEST_IO_TIMING_ONLY") != nullptr;  // records then come back empty
        const auto t_loop = std::chrono::steady_clock::now();
        double scan_s = 0;
        while (src.next(&slab, &n)) {
          const u8* run = nullptr;
          u64 run_len = 0, run_base = 0;
          const auto t_scan = std::chrono::steady_clock::now();
          sc.scan(slab, n, &run, &run_len, &run_base, recs, nm);
          scan_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_scan).count();
          if (!timing_only) {
            if (text.size() < run_base + run_len) text.resize(run_base + run_len);
            if (run_len) std::memcpy(text.data() + run_base, run, run_len);
          }
          if (!first) src.release();
          first = false;
        }
        const double loop_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
        u8 extra = 0;
        const u64 at = sc.text_end();
        if (sc.finish(recs, nm, &extra)) {
          text.resize(std::max<u64>(text.size(), at + 1));
          text[at] = extra;
        }
        u64 total = 0, nb = 0;
        for (const io::TextRecord& r : recs) total += r.len;
        for (const std::string& x : nm) nb += x.size() + 1;
        u8* b = static_cast<u8*>(std::malloc(total + 1));
        u8* q = static_cast<u8*>(std::malloc(total + 1));
        uint32_t* l = static_cast<uint32_t*>(std::malloc((recs.size() + 1) * 4));
        char* names_out = static_cast<char*>(std::malloc(nb + 1));
        if (!b || !q || !l || !names_out) return fail(RVN_ENOMEM, "[raven_hip] out of memory");
        u64 o = 0, no = 0;
        if (timing_only) recs.clear();
        for (size_t i = 0; i < recs.size(); ++i) {
          std::memcpy(b + o, text.data() + recs[i].seq_off, recs[i].len);
          if (fastq) std::memcpy(q + o, text.data() + recs[i].qual_off, recs[i].len);
          l[i] = static_cast<uint32_t>(recs[i].len);
          o += recs[i].len;
          std::memcpy(names_out + no, nm[i].data(), nm[i].size());
          no += nm[i].size();
          names_out[no++] = '\n';
        }
        names_out[no] = 0;
        *bases = b;
        *quals = q;
        *lengths = l;
        *n_records = static_cast<uint32_t>(recs.size());
        *names = names_out;
        if (info) {
          info[0] = src.gzip();
          info[1] = src.streaming();
          info[2] = src.members();
          info[3] = src.threads();
          info[4] = static_cast<uint32_t>(attempt);
          info[5] = static_cast<uint32_t>(loop_s * 1e6);  // inflate + scan + assembling the kept text, microseconds
          info[6] = static_cast<uint32_t>(scan_s * 1e6);  // of which inside RecordScanner::scan
          info[7] = src.fast_stream() ? 1 : 0;            // the single stream went through inflate_fast.h
        }
        return RVN_OK;
      } catch (const io::SpeculationFailed&) {
        if (attempt == 1) return fail(RVN_EINVAL, "[bioparser] error: corrupt or truncated file");
      } catch (const std::invalid_argument&) {  // (as reads_load: only the zlib attempt reports an error)
        if (attempt == 1) throw;
      }
    }
    return RVN_OK;
  });
}

// inflate_fast.h on ONE gzip member (header and trailer handled here): dst gets the text, out[4] = {bytes produced, bytes
// of the member consumed incl. the trailer, CRC-32 found in the trailer, ISIZE found}; chunk > 0: the output is produced
// through a buffer of that many bytes that is drained whenever it fills (the way the input path uses the decoder).
// Returns 0, RVN_EINVAL with the decoder's message for an invalid stream.
int rvn_test_inflate_fast(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t chunk, uint64_t* out) {
  return guarded([&]() -> int {
    if (!src || !dst || !out) return fail(RVN_EINVAL, "[raven_hip] NULL argument");
    const u64 hdr = io::gz_header_len(src, n, nullptr);
    if (!hdr) return fail(RVN_EINVAL, "not a gzip member");
    io::FastInflate dec;
    dec.reset(src + hdr, src + n);
    u64 produced = 0;
    if (chunk == 0) {
      u8* o = dst;
      const io::FastInflate::Status st = dec.run(dst, &o, dst + cap);
      produced = static_cast<u64>(o - dst);
      if (st == io::FastInflate::kOutputFull) return fail(RVN_EINVAL, "output buffer too small");
      if (st == io::FastInflate::kError) return fail(RVN_EINVAL, dec.error());
    } else {
      const u64 hist = 32768;
      std::vector<u8> buf(hist + chunk + io::FastInflate::kOutMargin);
      u8* base = buf.data();
      u8* o = base;  // (no history yet)
      const u8* valid_from = base;
      for (;;) {
        const io::FastInflate::Status st = dec.run(valid_from, &o, base + buf.size());
        const u8* from = valid_from == base && produced == 0 ? base : base + hist;
        // drain what is new: everything behind the history area (or the whole buffer the first time round)
        const u64 fresh = static_cast<u64>(o - from);
        if (produced + fresh > cap) return fail(RVN_EINVAL, "output buffer too small");
        std::memcpy(dst + produced, from, fresh);
        produced += fresh;
        if (st == io::FastInflate::kError) return fail(RVN_EINVAL, dec.error());
        if (st == io::FastInflate::kStreamEnd) break;
        // keep the last 32 KB in front
        const u64 have = static_cast<u64>(o - base);
        const u64 keep = std::min<u64>(hist, have);
        std::memmove(base + hist - keep, o - keep, keep);
        valid_from = base + hist - keep;
        o = base + hist;
      }
    }
    const u8* p = dec.input_position();
    if (p + 8 > src + n) return fail(RVN_EINVAL, "unexpected end of file");
    out[0] = produced;
    out[1] = static_cast<u64>(p + 8 - src);
    out[2] = p[0] | (static_cast<u64>(p[1]) << 8) | (static_cast<u64>(p[2]) << 16) | (static_cast<u64>(p[3]) << 24);
    out[3] = p[4] | (static_cast<u64>(p[5]) << 8) | (static_cast<u64>(p[6]) << 16) | (static_cast<u64>(p[7]) << 24);
    return RVN_OK;
  });
}

// freelist.h (the bookkeeping of the device arena) driven by a list of operations: ops[i] > 0 = allocate that many bytes
// (out[i] = offset, or -1 if no hole holds it), ops[i] <= 0 = give back the block allocated by operation -ops[i] (out[i] = 1,
// 0 if that was no block in use).  state[3] = {bytes free, largest hole, blocks in use} at the end.
int rvn_test_freelist(uint64_t size, uint64_t grain, const int64_t* ops, uint32_t n_ops, int64_t* out, uint64_t* state) {
  if (!ops || !out || !state) return RVN_EINVAL;
  rvn::FreeList fl;
  fl.reset(size, grain);
  std::vector<char> given_back(n_ops, 0);  // (a block is named by the operation that made it: its offset may have a new owner)
  for (uint32_t i = 0; i < n_ops; ++i) {
    if (ops[i] > 0) {
      size_t off = 0;
      out[i] = fl.alloc(static_cast<size_t>(ops[i]), &off) ? static_cast<int64_t>(off) : -1;
    } else {
      const uint64_t j = static_cast<uint64_t>(-ops[i]);
      const bool ok = j < i && ops[j] > 0 && out[j] >= 0 && !given_back[j] && fl.release(static_cast<size_t>(out[j]));
      if (ok) given_back[j] = 1;
      out[i] = ok ? 1 : 0;
    }
  }
  state[0] = fl.free_total();
  state[1] = fl.free_largest();
  state[2] = fl.in_use.size();
  return RVN_OK;
}

int rvn_test_nw_breakpoints(const uint64_t* t_words, uint32_t t_len, const uint64_t* r_words, uint32_t r_len,
                            uint32_t t_begin, uint32_t n, uint32_t q_begin, uint32_t m, int rc, uint32_t w, uint32_t k,
                            int force_r, uint32_t* recs, uint32_t* distance, uint32_t* band) {
  if (!t_words || !r_words || !recs || !distance || w == 0) return RVN_EINVAL;
  if (static_cast<u64>(t_begin) + n > t_len || static_cast<u64>(q_begin) + m > r_len) return RVN_EINVAL;
  static_assert(sizeof(NwWindowRec) == 32, "record layout");
  return nw_breakpoints_host(t_words, t_len, r_words, r_len, t_begin, n, q_begin, m, rc, w, k, force_r,
                             reinterpret_cast<NwWindowRec*>(recs), distance, band);
}

int rvn_test_low_complexity(const uint8_t* codes, uint32_t k) { return lc_kmer_passes(codes, k) ? 1 : 0; }

void rvn_test_std_sort_lendesc(uint64_t* data, uint64_t n) { std_sort(data, data + n, LenDesc()); }
void rvn_test_heap_sort_lendesc(uint64_t* data, uint64_t n) { intro::heap_sort(data, data + n, LenDesc()); }
#endif  // RVN_TEST_HOOKS

}  // extern "C"
