// common.h — host-side plumbing shared by the HIP translation units.
#pragma once

#include <chrono>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <new>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

namespace rvn {

using u8 = std::uint8_t;
using u16 = std::uint16_t;
using u32 = std::uint32_t;
using u64 = std::uint64_t;
using i32 = std::int32_t;
using i16 = std::int16_t;

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
// hipMalloc failed with the block pool already empty: the stage entry points hand back every scratch buffer of the
// engine and run the stage once more before they report it (engine.hip: guarded)
struct DeviceOutOfMemory : HipError {
  using HipError::HipError;
};

#define RVN_HIP(expr)                                                                            \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      throw ::rvn::HipError(std::string("[raven_hip] HIP error: ") + hipGetErrorString(_e) +    \
                            " at " __FILE__ ":" + std::to_string(__LINE__) + " (" #expr ")");    \
    }                                                                                            \
  } while (0)

#define RVN_LAUNCH_CHECK() RVN_HIP(hipGetLastError())

// Environment switches.  The PRODUCT library reads none that changes what a call computes or how it is scheduled: tuning
// a deployment may want is an engine option (rvn_engine_set_option, include/raven_hip.h), and everything else —
// diagnostics, A/B switches of experiments, "skip this stage" profiling aids — exists only in builds with
// -DRVN_DEBUG_KNOBS (libraven_hip_test.so, which the tools load through RVN_LIB_PATH): knob() is nullptr otherwise.
inline const char* knob(const char* name) {
#ifdef RVN_DEBUG_KNOBS
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// Device ARENA behind the grow-only buffers, switched on when a workload turns out not to fit the HBM with both of its
// phases' scratch alive (the HiFi one: 1.1 matches per read base; engine_release_scratch_if_tight).  Such a workload hands
// ~200 GB of scratch back and takes it again at every change of phase, and hipMalloc of FRESH memory runs at ~25 GB/s on
// MI355X (17 GB: 0.64 s; a traced step spent 14 s in three such calls).  With the arena that traffic is a first-fit
// search in a free list: one hipMalloc of (free memory - margin) when it starts, no driver call afterwards; a request
// the arena cannot hold falls through to the driver, and if that is out of memory too the stage entry point releases
// every scratch buffer and runs the stage once more (engine.hip: guarded).  Implemented in engine.hip.
namespace devpool {
bool active();                 // an arena exists on the current device
bool start(size_t bytes);      // one hipMalloc; false if the driver refuses
void* alloc(size_t bytes);     // nullptr: no arena, or no hole of that size
bool give_back(void* p);       // false: p is not an arena block (the caller hipFree's it)
size_t free_total();
size_t free_largest();
size_t size();
void stop();                   // back to the driver — only if no block of it is in use
}  // namespace devpool

// Growable device buffer; capacity only grows, so steady-state iterations do
// not touch the allocator.
struct DevBuf {
  void* ptr = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (ptr && !devpool::give_back(ptr)) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
  void reserve(size_t bytes) {
    if (bytes <= cap) return;
    release();
    size_t want = bytes + bytes / 8 + 256;
    static const bool trace = knob("RVN_DEBUG_MEM") != nullptr;  // allocator traffic of the grow-only buffers
    const auto t0 = std::chrono::steady_clock::now();
    const char* how = "arena";
    void* p = devpool::alloc(want);
    if (!p) {
      how = "driver";
      const hipError_t err = hipMalloc(&p, want);
      if (err == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        throw DeviceOutOfMemory("[raven_hip] HIP error: out of memory (device buffer of " + std::to_string(want >> 20) + " MB)");
      }
      RVN_HIP(err);
    }
    ptr = p;
    cap = want;
    if (trace && want >= (256ULL << 20))
      std::fprintf(stderr, "[raven_hip] DevBuf grows to %.2f GB (%.1f ms, %s)\n", want / 1e9,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), how);
  }
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(ptr);
  }
  template <typename T>
  T* get(size_t count) {
    reserve(count * sizeof(T));
    return reinterpret_cast<T*>(ptr);
  }
};

// Growable pinned host buffer (device -> host read-backs of bulk results at PCIe speed, no page-fault zeroing).
struct PinBuf {
  void* ptr = nullptr;
  size_t cap = 0;
  PinBuf() = default;
  PinBuf(const PinBuf&) = delete;
  PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf() {
    if (ptr) (void)hipHostFree(ptr);
  }
  template <typename T>
  T* get(size_t count) {
    const size_t bytes = count * sizeof(T);
    if (bytes > cap) {
      if (ptr) RVN_HIP(hipHostFree(ptr));
      ptr = nullptr;
      cap = 0;
      const size_t want = bytes + bytes / 4 + 4096;  // pinning is slow (~1-2 GB/s): leave room for the next round
      RVN_HIP(hipHostMalloc(&ptr, want, hipHostMallocDefault));
      cap = want;
    }
    return reinterpret_cast<T*>(ptr);
  }
};

// Growable plain host buffer, never initialised (bulk read-backs too large to be worth pinning: pinning runs at
// ~2 GB/s, a pageable copy at ~10 GB/s).
struct HostBuf {
  void* ptr = nullptr;
  size_t cap = 0;
  HostBuf() = default;
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  ~HostBuf() { std::free(ptr); }
  template <typename T>
  T* get(size_t count) {
    const size_t bytes = count * sizeof(T);
    if (bytes > cap) {
      std::free(ptr);
      cap = bytes + bytes / 4 + 4096;
      ptr = std::malloc(cap);
      if (!ptr) {
        cap = 0;
        throw std::bad_alloc();
      }
    }
    return reinterpret_cast<T*>(ptr);
  }
};

// ---- per-kernel-site timing (HIP events on the engine stream, no host sync while recording) ----
enum KernelSite {
  kKSketchCount, kKSketchWrite, kKMinhashSelect, kKCompactSketch, kKScan, kKRsBits, kKRsUpsweep, kKRsDownsweep,
  kKHeads, kKUnique, kKTable, kKOccHist, kKMatchCount, kKMatchEmit, kKSegSortGroup, kKIntervals,
  kKIntervalsGather, kKSegSortPos, kKChain, kKCompactOverlaps, kKPileKeys, kKPileCounts, kKPileBuild,
  kKAddLayers, kKTruncateSort, kKKeptWrite, kKGather, kKPileSortUp, kKPileSortDown, kKChainSmall, kKJoinCount, kKJoinEmit, kKEditBanded, kKEditFull, kKPoa, kKAddKmers, kKPoaBanded, kKPileTrim, kKNwForward, kKNwTraceback, kKEditLane, kKNwLane, kKBestOverlap, kKLayerBuild, kKStitch, kKPoaRows, kKNumSites
};
extern const char* const kKernelSiteNames[kKNumSites];

// hipStreamSynchronize that polls the stream for a while before it blocks: a blocking wait costs a wake-up latency of
// several milliseconds on some hosts, and a pass has ~60 short waits (size read-backs between stages) — at C4 they added
// up to 0.5 s per pass on such hosts although the kernels took 0.29 s.  Long waits fall back to the blocking call.
inline hipError_t rvn_stream_sync(hipStream_t s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(s);
    if (q != hipErrorNotReady) return q;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(100)) return hipStreamSynchronize(s);
  }
}

struct KernelTimers {
  bool enabled = false;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  struct Rec {
    int site;
    size_t e0, e1;
  };
  std::vector<Rec> recs;
  double ms[kKNumSites] = {};
  u64 launches[kKNumSites] = {};
  ~KernelTimers();
  size_t next_event();
  void resolve();  // call after the stream is synchronised
  void reset();
};
// timers of the engine currently executing on this host thread (nullptr = no timing)
extern thread_local KernelTimers* g_kernel_timers;

struct KernelScope {
  KernelTimers* kt;
  size_t e1 = 0;
  hipStream_t st = nullptr;
  // `on`: the stream the kernel is launched on when it is not the engine's main stream (the event pair must sit on it)
  explicit KernelScope(int site, hipStream_t on = nullptr) : kt(g_kernel_timers) {
    if (kt && kt->enabled) {
      st = on ? on : kt->stream;
      size_t e0 = kt->next_event();
      e1 = kt->next_event();
      kt->recs.push_back({site, e0, e1});
      RVN_HIP(hipEventRecord(kt->pool[e0], st));
    } else {
      kt = nullptr;
    }
  }
  ~KernelScope() {
    if (kt) (void)hipEventRecord(kt->pool[e1], st);
  }
};

// launch a kernel (or a few) attributed to `site`
#define RVN_KLAUNCH(site, ...)        \
  do {                                \
    ::rvn::KernelScope _ks(site);     \
    __VA_ARGS__;                      \
    RVN_LAUNCH_CHECK();               \
  } while (0)

// the same on another stream of the engine
#define RVN_KLAUNCH_ON(site, stream, ...)     \
  do {                                        \
    ::rvn::KernelScope _ks(site, stream);     \
    __VA_ARGS__;                              \
    RVN_LAUNCH_CHECK();                       \
  } while (0)

// 8 x u32 overlap record == biosoup::Overlap minus the alignment string.
// Bit 63 of an index origin marks "this minimizer is also a (minhash) query minimizer" (self-join path);
// read ids must therefore stay below 2^31.  Every consumer of an origin's id masks it.
// f(begin, end) over [0, n) on a few host threads (the per-job loops of a stage's host planning: 300 000 jobs at C4; the GPU
// is idle while they run).  Serial below `grain` items per thread.
template <class F>
inline void parallel_for(size_t n, size_t grain, F f) {
  size_t nt = std::min<size_t>(std::min<size_t>(std::thread::hardware_concurrency(), 16), grain ? n / grain : 1);
  if (nt <= 1) {
    f(static_cast<size_t>(0), n);
    return;
  }
  std::vector<std::thread> ths;
  ths.reserve(nt);
  for (size_t t = 0; t < nt; ++t) ths.emplace_back([=]() { f(n * t / nt, n * (t + 1) / nt); });
  for (std::thread& th : ths) th.join();
}

constexpr u64 kQueryFlag = 1ULL << 63;
// An entry that is a QUERY ONLY: a minimizer of a read that does not belong to the index batch at hand (sharded pass with
// more than one index batch: reads of earlier batches are mapped against every later batch's index, construct.cc:59-64
// inside the loop of :32-37).  Such entries take part in the sort so that the self-join finds them beside the index's
// runs, but they are no members: they do not count towards a key's occurrence and are nobody's match.  Their reads have
// smaller ids than every member's, and the sort is stable, so they are the FRONT of their run.
constexpr u64 kForeignFlag = 1ULL << 62;
constexpr u32 kMaxReadId = 1u << 30;                  // read ids are 30 bits: bits 63 / 62 of an origin word are the two flags
constexpr u64 kForeignPieceBases = 1ULL << 31;        // a query-only (foreign) sketch is taken in pieces of at most this many bases
__host__ __device__ inline u32 origin_id(u64 org) { return static_cast<u32>(org >> 32) & 0x3FFFFFFFu; }
// number of foreign entries at the front of the run [s, s + c) of a sorted origin array (binary search: the flag is
// monotone within a run)
__host__ __device__ inline u32 run_foreign_prefix(const u64* s_org, u32 s, u32 c) {
  if (c == 0 || !(s_org[s] & kForeignFlag)) return 0;
  u32 lo = 1, hi = c;  // first index without the flag lies in [lo, hi]
  while (lo < hi) {
    const u32 mid = (lo + hi) >> 1;
    if (s_org[s + mid] & kForeignFlag) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

struct Overlap {
  u32 lhs_id, lhs_begin, lhs_end, rhs_id, rhs_begin, rhs_end, score, strand;
};

static inline u32 div_up(u64 a, u64 b) { return static_cast<u32>((a + b - 1) / b); }

// ---- device-wide primitives (scan.hip / radix_sort.hip) ----------------------

// out[0..n] = exclusive prefix sums of in[0..n-1] (out[n] = total). `tmp` is scratch.
void exclusive_scan_u32_u64(const u32* in, u64* out, u64 n, DevBuf& tmp, hipStream_t s);
void exclusive_scan_u32_u32(const u32* in, u32* out, u64 n, DevBuf& tmp, hipStream_t s);
void exclusive_scan_u8_u32(const u8* in, u32* out, u64 n, DevBuf& tmp, hipStream_t s);

// Stable LSD radix sort of (key, value) pairs on key bits [0, key_bits).
// Ping-pongs between (k0,v0) and (k1,v1); returns 0 if the sorted data ends in
// (k0,v0), 1 if in (k1,v1).
int radix_sort_pairs_u32_u64(u32* k0, u32* k1, u64* v0, u64* v1, u64 n, int key_bits, DevBuf& tmp, DevBuf& tmp2, hipStream_t s,
                             int site_up = kKRsUpsweep, int site_down = kKRsDownsweep,
                             bool skip_constant_digits = true);
int radix_sort_pairs_u64_u64(u64* k0, u64* k1, u64* v0, u64* v1, u64 n, int key_bits, DevBuf& tmp, DevBuf& tmp2, hipStream_t s,
                             int site_up = kKRsUpsweep, int site_down = kKRsDownsweep,
                             bool skip_constant_digits = true);
int radix_sort_pairs_u32_u32(u32* k0, u32* k1, u32* v0, u32* v1, u64 n, int key_bits, DevBuf& tmp, DevBuf& tmp2, hipStream_t s,
                             int site_up = kKRsUpsweep, int site_down = kKRsDownsweep,
                             bool skip_constant_digits = true);

}  // namespace rvn
