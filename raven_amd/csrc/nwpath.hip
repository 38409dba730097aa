// nwpath.hip — kernels and host driver of the alignment-path stage of a polishing round (see nwpath.h): for every
// read's best overlap, the global alignment path against its target span and racon's per-window breakpoints
// (racon Overlap::find_breaking_points, reached from RavenLib/src/polish.cc:51).
//
//   nw_sweep_kernel<R, G>   the forward sweep (nwsweep.h).  A wave is split into 64 / G lane groups; every group owns one
//                           alignment (its ring of L <= G lanes, R blocks per lane), the groups of a wave take a bundle
//                           of 64 / G jobs of similar length and step in lockstep.  Persistent waves, longest jobs
//                           first.  Block state in registers, match masks in LDS; per step one ds_bpermute, one LDS
//                           read, the Myers update.  Leaves the exact distance (if <= k), the hs stream and the
//                           checkpoints.
//   nw_trace_kernel         the backward walk (nwtrace.h), one lane per alignment, strip of <= 33 columns in LDS.
//
// Host side: threshold k from the running error-rate estimate of the engine (first call: a generous default), the
// narrowest kernel variant whose ring holds the band of k; alignments whose distance exceeds k are repeated with 2k —
// the result is always the exact optimal path, the estimate only decides how much band is computed.  hs + ck of a launch
// are budgeted (RVN_NW_BUDGET_MB, default: a quarter of the free HBM, at most 64 GB); more jobs than fit go in chunks.
#include <algorithm>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <vector>

#include "engine.h"
#include "nwpath.h"
#include "nwsweep.h"
#include "nwtrace.h"
#include "wave.h"

namespace rvn {

namespace {

// kernel variants: (blocks per lane R, lanes per alignment G); narrow rings share a wave (64 / G alignments).  Ordered by
// the band they hold (G (64 R + 1) - 64 R offsets), i.e. by alignment length.  Several blocks per lane are only used
// where one block per lane cannot hold the band: variants (2, 8) ... (4, 32) — fewer instructions per block step on
// paper — were measured and lost (C4 sweep 178 ms with R = 1 wherever possible, 187 ms allowing R = 2, 225 ms allowing
// R = 4: more registers = fewer waves, a coarser band, costlier ring events).
constexpr int kNwLaneStripCols = 16;        // columns a walker's strip keeps in LDS (nw_trace_kernel<true>)
constexpr int kNwGroupLanes = 16;           // lanes per alignment of the group walk (four alignments per wave)
constexpr u32 kNwGroupWalkMaxJobs = 8192;  // a walk launch of at most this many alignments takes the group walk (two rounds of the
                                           // machine's 4 096 resident groups: beyond that a group's ~3x shorter latency per column loses
                                           // to the lane walk's sixteen times as many alignments in flight — tools/walk_ab.sh)
constexpr u32 kSideSweepMaxWaves = 2048;    // a sweep launch of at most this many waves goes beside the main stream's (enqueue)
constexpr u32 kLevels = 8;
const u32 kRs[kLevels] = {1, 1, 1, 1, 1, 2, 4, 8};
const u32 kGs[kLevels] = {4, 8, 16, 32, 64, 64, 64, 64};

template <int G>
__device__ __forceinline__ u32 group_max(u32 v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) {
    const u32 o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

template <int R>
constexpr int sweep_waves_per_simd() {
  return R == 1 ? 8 : (R == 2 ? 6 : (R == 4 ? 4 : 2));
}

template <int R, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sweep_waves_per_simd<R>())))
void nw_sweep_kernel(const NwJob* __restrict__ jobs, const u32* __restrict__ idx, u32 n_idx,
                     const u64* __restrict__ t_words, const u64* __restrict__ r_words, u32* __restrict__ hs,
                     NwPm* __restrict__ ck, u32* __restrict__ result, u32* __restrict__ status, u32* __restrict__ next) {
  constexpr int NG = 64 / G;
  __shared__ u64 s_peq[4][R * 4 * 64];
  const int lane = lane_id();
  const int group = lane / G, lig = lane % G, gbase = group * G;
  u64* peq = s_peq[threadIdx.x >> 6];
  for (;;) {
    // every lane takes part in the fetch (lane 0 adds the bundle size, the others 0; the compiler folds it into one
    // wave-level atomic): with the usual `if (lane == 0)` around it hipcc 7.2 threaded the G = 64 variant's control flow
    // through the readfirstlane and left lanes 1..63 spinning on bundle 0
    u32 q0 = atomicAdd(next, lane == 0 ? static_cast<u32>(NG) : 0u);
    q0 = static_cast<u32>(__builtin_amdgcn_readfirstlane(static_cast<int>(q0)));
    if (q0 >= n_idx) break;
    const u32 q = q0 + static_cast<u32>(group);
    const bool has = q < n_idx;
    NwSweepLane<R, 64> ln;
    u32 ji = 0, k = 0;
    int L = 0, n_steps = 0;
    u32* hs_j = hs;
    NwPm* ck_j = ck;
    if (has) {
      ji = idx[q];
      const NwJob J = jobs[ji];
      const NwGeo geo = nw_geo(J.n, J.m, J.k, R);
      ln.init(J, t_words, r_words, geo, lig, peq, lane);
      L = geo.L;
      n_steps = geo.n_steps;
      k = J.k;
      hs_j = hs + J.hs;
      ck_j = ck + J.ckpt;
    } else {
      ln.init_idle(peq, lane);
    }
    int n_steps_w = n_steps;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(n_steps_w, off, 64);
      n_steps_w = o > n_steps_w ? o : n_steps_w;
    }
    n_steps_w = __builtin_amdgcn_readfirstlane((n_steps_w + 15) & ~15);
    const int prev = lig == 0 ? gbase + (L > 0 ? L - 1 : 0) : lane - 1;
    const bool st_ok = has && lig < L;
    const int n_g = (n_steps + kNwHsSteps - 1) / kNwHsSteps, n_q = n_steps / kNwCkSteps;
    const u64 row = static_cast<u64>(L) * R;
    for (int t = 1; t <= n_steps_w; ++t) {
      const int x = __shfl(ln.xf, prev, 64);
      if (__ballot(ln.has_event(t)) != 0) {
        const int sp = __shfl(ln.sc, prev, 64);
        if (ln.has_event(t)) ln.event(t, x, sp);
      }
      ln.step(t, x);
      if ((t & 15) == 0) {
        const int gi = (t >> 4) - 1;
        if (st_ok && gi < n_g) {
          u32* p = hs_j + static_cast<u64>(gi) * row + static_cast<u64>(lig) * R;
#pragma unroll
          for (int r = 0; r < R; ++r) p[r] = ln.acc[r];
        }
        if ((t & 31) == 0) {
          const int qi = (t >> 5) - 1;
          if (st_ok && qi < n_q) {
            NwPm* p = ck_j + static_cast<u64>(qi) * row + static_cast<u64>(lig) * R;
#pragma unroll
            for (int r = 0; r < R; ++r) p[r] = NwPm{ln.Pv[r], ln.Mv[r]};
          }
        }
        ln.next_group(t);
      }
    }
    const u32 res1 = group_max<G>(ln.result);  // exactly one lane of the group holds D(n, m) + 1
    if (has && lig == 0) {
      const u32 res = res1 - 1u;
      result[ji] = res;
      status[ji] = (res1 != 0 && res <= k) ? 0u : 2u;  // 2: beyond the threshold, the host repeats the job with 2k
    }
  }
}

// STRIP_LDS: the walker's strip in LDS (33 KB per wave: four waves per CU) or in a per-wave scratch in HBM / L2
// ([column][lane], coalesced; occupancy limited by registers only)
template <bool STRIP_LDS, int SC>
__global__ __launch_bounds__(64) void nw_trace_kernel(const NwJob* __restrict__ jobs, const u32* __restrict__ idx, u32 n_idx,
                                                      const u64* __restrict__ t_words, const u64* __restrict__ r_words,
                                                      const u32* __restrict__ hs, const NwPm* __restrict__ ck,
                                                      const u32* __restrict__ result, u32* __restrict__ status, u32 w,
                                                      NwWindowRec* __restrict__ recs, u64* __restrict__ scratch) {
  // (SC = 16: a strip keeps sixteen columns, 17.4 KB of LDS per wave, nine waves per CU instead of four — nwtrace.h; for
  // launches of more waves than the machine holds with whole strips.  A walker pays ~15 % for the second visit of a
  // checkpoint interval, so launches that fit keep whole strips.)
  __shared__ u64 s_pv[STRIP_LDS ? (SC + 1) * 64 : 1];
  __shared__ u64 s_mv[STRIP_LDS ? (SC + 1) * 64 : 1];
  // one lane per alignment, a long dependent chain per column: these waves run beside the next chunk's sweep (which keeps
  // the VALUs full) and must not queue behind it for every instruction
  __builtin_amdgcn_s_setprio(3);
  const u32 q = blockIdx.x * 64 + threadIdx.x;
  if (q >= n_idx) return;
  const u32 ji = idx[q];
  if (status[ji] != 0) return;
  const NwJob J = jobs[ji];
  const NwGeo geo = nw_geo(J.n, J.m, J.k, J.R);
  u64* g_pv = scratch + static_cast<u64>(blockIdx.x) * (2 * kNwStripCols * 64);
  const NwStripMem<64> mem{STRIP_LDS ? s_pv : g_pv, STRIP_LDS ? s_mv : g_pv + kNwStripCols * 64, static_cast<int>(threadIdx.x)};
  status[ji] = static_cast<u32>(nw_trace_job<64, SC>(J, geo, t_words, r_words, hs + J.hs, ck + J.ckpt, mem, result[ji], w, recs));
}

// The group walk (nwtrace.h: NwGroupWalk): GL lanes per alignment, 64 / GL alignments per wave; strips in the same LDS
// array as nw_trace_kernel's (column = lane), the heads beside it: 36.9 KB per wave, four waves per CU.
template <int GL>
__global__ __launch_bounds__(64) void nw_trace_group_kernel(const NwJob* __restrict__ jobs, const u32* __restrict__ idx, u32 n_idx,
                                                            const u64* __restrict__ t_words, const u64* __restrict__ r_words,
                                                            const u32* __restrict__ hs, const NwPm* __restrict__ ck,
                                                            const u32* __restrict__ result, u32* __restrict__ status, u32 w,
                                                            NwWindowRec* __restrict__ recs) {
  constexpr int NG = 64 / GL;
  __shared__ u64 s_pv[kNwStripCols * 64];
  __shared__ u64 s_mv[kNwStripCols * 64];
  __shared__ NwStripHead s_heads[64];
  __builtin_amdgcn_s_setprio(3);
  const int lane = static_cast<int>(threadIdx.x), grp = lane / GL, t = lane % GL;
  const u32 q = blockIdx.x * NG + static_cast<u32>(grp);
  if (q >= n_idx) return;
  const u32 ji = idx[q];
  if (status[ji] != 0) return;
  const NwJob J = jobs[ji];
  const NwGeo geo = nw_geo(J.n, J.m, J.k, J.R);
  NwGroupWalk<64, GL> G;
  G.init(J, t_words, r_words, NwStripMem<64>{s_pv, s_mv, grp * GL}, s_heads + grp * GL, result[ji], w, recs);
  int bad = 0;
  while (!G.done()) {
    G.fill(J, geo, hs + J.hs, ck + J.ckpt, t);
    __threadfence_block();  // (one wave: the strips and heads of the group's lanes, visible to all of them)
    __builtin_amdgcn_wave_barrier();
    bad = G.walk_batch(geo, t == 0);
    __threadfence_block();  // (the next batch overwrites them)
    __builtin_amdgcn_wave_barrier();
    if (bad) break;
  }
  const int rcode = bad ? 1 : G.wk.finish(t == 0);
  if (t == 0) status[ji] = static_cast<u32>(rcode);
}

template <int R, int G>
void launch_sweep(Engine& e, hipStream_t s, const NwJob* d_jobs, const u32* d_idx, u32 n_idx, const ReadsDev& T, const ReadsDev& Rd,
                  u32* d_hs, NwPm* d_ck, u32* d_result, u32* d_status, u32* d_next) {
  if (n_idx == 0) return;
  constexpr u32 NG = 64 / G;
  const u32 bundles = (n_idx + NG - 1) / NG;
  // persistent waves: not every slot of the machine — the walks of the previous chunk run beside this sweep on the other
  // stream and need wave slots (and LDS) of their own; the sweep is VALU-bound long before its last two waves per SIMD
  const u32 per_simd = static_cast<u32>(sweep_waves_per_simd<R>());
  const u32 waves = std::min<u32>(bundles, 256u * 4u * (per_simd > 4 ? per_simd - 2 : per_simd));
  RVN_HIP(hipMemsetAsync(d_next, 0, 4, s));
  RVN_KLAUNCH_ON(kKNwForward, s, (nw_sweep_kernel<R, G><<<(waves + 3) / 4, 256, 0, s>>>(
                               d_jobs, d_idx, n_idx, T.packed.as<u64>(), Rd.packed.as<u64>(), d_hs, d_ck, d_result,
                               d_status, d_next)));
}

// largest threshold whose band fits a ring of G lanes with R blocks each
u32 kcap_of(u32 n, u32 m, u32 R, u32 G) {
  const u32 d = n > m ? n - m : m - n;
  // nw_ring_lanes(lo, hi, R) <= G  <=>  64R + lo + hi <= G (64R + 1);  lo + hi = 2 floor((k - d) / 2) + d
  const u64 room = static_cast<u64>(G) * (64ULL * R + 1) - 64ULL * R;
  if (room < d) return 0;
  const u64 k = d + ((room - d) / 2) * 2 + 1;  // (k - d) / 2 floors: an odd surplus costs nothing
  return static_cast<u32>(std::min<u64>(k, static_cast<u64>(n) + m));
}

u32 level_of(const NwJob& J) {
  for (u32 x = 0; x < kLevels; ++x)
    if (kRs[x] == J.R && kGs[x] == J.G) return x;
  return kLevels - 1;
}

}  // namespace

// Fills the band fields of `jobs` (k, kcap, R, G, hs, ckpt) and produces every job's window records in d_recs
// (records of a job start at its bp_off; jobs that cannot be aligned keep all-invalid records and are counted).
//
// Schedule: the jobs of a pass go in chunks whose hs + ck fit half the budget, longest jobs first.  The sweeps run on the
// engine's stream, the walk of chunk i on a second stream beside the sweeps of chunk i + 1 (two buffer sets): a walk is
// one lane per alignment and latency-bound (~1 us per column), a sweep fills the VALUs — together they cost the time of
// the sweeps plus the walk of the last, shortest jobs.
void nw_breakpoints(Engine& e, const ReadsDev& T, const ReadsDev& Rd, std::vector<NwJob>& jobs, u32 w,
                    NwWindowRec* d_recs, u64 n_recs, NwStats& st) {
  st = NwStats();
  const u32 nj = static_cast<u32>(jobs.size());
  hipStream_t s = e.stream;
  RVN_HIP(hipMemsetAsync(d_recs, 0xFF, n_recs * sizeof(NwWindowRec), s));
  if (nj == 0) return;
  RVN_HIP(hipEventRecord(e.ev0, s));
  if (!e.nw_streams[0]) {
    // HIP multiplexes its streams over a handful of hardware queues (four by default): two walk streams that land on the
    // same queue run one after the other, and the ~100-ms walk of the longest alignments held the walk queued behind it —
    // and with it the sweep waiting for that walk's buffer set (profiles/r05_nw_timeline.csv: a walk starting the moment
    // the long one ended).  Streams of another PRIORITY get hardware queues of their own: the long pole's stream (set 3)
    // is created at the highest priority, which also suits a kernel of 38 latency-bound waves.
    int prio_least = 0, prio_greatest = 0;
    RVN_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    // created into locals and handed to the engine only when all exist: a creation that throws half-way must not leave
    // nw_streams[0] set with the others null — the next call would skip this block and launch walks on the null stream
    constexpr int kNwEv = static_cast<int>(sizeof(e.nw_ev) / sizeof(e.nw_ev[0]));
    hipStream_t made[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [5..7]: nw_side
    hipEvent_t made_ev[kNwEv + 4] = {};
    try {
      for (int b = 0; b < 8; ++b) {
        // ([4], the upload stream, must not share a hardware queue with the engine's stream either: its copies run while a
        // pass queued before is sweeping there)
        if (b >= 3 && prio_greatest != prio_least)
          RVN_HIP(hipStreamCreateWithPriority(&made[b], hipStreamNonBlocking, prio_greatest));
        else
          RVN_HIP(hipStreamCreateWithFlags(&made[b], hipStreamNonBlocking));
      }
      for (hipEvent_t& ev : made_ev) RVN_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    } catch (...) {
      for (hipStream_t st2 : made)
        if (st2) (void)hipStreamDestroy(st2);
      for (hipEvent_t ev : made_ev)
        if (ev) (void)hipEventDestroy(ev);
      throw;
    }
    for (int b = 0; b < 3; ++b) e.nw_side[b] = made[5 + b];
    for (int i = 0; i < 4; ++i) e.nw_side_ev[i] = made_ev[kNwEv + i];
    for (int i = 0; i < kNwEv; ++i) e.nw_ev[i] = made_ev[i];
    for (int b = 0; b < 5; ++b) e.nw_streams[b] = made[b];  // (last: what the test above looks at)
  }
  // an error in the middle of a pass (a walk that left its band, an allocation that failed) must not leave walks running
  // on the side streams against buffers the next call hands out again
  struct WalkGuard {
    Engine& e;
    ~WalkGuard() {
      if (!std::uncaught_exceptions()) return;
      for (hipStream_t st2 : e.nw_streams)
        if (st2) (void)hipStreamSynchronize(st2);
      for (hipStream_t st2 : e.nw_side)
        if (st2) (void)hipStreamSynchronize(st2);
      (void)hipStreamSynchronize(e.stream);
    }
  } walk_guard{e};
  double rate = e.nw_rate > 0 ? e.nw_rate : 0.13;  // first call: ONT-like; too small only costs a repeat
  if (const char* ev = knob("RVN_NW_RATE")) rate = std::atof(ev);  // (debug builds: force repeats)

  // plan: the narrowest variant whose ring holds the band of k
  auto plan = [&](NwJob& J, u64 k) -> bool {
    const u32 d = J.n > J.m ? J.n - J.m : J.m - J.n;
    k = std::max<u64>(std::max<u64>(k, d), 16);
    k = std::min<u64>(k, static_cast<u64>(J.n) + J.m);  // D(n, m) <= n + m: that threshold never fails
    for (u32 lvl = 0; lvl < kLevels; ++lvl) {
      const u32 cap = kcap_of(J.n, J.m, kRs[lvl], kGs[lvl]);
      if (cap >= k) {
        J.R = kRs[lvl];
        J.G = kGs[lvl];
        J.k = static_cast<u32>(k);
        J.kcap = cap;
        return true;
      }
    }
    return false;
  };
  std::vector<u32> valid;
  for (u32 i = 0; i < nj; ++i) {
    const NwJob& J = jobs[i];
    if (J.n == 0 || J.m == 0 || J.n >= (1u << 30) || J.m >= (1u << 30)) ++st.n_unaligned;
    else valid.push_back(i);
  }

  u64 budget = 0;
  {
    size_t free_b = 0, total_b = 0;
    RVN_HIP(hipMemGetInfo(&free_b, &total_b));
    const u64 held = e.nw_hs.cap + e.nw_ck.cap + e.nw_hs2.cap + e.nw_ck2.cap + e.nw_hs3.cap + e.nw_ck3.cap + e.nw_hs4.cap + e.nw_ck4.cap;
    budget = std::min<u64>((static_cast<u64>(free_b) + devpool::free_total()) / 4 + held, 64ULL << 30);  // parked blocks count as free
    if (e.opt.nw_budget_mb > 0) budget = static_cast<u64>(e.opt.nw_budget_mb) << 20;
    budget = std::max<u64>(budget, 64ULL << 20);
  }
  const bool trace_lds = !(knob("RVN_NW_TRACE_MEM") && std::atoi(knob("RVN_NW_TRACE_MEM")) == 1);
  // which walk a launch takes (engine option nw_group_walk): 1 the lane per alignment with whole strips, 2 the group of lanes
  // per alignment, 3 the lane per alignment with strips of sixteen kept columns, otherwise by the number of alignments in the launch
  const int group_walk = static_cast<int>(e.opt.nw_group_walk);
  const bool one_stream = knob("RVN_NW_ONE_STREAM") != nullptr;
  const bool dbg_sync = knob("RVN_NW_DEBUG") && std::atoi(knob("RVN_NW_DEBUG")) >= 2;
  std::vector<double> rates;
  rates.reserve(nj);  // (growing it inside collect() cost milliseconds of page faults with the GPU idle)
  std::vector<u32> h_result(nj), h_status(nj);
  constexpr u32 kHeadMax = 4096;  // jobs of the pass of the longest alignments (below)
  NwJob* d_jobs = e.nw_jobs.get<NwJob>(static_cast<size_t>(nj) + 1 + 2 * kHeadMax);
  u32* d_res = e.nw_res.get<u32>(3 * static_cast<size_t>(nj) + 32 + 6 * kHeadMax);
  u32* d_status = d_res + nj + 1;
  u32* d_idx = d_status + nj + 1;
  u32* d_next = d_idx + nj + 1;
  // A pass = a set of planned jobs queued together: its job records, order, results and states on the device.  The usual
  // pass addresses them by job id; a COMPACT pass (the head, below) has arrays of its own, addressed by position in its order.
  struct PassDev {
    NwJob* jobs;
    u32 *idx, *res, *status;
    bool compact;
  };
  const PassDev dev_all{d_jobs, d_idx, d_res, d_status, false};
  const PassDev dev_head{d_jobs + nj + 1, d_next + 4 + 2 * kHeadMax, d_next + 4, d_next + 4 + kHeadMax, true};
  const PassDev dev_retry{dev_head.jobs + kHeadMax, dev_head.idx + 3 * kHeadMax, dev_head.res + 3 * kHeadMax,
                          dev_head.status + 3 * kHeadMax, true};
  u32* d_side_next = d_next + 4 + 6 * kHeadMax;  // (behind the compact passes' arrays) job counters of the sweeps launched beside the main stream, one per variant
  bool side_pending[3] = {false, false, false};   // sweeps on a side stream that the main stream has not waited for yet
  auto join_side = [&]() {
    for (int x = 0; x < 3; ++x) {
      if (side_pending[x]) RVN_HIP(hipStreamWaitEvent(s, e.nw_side_ev[x], 0));
      side_pending[x] = false;
    }
  };
  struct Obs {
    double len, d;
  };
  using clk = std::chrono::steady_clock;
  double h_plan = 0, h_order = 0, h_up = 0, h_res = 0;  // host milliseconds between the launches (RVN_NW_DEBUG)
  auto since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
  std::vector<Obs> obs;
  struct Chunk {
    size_t c0, c1;
    u64 hs_w, ck_e;
    u32 coff[kLevels + 1];
  };
  DevBuf* hs_buf[4] = {&e.nw_hs, &e.nw_hs2, &e.nw_hs3, &e.nw_hs4};
  DevBuf* ck_buf[4] = {&e.nw_ck, &e.nw_ck2, &e.nw_ck3, &e.nw_ck4};
  bool set_used[4] = {false, false, false, false};  // a walk queued since the last collect() holds the buffer set
  struct Queued {  // what collect() reads back
    std::vector<u32> order;
    PassDev dev;
    bool sweep_only;
    std::vector<u32> res, status;  // (compact passes)
  };
  std::vector<Queued> queued;
  std::vector<u8> early;  // job was handed to retry_early(): its state in the pass that found it stays "above the threshold"
  // Queues the sweeps and walks of `todo` (already planned) and returns; collect() waits for everything queued and reads
  // the results.  Chunk layout: kOwnFirst — the first chunk (the longest alignments) on buffer set 3 with a twelfth of the
  // budget, the others on sets 0-2 in rotation with a third each; kHeadOnly — that first chunk ALONE, what does not fit it
  // is handed back in `left`; kRotating — sets 0-2 only (the jobs behind a head that is already running).  up: the stream
  // of the uploads (the host waits for them: not the engine's stream when a pass queued before is sweeping on it).
  enum Layout { kOwnFirst, kHeadOnly, kRotating };
  auto enqueue = [&](const std::vector<u32>& todo, bool sweep_only, Layout layout, const PassDev& dev, hipStream_t up,
                     std::vector<u32>* left) {
    if (todo.empty()) return;
    // longest jobs first: they go through the widest variants, and the pass ends with the walks of the shortest ones
    auto t_h = clk::now();
    std::vector<u32> order;
    {  // descending by (variant, columns), ties in job order: two counting passes (12 + 12 key bits) instead of a sort
      const size_t nt = todo.size();
      std::vector<u32> key(nt), tmp(nt), idx(nt);
      parallel_for(nt, 16384, [&](size_t x0, size_t x1) {  // (the GPU waits for this planning: a few host threads)
        for (size_t x = x0; x < x1; ++x) {
          const NwJob& J = jobs[todo[x]];
          const u32 mm = std::min<u32>(J.m >> 3, (1u << 20) - 1);  // 8-base resolution is plenty for the ordering
          static_assert(kLevels <= 16, "4 bits of variant + 20 bits of length = the 24 key bits of the two passes");
          key[x] = 0xFFFFFFu - ((level_of(J) << 20) | mm);
          idx[x] = static_cast<u32>(x);
        }
      });
      for (int pass = 0; pass < 2; ++pass) {
        u32 cnt[4097] = {};
        const int sh = 12 * pass;
        for (size_t x = 0; x < nt; ++x) cnt[((key[idx[x]] >> sh) & 4095u) + 1]++;
        for (int c = 0; c < 4096; ++c) cnt[c + 1] += cnt[c];
        for (size_t x = 0; x < nt; ++x) tmp[cnt[(key[idx[x]] >> sh) & 4095u]++] = idx[x];
        idx.swap(tmp);
      }
      order.resize(nt);
      for (size_t x = 0; x < nt; ++x) order[x] = todo[idx[x]];
    }
    // chunks of the order whose hs + ck fit a third of the budget (a job larger than that goes alone).  (Capping the jobs
    // per chunk — an even eighth, or a third of what is left, so that the uncovered walk of the last chunk gets shorter —
    // was measured at C4 and did not pay: every extra sweep launch brings its own ramp and tail, +19 ms of sweep time
    // against ~20 ms less at the end; profiles/r05_nw_timeline.csv.)
    std::vector<Chunk> chunks;
    struct Need {
      u64 hw, ce, cells;
      u32 level;
    };
    std::vector<Need> need(order.size());  // what a job stores and computes: per job in parallel, summed up in order below
    parallel_for(order.size(), 16384, [&](size_t x0, size_t x1) {
      for (size_t x = x0; x < x1; ++x) {
        const NwJob& J = jobs[order[x]];
        const NwGeo g = nw_geo(J.n, J.m, J.k, J.R);
        need[x] = Need{g.hs_words(), g.ck_entries(), static_cast<u64>(J.m) * (static_cast<u64>(g.lo) + g.hi + 1), level_of(J)};
      }
    });
    for (size_t c0 = 0; c0 < order.size();) {
      Chunk C{};
      C.c0 = c0;
      size_t c1 = c0;
      while (c1 < order.size()) {
        NwJob& J = jobs[order[c1]];
        const u64 hw = need[c1].hw, ce = need[c1].ce;
        // three buffer sets in rotation + one of its own for the first chunk (the longest alignments: its walk is
        // latency-bound — 106 ms for 2 432 alignments at C4 — and a set shared with chunk 3 made that chunk's sweep wait
        // 17 ms for it), a quarter of a share
        const u64 share = (chunks.empty() && layout != kRotating) ? budget / 12 : budget / 3;
        if (c1 > c0 && (C.hs_w + hw) * 4 + (C.ck_e + ce) * 16 > share) break;
        J.hs = C.hs_w;
        J.ckpt = C.ck_e;
        C.hs_w += hw;
        C.ck_e += ce;
        st.band_cells += need[c1].cells;
        C.coff[need[c1].level + 1]++;
        ++c1;
      }
      // the order is by descending level: offsets of the classes inside the chunk, in that order
      u32 run_off = 0, cnt[kLevels];
      for (u32 x = 0; x < kLevels; ++x) cnt[x] = C.coff[x + 1];
      for (int x = static_cast<int>(kLevels) - 1; x >= 0; --x) {
        C.coff[x] = run_off;
        run_off += cnt[x];
      }
      C.coff[kLevels] = run_off;  // count of class x = offset of class x - 1 (or the chunk's end) - its own offset
      C.c1 = c1;
      st.store_bytes = std::max<u64>(st.store_bytes, C.hs_w * 4 + C.ck_e * 16);
      chunks.push_back(C);
      c0 = c1;
      if (layout == kHeadOnly) break;
    }
    if (layout == kHeadOnly && chunks[0].c1 < order.size()) {  // what the head's share does not hold goes with the rest
      for (size_t x = chunks[0].c1; x < order.size(); ++x) left->push_back(order[x]);
      order.resize(chunks[0].c1);
    }
    // chunk 0 -> set 3 (its own), chunk ci >= 1 -> set (ci - 1) % 3;  kRotating: chunk ci -> set ci % 3
    auto set_of = [&](size_t ci) -> int {
      if (layout == kRotating) return static_cast<int>(ci % 3);
      return ci == 0 ? 3 : static_cast<int>((ci - 1) % 3);
    };
    for (int b = 0; b < 4; ++b) {
      // sized for the chunks the set serves (a lone job beyond its share enlarges one set, not all)
      u64 set_hs = 0, set_ck = 0;
      bool used = false;
      for (size_t ci = 0; ci < chunks.size(); ++ci) {
        if (set_of(ci) != b) continue;
        used = true;
        set_hs = std::max(set_hs, chunks[ci].hs_w);
        set_ck = std::max(set_ck, chunks[ci].ck_e);
      }
      if (!used) continue;
      (void)hs_buf[b]->get<u32>(set_hs + 4 * 64 * 8 + 16);  // + slack: the walk reads up to two words past a job's last one
      (void)ck_buf[b]->get<NwPm>(set_ck + 16);
    }
    u64* d_strip = nullptr;
    if (!trace_lds) {
      size_t mc = 0;
      for (const Chunk& C : chunks) mc = std::max(mc, C.c1 - C.c0);
      d_strip = e.nw_strip.get<u64>(static_cast<size_t>((mc + 63) / 64) * 2 * kNwStripCols * 64 * 4 + 64);
    }
    h_order += since(t_h);
    t_h = clk::now();
    if (dev.compact) {  // records in the order of the pass, addressed by position
      std::vector<NwJob> hj(order.size());
      std::vector<u32> iota(order.size());
      for (size_t x = 0; x < order.size(); ++x) {
        hj[x] = jobs[order[x]];
        iota[x] = static_cast<u32>(x);
      }
      RVN_HIP(hipMemcpyAsync(dev.jobs, hj.data(), hj.size() * sizeof(NwJob), hipMemcpyHostToDevice, up));
      RVN_HIP(hipMemcpyAsync(dev.idx, iota.data(), iota.size() * 4, hipMemcpyHostToDevice, up));
      RVN_HIP(rvn_stream_sync(up));  // (locals)
    } else {
      RVN_HIP(hipMemcpyAsync(dev.jobs, jobs.data(), static_cast<size_t>(nj) * sizeof(NwJob), hipMemcpyHostToDevice, up));
      RVN_HIP(hipMemcpyAsync(dev.idx, order.data(), order.size() * 4, hipMemcpyHostToDevice, up));
      RVN_HIP(rvn_stream_sync(up));  // `jobs` / `order` are pageable: the copies must be done before the host goes on
    }
    h_up += since(t_h);
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
      const Chunk& C = chunks[ci];
      const int b = set_of(ci);
      u32* hs = hs_buf[b]->as<u32>();
      NwPm* ck = ck_buf[b]->as<NwPm>();
      const u32* idx_c = dev.idx + C.c0;
      const u32 cn = static_cast<u32>(C.c1 - C.c0);
      // the walk that used this buffer set before (three chunks back in the rotation) is done with it
      if (set_used[b]) RVN_HIP(hipStreamWaitEvent(s, e.nw_ev[b], 0));
      auto count_of = [&](u32 x) -> u32 {  // classes are laid out from the widest variant down
        const u32 next_off = x == 0 ? cn : C.coff[x - 1];
        return next_off - C.coff[x];
      };
      // one walk stream per buffer set: the walk of the longest alignments (chunk 0: few waves, tens of milliseconds of
      // latency) must not hold back the walks of the chunks behind it
      hipStream_t ts = one_stream ? s : e.nw_streams[b];
      auto launch_walk = [&](hipStream_t wst, const u32* idx_w, u32 n_w) {
        if (group_walk == 2 || (group_walk != 1 && n_w <= kNwGroupWalkMaxJobs)) {
          // few alignments: a group of lanes each (nwtrace.h) — the walk of a few thousand alignments costs its longest one's
          // latency, and a group walks a column in a fraction of a lane's time
          constexpr u32 NG = 64 / kNwGroupLanes;
          RVN_KLAUNCH_ON(kKNwTraceback, wst, (nw_trace_group_kernel<kNwGroupLanes><<<(n_w + NG - 1) / NG, 64, 0, wst>>>(
                                                 dev.jobs, idx_w, n_w, T.packed.as<u64>(), Rd.packed.as<u64>(), hs, ck, dev.res,
                                                 dev.status, w, d_recs)));
        } else if (trace_lds && (group_walk == 3 || (group_walk != 1 && n_w > 64u * 4u * 256u))) {
          // more waves than four per CU hold: strips of sixteen kept columns, nine waves per CU (C4: the walk of the last
          // chunk's 140 000 shortest alignments, alone on the GPU, 18 -> 13 ms; align_ms 185 -> 174)
          RVN_KLAUNCH_ON(kKNwTraceback, wst, (nw_trace_kernel<true, kNwLaneStripCols><<<(n_w + 63) / 64, 64, 0, wst>>>(
                                                 dev.jobs, idx_w, n_w, T.packed.as<u64>(), Rd.packed.as<u64>(), hs, ck, dev.res,
                                                 dev.status, w, d_recs, nullptr)));
        } else if (trace_lds) {
          RVN_KLAUNCH_ON(kKNwTraceback, wst, (nw_trace_kernel<true, kNwCkSteps><<<(n_w + 63) / 64, 64, 0, wst>>>(
                                                 dev.jobs, idx_w, n_w, T.packed.as<u64>(), Rd.packed.as<u64>(), hs, ck, dev.res,
                                                 dev.status, w, d_recs, nullptr)));
        } else {
          RVN_KLAUNCH_ON(kKNwTraceback, wst, (nw_trace_kernel<false, kNwCkSteps><<<(n_w + 63) / 64, 64, 0, wst>>>(
                                                 dev.jobs, idx_w, n_w, T.packed.as<u64>(), Rd.packed.as<u64>(), hs, ck, dev.res,
                                                 dev.status, w, d_recs,
                                                 d_strip + static_cast<size_t>(b) * ((e.nw_strip.cap / 32) & ~size_t(63)))));
        }
      };
      // A variant's launch of few waves — the pilot's five, the several-blocks-per-lane variants of a round's longest
      // alignments (a few hundred jobs whose sweep is tens of thousands of dependent steps: 14-18 ms on 60-300 of the
      // machine's 6 000 wave slots) — runs BESIDE the others on a side stream instead of in front of them; the walk (and
      // whoever reads the states) waits for all of them, the main stream goes on with the next chunk.
      bool side[kLevels] = {};
      bool side_used[3] = {false, false, false};
      {
        u32 n_levels = 0;
        for (u32 x = 0; x < kLevels; ++x) n_levels += count_of(x) ? 1u : 0u;
        if (n_levels >= 2 && !one_stream && !dbg_sync)
          for (u32 x = 0; x < kLevels; ++x) {
            const u32 bundles = (count_of(x) + 64 / kGs[x] - 1) / (64 / kGs[x]);
            side[x] = count_of(x) && bundles <= kSideSweepMaxWaves;
            if (side[x]) side_used[x % 3] = true;
          }
        if (side_used[0] || side_used[1] || side_used[2]) {
          RVN_HIP(hipEventRecord(e.nw_side_ev[3], s));
          for (int y = 0; y < 3; ++y)
            if (side_used[y]) RVN_HIP(hipStreamWaitEvent(e.nw_side[y], e.nw_side_ev[3], 0));
        }
      }
#define RVN_SWEEP(x, R_, G_)                                                                                         \
  do {                                                                                                               \
    launch_sweep<R_, G_>(e, side[x] ? e.nw_side[(x) % 3] : s, dev.jobs, idx_c + C.coff[x], count_of(x), T, Rd, hs, ck, \
                         dev.res, dev.status, side[x] ? d_side_next + (x) : d_next);                                   \
    if (dbg_sync && count_of(x)) {                                                                                   \
      RVN_HIP(hipStreamSynchronize(s));                                                                              \
      std::fprintf(stderr, "[raven_hip] nw: sweep R=%d G=%d done, %u jobs\n", R_, G_, count_of(x));                   \
    }                                                                                                                \
  } while (0)
      RVN_SWEEP(7, 8, 64);
      RVN_SWEEP(6, 4, 64);
      RVN_SWEEP(5, 2, 64);
      RVN_SWEEP(4, 1, 64);
      RVN_SWEEP(3, 1, 32);
      RVN_SWEEP(2, 1, 16);
      RVN_SWEEP(1, 1, 8);
      RVN_SWEEP(0, 1, 4);
#undef RVN_SWEEP
      for (int y = 0; y < 3; ++y)
        if (side_used[y]) {
          RVN_HIP(hipEventRecord(e.nw_side_ev[y], e.nw_side[y]));
          side_pending[y] = true;
        }
      ++st.n_batches;
      if (sweep_only) continue;
      if (!one_stream) {
        RVN_HIP(hipEventRecord(e.nw_ev[4], s));
        RVN_HIP(hipStreamWaitEvent(ts, e.nw_ev[4], 0));
        for (int y = 0; y < 3; ++y)
          if (side_used[y]) RVN_HIP(hipStreamWaitEvent(ts, e.nw_side_ev[y], 0));
      }
      launch_walk(ts, idx_c, cn);
      if (!one_stream) {
        RVN_HIP(hipEventRecord(e.nw_ev[b], ts));
        set_used[b] = true;
      }
      if (dbg_sync) {
        RVN_HIP(hipStreamSynchronize(ts));
        std::fprintf(stderr, "[raven_hip] nw: trace done, %u jobs\n", cn);
      }
    }
    queued.push_back(Queued{std::move(order), dev, sweep_only, {}, {}});
  };
  // Waits for everything queued, reads the results; the jobs beyond their thresholds come back in `again` (planned with
  // twice the band and the variant that holds it).
  auto collect = [&](std::vector<u32>& again) {
    again.clear();
    join_side();
    for (int b = 0; b < 4; ++b) {  // (an event stands for the LAST walk recorded on its set)
      if (set_used[b]) RVN_HIP(hipStreamWaitEvent(s, e.nw_ev[b], 0));
      set_used[b] = false;
    }
    for (Queued& q : queued) {
      if (q.dev.compact) {
        q.res.resize(q.order.size());
        q.status.resize(q.order.size());
        RVN_HIP(hipMemcpyAsync(q.res.data(), q.dev.res, q.order.size() * 4, hipMemcpyDeviceToHost, s));
        RVN_HIP(hipMemcpyAsync(q.status.data(), q.dev.status, q.order.size() * 4, hipMemcpyDeviceToHost, s));
      }
    }
    RVN_HIP(hipMemcpyAsync(h_result.data(), d_res, static_cast<size_t>(nj) * 4, hipMemcpyDeviceToHost, s));
    RVN_HIP(hipMemcpyAsync(h_status.data(), d_status, static_cast<size_t>(nj) * 4, hipMemcpyDeviceToHost, s));
    RVN_HIP(rvn_stream_sync(s));
    auto t_h = clk::now();
    for (const Queued& q : queued)
      if (q.dev.compact)
        for (size_t x = 0; x < q.order.size(); ++x) {
          h_result[q.order[x]] = q.res[x];
          h_status[q.order[x]] = q.status[x];
        }
    for (size_t qi = 0; qi < queued.size(); ++qi) {
      const Queued& q = queued[qi];
      const bool is_early_retry = q.dev.jobs == dev_retry.jobs;
      // the aligned jobs of a pass (all but a handful) are only counted: on a few threads — the pass order is by length,
      // i.e. random in the 28 MB of job records, and one thread's cache misses were ~10 ms with the GPU idle
      std::vector<u32> rest_of;  // everything that is not a plain "aligned"
      if (q.sweep_only) {
        rest_of = q.order;
      } else {
        std::mutex mu_;
        parallel_for(q.order.size(), 16384, [&](size_t x0, size_t x1) {
          u64 n_al = 0, sum_d = 0;
          std::vector<double> rr;
          std::vector<u32> other;
          rr.reserve(x1 - x0);
          for (size_t x = x0; x < x1; ++x) {
            const u32 i = q.order[x];
            if (!is_early_retry && !early.empty() && early[i]) continue;  // its result is the early retry pass's
            if (h_status[i] != 0) {
              other.push_back(i);
              continue;
            }
            const NwJob& J = jobs[i];
            ++n_al;
            sum_d += h_result[i];
            rr.push_back(static_cast<double>(h_result[i]) / std::max(J.n, J.m));
          }
          std::lock_guard<std::mutex> lk(mu_);
          st.n_aligned += n_al;
          st.sum_distance += sum_d;
          rates.insert(rates.end(), rr.begin(), rr.end());
          rest_of.insert(rest_of.end(), other.begin(), other.end());
        });
        std::sort(rest_of.begin(), rest_of.end());  // (the threads finish in any order)
      }
      for (u32 i : rest_of) {
        NwJob& J = jobs[i];
        if (h_status[i] == 2) {  // distance above the threshold: twice the band (and the variant that holds it)
          ++st.n_retries;
          if (J.k >= static_cast<u64>(J.n) + J.m || !plan(J, static_cast<u64>(J.k) * 2)) ++st.n_unaligned;
          else again.push_back(i);
        } else if (h_status[i] != 0) {
          throw HipError("[raven_hip] alignment path: the walk left the stored band (internal error)");
        } else {  // (a sweep-only pass: the pilot)
          obs.push_back(Obs{static_cast<double>(std::max(J.n, J.m)), static_cast<double>(h_result[i])});
        }
      }
    }
    early.clear();
    queued.clear();
    h_res += since(t_h);
  };
  // Whether an alignment is beyond its threshold is known when its SWEEP is done: the states are read once the sweeps of
  // everything queued are through — the last walks still run on their streams — and the few jobs above their thresholds
  // (two or three of 295 000 at C4) are planned again and queued at once, as a compact pass on the head's buffer set,
  // instead of as a pass of their own behind the last walk (~9 ms per round of sweep + lonely walk + host round trips,
  // profiles/r05_nw_timeline.csv).  A job handed on here is skipped by collect() in the pass that found it.
  auto retry_early = [&]() {
    if (one_stream || queued.empty()) return;
    join_side();
    for (Queued& q : queued)
      if (q.dev.compact) {
        q.status.resize(q.order.size());
        RVN_HIP(hipMemcpyAsync(q.status.data(), q.dev.status, q.order.size() * 4, hipMemcpyDeviceToHost, s));
      }
    RVN_HIP(hipMemcpyAsync(h_status.data(), d_status, static_cast<size_t>(nj) * 4, hipMemcpyDeviceToHost, s));
    RVN_HIP(rvn_stream_sync(s));  // (the sweeps; not the walks)
    std::vector<u32> todo;
    for (const Queued& q : queued)
      for (size_t x = 0; x < q.order.size(); ++x) {
        const u32 i = q.order[x];
        if ((q.dev.compact ? q.status[x] : h_status[i]) != 2) continue;
        if (todo.size() >= kHeadMax) break;  // (a wrong rate estimate: the ordinary repeat pass takes them)
        NwJob& J = jobs[i];
        if (J.k >= static_cast<u64>(J.n) + J.m) continue;  // (collect() counts it as unaligned)
        const NwJob before = J;
        if (!plan(J, static_cast<u64>(J.k) * 2)) {
          J = before;
          continue;
        }
        todo.push_back(i);
      }
    if (todo.empty()) return;
    early.assign(nj, 0);
    for (u32 i : todo) early[i] = 1;
    st.n_retries += todo.size();
    std::vector<u32> left;
    // (set 3 may have to grow for a doubled band, and growing hands the old block back: the walk on it — the head's, queued
    // long before — must be through)
    if (set_used[3]) RVN_HIP(hipEventSynchronize(e.nw_ev[3]));
    enqueue(todo, false, kHeadOnly, dev_retry, e.nw_streams[4], &left);
    for (u32 i : left) early[i] = 0;  // (beyond the set's share: their doubled plan stands, collect() doubles it once more)
    st.n_retries -= left.size();
  };
  // aligns every job of `todo` (already planned), repeating the ones beyond their threshold with twice the band
  auto run = [&](std::vector<u32> todo, bool sweep_only) {
    while (!todo.empty()) {
      enqueue(todo, sweep_only, kOwnFirst, dev_all, s, nullptr);
      std::vector<u32> again;
      collect(again);
      todo.swap(again);
    }
  };

  // Thresholds.  A pilot sample (every call: the error level changes from round to round) is aligned with a generous
  // band; its distances give the mean rate mu and the spread around mu x len as  var = a len + b len^2  (a: base-level
  // noise, b: differences between reads).  Everyone gets  k = mu len + 4.5 sqrt(a len + b len^2) + 6 : with a normal
  // spread a few alignments in a million are repeated (a repeat pass ends with lonely, latency-bound walks, so it is
  // worth ~3 % more band to make it rare), against one in ten with a 90th-percentile rule.  The pilot is swept for its
  // distances only and skips the longest quarter of the reads.  Small batches keep the previous call's estimate.
  std::vector<u32> rest, head;
  double mu = -1, va = 0, vb = 0;
  if (valid.size() >= 4096 && !knob("RVN_NW_RATE")) {
    std::vector<u32> lens;
    for (u32 i : valid) lens.push_back(std::max(jobs[i].n, jobs[i].m));
    std::nth_element(lens.begin(), lens.begin() + lens.size() * 3 / 4, lens.end());
    const u32 len_cap = lens[lens.size() * 3 / 4];
    std::vector<u32> pilot;
    const size_t stride = valid.size() / 1400;
    for (size_t x = 0; x < valid.size(); ++x) {
      const NwJob& J = jobs[valid[x]];
      if (x % stride == stride / 2 && pilot.size() < 1024 && std::max(J.n, J.m) <= len_cap) pilot.push_back(valid[x]);
      rest.push_back(valid[x]);  // the pilot is swept for its distances only; its jobs are aligned with everyone else
    }
    std::vector<u32> ok;
    for (u32 i : pilot) {
      NwJob& J = jobs[i];
      if (plan(J, static_cast<u64>((rate * 1.25 + 0.02) * std::max(J.n, J.m)) + 16)) ok.push_back(i);
    }
    enqueue(ok, true, kOwnFirst, dev_all, s, nullptr);
    // (while the pilot is swept) the longest alignments: they go first and by themselves, see below
    if (!one_stream) {
      const size_t n_head = std::min<size_t>(kHeadMax, rest.size() / 8);
      auto longer = [&](u32 x, u32 y) {
        const u32 lx = std::max(jobs[x].n, jobs[x].m), ly = std::max(jobs[y].n, jobs[y].m);
        return lx != ly ? lx > ly : x < y;
      };
      std::nth_element(rest.begin(), rest.begin() + n_head, rest.end(), longer);
      head.assign(rest.begin(), rest.begin() + n_head);
      rest.erase(rest.begin(), rest.begin() + n_head);
    }
    {
      std::vector<u32> again;
      collect(again);
      run(again, true);
    }
    if (obs.size() >= 256) {
      double sl = 0, sd = 0;
      for (const Obs& o : obs) {
        sl += o.len;
        sd += o.d;
      }
      mu = sd / sl;
      // least squares of the squared residuals on (len, len^2), the 2 % largest standardised residuals left out
      std::vector<double> zs;
      for (const Obs& o : obs) zs.push_back(std::fabs(o.d - mu * o.len) / std::sqrt(o.len));
      std::vector<double> zsorted = zs;
      std::sort(zsorted.begin(), zsorted.end());
      const double zcut = zsorted[static_cast<size_t>(zsorted.size() * 0.98)];
      double s11 = 0, s12 = 0, s22 = 0, t1 = 0, t2 = 0, sr = 0, sn = 0;
      for (size_t x = 0; x < obs.size(); ++x) {
        if (zs[x] > zcut) continue;
        const double L = obs[x].len * 1e-3, r2 = (obs[x].d - mu * obs[x].len) * (obs[x].d - mu * obs[x].len);
        s11 += L * L;
        s12 += L * L * L;
        s22 += L * L * L * L;
        t1 += r2 * L;
        t2 += r2 * L * L;
        sr += r2 / L;
        sn += 1;
      }
      const double det = s11 * s22 - s12 * s12;
      double a = -1, b = -1;
      if (det > 1e-9 * s11 * s22) {
        a = (t1 * s22 - t2 * s12) / det;
        b = (t2 * s11 - t1 * s12) / det;
      }
      if (!(a >= 0) || !(b >= 0)) {  // one length only, or a fit outside the model: all spread on the linear term
        a = sn > 0 ? sr / sn : mu * 1e3;
        b = 0;
      }
      va = std::max(a * 1e-3, 0.25 * mu);  // back to bases; never below a quarter of the Poisson level
      vb = b * 1e-6;
    }
  } else {
    rest = valid;
  }
  {
    auto t_h = clk::now();
    auto threshold = [&](const NwJob& J) -> u64 {
      const double len = std::max(J.n, J.m);
      const double z = 4.5;
      return mu > 0 ? static_cast<u64>(mu * len + z * std::sqrt(va * len + vb * len * len)) + 6 : static_cast<u64>(rate * len) + 16;
    };
    // The longest alignments are a pass of their own, queued the moment the thresholds exist: planning, ordering and
    // uploading 300 000 jobs took the host ~20 ms in which the GPU did nothing (profiles/r05_nw_timeline.csv: 6.6 -> 28.7 ms),
    // every round; now the head's sweeps (~23 ms at C4) run in that time, and its walk — the long pole, one lane per
    // alignment — starts that much earlier.  (Results do not depend on which pass or chunk a job is in.)
    std::vector<u32> ok_head, left;
    for (u32 i : head) {
      if (plan(jobs[i], threshold(jobs[i]))) ok_head.push_back(i);
      else ++st.n_unaligned;
    }
    const bool split = !ok_head.empty();
    if (split) enqueue(ok_head, false, kHeadOnly, dev_head, s, &left);
    std::vector<u8> planned(rest.size());
    parallel_for(rest.size(), 16384, [&](size_t x0, size_t x1) {
      for (size_t x = x0; x < x1; ++x) planned[x] = plan(jobs[rest[x]], threshold(jobs[rest[x]])) ? 1 : 0;
    });
    std::vector<u32> ok;
    ok.reserve(rest.size() + left.size());
    for (size_t x = 0; x < rest.size(); ++x) {
      if (planned[x]) ok.push_back(rest[x]);
      else ++st.n_unaligned;
    }
    ok.insert(ok.end(), left.begin(), left.end());
    h_plan += since(t_h);
    enqueue(ok, false, split ? kRotating : kOwnFirst, dev_all, split ? e.nw_streams[4] : s, nullptr);
    retry_early();
    std::vector<u32> again;
    collect(again);
    run(again, false);
  }
  RVN_HIP(hipEventRecord(e.ev1, s));
  if (rates.size() >= 32) {  // rate estimate for the next call (the pilot's first threshold / small batches)
    const size_t at = std::min(rates.size() - 1, static_cast<size_t>(rates.size() * 0.9));
    std::nth_element(rates.begin(), rates.begin() + at, rates.end());  // (the order statistic a full sort gave: ~15 ms of host time per round)
    e.nw_rate = rates[at] * 1.05 + 0.002;
  }
  RVN_HIP(hipEventSynchronize(e.ev1));
  float ms = 0;
  RVN_HIP(hipEventElapsedTime(&ms, e.ev0, e.ev1));
  st.ms = ms;
  if (knob("RVN_NW_DEBUG"))
    std::fprintf(stderr, "[raven_hip] nw host: plan %.1f ms, order + chunks %.1f ms, uploads %.1f ms, results %.1f ms\n", h_plan, h_order, h_up,
                 h_res);
  if (knob("RVN_NW_DEBUG"))
    std::fprintf(stderr, "[raven_hip] nw: %u jobs, %llu aligned, %llu retries, %llu chunks, %.3e band cells, %.1f MB hs + ck, %.1f ms; pilot mu %.4f a %.4f b %.3e\n", nj,
                 static_cast<unsigned long long>(st.n_aligned), static_cast<unsigned long long>(st.n_retries),
                 static_cast<unsigned long long>(st.n_batches), static_cast<double>(st.band_cells), st.store_bytes / 1048576.0, ms, mu, va, vb);
}

#ifdef RVN_TEST_HOOKS
// ---- CPU stepper of the same code (test hook rvn_test_nw_breakpoints): 64 emulated lanes, host arrays --------------
static u64 g_group_batches_store = 0;
static u64* const g_group_batches = &g_group_batches_store;  // batches of the hook's last group walk (band[3] when a group walk is asked for)
template <int R>
static int emulate_job(NwJob J, u32 G, const u64* t_words, const u64* r_words, u32 w, NwWindowRec* recs, u32* distance,
                       u32* band, int walk_gl) {
  std::vector<u64> peq(static_cast<size_t>(R) * 4 * 64);
  std::vector<NwSweepLane<R, 64>> lanes(64);
  std::vector<int> xp(64), sp(64);
  std::vector<u32> hs;
  std::vector<NwPm> ck;
  NwGeo g;
  u32 res = 0;
  for (;;) {
    g = nw_geo(J.n, J.m, J.k, R);
    if (static_cast<u32>(g.L) > G) return -3;  // beyond this variant's ring
    hs.assign(g.hs_words() + 4 * 64 * 8 + 16, 0xA5A5A5A5u);
    ck.assign(g.ck_entries() + 16, NwPm{0x1234567887654321ULL, 0x0FEDCBA99ABCDEF0ULL});
    for (int l = 0; l < 64; ++l) {
      if (l < static_cast<int>(G)) lanes[l].init(J, t_words, r_words, g, l, peq.data(), l);
      else lanes[l].init_idle(peq.data(), l);
    }
    const int n_steps_w = (g.n_steps + 15) & ~15;
    const int n_g = (g.n_steps + kNwHsSteps - 1) / kNwHsSteps, n_q = g.n_steps / kNwCkSteps;
    const u64 row = static_cast<u64>(g.L) * R;
    for (int t = 1; t <= n_steps_w; ++t) {
      for (int l = 0; l < 64; ++l) {  // the shuffles read the previous lane's values of the previous step
        const int src = l == 0 ? g.L - 1 : l - 1;
        xp[l] = lanes[src].xf;
        sp[l] = lanes[src].sc;
      }
      for (int l = 0; l < 64; ++l)
        if (lanes[l].has_event(t)) lanes[l].event(t, xp[l], sp[l]);
      for (int l = 0; l < 64; ++l) lanes[l].step(t, xp[l]);
      if ((t & 15) == 0) {
        const int gi = (t >> 4) - 1;
        for (int l = 0; l < g.L; ++l) {
          if (gi < n_g)
            for (int r = 0; r < R; ++r) hs[static_cast<u64>(gi) * row + static_cast<u64>(l) * R + r] = lanes[l].acc[r];
          if ((t & 31) == 0 && (t >> 5) - 1 < n_q)
            for (int r = 0; r < R; ++r)
              ck[static_cast<u64>((t >> 5) - 1) * row + static_cast<u64>(l) * R + r] = NwPm{lanes[l].Pv[r], lanes[l].Mv[r]};
        }
        for (int l = 0; l < 64; ++l) lanes[l].next_group(t);
      }
    }
    u32 res1 = 0;
    for (int l = 0; l < 64; ++l) res1 = std::max(res1, lanes[l].result);
    if (res1 == 0) return -5;
    res = res1 - 1u;
    if (res <= J.k) break;
    if (J.k >= J.n + J.m) return -5;
    J.k = static_cast<u32>(std::min<u64>(2ULL * J.k, static_cast<u64>(J.n) + J.m));
  }
  *distance = res;
  if (band) {
    band[0] = J.k;
    band[1] = static_cast<u32>(g.L);
    band[2] = R;
  }
  if (walk_gl == 16) return nw_trace_group_host<16>(J, g, t_words, r_words, hs.data(), ck.data(), res, w, recs, band ? g_group_batches : nullptr);
  if (walk_gl == 64) return nw_trace_group_host<64>(J, g, t_words, r_words, hs.data(), ck.data(), res, w, recs, band ? g_group_batches : nullptr);
  if (walk_gl == 4) return nw_trace_group_host<4>(J, g, t_words, r_words, hs.data(), ck.data(), res, w, recs, band ? g_group_batches : nullptr);
  if (walk_gl == 1) {  // the lane walk as nw_trace_kernel<true> runs it: strips of sixteen kept columns
    u64 pv16[kNwLaneStripCols + 1], mv16[kNwLaneStripCols + 1];
    const NwStripMem<1> mem16{pv16, mv16, 0};
    return nw_trace_job<1, kNwLaneStripCols>(J, g, t_words, r_words, hs.data(), ck.data(), mem16, res, w, recs);
  }
  if (walk_gl != 0) return -2;
  u64 pv[kNwStripCols], mv[kNwStripCols];
  const NwStripMem<1> mem{pv, mv, 0};
  return nw_trace_job<1>(J, g, t_words, r_words, hs.data(), ck.data(), mem, res, w, recs);
}

// force_R > 0: that many blocks per lane (whole-wave ring); force_R < 0: R = 1 with a ring of at most -force_R lanes
// (the lane groups of the narrow variants); force_R >= 1000: the variant (R, G) = (force_R / 1000, force_R % 1000);
// 0: the narrowest variant that holds k, the next one on overflow
int nw_breakpoints_host(const u64* t_words, u32 t_len, const u64* r_words, u32 r_len, u32 t_begin, u32 n, u32 q_begin,
                        u32 m, int rc, u32 w, u32 k, int force_R, NwWindowRec* recs, u32* distance, u32* band) {
  (void)t_len;
  const int walk_gl = (rc >> 8) & 0xFF;  // bits 8-15 of rc: lanes per alignment of the group walk (0: the lane walk)
  rc &= 1;
  g_group_batches_store = 0;
  if (n == 0 || m == 0) return -1;
  NwJob J{};
  J.t_begin = t_begin;
  J.n = n;
  J.q_begin = q_begin;
  J.m = m;
  J.r_len = r_len;
  J.rc = rc ? 1 : 0;
  J.n_windows = (t_begin + n - 1) / w - t_begin / w + 1;
  for (u32 x = 0; x < J.n_windows; ++x) {
    recs[x].first_t = recs[x].first_q = recs[x].last_t = recs[x].last_q = 0xFFFFFFFFu;
    for (int g = 0; g < 8; ++g) recs[x].grid[g] = 0xFFFFu;
  }
  const u32 d = n > m ? n - m : m - n;
  u64 kk = std::min<u64>(std::max<u64>(std::max<u64>(k, d), 1), static_cast<u64>(n) + m);
  u32 lvl = 0;
  if (force_R >= 1000) {  // R * 1000 + G: one particular variant
    while (lvl < kLevels && !(kRs[lvl] == static_cast<u32>(force_R / 1000) && kGs[lvl] == static_cast<u32>(force_R % 1000))) ++lvl;
    if (lvl == kLevels) return -2;
  } else if (force_R > 0) {  // that many blocks per lane, whole-wave ring
    while (lvl < kLevels && !(kRs[lvl] == static_cast<u32>(force_R) && kGs[lvl] == 64)) ++lvl;
    if (lvl == kLevels) return -2;
  } else if (force_R < 0) {  // one block per lane, ring of at most -force_R lanes
    while (lvl < kLevels && !(kRs[lvl] == 1 && kGs[lvl] == static_cast<u32>(-force_R))) ++lvl;
    if (lvl == kLevels) return -2;
  }
  for (; lvl < kLevels; ++lvl) {
    const u32 cap = kcap_of(n, m, kRs[lvl], kGs[lvl]);
    if (cap < kk) {
      if (force_R) return -2;
      continue;
    }
    J.R = kRs[lvl];
    J.G = kGs[lvl];
    J.k = static_cast<u32>(kk);
    J.kcap = cap;
    int rcode;
    switch (J.R) {
      case 1: rcode = emulate_job<1>(J, J.G, t_words, r_words, w, recs, distance, band, walk_gl); break;
      case 2: rcode = emulate_job<2>(J, J.G, t_words, r_words, w, recs, distance, band, walk_gl); break;
      case 4: rcode = emulate_job<4>(J, J.G, t_words, r_words, w, recs, distance, band, walk_gl); break;
      default: rcode = emulate_job<8>(J, J.G, t_words, r_words, w, recs, distance, band, walk_gl); break;
    }
    if (walk_gl && band) band[3] = static_cast<u32>(g_group_batches_store);
    if (rcode == -3 && !force_R) {  // the doubled threshold no longer fits this variant's ring
      kk = std::min<u64>(static_cast<u64>(cap) + 1, static_cast<u64>(n) + m);
      continue;
    }
    return rcode;
  }
  return -3;
}

#endif  // RVN_TEST_HOOKS

}  // namespace rvn
