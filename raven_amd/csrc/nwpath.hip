// nwpath.hip — kernel and host driver of the alignment-path stage of a polishing round (see nwpath.h): for every
// read's best overlap, the global alignment path against its target span and racon's per-window breakpoints
// (racon Overlap::find_breaking_points, reached from RavenLib/src/polish.cc:51).
//
//   nw_path_kernel<R>   persistent waves, one alignment per wave at a time (longest first): pass 1 (banded Myers sweep
//                       keeping a checkpoint every kNwSeg columns; threshold doubled in place until exact), then the
//                       walk back through the segments (re-sweep of a segment into the wave's scratch, walk, next).
//
// Host side: first thresholds k from the running error-rate estimate of the engine (first call: a generous default),
// blocks-per-lane R from k, jobs that outgrow their R (distance > kcap) are relaunched with the next R — the result is
// always the exact optimal path, the estimate only decides how much band is computed.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "engine.h"
#include "nwlane.h"
#include "nwpath.h"
#include "wave.h"

namespace rvn {

namespace {

__device__ __forceinline__ void nw_wsync() {
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
}

// alignment slots (lane groups) of a launch of the wave kernel: waves per SIMD by register budget x groups per wave
inline u32 path_slots(u32 R, u32 G) {
  const u32 occ = R == 1 ? 4u : 2u;
  return 256u * 4u * occ * (64u / G);
}

// A wave is split into 64 / G lane groups of G lanes; every group owns one alignment at a time (its ring of L <= G lanes).
// Control flow is uniform inside a group and may diverge between groups; every cross-lane operation below stays inside
// the caller's group, whose lanes are always active together.
template <int G>
__device__ __forceinline__ int group_prev(int v, int lig, int gbase, int L) {  // lane - 1 of the ring (lane 0 <- lane L - 1)
  return __shfl(v, lig == 0 ? gbase + L - 1 : gbase + lig - 1, 64);
}
template <int G>
__device__ __forceinline__ u32 group_max(u32 v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) {
    const u32 o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// wave-cycles per phase (pass 1 / segment re-sweeps / walks / whole kernel), printed by nw_breakpoints under RVN_NW_DEBUG
__device__ unsigned long long g_nw_phase[8];

template <int R, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(R == 1 ? 4 : 1))) void nw_path_kernel(const NwJob* __restrict__ jobs, const u32* __restrict__ idx,
                                                     u32 n_idx, const u64* __restrict__ t_words,
                                                     const u64* __restrict__ r_words, NwPm* __restrict__ ck_pm,
                                                     int* __restrict__ ck_sc, NwPm* __restrict__ seg_pm,
                                                     int* __restrict__ seg_sc, u64 seg_stride, u32 n_slots, u32 w,
                                                     NwWindowRec* __restrict__ recs, u32* __restrict__ result,
                                                     u32* __restrict__ status, u32* __restrict__ k_used,
                                                     u32* __restrict__ next) {
  constexpr int NG = 64 / G;
  __shared__ NwWalker s_walker[4][NG];
  const int lane = lane_id();
  const int group = lane / G, lig = lane % G, gbase = group * G;
  const u32 slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NG + static_cast<u32>(group);
  if (slot >= n_slots) return;
  unsigned long long c_p1 = 0, c_sw = 0, c_wk = 0, n_fast = 0, n_slow = 0;
  const unsigned long long c_begin = __builtin_readcyclecounter();
  // The groups of a wave take a bundle of 64 / G consecutive jobs together and run through its phases in lockstep
  // (pass 1, then segment by segment: sweep, walk): control flow that differs between the groups of a wave is serialised
  // by the hardware, so groups drifting into different phases would cost more than they share.  Jobs are sorted by
  // size, the jobs of a bundle have nearly the same number of columns and segments.
  for (;;) {
    u32 q0 = 0;
    if (lane == 0) q0 = atomicAdd(next, static_cast<u32>(NG));
    q0 = static_cast<u32>(__builtin_amdgcn_readfirstlane(static_cast<int>(q0)));
    if (q0 >= n_idx) break;
    const u32 q = q0 + static_cast<u32>(group);
    if (q < n_idx) {
      const u32 ji = idx[q];
      const NwJob J = jobs[ji];
      NwStore st;
      st.ck_pm = ck_pm + J.ckpt;
      st.ck_sc = ck_sc + J.ckpt;
      st.ckpt_nb = J.ckpt_nb;
      st.seg_pm = seg_pm + static_cast<u64>(slot) * seg_stride;
      st.seg_sc = seg_sc + static_cast<u64>(slot) * seg_stride;
      // ---- pass 1: distance + checkpoints; the threshold is doubled until the banded result is exact ----
      u32 k = J.k;
      NwBand B;
      NwLane<R> ln;
      u32 res = 0;
      bool ok = false;
      const unsigned long long c0 = __builtin_readcyclecounter();
      for (;;) {
        B = nw_band(J.n, J.m, k, R);
        ln.init(J, t_words, r_words, B, st, lig);
        ln.begin_sweep(0, J.m, 0);
        const int t1 = NwLane<R>::sweep_t1(B, static_cast<int>(J.m));
        for (int t = ln.t0; t <= t1; ++t) {
          const int hp = group_prev<G>(ln.xfer_last, lig, gbase, B.L);
          {  // most steps are plain block updates on every lane of the wave
            const int cls = ln.classify(t);
            if (__ballot(cls == 2) == 0) {
              if (cls == 1) ln.fast_step(t, hp);
              ++n_fast;
              continue;
            }
            ++n_slow;
          }
          const int sp = group_prev<G>(ln.score_last, lig, gbase, B.L);
          ln.step(t, hp, sp);
          ln.refresh_cache();
        }
        res = group_max<G>(ln.result) - 1u;  // exactly one lane of the group holds D(n, m) + 1
        if (res <= k) {
          ok = true;
          break;
        }
        if (k >= J.kcap) break;
        k = 2 * k < J.kcap ? 2 * k : J.kcap;
      }
      if (lig == 0) {
        result[ji] = res;
        k_used[ji] = k;
        if (!ok) status[ji] = 2;  // beyond this launch's ring: the host relaunches the job with a larger ring
      }
      nw_wsync();  // checkpoints visible to every lane
      c_p1 += __builtin_readcyclecounter() - c0;
      if (ok) {
        // ---- the walk, segment by segment from the end ----
        // The walker's state lives in LDS between the segments (it is not needed while the group re-sweeps a segment,
        // and keeping it in registers across the sweep loop costs occupancy); every lane of the group holds a copy.
        NwWalker& swk = s_walker[threadIdx.x >> 6][group];
        {
          NwWalker wk;
          wk.init(J, t_words, r_words, B, st, res, w, recs);
          if (lig == 0) swk = wk;
        }
        int rows_left = static_cast<int>(J.n);
        for (int sg = (static_cast<int>(J.m) - 1) / kNwSeg; sg >= 0 && rows_left > 0; --sg) {
          const int j0 = sg * kNwSeg;
          const int j_end = j0 + kNwSeg < static_cast<int>(J.m) ? j0 + kNwSeg : static_cast<int>(J.m);
          const unsigned long long c1 = __builtin_readcyclecounter();
          // The walk only moves up: rows below the one it stands on are never read again, and a block never depends
          // on the blocks below it — the re-sweep stops at the block of the walker's row (on average half the band).
          NwBand Bs = B;
          const int nb_need = ((rows_left - 1) >> 6) + 1;
          if (nb_need < Bs.nb) {
            Bs.nb = nb_need;
            Bs.n_super = (nb_need + R - 1) / R;
          }
          ln.B = Bs;
          ln.begin_sweep(j0, j_end, 1);
          const int t1 = NwLane<R>::sweep_t1(Bs, j_end);
          for (int t = ln.t0; t <= t1; ++t) {
            const int hp = group_prev<G>(ln.xfer_last, lig, gbase, B.L);
            {
              const int cls = ln.classify(t);
              if (__ballot(cls == 2) == 0) {
                if (cls == 1) ln.fast_step(t, hp);
                ++n_fast;
                continue;
              }
              ++n_slow;
            }
            const int sp = group_prev<G>(ln.score_last, lig, gbase, B.L);
            ln.step(t, hp, sp);
            ln.refresh_cache();
          }
          nw_wsync();  // the segment's block states (and the walker in LDS) visible to every lane
          const unsigned long long c2 = __builtin_readcyclecounter();
          c_sw += c2 - c1;
          NwWalker wk = swk;
          wk.set_segment(j0, ln.t0);
          wk.walk(lig == 0);  // every lane of the group walks the same path; its first lane writes the records
          rows_left = wk.i;
          nw_wsync();         // all reads of the scratch done before the next segment overwrites it
          c_wk += __builtin_readcyclecounter() - c2;
          if (lig == 0) swk = wk;
        }
        nw_wsync();
        NwWalker wk = swk;
        const int bad = wk.finish(lig == 0);
        if (lig == 0) status[ji] = static_cast<u32>(bad);
      }
    }
    nw_wsync();
  }
  if (lane == 0) {
    atomicAdd(&g_nw_phase[0], c_p1);
    atomicAdd(&g_nw_phase[1], c_sw);
    atomicAdd(&g_nw_phase[2], c_wk);
    atomicAdd(&g_nw_phase[3], __builtin_readcyclecounter() - c_begin);
    atomicAdd(&g_nw_phase[4], n_fast);
    atomicAdd(&g_nw_phase[5], n_slow);
  }
}

template <int R, int G>
void launch_path(Engine& e, const NwJob* d_jobs, const u32* d_idx, u32 n_idx, const ReadsDev& T, const ReadsDev& Rd, u32 w,
                 NwWindowRec* d_recs, u32* d_result, u32* d_status, u32* d_kused, u32* d_next) {
  if (n_idx == 0) return;
  hipStream_t s = e.stream;
  constexpr u32 NG = 64 / G;
  // per-group scratch of one segment: nw_seg_rows() systolic steps x G lanes x R blocks
  const u64 seg_stride = static_cast<u64>(nw_seg_rows()) * G * R;
  u32 n_slots = std::min<u32>(n_idx, path_slots(R, G));
  n_slots = ((n_slots + 4 * NG - 1) / (4 * NG)) * (4 * NG);
  NwPm* seg_pm = e.nw_pm.as<NwPm>();  // sized by nw_breakpoints for the largest launch of the batch
  int* seg_sc = e.nw_sc.as<int>();
  if (static_cast<u64>(n_slots) * seg_stride * sizeof(NwPm) > e.nw_pm.cap) throw HipError("[raven_hip] alignment path: scratch too small");
  RVN_HIP(hipMemsetAsync(d_next, 0, 4, s));
  RVN_KLAUNCH(kKNwForward, (nw_path_kernel<R, G><<<n_slots / (4 * NG), 256, 0, s>>>(
                               d_jobs, d_idx, n_idx, T.packed.as<u64>(), Rd.packed.as<u64>(), e.nw_ck_pm.as<NwPm>(),
                               e.nw_ck_sc.as<int>(), seg_pm, seg_sc, seg_stride, n_slots, w, d_recs, d_result, d_status,
                               d_kused, d_next)));
}

// One lane per alignment (nwlane.h): persistent workgroups of one wave; group g = 64 consecutive jobs of the bin's list
template <int NB>
__global__ __launch_bounds__(64) void nw_lane_kernel(const NwJob* __restrict__ jobs, const u32* __restrict__ idx, u32 n_idx,
                                                    const u64* __restrict__ t_words, const u64* __restrict__ r_words,
                                                    NwPm* __restrict__ ck_pm, int* __restrict__ ck_sc,
                                                    NwPm* __restrict__ seg_pm, int* __restrict__ seg_sc, u64 seg_stride, u32 w,
                                                    NwWindowRec* __restrict__ recs, u32* __restrict__ result,
                                                    u32* __restrict__ status, u32* __restrict__ k_used) {
  extern __shared__ __attribute__((aligned(16))) unsigned char nw_lds[];
  u64* s_pv = reinterpret_cast<u64*>(nw_lds);
  u64* s_mv = s_pv + NB * 64;
  u64* s_plo = s_mv + NB * 64;
  u64* s_phi = s_plo + NB * 64;
  int* s_sc = reinterpret_cast<int*>(s_phi + NB * 64);
  const int lane = static_cast<int>(threadIdx.x);
  const u32 n_groups = (n_idx + 63) / 64;
  for (u32 g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const u32 q = g * 64 + threadIdx.x;
    if (q >= n_idx) continue;
    const u32 ji = idx[q];
    const NwJob J = jobs[ji];
    NwLaneMem<NB, 64> M{s_pv, s_mv, s_plo, s_phi, s_sc, lane};
    NwLaneStore st;
    st.ck_pm = ck_pm + J.ckpt;
    st.ck_sc = ck_sc + J.ckpt;
    st.ckpt_nb = J.ckpt_nb;
    st.seg_pm = seg_pm + static_cast<u64>(blockIdx.x) * seg_stride;
    st.seg_sc = seg_sc + static_cast<u64>(blockIdx.x) * seg_stride;
    u32 dist = 0, ku = 0;
    const int rc = nw_lane_job<NB, 64>(J, t_words, r_words, M, st, w, recs, &dist, &ku);
    result[ji] = dist;
    k_used[ji] = ku;
    status[ji] = static_cast<u32>(rc);
  }
}

template <int NB>
void launch_lane(Engine& e, const NwJob* d_jobs, const u32* d_idx, u32 n_idx, const ReadsDev& T, const ReadsDev& Rd, u32 w,
                 NwWindowRec* d_recs, u32* d_result, u32* d_status, u32* d_kused) {
  if (n_idx == 0) return;
  hipStream_t s = e.stream;
  const size_t lds = static_cast<size_t>(NB) * 64 * 36;
  static bool attr_set = false;
  if (!attr_set) {
    RVN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(nw_lane_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds)));
    attr_set = true;
  }
  const u32 per_cu = static_cast<u32>(std::max<size_t>(1, std::min<size_t>(8, (160u << 10) / lds)));
  const u32 n_groups = (n_idx + 63) / 64;
  const u32 n_blocks = std::min<u32>(n_groups, 256u * per_cu);
  const u64 seg_stride = static_cast<u64>(kNwSeg) * NB * 64;
  NwPm* seg_pm = e.nw_pm.as<NwPm>();
  int* seg_sc = e.nw_sc.as<int>();
  if (static_cast<u64>(n_blocks) * seg_stride * sizeof(NwPm) > e.nw_pm.cap) throw HipError("[raven_hip] alignment path: scratch too small");
  RVN_KLAUNCH(kKNwLane, nw_lane_kernel<NB><<<n_blocks, 64, lds, s>>>(d_jobs, d_idx, n_idx, T.packed.as<u64>(), Rd.packed.as<u64>(),
                                                                      e.nw_ck_pm.as<NwPm>(), e.nw_ck_sc.as<int>(), seg_pm, seg_sc,
                                                                      seg_stride, w, d_recs, d_result, d_status, d_kused));
}

// largest threshold whose band (checkpoint row) fits NB blocks: (lo + hi) / 64 + 3 <= NB
u32 kcap_of_blocks(u32 n, u32 m, u32 NB) {
  const u32 d = n > m ? n - m : m - n;
  const u64 room = 64ULL * (NB - 3) + 63;  // largest lo + hi
  if (room < d) return 0;
  const u64 k = d + ((room - d) / 2) * 2 + 1;
  return static_cast<u32>(std::min<u64>(k, static_cast<u64>(n) + m));
}

// largest threshold whose band fits a ring of 64 lanes with R blocks each
u32 kcap_of(u32 n, u32 m, u32 R, u32 G = 64) {
  const u32 d = n > m ? n - m : m - n;
  // nw_ring_lanes(lo, hi, R) <= G  <=>  64R + lo + hi <= G (64R + 1);  lo + hi = 2 floor((k - d) / 2) + d
  const u64 room = static_cast<u64>(G) * (64ULL * R + 1) - 64ULL * R;
  if (room < d) return 0;
  const u64 k = d + ((room - d) / 2) * 2 + 1;  // (k - d) / 2 floors: an odd surplus costs nothing
  return static_cast<u32>(std::min<u64>(k, static_cast<u64>(n) + m));
}

}  // namespace

// Fills the band fields of `jobs` (k, kcap, R, ckpt) and produces every job's window records in d_recs
// (records of a job start at its bp_off; jobs that cannot be aligned keep all-invalid records and are counted).
void nw_breakpoints(Engine& e, const ReadsDev& T, const ReadsDev& Rd, std::vector<NwJob>& jobs, u32 w,
                    NwWindowRec* d_recs, u64 n_recs, NwStats& st) {
  st = NwStats();
  const u32 nj = static_cast<u32>(jobs.size());
  hipStream_t s = e.stream;
  RVN_HIP(hipMemsetAsync(d_recs, 0xFF, n_recs * sizeof(NwWindowRec), s));
  if (nj == 0) return;
  RVN_HIP(hipEventRecord(e.ev0, s));
  // levels of the wave kernel: (blocks per lane R, lanes per alignment G); narrow rings share a wave (64 / G alignments)
  static const u32 kRs[6] = {1, 1, 1, 2, 4, 8};
  static const u32 kGs[6] = {16, 32, 64, 64, 64, 64};
  constexpr u32 kLevels = 6;
  const double rate = e.nw_rate > 0 ? e.nw_rate : 0.13;  // first call: ONT-like; too small only costs a doubling

  // plan: first threshold from the estimate, the smallest R whose ring holds twice that, checkpoint rows for kcap
  std::vector<u32> level(nj, 0);  // index into kRs
  std::vector<u32> todo;
  // lane-per-alignment bins in use: rings of up to 16 blocks by default (HiFi-like / short alignments).  Wider rings fit
  // too few alignments per CU (LDS) to beat the wave-per-alignment kernel (measured: r02_j / r02_k); RVN_NW_LANE_BINS=4
  // enables all of them, 0 none.
  u32 max_bin = 2;
  if (const char* ev = std::getenv("RVN_NW_LANE_BINS")) max_bin = static_cast<u32>(std::atoi(ev));
  const bool lane_ok = max_bin > 0;
  auto plan = [&](NwJob& J, u32 lvl, u64 k_first) -> bool {
    const u32 d = J.n > J.m ? J.n - J.m : J.m - J.n;
    k_first = std::max<u64>(std::max<u64>(k_first, d), 16);
    k_first = std::min<u64>(k_first, static_cast<u64>(J.n) + J.m);  // D(n, m) <= n + m: that threshold never fails
    J.bin = 0;
    if (lane_ok && lvl == 0) {  // narrow band: one lane per alignment, ring of 8 / 16 / 24 / 32 blocks in LDS
      // the ring must hold the first threshold with a little headroom; when it cannot also hold a doubling, the sweep
      // starts at the ring's largest threshold right away (a wider band on an efficient kernel beats a retry)
      const u64 total = static_cast<u64>(J.n) + J.m;
      const u64 want = std::min<u64>(k_first + k_first / 8, total);
      for (u32 bin = 1; bin <= max_bin && bin <= 4; ++bin) {
        const u32 cap = kcap_of_blocks(J.n, J.m, 8 * bin);
        if (cap >= want) {
          J.bin = bin;
          J.R = 1;
          J.k = static_cast<u32>(cap >= 2 * k_first ? k_first : cap);
          J.kcap = static_cast<u32>(std::min<u64>(cap, std::max<u64>(4 * k_first, 64)));
          if (J.kcap < J.k) J.kcap = J.k;
          J.ckpt_nb = nw_ckpt_blocks(J.n, J.m, J.kcap);
          return true;
        }
      }
    }
    for (; lvl < kLevels; ++lvl) {
      const u32 cap = kcap_of(J.n, J.m, kRs[lvl], kGs[lvl]);
      // headroom of a quarter over the first threshold (the p90-based estimate): the few alignments beyond it come
      // back with status 2 and are redone one level up
      if (cap >= k_first && (cap >= k_first + k_first / 4 || lvl == kLevels - 1 || cap >= static_cast<u64>(J.n) + J.m)) {
        J.R = kRs[lvl];
        J.bin = kGs[lvl] == 64 ? 0 : kGs[lvl];  // 16 / 32: lanes per alignment of the wave kernel (0 = the whole wave)
        J.k = static_cast<u32>(std::min<u64>(k_first, cap));
        J.kcap = static_cast<u32>(std::min<u64>(cap, std::max<u64>(4 * k_first, 64)));
        J.ckpt_nb = nw_ckpt_blocks(J.n, J.m, J.kcap);
        return true;
      }
    }
    return false;
  };
  for (u32 i = 0; i < nj; ++i) {
    NwJob& J = jobs[i];
    if (J.n == 0 || J.m == 0) {
      ++st.n_unaligned;
      continue;
    }
    const u32 len = std::max(J.n, J.m);
    if (plan(J, 0, static_cast<u64>(rate * len) + 16)) todo.push_back(i);
    else ++st.n_unaligned;
  }

  std::vector<double> rates;
  std::vector<u32> h_result, h_status, h_kused, order;
  while (!todo.empty()) {
    // checkpoints of all jobs of this launch; longest alignments first (persistent waves: no long tail)
    u64 ck = 0;
    for (u32 i : todo) {
      jobs[i].ckpt = ck;
      ck += nw_ckpt_slots(jobs[i].m, jobs[i].ckpt_nb);
    }
    (void)e.nw_ck_pm.get<NwPm>(ck + 1);
    (void)e.nw_ck_sc.get<int>(ck + 1);
    st.store_bytes = std::max<u64>(st.store_bytes, ck * 20);
    NwJob* d_jobs = e.nw_jobs.get<NwJob>(nj + 1);
    RVN_HIP(hipMemcpyAsync(d_jobs, jobs.data(), static_cast<size_t>(nj) * sizeof(NwJob), hipMemcpyHostToDevice, s));
    u32* d_res = e.nw_res.get<u32>(4 * static_cast<size_t>(nj) + 16);
    u32* d_status = d_res + nj + 1;
    u32* d_kused = d_status + nj + 1;
    u32* d_idx = d_kused + nj + 1;
    u32* d_next = d_idx + nj + 1;
    order = todo;
    // classes: lane bins 1..4 first, then the wave kernel's levels; inside a class the largest jobs first
    auto level_of = [&](const NwJob& J) -> u32 {
      if (J.R == 1) return J.bin == 16 ? 0 : (J.bin == 32 ? 1 : 2);
      return J.R == 2 ? 3 : (J.R == 4 ? 4 : 5);
    };
    auto cls = [&](u32 i) -> u32 {
      const NwJob& J = jobs[i];
      return (J.bin >= 1 && J.bin <= 4) ? J.bin - 1 : 4 + level_of(J);
    };
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) {
      if (cls(a) != cls(b)) return cls(a) < cls(b);
      if (cls(a) < 4) return jobs[a].m > jobs[b].m;  // lanes of a wave run loops of similar length
      return jobs[a].m > jobs[b].m;  // longest first; the groups of a wave get jobs of (nearly) the same length
    });
    u32 coff[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 i : order) coff[cls(i) + 1]++;
    for (int x = 0; x < 10; ++x) coff[x + 1] += coff[x];
    const u32* off = coff + 4;
    {  // one scratch allocation for the largest launch (the launches are asynchronous: no reallocation in between)
      u64 need = 1;
      for (u32 bin = 1; bin <= 4; ++bin) {
        const u32 cnt = coff[bin] - coff[bin - 1];
        if (!cnt) continue;
        const size_t lds = static_cast<size_t>(8 * bin) * 64 * 36;
        const u32 per_cu = static_cast<u32>(std::max<size_t>(1, std::min<size_t>(8, (160u << 10) / lds)));
        const u64 blocks = std::min<u64>((cnt + 63) / 64, 256ULL * per_cu);
        need = std::max<u64>(need, blocks * kNwSeg * (8ULL * bin) * 64);
      }
      for (u32 x = 0; x < kLevels; ++x) {
        const u32 cnt = off[x + 1] - off[x];
        if (!cnt) continue;
        const u32 R = kRs[x], G = kGs[x], NG = 64 / G;
        u64 slots = std::min<u64>(cnt, path_slots(R, G));
        slots = ((slots + 4 * NG - 1) / (4 * NG)) * (4 * NG);
        need = std::max<u64>(need, slots * nw_seg_rows() * static_cast<u64>(G) * R);
      }
      (void)e.nw_pm.get<NwPm>(need + 1);
      (void)e.nw_sc.get<int>(need + 1);
    }
    RVN_HIP(hipMemcpyAsync(d_idx, order.data(), order.size() * 4, hipMemcpyHostToDevice, s));
    RVN_HIP(rvn_stream_sync(s));
    launch_lane<8>(e, d_jobs, d_idx + coff[0], coff[1] - coff[0], T, Rd, w, d_recs, d_res, d_status, d_kused);
    launch_lane<16>(e, d_jobs, d_idx + coff[1], coff[2] - coff[1], T, Rd, w, d_recs, d_res, d_status, d_kused);
    launch_lane<24>(e, d_jobs, d_idx + coff[2], coff[3] - coff[2], T, Rd, w, d_recs, d_res, d_status, d_kused);
    launch_lane<32>(e, d_jobs, d_idx + coff[3], coff[4] - coff[3], T, Rd, w, d_recs, d_res, d_status, d_kused);
    launch_path<1, 16>(e, d_jobs, d_idx + off[0], off[1] - off[0], T, Rd, w, d_recs, d_res, d_status, d_kused, d_next);
    launch_path<1, 32>(e, d_jobs, d_idx + off[1], off[2] - off[1], T, Rd, w, d_recs, d_res, d_status, d_kused, d_next);
    launch_path<1, 64>(e, d_jobs, d_idx + off[2], off[3] - off[2], T, Rd, w, d_recs, d_res, d_status, d_kused, d_next);
    launch_path<2, 64>(e, d_jobs, d_idx + off[3], off[4] - off[3], T, Rd, w, d_recs, d_res, d_status, d_kused, d_next);
    launch_path<4, 64>(e, d_jobs, d_idx + off[4], off[5] - off[4], T, Rd, w, d_recs, d_res, d_status, d_kused, d_next);
    launch_path<8, 64>(e, d_jobs, d_idx + off[5], off[6] - off[5], T, Rd, w, d_recs, d_res, d_status, d_kused, d_next);
    h_result.resize(nj);
    h_status.resize(nj);
    h_kused.resize(nj);
    RVN_HIP(hipMemcpyAsync(h_result.data(), d_res, static_cast<size_t>(nj) * 4, hipMemcpyDeviceToHost, s));
    RVN_HIP(hipMemcpyAsync(h_status.data(), d_status, static_cast<size_t>(nj) * 4, hipMemcpyDeviceToHost, s));
    RVN_HIP(hipMemcpyAsync(h_kused.data(), d_kused, static_cast<size_t>(nj) * 4, hipMemcpyDeviceToHost, s));
    RVN_HIP(rvn_stream_sync(s));
    ++st.n_batches;
    std::vector<u32> again;
    for (u32 i : todo) {
      NwJob& J = jobs[i];
      // cells of every attempt: thresholds k, 2k, .. up to the one used
      for (u64 kk = J.k;; kk = std::min<u64>(2 * kk, J.kcap)) {
        st.band_cells += static_cast<u64>(J.m) * (nw_band_lo(J.n, J.m, static_cast<u32>(kk)) + nw_band_hi(J.n, J.m, static_cast<u32>(kk)) + 1);
        if (kk >= h_kused[i]) break;
        ++st.n_retries;
      }
      if (h_status[i] == 2) {  // distance above this launch's largest threshold: next blocks-per-lane level
        const u64 k2 = static_cast<u64>(h_kused[i]) * 2;
        const bool was_lane = J.bin >= 1 && J.bin <= 4;
        const u32 lvl = was_lane ? 0 : level_of(J);
        if (J.kcap >= static_cast<u64>(J.n) + J.m ||
            !plan(J, was_lane ? 0 : (J.kcap < kcap_of(J.n, J.m, kRs[lvl], kGs[lvl]) ? lvl : lvl + 1), k2))
          ++st.n_unaligned;
        else again.push_back(i);
      } else if (h_status[i] != 0) {
        throw HipError("[raven_hip] alignment path: the walk left the stored band (internal error)");
      } else {
        ++st.n_aligned;
        st.sum_distance += h_result[i];
        st.band_cells += static_cast<u64>(J.m) * (nw_band_lo(J.n, J.m, h_kused[i]) + nw_band_hi(J.n, J.m, h_kused[i]) + 1);  // the re-sweeps
        rates.push_back(static_cast<double>(h_result[i]) / std::max(J.n, J.m));
      }
    }
    todo.swap(again);
  }
  if (rates.size() >= 32) {  // threshold estimate for the next call: most alignments succeed at the first attempt
    std::sort(rates.begin(), rates.end());
    e.nw_rate = rates[std::min(rates.size() - 1, static_cast<size_t>(rates.size() * 0.9))] * 1.05 + 0.002;
  }
  RVN_HIP(hipEventRecord(e.ev1, s));
  RVN_HIP(hipEventSynchronize(e.ev1));
  float ms = 0;
  RVN_HIP(hipEventElapsedTime(&ms, e.ev0, e.ev1));
  st.ms = ms;
  if (std::getenv("RVN_NW_DEBUG")) {  // cumulative since the library was loaded
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    RVN_HIP(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_nw_phase), sizeof(ph)));
    std::fprintf(stderr, "[raven_hip] nw_path wave-cycles (cumulative): pass1 %.3e  re-sweep %.3e  walk %.3e  total %.3e; fast steps %.3e slow %.3e; this call %.1f ms, %u jobs\n",
                 static_cast<double>(ph[0]), static_cast<double>(ph[1]), static_cast<double>(ph[2]), static_cast<double>(ph[3]), static_cast<double>(ph[4]), static_cast<double>(ph[5]), ms, nj);
  }
}

// ---- CPU stepper of the same code (test hook rvn_test_nw_breakpoints): 64 emulated lanes, host arrays --------------
template <int R>
static int emulate_job(const NwJob& J, const u64* t_words, const u64* r_words, u32 w, NwWindowRec* recs, u32* distance,
                       u32* band) {
  std::vector<NwPm> ck_pm(nw_ckpt_slots(J.m, J.ckpt_nb) + 1), seg_pm(static_cast<size_t>(nw_seg_rows()) * 64 * R + 1);
  std::vector<int> ck_sc(ck_pm.size()), seg_sc(seg_pm.size());
  NwStore st{ck_pm.data(), ck_sc.data(), J.ckpt_nb, seg_pm.data(), seg_sc.data()};
  std::vector<NwLane<R>> lanes(64);
  std::vector<int> hp(64), sp(64);
  NwBand B;
  auto sweep = [&](int j0, int j_end, int mode) {
    for (int l = 0; l < 64; ++l) lanes[l].begin_sweep(j0, j_end, mode);
    const int t1 = NwLane<R>::sweep_t1(B, j_end);
    for (int t = lanes[0].t0; t <= t1; ++t) {
      for (int l = 0; l < 64; ++l) {  // the shuffles read the producer's values of the previous step
        const int src = l == 0 ? B.L - 1 : l - 1;
        hp[l] = lanes[src].xfer_last;
        sp[l] = lanes[src].score_last;
      }
      bool any_slow = false;
      for (int l = 0; l < 64; ++l) any_slow = any_slow || lanes[l].classify(t) == 2;
      if (!any_slow) {  // the kernel's short path: no lane of the wave needs more than the plain block update
        for (int l = 0; l < 64; ++l)
          if (lanes[l].classify(t) == 1) lanes[l].fast_step(t, hp[l]);
      } else {
        for (int l = 0; l < 64; ++l) {
          lanes[l].step(t, hp[l], sp[l]);
          lanes[l].refresh_cache();
        }
      }
      if (mode == 1 && static_cast<u64>(t - lanes[0].t0) >= nw_seg_rows()) return false;  // scratch rows exceeded
    }
    return true;
  };
  u32 k = J.k, res = 0;
  for (;;) {
    B = nw_band(J.n, J.m, k, R);
    if (B.L > 64) return -2;
    for (int l = 0; l < 64; ++l) lanes[l].init(J, t_words, r_words, B, st, l);
    sweep(0, static_cast<int>(J.m), 0);
    res = 0;
    for (int l = 0; l < 64; ++l) res = std::max(res, lanes[l].result);
    res -= 1u;
    if (res <= k) break;
    if (k >= J.kcap) return -3;
    k = std::min<u32>(2 * k, J.kcap);
  }
  *distance = res;
  if (band) {
    band[0] = k;
    band[1] = static_cast<u32>(B.L);
    band[2] = R;
  }
  NwWalker wk;
  wk.init(J, t_words, r_words, B, st, res, w, recs);
  for (int sg = (static_cast<int>(J.m) - 1) / kNwSeg; sg >= 0 && wk.i > 0; --sg) {
    const int j0 = sg * kNwSeg;
    const int j_end = std::min<int>(j0 + kNwSeg, static_cast<int>(J.m));
    {  // as in the kernel: the re-sweep stops at the block of the walker's row
      NwBand Bs = B;
      const int nb_need = ((wk.i - 1) >> 6) + 1;
      if (nb_need < Bs.nb) {
        Bs.nb = nb_need;
        Bs.n_super = (nb_need + R - 1) / R;
      }
      for (int l = 0; l < 64; ++l) lanes[l].B = Bs;
      const NwBand keep = B;
      B = Bs;
      const bool ok_sweep = sweep(j0, j_end, 1);
      B = keep;
      if (!ok_sweep) return -4;
    }
    wk.set_segment(j0, lanes[0].t0);
    wk.walk(true);
  }
  return wk.finish(true);
}

template <int NB>
static int emulate_lane_job(const NwJob& J, const u64* t_words, const u64* r_words, u32 w, NwWindowRec* recs, u32* distance,
                            u32* band) {
  std::vector<u64> pv(NB), mv(NB), plo(NB), phi(NB);
  std::vector<int> sc(NB);
  std::vector<NwPm> ck_pm(nw_ckpt_slots(J.m, J.ckpt_nb) + 1), seg_pm(static_cast<size_t>(kNwSeg) * NB + 1);
  std::vector<int> ck_sc(ck_pm.size()), seg_sc(seg_pm.size());
  NwLaneMem<NB, 1> M{pv.data(), mv.data(), plo.data(), phi.data(), sc.data(), 0};
  NwLaneStore st{ck_pm.data(), ck_sc.data(), J.ckpt_nb, seg_pm.data(), seg_sc.data()};
  u32 ku = 0;
  const int rcode = nw_lane_job<NB, 1>(J, t_words, r_words, M, st, w, recs, distance, &ku);
  if (band) {
    band[0] = ku;
    band[1] = NB;
    band[2] = 0;
  }
  return rcode == 2 ? -3 : rcode;
}

int nw_breakpoints_host(const u64* t_words, u32 t_len, const u64* r_words, u32 r_len, u32 t_begin, u32 n, u32 q_begin,
                        u32 m, int rc, u32 w, u32 k, int force_R, NwWindowRec* recs, u32* distance, u32* band) {
  (void)t_len;
  if (n == 0 || m == 0) return -1;
  if (force_R < 0) {  // the lane-per-alignment kernel's code with a ring of -force_R blocks
    const u32 NB = static_cast<u32>(-force_R);
    if (NB != 8 && NB != 16 && NB != 24 && NB != 32) return -2;
    NwJob J{};
    J.t_begin = t_begin;
    J.n = n;
    J.q_begin = q_begin;
    J.m = m;
    J.r_len = r_len;
    J.rc = rc ? 1 : 0;
    J.R = 1;
    J.bin = NB / 8;
    J.n_windows = (t_begin + n - 1) / w - t_begin / w + 1;
    for (u32 x = 0; x < J.n_windows; ++x) {
      recs[x].first_t = recs[x].first_q = recs[x].last_t = recs[x].last_q = 0xFFFFFFFFu;
      for (int g = 0; g < 8; ++g) recs[x].grid[g] = 0xFFFFu;
    }
    const u32 d = n > m ? n - m : m - n;
    const u32 cap = kcap_of_blocks(n, m, NB);
    const u64 kk = std::min<u64>(std::max<u64>(std::max<u64>(k, d), 1), static_cast<u64>(n) + m);
    if (cap < kk) return -2;
    J.k = static_cast<u32>(kk);
    J.kcap = cap;
    J.ckpt_nb = nw_ckpt_blocks(n, m, J.kcap);
    switch (NB) {
      case 8: return emulate_lane_job<8>(J, t_words, r_words, w, recs, distance, band);
      case 16: return emulate_lane_job<16>(J, t_words, r_words, w, recs, distance, band);
      case 24: return emulate_lane_job<24>(J, t_words, r_words, w, recs, distance, band);
      default: return emulate_lane_job<32>(J, t_words, r_words, w, recs, distance, band);
    }
  }
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
  static const u32 kRs[4] = {1, 2, 4, 8};  // the stepper emulates whole-wave rings (lane groups only change which lanes a ring uses)
  u32 lvl = 0;
  if (force_R) {
    while (lvl < 4 && kRs[lvl] != static_cast<u32>(force_R)) ++lvl;
    if (lvl == 4) return -2;
  }
  u64 kk = std::min<u64>(std::max<u64>(std::max<u64>(k, d), 1), static_cast<u64>(n) + m);
  for (; lvl < 4; ++lvl) {
    J.R = kRs[lvl];
    const u32 cap = kcap_of(n, m, J.R);
    if (cap < kk) {
      if (force_R) return -2;
      continue;
    }
    J.k = static_cast<u32>(kk);
    J.kcap = cap;
    J.ckpt_nb = nw_ckpt_blocks(n, m, J.kcap);
    int rcode;
    switch (J.R) {
      case 1: rcode = emulate_job<1>(J, t_words, r_words, w, recs, distance, band); break;
      case 2: rcode = emulate_job<2>(J, t_words, r_words, w, recs, distance, band); break;
      case 4: rcode = emulate_job<4>(J, t_words, r_words, w, recs, distance, band); break;
      default: rcode = emulate_job<8>(J, t_words, r_words, w, recs, distance, band); break;
    }
    if (rcode == -3 && !force_R) {  // distance above this R's largest threshold
      kk = std::min<u64>(static_cast<u64>(cap) * 2, static_cast<u64>(n) + m);
      continue;
    }
    return rcode;
  }
  return -3;
}

}  // namespace rvn
