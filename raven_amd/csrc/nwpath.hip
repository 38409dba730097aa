// nwpath.hip — kernels and host driver of the alignment-path stage of a polishing round (see nwpath.h): for every
// read's best overlap, the global alignment path against its target span and racon's per-window breakpoints
// (racon Overlap::find_breaking_points, reached from RavenLib/src/polish.cc:51).
//
//   nw_forward_kernel<R>   one wave per alignment: banded Myers sweep that stores every block's (Pv, Mv, score)
//   nw_traceback_kernel    one thread per alignment: path walk + breakpoints + band-guide samples -> NwWindowRec
//
// Host side: band thresholds k from the running error-rate estimate of the engine (first call: a pilot sample with
// threshold doubling), jobs packed into batches that fit the store budget, failed attempts (distance > k) redone
// with 2k — the result is always the exact optimal path, the estimate only decides how much band is computed.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "engine.h"
#include "nwpath.h"
#include "wave.h"

namespace rvn {

namespace {

template <int R>
__global__ __launch_bounds__(256) void nw_forward_kernel(const NwJob* __restrict__ jobs, const u32* __restrict__ idx,
                                                        u32 n_idx, const u64* __restrict__ t_words,
                                                        const u64* __restrict__ r_words, NwPm* __restrict__ pm,
                                                        int* __restrict__ sc, u32* __restrict__ result) {
  const u32 q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n_idx) return;
  const u32 ji = idx[q];
  const NwJob J = jobs[ji];
  const int lane = lane_id();
  NwLane<R> ln;
  ln.init(J, t_words, r_words, pm, sc, lane);
  const int src = lane == 0 ? static_cast<int>(J.L) - 1 : lane - 1;
  const long long t_end = static_cast<long long>(J.m) + ln.n_super;
  for (long long t = 0; t < t_end; ++t) {
    const int hp = __shfl(ln.hout_last, src, 64);
    const int sp = __shfl(ln.score_last, src, 64);
    ln.step(t, hp, sp);
  }
  const u32 res = wave_max(ln.result);  // exactly one lane holds D(n, m) + 1
  if (lane == 0) result[ji] = res - 1u;
}

__global__ __launch_bounds__(64) void nw_traceback_kernel(const NwJob* __restrict__ jobs, u32 n_jobs,
                                                         const u64* __restrict__ t_words, const u64* __restrict__ r_words,
                                                         const NwPm* __restrict__ pm, const int* __restrict__ sc,
                                                         const u32* __restrict__ result, u32 w,
                                                         NwWindowRec* __restrict__ recs, u32* __restrict__ status) {
  const u32 ji = blockIdx.x * 64 + threadIdx.x;
  if (ji >= n_jobs) return;
  const NwJob J = jobs[ji];
  const u32 d = result[ji];
  if (d > J.k) {  // the band was too narrow for this pair: redone with a larger threshold
    status[ji] = 2;
    return;
  }
  status[ji] = static_cast<u32>(nw_traceback(J, t_words, r_words, pm, sc, d, w, recs));
}

template <int R>
void launch_forward(Engine& e, const NwJob* d_jobs, const u32* d_idx, u32 n_idx, const ReadsDev& T, const ReadsDev& Rd,
                    NwPm* pm, int* sc, u32* d_result) {
  if (n_idx == 0) return;
  RVN_KLAUNCH(kKNwForward, nw_forward_kernel<R><<<div_up(n_idx, 4), 256, 0, e.stream>>>(
                               d_jobs, d_idx, n_idx, T.packed.as<u64>(), Rd.packed.as<u64>(), pm, sc, d_result));
}

// smallest supported R whose ring holds the band of threshold k; 0 = beyond the kernel (k > ~32 000)
u32 pick_R(u32 lo, u32 hi, u32* L) {
  for (u32 R : {1u, 2u, 4u, 8u}) {
    const u32 l = nw_ring_lanes(lo, hi, R);
    if (l <= 64) {
      *L = l < 1 ? 1 : l;
      return R;
    }
  }
  return 0;
}

}  // namespace

// Fills the band fields of `jobs` (k, lo, hi, L, R, store) and produces every job's window records in d_recs
// (records of a job start at its bp_off; jobs that cannot be aligned keep all-invalid records and are counted).
void nw_breakpoints(Engine& e, const ReadsDev& T, const ReadsDev& Rd, std::vector<NwJob>& jobs, u32 w,
                    NwWindowRec* d_recs, u64 n_recs, NwStats& st) {
  st = NwStats();
  const u32 nj = static_cast<u32>(jobs.size());
  hipStream_t s = e.stream;
  RVN_HIP(hipMemsetAsync(d_recs, 0xFF, n_recs * sizeof(NwWindowRec), s));
  if (nj == 0) return;
  RVN_HIP(hipEventRecord(e.ev0, s));

  size_t free_b = 0, total_b = 0;
  RVN_HIP(hipMemGetInfo(&free_b, &total_b));
  u64 budget = std::min<u64>(32ULL << 30, static_cast<u64>((free_b + e.nw_pm.cap + e.nw_sc.cap) * 0.4));
  if (const char* ev = std::getenv("RVN_NW_BUDGET_MB")) budget = static_cast<u64>(std::atoll(ev)) << 20;
  const u64 budget_slots = std::max<u64>(budget / 20, 1);

  auto set_band = [&](NwJob& J, u64 k) -> bool {
    const u32 d = J.n > J.m ? J.n - J.m : J.m - J.n;
    k = std::max<u64>(k, d);
    k = std::min<u64>(k, static_cast<u64>(J.n) + J.m);  // D(n, m) <= n + m: this threshold always succeeds
    J.k = static_cast<u32>(k);
    J.lo = nw_band_lo(J.n, J.m, J.k);
    J.hi = nw_band_hi(J.n, J.m, J.k);
    J.R = pick_R(J.lo, J.hi, &J.L);
    return J.R != 0;
  };

  // thresholds: from the engine's running estimate of distance / length, or a pilot sample on the first call
  std::vector<u32> pending;
  std::vector<u32> later;
  const bool pilot = e.nw_rate < 0;
  std::vector<u8> in_pilot(nj, 0);
  if (pilot) {
    const u32 n_pilot = std::min<u32>(nj, 256);
    for (u32 x = 0; x < n_pilot; ++x) in_pilot[static_cast<u64>(x) * nj / n_pilot] = 1;
  }
  for (u32 i = 0; i < nj; ++i) {
    NwJob& J = jobs[i];
    if (J.n == 0 || J.m == 0) {
      ++st.n_unaligned;
      continue;
    }
    const u32 len = std::max(J.n, J.m);
    const u64 k0 = pilot ? std::max<u64>(64, static_cast<u64>(0.03 * len))
                         : std::max<u64>(32, static_cast<u64>(e.nw_rate * len) + 16);
    if (!set_band(J, k0)) {
      ++st.n_unaligned;
      continue;
    }
    if (pilot && !in_pilot[i]) later.push_back(i);
    else pending.push_back(i);
  }

  std::vector<double> rates;
  std::vector<NwJob> batch;
  std::vector<u32> batch_src, idxR[4], h_result, h_status;
  while (!pending.empty() || !later.empty()) {
    if (pending.empty()) {  // the pilot is done: thresholds of everything else from its distances
      if (!rates.empty()) {
        std::sort(rates.begin(), rates.end());
        e.nw_rate = rates[std::min(rates.size() - 1, static_cast<size_t>(rates.size() * 0.9))] * 1.1 + 0.002;
      } else {
        e.nw_rate = 0.15;
      }
      for (u32 i : later) {
        NwJob& J = jobs[i];
        const u32 len = std::max(J.n, J.m);
        if (set_band(J, std::max<u64>(32, static_cast<u64>(e.nw_rate * len) + 16))) pending.push_back(i);
        else ++st.n_unaligned;
      }
      later.clear();
      continue;
    }
    // ---- one batch: as many pending jobs as the store budget holds ----
    batch.clear();
    batch_src.clear();
    for (auto& v : idxR) v.clear();
    u64 slots = 0;
    size_t taken = 0;
    for (; taken < pending.size(); ++taken) {
      NwJob& J = jobs[pending[taken]];
      const u64 need = nw_store_slots(J.n, J.m, J.L, J.R);
      if (!batch.empty() && slots + need > budget_slots) break;
      J.store = slots;
      slots += need;
      const u32 bi = static_cast<u32>(batch.size());
      idxR[J.R == 1 ? 0 : (J.R == 2 ? 1 : (J.R == 4 ? 2 : 3))].push_back(bi);
      batch.push_back(J);
      batch_src.push_back(pending[taken]);
    }
    pending.erase(pending.begin(), pending.begin() + taken);
    const u32 nb = static_cast<u32>(batch.size());
    NwPm* pm = e.nw_pm.get<NwPm>(slots + 1);
    int* sc = e.nw_sc.get<int>(slots + 1);
    NwJob* d_jobs = e.nw_jobs.get<NwJob>(nb + 1);
    u32* d_res = e.nw_res.get<u32>(3 * static_cast<size_t>(nb) + 4);
    u32* d_status = d_res + nb + 1;
    u32* d_idx = d_status + nb + 1;
    RVN_HIP(hipMemcpyAsync(d_jobs, batch.data(), nb * sizeof(NwJob), hipMemcpyHostToDevice, s));
    {
      std::vector<u32> all_idx;
      u32 off[5] = {0, 0, 0, 0, 0};
      for (int x = 0; x < 4; ++x) {
        all_idx.insert(all_idx.end(), idxR[x].begin(), idxR[x].end());
        off[x + 1] = static_cast<u32>(all_idx.size());
      }
      RVN_HIP(hipMemcpyAsync(d_idx, all_idx.data(), all_idx.size() * 4, hipMemcpyHostToDevice, s));
      RVN_HIP(hipStreamSynchronize(s));  // all_idx is a local
      launch_forward<1>(e, d_jobs, d_idx + off[0], off[1] - off[0], T, Rd, pm, sc, d_res);
      launch_forward<2>(e, d_jobs, d_idx + off[1], off[2] - off[1], T, Rd, pm, sc, d_res);
      launch_forward<4>(e, d_jobs, d_idx + off[2], off[3] - off[2], T, Rd, pm, sc, d_res);
      launch_forward<8>(e, d_jobs, d_idx + off[3], off[4] - off[3], T, Rd, pm, sc, d_res);
    }
    RVN_KLAUNCH(kKNwTraceback, nw_traceback_kernel<<<div_up(nb, 64), 64, 0, s>>>(
                                   d_jobs, nb, T.packed.as<u64>(), Rd.packed.as<u64>(), pm, sc, d_res, w, d_recs, d_status));
    h_result.resize(nb);
    h_status.resize(nb);
    RVN_HIP(hipMemcpyAsync(h_result.data(), d_res, nb * 4, hipMemcpyDeviceToHost, s));
    RVN_HIP(hipMemcpyAsync(h_status.data(), d_status, nb * 4, hipMemcpyDeviceToHost, s));
    RVN_HIP(hipStreamSynchronize(s));
    ++st.n_batches;
    st.store_bytes = std::max<u64>(st.store_bytes, slots * 20);
    for (u32 bi = 0; bi < nb; ++bi) {
      NwJob& J = jobs[batch_src[bi]];
      st.band_cells += static_cast<u64>(J.m) * (J.lo + J.hi + 1);
      if (h_status[bi] == 2) {  // distance above the threshold: double it
        ++st.n_retries;
        const u64 k2 = std::max<u64>(static_cast<u64>(J.k) * 2, 64);
        if (J.k >= static_cast<u64>(J.n) + J.m || !set_band(J, k2)) ++st.n_unaligned;
        else pending.push_back(batch_src[bi]);
      } else if (h_status[bi] != 0) {
        throw HipError("[raven_hip] alignment path: traceback left the stored band (internal error)");
      } else {
        ++st.n_aligned;
        st.sum_distance += h_result[bi];
        rates.push_back(static_cast<double>(h_result[bi]) / std::max(J.n, J.m));
      }
    }
  }
  if (!pilot && rates.size() >= 64) {  // keep the estimate current (rounds get more accurate)
    std::sort(rates.begin(), rates.end());
    e.nw_rate = rates[std::min(rates.size() - 1, static_cast<size_t>(rates.size() * 0.9))] * 1.1 + 0.002;
  }
  RVN_HIP(hipEventRecord(e.ev1, s));
  RVN_HIP(hipEventSynchronize(e.ev1));
  float ms = 0;
  RVN_HIP(hipEventElapsedTime(&ms, e.ev0, e.ev1));
  st.ms = ms;
}

// ---- CPU stepper of the same code (test hook rvn_test_nw_breakpoints): 64 emulated lanes, host arrays --------------
template <int R>
static u32 emulate_forward(const NwJob& J, const u64* t_words, const u64* r_words, NwPm* pm, int* sc) {
  std::vector<NwLane<R>> lanes(64);
  for (int l = 0; l < 64; ++l) lanes[l].init(J, t_words, r_words, pm, sc, l);
  const long long t_end = static_cast<long long>(J.m) + lanes[0].n_super;
  std::vector<int> hp(64), sp(64);
  for (long long t = 0; t < t_end; ++t) {
    for (int l = 0; l < 64; ++l) {  // the shuffles read the producer's values of the previous step
      const int src = l == 0 ? static_cast<int>(J.L) - 1 : l - 1;
      hp[l] = lanes[src].hout_last;
      sp[l] = lanes[src].score_last;
    }
    for (int l = 0; l < 64; ++l) lanes[l].step(t, hp[l], sp[l]);
  }
  u32 res = 0;
  for (int l = 0; l < 64; ++l) res = std::max(res, lanes[l].result);
  return res - 1u;
}

int nw_breakpoints_host(const u64* t_words, u32 t_len, const u64* r_words, u32 r_len, u32 t_begin, u32 n, u32 q_begin,
                        u32 m, int rc, u32 w, u32 k, int force_R, NwWindowRec* recs, u32* distance, u32* band) {
  NwJob J{};
  J.t_word = 0;
  J.r_word = 0;
  J.store = 0;
  J.bp_off = 0;
  J.t_begin = t_begin;
  J.n = n;
  J.q_begin = q_begin;
  J.m = m;
  J.r_len = r_len;
  J.rc = rc ? 1 : 0;
  (void)t_len;
  if (n == 0 || m == 0) return -1;
  const u32 d = n > m ? n - m : m - n;
  J.n_windows = (t_begin + n - 1) / w - t_begin / w + 1;
  u64 kk = std::max<u64>(k, d);
  for (;;) {
    kk = std::min<u64>(kk, static_cast<u64>(n) + m);
    J.k = static_cast<u32>(kk);
    J.lo = nw_band_lo(n, m, J.k);
    J.hi = nw_band_hi(n, m, J.k);
    J.R = force_R ? static_cast<u32>(force_R) : pick_R(J.lo, J.hi, &J.L);
    if (force_R) J.L = nw_ring_lanes(J.lo, J.hi, J.R);
    if (J.R == 0 || J.L > 64) return -2;
    const u64 slots = nw_store_slots(n, m, J.L, J.R);
    std::vector<NwPm> pm(slots + 1);
    std::vector<int> sc(slots + 1);
    u32 res = 0;
    switch (J.R) {
      case 1: res = emulate_forward<1>(J, t_words, r_words, pm.data(), sc.data()); break;
      case 2: res = emulate_forward<2>(J, t_words, r_words, pm.data(), sc.data()); break;
      case 4: res = emulate_forward<4>(J, t_words, r_words, pm.data(), sc.data()); break;
      case 8: res = emulate_forward<8>(J, t_words, r_words, pm.data(), sc.data()); break;
      default: return -2;
    }
    if (res > J.k) {
      if (J.k >= static_cast<u64>(n) + m) return -3;
      kk = std::max<u64>(2 * kk, 64);
      continue;
    }
    *distance = res;
    if (band) {
      band[0] = J.k;
      band[1] = J.L;
      band[2] = J.R;
    }
    return nw_traceback(J, t_words, r_words, pm.data(), sc.data(), res, w, recs);
  }
}

}  // namespace rvn
