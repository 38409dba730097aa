// index.hip — minimizer index: stable sort by value, distinct-key table, direct-address bucket table,
// occurrence filter (replaces ram::MinimizerEngine::Minimize(first,last,minhash) index build and
// ::Filter(f); call sites RavenLib/src/construct.cc:42-44).
//
// ram buckets by the low 14 bits, stable-sorts each bucket by value and keeps an unordered_map
// value -> (offset,count) into an origins array ordered by (read iteration order, position).  The same
// mapping is obtained here from ONE stable device radix sort of the (value, origin) stream (already in
// iteration/position order), a run-head compaction (distinct keys + start offsets) and a direct-address
// table over the top bits of the (uniformly distributed) hash: lookup = 2 table loads + a <=few-entry scan.
#include <algorithm>
#include <vector>

#include "engine.h"
#include "wave.h"

namespace rvn {

namespace {

template <typename V>
__global__ void heads_kernel(const V* __restrict__ val, u64 n, u8* __restrict__ flags) {
  u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = (i == 0 || val[i] != val[i - 1]) ? 1 : 0;
}

template <typename V>
__global__ void unique_kernel(const V* __restrict__ val, const u8* __restrict__ flags, const u32* __restrict__ scan,
                              u64 n, V* __restrict__ u_val, u32* __restrict__ u_start, u32 u) {
  u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n && flags[i]) {
    u32 j = scan[i];
    u_val[j] = val[i];
    u_start[j] = static_cast<u32>(i);
  }
  if (i == 0) u_start[u] = static_cast<u32>(n);
}

// table[b] = first distinct-key index j with (u_val[j] >> shift) >= b, for b in [0, B]; B = 1 << bits.
// Minimizer hashes are window MINIMA, so they crowd the low end of the value range and the top of
// the table is sparse: a lane may own a gap of 10^3..10^6 buckets.  Short gaps are filled by their
// lane; long gaps are filled by the whole wave (64 coalesced stores per step).
template <typename V>
__global__ __launch_bounds__(256) void table_kernel(const V* __restrict__ u_val, u32 u, int shift,
                                                   u32* __restrict__ table) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = lane_id();
  u32 start = 0, len = 0;
  if (j < u) {
    const long long bj = static_cast<long long>(static_cast<u64>(u_val[j]) >> shift);
    const long long bp = j ? static_cast<long long>(static_cast<u64>(u_val[j - 1]) >> shift) : -1;
    start = static_cast<u32>(bp + 1);
    len = static_cast<u32>(bj - bp);
  }
  if (len <= 8) {
    for (u32 i = 0; i < len; ++i) table[start + i] = j;
  }
  unsigned long long longmask = __ballot(len > 8);
  while (longmask) {
    const int l = __ffsll(static_cast<long long>(longmask)) - 1;
    longmask &= longmask - 1;
    const u32 s = __shfl(start, l, 64), n = __shfl(len, l, 64), v = __shfl(j, l, 64);
    for (u32 i = lane; i < n; i += 64) table[s + i] = v;
  }
}

// buckets above the largest key: table[b] = u for b in (u_val[u-1] >> shift, B]
template <typename V>
__global__ __launch_bounds__(256) void table_tail_kernel(const V* __restrict__ u_val, u32 u, int shift, u32 B,
                                                        u32* __restrict__ table) {
  const u64 first = (static_cast<u64>(u_val[u - 1]) >> shift) + 1;
  for (u64 b = first + static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x; b <= B;
       b += static_cast<u64>(gridDim.x) * blockDim.x)
    table[b] = u;
}

// direct-address form of the index: direct[v] = (run length << 32) | first entry of v's run in the sorted origins
template <typename V>
__global__ __launch_bounds__(256) void direct_fill_kernel(const V* __restrict__ u_val, const u32* __restrict__ u_start, u32 u,
                                                         u64* __restrict__ direct) {
  const u32 j = blockIdx.x * 256 + threadIdx.x;
  if (j >= u) return;
  const u32 a = u_start[j], b = u_start[j + 1];
  direct[static_cast<u64>(u_val[j])] = (static_cast<u64>(b - a) << 32) | a;
}

__global__ void table_empty_kernel(u32* table, u32 B) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= B) table[i] = 0;
}

constexpr u32 kHistBins = 65536;

// count-of-counts histogram of run lengths (bin kHistBins-1 = overflow ">= 65535")
// (s_org != nullptr: the sorted origins may hold query-only entries, kForeignFlag — a run's count is its MEMBERS', and a
// run without members is no key of the index)
__global__ __launch_bounds__(256) void occ_hist_kernel(const u32* __restrict__ u_start, u32 u,
                                                      u32* __restrict__ hist, u32* __restrict__ overflow_list,
                                                      u32* __restrict__ overflow_n, u32 overflow_cap,
                                                      const u64* __restrict__ s_org) {
  __shared__ u32 lh[256];
  lh[threadIdx.x] = 0;
  __syncthreads();
  for (u32 j = blockIdx.x * 256 + threadIdx.x; j < u; j += gridDim.x * 256) {
    u32 c = u_start[j + 1] - u_start[j];
    if (s_org) {
      c -= run_foreign_prefix(s_org, u_start[j], c);
      if (c == 0) continue;
    }
    if (c < 256) {
      atomicAdd(&lh[c], 1u);
    } else if (c < kHistBins - 1) {
      atomicAdd(&hist[c], 1u);
    } else {
      atomicAdd(&hist[kHistBins - 1], 1u);
      u32 slot = atomicAdd(overflow_n, 1u);
      if (slot < overflow_cap) overflow_list[slot] = c;
    }
  }
  __syncthreads();
  if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// (1-f) quantile of the per-key counts from the count-of-counts histogram: smallest c with
// cum(c) > nth.  One workgroup; out[0] = c, out[1] = number of keys with count < c ("below").
__global__ __launch_bounds__(256) void occ_quantile_kernel(const u32* __restrict__ hist, u64 nth,
                                                          u64* __restrict__ out) {
  __shared__ u64 smem[4];
  __shared__ u64 s_base[256];
  const u32 t = threadIdx.x;
  u64 sum = 0;
  for (u32 i = 0; i < kHistBins / 256; ++i) sum += hist[t * (kHistBins / 256) + i];
  u64 total;
  const u64 ex = block_exclusive_sum_256<u64>(sum, smem, &total);
  s_base[t] = ex;
  __syncthreads();
  if (ex <= nth && nth < ex + sum) {
    u64 cum = ex;
    for (u32 i = 0; i < kHistBins / 256; ++i) {
      const u32 c = t * (kHistBins / 256) + i;
      const u64 h = hist[c];
      if (cum + h > nth) {
        out[0] = c;
        out[1] = cum;
        break;
      }
      cum += h;
    }
  }
}

// run boundaries: u_start[U+1] (and, for the probe path, the distinct keys u_val[U])
template <typename V>
void index_runs_impl(Engine& e) {
  hipStream_t s = e.stream;
  Index& ix = e.index;
  const u64 m = ix.m;
  StageTimer t(e, StageTimes::kIndex);
  const V* sv = ix.s_val[ix.cur].as<V>();
  u8* flags = e.tmp_c.get<u8>(m + 1);
  u32* fscan = e.tmp_d.get<u32>(m + 1);
  RVN_KLAUNCH(kKHeads, heads_kernel<V><<<div_up(m, 256), 256, 0, s>>>(sv, m, flags));
  exclusive_scan_u8_u32(flags, fscan, m, e.scan_tmp, s);
  const u32 u = static_cast<u32>(read_back(e, fscan + m, 4));
  ix.u = u;
  V* u_val = ix.u_val.get<V>(static_cast<size_t>(u) + 1);
  u32* u_start = ix.u_start.get<u32>(static_cast<size_t>(u) + 2);
  RVN_KLAUNCH(kKUnique, unique_kernel<V><<<div_up(m, 256), 256, 0, s>>>(sv, flags, fscan, m, u_val, u_start, u));
  t.stop();
}

constexpr u32 kDirectMinKeys = 8u << 20;  // distinct values from which the direct-address table is built

template <typename V>
void index_table_impl(Engine& e) {
  hipStream_t s = e.stream;
  Index& ix = e.index;
  if (ix.table_built || ix.m == 0) return;
  StageTimer t(e, StageTimes::kIndex);
  const u32 u = static_cast<u32>(ix.u);
  int bits = 1;
  while ((1ULL << bits) < 2ULL * u) ++bits;
  bits = std::max(8, bits);
  bits = std::min(bits, std::min<int>(2 * e.k, 26));
  ix.table_bits = bits;
  ix.shift = 2 * e.k - bits;
  const u32 B = 1u << bits;
  u32* table = ix.table.get<u32>(static_cast<size_t>(B) + 2);
  const V* u_val = ix.u_val.as<V>();
  RVN_KLAUNCH(kKTable, table_kernel<V><<<div_up(u, 256), 256, 0, s>>>(u_val, u, ix.shift, table);
              table_tail_kernel<V><<<256, 256, 0, s>>>(u_val, u, ix.shift, B, table));
  ix.table_built = true;
  // A probe through the bucket table touches four to six cache lines (table, a few entries of u_val, u_start, the origins);
  // with every possible value addressed directly it is the entry and the origins.  4^k entries of 8 bytes = 8 GB at k = 15:
  // worth its memset and fill (~6 ms at 100 Mb x 30) where a pass probes hundreds of millions of times, i.e. for large indexes
  // only (a third to a half of all 15-mers occur among 3 Gbases of reads with 10 % errors: the fill's stores, in ascending
  // address order, are nearly a stream).
  ix.direct_built = false;
  const u64 direct_min = e.opt.index_direct_min_keys > 0 ? static_cast<u64>(e.opt.index_direct_min_keys) : kDirectMinKeys;
  if (sizeof(V) == 4 && 2 * e.k <= 30 && u >= direct_min && !knob("RVN_NO_DIRECT_INDEX")) {
    const size_t n_dir = static_cast<size_t>(1) << (2 * e.k);
    u64* direct = nullptr;
    try {
      direct = ix.direct.get<u64>(n_dir);
    } catch (const DeviceOutOfMemory&) {  // (no room for it: the bucket table alone serves every probe, only slower)
      direct = nullptr;
    }
    if (direct) {
      RVN_HIP(hipMemsetAsync(direct, 0, n_dir * 8, s));
      RVN_KLAUNCH(kKTable, direct_fill_kernel<V><<<div_up(u, 256), 256, 0, s>>>(u_val, ix.u_start.as<u32>(), u, direct));
      ix.direct_built = true;
    }
  }
  t.stop();
}

template <typename V>
void index_build_impl(Engine& e, Sketch& sk, bool build_table) {
  hipStream_t s = e.stream;
  Index& ix = e.index;
  const u64 m = sk.count;
  ix.m = m;
  ix.u = 0;
  ix.occurrence = 0xFFFFFFFFu;
  if (m >= (1ULL << 32)) throw HipError("[raven_hip] index batch with >= 2^32 minimizers is not supported");

  // adopt the sketch buffers as ping side 0 (swap ownership, no copy)
  std::swap(ix.s_val[0].ptr, sk.val.ptr);
  std::swap(ix.s_val[0].cap, sk.val.cap);
  std::swap(ix.s_org[0].ptr, sk.org.ptr);
  std::swap(ix.s_org[0].cap, sk.org.cap);
  ix.cur = 0;
  if (m == 0) {
    ix.table_bits = 1;
    ix.shift = 2 * e.k > 1 ? 2 * e.k - 1 : 0;
    u32* table = ix.table.get<u32>(3);
    RVN_KLAUNCH(kKTable, table_empty_kernel<<<1, 64, 0, s>>>(table, 2));
    ix.u_val.reserve(16);
    ix.u_start.reserve(16);
    RVN_HIP(hipMemsetAsync(ix.u_start.ptr, 0, 8, s));
    ix.table_built = true;
    ix.direct_built = false;
    return;
  }
  V* v0 = ix.s_val[0].as<V>();
  u64* o0 = ix.s_org[0].as<u64>();
  V* v1 = ix.s_val[1].get<V>(m + 1);
  u64* o1 = ix.s_org[1].get<u64>(m + 1);
  {
    StageTimer t(e, StageTimes::kSort);
    if (sizeof(V) == 4)
      ix.cur = radix_sort_pairs_u32_u64(reinterpret_cast<u32*>(v0), reinterpret_cast<u32*>(v1), o0, o1, m, 2 * e.k,
                                        e.sort_tmp, e.scan_tmp, s, kKRsUpsweep, kKRsDownsweep, false);
    else
      ix.cur = radix_sort_pairs_u64_u64(reinterpret_cast<u64*>(v0), reinterpret_cast<u64*>(v1), o0, o1, m, 2 * e.k,
                                        e.sort_tmp, e.scan_tmp, s, kKRsUpsweep, kKRsDownsweep, false);
    t.stop();
  }
  ix.table_built = false;
  ix.direct_built = false;
  index_runs_impl<V>(e);
  if (build_table) index_table_impl<V>(e);
}

}  // namespace

void index_build(Engine& e, Sketch& sk, bool build_table) {
  e.index.first = sk.first;
  e.index.last = sk.last;
  if (e.val64) index_build_impl<u64>(e, sk, build_table);
  else index_build_impl<u32>(e, sk, build_table);
}

void index_build_table(Engine& e) {
  if (e.val64) index_table_impl<u64>(e);
  else index_table_impl<u32>(e);
}

// Count-of-counts histogram of the per-key run lengths (bins 0..65534; `over` = the run lengths >= 65535): what a
// rank contributes to the all-reduce behind the sharded pass's global Filter.
void index_key_histogram(Engine& e, std::vector<u64>& hist, std::vector<u32>& over) {
  Index& ix = e.index;
  hist.assign(kHistBins, 0);
  over.clear();
  if (ix.u == 0) return;
  hipStream_t s = e.stream;
  const u32 overflow_cap = 1u << 20;
  u32* d_hist = e.tmp_a.get<u32>(kHistBins + 1);
  u32* d_over = e.tmp_b.get<u32>(overflow_cap + 1);
  RVN_HIP(hipMemsetAsync(d_hist, 0, (kHistBins + 1) * 4, s));
  const u32 u = static_cast<u32>(ix.u);
  const u32 grid = std::min<u32>(div_up(u, 256), 2048);
  RVN_KLAUNCH(kKOccHist, occ_hist_kernel<<<grid, 256, 0, s>>>(ix.u_start.as<u32>(), u, d_hist, d_over, d_hist + kHistBins, overflow_cap,
                                                              ix.s_org[ix.cur].as<u64>()));
  std::vector<u32> h(kHistBins + 1);
  RVN_HIP(hipMemcpyAsync(h.data(), d_hist, (kHistBins + 1) * 4, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  const u32 n_over = h[kHistBins];
  if (n_over > overflow_cap) throw HipError("[raven_hip] key histogram: overflow list too small");
  for (u32 i = 0; i < kHistBins; ++i) hist[i] = h[i];
  hist[kHistBins - 1] = n_over;
  over.resize(n_over);
  if (n_over) RVN_HIP(hipMemcpy(over.data(), d_over, static_cast<size_t>(n_over) * 4, hipMemcpyDeviceToHost));
}

// ram Filter: occurrence_ = (value at index (1-f)*U of the sorted per-key counts) + 1; f == 0 -> no filter.
void index_filter(Engine& e, double freq) {
  Index& ix = e.index;
  if (freq == 0 || ix.u == 0) {
    ix.occurrence = 0xFFFFFFFFu;
    return;
  }
  StageTimer t(e, StageTimes::kFilter);
  hipStream_t s = e.stream;
  const u32 overflow_cap = 1u << 20;
  u32* hist = e.tmp_a.get<u32>(kHistBins + 1);
  u32* ovl = e.tmp_b.get<u32>(overflow_cap + 1);
  RVN_HIP(hipMemsetAsync(hist, 0, (kHistBins + 1) * 4, s));
  const u32 u = static_cast<u32>(ix.u);
  const u32 grid = std::min<u32>(div_up(u, 256), 2048);
  RVN_KLAUNCH(kKOccHist, occ_hist_kernel<<<grid, 256, 0, s>>>(ix.u_start.as<u32>(), u, hist, ovl, hist + kHistBins, overflow_cap, nullptr));
  size_t nth = static_cast<size_t>((1 - freq) * u);
  if (nth >= u) nth = u - 1;
  u64* qout = e.tmp_e.get<u64>(4);
  RVN_KLAUNCH(kKOccHist, occ_quantile_kernel<<<1, 256, 0, s>>>(hist, nth, qout));
  RVN_HIP(hipMemcpyAsync(e.h_pin, qout, 16, hipMemcpyDeviceToHost, s));
  RVN_HIP(hipMemcpyAsync(e.h_pin + 2, hist + kHistBins, 4, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  u32 c = static_cast<u32>(e.h_pin[0]);
  if (c >= kHistBins - 1) {
    // quantile lands among run lengths >= 65535: resolve exactly from the overflow list
    const u64 below = e.h_pin[1];
    const u32 n_over = static_cast<u32>(e.h_pin[2]);
    if (n_over > overflow_cap) throw HipError("[raven_hip] Filter: overflow list too small");
    std::vector<u32> over(n_over);
    RVN_HIP(hipMemcpy(over.data(), ovl, static_cast<size_t>(n_over) * 4, hipMemcpyDeviceToHost));
    std::sort(over.begin(), over.end());
    c = over[nth - below];
  }
  ix.occurrence = c + 1;
  t.stop();
}

}  // namespace rvn
