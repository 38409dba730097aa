// scan.hip — device-wide exclusive prefix sums (three-kernel reduce / scan / downsweep).
// HBM-bound: reads the input twice, writes the output once.
#include "common.h"
#include "wave.h"

namespace rvn {

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;  // per thread
constexpr int kScanTile = kScanThreads * kScanItems;

template <typename In>
__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(const In* __restrict__ in, u64 n,
                                                                  u64* __restrict__ block_sums);

// Single block: in-place exclusive scan of block sums; total written to sums[nb].
__global__ __launch_bounds__(kScanThreads) void scan_block_sums_kernel(u64* __restrict__ sums, u32 nb) {
  __shared__ u64 smem[4];
  __shared__ u64 carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (u32 start = 0; start < nb; start += kScanThreads) {
    u32 idx = start + threadIdx.x;
    u64 v = idx < nb ? sums[idx] : 0;
    u64 total;
    u64 ex = block_exclusive_sum_256<u64>(v, smem, &total);
    u64 carry = carry_s;
    if (idx < nb) sums[idx] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[nb] = carry_s;
}

// 16 consecutive items of one thread, vectorised (16-byte accesses) when the chunk is in range and aligned.
template <typename T>
__device__ __forceinline__ void load16(const T* __restrict__ in, u64 base, u64 n, u64 (&vals)[kScanItems]) {
  constexpr int kVec = 16 / sizeof(T);  // items per 16-byte access
  const bool fast = base + kScanItems <= n && (reinterpret_cast<uintptr_t>(in + base) & 15) == 0;
  if (fast) {
    const uint4* src = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
    for (int v = 0; v < kScanItems / kVec; ++v) {
      const uint4 q = src[v];
      const T* e = reinterpret_cast<const T*>(&q);
#pragma unroll
      for (int j = 0; j < kVec; ++j) vals[v * kVec + j] = static_cast<u64>(e[j]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) vals[i] = (base + i < n) ? static_cast<u64>(in[base + i]) : 0;
  }
}

template <typename T>
__device__ __forceinline__ void store16(T* __restrict__ out, u64 base, u64 n, const u64 (&vals)[kScanItems]) {
  constexpr int kVec = 16 / sizeof(T);
  const bool fast = base + kScanItems <= n && (reinterpret_cast<uintptr_t>(out + base) & 15) == 0;
  if (fast) {
    uint4* dst = reinterpret_cast<uint4*>(out + base);
#pragma unroll
    for (int v = 0; v < kScanItems / kVec; ++v) {
      uint4 q;
      T* e = reinterpret_cast<T*>(&q);
#pragma unroll
      for (int j = 0; j < kVec; ++j) e[j] = static_cast<T>(vals[v * kVec + j]);
      dst[v] = q;
    }
  } else {
#pragma unroll
    for (int i = 0; i < kScanItems; ++i)
      if (base + i < n) out[base + i] = static_cast<T>(vals[i]);
  }
}

template <typename In>
__global__ __launch_bounds__(kScanThreads) void scan_reduce_kernel(const In* __restrict__ in, u64 n,
                                                                  u64* __restrict__ block_sums) {
  __shared__ u64 smem[4];
  const u64 base = static_cast<u64>(blockIdx.x) * kScanTile + static_cast<u64>(threadIdx.x) * kScanItems;
  u64 vals[kScanItems];
  load16<In>(in, base, n, vals);
  u64 sum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) sum += vals[i];
  sum = wave_sum(sum);
  if (lane_id() == 0) smem[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = smem[0] + smem[1] + smem[2] + smem[3];
}

template <typename In, typename Out>
__global__ __launch_bounds__(kScanThreads) void scan_downsweep_kernel(const In* __restrict__ in, Out* __restrict__ out,
                                                                     u64 n, const u64* __restrict__ block_sums,
                                                                     u32 nb) {
  __shared__ u64 smem[4];
  // blocked arrangement: thread t owns items [t*16, t*16+16) of the tile
  const u64 base = static_cast<u64>(blockIdx.x) * kScanTile + static_cast<u64>(threadIdx.x) * kScanItems;
  u64 vals[kScanItems];
  load16<In>(in, base, n, vals);
  u64 sum = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) sum += vals[i];
  u64 total;
  u64 ex = block_exclusive_sum_256<u64>(sum, smem, &total);
  u64 run = block_sums[blockIdx.x] + ex;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    const u64 v = vals[i];
    vals[i] = run;
    run += v;
  }
  store16<Out>(out, base, n, vals);
  if (blockIdx.x == nb - 1 && threadIdx.x == 0) out[n] = static_cast<Out>(block_sums[nb]);
}

template <typename Out>
__global__ void scan_empty_kernel(Out* out) { out[0] = 0; }

template <typename In, typename Out>
void exclusive_scan_impl(const In* in, Out* out, u64 n, DevBuf& tmp, hipStream_t s) {
  if (n == 0) {
    RVN_KLAUNCH(kKScan, scan_empty_kernel<Out><<<1, 1, 0, s>>>(out));
    return;
  }
  u32 nb = div_up(n, kScanTile);
  u64* sums = tmp.get<u64>(static_cast<size_t>(nb) + 1);
  RVN_KLAUNCH(kKScan, scan_reduce_kernel<In><<<nb, kScanThreads, 0, s>>>(in, n, sums);
              scan_block_sums_kernel<<<1, kScanThreads, 0, s>>>(sums, nb);
              scan_downsweep_kernel<In, Out><<<nb, kScanThreads, 0, s>>>(in, out, n, sums, nb));
}

}  // namespace

const char* const kKernelSiteNames[kKNumSites] = {
    "sketch_count", "sketch_write", "minhash_select", "compact_sketch", "scan", "rs_bits", "rs_upsweep",
    "rs_downsweep", "heads", "unique", "table", "occ_hist", "match_count", "match_emit", "seg_sort_group",
    "intervals", "intervals_gather", "seg_sort_pos", "chain", "compact_overlaps", "pile_keys", "pile_counts",
    "pile_build", "add_layers", "truncate_sort", "kept_write", "gather", "pile_sort_up", "pile_sort_down", "chain_small", "join_count", "join_emit", "edit_banded", "edit_full", "poa", "add_kmers", "poa_banded", "pile_trim", "nw_forward", "nw_traceback", "edit_lane", "nw_lane", "best_overlap", "layer_build", "stitch", "poa_rows"};

thread_local KernelTimers* g_kernel_timers = nullptr;

KernelTimers::~KernelTimers() {
  for (auto ev : pool) (void)hipEventDestroy(ev);
}
size_t KernelTimers::next_event() {
  if (used == pool.size()) {
    hipEvent_t ev;
    RVN_HIP(hipEventCreate(&ev));
    pool.push_back(ev);
  }
  return used++;
}
void KernelTimers::resolve() {
  for (const auto& r : recs) {
    float t = 0;
    if (hipEventElapsedTime(&t, pool[r.e0], pool[r.e1]) == hipSuccess) {
      ms[r.site] += t;
      launches[r.site] += 1;
    }
  }
  recs.clear();
  used = 0;
}
void KernelTimers::reset() {
  recs.clear();
  used = 0;
  for (int i = 0; i < kKNumSites; ++i) {
    ms[i] = 0;
    launches[i] = 0;
  }
}

void exclusive_scan_u32_u64(const u32* in, u64* out, u64 n, DevBuf& tmp, hipStream_t s) {
  exclusive_scan_impl<u32, u64>(in, out, n, tmp, s);
}
void exclusive_scan_u32_u32(const u32* in, u32* out, u64 n, DevBuf& tmp, hipStream_t s) {
  exclusive_scan_impl<u32, u32>(in, out, n, tmp, s);
}
void exclusive_scan_u8_u32(const u8* in, u32* out, u64 n, DevBuf& tmp, hipStream_t s) {
  exclusive_scan_impl<u8, u32>(in, out, n, tmp, s);
}

}  // namespace rvn
