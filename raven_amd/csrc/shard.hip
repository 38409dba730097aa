// shard.hip — partition / regroup steps of the sharded single-genome pass (raven_amd/sharded.py): the data
// movement between the stages of raven::FindOverlapsAndCreatePiles (RavenLib/src/construct.cc:14-121) when minimizers
// are owned by hash class and reads / piles by contiguous read range.
//   * stable multi-split by owner rank (minimizers by hash class before exchange 1; overlaps by the owner of their rhs
//     read before exchange 3): per-tile bucket counts, one scan, stable scatter — the order inside a bucket is the
//     input order, which is what keeps ram's (read, position) order and the reference's overlap order;
//   * regroup of the matches received from every index owner into per-read segments (exchange 2);
//   * per-read offsets of a list of overlaps grouped by lhs read (input of the pile merge).
// All buffers are device pointers (the exchange buffers are allocated by the host side and handed to RCCL as they are).
// HBM-bound byte movement: every element is read twice and written once.
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "engine.h"
#include "wave.h"

namespace rvn {

namespace {

constexpr int kSplitThreads = 256;
constexpr int kSplitItems = 8;
constexpr int kSplitTile = kSplitThreads * kSplitItems;
constexpr int kSplitMaxBuckets = 17;  // up to 16 ranks + the bucket of what stays behind

// owner of a minimizer: its hash class (raven_amd/sharded.py hash_owner)
struct HashOwnerKey {
  const u64* val;
  u32 world;
  __device__ u32 operator()(u64 i) const {
    if (world == 1) return 0;
    const u64 h = (val[i] * 0x9E3779B97F4A7C15ULL) >> 33;
    return static_cast<u32>(h % world);
  }
};

// owner of an overlap: the rank whose read range holds its rhs read; overlaps of `self` stay (bucket `world`)
struct RhsOwnerKey {
  const Overlap* ovl;
  u32 bounds[kSplitMaxBuckets + 1];
  u32 world, self;
  __device__ u32 operator()(u64 i) const {
    const u32 rhs = ovl[i].rhs_id;
    u32 o = 0;
    while (o + 1 < world && rhs >= bounds[o + 1]) ++o;
    return o == self ? world : o;
  }
};

template <class Key>
__global__ __launch_bounds__(kSplitThreads) void split_count_kernel(Key key, u64 n, u32 nb, u32* __restrict__ block_hist,
                                                                     u32 n_blocks) {
  __shared__ u32 s_cnt[kSplitMaxBuckets];
  if (threadIdx.x < nb) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const u64 base = static_cast<u64>(blockIdx.x) * kSplitTile + static_cast<u64>(threadIdx.x) * kSplitItems;
  for (int k = 0; k < kSplitItems; ++k)
    if (base + k < n) atomicAdd(&s_cnt[key(base + k)], 1u);
  __syncthreads();
  if (threadIdx.x < nb) block_hist[static_cast<size_t>(threadIdx.x) * n_blocks + blockIdx.x] = s_cnt[threadIdx.x];
}

// Element i of column c is ROW consecutive 64-bit words at src[c] + i * ROW.
template <class Key, int NCOL, int ROW>
__global__ __launch_bounds__(kSplitThreads) void split_scatter_kernel(Key key, const u64* const* __restrict__ src_cols,
                                                                       u64* const* __restrict__ dst_cols, u64 n, u32 nb,
                                                                       const u64* __restrict__ block_off, u32 n_blocks) {
  __shared__ u16 s_pre[kSplitMaxBuckets][kSplitThreads];
  const u32 tid = threadIdx.x;
  for (u32 b = 0; b < nb; ++b) s_pre[b][tid] = 0;
  const u64 base = static_cast<u64>(blockIdx.x) * kSplitTile + static_cast<u64>(tid) * kSplitItems;
  u32 keys[kSplitItems];
#pragma unroll
  for (int k = 0; k < kSplitItems; ++k) {
    keys[k] = 0xFFFFFFFFu;
    if (base + k < n) {
      keys[k] = key(base + k);
      s_pre[keys[k]][tid] += 1;  // column `tid` belongs to this thread
    }
  }
  __syncthreads();
  {  // exclusive prefix over the threads, per bucket; wave w takes buckets w, w + 4, ...
    const int lane = lane_id();
    for (u32 b = tid >> 6; b < nb; b += kSplitThreads / 64) {
      u32 carry = 0;
      for (int c = 0; c < kSplitThreads / 64; ++c) {
        const u32 v = s_pre[b][c * 64 + lane];
        const u32 inc = wave_inclusive_sum(v);
        s_pre[b][c * 64 + lane] = static_cast<u16>(carry + inc - v);
        carry += __shfl(inc, 63, 64);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kSplitItems; ++k) {
    if (keys[k] == 0xFFFFFFFFu) continue;
    const u32 b = keys[k];
    const u64 dest = block_off[static_cast<size_t>(b) * n_blocks + blockIdx.x] + s_pre[b][tid];
    s_pre[b][tid] += 1;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      const u64* s = src_cols[c] + (base + k) * ROW;
      u64* d = dst_cols[c] + dest * ROW;
#pragma unroll
      for (int r = 0; r < ROW; ++r) d[r] = s[r];
    }
  }
}

__global__ void gather_bucket_starts_kernel(const u64* __restrict__ block_off, u32 n_blocks, u32 nb, u64* __restrict__ out) {
  const u32 b = threadIdx.x;
  if (b <= nb) out[b] = block_off[static_cast<size_t>(b) * n_blocks];  // b == nb: the total (scan output has n + 1 entries)
}

template <class Key, int NCOL, int ROW>
void multi_split(Engine& e, Key key, const u64* const* src_cols, u64* const* dst_cols, u64 n, u32 nb, u64* counts) {
  hipStream_t s = e.stream;
  for (u32 b = 0; b < nb; ++b) counts[b] = 0;
  if (n == 0) return;
  if (nb > kSplitMaxBuckets) throw std::invalid_argument("[raven_hip] sharded pass: at most 16 ranks");
  const u32 n_blocks = div_up(n, kSplitTile);
  const size_t cells = static_cast<size_t>(nb) * n_blocks;
  u32* d_hist = e.sh_hist.get<u32>(cells + 1);
  u64* d_off = e.sh_off.get<u64>(cells + 2);
  split_count_kernel<Key><<<n_blocks, kSplitThreads, 0, s>>>(key, n, nb, d_hist, n_blocks);
  RVN_LAUNCH_CHECK();
  exclusive_scan_u32_u64(d_hist, d_off, cells, e.scan_tmp, s);
  // device copies of the column pointer tables
  const u64** d_src = reinterpret_cast<const u64**>(e.sh_ptrs.get<u64>(2 * NCOL + 2 + kSplitMaxBuckets + 2));
  u64** d_dst = const_cast<u64**>(d_src) + NCOL;
  u64* d_starts = reinterpret_cast<u64*>(const_cast<u64**>(d_src) + 2 * NCOL);
  RVN_HIP(hipMemcpyAsync(d_src, src_cols, NCOL * sizeof(u64*), hipMemcpyHostToDevice, s));
  RVN_HIP(hipMemcpyAsync(d_dst, dst_cols, NCOL * sizeof(u64*), hipMemcpyHostToDevice, s));
  split_scatter_kernel<Key, NCOL, ROW><<<n_blocks, kSplitThreads, 0, s>>>(key, d_src, d_dst, n, nb, d_off, n_blocks);
  RVN_LAUNCH_CHECK();
  gather_bucket_starts_kernel<<<1, 64, 0, s>>>(d_off, n_blocks, nb, d_starts);
  RVN_LAUNCH_CHECK();
  RVN_HIP(hipMemcpyAsync(e.h_pin, d_starts, (nb + 1) * 8, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  for (u32 b = 0; b < nb; ++b) counts[b] = e.h_pin[b + 1] - e.h_pin[b];
}

__global__ void count_flagged_kernel(const u64* __restrict__ org, u64 n, unsigned long long* __restrict__ out) {
  u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  const u64 stride = static_cast<u64>(gridDim.x) * blockDim.x;
  u32 c = 0;
  for (; i < n; i += stride) c += static_cast<u32>(org[i] >> 63);
  c = wave_sum(c);
  if (lane_id() == 0 && c) atomicAdd(out, static_cast<unsigned long long>(c));
}

__global__ void adjacent_diff_kernel(const u64* __restrict__ seg, u64 n, u64* __restrict__ cnt) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) cnt[i] = seg[i + 1] - seg[i];
}

__global__ void sum_counts_kernel(const u64* const* __restrict__ cnts, u32 world, u32 n, u32* __restrict__ total) {
  const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  u64 t = 0;
  for (u32 s = 0; s < world; ++s) t += cnts[s][r];
  total[r] = static_cast<u32>(t);
}

__global__ void narrow_u64_u32_kernel(const u64* __restrict__ src, u32 n, u32* __restrict__ dst) {
  const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) dst[r] = static_cast<u32>(src[r]);
}

// element i of source s belongs to the read r with src_off[r] <= i < src_off[r + 1]; it goes to start[r] + (i - src_off[r])
__global__ void regroup_scatter_kernel(const u64* __restrict__ src_off, const u64* __restrict__ start, u32 n_reads, u64 m,
                                       const u64* __restrict__ grp, const u64* __restrict__ pos, u64* __restrict__ grp_out,
                                       u64* __restrict__ pos_out) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m) return;
  u32 lo = 0, hi = n_reads;  // last r with src_off[r] <= i
  while (hi - lo > 1) {
    const u32 mid = lo + (hi - lo) / 2;
    if (src_off[mid] <= i) lo = mid;
    else hi = mid;
  }
  const u64 dest = start[lo] + (i - src_off[lo]);
  grp_out[dest] = grp[i];
  pos_out[dest] = pos[i];
}

__global__ void advance_start_kernel(u64* __restrict__ start, const u64* __restrict__ cnt, u32 n) {
  const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) start[r] += cnt[r];
}

__global__ void lhs_count_kernel(const Overlap* __restrict__ ovl, u64 n, u32 n_reads, u32* __restrict__ cnt) {
  const u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n && ovl[i].lhs_id < n_reads) atomicAdd(&cnt[ovl[i].lhs_id], 1u);
}

}  // namespace

void shard_split_minimizers(Engine& e, const u64* d_val, const u64* d_org, u64 n, u32 world, u64* d_val_out, u64* d_org_out,
                            u64* counts) {
  HashOwnerKey key{d_val, world};
  const u64* src[2] = {d_val, d_org};
  u64* dst[2] = {d_val_out, d_org_out};
  multi_split<HashOwnerKey, 2, 1>(e, key, src, dst, n, world, counts);
}

void shard_split_overlaps(Engine& e, const Overlap* d_ovl, u64 n, const u32* bounds, u32 world, u32 self, Overlap* d_out,
                          u64* counts) {
  static_assert(sizeof(Overlap) == 32, "an overlap is four 64-bit words");
  if (world > 16) throw std::invalid_argument("[raven_hip] sharded pass: at most 16 ranks");
  RhsOwnerKey key{};
  key.ovl = d_ovl;
  for (u32 i = 0; i <= world; ++i) key.bounds[i] = bounds[i];
  key.world = world;
  key.self = self;
  const u64* src[1] = {reinterpret_cast<const u64*>(d_ovl)};
  u64* dst[1] = {reinterpret_cast<u64*>(d_out)};
  multi_split<RhsOwnerKey, 1, 4>(e, key, src, dst, n, world + 1, counts);
}

u64 shard_count_flagged(Engine& e, const u64* d_org, u64 n) {
  if (n == 0) return 0;
  unsigned long long* d = e.sh_ptrs.get<unsigned long long>(64);
  RVN_HIP(hipMemsetAsync(d, 0, 8, e.stream));
  count_flagged_kernel<<<std::min<u32>(div_up(n, 256), 2048), 256, 0, e.stream>>>(d_org, n, d);
  RVN_LAUNCH_CHECK();
  return read_back(e, d, 8);
}

void shard_adjacent_diff(Engine& e, const u64* d_seg, u64 n, u64* d_cnt) {
  if (n == 0) return;
  adjacent_diff_kernel<<<div_up(n, 256), 256, 0, e.stream>>>(d_seg, n, d_cnt);
  RVN_LAUNCH_CHECK();
}

// Matches received from `world` index owners: source s delivers, for every own read in order, cnt[s][r] entries of
// (group, position).  Output: per read the sources' segments in rank order (the order ram's per-read match list has when
// one index holds every hash class is restored by the chain stage's sort); seg[n_reads + 1] = per-read offsets.
void shard_regroup(Engine& e, u32 world, const u64* const* d_cnt, const u64* const* d_grp, const u64* const* d_pos,
                   const u64* n_src, u32 n_reads, u64* d_seg, u64* d_grp_out, u64* d_pos_out) {
  hipStream_t s = e.stream;
  if (n_reads == 0) {
    RVN_HIP(hipMemsetAsync(d_seg, 0, 8, s));
    return;
  }
  const u64** d_tab = reinterpret_cast<const u64**>(e.sh_ptrs.get<u64>(world + 64));
  RVN_HIP(hipMemcpyAsync(d_tab, d_cnt, world * sizeof(u64*), hipMemcpyHostToDevice, s));
  u32* d_total = e.sh_hist.get<u32>(static_cast<size_t>(n_reads) + 1);
  sum_counts_kernel<<<div_up(n_reads, 256), 256, 0, s>>>(d_tab, world, n_reads, d_total);
  RVN_LAUNCH_CHECK();
  exclusive_scan_u32_u64(d_total, d_seg, n_reads, e.scan_tmp, s);
  u64* d_start = e.sh_off.get<u64>(2 * static_cast<size_t>(n_reads) + 4);
  u64* d_src_off = d_start + n_reads + 1;
  RVN_HIP(hipMemcpyAsync(d_start, d_seg, static_cast<size_t>(n_reads) * 8, hipMemcpyDeviceToDevice, s));
  for (u32 src = 0; src < world; ++src) {
    if (n_src[src]) {
      narrow_u64_u32_kernel<<<div_up(n_reads, 256), 256, 0, s>>>(d_cnt[src], n_reads, d_total);
      RVN_LAUNCH_CHECK();
      exclusive_scan_u32_u64(d_total, d_src_off, n_reads, e.scan_tmp, s);
      regroup_scatter_kernel<<<div_up(n_src[src], 256), 256, 0, s>>>(d_src_off, d_start, n_reads, n_src[src], d_grp[src],
                                                                      d_pos[src], d_grp_out, d_pos_out);
      RVN_LAUNCH_CHECK();
    }
    advance_start_kernel<<<div_up(n_reads, 256), 256, 0, s>>>(d_start, d_cnt[src], n_reads);
    RVN_LAUNCH_CHECK();
  }
}

// d_off[r] = first overlap of lhs read r in a list grouped by ascending lhs id (n_reads + 1 entries)
void shard_lhs_offsets(Engine& e, const Overlap* d_ovl, u64 n, u32 n_reads, u32* d_off) {
  hipStream_t s = e.stream;
  u32* d_cnt = e.sh_hist.get<u32>(static_cast<size_t>(n_reads) + 1);
  RVN_HIP(hipMemsetAsync(d_cnt, 0, (static_cast<size_t>(n_reads) + 1) * 4, s));
  if (n) {
    lhs_count_kernel<<<div_up(n, 256), 256, 0, s>>>(d_ovl, n, n_reads, d_cnt);
    RVN_LAUNCH_CHECK();
  }
  exclusive_scan_u32_u32(d_cnt, d_off, n_reads, e.scan_tmp, s);
}

}  // namespace rvn
