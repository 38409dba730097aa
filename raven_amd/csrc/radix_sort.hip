// radix_sort.hip — stable device-wide LSD radix sort of (key, value) pairs, 8 bits per pass.
//
// Per pass: upsweep (per-tile digit histogram, reads keys only) -> scan of the
// digit-major histogram -> downsweep (re-read tile, stable in-tile rank by
// wave ballots, scatter).  Each 256-thread block owns a 4096-item tile; wave w
// owns 1024 consecutive items as 16 rows of 64 (lane = item within row), so a
// row is one coalesced wave load and item order == (wave, row, lane).
// Passes whose digit is constant over the whole input are skipped.
// HBM-bound: 4|8 B (upsweep) + 2 x (key + value) bytes per item per pass.
#include "common.h"
#include "wave.h"

namespace rvn {

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kRows = 16;
constexpr int kWaveItems = kRows * 64;    // 1024
constexpr int kTile = kWaves * kWaveItems;  // 4096

template <typename K>
__global__ __launch_bounds__(kThreads) void rs_upsweep_kernel(const K* __restrict__ keys, u64 n, int shift,
                                                             u32* __restrict__ block_hist, u32 nb) {
  __shared__ u32 hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const u64 base = static_cast<u64>(blockIdx.x) * kTile;
#pragma unroll
  for (int i = 0; i < kTile / kThreads; ++i) {
    u64 idx = base + static_cast<u64>(i) * kThreads + threadIdx.x;
    if (idx < n) atomicAdd(&hist[(keys[idx] >> shift) & 0xFF], 1u);
  }
  __syncthreads();
  block_hist[static_cast<u64>(threadIdx.x) * nb + blockIdx.x] = hist[threadIdx.x];
}

// Downsweep with an LDS-staged scatter: items are first placed in tile-sorted order in LDS, then written
// out so that consecutive lanes store consecutive addresses of the same digit run (the direct scatter wrote
// 4-8 byte fragments: PMC WRITE_SIZE was 2.1x the algorithmic bytes, profiles/r01_c_pmc_write_size.csv).
template <typename K, typename V>
__global__ __launch_bounds__(kThreads) void rs_downsweep_kernel(const K* __restrict__ keys_in,
                                                               const V* __restrict__ vals_in,
                                                               K* __restrict__ keys_out, V* __restrict__ vals_out,
                                                               u64 n, int shift,
                                                               const u32* __restrict__ block_hist_scanned, u32 nb) {
  __shared__ K s_keys[kTile];
  __shared__ V s_vals[kTile];
  __shared__ u16 wave_cnt[kWaves][256];  // per-wave digit counts, then per-wave tile-local offsets
  __shared__ u32 digit_gbase[256];       // global base of (digit, block) minus the digit's tile-local start
  __shared__ u32 smem4[4];
  const int w = threadIdx.x >> 6;
  const int lane = lane_id();
  for (int i = threadIdx.x; i < kWaves * 256; i += kThreads) (&wave_cnt[0][0])[i] = 0;
  __syncthreads();

  const u64 tile_base = static_cast<u64>(blockIdx.x) * kTile;
  const u64 wbase = tile_base + static_cast<u64>(w) * kWaveItems;
  K key[kRows];
  V val[kRows];
  u16 rank[kRows];
  const unsigned long long lt = lanemask_lt();
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const u64 idx = wbase + static_cast<u64>(r) * 64 + lane;
    const bool valid = idx < n;
    if (valid) {
      key[r] = keys_in[idx];
      val[r] = vals_in[idx];
    } else {
      key[r] = 0;
      val[r] = V{};
    }
  }
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const u64 idx = wbase + static_cast<u64>(r) * 64 + lane;
    const bool valid = idx < n;
    const unsigned d = static_cast<unsigned>((key[r] >> shift) & 0xFF);
    const unsigned long long peers = match_digit8(d, valid);
    u32 before = 0;
    if (valid) before = wave_cnt[w][d];
    __builtin_amdgcn_wave_barrier();
    rank[r] = static_cast<u16>(before + __popcll(peers & lt));
    if (valid && (peers & lt) == 0) wave_cnt[w][d] = static_cast<u16>(before + __popcll(peers));
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {
    // digit t: counts per wave -> tile-local start of the digit (block scan) and per-wave offsets
    const int t = threadIdx.x;
    u32 c[kWaves];
    u32 tot = 0;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) {
      c[i] = wave_cnt[i][t];
      tot += c[i];
    }
    u32 total;
    const u32 start = block_exclusive_sum_256<u32>(tot, smem4, &total);
    digit_gbase[t] = block_hist_scanned[static_cast<u64>(t) * nb + blockIdx.x] - start;
    u32 run = start;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) {
      wave_cnt[i][t] = static_cast<u16>(run);
      run += c[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const u64 idx = wbase + static_cast<u64>(r) * 64 + lane;
    if (idx < n) {
      const unsigned d = static_cast<unsigned>((key[r] >> shift) & 0xFF);
      const u32 lp = static_cast<u32>(wave_cnt[w][d]) + rank[r];
      s_keys[lp] = key[r];
      s_vals[lp] = val[r];
    }
  }
  __syncthreads();
  const u32 valid_n = static_cast<u32>(n - tile_base < static_cast<u64>(kTile) ? n - tile_base : kTile);
#pragma unroll
  for (int i = 0; i < kTile / kThreads; ++i) {
    const u32 lp = i * kThreads + threadIdx.x;
    if (lp < valid_n) {
      const K kk = s_keys[lp];
      const unsigned d = static_cast<unsigned>((kk >> shift) & 0xFF);
      const u32 dst = digit_gbase[d] + lp;
      keys_out[dst] = kk;
      vals_out[dst] = s_vals[lp];
    }
  }
}

// OR / AND reduction of all keys, to skip constant digits.
template <typename K>
__global__ __launch_bounds__(kThreads) void rs_bits_kernel(const K* __restrict__ keys, u64 n, u64* __restrict__ out2) {
  u64 o = 0, a = ~0ULL;
  for (u64 idx = static_cast<u64>(blockIdx.x) * kThreads + threadIdx.x; idx < n;
       idx += static_cast<u64>(gridDim.x) * kThreads) {
    u64 k = keys[idx];
    o |= k;
    a &= k;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    o |= __shfl_xor(o, off, 64);
    a &= __shfl_xor(a, off, 64);
  }
  if (lane_id() == 0) {
    atomicOr(reinterpret_cast<unsigned long long*>(&out2[0]), static_cast<unsigned long long>(o));
    atomicAnd(reinterpret_cast<unsigned long long*>(&out2[1]), static_cast<unsigned long long>(a));
  }
}

__global__ void rs_bits_init_kernel(u64* out2) {
  out2[0] = 0;
  out2[1] = ~0ULL;
}

template <typename K, typename V>
int radix_sort_impl(K* k0, K* k1, V* v0, V* v1, u64 n, int key_bits, DevBuf& tmp, DevBuf& tmp2, hipStream_t s,
                    int site_up, int site_down, bool skip_constant_digits) {
  if (n <= 1 || key_bits <= 0) return 0;
  if (n >= (1ULL << 32)) throw HipError("[raven_hip] radix_sort: n >= 2^32 not supported");
  u32 nb = div_up(n, kTile);
  size_t hist_entries = static_cast<size_t>(nb) * 256;
  // tmp layout: [bits: 2 x u64][hist: hist_entries u32][hist_scanned: hist_entries+1 u32]; tmp2 = scan scratch
  size_t off_hist = 16;
  size_t off_scanned = off_hist + hist_entries * 4;
  size_t off_end = off_scanned + (hist_entries + 1) * 4;
  tmp.reserve(off_end + 256);
  char* base = tmp.as<char>();
  u64* bits = reinterpret_cast<u64*>(base);
  u32* hist = reinterpret_cast<u32*>(base + off_hist);
  u32* scanned = reinterpret_cast<u32*>(base + off_scanned);

  u64 varying = ~0ULL;
  if (skip_constant_digits) {
  RVN_KLAUNCH(kKRsBits, rs_bits_init_kernel<<<1, 1, 0, s>>>(bits));
  u32 gb = nb < 2048 ? (nb * 16 < 1 ? 1 : (nb * 16 > 2048 ? 2048 : nb * 16)) : 2048;
  RVN_KLAUNCH(kKRsBits, rs_bits_kernel<K><<<gb, kThreads, 0, s>>>(k0, n, bits));
  u64 hbits[2];
  RVN_HIP(hipMemcpyAsync(hbits, bits, 16, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  varying = hbits[0] ^ hbits[1];
  }

  int cur = 0;
  for (int shift = 0; shift < key_bits; shift += 8) {
    if (((varying >> shift) & 0xFF) == 0) continue;
    K* kin = cur ? k1 : k0;
    K* kout = cur ? k0 : k1;
    V* vin = cur ? v1 : v0;
    V* vout = cur ? v0 : v1;
    RVN_KLAUNCH(site_up, rs_upsweep_kernel<K><<<nb, kThreads, 0, s>>>(kin, n, shift, hist, nb));
    exclusive_scan_u32_u32(hist, scanned, hist_entries, tmp2, s);
    RVN_KLAUNCH(site_down,
                rs_downsweep_kernel<K, V><<<nb, kThreads, 0, s>>>(kin, vin, kout, vout, n, shift, scanned, nb));
    cur ^= 1;
  }
  return cur;
}

}  // namespace

int radix_sort_pairs_u32_u64(u32* k0, u32* k1, u64* v0, u64* v1, u64 n, int key_bits, DevBuf& tmp, DevBuf& tmp2, hipStream_t s,
                    int site_up, int site_down, bool skip_constant_digits) {
  return radix_sort_impl<u32, u64>(k0, k1, v0, v1, n, key_bits, tmp, tmp2, s, site_up, site_down, skip_constant_digits);
}
int radix_sort_pairs_u64_u64(u64* k0, u64* k1, u64* v0, u64* v1, u64 n, int key_bits, DevBuf& tmp, DevBuf& tmp2, hipStream_t s,
                    int site_up, int site_down, bool skip_constant_digits) {
  return radix_sort_impl<u64, u64>(k0, k1, v0, v1, n, key_bits, tmp, tmp2, s, site_up, site_down, skip_constant_digits);
}
int radix_sort_pairs_u32_u32(u32* k0, u32* k1, u32* v0, u32* v1, u64 n, int key_bits, DevBuf& tmp, DevBuf& tmp2, hipStream_t s,
                    int site_up, int site_down, bool skip_constant_digits) {
  return radix_sort_impl<u32, u32>(k0, k1, v0, v1, n, key_bits, tmp, tmp2, s, site_up, site_down, skip_constant_digits);
}

}  // namespace rvn
