// map.hip — batched ram::MinimizerEngine::Map (call site RavenLib/src/construct.cc:59-64, :377-381):
// query sketch -> index probe -> match emit -> sort by group -> diagonal-band intervals ->
// per-interval sort by positions -> LIS chain -> overlaps, for a whole batch of query reads at once.
// Output order equals the reference's "for each query read in order, Map() output order".
//
// Kernels (all integer, HBM/L2-bound; no MFMA):
//   match_count / match_emit   one lane per query minimizer, direct-address table probe
//   seg_sort                   one wave per segment, LSD radix on 64-bit keys, LDS histogram,
//                              stable rank by wave ballots (used for both sorts)
//   intervals                  one wave per read segment; ram's sequential (i, j) sweep restated as
//                              a per-element binary search + wave max-scan
//   chain                      one lane per interval: ram's exact patience/binary-search LIS, gap split,
//                              covered-bases score
#include <algorithm>

#include <cstring>

#include "engine.h"
#include "wave.h"

namespace rvn {

namespace {

template <typename V>
__device__ __forceinline__ bool index_find(const V* __restrict__ u_val, const u32* __restrict__ u_start,
                                           const u32* __restrict__ table, int shift, V v, u32* start, u32* count) {
  const u32 b = static_cast<u32>(static_cast<u64>(v) >> shift);
  u32 lo = table[b], hi = table[b + 1];
  const u32 end = hi;
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);
    if (u_val[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  if (lo < end && u_val[lo] == v) {
    *start = u_start[lo];
    *count = u_start[lo + 1] - *start;
    return true;
  }
  return false;
}

template <typename V>
__global__ void match_count_kernel(const V* __restrict__ q_val, const u64* __restrict__ q_org, u64 nq,
                                   const V* __restrict__ u_val, const u32* __restrict__ u_start,
                                   const u32* __restrict__ table, int shift, u32 n_keys,
                                   const u64* __restrict__ s_org, u32 occurrence, int avoid_equal,
                                   int avoid_symmetric, u32* __restrict__ q_start, u32* __restrict__ q_n,
                                   u32* __restrict__ q_cnt, u8* __restrict__ filtered, const u64* __restrict__ direct) {
  u64 q = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const V v = q_val[q];
  const u32 qid = origin_id(q_org[q]);
  u32 start = 0, count = 0, cnt = 0;
  u8 filt = 0;
  bool found;
  if (direct) {  // (index.hip: every value addressed directly — the same start / count the search below finds)
    const u64 ent = direct[static_cast<u64>(v)];
    count = static_cast<u32>(ent >> 32);
    found = count != 0;
    start = found ? static_cast<u32>(ent) : 0u;
  } else {
    found = n_keys && index_find<V>(u_val, u_start, table, shift, v, &start, &count);
  }
  if (found) {
    if (count > occurrence) {
      filt = 1;
      count = 0;
    } else {
      for (u32 j = 0; j < count; ++j) {
        const u32 rid = origin_id(s_org[start + j]);
        if (avoid_equal && qid == rid) continue;
        if (avoid_symmetric && qid > rid) continue;
        ++cnt;
      }
    }
  }
  q_start[q] = start;
  q_n[q] = count;
  q_cnt[q] = cnt;
  if (filtered) filtered[q] = filt;
}

__global__ void match_emit_kernel(const u64* __restrict__ q_org, u64 nq, const u64* __restrict__ s_org,
                                  const u32* __restrict__ q_start, const u32* __restrict__ q_n,
                                  const u64* __restrict__ m_off, int avoid_equal, int avoid_symmetric,
                                  u64* __restrict__ m_grp, u64* __restrict__ m_pos) {
  u64 q = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const u32 count = q_n[q];
  if (count == 0) return;
  const u64 qo = q_org[q];
  const u32 qid = origin_id(qo);
  const u64 lhs_pos = static_cast<u32>(qo) >> 1;
  const u32 qstrand = static_cast<u32>(qo) & 1u;
  const u32 start = q_start[q];
  u64 o = m_off[q];
  for (u32 j = 0; j < count; ++j) {
    const u64 ro = s_org[start + j];
    const u32 rid = origin_id(ro);
    if (avoid_equal && qid == rid) continue;
    if (avoid_symmetric && qid > rid) continue;
    const u64 rhs_pos = static_cast<u32>(ro) >> 1;
    const u64 strand = (qstrand == (static_cast<u32>(ro) & 1u)) ? 1 : 0;
    const u64 diagonal = !strand ? rhs_pos + lhs_pos : rhs_pos - lhs_pos + (3ULL << 30);
    m_grp[o] = (((static_cast<u64>(rid) << 1) | strand) << 32) | diagonal;
    m_pos[o] = (lhs_pos << 32) | rhs_pos;
    ++o;
  }
}

// ---- self-join over the sorted index ----------------------------------------------------------------
// When the query reads are exactly the indexed reads, every query minimizer is itself an index entry
// (flagged kQueryFlag), so Map's probes become one streaming pass over the runs of equal value: for each
// flagged entry of a run (count <= occurrence), every other entry of the run is a match.  No table, no random
// access (the probe path fetched ~3 GB of cache lines per pass, profiles/r01_c_pmc_fetch_size.csv).
// Matches of one read land in that read's segment in arbitrary order; the result does not depend on it
// because the following sorts are total orders (group, then positions).
template <bool EMIT>
__global__ __launch_bounds__(256) void join_kernel(const u32* __restrict__ u_start, u32 n_runs,
                                                  const u64* __restrict__ s_org, u32 occurrence, int all_query,
                                                  int avoid_equal, int avoid_symmetric, u32 first, u32 q_lo, u32 q_hi,
                                                  u32* __restrict__ read_cnt, const u64* __restrict__ seg_off,
                                                  u32* __restrict__ cursor, u64* __restrict__ m_grp,
                                                  u64* __restrict__ m_pos) {
  const u32 run = blockIdx.x * blockDim.x + threadIdx.x;
  if (run >= n_runs) return;
  const u32 s = u_start[run];
  const u32 c = u_start[run + 1] - s;
  // entries [0, f) of the run are queries of reads outside this index batch (kForeignFlag): no members of the index
  const u32 f = run_foreign_prefix(s_org, s, c);
  if (c - f > occurrence || c == f) return;
  if (c == 1 && avoid_equal) return;
  for (u32 i = 0; i < c; ++i) {
    const u64 qo = s_org[s + i];
    if (!all_query && !(qo & kQueryFlag)) continue;
    const u32 qid = origin_id(qo);
    if (qid < q_lo || qid >= q_hi) continue;  // query reads of another flush window (sharded pass)
    u32 cnt = 0;
    for (u32 j = f; j < c; ++j) {
      const u32 rid = origin_id(s_org[s + j]);
      if (avoid_equal && qid == rid) continue;
      if (avoid_symmetric && qid > rid) continue;
      ++cnt;
    }
    if (cnt == 0) continue;
    if (!EMIT) {
      atomicAdd(&read_cnt[qid - first], cnt);
    } else {
      u64 o = seg_off[qid - first] + atomicAdd(&cursor[qid - first], cnt);
      const u64 lhs_pos = static_cast<u32>(qo) >> 1;
      const u32 qstrand = static_cast<u32>(qo) & 1u;
      for (u32 j = f; j < c; ++j) {
        const u64 ro = s_org[s + j];
        const u32 rid = origin_id(ro);
        if (avoid_equal && qid == rid) continue;
        if (avoid_symmetric && qid > rid) continue;
        const u64 rhs_pos = static_cast<u32>(ro) >> 1;
        const u64 strand = (qstrand == (static_cast<u32>(ro) & 1u)) ? 1 : 0;
        const u64 diagonal = !strand ? rhs_pos + lhs_pos : rhs_pos - lhs_pos + (3ULL << 30);
        m_grp[o] = (((static_cast<u64>(rid) << 1) | strand) << 32) | diagonal;
        m_pos[o] = (lhs_pos << 32) | rhs_pos;
        ++o;
      }
    }
  }
}

__global__ void gather_u64_by_u32_kernel(const u64* __restrict__ src, const u32* __restrict__ idx,
                                         u64* __restrict__ dst, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

// ---- wave-level segmented LSD radix sort ---------------------------------------------------------
// Sorts keys k0[b, b+n) (payload p0 when PAY) stably by the full 64-bit key; result always ends in k0/p0.  For the segments
// that do not fit a workgroup's LDS (HiFi: 10 000 - 100 000 matches of one read).  hist: 2 x 256 counters of the wave.
// Round 6: a pass is ONE sweep over the segment — the counts of the next digit are taken while the current one is
// scattered (and those of byte 0 while the varying bits are found), where rounds 1-5 read the keys a second time per pass —
// and a sweep takes four batches of 64 at a time, all their loads issued before the first rank is computed (a wave's
// passes are chains of dependent global loads: the stage ran at 4 % of the HBM rate, bound by their latency).
template <bool PAY>
__device__ void wave_sort_segment(u64* __restrict__ k0, u64* __restrict__ k1, u64* __restrict__ p0,
                                  u64* __restrict__ p1, u64 b, u64 n, u32* hist) {
  const int lane = lane_id();
  const unsigned long long lt = lanemask_lt();
  u32* h_cur = hist;         // counts / running offsets of the digit being scattered
  u32* h_nxt = hist + 256;   // counts of the next digit
  for (int i = lane; i < 256; i += 64) h_nxt[i] = 0;
  __builtin_amdgcn_wave_barrier();
  u64 o = 0, a = ~0ULL;
  for (u64 i = lane; i < n; i += 64) {
    const u64 key = k0[b + i];
    o |= key;
    a &= key;
    atomicAdd(&h_nxt[key & 0xFF], 1u);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    o |= __shfl_xor(o, off, 64);
    a &= __shfl_xor(a, off, 64);
  }
  const u64 varying = o ^ a;
  int cur = 0;
  bool have_counts = true;  // h_nxt holds the counts of byte 0
  int counted_shift = 0;
  for (int shift = 0; shift < 64; shift += 8) {
    if (((varying >> shift) & 0xFF) == 0) continue;
    const u64* kin = cur ? k1 : k0;
    const u64* pin = cur ? p1 : p0;
    u64* kout = cur ? k0 : k1;
    u64* pout = cur ? p0 : p1;
    __builtin_amdgcn_wave_barrier();
    if (have_counts && counted_shift == shift) {
      u32* t = h_cur;
      h_cur = h_nxt;
      h_nxt = t;
    } else {  // (the first varying byte is not byte 0: one counting sweep)
      for (int i = lane; i < 256; i += 64) h_cur[i] = 0;
      __builtin_amdgcn_wave_barrier();
      for (u64 i = lane; i < n; i += 64) atomicAdd(&h_cur[(kin[b + i] >> shift) & 0xFF], 1u);
    }
    int nshift = shift + 8;
    while (nshift < 64 && ((varying >> nshift) & 0xFF) == 0) nshift += 8;
    have_counts = nshift < 64;
    counted_shift = nshift;
    __builtin_amdgcn_wave_barrier();
    {
      const u32 h0 = h_cur[4 * lane], h1 = h_cur[4 * lane + 1], h2 = h_cur[4 * lane + 2], h3 = h_cur[4 * lane + 3];
      const u32 s = h0 + h1 + h2 + h3;
      const u32 ex = wave_inclusive_sum(s) - s;
      __builtin_amdgcn_wave_barrier();
      h_cur[4 * lane] = ex;
      h_cur[4 * lane + 1] = ex + h0;
      h_cur[4 * lane + 2] = ex + h0 + h1;
      h_cur[4 * lane + 3] = ex + h0 + h1 + h2;
      for (int i = lane; i < 256; i += 64) h_nxt[i] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    for (u64 base = 0; base < n; base += 256) {
      u64 key[4], pay[4];
      bool valid[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const u64 i = base + 64 * x + lane;
        valid[x] = i < n;
        key[x] = 0;
        pay[x] = 0;
        if (valid[x]) {
          key[x] = kin[b + i];
          if (PAY) pay[x] = pin[b + i];
        }
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        if (base + 64 * x >= n) break;  // (uniform)
        const unsigned d = static_cast<unsigned>((key[x] >> shift) & 0xFF);
        const unsigned long long peers = match_digit8(d, valid[x]);
        u32 before = 0;
        if (valid[x]) before = h_cur[d];
        __builtin_amdgcn_wave_barrier();
        const u32 rank = before + __popcll(peers & lt);
        if (valid[x] && (peers & lt) == 0) h_cur[d] = before + __popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (valid[x]) {
          kout[b + rank] = key[x];
          if (PAY) pout[b + rank] = pay[x];
          if (have_counts) atomicAdd(&h_nxt[(key[x] >> nshift) & 0xFF], 1u);
        }
      }
    }
    __threadfence_block();  // this wave's stores must be visible to its other lanes' loads next pass
    cur ^= 1;
  }
  if (cur) {
    for (u64 i = lane; i < n; i += 64) {
      k0[b + i] = k1[b + i];
      if (PAY) p0[b + i] = p1[b + i];
    }
    __threadfence_block();
  }
}

// ---- block-level segmented sort in LDS ----------------------------------------------------------------
// One workgroup of NT threads per segment of <= cap keys (payloads beside them when PAY): all LSD passes ping-pong between
// two LDS buffers, HBM sees one coalesced read and one coalesced write (the global-memory version moved ~20x the data,
// profiles/r01_g_final_pmc_*).  smem: seg_sort_lds_bytes(cap).
template <int NT, bool PAY>
constexpr size_t seg_sort_lds_bytes(u32 cap) {
  return static_cast<size_t>(cap) * 8 * (PAY ? 4 : 2) + (NT / 64) * 256 * 2 + 32 + (NT / 64) * 16;
}
template <int NT, bool PAY>
__device__ void block_sort_lds(u64* __restrict__ keys, u64* __restrict__ pays, u64 b, u32 n, u32 cap, unsigned char* smem) {
  constexpr int NW = NT / 64;
  static_assert(NT == 64 || NT == 256, "one wave or four");
  u64* s_k = reinterpret_cast<u64*>(smem);                         // [2][cap]
  u64* s_p = s_k + 2 * static_cast<size_t>(cap);                   // [2][cap] when PAY
  u16* wave_cnt = reinterpret_cast<u16*>(s_p + (PAY ? 2 * static_cast<size_t>(cap) : 0));  // [NW][256]
  u32* s4 = reinterpret_cast<u32*>(wave_cnt + NW * 256);           // 8 words
  u64* s_or = reinterpret_cast<u64*>(s4 + 8);                      // [NW]
  u64* s_and = s_or + NW;                                          // [NW]
  const int lane = lane_id();
  const int w = threadIdx.x >> 6;
  u64 o = 0, a = ~0ULL;
  for (u32 i = threadIdx.x; i < n; i += NT) {
    const u64 k = keys[b + i];
    s_k[i] = k;
    if (PAY) s_p[i] = pays[b + i];
    o |= k;
    a &= k;
  }
#pragma unroll
  for (int x = 32; x > 0; x >>= 1) {
    o |= __shfl_xor(o, x, 64);
    a &= __shfl_xor(a, x, 64);
  }
  if (lane == 0) {
    s_or[w] = o;
    s_and[w] = a;
  }
  __syncthreads();
  u64 all_or = 0, all_and = ~0ULL;
#pragma unroll
  for (int x = 0; x < NW; ++x) {
    all_or |= s_or[x];
    all_and &= s_and[x];
  }
  const u64 varying = all_or ^ all_and;
  // wave w owns the contiguous share [q0, q1) of the segment (stable: rank order == index order)
  const u32 per = (n + NW - 1) / NW;
  const u32 q0 = min(n, w * per), q1 = min(n, q0 + per);
  const unsigned long long lt = lanemask_lt();
  u32 cur = 0;
  for (int shift = 0; shift < 64; shift += 8) {
    if (((varying >> shift) & 0xFF) == 0) continue;
    const u64* kin = s_k + static_cast<size_t>(cur) * cap;
    u64* kout = s_k + static_cast<size_t>(cur ^ 1) * cap;
    const u64* pin = s_p + static_cast<size_t>(cur) * cap;
    u64* pout = s_p + static_cast<size_t>(cur ^ 1) * cap;
    for (int i = threadIdx.x; i < NW * 256; i += NT) wave_cnt[i] = 0;
    __syncthreads();
    // pass 1: per-wave digit counts (ranks are recomputed in pass 2 with the same ballot order)
    u16* my_cnt = wave_cnt + w * 256;
    for (u32 base = q0; base < q1; base += 64) {
      const u32 i = base + lane;
      const bool valid = i < q1;
      const unsigned d = valid ? static_cast<unsigned>((kin[i] >> shift) & 0xFF) : 0;
      const unsigned long long peers = match_digit8(d, valid);
      if (valid && (peers & lt) == 0) my_cnt[d] += static_cast<u16>(__popcll(peers));
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (NT == 256) {
      const int t = threadIdx.x;
      u32 c[NW], tot = 0;
#pragma unroll
      for (int x = 0; x < NW; ++x) {
        c[x] = wave_cnt[x * 256 + t];
        tot += c[x];
      }
      u32 total;
      u32 run = block_exclusive_sum_256<u32>(tot, s4, &total);
#pragma unroll
      for (int x = 0; x < NW; ++x) {
        wave_cnt[x * 256 + t] = static_cast<u16>(run);
        run += c[x];
      }
    } else {  // one wave: lane l owns the digits 4 l .. 4 l + 3
      const u32 h0 = wave_cnt[4 * lane], h1 = wave_cnt[4 * lane + 1], h2 = wave_cnt[4 * lane + 2], h3 = wave_cnt[4 * lane + 3];
      const u32 sum = h0 + h1 + h2 + h3;
      const u32 ex = wave_inclusive_sum(sum) - sum;
      __builtin_amdgcn_wave_barrier();
      wave_cnt[4 * lane] = static_cast<u16>(ex);
      wave_cnt[4 * lane + 1] = static_cast<u16>(ex + h0);
      wave_cnt[4 * lane + 2] = static_cast<u16>(ex + h0 + h1);
      wave_cnt[4 * lane + 3] = static_cast<u16>(ex + h0 + h1 + h2);
    }
    __syncthreads();
    // pass 2: scatter in index order
    for (u32 base = q0; base < q1; base += 64) {
      const u32 i = base + lane;
      const bool valid = i < q1;
      u64 k = 0, pv = 0;
      if (valid) {
        k = kin[i];
        if (PAY) pv = pin[i];
      }
      const unsigned d = static_cast<unsigned>((k >> shift) & 0xFF);
      const unsigned long long peers = match_digit8(d, valid);
      u32 before = 0;
      if (valid) before = my_cnt[d];
      __builtin_amdgcn_wave_barrier();
      if (valid && (peers & lt) == 0) my_cnt[d] = static_cast<u16>(before + __popcll(peers));
      __builtin_amdgcn_wave_barrier();
      if (valid) {
        const u32 dst = before + __popcll(peers & lt);
        kout[dst] = k;
        if (PAY) pout[dst] = pv;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  const u64* kfin = s_k + static_cast<size_t>(cur) * cap;
  const u64* pfin = s_p + static_cast<size_t>(cur) * cap;
  for (u32 i = threadIdx.x; i < n; i += NT) {
    keys[b + i] = kfin[i];
    if (PAY) pays[b + i] = pfin[i];
  }
}

// Per-read group sort by size class (round 6; rounds 1-5: one workgroup of 256 threads with LDS for 2048 pairs per read
// whatever its size — 67 KB, two workgroups per CU — and a wave through global memory beyond that).  A read's segment of
// <= cap (key, payload) pairs is sorted by one workgroup whose LDS is sized for the class: the typical read of an ONT
// pass (1 000 - 3 000 matches per launch) gets a workgroup that fits four to eight times per CU.
constexpr int kSegClasses = 5;
__constant__ u32 kSegClassCap[kSegClasses] = {256, 512, 1024, 2048, 4096};
static const u32 kSegClassCapHost[kSegClasses] = {256, 512, 1024, 2048, 4096};
// lists[c * n_seg ...] <- segments of class c (kSegClasses: beyond the largest: a wave each through global memory)
__global__ __launch_bounds__(256) void seg_class_list_kernel(const u64* __restrict__ off, u32 n_seg, u32* __restrict__ lists,
                                                            u32* __restrict__ cnt) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  int cls = -1;
  if (t < n_seg) {
    const u64 n = off[t + 1] - off[t];
    if (n >= 2) {
      cls = kSegClasses;
      for (int c = kSegClasses - 1; c >= 0; --c)
        if (n <= kSegClassCap[c]) cls = c;
    }
  }
  const int lane = lane_id();
  for (int c = 0; c <= kSegClasses; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if (!m) continue;
    u32 base = 0;
    if (lane == 0) base = atomicAdd(cnt + c, static_cast<u32>(__popcll(m)));
    base = __shfl(base, 0, 64);
    if (cls == c) lists[static_cast<size_t>(c) * n_seg + base + __popcll(m & ((1ULL << lane) - 1ULL))] = t;
  }
}
template <int NT>
__global__ __launch_bounds__(NT) void seg_sort_group_lds_kernel(u64* __restrict__ keys, u64* __restrict__ pays,
                                                               const u64* __restrict__ off, const u32* __restrict__ list,
                                                               u32 n_list, u32 cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char seg_smem[];
  if (blockIdx.x >= n_list) return;
  const u32 seg = list[blockIdx.x];
  const u64 b = off[seg];
  block_sort_lds<NT, true>(keys, pays, b, static_cast<u32>(off[seg + 1] - b), cap, seg_smem);
}
__global__ __launch_bounds__(256) void seg_sort_group_big_kernel(u64* k0, u64* k1, u64* p0, u64* p1, const u64* __restrict__ off,
                                                                const u32* __restrict__ list, u32 n_list) {
  __shared__ u32 hist[4][512];
  const u32 q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n_list) return;
  const u32 seg = list[q];
  const u64 b = off[seg], e = off[seg + 1];
  wave_sort_segment<true>(k0, k1, p0, p1, b, e - b, hist[threadIdx.x >> 6]);
}

// Position sort of the intervals of one size class (the chain stage's lists: chain_class_list_kernel), KEYS ONLY — the
// payload of rounds 1-5, the group words, is the same (target, strand) for every match of an interval and nothing behind
// this sort reads a group word's diagonal (the chain kernels take target and strand from the interval's first word), so it
// stays where it is.  One workgroup per interval, the interval in LDS sized for the class.
template <int NT>
__global__ __launch_bounds__(NT) void seg_sort_pos_lds_kernel(u64* __restrict__ pos, const u64* __restrict__ iv_begin,
                                                             const u64* __restrict__ iv_end, const u32* __restrict__ list,
                                                             u32 n_list, u32 cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char seg_smem[];
  if (blockIdx.x >= n_list) return;
  const u32 t = list[blockIdx.x];
  const u64 b = iv_begin[t];
  block_sort_lds<NT, false>(pos, nullptr, b, static_cast<u32>(iv_end[t] - b), cap, seg_smem);
}
// ... and of the intervals beyond the largest class: a wave each, through global memory
__global__ __launch_bounds__(256) void seg_sort_pos_big_kernel(u64* k0, u64* k1, const u64* __restrict__ iv_begin,
                                                              const u64* __restrict__ iv_end, const u32* __restrict__ list,
                                                              u32 n_list) {
  __shared__ u32 hist[4][512];
  const u32 q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= n_list) return;
  const u32 t = list[q];
  const u64 b = iv_begin[t], e = iv_end[t];
  wave_sort_segment<false>(k0, k1, nullptr, nullptr, b, e - b, hist[threadIdx.x >> 6]);
}

// ---- diagonal-band intervals (ram Chain, first loop) ------------------------------------------
// One wave per read segment. Slots: segment [b, e) may hold at most (e-b)/4 intervals, stored at
// slot ceil(b/4)+id (disjoint across segments).
__global__ __launch_bounds__(256) void intervals_kernel(const u64* __restrict__ grp, const u64* __restrict__ seg_off,
                                                       u32 n_seg, u64 bandwidth, u64* __restrict__ slot_begin,
                                                       u64* __restrict__ slot_end, u32* __restrict__ iv_cnt) {
  const u32 seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= n_seg) return;
  const int lane = lane_id();
  const u64 b = seg_off[seg], e = seg_off[seg + 1];
  const long long n = static_cast<long long>(e - b);
  if (n < 4) {
    if (lane == 0) iv_cnt[seg] = 0;
    return;
  }
  const u64 slot_base = (b + 3) >> 2;
  const u64* g = grp + b;
  long long carry_prev = -1;  // local index (end) of the last valid candidate so far
  u32 n_iv = 0;
  const unsigned long long lt = lanemask_lt();
  for (long long base = 1; base <= n; base += 64) {
    const long long il = base + lane;
    const bool in = il <= n;
    long long lprev = 0;
    bool cand = false;
    if (in) {
      const u64 g_prev = g[il - 1];
      long long lo = 0, hi = il - 1;
      while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (g_prev - g[mid] <= bandwidth) hi = mid;
        else lo = mid + 1;
      }
      lprev = lo;
      const u64 gi = il == n ? ~0ULL : g[il];
      cand = (gi - g[lprev] > bandwidth) && (il - lprev >= 4);
    }
    const long long mine = cand ? il : -1;
    const long long incl = wave_inclusive_max(mine);
    long long excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = -1;
    excl = excl > carry_prev ? excl : carry_prev;
    const bool is_new = cand && !(excl > lprev);
    const unsigned long long newmask = __ballot(is_new);
    if (is_new) {
      const u32 id = n_iv + __popcll(newmask & lt);
      slot_begin[slot_base + id] = b + static_cast<u64>(lprev);
      if (excl >= 0) slot_end[slot_base + id - 1] = b + static_cast<u64>(excl);
    }
    const long long wmax = __shfl(incl, 63, 64);
    carry_prev = wmax > carry_prev ? wmax : carry_prev;
    n_iv += __popcll(newmask);
  }
  if (lane == 0) {
    if (n_iv) slot_end[slot_base + n_iv - 1] = b + static_cast<u64>(carry_prev);
    iv_cnt[seg] = n_iv;
  }
}

__global__ __launch_bounds__(256) void intervals_gather_kernel(const u64* __restrict__ slot_begin,
                                                              const u64* __restrict__ slot_end,
                                                              const u64* __restrict__ seg_off,
                                                              const u32* __restrict__ iv_off, u32 n_seg,
                                                              u64* __restrict__ iv_begin, u64* __restrict__ iv_end,
                                                              u32* __restrict__ iv_read) {
  const u32 seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= n_seg) return;
  const u32 o0 = iv_off[seg], o1 = iv_off[seg + 1];
  const u64 slot_base = (seg_off[seg] + 3) >> 2;
  for (u32 i = lane_id(); i < o1 - o0; i += 64) {
    iv_begin[o0 + i] = slot_begin[slot_base + i];
    iv_end[o0 + i] = slot_end[slot_base + i];
    iv_read[o0 + i] = seg;
  }
}

// ---- LIS chain + overlap emission (ram Chain second loop + LongestSubsequence) ----------------
// ram Chain: split the chain where consecutive lhs positions differ by more than `gap`, score each piece by
// covered bases, emit the overlaps.  cp(m) = packed positions of chain element m (ascending).
template <typename CP>
__device__ __forceinline__ void chain_emit(CP cp, u32 longest, bool strand, u64 g0, u32 lhs_id, u32 k, u32 chain,
                                           u32 min_matches, u32 gap, Overlap* __restrict__ slots,
                                           u8* __restrict__ slot_flags, bool writer,
                                           u64* __restrict__ anchors /* region of this interval or null */,
                                           u64 anchor_base, u64* __restrict__ slot_aoff,
                                           u32* __restrict__ slot_acnt) {
  u32 emitted = 0;
  u32 l = 0;
  for (u32 kk = 1; kk <= longest; ++kk) {
    const u32 lhs_k = kk < longest ? static_cast<u32>(cp(kk) >> 32) : 0xFFFFFFFFu;
    const u32 lhs_km1 = static_cast<u32>(cp(kk - 1) >> 32);
    if (lhs_k - lhs_km1 > gap) {
      if (kk - l < chain) {
        l = kk;
        continue;
      }
      u32 lhs_matches = 0, lhs_begin = 0, lhs_end = 0;
      u32 rhs_matches = 0, rhs_begin = 0, rhs_end = 0;
      for (u32 m = l; m < kk; ++m) {
        const u64 mm = cp(m);
        const u32 lhs_pos = static_cast<u32>(mm >> 32);
        if (lhs_pos > lhs_end) {
          lhs_matches += lhs_end - lhs_begin;
          lhs_begin = lhs_pos;
        }
        lhs_end = lhs_pos + k;
        u32 rhs_pos = static_cast<u32>(mm);
        rhs_pos = strand ? rhs_pos : (1U << 31) - (rhs_pos + k - 1);
        if (rhs_pos > rhs_end) {
          rhs_matches += rhs_end - rhs_begin;
          rhs_begin = rhs_pos;
        }
        rhs_end = rhs_pos + k;
      }
      lhs_matches += lhs_end - lhs_begin;
      rhs_matches += rhs_end - rhs_begin;
      const u32 score = lhs_matches < rhs_matches ? lhs_matches : rhs_matches;
      if (score < min_matches) {
        l = kk;
        continue;
      }
      if (writer) {
        const u64 ml = cp(l), mr = cp(kk - 1);
        Overlap o;
        o.lhs_id = lhs_id;
        o.lhs_begin = static_cast<u32>(ml >> 32);
        o.lhs_end = k + static_cast<u32>(mr >> 32);
        o.rhs_id = static_cast<u32>(g0 >> 33);
        o.rhs_begin = strand ? static_cast<u32>(ml) : static_cast<u32>(mr);
        o.rhs_end = k + (strand ? static_cast<u32>(mr) : static_cast<u32>(ml));
        o.score = score;
        o.strand = strand ? 1u : 0u;
        slots[emitted] = o;
        slot_flags[emitted] = 1;
        if (anchors) {  // the chain pieces are disjoint sub-ranges of [0, longest): store them in place
          for (u32 m = l; m < kk; ++m) anchors[m] = cp(m);
          slot_aoff[emitted] = anchor_base + l;
          slot_acnt[emitted] = kk - l;
        }
      }
      ++emitted;
      l = kk;
    }
  }
}

// The same for a WAVE that holds the chain (chain_wave): the pieces from ballots of the gap test, a piece's covered bases
// summed by all lanes — the serial loop above adds up the measure of the union of the k-mers [pos, pos + k) of ascending
// positions, which is sum over consecutive pairs of min(k, difference) + k.
template <typename CP>
__device__ __forceinline__ void chain_emit_wave(CP cp, u32 longest, bool strand, u64 g0, u32 lhs_id, u32 k, u32 chain,
                                                u32 min_matches, u32 gap, Overlap* __restrict__ slots,
                                                u8* __restrict__ slot_flags, u64* __restrict__ anchors, u64 anchor_base,
                                                u64* __restrict__ slot_aoff, u32* __restrict__ slot_acnt) {
  const int lane = lane_id();
  auto rhs_of = [&](u64 mm) -> u32 {
    const u32 rhs_pos = static_cast<u32>(mm);
    return strand ? rhs_pos : (1U << 31) - (rhs_pos + k - 1);
  };
  u32 emitted = 0;
  u32 l = 0;
  for (u32 base = 1; base <= longest; base += 64) {
    const u32 kk_mine = base + static_cast<u32>(lane);
    bool brk = false;
    if (kk_mine <= longest) {
      const u32 lhs_k = kk_mine < longest ? static_cast<u32>(cp(kk_mine) >> 32) : 0xFFFFFFFFu;
      const u32 lhs_km1 = static_cast<u32>(cp(kk_mine - 1) >> 32);
      brk = lhs_k - lhs_km1 > gap;
    }
    unsigned long long pieces = __ballot(brk);
    while (pieces) {
      const u32 kk = base + static_cast<u32>(__builtin_ctzll(pieces));
      pieces &= pieces - 1;
      if (kk - l >= chain) {
        u32 lsum = 0, rsum = 0;
        for (u32 m = l + static_cast<u32>(lane); m + 1 < kk; m += 64) {
          const u64 a = cp(m), b = cp(m + 1);
          const u32 dl = static_cast<u32>(b >> 32) - static_cast<u32>(a >> 32), dr = rhs_of(b) - rhs_of(a);
          lsum += dl < k ? dl : k;
          rsum += dr < k ? dr : k;
        }
        const u32 lhs_matches = wave_sum(lsum) + k, rhs_matches = wave_sum(rsum) + k;
        const u32 score = lhs_matches < rhs_matches ? lhs_matches : rhs_matches;
        if (score >= min_matches) {
          if (lane == 0) {
            const u64 ml = cp(l), mr = cp(kk - 1);
            Overlap o;
            o.lhs_id = lhs_id;
            o.lhs_begin = static_cast<u32>(ml >> 32);
            o.lhs_end = k + static_cast<u32>(mr >> 32);
            o.rhs_id = static_cast<u32>(g0 >> 33);
            o.rhs_begin = strand ? static_cast<u32>(ml) : static_cast<u32>(mr);
            o.rhs_end = k + (strand ? static_cast<u32>(mr) : static_cast<u32>(ml));
            o.score = score;
            o.strand = strand ? 1u : 0u;
            slots[emitted] = o;
            slot_flags[emitted] = 1;
            if (anchors) {
              slot_aoff[emitted] = anchor_base + l;
              slot_acnt[emitted] = kk - l;
            }
          }
          if (anchors)  // the chain pieces are disjoint sub-ranges of [0, longest): stored in place
            for (u32 m = l + static_cast<u32>(lane); m < kk; m += 64) anchors[m] = cp(m);
          ++emitted;
        }
      }
      l = kk;
    }
  }
}

// Small intervals (chain <= n <= kChainSmallCap): one LANE per interval, every array private to the lane in LDS
// ([element][lane] layout: bank = lane, conflict-free), no cross-lane traffic at all.
constexpr u32 kChainSmallCap = 32;
__global__ __launch_bounds__(64) void chain_small_kernel(const u64* __restrict__ grp, const u64* __restrict__ pos,
                                                        const u64* __restrict__ iv_begin,
                                                        const u64* __restrict__ iv_end,
                                                        const u32* __restrict__ iv_read, u32 n_iv,
                                                        const u32* __restrict__ ids, u32 first, u32 k, u32 chain,
                                                        u32 min_matches, u32 gap, u32 slot_div,
                                                        Overlap* __restrict__ slots, u8* __restrict__ slot_flags,
                                                        u64* __restrict__ anchors, u64* __restrict__ slot_aoff,
                                                        u32* __restrict__ slot_acnt) {
  __shared__ u64 s_pos[kChainSmallCap][64];
  __shared__ u64 s_tail[kChainSmallCap + 1][64];
  __shared__ u8 s_tidx[kChainSmallCap + 1][64];
  __shared__ u8 s_pred[kChainSmallCap][64];
  const u32 lane = threadIdx.x;
  const u32 t = blockIdx.x * 64 + lane;
  if (t >= n_iv) return;
  const u64 b = iv_begin[t];
  const u64 n64 = iv_end[t] - b;
  if (n64 < chain || n64 > kChainSmallCap) return;  // larger intervals: chain_kernel
  const u32 n = static_cast<u32>(n64);
  const u64* p = pos + b;
  const u64 g0 = grp[b];
  const bool strand = (g0 >> 32) & 1;
#pragma unroll 8
  for (u32 i = 0; i < kChainSmallCap; ++i)
    if (i < n) s_pos[i][lane] = p[i];
  // ram sorts the interval by positions before the LIS: <= 32 elements, private to the lane -> insertion sort
  // (positions are distinct, so the result is the unique sorted order)
  for (u32 i = 1; i < n; ++i) {
    const u64 key = s_pos[i][lane];
    u32 j = i;
    while (j > 0 && s_pos[j - 1][lane] > key) {
      s_pos[j][lane] = s_pos[j - 1][lane];
      --j;
    }
    s_pos[j][lane] = key;
  }
  u32 longest = 0;
  for (u32 it = 0; it < n; ++it) {
    const u64 cur = s_pos[it][lane];
    const u32 lhs = static_cast<u32>(cur >> 32), rhs = static_cast<u32>(cur);
    u32 lo = 1, hi = longest;
    while (lo <= hi) {
      const u32 mid = lo + (hi - lo) / 2;
      const u64 q = s_tail[mid][lane];
      const u32 ql = static_cast<u32>(q >> 32), qr = static_cast<u32>(q);
      if (ql < lhs && (strand ? qr < rhs : qr > rhs)) lo = mid + 1;
      else hi = mid - 1;
    }
    s_pred[it][lane] = lo > 1 ? s_tidx[lo - 1][lane] : static_cast<u8>(0);
    s_tidx[lo][lane] = static_cast<u8>(it);
    s_tail[lo][lane] = cur;
    longest = longest > lo ? longest : lo;
  }
  if (longest < chain) return;
  {
    u32 j = s_tidx[longest][lane];
    for (u32 i = 0; i < longest; ++i) {
      const u32 nj = s_pred[j][lane];
      s_tidx[longest - 1 - i][lane] = static_cast<u8>(j);
      j = nj;
    }
    for (u32 m = 0; m < longest; ++m) s_tail[m][lane] = s_pos[s_tidx[m][lane]][lane];
  }
  const u64 slot_base = (b + slot_div - 1) / slot_div;
  chain_emit([&](u32 m) { return s_tail[m][lane]; }, longest, strand, g0, ids[first + iv_read[t]], k, chain,
             min_matches, gap, slots + slot_base, slot_flags + slot_base, true, anchors ? anchors + b : nullptr, b,
             slot_aoff ? slot_aoff + slot_base : nullptr, slot_acnt ? slot_acnt + slot_base : nullptr);
}

// One WAVE per interval.  ram's patience LIS is sequential over the elements, but its binary search only
// needs the predicate "tail[len] precedes cur" at the probed lengths: all 64 lanes evaluate the predicate
// for 64 lengths at once (tails live in LDS), a ballot turns it into a bit mask and the *same* probe
// sequence ram would follow is then replayed on the mask with scalar ALU only.  Intervals of up to
// kChainBigCap matches run entirely out of LDS; larger ones use the same code on global scratch.
constexpr u32 kChainBigCap = 8192;  // intervals of up to this many matches run out of a whole workgroup's LDS (chain_big_kernel)

template <bool GLOBAL>
__device__ __forceinline__ void chain_sync() {
  if (GLOBAL) __threadfence_block();  // lane 0's global stores must be visible to the other lanes' loads
  __builtin_amdgcn_wave_barrier();
}

template <typename IdxT, bool GLOBAL>
__device__ void chain_wave(const u64* __restrict__ p, u32 n, bool strand, u64 g0, u32 lhs_id, u32 k, u32 chain,
                           u32 min_matches, u32 gap, u64* tail_pos, IdxT* tail_idx, IdxT* pred, u64* maskbuf,
                           Overlap* __restrict__ slots, u8* __restrict__ slot_flags, u64* __restrict__ anchors,
                           u64 anchor_base, u64* __restrict__ slot_aoff, u32* __restrict__ slot_acnt) {
  const int lane = lane_id();
  u32 longest = 0;
  for (u32 base = 0; base < n; base += 64) {
    const u64 mine = (base + lane < n) ? p[base + lane] : 0;
    const u32 cnt = min(64u, n - base);
    // A colinear stretch is a run of elements that each EXTEND the longest chain: ram's search for such an element
    // answers "precedes" at every probe, and both the probed lengths and the tails found there are then known in
    // advance for the whole run (element t of the run meets the chain at length longest + t: its probes see old tails
    // and the run's own earlier elements).  So the run is tried as a whole: lanes t0 .. cnt-1 store their elements at
    // the lengths they would get, every lane replays its own ~log2(longest) probes against that speculated tail array,
    // and one ballot gives the first lane whose search would have turned: the elements before it are appended at once
    // (exactly the state the one-by-one loop below would leave), that element goes through the general replay.  A run
    // is tried at the start of a block and after every element that extended the chain.
    bool try_run = true;
    for (u32 t = 0; t < cnt;) {
      if (try_run && cnt - t >= 2) {
        const bool active = static_cast<u32>(lane) >= t && static_cast<u32>(lane) < cnt;
        const u32 my_long = longest + (static_cast<u32>(lane) - t);  // the chain length this lane's element meets
        const IdxT first_pred = longest ? tail_idx[longest] : static_cast<IdxT>(0);
        if (active) tail_pos[my_long + 1] = mine;
        chain_sync<GLOBAL>();
        bool turned = false;
        if (active) {
          const u32 lhs = static_cast<u32>(mine >> 32), rhs = static_cast<u32>(mine);
          for (u32 a = my_long; a >= 1 && !turned; a >>= 1) {
            const u64 q = tail_pos[my_long + 1 - a + ((a - 1) >> 1)];
            const u32 ql = static_cast<u32>(q >> 32), qr = static_cast<u32>(q);
            turned = !(ql < lhs && (strand ? qr < rhs : qr > rhs));
          }
        }
        const unsigned long long bad = __ballot(turned);
        const u32 stop = bad ? static_cast<u32>(__builtin_ctzll(bad)) : cnt;  // first lane whose search turns
        if (active && static_cast<u32>(lane) < stop) {
          tail_idx[my_long + 1] = static_cast<IdxT>(base + lane);
          pred[base + lane] = static_cast<u32>(lane) == t ? first_pred : static_cast<IdxT>(base + lane - 1);
        }
        chain_sync<GLOBAL>();
        longest += stop - t;
        t = stop;
        try_run = false;
        if (t >= cnt) break;
      }
      const u64 cur = __shfl(mine, static_cast<int>(t), 64);
      const u32 lhs = static_cast<u32>(cur >> 32), rhs = static_cast<u32>(cur);
      u32 lo = 1, hi = longest;
      bool appended = false;
      if (longest > 64) {
        // The common case of a colinear interval: cur extends the longest chain.  ram's search then answers "precedes" at
        // every probe, and that probe path is known in advance — probe i looks at length
        // longest + 1 - a + (a - 1) / 2 with a = longest >> i, while a >= 1 — so lane i tests probe i, and one ballot
        // says whether the search ends at longest + 1.  (Any failed probe: the general replay below.)
        const u32 a = lane < 32 ? longest >> lane : 0u;
        bool fail = false;
        if (a >= 1) {
          const u64 q = tail_pos[longest + 1 - a + ((a - 1) >> 1)];
          const u32 ql = static_cast<u32>(q >> 32), qr = static_cast<u32>(q);
          fail = !(ql < lhs && (strand ? qr < rhs : qr > rhs));
        }
        if (!__ballot(fail)) {
          lo = longest + 1;
          appended = true;
        }
      }
      if (appended) {
      } else if (longest > 512) {
        // Long chains: ram's binary search probes ~log2(longest) tails, one after the other.  Here six levels of its
        // search tree are evaluated at once: lane h (heap index 1..63) derives the (lo, hi) range ram would have at tree
        // node h from the current range, tests the predicate at that node's midpoint, a ballot collects the 63 answers
        // and the walk down the six levels is scalar bit tests — ram's exact probe sequence (the predicate need not be
        // monotone), two round trips to the tails for chains of up to 4096 instead of twelve dependent reads.
        while (lo <= hi) {
          const u32 h = static_cast<u32>(lane);
          u32 l = lo, r = hi;
          bool valid = h >= 1;
          if (valid) {
            const int depth = 31 - __clz(static_cast<int>(h));
            for (int d = depth - 1; d >= 0; --d) {
              if (l > r) break;
              const u32 mid = l + (r - l) / 2;
              if ((h >> d) & 1u) l = mid + 1;
              else r = mid - 1;
            }
            valid = l <= r;
          }
          bool ok = false;
          if (valid) {
            const u64 q = tail_pos[l + (r - l) / 2];
            const u32 ql = static_cast<u32>(q >> 32), qr = static_cast<u32>(q);
            ok = ql < lhs && (strand ? qr < rhs : qr > rhs);
          }
          const unsigned long long m = __ballot(ok);
          u32 node = 1;
          for (int level = 0; level < 6 && lo <= hi; ++level) {
            const u32 mid = lo + (hi - lo) / 2;
            const u32 bit = static_cast<u32>((m >> node) & 1ULL);
            if (bit) lo = mid + 1;
            else hi = mid - 1;
            node = 2 * node + bit;
          }
        }
      } else if (longest <= 64) {
        bool ok = false;
        if (static_cast<u32>(lane) < longest) {
          const u64 q = tail_pos[lane + 1];
          const u32 ql = static_cast<u32>(q >> 32), qr = static_cast<u32>(q);
          ok = ql < lhs && (strand ? qr < rhs : qr > rhs);
        }
        const unsigned long long m = __ballot(ok);
        while (lo <= hi) {
          const u32 mid = lo + (hi - lo) / 2;
          if ((m >> (mid - 1)) & 1ULL) lo = mid + 1;
          else hi = mid - 1;
        }
      } else {
        const u32 nchunks = (longest + 63) >> 6;
        for (u32 c = 0; c < nchunks; ++c) {
          const u32 len = c * 64 + lane + 1;
          bool ok = false;
          if (len <= longest) {
            const u64 q = tail_pos[len];
            const u32 ql = static_cast<u32>(q >> 32), qr = static_cast<u32>(q);
            ok = ql < lhs && (strand ? qr < rhs : qr > rhs);
          }
          const unsigned long long m = __ballot(ok);
          if (lane == 0) maskbuf[c] = m;
        }
        chain_sync<GLOBAL>();
        while (lo <= hi) {
          const u32 mid = lo + (hi - lo) / 2;
          if ((maskbuf[(mid - 1) >> 6] >> ((mid - 1) & 63)) & 1ULL) lo = mid + 1;
          else hi = mid - 1;
        }
      }
      const u32 it = base + t;
      const IdxT prev = lo > 1 ? tail_idx[lo - 1] : static_cast<IdxT>(0);  // minimal[0] == 0 in ram
      chain_sync<GLOBAL>();  // every lane has read maskbuf / tail_idx before lane 0 overwrites
      if (lane == 0) {
        pred[it] = prev;
        tail_idx[lo] = static_cast<IdxT>(it);
        tail_pos[lo] = cur;
      }
      chain_sync<GLOBAL>();
      try_run = lo > longest;  // it extended the chain: the next ones may well do so too
      longest = longest > lo ? longest : lo;
      ++t;
    }
  }
  if (longest < chain) return;
  {
    // backtrack: chain indices ascending into tail_idx[0 .. longest).  Rounds 1-5 followed pred[] one element at a time out
    // of LDS — `longest` dependent reads of ~100 cycles, every lane doing the same — and the emission below ran two more
    // serial loops over the chain on all 64 lanes: for the long colinear intervals of HiFi reads (4 500 matches per read in a
    // polishing round's mapping, one wave per CU in that size class) the three were ~95 % of the stage (round 6: 0.63 ms per
    // interval of which ~0.02 ms the LIS itself).  Now: the walk back goes 64 elements at a time out of REGISTERS (a block's
    // predecessors, one per lane; a run of elements that each follow the one before is taken in one step from a ballot),
    // leaving one membership mask per block, and the indices are written block by block from the masks.
    const u32 nblk = (n + 63) >> 6;
    for (u32 b2 = lane; b2 < nblk; b2 += 64) maskbuf[b2] = 0;
    u32 j = tail_idx[longest];
    chain_sync<GLOBAL>();
    u32 left = longest;
    for (int b2 = static_cast<int>(j >> 6); b2 >= 0 && left; --b2) {
      const u32 idx = (static_cast<u32>(b2) << 6) + static_cast<u32>(lane);
      const u32 pr = idx < n ? static_cast<u32>(pred[idx]) : 0u;
      // bit i: element i of the block follows element i - 1 of the block
      const unsigned long long cont = __ballot(lane > 0 && idx < n && pr + 1 == idx);
      unsigned long long m = 0;
      while (left && static_cast<int>(j >> 6) == b2) {
        const u32 l = j & 63u;
        const unsigned long long upto = l == 63 ? ~0ULL : ((2ULL << l) - 1ULL);
        const unsigned long long stops = ~cont & upto;                                   // (bit 0 is always one of them)
        u32 r = 63u - static_cast<u32>(__builtin_clzll(stops));                           // the run l, l - 1, .., r
        if (l - r + 1 > left) r = l + 1 - left;
        m |= upto & ~((1ULL << r) - 1ULL);
        left -= l - r + 1;
        j = static_cast<u32>(__shfl(static_cast<int>(pr), static_cast<int>(r), 64));       // where element r came from
      }
      if (lane == 0) maskbuf[b2] = m;
    }
    chain_sync<GLOBAL>();
    u32 at = 0;
    for (u32 b2 = 0; b2 < nblk; ++b2) {
      const unsigned long long m = maskbuf[b2];
      if ((m >> lane) & 1ULL) tail_idx[at + static_cast<u32>(__popcll(m & ((1ULL << lane) - 1ULL)))] = static_cast<IdxT>((b2 << 6) + lane);
      at += static_cast<u32>(__popcll(m));
    }
    chain_sync<GLOBAL>();
    // positions of the chain elements into tail_pos[0 .. longest)
    for (u32 m = lane; m < longest; m += 64) tail_pos[m] = p[tail_idx[m]];
    chain_sync<GLOBAL>();
  }
  chain_emit_wave([&](u32 m) { return tail_pos[m]; }, longest, strand, g0, lhs_id, k, chain, min_matches, gap, slots,
                  slot_flags, anchors, anchor_base, slot_aoff, slot_acnt);
}

// Size classes of the chain stage.  Every interval of more than kChainSmallCap matches is handled by ONE wave = one
// workgroup whose dynamic LDS is sized for the class (tails, tail indices, predecessors of the whole interval), so the
// number of resident waves per CU follows the interval size (3 KB for <= 256 matches ... 99 KB for <= 8192) and no wave
// waits for a sibling's longer interval.  Larger intervals run the same code on global scratch (class kChainClasses).
// (finer classes in the range where the intervals of a polishing round's read-to-contig maps fall, 300 - 700 matches: a
// wave's LDS is ~12 B per match of its class, and the stage is a serial LIS per wave — occupancy is its throughput)
constexpr int kChainClasses = 8;
__constant__ u32 kChainClassCap[kChainClasses] = {128, 256, 512, 768, 1024, 2048, 4096, kChainBigCap};
static const u32 kChainClassCapHost[kChainClasses] = {128, 256, 512, 768, 1024, 2048, 4096, kChainBigCap};
inline size_t chain_class_lds(u32 cap) {
  return static_cast<size_t>(cap + 1) * 8 + (cap / 64 + 2) * 8 + static_cast<size_t>(cap + 2) * 2 + static_cast<size_t>(cap) * 2 + 64;
}

__global__ __launch_bounds__(64) void chain_kernel(const u64* __restrict__ grp, const u64* __restrict__ pos,
                                                  const u64* __restrict__ iv_begin, const u64* __restrict__ iv_end,
                                                  const u32* __restrict__ iv_read, const u32* __restrict__ list,
                                                  u32 n_list, u32 cap, const u32* __restrict__ ids, u32 first, u32 k,
                                                  u32 chain, u32 min_matches, u32 gap, u32 slot_div,
                                                  u64* __restrict__ g_tail_pos, u32* __restrict__ g_tail_idx,
                                                  u32* __restrict__ g_pred, u64* __restrict__ g_mask,
                                                  Overlap* __restrict__ slots, u8* __restrict__ slot_flags,
                                                  u64* __restrict__ anchors, u64* __restrict__ slot_aoff,
                                                  u32* __restrict__ slot_acnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chain_smem[];
  if (blockIdx.x >= n_list) return;
  const u32 t = list[blockIdx.x];
  const u64 b = iv_begin[t];
  const u32 n = static_cast<u32>(iv_end[t] - b);
  const u64 g0 = grp[b];
  const bool strand = (g0 >> 32) & 1;
  const u32 lhs_id = ids[first + iv_read[t]];
  const u64 slot_base = (b + slot_div - 1) / slot_div;
  if (cap) {
    u64* tail_pos = reinterpret_cast<u64*>(chain_smem);
    u64* maskbuf = tail_pos + (cap + 1);
    u16* tail_idx = reinterpret_cast<u16*>(maskbuf + (cap / 64 + 2));
    u16* pred = tail_idx + (cap + 2);
    chain_wave<u16, false>(pos + b, n, strand, g0, lhs_id, k, chain, min_matches, gap, tail_pos, tail_idx, pred, maskbuf,
                           slots + slot_base, slot_flags + slot_base, anchors ? anchors + b : nullptr, b,
                           slot_aoff ? slot_aoff + slot_base : nullptr, slot_acnt ? slot_acnt + slot_base : nullptr);
  } else {
    chain_wave<u32, true>(pos + b, n, strand, g0, lhs_id, k, chain, min_matches, gap, g_tail_pos + b + t,
                          g_tail_idx + b + t, g_pred + b, g_mask + (b >> 6) + t, slots + slot_base,
                          slot_flags + slot_base, anchors ? anchors + b : nullptr, b,
                          slot_aoff ? slot_aoff + slot_base : nullptr, slot_acnt ? slot_acnt + slot_base : nullptr);
  }
}

// lists[c * n_iv ...] <- intervals of class c (order within a list is irrelevant: results go to per-interval slots)
__global__ __launch_bounds__(256) void chain_class_list_kernel(const u64* __restrict__ iv_begin,
                                                              const u64* __restrict__ iv_end, u32 n_iv, u32 chain,
                                                              u32* __restrict__ lists, u32* __restrict__ cnt) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  int cls = -1;
  if (t < n_iv) {
    const u64 n = iv_end[t] - iv_begin[t];
    if (n > kChainSmallCap && n >= chain) {
      cls = kChainClasses;
      for (int c = kChainClasses - 1; c >= 0; --c)
        if (n <= kChainClassCap[c]) cls = c;
    }
  }
  const int lane = lane_id();
  for (int c = 0; c <= kChainClasses; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if (!m) continue;
    u32 base = 0;
    if (lane == 0) base = atomicAdd(cnt + c, static_cast<u32>(__popcll(m)));
    base = __shfl(base, 0, 64);
    if (cls == c) lists[static_cast<size_t>(c) * n_iv + base + __popcll(m & ((1ULL << lane) - 1ULL))] = t;
  }
}

__global__ void compact_overlaps_kernel(const Overlap* __restrict__ slots, const u8* __restrict__ flags,
                                        const u32* __restrict__ scan, u64 n_slots, Overlap* __restrict__ out) {
  u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n_slots && flags[i]) out[scan[i]] = slots[i];
}

__global__ void compact_aux_kernel(const u64* __restrict__ slot_aoff, const u32* __restrict__ slot_acnt,
                                   const u8* __restrict__ flags, const u32* __restrict__ scan, u64 n_slots,
                                   u64* __restrict__ aoff, u32* __restrict__ acnt) {
  u64 i = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n_slots && flags[i]) {
    aoff[scan[i]] = slot_aoff[i];
    acnt[scan[i]] = slot_acnt[i];
  }
}

__global__ void read_ovl_off_kernel(const u64* __restrict__ seg_off, const u32* __restrict__ scan, u32 slot_div,
                                    u32 n, u32* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = scan[(seg_off[i] + slot_div - 1) / slot_div];
}

}  // namespace

// Chain stage: matches of reads [first, last) are in e.m_grp[0] / e.m_pos[0] (both ping-pong sides reserved for
// H + 1 entries), segmented per read by e.seg_off[nr + 1] -> overlaps in (read, emission) order in `out`.
// (ram Map after the index probes: sort by group, diagonal bands, per-band LIS, overlap emission.)
void chain_matches(Engine& e, const ReadsDev& r, u32 first, u32 last, u64 H, MapOut& out) {
  hipStream_t s = e.stream;
  const u32 nr = last - first;
  u32* ovl_read_off = out.ovl_read_off.get<u32>(static_cast<size_t>(nr) + 1);
  u64* seg_off = e.seg_off.as<u64>();
  if (H == 0) {
    RVN_HIP(hipMemsetAsync(ovl_read_off, 0, (static_cast<size_t>(nr) + 1) * 4, s));
    return;
  }
  u64* g0 = e.m_grp[0].as<u64>();
  u64* g1 = e.m_grp[1].as<u64>();
  u64* p0 = e.m_pos[0].as<u64>();
  u64* p1 = e.m_pos[1].as<u64>();
  {
    StageTimer t(e, StageTimes::kSegSort);
    u32* lists = e.chain_big.get<u32>(static_cast<size_t>(nr) * (kSegClasses + 1) + 16);
    u32* d_cnt = lists + static_cast<size_t>(nr) * (kSegClasses + 1);
    RVN_HIP(hipMemsetAsync(d_cnt, 0, (kSegClasses + 1) * 4, s));
    seg_class_list_kernel<<<div_up(nr, 256), 256, 0, s>>>(seg_off, nr, lists, d_cnt);
    RVN_LAUNCH_CHECK();
    read_back(e, d_cnt, (kSegClasses + 1) * 4);
    u32 n_cls[kSegClasses + 1];
    std::memcpy(n_cls, e.h_pin, sizeof(n_cls));
    static bool attr_set = false;
    if (!attr_set) {
      RVN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(seg_sort_group_lds_kernel<256>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(seg_sort_lds_bytes<256, true>(kSegClassCapHost[kSegClasses - 1]))));
      attr_set = true;
    }
    // the largest first: their workgroups run longest
    if (n_cls[kSegClasses])
      RVN_KLAUNCH(kKSegSortGroup, seg_sort_group_big_kernel<<<div_up(n_cls[kSegClasses], 4), 256, 0, s>>>(
                                      g0, g1, p0, p1, seg_off, lists + static_cast<size_t>(kSegClasses) * nr, n_cls[kSegClasses]));
    for (int c = kSegClasses - 1; c >= 0; --c) {
      if (!n_cls[c]) continue;
      const u32 cap = kSegClassCapHost[c];
      const u32* list = lists + static_cast<size_t>(c) * nr;
      if (cap <= 512)
        RVN_KLAUNCH(kKSegSortGroup, seg_sort_group_lds_kernel<64><<<n_cls[c], 64, seg_sort_lds_bytes<64, true>(cap), s>>>(
                                        g0, p0, seg_off, list, n_cls[c], cap));
      else
        RVN_KLAUNCH(kKSegSortGroup, seg_sort_group_lds_kernel<256><<<n_cls[c], 256, seg_sort_lds_bytes<256, true>(cap), s>>>(
                                        g0, p0, seg_off, list, n_cls[c], cap));
    }
    t.stop();
  }
  u32 NI = 0;
  const u64 n_slots4 = (H + 3) / 4 + 1;
  {
    StageTimer t(e, StageTimes::kIntervals);
    u64* slot_begin = e.iv_slot_begin.get<u64>(n_slots4 + 1);
    u64* slot_end = e.iv_slot_end.get<u64>(n_slots4 + 1);
    u32* iv_cnt = e.iv_cnt.get<u32>(static_cast<size_t>(nr) + 1);
    u32* iv_off = e.iv_off.get<u32>(static_cast<size_t>(nr) + 2);
    RVN_KLAUNCH(kKIntervals, intervals_kernel<<<div_up(nr, 4), 256, 0, s>>>(g0, seg_off, nr, e.bandwidth, slot_begin, slot_end, iv_cnt));
    exclusive_scan_u32_u32(iv_cnt, iv_off, nr, e.scan_tmp, s);
    NI = static_cast<u32>(read_back(e, iv_off + nr, 4));
    out.n_intervals = NI;
    if (NI) {
      u64* iv_begin = e.iv_begin.get<u64>(static_cast<size_t>(NI) + 1);
      u64* iv_end = e.iv_end.get<u64>(static_cast<size_t>(NI) + 1);
      u32* iv_read = e.tmp_b.get<u32>(static_cast<size_t>(NI) + 1);
      RVN_KLAUNCH(kKIntervalsGather, intervals_gather_kernel<<<div_up(nr, 4), 256, 0, s>>>(
                                         slot_begin, slot_end, seg_off, iv_off, nr, iv_begin, iv_end, iv_read));
    }
    t.stop();
  }
  if (NI == 0) {
    RVN_HIP(hipMemsetAsync(ovl_read_off, 0, (static_cast<size_t>(nr) + 1) * 4, s));
    return;
  }
  const u32 slot_div = std::max(1u, std::min(4u, e.chain));
  const u64 n_slots = (H + slot_div - 1) / slot_div + 1;
  Overlap* slots = e.ovl_slots.get<Overlap>(n_slots + 1);
  u8* slot_flags = e.ovl_flags.get<u8>(n_slots + 1);
  {
    StageTimer t(e, StageTimes::kChain);
    u64* iv_begin = e.iv_begin.as<u64>();
    u64* iv_end = e.iv_end.as<u64>();
    u32* iv_read = e.tmp_b.as<u32>();
    // size classes of the intervals (position sort and chain kernels both go by them)
    u32* lists = e.chain_big.get<u32>(static_cast<size_t>(NI) * (kChainClasses + 1) + 16);
    u32* d_cnt = lists + static_cast<size_t>(NI) * (kChainClasses + 1);
    RVN_HIP(hipMemsetAsync(d_cnt, 0, (kChainClasses + 1) * 4, s));
    chain_class_list_kernel<<<div_up(NI, 256), 256, 0, s>>>(iv_begin, iv_end, NI, e.chain, lists, d_cnt);
    RVN_LAUNCH_CHECK();
    read_back(e, d_cnt, (kChainClasses + 1) * 4);
    u32 n_cls[kChainClasses + 1];
    std::memcpy(n_cls, e.h_pin, sizeof(n_cls));
    static bool attr_set = false;
    if (!attr_set) {
      RVN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(chain_class_lds(kChainBigCap))));
      RVN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(seg_sort_pos_lds_kernel<256>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                  static_cast<int>(seg_sort_lds_bytes<256, false>(kChainBigCap))));
      attr_set = true;
    }
    // sort every interval by positions: the largest classes first (their workgroups run longest)
    if (n_cls[kChainClasses])
      RVN_KLAUNCH(kKSegSortPos, seg_sort_pos_big_kernel<<<div_up(n_cls[kChainClasses], 4), 256, 0, s>>>(
                                    p0, p1, iv_begin, iv_end, lists + static_cast<size_t>(kChainClasses) * NI, n_cls[kChainClasses]));
    for (int c = kChainClasses - 1; c >= 0; --c) {
      if (!n_cls[c]) continue;
      const u32 cap = kChainClassCapHost[c];
      const u32* list = lists + static_cast<size_t>(c) * NI;
      if (cap <= 512)
        RVN_KLAUNCH(kKSegSortPos, seg_sort_pos_lds_kernel<64><<<n_cls[c], 64, seg_sort_lds_bytes<64, false>(cap), s>>>(
                                      p0, iv_begin, iv_end, list, n_cls[c], cap));
      else
        RVN_KLAUNCH(kKSegSortPos, seg_sort_pos_lds_kernel<256><<<n_cls[c], 256, seg_sort_lds_bytes<256, false>(cap), s>>>(
                                      p0, iv_begin, iv_end, list, n_cls[c], cap));
    }
    u32* lis_min = e.lis_min.get<u32>(H + NI + 1);
    u32* lis_pred = e.lis_pred.get<u32>(H + 1);
    u64* lis_tail = e.lis_tail.get<u64>(H + NI + 1);
    u64* lis_mask = e.lis_mask.get<u64>((H >> 6) + NI + 2);
    RVN_HIP(hipMemsetAsync(slot_flags, 0, n_slots + 1, s));
    u64* anchors = nullptr;
    u64* slot_aoff = nullptr;
    u32* slot_acnt = nullptr;
    if (e.keep_anchors) {
      anchors = out.anchors.get<u64>(H + 1);
      slot_aoff = e.anc_slot_off.get<u64>(n_slots + 1);
      slot_acnt = e.anc_slot_cnt.get<u32>(n_slots + 1);
    }
    RVN_KLAUNCH(kKChainSmall, chain_small_kernel<<<div_up(NI, 64), 64, 0, s>>>(
                                  g0, p0, iv_begin, iv_end, iv_read, NI, r.id.as<u32>(), first, e.k, e.chain, e.matches,
                                  e.gap, slot_div, slots, slot_flags, anchors, slot_aoff, slot_acnt));
    {
      // largest classes first: their waves run longest
      for (int c = kChainClasses; c >= 0; --c) {
        if (!n_cls[c]) continue;
        const u32 cap = c < kChainClasses ? kChainClassCapHost[c] : 0u;
        const size_t lds = cap ? chain_class_lds(cap) : 0;
        RVN_KLAUNCH(kKChain, chain_kernel<<<n_cls[c], 64, lds, s>>>(g0, p0, iv_begin, iv_end, iv_read,
                                                                   lists + static_cast<size_t>(c) * NI, n_cls[c], cap,
                                                                   r.id.as<u32>(), first, e.k, e.chain, e.matches, e.gap,
                                                                   slot_div, lis_tail, lis_min, lis_pred, lis_mask, slots,
                                                                   slot_flags, anchors, slot_aoff, slot_acnt));
      }
    }
    t.stop();
  }
  {
    StageTimer t(e, StageTimes::kCompact);
    u32* scan = e.ovl_scan.get<u32>(n_slots + 2);
    exclusive_scan_u8_u32(slot_flags, scan, n_slots, e.scan_tmp, s);
    const u32 O = static_cast<u32>(read_back(e, scan + n_slots, 4));
    out.n_overlaps = O;
    e.c_overlaps += O;
    Overlap* ovl = out.ovl.get<Overlap>(static_cast<size_t>(O) + 1);
    RVN_KLAUNCH(kKCompactOverlaps, compact_overlaps_kernel<<<div_up(n_slots, 256), 256, 0, s>>>(slots, slot_flags, scan, n_slots, ovl));
    out.has_anchors = e.keep_anchors;
    if (e.keep_anchors) {
      u64* aoff = out.anchor_off.get<u64>(static_cast<size_t>(O) + 1);
      u32* acnt = out.anchor_cnt.get<u32>(static_cast<size_t>(O) + 1);
      RVN_KLAUNCH(kKCompactOverlaps, compact_aux_kernel<<<div_up(n_slots, 256), 256, 0, s>>>(
                                         e.anc_slot_off.as<u64>(), e.anc_slot_cnt.as<u32>(), slot_flags, scan, n_slots,
                                         aoff, acnt));
    }
    RVN_KLAUNCH(kKGather, read_ovl_off_kernel<<<div_up(nr + 1, 256), 256, 0, s>>>(seg_off, scan, slot_div, nr + 1, ovl_read_off));
    t.stop();
  }
}

namespace {

template <typename V>
void map_batch_impl(Engine& e, const ReadsDev& r, u32 first, u32 last, bool avoid_equal, bool avoid_symmetric,
                    bool minhash, bool want_filtered, MapOut& out) {
  hipStream_t s = e.stream;
  Index& ix = e.index;
  const u32 nr = last - first;
  out.first = first;
  out.last = last;
  out.n_query = out.n_matches = out.n_intervals = out.n_overlaps = 0;
  u32* ovl_read_off = out.ovl_read_off.get<u32>(static_cast<size_t>(nr) + 1);

  u64* seg_off = e.seg_off.get<u64>(static_cast<size_t>(nr) + 2);
  u64 H = 0;
  // self-join path: queries == indexed reads, flags in the index, bounded run lengths, ids == read indices
  const bool join = minhash && !want_filtered && ix.m != 0 && ix.first == first && ix.last == last &&
                    (ix.has_query_flags || ix.all_query) && ix.occurrence <= 4096 && r.ids_are_indices;
  if (join) {
    StageTimer t(e, StageTimes::kMatch);
    e.query_ready = false;
    for (u32 i = first; i < last; ++i) e.c_query_bases += r.h_len[i];
    out.n_query = ix.all_query ? ix.m : e.join_query_count;
    e.c_query_min += out.n_query;
    u32* read_cnt = e.q_cnt.get<u32>(2 * (static_cast<size_t>(nr) + 1));
    u32* cursor = read_cnt + nr + 1;
    RVN_HIP(hipMemsetAsync(read_cnt, 0, 2 * (static_cast<size_t>(nr) + 1) * 4, s));
    const u32 n_runs = static_cast<u32>(ix.u);
    const u64* sorg = ix.s_org[ix.cur].as<u64>();
    RVN_KLAUNCH(kKJoinCount, join_kernel<false><<<div_up(n_runs, 256), 256, 0, s>>>(
                                 ix.u_start.as<u32>(), n_runs, sorg, ix.occurrence, ix.all_query, avoid_equal,
                                 avoid_symmetric, r.h_id.empty() ? 0 : r.h_id[first], 0u, 0xFFFFFFFFu, read_cnt, nullptr, nullptr,
                                 nullptr, nullptr));
    exclusive_scan_u32_u64(read_cnt, seg_off, nr, e.scan_tmp, s);
    H = read_back(e, seg_off + nr, 8);
    out.n_matches = H;
    e.c_matches += H;
    if (H) {
      u64* g0 = e.m_grp[0].get<u64>(H + 1);
      u64* p0 = e.m_pos[0].get<u64>(H + 1);
      e.m_grp[1].reserve((H + 1) * 8);
      e.m_pos[1].reserve((H + 1) * 8);
      RVN_KLAUNCH(kKJoinEmit, join_kernel<true><<<div_up(n_runs, 256), 256, 0, s>>>(
                                  ix.u_start.as<u32>(), n_runs, sorg, ix.occurrence, ix.all_query, avoid_equal,
                                  avoid_symmetric, r.h_id[first], 0u, 0xFFFFFFFFu, nullptr, seg_off, cursor, g0, p0));
    }
    t.stop();
  }
  Sketch& qs = e.query_sketch;
  if (!join) {
    index_build_table(e);
  {
    StageTimer t(e, StageTimes::kQuery);
    const bool ready = e.query_ready && e.query_ready_first == first && e.query_ready_last == last &&
                       e.query_ready_minhash == minhash;
    e.query_ready = false;
    if (!ready) sketch_range(e, r, first, last, minhash, qs);
    t.stop();
  }
  const u64 nq = qs.count;
  out.n_query = nq;
  for (u32 i = first; i < last; ++i) e.c_query_bases += r.h_len[i];
  e.c_query_min += nq;
  if (nq == 0 || ix.m == 0) {
    RVN_HIP(hipMemsetAsync(ovl_read_off, 0, (static_cast<size_t>(nr) + 1) * 4, s));
    if (want_filtered) {
      u8* f = out.filtered.get<u8>(nq + 1);
      RVN_HIP(hipMemsetAsync(f, 0, nq + 1, s));
    }
    return;
  }
  {
    StageTimer t(e, StageTimes::kMatch);
    u32* q_start = e.q_start.get<u32>(nq + 1);
    u32* q_n = e.tmp_a.get<u32>(nq + 1);
    u32* q_cnt = e.q_cnt.get<u32>(nq + 1);
    u8* filt = want_filtered ? out.filtered.get<u8>(nq + 1) : nullptr;
    u64* m_off = e.m_off.get<u64>(nq + 2);
    RVN_KLAUNCH(kKMatchCount, match_count_kernel<V><<<div_up(nq, 256), 256, 0, s>>>(
        qs.val.as<V>(), qs.org.as<u64>(), nq, ix.u_val.as<V>(), ix.u_start.as<u32>(), ix.table.as<u32>(), ix.shift,
        static_cast<u32>(ix.u), ix.s_org[ix.cur].as<u64>(), ix.occurrence, avoid_equal, avoid_symmetric, q_start, q_n,
        q_cnt, filt, ix.direct_built ? ix.direct.as<u64>() : nullptr));
    exclusive_scan_u32_u64(q_cnt, m_off, nq, e.scan_tmp, s);
    H = read_back(e, m_off + nq, 8);
    out.n_matches = H;
    e.c_matches += H;
    if (H) {
      u64* g0 = e.m_grp[0].get<u64>(H + 1);
      u64* p0 = e.m_pos[0].get<u64>(H + 1);
      e.m_grp[1].reserve((H + 1) * 8);
      e.m_pos[1].reserve((H + 1) * 8);
      RVN_KLAUNCH(kKMatchEmit, match_emit_kernel<<<div_up(nq, 256), 256, 0, s>>>(qs.org.as<u64>(), nq, ix.s_org[ix.cur].as<u64>(), q_start,
                                                        q_n, m_off, avoid_equal, avoid_symmetric, g0, p0));
    }
    if (H) {
      RVN_KLAUNCH(kKGather, gather_u64_by_u32_kernel<<<div_up(nr + 1, 256), 256, 0, s>>>(
                                e.m_off.as<u64>(), qs.read_off.as<u32>(), seg_off, nr + 1));
    }
    t.stop();
  }
  }  // !join
  out.n_matches = H;
  chain_matches(e, r, first, last, H, out);
}

}  // namespace

// Self-join of the whole (shard of the) index for query read ids 0..n_reads-1 (ids are global read indices):
// matches land in e.m_grp[0] / e.m_pos[0], segmented by query id through e.seg_off[n_reads + 1].  The hash-owner
// side of the sharded pass: the owner of a hash class joins its runs and ships every read's matches
// to the GPU that owns the read.  Returns the number of matches.
u64 join_index_matches(Engine& e, u32 n_reads, bool avoid_equal, bool avoid_symmetric, u32 q_lo, u32 q_hi) {
  hipStream_t s = e.stream;
  Index& ix = e.index;
  u64* seg_off = e.seg_off.get<u64>(static_cast<size_t>(n_reads) + 2);
  RVN_HIP(hipMemsetAsync(seg_off, 0, (static_cast<size_t>(n_reads) + 2) * 8, s));
  if (ix.m == 0 || ix.u == 0) return 0;
  if (!(ix.has_query_flags || ix.all_query)) throw std::invalid_argument("[raven_hip] join: index has no query flags");
  StageTimer t(e, StageTimes::kMatch);
  u32* read_cnt = e.q_cnt.get<u32>(2 * (static_cast<size_t>(n_reads) + 1));
  u32* cursor = read_cnt + n_reads + 1;
  RVN_HIP(hipMemsetAsync(read_cnt, 0, 2 * (static_cast<size_t>(n_reads) + 1) * 4, s));
  const u32 n_runs = static_cast<u32>(ix.u);
  const u64* sorg = ix.s_org[ix.cur].as<u64>();
  RVN_KLAUNCH(kKJoinCount, join_kernel<false><<<div_up(n_runs, 256), 256, 0, s>>>(
                               ix.u_start.as<u32>(), n_runs, sorg, ix.occurrence, ix.all_query, avoid_equal,
                               avoid_symmetric, 0, q_lo, q_hi, read_cnt, nullptr, nullptr, nullptr, nullptr));
  exclusive_scan_u32_u64(read_cnt, seg_off, n_reads, e.scan_tmp, s);
  const u64 H = read_back(e, seg_off + n_reads, 8);
  e.c_matches += H;
  u64* g0 = e.m_grp[0].get<u64>(H + 1);
  u64* p0 = e.m_pos[0].get<u64>(H + 1);
  if (H)
    RVN_KLAUNCH(kKJoinEmit, join_kernel<true><<<div_up(n_runs, 256), 256, 0, s>>>(
                                ix.u_start.as<u32>(), n_runs, sorg, ix.occurrence, ix.all_query, avoid_equal,
                                avoid_symmetric, 0, q_lo, q_hi, nullptr, seg_off, cursor, g0, p0));
  t.stop();
  return H;
}

// Map() of the reads [first, last) against an index in which their OWN minimizers sit as query-only entries (kQueryFlag |
// kForeignFlag, ahead of the members of every run: the polishing round's mapping, polish.hip).  One streaming pass over the
// runs instead of one random probe per query minimizer — what the first pass does when the queries are the indexed reads
// (join_kernel above), extended to queries that are not members.  n_query: the query-only entries (statistics).
void map_batch_query_only(Engine& e, const ReadsDev& r, u32 first, u32 last, u64 n_query, MapOut& out) {
  hipStream_t s = e.stream;
  Index& ix = e.index;
  const u32 nr = last - first;
  out.first = first;
  out.last = last;
  out.n_query = n_query;
  out.n_matches = out.n_intervals = out.n_overlaps = 0;
  (void)out.ovl_read_off.get<u32>(static_cast<size_t>(nr) + 1);
  u64* seg_off = e.seg_off.get<u64>(static_cast<size_t>(nr) + 2);
  u64 H = 0;
  {
    StageTimer t(e, StageTimes::kMatch);
    e.query_ready = false;
    for (u32 i = first; i < last; ++i) e.c_query_bases += r.h_len[i];
    e.c_query_min += n_query;
    u32* read_cnt = e.q_cnt.get<u32>(2 * (static_cast<size_t>(nr) + 1));
    u32* cursor = read_cnt + nr + 1;
    RVN_HIP(hipMemsetAsync(read_cnt, 0, 2 * (static_cast<size_t>(nr) + 1) * 4, s));
    RVN_HIP(hipMemsetAsync(seg_off, 0, (static_cast<size_t>(nr) + 2) * 8, s));
    const u32 n_runs = static_cast<u32>(ix.u);
    const u64* sorg = ix.s_org[ix.cur].as<u64>();
    if (n_runs) {
      RVN_KLAUNCH(kKJoinCount, join_kernel<false><<<div_up(n_runs, 256), 256, 0, s>>>(
                                   ix.u_start.as<u32>(), n_runs, sorg, ix.occurrence, 0, 0, 0, r.h_id[first], 0u, 0xFFFFFFFFu,
                                   read_cnt, nullptr, nullptr, nullptr, nullptr));
      exclusive_scan_u32_u64(read_cnt, seg_off, nr, e.scan_tmp, s);
      H = read_back(e, seg_off + nr, 8);
    }
    e.c_matches += H;
    if (H) {
      u64* g0 = e.m_grp[0].get<u64>(H + 1);
      u64* p0 = e.m_pos[0].get<u64>(H + 1);
      e.m_grp[1].reserve((H + 1) * 8);
      e.m_pos[1].reserve((H + 1) * 8);
      RVN_KLAUNCH(kKJoinEmit, join_kernel<true><<<div_up(n_runs, 256), 256, 0, s>>>(
                                  ix.u_start.as<u32>(), n_runs, sorg, ix.occurrence, 0, 0, 0, r.h_id[first], 0u, 0xFFFFFFFFu,
                                  nullptr, seg_off, cursor, g0, p0));
    }
    t.stop();
  }
  out.n_matches = H;
  chain_matches(e, r, first, last, H, out);
}

void map_batch(Engine& e, const ReadsDev& r, u32 first, u32 last, bool avoid_equal, bool avoid_symmetric,
               bool minhash, bool want_filtered, MapOut& out) {
  if (e.val64) map_batch_impl<u64>(e, r, first, last, avoid_equal, avoid_symmetric, minhash, want_filtered, out);
  else map_batch_impl<u32>(e, r, first, last, avoid_equal, avoid_symmetric, minhash, want_filtered, out);
}

}  // namespace rvn
