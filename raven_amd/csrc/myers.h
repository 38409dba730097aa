// myers.h — Myers' bit-vector block update and the packed-sequence accessors shared by the edit-distance kernel
// (edit_distance.hip: edlibAlign distance, construct.cc:190-199) and the alignment-path kernels (nwpath.hip: the
// edlib NW path racon takes its window breakpoints from).  Everything is __host__ __device__ so that the exact code
// the kernels run can be driven lane by lane on the CPU (tests/test_nwpath.py through rvn_test_nw_breakpoints).
#pragma once

#include "common.h"

namespace rvn {

#if defined(__HIP_DEVICE_COMPILE__)
#define RVN_POPC64(x) __popcll(x)
#else
#define RVN_POPC64(x) __builtin_popcountll(x)
#endif

// edlib calculateBlock: advance one 64-row block by one column. hin/hout in {-1, 0, +1}.
__host__ __device__ __forceinline__ int myers_block(u64& Pv, u64& Mv, u64 Eq, int hin) {
  const u64 Xv = Eq | Mv;
  if (hin < 0) Eq |= 1ULL;
  const u64 Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
  u64 Ph = Mv | ~(Xh | Pv);
  u64 Mh = Pv & Xh;
  const int hout = static_cast<int>(Ph >> 63) - static_cast<int>(Mh >> 63);
  Ph <<= 1;
  Mh <<= 1;
  if (hin < 0) Mh |= 1ULL;
  else if (hin > 0) Ph |= 1ULL;
  Pv = Mh | ~(Xv | Ph);
  Mv = Ph & Xv;
  return hout;
}

// 64 bits of the 2-bit stream starting at base index `base` (bases base .. base+31)
__host__ __device__ __forceinline__ u64 load_bases32(const u64* __restrict__ words, u64 base) {
  const u64 bit = base * 2;
  const u64 wi = bit >> 6;
  const unsigned off = static_cast<unsigned>(bit & 63);
  u64 x = words[wi] >> off;
  if (off) x |= words[wi + 1] << (64 - off);
  return x;
}

// even bits of x (bit 2i -> bit i), 32 result bits
__host__ __device__ __forceinline__ u64 compress_even(u64 x) {
  x &= 0x5555555555555555ULL;
  x = (x | (x >> 1)) & 0x3333333333333333ULL;
  x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0FULL;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFULL;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFULL;
  x = (x | (x >> 16)) & 0x00000000FFFFFFFFULL;
  return x;
}

// Peq masks of pattern block b (rows 64b .. 64b+63 of the span starting at a_base, n rows in total)
__host__ __device__ __forceinline__ void load_peq(const u64* __restrict__ words, u64 a_base, u32 n, u32 b, u64 (&peq)[4]) {
  const u32 row0 = b * 64;
  u64 lo = 0, hi = 0;
  if (row0 < n) lo = load_bases32(words, a_base + row0);
  if (row0 + 32 < n) hi = load_bases32(words, a_base + row0 + 32);
  const u32 valid = n > row0 ? (n - row0 >= 64 ? 64u : n - row0) : 0u;
  const u64 vmask = valid >= 64 ? ~0ULL : ((1ULL << valid) - 1ULL);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const u64 rep = 0x5555555555555555ULL * static_cast<u64>(c);
    const u64 ml = lo ^ rep, mh = hi ^ rep;
    const u64 el = compress_even(~(ml | (ml >> 1)));
    const u64 eh = compress_even(~(mh | (mh >> 1)));
    peq[c] = (el | (eh << 32)) & vmask;
  }
}

// Bit planes of pattern block b: bit i of lo / hi = low / high bit of the code of row 64b + i; rows beyond n are marked in
// `pad` (they never match).  The match mask of symbol c is then 4 bit operations instead of a 4-entry table of 64-bit
// words per block: eq(c) = (c & 1 ? lo : ~lo) & (c & 2 ? hi : ~hi) & ~pad.
struct BlockPlanes {
  u64 lo, hi, valid;
};
__host__ __device__ __forceinline__ BlockPlanes load_planes(const u64* __restrict__ words, u64 a_base, u32 n, u32 b) {
  const u32 row0 = b * 64;
  u64 w0 = 0, w1 = 0;
  if (row0 < n) w0 = load_bases32(words, a_base + row0);
  if (row0 + 32 < n) w1 = load_bases32(words, a_base + row0 + 32);
  const u32 valid = n > row0 ? (n - row0 >= 64 ? 64u : n - row0) : 0u;
  BlockPlanes p;
  p.lo = compress_even(w0) | (compress_even(w1) << 32);
  p.hi = compress_even(w0 >> 1) | (compress_even(w1 >> 1) << 32);
  p.valid = valid >= 64 ? ~0ULL : ((1ULL << valid) - 1ULL);
  return p;
}
__host__ __device__ __forceinline__ u64 planes_eq(const BlockPlanes& p, unsigned c) {
  const u64 a = (c & 1u) ? p.lo : ~p.lo;
  const u64 b = (c & 2u) ? p.hi : ~p.hi;
  return a & b & p.valid;
}

// text symbol of column j (1-based) with a one-word look-ahead so the load latency is off the critical path
struct TextCursor {
  const u64* words;
  long long first;  // base index of column 1 (forward) / of column 1 in the rc direction
  bool rc;
  long long widx;   // word index currently held
  u64 w_cur, w_next;
  __host__ __device__ __forceinline__ void init(const u64* w, u64 b_base, u32 m, bool rc_, long long j) {
    words = w;
    rc = rc_;
    first = rc_ ? static_cast<long long>(b_base) + m - 1 : static_cast<long long>(b_base);
    const long long pos = rc ? first - (j - 1) : first + (j - 1);
    widx = pos >> 5;
    w_cur = words[widx];
    w_next = words[rc ? (widx > 0 ? widx - 1 : 0) : widx + 1];
  }
  __host__ __device__ __forceinline__ unsigned get(long long j) {
    const long long pos = rc ? first - (j - 1) : first + (j - 1);
    const long long wi = pos >> 5;
    if (wi != widx) {  // crossed into the neighbouring word: rotate, prefetch the one after
      widx = wi;
      w_cur = w_next;
      w_next = words[rc ? (wi > 0 ? wi - 1 : 0) : wi + 1];
    }
    const unsigned c = static_cast<unsigned>(w_cur >> ((pos & 31) * 2)) & 3u;
    return rc ? 3u - c : c;
  }
};

// 2-bit code of base `i` of a packed sequence
__host__ __device__ __forceinline__ u32 packed_code(const u64* __restrict__ words, u64 i) {
  return static_cast<u32>(words[i >> 5] >> ((i & 31) << 1)) & 3u;
}

}  // namespace rvn
