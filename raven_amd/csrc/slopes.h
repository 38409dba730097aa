// slopes.h — raven::Pile::FindChimericRegions (RavenLib/src/pile.cc:176-187) = Pile::FindSlopes(1.82)
// (pile.cc:403-600) + pairing of consecutive (down, up) slopes + Pile::MergeRegions (pile.cc:373-400), on one pile's
// coverage array, as __host__ __device__ code: one thread per pile on the device (pile.hip), the same function behind
// rvn_test_find_chimeric_regions on the CPU.
//
// A "slope" is a run of cells whose coverage, multiplied by q, is still below the highest coverage within w = 52 cells
// on their left (a DOWN slope: coverage drops) or on their right (an UP slope); a chimeric read shows a coverage pit, i.e.
// a down slope followed by an up slope.  The reference finds the windowed maxima with monotone deques; here they are
// plain windowed scans of the (short, L1-resident) coverage array — only the maxima matter, not how they are kept.
// All comparisons are the reference's: coverage * q in double, clamped at 65535, compared either as a truncated uint16
// (first sweep) or as the double itself (later sweeps), exactly where the reference does each.
#pragma once

#include "common.h"

namespace rvn {

struct SlopeRegion {
  u32 first;   // cell << 1 | type (0 = down, 1 = up)
  u32 second;  // last cell
};

__host__ __device__ inline double slope_clamp(double v) { return v < 65535.0 ? v : 65535.0; }
__host__ __device__ inline bool region_less(const SlopeRegion& a, const SlopeRegion& b) {
  return a.first < b.first || (a.first == b.first && a.second < b.second);
}
__host__ __device__ inline void regions_sort(SlopeRegion* r, u32 n) {  // std::sort on pairs; equal pairs are interchangeable
  for (u32 i = 1; i < n; ++i) {
    const SlopeRegion key = r[i];
    u32 j = i;
    while (j > 0 && region_less(key, r[j - 1])) {
      r[j] = r[j - 1];
      --j;
    }
    r[j] = key;
  }
}

// Appends the runs of flagged cells of one sweep: consecutive flagged cells (at most one unflagged cell apart is NOT
// enough: the reference starts a new run when j - last > 1) form one region of the given type.
struct RunBuilder {
  SlopeRegion* dst;
  u32 cap;
  u32* n;
  bool* overflow;
  u32 type;
  bool found = false;
  u32 first = 0, last = 0;
  __host__ __device__ void push(u32 a, u32 b) {
    if (*n < cap) dst[(*n)++] = SlopeRegion{a << 1 | type, b};
    else *overflow = true;
  }
  __host__ __device__ void hit(u32 j) {
    if (found) {
      if (j - last > 1) {
        push(first, last);
        first = j;
      }
    } else {
      found = true;
      first = j;
    }
    last = j;
  }
  __host__ __device__ void finish() {
    if (found) push(first, last);
  }
};

__host__ __device__ inline u32 find_slopes_rest(const u16* __restrict__ data, double q, SlopeRegion* dst, u32 cap, u32 n,
                                                u16* __restrict__ tmp, bool* overflow);

// Pile::FindSlopes(q) on data[0, size); dst has room for cap regions; tmp: size u16 cells of scratch.
// Returns the number of regions (sorted as the reference leaves them); *overflow when cap was too small.
__host__ __device__ inline u32 find_slopes(const u16* __restrict__ data, int size, double q, SlopeRegion* dst, u32 cap,
                                           u16* __restrict__ tmp, bool* overflow) {
  const int w = 847 >> 4;
  u32 n = 0;
  *overflow = false;
  if (size <= 0) return 0;
  {  // first sweep over every cell
    RunBuilder down{dst, cap, &n, overflow, 0u}, up{dst, cap, &n, overflow, 1u};
    // The reference interleaves the two kinds in discovery order and sorts afterwards; the order of discovery does not
    // matter for a sorted list of distinct runs, so downs and ups are collected in two passes.
    for (int i = 0; i < size; ++i) {
      const u16 d = static_cast<u16>(slope_clamp(static_cast<double>(data[i]) * q));
      if (i != 0) {
        u16 lmax = 0;
        const int lo = i - w < 0 ? 0 : i - w;
        for (int x = lo; x < i; ++x) lmax = data[x] > lmax ? data[x] : lmax;
        if (lmax > d) down.hit(static_cast<u32>(i));
      }
    }
    down.finish();
    for (int i = 0; i < size; ++i) {
      const u16 d = static_cast<u16>(slope_clamp(static_cast<double>(data[i]) * q));
      if (i != size - 1) {
        u16 rmax = 0;
        const int hi = i + w > size - 1 ? size - 1 : i + w;
        for (int x = i + 1; x <= hi; ++x) rmax = data[x] > rmax ? data[x] : rmax;
        if (rmax > d) up.hit(static_cast<u32>(i));
      }
    }
    up.finish();
  }
  return find_slopes_rest(data, q, dst, cap, n, tmp, overflow);
}

// The part of Pile::FindSlopes after the first sweep: dst[0, n) holds the runs of the first sweep (any order).
__host__ __device__ inline u32 find_slopes_rest(const u16* __restrict__ data, double q, SlopeRegion* dst, u32 cap, u32 n,
                                                u16* __restrict__ tmp, bool* overflow) {
  const int w = 847 >> 4;
  if (n == 0) return 0;
  // separate overlapping slopes
  for (;;) {
    regions_sort(dst, n);
    bool changed = false;
    for (u32 i = 0; i + 1 < n; ++i) {
      if (dst[i].second < (dst[i + 1].first >> 1)) continue;
      if (dst[i].first & 1u) {  // an up slope reaching into the next region: re-evaluate its cells against the maximum
        const u32 sb = dst[i].first >> 1;                                           // of the cells after them only
        const u32 se = dst[i].second < dst[i + 1].second ? dst[i].second : dst[i + 1].second;
        if (se > sb) {  // suffix maxima of (j, se]
          u16 m = 0;
          for (u32 j = se; j > sb; --j) {
            m = data[j] > m ? data[j] : m;
            tmp[j - 1] = m;  // tmp[j'] = max(data[j' + 1 .. se])
          }
        }
        RunBuilder up{dst, cap, &n, overflow, 1u};
        for (u32 j = sb; j < se; ++j)
          if (slope_clamp(static_cast<double>(data[j]) * q) < static_cast<double>(tmp[j])) up.hit(j);
        up.finish();
        dst[i].first = se << 1 | 1u;
      } else {
        if (dst[i].second == (dst[i + 1].first >> 1)) continue;
        const u32 a = dst[i].first >> 1, b = dst[i + 1].first >> 1;
        const u32 sb = a > b ? a : b;
        const u32 se = dst[i].second;
        RunBuilder down{dst, cap, &n, overflow, 0u};
        bool have = false;
        u16 pmax = 0;  // running maximum of data[sb .. j-1]
        for (u32 j = sb; j <= se; ++j) {
          if (have && slope_clamp(static_cast<double>(data[j]) * q) < static_cast<double>(pmax)) down.hit(j);
          pmax = (!have || data[j] >= pmax) ? data[j] : pmax;
          have = true;
        }
        down.finish();
        dst[i].second = sb;
      }
      changed = true;
      break;
    }
    if (!changed || *overflow) break;
  }
  // narrow the slopes around a short plateau between an up and a down slope
  for (u32 i = 0; i + 1 < n; ++i) {
    if ((dst[i].first & 1u) && !(dst[i + 1].first & 1u)) {
      const u32 sb = dst[i].second, se = dst[i + 1].first >> 1;
      if (se - sb > static_cast<u32>(w)) continue;
      u16 max_cov = 0;
      for (u32 j = sb + 1; j < se; ++j) max_cov = data[j] > max_cov ? data[j] : max_cov;
      u32 vp = dst[i].first >> 1;
      for (u32 j = dst[i].first >> 1; j <= sb; ++j)
        if (static_cast<double>(max_cov) > slope_clamp(static_cast<double>(data[j]) * q)) vp = j;
      dst[i].second = vp;
      vp = dst[i + 1].second;
      for (u32 j = se; j <= dst[i + 1].second; ++j) {
        if (static_cast<double>(max_cov) > slope_clamp(static_cast<double>(data[j]) * q)) {
          vp = j;
          break;
        }
      }
      dst[i + 1].first = vp << 1 | 0u;
    }
  }
  return n;
}

__host__ __device__ inline u32 pair_and_merge_slopes(SlopeRegion* slopes, u32 ns, u16* __restrict__ tmp, u32* __restrict__ out,
                                                     u32 out_cap, bool* overflow);

// Pile::FindChimericRegions: out[2 * r], out[2 * r + 1] = the merged (begin, end) cells of region r; returns their number.
// slopes / tmp: scratch (cap regions / size cells); merged flags reuse tmp.
__host__ __device__ inline u32 find_chimeric_regions(const u16* __restrict__ data, int size, SlopeRegion* slopes, u32 cap,
                                                     u16* __restrict__ tmp, u32* __restrict__ out, u32 out_cap,
                                                     bool* overflow) {
  const u32 ns = find_slopes(data, size, 1.82, slopes, cap, tmp, overflow);
  return pair_and_merge_slopes(slopes, ns, tmp, out, out_cap, overflow);
}

// pit pairing + Pile::MergeRegions on the final slope list
__host__ __device__ inline u32 pair_and_merge_slopes(SlopeRegion* slopes, u32 ns, u16* __restrict__ tmp, u32* __restrict__ out,
                                                     u32 out_cap, bool* overflow) {
  if (ns == 0 || *overflow) return 0;
  // a down slope directly followed by an up slope: the pit between them (pile.cc:181-186); collected in place
  u32 nr = 0;
  for (u32 i = 0; i + 1 < ns; ++i) {
    if (!(slopes[i].first & 1u) && (slopes[i + 1].first & 1u)) {
      const SlopeRegion r{slopes[i].first >> 1, slopes[i + 1].second};
      slopes[nr++] = r;  // nr <= i: never overwrites an unread entry
    }
  }
  // Pile::MergeRegions: every not yet merged region absorbs, repeatedly, all later regions it overlaps
  for (u32 i = 0; i < nr; ++i) tmp[i] = 0;  // is_merged (nr <= ns / 2 <= cap / 2 <= size is guaranteed by the caller)
  u32 no = 0;
  for (u32 i = 0; i < nr; ++i) {
    if (tmp[i]) continue;
    SlopeRegion r = slopes[i];
    for (;;) {
      bool grew = false;
      for (u32 j = i + 1; j < nr; ++j) {
        if (tmp[j]) continue;
        if (r.first < slopes[j].second && r.second > slopes[j].first) {
          grew = true;
          tmp[j] = 1;
          r.first = r.first < slopes[j].first ? r.first : slopes[j].first;
          r.second = r.second > slopes[j].second ? r.second : slopes[j].second;
        }
      }
      if (!grew) break;
    }
    if (no < out_cap) {
      out[2 * no] = r.first;
      out[2 * no + 1] = r.second;
      ++no;
    } else {
      *overflow = true;
    }
  }
  return no;
}

}  // namespace rvn
