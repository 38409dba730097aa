// overlap_rules.h — an overlap against the valid regions of its two piles: the per-overlap integer rules of
// RavenLib/src/overlap_utils.cc (OverlapUpdate :14-80, GetOverlapType :82-113, OverlapFinalize :115-133) as
// __host__ __device__ functions for the second mapping pass and the identity filters (pass2.hip).
// All arithmetic is the reference's 32-bit unsigned arithmetic (differences may wrap exactly as they do there).
#pragma once

#include "common.h"

namespace rvn {

struct PileRegion {  // Pile::begin(), Pile::end() in bases and Pile::is_invalid() (pile.h:37-45)
  u32 begin, end;
  u32 invalid;
};

// how far [b, e) sticks out of the valid region [rb, re) at its head / tail
__host__ __device__ inline u32 stick_out_head(u32 b, u32 rb) { return b < rb ? rb - b : 0u; }
__host__ __device__ inline u32 stick_out_tail(u32 e, u32 re) { return e > re ? e - re : 0u; }

// OverlapUpdate: clips the overlap to the two valid regions (what one read loses outside its region, the other loses
// at the matching end — which end depends on the strand); false = dropped (an invalid pile, nothing left inside the
// regions, or fewer than 84 bases on a side).  Updates `o` only when it returns true.
__host__ __device__ inline bool overlap_update(Overlap& o, const PileRegion& L, const PileRegion& R) {
  if (L.invalid || R.invalid) return false;
  if (o.lhs_begin >= L.end || o.lhs_end <= L.begin || o.rhs_begin >= R.end || o.rhs_end <= R.begin) return false;
  const u32 r_head = stick_out_head(o.rhs_begin, R.begin), r_tail = stick_out_tail(o.rhs_end, R.end);
  const u32 l_head = stick_out_head(o.lhs_begin, L.begin), l_tail = stick_out_tail(o.lhs_end, L.end);
  u32 lb = o.lhs_begin + (o.strand ? r_head : r_tail);
  u32 le = o.lhs_end - (o.strand ? r_tail : r_head);
  u32 rb = o.rhs_begin + (o.strand ? l_head : l_tail);
  u32 re = o.rhs_end - (o.strand ? l_tail : l_head);
  if (lb >= L.end || le <= L.begin || rb >= R.end || re <= R.begin) return false;
  lb = lb > L.begin ? lb : L.begin;
  le = le < L.end ? le : L.end;
  rb = rb > R.begin ? rb : R.begin;
  re = re < R.end ? re : R.end;
  if (lb >= le || le - lb < 84u || rb >= re || re - rb < 84u) return false;
  o.lhs_begin = lb;
  o.lhs_end = le;
  o.rhs_begin = rb;
  o.rhs_end = re;
  return true;
}

// GetOverlapType: 0 internal, 1 lhs contained, 2 rhs contained, 3 lhs -> rhs, 4 rhs -> lhs
__host__ __device__ inline u32 overlap_type(const Overlap& o, const PileRegion& L, const PileRegion& R) {
  const u32 l_len = L.end - L.begin;
  const u32 lb = o.lhs_begin - L.begin, le = o.lhs_end - L.begin;
  const u32 r_len = R.end - R.begin;
  const u32 rb = o.strand ? o.rhs_begin - R.begin : r_len - (o.rhs_end - R.begin);
  const u32 re = o.strand ? o.rhs_end - R.begin : r_len - (o.rhs_begin - R.begin);
  const u32 head = lb < rb ? lb : rb;
  const u32 l_rest = l_len - le, r_rest = r_len - re;
  const u32 overhang = head + (l_rest < r_rest ? l_rest : r_rest);
  // the reference compares uint32 < uint32 * 0.875 in double
  if (static_cast<double>(le - lb) < static_cast<double>(le - lb + overhang) * 0.875 ||
      static_cast<double>(re - rb) < static_cast<double>(re - rb + overhang) * 0.875)
    return 0;
  if (lb <= rb && l_rest <= r_rest) return 1;
  if (rb <= lb && r_rest <= l_rest) return 2;
  return lb > rb ? 3u : 4u;
}

// OverlapFinalize: type into `score`, coordinates relative to the valid regions (rhs mirrored on the opposite strand)
__host__ __device__ inline bool overlap_finalize(Overlap& o, const PileRegion& L, const PileRegion& R) {
  o.score = overlap_type(o, L, R);
  if (o.score < 3) return false;
  o.lhs_begin -= L.begin;
  o.lhs_end -= L.begin;
  o.rhs_begin -= R.begin;
  o.rhs_end -= R.begin;
  if (!o.strand) {
    const u32 rb = o.rhs_begin, r_len = R.end - R.begin;
    o.rhs_begin = r_len - o.rhs_end;
    o.rhs_end = r_len - rb;
  }
  return true;
}

__host__ __device__ inline u32 overlap_length(const Overlap& o) {  // GetOverlapLength (overlap_utils.cc:10-12)
  const u32 a = o.rhs_end - o.rhs_begin, b = o.lhs_end - o.lhs_begin;
  return a > b ? a : b;
}

}  // namespace rvn
