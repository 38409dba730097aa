// introsort.h — exact restatement of libstdc++'s std::sort (bits/stl_algo.h, GCC 11: __introsort_loop,
// __unguarded_partition_pivot, __move_median_to_first, __final_insertion_sort, heap fallback from
// bits/stl_heap.h) as a __host__ __device__ template.
//
// Why: RavenLib/src/construct.cc:98-107 truncates each pile's overlap list with the *unstable*
// std::sort(GetOverlapLength desc) + "keep first kMax".  Which equal-length overlaps survive, and
// in what order, is defined only by this algorithm; to stay bit-exact on the GPU the device runs the
// same sequence of comparisons and moves.  tests/test_hostdev.py checks it against std::sort itself.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rvn {

// Elements are u64 with the sort key in the HIGH 32 bits (descending) and a payload in the low 32.
struct LenDesc {
  __host__ __device__ __forceinline__ bool operator()(std::uint64_t a, std::uint64_t b) const {
    return (a >> 32) > (b >> 32);
  }
};

namespace intro {

template <typename T, typename C>
__host__ __device__ inline void push_heap(T* first, long hole, long top, T value, C comp) {
  long parent = (hole - 1) / 2;
  while (hole > top && comp(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

template <typename T, typename C>
__host__ __device__ inline void adjust_heap(T* first, long hole, long len, T value, C comp) {
  const long top = hole;
  long second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (comp(first[second], first[second - 1])) second--;
    first[hole] = first[second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    first[hole] = first[second - 1];
    hole = second - 1;
  }
  push_heap(first, hole, top, value, comp);
}

// std::__partial_sort(first, last, last): make_heap + sort_heap
template <typename T, typename C>
__host__ __device__ inline void heap_sort(T* first, T* last, C comp) {
  const long len = last - first;
  if (len >= 2) {
    long parent = (len - 2) / 2;
    while (true) {
      T value = first[parent];
      adjust_heap(first, parent, len, value, comp);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {
    --last;
    T value = *last;
    *last = *first;
    adjust_heap(first, 0L, static_cast<long>(last - first), value, comp);
  }
}

template <typename T>
__host__ __device__ __forceinline__ void iter_swap(T* a, T* b) {
  T t = *a;
  *a = *b;
  *b = t;
}

template <typename T, typename C>
__host__ __device__ inline void move_median_to_first(T* result, T* a, T* b, T* c, C comp) {
  if (comp(*a, *b)) {
    if (comp(*b, *c)) iter_swap(result, b);
    else if (comp(*a, *c)) iter_swap(result, c);
    else iter_swap(result, a);
  } else if (comp(*a, *c)) iter_swap(result, a);
  else if (comp(*b, *c)) iter_swap(result, c);
  else iter_swap(result, b);
}

template <typename T, typename C>
__host__ __device__ inline T* unguarded_partition(T* first, T* last, T* pivot, C comp) {
  while (true) {
    while (comp(*first, *pivot)) ++first;
    --last;
    while (comp(*pivot, *last)) --last;
    if (!(first < last)) return first;
    iter_swap(first, last);
    ++first;
  }
}

template <typename T, typename C>
__host__ __device__ inline void unguarded_linear_insert(T* last, C comp) {
  T val = *last;
  T* next = last;
  --next;
  while (comp(val, *next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}

template <typename T, typename C>
__host__ __device__ inline void insertion_sort(T* first, T* last, C comp) {
  if (first == last) return;
  for (T* i = first + 1; i != last; ++i) {
    if (comp(*i, *first)) {
      T val = *i;
      for (T* p = i; p != first; --p) *p = *(p - 1);  // move_backward(first, i, i + 1)
      *first = val;
    } else {
      unguarded_linear_insert(i, comp);
    }
  }
}

}  // namespace intro

template <typename T, typename C>
__host__ __device__ inline void std_sort(T* first, T* last, C comp) {
  if (first == last) return;
  const long n = last - first;
  long lg = 0;
  for (long t = n; t > 1; t >>= 1) ++lg;  // std::__lg
  // __introsort_loop with the right-hand recursion on an explicit stack
  struct Frame {
    T* first;
    T* last;
    long depth;
  };
  Frame stack[96];
  int sp = 0;
  stack[sp++] = Frame{first, last, lg * 2};
  while (sp > 0) {
    Frame f = stack[--sp];
    T* lo = f.first;
    T* hi = f.last;
    long depth = f.depth;
    while (hi - lo > 16) {
      if (depth == 0) {
        intro::heap_sort(lo, hi, comp);
        break;
      }
      --depth;
      T* mid = lo + (hi - lo) / 2;
      intro::move_median_to_first(lo, lo + 1, mid, hi - 1, comp);
      T* cut = intro::unguarded_partition(lo + 1, hi, lo, comp);
      // reference recurses into [cut, hi) FIRST, then continues with [lo, cut): the two ranges are
      // disjoint, so processing order does not change the result; push the right part.
      stack[sp++] = Frame{cut, hi, depth};
      hi = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    intro::insertion_sort(first, first + 16, comp);
    for (T* i = first + 16; i != last; ++i) intro::unguarded_linear_insert(i, comp);
  } else {
    intro::insertion_sort(first, last, comp);
  }
}

}  // namespace rvn
