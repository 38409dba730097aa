// freelist.h — first-fit free list over the offsets [0, size) of one block of memory: the bookkeeping of the device arena
// (engine.hip: devpool).  Offsets and lengths only, no memory behind it, so the CPU suite can drive it
// (rvn_test_freelist, tests/test_freelist.py).  Blocks are multiples of `grain`; free neighbours are coalesced.
#ifndef RVN_FREELIST_H_
#define RVN_FREELIST_H_

#include <cstddef>
#include <iterator>
#include <map>
#include <unordered_map>

namespace rvn {

struct FreeList {
  size_t size = 0, grain = 64 << 10;
  std::map<size_t, size_t> holes;             // offset -> length, coalesced
  std::unordered_map<size_t, size_t> in_use;  // offset -> length

  void reset(size_t bytes, size_t grain_bytes) {
    grain = grain_bytes ? grain_bytes : 1;
    size = bytes / grain * grain;
    holes.clear();
    in_use.clear();
    if (size) holes[0] = size;
  }
  // lowest hole that holds `bytes` (rounded up to the grain); false: none
  bool alloc(size_t bytes, size_t* off) {
    bytes = (bytes + grain - 1) / grain * grain;
    if (bytes == 0) bytes = grain;
    for (auto it = holes.begin(); it != holes.end(); ++it) {
      if (it->second < bytes) continue;
      const size_t o = it->first, len = it->second;
      holes.erase(it);
      if (len > bytes) holes[o + bytes] = len - bytes;
      in_use[o] = bytes;
      *off = o;
      return true;
    }
    return false;
  }
  bool owns(size_t off) const { return in_use.count(off) != 0; }
  // false: `off` is not the start of a block in use
  bool release(size_t off) {
    auto it = in_use.find(off);
    if (it == in_use.end()) return false;
    size_t len = it->second;
    in_use.erase(it);
    auto next = holes.lower_bound(off);
    if (next != holes.end() && off + len == next->first) {
      len += next->second;
      next = holes.erase(next);
    }
    if (next != holes.begin()) {
      auto prev = std::prev(next);
      if (prev->first + prev->second == off) {
        off = prev->first;
        len += prev->second;
        holes.erase(prev);
      }
    }
    holes[off] = len;
    return true;
  }
  size_t free_total() const {
    size_t t = 0;
    for (const auto& h : holes) t += h.second;
    return t;
  }
  size_t free_largest() const {
    size_t t = 0;
    for (const auto& h : holes) t = t > h.second ? t : h.second;
    return t;
  }
};

}  // namespace rvn

#endif  // RVN_FREELIST_H_
