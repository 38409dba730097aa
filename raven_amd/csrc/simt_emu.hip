// simt_emu.hip — host-side wavefront emulator behind simt.h: 64 fibres (one per lane) on the calling thread, switched
// at every cross-lane call.  TEST INFRASTRUCTURE: it exists so that the CPU suite can run a kernel written against
// sv:: (poa3.hip) line by line against the oracle; nothing on the product path calls it.
//
// A cross-lane call (exchange / ballot / sync) is a rendezvous: each fibre deposits its operand and yields until the
// last of the 64 has arrived.  The operand buffers are double-buffered by rendezvous parity, so one barrier per call is
// enough (a fibre cannot be two rendezvous ahead of another).  Every call carries its source line; if the fibres of a
// wave meet at different lines the kernel has a cross-lane call under a lane-dependent branch — reported and aborted,
// because the GPU would execute that with a partial EXEC mask and the emulation would no longer mean anything.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "simt.h"

#if !defined(__HIP_DEVICE_COMPILE__)

extern "C" void rvn_fiber_switch(void** save_sp, void* load_sp);
// callee-saved registers of the System V x86-64 ABI on the old stack, stack pointers swapped, the same popped off the new
asm(R"(
.text
.globl rvn_fiber_switch
.type rvn_fiber_switch,@function
rvn_fiber_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size rvn_fiber_switch,.-rvn_fiber_switch
)");

namespace rvn {
namespace simt_emu {

namespace {

constexpr int kLanes = 64;
constexpr size_t kStackBytes = 512 * 1024;

struct WaveState {
  void* main_sp = nullptr;
  void* sp[kLanes] = {};
  bool finished[kLanes] = {};
  int cur = -1;
  void (*fn)(void*) = nullptr;
  void* arg = nullptr;
  // rendezvous
  unsigned gen = 0;
  int arrived = 0;
  int xbuf[2][kLanes] = {};
  unsigned long long bal[2] = {};
  int site_of[kLanes] = {};
  unsigned count[kLanes] = {};  // rendezvous passed by each lane
  bool progressed = false;
  std::vector<unsigned char> stacks;
};

thread_local WaveState* W = nullptr;

void yield_to_main() { rvn_fiber_switch(&W->sp[W->cur], W->main_sp); }

[[noreturn]] void die(const char* what, int a, int b) {
  std::fprintf(stderr, "[simt_emu] %s (source lines %d / %d)\n", what, a, b);
  std::abort();
}

void trampoline() {
  WaveState* w = W;
  w->fn(w->arg);
  w->finished[w->cur] = true;
  w->progressed = true;
  for (;;) yield_to_main();
}

// every fibre calls this with its operand already deposited; returns once all 64 have arrived
void rendezvous(int site) {
  WaveState* w = W;
  const int l = w->cur;
  w->site_of[l] = site;
  const unsigned my = w->gen;
  if (++w->arrived == kLanes) {
    for (int i = 1; i < kLanes; ++i)
      if (w->site_of[i] != w->site_of[0]) die("lanes met at different cross-lane calls", w->site_of[0], w->site_of[i]);
    w->arrived = 0;
    ++w->gen;
    w->progressed = true;
  } else {
    while (w->gen == my) yield_to_main();
  }
  ++w->count[l];
}

}  // namespace

int lane() { return W->cur; }

int exchange(int v, int src_lane, int site) {
  WaveState* w = W;
  const unsigned par = w->count[w->cur] & 1u;
  w->xbuf[par][w->cur] = v;
  rendezvous(site);
  return w->xbuf[par][src_lane & 63];
}

unsigned long long ballot(bool p, int site) {
  WaveState* w = W;
  const unsigned par = w->count[w->cur] & 1u;
  w->xbuf[par][w->cur] = p ? 1 : 0;
  rendezvous(site);
  unsigned long long m = 0;
  for (int i = 0; i < kLanes; ++i) m |= static_cast<unsigned long long>(w->xbuf[par][i] & 1) << i;
  return m;
}

void sync(int site) { rendezvous(site); }

void run_wave(void (*fn)(void*), void* arg) {
  WaveState ws;
  ws.fn = fn;
  ws.arg = arg;
  ws.stacks.resize(kStackBytes * kLanes + 64);
  WaveState* outer = W;
  W = &ws;
  for (int l = 0; l < kLanes; ++l) {
    uintptr_t top = reinterpret_cast<uintptr_t>(ws.stacks.data()) + kStackBytes * (l + 1);
    top &= ~static_cast<uintptr_t>(15);
    uint64_t* sp = reinterpret_cast<uint64_t*>(top);
    *--sp = 0;                                         // where trampoline's caller would have its return address
    *--sp = reinterpret_cast<uint64_t>(&trampoline);   // `ret` of the first switch jumps here
    for (int i = 0; i < 6; ++i) *--sp = 0;             // rbp rbx r12 r13 r14 r15
    ws.sp[l] = sp;
  }
  for (;;) {
    bool all_done = true;
    ws.progressed = false;
    for (int l = 0; l < kLanes; ++l) {
      if (ws.finished[l]) continue;
      all_done = false;
      ws.cur = l;
      rvn_fiber_switch(&ws.main_sp, ws.sp[l]);
    }
    if (all_done) break;
    if (!ws.progressed) {
      int a = -1, b = -1;
      for (int l = 0; l < kLanes; ++l) {
        if (ws.finished[l]) b = l;
        else a = l;
      }
      die("deadlock: some lanes returned while others wait at a cross-lane call", a >= 0 ? ws.site_of[a] : -1, b);
    }
  }
  W = outer;
}

}  // namespace simt_emu
}  // namespace rvn

#endif  // host pass
