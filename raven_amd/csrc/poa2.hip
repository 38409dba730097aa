// poa2.hip — banded, LDS-staged POA window kernel: the fast path behind rvn_poa_consensus_batch (racon
// Window::GenerateConsensus over spoa, as driven by raven::Polish, RavenLib/src/polish.cc:43-51).
//
// Same algorithm and graph layout as poa.hip (one wave per window, SoA graph in a per-slot global scratch,
// incremental topological order), re-cut so that no serial step waits on HBM:
//   * NW is BANDED: every node carries its backbone coordinate (`bpos`), the band of its row is the 128 columns
//     around the layer position that coordinate maps to ((bpos - begin) * len / span).  4x fewer cells than the
//     full matrix for racon's 500-bp windows; a layer whose traceback comes within 2 cells of a band edge marks the
//     window kPoaBandHit and the host re-runs it through the full-matrix kernel, so banding never silently
//     changes a result.
//   * predecessor rows come from a 32-row score ring in LDS (hit rate ~all: an incremental order keeps a
//     node's in-edges within a few ranks); only ring misses read the int16 copy in HBM.
//   * the DP writes one BACKPOINTER byte per cell (which in-edge, diagonal/vertical/horizontal, chosen with
//     spoa's traceback priority), so the traceback is a walk over bytes: blocks of 64 rows x band B are staged
//     into LDS with one coalesced load, together with a per-row table (node, band start, first 4 predecessor
//     ranks), and the walk itself touches only LDS.
//   * spoa's AddAlignment runs lane-parallel, one sequence position per lane: the nodes on an alignment path are
//     distinct and belong to distinct aligned groups, so target lookup, node creation, group updates and the
//     edge (p-1 -> p) are conflict-free; new ids and order slots come from ballots/prefix counts in path order,
//     identical to the serial order of creation.
// Integer VALU + LDS bound; no MFMA.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "poa.h"
#include "poa2_window.h"

namespace rvn {

namespace {

using namespace p2;

template <int NCH, int WPB, int OCC>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(OCC))) void poa2_kernel(const PoaWindow* __restrict__ windows, u32 n_windows,
                                                  const PoaLayer* __restrict__ layers, const PoaSrc src,
                                                  unsigned char* __restrict__ scratch,
                                                  size_t slot_bytes, u32 n_slots, u32 nmax, u32 lmax, int m, int n_,
                                                  int gp, int trim, u8* __restrict__ out, u32* __restrict__ out_len,
                                                  u32* __restrict__ status,
                                                  unsigned long long* __restrict__ phase_cycles,
                                                  const u32* __restrict__ sched, u32* __restrict__ next, u32 probe) {
  __shared__ Poa2Lds<NCH> lds[WPB];
  const u32 wv = static_cast<u32>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
  const u32 slot = blockIdx.x * WPB + wv;
  if (slot >= n_slots) return;
  Poa2Slot g = poa2_carve(scratch + static_cast<size_t>(slot) * slot_bytes, nmax, lmax, 64 * NCH);
  for (;;) {
    const u32 wi = poa_next_window(next, sched, n_windows);
    if (wi == 0xFFFFFFFFu) break;
    const PoaWindow win = windows[wi];
    const u32 st = poa2_window<NCH>(win, layers, src, g, nmax, lmax, m, n_, gp, trim, lds[wv],
                               out + win.out_off, out_len + wi, phase_cycles, probe);
    if (lane_id() == 0) status[wi] = st;
    wsync();
  }
}

}  // namespace

void poa_v2_launch(Engine& e, const PoaBatchDev& b, int nch) {
  if (b.n_windows == 0) return;
  const size_t slot_bytes = poa2_slot_bytes(b.nmax, b.lmax, 64u * nch);
  size_t free_b = 0, total_b = 0;
  RVN_HIP(hipMemGetInfo(&free_b, &total_b));
  u32 occ = 6;  // waves per SIMD the 64-column kernel is built for (80 VGPRs; LDS allows 7 workgroups of 4 waves per CU): measured 5 / 6 / 7 -> 158.7k / 170.9k / 170.1k windows/s
  if (const char* ev = knob("RVN_POA_OCC")) occ = static_cast<u32>(std::atoi(ev));
  occ = occ < 5 ? 5 : (occ > 7 ? 7 : occ);
  u32 per_cu = nch == 1 ? 4 * occ : 16;
  if (const char* ev = knob("RVN_POA_WAVES_PER_CU")) per_cu = static_cast<u32>(std::atoi(ev));  // occupancy experiments
  u32 n_slots = std::min<u32>(b.n_windows, 256 * per_cu);
  const size_t budget = e.poa2_scratch.cap + (free_b + devpool::free_total()) / 2;
  if (static_cast<size_t>(n_slots) * slot_bytes > budget) n_slots = static_cast<u32>(std::max<size_t>(1, budget / slot_bytes));
  n_slots = ((n_slots + 3) / 4) * 4;
  unsigned char* d_scratch = e.poa2_scratch.get<unsigned char>(static_cast<size_t>(n_slots) * slot_bytes + 256);
  RVN_HIP(hipMemsetAsync(b.next, 0, 4, e.stream));
  if (nch == 1 && occ == 5) {
    RVN_KLAUNCH(kKPoaBanded, (poa2_kernel<1, 4, 5><<<n_slots / 4, 256, 0, e.stream>>>(
                                 b.wins, b.n_windows, b.layers, b.src, d_scratch, slot_bytes, n_slots, b.nmax,
                                 b.lmax, b.m, b.n, b.g, b.trim, b.out, b.out_len, b.status, b.phase_cycles, b.sched, b.next, b.probe)));
  } else if (nch == 1 && occ == 6) {
    RVN_KLAUNCH(kKPoaBanded, (poa2_kernel<1, 4, 6><<<n_slots / 4, 256, 0, e.stream>>>(
                                 b.wins, b.n_windows, b.layers, b.src, d_scratch, slot_bytes, n_slots, b.nmax,
                                 b.lmax, b.m, b.n, b.g, b.trim, b.out, b.out_len, b.status, b.phase_cycles, b.sched, b.next, b.probe)));
  } else if (nch == 1) {
    RVN_KLAUNCH(kKPoaBanded, (poa2_kernel<1, 4, 7><<<n_slots / 4, 256, 0, e.stream>>>(
                                 b.wins, b.n_windows, b.layers, b.src, d_scratch, slot_bytes, n_slots, b.nmax,
                                 b.lmax, b.m, b.n, b.g, b.trim, b.out, b.out_len, b.status, b.phase_cycles, b.sched, b.next, b.probe)));
  } else if (nch == 2) {
    RVN_KLAUNCH(kKPoaBanded, (poa2_kernel<2, 4, 1><<<n_slots / 4, 256, 0, e.stream>>>(
                                 b.wins, b.n_windows, b.layers, b.src, d_scratch, slot_bytes, n_slots, b.nmax,
                                 b.lmax, b.m, b.n, b.g, b.trim, b.out, b.out_len, b.status, b.phase_cycles, b.sched, b.next, b.probe)));
  } else {  // 256 columns: 21.6 KB of LDS per wave -> 2 waves per workgroup
    RVN_KLAUNCH(kKPoaBanded, (poa2_kernel<4, 2, 1><<<n_slots / 2, 128, 0, e.stream>>>(
                                 b.wins, b.n_windows, b.layers, b.src, d_scratch, slot_bytes, n_slots, b.nmax,
                                 b.lmax, b.m, b.n, b.g, b.trim, b.out, b.out_len, b.status, b.phase_cycles, b.sched, b.next, b.probe)));
  }
}

}  // namespace rvn
