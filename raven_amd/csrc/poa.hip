// poa.hip — batched partial-order-alignment window consensus on the device (the kernel behind the racon
// polishing rounds that raven::Polish drives, RavenLib/src/polish.cc:43-51; racon Window::GenerateConsensus
// over spoa: graph + linear-gap NW (m,n,g of polish.hpp:13-17) + heaviest bundle + TGS trim).
//
// One WAVE per window, persistent over windows (slot = wave): the graph lives in a per-slot global scratch
// (SoA: code, in-edge lists with weights, aligned groups, visit counts, topological order), the current layer
// (bases + weights) in LDS.  Per layer:
//   1. (partial layers only) spoa's Subgraph = ancestors of backbone node `end` with id >= begin (lane-0 DFS);
//   2. NW over the nodes in topological order: one DP row per node, 64 columns per step; diagonal/vertical
//      terms from every predecessor row, the horizontal gap chain as a wave prefix-max of (H - j*g);
//      int16 scores in an (N+1) x (L+1) matrix in HBM;
//   3. traceback with spoa's move priority (diagonal over in-edges in insertion order, vertical, horizontal);
//   4. spoa Graph::AddAlignment (lane 0), new nodes created in path order;
//   5. the topological order is maintained incrementally instead of re-sorted: a new node goes right after the
//      last old node of its path (new rank of old r = r + #new nodes anchored before it; of the t-th new node =
//      anchor slot + t) — any valid order gives the same DP matrix; only equal-score ties can resolve differently
//      from spoa's DFS order, which is why consensus parity is tolerance-based.
// Integer VALU + L2 bound; no MFMA.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "poa.h"

namespace rvn {

namespace {

struct PoaSlot {
  i16* H;
  u8* code;
  u8* in_cnt;
  u16* in_tail;
  i32* in_w;
  u16* out_cnt;
  u8* al_cnt;
  u16* al;
  u16* visits;
  u16* rank_of;
  u16* order;
  u16* order2;
  u8* mark;
  u16* sub_out;
  i32* aln_node;
  i32* aln_pos;
  u16* new_slot;
  i32* scores;
  i32* preds;
  u16* stack;
};

__host__ __device__ inline size_t poa_slot_bytes(u32 nmax, u32 lmax) {
  size_t b = 0;
  auto add = [&](size_t x) { b += (x + 255) & ~size_t(255); };
  add(static_cast<size_t>(nmax + 1) * (lmax + 1) * 2);  // H
  add(nmax);                                            // code
  add(nmax);                                            // in_cnt
  add(static_cast<size_t>(nmax) * kPoaMaxIn * 2);       // in_tail
  add(static_cast<size_t>(nmax) * kPoaMaxIn * 4);       // in_w
  add(nmax * 2);                                        // out_cnt
  add(nmax);                                            // al_cnt
  add(nmax * 4 * 2);                                    // al
  add(nmax * 2);                                        // visits
  add(nmax * 2);                                        // rank_of
  add(nmax * 2);                                        // order
  add(nmax * 2);                                        // order2
  add(nmax);                                            // mark
  add(nmax * 2);                                        // sub_out
  add(static_cast<size_t>(nmax + lmax + 2) * 4);        // aln_node
  add(static_cast<size_t>(nmax + lmax + 2) * 4);        // aln_pos
  add(static_cast<size_t>(lmax + 2) * 2);               // new_slot
  add(nmax * 4);                                        // scores
  add(nmax * 4);                                        // preds
  add(nmax * 2);                                        // stack
  return b;
}

__device__ inline PoaSlot poa_carve(unsigned char* base, u32 nmax, u32 lmax) {
  PoaSlot s;
  size_t o = 0;
  auto take = [&](size_t x) {
    unsigned char* p = base + o;
    o += (x + 255) & ~size_t(255);
    return p;
  };
  s.H = reinterpret_cast<i16*>(take(static_cast<size_t>(nmax + 1) * (lmax + 1) * 2));
  s.code = take(nmax);
  s.in_cnt = take(nmax);
  s.in_tail = reinterpret_cast<u16*>(take(static_cast<size_t>(nmax) * kPoaMaxIn * 2));
  s.in_w = reinterpret_cast<i32*>(take(static_cast<size_t>(nmax) * kPoaMaxIn * 4));
  s.out_cnt = reinterpret_cast<u16*>(take(nmax * 2));
  s.al_cnt = take(nmax);
  s.al = reinterpret_cast<u16*>(take(nmax * 4 * 2));
  s.visits = reinterpret_cast<u16*>(take(nmax * 2));
  s.rank_of = reinterpret_cast<u16*>(take(nmax * 2));
  s.order = reinterpret_cast<u16*>(take(nmax * 2));
  s.order2 = reinterpret_cast<u16*>(take(nmax * 2));
  s.mark = take(nmax);
  s.sub_out = reinterpret_cast<u16*>(take(nmax * 2));
  s.aln_node = reinterpret_cast<i32*>(take(static_cast<size_t>(nmax + lmax + 2) * 4));
  s.aln_pos = reinterpret_cast<i32*>(take(static_cast<size_t>(nmax + lmax + 2) * 4));
  s.new_slot = reinterpret_cast<u16*>(take(static_cast<size_t>(lmax + 2) * 2));
  s.scores = reinterpret_cast<i32*>(take(nmax * 4));
  s.preds = reinterpret_cast<i32*>(take(nmax * 4));
  s.stack = reinterpret_cast<u16*>(take(nmax * 2));
  return s;
}

__device__ inline u32 poa_add_node(PoaSlot& g, u32& n_nodes, u32 code) {
  const u32 id = n_nodes++;
  g.code[id] = static_cast<u8>(code);
  g.in_cnt[id] = 0;
  g.out_cnt[id] = 0;
  g.al_cnt[id] = 0;
  g.visits[id] = 0;
  return id;
}

// One window. Returns status: 0 = backbone returned (< 3 sequences), 1 = polished, 2 = limits exceeded.
__device__ u32 poa_window(const PoaWindow& win, const PoaLayer* __restrict__ layers, const PoaSrc& src,
                          PoaSlot& g, u32 nmax, u32 lmax, int m, int n_, int gp, int trim,
                          u8* s_seq, u8* s_w, u8* __restrict__ out, u32* out_len,
                          unsigned long long* __restrict__ phase_cycles) {
  const int lane = lane_id();
  unsigned long long t_sub = 0, t_dp = 0, t_tb = 0, t_add = 0, t_ord = 0, t_cons = 0, t0 = 0;
  auto tick = [&]() { t0 = __builtin_readcyclecounter(); };
  auto tock = [&](unsigned long long& acc) { acc += __builtin_readcyclecounter() - t0; };
  const PoaLayer bb = layers[win.layer_first];
  const u32 blen = bb.len;
  auto copy_backbone = [&]() {
    const u32 n = blen < win.out_cap ? blen : win.out_cap;
    for (u32 i = lane; i < n; i += 64) out[i] = static_cast<u8>(poa_layer_code(src, bb, i));
    if (lane == 0) *out_len = n;
  };
  // layers dropped by racon's mean-quality filter do not count as sequences of the window
  u32 n_eff = win.n_layers;
  if (src.layer_ok) {
    u32 cnt = 0;
    for (u32 i = 1 + lane; i < win.n_layers; i += 64) cnt += src.layer_ok[win.layer_first + i] ? 1u : 0u;
    n_eff = 1 + wave_sum(cnt);
  }
  if (n_eff < 3) {
    copy_backbone();
    return 0;
  }
  if (blen == 0 || blen > nmax || blen > lmax) {
    copy_backbone();
    return 4;
  }
  // ---- backbone graph (spoa AddAlignment with an empty alignment) ----
  u32 n_nodes = blen;
  for (u32 i = lane; i < blen; i += 64) {
    g.code[i] = static_cast<u8>(poa_layer_code(src, bb, i));
    g.al_cnt[i] = 0;
    g.visits[i] = blen >= 2 ? 1 : 0;
    g.rank_of[i] = static_cast<u16>(i);
    g.order[i] = static_cast<u16>(i);
    const i32 wi = poa_layer_weight(src, bb, i);
    if (i > 0) {
      const i32 wp = poa_layer_weight(src, bb, i - 1);
      g.in_cnt[i] = 1;
      g.in_tail[i * kPoaMaxIn] = static_cast<u16>(i - 1);
      g.in_w[i * kPoaMaxIn] = wp + wi;
    } else {
      g.in_cnt[i] = 0;
    }
    g.out_cnt[i] = i + 1 < blen ? 1 : 0;
  }
  wsync();
  const u32 offset = static_cast<u32>(0.01 * blen);
  u32 failed = 0;  // 2 nodes, 3 in-degree, 4 layer length, 5 internal

  for (u32 li = 1; li < win.n_layers && !failed; ++li) {
    const PoaLayer L = layers[win.layer_first + li];
    const u32 len = L.len;
    if (len == 0 || (src.layer_ok && !src.layer_ok[win.layer_first + li])) continue;
    if (len > lmax || len > kPoaMaxSeq) {
      failed = 4;
      break;
    }
    for (u32 i = lane; i < len; i += 64) {
      s_seq[i] = static_cast<u8>(poa_layer_code(src, L, i));
      s_w[i] = static_cast<u8>(poa_layer_weight(src, L, i));
    }
    const bool full = L.begin < offset && L.end > blen - offset;
    tick();
    // ---- 1. subgraph marks ----
    if (!full) {
      poa_subgraph_marks(g, n_nodes, nmax, L.begin, L.end);
    }
    tock(t_sub);
    tick();
    // ---- 2. NW matrix ----
    const u32 w = len + 1;
    for (u32 j = lane; j < w; j += 64) g.H[j] = static_cast<i16>(static_cast<i32>(j) * gp);
    wsync();
    i32 best_score = -0x7FFFFFFF;
    u32 best_row = 0, best_node = 0;  // end node: equal scores -> smallest node id (as poa2.hip / poa4.hip)
    // the row computed last stays in registers (lane l holds columns c*64+l): it is the predecessor of most
    // rows, so the common case needs no global load and no store->load fence
    i32 lastrow[kPoaMaxSeq / 64 + 1];
#pragma unroll
    for (int c = 0; c < kPoaMaxSeq / 64 + 1; ++c) {
      const u32 j = c * 64 + lane;
      lastrow[c] = j < w ? static_cast<i32>(j) * gp : kNegInf16;
    }
    u32 last_row_idx = 0;
    bool dirty = false;  // rows stored since the last fence
    for (u32 r0 = 0; r0 < n_nodes; r0 += 64) {
     // Row metadata for 64 rows at once, one row per lane: the order -> in-edges -> rank chain of dependent
     // loads is paid once per 64 rows instead of once per row (it was ~80 % of the row time).
     int m_v = 0, m_np = 0, m_p0 = 0, m_p1 = 0, m_code = 0, m_outc = 1, m_marked = 0;
     if (r0 + lane < n_nodes) {
       m_v = g.order[r0 + lane];
       m_marked = (full || g.mark[m_v]) ? 1 : 0;
       if (m_marked) {
         m_code = g.code[m_v];
         m_outc = full ? g.out_cnt[m_v] : g.sub_out[m_v];
         const u32 c = g.in_cnt[m_v];
         for (u32 k = 0; k < c; ++k) {
           const u32 t = g.in_tail[m_v * kPoaMaxIn + k];
           if (full || g.mark[t]) {
             const int pr = static_cast<int>(g.rank_of[t]) + 1;
             if (m_np == 0) m_p0 = pr;
             else if (m_np == 1) m_p1 = pr;
             ++m_np;
           }
         }
       }
     }
     const u32 rows_here = n_nodes - r0 < 64 ? n_nodes - r0 : 64;
     for (u32 ri = 0; ri < rows_here; ++ri) {
      const u32 r = r0 + ri;
      if (!__shfl(m_marked, static_cast<int>(ri), 64)) continue;
      const u32 v = static_cast<u32>(__shfl(m_v, static_cast<int>(ri), 64));
      const u32 row = r + 1;
      // predecessor rows (in-edges whose tail is inside the subgraph), in insertion order
      u32 prow[kPoaMaxIn];
      u32 np = static_cast<u32>(__shfl(m_np, static_cast<int>(ri), 64));
      if (np <= 2) {
        prow[0] = static_cast<u32>(__shfl(m_p0, static_cast<int>(ri), 64));
        prow[1] = static_cast<u32>(__shfl(m_p1, static_cast<int>(ri), 64));
      } else {
        np = 0;
        const u32 c = g.in_cnt[v];
        for (u32 k = 0; k < c; ++k) {
          const u32 t = g.in_tail[v * kPoaMaxIn + k];
          if (full || g.mark[t]) prow[np++] = static_cast<u32>(g.rank_of[t]) + 1;
        }
      }
      const bool no_pred = np == 0;
      if (no_pred) {
        prow[0] = 0;
        np = 1;
      }
      const u32 vc = static_cast<u32>(__shfl(m_code, static_cast<int>(ri), 64));
      i16* Hr = g.H + static_cast<size_t>(row) * w;
      i32 carry_h = 0;  // H[row][c0 - 1] of the previous chunk
      bool far = false;
      for (u32 k = 0; k < np; ++k) far |= prow[k] != last_row_idx;
      if (far && dirty) {  // a predecessor row comes from memory: earlier stores must have landed
        wsync();
        dirty = false;
      }
      i32 prev_chunk_last = kNegInf16;  // lastrow value of column c0-1 (diag for lane 0)
      i32 end_score = kNegInf16;
#pragma unroll
      for (int ci = 0; ci < kPoaMaxSeq / 64 + 1; ++ci) {
        const u32 c0 = ci * 64;
        if (c0 >= w) continue;
        const u32 j = c0 + lane;
        const bool valid = j < w;
        i32 best = -0x3FFFFFFF;
        i32 col0 = -0x3FFFFFFF;
        for (u32 k = 0; k < np; ++k) {
          i32 up, diag;
          if (prow[k] == last_row_idx) {
            up = lastrow[ci];
            diag = dpp_wave_shr1(up, prev_chunk_last);
          } else {
            const i16* Hp = g.H + static_cast<size_t>(prow[k]) * w;
            up = valid ? static_cast<i32>(Hp[j]) : kNegInf16;
            const i32 edge = c0 ? static_cast<i32>(Hp[c0 - 1]) : kNegInf16;
            diag = dpp_wave_shr1(up, edge);
          }
          if (j >= 1 && valid) {
            const i32 s = (vc == s_seq[j - 1]) ? m : n_;
            const i32 a = diag + s, b = up + gp;
            const i32 x = a > b ? a : b;
            best = x > best ? x : best;
          }
          col0 = up > col0 ? up : col0;
        }
        if (j == 0) best = (no_pred ? 0 : col0) + gp;
        // horizontal chain: H[j] = max_k<=j (best[k] + (j-k) g)  ->  prefix max of (best - j g)
        i32 x = valid ? best - static_cast<i32>(j) * gp : -0x3FFFFFFF;
        x = wave_inclusive_max_dpp(x, -0x3FFFFFFF);
        i32 h = x + static_cast<i32>(j) * gp;
        if (c0) {
          const i32 viac = carry_h + static_cast<i32>(lane + 1) * gp;
          h = viac > h ? viac : h;
        }
        h = h < kNegInf16 ? kNegInf16 : h;
        if (valid) Hr[j] = static_cast<i16>(h);
        carry_h = __builtin_amdgcn_readlane(h, 63);
        prev_chunk_last = __builtin_amdgcn_readlane(lastrow[ci], 63);
        lastrow[ci] = valid ? h : kNegInf16;  // in place: this chunk's old values are no longer needed
        if (c0 + 64 >= w) end_score = __shfl(h, static_cast<int>((w - 1) & 63), 64);
      }
      last_row_idx = row;
      dirty = true;
      const u32 outc = static_cast<u32>(__shfl(m_outc, static_cast<int>(ri), 64));
      if (outc == 0) {
        const i32 sc = end_score;
        if (sc > best_score || (sc == best_score && best_row != 0 && v < best_node)) {
          best_score = sc;
          best_row = row;
          best_node = v;
        }
      }
     }
    }
    wsync();  // the whole matrix must be visible to the traceback
    tock(t_dp);
    tick();
    if (best_row == 0) {  // no end node inside the subgraph (cannot happen for a valid layer)
      failed = 5 | (li << 8);
      break;
    }
    // ---- 3. traceback (lane 0), pairs stored end -> start ----
    u32 n_aln = 0;
    if (lane == 0) {
      u32 i = best_row, j = w - 1;
      while (!(i == 0 && j == 0) && n_aln < nmax + lmax) {
        const i32 Hij = g.H[static_cast<size_t>(i) * w + j];
        bool found = false;
        u32 pi = 0, pj = 0;
        u32 v = 0, c = 0;
        if (i != 0) {
          v = g.order[i - 1];
          c = g.in_cnt[v];
        }
        if (i != 0 && j != 0) {
          const i32 mc = (g.code[v] == s_seq[j - 1]) ? m : n_;
          bool any = false;
          for (u32 k = 0; k < c && !found; ++k) {
            const u32 t = g.in_tail[v * kPoaMaxIn + k];
            if (!(full || g.mark[t])) continue;
            any = true;
            const u32 pr = static_cast<u32>(g.rank_of[t]) + 1;
            if (Hij == g.H[static_cast<size_t>(pr) * w + j - 1] + mc) {
              pi = pr;
              pj = j - 1;
              found = true;
            }
          }
          if (!any && Hij == g.H[j - 1] + mc) {
            pi = 0;
            pj = j - 1;
            found = true;
          }
        }
        if (!found && i != 0) {
          bool any = false;
          for (u32 k = 0; k < c && !found; ++k) {
            const u32 t = g.in_tail[v * kPoaMaxIn + k];
            if (!(full || g.mark[t])) continue;
            any = true;
            const u32 pr = static_cast<u32>(g.rank_of[t]) + 1;
            if (Hij == g.H[static_cast<size_t>(pr) * w + j] + gp) {
              pi = pr;
              pj = j;
              found = true;
            }
          }
          if (!any && Hij == g.H[j] + gp) {
            pi = 0;
            pj = j;
            found = true;
          }
        }
        if (!found && j != 0 && Hij == g.H[static_cast<size_t>(i) * w + j - 1] + gp) {
          pi = i;
          pj = j - 1;
          found = true;
        }
        if (!found) {
          n_aln = 0xFFFFFFFFu;
          break;
        }
        g.aln_node[n_aln] = (i == pi) ? -1 : static_cast<i32>(g.order[i - 1]);
        g.aln_pos[n_aln] = (j == pj) ? -1 : static_cast<i32>(j - 1);
        ++n_aln;
        i = pi;
        j = pj;
      }
    }
    n_aln = __shfl(n_aln, 0, 64);
    tock(t_tb);
    tick();
    if (n_aln == 0xFFFFFFFFu || n_aln == 0) {
      failed = (n_aln == 0 ? 7u : 6u) | (li << 8);
      break;
    }
    // ---- 4. spoa AddAlignment (lane 0); new nodes in path order, each with its order slot ----
    const u32 n_old = n_nodes;
    u32 n_new = 0;
    u32 ok = 1;
    u32 why = 3;  // failure reason if !ok: in-degree overflow unless a node limit was hit
    if (lane == 0) {
      // first / last aligned sequence positions
      i32 first_pos = -1, last_pos = -1;
      for (u32 a = n_aln; a-- > 0;) {
        if (g.aln_pos[a] != -1) {
          if (first_pos < 0) first_pos = g.aln_pos[a];
          last_pos = g.aln_pos[a];
        }
      }
      // A column = an aligned group.  New nodes anchored after a column go after ALL its members (a later read
      // may leave the column through any alternative); the unaligned prefix goes before all members of the
      // first column.
      auto group_max = [&](u32 v) -> u32 {
        u32 r = g.rank_of[v];
        for (u32 k = 0; k < g.al_cnt[v]; ++k) {
          const u32 a = g.al[v * 4 + k];
          if (a < n_old && g.rank_of[a] > r) r = g.rank_of[a];
        }
        return r;
      };
      u32 first_old_rank = n_old;
      for (u32 a = n_aln; a-- > 0;) {
        if (g.aln_node[a] != -1 && g.aln_pos[a] != -1) {
          const u32 v = g.aln_node[a];
          u32 r = g.rank_of[v];
          for (u32 k = 0; k < g.al_cnt[v]; ++k)
            if (g.rank_of[g.al[v * 4 + k]] < r) r = g.rank_of[g.al[v * 4 + k]];
          first_old_rank = r;
          break;
        }
      }
      i32 prev = -1;
      u32 cur_slot = first_old_rank;  // new nodes go right before old rank `cur_slot`
      auto new_node = [&](u32 code) -> i32 {
        if (n_nodes >= nmax || n_new >= lmax) {
          ok = 0;
          why = 2;
          return -1;
        }
        const u32 id = poa_add_node(g, n_nodes, code);
        g.new_slot[n_new++] = static_cast<u16>(cur_slot);
        return static_cast<i32>(id);
      };
      auto visit = [&](i32 node) { g.visits[node] += 1; };
      // unaligned prefix [0, first_pos)
      for (i32 p = 0; p < first_pos && ok; ++p) {
        const i32 curr = new_node(s_seq[p]);
        if (curr < 0) break;
        if (prev >= 0) ok &= poa_add_edge(g, prev, curr, static_cast<i32>(s_w[p - 1]) + s_w[p]);
        if (len >= 2) visit(curr);
        prev = curr;
      }
      // aligned part
      for (u32 a = n_aln; a-- > 0 && ok;) {
        const i32 sp = g.aln_pos[a];
        if (sp == -1) continue;
        const u32 code = s_seq[sp];
        const i32 an = g.aln_node[a];
        i32 curr = -1;
        if (an == -1) {
          curr = new_node(code);
        } else {
          cur_slot = group_max(static_cast<u32>(an)) + 1;  // later new nodes follow this column
          if (g.code[an] == code) {
            curr = an;
          } else {
            const u32 ac = g.al_cnt[an];
            for (u32 k = 0; k < ac; ++k) {
              const u32 kt = g.al[an * 4 + k];
              if (g.code[kt] == code) {
                curr = static_cast<i32>(kt);
                break;
              }
            }
            if (curr < 0) {
              curr = new_node(code);
              if (curr >= 0) {
                for (u32 k = 0; k < ac; ++k) {
                  const u32 kt = g.al[an * 4 + k];
                  if (g.al_cnt[kt] < 4) g.al[kt * 4 + g.al_cnt[kt]++] = static_cast<u16>(curr);
                  if (g.al_cnt[curr] < 4) g.al[curr * 4 + g.al_cnt[curr]++] = static_cast<u16>(kt);
                }
                if (g.al_cnt[an] < 4) g.al[an * 4 + g.al_cnt[an]++] = static_cast<u16>(curr);
                if (g.al_cnt[curr] < 4) g.al[curr * 4 + g.al_cnt[curr]++] = static_cast<u16>(an);
              }
            }
          }
        }
        if (curr < 0) {
          ok = 0;
          break;
        }
        if (prev >= 0) ok &= poa_add_edge(g, prev, curr, static_cast<i32>(s_w[sp - 1]) + s_w[sp]);
        if (len >= 2) visit(curr);
        prev = curr;
      }
      // unaligned suffix (last_pos, len)
      for (i32 p = last_pos + 1; p < static_cast<i32>(len) && ok; ++p) {
        const i32 curr = new_node(s_seq[p]);
        if (curr < 0) break;
        if (prev >= 0) ok &= poa_add_edge(g, prev, curr, static_cast<i32>(s_w[p - 1]) + s_w[p]);
        if (len >= 2) visit(curr);
        prev = curr;
      }
    }
    wsync();
    ok = __shfl(ok, 0, 64);
    n_nodes = __shfl(n_nodes, 0, 64);
    n_new = __shfl(n_new, 0, 64);
    why = __shfl(why, 0, 64);
    if (!ok) {
      failed = why;
      break;
    }
    tock(t_add);
    tick();
    // ---- 5. order rebuild: old rank r -> r + #(new slots <= r); t-th new node -> slot_t + t ----
    if (n_new) {
      for (u32 r = lane; r < n_old; r += 64) {
        u32 lo = 0, hi = n_new;  // upper_bound(new_slot, r)
        while (lo < hi) {
          const u32 mid = (lo + hi) >> 1;
          if (g.new_slot[mid] <= r) lo = mid + 1;
          else hi = mid;
        }
        g.order2[r + lo] = g.order[r];
      }
      for (u32 t = lane; t < n_new; t += 64) g.order2[static_cast<u32>(g.new_slot[t]) + t] = static_cast<u16>(n_old + t);
      wsync();
      for (u32 r = lane; r < n_nodes; r += 64) {
        const u32 v = g.order2[r];
        g.order[r] = static_cast<u16>(v);
        g.rank_of[v] = static_cast<u16>(r);
      }
      wsync();
    }
    tock(t_ord);
  }
  if (failed) {
    copy_backbone();
    return failed;
  }
  // ---- consensus: spoa TraverseHeaviestBundle + BranchCompletion (lane 0) ----
  tick();
  PoaWindow weff = win;
  weff.n_layers = n_eff;
  if (lane == 0) poa_consensus_lane0(g, n_nodes, nmax, weff, trim, out, out_len);
  wsync();
  tock(t_cons);
  if (phase_cycles && lane == 0) {
    atomicAdd(&phase_cycles[0], t_sub);
    atomicAdd(&phase_cycles[1], t_dp);
    atomicAdd(&phase_cycles[2], t_tb);
    atomicAdd(&phase_cycles[3], t_add);
    atomicAdd(&phase_cycles[4], t_ord);
    atomicAdd(&phase_cycles[5], t_cons);
  }
  return 1;
}

__global__ __launch_bounds__(256) void poa_kernel(const PoaWindow* __restrict__ windows, u32 n_windows,
                                                 const PoaLayer* __restrict__ layers, const PoaSrc src,
                                                 unsigned char* __restrict__ scratch,
                                                 size_t slot_bytes, u32 n_slots, u32 nmax, u32 lmax, int m, int n_,
                                                 int gp, int trim, u8* __restrict__ out, u32* __restrict__ out_len,
                                                 u32* __restrict__ status,
                                                 unsigned long long* __restrict__ phase_cycles,
                                                 const u32* __restrict__ sched, u32* __restrict__ next) {
  __shared__ u8 s_seq[4][kPoaMaxSeq];
  __shared__ u8 s_w[4][kPoaMaxSeq];
  const u32 wv = static_cast<u32>(__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
  const u32 slot = blockIdx.x * 4 + wv;
  if (slot >= n_slots) return;
  PoaSlot g = poa_carve(scratch + static_cast<size_t>(slot) * slot_bytes, nmax, lmax);
  for (;;) {
    const u32 wi = poa_next_window(next, sched, n_windows);
    if (wi == 0xFFFFFFFFu) break;
    const PoaWindow win = windows[wi];
    const u32 st = poa_window(win, layers, src, g, nmax, lmax, m, n_, gp, trim, s_seq[wv], s_w[wv],
                              out + win.out_off, out_len + wi, phase_cycles);
    if (lane_id() == 0) status[wi] = st;
    wsync();
  }
}

}  // namespace

void poa_v1_launch(Engine& e, const PoaBatchDev& b) {
  if (b.n_windows == 0) return;
  const size_t slot_bytes = poa_slot_bytes(b.nmax, b.lmax);
  size_t free_b = 0, total_b = 0;
  RVN_HIP(hipMemGetInfo(&free_b, &total_b));
  u32 n_slots = std::min<u32>(b.n_windows, 256 * 8);
  const size_t budget = e.poa_scratch.cap + (free_b + devpool::free_total()) / 2;
  if (static_cast<size_t>(n_slots) * slot_bytes > budget) n_slots = static_cast<u32>(std::max<size_t>(1, budget / slot_bytes));
  n_slots = ((n_slots + 3) / 4) * 4;
  unsigned char* d_scratch = e.poa_scratch.get<unsigned char>(static_cast<size_t>(n_slots) * slot_bytes + 256);
  RVN_HIP(hipMemsetAsync(b.next, 0, 4, e.stream));
  RVN_KLAUNCH(kKPoa, poa_kernel<<<n_slots / 4, 256, 0, e.stream>>>(b.wins, b.n_windows, b.layers, b.src,
                                                                    d_scratch, slot_bytes, n_slots, b.nmax, b.lmax, b.m,
                                                                    b.n, b.g, b.trim, b.out, b.out_len, b.status,
                                                                    b.phase_cycles, b.sched, b.next));
}

namespace {

// LPT order of the persistent waves: windows by decreasing number of layers (stable), built on the device
__global__ void poa_sched_keys_kernel(const PoaWindow* __restrict__ wins, u32 n, u32* __restrict__ keys, u32* __restrict__ vals) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 nl = wins[i].n_layers;
  keys[i] = 0xFFFFu - (nl < 0xFFFFu ? nl : 0xFFFFu);
  vals[i] = i;
}
__global__ void poa_gather_windows_kernel(const PoaWindow* __restrict__ wins, const u32* __restrict__ idx, u32 n,
                                          PoaWindow* __restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = wins[idx[i]];
}
__global__ void poa_scatter_results_kernel(const u32* __restrict__ idx, u32 n, const u32* __restrict__ rlen,
                                           const u32* __restrict__ rstatus, u32* __restrict__ len, u32* __restrict__ status) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    len[idx[i]] = rlen[i];
    status[idx[i]] = rstatus[i];
  }
}

}  // namespace

// Core of a batch, everything resident in HBM: windows, (begin-sorted) layer descriptors, every base / quality source
// named by `src`, and the outputs (consensus bytes at each window's out_off, d_len / d_status per window).  Every
// window first goes through the banded LDS kernel (poa2.hip) with a 64-column band; windows whose alignment touches
// the band edge are repeated with 128 and then 256 columns, and what is left (or beyond a limit) is re-run by the
// full-matrix kernel above, so a status >= 2 in the result means the window is beyond ALL of them.  Only the
// per-window status words cross PCIe (the escalation lists are decided on the host); h_status receives them.
void poa_run_dev(Engine& e, const PoaWindow* d_wins, const PoaLayer* d_lays, u32 n_windows, const PoaSrc& src, u32 max_bb,
                 u32 max_len, int m, int n, int g, int trim, u8* d_out, u32* d_len, u32* d_status,
                 std::vector<u32>& h_status, double* device_ms, bool allow_full) {
  h_status.assign(n_windows, 0);
  if (n_windows == 0) return;
  hipStream_t s = e.stream;
  // limits: nodes <= nmax, layer length <= lmax (windows beyond them come back with status 2 / 4)
  PoaBatchDev b{};
  b.lmax = std::min<u32>(kPoaMaxSeq, std::max<u32>(64, ((max_len + 63) / 64) * 64));
  b.nmax = std::min<u32>(8192, std::max<u32>(512, max_bb * 6));
  if (const char* ev = knob("RVN_POA_NMAX_MULT")) b.nmax = std::min<u32>(8192, std::max<u32>(512, max_bb * static_cast<u32>(std::atoi(ev))));  // footprint experiments
  b.m = m;
  b.n = n;
  b.g = g;
  b.trim = trim;
  b.n_windows = n_windows;
  b.src = src;
  unsigned long long* d_phase = e.q_start.get<unsigned long long>(16);  // [0..7] phases + cells, [8] work counter, [10..15] kernel statistics
  RVN_HIP(hipMemsetAsync(d_phase, 0, 128, s));
  // heaviest windows first: cost ~ number of layers
  const u32 esc_cap = std::min<u32>(n_windows, 1u << 16);
  u32* d_sk = e.poa_sched.get<u32>(4 * static_cast<size_t>(n_windows) + 8 + esc_cap + 4);
  u32* d_sk1 = d_sk + n_windows + 1;
  u32* d_sv = d_sk1 + n_windows + 1;
  u32* d_sv1 = d_sv + n_windows + 1;
  u32* d_esc = d_sv1 + n_windows + 1;
  poa_sched_keys_kernel<<<div_up(n_windows, 256), 256, 0, s>>>(d_wins, n_windows, d_sk, d_sv);
  RVN_LAUNCH_CHECK();
  const int which = radix_sort_pairs_u32_u32(d_sk, d_sk1, d_sv, d_sv1, n_windows, 16, e.sort_tmp, e.scan_tmp, s,
                                             kKPileSortUp, kKPileSortDown, false);
  b.sched = which ? d_sv1 : d_sv;
  b.next = reinterpret_cast<u32*>(d_phase + 8);
  b.wins = d_wins;
  b.layers = d_lays;
  b.out = d_out;
  b.out_len = d_len;
  b.status = d_status;
  b.phase_cycles = d_phase;
  b.probe = knob("RVN_POA_BAND_PROBE") ? 1u : 0u;
  RVN_HIP(hipEventRecord(e.ev0, s));
  // the banded kernels keep scores as int16 (and add the match / mismatch / gap terms as packed int16): scoring
  // parameters far beyond spoa's usual single digits go straight to the int32 full-matrix kernel
  const auto mag = [](int x) { return x < 0 ? -x : x; };
  const bool int16_ok = mag(m) <= 24 && mag(n) <= 24 && mag(g) <= 24;
  const int mode = int16_ok ? e.poa_mode : 1;
  // first attempt: rows on lanes with a 32-column band (poa4.hip); mode 9 = poa4.hip alone.  A batch too small to fill the
  // chip with groups of four windows is faster in poa2's one-window-per-wave kernel (10 000 windows of a configs[2] round:
  // 108 ms in poa4's persistent kernel, 66 ms in poa2's; at 24 576 windows poa4 is ahead): below the threshold the default
  // mode goes straight to poa2 (engine option poa_rows_min_windows).
  const u32 v4_min_windows = e.opt.poa_rows_min_windows >= 0 ? static_cast<u32>(e.opt.poa_rows_min_windows) : kPoaRowsMinWindowsDefault;
  const bool v4 = mode == 9 || (mode == 0 && n_windows >= v4_min_windows);
  // the windows the 32-column attempt hands on go to the 64-column window function inside the same launch (poa4.hip,
  // poa4_esc_*) — when the escalation chain is on at all (mode 9 is the first attempt alone)
  u32 esc_head[4] = {};
  u32& esc_pushed = esc_head[0];
  if (v4 && mode == 0 && !knob("RVN_POA_NO_ESC")) {
    RVN_HIP(hipMemsetAsync(d_esc, 0xFF, (static_cast<size_t>(esc_cap) + 4) * 4, s));
    RVN_HIP(hipMemsetAsync(d_esc, 0, 16, s));
    b.esc = d_esc;
    b.esc_cap = esc_cap;
  }
  if (mode == 1) poa_v1_launch(e, b);
  else if (v4) poa_v4_launch(e, b);
  else poa_v2_launch(e, b, mode == 3 ? 2 : (mode == 4 ? 4 : 1));
  RVN_HIP(hipMemcpyAsync(h_status.data(), d_status, static_cast<size_t>(n_windows) * 4, hipMemcpyDeviceToHost, s));
  if (b.esc) RVN_HIP(hipMemcpyAsync(esc_head, d_esc, 16, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  esc_pushed = std::min(esc_pushed, esc_cap);
  if (b.probe) {  // diagnostics only: histogram of the paths' largest distance from the band centre, first attempt
    unsigned long long hist[33] = {};
    u32 polished = 0;
    for (u32 w = 0; w < n_windows; ++w) {
      if ((h_status[w] & 0xFFu) != 1u) continue;
      ++polished;
      const u32 dv = (h_status[w] >> 16) & 0xFFu;
      ++hist[dv > 32 ? 32 : dv];
      h_status[w] = 1;
    }
    std::fprintf(stderr, "[raven_hip] poa band probe: %u polished windows of %u; largest distance from the band centre:", polished, n_windows);
    unsigned long long acc = 0;
    for (int d = 0; d <= 32; ++d) {
      acc += hist[d];
      if (hist[d]) std::fprintf(stderr, " %d:%.4f", d, static_cast<double>(acc) / (polished ? polished : 1));
    }
    std::fprintf(stderr, " (cumulative share)\n");
  }
  e.poa_fallback_windows = 0;
  e.poa_narrow_windows = 0;
  e.poa_wide_windows = esc_head[2];  // (windows that went through the 128-column function inside the first launch)
  e.poa_fullmatrix_windows = 0;
  if (mode == 0) {
    // escalate what the 64-column band could not do: band hits -> 128-column band -> 256 -> full matrix; windows
    // beyond a limit (nodes / in-degree / length) -> full matrix directly
    auto rerun = [&](const std::vector<u32>& redo, int which_kernel) {
      const u32 nr = static_cast<u32>(redo.size());
      PoaWindow* d_rw = e.poa_redo_w.get<PoaWindow>(nr + 1);
      u32* d_ridx = e.poa_redo_i.get<u32>(3 * static_cast<size_t>(nr) + 4);
      u32* d_rlen = d_ridx + nr + 1;
      u32* d_rstatus = d_rlen + nr + 1;
      RVN_HIP(hipMemcpyAsync(d_ridx, redo.data(), nr * 4, hipMemcpyHostToDevice, s));
      poa_gather_windows_kernel<<<div_up(nr, 256), 256, 0, s>>>(d_wins, d_ridx, nr, d_rw);
      RVN_LAUNCH_CHECK();
      PoaBatchDev rb = b;
      rb.wins = d_rw;
      rb.n_windows = nr;
      rb.out_len = d_rlen;
      rb.status = d_rstatus;
      rb.sched = nullptr;
      if (which_kernel == 2 || which_kernel == 4) poa_v2_launch(e, rb, which_kernel);
      else if (which_kernel == 64) poa_v2_launch(e, rb, 1);
      else poa_v1_launch(e, rb);
      poa_scatter_results_kernel<<<div_up(nr, 256), 256, 0, s>>>(d_ridx, nr, d_rlen, d_rstatus, d_len, d_status);
      RVN_LAUNCH_CHECK();
      std::vector<u32> rs(nr);
      RVN_HIP(hipMemcpyAsync(rs.data(), d_rstatus, rs.size() * 4, hipMemcpyDeviceToHost, s));
      RVN_HIP(rvn_stream_sync(s));
      for (u32 i = 0; i < nr; ++i) h_status[redo[i]] = rs[i];
    };
    std::vector<u32> wide, wider, fullm;
    if (v4) {  // a 32-column first attempt: what touched its edge (or is beyond poa4.hip's limits) gets the 64-column band next
      std::vector<u32> narrow;
      u32 why[16] = {};
      for (u32 w = 0; w < n_windows; ++w)
        if ((h_status[w] & 0xFF) == kPoaBandHit && !(h_status[w] & kPoaTried64)) {  // (not queued: no queue, or a full one)
          narrow.push_back(w);
          ++why[(h_status[w] >> 24) & 15u];
        }
      if (knob("RVN_POA_STATS") && !narrow.empty())
        std::fprintf(stderr, "[raven_hip] poa: %zu of %u windows handed on by the first attempt: steps %u, in-degree %u, band step along an in-edge %u, last column outside the bands %u, walk near a band's edge %u, walk met a backpointer it cannot follow %u\n",
                     narrow.size(), n_windows, why[1], why[3], why[7], why[9], why[10], why[11]);
      if (!narrow.empty()) rerun(narrow, 64);
      e.poa_narrow_windows = static_cast<u32>(narrow.size()) + esc_pushed;
    }
    for (u32 w = 0; w < n_windows; ++w) {
      const u32 st = h_status[w] & 0xFF;
      // 7 = a predecessor row had left the 64-column kernel's LDS ring: the wider kernels keep a score copy in HBM
      // for that case (one window in 400 000 at C4 — not worth a full-matrix launch)
      if (st == kPoaBandHit || st == 7u) ((h_status[w] & kPoaTried128) ? wider : wide).push_back(w);  // (128 columns: already tried inside the first launch)
      else if (st >= 2) fullm.push_back(w);
    }
    if (knob("RVN_POA_DEBUG")) {
      for (u32 w : wide)
        std::fprintf(stderr, "[raven_hip] poa: window %u band hit at layer %u\n", w, (h_status[w] >> 8) & 0xFFFFu);
      for (u32 w : fullm) std::fprintf(stderr, "[raven_hip] poa: window %u status %u -> full matrix\n", w, h_status[w]);
    }
    if (!wide.empty()) {  // 128 columns
      rerun(wide, 2);
      e.poa_wide_windows += static_cast<u32>(wide.size());
      for (u32 w : wide) {
        const u32 st = h_status[w] & 0xFF;
        if (st == kPoaBandHit) wider.push_back(w);
        else if (st >= 2) fullm.push_back(w);
      }
    }
    const size_t direct_full = fullm.size();  // limits hit in the first two stages: not a matter of band width
    if (!wider.empty()) {  // 256 columns
      rerun(wider, 4);
      e.poa_fallback_windows = static_cast<u32>(wider.size() + direct_full);
      for (u32 w : wider)
        if ((h_status[w] & 0xFF) >= 2) fullm.push_back(w);
    }
    if (wider.empty()) e.poa_fallback_windows = static_cast<u32>(direct_full);
    if (!fullm.empty() && allow_full) {
      rerun(fullm, 1);
      e.poa_fullmatrix_windows = static_cast<u32>(fullm.size());
    }
  }
  RVN_HIP(hipEventRecord(e.ev1, s));
  RVN_HIP(hipMemcpyAsync(e.poa_phase_cycles, d_phase, 64, hipMemcpyDeviceToHost, s));
  unsigned long long kstats[7] = {};
  const bool want_stats = knob("RVN_POA_STATS") != nullptr;
  if (want_stats) RVN_HIP(hipMemcpyAsync(kstats, d_phase + 9, 56, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  if (want_stats)
    std::fprintf(stderr, "[raven_hip] poa kernel statistics (poa4.hip): descriptor-pass cycles %llu, traceback steps %llu round changes %llu (cycles in them %llu), dp wave-steps %llu (dp cycles %llu, in service points %llu), window set-up cycles %llu\n",
                 kstats[1], kstats[2], kstats[3], kstats[4], kstats[5], e.poa_phase_cycles[1], kstats[0], kstats[6]);
  e.poa_cells_full += e.poa_phase_cycles[6];
  e.poa_cells_band += e.poa_phase_cycles[7];
  e.poa_calls += 1;
  if (device_ms) {
    float ms = 0;
    RVN_HIP(hipEventElapsedTime(&ms, e.ev0, e.ev1));
    *device_ms = ms;
  }
}

// Same batch with the windows / layer descriptors and the results on the host (rvn_poa_consensus_batch).
void poa_run(Engine& e, const std::vector<PoaWindow>& wins, const std::vector<PoaLayer>& lays, const PoaSrc& src,
             u32 max_bb, u32 max_len, int m, int n, int g, int trim, u8* h_out, u64 out_total, u32* h_out_len,
             u32* h_status, double* device_ms, bool allow_full) {
  const u32 n_windows = static_cast<u32>(wins.size());
  if (n_windows == 0) return;
  hipStream_t s = e.stream;
  PoaWindow* d_wins = e.tmp_c.get<PoaWindow>(static_cast<size_t>(n_windows) + 2);
  PoaLayer* d_lays = e.tmp_d.get<PoaLayer>(lays.size() + 1);
  u8* d_out = e.tmp_e.get<u8>(out_total + 16);
  u32* d_len = e.tmp_f.get<u32>(2 * static_cast<size_t>(n_windows) + 4);
  u32* d_status = d_len + n_windows + 1;
  RVN_HIP(hipMemcpyAsync(d_wins, wins.data(), wins.size() * sizeof(PoaWindow), hipMemcpyHostToDevice, s));
  RVN_HIP(hipMemcpyAsync(d_lays, lays.data(), lays.size() * sizeof(PoaLayer), hipMemcpyHostToDevice, s));
  RVN_HIP(hipMemsetAsync(d_len, 0, static_cast<size_t>(n_windows) * 4, s));
  std::vector<u32> st;
  poa_run_dev(e, d_wins, d_lays, n_windows, src, max_bb, max_len, m, n, g, trim, d_out, d_len, d_status, st, device_ms,
              allow_full);
  RVN_HIP(hipMemcpyAsync(h_out_len, d_len, static_cast<size_t>(n_windows) * 4, hipMemcpyDeviceToHost, s));
  RVN_HIP(hipMemcpyAsync(h_out, d_out, out_total, hipMemcpyDeviceToHost, s));
  RVN_HIP(rvn_stream_sync(s));
  for (u32 w = 0; w < n_windows; ++w) h_status[w] = st[w];
}

// windows + begin-sorted layer descriptors of a caller-built batch (one-byte codes on the host)
static void poa_build_host_batch(const u64* h_layer_off, const u32* h_begins, const u32* h_ends, const u32* h_has_qual,
                                 bool have_quals, const u32* h_win_off, u32 n_windows, const u64* h_out_off,
                                 std::vector<PoaWindow>& wins, std::vector<PoaLayer>& lays, u32& max_bb, u32& max_len) {
  const u32 n_layers = h_win_off[n_windows];
  wins.assign(n_windows, PoaWindow{});
  lays.assign(n_layers, PoaLayer{});
  max_bb = 1;
  max_len = 1;
  for (u32 w = 0; w < n_windows; ++w) {
    const u32 f = h_win_off[w], l = h_win_off[w + 1];
    wins[w].layer_first = f;
    wins[w].n_layers = l - f;
    wins[w].out_off = static_cast<u32>(h_out_off[w]);
    wins[w].out_cap = static_cast<u32>(h_out_off[w + 1] - h_out_off[w]);
    // racon: rank = stable sort of layers 1.. by begin position
    std::vector<u32> rank(l - f);
    for (u32 i = 0; i < l - f; ++i) rank[i] = f + i;
    if (l - f > 1)
      std::stable_sort(rank.begin() + 1, rank.end(), [&](u32 a, u32 b) { return h_begins[a] < h_begins[b]; });
    for (u32 i = 0; i < l - f; ++i) {
      const u32 src = rank[i];
      PoaLayer& L = lays[f + i];
      L = PoaLayer{};
      L.code_off = h_layer_off[src];
      L.qual_off = h_layer_off[src];
      L.len = static_cast<u32>(h_layer_off[src + 1] - h_layer_off[src]);
      L.begin = h_begins[src];
      L.end = h_ends[src];
      L.flags = (have_quals && h_has_qual && h_has_qual[src]) ? kLayerQual : 0u;
      poa_layer_linear_way(L);
      if (i == 0) max_bb = std::max(max_bb, L.len);
      max_len = std::max(max_len, L.len);
    }
  }
}

// Host entry: see rvn_poa_consensus_batch in raven_hip.h (caller-built windows: one-byte codes on the host).
void poa_consensus_batch(Engine& e, const u8* h_codes, const u8* h_quals, const u64* h_layer_off,
                         const u32* h_begins, const u32* h_ends, const u32* h_has_qual, const u32* h_win_off,
                         u32 n_windows, int m, int n, int g, int trim, u8* h_out, const u64* h_out_off,
                         u32* h_out_len, u32* h_status, double* device_ms) {
  if (n_windows == 0) return;
  hipStream_t s = e.stream;
  const u32 n_layers = h_win_off[n_windows];
  const u64 total = h_layer_off[n_layers];
  std::vector<PoaWindow> wins;
  std::vector<PoaLayer> lays;
  u32 max_bb = 1, max_len = 1;
  poa_build_host_batch(h_layer_off, h_begins, h_ends, h_has_qual, h_quals != nullptr, h_win_off, n_windows, h_out_off, wins,
                       lays, max_bb, max_len);
  u8* d_codes = e.tmp_a.get<u8>(total + 16);
  u8* d_quals = h_quals ? e.tmp_b.get<u8>(total + 16) : nullptr;
  RVN_HIP(hipMemcpyAsync(d_codes, h_codes, total, hipMemcpyHostToDevice, s));
  if (d_quals) RVN_HIP(hipMemcpyAsync(d_quals, h_quals, total, hipMemcpyHostToDevice, s));
  PoaSrc src{};
  src.codes = d_codes;
  src.quals = d_quals;
  poa_run(e, wins, lays, src, max_bb, max_len, m, n, g, trim, h_out, h_out_off[n_windows], h_out_len, h_status,
          device_ms);
}

#ifdef RVN_TEST_HOOKS
// The rows-on-lanes banded kernel (poa4.hip) stepped through on the HOST by the wavefront emulator: same batch
// description as poa_consensus_batch, first attempt only (status 8 / 7 = the window needs the wider kernels).  Test
// infrastructure for the CPU suite; needs no GPU and no engine.
void poa_banded_emulate(const u8* h_codes, const u8* h_quals, const u64* h_layer_off, const u32* h_begins,
                        const u32* h_ends, const u32* h_has_qual, const u32* h_win_off, u32 n_windows, int m, int n, int g,
                        int trim, u8* h_out, const u64* h_out_off, u32* h_out_len, u32* h_status, int variant) {
  if (n_windows == 0) return;
  std::vector<PoaWindow> wins;
  std::vector<PoaLayer> lays;
  u32 max_bb = 1, max_len = 1;
  poa_build_host_batch(h_layer_off, h_begins, h_ends, h_has_qual, h_quals != nullptr, h_win_off, n_windows, h_out_off, wins,
                       lays, max_bb, max_len);
  PoaSrc src{};
  src.codes = h_codes;
  src.quals = h_quals;
  // variant 4: the per-round launches (graph side / alignment side), 5: the persistent kernel
  poa_v4_emulate(wins, lays, src, max_bb, max_len, m, n, g, trim, h_out, h_out_len, h_status, variant == 5);
}

#endif  // RVN_TEST_HOOKS

}  // namespace rvn
