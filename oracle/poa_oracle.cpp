// poa_oracle.cpp — CPU restatement of the window consensus that raven::Polish reaches through
// racon::Polisher::Polish (RavenLib/src/polish.cc:43-51): racon `Window::GenerateConsensus` over spoa's
// partial-order graph, linear-gap global (kNW) alignment (m=3, n=-5, g=-4: polish.hpp:13-17) and
// heaviest-bundle consensus with TGS trimming.
//
// *** TEST INFRASTRUCTURE ONLY (see raven_oracle.cpp header). ***
// PARITY STATUS: parity unpinned — racon (branch `library`, Raven.deps.cmake:39-44) and spoa are not in the
// reference tree; this follows their published sources as recollected:
//   spoa::Graph::{AddAlignment, AddSequence, AddEdge, TopologicalSort, Subgraph, UpdateAlignment,
//                 TraverseHeaviestBundle, BranchCompletion, GenerateConsensus(summary), Node::Coverage}
//   spoa::SisdAlignmentEngine::Linear (the SIMD engine computes the same matrix)
//   racon::Window::GenerateConsensus
// Definitional pins live in tests/test_oracle_poa.py (error-free layers reproduce the truth, majority vote
// beats a wrong backbone, alignment score equals an independent DP on linear graphs, ...).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <stack>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "poa_oracle.h"

namespace poa {

using Alignment = std::vector<std::pair<std::int32_t, std::int32_t>>;  // (node id | -1, sequence pos | -1)

struct Graph {
  struct Edge {
    std::uint32_t tail, head;
    std::vector<std::uint32_t> labels;
    std::int64_t weight;
  };
  struct Node {
    std::uint32_t code;
    std::vector<std::uint32_t> inedges, outedges;  // edge indices, insertion order
    std::vector<std::uint32_t> aligned;            // node ids, insertion order
  };
  std::vector<Node> nodes;
  std::vector<Edge> edges;
  std::vector<std::uint32_t> rank_to_node;
  std::uint32_t num_sequences = 0;
  std::vector<std::uint32_t> consensus;  // node ids
  // debug trace of the last AddAlignment: path node ids in path order and the DP node each was aligned to (-1: none)
  std::vector<std::int32_t> path_nodes, path_aligned;

  std::uint32_t AddNode(std::uint32_t code) {
    nodes.push_back(Node{code, {}, {}, {}});
    return nodes.size() - 1;
  }
  void AddEdge(std::uint32_t tail, std::uint32_t head, std::int64_t weight) {
    for (auto e : nodes[tail].outedges) {
      if (edges[e].head == head) {
        edges[e].labels.push_back(num_sequences);
        edges[e].weight += weight;
        return;
      }
    }
    edges.push_back(Edge{tail, head, {num_sequences}, weight});
    nodes[tail].outedges.push_back(edges.size() - 1);
    nodes[head].inedges.push_back(edges.size() - 1);
  }
  // returns first node of the run or -1
  std::int32_t AddSequence(const std::uint8_t* codes, const std::vector<std::uint32_t>& weights, std::uint32_t begin,
                           std::uint32_t end) {
    if (begin == end) return -1;
    std::int32_t prev = -1, first = -1;
    for (std::uint32_t i = begin; i < end; ++i) {
      std::int32_t curr = AddNode(codes[i]);
      if (first < 0) first = curr;
      if (prev >= 0) AddEdge(prev, curr, weights[i - 1] + weights[i]);
      prev = curr;
    }
    return first;
  }
  void AddAlignment(const Alignment& alignment, const std::uint8_t* codes, std::uint32_t len,
                    const std::vector<std::uint32_t>& weights) {
    if (len == 0) return;
    if (alignment.empty()) {
      AddSequence(codes, weights, 0, len);
      ++num_sequences;
      TopologicalSort();
      return;
    }
    std::vector<std::uint32_t> valid;
    for (const auto& it : alignment)
      if (it.second != -1) valid.push_back(it.second);
    std::uint32_t tmp = nodes.size();
    std::int32_t begin = AddSequence(codes, weights, 0, valid.front());
    std::int32_t prev = tmp == nodes.size() ? -1 : static_cast<std::int32_t>(nodes.size() - 1);
    path_nodes.clear();
    path_aligned.clear();
    for (std::uint32_t x = tmp; x < nodes.size(); ++x) {
      path_nodes.push_back(x);
      path_aligned.push_back(-1);
    }
    std::uint32_t suffix_first = nodes.size();
    std::int32_t last = AddSequence(codes, weights, valid.back() + 1, len);
    std::uint32_t suffix_end = nodes.size();
    for (const auto& it : alignment) {
      if (it.second == -1) continue;
      std::uint32_t code = codes[it.second];
      std::int32_t curr = -1;
      if (it.first == -1) {
        curr = AddNode(code);
      } else {
        std::uint32_t jt = it.first;
        if (nodes[jt].code == code) {
          curr = jt;
        } else {
          for (auto kt : nodes[jt].aligned) {
            if (nodes[kt].code == code) {
              curr = kt;
              break;
            }
          }
          if (curr < 0) {
            curr = AddNode(code);
            auto al = nodes[jt].aligned;  // copy: the loop mutates other nodes' lists only
            for (auto kt : al) {
              nodes[kt].aligned.push_back(curr);
              nodes[curr].aligned.push_back(kt);
            }
            nodes[jt].aligned.push_back(curr);
            nodes[curr].aligned.push_back(jt);
          }
        }
      }
      if (begin < 0) begin = curr;
      if (prev >= 0) AddEdge(prev, curr, weights[it.second - 1] + weights[it.second]);
      prev = curr;
      path_nodes.push_back(curr);
      path_aligned.push_back(it.first);
    }
    for (std::uint32_t x = suffix_first; x < suffix_end; ++x) {
      path_nodes.push_back(x);
      path_aligned.push_back(-1);
    }
    if (last >= 0) AddEdge(prev, last, weights[valid.back()] + weights[valid.back() + 1]);
    ++num_sequences;
    TopologicalSort();
  }
  void TopologicalSort() {
    rank_to_node.clear();
    std::vector<std::uint8_t> marks(nodes.size(), 0);
    std::vector<bool> ignored(nodes.size(), false);
    std::stack<std::uint32_t> stack;
    for (std::uint32_t it = 0; it < nodes.size(); ++it) {
      if (marks[it] != 0) continue;
      stack.push(it);
      while (!stack.empty()) {
        auto curr = stack.top();
        bool is_valid = true;
        if (marks[curr] != 2) {
          for (auto e : nodes[curr].inedges) {
            if (marks[edges[e].tail] != 2) {
              stack.push(edges[e].tail);
              is_valid = false;
            }
          }
          if (!ignored[curr]) {
            for (auto jt : nodes[curr].aligned) {
              if (marks[jt] != 2) {
                stack.push(jt);
                ignored[jt] = true;
                is_valid = false;
              }
            }
          }
          if (is_valid) {
            marks[curr] = 2;
            if (!ignored[curr]) {
              rank_to_node.push_back(curr);
              for (auto jt : nodes[curr].aligned) rank_to_node.push_back(jt);
            }
          } else {
            marks[curr] = 1;
          }
        }
        if (is_valid) stack.pop();
      }
    }
  }
  // spoa Graph::Subgraph(begin, end, &mapping): ancestors of node `end` with id >= begin
  // parent_rank != nullptr: the subgraph's rows keep the parent's order (what the device does: it marks the members and
  // walks the parent's ranks) instead of spoa's own topological sort of the copy.
  Graph Subgraph(std::uint32_t begin, std::uint32_t end, std::vector<std::uint32_t>* sub_to_graph,
                 const std::vector<std::uint32_t>* parent_rank = nullptr) const {
    std::vector<bool> in(nodes.size(), false);
    std::stack<std::uint32_t> stack;
    stack.push(end);
    while (!stack.empty()) {
      auto curr = stack.top();
      stack.pop();
      if (!in[curr] && curr >= begin) {
        for (auto e : nodes[curr].inedges) stack.push(edges[e].tail);
        for (auto jt : nodes[curr].aligned) stack.push(jt);
        in[curr] = true;
      }
    }
    Graph sub;
    sub_to_graph->clear();
    std::vector<std::int32_t> g2s(nodes.size(), -1);
    for (std::uint32_t it = 0; it < nodes.size(); ++it) {
      if (!in[it]) continue;
      g2s[it] = sub.AddNode(nodes[it].code);
      sub_to_graph->push_back(it);
    }
    for (std::uint32_t it = 0; it < nodes.size(); ++it) {
      if (!in[it]) continue;
      for (auto e : nodes[it].inedges)
        if (g2s[edges[e].tail] >= 0) sub.AddEdge(g2s[edges[e].tail], g2s[it], edges[e].weight);
      for (auto kt : nodes[it].aligned)
        if (g2s[kt] >= 0) sub.nodes[g2s[it]].aligned.push_back(g2s[kt]);
    }
    sub.TopologicalSort();
    if (parent_rank) {
      std::sort(sub.rank_to_node.begin(), sub.rank_to_node.end(), [&](std::uint32_t a, std::uint32_t b) {
        return (*parent_rank)[(*sub_to_graph)[a]] < (*parent_rank)[(*sub_to_graph)[b]];
      });
    }
    return sub;
  }
  std::uint32_t Coverage(std::uint32_t id) const {
    std::unordered_set<std::uint32_t> labels;
    for (auto e : nodes[id].inedges) labels.insert(edges[e].labels.begin(), edges[e].labels.end());
    for (auto e : nodes[id].outedges) labels.insert(edges[e].labels.begin(), edges[e].labels.end());
    return labels.size();
  }
  std::uint32_t BranchCompletion(std::uint32_t rank, std::vector<std::int64_t>* scores,
                                 std::vector<std::int32_t>* predecessors) {
    auto start = rank_to_node[rank];
    for (auto e : nodes[start].outedges)
      for (auto f : nodes[edges[e].head].inedges)
        if (edges[f].tail != start) (*scores)[edges[f].tail] = -1;
    std::int32_t max = -1;
    for (std::uint32_t i = rank + 1; i < rank_to_node.size(); ++i) {
      auto it = rank_to_node[i];
      (*scores)[it] = -1;
      (*predecessors)[it] = -1;
      for (auto e : nodes[it].inedges) {
        const auto& jt = edges[e];
        if ((*scores)[jt.tail] == -1) continue;
        if (((*scores)[it] < jt.weight) ||
            ((*scores)[it] == jt.weight && (*scores)[(*predecessors)[it]] <= (*scores)[jt.tail])) {
          (*scores)[it] = jt.weight;
          (*predecessors)[it] = jt.tail;
        }
      }
      if ((*predecessors)[it] != -1) (*scores)[it] += (*scores)[(*predecessors)[it]];
      if (max == -1 || (*scores)[max] < (*scores)[it]) max = it;
    }
    return max;
  }
  void TraverseHeaviestBundle() {
    consensus.clear();
    if (rank_to_node.empty()) return;
    std::vector<std::int32_t> predecessors(nodes.size(), -1);
    std::vector<std::int64_t> scores(nodes.size(), -1);
    std::int32_t max = -1;
    for (auto it : rank_to_node) {
      for (auto e : nodes[it].inedges) {
        const auto& jt = edges[e];
        if ((scores[it] < jt.weight) ||
            (scores[it] == jt.weight && scores[predecessors[it]] <= scores[jt.tail])) {
          scores[it] = jt.weight;
          predecessors[it] = jt.tail;
        }
      }
      if (predecessors[it] != -1) scores[it] += scores[predecessors[it]];
      if (max == -1 || scores[max] < scores[it]) max = it;
    }
    if (!nodes[max].outedges.empty()) {
      std::vector<std::uint32_t> node_id_to_rank(nodes.size(), 0);
      for (std::uint32_t i = 0; i < rank_to_node.size(); ++i) node_id_to_rank[rank_to_node[i]] = i;
      while (!nodes[max].outedges.empty()) max = BranchCompletion(node_id_to_rank[max], &scores, &predecessors);
    }
    while (predecessors[max] != -1) {
      consensus.push_back(max);
      max = predecessors[max];
    }
    consensus.push_back(max);
    std::reverse(consensus.begin(), consensus.end());
  }
};

// End node of an alignment among equal scores: 0 = spoa's rule, the first end node in rank order; 1 / 2 = the one with the
// smallest / largest node id.  1 is what the device kernels do since round 5 (a rule that does not depend on the order of
// the rows; with the device's row order it reproduces spoa's consensus on 19 998 of 20 000 C4-like windows, rule 0 in
// that order on 19 948).  Per CALL and per thread: set from the flags word of orc_poa_window, restored when it returns.
static thread_local int g_end_tie_rule = 0;
// spoa SisdAlignmentEngine::Linear, AlignmentType::kNW
static Alignment AlignNW(const std::uint8_t* seq, std::uint32_t len, const Graph& graph, std::int8_t m, std::int8_t n,
                         std::int8_t g, std::int32_t* score_out = nullptr) {
  if (graph.nodes.empty() || len == 0) return {};
  const std::uint32_t w = len + 1;
  const std::uint32_t rows = graph.rank_to_node.size() + 1;
  const std::int32_t kNegInf = std::numeric_limits<std::int32_t>::min() + 1024;
  std::vector<std::int32_t> H(static_cast<std::size_t>(rows) * w);
  std::vector<std::uint32_t> node_id_to_rank(graph.nodes.size(), 0);
  for (std::uint32_t i = 0; i < graph.rank_to_node.size(); ++i) node_id_to_rank[graph.rank_to_node[i]] = i;
  for (std::uint32_t j = 0; j < w; ++j) H[j] = static_cast<std::int32_t>(j) * g;
  for (std::uint32_t i = 1; i < rows; ++i) {
    const auto& node = graph.nodes[graph.rank_to_node[i - 1]];
    std::int32_t penalty = node.inedges.empty() ? 0 : kNegInf;
    for (auto e : node.inedges) {
      std::uint32_t pred_i = node_id_to_rank[graph.edges[e].tail] + 1;
      penalty = std::max(penalty, H[static_cast<std::size_t>(pred_i) * w]);
    }
    H[static_cast<std::size_t>(i) * w] = penalty + g;
  }
  std::int32_t max_score = kNegInf;
  std::uint32_t max_i = 0, max_j = 0;
  for (std::uint32_t i = 1; i < rows; ++i) {
    const auto& node = graph.nodes[graph.rank_to_node[i - 1]];
    std::int32_t* H_row = &H[static_cast<std::size_t>(i) * w];
    std::uint32_t pred_i = node.inedges.empty() ? 0 : node_id_to_rank[graph.edges[node.inedges[0]].tail] + 1;
    const std::int32_t* H_pred = &H[static_cast<std::size_t>(pred_i) * w];
    for (std::uint32_t j = 1; j < w; ++j) {
      std::int32_t s = node.code == seq[j - 1] ? m : n;
      H_row[j] = std::max(H_pred[j - 1] + s, H_pred[j] + g);
    }
    for (std::uint32_t p = 1; p < node.inedges.size(); ++p) {
      pred_i = node_id_to_rank[graph.edges[node.inedges[p]].tail] + 1;
      H_pred = &H[static_cast<std::size_t>(pred_i) * w];
      for (std::uint32_t j = 1; j < w; ++j) {
        std::int32_t s = node.code == seq[j - 1] ? m : n;
        H_row[j] = std::max(H_pred[j - 1] + s, std::max(H_row[j], H_pred[j] + g));
      }
    }
    for (std::uint32_t j = 1; j < w; ++j) H_row[j] = std::max(H_row[j - 1] + g, H_row[j]);
    if (node.outedges.empty() &&
        (max_score < H_row[w - 1] ||
         (g_end_tie_rule && max_score == H_row[w - 1] && max_i != 0 &&
          (g_end_tie_rule == 1 ? graph.rank_to_node[i - 1] < graph.rank_to_node[max_i - 1]
                               : graph.rank_to_node[i - 1] > graph.rank_to_node[max_i - 1])))) {
      max_score = H_row[w - 1];
      max_i = i;
      max_j = w - 1;
    }
  }
  if (score_out) *score_out = max_score;
  Alignment alignment;
  std::uint32_t i = max_i, j = max_j, prev_i = 0, prev_j = 0;
  while (!(i == 0 && j == 0)) {
    const std::int32_t H_ij = H[static_cast<std::size_t>(i) * w + j];
    bool found = false;
    if (i != 0 && j != 0) {
      const auto& it = graph.nodes[graph.rank_to_node[i - 1]];
      std::int32_t match_cost = it.code == seq[j - 1] ? m : n;
      std::uint32_t pred_i = it.inedges.empty() ? 0 : node_id_to_rank[graph.edges[it.inedges[0]].tail] + 1;
      if (H_ij == H[static_cast<std::size_t>(pred_i) * w + (j - 1)] + match_cost) {
        prev_i = pred_i;
        prev_j = j - 1;
        found = true;
      } else {
        for (std::uint32_t p = 1; p < it.inedges.size(); ++p) {
          pred_i = node_id_to_rank[graph.edges[it.inedges[p]].tail] + 1;
          if (H_ij == H[static_cast<std::size_t>(pred_i) * w + (j - 1)] + match_cost) {
            prev_i = pred_i;
            prev_j = j - 1;
            found = true;
            break;
          }
        }
      }
    }
    if (!found && i != 0) {
      const auto& it = graph.nodes[graph.rank_to_node[i - 1]];
      std::uint32_t pred_i = it.inedges.empty() ? 0 : node_id_to_rank[graph.edges[it.inedges[0]].tail] + 1;
      if (H_ij == H[static_cast<std::size_t>(pred_i) * w + j] + g) {
        prev_i = pred_i;
        prev_j = j;
        found = true;
      } else {
        for (std::uint32_t p = 1; p < it.inedges.size(); ++p) {
          pred_i = node_id_to_rank[graph.edges[it.inedges[p]].tail] + 1;
          if (H_ij == H[static_cast<std::size_t>(pred_i) * w + j] + g) {
            prev_i = pred_i;
            prev_j = j;
            found = true;
            break;
          }
        }
      }
    }
    if (!found && j != 0 && H_ij == H[static_cast<std::size_t>(i) * w + j - 1] + g) {
      prev_i = i;
      prev_j = j - 1;
      found = true;
    }
    if (!found) break;  // cannot happen for a consistent matrix
    alignment.emplace_back(i == prev_i ? -1 : static_cast<std::int32_t>(graph.rank_to_node[i - 1]),
                           j == prev_j ? -1 : static_cast<std::int32_t>(j - 1));
    i = prev_i;
    j = prev_j;
  }
  std::reverse(alignment.begin(), alignment.end());
  return alignment;
}

static std::vector<std::uint32_t> Weights(const Layer& l) {
  std::vector<std::uint32_t> w(l.len, 1);
  if (l.qual)
    for (std::uint32_t i = 0; i < l.len; ++i) w[i] = static_cast<std::uint32_t>(l.qual[i]) - 33;
  return w;
}

// The device kernels' incremental order rule (raven_amd/csrc/poa4.hip poa4_update_graph): after a
// layer is added, old nodes keep their relative order and every new node takes the slot right behind the whole aligned
// group (column) of the last DP node its path met.  rank_of: node id -> rank, updated in place; n_old = nodes before
// the layer.  Returns false if the rule would break the topological order (never seen; orc_poa_order_check looks for it).
static bool DeviceOrderUpdate(const Graph& graph, std::uint32_t n_old, std::vector<std::uint32_t>* rank_of_io) {
  std::vector<std::uint32_t>& rank_of = *rank_of_io;
  auto gmax = [&](std::uint32_t v) {
    std::uint32_t r = rank_of[v];
    for (auto a : graph.nodes[v].aligned) if (a < n_old) r = std::max(r, rank_of[a]);
    return r;
  };
  auto gmin = [&](std::uint32_t v) {
    std::uint32_t r = rank_of[v];
    for (auto a : graph.nodes[v].aligned) if (a < n_old) r = std::min(r, rank_of[a]);
    return r;
  };
  std::uint32_t cur_slot = n_old;
  for (std::size_t q = 0; q < graph.path_nodes.size(); ++q)
    if (graph.path_aligned[q] != -1) { cur_slot = gmin(graph.path_aligned[q]); break; }
  std::vector<std::pair<std::uint32_t, std::uint32_t>> news;  // (slot, node id) in path order
  for (std::size_t q = 0; q < graph.path_nodes.size(); ++q) {
    const std::uint32_t curr = graph.path_nodes[q];
    const std::int32_t an = graph.path_aligned[q];
    if (an != -1) {
      if (gmax(an) + 1 < cur_slot) return false;
      cur_slot = gmax(an) + 1;
    }
    if (curr >= n_old) news.emplace_back(cur_slot, curr);
  }
  std::vector<std::uint32_t> nr(graph.nodes.size(), 0), slots;
  for (auto& p : news) slots.push_back(p.first);
  for (std::uint32_t v = 0; v < n_old; ++v)
    nr[v] = rank_of[v] + static_cast<std::uint32_t>(std::upper_bound(slots.begin(), slots.end(), rank_of[v]) - slots.begin());
  for (std::size_t t = 0; t < news.size(); ++t) nr[news[t].second] = news[t].first + t;
  rank_of = nr;
  for (const auto& e : graph.edges)
    if (rank_of[e.tail] >= rank_of[e.head]) return false;
  return true;
}

static thread_local int g_order_where = 3;  // diagnostic split of device_order (flags word of orc_poa_window): 1 = alignments, 2 = consensus
// racon Window::GenerateConsensus (TGS). layers[0] is the backbone. Returns polished flag.
// device_order: spoa's DFS rank (Graph::TopologicalSort) is replaced by the device kernels' incremental order wherever
// the order of the rows can decide a tie — the end node of an alignment, the node a traceback prefers among equal
// scores, the start of the heaviest bundle.  Same graph rules, same scores, another valid topological order: it tells
// whether a consensus that differs from spoa's differs because of such a tie and nothing else (tools/poa_parity.py).
// The graph of a window after all its layers (racon Window::GenerateConsensus up to the consensus call); false if the
// device order rule broke down.
// where: 1 = the device's order for the alignments (NW rows, Subgraph rows), 2 = for the consensus, 3 = both (diagnostic
// split: which of the two decides a tie)
bool BuildWindowGraph(const std::vector<Layer>& layers, std::int8_t m, std::int8_t n, std::int8_t g, bool device_order,
                      Graph* out, int where = 3) {
  const Layer& bb = layers.front();
  Graph& graph = *out;
  graph = Graph();
  graph.AddAlignment(Alignment(), bb.codes, bb.len, Weights(bb));
  std::vector<std::uint32_t> node_rank(bb.len);  // device_order: node id -> rank by the device's rule
  for (std::uint32_t i = 0; i < bb.len; ++i) node_rank[i] = i;
  auto impose = [&]() {
    if (!device_order) return;
    for (std::uint32_t v = 0; v < graph.nodes.size(); ++v) graph.rank_to_node[node_rank[v]] = v;
  };
  const bool in_alignment = device_order && (where & 1), in_consensus = device_order && (where & 2);
  std::vector<std::uint32_t> rank(layers.size());
  for (std::uint32_t i = 0; i < layers.size(); ++i) rank[i] = i;
  std::stable_sort(rank.begin() + 1, rank.end(),
                   [&](std::uint32_t lhs, std::uint32_t rhs) { return layers[lhs].begin < layers[rhs].begin; });
  std::uint32_t offset = 0.01 * bb.len;
  for (std::uint32_t j = 1; j < layers.size(); ++j) {
    const Layer& l = layers[rank[j]];
    Alignment alignment;
    if (l.begin < offset && l.end > bb.len - offset) {
      alignment = AlignNW(l.codes, l.len, graph, m, n, g);
    } else {
      std::vector<std::uint32_t> mapping;
      auto subgraph = graph.Subgraph(l.begin, l.end, &mapping, in_alignment ? &node_rank : nullptr);
      alignment = AlignNW(l.codes, l.len, subgraph, m, n, g);
      for (auto& it : alignment)
        if (it.first != -1) it.first = mapping[it.first];
    }
    const std::uint32_t n_old = graph.nodes.size();
    graph.AddAlignment(alignment, l.codes, l.len, Weights(l));
    if (device_order) {
      if (!DeviceOrderUpdate(graph, n_old, &node_rank)) {
        return false;
      }
      if (in_alignment) impose();
    }
  }
  if (in_consensus) impose();
  else if (device_order) graph.TopologicalSort();
  return true;
}

bool WindowConsensus(const std::vector<Layer>& layers, std::int8_t m, std::int8_t n, std::int8_t g, bool trim,
                            std::vector<std::uint8_t>* consensus, std::vector<std::uint32_t>* coverages_out,
                            bool device_order) {
  const Layer& bb = layers.front();
  if (layers.size() < 3) {
    consensus->assign(bb.codes, bb.codes + bb.len);
    return false;
  }
  Graph graph;
  if (!BuildWindowGraph(layers, m, n, g, device_order, &graph, g_order_where)) {
    consensus->clear();
    return false;
  }
  graph.TraverseHeaviestBundle();
  std::vector<std::uint32_t> coverages;
  consensus->clear();
  for (auto id : graph.consensus) {
    consensus->push_back(graph.nodes[id].code);
    std::uint32_t c = graph.Coverage(id);
    for (auto jt : graph.nodes[id].aligned) c += graph.Coverage(jt);
    coverages.push_back(c);
  }
  if (trim) {
    std::uint32_t average_coverage = (layers.size() - 1) / 2;
    std::int32_t begin = 0, end = static_cast<std::int32_t>(consensus->size()) - 1;
    for (; begin < static_cast<std::int32_t>(consensus->size()); ++begin)
      if (coverages[begin] >= average_coverage) break;
    for (; end >= 0; --end)
      if (coverages[end] >= average_coverage) break;
    if (begin < end) {  // else: racon warns "might be chimeric" and keeps the untrimmed consensus
      consensus->assign(consensus->begin() + begin, consensus->begin() + end + 1);
      coverages.assign(coverages.begin() + begin, coverages.begin() + end + 1);
    }
  }
  if (coverages_out) *coverages_out = coverages;
  return true;
}

}  // namespace poa

extern "C" {

// One window. Layers are concatenated: codes (0..3) [total], optional quals (Phred+33) [total] or NULL,
// offsets[n_layers+1], begins/ends[n_layers] (layer 0 = backbone; its begin/end are ignored).
// Returns 1 if polished (>= 3 sequences), 0 if the backbone was returned unchanged.
int orc_poa_window(const std::uint8_t* codes, const std::uint8_t* quals, const std::uint64_t* offsets,
                   const std::uint32_t* begins, const std::uint32_t* ends, std::uint32_t n_layers, int m, int n, int g,
                   int trim, std::uint8_t* out, std::uint32_t out_cap, std::uint32_t* out_len) {
  std::vector<poa::Layer> layers(n_layers);
  for (std::uint32_t i = 0; i < n_layers; ++i) {
    layers[i].codes = codes + offsets[i];
    layers[i].qual = quals ? quals + offsets[i] : nullptr;
    layers[i].len = static_cast<std::uint32_t>(offsets[i + 1] - offsets[i]);
    layers[i].begin = begins[i];
    layers[i].end = ends[i];
  }
  // racon Window::AddLayer rejects begin >= end and positions beyond the backbone
  for (std::uint32_t i = 1; i < n_layers; ++i)
    if (layers[i].len && (begins[i] >= ends[i] || ends[i] >= layers[0].len)) return -1;
  // flags word `trim`: bit 0 = racon's coverage trim, bit 1 = the device's row order, bits 2-3 = end-node tie rule
  // (0 spoa's, 1 smallest node id, 2 largest), bits 4-5 = where the device's order applies (0 = everywhere, 1 = in the
  // alignments only, 2 = in the consensus only).  The two diagnostic settings live for this call on this thread only.
  struct Knobs {
    int tie, where;
    Knobs(int t, int w) : tie(poa::g_end_tie_rule), where(poa::g_order_where) {
      poa::g_end_tie_rule = t;
      poa::g_order_where = w ? w : 3;
    }
    ~Knobs() {
      poa::g_end_tie_rule = tie;
      poa::g_order_where = where;
    }
  } knobs((trim >> 2) & 3, (trim >> 4) & 3);
  std::vector<std::uint8_t> cons;
  bool polished = poa::WindowConsensus(layers, m, n, g, (trim & 1) != 0, &cons, nullptr, (trim & 2) != 0);
  *out_len = cons.size();
  std::memcpy(out, cons.data(), std::min<std::size_t>(cons.size(), out_cap));
  return polished ? 1 : 0;
}

// DEBUG: replays the device kernel's incremental topological-order rule (raven_amd/csrc/poa.hip step 5) next to
// the real graph construction and returns the first layer after which some edge has rank(tail) >= rank(head)
// (-1 if the order stays valid). info[0..3] = tail, head, rank(tail), rank(head) of the first violation.
// Graph-shape statistics gathered while orc_poa_order_check replays the kernel's incremental order (design input for
// the rows-on-lanes POA kernel): per layer, BEFORE the layer is added, over the graph in kernel rank order.
//  [0] layers  [1] sum nodes  [2] sum edges  [3] sum over ranks of max in-degree within the aligned block of 16 ranks
//  [4] the same over blocks of 64 ranks  [5] max lookback (rank(head) - rank(tail))  [6] edges with lookback > 15
//  [7] edges with lookback > 31  [8] rank pairs where bpos decreases  [9] max in-degree  [10..26] in-degree histogram 0..16
//  [27] edges with lookback > 47  [28] sum over ranks of in-degree clipped at 4 block-of-64 max  [29] nodes with in-degree > 4
//  [30] nodes with in-degree > 6 [31] nodes with in-degree > 8
static std::int64_t* g_stats = nullptr;
void orc_poa_stats_buffer(std::int64_t* buf) { g_stats = buf; }
static void poa_graph_stats(const poa::Graph& graph, const std::vector<std::uint32_t>& rank_of,
                            const std::vector<std::uint32_t>& bpos) {
  if (!g_stats) return;
  const std::uint32_t n = graph.nodes.size();
  std::vector<std::uint32_t> deg_by_rank(n, 0), node_at(n, 0);
  for (std::uint32_t v = 0; v < n; ++v) {
    deg_by_rank[rank_of[v]] = graph.nodes[v].inedges.size();
    node_at[rank_of[v]] = v;
  }
  g_stats[0] += 1;
  g_stats[1] += n;
  g_stats[2] += graph.edges.size();
  for (std::uint32_t r0 = 0; r0 < n; r0 += 16) {
    std::uint32_t mx = 0, cnt = 0;
    for (std::uint32_t r = r0; r < std::min(n, r0 + 16); ++r) { mx = std::max(mx, deg_by_rank[r]); ++cnt; }
    g_stats[3] += static_cast<std::int64_t>(mx) * cnt;
  }
  for (std::uint32_t r0 = 0; r0 < n; r0 += 64) {
    std::uint32_t mx = 0, cnt = 0;
    for (std::uint32_t r = r0; r < std::min(n, r0 + 64); ++r) { mx = std::max(mx, deg_by_rank[r]); ++cnt; }
    g_stats[4] += static_cast<std::int64_t>(mx) * cnt;
  }
  for (const auto& e : graph.edges) {
    const std::int64_t lb = static_cast<std::int64_t>(rank_of[e.head]) - rank_of[e.tail];
    g_stats[5] = std::max(g_stats[5], lb);
    if (lb > 15) g_stats[6] += 1;
    if (lb > 31) g_stats[7] += 1;
    if (lb > 47) g_stats[27] += 1;
    if (lb > 23) g_stats[32] += 1;
    const std::int64_t db = static_cast<std::int64_t>(bpos[e.head]) - bpos[e.tail];
    g_stats[33] = std::max(g_stats[33], db);
    if (db > 8) g_stats[34] += 1;
    if (db > 14) g_stats[35] += 1;
  }
  for (std::uint32_t r = 1; r < n; ++r)
    if (bpos[node_at[r]] < bpos[node_at[r - 1]]) g_stats[8] += 1;
  for (std::uint32_t v = 0; v < n; ++v) {
    const std::uint32_t d = graph.nodes[v].inedges.size();
    g_stats[9] = std::max<std::int64_t>(g_stats[9], d);
    g_stats[10 + std::min<std::uint32_t>(d, 16)] += 1;
    if (d > 4) g_stats[29] += 1;
    if (d > 6) g_stats[30] += 1;
    if (d > 8) g_stats[31] += 1;
  }
}

int orc_poa_order_check(const std::uint8_t* codes, const std::uint64_t* offsets, const std::uint32_t* begins,
                        const std::uint32_t* ends, std::uint32_t n_layers, int m, int n, int g, std::int64_t* info) {
  std::vector<poa::Layer> layers(n_layers);
  for (std::uint32_t i = 0; i < n_layers; ++i) {
    layers[i].codes = codes + offsets[i];
    layers[i].qual = nullptr;
    layers[i].len = static_cast<std::uint32_t>(offsets[i + 1] - offsets[i]);
    layers[i].begin = begins[i];
    layers[i].end = ends[i];
  }
  poa::Graph graph;
  const poa::Layer& bb = layers.front();
  graph.AddAlignment(poa::Alignment(), bb.codes, bb.len, poa::Weights(bb));
  std::vector<std::uint32_t> rank_of(bb.len);
  for (std::uint32_t i = 0; i < bb.len; ++i) rank_of[i] = i;
  std::vector<std::uint32_t> bpos(bb.len);  // the kernel's backbone coordinate of a node (anchor column; carried over insertions)
  for (std::uint32_t i = 0; i < bb.len; ++i) bpos[i] = i;
  std::vector<std::uint32_t> rk(n_layers);
  for (std::uint32_t i = 0; i < n_layers; ++i) rk[i] = i;
  std::stable_sort(rk.begin() + 1, rk.end(), [&](std::uint32_t a, std::uint32_t b) { return layers[a].begin < layers[b].begin; });
  std::uint32_t offset = 0.01 * bb.len;
  for (std::uint32_t j = 1; j < n_layers; ++j) {
    const poa::Layer& l = layers[rk[j]];
    poa::Alignment alignment;
    if (l.begin < offset && l.end > bb.len - offset) {
      alignment = poa::AlignNW(l.codes, l.len, graph, m, n, g);
    } else {
      std::vector<std::uint32_t> mapping;
      auto sub = graph.Subgraph(l.begin, l.end, &mapping);
      alignment = poa::AlignNW(l.codes, l.len, sub, m, n, g);
      for (auto& it : alignment)
        if (it.first != -1) it.first = mapping[it.first];
    }
    poa_graph_stats(graph, rank_of, bpos);
    const std::uint32_t n_old = graph.nodes.size();
    graph.AddAlignment(alignment, l.codes, l.len, poa::Weights(l));
    {  // backbone coordinates of the new nodes, as the kernel assigns them
      bpos.resize(graph.nodes.size(), 0);
      std::uint32_t carry = l.begin;
      for (std::size_t q = 0; q < graph.path_nodes.size(); ++q)
        if (graph.path_aligned[q] != -1) { carry = bpos[graph.path_aligned[q]]; break; }
      for (std::size_t q = 0; q < graph.path_nodes.size(); ++q) {
        if (graph.path_aligned[q] != -1) carry = bpos[graph.path_aligned[q]];
        if (static_cast<std::uint32_t>(graph.path_nodes[q]) >= n_old) bpos[graph.path_nodes[q]] = carry;
      }
    }
    // kernel rule
    auto gmax = [&](std::uint32_t v) {
      std::uint32_t r = rank_of[v];
      for (auto a : graph.nodes[v].aligned) if (a < n_old) r = std::max(r, rank_of[a]);
      return r;
    };
    auto gmin = [&](std::uint32_t v) {
      std::uint32_t r = rank_of[v];
      for (auto a : graph.nodes[v].aligned) if (a < n_old) r = std::min(r, rank_of[a]);
      return r;
    };
    std::uint32_t first_old_rank = n_old;
    for (std::size_t q = 0; q < graph.path_nodes.size(); ++q)
      if (graph.path_aligned[q] != -1) { first_old_rank = gmin(graph.path_aligned[q]); break; }
    std::uint32_t cur_slot = first_old_rank;
    std::vector<std::pair<std::uint32_t, std::uint32_t>> news;  // (slot, node id) in path order
    for (std::size_t q = 0; q < graph.path_nodes.size(); ++q) {
      std::uint32_t curr = graph.path_nodes[q];
      std::int32_t an = graph.path_aligned[q];
      if (an != -1) {
        if (gmax(an) + 1 < cur_slot) {  // DEBUG: anchor slot would move backwards
          info[0] = -3; info[1] = an; info[2] = rank_of[an]; info[3] = cur_slot; info[4] = n_old; info[5] = q;
          info[6] = curr; info[7] = q ? graph.path_nodes[q - 1] : -1;
          return static_cast<int>(j);
        }
        cur_slot = gmax(an) + 1;  // after the whole aligned group (column) of the DP node
      }
      if (curr >= n_old) news.emplace_back(cur_slot, curr);
    }
    std::vector<std::uint32_t> nr(graph.nodes.size(), 0);
    std::vector<std::uint32_t> slots;
    for (auto& p : news) slots.push_back(p.first);
    for (std::uint32_t v = 0; v < n_old; ++v) {
      std::uint32_t r = rank_of[v];
      std::uint32_t ub = std::upper_bound(slots.begin(), slots.end(), r) - slots.begin();
      nr[v] = r + ub;
    }
    for (std::size_t t = 0; t < news.size(); ++t) nr[news[t].second] = news[t].first + t;
    rank_of = nr;
    for (const auto& e : graph.edges) {
      if (rank_of[e.tail] >= rank_of[e.head]) {
        info[0] = e.tail; info[1] = e.head; info[2] = rank_of[e.tail]; info[3] = rank_of[e.head];
        info[4] = n_old;
        info[5] = info[6] = info[7] = -9;
        for (std::size_t q = 0; q < graph.path_nodes.size(); ++q)
          if (graph.path_nodes[q] == static_cast<std::int32_t>(e.head)) {
            info[5] = q; info[6] = graph.path_aligned[q]; info[7] = q ? graph.path_nodes[q - 1] : -1;
          }
        return static_cast<int>(j);
      }
    }
    std::vector<std::uint32_t> seen(graph.nodes.size(), 0);
    for (auto r : rank_of) {
      if (r >= graph.nodes.size() || seen[r]) { info[0] = -2; info[1] = r; return static_cast<int>(j); }
      seen[r] = 1;
    }
  }
  return -1;
}


// DEBUG: the tail of a window's final graph (spoa order or device order): for the last `n_tail` ranks: node id, code,
// out-degree, heaviest-path score and predecessor BEFORE branch completion, in-edges (tail:weight ...).  Text to stderr.
int orc_poa_dump_tail(const std::uint8_t* codes, const std::uint64_t* offsets, const std::uint32_t* begins,
                      const std::uint32_t* ends, std::uint32_t n_layers, int m, int n, int g, int device_order, int n_tail) {
  std::vector<poa::Layer> layers(n_layers);
  for (std::uint32_t i = 0; i < n_layers; ++i) {
    layers[i].codes = codes + offsets[i];
    layers[i].qual = nullptr;
    layers[i].len = static_cast<std::uint32_t>(offsets[i + 1] - offsets[i]);
    layers[i].begin = begins[i];
    layers[i].end = ends[i];
  }
  poa::Graph graph;
  poa::BuildWindowGraph(layers, m, n, g, device_order != 0, &graph);
  const std::uint32_t N = graph.rank_to_node.size();
  std::vector<std::int32_t> pred(graph.nodes.size(), -1);
  std::vector<std::int64_t> sc(graph.nodes.size(), -1);
  std::int32_t mx = -1;
  for (auto it : graph.rank_to_node) {
    for (auto e : graph.nodes[it].inedges) {
      const auto& jt = graph.edges[e];
      if (sc[it] < jt.weight || (sc[it] == jt.weight && sc[pred[it]] <= sc[jt.tail])) {
        sc[it] = jt.weight;
        pred[it] = jt.tail;
      }
    }
    if (pred[it] != -1) sc[it] += sc[pred[it]];
    if (mx == -1 || sc[mx] < sc[it]) mx = it;
  }
  if (const char* path = std::getenv("ORC_POA_DUMP_GRAPH")) {  // whole graph: "rank node code out_degree | tail:weight ..."
    if (FILE* f = std::fopen(path, "w")) {
      for (std::uint32_t r = 0; r < N; ++r) {
        const auto v = graph.rank_to_node[r];
        std::fprintf(f, "%u %u %u %zu |", r, v, graph.nodes[v].code & 3, graph.nodes[v].outedges.size());
        for (auto e : graph.nodes[v].inedges) std::fprintf(f, " %u:%lld", graph.edges[e].tail, static_cast<long long>(graph.edges[e].weight));
        std::fprintf(f, "\n");
      }
      std::fclose(f);
    }
  }
  std::fprintf(stderr, "nodes %u edges %zu best node %d (score %lld, out-degree %zu)\n", N, graph.edges.size(), mx,
               static_cast<long long>(sc[mx]), graph.nodes[mx].outedges.size());
  for (std::uint32_t r = N > static_cast<std::uint32_t>(n_tail) ? N - n_tail : 0; r < N; ++r) {
    const auto v = graph.rank_to_node[r];
    std::fprintf(stderr, "rank %u node %u %c out %zu score %lld pred %d aligned[", r, v, "ACGT"[graph.nodes[v].code & 3],
                 graph.nodes[v].outedges.size(), static_cast<long long>(sc[v]), pred[v]);
    for (auto a : graph.nodes[v].aligned) std::fprintf(stderr, " %u", a);
    std::fprintf(stderr, " ] in:");
    for (auto e : graph.nodes[v].inedges) std::fprintf(stderr, " %u:%lld", graph.edges[e].tail, static_cast<long long>(graph.edges[e].weight));
    std::fprintf(stderr, "\n");
  }
  return 0;
}

// NW score of a sequence against the LINEAR graph of another sequence (unit-test hook for AlignNW).
int orc_poa_align_score_linear(const std::uint8_t* target, std::uint32_t tlen, const std::uint8_t* query,
                               std::uint32_t qlen, int m, int n, int g) {
  poa::Graph graph;
  std::vector<std::uint32_t> w(tlen, 1);
  graph.AddAlignment(poa::Alignment(), target, tlen, w);
  std::int32_t score = 0;
  poa::AlignNW(query, qlen, graph, m, n, g, &score);
  return score;
}

}  // extern "C"
