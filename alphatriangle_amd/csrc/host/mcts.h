// Host PUCT tree search: the sequential specification that the HIP batched-tree
// kernels are tested against, and the search used by CPU self-play workers.
//
// Capability parity with the external `trimcts.run_mcts` consumed at
// alphatriangle/rl/self_play/worker.py:273-280 and configured by
// alphatriangle/config/mcts_config.py:21-62: PUCT selection, depth cap,
// Dirichlet root noise, discount, leaf batching and subtree reuse.
#pragma once
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <random>
#include <vector>

#include "../core/atri_core.h"
#include "../core/atri_search.h"
#include "geometry.h"

namespace atri {

struct Slot {
  int32_t kind = kSlotNone;
  int32_t leaf = -1;
  int32_t plen = 0;
  int32_t path_node[kMaxPath];
  int32_t path_action[kMaxPath];
};

// Evaluator: fills logits[n][action_dim] and values[n] for n leaf states.
using EvalFn = std::function<void(const std::vector<State>&, std::vector<float>&, std::vector<float>&)>;

class Tree {
 public:
  std::shared_ptr<const Geometry> geom;
  int A = 0;
  int capacity = 0;
  int n_nodes = 0;
  std::vector<State> states;
  std::vector<NodeMeta> meta;
  std::vector<Edge> edges;
  uint64_t noise_rng = 0x1234;
  int64_t total_simulations = 0;
  int max_depth_seen = 0;

  Tree(std::shared_ptr<const Geometry> g, int cap, uint64_t seed)
      : geom(std::move(g)), A(geom->action_dim), capacity(cap), noise_rng(seed) {
    states.resize(capacity);
    meta.resize(capacity);
    edges.resize((size_t)capacity * A);
  }

  void reset(const State& root) {
    n_nodes = 1;
    states[0] = root;
    NodeMeta m{};
    m.parent = -1;
    m.action = -1;
    m.terminal = root.over;
    meta[0] = m;
  }

  Edge* node_edges(int node) { return &edges[(size_t)node * A]; }
  const Edge* node_edges(int node) const { return &edges[(size_t)node * A]; }

  // Masked softmax of the network logits over the valid actions of `node`,
  // in wave order so the device reproduces the same bits.
  void init_edges(int node, const float* logits, float value) {
    GeomView g = geom->view();
    const State& st = states[node];
    std::vector<uint8_t> valid(A);
    float mx = -INFINITY;
    for (int a = 0; a < A; ++a) {
      valid[a] = action_valid(g, st, a) ? 1 : 0;
      if (valid[a] && logits[a] > mx) mx = logits[a];
    }
    std::vector<float> e(A, 0.0f);
    for (int a = 0; a < A; ++a)
      if (valid[a]) e[a] = exp_det(logits[a] - mx);
    float sum = wave_order_sum(A, [&](int i) { return e[i]; });
    Edge* ed = node_edges(node);
    for (int a = 0; a < A; ++a) {
      ed[a].prior = valid[a] ? e[a] / sum : -1.0f;
      ed[a].n = 0;
      ed[a].w = 0.0f;
      ed[a].child = -1;
    }
    meta[node].expanded = 1;
    meta[node].nn_value = value;
  }

  // Mix Dirichlet(alpha) noise into the root priors (statistical parity with the
  // device kernel only: gamma sampling uses libm).
  void add_root_noise(float alpha, float eps) {
    if (eps <= 0.0f || alpha <= 0.0f) return;
    Edge* ed = node_edges(0);
    std::mt19937_64 gen(rng_next(noise_rng));
    std::gamma_distribution<float> gamma(alpha, 1.0f);
    std::vector<float> eta(A, 0.0f);
    float tot = 0.0f;
    for (int a = 0; a < A; ++a)
      if (ed[a].prior >= 0.0f) {
        eta[a] = gamma(gen);
        tot += eta[a];
      }
    if (tot <= 0.0f) return;
    for (int a = 0; a < A; ++a)
      if (ed[a].prior >= 0.0f) ed[a].prior = (1.0f - eps) * ed[a].prior + eps * (eta[a] / tot);
  }

  int select_action(int node, const SearchParams& sp) const {
    const Edge* ed = node_edges(node);
    const NodeMeta& m = meta[node];
    float sqrt_parent = sqrtf((float)(m.n > 1 ? m.n : 1));
    float fpu = (sp.fpu_parent && m.n > 0) ? m.w / (float)m.n : 0.0f;
    int best = -1;
    float best_score = -INFINITY;
    for (int a = 0; a < A; ++a) {
      if (ed[a].prior < 0.0f) continue;
      float q = ed[a].n > 0 ? ed[a].w / (float)ed[a].n : fpu;
      float sc = puct_score(q, ed[a].prior, sqrt_parent, ed[a].n, sp.cpuct);
      if (sc > best_score) {
        best_score = sc;
        best = a;
      }
    }
    return best;
  }

  void undo_virtual_loss(const Slot& s, const SearchParams& sp) {
    for (int i = 0; i < s.plen; ++i) {
      Edge& e = node_edges(s.path_node[i])[s.path_action[i]];
      e.n -= 1;
      e.w = e.w - sp.virtual_loss;
    }
  }

  Slot select(const SearchParams& sp) {
    GeomView g = geom->view();
    Slot s;
    int node = 0, depth = 0;
    bool vl = sp.leaf_batch > 1;
    while (true) {
      const NodeMeta& m = meta[node];
      if (m.terminal) {
        s.kind = kSlotTerminal;
        s.leaf = node;
        break;
      }
      if (!m.expanded) {  // in-flight leaf of an earlier slot in this batch
        if (vl) undo_virtual_loss(s, sp);
        s.kind = kSlotNone;
        s.plen = 0;
        break;
      }
      if (depth >= sp.max_depth || depth >= kMaxPath) {
        s.kind = kSlotDepthCap;
        s.leaf = node;
        break;
      }
      int a = select_action(node, sp);
      if (a < 0) {  // expanded non-terminal nodes always have a valid action
        s.kind = kSlotDepthCap;
        s.leaf = node;
        break;
      }
      Edge& e = node_edges(node)[a];
      if (e.child < 0 && n_nodes >= capacity) {
        s.kind = kSlotDepthCap;
        s.leaf = node;
        break;
      }
      s.path_node[s.plen] = node;
      s.path_action[s.plen] = a;
      s.plen++;
      if (vl) {
        e.n += 1;
        e.w = e.w + sp.virtual_loss;
      }
      if (e.child < 0) {
        int nn = n_nodes++;
        State st = states[node];
        int placed, cleared;
        apply_action(g, st, a, placed, cleared);
        bool hv = any_valid_action(g, st);
        float reward = finish_step(g, st, placed, cleared, hv);
        states[nn] = st;
        NodeMeta nm{};
        nm.reward = reward;
        nm.parent = node;
        nm.action = (int16_t)a;
        nm.terminal = st.over;
        nm.depth_hint = depth + 1;
        meta[nn] = nm;
        e.child = nn;
        s.leaf = nn;
        s.kind = st.over ? kSlotTerminal : kSlotExpand;
        if (depth + 1 > max_depth_seen) max_depth_seen = depth + 1;
        break;
      }
      node = e.child;
      depth++;
    }
    return s;
  }

  void backup(const Slot& s, float leaf_value, const SearchParams& sp) {
    bool vl = sp.leaf_batch > 1;
    float ret = leaf_value;
    meta[s.leaf].n += 1;
    meta[s.leaf].w = meta[s.leaf].w + ret;
    for (int i = s.plen - 1; i >= 0; --i) {
      int node = s.path_node[i];
      Edge& e = node_edges(node)[s.path_action[i]];
      ret = meta[e.child].reward + sp.discount * ret;
      if (vl) {
        e.w = e.w - sp.virtual_loss;
      } else {
        e.n += 1;
      }
      e.w = e.w + ret;
      meta[node].n += 1;
      meta[node].w = meta[node].w + ret;
    }
  }

  // Make the child reached by `action` the new root by compacting its subtree
  // into fresh arrays (breadth first).  Returns false when the child does not exist.
  bool reroot(int action) {
    if (n_nodes <= 0 || action < 0 || action >= A || !meta[0].expanded) return false;
    int child = node_edges(0)[action].child;
    if (child < 0) return false;
    std::vector<int> order;  // old indices in BFS order
    order.push_back(child);
    std::vector<int> remap(n_nodes, -1);
    remap[child] = 0;
    for (size_t head = 0; head < order.size(); ++head) {
      int o = order[head];
      if (!meta[o].expanded) continue;
      const Edge* ed = node_edges(o);
      for (int a = 0; a < A; ++a)
        if (ed[a].child >= 0) {
          remap[ed[a].child] = (int)order.size();
          order.push_back(ed[a].child);
        }
    }
    std::vector<State> ns(capacity);
    std::vector<NodeMeta> nm(capacity);
    std::vector<Edge> ne((size_t)capacity * A);
    for (size_t i = 0; i < order.size(); ++i) {
      int o = order[i];
      ns[i] = states[o];
      nm[i] = meta[o];
      nm[i].parent = i == 0 ? -1 : remap[meta[o].parent];
      if (meta[o].expanded) {
        const Edge* src = node_edges(o);
        Edge* dst = &ne[i * (size_t)A];
        for (int a = 0; a < A; ++a) {
          dst[a] = src[a];
          if (src[a].child >= 0) dst[a].child = remap[src[a].child];
        }
      }
    }
    nm[0].action = -1;
    states.swap(ns);
    meta.swap(nm);
    edges.swap(ne);
    n_nodes = (int)order.size();
    return true;
  }

  // Full search from the current root.  Returns root visit counts per action.
  std::vector<int32_t> search(const SearchParams& sp, const EvalFn& eval) {
    std::vector<State> batch;
    std::vector<float> logits, values;
    if (!meta[0].terminal && !meta[0].expanded) {
      batch.assign(1, states[0]);
      eval(batch, logits, values);
      init_edges(0, logits.data(), values[0]);
      meta[0].n = 1;
      meta[0].w = values[0];
    }
    if (!meta[0].terminal) add_root_noise(sp.dirichlet_alpha, sp.dirichlet_epsilon);
    int done = 0;
    int L = sp.leaf_batch < 1 ? 1 : sp.leaf_batch;
    std::vector<Slot> slots;
    while (done < sp.max_simulations && !meta[0].terminal) {
      int nb = std::min(L, sp.max_simulations - done);
      slots.clear();
      batch.clear();
      for (int i = 0; i < nb; ++i) {
        slots.push_back(select(sp));
        if (slots.back().kind == kSlotExpand) batch.push_back(states[slots.back().leaf]);
      }
      if (!batch.empty()) eval(batch, logits, values);
      int bi = 0;
      for (auto& s : slots) {
        if (s.kind == kSlotExpand) {
          init_edges(s.leaf, logits.data() + (size_t)bi * A, values[bi]);
          backup(s, values[bi], sp);
          ++bi;
        } else if (s.kind == kSlotTerminal) {
          backup(s, 0.0f, sp);
        } else if (s.kind == kSlotDepthCap) {
          backup(s, meta[s.leaf].nn_value, sp);
        }
      }
      done += nb;
    }
    total_simulations += done;
    std::vector<int32_t> visits(A, 0);
    if (meta[0].expanded) {
      const Edge* ed = node_edges(0);
      for (int a = 0; a < A; ++a) visits[a] = ed[a].prior >= 0.0f ? ed[a].n : 0;
    }
    return visits;
  }
};

}  // namespace atri
