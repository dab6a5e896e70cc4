// HBM-resident batched game engine and PUCT tree search for gfx950.
//
// Replaces, on the GPU, the two native libraries the reference calls into from its
// CPU workers: the game core (alphatriangle/rl/self_play/worker.py:190,377-379) and
// the tree search (alphatriangle/rl/self_play/worker.py:273-280), plus the per-move
// Python glue around them (policy target / action sampling:
// alphatriangle/rl/self_play/mcts_helpers.py:19-189; n-step returns:
// alphatriangle/rl/self_play/worker.py:410-485).
//
// Execution model: thousands of independent games per GPU, ONE 64-lane wavefront per
// game tree (workgroup = one wave, so workgroup barriers are free and every tree's
// memory is private to its wave).  Per simulation wave the host launches
// tree_wave_kernel -> network forward over all leaves -> tree_wave_kernel ...; nothing
// leaves the device between the root state and the visit counts.
//   * PUCT scan: lanes stride over the node's dense edge array (16-byte edges, one
//     dwordx4 load per lane), wave argmax by xor-butterfly with lowest-action tie break.
//   * Expansion: the wave steps the parent's 64-byte state with the shared rule core
//     (csrc/core/atri_core.h), finds the valid mask in parallel, and writes the leaf's
//     network input row.
//   * All float math that decides the search uses the deterministic helpers of
//     atri_search.h in wave order, so a device tree equals the host tree bit for bit.
#include "common.hpp"
#include "../core/atri_core.h"
#include "../core/atri_search.h"

using namespace atri;

#pragma clang fp contract(off)

// The file is compiled twice into the kernel library: with the 64-byte game state (boards up to 256 cells; C entry points
// atri_*) and, with -DATRI_MAX_WORDS=16, with the 160-byte state of the wide host engine (boards up to 1024 cells; entry
// points atriw_*).  Kernels and structures of the two builds live in different namespaces.
#if ATRI_MAX_WORDS != 4
#define ATRI_ENG_NS atri_engine_wide
#define ATRI_ENG(name) atriw_##name
// a 160-byte state per lane: the one-wave workgroups get the whole register file instead of the default 128 registers
#define ATRI_ENG_BOUNDS __launch_bounds__(64)
#else
#define ATRI_ENG_NS atri_engine
#define ATRI_ENG(name) atri_##name
#define ATRI_ENG_BOUNDS
#endif

namespace ATRI_ENG_NS {

struct TreeBufs {
  State* states[2];    // [G][C]
  NodeMeta* meta[2];   // [G][C]
  Edge* edges[2];      // [G][C][A]
  int32_t* cur;        // [G] active arena
  int32_t* n_nodes;    // [G]
  int32_t* slot_kind;  // [G*L]
  int32_t* slot_leaf;  // [G*L]
  int32_t* slot_plen;  // [G*L]
  int32_t* path_node;  // [G*L][P]
  int32_t* path_action;  // [G*L][P]
  float* nn_grid;      // [G*L][cells]
  float* nn_other;     // [G*L][other_dim]
  const float* nn_logits;  // [G*L][logit_stride]
  const float* nn_value;   // [G*L]
  int32_t* order;      // [G][C] scratch for re-rooting
  // per-tree counters, private to the tree's wave (plain read-modify-write, summed on the host once per move):
  // [0..3] leaf slots by kind (none / expand / terminal / depth cap), [4] network evaluations that were not finite
  int32_t* census;     // [G][kCensus]
  // Batch compaction: with a live-row counter (live_cursor != nullptr) the selecting wave only marks its slots; a scan
  // over the slot kinds then numbers the slots that need an evaluation in slot order (leaf_rows_kernel: slot_row, live
  // count) and leaf_features_kernel writes their feature rows densely.  The network kernels that follow read the count
  // on the device.  Without a counter a slot's row is the slot itself and the wave writes the features directly.
  // (One atomicAdd per leaf on a shared counter instead of the scan was measured first: 4096 same-address atomics
  // serialise and cost ~110 us per wave.)
  int32_t* slot_row;     // [G*L]
  int32_t* live_cursor;  // [1] number of dense rows written by the last selecting wave
  int32_t G, C, A, L, P, other_dim, logit_stride, pad;
};
constexpr int kCensus = 8;
constexpr int kCensusNonFinite = 4;

__device__ __forceinline__ State* tree_states(const TreeBufs& tb, int g) { return tb.states[tb.cur[g]] + (size_t)g * tb.C; }
__device__ __forceinline__ NodeMeta* tree_meta(const TreeBufs& tb, int g) { return tb.meta[tb.cur[g]] + (size_t)g * tb.C; }
__device__ __forceinline__ Edge* tree_edges(const TreeBufs& tb, int g) { return tb.edges[tb.cur[g]] + (size_t)g * tb.C * tb.A; }

// ------------------------------------------------------------------ wave helpers
// Lane-strided walk over the action space without integer division: (slot, pos) of action lane + 64k.
struct ActionCursor {
  int slot, pos;
  __device__ __forceinline__ ActionCursor(const GeomView& g, int lane) : slot(0), pos(lane) { wrap(g); }
  __device__ __forceinline__ void wrap(const GeomView& g) {
    while (pos >= g.cells) {
      pos -= g.cells;
      ++slot;
    }
  }
  __device__ __forceinline__ void advance(const GeomView& g) {
    pos += 64;
    wrap(g);
  }
};

__device__ __forceinline__ bool wave_any_valid(const GeomView& g, const State& st, int lane) {
  bool any = false;
  ActionCursor cur(g, lane);
  for (int a = lane; a < g.action_dim; a += 64) {
    any |= action_valid_sp(g, st, cur.slot, cur.pos);
    cur.advance(g);
  }
  return __ballot(any) != 0ULL;
}

// Writes the network input row of `st` (grid channel 0 + other features).
__device__ void wave_features(const GeomView& g, const State& st, float* grid, float* other, int other_dim, int lane) {
  for (int c = lane; c < g.cells; c += 64) grid[c] = grid_value(g, st, c);
  int sum_h = 0, max_h = 0, holes = 0, bump = 0;
  for (int c0 = 0; c0 < g.cols; c0 += 64) {
    int c = c0 + lane;
    int h = 0, ho = 0;
    if (c < g.cols) column_stats(g, st, c, h, ho);
    int hn = __shfl_down(h, 1, 64);
    int hb = 0;
    if (c + 1 < g.cols && lane < 63) hb = h > hn ? h - hn : hn - h;
    // the pair straddling a 64-column chunk boundary (only grids wider than 64 columns)
    if (lane == 63 && c + 1 < g.cols) {
      int h2, ho2;
      column_stats(g, st, c + 1, h2, ho2);
      hb = h > h2 ? h - h2 : h2 - h;
    }
    sum_h += atrihip::wave_sum_i(h);
    holes += atrihip::wave_sum_i(ho);
    bump += atrihip::wave_sum_i(hb);
    int mh = h;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mh = max(mh, __shfl_xor(mh, off, 64));
    max_h = max(max_h, mh);
  }
  const int nshape = g.slots * kShapeFeat;
  for (int k = lane; k < nshape; k += 64) {
    int slot = k / kShapeFeat, f = k - slot * kShapeFeat;
    int t = slot_shape(st, slot);
    if (k < other_dim) other[k] = t >= 0 ? g.shape_feat[t * kShapeFeat + f] : 0.0f;
  }
  if (lane < g.slots && nshape + lane < other_dim) other[nshape + lane] = slot_shape(st, lane) >= 0 ? 1.0f : 0.0f;
  if (lane == 0) {
    float ex[kExplicitFeat];
    explicit_features(g, st, sum_h, max_h, holes, bump, ex);
    int k0 = nshape + g.slots;
    for (int f = 0; f < kExplicitFeat; ++f)
      if (k0 + f < other_dim) other[k0 + f] = ex[f];
  }
  for (int k = g.native_other_dim + lane; k < other_dim; k += 64) other[k] = 0.0f;
}

// Masked softmax of the network logits over the valid actions (wave order) -> fresh edges.
// For action spaces up to 6 x 64 the per-lane validity bits and exponentials are computed once and kept
// in registers across the max / sum / write passes (the arithmetic and its order are unchanged).
// Finite guard (the reference checks every evaluation: alphatriangle/nn/network.py:177-191,260-274): a NaN or an
// infinity among the logits of the valid actions makes the softmax denominator non-finite.  Such a node gets a uniform
// prior over its valid actions - a NaN prior would fail every PUCT comparison and freeze the tree - and the
// function returns true so that the caller counts the evaluation; the host raises after the move.  For finite logits
// neither the arithmetic nor its order changes.
constexpr int kLaneActs = 6;
__device__ bool wave_init_edges(const GeomView& g, const State& st, Edge* ed, const float* logits, int lane) {
  if (g.action_dim <= kLaneActs * 64) {
    float lg[kLaneActs];
    unsigned vbits = 0;
    ActionCursor cur(g, lane);
#pragma unroll
    for (int k = 0; k < kLaneActs; ++k) {
      const int a = lane + 64 * k;
      lg[k] = 0.0f;
      if (a < g.action_dim && action_valid_sp(g, st, cur.slot, cur.pos)) {
        vbits |= 1u << k;
        lg[k] = logits[a];
      }
      cur.advance(g);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < kLaneActs; ++k)
      if (vbits & (1u << k)) mx = fmaxf(mx, lg[k]);
    mx = atrihip::wave_max(mx);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < kLaneActs; ++k)
      if (vbits & (1u << k)) {
        lg[k] = exp_det(lg[k] - mx);
        s = s + lg[k];
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
    const bool bad = !(s > 0.0f && s < INFINITY);  // wave-uniform
    float uni = 0.0f;
    if (bad) uni = 1.0f / (float)max(1, atrihip::wave_sum_i(__popc(vbits)));
#pragma unroll
    for (int k = 0; k < kLaneActs; ++k) {
      const int a = lane + 64 * k;
      if (a < g.action_dim) {
        Edge e;
        e.prior = (vbits & (1u << k)) ? (bad ? uni : lg[k] / s) : -1.0f;
        e.n = 0;
        e.w = 0.0f;
        e.child = -1;
        ed[a] = e;
      }
    }
    return bad;
  }
  float mx = -INFINITY;
  for (int a = lane; a < g.action_dim; a += 64)
    if (action_valid(g, st, a)) mx = fmaxf(mx, logits[a]);
  mx = atrihip::wave_max(mx);
  float s = 0.0f;
  for (int a = lane; a < g.action_dim; a += 64)
    if (action_valid(g, st, a)) s = s + exp_det(logits[a] - mx);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
  const bool bad = !(s > 0.0f && s < INFINITY);
  float uni = 0.0f;
  if (bad) {
    int nv = 0;
    for (int a = lane; a < g.action_dim; a += 64) nv += action_valid(g, st, a) ? 1 : 0;
    uni = 1.0f / (float)max(1, atrihip::wave_sum_i(nv));
  }
  for (int a = lane; a < g.action_dim; a += 64) {
    Edge e;
    e.prior = action_valid(g, st, a) ? (bad ? uni : exp_det(logits[a] - mx) / s) : -1.0f;
    e.n = 0;
    e.w = 0.0f;
    e.child = -1;
    ed[a] = e;
  }
  return bad;
}

// PUCT argmax over a node's edges: lanes stride over the dense edge array, xor-butterfly reduction with
// lowest-action tie break.  `pre` holds this lane's first kLaneActs edges when the caller loaded them
// ahead of time (together with the node's meta, so the two HBM round trips overlap); the child index of
// the winning edge travels through the reduction so the caller needs no second load.
struct Pick {
  int action, child;
};
template <bool PRE>
__device__ __forceinline__ Pick wave_select_action(const Edge* ed, const Edge (&pre)[kLaneActs], const NodeMeta& m, int A,
                                                   const SearchParams& sp, int lane) {
  const float sqrt_parent = sqrtf((float)(m.n > 1 ? m.n : 1));
  const float fpu = (sp.fpu_parent && m.n > 0) ? m.w / (float)m.n : 0.0f;
  float bs = -INFINITY;
  int ba = 0x7FFFFFFF, bc = -1;
  if (PRE) {
#pragma unroll
    for (int k = 0; k < kLaneActs; ++k) {
      const Edge e = pre[k];
      if (e.prior < 0.0f) continue;
      float q = e.n > 0 ? e.w / (float)e.n : fpu;
      float sc = puct_score(q, e.prior, sqrt_parent, e.n, sp.cpuct);
      if (sc > bs) {
        bs = sc;
        ba = lane + 64 * k;
        bc = e.child;
      }
    }
  }
  for (int a = lane + (PRE ? kLaneActs * 64 : 0); a < A; a += 64) {
    Edge e = ed[a];
    if (e.prior < 0.0f) continue;
    float q = e.n > 0 ? e.w / (float)e.n : fpu;
    float sc = puct_score(q, e.prior, sqrt_parent, e.n, sp.cpuct);
    if (sc > bs) {
      bs = sc;
      ba = a;
      bc = e.child;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    float os = __shfl_xor(bs, off, 64);
    int oa = __shfl_xor(ba, off, 64);
    int oc = __shfl_xor(bc, off, 64);
    if (os > bs || (os == bs && oa < ba)) {
      bs = os;
      ba = oa;
      bc = oc;
    }
  }
  Pick p;
  p.action = ba == 0x7FFFFFFF ? -1 : ba;
  p.child = bc;
  return p;
}

// ------------------------------------------------------------------ batched engine ops (parity tests, resets)
__global__ void reset_states_kernel(GeomView g, State* states, const uint64_t* seeds, int n) {
  const int i = blockIdx.x;
  const int lane = threadIdx.x;
  if (i >= n) return;
  State st;
  reset_state(g, st, seeds[i]);
  bool hv = wave_any_valid(g, st, lane);
  if (!hv) st.over = 1;
  if (lane == 0) states[i] = st;
}

__global__ void step_states_kernel(GeomView g, State* states, const int32_t* actions, float* rewards, int32_t* done,
                                   int n) {
  const int i = blockIdx.x;
  const int lane = threadIdx.x;
  if (i >= n) return;
  State st = states[i];
  const int a = actions[i];
  if (st.over || a < 0 || a >= g.action_dim || !action_valid(g, st, a)) {
    if (lane == 0) {
      rewards[i] = 0.0f;
      done[i] = st.over ? 1 : -1;  // -1 flags an invalid action
    }
    return;
  }
  int placed, cleared;
  apply_action(g, st, a, placed, cleared);
  bool hv = wave_any_valid(g, st, lane);
  float r = finish_step(g, st, placed, cleared, hv);
  if (lane == 0) {
    states[i] = st;
    rewards[i] = r;
    done[i] = st.over;
  }
}

__global__ void valid_mask_kernel(GeomView g, const State* states, uint8_t* mask, int n) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const State st = states[i];
  for (int a = threadIdx.x; a < g.action_dim; a += 64) mask[(size_t)i * g.action_dim + a] = (!st.over && action_valid(g, st, a)) ? 1 : 0;
}

__global__ void features_kernel(GeomView g, const State* states, float* grid, float* other, int other_dim, int n) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const State st = states[i];
  wave_features(g, st, grid + (size_t)i * g.cells, other + (size_t)i * other_dim, other_dim, threadIdx.x);
}

// ------------------------------------------------------------------ tree kernels
// Fresh single-node trees from the given root states (mask[g] != 0 selects the trees to reset).
__global__ void tree_reset_kernel(TreeBufs tb, const State* roots, const uint8_t* mask) {
  const int g = blockIdx.x;
  if (mask != nullptr && !mask[g]) return;
  if (threadIdx.x == 0) {
    State* st = tree_states(tb, g);
    NodeMeta* mt = tree_meta(tb, g);
    st[0] = roots[g];
    NodeMeta m;
    m.n = 0; m.w = 0.f; m.nn_value = 0.f; m.reward = 0.f; m.parent = -1; m.action = -1;
    m.expanded = 0; m.terminal = roots[g].over; m.depth_hint = 0; m.pad = 0;
    mt[0] = m;
    tb.n_nodes[g] = 1;
  }
}

// Roots that still need their first network evaluation write their input row into slot 0.
__global__ void prepare_roots_kernel(GeomView g, TreeBufs tb) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const NodeMeta m = tree_meta(tb, t)[0];
  const int slot = t * tb.L;
  const bool need = !m.expanded && !m.terminal;
  if (need) {
    const State st = tree_states(tb, t)[0];
    wave_features(g, st, tb.nn_grid + (size_t)slot * g.cells, tb.nn_other + (size_t)slot * tb.other_dim, tb.other_dim, lane);
  }
  if (lane == 0) {
    tb.slot_kind[slot] = need ? kSlotExpand : kSlotNone;
    tb.slot_leaf[slot] = 0;
    tb.slot_plen[slot] = 0;
    for (int s = 1; s < tb.L; ++s) tb.slot_kind[slot + s] = kSlotNone;
  }
}

// Gamma(alpha, 1) by Marsaglia-Tsang (with the alpha < 1 boost); per-lane counter RNG.
__device__ float gamma_sample(float alpha, uint64_t& rng) {
  float boost = 1.0f;
  if (alpha < 1.0f) {
    float u = fmaxf(rng_uniform(rng), 1e-12f);
    boost = powf(u, 1.0f / alpha);
    alpha += 1.0f;
  }
  const float d = alpha - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
  for (int it = 0; it < 64; ++it) {
    float u1 = fmaxf(rng_uniform(rng), 1e-12f), u2 = rng_uniform(rng);
    float x = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307f * u2);
    float v = 1.0f + c * x;
    if (v <= 0.0f) continue;
    v = v * v * v;
    float u = fmaxf(rng_uniform(rng), 1e-12f);
    if (logf(u) < 0.5f * x * x + d - d * v + d * logf(v)) return d * v * boost;
  }
  return d * boost;
}

// Expanded roots get their edges from the network output; every live root gets Dirichlet noise.
__global__ void init_roots_kernel(GeomView g, TreeBufs tb, SearchParams sp, uint64_t* noise_rng) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const int slot = t * tb.L;
  NodeMeta* mt = tree_meta(tb, t);
  Edge* ed = tree_edges(tb, t);
  const State st = tree_states(tb, t)[0];
  if (tb.slot_kind[slot] == kSlotExpand) {
    bool bad = wave_init_edges(g, st, ed, tb.nn_logits + (size_t)slot * tb.logit_stride, lane);
    if (lane == 0) {
      NodeMeta m = mt[0];
      m.expanded = 1;
      float v = tb.nn_value[slot];
      if (!(fabsf(v) < INFINITY)) {
        bad = true;
        v = 0.0f;
      }
      if (bad) tb.census[(size_t)t * kCensus + kCensusNonFinite] += 1;
      m.nn_value = v;
      m.n = 1;
      m.w = m.nn_value;
      mt[0] = m;
      tb.slot_kind[slot] = kSlotNone;
    }
  }
  __syncthreads();
  if (sp.dirichlet_epsilon > 0.0f && sp.dirichlet_alpha > 0.0f && !mt[0].terminal && mt[0].expanded) {
    // the per-action sample is a pure function of (tree stream, action): pass 1 sums it, pass 2
    // recomputes it and mixes it into the prior; visit statistics of a reused root stay untouched
    const uint64_t base = noise_rng[t];
    float tot = 0.0f;
    for (int a = lane; a < g.action_dim; a += 64) {
      if (ed[a].prior < 0.0f) continue;
      uint64_t r = mix64(base ^ ((uint64_t)(a + 1) * 0x9E3779B97F4A7C15ULL));
      tot += gamma_sample(sp.dirichlet_alpha, r);
    }
    tot = atrihip::wave_sum(tot);
    if (tot > 0.0f) {
      for (int a = lane; a < g.action_dim; a += 64) {
        const float p = ed[a].prior;
        if (p < 0.0f) continue;
        uint64_t r = mix64(base ^ ((uint64_t)(a + 1) * 0x9E3779B97F4A7C15ULL));
        const float eta = gamma_sample(sp.dirichlet_alpha, r);
        ed[a].prior = (1.0f - sp.dirichlet_epsilon) * p + sp.dirichlet_epsilon * (eta / tot);
      }
    }
    __syncthreads();
    if (lane == 0) noise_rng[t] = mix64(base + 0x632BE59BD9B4E019ULL);
  }
}

// One simulation wave for one tree: (1) consume the network outputs of the previous wave
// (edge initialisation + backup), (2) select up to L new leaves and expand them.
__global__ ATRI_ENG_BOUNDS void tree_wave_kernel(GeomView g, TreeBufs tb, SearchParams sp, int do_consume, int do_select) {
  const int t = blockIdx.x, lane = threadIdx.x;
  State* states = tree_states(tb, t);
  NodeMeta* meta = tree_meta(tb, t);
  Edge* edges = tree_edges(tb, t);
  const int A = tb.A;
  const bool vl = sp.leaf_batch > 1;

  if (do_consume) {
    for (int s = 0; s < tb.L; ++s) {
      const int slot = t * tb.L + s;
      const int kind = tb.slot_kind[slot];
      if (kind == kSlotNone) continue;
      const int leaf = tb.slot_leaf[slot];
      float value = 0.0f;
      if (kind == kSlotExpand) {
        const State st = states[leaf];
        const int row = tb.live_cursor != nullptr ? tb.slot_row[slot] : slot;
        bool bad = wave_init_edges(g, st, edges + (size_t)leaf * A, tb.nn_logits + (size_t)row * tb.logit_stride, lane);
        value = tb.nn_value[row];
        if (!(fabsf(value) < INFINITY)) {
          bad = true;
          value = 0.0f;
        }
        if (bad && lane == 0) tb.census[(size_t)t * kCensus + kCensusNonFinite] += 1;
      } else if (kind == kSlotDepthCap) {
        value = meta[leaf].nn_value;
      }
      __syncthreads();
      // Backup.  Lane i owns path step i: its loads and read-modify-writes touch a node no other lane touches,
      // so the whole path costs one memory round trip instead of one per level.  The discounted return is the
      // same left-to-right float recurrence as the host search, run in registers over shuffled rewards.
      const int plen = tb.slot_plen[slot];
      const int32_t* pn = tb.path_node + (size_t)slot * tb.P;
      const int32_t* pa = tb.path_action + (size_t)slot * tb.P;
      int node_i = 0, act_i = 0;
      float r_i = 0.0f;
      if (lane < plen) {
        node_i = pn[lane];
        act_i = pa[lane];
        const int child_i = (lane + 1 < plen) ? pn[lane + 1] : leaf;
        r_i = meta[child_i].reward;
      }
      float ret = value, my_ret = 0.0f;
      for (int i = plen - 1; i >= 0; --i) {
        ret = __shfl(r_i, i, 64) + sp.discount * ret;
        if (lane == i) my_ret = ret;
      }
      if (lane < plen) {
        Edge* e = edges + (size_t)node_i * A + act_i;
        Edge ev = *e;
        if (vl) {
          ev.w = ev.w - sp.virtual_loss;
        } else {
          ev.n += 1;
        }
        ev.w = ev.w + my_ret;
        *e = ev;
        NodeMeta* mp = meta + node_i;
        mp->n += 1;
        mp->w = mp->w + my_ret;
      }
      if (lane == 0) {
        if (kind == kSlotExpand) {
          meta[leaf].expanded = 1;
          meta[leaf].nn_value = value;
        }
        meta[leaf].n += 1;
        meta[leaf].w = meta[leaf].w + value;
        tb.slot_kind[slot] = kSlotNone;
      }
      __syncthreads();
    }
  }
  if (!do_select) return;

  for (int s = 0; s < tb.L; ++s) {
    const int slot = t * tb.L + s;
    int32_t* pn = tb.path_node + (size_t)slot * tb.P;
    int32_t* pa = tb.path_action + (size_t)slot * tb.P;
    int node = 0, depth = 0, plen = 0, kind = kSlotNone, leaf = -1;
    const bool pre_ok = A <= kLaneActs * 64;
    while (true) {
      // the lane's edges are requested together with the node's meta (one HBM round trip per level)
      Edge pre[kLaneActs];
      if (pre_ok) {
        const Edge* ed = edges + (size_t)node * A;
#pragma unroll
        for (int k = 0; k < kLaneActs; ++k) {
          const int a = lane + 64 * k;
          pre[k].prior = -1.0f;
          pre[k].n = 0;
          pre[k].w = 0.0f;
          pre[k].child = -1;
          if (a < A) pre[k] = ed[a];
        }
      }
      const NodeMeta m = meta[node];
      if (m.terminal) {
        kind = kSlotTerminal;
        leaf = node;
        break;
      }
      if (!m.expanded) {  // in-flight leaf of an earlier slot of this wave (or an unexpanded root)
        if (vl && lane == 0)
          for (int i = 0; i < plen; ++i) {
            Edge* e = edges + (size_t)pn[i] * A + pa[i];
            e->n -= 1;
            e->w = e->w - sp.virtual_loss;
          }
        kind = kSlotNone;
        plen = 0;
        break;
      }
      if (depth >= sp.max_depth || depth >= tb.P) {
        kind = kSlotDepthCap;
        leaf = node;
        break;
      }
      const Pick pick = pre_ok ? wave_select_action<true>(edges + (size_t)node * A, pre, m, A, sp, lane)
                               : wave_select_action<false>(edges + (size_t)node * A, pre, m, A, sp, lane);
      const int a = pick.action;
      if (a < 0) {
        kind = kSlotDepthCap;
        leaf = node;
        break;
      }
      Edge* e = edges + (size_t)node * A + a;
      const int child = pick.child;
      const int nn = tb.n_nodes[t];
      if (child < 0 && nn >= tb.C) {
        kind = kSlotDepthCap;
        leaf = node;
        break;
      }
      if (lane == 0) {
        pn[plen] = node;
        pa[plen] = a;
        if (vl) {
          e->n += 1;
          e->w = e->w + sp.virtual_loss;
        }
      }
      plen++;
      if (child < 0) {
        State st = states[node];
        int placed, cleared;
        apply_action(g, st, a, placed, cleared);
        const bool hv = wave_any_valid(g, st, lane);
        const float reward = finish_step(g, st, placed, cleared, hv);
        if (lane == 0) {
          states[nn] = st;
          NodeMeta nm;
          nm.n = 0; nm.w = 0.f; nm.nn_value = 0.f; nm.reward = reward; nm.parent = node; nm.action = (int16_t)a;
          nm.expanded = 0; nm.terminal = st.over; nm.depth_hint = depth + 1; nm.pad = 0;
          meta[nn] = nm;
          e->child = nn;
          tb.n_nodes[t] = nn + 1;
        }
        leaf = nn;
        kind = st.over ? kSlotTerminal : kSlotExpand;
        if (kind == kSlotExpand && tb.live_cursor == nullptr)  // with compaction the features follow in leaf_features_kernel
          wave_features(g, st, tb.nn_grid + (size_t)slot * g.cells, tb.nn_other + (size_t)slot * tb.other_dim, tb.other_dim, lane);
        __syncthreads();
        break;
      }
      __syncthreads();
      node = child;
      depth++;
    }
    if (lane == 0) {
      tb.slot_kind[slot] = kind;
      tb.slot_leaf[slot] = leaf;
      tb.slot_plen[slot] = plen;
      tb.census[(size_t)t * kCensus + kind] += 1;
    }
    __syncthreads();
  }
}

// Batch compaction, step 1: number the slots that need an evaluation in slot order.  One workgroup; every thread owns a
// contiguous run of slots, the runs' counts go through an LDS scan.
__global__ __launch_bounds__(1024) void leaf_rows_kernel(TreeBufs tb) {
  __shared__ int sums[1024];
  const int n = tb.G * tb.L, tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = min(tid * per, n), hi = min(lo + per, n);
  int c = 0;
  for (int i = lo; i < hi; ++i) c += tb.slot_kind[i] == kSlotExpand;
  sums[tid] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele scan
    const int v = tid >= off ? sums[tid - off] : 0;
    __syncthreads();
    sums[tid] += v;
    __syncthreads();
  }
  int row = sums[tid] - c;
  for (int i = lo; i < hi; ++i)
    if (tb.slot_kind[i] == kSlotExpand) tb.slot_row[i] = row++;
  if (tid == 1023) *tb.live_cursor = sums[1023];
}

// Batch compaction, step 2: feature rows of the numbered slots, one wave per slot.
__global__ void leaf_features_kernel(GeomView g, TreeBufs tb) {
  const int slot = blockIdx.x, lane = threadIdx.x;
  if (tb.slot_kind[slot] != kSlotExpand) return;
  const int t = slot / tb.L;
  const State st = tree_states(tb, t)[tb.slot_leaf[slot]];
  const int row = tb.slot_row[slot];
  wave_features(g, st, tb.nn_grid + (size_t)row * g.cells, tb.nn_other + (size_t)row * tb.other_dim, tb.other_dim, lane);
}

// Root visit counts -> dense int32 [G][A] (for tests and host-side inspection).
__global__ void root_visits_kernel(TreeBufs tb, int32_t* visits) {
  const int t = blockIdx.x;
  const Edge* ed = tree_edges(tb, t);
  const bool ok = tree_meta(tb, t)[0].expanded;
  for (int a = threadIdx.x; a < tb.A; a += 64) visits[(size_t)t * tb.A + a] = (ok && ed[a].prior >= 0.0f) ? ed[a].n : 0;
}

// ------------------------------------------------------------------ move commit
struct GameBufs {
  uint64_t* game_rng;     // [G] action-sampling stream
  int32_t* episode_idx;   // [G]
  int32_t* ep_sims;       // [G] simulations of the running episode
  // pending n-step experiences, ring of R entries per game
  float* pend_grid;       // [G][R][cells]
  float* pend_other;      // [G][R][other_dim]
  float* pend_policy;     // [G][R][A]
  float* pend_acc;        // [G][R]
  int32_t* pend_count;    // [G][R]
  int32_t* pend_head;     // [G]
  int32_t* pend_len;      // [G]
  // replay arrays (HBM resident)
  float* rp_grid;         // [cap][cells]
  float* rp_other;        // [cap][other_dim]
  float* rp_policy;       // [cap][A]
  float* rp_ret;          // [cap]
  unsigned long long* rp_head;  // monotonically increasing write counter
  int32_t* emitted_idx;   // [emit_cap] replay slots written by this commit
  int32_t* emitted_count; // [1]
  // episode statistics ring
  float* ep_score;        // [ep_ring]
  int32_t* ep_len;
  int32_t* ep_sims_out;
  int32_t* ep_cleared;
  unsigned long long* ep_count;  // [1]
  // per-call aggregates: [0]=sum reward, [1]=sum cleared, [2]=moves, [3]=episodes, [4]=sum root visits
  float* move_stats;
  // outputs for re-rooting
  State* next_root;       // [G]
  int32_t* reroot_child;  // [G] node index of the chosen child, -1 = fresh tree
  int32_t* chosen_action; // [G]
  const float* gamma_pow; // [n_step+1]
  // what this move did to each game (one metric sample per game step on the host): reward (NaN = the game did not
  // move in this call), triangles cleared, score after the move
  float* step_reward;     // [G]
  float* step_cleared;    // [G]
  float* step_score;      // [G]
  uint64_t base_seed;
  int32_t R, n_step, cap, emit_cap, ep_ring, temp_switch_step, tree_reuse, reuse_limit;
  float temp_hi, temp_lo;
};

__device__ void emit_experience(const GeomView& g, const TreeBufs& tb, GameBufs& gb, int game, int entry, float ret, int lane) {
  unsigned long long pos = 0;
  if (lane == 0) {
    pos = atomicAdd(gb.rp_head, 1ULL);
    int e = atomicAdd(gb.emitted_count, 1);
    if (e < gb.emit_cap) gb.emitted_idx[e] = (int)(pos % (unsigned long long)gb.cap);
  }
  pos = __shfl(pos, 0, 64);
  const size_t dst = (size_t)(pos % (unsigned long long)gb.cap);
  const size_t src = (size_t)game * gb.R + entry;
  for (int c = lane; c < g.cells; c += 64) gb.rp_grid[dst * g.cells + c] = gb.pend_grid[src * g.cells + c];
  for (int c = lane; c < tb.other_dim; c += 64) gb.rp_other[dst * tb.other_dim + c] = gb.pend_other[src * tb.other_dim + c];
  for (int c = lane; c < tb.A; c += 64) gb.rp_policy[dst * tb.A + c] = gb.pend_policy[src * tb.A + c];
  if (lane == 0) gb.rp_ret[dst] = ret;
}

__global__ ATRI_ENG_BOUNDS void commit_moves_kernel(GeomView g, TreeBufs tb, GameBufs gb) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const State* states = tree_states(tb, t);
  const NodeMeta* meta = tree_meta(tb, t);
  const Edge* ed = tree_edges(tb, t);
  const int A = tb.A;
  State st = states[0];
  const NodeMeta rootm = meta[0];

  if (st.over || !rootm.expanded) {  // defensive: restart a dead game
    if (lane == 0) {
      State ns;
      gb.episode_idx[t] += 1;
      reset_state(g, ns, mix64(gb.base_seed ^ ((uint64_t)t << 32) ^ (uint64_t)gb.episode_idx[t]));
      gb.next_root[t] = ns;
      gb.reroot_child[t] = -1;
      gb.chosen_action[t] = -1;
      gb.pend_len[t] = 0;
      gb.ep_sims[t] = 0;
      gb.step_reward[t] = NAN;
      gb.step_cleared[t] = 0.0f;
      gb.step_score[t] = 0.0f;
    }
    return;
  }

  // ---- 1. experiences whose n-step window closed on this state: bootstrap with V(root)
  {
    int head = gb.pend_head[t], len = gb.pend_len[t];
    while (len > 0 && gb.pend_count[(size_t)t * gb.R + head] >= gb.n_step) {
      float ret = gb.pend_acc[(size_t)t * gb.R + head] + gb.gamma_pow[gb.n_step] * rootm.nn_value;
      emit_experience(g, tb, gb, t, head, ret, lane);
      head = (head + 1) % gb.R;
      len--;
    }
    __syncthreads();
    if (lane == 0) {
      gb.pend_head[t] = head;
      gb.pend_len[t] = len;
    }
    __syncthreads();
  }

  // ---- 2. visit counts -> policy target (T = 1) and sampled action
  const int chunk = (A + 63) / 64;
  const int a0 = lane * chunk, a1 = min(A, a0 + chunk);
  int local = 0;
  for (int a = a0; a < a1; ++a)
    if (ed[a].prior >= 0.0f) local += ed[a].n;
  const int total = atrihip::wave_sum_i(local);
  const float temp = st.step < gb.temp_switch_step ? gb.temp_hi : gb.temp_lo;
  uint64_t rng = gb.game_rng[t];
  const float u = rng_uniform(rng);
  const uint64_t rint = rng_next(rng);
  int action = -1;
  if (total > 0) {
    if (temp == 1.0f) {
      // exact integer sampling proportional to the visit counts
      int incl = local;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
      }
      const int target = (int)(rint % (uint64_t)total);
      int cand = -1;
      if (target < incl && target >= incl - local) {
        int acc = incl - local;
        for (int a = a0; a < a1; ++a) {
          if (ed[a].prior < 0.0f) continue;
          acc += ed[a].n;
          if (target < acc) {
            cand = a;
            break;
          }
        }
      }
      unsigned long long b = __ballot(cand >= 0);
      int src = b ? __ffsll((long long)b) - 1 : 0;
      action = __shfl(cand, src, 64);
    } else {
      const float inv_t = 1.0f / temp;
      float lw = 0.0f;
      for (int a = a0; a < a1; ++a)
        if (ed[a].prior >= 0.0f && ed[a].n > 0) lw += powf((float)ed[a].n, inv_t);
      float incl = lw;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        float v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
      }
      const float totw = __shfl(incl, 63, 64);
      const float target = u * totw;
      int cand = -1;
      if (lw > 0.0f && target < incl && target >= incl - lw) {
        float acc = incl - lw;
        for (int a = a0; a < a1; ++a) {
          if (ed[a].prior < 0.0f || ed[a].n <= 0) continue;
          acc += powf((float)ed[a].n, inv_t);
          cand = a;
          if (target < acc) break;
        }
      }
      unsigned long long b = __ballot(cand >= 0);
      int src = b ? __ffsll((long long)b) - 1 : 0;
      action = __shfl(cand, src, 64);
    }
  }
  if (action < 0) {  // no visits (or float rounding at the upper edge): most visited, else first valid action
    int best = -1, bn = -1;
    for (int a = a0; a < a1; ++a)
      if (ed[a].prior >= 0.0f && ed[a].n > bn) {
        bn = ed[a].n;
        best = a;
      }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      int on = __shfl_xor(bn, off, 64), ob = __shfl_xor(best, off, 64);
      if (on > bn || (on == bn && ob >= 0 && (best < 0 || ob < best))) {
        bn = on;
        best = ob;
      }
    }
    action = best;
  }

  // ---- 3. new pending experience: features of the root + policy target
  {
    const int head = gb.pend_head[t], len = gb.pend_len[t];
    const int entry = (head + len) % gb.R;
    const size_t pe = (size_t)t * gb.R + entry;
    wave_features(g, st, gb.pend_grid + pe * g.cells, gb.pend_other + pe * tb.other_dim, tb.other_dim, lane);
    const float inv_total = total > 0 ? 1.0f / (float)total : 0.0f;
    for (int a = lane; a < A; a += 64) {
      float p = 0.0f;
      if (ed[a].prior >= 0.0f) p = total > 0 ? (float)ed[a].n * inv_total : 0.0f;
      gb.pend_policy[pe * A + a] = p;
    }
    if (lane == 0) {
      gb.pend_acc[pe] = 0.0f;
      gb.pend_count[pe] = 0;
      gb.pend_len[t] = len + 1;
    }
    __syncthreads();
  }

  // ---- 4. play the move
  const int child = ed[action].child;
  int placed, cleared;
  apply_action(g, st, action, placed, cleared);
  const bool hv = wave_any_valid(g, st, lane);
  const float reward = finish_step(g, st, placed, cleared, hv);
  const bool done = st.over != 0;
  {
    const int head = gb.pend_head[t], len = gb.pend_len[t];
    if (lane == 0)
      for (int i = 0; i < len; ++i) {
        const size_t pe = (size_t)t * gb.R + (head + i) % gb.R;
        const int c = gb.pend_count[pe];
        gb.pend_acc[pe] = gb.pend_acc[pe] + gb.gamma_pow[c] * reward;
        gb.pend_count[pe] = c + 1;
      }
    __syncthreads();
    if (done) {  // flush: truncated discounted sums, no bootstrap
      for (int i = 0; i < len; ++i) {
        const int entry = (head + i) % gb.R;
        emit_experience(g, tb, gb, t, entry, gb.pend_acc[(size_t)t * gb.R + entry], lane);
      }
      __syncthreads();
      if (lane == 0) gb.pend_len[t] = 0;
    }
  }

  if (lane == 0) {
    gb.game_rng[t] = rng;
    gb.chosen_action[t] = action;
    gb.step_reward[t] = reward;
    gb.step_cleared[t] = (float)cleared;
    gb.step_score[t] = st.score;
    const int sims = gb.ep_sims[t] + total;
    atomicAdd(&gb.move_stats[0], reward);
    atomicAdd(&gb.move_stats[1], (float)cleared);
    atomicAdd(&gb.move_stats[2], 1.0f);
    atomicAdd(&gb.move_stats[4], (float)total);
    if (done) {
      unsigned long long ei = atomicAdd(gb.ep_count, 1ULL) % (unsigned long long)gb.ep_ring;
      gb.ep_score[ei] = st.score;
      gb.ep_len[ei] = st.step;
      gb.ep_sims_out[ei] = sims;
      gb.ep_cleared[ei] = st.tris_cleared_total;
      atomicAdd(&gb.move_stats[3], 1.0f);
      State ns;
      gb.episode_idx[t] += 1;
      reset_state(g, ns, mix64(gb.base_seed ^ ((uint64_t)t << 32) ^ (uint64_t)gb.episode_idx[t]));
      gb.next_root[t] = ns;
      gb.reroot_child[t] = -1;
      gb.ep_sims[t] = 0;
    } else {
      gb.next_root[t] = st;
      gb.reroot_child[t] = (gb.tree_reuse && child >= 0) ? child : -1;
      gb.ep_sims[t] = sims;
    }
  }
}

// Move the tree to its next root: compact the chosen child's subtree into the other arena
// (breadth first) or start a fresh single-node tree.
__global__ void reroot_kernel(GeomView g, TreeBufs tb, GameBufs gb) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const int src_a = tb.cur[t], dst_a = 1 - src_a;
  State* sst = tb.states[src_a] + (size_t)t * tb.C;
  NodeMeta* smt = tb.meta[src_a] + (size_t)t * tb.C;
  Edge* sed = tb.edges[src_a] + (size_t)t * tb.C * tb.A;
  State* dst = tb.states[dst_a] + (size_t)t * tb.C;
  NodeMeta* dmt = tb.meta[dst_a] + (size_t)t * tb.C;
  Edge* ded = tb.edges[dst_a] + (size_t)t * tb.C * tb.A;
  int32_t* order = tb.order + (size_t)t * tb.C;
  const int child = gb.reroot_child[t];
  const int A = tb.A;
  if (child < 0) {
    if (lane == 0) {
      State ns = gb.next_root[t];
      NodeMeta m;
      m.n = 0; m.w = 0.f; m.nn_value = 0.f; m.reward = 0.f; m.parent = -1; m.action = -1;
      m.expanded = 0; m.terminal = 0; m.depth_hint = 0; m.pad = 0;
      dst[0] = ns;
      dmt[0] = m;
      tb.n_nodes[t] = 1;
      tb.cur[t] = dst_a;
    }
    // a fresh game may already be terminal (no placement fits): flag it for the next commit
    const State ns = gb.next_root[t];
    const bool hv = wave_any_valid(g, ns, lane);
    __syncthreads();
    if (lane == 0 && !hv) {
      dst[0].over = 1;
      dmt[0].terminal = 1;
    }
    return;
  }
  if (lane == 0) {
    order[0] = child;
    smt[child].parent = -1;
  }
  __syncthreads();
  int head = 0, tail = 1;
  const int limit = max(1, min(tb.C, gb.reuse_limit));
  while (head < tail) {
    const int o = order[head];
    NodeMeta m = smt[o];
    if (lane == 0) {
      dst[head] = sst[o];
      NodeMeta nm = m;
      if (head == 0) {
        nm.parent = -1;
        nm.action = -1;
      }
      dmt[head] = nm;
    }
    if (m.expanded) {
      for (int a0 = 0; a0 < A; a0 += 64) {
        const int a = a0 + lane;
        Edge e;
        e.prior = -1.f; e.n = 0; e.w = 0.f; e.child = -1;
        if (a < A) e = sed[(size_t)o * A + a];
        const bool has = a < A && e.child >= 0;
        const unsigned long long b = __ballot(has);
        const int rank = __popcll(b & ((1ULL << lane) - 1ULL));
        if (has) {
          const int ni = tail + rank;
          if (ni < limit) {
            order[ni] = e.child;
            smt[e.child].parent = head;
            e.child = ni;
          } else {
            e.child = -1;  // truncated: the edge keeps its statistics, the subtree is dropped
          }
        }
        if (a < A) ded[(size_t)head * A + a] = e;
        tail = min(limit, tail + __popcll(b));
      }
    }
    __syncthreads();
    head++;
  }
  if (lane == 0) {
    tb.n_nodes[t] = tail;
    tb.cur[t] = dst_a;
  }
}

}  // namespace ATRI_ENG_NS
using namespace ATRI_ENG_NS;

// ------------------------------------------------------------------ C ABI
extern "C" {

int ATRI_ENG(reset_states)(const GeomView* g, void* states, const void* seeds, int n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(reset_states_kernel, dim3(n), dim3(64), 0, s, *g, (State*)states, (const uint64_t*)seeds, n);
  return (int)hipGetLastError();
}
int ATRI_ENG(step_states)(const GeomView* g, void* states, const int32_t* actions, float* rewards, int32_t* done, int n,
                     hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(step_states_kernel, dim3(n), dim3(64), 0, s, *g, (State*)states, actions, rewards, done, n);
  return (int)hipGetLastError();
}
int ATRI_ENG(valid_mask)(const GeomView* g, const void* states, uint8_t* mask, int n, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(valid_mask_kernel, dim3(n), dim3(64), 0, s, *g, (const State*)states, mask, n);
  return (int)hipGetLastError();
}
int ATRI_ENG(features)(const GeomView* g, const void* states, float* grid, float* other, int other_dim, int n,
                  hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(features_kernel, dim3(n), dim3(64), 0, s, *g, (const State*)states, grid, other, other_dim, n);
  return (int)hipGetLastError();
}
int ATRI_ENG(tree_reset)(const TreeBufs* tb, const void* roots, const uint8_t* mask, hipStream_t s) {
  hipLaunchKernelGGL(tree_reset_kernel, dim3(tb->G), dim3(64), 0, s, *tb, (const State*)roots, mask);
  return (int)hipGetLastError();
}
int ATRI_ENG(prepare_roots)(const GeomView* g, const TreeBufs* tb, hipStream_t s) {
  hipLaunchKernelGGL(prepare_roots_kernel, dim3(tb->G), dim3(64), 0, s, *g, *tb);
  return (int)hipGetLastError();
}
int ATRI_ENG(init_roots)(const GeomView* g, const TreeBufs* tb, const SearchParams* sp, void* noise_rng, hipStream_t s) {
  hipLaunchKernelGGL(init_roots_kernel, dim3(tb->G), dim3(64), 0, s, *g, *tb, *sp, (uint64_t*)noise_rng);
  return (int)hipGetLastError();
}
int ATRI_ENG(tree_wave)(const GeomView* g, const TreeBufs* tb, const SearchParams* sp, int do_consume, int do_select,
                   hipStream_t s) {
  if (sp->max_depth > tb->P) return -10;
  hipLaunchKernelGGL(tree_wave_kernel, dim3(tb->G), dim3(64), 0, s, *g, *tb, *sp, do_consume, do_select);
  if (do_select && tb->live_cursor != nullptr) {
    hipLaunchKernelGGL(leaf_rows_kernel, dim3(1), dim3(1024), 0, s, *tb);
    hipLaunchKernelGGL(leaf_features_kernel, dim3(tb->G * tb->L), dim3(64), 0, s, *g, *tb);
  }
  return (int)hipGetLastError();
}
int ATRI_ENG(root_visits)(const TreeBufs* tb, int32_t* visits, hipStream_t s) {
  hipLaunchKernelGGL(root_visits_kernel, dim3(tb->G), dim3(64), 0, s, *tb, visits);
  return (int)hipGetLastError();
}
int ATRI_ENG(commit_moves)(const GeomView* g, const TreeBufs* tb, const GameBufs* gb, hipStream_t s) {
  hipLaunchKernelGGL(commit_moves_kernel, dim3(tb->G), dim3(64), 0, s, *g, *tb, *gb);
  return (int)hipGetLastError();
}
int ATRI_ENG(reroot)(const GeomView* g, const TreeBufs* tb, const GameBufs* gb, hipStream_t s) {
  hipLaunchKernelGGL(reroot_kernel, dim3(tb->G), dim3(64), 0, s, *g, *tb, *gb);
  return (int)hipGetLastError();
}
int ATRI_ENG(sizeof_treebufs)() { return (int)sizeof(TreeBufs); }
int ATRI_ENG(sizeof_gamebufs)() { return (int)sizeof(GameBufs); }
int ATRI_ENG(sizeof_geomview)() { return (int)sizeof(GeomView); }
int ATRI_ENG(sizeof_searchparams)() { return (int)sizeof(SearchParams); }

}  // extern "C"
