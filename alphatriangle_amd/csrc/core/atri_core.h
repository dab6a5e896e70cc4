// Single-source game core shared by the host C++ engine (g++) and the HIP
// device engine (hipcc, gfx950).  Every rule of the triangle puzzle, the state
// RNG, the feature formulas and the PUCT score live here so that the device
// engine is bit-exact with the host engine by construction.
//
// Reference API surface this implements (the reference delegates all of it to
// the external `trianglengin` package, which is not vendored):
//   alphatriangle/rl/self_play/worker.py:190-379   (GameState ctor/step/score)
//   alphatriangle/features/extractor.py:33-118     (feature formulas)
//   alphatriangle/features/grid_features.py:7-42   (heights / holes / bumpiness)
//
// Floating point: contraction is disabled so that host and device evaluate the
// same IEEE-754 single operations in the same order.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define ATRI_HD __host__ __device__ inline
#else
#define ATRI_HD inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

// Occupancy words of a game state.  4 (256 cells, the 64-byte state of the device engine) unless the build says otherwise:
// the wide host module is the same code compiled with 16 words (1024 cells).  A wide build lives in a namespace of its
// own, so that both host modules can be loaded into one process without sharing a single type.
#ifndef ATRI_MAX_WORDS
#define ATRI_MAX_WORDS 4
#endif
#if ATRI_MAX_WORDS != 4
#define atri atri_wide
#endif

namespace atri {

constexpr int kMaxWords = ATRI_MAX_WORDS;   // 64 grid cells per word
constexpr int kMaxSlots = 4;
constexpr int kMaxTris = 5;
constexpr int kShapeFeat = 7;
constexpr int kExplicitFeat = 6;

// Flat, pointer-based view of the precomputed geometry tables.  Built on the
// host (geometry.h) and mirrored into HBM for the device engine.
struct GeomView {
  int32_t rows, cols, cells, nwords, slots, n_templates, n_lines, action_dim;
  int32_t playable_count, native_other_dim;
  float r_place, r_clear, r_alive, r_over;
  const uint64_t* playable;    // [kMaxWords] mask of playable (non-death) cells
  const uint64_t* place_mask;  // [n_templates][cells][nwords]
  const uint8_t* place_ok;     // [n_templates][cells] 1 = in-bounds, playable, parity ok
  const uint64_t* line_mask;   // [n_lines][nwords]
  const uint32_t* shape_cdf;   // [n_templates] cumulative 32-bit thresholds
  const float* shape_feat;     // [n_templates][7]
  const uint8_t* tmpl_ntris;   // [n_templates]
};

// Compact game state: 64 bytes with four occupancy words, one per tree node in HBM.
struct State {
  uint64_t occ[kMaxWords];
  uint64_t rng;
  float score;
  int32_t step;
  int16_t shapes[kMaxSlots];  // template id, -1 = empty slot
  int16_t last_cleared;
  uint8_t over;
  uint8_t pad0;
  int32_t tris_cleared_total;
};
static_assert(sizeof(State) == 32 + 8 * kMaxWords, "State must stay 32 bytes + the occupancy words (64 bytes on the device)");

// ---------------------------------------------------------------- RNG
// splitmix64: counter based, identical on host and device.
ATRI_HD uint64_t rng_next(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ULL;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
ATRI_HD uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
// Uniform float in [0,1) with 24 bits.
ATRI_HD float rng_uniform(uint64_t& s) {
  return (float)(rng_next(s) >> 40) * (1.0f / 16777216.0f);
}

ATRI_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(x);
#else
  return __builtin_popcountll(x);
#endif
}

// ---------------------------------------------------------------- shapes
ATRI_HD int sample_shape(const GeomView& g, uint64_t& rng) {
  uint32_t r = (uint32_t)(rng_next(rng) >> 32);
  int n = g.n_templates;
  for (int i = 0; i < n - 1; ++i)
    if (r < g.shape_cdf[i]) return i;
  return n - 1;
}

ATRI_HD void refill_slots(const GeomView& g, State& s) {
  for (int i = 0; i < kMaxSlots; ++i)
    if (i < g.slots) s.shapes[i] = (int16_t)sample_shape(g, s.rng);
}

ATRI_HD bool slots_all_empty(const GeomView& g, const State& s) {
  bool empty = true;
  for (int i = 0; i < kMaxSlots; ++i)
    if (i < g.slots && s.shapes[i] >= 0) empty = false;
  return empty;
}

// State fields are always indexed with compile-time constants (select chains / guarded full-length loops):
// on the device a State then lives in registers instead of scratch memory.
ATRI_HD int slot_shape(const State& s, int slot) {
  // the four int16 slots as one 64-bit word, extracted with a shift (pure ALU: a select chain over the
  // array elements gets folded back into an indexed load by the optimiser)
  uint64_t pk;
  __builtin_memcpy(&pk, s.shapes, sizeof(pk));
  return (int)(int16_t)(uint16_t)(pk >> (16 * slot));
}
ATRI_HD uint64_t occ_word(const State& s, int w) {
  uint64_t v = 0;
  for (int i = 0; i < kMaxWords; ++i) v |= s.occ[i] & (uint64_t)(-(int64_t)(w == i));
  return v;
}

// ---------------------------------------------------------------- rules
// Validity of placing the shape of `slot` at cell `pos` (action = slot * cells + pos).
ATRI_HD bool action_valid_sp(const GeomView& g, const State& s, int slot, int pos) {
  int t = slot_shape(s, slot);
  if (t < 0) return false;
  if (!g.place_ok[t * g.cells + pos]) return false;
  const uint64_t* pm = g.place_mask + ((size_t)t * g.cells + pos) * g.nwords;
  uint64_t hit = 0;
  for (int w = 0; w < kMaxWords; ++w)
    if (w < g.nwords) hit |= (s.occ[w] & pm[w]);
  return hit == 0;
}

ATRI_HD bool action_valid(const GeomView& g, const State& s, int action) {
  int slot = action / g.cells;
  return action_valid_sp(g, s, slot, action - slot * g.cells);
}

ATRI_HD bool any_valid_action(const GeomView& g, const State& s) {
  for (int a = 0; a < g.action_dim; ++a)
    if (action_valid(g, s, a)) return true;
  return false;
}

// Fresh game.  The terminal flag is left to the caller (needs a valid scan).
ATRI_HD void reset_state(const GeomView& g, State& s, uint64_t seed) {
  for (int w = 0; w < kMaxWords; ++w) s.occ[w] = 0;
  s.rng = mix64(seed ^ 0xA5A5A5A55A5A5A5AULL);
  s.score = 0.0f;
  s.step = 0;
  for (int i = 0; i < kMaxSlots; ++i) s.shapes[i] = -1;
  s.last_cleared = 0;
  s.over = 0;
  s.pad0 = 0;
  s.tris_cleared_total = 0;
  refill_slots(g, s);
}

// Place the shape of `action`, clear completed lines, consume the slot and
// refill all slots once every slot is empty.  Caller guarantees validity.
ATRI_HD void apply_action(const GeomView& g, State& s, int action, int& placed, int& cleared) {
  int slot = action / g.cells;
  int pos = action - slot * g.cells;
  int t = slot_shape(s, slot);
  const uint64_t* pm = g.place_mask + ((size_t)t * g.cells + pos) * g.nwords;
  uint64_t pl[kMaxWords];
  for (int w = 0; w < kMaxWords; ++w) {
    pl[w] = w < g.nwords ? pm[w] : 0;
    s.occ[w] |= pl[w];
  }
  placed = g.tmpl_ntris[t];
  uint64_t clr[kMaxWords];
  for (int w = 0; w < kMaxWords; ++w) clr[w] = 0;
  for (int l = 0; l < g.n_lines; ++l) {
    const uint64_t* lm = g.line_mask + (size_t)l * g.nwords;
    uint64_t lw[kMaxWords];
    bool touches = false, full = true;
    for (int w = 0; w < kMaxWords; ++w) {
      lw[w] = w < g.nwords ? lm[w] : 0;
      touches |= (lw[w] & pl[w]) != 0;
      full &= (s.occ[w] & lw[w]) == lw[w];
    }
    if (touches && full)
      for (int w = 0; w < kMaxWords; ++w) clr[w] |= lw[w];
  }
  cleared = 0;
  for (int w = 0; w < kMaxWords; ++w) {
    cleared += popc64(clr[w]);
    s.occ[w] &= ~clr[w];
  }
  for (int i = 0; i < kMaxSlots; ++i)
    if (i == slot) s.shapes[i] = -1;
  if (slots_all_empty(g, s)) refill_slots(g, s);
  s.step += 1;
  s.last_cleared = (int16_t)cleared;
  s.tris_cleared_total += cleared;
  s.score += (float)(placed + 2 * cleared);
}

// Reward of the step just applied; marks the state terminal when no action is left.
ATRI_HD float finish_step(const GeomView& g, State& s, int placed, int cleared, bool has_valid) {
  float reward = (float)placed * g.r_place;
  reward = reward + (float)cleared * g.r_clear;
  if (has_valid) {
    reward = reward + g.r_alive;
  } else {
    s.over = 1;
    reward = reward + g.r_over;
  }
  return reward;
}

ATRI_HD float step_state(const GeomView& g, State& s, int action) {
  int placed, cleared;
  apply_action(g, s, action, placed, cleared);
  bool hv = any_valid_action(g, s);
  return finish_step(g, s, placed, cleared, hv);
}

// ---------------------------------------------------------------- features
ATRI_HD bool cell_bit(const uint64_t* m, int cell) { return (m[cell >> 6] >> (cell & 63)) & 1ULL; }

// Grid channel 0: +1 occupied playable, 0 empty playable, -1 death cell.
ATRI_HD bool occ_bit(const State& s, int cell) { return (occ_word(s, cell >> 6) >> (cell & 63)) & 1ULL; }

ATRI_HD float grid_value(const GeomView& g, const State& s, int cell) {
  if (!cell_bit(g.playable, cell)) return -1.0f;
  return occ_bit(s, cell) ? 1.0f : 0.0f;
}

// Column height = (row of the last occupied playable cell) + 1; holes = empty
// playable cells above that row index (row < height).
ATRI_HD void column_stats(const GeomView& g, const State& s, int c, int& height, int& holes) {
  int max_r = -1;
  for (int r = 0; r < g.rows; ++r) {
    int cell = r * g.cols + c;
    if (occ_bit(s, cell) && cell_bit(g.playable, cell)) max_r = r;
  }
  height = max_r + 1;
  holes = 0;
  for (int r = 0; r < height; ++r) {
    int cell = r * g.cols + c;
    if (!occ_bit(s, cell) && cell_bit(g.playable, cell)) holes += 1;
  }
}

ATRI_HD float clipf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// The six explicit scalar features from integer board statistics.
ATRI_HD void explicit_features(const GeomView& g, const State& s, int sum_h, int max_h, int holes,
                               int bump, float* out) {
  float rows = (float)g.rows;
  out[0] = clipf(s.score / 100.0f, -5.0f, 5.0f);
  out[1] = g.rows > 0 ? ((float)sum_h / (float)g.cols) / rows : 0.0f;
  out[2] = g.rows > 0 ? (float)max_h / rows : 0.0f;
  out[3] = g.playable_count > 0 ? (float)holes / (float)g.playable_count : 0.0f;
  out[4] = (g.cols > 1 && g.rows > 0) ? ((float)bump / (float)(g.cols - 1)) / rows : 0.0f;
  out[5] = clipf((float)s.step / 1000.0f, 0.0f, 1.0f);
}

// Serial extraction of the "other" feature vector, padded / truncated to `dim`.
// Layout: [7 per slot][1 availability per slot][6 explicit].
ATRI_HD void other_features_serial(const GeomView& g, const State& s, float* out, int dim) {
  int k = 0;
  for (int i = 0; i < g.slots; ++i) {
    int t = s.shapes[i];
    for (int f = 0; f < kShapeFeat; ++f) {
      float v = t >= 0 ? g.shape_feat[t * kShapeFeat + f] : 0.0f;
      if (k < dim) out[k] = v;
      ++k;
    }
  }
  for (int i = 0; i < g.slots; ++i) {
    if (k < dim) out[k] = s.shapes[i] >= 0 ? 1.0f : 0.0f;
    ++k;
  }
  int sum_h = 0, max_h = 0, holes = 0, bump = 0, prev = 0;
  for (int c = 0; c < g.cols; ++c) {
    int h, ho;
    column_stats(g, s, c, h, ho);
    sum_h += h;
    if (h > max_h) max_h = h;
    holes += ho;
    if (c > 0) bump += (h > prev ? h - prev : prev - h);
    prev = h;
  }
  float ex[kExplicitFeat];
  explicit_features(g, s, sum_h, max_h, holes, bump, ex);
  for (int f = 0; f < kExplicitFeat; ++f) {
    if (k < dim) out[k] = ex[f];
    ++k;
  }
  for (; k < dim; ++k) out[k] = 0.0f;
}

// ---------------------------------------------------------------- search math
// PUCT score of one edge.  `sqrt_parent` = sqrt(parent visit count).
ATRI_HD float puct_score(float q, float prior, float sqrt_parent, int n, float cpuct) {
  float u = cpuct * prior;
  u = u * sqrt_parent;
  u = u / (float)(1 + n);
  return q + u;
}

// Search parameters shared by the host and the device tree search.
struct SearchParams {
  int32_t max_simulations;
  int32_t max_depth;
  int32_t leaf_batch;       // leaves per tree per wave (virtual loss when > 1)
  int32_t fpu_parent;       // 1: unvisited edges take the parent's mean value, 0: zero
  float cpuct;
  float dirichlet_alpha;
  float dirichlet_epsilon;
  float discount;
  float virtual_loss;       // value charged to an in-flight edge
};

// One tree edge, 16 bytes so a lane moves it with a single dwordx4 access.
struct Edge {
  float prior;    // < 0 marks an invalid action
  int32_t n;      // visit count
  float w;        // sum of backed-up returns
  int32_t child;  // node index, -1 = not expanded
};
static_assert(sizeof(Edge) == 16, "Edge must stay 16 bytes");

// Per-node bookkeeping next to the 64-byte State.
struct NodeMeta {
  int32_t n;        // visits through this node (including its own expansion)
  float w;          // sum of returns seen from this node
  float nn_value;   // network value at expansion
  float reward;     // reward on the edge into this node
  int32_t parent;   // parent node index, -1 for the root
  int16_t action;   // action taken from the parent
  uint8_t expanded; // edges initialised from the network priors
  uint8_t terminal;
  int32_t depth_hint;
  int32_t pad;
};
static_assert(sizeof(NodeMeta) == 32, "NodeMeta must stay 32 bytes");

}  // namespace atri
