// Host-side construction of the precomputed geometry tables (shape templates,
// placement masks, line masks) consumed through atri::GeomView by both the
// host engine and the HIP engine.
//
// Replaces the geometry part of the external `trianglengin` C++ core that the
// reference imports (alphatriangle/config/__init__.py:2, tests/conftest.py:33-41
// show the EnvConfig fields ROWS / COLS / PLAYABLE_RANGE_PER_ROW / NUM_SHAPE_SLOTS).
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <set>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../core/atri_core.h"

namespace atri {

struct Tri {
  int dr, dc;
  bool up;
  bool operator<(const Tri& o) const { return std::tie(dr, dc, up) < std::tie(o.dr, o.dc, o.up); }
  bool operator==(const Tri& o) const { return dr == o.dr && dc == o.dc && up == o.up; }
};
using Template = std::vector<Tri>;  // sorted, bbox top-left at (0,0)

struct EnvParams {
  int rows = 8, cols = 15, slots = 3, min_line_length = 3, max_shape_tris = 5;
  std::vector<std::pair<int, int>> playable_range;  // [start, end) per row
  float r_place = 0.01f, r_clear = 0.5f, r_alive = 0.005f, r_over = -10.0f;
  std::vector<double> size_weights = {0.10, 0.20, 0.30, 0.25, 0.15};  // P(shape has n triangles)
};

inline bool cell_is_up(int r, int c) { return ((r + c) & 1) != 0; }

inline Template normalize_template(std::vector<std::pair<int, int>> cells) {
  int min_r = 1 << 30, min_c = 1 << 30;
  for (auto& rc : cells) {
    min_r = std::min(min_r, rc.first);
    min_c = std::min(min_c, rc.second);
  }
  Template t;
  for (auto& rc : cells) t.push_back({rc.first - min_r, rc.second - min_c, cell_is_up(rc.first, rc.second)});
  std::sort(t.begin(), t.end());
  return t;
}

// All fixed polyiamonds (translations only) with 1..max_n triangles, in a
// deterministic order: by size, then lexicographic triangle list.
inline std::vector<Template> generate_templates(int max_n) {
  using Cells = std::vector<std::pair<int, int>>;
  std::vector<Template> out;
  std::set<Template> seen_all;
  std::vector<Cells> frontier = {{{0, 0}}, {{0, 1}}};  // one down, one up triangle
  for (int n = 1; n <= max_n; ++n) {
    std::set<Template> level;
    std::vector<Cells> level_cells;
    for (auto& cells : frontier) {
      Template t = normalize_template(cells);
      if (level.insert(t).second) level_cells.push_back(cells);
    }
    for (auto& t : level) out.push_back(t);
    if (n == max_n) break;
    std::vector<Cells> next;
    std::set<Cells> seen_next;
    for (auto& cells : level_cells) {
      for (auto& rc : cells) {
        int r = rc.first, c = rc.second;
        std::pair<int, int> nb[3] = {{r, c - 1}, {r, c + 1}, {cell_is_up(r, c) ? r + 1 : r - 1, c}};
        for (auto& q : nb) {
          if (std::find(cells.begin(), cells.end(), q) != cells.end()) continue;
          Cells grown = cells;
          grown.push_back(q);
          std::sort(grown.begin(), grown.end());
          if (seen_next.insert(grown).second) next.push_back(grown);
        }
      }
    }
    frontier.swap(next);
  }
  return out;
}

class Geometry {
 public:
  EnvParams p;
  int cells = 0, nwords = 0, action_dim = 0;
  std::vector<Template> templates;
  std::vector<uint8_t> tmpl_ntris;
  std::vector<uint8_t> tmpl_parity;  // required (r+c)&1 of the anchor cell
  std::vector<uint64_t> playable;    // kMaxWords
  std::vector<uint64_t> place_mask;
  std::vector<uint8_t> place_ok;
  std::vector<uint64_t> line_mask;
  std::vector<std::vector<int>> lines;  // cell lists, for inspection
  std::vector<uint32_t> shape_cdf;
  std::vector<float> shape_feat;
  int playable_count = 0;

  explicit Geometry(const EnvParams& params) : p(params) { build(); }

  GeomView view() const {
    GeomView g;
    g.rows = p.rows;
    g.cols = p.cols;
    g.cells = cells;
    g.nwords = nwords;
    g.slots = p.slots;
    g.n_templates = (int)templates.size();
    g.n_lines = (int)lines.size();
    g.action_dim = action_dim;
    g.playable_count = playable_count;
    g.native_other_dim = 8 * p.slots + kExplicitFeat;
    g.r_place = p.r_place;
    g.r_clear = p.r_clear;
    g.r_alive = p.r_alive;
    g.r_over = p.r_over;
    g.playable = playable.data();
    g.place_mask = place_mask.data();
    g.place_ok = place_ok.data();
    g.line_mask = line_mask.data();
    g.shape_cdf = shape_cdf.data();
    g.shape_feat = shape_feat.data();
    g.tmpl_ntris = tmpl_ntris.data();
    return g;
  }

  bool is_playable(int r, int c) const {
    if (r < 0 || r >= p.rows || c < 0 || c >= p.cols) return false;
    return c >= p.playable_range[r].first && c < p.playable_range[r].second;
  }

 private:
  void build() {
    if (p.rows <= 0 || p.cols <= 0) throw std::invalid_argument("ROWS and COLS must be positive");
    cells = p.rows * p.cols;
    if (cells > 64 * kMaxWords) throw std::invalid_argument("grid larger than " + std::to_string(64 * kMaxWords) + " cells is not supported by this build of the host engine");
    if (p.slots < 1 || p.slots > kMaxSlots) throw std::invalid_argument("NUM_SHAPE_SLOTS must be in 1..4");
    if ((int)p.playable_range.size() != p.rows)
      throw std::invalid_argument("PLAYABLE_RANGE_PER_ROW needs one (start, end) pair per row");
    if (p.max_shape_tris < 1 || p.max_shape_tris > kMaxTris)
      throw std::invalid_argument("MAX_SHAPE_TRIANGLES must be in 1..5");
    nwords = (cells + 63) / 64;
    action_dim = p.slots * cells;
    playable.assign(kMaxWords, 0);
    for (int r = 0; r < p.rows; ++r) {
      auto rg = p.playable_range[r];
      if (rg.first < 0 || rg.second > p.cols || rg.first > rg.second)
        throw std::invalid_argument("invalid playable range in row " + std::to_string(r));
      for (int c = rg.first; c < rg.second; ++c) {
        int cell = r * p.cols + c;
        playable[cell >> 6] |= 1ULL << (cell & 63);
        ++playable_count;
      }
    }
    build_templates();
    build_lines();
  }

  void build_templates() {
    templates = generate_templates(p.max_shape_tris);
    int nt = (int)templates.size();
    tmpl_ntris.resize(nt);
    tmpl_parity.resize(nt);
    place_mask.assign((size_t)nt * cells * nwords, 0);
    place_ok.assign((size_t)nt * cells, 0);
    shape_feat.assign((size_t)nt * kShapeFeat, 0.0f);
    for (int t = 0; t < nt; ++t) {
      const Template& tp = templates[t];
      int n = (int)tp.size();
      tmpl_ntris[t] = (uint8_t)n;
      // anchor parity: triangle 0 at (r+dr, c+dc) must be `up` iff (r+dr+c+dc) is odd
      int par0 = (tp[0].dr + tp[0].dc) & 1;
      tmpl_parity[t] = (uint8_t)((tp[0].up ? 1 : 0) ^ par0);
      int ups = 0, max_r = 0, max_c = 0;
      for (auto& tr : tp) {
        ups += tr.up ? 1 : 0;
        max_r = std::max(max_r, tr.dr);
        max_c = std::max(max_c, tr.dc);
      }
      // Feature formulas of alphatriangle/features/extractor.py:62-84 (bbox min is 0 by normalisation).
      float* f = &shape_feat[(size_t)t * kShapeFeat];
      float height = (float)(max_r + 1);
      float width_eff = (float)(max_c + 1) * 0.75f + 0.25f;
      f[0] = clipf((float)n / 5.0f, 0.0f, 1.0f);
      f[1] = (float)ups / (float)n;
      f[2] = (float)(n - ups) / (float)n;
      f[3] = clipf(height / (float)p.rows, 0.0f, 1.0f);
      f[4] = clipf(width_eff / (float)p.cols, 0.0f, 1.0f);
      f[5] = clipf(((float)max_r / 2.0f) / (float)p.rows, 0.0f, 1.0f);
      f[6] = clipf(((float)max_c / 2.0f) / (float)p.cols, 0.0f, 1.0f);
      for (int r = 0; r < p.rows; ++r)
        for (int c = 0; c < p.cols; ++c) {
          int pos = r * p.cols + c;
          if (((r + c) & 1) != tmpl_parity[t]) continue;
          bool ok = true;
          std::array<uint64_t, kMaxWords> m{};
          for (auto& tr : tp) {
            int rr = r + tr.dr, cc = c + tr.dc;
            if (!is_playable(rr, cc)) {
              ok = false;
              break;
            }
            int cell = rr * p.cols + cc;
            m[cell >> 6] |= 1ULL << (cell & 63);
          }
          if (!ok) continue;
          place_ok[(size_t)t * cells + pos] = 1;
          for (int w = 0; w < nwords; ++w) place_mask[((size_t)t * cells + pos) * nwords + w] = m[w];
        }
    }
    // cumulative sampling thresholds: P(size n) = size_weights[n-1], uniform within a size
    std::vector<int> per_size(kMaxTris + 1, 0);
    for (int t = 0; t < nt; ++t) per_size[tmpl_ntris[t]]++;
    double total_w = 0;
    for (int n = 1; n <= p.max_shape_tris; ++n)
      total_w += (n - 1 < (int)p.size_weights.size()) ? p.size_weights[n - 1] : 0.0;
    if (total_w <= 0) throw std::invalid_argument("shape size weights must sum to a positive value");
    shape_cdf.resize(nt);
    double acc = 0;
    for (int t = 0; t < nt; ++t) {
      int n = tmpl_ntris[t];
      double wn = (n - 1 < (int)p.size_weights.size()) ? p.size_weights[n - 1] : 0.0;
      acc += wn / total_w / per_size[n];
      double thr = acc * 4294967296.0;
      shape_cdf[t] = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
    }
    shape_cdf[nt - 1] = 0xFFFFFFFFu;
  }

  // Three line families of the triangular grid.  A cell's successor along a family:
  //   horizontal : (r, c+1)
  //   diagonal A : down -> (r, c+1), up -> (r+1, c)
  //   diagonal B : down -> (r, c-1), up -> (r+1, c)
  std::pair<int, int> succ(int fam, int r, int c) const {
    bool up = cell_is_up(r, c);
    if (fam == 0) return {r, c + 1};
    if (fam == 1) return up ? std::make_pair(r + 1, c) : std::make_pair(r, c + 1);
    return up ? std::make_pair(r + 1, c) : std::make_pair(r, c - 1);
  }
  std::pair<int, int> pred(int fam, int r, int c) const {
    bool up = cell_is_up(r, c);
    if (fam == 0) return {r, c - 1};
    if (fam == 1) return up ? std::make_pair(r, c - 1) : std::make_pair(r - 1, c);
    return up ? std::make_pair(r, c + 1) : std::make_pair(r - 1, c);
  }

  void build_lines() {
    lines.clear();
    for (int fam = 0; fam < 3; ++fam)
      for (int r = 0; r < p.rows; ++r)
        for (int c = 0; c < p.cols; ++c) {
          if (!is_playable(r, c)) continue;
          auto pr = pred(fam, r, c);
          if (is_playable(pr.first, pr.second)) continue;  // not the start of a maximal run
          std::vector<int> run;
          int rr = r, cc = c;
          while (is_playable(rr, cc)) {
            run.push_back(rr * p.cols + cc);
            auto nx = succ(fam, rr, cc);
            rr = nx.first;
            cc = nx.second;
          }
          if ((int)run.size() >= p.min_line_length) lines.push_back(run);
        }
    line_mask.assign(lines.size() * (size_t)nwords, 0);
    for (size_t l = 0; l < lines.size(); ++l)
      for (int cell : lines[l]) line_mask[l * nwords + (cell >> 6)] |= 1ULL << (cell & 63);
  }
};

}  // namespace atri
