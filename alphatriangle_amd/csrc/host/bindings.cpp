// pybind11 bindings of the host runtime: geometry tables, game state, feature
// extraction, host tree search, sum tree and the metric ring.
//
// API parity targets (the reference consumes these from external packages):
//   GameState surface  - alphatriangle/rl/self_play/worker.py:190-379,
//                        alphatriangle/features/extractor.py:22-31,58-66
//   run_mcts surface   - alphatriangle/rl/self_play/worker.py:273-280
//   SumTree            - alphatriangle/utils/sumtree.py:12-98
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <array>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "geometry.h"
#include "mcts.h"
#include "metric_ring.h"
#include "sumtree.h"

namespace py = pybind11;
using namespace atri;

// Module name: `_atri_host` (256-cell boards, the state layout of the device engine) or, compiled with
// -DATRI_MAX_WORDS=16 -DATRI_HOST_MODULE=_atri_host_wide, the wide module for boards of up to 1024 cells.
#ifndef ATRI_HOST_MODULE
#define ATRI_HOST_MODULE _atri_host
#endif

// The binding helpers live in the core's namespace (not an anonymous one): the wide build renames that namespace, which
// keeps every bound C++ type of the two modules distinct in pybind11's process-wide registry.
namespace atri {

struct PyGeometry {
  std::shared_ptr<Geometry> g;
};

struct PyState {
  std::shared_ptr<Geometry> g;
  State st;
  std::array<int8_t, 64 * kMaxWords> color;

  PyState(std::shared_ptr<Geometry> geom, uint64_t seed) : g(std::move(geom)) {
    GeomView v = g->view();
    reset_state(v, st, seed);
    color.fill(-1);
    if (!any_valid_action(v, st)) st.over = 1;
  }

  std::pair<float, bool> step(int action) {
    GeomView v = g->view();
    if (st.over) throw std::runtime_error("step() called on a finished game");
    if (action < 0 || action >= v.action_dim || !action_valid(v, st, action))
      throw std::invalid_argument("invalid action " + std::to_string(action));
    int slot = action / v.cells, pos = action % v.cells;
    int t = st.shapes[slot];
    const uint64_t* pm = v.place_mask + ((size_t)t * v.cells + pos) * v.nwords;
    for (int c = 0; c < v.cells; ++c)
      if (cell_bit(pm, c)) color[c] = (int8_t)(t % 8);
    float r = step_state(v, st, action);
    for (int c = 0; c < v.cells; ++c)
      if (!cell_bit(st.occ, c)) color[c] = -1;
    return {r, st.over != 0};
  }

  std::vector<int> valid_actions() const {
    GeomView v = g->view();
    std::vector<int> out;
    if (st.over) return out;
    for (int a = 0; a < v.action_dim; ++a)
      if (action_valid(v, st, a)) out.push_back(a);
    return out;
  }
};

py::array_t<float> make_f32(std::vector<ssize_t> shape) { return py::array_t<float>(shape); }

void fill_features(const GeomView& v, const State& st, int channels, int other_dim, float* grid, float* other) {
  size_t hw = (size_t)v.cells;
  for (size_t i = 0; i < (size_t)channels * hw; ++i) grid[i] = 0.0f;
  for (int c = 0; c < v.cells; ++c) grid[c] = grid_value(v, st, c);
  other_features_serial(v, st, other, other_dim);
}

SearchParams make_params(int max_simulations, int max_depth, float cpuct, float alpha, float eps, float discount,
                         int leaf_batch, bool fpu_parent, float virtual_loss) {
  SearchParams sp{};
  sp.max_simulations = max_simulations;
  sp.max_depth = max_depth > kMaxPath ? kMaxPath : max_depth;
  sp.leaf_batch = leaf_batch;
  sp.fpu_parent = fpu_parent ? 1 : 0;
  sp.cpuct = cpuct;
  sp.dirichlet_alpha = alpha;
  sp.dirichlet_epsilon = eps;
  sp.discount = discount;
  sp.virtual_loss = virtual_loss;
  return sp;
}

struct PyTree {
  std::shared_ptr<Geometry> g;
  std::unique_ptr<Tree> tree;
  bool has_root = false;

  PyTree(const PyGeometry& geom, int capacity, uint64_t seed) : g(geom.g) {
    tree.reset(new Tree(g, capacity, seed));
  }

  // One search from `root`.  When `reuse_action` >= 0 and the stored tree holds a
  // child for it whose state equals `root`, that subtree is kept.
  py::array_t<int32_t> search(const PyState& root, const SearchParams& sp, py::function eval, int reuse_action) {
    bool reused = false;
    if (has_root && reuse_action >= 0) {
      if (tree->reroot(reuse_action) && std::memcmp(&tree->states[0], &root.st, sizeof(State)) == 0) reused = true;
    }
    if (!reused) tree->reset(root.st);
    has_root = true;
    int A = tree->A;
    std::shared_ptr<Geometry> geom = g;
    EvalFn fn = [&](const std::vector<State>& states, std::vector<float>& logits, std::vector<float>& values) {
      py::gil_scoped_acquire gil;
      py::list lst;
      for (auto& s : states) {
        PyState ps(geom, 0);
        ps.st = s;
        lst.append(py::cast(ps));
      }
      py::tuple res = eval(lst).cast<py::tuple>();
      auto lg = res[0].cast<py::array_t<float, py::array::c_style | py::array::forcecast>>();
      auto vl = res[1].cast<py::array_t<float, py::array::c_style | py::array::forcecast>>();
      if (lg.size() != (ssize_t)(states.size() * (size_t)A) || vl.size() != (ssize_t)states.size())
        throw std::runtime_error("evaluator returned arrays of the wrong size");
      logits.assign(lg.data(), lg.data() + lg.size());
      values.assign(vl.data(), vl.data() + vl.size());
      for (float x : values)
        if (!std::isfinite(x)) throw std::runtime_error("evaluator returned a non-finite value");
    };
    std::vector<int32_t> visits;
    {
      py::gil_scoped_release rel;
      visits = tree->search(sp, fn);
    }
    py::array_t<int32_t> out((ssize_t)visits.size());
    std::memcpy(out.mutable_data(), visits.data(), visits.size() * sizeof(int32_t));
    return out;
  }
};

}  // namespace atri

PYBIND11_MODULE(ATRI_HOST_MODULE, m) {
  m.doc() = "Host runtime of alphatriangle_amd (game engine, features, tree search, sum tree, metric ring)";
  m.attr("MAX_WORDS") = kMaxWords;
  m.attr("MAX_SLOTS") = kMaxSlots;
  m.attr("STATE_BYTES") = (int)sizeof(State);
  m.attr("MAX_PATH") = kMaxPath;
  m.attr("MAX_CTX") = kMaxCtx;

  py::class_<PyGeometry>(m, "Geometry")
      .def(py::init([](int rows, int cols, std::vector<std::pair<int, int>> ranges, int slots, int min_line,
                       int max_tris, float r_place, float r_clear, float r_alive, float r_over,
                       std::vector<double> size_weights) {
             EnvParams p;
             p.rows = rows;
             p.cols = cols;
             p.playable_range = std::move(ranges);
             p.slots = slots;
             p.min_line_length = min_line;
             p.max_shape_tris = max_tris;
             p.r_place = r_place;
             p.r_clear = r_clear;
             p.r_alive = r_alive;
             p.r_over = r_over;
             if (!size_weights.empty()) p.size_weights = std::move(size_weights);
             PyGeometry pg;
             pg.g = std::make_shared<Geometry>(p);
             return pg;
           }),
           py::arg("rows"), py::arg("cols"), py::arg("playable_range"), py::arg("slots"), py::arg("min_line_length"),
           py::arg("max_shape_tris"), py::arg("r_place"), py::arg("r_clear"), py::arg("r_alive"), py::arg("r_over"),
           py::arg("size_weights") = std::vector<double>())
      .def_property_readonly("rows", [](const PyGeometry& s) { return s.g->p.rows; })
      .def_property_readonly("cols", [](const PyGeometry& s) { return s.g->p.cols; })
      .def_property_readonly("cells", [](const PyGeometry& s) { return s.g->cells; })
      .def_property_readonly("nwords", [](const PyGeometry& s) { return s.g->nwords; })
      .def_property_readonly("slots", [](const PyGeometry& s) { return s.g->p.slots; })
      .def_property_readonly("n_templates", [](const PyGeometry& s) { return (int)s.g->templates.size(); })
      .def_property_readonly("n_lines", [](const PyGeometry& s) { return (int)s.g->lines.size(); })
      .def_property_readonly("action_dim", [](const PyGeometry& s) { return s.g->action_dim; })
      .def_property_readonly("playable_count", [](const PyGeometry& s) { return s.g->playable_count; })
      .def("rewards",
           [](const PyGeometry& s) {
             return py::make_tuple(s.g->p.r_place, s.g->p.r_clear, s.g->p.r_alive, s.g->p.r_over);
           })
      .def("templates",
           [](const PyGeometry& s) {
             py::list out;
             for (auto& t : s.g->templates) {
               py::list tl;
               for (auto& tr : t) tl.append(py::make_tuple(tr.dr, tr.dc, tr.up));
               out.append(tl);
             }
             return out;
           })
      .def("lines", [](const PyGeometry& s) { return s.g->lines; })
      .def("playable_mask",
           [](const PyGeometry& s) {
             py::array_t<uint64_t> a(kMaxWords);
             std::memcpy(a.mutable_data(), s.g->playable.data(), kMaxWords * 8);
             return a;
           })
      .def("place_mask",
           [](const PyGeometry& s) {
             py::array_t<uint64_t> a({(ssize_t)s.g->templates.size(), (ssize_t)s.g->cells, (ssize_t)s.g->nwords});
             std::memcpy(a.mutable_data(), s.g->place_mask.data(), s.g->place_mask.size() * 8);
             return a;
           })
      .def("place_ok",
           [](const PyGeometry& s) {
             py::array_t<uint8_t> a({(ssize_t)s.g->templates.size(), (ssize_t)s.g->cells});
             std::memcpy(a.mutable_data(), s.g->place_ok.data(), s.g->place_ok.size());
             return a;
           })
      .def("line_mask",
           [](const PyGeometry& s) {
             py::array_t<uint64_t> a({(ssize_t)s.g->lines.size(), (ssize_t)s.g->nwords});
             if (!s.g->line_mask.empty())
               std::memcpy(a.mutable_data(), s.g->line_mask.data(), s.g->line_mask.size() * 8);
             return a;
           })
      .def("shape_cdf",
           [](const PyGeometry& s) {
             py::array_t<uint32_t> a((ssize_t)s.g->shape_cdf.size());
             std::memcpy(a.mutable_data(), s.g->shape_cdf.data(), s.g->shape_cdf.size() * 4);
             return a;
           })
      .def("shape_feat",
           [](const PyGeometry& s) {
             py::array_t<float> a({(ssize_t)s.g->templates.size(), (ssize_t)kShapeFeat});
             std::memcpy(a.mutable_data(), s.g->shape_feat.data(), s.g->shape_feat.size() * 4);
             return a;
           })
      .def("tmpl_ntris", [](const PyGeometry& s) {
        py::array_t<uint8_t> a((ssize_t)s.g->tmpl_ntris.size());
        std::memcpy(a.mutable_data(), s.g->tmpl_ntris.data(), s.g->tmpl_ntris.size());
        return a;
      });

  py::class_<PyState>(m, "CoreState")
      .def(py::init([](const PyGeometry& g, uint64_t seed) { return PyState(g.g, seed); }), py::arg("geometry"),
           py::arg("seed"))
      .def("copy", [](const PyState& s) { return PyState(s); })
      .def("step", &PyState::step)
      .def("valid_actions", &PyState::valid_actions)
      .def("valid_mask",
           [](const PyState& s) {
             GeomView v = s.g->view();
             py::array_t<bool> a(v.action_dim);
             bool* d = a.mutable_data();
             for (int i = 0; i < v.action_dim; ++i) d[i] = !s.st.over && action_valid(v, s.st, i);
             return a;
           })
      .def("is_over", [](const PyState& s) { return s.st.over != 0; })
      .def("game_score", [](const PyState& s) { return s.st.score; })
      .def_property_readonly("current_step", [](const PyState& s) { return s.st.step; })
      .def_property_readonly("last_cleared", [](const PyState& s) { return (int)s.st.last_cleared; })
      .def_property_readonly("tris_cleared_total", [](const PyState& s) { return s.st.tris_cleared_total; })
      .def("shape_ids",
           [](const PyState& s) {
             std::vector<int> out;
             for (int i = 0; i < s.g->p.slots; ++i) out.push_back(s.st.shapes[i]);
             return out;
           })
      .def("set_shape_ids",
           [](PyState& s, std::vector<int> ids) {
             if ((int)ids.size() != s.g->p.slots) throw std::invalid_argument("need one id per slot");
             for (int i = 0; i < s.g->p.slots; ++i) {
               if (ids[i] >= (int)s.g->templates.size()) throw std::invalid_argument("template id out of range");
               s.st.shapes[i] = (int16_t)ids[i];
             }
             s.st.over = any_valid_action(s.g->view(), s.st) ? 0 : 1;
           })
      .def("occupied",
           [](const PyState& s) {
             py::array_t<bool> a({(ssize_t)s.g->p.rows, (ssize_t)s.g->p.cols});
             bool* d = a.mutable_data();
             for (int c = 0; c < s.g->cells; ++c) d[c] = cell_bit(s.st.occ, c);
             return a;
           })
      .def("set_occupied",
           [](PyState& s, py::array_t<bool, py::array::c_style | py::array::forcecast> occ) {
             if (occ.size() != s.g->cells) throw std::invalid_argument("occupied array has the wrong size");
             for (int w = 0; w < kMaxWords; ++w) s.st.occ[w] = 0;
             const bool* d = occ.data();
             for (int c = 0; c < s.g->cells; ++c)
               if (d[c] && cell_bit(s.g->playable.data(), c)) {
                 s.st.occ[c >> 6] |= 1ULL << (c & 63);
                 if (s.color[c] < 0) s.color[c] = 0;
               } else {
                 s.color[c] = -1;
               }
             s.st.over = any_valid_action(s.g->view(), s.st) ? 0 : 1;
           })
      .def("death",
           [](const PyState& s) {
             py::array_t<bool> a({(ssize_t)s.g->p.rows, (ssize_t)s.g->p.cols});
             bool* d = a.mutable_data();
             for (int c = 0; c < s.g->cells; ++c) d[c] = !cell_bit(s.g->playable.data(), c);
             return a;
           })
      .def("color_id",
           [](const PyState& s) {
             py::array_t<int8_t> a({(ssize_t)s.g->p.rows, (ssize_t)s.g->p.cols});
             std::memcpy(a.mutable_data(), s.color.data(), (size_t)s.g->cells);
             return a;
           })
      .def("pack",
           [](const PyState& s) { return py::bytes(reinterpret_cast<const char*>(&s.st), sizeof(State)); })
      .def("unpack",
           [](PyState& s, py::bytes b) {
             std::string raw = b;
             if (raw.size() != sizeof(State)) throw std::invalid_argument("packed state must be " + std::to_string(sizeof(State)) + " bytes");
             std::memcpy(&s.st, raw.data(), sizeof(State));
             for (int c = 0; c < s.g->cells; ++c) s.color[c] = cell_bit(s.st.occ, c) ? (s.color[c] < 0 ? 0 : s.color[c]) : -1;
           })
      .def("core_equal", [](const PyState& a, const PyState& b) { return std::memcmp(&a.st, &b.st, sizeof(State)) == 0; });

  m.def(
      "extract_features",
      [](const PyState& s, int channels, int other_dim) {
        GeomView v = s.g->view();
        py::array_t<float> grid({(ssize_t)channels, (ssize_t)v.rows, (ssize_t)v.cols});
        py::array_t<float> other((ssize_t)other_dim);
        fill_features(v, s.st, channels, other_dim, grid.mutable_data(), other.mutable_data());
        return py::make_tuple(grid, other);
      },
      py::arg("state"), py::arg("grid_channels"), py::arg("other_dim"));

  m.def(
      "extract_features_batch",
      [](const std::vector<PyState*>& states, int channels, int other_dim, int rows, int cols) {
        ssize_t n = (ssize_t)states.size();
        py::array_t<float> grid({n, (ssize_t)channels, (ssize_t)rows, (ssize_t)cols});
        py::array_t<float> other({n, (ssize_t)other_dim});
        float* gp = grid.mutable_data();
        float* op = other.mutable_data();
        size_t gs = (size_t)channels * rows * cols;
        for (ssize_t i = 0; i < n; ++i) {
          GeomView v = states[i]->g->view();
          if (v.rows != rows || v.cols != cols) throw std::invalid_argument("state grid size mismatch");
          fill_features(v, states[i]->st, channels, other_dim, gp + i * gs, op + i * other_dim);
        }
        return py::make_tuple(grid, other);
      },
      py::arg("states"), py::arg("grid_channels"), py::arg("other_dim"), py::arg("rows"), py::arg("cols"));

  m.def("exp_det", [](float x) { return exp_det(x); });
  m.def("wave_order_sum", [](py::array_t<float, py::array::c_style | py::array::forcecast> a) {
    const float* d = a.data();
    return wave_order_sum((int)a.size(), [&](int i) { return d[i]; });
  });

  py::class_<SearchParams>(m, "SearchParams")
      .def(py::init(&make_params), py::arg("max_simulations"), py::arg("max_depth"), py::arg("cpuct"),
           py::arg("dirichlet_alpha"), py::arg("dirichlet_epsilon"), py::arg("discount"), py::arg("leaf_batch"),
           py::arg("fpu_parent") = true, py::arg("virtual_loss") = -1.0f)
      .def_readwrite("max_simulations", &SearchParams::max_simulations)
      .def_readwrite("max_depth", &SearchParams::max_depth)
      .def_readwrite("leaf_batch", &SearchParams::leaf_batch)
      .def_readwrite("fpu_parent", &SearchParams::fpu_parent)
      .def_readwrite("cpuct", &SearchParams::cpuct)
      .def_readwrite("dirichlet_alpha", &SearchParams::dirichlet_alpha)
      .def_readwrite("dirichlet_epsilon", &SearchParams::dirichlet_epsilon)
      .def_readwrite("discount", &SearchParams::discount)
      .def_readwrite("virtual_loss", &SearchParams::virtual_loss);

  py::class_<PyTree>(m, "SearchTree")
      .def(py::init<const PyGeometry&, int, uint64_t>(), py::arg("geometry"), py::arg("capacity"), py::arg("seed"))
      .def("search", &PyTree::search, py::arg("root"), py::arg("params"), py::arg("evaluator"),
           py::arg("reuse_action") = -1)
      .def_property_readonly("n_nodes", [](const PyTree& t) { return t.tree->n_nodes; })
      .def_property_readonly("capacity", [](const PyTree& t) { return t.tree->capacity; })
      .def_property_readonly("total_simulations", [](const PyTree& t) { return t.tree->total_simulations; })
      .def_property_readonly("max_depth_seen", [](const PyTree& t) { return t.tree->max_depth_seen; })
      .def("root_value", [](const PyTree& t) { return t.tree->meta[0].nn_value; })
      .def("root_mean_value",
           [](const PyTree& t) { return t.tree->meta[0].n > 0 ? t.tree->meta[0].w / t.tree->meta[0].n : 0.0f; })
      .def("root_priors",
           [](const PyTree& t) {
             py::array_t<float> a(t.tree->A);
             const Edge* e = t.tree->node_edges(0);
             for (int i = 0; i < t.tree->A; ++i) a.mutable_data()[i] = e[i].prior;
             return a;
           })
      .def("node_meta",
           [](const PyTree& t) {
             py::array_t<uint8_t> a({(ssize_t)t.tree->n_nodes, (ssize_t)sizeof(NodeMeta)});
             std::memcpy(a.mutable_data(), t.tree->meta.data(), (size_t)t.tree->n_nodes * sizeof(NodeMeta));
             return a;
           })
      .def("node_states",
           [](const PyTree& t) {
             py::array_t<uint8_t> a({(ssize_t)t.tree->n_nodes, (ssize_t)sizeof(State)});
             std::memcpy(a.mutable_data(), t.tree->states.data(), (size_t)t.tree->n_nodes * sizeof(State));
             return a;
           })
      .def("node_edges", [](const PyTree& t) {
        py::array_t<uint8_t> a({(ssize_t)t.tree->n_nodes, (ssize_t)t.tree->A, (ssize_t)sizeof(Edge)});
        std::memcpy(a.mutable_data(), t.tree->edges.data(), (size_t)t.tree->n_nodes * t.tree->A * sizeof(Edge));
        return a;
      });

  py::class_<SumTreeCore>(m, "SumTreeCore")
      .def(py::init<int64_t>(), py::arg("capacity"))
      .def_readonly("capacity", &SumTreeCore::capacity)
      .def_readwrite("data_pointer", &SumTreeCore::data_pointer)
      .def_readwrite("n_entries", &SumTreeCore::n_entries)
      .def_readwrite("max_priority_seen", &SumTreeCore::max_priority_seen)
      .def("add", &SumTreeCore::add)
      .def("update", &SumTreeCore::update)
      .def("get_leaf", &SumTreeCore::get_leaf)
      .def("total", &SumTreeCore::total)
      .def("max_priority", &SumTreeCore::max_priority)
      .def("tree_view",
           [](SumTreeCore& s) {
             return py::array_t<double>({(ssize_t)s.tree.size()}, {sizeof(double)}, s.tree.data(), py::cast(&s));
           })
      .def("update_batch",
           [](SumTreeCore& s, py::array_t<int64_t, py::array::c_style | py::array::forcecast> idx,
              py::array_t<double, py::array::c_style | py::array::forcecast> pr) {
             if (idx.size() != pr.size()) throw std::invalid_argument("length mismatch");
             for (ssize_t i = 0; i < idx.size(); ++i) s.update(idx.data()[i], pr.data()[i]);
           })
      .def("get_leaf_batch", [](const SumTreeCore& s, py::array_t<double, py::array::c_style | py::array::forcecast> v) {
        py::array_t<int64_t> idx(v.size());
        py::array_t<double> pr(v.size());
        for (ssize_t i = 0; i < v.size(); ++i) {
          int64_t leaf = s.get_leaf(v.data()[i]);
          idx.mutable_data()[i] = leaf;
          pr.mutable_data()[i] = s.tree[leaf];
        }
        return py::make_tuple(idx, pr);
      });

  py::class_<MetricRing>(m, "MetricRing")
      .def(py::init<size_t>(), py::arg("capacity"))
      .def("push",
           [](MetricRing& r, int name_id, double value, int64_t step, double ts, std::vector<int> keys,
              std::vector<double> vals) {
             MetricEvent ev{};
             ev.name_id = name_id;
             ev.value = value;
             ev.global_step = step;
             ev.timestamp = ts;
             ev.n_ctx = (int)std::min<size_t>(std::min(keys.size(), vals.size()), kMaxCtx);
             for (int i = 0; i < ev.n_ctx; ++i) {
               ev.ctx_key[i] = keys[i];
               ev.ctx_val[i] = vals[i];
             }
             return r.push(ev);
           },
           py::arg("name_id"), py::arg("value"), py::arg("global_step"), py::arg("timestamp"),
           py::arg("ctx_keys") = std::vector<int>(), py::arg("ctx_vals") = std::vector<double>())
      .def("drain",
           [](MetricRing& r, size_t max_events) {
             py::list out;
             MetricEvent ev;
             size_t n = 0;
             while (n < max_events && r.pop(ev)) {
               py::list ck, cv;
               for (int i = 0; i < ev.n_ctx; ++i) {
                 ck.append(ev.ctx_key[i]);
                 cv.append(ev.ctx_val[i]);
               }
               out.append(py::make_tuple(ev.name_id, ev.value, ev.global_step, ev.timestamp, ck, cv));
               ++n;
             }
             return out;
           },
           py::arg("max_events") = 1 << 20)
      .def_property_readonly("capacity", &MetricRing::capacity)
      .def_property_readonly("dropped", &MetricRing::dropped)
      .def("__len__", &MetricRing::approx_size);
}
