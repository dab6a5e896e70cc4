// Stand-alone exerciser of the host runtime for sanitizer builds (no Python in the process):
//   * game engine: random playouts with the invariants valid <=> placement fits, score/step bookkeeping;
//   * PUCT search: visit-count invariants with leaf batching and subtree reuse;
//   * sum tree: root equals the sum of the leaves after random updates;
//   * metric ring: several producer threads against one consumer, every event delivered exactly once.
// Built by `python -m alphatriangle_amd.build_native --selftest {address|thread|undefined}` with
// -fsanitize=<kind>; tests/test_host_sanitizers.py runs the binaries.  Exit code 0 = all checks passed.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../host/geometry.h"
#include "../host/mcts.h"
#include "../host/metric_ring.h"
#include "../host/sumtree.h"

using namespace atri;

#define CHECK(cond)                                                              \
  do {                                                                           \
    if (!(cond)) {                                                               \
      std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
      std::exit(1);                                                              \
    }                                                                            \
  } while (0)

static EnvParams default_params() {
  EnvParams p;
  p.playable_range = {{3, 12}, {2, 13}, {1, 14}, {0, 15}, {0, 15}, {1, 14}, {2, 13}, {3, 12}};
  return p;
}

static void test_engine(const Geometry& geom) {
  GeomView g = geom.view();
  uint64_t pick = 12345;
  for (int game = 0; game < 20; ++game) {
    State s;
    reset_state(g, s, 1000 + game);
    s.over = any_valid_action(g, s) ? 0 : 1;
    int steps = 0;
    while (!s.over && steps < 400) {
      std::vector<int> valid;
      for (int a = 0; a < g.action_dim; ++a)
        if (action_valid(g, s, a)) valid.push_back(a);
      CHECK(!valid.empty());
      int a = valid[rng_next(pick) % valid.size()];
      int occ_before = 0;
      for (int w = 0; w < g.nwords; ++w) occ_before += popc64(s.occ[w]);
      float score_before = s.score;
      State t = s;
      float r = step_state(g, t, a);
      int occ_after = 0;
      for (int w = 0; w < g.nwords; ++w) occ_after += popc64(t.occ[w]);
      int placed = g.tmpl_ntris[slot_shape(s, a / g.cells)];
      CHECK(occ_after == occ_before + placed - t.last_cleared);
      CHECK(t.score == score_before + (float)(placed + 2 * t.last_cleared));
      CHECK(t.step == s.step + 1);
      CHECK(r == r);  // not NaN
      for (int w = 0; w < kMaxWords; ++w) CHECK((t.occ[w] & ~g.playable[w]) == 0);
      s = t;
      ++steps;
    }
  }
}

static void test_search(std::shared_ptr<const Geometry> geom) {
  GeomView g = geom->view();
  const int A = g.action_dim;
  EvalFn eval = [&](const std::vector<State>& states, std::vector<float>& logits, std::vector<float>& values) {
    logits.assign(states.size() * (size_t)A, 0.0f);
    values.assign(states.size(), 0.0f);
    for (size_t i = 0; i < states.size(); ++i) {
      uint64_t h = states[i].rng ^ (uint64_t)states[i].step;
      for (int a = 0; a < A; ++a) logits[i * A + a] = (float)((mix64(h + a) >> 40) & 1023) / 512.0f - 1.0f;
      values[i] = (float)((mix64(h) >> 20) & 1023) / 512.0f - 1.0f;
    }
  };
  for (int leaf_batch : {1, 4}) {
    SearchParams sp{};
    sp.max_simulations = 64;
    sp.max_depth = 6;
    sp.leaf_batch = leaf_batch;
    sp.fpu_parent = 1;
    sp.cpuct = 1.5f;
    sp.dirichlet_alpha = 0.3f;
    sp.dirichlet_epsilon = 0.25f;
    sp.discount = 1.0f;
    sp.virtual_loss = 1.0f;
    State root;
    reset_state(g, root, 77);
    Tree tree(geom, 2 * sp.max_simulations + 2, 99);
    tree.reset(root);
    for (int move = 0; move < 6 && !root.over; ++move) {
      std::vector<int32_t> visits = tree.search(sp, eval);
      long total = 0;
      int best = -1;
      for (int a = 0; a < A; ++a) {
        CHECK(visits[a] >= 0);
        if (visits[a] > 0) CHECK(action_valid(g, root, a));
        total += visits[a];
        if (best < 0 || visits[a] > visits[best]) best = a;
      }
      CHECK(total > 0 && total <= (long)sp.max_simulations * (move + 1));
      CHECK(tree.n_nodes <= tree.capacity);
      step_state(g, root, best);
      if (!tree.reroot(best)) tree.reset(root);
      CHECK(std::memcmp(&tree.states[0], &root, sizeof(State)) == 0);
    }
  }
}

static void test_sumtree() {
  SumTreeCore t(37);
  uint64_t r = 5;
  std::vector<double> leaves(37, 0.0);
  for (int i = 0; i < 500; ++i) {
    int64_t slot = (int64_t)(rng_next(r) % 37);
    double p = (double)(rng_next(r) % 1000) / 100.0;
    t.update(slot + t.capacity - 1, p);
    leaves[slot] = p;
  }
  double sum = 0;
  for (double v : leaves) sum += v;
  CHECK(std::fabs(t.tree[0] - sum) < 1e-6);
}

static void test_ring() {
  MetricRing ring(1024);
  constexpr int kProducers = 4, kPer = 20000;
  std::atomic<int> live{kProducers};
  std::vector<std::thread> producers;
  for (int p = 0; p < kProducers; ++p)
    producers.emplace_back([&, p] {
      for (int i = 0; i < kPer; ++i) {
        MetricEvent ev{};
        ev.name_id = p;
        ev.value = (double)i;
        ev.global_step = i;
        while (!ring.push(ev)) std::this_thread::yield();
      }
      live.fetch_sub(1);
    });
  std::vector<long> next(kProducers, 0);
  long got = 0;
  MetricEvent ev;
  while (live.load() > 0 || ring.approx_size() > 0) {
    if (ring.pop(ev)) {
      CHECK(ev.name_id >= 0 && ev.name_id < kProducers);
      CHECK(ev.global_step == next[ev.name_id]);  // per-producer order is preserved
      next[ev.name_id]++;
      ++got;
    } else {
      std::this_thread::yield();
    }
  }
  for (auto& th : producers) th.join();
  while (ring.pop(ev)) {
    next[ev.name_id]++;
    ++got;
  }
  CHECK(got == (long)kProducers * kPer);
}

int main() {
  auto geom = std::make_shared<const Geometry>(default_params());
  test_engine(*geom);
  test_search(geom);
  if (kMaxWords > 4) {
    // the wide build (-DATRI_MAX_WORDS=16): a 20 x 30 board, ten occupancy words
    EnvParams p;
    p.rows = 20;
    p.cols = 30;
    p.playable_range.assign(20, {1, 29});
    auto wide = std::make_shared<const Geometry>(p);
    CHECK(wide->view().cells == 600 && wide->view().nwords == 10);
    test_engine(*wide);
    test_search(wide);
  }
  test_sumtree();
  test_ring();
  std::puts("host selftest: ok");
  return 0;
}
