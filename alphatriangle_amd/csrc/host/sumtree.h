// Array-backed binary sum tree for prioritized replay (host twin of the HIP
// segment-tree kernels in hip/replay_kernels.hip).
//
// Behavioural parity: alphatriangle/utils/sumtree.py:12-98 (layout 2*cap-1,
// leaves at cap-1.., `value <= left` goes left, ring data pointer, max priority).
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace atri {

class SumTreeCore {
 public:
  int64_t capacity;
  std::vector<double> tree;
  int64_t data_pointer = 0;
  int64_t n_entries = 0;
  double max_priority_seen = 1.0;

  explicit SumTreeCore(int64_t cap) : capacity(cap) {
    if (cap <= 0) throw std::invalid_argument("SumTree capacity must be positive");
    tree.assign((size_t)(2 * cap - 1), 0.0);
  }

  void update(int64_t tree_idx, double priority) {
    if (tree_idx < 0 || tree_idx >= (int64_t)tree.size()) throw std::out_of_range("tree index out of range");
    double change = priority - tree[tree_idx];
    tree[tree_idx] = priority;
    while (tree_idx != 0) {
      tree_idx = (tree_idx - 1) / 2;
      tree[tree_idx] += change;
    }
  }

  // Returns the data slot that was written.
  int64_t add(double priority) {
    int64_t slot = data_pointer;
    update(slot + capacity - 1, priority);
    data_pointer = (data_pointer + 1) % capacity;
    if (n_entries < capacity) ++n_entries;
    max_priority_seen = std::max(max_priority_seen, priority);
    return slot;
  }

  int64_t get_leaf(double value) const {
    int64_t parent = 0;
    int64_t n = (int64_t)tree.size();
    while (true) {
      int64_t left = 2 * parent + 1;
      if (left >= n) return parent;
      if (value <= tree[left]) {
        parent = left;
      } else {
        value -= tree[left];
        parent = left + 1;
      }
    }
  }

  double total() const { return tree[0]; }
  double max_priority() const { return n_entries > 0 ? max_priority_seen : 1.0; }
};

}  // namespace atri
