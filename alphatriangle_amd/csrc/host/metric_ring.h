// Bounded lock-free multi-producer ring for metric events (Vyukov MPMC queue).
// Producers (self-play threads, the learner loop) never block: a full ring
// drops the event and counts it.  One stats thread drains it.
//
// Replaces the fire-and-forget `trieye_actor.log_event.remote(RawMetricEvent)`
// transport of alphatriangle/rl/self_play/worker.py:141-164 and
// alphatriangle/training/loop.py:97-121.
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <vector>

namespace atri {

constexpr int kMaxCtx = 6;

struct MetricEvent {
  int32_t name_id;
  int32_t n_ctx;
  double value;
  int64_t global_step;
  double timestamp;
  int32_t ctx_key[kMaxCtx];
  double ctx_val[kMaxCtx];
};

class MetricRing {
 public:
  explicit MetricRing(size_t capacity_pow2) {
    size_t cap = 2;
    while (cap < capacity_pow2) cap <<= 1;
    mask_ = cap - 1;
    cells_ = std::unique_ptr<Cell[]>(new Cell[cap]);
    for (size_t i = 0; i < cap; ++i) cells_[i].seq.store(i, std::memory_order_relaxed);
    enq_.store(0, std::memory_order_relaxed);
    deq_.store(0, std::memory_order_relaxed);
    dropped_.store(0, std::memory_order_relaxed);
  }

  bool push(const MetricEvent& ev) {
    size_t pos = enq_.load(std::memory_order_relaxed);
    Cell* c;
    while (true) {
      c = &cells_[pos & mask_];
      size_t seq = c->seq.load(std::memory_order_acquire);
      intptr_t dif = (intptr_t)seq - (intptr_t)pos;
      if (dif == 0) {
        if (enq_.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) break;
      } else if (dif < 0) {
        dropped_.fetch_add(1, std::memory_order_relaxed);
        return false;
      } else {
        pos = enq_.load(std::memory_order_relaxed);
      }
    }
    c->ev = ev;
    c->seq.store(pos + 1, std::memory_order_release);
    return true;
  }

  bool pop(MetricEvent& out) {
    size_t pos = deq_.load(std::memory_order_relaxed);
    Cell* c;
    while (true) {
      c = &cells_[pos & mask_];
      size_t seq = c->seq.load(std::memory_order_acquire);
      intptr_t dif = (intptr_t)seq - (intptr_t)(pos + 1);
      if (dif == 0) {
        if (deq_.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) break;
      } else if (dif < 0) {
        return false;
      } else {
        pos = deq_.load(std::memory_order_relaxed);
      }
    }
    out = c->ev;
    c->seq.store(pos + mask_ + 1, std::memory_order_release);
    return true;
  }

  size_t capacity() const { return mask_ + 1; }
  uint64_t dropped() const { return dropped_.load(std::memory_order_relaxed); }
  size_t approx_size() const {
    size_t e = enq_.load(std::memory_order_relaxed), d = deq_.load(std::memory_order_relaxed);
    return e >= d ? e - d : 0;
  }

 private:
  struct Cell {
    std::atomic<size_t> seq;
    MetricEvent ev;
  };
  std::unique_ptr<Cell[]> cells_;
  size_t mask_;
  alignas(64) std::atomic<size_t> enq_;
  alignas(64) std::atomic<size_t> deq_;
  alignas(64) std::atomic<uint64_t> dropped_;
};

}  // namespace atri
