// Deterministic numeric helpers of the tree search, shared by the host search
// (host/mcts.h) and the HIP tree kernels (hip/tree_kernels.hip).  The wave-order
// reduction helpers let the host reproduce bit-for-bit the sums a 64-lane
// wavefront computes, so a device tree can be compared exactly with a host tree.
//
// Replaces the numeric core of the external `trimcts` C++ search used at
// alphatriangle/rl/self_play/worker.py:273-280.
#pragma once
#include "atri_core.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace atri {

ATRI_HD float bits_to_float(uint32_t b) {
  union {
    uint32_t u;
    float f;
  } c;
  c.u = b;
  return c.f;
}

// expf with only IEEE add/mul/floor so host and device agree bit-for-bit.
// Relative error < 2e-7 on [-87, 88].
ATRI_HD float exp_det(float x) {
  x = clipf(x, -87.0f, 88.0f);
  float fn = floorf(x * 1.44269504088896341f + 0.5f);
  float r = x - fn * 0.693359375f;
  r = r - fn * -2.12194440e-4f;
  float z = r * r;
  float p = 1.9875691500E-4f;
  p = p * r + 1.3981999507E-3f;
  p = p * r + 8.3334519073E-3f;
  p = p * r + 4.1665795894E-2f;
  p = p * r + 1.6666665459E-1f;
  p = p * r + 5.0000001201E-1f;
  p = p * z;
  p = p + r;
  p = p + 1.0f;
  int n = (int)fn;
  return p * bits_to_float((uint32_t)(n + 127) << 23);
}

constexpr int kWave = 64;

#if !defined(__HIP_DEVICE_COMPILE__)
// Host emulation of a wave-order sum: lane l accumulates elements l, l+64, ...
// in increasing order, then an xor-butterfly combines the 64 lane partials.
template <typename F>
inline float wave_order_sum(int n, F value_at) {
  float lane[kWave];
  for (int l = 0; l < kWave; ++l) {
    float s = 0.0f;
    for (int i = l; i < n; i += kWave) s = s + value_at(i);
    lane[l] = s;
  }
  for (int off = kWave / 2; off >= 1; off >>= 1) {
    float next[kWave];
    for (int l = 0; l < kWave; ++l) next[l] = lane[l] + lane[l ^ off];
    for (int l = 0; l < kWave; ++l) lane[l] = next[l];
  }
  return lane[0];
}
#endif

// Slot kinds produced by one selection pass.
enum SlotKind : int32_t {
  kSlotNone = 0,      // nothing to back up (collision with an in-flight leaf)
  kSlotExpand = 1,    // new non-terminal node: needs a network evaluation
  kSlotTerminal = 2,  // leaf is terminal: value 0
  kSlotDepthCap = 3,  // depth / capacity cap on an expanded node: reuse its stored value
};

constexpr int kMaxPath = 64;  // upper bound for SearchParams::max_depth

}  // namespace atri
