// Shared device helpers for the gfx950 kernel library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace atrihip {

typedef __attribute__((ext_vector_type(8))) short short8;   // 8 x bf16 = one MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) float float4v;  // 16x16 MFMA accumulator fragment

constexpr int kWaveSize = 64;

__device__ __forceinline__ float bf16_to_float(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}

// Round-to-nearest-even f32 -> bf16.  NaN stays NaN (quiet bit forced).
__device__ __forceinline__ unsigned short float_to_bf16(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

enum Act : int { kActNone = 0, kActReLU = 1, kActGELU = 2, kActSiLU = 3, kActTanh = 4, kActSigmoid = 5 };

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case kActReLU: return fmaxf(x, 0.0f);
    case kActGELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    case kActSiLU: return x / (1.0f + __expf(-x));
    case kActTanh: return tanhf(x);
    case kActSigmoid: return 1.0f / (1.0f + __expf(-x));
    default: return x;
  }
}

inline int hip_err(hipError_t e) { return (int)e; }

}  // namespace atrihip
