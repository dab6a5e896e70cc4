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

typedef __attribute__((ext_vector_type(2))) float float2v;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
typedef __attribute__((ext_vector_type(4))) unsigned int uint4v;

// Round-to-nearest-even f32 -> bf16 on the gfx950 converter (v_cvt_pk_bf16_f32).
__device__ __forceinline__ unsigned short float_to_bf16(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
// Two floats -> one dword of two bf16 (lo in bits 0..15): a single v_cvt_pk_bf16_f32.
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  float2v v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2v));
}
// Eight floats -> one MFMA fragment / one 16-byte store.
__device__ __forceinline__ short8 pack_bf16x8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
  uint4v u = {pack_bf16x2(a0, a1), pack_bf16x2(a2, a3), pack_bf16x2(a4, a5), pack_bf16x2(a6, a7)};
  return __builtin_bit_cast(short8, u);
}

// fp32 -> (hi, lo) bf16 pair with hi + lo carrying 16 significant bits of x: hi = bf16(x), lo = bf16(x - hi).  Products of
// such pairs on the bf16 matrix cores (hi*hi + lo*hi + hi*lo, fp32 accumulation) are the fp32-grade GEMMs of the LEARNER
// kernels (learner_gemm_kernels.hip, learner_attn_kernels.hip): gradients span fp32's whole exponent range, which bf16
// keeps and fp16 does not.  A NaN or an infinity stays visible: hi carries it, lo becomes NaN.
__device__ __forceinline__ void split_bf16x8(const float (&v)[8], short8& hi, short8& lo) {
  hi = pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = v[j] - bf16_to_float((unsigned short)hi[j]);
  lo = pack_bf16x8(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
}
// acc += A B for operands given as (hi, lo) bf16 pairs: three MFMA passes, smallest terms first.
__device__ __forceinline__ float4v mfma_split(const short8& ah, const short8& al, const short8& bh, const short8& bl, float4v acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

// Operand format of the fp32-grade INFERENCE kernels (split_*.hip), chosen when the library is built:
//   1  fp16 pairs (default): hi = f16(x), lo = f16(x - hi) - 22 significant bits per operand for |x| >= 2^-3, an absolute
//      error of at most 2^-25 below that (fp16's subnormal spacing), values beyond 65504 become infinities (and surface
//      through the engine's non-finite census).  Activations of this network are O(1 .. 100) and probabilities are
//      taken relative to their row maximum, so the range is not a constraint; measured against an fp64 evaluation the
//      logits are within 0.6 - 1.2e-6, as close as PyTorch's own fp32 path (0.5 - 2.3e-6).
//   0  bf16 pairs: 16 significant bits, fp32's exponent range (logits within 4 - 6e-6); the first form of these kernels.
// Either way a product is three MFMA passes hi*hi + lo*hi + hi*lo with fp32 accumulation, at the same speed (the f16
// and bf16 MFMA of one shape take the same cycles).  The host packs weights in the same format (ops/inference.py
// reads atri_split_operand_format()).
#ifndef ATRI_SPLIT_F16
#define ATRI_SPLIT_F16 1
#endif
constexpr bool kSplitF16 = ATRI_SPLIT_F16 != 0;
#ifndef ATRI_SPLIT_NOMIX
#define ATRI_SPLIT_NOMIX 0
#endif
typedef __attribute__((ext_vector_type(8))) _Float16 half8v;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;

// Two floats -> one dword of two split operands (first value in bits 0..15), round to nearest even.
__device__ __forceinline__ unsigned int split_pack2(float a, float b) {
  if constexpr (kSplitF16) {
    float2v v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, half2v));
  } else {
    return pack_bf16x2(a, b);
  }
}
__device__ __forceinline__ float split_unpack(unsigned short h) {
  if constexpr (kSplitF16) {
    return (float)__builtin_bit_cast(_Float16, h);
  } else {
    return bf16_to_float(h);
  }
}
__device__ __forceinline__ float4v split_mma(const short8& a, const short8& b, float4v acc) {
  if constexpr (kSplitF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, a), __builtin_bit_cast(half8v, b), acc, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
}
// v - value(h): what the lo operand has to carry.  For fp16 pairs this is written as fma(h, -1, v) - the same single
// rounding as the subtraction - which the compiler selects as one v_fma_mix_f32 (the conversion of h rides in the
// instruction) instead of v_cvt_f32_f16 + v_sub_f32: one vector instruction less per split element.
__device__ __forceinline__ float split_residual(float v, unsigned short h) {
  if constexpr (kSplitF16) {
#if ATRI_SPLIT_NOMIX
    return v - split_unpack(h);
#else
    float minus_one = -1.0f;
    asm("" : "+s"(minus_one));  // opaque: a literal -1 is folded back into a subtraction of the converted value
    return __builtin_fmaf((float)__builtin_bit_cast(_Float16, h), minus_one, v);
#endif
  } else {
    return v - split_unpack(h);
  }
}
// fp32 -> (hi, lo) pair of split operands with hi + lo ~= x.
__device__ __forceinline__ void split_op8(const float (&v)[8], short8& hi, short8& lo) {
  const uint4v uh = {split_pack2(v[0], v[1]), split_pack2(v[2], v[3]), split_pack2(v[4], v[5]), split_pack2(v[6], v[7])};
  hi = __builtin_bit_cast(short8, uh);
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = split_residual(v[j], (unsigned short)hi[j]);
  const uint4v ul = {split_pack2(r[0], r[1]), split_pack2(r[2], r[3]), split_pack2(r[4], r[5]), split_pack2(r[6], r[7])};
  lo = __builtin_bit_cast(short8, ul);
}
// acc += A B for operands given as (hi, lo) pairs of the split format: three MFMA passes, smallest terms first.
__device__ __forceinline__ float4v mma_split3(const short8& ah, const short8& al, const short8& bh, const short8& bl, float4v acc) {
  acc = split_mma(al, bh, acc);
  acc = split_mma(ah, bl, acc);
  return split_mma(ah, bh, acc);
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

// Maximum over the four lanes that share (lane & 15): lane, lane ^ 16, lane ^ 32, lane ^ 48.  gfx950's row-swap
// instructions exchange whole 16-lane rows between two registers in one VALU operation
// (v_permlane16_swap: odd rows of the first operand <-> even rows of the second; v_permlane32_swap: upper half of
// the first <-> lower half of the second), so with both operands equal the pair holds each lane's own value and
// its partner's.  __shfl_xor goes through ds_bpermute instead, an LDS round trip per exchange.
__device__ __forceinline__ float quad_row_max(float m) {
  const unsigned u = __float_as_uint(m);
  const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float a = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
  const unsigned v = __float_as_uint(a);
  const auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}
__device__ __forceinline__ float quad_row_sum(float m) {
  const unsigned u = __float_as_uint(m);
  const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float a = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
  const unsigned v = __float_as_uint(a);
  const auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
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

// Live row count for the inference engines' batch kernels.  A search wave writes only the leaves that need an evaluation
// (densely, rows 0 .. n-1) and leaves n in device memory; the engine kernels are launched for the full batch capacity
// and every workgroup beyond row n ends at once.  The pointer is host state of the library: atri_set_live_count()
// sets it, every engine launch made while it is set carries it (also into a captured graph), nullptr = all rows.
const int* live_count_ptr();
__device__ __forceinline__ int live_rows(const int* live, int capacity) {
  if (live == nullptr) return capacity;
  const int n = *live;
  return n < capacity ? n : capacity;
}


// How a shared-FC launch's workgroups are dealt out for nB live rows.  The launch is sized for the full batch
// (full_row_blocks x SK workgroups, steps_per_split K-steps each); with max_splits > SK the same workgroups cover only
// the row blocks that hold leaves and split K further - up to max_splits ways - so that a thin batch still occupies every
// CU and each workgroup walks a shorter K range.  Both the FC kernel and the tail kernel that sums the slabs derive the
// plan from the device-side row count, so they agree without a host round trip.
struct FcPlan {
  int row_blocks, splits, steps;
};
__host__ __device__ inline FcPlan fc_plan(int nB, int rows_per_block, int full_row_blocks, int SK, int steps_per_split,
                                          int total_steps, int max_splits) {
  FcPlan p{full_row_blocks, SK, steps_per_split};
  if (max_splits > SK && nB > 0) {
    p.row_blocks = (nB + rows_per_block - 1) / rows_per_block;
    int splits = full_row_blocks * SK / p.row_blocks;
    if (splits > max_splits) splits = max_splits;
    p.steps = ((total_steps + splits - 1) / splits + 3) & ~3;  // whole pixel rows (4 K-steps) per split
    p.splits = (total_steps + p.steps - 1) / p.steps;          // no empty split at the end
  }
  return p;
}

}  // namespace atrihip
