// Helpers shared by the fp32-grade Transformer kernels (split_tf_kernels.hip: one workgroup per sample, up to 128
// tokens; split_tf_rows_kernels.hip: 129..256 tokens): (hi, lo) 16-bit planes of a 128-row token tile in LDS, a
// 64-register ring of packed weight fragments, the K = 128 GEMM step over them, LayerNorm of rows on their way from
// memory into the planes.
#pragma once
#include "common.hpp"

namespace atrihip {
namespace splittf {
// Softmax without the row maximum ("free" pass): p = exp2(score + kFreeShift), accepted when every row sum of the wave lies in
// (kFreeLo, kFreeHi); otherwise the wave repeats its tiles with the maximum subtracted.  bf16 pairs keep fp32's exponent
// range, so the window only excludes overflow, total underflow and NaN.  fp16 pairs need every probability below 65504
// (row sum < 2^15) and the row sum large enough that fp16's absolute resolution (2^-25 for the lo part) stays below
// 2^-22 of it (row sum > 2^4); the shift of 3 centres that window on rows of ~100 comparable scores.  The safe pass
// scales its probabilities by 2^kSafeShift for the same reason (largest term 16 instead of 1).
constexpr float kFreeShift = kSplitF16 ? 3.0f : 0.0f;
constexpr float kFreeLo = kSplitF16 ? 0x1p4f : 0x1p-100f;
constexpr float kFreeHi = kSplitF16 ? 0x1p15f : 0x1p100f;
constexpr float kSafeShift = kSplitF16 ? 4.0f : 0.0f;
// A wave whose free pass was rejected this many times in one launch goes straight to the safe pass for its remaining
// samples: weights that produce large scores cost a bounded number of repeated passes.
constexpr int kFreeStrikes = 2;
constexpr int kDs = 128;
constexpr int kStrideS = 288;                  // bytes per token row of a 16-bit plane
constexpr int kPlane = 128 * kStrideS;         // 36,864
constexpr int kRegion = 2 * kPlane;            // 73,728

typedef __attribute__((ext_vector_type(4))) short short4v;

struct WSplit {
  short8 h[4][2], l[4][2];
};
struct WRef {  // this lane's fragment of K-step 0 in the hi and lo streams; K-steps are 512 fragments apart
  const short8* h;
  const short8* l;
};
__device__ __forceinline__ WRef wref(const unsigned short* base, long long lo_off, size_t frag) {
  return {reinterpret_cast<const short8*>(base) + frag, reinterpret_cast<const short8*>(base + lo_off) + frag};
}
__device__ __forceinline__ void load_wslice(WSplit& w, int kc, const WRef& r) {
  w.h[kc][0] = r.h[kc * 512];
  w.h[kc][1] = r.h[kc * 512 + 64];
  w.l[kc][0] = r.l[kc * 512];
  w.l[kc][1] = r.l[kc * 512 + 64];
}
__device__ __forceinline__ void load_wsplit(WSplit& w, const WRef& r) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) load_wslice(w, kc, r);
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void load_x(short8 (&xh)[4], short8 (&xl)[4], const unsigned char* base, int kc) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    xh[t] = *reinterpret_cast<const short8*>(base + t * 16 * kStrideS + kc * 64);
    xl[t] = *reinterpret_cast<const short8*>(base + kPlane + t * 16 * kStrideS + kc * 64);
  }
}

// acc[t][n] (+)= W x over K = 128 for four token tiles (tile0 .. tile0 + 3) of the (hi, lo) planes at `tile`.
// SWAP = false: weights are the MFMA A operand (lane ends with 8 consecutive output channels of token l15);
// SWAP = true : the tile is the A operand (lane ends with 4 consecutive tokens of one output channel).
// NEXT: each K-step's weight registers are refilled from `next` as soon as its MFMAs are issued.
template <bool SWAP, bool NEXT>
__device__ __forceinline__ void gemm_split_k128(float4v (&acc)[4][2], WSplit& w, const unsigned char* tile, int tile0, int l15, int q4,
                                                const WRef& next) {
  const unsigned char* base = tile + (tile0 * 16 + l15) * kStrideS + q4 * 16;
  short8 xh[2][4], xl[2][4];
  load_x(xh[0], xl[0], base, 0);
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    if (kc < 3) load_x(xh[(kc + 1) & 1], xl[(kc + 1) & 1], base, kc + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (SWAP) {
        acc[t][0] = mma_split3(xh[kc & 1][t], xl[kc & 1][t], w.h[kc][0], w.l[kc][0], acc[t][0]);
        acc[t][1] = mma_split3(xh[kc & 1][t], xl[kc & 1][t], w.h[kc][1], w.l[kc][1], acc[t][1]);
      } else {
        acc[t][0] = mma_split3(w.h[kc][0], w.l[kc][0], xh[kc & 1][t], xl[kc & 1][t], acc[t][0]);
        acc[t][1] = mma_split3(w.h[kc][1], w.l[kc][1], xh[kc & 1][t], xl[kc & 1][t], acc[t][1]);
      }
    }
    if (NEXT) load_wslice(w, kc, next);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ void split8(const float4v& a, const float4v& b, short8& hi, short8& lo) {
  const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  split_op8(f, hi, lo);
}

// LayerNorm (no affine part) of a sample's rows from fp32 global memory in two steps, 8 lanes per row and two rows per
// thread: ln_load issues every load, ln_finish reduces, normalises and splits the result into the two planes of `dst`
// (rows >= nrows become zero rows).  Whatever runs between the two steps hides the memory latency.
// Sum over the 8 lanes that share (lane >> 3): two quad permutes and a half-row mirror, all DPP (no LDS round trip).
__device__ __forceinline__ float sum8(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}

struct LnRows {
  float4 v[2][4];
};
__device__ __forceinline__ void ln_load(LnRows& x, const float* src, int nrows, int tid) {
  const int l8 = tid & 7;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = pass * 64 + (tid >> 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x.v[pass][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nrows) x.v[pass][j] = *reinterpret_cast<const float4*>(src + (size_t)row * kDs + (j * 8 + l8) * 4);
    }
  }
}
// NORM = false: the rows are only split into the planes (no normalisation).
template <bool NORM = true>
__device__ __forceinline__ void ln_finish(LnRows& x, unsigned char* dst, float eps, int tid) {
  const int l8 = tid & 7;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int row = pass * 64 + (tid >> 3);
    float rstd = 1.0f;
    if (NORM) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += (x.v[pass][j].x + x.v[pass][j].y) + (x.v[pass][j].z + x.v[pass][j].w);
      const float mean = sum8(s) * (1.0f / kDs);
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4& q = x.v[pass][j];
        q.x -= mean; q.y -= mean; q.z -= mean; q.w -= mean;
        ss += (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
      }
      rstd = rsqrtf(sum8(ss) * (1.0f / kDs) + eps);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4& q = x.v[pass][j];
      const float o0 = q.x * rstd, o1 = q.y * rstd, o2 = q.z * rstd, o3 = q.w * rstd;
      const unsigned h0 = split_pack2(o0, o1), h1 = split_pack2(o2, o3);
      const float r0 = split_residual(o0, (unsigned short)(h0 & 0xFFFF)), r1 = split_residual(o1, (unsigned short)(h0 >> 16));
      const float r2 = split_residual(o2, (unsigned short)(h1 & 0xFFFF)), r3 = split_residual(o3, (unsigned short)(h1 >> 16));
      unsigned char* p = dst + row * kStrideS + (j * 8 + l8) * 8;
      *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p + kPlane) = make_uint2(split_pack2(r0, r1), split_pack2(r2, r3));
    }
  }
}
}  // namespace splittf
}  // namespace atrihip
