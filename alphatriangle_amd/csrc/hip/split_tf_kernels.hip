// fp32-grade pre-LN Transformer encoder layer in one launch, 8-wave workgroups that take one sample at a time (up to 128 tokens,
// d_model 128, 4 heads of 32): nn.TransformerEncoderLayer(norm_first=True) as instantiated at
// alphatriangle/nn/model.py:189-196, LayerNorm affine parts folded into the following projections by the host.
//
// Precision scheme as in split_conv_kernels.hip: fp32 activations in memory and in every element-wise step
// (LayerNorm statistics, softmax, residual adds), each matrix product as three 16-bit MFMA passes over (hi, lo)
// operand splits with fp32 accumulation.
//
// Work split: wave w = (channel block / head h = w & 3, token half = w >> 2); every GEMM gives a wave 64 tokens x
// 32 channels.  LDS holds two 72 KB regions:
//   A: (hi, lo) planes of the GEMM input tile - LayerNorm1(x), later the attention output, later LayerNorm2(y);
//   B: the K operand image of all heads (read at use by the attention loop), later the per-wave LayerNorm2 statistics,
//      later the (hi, lo) planes of the FFN hidden tile.
// Q and V^T stay in registers: a wave attends for its own 64 queries and computes V^T of its head for all 128 keys
// itself (the second half is redundant work on an otherwise idle matrix pipe and saves an exchange with two
// barriers); K of a head is produced half by each wave of the head's pair and read from a lane-linear operand image.
// The residual stream y stays in fp32 registers from the output projection to the end of the layer; LayerNorm2 works
// on those registers (per-wave mean / M2 over its 32 channels, combined across the four channel blocks through 4 KB
// of LDS).
//
// Latency hiding, one workgroup per CU: weights sit in one 64-register ring (the slice a GEMM step has just used is
// refilled with the next GEMM's slice), GEMM input fragments are double buffered, the scores of the next query tile
// are issued before the softmax of the current one, and the workgroup is persistent (one per CU, samples strided by the
// grid): the next sample's input rows are requested before the last GEMM of the current sample and normalised after it.
#include "common.hpp"
#include "split_tf_common.hpp"

using namespace atrihip;

struct SplitEncArgs {
  const float* x;             // [B, HW, 128] f32
  float* out;                 // [B, HW, 128] f32
  const unsigned short* wqkv; // packed [hi | lo], blocks Q (pre-scaled by log2(e)/sqrt(32)), K, V
  const float* bqkv;          // [384]
  const unsigned short* wo;
  const float* bo;
  const unsigned short* w1;   // [F/128 blocks][4 steps]
  const float* b1;            // [F]
  const unsigned short* w2;   // [1][F/32 steps]
  const float* b2;
  long long lo_qkv, lo_o, lo_1, lo_2;
  int B, HW, F, act;
  float eps1, eps2;
  const int* live;            // device row count (nullptr: B rows), see common.hpp
  long long* phase_clock;     // diagnostics: workgroup `phase_block` stores the shader clock at its phase boundaries
  int phase_block;
};

// Only the PROBE instantiation carries the stamps: their branches would otherwise sit between the weight prefetches and
// the GEMMs of the production kernel, where the compiler answers every control-flow join with a full wait (2 - 3 % slower).
#define ATRI_PHASE(i)                                                                                               \
  do {                                                                                                              \
    if constexpr (PROBE) {                                                                                          \
      if (a.phase_clock != nullptr && (int)blockIdx.x == a.phase_block && iter == 1 && tid == 0) a.phase_clock[i] = clock64(); \
    }                                                                                                               \
  } while (0)

namespace {
using namespace atrihip::splittf;
// Biases of the layer live in LDS behind the two regions (floats): [bqkv 384][bo 128][b2 128][b1 up to 512].  A bias
// read from global memory right before its GEMM would be the youngest outstanding load, and waiting for it drains
// every older one - the weight slices and residual rows that are meant to stay in flight.
constexpr int kBiasQkv = 0, kBiasO = 384, kBias2 = 512, kBias1 = 640, kBiasFloats = 640 + 512;
constexpr int kEncLds = 2 * kRegion + kBiasFloats * 4;  // 152,064

// Scores of query tile t of this wave against all 128 keys of its head, key-major: st[kt][j] = score of key
// 16 kt + 4 q4 + j for query l15 of the tile, in log2 units.  Padded keys get -1e30 (probability 0): branch-free, and
// with TAIL (more than 112 tokens) only the last key tile can hold padding.
// The accumulators start at `init`: a constant added to every score costs nothing there.
template <bool TAIL>
__device__ __forceinline__ void attn_scores(float4v (&st)[8], const short8* kimg, const short8& qh, const short8& ql, int HW, int q4,
                                            float init) {
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    const short8 kh = kimg[(size_t)kt * 128], kl = kimg[(size_t)kt * 128 + 64];
    st[kt] = mma_split3(kh, kl, qh, ql, (float4v){init, init, init, init});
  }
#pragma unroll
  for (int kt = TAIL ? 7 : 0; kt < 8; ++kt)
#pragma unroll
    for (int j = 0; j < 4; ++j) st[kt][j] = kt * 16 + q4 * 4 + j >= HW ? -1e30f : st[kt][j];
}

// Attention of head cb for this wave's four query tiles; the output rows go to the planes at A.  SAFE subtracts the
// row maxima before exp2; the other form leaves range checking of the denominators to the caller (returned: true when
// a lane saw one out of range).  The max-free form also starts the loads of the weights `after`.  Softmax is shift invariant and the (hi, lo) split keeps fp32's exponent range, so the
// max-free form is exact as long as no denominator overflows or vanishes.
template <bool SAFE, bool TAIL>
__device__ __forceinline__ bool attention_tiles(const short8 (&qh)[4], const short8 (&ql)[4], const short8 (&vth)[4][2],
                                                const short8 (&vtl)[4][2], const short8* kimg, unsigned char* A, int HW, int tile0,
                                                int cb, int q4, int l15, WSplit& w, const WRef& after) {
  bool suspect = false;
  // One score tile at a time: a second one in flight (scores of tile t + 1 under the softmax of tile t) costs 32
  // registers that the kernel does not have - they came back as spills whose reloads drain the in-flight loads.  The
  // phase is bound by the vector pipe over both waves of a SIMD, so the matrix work of one wave hides under the other's.
  float4v s[8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    attn_scores<TAIL>(s, kimg, qh[t], ql[t], HW, q4, SAFE ? 0.f : kFreeShift);  // the free pass's shift rides in the accumulators
    if (!SAFE && t == 3) {  // the weights of the GEMM after attention fly during the last tile (the ring is idle until then)
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) load_wslice(w, kc, after);
    }
    float m = 0.f;
    if (SAFE) {
      m = -1e30f;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, s[kt][j]);
      m = quad_row_max(m) - kSafeShift;
    }
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[kt][j] = __builtin_amdgcn_exp2f(SAFE ? s[kt][j] - m : s[kt][j]);
        sum += s[kt][j];
      }
    sum = quad_row_sum(sum);
    if (!SAFE) suspect |= !(sum > kFreeLo && sum < kFreeHi);  // outside the window of the operand format, or NaN
    float4v o0 = (float4v){0.f, 0.f, 0.f, 0.f}, o1 = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      short8 ph, pl;
      split8(s[2 * kb], s[2 * kb + 1], ph, pl);
      o0 = mma_split3(vth[kb][0], vtl[kb][0], ph, pl, o0);
      o1 = mma_split3(vth[kb][1], vtl[kb][1], ph, pl, o1);
    }
    const float inv = __builtin_amdgcn_rcpf(sum);  // 1 ulp: the division's scale / fixup sequence is ten more vector instructions per tile
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o0[j] *= inv;
      o1[j] *= inv;
    }
    // lane holds channels cb*32 + q4*8 .. +7 of query (tile0 + t)*16 + l15
    short8 oh, ol;
    split8(o0, o1, oh, ol);
    unsigned char* p = A + ((tile0 + t) * 16 + l15) * kStrideS + cb * 64 + q4 * 16;
    *reinterpret_cast<short8*>(p) = oh;
    *reinterpret_cast<short8*>(p + kPlane) = ol;
  }
  return suspect;
}
}  // namespace

template <int HALVES, bool PROBE, int ACT>
__global__ __launch_bounds__(512, 2) void encoder_layer_split_kernel(SplitEncArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* A = smem;
  unsigned char* Bq = smem + kRegion;
  float* sb = reinterpret_cast<float*>(smem + 2 * kRegion);
  const int HW = a.HW;
  const int nB = live_rows(a.live, a.B);
  if ((int)blockIdx.x >= nB) return;  // whole workgroup, before any barrier
  // Two waves share a SIMD (w and w + 4); issue arbitration goes by priority, then age, so the second-dispatched half
  // would lose every contested slot.  One static priority step for that half evens the pair out.
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
  const WRef none{nullptr, nullptr};

  // The workgroup is persistent: it walks samples blockIdx.x, blockIdx.x + gridDim.x, ...  The input rows of the next
  // sample are requested before the last GEMM of the current one and normalised right after it, so that the only
  // exposed memory latency is that of the first sample.
  WSplit w;
  {
    const int tid = threadIdx.x;
    if (tid < 384) sb[kBiasQkv + tid] = a.bqkv[tid];
    if (tid < 128) {
      sb[kBiasO + tid] = a.bo[tid];
      sb[kBias2 + tid] = a.b2[tid];
    }
    if (tid < HALVES * 128) sb[kBias1 + tid] = a.b1[tid];  // visible after the first barrier of the sample loop
    load_wsplit(w, wref(a.wqkv, a.lo_qkv, (size_t)((tid >> 6) & 3) * 128 + (tid & 63) + (size_t)4 * 512));  // K weights fly during LayerNorm1
    LnRows first;
    ln_load(first, a.x + (size_t)blockIdx.x * HW * kDs, HW, tid);
    ln_finish(first, A, a.eps1, tid);
  }
  int iter = 0;
  int strikes = 0;  // rejected free softmax passes of this wave (wave-uniform)
  for (int b = blockIdx.x; b < nB; b += gridDim.x, ++iter) {
  // Everything below derives from a thread index the optimiser cannot see through: hoisting the sample-invariant
  // bias loads and addresses out of the loop would cost more registers than the kernel has.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int cb = wave & 3, half = wave >> 2, tile0 = half * 4;
  const size_t frag = (size_t)cb * 128 + lane;
  const int ch0 = cb * 32 + q4 * 8;
  const WRef wk = wref(a.wqkv, a.lo_qkv, frag + (size_t)4 * 512);
  const float* xg = a.x + (size_t)b * HW * kDs;
  const bool more = b + (int)gridDim.x < nB;
  LnRows rows;  // the next sample's input rows (this sample's again when there is none: loaded, never used)
  ATRI_PHASE(0);
  __syncthreads();  // LayerNorm1(x) planes complete
  ATRI_PHASE(1);

  // ---- K (own 64 tokens of head cb) -> operand image in region B: [head][key tile][plane][lane]
  {
    float bias[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bias[i] = sb[kBiasQkv + kDs + ch0 + i];
    float4v acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t][0] = (float4v){bias[0], bias[1], bias[2], bias[3]};
      acc[t][1] = (float4v){bias[4], bias[5], bias[6], bias[7]};
    }
    gemm_split_k128<false, true>(acc, w, A, tile0, l15, q4, wref(a.wqkv, a.lo_qkv, frag + (size_t)8 * 512));  // V weights refill the ring
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      short8 kh, kl;
      split8(acc[t][0], acc[t][1], kh, kl);
      short8* img = reinterpret_cast<short8*>(Bq) + ((size_t)(cb * 8 + tile0 + t) * 2) * 64 + lane;
      img[0] = kh;
      img[64] = kl;
    }
  }
  ATRI_PHASE(2);
  // ---- V^T of head cb for all 128 keys, swapped operands: acc[t][n][j] = V[token (4 r + t) * 16 + 4 q4 + j][dim perm(l15, n)]
  short8 vth[4][2], vtl[4][2];
  {
    const int chp = (l15 >> 2) * 8 + (l15 & 3);
    const float bv0 = sb[kBiasQkv + 2 * kDs + cb * 32 + chp], bv1 = sb[kBiasQkv + 2 * kDs + cb * 32 + chp + 4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float4v acc[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t][0] = (float4v){bv0, bv0, bv0, bv0};
        acc[t][1] = (float4v){bv1, bv1, bv1, bv1};
      }
      if (r == 0) {
        gemm_split_k128<true, false>(acc, w, A, 0, l15, q4, none);
      } else {
        gemm_split_k128<true, true>(acc, w, A, 4, l15, q4, wref(a.wqkv, a.lo_qkv, frag));  // Q weights refill the ring
      }
#pragma unroll
      for (int kbl = 0; kbl < 2; ++kbl)
#pragma unroll
        for (int n = 0; n < 2; ++n) split8(acc[2 * kbl][n], acc[2 * kbl + 1][n], vth[2 * r + kbl][n], vtl[2 * r + kbl][n]);
    }
  }
  ATRI_PHASE(3);
  // ---- Q: lane ends with dims 8 q4 .. 8 q4 + 7 of token (tile0 + t) * 16 + l15
  short8 qh[4], ql[4];
  {
    float bias[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bias[i] = sb[kBiasQkv + ch0 + i];
    float4v acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t][0] = (float4v){bias[0], bias[1], bias[2], bias[3]};
      acc[t][1] = (float4v){bias[4], bias[5], bias[6], bias[7]};
    }
    gemm_split_k128<false, false>(acc, w, A, tile0, l15, q4, none);
#pragma unroll
    for (int t = 0; t < 4; ++t) split8(acc[t][0], acc[t][1], qh[t], ql[t]);
  }
  // One dword of every 128-byte line of this sample's input rows: the output projection adds them as the residual, and
  // by then they may have left L2 (the other workgroups of the XCD stream their own rows and outputs through it).
  const float touch = xg[min(tid * 32, HW * kDs - 1)];
  ATRI_PHASE(4);
  __syncthreads();  // K image complete; every wave is past its reads of LayerNorm1(x): region A takes the attention output
  ATRI_PHASE(5);

  // ---- attention of head cb for this wave's four query tiles
  {
    const short8* kimg = reinterpret_cast<const short8*>(Bq) + (size_t)(cb * 8) * 2 * 64 + lane;
    const WRef wo = wref(a.wo, a.lo_o, frag);
    bool need_safe = strikes >= kFreeStrikes, free_ran = false;
    if (!need_safe) {
      bool suspect;
      if (HW > 112) {
        suspect = attention_tiles<false, true>(qh, ql, vth, vtl, kimg, A, HW, tile0, cb, q4, l15, w, wo);
      } else {
        suspect = attention_tiles<false, false>(qh, ql, vth, vtl, kimg, A, HW, tile0, cb, q4, l15, w, wo);
      }
      free_ran = true;
      need_safe = __builtin_amdgcn_ballot_w64(suspect) != 0;
      if (need_safe) ++strikes;
    }
    if (need_safe) {
      (void)attention_tiles<true, false>(qh, ql, vth, vtl, kimg, A, HW, tile0, cb, q4, l15, w, wo);
      if (!free_ran) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) load_wslice(w, kc, wo);  // the free pass starts these loads itself
      }
    }
  }
  ATRI_PHASE(6);
  __syncthreads();  // attention output complete; the K image in region B is dead
  ATRI_PHASE(7);

  // ---- output projection + residual: y stays in fp32 registers until the end of the layer
  float4v y[4][2];
  float2* stats = reinterpret_cast<float2*>(Bq);  // [token][channel block] (mean, M2) of 32 channels
  {
    float4 res[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // unpredicated (padding rows read the last real row; their sums are never stored): no branch between these loads
      // and the GEMM, so the compiler lets them fly through it
      const int r = min((tile0 + t) * 16 + l15, HW - 1);
      const float4* p = reinterpret_cast<const float4*>(xg + (size_t)r * kDs + ch0);
      res[t][0] = p[0];
      res[t][1] = p[1];
    }
    asm volatile("" ::"v"(touch));  // keeps the touch load alive
    if constexpr (PROBE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATRI_PHASE(17);
    float bias[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bias[i] = sb[kBiasO + ch0 + i];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      y[t][0] = (float4v){bias[0], bias[1], bias[2], bias[3]};
      y[t][1] = (float4v){bias[4], bias[5], bias[6], bias[7]};
    }
    gemm_split_k128<false, true>(y, w, A, tile0, l15, q4, wref(a.w1, a.lo_1, frag));  // first W1 block refills the ring
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      y[t][0][0] += res[t][0].x; y[t][0][1] += res[t][0].y; y[t][0][2] += res[t][0].z; y[t][0][3] += res[t][0].w;
      y[t][1][0] += res[t][1].x; y[t][1][1] += res[t][1].y; y[t][1][2] += res[t][1].z; y[t][1][3] += res[t][1].w;
      // this wave's share of the row statistics: mean and centred sum of squares over its 32 channels
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += y[t][0][j] + y[t][1][j];
      const float mean_w = quad_row_sum(s) * (1.0f / 32.0f);
      float m2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d0 = y[t][0][j] - mean_w, d1 = y[t][1][j] - mean_w;
        m2 += d0 * d0 + d1 * d1;
      }
      m2 = quad_row_sum(m2);
      if (q4 == 0) stats[((tile0 + t) * 16 + l15) * 4 + cb] = make_float2(mean_w, m2);
    }
  }
  ATRI_PHASE(8);
  __syncthreads();  // statistics of all four channel blocks in place; every wave is past its out-projection reads of region A
  ATRI_PHASE(9);
  // ---- LayerNorm2(y) from registers into the planes of region A
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = (tile0 + t) * 16 + l15;
    const float4 s01 = *reinterpret_cast<const float4*>(stats + r * 4), s23 = *reinterpret_cast<const float4*>(stats + r * 4 + 2);
    const float mean = (s01.x + s01.z + s23.x + s23.z) * 0.25f;
    const float e0 = s01.x - mean, e1 = s01.z - mean, e2 = s23.x - mean, e3 = s23.z - mean;
    const float m2 = (s01.y + s01.w + s23.y + s23.w) + 32.0f * ((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3));
    const float rstd = r < HW ? rsqrtf(m2 * (1.0f / kDs) + a.eps2) : 0.f;  // rows beyond the sample become zero rows
    float4v n0, n1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      n0[j] = (y[t][0][j] - mean) * rstd;
      n1[j] = (y[t][1][j] - mean) * rstd;
    }
    short8 nh, nl;
    split8(n0, n1, nh, nl);
    unsigned char* p = A + r * kStrideS + cb * 64 + q4 * 16;
    *reinterpret_cast<short8*>(p) = nh;
    *reinterpret_cast<short8*>(p + kPlane) = nl;
  }
  __syncthreads();  // LayerNorm2(y) planes complete; the statistics in region B are dead
  ATRI_PHASE(10);

  // ---- feed-forward: hidden layer in 128-wide halves through region B, second GEMM accumulates onto y (+ b2)
  {
    float b2v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b2v[i] = sb[kBias2 + ch0 + i];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[t][0][j] += b2v[j];
        y[t][1][j] += b2v[4 + j];
      }
  }
#pragma unroll
  for (int h = 0; h < HALVES; ++h) {
    {
      float bias[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) bias[i] = sb[kBias1 + h * 128 + ch0 + i];
      float4v acc[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t][0] = (float4v){bias[0], bias[1], bias[2], bias[3]};
        acc[t][1] = (float4v){bias[4], bias[5], bias[6], bias[7]};
      }
      gemm_split_k128<false, true>(acc, w, A, tile0, l15, q4, wref(a.w2, a.lo_2, frag + (size_t)h * 4 * 512));  // this half's W2 steps
      if (h > 0) __syncthreads();  // the previous half's second GEMM has finished reading the hidden tile
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4v lo = acc[t][0], hi = acc[t][1];
        if (ACT == kActReLU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lo[j] = fmaxf(lo[j], 0.f);
            hi[j] = fmaxf(hi[j], 0.f);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lo[j] = 0.5f * lo[j] * (1.0f + erff(lo[j] * 0.70710678118654752f));
            hi[j] = 0.5f * hi[j] * (1.0f + erff(hi[j] * 0.70710678118654752f));
          }
        }
        short8 hh, hl;
        split8(lo, hi, hh, hl);
        unsigned char* p = Bq + ((tile0 + t) * 16 + l15) * kStrideS + cb * 64 + q4 * 16;
        *reinterpret_cast<short8*>(p) = hh;
        *reinterpret_cast<short8*>(p + kPlane) = hl;
      }
    }
    ATRI_PHASE(11 + 2 * h);
    __syncthreads();  // hidden tile complete; with the last half, every wave is also past its reads of region A
    if (h + 1 < HALVES) {
      gemm_split_k128<false, true>(y, w, Bq, tile0, l15, q4, wref(a.w1, a.lo_1, frag + (size_t)(h + 1) * 4 * 512));
    } else {
      ln_load(rows, more ? xg + (size_t)gridDim.x * HW * kDs : xg, HW, tid);  // next sample's rows fly during the last GEMM
      gemm_split_k128<false, true>(y, w, Bq, tile0, l15, q4, wk);           // and the K weights refill the ring
    }
    ATRI_PHASE(12 + 2 * h);
  }

  float* og = a.out + (size_t)b * HW * kDs;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = (tile0 + t) * 16 + l15;
    if (r >= HW) continue;
    float* op = og + (size_t)r * kDs + ch0;
    *reinterpret_cast<float4*>(op) = make_float4(y[t][0][0], y[t][0][1], y[t][0][2], y[t][0][3]);
    *reinterpret_cast<float4*>(op + 4) = make_float4(y[t][1][0], y[t][1][1], y[t][1][2], y[t][1][3]);
  }
  ATRI_PHASE(18);
  if (more) ln_finish(rows, A, a.eps1, tid);  // region A is free since the barrier before the last GEMM
  ATRI_PHASE(19);
  }
}

namespace {
template <int HALVES, bool PROBE, int ACT>
int launch_enc_inst(const SplitEncArgs& a, int grid, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)encoder_layer_split_kernel<HALVES, PROBE, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, kEncLds);
    attr = true;
  }
  hipLaunchKernelGGL((encoder_layer_split_kernel<HALVES, PROBE, ACT>), dim3(grid), dim3(512), kEncLds, stream, a);
  return hip_err(hipGetLastError());
}

template <int HALVES>
int launch_enc(const SplitEncArgs& a, hipStream_t stream) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -20;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int grid = a.B < cus ? a.B : cus;
  const bool relu = a.act == kActReLU;
  if (a.phase_clock != nullptr)  // diagnostics build of the kernel
    return relu ? launch_enc_inst<HALVES, true, kActReLU>(a, grid, stream) : launch_enc_inst<HALVES, true, kActGELU>(a, grid, stream);
  return relu ? launch_enc_inst<HALVES, false, kActReLU>(a, grid, stream) : launch_enc_inst<HALVES, false, kActGELU>(a, grid, stream);
}
}  // namespace

namespace {
long long* g_phase_clock = nullptr;
int g_phase_block = 0;
}  // namespace

extern "C" {

// Diagnostics: workgroup `block` of every following atri_split_encoder_layer launch stores the shader clock at its
// phase boundaries into clock[0..19] (device memory); nullptr switches it off.
int atri_split_encoder_phase_clock(long long* clock, int block) {
  g_phase_clock = clock;
  g_phase_block = block;
  return 0;
}

// One encoder layer, fp32-grade: x [B, HW, 128] f32 -> out [B, HW, 128] f32 (out != x).  LayerNorm affine parts are
// folded into (wqkv, bqkv) and (w1, b1) by the host; Q weights are pre-scaled by log2(e) / sqrt(32).
int atri_split_encoder_layer(const float* x, float* out, const void* wqkv, long long lo_qkv, const float* bqkv, const void* wo,
                             long long lo_o, const float* bo, float eps1, const void* w1, long long lo_1, const float* b1,
                             const void* w2, long long lo_2, const float* b2, float eps2, int B, int HW, int F, int act,
                             hipStream_t stream) {
  if (B <= 0) return 0;
  if (HW < 1 || HW > 128 || F % 128 != 0 || F < 128 || F > 512) return -10;
  if (act != kActReLU && act != kActGELU) return -11;
  if (x == out) return -13;
  SplitEncArgs a{};
  a.x = x;
  a.out = out;
  a.wqkv = (const unsigned short*)wqkv; a.lo_qkv = lo_qkv; a.bqkv = bqkv;
  a.wo = (const unsigned short*)wo; a.lo_o = lo_o; a.bo = bo;
  a.w1 = (const unsigned short*)w1; a.lo_1 = lo_1; a.b1 = b1;
  a.w2 = (const unsigned short*)w2; a.lo_2 = lo_2; a.b2 = b2;
  a.B = B; a.HW = HW; a.F = F; a.act = act;
  a.eps1 = eps1; a.eps2 = eps2;
  a.phase_clock = g_phase_clock; a.phase_block = g_phase_block;
  a.live = live_count_ptr();
  switch (F / 128) {
    case 1: return launch_enc<1>(a, stream);
    case 2: return launch_enc<2>(a, stream);
    case 3: return launch_enc<3>(a, stream);
    default: return launch_enc<4>(a, stream);
  }
}

}  // extern "C"
