// fp32-grade pre-LN Transformer encoder layer for boards of 129..256 tokens (the 12x21 grid of the large preset), two
// launches per layer: nn.TransformerEncoderLayer(norm_first=True) as instantiated at alphatriangle/nn/model.py:189-196
// with d_model 128 and 4 heads of 32, LayerNorm affine parts folded into the following projections by the host.  The
// reference evaluates the layer in fp32 on any grid (alphatriangle/nn/model.py:283-289, tests/nn/test_model.py:80-81).
//
// Precision scheme as in split_conv_kernels.hip: fp32 activations in memory and in every element-wise step (LayerNorm
// statistics, softmax, residual adds), each matrix product as three 16-bit MFMA passes over (hi, lo) operand splits
// with fp32 accumulation.
//
//   qkv_attn_split_kernel      LayerNorm1 + QKV projection + attention; Q, K and V never go to memory.  Persistent
//                              8-wave workgroups, one sample at a time, the four heads one after the other.  Wave
//                              `oct` owns 32 tokens: it loads their rows once per sample, normalises them in registers
//                              and keeps them as MFMA operand fragments (hi, lo) for all four heads.  Per head the
//                              packed Q / K / V weight fragments (48 KB) sit in LDS - every wave multiplies by the same
//                              weights, so they are fetched once per workgroup by LDS-DMA instead of eight times
//                              through the vector-memory path - and the next head's weights are requested as soon as
//                              the projections of the current one are done, so they arrive under its attention phase.
//                              K and V^T operand images of all 256 key slots (hi, lo: 64 KB) go through LDS
//                              lane-linearly and are read at every use, as in the bf16 kernel (tf_kernels.hip).
//   proj_ffn_rows_split_kernel output projection + residual + LayerNorm2 + feed-forward block over 128-row tiles of the
//                              flat token space: the second half of encoder_layer_split_kernel (split_tf_kernels.hip)
//                              with the attention output staged from memory; persistent workgroups, the next tile's
//                              rows requested before the last GEMM of the current one.
#include "common.hpp"
#include "split_tf_common.hpp"

using namespace atrihip;
using namespace atrihip::splittf;

struct SplitQkvAttnArgs {
  const float* x;              // [B, HW, 128] f32
  float* att;                  // [B, HW, 128] f32 attention output (heads concatenated)
  const unsigned short* wqkv;  // packed [hi | lo], blocks Q (pre-scaled by log2(e)/sqrt(32)), K, V; LayerNorm1 folded in
  const float* bqkv;           // [384]
  long long lo_qkv;
  int B, HW;
  float eps;
  const int* live;             // device row count (nullptr: B rows), see common.hpp
};

struct SplitProjFfnArgs {
  const float* att;            // [rows, 128] f32 attention output
  const float* x;              // [rows, 128] f32 layer input (residual)
  float* out;                  // [rows, 128] f32
  const unsigned short* wo;
  const float* bo;
  const unsigned short* w1;    // [F/128 blocks][4 steps], LayerNorm2 folded in
  const float* b1;             // [F]
  const unsigned short* w2;    // [1][F/32 steps]
  const float* b2;
  long long lo_o, lo_1, lo_2;
  int B, HW, F, act;
  float eps2;
  const int* live;             // device sample count (nullptr: B samples): only the rows of the first *live samples exist
};

namespace {
// ---------------------------------------------------------------------------------------------------------------------
// qkv_attn_split_kernel
constexpr int kQaW = 48 * 1024;  // one head's weights: [Q K V][kc 4][plane 2][n 2][lane 64] x 16 B
constexpr int kQaK = 32 * 1024;  // K image  [key tile 16][plane 2][lane 64] x 16 B
constexpr int kQaV = 32 * 1024;  // V^T image [key block 8][n 2][plane 2][lane 64] x 16 B
constexpr int kQaLds = kQaW + kQaK + kQaV + 384 * 4;

// Requests the 48 one-KB fragment chunks of `head` (six per wave) from the packed streams straight into LDS.
__device__ __forceinline__ void qa_stage_weights(const SplitQkvAttnArgs& a, unsigned char* W, int head, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = wave + i * 8;  // chunk = ((step * 2 + plane) * 2 + n), step = block * 4 + kc
    const int n = c & 1, plane = (c >> 1) & 1, step = c >> 2;
    const unsigned short* base = plane ? a.wqkv + a.lo_qkv : a.wqkv;
    const short8* src = reinterpret_cast<const short8*>(base) + (size_t)step * 512 + head * 128 + n * 64 + lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(W + c * 1024), 16, 0, 0);
  }
}

// acc[t][n] += W_m x for this wave's two token tiles; the weights of block m (0 Q, 1 K, 2 V) come from LDS.
// SWAP = false: weights are the MFMA A operand (lane ends with 8 consecutive output dims of token l15);
// SWAP = true : the tokens are the A operand (lane ends with 4 consecutive tokens of one output dim).
template <bool SWAP>
__device__ __forceinline__ void qa_gemm(float4v (&acc)[2][2], const short8* wlane, int m, const short8 (&xh)[2][4],
                                        const short8 (&xl)[2][4]) {
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    const short8* ws = wlane + (size_t)((m * 4 + kc) * 4) * 64;
    const short8 wh0 = ws[0], wh1 = ws[64], wl0 = ws[128], wl1 = ws[192];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (SWAP) {
        acc[t][0] = mma_split3(xh[t][kc], xl[t][kc], wh0, wl0, acc[t][0]);
        acc[t][1] = mma_split3(xh[t][kc], xl[t][kc], wh1, wl1, acc[t][1]);
      } else {
        acc[t][0] = mma_split3(wh0, wl0, xh[t][kc], xl[t][kc], acc[t][0]);
        acc[t][1] = mma_split3(wh1, wl1, xh[t][kc], xl[t][kc], acc[t][1]);
      }
    }
  }
}

// The two query tiles of one wave against all key slots of one head.  SAFE subtracts the row maxima before exp2; the
// other form relies on the range check of the denominators (returned: true when a lane saw one out of range).
// KTF >= 0: index of the first key tile that holds padding, known at compile time (token count in (16 KTF, 16 KTF + 16]);
// KTF < 0 derives it from the token count.
template <int KTF, bool SAFE>
__device__ __forceinline__ bool qa_tiles(const short8* kimg, const short8* vimg, const short8 (&Qh)[2], const short8 (&Ql)[2],
                                         float* og, int HW, int oct, int q4, int l15) {
  const int kt_full = KTF >= 0 ? KTF : HW >> 4;  // key tiles below this index hold no padding
  bool suspect = false;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int q0 = oct * 32 + t * 16;
    if (q0 >= HW) break;  // wave-uniform
    __builtin_amdgcn_sched_barrier(0);  // keep the query tiles from interleaving: each holds 64 score registers
    float4v st[16];
    float m = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 16; ++kt) {
      const short8 kh = kimg[(kt * 2) * 64], kl = kimg[(kt * 2 + 1) * 64];
      const float init = SAFE ? 0.f : kFreeShift;  // the free pass's shift rides in the accumulators
      st[kt] = mma_split3(kh, kl, Qh[t], Ql[t], (float4v){init, init, init, init});
      if (kt >= kt_full) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (kt * 16 + q4 * 4 + j >= HW) st[kt][j] = -1e30f;
      }
      if (SAFE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, st[kt][j]);
      }
    }
    if (SAFE) m = quad_row_max(m) - kSafeShift;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 16; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st[kt][j] = __builtin_amdgcn_exp2f(SAFE ? st[kt][j] - m : st[kt][j]);  // log2 units
        sum += st[kt][j];
      }
    sum = quad_row_sum(sum);
    if (!SAFE) suspect |= !(sum > kFreeLo && sum < kFreeHi);  // outside the window of the operand format, or NaN
    // O^T = V^T P^T over eight blocks of 32 keys
    float4v o0 = (float4v){0.f, 0.f, 0.f, 0.f}, o1 = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      short8 ph, pl;
      split8(st[2 * kb], st[2 * kb + 1], ph, pl);
      const short8* vb = vimg + (size_t)(kb * 4) * 64;
      o0 = mma_split3(vb[0], vb[64], ph, pl, o0);
      o1 = mma_split3(vb[128], vb[192], ph, pl, o1);
    }
    const float inv = __builtin_amdgcn_rcpf(sum);  // 1 ulp, one instruction instead of the division's ten
    // lane holds dims q4 * 8 .. + 7 of query q0 + l15
    if (q0 + l15 < HW) {
      float* op = og + (size_t)(q0 + l15) * kDs;
      *reinterpret_cast<float4*>(op) = make_float4(o0[0] * inv, o0[1] * inv, o0[2] * inv, o0[3] * inv);
      *reinterpret_cast<float4*>(op + 4) = make_float4(o1[0] * inv, o1[1] * inv, o1[2] * inv, o1[3] * inv);
    }
  }
  return suspect;
}
}  // namespace

template <int KTF>
__global__ __launch_bounds__(512, 2) void qkv_attn_split_kernel(SplitQkvAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* W = smem;
  float* sb = reinterpret_cast<float*>(smem + kQaW + kQaK + kQaV);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), oct = wave;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int HW = a.HW;
  const int nB = live_rows(a.live, a.B);
  if ((int)blockIdx.x >= nB) return;  // whole workgroup, before any barrier
  const short8* wlane = reinterpret_cast<const short8*>(W) + lane;
  short8* kimg = reinterpret_cast<short8*>(smem + kQaW) + lane;
  short8* vimg = reinterpret_cast<short8*>(smem + kQaW + kQaK) + lane;

  if (tid < 384) sb[tid] = a.bqkv[tid];
  qa_stage_weights(a, W, 0, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // biases and the first head's weights are in LDS

  int strikes = 0;  // rejected free softmax passes of this wave (wave-uniform)
  for (int b = blockIdx.x; b < nB; b += gridDim.x) {
    const float* xg = a.x + (size_t)b * HW * kDs;
    // ---- this wave's 32 token rows -> LayerNorm in registers -> (hi, lo) operand fragments for all four heads.
    // A token's 128 channels sit in the four lanes that share l15, 32 values each (channel = 32 kc + 8 q4 + j).
    short8 xh[2][4], xl[2][4];
    {
      float4 xr[2][4][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int tok = min(oct * 32 + t * 16 + l15, HW - 1);  // padding rows repeat the last token (finite, masked as keys)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          const float4* p = reinterpret_cast<const float4*>(xg + (size_t)tok * kDs + kc * 32 + q4 * 8);
          xr[t][kc][0] = p[0];
          xr[t][kc][1] = p[1];
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float s = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
          s += ((xr[t][kc][0].x + xr[t][kc][0].y) + (xr[t][kc][0].z + xr[t][kc][0].w)) +
               ((xr[t][kc][1].x + xr[t][kc][1].y) + (xr[t][kc][1].z + xr[t][kc][1].w));
        const float mean = quad_row_sum(s) * (1.0f / kDs);
        float ss = 0.f;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            float4& q = xr[t][kc][u];
            q.x -= mean; q.y -= mean; q.z -= mean; q.w -= mean;
            ss += (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
          }
        const float rstd = rsqrtf(quad_row_sum(ss) * (1.0f / kDs) + a.eps);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          const float4 q0 = xr[t][kc][0], q1 = xr[t][kc][1];
          const float f[8] = {q0.x * rstd, q0.y * rstd, q0.z * rstd, q0.w * rstd, q1.x * rstd, q1.y * rstd, q1.z * rstd, q1.w * rstd};
          split_op8(f, xh[t][kc], xl[t][kc]);
        }
      }
    }
    const bool more = b + (int)gridDim.x < nB;
    const float* xnext = more ? xg + (size_t)gridDim.x * HW * kDs : xg;

#pragma unroll 1
    for (int head = 0; head < 4; ++head) {
      // ---- K of these 32 tokens -> operand image
      {
        float bias[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bias[i] = sb[kDs + head * 32 + q4 * 8 + i];
        float4v acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t][0] = (float4v){bias[0], bias[1], bias[2], bias[3]};
          acc[t][1] = (float4v){bias[4], bias[5], bias[6], bias[7]};
        }
        qa_gemm<false>(acc, wlane, 1, xh, xl);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          short8 kh, kl;
          split8(acc[t][0], acc[t][1], kh, kl);
          kimg[((oct * 2 + t) * 2) * 64] = kh;
          kimg[((oct * 2 + t) * 2 + 1) * 64] = kl;
        }
      }
      // ---- V with swapped operands: acc[t][n][j] = V[token 16 t + 4 q4 + j][dim perm(l15, n)] -> V^T operand images
      {
        const int chp = (l15 >> 2) * 8 + (l15 & 3);
        const float bv0 = sb[2 * kDs + head * 32 + chp], bv1 = sb[2 * kDs + head * 32 + chp + 4];
        float4v acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t][0] = (float4v){bv0, bv0, bv0, bv0};
          acc[t][1] = (float4v){bv1, bv1, bv1, bv1};
        }
        qa_gemm<true>(acc, wlane, 2, xh, xl);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          short8 vh, vl;
          split8(acc[0][n], acc[1][n], vh, vl);
          vimg[((oct * 2 + n) * 2) * 64] = vh;
          vimg[((oct * 2 + n) * 2 + 1) * 64] = vl;
        }
      }
      // ---- Q of these 32 tokens stays in registers
      short8 Qh[2], Ql[2];
      {
        float bias[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bias[i] = sb[head * 32 + q4 * 8 + i];
        float4v acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t][0] = (float4v){bias[0], bias[1], bias[2], bias[3]};
          acc[t][1] = (float4v){bias[4], bias[5], bias[6], bias[7]};
        }
        qa_gemm<false>(acc, wlane, 0, xh, xl);
#pragma unroll
        for (int t = 0; t < 2; ++t) split8(acc[t][0], acc[t][1], Qh[t], Ql[t]);
      }
      __syncthreads();  // K / V^T images complete; every wave is past its reads of this head's weights
      // the next head's weights (the first head's again after the last: the same for every sample) fly under the attention
      qa_stage_weights(a, W, (head + 1) & 3, wave, lane);
      float touch = 0.f;
      if (head == 3) {
        // one dword of every 128-byte line of the next sample's rows of this wave: they are in L2 when it starts
        const int r = min(oct * 32 + (lane >> 1), HW - 1);
        touch = xnext[(size_t)r * kDs + (lane & 1) * 32] + xnext[(size_t)r * kDs + 64 + (lane & 1) * 32];
      }
      float* og = a.att + (size_t)b * HW * kDs + head * 32 + q4 * 8;
      // Softmax is shift invariant, so the fast pass skips the row maximum and checks the denominators instead: a wave
      // that saw one outside the window of the operand format (split_tf_common.hpp) repeats its tiles with the maximum
      // subtracted and overwrites its outputs; after kFreeStrikes rejections it stops trying for the rest of the launch.
      bool need_safe = strikes >= kFreeStrikes;
      if (!need_safe) {
        const bool suspect = qa_tiles<KTF, false>(kimg, vimg, Qh, Ql, og, HW, oct, q4, l15);
        need_safe = __builtin_amdgcn_ballot_w64(suspect) != 0;
        if (need_safe) ++strikes;
      }
      if (need_safe) (void)qa_tiles<KTF, true>(kimg, vimg, Qh, Ql, og, HW, oct, q4, l15);
      asm volatile("" ::"v"(touch));  // keeps the touch loads alive
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // every wave is done with the images; the next head's weights are in LDS
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// proj_ffn_rows_split_kernel
namespace {
// biases in LDS behind the two plane regions (floats): [bo 128][b2 128][b1 up to 512]
constexpr int kPfBiasO = 0, kPfBias2 = 128, kPfBias1 = 256, kPfBiasFloats = 256 + 512;
constexpr int kPfLds = 2 * kRegion + kPfBiasFloats * 4;
}  // namespace

// Wave w = (channel block cb = w & 3, row half = w >> 2): every GEMM gives a wave 64 rows x 32 channels.  Region A of LDS
// holds the (hi, lo) planes of the GEMM input tile (attention output, later LayerNorm2(y)), region B the per-wave
// LayerNorm2 statistics, later the planes of the hidden tile.  The residual stream y stays in fp32 registers from the
// output projection to the end.  Weights sit in one 64-register ring (the slice a GEMM step has just used is refilled
// with the next GEMM's slice).
template <int HALVES, int ACT>
__global__ __launch_bounds__(512, 2) void proj_ffn_rows_split_kernel(SplitProjFfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* A = smem;
  unsigned char* Bq = smem + kRegion;
  float* sb = reinterpret_cast<float*>(smem + 2 * kRegion);
  const int rows = live_rows(a.live, a.B) * a.HW;
  const int ntiles = (rows + 127) >> 7;
  if ((int)blockIdx.x >= ntiles) return;  // whole workgroup, before any barrier
  // Two waves share a SIMD (w and w + 4); issue arbitration goes by priority, then age, so the second-dispatched half
  // would lose every contested slot.  One static priority step for that half evens the pair out.
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);

  WSplit w;
  {
    const int tid = threadIdx.x;
    if (tid < 128) {
      sb[kPfBiasO + tid] = a.bo[tid];
      sb[kPfBias2 + tid] = a.b2[tid];
    }
    if (tid < HALVES * 128) sb[kPfBias1 + tid] = a.b1[tid];  // visible after the first barrier of the tile loop
    load_wsplit(w, wref(a.wo, a.lo_o, (size_t)((tid >> 6) & 3) * 128 + (tid & 63)));
    LnRows first;
    const int r0 = (int)blockIdx.x * 128;
    ln_load(first, a.att + (size_t)r0 * kDs, min(128, rows - r0), tid);
    ln_finish<false>(first, A, 0.f, tid);
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // Everything below derives from a thread index the optimiser cannot see through: hoisting the tile-invariant bias
    // loads and addresses out of the loop would cost more registers than the kernel has.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int q4 = lane >> 4, l15 = lane & 15;
    const int cb = wave & 3, half = wave >> 2, tile0 = half * 4;
    const size_t frag = (size_t)cb * 128 + lane;
    const int ch0 = cb * 32 + q4 * 8;
    const int row0 = tile * 128, nrows = min(128, rows - row0);
    const float* xg = a.x + (size_t)row0 * kDs;
    const bool more = tile + (int)gridDim.x < ntiles;
    const int next0 = more ? row0 + (int)gridDim.x * 128 : row0;
    LnRows nxt;  // the next tile's attention rows (this tile's again when there is none: loaded, never used)
    __syncthreads();  // attention-output planes complete

    // ---- output projection + residual: y stays in fp32 registers until the end
    float4v y[4][2];
    float2* stats = reinterpret_cast<float2*>(Bq);  // [row][channel block] (mean, M2) of 32 channels
    {
      float4 res[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // unpredicated (padding rows read the last real row; their sums are never stored): no branch between these loads
        // and the GEMM, so the compiler lets them fly through it
        const int r = min((tile0 + t) * 16 + l15, nrows - 1);
        const float4* p = reinterpret_cast<const float4*>(xg + (size_t)r * kDs + ch0);
        res[t][0] = p[0];
        res[t][1] = p[1];
      }
      float bias[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) bias[i] = sb[kPfBiasO + ch0 + i];
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
    __syncthreads();  // statistics of all four channel blocks in place; every wave is past its out-projection reads of region A
    // ---- LayerNorm2(y) from registers into the planes of region A (per-wave mean / M2 combined with Chan's formula)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = (tile0 + t) * 16 + l15;
      const float4 s01 = *reinterpret_cast<const float4*>(stats + r * 4), s23 = *reinterpret_cast<const float4*>(stats + r * 4 + 2);
      const float mean = (s01.x + s01.z + s23.x + s23.z) * 0.25f;
      const float e0 = s01.x - mean, e1 = s01.z - mean, e2 = s23.x - mean, e3 = s23.z - mean;
      const float m2 = (s01.y + s01.w + s23.y + s23.w) + 32.0f * ((e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3));
      const float rstd = r < nrows ? rsqrtf(m2 * (1.0f / kDs) + a.eps2) : 0.f;  // rows beyond the tile's end become zero rows
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

    // ---- feed-forward: hidden layer in 128-wide halves through region B, second GEMM accumulates onto y (+ b2)
    {
      float b2v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) b2v[i] = sb[kPfBias2 + ch0 + i];
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
        for (int i = 0; i < 8; ++i) bias[i] = sb[kPfBias1 + h * 128 + ch0 + i];
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
      __syncthreads();  // hidden tile complete; with the last half, every wave is also past its reads of region A
      if (h + 1 < HALVES) {
        gemm_split_k128<false, true>(y, w, Bq, tile0, l15, q4, wref(a.w1, a.lo_1, frag + (size_t)(h + 1) * 4 * 512));
      } else {
        // the next tile's attention rows fly during the last GEMM, and the output-projection weights refill the ring
        ln_load(nxt, a.att + (size_t)next0 * kDs, min(128, rows - next0), tid);
        gemm_split_k128<false, true>(y, w, Bq, tile0, l15, q4, wref(a.wo, a.lo_o, frag));
      }
    }

    float* og = a.out + (size_t)row0 * kDs;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = (tile0 + t) * 16 + l15;
      if (r >= nrows) continue;
      float* op = og + (size_t)r * kDs + ch0;
      *reinterpret_cast<float4*>(op) = make_float4(y[t][0][0], y[t][0][1], y[t][0][2], y[t][0][3]);
      *reinterpret_cast<float4*>(op + 4) = make_float4(y[t][1][0], y[t][1][1], y[t][1][2], y[t][1][3]);
    }
    if (more) ln_finish<false>(nxt, A, 0.f, tid);  // region A is free since the barrier before the last GEMM
  }
}

namespace {
int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus;
}

template <int KTF>
int launch_qa(const SplitQkvAttnArgs& a, int grid, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)qkv_attn_split_kernel<KTF>, hipFuncAttributeMaxDynamicSharedMemorySize, kQaLds);
    attr = true;
  }
  hipLaunchKernelGGL((qkv_attn_split_kernel<KTF>), dim3(grid), dim3(512), kQaLds, stream, a);
  return hip_err(hipGetLastError());
}

template <int HALVES, int ACT>
int launch_pf_inst(const SplitProjFfnArgs& a, int grid, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)proj_ffn_rows_split_kernel<HALVES, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, kPfLds);
    attr = true;
  }
  hipLaunchKernelGGL((proj_ffn_rows_split_kernel<HALVES, ACT>), dim3(grid), dim3(512), kPfLds, stream, a);
  return hip_err(hipGetLastError());
}
template <int HALVES>
int launch_pf(const SplitProjFfnArgs& a, int grid, hipStream_t stream) {
  return a.act == kActReLU ? launch_pf_inst<HALVES, kActReLU>(a, grid, stream) : launch_pf_inst<HALVES, kActGELU>(a, grid, stream);
}
}  // namespace

extern "C" {

// LayerNorm1 + QKV + 4-head attention of one encoder layer, fp32-grade, up to 256 tokens:
// x [B, HW, 128] f32 -> att [B, HW, 128] f32 (att != x).  LayerNorm1's affine part is folded into (wqkv, bqkv) by the
// host; Q weights are pre-scaled by log2(e) / sqrt(32).
int atri_split_qkv_attn(const float* x, float* att, const void* wqkv, long long lo_qkv, const float* bqkv, float eps, int B,
                        int HW, hipStream_t stream) {
  if (B <= 0) return 0;
  if (HW < 1 || HW > 256) return -10;
  if (x == att) return -13;
  const int cus = device_cus();
  if (cus < 0) return -20;
  SplitQkvAttnArgs a{};
  a.x = x;
  a.att = att;
  a.wqkv = (const unsigned short*)wqkv;
  a.lo_qkv = lo_qkv;
  a.bqkv = bqkv;
  a.B = B;
  a.HW = HW;
  a.eps = eps;
  a.live = live_count_ptr();
  const int grid = B < cus ? B : cus;
  return HW > 240 ? launch_qa<15>(a, grid, stream) : launch_qa<-1>(a, grid, stream);
}

// Output projection + residual + LayerNorm2 + feed-forward block of one encoder layer, fp32-grade, over the flat token
// rows of B samples: out = y + W2 act(W1 LN2(y)) + b2, y = x + att Wo + bo (out may not alias x or att).
int atri_split_proj_ffn_rows(const float* att, const float* x, float* out, const void* wo, long long lo_o, const float* bo,
                             const void* w1, long long lo_1, const float* b1, const void* w2, long long lo_2, const float* b2,
                             float eps2, int B, int HW, int F, int act, hipStream_t stream) {
  if (B <= 0) return 0;
  if (HW < 1 || F % 128 != 0 || F < 128 || F > 512) return -10;
  if (act != kActReLU && act != kActGELU) return -11;
  if (out == x || out == att) return -13;
  const int cus = device_cus();
  if (cus < 0) return -20;
  SplitProjFfnArgs a{};
  a.att = att;
  a.x = x;
  a.out = out;
  a.wo = (const unsigned short*)wo; a.lo_o = lo_o; a.bo = bo;
  a.w1 = (const unsigned short*)w1; a.lo_1 = lo_1; a.b1 = b1;
  a.w2 = (const unsigned short*)w2; a.lo_2 = lo_2; a.b2 = b2;
  a.B = B; a.HW = HW; a.F = F; a.act = act;
  a.eps2 = eps2;
  a.live = live_count_ptr();
  const long long tiles = ((long long)B * HW + 127) / 128;
  const int grid = tiles < cus ? (int)tiles : cus;
  switch (F / 128) {
    case 1: return launch_pf<1>(a, grid, stream);
    case 2: return launch_pf<2>(a, grid, stream);
    case 3: return launch_pf<3>(a, grid, stream);
    default: return launch_pf<4>(a, grid, stream);
  }
}

}  // extern "C"
