// fp32-grade dense tail of the network: shared FC over the flattened body (split-K, final LayerNorm fused into the
// staging), policy / value heads and the C51 expectation (alphatriangle/nn/model.py:232-266,
// alphatriangle/nn/network.py:157-162).  Activations are fp32 in memory and in every element-wise step; each GEMM is
// three 16-bit MFMA passes over (hi, lo) operand splits (see split_conv_kernels.hip for the scheme).
#include "common.hpp"

using namespace atrihip;

namespace {
constexpr int kPitch = 288;  // bytes per 128-wide 16-bit LDS row (256 + 32): 16-byte fragment reads of 16 rows x 4 column groups hit distinct banks
}

// Shared FC fed directly by the body's [B][HW][128] fp32 rows, 64 samples x one K split per workgroup.  The final
// LayerNorm - when there is one - is applied in fp32 to each 128-wide pixel row while it is staged (16 lanes per row;
// LN = 1: normalisation only, the affine part folded into the weights by the host; LN = 2: with the affine part),
// the row is then split into the two 16-bit planes; the "other" features are read from the caller's f32 rows as the last
// K steps.  steps_per_split is a multiple of 4, so every staged chunk of 4 K-steps is exactly one pixel row.
//
// The workgroup is specialised: waves 4-7 are producers (global rows -> LayerNorm -> split -> LDS tile), waves 0-3 are
// consumers (wave = 32 output channels; three MFMA passes per product out of the tile).  One pixel row is only 96
// MFMAs per consumer wave - far less than a memory round trip - so the producers keep kFcDepth pixel rows in flight in
// the registers the consumers spend on accumulators and weights; with one barrier per pixel row the consumers never
// wait for memory, only for the producers' staging arithmetic, which runs on the vector pipe beside their MFMAs.
// Both loops are straight-line code (addresses are clamped instead of predicated, rows past the live count compute
// on a valid row and are never stored): only then does the compiler's wait-count bookkeeping leave the younger loads
// in flight while it waits for the oldest.
// kFcDepth: pixel rows requested ahead of the one being staged.
template <int MT, int LN, int kFcDepth>
__global__ __launch_bounds__(512, 1) void fc_rows_split_kernel(const float* __restrict__ rows, const float* __restrict__ other,
                                                              int other_dim, const float* __restrict__ g,
                                                              const float* __restrict__ bt, float eps,
                                                              const unsigned short* __restrict__ Wp, long long lo_off,
                                                              float* __restrict__ slabs, int B, int HW, int other_pad, int N,
                                                              int steps_per_split, const int* __restrict__ live,
                                                              int max_splits) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[2][2][MT * 16 * kPitch];  // [buffer][plane]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nB = live_rows(live, B);  // rows that hold leaves; B stays the slab pitch
  const int feat_steps = HW * 4;
  const int total_steps = feat_steps + (other_pad >> 5);
  const FcPlan plan = fc_plan(nB, MT * 16, gridDim.x, gridDim.y, steps_per_split, total_steps, max_splits);
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;  // (row block, split) of this workgroup under the plan
  const int split = wg / plan.row_blocks;
  if (split >= plan.splits) return;   // whole workgroup, before any barrier
  const int b0 = (wg - split * plan.row_blocks) * (MT * 16);
  if (b0 >= nB) return;
  const int s_begin = split * plan.steps;
  const int s_end = min(s_begin + plan.steps, total_steps);
  if (s_begin >= s_end) return;
  const int n_rows = (s_end - s_begin + 3) >> 2;                                          // chunks of 4 K-steps in this split
  const int n_feat = s_begin < feat_steps ? (min(s_end, feat_steps) - s_begin) >> 2 : 0;  // of which pixel rows of the body

  if (wave >= 4) {
    // ------------------------------------------------------------------ producers
    const int pt = tid - 256;  // chunk c = pt + i * 256 -> sample row c >> 4, 8-channel piece c & 15
    float gg[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gg[j] = LN == 2 ? g[(pt & 15) * 8 + j] : 1.f;
      bb[j] = LN == 2 ? bt[(pt & 15) * 8 + j] : 0.f;
    }
    // Sum over the 16 lanes that share (lane >> 4): quad permutes, half-row mirror, row mirror - all DPP, no LDS round trip.
    auto sum16 = [](float x) {
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
      return x;
    };
    // this thread's four (sample, 8-channel piece) sources at pixel row 0; samples past the live count read the last one
    const float* src[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int c = pt + i * 256, r = c >> 4, cc = c & 15;
      src[i] = rows + (size_t)min(b0 + r, nB - 1) * HW * 128 + cc * 8;
    }
    const int p0 = s_begin >> 2, p_last = max(p0 + n_feat - 1, 0);
    // first 128-feature chunk of the "other" features in this split (a split may start behind the body rows)
    const int k0 = max(s_begin - feat_steps, 0) >> 2;
    // body pixel row f of this split into registers; rows past the split's last one re-read it (loaded, never staged)
    auto issue = [&](int f, float4 (&v)[MT][2]) {
      const size_t off = (size_t)min(p0 + f, p_last) * 128;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float4* p = reinterpret_cast<const float4*>(src[i] + off);
        v[i][0] = p[0];
        v[i][1] = p[1];
      }
    };
    auto put = [&](const float (&f)[8], int i, int buf) {
      const int c = pt + i * 256, r = c >> 4, cc = c & 15;
      short8 xh, xl;
      split_op8(f, xh, xl);
      *reinterpret_cast<short8*>(xs[buf][0] + r * kPitch + cc * 16) = xh;
      *reinterpret_cast<short8*>(xs[buf][1] + r * kPitch + cc * 16) = xl;
    };
    // registers -> (LayerNorm) -> the two 16-bit planes of tile buffer `buf`
    auto stage = [&](const float4 (&v)[MT][2], int buf) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float f[8] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w, v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w};
        if (LN != 0) {
          float s = 0.f, ss = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) s += f[j];
          const float mean = sum16(s) * (1.0f / 128.0f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f[j] -= mean;
            ss += f[j] * f[j];
          }
          const float rstd = __builtin_amdgcn_rsqf(sum16(ss) * (1.0f / 128.0f) + eps);  // the argument is >= eps: no denormal case
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = LN == 2 ? f[j] * rstd * gg[j] + bb[j] : f[j] * rstd;
        }
        put(f, i, buf);
      }
    };
    // chunk k of the "other" features (zero padded to other_pad; steps past the split's end become zero columns)
    auto stage_other = [&](int k, int buf) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int c = pt + i * 256, r = c >> 4, cc = c & 15;
        const int d0 = k * 128 + cc * 8;
        const float* orow = other + (size_t)min(b0 + r, nB - 1) * other_dim;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = d0 + j < other_dim ? orow[d0 + j] : 0.f;
        put(f, i, buf);
      }
    };
    float4 v[kFcDepth][MT][2];
#pragma unroll
    for (int d = 0; d < kFcDepth; ++d) issue(d, v[d]);
    // chunk i is staged during iteration i - 1 (chunk 0 before the loop); a body row's register slot is then refilled
    // with the row kFcDepth further on
    if (n_feat > 0) {
      stage(v[0], 0);
      issue(kFcDepth, v[0]);
    } else {
      stage_other(k0, 0);
    }
    __syncthreads();
    int i = 0;  // the chunk the consumers multiply now
    for (; i + kFcDepth < n_feat; i += kFcDepth) {  // kFcDepth iterations with static register slots, no branches inside
#pragma unroll
      for (int d = 0; d < kFcDepth; ++d) {
        constexpr int kD = kFcDepth;
        stage(v[(d + 1) % kD], (i + d + 1) & 1);
        issue(i + d + 1 + kD, v[(d + 1) % kD]);
        __syncthreads();
      }
    }
    // remaining iterations: fewer than kFcDepth body rows, then the chunks of "other" features
#pragma unroll
    for (int d = 0; d < kFcDepth; ++d) {
      if (i + d + 1 < n_feat) {
        stage(v[(d + 1) % kFcDepth], (i + d + 1) & 1);
        __syncthreads();
      }
    }
    i = max(n_feat - 1, 0);
    for (; i < n_rows; ++i) {
      if (i + 1 < n_rows) stage_other(k0 + i + 1 - n_feat, (i + 1) & 1);
      __syncthreads();
    }
    return;
  }

  // -------------------------------------------------------------------- consumers
  const int q4 = lane >> 4, l15 = lane & 15;
  const bool wave_active = wave * 32 < N;
  float4v acc[MT][2];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    acc[t][0] = (float4v){0.f, 0.f, 0.f, 0.f};
    acc[t][1] = (float4v){0.f, 0.f, 0.f, 0.f};
  }
  const int wv = wave_active ? wave : 0;  // idle waves read valid fragments and drop the result
  const short8* wph = reinterpret_cast<const short8*>(Wp) + (size_t)wv * 128 + lane;
  const short8* wpl = reinterpret_cast<const short8*>(Wp + lo_off) + (size_t)wv * 128 + lane;
  short8 wh[4][2], wl[4][2];
  // K-steps past the split's end are staged as zero columns; their weights are the last step's again (finite)
  auto load_wslice = [&](int s0, int ks) {
    const size_t step = (size_t)min(s0 + ks, s_end - 1) * 512;
    wh[ks][0] = wph[step];
    wh[ks][1] = wph[step + 64];
    wl[ks][0] = wpl[step];
    wl[ks][1] = wpl[step + 64];
  };
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) load_wslice(s_begin, ks);
  __syncthreads();  // chunk 0 is staged
  for (int i = 0; i < n_rows; ++i) {
    const int s0 = s_begin + 4 * i;
    const unsigned char* xh_base = xs[i & 1][0] + l15 * kPitch + q4 * 16;
    const unsigned char* xl_base = xs[i & 1][1] + l15 * kPitch + q4 * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const short8 xh = *reinterpret_cast<const short8*>(xh_base + t * 16 * kPitch + ks * 64);
        const short8 xl = *reinterpret_cast<const short8*>(xl_base + t * 16 * kPitch + ks * 64);
        acc[t][0] = mma_split3(wh[ks][0], wl[ks][0], xh, xl, acc[t][0]);
        acc[t][1] = mma_split3(wh[ks][1], wl[ks][1], xh, xl, acc[t][1]);
      }
      load_wslice(s0 + 4, ks);  // the slice just used takes the next chunk's weights
    }
    __syncthreads();  // chunk i + 1 is staged; every consumer is past its reads of this buffer
  }
  if (!wave_active) return;
  float* slab = slabs + (size_t)split * B * 128;
  const int ch0 = wave * 32 + q4 * 8;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int bi = b0 + t * 16 + l15;
    if (bi >= nB) continue;
    float* op = slab + (size_t)bi * 128 + ch0;
    *reinterpret_cast<float4*>(op) = make_float4(acc[t][0][0], acc[t][0][1], acc[t][0][2], acc[t][0][3]);
    *reinterpret_cast<float4*>(op + 4) = make_float4(acc[t][1][0], acc[t][1][1], acc[t][1][2], acc[t][1][3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct SplitHeadsArgs {
  const float* slabs;          // [SK][B][128] split-K partial sums of the shared FC layer
  const float* fc_bias;        // [128]
  const unsigned short* wp1;   // packed [hi | lo] fragments: policy hidden 128 -> 128
  const float* bp1;
  const unsigned short* wp2;   // policy logits 128 -> A
  const float* bp2;
  const unsigned short* wv1;   // value hidden 128 -> 128
  const float* bv1;
  const unsigned short* wv2;   // value logits 128 -> atoms (one block)
  const float* bv2;
  long long lo_p1, lo_p2, lo_v1, lo_v2;
  float* logits;               // [B][logit_stride]
  float* vlogits;              // [B][value_stride]
  float* values;               // [B]
  int B, SK, act, nbp, a_store, logit_stride, atoms, atoms_store, value_stride;
  float vmin, dz;
  const int* live;             // device row count (nullptr: B rows), see common.hpp; B stays the slab pitch
  // the FC launch whose slabs these are (fc_plan): with fc_max_splits > SK the number of slabs follows the live count
  int fc_rows_per_block, fc_row_blocks, fc_steps_per_split, fc_total_steps, fc_max_splits;
};

namespace {
constexpr int kHeadRowsS = 16;
constexpr int kHeadPitchS = 288;

__device__ __forceinline__ float4v act4h(float4v v, int act) {
  if (act == kActReLU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
  } else if (act != kActNone) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], act);
  }
  return v;
}

// The sixteen (hi, lo) weight fragments of this lane for one 128-deep, 32-channel-per-wave layer block, and the product
// with the LDS tile rows of this lane.  Separate, so that the loads of the next layer's weights can be issued before the
// current layer is multiplied: the kernel is a chain of five small layers, each of which would otherwise start with a
// round trip to L2.
struct W128 {
  short8 h[4][2], l[4][2];
};
__device__ __forceinline__ void load_w128(W128& f, const unsigned short* __restrict__ w, long long lo_off, size_t frag_index) {
  const short8* wph = reinterpret_cast<const short8*>(w) + frag_index;
  const short8* wpl = reinterpret_cast<const short8*>(w + lo_off) + frag_index;
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    f.h[kc][0] = wph[kc * 512];
    f.h[kc][1] = wph[kc * 512 + 64];
    f.l[kc][0] = wpl[kc * 512];
    f.l[kc][1] = wpl[kc * 512 + 64];
  }
}
__device__ __forceinline__ void mma_w128(const W128& f, const unsigned char* __restrict__ tile_h, const unsigned char* __restrict__ tile_l,
                                         float4v (&acc)[2]) {
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    const short8 xh = *reinterpret_cast<const short8*>(tile_h + kc * 64);
    const short8 xl = *reinterpret_cast<const short8*>(tile_l + kc * 64);
    acc[0] = mma_split3(f.h[kc][0], f.l[kc][0], xh, xl, acc[0]);
    acc[1] = mma_split3(f.h[kc][1], f.l[kc][1], xh, xl, acc[1]);
  }
}
}  // namespace

__global__ __launch_bounds__(256) void heads_split_kernel(SplitHeadsArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char s_tile[2][kHeadRowsS * kHeadPitchS];
  __shared__ __attribute__((aligned(16))) unsigned char p_tile[2][kHeadRowsS * kHeadPitchS];
  __shared__ __attribute__((aligned(16))) unsigned char v_tile[2][kHeadRowsS * kHeadPitchS];
  __shared__ float vl[kHeadRowsS][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int b0 = blockIdx.x * kHeadRowsS;
  const int nB = live_rows(a.live, a.B);
  if (b0 >= nB) return;  // whole workgroup, before any barrier

  // weights of both hidden layers: requested now, used after the slab sums
  const size_t frag = (size_t)wave * 128 + lane;
  W128 w_p1, w_v1;
  load_w128(w_p1, a.wp1, a.lo_p1, frag);
  load_w128(w_v1, a.wv1, a.lo_v1, frag);

  // ---- shared FC: sum the split-K slabs, bias, activation (fp32) -> s_tile planes (thread = sample, 8 channels) ----
  {
    const int r = tid >> 4, n0 = (tid & 15) * 8;
    const int b = b0 + r;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (b < nB) {
      const float4* bp = reinterpret_cast<const float4*>(a.fc_bias + n0);
      const float4 c0 = bp[0], c1 = bp[1];
      v[0] = c0.x; v[1] = c0.y; v[2] = c0.z; v[3] = c0.w;
      v[4] = c1.x; v[5] = c1.y; v[6] = c1.z; v[7] = c1.w;
      const int n_slabs = fc_plan(nB, a.fc_rows_per_block, a.fc_row_blocks, a.SK, a.fc_steps_per_split, a.fc_total_steps,
                                  a.fc_max_splits).splits;
      const size_t slab_pitch = (size_t)a.B * 128;
      const float* sp0 = a.slabs + (size_t)b * 128 + n0;
      int s = 0;
      for (; s + 3 < n_slabs; s += 4) {  // four slabs per round trip (a thin wave has up to sixteen)
        float4 p[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4* sp = reinterpret_cast<const float4*>(sp0 + (size_t)(s + i) * slab_pitch);
          p[i][0] = sp[0];
          p[i][1] = sp[1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[0] += p[i][0].x; v[1] += p[i][0].y; v[2] += p[i][0].z; v[3] += p[i][0].w;
          v[4] += p[i][1].x; v[5] += p[i][1].y; v[6] += p[i][1].z; v[7] += p[i][1].w;
        }
      }
      for (; s < n_slabs; ++s) {
        const float4* sp = reinterpret_cast<const float4*>(sp0 + (size_t)s * slab_pitch);
        const float4 p0 = sp[0], p1 = sp[1];
        v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
        v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
      }
      const float4v lo = act4h((float4v){v[0], v[1], v[2], v[3]}, a.act);
      const float4v hi = act4h((float4v){v[4], v[5], v[6], v[7]}, a.act);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = lo[j];
        v[4 + j] = hi[j];
      }
    }
    short8 xh, xl;
    split_op8(v, xh, xl);
    *reinterpret_cast<short8*>(s_tile[0] + r * kHeadPitchS + n0 * 2) = xh;
    *reinterpret_cast<short8*>(s_tile[1] + r * kHeadPitchS + n0 * 2) = xl;
  }
  __syncthreads();

  // ---- hidden layers of both heads: wave = 32 output channels, lane ends with 8 consecutive channels of sample l15 ----
  const int lane_off = l15 * kHeadPitchS + q4 * 16;
  const int ch = wave * 32 + q4 * 8;
  // weights of the value logits and of the first policy block: requested before the hidden layers are multiplied
  W128 w_v2, w_p2;
  const bool value_wave = wave * 32 < a.atoms_store;
  load_w128(w_v2, a.wv2, a.lo_v2, value_wave ? frag : (size_t)lane);
  load_w128(w_p2, a.wp2, a.lo_p2, frag);
#pragma unroll
  for (int head = 0; head < 2; ++head) {
    float4v acc[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
    mma_w128(head == 0 ? w_p1 : w_v1, s_tile[0] + lane_off, s_tile[1] + lane_off, acc);
    const float* bias = (head == 0 ? a.bp1 : a.bv1) + ch;
    float4v lo = acc[0], hi = acc[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo[j] += bias[j];
      hi[j] += bias[4 + j];
    }
    lo = act4h(lo, a.act);
    hi = act4h(hi, a.act);
    const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    short8 xh, xl;
    split_op8(f, xh, xl);
    unsigned char* dh = (head == 0 ? p_tile[0] : v_tile[0]) + l15 * kHeadPitchS + ch * 2;
    unsigned char* dl = (head == 0 ? p_tile[1] : v_tile[1]) + l15 * kHeadPitchS + ch * 2;
    *reinterpret_cast<short8*>(dh) = xh;
    *reinterpret_cast<short8*>(dl) = xl;
  }
  __syncthreads();

  // ---- value logits (one 128-channel block; waves beyond the atom count idle) -> LDS + memory ----
  const int b = b0 + l15;
  if (value_wave) {
    float4v acc[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
    mma_w128(w_v2, v_tile[0] + lane_off, v_tile[1] + lane_off, acc);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = ch + n * 4 + j;
        const float v = acc[n][j] + a.bv2[c];
        vl[l15][c] = v;
        if (b < nB && c < a.atoms_store) a.vlogits[(size_t)b * a.value_stride + c] = v;
      }
    }
  }
  // ---- policy logits: NBP blocks of 128 channels ----
  for (int nb = 0; nb < a.nbp; ++nb) {
    if (nb * 128 + wave * 32 >= a.a_store) break;  // wave-uniform
    float4v acc[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
    const W128 w_now = w_p2;
    if (nb + 1 < a.nbp) load_w128(w_p2, a.wp2, a.lo_p2, ((size_t)(nb + 1) * 16 + wave) * 128 + lane);  // next block's weights
    mma_w128(w_now, p_tile[0] + lane_off, p_tile[1] + lane_off, acc);
    const int c0 = nb * 128 + ch;
    if (b < nB && c0 < a.a_store) {
      const float4* bp = reinterpret_cast<const float4*>(a.bp2 + c0);
      const float4 bl = bp[0], bh = bp[1];
      float4* out = reinterpret_cast<float4*>(a.logits + (size_t)b * a.logit_stride + c0);
      out[0] = make_float4(acc[0][0] + bl.x, acc[0][1] + bl.y, acc[0][2] + bl.z, acc[0][3] + bl.w);
      out[1] = make_float4(acc[1][0] + bh.x, acc[1][1] + bh.y, acc[1][2] + bh.z, acc[1][3] + bh.w);
    }
  }
  __syncthreads();

  // ---- C51 expectation: 16 lanes per sample, atoms strided over them ----
  {
    const int r = tid >> 4, l = tid & 15;
    float m = -1e30f;
    for (int i = l; i < a.atoms; i += 16) m = fmaxf(m, vl[r][i]);
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float s = 0.f, e = 0.f;
    for (int i = l; i < a.atoms; i += 16) {
      const float p = expf(vl[r][i] - m);
      s += p;
      e += p * (a.vmin + (float)i * a.dz);
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      s += __shfl_xor(s, off, 64);
      e += __shfl_xor(e, off, 64);
    }
    if (l == 0 && b0 + r < nB) a.values[b0 + r] = e / s;
  }
}

extern "C" {

// Operand format the split kernels of this library were built for: 0 = bf16 (hi, lo) pairs, 1 = fp16 pairs.  The host
// packs weight fragments accordingly.
int atri_split_operand_format() { return kSplitF16 ? 1 : 0; }

// max_splits > SK: slabs must hold max_splits x B x 128 floats; the split count then follows the live row count (fc_plan)
// and atri_split_heads_dyn has to be given the same numbers.
int atri_split_fc_rows_dyn(const float* rows, const float* other, int other_dim, const float* g, const float* bt, float eps,
                           const void* Wp, long long lo_off, float* slabs, int B, int HW, int D, int other_pad, int N, int SK,
                           int steps_per_split, int max_splits, hipStream_t stream) {
  if (B <= 0) return 0;
  if (D != 128 || (other_pad & 31) != 0 || other_dim < 0 || other_dim > other_pad || N > 128 || (N & 31) != 0) return -10;
  if ((steps_per_split & 3) != 0 || steps_per_split <= 0) return -11;
  if ((long long)SK * steps_per_split < (long long)HW * 4 + (other_pad >> 5)) return -12;
  // 64 samples x one K split per 8-wave workgroup (one per CU).  g == nullptr with eps > 0: LayerNorm without an affine part
  // (the host folded it into Wp and the bias); eps == 0: no LayerNorm.
  const dim3 grid((B + 63) / 64, SK);
#define ATRI_FC_LAUNCH(LN)                                                                                                \
  hipLaunchKernelGGL((fc_rows_split_kernel<4, LN, 4>), grid, dim3(512), 0, stream, rows, other, other_dim, g, bt, eps,    \
                     (const unsigned short*)Wp, lo_off, slabs, B, HW, other_pad, N, steps_per_split, live_count_ptr(), max_splits)
  if (g != nullptr) {
    ATRI_FC_LAUNCH(2);
  } else if (eps > 0.f) {
    ATRI_FC_LAUNCH(1);
  } else {
    ATRI_FC_LAUNCH(0);
  }
#undef ATRI_FC_LAUNCH
  return hip_err(hipGetLastError());
}

int atri_split_fc_rows(const float* rows, const float* other, int other_dim, const float* g, const float* bt, float eps,
                       const void* Wp, long long lo_off, float* slabs, int B, int HW, int D, int other_pad, int N, int SK,
                       int steps_per_split, hipStream_t stream) {
  return atri_split_fc_rows_dyn(rows, other, other_dim, g, bt, eps, Wp, lo_off, slabs, B, HW, D, other_pad, N, SK,
                                steps_per_split, 0, stream);
}

// The plan atri_split_fc_rows_dyn (64 rows per workgroup) follows for n_live rows of a batch of B:
// out = {row blocks, splits, steps per split}.
int atri_split_fc_plan(int n_live, int B, int SK, int steps_per_split, int total_steps, int max_splits, int* out) {
  const FcPlan p = fc_plan(n_live < B ? n_live : B, 64, (B + 63) / 64, SK, steps_per_split, total_steps, max_splits);
  out[0] = p.row_blocks;
  out[1] = p.splits;
  out[2] = p.steps;
  return 0;
}

int atri_split_heads_dyn(const float* slabs, int SK, int fc_steps_per_split, int fc_total_steps, int fc_max_splits,
                         const float* fc_bias, const void* wp1, long long lo_p1, const float* bp1,
                         const void* wp2, long long lo_p2, const float* bp2, int nbp, int a_store, const void* wv1, long long lo_v1,
                         const float* bv1, const void* wv2, long long lo_v2, const float* bv2, int atoms, int atoms_store,
                         float* logits, int logit_stride, float* vlogits, int value_stride, float* values, float vmin, float dz,
                         int B, int act, hipStream_t stream) {
  if (B <= 0) return 0;
  if (SK < 1 || nbp < 1 || a_store < 8 || (a_store & 7) || a_store > nbp * 128 || a_store > logit_stride) return -20;
  if (atoms < 1 || atoms > atoms_store || atoms_store > 128 || (atoms_store & 7) || atoms_store > value_stride) return -21;
  if (logit_stride & 3) return -22;
  SplitHeadsArgs a{};
  a.slabs = slabs;
  a.fc_bias = fc_bias;
  a.wp1 = (const unsigned short*)wp1; a.lo_p1 = lo_p1; a.bp1 = bp1;
  a.wp2 = (const unsigned short*)wp2; a.lo_p2 = lo_p2; a.bp2 = bp2;
  a.wv1 = (const unsigned short*)wv1; a.lo_v1 = lo_v1; a.bv1 = bv1;
  a.wv2 = (const unsigned short*)wv2; a.lo_v2 = lo_v2; a.bv2 = bv2;
  a.logits = logits;
  a.vlogits = vlogits;
  a.values = values;
  a.B = B; a.SK = SK; a.act = act; a.nbp = nbp; a.a_store = a_store; a.logit_stride = logit_stride;
  a.atoms = atoms; a.atoms_store = atoms_store; a.value_stride = value_stride;
  a.vmin = vmin; a.dz = dz;
  a.live = live_count_ptr();
  a.fc_rows_per_block = 64;
  a.fc_row_blocks = (B + 63) / 64;
  a.fc_steps_per_split = fc_steps_per_split;
  a.fc_total_steps = fc_total_steps;
  a.fc_max_splits = fc_max_splits;
  hipLaunchKernelGGL(heads_split_kernel, dim3((B + kHeadRowsS - 1) / kHeadRowsS), dim3(256), 0, stream, a);
  return hip_err(hipGetLastError());
}

int atri_split_heads(const float* slabs, int SK, const float* fc_bias, const void* wp1, long long lo_p1, const float* bp1,
                     const void* wp2, long long lo_p2, const float* bp2, int nbp, int a_store, const void* wv1, long long lo_v1,
                     const float* bv1, const void* wv2, long long lo_v2, const float* bv2, int atoms, int atoms_store,
                     float* logits, int logit_stride, float* vlogits, int value_stride, float* values, float vmin, float dz,
                     int B, int act, hipStream_t stream) {
  return atri_split_heads_dyn(slabs, SK, 0, 0, 0, fc_bias, wp1, lo_p1, bp1, wp2, lo_p2, bp2, nbp, a_store, wv1, lo_v1, bv1, wv2,
                              lo_v2, bv2, atoms, atoms_store, logits, logit_stride, vlogits, value_stride, values, vmin, dz, B,
                              act, stream);
}

}  // extern "C"
