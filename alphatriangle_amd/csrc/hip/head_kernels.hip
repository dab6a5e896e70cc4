// Network tail in one launch: split-K FC reduction + bias + activation, policy head (128 -> 128 -> A logits),
// value head (128 -> 128 -> atoms) and the C51 expectation (alphatriangle/nn/model.py:232-266 heads,
// alphatriangle/nn/network.py:157-162 expectation), BatchNorm folded.
//
// These layers are a few hundred kFLOP per sample; as separate launches (reduction, four small GEMMs,
// expectation) they cost six kernel boundaries and ~55 us per MCTS wave of pure latency.  Here one workgroup
// takes 16 samples through the whole chain: activations stay in LDS, each wave owns 32 output channels of every
// layer and reads its pre-packed weight fragments straight from L2 (first-layer fragments are requested before
// the split-K slabs are reduced).
#include "common.hpp"

using namespace atrihip;

struct HeadsArgs {
  const float* slabs;          // [SK][B][128] split-K partial sums of the shared FC layer
  const float* fc_bias;        // [128]
  const unsigned short* wp1;   // packed fragments [1][4][wave 4][n 2][lane 64][8]: policy hidden 128 -> 128
  const float* bp1;            // [128]
  const unsigned short* wp2;   // packed fragments [NBP][4][4][2][64][8]: policy logits 128 -> A
  const float* bp2;            // [NBP*128]
  const unsigned short* wv1;   // value hidden 128 -> 128
  const float* bv1;
  const unsigned short* wv2;   // value logits 128 -> atoms (one block)
  const float* bv2;            // [128]
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
constexpr int kHeadRows = 16;     // samples per workgroup = one MFMA N tile
constexpr int kHeadPitch = 288;   // bytes per LDS row of a 128-wide bf16 activation tile (256 + 32: conflict-free b128 reads)

__device__ __forceinline__ float4v act4(float4v v, int act) {
  if (act == kActReLU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
  } else if (act != kActNone) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], act);
  }
  return v;
}

// acc[n] += W-fragments(step kc) x activation fragment(kc) for the 128-deep layer whose packed weights start at wp
__device__ __forceinline__ void gemm128(const short8* __restrict__ wp, const unsigned char* __restrict__ tile_lane,
                                        float4v (&acc)[2]) {
  short8 w[4][2];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    w[kc][0] = wp[kc * 512];
    w[kc][1] = wp[kc * 512 + 64];
  }
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    const short8 x = *reinterpret_cast<const short8*>(tile_lane + kc * 64);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[kc][0], x, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[kc][1], x, acc[1], 0, 0, 0);
  }
}
}  // namespace

__global__ __launch_bounds__(256) void heads_kernel(HeadsArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char s_tile[kHeadRows * kHeadPitch];
  __shared__ __attribute__((aligned(16))) unsigned char p_tile[kHeadRows * kHeadPitch];
  __shared__ __attribute__((aligned(16))) unsigned char v_tile[kHeadRows * kHeadPitch];
  __shared__ float vl[kHeadRows][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int b0 = blockIdx.x * kHeadRows;
  const int nB = live_rows(a.live, a.B);
  if (b0 >= nB) return;  // whole workgroup, before any barrier

  // first-layer weight fragments of both heads: requested now, consumed after the slab reduction
  const short8* wp1 = reinterpret_cast<const short8*>(a.wp1) + wave * 128 + lane;
  const short8* wv1 = reinterpret_cast<const short8*>(a.wv1) + wave * 128 + lane;
  short8 w1[2][4][2];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    w1[0][kc][0] = wp1[kc * 512];
    w1[0][kc][1] = wp1[kc * 512 + 64];
    w1[1][kc][0] = wv1[kc * 512];
    w1[1][kc][1] = wv1[kc * 512 + 64];
  }

  // ---- shared FC: sum the split-K slabs, bias, activation -> s_tile (thread = sample, 8 channels) ----
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
      for (int s = 0; s < n_slabs; ++s) {
        const float4* sp = reinterpret_cast<const float4*>(a.slabs + ((size_t)s * a.B + b) * 128 + n0);
        const float4 p0 = sp[0], p1 = sp[1];
        v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
        v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
      }
      const float4v lo = act4((float4v){v[0], v[1], v[2], v[3]}, a.act);
      const float4v hi = act4((float4v){v[4], v[5], v[6], v[7]}, a.act);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = lo[j];
        v[4 + j] = hi[j];
      }
    }
    *reinterpret_cast<short8*>(s_tile + r * kHeadPitch + n0 * 2) = pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  }
  __syncthreads();

  // ---- hidden layers of both heads: wave = 32 output channels, lane ends with 8 consecutive channels of sample l15 ----
  const int lane_off = l15 * kHeadPitch + q4 * 16;
  const int ch = wave * 32 + q4 * 8;
#pragma unroll
  for (int head = 0; head < 2; ++head) {
    float4v acc[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const short8 x = *reinterpret_cast<const short8*>(s_tile + lane_off + kc * 64);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[head][kc][0], x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[head][kc][1], x, acc[1], 0, 0, 0);
    }
    const float* bias = (head == 0 ? a.bp1 : a.bv1) + ch;
    float4v lo = acc[0], hi = acc[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      lo[j] += bias[j];
      hi[j] += bias[4 + j];
    }
    lo = act4(lo, a.act);
    hi = act4(hi, a.act);
    unsigned char* dst = (head == 0 ? p_tile : v_tile) + l15 * kHeadPitch + ch * 2;
    *reinterpret_cast<short8*>(dst) = pack_bf16x8(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
  }
  __syncthreads();

  // ---- value logits (one 128-channel block; waves beyond the atom count idle) -> LDS + memory ----
  const int b = b0 + l15;
  if (wave * 32 < a.atoms_store) {
    float4v acc[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
    gemm128(reinterpret_cast<const short8*>(a.wv2) + wave * 128 + lane, v_tile + lane_off, acc);
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
    gemm128(reinterpret_cast<const short8*>(a.wp2) + ((size_t)nb * 16 + wave) * 128 + lane, p_tile + lane_off, acc);
    const int c0 = nb * 128 + ch;
    if (b < nB && c0 < a.a_store) {  // a_store is a multiple of 8: a lane's 8 channels are all inside or all outside
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
      const float p = __expf(vl[r][i] - m);
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

// Tail of the network for B samples.  Every hidden width is 128; `a_store` / `atoms_store` are the stored (multiple
// of 8) widths of the two output layers, `nbp` the number of 128-channel weight blocks of the policy output layer.
// fc_*: the atri_fc_rows_splitk_dyn launch that filled the slabs (samples per workgroup, steps per split, total
// K-steps, max_splits); fc_max_splits <= SK: exactly SK slabs.
int atri_heads_dyn(const float* slabs, int SK, int fc_rows_per_block, int fc_steps_per_split, int fc_total_steps,
                   int fc_max_splits, const float* fc_bias, const void* wp1, const float* bp1, const void* wp2,
                   const float* bp2, int nbp, int a_store, const void* wv1, const float* bv1, const void* wv2,
                   const float* bv2, int atoms, int atoms_store, float* logits, int logit_stride, float* vlogits,
                   int value_stride, float* values, float vmin, float dz, int B, int act, hipStream_t stream) {
  if (B <= 0) return 0;
  if (fc_max_splits > SK && (fc_rows_per_block <= 0 || fc_steps_per_split <= 0 || fc_total_steps <= 0)) return -23;
  if (SK < 1 || nbp < 1 || a_store < 8 || (a_store & 7) || a_store > nbp * 128 || a_store > logit_stride) return -20;
  if (atoms < 1 || atoms > atoms_store || atoms_store > 128 || (atoms_store & 7) || atoms_store > value_stride) return -21;
  if (logit_stride & 3) return -22;  // float4 stores
  HeadsArgs a{};
  a.slabs = slabs;
  a.fc_bias = fc_bias;
  a.wp1 = (const unsigned short*)wp1;
  a.bp1 = bp1;
  a.wp2 = (const unsigned short*)wp2;
  a.bp2 = bp2;
  a.wv1 = (const unsigned short*)wv1;
  a.bv1 = bv1;
  a.wv2 = (const unsigned short*)wv2;
  a.bv2 = bv2;
  a.logits = logits;
  a.vlogits = vlogits;
  a.values = values;
  a.B = B;
  a.live = live_count_ptr();
  a.SK = SK;
  a.act = act;
  a.nbp = nbp;
  a.a_store = a_store;
  a.logit_stride = logit_stride;
  a.atoms = atoms;
  a.atoms_store = atoms_store;
  a.value_stride = value_stride;
  a.vmin = vmin;
  a.dz = dz;
  a.fc_rows_per_block = fc_rows_per_block > 0 ? fc_rows_per_block : 64;
  a.fc_row_blocks = (B + a.fc_rows_per_block - 1) / a.fc_rows_per_block;
  a.fc_steps_per_split = fc_steps_per_split;
  a.fc_total_steps = fc_total_steps;
  a.fc_max_splits = fc_max_splits;
  hipLaunchKernelGGL(heads_kernel, dim3((B + kHeadRows - 1) / kHeadRows), dim3(256), 0, stream, a);
  return hip_err(hipGetLastError());
}

int atri_heads(const float* slabs, int SK, const float* fc_bias, const void* wp1, const float* bp1, const void* wp2,
               const float* bp2, int nbp, int a_store, const void* wv1, const float* bv1, const void* wv2,
               const float* bv2, int atoms, int atoms_store, float* logits, int logit_stride, float* vlogits,
               int value_stride, float* values, float vmin, float dz, int B, int act, hipStream_t stream) {
  return atri_heads_dyn(slabs, SK, 0, 0, 0, 0, fc_bias, wp1, bp1, wp2, bp2, nbp, a_store, wv1, bv1, wv2, bv2, atoms,
                        atoms_store, logits, logit_stride, vlogits, value_stride, values, vmin, dz, B, act, stream);
}

}  // extern "C"
