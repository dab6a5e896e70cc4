// Hand-written gfx950 inference kernels of the policy / value network.
//
// Network being evaluated: alphatriangle/nn/model.py:109-297 (conv stem, residual
// blocks, pre-LN Transformer encoder, shared FC, policy head, C51 value head) with
// BatchNorm folded into the preceding conv / linear weights and bf16 activations.
//
// Data layout: activations are NHWC bf16, i.e. a row-major [rows = B*H*W, C] matrix,
// so every conv / linear layer is a GEMM over rows.
//
// tile_gemm_kernel - the workhorse.  One workgroup keeps the complete input tile
// of up to 256 rows (two 8x15 samples) resident in LDS and produces all taps of a
// 3x3 convolution (or a 1x1 / linear layer) from it: no im2col traffic, the only
// streamed operand is the weight panel, which is pre-packed in MFMA fragment order
// and read straight from L2 into registers (each weight byte once per workgroup).
// MFMA shape v_mfma_f32_16x16x32_bf16 is used with the weights as the A operand and
// the activations as the B operand, so a lane ends up holding 8 consecutive output
// channels of one pixel and the epilogue (bias, residual, activation, positional
// table, bf16 pack) stores 16 bytes per lane.  240 rows = 15 M-tiles of 16, so the
// default grid wastes no MFMA rows.  64-lane waves: the four waves split the 128
// output channels of a column block (32 each).
#include "common.hpp"

using namespace atrihip;

struct TileGemmArgs {
  const unsigned short* in;        // [rows_total, Cin] bf16
  const unsigned short* w;         // packed fragments, see pack_gemm_weights in ops/inference.py
  const float* bias;               // [n_blocks*128]
  const unsigned short* residual;  // [rows_total, Cout] bf16 or nullptr
  const float* post_add;           // [HW, Cout] f32 or nullptr (added after the activation)
  const float* ln_g;               // [Cin] LayerNorm prologue (nullptr = none)
  const float* ln_b;
  void* out;                       // bf16 or f32, row stride out_stride elements
  int rows_total, rows_per_wg, HW, H, W, Cin, Cout, taps, act, out_f32, out_stride, n_blocks, log2_cpr;
  float ln_eps;
};

// Non-ReLU activations of eight values; kept out of line so the unrolled epilogue stays small.
// By value on purpose: an address-taken float[8] would live in scratch memory on every path.
__device__ __noinline__ float4v act8_slow(float4v v, int act) {
  for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], act);
  return v;
}

// DBG != 0 instantiations are roofline probes (bench/conv_probe.py): they drop one ingredient of the main
// loop so its cost can be read off the timing difference; their results are numerically meaningless.
// DBG is a bit mask: 1 no LDS fragment reads, 2 no weight-fragment loads, 4 no main loop, 8 no tap address math,
// 16 no input staging, 32 no output stores.
template <int MT, int DBG = 0>
__global__ __launch_bounds__(256, 2) void tile_gemm_kernel(TileGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int row0 = blockIdx.x * a.rows_per_wg;
  const int rows = min(a.rows_per_wg, a.rows_total - row0);
  const int stride = a.Cin * 2 + 16;  // bytes per LDS row (16-byte pad breaks the 256-byte bank period)
  const int zoff = MT * 16 * stride;  // one all-zero row serves out-of-image taps and padding rows

  // ---- stage the input tile (coalesced 16-byte chunks) ----
  {
    // rows of the tile are contiguous in memory: chunk c is 16 bytes at in + row0*Cin + 8c.
    // Eight loads are kept in flight per thread before the first LDS write (HBM latency paid once per batch).
    const int cmask = (1 << a.log2_cpr) - 1;  // 16-byte chunks per row - 1
    const int nchunks = rows << a.log2_cpr;
    const uint4* src = reinterpret_cast<const uint4*>(a.in + (size_t)row0 * a.Cin);
    for (int c0 = tid; c0 < ((DBG & 16) ? 0 : nchunks); c0 += 256 * 8) {
      uint4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + i * 256;
        v[i] = make_uint4(0, 0, 0, 0);  // unconditional definition keeps the batch in registers
        if (c < nchunks) v[i] = src[c];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + i * 256;
        if (c < nchunks) *reinterpret_cast<uint4*>(smem + (c >> a.log2_cpr) * stride + (c & cmask) * 16) = v[i];
      }
    }
    for (int c = tid; c < (stride >> 4); c += 256) *reinterpret_cast<uint4*>(smem + zoff + c * 16) = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();

  // ---- optional LayerNorm over Cin, in place (pre-LN Transformer blocks) ----
  if (a.ln_g != nullptr) {
    if (tid < rows) {
      unsigned char* rowp = smem + tid * stride;
      float s = 0.f, ss = 0.f;
      for (int c = 0; c < (a.Cin >> 3); ++c) {
        short8 v = *reinterpret_cast<const short8*>(rowp + c * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float x = bf16_to_float((unsigned short)v[j]);
          s += x;
          ss += x * x;
        }
      }
      float mean = s / (float)a.Cin;
      float var = fmaxf(ss / (float)a.Cin - mean * mean, 0.f);
      float rstd = rsqrtf(var + a.ln_eps);
      for (int c = 0; c < (a.Cin >> 3); ++c) {
        short8 v = *reinterpret_cast<const short8*>(rowp + c * 16);
        short8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float x = (bf16_to_float((unsigned short)v[j]) - mean) * rstd;
          o[j] = (short)float_to_bf16(x * a.ln_g[c * 8 + j] + a.ln_b[c * 8 + j]);
        }
        *reinterpret_cast<short8*>(rowp + c * 16) = o;
      }
    }
    __syncthreads();
  }

  // ---- per-lane tile bookkeeping: LDS offset of "its" pixel + 9-bit tap validity ----
  int packed[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int r = t * 16 + l15;
    int mask = 0;
    if (r < rows) {
      if (a.taps == 9) {
        int p = r % a.HW;
        int y = p / a.W, x = p - y * a.W;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
          if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) mask |= 1 << tap;
        }
      } else {
        mask = 1;
      }
    }
    packed[t] = (r * stride + q4 * 16) | (mask << 18);
  }

  const int KC = a.Cin >> 5;
  const int steps = a.taps * KC;
  for (int nb = 0; nb < a.n_blocks; ++nb) {
    const int ch_wave = nb * 128 + wave * 32;
    if (ch_wave >= a.Cout) continue;  // wave-uniform
    float4v acc[MT][2];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      acc[t][0] = (float4v){0.f, 0.f, 0.f, 0.f};
      acc[t][1] = (float4v){0.f, 0.f, 0.f, 0.f};
    }
    // fragment stream of this wave: [(nb*steps + s)][wave][n][lane] in short8 units
    const short8* wp = reinterpret_cast<const short8*>(a.w) + ((size_t)nb * steps * 4 + wave) * 128 + lane;
    short8 b0 = wp[0], b1 = wp[64];
    int s = 0;
    for (int tap = 0; tap < ((DBG & 4) ? 0 : a.taps); ++tap) {
      const int toff = (a.taps == 9) ? ((tap / 3 - 1) * a.W + (tap % 3 - 1)) * stride : 0;
      const int tapbit = 18 + tap;
      const int zlane = zoff + q4 * 16;
      for (int kc = 0; kc < KC; ++kc, ++s) {
        // prefetch the next step's weight fragments (the packed buffer carries one spare step)
        const short8 nb0 = (DBG & 2) ? b0 : wp[(size_t)(s + 1) * 512];
        const short8 nb1 = (DBG & 2) ? b1 : wp[(size_t)(s + 1) * 512 + 64];
        const int koff = kc * 64;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          const int ad = (DBG & 8) ? (packed[t] & 0x3FFFF) : ((packed[t] >> tapbit) & 1) ? (packed[t] & 0x3FFFF) + toff : zlane;
          const short8 av = (DBG & 1) ? b1 : *reinterpret_cast<const short8*>(smem + ad + koff);
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, av, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, av, acc[t][1], 0, 0, 0);
          if ((t % 5) == 4) __builtin_amdgcn_sched_barrier(0);  // bound the LDS reads in flight (VGPR budget)
        }
        b0 = nb0;
        b1 = nb1;
      }
    }

    // ---- epilogue: lane owns channels ch0..ch0+7 of pixel (t*16 + l15) ----
    const int ch0 = ch_wave + q4 * 8;
    if (ch0 < a.Cout) {
      float bias[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) bias[i] = a.bias[ch0 + i];
      constexpr int EG = (MT % 5 == 0) ? 5 : 4;  // tiles per epilogue group: residual loads of a group fly together
#pragma unroll
      for (int t0 = 0; t0 < MT; t0 += EG) {
        short8 rv[EG];
        if (a.residual != nullptr) {
#pragma unroll
          for (int u = 0; u < EG; ++u) {
            int r = (t0 + u) * 16 + l15;
            asm volatile("" : "+v"(r));  // keep the per-row address math out of the MFMA loop's live range
            rv[u] = (short8){0, 0, 0, 0, 0, 0, 0, 0};
            if (r < rows) rv[u] = *reinterpret_cast<const short8*>(a.residual + (size_t)(row0 + r) * a.Cout + ch0);
          }
        }
#pragma unroll
        for (int u = 0; u < EG; ++u) {
          const int t = t0 + u;
          int r = t * 16 + l15;
          asm volatile("" : "+v"(r));
          if (r >= rows) continue;
          if ((DBG & 32) && acc[t][0][0] != 12345.f) continue;
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = acc[t][0][j] + bias[j];
            v[4 + j] = acc[t][1][j] + bias[4 + j];
          }
          const size_t grow = (size_t)(row0 + r);
          if (a.residual != nullptr) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += bf16_to_float((unsigned short)rv[u][i]);
          }
          if (a.act == kActReLU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
          } else if (a.act != kActNone) {
            const float4v lo = act8_slow((float4v){v[0], v[1], v[2], v[3]}, a.act);
            const float4v hi = act8_slow((float4v){v[4], v[5], v[6], v[7]}, a.act);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j] = lo[j];
              v[4 + j] = hi[j];
            }
          }
          if (a.post_add != nullptr) {
            const float* pa = a.post_add + (size_t)(r % a.HW) * a.Cout + ch0;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += pa[i];
          }
          if (a.out_f32) {
            float* op = reinterpret_cast<float*>(a.out) + grow * a.out_stride + ch0;
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
            *reinterpret_cast<short8*>(reinterpret_cast<unsigned short*>(a.out) + grow * a.out_stride + ch0) =
                pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// First conv layer: Cin = 1, direct 3x3 on the f32 grid -> bf16 NHWC.  Memory bound.
// One thread per (row, 8 output channels).
__global__ void conv_first_kernel(const float* __restrict__ grid, const float* __restrict__ w /*[9][Cout]*/,
                                  const float* __restrict__ bias, unsigned short* __restrict__ out, int rows_total,
                                  int HW, int H, int W, int Cout, int act) {
  const int groups = Cout >> 3;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)rows_total * groups) return;
  const int row = (int)(idx / groups), g = (int)(idx - (long long)row * groups);
  const int p = row % HW, y = p / W, x = p - y * W;
  const float* gs = grid + (size_t)(row - p);
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = bias[g * 8 + i];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    float gv = gs[yy * W + xx];
    const float* wp = w + tap * Cout + g * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += gv * wp[i];
  }
  short8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (short)float_to_bf16(apply_act(v[i], act));
  *reinterpret_cast<short8*>(out + (size_t)row * Cout + g * 8) = o;
}

// ---------------------------------------------------------------------------
// Multi-head self-attention over one sample's tokens, head_dim = 32, no mask.
// One wave per (sample, head).  S^T = K Q^T is computed with the KEY on the MFMA
// row, so the softmax row reduction is in-lane plus two shuffles and the
// probability tile feeds the P.V product as the B operand straight from registers
// (only V goes through LDS, to gather its column-major fragments).
// qkv: [B, HW, 3*D] bf16 (Q pre-scaled by log2(e)/sqrt(32): the softmax uses exp2), out: [B, HW, D] bf16.
template <int NKT>
__global__ __launch_bounds__(256) void attention_kernel(const unsigned short* __restrict__ qkv,
                                                        unsigned short* __restrict__ out, int B, int HW, int D,
                                                        int heads) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int VSTRIDE = 80;  // bytes per staged V row (64 + 16 pad)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.x;
  unsigned char* vl = smem + wave * (NKT * 16 * VSTRIDE);
  const size_t rs = (size_t)3 * D;  // row stride in elements
  const unsigned short* base = qkv + (size_t)b * HW * rs;

  for (int head = wave; head < heads; head += 4) {
    // stage V[token][32] of this head
    for (int c = lane; c < NKT * 16 * 4; c += 64) {
      int tok = c >> 2, ch = c & 3;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (tok < HW) v = *reinterpret_cast<const uint4*>(base + tok * rs + 2 * D + head * 32 + ch * 8);
      *reinterpret_cast<uint4*>(vl + tok * VSTRIDE + ch * 16) = v;
    }
    // K fragments (A operand): K[token 16kt + l15][dims 8q4 .. 8q4+7]
    short8 kf[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      int tok = kt * 16 + l15;
      kf[kt] = (short8){0, 0, 0, 0, 0, 0, 0, 0};
      if (tok < HW) kf[kt] = *reinterpret_cast<const short8*>(base + tok * rs + D + head * 32 + q4 * 8);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // V^T fragments (A operand of P.V): element e of lane group q4 in key block kb is
    // key 32kb + (e<4 ? 4q4+e : 16+4q4+e-4), matching the register order of the S^T tiles.
    short8 vf[NKT / 2][2];
#pragma unroll
    for (int kb = 0; kb < NKT / 2; ++kb)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        short8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          int key = 32 * kb + (e < 4 ? 4 * q4 + e : 16 + 4 * q4 + (e - 4));
          f[e] = *reinterpret_cast<const short*>(vl + key * VSTRIDE + (dt * 16 + l15) * 2);
        }
        vf[kb][dt] = f;
      }

    const int n_qt = (HW + 15) >> 4;
    for (int qt = 0; qt < n_qt; ++qt) {
      const int qtok = qt * 16 + l15;
      short8 qf = (short8){0, 0, 0, 0, 0, 0, 0, 0};
      if (qtok < HW) qf = *reinterpret_cast<const short8*>(base + qtok * rs + head * 32 + q4 * 8);
      float4v st[NKT];
      float m = -1e30f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf, (float4v){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int key = kt * 16 + q4 * 4 + j;
          if (key >= HW) st[kt][j] = -1e30f;
          m = fmaxf(m, st[kt][j]);
        }
      }
      m = quad_row_max(m);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float p = __builtin_amdgcn_exp2f(st[kt][j] - m);  // Q is pre-scaled by log2(e)/sqrt(32)
          st[kt][j] = p;
          sum += p;
        }
      sum = quad_row_sum(sum);
      const float inv = 1.0f / sum;
      float4v o0 = (float4v){0.f, 0.f, 0.f, 0.f}, o1 = (float4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKT / 2; ++kb) {
        short8 pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pf[j] = (short)float_to_bf16(st[2 * kb][j]);
          pf[4 + j] = (short)float_to_bf16(st[2 * kb + 1][j]);
        }
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kb][0], pf, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kb][1], pf, o1, 0, 0, 0);
      }
      if (qtok < HW) {
        unsigned short* op = out + ((size_t)b * HW + qtok) * D + head * 32 + q4 * 4;
        ushort4 w0, w1;
        w0.x = float_to_bf16(o0[0] * inv); w0.y = float_to_bf16(o0[1] * inv);
        w0.z = float_to_bf16(o0[2] * inv); w0.w = float_to_bf16(o0[3] * inv);
        w1.x = float_to_bf16(o1[0] * inv); w1.y = float_to_bf16(o1[1] * inv);
        w1.z = float_to_bf16(o1[2] * inv); w1.w = float_to_bf16(o1[3] * inv);
        *reinterpret_cast<ushort4*>(op) = w0;
        *reinterpret_cast<ushort4*>(op + 16) = w1;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// Row LayerNorm (or plain copy when gamma == nullptr) of [rows, D] bf16 into the FC input
// buffer: sample b, pixel p lands at dst[b*dst_stride + p*D .. +D).  16 lanes per row.
__global__ void rows_to_fc_kernel(const unsigned short* __restrict__ src, const float* __restrict__ g,
                                  const float* __restrict__ bt, unsigned short* __restrict__ dst, int rows_total,
                                  int HW, int D, int dst_stride, float eps) {
  const int lane16 = threadIdx.x & 15;
  const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  if (row >= rows_total) return;  // whole 16-lane group exits together
  const int b = (int)(row / HW), p = (int)(row - (long long)b * HW);
  const unsigned short* sp = src + (size_t)row * D;
  unsigned short* dp = dst + (size_t)b * dst_stride + (size_t)p * D;
  float s = 0.f, ss = 0.f;
  if (g != nullptr) {
    for (int c = lane16; c < (D >> 3); c += 16) {
      short8 v = *reinterpret_cast<const short8*>(sp + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = bf16_to_float((unsigned short)v[j]);
        s += x;
        ss += x * x;
      }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      s += __shfl_xor(s, off, 64);
      ss += __shfl_xor(ss, off, 64);
    }
  }
  const float mean = s / (float)D;
  const float rstd = rsqrtf(fmaxf(ss / (float)D - mean * mean, 0.f) + eps);
  for (int c = lane16; c < (D >> 3); c += 16) {
    short8 v = *reinterpret_cast<const short8*>(sp + c * 8);
    if (g != nullptr) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = (bf16_to_float((unsigned short)v[j]) - mean) * rstd;
        v[j] = (short)float_to_bf16(x * g[c * 8 + j] + bt[c * 8 + j]);
      }
    }
    *reinterpret_cast<short8*>(dp + c * 8) = v;
  }
}

// "other" features f32 [B, Do] -> bf16 into the tail columns of the FC input buffer (zero padded).
__global__ void other_to_fc_kernel(const float* __restrict__ other, unsigned short* __restrict__ dst, int B, int Do,
                                   int Dpad, int dst_stride, int col0) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Dpad) return;
  const int b = (int)(idx / Dpad), c = (int)(idx - (long long)b * Dpad);
  float v = c < Do ? other[(size_t)b * Do + c] : 0.f;
  dst[(size_t)b * dst_stride + col0 + c] = float_to_bf16(v);
}

// ---------------------------------------------------------------------------
// Shared FC over the flattened body: out[b][n] = sum_k X[b][k] W[k][n], K ~ 15k, N <= 128.
// Split-K over blockIdx.y into f32 slabs [SK][B][128] (summed, deterministically, by fc_finish).
// Workgroup tile: 64 samples x 128 outputs; X chunks of 128 k go through LDS, weights come
// as packed fragments from L2.
__global__ __launch_bounds__(256, 2) void fc_splitk_kernel(const unsigned short* __restrict__ X,
                                                           const unsigned short* __restrict__ Wp,
                                                           float* __restrict__ slabs, int B, int Kp, int N,
                                                           int steps_per_split) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[64 * 272];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int b0 = blockIdx.x * 64;
  const int total_steps = Kp >> 5;
  const int s_begin = blockIdx.y * steps_per_split;
  const int s_end = min(s_begin + steps_per_split, total_steps);
  float4v acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t][0] = (float4v){0.f, 0.f, 0.f, 0.f};
    acc[t][1] = (float4v){0.f, 0.f, 0.f, 0.f};
  }
  const bool wave_active = wave * 32 < N;
  const short8* wp = reinterpret_cast<const short8*>(Wp) + (size_t)wave * 128 + lane;
  for (int s0 = s_begin; s0 < s_end; s0 += 4) {
    // stage X[64 samples][128 k] (zero beyond the batch / the split's K range)
    __syncthreads();
    for (int c = tid; c < 64 * 16; c += 256) {
      int r = c >> 4, cc = c & 15;
      int step = s0 + (cc >> 2);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (b0 + r < B && step < s_end) v = *reinterpret_cast<const uint4*>(X + (size_t)(b0 + r) * Kp + (size_t)s0 * 32 + cc * 8);
      *reinterpret_cast<uint4*>(xs + r * 272 + cc * 16) = v;
    }
    __syncthreads();
    if (!wave_active) continue;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int step = s0 + ks;
      if (step >= s_end) break;
      const short8 w0 = wp[(size_t)step * 512];
      const short8 w1 = wp[(size_t)step * 512 + 64];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const short8 xv = *reinterpret_cast<const short8*>(xs + (t * 16 + l15) * 272 + ks * 64 + q4 * 16);
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xv, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xv, acc[t][1], 0, 0, 0);
      }
    }
  }
  if (!wave_active) return;
  float* slab = slabs + (size_t)blockIdx.y * B * 128;
  const int ch0 = wave * 32 + q4 * 8;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int bi = b0 + t * 16 + l15;
    if (bi >= B) continue;
    float* op = slab + (size_t)bi * 128 + ch0;
    *reinterpret_cast<float4*>(op) = make_float4(acc[t][0][0], acc[t][0][1], acc[t][0][2], acc[t][0][3]);
    *reinterpret_cast<float4*>(op + 4) = make_float4(acc[t][1][0], acc[t][1][1], acc[t][1][2], acc[t][1][3]);
  }
}

// Shared FC fed directly by the body's NHWC rows (128 channels per pixel): the flatten is free (a sample's
// rows are contiguous), the final LayerNorm - when there is one - is applied to each 128-wide pixel row while
// it is staged into LDS (16 lanes per row), and the "other" features are read from the caller's f32 rows as the last
// K steps.  Replaces rows_to_fc + fc_splitk and their 2 x 126 MB round trip.
// steps_per_split is a multiple of 4, so every staged chunk of 4 K-steps is exactly one pixel row.
//
// MT * 16 samples x one K split per 8-wave workgroup, specialised: waves 4-7 are producers (global rows ->
// LayerNorm -> LDS tile, DEPTH pixel rows in flight in registers), waves 0-3 are consumers (wave = 32 output
// channels, MFMAs out of the tile, the weight slice just used refilled with the next pixel row's).  One barrier per
// pixel row, two tile buffers.  Both loops are straight-line code - addresses are clamped instead of predicated,
// samples past the live count compute on a valid row and are never stored - so that the compiler's wait counts leave
// the younger loads in flight (with branches in the loop it drains them all at every join).
template <int MT, bool LN, int DEPTH>
__global__ __launch_bounds__(512, 1) void fc_rows_splitk_kernel(const unsigned short* __restrict__ rows /*[B][HW][128]*/,
                                                                const float* __restrict__ other /*[B][other_dim] f32*/,
                                                                int other_dim,
                                                                const float* __restrict__ g, const float* __restrict__ bt,
                                                                float eps, const unsigned short* __restrict__ Wp,
                                                                float* __restrict__ slabs, int B, int HW, int other_pad,
                                                                int N, int steps_per_split, const int* __restrict__ live,
                                                                int max_splits) {
  constexpr int kPitch = 288;  // bytes per tile row: 16-byte fragment reads of 16 rows x 4 column groups hit distinct banks
  constexpr int CH = MT;       // 16-byte chunks per producer thread and pixel row
  __shared__ __attribute__((aligned(16))) unsigned char xs[2][MT * 16 * kPitch];
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
      gg[j] = LN ? g[(pt & 15) * 8 + j] : 1.f;
      bb[j] = LN ? bt[(pt & 15) * 8 + j] : 0.f;
    }
    // Sum over the 16 lanes that share (lane >> 4): quad permutes, half-row mirror, row mirror - all DPP, no LDS round trip.
    auto sum16 = [](float x) {
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
      return x;
    };
    const unsigned short* src[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = pt + i * 256, r = c >> 4, cc = c & 15;
      src[i] = rows + (size_t)min(b0 + r, nB - 1) * HW * 128 + cc * 8;
    }
    const int p0 = s_begin >> 2, p_last = max(p0 + n_feat - 1, 0);
    // first 128-feature chunk of the "other" features in this split (a split may start behind the body rows)
    const int k0 = max(s_begin - feat_steps, 0) >> 2;
    // body pixel row f of this split into registers; rows past the split's last one re-read it (loaded, never staged)
    auto issue = [&](int f, uint4 (&v)[CH]) {
      const size_t off = (size_t)min(p0 + f, p_last) * 128;
#pragma unroll
      for (int i = 0; i < CH; ++i) v[i] = *reinterpret_cast<const uint4*>(src[i] + off);
    };
    auto put = [&](short8 x, int i, int buf) {
      const int c = pt + i * 256, r = c >> 4, cc = c & 15;
      *reinterpret_cast<short8*>(xs[buf] + r * kPitch + cc * 16) = x;
    };
    auto stage = [&](const uint4 (&v)[CH], int buf) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        short8 x = __builtin_bit_cast(short8, v[i]);
        if (LN) {
          float f[8], s = 0.f, ss = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f[j] = bf16_to_float((unsigned short)x[j]);
            s += f[j];
            ss += f[j] * f[j];
          }
          s = sum16(s);
          ss = sum16(ss);
          const float mean = s * (1.0f / 128.0f);
          const float rstd = rsqrtf(fmaxf(ss * (1.0f / 128.0f) - mean * mean, 0.f) + eps);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * rstd * gg[j] + bb[j];
          x = pack_bf16x8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
        }
        put(x, i, buf);
      }
    };
    // chunk k of the "other" features (zero padded to other_pad; steps past the split's end become zero columns)
    auto stage_other = [&](int k, int buf) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = pt + i * 256, r = c >> 4, cc = c & 15;
        const int d0 = k * 128 + cc * 8;
        const float* orow = other + (size_t)min(b0 + r, nB - 1) * other_dim;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = d0 + j < other_dim ? orow[d0 + j] : 0.f;
        put(pack_bf16x8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]), i, buf);
      }
    };
    uint4 v[DEPTH][CH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, v[d]);
    // chunk i is staged during iteration i - 1 (chunk 0 before the loop); a body row's register slot is then refilled
    // with the row DEPTH further on
    if (n_feat > 0) {
      stage(v[0], 0);
      issue(DEPTH, v[0]);
    } else {
      stage_other(k0, 0);
    }
    __syncthreads();
    int i = 0;  // the chunk the consumers multiply now
    for (; i + DEPTH < n_feat; i += DEPTH) {  // DEPTH iterations with static register slots, no branches inside
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        stage(v[(d + 1) % DEPTH], (i + d + 1) & 1);
        issue(i + d + 1 + DEPTH, v[(d + 1) % DEPTH]);
        __syncthreads();
      }
    }
    // remaining iterations: fewer than DEPTH body rows, then the chunks of "other" features
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (i + d + 1 < n_feat) {
        stage(v[(d + 1) % DEPTH], (i + d + 1) & 1);
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
  const short8* wp = reinterpret_cast<const short8*>(Wp) + (size_t)wv * 128 + lane;
  short8 wf[4][2];
  // K-steps past the split's end are staged as zero columns; their weights are the last step's again (finite)
  auto load_wslice = [&](int s0, int ks) {
    const size_t step = (size_t)min(s0 + ks, s_end - 1) * 512;
    wf[ks][0] = wp[step];
    wf[ks][1] = wp[step + 64];
  };
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) load_wslice(s_begin, ks);
  __syncthreads();  // chunk 0 is staged
  for (int i = 0; i < n_rows; ++i) {
    const int s0 = s_begin + 4 * i;
    const unsigned char* x_base = xs[i & 1] + l15 * kPitch + q4 * 16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const short8 xv = *reinterpret_cast<const short8*>(x_base + t * 16 * kPitch + ks * 64);
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][0], xv, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][1], xv, acc[t][1], 0, 0, 0);
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

// h[b][n] = act(sum_s slabs[s][b][n] + bias[n]) -> bf16 [B, N]
__global__ void fc_finish_kernel(const float* __restrict__ slabs, const float* __restrict__ bias,
                                 unsigned short* __restrict__ out, int B, int N, int SK, int act) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * N) return;
  const int b = (int)(idx / N), n = (int)(idx - (long long)b * N);
  float v = bias[n];
  for (int s = 0; s < SK; ++s) v += slabs[((size_t)s * B + b) * 128 + n];
  out[(size_t)b * N + n] = float_to_bf16(apply_act(v, act));
}

// C51 expectation: value[b] = sum_i softmax(logits[b])_i * (vmin + i*dz).  16 lanes per sample
// (atoms strided over the lanes, reductions by xor-shuffles inside the 16-lane group).
__global__ void c51_expect_kernel(const float* __restrict__ logits, int stride, int atoms, float vmin, float dz,
                                  float* __restrict__ value, int B) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = gid >> 4, l = gid & 15;
  const bool live = b < B;
  const float* lp = logits + (size_t)(live ? b : 0) * stride;
  float m = -1e30f;
  for (int i = l; i < atoms; i += 16) m = fmaxf(m, lp[i]);
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  float s = 0.f, e = 0.f;
  for (int i = l; i < atoms; i += 16) {
    float p = __expf(lp[i] - m);
    s += p;
    e += p * (vmin + dz * (float)i);
  }
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    e += __shfl_xor(e, off, 64);
  }
  if (live && l == 0) value[b] = e / s;
}

// ---------------------------------------------------------------------------
// C ABI
namespace {
template <int MT, int DBG>
int launch_tile_gemm_inst(const TileGemmArgs& a, int grid, size_t lds, hipStream_t stream) {
  // the dynamic-LDS ceiling is raised once per kernel (not a stream operation: keep it out of graph capture)
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)tile_gemm_kernel<MT, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((tile_gemm_kernel<MT, DBG>), dim3(grid), dim3(256), lds, stream, a);
  return hip_err(hipGetLastError());
}

int launch_tile_gemm(const TileGemmArgs& a, hipStream_t stream, int dbg) {
  const int grid = (a.rows_total + a.rows_per_wg - 1) / a.rows_per_wg;
  const int stride = a.Cin * 2 + 16;
  if (a.rows_per_wg > 240) return launch_tile_gemm_inst<16, 0>(a, grid, (size_t)(16 * 16 + 1) * stride, stream);
  const size_t lds = (size_t)(15 * 16 + 1) * stride;
  switch (dbg) {
    case 0: return launch_tile_gemm_inst<15, 0>(a, grid, lds, stream);
    case 1: return launch_tile_gemm_inst<15, 1>(a, grid, lds, stream);
    case 2: return launch_tile_gemm_inst<15, 2>(a, grid, lds, stream);
    case 4: return launch_tile_gemm_inst<15, 4>(a, grid, lds, stream);
    case 8: return launch_tile_gemm_inst<15, 8>(a, grid, lds, stream);
    case 20: return launch_tile_gemm_inst<15, 20>(a, grid, lds, stream);
    case 36: return launch_tile_gemm_inst<15, 36>(a, grid, lds, stream);
    case 52: return launch_tile_gemm_inst<15, 52>(a, grid, lds, stream);
    default: return -14;
  }
}
}  // namespace

namespace {
const int* g_live_count = nullptr;
}  // namespace
namespace atrihip {
const int* live_count_ptr() { return g_live_count; }
}  // namespace atrihip

extern "C" {

// Device row count consulted by the engines' batch kernels launched from now on (nullptr: every launch covers its
// whole batch).  See common.hpp: live_rows.
int atri_set_live_count(const int* count) {
  g_live_count = count;
  return 0;
}



int atri_tile_gemm(const void* in, const void* w, const float* bias, const void* residual, const float* post_add,
                   const float* ln_g, const float* ln_b, void* out, int rows_total, int rows_per_wg, int HW, int H,
                   int W, int Cin, int Cout, int taps, int act, int out_f32, int out_stride, int n_blocks,
                   float ln_eps, hipStream_t stream) {
  if (rows_total <= 0) return 0;
  if (Cin < 32 || (Cin & (Cin - 1)) != 0 || Cin > 512) return -10;
  if (rows_per_wg <= 0 || rows_per_wg > 256) return -11;
  if (taps != 1 && taps != 9) return -12;
  if ((Cout & 7) != 0 || n_blocks * 128 < Cout) return -13;
  TileGemmArgs a;
  a.in = (const unsigned short*)in;
  a.w = (const unsigned short*)w;
  a.bias = bias;
  a.residual = (const unsigned short*)residual;
  a.post_add = post_add;
  a.ln_g = ln_g;
  a.ln_b = ln_b;
  a.out = out;
  a.rows_total = rows_total;
  a.rows_per_wg = rows_per_wg;
  a.HW = HW;
  a.H = H;
  a.W = W;
  a.Cin = Cin;
  a.Cout = Cout;
  a.taps = taps;
  a.act = act;
  a.out_f32 = out_f32;
  a.out_stride = out_stride;
  a.n_blocks = n_blocks;
  int l2 = 0;
  while ((1 << l2) < (Cin >> 3)) ++l2;
  a.log2_cpr = l2;
  a.ln_eps = ln_eps;
  return launch_tile_gemm(a, stream, 0);
}

// Roofline probe: same launch with one main-loop ingredient removed (see the kernel's DBG parameter).
int atri_tile_gemm_probe(const void* in, const void* w, const float* bias, void* out, int rows_total, int rows_per_wg,
                         int HW, int H, int W, int Cin, int Cout, int taps, int n_blocks, int dbg, hipStream_t stream) {
  if (rows_total <= 0) return 0;
  if (Cin < 32 || Cin > 512 || (Cin & (Cin - 1)) != 0) return -10;
  if (rows_per_wg <= 0 || rows_per_wg > 240) return -11;
  if (taps != 1 && taps != 9) return -12;
  if ((Cout & 7) != 0 || n_blocks * 128 < Cout || dbg < 0) return -13;
  TileGemmArgs a{};
  a.in = (const unsigned short*)in;
  a.w = (const unsigned short*)w;
  a.bias = bias;
  a.out = out;
  a.rows_total = rows_total;
  a.rows_per_wg = rows_per_wg;
  a.HW = HW;
  a.H = H;
  a.W = W;
  a.Cin = Cin;
  a.Cout = Cout;
  a.taps = taps;
  a.act = kActReLU;
  a.out_stride = Cout;
  a.n_blocks = n_blocks;
  int l2 = 0;
  while ((1 << l2) < (Cin >> 3)) ++l2;
  a.log2_cpr = l2;
  a.ln_eps = 1e-5f;
  return launch_tile_gemm(a, stream, dbg);
}

int atri_conv_first(const float* grid, const float* w, const float* bias, void* out, int rows_total, int HW, int H,
                    int W, int Cout, int act, hipStream_t stream) {
  if (rows_total <= 0) return 0;
  if ((Cout & 7) != 0) return -10;
  long long n = (long long)rows_total * (Cout >> 3);
  hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, grid, w, bias,
                     (unsigned short*)out, rows_total, HW, H, W, Cout, act);
  return hip_err(hipGetLastError());
}

int atri_attention(const void* qkv, void* out, int B, int HW, int D, int heads, hipStream_t stream) {
  if (B <= 0) return 0;
  if (D != heads * 32) return -10;
  if (HW <= 128) {
    size_t lds = (size_t)4 * 8 * 16 * 80;
    hipLaunchKernelGGL(attention_kernel<8>, dim3(B), dim3(256), lds, stream, (const unsigned short*)qkv,
                       (unsigned short*)out, B, HW, D, heads);
  } else if (HW <= 256) {
    size_t lds = (size_t)4 * 16 * 16 * 80;
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute((const void*)attention_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = true;
    }
    hipLaunchKernelGGL(attention_kernel<16>, dim3(B), dim3(256), lds, stream, (const unsigned short*)qkv,
                       (unsigned short*)out, B, HW, D, heads);
  } else {
    return -11;
  }
  return hip_err(hipGetLastError());
}

int atri_rows_to_fc(const void* src, const float* g, const float* bt, void* dst, int rows_total, int HW, int D,
                    int dst_stride, float eps, hipStream_t stream) {
  if (rows_total <= 0) return 0;
  if ((D & 7) != 0) return -10;
  long long threads = (long long)rows_total * 16;
  hipLaunchKernelGGL(rows_to_fc_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream,
                     (const unsigned short*)src, g, bt, (unsigned short*)dst, rows_total, HW, D, dst_stride, eps);
  return hip_err(hipGetLastError());
}

int atri_other_to_fc(const float* other, void* dst, int B, int Do, int Dpad, int dst_stride, int col0,
                     hipStream_t stream) {
  if (B <= 0) return 0;
  long long n = (long long)B * Dpad;
  hipLaunchKernelGGL(other_to_fc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, other,
                     (unsigned short*)dst, B, Do, Dpad, dst_stride, col0);
  return hip_err(hipGetLastError());
}

int atri_fc_splitk(const void* X, const void* Wp, float* slabs, int B, int Kp, int N, int SK, hipStream_t stream) {
  if (B <= 0) return 0;
  if ((Kp & 31) != 0 || N > 128 || (N & 7) != 0 || SK < 1) return -10;
  int total_steps = Kp >> 5;
  int per = (total_steps + SK - 1) / SK;
  per = (per + 3) & ~3;
  hipLaunchKernelGGL(fc_splitk_kernel, dim3((B + 63) / 64, SK), dim3(256), 0, stream, (const unsigned short*)X,
                     (const unsigned short*)Wp, slabs, B, Kp, N, per);
  return hip_err(hipGetLastError());
}

// Fused flatten (+ final LayerNorm) + shared FC for bodies with 128 channels per pixel.  SK splits of
// steps_per_split K-steps each (a multiple of 4); slabs is [SK][B][128] f32.
// Samples per workgroup of atri_fc_rows_splitk: 128-sample workgroups read the weights half as often as 64-sample
// ones; taken when they still give every CU one.
int atri_fc_rows_splitk_block_rows(int B, int SK) { return (long long)((B + 127) / 128) * SK >= 256 ? 128 : 64; }

// max_splits > SK: slabs must hold max_splits x B x 128 floats; the split count then follows the live row count (fc_plan)
// and atri_heads_dyn has to be given the same numbers.
int atri_fc_rows_splitk_dyn(const void* rows, const float* other, int other_dim, const float* g, const float* bt, float eps,
                            const void* Wp, float* slabs, int B, int HW, int D, int other_pad, int N, int SK,
                            int steps_per_split, int max_splits, hipStream_t stream) {
  if (B <= 0) return 0;
  if (D != 128 || (other_pad & 31) != 0 || other_dim < 0 || other_dim > other_pad || N > 128 || (N & 31) != 0) return -10;
  if ((steps_per_split & 3) != 0 || steps_per_split <= 0) return -11;
  if ((long long)SK * steps_per_split < (long long)HW * 4 + (other_pad >> 5)) return -12;
#define ATRI_FC_LAUNCH(MT, LN)                                                                                      \
  hipLaunchKernelGGL((fc_rows_splitk_kernel<MT, LN, 4>), dim3((B + MT * 16 - 1) / (MT * 16), SK), dim3(512), 0, stream, \
                     (const unsigned short*)rows, other, other_dim, g, bt, eps, (const unsigned short*)Wp, slabs, B, HW, \
                     other_pad, N, steps_per_split, live_count_ptr(), max_splits)
  if (atri_fc_rows_splitk_block_rows(B, SK) == 128) {
    if (g != nullptr) ATRI_FC_LAUNCH(8, true); else ATRI_FC_LAUNCH(8, false);
  } else {
    if (g != nullptr) ATRI_FC_LAUNCH(4, true); else ATRI_FC_LAUNCH(4, false);
  }
#undef ATRI_FC_LAUNCH
  return hip_err(hipGetLastError());
}

int atri_fc_rows_splitk(const void* rows, const float* other, int other_dim, const float* g, const float* bt, float eps,
                        const void* Wp, float* slabs, int B, int HW, int D, int other_pad, int N, int SK, int steps_per_split,
                        hipStream_t stream) {
  return atri_fc_rows_splitk_dyn(rows, other, other_dim, g, bt, eps, Wp, slabs, B, HW, D, other_pad, N, SK, steps_per_split,
                                 0, stream);
}

int atri_fc_finish(const float* slabs, const float* bias, void* out, int B, int N, int SK, int act,
                   hipStream_t stream) {
  if (B <= 0) return 0;
  long long n = (long long)B * N;
  hipLaunchKernelGGL(fc_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, slabs, bias,
                     (unsigned short*)out, B, N, SK, act);
  return hip_err(hipGetLastError());
}

int atri_c51_expect(const float* logits, int stride, int atoms, float vmin, float dz, float* value, int B,
                    hipStream_t stream) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(c51_expect_kernel, dim3((B * 16 + 255) / 256), dim3(256), 0, stream, logits, stride, atoms, vmin, dz,
                     value, B);
  return hip_err(hipGetLastError());
}

}  // extern "C"
