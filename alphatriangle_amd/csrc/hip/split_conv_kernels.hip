// fp32-grade 3x3 convolutions for boards with W <= 15 and H <= 8 (the default 8x15 grid): fp32 activations in
// memory, every GEMM as three 16-bit MFMA passes over (hi, lo) operand splits with fp32 accumulation.
//
// The reference evaluates its network in fp32 (alphatriangle/nn/network.py:243-318, tests/nn/test_model.py:80-81);
// gfx950's f32-input MFMA runs at 1/16 of the 16-bit rate, so the reference-precision path of this engine writes each
// fp32 operand as x = hi + lo with hi = r16(x), lo = r16(x - hi) and computes
//     W x  ~=  W_hi x_hi + W_lo x_hi + W_hi x_lo          (the dropped W_lo x_lo term is below the operands' own rounding)
// on the 16-bit matrix cores: 3 MFMAs per product instead of 1, accumulators and everything element-wise in fp32.
// r16 is fp16 (22 significant bits per operand) or bf16 (16 bits), fixed when the library is built (common.hpp).
// Measured against an fp64 evaluation of the default network the logits are within 0.6 - 1.2e-6 with fp16 pairs and
// 4 - 6e-6 with bf16 pairs (PyTorch's fp32 path: 0.5 - 2.3e-6; a plain bf16 evaluation: ~1e-2) - see
// tests/test_gpu_nn_fp32.py.
//
// Layers served: the conv stem and both convolutions of every residual block
// (alphatriangle/nn/model.py:20-37 conv_block, :40-59 ResidualBlock), BatchNorm folded.
//
// Structure (same halo-tile idea as conv_kernels.hip, re-balanced for a loop that is matrix-core bound):
//   * one workgroup per sample; the sample sits in LDS as two 16-bit planes (hi, lo) with a physical zero halo at a
//     16-pixel row pitch, so a 3x3 tap is a constant LDS offset and one fragment read of image row r serves the
//     three vertical taps.  Only the 130 tile rows the main loop can touch are kept (37.4 KB per 128-channel
//     plane): two planes are 75 KB, two workgroups per CU.
//   * the loop issues 18 MFMAs per pair of 16-byte LDS reads; weight fragments (hi and lo streams) of the next
//     (channel chunk, horizontal tap) group are requested a whole group ahead.
//   * accumulators leave through an fp32 staging tile in LDS so that global stores, the skip connection and the
//     positional table move as whole 512-byte pixel rows.
#include "common.hpp"

using namespace atrihip;

struct SplitConvArgs {
  const float* in;         // [B, HW, Cin] f32
  const unsigned short* w; // packed 16-bit fragments, hi stream then lo stream (lo_off elements apart)
  const float* bias;       // [>= Cout]
  const float* residual;   // [B, HW, Cout] f32 or nullptr (added before the activation)
  const float* post_add;   // [HW, Cout] f32 or nullptr (added after the activation)
  float* out;              // [B, HW, Cout] f32
  long long lo_off, lo_off2;
  int B, H, W, HW, Cin, Cout, act, ymagic;
  const int* live;       // device row count (nullptr: B rows), see common.hpp
  // FIRST: the network's first layer (1 input channel, 3x3) is evaluated while staging
  const float* grid;     // [B, HW] f32 board planes
  const float* w_first;  // [9][Cin] f32 (BatchNorm folded)
  const float* b_first;  // [Cin]
  int first_act, aux_off;
  // PAIR: the second convolution (residual block: in place, Cin = Cout = 128; conv stem: CIN2 channels in a second tile)
  const unsigned short* w2;
  const float* bias2;
  int tile2_off;
};

namespace {
constexpr int kTileRows = 130;    // tile row = 16 y + x + 1 for pixel (y, x); row 0 and the x = 15 column are halo
constexpr int kOutPitch = 528;    // bytes per pixel row of the fp32 output staging tile (128 channels + 16 B pad)
constexpr int kAheadS = 2;

__device__ __noinline__ float4v act4_slow(float4v v, int act) {
  for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], act);
  return v;
}
__device__ __forceinline__ float4v act4s(float4v v, int act) {
  if (act == kActReLU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    return v;
  }
  return act == kActNone ? v : act4_slow(v, act);
}

// One 3x3 convolution of this wave's 32 output channels over the (hi, lo) halo tiles behind base_hi / base_lo, output
// image rows T0 .. T0 + NT - 1 (the image rows one above and one below that range are read as well).
template <int CIN, int T0 = 0, int NT = 8>
__device__ __forceinline__ void split_mainloop(float4v (&acc)[8][2], const unsigned short* __restrict__ wsrc, long long lo_off,
                                               const unsigned char* base_hi, const unsigned char* base_lo, int wave, int lane) {
  constexpr int stride = CIN * 2 + 32;
  constexpr int KC = CIN / 32, STEPS = 9 * KC, GROUPS = 3 * KC;
  const short8* wph = reinterpret_cast<const short8*>(wsrc) + (size_t)wave * 128 + lane;
  const short8* wpl = reinterpret_cast<const short8*>(wsrc + lo_off) + (size_t)wave * 128 + lane;
  (void)STEPS;
  constexpr int R0 = T0 > 0 ? T0 - 1 : 0, R1 = T0 + NT < 8 ? T0 + NT : 7, NR = R1 - R0 + 1;  // image rows read
#define SPLIT_WSTEP(g, dyi) ((((dyi) * 3 + (g) % 3) * KC) + (g) / 3)
  // tile row of image row r at horizontal tap dxi: (r * 16 + l15 + 1) + (dxi - 1); l15 is in the lane base
#define SPLIT_ROW(g, r) (((r) * 16 + ((g) % 3)) * stride + ((g) / 3) * 64)
  short8 wh[2][3][2], wl[2][3][2], xh[kAheadS + 1], xl[kAheadS + 1];
#pragma unroll
  for (int dyi = 0; dyi < 3; ++dyi) {
    wh[0][dyi][0] = wph[(size_t)SPLIT_WSTEP(0, dyi) * 512];
    wh[0][dyi][1] = wph[(size_t)SPLIT_WSTEP(0, dyi) * 512 + 64];
    wl[0][dyi][0] = wpl[(size_t)SPLIT_WSTEP(0, dyi) * 512];
    wl[0][dyi][1] = wpl[(size_t)SPLIT_WSTEP(0, dyi) * 512 + 64];
  }
#pragma unroll
  for (int i = 0; i < kAheadS; ++i) {
    xh[i] = *reinterpret_cast<const short8*>(base_hi + SPLIT_ROW(i / NR, R0 + i % NR));
    xl[i] = *reinterpret_cast<const short8*>(base_lo + SPLIT_ROW(i / NR, R0 + i % NR));
  }
  // unroll(full), not `#pragma unroll`: the 128-channel instantiation (96 iterations) was left as two half-rolled trips
#pragma clang loop unroll(full)
  for (int i = 0; i < GROUPS * NR; ++i) {
    const int g = i / NR, r = R0 + i % NR, wc = g & 1, wn = wc ^ 1;
    __builtin_amdgcn_sched_barrier(0);
    if (r == R0 && g + 1 < GROUPS) {
#pragma unroll
      for (int dyi = 0; dyi < 3; ++dyi) {
        wh[wn][dyi][0] = wph[(size_t)SPLIT_WSTEP(g + 1, dyi) * 512];
        wh[wn][dyi][1] = wph[(size_t)SPLIT_WSTEP(g + 1, dyi) * 512 + 64];
        wl[wn][dyi][0] = wpl[(size_t)SPLIT_WSTEP(g + 1, dyi) * 512];
        wl[wn][dyi][1] = wpl[(size_t)SPLIT_WSTEP(g + 1, dyi) * 512 + 64];
      }
    }
    if (i + kAheadS < GROUPS * NR) {
      xh[(i + kAheadS) % (kAheadS + 1)] = *reinterpret_cast<const short8*>(base_hi + SPLIT_ROW((i + kAheadS) / NR, R0 + (i + kAheadS) % NR));
      xl[(i + kAheadS) % (kAheadS + 1)] = *reinterpret_cast<const short8*>(base_lo + SPLIT_ROW((i + kAheadS) / NR, R0 + (i + kAheadS) % NR));
    }
    __builtin_amdgcn_sched_barrier(0);
    const short8 ah = xh[i % (kAheadS + 1)], al = xl[i % (kAheadS + 1)];
    // image row r feeds output row r - 1 through the dy = +1 tap, row r through dy = 0, row r + 1 through dy = -1
#pragma unroll
    for (int dyi = 0; dyi < 3; ++dyi) {
      const int t = r + 1 - dyi;
      if (t < T0 || t >= T0 + NT) continue;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        acc[t][n] = split_mma(wl[wc][dyi][n], ah, acc[t][n]);
        acc[t][n] = split_mma(wh[wc][dyi][n], al, acc[t][n]);
        acc[t][n] = split_mma(wh[wc][dyi][n], ah, acc[t][n]);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#undef SPLIT_WSTEP
#undef SPLIT_ROW
}
}  // namespace

// FIRST: the input tile is computed from the raw board (first conv layer, Cin_first = 1) instead of being read.
// PAIR: two convolutions per launch.  CIN2 == CIN: a residual block (alphatriangle/nn/model.py:40-59) - the first
// convolution's activated output is split and written straight back into the halo tile, the second one runs from
// there and takes the skip connection from the block input in memory.  CIN2 != CIN: two consecutive stem layers; the
// first one's output goes into a second halo tile of CIN2 channels.
// RELU: the activation is ReLU (the reference's default) - compiled in, so that the epilogues and the write-back have
// no per-chunk branches on the activation code; other activations go through the run-time form.
template <int CIN, bool FIRST, bool PAIR, int CIN2, bool RELU>
__global__ __launch_bounds__(256, 2) void rowconv_split_kernel(SplitConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.x;
  if (b >= live_rows(a.live, a.B)) return;  // whole workgroup, before any barrier
  const int act = RELU ? (int)kActReLU : a.act;
  const int first_act = RELU ? (int)kActReLU : a.first_act;
  constexpr int stride = CIN * 2 + 32;
  constexpr int stride2 = CIN2 * 2 + 32;
  constexpr int plane = kTileRows * stride, plane2 = kTileRows * stride2;
  const int H = a.H, W = a.W;
  unsigned char* t1h = smem;
  unsigned char* t1l = smem + plane;
  unsigned char* t2h = (CIN2 != CIN) ? smem + a.tile2_off : t1h;
  unsigned char* t2l = (CIN2 != CIN) ? t2h + plane2 : t1l;
  if (PAIR && CIN2 != CIN) {  // the second tile's halo has to be zero; its interior is written by the first convolution
    for (int c = tid; c < 2 * plane2 / 16; c += 256) reinterpret_cast<uint4*>(t2h)[c] = make_uint4(0, 0, 0, 0);
  }

  // ---- stage the sample with its zero halo, split into the two planes ----
  constexpr int CPR = CIN / 8;  // 8-channel chunks per pixel row
  if (FIRST) {
    float* aux = reinterpret_cast<float*>(smem + a.aux_off);
    float* gplane = aux;                   // [(H + 2) * (W + 2)] board plane with a zero border
    float* wl = aux + 192;                 // [9][CIN]
    float* bl = wl + 9 * CIN;              // [CIN]
    const int PW = W + 2;
    for (int i = tid; i < (H + 2) * PW; i += 256) {
      const int yy = i / PW - 1, xx = i - (yy + 1) * PW - 1;
      gplane[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? a.grid[(size_t)b * a.HW + yy * W + xx] : 0.f;
    }
    for (int i = tid; i < 9 * CIN; i += 256) wl[i] = a.w_first[i];
    for (int i = tid; i < CIN; i += 256) bl[i] = a.b_first[i];
    __syncthreads();
    for (int c = tid; c < kTileRows * CPR; c += 256) {
      const int r = c / CPR, cc = c - r * CPR;
      const int rr = r - 1, y = rr >> 4, x = rr & 15;
      short8 oh = (short8){0, 0, 0, 0, 0, 0, 0, 0}, ol = oh;
      if (rr >= 0 && y < H && x < W) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bl[cc * 8 + j];
        const float* gp = gplane + y * PW + x;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float gv = gp[(tap / 3) * PW + tap % 3];
          const float4* wq = reinterpret_cast<const float4*>(wl + tap * CIN + cc * 8);
          const float4 w0 = wq[0], w1 = wq[1];
          v[0] += gv * w0.x; v[1] += gv * w0.y; v[2] += gv * w0.z; v[3] += gv * w0.w;
          v[4] += gv * w1.x; v[5] += gv * w1.y; v[6] += gv * w1.z; v[7] += gv * w1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j], first_act);
        split_op8(v, oh, ol);
      }
      *reinterpret_cast<short8*>(t1h + r * stride + cc * 16) = oh;
      *reinterpret_cast<short8*>(t1l + r * stride + cc * 16) = ol;
    }
  } else {
    const float* src = a.in + (size_t)b * a.HW * CIN;
    constexpr int total = kTileRows * CPR;
    for (int c0 = tid; c0 < total; c0 += 256 * 4) {
      float4 v[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + i * 256;
        v[i][0] = v[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int r = c / CPR, cc = c - r * CPR;
        const int rr = r - 1, y = rr >> 4, x = rr & 15;
        if (c < total && rr >= 0 && y < H && x < W) {
          const float4* p = reinterpret_cast<const float4*>(src + (size_t)(y * W + x) * CIN + cc * 8);
          v[i][0] = p[0];
          v[i][1] = p[1];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + i * 256;
        if (c >= total) continue;
        const int r = c / CPR, cc = c - r * CPR;
        const float f[8] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w, v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w};
        short8 oh, ol;
        split_op8(f, oh, ol);
        *reinterpret_cast<short8*>(t1h + r * stride + cc * 16) = oh;
        *reinterpret_cast<short8*>(t1l + r * stride + cc * 16) = ol;
      }
    }
  }
  __syncthreads();
  // The main loops run at a raised issue priority, staging and write-back at the default one: of the two workgroups that
  // share a CU the one in a matrix phase then never queues behind the other's I/O instructions (measured both ways:
  // I/O phases raised +-0, matrix phases raised +0.5 .. +1.9 % moves/s over four same-box pairs).
  __builtin_amdgcn_s_setprio(1);

  // lane base: tile row l15 (+1 is folded into the tap offset table: dx = -1 reads row l15 + 0)
  const unsigned char* b1h = t1h + l15 * stride + q4 * 16;
  const unsigned char* b1l = t1l + l15 * stride + q4 * 16;
  const unsigned char* b2h = t2h + l15 * stride2 + q4 * 16;
  const unsigned char* b2l = t2l + l15 * stride2 + q4 * 16;
  float* otile = reinterpret_cast<float*>(smem);

  const int ch_wave = wave * 32;
  // A 64-channel first convolution of a pair (the conv stem's middle layer) would occupy two of the four waves: it is
  // split by image rows as well, wave -> (channel block wave & 1, rows 4 (wave >> 1) .. + 3).
  constexpr bool kRowSplit = PAIR && CIN2 != CIN && CIN2 == 64;
  float4v acc[8][2];
#pragma unroll(CIN2 != CIN ? 2 : 1)
  for (int stage = 0; stage < (PAIR ? 2 : 1); ++stage) {
    const unsigned short* wsrc = (PAIR && stage == 1) ? a.w2 : a.w;
    const long long lo_off = (PAIR && stage == 1) ? a.lo_off2 : a.lo_off;
    const float* bsrc = (PAIR && stage == 1) ? a.bias2 : a.bias;
    const bool active = (PAIR && stage == 0) ? (kRowSplit || ch_wave < CIN2) : (ch_wave < a.Cout);  // wave-uniform
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      acc[t][0] = (float4v){0.f, 0.f, 0.f, 0.f};
      acc[t][1] = (float4v){0.f, 0.f, 0.f, 0.f};
    }
    // the bias is requested before the main loop: read after it, it would be a memory round trip in front of the epilogue
    const int cw = (kRowSplit && stage == 0) ? (wave & 1) : wave;  // channel block of this wave in this stage
    float bias[8];
    if (active) {
#pragma unroll
      for (int i = 0; i < 8; ++i) bias[i] = bsrc[cw * 32 + q4 * 8 + i];
    }
    if (active) {
      if (PAIR && stage == 1 && CIN2 != CIN)
        split_mainloop<CIN2>(acc, wsrc, lo_off, b2h, b2l, wave, lane);
      else if (kRowSplit && stage == 0 && wave < 2)
        split_mainloop<CIN, 0, 4>(acc, wsrc, lo_off, b1h, b1l, wave & 1, lane);
      else if (kRowSplit && stage == 0)
        split_mainloop<CIN, 4, 4>(acc, wsrc, lo_off, b1h, b1l, wave & 1, lane);
      else
        split_mainloop<CIN>(acc, wsrc, lo_off, b1h, b1l, wave, lane);
    }
    __syncthreads();  // every wave is done reading the input tile (the next writes alias it)
    if (PAIR && stage == 0) {
      // first convolution: bias + activation in fp32, split, back into the halo tile as the second one's input
      if (active && l15 < W) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (t >= H || (kRowSplit && (t >> 2) != (wave >> 1))) continue;
          float4v lo = acc[t][0], hi = acc[t][1];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            lo[j] += bias[j];
            hi[j] += bias[4 + j];
          }
          lo = act4s(lo, act);
          hi = act4s(hi, act);
          const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          short8 oh, ol;
          split_op8(f, oh, ol);
          const int off = (t * 16 + l15 + 1) * stride2 + cw * 64 + q4 * 16;
          *reinterpret_cast<short8*>(t2h + off) = oh;
          *reinterpret_cast<short8*>(t2l + off) = ol;
        }
      }
      __syncthreads();
      continue;
    }
    if (active) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float* op = otile + (size_t)(t * 16 + l15) * (kOutPitch / 4) + wave * 32 + q4 * 8;
        *reinterpret_cast<float4*>(op) = make_float4(acc[t][0][0] + bias[0], acc[t][0][1] + bias[1], acc[t][0][2] + bias[2], acc[t][0][3] + bias[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(acc[t][1][0] + bias[4], acc[t][1][1] + bias[5], acc[t][1][2] + bias[6], acc[t][1][3] + bias[7]);
      }
    }
    break;  // last stage: the barrier that publishes the output tile comes after the skip-connection loads below
  }
  __builtin_amdgcn_s_setprio(0);

  // ---- coalesced write-back: thread -> (pixel, 4-channel chunk), whole pixel rows per 32 lanes ----
  const int n_ch = min(128, a.Cout);
  const int cpp = n_ch >> 2;  // float4 chunks per pixel (power of two)
  int l2c = 0;
  while ((1 << l2c) < cpp) ++l2c;
  const int chunks = a.HW << l2c;
  const size_t gbase = (size_t)b * a.HW * a.Cout;
  // The skip-connection rows of the whole sample (at most 16 chunks per thread: 128 pixels x 32 chunks) are requested
  // now that the accumulators are dead, before the barrier: they arrive while the workgroup waits for its last wave.
  float4 rall[16];
  if (a.residual != nullptr) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = min(tid + i * 256, chunks - 1);
      rall[i] = *reinterpret_cast<const float4*>(a.residual + gbase + (size_t)(c >> l2c) * a.Cout + (c & (cpp - 1)) * 4);
    }
  }
  __syncthreads();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int c0 = tid + half * 256 * 8;
    if (c0 >= chunks) break;
    // positional-table values of all eight chunks are requested before the first is used: a table value read inside
    // the chunk loop would wait for the previous chunk's store as well (one counter for both)
    float4 rv[8], pv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) rv[i] = rall[half * 8 + i];
    if (a.post_add != nullptr) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = min(c0 + i * 256, chunks - 1);
        pv[i] = *reinterpret_cast<const float4*>(a.post_add + (size_t)(c >> l2c) * a.Cout + (c & (cpp - 1)) * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i * 256;
      if (c >= chunks) continue;
      const int p = c >> l2c, cc = c & (cpp - 1);
      const int y = (p * a.ymagic) >> 16, x = p - y * W;
      const float4 hv = *reinterpret_cast<const float4*>(otile + (size_t)(y * 16 + x) * (kOutPitch / 4) + cc * 4);
      float4v v = (float4v){hv.x, hv.y, hv.z, hv.w};
      if (a.residual != nullptr) {
        v[0] += rv[i].x; v[1] += rv[i].y; v[2] += rv[i].z; v[3] += rv[i].w;
      }
      v = act4s(v, act);
      if (a.post_add != nullptr) {
        v[0] += pv[i].x; v[1] += pv[i].y; v[2] += pv[i].z; v[3] += pv[i].w;
      }
      *reinterpret_cast<float4*>(a.out + gbase + (size_t)p * a.Cout + cc * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

namespace {
template <int CIN, bool FIRST, bool PAIR, int CIN2, bool RELU>
int launch_split_inst(const SplitConvArgs& a, int lds, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)rowconv_split_kernel<CIN, FIRST, PAIR, CIN2, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((rowconv_split_kernel<CIN, FIRST, PAIR, CIN2, RELU>), dim3(a.B), dim3(256), lds, stream, a);  // a.live set by the callers
  return hip_err(hipGetLastError());
}
template <int CIN, bool FIRST, bool PAIR, int CIN2>
int launch_split(const SplitConvArgs& a, int lds, hipStream_t stream) {
  const bool relu = a.act == kActReLU && (!FIRST || a.first_act == kActReLU);
  return relu ? launch_split_inst<CIN, FIRST, PAIR, CIN2, true>(a, lds, stream) : launch_split_inst<CIN, FIRST, PAIR, CIN2, false>(a, lds, stream);
}

int check_geometry(int H, int W, int HW) {
  if (H < 1 || H > 8 || W < 1 || W > 15 || HW != H * W) return -10;
  const int ymagic = 65536 / W + 1;
  for (int p = 0; p < HW; ++p)
    if (((p * ymagic) >> 16) != p / W) return -12;
  return 0;
}
constexpr int kOutBytes = 128 * kOutPitch;
}  // namespace

extern "C" {

// Residual block in one launch, fp32-grade: out = act(conv2(act(conv1(in))) + in) (+ post_add), 128 channels.
// w1 / w2: packed 16-bit fragments [hi stream | lo stream], lo1 / lo2 = element offset of the lo stream.
int atri_split_conv_pair(const float* in, const void* w1, long long lo1, const float* b1, const void* w2, long long lo2,
                         const float* b2, const float* post_add, float* out, int B, int H, int W, int C, int act,
                         hipStream_t stream) {
  if (B <= 0) return 0;
  if (C != 128) return -10;
  if (int rc = check_geometry(H, W, H * W)) return rc;
  if (in == out) return -13;  // the skip connection is read after other workgroups may have written their output
  SplitConvArgs a{};
  a.in = in;
  a.w = (const unsigned short*)w1;
  a.lo_off = lo1;
  a.bias = b1;
  a.w2 = (const unsigned short*)w2;
  a.lo_off2 = lo2;
  a.bias2 = b2;
  a.residual = in;
  a.post_add = post_add;
  a.out = out;
  a.B = B; a.H = H; a.W = W; a.HW = H * W; a.Cin = C; a.Cout = C; a.act = act;
  a.live = live_count_ptr();
  a.ymagic = 65536 / W + 1;
  const int tiles = 2 * kTileRows * (C * 2 + 32);
  const int lds = tiles > kOutBytes ? tiles : kOutBytes;
  return launch_split<128, false, true, 128>(a, lds, stream);
}

// Conv stem in one launch, fp32-grade: board plane -> conv(1 -> 32) -> conv(32 -> 64) -> conv(64 -> 128), every layer
// followed by the activation; only the last layer's output goes to memory.
int atri_split_conv_stem(const float* grid, const float* w_first, const float* b_first, const void* w1, long long lo1,
                         const float* b1, const void* w2, long long lo2, const float* b2, const float* post_add, float* out,
                         int B, int H, int W, int C0, int C1, int C2, int act, hipStream_t stream) {
  if (B <= 0) return 0;
  if (C0 != 32 || C1 != 64 || C2 != 128) return -10;
  if (int rc = check_geometry(H, W, H * W)) return rc;
  SplitConvArgs a{};
  a.w = (const unsigned short*)w1;
  a.lo_off = lo1;
  a.bias = b1;
  a.w2 = (const unsigned short*)w2;
  a.lo_off2 = lo2;
  a.bias2 = b2;
  a.post_add = post_add;
  a.out = out;
  a.B = B; a.H = H; a.W = W; a.HW = H * W; a.Cin = C0; a.Cout = C2; a.act = act;
  a.live = live_count_ptr();
  a.grid = grid;
  a.w_first = w_first;
  a.b_first = b_first;
  a.first_act = act;
  a.ymagic = 65536 / W + 1;
  const int tile1 = 2 * kTileRows * (C0 * 2 + 32), tile2 = 2 * kTileRows * (C1 * 2 + 32);
  a.tile2_off = (tile1 + 15) & ~15;
  int lds = a.tile2_off + tile2 > kOutBytes ? a.tile2_off + tile2 : kOutBytes;
  a.aux_off = (lds + 15) & ~15;  // board plane + first-layer weights live behind the tiles (and behind the output staging tile)
  lds = a.aux_off + (192 + 10 * C0) * 4;
  return launch_split<32, true, true, 64>(a, lds, stream);
}

// One fp32-grade 3x3 convolution (+ bias, residual, activation, positional table), Cin in {32, 64, 128}, Cout <= 128.
int atri_split_conv(const float* in, const void* w, long long lo, const float* bias, const float* residual,
                    const float* post_add, float* out, int B, int H, int W, int Cin, int Cout, int act, hipStream_t stream) {
  if (B <= 0) return 0;
  if (Cout < 32 || Cout > 128 || (Cout & (Cout - 1)) != 0) return -10;
  if (int rc = check_geometry(H, W, H * W)) return rc;
  SplitConvArgs a{};
  a.in = in;
  a.w = (const unsigned short*)w;
  a.lo_off = lo;
  a.bias = bias;
  a.residual = residual;
  a.post_add = post_add;
  a.out = out;
  a.B = B; a.H = H; a.W = W; a.HW = H * W; a.Cin = Cin; a.Cout = Cout; a.act = act;
  a.live = live_count_ptr();
  a.ymagic = 65536 / W + 1;
  const int tiles = 2 * kTileRows * (Cin * 2 + 32);
  const int lds = tiles > kOutBytes ? tiles : kOutBytes;
  if (Cin == 32) return launch_split<32, false, false, 32>(a, lds, stream);
  if (Cin == 64) return launch_split<64, false, false, 64>(a, lds, stream);
  if (Cin == 128) return launch_split<128, false, false, 128>(a, lds, stream);
  return -11;
}

}  // extern "C"
