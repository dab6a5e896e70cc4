// 3x3 / stride 1 / pad 1 convolution as an MFMA implicit GEMM for boards with W <= 15 and H <= 8
// (the default 8x15 grid), NHWC bf16 in and out, one workgroup per sample.
//
// Layers served: conv stem layers 1.. and both convs of every residual block
// (alphatriangle/nn/model.py:20-37 conv_block, :40-59 ResidualBlock), BatchNorm folded.
//
// Design points, each one the answer to a measured cost of the generic tile kernel
// (bench/conv_probe.py removes one ingredient at a time):
//   * The sample is staged into LDS with a physical zero halo at a row pitch of 16 pixels
//     (pixel (y, x) -> LDS row 16y + x + 17).  One MFMA M-tile is one image row, a 3x3 tap is a
//     constant LDS offset, out-of-image taps read zeros: the main loop has no per-lane address
//     math at all (it was ~40 % of the old loop's time), only ds_read with immediate offsets.
//   * The accumulators leave through LDS: lanes hold (pixel, 8 channels) fragments whose direct
//     16-byte stores touch 16 different rows per instruction (1.5 TB/s measured); after the
//     transpose each wave writes whole 256-byte rows, and the residual / positional table
//     are read with the same coalesced pattern.
//   * 44 KB of LDS and ~110 VGPRs per workgroup: three workgroups (12 waves) per CU overlap
//     staging, MFMA and write-back phases of different samples.
#include "common.hpp"

using namespace atrihip;

struct RowConvArgs {
  const unsigned short* in;        // [B, HW, Cin] bf16
  const unsigned short* w;         // packed fragments [NB][9*KC][wave 4][n 2][lane 64][8] (+1 spare step)
  const float* bias;               // [NB*128]
  const unsigned short* residual;  // [B, HW, Cout] bf16 or nullptr (added before the activation)
  const float* post_add;           // [HW, Cout] f32 or nullptr (added after the activation)
  unsigned short* out;             // [B, HW, Cout] bf16
  int B, H, W, HW, Cin, Cout, act, n_blocks, log2_cpr, out_off, ymagic;
  const int* live;                 // device row count (nullptr: B rows), see common.hpp
  // FIRST instantiation only: the network's first layer (1 input channel, 3x3) is evaluated while staging
  const float* grid;     // [B, HW] f32 board planes
  const float* w_first;  // [9][Cin] f32 (BatchNorm folded)
  const float* b_first;  // [Cin]
  int first_act, aux_off;
  // PAIR instantiation only: the second convolution (residual block: in place, Cin = Cout = 128; conv stem: Cin2
  // channels in a second halo tile at tile2_off)
  const unsigned short* w2;
  const float* bias2;
  int tile2_off;
};

namespace {
constexpr int kInRows = 162;     // (8 + 2) * 16 + 2 halo-padded pixel rows
constexpr int kOutStride = 272;  // bytes per row of the 128-channel output staging tile
constexpr int kAhead = 2;        // image-row fragments requested ahead of the MFMAs that consume them

// By value on purpose: an address-taken float[8] would live in scratch memory on every path.
__device__ __noinline__ float4v act8_slow_rc(float4v v, int act) {
  for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], act);
  return v;
}
}  // namespace

namespace {
// One 3x3 convolution of this wave's 32 output channels over the halo tile behind lane_base:
// acc[t][n] += W x tile for the output image rows t = T0 .. T0 + NT - 1 (the image rows one above and one below
// that range are read as well).
template <int CIN, int DBG, int T0 = 0, int NT = 8>
__device__ __forceinline__ void rowconv_mainloop(float4v (&acc)[8][2], const unsigned short* __restrict__ wsrc, int nb,
                                                 const unsigned char* lane_base, int wave, int lane) {
  constexpr int stride = CIN * 2 + 32;
  constexpr int KC = CIN / 32, STEPS = 9 * KC;
  // GEMM steps are grouped by (channel chunk kc, horizontal tap dx).  Inside a group the B fragment of output
  // row t and vertical tap dy is the image-row fragment r = t + dy, so ONE LDS read per image row serves the
  // three vertical taps (24 fragment reads per channel chunk instead of 72: the LDS pipe and the MFMA pipe
  // were both saturated before), and the all-zero halo rows r = -1 and r = 8 are not multiplied at all
  // (44 MFMAs per group instead of 48).  Fully unrolled; row fragments are requested kAhead rows ahead, the six
  // weight fragments of the next group at the start of the current one; sched_barrier(0) fences keep the
  // compiler from sinking those loads down to their first use.
  const short8* wp = reinterpret_cast<const short8*>(wsrc) + ((size_t)nb * STEPS * 4 + wave) * 128 + lane;
  constexpr int GROUPS = 3 * KC;
  constexpr int R0 = T0 > 0 ? T0 - 1 : 0, R1 = T0 + NT < 8 ? T0 + NT : 7, NR = R1 - R0 + 1;  // image rows read
  // group g -> kc = g / 3, dxi = g % 3; weight step of vertical tap dyi: (dyi * 3 + dxi) * KC + kc
#define ROWCONV_WSTEP(g, dyi) ((((dyi) * 3 + (g) % 3) * KC) + (g) / 3)
  // LDS byte offset (relative to lane_base) of image row r at horizontal tap dxi of channel chunk kc
#define ROWCONV_ROW(g, r) ((17 + (r) * 16 + ((g) % 3 - 1)) * stride + ((g) / 3) * 64)
  short8 wg[2][3][2], av[kAhead + 1];
#pragma unroll
  for (int dyi = 0; dyi < 3; ++dyi) {
    wg[0][dyi][0] = wp[(size_t)ROWCONV_WSTEP(0, dyi) * 512];
    wg[0][dyi][1] = wp[(size_t)ROWCONV_WSTEP(0, dyi) * 512 + 64];
  }
#pragma unroll
  for (int i = 0; i < kAhead; ++i) av[i] = *reinterpret_cast<const short8*>(lane_base + ROWCONV_ROW(i / NR, R0 + i % NR));
#pragma unroll
  for (int i = 0; i < GROUPS * NR; ++i) {
    const int g = i / NR, r = R0 + i % NR, wc = g & 1, wn = wc ^ 1;
    __builtin_amdgcn_sched_barrier(0);
    if (r == R0 && g + 1 < GROUPS) {
#pragma unroll
      for (int dyi = 0; dyi < 3; ++dyi) {
        wg[wn][dyi][0] = (DBG & 2) ? wg[wc][dyi][0] : wp[(size_t)ROWCONV_WSTEP(g + 1, dyi) * 512];
        wg[wn][dyi][1] = (DBG & 2) ? wg[wc][dyi][1] : wp[(size_t)ROWCONV_WSTEP(g + 1, dyi) * 512 + 64];
      }
    }
    if (i + kAhead < GROUPS * NR)
      av[(i + kAhead) % (kAhead + 1)] =
          (DBG & 1) ? av[i % (kAhead + 1)]
                    : *reinterpret_cast<const short8*>(lane_base + ROWCONV_ROW((i + kAhead) / NR, R0 + (i + kAhead) % NR));
    __builtin_amdgcn_sched_barrier(0);
    const short8 x = av[i % (kAhead + 1)];
    // image row r feeds output row r - 1 through the dy = +1 tap, row r through dy = 0, row r + 1 through dy = -1
    if (r - 1 >= T0 && r - 1 < T0 + NT) {
      acc[r - 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[wc][2][0], x, acc[r - 1][0], 0, 0, 0);
      acc[r - 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[wc][2][1], x, acc[r - 1][1], 0, 0, 0);
    }
    if (r >= T0 && r < T0 + NT) {
      acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[wc][1][0], x, acc[r][0], 0, 0, 0);
      acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[wc][1][1], x, acc[r][1], 0, 0, 0);
    }
    if (r + 1 >= T0 && r + 1 < T0 + NT) {
      acc[r + 1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[wc][0][0], x, acc[r + 1][0], 0, 0, 0);
      acc[r + 1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[wc][0][1], x, acc[r + 1][1], 0, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#undef ROWCONV_WSTEP
#undef ROWCONV_ROW
}
}  // namespace

// DBG != 0 instantiations are roofline probes (bench/conv_probe.py), bit mask: 1 no LDS fragment reads,
// 2 no weight-fragment loads, 4 no main loop, 16 no input staging loads, 32 no write-back phase, 64 no first-layer
// evaluation while staging (FIRST instantiations).
// FIRST: the input tile is not read from memory but computed from the raw board (first conv layer, Cin_first = 1),
// which removes a separate memory-bound kernel and the round trip of its output.
// PAIR: a whole residual block (alphatriangle/nn/model.py:40-59) per launch.  The first convolution's activated
// output goes from the accumulators straight back into the halo tile (its lanes hold (pixel, 8 channels) chunks,
// which is the tile's row layout; the halo stays zero), the second convolution runs from there and takes the skip
// connection from the block input in memory.  Saves the write-back phase of one convolution, the staging phase
// of the other and a kernel boundary per block.
// PAIR with CIN2 != CIN: two consecutive convolutions of different widths (the conv stem: CIN -> CIN2 -> Cout); the
// first one's output goes into a second halo tile of CIN2 channels instead of back in place.
template <int CIN, int DBG, bool FIRST = false, bool PAIR = false, int CIN2 = CIN>
__global__ __launch_bounds__(256, 3) void rowconv_kernel(RowConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.x;
  if (b >= live_rows(a.live, a.B)) return;  // whole workgroup, before any barrier
  // row pitch = Cin*2 + 32 bytes = an even, non-multiple-of-4 count of 16-byte slots: the 16 lanes that one
  // ds_read_b128 cycle serves (8 rows at slot q, 8 rows at slot q+1) then hit 16 different bank quads for every tap
  constexpr int stride = CIN * 2 + 32;
  constexpr int stride2 = CIN2 * 2 + 32;
  const int H = a.H, W = a.W;
  unsigned char* tile2 = smem + (CIN2 != CIN ? a.tile2_off : 0);
  if (PAIR && CIN2 != CIN) {  // the second tile's halo has to be zero; its interior is written by the first convolution
    for (int c = tid; c < kInRows * stride2 / 16; c += 256) reinterpret_cast<uint4*>(tile2)[c] = make_uint4(0, 0, 0, 0);
  }

  // ---- stage the sample with its zero halo; eight global loads in flight per thread ----
  if (FIRST) {
    // board plane, first-layer weights and bias go to a side area of LDS, then every thread evaluates the
    // 3x3 single-channel conv for its (pixel, 8-channel) chunks straight into the halo tile
    float* aux = reinterpret_cast<float*>(smem + a.aux_off);
    float* gplane = aux;                   // [(H + 2) * (W + 2)] board plane with a zero border: taps need no bounds checks
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
    constexpr int CPR = CIN / 8;
    for (int c = tid; c < kInRows * CPR; c += 256) {
      const int r = c / CPR, cc = c - r * CPR;
      const int rr = r - 17, y = rr >> 4, x = rr & 15;
      short8 o = (short8){0, 0, 0, 0, 0, 0, 0, 0};
      if (!(DBG & 64) && rr >= 0 && y < H && x < W) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bl[cc * 8 + j];
        const float* gp = gplane + y * PW + x;  // top-left tap of this pixel in the padded plane
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float gv = gp[(tap / 3) * PW + tap % 3];
          const float4* wq = reinterpret_cast<const float4*>(wl + tap * CIN + cc * 8);
          const float4 w0 = wq[0], w1 = wq[1];
          v[0] += gv * w0.x; v[1] += gv * w0.y; v[2] += gv * w0.z; v[3] += gv * w0.w;
          v[4] += gv * w1.x; v[5] += gv * w1.y; v[6] += gv * w1.z; v[7] += gv * w1.w;
        }
        if (a.first_act == kActReLU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j], a.first_act);
        }
        o = pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      }
      *reinterpret_cast<short8*>(smem + r * stride + cc * 16) = o;
    }
  } else {
    const int cmask = (1 << a.log2_cpr) - 1;
    const int total = kInRows << a.log2_cpr;
    const unsigned short* src = a.in + (size_t)b * a.HW * a.Cin;
    for (int c0 = tid; c0 < total; c0 += 256 * 8) {
      uint4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + i * 256;
        v[i] = make_uint4(0, 0, 0, 0);
        const int rr = (c >> a.log2_cpr) - 17, y = rr >> 4, x = rr & 15;
        if (!(DBG & 16) && c < total && rr >= 0 && y < H && x < W)
          v[i] = *reinterpret_cast<const uint4*>(src + (size_t)(y * W + x) * a.Cin + (c & cmask) * 8);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + i * 256;
        if (c < total) *reinterpret_cast<uint4*>(smem + (c >> a.log2_cpr) * stride + (c & cmask) * 16) = v[i];
      }
    }
  }
  __syncthreads();

  // Fragment addresses are lane_base + a compile-time offset after unrolling: every ds_read of the main loop
  // uses an immediate offset.
  const unsigned char* lane_base = smem + l15 * stride + q4 * 16;
  const unsigned char* lane_base2 = tile2 + l15 * stride2 + q4 * 16;
  unsigned char* otile = smem + a.out_off;

  // A 64-channel first convolution of a pair (the conv stem's middle layer) would occupy two of the four waves: it is
  // split by image rows as well, wave -> (channel block wave & 1, rows 4 (wave >> 1) .. + 3).
  constexpr bool kRowSplit = PAIR && CIN2 != CIN && CIN2 == 64;
  for (int nb = 0; nb < a.n_blocks; ++nb) {
    const int ch_wave = nb * 128 + wave * 32;
    float4v acc[8][2];
    // same-width pair: one copy of the unrolled main loop, run twice; different widths: two straight-line stages
#pragma unroll(CIN2 != CIN ? 2 : 1)
    for (int stage = 0; stage < (PAIR ? 2 : 1); ++stage) {
    const unsigned short* wsrc = (PAIR && stage == 1) ? a.w2 : a.w;
    const float* bsrc = (PAIR && stage == 1) ? a.bias2 : a.bias;
    // wave-uniform; the first convolution of a pair has CIN2 output channels
    const bool active = (PAIR && stage == 0) ? (kRowSplit || wave * 32 < CIN2) : (ch_wave < a.Cout);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      acc[t][0] = (float4v){0.f, 0.f, 0.f, 0.f};
      acc[t][1] = (float4v){0.f, 0.f, 0.f, 0.f};
    }
    if (active && !(DBG & 4)) {
      if (PAIR && stage == 1 && CIN2 != CIN)
        rowconv_mainloop<CIN2, DBG>(acc, wsrc, nb, lane_base2, wave, lane);
      else if (kRowSplit && stage == 0 && wave < 2)
        rowconv_mainloop<CIN, DBG, 0, 4>(acc, wsrc, nb, lane_base, wave & 1, lane);
      else if (kRowSplit && stage == 0)
        rowconv_mainloop<CIN, DBG, 4, 4>(acc, wsrc, nb, lane_base, wave & 1, lane);
      else
        rowconv_mainloop<CIN, DBG>(acc, wsrc, nb, lane_base, wave, lane);
    }
    __syncthreads();  // every wave is done reading the input tile (the output tile may alias it)
    const int cw = (kRowSplit && stage == 0) ? (wave & 1) : wave;  // channel block of this wave in this stage
    float bias[8];
    if (active) {
#pragma unroll
      for (int i = 0; i < 8; ++i) bias[i] = bsrc[nb * 128 + cw * 32 + q4 * 8 + i];
    }
    if (PAIR && stage == 0) {
      // first convolution of the block: bias + activation, back into the halo tile as the second one's input
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
          if (a.act == kActReLU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              lo[j] = fmaxf(lo[j], 0.f);
              hi[j] = fmaxf(hi[j], 0.f);
            }
          } else if (a.act != kActNone) {
            lo = act8_slow_rc(lo, a.act);
            hi = act8_slow_rc(hi, a.act);
          }
          *reinterpret_cast<short8*>(tile2 + (17 + t * 16 + l15) * stride2 + cw * 64 + q4 * 16) =
              pack_bf16x8(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
        }
      }
      __syncthreads();
      continue;
    }
    if (active) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        *reinterpret_cast<short8*>(otile + (t * 16 + l15) * kOutStride + wave * 64 + q4 * 16) =
            pack_bf16x8(acc[t][0][0] + bias[0], acc[t][0][1] + bias[1], acc[t][0][2] + bias[2], acc[t][0][3] + bias[3],
                        acc[t][1][0] + bias[4], acc[t][1][1] + bias[5], acc[t][1][2] + bias[6], acc[t][1][3] + bias[7]);
      }
    }
    __syncthreads();
    }  // stage

    // ---- coalesced write-back: thread -> (pixel, 8-channel chunk), whole rows per wave ----
    const int n_ch = min(128, a.Cout - nb * 128);
    int l2c = 0;
    while ((8 << l2c) < n_ch) ++l2c;  // chunks per pixel = n_ch / 8 (power of two)
    const int chunks = a.HW << l2c;
    const size_t gbase = (size_t)b * a.HW * a.Cout + nb * 128;
    for (int c0 = tid; c0 < (((DBG & 32) && acc[0][0][0] != 12345.f) ? 0 : chunks); c0 += 256 * 8) {
      short8 rv[8];
      if (a.residual != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int c = c0 + i * 256;
          if (c < chunks)
            rv[i] = *reinterpret_cast<const short8*>(a.residual + gbase + (size_t)(c >> l2c) * a.Cout + (c & ((1 << l2c) - 1)) * 8);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + i * 256;
        if (c >= chunks) continue;
        const int p = c >> l2c, cc = c & ((1 << l2c) - 1);
        const int y = (p * a.ymagic) >> 16, x = p - y * W;
        const short8 hv = *reinterpret_cast<const short8*>(otile + (y * 16 + x) * kOutStride + cc * 16);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf16_to_float((unsigned short)hv[j]);
        if (a.residual != nullptr) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += bf16_to_float((unsigned short)rv[i][j]);
        }
        if (a.act == kActReLU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        } else if (a.act != kActNone) {
          const float4v lo = act8_slow_rc((float4v){v[0], v[1], v[2], v[3]}, a.act);
          const float4v hi = act8_slow_rc((float4v){v[4], v[5], v[6], v[7]}, a.act);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = lo[j];
            v[4 + j] = hi[j];
          }
        }
        if (a.post_add != nullptr) {
          const float4* pa = reinterpret_cast<const float4*>(a.post_add + (size_t)p * a.Cout + nb * 128 + cc * 8);
          const float4 p0 = pa[0], p1 = pa[1];
          v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
          v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
        }
        *reinterpret_cast<short8*>(a.out + gbase + (size_t)p * a.Cout + cc * 8) =
            pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      }
    }
  }
}

extern "C" {

// Whether atri_rowconv accepts this geometry (callers fall back to atri_tile_gemm otherwise).
int atri_rowconv_supported(int H, int W, int Cin, int Cout) {
  if (H < 1 || H > 8 || W < 1 || W > 15) return 0;
  if (Cin < 32 || Cin > 128 || (Cin & (Cin - 1)) != 0) return 0;
  if (Cout < 32 || (Cout & (Cout - 1)) != 0) return 0;
  return 1;
}

}  // extern "C"

namespace {
template <int CIN, int DBG, bool FIRST, bool PAIR = false, int CIN2 = CIN>
int launch_rowconv_inst(const RowConvArgs& a, int lds, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)rowconv_kernel<CIN, DBG, FIRST, PAIR, CIN2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((rowconv_kernel<CIN, DBG, FIRST, PAIR, CIN2>), dim3(a.B), dim3(256), lds, stream, a);
  return hip_err(hipGetLastError());
}

template <int DBG>
int launch_rowconv(const RowConvArgs& a, int lds, hipStream_t stream) {
  if (a.Cin == 32) return launch_rowconv_inst<32, DBG, false>(a, lds, stream);
  if (a.Cin == 64) return launch_rowconv_inst<64, DBG, false>(a, lds, stream);
  return launch_rowconv_inst<128, DBG, false>(a, lds, stream);
}
}  // namespace

static int rowconv_impl(const void* in, const void* w, const float* bias, const void* residual, const float* post_add,
                        void* out, int B, int H, int W, int Cin, int Cout, int act, int n_blocks, int dbg,
                        hipStream_t stream) {
  if (B <= 0) return 0;
  if (!atri_rowconv_supported(H, W, Cin, Cout)) return -10;
  if (n_blocks * 128 < Cout) return -11;
  RowConvArgs a{};
  a.in = (const unsigned short*)in;
  a.w = (const unsigned short*)w;
  a.bias = bias;
  a.residual = (const unsigned short*)residual;
  a.post_add = post_add;
  a.out = (unsigned short*)out;
  a.B = B;
  a.live = live_count_ptr();
  a.H = H;
  a.W = W;
  a.HW = H * W;
  a.Cin = Cin;
  a.Cout = Cout;
  a.act = act;
  a.n_blocks = n_blocks;
  int l2 = 0;
  while ((1 << l2) < (Cin >> 3)) ++l2;
  a.log2_cpr = l2;
  const int stride = Cin * 2 + 32;
  const int in_bytes = kInRows * stride;
  const int out_bytes = 128 * kOutStride;
  // one output block may reuse the input tile's LDS; several blocks need the input to survive
  a.out_off = (n_blocks > 1) ? in_bytes : 0;
  const int lds = (n_blocks > 1) ? in_bytes + out_bytes : (in_bytes > out_bytes ? in_bytes : out_bytes);
  a.ymagic = 65536 / W + 1;
  for (int p = 0; p < a.HW; ++p)
    if (((p * a.ymagic) >> 16) != p / W) return -12;
  switch (dbg) {
    case 0: return launch_rowconv<0>(a, lds, stream);
    case 1: return launch_rowconv<1>(a, lds, stream);
    case 2: return launch_rowconv<2>(a, lds, stream);
    case 4: return launch_rowconv<4>(a, lds, stream);
    case 20: return launch_rowconv<20>(a, lds, stream);
    case 36: return launch_rowconv<36>(a, lds, stream);
    case 48: return launch_rowconv<48>(a, lds, stream);
    case 52: return launch_rowconv<52>(a, lds, stream);
    default: return -14;
  }
}

extern "C" {

int atri_rowconv(const void* in, const void* w, const float* bias, const void* residual, const float* post_add, void* out,
                 int B, int H, int W, int Cin, int Cout, int act, int n_blocks, hipStream_t stream) {
  return rowconv_impl(in, w, bias, residual, post_add, out, B, H, W, Cin, Cout, act, n_blocks, 0, stream);
}

// First conv layer (1 input channel) fused into the staging of the second one: grid f32 [B, HW] -> out [B, HW, Cout].
// w_first [9][Cin] / b_first [Cin] are the first layer's folded weights, (w, bias) the second layer's.
int atri_rowconv_first(const float* grid, const float* w_first, const float* b_first, int first_act, const void* w,
                       const float* bias, void* out, int B, int H, int W, int Cin, int Cout, int act, int n_blocks,
                       hipStream_t stream) {
  if (B <= 0) return 0;
  if (!atri_rowconv_supported(H, W, Cin, Cout) || H * W > 128) return -10;
  if (n_blocks * 128 < Cout) return -11;
  RowConvArgs a{};
  a.w = (const unsigned short*)w;
  a.bias = bias;
  a.out = (unsigned short*)out;
  a.B = B;
  a.live = live_count_ptr();
  a.H = H;
  a.W = W;
  a.HW = H * W;
  a.Cin = Cin;
  a.Cout = Cout;
  a.act = act;
  a.n_blocks = n_blocks;
  a.grid = grid;
  a.w_first = w_first;
  a.b_first = b_first;
  a.first_act = first_act;
  const int stride = Cin * 2 + 32;
  const int in_bytes = kInRows * stride;
  const int out_bytes = 128 * kOutStride;
  a.out_off = (n_blocks > 1) ? in_bytes : 0;
  int lds = (n_blocks > 1) ? in_bytes + out_bytes : (in_bytes > out_bytes ? in_bytes : out_bytes);
  a.aux_off = (lds + 15) & ~15;              // board plane + first-layer weights live behind the tiles
  lds = a.aux_off + (192 + 10 * Cin) * 4;
  a.ymagic = 65536 / W + 1;
  for (int p = 0; p < a.HW; ++p)
    if (((p * a.ymagic) >> 16) != p / W) return -12;
  if (Cin == 32) return launch_rowconv_inst<32, 0, true>(a, lds, stream);
  if (Cin == 64) return launch_rowconv_inst<64, 0, true>(a, lds, stream);
  return launch_rowconv_inst<128, 0, true>(a, lds, stream);
}

// Residual block in one launch: out = act(conv2(act(conv1(in))) + in) (+ post_add), 128 channels throughout.
// (w1, b1) / (w2, b2) are the two folded convolutions; the skip connection is read from `in`.
int atri_rowconv_pair(const void* in, const void* w1, const float* b1, const void* w2, const float* b2,
                      const float* post_add, void* out, int B, int H, int W, int C, int act, hipStream_t stream) {
  if (B <= 0) return 0;
  if (C != 128 || !atri_rowconv_supported(H, W, C, C)) return -10;
  if (in == out) return -13;  // the skip connection is read after other workgroups may have written their output
  RowConvArgs a{};
  a.in = (const unsigned short*)in;
  a.w = (const unsigned short*)w1;
  a.bias = b1;
  a.w2 = (const unsigned short*)w2;
  a.bias2 = b2;
  a.residual = (const unsigned short*)in;
  a.post_add = post_add;
  a.out = (unsigned short*)out;
  a.B = B;
  a.live = live_count_ptr();
  a.H = H;
  a.W = W;
  a.HW = H * W;
  a.Cin = C;
  a.Cout = C;
  a.act = act;
  a.n_blocks = 1;
  a.log2_cpr = 4;  // 16 chunks of 8 channels per pixel
  const int in_bytes = kInRows * (C * 2 + 32);
  const int out_bytes = 128 * kOutStride;
  a.out_off = 0;
  const int lds = in_bytes > out_bytes ? in_bytes : out_bytes;
  a.ymagic = 65536 / W + 1;
  for (int p = 0; p < a.HW; ++p)
    if (((p * a.ymagic) >> 16) != p / W) return -12;
  return launch_rowconv_inst<128, 0, false, true>(a, lds, stream);
}

// Conv stem in one launch: board plane -> conv(1 -> C0) -> conv(C0 -> C1) -> conv(C1 -> C2), every layer followed by
// the activation; only the last layer's output goes to memory.  (C0, C1, C2) = (32, 64, 128).
// w_first [9][C0] / b_first [C0]: first layer (evaluated while staging); (w1, b1), (w2, b2): packed second and third layer.
static int rowconv_stem_impl(const float* grid, const float* w_first, const float* b_first, const void* w1, const float* b1,
                             const void* w2, const float* b2, const float* post_add, void* out, int B, int H, int W, int C0,
                             int C1, int C2, int act, int dbg, hipStream_t stream) {
  if (B <= 0) return 0;
  if (C0 != 32 || C1 != 64 || C2 != 128 || !atri_rowconv_supported(H, W, C0, C1) || H * W > 128) return -10;
  RowConvArgs a{};
  a.w = (const unsigned short*)w1;
  a.bias = b1;
  a.w2 = (const unsigned short*)w2;
  a.bias2 = b2;
  a.post_add = post_add;
  a.out = (unsigned short*)out;
  a.B = B;
  a.live = live_count_ptr();
  a.H = H;
  a.W = W;
  a.HW = H * W;
  a.Cin = C0;
  a.Cout = C2;
  a.act = act;
  a.n_blocks = 1;
  a.grid = grid;
  a.w_first = w_first;
  a.b_first = b_first;
  a.first_act = act;
  const int tile1 = kInRows * (C0 * 2 + 32), tile2 = kInRows * (C1 * 2 + 32);
  const int out_bytes = 128 * kOutStride;
  a.tile2_off = (tile1 + 15) & ~15;
  a.out_off = 0;  // the output staging tile reuses both input tiles once the last main loop is done
  int lds = a.tile2_off + tile2 > out_bytes ? a.tile2_off + tile2 : out_bytes;
  a.aux_off = (lds + 15) & ~15;
  lds = a.aux_off + (192 + 10 * C0) * 4;
  a.ymagic = 65536 / W + 1;
  for (int p = 0; p < a.HW; ++p)
    if (((p * a.ymagic) >> 16) != p / W) return -12;
  switch (dbg) {
    case 0: return launch_rowconv_inst<32, 0, true, true, 64>(a, lds, stream);
    case 4: return launch_rowconv_inst<32, 4, true, true, 64>(a, lds, stream);
    case 32: return launch_rowconv_inst<32, 32, true, true, 64>(a, lds, stream);
    case 36: return launch_rowconv_inst<32, 36, true, true, 64>(a, lds, stream);
    case 64: return launch_rowconv_inst<32, 64, true, true, 64>(a, lds, stream);
    case 100: return launch_rowconv_inst<32, 100, true, true, 64>(a, lds, stream);
    default: return -14;
  }
}

int atri_rowconv_stem(const float* grid, const float* w_first, const float* b_first, const void* w1, const float* b1,
                      const void* w2, const float* b2, const float* post_add, void* out, int B, int H, int W, int C0, int C1,
                      int C2, int act, hipStream_t stream) {
  return rowconv_stem_impl(grid, w_first, b_first, w1, b1, w2, b2, post_add, out, B, H, W, C0, C1, C2, act, 0, stream);
}

// Roofline probe of the conv stem (see the kernel's DBG parameter; 100 = 4 + 32 + 64: launch, tile zeroing and barriers only).
int atri_rowconv_stem_probe(const float* grid, const float* w_first, const float* b_first, const void* w1, const float* b1,
                            const void* w2, const float* b2, void* out, int B, int H, int W, int act, int dbg,
                            hipStream_t stream) {
  return rowconv_stem_impl(grid, w_first, b_first, w1, b1, w2, b2, nullptr, out, B, H, W, 32, 64, 128, act, dbg, stream);
}

// Roofline probe: the same launch with ingredients removed (see the kernel's DBG parameter).
int atri_rowconv_probe(const void* in, const void* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                       int n_blocks, int dbg, hipStream_t stream) {
  return rowconv_impl(in, w, bias, nullptr, nullptr, out, B, H, W, Cin, Cout, kActReLU, n_blocks, dbg, stream);
}

}  // extern "C"
