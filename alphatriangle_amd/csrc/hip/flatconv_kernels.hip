// 3x3 / stride 1 / pad 1 convolution as an MFMA implicit GEMM for boards that do not fit the 8x15 halo tile of
// conv_kernels.hip (the 12x21 grid of the large preset; any board with H * (W + 1) <= 320), NHWC bf16 in and out,
// one workgroup of eight waves per sample.
//
// Layers served: conv stem layers 1.. and both convs of every residual block
// (alphatriangle/nn/model.py:20-37 conv_block, :40-59 ResidualBlock), BatchNorm folded.
//
//   * The sample sits in LDS as ONE flat run of pixels at a row pitch of P = W + 1: pixel (y, x) is LDS row
//     y * P + x + P + 1.  The single spare column per image row is zero and serves as the right halo of its own row
//     and as the left halo of the next one; P + 1 zero rows in front and behind are the top and bottom halo.  A 3x3
//     tap is then a constant row offset (dy * P + dx) for EVERY pixel, so an MFMA tile is simply 16 consecutive flat
//     positions: 12 x 22 = 264 positions make 17 tiles, against 24 tiles if every image row were padded to 32.
//   * wave -> (64 of the 128 output channels, a quarter of the tiles): one 16-byte LDS fragment read feeds four
//     MFMAs, which keeps the LDS pipe at half of what the matrix pipe asks for with eight waves on the CU.  Tile
//     counts that are no multiple of four are dealt out unevenly (17 tiles = 5 + 4 + 4 + 4): waves with one tile
//     less run their own instantiation of the convolution code and meet the others at the same barriers.
//   * Residual blocks run both convolutions in one launch; the first one's activated output goes from the
//     accumulators back into the same flat tile (spare column and overhang written as zeros), and the skip
//     connection stays in the registers that staged the block input (staging and write-back use one thread mapping).
//   * The conv stem is one launch as well: the single-channel first layer is evaluated while staging, the
//     32 -> 64 -> 128 pair keeps its 64-channel intermediate in a second flat tile.
//   * The accumulators leave through an LDS transpose and whole 256-byte rows, as in conv_kernels.hip.
#include "common.hpp"

using namespace atrihip;

struct FlatConvArgs {
  const unsigned short* in;        // [B, HW, Cin] bf16
  const unsigned short* w;         // packed fragments [1][9*KC][wave 4][n 2][lane 64][8] (pack_gemm_weights)
  const float* bias;               // [128]
  const unsigned short* residual;  // [B, HW, Cout] bf16 or nullptr (added before the activation)
  const float* post_add;           // [HW, Cout] f32 or nullptr (added after the activation)
  unsigned short* out;             // [B, HW, Cout] bf16
  const unsigned short* w2;        // PAIR: second convolution
  const float* bias2;
  int B, H, W, P, HW, Cin, Cout, act, log2_cpr, rows, pmagic, wmagic, full_groups;
  const int* live;                 // device row count (nullptr: B rows), see common.hpp
  long long* phase_clock;          // diagnostics (PROBE instantiation): workgroup `phase_block` stores the shader clock
  int phase_block;                 // at its phase boundaries
  // FIRST instantiation only (conv stem in one launch): the network's first layer (1 input channel, 3x3) is evaluated
  // while staging, its output feeds the pair CIN -> CIN2 -> Cout whose intermediate lives in a second flat tile
  const float* grid;     // [B, HW] f32 board planes
  const float* w_first;  // [9][CIN] f32 (BatchNorm folded)
  const float* b_first;  // [CIN]
  int tile2_off, aux_off, plane_w;
};

namespace {
constexpr int kFlatOutStride = 272;  // bytes per flat position of the 128-channel output staging tile
constexpr int kFlatAhead = 2;        // LDS fragments requested ahead of the MFMAs that consume them
constexpr int kFlatThreads = 512;

__device__ __noinline__ float4v act4_slow_fc(float4v v, int act) {
  for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], act);
  return v;
}

// One 3x3 convolution of this wave's 64 output channels (packed channel groups 2h and 2h + 1) over its TPW tiles.
// Steps run in the packed order s = tap * KC + kc; the four weight fragments of a step are requested two steps ahead
// (one step is TPW * 4 MFMAs, shorter than an L2 round trip when the partner wave on the SIMD is stalled as well).
template <int CIN, int TPW>
__device__ __forceinline__ void flatconv_mainloop(float4v (&acc)[TPW][4], const unsigned short* __restrict__ wsrc,
                                                  const unsigned char* tile_base, int P, int h, int lane) {
  constexpr int stride = CIN * 2 + 32;
  constexpr int KC = CIN / 32, STEPS = 9 * KC, TOTAL = STEPS * TPW;
  const short8* wp = reinterpret_cast<const short8*>(wsrc) + (size_t)(2 * h) * 128 + lane;
#define FLATCONV_ROW(i) \
  (tile_base + ((((i) / TPW / KC) / 3) * P + (((i) / TPW / KC) % 3) + ((i) % TPW) * 16) * stride + (((i) / TPW) % KC) * 64)
  short8 wg[3][4], av[kFlatAhead + 1];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    if (d < STEPS) {
#pragma unroll
      for (int k = 0; k < 4; ++k) wg[d][k] = wp[(size_t)d * 512 + k * 64];
    }
  }
#pragma unroll
  for (int i = 0; i < kFlatAhead; ++i) av[i] = *reinterpret_cast<const short8*>(FLATCONV_ROW(i));
  // unroll(full), not `#pragma unroll`: the latter left this loop half rolled (90 iterations x 2 trips for 180), and
  // every fragment address was then recomputed from the loop counter by ~18 scalar instructions per four MFMAs
#pragma clang loop unroll(full)
  for (int i = 0; i < TOTAL; ++i) {
    const int s = i / TPW, t = i % TPW, wc = s % 3;
    __builtin_amdgcn_sched_barrier(0);
    if (t == 0 && s + 2 < STEPS) {
#pragma unroll
      for (int k = 0; k < 4; ++k) wg[(s + 2) % 3][k] = wp[(size_t)(s + 2) * 512 + k * 64];
    }
    if (i + kFlatAhead < TOTAL)
      av[(i + kFlatAhead) % (kFlatAhead + 1)] = *reinterpret_cast<const short8*>(FLATCONV_ROW(i + kFlatAhead));
    __builtin_amdgcn_sched_barrier(0);
    const short8 x = av[i % (kFlatAhead + 1)];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[t][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[wc][k], x, acc[t][k], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
#undef FLATCONV_ROW
}
}  // namespace

#define FLAT_PHASE(i)                                                                                  \
  if constexpr (PROBE) {                                                                               \
    if ((int)blockIdx.x == a.phase_block && threadIdx.x == 0) a.phase_clock[i] = clock64();             \
  }

// The convolution(s) of one wave over its NT tiles starting at tile0, from the staged flat tile to the output staging
// tile (every barrier of the workgroup between those two points is in here: waves with NT and NT - 1 tiles run
// different instantiations and meet at the same barrier count).
template <int CIN, int NT, bool PAIR, bool PROBE, int CIN2 = CIN>
__device__ __forceinline__ void flat_convs(const FlatConvArgs& a, unsigned char* smem, int tile0, int h, int lane) {
  constexpr int TPW = NT;
  constexpr int stride = CIN * 2 + 32, stride2 = CIN2 * 2 + 32;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int H = a.H, W = a.W, P = a.P;
  // a pair of two widths (the conv stem) keeps its intermediate in a second tile; a residual block writes it in place
  unsigned char* tile2 = smem + (CIN2 != CIN ? a.tile2_off : 0);
  const unsigned char* tile_base = smem + (tile0 * 16 + l15) * stride + q4 * 16;
  const unsigned char* tile_base2 = tile2 + (tile0 * 16 + l15) * stride2 + q4 * 16;
  float4v acc[TPW][4];
  // same width: one copy of the unrolled main loop, run twice (two copies of it only cost instruction-cache misses);
  // two widths: two straight-line stages
#pragma unroll(CIN2 != CIN ? 2 : 1)
  for (int stage = 0; stage < (PAIR ? 2 : 1); ++stage) {
    const unsigned short* wsrc = (PAIR && stage == 1) ? a.w2 : a.w;
    const float* bsrc = (PAIR && stage == 1) ? a.bias2 : a.bias;
    // wave-uniform: a 64-channel layer occupies the waves of half 0 only
    const bool active = h * 64 < ((PAIR && stage == 0 && CIN2 != CIN) ? CIN2 : a.Cout);
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[t][k] = (float4v){0.f, 0.f, 0.f, 0.f};
    float bias[2][8];
    if (active) {
#pragma unroll
      for (int ws = 0; ws < 2; ++ws)
#pragma unroll
        for (int i = 0; i < 8; ++i) bias[ws][i] = bsrc[(2 * h + ws) * 32 + q4 * 8 + i];
      if (PAIR && stage == 1 && CIN2 != CIN)
        flatconv_mainloop<CIN2, TPW>(acc, wsrc, tile_base2, P, h, lane);
      else
        flatconv_mainloop<CIN, TPW>(acc, wsrc, tile_base, P, h, lane);
    }
    FLAT_PHASE(3 + stage * 4);
    __syncthreads();  // every wave is done reading the tile: it is overwritten in place / reused as the output tile
    FLAT_PHASE(4 + stage * 4);
    if (PAIR && stage == 0) {
      // first convolution of the block: bias + activation back into the flat tile; positions that are no pixel
      // (the spare column, the overhang behind the last image row) are written as zeros and stay the halo
      if (active) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int f = (tile0 + t) * 16 + l15;
          const int y = (f * a.pmagic) >> 16, x = f - y * P;
          const bool pixel = y < H && x < W;
#pragma unroll
          for (int ws = 0; ws < 2; ++ws) {
            float4v lo = acc[t][ws * 2], hi = acc[t][ws * 2 + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              lo[j] += bias[ws][j];
              hi[j] += bias[ws][4 + j];
            }
            if (a.act == kActReLU) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                lo[j] = fmaxf(lo[j], 0.f);
                hi[j] = fmaxf(hi[j], 0.f);
              }
            } else if (a.act != kActNone) {
              lo = act4_slow_fc(lo, a.act);
              hi = act4_slow_fc(hi, a.act);
            }
            const short8 o = pixel ? pack_bf16x8(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3])
                                   : (short8){0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<short8*>(tile2 + (f + P + 1) * stride2 + (2 * h + ws) * 64 + q4 * 16) = o;
          }
        }
      }
      FLAT_PHASE(5);
      __syncthreads();
      FLAT_PHASE(6);
      continue;
    }
    if (active) {
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int f = (tile0 + t) * 16 + l15;
#pragma unroll
        for (int ws = 0; ws < 2; ++ws) {
          const float4v lo = acc[t][ws * 2], hi = acc[t][ws * 2 + 1];
          *reinterpret_cast<short8*>(smem + f * kFlatOutStride + (2 * h + ws) * 64 + q4 * 16) =
              pack_bf16x8(lo[0] + bias[ws][0], lo[1] + bias[ws][1], lo[2] + bias[ws][2], lo[3] + bias[ws][3],
                          hi[0] + bias[ws][4], hi[1] + bias[ws][5], hi[2] + bias[ws][6], hi[3] + bias[ws][7]);
        }
      }
    }
    FLAT_PHASE(9);
    __syncthreads();
    FLAT_PHASE(10);
  }

}

// PAIR: a whole residual block (alphatriangle/nn/model.py:40-59) per launch, 128 channels throughout; the skip
// connection stays in the registers that staged the block input.
// FIRST (with PAIR and CIN2): the conv stem in one launch - board plane -> conv(1 -> CIN) while staging ->
// conv(CIN -> CIN2) -> conv(CIN2 -> Cout); only the last layer's output goes to memory.
template <int CIN, int TPW, bool PAIR, bool PROBE = false, bool FIRST = false, int CIN2 = CIN>
__global__ __launch_bounds__(kFlatThreads, 1) void flatconv_kernel(FlatConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.x;
  if (b >= live_rows(a.live, a.B)) return;  // whole workgroup, before any barrier
  constexpr int stride = CIN * 2 + 32;
  const int H = a.H, W = a.W, P = a.P;
  FLAT_PHASE(0);

  // ---- stage the sample into the flat tile: thread -> (pixel, 8-channel chunk) as in the write-back, all loads of a
  // thread in flight at once (H * W <= 256 pixels: at most eight chunks per thread); rows that are no pixel (halo
  // runs, spare column, overhang) are zero-filled while the loads are under way.  A residual block keeps the chunks
  // in registers: they are its skip connection.
  uint4 xin[8];
  if constexpr (FIRST) {
    // board plane (with a zero border: taps need no bounds checks), first-layer weights and bias go to a side area of
    // LDS; then every thread evaluates the single-channel 3x3 conv for its (flat position, 8-channel chunk) slots
    // straight into the flat tile, zeros for the positions that are no pixel.  The second tile starts all zero.
    float* aux = reinterpret_cast<float*>(smem + a.aux_off);
    const int PW = W + 2, plane = (H + 2) * PW;
    float* gplane = aux;
    float* wl = aux + a.plane_w;  // [9][CIN]
    float* bl = wl + 9 * CIN;     // [CIN]
    for (int i = tid; i < plane; i += kFlatThreads) {
      const int yy = i / PW - 1, xx = i - (yy + 1) * PW - 1;
      gplane[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? a.grid[(size_t)b * a.HW + yy * W + xx] : 0.f;
    }
    for (int i = tid; i < 9 * CIN; i += kFlatThreads) wl[i] = a.w_first[i];
    for (int i = tid; i < CIN; i += kFlatThreads) bl[i] = a.b_first[i];
    constexpr int stride2 = CIN2 * 2 + 32;
    for (int c = tid; c < a.rows * stride2 / 16; c += kFlatThreads)
      reinterpret_cast<uint4*>(smem + a.tile2_off)[c] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    constexpr int CPR = CIN / 8;
    for (int c = tid; c < a.rows * CPR; c += kFlatThreads) {
      const int r = c / CPR, cc = c - r * CPR;
      const int rr = r - (P + 1);
      const int y = (rr * a.pmagic) >> 16, x = rr - y * P;
      short8 o = (short8){0, 0, 0, 0, 0, 0, 0, 0};
      if (rr >= 0 && y < H && x < W) {
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
        if (a.act == kActReLU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j], a.act);
        }
        o = pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      }
      *reinterpret_cast<short8*>(smem + r * stride + cc * 16) = o;
    }
  } else {
    const int l2 = a.log2_cpr, cmask = (1 << l2) - 1;
    const int total = a.HW << l2;
    const unsigned short* src = a.in + (size_t)b * a.HW * a.Cin;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = min(tid + i * kFlatThreads, total - 1);
      xin[i] = *reinterpret_cast<const uint4*>(src + (size_t)(c >> l2) * a.Cin + (c & cmask) * 8);
    }
    for (int c = tid; c < (a.rows << l2); c += kFlatThreads) {
      const int r = c >> l2, rr = r - (P + 1);
      const int y = (rr * a.pmagic) >> 16, x = rr - y * P;
      if (!(rr >= 0 && y < H && x < W)) *reinterpret_cast<uint4*>(smem + r * stride + (c & cmask) * 16) = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + i * kFlatThreads;
      if (c < total) {
        const int pp = c >> l2;
        const int y = (pp * a.wmagic) >> 16, x = pp - y * W;
        *reinterpret_cast<uint4*>(smem + (y * P + x + P + 1) * stride + (c & cmask) * 16) = xin[i];
      }
    }
  }
  FLAT_PHASE(1);
  __syncthreads();
  FLAT_PHASE(2);

  // channel half and tile quarter of this wave; the first `full_groups` quarters hold TPW tiles, the others TPW - 1
  const int h = wave & 1, tq = wave >> 1;
  if (tq < a.full_groups)  // wave-uniform
    flat_convs<CIN, TPW, PAIR, PROBE, CIN2>(a, smem, tq * TPW, h, lane);
  else
    flat_convs<CIN, TPW - 1, PAIR, PROBE, CIN2>(a, smem, a.full_groups + tq * (TPW - 1), h, lane);

  // ---- coalesced write-back: thread -> (pixel, 8-channel chunk), whole rows per wave ----
  const int n_ch = min(128, a.Cout);
  int l2c = 0;
  while ((8 << l2c) < n_ch) ++l2c;  // chunks per pixel = n_ch / 8 (power of two)
  const int chunks = a.HW << l2c;
  const size_t gbase = (size_t)b * a.HW * a.Cout;
  for (int c0 = tid; c0 < chunks; c0 += kFlatThreads * 8) {
    short8 rv[8];
    if (PAIR && !FIRST) {  // the skip connection is the block input: same (pixel, chunk) mapping as the staging, still in registers
#pragma unroll
      for (int i = 0; i < 8; ++i) rv[i] = __builtin_bit_cast(short8, xin[i]);
    } else if (a.residual != nullptr) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = min(c0 + i * kFlatThreads, chunks - 1);
        rv[i] = *reinterpret_cast<const short8*>(a.residual + gbase + (size_t)(c >> l2c) * a.Cout + (c & ((1 << l2c) - 1)) * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i * kFlatThreads;
      if (c >= chunks) continue;
      const int p = c >> l2c, cc = c & ((1 << l2c) - 1);
      const int y = (p * a.wmagic) >> 16, x = p - y * W;
      const short8 hv = *reinterpret_cast<const short8*>(smem + (y * P + x) * kFlatOutStride + cc * 16);
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = bf16_to_float((unsigned short)hv[j]);
      if ((PAIR && !FIRST) || a.residual != nullptr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += bf16_to_float((unsigned short)rv[i][j]);
      }
      if (a.act == kActReLU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
      } else if (a.act != kActNone) {
        const float4v lo = act4_slow_fc((float4v){v[0], v[1], v[2], v[3]}, a.act);
        const float4v hi = act4_slow_fc((float4v){v[4], v[5], v[6], v[7]}, a.act);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = lo[j];
          v[4 + j] = hi[j];
        }
      }
      if (a.post_add != nullptr) {
        const float4* pa = reinterpret_cast<const float4*>(a.post_add + (size_t)p * a.Cout + cc * 8);
        const float4 p0 = pa[0], p1 = pa[1];
        v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
        v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
      }
      *reinterpret_cast<short8*>(a.out + gbase + (size_t)p * a.Cout + cc * 8) =
          pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    }
  }
  FLAT_PHASE(11);
#undef FLAT_PHASE
}

namespace {
long long* g_flat_phase_clock = nullptr;
int g_flat_phase_block = 0;

int flat_tiles(int H, int W) { return (H * (W + 1) + 15) / 16; }

// Largest tile count of a wave for a board: ceil(H * (W + 1) / 16) tiles over four tile quarters; 0 = not served.
int flat_tiles_per_wave(int H, int W) {
  if (H < 1 || W < 1 || H * W > 256) return 0;  // staging and write-back move a sample in one pass of 8 chunks per thread
  const int tpw = (flat_tiles(H, W) + 3) / 4;
  return tpw <= 3 ? 3 : (tpw <= 5 ? tpw : 0);
}

template <int CIN, int TPW, bool PAIR>
int launch_flat_inst(const FlatConvArgs& a, int lds, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)flatconv_kernel<CIN, TPW, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((flatconv_kernel<CIN, TPW, PAIR>), dim3(a.B), dim3(kFlatThreads), lds, stream, a);
  return hip_err(hipGetLastError());
}

template <int CIN, bool PAIR>
int launch_flat_tpw(const FlatConvArgs& a, int tpw, int lds, hipStream_t stream) {
  if (tpw == 3) return launch_flat_inst<CIN, 3, PAIR>(a, lds, stream);
  if (tpw == 4) return launch_flat_inst<CIN, 4, PAIR>(a, lds, stream);
  return launch_flat_inst<CIN, 5, PAIR>(a, lds, stream);
}

// Geometry fields shared by the single and the paired launch; returns the LDS bytes or a negative error code.
int flat_fill_geometry(FlatConvArgs& a, int H, int W, int Cin, int tpw) {
  a.H = H;
  a.W = W;
  a.P = W + 1;
  a.HW = H * W;
  a.Cin = Cin;
  int l2 = 0;
  while ((1 << l2) < (Cin >> 3)) ++l2;
  a.log2_cpr = l2;
  // tpw tiles for the first full_groups tile quarters, tpw - 1 for the others (a board of fewer than 4 * (tpw - 1)
  // tiles - only possible at tpw = 3 - computes a few tiles of overhang)
  const int tiles = flat_tiles(H, W) > 4 * (tpw - 1) ? flat_tiles(H, W) : 4 * (tpw - 1);
  a.full_groups = tiles - 4 * (tpw - 1);
  const int slots = 16 * tiles;     // flat positions covered by the tiles
  a.rows = slots + 2 * a.P + 2;     // + the halo runs in front and behind
  a.pmagic = 65536 / a.P + 1;
  a.wmagic = 65536 / W + 1;
  for (int r = 0; r < a.rows; ++r)
    if (((r * a.pmagic) >> 16) != r / a.P) return -12;
  for (int p = 0; p < a.HW; ++p)
    if (((p * a.wmagic) >> 16) != p / W) return -12;
  const int in_bytes = a.rows * (Cin * 2 + 32);
  const int out_bytes = slots * kFlatOutStride;
  const int lds = in_bytes > out_bytes ? in_bytes : out_bytes;
  return lds <= 160 * 1024 ? lds : -15;
}
}  // namespace

extern "C" {

// Whether atri_flatconv accepts this geometry (boards the 8x15 halo-tile kernel refuses; callers fall back to
// atri_tile_gemm otherwise).
int atri_flatconv_supported(int H, int W, int Cin, int Cout) {
  if (flat_tiles_per_wave(H, W) == 0) return 0;
  if (Cin != 32 && Cin != 64 && Cin != 128) return 0;
  if (Cout != 64 && Cout != 128) return 0;
  FlatConvArgs a{};
  return flat_fill_geometry(a, H, W, Cin, flat_tiles_per_wave(H, W)) > 0 ? 1 : 0;
}

int atri_flatconv(const void* in, const void* w, const float* bias, const void* residual, const float* post_add, void* out,
                  int B, int H, int W, int Cin, int Cout, int act, hipStream_t stream) {
  if (B <= 0) return 0;
  if (!atri_flatconv_supported(H, W, Cin, Cout)) return -10;
  FlatConvArgs a{};
  const int tpw = flat_tiles_per_wave(H, W);
  const int lds = flat_fill_geometry(a, H, W, Cin, tpw);
  if (lds < 0) return lds;
  a.in = (const unsigned short*)in;
  a.w = (const unsigned short*)w;
  a.bias = bias;
  a.residual = (const unsigned short*)residual;
  a.post_add = post_add;
  a.out = (unsigned short*)out;
  a.B = B;
  a.live = live_count_ptr();
  a.Cout = Cout;
  a.act = act;
  if (Cin == 32) return launch_flat_tpw<32, false>(a, tpw, lds, stream);
  if (Cin == 64) return launch_flat_tpw<64, false>(a, tpw, lds, stream);
  return launch_flat_tpw<128, false>(a, tpw, lds, stream);
}

// Residual block in one launch: out = act(conv2(act(conv1(in))) + in) (+ post_add), 128 channels throughout.
int atri_flatconv_pair(const void* in, const void* w1, const float* b1, const void* w2, const float* b2,
                       const float* post_add, void* out, int B, int H, int W, int C, int act, hipStream_t stream) {
  if (B <= 0) return 0;
  if (C != 128 || !atri_flatconv_supported(H, W, C, C)) return -10;
  if (in == out) return -13;  // the skip connection is read after other workgroups may have written their output
  FlatConvArgs a{};
  const int tpw = flat_tiles_per_wave(H, W);
  const int lds = flat_fill_geometry(a, H, W, C, tpw);
  if (lds < 0) return lds;
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
  a.Cout = C;
  a.act = act;
  if (g_flat_phase_clock != nullptr && tpw == 5) {  // diagnostics build of the kernel
    a.phase_clock = g_flat_phase_clock;
    a.phase_block = g_flat_phase_block;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)flatconv_kernel<128, 5, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL((flatconv_kernel<128, 5, true, true>), dim3(a.B), dim3(kFlatThreads), lds, stream, a);
    return hip_err(hipGetLastError());
  }
  return launch_flat_tpw<128, true>(a, tpw, lds, stream);
}

// Conv stem in one launch: board plane -> conv(1 -> C0) -> conv(C0 -> C1) -> conv(C1 -> C2), every layer followed by
// the activation; only the last layer's output goes to memory.  (C0, C1, C2) = (32, 64, 128).
// w_first [9][C0] / b_first [C0]: first layer (evaluated while staging); (w1, b1), (w2, b2): packed second and third layer.
int atri_flatconv_stem(const float* grid, const float* w_first, const float* b_first, const void* w1, const float* b1,
                       const void* w2, const float* b2, const float* post_add, void* out, int B, int H, int W, int C0, int C1,
                       int C2, int act, hipStream_t stream) {
  if (B <= 0) return 0;
  if (C0 != 32 || C1 != 64 || C2 != 128 || !atri_flatconv_supported(H, W, C0, C1) || !atri_flatconv_supported(H, W, C1, C2)) return -10;
  FlatConvArgs a{};
  const int tpw = flat_tiles_per_wave(H, W);
  if (flat_fill_geometry(a, H, W, C0, tpw) < 0) return -15;
  a.w = (const unsigned short*)w1;
  a.bias = b1;
  a.w2 = (const unsigned short*)w2;
  a.bias2 = b2;
  a.post_add = post_add;
  a.out = (unsigned short*)out;
  a.B = B;
  a.live = live_count_ptr();
  a.Cout = C2;
  a.act = act;
  a.grid = grid;
  a.w_first = w_first;
  a.b_first = b_first;
  const int tile1 = a.rows * (C0 * 2 + 32), tile2 = a.rows * (C1 * 2 + 32);
  const int out_bytes = (a.rows - 2 * a.P - 2) * kFlatOutStride;
  a.tile2_off = (tile1 + 15) & ~15;
  int lds = a.tile2_off + tile2 > out_bytes ? a.tile2_off + tile2 : out_bytes;  // the output tile reuses both input tiles
  a.aux_off = (lds + 15) & ~15;                                                  // board plane + first-layer weights behind them
  a.plane_w = ((H + 2) * (W + 2) + 3) & ~3;
  lds = a.aux_off + (a.plane_w + 10 * C0) * 4;
  if (lds > 160 * 1024) return -15;
#define ATRI_FLAT_STEM(T)                                                                                                          \
  {                                                                                                                               \
    static bool attr = false;                                                                                                     \
    if (!attr) {                                                                                                                  \
      (void)hipFuncSetAttribute((const void*)flatconv_kernel<32, T, true, false, true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
      attr = true;                                                                                                                \
    }                                                                                                                             \
    hipLaunchKernelGGL((flatconv_kernel<32, T, true, false, true, 64>), dim3(B), dim3(kFlatThreads), lds, stream, a);             \
  }
  if (tpw == 3) ATRI_FLAT_STEM(3) else if (tpw == 4) ATRI_FLAT_STEM(4) else ATRI_FLAT_STEM(5)
#undef ATRI_FLAT_STEM
  return hip_err(hipGetLastError());
}

// Diagnostics: while `clock` (12 x int64 of device memory) is set, atri_flatconv_pair launches on 5-tile boards run
// the instantiation in which workgroup `block` stores the shader clock at its phase boundaries: 0 start, 1 staged,
// 2 barrier, 3 first main loop done, 4 barrier, 5 intermediate written, 6 barrier, 7 second main loop done, 8 barrier,
// 9 output tile written, 10 barrier, 11 written back.  nullptr switches it off.
int atri_flatconv_phase_clock(long long* clock, int block) {
  g_flat_phase_clock = clock;
  g_flat_phase_block = block;
  return 0;
}

}  // extern "C"
