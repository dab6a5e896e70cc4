// fp32-grade 3x3 convolutions for boards that do not fit the 8x15 halo tile of split_conv_kernels.hip (the 12x21 grid of
// the large preset; any board with 16 * ceil(H * (W + 1) / 16) + 2 * W + 4 <= 320 flat rows): fp32 activations in memory,
// every GEMM as three 16-bit MFMA passes over (hi, lo) operand splits with fp32 accumulation (the scheme of
// split_conv_kernels.hip), on the flat-tile geometry of flatconv_kernels.hip.
//
// Layers served: the conv stem and both convolutions of every residual block
// (alphatriangle/nn/model.py:20-37 conv_block, :40-59 ResidualBlock), BatchNorm folded; the reference evaluates them in
// fp32 on any grid (alphatriangle/nn/model.py:123-125,268-297, tests/nn/test_model.py:80-81).
//
//   * The sample sits in LDS as ONE flat run of pixels at a row pitch of P = W + 1 (pixel (y, x) is LDS row
//     y * P + x + P + 1; the spare column and P + 1 rows in front and behind are the zero halo), so a 3x3 tap is a
//     constant row offset and an MFMA tile is 16 consecutive flat positions.
//   * An LDS row holds both planes of a position: [hi: C bf16][lo: C bf16].  For 32 and 64 channels the row is padded by
//     32 bytes (fragment reads of 16 rows then hit 16 different bank quads).  A 128-channel row is exactly 512 bytes -
//     318 rows of a 12x21 board are 159 KB, which leaves no room for padding inside the 160 KB of a CU - and its
//     sixteen 16-byte slots per plane are XOR-swizzled with the row index instead (slot ^ (row & 15)).
//   * wave -> (64 of the 128 output channels, a quarter of the tiles); uneven tile counts are dealt out as in
//     flatconv_kernels.hip (waves with one tile less run their own instantiation and meet the others at the same
//     barriers).  The loop over the nine taps stays rolled: one tap is 12 MFMAs x KC channel chunks x TPW tiles, long
//     enough to hide the loop overhead, and the code of both instantiations then fits the instruction cache.  The
//     weight fragments (hi and lo streams) of the next K-step are requested one step ahead - a step is 12 * TPW MFMAs,
//     longer than an L2 round trip - and the LDS fragments one tile ahead.
//   * A residual block runs both convolutions in one launch: the first one's activated output is split and written back
//     into the same tile; the skip connection is re-read from memory while the workgroup waits at its last barrier.
//   * The conv stem is one launch as well: the single-channel first layer is evaluated while staging; the 32 -> 64
//     layer runs with all eight waves (wave -> 32 channels) into a second tile, the 64 -> 128 layer from there.
//   * Accumulators leave through an fp32 staging tile in LDS so that global stores, the skip connection and the
//     positional table move as whole 512-byte pixel rows.
#include "common.hpp"

using namespace atrihip;

struct SplitFlatArgs {
  const float* in;          // [B, HW, 128] f32 (residual block)
  const unsigned short* w;  // first convolution: packed 16-bit fragments, hi stream then lo stream (lo_off elements apart)
  const float* bias;
  const unsigned short* w2; // second convolution
  const float* bias2;
  long long lo_off, lo_off2;
  const float* post_add;    // [HW, 128] f32 or nullptr (added after the activation)
  float* out;               // [B, HW, 128] f32
  int B, H, W, P, HW, act, rows, pmagic, wmagic, full_groups;
  const int* live;          // device row count (nullptr: B rows), see common.hpp
  // conv stem: the network's first layer (1 input channel, 3x3) is evaluated while staging
  const float* grid;        // [B, HW] f32 board planes
  const float* w_first;     // [9][32] f32 (BatchNorm folded)
  const float* b_first;     // [32]
  int tile2_off, aux_off, plane_w;
};

namespace {
constexpr int kSfThreads = 512;
constexpr int kSfOutPitch = 528;  // bytes per flat position of the fp32 output staging tile (128 channels + 16 B pad)

// Geometry of one LDS row holding C channels as [hi plane | lo plane].
template <int C>
struct SfRow {
  static constexpr bool kSwz = C == 128;
  static constexpr int kStride = kSwz ? 512 : C * 4 + 32;
  static constexpr int kLo = C * 2;  // byte offset of the lo plane inside a row
  // byte offset of 16-byte slot `slot` (8 channels) of the hi plane of row `row`
  static __device__ __forceinline__ int off(int row, int slot) {
    return row * kStride + ((kSwz ? (slot ^ (row & 15)) : slot) << 4);
  }
};

__device__ __noinline__ float apply_act_slow(float v, int act) { return apply_act(v, act); }
template <bool RELU>
__device__ __forceinline__ void act8(float (&v)[8], int act) {
  if (RELU) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
  } else if (act != kActNone) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = apply_act_slow(v[j], act);
  }
}

// One 3x3 convolution of this wave's CB 32-channel groups (cgrp, cgrp + 1, ..) over its TPW tiles starting at flat
// position f0 - l15; acc[t][2 * ws + n] = channels (cgrp + ws) * 32 + q4 * 8 + n * 4 .. + 3 of position f0 + 16 t.
template <int CIN, int TPW, int CB>
__device__ __forceinline__ void sf_mainloop(float4v (&acc)[TPW][2 * CB], const unsigned short* __restrict__ wsrc, long long lo_off,
                                            const unsigned char* tile, int P, int f0, int q4, int cgrp, int lane) {
  using R = SfRow<CIN>;
  constexpr int KC = CIN / 32, NF = 2 * CB;
  const short8* wph = reinterpret_cast<const short8*>(wsrc) + (size_t)cgrp * 128 + lane;
  const short8* wpl = reinterpret_cast<const short8*>(wsrc + lo_off) + (size_t)cgrp * 128 + lane;
  short8 wh[NF], wl[NF];
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    wh[k] = wph[k * 64];
    wl[k] = wpl[k * 64];
  }
  // this lane's hi fragment of tile 0 at (tap, channel chunk kc); tile t is t * 16 rows further, the lo plane kLo bytes
  auto frag = [&](int tap, int kc) -> const unsigned char* {
    const int row = f0 + (tap / 3) * P + (tap % 3);
    return tile + R::off(row, kc * 4 + q4);
  };
  const unsigned char* p0 = frag(0, 0);
  short8 xh = *reinterpret_cast<const short8*>(p0), xl = *reinterpret_cast<const short8*>(p0 + R::kLo);
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int s = tap * KC + kc;
      // the packed streams end with a spare K-step, so step s + 1 is readable after the last one as well
      short8 nh[NF], nl[NF];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        nh[k] = wph[(size_t)(s + 1) * 512 + k * 64];
        nl[k] = wpl[(size_t)(s + 1) * 512 + k * 64];
      }
      const unsigned char* pc = frag(tap, kc);
      const unsigned char* pn = kc + 1 < KC ? frag(tap, kc + 1) : frag(min(tap + 1, 8), 0);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const unsigned char* py = t + 1 < TPW ? pc + (t + 1) * 16 * R::kStride : pn;
        const short8 yh = *reinterpret_cast<const short8*>(py), yl = *reinterpret_cast<const short8*>(py + R::kLo);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NF; ++k) acc[t][k] = split_mma(wl[k], xh, acc[t][k]);
#pragma unroll
        for (int k = 0; k < NF; ++k) acc[t][k] = split_mma(wh[k], xl, acc[t][k]);
#pragma unroll
        for (int k = 0; k < NF; ++k) acc[t][k] = split_mma(wh[k], xh, acc[t][k]);
        xh = yh;
        xl = yl;
      }
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        wh[k] = nh[k];
        wl[k] = nl[k];
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// Both convolutions of one wave over its NT tiles starting at tile0, from the staged tile to the fp32 output staging
// tile.  Every barrier of the workgroup between those two points is in here except the last one (which the kernel
// issues after requesting the skip connection): waves with NT and NT - 1 tiles run different instantiations and meet at
// the same barrier count.  CMID: channels of the intermediate (128: residual block, in place; 64: conv stem, second tile).
template <int CIN, int NT, bool FIRST, int CMID, bool RELU>
__device__ __forceinline__ void sf_convs(const SplitFlatArgs& a, unsigned char* smem, int tile0, int h, int lane) {
  using R2 = SfRow<CMID>;
  constexpr int CB1 = CMID / 64;  // 32-channel groups per wave in the first convolution
  const int q4 = lane >> 4, l15 = lane & 15;
  const int H = a.H, W = a.W, P = a.P;
  const int act = RELU ? (int)kActReLU : a.act;
  unsigned char* tile2 = smem + (FIRST ? a.tile2_off : 0);
  const int f0 = tile0 * 16 + l15;
  {
    float4v acc[NT][2 * CB1];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 2 * CB1; ++k) acc[t][k] = (float4v){0.f, 0.f, 0.f, 0.f};
    float bias[CB1][8];
#pragma unroll
    for (int ws = 0; ws < CB1; ++ws)
#pragma unroll
      for (int i = 0; i < 8; ++i) bias[ws][i] = a.bias[(h * CB1 + ws) * 32 + q4 * 8 + i];
    sf_mainloop<CIN, NT, CB1>(acc, a.w, a.lo_off, smem, P, f0, q4, h * CB1, lane);
    __syncthreads();  // every wave is done reading the input tile (a residual block overwrites it in place)
    // bias + activation in fp32, split, into the second convolution's tile; positions that are no pixel (the spare
    // column, the overhang behind the last image row) are written as zeros and stay the halo
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int f = f0 + t * 16;
      const int y = (f * a.pmagic) >> 16, x = f - y * P;
      const bool pixel = y < H && x < W;
#pragma unroll
      for (int ws = 0; ws < CB1; ++ws) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[t][2 * ws][j] + bias[ws][j];
          v[4 + j] = acc[t][2 * ws + 1][j] + bias[ws][4 + j];
        }
        act8<RELU>(v, act);
        short8 oh = (short8){0, 0, 0, 0, 0, 0, 0, 0}, ol = oh;
        if (pixel) split_op8(v, oh, ol);
        unsigned char* p = tile2 + R2::off(f + P + 1, (h * CB1 + ws) * 4 + q4);
        *reinterpret_cast<short8*>(p) = oh;
        *reinterpret_cast<short8*>(p + R2::kLo) = ol;
      }
    }
    __syncthreads();
  }
  {
    float4v acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[t][k] = (float4v){0.f, 0.f, 0.f, 0.f};
    float bias[2][8];
#pragma unroll
    for (int ws = 0; ws < 2; ++ws)
#pragma unroll
      for (int i = 0; i < 8; ++i) bias[ws][i] = a.bias2[(2 * h + ws) * 32 + q4 * 8 + i];
    sf_mainloop<CMID, NT, 2>(acc, a.w2, a.lo_off2, tile2, P, f0, q4, 2 * h, lane);
    __syncthreads();  // every wave is done reading: the tiles become the output staging tile
    float* otile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int ws = 0; ws < 2; ++ws) {
        float* op = otile + (size_t)(f0 + t * 16) * (kSfOutPitch / 4) + (2 * h + ws) * 32 + q4 * 8;
        const float4v lo = acc[t][2 * ws], hi = acc[t][2 * ws + 1];
        *reinterpret_cast<float4*>(op) = make_float4(lo[0] + bias[ws][0], lo[1] + bias[ws][1], lo[2] + bias[ws][2], lo[3] + bias[ws][3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(hi[0] + bias[ws][4], hi[1] + bias[ws][5], hi[2] + bias[ws][6], hi[3] + bias[ws][7]);
      }
    }
  }
}
}  // namespace

// FIRST = false: a whole residual block (alphatriangle/nn/model.py:40-59), 128 channels throughout, per launch:
// out = act(conv2(act(conv1(in))) + in) (+ post_add).
// FIRST = true: the conv stem - board plane -> conv(1 -> 32) while staging -> conv(32 -> 64) -> conv(64 -> 128), every
// layer followed by the activation; only the last layer's output goes to memory.
template <int TPW, bool FIRST, bool RELU>
__global__ __launch_bounds__(kSfThreads, 2) void flatconv_split_kernel(SplitFlatArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CIN = FIRST ? 32 : 128, CMID = FIRST ? 64 : 128;
  using R1 = SfRow<CIN>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  if (b >= live_rows(a.live, a.B)) return;  // whole workgroup, before any barrier
  const int H = a.H, W = a.W, P = a.P;
  const int act = RELU ? (int)kActReLU : a.act;

  // ---- stage the sample into the flat tile, split into the two planes; rows that are no pixel are zero rows ----
  constexpr int CPR = CIN / 8;  // 8-channel chunks per row
  if constexpr (FIRST) {
    using R2 = SfRow<CMID>;
    float* aux = reinterpret_cast<float*>(smem + a.aux_off);
    const int PW = W + 2, plane = (H + 2) * PW;
    float* gplane = aux;          // board plane with a zero border: taps need no bounds checks
    float* wl = aux + a.plane_w;  // [9][CIN]
    float* bl = wl + 9 * CIN;     // [CIN]
    for (int i = tid; i < plane; i += kSfThreads) {
      const int yy = i / PW - 1, xx = i - (yy + 1) * PW - 1;
      gplane[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? a.grid[(size_t)b * a.HW + yy * W + xx] : 0.f;
    }
    for (int i = tid; i < 9 * CIN; i += kSfThreads) wl[i] = a.w_first[i];
    for (int i = tid; i < CIN; i += kSfThreads) bl[i] = a.b_first[i];
    // the second tile starts all zero: its halo must be, its interior is written by the 32 -> 64 layer
    for (int c = tid; c < a.rows * R2::kStride / 16; c += kSfThreads)
      reinterpret_cast<uint4*>(smem + a.tile2_off)[c] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    for (int c = tid; c < a.rows * CPR; c += kSfThreads) {
      const int r = c / CPR, cc = c - r * CPR;
      const int rr = r - (P + 1);
      const int y = (rr * a.pmagic) >> 16, x = rr - y * P;
      short8 oh = (short8){0, 0, 0, 0, 0, 0, 0, 0}, ol = oh;
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
        act8<RELU>(v, act);
        split_op8(v, oh, ol);
      }
      unsigned char* p = smem + R1::off(r, cc);
      *reinterpret_cast<short8*>(p) = oh;
      *reinterpret_cast<short8*>(p + R1::kLo) = ol;
    }
  } else {
    // thread -> (pixel, 8-channel chunk); all loads of a thread in flight at once (H * W <= 256 pixels: at most eight
    // chunks per thread), the zero rows are written while they are under way
    const int total = a.HW * CPR;
    const float* src = a.in + (size_t)b * a.HW * CIN;
    float4 xin[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = min(tid + i * kSfThreads, total - 1);
      const float4* p = reinterpret_cast<const float4*>(src + (size_t)c * 8);
      xin[i][0] = p[0];
      xin[i][1] = p[1];
    }
    for (int c = tid; c < a.rows * CPR; c += kSfThreads) {
      const int r = c / CPR, cc = c - r * CPR, rr = r - (P + 1);
      const int y = (rr * a.pmagic) >> 16, x = rr - y * P;
      if (!(rr >= 0 && y < H && x < W)) {
        unsigned char* p = smem + R1::off(r, cc);
        *reinterpret_cast<uint4*>(p) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(p + R1::kLo) = make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + i * kSfThreads;
      if (c < total) {
        const int pp = c / CPR, cc = c - pp * CPR;
        const int y = (pp * a.wmagic) >> 16, x = pp - y * W;
        const float f[8] = {xin[i][0].x, xin[i][0].y, xin[i][0].z, xin[i][0].w, xin[i][1].x, xin[i][1].y, xin[i][1].z, xin[i][1].w};
        short8 oh, ol;
        split_op8(f, oh, ol);
        unsigned char* p = smem + R1::off(y * P + x + P + 1, cc);
        *reinterpret_cast<short8*>(p) = oh;
        *reinterpret_cast<short8*>(p + R1::kLo) = ol;
      }
    }
  }
  __syncthreads();

  // channel half and tile quarter of this wave; the first `full_groups` quarters hold TPW tiles, the others TPW - 1
  const int h = wave & 1, tq = wave >> 1;
  if (tq < a.full_groups)  // wave-uniform
    sf_convs<CIN, TPW, FIRST, CMID, RELU>(a, smem, tq * TPW, h, lane);
  else
    sf_convs<CIN, TPW - 1, FIRST, CMID, RELU>(a, smem, a.full_groups + tq * (TPW - 1), h, lane);

  // ---- coalesced write-back: thread -> (pixel, 4-channel chunk), whole 512-byte pixel rows per 32 lanes ----
  const int chunks = a.HW * 32;
  const size_t gbase = (size_t)b * a.HW * 128;
  // The skip-connection rows of the whole sample (at most 16 chunks per thread) are requested now that the accumulators
  // are dead, before the barrier: they arrive while the workgroup waits for its last wave.
  float4 rall[16];
  if constexpr (!FIRST) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = min(tid + i * kSfThreads, chunks - 1);
      rall[i] = *reinterpret_cast<const float4*>(a.in + gbase + (size_t)c * 4);
    }
  }
  __syncthreads();
  const float* otile = reinterpret_cast<const float*>(smem);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int c0 = tid + half * kSfThreads * 8;
    if (c0 >= chunks) break;
    // positional-table values of all eight chunks are requested before the first is used: a table value read inside
    // the chunk loop would wait for the previous chunk's store as well (one counter for both)
    float4 pv[8];
    if (a.post_add != nullptr) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = min(c0 + i * kSfThreads, chunks - 1);
        pv[i] = *reinterpret_cast<const float4*>(a.post_add + (size_t)c * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i * kSfThreads;
      if (c >= chunks) continue;
      const int p = c >> 5, cc = c & 31;
      const int y = (p * a.wmagic) >> 16, x = p - y * W;
      const float4 hv = *reinterpret_cast<const float4*>(otile + (size_t)(y * P + x) * (kSfOutPitch / 4) + cc * 4);
      float v[4] = {hv.x, hv.y, hv.z, hv.w};
      if constexpr (!FIRST) {
        const float4 rv = rall[half * 8 + i];
        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
      }
      if (RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      } else if (act != kActNone) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = apply_act_slow(v[j], act);
      }
      if (a.post_add != nullptr) {
        v[0] += pv[i].x; v[1] += pv[i].y; v[2] += pv[i].z; v[3] += pv[i].w;
      }
      *reinterpret_cast<float4*>(a.out + gbase + (size_t)c * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

namespace {
int sf_tiles(int H, int W) { return (H * (W + 1) + 15) / 16; }

// Largest tile count of a wave for a board: ceil(H * (W + 1) / 16) tiles over four tile quarters; 0 = not served.
int sf_tiles_per_wave(int H, int W) {
  if (H < 1 || W < 1 || H * W > 256) return 0;  // staging and write-back move a sample in one pass per thread
  const int tpw = (sf_tiles(H, W) + 3) / 4;
  return tpw <= 3 ? 3 : (tpw <= 5 ? tpw : 0);
}

// Geometry fields; returns the flat positions covered by the tiles or a negative error code.
int sf_fill_geometry(SplitFlatArgs& a, int H, int W, int tpw) {
  a.H = H;
  a.W = W;
  a.P = W + 1;
  a.HW = H * W;
  const int tiles = sf_tiles(H, W) > 4 * (tpw - 1) ? sf_tiles(H, W) : 4 * (tpw - 1);
  a.full_groups = tiles - 4 * (tpw - 1);
  const int slots = 16 * tiles;
  a.rows = slots + 2 * a.P + 2;
  a.pmagic = 65536 / a.P + 1;
  a.wmagic = 65536 / W + 1;
  for (int r = 0; r < a.rows; ++r)
    if (((r * a.pmagic) >> 16) != r / a.P) return -12;
  for (int p = 0; p < a.HW; ++p)
    if (((p * a.wmagic) >> 16) != p / W) return -12;
  return slots;
}

constexpr int kSfMaxLds = 160 * 1024;

template <int TPW, bool FIRST>
int launch_sf(const SplitFlatArgs& a, int lds, hipStream_t stream) {
  const bool relu = a.act == kActReLU;
#define ATRI_SF_LAUNCH(R)                                                                                                    \
  {                                                                                                                         \
    static bool attr = false;                                                                                               \
    if (!attr) {                                                                                                            \
      (void)hipFuncSetAttribute((const void*)flatconv_split_kernel<TPW, FIRST, R>, hipFuncAttributeMaxDynamicSharedMemorySize, kSfMaxLds); \
      attr = true;                                                                                                          \
    }                                                                                                                       \
    hipLaunchKernelGGL((flatconv_split_kernel<TPW, FIRST, R>), dim3(a.B), dim3(kSfThreads), lds, stream, a);                \
  }
  if (relu) ATRI_SF_LAUNCH(true) else ATRI_SF_LAUNCH(false)
#undef ATRI_SF_LAUNCH
  return hip_err(hipGetLastError());
}

template <bool FIRST>
int launch_sf_tpw(const SplitFlatArgs& a, int tpw, int lds, hipStream_t stream) {
  if (tpw == 3) return launch_sf<3, FIRST>(a, lds, stream);
  if (tpw == 4) return launch_sf<4, FIRST>(a, lds, stream);
  return launch_sf<5, FIRST>(a, lds, stream);
}

// LDS bytes of the residual-block launch / the stem launch for a board, or a negative error code.
int sf_pair_lds(SplitFlatArgs& a, int H, int W) {
  const int tpw = sf_tiles_per_wave(H, W);
  if (tpw == 0) return -10;
  const int slots = sf_fill_geometry(a, H, W, tpw);
  if (slots < 0) return slots;
  const int in_bytes = a.rows * SfRow<128>::kStride, out_bytes = slots * kSfOutPitch;
  const int lds = in_bytes > out_bytes ? in_bytes : out_bytes;
  return lds <= kSfMaxLds ? lds : -15;
}
int sf_stem_lds(SplitFlatArgs& a, int H, int W) {
  const int tpw = sf_tiles_per_wave(H, W);
  if (tpw == 0) return -10;
  const int slots = sf_fill_geometry(a, H, W, tpw);
  if (slots < 0) return slots;
  const int tile1 = a.rows * SfRow<32>::kStride, tile2 = a.rows * SfRow<64>::kStride;
  const int out_bytes = slots * kSfOutPitch;
  a.tile2_off = (tile1 + 15) & ~15;
  int lds = a.tile2_off + tile2 > out_bytes ? a.tile2_off + tile2 : out_bytes;  // the output tile reuses both input tiles
  a.aux_off = (lds + 15) & ~15;                                                  // board plane + first-layer weights behind them
  a.plane_w = ((H + 2) * (W + 2) + 3) & ~3;
  lds = a.aux_off + (a.plane_w + 10 * 32) * 4;
  return lds <= kSfMaxLds ? lds : -15;
}
}  // namespace

extern "C" {

// Whether the fp32-grade flat-tile kernels serve this board (boards the 8x15 halo-tile kernels refuse).
int atri_split_flatconv_supported(int H, int W) {
  SplitFlatArgs a{};
  return sf_pair_lds(a, H, W) > 0 && sf_stem_lds(a, H, W) > 0 ? 1 : 0;
}

// Residual block in one launch, fp32-grade: out = act(conv2(act(conv1(in))) + in) (+ post_add), 128 channels.
// w1 / w2: packed 16-bit fragments [hi stream | lo stream], lo1 / lo2 = element offset of the lo stream.
int atri_split_flatconv_pair(const float* in, const void* w1, long long lo1, const float* b1, const void* w2, long long lo2,
                             const float* b2, const float* post_add, float* out, int B, int H, int W, int C, int act,
                             hipStream_t stream) {
  if (B <= 0) return 0;
  if (C != 128) return -10;
  if (in == out) return -13;  // the skip connection is read after other workgroups may have written their output
  SplitFlatArgs a{};
  const int lds = sf_pair_lds(a, H, W);
  if (lds < 0) return lds;
  a.in = in;
  a.w = (const unsigned short*)w1;
  a.lo_off = lo1;
  a.bias = b1;
  a.w2 = (const unsigned short*)w2;
  a.lo_off2 = lo2;
  a.bias2 = b2;
  a.post_add = post_add;
  a.out = out;
  a.B = B;
  a.act = act;
  a.live = live_count_ptr();
  return launch_sf_tpw<false>(a, sf_tiles_per_wave(H, W), lds, stream);
}

// Conv stem in one launch, fp32-grade: board plane -> conv(1 -> 32) -> conv(32 -> 64) -> conv(64 -> 128), every layer
// followed by the activation; only the last layer's output goes to memory.
int atri_split_flatconv_stem(const float* grid, const float* w_first, const float* b_first, const void* w1, long long lo1,
                             const float* b1, const void* w2, long long lo2, const float* b2, const float* post_add, float* out,
                             int B, int H, int W, int C0, int C1, int C2, int act, hipStream_t stream) {
  if (B <= 0) return 0;
  if (C0 != 32 || C1 != 64 || C2 != 128) return -10;
  SplitFlatArgs a{};
  const int lds = sf_stem_lds(a, H, W);
  if (lds < 0) return lds;
  a.grid = grid;
  a.w_first = w_first;
  a.b_first = b_first;
  a.w = (const unsigned short*)w1;
  a.lo_off = lo1;
  a.bias = b1;
  a.w2 = (const unsigned short*)w2;
  a.lo_off2 = lo2;
  a.bias2 = b2;
  a.post_add = post_add;
  a.out = out;
  a.B = B;
  a.act = act;
  a.live = live_count_ptr();
  return launch_sf_tpw<true>(a, sf_tiles_per_wave(H, W), lds, stream);
}

}  // extern "C"
