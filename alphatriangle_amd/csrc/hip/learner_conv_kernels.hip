// 3x3, padding-1 convolution of the learner's fp32 step on channel-major activations, without a patch matrix:
//
//   out[co][n] = sum over (tap, ci) of W[co][ci][tap] * x[ci][n + shift(tap)]       x: [Cin][B * H * W], out: [Cout][B * H * W]
//
// (alphatriangle/nn/model.py:15-60 builds these layers; the same kernel computes the data gradient when it is given the
// output gradient and the flipped, transposed filter.)  The GEMM-lowered form writes a nine times larger patch matrix
// and reads it back (2 x 141 MB per layer and direction at batch 256); here one workgroup owns one board: it stages
// the board's Cin x (H * W) input once into LDS - transposed to [position][ci] and split into (hi, lo) bf16 planes - and
// the nine taps are nine row-shifted reads of that image.  Every product is three bf16 MFMA passes over (hi, lo) pairs
// with fp32 accumulation, like the learner's GEMMs (learner_gemm_kernels.hip).
//
// Geometry: MFMA rows are positions (16 per tile, NT tiles cover the board), columns are output channels.  A tap that
// falls outside the board reads an all-zero LDS row instead (one address select per tile and step, no masking of
// fragments).  Eight waves: four groups of output channels (COT tiles of 16 each) times two halves of the position
// tiles.  The filter fragments come from global memory (L2-resident, prepacked by conv3_pack_kernel into
// [tap][ci / 32][co][32] (hi, lo) planes, loaded one step ahead), the input fragments from LDS.  (Four waves that each
// walk all position tiles read the image twice as often and leave one wave per SIMD: 41 us against the patch-matrix
// form's 101 us at 128 -> 128 channels and 256 boards.)
#include <type_traits>
#include <utility>

#include "common.hpp"

using namespace atrihip;

struct Conv3CmArgs {
  const float* x;             // [cin][n_total]
  const unsigned short* whi;  // [9][cin / 32][cout][32]
  const unsigned short* wlo;
  float* out;                 // [cout][n_total]
  long long n_total;          // boards * h * w
  int h, w;
  int vec;                    // widest aligned access along the positions: 4 or 1 floats
};

namespace {
__device__ __forceinline__ void c3_split4(const float4& v, uint2& hi, uint2& lo) {
  const unsigned h0 = pack_bf16x2(v.x, v.y), h1 = pack_bf16x2(v.z, v.w);
  const float r0 = v.x - bf16_to_float((unsigned short)(h0 & 0xFFFF)), r1 = v.y - bf16_to_float((unsigned short)(h0 >> 16));
  const float r2 = v.z - bf16_to_float((unsigned short)(h1 & 0xFFFF)), r3 = v.w - bf16_to_float((unsigned short)(h1 >> 16));
  hi = make_uint2(h0, h1);
  lo = make_uint2(pack_bf16x2(r0, r1), pack_bf16x2(r2, r3));
}

__device__ __forceinline__ float4 c3_load_quad(const float* p, int left, int vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec == 4) {
    if (left > 0) v = *reinterpret_cast<const float4*>(p);
  } else {
    if (left > 0) v.x = p[0];
    if (left > 1) v.y = p[1];
    if (left > 2) v.z = p[2];
    if (left > 3) v.w = p[3];
  }
  return v;
}

template <typename F, int... Is>
__device__ __forceinline__ void c3_static_for_impl(F& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void c3_static_for(F& f) {  // f(integral_constant<0>) ... f(integral_constant<N - 1>), unrolled
  c3_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int CIN, int NT>
struct C3Geom {
  static constexpr int kRows = NT * 16;             // position rows of the LDS image; row kRows is all zeros
  static constexpr int kRowBytes = CIN * 2;         // one plane
  static constexpr int kSlots = CIN / 8;            // 16-byte pieces per row
  static constexpr int kSwz = (kSlots < 16 ? kSlots : 16) - 1;
  static constexpr int kPlane = (kRows + 1) * kRowBytes;
  static constexpr int kLds = 2 * kPlane;
};
}  // namespace

template <int CIN, int COT, int NT>
__global__ __launch_bounds__(512) void conv3_cm_kernel(Conv3CmArgs a) {
  using G = C3Geom<CIN, NT>;
  constexpr int COUT = 64 * COT;
  constexpr int KC = CIN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTW = NT / 2;  // position tiles per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wc = wave & 3, t0 = (wave >> 2) * NTW;  // channel group, first position tile
  const int hw = a.h * a.w;
  const long long n0 = (long long)blockIdx.x * hw;

  // ---- the board -> LDS: 4 (ci) x 4 (positions) blocks, float4 along the positions, transposed in registers
  {
    constexpr int kPosQuads = G::kRows / 4;
    const float* xb = a.x + n0;
    for (int blk = tid; blk < (CIN / 4) * kPosQuads; blk += 512) {
      const int pq = blk % kPosQuads, cq = blk / kPosQuads;
      const int pos = pq * 4;
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = c3_load_quad(xb + (long long)(cq * 4 + i) * a.n_total + pos, hw - pos, a.vec);
      const float f[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                             {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = pos + j;
        uint2 hi, lo;
        c3_split4(make_float4(f[0][j], f[1][j], f[2][j], f[3][j]), hi, lo);
        unsigned char* p = smem + row * G::kRowBytes + ((((cq >> 1) ^ (row & G::kSwz))) << 4) + ((cq & 1) << 3);
        *reinterpret_cast<uint2*>(p) = hi;
        *reinterpret_cast<uint2*>(p + G::kPlane) = lo;
      }
    }
    for (int i = tid; i < G::kRowBytes / 8; i += 512) {  // the zero row of both planes
      *reinterpret_cast<uint2*>(smem + G::kRows * G::kRowBytes + i * 8) = make_uint2(0u, 0u);
      *reinterpret_cast<uint2*>(smem + G::kPlane + G::kRows * G::kRowBytes + i * 8) = make_uint2(0u, 0u);
    }
  }

  // ---- which taps of this lane's positions lie inside the board (bit tap of valid[t])
  unsigned valid[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int p = (t0 + t) * 16 + l15;
    const int y = p / a.w, x = p - y * a.w;
    unsigned m = 0;
    if (p < hw) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
        if (yy >= 0 && yy < a.h && xx >= 0 && xx < a.w) m |= 1u << tp;
      }
    }
    valid[t] = m;
  }

  float4v acc[COT][NTW];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[c][t] = (float4v){0.f, 0.f, 0.f, 0.f};

  // filter fragments of this wave's COT channel tiles for step s = tap * KC + kc: 16 bytes per lane and plane
  const int co_lane = (wc * COT) * 16 + l15;
  auto w_offset = [&](int s, int c) { return ((long long)s * COUT + co_lane + c * 16) * 32 + q4 * 8; };
  // KC >= 2 (even): fragments of step s + 1 are loaded while step s is multiplied; the buffer parity is that of kc.
  // KC == 1 (32 input channels, the small early layer): loaded at the start of their step.
  constexpr bool kAhead = KC > 1;
  short8 wh[2][COT], wl[2][COT];
  if constexpr (kAhead) {
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      wh[0][c] = *reinterpret_cast<const short8*>(a.whi + w_offset(0, c));
      wl[0][c] = *reinterpret_cast<const short8*>(a.wlo + w_offset(0, c));
    }
  }
  __syncthreads();

  constexpr int kZeroRow = G::kRows * G::kRowBytes;
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int shift = (tap / 3 - 1) * a.w + (tap % 3 - 1);
    int row_off[NTW], row_swz[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int row = (t0 + t) * 16 + l15 + shift;
      const bool ok = (valid[t] >> tap) & 1u;
      row_off[t] = ok ? row * G::kRowBytes : kZeroRow;
      row_swz[t] = ok ? (row & G::kSwz) : 0;
    }
    // One tap = KC * NG units of TG position tiles at one 32-channel step.  The fragments of unit u + 1 are read from LDS
    // while unit u is multiplied, and a unit's MFMAs go pass by pass over TG * COT independent accumulators (three
    // chained passes on one accumulator would wait for each other).
    constexpr int TG = 4, NG = NTW / TG, U = KC * NG;
    short8 xh[2][TG], xl[2][TG];
    auto load_unit = [&](auto UC, auto BUF) {
      constexpr int u = decltype(UC)::value, buf = decltype(BUF)::value;
      constexpr int kc = u / NG, g = u % NG;
      const int slot = kc * 4 + q4;
#pragma unroll
      for (int i = 0; i < TG; ++i) {
        const int t = g * TG + i;
        const unsigned char* p = smem + row_off[t] + ((slot ^ row_swz[t]) << 4);
        xh[buf][i] = *reinterpret_cast<const short8*>(p);
        xl[buf][i] = *reinterpret_cast<const short8*>(p + G::kPlane);
      }
    };
    load_unit(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    auto unit = [&](auto UC) {
      constexpr int u = decltype(UC)::value;
      constexpr int kc = u / NG, g = u % NG, buf = u & 1;
      constexpr int cur = kAhead ? (kc & 1) : 0, nxt = cur ^ 1;
      if constexpr (g == 0) {  // filter fragments: one 32-channel step ahead (or for this step when KC == 1)
        const int s = tap * KC + kc;
        if constexpr (kAhead) {
          if (s + 1 < 9 * KC) {
#pragma unroll
            for (int c = 0; c < COT; ++c) {
              wh[nxt][c] = *reinterpret_cast<const short8*>(a.whi + w_offset(s + 1, c));
              wl[nxt][c] = *reinterpret_cast<const short8*>(a.wlo + w_offset(s + 1, c));
            }
          }
        } else {
#pragma unroll
          for (int c = 0; c < COT; ++c) {
            wh[0][c] = *reinterpret_cast<const short8*>(a.whi + w_offset(s, c));
            wl[0][c] = *reinterpret_cast<const short8*>(a.wlo + w_offset(s, c));
          }
        }
      }
      if constexpr (u + 1 < U) load_unit(std::integral_constant<int, u + 1>{}, std::integral_constant<int, (u + 1) & 1>{});
#pragma unroll
      for (int i = 0; i < TG; ++i)
#pragma unroll
        for (int c = 0; c < COT; ++c)
          acc[c][g * TG + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[buf][i], wh[cur][c], acc[c][g * TG + i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TG; ++i)
#pragma unroll
        for (int c = 0; c < COT; ++c)
          acc[c][g * TG + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[buf][i], wl[cur][c], acc[c][g * TG + i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TG; ++i)
#pragma unroll
        for (int c = 0; c < COT; ++c)
          acc[c][g * TG + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[buf][i], wh[cur][c], acc[c][g * TG + i], 0, 0, 0);
    };
    c3_static_for<U>(unit);
  }

  // acc[c][t][e] = out(co = (wc * COT + c) * 16 + l15, position (t0 + t) * 16 + 4 q4 + e)
#pragma unroll
  for (int c = 0; c < COT; ++c) {
    float* orow = a.out + (long long)(co_lane + c * 16) * a.n_total + n0;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int pos = (t0 + t) * 16 + q4 * 4;
      const int left = hw - pos;
      if (left <= 0) continue;
      if (a.vec == 4) {
        *reinterpret_cast<float4*>(orow + pos) = make_float4(acc[c][t][0], acc[c][t][1], acc[c][t][2], acc[c][t][3]);
      } else {
        orow[pos] = acc[c][t][0];
        if (left > 1) orow[pos + 1] = acc[c][t][1];
        if (left > 2) orow[pos + 2] = acc[c][t][2];
        if (left > 3) orow[pos + 3] = acc[c][t][3];
      }
    }
  }
}

// Filter -> (hi, lo) bf16 planes [9][K / 32][M][32].  Forward: K = cin, M = cout, element (m, k, tap) = w[m][k][tap].
// Data gradient: K = cout, M = cin, element (m, k, tap) = w[k][m][8 - tap] (flipped taps, channels swapped).
template <bool DGRAD>
__global__ __launch_bounds__(256) void conv3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ hi,
                                                         unsigned short* __restrict__ lo, int cout, int cin) {
  const int M = DGRAD ? cin : cout, K = DGRAD ? cout : cin;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 9 * K * M) return;
  const int kk = i & 31, m = (i >> 5) % M, rest = (i >> 5) / M;
  const int kc = rest % (K / 32), tap = rest / (K / 32);
  const int k = kc * 32 + kk;
  const float v = DGRAD ? w[((size_t)k * cin + m) * 9 + (8 - tap)] : w[((size_t)m * cin + k) * 9 + tap];
  const unsigned short h = float_to_bf16(v);
  hi[i] = h;
  lo[i] = float_to_bf16(v - bf16_to_float(h));
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution without a patch matrix:
//
//   dW[co][ci][dy, dx] = sum over boards and positions q of g[co][q] * x[ci][q + dy * P + dx]
//
// in PADDED board coordinates (pitch P = W + 1 rounded up to a multiple of 8, one zero row above and below): every
// position outside the board holds a zero in the staged copy of x, and g is zero in the padding columns, so no product
// needs a mask.  The reduction index of the MFMA is the position, contiguous for both operands as they lie in memory
// (no transposition).  A tap shifts the x operand by dy * P + dx positions; with P a multiple of 8 only dx moves a
// 16-byte LDS read off its alignment, so x is staged three times, shifted by -1, 0 and +1 positions, and every read is
// aligned.  One workgroup = (a slice of the boards) x (32 input channels) x all output channels x 9 taps; its eight waves
// are 4 groups of output-channel tiles x 2 input-channel tiles, 9 * COT accumulator tiles each.  The slices are added by
// conv3_wgrad_reduce_kernel, which also turns [tap][co][ci] into the parameter's [co][ci][tap].
struct Conv3WgArgs {
  const float* g;    // [cout][n_total]
  const float* x;    // [cin][n_total]
  float* part;       // [slices][9][cout][cin]
  long long n_total;
  int cin, boards, boards_per_wg, h, w, P, ksteps;
  int g_stride, x_stride;  // bytes per LDS row (one plane), padded against bank conflicts
};

namespace {
__device__ __forceinline__ void c3_split8(const float (&v)[8], uint4& hi, uint4& lo) {
  uint2 h0, l0, h1, l1;
  c3_split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
  c3_split4(make_float4(v[4], v[5], v[6], v[7]), h1, l1);
  hi = make_uint4(h0.x, h0.y, h1.x, h1.y);
  lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
}
}  // namespace

template <int COT>
__global__ __launch_bounds__(512) void conv3_wgrad_kernel(Conv3WgArgs a) {
  constexpr int COUT = 64 * COT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int cg = wave & 3, half = wave >> 2;
  const int hw = a.h * a.w, P = a.P;
  const int K32 = a.ksteps * 32;      // padded positions of one board the reduction runs over (rows 1 .. H of the padded image)
  const int Lq = K32 + 2 * P;         // staged x positions: one padded row before and after
  unsigned char* g_hi = smem;
  unsigned char* g_lo = g_hi + COUT * a.g_stride;
  unsigned char* x_hi = g_lo + COUT * a.g_stride;
  unsigned char* x_lo = x_hi + 3 * 32 * a.x_stride;
  const int ci0 = blockIdx.y * 32;

  float4v acc[9][COT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[t][c] = (float4v){0.f, 0.f, 0.f, 0.f};

  for (int bi = 0; bi < a.boards_per_wg; ++bi) {
    const int board = blockIdx.x * a.boards_per_wg + bi;
    if (board >= a.boards) break;  // uniform over the workgroup
    const long long n0 = (long long)board * hw;
    __syncthreads();  // the previous board's fragments have been read
    // g -> [co][k] planes, k = y * P + c with board column c - 1 (zeros in the padding columns and beyond the last row)
    for (int o = tid; o < COUT * (K32 / 8); o += 512) {
      const int co = o / (K32 / 8), k8 = (o - co * (K32 / 8)) * 8;
      const int y = k8 / P, c0 = k8 - y * P;
      const float* src = a.g + (long long)co * a.n_total + n0 + y * a.w + (c0 - 1);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int xb = c0 + e - 1;
        v[e] = (y < a.h && xb >= 0 && xb < a.w) ? src[e] : 0.f;
      }
      uint4 hi, lo;
      c3_split8(v, hi, lo);
      *reinterpret_cast<uint4*>(g_hi + co * a.g_stride + k8 * 2) = hi;
      *reinterpret_cast<uint4*>(g_lo + co * a.g_stride + k8 * 2) = lo;
    }
    // x -> three copies [d][ci][j], copy d holding the padded image shifted by d - 1 positions
    for (int o = tid; o < 32 * (Lq / 8); o += 512) {
      const int ci = o / (Lq / 8), j8 = (o - ci * (Lq / 8)) * 8;
      const float* src = a.x + (long long)(ci0 + ci) * a.n_total + n0;
      float v[10];  // padded positions j8 - 1 .. j8 + 8
#pragma unroll
      for (int e = 0; e < 10; ++e) {
        const int q = j8 - 1 + e;
        const int r = q >= 0 ? q / P : -1;
        const int yy = r - 1, xb = q - r * P - 1;
        const bool ok = q >= 0 && yy >= 0 && yy < a.h && xb >= 0 && xb < a.w;
        v[e] = ok ? src[yy * a.w + xb] : 0.f;
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float f[8] = {v[d], v[d + 1], v[d + 2], v[d + 3], v[d + 4], v[d + 5], v[d + 6], v[d + 7]};
        uint4 hi, lo;
        c3_split8(f, hi, lo);
        *reinterpret_cast<uint4*>(x_hi + (d * 32 + ci) * a.x_stride + j8 * 2) = hi;
        *reinterpret_cast<uint4*>(x_lo + (d * 32 + ci) * a.x_stride + j8 * 2) = lo;
      }
    }
    __syncthreads();
    for (int ks = 0; ks < a.ksteps; ++ks) {
      const int kb = ks * 32 + q4 * 8;
      short8 gh[COT], gl[COT];
#pragma unroll
      for (int c = 0; c < COT; ++c) {
        const int off = ((cg * COT + c) * 16 + l15) * a.g_stride + kb * 2;
        gh[c] = *reinterpret_cast<const short8*>(g_hi + off);
        gl[c] = *reinterpret_cast<const short8*>(g_lo + off);
      }
      short8 xh[9], xl[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dy = t / 3 - 1, d = t % 3;
        const int off = (d * 32 + half * 16 + l15) * a.x_stride + (kb + P + dy * P) * 2;
        xh[t] = *reinterpret_cast<const short8*>(x_hi + off);
        xl[t] = *reinterpret_cast<const short8*>(x_lo + off);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[t], gh[c], acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[t], gl[c], acc[t][c], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[t], gh[c], acc[t][c], 0, 0, 0);
    }
  }
  // acc[t][c][e] = dW(tap t, co = (cg * COT + c) * 16 + l15, ci = ci0 + half * 16 + 4 q4 + e) of this slice
  float* part = a.part + (long long)blockIdx.x * 9 * COUT * a.cin;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      const int co = (cg * COT + c) * 16 + l15;
      *reinterpret_cast<float4*>(part + ((long long)t * COUT + co) * a.cin + ci0 + half * 16 + q4 * 4) =
          make_float4(acc[t][c][0], acc[t][c][1], acc[t][c][2], acc[t][c][3]);
    }
}

// dW[co][ci][tap] = sum over the slices of part[s][tap][co][ci].  64 quads of ci per workgroup, the four waves take every
// fourth slice each.
__global__ __launch_bounds__(256) void conv3_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int cout,
                                                                 int cin) {
  __shared__ float4 sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long F = 9LL * cout * cin;
  const long long f = ((long long)blockIdx.x * 64 + lane) * 4;
  float4 acc4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (f < F) {
    const float* p = part + f;
    int s = wave;
    for (; s + 12 < S; s += 16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(p + (long long)(s + 4 * i) * F);
        acc4[i].x += v.x; acc4[i].y += v.y; acc4[i].z += v.z; acc4[i].w += v.w;
      }
    }
    for (; s < S; s += 4) {
      const float4 v = *reinterpret_cast<const float4*>(p + (long long)s * F);
      acc4[0].x += v.x; acc4[0].y += v.y; acc4[0].z += v.z; acc4[0].w += v.w;
    }
  }
  sm[wave][lane] = make_float4((acc4[0].x + acc4[1].x) + (acc4[2].x + acc4[3].x), (acc4[0].y + acc4[1].y) + (acc4[2].y + acc4[3].y),
                               (acc4[0].z + acc4[1].z) + (acc4[2].z + acc4[3].z), (acc4[0].w + acc4[1].w) + (acc4[2].w + acc4[3].w));
  __syncthreads();
  if (wave == 0 && f < F) {
    float4 t = sm[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      t.x += sm[w][lane].x; t.y += sm[w][lane].y; t.z += sm[w][lane].z; t.w += sm[w][lane].w;
    }
    const int ci = (int)(f % cin), co = (int)((f / cin) % cout), tap = (int)(f / ((long long)cin * cout));
    float* o = dw + ((long long)co * cin + ci) * 9 + tap;
    o[0] = t.x; o[9] = t.y; o[18] = t.z; o[27] = t.w;
  }
}

// Every filter of a step in one launch: entry e = blockIdx.y (a filter and a direction), blockIdx.x walks its elements.
constexpr int kPackMany = 24;
struct Conv3PackMany {
  const float* w[kPackMany];
  unsigned short* hi[kPackMany];
  unsigned short* lo[kPackMany];
  int cout[kPackMany], cin[kPackMany], dgrad[kPackMany];
};
__global__ __launch_bounds__(256) void conv3_pack_many_kernel(Conv3PackMany a) {
  const int e = blockIdx.y;
  const int cout = a.cout[e], cin = a.cin[e];
  const bool dg = a.dgrad[e] != 0;
  const int M = dg ? cin : cout, K = dg ? cout : cin;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 9 * K * M) return;
  const int kk = i & 31, m = (i >> 5) % M, rest = (i >> 5) / M;
  const int kc = rest % (K / 32), tap = rest / (K / 32);
  const int k = kc * 32 + kk;
  const float* w = a.w[e];
  const float v = dg ? w[((size_t)k * cin + m) * 9 + (8 - tap)] : w[((size_t)m * cin + k) * 9 + tap];
  const unsigned short h = float_to_bf16(v);
  a.hi[e][i] = h;
  a.lo[e][i] = float_to_bf16(v - bf16_to_float(h));
}

namespace {
template <int CIN, int COT, int NT>
int launch_conv3(const Conv3CmArgs& a, int boards, hipStream_t s) {
  using G = C3Geom<CIN, NT>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv3_cm_kernel<CIN, COT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLds);
    attr = true;
  }
  hipLaunchKernelGGL((conv3_cm_kernel<CIN, COT, NT>), dim3(boards), dim3(512), G::kLds, s, a);
  return hip_err(hipGetLastError());
}
}  // namespace

extern "C" {

// 1 when atri_conv3_cm handles these sizes: cin in {32, 64, 128}, cout in {64, 128}, boards of at most 256 positions.
int atri_conv3_cm_supported(int cin, int cout, int h, int w) {
  if (h < 1 || w < 1 || h * w > 256) return 0;
  if (cin != 32 && cin != 64 && cin != 128) return 0;
  if (cout != 64 && cout != 128) return 0;
  if (h * w > 128 && !(cin == 128 || cin == 64)) return 0;
  return 1;
}

// w: fp32 [cout][cin][3][3]; hi / lo: 9 * cin * cout bf16 each.
int atri_conv3_pack(const float* w, unsigned short* hi, unsigned short* lo, int cout, int cin, int dgrad, hipStream_t s) {
  const int K = dgrad ? cout : cin;
  if (cout <= 0 || cin <= 0 || (K & 31)) return (int)hipErrorInvalidValue;
  const dim3 grid((9 * cout * cin + 255) / 256);
  if (dgrad)
    hipLaunchKernelGGL(conv3_pack_kernel<true>, grid, dim3(256), 0, s, w, hi, lo, cout, cin);
  else
    hipLaunchKernelGGL(conv3_pack_kernel<false>, grid, dim3(256), 0, s, w, hi, lo, cout, cin);
  return hip_err(hipGetLastError());
}

// n <= 24 filters (each fp32 [cout][cin][3][3]) packed in one launch; entry e is packed for the data gradient when dgrad[e] != 0.
int atri_conv3_pack_many(int n, const float* const* w, unsigned short* const* hi, unsigned short* const* lo, const int* cout,
                         const int* cin, const int* dgrad, hipStream_t s) {
  if (n <= 0) return 0;
  if (n > kPackMany) return (int)hipErrorInvalidValue;
  Conv3PackMany a{};
  int most = 0;
  for (int e = 0; e < n; ++e) {
    const int K = dgrad[e] ? cout[e] : cin[e];
    if (cout[e] <= 0 || cin[e] <= 0 || (K & 31)) return (int)hipErrorInvalidValue;
    a.w[e] = w[e]; a.hi[e] = hi[e]; a.lo[e] = lo[e];
    a.cout[e] = cout[e]; a.cin[e] = cin[e]; a.dgrad[e] = dgrad[e];
    most = most > 9 * cout[e] * cin[e] ? most : 9 * cout[e] * cin[e];
  }
  hipLaunchKernelGGL(conv3_pack_many_kernel, dim3((most + 255) / 256, n), dim3(256), 0, s, a);
  return hip_err(hipGetLastError());
}

// out[cout][boards * h * w] = conv3x3(x[cin][boards * h * w]) with the packed filter (atri_conv3_pack).
int atri_conv3_cm(const float* x, const unsigned short* whi, const unsigned short* wlo, float* out, int cin, int cout, int boards,
                  int h, int w, hipStream_t s) {
  if (boards <= 0) return 0;
  if (!atri_conv3_cm_supported(cin, cout, h, w)) return (int)hipErrorInvalidValue;
  if (((uintptr_t)whi | (uintptr_t)wlo) & 15) return (int)hipErrorInvalidValue;
  Conv3CmArgs a{x, whi, wlo, out, (long long)boards * h * w, h, w, 4};
  const int hw = h * w;
  if ((hw & 3) || (((uintptr_t)x | (uintptr_t)out) & 15)) a.vec = 1;
  const bool big = hw > 128;
#define ATRI_C3(CIN, COT)                                              \
  if (cin == CIN && cout == 64 * COT) {                                \
    if (big) return launch_conv3<CIN, COT, 16>(a, boards, s);          \
    return launch_conv3<CIN, COT, 8>(a, boards, s);                    \
  }
  ATRI_C3(128, 2)
  ATRI_C3(128, 1)
  ATRI_C3(64, 2)
  ATRI_C3(64, 1)
#undef ATRI_C3
  if (!big) {
    if (cin == 32 && cout == 128) return launch_conv3<32, 2, 8>(a, boards, s);
    if (cin == 32 && cout == 64) return launch_conv3<32, 1, 8>(a, boards, s);
  }
  return (int)hipErrorInvalidValue;
}

// Sizes of the weight-gradient kernel for a board: pitch, steps, LDS bytes; 0 when the board does not fit (then the caller
// multiplies with the patch matrix instead).
static int c3_wgrad_geometry(int cout, int h, int w, int* P, int* ksteps, int* g_stride, int* x_stride) {
  *P = (w + 1 + 7) / 8 * 8;
  const int K32 = (h * *P + 31) / 32 * 32;
  *ksteps = K32 / 32;
  const int Lq = K32 + 2 * *P;
  // row pitches in bytes: a multiple of 16 that is 16 mod 256 apart from a multiple of 256 spreads 16 rows over all banks
  auto pitch = [](int bytes) {
    int p = (bytes + 15) / 16 * 16;
    while ((p / 4) % 64 != 4 && (p / 4) % 64 != 20 && (p / 4) % 64 != 36 && (p / 4) % 64 != 52) p += 16;
    return p;
  };
  *g_stride = pitch(K32 * 2);
  *x_stride = pitch(Lq * 2);
  return 2 * cout * *g_stride + 2 * 3 * 32 * *x_stride;
}

// 1 when atri_conv3_wgrad handles these sizes.
int atri_conv3_wgrad_supported(int cin, int cout, int h, int w) {
  if (h < 1 || w < 1 || (cin & 31) || cin < 32 || (cout != 64 && cout != 128)) return 0;
  int P, ks, gs, xs;
  return c3_wgrad_geometry(cout, h, w, &P, &ks, &gs, &xs) <= 160 * 1024;
}

// Number of board slices (= partial results) atri_conv3_wgrad uses for this problem.
int atri_conv3_wgrad_slices(int cin, int boards) {
  const int blocks = cin / 32;
  int s = 256 / (blocks > 0 ? blocks : 1);
  if (s < 1) s = 1;
  if (s > boards) s = boards;
  const int per = (boards + s - 1) / s;
  return (boards + per - 1) / per;
}

// dw[cout][cin][3][3] = weight gradient of the 3x3, padding-1 convolution: g [cout][boards*h*w], x [cin][boards*h*w] (fp32,
// channel-major); part: atri_conv3_wgrad_slices(cin, boards) * 9 * cout * cin floats of scratch.
int atri_conv3_wgrad(const float* g, const float* x, float* part, float* dw, int cin, int cout, int boards, int h, int w,
                     hipStream_t s) {
  if (boards <= 0) return (int)hipErrorInvalidValue;
  if (!atri_conv3_wgrad_supported(cin, cout, h, w)) return (int)hipErrorInvalidValue;
  if (((uintptr_t)part | (uintptr_t)dw) & 15) return (int)hipErrorInvalidValue;
  Conv3WgArgs a{};
  a.g = g; a.x = x; a.part = part; a.n_total = (long long)boards * h * w;
  a.cin = cin; a.boards = boards; a.h = h; a.w = w;
  const int lds = c3_wgrad_geometry(cout, h, w, &a.P, &a.ksteps, &a.g_stride, &a.x_stride);
  const int slices = atri_conv3_wgrad_slices(cin, boards);
  a.boards_per_wg = (boards + slices - 1) / slices;
  const dim3 grid(slices, cin / 32);
  if (cout == 128) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)conv3_wgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL(conv3_wgrad_kernel<2>, grid, dim3(512), lds, s, a);
  } else {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)conv3_wgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    hipLaunchKernelGGL(conv3_wgrad_kernel<1>, grid, dim3(512), lds, s, a);
  }
  const long long F = 9LL * cout * cin;
  hipLaunchKernelGGL(conv3_wgrad_reduce_kernel, dim3((unsigned)((F / 4 + 63) / 64)), dim3(256), 0, s, part, dw, slices, cout, cin);
  return hip_err(hipGetLastError());
}

}  // extern "C"
