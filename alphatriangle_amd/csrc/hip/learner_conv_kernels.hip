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
// fragments).  The four waves split the output channels (COT tiles of 16 each); every wave walks all position tiles,
// so the filter fragments come from global memory (L2-resident, prepacked by conv3_pack_kernel into [tap][ci / 32][co][32]
// (hi, lo) planes, loaded one step ahead) and the input fragments from LDS.
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
__global__ __launch_bounds__(256) void conv3_cm_kernel(Conv3CmArgs a) {
  using G = C3Geom<CIN, NT>;
  constexpr int COUT = 64 * COT;
  constexpr int KC = CIN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int hw = a.h * a.w;
  const long long n0 = (long long)blockIdx.x * hw;

  // ---- the board -> LDS: 4 (ci) x 4 (positions) blocks, float4 along the positions, transposed in registers
  {
    constexpr int kPosQuads = G::kRows / 4;
    const float* xb = a.x + n0;
    for (int blk = tid; blk < (CIN / 4) * kPosQuads; blk += 256) {
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
    for (int i = tid; i < G::kRowBytes / 8; i += 256) {  // the zero row of both planes
      *reinterpret_cast<uint2*>(smem + G::kRows * G::kRowBytes + i * 8) = make_uint2(0u, 0u);
      *reinterpret_cast<uint2*>(smem + G::kPlane + G::kRows * G::kRowBytes + i * 8) = make_uint2(0u, 0u);
    }
  }

  // ---- which taps of this lane's positions lie inside the board (bit tap of valid[t])
  unsigned valid[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int p = t * 16 + l15;
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

  float4v acc[COT][NT];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[c][t] = (float4v){0.f, 0.f, 0.f, 0.f};

  // filter fragments of this wave's COT channel tiles for step s = tap * KC + kc: 16 bytes per lane and plane
  const int co_lane = (wave * COT) * 16 + l15;
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
    int row_off[NT], row_swz[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int row = t * 16 + l15 + shift;
      const bool ok = (valid[t] >> tap) & 1u;
      row_off[t] = ok ? row * G::kRowBytes : kZeroRow;
      row_swz[t] = ok ? (row & G::kSwz) : 0;
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int cur = kAhead ? (kc & 1) : 0, nxt = cur ^ 1;
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
      const int slot = kc * 4 + q4;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const unsigned char* p = smem + row_off[t] + ((slot ^ row_swz[t]) << 4);
        const short8 xh = *reinterpret_cast<const short8*>(p);
        const short8 xl = *reinterpret_cast<const short8*>(p + G::kPlane);
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[c][t] = mfma_split(xh, xl, wh[cur][c], wl[cur][c], acc[c][t]);
      }
    }
  }

  // acc[c][t][e] = out(co = (wave * COT + c) * 16 + l15, position t * 16 + 4 q4 + e)
#pragma unroll
  for (int c = 0; c < COT; ++c) {
    float* orow = a.out + (long long)(co_lane + c * 16) * a.n_total + n0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int pos = t * 16 + q4 * 4;
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

namespace {
template <int CIN, int COT, int NT>
int launch_conv3(const Conv3CmArgs& a, int boards, hipStream_t s) {
  using G = C3Geom<CIN, NT>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv3_cm_kernel<CIN, COT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLds);
    attr = true;
  }
  hipLaunchKernelGGL((conv3_cm_kernel<CIN, COT, NT>), dim3(boards), dim3(256), G::kLds, s, a);
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

}  // extern "C"
