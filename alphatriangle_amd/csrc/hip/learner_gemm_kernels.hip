// fp32 GEMM on the bf16 matrix cores for the learner's fp32 mode: C[M, N] = A[M, K] B[K, N] (+ bias[N]) with fp32 inputs and
// outputs, every product as three bf16 MFMA passes over (hi, lo) operand splits (hi*hi + lo*hi + hi*lo, fp32
// accumulation) - the precision scheme of the fp32-grade inference kernels (split_conv_kernels.hip).  gfx950's native
// fp32 MFMA runs at 1/16 of the bf16 rate, which is where the library GEMMs of an fp32 training step spend their time
// (the reference trains in fp32: alphatriangle/rl/core/trainer.py:204-310).
//
// One 4-wave workgroup computes a BM x 128 tile (BM = 128 / 64 / 32), K in steps of 32.  Operands are read from global
// memory as fp32 with either dimension contiguous (so the forward, data-gradient and weight-gradient products of the
// GEMM-lowered layers need no transposed copies), split in registers and stored to LDS as [row][32 k] bf16 planes
// whose 16-byte chunks are XOR-swizzled by the row (conflict-free ds_read_b128 without padding).  Loads of step k + 1
// are in flight while step k is multiplied (register staging, two LDS buffers).  A batch dimension (grid z) with
// element strides serves split-K weight gradients.
#include <type_traits>

#include "common.hpp"

using namespace atrihip;

struct Gemm3Args {
  const float* A;
  const float* B;
  float* C;
  const float* bias;       // [N] or nullptr
  long long a_row, a_k;    // element strides of A: A(m, k) = A[m * a_row + k * a_k]; one of them is 1
  long long b_k, b_col;    // element strides of B: B(k, n) = B[k * b_k + n * b_col]; one of them is 1
  long long ldc;           // C(m, n) = C[m * ldc + n]
  long long a_batch, b_batch, c_batch;
  int M, N, K;
  // B_MODE 2 / 3: B is the patch matrix of a 3x3, padding-1 convolution over the channel-major activation B = x[cin][nb][h][w],
  // generated while loading (never in memory): patch row (ci, ky, kx) = ci * 9 + ky * 3 + kx, patch column = position
  // (b * h + y) * w + x.  Mode 2: B(k, n) = patch[k][n] (forward / data gradient, N = positions).  Mode 3:
  // B(k, n) = patch[n][k_off + k] (weight gradient, K = a chunk of positions starting at k_off = blockIdx.z * b_batch).
  int cin, nb, h, w;
  // widest aligned access of each operand along its contiguous dimension: 4 (float4: strides, extents, base addresses all
  // multiples of 4 floats), 2 (all even) or 1.  Narrower accesses mask per pair / per element, so that rows of any
  // length and pitch work (the shared layer's 15 390 inputs, the value head's 51 atoms).
  int va, vb, vc;
};

namespace {
constexpr int kBK = 32;
constexpr int kBN = 128;
constexpr int kRowBytes = 64;  // one tile row: 32 bf16

// chunk (16 bytes = 8 k-values) c of tile row r lives at chunk c ^ swz(r): the four lane groups of a ds_read_b128
// (lanes {0-3, 12-15, 20-27}, ...) then touch 16 different 16-byte slots of the 256-byte bank row
__device__ __forceinline__ int swz(int r) { return (0x6C >> (((r >> 2) & 3) * 2)) & 3; }  // table {0, 3, 2, 1}

__device__ __forceinline__ void split4(const float4& v, uint2& hi, uint2& lo) {
  const unsigned h0 = pack_bf16x2(v.x, v.y), h1 = pack_bf16x2(v.z, v.w);
  const float r0 = v.x - bf16_to_float((unsigned short)(h0 & 0xFFFF)), r1 = v.y - bf16_to_float((unsigned short)(h0 >> 16));
  const float r2 = v.z - bf16_to_float((unsigned short)(h1 & 0xFFFF)), r3 = v.w - bf16_to_float((unsigned short)(h1 >> 16));
  hi = make_uint2(h0, h1);
  lo = make_uint2(pack_bf16x2(r0, r1), pack_bf16x2(r2, r3));
}

// Four consecutive floats at p of which the first `left` (may be <= 0 or > 4) exist, read in pieces of `vec` floats.
__device__ __forceinline__ float4 load_quad(const float* p, int left, int vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec == 4) {
    if (left > 0) v = *reinterpret_cast<const float4*>(p);
  } else if (vec == 2) {
    if (left > 0) {
      const float2 a = *reinterpret_cast<const float2*>(p);
      v.x = a.x; v.y = a.y;
    }
    if (left > 2) {
      const float2 b = *reinterpret_cast<const float2*>(p + 2);
      v.z = b.x; v.w = b.y;
    }
  } else {
    if (left > 0) v.x = p[0];
    if (left > 1) v.y = p[1];
    if (left > 2) v.z = p[2];
    if (left > 3) v.w = p[3];
  }
  return v;
}

// Global -> registers -> LDS for one ROWS x 32 operand tile.
// KCONTIG: a thread takes float4s along k (row r, k-quad c) and stores each as one 8-byte piece per plane.
// Otherwise the row dimension is the contiguous one: a thread takes a 4 x 4 block (k = 4 kq .. + 3, rows 4 rq .. + 3) as
// four float4s along the rows, transposes it in registers and stores one 8-byte piece per row and plane (one ds_write_b64
// per four values, where a naive transposition costs one ds_write_b16 per value).  Out-of-range rows / k read as zero.
template <int ROWS, bool KCONTIG>
struct TileLoader {
  static constexpr int kVec = KCONTIG ? ROWS * kBK / 4 / 256 : 4;  // float4 per thread
  static constexpr int kBlocks = 2 * ROWS;                         // 4 x 4 blocks per tile (row-contiguous form)
  float4 v[kVec];
  __device__ __forceinline__ void load(const float* base, long long s_row, long long s_k, int row0, int n_rows, int k0, int K, int tid,
                                       int vec) {
    if constexpr (KCONTIG) {
#pragma unroll
      for (int i = 0; i < kVec; ++i) {
        const int id = tid + i * 256;
        const int r = id >> 3, c = id & 7;
        const int left = row0 + r < n_rows ? K - (k0 + c * 4) : 0;
        v[i] = load_quad(base + (long long)(row0 + r) * s_row + (k0 + c * 4), left, vec);
      }
    } else {
      const int kq = tid / (ROWS / 4), rq = tid % (ROWS / 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int left = (tid < kBlocks && k0 + kq * 4 + i < K) ? n_rows - (row0 + rq * 4) : 0;
        v[i] = load_quad(base + (long long)(k0 + kq * 4 + i) * s_k + (row0 + rq * 4), left, vec);
      }
    }
  }
  // registers -> the (hi, lo) planes of one LDS tile buffer (lo plane ROWS * 64 bytes after the hi plane)
  __device__ __forceinline__ void store(unsigned char* tile, int tid) const {
    if constexpr (KCONTIG) {
#pragma unroll
      for (int i = 0; i < kVec; ++i) {
        const int id = tid + i * 256;
        const int r = id >> 3, c = id & 7;
        uint2 hi, lo;
        split4(v[i], hi, lo);
        unsigned char* p = tile + r * kRowBytes + (((c >> 1) ^ swz(r)) << 4) + ((c & 1) << 3);
        *reinterpret_cast<uint2*>(p) = hi;
        *reinterpret_cast<uint2*>(p + ROWS * kRowBytes) = lo;
      }
    } else {
      if (tid >= kBlocks) return;
      const int kq = tid / (ROWS / 4), rq = tid % (ROWS / 4);
      const float f[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                             {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = rq * 4 + j;
        uint2 hi, lo;
        split4(make_float4(f[0][j], f[1][j], f[2][j], f[3][j]), hi, lo);
        unsigned char* p = tile + r * kRowBytes + (((kq >> 1) ^ swz(r)) << 4) + ((kq & 1) << 3);
        *reinterpret_cast<uint2*>(p) = hi;
        *reinterpret_cast<uint2*>(p + ROWS * kRowBytes) = lo;
      }
    }
  }
};

__device__ __forceinline__ void div_small(int n, int d, float inv, int& q, int& r) {  // n < 2^24, d > 0
  q = (int)((float)n * inv);
  r = n - q * d;
  if (r < 0) {
    --q;
    r += d;
  } else if (r >= d) {
    ++q;
    r -= d;
  }
}

// Patch-matrix tile with the positions as tile rows (B_MODE 2).  A thread holds a 4 x 4 block: k = 4 kq .. + 3 and the
// four positions n0 + rq + 32 j (consecutive lanes read consecutive positions: coalesced 4-byte loads).  Its positions are
// fixed for the whole kernel, so which of the nine taps fall inside the board is decided once (one bit per tap).
// Loads are unconditional from a clamped index and masked afterwards (no divergent branches).  Two aligned 16-byte loads
// plus a select per four values instead of four 4-byte loads measured 25 % slower (more arithmetic per value).
struct PatchRowsLoader {
  float v[4][4];  // [k][position]
  int n[4];
  unsigned taps[4];
  __device__ __forceinline__ void init(const Gemm3Args& g, int n0, int tid) {
    const int rq = tid & 31, hw = g.h * g.w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      n[j] = n0 + rq + 32 * j;
      const int p = n[j] % hw, y = p / g.w, x = p % g.w;
      unsigned m = 0;
      if (n[j] < g.N) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
          if (yy >= 0 && yy < g.h && xx >= 0 && xx < g.w) m |= 1u << tp;
        }
      }
      taps[j] = m;
    }
  }
  __device__ __forceinline__ void load(const Gemm3Args& g, const float* X, int k0, int tid) {
    const int kq = tid >> 5;
    const int plane = g.nb * g.h * g.w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + kq * 4 + i;
      const int ci = k / 9, tp = k - ci * 9, ky = tp / 3, kx = tp - ky * 3;
      const int off = ci * plane + (ky - 1) * g.w + (kx - 1);
      const unsigned bit = k < g.K ? 1u << tp : 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = (taps[j] & bit) != 0;  // ok implies 0 <= off + n[j] < cin * plane
        const float val = X[ok ? off + n[j] : 0];
        v[i][j] = ok ? val : 0.f;
      }
    }
  }
  __device__ __forceinline__ void store(unsigned char* tile, int tid) const {
    const int kq = tid >> 5, rq = tid & 31;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = rq + 32 * j;
      uint2 hi, lo;
      split4(make_float4(v[0][j], v[1][j], v[2][j], v[3][j]), hi, lo);
      unsigned char* p = tile + r * kRowBytes + (((kq >> 1) ^ swz(r)) << 4) + ((kq & 1) << 3);
      *reinterpret_cast<uint2*>(p) = hi;
      *reinterpret_cast<uint2*>(p + kBN * kRowBytes) = lo;
    }
  }
};

// Patch-matrix tile with the patch rows (ci, ky, kx) as tile rows and positions along k (B_MODE 3): the register layout
// of TileLoader<128, true> - float4s over four consecutive positions.  A thread's four patch rows are fixed for the
// whole kernel; its four positions of a step are decoded once and shared by the four rows.
struct PatchColsLoader {
  TileLoader<kBN, true> t;
  int roff[4];
  int dy[4], dx[4];
  float inv_hw, inv_w;
  __device__ __forceinline__ void init(const Gemm3Args& g, int n0, int tid) {
    const int plane = g.nb * g.h * g.w;
    inv_hw = 1.0f / (float)(g.h * g.w);
    inv_w = 1.0f / (float)g.w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = n0 + ((tid + i * 256) >> 3);
      const int ci = row / 9, r = row - ci * 9, ky = r / 3, kx = r - ky * 3;
      dy[i] = row < g.N ? ky - 1 : 1 << 20;  // a patch row beyond the matrix fails every bounds test below
      dx[i] = kx - 1;
      roff[i] = ci * plane + (ky - 1) * g.w + (kx - 1);
    }
  }
  __device__ __forceinline__ void load(const Gemm3Args& g, const float* X, int k_off, int k0, int tid) {
    const int kl = k0 + (tid & 7) * 4;  // first of this thread's four positions, relative to the chunk
    int b, p, y, x;
    div_small(k_off + kl, g.h * g.w, inv_hw, b, p);
    div_small(p, g.w, inv_w, y, x);
    int ys[4], xs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ys[e] = kl + e < g.K ? y : -(1 << 21);
      xs[e] = x;
      if (++x == g.w) {
        x = 0;
        if (++y == g.h) y = 0;
      }
    }
    const int pos = k_off + kl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float f[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // ok implies a source inside the same board, i.e. inside the tensor
        const bool ok = (unsigned)(ys[e] + dy[i]) < (unsigned)g.h && (unsigned)(xs[e] + dx[i]) < (unsigned)g.w;
        const float val = X[ok ? pos + roff[i] + e : 0];
        f[e] = ok ? val : 0.f;
      }
      t.v[i] = make_float4(f[0], f[1], f[2], f[3]);
    }
  }
};

// MFMA fragment (hi, lo) of tile row `r` (16 rows per MFMA tile, lane l15 = row), k-chunk q4
template <int ROWS>
__device__ __forceinline__ void read_frag(const unsigned char* tile, int r, int q4, short8& hi, short8& lo) {
  const unsigned char* p = tile + r * kRowBytes + ((q4 ^ swz(r)) << 4);
  hi = *reinterpret_cast<const short8*>(p);
  lo = *reinterpret_cast<const short8*>(p + ROWS * kRowBytes);
}
}  // namespace

// MTW = MFMA row tiles per wave (4 / 2 / 1): BM = 32 * MTW rows of C per workgroup, waves arranged 2 (rows) x 2 (columns)
// B_MODE: 0 = B with n contiguous, 1 = B with k contiguous, 2 / 3 = patch matrix generated on the fly (Gemm3Args)
template <int MTW, bool A_KCONTIG, int B_MODE>
__global__ __launch_bounds__(256, 2) void gemm3_kernel(Gemm3Args g) {
  constexpr int BM = 32 * MTW;
  constexpr int kATile = 2 * BM * kRowBytes;   // hi + lo planes
  constexpr int kBTile = 2 * kBN * kRowBytes;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (kATile + kBTile)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * kBN;
  const float* A = g.A + (long long)blockIdx.z * g.a_batch;
  const float* B = B_MODE < 2 ? g.B + (long long)blockIdx.z * g.b_batch : g.B;
  float* C = g.C + (long long)blockIdx.z * g.c_batch;

  // Two register staging sets per operand: the loads of step s + 2 are issued before step s is multiplied and stored to
  // LDS after step s + 1 was, so a load has two steps (about 2 x 770 matrix-pipe cycles per wave) to come back.
  struct PlainB {
    TileLoader<kBN, B_MODE == 1> t;
  };
  using BLoader = typename std::conditional<B_MODE == 2, PatchRowsLoader, typename std::conditional<B_MODE == 3, PatchColsLoader, PlainB>::type>::type;
  TileLoader<BM, A_KCONTIG> la[2];
  BLoader lb[2];  // B tile rows are the n index
  if constexpr (B_MODE >= 2) {
    lb[0].init(g, n0, tid);
    lb[1] = lb[0];
  }
  const int k_off = B_MODE == 3 ? (int)(blockIdx.z * g.b_batch) : 0;
  // operand strides in the loader's terms: rows = m (A) / n (B)
  const long long a_row = g.a_row, a_k = g.a_k, b_row = g.b_col, b_k = g.b_k;
  auto load = [&](auto P, int k0) {
    constexpr int p = decltype(P)::value;
    la[p].load(A, a_row, a_k, m0, g.M, k0, g.K, tid, g.va);
    if constexpr (B_MODE == 2) lb[p].load(g, B, k0, tid);
    else if constexpr (B_MODE == 3) lb[p].load(g, B, k_off, k0, tid);
    else lb[p].t.load(B, b_row, b_k, n0, g.N, k0, g.K, tid, g.vb);
  };
  auto store = [&](auto P, unsigned char* buf) {
    constexpr int p = decltype(P)::value;
    la[p].store(buf, tid);
    if constexpr (B_MODE == 2) lb[p].store(buf + kATile, tid);
    else lb[p].t.store(buf + kATile, tid);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  float4v acc[MTW][4];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

  const int steps = (g.K + kBK - 1) / kBK;
  load(P0{}, 0);
  store(P0{}, smem);
  if (steps > 1) load(P1{}, kBK);
  __syncthreads();
  // step s: LDS buffer s & 1 holds its operands, staging set (s + 1) & 1 holds (in flight) those of step s + 1
  auto step = [&](auto P, int s) {
    constexpr int p = decltype(P)::value;
    using Other = std::integral_constant<int, 1 - p>;
    const unsigned char* at = smem + p * (kATile + kBTile);
    const unsigned char* bt = at + kATile;
    if (s + 2 < steps) load(P, (s + 2) * kBK);  // set p is free: step s went to LDS before step s - 1 was multiplied
    short8 ah[MTW], al[MTW], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < MTW; ++i) read_frag<BM>(at, (wm * MTW + i) * 16 + l15, q4, ah[i], al[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) read_frag<kBN>(bt, (wn * 4 + j) * 16 + l15, q4, bh[j], bl[j]);
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma_split(bh[j], bl[j], ah[i], al[i], acc[i][j]);  // rows = n, columns = m
    if (s + 1 < steps) store(Other{}, smem + (1 - p) * (kATile + kBTile));
    __syncthreads();
  };
  for (int s = 0; s < steps; s += 2) {
    step(P0{}, s);
    if (s + 1 < steps) step(P1{}, s + 1);
  }

  // acc[i][j][e] = C(m = m0 + (wm * MTW + i) * 16 + l15, n = n0 + (wn * 4 + j) * 16 + 4 q4 + e)
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int m = m0 + (wm * MTW + i) * 16 + l15;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + (wn * 4 + j) * 16 + q4 * 4;
      const int left = g.N - n;
      if (left <= 0) continue;
      float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (g.bias != nullptr) {
        const float4 b = load_quad(g.bias + n, left, g.vc);
        o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
      }
      float* c = C + (long long)m * g.ldc + n;
      if (g.vc == 4) {
        *reinterpret_cast<float4*>(c) = o;
      } else if (g.vc == 2) {
        *reinterpret_cast<float2*>(c) = make_float2(o.x, o.y);
        if (left > 2) *reinterpret_cast<float2*>(c + 2) = make_float2(o.z, o.w);
      } else {
        c[0] = o.x;
        if (left > 1) c[1] = o.y;
        if (left > 2) c[2] = o.z;
        if (left > 3) c[3] = o.w;
      }
    }
  }
}

namespace {
template <int MTW>
int launch_gemm3(const Gemm3Args& g, int batch, bool a_kc, int b_mode, hipStream_t stream) {
  const dim3 grid((g.N + kBN - 1) / kBN, (g.M + 32 * MTW - 1) / (32 * MTW), batch);
  switch ((a_kc ? 4 : 0) + b_mode) {
    case 0: hipLaunchKernelGGL((gemm3_kernel<MTW, false, 0>), grid, dim3(256), 0, stream, g); break;
    case 1: hipLaunchKernelGGL((gemm3_kernel<MTW, false, 1>), grid, dim3(256), 0, stream, g); break;
    case 4: hipLaunchKernelGGL((gemm3_kernel<MTW, true, 0>), grid, dim3(256), 0, stream, g); break;
    case 5: hipLaunchKernelGGL((gemm3_kernel<MTW, true, 1>), grid, dim3(256), 0, stream, g); break;
    case 6: hipLaunchKernelGGL((gemm3_kernel<MTW, true, 2>), grid, dim3(256), 0, stream, g); break;
    case 7: hipLaunchKernelGGL((gemm3_kernel<MTW, true, 3>), grid, dim3(256), 0, stream, g); break;
    default: return -15;
  }
  return hip_err(hipGetLastError());
}
int launch_for_m(const Gemm3Args& g, int batch, bool a_kc, int b_mode, hipStream_t stream) {
  if (g.M > 64) return launch_gemm3<4>(g, batch, a_kc, b_mode, stream);
  if (g.M > 32) return launch_gemm3<2>(g, batch, a_kc, b_mode, stream);
  return launch_gemm3<1>(g, batch, a_kc, b_mode, stream);
}
}  // namespace

extern "C" {

// C[b] = A[b] B[b] (+ bias) for b < batch, fp32 in / out, three-pass bf16 MFMA.  Strides in elements; exactly one of
// (a_row, a_k) and one of (b_k, b_col) must be 1.  Each operand is accessed with the widest vectors its pitch, extent
// along the contiguous dimension, base address and batch stride allow (16, 8 or 4 bytes); M, N, K are arbitrary.
int atri_gemm3(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long long a_row, long long a_k,
               long long b_k, long long b_col, long long ldc, int batch, long long a_batch, long long b_batch, long long c_batch,
               hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
  const bool a_kc = a_k == 1, b_kc = b_k == 1;
  if ((!a_kc && a_row != 1) || (!b_kc && b_col != 1)) return -10;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)bias) & 3) return -12;
  if (batch > 65535) return -14;
  // widest vector (in floats) that divides every quantity an access of the operand depends on
  auto width = [](long long pitch, long long extent, long long batch_stride, uintptr_t p0, uintptr_t p1) {
    const unsigned long long bits = (unsigned long long)pitch | (unsigned long long)extent | (unsigned long long)batch_stride |
                                    (unsigned long long)(p0 >> 2) | (unsigned long long)(p1 >> 2);
    return (bits & 3) == 0 ? 4 : (bits & 1) == 0 ? 2 : 1;
  };
  Gemm3Args g{A, B, C, bias, a_row, a_k, b_k, b_col, ldc, a_batch, b_batch, c_batch, M, N, K, 0, 0, 0, 0, 4, 4, 4};
  g.va = width(a_kc ? a_row : a_k, a_kc ? K : M, a_batch, (uintptr_t)A, 0);
  g.vb = width(b_kc ? b_col : b_k, b_kc ? K : N, b_batch, (uintptr_t)B, 0);
  g.vc = width(ldc, N, c_batch, (uintptr_t)C, (uintptr_t)bias);
  return launch_for_m(g, batch, a_kc, b_kc ? 1 : 0, stream);
}

// 3x3, padding-1 convolution products over a channel-major activation X = x[cin][nb][h][w] without a patch matrix in
// memory (A row-major [M, K], fp32 in / out, three-pass bf16 MFMA):
//   transposed = 0: C[M, nb*h*w] = A[M, cin*9] patches(X)                       (forward; data gradient with flipped weights)
//   transposed = 1: C[z][M, cin*9] = A[:, z*chunk : (z+1)*chunk] patches(X)[:, same]^T for z < positions / chunk
//                   (split-K weight gradient; the caller adds the z slices)
// cin and chunk must be multiples of 4, nb*h*w a multiple of 4 (and of chunk when transposed).
int atri_conv3_gemm(const float* A, const float* X, float* C, int M, int cin, int nb, int h, int w, long long lda, int transposed,
                    int chunk, hipStream_t stream) {
  const long long positions = (long long)nb * h * w;
  if (M <= 0 || cin <= 0 || positions <= 0) return 0;
  if ((cin & 3) || (positions & 3) || (lda & 3) || positions >= (1 << 24) || h < 1 || w < 1) return -20;
  if ((long long)cin * positions >= (1LL << 31)) return -20;  // 32-bit element offsets inside the loaders
  if (((uintptr_t)A | (uintptr_t)X | (uintptr_t)C) & 15) return -21;
  Gemm3Args g{};
  g.A = A; g.B = X; g.C = C; g.bias = nullptr;
  g.a_row = lda; g.a_k = 1;
  g.M = M; g.cin = cin; g.nb = nb; g.h = h; g.w = w;
  g.va = g.vb = g.vc = 4;
  if (!transposed) {
    g.N = (int)positions; g.K = cin * 9; g.ldc = positions;
    return launch_for_m(g, 1, true, 2, stream);
  }
  if (chunk <= 0 || (chunk & 3) || positions % chunk != 0 || positions / chunk > 65535) return -22;
  g.N = cin * 9; g.K = chunk; g.ldc = g.N;
  g.a_batch = chunk; g.b_batch = chunk; g.c_batch = (long long)M * g.N;
  return launch_for_m(g, (int)(positions / chunk), true, 3, stream);
}

}  // extern "C"
