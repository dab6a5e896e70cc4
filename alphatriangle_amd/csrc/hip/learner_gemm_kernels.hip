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
  __device__ __forceinline__ void load(const float* base, long long s_row, long long s_k, int row0, int n_rows, int k0, int K, int tid) {
    if constexpr (KCONTIG) {
#pragma unroll
      for (int i = 0; i < kVec; ++i) {
        const int id = tid + i * 256;
        const int r = id >> 3, c = id & 7;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < n_rows && k0 + c * 4 < K) v[i] = *reinterpret_cast<const float4*>(base + (long long)(row0 + r) * s_row + (k0 + c * 4));
      }
    } else {
      const int kq = tid / (ROWS / 4), rq = tid % (ROWS / 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < kBlocks && row0 + rq * 4 < n_rows && k0 + kq * 4 + i < K)
          v[i] = *reinterpret_cast<const float4*>(base + (long long)(k0 + kq * 4 + i) * s_k + (row0 + rq * 4));
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

// MFMA fragment (hi, lo) of tile row `r` (16 rows per MFMA tile, lane l15 = row), k-chunk q4
template <int ROWS>
__device__ __forceinline__ void read_frag(const unsigned char* tile, int r, int q4, short8& hi, short8& lo) {
  const unsigned char* p = tile + r * kRowBytes + ((q4 ^ swz(r)) << 4);
  hi = *reinterpret_cast<const short8*>(p);
  lo = *reinterpret_cast<const short8*>(p + ROWS * kRowBytes);
}
}  // namespace

// MTW = MFMA row tiles per wave (4 / 2 / 1): BM = 32 * MTW rows of C per workgroup, waves arranged 2 (rows) x 2 (columns)
template <int MTW, bool A_KCONTIG, bool B_KCONTIG>
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
  const float* B = g.B + (long long)blockIdx.z * g.b_batch;
  float* C = g.C + (long long)blockIdx.z * g.c_batch;

  TileLoader<BM, A_KCONTIG> la;
  TileLoader<kBN, B_KCONTIG> lb;  // B tile rows are the n index
  // operand strides in the loader's terms: rows = m (A) / n (B)
  const long long a_row = g.a_row, a_k = g.a_k, b_row = g.b_col, b_k = g.b_k;

  float4v acc[MTW][4];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (float4v){0.f, 0.f, 0.f, 0.f};

  const int steps = (g.K + kBK - 1) / kBK;
  la.load(A, a_row, a_k, m0, g.M, 0, g.K, tid);
  lb.load(B, b_row, b_k, n0, g.N, 0, g.K, tid);
  la.store(smem, tid);
  lb.store(smem + kATile, tid);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const unsigned char* at = smem + (s & 1) * (kATile + kBTile);
    const unsigned char* bt = at + kATile;
    if (s + 1 < steps) {
      la.load(A, a_row, a_k, m0, g.M, (s + 1) * kBK, g.K, tid);
      lb.load(B, b_row, b_k, n0, g.N, (s + 1) * kBK, g.K, tid);
    }
    short8 ah[MTW], al[MTW], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < MTW; ++i) read_frag<BM>(at, (wm * MTW + i) * 16 + l15, q4, ah[i], al[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) read_frag<kBN>(bt, (wn * 4 + j) * 16 + l15, q4, bh[j], bl[j]);
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma_split(bh[j], bl[j], ah[i], al[i], acc[i][j]);  // rows = n, columns = m
    if (s + 1 < steps) {
      unsigned char* nt = smem + ((s + 1) & 1) * (kATile + kBTile);
      la.store(nt, tid);
      lb.store(nt + kATile, tid);
    }
    __syncthreads();
  }

  // acc[i][j][e] = C(m = m0 + (wm * MTW + i) * 16 + l15, n = n0 + (wn * 4 + j) * 16 + 4 q4 + e)
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    const int m = m0 + (wm * MTW + i) * 16 + l15;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + (wn * 4 + j) * 16 + q4 * 4;
      if (n >= g.N) continue;  // N is a multiple of 4
      float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (g.bias != nullptr) {
        const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
        o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
      }
      *reinterpret_cast<float4*>(C + (long long)m * g.ldc + n) = o;
    }
  }
}

namespace {
template <int MTW>
int launch_gemm3(const Gemm3Args& g, int batch, bool a_kc, bool b_kc, hipStream_t stream) {
  const dim3 grid((g.N + kBN - 1) / kBN, (g.M + 32 * MTW - 1) / (32 * MTW), batch);
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm3_kernel<MTW, true, true>), grid, dim3(256), 0, stream, g);
  else if (a_kc) hipLaunchKernelGGL((gemm3_kernel<MTW, true, false>), grid, dim3(256), 0, stream, g);
  else if (b_kc) hipLaunchKernelGGL((gemm3_kernel<MTW, false, true>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm3_kernel<MTW, false, false>), grid, dim3(256), 0, stream, g);
  return hip_err(hipGetLastError());
}
}  // namespace

extern "C" {

// C[b] = A[b] B[b] (+ bias) for b < batch, fp32 in / out, three-pass bf16 MFMA.  Strides in elements; exactly one of
// (a_row, a_k) and one of (b_k, b_col) must be 1, the other a multiple of 4; all base addresses, batch strides and ldc
// multiples of 4 elements (16-byte vector accesses); N a multiple of 4.  M, N, K need not be multiples of the tile sizes.
int atri_gemm3(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long long a_row, long long a_k,
               long long b_k, long long b_col, long long ldc, int batch, long long a_batch, long long b_batch, long long c_batch,
               hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
  const bool a_kc = a_k == 1, b_kc = b_k == 1;
  if ((!a_kc && a_row != 1) || (!b_kc && b_col != 1)) return -10;
  if (((a_kc ? a_row : a_k) & 3) || ((b_kc ? b_col : b_k) & 3) || (ldc & 3) || (N & 3)) return -11;
  if ((a_batch & 3) || (b_batch & 3) || (c_batch & 3)) return -11;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)bias) & 15) return -12;
  if (a_kc && (K & 3)) return -13;   // float4 along k
  if (!a_kc && (M & 3)) return -13;  // float4 along m
  if (b_kc && (K & 3)) return -13;
  if (batch > 65535) return -14;
  Gemm3Args g{A, B, C, bias, a_row, a_k, b_k, b_col, ldc, a_batch, b_batch, c_batch, M, N, K};
  if (M > 64) return launch_gemm3<4>(g, batch, a_kc, b_kc, stream);
  if (M > 32) return launch_gemm3<2>(g, batch, a_kc, b_kc, stream);
  return launch_gemm3<1>(g, batch, a_kc, b_kc, stream);
}

}  // extern "C"
