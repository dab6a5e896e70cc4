// Learner-side fused kernels on channel-major activations [C, N] (N = batch * H * W contiguous per channel):
//
//   training-mode BatchNorm2d (+ skip connection) (+ activation), forward and backward
//   (alphatriangle/nn/model.py:15-38 conv_block, :41-60 ResidualBlock - the reference runs these as separate
//   library ops; alphatriangle/rl/core/trainer.py:204-310 is the step they belong to).
//
// In the channel-major layout a channel is one contiguous row, so per-channel statistics are plain row reductions.
// Each pass is two launches - per-(channel, chunk) partial sums, then the element-wise apply, which first folds
// the handful of partials of its channel - with no atomics and no memsets, so a captured learner step replays
// bit-exactly.  Statistics are fp32 for both element types; the one-pass variance uses sums shifted by the
// channel's first element (Σ(x - p), Σ(x - p)²), which removes the cancellation of the textbook E[x²] - E[x]² form.
#include "common.hpp"

using namespace atrihip;

namespace {
constexpr int kBnChunk = 2048;  // elements per workgroup: 256 threads x 8

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<unsigned short>(const unsigned short* p, float (&v)[8]) {
  const short8 h = *reinterpret_cast<const short8*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = bf16_to_float((unsigned short)h[j]);
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8<unsigned short>(unsigned short* p, const float (&v)[8]) {
  *reinterpret_cast<short8*>(p) = pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}
template <typename T>
__device__ __forceinline__ float load1(const T* p);
template <>
__device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load1<unsigned short>(const unsigned short* p) { return bf16_to_float(*p); }

// d act(z) / dz
__device__ __forceinline__ float act_grad(float z, int act) {
  switch (act) {
    case kActReLU: return z > 0.f ? 1.f : 0.f;
    case kActGELU: {
      const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752f));
      return cdf + z * 0.3989422804014327f * __expf(-0.5f * z * z);
    }
    case kActSiLU: {
      const float s = 1.0f / (1.0f + __expf(-z));
      return s * (1.0f + z * (1.0f - s));
    }
    case kActTanh: {
      const float t = tanhf(z);
      return 1.0f - t * t;
    }
    case kActSigmoid: {
      const float s = 1.0f / (1.0f + __expf(-z));
      return s * (1.0f - s);
    }
    default: return 1.f;
  }
}

// Two values per thread -> two block totals (every thread gets them).
__device__ __forceinline__ void block_sum2(float& a, float& b, float (&red)[2][4]) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = a;
    red[1][wave] = b;
  }
  __syncthreads();
  a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  __syncthreads();
}
}  // namespace

// partial[c][chunk] = (Σ(x - p), Σ(x - p)²) over the chunk, p = x[c][0]
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const T* __restrict__ x, int N, float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int c = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
  const T* row = x + (size_t)c * N;
  const float pivot = load1<T>(row);
  const int i0 = chunk * kBnChunk + threadIdx.x * 8;
  float s = 0.f, ss = 0.f;
  if (i0 < N) {
    float v[8];
    load8<T>(row + i0, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v[j] - pivot;
      s += d;
      ss += d * d;
    }
  }
  block_sum2(s, ss, red);
  if (threadIdx.x == 0) {
    partial[((size_t)c * chunks + chunk) * 2] = s;
    partial[((size_t)c * chunks + chunk) * 2 + 1] = ss;
  }
}

// y = act(gamma * (x - mean) * rstd + beta (+ r)); block (0, c) also writes the saved statistics and updates the
// running ones (unbiased variance, like torch.nn.BatchNorm2d); block (0, 0) bumps num_batches_tracked.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const T* __restrict__ x, const T* __restrict__ r,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ partial, int N, float eps, float momentum,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          long long* __restrict__ num_batches, T* __restrict__ y,
                                                          float* __restrict__ save_mean, float* __restrict__ save_rstd, int act) {
  const int c = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
  const T* row = x + (size_t)c * N;
  const float pivot = load1<T>(row);
  float s = 0.f, ss = 0.f;
  for (int k = 0; k < chunks; ++k) {  // same order in every block of the channel: identical statistics everywhere
    s += partial[((size_t)c * chunks + k) * 2];
    ss += partial[((size_t)c * chunks + k) * 2 + 1];
  }
  const float inv_n = 1.0f / (float)N;
  const float ms = s * inv_n;
  const float mean = pivot + ms;
  const float var = fmaxf(ss * inv_n - ms * ms, 0.f);
  const float rstd = rsqrtf(var + eps);
  if (chunk == 0 && threadIdx.x == 0) {
    save_mean[c] = mean;
    save_rstd[c] = rstd;
    if (running_mean != nullptr) {
      running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.0f - momentum) * running_var[c] + momentum * var * ((float)N / (float)max(N - 1, 1));
    }
    if (c == 0 && num_batches != nullptr) *num_batches += 1;
  }
  const int i0 = chunk * kBnChunk + threadIdx.x * 8;
  if (i0 >= N) return;
  const float g = gamma[c] * rstd, b = beta[c] - mean * g;
  float v[8], rv[8];
  load8<T>(row + i0, v);
  if (r != nullptr) load8<T>(r + (size_t)c * N + i0, rv);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float z = v[j] * g + b;
    if (r != nullptr) z += rv[j];
    v[j] = apply_act(z, act);
  }
  store8<T>(y + (size_t)c * N + i0, v);
}

// partial[c][chunk] = (Σ dz, Σ dz * xhat) with dz = dy * act'(z), z = gamma * xhat + beta (+ r)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ x, const T* __restrict__ r, const T* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, int N,
                                                            int act, float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int c = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
  const int i0 = chunk * kBnChunk + threadIdx.x * 8;
  float s = 0.f, sx = 0.f;
  if (i0 < N) {
    const float m = mean[c], rs = rstd[c], g = gamma[c], b = beta[c];
    float v[8], d[8], rv[8];
    load8<T>(x + (size_t)c * N + i0, v);
    load8<T>(dy + (size_t)c * N + i0, d);
    if (r != nullptr) load8<T>(r + (size_t)c * N + i0, rv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - m) * rs;
      float z = xh * g + b;
      if (r != nullptr) z += rv[j];
      const float dz = d[j] * act_grad(z, act);
      s += dz;
      sx += dz * xh;
    }
  }
  block_sum2(s, sx, red);
  if (threadIdx.x == 0) {
    partial[((size_t)c * chunks + chunk) * 2] = s;
    partial[((size_t)c * chunks + chunk) * 2 + 1] = sx;
  }
}

// dx = gamma * rstd * (dz - Σdz / N - xhat * Σ(dz xhat) / N); dr = dz; block (0, c) writes dgamma[c], dbeta[c]
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ r, const T* __restrict__ dy,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ partial, int N, int act, T* __restrict__ dx,
                                                          T* __restrict__ dr, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
  float s = 0.f, sx = 0.f;
  for (int k = 0; k < chunks; ++k) {
    s += partial[((size_t)c * chunks + k) * 2];
    sx += partial[((size_t)c * chunks + k) * 2 + 1];
  }
  if (chunk == 0 && threadIdx.x == 0) {
    dbeta[c] = s;
    dgamma[c] = sx;
  }
  const int i0 = chunk * kBnChunk + threadIdx.x * 8;
  if (i0 >= N) return;
  const float m = mean[c], rs = rstd[c], g = gamma[c], b = beta[c];
  const float inv_n = 1.0f / (float)N;
  const float k1 = s * inv_n, k2 = sx * inv_n, gs = g * rs;
  float v[8], d[8], rv[8], o[8];
  load8<T>(x + (size_t)c * N + i0, v);
  load8<T>(dy + (size_t)c * N + i0, d);
  if (r != nullptr) load8<T>(r + (size_t)c * N + i0, rv);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float xh = (v[j] - m) * rs;
    float z = xh * g + b;
    if (r != nullptr) z += rv[j];
    const float dz = d[j] * act_grad(z, act);
    d[j] = dz;
    o[j] = gs * (dz - k1 - xh * k2);
  }
  store8<T>(dx + (size_t)c * N + i0, o);
  if (dr != nullptr) store8<T>(dr + (size_t)c * N + i0, d);
}

namespace {
template <typename T>
int bn_fwd_t(const void* x, const void* r, const float* gamma, const float* beta, float* partial, int C, int N, float eps,
             float momentum, float* rm, float* rv, long long* nbt, void* y, float* save_mean, float* save_rstd, int act,
             hipStream_t s) {
  const dim3 grid((N + kBnChunk - 1) / kBnChunk, C);
  hipLaunchKernelGGL(bn_stats_partial_kernel<T>, grid, dim3(256), 0, s, (const T*)x, N, partial);
  hipLaunchKernelGGL(bn_apply_fwd_kernel<T>, grid, dim3(256), 0, s, (const T*)x, (const T*)r, gamma, beta, partial, N, eps,
                     momentum, rm, rv, nbt, (T*)y, save_mean, save_rstd, act);
  return (int)hipGetLastError();
}
template <typename T>
int bn_bwd_t(const void* x, const void* r, const void* dy, const float* gamma, const float* beta, const float* mean,
             const float* rstd, float* partial, int C, int N, int act, void* dx, void* dr, float* dgamma, float* dbeta,
             hipStream_t s) {
  const dim3 grid((N + kBnChunk - 1) / kBnChunk, C);
  hipLaunchKernelGGL(bn_bwd_partial_kernel<T>, grid, dim3(256), 0, s, (const T*)x, (const T*)r, (const T*)dy, gamma, beta, mean,
                     rstd, N, act, partial);
  hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, grid, dim3(256), 0, s, (const T*)x, (const T*)r, (const T*)dy, gamma, beta, mean,
                     rstd, partial, N, act, (T*)dx, (T*)dr, dgamma, dbeta);
  return (int)hipGetLastError();
}
}  // namespace

extern "C" {

// Number of floats the `partial` scratch needs for C channels of N elements.
int atri_bn_cm_partial_floats(int C, int N) { return C * ((N + kBnChunk - 1) / kBnChunk) * 2; }

// Training-mode BatchNorm (+ r) (+ act) of channel-major x [C, N]; N a multiple of 8.  r, running_mean / running_var and
// num_batches may be null.
int atri_bn_cm_fwd(const void* x, const void* r, int is_bf16, const float* gamma, const float* beta, float* partial, int C, int N,
                   float eps, float momentum, float* running_mean, float* running_var, long long* num_batches, void* y,
                   float* save_mean, float* save_rstd, int act, hipStream_t s) {
  if (C <= 0 || N <= 0) return 0;
  if ((N & 7) != 0 || C > 65535) return -10;
  return is_bf16 ? bn_fwd_t<unsigned short>(x, r, gamma, beta, partial, C, N, eps, momentum, running_mean, running_var, num_batches,
                                            y, save_mean, save_rstd, act, s)
                 : bn_fwd_t<float>(x, r, gamma, beta, partial, C, N, eps, momentum, running_mean, running_var, num_batches, y,
                                   save_mean, save_rstd, act, s);
}

// Backward of the same: dx (and dr = gradient of the skip connection when r was given), dgamma, dbeta.
int atri_bn_cm_bwd(const void* x, const void* r, const void* dy, int is_bf16, const float* gamma, const float* beta,
                   const float* mean, const float* rstd, float* partial, int C, int N, int act, void* dx, void* dr,
                   float* dgamma, float* dbeta, hipStream_t s) {
  if (C <= 0 || N <= 0) return 0;
  if ((N & 7) != 0 || C > 65535) return -10;
  return is_bf16 ? bn_bwd_t<unsigned short>(x, r, dy, gamma, beta, mean, rstd, partial, C, N, act, dx, dr, dgamma, dbeta, s)
                 : bn_bwd_t<float>(x, r, dy, gamma, beta, mean, rstd, partial, C, N, act, dx, dr, dgamma, dbeta, s);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Dropout fused with its neighbours in the encoder layer (alphatriangle/nn/model.py:189-196, the stock
// nn.TransformerEncoderLayer): "stream + dropout(branch)" and "dropout(activation(hidden))", forward and backward.
// The keep mask is a hash of (seed, element index), recomputed by the backward kernel: no mask tensor, no generator
// state, and a captured learner step draws a new seed tensor on every replay.  Elements in eights (n % 8 == 0).
namespace {
__device__ __forceinline__ unsigned drop_hash32(unsigned idx, unsigned s0, unsigned s1) {
  unsigned x = idx * 0x9E3779B1u + s0;
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  x ^= x >> 16; x += s1;
  x ^= x >> 15; x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  return x;
}
struct DropSeed {
  unsigned s0, s1;
};
__device__ __forceinline__ DropSeed drop_seed(const long long* seed) {
  const unsigned long long s = (unsigned long long)*seed;
  return {(unsigned)s, (unsigned)(s >> 32)};
}
}  // namespace

// out = a + dropout(b)  (dtype of a)
template <typename TA, typename TB>
__global__ __launch_bounds__(256) void add_dropout_fwd_kernel(const TA* __restrict__ a, const TB* __restrict__ b, TA* __restrict__ out,
                                                              const long long* __restrict__ seed, long long n, unsigned thresh, float scale) {
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i0 >= n) return;
  const DropSeed s = drop_seed(seed);
  float va[8], vb[8];
  load8<TA>(a + i0, va);
  load8<TB>(b + i0, vb);
#pragma unroll
  for (int j = 0; j < 8; ++j) va[j] += drop_hash32((unsigned)(i0 + j), s.s0, s.s1) >= thresh ? vb[j] * scale : 0.f;
  store8<TA>(out + i0, va);
}

// db = dropout'(g) = keep ? g * scale : 0  (g in the dtype of the sum, db in the dtype of the branch)
template <typename TA, typename TB>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const TA* __restrict__ g, TB* __restrict__ db, const long long* __restrict__ seed,
                                                          long long n, unsigned thresh, float scale) {
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i0 >= n) return;
  const DropSeed s = drop_seed(seed);
  float v[8];
  load8<TA>(g + i0, v);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = drop_hash32((unsigned)(i0 + j), s.s0, s.s1) >= thresh ? v[j] * scale : 0.f;
  store8<TB>(db + i0, v);
}

// out = dropout(act(h)); BWD: dh = keep ? g * scale * act'(h) : 0
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void act_dropout_kernel(const T* __restrict__ h, const T* __restrict__ g, T* __restrict__ out,
                                                          const long long* __restrict__ seed, long long n, unsigned thresh, float scale,
                                                          int act) {
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i0 >= n) return;
  const DropSeed s = thresh ? drop_seed(seed) : DropSeed{0u, 0u};
  float v[8], gv[8];
  load8<T>(h + i0, v);
  if (BWD) load8<T>(g + i0, gv);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool keep = thresh == 0u || drop_hash32((unsigned)(i0 + j), s.s0, s.s1) >= thresh;
    const float y = BWD ? gv[j] * act_grad(v[j], act) : apply_act(v[j], act);
    v[j] = keep ? y * scale : 0.f;
  }
  store8<T>(out + i0, v);
}

// Channel-major body output -> encoder tokens, with the positional table and its dropout in the same pass:
//   forward : tokens[p][b][c] = dropout(x[c][b * HW + p] + pe[p][c])        x: [D][B * HW] (T: fp32 or bf16), tokens: fp32
//   backward: dx[c][b * HW + p] = keep ? g[p][b][c] * scale : 0             (the table takes no gradient)
// (alphatriangle/nn/model.py:268-283 permutes, adds the table and applies nn.Dropout as three ops.)  One workgroup moves a
// 32 x 32 tile through LDS, so both sides are read and written in whole 128-byte runs; the keep mask hashes the flat
// token index, like the other dropout kernels here.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void tokens_pe_kernel(const T* __restrict__ x, const float* __restrict__ pe, const float* __restrict__ g,
                                                        float* __restrict__ tokens, T* __restrict__ dx, const long long* __restrict__ seed,
                                                        int D, int B, int HW, unsigned thresh, float scale) {
  __shared__ float tile[32][33];
  const int N = B * HW;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tid = threadIdx.x;
  const DropSeed s = thresh ? drop_seed(seed) : DropSeed{0u, 0u};
  // token side: thread = (position nl, four channels c4 ..)
  const int nl_t = tid >> 3, c4 = (tid & 7) * 4;
  const int n_t = n0 + nl_t;
  const int b_t = n_t / HW, p_t = n_t - b_t * HW;
  const long long tok = ((long long)p_t * B + b_t) * D + c0 + c4;
  // channel-major side: thread = (position nl, channels cr, cr + 8, ..)
  const int nl_c = tid & 31, cr = tid >> 5;
  if (!BWD) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = cr + 8 * i;
      tile[c][nl_c] = n0 + nl_c < N ? load1<T>(x + (size_t)(c0 + c) * N + n0 + nl_c) : 0.f;
    }
    __syncthreads();
    if (n_t < N) {
      const float4 pv = *reinterpret_cast<const float4*>(pe + (size_t)p_t * D + c0 + c4);
      float v[4] = {tile[c4][nl_t] + pv.x, tile[c4 + 1][nl_t] + pv.y, tile[c4 + 2][nl_t] + pv.z, tile[c4 + 3][nl_t] + pv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool keep = thresh == 0u || drop_hash32((unsigned)(tok + j), s.s0, s.s1) >= thresh;
        v[j] = keep ? v[j] * scale : 0.f;
      }
      *reinterpret_cast<float4*>(tokens + tok) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
    if (n_t < N) {
      const float4 gv = *reinterpret_cast<const float4*>(g + tok);
      const float v[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool keep = thresh == 0u || drop_hash32((unsigned)(tok + j), s.s0, s.s1) >= thresh;
        tile[c4 + j][nl_t] = keep ? v[j] * scale : 0.f;
      }
    }
    __syncthreads();
    if (n0 + nl_c < N) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = cr + 8 * i;
        const float v = tile[c][nl_c];
        if constexpr (sizeof(T) == 2) dx[(size_t)(c0 + c) * N + n0 + nl_c] = float_to_bf16(v);
        else dx[(size_t)(c0 + c) * N + n0 + nl_c] = v;
      }
    }
  }
}

extern "C" {

// out = a + dropout(b); a / out share a dtype, b may be bf16 next to fp32 a.  seed: one int64 on the device.
int atri_add_dropout_fwd(const void* a, int a_bf16, const void* b, int b_bf16, void* out, const long long* seed, long long n,
                         float p_drop, hipStream_t s) {
  if (n <= 0) return 0;
  if ((n & 7) || p_drop <= 0.f || p_drop >= 1.f || (a_bf16 && !b_bf16)) return (int)hipErrorInvalidValue;
  const unsigned thresh = (unsigned)((double)p_drop * 4294967296.0);
  const float scale = 1.0f / (1.0f - p_drop);
  const dim3 grid((unsigned)((n / 8 + 255) / 256));
  if (a_bf16)
    hipLaunchKernelGGL((add_dropout_fwd_kernel<unsigned short, unsigned short>), grid, dim3(256), 0, s, (const unsigned short*)a,
                       (const unsigned short*)b, (unsigned short*)out, seed, n, thresh, scale);
  else if (b_bf16)
    hipLaunchKernelGGL((add_dropout_fwd_kernel<float, unsigned short>), grid, dim3(256), 0, s, (const float*)a, (const unsigned short*)b,
                       (float*)out, seed, n, thresh, scale);
  else
    hipLaunchKernelGGL((add_dropout_fwd_kernel<float, float>), grid, dim3(256), 0, s, (const float*)a, (const float*)b, (float*)out, seed,
                       n, thresh, scale);
  return (int)hipGetLastError();
}

// db = keep ? g / (1 - p) : 0 with the mask of atri_add_dropout_fwd for the same seed.
int atri_dropout_bwd(const void* g, int g_bf16, void* db, int db_bf16, const long long* seed, long long n, float p_drop, hipStream_t s) {
  if (n <= 0) return 0;
  if ((n & 7) || p_drop <= 0.f || p_drop >= 1.f || (g_bf16 && !db_bf16)) return (int)hipErrorInvalidValue;
  const unsigned thresh = (unsigned)((double)p_drop * 4294967296.0);
  const float scale = 1.0f / (1.0f - p_drop);
  const dim3 grid((unsigned)((n / 8 + 255) / 256));
  if (g_bf16)
    hipLaunchKernelGGL((dropout_bwd_kernel<unsigned short, unsigned short>), grid, dim3(256), 0, s, (const unsigned short*)g,
                       (unsigned short*)db, seed, n, thresh, scale);
  else if (db_bf16)
    hipLaunchKernelGGL((dropout_bwd_kernel<float, unsigned short>), grid, dim3(256), 0, s, (const float*)g, (unsigned short*)db, seed, n,
                       thresh, scale);
  else
    hipLaunchKernelGGL((dropout_bwd_kernel<float, float>), grid, dim3(256), 0, s, (const float*)g, (float*)db, seed, n, thresh, scale);
  return (int)hipGetLastError();
}

// forward (g == nullptr): out = dropout(act(h)); backward: out = dropout'(g) * act'(h).  p_drop == 0: no dropout.
int atri_act_dropout(const void* h, const void* g, void* out, int is_bf16, const long long* seed, long long n, float p_drop, int act,
                     hipStream_t s) {
  if (n <= 0) return 0;
  if ((n & 7) || p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && seed == nullptr)) return (int)hipErrorInvalidValue;
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const float scale = 1.0f / (1.0f - p_drop);
  const dim3 grid((unsigned)((n / 8 + 255) / 256));
#define ATRI_ACT_DROPOUT(T, B) \
  hipLaunchKernelGGL((act_dropout_kernel<T, B>), grid, dim3(256), 0, s, (const T*)h, (const T*)g, (T*)out, seed, n, thresh, scale, act)
  if (is_bf16) {
    if (g) ATRI_ACT_DROPOUT(unsigned short, true); else ATRI_ACT_DROPOUT(unsigned short, false);
  } else {
    if (g) ATRI_ACT_DROPOUT(float, true); else ATRI_ACT_DROPOUT(float, false);
  }
#undef ATRI_ACT_DROPOUT
  return (int)hipGetLastError();
}

// Forward (g == nullptr): tokens [HW][B][D] (fp32) from x [D][B * HW] (fp32 or bf16) and the table pe [HW][D]; backward: dx from
// the token gradient g.  D a multiple of 32; p_drop == 0: no dropout (seed may be null).
int atri_tokens_pe(const void* x, int x_bf16, const float* pe, const float* g, float* tokens, void* dx, const long long* seed, int D,
                   int B, int HW, float p_drop, hipStream_t s) {
  if (D <= 0 || B <= 0 || HW <= 0) return 0;
  if ((D & 31) || p_drop < 0.f || p_drop >= 1.f || (p_drop > 0.f && seed == nullptr)) return (int)hipErrorInvalidValue;
  if ((long long)B * HW * D >= (1LL << 32)) return (int)hipErrorInvalidValue;  // the mask hashes 32-bit token indices
  const unsigned thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  const float scale = 1.0f / (1.0f - p_drop);
  const dim3 grid((B * HW + 31) / 32, D / 32);
#define ATRI_TOKENS(T, BWD) \
  hipLaunchKernelGGL((tokens_pe_kernel<T, BWD>), grid, dim3(256), 0, s, (const T*)x, pe, g, tokens, (T*)dx, seed, D, B, HW, thresh, scale)
  if (g == nullptr) {
    if (x_bf16) ATRI_TOKENS(unsigned short, false); else ATRI_TOKENS(float, false);
  } else {
    if (x_bf16) ATRI_TOKENS(unsigned short, true); else ATRI_TOKENS(float, true);
  }
#undef ATRI_TOKENS
  return (int)hipGetLastError();
}

}  // extern "C"
