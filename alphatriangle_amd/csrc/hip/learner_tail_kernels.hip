// Small learner-side kernels around the GEMMs of the training step (alphatriangle/rl/core/trainer.py:204-310 is the
// step; alphatriangle/nn/model.py:180-229 the shared layer and the two heads they serve):
//
//   sum_slices      out[f] = sum_s parts[s][f] (+ bias[f % period]): the second half of every split-K product (weight
//                   gradients over 30 720 rows, the shared layer's 15 390-long dot products)
//   bn1d_*          training-mode BatchNorm1d (+ activation) on row-major [B, C] rows, forward and backward, one launch each
//   conv_wt         the flipped, transposed filter [Cin, Cout * taps] a convolution's data gradient multiplies with
//   loss_reduce     the four scalars of the loss from its per-sample terms; scale_pair: the logit gradients times the
//                   incoming gradient of the total loss
//
// Like the other learner kernels they use neither atomics nor memsets, so a captured step holds kernel nodes only.
#include "common.hpp"

using namespace atrihip;

namespace {
template <typename T>
__device__ __forceinline__ float ld(const T* p);
template <>
__device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld<unsigned short>(const unsigned short* p) { return bf16_to_float(*p); }
template <typename T>
__device__ __forceinline__ void st(T* p, float v);
template <>
__device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st<unsigned short>(unsigned short* p, float v) { *p = float_to_bf16(v); }

__device__ __forceinline__ float act_slope(float z, int act) {  // d act(z) / dz
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
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// out[f] = sum over S slices (+ bias[f % period]).  fp32 slices whose length is a multiple of 4: a workgroup covers 64
// column quads and its four waves take every fourth slice each (four loads in flight per lane), then add up through LDS.
__global__ __launch_bounds__(256) void sum_slices4_kernel(const float* __restrict__ parts, float* __restrict__ out, int S, long long F,
                                                          const float* __restrict__ bias, int period) {
  __shared__ float4 sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long f = ((long long)blockIdx.x * 64 + lane) * 4;
  float4 a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (f < F) {
    const float* p = parts + f;
    int s = wave;
    for (; s + 12 < S; s += 16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(p + (long long)(s + 4 * i) * F);
        a[i].x += v.x; a[i].y += v.y; a[i].z += v.z; a[i].w += v.w;
      }
    }
    for (; s < S; s += 4) {
      const float4 v = *reinterpret_cast<const float4*>(p + (long long)s * F);
      a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
    }
  }
  sm[wave][lane] = make_float4((a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y),
                               (a[0].z + a[1].z) + (a[2].z + a[3].z), (a[0].w + a[1].w) + (a[2].w + a[3].w));
  __syncthreads();
  if (wave == 0 && f < F) {
    float4 t = sm[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      t.x += sm[w][lane].x; t.y += sm[w][lane].y; t.z += sm[w][lane].z; t.w += sm[w][lane].w;
    }
    if (bias != nullptr) {
      t.x += bias[f % period]; t.y += bias[(f + 1) % period]; t.z += bias[(f + 2) % period]; t.w += bias[(f + 3) % period];
    }
    *reinterpret_cast<float4*>(out + f) = t;
  }
}

// Any length, fp32 or bf16 slices: one column per thread.
template <typename T>
__global__ __launch_bounds__(256) void sum_slices_kernel(const T* __restrict__ parts, float* __restrict__ out, int S, long long F,
                                                         const float* __restrict__ bias, int period) {
  const long long f = (long long)blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  float a0 = 0.f, a1 = 0.f;
  int s = 0;
  for (; s + 1 < S; s += 2) {
    a0 += ld<T>(parts + (long long)s * F + f);
    a1 += ld<T>(parts + (long long)(s + 1) * F + f);
  }
  if (s < S) a0 += ld<T>(parts + (long long)s * F + f);
  float v = a0 + a1;
  if (bias != nullptr) v += bias[f % period];
  out[f] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm1d over the rows of x[B][C].  One workgroup per 16 channels: thread = (row group 0..15, channel 0..15), so a
// wave reads four 64-byte row segments per step.  The rows are read again for each pass (32 K values: they stay in L2).
namespace {
constexpr int kBnRows = 16;  // rows per thread held in registers (batches up to 256)
__device__ __forceinline__ float group_total(float v, float (&red)[16][17], int rg, int ch) {
  __syncthreads();  // previous use of `red`
  red[rg][ch] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += red[i][ch];
  return t;
}
}  // namespace

template <typename T>
__global__ __launch_bounds__(256) void bn1d_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int B, int C, float eps, float momentum, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, long long* __restrict__ num_batches,
                                                       T* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                       int act) {
  __shared__ float red[16][17];
  const int ch = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + ch;
  const bool live = c < C;
  // up to 16 * kBnRows rows: the thread's values stay in registers (one round of independent loads instead of three
  // dependent passes); longer batches read the rows again for each pass
  const bool cached = B <= 16 * kBnRows;
  float v[kBnRows];
  float s = 0.f;
  if (live) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < kBnRows; ++i) {
        const int r = rg + 16 * i;
        v[i] = r < B ? ld<T>(x + (size_t)r * C + c) : 0.f;
        s += v[i];
      }
    } else {
      for (int r = rg; r < B; r += 16) s += ld<T>(x + (size_t)r * C + c);
    }
  }
  const float mean = group_total(s, red, rg, ch) / (float)B;
  float ss = 0.f;
  if (live) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < kBnRows; ++i) {
        const float d = rg + 16 * i < B ? v[i] - mean : 0.f;
        ss += d * d;
      }
    } else {
      for (int r = rg; r < B; r += 16) {
        const float d = ld<T>(x + (size_t)r * C + c) - mean;
        ss += d * d;
      }
    }
  }
  const float var = group_total(ss, red, rg, ch) / (float)B;
  const float rstd = rsqrtf(var + eps);
  if (!live) return;
  const float g = gamma[c] * rstd, b = beta[c] - mean * g;
  if (cached) {
#pragma unroll
    for (int i = 0; i < kBnRows; ++i) {
      const int r = rg + 16 * i;
      if (r < B) st<T>(y + (size_t)r * C + c, apply_act(v[i] * g + b, act));
    }
  } else {
    for (int r = rg; r < B; r += 16) st<T>(y + (size_t)r * C + c, apply_act(ld<T>(x + (size_t)r * C + c) * g + b, act));
  }
  if (rg == 0) {
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    if (running_mean != nullptr) {
      const float unbiased = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
    if (num_batches != nullptr && c == 0) *num_batches += 1;
  }
}

// dx, dgamma, dbeta of y = act(gamma * xhat + beta), xhat = (x - mean) * rstd
template <typename T>
__global__ __launch_bounds__(256) void bn1d_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mean_in,
                                                       const float* __restrict__ rstd_in, int B, int C, int act, T* __restrict__ dx,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[16][17];
  const int ch = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + ch;
  const bool live = c < C;
  const float mean = live ? mean_in[c] : 0.f, rstd = live ? rstd_in[c] : 0.f;
  const float g = live ? gamma[c] : 0.f, b = live ? beta[c] : 0.f;
  const bool cached = B <= 16 * kBnRows;
  float xh_c[kBnRows], dz_c[kBnRows];
  float s1 = 0.f, s2 = 0.f;
  if (live) {
    if (cached) {
#pragma unroll
      for (int i = 0; i < kBnRows; ++i) {
        const int r = rg + 16 * i;
        const bool in = r < B;
        const float xv = in ? ld<T>(x + (size_t)r * C + c) : mean;
        const float dv = in ? ld<T>(dy + (size_t)r * C + c) : 0.f;
        xh_c[i] = (xv - mean) * rstd;
        dz_c[i] = dv * act_slope(g * xh_c[i] + b, act);
        s1 += dz_c[i];
        s2 += dz_c[i] * xh_c[i];
      }
    } else {
      for (int r = rg; r < B; r += 16) {
        const float xh = (ld<T>(x + (size_t)r * C + c) - mean) * rstd;
        const float dz = ld<T>(dy + (size_t)r * C + c) * act_slope(g * xh + b, act);
        s1 += dz;
        s2 += dz * xh;
      }
    }
  }
  const float sum_dz = group_total(s1, red, rg, ch);
  const float sum_dzx = group_total(s2, red, rg, ch);
  if (!live) return;
  const float inv_b = 1.0f / (float)B;
  if (cached) {
#pragma unroll
    for (int i = 0; i < kBnRows; ++i) {
      const int r = rg + 16 * i;
      if (r < B) st<T>(dx + (size_t)r * C + c, g * rstd * (dz_c[i] - sum_dz * inv_b - xh_c[i] * sum_dzx * inv_b));
    }
  } else {
    for (int r = rg; r < B; r += 16) {
      const float xh = (ld<T>(x + (size_t)r * C + c) - mean) * rstd;
      const float dz = ld<T>(dy + (size_t)r * C + c) * act_slope(g * xh + b, act);
      st<T>(dx + (size_t)r * C + c, g * rstd * (dz - sum_dz * inv_b - xh * sum_dzx * inv_b));
    }
  }
  if (rg == 0) {
    dgamma[c] = sum_dzx;
    dbeta[c] = sum_dz;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// out[ci][co * taps + t] = w[co][ci][taps - 1 - t]: the filter of the data gradient (a convolution of the output gradient
// with the spatially flipped filter, input and output channels swapped) as the row-major A operand of its GEMM.
template <typename T>
__global__ __launch_bounds__(256) void conv_wt_kernel(const T* __restrict__ w, T* __restrict__ out, int cout, int cin, int taps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cout * cin * taps) return;
  const int k = i % (cout * taps), ci = i / (cout * taps);
  const int co = k / taps, t = k - co * taps;
  out[i] = w[((size_t)co * cin + ci) * taps + (taps - 1 - t)];
}

// out = (total, policy, value, entropy): policy = mean(pce * w), value = mean(vce * w), entropy = mean(ent) (0 when its weight
// is 0), total = wp * policy + wv * value - we * entropy (alphatriangle/rl/core/trainer.py:250-272).  One workgroup.
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ pce, const float* __restrict__ vce,
                                                          const float* __restrict__ ent, const float* __restrict__ isw, int B, float wp,
                                                          float wv, float we, float* __restrict__ out) {
  __shared__ float red[3][4];
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float w = isw[i];
    a += pce[i] * w;
    b += vce[i] * w;
    c += ent[i];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
    red[2][threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float inv = 1.0f / (float)B;
    const float pl = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) * inv;
    const float vl = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) * inv;
    const float en = we > 0.f ? (red[2][0] + red[2][1] + red[2][2] + red[2][3]) * inv : 0.f;
    out[0] = wp * pl + wv * vl - we * en;
    out[1] = pl;
    out[2] = vl;
    out[3] = en;
  }
}

// oa = a * g[0], ob = b * g[0] (TO: float or bf16 outputs)
template <typename TO>
__global__ __launch_bounds__(256) void scale_pair_kernel(const float* __restrict__ a, long long na, const float* __restrict__ b,
                                                         long long nb, const float* __restrict__ g, TO* __restrict__ oa,
                                                         TO* __restrict__ ob) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const float s = g[0];
  if (i < na) st<TO>(oa + i, a[i] * s);
  else if (i < na + nb) st<TO>(ob + (i - na), b[i - na] * s);
}

extern "C" {

int atri_loss_reduce(const float* pce, const float* vce, const float* ent, const float* isw, int B, float wp, float wv, float we,
                     float* out, hipStream_t s) {
  if (B <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, pce, vce, ent, isw, B, wp, wv, we, out);
  return hip_err(hipGetLastError());
}

int atri_scale_pair(const float* a, long long na, const float* b, long long nb, const float* g, void* oa, void* ob, int out_bf16,
                    hipStream_t s) {
  if (na + nb <= 0) return 0;
  const dim3 grid((unsigned)((na + nb + 255) / 256));
  if (out_bf16)
    hipLaunchKernelGGL(scale_pair_kernel<unsigned short>, grid, dim3(256), 0, s, a, na, b, nb, g, (unsigned short*)oa, (unsigned short*)ob);
  else
    hipLaunchKernelGGL(scale_pair_kernel<float>, grid, dim3(256), 0, s, a, na, b, nb, g, (float*)oa, (float*)ob);
  return hip_err(hipGetLastError());
}

// parts: S contiguous slices of F values (fp32 or bf16); out: F fp32 values.  bias (optional) has `period` entries.
int atri_sum_slices(const void* parts, int is_bf16, float* out, int S, long long F, const float* bias, int period, hipStream_t s) {
  if (S <= 0 || F <= 0) return 0;
  if (bias != nullptr && period <= 0) return (int)hipErrorInvalidValue;
  const bool vec4 = !is_bf16 && (F & 3) == 0 && (((uintptr_t)parts | (uintptr_t)out) & 15) == 0;
  if (vec4)
    hipLaunchKernelGGL(sum_slices4_kernel, dim3((unsigned)((F / 4 + 63) / 64)), dim3(256), 0, s, (const float*)parts, out, S, F, bias,
                       period);
  else if (is_bf16)
    hipLaunchKernelGGL(sum_slices_kernel<unsigned short>, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, s,
                       (const unsigned short*)parts, out, S, F, bias, period);
  else
    hipLaunchKernelGGL(sum_slices_kernel<float>, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, s, (const float*)parts, out, S, F, bias,
                       period);
  return hip_err(hipGetLastError());
}

int atri_bn1d_fwd(const void* x, int is_bf16, const float* gamma, const float* beta, int B, int C, float eps, float momentum,
                  float* running_mean, float* running_var, long long* num_batches, void* y, float* mean, float* rstd, int act,
                  hipStream_t s) {
  if (B <= 0 || C <= 0) return (int)hipErrorInvalidValue;
  const dim3 grid((C + 15) / 16);
  if (is_bf16)
    hipLaunchKernelGGL(bn1d_fwd_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short*)x, gamma, beta, B, C, eps, momentum,
                       running_mean, running_var, num_batches, (unsigned short*)y, mean, rstd, act);
  else
    hipLaunchKernelGGL(bn1d_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)x, gamma, beta, B, C, eps, momentum, running_mean,
                       running_var, num_batches, (float*)y, mean, rstd, act);
  return hip_err(hipGetLastError());
}

int atri_bn1d_bwd(const void* x, const void* dy, int is_bf16, const float* gamma, const float* beta, const float* mean, const float* rstd,
                  int B, int C, int act, void* dx, float* dgamma, float* dbeta, hipStream_t s) {
  if (B <= 0 || C <= 0) return (int)hipErrorInvalidValue;
  const dim3 grid((C + 15) / 16);
  if (is_bf16)
    hipLaunchKernelGGL(bn1d_bwd_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short*)x, (const unsigned short*)dy, gamma,
                       beta, mean, rstd, B, C, act, (unsigned short*)dx, dgamma, dbeta);
  else
    hipLaunchKernelGGL(bn1d_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)dy, gamma, beta, mean, rstd, B, C, act,
                       (float*)dx, dgamma, dbeta);
  return hip_err(hipGetLastError());
}

int atri_conv_wt(const void* w, void* out, int is_bf16, int cout, int cin, int taps, hipStream_t s) {
  const long long n = (long long)cout * cin * taps;
  if (n <= 0) return 0;
  if (n >= (1LL << 31)) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (is_bf16)
    hipLaunchKernelGGL(conv_wt_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short*)w, (unsigned short*)out, cout, cin, taps);
  else
    hipLaunchKernelGGL(conv_wt_kernel<float>, grid, dim3(256), 0, s, (const float*)w, (float*)out, cout, cin, taps);
  return hip_err(hipGetLastError());
}

}  // extern "C"
