// Learner-side multi-head self-attention (forward and backward) for 32-wide heads and up to 256 tokens, one 4-wave
// workgroup per (sample, head).  Scores, probabilities, the dropout mask and every gradient of them live in MFMA
// accumulator registers only: memory sees the packed QKV rows, the attention output, one log-sum-exp per query and the
// gradient of the QKV rows.  It replaces the batched-GEMM / softmax / dropout chain of
// nn.MultiheadAttention inside nn.TransformerEncoderLayer (alphatriangle/nn/model.py:189-196) in training.
//
// Operand precision follows the input type: bf16 rows use one bf16 MFMA pass, fp32 rows are split into (hi, lo) bf16
// planes and every product takes three passes (hi*hi + lo*hi + hi*lo, fp32 accumulation), which keeps fp32 training at
// fp32-grade accuracy on the bf16 matrix cores.
//
// Layouts.  An MFMA A / B fragment gives lane (l15 = lane & 15, q4 = lane >> 4) eight k-values of row / column l15;
// an accumulator gives it C[4 q4 + j][l15], j = 0..3.  Two operand images are built in LDS, both lane-linear
// ([fragment][plane][lane] x 16 B, conflict free):
//   row image  of M: fragment kt         = M[token 16 kt + l15][dims 8 q4 .. 8 q4 + 7]
//   transposed image: fragment (kb, n)   = M[token 32 kb + slot(q4, e)][dim 16 n + l15],  e = 0..7,
//                                          slot(q4, e) = e < 4 ? 4 q4 + e : 16 + 4 q4 + (e - 4),
// the slot order being the one in which two stacked 16 x 16 accumulators turn into a B fragment without any lane
// exchange.  Scores are computed transposed (keys x queries) where the wave owns queries and plain (queries x keys)
// where it owns keys, so that softmax sums and the products that follow need no transposition either.
//
// Dropout is a counter-based hash of (seed, sample-head, query, key), identical in both kernels; the seed is read from
// device memory so that a captured launch draws a new mask on every replay.
#include "common.hpp"

using namespace atrihip;

struct AttnArgs {
  const void* qkv;                  // [L * B, 3 * 32 * H], row = token * B + sample
  const void* dout;                 // backward: [L * B, 32 * H]
  void* out;                        // forward: [L * B, 32 * H]; backward: [L * B, 3 * 32 * H]
  float* lse;                       // [B * H, 16 * KT] log2-domain log-sum-exp of every query
  const unsigned long long* seed;   // dropout seed (device memory); unused when thresh == 0
  int L, B, H;
  float scale;                      // 1 / sqrt(32)
  unsigned thresh;                  // an element is dropped when its hash is below this
  float keep_scale;                 // 1 / (1 - p)
  int batch_major;                  // rows are sample * L + token instead of token * B + sample (a sample's rows contiguous)
};

namespace {
constexpr float kLog2e = 1.4426950408889634f;

template <int P>
__device__ __forceinline__ float4v mm(const short8 (&a)[P], const short8 (&b)[P], float4v acc) {
  if constexpr (P == 2) {
    return mfma_split(a[0], a[1], b[0], b[1], acc);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  }
}

template <int P>
__device__ __forceinline__ void make_frag(const float (&v)[8], short8 (&f)[P]) {
  if constexpr (P == 2) {
    split_bf16x8(v, f[0], f[1]);
  } else {
    f[0] = pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  }
}
template <int P>
__device__ __forceinline__ void make_frag(const float4v& a, const float4v& b, short8 (&f)[P]) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  make_frag<P>(v, f);
}

__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const unsigned short* p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const unsigned short* p) { return bf16_to_float(*p); }

__device__ __forceinline__ void store4(float* p, const float4v& v, float mul) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0] * mul, v[1] * mul, v[2] * mul, v[3] * mul);
}
__device__ __forceinline__ void store4(unsigned short* p, const float4v& v, float mul) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0] * mul, v[1] * mul), pack_bf16x2(v[2] * mul, v[3] * mul));
}

// Row-form fragment of the matrix at `base` (row pitch `pitch` elements): token 16 kt + l15, dims 8 q4 .. + 7.
template <typename T, int P>
__device__ __forceinline__ void row_frag(const T* base, size_t pitch, int token, int L, int q4, float mul, short8 (&f)[P]) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  if (token < L) {
    load8(base + (size_t)token * pitch + q4 * 8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= mul;
  }
  make_frag<P>(v, f);
}

template <typename T, int P, int KT>
__device__ __forceinline__ void build_row_image(short8* img, const T* base, size_t pitch, int L, float mul, int wave, int lane) {
  const int l15 = lane & 15, q4 = lane >> 4;
  for (int kt = wave; kt < KT; kt += 4) {
    short8 f[P];
    row_frag<T, P>(base, pitch, kt * 16 + l15, L, q4, mul, f);
#pragma unroll
    for (int p = 0; p < P; ++p) img[(size_t)(kt * P + p) * 64 + lane] = f[p];
  }
}

template <typename T, int P, int KT>
__device__ __forceinline__ void build_transposed_image(short8* img, const T* base, size_t pitch, int L, float mul, int wave, int lane) {
  const int l15 = lane & 15, q4 = lane >> 4;
  for (int fi = wave; fi < KT; fi += 4) {  // fragment (kb, n) = (fi >> 1, fi & 1)
    const int kb = fi >> 1, n = fi & 1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int token = kb * 32 + (e < 4 ? 4 * q4 + e : 12 + 4 * q4 + e);
      v[e] = token < L ? load1(base + (size_t)token * pitch + n * 16 + l15) * mul : 0.f;
    }
    short8 f[P];
    make_frag<P>(v, f);
#pragma unroll
    for (int p = 0; p < P; ++p) img[(size_t)(fi * P + p) * 64 + lane] = f[p];
  }
}

template <int P>
__device__ __forceinline__ void read_frag(const short8* img, int fi, int lane, short8 (&f)[P]) {
#pragma unroll
  for (int p = 0; p < P; ++p) f[p] = img[(size_t)(fi * P + p) * 64 + lane];
}

__device__ __forceinline__ unsigned drop_hash(unsigned idx, unsigned s0, unsigned s1) {
  unsigned x = idx * 0x9E3779B1u + s0;
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  x ^= x >> 16; x += s1;
  x ^= x >> 15; x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  return x;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------- forward
template <typename T, int KT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  constexpr int P = sizeof(T) == 4 ? 2 : 1;
  constexpr int LP = KT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  short8* kimg = reinterpret_cast<short8*>(smem);  // row image of K
  short8* vimg = kimg + (size_t)KT * P * 64;       // transposed image of V
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int D = a.H * 32, L = a.L;
  const size_t pitch = a.batch_major ? (size_t)3 * D : (size_t)a.B * 3 * D;
  const T* qbase = reinterpret_cast<const T*>(a.qkv) + (a.batch_major ? (size_t)b * L * 3 * D : (size_t)b * 3 * D) + h * 32;
  build_row_image<T, P, KT>(kimg, qbase + D, pitch, L, 1.0f, wave, lane);
  build_transposed_image<T, P, KT>(vimg, qbase + 2 * D, pitch, L, 1.0f, wave, lane);
  unsigned s0 = 0, s1 = 0;
  if (a.thresh) {
    const unsigned long long s = *a.seed;
    s0 = (unsigned)s;
    s1 = (unsigned)(s >> 32);
  }
  __syncthreads();

  T* obase = reinterpret_cast<T*>(a.out) + (a.batch_major ? (size_t)b * L * D : (size_t)b * D) + h * 32;
  const size_t opitch = a.batch_major ? (size_t)D : (size_t)a.B * D;
  for (int qt = wave; qt < KT; qt += 4) {
    if (qt * 16 >= L) break;  // wave-uniform
    const int query = qt * 16 + l15;
    short8 qf[P];
    row_frag<T, P>(qbase, pitch, query, L, q4, a.scale * kLog2e, qf);
    float4v st[KT];
    float m = -1e30f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      short8 kf[P];
      read_frag<P>(kimg, kt, lane, kf);
      st[kt] = mm<P>(kf, qf, (float4v){0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kt * 16 + q4 * 4 + j >= L) st[kt][j] = -1e30f;
        m = fmaxf(m, st[kt][j]);
      }
    }
    m = quad_row_max(m);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st[kt][j] = __builtin_amdgcn_exp2f(st[kt][j] - m);
        sum += st[kt][j];
      }
    sum = quad_row_sum(sum);
    if (q4 == 0) a.lse[(size_t)bh * LP + query] = m + __log2f(sum);
    const float inv = 1.0f / sum;
    const float kept = inv * a.keep_scale;
    const unsigned idx0 = ((unsigned)bh * LP + (unsigned)query) * LP;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float p = st[kt][j] * inv;
        if (a.thresh) p = drop_hash(idx0 + kt * 16 + q4 * 4 + j, s0, s1) >= a.thresh ? st[kt][j] * kept : 0.f;
        st[kt][j] = p;
      }
    float4v o[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kb = 0; kb < KT / 2; ++kb) {
      short8 pf[P];
      make_frag<P>(st[2 * kb], st[2 * kb + 1], pf);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        short8 vf[P];
        read_frag<P>(vimg, kb * 2 + n, lane, vf);
        o[n] = mm<P>(vf, pf, o[n]);
      }
    }
    if (query < L) {
#pragma unroll
      for (int n = 0; n < 2; ++n) store4(obase + (size_t)query * opitch + n * 16 + q4 * 4, o[n], 1.0f);
    }
  }
}

// --------------------------------------------------------------------------------------------------------- backward
template <typename T, int KT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a) {
  constexpr int P = sizeof(T) == 4 ? 2 : 1;
  constexpr int LP = KT * 16;
  constexpr size_t kImage = (size_t)KT * P * 64;  // short8 units
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  short8* img0 = reinterpret_cast<short8*>(smem);
  short8* img1 = img0 + kImage;
  short8* img2 = img1 + kImage;
  short8* img3 = img2 + kImage;
  float* lse_s = reinterpret_cast<float*>(img3 + kImage);
  float* delta_s = lse_s + LP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const int D = a.H * 32, L = a.L;
  const size_t pitch = a.batch_major ? (size_t)3 * D : (size_t)a.B * 3 * D, opitch = a.batch_major ? (size_t)D : (size_t)a.B * D;
  const size_t q_off = (a.batch_major ? (size_t)b * L * 3 * D : (size_t)b * 3 * D) + h * 32;
  const T* qbase = reinterpret_cast<const T*>(a.qkv) + q_off;
  const T* dobase = reinterpret_cast<const T*>(a.dout) + (a.batch_major ? (size_t)b * L * D : (size_t)b * D) + h * 32;
  T* gbase = reinterpret_cast<T*>(a.out) + q_off;
  const float s2 = a.scale * kLog2e;
  unsigned s0 = 0, s1 = 0;
  if (a.thresh) {
    const unsigned long long s = *a.seed;
    s0 = (unsigned)s;
    s1 = (unsigned)(s >> 32);
  }

  // ---- pass A: the wave owns query tiles.  Images: K rows, V rows, K transposed.
  build_row_image<T, P, KT>(img0, qbase + D, pitch, L, 1.0f, wave, lane);
  build_row_image<T, P, KT>(img1, qbase + 2 * D, pitch, L, 1.0f, wave, lane);
  build_transposed_image<T, P, KT>(img2, qbase + D, pitch, L, 1.0f, wave, lane);
  for (int i = tid; i < LP; i += 256) {
    lse_s[i] = 1e30f;  // a query beyond L: probability 0 in pass B
    delta_s[i] = 0.f;
  }
  __syncthreads();
  for (int qt = wave; qt < KT; qt += 4) {
    if (qt * 16 >= L) break;
    const int query = qt * 16 + l15;
    short8 qf[P], dof[P];
    row_frag<T, P>(qbase, pitch, query, L, q4, s2, qf);
    row_frag<T, P>(dobase, opitch, query, L, q4, 1.0f, dof);
    const float lse_q = query < L ? a.lse[(size_t)bh * LP + query] : 0.f;
    float4v st[KT], dp[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      short8 kf[P], vf[P];
      read_frag<P>(img0, kt, lane, kf);
      read_frag<P>(img1, kt, lane, vf);
      st[kt] = mm<P>(kf, qf, (float4v){0.f, 0.f, 0.f, 0.f});
      dp[kt] = mm<P>(vf, dof, (float4v){0.f, 0.f, 0.f, 0.f});
    }
    const unsigned idx0 = ((unsigned)bh * LP + (unsigned)query) * LP;
    float delta = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = kt * 16 + q4 * 4 + j;
        const float p = key < L ? __builtin_amdgcn_exp2f(st[kt][j] - lse_q) : 0.f;
        float g = dp[kt][j];
        if (a.thresh) g = drop_hash(idx0 + key, s0, s1) >= a.thresh ? g * a.keep_scale : 0.f;
        st[kt][j] = p;
        dp[kt][j] = g;       // gradient of the undropped probability
        delta += g * p;
      }
    delta = quad_row_sum(delta);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) st[kt][j] *= dp[kt][j] - delta;  // gradient of the logit
    float4v dq[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kb = 0; kb < KT / 2; ++kb) {
      short8 sf[P];
      make_frag<P>(st[2 * kb], st[2 * kb + 1], sf);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        short8 ktf[P];
        read_frag<P>(img2, kb * 2 + n, lane, ktf);
        dq[n] = mm<P>(ktf, sf, dq[n]);
      }
    }
    if (query < L) {
#pragma unroll
      for (int n = 0; n < 2; ++n) store4(gbase + (size_t)query * pitch + n * 16 + q4 * 4, dq[n], a.scale);
      if (q4 == 0) {
        lse_s[query] = lse_q;
        delta_s[query] = delta;
      }
    }
  }
  __syncthreads();  // every wave is past its reads of the pass-A images

  // ---- pass B: the wave owns key tiles.  Images: Q rows (log2-scaled), dO rows, Q transposed (scaled), dO transposed.
  build_row_image<T, P, KT>(img0, qbase, pitch, L, s2, wave, lane);
  build_row_image<T, P, KT>(img1, dobase, opitch, L, 1.0f, wave, lane);
  build_transposed_image<T, P, KT>(img2, qbase, pitch, L, a.scale, wave, lane);
  build_transposed_image<T, P, KT>(img3, dobase, opitch, L, 1.0f, wave, lane);
  __syncthreads();
  for (int jt = wave; jt < KT; jt += 4) {
    if (jt * 16 >= L) break;
    const int key = jt * 16 + l15;
    short8 kf[P], vf[P];
    row_frag<T, P>(qbase + D, pitch, key, L, q4, 1.0f, kf);
    row_frag<T, P>(qbase + 2 * D, pitch, key, L, q4, 1.0f, vf);
    float4v dv[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
    float4v dk[2] = {(float4v){0.f, 0.f, 0.f, 0.f}, (float4v){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ib = 0; ib < KT / 2; ++ib) {
      float4v pd[2], ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int it = ib * 2 + u;
        short8 qf[P], dof[P];
        read_frag<P>(img0, it, lane, qf);
        read_frag<P>(img1, it, lane, dof);
        const float4v s = mm<P>(qf, kf, (float4v){0.f, 0.f, 0.f, 0.f});
        const float4v g = mm<P>(dof, vf, (float4v){0.f, 0.f, 0.f, 0.f});
        const float4 lse4 = *reinterpret_cast<const float4*>(lse_s + it * 16 + q4 * 4);
        const float4 del4 = *reinterpret_cast<const float4*>(delta_s + it * 16 + q4 * 4);
        const float lq[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, dl[4] = {del4.x, del4.y, del4.z, del4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int query = it * 16 + q4 * 4 + j;
          const float p = key < L ? __builtin_amdgcn_exp2f(s[j] - lq[j]) : 0.f;
          float pk = p, gk = g[j];
          if (a.thresh) {
            const bool keep = drop_hash(((unsigned)bh * LP + (unsigned)query) * LP + key, s0, s1) >= a.thresh;
            pk = keep ? p * a.keep_scale : 0.f;
            gk = keep ? gk * a.keep_scale : 0.f;
          }
          pd[u][j] = pk;
          ds[u][j] = p * (gk - dl[j]);
        }
      }
      short8 pf[P], sf[P];
      make_frag<P>(pd[0], pd[1], pf);
      make_frag<P>(ds[0], ds[1], sf);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        short8 dotf[P], qtf[P];
        read_frag<P>(img3, ib * 2 + n, lane, dotf);
        read_frag<P>(img2, ib * 2 + n, lane, qtf);
        dv[n] = mm<P>(dotf, pf, dv[n]);
        dk[n] = mm<P>(qtf, sf, dk[n]);
      }
    }
    if (key < L) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        store4(gbase + (size_t)key * pitch + D + n * 16 + q4 * 4, dk[n], 1.0f);
        store4(gbase + (size_t)key * pitch + 2 * D + n * 16 + q4 * 4, dv[n], 1.0f);
      }
    }
  }
}

namespace {
template <typename T, int KT>
int launch_attn(const AttnArgs& a, bool backward, hipStream_t stream) {
  constexpr int P = sizeof(T) == 4 ? 2 : 1;
  const size_t image = (size_t)KT * P * 1024;
  const size_t lds = backward ? 4 * image + 2 * KT * 16 * sizeof(float) : 2 * image;
  const void* fn = backward ? (const void*)attn_bwd_kernel<T, KT> : (const void*)attn_fwd_kernel<T, KT>;
  static bool attr[2] = {false, false};
  if (!attr[backward]) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr[backward] = true;
  }
  if (backward) {
    hipLaunchKernelGGL((attn_bwd_kernel<T, KT>), dim3(a.B * a.H), dim3(256), lds, stream, a);
  } else {
    hipLaunchKernelGGL((attn_fwd_kernel<T, KT>), dim3(a.B * a.H), dim3(256), lds, stream, a);
  }
  return hip_err(hipGetLastError());
}

int dispatch_attn(const AttnArgs& a, int is_bf16, bool backward, hipStream_t stream) {
  const bool wide = a.L > 128;
  if (is_bf16) return wide ? launch_attn<unsigned short, 16>(a, backward, stream) : launch_attn<unsigned short, 8>(a, backward, stream);
  return wide ? launch_attn<float, 16>(a, backward, stream) : launch_attn<float, 8>(a, backward, stream);
}

int fill_args(AttnArgs& a, const void* qkv, const void* dout, void* out, float* lse, const void* seed, int L, int B, int H, float p_drop) {
  if (L < 1 || L > 256 || B < 1 || H < 1) return -10;
  const long long lp = L > 128 ? 256 : 128;
  if ((long long)B * H * lp * lp >= (1LL << 32)) return -11;  // the dropout counter is 32 bits wide
  if (!(p_drop >= 0.f && p_drop < 1.f)) return -12;
  if (p_drop > 0.f && seed == nullptr) return -13;
  a.qkv = qkv; a.dout = dout; a.out = out; a.lse = lse;
  a.seed = reinterpret_cast<const unsigned long long*>(seed);
  a.L = L; a.B = B; a.H = H;
  a.scale = 0.17677669529663687f;
  a.thresh = p_drop > 0.f ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  a.keep_scale = 1.0f / (1.0f - p_drop);
  return 0;
}
}  // namespace

extern "C" {

// Floats per (sample, head) row of the log-sum-exp buffer for sequences of L tokens.
int atri_attn_lse_pitch(int L) { return L > 128 ? 256 : 128; }

// ctx[L*B, 32H] = dropout(softmax(Q K^T / sqrt(32))) V per (sample, head) from packed qkv [L*B, 96H]; lse receives the
// log2-domain log-sum-exp of every query ([B*H, atri_attn_lse_pitch(L)] floats).
// batch_major: qkv / ctx rows are sample * L + token ([B, L, ...]) instead of token * B + sample ([L, B, ...]).
int atri_attn_fwd_layout(const void* qkv, int is_bf16, void* ctx, float* lse, const void* seed, int L, int B, int H, float p_drop,
                         int batch_major, hipStream_t stream) {
  AttnArgs a{};
  const int rc = fill_args(a, qkv, nullptr, ctx, lse, seed, L, B, H, p_drop);
  if (rc) return rc;
  a.batch_major = batch_major ? 1 : 0;
  return dispatch_attn(a, is_bf16, false, stream);
}

int atri_attn_fwd(const void* qkv, int is_bf16, void* ctx, float* lse, const void* seed, int L, int B, int H, float p_drop,
                  hipStream_t stream) {
  return atri_attn_fwd_layout(qkv, is_bf16, ctx, lse, seed, L, B, H, p_drop, 0, stream);
}

// dqkv[L*B, 96H] from qkv, the gradient of ctx and the forward's lse / seed / p_drop.  Every element of dqkv is written.
int atri_attn_bwd_layout(const void* qkv, const void* dctx, int is_bf16, void* dqkv, float* lse, const void* seed, int L, int B, int H,
                         float p_drop, int batch_major, hipStream_t stream) {
  AttnArgs a{};
  const int rc = fill_args(a, qkv, dctx, dqkv, lse, seed, L, B, H, p_drop);
  if (rc) return rc;
  a.batch_major = batch_major ? 1 : 0;
  return dispatch_attn(a, is_bf16, true, stream);
}

int atri_attn_bwd(const void* qkv, const void* dctx, int is_bf16, void* dqkv, float* lse, const void* seed, int L, int B, int H,
                  float p_drop, hipStream_t stream) {
  return atri_attn_bwd_layout(qkv, dctx, is_bf16, dqkv, lse, seed, L, B, H, p_drop, 0, stream);
}

}  // extern "C"
