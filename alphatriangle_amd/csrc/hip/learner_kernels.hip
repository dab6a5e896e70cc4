// Learner-side fused kernels for gfx950: AlphaZero/C51 loss forward+backward, flat-buffer
// gradient-norm clipping + optimizer update, and the HBM-resident prioritized-replay sum tree.
//
// Definitions being implemented:
//   loss        alphatriangle/rl/core/trainer.py:159-271 (two-hot C51 projection, value CE,
//               policy CE, unweighted entropy bonus, importance-weighted means)
//   clip + opt  alphatriangle/rl/core/trainer.py:48-64,276-286 (Adam / AdamW / SGD momentum 0.9,
//               clip_grad_norm_)
//   PER         alphatriangle/utils/sumtree.py:25-98, alphatriangle/rl/core/buffer.py:96-195
#include "common.hpp"

#include <vector>

using namespace atrihip;

// ---------------------------------------------------------------------------
// One wave per sample.  Writes per-sample losses and the gradients of
//   total = wp * mean_i(w_i * pce_i) + wv * mean_i(w_i * vce_i) - we * mean_i(H_i)
// with respect to both logit tensors.
__global__ void az_loss_kernel(const float* __restrict__ plog, const float* __restrict__ vlog,
                               const float* __restrict__ ptgt, const float* __restrict__ ret,
                               const float* __restrict__ isw, float* __restrict__ pce, float* __restrict__ vce,
                               float* __restrict__ ent, float* __restrict__ dplog, float* __restrict__ dvlog, int B,
                               int A, int atoms, float vmin, float vmax, float wp, float wv, float we) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= B) return;
  const float* pl = plog + (size_t)wave * A;
  const float* pt = ptgt + (size_t)wave * A;
  const float w = isw[wave];
  const float invB = 1.0f / (float)B;
  // ---- policy head
  float m = -INFINITY;
  for (int a = lane; a < A; a += 64) m = fmaxf(m, pl[a]);
  m = wave_max(m);
  float se = 0.f;
  for (int a = lane; a < A; a += 64) se += __expf(pl[a] - m);
  se = wave_sum(se);
  const float lse = m + __logf(se);
  float ce = 0.f, H = 0.f, gbar = 0.f, tsum = 0.f;
  for (int a = lane; a < A; a += 64) {
    float t = pt[a];
    if (t != t) t = 0.f;  // nan_to_num on the target
    const float logp = pl[a] - lse;
    const float p = __expf(logp);
    ce -= t * logp;
    tsum += t;
    const float lpe = __logf(p + 1e-9f);
    H -= p * lpe;
    gbar += p * (-lpe - p / (p + 1e-9f));
  }
  ce = wave_sum(ce);
  H = wave_sum(H);
  gbar = wave_sum(gbar);
  tsum = wave_sum(tsum);
  const float cp = wp * w * invB, cee = we * invB;
  for (int a = lane; a < A; a += 64) {
    float t = pt[a];
    if (t != t) t = 0.f;
    const float p = __expf(pl[a] - lse);
    float gr = cp * (p * tsum - t);
    if (we > 0.f) {
      const float gk = -__logf(p + 1e-9f) - p / (p + 1e-9f);
      gr -= cee * p * (gk - gbar);
    }
    dplog[(size_t)wave * A + a] = gr;
  }
  // ---- value head: two-hot projection of the return onto the support
  const float* vl = vlog + (size_t)wave * atoms;
  const float dz = (vmax - vmin) / (float)(atoms - 1);
  const float g = fminf(fmaxf(ret[wave], vmin), vmax);
  const float b = (g - vmin) / dz;
  int lo = (int)floorf(b), hi = (int)ceilf(b);
  lo = min(max(lo, 0), atoms - 1);
  hi = min(max(hi, 0), atoms - 1);
  const float mlo = lo == hi ? 1.0f : (float)hi - b;
  const float mhi = lo == hi ? 0.0f : b - (float)lo;
  float vm = -INFINITY;
  for (int i = lane; i < atoms; i += 64) vm = fmaxf(vm, vl[i]);
  vm = wave_max(vm);
  float vs = 0.f;
  for (int i = lane; i < atoms; i += 64) vs += __expf(vl[i] - vm);
  vs = wave_sum(vs);
  const float vlse = vm + __logf(vs);
  float vc = 0.f;
  const float cv = wv * w * invB;
  for (int i = lane; i < atoms; i += 64) {
    const float tgt = (i == lo ? mlo : 0.f) + (i == hi ? mhi : 0.f);
    const float logq = vl[i] - vlse;
    vc -= tgt * logq;
    dvlog[(size_t)wave * atoms + i] = cv * (__expf(logq) - tgt);
  }
  vc = wave_sum(vc);
  if (lane == 0) {
    pce[wave] = ce;
    vce[wave] = vc;
    ent[wave] = H;
  }
}

// ---------------------------------------------------------------------------
// Two-stage deterministic sum of squares of the (scaled) flat gradient buffer.
__global__ void sumsq_partial_kernel(const float* __restrict__ g, long long n, float scale, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x * 4;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    float4 v = *reinterpret_cast<const float4*>(g + i);  // buffers are padded to multiples of 64 floats
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ norm_out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm_out[0] = sqrtf(red[0] + red[1] + red[2] + red[3]);
}

// kind: 0 = Adam (L2 in the gradient), 1 = AdamW (decoupled decay), 2 = SGD with momentum.
__global__ void flat_optimizer_kernel(int kind, float* __restrict__ p, const float* __restrict__ g,
                                      float* __restrict__ m, float* __restrict__ v, long long n,
                                      const float* __restrict__ norm, float lr, float b1, float b2, float eps,
                                      float wd, float mom, float bc1, float bc2_sqrt, float max_norm,
                                      float grad_scale, int first_step) {
  float coef = grad_scale;
  if (max_norm > 0.f) coef *= fminf(1.0f, max_norm / (norm[0] + 1e-6f));
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pv = p[i];
    float gv = g[i] * coef;
    if (kind == 2) {
      if (wd != 0.f) gv += wd * pv;
      float mv = first_step ? gv : mom * m[i] + gv;
      m[i] = mv;
      p[i] = pv - lr * mv;
    } else {
      if (kind == 1) {
        pv *= 1.0f - lr * wd;
      } else if (wd != 0.f) {
        gv += wd * pv;
      }
      const float mv = b1 * m[i] + (1.0f - b1) * gv;
      const float vv = b2 * v[i] + (1.0f - b2) * gv * gv;
      m[i] = mv;
      v[i] = vv;
      const float denom = sqrtf(vv) / bc2_sqrt + eps;
      p[i] = pv - (lr / bc1) * (mv / denom);
    }
  }
}

// ---------------------------------------------------------------------------
// Sum tree in HBM: power-of-two layout, root at 1, leaves at [P, 2P), float64 sums.
__global__ void per_set_leaves_kernel(double* tree, long long P, const int* idx, const int* count_ptr, int count_max,
                                      const float* prio_ptr) {
  const int n = count_ptr ? min(*count_ptr, count_max) : count_max;
  const double pr = (double)prio_ptr[0];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) tree[P + idx[i]] = pr;
}

// Recompute the ancestors of the given leaves, level by level, inside ONE workgroup (the
// barrier between levels makes every level-k sum visible before level k-1 reads it; __syncthreads orders the
// workgroup's global writes and reads).
__global__ void per_update_paths_kernel(double* tree, long long P, int levels, const int* idx, const int* count_ptr,
                                        int count_max) {
  const int n = count_ptr ? min(*count_ptr, count_max) : count_max;
  // Level-major: one barrier per tree level however many leaves there are, and a thread's loads of one level are all
  // in flight together.  Leaves that share an ancestor write the same sum to it.
  for (int l = 1; l <= levels; ++l) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const long long node = (P + idx[i]) >> l;
      tree[node] = tree[2 * node] + tree[2 * node + 1];
    }
    __syncthreads();
  }
}

// The same recomputation spread over the chip for trees of more than kPerTopLevels levels.  The tree is cut kPerTopLevels
// below the root: each of the 2^kPerTopLevels subtrees is walked by one workgroup, which first collects the touched leaves
// that fall into its subtree (a compaction through an LDS counter; a list that overflows falls back to re-scanning all
// indices per level) and then recomputes their ancestors level by level up to the subtree's root, with workgroup
// barriers only - no subtree depends on another one.  per_update_top_kernel then rebuilds the 2^kPerTopLevels - 1 nodes
// above the cut in one small workgroup.  No atomics on global memory, no flags between workgroups.
constexpr int kPerTopLevels = 8;
constexpr int kPerListCap = 8192;

__global__ __launch_bounds__(256) void per_update_subtrees_kernel(double* tree, long long P, int sub_levels, const int* idx,
                                                                  const int* count_ptr, int count_max) {
  __shared__ int list[kPerListCap];
  __shared__ int cnt;
  const int n = count_ptr ? min(*count_ptr, count_max) : count_max;
  const int tid = threadIdx.x;
  for (int sub = blockIdx.x; sub < (1 << kPerTopLevels); sub += gridDim.x) {
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
      const int leaf = idx[i];
      if ((leaf >> sub_levels) == sub) {
        const int k = atomicAdd(&cnt, 1);
        if (k < kPerListCap) list[k] = leaf;
      }
    }
    __syncthreads();
    const int m = cnt;
    if (m == 0) continue;  // workgroup-uniform
    if (m <= kPerListCap) {
      for (int l = 1; l <= sub_levels; ++l) {
        const int level_nodes = 1 << (sub_levels - l);  // nodes of this subtree at this level
        if (m >= level_nodes) {
          // more touched leaves than nodes: recompute the whole level of the subtree (each node once)
          const long long first = ((P >> l) + ((long long)sub << (sub_levels - l)));
          for (int j = tid; j < level_nodes; j += blockDim.x) {
            const long long node = first + j;
            tree[node] = tree[2 * node] + tree[2 * node + 1];
          }
        } else {
          for (int j = tid; j < m; j += blockDim.x) {
            const long long node = (P + list[j]) >> l;
            tree[node] = tree[2 * node] + tree[2 * node + 1];
          }
        }
        __syncthreads();
      }
    } else {
      for (int l = 1; l <= sub_levels; ++l) {
        for (int i = tid; i < n; i += blockDim.x) {
          const int leaf = idx[i];
          if ((leaf >> sub_levels) != sub) continue;
          const long long node = (P + leaf) >> l;
          tree[node] = tree[2 * node] + tree[2 * node + 1];
        }
        __syncthreads();
      }
    }
  }
}

__global__ __launch_bounds__(256) void per_update_top_kernel(double* tree) {
  for (int level = 1 << (kPerTopLevels - 1); level >= 1; level >>= 1) {
    for (int i = level + threadIdx.x; i < 2 * level; i += blockDim.x) tree[i] = tree[2 * i] + tree[2 * i + 1];
    __syncthreads();
  }
}

__global__ void per_rebuild_kernel(double* tree, long long P) {
  for (long long level = P >> 1; level >= 1; level >>= 1) {
    for (long long i = level + threadIdx.x; i < 2 * level; i += blockDim.x) tree[i] = tree[2 * i] + tree[2 * i + 1];
    __syncthreads();
  }
}

// Stratified proportional sampling of B leaves + importance weights normalised by the batch max.
// Single workgroup, B <= blockDim.x.
__global__ void per_sample_kernel(const double* tree, long long P, int size, int B, const float* rnd, float beta,
                                  int* out_idx, float* out_w) {
  __shared__ float red[16];
  const int i = threadIdx.x;
  float wgt = 0.f;
  if (i < B) {
    const double total = tree[1];
    const double seg = total / (double)B;
    double value = ((double)i + (double)rnd[i]) * seg;
    long long node = 1;
    while (node < P) {
      const double left = tree[2 * node];
      if (value <= left) {
        node = 2 * node;
      } else {
        value -= left;
        node = 2 * node + 1;
      }
    }
    int leaf = (int)(node - P);
    if (leaf >= size) leaf = size - 1;  // rounding at the upper edge of the mass
    const double pr = tree[P + leaf];
    const double prob = total > 0.0 ? pr / total : 0.0;
    wgt = prob > 1e-9 ? (float)pow((double)size * prob, -(double)beta) : 0.f;
    out_idx[i] = leaf;
  }
  float mx = wave_max(wgt);
  if ((i & 63) == 0) red[i >> 6] = mx;
  __syncthreads();
  float gm = 0.f;
  for (int k = 0; k < (blockDim.x >> 6); ++k) gm = fmaxf(gm, red[k]);
  if (i < B) out_w[i] = gm > 1e-9f ? wgt / gm : 1.0f;
}

// The same for batches above one workgroup: 256 samples per workgroup, raw weights and one maximum per workgroup first,
// the normalisation by the batch maximum in a second launch (per_sample_norm_kernel).
__global__ __launch_bounds__(256) void per_sample_multi_kernel(const double* tree, long long P, int size, int B, const float* rnd,
                                                               float beta, int* out_idx, float* out_w, float* wg_max) {
  __shared__ float red[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float wgt = 0.f;
  if (i < B) {
    const double total = tree[1];
    const double seg = total / (double)B;
    double value = ((double)i + (double)rnd[i]) * seg;
    long long node = 1;
    while (node < P) {
      const double left = tree[2 * node];
      if (value <= left) {
        node = 2 * node;
      } else {
        value -= left;
        node = 2 * node + 1;
      }
    }
    int leaf = (int)(node - P);
    if (leaf >= size) leaf = size - 1;  // rounding at the upper edge of the mass
    const double pr = tree[P + leaf];
    const double prob = total > 0.0 ? pr / total : 0.0;
    wgt = prob > 1e-9 ? (float)pow((double)size * prob, -(double)beta) : 0.f;
    out_idx[i] = leaf;
    out_w[i] = wgt;
  }
  const float mx = wave_max(wgt);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) wg_max[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void per_sample_norm_kernel(float* out_w, int B, const float* wg_max, int n_wg) {
  float gm = 0.f;
  for (int k = 0; k < n_wg; ++k) gm = fmaxf(gm, wg_max[k]);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < B) out_w[i] = gm > 1e-9f ? out_w[i] / gm : 1.0f;
}

// priority = (|err| + eps)^alpha, written to the leaves; tracks the running maximum.
__global__ void per_priorities_kernel(double* tree, long long P, const int* idx, const float* err, int n, float alpha,
                                      float eps, float* max_prio) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float e = fabsf(err[i]);
  float pr = powf(e + eps, alpha);
  if (!(pr == pr) || pr < eps) pr = eps;            // NaN / underflow
  if (isinf(pr)) pr = max_prio[0];                   // +inf -> current maximum
  tree[P + idx[i]] = (double)pr;
  atomicMax(reinterpret_cast<unsigned int*>(max_prio), __float_as_uint(pr));  // positive floats order like uints
}

// ---------------------------------------------------------------------------------------------------------------
// Column sums of a row-major [N, F] matrix (bias gradients of the encoder's linear layers: N = tokens * batch).
// Two plain stages - per-row-block partial sums, then their sum - with no atomics and no semaphores, so the op
// records nothing but two kernel nodes when the learner step is captured into a HIP graph.
template <typename T>
__device__ __forceinline__ float colsum_load(const T* p);
template <>
__device__ __forceinline__ float colsum_load<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float colsum_load<unsigned short>(const unsigned short* p) { return bf16_to_float(*p); }

template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ g, float* __restrict__ partial,
                                                             int N, int F, int rows_per_block) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(N, r0 + rows_per_block);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < F) {
    int r = r0 + wave;
    for (; r + 12 < r1; r += 16) {
      a0 += colsum_load(g + (size_t)r * F + col);
      a1 += colsum_load(g + (size_t)(r + 4) * F + col);
      a2 += colsum_load(g + (size_t)(r + 8) * F + col);
      a3 += colsum_load(g + (size_t)(r + 12) * F + col);
    }
    for (; r < r1; r += 4) a0 += colsum_load(g + (size_t)r * F + col);
  }
  sm[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && col < F) partial[(size_t)blockIdx.y * F + col] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                           int blocks, int F) {
  // 64 columns per workgroup; the four waves split the partial rows, four loads in flight per lane
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < F) {
    int b = wave;
    for (; b + 12 < blocks; b += 16) {
      a0 += partial[(size_t)b * F + col];
      a1 += partial[(size_t)(b + 4) * F + col];
      a2 += partial[(size_t)(b + 8) * F + col];
      a3 += partial[(size_t)(b + 12) * F + col];
    }
    for (; b < blocks; b += 4) a0 += partial[(size_t)b * F + col];
  }
  sm[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && col < F) out[col] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// The same sum for many partial rows: 16 columns per workgroup, the rows spread over 16 thread groups (64 columns per
// workgroup leave four workgroups for a 256-wide LayerNorm gradient, each walking 500 rows four at a time).
__global__ __launch_bounds__(256) void colsum_final16_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                             int blocks, int F) {
  __shared__ float sm[16][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < F) {
    int b = rg;
    for (; b + 48 < blocks; b += 64) {
      a0 += partial[(size_t)b * F + col];
      a1 += partial[(size_t)(b + 16) * F + col];
      a2 += partial[(size_t)(b + 32) * F + col];
      a3 += partial[(size_t)(b + 48) * F + col];
    }
    for (; b < blocks; b += 16) a0 += partial[(size_t)b * F + col];
  }
  sm[rg][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rg == 0 && col < F) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sm[i][cl];
    out[col] = t;
  }
}

static inline void launch_colsum_final(const float* partial, float* out, int blocks, int F, hipStream_t s) {
  if (blocks >= 48)
    hipLaunchKernelGGL(colsum_final16_kernel, dim3((F + 15) / 16), dim3(256), 0, s, partial, out, blocks, F);
  else
    hipLaunchKernelGGL(colsum_final_kernel, dim3((F + 63) / 64), dim3(256), 0, s, partial, out, blocks, F);
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm forward / backward for the learner (rows x D, D = 8 * LPR with LPR lanes per row, bf16 or fp32 rows,
// fp32 statistics).  torch's kernels take 34 us (forward), 40 + 28 us (backward) per call on [30720, 128] and run in
// fp32 under autocast with cast kernels on both sides; these move each row once per pass.
template <typename T>
__device__ __forceinline__ void ln_load8(const T* p, float (&f)[8]);
template <>
__device__ __forceinline__ void ln_load8<float>(const float* p, float (&f)[8]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <>
__device__ __forceinline__ void ln_load8<unsigned short>(const unsigned short* p, float (&f)[8]) {
  const uint4 q = *reinterpret_cast<const uint4*>(p);
  f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xFFFF0000u);
  f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xFFFF0000u);
  f[4] = __uint_as_float(q.z << 16); f[5] = __uint_as_float(q.z & 0xFFFF0000u);
  f[6] = __uint_as_float(q.w << 16); f[7] = __uint_as_float(q.w & 0xFFFF0000u);
}
template <typename T>
__device__ __forceinline__ void ln_store8(T* p, const float (&f)[8]);
template <>
__device__ __forceinline__ void ln_store8<float>(float* p, const float (&f)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}
template <>
__device__ __forceinline__ void ln_store8<unsigned short>(unsigned short* p, const float (&f)[8]) {
  *reinterpret_cast<short8*>(p) = pack_bf16x8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
}

// T: dtype of the rows; TO: dtype of the normalised rows (bf16 out of fp32 rows under autocast: the residual stream
// stays fp32 and the next GEMM reads bf16, without a cast kernel in between).
template <typename T, typename TO, int LPR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ b, TO* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out, int rows,
                                                     float eps, int perm_b) {
  constexpr int D = LPR * 8, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR;
  const int row = (blockIdx.x * 4 + wave) * RPW + lane / LPR;
  const bool live = row < rows;
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = 0.f;
  if (live) ln_load8<T>(x + (size_t)row * D + sub * 8, f);
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s += f[j];
    ss += f[j] * f[j];
  }
#pragma unroll
  for (int off = LPR / 2; off >= 1; off >>= 1) {
    s += __shfl_xor(s, off, 64);
    ss += __shfl_xor(ss, off, 64);
  }
  const float mean = s * (1.0f / D);
  const float rstd = rsqrtf(fmaxf(ss * (1.0f / D) - mean * mean, 0.f) + eps);
  if (!live) return;
  float gg[8], bb[8];
  ln_load8<float>(g + sub * 8, gg);
  ln_load8<float>(b + sub * 8, bb);
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * rstd * gg[j] + bb[j];
  // perm_b > 0: input row p * perm_b + b (token-major [tokens][batch]) goes to output row b * (rows / perm_b) + p
  // (batch-major), which is the flattening the layer after the encoder wants; statistics stay indexed by the input row
  const int orow = perm_b > 0 ? (row % perm_b) * (rows / perm_b) + row / perm_b : row;
  ln_store8<TO>(y + (size_t)orow * D + sub * 8, f);
  if (sub == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// dx for every row plus per-workgroup partial sums of dgamma / dbeta ([gridDim.x][2][D]); each workgroup walks
// rows_per_block rows so the partial buffer stays small.
// T: dtype of x and dx; TG: dtype of the incoming gradient (that of the forward's output).
template <typename T, typename TG, int LPR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, const TG* __restrict__ dy,
                                                     const float* __restrict__ g, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, T* __restrict__ dx,
                                                     float* __restrict__ partial, int rows, int rows_per_block, int perm_b,
                                                     const T* __restrict__ skip) {
  constexpr int D = LPR * 8, RPW = 64 / LPR, RPB = 4 * RPW;  // rows per pass of the workgroup
  __shared__ float red[2][RPB][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rloc = wave * RPW + lane / LPR;
  float gg[8], dg[8], db[8];
  ln_load8<float>(g + sub * 8, gg);
#pragma unroll
  for (int j = 0; j < 8; ++j) dg[j] = db[j] = 0.f;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(rows, r_begin + rows_per_block);
  for (int r0 = r_begin; r0 < r_end; r0 += RPB) {
    const int row = r0 + rloc;
    const bool live = row < r_end;
    float xv[8], dv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = dv[j] = 0.f;
    float mean = 0.f, rstd = 0.f;
    if (live) {
      ln_load8<T>(x + (size_t)row * D + sub * 8, xv);
      const int grow = perm_b > 0 ? (row % perm_b) * (rows / perm_b) + row / perm_b : row;  // the forward's output row
      ln_load8<TG>(dy + (size_t)grow * D + sub * 8, dv);
      mean = mean_in[row];
      rstd = rstd_in[row];
    }
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xv[j] = (xv[j] - mean) * rstd;  // x_hat
      dg[j] += dv[j] * xv[j];
      db[j] += dv[j];
      dv[j] *= gg[j];                 // dy * gamma
      c1 += dv[j];
      c2 += dv[j] * xv[j];
    }
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) {
      c1 += __shfl_xor(c1, off, 64);
      c2 += __shfl_xor(c2, off, 64);
    }
    c1 *= 1.0f / D;
    c2 *= 1.0f / D;
    if (live) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dv[j] = rstd * (dv[j] - c1 - xv[j] * c2);
      if (skip != nullptr) {  // gradient that reaches x past the normalisation (the residual connection): added here
        float sk[8];
        ln_load8<T>(skip + (size_t)row * D + sub * 8, sk);
#pragma unroll
        for (int j = 0; j < 8; ++j) dv[j] += sk[j];
      }
      ln_store8<T>(dx + (size_t)row * D + sub * 8, dv);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[0][rloc][sub * 8 + j] = dg[j];
    red[1][rloc][sub * 8 + j] = db[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    const int which = i / D, c = i - which * D;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < RPB; ++r) acc += red[which][r][c];
    partial[((size_t)blockIdx.x * 2 + which) * D + c] = acc;
  }
}

namespace {
template <typename T, typename TO, int LPR>
void ln_fwd_launch(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, int rows, float eps,
                   int perm_b, hipStream_t s) {
  constexpr int RPB = 4 * (64 / LPR);
  hipLaunchKernelGGL((ln_fwd_kernel<T, TO, LPR>), dim3((rows + RPB - 1) / RPB), dim3(256), 0, s, (const T*)x, g, b, (TO*)y, mean,
                     rstd, rows, eps, perm_b);
}
template <typename T, typename TG, int LPR>
void ln_bwd_launch(const void* x, const void* dy, const float* g, const float* mean, const float* rstd, void* dx,
                   float* partial, int rows, int rows_per_block, int blocks, int perm_b, const void* skip, hipStream_t s) {
  hipLaunchKernelGGL((ln_bwd_kernel<T, TG, LPR>), dim3(blocks), dim3(256), 0, s, (const T*)x, (const TG*)dy, g, mean, rstd, (T*)dx,
                     partial, rows, rows_per_block, perm_b, (const T*)skip);
}
}  // namespace


// ---------------------------------------------------------------------------------------------------------
// Patch matrix (im2col, stride 1) of a channel-major activation: x [C, B, H, W] bf16 -> cols [C*kh*kw, B*oh*ow],
// row (c, ky, kx), column (b, oy, ox), zeros where the tap falls off the board.  This is the B operand of the
// learner's convolution GEMMs (forward: the layer input; data gradient: the output gradient); channel-major
// activations make the GEMM output [Cout, B*oh*ow] the next layer's input as it stands, so the convolution body of
// the learner needs no layout copies.
// One workgroup = 2048 consecutive columns of all kh*kw rows of one channel.  The columns' source pixels are one
// contiguous run of whole boards of that channel: they are staged in LDS with coalesced loads, and the scattered
// 2-byte tap reads then hit LDS instead of global memory; one thread = 8 consecutive columns (the column decode is
// shared by the taps; every store is 16 bytes, coalesced along the row).
constexpr int kPatchCols = 2048;

template <typename T, int KH, int KW>
__global__ __launch_bounds__(256) void patch_matrix_kernel(const T* __restrict__ x, T* __restrict__ cols,
                                                          int B, int H, int W, int ph, int pw, int oh, int ow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  T* tile = reinterpret_cast<T*>(tile_raw);
  const int c = blockIdx.y, tid = threadIdx.x;
  const int plane = oh * ow, n_total = B * plane;
  const int n_begin = blockIdx.x * kPatchCols, n_end = min(n_total, n_begin + kPatchCols);
  const int b_lo = n_begin / plane, b_hi = (n_end - 1) / plane;
  const int board = H * W, count = (b_hi - b_lo + 1) * board;
  const T* src = x + ((size_t)c * B + b_lo) * board;
  if (sizeof(T) == 2 && ((((size_t)c * B + b_lo) * board) & 1) == 0) {  // 2-byte elements on a 4-byte boundary: dword loads
    const unsigned* src2 = reinterpret_cast<const unsigned*>(src);
    unsigned* tile2 = reinterpret_cast<unsigned*>(tile);
    for (int i = tid; i < (count >> 1); i += 256) tile2[i] = src2[i];
    if ((count & 1) && tid == 0) tile[count - 1] = src[count - 1];
  } else {
    for (int i = tid; i < count; i += 256) tile[i] = src[i];
  }
  __syncthreads();
  const int n0 = n_begin + (tid << 3);
  if (n0 >= n_end) return;  // after the only barrier
  int b = n0 / plane, rem = n0 - b * plane;
  int oy = rem / ow, ox = rem - oy * ow;
  b -= b_lo;
  int iy[8], ix[8], base[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    iy[i] = oy - ph;
    ix[i] = ox - pw;
    base[i] = (b * H + iy[i]) * W + ix[i];
    if (++ox == ow) {
      ox = 0;
      if (++oy == oh) {
        oy = 0;
        ++b;
      }
    }
  }
  const size_t row_pitch = (size_t)n_total;
  T* out = cols + (size_t)c * KH * KW * row_pitch + n0;
#pragma unroll
  for (int ky = 0; ky < KH; ++ky)
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) {
      T v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int yy = iy[i] + ky, xx = ix[i] + kx;
        const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        v[i] = in ? tile[base[i] + ky * W + kx] : (T)0;
      }
      T* o = out + (size_t)(ky * KW + kx) * row_pitch;
      if (sizeof(T) == 2) {
        uint4 q;
        q.x = (unsigned)v[0] | ((unsigned)v[1] << 16);
        q.y = (unsigned)v[2] | ((unsigned)v[3] << 16);
        q.z = (unsigned)v[4] | ((unsigned)v[5] << 16);
        q.w = (unsigned)v[6] | ((unsigned)v[7] << 16);
        *reinterpret_cast<uint4*>(o) = q;
      } else {
        float4 q0, q1;
        q0.x = (float)v[0]; q0.y = (float)v[1]; q0.z = (float)v[2]; q0.w = (float)v[3];
        q1.x = (float)v[4]; q1.y = (float)v[5]; q1.z = (float)v[6]; q1.w = (float)v[7];
        reinterpret_cast<float4*>(o)[0] = q0;
        reinterpret_cast<float4*>(o)[1] = q1;
      }
    }
}

template <typename T, int KH, int KW>
static int launch_patch_matrix(const void* x, void* cols, int C, int B, int H, int W, int ph, int pw, hipStream_t s) {
  const int oh = H + 2 * ph - KH + 1, ow = W + 2 * pw - KW + 1;
  if (oh <= 0 || ow <= 0 || (((long long)B * oh * ow) & 7) != 0) return -2;
  if ((long long)C * B * H * W > 0x7fffffffLL || (long long)B * oh * ow > 0x7fffffffLL || C > 65535) return -3;
  const int n_total = B * oh * ow;
  const int boards = kPatchCols / (oh * ow) + 2;  // most boards one workgroup's columns can touch
  const size_t lds = ((size_t)boards * H * W * sizeof(T) + 15) & ~(size_t)15;
  if (lds > 64 * 1024) return -4;
  hipLaunchKernelGGL((patch_matrix_kernel<T, KH, KW>), dim3((n_total + kPatchCols - 1) / kPatchCols, C), dim3(256), lds, s,
                     (const T*)x, (T*)cols, B, H, W, ph, pw, oh, ow);
  return (int)hipGetLastError();
}

extern "C" {

int atri_az_loss(const float* plog, const float* vlog, const float* ptgt, const float* ret, const float* isw,
                 float* pce, float* vce, float* ent, float* dplog, float* dvlog, int B, int A, int atoms, float vmin,
                 float vmax, float wp, float wv, float we, hipStream_t s) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(az_loss_kernel, dim3((B + 3) / 4), dim3(256), 0, s, plog, vlog, ptgt, ret, isw, pce, vce, ent,
                     dplog, dvlog, B, A, atoms, vmin, vmax, wp, wv, we);
  return (int)hipGetLastError();
}

int atri_flat_optimizer(int kind, float* p, const float* g, float* m, float* v, long long n, float* norm_buf,
                        float* partial, int partial_cap, float lr, float b1, float b2, float eps, float wd, float mom,
                        float bc1, float bc2_sqrt, float max_norm, float grad_scale, int first_step, hipStream_t s) {
  if (n <= 0) return 0;
  if ((n & 3) != 0) return -10;
  int nb = (int)((n / 4 + 255) / 256);
  if (nb > partial_cap) nb = partial_cap;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, s, g, n, grad_scale, partial);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, partial, nb, norm_buf);
  int ob = (int)((n + 255) / 256);
  if (ob > 4096) ob = 4096;
  hipLaunchKernelGGL(flat_optimizer_kernel, dim3(ob), dim3(256), 0, s, kind, p, g, m, v, n, norm_buf, lr, b1, b2, eps,
                     wd, mom, bc1, bc2_sqrt, max_norm, grad_scale, first_step);
  return (int)hipGetLastError();
}

int atri_per_set_leaves(double* tree, long long P, const int* idx, const int* count_ptr, int count_max,
                        const float* prio_ptr, hipStream_t s) {
  if (count_max <= 0) return 0;
  int nb = (count_max + 255) / 256;
  if (nb > 256) nb = 256;
  hipLaunchKernelGGL(per_set_leaves_kernel, dim3(nb), dim3(256), 0, s, tree, P, idx, count_ptr, count_max, prio_ptr);
  return (int)hipGetLastError();
}
int atri_per_update_paths(double* tree, long long P, const int* idx, const int* count_ptr, int count_max,
                          hipStream_t s) {
  if (count_max <= 0) return 0;
  int levels = 0;
  while ((1LL << levels) < P) ++levels;
  if (levels <= kPerTopLevels + 2 || count_max <= 256) {
    // small trees, or so few leaves that one workgroup walks them faster than 256 workgroups scan for them
    hipLaunchKernelGGL(per_update_paths_kernel, dim3(1), dim3(1024), 0, s, tree, P, levels, idx, count_ptr, count_max);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(per_update_subtrees_kernel, dim3(1 << kPerTopLevels), dim3(256), 0, s, tree, P, levels - kPerTopLevels, idx,
                     count_ptr, count_max);
  hipLaunchKernelGGL(per_update_top_kernel, dim3(1), dim3(256), 0, s, tree);
  return (int)hipGetLastError();
}
int atri_per_rebuild(double* tree, long long P, hipStream_t s) {
  hipLaunchKernelGGL(per_rebuild_kernel, dim3(1), dim3(1024), 0, s, tree, P);
  return (int)hipGetLastError();
}
int atri_per_sample(const double* tree, long long P, int size, int B, const float* rnd, float beta, int* out_idx,
                    float* out_w, hipStream_t s) {
  if (B <= 0 || B > 1024 || size <= 0) return -10;
  int threads = ((B + 63) / 64) * 64;
  hipLaunchKernelGGL(per_sample_kernel, dim3(1), dim3(threads), 0, s, tree, P, size, B, rnd, beta, out_idx, out_w);
  return (int)hipGetLastError();
}
// Batches of any size: wg_max is scratch of ceil(B / 256) floats.
int atri_per_sample_large(const double* tree, long long P, int size, int B, const float* rnd, float beta, int* out_idx,
                          float* out_w, float* wg_max, hipStream_t s) {
  if (B <= 0 || size <= 0 || wg_max == nullptr) return -10;
  const int nb = (B + 255) / 256;
  hipLaunchKernelGGL(per_sample_multi_kernel, dim3(nb), dim3(256), 0, s, tree, P, size, B, rnd, beta, out_idx, out_w, wg_max);
  hipLaunchKernelGGL(per_sample_norm_kernel, dim3(nb), dim3(256), 0, s, out_w, B, wg_max, nb);
  return (int)hipGetLastError();
}
int atri_per_priorities(double* tree, long long P, const int* idx, const float* err, int n, float alpha, float eps,
                        float* max_prio, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(per_priorities_kernel, dim3((n + 255) / 256), dim3(256), 0, s, tree, P, idx, err, n, alpha, eps,
                     max_prio);
  return (int)hipGetLastError();
}

// out[F] = column sums of g[N, F]; partial holds ceil(N / rows_per_block) * F floats.
int atri_colsum(const void* g, int is_bf16, float* partial, float* out, int N, int F, int rows_per_block,
                hipStream_t s) {
  if (N <= 0 || F <= 0 || rows_per_block <= 0) return (int)hipErrorInvalidValue;
  if (!is_bf16 && N <= 1024) {  // few rows (the heads' [batch, features] gradients): the second stage alone adds them all
    launch_colsum_final((const float*)g, out, N, F, s);
    return (int)hipGetLastError();
  }
  const int blocks = (N + rows_per_block - 1) / rows_per_block;
  dim3 grid((F + 63) / 64, blocks);
  if (is_bf16)
    hipLaunchKernelGGL(colsum_partial_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short*)g, partial,
                       N, F, rows_per_block);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)g, partial, N, F,
                       rows_per_block);
  launch_colsum_final(partial, out, blocks, F, s);
  return (int)hipGetLastError();
}

// y = LayerNorm(x) over the last dimension D in {64, 128, 256, 512}; mean / rstd [rows] are saved for the backward pass.
// is_bf16: dtype of x; out_bf16: dtype of y (fp32 rows may be written as bf16, not the other way round).
// perm_b > 0 (rows a multiple of it): the rows are token-major [rows / perm_b tokens][perm_b samples] and y is written
// batch-major [perm_b][rows / perm_b] - LayerNorm and the encoder's final flattening in one pass.
int atri_ln_fwd_perm(const void* x, int is_bf16, const float* g, const float* b, void* y, int out_bf16, float* mean, float* rstd,
                     int rows, int D, float eps, int perm_b, hipStream_t s) {
  if (rows <= 0) return 0;
  if (is_bf16 && !out_bf16) return (int)hipErrorInvalidValue;
  if (perm_b < 0 || (perm_b > 0 && rows % perm_b != 0)) return (int)hipErrorInvalidValue;
#define ATRI_LN_FWD(LPR)                                                                                             \
  if (is_bf16) ln_fwd_launch<unsigned short, unsigned short, LPR>(x, g, b, y, mean, rstd, rows, eps, perm_b, s);     \
  else if (out_bf16) ln_fwd_launch<float, unsigned short, LPR>(x, g, b, y, mean, rstd, rows, eps, perm_b, s);        \
  else ln_fwd_launch<float, float, LPR>(x, g, b, y, mean, rstd, rows, eps, perm_b, s)
  switch (D) {
    case 64: ATRI_LN_FWD(8); break;
    case 128: ATRI_LN_FWD(16); break;
    case 256: ATRI_LN_FWD(32); break;
    case 512: ATRI_LN_FWD(64); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef ATRI_LN_FWD
  return (int)hipGetLastError();
}

int atri_ln_fwd(const void* x, int is_bf16, const float* g, const float* b, void* y, int out_bf16, float* mean, float* rstd,
                int rows, int D, float eps, hipStream_t s) {
  return atri_ln_fwd_perm(x, is_bf16, g, b, y, out_bf16, mean, rstd, rows, D, eps, 0, s);
}

// dx, dgamma, dbeta of LayerNorm.  partial holds blocks * 2 * D floats with blocks = ceil(rows / rows_per_block);
// rows_per_block must be a multiple of 32.
// is_bf16: dtype of x and dx; dy_bf16: dtype of dy (that of the forward's output).
// skip (optional, dtype of x): a gradient that reaches x around the normalisation; dx = skip + LayerNorm'(dy).
int atri_ln_bwd_perm(const void* x, const void* dy, int is_bf16, int dy_bf16, const float* g, const float* mean, const float* rstd,
                     void* dx, float* partial, float* dgamma, float* dbeta, int rows, int D, int rows_per_block, int perm_b,
                     const void* skip, hipStream_t s) {
  if (rows <= 0) return 0;
  if (rows_per_block <= 0 || (rows_per_block & 31)) return (int)hipErrorInvalidValue;
  if (perm_b < 0 || (perm_b > 0 && rows % perm_b != 0)) return (int)hipErrorInvalidValue;
  if (is_bf16 && !dy_bf16) return (int)hipErrorInvalidValue;
  const int blocks = (rows + rows_per_block - 1) / rows_per_block;
#define ATRI_LN_BWD(LPR)                                                                                                               \
  if (is_bf16) ln_bwd_launch<unsigned short, unsigned short, LPR>(x, dy, g, mean, rstd, dx, partial, rows, rows_per_block, blocks, perm_b, skip, s); \
  else if (dy_bf16) ln_bwd_launch<float, unsigned short, LPR>(x, dy, g, mean, rstd, dx, partial, rows, rows_per_block, blocks, perm_b, skip, s);     \
  else ln_bwd_launch<float, float, LPR>(x, dy, g, mean, rstd, dx, partial, rows, rows_per_block, blocks, perm_b, skip, s)
  switch (D) {
    case 64: ATRI_LN_BWD(8); break;
    case 128: ATRI_LN_BWD(16); break;
    case 256: ATRI_LN_BWD(32); break;
    case 512: ATRI_LN_BWD(64); break;
    default: return (int)hipErrorInvalidValue;
  }
#undef ATRI_LN_BWD
  // partial is [blocks][2][D]: one column sum over blocks of the 2 * D wide rows gives dgamma | dbeta
  launch_colsum_final(partial, dgamma, blocks, 2 * D, s);
  (void)dbeta;  // dgamma and dbeta are the two halves of one [2 * D] buffer
  return (int)hipGetLastError();
}

int atri_ln_bwd(const void* x, const void* dy, int is_bf16, int dy_bf16, const float* g, const float* mean, const float* rstd,
                void* dx, float* partial, float* dgamma, float* dbeta, int rows, int D, int rows_per_block, hipStream_t s) {
  return atri_ln_bwd_perm(x, dy, is_bf16, dy_bf16, g, mean, rstd, dx, partial, dgamma, dbeta, rows, D, rows_per_block, 0, nullptr, s);
}

// cols [C*kh*kw, B*oh*ow] = patch matrix of x [C, B, H, W] (bf16 or f32, stride 1, zero padding ph / pw); 3x3 and 5x5 kernels.
int atri_patch_matrix(const void* x, void* cols, int is_bf16, int C, int B, int H, int W, int kh, int kw, int ph, int pw, hipStream_t s) {
  if (C <= 0 || B <= 0) return 0;
  if (ph < 0 || pw < 0 || ph >= kh || pw >= kw) return -1;
  if (is_bf16) {
    if (kh == 3 && kw == 3) return launch_patch_matrix<unsigned short, 3, 3>(x, cols, C, B, H, W, ph, pw, s);
    if (kh == 5 && kw == 5) return launch_patch_matrix<unsigned short, 5, 5>(x, cols, C, B, H, W, ph, pw, s);
  } else {
    if (kh == 3 && kw == 3) return launch_patch_matrix<float, 3, 3>(x, cols, C, B, H, W, ph, pw, s);
    if (kh == 5 && kw == 5) return launch_patch_matrix<float, 5, 5>(x, cols, C, B, H, W, ph, pw, s);
  }
  return -1;
}

// Node census of a captured (not yet instantiated) HIP graph: counts[0] = kernel nodes, counts[1] = memset nodes,
// counts[2] = every other node type.  Memset nodes recorded by library calls do not replay reliably, so callers
// refuse to replay a graph that holds any.
int atri_graph_census(void* graph, int* counts, hipStream_t) {
  hipGraph_t g = (hipGraph_t)graph;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(g, nullptr, &n);
  if (e != hipSuccess) return (int)e;
  std::vector<hipGraphNode_t> nodes(n);
  if (n) {
    e = hipGraphGetNodes(g, nodes.data(), &n);
    if (e != hipSuccess) return (int)e;
  }
  counts[0] = counts[1] = counts[2] = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    e = hipGraphNodeGetType(nodes[i], &t);
    if (e != hipSuccess) return (int)e;
    if (t == hipGraphNodeTypeKernel) ++counts[0];
    else if (t == hipGraphNodeTypeMemset) ++counts[1];
    else ++counts[2];
  }
  return 0;
}

}  // extern "C"
