// Fused gradient-norm clip + Novograd over flat f32 arenas, and the f32 -> T weight refresh.
//   ccml/trainer.py:541-543      torch.nn.utils.clip_grad_norm_(params, max_norm=20)
//   ccml/optim/novograd.py:75-145 per-tensor second moment of ||g||^2 (amsgrad=False, luc=False)
// Parameters, gradients and first moments are three parallel flat arenas, so the whole optimizer is three launches:
// (1) per-chunk sum of squares, (2) one-workgroup per-tensor bookkeeping (norms, clip coefficient, second moments),
// (3) the element-wise update.  Only tensors listed in `work` are touched, which is how "p.grad is None" (other
// language heads, layers skipped by stochastic depth) is expressed.
#include "common.h"
#include <math.h>

__global__ void __launch_bounds__(256)
sumsq_chunks_kernel(const float* __restrict__ grads, const int64_t* __restrict__ work, float* __restrict__ chunk_sumsq) {
  __shared__ float red[4];
  const int64_t off = work[(size_t)blockIdx.x * 3 + 1], len = work[(size_t)blockIdx.x * 3 + 2];
  const float* g = grads + off;
  float s = 0.f;
  // 16-byte loads (a chunk starts on a 32-byte boundary: tensor offsets are 8-float aligned, chunks are LIDK_OPT_CHUNK long);
  // the last chunk of a tensor may end off a multiple of 4
  const int64_t n4 = len >> 2;
  for (int64_t i = threadIdx.x; i < n4; i += 256) {
    const float4 v = load4(g + 4 * i);
    s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
  }
  for (int64_t i = 4 * n4 + threadIdx.x; i < len; i += 256) { float v = g[i]; s = fmaf(v, v, s); }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) chunk_sumsq[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

#define NOVO_MAX_TENSORS 4096
// scratch layout: [0, n_work) chunk sums | [n_work, n_work+n_tensors) per-tensor scale | [n_work+n_tensors] total sumsq
__global__ void __launch_bounds__(256)
novograd_prepare_kernel(const int64_t* __restrict__ work, int n_work, int n_tensors, float* __restrict__ scratch,
                        float* __restrict__ exp_avg_sq, float beta2, float eps, float max_norm,
                        float* __restrict__ total_norm_out) {
  __shared__ float s_total;
  __shared__ float red[4];
  float* chunk = scratch;
  float* scale = scratch + n_work;
  // pass 1: per-tensor sums.  Run boundaries of every tensor's contiguous chunk run are found in parallel first, so the
  // summation loop has known bounds and its loads are independent (a scan that discovers the end of a run while it sums
  // is a chain of dependent global loads: 78 us for cfg2).  Fixed order -> deterministic.
  __shared__ int run_lo[NOVO_MAX_TENSORS], run_hi[NOVO_MAX_TENSORS];
  for (int t = threadIdx.x; t < n_tensors; t += 256) { scale[t] = -1.f; if (t < NOVO_MAX_TENSORS) run_lo[t] = -1; }   // -1: "not in this step"
  __syncthreads();
  const bool fast = n_tensors <= NOVO_MAX_TENSORS;
  float mine = 0.f;
  if (fast) {
    for (int w = threadIdx.x; w < n_work; w += 256) {
      const int tid_ = (int)work[(size_t)w * 3];
      if (w == 0 || work[(size_t)(w - 1) * 3] != tid_) run_lo[tid_] = w;
      if (w == n_work - 1 || work[(size_t)(w + 1) * 3] != tid_) run_hi[tid_] = w + 1;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_tensors; t += 256) {
      if (run_lo[t] < 0) continue;
      float acc = 0.f;
      for (int u = run_lo[t]; u < run_hi[t]; ++u) acc += chunk[u];
      scale[t] = acc;                                                   // holds ||g_t||^2 for now
      mine += acc;
    }
  } else {
    for (int w = threadIdx.x; w < n_work; w += 256) {
      int64_t tid_ = work[(size_t)w * 3];
      if (w == 0 || work[(size_t)(w - 1) * 3] != tid_) {               // first chunk of a tensor
        float acc = 0.f;
        for (int u = w; u < n_work && work[(size_t)u * 3] == tid_; ++u) acc += chunk[u];
        scale[tid_] = acc;
        mine += acc;
      }
    }
  }
  mine = wave_sum(mine);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = red[0] + red[1] + red[2] + red[3];
    s_total = tot;
    if (total_norm_out) *total_norm_out = sqrtf(tot);
  }
  __syncthreads();
  const float total = sqrtf(s_total);
  const float clip = (max_norm > 0.f) ? fminf(1.f, max_norm / (total + 1e-6f)) : 1.f;
  for (int t = threadIdx.x; t < n_tensors; t += 256) {
    float n = scale[t];
    if (n < 0.f) continue;
    n *= clip * clip;                                                   // norm of the clipped gradient
    float v = exp_avg_sq[t];
    v = (v == 0.f) ? n : beta2 * v + (1.f - beta2) * n;                 // novograd.py:115-118
    exp_avg_sq[t] = v;
    scale[t] = clip / (sqrtf(v) + eps);
  }
}

__global__ void __launch_bounds__(256)
novograd_apply_kernel(float* __restrict__ params, float* __restrict__ grads, float* __restrict__ exp_avg,
                      const int64_t* __restrict__ work, const float* __restrict__ scale, float lr, float beta1,
                      float weight_decay, float ga) {
  const int64_t t = work[(size_t)blockIdx.x * 3], off = work[(size_t)blockIdx.x * 3 + 1], len = work[(size_t)blockIdx.x * 3 + 2];
  const float sc = scale[t];
  const int64_t n4 = len >> 2;
  for (int64_t i = threadIdx.x; i < n4; i += 256) {                     // 16-byte accesses on all three arenas
    const int64_t e = off + 4 * i;
    const float4 p = load4(params + e), gr = load4(grads + e), mo = load4(exp_avg + e);
    float4 m, q;
    m.x = beta1 * mo.x + (gr.x * sc + weight_decay * p.x) * ga; m.y = beta1 * mo.y + (gr.y * sc + weight_decay * p.y) * ga;
    m.z = beta1 * mo.z + (gr.z * sc + weight_decay * p.z) * ga; m.w = beta1 * mo.w + (gr.w * sc + weight_decay * p.w) * ga;
    q.x = p.x - lr * m.x; q.y = p.y - lr * m.y; q.z = p.z - lr * m.z; q.w = p.w - lr * m.w;
    store4(exp_avg + e, m);
    store4(params + e, q);
    store4(grads + e, make_float4(0.f, 0.f, 0.f, 0.f));
  }
  for (int64_t i = 4 * n4 + threadIdx.x; i < len; i += 256) {
    float p = params[off + i];
    float g = grads[off + i] * sc + weight_decay * p;
    g *= ga;
    float m = beta1 * exp_avg[off + i] + g;
    exp_avg[off + i] = m;
    params[off + i] = p - lr * m;
    grads[off + i] = 0.f;                      // consumed: the next step accumulates into zeros (no separate 186 MB memset)
  }
}

extern "C" int lidk_novograd_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* work,
                                  int n_work, int n_tensors, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, int grad_averaging, float max_norm, float* scratch,
                                  float* total_norm_out, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !work || !scratch || n_work <= 0 || n_tensors <= 0) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  sumsq_chunks_kernel<<<n_work, 256, 0, s>>>(grads, work, scratch);
  novograd_prepare_kernel<<<1, 256, 0, s>>>(work, n_work, n_tensors, scratch, exp_avg_sq, beta2, eps, max_norm, total_norm_out);
  novograd_apply_kernel<<<n_work, 256, 0, s>>>(params, grads, exp_avg, work, scratch + n_work, lr, beta1, weight_decay,
                                               grad_averaging ? (1.f - beta1) : 1.f);
  return launch_status();
}

// ------------------------------------------------------------------------------------ weight refresh (f32 master -> T operands)
// ONE launch for all matrices: mats [n][8] int64 (device) = (src offset, rows R, cols C, dst offset of W or -1, dst offset of
// W^T or -1, leading dim of W^T, first tile index, tiles per row).  A workgroup handles one 32x32 tile; it finds its matrix
// by binary search over the tile prefix.
template <typename T>
__global__ void __launch_bounds__(256)
cast_weights_kernel(const float* __restrict__ params, T* __restrict__ wT, const int64_t* __restrict__ mats, int n_mats) {
  __shared__ float tile[32][33];
  const long tid = blockIdx.x;
  int lo = 0, hi = n_mats - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (mats[(size_t)mid * 8 + 6] <= tid) lo = mid; else hi = mid - 1;
  }
  const int64_t* m = mats + (size_t)lo * 8;
  const int R = (int)m[1], C = (int)m[2], ldt = (int)m[5], tx_n = (int)m[7];
  const long local = tid - m[6];
  const int r0 = (int)(local / tx_n) * 32, c0 = (int)(local % tx_n) * 32;
  const float* src = params + m[0];
  T* w = m[3] >= 0 ? wT + m[3] : nullptr;
  T* wt = m[4] >= 0 ? wT + m[4] : nullptr;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    int r = r0 + i, c = c0 + tx;
    if (r < R && c < C) {
      float v = src[(size_t)r * C + c];
      tile[i][tx] = v;
      if (w) w[(size_t)r * C + c] = from_f<T>(v);
    }
  }
  __syncthreads();
  if (!wt) return;
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, r = r0 + tx;
    if (r < R && c < C) wt[(size_t)c * ldt + r] = from_f<T>(tile[tx][i]);
  }
}

extern "C" int lidk_cast_weights(const float* params, void* wT, const int64_t* mats, int n_mats, long total_tiles, int dtype,
                                 void* stream) {
  if (!params || !wT || !mats || n_mats <= 0 || total_tiles <= 0) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  LIDK_DISPATCH(dtype, cast_weights_kernel<T><<<(unsigned)total_tiles, 256, 0, s>>>(params, (T*)wT, mats, n_mats));
  return launch_status();
}

extern "C" int lidk_version(void) { return 1; }

// ------------------------------------------------------------------------------------ multi-tensor Adam / SGD
// torch.optim.Adam / SGD over a LIST of parameter tensors (the reference's optimizers for the wav2vec2 / WavLM confs:
// lid/LidModule_ASR.py:143-150 with lid/conf/xf_asr_wav2vec.yaml:25, xf_asr_extra_finetune.yaml:22) in ONE launch: a device table of
// chunks {p, g, m, v, n <= LIDK_MT_CHUNK}, one workgroup per chunk.  torch's default (foreach) implementation runs ~10 passes of ~25
// launches each over the 300 M parameters of XLS-R: 11 of a 98 ms fine-tune step; this is one read-modify-write pass.
// Same update rule as torch (single-tensor form, torch/optim/adam.py _single_tensor_adam; sgd.py _single_tensor_sgd): weight decay
// added to the gradient, exp_avg by lerp, denom = sqrt(v) / sqrt(1 - beta2^t) + eps, step lr / (1 - beta1^t).
struct MtChunk { float* p; const float* g; float* m; float* v; int n; int pad; };
extern "C" int lidk_mt_chunk_bytes(void) { return (int)sizeof(MtChunk); }
extern "C" int lidk_mt_chunk_elems(void) { return LIDK_MT_CHUNK; }

struct AdamArgs { float lr_bc1, beta1, beta2, eps, wd, rsqrt_bc2; int maximize; };

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
  if (a.maximize) g = -g;
  if (a.wd != 0.f) g = fmaf(a.wd, p, g);
  m = fmaf(1.f - a.beta1, g - m, m);
  v = fmaf(1.f - a.beta2, g * g, a.beta2 * v);
  const float denom = fmaf(sqrtf(v), a.rsqrt_bc2, a.eps);
  p = p - a.lr_bc1 * (m / denom);
}

__global__ void __launch_bounds__(256) adam_multi_kernel(const MtChunk* __restrict__ chunks, AdamArgs a) {
  const MtChunk c = chunks[blockIdx.x];
  const bool vec = !(((size_t)c.p | (size_t)c.g | (size_t)c.m | (size_t)c.v) & 15);
  const int n4 = vec ? c.n >> 2 : 0;
  for (int i = threadIdx.x; i < n4; i += 256) {
    float4 p = load4(c.p + 4 * i), m = load4(c.m + 4 * i), v = load4(c.v + 4 * i);
    const float4 g = load4(c.g + 4 * i);
    adam_one(p.x, g.x, m.x, v.x, a); adam_one(p.y, g.y, m.y, v.y, a); adam_one(p.z, g.z, m.z, v.z, a); adam_one(p.w, g.w, m.w, v.w, a);
    store4(c.p + 4 * i, p); store4(c.m + 4 * i, m); store4(c.v + 4 * i, v);
  }
  for (int i = 4 * n4 + threadIdx.x; i < c.n; i += 256) {
    float p = c.p[i], m = c.m[i], v = c.v[i];
    adam_one(p, c.g[i], m, v, a);
    c.p[i] = p; c.m[i] = m; c.v[i] = v;
  }
}

extern "C" int lidk_adam_multi(const void* chunks, int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                               double bias_correction1, double bias_correction2, int maximize, void* stream) {
  if (!chunks || n_chunks <= 0 || bias_correction1 <= 0.0 || bias_correction2 <= 0.0) return LIDK_ERR_ARG;
  AdamArgs a{(float)(lr / bias_correction1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bias_correction2)), maximize};
  adam_multi_kernel<<<n_chunks, 256, 0, as_stream(stream)>>>((const MtChunk*)chunks, a);
  return launch_status();
}

struct SgdArgs { float lr, momentum, dampening, wd; int nesterov, first, maximize; };

__global__ void __launch_bounds__(256) sgd_multi_kernel(const MtChunk* __restrict__ chunks, SgdArgs a) {
  const MtChunk c = chunks[blockIdx.x];                       // m = momentum buffer (NULL without momentum), v unused
  for (int i = threadIdx.x; i < c.n; i += 256) {
    float p = c.p[i], g = c.g[i];
    if (a.maximize) g = -g;
    if (a.wd != 0.f) g = fmaf(a.wd, p, g);
    if (c.m) {
      const float b = a.first ? g : fmaf(a.momentum, c.m[i], (1.f - a.dampening) * g);
      c.m[i] = b;
      g = a.nesterov ? fmaf(a.momentum, b, g) : b;
    }
    c.p[i] = p - a.lr * g;
  }
}

extern "C" int lidk_sgd_multi(const void* chunks, int n_chunks, float lr, float momentum, float dampening, float weight_decay,
                              int nesterov, int first_step, int maximize, void* stream) {
  if (!chunks || n_chunks <= 0) return LIDK_ERR_ARG;
  SgdArgs a{lr, momentum, dampening, weight_decay, nesterov, first_step, maximize};
  sgd_multi_kernel<<<n_chunks, 256, 0, as_stream(stream)>>>((const MtChunk*)chunks, a);
  return launch_status();
}
