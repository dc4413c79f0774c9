// Fused gradient-norm clip + Novograd over flat f32 arenas, and the f32 -> T weight refresh.
//   ccml/trainer.py:541-543      torch.nn.utils.clip_grad_norm_(params, max_norm=20)
//   ccml/optim/novograd.py:75-145 per-tensor second moment of ||g||^2 (amsgrad=False, luc=False)
// Parameters, gradients and first moments are three parallel flat arenas, so the whole optimizer is three launches:
// (1) per-chunk sum of squares, (2) one-workgroup per-tensor bookkeeping (norms, clip coefficient, second moments),
// (3) the element-wise update.  Only tensors listed in `work` are touched, which is how "p.grad is None" (other
// language heads, layers skipped by stochastic depth) is expressed.
#include "common.h"

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
