// CTC loss (forward + gradient w.r.t. logits) and the per-language LID score reduction.
//   lid/LidModule_ASR_Supervised.py:162-168  CTCLoss(blank=V, reduction='none', zero_infinity=True)(log_softmax(logits).T, ...)
//   lid/ConformerLangModel.py:386-393        LangDiscriminator score of one language head
// One workgroup per utterance; the alpha lattice lives in a global workspace (L2-resident), beta is kept as two
// LDS rows; all arithmetic is f32 log-space, matching torch's ctc_loss for f32 inputs.
#include "common.h"

#define NEG_INF (-INFINITY)

__device__ __forceinline__ float lse2(float a, float b) {
  float m = fmaxf(a, b);
  if (m == NEG_INF) return NEG_INF;
  return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  float m = fmaxf(fmaxf(a, b), c);
  if (m == NEG_INF) return NEG_INF;
  return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

extern "C" long lidk_ctc_workspace_bytes(int B, int T_, int V1, int Lmax) {
  (void)V1;
  return (long)B * T_ * (2 * Lmax + 1) * (long)sizeof(float);
}

// dynamic LDS: lse[T] | beta0[Smax] | beta1[Smax] | ab[Smax] | lab[Smax] (int) | nxt[Smax] (int) | first[Smax] (int)
__global__ void __launch_bounds__(256)
ctc_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets, const int64_t* __restrict__ in_len,
           const int64_t* __restrict__ tg_len, float* __restrict__ loss, float* __restrict__ dlogits,
           float* __restrict__ alpha_ws, int T_, int V1, int Lmax, int blank, float gscale, int zero_inf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Smax = 2 * Lmax + 1;
  float* lse = reinterpret_cast<float*>(smem);
  float* beta0 = lse + T_;
  float* beta1 = beta0 + Smax;
  float* ab = beta1 + Smax;
  int* lab = reinterpret_cast<int*>(ab + Smax);
  int* nxt = lab + Smax;
  int* first = nxt + Smax;
  __shared__ float s_nll;

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* lg = logits + (size_t)b * T_ * V1;
  float* dl = dlogits ? dlogits + (size_t)b * T_ * V1 : nullptr;
  float* alpha = alpha_ws + (size_t)b * T_ * Smax;
  int Tb = (int)in_len[b]; if (Tb > T_) Tb = T_; if (Tb < 0) Tb = 0;
  int Lb = (int)tg_len[b]; if (Lb > Lmax) Lb = Lmax; if (Lb < 0) Lb = 0;
  const int S = 2 * Lb + 1;

  // row-wise log-sum-exp of the logits (one wave per row)
  for (int t = wave; t < T_; t += 4) {
    float mx = NEG_INF;
    for (int c = lane; c < V1; c += 64) mx = fmaxf(mx, lg[(size_t)t * V1 + c]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int c = lane; c < V1; c += 64) sm += __expf(lg[(size_t)t * V1 + c] - mx);
    sm = wave_sum(sm);
    if (lane == 0) lse[t] = mx + __logf(sm);
  }
  // extended label sequence l' and same-label chains
  for (int s = tid; s < S; s += 256) lab[s] = (s & 1) ? (int)targets[(size_t)b * Lmax + (s >> 1)] : blank;
  __syncthreads();
  for (int s = tid; s < S; s += 256) {
    int me = lab[s], n = -1, f = 1;
    for (int u = s + 1; u < S; ++u) if (lab[u] == me) { n = u; break; }
    for (int u = 0; u < s; ++u) if (lab[u] == me) { f = 0; break; }
    nxt[s] = n; first[s] = f;
  }
  __syncthreads();
#define LP(t, c) (lg[(size_t)(t) * V1 + (c)] - lse[t])

  if (Tb == 0) {      // torch: zero-length input -> loss 0 if the target is empty, else inf (-> 0 under zero_infinity)
    if (tid == 0) loss[b] = (Lb == 0) ? 0.f : (zero_inf ? 0.f : INFINITY);
    if (dl) for (int i = tid; i < T_ * V1; i += 256) dl[i] = 0.f;
    return;
  }
  // alpha
  for (int s = tid; s < S; s += 256) alpha[s] = (s < 2) ? LP(0, lab[s]) : NEG_INF;
  __syncthreads();
  for (int t = 1; t < Tb; ++t) {
    const float* ap = alpha + (size_t)(t - 1) * Smax;
    float* ac = alpha + (size_t)t * Smax;
    for (int s = tid; s < S; s += 256) {
      float a0 = ap[s];
      float a1 = s >= 1 ? ap[s - 1] : NEG_INF;
      float a2 = (s >= 2 && lab[s] != blank && lab[s] != lab[s - 2]) ? ap[s - 2] : NEG_INF;
      ac[s] = lse3(a0, a1, a2) + LP(t, lab[s]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float* al = alpha + (size_t)(Tb - 1) * Smax;
    float ll = lse2(al[S - 1], S >= 2 ? al[S - 2] : NEG_INF);
    s_nll = -ll;
  }
  __syncthreads();
  const float nll = s_nll;
  const bool inf_loss = (nll == INFINITY);
  if (tid == 0) loss[b] = (inf_loss && zero_inf) ? 0.f : nll;
  if (!dl) return;
  if (inf_loss || nll != nll) {            // zero_infinity: zero gradient as torch does
    for (int i = tid; i < T_ * V1; i += 256) dl[i] = 0.f;
    return;
  }
  // padded frames get zero gradient
  for (int i = tid + Tb * V1; i < T_ * V1; i += 256) dl[i] = 0.f;

  // beta + gradient, walking t backwards
  float* bc = beta0; float* bn = beta1;
  for (int t = Tb - 1; t >= 0; --t) {
    const float* at = alpha + (size_t)t * Smax;
    for (int s = tid; s < S; s += 256) {
      float v;
      if (t == Tb - 1) {
        v = (s >= S - 2) ? LP(t, lab[s]) : NEG_INF;
      } else {
        float b0 = bn[s];
        float b1 = s + 1 < S ? bn[s + 1] : NEG_INF;
        float b2 = (s + 2 < S && lab[s + 2] != blank && lab[s] != lab[s + 2]) ? bn[s + 2] : NEG_INF;
        v = lse3(b0, b1, b2) + LP(t, lab[s]);
      }
      bc[s] = v;
      ab[s] = at[s] + v;
    }
    for (int c = tid; c < V1; c += 256) dl[(size_t)t * V1 + c] = gscale * __expf(LP(t, c));
    __syncthreads();
    for (int s = tid; s < S; s += 256) {
      if (!first[s]) continue;
      float acc = ab[s];
      for (int u = nxt[s]; u >= 0; u = nxt[u]) acc = lse2(acc, ab[u]);
      int c = lab[s];
      float lp = LP(t, c);
      dl[(size_t)t * V1 + c] -= gscale * __expf(acc + nll - lp);
    }
    __syncthreads();
    float* tmp = bc; bc = bn; bn = tmp;
  }
#undef LP
}

extern "C" int lidk_ctc_loss(const float* logits, const int64_t* targets, const int64_t* in_len, const int64_t* tg_len,
                             float* loss, float* dlogits, void* workspace, int B, int T_, int V1, int Lmax, int blank,
                             float grad_scale, int zero_infinity, void* stream) {
  if (!logits || !targets || !in_len || !tg_len || !loss || !workspace || B <= 0 || T_ <= 0 || V1 <= 1 || Lmax <= 0 ||
      blank < 0 || blank >= V1)
    return LIDK_ERR_ARG;
  const int Smax = 2 * Lmax + 1;
  size_t lds = (size_t)(T_ + 3 * Smax) * sizeof(float) + (size_t)3 * Smax * sizeof(int);
  if (lds > 150 * 1024) return LIDK_ERR_UNSUPPORTED;
  (void)hipFuncSetAttribute((const void*)ctc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  ctc_kernel<<<B, 256, lds, as_stream(stream)>>>(logits, targets, in_len, tg_len, loss, dlogits, (float*)workspace, T_, V1,
                                                 Lmax, blank, grad_scale, zero_infinity);
  return launch_status();
}

// ------------------------------------------------------------------------------------ LID score
__global__ void __launch_bounds__(256)
lid_score_kernel(const float* __restrict__ logits, float* __restrict__ scores, int score_stride, int T_, int V1, int blank) {
  __shared__ float r_sum[4];
  __shared__ float r_cnt[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* lg = logits + (size_t)b * T_ * V1;
  float sum = 0.f, cnt = 0.f;
  for (int t = wave; t < T_; t += 4) {
    float mx = NEG_INF; int arg = 0x7fffffff;
    for (int c = lane; c < V1; c += 64) {
      float v = lg[(size_t)t * V1 + c];
      if (v > mx) { mx = v; arg = c; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      float om = __shfl_xor(mx, o, 64); int oa = __shfl_xor(arg, o, 64);
      if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    float sm = 0.f;
    for (int c = lane; c < V1; c += 64) sm += __expf(lg[(size_t)t * V1 + c] - mx);
    sm = wave_sum(sm);
    float maxlp = -__logf(sm);                       // max - lse = -log(sum exp(x - max))
    if (arg != blank) { sum += maxlp; cnt += 1.f; }
  }
  if (lane == 0) { r_sum[wave] = sum; r_cnt[wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = r_sum[0] + r_sum[1] + r_sum[2] + r_sum[3];
    float n = r_cnt[0] + r_cnt[1] + r_cnt[2] + r_cnt[3];
    scores[(size_t)b * score_stride] = s / (n * logf((float)blank) + 1e-5f);
  }
}

extern "C" int lidk_lid_score(const float* logits, float* scores, int score_stride, int B, int T_, int V1, int blank,
                              void* stream) {
  if (!logits || !scores || B <= 0 || T_ <= 0 || V1 <= 1 || blank != V1 - 1 || score_stride <= 0) return LIDK_ERR_ARG;
  lid_score_kernel<<<B, 256, 0, as_stream(stream)>>>(logits, scores, score_stride, T_, V1, blank);
  return launch_status();
}
