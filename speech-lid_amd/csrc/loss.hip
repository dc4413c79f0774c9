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


// ------------------------------------------------------------------------------------ CTC fast path (3 launches)
// The O(T) recursions are the only sequential part of CTC, so they get a kernel of their own that touches nothing but LDS;
// everything that is parallel over frames runs chip-wide, one wave per (utterance, frame) row:
//   1. ctc_prep_kernel     row log-sum-exp + label log-probs lpl[b][t][s] = log_softmax(logits)[b][t][l'_s]
//   2. ctc_lattice_kernel  per utterance: wave 0 runs alpha, wave 1 runs beta (independent recursions, concurrently),
//                          wave 2 builds the same-label chains; rows are streamed to the workspace as they are produced
//   3. ctc_grad_kernel     dlogits row = gscale * (softmax - sum_{s: l'_s = c} posterior(s, t))
// workspace (4-byte words): lse[B*T] | nll[B] | lpl[B*T*Smax] | alpha[B*T*Smax] | beta[B*T*Smax] | nxt[B*Smax] | first[B*Smax]
struct CtcWs { float *lse, *nll, *lpl, *alpha, *beta; int *nxt, *first; };
static inline CtcWs ctc_ws(void* w, int B, int T_, int Smax) {
  CtcWs r; float* f = (float*)w; size_t lat = (size_t)B * T_ * Smax;
  r.lse = f; f += (size_t)B * T_; r.nll = f; f += B; r.lpl = f; f += lat; r.alpha = f; f += lat; r.beta = f; f += lat;
  r.nxt = (int*)f; r.first = r.nxt + (size_t)B * Smax; return r;
}
extern "C" long lidk_ctc_workspace_bytes(int B, int T_, int V1, int Lmax) {
  (void)V1; long S = 2 * Lmax + 1;
  return ((long)B * T_ * (3 * S + 1) + B + 2 * (long)B * S) * (long)sizeof(float);
}

__global__ void __launch_bounds__(256)
ctc_prep_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets, const int64_t* __restrict__ tg_len,
                float* __restrict__ lse, float* __restrict__ lpl, int rows, int T_, int V1, int Lmax, int blank) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / T_, Smax = 2 * Lmax + 1;
  const float* lg = logits + (size_t)row * V1;
  float mx = NEG_INF;
  for (int c = lane; c < V1; c += 64) mx = fmaxf(mx, lg[c]);
  mx = wave_max(mx);
  float sm = 0.f;
  for (int c = lane; c < V1; c += 64) sm += __expf(lg[c] - mx);
  sm = wave_sum(sm);
  const float l = mx + __logf(sm);
  if (lane == 0) lse[row] = l;
  int Lb = (int)tg_len[b]; if (Lb > Lmax) Lb = Lmax; if (Lb < 0) Lb = 0;
  const int S = 2 * Lb + 1;
  for (int s = lane; s < S; s += 64) {
    int c = (s & 1) ? (int)targets[(size_t)b * Lmax + (s >> 1)] : blank;
    lpl[(size_t)row * Smax + s] = lg[c] - l;
  }
}

// dynamic LDS: lp[T][Smax] | a0 a1 b0 b1 [Smax] | lab[Smax] (int)
__global__ void __launch_bounds__(192)
ctc_lattice_kernel(const int64_t* __restrict__ targets, const int64_t* __restrict__ in_len, const int64_t* __restrict__ tg_len,
                   float* __restrict__ loss, CtcWs ws, int T_, int Lmax, int blank, int zero_inf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Smax = 2 * Lmax + 1;
  float* lp = reinterpret_cast<float*>(smem);
  float* rowbuf = lp + (size_t)T_ * Smax;
  int* lab = reinterpret_cast<int*>(rowbuf + 4 * Smax);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int Tb = (int)in_len[b]; if (Tb > T_) Tb = T_; if (Tb < 0) Tb = 0;
  int Lb = (int)tg_len[b]; if (Lb > Lmax) Lb = Lmax; if (Lb < 0) Lb = 0;
  const int S = 2 * Lb + 1;
  const float* glp = ws.lpl + (size_t)b * T_ * Smax;
  for (int i = tid; i < Tb * Smax; i += 192) lp[i] = glp[i];
  for (int s = tid; s < S; s += 192) lab[s] = (s & 1) ? (int)targets[(size_t)b * Lmax + (s >> 1)] : blank;
  __syncthreads();
  if (Tb == 0) {
    if (tid == 0) { float v = (Lb == 0) ? 0.f : INFINITY; ws.nll[b] = v; loss[b] = (v == INFINITY && zero_inf) ? 0.f : v; }
    return;
  }
  if (S <= 64 && wave < 2) {
    // One lane per lattice state: the previous row lives in a register and the neighbours s -+ 1, s -+ 2 come by wave shuffles -
    // no LDS ping-pong and no barrier inside the only sequential loop of the loss (T steps of ~150 clocks instead of ~800).
    const int s = lane;
    const bool in = s < S;
    if (wave == 0) {                                 // alpha, forwards
      float* ga = ws.alpha + (size_t)b * T_ * Smax;
      const bool skip = in && s >= 2 && (s & 1) && lab[s] != lab[s - 2];
      float a = (in && s < 2) ? lp[s] : NEG_INF;
      if (in) ga[s] = a;
      float nxt_lp = (in && Tb > 1) ? lp[(size_t)Smax + s] : 0.f;
      for (int t = 1; t < Tb; ++t) {
        const float lpt = nxt_lp;
        if (in && t + 1 < Tb) nxt_lp = lp[(size_t)(t + 1) * Smax + s];
        float a1 = __shfl_up(a, 1, 64), a2 = __shfl_up(a, 2, 64);
        if (s < 1) a1 = NEG_INF;
        if (!skip) a2 = NEG_INF;
        const float v = lse3(a, a1, a2) + lpt;
        a = in ? v : NEG_INF;
        if (in) ga[(size_t)t * Smax + s] = a;
      }
      const float e1 = __shfl(a, S - 1, 64), e2 = __shfl(a, S >= 2 ? S - 2 : 0, 64);
      if (lane == 0) {
        float nll = -lse2(e1, S >= 2 ? e2 : NEG_INF);
        ws.nll[b] = nll;
        loss[b] = (nll == INFINITY && zero_inf) ? 0.f : nll;
      }
    } else {                                         // beta, backwards
      float* gb = ws.beta + (size_t)b * T_ * Smax;
      const bool skip = in && s + 2 < S && (s & 1) && lab[s] != lab[s + 2];
      float a = (in && s >= S - 2) ? lp[(size_t)(Tb - 1) * Smax + s] : NEG_INF;
      if (in) gb[(size_t)(Tb - 1) * Smax + s] = a;
      float nxt_lp = (in && Tb > 1) ? lp[(size_t)(Tb - 2) * Smax + s] : 0.f;
      for (int t = Tb - 2; t >= 0; --t) {
        const float lpt = nxt_lp;
        if (in && t > 0) nxt_lp = lp[(size_t)(t - 1) * Smax + s];
        float b1 = __shfl_down(a, 1, 64), b2 = __shfl_down(a, 2, 64);
        if (s + 1 >= S) b1 = NEG_INF;
        if (!skip) b2 = NEG_INF;
        const float v = lse3(a, b1, b2) + lpt;
        a = in ? v : NEG_INF;
        if (in) gb[(size_t)t * Smax + s] = a;
      }
    }
  } else if (wave == 0) {                            // alpha, forwards
    float* ga = ws.alpha + (size_t)b * T_ * Smax;
    float* cur = rowbuf; float* prv = rowbuf + Smax;
    for (int s = lane; s < S; s += 64) { float v = (s < 2) ? lp[s] : NEG_INF; cur[s] = v; ga[s] = v; }
    __builtin_amdgcn_wave_barrier();
    for (int t = 1; t < Tb; ++t) {
      float* tmp = cur; cur = prv; prv = tmp;
      for (int s = lane; s < S; s += 64) {
        float a0 = prv[s];
        float a1 = s >= 1 ? prv[s - 1] : NEG_INF;
        float a2 = (s >= 2 && (s & 1) && lab[s] != lab[s - 2]) ? prv[s - 2] : NEG_INF;
        float v = lse3(a0, a1, a2) + lp[(size_t)t * Smax + s];
        cur[s] = v; ga[(size_t)t * Smax + s] = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
      float nll = -lse2(cur[S - 1], S >= 2 ? cur[S - 2] : NEG_INF);
      ws.nll[b] = nll;
      loss[b] = (nll == INFINITY && zero_inf) ? 0.f : nll;
    }
  } else if (wave == 1) {                            // beta, backwards
    float* gb = ws.beta + (size_t)b * T_ * Smax;
    float* cur = rowbuf + 2 * Smax; float* nx = rowbuf + 3 * Smax;
    for (int s = lane; s < S; s += 64) {
      float v = (s >= S - 2) ? lp[(size_t)(Tb - 1) * Smax + s] : NEG_INF; cur[s] = v; gb[(size_t)(Tb - 1) * Smax + s] = v;
    }
    __builtin_amdgcn_wave_barrier();
    for (int t = Tb - 2; t >= 0; --t) {
      float* tmp = cur; cur = nx; nx = tmp;
      for (int s = lane; s < S; s += 64) {
        float b0 = nx[s];
        float b1 = s + 1 < S ? nx[s + 1] : NEG_INF;
        float b2 = (s + 2 < S && (s & 1) && lab[s] != lab[s + 2]) ? nx[s + 2] : NEG_INF;
        float v = lse3(b0, b1, b2) + lp[(size_t)t * Smax + s];
        cur[s] = v; gb[(size_t)t * Smax + s] = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else {                                           // same-label chains (first occurrence + next occurrence)
    for (int s = lane; s < S; s += 64) {
      int me = lab[s], n = -1, f = 1;
      for (int u = s + 1; u < S; ++u) if (lab[u] == me) { n = u; break; }
      for (int u = 0; u < s; ++u) if (lab[u] == me) { f = 0; break; }
      ws.nxt[(size_t)b * Smax + s] = n; ws.first[(size_t)b * Smax + s] = f;
    }
  }
}

// dynamic LDS per wave: post[Smax] | corr[V1].  dlogits [rows][ld] of T (columns V1 .. ld-1, the GEMM operand's padding, are
// written as zero); gscale_dev (optional): a device scalar multiplied into gscale - the upstream gradient of the mean loss.
template <typename T>
__global__ void __launch_bounds__(256)
ctc_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets, const int64_t* __restrict__ in_len,
                const int64_t* __restrict__ tg_len, T* __restrict__ dlogits, int ld, CtcWs ws, int rows, int T_, int V1, int Lmax,
                int blank, float gscale, const float* __restrict__ gscale_dev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Smax = 2 * Lmax + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  float* post = reinterpret_cast<float*>(smem) + (size_t)wave * (Smax + V1);
  float* corr = post + Smax;
  const int b = row / T_, t = row - b * T_;
  T* dl = dlogits + (size_t)row * ld;
  if (gscale_dev) gscale *= *gscale_dev;
  int Tb = (int)in_len[b]; if (Tb > T_) Tb = T_; if (Tb < 0) Tb = 0;
  const float nll = ws.nll[b];
  if (t >= Tb || nll == INFINITY || nll != nll) {    // padded frame, or zero_infinity: zero gradient as torch does
    for (int c = lane; c < ld; c += 64) dl[c] = from_f<T>(0.f);
    return;
  }
  int Lb = (int)tg_len[b]; if (Lb > Lmax) Lb = Lmax; if (Lb < 0) Lb = 0;
  const int S = 2 * Lb + 1;
  const size_t lo = (size_t)row * Smax;
  for (int s = lane; s < S; s += 64) post[s] = ws.alpha[lo + s] + ws.beta[lo + s] + nll - ws.lpl[lo + s];   // log posterior
  for (int c = lane; c < V1; c += 64) corr[c] = 0.f;
  __builtin_amdgcn_wave_barrier();
  for (int s = lane; s < S; s += 64) {
    if (!ws.first[(size_t)b * Smax + s]) continue;
    float acc = post[s];
    for (int u = ws.nxt[(size_t)b * Smax + s]; u >= 0; u = ws.nxt[(size_t)b * Smax + u]) acc = lse2(acc, post[u]);
    int c = (s & 1) ? (int)targets[(size_t)b * Lmax + (s >> 1)] : blank;
    corr[c] = __expf(acc);
  }
  __builtin_amdgcn_wave_barrier();
  const float* lg = logits + (size_t)row * V1;
  const float l = ws.lse[row];
  for (int c = lane; c < ld; c += 64) dl[c] = from_f<T>(c < V1 ? gscale * (__expf(lg[c] - l) - corr[c]) : 0.f);
}

// in_len[b] = (long)(T * wav_pct[b]), tg_len[b] = (long)(Lmax * txt_pct[b]) in f32, truncated toward zero - the module's
// ``(out.shape[1] * wav_percents).long()`` / ``(texts.shape[-1] * text_percents).long()`` (lid/LidModule_ASR_Supervised.py:163-166)
__global__ void ctc_lengths_kernel(const float* __restrict__ wav_pct, const float* __restrict__ txt_pct, int64_t* __restrict__ in_len,
                                   int64_t* __restrict__ tg_len, int B, int T_, int Lmax) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  in_len[b] = (int64_t)((float)T_ * wav_pct[b]);
  tg_len[b] = (int64_t)((float)Lmax * txt_pct[b]);
}
// mean of the per-utterance losses (reduction='none' then .mean(), :167-168), one wave
__global__ void __launch_bounds__(64) ctc_mean_kernel(const float* __restrict__ loss, float* __restrict__ mean, int B) {
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 64) s += loss[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) *mean = s / (float)B;
}

extern "C" int lidk_ctc_loss(const float* logits, const int64_t* targets, const int64_t* in_len, const int64_t* tg_len,
                             float* loss, float* dlogits, void* workspace, int B, int T_, int V1, int Lmax, int blank,
                             float grad_scale, int zero_infinity, void* stream) {
  if (!logits || !targets || !in_len || !tg_len || !loss || !workspace || B <= 0 || T_ <= 0 || V1 <= 1 || Lmax <= 0 ||
      blank < 0 || blank >= V1)
    return LIDK_ERR_ARG;
  const int Smax = 2 * Lmax + 1;
  const size_t fast = ((size_t)T_ * Smax + 4 * Smax) * sizeof(float) + (size_t)Smax * sizeof(int);
  const size_t glds = (size_t)4 * (Smax + V1) * sizeof(float);
  if (fast <= 150 * 1024 && glds <= 150 * 1024) {
    hipStream_t st = as_stream(stream);
    CtcWs ws = ctc_ws(workspace, B, T_, Smax);
    const int rows = B * T_;
    ctc_prep_kernel<<<(rows + 3) / 4, 256, 0, st>>>(logits, targets, tg_len, ws.lse, ws.lpl, rows, T_, V1, Lmax, blank);
    (void)hipFuncSetAttribute((const void*)ctc_lattice_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast);
    ctc_lattice_kernel<<<B, 192, fast, st>>>(targets, in_len, tg_len, loss, ws, T_, Lmax, blank, zero_infinity);
    if (dlogits) {
      (void)hipFuncSetAttribute((const void*)ctc_grad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds);
      ctc_grad_kernel<float><<<(rows + 3) / 4, 256, glds, st>>>(logits, targets, in_len, tg_len, dlogits, V1, ws, rows, T_, V1, Lmax,
                                                                blank, grad_scale, nullptr);
    }
    return launch_status();
  }
  size_t lds = (size_t)(T_ + 3 * Smax) * sizeof(float) + (size_t)3 * Smax * sizeof(int);
  if (lds > 150 * 1024) return LIDK_ERR_UNSUPPORTED;
  (void)hipFuncSetAttribute((const void*)ctc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  ctc_kernel<<<B, 256, lds, as_stream(stream)>>>(logits, targets, in_len, tg_len, loss, dlogits, (float*)workspace, T_, V1,
                                                 Lmax, blank, grad_scale, zero_infinity);
  return launch_status();
}

// The training step's form of the same loss (lid/LidModule_ASR_Supervised.py:162-168 + ccml/trainer.py:521,531): lengths from the
// batch's percents, per-utterance losses AND their mean in one call; the gradient is produced later, at backward time, straight
// into the T-typed, column-padded operand of the vocabulary projection's data / weight gradient GEMMs, scaled by
// grad_scale * (*grad_scale_dev) - the upstream gradient autograd hands the mean (1 / accumulate_grad).  Fast path only
// (LIDK_ERR_UNSUPPORTED when the lattice does not fit LDS: callers fall back to lidk_ctc_loss).
extern "C" int lidk_ctc_forward(const float* logits, const int64_t* targets, const float* wav_pct, const float* txt_pct,
                                int64_t* in_len, int64_t* tg_len, float* loss, float* loss_mean, void* workspace, int B, int T_, int V1,
                                int Lmax, int blank, int zero_infinity, void* stream) {
  if (!logits || !targets || !wav_pct || !txt_pct || !in_len || !tg_len || !loss || !loss_mean || !workspace || B <= 0 || T_ <= 0 ||
      V1 <= 1 || Lmax <= 0 || blank < 0 || blank >= V1)
    return LIDK_ERR_ARG;
  const int Smax = 2 * Lmax + 1;
  const size_t fast = ((size_t)T_ * Smax + 4 * Smax) * sizeof(float) + (size_t)Smax * sizeof(int);
  if (fast > 150 * 1024 || (size_t)4 * (Smax + V1) * sizeof(float) > 150 * 1024) return LIDK_ERR_UNSUPPORTED;
  hipStream_t st = as_stream(stream);
  CtcWs ws = ctc_ws(workspace, B, T_, Smax);
  const int rows = B * T_;
  ctc_lengths_kernel<<<cdiv(B, 64), 64, 0, st>>>(wav_pct, txt_pct, in_len, tg_len, B, T_, Lmax);
  ctc_prep_kernel<<<(rows + 3) / 4, 256, 0, st>>>(logits, targets, tg_len, ws.lse, ws.lpl, rows, T_, V1, Lmax, blank);
  (void)hipFuncSetAttribute((const void*)ctc_lattice_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast);
  ctc_lattice_kernel<<<B, 192, fast, st>>>(targets, in_len, tg_len, loss, ws, T_, Lmax, blank, zero_infinity);
  ctc_mean_kernel<<<1, 64, 0, st>>>(loss, loss_mean, B);
  return launch_status();
}
extern "C" int lidk_ctc_backward(const float* logits, const int64_t* targets, const int64_t* in_len, const int64_t* tg_len,
                                 void* dlogits, int ld, int dtype, void* workspace, int B, int T_, int V1, int Lmax, int blank,
                                 float grad_scale, const float* grad_scale_dev, void* stream) {
  if (!logits || !targets || !in_len || !tg_len || !dlogits || !workspace || B <= 0 || T_ <= 0 || V1 <= 1 || ld < V1 || Lmax <= 0)
    return LIDK_ERR_ARG;
  const int Smax = 2 * Lmax + 1;
  const size_t glds = (size_t)4 * (Smax + V1) * sizeof(float);
  if (glds > 150 * 1024) return LIDK_ERR_UNSUPPORTED;
  CtcWs ws = ctc_ws(workspace, B, T_, Smax);
  const int rows = B * T_;
  LIDK_DISPATCH(dtype, {
    (void)hipFuncSetAttribute((const void*)ctc_grad_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds);
    ctc_grad_kernel<T><<<(rows + 3) / 4, 256, glds, as_stream(stream)>>>(logits, targets, in_len, tg_len, (T*)dlogits, ld, ws, rows, T_,
                                                                        V1, Lmax, blank, grad_scale, grad_scale_dev);
  });
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
    float mx = NEG_INF; int arg = blank;             // an all-NaN / all -inf row decodes as blank, never as an out-of-range id
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

// ------------------------------------------------------------------------------------ LangDiscriminator MLP
// One wave per utterance: h = relu(W0 s + b0) (H <= 64 hidden units, one per lane), out_c = W2[c,:] . h + b2[c] by a wave
// reduction per class.  Rows are independent of the batch and of each other, the summation order is fixed.
__global__ void __launch_bounds__(64)
lid_mlp_kernel(const float* __restrict__ scores, const float* __restrict__ w0, const float* __restrict__ b0,
               const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ out, int C, int H) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float h = 0.f;
  if (lane < H) {
    float acc = b0[lane];
    for (int c = 0; c < C; ++c) acc = fmaf(w0[(size_t)lane * C + c], scores[(size_t)b * C + c], acc);
    h = fmaxf(acc, 0.f);
  }
  for (int c = 0; c < C; ++c) {
    float v = lane < H ? w2[(size_t)c * H + lane] * h : 0.f;
    v = wave_sum(v);
    if (lane == 0) out[(size_t)b * C + c] = v + b2[c];
  }
}

extern "C" int lidk_lid_mlp(const float* scores, const float* w0, const float* b0, const float* w2, const float* b2,
                            float* out, int B, int C, int H, void* stream) {
  if (!scores || !w0 || !b0 || !w2 || !b2 || !out || B <= 0 || C <= 0) return LIDK_ERR_ARG;
  if (H <= 0 || H > 64) return LIDK_ERR_UNSUPPORTED;
  lid_mlp_kernel<<<B, 64, 0, as_stream(stream)>>>(scores, w0, b0, w2, b2, out, C, H);
  return launch_status();
}

// ------------------------------------------------------------------------------------ greedy CTC decode
// CTCTokenizer.ctc_decode (lid/tokenizer.py:55-70) on the device: per frame argmax over the vocabulary (ties -> lowest index,
// as torch.argmax), then keep a frame's symbol iff it is not blank and differs from the previous FRAME's symbol.  One
// workgroup per utterance: waves take frames round-robin for the argmax, a wave-level prefix sum packs the kept symbols.
__global__ void __launch_bounds__(256)
ctc_greedy_kernel(const float* __restrict__ logits, const int64_t* __restrict__ in_len, int* __restrict__ ids,
                  int* __restrict__ out_len, int T_, int V1, int blank) {
  extern __shared__ int s_arg[];                     // [T_]
  __shared__ int s_base;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = in_len ? (int)min((int64_t)T_, max((int64_t)0, in_len[b])) : T_;
  const float* lg = logits + (size_t)b * T_ * V1;
  for (int t = wave; t < n; t += 4) {
    float mx = NEG_INF; int arg = blank;             // an all-NaN / all -inf row decodes as blank, never as an out-of-range id
    for (int c = lane; c < V1; c += 64) {
      float v = lg[(size_t)t * V1 + c];
      if (v > mx) { mx = v; arg = c; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      float om = __shfl_xor(mx, o, 64); int oa = __shfl_xor(arg, o, 64);
      if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    if (lane == 0) s_arg[t] = arg;
  }
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  if (wave == 0) {                                    // one wave packs: 64 frames per round, ballot prefix
    for (int t0 = 0; t0 < n; t0 += 64) {
      const int t = t0 + lane;
      int sym = blank, prev = blank;
      if (t < n) { sym = s_arg[t]; prev = t > 0 ? s_arg[t - 1] : blank; }
      const bool keep = t < n && sym != blank && sym != prev;
      const unsigned long long m = __ballot(keep);
      const int pos = s_base + __popcll(m & ((1ull << lane) - 1ull));
      if (keep) ids[(size_t)b * T_ + pos] = sym;
      if (lane == 0) s_base += __popcll(m);          // same wave: program order, no barrier needed
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) out_len[b] = s_base;
  }
}

extern "C" int lidk_ctc_greedy(const float* logits, const int64_t* in_len, int* ids, int* out_len, int B, int T_, int V1,
                               int blank, void* stream) {
  if (!logits || !ids || !out_len || B <= 0 || T_ <= 0 || V1 <= 1 || blank < 0 || blank >= V1) return LIDK_ERR_ARG;
  if ((size_t)T_ * 4 > 150 * 1024) return LIDK_ERR_UNSUPPORTED;      // frame symbols live in LDS
  if ((size_t)T_ * 4 > 48 * 1024)                                     // beyond the default dynamic-LDS limit: ask for it
    (void)hipFuncSetAttribute((const void*)ctc_greedy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)T_ * 4));
  ctc_greedy_kernel<<<B, 256, (size_t)T_ * 4, as_stream(stream)>>>(logits, in_len, ids, out_len, T_, V1, blank);
  return launch_status();
}
