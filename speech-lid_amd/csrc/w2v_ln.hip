// The "layer_norm" form of the convolutional feature extractor and the waveform normalisation of wav2vec2 Large / XLS-R and
// WavLM Large (the checkpoints every wav2vec conf of the reference loads: lid/conf/xf_asr_wav2vec.yaml:12 xlsr2_300m.pt,
// lid/conf/xf_asr_extra_finetune.yaml:12 WavLM-Large.pt):
//   * ConvFeatureExtractionModel(mode="layer_norm", conv_bias) - lid/wavlm/WavLM.py:415-477 (the same class as
//     lid/s3prl_updream/wav2vec/wav2vec2.py:769-848): EVERY layer is Conv1d(+bias) -> LayerNorm over the channels of a time step
//     (Fp32LayerNorm, affine) -> GELU.  Activations are channel-last here, so that LayerNorm is a plain row LayerNorm of C = 512
//     values: one wave per row, 8 channels per lane, 16-byte accesses, statistics by wave shuffles.  Layers 1-6 stay GEMMs over
//     strided views (lidk_gemm_nt with the bias in its epilogue) followed by lidk_ln_gelu_fwd; layer 0 (k10 s5 on the raw
//     waveform, K = 10 is too thin for MFMA) computes its 512 channels, their LayerNorm and the GELU in registers.
//   * task.normalize (lid/s3prl_updream/wav2vec/wav2vec2_expert.py:71-72): F.layer_norm(wav, wav.shape) per utterance - biased
//     variance, eps inside the root - on the utterance's own samples; the zero padding behind them stays zero.
#include "common.h"

#define LNX_C 512          // channels of the extractor (every released wav2vec2 / WavLM model): 64 lanes x 8
#define LNX_K 10
#define LNX_S 5
#define LNX_TC 64          // frames per workgroup of the layer-0 kernels (4 waves x 16)

// ------------------------------------------------------------------------------------ waveform layer-norm
__global__ void __launch_bounds__(1024) wav_layernorm_kernel(const float* __restrict__ wav, float* __restrict__ out, int Lrow,
                                                             const int32_t* __restrict__ n_samples, float eps) {
  __shared__ float red[16];
  __shared__ float bc;
  const float* x = wav + (size_t)blockIdx.x * Lrow;
  float* y = out + (size_t)blockIdx.x * Lrow;
  const int L = n_samples ? max(1, min(Lrow, n_samples[blockIdx.x])) : Lrow;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) s += x[i];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < 16; ++w) t += red[w]; bc = t / (float)L; }
  __syncthreads();
  const float mu = bc;
  float q = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) { float d = x[i] - mu; q = fmaf(d, d, q); }
  q = wave_sum(q);
  __syncthreads();
  if (lane == 0) red[wave] = q;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < 16; ++w) t += red[w]; bc = rsqrtf(t / (float)L + eps); }
  __syncthreads();
  const float inv = bc;
  for (int i = threadIdx.x; i < L; i += blockDim.x) y[i] = (x[i] - mu) * inv;
  for (int i = L + threadIdx.x; i < Lrow; i += blockDim.x) y[i] = 0.f;
}
extern "C" int lidk_wav_layernorm(const float* wav, float* out, int B, int L, const int32_t* n_samples, float eps, void* stream) {
  if (!wav || !out || B <= 0 || L < 1) return LIDK_ERR_ARG;
  wav_layernorm_kernel<<<B, 1024, 0, as_stream(stream)>>>(wav, out, L, n_samples, eps);
  return launch_status();
}

// ------------------------------------------------------------------------------------ helpers: one row of 512 values per wave
struct row8 { float v[8]; };
__device__ __forceinline__ row8 load_row8(const bf16* p) {
  f32x8_t t = load8(p);
  row8 r; r.v[0] = t.lo.x; r.v[1] = t.lo.y; r.v[2] = t.lo.z; r.v[3] = t.lo.w; r.v[4] = t.hi.x; r.v[5] = t.hi.y; r.v[6] = t.hi.z; r.v[7] = t.hi.w;
  return r;
}
__device__ __forceinline__ row8 load_row8(const float* p) {
  float4 a = load4(p), b = load4(p + 4);
  row8 r; r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  return r;
}
__device__ __forceinline__ void store_row8(bf16* p, const row8& r) {
  f32x8_t t; t.lo = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]); t.hi = make_float4(r.v[4], r.v[5], r.v[6], r.v[7]);
  store8(p, t);
}
__device__ __forceinline__ void store_row8(float* p, const row8& r) {
  store4(p, make_float4(r.v[0], r.v[1], r.v[2], r.v[3])); store4(p + 4, make_float4(r.v[4], r.v[5], r.v[6], r.v[7]));
}
// mean / rstd of the wave's row (two-pass in registers: the values are already here)
__device__ __forceinline__ void row_stats(const row8& y, float eps, float& mu, float& rs) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += y.v[j];
  mu = wave_sum(s) * (1.0f / LNX_C);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const float d = y.v[j] - mu; q = fmaf(d, d, q); }
  rs = rsqrtf(wave_sum(q) * (1.0f / LNX_C) + eps);
}

// ------------------------------------------------------------------------------------ layers 1..: LayerNorm(C) + GELU on rows
// pre [rows][512] T -> out [rows][512] T.  One wave per row, grid-stride.
template <typename T>
__global__ void __launch_bounds__(256) ln_gelu_fwd_kernel(const T* __restrict__ pre, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ out, long rows, float eps) {
  const int lane = threadIdx.x & 63, c0 = lane * 8;
  const row8 g = load_row8(gamma + c0), be = load_row8(beta + c0);
  const long w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  for (long r = w0; r < rows; r += nw) {
    row8 y = load_row8(pre + r * LNX_C + c0);
    float mu, rs;
    row_stats(y, eps, mu, rs);
#pragma unroll
    for (int j = 0; j < 8; ++j) y.v[j] = gelu_((y.v[j] - mu) * rs * g.v[j] + be.v[j]);
    store_row8(out + r * LNX_C + c0, y);
  }
}
extern "C" int lidk_ln_gelu_fwd(const void* pre, const float* gamma, const float* beta, void* out, long rows, int C, float eps,
                                int dtype, void* stream) {
  if (!pre || !gamma || !beta || !out || rows <= 0) return LIDK_ERR_ARG;
  if (C != LNX_C) return LIDK_ERR_UNSUPPORTED;
  int blocks = (int)((rows + 3) / 4); if (blocks > 8192) blocks = 8192;
  LIDK_DISPATCH(dtype, (ln_gelu_fwd_kernel<T><<<blocks, 256, 0, as_stream(stream)>>>((const T*)pre, gamma, beta, (T*)out, rows, eps)));
  return launch_status();
}

// Backward of the same: dy [B*P][512] T = gradient at the GELU output, pre as in the forward (statistics are recomputed: a row is
// 1 KB) -> dpre [B*P][512] T; rows t >= Tv of an utterance (pitch padding) get dpre = 0 and add nothing to (dgamma | dbeta), which
// are ACCUMULATED with one atomic per channel and workgroup.
template <typename T>
__global__ void __launch_bounds__(256) ln_gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ pre,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          T* __restrict__ dpre, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          int B, int P, int Tv, float eps) {
  __shared__ float red[4][2][LNX_C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c0 = lane * 8;
  const row8 g = load_row8(gamma + c0), be = load_row8(beta + c0);
  float ag[8], ab[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ag[j] = ab[j] = 0.f;
  const long rows = (long)B * P;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const int t = (int)(r % P);
    row8 d;
    if (t < Tv) {
      const row8 y = load_row8(pre + r * LNX_C + c0);
      const row8 up = load_row8(dy + r * LNX_C + c0);
      float mu, rs;
      row_stats(y, eps, mu, rs);
      float xh[8], dxh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[j] = (y.v[j] - mu) * rs;
        const float dz = up.v[j] * gelu_grad_(xh[j] * g.v[j] + be.v[j]);
        ag[j] = fmaf(dz, xh[j], ag[j]); ab[j] += dz;
        dxh[j] = dz * g.v[j];
        s1 += dxh[j]; s2 = fmaf(dxh[j], xh[j], s2);
      }
      s1 = wave_sum(s1) * (1.0f / LNX_C); s2 = wave_sum(s2) * (1.0f / LNX_C);
#pragma unroll
      for (int j = 0; j < 8; ++j) d.v[j] = rs * (dxh[j] - s1 - xh[j] * s2);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) d.v[j] = 0.f;
    }
    store_row8(dpre + r * LNX_C + c0, d);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[wave][0][c0 + j] = ag[j]; red[wave][1][c0 + j] = ab[j]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * LNX_C; i += 256) {
    const int which = i / LNX_C, c = i % LNX_C;
    const float v = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    if (v != 0.f) atomicAdd((which ? dbeta : dgamma) + c, v);
  }
}
extern "C" int lidk_ln_gelu_bwd(const void* dy, const void* pre, const float* gamma, const float* beta, void* dpre, float* dgamma,
                                float* dbeta, int B, int P, int Tv, int C, float eps, int dtype, void* stream) {
  if (!dy || !pre || !gamma || !beta || !dpre || !dgamma || !dbeta || B <= 0 || P <= 0 || Tv <= 0 || Tv > P) return LIDK_ERR_ARG;
  if (C != LNX_C) return LIDK_ERR_UNSUPPORTED;
  const long rows = (long)B * P;
  int blocks = (int)((rows + 3) / 4); if (blocks > 1024) blocks = 1024;
  LIDK_DISPATCH(dtype, (ln_gelu_bwd_kernel<T><<<blocks, 256, 0, as_stream(stream)>>>((const T*)dy, (const T*)pre, gamma, beta, (T*)dpre,
                                                                                    dgamma, dbeta, B, P, Tv, eps)));
  return launch_status();
}

// ------------------------------------------------------------------------------------ layer 0: conv k10 s5 (+bias) + LN(C) + GELU
// wav [B][L] f32 -> out [B*P0][512] bf16 (rows T0..P0-1 of an utterance zero).  A workgroup stages the 64*5+5 samples of its 64
// frames in LDS; a wave walks 16 frames with the 8 x 10 filter taps of its lane's channels in registers.
__device__ __forceinline__ row8 conv0_row(const float (&wk)[8][LNX_K], const row8& b, const float* xs) {
  row8 y = b;
#pragma unroll
  for (int k = 0; k < LNX_K; ++k) {
    const float x = xs[k];
#pragma unroll
    for (int j = 0; j < 8; ++j) y.v[j] = fmaf(wk[j][k], x, y.v[j]);
  }
  return y;
}
__global__ void __launch_bounds__(256)
conv0_ln_fwd_kernel(const float* __restrict__ wav, int L, const float* __restrict__ w, const float* __restrict__ bias,
                    const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ out, int T0, int P0, float eps) {
  __shared__ float xs[LNX_TC * LNX_S + LNX_K];
  const int b = blockIdx.y, t0 = blockIdx.x * LNX_TC;
  const int nt = max(0, min(LNX_TC, T0 - t0)), np = min(LNX_TC, P0 - t0);
  const float* x = wav + (size_t)b * L + (size_t)t0 * LNX_S;
  if (nt > 0) for (int i = threadIdx.x; i < nt * LNX_S + LNX_K - LNX_S; i += 256) xs[i] = x[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c0 = lane * 8;
  float wk[8][LNX_K];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < LNX_K; ++k) wk[j][k] = w[(c0 + j) * LNX_K + k];
  row8 bi;
#pragma unroll
  for (int j = 0; j < 8; ++j) bi.v[j] = bias ? bias[c0 + j] : 0.f;
  const row8 g = load_row8(gamma + c0), be = load_row8(beta + c0);
  for (int t = wave; t < np; t += 4) {
    row8 y;
    if (t < nt) {
      y = conv0_row(wk, bi, xs + t * LNX_S);
      float mu, rs;
      row_stats(y, eps, mu, rs);
#pragma unroll
      for (int j = 0; j < 8; ++j) y.v[j] = gelu_((y.v[j] - mu) * rs * g.v[j] + be.v[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) y.v[j] = 0.f;
    }
    store_row8(out + ((size_t)b * P0 + t0 + t) * LNX_C + c0, y);
  }
}
extern "C" int lidk_conv0_ln_fwd(const float* wav, int B, int L, const float* w, const float* bias, const float* gamma,
                                 const float* beta, float eps, void* out, int T0, int P0, int C, void* stream) {
  if (!wav || !w || !gamma || !beta || !out || B <= 0 || T0 <= 0 || P0 < T0) return LIDK_ERR_ARG;
  if ((long)(T0 - 1) * LNX_S + LNX_K > L) return LIDK_ERR_ARG;
  if (C != LNX_C) return LIDK_ERR_UNSUPPORTED;
  conv0_ln_fwd_kernel<<<dim3(cdiv(P0, LNX_TC), B), 256, 0, as_stream(stream)>>>(wav, L, w, bias, gamma, beta, (bf16*)out, T0, P0, eps);
  return launch_status();
}

// Layer 0 backward (the extractor un-frozen): dy0 [B*P0][512] bf16 = gradient at the layer's output.  The convolution and its row
// statistics are recomputed from the waveform; per row dz = dy0 * gelu'(z), the LayerNorm backward gives the gradient at the conv
// output dc, and dW[c][k] += dc * wav[5 t + k], dbias[c] += dc, dgamma[c] += dz * xhat, dbeta[c] += dz - accumulated in registers
// over all the frames a wave meets (a bounded grid walks the (utterance, 64-frame chunk) items) and flushed once per workgroup.
#define LNX_NACC (LNX_K + 3)      // per channel: 10 filter taps, bias, gamma, beta
__global__ void __launch_bounds__(256)
conv0_ln_bwd_kernel(const float* __restrict__ wav, int L, const float* __restrict__ w, const float* __restrict__ bias,
                    const float* __restrict__ gamma, const float* __restrict__ beta, const bf16* __restrict__ dy0,
                    float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dgamma, float* __restrict__ dbeta, int B,
                    int T0, int P0, float eps) {
  __shared__ float xs[LNX_TC * LNX_S + LNX_K];
  __shared__ float red[LNX_C * LNX_NACC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c0 = lane * 8;
  float wk[8][LNX_K], acc[8][LNX_NACC];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int k = 0; k < LNX_K; ++k) wk[j][k] = w[(c0 + j) * LNX_K + k];
#pragma unroll
    for (int k = 0; k < LNX_NACC; ++k) acc[j][k] = 0.f;
  }
  row8 bi;
#pragma unroll
  for (int j = 0; j < 8; ++j) bi.v[j] = bias ? bias[c0 + j] : 0.f;
  const row8 g = load_row8(gamma + c0), be = load_row8(beta + c0);
  const int nchunk = (T0 + LNX_TC - 1) / LNX_TC;
  for (long item = blockIdx.x; item < (long)B * nchunk; item += gridDim.x) {
    const int b = (int)(item / nchunk), t0 = (int)(item % nchunk) * LNX_TC;
    const int nt = min(LNX_TC, T0 - t0);
    __syncthreads();
    const float* x = wav + (size_t)b * L + (size_t)t0 * LNX_S;
    for (int i = threadIdx.x; i < nt * LNX_S + LNX_K - LNX_S; i += 256) xs[i] = x[i];
    __syncthreads();
    for (int t = wave; t < nt; t += 4) {
      const row8 y = conv0_row(wk, bi, xs + t * LNX_S);
      const row8 up = load_row8(dy0 + ((size_t)b * P0 + t0 + t) * LNX_C + c0);
      float mu, rs;
      row_stats(y, eps, mu, rs);
      float xh[8], dxh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[j] = (y.v[j] - mu) * rs;
        const float dz = up.v[j] * gelu_grad_(xh[j] * g.v[j] + be.v[j]);
        acc[j][LNX_K + 1] = fmaf(dz, xh[j], acc[j][LNX_K + 1]); acc[j][LNX_K + 2] += dz;
        dxh[j] = dz * g.v[j];
        s1 += dxh[j]; s2 = fmaf(dxh[j], xh[j], s2);
      }
      s1 = wave_sum(s1) * (1.0f / LNX_C); s2 = wave_sum(s2) * (1.0f / LNX_C);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dc = rs * (dxh[j] - s1 - xh[j] * s2);
        acc[j][LNX_K] += dc;
#pragma unroll
        for (int k = 0; k < LNX_K; ++k) acc[j][k] = fmaf(dc, xs[t * LNX_S + k], acc[j][k]);
      }
    }
  }
  // workgroup reduction: the four waves add their registers into LDS one after the other, then one atomic per value
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < LNX_NACC; ++k) {
          float* p = red + (c0 + j) * LNX_NACC + k;
          *p = (wv == 0 ? 0.f : *p) + acc[j][k];
        }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < LNX_C * LNX_NACC; i += 256) {
    const int c = i / LNX_NACC, k = i % LNX_NACC;
    const float v = red[i];
    if (v == 0.f) continue;
    if (k < LNX_K) atomicAdd(dw + c * LNX_K + k, v);
    else if (k == LNX_K) { if (dbias) atomicAdd(dbias + c, v); }
    else if (k == LNX_K + 1) atomicAdd(dgamma + c, v);
    else atomicAdd(dbeta + c, v);
  }
}
extern "C" int lidk_conv0_ln_bwd(const float* wav, int B, int L, const float* w, const float* bias, const float* gamma,
                                 const float* beta, float eps, const void* dy0, float* dw, float* dbias, float* dgamma, float* dbeta,
                                 int T0, int P0, int C, void* stream) {
  if (!wav || !w || !gamma || !beta || !dy0 || !dw || !dgamma || !dbeta || B <= 0 || T0 <= 0 || P0 < T0) return LIDK_ERR_ARG;
  if ((long)(T0 - 1) * LNX_S + LNX_K > L) return LIDK_ERR_ARG;
  if (C != LNX_C) return LIDK_ERR_UNSUPPORTED;
  const long items = (long)B * cdiv(T0, LNX_TC);
  const int blocks = (int)(items < 512 ? items : 512);
  conv0_ln_bwd_kernel<<<blocks, 256, 0, as_stream(stream)>>>(wav, L, w, bias, gamma, beta, (const bf16*)dy0, dw, dbias, dgamma, dbeta,
                                                            B, T0, P0, eps);
  return launch_status();
}
