// WavLM backbone pieces that are not plain GEMM / LayerNorm launches (lid/wavlm/WavLM.py, lid/wavlm/modules.py), forward pass:
//   * conv layer 0 of the feature extractor: Conv1d(1, C, k10, s5, no bias) + GroupNorm(C groups = per-channel statistics over
//     time) + GELU, channel-last output.  K = 10 is far too thin for MFMA and the input is the raw waveform, so the convolution is
//     recomputed in both passes instead of storing its f32 pre-norm output (9,599 x 512 per 3 s utterance): pass 1 reduces
//     per-(utterance, channel) sums, pass 2 normalises + GELU and writes bf16.  (Layers 1-6 are GEMMs over strided views of the
//     channel-last signal, see lidk_gemm_nt: lda < K.)
//   * the grouped positional convolution's operand layout: group-major, zero-padded copies so that each group's k128 window is
//     contiguous and the convolution is 16 GEMMs with lda = 48;
//   * the gated relative position bias (GRU-style gate from the layer input, modules.py:519-528) and the attention core with an
//     additive (gate x bucketed relative bias) term: one workgroup per (batch, head), K / V^T / the head's 1-D bias table in LDS,
//     S = Q.K^T and O = P.V on v_mfma_f32_16x16x32_bf16, softmax by 16-lane shuffles.
#include "common.h"

#define W0_K 10
#define W0_S 5
#define W0_TC 128          // time steps per workgroup

// ------------------------------------------------------------------------------------ conv layer 0
// partial [B][nchunk][C][2]
__global__ void __launch_bounds__(256)
wavlm_conv0_stats_kernel(const float* __restrict__ wav, int L, const float* __restrict__ w, float* __restrict__ partial, int T0,
                         int C) {
  __shared__ float xs[W0_TC * W0_S + W0_K];
  const int b = blockIdx.y, t0 = blockIdx.x * W0_TC, nt = min(W0_TC, T0 - t0);
  const float* x = wav + (size_t)b * L + (size_t)t0 * W0_S;
  for (int i = threadIdx.x; i < nt * W0_S + W0_K - W0_S; i += 256) xs[i] = x[i];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float wk[W0_K];
#pragma unroll
    for (int k = 0; k < W0_K; ++k) wk[k] = w[c * W0_K + k];
    float s = 0.f, ss = 0.f;
    for (int t = 0; t < nt; ++t) {
      float y = 0.f;
#pragma unroll
      for (int k = 0; k < W0_K; ++k) y = fmaf(wk[k], xs[t * W0_S + k], y);
      s += y; ss = fmaf(y, y, ss);
    }
    float* p = partial + (((size_t)b * gridDim.x + blockIdx.x) * C + c) * 2;
    p[0] = s; p[1] = ss;
  }
}
// stats [B][C][2] = (mean, rstd), biased variance (nn.GroupNorm)
__global__ void wavlm_gn_finalize_kernel(const float* __restrict__ partial, int nchunk, int C, int T0, float eps,
                                         float* __restrict__ stats) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < nchunk; ++k) {
      const float* p = partial + (((size_t)b * nchunk + k) * C + c) * 2;
      s += (double)p[0]; ss += (double)p[1];
    }
    const double mu = s / T0;
    double var = ss / T0 - mu * mu;
    if (var < 0) var = 0;
    stats[((size_t)b * C + c) * 2] = (float)mu;
    stats[((size_t)b * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
// out [B*P0][C] bf16 (row pitch P0 >= T0; rows T0..P0-1 are zero)
__global__ void __launch_bounds__(256)
wavlm_conv0_apply_kernel(const float* __restrict__ wav, int L, const float* __restrict__ w, const float* __restrict__ stats,
                         const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ out, int T0, int P0,
                         int C) {
  __shared__ float xs[W0_TC * W0_S + W0_K];
  const int b = blockIdx.y, t0 = blockIdx.x * W0_TC;
  const int nt = max(0, min(W0_TC, T0 - t0)), np = min(W0_TC, P0 - t0);
  const float* x = wav + (size_t)b * L + (size_t)t0 * W0_S;
  if (nt > 0) for (int i = threadIdx.x; i < nt * W0_S + W0_K - W0_S; i += 256) xs[i] = x[i];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float wk[W0_K];
#pragma unroll
    for (int k = 0; k < W0_K; ++k) wk[k] = w[c * W0_K + k];
    const float mu = stats[((size_t)b * C + c) * 2], rs = stats[((size_t)b * C + c) * 2 + 1], g = gamma[c], be = beta[c];
    bf16* o = out + ((size_t)b * P0 + t0) * C + c;
    for (int t = 0; t < np; ++t) {
      float v = 0.f;
      if (t < nt) {
        float y = 0.f;
#pragma unroll
        for (int k = 0; k < W0_K; ++k) y = fmaf(wk[k], xs[t * W0_S + k], y);
        v = gelu_((y - mu) * rs * g + be);
      }
      o[(size_t)t * C] = (bf16)v;
    }
  }
}

extern "C" long lidk_wavlm_conv0_workspace(int B, int T0, int C) {
  return ((long)B * cdiv(T0, W0_TC) * C * 2 + (long)B * C * 2) * (long)sizeof(float);
}
extern "C" int lidk_wavlm_conv0(const float* wav, int B, int L, const float* w, const float* gamma, const float* beta, float eps,
                                void* out, int T0, int P0, int C, float* workspace, void* stream) {
  if (!wav || !w || !gamma || !beta || !out || !workspace || B <= 0 || C <= 0 || T0 <= 0 || P0 < T0) return LIDK_ERR_ARG;
  if ((long)(T0 - 1) * W0_S + W0_K > L) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  const int nchunk = cdiv(T0, W0_TC);
  float* partial = workspace;
  float* stats = workspace + (size_t)B * nchunk * C * 2;
  wavlm_conv0_stats_kernel<<<dim3(nchunk, B), 256, 0, s>>>(wav, L, w, partial, T0, C);
  wavlm_gn_finalize_kernel<<<B, 256, 0, s>>>(partial, nchunk, C, T0, eps, stats);
  wavlm_conv0_apply_kernel<<<dim3(cdiv(P0, W0_TC), B), 256, 0, s>>>(wav, L, w, stats, gamma, beta, (bf16*)out, T0, P0, C);
  return launch_status();
}

// ------------------------------------------------------------------------------------ positional conv operand layout
// x [B*T][C] f32 -> xg [G][B*Pp + slack][C/G] bf16: row (b, u) = x[b][u - pad_left] (zero outside [0, T)); rows behind B*Pp zero.
__global__ void wavlm_posconv_prep_kernel(const float* __restrict__ x, bf16* __restrict__ xg, int B, int T_, int C, int G, int Pp,
                                          int pad_left, long rows_total) {
  const int cg = C / G;
  const long n = rows_total * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long row = i / C;                         // over rows_total = B*Pp + slack
    float v = 0.f;
    if (row < (long)B * Pp) {
      const int b = (int)(row / Pp), u = (int)(row % Pp) - pad_left;
      if (u >= 0 && u < T_) v = x[((size_t)b * T_ + u) * C + c];
    }
    const int g = c / cg, cc = c - g * cg;
    xg[((size_t)g * rows_total + row) * cg + cc] = (bf16)v;
  }
}
extern "C" int lidk_wavlm_posconv_prep(const float* x, void* xg, int B, int T_, int C, int G, int Pp, int pad_left, long rows_total,
                                       void* stream) {
  if (!x || !xg || B <= 0 || T_ <= 0 || C <= 0 || G <= 0 || C % G || Pp < T_ + pad_left || rows_total < (long)B * Pp) return LIDK_ERR_ARG;
  const long n = rows_total * C;
  int blocks = (int)((n + 255) / 256); if (blocks > 16384) blocks = 16384;
  wavlm_posconv_prep_kernel<<<blocks, 256, 0, as_stream(stream)>>>(x, (bf16*)xg, B, T_, C, G, Pp, pad_left, rows_total);
  return launch_status();
}
// out[b][t][:] = x[b][t][:] + y[b*Pp + t][:]   (y holds Pp rows per utterance, the first T are valid)
__global__ void wavlm_add_rows_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int T_,
                                      int Pp, int C, long n4) {
  const int c4 = C / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long row = i / c4; const int c = (int)(i % c4) * 4;
    const long b = row / T_, t = row % T_;
    float4 a = load4(x + row * C + c), d = load4(y + (b * Pp + t) * C + c);
    a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
    store4(out + row * C + c, a);
  }
}
extern "C" int lidk_wavlm_add_rows(const float* x, const float* y, float* out, int B, int T_, int Pp, int C, void* stream) {
  if (!x || !y || !out || B <= 0 || T_ <= 0 || Pp < T_ || C <= 0 || (C & 3)) return LIDK_ERR_ARG;
  const long n4 = (long)B * T_ * C / 4;
  int blocks = (int)((n4 + 255) / 256); if (blocks > 16384) blocks = 16384;
  wavlm_add_rows_kernel<<<blocks, 256, 0, as_stream(stream)>>>(x, y, out, T_, Pp, C, n4);
  return launch_status();
}

// ------------------------------------------------------------------------------------ span masking (training)
// WavLM.apply_mask (WavLM.py:300-337): x[b][t][:] = mask_emb where time_mask[b][t]; then x[b][:][c] = 0 where chan_mask[b][c].
__global__ void wavlm_apply_mask_kernel(float* __restrict__ x, const unsigned char* __restrict__ tmask,
                                        const unsigned char* __restrict__ cmask, const float* __restrict__ mask_emb, int T_, int C,
                                        long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long row = i / C; const long b = row / T_;
    float v = x[i];
    if (tmask && tmask[row]) v = mask_emb[c];
    if (cmask && cmask[b * C + c]) v = 0.f;
    x[i] = v;
  }
}
extern "C" int lidk_wavlm_apply_mask(float* x, const unsigned char* time_mask, const unsigned char* chan_mask, const float* mask_emb,
                                     int B, int T_, int C, void* stream) {
  if (!x || (time_mask && !mask_emb) || B <= 0 || T_ <= 0 || C <= 0) return LIDK_ERR_ARG;
  const long n = (long)B * T_ * C;
  int blocks = (int)((n + 255) / 256); if (blocks > 16384) blocks = 16384;
  wavlm_apply_mask_kernel<<<blocks, 256, 0, as_stream(stream)>>>(x, time_mask, chan_mask, mask_emb, T_, C, n);
  return launch_status();
}

// ------------------------------------------------------------------------------------ gate of the relative position bias
// modules.py:519-528: per (b, h, t): u = grep_linear(x[b][t][h*dh : (h+1)*dh]) (8 values) ; (ga, gb) = sigmoid of the sums of
// u[0:4], u[4:8] ; gate = ga * (gb * grep_a[h] - 1) + 2.   x is the LAYER INPUT (not the projected query).
__global__ void __launch_bounds__(256)
wavlm_gate_kernel(const float* __restrict__ x, const float* __restrict__ wg, const float* __restrict__ bg,
                  const float* __restrict__ grep_a, float* __restrict__ gate, int B, int T_, int H, int dh) {
  const long n = (long)B * T_ * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int h = (int)(i % H); const long m = i / H;
    const float* xr = x + m * (long)H * dh + (long)h * dh;
    float u[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) u[q] = bg[q];
    for (int d = 0; d < dh; ++d) {
      const float v = xr[d];
#pragma unroll
      for (int q = 0; q < 8; ++q) u[q] = fmaf(wg[q * dh + d], v, u[q]);
    }
    const float ga = 1.f / (1.f + __expf(-(u[0] + u[1] + u[2] + u[3])));
    const float gb = 1.f / (1.f + __expf(-(u[4] + u[5] + u[6] + u[7])));
    const long b = m / T_, t = m % T_;
    gate[(b * H + h) * T_ + t] = ga * (gb * grep_a[h] - 1.f) + 2.f;
  }
}
extern "C" int lidk_wavlm_gate(const float* x, const float* wg, const float* bg, const float* grep_a, float* gate, int B, int T_,
                               int H, int dh, void* stream) {
  if (!x || !wg || !bg || !grep_a || !gate || B <= 0 || T_ <= 0 || H <= 0 || dh <= 0) return LIDK_ERR_ARG;
  const long n = (long)B * T_ * H;
  int blocks = (int)((n + 255) / 256); if (blocks > 16384) blocks = 16384;
  wavlm_gate_kernel<<<blocks, 256, 0, as_stream(stream)>>>(x, wg, bg, grep_a, gate, B, T_, H, dh);
  return launch_status();
}

// ------------------------------------------------------------------------------------ attention with gated relative bias (forward)
// scores[i][j] = scale * q_i.k_j + gate[b][h][i] * rb[h][j - i] ; probs = softmax_j ; out = probs.v
// qkv [B*T][3*H*DH] bf16 (q | k | v column blocks, head h at columns h*DH); rb [H][2*RB-1] f32, entry r + RB - 1 for offset r = j - i.
#define WA_KPAD 8
template <int DH, bool WP>         // WP: also write the probabilities (training: the backward pass reads them)
__global__ void __launch_bounds__(1024)
wavlm_attn_fwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ gate, const float* __restrict__ rb,
                      bf16* __restrict__ out, bf16* __restrict__ probs, int T_, int H, int RB, float scale, int NJ) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Tp = NJ * 16, Tp32 = (Tp + 31) / 32 * 32, LDK = DH + WA_KPAD, LDV = Tp32 + WA_KPAD;
  bf16* Ks = reinterpret_cast<bf16*>(smem);                      // [Tp][LDK]
  bf16* Vt = Ks + (size_t)Tp * LDK;                              // [DH][LDV]
  float* rbs = reinterpret_cast<float*>(Vt + (size_t)DH * LDV);  // [2*Tp]: rbs[r + Tp]
  bf16* Pw = reinterpret_cast<bf16*>(rbs + 2 * Tp);              // [nw][16][LDV]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x % H, inner = H * DH, ld = 3 * inner;
  const bf16* base = qkv + (size_t)b * T_ * ld + h * DH;
  for (int idx = threadIdx.x; idx < Tp * (DH / 8); idx += blockDim.x) {            // K rows, 16 B per access
    const int j = idx / (DH / 8), c8 = (idx % (DH / 8)) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (j < T_) v = *reinterpret_cast<const uint4*>(base + (size_t)j * ld + inner + c8);
    *reinterpret_cast<uint4*>(&Ks[j * LDK + c8]) = v;
  }
  for (int idx = threadIdx.x; idx < Tp32 * DH; idx += blockDim.x) {                 // V transposed
    const int j = idx / DH, d = idx % DH;
    Vt[d * LDV + j] = j < T_ ? base[(size_t)j * ld + 2 * inner + d] : (bf16)0.f;
  }
  for (int idx = threadIdx.x; idx < 2 * Tp; idx += blockDim.x) {
    const int r = idx - Tp;
    rbs[idx] = (r > -T_ && r < T_) ? rb[(size_t)h * (2 * RB - 1) + r + RB - 1] : 0.f;
  }
  __syncthreads();
  bf16* P = Pw + (size_t)wave * 16 * LDV;
  const float* grow = gate + ((size_t)b * H + h) * T_;
  for (int qb = wave; qb * 16 < T_; qb += nw) {
    const int i0 = qb * 16;
    bf16x8 aq[DH / 32];
#pragma unroll
    for (int kk = 0; kk < DH / 32; ++kk)
      aq[kk] = *reinterpret_cast<const bf16x8*>(base + (size_t)min(i0 + fr, T_ - 1) * ld + kk * 32 + fq * 8);
    float gi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) gi[r] = grow[min(i0 + 4 * fq + r, T_ - 1)];
    // ---- S tiles: lane holds rows i0 + 4*fq + r (r = 0..3), key column jt*16 + fr
    f32x4 S[16];                                          // NJ <= 16
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) {
      if (jt < NJ) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DH / 32; ++kk) {
          const bf16x8 bk = *reinterpret_cast<const bf16x8*>(&Ks[(jt * 16 + fr) * LDK + kk * 32 + fq * 8]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[kk], bk, acc, 0, 0, 0);
        }
        const int j = jt * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + 4 * fq + r;
          float s = acc[r] * scale + gi[r] * rbs[j - i + Tp];
          if (j >= T_) s = -INFINITY;
          acc[r] = s;
          mx[r] = fmaxf(mx[r], s);
        }
        S[jt] = acc;
      }
    }
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      for (int o = 1; o < 16; o <<= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
    }
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) {
      if (jt < NJ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float e = __expf(S[jt][r] - mx[r]); S[jt][r] = e; sum[r] += e; }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      for (int o = 1; o < 16; o <<= 1) sum[r] += __shfl_xor(sum[r], o, 64);
      sum[r] = 1.f / sum[r];
    }
    // ---- P (bf16) -> this wave's LDS tile [16][LDV], zero padded to Tp32
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) {
      if (jt < NJ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(4 * fq + r) * LDV + jt * 16 + fr] = (bf16)(S[jt][r] * sum[r]);
      }
    }
    if (Tp32 > Tp) {
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(4 * fq + r) * LDV + Tp + fr] = (bf16)0.f;
    }
    __builtin_amdgcn_wave_barrier();
    if (WP) {                                             // probs [B][H][T][Tp32] bf16: this wave's 16 rows, 16 bytes per store
      const int cpr = Tp32 / 8;
      for (int c = lane; c < 16 * cpr; c += 64) {
        const int rr = c / cpr, cc = (c - rr * cpr) * 8, i = i0 + rr;
        if (i < T_)
          *reinterpret_cast<uint4*>(probs + (((size_t)b * H + h) * T_ + i) * Tp32 + cc) = *reinterpret_cast<const uint4*>(&P[rr * LDV + cc]);
      }
    }
    // ---- O = P.V : A = P rows (queries, k = keys), B = V^T rows (d, k = keys)
    f32x4 O[DH / 16];
#pragma unroll
    for (int nt = 0; nt < DH / 16; ++nt) O[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < Tp32 / 32; ++kt) {
      const bf16x8 ap = *reinterpret_cast<const bf16x8*>(&P[fr * LDV + kt * 32 + fq * 8]);
#pragma unroll
      for (int nt = 0; nt < DH / 16; ++nt) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(&Vt[(nt * 16 + fr) * LDV + kt * 32 + fq * 8]);
        O[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap, bv, O[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < DH / 16; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * fq + r;
        if (i < T_) out[((size_t)b * T_ + i) * inner + h * DH + nt * 16 + fr] = (bf16)O[nt][r];
      }
    __builtin_amdgcn_wave_barrier();
  }
}

static size_t wavlm_attn_lds(int NJ, int DH, int nw) {
  const int Tp = NJ * 16, Tp32 = (Tp + 31) / 32 * 32;
  return (size_t)Tp * (DH + WA_KPAD) * 2 + (size_t)DH * (Tp32 + WA_KPAD) * 2 + (size_t)2 * Tp * 4 + (size_t)nw * 16 * (Tp32 + WA_KPAD) * 2;
}
extern "C" int lidk_wavlm_attn_max_frames(int dh) { return dh == 64 ? 256 : 0; }
extern "C" int lidk_wavlm_attn_fwd(const void* qkv, const float* gate, const float* rb, void* out, int B, int T_, int H, int dh,
                                   int RB, void* stream) {
  if (!qkv || !gate || !rb || !out || B <= 0 || T_ <= 0 || H <= 0 || RB < T_) return LIDK_ERR_ARG;
  if (dh != 64 || T_ > 256) return LIDK_ERR_UNSUPPORTED;
  const int NJ = cdiv(T_, 16);
  int nw = NJ < 16 ? NJ : 16;
  while (nw > 1 && wavlm_attn_lds(NJ, 64, nw) > 160 * 1024) --nw;
  const size_t lds = wavlm_attn_lds(NJ, 64, nw);
  if (lds > 160 * 1024) return LIDK_ERR_UNSUPPORTED;
  (void)hipFuncSetAttribute((const void*)wavlm_attn_fwd_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  wavlm_attn_fwd_kernel<64, false><<<B * H, 64 * nw, lds, as_stream(stream)>>>((const bf16*)qkv, gate, rb, (bf16*)out, nullptr, T_, H,
                                                                               RB, 1.0f / sqrtf(64.0f), NJ);
  return launch_status();
}
// The same forward, also storing probs [B][H][T][ldp] bf16 (ldp = lidk_wavlm_attn_ldp(T)) for the backward pass.
extern "C" int lidk_wavlm_attn_fwd_probs(const void* qkv, const float* gate, const float* rb, void* out, void* probs, int B, int T_,
                                         int H, int dh, int RB, void* stream) {
  if (!qkv || !gate || !rb || !out || !probs || B <= 0 || T_ <= 0 || H <= 0 || RB < T_) return LIDK_ERR_ARG;
  if (dh != 64 || T_ > 256) return LIDK_ERR_UNSUPPORTED;
  const int NJ = cdiv(T_, 16);
  int nw = NJ < 16 ? NJ : 16;
  while (nw > 1 && wavlm_attn_lds(NJ, 64, nw) > 160 * 1024) --nw;
  const size_t lds = wavlm_attn_lds(NJ, 64, nw);
  if (lds > 160 * 1024) return LIDK_ERR_UNSUPPORTED;
  (void)hipFuncSetAttribute((const void*)wavlm_attn_fwd_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  wavlm_attn_fwd_kernel<64, true><<<B * H, 64 * nw, lds, as_stream(stream)>>>((const bf16*)qkv, gate, rb, (bf16*)out, (bf16*)probs, T_,
                                                                              H, RB, 1.0f / sqrtf(64.0f), NJ);
  return launch_status();
}

// =====================================================================================================================
// Backward pass of the transformer part (un-frozen encoder: the reference's regime after freeze_tranformer_epoch).
// First correct versions: the attention backward is a VALU two-pass form (row pass + column pass through an f32 dS buffer),
// like attn.hip's v1 kernels, with the bias-table and gate gradients added.
// =====================================================================================================================

// Attention forward variant that also stores the probabilities (needed by backward): same kernel as above, P written out.
// probs [B][H][T][Tp32] bf16.  Implemented by a flag of wavlm_attn_fwd_kernel would cost registers in the inference path, so the
// training path runs this small extra kernel instead: it recomputes S rows (VALU) - only T*T*dh per (b,h), once per layer.
template <int DH>
__global__ void __launch_bounds__(256)
wavlm_attn_probs_kernel(const bf16* __restrict__ qkv, const float* __restrict__ gate, const float* __restrict__ rb,
                        bf16* __restrict__ probs, int T_, int H, int RB, int ldp, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int LDK = DH + 2;
  bf16* Ks = reinterpret_cast<bf16*>(smem);                                  // [T][LDK]
  float* fb = reinterpret_cast<float*>(smem + (((size_t)T_ * LDK * 2) + 15) / 16 * 16);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* qs = fb + wave * (DH + T_);
  float* ps = qs + DH;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * 16, inner = H * DH, ld = 3 * inner;
  const bf16* base = qkv + (size_t)b * T_ * ld + h * DH;
  for (int idx = threadIdx.x; idx < T_ * DH; idx += blockDim.x) {
    const int j = idx / DH, d = idx - j * DH;
    Ks[j * LDK + d] = base[(size_t)j * ld + inner + d];
  }
  __syncthreads();
  const float* rbh = rb + (size_t)h * (2 * RB - 1) + RB - 1;
  for (int ii = wave; ii < 16; ii += 4) {
    const int i = i0 + ii;
    if (i >= T_) break;
    for (int d = lane; d < DH; d += 64) qs[d] = to_f(base[(size_t)i * ld + d]);
    __builtin_amdgcn_wave_barrier();
    const float g = gate[((size_t)b * H + h) * T_ + i];
    float mx = -INFINITY;
    for (int j = lane; j < T_; j += 64) {
      float s = 0.f;
      for (int d = 0; d < DH; ++d) s = fmaf(qs[d], to_f(Ks[j * LDK + d]), s);
      s = s * scale + g * rbh[j - i];
      ps[j] = s; mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < T_; j += 64) { float e = __expf(ps[j] - mx); ps[j] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    bf16* prow = probs + (((size_t)b * H + h) * T_ + i) * ldp;
    for (int j = lane; j < ldp; j += 64) prow[j] = (bf16)(j < T_ ? ps[j] * inv : 0.f);
    __builtin_amdgcn_wave_barrier();
  }
}

// row pass: dP = dO.V^T ; delta = sum_j P dP ; dS = P (dP - delta) -> dscores (f32 [B][H][T][T]) ; dq = scale dS.K ;
// dgate[b][h][i] = sum_j dS[i][j] rb[h][j - i]
template <int DH>
__global__ void __launch_bounds__(256)
wavlm_attn_bwd_rows_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ probs, const bf16* __restrict__ dout,
                           const float* __restrict__ rb, bf16* __restrict__ dqkv, float* __restrict__ dscores,
                           float* __restrict__ dgate, int T_, int H, int RB, int ldp, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int LDK = DH + 2;
  bf16* Ks = reinterpret_cast<bf16*>(smem);
  bf16* Vs = Ks + (size_t)T_ * LDK;
  float* fb = reinterpret_cast<float*>(smem + (((size_t)2 * T_ * LDK * 2) + 15) / 16 * 16);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* dos = fb + wave * (DH + T_);
  float* ps = dos + DH;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * 16, inner = H * DH, ld = 3 * inner;
  const bf16* base = qkv + (size_t)b * T_ * ld + h * DH;
  for (int idx = threadIdx.x; idx < T_ * DH; idx += blockDim.x) {
    const int j = idx / DH, d = idx - j * DH;
    Ks[j * LDK + d] = base[(size_t)j * ld + inner + d];
    Vs[j * LDK + d] = base[(size_t)j * ld + 2 * inner + d];
  }
  __syncthreads();
  const float* rbh = rb + (size_t)h * (2 * RB - 1) + RB - 1;
  for (int ii = wave; ii < 16; ii += 4) {
    const int i = i0 + ii;
    if (i >= T_) break;
    for (int d = lane; d < DH; d += 64) dos[d] = to_f(dout[((size_t)b * T_ + i) * inner + h * DH + d]);
    __builtin_amdgcn_wave_barrier();
    const bf16* prow = probs + (((size_t)b * H + h) * T_ + i) * ldp;
    float delta = 0.f;
    for (int j = lane; j < T_; j += 64) {
      float dp = 0.f;
      for (int d = 0; d < DH; ++d) dp = fmaf(dos[d], to_f(Vs[j * LDK + d]), dp);
      ps[j] = dp;
      delta = fmaf(to_f(prow[j]), dp, delta);
    }
    delta = wave_sum(delta);
    float* dsrow = dscores + (((size_t)b * H + h) * T_ + i) * T_;
    float dg = 0.f;
    for (int j = lane; j < T_; j += 64) {
      const float ds = to_f(prow[j]) * (ps[j] - delta);
      ps[j] = ds; dsrow[j] = ds;
      dg = fmaf(ds, rbh[j - i], dg);
    }
    dg = wave_sum(dg);
    if (lane == 0) dgate[((size_t)b * H + h) * T_ + i] = dg;
    __builtin_amdgcn_wave_barrier();
    float a = 0.f;                                              // DH == 64: lane <-> d
    for (int j = 0; j < T_; ++j) a = fmaf(ps[j], to_f(Ks[j * LDK + lane]), a);
    dqkv[((size_t)b * T_ + i) * ld + h * DH + lane] = (bf16)(a * scale);
    __builtin_amdgcn_wave_barrier();
  }
}

// column pass, one item per wave: item < nkc: lane <-> key j: dk[j] = scale sum_i dS[i][j] q[i], dv[j] = sum_i P[i][j] dO[i];
// item >= nkc: lane <-> offset r = j - i: drb[h][r] += sum_i gate[i] dS[i][i + r]   (atomics over the batch)
template <int DH>
__global__ void __launch_bounds__(256)
wavlm_attn_bwd_cols_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ probs, const bf16* __restrict__ dout,
                           const float* __restrict__ dscores, const float* __restrict__ gate, bf16* __restrict__ dqkv,
                           float* __restrict__ drb, int T_, int H, int RB, int ldp, float scale) {
  __shared__ float Qs[64 * DH];
  __shared__ float Ds[64 * DH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H, inner = H * DH, ld = 3 * inner;
  const int nkc = (T_ + 63) / 64, nrc = (2 * T_ - 1 + 63) / 64;
  const int item = blockIdx.y * 4 + wave;
  const bool is_key = item < nkc, is_rel = !is_key && item < nkc + nrc;
  const int j = item * 64 + lane, r = (item - nkc) * 64 + lane - (T_ - 1);
  const size_t sbase = ((size_t)b * H + h) * T_ * T_, pbase = ((size_t)b * H + h) * T_ * ldp;
  const float* grow = gate + ((size_t)b * H + h) * T_;
  float ak[DH], av[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { ak[d] = 0.f; av[d] = 0.f; }
  float arel = 0.f;
  for (int ic0 = 0; ic0 < T_; ic0 += 64) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * DH; idx += blockDim.x) {
      const int ii = idx / DH, d = idx - ii * DH, i = min(ic0 + ii, T_ - 1);
      Qs[idx] = to_f(qkv[((size_t)b * T_ + i) * ld + h * DH + d]);
      Ds[idx] = to_f(dout[((size_t)b * T_ + i) * inner + h * DH + d]);
    }
    __syncthreads();
    const int ihi_c = min(ic0 + 64, T_);
    if (is_key && j < T_) {
      for (int i = ic0; i < ihi_c; ++i) {
        const float s = dscores[sbase + (size_t)i * T_ + j];
        const float pr = to_f(probs[pbase + (size_t)i * ldp + j]);
        const float* q = Qs + (i - ic0) * DH;
        const float* dd = Ds + (i - ic0) * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) { ak[d] = fmaf(s, q[d], ak[d]); av[d] = fmaf(pr, dd[d], av[d]); }
      }
    } else if (is_rel && r <= T_ - 1) {
      const int ilo = max(max(0, -r), ic0), ihi = min(min(T_ - 1, T_ - 1 - r), ihi_c - 1);      // 0 <= i + r <= T-1
      for (int i = ilo; i <= ihi; ++i) arel = fmaf(grow[i], dscores[sbase + (size_t)i * T_ + (i + r)], arel);
    }
  }
  if (is_key && j < T_) {
    bf16* krow = dqkv + ((size_t)b * T_ + j) * ld + inner + h * DH;
    bf16* vrow = krow + inner;
#pragma unroll
    for (int d = 0; d < DH; ++d) { krow[d] = (bf16)(ak[d] * scale); vrow[d] = (bf16)av[d]; }
  } else if (is_rel && r <= T_ - 1 && r >= -(T_ - 1)) {
    atomicAdd(&drb[(size_t)h * (2 * RB - 1) + r + RB - 1], arel);
  }
}

extern "C" int lidk_wavlm_attn_ldp(int T_) { return (T_ + 31) / 32 * 32; }
extern "C" int lidk_wavlm_attn_probs(const void* qkv, const float* gate, const float* rb, void* probs, int B, int T_, int H, int dh,
                                     int RB, void* stream) {
  if (!qkv || !gate || !rb || !probs || B <= 0 || T_ <= 0 || H <= 0 || RB < T_) return LIDK_ERR_ARG;
  if (dh != 64 || T_ > 256) return LIDK_ERR_UNSUPPORTED;
  const size_t lds = (((size_t)T_ * 66 * 2) + 15) / 16 * 16 + 4 * (size_t)(64 + T_) * 4;
  (void)hipFuncSetAttribute((const void*)wavlm_attn_probs_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  wavlm_attn_probs_kernel<64><<<dim3(cdiv(T_, 16), H, B), 256, lds, as_stream(stream)>>>(
      (const bf16*)qkv, gate, rb, (bf16*)probs, T_, H, RB, lidk_wavlm_attn_ldp(T_), 0.125f);
  return launch_status();
}
extern "C" int lidk_wavlm_attn_bwd(const void* qkv, const void* probs, const void* dout, const float* gate, const float* rb,
                                   void* dqkv, float* dgate, float* drb, float* dscores, int B, int T_, int H, int dh, int RB,
                                   void* stream) {
  if (!qkv || !probs || !dout || !gate || !rb || !dqkv || !dgate || !drb || !dscores || B <= 0 || T_ <= 0 || H <= 0 || RB < T_)
    return LIDK_ERR_ARG;
  if (dh != 64 || T_ > 256) return LIDK_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const int ldp = lidk_wavlm_attn_ldp(T_);
  const size_t lds = (((size_t)2 * T_ * 66 * 2) + 15) / 16 * 16 + 4 * (size_t)(64 + T_) * 4;
  (void)hipFuncSetAttribute((const void*)wavlm_attn_bwd_rows_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  wavlm_attn_bwd_rows_kernel<64><<<dim3(cdiv(T_, 16), H, B), 256, lds, s>>>(
      (const bf16*)qkv, (const bf16*)probs, (const bf16*)dout, rb, (bf16*)dqkv, dscores, dgate, T_, H, RB, ldp, 0.125f);
  const int items = (T_ + 63) / 64 + (2 * T_ - 1 + 63) / 64;
  wavlm_attn_bwd_cols_kernel<64><<<dim3(B * H, cdiv(items, 4)), 256, 0, s>>>(
      (const bf16*)qkv, (const bf16*)probs, (const bf16*)dout, dscores, gate, (bf16*)dqkv, drb, T_, H, RB, ldp, 0.125f);
  return launch_status();
}

// ------------------------------------------------------------------------------------ gradients of the gated bias
// From dS [B][H][T][ldp] bf16 (left behind by lidk_attn_bwd's MFMA path, which computes dQ/dK/dV: with a zero relative-position
// table the Conformer's attention backward IS this layer's - the additive bias only enters through the saved probabilities):
//   dgate[b][h][i] = sum_j dS[i][j] rb[h][j - i]          drb[h][r] += sum_{b,i} gate[b][h][i] dS[i][i + r]
// One workgroup per (b, h): the dS tile is staged in LDS once (16-byte loads); a thread then owns one row (dgate) and, in a
// second pass, one offset r (drb: a walk down the r-th diagonal, conflict-free for consecutive r); one global atomic per
// offset and workgroup.
__global__ void __launch_bounds__(256)
wavlm_attn_bias_grads_kernel(const bf16* __restrict__ ds, const float* __restrict__ gate, const float* __restrict__ rb,
                             float* __restrict__ dgate, float* __restrict__ drb, int T_, int H, int RB, int ldp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int LDD = ldp + 8;                                        // bf16 elements per LDS row (16-byte aligned rows)
  bf16* D = reinterpret_cast<bf16*>(smem);                        // [T][LDD]
  float* gs = reinterpret_cast<float*>(smem + (((size_t)T_ * LDD * 2) + 15) / 16 * 16);   // [T]
  float* rs = gs + T_;                                            // [2T - 1]: rb[h][r], entry r + T - 1
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const bf16* base = ds + ((size_t)b * H + h) * T_ * ldp;
  const int cpr = ldp / 8;
  for (int c = threadIdx.x; c < T_ * cpr; c += blockDim.x) {
    const int i = c / cpr, cc = (c - i * cpr) * 8;
    *reinterpret_cast<uint4*>(&D[i * LDD + cc]) = *reinterpret_cast<const uint4*>(base + (size_t)i * ldp + cc);
  }
  for (int i = threadIdx.x; i < T_; i += blockDim.x) gs[i] = gate[((size_t)b * H + h) * T_ + i];
  for (int i = threadIdx.x; i < 2 * T_ - 1; i += blockDim.x) rs[i] = rb[(size_t)h * (2 * RB - 1) + (i - (T_ - 1)) + RB - 1];
  __syncthreads();
  for (int i = threadIdx.x; i < T_; i += blockDim.x) {            // dgate[i] = sum_j dS[i][j] rb[j - i]
    float dg = 0.f;
    for (int j = 0; j < T_; ++j) dg = fmaf(to_f(D[i * LDD + j]), rs[j - i + T_ - 1], dg);
    dgate[((size_t)b * H + h) * T_ + i] = dg;
  }
  for (int q = threadIdx.x; q < 2 * T_ - 1; q += blockDim.x) {    // drb[r] += sum_i gate[i] dS[i][i + r]
    const int r = q - (T_ - 1);
    const int ilo = r < 0 ? -r : 0, ihi = r > 0 ? T_ - 1 - r : T_ - 1;
    float a = 0.f;
    for (int i = ilo; i <= ihi; ++i) a = fmaf(gs[i], to_f(D[i * LDD + i + r]), a);
    if (a != 0.f) atomicAdd(&drb[(size_t)h * (2 * RB - 1) + r + RB - 1], a);
  }
}
extern "C" int lidk_wavlm_attn_bias_grads(const void* ds, const float* gate, const float* rb, float* dgate, float* drb, int B, int T_,
                                          int H, int RB, int ldp, void* stream) {
  if (!ds || !gate || !rb || !dgate || !drb || B <= 0 || T_ <= 0 || H <= 0 || RB < T_ || ldp < T_) return LIDK_ERR_ARG;
  if (ldp & 7) return LIDK_ERR_ARG;
  const size_t lds = (((size_t)T_ * (ldp + 8) * 2) + 15) / 16 * 16 + (size_t)(3 * T_ - 1) * 4;
  if (lds > 160 * 1024) return LIDK_ERR_UNSUPPORTED;
  (void)hipFuncSetAttribute((const void*)wavlm_attn_bias_grads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  wavlm_attn_bias_grads_kernel<<<B * H, 256, lds, as_stream(stream)>>>((const bf16*)ds, gate, rb, dgate, drb, T_, H, RB, ldp);
  return launch_status();
}

// ------------------------------------------------------------------------------------ gate backward
// dgate [B][H][T] -> dx[m][h*dh + d] += sum_q du[q] wg[q][d] ; dwg [8][dh], dbg [8], dgrep_a [H] accumulated.
// One wave per (m, h) row, lane = channel d (dh <= 64): the 8 projections collapse to two (rows 0-3 feed gate a, rows 4-7 gate b,
// and du is the same within each group), so a lane carries two weight sums and two dW accumulators; per workgroup the four
// waves' accumulators meet in LDS and leave through one atomic per element.
__global__ void __launch_bounds__(256)
wavlm_gate_bwd_kernel(const float* __restrict__ x, const float* __restrict__ wg, const float* __restrict__ bg,
                      const float* __restrict__ grep_a, const float* __restrict__ dgate, float* __restrict__ dx,
                      float* __restrict__ dwg, float* __restrict__ dbg, float* __restrict__ dgrep_a, int B, int T_, int H, int dh) {
  extern __shared__ float red[];                       // [4][2][64] dW partials | [2] db | [H] dgrep_a
  float* red_a = red + 4 * 2 * 64 + 2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < H + 2; i += blockDim.x) red[4 * 2 * 64 + i] = 0.f;
  __syncthreads();
  float wsa = 0.f, wsb = 0.f;
  if (lane < dh) {
    wsa = wg[lane] + wg[dh + lane] + wg[2 * dh + lane] + wg[3 * dh + lane];
    wsb = wg[4 * dh + lane] + wg[5 * dh + lane] + wg[6 * dh + lane] + wg[7 * dh + lane];
  }
  const float ba = bg[0] + bg[1] + bg[2] + bg[3], bb = bg[4] + bg[5] + bg[6] + bg[7];
  float acc_wa = 0.f, acc_wb = 0.f, acc_ba = 0.f, acc_bb = 0.f;
  const long n = (long)B * T_ * H;
  for (long i = (long)blockIdx.x * 4 + wv; i < n; i += (long)gridDim.x * 4) {
    const int h = (int)(i % H); const long m = i / H;
    const long b = m / T_, t = m % T_;
    const long off = m * (long)H * dh + (long)h * dh + lane;
    const float v = lane < dh ? x[off] : 0.f;
    const float ua = wave_sum(v * wsa) + ba, ub = wave_sum(v * wsb) + bb;
    const float ga = 1.f / (1.f + __expf(-ua)), gb = 1.f / (1.f + __expf(-ub));
    const float a = grep_a[h], dgt = dgate[(b * H + h) * T_ + t];
    const float dsa = dgt * (gb * a - 1.f) * ga * (1.f - ga), dsb = dgt * ga * a * gb * (1.f - gb);
    if (lane < dh) dx[off] += dsa * wsa + dsb * wsb;
    acc_wa = fmaf(dsa, v, acc_wa); acc_wb = fmaf(dsb, v, acc_wb);
    acc_ba += dsa; acc_bb += dsb;
    if (lane == 0) atomicAdd(&red_a[h], dgt * ga * gb);
  }
  red[(wv * 2 + 0) * 64 + lane] = acc_wa;
  red[(wv * 2 + 1) * 64 + lane] = acc_wb;
  if (lane == 0) { atomicAdd(&red[4 * 2 * 64 + 0], acc_ba); atomicAdd(&red[4 * 2 * 64 + 1], acc_bb); }
  __syncthreads();
  if (threadIdx.x < 128) {                             // thread = (group g, channel d): rows 4g .. 4g+3 of dwg share the value
    const int g = threadIdx.x >> 6, d = threadIdx.x & 63;
    const float v = red[(0 * 2 + g) * 64 + d] + red[(1 * 2 + g) * 64 + d] + red[(2 * 2 + g) * 64 + d] + red[(3 * 2 + g) * 64 + d];
    if (d < dh && v != 0.f)
      for (int q = 0; q < 4; ++q) atomicAdd(&dwg[(4 * g + q) * dh + d], v);
  } else if (threadIdx.x < 136) {
    const int q = threadIdx.x - 128;
    const float v = red[4 * 2 * 64 + (q >> 2)];
    if (v != 0.f) atomicAdd(&dbg[q], v);
  }
  for (int i = threadIdx.x; i < H; i += blockDim.x)
    if (red_a[i] != 0.f) atomicAdd(&dgrep_a[i], red_a[i]);
}
extern "C" int lidk_wavlm_gate_bwd(const float* x, const float* wg, const float* bg, const float* grep_a, const float* dgate, float* dx,
                                   float* dwg, float* dbg, float* dgrep_a, int B, int T_, int H, int dh, void* stream) {
  if (!x || !wg || !bg || !grep_a || !dgate || !dx || !dwg || !dbg || !dgrep_a || B <= 0 || T_ <= 0 || H <= 0 || dh <= 0 || dh > 64)
    return LIDK_ERR_ARG;
  const long n = (long)B * T_ * H;
  int blocks = (int)((n + 63) / 64); if (blocks > 512) blocks = 512;         // >= 16 rows per wave
  wavlm_gate_bwd_kernel<<<blocks, 256, (size_t)(4 * 2 * 64 + 2 + H) * 4, as_stream(stream)>>>(x, wg, bg, grep_a, dgate, dx, dwg, dbg,
                                                                                              dgrep_a, B, T_, H, dh);
  return launch_status();
}

// ------------------------------------------------------------------------------------ positional conv backward operand
// dpc [B*Pp + slack][C] bf16: row (b, t) = dy[b][t] * gelu'(pre[b*Pp + t]) for t < T, zero elsewhere (pitch padding, slack).
// dpg (optional) [G][rows_total][C/G] bf16: the same values group-major at row b*Pp + goff + t - the operand of the DATA gradient
// (a strided-view GEMM against the flipped kernel); its other rows are never written and must be zero from allocation.
__global__ void wavlm_posconv_dprep_kernel(const float* __restrict__ dy, const bf16* __restrict__ pre, bf16* __restrict__ dpc,
                                           bf16* __restrict__ dpg, int B, int T_, int Pp, int C, int G, int goff, long rows_total) {
  const long n = rows_total * C;
  const int cg = C / G;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long row = i / C;
    float v = 0.f;
    if (row < (long)B * Pp) {
      const long b = row / Pp, t = row % Pp;
      if (t < T_) {
        v = dy[(b * T_ + t) * C + c] * gelu_grad_(to_f(pre[row * C + c]));
        if (dpg) { const int g = c / cg; dpg[((size_t)g * rows_total + row + goff) * cg + (c - g * cg)] = (bf16)v; }
      }
    }
    dpc[i] = (bf16)v;
  }
}
extern "C" int lidk_wavlm_posconv_dprep(const float* dy, const void* pre, void* dpc, void* dpg, int B, int T_, int Pp, int C, int G,
                                        int goff, long rows_total, void* stream) {
  if (!dy || !pre || !dpc || B <= 0 || T_ <= 0 || Pp < T_ || C <= 0 || rows_total < (long)B * Pp) return LIDK_ERR_ARG;
  if (dpg && (G <= 0 || C % G || goff < 0 || goff + T_ > Pp)) return LIDK_ERR_ARG;
  const long n = rows_total * C;
  int blocks = (int)((n + 255) / 256); if (blocks > 16384) blocks = 16384;
  wavlm_posconv_dprep_kernel<<<blocks, 256, 0, as_stream(stream)>>>(dy, (const bf16*)pre, (bf16*)dpc, (bf16*)dpg, B, T_, Pp, C,
                                                                    G > 0 ? G : 1, goff, rows_total);
  return launch_status();
}

// =====================================================================================================================
// wav2vec2 pieces (lid/s3prl_updream/wav2vec/wav2vec2.py, lid/s3prl_updream/interfaces.py).
// =====================================================================================================================
// x[b][t][:] = 0 for t >= klen[b]: TransformerEncoder.extract_features zeroes the padded frames before the positional
// convolution (wav2vec2.py:906-907 index_put(x, padding_mask, 0)); the same launch zeroes their gradient rows in backward.
__global__ void zero_padded_rows_kernel(float* __restrict__ x, const int* __restrict__ klen, int T_, int C, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int b = (int)(row / T_), t = (int)(row % T_);
    if (t >= klen[b]) x[i] = 0.f;
  }
}
extern "C" int lidk_zero_padded_rows(float* x, const int* klen, int B, int T_, int C, void* stream) {
  if (!x || !klen || B <= 0 || T_ <= 0 || C <= 0) return LIDK_ERR_ARG;
  const long n = (long)B * T_ * C;
  int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
  zero_padded_rows_kernel<<<blocks, 256, 0, as_stream(stream)>>>(x, klen, T_, C, n);
  return launch_status();
}

// s3prl Featurizer._weighted_sum (interfaces.py:227-252): feature = sum_l softmax(weights)[l] * hidden_state_l over the L + 1
// hidden states (the input of every transformer layer and the encoder output).  One launch per state:
//   fwd : feat = (l == 0 ? 0 : feat) + sm[l] * h_l
//   bwd : dots[l] = <dfeat, h_l> ; dh_l += sm[l] * dfeat (axpy) ; dw[l] += sm[l] * (dots[l] - sum_k sm[k] dots[k])
#define HM_MAX 64
__device__ __forceinline__ float hm_softmax_at(const float* __restrict__ w, int n_states, int l) {
  float mx = -INFINITY, sum = 0.f;
  for (int k = 0; k < n_states; ++k) mx = fmaxf(mx, w[k]);
  for (int k = 0; k < n_states; ++k) sum += __expf(w[k] - mx);
  return __expf(w[l] - mx) / sum;
}
__global__ void hidden_mix_axpy_kernel(const float* __restrict__ h, const float* __restrict__ w, int n_states, int l,
                                       float* __restrict__ out, long n, int overwrite) {
  const float s = hm_softmax_at(w, n_states, l);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = (overwrite ? 0.f : out[i]) + s * h[i];
}
extern "C" int lidk_hidden_mix_axpy(const float* h, const float* w, int n_states, int l, float* out, long n, int overwrite,
                                    void* stream) {
  if (!h || !w || !out || n_states <= 0 || n_states > HM_MAX || l < 0 || l >= n_states || n <= 0) return LIDK_ERR_ARG;
  int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
  hidden_mix_axpy_kernel<<<blocks, 256, 0, as_stream(stream)>>>(h, w, n_states, l, out, n, overwrite);
  return launch_status();
}
__global__ void hidden_mix_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dot, long n) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc = fmaf(a[i], b[i], acc);
  acc = wave_sum(acc);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dot, red[0] + red[1] + red[2] + red[3]);
}
extern "C" int lidk_hidden_mix_dot(const float* a, const float* b, float* dot, long n, void* stream) {
  if (!a || !b || !dot || n <= 0) return LIDK_ERR_ARG;
  int blocks = (int)((n + 255) / 256); if (blocks > 1024) blocks = 1024;
  hidden_mix_dot_kernel<<<blocks, 256, 0, as_stream(stream)>>>(a, b, dot, n);
  return launch_status();
}
__global__ void hidden_mix_wgrad_kernel(const float* __restrict__ w, const float* __restrict__ dots, float* __restrict__ dw,
                                        int n_states) {
  const int l = threadIdx.x;
  if (l >= n_states) return;
  float mean = 0.f;
  for (int k = 0; k < n_states; ++k) mean = fmaf(hm_softmax_at(w, n_states, k), dots[k], mean);
  dw[l] += hm_softmax_at(w, n_states, l) * (dots[l] - mean);
}
extern "C" int lidk_hidden_mix_wgrad(const float* w, const float* dots, float* dw, int n_states, void* stream) {
  if (!w || !dots || !dw || n_states <= 0 || n_states > HM_MAX) return LIDK_ERR_ARG;
  hidden_mix_wgrad_kernel<<<1, HM_MAX, 0, as_stream(stream)>>>(w, dots, dw, n_states);
  return launch_status();
}

// =====================================================================================================================
// Backward of the convolutional feature extractor (lid/wavlm/WavLM.py:409-531; un-frozen by
// lid/WavLMMutiLangModel.py:86-94 after freeze_encoder_epoch).  Layers 1-6 mirror the forward's strided-view GEMMs:
//   dpre_l = dY_l * gelu'(pre_l)            (pre_l kept by the forward's GELU epilogue, bf16)
//   dW_l  += dpre_l^T . A_l                 TN GEMM on the SAME strided view of layer l-1's output (lidk_gemm_tn, ldy < N2)
//   dcol_l = dpre_l . W_l                   NT GEMM: [rows][kW*C], the gradient of every window position
//   dY_{l-1}[2u + kk] = sum over windows    col2im below (k2 s2: windows do not overlap, k3 s2: rows 2u get two terms)
// and layer 0 (k10 s5 on the raw waveform + per-channel GroupNorm over time + GELU) is recomputed from the waveform as in the
// forward: one pass for the GroupNorm backward sums, one for dW0.
// =====================================================================================================================
// dprev [B*Pprev][C] bf16, Pprev = 2 * P: row (b, t) = (sum of the window terms of dcol that touch input row t) * gelu'(pre[b][t])
// for t < Tprev, zero behind (pitch padding rows must carry no gradient).  pre == NULL: no activation factor (layer 0's output:
// its GELU is handled with the GroupNorm in lidk_wavlm_conv0_bwd).  dcol [B*P][kW*C] bf16.
__global__ void wavlm_conv_col2im_kernel(const bf16* __restrict__ dcol, const bf16* __restrict__ pre, bf16* __restrict__ dprev,
                                         int B, int P, int T_, int Tprev, int kW, int C) {
  const int Pprev = 2 * P;
  const long n8 = (long)B * Pprev * C / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 8;
    const int c = (int)(e % C);
    const long row = e / C;
    const int b = (int)(row / Pprev), t = (int)(row % Pprev);
    f32x8_t v;
    v.lo = make_float4(0.f, 0.f, 0.f, 0.f); v.hi = v.lo;
    if (t < Tprev) {
      const int u = t >> 1, kk = t & 1;
      if (u < T_) v = load8(dcol + ((size_t)b * P + u) * kW * C + (size_t)kk * C + c);
      if (kW == 3 && kk == 0 && u >= 1 && u - 1 < T_) {
        const f32x8_t w = load8(dcol + ((size_t)b * P + u - 1) * kW * C + (size_t)2 * C + c);
        v.lo.x += w.lo.x; v.lo.y += w.lo.y; v.lo.z += w.lo.z; v.lo.w += w.lo.w;
        v.hi.x += w.hi.x; v.hi.y += w.hi.y; v.hi.z += w.hi.z; v.hi.w += w.hi.w;
      }
      if (pre) {
        const f32x8_t p = load8(pre + (size_t)row * C + c);
        v.lo.x *= gelu_grad_(p.lo.x); v.lo.y *= gelu_grad_(p.lo.y); v.lo.z *= gelu_grad_(p.lo.z); v.lo.w *= gelu_grad_(p.lo.w);
        v.hi.x *= gelu_grad_(p.hi.x); v.hi.y *= gelu_grad_(p.hi.y); v.hi.z *= gelu_grad_(p.hi.z); v.hi.w *= gelu_grad_(p.hi.w);
      }
    }
    store8(dprev + e, v);
  }
}
extern "C" int lidk_wavlm_conv_col2im(const void* dcol, const void* pre, void* dprev, int B, int P, int T_, int Tprev, int kW,
                                      int C, void* stream) {
  if (!dcol || !dprev || B <= 0 || P <= 0 || T_ <= 0 || T_ > P || Tprev <= 0 || Tprev > 2 * P || (kW != 2 && kW != 3) || C <= 0 || (C & 7))
    return LIDK_ERR_ARG;
  const long n8 = (long)B * 2 * P * C / 8;
  int blocks = (int)((n8 + 255) / 256); if (blocks > 16384) blocks = 16384;
  wavlm_conv_col2im_kernel<<<blocks, 256, 0, as_stream(stream)>>>((const bf16*)dcol, (const bf16*)pre, (bf16*)dprev, B, P, T_, Tprev, kW, C);
  return launch_status();
}

// dsrc [B*T][C] f32 (gradient at the extractor's output, channel-last without pitch) -> dpre [B*P][C] bf16 =
// dsrc * gelu'(pre) for t < T, zero in the pitch padding rows: the last conv layer's operand.
__global__ void wavlm_conv_dlast_kernel(const float* __restrict__ dsrc, const bf16* __restrict__ pre, bf16* __restrict__ dpre, int B,
                                        int T_, int P, int C) {
  const long n = (long)B * P * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long row = i / C;
    const int b = (int)(row / P), t = (int)(row % P);
    float v = 0.f;
    if (t < T_) v = dsrc[((size_t)b * T_ + t) * C + c] * (pre ? gelu_grad_((float)pre[i]) : 1.0f);
    dpre[i] = (bf16)v;
  }
}
extern "C" int lidk_wavlm_conv_dlast(const float* dsrc, const void* pre, void* dpre, int B, int T_, int P, int C, void* stream) {
  if (!dsrc || !dpre || B <= 0 || T_ <= 0 || P < T_ || C <= 0) return LIDK_ERR_ARG;
  const long n = (long)B * P * C;
  int blocks = (int)((n + 255) / 256); if (blocks > 16384) blocks = 16384;
  wavlm_conv_dlast_kernel<<<blocks, 256, 0, as_stream(stream)>>>(dsrc, (const bf16*)pre, (bf16*)dpre, B, T_, P, C);
  return launch_status();
}

// Layer 0 backward.  dy0 [B*P0][C] bf16 = gradient at the layer's (post-GELU) output.  stats [B][C][2] = (mean, rstd) of the
// forward (lidk_wavlm_conv0's workspace tail).  Pass 1: per (utterance, channel) S1 = sum_t dz, S2 = sum_t dz * yhat with
// z = yhat * gamma + beta, dz = dy0 * gelu'(z); pass 2: dy = rstd * gamma * (dz - S1/T - yhat * S2/T) and
// dW0[c][k] += sum_t dy * wav[5 t + k]; dgamma[c] += S2, dbeta[c] += S1.  (lid/wavlm/WavLM.py:433-470: conv, GroupNorm, GELU.)
#define W0B_T 1024       // time steps per workgroup in the backward passes
__global__ void __launch_bounds__(256)
wavlm_conv0_bwd_sums_kernel(const float* __restrict__ wav, int L, const float* __restrict__ w, const float* __restrict__ stats,
                            const float* __restrict__ gamma, const float* __restrict__ beta, const bf16* __restrict__ dy0,
                            float* __restrict__ sums, int T0, int P0, int C) {
  __shared__ float xs[W0_TC * W0_S + W0_K];
  const int b = blockIdx.y, tb = blockIdx.x * W0B_T;
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  float wk[2][W0_K];
  for (int q = 0; q < 2; ++q) {
    const int c = threadIdx.x + 256 * q;
    for (int k = 0; k < W0_K; ++k) wk[q][k] = c < C ? w[c * W0_K + k] : 0.f;
  }
  for (int t0 = tb; t0 < min(T0, tb + W0B_T); t0 += W0_TC) {
    const int nt = min(W0_TC, T0 - t0);
    __syncthreads();
    const float* x = wav + (size_t)b * L + (size_t)t0 * W0_S;
    for (int i = threadIdx.x; i < nt * W0_S + W0_K - W0_S; i += 256) xs[i] = x[i];
    __syncthreads();
    for (int q = 0; q < 2; ++q) {
      const int c = threadIdx.x + 256 * q;
      if (c >= C) continue;
      const float mu = stats[((size_t)b * C + c) * 2], rs = stats[((size_t)b * C + c) * 2 + 1], g = gamma[c], be = beta[c];
      for (int t = 0; t < nt; ++t) {
        float y = 0.f;
#pragma unroll
        for (int k = 0; k < W0_K; ++k) y = fmaf(wk[q][k], xs[t * W0_S + k], y);
        const float yh = (y - mu) * rs;
        const float dz = (float)dy0[((size_t)b * P0 + t0 + t) * C + c] * gelu_grad_(yh * g + be);
        s1[q] += dz; s2[q] = fmaf(dz, yh, s2[q]);
      }
    }
  }
  for (int q = 0; q < 2; ++q) {
    const int c = threadIdx.x + 256 * q;
    if (c < C) { atomicAdd(&sums[((size_t)b * C + c) * 2], s1[q]); atomicAdd(&sums[((size_t)b * C + c) * 2 + 1], s2[q]); }
  }
}
__global__ void __launch_bounds__(256)
wavlm_conv0_bwd_apply_kernel(const float* __restrict__ wav, int L, const float* __restrict__ w, const float* __restrict__ stats,
                             const float* __restrict__ gamma, const float* __restrict__ beta, const bf16* __restrict__ dy0,
                             const float* __restrict__ sums, float* __restrict__ dw, float* __restrict__ dgamma,
                             float* __restrict__ dbeta, int T0, int P0, int C) {
  __shared__ float xs[W0_TC * W0_S + W0_K];
  const int b = blockIdx.y, tb = blockIdx.x * W0B_T;
  float acc[2][W0_K], wk[2][W0_K];
  for (int q = 0; q < 2; ++q) {
    const int c = threadIdx.x + 256 * q;
    for (int k = 0; k < W0_K; ++k) { wk[q][k] = c < C ? w[c * W0_K + k] : 0.f; acc[q][k] = 0.f; }
  }
  const float invT = 1.0f / (float)T0;
  for (int t0 = tb; t0 < min(T0, tb + W0B_T); t0 += W0_TC) {
    const int nt = min(W0_TC, T0 - t0);
    __syncthreads();
    const float* x = wav + (size_t)b * L + (size_t)t0 * W0_S;
    for (int i = threadIdx.x; i < nt * W0_S + W0_K - W0_S; i += 256) xs[i] = x[i];
    __syncthreads();
    for (int q = 0; q < 2; ++q) {
      const int c = threadIdx.x + 256 * q;
      if (c >= C) continue;
      const float mu = stats[((size_t)b * C + c) * 2], rs = stats[((size_t)b * C + c) * 2 + 1], g = gamma[c], be = beta[c];
      const float m1 = sums[((size_t)b * C + c) * 2] * invT, m2 = sums[((size_t)b * C + c) * 2 + 1] * invT;
      for (int t = 0; t < nt; ++t) {
        float y = 0.f;
#pragma unroll
        for (int k = 0; k < W0_K; ++k) y = fmaf(wk[q][k], xs[t * W0_S + k], y);
        const float yh = (y - mu) * rs;
        const float dz = (float)dy0[((size_t)b * P0 + t0 + t) * C + c] * gelu_grad_(yh * g + be);
        const float dy = rs * g * (dz - m1 - yh * m2);
#pragma unroll
        for (int k = 0; k < W0_K; ++k) acc[q][k] = fmaf(dy, xs[t * W0_S + k], acc[q][k]);
      }
    }
  }
  for (int q = 0; q < 2; ++q) {
    const int c = threadIdx.x + 256 * q;
    if (c >= C) continue;
    for (int k = 0; k < W0_K; ++k) atomicAdd(&dw[c * W0_K + k], acc[q][k]);
    if (blockIdx.x == 0) {                               // one workgroup per utterance adds the affine parameters' share
      atomicAdd(&dgamma[c], sums[((size_t)b * C + c) * 2 + 1]);
      atomicAdd(&dbeta[c], sums[((size_t)b * C + c) * 2]);
    }
  }
}
// sums: scratch [B][C][2] f32 (zeroed here).  dw [C][10], dgamma [C], dbeta [C] are ACCUMULATED.  C <= 512.
extern "C" int lidk_wavlm_conv0_bwd(const float* wav, int B, int L, const float* w, const float* gamma, const float* beta,
                                    const float* stats, const void* dy0, float* sums, float* dw, float* dgamma, float* dbeta,
                                    int T0, int P0, int C, void* stream) {
  if (!wav || !w || !gamma || !beta || !stats || !dy0 || !sums || !dw || !dgamma || !dbeta || B <= 0 || C <= 0 || C > 512 || T0 <= 0 ||
      P0 < T0)
    return LIDK_ERR_ARG;
  if ((long)(T0 - 1) * W0_S + W0_K > L) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(sums, 0, (size_t)B * C * 2 * sizeof(float), s) != hipSuccess) return LIDK_ERR_LAUNCH;
  const dim3 grid(cdiv(T0, W0B_T), B);
  wavlm_conv0_bwd_sums_kernel<<<grid, 256, 0, s>>>(wav, L, w, stats, gamma, beta, (const bf16*)dy0, sums, T0, P0, C);
  wavlm_conv0_bwd_apply_kernel<<<grid, 256, 0, s>>>(wav, L, w, stats, gamma, beta, (const bf16*)dy0, sums, dw, dgamma, dbeta, T0, P0, C);
  return launch_status();
}

// ------------------------------------------------------------------------------------ operand refresh after an optimizer step
// The backbones keep, per Linear, the f32 parameter (optimizer's), its bf16 copy W [N][K] (forward: out = x . W^T) and the bf16
// transpose W^T [K][N] (data gradient as an NT GEMM).  Refreshing those with torch copies cost ~310 us per XLS-R layer (14 copy /
// transposing-copy launches, 7.5 of a 98 ms fine-tune step).  Here one launch per layer: every record is a f32 matrix read ONCE
// through 64 x 64 LDS tiles and written as bf16 (dst), transposed bf16 (dstT) and / or f32 (dst32: the packed q|k|v bias).
struct CtDesc {
  const float* src; bf16* dst; bf16* dstT; float* dst32;
  int R, C, lds, ldd, ldt, ld32, tiles_c, tile0;          // rows, columns, leading dimensions (elements), column tiles, first tile
};
extern "C" int lidk_cast_transpose_desc_bytes(void) { return (int)sizeof(CtDesc); }

__global__ void __launch_bounds__(256) cast_transpose_grouped_kernel(const CtDesc* __restrict__ descs, int n) {
  __shared__ float tile[64][65];
  const int b = blockIdx.x, tid = threadIdx.x;
  int d = 0;
  while (d + 1 < n && descs[d + 1].tile0 <= b) ++d;         // block-uniform; n is a handful
  const CtDesc D = descs[d];
  const int t = b - D.tile0, r0 = (t / D.tiles_c) * 64, c0 = (t % D.tiles_c) * 64;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int rl = (tid >> 4) + 16 * it, cl = 4 * (tid & 15), r = r0 + rl, c = c0 + cl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < D.R && c < D.C) {                               // C % 4 == 0 (host check): whole vectors in or out
      v = load4(D.src + (size_t)r * D.lds + c);
      if (D.dst) store4(D.dst + (size_t)r * D.ldd + c, v);
      if (D.dst32) store4(D.dst32 + (size_t)r * D.ld32 + c, v);
    }
    tile[rl][cl] = v.x; tile[rl][cl + 1] = v.y; tile[rl][cl + 2] = v.z; tile[rl][cl + 3] = v.w;
  }
  if (!D.dstT) return;                                      // block-uniform
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int cl = (tid >> 4) + 16 * it, rl = 4 * (tid & 15), c = c0 + cl, r = r0 + rl;
    if (c < D.C && r < D.R)                                 // R % 4 == 0 for records with a transpose (host check)
      store4(D.dstT + (size_t)c * D.ldt + r, make_float4(tile[rl][cl], tile[rl + 1][cl], tile[rl + 2][cl], tile[rl + 3][cl]));
  }
}

extern "C" int lidk_cast_transpose_grouped(const void* descs, int n, int total_tiles, void* stream) {
  if (!descs || n <= 0 || total_tiles <= 0) return LIDK_ERR_ARG;
  cast_transpose_grouped_kernel<<<total_tiles, 256, 0, as_stream(stream)>>>((const CtDesc*)descs, n);
  return launch_status();
}
