// Element-wise helpers, column reductions, transpose and LayerNorm (HBM-bound kernels).
// One wave (64 lanes) per row for row-wise ops, 16-byte accesses, wave-shuffle reductions.
#include "common.h"

// ------------------------------------------------------------------------------------ scale_cast
template <typename TI, typename TO>
__global__ void scale_cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long n, float scale) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    float4 v = load4(x + i);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    store4(y + i, v);
  }
  if (i < n) for (long j = i; j < n; ++j) y[j] = from_f<TO>(to_f(x[j]) * scale);
}

template <typename TI, typename TO>
static int scale_cast_launch(const void* x, void* y, long n, float scale, hipStream_t s) {
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  scale_cast_kernel<TI, TO><<<blocks, 256, 0, s>>>((const TI*)x, (TO*)y, n, scale);
  return launch_status();
}

extern "C" int lidk_scale_cast(const void* x, int xd, void* y, int yd, long n, float scale, void* stream) {
  if (!x || !y || n < 0) return LIDK_ERR_ARG;
  if (n == 0) return LIDK_OK;
  hipStream_t s = as_stream(stream);
  if (xd == LIDK_F32 && yd == LIDK_F32) return scale_cast_launch<float, float>(x, y, n, scale, s);
  if (xd == LIDK_F32 && yd == LIDK_BF16) return scale_cast_launch<float, bf16>(x, y, n, scale, s);
  if (xd == LIDK_BF16 && yd == LIDK_F32) return scale_cast_launch<bf16, float>(x, y, n, scale, s);
  if (xd == LIDK_BF16 && yd == LIDK_BF16) return scale_cast_launch<bf16, bf16>(x, y, n, scale, s);
  return LIDK_ERR_ARG;
}

// strided 2-D variant: y[m][n] = scale * x[m][n] for n < N (pad columns of y are left untouched)
template <typename TI, typename TO>
__global__ void scale_cast_2d_kernel(const TI* __restrict__ x, int ldx, TO* __restrict__ y, int ldy, int M, int N, float scale) {
  long n_el = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (long)gridDim.x * blockDim.x) {
    long m = i / N; int n = (int)(i - m * N);
    y[m * ldy + n] = from_f<TO>(to_f(x[m * ldx + n]) * scale);
  }
}
template <typename TI, typename TO>
static int scale_cast_2d_launch(const void* x, int ldx, void* y, int ldy, int M, int N, float scale, hipStream_t s) {
  long n_el = (long)M * N;
  int blocks = (int)((n_el + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  scale_cast_2d_kernel<TI, TO><<<blocks, 256, 0, s>>>((const TI*)x, ldx, (TO*)y, ldy, M, N, scale);
  return launch_status();
}
extern "C" int lidk_scale_cast_2d(const void* x, int ldx, int xd, void* y, int ldy, int yd, int M, int N, float scale,
                                  void* stream) {
  if (!x || !y || M <= 0 || N <= 0 || ldx < N || ldy < N) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  if (xd == LIDK_F32 && yd == LIDK_F32) return scale_cast_2d_launch<float, float>(x, ldx, y, ldy, M, N, scale, s);
  if (xd == LIDK_F32 && yd == LIDK_BF16) return scale_cast_2d_launch<float, bf16>(x, ldx, y, ldy, M, N, scale, s);
  if (xd == LIDK_BF16 && yd == LIDK_F32) return scale_cast_2d_launch<bf16, float>(x, ldx, y, ldy, M, N, scale, s);
  if (xd == LIDK_BF16 && yd == LIDK_BF16) return scale_cast_2d_launch<bf16, bf16>(x, ldx, y, ldy, M, N, scale, s);
  return LIDK_ERR_ARG;
}

// ------------------------------------------------------------------------------------ dropout
// Counter-based generator uniform_from(seed, index): common.h
template <typename TI, typename TO>
__global__ void dropout_kernel(const TI* __restrict__ x, TO* __restrict__ y, const uint8_t* __restrict__ keep_in,
                               uint8_t* __restrict__ keep_out, long n, float p, float inv_keep, uint64_t seed) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    bool keep = keep_in ? (keep_in[i] != 0) : (uniform_from(seed, (uint64_t)i) >= p);
    if (keep_out) keep_out[i] = keep ? 1 : 0;
    y[i] = from_f<TO>(keep ? to_f(x[i]) * inv_keep : 0.0f);
  }
}

template <typename TI, typename TO>
static int dropout_launch(const void* x, void* y, const uint8_t* ki, uint8_t* ko, long n, float p, uint64_t seed,
                          hipStream_t s) {
  int blocks = (int)((n + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  dropout_kernel<TI, TO><<<blocks, 256, 0, s>>>((const TI*)x, (TO*)y, ki, ko, n, p, 1.0f / (1.0f - p), seed);
  return launch_status();
}

extern "C" int lidk_dropout(const void* x, int xd, void* y, int yd, const uint8_t* keep_in, uint8_t* keep_out, long n,
                            float p, uint64_t seed, void* stream) {
  if (!x || !y || n < 0 || p < 0.f || p >= 1.f) return LIDK_ERR_ARG;
  if (n == 0) return LIDK_OK;
  hipStream_t s = as_stream(stream);
  if (xd == LIDK_F32 && yd == LIDK_F32) return dropout_launch<float, float>(x, y, keep_in, keep_out, n, p, seed, s);
  if (xd == LIDK_F32 && yd == LIDK_BF16) return dropout_launch<float, bf16>(x, y, keep_in, keep_out, n, p, seed, s);
  if (xd == LIDK_BF16 && yd == LIDK_F32) return dropout_launch<bf16, float>(x, y, keep_in, keep_out, n, p, seed, s);
  if (xd == LIDK_BF16 && yd == LIDK_BF16) return dropout_launch<bf16, bf16>(x, y, keep_in, keep_out, n, p, seed, s);
  return LIDK_ERR_ARG;
}

// out = res + dropout(x): the transformer layers' dropout1 / dropout3 in front of the residual add (lid/wavlm/WavLM.py:745-771),
// same decisions as dropout_kernel for the same (seed, index).
__global__ void dropout_add_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ out,
                                   const uint8_t* __restrict__ keep_in, long n, float p, float inv_keep, uint64_t seed) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const bool keep = keep_in ? (keep_in[i] != 0) : (uniform_from(seed, (uint64_t)i) >= p);
    out[i] = res[i] + (keep ? x[i] * inv_keep : 0.0f);
  }
}
extern "C" int lidk_dropout_add(const float* x, const float* res, float* out, const uint8_t* keep_in, long n, float p,
                                uint64_t seed, void* stream) {
  if (!x || !res || !out || n < 0 || p < 0.f || p >= 1.f) return LIDK_ERR_ARG;
  if (n == 0) return LIDK_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  dropout_add_kernel<<<blocks, 256, 0, as_stream(stream)>>>(x, res, out, keep_in, n, p, 1.0f / (1.0f - p), seed);
  return launch_status();
}

// ------------------------------------------------------------------------------------ relu backward
template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dx[i] = to_f(y[i]) > 0.f ? dy[i] : from_f<T>(0.f);
}

extern "C" int lidk_relu_bwd(const void* dy, const void* y, void* dx, long n, int dtype, void* stream) {
  if (!dy || !y || !dx || n < 0) return LIDK_ERR_ARG;
  if (n == 0) return LIDK_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  LIDK_DISPATCH(dtype, relu_bwd_kernel<T><<<blocks, 256, 0, as_stream(stream)>>>((const T*)dy, (const T*)y, (T*)dx, n));
  return launch_status();
}

// ------------------------------------------------------------------------------------ column sums
// Stage 1: block b sums rows b, b+G, ... for every column (lanes over columns -> coalesced); stage 2 adds the
// G partial rows in fixed order (deterministic) into out[] with a scale.
template <typename T>
__global__ void colsum_partial_kernel(const T* __restrict__ x, int ldx, float* __restrict__ partial, int M, int N) {
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    float acc = 0.f;
    for (int m = blockIdx.x; m < M; m += gridDim.x) acc += to_f(x[(size_t)m * ldx + n]);
    partial[(size_t)blockIdx.x * N + n] = acc;
  }
}

// Second stage of every column reduction: out[c] (+)= scale * sum_p partial[p][c].  1024 threads = 16 waves: lane -> column,
// wave -> every 16th partial row (independent loads in flight), then a fixed-order LDS combine: deterministic.
// Columns c < split go to out0[c], the rest to out1[c - split] (LayerNorm: dgamma | dbeta).
template <typename ACC, typename TOUT, bool ACCUM>
__global__ void __launch_bounds__(1024)
colreduce_kernel(const float* __restrict__ partial, int nparts, int ncols, TOUT* __restrict__ out0, TOUT* __restrict__ out1,
                 int split, float scale, TOUT* __restrict__ dup = nullptr, double tail = 0.0) {
  // 16 columns per workgroup (32 workgroups for the 512 LayerNorm columns instead of 8): a wave reads 4 partial rows x 16
  // columns per instruction; fixed summation order -> deterministic
  __shared__ ACC red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane >> 4;
  const int c = blockIdx.x * 16 + (lane & 15);
  if (tail > 0.0 && blockIdx.x == 0 && threadIdx.x == 0) {       // the row count behind the sums (lidk_reduce_partials_f64)
    out0[ncols] = (TOUT)tail;
    if (dup) dup[ncols] = (TOUT)tail;
  }
  ACC acc = 0;
  if (c < ncols) {
#pragma unroll 4
    for (int p = w * 4 + sub; p < nparts; p += 64) acc += (ACC)partial[(size_t)p * ncols + c];
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w != 0) return;
  // second stage on the whole first wave: lane (sub, column) adds its quarter's 16 rows, the four quarters meet by two xor
  // shuffles (fixed order: deterministic).  (One lane per column adding all 64 values serially cost the f64 instantiation 68
  // bytes of scratch per lane under the 1024-thread register cap.)
  ACC s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += red[i][lane];
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if (lane < 16 && c < ncols) {
    TOUT* dst = (c < split) ? (out0 ? out0 + c : nullptr) : (out1 ? out1 + (c - split) : nullptr);
    if (dst) *dst = ACCUM ? (TOUT)(*dst + (TOUT)scale * (TOUT)s) : (TOUT)s;
    if (dup) dup[c] = (TOUT)s;
  }
}

extern "C" int lidk_colsum(const void* x, int ldx, int xd, float* out, float* partial, int M, int N, float scale,
                           void* stream) {
  if (!x || !out || !partial || M <= 0 || N <= 0) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  int G = M < LIDK_LN_PARTIAL_BLOCKS ? M : LIDK_LN_PARTIAL_BLOCKS;
  LIDK_DISPATCH(xd, colsum_partial_kernel<T><<<G, 256, 0, s>>>((const T*)x, ldx, partial, M, N));
  colreduce_kernel<float, float, true><<<cdiv(N, 16), 1024, 0, s>>>(partial, G, N, out, (float*)nullptr, N, scale);
  return launch_status();
}

extern "C" int lidk_reduce_partials_f64(const float* partial, int nparts, int ncols, double* out, double* out2, double tail,
                                        void* stream) {
  if (!partial || !out || nparts <= 0 || ncols <= 0) return LIDK_ERR_ARG;
  colreduce_kernel<double, double, false><<<cdiv(ncols, 16), 1024, 0, as_stream(stream)>>>(partial, nparts, ncols, out,
                                                                                          (double*)nullptr, ncols, 1.0f, out2,
                                                                                          tail);
  return launch_status();
}

// ------------------------------------------------------------------------------------ transpose
template <typename T>
__global__ void transpose_kernel(const T* __restrict__ in, int ldi, T* __restrict__ out, int ldo, int R, int C) {
  __shared__ T tile[64][65];
  int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 4 rows per pass
  for (int i = ty; i < 64; i += 4) {
    int r = r0 + i, c = c0 + tx;
    if (r < R && c < C) tile[i][tx] = in[(size_t)r * ldi + c];
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    int c = c0 + i, r = r0 + tx;
    if (r < R && c < C) out[(size_t)c * ldo + r] = tile[tx][i];
  }
}

extern "C" int lidk_transpose(const void* in, int ldi, void* out, int ldo, int R, int C, int dtype, void* stream) {
  if (!in || !out || R <= 0 || C <= 0 || ldi < C || ldo < R) return LIDK_ERR_ARG;
  dim3 grid(cdiv(C, 64), cdiv(R, 64));
  LIDK_DISPATCH(dtype, transpose_kernel<T><<<grid, 256, 0, as_stream(stream)>>>((const T*)in, ldi, (T*)out, ldo, R, C));
  return launch_status();
}

// ------------------------------------------------------------------------------------ LayerNorm forward
#define LN_MAX_VEC 4   // up to 4 float4 per lane -> C <= 1024
template <typename T>
__global__ void __launch_bounds__(256)
ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
              T* __restrict__ yT, float* __restrict__ y32, float* __restrict__ mean, float* __restrict__ rstd, int M,
              int C, float eps) {
  int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * C;
  float4 v[LN_MAX_VEC];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX_VEC; ++k) {
    int c = lane * 4 + k * 256;
    if (c < C) { v[k] = load4(xr + c); s += v[k].x + v[k].y + v[k].z + v[k].w; }
  }
  float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAX_VEC; ++k) {
    int c = lane * 4 + k * 256;
    if (c < C) {
      float a = v[k].x - mu, b = v[k].y - mu, d = v[k].z - mu, e = v[k].w - mu;
      q += a * a + b * b + d * d + e * e;
    }
  }
  float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#pragma unroll
  for (int k = 0; k < LN_MAX_VEC; ++k) {
    int c = lane * 4 + k * 256;
    if (c < C) {
      float4 g = load4(gamma + c), b = load4(beta + c), o;
      o.x = (v[k].x - mu) * rs * g.x + b.x; o.y = (v[k].y - mu) * rs * g.y + b.y;
      o.z = (v[k].z - mu) * rs * g.z + b.z; o.w = (v[k].w - mu) * rs * g.w + b.w;
      if (yT) store4(yT + (size_t)row * C + c, o);
      if (y32) store4(y32 + (size_t)row * C + c, o);
    }
  }
}

// C <= 256: two rows per wave, both rows' loads in flight together (half the waves, one scheduling round at the training shapes)
template <typename T>
__global__ void __launch_bounds__(256)
ln_fwd_c256_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                   T* __restrict__ yT, float* __restrict__ y32, float* __restrict__ mean, float* __restrict__ rstd, int M,
                   int C, float eps) {
  const int lane = threadIdx.x & 63, c = lane * 4;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
  if (row0 >= M) return;
  const bool has1 = row0 + 1 < M, ok = c < C;                 // has1 is wave-uniform
  const int row1 = has1 ? row0 + 1 : row0;
  float4 v0 = make_float4(0, 0, 0, 0), v1 = v0, g = v0, b = v0;
  if (ok) { v0 = load4(x + (size_t)row0 * C + c); v1 = load4(x + (size_t)row1 * C + c); g = load4(gamma + c); b = load4(beta + c); }
  float s0 = v0.x + v0.y + v0.z + v0.w, s1 = v1.x + v1.y + v1.z + v1.w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
  const float mu0 = s0 / (float)C, mu1 = s1 / (float)C;        // the same expressions as ln_fwd_kernel: bit-identical outputs
  float q0 = 0.f, q1 = 0.f;
  if (ok) {
    float a = v0.x - mu0, b2 = v0.y - mu0, d = v0.z - mu0, e = v0.w - mu0;
    q0 += a * a + b2 * b2 + d * d + e * e;
    a = v1.x - mu1; b2 = v1.y - mu1; d = v1.z - mu1; e = v1.w - mu1;
    q1 += a * a + b2 * b2 + d * d + e * e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { q0 += __shfl_xor(q0, o, 64); q1 += __shfl_xor(q1, o, 64); }
  const float rs0 = rsqrtf(q0 / (float)C + eps), rs1 = rsqrtf(q1 / (float)C + eps);
  if (lane == 0) {
    if (mean) { mean[row0] = mu0; if (has1) mean[row1] = mu1; }
    if (rstd) { rstd[row0] = rs0; if (has1) rstd[row1] = rs1; }
  }
  if (!ok) return;
  float4 o;
  o.x = (v0.x - mu0) * rs0 * g.x + b.x; o.y = (v0.y - mu0) * rs0 * g.y + b.y;
  o.z = (v0.z - mu0) * rs0 * g.z + b.z; o.w = (v0.w - mu0) * rs0 * g.w + b.w;
  if (yT) store4(yT + (size_t)row0 * C + c, o);
  if (y32) store4(y32 + (size_t)row0 * C + c, o);
  if (has1) {
    o.x = (v1.x - mu1) * rs1 * g.x + b.x; o.y = (v1.y - mu1) * rs1 * g.y + b.y;
    o.z = (v1.z - mu1) * rs1 * g.z + b.z; o.w = (v1.w - mu1) * rs1 * g.w + b.w;
    if (yT) store4(yT + (size_t)row1 * C + c, o);
    if (y32) store4(y32 + (size_t)row1 * C + c, o);
  }
}

extern "C" int lidk_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* yT, float* y32,
                                  float* mean, float* rstd, int M, int C, float eps, int dtype, void* stream) {
  if (!x || !gamma || !beta || (!yT && !y32) || M <= 0 || C <= 0 || (C & 3) || C > 256 * LN_MAX_VEC) return LIDK_ERR_ARG;
  static const int small = getenv("LIDK_LN_FWD_SMALL") ? atoi(getenv("LIDK_LN_FWD_SMALL")) : 1;
  if (small && C <= 256) {
    LIDK_DISPATCH(dtype, ln_fwd_c256_kernel<T><<<cdiv(M, 8), 256, 0, as_stream(stream)>>>(x, gamma, beta, (T*)yT, y32, mean,
                                                                                         rstd, M, C, eps));
    return launch_status();
  }
  LIDK_DISPATCH(dtype, ln_fwd_kernel<T><<<cdiv(M, 4), 256, 0, as_stream(stream)>>>(x, gamma, beta, (T*)yT, y32, mean,
                                                                                  rstd, M, C, eps));
  return launch_status();
}

// ------------------------------------------------------------------------------------ LayerNorm backward
// Each wave walks rows (grid-stride), keeps per-column dgamma/dbeta partials in registers, then the 4 waves of a
// block combine through LDS and write one partial row; partial_finalize adds them into the gradient arena.
template <typename T, typename TDY>
__global__ void __launch_bounds__(256)
ln_bwd_kernel(const TDY* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
              const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ dres,
              float* __restrict__ dx, T* __restrict__ dxT, float dxT_scale, float* __restrict__ partial, int M, int C) {
  __shared__ float red[4][2 * 256 * LN_MAX_VEC];   // [wave][2*C], worst case C=1024 -> 2048 floats per wave (32 KiB)
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 dg[LN_MAX_VEC], db[LN_MAX_VEC], gm[LN_MAX_VEC];
#pragma unroll
  for (int k = 0; k < LN_MAX_VEC; ++k) {
    dg[k] = make_float4(0, 0, 0, 0); db[k] = make_float4(0, 0, 0, 0);
    int c = lane * 4 + k * 256;
    gm[k] = c < C ? load4(gamma + c) : make_float4(0, 0, 0, 0);
  }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    float mu = mean[row], rs = rstd[row];
    float4 xh[LN_MAX_VEC], g[LN_MAX_VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAX_VEC; ++k) {
      int c = lane * 4 + k * 256;
      if (c < C) {
        float4 xv = load4(x + (size_t)row * C + c);
        float4 d = load4(dy + (size_t)row * C + c);
        xh[k].x = (xv.x - mu) * rs; xh[k].y = (xv.y - mu) * rs; xh[k].z = (xv.z - mu) * rs; xh[k].w = (xv.w - mu) * rs;
        g[k].x = d.x * gm[k].x; g[k].y = d.y * gm[k].y; g[k].z = d.z * gm[k].z; g[k].w = d.w * gm[k].w;
        s1 += g[k].x + g[k].y + g[k].z + g[k].w;
        s2 += g[k].x * xh[k].x + g[k].y * xh[k].y + g[k].z * xh[k].z + g[k].w * xh[k].w;
        dg[k].x += d.x * xh[k].x; dg[k].y += d.y * xh[k].y; dg[k].z += d.z * xh[k].z; dg[k].w += d.w * xh[k].w;
        db[k].x += d.x; db[k].y += d.y; db[k].z += d.z; db[k].w += d.w;
      }
    }
    float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int k = 0; k < LN_MAX_VEC; ++k) {
      int c = lane * 4 + k * 256;
      if (c < C) {
        float4 o;
        o.x = rs * (g[k].x - m1 - xh[k].x * m2); o.y = rs * (g[k].y - m1 - xh[k].y * m2);
        o.z = rs * (g[k].z - m1 - xh[k].z * m2); o.w = rs * (g[k].w - m1 - xh[k].w * m2);
        if (dres) { float4 r = load4(dres + (size_t)row * C + c); o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        if (dx) store4(dx + (size_t)row * C + c, o);
        if (dxT) {
          o.x *= dxT_scale; o.y *= dxT_scale; o.z *= dxT_scale; o.w *= dxT_scale;
          store4(dxT + (size_t)row * C + c, o);
        }
      }
    }
  }
  // combine the block's 4 waves: red[wave][0..C) = dgamma, red[wave][C..2C) = dbeta
#pragma unroll
  for (int k = 0; k < LN_MAX_VEC; ++k) {
    int c = lane * 4 + k * 256;
    if (c < C) { store4(&red[wave][c], dg[k]); store4(&red[wave][C + c], db[k]); }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256)
    partial[(size_t)blockIdx.x * 2 * C + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// C <= 256 (one float4 per lane): the same arithmetic with TWO rows of loads in flight per wave - with ~2.4 rows per wave at the
// training shapes the one-row loop is a chain of exposed memory latencies - and 8 KB of LDS instead of 32.
template <typename T, typename TDY>
__global__ void __launch_bounds__(256)
ln_bwd_c256_kernel(const TDY* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                   const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ dres,
                   float* __restrict__ dx, T* __restrict__ dxT, float dxT_scale, float* __restrict__ partial, int M, int C) {
  __shared__ float red[4][2 * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane * 4;
  const bool ok = c < C;
  float4 dg = make_float4(0, 0, 0, 0), db = dg;
  const float4 gm = ok ? load4(gamma + c) : make_float4(0, 0, 0, 0);
  const int stride = gridDim.x * 4;
  for (int row0 = blockIdx.x * 4 + wave; row0 < M; row0 += 2 * stride) {
    const int row1 = row0 + stride;
    const bool has1 = row1 < M;                                  // wave-uniform
    const int r1 = has1 ? row1 : row0;                           // clamped: the second set of loads is unconditional
    const float mu0 = mean[row0], rs0 = rstd[row0], mu1 = mean[r1], rs1 = rstd[r1];
    float4 xv0 = make_float4(0, 0, 0, 0), d0 = xv0, xv1 = xv0, d1 = xv0, q0 = xv0, q1 = xv0;
    if (ok) {
      xv0 = load4(x + (size_t)row0 * C + c); d0 = load4(dy + (size_t)row0 * C + c);
      xv1 = load4(x + (size_t)r1 * C + c);   d1 = load4(dy + (size_t)r1 * C + c);
      if (dres) { q0 = load4(dres + (size_t)row0 * C + c); q1 = load4(dres + (size_t)r1 * C + c); }
    }
    float4 xh0, g0, xh1, g1;
    xh0.x = (xv0.x - mu0) * rs0; xh0.y = (xv0.y - mu0) * rs0; xh0.z = (xv0.z - mu0) * rs0; xh0.w = (xv0.w - mu0) * rs0;
    xh1.x = (xv1.x - mu1) * rs1; xh1.y = (xv1.y - mu1) * rs1; xh1.z = (xv1.z - mu1) * rs1; xh1.w = (xv1.w - mu1) * rs1;
    g0.x = d0.x * gm.x; g0.y = d0.y * gm.y; g0.z = d0.z * gm.z; g0.w = d0.w * gm.w;
    g1.x = d1.x * gm.x; g1.y = d1.y * gm.y; g1.z = d1.z * gm.z; g1.w = d1.w * gm.w;
    float a0 = g0.x + g0.y + g0.z + g0.w, b0 = g0.x * xh0.x + g0.y * xh0.y + g0.z * xh0.z + g0.w * xh0.w;
    float a1 = g1.x + g1.y + g1.z + g1.w, b1 = g1.x * xh1.x + g1.y * xh1.y + g1.z * xh1.z + g1.w * xh1.w;
    if (!ok) { a0 = b0 = a1 = b1 = 0.f; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a0 += __shfl_xor(a0, o, 64); b0 += __shfl_xor(b0, o, 64); a1 += __shfl_xor(a1, o, 64); b1 += __shfl_xor(b1, o, 64);
    }
    a0 /= (float)C; b0 /= (float)C; a1 /= (float)C; b1 /= (float)C;      // as ln_bwd_kernel: bit-identical outputs
    if (ok) {
      dg.x += d0.x * xh0.x; dg.y += d0.y * xh0.y; dg.z += d0.z * xh0.z; dg.w += d0.w * xh0.w;
      db.x += d0.x; db.y += d0.y; db.z += d0.z; db.w += d0.w;
      float4 o;
      o.x = rs0 * (g0.x - a0 - xh0.x * b0) + q0.x; o.y = rs0 * (g0.y - a0 - xh0.y * b0) + q0.y;
      o.z = rs0 * (g0.z - a0 - xh0.z * b0) + q0.z; o.w = rs0 * (g0.w - a0 - xh0.w * b0) + q0.w;
      if (dx) store4(dx + (size_t)row0 * C + c, o);
      if (dxT) { o.x *= dxT_scale; o.y *= dxT_scale; o.z *= dxT_scale; o.w *= dxT_scale; store4(dxT + (size_t)row0 * C + c, o); }
      if (has1) {
        dg.x += d1.x * xh1.x; dg.y += d1.y * xh1.y; dg.z += d1.z * xh1.z; dg.w += d1.w * xh1.w;
        db.x += d1.x; db.y += d1.y; db.z += d1.z; db.w += d1.w;
        o.x = rs1 * (g1.x - a1 - xh1.x * b1) + q1.x; o.y = rs1 * (g1.y - a1 - xh1.y * b1) + q1.y;
        o.z = rs1 * (g1.z - a1 - xh1.z * b1) + q1.z; o.w = rs1 * (g1.w - a1 - xh1.w * b1) + q1.w;
        if (dx) store4(dx + (size_t)row1 * C + c, o);
        if (dxT) { o.x *= dxT_scale; o.y *= dxT_scale; o.z *= dxT_scale; o.w *= dxT_scale; store4(dxT + (size_t)row1 * C + c, o); }
      }
    }
  }
  if (ok) { store4(&red[wave][c], dg); store4(&red[wave][C + c], db); }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256)
    partial[(size_t)blockIdx.x * 2 * C + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

extern "C" int lidk_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                                  const float* gamma, const float* dres, float* dx, void* dxT, float dxT_scale,
                                  float* dgamma, float* dbeta, float* partial, int M, int C, int dtype, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !partial || (!dx && !dxT) || M <= 0 || C <= 0 || (C & 3) ||
      C > 256 * LN_MAX_VEC)
    return LIDK_ERR_ARG;
  if (dy_dtype != LIDK_F32 && dy_dtype != dtype) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  int G = cdiv(M, 4) < LIDK_LN_BWD_BLOCKS ? cdiv(M, 4) : LIDK_LN_BWD_BLOCKS;   // one wave per row in flight: latency-bound otherwise
  static const int small = getenv("LIDK_LN_BWD_SMALL") ? atoi(getenv("LIDK_LN_BWD_SMALL")) : 1;
  if (small && C <= 256) {
    if (dy_dtype == LIDK_F32) {
      LIDK_DISPATCH(dtype, ln_bwd_c256_kernel<T, float><<<G, 256, 0, s>>>((const float*)dy, x, mean, rstd, gamma, dres, dx,
                                                                         (T*)dxT, dxT_scale, partial, M, C));
    } else {
      LIDK_DISPATCH(dtype, ln_bwd_c256_kernel<T, T><<<G, 256, 0, s>>>((const T*)dy, x, mean, rstd, gamma, dres, dx, (T*)dxT,
                                                                     dxT_scale, partial, M, C));
    }
  } else if (dy_dtype == LIDK_F32) {
    LIDK_DISPATCH(dtype, ln_bwd_kernel<T, float><<<G, 256, 0, s>>>((const float*)dy, x, mean, rstd, gamma, dres, dx,
                                                                  (T*)dxT, dxT_scale, partial, M, C));
  } else {
    LIDK_DISPATCH(dtype, ln_bwd_kernel<T, T><<<G, 256, 0, s>>>((const T*)dy, x, mean, rstd, gamma, dres, dx, (T*)dxT,
                                                              dxT_scale, partial, M, C));
  }
  if (dgamma || dbeta)      // both NULL: the caller finishes the parameter gradients later with lidk_layernorm_param_grads
    colreduce_kernel<float, float, true><<<cdiv(2 * C, 16), 1024, 0, s>>>(partial, G, 2 * C, dgamma, dbeta, C, 1.0f);
  return launch_status();
}

// ------------------------------------------------------------------------------------ two LayerNorms in a row (C <= 256)
// Between two ConformerBlocks the residual stream passes through post_norm of block i and straight into the first FeedForward's
// PreNorm of block i+1 (lid/conformer.py:252-259, 153-171): y1 = LN1(x) (f32, kept: it is block i's output and block i+1's
// residual), y2 = LN2(y1) (T: the GEMM operand).  Row-local, so one launch does both - one read of x instead of two passes.
// One row per wave; the arithmetic of each LayerNorm is ln_fwd_kernel's, applied to the f32 y1 exactly as stored.
template <typename T>
__global__ void __launch_bounds__(256)
ln2_fwd_c256_kernel(const float* __restrict__ x, const float* __restrict__ g1, const float* __restrict__ b1,
                    float* __restrict__ y1, float* __restrict__ mean1, float* __restrict__ rstd1,
                    const float* __restrict__ g2, const float* __restrict__ b2, T* __restrict__ y2, float* __restrict__ mean2,
                    float* __restrict__ rstd2, int M, int C, float eps) {
  const int lane = threadIdx.x & 63, c = lane * 4;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const bool ok = c < C;
  float4 v = make_float4(0, 0, 0, 0), ga = v, ba = v, gb = v, bb = v;
  if (ok) { v = load4(x + (size_t)row * C + c); ga = load4(g1 + c); ba = load4(b1 + c); gb = load4(g2 + c); bb = load4(b2 + c); }
  float s = v.x + v.y + v.z + v.w;
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
  if (ok) { float a = v.x - mu, b = v.y - mu, d = v.z - mu, e = v.w - mu; q += a * a + b * b + d * d + e * e; }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  float4 o = make_float4(0, 0, 0, 0);
  if (ok) {
    o.x = (v.x - mu) * rs * ga.x + ba.x; o.y = (v.y - mu) * rs * ga.y + ba.y;
    o.z = (v.z - mu) * rs * ga.z + ba.z; o.w = (v.w - mu) * rs * ga.w + ba.w;
    store4(y1 + (size_t)row * C + c, o);
  }
  s = o.x + o.y + o.z + o.w;
  const float mu2 = wave_sum(s) / (float)C;
  q = 0.f;
  if (ok) { float a = o.x - mu2, b = o.y - mu2, d = o.z - mu2, e = o.w - mu2; q += a * a + b * b + d * d + e * e; }
  const float rs2 = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) { mean1[row] = mu; rstd1[row] = rs; mean2[row] = mu2; rstd2[row] = rs2; }
  if (ok) {
    float4 p;
    p.x = (o.x - mu2) * rs2 * gb.x + bb.x; p.y = (o.y - mu2) * rs2 * gb.y + bb.y;
    p.z = (o.z - mu2) * rs2 * gb.z + bb.z; p.w = (o.w - mu2) * rs2 * gb.w + bb.w;
    store4(y2 + (size_t)row * C + c, p);
  }
}
extern "C" int lidk_layernorm2_fwd(const float* x, const float* g1, const float* b1, float* y1, float* mean1, float* rstd1,
                                   const float* g2, const float* b2, void* y2, float* mean2, float* rstd2, int M, int C, float eps,
                                   int dtype, void* stream) {
  if (!x || !g1 || !b1 || !y1 || !mean1 || !rstd1 || !g2 || !b2 || !y2 || !mean2 || !rstd2 || M <= 0 || C <= 0 || (C & 3) || C > 256)
    return LIDK_ERR_ARG;
  LIDK_DISPATCH(dtype, ln2_fwd_c256_kernel<T><<<cdiv(M, 4), 256, 0, as_stream(stream)>>>(x, g1, b1, y1, mean1, rstd1, g2, b2, (T*)y2,
                                                                                        mean2, rstd2, M, C, eps));
  return launch_status();
}

// Backward of the same pair: dy (T) is the gradient at y2, dres (f32) the gradient reaching y1 along the residual path.
//   dv = LN2'(dy; y1, mean2, rstd2, g2) + dres ;  dx = LN1'(dv; x, mean1, rstd1, g1)  -> dx (f32) and dxT = dxT_scale * dx (T)
// partial1 / partial2 receive the per-workgroup (dgamma | dbeta) rows of LN1 / LN2 in lidk_layernorm_bwd's layout (finish them
// with lidk_layernorm_param_grads).  The intermediate dv is never written.
template <typename T>
__global__ void __launch_bounds__(256)
ln2_bwd_c256_kernel(const T* __restrict__ dy, const float* __restrict__ dres, const float* __restrict__ y1,
                    const float* __restrict__ mean2, const float* __restrict__ rstd2, const float* __restrict__ g2,
                    const float* __restrict__ x, const float* __restrict__ mean1, const float* __restrict__ rstd1,
                    const float* __restrict__ g1, float* __restrict__ dx, T* __restrict__ dxT, float dxT_scale,
                    float* __restrict__ partial1, float* __restrict__ partial2, int M, int C) {
  __shared__ float red[4][4 * 256];                          // [wave][dg1 | db1 | dg2 | db2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane * 4;
  const bool ok = c < C;
  float4 dg1 = make_float4(0, 0, 0, 0), db1 = dg1, dg2 = dg1, db2 = dg1;
  const float4 gm1 = ok ? load4(g1 + c) : dg1, gm2 = ok ? load4(g2 + c) : dg1;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mu2 = mean2[row], rs2 = rstd2[row], mu1 = mean1[row], rs1 = rstd1[row];
    float4 d = make_float4(0, 0, 0, 0), v = d, r = d, u = d;
    if (ok) {
      d = load4(dy + (size_t)row * C + c); v = load4(y1 + (size_t)row * C + c);
      r = load4(dres + (size_t)row * C + c); u = load4(x + (size_t)row * C + c);
    }
    // ---- LN2 backward (+ residual)
    float4 xh, g;
    xh.x = (v.x - mu2) * rs2; xh.y = (v.y - mu2) * rs2; xh.z = (v.z - mu2) * rs2; xh.w = (v.w - mu2) * rs2;
    g.x = d.x * gm2.x; g.y = d.y * gm2.y; g.z = d.z * gm2.z; g.w = d.w * gm2.w;
    float s1 = 0.f, s2 = 0.f;
    if (ok) {
      s1 += g.x + g.y + g.z + g.w;
      s2 += g.x * xh.x + g.y * xh.y + g.z * xh.z + g.w * xh.w;
      dg2.x += d.x * xh.x; dg2.y += d.y * xh.y; dg2.z += d.z * xh.z; dg2.w += d.w * xh.w;
      db2.x += d.x; db2.y += d.y; db2.z += d.z; db2.w += d.w;
    }
    float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
    float4 dv;
    dv.x = rs2 * (g.x - m1 - xh.x * m2) + r.x; dv.y = rs2 * (g.y - m1 - xh.y * m2) + r.y;
    dv.z = rs2 * (g.z - m1 - xh.z * m2) + r.z; dv.w = rs2 * (g.w - m1 - xh.w * m2) + r.w;
    // ---- LN1 backward
    xh.x = (u.x - mu1) * rs1; xh.y = (u.y - mu1) * rs1; xh.z = (u.z - mu1) * rs1; xh.w = (u.w - mu1) * rs1;
    g.x = dv.x * gm1.x; g.y = dv.y * gm1.y; g.z = dv.z * gm1.z; g.w = dv.w * gm1.w;
    s1 = 0.f; s2 = 0.f;
    if (ok) {
      s1 += g.x + g.y + g.z + g.w;
      s2 += g.x * xh.x + g.y * xh.y + g.z * xh.z + g.w * xh.w;
      dg1.x += dv.x * xh.x; dg1.y += dv.y * xh.y; dg1.z += dv.z * xh.z; dg1.w += dv.w * xh.w;
      db1.x += dv.x; db1.y += dv.y; db1.z += dv.z; db1.w += dv.w;
    }
    m1 = wave_sum(s1) / (float)C; m2 = wave_sum(s2) / (float)C;
    if (ok) {
      float4 o;
      o.x = rs1 * (g.x - m1 - xh.x * m2); o.y = rs1 * (g.y - m1 - xh.y * m2);
      o.z = rs1 * (g.z - m1 - xh.z * m2); o.w = rs1 * (g.w - m1 - xh.w * m2);
      store4(dx + (size_t)row * C + c, o);
      if (dxT) {
        o.x *= dxT_scale; o.y *= dxT_scale; o.z *= dxT_scale; o.w *= dxT_scale;
        store4(dxT + (size_t)row * C + c, o);
      }
    }
  }
  if (ok) {
    store4(&red[wave][c], dg1); store4(&red[wave][C + c], db1);
    store4(&red[wave][2 * C + c], dg2); store4(&red[wave][3 * C + c], db2);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    partial1[(size_t)blockIdx.x * 2 * C + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    partial2[(size_t)blockIdx.x * 2 * C + i] = red[0][2 * C + i] + red[1][2 * C + i] + red[2][2 * C + i] + red[3][2 * C + i];
  }
}
extern "C" int lidk_layernorm2_bwd(const void* dy, const float* dres, const float* y1, const float* mean2, const float* rstd2,
                                   const float* g2, const float* x, const float* mean1, const float* rstd1, const float* g1,
                                   float* dx, void* dxT, float dxT_scale, float* partial1, float* partial2, int M, int C, int dtype,
                                   void* stream) {
  if (!dy || !dres || !y1 || !mean2 || !rstd2 || !g2 || !x || !mean1 || !rstd1 || !g1 || !dx || !partial1 || !partial2 || M <= 0 ||
      C <= 0 || (C & 3) || C > 256)
    return LIDK_ERR_ARG;
  const int G = cdiv(M, 4) < LIDK_LN_BWD_BLOCKS ? cdiv(M, 4) : LIDK_LN_BWD_BLOCKS;       // = lidk_layernorm_bwd's partial rows
  LIDK_DISPATCH(dtype, ln2_bwd_c256_kernel<T><<<G, 256, 0, as_stream(stream)>>>((const T*)dy, dres, y1, mean2, rstd2, g2, x, mean1,
                                                                               rstd1, g1, dx, (T*)dxT, dxT_scale, partial1,
                                                                               partial2, M, C));
  return launch_status();
}

extern "C" int lidk_layernorm_param_grads(const float* partial, int M, int C, float* dgamma, float* dbeta, void* stream) {
  if (!partial || M <= 0 || C <= 0 || (C & 3) || C > 256 * LN_MAX_VEC || (!dgamma && !dbeta)) return LIDK_ERR_ARG;
  int G = cdiv(M, 4) < LIDK_LN_BWD_BLOCKS ? cdiv(M, 4) : LIDK_LN_BWD_BLOCKS;     // the partial rows lidk_layernorm_bwd(M) wrote
  colreduce_kernel<float, float, true><<<cdiv(2 * C, 16), 1024, 0, as_stream(stream)>>>(partial, G, 2 * C, dgamma, dbeta, C, 1.0f);
  return launch_status();
}

// Several finalisers in one launch (a ConformerBlock has five LayerNorms: five 5 us launches on the weight-gradient stream
// otherwise).  descs: device array of n records; record i reduces rows_i partial rows of (dgamma | dbeta) [2*C].
struct LnPgDesc { const float* partial; float* dgamma; float* dbeta; int rows; int C; };
__global__ void __launch_bounds__(1024)
ln_param_grads_grouped_kernel(const LnPgDesc* __restrict__ descs) {
  __shared__ float red[16][64];
  const LnPgDesc e = descs[blockIdx.y];
  const int ncols = 2 * e.C;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane >> 4;
  const int c = blockIdx.x * 16 + (lane & 15);
  float acc = 0.f;
  if (c < ncols) {
#pragma unroll 4
    for (int p = w * 4 + sub; p < e.rows; p += 64) acc += e.partial[(size_t)p * ncols + c];
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w != 0) return;
  float s = 0.f;                                       // same order as colreduce_kernel's second stage: bit-identical sums
#pragma unroll
  for (int i = 0; i < 16; ++i) s += red[i][lane];
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if (lane < 16 && c < ncols) {
    float* dst = (c < e.C) ? e.dgamma + c : e.dbeta + (c - e.C);
    *dst += s;
  }
}
extern "C" int lidk_ln_param_grads_desc_bytes(void) { return (int)sizeof(LnPgDesc); }
extern "C" int lidk_layernorm_param_grads_grouped(const void* descs, int n, int C, void* stream) {
  if (!descs || n <= 0 || C <= 0) return LIDK_ERR_ARG;
  ln_param_grads_grouped_kernel<<<dim3(cdiv(2 * C, 16), n), 1024, 0, as_stream(stream)>>>((const LnPgDesc*)descs);
  return launch_status();
}

extern "C" int lidk_layernorm_param_grads_rows(const float* partial, int rows, int C, float* dgamma, float* dbeta, void* stream) {
  if (!partial || rows <= 0 || C <= 0 || (C & 3) || C > 256 * LN_MAX_VEC || (!dgamma && !dbeta)) return LIDK_ERR_ARG;
  colreduce_kernel<float, float, true><<<cdiv(2 * C, 16), 1024, 0, as_stream(stream)>>>(partial, rows, 2 * C, dgamma, dbeta, C, 1.0f);
  return launch_status();
}
