// Row-panel GEMM for the K = 256 (= encoder width d) projections, optionally with the LayerNorm that precedes them fused
// into the operand load:  out[M][N] = epilogue( LN(x)[M][256] . W[N][256]^T )   (see include/lidk.h, lidk_ln_gemm_nt).
//
// In the Conformer block every LayerNorm output feeds exactly one GEMM whose K is the model width (ff up-projection N = 4d,
// fused QKV N = 3*inner, pointwise conv 1 N = 2*ci), and the whole K extent of a 64-row panel is 64 x 256 bf16 = 32 KB: it
// fits in LDS next to a 64-column weight chunk.  So a workgroup
//   1. reads its 64 rows of the f32 residual stream ONCE (one wave per row: 64 lanes x 16 B = the 1 KB row), computes the row
//      statistics with wave shuffles exactly as ln_fwd_kernel does (two-pass, rsqrtf), normalises, and parks the bf16 panel in
//      LDS in MFMA-fragment order (column group 0 also writes h / mean / rstd, which the backward pass consumes);
//   2. walks NCH consecutive 64-column chunks of W with the panel resident: W chunk -> registers -> LDS, 32 MFMAs per wave,
//      fused epilogue, while the next chunk's loads are already in flight and the previous chunk's stores drain - the
//      stores of a chunk never sit on the critical path of a workgroup's slot (in the per-tile kernels they are half a launch).
// The standalone LayerNorm launch, its read of x and the re-read of h by 4..16 column tiles disappear (h is still written
// once: the weight-gradient GEMM consumes it).  With LN = false the same kernel takes a bf16 A panel (dgrad GEMMs with K = d).
//
// LDS images: [kt][row][64] bf16 per 64-wide K sub-tile, 16-byte chunk index XOR (row & 7) - conflict-free ds_read_b128
// fragments and full-row ds_write_b128, as in gemm_nt_bf16_direct_kernel; W rows are permuted (rho) so that a lane owns 8
// consecutive output columns (one 16-byte store), MFMA issued transposed (W fragment first).
#include "common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { PM_PLAIN = 0, PM_BIAS = 1, PM_BIAS_SWISH_PRE = 2, PM_SWISH_GRAD = 3 };

struct PanelArgs {
  const float* x; int ldx; const float* gamma; const float* beta; float eps;      // LN = true: f32 rows to normalise
  const bf16* A; int lda;                                                         // LN = false: bf16 operand
  bf16* h; float* mean; float* rstd;                                              // LN side outputs (column group 0 writes them)
  const bf16* W; int ldb;
  int M, N;
  const float* bias; bf16* out; int ldo; bf16* out2; int ldo2; const bf16* aux; int ldaux;
};

__device__ __forceinline__ int xcd_unit(int L, int total) {       // see gemm.hip: each XCD walks one contiguous range of units
  const int xcd = L & 7, slot = L >> 3, q = total >> 3, r = total & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}
__device__ __forceinline__ uint4 pack8p(float4 a, float4 b) {
  union { bf16 h[8]; uint4 u; } p;
  p.h[0] = (bf16)a.x; p.h[1] = (bf16)a.y; p.h[2] = (bf16)a.z; p.h[3] = (bf16)a.w;
  p.h[4] = (bf16)b.x; p.h[5] = (bf16)b.y; p.h[6] = (bf16)b.z; p.h[7] = (bf16)b.w;
  return p.u;
}

template <int MODE, bool LN>
__global__ void __launch_bounds__(256)
gemm_k256_panel_kernel(PanelArgs p, int col_groups, int nch) {
  constexpr int BK = 64;
  __shared__ __attribute__((aligned(16))) bf16 As[4 * 64 * BK];
  __shared__ __attribute__((aligned(16))) bf16 Ws[4 * 64 * BK];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int unit = xcd_unit(blockIdx.x, gridDim.x);
  const int rb = unit / col_groups, cg = unit % col_groups;
  const int m0 = rb * 64, ncol0 = cg * nch * 64;

  // ---- W chunk staging: 64 rows x 256 k = 2048 16-byte pieces, 8 per thread; a row is one 512-byte run of 32 threads
  u32x4 rw[8];
  auto wload = [&](int j) __attribute__((always_inline)) {
    const bf16* base = p.W + (size_t)(ncol0 + j * 64) * p.ldb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + i * 256, r = c >> 5, kc = c & 31;
      rw[i] = *reinterpret_cast<const u32x4*>(base + (size_t)r * p.ldb + kc * 8);
    }
  };
  auto wstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + i * 256, r = c >> 5, kc = c & 31, kt = kc >> 3, ch = kc & 7;
      const int rho = (r & ~31) + (((r >> 2) & 1) << 4) + (((r >> 3) & 3) << 2) + (r & 3);
      *reinterpret_cast<u32x4*>(&Ws[(kt * 64 + rho) * BK + ((ch ^ (rho & 7)) << 3)]) = rw[i];
    }
  };
  wload(0);

  // ---- A panel
  if (LN) {
    const float4 g = load4(p.gamma + lane * 4), b = load4(p.beta + lane * 4);
    const int kt = lane >> 4, ch = (lane & 15) >> 1, sub = (lane & 1) * 4;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
      const int rl = wid * 16 + r, row = min(m0 + rl, p.M - 1);
      const float4 v = load4(p.x + (size_t)row * p.ldx + lane * 4);
      const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / 256.0f);
      const float a0 = v.x - mu, a1 = v.y - mu, a2 = v.z - mu, a3 = v.w - mu;
      const float rs = rsqrtf(wave_sum(a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3) * (1.0f / 256.0f) + p.eps);
      float4 o;
      o.x = a0 * rs * g.x + b.x; o.y = a1 * rs * g.y + b.y; o.z = a2 * rs * g.z + b.z; o.w = a3 * rs * g.w + b.w;
      union { bf16 hh[4]; uint2 u; } t;
      t.hh[0] = (bf16)o.x; t.hh[1] = (bf16)o.y; t.hh[2] = (bf16)o.z; t.hh[3] = (bf16)o.w;
      *reinterpret_cast<uint2*>(&As[(kt * 64 + rl) * BK + ((ch ^ (rl & 7)) << 3) + sub]) = t.u;
      if (cg == 0 && m0 + rl < p.M) {
        if (p.h) *reinterpret_cast<uint2*>(p.h + (size_t)row * 256 + lane * 4) = t.u;
        if (lane == 0) { if (p.mean) p.mean[row] = mu; if (p.rstd) p.rstd[row] = rs; }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + i * 256, r = c >> 5, kc = c & 31, kt = kc >> 3, ch = kc & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(p.A + (size_t)min(m0 + r, p.M - 1) * p.lda + kc * 8);
      *reinterpret_cast<u32x4*>(&As[(kt * 64 + r) * BK + ((ch ^ (r & 7)) << 3)]) = v;
    }
  }

  f32x4 acc[2][2];
  auto epi = [&](int i, int n0) __attribute__((always_inline)) {
    const int m = m0 + wm * 32 + i * 16 + fr, n = n0 + wn * 32 + 8 * fq;
    if (m >= p.M) return;
    float4 v0 = make_float4(acc[i][0][0], acc[i][0][1], acc[i][0][2], acc[i][0][3]);
    float4 v1 = make_float4(acc[i][1][0], acc[i][1][1], acc[i][1][2], acc[i][1][3]);
    if (MODE == PM_BIAS || MODE == PM_BIAS_SWISH_PRE) {
      const float4 b0 = load4(p.bias + n), b1 = load4(p.bias + n + 4);
      v0.x += b0.x; v0.y += b0.y; v0.z += b0.z; v0.w += b0.w; v1.x += b1.x; v1.y += b1.y; v1.z += b1.z; v1.w += b1.w;
    }
    if (MODE == PM_BIAS_SWISH_PRE) {
      *reinterpret_cast<uint4*>(p.out2 + (size_t)m * p.ldo2 + n) = pack8p(v0, v1);
      v0.x *= sigmoidf_(v0.x); v0.y *= sigmoidf_(v0.y); v0.z *= sigmoidf_(v0.z); v0.w *= sigmoidf_(v0.w);
      v1.x *= sigmoidf_(v1.x); v1.y *= sigmoidf_(v1.y); v1.z *= sigmoidf_(v1.z); v1.w *= sigmoidf_(v1.w);
    }
    if (MODE == PM_SWISH_GRAD) {
      const bf16* ap = p.aux + (size_t)m * p.ldaux + n;
      const float4 a0 = load4(ap), a1 = load4(ap + 4);
      float s;
      s = sigmoidf_(a0.x); v0.x *= s * (1.f + a0.x * (1.f - s)); s = sigmoidf_(a0.y); v0.y *= s * (1.f + a0.y * (1.f - s));
      s = sigmoidf_(a0.z); v0.z *= s * (1.f + a0.z * (1.f - s)); s = sigmoidf_(a0.w); v0.w *= s * (1.f + a0.w * (1.f - s));
      s = sigmoidf_(a1.x); v1.x *= s * (1.f + a1.x * (1.f - s)); s = sigmoidf_(a1.y); v1.y *= s * (1.f + a1.y * (1.f - s));
      s = sigmoidf_(a1.z); v1.z *= s * (1.f + a1.z * (1.f - s)); s = sigmoidf_(a1.w); v1.w *= s * (1.f + a1.w * (1.f - s));
    }
    *reinterpret_cast<uint4*>(p.out + (size_t)m * p.ldo + n) = pack8p(v0, v1);
  };

  for (int j = 0; j < nch; ++j) {
    wstore();
    __syncthreads();                                   // W chunk j (and, the first time, the A panel) visible to every wave
    wload(min(j + 1, nch - 1));                        // unconditional (the last chunk is re-read once from L2): counted vmcnt
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int kk = 0; kk < 8; kk += 4) {
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          af[i] = *reinterpret_cast<const bf16x8*>(&As[(kt * 64 + wm * 32 + i * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          bfr[jj] = *reinterpret_cast<const bf16x8*>(&Ws[(kt * 64 + wn * 32 + jj * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jj], af[i], acc[i][jj], 0, 0, 0);
      }
    }
    const int n0 = ncol0 + j * 64;
    epi(0, n0);
    epi(1, n0);
    __syncthreads();                                   // every wave is done with Ws before the next chunk overwrites it
  }
}

extern "C" int lidk_ln_gemm_supported(int M, int N, int K, int dtype) {
  return dtype == LIDK_BF16 && K == 256 && M > 0 && N >= 256 && (N % 256) == 0;
}

extern "C" int lidk_ln_gemm_nt(const lidk_gemm_args* g, const float* x, int ldx, const float* gamma, const float* beta,
                               float eps, void* h, float* mean, float* rstd, int dtype, void* stream) {
  if (!g || !g->B || !g->out) return LIDK_ERR_ARG;
  if (!lidk_ln_gemm_supported(g->M, g->N, g->K, dtype)) return LIDK_ERR_UNSUPPORTED;
  const bool ln = x != nullptr;
  if (ln && (!gamma || !beta || ldx < 256 || (ldx & 3))) return LIDK_ERR_ARG;
  if (!ln && (!g->A || (g->lda & 7) || g->lda < 256)) return LIDK_ERR_ARG;
  if ((g->ldb & 7) || g->ldb < 256 || (g->ldo & 7) || g->ldo < g->N || g->out_f32 || g->res || g->alpha != 1.0f || g->splitk > 1)
    return LIDK_ERR_ARG;
  int mode = -1;
  if (g->act == LIDK_ACT_NONE) mode = g->bias ? PM_BIAS : PM_PLAIN;
  else if (g->act == LIDK_ACT_SWISH && g->bias && g->out2 && !(g->ldo2 & 7)) mode = PM_BIAS_SWISH_PRE;
  else if (g->act == LIDK_ACT_SWISH_GRAD && !g->bias && g->aux && !(g->ldaux & 7)) mode = PM_SWISH_GRAD;
  if (mode < 0) return LIDK_ERR_UNSUPPORTED;
  PanelArgs p{x, ldx, gamma, beta, eps, (const bf16*)g->A, g->lda, (bf16*)h, mean, rstd, (const bf16*)g->B, g->ldb, g->M, g->N,
              g->bias, (bf16*)g->out, g->ldo, (bf16*)g->out2, g->ldo2, (const bf16*)g->aux, g->ldaux};
  const int nch = 4, col_groups = g->N / (64 * nch), row_blocks = cdiv(g->M, 64);
  const int grid = row_blocks * col_groups;
  hipStream_t s = as_stream(stream);
#define LIDK_PANEL_LAUNCH(MODE_)                                                                          \
  do {                                                                                                    \
    if (ln) gemm_k256_panel_kernel<MODE_, true><<<grid, 256, 0, s>>>(p, col_groups, nch);                  \
    else gemm_k256_panel_kernel<MODE_, false><<<grid, 256, 0, s>>>(p, col_groups, nch);                    \
  } while (0)
  if (mode == PM_PLAIN) LIDK_PANEL_LAUNCH(PM_PLAIN);
  else if (mode == PM_BIAS) LIDK_PANEL_LAUNCH(PM_BIAS);
  else if (mode == PM_BIAS_SWISH_PRE) LIDK_PANEL_LAUNCH(PM_BIAS_SWISH_PRE);
  else LIDK_PANEL_LAUNCH(PM_SWISH_GRAD);
#undef LIDK_PANEL_LAUNCH
  return launch_status();
}
