// C[M][N] = A[M][K] * B[N][K]^T with a fused epilogue (see include/lidk.h, lidk_gemm_nt).
//
// bf16 path: v_mfma_f32_16x16x32_bf16, 256 threads = 4 waves arranged 2x2 over a BMxBN block tile, BK = 64.
//   Both operands are K-contiguous, so a lane's MFMA fragment (8 consecutive k of one row) is one 16-byte LDS read.
//   Global -> registers (16 B per lane, 128-B row segments) -> LDS; the next K tile's global loads are issued
//   before the MFMAs of the current tile so HBM latency hides under compute.  LDS rows are padded by 16 B.
//   Fragment maps (cdna_hip_programming.md section 3): A/B lane l holds row (l&15), k = 8*(l>>4)+j;
//   C/D lane l, reg r -> row 4*(l>>4)+r, col (l&15).
// f32 path (parity mode): 64x64 tile, BK = 16, 4x4 outputs per thread, plain FMA.
#include "common.h"

struct Epi {
  const float* bias; int act; float alpha;
  const float* res; int ldres;
  void* out; int ldo; int out_f32;
  void* out2; int ldo2;
  const void* aux; int ldaux;
  int atomic;
};

template <typename T>
__device__ __forceinline__ void epi_store(const Epi& e, int m, int n, float acc) {
  float v = acc;
  if (e.bias) v += e.bias[n];
  if (e.act == LIDK_ACT_SWISH) {
    if (e.out2) ((T*)e.out2)[(size_t)m * e.ldo2 + n] = from_f<T>(v);
    v = v * sigmoidf_(v);
  } else if (e.act == LIDK_ACT_RELU) {
    v = fmaxf(v, 0.f);
  } else if (e.act == LIDK_ACT_SWISH_GRAD) {
    float a = to_f(((const T*)e.aux)[(size_t)m * e.ldaux + n]);
    float s = sigmoidf_(a);
    v *= s * (1.f + a * (1.f - s));
  }
  v *= e.alpha;
  if (e.res) v += e.res[(size_t)m * e.ldres + n];
  if (e.atomic) atomicAdd(&((float*)e.out)[(size_t)m * e.ldo + n], v);
  else if (e.out_f32) ((float*)e.out)[(size_t)m * e.ldo + n] = v;
  else ((T*)e.out)[(size_t)m * e.ldo + n] = from_f<T>(v);
}

// ------------------------------------------------------------------------------------ bf16 MFMA kernel
#define BK 64
#define LDS_STRIDE (BK + 8)   // bf16 elements; 144-byte rows keep 16-byte alignment and break the power-of-2 stride

template <int BM, int BN>
__global__ void __launch_bounds__(256)
gemm_nt_bf16_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda, int ldb,
                    int kchunk, Epi e) {
  constexpr int TM = BM / 32, TN = BN / 32;        // 16x16 tiles per wave along M / N
  constexpr int CA = BM / 32, CB = BN / 32;        // 16-byte chunks per thread per K tile
  __shared__ __attribute__((aligned(16))) bf16 As[BM * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) bf16 Bs[BN * LDS_STRIDE];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * kchunk;
  const int kend = min(K, kbeg + kchunk);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 ra[CA], rb[CB];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      int gm = m0 + row, gk = k0 + kc;
      ra[i] = (gm < M && gk < kend) ? *reinterpret_cast<const uint4*>(A + (size_t)gm * lda + gk) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      int gn = n0 + row, gk = k0 + kc;
      rb[i] = (gn < N && gk < kend) ? *reinterpret_cast<const uint4*>(B + (size_t)gn * ldb + gk) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      *reinterpret_cast<uint4*>(&As[row * LDS_STRIDE + kc]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      *reinterpret_cast<uint4*>(&Bs[row * LDS_STRIDE + kc]) = rb[i];
    }
  };

  gload(kbeg);
  lstore();
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) gload(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 32) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * (BM / 2) + i * 16 + fr) * LDS_STRIDE + kk + fq * 8]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * (BN / 2) + j * 16 + fr) * LDS_STRIDE + kk + fq * 8]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (more) { lstore(); __syncthreads(); }
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = m0 + wm * (BM / 2) + i * 16 + fq * 4 + r;
        int n = n0 + wn * (BN / 2) + j * 16 + fr;
        if (m < M && n < N) epi_store<bf16>(e, m, n, acc[i][j][r]);
      }
}

// ------------------------------------------------------------------------------------ f32 kernel (parity mode)
__global__ void __launch_bounds__(256)
gemm_nt_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, int M, int N, int K, int lda, int ldb,
                   int kchunk, Epi e) {
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
  float acc[4][4] = {};
  const int lr = tid >> 2, lk = (tid & 3) * 4;     // this thread stages row lr, k offsets lk..lk+3
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0);
    if (m0 + lr < M && k0 + lk < kend) a = *reinterpret_cast<const float4*>(A + (size_t)(m0 + lr) * lda + k0 + lk);
    if (n0 + lr < N && k0 + lk < kend) b = *reinterpret_cast<const float4*>(B + (size_t)(n0 + lr) * ldb + k0 + lk);
    As[lk + 0][lr] = a.x; As[lk + 1][lr] = a.y; As[lk + 2][lr] = a.z; As[lk + 3][lr] = a.w;
    Bs[lk + 0][lr] = b.x; Bs[lk + 1][lr] = b.y; Bs[lk + 2][lr] = b.z; Bs[lk + 3][lr] = b.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = As[k][ty * 4 + i]; bv[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < M && n < N) epi_store<float>(e, m, n, acc[i][j]);
    }
}

// ------------------------------------------------------------------------------------ host dispatch
extern "C" int lidk_gemm_nt(const lidk_gemm_args* g, int dtype, void* stream) {
  if (!g || !g->A || !g->B || !g->out || g->M <= 0 || g->N <= 0 || g->K <= 0) return LIDK_ERR_ARG;
  if ((g->K & 7) || (g->lda & 7) || (g->ldb & 7) || g->lda < g->K || g->ldb < g->K || g->ldo < g->N) return LIDK_ERR_ARG;
  if (g->act == LIDK_ACT_SWISH_GRAD && !g->aux) return LIDK_ERR_ARG;
  int splitk = g->splitk > 1 ? g->splitk : 1;
  if (splitk > 1 && (g->bias || g->res || g->act != LIDK_ACT_NONE || !g->out_f32)) return LIDK_ERR_ARG;
  Epi e{g->bias, g->act, g->alpha, g->res, g->ldres, g->out, g->ldo, g->out_f32, g->out2, g->ldo2, g->aux, g->ldaux,
        splitk > 1 ? 1 : 0};
  hipStream_t s = as_stream(stream);
  if (dtype == LIDK_BF16) {
    int kchunk = cdiv(cdiv(g->K, splitk), BK) * BK;
    splitk = cdiv(g->K, kchunk);
    // tile choice: big tiles when the grid still fills 256 CUs, otherwise 64x64 for more workgroups
    long big = (long)cdiv(g->M, 128) * cdiv(g->N, 128) * splitk;
    if (big >= 384) {
      dim3 grid(cdiv(g->N, 128), cdiv(g->M, 128), splitk);
      gemm_nt_bf16_kernel<128, 128><<<grid, 256, 0, s>>>((const bf16*)g->A, (const bf16*)g->B, g->M, g->N, g->K, g->lda,
                                                        g->ldb, kchunk, e);
    } else {
      dim3 grid(cdiv(g->N, 64), cdiv(g->M, 64), splitk);
      gemm_nt_bf16_kernel<64, 64><<<grid, 256, 0, s>>>((const bf16*)g->A, (const bf16*)g->B, g->M, g->N, g->K, g->lda,
                                                      g->ldb, kchunk, e);
    }
  } else if (dtype == LIDK_F32) {
    int kchunk = cdiv(cdiv(g->K, splitk), 16) * 16;
    splitk = cdiv(g->K, kchunk);
    dim3 grid(cdiv(g->N, 64), cdiv(g->M, 64), splitk);
    gemm_nt_f32_kernel<<<grid, 256, 0, s>>>((const float*)g->A, (const float*)g->B, g->M, g->N, g->K, g->lda, g->ldb,
                                            kchunk, e);
  } else {
    return LIDK_ERR_ARG;
  }
  return launch_status();
}
