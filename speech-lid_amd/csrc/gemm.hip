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
#include <string.h>
#include <type_traits>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Workgroups are dispatched round-robin over the 8 XCDs (each with a private L2) in launch order.  Map launch index L to a
// logical tile index so that every XCD works through ONE contiguous range of tiles: tiles that share an operand panel
// (same row block, neighbouring column tiles) then hit the same L2 instead of fetching the panel once per XCD
// (rocprofv3 FETCH_SIZE on the ff-up GEMM: 39 MB with the plain grid for 5.5 MB of operands).  Bijective for any count.
__device__ __forceinline__ int xcd_tile(int L, int total) {
  const int xcd = L & 7, slot = L >> 3, q = total >> 3, r = total & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

struct Epi {
  int dbg;                 // tuning aid (LIDK_GEMM_DBG): 1 = skip the epilogue stores, 2 = skip the K loop
  const float* bias; int act; float alpha;
  const float* res; int ldres;
  void* out; int ldo; int out_f32;
  void* out2; int ldo2;
  const void* aux; int ldaux;
  int atomic;
  // PIPE_BN_SUMS (lidk_gemm_nt_bn_sums): BatchNorm + Swish backward statistics of the output tile, see the pipelined kernel
  const float* bn_mean; const float* bn_rstd; const float* bn_gamma; const float* bn_beta; float* bn_partial;
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
  } else if (e.act == LIDK_ACT_GELU) {
    if (e.out2) ((T*)e.out2)[(size_t)m * e.ldo2 + n] = from_f<T>(v);
    v = gelu_(v);
  } else if (e.act == LIDK_ACT_SWISH_GRAD) {
    float a = to_f(((const T*)e.aux)[(size_t)m * e.ldaux + n]);
    float s = sigmoidf_(a);
    v *= s * (1.f + a * (1.f - s));
  } else if (e.act == LIDK_ACT_GELU_GRAD) {
    v *= gelu_grad_(to_f(((const T*)e.aux)[(size_t)m * e.ldaux + n]));
  }
  v *= e.alpha;
  if (e.res) v += e.res[(size_t)m * e.ldres + n];
  if (e.atomic) atomicAdd(&((float*)e.out)[(size_t)m * e.ldo + n], v);
  else if (e.out_f32) ((float*)e.out)[(size_t)m * e.ldo + n] = v;
  else ((T*)e.out)[(size_t)m * e.ldo + n] = from_f<T>(v);
}

// 4 consecutive columns n..n+3 of row m (n % 4 == 0): 16-byte loads of bias/res/aux and one 8/16-byte store.  Falls back to
// the scalar form at the N tail or when a leading dimension is not a multiple of 4.
template <typename T>
__device__ __forceinline__ void epi_store4(const Epi& e, int m, int n, int N, float4 acc) {
  const bool vec = (n + 3 < N) && !(e.ldo & 3) && !e.atomic && (!e.res || !(e.ldres & 3)) && (!e.out2 || !(e.ldo2 & 3)) &&
                   (!e.aux || !(e.ldaux & 3));
  if (!vec) {
    const float a[4] = {acc.x, acc.y, acc.z, acc.w};
    for (int i = 0; i < 4; ++i) if (n + i < N) epi_store<T>(e, m, n + i, a[i]);
    return;
  }
  float4 v = acc;
  if (e.bias) { float4 b = load4(e.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
  if (e.act == LIDK_ACT_SWISH) {
    if (e.out2) store4((T*)e.out2 + (size_t)m * e.ldo2 + n, v);
    v.x *= sigmoidf_(v.x); v.y *= sigmoidf_(v.y); v.z *= sigmoidf_(v.z); v.w *= sigmoidf_(v.w);
  } else if (e.act == LIDK_ACT_RELU) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  } else if (e.act == LIDK_ACT_GELU) {
    if (e.out2) store4((T*)e.out2 + (size_t)m * e.ldo2 + n, v);
    v.x = gelu_(v.x); v.y = gelu_(v.y); v.z = gelu_(v.z); v.w = gelu_(v.w);
  } else if (e.act == LIDK_ACT_GELU_GRAD) {
    float4 a = load4((const T*)e.aux + (size_t)m * e.ldaux + n);
    v.x *= gelu_grad_(a.x); v.y *= gelu_grad_(a.y); v.z *= gelu_grad_(a.z); v.w *= gelu_grad_(a.w);
  } else if (e.act == LIDK_ACT_SWISH_GRAD) {
    float4 a = load4((const T*)e.aux + (size_t)m * e.ldaux + n);
    float s;
    s = sigmoidf_(a.x); v.x *= s * (1.f + a.x * (1.f - s));
    s = sigmoidf_(a.y); v.y *= s * (1.f + a.y * (1.f - s));
    s = sigmoidf_(a.z); v.z *= s * (1.f + a.z * (1.f - s));
    s = sigmoidf_(a.w); v.w *= s * (1.f + a.w * (1.f - s));
  }
  v.x *= e.alpha; v.y *= e.alpha; v.z *= e.alpha; v.w *= e.alpha;
  if (e.res) { float4 r = load4(e.res + (size_t)m * e.ldres + n); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
  if (e.out_f32) store4((float*)e.out + (size_t)m * e.ldo + n, v);
  else store4((T*)e.out + (size_t)m * e.ldo + n, v);
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
  constexpr int WM = BM / 2, WN = BN / 2, CST = WN + 4;          // per-wave C tile, f32, padded rows
  constexpr int AB_BYTES = (BM + BN) * LDS_STRIDE * 2, C_BYTES = 4 * WM * CST * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES];
  bf16* As = reinterpret_cast<bf16*>(smem);
  bf16* Bs = As + BM * LDS_STRIDE;

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

  // PD K-tiles of global loads are kept in flight per thread (registers); loads complete in order, so consuming stage st
  // waits only for that stage.  (Ablation on MI355X, ff-up shape M=9664 N=1024 K=256: operand fetch ~9 us — the 64x64 tiles
  // re-read A and W through L2 — MFMA/LDS loop +6 us, epilogue stores +9 us, with little overlap between the phases.)
  constexpr int PD = 2;          // measured: 2 and 4 tiles in flight perform the same (tools/gemm_bench.py); 2 keeps 76 VGPRs
  uint4 ra[PD][CA], rb[PD][CB];
  auto gload = [&](int st, int k0) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      int gm = m0 + row, gk = k0 + kc;
      ra[st][i] = (gm < M && gk < kend) ? *reinterpret_cast<const uint4*>(A + (size_t)gm * lda + gk) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      int gn = n0 + row, gk = k0 + kc;
      rb[st][i] = (gn < N && gk < kend) ? *reinterpret_cast<const uint4*>(B + (size_t)gn * ldb + gk) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int st) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      *reinterpret_cast<uint4*>(&As[row * LDS_STRIDE + kc]) = ra[st][i];
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      *reinterpret_cast<uint4*>(&Bs[row * LDS_STRIDE + kc]) = rb[st][i];
    }
  };

#pragma unroll
  for (int st = 0; st < PD; ++st) gload(st, kbeg + st * BK);
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = kbeg; kt < ((e.dbg & 2) ? kbeg : kend); kt += PD * BK) {
#pragma unroll
    for (int st = 0; st < PD; ++st) {
      const int k0 = kt + st * BK;
      if (k0 < kend) {                                   // block-uniform
        lstore(st);
        __syncthreads();
        gload(st, k0 + PD * BK);                         // refill this stage (zero fill past the end: no memory access)
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
      }
    }
  }

  // Epilogue: accumulators -> this wave's f32 LDS tile -> row-major pass with 16-byte accesses (the K loop ended with a
  // workgroup barrier, so the A/B tiles are dead and the space is reused).
  float* Cw = reinterpret_cast<float*>(smem) + wid * WM * CST;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cw[(i * 16 + fq * 4 + r) * CST + j * 16 + fr] = acc[i][j][r];
  __builtin_amdgcn_wave_barrier();
  if (e.atomic) {
    // split-K accumulation: consecutive lanes -> consecutive columns, so one wave-instruction adds to 64 (or 2 x 32)
    // contiguous floats - the access shape float atomics run fastest on (MI355X_MICROARCH.md, "Global float atomics")
    constexpr int RPA = 64 / WN;
    for (int row = lane / WN; row < WM; row += RPA) {
      const int col = lane % WN, m = m0 + wm * WM + row, n = n0 + wn * WN + col;
      if (m < M && n < N) atomicAdd(&((float*)e.out)[(size_t)m * e.ldo + n], Cw[row * CST + col] * e.alpha);
    }
    return;
  }
  constexpr int LPR = WN / 4, RPI = 64 / LPR;                     // lanes per row, rows per wave-instruction
  const int crow = lane / LPR, ccol = (lane % LPR) * 4;
#pragma unroll 4
  for (int row = crow; row < WM; row += RPI) {
    const int m = m0 + wm * WM + row, n = n0 + wn * WN + ccol;
    if (m < M && n < N && !(e.dbg & 1)) epi_store4<bf16>(e, m, n, N, *reinterpret_cast<const float4*>(&Cw[row * CST + ccol]));
  }
}

// ------------------------------------------------------------------------------------ bf16 MFMA kernel, direct epilogue
// Same operand staging as gemm_nt_bf16_kernel (full 128-byte row segments global -> registers -> LDS, one K tile ahead),
// with the three changes the counters asked for (rocprofv3 --pmc on the ff-up shape: 32 % of the LDS cycles were bank
// conflicts, the epilogue was VALU-issue bound):
//   * LDS tiles are unpadded [rows][64] bf16 with the 16-byte chunk index XOR-ed by (row & 7): every 16-lane group of a
//     ds_read_b128 then covers all 16 slots of the 256-byte bank row, and the ds_write_b128 groups stay one full row;
//   * the MFMA is issued "transposed" (B fragment as the first operand), so a lane's 4 accumulator registers are 4
//     CONSECUTIVE output columns of one row: bias/residual/aux loads and the stores are 16-byte accesses straight from
//     registers, no LDS staging pass and no barrier after the K loop;
//   * for bf16 outputs the rows of the B tile are permuted when written to LDS so that two adjacent 16-column tiles give a
//     lane 8 consecutive columns (one 16-byte store; a wave instruction covers 16 rows x 64 contiguous bytes).
__device__ __forceinline__ float4 epi_math4(const Epi& e, int m, int n, float4 v, float4* pre) {
  if (e.bias) { float4 b = load4(e.bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
  if (e.act == LIDK_ACT_SWISH) {
    *pre = v;
    v.x *= sigmoidf_(v.x); v.y *= sigmoidf_(v.y); v.z *= sigmoidf_(v.z); v.w *= sigmoidf_(v.w);
  } else if (e.act == LIDK_ACT_RELU) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  } else if (e.act == LIDK_ACT_GELU) {
    *pre = v;
    v.x = gelu_(v.x); v.y = gelu_(v.y); v.z = gelu_(v.z); v.w = gelu_(v.w);
  } else if (e.act == LIDK_ACT_GELU_GRAD) {
    float4 a = load4((const bf16*)e.aux + (size_t)m * e.ldaux + n);
    v.x *= gelu_grad_(a.x); v.y *= gelu_grad_(a.y); v.z *= gelu_grad_(a.z); v.w *= gelu_grad_(a.w);
  } else if (e.act == LIDK_ACT_SWISH_GRAD) {
    float4 a = load4((const bf16*)e.aux + (size_t)m * e.ldaux + n);
    float s;
    s = sigmoidf_(a.x); v.x *= s * (1.f + a.x * (1.f - s));
    s = sigmoidf_(a.y); v.y *= s * (1.f + a.y * (1.f - s));
    s = sigmoidf_(a.z); v.z *= s * (1.f + a.z * (1.f - s));
    s = sigmoidf_(a.w); v.w *= s * (1.f + a.w * (1.f - s));
  }
  v.x *= e.alpha; v.y *= e.alpha; v.z *= e.alpha; v.w *= e.alpha;
  if (e.res) { float4 r = load4(e.res + (size_t)m * e.ldres + n); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
  return v;
}
__device__ __forceinline__ void st16(const Epi& e, void* p, uint4 v) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4 w = {v.x, v.y, v.z, v.w};
  if (e.dbg & 8) __builtin_nontemporal_store(w, reinterpret_cast<u4*>(p));
  else *reinterpret_cast<u4*>(p) = w;
}
__device__ __forceinline__ uint4 pack8(float4 a, float4 b) {
  union { bf16 h[8]; uint4 u; } p;
  p.h[0] = from_f<bf16>(a.x); p.h[1] = from_f<bf16>(a.y); p.h[2] = from_f<bf16>(a.z); p.h[3] = from_f<bf16>(a.w);
  p.h[4] = from_f<bf16>(b.x); p.h[5] = from_f<bf16>(b.y); p.h[6] = from_f<bf16>(b.z); p.h[7] = from_f<bf16>(b.w);
  return p.u;
}

template <int BM, int BN>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
gemm_nt_bf16_direct_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda, int ldb, Epi e) {
  const int n_tiles = (N + BN - 1) / BN;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);               // column tile fastest inside a row block
  constexpr int TM = BM / 32, TN = BN / 32;        // 16x16 tiles per wave along M / N (TN is even: tiles pair up)
  constexpr int CA = BM / 32, CB = BN / 32;        // 16-byte chunks per thread per K tile
  __shared__ __attribute__((aligned(16))) bf16 As[BM * BK];
  __shared__ __attribute__((aligned(16))) bf16 Bs[BN * BK];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = (tile / n_tiles) * BM, n0 = (tile % n_tiles) * BN;
  const bool pair = !e.out_f32;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Two K tiles of global loads are kept in flight per thread in two NAMED register sets.  Every load is unconditional
  // (host guarantees K % 64 == 0; row index clamped - rows past M / N only feed outputs that are never stored) and the
  // last tiles run in a peeled tail without prefetch: a predicated load becomes an exec-mask branch and hipcc then waits
  // vmcnt(0) before every LDS write, draining the other stage as well (measured: no gain from any prefetch depth);
  // stage arrays indexed by a variable fall to scratch memory.
  u32x4 ra0[CA], rb0[CB], ra1[CA], rb1[CB];
  auto gload = [&](u32x4 (&ra)[CA], u32x4 (&rb)[CB], int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      ra[i] = *reinterpret_cast<const u32x4*>(A + (size_t)min(m0 + row, M - 1) * lda + k0 + kc);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      rb[i] = *reinterpret_cast<const u32x4*>(B + (size_t)min(n0 + row, N - 1) * ldb + k0 + kc);
    }
  };
  auto lstore = [&](const u32x4 (&ra)[CA], const u32x4 (&rb)[CB]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int c = tid + i * 256, row = c >> 3, ch = c & 7;
      *reinterpret_cast<u32x4*>(&As[row * BK + ((ch ^ (row & 7)) << 3)]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int c = tid + i * 256, row = c >> 3, ch = c & 7;
      // bf16 outputs: column j of a 32-column group sits at LDS row (j>>2 & 1)*16 + (j>>3)*4 + (j&3) of that group
      int rho = pair ? (row & ~31) + (((row >> 2) & 1) << 4) + (((row >> 3) & 3) << 2) + (row & 3) : row;
      *reinterpret_cast<u32x4*>(&Bs[rho * BK + ((ch ^ (rho & 7)) << 3)]) = rb[i];
    }
  };
  auto mma = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < BK / 8; kk += 4) {             // kk: 16-byte chunk index of this 32-wide K step
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * (BM / 2) + i * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * (BN / 2) + j * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  // one K tile: registers -> LDS, refill the register set with the tile two ahead, MFMAs.  The refill is a COMPILE-TIME choice
  // (pf is std::true_type / std::false_type): a load behind any runtime condition - even one that is always true - makes
  // hipcc's waitcnt pass assume it may not have been issued, and the counted wait in front of the next LDS write
  // (vmcnt(7..4): leave the other register set's 4 loads in flight) degrades to vmcnt(3..0), i.e. the prefetch is drained
  // every K tile and each tile pays a full memory latency (seen in the ISA of the previous version of this loop).
  auto stage = [&](auto pf, u32x4 (&ra)[CA], u32x4 (&rb)[CB], int k_next) __attribute__((always_inline)) {
    lstore(ra, rb);
    __syncthreads();
    if constexpr (decltype(pf)::value) gload(ra, rb, k_next);
    mma();
    __syncthreads();
  };
  constexpr std::true_type PF{};
  constexpr std::false_type NOPF{};
  const int nt = K / BK;
  gload(ra0, rb0, 0);
  gload(ra1, rb1, nt > 1 ? BK : 0);          // unconditional (a one-tile problem re-reads tile 0): the loop header's counted
  int t = 0;                                 // wait is the minimum over all paths that reach it
  for (; t + 3 < nt; t += 2) {
    stage(PF, ra0, rb0, (t + 2) * BK);
    stage(PF, ra1, rb1, (t + 3) * BK);
  }
  if (nt - t == 3) {
    stage(PF, ra0, rb0, (t + 2) * BK);
    stage(NOPF, ra1, rb1, 0);
    stage(NOPF, ra0, rb0, 0);
  } else if (nt - t == 2) {
    stage(NOPF, ra0, rb0, 0);
    stage(NOPF, ra1, rb1, 0);
  } else {
    stage(NOPF, ra0, rb0, 0);
  }

  // lane (fr, fq) holds, for row tile i and column tile j, row m = ... + fr and 4 consecutive columns
  const bool vec = !(e.ldo & 7) && (!e.res || !(e.ldres & 3)) && (!e.out2 || !(e.ldo2 & 7)) && (!e.aux || !(e.ldaux & 7));
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / 2) + i * 16 + fr;
    if (m >= M) continue;
    if (pair) {
#pragma unroll
      for (int c = 0; c < TN / 2; ++c) {
        const int n = n0 + wn * (BN / 2) + 32 * c + 8 * fq;
        const float4 a0 = make_float4(acc[i][2 * c][0], acc[i][2 * c][1], acc[i][2 * c][2], acc[i][2 * c][3]);
        const float4 a1 = make_float4(acc[i][2 * c + 1][0], acc[i][2 * c + 1][1], acc[i][2 * c + 1][2], acc[i][2 * c + 1][3]);
        if (vec && n + 7 < N) {
          float4 p0, p1;
          float4 v0 = epi_math4(e, m, n, a0, &p0);
          float4 v1 = epi_math4(e, m, n + 4, a1, &p1);
          if ((e.act == LIDK_ACT_SWISH || e.act == LIDK_ACT_GELU) && e.out2) st16(e, (bf16*)e.out2 + (size_t)m * e.ldo2 + n, pack8(p0, p1));
          st16(e, (bf16*)e.out + (size_t)m * e.ldo + n, pack8(v0, v1));
        } else {
          epi_store4<bf16>(e, m, n, N, a0);
          epi_store4<bf16>(e, m, n + 4, N, a1);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + 16 * j + 4 * fq;
        epi_store4<bf16>(e, m, n, N, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
      }
    }
  }
}

// ------------------------------------------------------------------------------------ bf16 MFMA kernel, 128x128 tile
// For long-K large shapes only (WavLM's ffn-down GEMM and its data gradient: K = 3072, hundreds of 128x128 tiles): twice the MFMA
// work per byte staged through LDS.  Measured at M = 9536 (tools/gemm_bench_wavlm.py): 71 vs 82 us at K = 3072 (631 vs 547 TFLOP/s);
// at K = 768 / 1536 the 64x64 tiles' many small workgroups still hide latency better (68 vs 66 us, 130 vs 108 us with the two-output
// GELU epilogue), so the dispatch asks for K >= 2048.  Register plan for two waves per SIMD: 64
// accumulator registers, ONE set of 8 prefetch registers x 4, operands double-buffered in LDS (2 x 32 KB) so a K tile costs one
// barrier: issue the loads of tile t+1, multiply tile t from LDS buffer t&1, then park tile t+1 in the other buffer.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_nt_bf16_big_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda, int ldb, Epi e) {
  constexpr int BM = 128, BN = 128, TM = 4, TN = 4, CA = 4, CB = 4;
  const int n_tiles = N / BN;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
  bf16* As = reinterpret_cast<bf16*>(big_smem);                     // [2][BM * BK]
  bf16* Bs = As + 2 * BM * BK;                                      // [2][BN * BK]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = (tile / n_tiles) * BM, n0 = (tile % n_tiles) * BN;
  const bool pair = !e.out_f32;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 ra[CA], rb[CB];
  auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      ra[i] = *reinterpret_cast<const u32x4*>(A + (size_t)min(m0 + row, M - 1) * lda + k0 + kc);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      rb[i] = *reinterpret_cast<const u32x4*>(B + (size_t)(n0 + row) * ldb + k0 + kc);
    }
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    bf16* a = As + buf * BM * BK;
    bf16* b = Bs + buf * BN * BK;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int c = tid + i * 256, row = c >> 3, ch = c & 7;
      *reinterpret_cast<u32x4*>(&a[row * BK + ((ch ^ (row & 7)) << 3)]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int c = tid + i * 256, row = c >> 3, ch = c & 7;
      const int rho = pair ? (row & ~31) + (((row >> 2) & 1) << 4) + (((row >> 3) & 3) << 2) + (row & 3) : row;
      *reinterpret_cast<u32x4*>(&b[rho * BK + ((ch ^ (rho & 7)) << 3)]) = rb[i];
    }
  };
  auto mma = [&](int buf) __attribute__((always_inline)) {
    const bf16* a = As + buf * BM * BK;
    const bf16* b = Bs + buf * BN * BK;
#pragma unroll
    for (int kk = 0; kk < BK / 8; kk += 4) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(&a[(wm * 64 + i * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(&b[(wn * 64 + j * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  const int nt = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int t = 0; t + 1 < nt; ++t) {
    gload((t + 1) * BK);              // unconditional: in flight during the whole multiply of tile t
    mma(t & 1);
    lstore((t + 1) & 1);
    __syncthreads();
  }
  mma((nt - 1) & 1);

  const bool vec = !(e.ldo & 7) && (!e.res || !(e.ldres & 3)) && (!e.out2 || !(e.ldo2 & 7)) && (!e.aux || !(e.ldaux & 7));
#pragma clang loop unroll(full)
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= M) continue;
    if (pair) {
#pragma clang loop unroll(full)
      for (int c = 0; c < TN / 2; ++c) {
        const int n = n0 + wn * 64 + 32 * c + 8 * fq;
        const float4 a0 = make_float4(acc[i][2 * c][0], acc[i][2 * c][1], acc[i][2 * c][2], acc[i][2 * c][3]);
        const float4 a1 = make_float4(acc[i][2 * c + 1][0], acc[i][2 * c + 1][1], acc[i][2 * c + 1][2], acc[i][2 * c + 1][3]);
        if (vec) {
          float4 p0, p1;
          const float4 v0 = epi_math4(e, m, n, a0, &p0);
          const float4 v1 = epi_math4(e, m, n + 4, a1, &p1);
          if ((e.act == LIDK_ACT_SWISH || e.act == LIDK_ACT_GELU) && e.out2) st16(e, (bf16*)e.out2 + (size_t)m * e.ldo2 + n, pack8(p0, p1));
          st16(e, (bf16*)e.out + (size_t)m * e.ldo + n, pack8(v0, v1));
        } else {
          epi_store4<bf16>(e, m, n, N, a0);
          epi_store4<bf16>(e, m, n + 4, N, a1);
        }
      }
    } else {
#pragma clang loop unroll(full)
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * 64 + 16 * j + 4 * fq;
        epi_store4<bf16>(e, m, n, N, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
      }
    }
  }
}

// ------------------------------------------------------------------------------------ bf16 MFMA kernel, 128x128 tile, LDS-DMA
// Round 3.  The 128x128 kernel above stages its operands through registers (8 prefetch registers x 4, one barrier per K tile,
// two waves per SIMD by register count) and only pays for K >= 2048.  Here both operand tiles arrive by global_load_lds_dwordx4
// (no staging registers, no ds_write pass; csrc/ffn.hip explains the inline asm and the scalar-base addressing): 2 x 32 KB of
// LDS and ~110 VGPRs per workgroup, so two workgroups (8 waves) share a CU and one's MFMAs cover the other's barrier and DMA
// waits.  LDS images are lane-linear per wave instruction (8 rows of 128 B); the XOR swizzle of the fragment reads - and, for
// bf16 outputs, the row permutation that gives a lane 8 consecutive output columns - are applied on the SOURCE addresses.
// For the d = 768 / ffn 3072 shapes of the transformer backbones (K = 768 ... 3072, hundreds of 128-tiles).
__device__ __forceinline__ void gemm_glds16s(const void* sbase, unsigned voff, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_nt_bf16_dma_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda, int ldb, Epi e) {
  constexpr int BM = 128, BN = 128, TM = 4, TN = 4;
  typedef __attribute__((address_space(3))) void lds_v;
  const int n_tiles = N / BN;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  extern __shared__ __attribute__((aligned(16))) unsigned char dma_smem[];       // [2][A 16 KB | B 16 KB]
  const unsigned smem0 = (unsigned)(size_t)(lds_v*)dma_smem;
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
  const int m0 = (tile / n_tiles) * BM, n0 = (tile % n_tiles) * BN;
  const bool pair = !e.out_f32;
  // per-lane byte offsets of this wave's 4 + 4 wave instructions per K tile (tile-row 32 wid + 8 i + lane / 8, chunk lane % 8)
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 32 * wid + 8 * i + (lane >> 3), c = (lane & 7) ^ (row & 7);
    aoff[i] = (unsigned)(((size_t)(min(m0 + row, M - 1) - m0) * lda + c * 8) * 2);
    const int n = pair ? (row & ~31) + (((row >> 2) & 3) << 3) + (((row >> 4) & 1) << 2) + (row & 3) : row;   // LDS row -> column
    boff[i] = (unsigned)(((size_t)n * ldb + c * 8) * 2);
  }
  const bf16* abase = A + (size_t)m0 * lda;
  const bf16* bbase = B + (size_t)n0 * ldb;
  auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
    const unsigned dst = smem0 + buf * 32768 + (32 * wid) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) gemm_glds16s(abase + kt * BK, aoff[i], dst + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) gemm_glds16s(bbase + kt * BK, boff[i], dst + 16384 + i * 1024);
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nt = K / BK;
  stage(0, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tile t landed (this wave's pieces), barrier: everybody's; and
    __syncthreads();                                       // every wave is done reading the other buffer
    if (t + 1 < nt) stage(t + 1, (t + 1) & 1);
    const bf16* a = reinterpret_cast<const bf16*>(dma_smem + (t & 1) * 32768);
    const bf16* b = a + BM * BK;
#pragma unroll
    for (int kk = 0; kk < BK / 8; kk += 4) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(&a[(wm * 64 + i * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(&b[(wn * 64 + j * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }
  const bool vec = !(e.ldo & 7) && (!e.res || !(e.ldres & 3)) && (!e.out2 || !(e.ldo2 & 7)) && (!e.aux || !(e.ldaux & 7));
#pragma clang loop unroll(full)
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= M) continue;
    if (pair) {
#pragma clang loop unroll(full)
      for (int c = 0; c < TN / 2; ++c) {
        const int n = n0 + wn * 64 + 32 * c + 8 * fq;
        const float4 a0 = make_float4(acc[i][2 * c][0], acc[i][2 * c][1], acc[i][2 * c][2], acc[i][2 * c][3]);
        const float4 a1 = make_float4(acc[i][2 * c + 1][0], acc[i][2 * c + 1][1], acc[i][2 * c + 1][2], acc[i][2 * c + 1][3]);
        if (vec) {
          float4 p0, p1;
          const float4 v0 = epi_math4(e, m, n, a0, &p0);
          const float4 v1 = epi_math4(e, m, n + 4, a1, &p1);
          if ((e.act == LIDK_ACT_SWISH || e.act == LIDK_ACT_GELU) && e.out2) st16(e, (bf16*)e.out2 + (size_t)m * e.ldo2 + n, pack8(p0, p1));
          st16(e, (bf16*)e.out + (size_t)m * e.ldo + n, pack8(v0, v1));
        } else {
          epi_store4<bf16>(e, m, n, N, a0);
          epi_store4<bf16>(e, m, n + 4, N, a1);
        }
      }
    } else {
#pragma clang loop unroll(full)
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * 64 + 16 * j + 4 * fq;
        epi_store4<bf16>(e, m, n, N, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
      }
    }
  }
}

// ------------------------------------------------------------------------------------ bf16 MFMA kernel, 256x256 tile, LDS-DMA
// Round 4.  Through L2 a 128 x 128 tile moves one operand byte per 64 FLOP; at the ~21 B/clk a CU's LDS-DMA fills at that caps the
// kernel above near a third of the MFMA peak (798 TFLOP/s at K = 3072, 470 at the XLS-R shapes).  A 256 x 256 tile halves the
// bytes per FLOP: 8 waves as 2 x 4, a wave owns 128 x 64 of the output (32 accumulator tiles = 128 registers), K tiles of 64 in
// two 64 KB buffers, one workgroup per CU.  Same LDS images and source-side swizzle as the 128 x 128 kernel.
// PERSISTENT: one workgroup per CU walks tiles b, b + G, b + 2 G ... (XCD-aware order), because with one workgroup per CU nothing
// else hides a tile's epilogue and prologue - ablation at M 16000 N 3072 K 1024 (tools/gemm_time.py, LIDK_GEMM_DBG bits 32 / 64 /
// 128): 155 us per launch of which the K loops 82, the epilogues 67.  Here the first K tile of the NEXT tile is requested before
// the epilogue of this one, and the epilogue's stores drain under the next K loop.
// Stores go through LDS: straight from the accumulator layout a wave instruction writes 16 rows x 64 B (half cache lines); each
// wave transposes its 128 x 64 block through a private 4 KB (32 rows at a time) and stores whole 128 B lines, 8 lanes per row; the
// residual of an f32 output is added on the way out with the same full-line pattern.
// For the transformer backbones (d = 768 / 1024, ffn 3072 / 4096, conv stack): launches with >= LIDK_GEMM_DMA256 (default 200)
// tiles of 256 x 256.
// BN = 256: waves 2 x 4, a wave owns 128 x 64; BN = 128 (launches whose 256 x 256 tile count leaves a poorly filled last round):
// waves 4 x 2, a wave owns 64 x 64, K tiles of 32 + 16 KB.
template <int EPI_F32, int BN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_nt_bf16_dma256_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int K, int lda, int ldb,
                           int total_tiles, Epi e) {
  constexpr int BM = 256, NWN = BN / 64, NWM = 8 / NWN, RW = BM / NWM, TM = RW / 16, TN = 4, STAGE = (BM + BN) * BK * 2;
  constexpr int NBI = BN / 32;                               // B-tile wave instructions per loader and K tile
  typedef __attribute__((address_space(3))) void lds_v;
  const int n_tiles = N / BN;
  extern __shared__ __attribute__((aligned(16))) unsigned char dma_smem[];       // [2][A 32 KB | B 32 KB] + 8 x 4 KB epilogue staging
  const unsigned smem0 = (unsigned)(size_t)(lds_v*)dma_smem;
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid / NWN, wn = wid % NWN;
  constexpr bool pair = !EPI_F32;
  const int G = gridDim.x, slot = xcd_tile(blockIdx.x, G);
  const int nt = K / BK;
  // ROLES.  vmcnt retires a wave's loads AND stores in issue order, so a wave that has stored a tile cannot wait for the next
  // tile's operands without waiting for those stores as well.  Waves 0-3 therefore issue every LDS-DMA load (8 + 8 wave
  // instructions per K tile each) and no global store; waves 4-7 issue every global store (their own 128 x 64 block and, handed
  // over through LDS, the block of wave w - 4 above it) and never wait on vmcnt in the K loop: a tile's stores drain under the next
  // tile's K loop.
  const bool loader = wid < 4;
  // a loader's 8 + 8 wave instructions per K tile: tile-row 64 w + 8 i + lane / 8, 16-byte chunk (lane % 8) ^ (lane / 8).  One lane
  // offset per operand; the row step of instruction i rides in the scalar base (A: 8 i rows; B: the LDS-row -> output-column
  // permutation of the 8-consecutive-columns epilogue is affine in the bits of i), except in the last row tile of a ragged M,
  // whose rows are clamped per lane.
  const int arow0 = 64 * (wid & 3) + (lane >> 3);
  const unsigned cofs = (unsigned)(((lane & 7) ^ (lane >> 3)) * 16);
  const unsigned aoff0 = (unsigned)((size_t)arow0 * lda * 2) + cofs;
  const int brow_l = (BN / 4) * (wid & 3) + (lane >> 3);         // this loader's first LDS row of the B tile
  const int brow0 = pair ? (brow_l & ~31) + (((brow_l >> 2) & 3) << 3) + (((brow_l >> 4) & 1) << 2) + (brow_l & 3) : brow_l;
  const unsigned boff0 = (unsigned)((size_t)brow0 * ldb * 2) + cofs;
  const bf16* abase = A;
  const bf16* bbase = B;
  int mlim = 0;                                              // last valid row of the tile, relative to its first
  bool tail = false;
  auto set_tile = [&](int tile) __attribute__((always_inline)) {
    const int m0 = (tile / n_tiles) * BM, n0 = (tile % n_tiles) * BN;
    abase = A + (size_t)m0 * lda;
    bbase = B + (size_t)n0 * ldb;
    mlim = M - 1 - m0;
    tail = mlim < BM - 1;
  };
  auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
    if (loader) {
      const unsigned dst = smem0 + buf * STAGE + (64 * wid) * 128;
      const bf16* ak = abase + kt * BK;
      const bf16* bk = bbase + kt * BK;
      if (tail) {
        int r0 = arow0;
        asm volatile("" : "+v"(r0));                         // recomputed per call: hoisted out of the K loop these 8 offsets spill
#pragma unroll
        for (int i = 0; i < 8; ++i)
          gemm_glds16s(ak, (unsigned)((size_t)min(r0 + 8 * i, mlim) * lda * 2) + cofs, dst + i * 1024);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) gemm_glds16s(ak + (size_t)(8 * i) * lda, aoff0, dst + i * 1024);
      }
#pragma unroll
      for (int i = 0; i < NBI; ++i) {
        const int dn = pair ? 16 * (i & 1) + 4 * ((i >> 1) & 1) + 32 * (i >> 2) : 8 * i;
        gemm_glds16s(bk + (size_t)dn * ldb, boff0, smem0 + buf * STAGE + BM * BK * 2 + (BN / 4) * (wid & 3) * 128 + i * 1024);
      }
    }
  };
  unsigned char* wl = dma_smem + 2 * STAGE + wid * 4096;          // this wave's staging slice; wl - 4 * 4096: the partner's (waves 4-7)

  int tile = slot;
  if (tile >= total_tiles) return;
  set_tile(tile);
  stage(0, 0);
  while (true) {
    const int m0 = (tile / n_tiles) * BM, n0 = (tile % n_tiles) * BN;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nt; ++t) {
      if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K tile t landed (the loaders' pieces = all of it); barrier:
      __syncthreads();                                                // everybody sees it, and is done reading the other buffer
      if (t + 1 < nt && !((e.dbg & 32) && t > 0)) stage(t + 1, (t + 1) & 1);
      if (e.dbg & 64) continue;
      const bf16* a = reinterpret_cast<const bf16*>(dma_smem + (t & 1) * STAGE);
      const bf16* b = a + BM * BK;
#pragma unroll
      for (int kk = 0; kk < BK / 8; kk += 4) {
        bf16x8 bfr[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bfr[j] = *reinterpret_cast<const bf16x8*>(&b[(wn * 64 + j * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
        for (int ih = 0; ih < TM; ih += 4) {                 // A fragments four row tiles at a time: 32 instead of 48 fragment registers
          bf16x8 af[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            af[i] = *reinterpret_cast<const bf16x8*>(&a[(wm * RW + (ih + i) * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[ih + i][j], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
        }
      }
    }
    // the next tile's first K tile is requested now: it lands while this tile's epilogue runs
    const int next = tile + G;
    const bool more = next < total_tiles;
    if (more) {
      __syncthreads();                                       // every wave is done with both stage buffers
      set_tile(next);
      stage(0, 0);
    }
    if (e.dbg & 128) {                                       // ablation: no epilogue (accumulators kept alive)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(acc[i][j]));
    } else {
      // Stores through LDS, whole 128 B lines, 32 rows (bf16) / 16 rows (f32) of a wave's block at a time: every wave leaves the
      // slice in its 4 KB, barrier, waves 4-7 write out their own and wave (w - 4)'s, barrier.
      if (pair) {
        const bool two = (e.act == LIDK_ACT_SWISH || e.act == LIDK_ACT_GELU) && e.out2;
        // the bias of this lane's 16 columns once per tile, ahead of the first store: a load issued behind stores is not
        // returned before they are acknowledged
        float4 bz[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int n = n0 + wn * 64 + 32 * c + 8 * fq;
          bz[c][0] = e.bias ? load4(e.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
          bz[c][1] = e.bias ? load4(e.bias + n + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        Epi eb = e;
        eb.bias = nullptr;
#pragma clang loop unroll(full)
        for (int i2 = 0; i2 < TM / 2; ++i2) {
          uint4 pk[2][2], pp[2][2];
#pragma clang loop unroll(full)
          for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * i2 + ii, m = min(m0 + wm * RW + i * 16 + fr, M - 1);
#pragma clang loop unroll(full)
            for (int c = 0; c < TN / 2; ++c) {
              const int n = n0 + wn * 64 + 32 * c + 8 * fq;
              const float4 a0 = make_float4(acc[i][2 * c][0] + bz[c][0].x, acc[i][2 * c][1] + bz[c][0].y, acc[i][2 * c][2] + bz[c][0].z,
                                            acc[i][2 * c][3] + bz[c][0].w);
              const float4 a1 = make_float4(acc[i][2 * c + 1][0] + bz[c][1].x, acc[i][2 * c + 1][1] + bz[c][1].y,
                                            acc[i][2 * c + 1][2] + bz[c][1].z, acc[i][2 * c + 1][3] + bz[c][1].w);
              float4 p0 = a0, p1 = a1;
              const float4 v0 = epi_math4(eb, m, n, a0, &p0);
              const float4 v1 = epi_math4(eb, m, n + 4, a1, &p1);
              pk[ii][c] = pack8(v0, v1);
              pp[ii][c] = pack8(p0, p1);
            }
          }
#pragma clang loop unroll(full)
          for (int ps = 0; ps < 2; ++ps) {
            if (ps == 1 && !two) break;
#pragma clang loop unroll(full)
            for (int ii = 0; ii < 2; ++ii)
#pragma clang loop unroll(full)
              for (int c = 0; c < TN / 2; ++c) {
                const int row = ii * 16 + fr;
                *reinterpret_cast<uint4*>(wl + row * 128 + (((4 * c + fq) ^ (row & 7)) << 4)) = ps ? pp[ii][c] : pk[ii][c];
              }
            __syncthreads();
            if (!loader) {
              bf16* dst = ps ? (bf16*)e.out2 : (bf16*)e.out;
              const int ldd = ps ? e.ldo2 : e.ldo;
#pragma unroll
              for (int own = 1; own >= 0; --own) {             // own = 1: this wave's rows; 0: the partner's (wave w - 4, same columns)
                const unsigned char* src = wl - (own ? 0 : 4 * 4096);
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                  const int row = it * 8 + (lane >> 3), m = m0 + (own ? wm : wm - NWM / 2) * RW + i2 * 32 + row;
                  const uint4 v = *reinterpret_cast<const uint4*>(src + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
                  if (m < M) st16(e, dst + (size_t)m * ldd + n0 + wn * 64 + (lane & 7) * 8, v);
                }
              }
            }
            __syncthreads();
          }
          __builtin_amdgcn_sched_barrier(0);                 // one slice at a time: the aux loads of all would not fit the registers
        }
      } else {
        Epi e2 = e;
        e2.res = nullptr;
#pragma clang loop unroll(full)
        for (int i = 0; i < TM; ++i) {                       // 16 rows x 256 B at a time
          const int m_ = min(m0 + wm * RW + i * 16 + fr, M - 1);
#pragma clang loop unroll(full)
          for (int j = 0; j < TN; ++j) {
            float4 p0;
            const float4 v = epi_math4(e2, m_, n0 + wn * 64 + 16 * j + 4 * fq,
                                       make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]), &p0);
            *reinterpret_cast<float4*>(wl + fr * 256 + (((4 * j + fq) ^ fr) << 4)) = v;
          }
          __syncthreads();
          if (!loader) {
            // all 8 residual loads of the slice first, then its 8 stores (a load behind a store waits for the store's acknowledgement)
            float4 rr[2][4];
#pragma unroll
            for (int own = 1; own >= 0; --own)
#pragma unroll
              for (int it = 0; it < 4; ++it) {
                const int row = it * 4 + (lane >> 4), m = min(m0 + (own ? wm : wm - NWM / 2) * RW + i * 16 + row, M - 1), n = n0 + wn * 64 + (lane & 15) * 4;
                rr[own][it] = e.res ? load4(e.res + (size_t)m * e.ldres + n) : make_float4(0.f, 0.f, 0.f, 0.f);
              }
#pragma unroll
            for (int own = 1; own >= 0; --own) {
              const unsigned char* src = wl - (own ? 0 : 4 * 4096);
#pragma unroll
              for (int it = 0; it < 4; ++it) {
                const int row = it * 4 + (lane >> 4), m = m0 + (own ? wm : wm - NWM / 2) * RW + i * 16 + row, n = n0 + wn * 64 + (lane & 15) * 4;
                float4 v = *reinterpret_cast<const float4*>(src + row * 256 + (((lane & 15) ^ row) << 4));
                const float4 r = rr[own][it];
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                if (m < M) store4((float*)e.out + (size_t)m * e.ldo + n, v);
              }
            }
          }
          __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (!more) break;
    tile = next;
  }
}

// ------------------------------------------------------------------------------------ bf16 MFMA kernel, pipelined tiles
// For the wide K = 256 GEMMs (ff up-projection, its data gradient, QKV, pointwise conv 1) the output stores are half of a
// launch (ablation in DESIGN.md section 5) and, in gemm_nt_bf16_direct_kernel, purely additive: a wave keeps its slot until
// its stores have drained, so stores never overlap the next tile's operand fetch.  Here every workgroup walks TPB tiles and
// the work of consecutive tiles is interleaved in straight-line code:
//   * the epilogue of tile j-1 (its accumulators are kept in 16 spare registers) is issued in two slices inside the first
//     two K stages of tile j;
//   * the last two K stages of tile j issue the operand loads of tile j+1's first two K tiles.
// Everything in the steady state is unconditional (M % 64 == 0, N % 64 == 0, K == 256, epilogue chosen at compile time,
// tile indices past the end are clamped to the last tile, which is then simply computed twice with identical stores), so
// hipcc can keep counted vmcnt waits instead of draining the stores.
// PIPE_BN_SUMS: the output is ds, the gradient at a BatchNorm + Swish output (the conv module's second pointwise convolution run
// backwards, lid/conformer.py:197-199); besides storing it the epilogue forms dz = ds * swish'(xhat * gamma + beta), xhat = (c -
// mean) * rstd from the saved BatchNorm input c (aux) and leaves per-channel partial sums (sum dz | sum dz * xhat) over the wave's
// 32 rows: bn_partial[(m_tile * 2 + wm)][2][N] - what lidk_bn_swish_bwd_reduce computes in a launch of its own that re-reads ds.
enum { PIPE_PLAIN = 0, PIPE_BIAS = 1, PIPE_BIAS_SWISH_PRE = 2, PIPE_SWISH_GRAD = 3, PIPE_BN_SUMS = 4 };

template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4)))
gemm_nt_bf16_pipe_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int lda, int ldb, int tiles,
                         int G, int TPB, Epi e) {
  constexpr int K = 256;
  __shared__ __attribute__((aligned(16))) bf16 As[64 * BK];
  __shared__ __attribute__((aligned(16))) bf16 Bs[64 * BK];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int n_tiles = N / 64, span = G * TPB;

  auto coords = [&](int j, int& m0, int& n0) __attribute__((always_inline)) {
    int t = xcd_tile(blockIdx.x + j * G, span);          // G % 8 == 0: the XCD of launch index b + j*G is that of b
    t = min(t, tiles - 1);
    m0 = (t / n_tiles) * 64; n0 = (t % n_tiles) * 64;
  };
  u32x4 ra0[2], rb0[2], ra1[2], rb1[2];
  auto gload = [&](u32x4 (&ra)[2], u32x4 (&rb)[2], int m0, int n0, int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      ra[i] = *reinterpret_cast<const u32x4*>(A + (size_t)(m0 + row) * lda + k0 + kc);
      rb[i] = *reinterpret_cast<const u32x4*>(B + (size_t)(n0 + row) * ldb + k0 + kc);
    }
  };
  auto lstore = [&](const u32x4 (&ra)[2], const u32x4 (&rb)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int c = tid + i * 256, row = c >> 3, ch = c & 7;
      *reinterpret_cast<u32x4*>(&As[row * BK + ((ch ^ (row & 7)) << 3)]) = ra[i];
      int rho = (row & ~31) + (((row >> 2) & 1) << 4) + (((row >> 3) & 3) << 2) + (row & 3);     // bf16 outputs: pair layout
      *reinterpret_cast<u32x4*>(&Bs[rho * BK + ((ch ^ (rho & 7)) << 3)]) = rb[i];
    }
  };
  f32x4 acc[2][2], prv[2][2];
  auto mma = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < BK / 8; kk += 4) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * 32 + i * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 32 + j * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  float bns0[8], bns1[8];                                  // PIPE_BN_SUMS: this lane's two rows of the tile, 8 channels
  // rows i*16 + fr of this wave's 32x32 block of the tile at (m0, n0): 8 consecutive columns per lane
  auto epi = [&](const f32x4 (&a)[2][2], int i, int m0, int n0) __attribute__((always_inline)) {
    const int m = m0 + wm * 32 + i * 16 + fr, n = n0 + wn * 32 + 8 * fq;
    float4 v0 = make_float4(a[i][0][0], a[i][0][1], a[i][0][2], a[i][0][3]);
    float4 v1 = make_float4(a[i][1][0], a[i][1][1], a[i][1][2], a[i][1][3]);
    if (MODE == PIPE_BIAS || MODE == PIPE_BIAS_SWISH_PRE) {
      float4 b0 = load4(e.bias + n), b1 = load4(e.bias + n + 4);
      v0.x += b0.x; v0.y += b0.y; v0.z += b0.z; v0.w += b0.w; v1.x += b1.x; v1.y += b1.y; v1.z += b1.z; v1.w += b1.w;
    }
    if (MODE == PIPE_BIAS_SWISH_PRE) {
      *reinterpret_cast<uint4*>((bf16*)e.out2 + (size_t)m * e.ldo2 + n) = pack8(v0, v1);
      v0.x *= sigmoidf_(v0.x); v0.y *= sigmoidf_(v0.y); v0.z *= sigmoidf_(v0.z); v0.w *= sigmoidf_(v0.w);
      v1.x *= sigmoidf_(v1.x); v1.y *= sigmoidf_(v1.y); v1.z *= sigmoidf_(v1.z); v1.w *= sigmoidf_(v1.w);
    }
    if (MODE == PIPE_BN_SUMS) {
      const uint4 pk = pack8(v0, v1);                    // the statistics are those of the ROUNDED ds the later kernels read
      *reinterpret_cast<uint4*>((bf16*)e.out + (size_t)m * e.ldo + n) = pk;
      const f32x8_t dsr = unpack8(pk);
      const f32x8_t cv = load8((const bf16*)e.aux + (size_t)m * e.ldaux + n);
      const float dv[8] = {dsr.lo.x, dsr.lo.y, dsr.lo.z, dsr.lo.w, dsr.hi.x, dsr.hi.y, dsr.hi.z, dsr.hi.w};
      const float cc[8] = {cv.lo.x, cv.lo.y, cv.lo.z, cv.lo.w, cv.hi.x, cv.hi.y, cv.hi.z, cv.hi.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (cc[j] - e.bn_mean[n + j]) * e.bn_rstd[n + j];
        const float z = xh * e.bn_gamma[n + j] + e.bn_beta[n + j];
        const float sg = sigmoidf_(z);
        const float dz = dv[j] * (sg * (1.f + z * (1.f - sg)));
        if (i == 0) { bns0[j] = dz; bns1[j] = dz * xh; } else { bns0[j] += dz; bns1[j] = fmaf(dz, xh, bns1[j]); }
      }
      if (i == 1) {                                      // both row groups of the tile are in: 16 row-lanes -> one
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { bns0[j] += __shfl_xor(bns0[j], o, 64); bns1[j] += __shfl_xor(bns1[j], o, 64); }
        }
        if (fr == 0) {
          float* pr = e.bn_partial + ((size_t)(m0 / 64) * 2 + wm) * 2 * N + n;
          store4(pr, make_float4(bns0[0], bns0[1], bns0[2], bns0[3])); store4(pr + 4, make_float4(bns0[4], bns0[5], bns0[6], bns0[7]));
          store4(pr + N, make_float4(bns1[0], bns1[1], bns1[2], bns1[3])); store4(pr + N + 4, make_float4(bns1[4], bns1[5], bns1[6], bns1[7]));
        }
      }
      return;
    }
    if (MODE == PIPE_SWISH_GRAD) {
      const bf16* ap = (const bf16*)e.aux + (size_t)m * e.ldaux + n;
      float4 a0 = load4(ap), a1 = load4(ap + 4);
      float s;
      s = sigmoidf_(a0.x); v0.x *= s * (1.f + a0.x * (1.f - s)); s = sigmoidf_(a0.y); v0.y *= s * (1.f + a0.y * (1.f - s));
      s = sigmoidf_(a0.z); v0.z *= s * (1.f + a0.z * (1.f - s)); s = sigmoidf_(a0.w); v0.w *= s * (1.f + a0.w * (1.f - s));
      s = sigmoidf_(a1.x); v1.x *= s * (1.f + a1.x * (1.f - s)); s = sigmoidf_(a1.y); v1.y *= s * (1.f + a1.y * (1.f - s));
      s = sigmoidf_(a1.z); v1.z *= s * (1.f + a1.z * (1.f - s)); s = sigmoidf_(a1.w); v1.w *= s * (1.f + a1.w * (1.f - s));
    }
    *reinterpret_cast<uint4*>((bf16*)e.out + (size_t)m * e.ldo + n) = pack8(v0, v1);
  };
  auto zero = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto keep = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) prv[i][j] = acc[i][j];
  };
#define PIPE_STAGE(RA, RB, LOADS, EPI)   \
  do { lstore(RA, RB); __syncthreads(); LOADS; mma(); EPI; __syncthreads(); } while (0)

  int m0, n0, mn, nn, mp = 0, np = 0;
  coords(0, m0, n0);
  gload(ra0, rb0, m0, n0, 0);
  gload(ra1, rb1, m0, n0, BK);
  // ---- first tile: nothing to store yet
  coords(1, mn, nn);
  zero();
  PIPE_STAGE(ra0, rb0, gload(ra0, rb0, m0, n0, 2 * BK), (void)0);
  PIPE_STAGE(ra1, rb1, gload(ra1, rb1, m0, n0, 3 * BK), (void)0);
  PIPE_STAGE(ra0, rb0, gload(ra0, rb0, mn, nn, 0), (void)0);
  PIPE_STAGE(ra1, rb1, gload(ra1, rb1, mn, nn, BK), (void)0);
  keep(); mp = m0; np = n0; m0 = mn; n0 = nn;
  // ---- middle tiles
  for (int j = 1; j + 1 < TPB; ++j) {
    coords(j + 1, mn, nn);
    zero();
    PIPE_STAGE(ra0, rb0, gload(ra0, rb0, m0, n0, 2 * BK), epi(prv, 0, mp, np));
    PIPE_STAGE(ra1, rb1, gload(ra1, rb1, m0, n0, 3 * BK), epi(prv, 1, mp, np));
    PIPE_STAGE(ra0, rb0, gload(ra0, rb0, mn, nn, 0), (void)0);
    PIPE_STAGE(ra1, rb1, gload(ra1, rb1, mn, nn, BK), (void)0);
    keep(); mp = m0; np = n0; m0 = mn; n0 = nn;
  }
  // ---- last tile (TPB >= 2): no further prefetch, then its own epilogue
  zero();
  PIPE_STAGE(ra0, rb0, gload(ra0, rb0, m0, n0, 2 * BK), epi(prv, 0, mp, np));
  PIPE_STAGE(ra1, rb1, gload(ra1, rb1, m0, n0, 3 * BK), epi(prv, 1, mp, np));
  PIPE_STAGE(ra0, rb0, (void)0, (void)0);
  PIPE_STAGE(ra1, rb1, (void)0, (void)0);
  epi(acc, 0, m0, n0);
  epi(acc, 1, m0, n0);
#undef PIPE_STAGE
}

// The same stage machinery for K = 64 KS (KS even, 8 ... 16: the K = 512 / 768 / 1024 data gradients of the QKV projection and the
// first pointwise convolution).  gemm_nt_bf16_direct_kernel reads every operand fragment twice from L2 (its 2 x 2 waves share rows
// and columns but not registers): at K = 768, N = 256 that is 232 MB through the texture path of 256 CUs, ~30 us for 3.8 GFLOP;
// staged through LDS each operand byte enters the CU once.  TPB = 1: one tile per workgroup (no cross-tile overlap).
template <int MODE, int KS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4)))
gemm_nt_bf16_pipek_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, int M, int N, int lda, int ldb, int tiles,
                         int G, int TPB, Epi e) {
  __shared__ __attribute__((aligned(16))) bf16 As[64 * BK];
  __shared__ __attribute__((aligned(16))) bf16 Bs[64 * BK];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int n_tiles = N / 64, span = G * TPB;

  auto coords = [&](int j, int& m0, int& n0) __attribute__((always_inline)) {
    int t = xcd_tile(blockIdx.x + j * G, span);          // G % 8 == 0: the XCD of launch index b + j*G is that of b
    t = min(t, tiles - 1);
    m0 = (t / n_tiles) * 64; n0 = (t % n_tiles) * 64;
  };
  u32x4 ra0[2], rb0[2], ra1[2], rb1[2];
  auto gload = [&](u32x4 (&ra)[2], u32x4 (&rb)[2], int m0, int n0, int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = (c & 7) * 8;
      ra[i] = *reinterpret_cast<const u32x4*>(A + (size_t)(m0 + row) * lda + k0 + kc);
      rb[i] = *reinterpret_cast<const u32x4*>(B + (size_t)(n0 + row) * ldb + k0 + kc);
    }
  };
  auto lstore = [&](const u32x4 (&ra)[2], const u32x4 (&rb)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int c = tid + i * 256, row = c >> 3, ch = c & 7;
      *reinterpret_cast<u32x4*>(&As[row * BK + ((ch ^ (row & 7)) << 3)]) = ra[i];
      int rho = (row & ~31) + (((row >> 2) & 1) << 4) + (((row >> 3) & 3) << 2) + (row & 3);     // bf16 outputs: pair layout
      *reinterpret_cast<u32x4*>(&Bs[rho * BK + ((ch ^ (rho & 7)) << 3)]) = rb[i];
    }
  };
  f32x4 acc[2][2], prv[2][2];
  auto mma = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < BK / 8; kk += 4) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * 32 + i * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 32 + j * 16 + fr) * BK + (((kk + fq) ^ (fr & 7)) << 3)]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  // rows i*16 + fr of this wave's 32x32 block of the tile at (m0, n0): 8 consecutive columns per lane
  auto epi = [&](const f32x4 (&a)[2][2], int i, int m0, int n0) __attribute__((always_inline)) {
    const int m = m0 + wm * 32 + i * 16 + fr, n = n0 + wn * 32 + 8 * fq;
    float4 v0 = make_float4(a[i][0][0], a[i][0][1], a[i][0][2], a[i][0][3]);
    float4 v1 = make_float4(a[i][1][0], a[i][1][1], a[i][1][2], a[i][1][3]);
    if (MODE == PIPE_BIAS || MODE == PIPE_BIAS_SWISH_PRE) {
      float4 b0 = load4(e.bias + n), b1 = load4(e.bias + n + 4);
      v0.x += b0.x; v0.y += b0.y; v0.z += b0.z; v0.w += b0.w; v1.x += b1.x; v1.y += b1.y; v1.z += b1.z; v1.w += b1.w;
    }
    if (MODE == PIPE_BIAS_SWISH_PRE) {
      *reinterpret_cast<uint4*>((bf16*)e.out2 + (size_t)m * e.ldo2 + n) = pack8(v0, v1);
      v0.x *= sigmoidf_(v0.x); v0.y *= sigmoidf_(v0.y); v0.z *= sigmoidf_(v0.z); v0.w *= sigmoidf_(v0.w);
      v1.x *= sigmoidf_(v1.x); v1.y *= sigmoidf_(v1.y); v1.z *= sigmoidf_(v1.z); v1.w *= sigmoidf_(v1.w);
    }
    if (MODE == PIPE_SWISH_GRAD) {
      const bf16* ap = (const bf16*)e.aux + (size_t)m * e.ldaux + n;
      float4 a0 = load4(ap), a1 = load4(ap + 4);
      float s;
      s = sigmoidf_(a0.x); v0.x *= s * (1.f + a0.x * (1.f - s)); s = sigmoidf_(a0.y); v0.y *= s * (1.f + a0.y * (1.f - s));
      s = sigmoidf_(a0.z); v0.z *= s * (1.f + a0.z * (1.f - s)); s = sigmoidf_(a0.w); v0.w *= s * (1.f + a0.w * (1.f - s));
      s = sigmoidf_(a1.x); v1.x *= s * (1.f + a1.x * (1.f - s)); s = sigmoidf_(a1.y); v1.y *= s * (1.f + a1.y * (1.f - s));
      s = sigmoidf_(a1.z); v1.z *= s * (1.f + a1.z * (1.f - s)); s = sigmoidf_(a1.w); v1.w *= s * (1.f + a1.w * (1.f - s));
    }
    *reinterpret_cast<uint4*>((bf16*)e.out + (size_t)m * e.ldo + n) = pack8(v0, v1);
  };
  auto zero = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  auto keep = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) prv[i][j] = acc[i][j];
  };
  int m0, n0, mn = 0, nn = 0, mp = 0, np = 0;
  // one tile = KS stages in two alternating register sets; stage s refills its set with stage s + 2 of the same tile or, in the
  // last two stages, with the first two stages of the next tile; the previous tile's epilogue rides in stages 0 and 1.  After
  // full unrolling every condition below is a compile-time constant (no refill sits behind a runtime branch).
  auto stage = [&](u32x4 (&ra)[2], u32x4 (&rb)[2], int s_, auto first, auto last) __attribute__((always_inline)) {
    lstore(ra, rb);
    __syncthreads();
    if (s_ + 2 < KS) gload(ra, rb, m0, n0, (s_ + 2) * BK);
    else if constexpr (!decltype(last)::value) gload(ra, rb, mn, nn, (s_ + 2 - KS) * BK);
    mma();
    if constexpr (!decltype(first)::value) { if (s_ < 2) epi(prv, s_, mp, np); }
    __syncthreads();
  };
  auto tile = [&](auto first, auto last) __attribute__((always_inline)) {
    zero();
#pragma unroll
    for (int sp = 0; sp < KS; sp += 2) {
      stage(ra0, rb0, sp, first, last);
      stage(ra1, rb1, sp + 1, first, last);
    }
  };
  constexpr std::true_type YES{};
  constexpr std::false_type NO{};
  coords(0, m0, n0);
  gload(ra0, rb0, m0, n0, 0);
  gload(ra1, rb1, m0, n0, BK);
  if (TPB == 1) {                 // block-uniform
    tile(YES, YES);
    epi(acc, 0, m0, n0);
    epi(acc, 1, m0, n0);
    return;
  }
  coords(1, mn, nn);
  tile(YES, NO);
  keep(); mp = m0; np = n0; m0 = mn; n0 = nn;
  for (int j = 1; j + 1 < TPB; ++j) {
    coords(j + 1, mn, nn);
    tile(NO, NO);
    keep(); mp = m0; np = n0; m0 = mn; n0 = nn;
  }
  tile(NO, YES);
  epi(acc, 0, m0, n0);
  epi(acc, 1, m0, n0);
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
// Kernel-family knobs of lidk_gemm_nt: read from the environment ONCE (LIDK_GEMM_PIPEK, LIDK_GEMM_DMA, LIDK_GEMM_DMA_TILES) and
// changed afterwards only through lidk_gemm_option - tests and the micro-benchmarks flip them inside one process.
static int g_opt_pipek = -1, g_opt_dma = -1;
static long g_opt_dma_tiles = -1, g_opt_dma256 = -1, g_opt_dma256_bn = -1;
static void gemm_opts_init() {
  if (g_opt_pipek < 0) { const char* v = getenv("LIDK_GEMM_PIPEK"); g_opt_pipek = v ? atoi(v) : 1; }
  if (g_opt_dma < 0) { const char* v = getenv("LIDK_GEMM_DMA"); g_opt_dma = v ? atoi(v) : 1024; }
  if (g_opt_dma_tiles < 0) { const char* v = getenv("LIDK_GEMM_DMA_TILES"); g_opt_dma_tiles = v ? atol(v) : 384; }
  if (g_opt_dma256 < 0) { const char* v = getenv("LIDK_GEMM_DMA256"); g_opt_dma256 = v ? atol(v) : 150; }
  if (g_opt_dma256_bn < 0) { const char* v = getenv("LIDK_GEMM_DMA256_BN"); g_opt_dma256_bn = v ? atol(v) : 0; }
}
extern "C" int lidk_gemm_option(const char* name, long value) {
  if (!name) return LIDK_ERR_ARG;
  gemm_opts_init();
  if (!strcmp(name, "LIDK_GEMM_PIPEK")) g_opt_pipek = (int)value;
  else if (!strcmp(name, "LIDK_GEMM_DMA")) g_opt_dma = (int)value;
  else if (!strcmp(name, "LIDK_GEMM_DMA_TILES")) g_opt_dma_tiles = value;
  else if (!strcmp(name, "LIDK_GEMM_DMA256")) g_opt_dma256 = value;
  else if (!strcmp(name, "LIDK_GEMM_DMA256_BN")) g_opt_dma256_bn = value;
  else return LIDK_ERR_ARG;
  return LIDK_OK;                                        // (a negative value: re-read the environment on the next launch)
}

// out [M][N] bf16 = A [M][256] . B [N][256]^T with the BatchNorm + Swish backward statistics of the output in the epilogue
// (PIPE_BN_SUMS above).  partial: (M / 64) * 2 rows of (sum dz | sum dz * xhat) [2 N] f32, *nparts receives that row count.
extern "C" int lidk_gemm_nt_bn_sums(const lidk_gemm_args* g, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, float* partial, int* nparts, int dtype, void* stream) {
  if (!g || !g->A || !g->B || !g->out || !g->aux || !mean || !rstd || !gamma || !beta || !partial || !nparts) return LIDK_ERR_ARG;
  const long tiles = (long)(g->M / 64) * (g->N / 64);
  if (dtype != LIDK_BF16 || g->K != 256 || (g->M & 63) || (g->N & 63) || g->out_f32 || g->res || g->bias || g->out2 ||
      g->act != LIDK_ACT_NONE || g->alpha != 1.0f || (g->ldo & 7) || (g->ldaux & 7) || (g->lda & 7) || (g->ldb & 7) || g->ldb < g->K ||
      tiles <= 1024)
    return LIDK_ERR_UNSUPPORTED;
  Epi e{0, nullptr, LIDK_ACT_NONE, 1.0f, nullptr, 0, g->out, g->ldo, 0, nullptr, 0, g->aux, g->ldaux, 0, mean, rstd, gamma, beta, partial};
  const int tpb = 2, G = cdiv(cdiv((int)tiles, tpb), 8) * 8;
  gemm_nt_bf16_pipe_kernel<PIPE_BN_SUMS><<<G, 256, 0, as_stream(stream)>>>((const bf16*)g->A, (const bf16*)g->B, g->M, g->N, g->lda,
                                                                          g->ldb, (int)tiles, G, tpb, e);
  *nparts = (g->M / 64) * 2;
  return launch_status();
}

extern "C" int lidk_gemm_nt(const lidk_gemm_args* g, int dtype, void* stream) {
  if (!g || !g->A || !g->B || !g->out || g->M <= 0 || g->N <= 0 || g->K <= 0) return LIDK_ERR_ARG;
  if ((g->K & 7) || (g->lda & 7) || (g->ldb & 7) || g->ldb < g->K || g->ldo < g->N) return LIDK_ERR_ARG;
  if ((g->act == LIDK_ACT_SWISH_GRAD || g->act == LIDK_ACT_GELU_GRAD) && !g->aux) return LIDK_ERR_ARG;
  int splitk = g->splitk > 1 ? g->splitk : 1;
  if (splitk > 1 && (g->bias || g->res || g->act != LIDK_ACT_NONE || !g->out_f32)) return LIDK_ERR_ARG;
  static const int dbg = getenv("LIDK_GEMM_DBG") ? atoi(getenv("LIDK_GEMM_DBG")) : 0;
  Epi e{dbg, g->bias, g->act, g->alpha, g->res, g->ldres, g->out, g->ldo, g->out_f32, g->out2, g->ldo2, g->aux, g->ldaux,
        splitk > 1 ? 1 : 0};
  hipStream_t s = as_stream(stream);
  if (dtype == LIDK_BF16) {
    int kchunk = cdiv(cdiv(g->K, splitk), BK) * BK;
    splitk = cdiv(g->K, kchunk);
    // tile choice: measured on MI355X (tools/gemm_bench.py), the 64x64 tile wins on every shape of this model (K = 256..1024:
    // only 4..16 K-tiles, so many small workgroups per CU hide the load latency better than one big tile does); the 128x128
    // variant is kept for long-K problems and can be forced with LIDK_GEMM_TILE=128.
    static const int direct = getenv("LIDK_GEMM_DIRECT") ? atoi(getenv("LIDK_GEMM_DIRECT")) : 1;
    static const int pipe = getenv("LIDK_GEMM_PIPE") ? atoi(getenv("LIDK_GEMM_PIPE")) : 1;
    if (pipe && splitk == 1 && !dbg && g->K == 256 && !(g->M & 63) && !(g->N & 63) && !g->out_f32 && !g->res && g->alpha == 1.0f &&
        !(g->ldo & 7) && (!g->out2 || !(g->ldo2 & 7)) && (!g->aux || !(g->ldaux & 7))) {
      int mode = -1;
      if (g->act == LIDK_ACT_NONE) mode = g->bias ? PIPE_BIAS : PIPE_PLAIN;
      else if (g->act == LIDK_ACT_SWISH && g->bias && g->out2) mode = PIPE_BIAS_SWISH_PRE;
      else if (g->act == LIDK_ACT_SWISH_GRAD && !g->bias) mode = PIPE_SWISH_GRAD;
      const int tiles = (g->M / 64) * (g->N / 64);
      const int tpb = pipe > 1 ? pipe : (tiles > 1024 ? 2 : 1);     // tiles per workgroup (measured: 2 beats 3 and 4 on every shape)
      if (mode >= 0 && tpb >= 2) {
        const int G = cdiv(cdiv(tiles, tpb), 8) * 8;
#define LIDK_PIPE_LAUNCH(MODE_)                                                                                              \
  gemm_nt_bf16_pipe_kernel<MODE_><<<G, 256, 0, s>>>((const bf16*)g->A, (const bf16*)g->B, g->M, g->N, g->lda, g->ldb, tiles, \
                                                    G, tpb, e)
        if (mode == PIPE_PLAIN) LIDK_PIPE_LAUNCH(PIPE_PLAIN);
        else if (mode == PIPE_BIAS) LIDK_PIPE_LAUNCH(PIPE_BIAS);
        else if (mode == PIPE_BIAS_SWISH_PRE) LIDK_PIPE_LAUNCH(PIPE_BIAS_SWISH_PRE);
        else LIDK_PIPE_LAUNCH(PIPE_SWISH_GRAD);
#undef LIDK_PIPE_LAUNCH
        return launch_status();
      }
    }
    // LDS-DMA 256x256 persistent kernel (comment at the kernel): N % 256 == 0, K % 64 == 0, K >= 512, >= LIDK_GEMM_DMA256 tiles
    // (default 150, 0 = never, 1 = always) whose last round fills >= 65 % of the CUs (or >= 3 rounds).  tools/gemm_shapes_ab.py
    // (profiles/r04/gemm_shapes_ab.txt), default dispatch -> this kernel forced, us per launch: 189 tiles (0.74 rounds; M 16000 N 768)
    // K 768 50 -> 46, K 3072 142 -> 101; 342 tiles (1.34) 72 -> 58; 567 (2.21) 111 -> 85; 756 (2.95) 90 -> 88; but 114 tiles (0.45)
    // 30 -> 42 / 62 -> 94 and 282 tiles (1.10 rounds: a second round for 26 tiles) 84 -> 87 / 193 -> 207: that is the fill rule.
    // XLS-R width (M 16000; tools/gemm_bench_wavlm.py, before the kernel -> with it): qkv 213 -> 104, out 69 -> 59, fc1 + GELU 327 ->
    // 186, fc2 204 -> 136; conv stack layers 1 / 2 838 -> 749 / 411 -> 349.
    {
      gemm_opts_init();
      static int n_cu = 0;
      if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
      }
      // tile width: 256 columns.  The 256 x 128 form (LIDK_GEMM_DMA256_BN = 128; waves 4 x 2) is built and tested but is not chosen
      // automatically: on the d = 768 shapes it loses to the 256 x 256 form or to the older kernels at every M tried
      // (profiles/r04/gemm_shapes_ab2.txt: e.g. M 9536 N 768 K 3072 58 us without the persistent kernel, 87 / 63 with 256 / 128 columns).
      const int force_bn = (int)g_opt_dma256_bn;
      const bool vec256 = !(g->ldo & 7) && (!g->res || !(g->ldres & 3)) && (!g->out2 || !(g->ldo2 & 7)) && (!g->aux || !(g->ldaux & 7));
      const bool shape_ok = direct && splitk == 1 && g_opt_dma256 > 0 && (g->K & 63) == 0 && g->K >= 512 && vec256 &&
                            (size_t)256 * g->lda * 2 < (1ull << 31) && (size_t)256 * g->ldb * 2 < (1ull << 31);
      int bn = 0;
      long tiles = 0;
      for (int cand = 256; cand >= 128 && shape_ok && !bn; cand >>= 1) {
        if ((g->N % cand) || (force_bn ? force_bn != cand : cand != 256)) continue;
        const long t = (long)cdiv(g->M, 256) * (g->N / cand), rounds = (t + n_cu - 1) / n_cu;
        const bool filled = t * 100 >= rounds * n_cu * 65 || rounds >= 3 || g_opt_dma256 == 1;
        if (t >= g_opt_dma256 && filled) { bn = cand; tiles = t; }
      }
      if (bn) {
        static bool attr256 = false;
        constexpr int lds256 = 2 * (256 + 256) * BK * 2 + 8 * 4096, lds128 = 2 * (256 + 128) * BK * 2 + 8 * 4096;
        if (!attr256) {
          (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_dma256_kernel<0, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
          (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_dma256_kernel<1, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, lds256);
          (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_dma256_kernel<0, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds128);
          (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_dma256_kernel<1, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds128);
          attr256 = true;
        }
        const int G = (int)(tiles < n_cu ? tiles : n_cu);    // persistent: one workgroup per CU
#define LIDK_DMA256_LAUNCH(F32_, BN_, LDS_)                                                                                      \
  gemm_nt_bf16_dma256_kernel<F32_, BN_><<<G, 512, LDS_, s>>>((const bf16*)g->A, (const bf16*)g->B, g->M, g->N, g->K, g->lda, g->ldb, \
                                                             (int)tiles, e)
        if (bn == 256) { if (g->out_f32) LIDK_DMA256_LAUNCH(1, 256, lds256); else LIDK_DMA256_LAUNCH(0, 256, lds256); }
        else { if (g->out_f32) LIDK_DMA256_LAUNCH(1, 128, lds128); else LIDK_DMA256_LAUNCH(0, 128, lds128); }
#undef LIDK_DMA256_LAUNCH
        return launch_status();
      }
    }
    // K = 512 / 768 / 1024 with plain / bias epilogues: the K-generic pipelined kernel (LIDK_GEMM_PIPEK = tiles per workgroup, default 1,
    // 0 = off).  tools/gemm_longk_bench.py, us per launch, direct kernel / 1 / 2 / 3 tiles per workgroup: M 9664 N 256 K 768: 12.2 /
    // 11.2 / 13.6 / 13.0; K 1024: 14.2 / 12.8 / 16.5 / 16.0; K 512: 9.0 / 8.1 / 10.1 / 9.9; M 9536 N 768 K 768: 23.1 / 21.2 / 23.4 /
    // 24.8.  End to end at cfg2 it is neutral (7.13 / 7.21 against 7.13 / 7.17 ms per step): beside the weight-gradient stream these
    // launches take 2-3 x their isolated time whichever kernel runs.
    {
      gemm_opts_init();
      const int pk = g_opt_pipek;
      if (pk > 0 && splitk == 1 && !dbg && (g->K == 512 || g->K == 768 || g->K == 1024) && !(g->M & 63) && !(g->N & 63) && !g->out_f32 &&
          !g->res && g->alpha == 1.0f && !(g->ldo & 7) && g->act == LIDK_ACT_NONE && !g->out2) {
        const int tiles = (g->M / 64) * (g->N / 64);
        const int tpb = tiles >= 16 * pk ? pk : 1;
        const int G = tpb == 1 ? tiles : cdiv(cdiv(tiles, tpb), 8) * 8;
#define LIDK_PIPEK_LAUNCH(MODE_, KS_)                                                                                          \
  gemm_nt_bf16_pipek_kernel<MODE_, KS_><<<G, 256, 0, s>>>((const bf16*)g->A, (const bf16*)g->B, g->M, g->N, g->lda, g->ldb,   \
                                                          tiles, G, tpb, e)
        if (g->bias) {
          if (g->K == 512) LIDK_PIPEK_LAUNCH(PIPE_BIAS, 8); else if (g->K == 768) LIDK_PIPEK_LAUNCH(PIPE_BIAS, 12); else LIDK_PIPEK_LAUNCH(PIPE_BIAS, 16);
        } else {
          if (g->K == 512) LIDK_PIPEK_LAUNCH(PIPE_PLAIN, 8); else if (g->K == 768) LIDK_PIPEK_LAUNCH(PIPE_PLAIN, 12); else LIDK_PIPEK_LAUNCH(PIPE_PLAIN, 16);
        }
#undef LIDK_PIPEK_LAUNCH
        return launch_status();
      }
    }
    // LDS-DMA 128x128 kernel: K >= LIDK_GEMM_DMA (default 1024; 0 = never), N % 128 == 0, >= LIDK_GEMM_DMA_TILES tiles (default 384).
    // Per launch against the 64x64 / register-staged 128x128 kernels on the backbone shapes (tools/gemm_bench_wavlm.py): K = 3072
    // +36 % (798 TFLOP/s), conv stack K = 1536 +7...20 %, K = 768 single output +0...7 %; it LOSES with the two-output GELU
    // epilogue at K = 768 (store-bound) and when the tile count leaves a mostly empty second round (512 < tiles < 1024).
    // End to end (tools/gpu_ab_finetune.sh, 3 alternating rounds of 60 steps, spread +-0.05 ms once the per-step host syncs of the
    // backbone modules were gone - the A/B runs before that were noise): WavLM fine-tune 26.79 (off) / 26.76 (K >= 512) / 26.42
    // (K >= 2048) / 26.12 (K >= 1024) ms per step; frozen regime 21.21 (off) / 20.64 (2048) / 20.38 (1536) / 20.38 (1024).  The
    // cfg2 Conformer has no such shape (its K >= 1024 data gradients have 152 tiles: with the tile floor at 100 the step goes
    // 7.16 -> 7.28 ms (K >= 1024) / 7.68 (K >= 512)).
    const int dma_min_k = g_opt_dma;
    const long dma_tiles = (long)cdiv(g->M, 128) * (g->N / 128);
    const long dma_min_tiles = g_opt_dma_tiles;
    if (direct && splitk == 1 && dma_min_k > 0 && g->K >= dma_min_k && (g->K & 63) == 0 && !(g->N & 127) &&
        (g->K >= 1024 || !g->out2) && dma_tiles >= dma_min_tiles && (dma_tiles <= 512 || dma_tiles >= 1024) &&
        (size_t)128 * g->lda * 2 < (1ull << 31) && (size_t)128 * g->ldb * 2 < (1ull << 31)) {
      const int grid = (g->N / 128) * cdiv(g->M, 128);
      static bool dma_attr = false;
      if (!dma_attr) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        dma_attr = true;
      }
      gemm_nt_bf16_dma_kernel<<<grid, 256, 65536, s>>>((const bf16*)g->A, (const bf16*)g->B, g->M, g->N, g->K, g->lda, g->ldb, e);
      return launch_status();
    }
    static const int big_min = getenv("LIDK_GEMM_BIG") ? atoi(getenv("LIDK_GEMM_BIG")) : 256;      // 0 = never
    if (direct && splitk == 1 && (g->K & 63) == 0 && big_min > 0 && g->K >= 2048 && !(g->N & 127) &&
        (long)cdiv(g->M, 128) * (g->N / 128) >= big_min) {
      const int grid = (g->N / 128) * cdiv(g->M, 128);
      const int lds = 2 * (128 + 128) * BK * 2;
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
      }
      gemm_nt_bf16_big_kernel<<<grid, 256, lds, s>>>((const bf16*)g->A, (const bf16*)g->B, g->M, g->N, g->K, g->lda, g->ldb, e);
      return launch_status();
    }
    if (direct && splitk == 1 && (g->K & 63) == 0) {
      const int grid = cdiv(g->N, 64) * cdiv(g->M, 64);
      gemm_nt_bf16_direct_kernel<64, 64><<<grid, 256, 0, s>>>((const bf16*)g->A, (const bf16*)g->B, g->M, g->N, g->K, g->lda,
                                                             g->ldb, e);
      return launch_status();
    }
    long big = (long)cdiv(g->M, 128) * cdiv(g->N, 128) * splitk;
    static const int force_tile = getenv("LIDK_GEMM_TILE") ? atoi(getenv("LIDK_GEMM_TILE")) : 0;
    if (force_tile == 128 || (force_tile != 64 && big >= 384 && g->K >= 4096)) {
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


// =====================================================================================================================
// Weight-gradient GEMM, "TN":  C[N1][N2] += alpha * sum_m X[m][N1]^T-row * Y[m][N2]   (dW = dY^T . X), f32 accumulate.
// Both operands are consumed as stored (row-major, contraction index = row): tiles of BKM rows are staged row-major in LDS
// with 16-byte loads and the MFMA fragments come from transposed LDS reads (tr_frag), so no transposed copies exist in
// HBM.  The contraction (M = B*T rows) is split over grid.z; partial tiles are added with contiguous float atomics.
// colsum (optional): colsum[n1] += sum_m X[m][n1] — the bias gradient, accumulated from the X chunks as they are staged.
// =====================================================================================================================
#define BKM 64

// One output tile over one row chunk: tile = (bz, by, bx) of a (t1 x t2 tiles) x (row chunks of mchunk) decomposition.
// ATOMIC: partial tiles of several row chunks meet in C through float atomics; otherwise the tile is the only writer (C += ...).
template <int BN1, int BN2, bool FULL>    // FULL: N1 % BN1 == 0, N2 % BN2 == 0, every row tile complete -> unpredicated loads
__device__ __forceinline__ void
gemm_tn_tile(const bf16* __restrict__ X, int ldx, const bf16* __restrict__ Y, int ldy, float* __restrict__ C, int ldc,
             float* __restrict__ colsum, int M, int N1, int N2, int mchunk, float alpha, int tile, bool atomic) {
  constexpr int LDX = BN1 + 8, LDY = BN2 + 8, TM = BN1 / 32, TN = BN2 / 32, WM = BN1 / 2, WN = BN2 / 2, CST = WN + 4;
  constexpr int CX = BKM * BN1 / 8 / 256, CY = BKM * BN2 / 8 / 256, PX = BN1 / 8, PY = BN2 / 8;
  constexpr int AB_BYTES = BKM * (LDX + LDY) * 2, C_BYTES = 4 * WM * CST * 4, R_BYTES = 256 * 8 * 4;
  constexpr int SM = AB_BYTES > C_BYTES ? (AB_BYTES > R_BYTES ? AB_BYTES : R_BYTES) : (C_BYTES > R_BYTES ? C_BYTES : R_BYTES);
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  bf16* Xs = reinterpret_cast<bf16*>(smem);
  bf16* Ys = Xs + BKM * LDX;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1, fr = lane & 15, fq = lane >> 4;
  const int t1 = (N1 + BN1 - 1) / BN1, t2 = (N2 + BN2 - 1) / BN2;
  const int bz = tile / (t1 * t2), by = (tile / t2) % t1, bx = tile % t2;
  const int n1_0 = by * BN1, n2_0 = bx * BN2;
  const int mbeg = bz * mchunk, mend = min(M, mbeg + mchunk);
  const int N1p = (N1 + 7) & ~7, N2p = (N2 + 7) & ~7;           // operands are readable (zero padded) up to a multiple of 8

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool want_cs = colsum != nullptr && bx == 0;

  // Two row tiles of global loads in flight, in two named register sets (see gemm_nt_bf16_direct_kernel: predicated loads
  // or stage arrays indexed by a variable make hipcc drain vmcnt(0) before every LDS write / fall to scratch).
  u32x4 rx0[CX], ry0[CY], rx1[CX], ry1[CY];
  auto gload = [&](u32x4 (&rx)[CX], u32x4 (&ry)[CY], int m0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CX; ++i) {
      int c = tid + i * 256, row = c / PX, col = (c % PX) * 8;
      if (FULL) rx[i] = *reinterpret_cast<const u32x4*>(X + (size_t)(m0 + row) * ldx + n1_0 + col);
      else rx[i] = (m0 + row < mend && n1_0 + col < N1p) ? *reinterpret_cast<const u32x4*>(X + (size_t)(m0 + row) * ldx + n1_0 + col)
                                                          : (u32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < CY; ++i) {
      int c = tid + i * 256, row = c / PY, col = (c % PY) * 8;
      if (FULL) ry[i] = *reinterpret_cast<const u32x4*>(Y + (size_t)(m0 + row) * ldy + n2_0 + col);
      else ry[i] = (m0 + row < mend && n2_0 + col < N2p) ? *reinterpret_cast<const u32x4*>(Y + (size_t)(m0 + row) * ldy + n2_0 + col)
                                                          : (u32x4){0, 0, 0, 0};
    }
  };
  auto lstore = [&](const u32x4 (&rx)[CX], const u32x4 (&ry)[CY]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CX; ++i) {
      int c = tid + i * 256, row = c / PX, col = (c % PX) * 8;
      *reinterpret_cast<u32x4*>(&Xs[row * LDX + col]) = rx[i];
      if (want_cs) {
        const bf16* e = reinterpret_cast<const bf16*>(&rx[i]);
#pragma unroll
        for (int q = 0; q < 8; ++q) cs[q] += (float)e[q];
      }
    }
#pragma unroll
    for (int i = 0; i < CY; ++i) {
      int c = tid + i * 256, row = c / PY, col = (c % PY) * 8;
      *reinterpret_cast<u32x4*>(&Ys[row * LDY + col]) = ry[i];
    }
  };
  // pf: compile-time "refill this register set" (see gemm_nt_bf16_direct_kernel: a refill behind a runtime condition costs
  // the counted vmcnt waits and with them the whole prefetch)
  auto stage = [&](auto pf, u32x4 (&rx)[CX], u32x4 (&ry)[CY], int m_next) __attribute__((always_inline)) {
    lstore(rx, ry);
    __syncthreads();
    if constexpr (decltype(pf)::value) gload(rx, ry, m_next);
#pragma unroll
    for (int kk = 0; kk < BKM; kk += 32) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = tr_frag(Xs, LDX, kk, wm * WM + i * 16, fq, fr);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = tr_frag(Ys, LDY, kk, wn * WN + j * 16, fq, fr);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  };
  constexpr std::true_type PF{};
  constexpr std::false_type NOPF{};
  const int nt = (mend - mbeg + BKM - 1) / BKM;            // block-uniform; >= 1
  gload(rx0, ry0, mbeg);
  gload(rx1, ry1, nt > 1 ? mbeg + BKM : mbeg);       // unconditional, like the NT kernel's prologue
  int t = 0;
  for (; t + 3 < nt; t += 2) {
    stage(PF, rx0, ry0, mbeg + (t + 2) * BKM);
    stage(PF, rx1, ry1, mbeg + (t + 3) * BKM);
  }
  if (nt - t == 3) {
    stage(PF, rx0, ry0, mbeg + (t + 2) * BKM);
    stage(NOPF, rx1, ry1, 0);
    stage(NOPF, rx0, ry0, 0);
  } else if (nt - t == 2) {
    stage(NOPF, rx0, ry0, 0);
    stage(NOPF, rx1, ry1, 0);
  } else {
    stage(NOPF, rx0, ry0, 0);
  }
  // bias gradient: threads with equal tid % PX hold partial sums of the same 8 columns
  if (want_cs) {      // block-uniform
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int q = 0; q < 8; ++q) red[tid * 8 + q] = cs[q];
    __syncthreads();
    if (tid < BN1) {
      const int c8 = tid >> 3, q = tid & 7;
      float t = 0.f;
      for (int u = c8; u < 256; u += PX) t += red[u * 8 + q];
      if (n1_0 + tid < N1) {
        if (atomic) atomicAdd(&colsum[n1_0 + tid], t * alpha);
        else colsum[n1_0 + tid] += t * alpha;
      }
    }
    __syncthreads();
  }
  float* Cw = reinterpret_cast<float*>(smem) + wid * WM * CST;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cw[(i * 16 + fq * 4 + r) * CST + j * 16 + fr] = acc[i][j][r];
  __builtin_amdgcn_wave_barrier();
  constexpr int RPA = 64 / WN;
  for (int row = lane / WN; row < WM; row += RPA) {
    const int col = lane % WN, n1 = n1_0 + wm * WM + row, n2 = n2_0 + wn * WN + col;
    if (n1 < N1 && n2 < N2) {
      if (atomic) atomicAdd(&C[(size_t)n1 * ldc + n2], Cw[row * CST + col] * alpha);
      else C[(size_t)n1 * ldc + n2] += Cw[row * CST + col] * alpha;
    }
  }
}

template <int BN1, int BN2, bool FULL>
__global__ void __launch_bounds__(256)
gemm_tn_bf16_kernel(const bf16* __restrict__ X, int ldx, const bf16* __restrict__ Y, int ldy, float* __restrict__ C, int ldc,
                    float* __restrict__ colsum, int M, int N1, int N2, int mchunk, float alpha) {
  // 1-D launch, XCD-aware: all tiles of one row chunk (they share its X and Y rows) run on the same XCD back to back
  gemm_tn_tile<BN1, BN2, FULL>(X, ldx, Y, ldy, C, ldc, colsum, M, N1, N2, mchunk, alpha, xcd_tile(blockIdx.x, gridDim.x), true);
}

// Grouped form: ONE launch for the weight gradients of a whole ConformerBlock (ff1 x 2, ff2 x 2, conv pw1 / pw2, attention out,
// fused qkv).  Ten separate launches of ~512 short workgroups (19 row tiles each, 8-16 partial tiles meeting in atomics) kept
// the second stream busy for ~210 us per block at a fifth of the L2 rate; here every item runs a long row chunk (M / split
// rows, split = 4 by default), the launch holds ~1 700 items - the chip stays full from first to last wave - and a 4-way
// atomic fan-in replaces the 8-16-way one.  descs: device array, item0 ascending.
struct TnDesc {
  const bf16* X; const bf16* Y; float* C; float* colsum;
  int ldx, ldy, ldc, M, N1, N2, mchunk, item0, nsplit, pad0;
  float alpha; int pad1;
};
template <bool FULL, int BT>
__global__ void __launch_bounds__(256)
gemm_tn_grouped_kernel(const TnDesc* __restrict__ descs, int n_desc, int total_items) {
  // grid == total_items: one item per workgroup.  A smaller grid (LIDK_WGRAD_GRID) walks the items with a stride: the launch then
  // occupies a bounded share of every CU for longer, leaving room for the data-gradient chain it runs beside.
  for (int it = blockIdx.x; it < total_items; it += gridDim.x) {
    const int item = xcd_tile(it, total_items);
    int g = 0;
    while (g + 1 < n_desc && item >= descs[g + 1].item0) ++g;
    const TnDesc d = descs[g];
    gemm_tn_tile<BT, BT, FULL>(d.X, d.ldx, d.Y, d.ldy, d.C, d.ldc, d.colsum, d.M, d.N1, d.N2, d.mchunk, d.alpha, item - d.item0,
                               d.nsplit > 1);
    __syncthreads();                  // the tile's LDS staging area is reused by the next item
  }
}

// ------------------------------------------------------------------------------------ grouped TN, 128x128 tiles, LDS-DMA ring
// The register-staged tile above keeps two 64-row tiles of global loads in flight per workgroup and pays a ds_write pass and
// two barriers per 64 rows; with one item per CU (~200 items of ~4 800 rows at cfg2) it runs at ~13 % of the MFMA rate.  Here
// both operand tiles of a stage (64 rows x 128 columns each, 256 B per row) arrive by global_load_lds_dwordx4 into a ring of
// NST stages (NST - 1 stages = (NST - 1) x 32 KB in flight per CU, no staging registers, no ds_write pass, ONE barrier per
// stage).  A wave instruction carries 4 rows (lane-linear image), the 16-byte chunks of row r are XOR-permuted on the SOURCE
// address by sw(r) = 2 (r & 3) | 8 ((r >> 3) & 1): the 16 row segments one ds_read_b64_tr_b16 gathers (rows 8 fq + q) then
// fall on 8 different bank groups per half wave.  The bias gradient (column sums of X) is taken by the matrix cores too:
// X^T . 1 with a fragment of ones, two extra MFMAs per wave and K step on the tiles that own it.
constexpr int TN_STG = 32768;       // one stage: X rows [64][256 B] | Y rows [64][256 B]

__device__ __forceinline__ bf16x8 tr_pair256(const unsigned char* p) {      // rows k .. k+3 | k+4 .. k+7 of a 256-byte-row image
  union { struct { s16x4_t lo, hi; } h; bf16x8 v; } u;
  u.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)p);
  u.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p + 1024));
  return u.v;
}

__device__ __forceinline__ bf16x8 tr_pair512(const unsigned char* p) {      // the same for a 512-byte-row image
  union { struct { s16x4_t lo, hi; } h; bf16x8 v; } u;
  u.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)p);
  u.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p + 2048));
  return u.v;
}

// ABL (ablation builds, -DLIDK_TN_ABLATION, wrong results): 1 no DMA in the loop, 2 no fragment reads / MFMAs, 4 no output stores,
// 8 fragment reads without MFMAs, 16 MFMAs without fragment reads
template <int NST, bool CS, int ABL = 0>
__device__ __forceinline__ void gemm_tn_dma_item(const TnDesc d, int tile, unsigned char* smem, unsigned smem0) {
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
  const int t1 = d.N1 / 128, t2 = d.N2 / 128;
  const int bz = tile / (t1 * t2), by = (tile / t2) % t1, bx = tile % t2;
  const int n1_0 = by * 128, n2_0 = bx * 128;
  const int mbeg = bz * d.mchunk, mend = min(d.M, mbeg + d.mchunk), nt = (mend - mbeg) / 64;       // >= 1
  const bool atomic = d.nsplit > 1;

  // per-lane source offsets of this wave's 4 + 4 wave instructions per stage (stage rows 16 wid + 4 i + lane / 16)
  unsigned xoff[4], yoff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 16 * wid + 4 * i + (lane >> 4), c = (lane & 15) ^ (((r & 3) << 1) | (((r >> 3) & 1) << 3));
    xoff[i] = (unsigned)((r * d.ldx + 8 * c) * 2);
    yoff[i] = (unsigned)((r * d.ldy + 8 * c) * 2);
  }
  const bf16* xb = d.X + (size_t)mbeg * d.ldx + n1_0;
  const bf16* yb = d.Y + (size_t)mbeg * d.ldy + n2_0;
  const size_t xstep = (size_t)64 * d.ldx, ystep = (size_t)64 * d.ldy;
  auto issue = [&](int ti, int slot) __attribute__((always_inline)) {
    ti = min(ti, nt - 1);                              // past the end: the last tile again, into a slot nobody reads any more
    const bf16* xs = xb + ti * xstep;
    const bf16* ys = yb + ti * ystep;
    const unsigned dst = smem0 + slot * TN_STG + wid * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) gemm_glds16s(xs, xoff[i], dst + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) gemm_glds16s(ys, yoff[i], dst + 16384 + i * 1024);
  };

  // per-lane fragment offsets inside a stage: row 8 fq + fr / 4, 16-byte chunk (tile chunk ^ sw(row)), 8-byte half fr & 1
  const int swl = (((fr >> 2) & 3) << 1) | ((fq & 1) << 3);
  const int rowb = (8 * fq + (fr >> 2)) * 256 + (((fr & 3) >> 1) << 4) + (fr & 1) * 8;
  int aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = rowb + (((wm * 8 + 2 * i) ^ swl) << 4);
    boff[i] = 16384 + rowb + (((wn * 8 + 2 * i) ^ swl) << 4);
  }

  f32x4 acc[4][4], accs[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accs[0] = accs[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int q = 0; q < 8; ++q) ones[q] = (bf16)1.0f;

  // ABL & 32 (experiment): the same LDS image filled through registers (global_load_dwordx4 + ds_write_b128) instead of LDS-DMA
  u32x4 rg[8];
  auto gload = [&](int ti) __attribute__((always_inline)) {
    ti = min(ti, nt - 1);
    const char* xs = reinterpret_cast<const char*>(xb + ti * xstep);
    const char* ys = reinterpret_cast<const char*>(yb + ti * ystep);
#pragma unroll
    for (int i = 0; i < 4; ++i) { rg[i] = *reinterpret_cast<const u32x4*>(xs + xoff[i]); rg[4 + i] = *reinterpret_cast<const u32x4*>(ys + yoff[i]); }
  };
  auto lwrite = [&](int slot_) __attribute__((always_inline)) {
    unsigned char* dst = smem + slot_ * TN_STG + wid * 4096 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) { *reinterpret_cast<u32x4*>(dst + i * 1024) = rg[i]; *reinterpret_cast<u32x4*>(dst + 16384 + i * 1024) = rg[4 + i]; }
  };
  if (ABL & 32) {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) { gload(s); lwrite(s); }
    gload(NST - 1);
  } else {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s, s);
  }
  int slot = 0, islot = NST - 1;
  for (int t = 0; t < nt; ++t) {
    if (!(ABL & 33)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 8) : "memory");      // this wave's pieces of stage t have landed
    __syncthreads();                                  // everybody's have; and every wave is done reading the slot refilled next
    if (ABL & 32) { lwrite(islot); gload(t + NST); }
    else if (!(ABL & 1)) issue(t + NST - 1, islot);
    const unsigned char* sb = smem + slot * TN_STG;
    slot = slot + 1 == NST ? 0 : slot + 1;
    islot = islot + 1 == NST ? 0 : islot + 1;
    if (ABL & 2) continue;
    bf16x8 a0[4], b0[4], a1[4], b1[4];
    if (ABL & 16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a0[i] = b0[i] = a1[i] = b1[i] = ones;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) { a0[i] = tr_pair256(sb + aoff[i]); b0[i] = tr_pair256(sb + boff[i]); }
#pragma unroll
      for (int i = 0; i < 4; ++i) { a1[i] = tr_pair256(sb + aoff[i] + 8192); b1[i] = tr_pair256(sb + boff[i] + 8192); }
    }
    if (ABL & 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a0[i]), "v"(b0[i]), "v"(a1[i]), "v"(b1[i]));
      continue;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
    if (CS) {
      if (wn == 0) { accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[0], ones, accs[0], 0, 0, 0);
                     accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[1], ones, accs[1], 0, 0, 0); }
      else         { accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[2], ones, accs[0], 0, 0, 0);
                     accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[3], ones, accs[1], 0, 0, 0); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
    if (CS) {
      if (wn == 0) { accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[0], ones, accs[0], 0, 0, 0);
                     accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[1], ones, accs[1], 0, 0, 0); }
      else         { accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[2], ones, accs[0], 0, 0, 0);
                     accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[3], ones, accs[1], 0, 0, 0); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the clamped refills of the tail still write LDS
  __syncthreads();
  if (ABL & 4) return;

  if (CS && fr == 0) {                               // every column of X^T . 1 is the column sum; lanes fr == 0 hold rows 4 fq + r
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* dst = d.colsum + n1_0 + wm * 64 + (2 * wn + ii) * 16 + 4 * fq + r;
        if (atomic) atomicAdd(dst, accs[ii][r] * d.alpha);
        else *dst += accs[ii][r] * d.alpha;
      }
  }
  constexpr int CST = 64 + 4;
  float* Cw = reinterpret_cast<float*>(smem) + wid * 64 * CST;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) Cw[(i * 16 + fq * 4 + r) * CST + j * 16 + fr] = acc[i][j][r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll 4
  for (int row = 0; row < 64; ++row) {
    float* dst = d.C + (size_t)(n1_0 + wm * 64 + row) * d.ldc + n2_0 + wn * 64 + lane;
    const float v = Cw[row * CST + lane] * d.alpha;
    if (atomic) atomicAdd(dst, v);
    else *dst += v;
  }
}

template <int NST, int ABL = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_tn_grouped_dma_kernel(const TnDesc* __restrict__ descs, int n_desc, int total_items) {
  typedef __attribute__((address_space(3))) void lds_v;
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_dma_smem[];
  const unsigned smem0 = (unsigned)(size_t)(lds_v*)tn_dma_smem;
  // grid == total_items: one item per workgroup.  A smaller grid (LIDK_TN_GRID) walks the items with a stride: the launch then
  // holds a bounded number of CUs for longer and leaves the others to the LDS-heavy kernels of the data-gradient chain.  Measured
  // SLOWER at cfg2 (ms per step, two rounds): 0 (= 208 workgroups) 7.08 / 7.16, 152: 7.49 / 7.54, 104: 7.57 / 7.55, 70: 7.74 / 7.81.
  for (int it = blockIdx.x; it < total_items; it += gridDim.x) {
    const int item = xcd_tile(it, total_items);
    int g = 0;
    while (g + 1 < n_desc && item >= descs[g + 1].item0) ++g;
    const TnDesc d = descs[g];
    const int tile = item - d.item0;
    if (d.colsum != nullptr && tile % (d.N2 / 128) == 0) gemm_tn_dma_item<NST, true, ABL>(d, tile, tn_dma_smem, smem0);
    else gemm_tn_dma_item<NST, false, ABL>(d, tile, tn_dma_smem, smem0);
    __syncthreads();                  // the ring doubles as the output staging area of the item just finished
  }
}

// one weight gradient per launch: the record travels as a kernel argument
template <int NST>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_tn_dma_single_kernel(const TnDesc d, int total_items) {
  typedef __attribute__((address_space(3))) void lds_v;
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_dma1_smem[];
  const unsigned smem0 = (unsigned)(size_t)(lds_v*)tn_dma1_smem;
  const int tile = xcd_tile(blockIdx.x, total_items);
  if (d.colsum != nullptr && tile % (d.N2 / 128) == 0) gemm_tn_dma_item<NST, true>(d, tile, tn_dma1_smem, smem0);
  else gemm_tn_dma_item<NST, false>(d, tile, tn_dma1_smem, smem0);
}

template <int NST>
static void tn_dma_single_launch(const TnDesc& d, int total_items, hipStream_t s) {
  constexpr int lds = NST * TN_STG > 4 * 64 * 68 * 4 ? NST * TN_STG : 4 * 64 * 68 * 4;
  (void)hipFuncSetAttribute((const void*)gemm_tn_dma_single_kernel<NST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  gemm_tn_dma_single_kernel<NST><<<total_items, 256, lds, s>>>(d, total_items);
}

template <int NST, int ABL = 0>
static void tn_dma_launch(const void* descs, int n_desc, int total_items, hipStream_t s) {
  constexpr int lds = NST * TN_STG > 4 * 64 * 68 * 4 ? NST * TN_STG : 4 * 64 * 68 * 4;
  (void)hipFuncSetAttribute((const void*)gemm_tn_grouped_dma_kernel<NST, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  static const int cap = getenv("LIDK_TN_GRID") ? atoi(getenv("LIDK_TN_GRID")) : 0;          // 0: one workgroup per item
  const int grid = cap > 0 && cap < total_items ? cap : total_items;
  gemm_tn_grouped_dma_kernel<NST, ABL><<<grid, 256, lds, s>>>((const TnDesc*)descs, n_desc, total_items);
}

// ------------------------------------------------------------------------------------ grouped TN, 256x256 tiles, LDS-DMA
// The 128-tile ring above is bound by the L2 -> LDS fill rate, not by HBM or the matrix cores: a 128x128 tile moves one operand
// byte per 64 FLOP, ~510 MB per launch at cfg2, and the fabric delivers ~16-20 B/clk per CU when every CU pulls (the ring depth
// makes no difference: 63.5 / 65.4 us with 3 / 4 stages, tools/wgrad_bench.py).  A 256x256 tile halves the bytes per FLOP.
// 8 waves (2 x 4), each 128 x 64 outputs = 32 accumulator tiles; a stage is 64 rows of X [256 cols] and Y [256 cols] = 64 KB,
// two stages = 128 KB per workgroup.  A wave instruction carries 2 rows of 512 B; chunk permutation as above (low 4 bits).
constexpr int TN256_STG = 65536;

template <bool CS>
__device__ __forceinline__ void gemm_tn_dma256_item(const TnDesc d, int tile, unsigned char* smem, unsigned smem0) {
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 2, wn = wid & 3;
  const int t1 = d.N1 / 256, t2 = d.N2 / 256;
  const int bz = tile / (t1 * t2), by = (tile / t2) % t1, bx = tile % t2;
  const int n1_0 = by * 256, n2_0 = bx * 256;
  const int mbeg = bz * d.mchunk, mend = min(d.M, mbeg + d.mchunk), nt = (mend - mbeg) / 64;       // >= 1
  const bool atomic = d.nsplit > 1;

  unsigned xoff[4], yoff[4];                       // stage rows 8 wid + 2 i + lane / 32, chunk (lane % 32) ^ sw(row)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8 * wid + 2 * i + (lane >> 5), c = (lane & 31) ^ (((r & 3) << 1) | (((r >> 3) & 1) << 3));
    xoff[i] = (unsigned)((r * d.ldx + 8 * c) * 2);
    yoff[i] = (unsigned)((r * d.ldy + 8 * c) * 2);
  }
  const bf16* xb = d.X + (size_t)mbeg * d.ldx + n1_0;
  const bf16* yb = d.Y + (size_t)mbeg * d.ldy + n2_0;
  const size_t xstep = (size_t)64 * d.ldx, ystep = (size_t)64 * d.ldy;
  auto issue = [&](int ti, int slot) __attribute__((always_inline)) {
    const bf16* xs = xb + ti * xstep;
    const bf16* ys = yb + ti * ystep;
    const unsigned dst = smem0 + slot * TN256_STG + wid * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) gemm_glds16s(xs, xoff[i], dst + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) gemm_glds16s(ys, yoff[i], dst + 32768 + i * 1024);
  };

  const int swl = (((fr >> 2) & 3) << 1) | ((fq & 1) << 3);
  const int rowb = (8 * fq + (fr >> 2)) * 512 + (((fr & 3) >> 1) << 4) + (fr & 1) * 8;
  const int a_base = rowb + (((wm * 16) ^ swl) << 4);              // fragment i: a_base ^ (i << 5)  (bits 5-7: 2 i ^ swl)
  const int b_base = 32768 + rowb + (((wn * 8) ^ swl) << 4);       // fragment j: b_base ^ (j << 5)

  f32x4 acc[8][4], accs[2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accs[0] = accs[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int q = 0; q < 8; ++q) ones[q] = (bf16)1.0f;

  issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of stage t have landed
    __syncthreads();                                      // everybody's have; every wave is done reading the other slot
    if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
    const unsigned char* sb = smem + (t & 1) * TN256_STG;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[8], b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = tr_pair512(sb + (b_base ^ (j << 5)) + kk * 16384);
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = tr_pair512(sb + (a_base ^ (i << 5)) + kk * 16384);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if (CS) {
        if (wn == 0)      { accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], ones, accs[0], 0, 0, 0);
                            accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], ones, accs[1], 0, 0, 0); }
        else if (wn == 1) { accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], ones, accs[0], 0, 0, 0);
                            accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[3], ones, accs[1], 0, 0, 0); }
        else if (wn == 2) { accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[4], ones, accs[0], 0, 0, 0);
                            accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[5], ones, accs[1], 0, 0, 0); }
        else              { accs[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[6], ones, accs[0], 0, 0, 0);
                            accs[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[7], ones, accs[1], 0, 0, 0); }
      }
    }
  }
  __syncthreads();                                     // the last stage's slot becomes the output staging area

  if (CS && fr == 0) {                                // every column of X^T . 1 is the column sum; lanes fr == 0 hold rows 4 fq + r
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* dst = d.colsum + n1_0 + wm * 128 + (2 * wn + ii) * 16 + 4 * fq + r;
        if (atomic) atomicAdd(dst, accs[ii][r] * d.alpha);
        else *dst += accs[ii][r] * d.alpha;
      }
  }
  constexpr int CST = 64 + 4;
  float* Cw = reinterpret_cast<float*>(smem) + wid * 32 * CST;       // 32 output rows of the wave at a time
#pragma unroll
  for (int h = 0; h < 4; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cw[(i * 16 + fq * 4 + r) * CST + j * 16 + fr] = acc[2 * h + i][j][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int row = 0; row < 32; ++row) {
      float* dst = d.C + (size_t)(n1_0 + wm * 128 + h * 32 + row) * d.ldc + n2_0 + wn * 64 + lane;
      const float v = Cw[row * CST + lane] * d.alpha;
      if (atomic) atomicAdd(dst, v);
      else *dst += v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void __launch_bounds__(512)
gemm_tn_grouped_dma256_kernel(const TnDesc* __restrict__ descs, int n_desc, int total_items) {
  typedef __attribute__((address_space(3))) void lds_v;
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_dma256_smem[];
  const unsigned smem0 = (unsigned)(size_t)(lds_v*)tn_dma256_smem;
  const int item = xcd_tile(blockIdx.x, total_items);
  int g = 0;
  while (g + 1 < n_desc && item >= descs[g + 1].item0) ++g;
  const TnDesc d = descs[g];
  const int tile = item - d.item0;
  if (d.colsum != nullptr && tile % (d.N2 / 256) == 0) gemm_tn_dma256_item<true>(d, tile, tn_dma256_smem, smem0);
  else gemm_tn_dma256_item<false>(d, tile, tn_dma256_smem, smem0);
}

// 256x256 output tiles (item0 / total_items counted in 256-tiles; every record needs N1 % 256 == N2 % 256 == M % 64 == 0)
extern "C" int lidk_gemm_tn_grouped256(const void* descs, int n_desc, int total_items, void* stream) {
  if (!descs || n_desc <= 0 || total_items <= 0) return LIDK_ERR_ARG;
  (void)hipFuncSetAttribute((const void*)gemm_tn_grouped_dma256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TN256_STG);
  gemm_tn_grouped_dma256_kernel<<<total_items, 512, 2 * TN256_STG, as_stream(stream)>>>((const TnDesc*)descs, n_desc, total_items);
  return launch_status();
}

extern "C" int lidk_gemm_tn_desc_bytes(void) { return (int)sizeof(TnDesc); }

// descs: n_desc TnDesc records on the DEVICE (layout: lidk_gemm_tn_desc_bytes(), built by the host binding); total_items =
// sum over records of tiles(N1, 64) * tiles(N2, 64) * nsplit; full != 0 asserts that every record has N1 % 64 == N2 % 64 ==
// M % 64 == 0 (unpredicated loads).
extern "C" int lidk_gemm_tn_grouped(const void* descs, int n_desc, int total_items, int full, void* stream) {
  if (!descs || n_desc <= 0 || total_items <= 0) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  static const int cap = getenv("LIDK_WGRAD_GRID") ? atoi(getenv("LIDK_WGRAD_GRID")) : 0;      // 0: one workgroup per item
  const int grid = cap > 0 && cap < total_items ? cap : total_items;
  if (full) gemm_tn_grouped_kernel<true, 64><<<grid, 256, 0, s>>>((const TnDesc*)descs, n_desc, total_items);
  else gemm_tn_grouped_kernel<false, 64><<<grid, 256, 0, s>>>((const TnDesc*)descs, n_desc, total_items);
  return launch_status();
}

// The same launch on 128x128 output tiles (item0 / total_items counted in 128-tiles; every record needs N1 % 128 == N2 % 128 ==
// M % 64 == 0): twice the MFMA work per operand byte staged through LDS.
extern "C" int lidk_gemm_tn_grouped128(const void* descs, int n_desc, int total_items, void* stream) {
  if (!descs || n_desc <= 0 || total_items <= 0) return LIDK_ERR_ARG;
  // LIDK_TN_DMA (read per call): ring depth of the LDS-DMA kernel (2 / 3 / 4), 0: the register-staged tiles.  Default: 3 stages
  // (96 KB, one workgroup per CU) while every item finds a CU of its own, 2 stages (64 KB, two per CU) beyond that.  Measured on
  // a cfg2 block (tools/wgrad_bench.py, us per launch): register-staged 102 (208 items) / 108 (416); ring of 2: 75 / 62,
  // of 3: 63.5 / 70.5, of 4: 65 / 72.
  const char* env = getenv("LIDK_TN_DMA");
  const int nst = env ? atoi(env) : (total_items > 256 ? 2 : 3);
#ifdef LIDK_TN_ABLATION
  const int abl = getenv("LIDK_TN_ABL") ? atoi(getenv("LIDK_TN_ABL")) : 0;
  hipStream_t as = as_stream(stream);
  switch (abl) {
    case 1: tn_dma_launch<3, 1>(descs, n_desc, total_items, as); return launch_status();
    case 2: tn_dma_launch<3, 2>(descs, n_desc, total_items, as); return launch_status();
    case 4: tn_dma_launch<3, 4>(descs, n_desc, total_items, as); return launch_status();
    case 5: tn_dma_launch<3, 5>(descs, n_desc, total_items, as); return launch_status();
    case 6: tn_dma_launch<3, 6>(descs, n_desc, total_items, as); return launch_status();
    case 8: tn_dma_launch<3, 8>(descs, n_desc, total_items, as); return launch_status();
    case 9: tn_dma_launch<3, 9>(descs, n_desc, total_items, as); return launch_status();
    case 16: tn_dma_launch<3, 16>(descs, n_desc, total_items, as); return launch_status();
    case 17: tn_dma_launch<3, 17>(descs, n_desc, total_items, as); return launch_status();
    case 32: tn_dma_launch<3, 32>(descs, n_desc, total_items, as); return launch_status();
    case 34: tn_dma_launch<3, 34>(descs, n_desc, total_items, as); return launch_status();
    case 38: tn_dma_launch<3, 38>(descs, n_desc, total_items, as); return launch_status();
    default: break;
  }
#endif
  if (nst == 2) tn_dma_launch<2>(descs, n_desc, total_items, as_stream(stream));
  else if (nst == 3) tn_dma_launch<3>(descs, n_desc, total_items, as_stream(stream));
  else if (nst >= 4) tn_dma_launch<4>(descs, n_desc, total_items, as_stream(stream));
  else gemm_tn_grouped_kernel<true, 128><<<total_items, 256, 0, as_stream(stream)>>>((const TnDesc*)descs, n_desc, total_items);
  return launch_status();
}

// f32 (parity mode): 64x64 tile, 16 rows of m per step, 4x4 outputs per thread; tiles are used as stored.
__global__ void __launch_bounds__(256)
gemm_tn_f32_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ Y, int ldy, float* __restrict__ C, int ldc,
                   float* __restrict__ colsum, int M, int N1, int N2, int mchunk, float alpha) {
  __shared__ float Xs[16][64 + 4];
  __shared__ float Ys[16][64 + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int n1_0 = blockIdx.y * 64, n2_0 = blockIdx.x * 64;
  const int mbeg = blockIdx.z * mchunk, mend = min(M, mbeg + mchunk);
  const int lr = tid >> 4, lc = (tid & 15) * 4;          // stage row lr (of 16), columns lc..lc+3
  float acc[4][4] = {};
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  for (int m0 = mbeg; m0 < mend; m0 += 16) {
    float xv[4] = {0.f, 0.f, 0.f, 0.f}, yv[4] = {0.f, 0.f, 0.f, 0.f};
    if (m0 + lr < mend) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (n1_0 + lc + q < N1) xv[q] = X[(size_t)(m0 + lr) * ldx + n1_0 + lc + q];
        if (n2_0 + lc + q < N2) yv[q] = Y[(size_t)(m0 + lr) * ldy + n2_0 + lc + q];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { Xs[lr][lc + q] = xv[q]; Ys[lr][lc + q] = yv[q]; cs[q] += xv[q]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = Xs[k][ty * 4 + i]; bv[i] = Ys[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  if (colsum != nullptr && blockIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) if (n1_0 + lc + q < N1) atomicAdd(&colsum[n1_0 + lc + q], cs[q] * alpha);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n1 = n1_0 + ty * 4 + i, n2 = n2_0 + tx * 4 + j;
      if (n1 < N1 && n2 < N2) atomicAdd(&C[(size_t)n1 * ldc + n2], acc[i][j] * alpha);
    }
}

extern "C" int lidk_gemm_tn(const void* X, int ldx, const void* Y, int ldy, float* C, int ldc, float* colsum, int M, int N1,
                            int N2, float alpha, int splitk, int dtype, void* stream) {
  if (!X || !Y || !C || M <= 0 || N1 <= 0 || N2 <= 0 || ldc < N2) return LIDK_ERR_ARG;
  if ((ldx & 7) || (ldy & 7)) return LIDK_ERR_ARG;        // ldx < N1 / ldy < N2: overlapping rows (strided-view convolution operands)
  hipStream_t s = as_stream(stream);
  if (splitk < 1) splitk = 1;
  if (dtype == LIDK_BF16) {
    int mchunk = cdiv(cdiv(M, splitk), BKM) * BKM;
    splitk = cdiv(M, mchunk);
    // Full 128-tile shapes with enough tiles: the LDS-DMA ring (gemm_tn_dma_item).  It chooses its own row split - as many
    // chunks (<= 8) as keep the launch within 512 items, two 64 KB workgroups per CU - because every extra chunk is another
    // pass of float atomics over the output.  LIDK_TN_DMA=0 turns it off (read per call).
    {
      const char* env = getenv("LIDK_TN_DMA");
      const long tiles = (long)(N1 / 128) * (N2 / 128);
      if (!(env && atoi(env) == 0) && !(N1 % 128) && !(N2 % 128) && !(M % 64) && tiles >= 16) {
        int split = (int)std::min<long>(8, std::max<long>(1, 512 / tiles));
        const int mc = cdiv(cdiv(M, split), 64) * 64;
        split = cdiv(M, mc);
        TnDesc d{(const bf16*)X, (const bf16*)Y, C, colsum, ldx, ldy, ldc, M, N1, N2, mc, 0, split, 0, alpha, 0};
        const int items = (int)tiles * split;
        if (items > 256) tn_dma_single_launch<2>(d, items, s);
        else tn_dma_single_launch<3>(d, items, s);
        return launch_status();
      }
    }
    static const int tn_tile = getenv("LIDK_TN_TILE") ? atoi(getenv("LIDK_TN_TILE")) : 0;
    const bool big = tn_tile == 128 || (tn_tile != 64 && N1 >= 128 && N2 >= 128 && (long)cdiv(N1, 128) * cdiv(N2, 128) * splitk >= 256);
    const int bn = big ? 128 : 64;
    const bool full = !(N1 % bn) && !(N2 % bn) && !(M % BKM);        // mchunk is a multiple of BKM
    const int grid = cdiv(N2, bn) * cdiv(N1, bn) * splitk;
#define LIDK_TN_LAUNCH(BN_, FULL_)                                                                                          \
  gemm_tn_bf16_kernel<BN_, BN_, FULL_><<<grid, 256, 0, s>>>((const bf16*)X, ldx, (const bf16*)Y, ldy, C, ldc, colsum, M, N1, \
                                                           N2, mchunk, alpha)
    if (big && full) LIDK_TN_LAUNCH(128, true);
    else if (big) LIDK_TN_LAUNCH(128, false);
    else if (full) LIDK_TN_LAUNCH(64, true);
    else LIDK_TN_LAUNCH(64, false);
#undef LIDK_TN_LAUNCH
  } else if (dtype == LIDK_F32) {
    int mchunk = cdiv(cdiv(M, splitk), 16) * 16;
    splitk = cdiv(M, mchunk);
    dim3 grid(cdiv(N2, 64), cdiv(N1, 64), splitk);
    gemm_tn_f32_kernel<<<grid, 256, 0, s>>>((const float*)X, ldx, (const float*)Y, ldy, C, ldc, colsum, M, N1, N2, mchunk, alpha);
  } else {
    return LIDK_ERR_ARG;
  }
  return launch_status();
}
