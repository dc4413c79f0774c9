// Fused FeedForward module (lid/conformer.py:153-171 with its PreNorm :81-89 and Scale(0.5) + residual :247-248,252-259) for the
// model width d = 256, bf16 operands:
//     xo = x + alpha * ( swish( LN(x) . W1^T + b1 ) . W2^T + b2 )
// One workgroup owns 64 (or 48: ffn_row_groups) complete rows; a wave owns 16 of them and half of every chunk's hidden columns:
//   * prologue: the wave normalises its rows straight into MFMA operand registers (lane (fr, fq) holds row fr, columns
//     32*ks + 8*fq .. +7 of every 32-wide K step: 8 x bf16x8 = the A fragments of the up-projection), writes h / mean / rstd for
//     the backward pass;
//   * the hidden dimension is walked in chunks of 64 columns.  The chunk's W1 rows [64][256] and W2 columns [256][64] (32 KB each)
//     arrive in LDS by global_load_lds (16 B per lane, no staging registers), one chunk ahead of the MFMAs, in two buffers;
//   * up-projection of the chunk (32 MFMAs per wave), bias + Swish in registers, pre-activation and activation stored for the
//     backward pass; issued "transposed" and with the W1 rows permuted in LDS, a lane's accumulators of two adjacent 16-column
//     tiles are 8 CONSECUTIVE hidden columns of its row - i.e. packed to bf16 they ARE the A fragment of the down-projection:
//     the [M][ff] activation never goes through LDS or HBM on its way to the second GEMM;
//   * down-projection accumulates the chunk into 16 output tiles (64 f32 registers per lane); epilogue adds b2, scales, adds x.
// LDS images are lane-linear per wave instruction (what global_load_lds writes); bank conflicts are avoided by permuting the
// 16-byte chunks of a row on the SOURCE address and applying the same XOR on the fragment reads (cdna_hip_programming.md rule 21).
// HBM traffic per row: x read twice (second time from L2), h + a + u + xo written once; the weights (1 MB) stream from L2 once
// per workgroup.  Replaces lidk_layernorm_fwd + 2 x lidk_gemm_nt (three launches, u re-read, h re-read).
#include "common.h"
#include <stdlib.h>
#include <string.h>

typedef __attribute__((address_space(3))) void lds_void_t;

// LIDK_FFN_DBG (ablations, wrong results): 1 no a / u / da stores, 2 no weight DMA after chunk 0, 16 no chunk loop.  LIDK_FFN_ROT=0:
// lock-step chunk order.
static int ffn_dbg() { static const int v = getenv("LIDK_FFN_DBG") ? atoi(getenv("LIDK_FFN_DBG")) : 0; return v; }
static int ffn_rot() { static const int v = getenv("LIDK_FFN_ROT") ? atoi(getenv("LIDK_FFN_ROT")) : 1; return v; }
// Row groups (of 16 rows) per workgroup, one policy for the three kernels and lidk_ffn_bwd_partial_rows: 48-row workgroups when
// that grid still fits ONE round of the chip's 256 CUs (these kernels run one workgroup per CU), else 64-row ones (fewer weight
// streams per row).  LIDK_FFN_RG = 3 / 4 forces one form (A/B runs).
static int g_ffn_rg = -1;
static int ffn_row_groups(int M) {
  if (g_ffn_rg < 0) g_ffn_rg = getenv("LIDK_FFN_RG") ? atoi(getenv("LIDK_FFN_RG")) : 0;
  if (g_ffn_rg == 3 || g_ffn_rg == 4) return g_ffn_rg;
  return (M + 47) / 48 <= 256 ? 3 : 4;
}
extern "C" int lidk_ffn_option(const char* name, long value) {
  if (!name || strcmp(name, "LIDK_FFN_RG")) return LIDK_ERR_ARG;
  g_ffn_rg = (int)value;                                 // 3 / 4: forced, 0: by M, negative: re-read the environment
  return LIDK_OK;
}
typedef __attribute__((address_space(1))) const void gbl_void_t;

#define FFN_D 256
#define FFN_BM 64
#define FFN_CH 64
#define FFN_BUF (2 * FFN_CH * FFN_D * 2)      // bytes of one chunk buffer: W1 rows + W2 columns

struct FfnFwd {
  const float* x; const bf16* h_in; const float* gamma; const float* beta; float eps;
  const bf16* W1; const float* b1; const bf16* W2; const float* b2;
  bf16* h; float* mean; float* rstd; bf16* a; bf16* u; float* xo;
  float alpha; int M; int FF; int rot; int dbg;
  // LayerNorm(s) that consume xo, applied in the epilogue (rows are complete): A on xo, optionally B on A's output
  const float* gA; const float* bA; float* yA32; bf16* yAT; float* meanA; float* rstdA;
  const float* gB; const float* bB; bf16* yBT; float* meanB; float* rstdB;
};

// LDS-DMA by inline asm: hipcc tracks a builtin global_load_lds as a pending LDS write and drains it (vmcnt(0)) in front of the
// first LDS read it cannot prove disjoint - in the middle of the chunk the copy is meant to overlap.  Issued this way the compiler
// does not see the copies; the kernel waits for them itself (vmcnt(0) + barrier at the top of a chunk).  lds_off: wave-uniform
// byte offset of the wave instruction's 1 KB destination (M0), lane l lands at lds_off + 16 l.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_off) : "memory");
}
// the same with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: the per-lane part is loop-invariant in
// the chunk loops, so a chunk's staging costs scalar adds only - no vector address arithmetic between the MFMAs
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_off) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}

__device__ __forceinline__ bf16x8 pack_bf16x8(float4 a, float4 b) {
  bf16x8 r;
  r[0] = (bf16)a.x; r[1] = (bf16)a.y; r[2] = (bf16)a.z; r[3] = (bf16)a.w;
  r[4] = (bf16)b.x; r[5] = (bf16)b.y; r[6] = (bf16)b.z; r[7] = (bf16)b.w;
  return r;
}

// Per-lane byte offsets (chunk-invariant) of this wave's 8 of the 64 wave instructions that stage a chunk: 4 for the W1 rows, 4 for
// the W2 columns.  W1 rows: LDS row rho (512 B) holds global row n(rho) with its 16-byte chunks XOR-permuted by (rho & 15); rho -> n
// undoes the "pair" permutation that makes two adjacent accumulator tiles 8 consecutive columns.  W2 columns: LDS row = output
// column (128 B), chunks XOR-permuted by ((row >> 1) & 7).
// RG = row groups of 16 rows per workgroup (4: 64 rows, 8 waves; 3: 48 rows, 6 waves).  The 32 + 32 wave instructions are dealt to
// the 2 RG waves: wave w issues instructions q = 4 w + i (RG = 4) or q = w + 6 i while q < 32 (RG = 3: two waves issue 6, four 5).
template <int RG> struct FfnStageOff { unsigned w1[RG == 4 ? 4 : 6], w2[RG == 4 ? 4 : 6]; };
template <int RG> __device__ __forceinline__ int ffn_stage_q(int wid, int i) { return RG == 4 ? 4 * wid + i : wid + 2 * RG * i; }
template <int RG>
__device__ __forceinline__ FfnStageOff<RG> ffn_stage_offsets(int FF, int wid, int lane) {
  FfnStageOff<RG> o;
#pragma unroll
  for (int i = 0; i < (RG == 4 ? 4 : 6); ++i) {
    const int q = ffn_stage_q<RG>(wid, i) & 31;
    const int rho = 2 * q + (lane >> 5), p = lane & 31, cc = p ^ (rho & 15);
    const int n = (rho & 32) + 8 * ((rho >> 2) & 3) + 4 * ((rho >> 4) & 1) + (rho & 3);
    o.w1[i] = (unsigned)((n * FFN_D + cc * 8) * 2);
    const int row = 8 * q + (lane >> 3), p2 = lane & 7, c2 = p2 ^ ((row >> 1) & 7);
    o.w2[i] = (unsigned)((row * FF + c2 * 8) * 2);
  }
  return o;
}
// the W2-type half of a chunk (columns c FFN_CH .. of a [256][ld] matrix) -> LDS at byte offset `dst`
template <int RG>
__device__ __forceinline__ void ffn_stage_cols(const bf16* __restrict__ b2, const FfnStageOff<RG>& o, unsigned dst, int wid) {
#pragma unroll
  for (int i = 0; i < (RG == 4 ? 4 : 6); ++i) {
    const int q = ffn_stage_q<RG>(wid, i);
    if (RG == 4 || q < 32) glds16s(b2, o.w2[i], dst + q * 1024);
  }
}
// chunk c of W1 / W2 -> LDS buffer at byte offset `buf` (wid is an SGPR value: all address arithmetic here is scalar)
template <int RG>
__device__ __forceinline__ void ffn_stage(const bf16* __restrict__ W1, const bf16* __restrict__ W2, const FfnStageOff<RG>& o, int c,
                                          unsigned buf, int wid) {
  const bf16* b1 = W1 + (size_t)c * FFN_CH * FFN_D;
#pragma unroll
  for (int i = 0; i < (RG == 4 ? 4 : 6); ++i) {
    const int q = ffn_stage_q<RG>(wid, i);
    if (RG == 4 || q < 32) glds16s(b1, o.w1[i], buf + q * 1024);
  }
  ffn_stage_cols<RG>(W2 + (size_t)c * FFN_CH, o, buf + FFN_CH * FFN_D * 2, wid);
}

// one 1 KB wave instruction of a small f32 vector (bias, LayerNorm weights) -> LDS, unpermuted
__device__ __forceinline__ void ffn_stage_vec(const float* __restrict__ v, int n_floats, unsigned lds_off, int k, int lane) {
  const int i = min(k * 256 + lane * 4, n_floats - 4);          // past the end: re-read the last 16 bytes into the image's padding
  glds16(v + i, lds_off + k * 1024);
}

// the workgroup's 64 rows of the f32 residual stream -> LDS [64][1 KB], 16-byte chunks XOR-permuted by (row & 15); wave w copies
// rows 8 w .. 8 w + 7, one row per wave instruction
__device__ __forceinline__ void ffn_stage_rows(const float* __restrict__ x, int m0, int M, unsigned lds_off, int wid, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 8 * wid + i;
    glds16(x + (size_t)min(m0 + row, M - 1) * FFN_D + ((lane ^ (row & 15)) << 2), lds_off + row * 1024);
  }
}

// 8 waves: wave (half = wid >> 2, rw = wid & 3) owns rows 16 rw .. + 15 of the workgroup's 64 and HALF of every chunk's hidden
// columns (32 half .. + 31): its up-projection is 2 column tiles x 8 K steps, its down-projection the K = 32 slice of all 16 output
// tiles that those hidden columns feed.  The two partial outputs of a row group meet once, after the last chunk, through LDS.
// Two waves per SIMD, so one wave's LDS latency and barrier waits hide behind the other's MFMAs.
// Latency chain of a workgroup (all of them run in lock-step, there is no second workgroup on the CU to hide behind): ONE HBM round
// trip in the prologue - the row loads are issued first, the small vectors (b1, b2, gamma, beta) and chunk 0 follow by LDS-DMA - and
// none in the epilogue: the residual rows the epilogue adds come back by LDS-DMA during the last chunk, into the free chunk buffer.
// RG = 3: 48-row workgroups of 6 waves (3 row groups x 2 halves) - the grid the launcher picks when it then fits one round of the
// 256 CUs (M <= 12 288: 202 workgroups instead of 151 at the benchmarked M = 9 664): a workgroup's LDS fragment reads - what bounds
// the chunk loop - and its prologue / epilogue rows fall by a quarter.
template <bool LN_IN, int ABL, int RG>      // ABL (ablation builds, wrong results): 1 no fragment reads, 2 no MFMAs, 4 no bias / Swish math
__global__ void __launch_bounds__(128 * RG) __attribute__((amdgpu_waves_per_eu(2, 2)))
ffn_fwd_kernel(FfnFwd p) {
  constexpr int NW = 2 * RG, BM = 16 * RG;
  extern __shared__ __attribute__((aligned(16))) unsigned char ffn_smem[];
  float* lb1 = reinterpret_cast<float*>(ffn_smem + 2 * FFN_BUF);
  const int FFp = (p.FF + 255) & ~255;                        // the b1 image is padded to whole 1 KB wave instructions
  float* lb2 = lb1 + FFp;
  float* lgam = lb2 + FFN_D;
  float* lbet = lgam + FFN_D;
  const unsigned smem0 = (unsigned)(size_t)(lds_void_t*)ffn_smem;
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), half = wid >= RG ? 1 : 0, rw = wid - RG * half;
  const int m0 = blockIdx.x * BM;
  const int m_raw = m0 + 16 * rw + fr;
  const bool m_ok = m_raw < p.M;
  const int m = m_ok ? m_raw : p.M - 1;
  const int NC = (p.dbg & 16) ? 0 : p.FF / FFN_CH;
  // Chunk order is rotated per workgroup (the sums over chunks commute): the ~19 workgroups sharing an XCD stream different chunks
  // at any moment, so a chunk is usually already in the XCD's L2 when a workgroup asks for it.
  const int rot = p.rot ? (blockIdx.x >> 3) % (p.FF / FFN_CH) : 0;
  auto chunk_of = [&](int c) __attribute__((always_inline)) { const int t = c + rot; return t >= NC ? t - NC : t; };

  const FfnStageOff<RG> so = ffn_stage_offsets<RG>(p.FF, wid, lane);

  // ---- register loads first: this lane's 64 row elements (or the ready-made h fragments)
  bf16x8 hA[8];
  float4 xv[16];
  if (LN_IN) {
    const float* xr = p.x + (size_t)m * FFN_D + 8 * fq;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) { xv[2 * ks] = load4(xr + 32 * ks); xv[2 * ks + 1] = load4(xr + 32 * ks + 4); }
  } else {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) hA[ks] = *reinterpret_cast<const bf16x8*>(p.h_in + (size_t)m * FFN_D + 32 * ks + 8 * fq);
  }
  // ---- small vectors and chunk 0 by LDS-DMA
  {
    const int nb1 = FFp / 256;                                    // wave instructions for b1; then b2, gamma, beta: one each
    const unsigned vec0 = smem0 + 2 * FFN_BUF;
    for (int k = wid; k < nb1; k += NW) ffn_stage_vec(p.b1, p.FF, vec0, k, lane);
    if (wid == nb1 % NW) ffn_stage_vec(p.b2, FFN_D, vec0 + FFp * 4, 0, lane);
    if (LN_IN) {
      if (wid == (nb1 + 1) % NW) ffn_stage_vec(p.gamma, FFN_D, vec0 + FFp * 4 + 1024, 0, lane);
      if (wid == (nb1 + 2) % NW) ffn_stage_vec(p.beta, FFN_D, vec0 + FFp * 4 + 2048, 0, lane);
    }
    if (p.gA) {
      if (wid == (nb1 + 3) % NW) ffn_stage_vec(p.gA, FFN_D, vec0 + FFp * 4 + 3072, 0, lane);
      if (wid == (nb1 + 4) % NW) ffn_stage_vec(p.bA, FFN_D, vec0 + FFp * 4 + 4096, 0, lane);
    }
    if (p.gB) {
      if (wid == (nb1 + 5) % NW) ffn_stage_vec(p.gB, FFN_D, vec0 + FFp * 4 + 5120, 0, lane);
      if (wid == (nb1 + 6) % NW) ffn_stage_vec(p.bB, FFN_D, vec0 + FFp * 4 + 6144, 0, lane);
    }
  }
  if (NC > 0) ffn_stage<RG>(p.W1, p.W2, so, chunk_of(0), smem0, wid);

  if (LN_IN) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    const float mu = s * (1.0f / FFN_D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float a = xv[i].x - mu, b = xv[i].y - mu, c = xv[i].z - mu, d = xv[i].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
    q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
    const float rs = rsqrtf(q * (1.0f / FFN_D) + p.eps);
    const bool writer = m_ok && half == 0;
    if (fq == 0 && writer) { p.mean[m] = mu; p.rstd[m] = rs; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // gamma / beta (and everything else staged so far) are in LDS
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int col = 32 * ks + 8 * fq;
      const float4 g0 = *reinterpret_cast<const float4*>(lgam + col), g1 = *reinterpret_cast<const float4*>(lgam + col + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(lbet + col), b1 = *reinterpret_cast<const float4*>(lbet + col + 4);
      float4 v0 = xv[2 * ks], v1 = xv[2 * ks + 1];
      v0.x = (v0.x - mu) * rs * g0.x + b0.x; v0.y = (v0.y - mu) * rs * g0.y + b0.y;
      v0.z = (v0.z - mu) * rs * g0.z + b0.z; v0.w = (v0.w - mu) * rs * g0.w + b0.w;
      v1.x = (v1.x - mu) * rs * g1.x + b1.x; v1.y = (v1.y - mu) * rs * g1.y + b1.y;
      v1.z = (v1.z - mu) * rs * g1.z + b1.z; v1.w = (v1.w - mu) * rs * g1.w + b1.w;
      hA[ks] = pack_bf16x8(v0, v1);
      if (p.h && writer) *reinterpret_cast<bf16x8*>(p.h + (size_t)m * FFN_D + col) = hA[ks];
    }
  }

  f32x4 acc2[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside a chunk buffer: up-projection tile t (0/1) of this wave's half, K step ks; down-projection tile j
  const int w1_lane = (32 * half + fr) * 512, w2_lane = FFN_CH * FFN_D * 2 + fr * 128 + (((4 * half + fq) ^ ((fr >> 1) & 7)) << 4);
#define W1F(buf_, t_, ks_) ((ABL & 1) ? hA[ks_] : *reinterpret_cast<const bf16x8*>((buf_) + w1_lane + (t_) * (16 * 512) + (((4 * (ks_) + fq) ^ fr) << 4)))
#define W2F(buf_, j_) ((ABL & 1) ? hA[(j_) & 7] : *reinterpret_cast<const bf16x8*>((buf_) + w2_lane + (j_) * (16 * 128)))

  // Software pipeline across the chunk barrier.  A chunk's work per wave is 4 + 4 sub-steps of 4 MFMAs; every sub-step is followed
  // by the 4 fragment reads that refill the register set it has just consumed, with the fragments needed 4 sub-steps later:
  //     ... | M1(c)[s] , read W2(c)[s] | bias + Swish (c) | BARRIER c | M2(c)[s] , read W1(c+1)[s] | M1(c+1)[s] , read W2(c+1)[s] | ...
  // so LDS reads and MFMAs alternate through the whole chunk (16 fragments = 64 registers in flight, as before), and the barrier
  // sits where a wave holds all the operands of its next 16 MFMAs in registers: nobody starts a chunk with a burst of reads in
  // front of idle matrix cores.  At barrier c every wave has finished reading buffer c & 1 (its W2(c) fragments are in registers),
  // so chunk c + 2 is staged into it there - one whole chunk of time to land - and chunk c + 1, staged at barrier c - 1, is complete.
  bf16x8 f[4][4];
  bf16x8 uA;
  f32x4 acc1[2];
  auto act = [&](int cc) __attribute__((always_inline)) {           // bias + Swish of the chunk just projected: a, u out, uA kept
    const int n = cc * FFN_CH + 32 * half + 8 * fq;
    const float4 ba = *reinterpret_cast<const float4*>(lb1 + n), bb = *reinterpret_cast<const float4*>(lb1 + n + 4);
    float4 v0 = make_float4(acc1[0][0] + ba.x, acc1[0][1] + ba.y, acc1[0][2] + ba.z, acc1[0][3] + ba.w);
    float4 v1 = make_float4(acc1[1][0] + bb.x, acc1[1][1] + bb.y, acc1[1][2] + bb.z, acc1[1][3] + bb.w);
    if (p.a && m_ok && !(p.dbg & 1)) *reinterpret_cast<bf16x8*>(p.a + (size_t)m * p.FF + n) = pack_bf16x8(v0, v1);
    if constexpr (!(ABL & 4)) {
      v0.x *= sigmoidf_(v0.x); v0.y *= sigmoidf_(v0.y); v0.z *= sigmoidf_(v0.z); v0.w *= sigmoidf_(v0.w);
      v1.x *= sigmoidf_(v1.x); v1.y *= sigmoidf_(v1.y); v1.z *= sigmoidf_(v1.z); v1.w *= sigmoidf_(v1.w);
    }
    uA = pack_bf16x8(v0, v1);
    if (p.u && m_ok && !(p.dbg & 1)) *reinterpret_cast<bf16x8*>(p.u + (size_t)m * p.FF + n) = uA;
  };
  // up-projection of the chunk in `nb` from the W1 fragments in f, refilled with that chunk's W2 fragments
  auto up = [&](const unsigned char* nb) __attribute__((always_inline)) {
    acc1[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { if constexpr (ABL & 2) acc1[i & 1][0] += (float)f[s4][i][0]; else acc1[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s4][i], hA[2 * s4 + (i >> 1)], acc1[i & 1], 0, 0, 0); }
#pragma unroll
      for (int i = 0; i < 4; ++i) f[s4][i] = W2F(nb, 4 * s4 + i);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (NC > 0) {
    if (NC == 1) ffn_stage_rows(p.x, m0, p.M, smem0 + FFN_BUF, wid, lane);
    else if (!(p.dbg & 2)) ffn_stage<RG>(p.W1, p.W2, so, chunk_of(1), smem0 + FFN_BUF, wid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // chunk 0 (and the small vectors) landed; chunk 1 rides along
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int i = 0; i < 4; ++i) f[s4][i] = W1F(ffn_smem, i & 1, 2 * s4 + (i >> 1));
    up(ffn_smem);
    act(chunk_of(0));
    for (int c = 0; c + 1 < NC; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // BARRIER c
      __syncthreads();
      if (c + 2 < NC) {
        if (!(p.dbg & 2)) ffn_stage<RG>(p.W1, p.W2, so, chunk_of(c + 2), smem0 + (c & 1) * FFN_BUF, wid);
      } else {
        ffn_stage_rows(p.x, m0, p.M, smem0 + (c & 1) * FFN_BUF, wid, lane);          // c == NC - 2: the epilogue's residual rows
      }
      const unsigned char* nb = ffn_smem + ((c + 1) & 1) * FFN_BUF;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { if constexpr (ABL & 2) acc2[4 * s4 + i][0] += (float)f[s4][i][0]; else acc2[4 * s4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s4][i], uA, acc2[4 * s4 + i], 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < 4; ++i) f[s4][i] = W1F(nb, i & 1, 2 * s4 + (i >> 1));
        __builtin_amdgcn_sched_barrier(0);
      }
      up(nb);
      act(chunk_of(c + 1));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // BARRIER NC - 1: the residual rows have landed too
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int i = 0; i < 4; ++i) { if constexpr (ABL & 2) acc2[4 * s4 + i][0] += (float)f[s4][i][0]; else acc2[4 * s4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s4][i], uA, acc2[4 * s4 + i], 0, 0, 0); }
  }
#undef W1F
#undef W2F
  if (NC == 0) ffn_stage_rows(p.x, m0, p.M, smem0 + FFN_BUF, wid, lane);       // (ablation only)

  // ---- the two halves of a row group exchange partial sums: half 0 finishes output tiles 0..7, half 1 tiles 8..15
  // (register arrays are indexed by compile-time constants only: the two roles are two branches of a wave-uniform condition).
  // The residual rows sit in the buffer the last chunk did NOT use; the exchange goes through the one it did.
  if (NC == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const int xb = NC == 0 ? 1 : (NC & 1);                         // buffer holding the rows; nobody reads the other one any more
  float4* xch = reinterpret_cast<float4*>(ffn_smem + (1 - xb) * FFN_BUF);
  const unsigned char* xt = ffn_smem + xb * FFN_BUF;
  if (half == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) xch[(wid * 8 + i) * 64 + lane] = make_float4(acc2[8 + i][0], acc2[8 + i][1], acc2[8 + i][2], acc2[8 + i][3]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) xch[(wid * 8 + i) * 64 + lane] = make_float4(acc2[i][0], acc2[i][1], acc2[i][2], acc2[i][3]);
  }
  __syncthreads();
  const int pw = half ? wid - RG : wid + RG;
  float* orow = p.xo + (size_t)m * FFN_D + 4 * fq + 128 * half;
  const float* bl = lb2 + 4 * fq + 128 * half;
  const unsigned char* xrow = xt + (16 * rw + fr) * 1024;
  f32x4 mine[8];
  if (half == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) mine[i] = acc2[i];
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) mine[i] = acc2[8 + i];
  }
  float4 ov[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 o2 = xch[(pw * 8 + i) * 64 + lane];
    const float4 r = *reinterpret_cast<const float4*>(xrow + (((32 * half + 4 * i + fq) ^ fr) << 4));
    const float4 b = *reinterpret_cast<const float4*>(bl + 16 * i);
    ov[i].x = r.x + p.alpha * (mine[i][0] + o2.x + b.x); ov[i].y = r.y + p.alpha * (mine[i][1] + o2.y + b.y);
    ov[i].z = r.z + p.alpha * (mine[i][2] + o2.z + b.z); ov[i].w = r.w + p.alpha * (mine[i][3] + o2.w + b.w);
    if (m_ok) store4(orow + 16 * i, ov[i]);
  }
  if (!p.gA) return;                                  // kernel-uniform

  // ---- the LayerNorm(s) that consume xo (lid/conformer.py:81-89 PreNorm of the next module; :252-259 post_norm, then the next
  // block's first PreNorm): this wave holds 128 of a row's 256 columns, its partner (wid ^ 4) the rest; two-pass statistics,
  // the halves meeting through a small LDS array.  Same arithmetic as ln_fwd_kernel / ln2_fwd_c256_kernel.
  const float* lgA = lbet + FFN_D, *lbA = lgA + FFN_D, *lgB = lbA + FFN_D, *lbB = lgB + FFN_D;
  float* rs_x = lbet + 5 * FFN_D;                     // [4 slots][8 waves][16 rows] behind the staged vectors
  const int ccol = 128 * half + 4 * fq;               // this lane's columns: ccol + 16 i + (0..3)
  auto row_stat = [&](float v, int slot) __attribute__((always_inline)) {      // sum over the row's 256 columns
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    if (fq == 0) rs_x[(slot * 8 + wid) * 16 + fr] = v;
    __syncthreads();
    return v + rs_x[(slot * 8 + pw) * 16 + fr];
  };
  auto layer_norm = [&](float4 (&v)[8], const float* lg, const float* lb, float* mean_out, float* rstd_out, int slot)
      __attribute__((always_inline)) {
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sm += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mu = row_stat(sm, slot) * (1.0f / FFN_D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
    const float rs = rsqrtf(row_stat(q, slot + 1) * (1.0f / FFN_D) + p.eps);
    if (fq == 0 && half == 0 && m_ok) { mean_out[m] = mu; rstd_out[m] = rs; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 g = *reinterpret_cast<const float4*>(lg + ccol + 16 * i), b = *reinterpret_cast<const float4*>(lb + ccol + 16 * i);
      v[i].x = (v[i].x - mu) * rs * g.x + b.x; v[i].y = (v[i].y - mu) * rs * g.y + b.y;
      v[i].z = (v[i].z - mu) * rs * g.z + b.z; v[i].w = (v[i].w - mu) * rs * g.w + b.w;
    }
  };
  layer_norm(ov, lgA, lbA, p.meanA, p.rstdA, 0);
  if (m_ok) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t at = (size_t)m * FFN_D + ccol + 16 * i;
      if (p.yA32) store4(p.yA32 + at, ov[i]);
      if (p.yAT) store4(p.yAT + at, ov[i]);
    }
  }
  if (!p.gB) return;
  layer_norm(ov, lgB, lbB, p.meanB, p.rstdB, 2);
  if (m_ok) {
#pragma unroll
    for (int i = 0; i < 8; ++i) store4(p.yBT + (size_t)m * FFN_D + ccol + 16 * i, ov[i]);
  }
}

extern "C" int lidk_ffn_fwd_supported(int M, int d, int ff, int dtype) {
  return dtype == LIDK_BF16 && d == FFN_D && M > 0 && ff >= FFN_CH && ff % FFN_CH == 0 && 2 * FFN_BUF + (((ff + 255) & ~255) + 7 * FFN_D) * 4 + 2048 <= 160 * 1024;
}

static int ffn_fwd_launch(const float* gA, const float* bA, float* yA32, void* yAT, float* meanA, float* rstdA, const float* gB,
                          const float* bB, void* yBT, float* meanB, float* rstdB, const float* x, const void* h_in, const float* gamma, const float* beta, float eps, const void* W1,
                            const float* b1, const void* W2, const float* b2, void* h, float* mean, float* rstd, void* a, void* u,
                            float* xo, float alpha, int M, int d, int ff, int dtype, void* stream) {
  if (!x || !W1 || !b1 || !W2 || !b2 || !xo) return LIDK_ERR_ARG;
  if (!h_in && (!gamma || !beta || !mean || !rstd)) return LIDK_ERR_ARG;
  if (!lidk_ffn_fwd_supported(M, d, ff, dtype)) return LIDK_ERR_UNSUPPORTED;
  FfnFwd p{x, (const bf16*)h_in, gamma, beta, eps, (const bf16*)W1, b1, (const bf16*)W2, b2, (bf16*)h, mean, rstd, (bf16*)a,
           (bf16*)u, xo, alpha, M, ff, ffn_rot(), ffn_dbg(), gA, bA, yA32, (bf16*)yAT, meanA, rstdA, gB, bB, (bf16*)yBT, meanB, rstdB};
  const int lds = 2 * FFN_BUF + (((ff + 255) & ~255) + 7 * FFN_D) * 4 + 2048;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<true, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<false, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<true, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<false, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef LIDK_FFN_ABLATION
    (void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<true, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<true, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<true, 3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<true, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_fwd_kernel<true, 7, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
    attr_set = true;
  }
#ifdef LIDK_FFN_ABLATION
  const int abl = (p.dbg >> 5) & 7;                // LIDK_FFN_DBG bits 32 / 64 / 128: compile-time ablations of the chunk loop
  if (!h_in && abl) {
    const int grid = cdiv(M, FFN_BM);
    if (abl == 1) ffn_fwd_kernel<true, 1, 4><<<grid, 512, lds, as_stream(stream)>>>(p);
    else if (abl == 2) ffn_fwd_kernel<true, 2, 4><<<grid, 512, lds, as_stream(stream)>>>(p);
    else if (abl == 3) ffn_fwd_kernel<true, 3, 4><<<grid, 512, lds, as_stream(stream)>>>(p);
    else if (abl == 4) ffn_fwd_kernel<true, 4, 4><<<grid, 512, lds, as_stream(stream)>>>(p);
    else ffn_fwd_kernel<true, 7, 4><<<grid, 512, lds, as_stream(stream)>>>(p);
    return launch_status();
  }
#endif
  if (ffn_row_groups(M) == 3) {
    const int grid = cdiv(M, 48);
    if (h_in) ffn_fwd_kernel<false, 0, 3><<<grid, 384, lds, as_stream(stream)>>>(p);
    else ffn_fwd_kernel<true, 0, 3><<<grid, 384, lds, as_stream(stream)>>>(p);
  } else {
    const int grid = cdiv(M, FFN_BM);
    if (h_in) ffn_fwd_kernel<false, 0, 4><<<grid, 512, lds, as_stream(stream)>>>(p);
    else ffn_fwd_kernel<true, 0, 4><<<grid, 512, lds, as_stream(stream)>>>(p);
  }
  return launch_status();
}

extern "C" int lidk_ffn_fwd(const float* x, const void* h_in, const float* gamma, const float* beta, float eps, const void* W1,
                            const float* b1, const void* W2, const float* b2, void* h, float* mean, float* rstd, void* a, void* u,
                            float* xo, float alpha, int M, int d, int ff, int dtype, void* stream) {
  return ffn_fwd_launch(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, x, h_in,
                        gamma, beta, eps, W1, b1, W2, b2, h, mean, rstd, a, u, xo, alpha, M, d, ff, dtype, stream);
}

extern "C" int lidk_ffn_fwd_ln(const float* x, const void* h_in, const float* gamma, const float* beta, float eps, const void* W1,
                               const float* b1, const void* W2, const float* b2, void* h, float* mean, float* rstd, void* a, void* u,
                               float* xo, float alpha, const float* gA, const float* bA, float* yA32, void* yAT, float* meanA,
                               float* rstdA, const float* gB, const float* bB, void* yBT, float* meanB, float* rstdB, int M, int d,
                               int ff, int dtype, void* stream) {
  if (!gA || !bA || !meanA || !rstdA || (!yA32 && !yAT)) return LIDK_ERR_ARG;
  if (gB && (!bB || !yBT || !meanB || !rstdB)) return LIDK_ERR_ARG;
  return ffn_fwd_launch(gA, bA, yA32, yAT, meanA, rstdA, gB, bB, yBT, meanB, rstdB, x, h_in, gamma, beta, eps, W1, b1, W2, b2, h, mean,
                        rstd, a, u, xo, alpha, M, d, ff, dtype, stream);
}

// =====================================================================================================================
// Backward of the same module, data path:  given dyT = alpha * d(loss)/d(xo) (T) and the saved pre-activation a,
//     da = (dyT . W2) * swish'(a)            [M][ff]  (T, stored: operand of the weight gradients dW1 = da^T h, db1)
//     dh = da . W1                           [M][256]
//     dx = dres + LN'(dh; x, mean, rstd, gamma)   (f32) and dxT = dxT_scale * dx (T); per-workgroup (dgamma | dbeta) rows
// Same decomposition as the forward kernel with the roles of the weights exchanged: W2T [ff][256] (row n = W2[:, n]) feeds the
// first product, W1T [256][ff] (row j = W1[:, j]) the second; the chunk's [64][64] tile of a arrives by LDS-DMA beside them, so the
// loop holds no register-destination load (hipcc would drain the DMA queue in front of its use).  The workgroup owns complete rows,
// so the PreNorm's backward runs in the epilogue: dh never exists in HBM (LN_OUT = false writes dh instead, for the site whose
// LayerNorm backward is fused with the neighbouring block's post_norm).  Replaces 2 x lidk_gemm_nt + lidk_layernorm_bwd.
// =====================================================================================================================
#define FFN_ATILE (FFN_BM * FFN_CH * 2)

struct FfnBwd {
  const bf16* dyT; const bf16* a; const bf16* W2T; const bf16* W1T; bf16* da;
  const float* x; const float* mean; const float* rstd; const float* gamma; const float* dres;
  float* dx; bf16* dxT; float dxT_scale; float* partial; bf16* dh;
  int M; int FF; int rot; int dbg;
  // optional SECOND LayerNorm backward in the epilogue (the pair post_norm -> this PreNorm): the gradient the first one yields
  // (+ dres) is the upstream gradient of LN1 whose input rows are x1: dx = LN1'(dv; x1, mean1, rstd1, gamma1); partial1 its rows
  const float* x1; const float* mean1; const float* rstd1; const float* gamma1; float* partial1;
};

// GEMM = true: only the second product - dh = A . W with A = p.a [M][K = p.FF] (T) and p.W1T = W^T [256][K] - followed by the same
// epilogue: the data gradient of any Linear / 1x1 conv that feeds a PreNorm (N = 256), fused with that LayerNorm's backward.
template <bool LN_OUT, bool GEMM, int RG>
__global__ void __launch_bounds__(128 * RG) __attribute__((amdgpu_waves_per_eu(2, 2)))
ffn_bwd_kernel(FfnBwd p) {
  constexpr int NW = 2 * RG, BM = 16 * RG, NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char ffn_smem[];
  const unsigned smem0 = (unsigned)(size_t)(lds_void_t*)ffn_smem;
  const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), half = wid >= RG ? 1 : 0, rw = wid - RG * half;
  const int m0 = blockIdx.x * BM;
  const int m_raw = m0 + 16 * rw + fr;
  const bool m_ok = m_raw < p.M;
  const int m = m_ok ? m_raw : p.M - 1;
  const int NC = p.FF / FFN_CH;
  const int rot = p.rot ? (blockIdx.x >> 3) % NC : 0;  // rotated chunk order, as in the forward kernel
  auto chunk_of = [&](int c) __attribute__((always_inline)) { const int t = c + rot; return t >= NC ? t - NC : t; };

  // register loads first (one HBM round trip in the prologue)
  bf16x8 dyA[8];
  if (!GEMM) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) dyA[ks] = *reinterpret_cast<const bf16x8*>(p.dyT + (size_t)m * FFN_D + 32 * ks + 8 * fq);
  }

  // the chunk's tile of a: wave w copies rows 8 w .. + 7 (128 B each), 16-byte chunks XOR-permuted by (row & 7)
  const int a_row = 8 * wid + (lane >> 3), a_cc = (lane & 7) ^ (a_row & 7);
  const FfnStageOff<RG> so = ffn_stage_offsets<RG>(p.FF, wid, lane);
  const unsigned a_off = (unsigned)((8 * wid + (lane >> 3)) * p.FF + a_cc * 8) * 2;      // relative to the workgroup's first row of a
  const bf16* a_base = p.a + (size_t)m0 * p.FF;
  const bool a_tail = m0 + BM > p.M;                     // last workgroup of a ragged M: per-lane pointers with the row clamped
  const bf16* a_src = p.a + (size_t)min(m0 + a_row, p.M - 1) * p.FF + a_cc * 8;
  auto stage = [&](int c, int b) __attribute__((always_inline)) {
    if (GEMM) ffn_stage_cols<RG>(p.W1T + (size_t)c * FFN_CH, so, smem0 + b * FFN_BUF + FFN_CH * FFN_D * 2, wid);
    else ffn_stage<RG>(p.W2T, p.W1T, so, c, smem0 + b * FFN_BUF, wid);
    if (a_tail) glds16(a_src + c * FFN_CH, smem0 + 2 * FFN_BUF + b * FFN_ATILE + wid * 1024);
    else glds16s(a_base + c * FFN_CH, a_off, smem0 + 2 * FFN_BUF + b * FFN_ATILE + wid * 1024);
  };
  stage(chunk_of(0), 0);

  f32x4 acc2[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc2[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int w1_lane = (32 * half + fr) * 512, w2_lane = FFN_CH * FFN_D * 2 + fr * 128 + (((4 * half + fq) ^ ((fr >> 1) & 7)) << 4);
  const int at_lane = (16 * rw + fr) * 128 + (((4 * half + fq) ^ (fr & 7)) << 4);
#define W1F(buf_, t_, ks_) (*reinterpret_cast<const bf16x8*>((buf_) + w1_lane + (t_) * (16 * 512) + (((4 * (ks_) + fq) ^ fr) << 4)))
#define W2F(buf_, j_) (*reinterpret_cast<const bf16x8*>((buf_) + w2_lane + (j_) * (16 * 128)))

  if (GEMM) {
    // one product per chunk: 16 fragments of W^T, the wave's K slice of the chunk's A tile from LDS, 16 MFMAs
    for (int c = 0; c < NC; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (c + 1 < NC) {
        stage(chunk_of(c + 1), (c + 1) & 1);
      } else if (LN_OUT) {
        ffn_stage_rows(p.x, m0, p.M, smem0 + ((c + 1) & 1) * FFN_BUF, wid, lane);
      }
      const unsigned char* buf = ffn_smem + (c & 1) * FFN_BUF;
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(ffn_smem + 2 * FFN_BUF + (c & 1) * FFN_ATILE + at_lane);
      bf16x8 fw[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) fw[j] = W2F(buf, j);
#pragma unroll
      for (int j = 0; j < 16; ++j) acc2[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], av, acc2[j], 0, 0, 0);
    }
  } else {
  // the forward kernel's software pipeline across the chunk barrier (see there): | M1(c)[s] , read W1T(c)[s] | swish'(a) (c) |
  // BARRIER c | M2(c)[s] , read W2T(c+1)[s] | M1(c+1)[s] , read W1T(c+1)[s] | ...
  bf16x8 f[4][4];
  bf16x8 daA;
  f32x4 acc1[2];
  auto act = [&](int cc, int b) __attribute__((always_inline)) {     // da = (dyT . W2) * swish'(a) of the chunk in buffer b
    const bf16x8 av = *reinterpret_cast<const bf16x8*>(ffn_smem + 2 * FFN_BUF + b * FFN_ATILE + at_lane);
    float dv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float a = (float)av[i], sg = sigmoidf_(a);
      dv[i] = acc1[i >> 2][i & 3] * (sg * (1.f + a * (1.f - sg)));
    }
    daA = pack_bf16x8(make_float4(dv[0], dv[1], dv[2], dv[3]), make_float4(dv[4], dv[5], dv[6], dv[7]));
    if (m_ok && !(p.dbg & 1)) *reinterpret_cast<bf16x8*>(p.da + (size_t)m * p.FF + cc * FFN_CH + 32 * half + 8 * fq) = daA;
  };
  auto up = [&](const unsigned char* nb) __attribute__((always_inline)) {
    acc1[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc1[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s4][i], dyA[2 * s4 + (i >> 1)], acc1[i & 1], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) f[s4][i] = W2F(nb, 4 * s4 + i);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  {
    if (NC == 1) { if (LN_OUT) ffn_stage_rows(p.x, m0, p.M, smem0 + FFN_BUF, wid, lane); }
    else if (!(p.dbg & 2)) stage(chunk_of(1), 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int i = 0; i < 4; ++i) f[s4][i] = W1F(ffn_smem, i & 1, 2 * s4 + (i >> 1));
    up(ffn_smem);
    act(chunk_of(0), 0);
    for (int c = 0; c + 1 < NC; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // BARRIER c
      __syncthreads();
      if (c + 2 < NC) {
        if (!(p.dbg & 2)) stage(chunk_of(c + 2), c & 1);
      } else if (LN_OUT) {
        ffn_stage_rows(p.x, m0, p.M, smem0 + (c & 1) * FFN_BUF, wid, lane);           // c == NC - 2: the LayerNorm backward's rows of x
      }
      const unsigned char* nb = ffn_smem + ((c + 1) & 1) * FFN_BUF;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[4 * s4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s4][i], daA, acc2[4 * s4 + i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[s4][i] = W1F(nb, i & 1, 2 * s4 + (i >> 1));
        __builtin_amdgcn_sched_barrier(0);
      }
      up(nb);
      act(chunk_of(c + 1), (c + 1) & 1);
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc2[4 * s4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s4][i], daA, acc2[4 * s4 + i], 0, 0, 0);
  }
  }
#undef W1F
#undef W2F

  // ---- halves of a row group exchange partial sums: half h keeps output tiles 8 h .. 8 h + 7 (columns 128 h + 16 i + 4 fq + r).
  // The rows of x sit in the buffer the last chunk did not use; the exchange goes through the one it did.  The epilogue's HBM
  // operands (dres, gamma, mean, rstd) are requested before the barriers so that their latency hides behind the exchange.
  const int col0 = 128 * half + 4 * fq;               // this lane's columns: col0 + 16 i + (0..3)
  float4 rv[8], gmv[8];
  float mu = 0.f, rs = 0.f;
  if (LN_OUT) {
    mu = p.mean[m]; rs = p.rstd[m];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      rv[i] = p.dres ? load4(p.dres + (size_t)m * FFN_D + col0 + 16 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
      gmv[i] = load4(p.gamma + col0 + 16 * i);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int xb = NC & 1;
  float4* xch = reinterpret_cast<float4*>(ffn_smem + (1 - xb) * FFN_BUF);
  const unsigned char* xrow = ffn_smem + xb * FFN_BUF + (16 * rw + fr) * 1024;
  if (half == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) xch[(wid * 8 + i) * 64 + lane] = make_float4(acc2[8 + i][0], acc2[8 + i][1], acc2[8 + i][2], acc2[8 + i][3]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) xch[(wid * 8 + i) * 64 + lane] = make_float4(acc2[i][0], acc2[i][1], acc2[i][2], acc2[i][3]);
  }
  __syncthreads();
  const int pw = half ? wid - RG : wid + RG;
  float4 dhv[8];
  if (half == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dhv[i] = make_float4(acc2[i][0], acc2[i][1], acc2[i][2], acc2[i][3]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) dhv[i] = make_float4(acc2[8 + i][0], acc2[8 + i][1], acc2[8 + i][2], acc2[8 + i][3]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 o = xch[(pw * 8 + i) * 64 + lane];
    dhv[i].x += o.x; dhv[i].y += o.y; dhv[i].z += o.z; dhv[i].w += o.w;
  }
  if (!LN_OUT) {
    if (m_ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) store4(p.dh + (size_t)m * FFN_D + col0 + 16 * i, dhv[i]);
    }
    return;
  }

  // ---- LayerNorm backward of the complete rows (once, or twice for the pair post_norm -> this PreNorm)
  float4 xin[8], x1v[8], g1v[8];
  float mu1 = 0.f, rs1 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) xin[i] = *reinterpret_cast<const float4*>(xrow + (((32 * half + 4 * i + fq) ^ fr) << 4));
  if (p.x1) {                                        // kernel-uniform; requested now, used after the first LayerNorm's backward
    mu1 = p.mean1[m]; rs1 = p.rstd1[m];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x1v[i] = load4(p.x1 + (size_t)m * FFN_D + col0 + 16 * i);
      g1v[i] = load4(p.gamma1 + col0 + 16 * i);
    }
  }
  __syncthreads();                                   // everybody has read the exchanged tiles and its rows of x: the area is free
  float2* rsum = reinterpret_cast<float2*>(ffn_smem);                 // [2 rounds][wave][16 rows]
  float* dgs = reinterpret_cast<float*>(ffn_smem + 2048);             // per-column (dgamma | dbeta) terms: [row][256] each
  float* dbs = dgs + BM * FFN_D;
  const int lrow = 16 * rw + fr;
  // dy: upstream gradient at the LayerNorm's output (this lane's 8 x 4 columns); xin: its input rows; -> gradient at its input,
  // and this workgroup's partial (dgamma | dbeta) row
  auto ln_bwd = [&](float4 (&dy)[8], const float4 (&xin)[8], const float4 (&gm)[8], float mu_, float rs_, float* part, int round)
      __attribute__((always_inline)) {
    float4 xh[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xh[i].x = (xin[i].x - mu_) * rs_; xh[i].y = (xin[i].y - mu_) * rs_; xh[i].z = (xin[i].z - mu_) * rs_; xh[i].w = (xin[i].w - mu_) * rs_;
      float4 tg = make_float4(0.f, 0.f, 0.f, 0.f), tb = tg;
      if (m_ok) {
        tg = make_float4(dy[i].x * xh[i].x, dy[i].y * xh[i].y, dy[i].z * xh[i].z, dy[i].w * xh[i].w);
        tb = dy[i];
      }
      *reinterpret_cast<float4*>(dgs + lrow * FFN_D + col0 + 16 * i) = tg;
      *reinterpret_cast<float4*>(dbs + lrow * FFN_D + col0 + 16 * i) = tb;
      dy[i].x *= gm[i].x; dy[i].y *= gm[i].y; dy[i].z *= gm[i].z; dy[i].w *= gm[i].w;       // g = dy * gamma
      s1 += (dy[i].x + dy[i].y) + (dy[i].z + dy[i].w);
      s2 += (dy[i].x * xh[i].x + dy[i].y * xh[i].y) + (dy[i].z * xh[i].z + dy[i].w * xh[i].w);
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    if (fq == 0) rsum[(round * 8 + wid) * 16 + fr] = make_float2(s1, s2);
    __syncthreads();
    const float2 o = rsum[(round * 8 + pw) * 16 + fr];
    const float m1 = (s1 + o.x) * (1.0f / FFN_D), m2 = (s2 + o.y) * (1.0f / FFN_D);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dy[i].x = rs_ * (dy[i].x - m1 - xh[i].x * m2); dy[i].y = rs_ * (dy[i].y - m1 - xh[i].y * m2);
      dy[i].z = rs_ * (dy[i].z - m1 - xh[i].z * m2); dy[i].w = rs_ * (dy[i].w - m1 - xh[i].w * m2);
    }
    // column sums over the workgroup's rows: item t -> (which = t >> 8, column = t & 255)
    for (int t0 = tid; t0 < 2 * FFN_D; t0 += NT) {
      const float* src = (t0 >> 8) ? dbs : dgs;
      const int col = t0 & 255;
      float t = 0.f;
#pragma unroll 8
      for (int r = 0; r < BM; ++r) t += src[r * FFN_D + col];
      part[(size_t)blockIdx.x * 2 * FFN_D + t0] = t;
    }
  };
  ln_bwd(dhv, xin, gmv, mu, rs, p.partial, 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) { dhv[i].x += rv[i].x; dhv[i].y += rv[i].y; dhv[i].z += rv[i].z; dhv[i].w += rv[i].w; }
  if (p.x1) {                                        // the gradient just formed is LN1's upstream gradient
    __syncthreads();                                 // the column sums of the first round are done with dgs / dbs
    ln_bwd(dhv, x1v, g1v, mu1, rs1, p.partial1, 1);
  }
  if (m_ok) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t at = (size_t)m * FFN_D + col0 + 16 * i;
      float4 o4 = dhv[i];
      if (p.dx) store4(p.dx + at, o4);
      if (p.dxT) {
        o4.x *= p.dxT_scale; o4.y *= p.dxT_scale; o4.z *= p.dxT_scale; o4.w *= p.dxT_scale;
        store4(p.dxT + at, o4);
      }
    }
  }
}

extern "C" int lidk_ffn_bwd_partial_rows(int M) { return cdiv(M, 16 * ffn_row_groups(M)); }

static int ffn_bwd_launch(const float* x1, const float* mean1, const float* rstd1, const float* gamma1, float* partial1,
                          const void* dyT, const void* a, const void* W2T, int ldw2t, const void* W1T, int ldw1t, void* da,
                          const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx,
                          void* dxT, float dxT_scale, float* partial, void* dh, int M, int d, int ff, int dtype, void* stream) {
  if (!dyT || !a || !W2T || !W1T || !da) return LIDK_ERR_ARG;
  if (!dh && (!x || !mean || !rstd || !gamma || !partial || (!dx && !dxT))) return LIDK_ERR_ARG;
  if (!lidk_ffn_fwd_supported(M, d, ff, dtype) || ldw2t != FFN_D || ldw1t != ff) return LIDK_ERR_UNSUPPORTED;
  FfnBwd p{(const bf16*)dyT, (const bf16*)a, (const bf16*)W2T, (const bf16*)W1T, (bf16*)da, x, mean, rstd, gamma, dres, dx, (bf16*)dxT,
           dxT_scale, partial, (bf16*)dh, M, ff, ffn_rot(), ffn_dbg(), x1, mean1, rstd1, gamma1, partial1};
  const int lds = 2 * FFN_BUF + 2 * FFN_ATILE;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)ffn_bwd_kernel<true, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_bwd_kernel<false, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_bwd_kernel<true, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_bwd_kernel<false, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (ffn_row_groups(M) == 3) {
    if (dh) ffn_bwd_kernel<false, false, 3><<<cdiv(M, 48), 384, lds, as_stream(stream)>>>(p);
    else ffn_bwd_kernel<true, false, 3><<<cdiv(M, 48), 384, lds, as_stream(stream)>>>(p);
  } else {
    if (dh) ffn_bwd_kernel<false, false, 4><<<cdiv(M, FFN_BM), 512, lds, as_stream(stream)>>>(p);
    else ffn_bwd_kernel<true, false, 4><<<cdiv(M, FFN_BM), 512, lds, as_stream(stream)>>>(p);
  }
  return launch_status();
}

extern "C" int lidk_ffn_bwd(const void* dyT, const void* a, const void* W2T, int ldw2t, const void* W1T, int ldw1t, void* da,
                            const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx,
                            void* dxT, float dxT_scale, float* partial, void* dh, int M, int d, int ff, int dtype, void* stream) {
  return ffn_bwd_launch(nullptr, nullptr, nullptr, nullptr, nullptr, dyT, a, W2T, ldw2t, W1T, ldw1t, da, x, mean, rstd, gamma, dres, dx,
                        dxT, dxT_scale, partial, dh, M, d, ff, dtype, stream);
}

extern "C" int lidk_ffn_bwd_ln2(const void* dyT, const void* a, const void* W2T, int ldw2t, const void* W1T, int ldw1t, void* da,
                                const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                                const float* x1, const float* mean1, const float* rstd1, const float* gamma1, float* dx, void* dxT,
                                float dxT_scale, float* partial, float* partial1, int M, int d, int ff, int dtype, void* stream) {
  if (!x1 || !mean1 || !rstd1 || !gamma1 || !partial1) return LIDK_ERR_ARG;
  return ffn_bwd_launch(x1, mean1, rstd1, gamma1, partial1, dyT, a, W2T, ldw2t, W1T, ldw1t, da, x, mean, rstd, gamma, dres, dx, dxT,
                        dxT_scale, partial, nullptr, M, d, ff, dtype, stream);
}

extern "C" int lidk_dgrad_ln_bwd_supported(int M, int N, int K, int dtype) {
  return dtype == LIDK_BF16 && N == FFN_D && M > 0 && K >= FFN_CH && K % FFN_CH == 0;
}

extern "C" int lidk_dgrad_ln_bwd(const void* dy, const void* WT, int ldwt, const float* x, const float* mean, const float* rstd,
                                 const float* gamma, const float* dres, float* dx, void* dxT, float dxT_scale, float* partial, int M,
                                 int N, int K, int dtype, void* stream) {
  if (!dy || !WT || !x || !mean || !rstd || !gamma || !partial || (!dx && !dxT)) return LIDK_ERR_ARG;
  if (!lidk_dgrad_ln_bwd_supported(M, N, K, dtype) || ldwt != K) return LIDK_ERR_UNSUPPORTED;
  FfnBwd p{nullptr, (const bf16*)dy, nullptr, (const bf16*)WT, nullptr, x, mean, rstd, gamma, dres, dx, (bf16*)dxT, dxT_scale, partial,
           nullptr, M, K, ffn_rot(), ffn_dbg(), nullptr, nullptr, nullptr, nullptr, nullptr};
  const int lds = 2 * FFN_BUF + 2 * FFN_ATILE;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)ffn_bwd_kernel<true, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_bwd_kernel<true, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (ffn_row_groups(M) == 3) ffn_bwd_kernel<true, true, 3><<<cdiv(M, 48), 384, lds, as_stream(stream)>>>(p);
  else ffn_bwd_kernel<true, true, 4><<<cdiv(M, FFN_BM), 512, lds, as_stream(stream)>>>(p);
  return launch_status();
}
