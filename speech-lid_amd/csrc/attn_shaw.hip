// Key-tiled MFMA attention with Shaw relative positions for LONG sequences (lid/conformer.py:117-148), bf16, dh in {32, 64}:
//     S[i][j] = scale * ( q_i . k_j + q_i . E[clamp(i - j, -P, P) + P] ) ;  A = softmax_j(S) ;  O = A . V
// The resident kernels of attn.hip keep a whole (batch, head)'s K, V and relative-embedding slice in LDS (T <= 256); beyond that
// the first implementation fell to the VALU kernels of attn_long.hip - 3.6-4.1 ms per launch at T ~ 500 against 15 us at T = 151
// (profiles/r03/conformer_ragged_kernel_stats.csv).  This file is the decomposition of xattn.hip with the relative term added:
//   workgroup = 64 query rows of one (batch, head), 4 waves x 16 rows; K / V / the E window staged one 64-key tile at a time;
//   S^T tiles by MFMA (A = K rows, B = Q rows): a lane holds, for ITS query i = fr, keys 4 fq + r of four 16-key tiles;
//   relative term: for the wave's 16 queries and the tile's 64 keys the offsets i - j span 79 consecutive values, so
//   R^T = E_window . Q^T is five more 16 x 16 tiles (same orientation: lane = query fr, window rows 4 fq + r); the skew
//   S^T[j][i] += R^T[(i - j) - base][i] goes through a per-wave [16][84] f32 LDS tile (write float4 per tile, read 4 scalars per
//   key tile: the source register index depends on the lane, which rules out a single ds_bpermute).
// The interface is the long path's (lidk_attn_fwd / lidk_attn_bwd with STORED probabilities, row pitch ldp): the forward makes two
// passes over the key tiles - row statistics, then normalised probabilities (written, bf16) and O - so the backward kernels of
// attn_long.hip keep working on its output.  Parity: tests/test_gpu_ops.py::test_attention_long_sequences (T = 420 ... 1100).
#include "common.h"

#define SH_TILE 64
#define SH_EROWS 144                // E window rows staged per key tile: offsets I0 - jt0 - 63 .. + 143 (127 used)
#define SH_RLD 84                   // row pitch (floats) of the per-wave skew tile [16][80]

struct ShGeom { int B, T, H, max_pos, inner, ld; float scale; };

template <int DH>
__device__ __forceinline__ void sh_stage_rows(bf16* dst, const bf16* src, size_t row_stride, int row0, int nrows_valid) {
  constexpr int CH = DH / 8, LDK = DH + 8;
  for (int c = threadIdx.x; c < SH_TILE * CH; c += blockDim.x) {
    const int row = c / CH, dc = (c % CH) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + row < nrows_valid) v = *reinterpret_cast<const uint4*>(src + (size_t)(row0 + row) * row_stride + dc);
    *reinterpret_cast<uint4*>(&dst[row * LDK + dc]) = v;
  }
}
// Es[x] = E[clamp(o0 + x, -P, P) + P], x = 0 .. SH_EROWS - 1 (embT: bf16 copy of rel_pos_emb.weight, [2P + 1][DH])
template <int DH>
__device__ __forceinline__ void sh_stage_e(bf16* Es, const bf16* __restrict__ embT, int o0, int P) {
  constexpr int CH = DH / 8, LDK = DH + 8;
  for (int c = threadIdx.x; c < SH_EROWS * CH; c += blockDim.x) {
    const int x = c / CH, dc = (c % CH) * 8;
    const int row = max(-P, min(P, o0 + x)) + P;
    *reinterpret_cast<uint4*>(&Es[x * LDK + dc]) = *reinterpret_cast<const uint4*>(embT + (size_t)row * DH + dc);
  }
}
__device__ __forceinline__ bf16x8 sh_pack(const float* a, const float* b) {
  union { bf16 e[8]; bf16x8 v; } u;
#pragma unroll
  for (int r = 0; r < 4; ++r) { u.e[r] = (bf16)a[r]; u.e[4 + r] = (bf16)b[r]; }
  return u.v;
}

// scaled scores of this lane's query (i = fr of the wave's 16 rows) against the 64 staged keys: s[t][r] <-> key jt0 + 16 t + 4 fq + r
template <int DH>
__device__ __forceinline__ void sh_scores(float (&s)[4][4], const bf16* Ks, const bf16* Es, float* R, const bf16x8 (&qf)[DH / 32],
                                          int wave, int fr, int fq, int jt0, int T_, float scale) {
  constexpr int LDK = DH + 8, KS = DH / 32;
#pragma unroll
  for (int et = 0; et < 5; ++et) {
    f32x4 racc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 ef = *reinterpret_cast<const bf16x8*>(&Es[(16 * wave + 16 * et + fr) * LDK + ks * 32 + 8 * fq]);
      racc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ef, qf[ks], racc, 0, 0, 0);
    }
    *reinterpret_cast<float4*>(&R[fr * SH_RLD + 16 * et + 4 * fq]) = make_float4(racc[0], racc[1], racc[2], racc[3]);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(16 * t + fr) * LDK + ks * 32 + 8 * fq]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], acc, 0, 0, 0);
    }
    const float* rrow = R + fr * SH_RLD + fr + 63 - 16 * t - 4 * fq;          // window index of (i, key 16 t + 4 fq) ; - r below
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = jt0 + 16 * t + 4 * fq + r;
      s[t][r] = (j < T_) ? (acc[r] + rrow[-r]) * scale : -INFINITY;
    }
  }
}

template <int DH>
__global__ void __launch_bounds__(256)
attn_fwd_shaw_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ embT, bf16* __restrict__ out, bf16* __restrict__ probs,
                     int ldp, ShGeom g) {
  constexpr int LDK = DH + 8, KS = DH / 32, NT = DH / 16;
  __shared__ __attribute__((aligned(16))) bf16 Ks[SH_TILE * LDK];
  __shared__ __attribute__((aligned(16))) bf16 Vs[SH_TILE * LDK];
  __shared__ __attribute__((aligned(16))) bf16 Es[SH_EROWS * LDK];
  __shared__ __attribute__((aligned(16))) float Rl[4 * 16 * SH_RLD];
  const int T_ = g.T, H = g.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh % H, I0 = blockIdx.y * SH_TILE;
  const bf16* base = qkv + (size_t)b * T_ * g.ld + h * DH;
  const int i = I0 + 16 * wave + fr, ic = min(i, T_ - 1);
  float* R = Rl + wave * 16 * SH_RLD;
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)ic * g.ld + ks * 32 + 8 * fq);

  // ---- pass 1: row maximum and sum
  float m = -INFINITY, l = 0.f;
  for (int jt0 = 0; jt0 < T_; jt0 += SH_TILE) {
    __syncthreads();
    sh_stage_rows<DH>(Ks, base + g.inner, g.ld, jt0, T_);
    sh_stage_e<DH>(Es, embT, I0 - jt0 - 63, g.max_pos);
    __syncthreads();
    float s[4][4];
    sh_scores<DH>(s, Ks, Es, R, qf, wave, fr, fq, jt0, T_, g.scale);
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);                               // finite: key jt0 < T is never masked
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) psum += __expf(s[t][r] - m_new);
    l = l * ((m == -INFINITY) ? 0.f : __expf(m - m_new)) + psum;
    m = m_new;
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float lse = m + __logf(l);

  // ---- pass 2: normalised probabilities (stored for the backward) and O = P . V
  f32x4 O[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) O[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16* prow = probs ? probs + ((size_t)bh * T_ + ic) * ldp : nullptr;
  for (int jt0 = 0; jt0 < T_; jt0 += SH_TILE) {
    __syncthreads();
    sh_stage_rows<DH>(Ks, base + g.inner, g.ld, jt0, T_);
    sh_stage_rows<DH>(Vs, base + 2 * g.inner, g.ld, jt0, T_);
    sh_stage_e<DH>(Es, embT, I0 - jt0 - 63, g.max_pos);
    __syncthreads();
    float s[4][4];
    sh_scores<DH>(s, Ks, Es, R, qf, wave, fr, fq, jt0, T_, g.scale);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s[t][r] = __expf(s[t][r] - lse);             // keys >= T: exp(-inf) = 0
      if (prow && i < T_) {
        const int j = jt0 + 16 * t + 4 * fq;
        if (j + 3 < ldp && !(ldp & 3)) {
          union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
          for (int r = 0; r < 4; ++r) pk.e[r] = (bf16)s[t][r];
          *reinterpret_cast<uint2*>(prow + j) = pk.u;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (j + r < ldp) prow[j + r] = (bf16)s[t][r];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bf16x8 pf = sh_pack(s[2 * c], s[2 * c + 1]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        O[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, tr_frag_split(Vs, LDK, 32 * c, nt * 16, fq, fr), O[nt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int io = I0 + 16 * wave + 4 * fq + r;
    if (io < T_) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) out[((size_t)b * T_ + io) * g.inner + h * DH + nt * 16 + fr] = (bf16)O[nt][r];
    }
  }
}

// host side (called from lidk_attn_fwd, attn.hip)
int att_shaw_fwd(const void* qkv, const void* embT, void* out, void* probs, int ldp, int B, int T_, int H, int dh, int max_pos,
                 hipStream_t s) {
  ShGeom g{B, T_, H, max_pos, H * dh, 3 * H * dh, 1.0f / sqrtf((float)dh)};
  const dim3 grid(B * H, cdiv(T_, SH_TILE));
  if (dh == 64) attn_fwd_shaw_kernel<64><<<grid, 256, 0, s>>>((const bf16*)qkv, (const bf16*)embT, (bf16*)out, (bf16*)probs, ldp, g);
  else if (dh == 32) attn_fwd_shaw_kernel<32><<<grid, 256, 0, s>>>((const bf16*)qkv, (const bf16*)embT, (bf16*)out, (bf16*)probs, ldp, g);
  else return LIDK_ERR_UNSUPPORTED;
  return launch_status();
}
