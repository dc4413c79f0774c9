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

// ------------------------------------------------------------------------------------ backward: dS rows and dQ
// Same decomposition, STORED probabilities (the forward above wrote them):
//   pass A (V tiles): dP^T = V . dO^T by MFMA, delta_i = sum_j P_ij dP_ij;
//   pass B (K / V / E tiles): dS = P (dP - delta) -> dscores [B][H][T][T] f32 (unscaled: the key-side kernel of attn_long.hip reads
//           it for dK, dV and the relative-embedding gradient), dQ += dS . K (the dS accumulators, packed, are the A operand) and
//           dQ += dR . E_window, where dR[i][e] = dS[i][j = i - base - e] is scattered into a per-wave [16][96] bf16 LDS tile
//           (each (i, e) is hit by exactly one key of the tile, the rest of the zeroed tile stays 0) and read back as A fragments.
#define SH_DLD 104                  // row pitch (bf16) of the per-wave dR tile [16][96]

template <int DH>
__global__ void __launch_bounds__(256)
attn_bwd_rows_shaw_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ embT, const bf16* __restrict__ probs, int ldp,
                          const bf16* __restrict__ dout, bf16* __restrict__ dqkv, float* __restrict__ dscores, ShGeom g) {
  constexpr int LDK = DH + 8, KS = DH / 32, NT = DH / 16;
  __shared__ __attribute__((aligned(16))) bf16 Ks[SH_TILE * LDK];
  __shared__ __attribute__((aligned(16))) bf16 Vs[SH_TILE * LDK];
  __shared__ __attribute__((aligned(16))) bf16 Es[SH_EROWS * LDK];
  __shared__ __attribute__((aligned(16))) bf16 Dl[4 * 16 * SH_DLD];
  const int T_ = g.T, H = g.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh % H, I0 = blockIdx.y * SH_TILE;
  const bf16* base = qkv + (size_t)b * T_ * g.ld + h * DH;
  const int i = I0 + 16 * wave + fr, ic = min(i, T_ - 1);
  bf16* D = Dl + wave * 16 * SH_DLD;
  bf16x8 dof[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    dof[ks] = *reinterpret_cast<const bf16x8*>(dout + ((size_t)b * T_ + ic) * g.inner + h * DH + ks * 32 + 8 * fq);
  const bf16* prow = probs + ((size_t)bh * T_ + ic) * ldp;
  const bool pvec = !(ldp & 3);
  // P of this lane's query against keys jt0 + 16 t + 4 fq + r (0 beyond T)
  auto load_p = [&](float (&pv)[4][4], int jt0) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = jt0 + 16 * t + 4 * fq;
      if (pvec && j + 3 < T_) {
        const float4 v = load4(prow + j);
        pv[t][0] = v.x; pv[t][1] = v.y; pv[t][2] = v.z; pv[t][3] = v.w;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) pv[t][r] = (j + r < T_) ? (float)prow[j + r] : 0.f;
      }
    }
  };
  auto dp_tiles = [&](float (&dp)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vs[(16 * t + fr) * LDK + ks * 32 + 8 * fq]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dp[t][r] = acc[r];
    }
  };

  // ---- pass A: delta_i = sum_j P_ij * dP_ij
  float delta = 0.f;
  for (int jt0 = 0; jt0 < T_; jt0 += SH_TILE) {
    __syncthreads();
    sh_stage_rows<DH>(Vs, base + 2 * g.inner, g.ld, jt0, T_);
    __syncthreads();
    float pv[4][4], dp[4][4];
    load_p(pv, jt0);
    dp_tiles(dp);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) delta = fmaf(pv[t][r], dp[t][r], delta);
  }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);

  // ---- pass B: dS (stored), dQ
  f32x4 dq[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) dq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float* dsrow = dscores + ((size_t)bh * T_ + ic) * T_;
  const bool svec = !(T_ & 3);
  for (int jt0 = 0; jt0 < T_; jt0 += SH_TILE) {
    __syncthreads();
    sh_stage_rows<DH>(Ks, base + g.inner, g.ld, jt0, T_);
    sh_stage_rows<DH>(Vs, base + 2 * g.inner, g.ld, jt0, T_);
    sh_stage_e<DH>(Es, embT, I0 - jt0 - 63, g.max_pos);
    // zero this wave's dR tile (16 x 104 bf16 = 208 16-byte pieces)
    for (int c = lane; c < 16 * SH_DLD / 8; c += 64) *reinterpret_cast<uint4*>(&D[c * 8]) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    float pv[4][4], ds[4][4];
    load_p(pv, jt0);
    dp_tiles(ds);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[t][r] = pv[t][r] * (ds[t][r] - delta);
      const int j = jt0 + 16 * t + 4 * fq;
      if (i < T_) {
        if (svec && j + 3 < T_) {
          *reinterpret_cast<float4*>(dsrow + j) = make_float4(ds[t][0], ds[t][1], ds[t][2], ds[t][3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (j + r < T_) dsrow[j + r] = ds[t][r];
        }
      }
      // scatter into dR[i = fr][e = fr + 63 - (16 t + 4 fq + r)]
      bf16* drow = D + fr * SH_DLD + fr + 63 - 16 * t - 4 * fq;
#pragma unroll
      for (int r = 0; r < 4; ++r) drow[-r] = (bf16)ds[t][r];
    }
    // dQ += dS . K : the dS accumulators (two stacked 16-key tiles, slot-permuted) are the A operand
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bf16x8 sf = sh_pack(ds[2 * c], ds[2 * c + 1]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        dq[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, tr_frag_split(Ks, LDK, 32 * c, nt * 16, fq, fr), dq[nt], 0, 0, 0);
    }
    // dQ += dR . E_window : A = dR rows (LDS), B = E_window^T by transposed reads; 96 window rows = 3 K steps
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3) {
      const bf16x8 rf = *reinterpret_cast<const bf16x8*>(&D[fr * SH_DLD + 32 * k3 + 8 * fq]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        dq[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rf, tr_frag(Es + 16 * wave * LDK, LDK, 32 * k3, nt * 16, fq, fr), dq[nt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int io = I0 + 16 * wave + 4 * fq + r;
    if (io < T_) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) dqkv[((size_t)b * T_ + io) * g.ld + h * DH + nt * 16 + fr] = (bf16)(dq[nt][r] * g.scale);
    }
  }
}

// ------------------------------------------------------------------------------------ backward: dK, dV
// Workgroup = 64 keys of one (batch, head), a wave owns 16 of them; loop over 64-row query tiles.  The stored P (bf16) and dS (f32,
// rounded to bf16 here) tiles are staged row-major [query][key] in LDS with coalesced loads and enter the MFMAs through transposed
// reads: dV += P^T . dO, dK += dS^T . Q with A[row = key][k = query] = tr_frag(P or dS tile), B[k = query][n = d] = tr_frag(dO or Q tile).
#define SH_PLD 72                   // row pitch (bf16) of the staged [64][64] P / dS tiles

// tile[il][jl] = src[(row0 + il) * pitch + col0 + jl] (0 outside [0, T) x [0, T)), 64 x 64, T = float or bf16 -> bf16 in LDS
template <typename TS>
__device__ __forceinline__ void sh_stage_tile(bf16* dst, const TS* __restrict__ src, size_t pitch, int row0, int col0, int T_) {
  for (int c = threadIdx.x; c < 64 * 16; c += blockDim.x) {
    const int il = c >> 4, jl = (c & 15) * 4, i = row0 + il, j = col0 + jl;
    union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
    for (int r = 0; r < 4; ++r) pk.e[r] = (i < T_ && j + r < T_) ? (bf16)(float)src[(size_t)i * pitch + j + r] : (bf16)0.f;
    *reinterpret_cast<uint2*>(&dst[il * SH_PLD + jl]) = pk.u;
  }
}

template <int DH>
__global__ void __launch_bounds__(256)
attn_bwd_kv_shaw_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ probs, int ldp, const bf16* __restrict__ dout,
                        const float* __restrict__ dscores, bf16* __restrict__ dqkv, ShGeom g) {
  constexpr int LDK = DH + 8, NT = DH / 16;
  __shared__ __attribute__((aligned(16))) bf16 Qs[SH_TILE * LDK];
  __shared__ __attribute__((aligned(16))) bf16 Ds[SH_TILE * LDK];
  __shared__ __attribute__((aligned(16))) bf16 Ps[SH_TILE * SH_PLD];
  __shared__ __attribute__((aligned(16))) bf16 Ss[SH_TILE * SH_PLD];
  const int T_ = g.T, H = g.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh % H, J0 = blockIdx.y * SH_TILE;
  const bf16* base = qkv + (size_t)b * T_ * g.ld + h * DH;
  const bf16* pb = probs + (size_t)bh * T_ * ldp;
  const float* sb = dscores + (size_t)bh * T_ * T_;
  f32x4 dk[NT], dv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { dk[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  for (int it0 = 0; it0 < T_; it0 += SH_TILE) {
    __syncthreads();
    sh_stage_rows<DH>(Qs, base, g.ld, it0, T_);
    sh_stage_rows<DH>(Ds, dout + (size_t)b * T_ * g.inner + h * DH, g.inner, it0, T_);
    sh_stage_tile<bf16>(Ps, pb, ldp, it0, J0, T_);
    sh_stage_tile<float>(Ss, sb, T_, it0, J0, T_);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bf16x8 pa = tr_frag(Ps, SH_PLD, 32 * c, 16 * wave, fq, fr), sa = tr_frag(Ss, SH_PLD, 32 * c, 16 * wave, fq, fr);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        dv[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, tr_frag(Ds, LDK, 32 * c, nt * 16, fq, fr), dv[nt], 0, 0, 0);
        dk[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, tr_frag(Qs, LDK, 32 * c, nt * 16, fq, fr), dk[nt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int jo = J0 + 16 * wave + 4 * fq + r;
    if (jo < T_) {
      bf16* row = dqkv + ((size_t)b * T_ + jo) * g.ld + h * DH;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        row[g.inner + nt * 16 + fr] = (bf16)(dk[nt][r] * g.scale);
        row[2 * g.inner + nt * 16 + fr] = (bf16)dv[nt][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------ backward: relative-embedding gradient
// dE[clamp(o) + P][d] += scale * sum_i dS[i][i - o] q_i[d].  Workgroup = 64 consecutive offsets o0 .. o0 + 63 of one (batch, head), a
// wave owns 16 of them; loop over the 64-row query tiles whose diagonals intersect the score matrix.  The skewed tile
// Sk[i][o] = dS[i][i - o] is built in LDS from contiguous pieces of dS rows (64 consecutive keys per query, reversed), then
// dE_tile += Sk^T . Q by MFMA (contraction over the 64 queries); one atomic flush per workgroup at the end.
template <int DH>
__global__ void __launch_bounds__(256)
attn_bwd_rel_shaw_kernel(const bf16* __restrict__ qkv, const float* __restrict__ dscores, float* __restrict__ demb, ShGeom g) {
  constexpr int LDK = DH + 8, NT = DH / 16;
  __shared__ __attribute__((aligned(16))) bf16 Qs[SH_TILE * LDK];
  __shared__ __attribute__((aligned(16))) bf16 Sk[SH_TILE * SH_PLD];
  const int T_ = g.T, H = g.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int o0 = (int)blockIdx.y * SH_TILE - (T_ - 1);                 // offsets o0 .. o0 + 63 (i - j ranges over -(T-1) .. T-1)
  const bf16* base = qkv + (size_t)b * T_ * g.ld + h * DH;
  const float* sb = dscores + (size_t)bh * T_ * T_;
  f32x4 de[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) de[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // queries with a valid key for some offset of the window: 0 <= i - o < T for o in [o0, o0 + 63]  ->  i in [max(0, o0), min(T, T + o0 + 63))
  const int ilo = max(0, o0) / SH_TILE * SH_TILE, ihi = min(T_, T_ + o0 + 63);
  for (int it0 = ilo; it0 < ihi; it0 += SH_TILE) {
    __syncthreads();
    sh_stage_rows<DH>(Qs, base, g.ld, it0, T_);
    for (int c = threadIdx.x; c < 64 * 16; c += blockDim.x) {
      const int il = c >> 4, ol = (c & 15) * 4, i = it0 + il;
      union { bf16 e[4]; uint2 u; } pk;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = i - (o0 + ol + r);
        pk.e[r] = (i < T_ && j >= 0 && j < T_) ? (bf16)sb[(size_t)i * T_ + j] : (bf16)0.f;
      }
      *reinterpret_cast<uint2*>(&Sk[il * SH_PLD + ol]) = pk.u;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bf16x8 sa = tr_frag(Sk, SH_PLD, 32 * c, 16 * wave, fq, fr);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        de[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa, tr_frag(Qs, LDK, 32 * c, nt * 16, fq, fr), de[nt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = o0 + 16 * wave + 4 * fq + r;
    if (o <= T_ - 1) {
      const int row = max(-g.max_pos, min(g.max_pos, o)) + g.max_pos;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) atomicAdd(&demb[(size_t)row * DH + nt * 16 + fr], de[nt][r] * g.scale);
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

// writes the dQ third of dqkv and dscores [B][H][T][T] (f32); the key-side kernel of attn_long.hip follows (lidk_attn_bwd)
int att_shaw_bwd_rows(const void* qkv, const void* embT, const void* probs, int ldp, const void* dout, void* dqkv, float* dscores,
                      int B, int T_, int H, int dh, int max_pos, hipStream_t s) {
  ShGeom g{B, T_, H, max_pos, H * dh, 3 * H * dh, 1.0f / sqrtf((float)dh)};
  const dim3 grid(B * H, cdiv(T_, SH_TILE));
  if (dh == 64) attn_bwd_rows_shaw_kernel<64><<<grid, 256, 0, s>>>((const bf16*)qkv, (const bf16*)embT, (const bf16*)probs, ldp,
                                                                   (const bf16*)dout, (bf16*)dqkv, dscores, g);
  else if (dh == 32) attn_bwd_rows_shaw_kernel<32><<<grid, 256, 0, s>>>((const bf16*)qkv, (const bf16*)embT, (const bf16*)probs, ldp,
                                                                        (const bf16*)dout, (bf16*)dqkv, dscores, g);
  else return LIDK_ERR_UNSUPPORTED;
  return launch_status();
}

// key side on the MFMA path: dK, dV thirds of dqkv from the stored P and dS, then the relative-embedding gradient (accumulated)
int att_shaw_bwd_cols(const void* qkv, const void* probs, int ldp, const void* dout, const float* dscores, void* dqkv, float* demb,
                      int B, int T_, int H, int dh, int max_pos, hipStream_t s) {
  ShGeom g{B, T_, H, max_pos, H * dh, 3 * H * dh, 1.0f / sqrtf((float)dh)};
  const dim3 grid(B * H, cdiv(T_, SH_TILE)), grid_e(B * H, cdiv(2 * T_ - 1, SH_TILE));
  if (dh == 64) {
    attn_bwd_kv_shaw_kernel<64><<<grid, 256, 0, s>>>((const bf16*)qkv, (const bf16*)probs, ldp, (const bf16*)dout, dscores, (bf16*)dqkv, g);
    if (demb) attn_bwd_rel_shaw_kernel<64><<<grid_e, 256, 0, s>>>((const bf16*)qkv, dscores, demb, g);
  } else if (dh == 32) {
    attn_bwd_kv_shaw_kernel<32><<<grid, 256, 0, s>>>((const bf16*)qkv, (const bf16*)probs, ldp, (const bf16*)dout, dscores, (bf16*)dqkv, g);
    if (demb) attn_bwd_rel_shaw_kernel<32><<<grid_e, 256, 0, s>>>((const bf16*)qkv, dscores, demb, g);
  } else {
    return LIDK_ERR_UNSUPPORTED;
  }
  return launch_status();
}
