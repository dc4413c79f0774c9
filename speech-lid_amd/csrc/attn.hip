// Attention core with Shaw relative-position scores (lid/conformer.py:117-148), v1: K/V/E tiles of one (batch, head)
// live in LDS, one wave per query row, scores lane-parallel over keys, softmax by wave shuffles, P.V lane-parallel
// over the head dimension.  The (T,T,dh) gather of the reference (E = emb[dist]) is never materialised: the
// relative term is q.(E[i-j]) read from an LDS slice of the embedding table.
//   scores[i][j] = scale * sum_d q[i][d] * (k[j][d] + emb[clamp(i-j,-P,P)+P][d])
#include "common.h"
#include <stdlib.h>

#define ATT_ROWS 16   // query rows per workgroup (4 waves x 4 rows)

template <typename T> struct AttPad { static constexpr int v = 1; };
template <> struct AttPad<bf16> { static constexpr int v = 2; };   // keep 4-byte row alignment, odd dword stride

// dot of q (f32, LDS broadcast) with (krow + erow), both T rows in LDS
template <typename T>
__device__ __forceinline__ float dot_qke(const float* q, const T* krow, const T* erow, int dh) {
  float s = 0.f;
  for (int d = 0; d < dh; ++d) s = fmaf(q[d], to_f(krow[d]) + to_f(erow[d]), s);
  return s;
}
template <typename T>
__device__ __forceinline__ float dot_qv(const float* q, const T* vrow, int dh) {
  float s = 0.f;
  for (int d = 0; d < dh; ++d) s = fmaf(q[d], to_f(vrow[d]), s);
  return s;
}

struct AttGeom { int B, T, H, dh, max_pos, inner, ld; };   // ld = 3*inner (row stride of qkv)

template <typename T>
__device__ __forceinline__ void att_stage_kve(const T* __restrict__ qkv, const float* __restrict__ emb, const AttGeom g,
                                              int b, int h, int i0, T* Ks, T* Vs, T* Es, int DHP) {
  const int T_ = g.T, dh = g.dh;
  for (int idx = threadIdx.x; idx < T_ * dh; idx += blockDim.x) {
    int j = idx / dh, d = idx - j * dh;
    const T* row = qkv + (size_t)(b * T_ + j) * g.ld + h * dh + d;
    Ks[j * DHP + d] = row[g.inner];
    Vs[j * DHP + d] = row[2 * g.inner];
  }
  const int NE = T_ + ATT_ROWS - 1;
  for (int idx = threadIdx.x; idx < NE * dh; idx += blockDim.x) {
    int e = idx / dh, d = idx - e * dh;
    int r = e + i0 - (T_ - 1);
    r = max(-g.max_pos, min(g.max_pos, r)) + g.max_pos;
    Es[e * DHP + d] = from_f<T>(emb[(size_t)r * dh + d]);
  }
}

// ------------------------------------------------------------------------------------ forward
template <typename T>
__global__ void __launch_bounds__(256)
attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ emb, T* __restrict__ out, T* __restrict__ probs,
                AttGeom g, int ldp, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T_ = g.T, dh = g.dh, DHP = dh + AttPad<T>::v, NE = T_ + ATT_ROWS - 1;
  T* Ks = reinterpret_cast<T*>(smem);
  T* Vs = Ks + T_ * DHP;
  T* Es = Vs + T_ * DHP;
  float* fbase = reinterpret_cast<float*>(smem + (((size_t)(2 * T_ + NE) * DHP * sizeof(T)) + 15) / 16 * 16);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* qs = fbase + wave * (dh + T_);
  float* ps = qs + dh;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * ATT_ROWS;
  att_stage_kve<T>(qkv, emb, g, b, h, i0, Ks, Vs, Es, DHP);
  __syncthreads();
  const int nparts = 64 / dh, dlane = lane % dh, part = lane / dh;
  for (int ii = wave; ii < ATT_ROWS; ii += 4) {
    const int i = i0 + ii;
    if (i >= T_) break;                                   // wave-uniform
    const T* qrow = qkv + (size_t)(b * T_ + i) * g.ld + h * dh;
    for (int d = lane; d < dh; d += 64) qs[d] = to_f(qrow[d]);
    __builtin_amdgcn_wave_barrier();
    float mx = -INFINITY;
    for (int j = lane; j < T_; j += 64) {
      float s = scale * dot_qke<T>(qs, Ks + j * DHP, Es + (i - j - i0 + T_ - 1) * DHP, dh);
      ps[j] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < T_; j += 64) { float p = __expf(ps[j] - mx); ps[j] = p; sum += p; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    T* prow = probs + ((size_t)(b * g.H + h) * T_ + i) * ldp;
    for (int j = lane; j < T_; j += 64) { float p = ps[j] * inv; ps[j] = p; prow[j] = from_f<T>(p); }
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
    for (int j = part; j < T_; j += nparts) o = fmaf(ps[j], to_f(Vs[j * DHP + dlane]), o);
    for (int off = dh; off < 64; off <<= 1) o += __shfl_xor(o, off, 64);
    if (lane < dh) out[(size_t)(b * T_ + i) * g.inner + h * dh + lane] = from_f<T>(o);
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------ backward, pass 1 (row-wise)
// dP = dO.V^T ; delta = sum_j P*dP ; dS = P*(dP - delta) -> dscores (f32, gradient w.r.t. the scaled scores)
// dq[i] = scale * sum_j dS[i][j] * (k[j] + E[i-j])
template <typename T>
__global__ void __launch_bounds__(256)
attn_bwd_rows_kernel(const T* __restrict__ qkv, const float* __restrict__ emb, const T* __restrict__ probs,
                     const T* __restrict__ dout, T* __restrict__ dqkv, float* __restrict__ dscores, AttGeom g,
                     int ldp, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T_ = g.T, dh = g.dh, DHP = dh + AttPad<T>::v, NE = T_ + ATT_ROWS - 1;
  T* Ks = reinterpret_cast<T*>(smem);
  T* Vs = Ks + T_ * DHP;
  T* Es = Vs + T_ * DHP;
  float* fbase = reinterpret_cast<float*>(smem + (((size_t)(2 * T_ + NE) * DHP * sizeof(T)) + 15) / 16 * 16);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* dos = fbase + wave * (dh + T_);
  float* ps = dos + dh;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * ATT_ROWS;
  att_stage_kve<T>(qkv, emb, g, b, h, i0, Ks, Vs, Es, DHP);
  __syncthreads();
  const int nparts = 64 / dh, dlane = lane % dh, part = lane / dh;
  for (int ii = wave; ii < ATT_ROWS; ii += 4) {
    const int i = i0 + ii;
    if (i >= T_) break;
    const T* dorow = dout + (size_t)(b * T_ + i) * g.inner + h * dh;
    for (int d = lane; d < dh; d += 64) dos[d] = to_f(dorow[d]);
    __builtin_amdgcn_wave_barrier();
    const size_t prow = ((size_t)(b * g.H + h) * T_ + i) * T_;
    const size_t pprow = ((size_t)(b * g.H + h) * T_ + i) * ldp;
    float delta = 0.f;
    for (int j = lane; j < T_; j += 64) {
      float dp = dot_qv<T>(dos, Vs + j * DHP, dh);
      float p = to_f(probs[pprow + j]);
      ps[j] = dp;
      delta = fmaf(p, dp, delta);
    }
    delta = wave_sum(delta);
    for (int j = lane; j < T_; j += 64) {
      float p = to_f(probs[pprow + j]);
      float ds = p * (ps[j] - delta);
      ps[j] = ds;
      dscores[prow + j] = ds;
    }
    __builtin_amdgcn_wave_barrier();
    float a = 0.f;
    for (int j = part; j < T_; j += nparts)
      a = fmaf(ps[j], to_f(Ks[j * DHP + dlane]) + to_f(Es[(i - j - i0 + T_ - 1) * DHP + dlane]), a);
    for (int off = dh; off < 64; off <<= 1) a += __shfl_xor(a, off, 64);
    if (lane < dh) dqkv[(size_t)(b * T_ + i) * g.ld + h * dh + lane] = from_f<T>(a * scale);
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------ backward, pass 2 (per batch*head)
// lane <-> key j:  dk[j][:] = scale * sum_i dS[i][j] q[i][:] ;  dv[j][:] = sum_i P[i][j] dO[i][:]
// lane <-> offset r: dE[clamp(r)+P][:] += scale * sum_i dS[i][i-r] q[i][:]   (one atomic per (b,h,r,d))
template <typename T, int DH>
__global__ void __launch_bounds__(256)
attn_bwd_cols_kernel(const T* __restrict__ qkv, const T* __restrict__ probs, const T* __restrict__ dout,
                     const float* __restrict__ dscores, T* __restrict__ dqkv, float* __restrict__ demb, AttGeom g,
                     int ldp, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T_ = g.T;
  float* Qs = reinterpret_cast<float*>(smem);          // [T][DH] f32 (broadcast reads, no padding needed)
  float* Ds = Qs + (size_t)T_ * DH;                     // dO
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / g.H, h = blockIdx.x % g.H;
  for (int idx = threadIdx.x; idx < T_ * DH; idx += blockDim.x) {
    int i = idx / DH, d = idx - i * DH;
    Qs[idx] = to_f(qkv[(size_t)(b * T_ + i) * g.ld + h * DH + d]);
    Ds[idx] = to_f(dout[(size_t)(b * T_ + i) * g.inner + h * DH + d]);
  }
  __syncthreads();
  const size_t base = (size_t)(b * g.H + h) * T_ * T_;
  const size_t pbase = (size_t)(b * g.H + h) * T_ * ldp;
  const int nkc = (T_ + 63) / 64, nrc = (2 * T_ - 1 + 63) / 64;
  for (int item = wave; item < nkc + nrc; item += 4) {
    if (item < nkc) {
      const int j = item * 64 + lane;
      float ak[DH], av[DH];
#pragma unroll
      for (int d = 0; d < DH; ++d) { ak[d] = 0.f; av[d] = 0.f; }
      if (j < T_) {
        for (int i = 0; i < T_; ++i) {
          const float s = dscores[base + (size_t)i * T_ + j];
          const float p = to_f(probs[pbase + (size_t)i * ldp + j]);
#pragma unroll
          for (int d = 0; d < DH; ++d) { ak[d] = fmaf(s, Qs[i * DH + d], ak[d]); av[d] = fmaf(p, Ds[i * DH + d], av[d]); }
        }
        T* krow = dqkv + (size_t)(b * T_ + j) * g.ld + g.inner + h * DH;
        T* vrow = krow + g.inner;
#pragma unroll
        for (int d = 0; d < DH; ++d) { krow[d] = from_f<T>(ak[d] * scale); vrow[d] = from_f<T>(av[d]); }
      }
    } else {
      const int r = (item - nkc) * 64 + lane - (T_ - 1);        // i - j
      if (r <= T_ - 1) {
        float ae[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) ae[d] = 0.f;
        const int ilo = max(0, r), ihi = min(T_ - 1, T_ - 1 + r);
        for (int i = ilo; i <= ihi; ++i) {
          const float s = dscores[base + (size_t)i * T_ + (i - r)];
#pragma unroll
          for (int d = 0; d < DH; ++d) ae[d] = fmaf(s, Qs[i * DH + d], ae[d]);
        }
        const int row = max(-g.max_pos, min(g.max_pos, r)) + g.max_pos;
#pragma unroll
        for (int d = 0; d < DH; ++d) atomicAdd(&demb[(size_t)row * DH + d], ae[d] * scale);
      }
    }
  }
}

// =====================================================================================================================
// MFMA forward (bf16, dh in {32, 64}, T <= 256): one workgroup per (batch, head).
//   LDS: Ks [Tp][DH+8], Vt [DH][Tp+8] (V transposed), Es [2*Tp+8][DH+8] (relative embeddings, index e = r + Tp), and one
//   P tile [16][Tp+8] per wave.  Each wave walks 16-row query blocks:
//     S tile (16x16)   = Q.K^T              (DH/32 MFMAs, v_mfma_f32_16x16x32_bf16)
//     R tiles (16x32)  = Q.E[r0..r0+31]^T   with r0 = i0 - j0 - 15: exactly the offsets i-j this tile needs
//     skew             : S[a][c] += R[a][a-c+15]; the source sits in the same register of a lane of the same 16-lane
//                        group, so it is one ds_bpermute per R tile and register
//     softmax          : row = one register across the 16 lanes of a group -> 4 xor-shuffles
//     P -> LDS (bf16) -> 16-byte stores to probs[b][h][i][0..ldp) and A operand of O = P.V (B operand from Vt)
// =====================================================================================================================
#define AF_NJ_MAX 16

template <int DH>
__global__ void __launch_bounds__(640)
attn_fwd_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ embT, bf16* __restrict__ out,
                     bf16* __restrict__ probs, AttGeom g, int ldp, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KS = DH / 32, LDK = DH + 8;
  const int T_ = g.T, Tp = (T_ + 31) / 32 * 32, NJ = Tp / 16, LDV = Tp + 8, NE = 2 * Tp + 8;
  bf16* Ks = reinterpret_cast<bf16*>(smem);
  bf16* Vs = Ks + Tp * LDK;                             // row-major; P.V reads it with transposed LDS reads (tr_frag)
  bf16* Es = Vs + Tp * LDK;
  bf16* Ps = Es + NE * LDK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, NW = nthr >> 6;
  const int b = blockIdx.x / g.H, h = blockIdx.x % g.H;
  const bf16* base = qkv + (size_t)b * T_ * g.ld + h * DH;

  // ---- stage K (row-major), V (transposed), E slice; 16-byte global loads
  constexpr int CH = DH / 8;
  for (int c = tid; c < Tp * CH; c += nthr) {
    int j = c / CH, dc = (c % CH) * 8;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
    if (j < T_) {
      kv = *reinterpret_cast<const uint4*>(base + (size_t)j * g.ld + g.inner + dc);
      vv = *reinterpret_cast<const uint4*>(base + (size_t)j * g.ld + 2 * g.inner + dc);
    }
    *reinterpret_cast<uint4*>(&Ks[j * LDK + dc]) = kv;
    *reinterpret_cast<uint4*>(&Vs[j * LDK + dc]) = vv;
  }
  for (int c = tid; c < NE * CH; c += nthr) {
    int e = c / CH, dc = (c % CH) * 8;
    int r = max(-g.max_pos, min(g.max_pos, e - Tp)) + g.max_pos;
    *reinterpret_cast<uint4*>(&Es[e * LDK + dc]) = *reinterpret_cast<const uint4*>(embT + (size_t)r * DH + dc);
  }
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  bf16* Pw = Ps + wave * 16 * LDV;
  const int nrb = (T_ + 15) / 16;
  for (int rb = wave; rb < nrb; rb += NW) {          // wave-uniform
    const int i0 = rb * 16;
    // Q fragments straight from global: lane holds row i0+fr, k = 32*ks + 8*fq .. +8
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i0 + fr < T_) v = *reinterpret_cast<const uint4*>(base + (size_t)(i0 + fr) * g.ld + ks * 32 + fq * 8);
      qf[ks] = *reinterpret_cast<bf16x8*>(&v);
    }
    float s[AF_NJ_MAX][4];
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int jt = 0; jt < AF_NJ_MAX; ++jt) {
      if (jt < NJ) {
        const int j0 = jt * 16;
        f32x4 as = {0.f, 0.f, 0.f, 0.f}, r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
        const int eb = i0 - j0 - 15 + Tp;                       // LDS row of offset r0 = i0 - j0 - 15
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(j0 + fr) * LDK + ks * 32 + fq * 8]);
          bf16x8 e0 = *reinterpret_cast<const bf16x8*>(&Es[(eb + fr) * LDK + ks * 32 + fq * 8]);
          bf16x8 e1 = *reinterpret_cast<const bf16x8*>(&Es[(eb + 16 + fr) * LDK + ks * 32 + fq * 8]);
          as = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf, as, 0, 0, 0);
          r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], e0, r0, 0, 0, 0);
          r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], e1, r1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = fq * 4 + r;                               // row inside the tile
          const int cp = a - fr + 15;                             // column of the R sub-block: 0..30
          const int src = (lane & 48) | (cp & 15);
          const float v0 = __shfl(r0[r], src, 64), v1 = __shfl(r1[r], src, 64);
          float v = scale * (as[r] + (cp < 16 ? v0 : v1));
          if (j0 + fr >= T_) v = -INFINITY;
          s[jt][r] = v;
          mx[r] = fmaxf(mx[r], v);
        }
      }
    }
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
    }
#pragma unroll
    for (int jt = 0; jt < AF_NJ_MAX; ++jt) {
      if (jt < NJ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { float p = __expf(s[jt][r] - mx[r]); s[jt][r] = p; sum[r] += p; }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) sum[r] += __shfl_xor(sum[r], o, 64);
      sum[r] = __builtin_amdgcn_rcpf(sum[r]);
    }
#pragma unroll
    for (int jt = 0; jt < AF_NJ_MAX; ++jt) {
      if (jt < NJ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Pw[(fq * 4 + r) * LDV + jt * 16 + fr] = (bf16)(s[jt][r] * sum[r]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // P rows -> global, 16 bytes per lane (probs == NULL: the backward recomputes them, nothing T x T is stored)
    if (probs) {
      bf16* pg = probs + ((size_t)(b * g.H + h) * T_ + i0) * ldp;
      for (int c = lane; c < 16 * (Tp / 8); c += 64) {
        int row = c / (Tp / 8), col = (c % (Tp / 8)) * 8;
        if (i0 + row < T_) *reinterpret_cast<uint4*>(pg + (size_t)row * ldp + col) = *reinterpret_cast<const uint4*>(&Pw[row * LDV + col]);
      }
    }
    // O = P.V
    f32x4 ao[DH / 16];
#pragma unroll
    for (int nt = 0; nt < DH / 16; ++nt) ao[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < Tp; k0 += 32) {
      bf16x8 pf = *reinterpret_cast<const bf16x8*>(&Pw[fr * LDV + k0 + fq * 8]);
#pragma unroll
      for (int nt = 0; nt < DH / 16; ++nt) {
        ao[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, tr_frag(Vs, LDK, k0, nt * 16, fq, fr), ao[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < DH / 16; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int i = i0 + fq * 4 + r;
        if (i < T_) out[(size_t)(b * T_ + i) * g.inner + h * DH + nt * 16 + fr] = (bf16)ao[nt][r];
      }
    __builtin_amdgcn_wave_barrier();
  }
}

// waves per workgroup: one 16-row query block per wave where LDS allows (each wave owns a [16][Tp+8] P tile), at least 4
#define ATT_NW_MAX 10
static size_t att_mfma_lds(int T_, int dh, int nw = 4) {
  int Tp = (T_ + 31) / 32 * 32;
  return (size_t)2 * ((size_t)2 * Tp * (dh + 8) + (size_t)(2 * Tp + 8) * (dh + 8) + (size_t)nw * 16 * (Tp + 8));
}
static int att_pick_nw(int T_, size_t (*lds)(int, int, int), int dh) {
  int nw = (T_ + 15) / 16;
  if (nw > ATT_NW_MAX) nw = ATT_NW_MAX;
  while (nw > 4 && lds(T_, dh, nw) > 160 * 1024) --nw;
  return nw < 4 ? 4 : nw;
}

extern "C" int lidk_attn_recompute_supported(int T_, int dh, int dtype);
extern "C" int lidk_attn_ldp(int T_, int dh, int dtype) {
  // row stride of the probs buffer: padded to a multiple of 32 for the MFMA kernels, plain T otherwise
  bool mfma = dtype == LIDK_BF16 && (dh == 32 || dh == 64) && T_ <= 16 * AF_NJ_MAX && att_mfma_lds(T_, dh) <= 160 * 1024;
  return mfma ? (T_ + 31) / 32 * 32 : T_;
}


// =====================================================================================================================
// MFMA backward (bf16).  Operands whose contraction index is the ROW index of a row-major LDS tile are read with
// ds_read_b64_tr_b16 (tr_frag): lane (fq, fr) gets X[k0 + 8*fq + jj][n0 + fr], jj = 0..7 — the 16x16x32 A/B fragment of
// X^T — from two transposed reads, so nothing is ever transposed in memory.
//   K1 (row blocks): dP = dO.V^T ; delta = rowsum(P*dP) ; dS = P*(dP - delta) -> dS (bf16, global + LDS)
//                    dq = scale * (dS.K + skew(dS).E)
//   K2 (per b,h)   : dv = P^T.dO ; dk = scale * dS^T.Q ; dE[r] += scale * sum_i dS[i][i-r] q[i]
// =====================================================================================================================
// NJM: compile-time bound on the key tiles NJ = Tp / 16 of the call (the score / dP tiles of a row block live in registers:
// with the bound at 16 for every T the arrays cost 128 VGPRs and the kernel spilled 188 bytes per lane to scratch)
template <int DH, int NJM>
__global__ void __launch_bounds__(640)
attn_bwd_rows_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ embT, const bf16* __restrict__ probs,
                          const bf16* __restrict__ dout, bf16* __restrict__ dqkv, bf16* __restrict__ dsT, AttGeom g,
                          int ldp, float scale, float* __restrict__ row_stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KS = DH / 32, LDK = DH + 8, CH = DH / 8;
  const int T_ = g.T, Tp = (T_ + 31) / 32 * 32, NJ = Tp / 16, LDV = Tp + 8, NE = 2 * Tp + 8;
  bf16* Ks = reinterpret_cast<bf16*>(smem);
  bf16* Vs = Ks + Tp * LDK;
  bf16* Es = Vs + Tp * LDK;
  bf16* Pall = Es + NE * LDK;                           // per wave: P rows, overwritten in place by dS
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, NW = nthr >> 6;
  const int b = blockIdx.x / g.H, h = blockIdx.x % g.H;
  const bf16* base = qkv + (size_t)b * T_ * g.ld + h * DH;
  for (int c = tid; c < Tp * CH; c += nthr) {
    int j = c / CH, dc = (c % CH) * 8;
    uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
    if (j < T_) {
      kv = *reinterpret_cast<const uint4*>(base + (size_t)j * g.ld + g.inner + dc);
      vv = *reinterpret_cast<const uint4*>(base + (size_t)j * g.ld + 2 * g.inner + dc);
    }
    *reinterpret_cast<uint4*>(&Ks[j * LDK + dc]) = kv;
    *reinterpret_cast<uint4*>(&Vs[j * LDK + dc]) = vv;
  }
  for (int c = tid; c < NE * CH; c += nthr) {
    int e = c / CH, dc = (c % CH) * 8;
    int r = max(-g.max_pos, min(g.max_pos, e - Tp)) + g.max_pos;
    *reinterpret_cast<uint4*>(&Es[e * LDK + dc]) = *reinterpret_cast<const uint4*>(embT + (size_t)r * DH + dc);
  }
  __syncthreads();

  const int fr = lane & 15, fq = lane >> 4;
  bf16* Pw = Pall + wave * 16 * LDV;
  bf16* Sw = Pw;                                        // every element is read (P) and rewritten (dS) by the same lane
  const int nrb = (T_ + 15) / 16;
  const size_t bh = (size_t)(b * g.H + h) * T_;
  for (int rb = wave; rb < nrb; rb += NW) {          // wave-uniform
    const int i0 = rb * 16;
    if (probs) {
      // P rows of this block -> LDS (16-byte chunks); rows >= T are zero
      for (int c = lane; c < 16 * (Tp / 8); c += 64) {
        int row = c / (Tp / 8), col = (c % (Tp / 8)) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i0 + row < T_) v = *reinterpret_cast<const uint4*>(probs + (bh + i0 + row) * ldp + col);
        *reinterpret_cast<uint4*>(&Pw[row * LDV + col]) = v;
      }
    } else {
      // recompute P = softmax(scale * (Q.K^T + skew(Q.E^T))) exactly as attn_fwd_mfma_kernel does (same tiles, same order), and
      // leave each row's log-sum-exp for the key-block kernel; nothing T x T is read from HBM
      bf16x8 qf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i0 + fr < T_) v = *reinterpret_cast<const uint4*>(base + (size_t)(i0 + fr) * g.ld + ks * 32 + fq * 8);
        qf[ks] = *reinterpret_cast<bf16x8*>(&v);
      }
      float sc[NJM][4];
      float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int jt = 0; jt < NJM; ++jt) {
        if (jt < NJ) {
          const int j0 = jt * 16;
          f32x4 as = {0.f, 0.f, 0.f, 0.f}, r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
          const int eb = i0 - j0 - 15 + Tp;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(j0 + fr) * LDK + ks * 32 + fq * 8]);
            bf16x8 e0 = *reinterpret_cast<const bf16x8*>(&Es[(eb + fr) * LDK + ks * 32 + fq * 8]);
            bf16x8 e1 = *reinterpret_cast<const bf16x8*>(&Es[(eb + 16 + fr) * LDK + ks * 32 + fq * 8]);
            as = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf, as, 0, 0, 0);
            r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], e0, r0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], e1, r1, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int a = fq * 4 + r, cp = a - fr + 15;
            const int src = (lane & 48) | (cp & 15);
            const float v0 = __shfl(r0[r], src, 64), v1 = __shfl(r1[r], src, 64);
            float v = scale * (as[r] + (cp < 16 ? v0 : v1));
            if (j0 + fr >= T_) v = -INFINITY;
            sc[jt][r] = v;
            mx[r] = fmaxf(mx[r], v);
          }
        }
      }
      float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
      }
#pragma unroll
      for (int jt = 0; jt < NJM; ++jt) {
        if (jt < NJ) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { float p = __expf(sc[jt][r] - mx[r]); sc[jt][r] = p; sum[r] += p; }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sum[r] += __shfl_xor(sum[r], o, 64);
        if (fr == 0 && i0 + fq * 4 + r < T_) row_stats[bh + i0 + fq * 4 + r] = mx[r] + __logf(sum[r]);
        sum[r] = __builtin_amdgcn_rcpf(sum[r]);
      }
#pragma unroll
      for (int jt = 0; jt < NJM; ++jt) {
        if (jt < NJ) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            Pw[(fq * 4 + r) * LDV + jt * 16 + fr] = (i0 + fq * 4 + r < T_) ? (bf16)(sc[jt][r] * sum[r]) : (bf16)0.f;
        }
      }
    }
    bf16x8 dof[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i0 + fr < T_) v = *reinterpret_cast<const uint4*>(dout + (size_t)(b * T_ + i0 + fr) * g.inner + h * DH + ks * 32 + fq * 8);
      dof[ks] = *reinterpret_cast<bf16x8*>(&v);
    }
    __builtin_amdgcn_wave_barrier();
    float dp[NJM][4];
    float delta[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < NJM; ++jt) {
      if (jt < NJ) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vs[(jt * 16 + fr) * LDK + ks * 32 + fq * 8]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[ks], vf, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dp[jt][r] = acc[r];
          delta[r] = fmaf((float)Pw[(fq * 4 + r) * LDV + jt * 16 + fr], acc[r], delta[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) delta[r] += __shfl_xor(delta[r], o, 64);
      if (!probs && fr == 0 && i0 + fq * 4 + r < T_) row_stats[(size_t)g.B * g.H * T_ + bh + i0 + fq * 4 + r] = delta[r];
    }
#pragma unroll
    for (int jt = 0; jt < NJM; ++jt) {
      if (jt < NJ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = (float)Pw[(fq * 4 + r) * LDV + jt * 16 + fr];
          Sw[(fq * 4 + r) * LDV + jt * 16 + fr] = (bf16)(p * (dp[jt][r] - delta[r]));
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // dS rows -> global (consumed by the column kernel)
    for (int c = lane; c < 16 * (Tp / 8); c += 64) {
      int row = c / (Tp / 8), col = (c % (Tp / 8)) * 8;
      if (i0 + row < T_) *reinterpret_cast<uint4*>(dsT + (bh + i0 + row) * ldp + col) = *reinterpret_cast<const uint4*>(&Sw[row * LDV + col]);
    }
    // dq = scale * (dS.K + skew(dS).E)
    f32x4 aq[DH / 16];
#pragma unroll
    for (int nt = 0; nt < DH / 16; ++nt) aq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < Tp; k0 += 32) {
      bf16x8 sf = *reinterpret_cast<const bf16x8*>(&Sw[fr * LDV + k0 + fq * 8]);
#pragma unroll
      for (int nt = 0; nt < DH / 16; ++nt)
        aq[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, tr_frag(Ks, LDK, k0, nt * 16, fq, fr), aq[nt], 0, 0, 0);
    }
    for (int jt = 0; jt < NJ; ++jt) {
      const int j0 = jt * 16;
      const int eb = i0 - j0 - 16 + Tp;                 // Es row of offset r0 = i0 - j0 - 16;  k index c' <-> r = r0 + c'
      union { bf16 e[8]; bf16x8 v; } sk;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int c = fr - (8 * fq + jj) + 16;           // column of the dS tile that has offset r0 + c' on row fr
        const bf16 val = Sw[fr * LDV + j0 + (c & 15)];
        sk.e[jj] = (c >= 0 && c <= 15) ? val : (bf16)0.f;
      }
#pragma unroll
      for (int nt = 0; nt < DH / 16; ++nt)
        aq[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sk.v, tr_frag(Es, LDK, eb, nt * 16, fq, fr), aq[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < DH / 16; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int i = i0 + fq * 4 + r;
        if (i < T_) dqkv[(size_t)(b * T_ + i) * g.ld + h * DH + nt * 16 + fr] = (bf16)(aq[nt][r] * scale);
      }
    __builtin_amdgcn_wave_barrier();
  }
}

// PART 0: dV, dK and dE; 1: dV and dK only (dE = the relative-position embedding's weight gradient is left to a later
// PART-2 launch, which the engine runs on the weight-gradient stream); 2: dE only (stages dS and Q only).
template <int DH, bool DUAL, int PART = 0>     // DUAL: LDS holds both phases' operands at once
__global__ void __launch_bounds__(1024)
attn_bwd_cols_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ probs, const bf16* __restrict__ dout,
                          const bf16* __restrict__ dsT, bf16* __restrict__ dqkv, float* __restrict__ demb, AttGeom g, int ldp,
                          float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LDK = DH + 8, CH = DH / 8;
  const int T_ = g.T, Tp = (T_ + 31) / 32 * 32, NJ = Tp / 16, LDV = Tp + 8;
  // both phases' operands are staged up front (P, dO | dS, Q in separate buffers) so their global loads overlap and the
  // second phase starts without another load round trip
  bf16* X = reinterpret_cast<bf16*>(smem);             // [Tp][LDV]: P
  bf16* Y = X + Tp * LDV;                               // [Tp][LDK]: dO
  bf16* X2 = DUAL ? Y + Tp * LDK : X;                   // [Tp][LDV]: dS
  bf16* Y2 = DUAL ? X2 + Tp * LDV : Y;                  // [Tp][LDK]: Q
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4, nthr = blockDim.x, NW = nthr >> 6;
  const int b = blockIdx.x / g.H, h = blockIdx.x % g.H;
  const size_t bh = (size_t)(b * g.H + h) * T_;
  auto stage = [&](bf16* X, bf16* Y, const bf16* sq, const bf16* rows, size_t row_stride) {
    for (int c = tid; c < Tp * (Tp / 8); c += nthr) {
      int i = c / (Tp / 8), col = (c % (Tp / 8)) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i < T_) v = *reinterpret_cast<const uint4*>(sq + (bh + i) * ldp + col);
      *reinterpret_cast<uint4*>(&X[i * LDV + col]) = v;
    }
    for (int c = tid; c < Tp * CH; c += nthr) {
      int i = c / CH, dc = (c % CH) * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (i < T_) v = *reinterpret_cast<const uint4*>(rows + (size_t)i * row_stride + dc);
      *reinterpret_cast<uint4*>(&Y[i * LDK + dc]) = v;
    }
  };
  // ---- phase A: dv[j][d] = sum_i P[i][j] dO[i][d]
  if (PART != 2) stage(X, Y, probs, dout + (size_t)b * T_ * g.inner + h * DH, g.inner);
  if (DUAL || PART == 2) stage(X2, Y2, dsT, qkv + (size_t)b * T_ * g.ld + h * DH, g.ld);
  __syncthreads();
  for (int jt = wave; jt < (PART == 2 ? 0 : NJ); jt += NW) {
    f32x4 acc[DH / 16];
#pragma unroll
    for (int nt = 0; nt < DH / 16; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < Tp; k0 += 32) {
      bf16x8 af = tr_frag(X, LDV, k0, jt * 16, fq, fr);
#pragma unroll
      for (int nt = 0; nt < DH / 16; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, tr_frag(Y, LDK, k0, nt * 16, fq, fr), acc[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < DH / 16; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int j = jt * 16 + fq * 4 + r;
        if (j < T_) dqkv[(size_t)(b * T_ + j) * g.ld + 2 * g.inner + h * DH + nt * 16 + fr] = (bf16)acc[nt][r];
      }
  }
  // ---- phase B: dk[j][d] = scale * sum_i dS[i][j] q[i][d] ; dE[r][d] += scale * sum_i dS[i][i-r] q[i][d]
  if (!DUAL && PART != 2) {
    __syncthreads();
    stage(X2, Y2, dsT, qkv + (size_t)b * T_ * g.ld + h * DH, g.ld);
    __syncthreads();
  }
  const int n_rt = 2 * Tp / 16;                          // offset tiles covering r in [-Tp, Tp)
  for (int item = (PART == 2 ? NJ : 0) + wave; item < (PART == 1 ? NJ : NJ + n_rt); item += NW) {
    f32x4 acc[DH / 16];
#pragma unroll
    for (int nt = 0; nt < DH / 16; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (item < NJ) {
      const int jt = item;
      for (int k0 = 0; k0 < Tp; k0 += 32) {
        bf16x8 af = tr_frag(X2, LDV, k0, jt * 16, fq, fr);
#pragma unroll
        for (int nt = 0; nt < DH / 16; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, tr_frag(Y2, LDK, k0, nt * 16, fq, fr), acc[nt], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < DH / 16; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int j = jt * 16 + fq * 4 + r;
          if (j < T_) dqkv[(size_t)(b * T_ + j) * g.ld + g.inner + h * DH + nt * 16 + fr] = (bf16)(acc[nt][r] * scale);
        }
    } else {
      const int rt = -Tp + 16 * (item - NJ);             // offsets rt .. rt+15
      const int klo = max(0, rt) / 32 * 32, khi = min(Tp, rt + 15 + Tp);
      for (int k0 = klo; k0 < khi; k0 += 32) {           // wave-uniform bounds: every lane runs the transposed reads
        union { bf16 e[8]; bf16x8 v; } ga;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int i = k0 + 8 * fq + jj, col = i - rt - fr;    // dS[i][i - r], r = rt + fr
          const bool ok = col >= 0 && col < Tp;
          const bf16 val = X2[i * LDV + (ok ? col : 0)];
          ga.e[jj] = ok ? val : (bf16)0.f;
        }
#pragma unroll
        for (int nt = 0; nt < DH / 16; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga.v, tr_frag(Y2, LDK, k0, nt * 16, fq, fr), acc[nt], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int roff = rt + fq * 4 + r;
        if (roff > -T_ && roff < T_) {
          const int row = max(-g.max_pos, min(g.max_pos, roff)) + g.max_pos;
#pragma unroll
          for (int nt = 0; nt < DH / 16; ++nt) atomicAdd(&demb[(size_t)row * DH + nt * 16 + fr], acc[nt][r] * scale);
        }
      }
    }
  }
}

// Key-block kernel of the recompute path (probs == NULL): a wave owns 16 keys and walks the query rows 32 at a time; the S tiles
// are recomputed from Q, K and the relative embeddings (same MFMAs and the same skew as the forward), P = exp(S - lse), dS =
// P (dP - delta) with lse / delta left by attn_bwd_rows_mfma_kernel.  The accumulator tiles of P and dS, packed to bf16, ARE the
// A operands of dV += P^T.dO and dK += dS^T.Q (slot-permuted MFMA, common.h tr_frag_split): no probabilities, no dS and nothing
// transposed is read from HBM or staged through LDS.  (dE, the embedding table's gradient, stays with the PART-2 launch of
// attn_bwd_cols_mfma_kernel on the weight-gradient stream, fed by the dS rows the row kernel writes.)
template <int DH, int MAXW>      // MAXW: waves per workgroup the call may use (T <= 160: 10 waves, 170 VGPRs each instead of 128)
__global__ void __launch_bounds__(64 * MAXW)
attn_bwd_kv_mfma_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ embT, const bf16* __restrict__ dout,
                        const float* __restrict__ row_stats, bf16* __restrict__ dqkv, AttGeom g, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KS = DH / 32, LDK = DH + 8, CH = DH / 8, NT = DH / 16;
  const int T_ = g.T, Tp = (T_ + 31) / 32 * 32, NJ = Tp / 16, NE = 2 * Tp + 8;
  bf16* Qs = reinterpret_cast<bf16*>(smem);             // [Tp][LDK]
  bf16* Ds = Qs + Tp * LDK;                             // dO
  bf16* Es = Ds + Tp * LDK;                             // [NE][LDK], row e <-> offset e - Tp
  float* lse_s = reinterpret_cast<float*>(Es + NE * LDK);
  float* del_s = lse_s + Tp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4, nthr = blockDim.x, NW = nthr >> 6;
  const int b = blockIdx.x / g.H, h = blockIdx.x % g.H;
  const size_t bh = (size_t)(b * g.H + h) * T_;
  const bf16* base = qkv + (size_t)b * T_ * g.ld + h * DH;
  const bf16* dbase = dout + (size_t)b * T_ * g.inner + h * DH;
  for (int c = tid; c < Tp * CH; c += nthr) {
    const int i = c / CH, dc = (c % CH) * 8;
    uint4 qv = make_uint4(0, 0, 0, 0), dv4 = make_uint4(0, 0, 0, 0);
    if (i < T_) {
      qv = *reinterpret_cast<const uint4*>(base + (size_t)i * g.ld + dc);
      dv4 = *reinterpret_cast<const uint4*>(dbase + (size_t)i * g.inner + dc);
    }
    *reinterpret_cast<uint4*>(&Qs[i * LDK + dc]) = qv;
    *reinterpret_cast<uint4*>(&Ds[i * LDK + dc]) = dv4;
  }
  for (int c = tid; c < NE * CH; c += nthr) {
    const int e = c / CH, dc = (c % CH) * 8;
    const int r = max(-g.max_pos, min(g.max_pos, e - Tp)) + g.max_pos;
    *reinterpret_cast<uint4*>(&Es[e * LDK + dc]) = *reinterpret_cast<const uint4*>(embT + (size_t)r * DH + dc);
  }
  for (int i = tid; i < Tp; i += nthr) {
    lse_s[i] = i < T_ ? row_stats[bh + i] : 0.f;
    del_s[i] = i < T_ ? row_stats[(size_t)g.B * g.H * T_ + bh + i] : 0.f;
  }
  __syncthreads();
  for (int jt = wave; jt < NJ; jt += NW) {                // wave-uniform
    const int j0 = jt * 16, j = j0 + fr;
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (j < T_) {
        kv = *reinterpret_cast<const uint4*>(base + (size_t)j * g.ld + g.inner + ks * 32 + fq * 8);
        vv = *reinterpret_cast<const uint4*>(base + (size_t)j * g.ld + 2 * g.inner + ks * 32 + fq * 8);
      }
      kf[ks] = *reinterpret_cast<bf16x8*>(&kv);
      vf[ks] = *reinterpret_cast<bf16x8*>(&vv);
    }
    f32x4 dk[NT], dv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { dk[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int i0 = 0; i0 < Tp; i0 += 32) {
      float pd[2][4], ds[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ib = i0 + 16 * t, eb = ib - j0 - 15 + Tp;
        f32x4 as = {0.f, 0.f, 0.f, 0.f}, r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 qa = *reinterpret_cast<const bf16x8*>(&Qs[(ib + fr) * LDK + ks * 32 + fq * 8]);
          const bf16x8 da = *reinterpret_cast<const bf16x8*>(&Ds[(ib + fr) * LDK + ks * 32 + fq * 8]);
          const bf16x8 e0 = *reinterpret_cast<const bf16x8*>(&Es[(eb + fr) * LDK + ks * 32 + fq * 8]);
          const bf16x8 e1 = *reinterpret_cast<const bf16x8*>(&Es[(eb + 16 + fr) * LDK + ks * 32 + fq * 8]);
          as = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks], as, 0, 0, 0);
          r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, e0, r0, 0, 0, 0);
          r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, e1, r1, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = fq * 4 + r, cp = a - fr + 15, i = ib + a;
          const int src = (lane & 48) | (cp & 15);
          const float v0 = __shfl(r0[r], src, 64), v1 = __shfl(r1[r], src, 64);
          const float sv = scale * (as[r] + (cp < 16 ? v0 : v1));
          const float p = (i < T_ && j < T_) ? __expf(sv - lse_s[i]) : 0.f;
          pd[t][r] = p;
          ds[t][r] = p * (dp[r] - del_s[i]);
        }
      }
      union { bf16 e[8]; bf16x8 v; } pf, sf;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pf.e[r] = (bf16)pd[0][r]; pf.e[4 + r] = (bf16)pd[1][r];
        sf.e[r] = (bf16)ds[0][r]; sf.e[4 + r] = (bf16)ds[1][r];
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        dv[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf.v, tr_frag_split(Ds, LDK, i0, nt * 16, fq, fr), dv[nt], 0, 0, 0);
        dk[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf.v, tr_frag_split(Qs, LDK, i0, nt * 16, fq, fr), dk[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jo = j0 + fq * 4 + r;
      if (jo < T_) {
        bf16* row = dqkv + (size_t)(b * T_ + jo) * g.ld + h * DH;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          row[g.inner + nt * 16 + fr] = (bf16)(dk[nt][r] * scale);
          row[2 * g.inner + nt * 16 + fr] = (bf16)dv[nt][r];
        }
      }
    }
  }
}
static size_t att_bwd_kv_mfma_lds(int T_, int dh) {
  const int Tp = (T_ + 31) / 32 * 32;
  return (size_t)2 * ((size_t)2 * Tp * (dh + 8) + (size_t)(2 * Tp + 8) * (dh + 8)) + (size_t)2 * Tp * 4;
}

static size_t att_bwd_rows_mfma_lds(int T_, int dh, int nw = 4) {
  int Tp = (T_ + 31) / 32 * 32;
  return (size_t)2 * ((size_t)2 * Tp * (dh + 8) + (size_t)(2 * Tp + 8) * (dh + 8) + (size_t)nw * 16 * (Tp + 8));
}
static size_t att_bwd_cols_mfma_lds(int T_, int dh, bool dual = false) {
  int Tp = (T_ + 31) / 32 * 32;
  return (size_t)2 * (dual ? 2 : 1) * ((size_t)Tp * (Tp + 8) + (size_t)Tp * (dh + 8));
}

template <int DH>
static void att_bwd_mfma_launch(const void* qkv, const void* embT, const void* probs, int ldp, const void* dout, void* dqkv,
                                float* demb, void* dsT, AttGeom g, hipStream_t s) {
  const float scale = 1.0f / sqrtf((float)DH);
  const int nw = att_pick_nw(g.T, att_bwd_rows_mfma_lds, DH);
  size_t l1 = att_bwd_rows_mfma_lds(g.T, DH, nw), l2 = att_bwd_cols_mfma_lds(g.T, DH);
  // row statistics (log-sum-exp, delta) of the recompute path live behind the bf16 dS rows in the caller's f32 scratch
  float* row_stats = reinterpret_cast<float*>(dsT) + (size_t)g.B * g.H * g.T * ldp / 2;
  const int Tp = (g.T + 31) / 32 * 32;
#define LIDK_ROWS_LAUNCH(NJM_)                                                                                                  \
  do {                                                                                                                         \
    (void)hipFuncSetAttribute((const void*)attn_bwd_rows_mfma_kernel<DH, NJM_>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                              (int)l1);                                                                                        \
    attn_bwd_rows_mfma_kernel<DH, NJM_><<<g.B * g.H, 64 * nw, l1, s>>>((const bf16*)qkv, (const bf16*)embT, (const bf16*)probs, \
                                                                     (const bf16*)dout, (bf16*)dqkv, (bf16*)dsT, g, ldp, scale, \
                                                                     row_stats);                                               \
  } while (0)
  if (Tp <= 64) LIDK_ROWS_LAUNCH(4);
  else if (Tp <= 128) LIDK_ROWS_LAUNCH(8);
  else if (Tp <= 160) LIDK_ROWS_LAUNCH(10);
  else if (Tp <= 192) LIDK_ROWS_LAUNCH(12);
  else LIDK_ROWS_LAUNCH(16);
#undef LIDK_ROWS_LAUNCH
  if (!probs) {                                      // recompute path: dK / dV from recomputed tiles, then (optionally) dE
    const size_t l3 = att_bwd_kv_mfma_lds(g.T, DH);
    const int nwk = min(16, max(4, Tp / 16));
    if (nwk <= 10) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_kv_mfma_kernel<DH, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
      attn_bwd_kv_mfma_kernel<DH, 10><<<g.B * g.H, 64 * nwk, l3, s>>>((const bf16*)qkv, (const bf16*)embT, (const bf16*)dout,
                                                                    row_stats, (bf16*)dqkv, g, scale);
    } else {
      (void)hipFuncSetAttribute((const void*)attn_bwd_kv_mfma_kernel<DH, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
      attn_bwd_kv_mfma_kernel<DH, 16><<<g.B * g.H, 64 * nwk, l3, s>>>((const bf16*)qkv, (const bf16*)embT, (const bf16*)dout,
                                                                    row_stats, (bf16*)dqkv, g, scale);
    }
    if (demb) {
      const void* np = nullptr; const void* nd = nullptr; void* nq = nullptr;
      const int nwc = min(16, max(4, 2 * Tp / 16 / 2));
      (void)hipFuncSetAttribute((const void*)attn_bwd_cols_mfma_kernel<DH, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
      attn_bwd_cols_mfma_kernel<DH, false, 2><<<g.B * g.H, 64 * nwc, l2, s>>>((const bf16*)qkv, (const bf16*)np, (const bf16*)nd,
                                                                          (const bf16*)dsT, (bf16*)nq, demb, g, ldp, scale);
    }
    return;
  }
  const bool dual = att_bwd_cols_mfma_lds(g.T, DH, true) <= 160 * 1024;
  if (dual) l2 = att_bwd_cols_mfma_lds(g.T, DH, true);
#define LIDK_COLS_LAUNCH(DUAL_, PART_, NW_)                                                                                    \
  do {                                                                                                                         \
    (void)hipFuncSetAttribute((const void*)attn_bwd_cols_mfma_kernel<DH, DUAL_, PART_>,                                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);                                            \
    attn_bwd_cols_mfma_kernel<DH, DUAL_, PART_><<<g.B * g.H, 64 * (NW_), l2, s>>>(                                             \
        (const bf16*)qkv, (const bf16*)probs, (const bf16*)dout, (const bf16*)dsT, (bf16*)dqkv, demb, g, ldp, scale);           \
  } while (0)
  if (demb) {                                        // NJ + n_rt items in phase B: two rounds of 16 waves
    const int nwc = min(16, max(4, (Tp / 16 + 2 * Tp / 16) / 2));
    if (dual) LIDK_COLS_LAUNCH(true, 0, nwc); else LIDK_COLS_LAUNCH(false, 0, nwc);
  } else {                                           // NJ items per phase: one wave each
    const int nwc = min(16, max(4, Tp / 16));
    if (dual) LIDK_COLS_LAUNCH(true, 1, nwc); else LIDK_COLS_LAUNCH(false, 1, nwc);
  }
}

// dE only, from the dS a previous lidk_attn_bwd (with drel_emb == NULL) left in `dsT`
template <int DH>
static void att_bwd_relpos_launch(const void* qkv, const void* dsT, int ldp, float* demb, AttGeom g, hipStream_t s) {
  const float scale = 1.0f / sqrtf((float)DH);
  const int Tp = (g.T + 31) / 32 * 32;
  size_t l2 = att_bwd_cols_mfma_lds(g.T, DH);
  const void *probs = nullptr, *dout = nullptr;
  void* dqkv = nullptr;
  const int nwc = min(16, max(4, 2 * Tp / 16 / 2));
  LIDK_COLS_LAUNCH(false, 2, nwc);
}
#undef LIDK_COLS_LAUNCH

// ------------------------------------------------------------------------------------ host side
// key-tiled MFMA kernels with Shaw relative positions (attn_shaw.hip)
int att_shaw_fwd(const void* qkv, const void* embT, void* out, void* probs, int ldp, int B, int T_, int H, int dh, int max_pos,
                 hipStream_t s);
int att_shaw_bwd_rows(const void* qkv, const void* embT, const void* probs, int ldp, const void* dout, void* dqkv, float* dscores,
                      int B, int T_, int H, int dh, int max_pos, hipStream_t s);
int att_long_cols_bf16(const void* qkv, const void* probs, int ldp, const void* dout, const float* dscores, void* dqkv, float* demb,
                       int B, int T_, int H, int dh, int max_pos, hipStream_t s);
int att_shaw_bwd_cols(const void* qkv, const void* probs, int ldp, const void* dout, const float* dscores, void* dqkv, float* demb,
                      int B, int T_, int H, int dh, int max_pos, hipStream_t s);
// key-tiled kernels for sequences whose K / V / E do not fit in LDS (attn_long.hip)
template <typename T>
int att_long_fwd(const void* qkv, const float* emb, void* out, void* probs, int ldp, int B, int T_, int H, int dh, int max_pos,
                 hipStream_t s);
template <typename T>
int att_long_bwd(const void* qkv, const float* emb, const void* probs, int ldp, const void* dout, void* dqkv, float* demb,
                 float* dscores, int B, int T_, int H, int dh, int max_pos, hipStream_t s);

template <typename T>
static size_t att_rows_lds(int T_, int dh) {
  size_t a = (size_t)(2 * T_ + T_ + ATT_ROWS - 1) * (dh + AttPad<T>::v) * sizeof(T);
  a = (a + 15) / 16 * 16;
  return a + 4 * (size_t)(dh + T_) * sizeof(float);
}
static bool att_dh_ok(int dh) { return dh == 8 || dh == 16 || dh == 32 || dh == 64; }

extern "C" int lidk_attn_fwd(const void* qkv, const float* rel_emb, const void* rel_emb_T, void* out, void* probs, int ldp,
                             int B, int T_, int heads, int dh, int max_pos, int dtype, void* stream) {
  if (!qkv || !rel_emb || !out || B <= 0 || T_ <= 0 || heads <= 0 || ldp < T_) return LIDK_ERR_ARG;
  if (!att_dh_ok(dh)) return LIDK_ERR_UNSUPPORTED;
  if (!probs && !lidk_attn_recompute_supported(T_, dh, dtype)) return LIDK_ERR_ARG;       // only the MFMA path can do without
  AttGeom g{B, T_, heads, dh, max_pos, heads * dh, 3 * heads * dh};
  const float scale = 1.0f / sqrtf((float)dh);
  hipStream_t s = as_stream(stream);
  if (dtype == LIDK_BF16 && rel_emb_T && ldp == (T_ + 31) / 32 * 32 && lidk_attn_ldp(T_, dh, dtype) == ldp && (dh == 32 || dh == 64)) {
    const int nw = att_pick_nw(T_, att_mfma_lds, dh);
    size_t lds = att_mfma_lds(T_, dh, nw);
    if (dh == 64) {
      (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attn_fwd_mfma_kernel<64><<<B * heads, 64 * nw, lds, s>>>((const bf16*)qkv, (const bf16*)rel_emb_T, (bf16*)out, (bf16*)probs, g, ldp, scale);
    } else {
      (void)hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attn_fwd_mfma_kernel<32><<<B * heads, 64 * nw, lds, s>>>((const bf16*)qkv, (const bf16*)rel_emb_T, (bf16*)out, (bf16*)probs, g, ldp, scale);
    }
    return launch_status();
  }
  // beyond the resident MFMA kernels (T > 256): the key-tiled MFMA kernel of attn_shaw.hip (LIDK_ATTN_SHAW=0: the VALU kernels)
  static const bool shaw = !(getenv("LIDK_ATTN_SHAW") && atoi(getenv("LIDK_ATTN_SHAW")) == 0);
  if (shaw && dtype == LIDK_BF16 && rel_emb_T && (dh == 32 || dh == 64))
    return att_shaw_fwd(qkv, rel_emb_T, out, probs, ldp, B, T_, heads, dh, max_pos, s);
  dim3 grid(cdiv(T_, ATT_ROWS), heads, B);
  if (dtype == LIDK_BF16) {
    size_t lds = att_rows_lds<bf16>(T_, dh);
    if (lds > 160 * 1024) return att_long_fwd<bf16>(qkv, rel_emb, out, probs, ldp, B, T_, heads, dh, max_pos, s);
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attn_fwd_kernel<bf16><<<grid, 256, lds, s>>>((const bf16*)qkv, rel_emb, (bf16*)out, (bf16*)probs, g, ldp, scale);
  } else if (dtype == LIDK_F32) {
    size_t lds = att_rows_lds<float>(T_, dh);
    if (lds > 160 * 1024) return att_long_fwd<float>(qkv, rel_emb, out, probs, ldp, B, T_, heads, dh, max_pos, s);
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attn_fwd_kernel<float><<<grid, 256, lds, s>>>((const float*)qkv, rel_emb, (float*)out, (float*)probs, g, ldp, scale);
  } else {
    return LIDK_ERR_ARG;
  }
  return launch_status();
}

template <typename T, int DH>
static void att_cols_launch(const void* qkv, const void* probs, const void* dout, const float* dscores, void* dqkv,
                            float* demb, AttGeom g, int ldp, float scale, hipStream_t s) {
  size_t lds = (size_t)2 * g.T * DH * sizeof(float);
  (void)hipFuncSetAttribute((const void*)attn_bwd_cols_kernel<T, DH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_bwd_cols_kernel<T, DH><<<g.B * g.H, 256, lds, s>>>((const T*)qkv, (const T*)probs, (const T*)dout, dscores,
                                                           (T*)dqkv, demb, g, ldp, scale);
}

template <typename T>
static int att_bwd_launch(const void* qkv, const float* rel_emb, const void* probs, int ldp, const void* dout, void* dqkv,
                          float* drel_emb, float* dscores, AttGeom g, hipStream_t s) {
  const float scale = 1.0f / sqrtf((float)g.dh);
  size_t lds = att_rows_lds<T>(g.T, g.dh);
  if (lds > 160 * 1024 || (size_t)2 * g.T * g.dh * sizeof(float) > 160 * 1024)
    return att_long_bwd<T>(qkv, rel_emb, probs, ldp, dout, dqkv, drel_emb, dscores, g.B, g.T, g.H, g.dh, g.max_pos, s);
  dim3 grid(cdiv(g.T, ATT_ROWS), g.H, g.B);
  (void)hipFuncSetAttribute((const void*)attn_bwd_rows_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_bwd_rows_kernel<T><<<grid, 256, lds, s>>>((const T*)qkv, rel_emb, (const T*)probs, (const T*)dout, (T*)dqkv,
                                                 dscores, g, ldp, scale);
  switch (g.dh) {
    case 8: att_cols_launch<T, 8>(qkv, probs, dout, dscores, dqkv, drel_emb, g, ldp, scale, s); break;
    case 16: att_cols_launch<T, 16>(qkv, probs, dout, dscores, dqkv, drel_emb, g, ldp, scale, s); break;
    case 32: att_cols_launch<T, 32>(qkv, probs, dout, dscores, dqkv, drel_emb, g, ldp, scale, s); break;
    case 64: att_cols_launch<T, 64>(qkv, probs, dout, dscores, dqkv, drel_emb, g, ldp, scale, s); break;
    default: return LIDK_ERR_UNSUPPORTED;
  }
  return launch_status();
}

// probs may be NULL in lidk_attn_fwd / lidk_attn_bwd (the backward recomputes the probabilities from Q, K, E) for these shapes
extern "C" int lidk_attn_recompute_supported(int T_, int dh, int dtype) {
  static const bool off = getenv("LIDK_ATTN_RECOMPUTE") && atoi(getenv("LIDK_ATTN_RECOMPUTE")) == 0;
  return !off && dtype == LIDK_BF16 && (dh == 32 || dh == 64) && T_ <= 16 * AF_NJ_MAX && att_bwd_rows_mfma_lds(T_, dh) <= 160 * 1024 &&
         att_bwd_cols_mfma_lds(T_, dh) <= 160 * 1024 && att_bwd_kv_mfma_lds(T_, dh) <= 160 * 1024 && att_mfma_lds(T_, dh) <= 160 * 1024;
}

extern "C" int lidk_attn_bwd_relpos_supported(int T_, int dh, int dtype) {
  return dtype == LIDK_BF16 && (dh == 32 || dh == 64) && T_ <= 16 * AF_NJ_MAX && att_bwd_rows_mfma_lds(T_, dh) <= 160 * 1024 &&
         att_bwd_cols_mfma_lds(T_, dh) <= 160 * 1024;
}

extern "C" int lidk_attn_bwd_relpos(const void* qkv, const float* dscores, int ldp, float* drel_emb, int B, int T_, int heads,
                                    int dh, int max_pos, int dtype, void* stream) {
  if (!qkv || !dscores || !drel_emb || B <= 0 || T_ <= 0 || heads <= 0 || ldp != (T_ + 31) / 32 * 32) return LIDK_ERR_ARG;
  if (!lidk_attn_bwd_relpos_supported(T_, dh, dtype)) return LIDK_ERR_UNSUPPORTED;
  AttGeom g{B, T_, heads, dh, max_pos, heads * dh, 3 * heads * dh};
  if (dh == 64) att_bwd_relpos_launch<64>(qkv, dscores, ldp, drel_emb, g, as_stream(stream));
  else att_bwd_relpos_launch<32>(qkv, dscores, ldp, drel_emb, g, as_stream(stream));
  return launch_status();
}

extern "C" int lidk_attn_bwd(const void* qkv, const float* rel_emb, const void* rel_emb_T, const void* probs, int ldp,
                             const void* dout, void* dqkv, float* drel_emb, float* dscores, int B, int T_, int heads, int dh,
                             int max_pos, int dtype, void* stream) {
  if (!qkv || !rel_emb || !dout || !dqkv || !dscores || B <= 0 || T_ <= 0 || heads <= 0 || ldp < T_)
    return LIDK_ERR_ARG;
  if (!att_dh_ok(dh)) return LIDK_ERR_UNSUPPORTED;
  if (!probs && !(lidk_attn_recompute_supported(T_, dh, dtype) && rel_emb_T && ldp == (T_ + 31) / 32 * 32)) return LIDK_ERR_ARG;
  if (!drel_emb && !lidk_attn_bwd_relpos_supported(T_, dh, dtype)) return LIDK_ERR_ARG;      // split form: MFMA path only
  AttGeom g{B, T_, heads, dh, max_pos, heads * dh, 3 * heads * dh};
  hipStream_t s = as_stream(stream);
  if (dtype == LIDK_BF16 && rel_emb_T && (dh == 32 || dh == 64) && ldp == (T_ + 31) / 32 * 32 && T_ <= 16 * AF_NJ_MAX &&
      att_bwd_rows_mfma_lds(T_, dh) <= 160 * 1024 && att_bwd_cols_mfma_lds(T_, dh) <= 160 * 1024) {
    if (dh == 64) att_bwd_mfma_launch<64>(qkv, rel_emb_T, probs, ldp, dout, dqkv, drel_emb, dscores, g, s);
    else att_bwd_mfma_launch<32>(qkv, rel_emb_T, probs, ldp, dout, dqkv, drel_emb, dscores, g, s);
    return launch_status();
  }
  // beyond the resident MFMA kernels: row side (dS, dQ) by the key-tiled MFMA kernel of attn_shaw.hip, key side by attn_long.hip
  static const bool shaw = !(getenv("LIDK_ATTN_SHAW") && atoi(getenv("LIDK_ATTN_SHAW")) == 0);
  if (shaw && dtype == LIDK_BF16 && rel_emb_T && probs && drel_emb && (dh == 32 || dh == 64)) {
    const int rc = att_shaw_bwd_rows(qkv, rel_emb_T, probs, ldp, dout, dqkv, dscores, B, T_, heads, dh, max_pos, s);
    if (rc != LIDK_OK) return rc;
    static const bool shaw_cols = !(getenv("LIDK_ATTN_SHAW_COLS") && atoi(getenv("LIDK_ATTN_SHAW_COLS")) == 0);
    if (shaw_cols) return att_shaw_bwd_cols(qkv, probs, ldp, dout, dscores, dqkv, drel_emb, B, T_, heads, dh, max_pos, s);
    return att_long_cols_bf16(qkv, probs, ldp, dout, dscores, dqkv, drel_emb, B, T_, heads, dh, max_pos, s);
  }
  if (dtype == LIDK_BF16) return att_bwd_launch<bf16>(qkv, rel_emb, probs, ldp, dout, dqkv, drel_emb, dscores, g, s);
  if (dtype == LIDK_F32) return att_bwd_launch<float>(qkv, rel_emb, probs, ldp, dout, dqkv, drel_emb, dscores, g, s);
  return LIDK_ERR_ARG;
}

// ------------------------------------------------------------------------------------ hardware self-test
// ds_read_b64_tr_b16 (gfx950): within each 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4x16 block of 16-bit elements; lane i receives column i of the 4 rows (row q in element q).  The TN GEMM and the attention
// backward rely on exactly this mapping; the self-test lets the test-suite pin it on the device.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void tr16_selftest_kernel(const short* __restrict__ in, short* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) short tile[8][64];
  for (int i = threadIdx.x; i < 8 * 64; i += 64) tile[i / 64][i % 64] = in[i];
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, idx = lane & 15;
  for (int half = 0; half < 2; ++half) {
    const short* addr = &tile[(idx >> 2) + 4 * half][16 * g + 4 * (idx & 3)];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
    for (int q = 0; q < 4; ++q) out[(lane * 2 + half) * 4 + q] = v[q];
  }
}
extern "C" int lidk_selftest_tr16(const void* in_8x64_i16, void* out_64x8_i16, void* stream) {
  if (!in_8x64_i16 || !out_64x8_i16) return LIDK_ERR_ARG;
  tr16_selftest_kernel<<<1, 64, 0, as_stream(stream)>>>((const short*)in_8x64_i16, (short*)out_64x8_i16);
  return launch_status();
}
