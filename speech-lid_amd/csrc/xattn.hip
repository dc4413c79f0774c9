// Key-tiled ("flash") multi-head self-attention for the transformer backbones (WavLM: lid/wavlm/modules.py:455-700 with the
// gated bucketed relative-position bias; wav2vec2: lid/s3prl_updream/wav2vec/wav2vec2.py:1009-1078 = fairseq MultiheadAttention
// with a key padding mask), bf16 operands, dh = 64, ANY sequence length T <= RB:
//
//   S[i][j] = scale * q_i.k_j + gate[b][h][i] * rb[h][j - i]      (bias optional)      keys j >= klen[b] are masked (optional)
//   A = softmax_j(S) ; Ad = dropout(A) (optional, counter-based or forced mask) ; O = Ad.V
//
// Nothing T x T ever reaches HBM: the forward keeps one f32 log-sum-exp per query row, the backward recomputes S tiles from
// Q, K and that number (two kernels, no communication between workgroups, no atomics on the main outputs):
//   lidk_xattn_fwd   workgroup = 64 query rows of one (batch, head), 4 waves x 16 rows, K/V staged 64 keys at a time
//   xattn_bwd_q      same decomposition: delta_i = dO_i.O_i, dQ, and the bias gradients (dgate rows; drb through LDS)
//   xattn_bwd_kv     workgroup = 64 keys, loops over 64-row query tiles: dK, dV in registers
//
// MFMA orientation.  Forward and bwd_q compute S^T tiles (A = K rows, B = Q rows): a lane then holds, for ITS query i = fr,
// keys 4*fq + r of two stacked 16-key tiles - exactly the (slot-permuted, common.h tr_frag_split) A operand of P.V and dS.K,
// so probabilities never pass through LDS.  bwd_kv computes S tiles (A = Q rows, B = K rows): the accumulators are the A
// operand of P^T.dO and dS^T.Q the same way.  Row statistics live per lane (i = fr) and are reduced over the 4 lane groups.
#include "common.h"

#define XA_DH 64
#define XA_LDK (XA_DH + 8)
#define XA_TILE 64

struct XaGeom { int B, T, H, RB, inner, ld; float scale, drop_p, inv_keep; unsigned long long seed; };

__device__ __forceinline__ bf16x8 xa_pack(const float* a, const float* b) {
  union { bf16 e[8]; bf16x8 v; } u;
#pragma unroll
  for (int r = 0; r < 4; ++r) { u.e[r] = (bf16)a[r]; u.e[4 + r] = (bf16)b[r]; }
  return u.v;
}

__device__ __forceinline__ void xa_stage_rows(bf16* dst, const bf16* src, size_t row_stride, int row0, int nrows_valid) {
  // 64 rows x 64 columns, 16 bytes per access; rows >= nrows_valid are zero
  constexpr int CH = XA_DH / 8;
  for (int c = threadIdx.x; c < XA_TILE * CH; c += blockDim.x) {
    const int row = c / CH, dc = (c % CH) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + row < nrows_valid) v = *reinterpret_cast<const uint4*>(src + (size_t)(row0 + row) * row_stride + dc);
    *reinterpret_cast<uint4*>(&dst[row * XA_LDK + dc]) = v;
  }
}

__device__ __forceinline__ bool xa_keep(const unsigned char* keep, const XaGeom& g, size_t e) {
  return keep ? keep[e] != 0 : uniform32_from(g.seed, (uint64_t)e) >= g.drop_p;
}

// ------------------------------------------------------------------------------------------------------------ forward
template <bool BIAS, bool DROP>
__global__ void __launch_bounds__(256)
xattn_fwd_kernel(const bf16* __restrict__ qkv, const float* __restrict__ gate, const float* __restrict__ rb,
                 const int* __restrict__ klen, bf16* __restrict__ out, float* __restrict__ lse,
                 const unsigned char* __restrict__ keep, XaGeom g) {
  __shared__ __attribute__((aligned(16))) bf16 Ks[XA_TILE * XA_LDK];
  __shared__ __attribute__((aligned(16))) bf16 Vs[XA_TILE * XA_LDK];
  extern __shared__ float rbs[];                                  // [T + 63]: rb[h][j - i] at (j - i) + I0 + 63
  const int T_ = g.T, H = g.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh % H, I0 = blockIdx.y * XA_TILE;
  const int kend = klen ? min(T_, max(klen[b], 1)) : T_;
  const bf16* base = qkv + (size_t)b * T_ * g.ld + h * XA_DH;
  if (BIAS) {
    for (int x = threadIdx.x; x < T_ + 63; x += blockDim.x) {
      const int r = x - I0 - 63;
      rbs[x] = (r > -g.RB && r < g.RB) ? rb[(size_t)h * (2 * g.RB - 1) + r + g.RB - 1] : 0.f;
    }
  }
  const int i = I0 + 16 * wave + fr, ic = min(i, T_ - 1);
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)ic * g.ld + ks * 32 + 8 * fq);
  const float gi = BIAS ? gate[(size_t)bh * T_ + ic] : 0.f;
  float m = -INFINITY, l = 0.f;
  f32x4 O[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) O[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int jt0 = 0; jt0 < kend; jt0 += XA_TILE) {
    __syncthreads();
    xa_stage_rows(Ks, base + g.inner, g.ld, jt0, T_);
    xa_stage_rows(Vs, base + 2 * g.inner, g.ld, jt0, T_);
    __syncthreads();
    float s[4][4];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(16 * t + fr) * XA_LDK + ks * 32 + 8 * fq]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = jt0 + 16 * t + 4 * fq + r;
        float v = acc[r] * g.scale;
        if (BIAS) v += gi * rbs[min(j, T_ - 1) - ic + I0 + 63];
        if (j >= kend) v = -INFINITY;
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float alpha = (m == -INFINITY) ? 0.f : __expf(m - m_new);
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __expf(s[t][r] - m_new);
        psum += p;
        if (DROP) {
          const int j = jt0 + 16 * t + 4 * fq + r;
          const size_t e = ((size_t)bh * T_ + ic) * T_ + min(j, T_ - 1);
          p = xa_keep(keep, g, e) ? p * g.inv_keep : 0.f;
        }
        s[t][r] = p;
      }
    l = l * alpha + psum;
    m = m_new;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ar = __shfl(alpha, (lane & 48) | (4 * fq + r), 64);       // alpha of output row 4*fq + r
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) O[nt][r] *= ar;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bf16x8 pf = xa_pack(s[2 * c], s[2 * c + 1]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        O[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, tr_frag_split(Vs, XA_LDK, 32 * c, nt * 16, fq, fr), O[nt], 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.f / l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ir = __shfl(inv, (lane & 48) | (4 * fq + r), 64);
    const int io = I0 + 16 * wave + 4 * fq + r;
    if (io < T_) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) out[((size_t)b * T_ + io) * g.inner + h * XA_DH + nt * 16 + fr] = (bf16)(O[nt][r] * ir);
    }
  }
  if (fq == 0 && i < T_) lse[(size_t)bh * T_ + i] = m + __logf(l);
}

// ------------------------------------------------------------------------------------------------------------ backward: dQ
template <bool BIAS, bool DROP>
__global__ void __launch_bounds__(256)
xattn_bwd_q_kernel(const bf16* __restrict__ qkv, const float* __restrict__ gate, const float* __restrict__ rb,
                   const int* __restrict__ klen, const bf16* __restrict__ out, const bf16* __restrict__ dout,
                   const float* __restrict__ lse, const unsigned char* __restrict__ keep, bf16* __restrict__ dqkv,
                   float* __restrict__ delta, float* __restrict__ dgate, float* __restrict__ drb, XaGeom g) {
  __shared__ __attribute__((aligned(16))) bf16 Ks[XA_TILE * XA_LDK];
  __shared__ __attribute__((aligned(16))) bf16 Vs[XA_TILE * XA_LDK];
  extern __shared__ float dyn[];
  float* rbs = dyn;                                               // [T + 63]
  float* drbs = dyn + (g.T + 63);                                 // [T + 63] (only with drb)
  const int T_ = g.T, H = g.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh % H, I0 = blockIdx.y * XA_TILE;
  const int kend = klen ? min(T_, max(klen[b], 1)) : T_;
  const bf16* base = qkv + (size_t)b * T_ * g.ld + h * XA_DH;
  if (BIAS) {
    for (int x = threadIdx.x; x < T_ + 63; x += blockDim.x) {
      const int r = x - I0 - 63;
      rbs[x] = (r > -g.RB && r < g.RB) ? rb[(size_t)h * (2 * g.RB - 1) + r + g.RB - 1] : 0.f;
      if (drb) drbs[x] = 0.f;
    }
  }
  const int i = I0 + 16 * wave + fr, ic = min(i, T_ - 1);
  bf16x8 qf[2], dof[2];
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)ic * g.ld + ks * 32 + 8 * fq);
    const size_t off = ((size_t)b * T_ + ic) * g.inner + h * XA_DH + ks * 32 + 8 * fq;
    dof[ks] = *reinterpret_cast<const bf16x8*>(dout + off);
    const bf16x8 of = *reinterpret_cast<const bf16x8*>(out + off);
#pragma unroll
    for (int e = 0; e < 8; ++e) dl = fmaf((float)dof[ks][e], (float)of[e], dl);
  }
  dl += __shfl_xor(dl, 16, 64);
  dl += __shfl_xor(dl, 32, 64);                                   // delta_i = dO_i . O_i
  if (fq == 0 && i < T_) delta[(size_t)bh * T_ + i] = dl;
  const float gi = BIAS ? gate[(size_t)bh * T_ + ic] : 0.f;
  const float li = lse[(size_t)bh * T_ + ic];
  float dg = 0.f;
  f32x4 dq[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) dq[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int jt0 = 0; jt0 < kend; jt0 += XA_TILE) {
    __syncthreads();
    xa_stage_rows(Ks, base + g.inner, g.ld, jt0, T_);
    xa_stage_rows(Vs, base + 2 * g.inner, g.ld, jt0, T_);
    __syncthreads();
    float ds[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[(16 * t + fr) * XA_LDK + ks * 32 + 8 * fq]);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&Vs[(16 * t + fr) * XA_LDK + ks * 32 + 8 * fq]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], acc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = jt0 + 16 * t + 4 * fq + r, jc = min(j, T_ - 1);
        float v = acc[r] * g.scale, rbv = 0.f;
        if (BIAS) { rbv = rbs[jc - ic + I0 + 63]; v += gi * rbv; }
        const float p = (j < kend) ? __expf(v - li) : 0.f;
        float da = dp[r];
        if (DROP) da = xa_keep(keep, g, ((size_t)bh * T_ + ic) * T_ + jc) ? da * g.inv_keep : 0.f;
        const float d = p * (da - dl);
        ds[t][r] = d;
        if (BIAS) {
          dg = fmaf(d, rbv, dg);
          if (drb && i < T_ && j < kend) atomicAdd(&drbs[jc - ic + I0 + 63], gi * d);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bf16x8 sf = xa_pack(ds[2 * c], ds[2 * c + 1]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        dq[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, tr_frag_split(Ks, XA_LDK, 32 * c, nt * 16, fq, fr), dq[nt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int io = I0 + 16 * wave + 4 * fq + r;
    if (io < T_) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) dqkv[((size_t)b * T_ + io) * g.ld + h * XA_DH + nt * 16 + fr] = (bf16)(dq[nt][r] * g.scale);
    }
  }
  if (BIAS) {
    dg += __shfl_xor(dg, 16, 64);
    dg += __shfl_xor(dg, 32, 64);
    if (fq == 0 && i < T_ && dgate) dgate[(size_t)bh * T_ + i] = dg;
    if (drb) {
      __syncthreads();
      for (int x = threadIdx.x; x < T_ + 63; x += blockDim.x) {
        const int r = x - I0 - 63;
        const float v = drbs[x];
        if (v != 0.f && r > -g.RB && r < g.RB) atomicAdd(&drb[(size_t)h * (2 * g.RB - 1) + r + g.RB - 1], v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ backward: dK, dV
template <bool BIAS, bool DROP>
__global__ void __launch_bounds__(256)
xattn_bwd_kv_kernel(const bf16* __restrict__ qkv, const float* __restrict__ gate, const float* __restrict__ rb,
                    const int* __restrict__ klen, const bf16* __restrict__ dout, const float* __restrict__ lse,
                    const float* __restrict__ delta, const unsigned char* __restrict__ keep, bf16* __restrict__ dqkv, XaGeom g) {
  __shared__ __attribute__((aligned(16))) bf16 Qs[XA_TILE * XA_LDK];
  __shared__ __attribute__((aligned(16))) bf16 Ds[XA_TILE * XA_LDK];
  __shared__ float lse_s[XA_TILE], del_s[XA_TILE], gate_s[XA_TILE];
  extern __shared__ float rbs[];                                  // [T + 63]: rb[h][j - i] at (j - i) - J0 + T - 1
  const int T_ = g.T, H = g.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
  const int bh = blockIdx.x, b = bh / H, h = bh % H, J0 = blockIdx.y * XA_TILE;
  const int kend = klen ? min(T_, max(klen[b], 1)) : T_;
  const bf16* base = qkv + (size_t)b * T_ * g.ld + h * XA_DH;
  const int j = J0 + 16 * wave + fr, jc = min(j, T_ - 1);
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) { dk[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  if (J0 < kend) {                                                // block-uniform: a fully padded key block has zero gradients
    if (BIAS) {
      for (int x = threadIdx.x; x < T_ + 63; x += blockDim.x) {
        const int r = x + J0 - T_ + 1;
        rbs[x] = (r > -g.RB && r < g.RB) ? rb[(size_t)h * (2 * g.RB - 1) + r + g.RB - 1] : 0.f;
      }
    }
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)jc * g.ld + g.inner + ks * 32 + 8 * fq);
      vf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)jc * g.ld + 2 * g.inner + ks * 32 + 8 * fq);
    }
    const bool jok = j < kend;
    for (int it0 = 0; it0 < T_; it0 += XA_TILE) {
      __syncthreads();
      xa_stage_rows(Qs, base, g.ld, it0, T_);
      xa_stage_rows(Ds, dout + (size_t)b * T_ * g.inner + h * XA_DH, g.inner, it0, T_);
      if (threadIdx.x < XA_TILE) {
        const int ii = min(it0 + (int)threadIdx.x, T_ - 1);
        lse_s[threadIdx.x] = lse[(size_t)bh * T_ + ii];
        del_s[threadIdx.x] = delta[(size_t)bh * T_ + ii];
        gate_s[threadIdx.x] = BIAS ? gate[(size_t)bh * T_ + ii] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float pd[2][4], ds[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row0 = 32 * c + 16 * t;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 qa = *reinterpret_cast<const bf16x8*>(&Qs[(row0 + fr) * XA_LDK + ks * 32 + 8 * fq]);
            const bf16x8 da = *reinterpret_cast<const bf16x8*>(&Ds[(row0 + fr) * XA_LDK + ks * 32 + 8 * fq]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[ks], acc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[ks], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int il = row0 + 4 * fq + r, i = it0 + il, ic = min(i, T_ - 1);
            float v = acc[r] * g.scale;
            if (BIAS) v += gate_s[il] * rbs[jc - ic - J0 + T_ - 1];
            float p = (jok && i < T_) ? __expf(v - lse_s[il]) : 0.f;
            float da = dp[r];
            float pdrop = p;
            if (DROP) {
              const bool kp = xa_keep(keep, g, ((size_t)bh * T_ + ic) * T_ + jc);
              da = kp ? da * g.inv_keep : 0.f;
              pdrop = kp ? p * g.inv_keep : 0.f;
            }
            pd[t][r] = pdrop;
            ds[t][r] = p * (da - del_s[il]);
          }
        }
        const bf16x8 pf = xa_pack(pd[0], pd[1]), sf = xa_pack(ds[0], ds[1]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          dv[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, tr_frag_split(Ds, XA_LDK, 32 * c, nt * 16, fq, fr), dv[nt], 0, 0, 0);
          dk[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, tr_frag_split(Qs, XA_LDK, 32 * c, nt * 16, fq, fr), dk[nt], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int jo = J0 + 16 * wave + 4 * fq + r;
    if (jo < T_) {
      bf16* row = dqkv + ((size_t)b * T_ + jo) * g.ld + h * XA_DH;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        row[g.inner + nt * 16 + fr] = (bf16)(dk[nt][r] * g.scale);
        row[2 * g.inner + nt * 16 + fr] = (bf16)dv[nt][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ host side
static bool xa_args_ok(const void* qkv, const float* gate, const float* rb, int B, int T_, int H, int dh, int RB, float drop_p) {
  if (!qkv || B <= 0 || T_ <= 0 || H <= 0 || drop_p < 0.f || drop_p >= 1.f) return false;
  if ((gate == nullptr) != (rb == nullptr)) return false;
  if (gate && RB < T_) return false;
  (void)dh;
  return true;
}

extern "C" int lidk_xattn_max_frames(int dh) { return dh == XA_DH ? 8192 : 0; }

extern "C" int lidk_xattn_fwd(const void* qkv, const float* gate, const float* rb, const int* klen, void* out, float* lse,
                              const unsigned char* keep, float drop_p, unsigned long long seed, int B, int T_, int H, int dh,
                              int RB, void* stream) {
  if (!xa_args_ok(qkv, gate, rb, B, T_, H, dh, RB, drop_p) || !out || !lse) return LIDK_ERR_ARG;
  if (dh != XA_DH || T_ > 8192) return LIDK_ERR_UNSUPPORTED;
  XaGeom g{B, T_, H, RB, H * dh, 3 * H * dh, 1.0f / sqrtf((float)dh), drop_p, 1.0f / (1.0f - drop_p), seed};
  const dim3 grid(B * H, cdiv(T_, XA_TILE));
  const size_t dyn = gate ? (size_t)(T_ + 63) * 4 : 0;
  const bool drop = drop_p > 0.f || keep;
  hipStream_t s = as_stream(stream);
#define XA_FWD(BIAS_, DROP_)                                                                                                   \
  xattn_fwd_kernel<BIAS_, DROP_><<<grid, 256, dyn, s>>>((const bf16*)qkv, gate, rb, klen, (bf16*)out, lse, keep, g)
  if (gate) { if (drop) XA_FWD(true, true); else XA_FWD(true, false); }
  else { if (drop) XA_FWD(false, true); else XA_FWD(false, false); }
#undef XA_FWD
  return launch_status();
}

// delta: scratch [B][H][T] f32 (written by the first kernel, read by the second).  dgate [B][H][T] is overwritten; drb
// [H][2*RB-1] is ACCUMULATED into (zero it per step); either may be NULL.
extern "C" int lidk_xattn_bwd(const void* qkv, const float* gate, const float* rb, const int* klen, const void* out,
                              const void* dout, const float* lse, const unsigned char* keep, float drop_p,
                              unsigned long long seed, void* dqkv, float* delta, float* dgate, float* drb, int B, int T_, int H,
                              int dh, int RB, void* stream) {
  if (!xa_args_ok(qkv, gate, rb, B, T_, H, dh, RB, drop_p) || !out || !dout || !lse || !dqkv || !delta) return LIDK_ERR_ARG;
  if (dh != XA_DH || T_ > 8192) return LIDK_ERR_UNSUPPORTED;
  XaGeom g{B, T_, H, RB, H * dh, 3 * H * dh, 1.0f / sqrtf((float)dh), drop_p, 1.0f / (1.0f - drop_p), seed};
  const dim3 grid(B * H, cdiv(T_, XA_TILE));
  const size_t dyn_q = gate ? (size_t)(T_ + 63) * 4 * (drb ? 2 : 1) : 0, dyn_kv = gate ? (size_t)(T_ + 63) * 4 : 0;
  const bool drop = drop_p > 0.f || keep;
  hipStream_t s = as_stream(stream);
#define XA_BWD(BIAS_, DROP_)                                                                                                   \
  do {                                                                                                                         \
    xattn_bwd_q_kernel<BIAS_, DROP_><<<grid, 256, dyn_q, s>>>((const bf16*)qkv, gate, rb, klen, (const bf16*)out,              \
                                                              (const bf16*)dout, lse, keep, (bf16*)dqkv, delta, dgate, drb, g); \
    xattn_bwd_kv_kernel<BIAS_, DROP_><<<grid, 256, dyn_kv, s>>>((const bf16*)qkv, gate, rb, klen, (const bf16*)dout, lse,      \
                                                                delta, keep, (bf16*)dqkv, g);                                  \
  } while (0)
  if (gate) { if (drop) XA_BWD(true, true); else XA_BWD(true, false); }
  else { if (drop) XA_BWD(false, true); else XA_BWD(false, false); }
#undef XA_BWD
  return launch_status();
}
