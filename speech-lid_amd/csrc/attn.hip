// Attention core with Shaw relative-position scores (lid/conformer.py:117-148), v1: K/V/E tiles of one (batch, head)
// live in LDS, one wave per query row, scores lane-parallel over keys, softmax by wave shuffles, P.V lane-parallel
// over the head dimension.  The (T,T,dh) gather of the reference (E = emb[dist]) is never materialised: the
// relative term is q.(E[i-j]) read from an LDS slice of the embedding table.
//   scores[i][j] = scale * sum_d q[i][d] * (k[j][d] + emb[clamp(i-j,-P,P)+P][d])
#include "common.h"

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
                AttGeom g, float scale) {
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
    T* prow = probs + ((size_t)(b * g.H + h) * T_ + i) * T_;
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
                     float scale) {
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
    float delta = 0.f;
    for (int j = lane; j < T_; j += 64) {
      float dp = dot_qv<T>(dos, Vs + j * DHP, dh);
      float p = to_f(probs[prow + j]);
      ps[j] = dp;
      delta = fmaf(p, dp, delta);
    }
    delta = wave_sum(delta);
    for (int j = lane; j < T_; j += 64) {
      float p = to_f(probs[prow + j]);
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
                     float scale) {
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
          const float p = to_f(probs[base + (size_t)i * T_ + j]);
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

// ------------------------------------------------------------------------------------ host side
template <typename T>
static size_t att_rows_lds(int T_, int dh) {
  size_t a = (size_t)(2 * T_ + T_ + ATT_ROWS - 1) * (dh + AttPad<T>::v) * sizeof(T);
  a = (a + 15) / 16 * 16;
  return a + 4 * (size_t)(dh + T_) * sizeof(float);
}
static bool att_dh_ok(int dh) { return dh == 8 || dh == 16 || dh == 32 || dh == 64; }

extern "C" int lidk_attn_fwd(const void* qkv, const float* rel_emb, void* out, void* probs, int B, int T_, int heads,
                             int dh, int max_pos, int dtype, void* stream) {
  if (!qkv || !rel_emb || !out || !probs || B <= 0 || T_ <= 0 || heads <= 0) return LIDK_ERR_ARG;
  if (!att_dh_ok(dh)) return LIDK_ERR_UNSUPPORTED;
  AttGeom g{B, T_, heads, dh, max_pos, heads * dh, 3 * heads * dh};
  dim3 grid(cdiv(T_, ATT_ROWS), heads, B);
  const float scale = 1.0f / sqrtf((float)dh);
  hipStream_t s = as_stream(stream);
  if (dtype == LIDK_BF16) {
    size_t lds = att_rows_lds<bf16>(T_, dh);
    if (lds > 160 * 1024) return LIDK_ERR_UNSUPPORTED;
    hipFuncSetAttribute((const void*)attn_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attn_fwd_kernel<bf16><<<grid, 256, lds, s>>>((const bf16*)qkv, rel_emb, (bf16*)out, (bf16*)probs, g, scale);
  } else if (dtype == LIDK_F32) {
    size_t lds = att_rows_lds<float>(T_, dh);
    if (lds > 160 * 1024) return LIDK_ERR_UNSUPPORTED;
    hipFuncSetAttribute((const void*)attn_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attn_fwd_kernel<float><<<grid, 256, lds, s>>>((const float*)qkv, rel_emb, (float*)out, (float*)probs, g, scale);
  } else {
    return LIDK_ERR_ARG;
  }
  return launch_status();
}

template <typename T, int DH>
static void att_cols_launch(const void* qkv, const void* probs, const void* dout, const float* dscores, void* dqkv,
                            float* demb, AttGeom g, float scale, hipStream_t s) {
  size_t lds = (size_t)2 * g.T * DH * sizeof(float);
  hipFuncSetAttribute((const void*)attn_bwd_cols_kernel<T, DH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_bwd_cols_kernel<T, DH><<<g.B * g.H, 256, lds, s>>>((const T*)qkv, (const T*)probs, (const T*)dout, dscores,
                                                           (T*)dqkv, demb, g, scale);
}

template <typename T>
static int att_bwd_launch(const void* qkv, const float* rel_emb, const void* probs, const void* dout, void* dqkv,
                          float* drel_emb, float* dscores, AttGeom g, hipStream_t s) {
  const float scale = 1.0f / sqrtf((float)g.dh);
  size_t lds = att_rows_lds<T>(g.T, g.dh);
  if (lds > 160 * 1024 || (size_t)2 * g.T * g.dh * sizeof(float) > 160 * 1024) return LIDK_ERR_UNSUPPORTED;
  dim3 grid(cdiv(g.T, ATT_ROWS), g.H, g.B);
  hipFuncSetAttribute((const void*)attn_bwd_rows_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_bwd_rows_kernel<T><<<grid, 256, lds, s>>>((const T*)qkv, rel_emb, (const T*)probs, (const T*)dout, (T*)dqkv,
                                                 dscores, g, scale);
  switch (g.dh) {
    case 8: att_cols_launch<T, 8>(qkv, probs, dout, dscores, dqkv, drel_emb, g, scale, s); break;
    case 16: att_cols_launch<T, 16>(qkv, probs, dout, dscores, dqkv, drel_emb, g, scale, s); break;
    case 32: att_cols_launch<T, 32>(qkv, probs, dout, dscores, dqkv, drel_emb, g, scale, s); break;
    case 64: att_cols_launch<T, 64>(qkv, probs, dout, dscores, dqkv, drel_emb, g, scale, s); break;
    default: return LIDK_ERR_UNSUPPORTED;
  }
  return launch_status();
}

extern "C" int lidk_attn_bwd(const void* qkv, const float* rel_emb, const void* probs, const void* dout, void* dqkv,
                             float* drel_emb, float* dscores, int B, int T_, int heads, int dh, int max_pos, int dtype,
                             void* stream) {
  if (!qkv || !rel_emb || !probs || !dout || !dqkv || !drel_emb || !dscores || B <= 0 || T_ <= 0 || heads <= 0)
    return LIDK_ERR_ARG;
  if (!att_dh_ok(dh)) return LIDK_ERR_UNSUPPORTED;
  AttGeom g{B, T_, heads, dh, max_pos, heads * dh, 3 * heads * dh};
  hipStream_t s = as_stream(stream);
  if (dtype == LIDK_BF16) return att_bwd_launch<bf16>(qkv, rel_emb, probs, dout, dqkv, drel_emb, dscores, g, s);
  if (dtype == LIDK_F32) return att_bwd_launch<float>(qkv, rel_emb, probs, dout, dqkv, drel_emb, dscores, g, s);
  return LIDK_ERR_ARG;
}
