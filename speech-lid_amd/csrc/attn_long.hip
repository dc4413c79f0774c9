// Attention with Shaw relative positions for LONG sequences (lid/conformer.py:117-148): key-tiled forms of the v1 kernels of
// attn.hip.  The fast paths keep a whole (batch, head)'s K, V and relative-embedding slice in LDS, which bounds T (MFMA path
// T <= 256, v1 path T <= ~390 forward / ~320 backward with dh = 64).  The reference's confs admit 12 s training utterances
// (T = 600) and 16.7 s validation utterances (T = 835), so beyond those bounds the same arithmetic runs here with K / V / E
// staged through LDS one KEY TILE at a time:
//   forward : 16 query rows per workgroup keep their full score rows in LDS ([16][T] f32); pass A walks key tiles (K tile +
//             the slice of E the tile's offsets i-j need) filling the scores, softmax in place, pass B walks V tiles.
//   backward rows: pass A (V tiles) dP = dO.V^T and delta, dS = P*(dP - delta) in place (+ dscores), pass B (K/E tiles) dq.
//   backward cols: one item (64 keys, or 64 relative offsets) per wave, the query/dO rows staged in chunks of 64.
// Same formulas, summation over keys in ascending order; parity is tested against the torch reference at T = 600 and 835.
// Throughput is secondary here (VALU dot products): the point is that no admissible utterance aborts training.
#include "common.h"

struct AttGeomL { int B, T, H, dh, max_pos, inner, ld; };

#define ATL_ROWS 16
#define ATL_TK 64
#define ATL_IC 64

template <typename T> struct AtlPad { static constexpr int v = 1; };
template <> struct AtlPad<bf16> { static constexpr int v = 2; };

template <typename T>
static size_t atl_rows_lds(int T_, int dh) {
  const int DHP = dh + AtlPad<T>::v;
  size_t a = (size_t)(2 * ATL_TK + ATL_TK + ATL_ROWS - 1) * DHP * sizeof(T);      // K tile, V tile, E slice
  a = (a + 15) / 16 * 16;
  return a + (size_t)ATL_ROWS * (T_ + dh) * sizeof(float);                         // score rows + q (or dO) rows
}

// stage rows j0 .. j0+TK-1 of the `which`-th block (1 = K, 2 = V) of qkv for head h
template <typename T>
__device__ __forceinline__ void atl_stage_kv(const T* __restrict__ qkv, const AttGeomL g, int b, int h, int j0, int which,
                                             T* dst, int DHP) {
  const int dh = g.dh;
  for (int idx = threadIdx.x; idx < ATL_TK * dh; idx += blockDim.x) {
    const int j = idx / dh, d = idx - j * dh;
    const int jj = min(j0 + j, g.T - 1);
    dst[j * DHP + d] = qkv[(size_t)(b * g.T + jj) * g.ld + which * g.inner + h * dh + d];
  }
}
// E slice for rows i0..i0+15 against keys j0..j0+TK-1: entry e <-> offset r = e + (i0 - j0 - TK + 1)
template <typename T>
__device__ __forceinline__ void atl_stage_e(const float* __restrict__ emb, const AttGeomL g, int i0, int j0, T* Es, int DHP) {
  const int dh = g.dh, NE = ATL_TK + ATL_ROWS - 1;
  for (int idx = threadIdx.x; idx < NE * dh; idx += blockDim.x) {
    const int e = idx / dh, d = idx - e * dh;
    int r = e + (i0 - j0 - ATL_TK + 1);
    r = max(-g.max_pos, min(g.max_pos, r)) + g.max_pos;
    Es[e * DHP + d] = from_f<T>(emb[(size_t)r * dh + d]);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
attn_fwd_long_kernel(const T* __restrict__ qkv, const float* __restrict__ emb, T* __restrict__ out, T* __restrict__ probs,
                     AttGeomL g, int ldp, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T_ = g.T, dh = g.dh, DHP = dh + AtlPad<T>::v, NE = ATL_TK + ATL_ROWS - 1;
  T* Ks = reinterpret_cast<T*>(smem);
  T* Vs = Ks + ATL_TK * DHP;
  T* Es = Vs + ATL_TK * DHP;
  float* ps = reinterpret_cast<float*>(smem + (((size_t)(2 * ATL_TK + NE) * DHP * sizeof(T)) + 15) / 16 * 16);   // [16][T]
  float* qs = ps + (size_t)ATL_ROWS * T_;                                                                          // [16][dh]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * ATL_ROWS;
  for (int idx = threadIdx.x; idx < ATL_ROWS * dh; idx += blockDim.x) {
    const int r = idx / dh, d = idx - r * dh, i = min(i0 + r, T_ - 1);
    qs[idx] = to_f(qkv[(size_t)(b * T_ + i) * g.ld + h * dh + d]);
  }
  // ---- pass A: scores
  for (int j0 = 0; j0 < T_; j0 += ATL_TK) {
    __syncthreads();
    atl_stage_kv<T>(qkv, g, b, h, j0, 1, Ks, DHP);
    atl_stage_e<T>(emb, g, i0, j0, Es, DHP);
    __syncthreads();
    const int nj = min(ATL_TK, T_ - j0);
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave * 4 + rr;
      if (i0 + r >= T_) break;
      for (int j = lane; j < nj; j += 64) {
        const T* krow = Ks + j * DHP;
        const T* erow = Es + (r - j + ATL_TK - 1) * DHP;
        const float* q = qs + r * dh;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s = fmaf(q[d], to_f(krow[d]) + to_f(erow[d]), s);
        ps[(size_t)r * T_ + j0 + j] = scale * s;
      }
    }
  }
  __syncthreads();
  // ---- softmax per row (a wave owns its 4 rows), probabilities written out
  for (int rr = 0; rr < 4; ++rr) {
    const int r = wave * 4 + rr, i = i0 + r;
    if (i >= T_) break;
    float* p = ps + (size_t)r * T_;
    float mx = -INFINITY;
    for (int j = lane; j < T_; j += 64) mx = fmaxf(mx, p[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < T_; j += 64) { float e = __expf(p[j] - mx); p[j] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    T* prow = probs + ((size_t)(b * g.H + h) * T_ + i) * ldp;
    for (int j = lane; j < T_; j += 64) { float v = p[j] * inv; p[j] = v; prow[j] = from_f<T>(v); }
    for (int j = T_ + lane; j < ldp; j += 64) prow[j] = from_f<T>(0.f);
  }
  // ---- pass B: O = P.V
  const int nparts = 64 / dh, dlane = lane % dh, part = lane / dh;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < T_; j0 += ATL_TK) {
    __syncthreads();
    atl_stage_kv<T>(qkv, g, b, h, j0, 2, Vs, DHP);
    __syncthreads();
    const int nj = min(ATL_TK, T_ - j0);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave * 4 + rr;
      if (i0 + r < T_)
        for (int j = part; j < nj; j += nparts) o[rr] = fmaf(ps[(size_t)r * T_ + j0 + j], to_f(Vs[j * DHP + dlane]), o[rr]);
    }
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int i = i0 + wave * 4 + rr;
    float v = o[rr];
    for (int off = dh; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    if (i < T_ && lane < dh) out[(size_t)(b * T_ + i) * g.inner + h * dh + lane] = from_f<T>(v);
  }
}

// backward, row pass: dP = dO.V^T ; delta = sum_j P*dP ; dS = P*(dP - delta) -> dscores ; dq = scale * sum_j dS (k_j + E_{i-j})
template <typename T>
__global__ void __launch_bounds__(256)
attn_bwd_rows_long_kernel(const T* __restrict__ qkv, const float* __restrict__ emb, const T* __restrict__ probs,
                          const T* __restrict__ dout, T* __restrict__ dqkv, float* __restrict__ dscores, AttGeomL g, int ldp,
                          float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int T_ = g.T, dh = g.dh, DHP = dh + AtlPad<T>::v, NE = ATL_TK + ATL_ROWS - 1;
  T* Ks = reinterpret_cast<T*>(smem);
  T* Vs = Ks + ATL_TK * DHP;
  T* Es = Vs + ATL_TK * DHP;
  float* ps = reinterpret_cast<float*>(smem + (((size_t)(2 * ATL_TK + NE) * DHP * sizeof(T)) + 15) / 16 * 16);
  float* dos = ps + (size_t)ATL_ROWS * T_;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * ATL_ROWS;
  for (int idx = threadIdx.x; idx < ATL_ROWS * dh; idx += blockDim.x) {
    const int r = idx / dh, d = idx - r * dh, i = min(i0 + r, T_ - 1);
    dos[idx] = to_f(dout[(size_t)(b * T_ + i) * g.inner + h * dh + d]);
  }
  for (int j0 = 0; j0 < T_; j0 += ATL_TK) {                       // pass A: dP rows
    __syncthreads();
    atl_stage_kv<T>(qkv, g, b, h, j0, 2, Vs, DHP);
    __syncthreads();
    const int nj = min(ATL_TK, T_ - j0);
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave * 4 + rr;
      if (i0 + r >= T_) break;
      for (int j = lane; j < nj; j += 64) {
        const T* vrow = Vs + j * DHP;
        const float* dd = dos + r * dh;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s = fmaf(dd[d], to_f(vrow[d]), s);
        ps[(size_t)r * T_ + j0 + j] = s;
      }
    }
  }
  __syncthreads();
  for (int rr = 0; rr < 4; ++rr) {
    const int r = wave * 4 + rr, i = i0 + r;
    if (i >= T_) break;
    float* p = ps + (size_t)r * T_;
    const T* prow = probs + ((size_t)(b * g.H + h) * T_ + i) * ldp;
    float delta = 0.f;
    for (int j = lane; j < T_; j += 64) delta = fmaf(to_f(prow[j]), p[j], delta);
    delta = wave_sum(delta);
    float* dsrow = dscores + ((size_t)(b * g.H + h) * T_ + i) * T_;
    for (int j = lane; j < T_; j += 64) { float ds = to_f(prow[j]) * (p[j] - delta); p[j] = ds; dsrow[j] = ds; }
  }
  const int nparts = 64 / dh, dlane = lane % dh, part = lane / dh;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j0 = 0; j0 < T_; j0 += ATL_TK) {                       // pass B: dq
    __syncthreads();
    atl_stage_kv<T>(qkv, g, b, h, j0, 1, Ks, DHP);
    atl_stage_e<T>(emb, g, i0, j0, Es, DHP);
    __syncthreads();
    const int nj = min(ATL_TK, T_ - j0);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave * 4 + rr;
      if (i0 + r < T_)
        for (int j = part; j < nj; j += nparts)
          a[rr] = fmaf(ps[(size_t)r * T_ + j0 + j], to_f(Ks[j * DHP + dlane]) + to_f(Es[(r - j + ATL_TK - 1) * DHP + dlane]), a[rr]);
    }
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int i = i0 + wave * 4 + rr;
    float v = a[rr];
    for (int off = dh; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    if (i < T_ && lane < dh) dqkv[(size_t)(b * T_ + i) * g.ld + h * dh + lane] = from_f<T>(v * scale);
  }
}

// backward, column pass.  item < nkc: lane <-> key j = item*64 + lane: dk[j] = scale * sum_i dS[i][j] q[i], dv[j] = sum_i P[i][j] dO[i];
// item >= nkc: lane <-> offset r: dE[clamp(r)] += scale * sum_i dS[i][i-r] q[i].  One item per wave; q / dO rows staged 64 at a time.
template <typename T, int DH>
__global__ void __launch_bounds__(256)
attn_bwd_cols_long_kernel(const T* __restrict__ qkv, const T* __restrict__ probs, const T* __restrict__ dout,
                          const float* __restrict__ dscores, T* __restrict__ dqkv, float* __restrict__ demb, AttGeomL g,
                          int ldp, float scale) {
  __shared__ float Qs[ATL_IC * DH];
  __shared__ float Ds[ATL_IC * DH];
  const int T_ = g.T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / g.H, h = blockIdx.x % g.H;
  const int nkc = (T_ + 63) / 64, nrc = (2 * T_ - 1 + 63) / 64;
  const int item = blockIdx.y * 4 + wave;
  const bool is_key = item < nkc, is_rel = !is_key && item < nkc + nrc;
  const int j = item * 64 + lane;                              // key items
  const int r = (item - nkc) * 64 + lane - (T_ - 1);           // offset items: i - j
  const size_t base = (size_t)(b * g.H + h) * T_ * T_;
  const size_t pbase = (size_t)(b * g.H + h) * T_ * ldp;
  float ak[DH], av[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { ak[d] = 0.f; av[d] = 0.f; }
  for (int ic0 = 0; ic0 < T_; ic0 += ATL_IC) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < ATL_IC * DH; idx += blockDim.x) {
      const int ii = idx / DH, d = idx - ii * DH, i = min(ic0 + ii, T_ - 1);
      Qs[idx] = to_f(qkv[(size_t)(b * T_ + i) * g.ld + h * DH + d]);
      Ds[idx] = to_f(dout[(size_t)(b * T_ + i) * g.inner + h * DH + d]);
    }
    __syncthreads();
    const int ihi_c = min(ic0 + ATL_IC, T_);
    if (is_key && j < T_) {
      for (int i = ic0; i < ihi_c; ++i) {
        const float s = dscores[base + (size_t)i * T_ + j];
        const float p = to_f(probs[pbase + (size_t)i * ldp + j]);
        const float* q = Qs + (i - ic0) * DH;
        const float* dd = Ds + (i - ic0) * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) { ak[d] = fmaf(s, q[d], ak[d]); av[d] = fmaf(p, dd[d], av[d]); }
      }
    } else if (is_rel && r <= T_ - 1) {
      const int ilo = max(max(0, r), ic0), ihi = min(min(T_ - 1, T_ - 1 + r), ihi_c - 1);
      for (int i = ilo; i <= ihi; ++i) {
        const float s = dscores[base + (size_t)i * T_ + (i - r)];
        const float* q = Qs + (i - ic0) * DH;
#pragma unroll
        for (int d = 0; d < DH; ++d) ak[d] = fmaf(s, q[d], ak[d]);
      }
    }
  }
  if (is_key && j < T_) {
    T* krow = dqkv + (size_t)(b * T_ + j) * g.ld + g.inner + h * DH;
    T* vrow = krow + g.inner;
#pragma unroll
    for (int d = 0; d < DH; ++d) { krow[d] = from_f<T>(ak[d] * scale); vrow[d] = from_f<T>(av[d]); }
  } else if (is_rel && r <= T_ - 1 && demb) {
    const int row = max(-g.max_pos, min(g.max_pos, r)) + g.max_pos;
#pragma unroll
    for (int d = 0; d < DH; ++d) atomicAdd(&demb[(size_t)row * DH + d], ak[d] * scale);
  }
}

// ------------------------------------------------------------------------------------ host side (called from attn.hip)
template <typename T>
int att_long_fwd(const void* qkv, const float* emb, void* out, void* probs, int ldp, int B, int T_, int H, int dh, int max_pos,
                 hipStream_t s) {
  const size_t lds = atl_rows_lds<T>(T_, dh);
  if (lds > 160 * 1024) return LIDK_ERR_UNSUPPORTED;
  AttGeomL g{B, T_, H, dh, max_pos, H * dh, 3 * H * dh};
  dim3 grid(cdiv(T_, ATL_ROWS), H, B);
  (void)hipFuncSetAttribute((const void*)attn_fwd_long_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_fwd_long_kernel<T><<<grid, 256, lds, s>>>((const T*)qkv, emb, (T*)out, (T*)probs, g, ldp, 1.0f / sqrtf((float)dh));
  return launch_status();
}

template <typename T, int DH>
static void atl_cols_launch(const void* qkv, const void* probs, const void* dout, const float* dscores, void* dqkv, float* demb,
                            AttGeomL g, int ldp, float scale, hipStream_t s) {
  const int items = (g.T + 63) / 64 + (2 * g.T - 1 + 63) / 64;
  dim3 grid(g.B * g.H, cdiv(items, 4));
  attn_bwd_cols_long_kernel<T, DH><<<grid, 256, 0, s>>>((const T*)qkv, (const T*)probs, (const T*)dout, dscores, (T*)dqkv, demb,
                                                        g, ldp, scale);
}

template <typename T>
int att_long_bwd(const void* qkv, const float* emb, const void* probs, int ldp, const void* dout, void* dqkv, float* demb,
                 float* dscores, int B, int T_, int H, int dh, int max_pos, hipStream_t s) {
  const size_t lds = atl_rows_lds<T>(T_, dh);
  if (lds > 160 * 1024) return LIDK_ERR_UNSUPPORTED;
  AttGeomL g{B, T_, H, dh, max_pos, H * dh, 3 * H * dh};
  const float scale = 1.0f / sqrtf((float)dh);
  dim3 grid(cdiv(T_, ATL_ROWS), H, B);
  (void)hipFuncSetAttribute((const void*)attn_bwd_rows_long_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  attn_bwd_rows_long_kernel<T><<<grid, 256, lds, s>>>((const T*)qkv, emb, (const T*)probs, (const T*)dout, (T*)dqkv, dscores, g,
                                                      ldp, scale);
  switch (dh) {
    case 8: atl_cols_launch<T, 8>(qkv, probs, dout, dscores, dqkv, demb, g, ldp, scale, s); break;
    case 16: atl_cols_launch<T, 16>(qkv, probs, dout, dscores, dqkv, demb, g, ldp, scale, s); break;
    case 32: atl_cols_launch<T, 32>(qkv, probs, dout, dscores, dqkv, demb, g, ldp, scale, s); break;
    case 64: atl_cols_launch<T, 64>(qkv, probs, dout, dscores, dqkv, demb, g, ldp, scale, s); break;
    default: return LIDK_ERR_UNSUPPORTED;
  }
  return launch_status();
}

// only the key-side kernel (dK, dV, relative-embedding gradient from probs + dscores): attn_shaw.hip supplies the row side
int att_long_cols_bf16(const void* qkv, const void* probs, int ldp, const void* dout, const float* dscores, void* dqkv, float* demb,
                       int B, int T_, int H, int dh, int max_pos, hipStream_t s) {
  AttGeomL g{B, T_, H, dh, max_pos, H * dh, 3 * H * dh};
  const float scale = 1.0f / sqrtf((float)dh);
  if (dh == 64) atl_cols_launch<bf16, 64>(qkv, probs, dout, dscores, dqkv, demb, g, ldp, scale, s);
  else if (dh == 32) atl_cols_launch<bf16, 32>(qkv, probs, dout, dscores, dqkv, demb, g, ldp, scale, s);
  else return LIDK_ERR_UNSUPPORTED;
  return launch_status();
}

template int att_long_fwd<bf16>(const void*, const float*, void*, void*, int, int, int, int, int, int, hipStream_t);
template int att_long_fwd<float>(const void*, const float*, void*, void*, int, int, int, int, int, int, hipStream_t);
template int att_long_bwd<bf16>(const void*, const float*, const void*, int, const void*, void*, float*, float*, int, int, int,
                                int, int, hipStream_t);
template int att_long_bwd<float>(const void*, const float*, const void*, int, const void*, void*, float*, float*, int, int, int,
                                 int, int, hipStream_t);

// largest T the key-tiled kernels take for a head dimension (LDS: 16 score rows of T floats + tiles)
extern "C" int lidk_attn_max_frames(int dh, int dtype) {
  int lo = 1, hi = 1 << 16;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    const size_t need = dtype == LIDK_BF16 ? atl_rows_lds<bf16>(mid, dh) : atl_rows_lds<float>(mid, dh);
    if (need <= 160 * 1024) lo = mid; else hi = mid - 1;
  }
  return lo;
}
