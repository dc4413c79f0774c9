// ConformerConvModule pieces (lid/conformer.py:47-65,174-205), channel-last [B][T][C]:
// GLU, depthwise Conv1d (k <= 32) with fused BatchNorm partial statistics, BatchNorm(+Swish) forward/backward.
// All HBM-bound: lanes run over channels (coalesced), the time window slides through registers.
#include "common.h"

#define DW_TT 32      // output time steps per workgroup (4 waves x 8)
#define DW_KMAX 32
#define DW_ROWS (DW_TT + DW_KMAX)   // staged input rows (zero beyond the needed 32+K-1)

// ------------------------------------------------------------------------------------ GLU
template <typename T>
__global__ void glu_fwd_kernel(const T* __restrict__ y, T* __restrict__ g, long M, int C) {
  long n4 = M * C / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    long e = i * 4, m = e / C; int c = (int)(e - m * C);
    float4 a = load4(y + m * 2 * C + c), b = load4(y + m * 2 * C + C + c), o;
    o.x = a.x * sigmoidf_(b.x); o.y = a.y * sigmoidf_(b.y); o.z = a.z * sigmoidf_(b.z); o.w = a.w * sigmoidf_(b.w);
    store4(g + e, o);
  }
}
template <typename T>
__global__ void glu_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dg, T* __restrict__ dy, long M, int C) {
  long n4 = M * C / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    long e = i * 4, m = e / C; int c = (int)(e - m * C);
    float4 a = load4(y + m * 2 * C + c), b = load4(y + m * 2 * C + C + c), d = load4(dg + e), oa, ob;
    float s;
    s = sigmoidf_(b.x); oa.x = d.x * s; ob.x = d.x * a.x * s * (1.f - s);
    s = sigmoidf_(b.y); oa.y = d.y * s; ob.y = d.y * a.y * s * (1.f - s);
    s = sigmoidf_(b.z); oa.z = d.z * s; ob.z = d.z * a.z * s * (1.f - s);
    s = sigmoidf_(b.w); oa.w = d.w * s; ob.w = d.w * a.w * s * (1.f - s);
    store4(dy + m * 2 * C + c, oa);
    store4(dy + m * 2 * C + C + c, ob);
  }
}
static int ew_blocks(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

extern "C" int lidk_glu_fwd(const void* y, void* g, int M, int C, int dtype, void* stream) {
  if (!y || !g || M <= 0 || C <= 0 || (C & 3)) return LIDK_ERR_ARG;
  LIDK_DISPATCH(dtype, glu_fwd_kernel<T><<<ew_blocks((long)M * C / 4), 256, 0, as_stream(stream)>>>((const T*)y, (T*)g, M, C));
  return launch_status();
}
extern "C" int lidk_glu_bwd(const void* y, const void* dg, void* dy, int M, int C, int dtype, void* stream) {
  if (!y || !dg || !dy || M <= 0 || C <= 0 || (C & 3)) return LIDK_ERR_ARG;
  LIDK_DISPATCH(dtype, glu_bwd_kernel<T><<<ew_blocks((long)M * C / 4), 256, 0, as_stream(stream)>>>((const T*)y, (const T*)dg, (T*)dy, M, C));
  return launch_status();
}

__device__ __forceinline__ float swish_grad(float z) { float s = sigmoidf_(z); return s * (1.f + z * (1.f - s)); }

// BatchNorm(+Swish) backward "apply" step as a prologue of the depthwise-conv input gradient (MODE 3)
struct BnBwd { const void* c; const float *mean, *rstd, *gamma, *beta; const double* sums; double count; };

// ------------------------------------------------------------------------------------ depthwise conv (fwd and dgrad)
// out[b][t][ch] = bias[ch] + sum_k w[ch][k] * in[b][t + k - pad_left][ch]   (flip=1: w[ch][K-1-k], used for dgrad)
// MODE 1 (forward, GLU fused in front): `in` is the pre-GLU tensor y [.][2C]; the tile is filled with y_a * sigmoid(y_gate)
//         and the workgroup's own rows of that product are also written to gout (the weight gradient reads it later).
// MODE 2 (dgrad, GLU backward fused behind): the result dg is not stored; out is dy [.][2C] with
//         dy_a = dg * sigmoid(gate), dy_gate = dg * a * sigmoid(gate) * (1 - sigmoid(gate)), a/gate read from yglu.
// MODE 3 = MODE 2 with the BatchNorm+Swish backward in front: `in` is ds (gradient at the Swish output) and the tile is
//         filled with dc = gamma*rstd*(dz - sums0/count - xhat*sums1/count), dz = ds*swish'(z), computed from ds and bn.c.
template <typename T, int MODE>
__global__ void __launch_bounds__(256)
dwconv_kernel(const T* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ out,
              float* __restrict__ stat_partial, int B, int T_, int C, int K, int pad_left, int flip,
              const T* __restrict__ yglu, T* __restrict__ gout, BnBwd bn) {
  __shared__ __attribute__((aligned(16))) T tile[DW_ROWS][64];
  __shared__ float red[4][2][64];
  __shared__ __attribute__((aligned(16))) float wsh[64 * DW_KMAX];     // this block's 64 x K taps: one coalesced load; a lane's K taps are then a stride-K
                                          // LDS walk (K odd -> conflict free) instead of K global gathers of 64 cache lines each
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = blockIdx.x * DW_TT, c0 = blockIdx.y * 64, b = blockIdx.z;
  {
    const int nw = min(64, C - c0) * K;
    for (int i = threadIdx.x; i < nw; i += 256) wsh[i] = w[(size_t)c0 * K + i];
  }
  const int ch = c0 + lane;
  const bool chok = ch < C;
  // stage [rows][64 channels] with 16-byte loads (8 bf16 / 4 f32 channels per lane) when the channel chunk is full
  constexpr int VE = 16 / sizeof(T), CPR = 64 / VE;
  if (MODE == 1) {
    // GLU on the way in: 4 channels per thread (8/16-byte accesses of both halves of y); requires C % 4 == 0 (host check)
    for (int q = threadIdx.x; q < DW_ROWS * 16; q += 256) {
      int r = q >> 4, cc = (q & 15) * 4, t = t0 - pad_left + r;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool live = t >= 0 && t < T_ && r < DW_TT + K - 1 && c0 + cc < C;
      if (live) {
        const T* yr = in + ((size_t)b * T_ + t) * 2 * C + c0 + cc;
        float4 a = load4(yr), gt = load4(yr + C);
        o.x = a.x * sigmoidf_(gt.x); o.y = a.y * sigmoidf_(gt.y); o.z = a.z * sigmoidf_(gt.z); o.w = a.w * sigmoidf_(gt.w);
        if (gout && r >= pad_left && r < pad_left + DW_TT) store4(gout + ((size_t)b * T_ + t) * C + c0 + cc, o);
      }
      store4(&tile[r][cc], o);
    }
  } else if (MODE == 3) {
    // 4 channels per thread; a thread keeps the same channels for all its rows (256 % 16 == 0), so the per-channel
    // constants are loaded once
    const int cc = (threadIdx.x & 15) * 4;
    const bool cok = c0 + cc < C;
    float4 mu = make_float4(0, 0, 0, 0), rs = mu, gm = mu, bt = mu, m0 = mu, m1 = mu;
    if (cok) {
      const int cg = c0 + cc;
      const double cnt = bn.count > 0 ? bn.count : bn.sums[2 * C];      // count <= 0: the global row count rides behind the sums
      mu = load4(bn.mean + cg); rs = load4(bn.rstd + cg); gm = load4(bn.gamma + cg); bt = load4(bn.beta + cg);
      m0 = make_float4((float)(bn.sums[cg] / cnt), (float)(bn.sums[cg + 1] / cnt), (float)(bn.sums[cg + 2] / cnt),
                       (float)(bn.sums[cg + 3] / cnt));
      m1 = make_float4((float)(bn.sums[C + cg] / cnt), (float)(bn.sums[C + cg + 1] / cnt),
                       (float)(bn.sums[C + cg + 2] / cnt), (float)(bn.sums[C + cg + 3] / cnt));
    }
    const T* cbuf = (const T*)bn.c;
    for (int q = threadIdx.x; q < DW_ROWS * 16; q += 256) {
      int r = q >> 4, t = t0 - pad_left + r;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cok && t >= 0 && t < T_ && r < DW_TT + K - 1) {
        const size_t e = ((size_t)b * T_ + t) * C + c0 + cc;
        float4 x = load4(cbuf + e), d = load4(in + e);
        float xh, dz;
        xh = (x.x - mu.x) * rs.x; dz = d.x * swish_grad(xh * gm.x + bt.x); o.x = gm.x * rs.x * (dz - m0.x - xh * m1.x);
        xh = (x.y - mu.y) * rs.y; dz = d.y * swish_grad(xh * gm.y + bt.y); o.y = gm.y * rs.y * (dz - m0.y - xh * m1.y);
        xh = (x.z - mu.z) * rs.z; dz = d.z * swish_grad(xh * gm.z + bt.z); o.z = gm.z * rs.z * (dz - m0.z - xh * m1.z);
        xh = (x.w - mu.w) * rs.w; dz = d.w * swish_grad(xh * gm.w + bt.w); o.w = gm.w * rs.w * (dz - m0.w - xh * m1.w);
      }
      store4(&tile[r][cc], o);
    }
  } else if (c0 + 64 <= C && (C % VE) == 0) {
    for (int q = threadIdx.x; q < DW_ROWS * CPR; q += 256) {
      int r = q / CPR, cc = (q % CPR) * VE, t = t0 - pad_left + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (t >= 0 && t < T_ && r < DW_TT + K - 1) v = *reinterpret_cast<const uint4*>(in + ((size_t)b * T_ + t) * C + c0 + cc);
      *reinterpret_cast<uint4*>(&tile[r][cc]) = v;
    }
  } else {
    for (int r = wave; r < DW_ROWS; r += 4) {
      int t = t0 - pad_left + r;
      T v = from_f<T>(0.f);
      if (chok && t >= 0 && t < T_ && r < DW_TT + K - 1) v = in[((size_t)b * T_ + t) * C + ch];
      tile[r][lane] = v;
    }
  }
  const float bv = (bias && chok) ? bias[ch] : 0.f;
  __syncthreads();
  float wr[DW_KMAX];
#pragma unroll
  for (int k = 0; k < DW_KMAX; ++k) wr[k] = (chok && k < K) ? wsh[lane * K + (flip ? K - 1 - k : k)] : 0.f;
  float x[8 + DW_KMAX - 1];
#pragma unroll
  for (int r = 0; r < 8 + DW_KMAX - 1; ++r) x[r] = to_f(tile[wave * 8 + r][lane]);
  float s1 = 0.f, s2 = 0.f;
  float res[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float acc = bv;
#pragma unroll
    for (int k = 0; k < DW_KMAX; ++k) acc = fmaf(wr[k], x[o + k], acc);
    res[o] = acc;
    if (chok && t0 + wave * 8 + o < T_) { s1 += acc; s2 = fmaf(acc, acc, s2); }
  }
  if ((C & 3) == 0) {
    // results -> LDS (the input tile is dead) -> epilogue with 4 channels per thread: 8/16-byte loads of y and stores,
    // instead of one 2-byte access per lane and row
    __syncthreads();
    float (*ot)[64] = reinterpret_cast<float (*)[64]>(&wsh[0]);     // 32 x 64 f32 = the taps' 8 KB, no longer needed
#pragma unroll
    for (int o = 0; o < 8; ++o) ot[wave * 8 + o][lane] = res[o];
    __syncthreads();
    for (int q = threadIdx.x; q < DW_TT * 16; q += 256) {
      const int r = q >> 4, cc = (q & 15) * 4, t = t0 + r;
      if (t >= T_ || c0 + cc >= C) continue;
      const float4 v = *reinterpret_cast<const float4*>(&ot[r][cc]);
      if (MODE == 2 || MODE == 3) {
        const T* yr = yglu + ((size_t)b * T_ + t) * 2 * C + c0 + cc;
        const float4 a = load4(yr), gt = load4(yr + C);
        float4 da, dgt;
        float sg;
        sg = sigmoidf_(gt.x); da.x = v.x * sg; dgt.x = v.x * a.x * sg * (1.f - sg);
        sg = sigmoidf_(gt.y); da.y = v.y * sg; dgt.y = v.y * a.y * sg * (1.f - sg);
        sg = sigmoidf_(gt.z); da.z = v.z * sg; dgt.z = v.z * a.z * sg * (1.f - sg);
        sg = sigmoidf_(gt.w); da.w = v.w * sg; dgt.w = v.w * a.w * sg * (1.f - sg);
        T* dr = out + ((size_t)b * T_ + t) * 2 * C + c0 + cc;
        store4(dr, da);
        store4(dr + C, dgt);
      } else {
        store4(out + ((size_t)b * T_ + t) * C + c0 + cc, v);
      }
    }
  } else {
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      int t = t0 + wave * 8 + o;
      if (chok && t < T_) {
        if (MODE == 2 || MODE == 3) {
          const T* yr = yglu + ((size_t)b * T_ + t) * 2 * C + ch;
          const float a = to_f(yr[0]), sg = sigmoidf_(to_f(yr[C]));
          T* dr = out + ((size_t)b * T_ + t) * 2 * C + ch;
          dr[0] = from_f<T>(res[o] * sg);
          dr[C] = from_f<T>(res[o] * a * sg * (1.f - sg));
        } else {
          out[((size_t)b * T_ + t) * C + ch] = from_f<T>(res[o]);
        }
      }
    }
  }
  if (stat_partial) {
    red[wave][0][lane] = s1; red[wave][1][lane] = s2;
    __syncthreads();
    if (wave == 0 && chok) {
      size_t p = (size_t)b * gridDim.x + blockIdx.x;
      stat_partial[(p * 2 + 0) * C + ch] = red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane];
      stat_partial[(p * 2 + 1) * C + ch] = red[0][1][lane] + red[1][1][lane] + red[2][1][lane] + red[3][1][lane];
    }
  }
}


// ------------------------------------------------------------------------------------ depthwise conv, long-tile bf16 form
// The same arithmetic as dwconv_kernel MODE 1 / MODE 3 (bf16-rounded tile, f32 taps and accumulation: bit-identical results),
// re-tiled for the training shapes: DW2_TT = 80 output steps per workgroup instead of 32 (halo re-reads 1.94x -> 1.375x and 2.5x
// fewer workgroups re-staging the taps), 16-byte accesses (8 channels per thread) on every global stream, per-channel
// BatchNorm constants staged once in LDS.  A wave slides a 20-output window through registers.
template <int MODE, int DW2_TT>
__global__ void __launch_bounds__(256)
dwconv2_bf16_kernel(const bf16* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, bf16* __restrict__ out,
                    float* __restrict__ stat_partial, int B, int T_, int C, int K, int pad_left, int flip,
                    const bf16* __restrict__ yglu, bf16* __restrict__ gout, BnBwd bn) {
  constexpr int OPW = DW2_TT / 4, DW2_ROWS = DW2_TT + DW_KMAX;
  constexpr int TILE_BYTES = DW2_ROWS * 64 * 2, OT_BYTES = DW2_TT * 64 * 4;
  constexpr int SM_BYTES = (TILE_BYTES + 64 * DW_KMAX * 4) > OT_BYTES ? (TILE_BYTES + 64 * DW_KMAX * 4) : OT_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM_BYTES];
  __shared__ float red[4][2][64];
  __shared__ __attribute__((aligned(16))) float cst[6][64];          // MODE 3: mean, rstd, gamma, beta, sum0/count, sum1/count
  bf16 (*tile)[64] = reinterpret_cast<bf16 (*)[64]>(smem);
  float* wsh = reinterpret_cast<float*>(smem + TILE_BYTES);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t0 = blockIdx.x * DW2_TT, c0 = blockIdx.y * 64, b = blockIdx.z;
  {
    const int nw = min(64, C - c0) * K;
    for (int i = threadIdx.x; i < nw; i += 256) wsh[i] = w[(size_t)c0 * K + i];
  }
  const int ch = c0 + lane;
  const bool chok = ch < C;
  if (MODE == 3) {
    if (threadIdx.x < 64) {
      float mu = 0.f, rs = 0.f, gm = 0.f, bt = 0.f, m0 = 0.f, m1 = 0.f;
      if (chok) {
        const double cnt = bn.count > 0 ? bn.count : bn.sums[2 * C];
        mu = bn.mean[ch]; rs = bn.rstd[ch]; gm = bn.gamma[ch]; bt = bn.beta[ch];
        m0 = (float)(bn.sums[ch] / cnt); m1 = (float)(bn.sums[C + ch] / cnt);
      }
      cst[0][lane] = mu; cst[1][lane] = rs; cst[2][lane] = gm; cst[3][lane] = bt; cst[4][lane] = m0; cst[5][lane] = m1;
    }
    __syncthreads();
  }
  // stage rows t0 - pad_left .. of 64 channels: 8 channels (16 bytes) per thread, 32 rows per pass; a thread keeps its channels
  {
    const int cc = (threadIdx.x & 7) * 8;
    const bool cok = c0 + cc < C;                                    // C % 8 == 0 (host check): whole vectors in or out
    const int need = DW2_TT + K - 1;
    f32x8_t mu, rs, gm, bt, m0, m1;
    if (MODE == 3) {
      mu.lo = *reinterpret_cast<const float4*>(&cst[0][cc]); mu.hi = *reinterpret_cast<const float4*>(&cst[0][cc + 4]);
      rs.lo = *reinterpret_cast<const float4*>(&cst[1][cc]); rs.hi = *reinterpret_cast<const float4*>(&cst[1][cc + 4]);
      gm.lo = *reinterpret_cast<const float4*>(&cst[2][cc]); gm.hi = *reinterpret_cast<const float4*>(&cst[2][cc + 4]);
      bt.lo = *reinterpret_cast<const float4*>(&cst[3][cc]); bt.hi = *reinterpret_cast<const float4*>(&cst[3][cc + 4]);
      m0.lo = *reinterpret_cast<const float4*>(&cst[4][cc]); m0.hi = *reinterpret_cast<const float4*>(&cst[4][cc + 4]);
      m1.lo = *reinterpret_cast<const float4*>(&cst[5][cc]); m1.hi = *reinterpret_cast<const float4*>(&cst[5][cc + 4]);
    }
    for (int r = threadIdx.x >> 3; r < DW2_ROWS; r += 32) {
      const int t = t0 - pad_left + r;
      f32x8_t o;
      o.lo = make_float4(0.f, 0.f, 0.f, 0.f); o.hi = o.lo;
      if (cok && t >= 0 && t < T_ && r < need) {
        if (MODE == 1) {
          const bf16* yr = in + ((size_t)b * T_ + t) * 2 * C + c0 + cc;
          const f32x8_t a = load8(yr), gt = load8(yr + C);
          o.lo.x = a.lo.x * sigmoidf_(gt.lo.x); o.lo.y = a.lo.y * sigmoidf_(gt.lo.y);
          o.lo.z = a.lo.z * sigmoidf_(gt.lo.z); o.lo.w = a.lo.w * sigmoidf_(gt.lo.w);
          o.hi.x = a.hi.x * sigmoidf_(gt.hi.x); o.hi.y = a.hi.y * sigmoidf_(gt.hi.y);
          o.hi.z = a.hi.z * sigmoidf_(gt.hi.z); o.hi.w = a.hi.w * sigmoidf_(gt.hi.w);
          if (gout && r >= pad_left && r < pad_left + DW2_TT) store8(gout + ((size_t)b * T_ + t) * C + c0 + cc, o);
        } else if (MODE == 0) {
          o = load8(in + ((size_t)b * T_ + t) * C + c0 + cc);
        } else {
          const size_t e = ((size_t)b * T_ + t) * C + c0 + cc;
          const f32x8_t x = load8(reinterpret_cast<const bf16*>(bn.c) + e), d = load8(in + e);
          float xh, dz;
#define DW2_BN(F, H)                                                                                                     \
          xh = (x.H.F - mu.H.F) * rs.H.F; dz = d.H.F * swish_grad(xh * gm.H.F + bt.H.F);                                 \
          o.H.F = gm.H.F * rs.H.F * (dz - m0.H.F - xh * m1.H.F);
          DW2_BN(x, lo) DW2_BN(y, lo) DW2_BN(z, lo) DW2_BN(w, lo) DW2_BN(x, hi) DW2_BN(y, hi) DW2_BN(z, hi) DW2_BN(w, hi)
#undef DW2_BN
        }
      }
      store8(&tile[r][cc], o);
    }
  }
  const float bv = (bias && chok) ? bias[ch] : 0.f;
  __syncthreads();
  float wr[DW_KMAX];
#pragma unroll
  for (int k = 0; k < DW_KMAX; ++k) wr[k] = (chok && k < K) ? wsh[lane * K + (flip ? K - 1 - k : k)] : 0.f;
  float res[OPW];
  float s1 = 0.f, s2 = 0.f;
  {
    float x[OPW + DW_KMAX - 1];
#pragma unroll
    for (int r = 0; r < OPW + DW_KMAX - 1; ++r) x[r] = to_f(tile[wave * OPW + r][lane]);
#pragma unroll
    for (int o = 0; o < OPW; ++o) {
      float acc = bv;
#pragma unroll
      for (int k = 0; k < DW_KMAX; ++k) acc = fmaf(wr[k], x[o + k], acc);
      res[o] = acc;
      if (chok && t0 + wave * OPW + o < T_) { s1 += acc; s2 = fmaf(acc, acc, s2); }
    }
  }
  __syncthreads();                                                   // the input tile is dead: the f32 result tile takes its place
  float (*ot)[64] = reinterpret_cast<float (*)[64]>(smem);          // DW2_TT x 64 f32 = 20 KB <= tile + taps
#pragma unroll
  for (int o = 0; o < OPW; ++o) ot[wave * OPW + o][lane] = res[o];
  if (stat_partial) { red[wave][0][lane] = s1; red[wave][1][lane] = s2; }
  __syncthreads();
  {
    const int cc = (threadIdx.x & 7) * 8;
    if (c0 + cc < C)
      for (int r = threadIdx.x >> 3; r < DW2_TT; r += 32) {
        const int t = t0 + r;
        if (t >= T_) break;
        f32x8_t v;
        v.lo = *reinterpret_cast<const float4*>(&ot[r][cc]); v.hi = *reinterpret_cast<const float4*>(&ot[r][cc + 4]);
        if (MODE == 3) {
          const bf16* yr = yglu + ((size_t)b * T_ + t) * 2 * C + c0 + cc;
          const f32x8_t a = load8(yr), gt = load8(yr + C);
          f32x8_t da, dgt;
          float sg;
#define DW2_GLU(F, H) sg = sigmoidf_(gt.H.F); da.H.F = v.H.F * sg; dgt.H.F = v.H.F * a.H.F * sg * (1.f - sg);
          DW2_GLU(x, lo) DW2_GLU(y, lo) DW2_GLU(z, lo) DW2_GLU(w, lo) DW2_GLU(x, hi) DW2_GLU(y, hi) DW2_GLU(z, hi) DW2_GLU(w, hi)
#undef DW2_GLU
          bf16* dr = out + ((size_t)b * T_ + t) * 2 * C + c0 + cc;
          store8(dr, da);
          store8(dr + C, dgt);
        } else {
          store8(out + ((size_t)b * T_ + t) * C + c0 + cc, v);
        }
      }
  }
  if (stat_partial && wave == 0 && chok) {
    const size_t p = (size_t)b * gridDim.x + blockIdx.x;
    stat_partial[(p * 2 + 0) * C + ch] = red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane];
    stat_partial[(p * 2 + 1) * C + ch] = red[0][1][lane] + red[1][1][lane] + red[2][1][lane] + red[3][1][lane];
  }
}
static inline bool dw2_ok(int C, int dtype) {
  static const int off = getenv("LIDK_DWCONV_V1") ? atoi(getenv("LIDK_DWCONV_V1")) : 0;
  return dtype == LIDK_BF16 && (C & 7) == 0 && !off;
}
static inline int dw2_tt() {
  static const int tt = getenv("LIDK_DW2_TT") ? atoi(getenv("LIDK_DW2_TT")) : 80;
  return tt == 40 || tt == 64 || tt == 112 ? tt : 80;
}
#define DW2_LAUNCH(MODE_, ...)                                                                                   \
  do {                                                                                                           \
    const int tt_ = dw2_tt();                                                                                    \
    dim3 grid2(cdiv(T_, tt_), cdiv(C, 64), B);                                                                   \
    if (tt_ == 40) dwconv2_bf16_kernel<MODE_, 40><<<grid2, 256, 0, as_stream(stream)>>>(__VA_ARGS__);            \
    else if (tt_ == 64) dwconv2_bf16_kernel<MODE_, 64><<<grid2, 256, 0, as_stream(stream)>>>(__VA_ARGS__);       \
    else if (tt_ == 112) dwconv2_bf16_kernel<MODE_, 112><<<grid2, 256, 0, as_stream(stream)>>>(__VA_ARGS__);     \
    else dwconv2_bf16_kernel<MODE_, 80><<<grid2, 256, 0, as_stream(stream)>>>(__VA_ARGS__);                      \
  } while (0)

extern "C" int lidk_dwconv_stat_parts(int B, int T_, int C, int dtype) { return B * cdiv(T_, dw2_ok(C, dtype) ? dw2_tt() : DW_TT); }

extern "C" int lidk_dwconv_fwd(const void* g, const float* w, const float* bias, void* c, float* stat_partial, int B,
                               int T_, int C, int K, int pad_left, int dtype, void* stream) {
  if (!g || !w || !c || B <= 0 || T_ <= 0 || C <= 0 || K <= 0 || K > DW_KMAX || pad_left < 0 || pad_left >= K) return LIDK_ERR_ARG;
  if (dw2_ok(C, dtype)) {
    DW2_LAUNCH(0, (const bf16*)g, w, bias, (bf16*)c, stat_partial, B, T_, C, K, pad_left, 0, nullptr, nullptr, BnBwd{});
    return launch_status();
  }
  dim3 grid(cdiv(T_, DW_TT), cdiv(C, 64), B);
  LIDK_DISPATCH(dtype, (dwconv_kernel<T, 0><<<grid, 256, 0, as_stream(stream)>>>((const T*)g, w, bias, (T*)c, stat_partial, B,
                                                                                T_, C, K, pad_left, 0, nullptr, nullptr, BnBwd{})));
  return launch_status();
}

extern "C" int lidk_glu_dwconv_fwd(const void* y, const float* w, const float* bias, void* g, void* c, float* stat_partial,
                                   int B, int T_, int C, int K, int pad_left, int dtype, void* stream) {
  if (!y || !w || !c || B <= 0 || T_ <= 0 || C <= 0 || (C & 3) || K <= 0 || K > DW_KMAX || pad_left < 0 || pad_left >= K)
    return LIDK_ERR_ARG;
  if (dw2_ok(C, dtype)) {
    DW2_LAUNCH(1, (const bf16*)y, w, bias, (bf16*)c, stat_partial, B, T_, C, K, pad_left, 0, nullptr, (bf16*)g, BnBwd{});
    return launch_status();
  }
  dim3 grid(cdiv(T_, DW_TT), cdiv(C, 64), B);
  LIDK_DISPATCH(dtype, (dwconv_kernel<T, 1><<<grid, 256, 0, as_stream(stream)>>>((const T*)y, w, bias, (T*)c, stat_partial, B,
                                                                                T_, C, K, pad_left, 0, nullptr, (T*)g, BnBwd{})));
  return launch_status();
}

extern "C" int lidk_dwconv_bwd_input(const void* dc, const float* w, void* dg, int B, int T_, int C, int K, int pad_left,
                                     int dtype, void* stream) {
  if (!dc || !w || !dg || B <= 0 || T_ <= 0 || C <= 0 || K <= 0 || K > DW_KMAX || pad_left < 0 || pad_left >= K) return LIDK_ERR_ARG;
  dim3 grid(cdiv(T_, DW_TT), cdiv(C, 64), B);
  LIDK_DISPATCH(dtype, (dwconv_kernel<T, 0><<<grid, 256, 0, as_stream(stream)>>>((const T*)dc, w, nullptr, (T*)dg, nullptr, B,
                                                                                T_, C, K, K - 1 - pad_left, 1, nullptr, nullptr, BnBwd{})));
  return launch_status();
}

extern "C" int lidk_dwconv_bwd_input_glu(const void* dc, const float* w, const void* y, void* dy, int B, int T_, int C, int K,
                                         int pad_left, int dtype, void* stream) {
  if (!dc || !w || !y || !dy || B <= 0 || T_ <= 0 || C <= 0 || K <= 0 || K > DW_KMAX || pad_left < 0 || pad_left >= K) return LIDK_ERR_ARG;
  dim3 grid(cdiv(T_, DW_TT), cdiv(C, 64), B);
  LIDK_DISPATCH(dtype, (dwconv_kernel<T, 2><<<grid, 256, 0, as_stream(stream)>>>((const T*)dc, w, nullptr, (T*)dy, nullptr, B,
                                                                                T_, C, K, K - 1 - pad_left, 1, (const T*)y, nullptr, BnBwd{})));
  return launch_status();
}

extern "C" int lidk_dwconv_bwd_input_bn_glu(const void* ds, const void* c, const float* mean, const float* rstd,
                                            const float* gamma, const float* beta, const double* sums, double count,
                                            const float* w, const void* y, void* dy, int B, int T_, int C, int K, int pad_left,
                                            int dtype, void* stream) {
  if (!ds || !c || !mean || !rstd || !gamma || !beta || !sums || !w || !y || !dy || B <= 0 || T_ <= 0 || C <= 0 ||
      (C & 3) || K <= 0 || K > DW_KMAX || pad_left < 0 || pad_left >= K)
    return LIDK_ERR_ARG;
  BnBwd bn{c, mean, rstd, gamma, beta, sums, count};
  if (dw2_ok(C, dtype)) {
    DW2_LAUNCH(3, (const bf16*)ds, w, nullptr, (bf16*)dy, nullptr, B, T_, C, K, K - 1 - pad_left, 1, (const bf16*)y, nullptr, bn);
    return launch_status();
  }
  dim3 grid(cdiv(T_, DW_TT), cdiv(C, 64), B);
  LIDK_DISPATCH(dtype, (dwconv_kernel<T, 3><<<grid, 256, 0, as_stream(stream)>>>((const T*)ds, w, nullptr, (T*)dy, nullptr, B,
                                                                                T_, C, K, K - 1 - pad_left, 1, (const T*)y, nullptr, bn)));
  return launch_status();
}

// ------------------------------------------------------------------------------------ depthwise conv wgrad
// partial[b][ch][k] = sum_t dc[b][t][ch] * g[b][t+k-pad_left][ch]  (k < K),  partial[b][ch][K] = sum_t dc[b][t][ch]
template <typename T>
__global__ void __launch_bounds__(256)
dwconv_wgrad_kernel(const T* __restrict__ dc, const T* __restrict__ g, float* __restrict__ partial, int B, int T_, int C,
                    int K, int pad_left) {
  __shared__ __attribute__((aligned(16))) T gt[DW_ROWS][64];
  __shared__ __attribute__((aligned(16))) T dt[DW_TT][64];
  __shared__ float red[4][DW_KMAX + 1][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 64, b = blockIdx.y, ch = c0 + lane;
  const bool chok = ch < C;
  float acc[DW_KMAX + 1];
#pragma unroll
  for (int k = 0; k <= DW_KMAX; ++k) acc[k] = 0.f;
  for (int t0 = 0; t0 < T_; t0 += DW_TT) {
    __syncthreads();
    constexpr int VE = 16 / sizeof(T), CPR = 64 / VE;
    if (c0 + 64 <= C && (C % VE) == 0) {
      for (int q = threadIdx.x; q < DW_ROWS * CPR; q += 256) {
        int r = q / CPR, cc = (q % CPR) * VE, t = t0 - pad_left + r;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t >= 0 && t < T_) v = *reinterpret_cast<const uint4*>(g + ((size_t)b * T_ + t) * C + c0 + cc);
        *reinterpret_cast<uint4*>(&gt[r][cc]) = v;
      }
      for (int q = threadIdx.x; q < DW_TT * CPR; q += 256) {
        int r = q / CPR, cc = (q % CPR) * VE, t = t0 + r;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t < T_) v = *reinterpret_cast<const uint4*>(dc + ((size_t)b * T_ + t) * C + c0 + cc);
        *reinterpret_cast<uint4*>(&dt[r][cc]) = v;
      }
    } else {
      for (int r = wave; r < DW_ROWS; r += 4) {
        int t = t0 - pad_left + r;
        gt[r][lane] = (chok && t >= 0 && t < T_) ? g[((size_t)b * T_ + t) * C + ch] : from_f<T>(0.f);
      }
      for (int r = wave; r < DW_TT; r += 4) {
        int t = t0 + r;
        dt[r][lane] = (chok && t < T_) ? dc[((size_t)b * T_ + t) * C + ch] : from_f<T>(0.f);
      }
    }
    __syncthreads();
    float x[8 + DW_KMAX - 1];
#pragma unroll
    for (int r = 0; r < 8 + DW_KMAX - 1; ++r) x[r] = to_f(gt[wave * 8 + r][lane]);
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float d = to_f(dt[wave * 8 + o][lane]);
#pragma unroll
      for (int k = 0; k < DW_KMAX; ++k) acc[k] = fmaf(d, x[o + k], acc[k]);
      acc[DW_KMAX] += d;
    }
  }
#pragma unroll
  for (int k = 0; k <= DW_KMAX; ++k) red[wave][k][lane] = acc[k];
  __syncthreads();
  if (wave == 0 && chok) {
    float* p = partial + ((size_t)b * C + ch) * (K + 1);
    for (int k = 0; k < K; ++k) p[k] = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
    p[K] = red[0][DW_KMAX][lane] + red[1][DW_KMAX][lane] + red[2][DW_KMAX][lane] + red[3][DW_KMAX][lane];
  }
}

// dw[ch][k] += sum_b partial[b][ch][k], db[ch] += sum_b partial[b][ch][K]: 64 outputs per workgroup, 16 waves split the batch,
// fixed summation order
__global__ void __launch_bounds__(1024)
dwconv_wgrad_finalize_kernel(const float* __restrict__ partial, int B, int C, int K, float* __restrict__ dw,
                             float* __restrict__ db) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane, n = C * (K + 1);
  float acc = 0.f;
  if (idx < n)
    for (int b = wv; b < B; b += 16) acc += partial[(size_t)b * n + idx];
  red[wv][lane] = acc;
  __syncthreads();
  if (wv != 0 || idx >= n) return;
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += red[i][lane];
  const int ch = idx / (K + 1), k = idx - ch * (K + 1);
  if (k < K) dw[(size_t)ch * K + k] += t;
  else if (db) db[ch] += t;
}

extern "C" int lidk_dwconv_bwd_weight(const void* dc, const void* g, float* dw, float* db, float* partial, int B, int T_,
                                      int C, int K, int pad_left, int dtype, void* stream) {
  if (!dc || !g || !dw || !partial || B <= 0 || T_ <= 0 || C <= 0 || K <= 0 || K > DW_KMAX || pad_left < 0 || pad_left >= K) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  dim3 grid(cdiv(C, 64), B);
  LIDK_DISPATCH(dtype, dwconv_wgrad_kernel<T><<<grid, 256, 0, s>>>((const T*)dc, (const T*)g, partial, B, T_, C, K, pad_left));
  dwconv_wgrad_finalize_kernel<<<cdiv(C * (K + 1), 64), 1024, 0, s>>>(partial, B, C, K, dw, db);
  return launch_status();
}

// ------------------------------------------------------------------------------------ depthwise conv wgrad, fused bf16 form
// The depthwise-conv weight gradient with the BatchNorm+Swish backward "apply" step in its operand load: dc (the gradient at
// the conv output) is formed from ds and c while the tile is staged - exactly the values lidk_bn_swish_bwd_apply would have
// written (bf16-rounded) - so dc never exists in HBM (-2 x 9.9 MB per block at cfg2) and one launch disappears.  Workgroup
// (0, 0) also adds this rank's BatchNorm parameter gradients (dbeta += sum dz, dgamma += sum dz*xhat) from sums_local.
// 80-row time tiles, 16-byte accesses; partial[b][ch][k] as in dwconv_wgrad_kernel, finished by dwconv_wgrad_finalize_kernel.
#define DWG_TT 80
__global__ void __launch_bounds__(256)
dwconv_wgrad_bn_bf16_kernel(const bf16* __restrict__ ds, const bf16* __restrict__ g, float* __restrict__ partial, int B, int T_, int C,
                            int K, int pad_left, BnBwd bn, const double* __restrict__ sums_local, float* __restrict__ dgamma,
                            float* __restrict__ dbeta) {
  constexpr int OPW = DWG_TT / 4, GROWS = DWG_TT + DW_KMAX;
  constexpr int G_BYTES = GROWS * 64 * 2, D_BYTES = DWG_TT * 64 * 2, R_BYTES = 4 * (DW_KMAX + 1) * 64 * 4;
  constexpr int SM = (G_BYTES + D_BYTES) > R_BYTES ? (G_BYTES + D_BYTES) : R_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  __shared__ __attribute__((aligned(16))) float cst[6][64];
  bf16 (*gt)[64] = reinterpret_cast<bf16 (*)[64]>(smem);
  bf16 (*dt)[64] = reinterpret_cast<bf16 (*)[64]>(smem + G_BYTES);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 64, b = blockIdx.y, ch = c0 + lane;
  const bool chok = ch < C;
  if (blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = threadIdx.x; i < C; i += 256) {
      if (dbeta) dbeta[i] += (float)sums_local[i];
      if (dgamma) dgamma[i] += (float)sums_local[C + i];
    }
  if (threadIdx.x < 64) {
    float mu = 0.f, rs = 0.f, gm = 0.f, bt = 0.f, m0 = 0.f, m1 = 0.f;
    if (chok) {
      const double cnt = bn.count > 0 ? bn.count : bn.sums[2 * C];
      mu = bn.mean[ch]; rs = bn.rstd[ch]; gm = bn.gamma[ch]; bt = bn.beta[ch];
      m0 = (float)(bn.sums[ch] / cnt); m1 = (float)(bn.sums[C + ch] / cnt);
    }
    cst[0][lane] = mu; cst[1][lane] = rs; cst[2][lane] = gm; cst[3][lane] = bt; cst[4][lane] = m0; cst[5][lane] = m1;
  }
  __syncthreads();
  const int cc = (threadIdx.x & 7) * 8;
  const bool cok = c0 + cc < C;
  f32x8_t mu, rs, gm, bt, m0, m1;
  mu.lo = *reinterpret_cast<const float4*>(&cst[0][cc]); mu.hi = *reinterpret_cast<const float4*>(&cst[0][cc + 4]);
  rs.lo = *reinterpret_cast<const float4*>(&cst[1][cc]); rs.hi = *reinterpret_cast<const float4*>(&cst[1][cc + 4]);
  gm.lo = *reinterpret_cast<const float4*>(&cst[2][cc]); gm.hi = *reinterpret_cast<const float4*>(&cst[2][cc + 4]);
  bt.lo = *reinterpret_cast<const float4*>(&cst[3][cc]); bt.hi = *reinterpret_cast<const float4*>(&cst[3][cc + 4]);
  m0.lo = *reinterpret_cast<const float4*>(&cst[4][cc]); m0.hi = *reinterpret_cast<const float4*>(&cst[4][cc + 4]);
  m1.lo = *reinterpret_cast<const float4*>(&cst[5][cc]); m1.hi = *reinterpret_cast<const float4*>(&cst[5][cc + 4]);
  const bf16* cbuf = reinterpret_cast<const bf16*>(bn.c);
  float acc[DW_KMAX + 1];
#pragma unroll
  for (int k = 0; k <= DW_KMAX; ++k) acc[k] = 0.f;
  for (int t0 = 0; t0 < T_; t0 += DWG_TT) {
    __syncthreads();
    for (int r = threadIdx.x >> 3; r < GROWS; r += 32) {
      const int t = t0 - pad_left + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (cok && t >= 0 && t < T_) v = *reinterpret_cast<const uint4*>(g + ((size_t)b * T_ + t) * C + c0 + cc);
      *reinterpret_cast<uint4*>(&gt[r][cc]) = v;
    }
    for (int r = threadIdx.x >> 3; r < DWG_TT; r += 32) {
      const int t = t0 + r;
      f32x8_t o;
      o.lo = make_float4(0.f, 0.f, 0.f, 0.f); o.hi = o.lo;
      if (cok && t < T_) {
        const size_t e = ((size_t)b * T_ + t) * C + c0 + cc;
        const f32x8_t x = load8(cbuf + e), d = load8(ds + e);
        float xh, dz;
#define DWG_BN(F, H)                                                                                                     \
        xh = (x.H.F - mu.H.F) * rs.H.F; dz = d.H.F * swish_grad(xh * gm.H.F + bt.H.F);                                   \
        o.H.F = gm.H.F * rs.H.F * (dz - m0.H.F - xh * m1.H.F);
        DWG_BN(x, lo) DWG_BN(y, lo) DWG_BN(z, lo) DWG_BN(w, lo) DWG_BN(x, hi) DWG_BN(y, hi) DWG_BN(z, hi) DWG_BN(w, hi)
#undef DWG_BN
      }
      store8(&dt[r][cc], o);
    }
    __syncthreads();
    float x[OPW + DW_KMAX - 1];
#pragma unroll
    for (int r = 0; r < OPW + DW_KMAX - 1; ++r) x[r] = to_f(gt[wave * OPW + r][lane]);
#pragma unroll
    for (int o = 0; o < OPW; ++o) {
      const float d = to_f(dt[wave * OPW + o][lane]);
#pragma unroll
      for (int k = 0; k < DW_KMAX; ++k) acc[k] = fmaf(d, x[o + k], acc[k]);
      acc[DW_KMAX] += d;
    }
  }
  __syncthreads();
  float (*red)[DW_KMAX + 1][64] = reinterpret_cast<float (*)[DW_KMAX + 1][64]>(smem);
#pragma unroll
  for (int k = 0; k <= DW_KMAX; ++k) red[wave][k][lane] = acc[k];
  __syncthreads();
  if (wave == 0 && chok) {
    float* p = partial + ((size_t)b * C + ch) * (K + 1);
    for (int k = 0; k < K; ++k) p[k] = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
    p[K] = red[0][DW_KMAX][lane] + red[1][DW_KMAX][lane] + red[2][DW_KMAX][lane] + red[3][DW_KMAX][lane];
  }
}
extern "C" int lidk_dwconv_bwd_weight_bn_supported(int C, int dtype) { return dw2_ok(C, dtype) ? 1 : 0; }
extern "C" int lidk_dwconv_bwd_weight_bn(const void* ds, const void* c, const float* mean, const float* rstd, const float* gamma,
                                         const float* beta, const double* sums, const double* sums_local, double count,
                                         const void* g, float* dw, float* db, float* dgamma, float* dbeta, float* partial, int B,
                                         int T_, int C, int K, int pad_left, int dtype, void* stream) {
  if (!ds || !c || !mean || !rstd || !gamma || !beta || !sums || !sums_local || !g || !dw || !partial || B <= 0 || T_ <= 0 ||
      C <= 0 || K <= 0 || K > DW_KMAX || pad_left < 0 || pad_left >= K || !dw2_ok(C, dtype))
    return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  BnBwd bn{c, mean, rstd, gamma, beta, sums, count};
  dim3 grid(cdiv(C, 64), B);
  dwconv_wgrad_bn_bf16_kernel<<<grid, 256, 0, s>>>((const bf16*)ds, (const bf16*)g, partial, B, T_, C, K, pad_left, bn, sums_local,
                                                  dgamma, dbeta);
  dwconv_wgrad_finalize_kernel<<<cdiv(C * (K + 1), 64), 1024, 0, s>>>(partial, B, C, K, dw, db);
  return launch_status();
}

// ------------------------------------------------------------------------------------ BatchNorm statistics
__global__ void bn_train_stats_kernel(const double* __restrict__ sums, double count_arg, float* __restrict__ mean,
                                      float* __restrict__ rstd, float* __restrict__ rmean, float* __restrict__ rvar,
                                      int64_t* __restrict__ nbt, float momentum, float eps, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  const double count = count_arg > 0 ? count_arg : sums[2 * C];       // count <= 0: the global row count rides behind the sums
  double mu = sums[c] / count;
  double var = sums[C + c] / count - mu * mu;
  if (var < 0) var = 0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * (count / (count > 1 ? count - 1 : 1)));
}
// Single-process form: column-reduce the depthwise conv's partial rows AND finish the statistics in one launch (under data
// parallelism the all-reduce sits between the two, so lidk_reduce_partials_f64 + lidk_bn_train_stats stay separate there).
__global__ void __launch_bounds__(1024)
bn_train_stats_from_partials_kernel(const float* __restrict__ partial, int nparts, double count,
                                    float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ rmean,
                                    float* __restrict__ rvar, int64_t* __restrict__ nbt, float momentum, float eps,
                                    int C) {
  // 64 channels x 16 part-lanes per workgroup: each lane strides over the partial rows (independent coalesced loads), the 16
  // lane sums are folded in fixed order through LDS -> deterministic.
  __shared__ double red[2][16][64];
  const int lane = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  if (c == 0 && pl == 0 && nbt) *nbt += 1;
  double s = 0.0, ss = 0.0;
  if (c < C)
    for (int p = pl; p < nparts; p += 16) {
      s += (double)partial[(size_t)p * 2 * C + c];
      ss += (double)partial[(size_t)p * 2 * C + C + c];
    }
  red[0][pl][lane] = s;
  red[1][pl][lane] = ss;
  __syncthreads();
  if (pl != 0 || c >= C) return;
  s = 0.0; ss = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s += red[0][i][lane]; ss += red[1][i][lane]; }
  const double mu = s / count;
  double var = ss / count - mu * mu;
  if (var < 0) var = 0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * (count / (count > 1 ? count - 1 : 1)));
}
extern "C" int lidk_bn_train_stats_from_partials(const float* partial, int nparts, double count, float* mean, float* rstd,
                                                 float* running_mean, float* running_var, int64_t* nbt, float momentum,
                                                 float eps, int C, void* stream) {
  if (!partial || !mean || !rstd || nparts <= 0 || count <= 0 || C <= 0) return LIDK_ERR_ARG;
  bn_train_stats_from_partials_kernel<<<cdiv(C, 64), 1024, 0, as_stream(stream)>>>(partial, nparts, count, mean, rstd, running_mean,
                                                                                running_var, nbt, momentum, eps, C);
  return launch_status();
}

extern "C" int lidk_bn_train_stats(const double* sums, double count, float* mean, float* rstd, float* running_mean,
                                   float* running_var, int64_t* nbt, float momentum, float eps, int C, void* stream) {
  if (!sums || !mean || !rstd || C <= 0) return LIDK_ERR_ARG;
  bn_train_stats_kernel<<<cdiv(C, 256), 256, 0, as_stream(stream)>>>(sums, count, mean, rstd, running_mean, running_var,
                                                                     nbt, momentum, eps, C);
  return launch_status();
}
__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                     float* __restrict__ mean, float* __restrict__ rstd, float eps, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rmean[c];
  rstd[c] = 1.0f / sqrtf(rvar[c] + eps);
}
extern "C" int lidk_bn_eval_stats(const float* running_mean, const float* running_var, float* mean, float* rstd, float eps,
                                  int C, void* stream) {
  if (!running_mean || !running_var || !mean || !rstd || C <= 0) return LIDK_ERR_ARG;
  bn_eval_stats_kernel<<<cdiv(C, 256), 256, 0, as_stream(stream)>>>(running_mean, running_var, mean, rstd, eps, C);
  return launch_status();
}

// ------------------------------------------------------------------------------------ BatchNorm + Swish forward
template <typename T>
__global__ void bn_swish_fwd_kernel(const T* __restrict__ c, const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ s,
                                    long M, int C) {
  long n4 = M * C / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    long e = i * 4; int ch = (int)(e % C);
    float4 x = load4(c + e), mu = load4(mean + ch), rs = load4(rstd + ch), g = load4(gamma + ch), b = load4(beta + ch), o;
    float z;
    z = (x.x - mu.x) * rs.x * g.x + b.x; o.x = z * sigmoidf_(z);
    z = (x.y - mu.y) * rs.y * g.y + b.y; o.y = z * sigmoidf_(z);
    z = (x.z - mu.z) * rs.z * g.z + b.z; o.z = z * sigmoidf_(z);
    z = (x.w - mu.w) * rs.w * g.w + b.w; o.w = z * sigmoidf_(z);
    store4(s + e, o);
  }
}
extern "C" int lidk_bn_swish_fwd(const void* c, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 void* s, int M, int C, int dtype, void* stream) {
  if (!c || !mean || !rstd || !gamma || !beta || !s || M <= 0 || C <= 0 || (C & 3)) return LIDK_ERR_ARG;
  LIDK_DISPATCH(dtype, bn_swish_fwd_kernel<T><<<ew_blocks((long)M * C / 4), 256, 0, as_stream(stream)>>>(
                           (const T*)c, mean, rstd, gamma, beta, (T*)s, M, C));
  return launch_status();
}

// ------------------------------------------------------------------------------------ BatchNorm + Swish backward

// Column reduction over M rows of (dz, dz*xhat): a thread owns 4 consecutive channels (8/16-byte loads) and every
// (256 / (Cs/4))-th row of its workgroup's row set; row-lanes are combined through LDS.  One launch covers a slab of Cs <= 1024
// channels starting at c0 (wider layers - the d = 768 heads have 1,536 conv channels - take one launch per slab).
template <typename T>
__global__ void __launch_bounds__(256)
bn_swish_bwd_reduce_kernel(const T* __restrict__ ds, const T* __restrict__ c, const float* __restrict__ mean,
                           const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                           float* __restrict__ partial, int M, int C, int c0, int Cs) {
  __shared__ float red[256][8];
  const int groups = Cs / 4, rl_n = 256 / groups;         // channel groups, row lanes
  const int cg = threadIdx.x % groups, rl = threadIdx.x / groups;
  float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
  if (rl < rl_n) {
    const int ch = c0 + cg * 4;
    const float4 mu = load4(mean + ch), rs = load4(rstd + ch), g = load4(gamma + ch), b = load4(beta + ch);
    auto row = [&](float4 x, float4 d) __attribute__((always_inline)) {
      float xh, dz;
      xh = (x.x - mu.x) * rs.x; dz = d.x * swish_grad(xh * g.x + b.x); a0[0] += dz; a1[0] = fmaf(dz, xh, a1[0]);
      xh = (x.y - mu.y) * rs.y; dz = d.y * swish_grad(xh * g.y + b.y); a0[1] += dz; a1[1] = fmaf(dz, xh, a1[1]);
      xh = (x.z - mu.z) * rs.z; dz = d.z * swish_grad(xh * g.z + b.z); a0[2] += dz; a1[2] = fmaf(dz, xh, a1[2]);
      xh = (x.w - mu.w) * rs.w; dz = d.w * swish_grad(xh * g.w + b.w); a0[3] += dz; a1[3] = fmaf(dz, xh, a1[3]);
    };
    const int step = gridDim.x * rl_n;
    int m = blockIdx.x * rl_n + rl;
    for (; m + step < M; m += 2 * step) {               // two rows per iteration: four independent loads in flight
      float4 x0 = load4(c + (size_t)m * C + ch), d0 = load4(ds + (size_t)m * C + ch);
      float4 x1 = load4(c + (size_t)(m + step) * C + ch), d1 = load4(ds + (size_t)(m + step) * C + ch);
      row(x0, d0); row(x1, d1);
    }
    if (m < M) row(load4(c + (size_t)m * C + ch), load4(ds + (size_t)m * C + ch));
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { red[threadIdx.x][q] = a0[q]; red[threadIdx.x][4 + q] = a1[q]; }
  __syncthreads();
  if (threadIdx.x < groups) {
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rl_n; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) { s0[q] += red[r * groups + cg][q]; s1[q] += red[r * groups + cg][4 + q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      partial[((size_t)blockIdx.x * 2 + 0) * C + c0 + cg * 4 + q] = s0[q];
      partial[((size_t)blockIdx.x * 2 + 1) * C + c0 + cg * 4 + q] = s1[q];
    }
  }
}
extern "C" int lidk_bn_swish_bwd_reduce(const void* ds, const void* c, const float* mean, const float* rstd,
                                        const float* gamma, const float* beta, float* partial, int M, int C, int dtype,
                                        void* stream) {
  if (!ds || !c || !mean || !rstd || !gamma || !beta || !partial || M <= 0 || C <= 0 || (C & 3)) return LIDK_ERR_ARG;
  // always LIDK_BN_PARTIAL_BLOCKS partial rows; blocks beyond M write zeros.  Channel slabs of <= 1024 (a multiple of 4 each).
  const int nslab = cdiv(C, 1024), per = cdiv(cdiv(C, nslab), 4) * 4;
  for (int c0 = 0; c0 < C; c0 += per) {
    const int Cs = C - c0 < per ? C - c0 : per;
    LIDK_DISPATCH(dtype, bn_swish_bwd_reduce_kernel<T><<<LIDK_BN_PARTIAL_BLOCKS, 256, 0, as_stream(stream)>>>(
                             (const T*)ds, (const T*)c, mean, rstd, gamma, beta, partial, M, C, c0, Cs));
  }
  return launch_status();
}

template <typename T>
__global__ void bn_swish_bwd_apply_kernel(const T* __restrict__ ds, const T* __restrict__ c, const float* __restrict__ mean,
                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, const double* __restrict__ sums, double count_arg,
                                          T* __restrict__ dc, long M, int C, const double* __restrict__ sums_local,
                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const double count = count_arg > 0 ? count_arg : sums[2 * C];       // count <= 0: the global row count rides behind the sums
  if (blockIdx.x == 0) {        // parameter gradients from this rank's own sums (DDP averages them like any other gradient)
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
      if (dbeta) dbeta[ch] += (float)sums_local[ch];
      if (dgamma) dgamma[ch] += (float)sums_local[C + ch];
    }
  }
  // 4 channels per thread (8/16-byte accesses).  When C/4 divides the block size a thread keeps the same channels for
  // its whole grid-stride walk, so the per-channel constants (two f64 divisions each) are computed once.
  const int groups = C / 4;
  const long n4 = M * (long)groups, stride = (long)gridDim.x * blockDim.x;
  const bool fixed = (blockDim.x % groups) == 0;
  float4 mu, rs, g, b, m0, m1;
  auto consts = [&](int ch) __attribute__((always_inline)) {
    mu = load4(mean + ch); rs = load4(rstd + ch); g = load4(gamma + ch); b = load4(beta + ch);
    m0 = make_float4((float)(sums[ch] / count), (float)(sums[ch + 1] / count), (float)(sums[ch + 2] / count),
                     (float)(sums[ch + 3] / count));
    m1 = make_float4((float)(sums[C + ch] / count), (float)(sums[C + ch + 1] / count), (float)(sums[C + ch + 2] / count),
                     (float)(sums[C + ch + 3] / count));
  };
  if (fixed) consts((int)(threadIdx.x % groups) * 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if (!fixed) consts((int)(i % groups) * 4);
    const long e = i * 4;
    float4 x = load4(c + e), d = load4(ds + e), o;
    float xh, dz;
    xh = (x.x - mu.x) * rs.x; dz = d.x * swish_grad(xh * g.x + b.x); o.x = g.x * rs.x * (dz - m0.x - xh * m1.x);
    xh = (x.y - mu.y) * rs.y; dz = d.y * swish_grad(xh * g.y + b.y); o.y = g.y * rs.y * (dz - m0.y - xh * m1.y);
    xh = (x.z - mu.z) * rs.z; dz = d.z * swish_grad(xh * g.z + b.z); o.z = g.z * rs.z * (dz - m0.z - xh * m1.z);
    xh = (x.w - mu.w) * rs.w; dz = d.w * swish_grad(xh * g.w + b.w); o.w = g.w * rs.w * (dz - m0.w - xh * m1.w);
    store4(dc + e, o);
  }
}
extern "C" int lidk_bn_swish_bwd_apply(const void* ds, const void* c, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, const double* sums,
                                       const double* sums_local, double count, void* dc, float* dgamma, float* dbeta,
                                       int M, int C, int dtype, void* stream) {
  if (!ds || !c || !mean || !rstd || !gamma || !beta || !sums || !sums_local || !dc || M <= 0 || C <= 0 || (C & 3)) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  LIDK_DISPATCH(dtype, bn_swish_bwd_apply_kernel<T><<<ew_blocks((long)M * C / 4), 256, 0, s>>>(
                           (const T*)ds, (const T*)c, mean, rstd, gamma, beta, sums, count, (T*)dc, M, C, sums_local, dgamma,
                           dbeta));
  return launch_status();
}
