// Shared device helpers for the lidk kernels (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lidk.h"

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LIDK_WAVE 64

__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float x) { return (bf16)x; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: the GEMM epilogues were VALU-issue bound on it
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// erf to 1.5e-7 ABSOLUTE (Abramowitz & Stegun 7.1.26; v_rcp_f32 adds 1 ulp of t): one rcp, one exp and seven FMAs instead of
// the ~40 instructions of erff.  GELU only ever uses 1 + erf, where that is a relative 1.5e-7 - a tenth of an f32 ulp's worth of
// the bf16 values these epilogues store.  (With erff the GELU epilogue of the XLS-R fc1 GEMM was VALU-issue bound: 8 192
// activations per lane and tile.)
__device__ __forceinline__ float erf_(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  return copysignf(1.0f - p * t * __expf(-ax * ax), x);
}
// GELU, erf form (torch.nn.functional.gelu default; lid/wavlm/modules.py gelu)
__device__ __forceinline__ float gelu_(float x) { return 0.5f * x * (1.0f + erf_(x * 0.70710678118654752f)); }
// d/dx gelu(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_grad_(float x) {
  return 0.5f * (1.0f + erf_(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// 4-wide load/store of activations held as T (float or bf16); p must be 4-element aligned.
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16* p) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  float4 v;
  v.x = __uint_as_float(r.x << 16); v.y = __uint_as_float(r.x & 0xffff0000u);
  v.z = __uint_as_float(r.y << 16); v.w = __uint_as_float(r.y & 0xffff0000u);
  return v;
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16* p, float4 v) {
  union { bf16 h[4]; uint2 u; } t;
  t.h[0] = (bf16)v.x; t.h[1] = (bf16)v.y; t.h[2] = (bf16)v.z; t.h[3] = (bf16)v.w;
  *reinterpret_cast<uint2*>(p) = t.u;
}

// 8-wide (16-byte) load/store of bf16 activations; p must be 8-element aligned.
struct f32x8_t { float4 lo, hi; };
__device__ __forceinline__ f32x8_t unpack8(uint4 r) {
  f32x8_t v;
  v.lo.x = __uint_as_float(r.x << 16); v.lo.y = __uint_as_float(r.x & 0xffff0000u);
  v.lo.z = __uint_as_float(r.y << 16); v.lo.w = __uint_as_float(r.y & 0xffff0000u);
  v.hi.x = __uint_as_float(r.z << 16); v.hi.y = __uint_as_float(r.z & 0xffff0000u);
  v.hi.z = __uint_as_float(r.w << 16); v.hi.w = __uint_as_float(r.w & 0xffff0000u);
  return v;
}
__device__ __forceinline__ f32x8_t load8(const bf16* p) { return unpack8(*reinterpret_cast<const uint4*>(p)); }
__device__ __forceinline__ void store8(bf16* p, const f32x8_t& v) {
  union { bf16 h[8]; uint4 u; } t;
  t.h[0] = (bf16)v.lo.x; t.h[1] = (bf16)v.lo.y; t.h[2] = (bf16)v.lo.z; t.h[3] = (bf16)v.lo.w;
  t.h[4] = (bf16)v.hi.x; t.h[5] = (bf16)v.hi.y; t.h[6] = (bf16)v.hi.z; t.h[7] = (bf16)v.hi.w;
  *reinterpret_cast<uint4*>(p) = t.u;
}

// Fragment of X^T for v_mfma_f32_16x16x32_bf16 from a ROW-MAJOR LDS tile X[k][n] (leading dimension ld elements, ld*2 % 8 == 0):
// lane (fq = lane>>4, fr = lane&15) receives X[k0 + 8*fq + jj][n0 + fr], jj = 0..7, through two ds_read_b64_tr_b16.  EXEC must be
// all ones at the call (the transposed read gathers across lanes); lane mapping pinned by lidk_selftest_tr16.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tr_frag(const bf16* X, int ld, int k0, int n0, int fq, int fr) {
  const bf16* p0 = X + (size_t)(k0 + 8 * fq + (fr >> 2)) * ld + n0 + 4 * (fr & 3);
  union { struct { s16x4_t lo, hi; } h; bf16x8 v; } u;
  u.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)p0);
  u.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p0 + 4 * ld));
  return u.v;
}

// Counter-based generator: splitmix64 of (seed, element index) -> 24-bit uniform.  Stateless, so the same (seed, index) gives
// the same decision on every rank and in the backward pass (lidk_dropout, attention dropout of lidk_xattn_*).
__device__ __forceinline__ float uniform_from(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// The cheap form for per-element decisions inside MFMA kernels (attention dropout of lidk_xattn_*): three 32-bit multiplies with
// xor-shifts ("lowbias32", the fused feature path's dither generator) of (seed, element index) -> 24-bit uniform.  splitmix64 costs
// ~60 VALU issue slots per element on CDNA (its 64-bit multiplies are built from quarter-rate 32-bit ones) - more than the two MFMAs
// an attention score takes part in.  Index bits above 32 are folded into the seed's high word.
__device__ __forceinline__ float uniform32_from(uint64_t seed, uint64_t idx) {
  uint32_t h = (uint32_t)idx * 0x9E3779B1u + (uint32_t)seed;
  h ^= h >> 16; h *= 0x21F0AAADu;
  h ^= (uint32_t)(seed >> 32) + (uint32_t)(idx >> 32) * 0x85EBCA6Bu;
  h ^= h >> 15; h *= 0x735A2D97u;
  h ^= h >> 15;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// Like tr_frag, for an MFMA whose 32 contraction slots are PERMUTED: slot (fq, jj) stands for row k0 + 4*fq + jj (jj < 4) and
// k0 + 16 + 4*fq + (jj - 4) (jj >= 4).  That is the order in which a lane holds two stacked 16x16 accumulator tiles
// (rows 4*fq + r of the first, 16 + 4*fq + r of the second), so those accumulators - packed to bf16 - ARE the other operand:
// P^T.dO, dS^T.Q, P.V and dS.K need no transposition through LDS.
__device__ __forceinline__ bf16x8 tr_frag_split(const bf16* X, int ld, int k0, int n0, int fq, int fr) {
  const bf16* p0 = X + (size_t)(k0 + 4 * fq + (fr >> 2)) * ld + n0 + 4 * (fr & 3);
  union { struct { s16x4_t lo, hi; } h; bf16x8 v; } u;
  u.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)p0);
  u.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p0 + 16 * ld));
  return u.v;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int launch_status() { return hipGetLastError() == hipSuccess ? LIDK_OK : LIDK_ERR_LAUNCH; }

#define LIDK_DISPATCH(dtype, ...)                         \
  do {                                                    \
    if ((dtype) == LIDK_F32) { using T = float; __VA_ARGS__; } \
    else if ((dtype) == LIDK_BF16) { using T = bf16; __VA_ARGS__; } \
    else return LIDK_ERR_ARG;                             \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
