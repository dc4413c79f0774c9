// Feature path: normalize_wav, dither + pre-emphasis, STFT -> |X|^2 -> mel -> dB -> top_db floor -> SpecAugment masks
// (lid/audio_processor.py:72-134,198-228) and the Conv1dSubSampling2 im2col (lid/conformer.py:328-348).
// The STFT is a 512-point radix-2 FFT per frame held entirely in LDS by ONE wave (4 butterflies per lane per
// stage, 9 stages); power, mel projection and log are fused behind it so a frame's samples are read once and
// only its 80 mel values are written (algorithmic bytes: 4*L in + 4*80*F out per utterance).
#include "common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------ normalize_wav
__global__ void __launch_bounds__(1024) normalize_wav_kernel(const float* __restrict__ wav, float* __restrict__ out, int Lrow,
                                                             const int32_t* __restrict__ n_samples) {
  __shared__ float red[16];
  __shared__ float bc;
  const float* x = wav + (size_t)blockIdx.x * Lrow;
  float* y = out + (size_t)blockIdx.x * Lrow;
  // ragged batch: statistics over the utterance's own samples, zeros behind them
  const int L = n_samples ? max(2, min(Lrow, n_samples[blockIdx.x])) : Lrow;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) s += x[i];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < 16; ++w) t += red[w]; bc = t / (float)L; }
  __syncthreads();
  const float mu = bc;
  float q = 0.f;
  for (int i = threadIdx.x; i < L; i += blockDim.x) { float d = x[i] - mu; q = fmaf(d, d, q); }
  q = wave_sum(q);
  __syncthreads();
  if (lane == 0) red[wave] = q;
  __syncthreads();
  if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < 16; ++w) t += red[w]; bc = 1.0f / (sqrtf(t / (float)(L - 1)) + 1e-6f); }
  __syncthreads();
  const float inv = bc;
  for (int i = threadIdx.x; i < L; i += blockDim.x) y[i] = (x[i] - mu) * inv;
  for (int i = L + threadIdx.x; i < Lrow; i += blockDim.x) y[i] = 0.f;
}

extern "C" int lidk_normalize_wav(const float* wav, float* out, int B, int L, const int32_t* n_samples, void* stream) {
  if (!wav || !out || B <= 0 || L < 2) return LIDK_ERR_ARG;
  normalize_wav_kernel<<<B, 1024, 0, as_stream(stream)>>>(wav, out, L, n_samples);
  return launch_status();
}

// ------------------------------------------------------------------------------------ dither + pre-emphasis
// uniform_from(seed, index): common.h

__global__ void dither_preemph_kernel(const float* __restrict__ wav, float* __restrict__ out, const float* __restrict__ noise,
                                      int L, long n, float coef, float dither, uint64_t seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int t = (int)(i % L);
    float u1 = dither != 0.f ? (noise ? noise[i] : uniform_from(seed, (uint64_t)i)) : 0.f;
    float cur = wav[i] + dither * u1;
    if (t > 0) {
      float u0 = dither != 0.f ? (noise ? noise[i - 1] : uniform_from(seed, (uint64_t)(i - 1))) : 0.f;
      cur -= coef * (wav[i - 1] + dither * u0);
    }
    out[i] = cur;
  }
}

extern "C" int lidk_dither_preemph(const float* wav, float* out, const float* noise, int B, int L, float coef, float dither,
                                   uint64_t seed, void* stream) {
  if (!wav || !out || wav == out || B <= 0 || L <= 0) return LIDK_ERR_ARG;
  long n = (long)B * L;
  int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
  dither_preemph_kernel<<<blocks, 256, 0, as_stream(stream)>>>(wav, out, noise, L, n, coef, dither, seed);
  return launch_status();
}

// ------------------------------------------------------------------------------------ speed perturbation (polyphase resampling)
// lid/audio_processor.py:136-156: sox effects ["speed", v], ["rate", sr] with v in {0.9, 1.0, 1.1} = band-limited resampling
// y[n] = x(n * v), output length round(L / v).  v = up/down... written as a rational p/q (11/10, 9/10): output sample n reads
// the input around n*p/q, i.e. integer base (n*p) / q and fractional phase (n*p) % q - one of q fixed FIR rows of a
// host-built windowed-sinc table (cut-off min(1, q/p) of Nyquist, Kaiser window).  A streaming kernel: every input sample is
// read from HBM once (the taps overlap in L1/L2), every output written once.  Utterance b uses row table_of[b] of a small
// set of tables (one per distinct factor in the batch); factor 1 is an exact copy (single unit tap).
struct ResampleTable { const float* taps; int p, q, ntaps, left; };   // taps [q][ntaps]; tap j of phase r weighs x[base - left + j]

__global__ void __launch_bounds__(256)
speed_perturb_kernel(const float* __restrict__ x, int Lin, const int32_t* __restrict__ n_in, float* __restrict__ y, int Lout,
                     const int32_t* __restrict__ n_out, const ResampleTable* __restrict__ tables,
                     const int32_t* __restrict__ table_of) {
  const int b = blockIdx.y;
  const ResampleTable tb = tables[table_of[b]];
  const int nin = n_in ? min(n_in[b], Lin) : Lin, nout = min(n_out[b], Lout);
  const float* xr = x + (size_t)b * Lin;
  float* yr = y + (size_t)b * Lout;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < Lout; n += gridDim.x * blockDim.x) {
    float acc = 0.f;
    if (n < nout) {
      const long pos = (long)n * tb.p;
      const int base = (int)(pos / tb.q) - tb.left, ph = (int)(pos % tb.q);
      const float* h = tb.taps + (size_t)ph * tb.ntaps;
      for (int j = 0; j < tb.ntaps; ++j) {
        const int k = base + j;
        if (k >= 0 && k < nin) acc = fmaf(h[j], xr[k], acc);
      }
    }
    yr[n] = acc;                                       // zeros behind the utterance's own samples (ragged batch layout)
  }
}

extern "C" int lidk_speed_perturb(const float* x, int B, int Lin, const int32_t* n_in, float* y, int Lout, const int32_t* n_out,
                                  const void* tables, int n_tables, const int32_t* table_of, void* stream) {
  if (!x || !y || x == y || !n_out || !tables || !table_of || B <= 0 || Lin <= 0 || Lout <= 0 || n_tables <= 0) return LIDK_ERR_ARG;
  dim3 grid(cdiv(Lout, 256 * 4) > 0 ? cdiv(Lout, 256 * 4) : 1, B);
  speed_perturb_kernel<<<grid, 256, 0, as_stream(stream)>>>(x, Lin, n_in, y, Lout, n_out, (const ResampleTable*)tables, table_of);
  return launch_status();
}

// ------------------------------------------------------------------------------------ log-mel
__global__ void fill_kernel(float* p, int n, float v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

#define LIDK_MEL_MAX 128
#define LIDK_MEL_TAPS 64
// One wave per frame, 16 waves per workgroup (the per-workgroup set-up - window, twiddles, the compact filterbank - is paid
// once per 16 waves); every wave walks a CONTIGUOUS run of frames, so it crosses an utterance boundary at most once or twice
// and the utterance maximum costs one atomic per run (a per-frame atomic on 64 addresses was 40 % of the kernel).
// One wave per frame.  LDS per wave: re[512], im[512]; shared twiddle[256][2], window[512].
#define STFT_WAVES 16
#define WAV_STAT_PARTS 8
// Raw-waveform form (lidk_wav2mel): the frame load applies normalize_wav ((x - mean) / (std + 1e-6), unbiased std, from the
// per-utterance partial sums of wav_stats_kernel), the dither and the pre-emphasis (lid/audio_processor.py:108-134) on the fly:
//   z(s) = (x[s] - mu) * inv + dither * u(b, s) ;  y(s) = z(s) - coef * z(s - 1)  (s > 0),
// so the normalised and the pre-emphasised waveforms never exist in HBM.  stats == NULL: wav is consumed as it is.
struct WavPrep { const double* stats; const float* noise; float coef, dither; unsigned long long seed; };

__global__ void __launch_bounds__(256)
wav_stats_kernel(const float* __restrict__ wav, int L, const int32_t* __restrict__ n_samples, double* __restrict__ stats,
                 float* __restrict__ utt_max, int* __restrict__ tickets) {
  __shared__ double red[4][2];
  const int b = blockIdx.x, part = blockIdx.y;
  const int Lb = n_samples ? max(2, min(L, n_samples[b])) : L;
  const int per = ((Lb + WAV_STAT_PARTS - 1) / WAV_STAT_PARTS + 3) / 4 * 4, lo = part * per, hi = min(Lb, lo + per);
  const float* x = wav + (size_t)b * L;
  // per-thread f32 partial sums over at most a few dozen samples (all loads independent: 16 bytes each where the row allows),
  // promoted to f64 before any two partial sums meet
  float s = 0.f, q = 0.f;
  if ((((size_t)b * L) & 3) == 0) {
    for (int i = lo + 4 * threadIdx.x; i < hi; i += 4 * 256) {
      if (i + 3 < hi) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        s += (v.x + v.y) + (v.z + v.w);
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      } else {
        for (int e = i; e < hi; ++e) { s += x[e]; q += x[e] * x[e]; }
      }
    }
  } else {
    for (int i = lo + threadIdx.x; i < hi; i += 256) { const float v = x[i]; s += v; q += v * v; }
  }
  double sd = (double)s, qd = (double)q;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sd += __shfl_xor(sd, o, 64); qd += __shfl_xor(qd, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sd; red[threadIdx.x >> 6][1] = qd; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[((size_t)b * WAV_STAT_PARTS + part) * 2 + 0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    stats[((size_t)b * WAV_STAT_PARTS + part) * 2 + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    if (part == 0) { utt_max[b] = -INFINITY; if (tickets) tickets[b] = 0; }
  }
}

__global__ void __launch_bounds__(64 * STFT_WAVES)
stft_mel_kernel(const float* __restrict__ wav, const float* __restrict__ window, const float* __restrict__ twiddle,
                const float* __restrict__ melfb, float* __restrict__ out, float* __restrict__ utt_max, int B, int L,
                int pad, int hop, int F, int n_mels, int dbg, const int32_t* __restrict__ n_samples, WavPrep prep) {
  __shared__ float s_re[STFT_WAVES][LIDK_N_FFT];
  __shared__ float s_im[STFT_WAVES][LIDK_N_FFT];
  __shared__ float s_tw[LIDK_N_FFT / 2][2];
  __shared__ float s_win[LIDK_N_FFT];
  __shared__ int s_lo[LIDK_MEL_MAX], s_hi[LIDK_MEL_MAX];         // nonzero k-range of every (triangular) mel filter
  __shared__ float s_coef[LIDK_MEL_MAX * LIDK_MEL_TAPS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // range scan, one filterbank row per thread (all loads independent; a per-filter serial scan costs 257 dependent L2
  // round trips per workgroup)
  for (int m = threadIdx.x; m < LIDK_MEL_MAX; m += 64 * STFT_WAVES) { s_lo[m] = LIDK_N_FFT / 2 + 1; s_hi[m] = -1; }
  __syncthreads();
  for (int k = threadIdx.x; k <= LIDK_N_FFT / 2; k += 64 * STFT_WAVES) {
    const float* row = melfb + (size_t)k * n_mels;
    const int nm = min(n_mels, LIDK_MEL_MAX);
    for (int m0 = 0; m0 < nm; m0 += 16) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = row[min(m0 + j, nm - 1)];        // 16 loads in flight
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (m0 + j < nm && v[j] != 0.f) { atomicMin(&s_lo[m0 + j], k); atomicMax(&s_hi[m0 + j], k); }
    }
  }
  for (int i = threadIdx.x; i < LIDK_N_FFT; i += 64 * STFT_WAVES) s_win[i] = window[i];
  for (int i = threadIdx.x; i < LIDK_N_FFT / 2; i += 64 * STFT_WAVES) { s_tw[i][0] = twiddle[2 * i]; s_tw[i][1] = twiddle[2 * i + 1]; }
  __syncthreads();
  // the nonzero taps of every filter (about two per FFT bin in total) -> LDS, LIDK_MEL_TAPS slots per filter; the per-frame
  // projection then never touches global memory (it was a chain of dependent L2 loads, 30 deep for the widest filter)
  for (int i = threadIdx.x; i < LIDK_MEL_MAX * LIDK_MEL_TAPS; i += 64 * STFT_WAVES) {
    const int m = i / LIDK_MEL_TAPS, j = i - m * LIDK_MEL_TAPS;
    float c = 0.f;
    if (m < n_mels && s_lo[m] + j <= s_hi[m]) c = melfb[(size_t)(s_lo[m] + j) * n_mels + m];
    s_coef[i] = c;
  }
  __syncthreads();
  float* re = s_re[wave];
  float* im = s_im[wave];
  const long nframes = (long)B * F;
  const long per_wave = (nframes + (long)gridDim.x * STFT_WAVES - 1) / ((long)gridDim.x * STFT_WAVES);
  const long fr0 = ((long)blockIdx.x * STFT_WAVES + wave) * per_wave, fr1 = min(nframes, fr0 + per_wave);
  int run_b = -1;
  float run_max = -INFINITY, mu = 0.f, inv = 1.f;
  for (long fr = fr0; fr < fr1; ++fr) {
    const int b = (int)(fr / F), f = (int)(fr - (long)b * F);
    if (b != run_b) {                                    // wave-uniform
      if (run_b >= 0 && lane == 0 && !(dbg & 4)) atomic_max_float(&utt_max[run_b], run_max);
      run_b = b; run_max = -INFINITY;
      if (prep.stats) {                                  // this utterance's mean and 1 / (std + 1e-6)
        const int Ln = n_samples ? max(2, min(L, n_samples[b])) : L;
        double sm = 0.0, sq = 0.0;
        for (int p = 0; p < WAV_STAT_PARTS; ++p) { sm += prep.stats[((size_t)b * WAV_STAT_PARTS + p) * 2]; sq += prep.stats[((size_t)b * WAV_STAT_PARTS + p) * 2 + 1]; }
        const double mean = sm / (double)Ln;
        double var = (sq - sm * mean) / (double)(Ln - 1);
        if (var < 0.0) var = 0.0;
        mu = (float)mean;
        inv = 1.0f / ((float)sqrt(var) + 1e-6f);
      }
    }
    const float* x = wav + (size_t)b * L;
    // ragged batch: the utterance has Lb samples and Fb frames of its own; the rows behind them are the zero padding the
    // reference's collate adds to the mel (lid/raw_datasets.py:345-365) and take no part in the utterance maximum
    const int Lb = n_samples ? min(L, n_samples[b]) : L, Lp = Lb + 2 * pad;
    if (f >= 1 + Lp / hop) {
      for (int m = lane; m < n_mels; m += 64) out[((size_t)b * F + f) * n_mels + m] = 0.f;
      continue;
    }
    // windowed frame, bit-reversed placement
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      int n = lane + 64 * q;
      int p = f * hop + n - LIDK_N_FFT / 2;
      if (p < 0) p = -p;
      if (p >= Lp) p = 2 * (Lp - 1) - p;
      int s = p - pad;
      float v = 0.f;
      if (s >= 0 && s < Lb) {
        if (prep.stats) {
          const size_t gi = (size_t)b * L + s;
          float z = (x[s] - mu) * inv;
          if (prep.dither != 0.f) z += prep.dither * (prep.noise ? prep.noise[gi] : uniform_from(prep.seed, gi));
          if (s > 0) {
            float z1 = (x[s - 1] - mu) * inv;
            if (prep.dither != 0.f) z1 += prep.dither * (prep.noise ? prep.noise[gi - 1] : uniform_from(prep.seed, gi - 1));
            z -= prep.coef * z1;
          }
          v = z * s_win[n];
        } else {
          v = x[s] * s_win[n];
        }
      }
      int r = (int)(__brev((unsigned)n) >> 23);
      re[r] = v; im[r] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    // 9 radix-2 DIT stages
#pragma unroll
    for (int st = 1; st <= 9; ++st) {
      if (dbg & 1) break;
      const int half = 1 << (st - 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int idx = lane + 64 * q;
        int pos = idx & (half - 1);
        int i = ((idx >> (st - 1)) << st) + pos;
        int j = i + half;
        int tk = pos << (9 - st);
        float wr = s_tw[tk][0], wi = -s_tw[tk][1];
        float xr = re[j], xi = im[j];
        float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
        float ur = re[i], ui = im[i];
        re[j] = ur - tr; im[j] = ui - ti;
        re[i] = ur + tr; im[i] = ui + ti;
      }
      __builtin_amdgcn_wave_barrier();
    }
    // power spectrum into re[0..256]
    float pw[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) { int k = lane + 64 * q; pw[q] = (k <= LIDK_N_FFT / 2) ? re[k] * re[k] + im[k] * im[k] : 0.f; }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 5; ++q) { int k = lane + 64 * q; if (k <= LIDK_N_FFT / 2) re[k] = pw[q]; }
    __builtin_amdgcn_wave_barrier();
    // mel projection + dB; lanes over mel bins (coalesced filterbank reads and output writes)
    float vmax = -INFINITY;
    for (int m = lane; m < n_mels && !(dbg & 2); m += 64) {
      float acc = 0.f;
      const int klo = m < LIDK_MEL_MAX ? s_lo[m] : 0, khi = m < LIDK_MEL_MAX ? s_hi[m] : LIDK_N_FFT / 2;
      if (m < LIDK_MEL_MAX && khi - klo < LIDK_MEL_TAPS) {
        const float* cf = s_coef + m * LIDK_MEL_TAPS;
        for (int k = klo; k <= khi; ++k) acc = fmaf(re[k], cf[k - klo], acc);
      } else {
        for (int k = klo; k <= khi; ++k) acc = fmaf(re[k], melfb[(size_t)k * n_mels + m], acc);
      }
      float db = 10.0f * log10f(fmaxf(acc, 1e-10f));
      out[((size_t)b * F + f) * n_mels + m] = db;
      vmax = fmaxf(vmax, db);
    }
    run_max = fmaxf(run_max, wave_max(vmax));
    __builtin_amdgcn_wave_barrier();
  }
  if (run_b >= 0 && lane == 0 && !(dbg & 4)) atomic_max_float(&utt_max[run_b], run_max);
}

// 16-byte form of the pass below for n_mels % 4 == 0 (a float4 never straddles a row): one workgroup-sized grid-stride sweep
__global__ void __launch_bounds__(256)
db_floor_mask4_kernel(float* __restrict__ out, const float* __restrict__ utt_max, const int32_t* __restrict__ spans, int mask_times,
                      int F, int n_mels, long n4, float top_db, const int32_t* __restrict__ n_samples, int L, int pad, int hop) {
  const int m4n = n_mels / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int m0 = (int)(i % m4n) * 4;
    const long bf = i / m4n;
    const int f = (int)(bf % F), b = (int)(bf / F);
    if (n_samples && f >= 1 + (min(L, n_samples[b]) + 2 * pad) / hop) continue;
    const float fl = utt_max[b] - top_db;
    float4 v = reinterpret_cast<const float4*>(out)[i];
    v.x = fmaxf(v.x, fl); v.y = fmaxf(v.y, fl); v.z = fmaxf(v.z, fl); v.w = fmaxf(v.w, fl);
    for (int r = 0; r < mask_times; ++r) {
      const int32_t* sp = spans + ((size_t)b * mask_times + r) * 4;
      const int t0 = sp[0], t1 = sp[1], c0 = sp[2], c1 = sp[3];
      const bool trow = f >= t0 && f < t1;
      if (trow || (m0 >= c0 && m0 < c1)) v.x = 0.f;
      if (trow || (m0 + 1 >= c0 && m0 + 1 < c1)) v.y = 0.f;
      if (trow || (m0 + 2 >= c0 && m0 + 2 < c1)) v.z = 0.f;
      if (trow || (m0 + 3 >= c0 && m0 + 3 < c1)) v.w = 0.f;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

__global__ void db_floor_mask_kernel(float* __restrict__ out, const float* __restrict__ utt_max, const int32_t* __restrict__ spans,
                                     int mask_times, int F, int n_mels, long n, float top_db,
                                     const int32_t* __restrict__ n_samples, int L, int pad, int hop) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int m = (int)(i % n_mels);
    long bf = i / n_mels;
    int f = (int)(bf % F), b = (int)(bf / F);
    if (n_samples && f >= 1 + (min(L, n_samples[b]) + 2 * pad) / hop) continue;       // zero padding rows stay zero
    float v = fmaxf(out[i], utt_max[b] - top_db);
    for (int r = 0; r < mask_times; ++r) {
      const int32_t* sp = spans + ((size_t)b * mask_times + r) * 4;
      if ((f >= sp[0] && f < sp[1]) || (m >= sp[2] && m < sp[3])) v = 0.f;
    }
    out[i] = v;
  }
}

// =====================================================================================================================
// STFT -> |X|^2 -> mel -> dB -> top_db floor + SpecAugment, round 3: radix-8 FFT in registers, two frames per wave.
//   * a workgroup (4 waves) owns a RUN of up to 40 consecutive frames of ONE utterance: the samples the run touches are
//     prepared once into LDS (normalise + dither: one hash per sample, not 6.4 - frames overlap 3.2 x and pre-emphasis needs a
//     neighbour); the frame load forms the pre-emphasis and the window from two LDS reads;
//   * a wave transforms frames f and f + 1 together as ONE complex FFT-512 (z = x_f + i x_{f+1}): 512 = 8 x 8 x 8, a lane
//     holds 8 complex points and does three radix-8 butterflies in registers with two transposes through LDS; the spectra
//     of the two real frames are separated by X_a[k] = (Z[k] + conj Z[N-k]) / 2, X_b[k] = (Z[k] - conj Z[N-k]) / 2i, the
//     partner Z[N-k] coming from the mirror lane by ONE shuffle per value;
//   * mel projection from the compact per-filter tap table (LDS), dB, row written once, running utterance maximum;
//   * the top_db floor and the SpecAugment fill need the utterance maximum, i.e. every run of the utterance: the run that
//     finishes LAST (a ticket counter per utterance, release / acquire fences) applies them to the utterance's rows while they
//     are still in L2 - there is no third launch.
// Index algebra (decimation in frequency): n = l + 64 q, k = k1 + 8 k2 + 64 k3:
//   stage 1 (lane l):            y_k1[l]    = W512^(l k1) * sum_q  x[l + 64 q]      W8^(q k1)
//   stage 2 (lane (k1, m1)):     z_k1k2[m1] = W64^(m1 k2) * sum_q2 y_k1[m1 + 8 q2]  W8^(q2 k2)
//   stage 3 (lane u = k1 + 8 k2): X[u + 64 k3] =            sum_m1 z_k1k2[m1]       W8^(m1 k3)
// =====================================================================================================================
#define R8_FRAMES 76                    // frames per workgroup: a 3 s utterance (301 frames) is four runs, 64 utterances = 256 CUs
#define R8_WAVES 16
#define R8_ZS 12544                     // staged samples: (R8_FRAMES - 1) * hop + 512 + margin, hop <= 160
#define R8_TAPS 32
#define R8_LDX 72                       // row pitch of the stage-1 -> stage-2 transpose (bank-conflict free per half wave)
#define R8_CHUNK 13                     // staged samples per thread: R8_CHUNK * 1024 >= R8_ZS
#define R8_XW 576                       // float2 slots of a wave's transpose buffer (8 * R8_LDX); the power spectra alias it

__device__ __forceinline__ void r8_dft(float (&ar)[8], float (&ai)[8]) {
  // forward DFT-8 in place: a[k] <- sum_q a[q] exp(-2 pi i q k / 8)
  const float c = 0.70710678118654752f;
  float br[8], bi[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) { br[q] = ar[q] + ar[q + 4]; bi[q] = ai[q] + ai[q + 4]; }
  {
    float dr = ar[0] - ar[4], di = ai[0] - ai[4];
    br[4] = dr; bi[4] = di;
    dr = ar[1] - ar[5]; di = ai[1] - ai[5];
    br[5] = c * (dr + di); bi[5] = c * (di - dr);
    dr = ar[2] - ar[6]; di = ai[2] - ai[6];
    br[6] = di; bi[6] = -dr;
    dr = ar[3] - ar[7]; di = ai[3] - ai[7];
    br[7] = c * (di - dr); bi[7] = -c * (di + dr);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {            // radix-4 on b[4h .. 4h+3] -> outputs a[2m + h]
    const float c0r = br[4 * h], c0i = bi[4 * h], c1r = br[4 * h + 1], c1i = bi[4 * h + 1];
    const float c2r = br[4 * h + 2], c2i = bi[4 * h + 2], c3r = br[4 * h + 3], c3i = bi[4 * h + 3];
    const float d0r = c0r + c2r, d0i = c0i + c2i, d1r = c1r + c3r, d1i = c1i + c3i;
    const float d2r = c0r - c2r, d2i = c0i - c2i, d3r = c1i - c3i, d3i = -(c1r - c3r);
    ar[0 + h] = d0r + d1r; ai[0 + h] = d0i + d1i;
    ar[4 + h] = d0r - d1r; ai[4 + h] = d0i - d1i;
    ar[2 + h] = d2r + d3r; ai[2 + h] = d2i + d3i;
    ar[6 + h] = d2r - d3r; ai[6 + h] = d2i - d3i;
  }
}

// Dither of the fused path: a 32-bit integer hash of (seed, sample index) -> 24-bit uniform.  splitmix64 (uniform_from) costs
// ~60 VALU issue slots per sample on CDNA (64-bit multiplies are built from quarter-rate 32-bit ones) and was 4 us of this
// kernel; three 32-bit multiplies with xor-shifts ("lowbias32") pass the same smoke checks for a 1e-5 * U[0,1) dither.
__device__ __forceinline__ float r8_uniform(uint32_t seed_lo, uint32_t seed_hi, uint32_t idx) {
  uint32_t h = idx * 0x9E3779B1u + seed_lo;
  h ^= h >> 16; h *= 0x21F0AAADu;
  h ^= seed_hi;
  h ^= h >> 15; h *= 0x735A2D97u;
  h ^= h >> 15;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// W512^j = exp(-2 pi i j / 512) from the [256][2] (cos, sin) table, any j >= 0
__device__ __forceinline__ void r8_twiddle(const float* __restrict__ tw, int j, float& wr, float& wi) {
  j &= 511;
  const float sg = j >= 256 ? -1.f : 1.f;
  j &= 255;
  wr = sg * tw[2 * j]; wi = -sg * tw[2 * j + 1];
}

__global__ void __launch_bounds__(64 * R8_WAVES)
stft_mel_r8_kernel(const float* __restrict__ wav, const float* __restrict__ window, const float* __restrict__ twiddle,
                   const float* __restrict__ melfb, float* __restrict__ out, float* __restrict__ utt_max,
                   int* __restrict__ tickets, int B, int L, int pad, int hop, int F, int n_mels, int runs,
                   const int32_t* __restrict__ n_samples, WavPrep prep, const int32_t* __restrict__ spans, int mask_times,
                   float top_db, const int32_t* __restrict__ mel_rng, const float* __restrict__ mel_coef, int fused_floor, int dbg) {
  __shared__ float s_y[R8_ZS];                                   // prepared + pre-emphasised samples y(s), slot 1 = y(s_first)
  __shared__ __attribute__((aligned(16))) float2 s_x[R8_WAVES][R8_XW];
  __shared__ float2 s_t1[8][64];                                 // stage-1 twiddles W512^(l k1)
  __shared__ int s_lo[LIDK_MEL_MAX], s_hi[LIDK_MEL_MAX];
  __shared__ __attribute__((aligned(16))) float s_coef[LIDK_MEL_MAX * R8_TAPS];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = 64 * R8_WAVES;
  const int b = blockIdx.x / runs, run = blockIdx.x - b * runs;
  const int Lb = n_samples ? min(L, n_samples[b]) : L, Lp = Lb + 2 * pad, Fb = 1 + Lp / hop;
  const int f0 = run * R8_FRAMES, f1 = min(F, f0 + R8_FRAMES);
  const int nm = min(n_mels, LIDK_MEL_MAX);
  // ---- the run's raw samples first (coalesced, R8_CHUNK per thread, all loads in flight while the tables below are fetched)
  const float coef = prep.stats ? prep.coef : 0.f;
  const int s_first = max(0, f0 * hop - LIDK_N_FFT / 2 - pad - 2);        // slot i holds y(s_first - 1 + i)
  const int s_end = min(Lb - 1, (f1 - 1) * hop + LIDK_N_FFT / 2 - 1 - pad);
  const int cnt = s_end - s_first + 2;                                     // slots 0 .. cnt - 1
  const float* x = wav + (size_t)b * L;
  const bool active = f0 < Fb;                                            // block-uniform: a run of padding rows only
  float xv[R8_CHUNK], nv[R8_CHUNK];
#pragma unroll
  for (int e = 0; e < R8_CHUNK; ++e) {
    const int sidx = s_first - 1 + tid + e * nthr;
    const bool ok = active && tid + e * nthr < cnt && sidx >= 0;
    xv[e] = ok ? x[sidx] : 0.f;
    nv[e] = (ok && prep.noise) ? prep.noise[(size_t)b * L + sidx] : 0.f;
  }
  // per-lane constants (window taps, stage-2 twiddles W64^(m1 k2)) are requested here too: every buffer this kernel reads is
  // touched in this first batch, so the address translations and the memory latencies of the half-dozen small tables overlap
  // instead of queueing up phase after phase (measured: ~4 us per phase of dependent first touches)
  float win[8], t2r[8], t2i[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    win[q] = window[lane + 64 * q];
    r8_twiddle(twiddle, 8 * (lane & 7) * q, t2r[q], t2i[q]);
  }
  double sm = 0.0, sq = 0.0;
  if (prep.stats)
    for (int p = 0; p < WAV_STAT_PARTS; ++p) { sm += prep.stats[((size_t)b * WAV_STAT_PARTS + p) * 2]; sq += prep.stats[((size_t)b * WAV_STAT_PARTS + p) * 2 + 1]; }
  // ---- compact mel table: nonzero range [lo, hi] of every filter and its taps - host-built (mel_rng [2][128], mel_coef
  // [128][R8_TAPS]: one coalesced copy) or, without it, scanned from the dense filterbank as stft_mel_kernel does
  if (mel_rng) {
    for (int m = tid; m < LIDK_MEL_MAX; m += nthr) { s_lo[m] = mel_rng[m]; s_hi[m] = mel_rng[LIDK_MEL_MAX + m]; }
    for (int i = tid; i < LIDK_MEL_MAX * R8_TAPS; i += nthr) {
      const int j = i / LIDK_MEL_MAX, m = i - j * LIDK_MEL_MAX;
      s_coef[i] = mel_coef[m * R8_TAPS + j];
    }
  } else {
    for (int m = tid; m < LIDK_MEL_MAX; m += nthr) { s_lo[m] = LIDK_N_FFT / 2 + 1; s_hi[m] = -1; }
    __syncthreads();
    for (int k = tid; k <= LIDK_N_FFT / 2; k += nthr) {
      const float* row = melfb + (size_t)k * n_mels;
      for (int m = 0; m < nm; ++m)
        if (row[m] != 0.f) { atomicMin(&s_lo[m], k); atomicMax(&s_hi[m], k); }
    }
  }
  for (int i = tid; i < 8 * 64; i += nthr) {
    float wr, wi;
    r8_twiddle(twiddle, (i >> 6) * (i & 63), wr, wi);
    s_t1[i >> 6][i & 63] = make_float2(wr, wi);
  }
  // ---- this utterance's mean and 1 / (std + 1e-6), then y(s) = z(s) - coef * z(s - 1) into LDS: z(s - 1) comes from the lane
  // below (one shuffle); lane 0 of a wave forms its neighbour itself (one extra load + hash per chunk element)
  float mu = 0.f, inv = 1.f;
  if (prep.stats) {
    const int Ln = max(2, Lb);
    const double mean = sm / (double)Ln;
    double var = (sq - sm * mean) / (double)(Ln - 1);
    if (var < 0.0) var = 0.0;
    mu = (float)mean;
    inv = 1.0f / ((float)sqrt(var) + 1e-6f);
  }
  if (active) {
    const uint32_t sd_lo = (uint32_t)prep.seed, sd_hi = (uint32_t)(prep.seed >> 32);
    const bool dith = prep.stats && prep.dither != 0.f && !(dbg & 4);
#pragma unroll
    for (int e = 0; e < R8_CHUNK; ++e) {
      const int i = tid + e * nthr, sidx = s_first - 1 + i;
      float z = 0.f;
      if (sidx >= 0 && i < cnt) {
        z = xv[e];
        if (prep.stats) {
          z = (z - mu) * inv;
          if (dith) z += prep.dither * (prep.noise ? nv[e] : r8_uniform(sd_lo, sd_hi, (uint32_t)(b * L + sidx)));
        }
      }
      float zp = __shfl_up(z, 1, 64);
      if (lane == 0) {                                                    // the sample in front of this wave's 64
        zp = 0.f;
        const int sp = sidx - 1;
        if (sp >= 0 && i < cnt) {
          zp = x[sp];
          if (prep.stats) {
            zp = (zp - mu) * inv;
            if (dith) zp += prep.dither * (prep.noise ? prep.noise[(size_t)b * L + sp] : r8_uniform(sd_lo, sd_hi, (uint32_t)(b * L + sp)));
          }
        }
      }
      if (i < cnt) s_y[i] = z - coef * zp;
    }
  }
  if (!mel_rng) {
    __syncthreads();
    for (int i = tid; i < LIDK_MEL_MAX * R8_TAPS; i += nthr) {
      const int m = i / R8_TAPS, j = i - m * R8_TAPS;
      float c = 0.f;
      if (m < nm && s_lo[m] + j <= s_hi[m]) c = melfb[(size_t)(s_lo[m] + j) * n_mels + m];
      s_coef[j * LIDK_MEL_MAX + m] = c;
    }
  }
  __syncthreads();
  // does every filter fit the compact table?  Host-built tables do by construction; the scan path checks (block-uniform)
  bool compact = true;
  if (!mel_rng)
    for (int m = 0; m < nm; ++m) compact = compact && (s_hi[m] - s_lo[m] < R8_TAPS);
  // taps of the widest filter met by this wave in each of its output rounds (output o = lane + 64 r: frame o / n_mels, filter
  // o % n_mels), rounded up to 4: the product below runs that many taps for all 64 lanes
  int jmax[4] = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = lane + 64 * r;
    int nt = 0;
    if (o < 2 * n_mels) { const int m = o % n_mels; nt = m < LIDK_MEL_MAX ? s_hi[m] - s_lo[m] + 1 : 0; }
    nt = (int)wave_max((float)nt);
    jmax[r] = min(R8_TAPS, (nt + 3) / 4 * 4);
  }
  float2* xw = s_x[wave];
  float* pw0 = reinterpret_cast<float*>(xw);                              // power spectra of the pair: [2][296] floats
  float run_max = -INFINITY;
  for (int fa = f0 + 2 * wave; fa < f1; fa += 2 * R8_WAVES) {            // wave-uniform
    const int fb = fa + 1;
    const bool has_a = fa < Fb, has_b = fb < f1 && fb < Fb;
    if (!has_a) {                                                         // zero rows behind the utterance's own frames
      for (int m = lane; m < n_mels; m += 64) {
        out[((size_t)b * F + fa) * n_mels + m] = 0.f;
        if (fb < f1) out[((size_t)b * F + fb) * n_mels + m] = 0.f;
      }
      continue;
    }
    float ar[8], ai[8];
    // interior pair (no reflection, no zero padding inside either frame): contiguous LDS reads, no branches
    const int pa0 = fa * hop - LIDK_N_FFT / 2 - pad;                      // sample index of tap 0 of frame a
    if (has_b && pa0 >= 0 && pa0 + hop + LIDK_N_FFT - 1 < Lb) {
      const float* ya = s_y + (pa0 - s_first + 1) + lane;
#pragma unroll
      for (int q = 0; q < 8; ++q) { ar[q] = ya[64 * q] * win[q]; ai[q] = ya[64 * q + hop] * win[q]; }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int n = lane + 64 * q;
        float va = 0.f, vb = 0.f;
        {
          int p = fa * hop + n - LIDK_N_FFT / 2;
          if (p < 0) p = -p;
          if (p >= Lp) p = 2 * (Lp - 1) - p;
          const int sidx = p - pad;
          if (sidx >= 0 && sidx < Lb) va = s_y[sidx - s_first + 1] * win[q];
        }
        if (has_b) {
          int p = fb * hop + n - LIDK_N_FFT / 2;
          if (p < 0) p = -p;
          if (p >= Lp) p = 2 * (Lp - 1) - p;
          const int sidx = p - pad;
          if (sidx >= 0 && sidx < Lb) vb = s_y[sidx - s_first + 1] * win[q];
        }
        ar[q] = va; ai[q] = vb;
      }
    }
    if (!(dbg & 1)) {
    // ---- stage 1
    r8_dft(ar, ai);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float2 t = s_t1[k][lane];
      xw[k * R8_LDX + lane] = make_float2(ar[k] * t.x - ai[k] * t.y, ar[k] * t.y + ai[k] * t.x);
    }
    // ---- stage 2: lane (k1 = lane >> 3, m1 = lane & 7)
    {
      const int k1 = lane >> 3, m1 = lane & 7;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float2 v = xw[k1 * R8_LDX + m1 + 8 * q]; ar[q] = v.x; ai[q] = v.y; }
      r8_dft(ar, ai);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        xw[(k1 * 8 + k) * 9 + m1] = make_float2(ar[k] * t2r[k] - ai[k] * t2i[k], ar[k] * t2i[k] + ai[k] * t2r[k]);
    }
    // ---- stage 3: lane u = k1 + 8 k2 holds X[u + 64 k3]
    {
      const int k1 = lane & 7, k2 = lane >> 3;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const float2 v = xw[(k1 * 8 + k2) * 9 + q]; ar[q] = v.x; ai[q] = v.y; }
      r8_dft(ar, ai);
    }
    }
    // ---- split the two real spectra: partner Z[N - k] sits in lane (64 - l) & 63, register 7 - j (lane 0: register (8 - j) & 7)
    const int mirror = (64 - lane) & 63;
    float par[5], pbr[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {                                        // k = lane + 64 j; j = 4 only for k = 256 (lane 0)
      const float pr_n = __shfl(ar[(7 - j) & 7], mirror, 64), pi_n = __shfl(ai[(7 - j) & 7], mirror, 64);
      const float pr_0 = __shfl(ar[(8 - j) & 7], 0, 64), pi_0 = __shfl(ai[(8 - j) & 7], 0, 64);
      const float zr = ar[j & 7], zi = ai[j & 7];
      const float qr = lane == 0 ? pr_0 : pr_n, qi = lane == 0 ? pi_0 : pi_n;         // Z[N - k]
      const float xar = 0.5f * (zr + qr), xai = 0.5f * (zi - qi);
      const float xbr = 0.5f * (zi + qi), xbi = -0.5f * (zr - qr);
      par[j] = xar * xar + xai * xai;
      pbr[j] = xbr * xbr + xbi * xbi;
    }
    // the transpose buffer is free now: power spectra [2][296], entries 257 .. 295 zero (the fixed-length mel product runs past
    // the last bin of the top filters)
#pragma unroll
    for (int j = 0; j < 4; ++j) { pw0[lane + 64 * j] = par[j]; pw0[296 + lane + 64 * j] = pbr[j]; }
    if (lane == 0) { pw0[256] = par[4]; pw0[296 + 256] = pbr[4]; }
    if (lane >= 1 && lane < 40) { pw0[256 + lane] = 0.f; pw0[296 + 256 + lane] = 0.f; }
    // ---- mel projection + dB for both frames.  Compact table: the 2 * n_mels (frame, filter) outputs are dealt to the lanes and
    // every output is a FIXED R8_TAPS-term product (taps beyond the filter are zero, the power buffer's tail is zero), fully
    // unrolled with four accumulators so the LDS reads are issued back to back instead of one latency per tap
    float vmax = -INFINITY;
    if (dbg & 2) {
      if (lane == 0) vmax = pw0[5];
    } else if (compact && n_mels <= LIDK_MEL_MAX) {
      const int nout = (has_b ? 2 : 1) * n_mels;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (64 * r >= nout) break;                                        // wave-uniform
        const int o = lane + 64 * r;
        const bool live = o < nout;
        const int oc = live ? o : 0, fr2 = oc >= n_mels, m = oc - fr2 * n_mels, f = fr2 ? fb : fa;
        const float* pw = pw0 + fr2 * 296 + s_lo[m];
        const float* cf = s_coef + m;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int j = 0; j < jmax[r]; j += 4) {
          a0 = fmaf(pw[j], cf[j * LIDK_MEL_MAX], a0);
          a1 = fmaf(pw[j + 1], cf[(j + 1) * LIDK_MEL_MAX], a1);
          a2 = fmaf(pw[j + 2], cf[(j + 2) * LIDK_MEL_MAX], a2);
          a3 = fmaf(pw[j + 3], cf[(j + 3) * LIDK_MEL_MAX], a3);
        }
        if (live) {
          const float db = 10.0f * log10f(fmaxf((a0 + a1) + (a2 + a3), 1e-10f));
          out[((size_t)b * F + f) * n_mels + m] = db;
          vmax = fmaxf(vmax, db);
        }
      }
    } else {
      for (int fr2 = 0; fr2 < 2; ++fr2) {
        if (fr2 == 1 && !has_b) break;
        const float* pw = pw0 + fr2 * 296;
        const int f = fr2 ? fb : fa;
        for (int m = lane; m < n_mels; m += 64) {
          float acc = 0.f;
          const int klo = m < LIDK_MEL_MAX ? s_lo[m] : 0, khi = m < LIDK_MEL_MAX ? s_hi[m] : LIDK_N_FFT / 2;
          for (int k = klo; k <= khi; ++k) acc = fmaf(pw[k], melfb[(size_t)k * n_mels + m], acc);
          const float db = 10.0f * log10f(fmaxf(acc, 1e-10f));
          out[((size_t)b * F + f) * n_mels + m] = db;
          vmax = fmaxf(vmax, db);
        }
      }
    }
    if (fb < f1 && !has_b)                                               // frame fb is a padding row of a ragged batch
      for (int m = lane; m < n_mels; m += 64) out[((size_t)b * F + fb) * n_mels + m] = 0.f;
    run_max = fmaxf(run_max, wave_max(vmax));
  }
  if (lane == 0 && run_max > -INFINITY) atomic_max_float(&utt_max[b], run_max);
  if (!fused_floor) return;
  // ---- (LIDK_STFT_TAIL=1) the run that finishes last applies the top_db floor and the SpecAugment fill to the whole utterance
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = (atomicAdd(&tickets[b], 1) == runs - 1);
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const float floor_db = __hip_atomic_load(&utt_max[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - top_db;
  const int nrow = min(Fb, F) * n_mels;                                  // the utterance's own rows; padding rows stay zero
  float* ob = out + (size_t)b * F * n_mels;
  for (int i = tid; i < nrow; i += nthr) {
    const int f = i / n_mels, m = i - f * n_mels;
    float v = fmaxf(ob[i], floor_db);
    for (int r = 0; r < mask_times; ++r) {
      const int32_t* sp = spans + ((size_t)b * mask_times + r) * 4;
      const int t0 = sp[0], t1 = sp[1], c0 = sp[2], c1 = sp[3];
      if ((f >= t0 && f < t1) || (m >= c0 && m < c1)) v = 0.f;
    }
    ob[i] = v;
  }
  if (tid == 0) tickets[b] = 0;                                           // ready for the next launch
}

__global__ void r8_init_kernel(float* utt_max, int* tickets, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { utt_max[i] = -INFINITY; tickets[i] = 0; }
}

static int r8_dbg() {
  static const int v = getenv("LIDK_STFT_DBG") ? atoi(getenv("LIDK_STFT_DBG")) : 0;      // tuning aid: 1 no FFT, 2 no mel, 4 no dither hash
  return v;
}
static int r8_fused_floor() {
  // 1: the utterance's last workgroup applies floor + masks (no third launch).  Measured SLOWER on MI355X (163 vs 72 us per
  // batch): the agent-scope release fence every workgroup must issue writes back its XCD's whole L2.  Kept for the record.
  static const int v = getenv("LIDK_STFT_TAIL") ? atoi(getenv("LIDK_STFT_TAIL")) : 0;
  return v;
}
static bool r8_usable(int hop, int n_mels) {
  static const bool off = getenv("LIDK_STFT_V1") && atoi(getenv("LIDK_STFT_V1")) != 0;
  return !off && hop > 0 && (R8_FRAMES - 1) * hop + LIDK_N_FFT + 8 <= R8_ZS && n_mels <= LIDK_MEL_MAX;
}

extern "C" int lidk_logmel(const float* wav, const float* window, const float* twiddle, const float* melfb, float* out,
                           float* utt_max, int B, int L, int pad, int hop, int n_mels, const int32_t* spans, int mask_times,
                           float top_db, const int32_t* n_samples, const int32_t* mel_rng, const float* mel_coef, void* stream) {
  if (!wav || !window || !twiddle || !melfb || !out || !utt_max || B <= 0 || hop <= 0 || pad < 0 || n_mels <= 0) return LIDK_ERR_ARG;
  if ((mel_rng == nullptr) != (mel_coef == nullptr)) return LIDK_ERR_ARG;
  if (L + 2 * pad <= LIDK_N_FFT / 2) return LIDK_ERR_ARG;         // reflect padding needs more than n_fft/2 samples
  if (mask_times > 0 && !spans) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  const int F = 1 + (L + 2 * pad) / hop;
  if (r8_usable(hop, n_mels)) {                       // two launches: reset, then STFT .. floor + masks in one kernel
    int* tickets = reinterpret_cast<int*>(utt_max + B);
    r8_init_kernel<<<cdiv(B, 256), 256, 0, s>>>(utt_max, tickets, B);
    const int runs = cdiv(F, R8_FRAMES);
    stft_mel_r8_kernel<<<B * runs, 64 * R8_WAVES, 0, s>>>(wav, window, twiddle, melfb, out, utt_max, tickets, B, L, pad, hop, F,
                                                         n_mels, runs, n_samples, WavPrep{nullptr, nullptr, 0.f, 0.f, 0ull}, spans,
                                                         mask_times > 0 ? mask_times : 0, top_db, mel_rng, mel_coef, r8_fused_floor(), r8_dbg());
    if (!r8_fused_floor()) {
      const long n = (long)B * F * n_mels;
      if (n_mels % 4 == 0 && ((uintptr_t)out & 15) == 0) {
        int eb = (int)((n / 4 + 255) / 256); if (eb > 4096) eb = 4096;
        db_floor_mask4_kernel<<<eb, 256, 0, s>>>(out, utt_max, spans, mask_times > 0 ? mask_times : 0, F, n_mels, n / 4, top_db, n_samples, L, pad, hop);
      } else {
        int eb = (int)((n + 255) / 256); if (eb > 8192) eb = 8192;
        db_floor_mask_kernel<<<eb, 256, 0, s>>>(out, utt_max, spans, mask_times > 0 ? mask_times : 0, F, n_mels, n, top_db, n_samples, L, pad, hop);
      }
    }
    return launch_status();
  }
  fill_kernel<<<cdiv(B, 256), 256, 0, s>>>(utt_max, B, -INFINITY);
  long nframes = (long)B * F;
  int blocks = (int)((nframes + STFT_WAVES - 1) / STFT_WAVES); if (blocks > 256) blocks = 256;      // one workgroup per CU
  static const int dbg = getenv("LIDK_STFT_DBG") ? atoi(getenv("LIDK_STFT_DBG")) : 0;     // tuning aid: 1 no FFT, 2 no mel, 4 no max
  stft_mel_kernel<<<blocks, 64 * STFT_WAVES, 0, s>>>(wav, window, twiddle, melfb, out, utt_max, B, L, pad, hop, F, n_mels, dbg, n_samples,
                                                     WavPrep{nullptr, nullptr, 0.f, 0.f, 0ull});
  long n = nframes * n_mels;
  int eb = (int)((n + 255) / 256); if (eb > 8192) eb = 8192;
  db_floor_mask_kernel<<<eb, 256, 0, s>>>(out, utt_max, spans, mask_times > 0 ? mask_times : 0, F, n_mels, n, top_db, n_samples, L, pad, hop);
  return launch_status();
}

// Raw waveform -> log-mel in three launches (statistics, STFT with the waveform preparation in its frame load, floor + masks)
// instead of five: normalize_wav + dither/pre-emphasis + log-mel of lid/audio_processor.py:72-134.  stats: B * 16 doubles of
// scratch.  noise (may be NULL): explicit dither values [B][L] instead of the counter-based generator (tests).
extern "C" int lidk_wav2mel(const float* wav, const float* window, const float* twiddle, const float* melfb, float* out,
                            float* utt_max, double* stats, int B, int L, int pad, int hop, int n_mels, const int32_t* spans,
                            int mask_times, float top_db, const int32_t* n_samples, float coef, float dither, uint64_t seed,
                            const float* noise, const int32_t* mel_rng, const float* mel_coef, void* stream) {
  if (!wav || !window || !twiddle || !melfb || !out || !utt_max || !stats || B <= 0 || L < 2 || hop <= 0 || pad < 0 || n_mels <= 0)
    return LIDK_ERR_ARG;
  if ((mel_rng == nullptr) != (mel_coef == nullptr)) return LIDK_ERR_ARG;
  if (L + 2 * pad <= LIDK_N_FFT / 2) return LIDK_ERR_ARG;
  if (mask_times > 0 && !spans) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  const int F = 1 + (L + 2 * pad) / hop;
  if (r8_usable(hop, n_mels)) {                       // two launches: statistics, then STFT .. floor + masks in one kernel
    int* tickets = reinterpret_cast<int*>(utt_max + B);
    wav_stats_kernel<<<dim3(B, WAV_STAT_PARTS), 256, 0, s>>>(wav, L, n_samples, stats, utt_max, tickets);
    const int runs = cdiv(F, R8_FRAMES);
    stft_mel_r8_kernel<<<B * runs, 64 * R8_WAVES, 0, s>>>(wav, window, twiddle, melfb, out, utt_max, tickets, B, L, pad, hop, F,
                                                         n_mels, runs, n_samples,
                                                         WavPrep{stats, noise, coef, dither, (unsigned long long)seed}, spans,
                                                         mask_times > 0 ? mask_times : 0, top_db, mel_rng, mel_coef, r8_fused_floor(), r8_dbg());
    if (!r8_fused_floor()) {
      const long n = (long)B * F * n_mels;
      if (n_mels % 4 == 0 && ((uintptr_t)out & 15) == 0) {
        int eb = (int)((n / 4 + 255) / 256); if (eb > 4096) eb = 4096;
        db_floor_mask4_kernel<<<eb, 256, 0, s>>>(out, utt_max, spans, mask_times > 0 ? mask_times : 0, F, n_mels, n / 4, top_db, n_samples, L, pad, hop);
      } else {
        int eb = (int)((n + 255) / 256); if (eb > 8192) eb = 8192;
        db_floor_mask_kernel<<<eb, 256, 0, s>>>(out, utt_max, spans, mask_times > 0 ? mask_times : 0, F, n_mels, n, top_db, n_samples, L, pad, hop);
      }
    }
    return launch_status();
  }
  wav_stats_kernel<<<dim3(B, WAV_STAT_PARTS), 256, 0, s>>>(wav, L, n_samples, stats, utt_max, nullptr);
  long nframes = (long)B * F;
  int blocks = (int)((nframes + STFT_WAVES - 1) / STFT_WAVES); if (blocks > 256) blocks = 256;
  stft_mel_kernel<<<blocks, 64 * STFT_WAVES, 0, s>>>(wav, window, twiddle, melfb, out, utt_max, B, L, pad, hop, F, n_mels, 0, n_samples,
                                                     WavPrep{stats, noise, coef, dither, (unsigned long long)seed});
  long n = nframes * n_mels;
  int eb = (int)((n + 255) / 256); if (eb > 8192) eb = 8192;
  db_floor_mask_kernel<<<eb, 256, 0, s>>>(out, utt_max, spans, mask_times > 0 ? mask_times : 0, F, n_mels, n, top_db, n_samples, L, pad, hop);
  return launch_status();
}

// ------------------------------------------------------------------------------------ im2col for Conv1d(k3, s2, p1)
template <typename T>
__global__ void im2col_k3s2_kernel(const float* __restrict__ mel, T* __restrict__ out, int B, int F, int C, int T_) {
  const long n4 = (long)B * T_ * 3 * C / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    long e = i * 4;
    int col = (int)(e % (3 * C));
    long bt = e / (3 * C);
    int t = (int)(bt % T_), b = (int)(bt / T_);
    int k = col / C, ci = col - k * C;
    int f = 2 * t + k - 1;
    float4 v = make_float4(0, 0, 0, 0);
    if (f >= 0 && f < F) v = load4(mel + ((size_t)b * F + f) * C + ci);
    store4(out + e, v);
  }
}

extern "C" int lidk_im2col_k3s2(const float* mel, void* out, int B, int F, int C, int T_, int dtype, void* stream) {
  if (!mel || !out || B <= 0 || F <= 0 || C <= 0 || (C & 3) || T_ != (F + 2 - 3) / 2 + 1) return LIDK_ERR_ARG;
  long n4 = (long)B * T_ * 3 * C / 4;
  int blocks = (int)((n4 + 255) / 256); if (blocks > 8192) blocks = 8192;
  LIDK_DISPATCH(dtype, im2col_k3s2_kernel<T><<<blocks, 256, 0, as_stream(stream)>>>(mel, (T*)out, B, F, C, T_));
  return launch_status();
}
