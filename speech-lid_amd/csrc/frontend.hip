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
                 float* __restrict__ utt_max) {
  __shared__ double red[4][2];
  const int b = blockIdx.x, part = blockIdx.y;
  const int Lb = n_samples ? max(2, min(L, n_samples[b])) : L;
  const int per = (Lb + WAV_STAT_PARTS - 1) / WAV_STAT_PARTS, lo = part * per, hi = min(Lb, lo + per);
  const float* x = wav + (size_t)b * L;
  double s = 0.0, q = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += 256) { const double v = (double)x[i]; s += v; q += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[((size_t)b * WAV_STAT_PARTS + part) * 2 + 0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    stats[((size_t)b * WAV_STAT_PARTS + part) * 2 + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    if (part == 0) utt_max[b] = -INFINITY;
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

extern "C" int lidk_logmel(const float* wav, const float* window, const float* twiddle, const float* melfb, float* out,
                           float* utt_max, int B, int L, int pad, int hop, int n_mels, const int32_t* spans, int mask_times,
                           float top_db, const int32_t* n_samples, void* stream) {
  if (!wav || !window || !twiddle || !melfb || !out || !utt_max || B <= 0 || hop <= 0 || pad < 0 || n_mels <= 0) return LIDK_ERR_ARG;
  if (L + 2 * pad <= LIDK_N_FFT / 2) return LIDK_ERR_ARG;         // reflect padding needs more than n_fft/2 samples
  if (mask_times > 0 && !spans) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  const int F = 1 + (L + 2 * pad) / hop;
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
                            const float* noise, void* stream) {
  if (!wav || !window || !twiddle || !melfb || !out || !utt_max || !stats || B <= 0 || L < 2 || hop <= 0 || pad < 0 || n_mels <= 0)
    return LIDK_ERR_ARG;
  if (L + 2 * pad <= LIDK_N_FFT / 2) return LIDK_ERR_ARG;
  if (mask_times > 0 && !spans) return LIDK_ERR_ARG;
  hipStream_t s = as_stream(stream);
  const int F = 1 + (L + 2 * pad) / hop;
  wav_stats_kernel<<<dim3(B, WAV_STAT_PARTS), 256, 0, s>>>(wav, L, n_samples, stats, utt_max);
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
