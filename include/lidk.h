/* lidk — C-ABI of the MI355X-native spoken-LID training hot path (gfx950 only).
 *
 * The reference (kouyt5/speech-lid) has no native code: every op below is what its Python reaches
 * through torch / torchaudio.  Each entry point names the reference call site it
 * replaces.  Conventions:
 *   - extern "C", plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*), never synchronise, never allocate;
 *   - return 0 (LIDK_OK) or a negative LIDK_ERR_* code; no exceptions cross the ABI; caller owns buffers;
 *   - `dtype` selects the storage type "T" of activations: LIDK_F32 (parity mode) or LIDK_BF16 (fast mode).
 *     The residual stream, statistics, losses, parameters (master copy) and gradients are always f32.
 *   - row-major everywhere; activations are [M, C] with M = B*T (utterance-major, then time).
 *   - outputs documented "+=" accumulate into the destination (gradient arena, zeroed once per step).
 */
#ifndef LIDK_H
#define LIDK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LIDK_OK 0
#define LIDK_ERR_ARG (-1)
#define LIDK_ERR_LAUNCH (-2)
#define LIDK_ERR_UNSUPPORTED (-3)

enum { LIDK_F32 = 0, LIDK_BF16 = 1 };
enum { LIDK_ACT_NONE = 0, LIDK_ACT_SWISH = 1, LIDK_ACT_RELU = 2, LIDK_ACT_SWISH_GRAD = 3, LIDK_ACT_GELU = 4,   /* GELU: exact erf form; */
       LIDK_ACT_GELU_GRAD = 5 };                                                                  /* out2 = pre-activation as for SWISH */

#define LIDK_N_FFT 512
#define LIDK_N_FREQ 257
#define LIDK_LN_PARTIAL_BLOCKS 256   /* rows of the column-reduction scratch used by *_bwd kernels */
#define LIDK_LN_BWD_BLOCKS 1024      /* rows of scratch used by lidk_layernorm_bwd */
#define LIDK_BN_PARTIAL_BLOCKS 1024   /* partial rows written by lidk_bn_swish_bwd_reduce */

int lidk_version(void);

/* ------------------------------------------------------------------ feature path (rows a1-a5) */
/* lid/audio_processor.py:108-115 normalize_wav: (x-mean)/(std_unbiased+1e-6) per utterance. in/out [B][L].
 * n_samples (may be NULL): true length of every utterance of a ragged, zero-padded batch; statistics then run over the
 * utterance's own samples and the tail of the row is zeroed. */
int lidk_normalize_wav(const float* wav, float* out, int B, int L, const int32_t* n_samples, void* stream);
/* lid/audio_processor.py:128-134 wav_augment dither + pre-emphasis: x += dither*U[0,1); y[0]=x[0], y[t]=x[t]-coef*x[t-1].
 * noise: optional [B][L] U[0,1) draws (parity tests); NULL -> counter-based generator keyed by (seed, index). */
int lidk_dither_preemph(const float* wav, float* out, const float* noise, int B, int L, float coef, float dither,
                        uint64_t seed, void* stream);

/* Speed perturbation (lid/audio_processor.py:136-156: sox "speed" v then "rate" sr, v drawn from {0.9, 1.0, 1.1}) as polyphase
 * resampling on the device: y[b][n] = sum_j taps[(n*p) % q][j] * x[b][(n*p) / q - left + j] for n < n_out[b], 0 behind.
 * tables: device array of n_tables records {const float* taps [q][ntaps]; int p, q, ntaps, left;} (v = p/q; host-built
 * windowed-sinc rows); table_of [B] selects the record of each utterance.  x [B][Lin] (n_in [B] true lengths or NULL),
 * y [B][Lout], n_out [B] = round(n_in / v).  The sox resampler's own filter is not part of the reference tree: parity unpinned,
 * checked against scipy.signal.resample_poly and a float64 restatement. */
int lidk_speed_perturb(const float* x, int B, int Lin, const int32_t* n_in, float* y, int Lout, const int32_t* n_out,
                       const void* tables, int n_tables, const int32_t* table_of, void* stream);
/* lid/audio_processor.py:72-105 _internal_wav2mel (torchaudio MelSpectrogram + AmplitudeToDB(top_db=80)) fused with
 * lid/audio_processor.py:225-227 spectrogram_augment masks and lid/raw_datasets.py:345-365 collate layout.
 * wav [B][L] -> out [B][F][n_mels] f32 dB, F = 1 + (L + 2*pad)/hop.  window [512] (hann(win) centred), twiddle [256][2]
 * (cos,sin of 2*pi*k/512), melfb [257][n_mels].  utt_max: scratch of 2*B floats (per-utterance dB max for the top_db floor +
 * one int ticket per utterance: the workgroup that finishes an utterance last applies the floor and the masks).
 * spans [B][mask_times][4] int32 = (t0,t1,f0,f1) filled with 0.0 dB, or NULL/mask_times=0 for none.
 * mel_rng [2][128] int32 + mel_coef [128][32] f32 (both may be NULL): the filterbank in compact form - first / last non-zero FFT
 * bin of every filter and its taps from the first one on (zero padded) - so a workgroup copies 17 KB instead of scanning melfb.
 * n_samples (may be NULL): ragged batch - utterance b has n_samples[b] samples, hence F_b = 1 + (n_samples[b] + 2*pad)/hop
 * frames computed as if it were alone (reflection at its own end, its own top_db maximum); rows F_b..F-1 are exactly 0.0,
 * the zero padding the reference's collate gives the mel. */
int lidk_logmel(const float* wav, const float* window, const float* twiddle, const float* melfb, float* out,
                float* utt_max, int B, int L, int pad, int hop, int n_mels, const int32_t* spans, int mask_times,
                float top_db, const int32_t* n_samples, const int32_t* mel_rng, const float* mel_coef, void* stream);
/* The same from the RAW waveform: normalize_wav, dither and pre-emphasis (lid/audio_processor.py:108-134) are applied in the
 * STFT's frame load from per-utterance partial sums, so the prepared waveforms never exist in memory: two launches
 * (statistics; STFT/mel/dB with the floor + masks applied by each utterance's last workgroup).  stats: scratch of B * 16 doubles.  noise [B][L] (may be NULL): explicit dither
 * values instead of the counter-based generator (what lidk_dither_preemph takes). */
int lidk_wav2mel(const float* wav, const float* window, const float* twiddle, const float* melfb, float* out, float* utt_max,
                 double* stats, int B, int L, int pad, int hop, int n_mels, const int32_t* spans, int mask_times, float top_db,
                 const int32_t* n_samples, float coef, float dither, uint64_t seed, const float* noise, const int32_t* mel_rng, const float* mel_coef, void* stream);

/* ------------------------------------------------------------------ generic element-wise helpers */
/* y = scale * x with dtype conversion (x_dtype/y_dtype in {LIDK_F32, LIDK_BF16}). */
int lidk_scale_cast(const void* x, int x_dtype, void* y, int y_dtype, long n, float scale, void* stream);
/* Strided 2-D variant: y[m][n] = scale*x[m][n], n < N; columns N..ldy of y are not written (zero padding stays). */
int lidk_scale_cast_2d(const void* x, int ldx, int x_dtype, void* y, int ldy, int y_dtype, int M, int N, float scale,
                       void* stream);
/* nn.Dropout (lid/conformer.py:491,590; lid/ConformerLangModel.py:349): y = x*keep/(1-p).  keep_in (uint8, optional)
 * supplies the mask (backward pass / parity tests); otherwise it is generated from (seed, index) and, if keep_out is
 * non-NULL, written there. */
int lidk_dropout(const void* x, int x_dtype, void* y, int y_dtype, const uint8_t* keep_in, uint8_t* keep_out, long n,
                 float p, uint64_t seed, void* stream);
/* out[i] = res[i] + dropout(x)[i] (f32), same decisions as lidk_dropout for the same (seed, index) / forced mask: the transformer
 * layers' dropout1 / dropout3 in front of the residual add (lid/wavlm/WavLM.py:745-771). */
int lidk_dropout_add(const float* x, const float* res, float* out, const uint8_t* keep_in, long n, float p, uint64_t seed,
                     void* stream);
/* dx = dy * (y > 0)   (backward of nn.ReLU in Conv1dSubSampling2, lid/conformer.py:331-333). */
int lidk_relu_bwd(const void* dy, const void* y, void* dx, long n, int dtype, void* stream);
/* out[n] += scale * sum_m x[m][n]   (bias gradients).  partial: >= LIDK_LN_PARTIAL_BLOCKS*N floats. */
int lidk_colsum(const void* x, int ldx, int x_dtype, float* out, float* partial, int M, int N, float scale, void* stream);
/* out [C][R] = in [R][C]^T */
int lidk_transpose(const void* in, int ldi, void* out, int ldo, int R, int C, int dtype, void* stream);
/* out[c] = sum_p partial[p][c] in float64 (deterministic column sum of scratch partials); out2 (may be NULL) gets a copy
 * (the SyncBatchNorm backward keeps the local sums while `out` is all-reduced in place).  tail > 0: out[ncols] (and
 * out2[ncols]) = tail - the BatchNorm kernels' row count, stored behind the sums so that ONE all-reduce(SUM) of ncols+1
 * doubles yields the global sums and the global count even when ranks hold different (B, T) shapes (what torch's
 * SyncBatchNorm gets by all-gathering per-rank counts, reference ccml/trainer.py:428). */
int lidk_reduce_partials_f64(const float* partial, int nparts, int ncols, double* out, double* out2, double tail,
                             void* stream);

/* ------------------------------------------------------------------ LayerNorm (nn.LayerNorm eps=1e-5, lid/conformer.py:85,190,250) */
/* x [M][C] f32 -> yT (T, may be NULL) and/or y32 (f32, may be NULL); mean/rstd [M] saved for backward (may be NULL). */
int lidk_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* yT, float* y32, float* mean,
                       float* rstd, int M, int C, float eps, int dtype, void* stream);
/* dx = dres + LN'(dy);  dxT = dxT_scale*dx (T, optional);  dgamma/dbeta += column sums.  dy is T or f32 (dy_dtype).
 * partial: >= LIDK_LN_BWD_BLOCKS*2*C floats.  With dgamma == dbeta == NULL only the partial rows are written and the
 * parameter gradients are finished later (off the critical path) by lidk_layernorm_param_grads on the same partial, M, C. */
int lidk_layernorm_bwd(const void* dy, int dy_dtype, const float* x, const float* mean, const float* rstd,
                       const float* gamma, const float* dres, float* dx, void* dxT, float dxT_scale, float* dgamma,
                       float* dbeta, float* partial, int M, int C, int dtype, void* stream);
int lidk_layernorm_param_grads(const float* partial, int M, int C, float* dgamma, float* dbeta, void* stream);
/* Same finaliser for `rows` partial rows of (dgamma | dbeta) [2*C] written by a kernel with its own grid (lidk_ffn_bwd). */
int lidk_layernorm_param_grads_rows(const float* partial, int rows, int C, float* dgamma, float* dbeta, void* stream);
/* n finalisers in one launch: descs = device array of n records {const float* partial; float* dgamma; float* dbeta; int rows; int C;}
 * (lidk_ln_param_grads_desc_bytes() bytes each, C equal in all records); sums are bit-identical to the single launches. */
int lidk_ln_param_grads_desc_bytes(void);
int lidk_layernorm_param_grads_grouped(const void* descs, int n, int C, void* stream);
/* Two LayerNorms in a row (C <= 256): post_norm of ConformerBlock i followed by the first FeedForward PreNorm of block i + 1
 * (lid/conformer.py:252-259, 153-171).  y1 = LN1(x) f32, y2 = LN2(y1) in T, statistics of both; one pass over x. */
int lidk_layernorm2_fwd(const float* x, const float* g1, const float* b1, float* y1, float* mean1, float* rstd1, const float* g2,
                        const float* b2, void* y2, float* mean2, float* rstd2, int M, int C, float eps, int dtype, void* stream);
/* Backward of that pair: dy (T) at y2, dres (f32) reaching y1 along the residual path -> dx (f32), dxT = dxT_scale * dx (T, may
 * be NULL); partial1 / partial2: LN1's / LN2's (dgamma | dbeta) rows as lidk_layernorm_bwd leaves them (finish with
 * lidk_layernorm_param_grads). */
int lidk_layernorm2_bwd(const void* dy, const float* dres, const float* y1, const float* mean2, const float* rstd2, const float* g2,
                        const float* x, const float* mean1, const float* rstd1, const float* g1, float* dx, void* dxT,
                        float dxT_scale, float* partial1, float* partial2, int M, int C, int dtype, void* stream);

/* ------------------------------------------------------------------ GEMM  C[M][N] = A[M][K] * B[N][K]^T  (+ fused epilogue)
 * Replaces every nn.Linear / 1x1 nn.Conv1d on the path (lid/conformer.py:98-100,163-166,192,199,334; lid/ConformerLangModel.py:350)
 * and their autograd (dgrad / wgrad run through the same kernel on transposed operands).
 * v = acc (+bias[n]);  act: SWISH -> out2 = v (pre-activation, T, optional), v = v*sigmoid(v);  RELU -> max(v,0);
 * GELU -> out2 = v (optional), v = gelu(v);  SWISH_GRAD / GELU_GRAD -> v *= act'(aux[m][n]);  v *= alpha;  v += res[m][n] (f32, optional);
 * out (T, or f32 when out_f32) = v; with splitk > 1 the K range is split over grid.z and out (f32) is accumulated atomically
 * (out must be pre-zeroed or hold the running gradient; bias/res/act must be unset).  K % 8 == 0, lda/ldb % 8 == 0.
 * lda may be SMALLER than K: rows of A then overlap, which is how a strided Conv1d over a channel-last [T][C] signal is a GEMM
 * without im2col (row t = the kW*C contiguous values starting at input row stride*t: lda = stride*C, K = kW*C; WavLM's
 * feature extractor, lid/wavlm/WavLM.py:409-531).  The caller guarantees (M-1)*lda + K elements are readable. */
typedef struct lidk_gemm_args {
  const void* A; const void* B;
  int M, N, K, lda, ldb;
  const float* bias;
  int act;
  float alpha;
  const float* res; int ldres;
  void* out; int ldo; int out_f32;
  void* out2; int ldo2;
  const void* aux; int ldaux;
  int splitk;
} lidk_gemm_args;
int lidk_gemm_nt(const lidk_gemm_args* args, int dtype, void* stream);
/* The conv module's second pointwise convolution run backwards (lid/conformer.py:197-199: BatchNorm1d -> Swish -> Conv1d(2d, d, 1)):
 * out = ds [M][N] bf16 = A [M][256] . B [N][256]^T, and in the same launch the BatchNorm + Swish backward statistics of that
 * output - dz = ds * swish'(xhat * gamma + beta), xhat = (c - mean) * rstd with c = g->aux [M][N] bf16 (the BatchNorm input) -
 * as per-channel partial sums: partial [(M / 64) * 2][2 N] f32 rows of (sum dz | sum dz * xhat); *nparts = the row count
 * (finish with lidk_reduce_partials_f64).  Replaces lidk_gemm_nt + lidk_bn_swish_bwd_reduce (a launch that re-reads ds).
 * K = 256, M % 64 == 0, N % 64 == 0, more than 1024 output tiles, bf16; otherwise LIDK_ERR_UNSUPPORTED. */
int lidk_gemm_nt_bn_sums(const lidk_gemm_args* g, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         float* partial, int* nparts, int dtype, void* stream);
/* Kernel-family knobs of lidk_gemm_nt, read from the environment once: "LIDK_GEMM_PIPEK" (tiles per workgroup of the K = 512 / 768 /
 * 1024 pipelined kernel, 0 = off), "LIDK_GEMM_DMA" (smallest K that takes the LDS-DMA 128 x 128 kernel, 0 = never),
 * "LIDK_GEMM_DMA_TILES" (its tile-count floor), "LIDK_GEMM_DMA256" (tile-count floor of the persistent 256-row LDS-DMA kernel; 0 = never,
 * 1 = whenever the shape allows), "LIDK_GEMM_DMA256_BN" (its tile width: 256 / 128, 0 = by the fill of the last round).  This call changes one afterwards (tests, micro-benchmarks); a negative value makes
 * the next launch re-read the environment. */
int lidk_gemm_option(const char* name, long value);

/* LayerNorm fused into the GEMM that consumes it (PreNorm -> Linear / 1x1 Conv1d: lid/conformer.py:81-89 with :163 ff
 * up-projection, :98-100 q/kv projections, :190-192 conv module) for K == 256 (the model width) and N % 256 == 0, bf16:
 *   out = epilogue( LN(x)[M][256] . B[N][256]^T ),  LN(x) = (x - mean) * rstd * gamma + beta per row, eps as nn.LayerNorm.
 * x [M][ldx] f32 is the residual stream; a workgroup normalises its 64-row panel once, keeps it in LDS and walks 4 column
 * chunks of B.  h [M][256] (T, optional), mean/rstd [M] (optional) receive what lidk_layernorm_fwd would have written (the
 * backward pass reads them).  args->A is ignored when x != NULL; with x == NULL the same row-panel kernel runs on the bf16
 * operand args->A (no LayerNorm).  Epilogues: none / bias / bias+Swish with out2 = pre-activation / SWISH_GRAD with aux;
 * alpha must be 1, res NULL, out of type T.  Returns LIDK_ERR_UNSUPPORTED for other shapes (lidk_ln_gemm_supported == 0):
 * callers then issue lidk_layernorm_fwd + lidk_gemm_nt. */
int lidk_ln_gemm_supported(int M, int N, int K, int dtype);
int lidk_ln_gemm_nt(const lidk_gemm_args* args, const float* x, int ldx, const float* gamma, const float* beta, float eps,
                    void* h, float* mean, float* rstd, int dtype, void* stream);

/* Whole FeedForward module in ONE launch (lid/conformer.py:153-171 FeedForward, its PreNorm :81-89, Scale(0.5) :247-248 and the
 * residual add of ConformerBlock.forward :252-259) for the model width d == 256, bf16:
 *     xo[M][256] (f32) = x + alpha * ( swish( LN(x) . W1[ff][256]^T + b1 ) . W2[256][ff]^T + b2 )
 * x [M][256] f32 residual stream; gamma/beta/eps: the PreNorm LayerNorm.  h_in (T, optional): LN(x) already produced by
 * lidk_layernorm2_fwd - then gamma/beta/mean/rstd are ignored and h is not rewritten.  Saved for the backward pass exactly as the
 * three-launch sequence (lidk_layernorm_fwd + 2 x lidk_gemm_nt) saves them: h [M][256] (T), mean/rstd [M], a [M][ff] (T, the
 * pre-activation), u [M][ff] (T, swish(a)); a, u, h may be NULL (inference).  A workgroup owns 64 complete rows, the hidden
 * activation of a row never leaves registers between the two projections, the weights stream through LDS by global_load_lds.
 * Returns LIDK_ERR_UNSUPPORTED when lidk_ffn_fwd_supported == 0 (callers then issue the three launches). */
int lidk_ffn_fwd_supported(int M, int d, int ff, int dtype);
int lidk_ffn_fwd(const float* x, const void* h_in, const float* gamma, const float* beta, float eps, const void* W1, const float* b1,
                 const void* W2, const float* b2, void* h, float* mean, float* rstd, void* a, void* u, float* xo, float alpha, int M,
                 int d, int ff, int dtype, void* stream);

/* lidk_ffn_fwd that also applies the LayerNorm(s) CONSUMING its output, in the epilogue (the workgroup owns complete rows):
 *   A: yA = LN(xo; gA, bA)  -> yA32 (f32, optional) and / or yAT (T, optional), meanA / rstdA [M]
 *   B (optional, gB != NULL): yB = LN(yA; gB, bB) -> yBT (T), meanB / rstdB [M]
 * i.e. the PreNorm of the module that follows (lid/conformer.py:81-89: attention after ff1), or post_norm followed by the next
 * block's first PreNorm (lid/conformer.py:252-259 then :153-171) - what lidk_layernorm_fwd / lidk_layernorm2_fwd would compute from
 * xo in one more launch each.  Same arithmetic (two-pass statistics, eps as the PreNorm's). */
int lidk_ffn_fwd_ln(const float* x, const void* h_in, const float* gamma, const float* beta, float eps, const void* W1, const float* b1,
                    const void* W2, const float* b2, void* h, float* mean, float* rstd, void* a, void* u, float* xo, float alpha,
                    const float* gA, const float* bA, float* yA32, void* yAT, float* meanA, float* rstdA, const float* gB,
                    const float* bB, void* yBT, float* meanB, float* rstdB, int M, int d, int ff, int dtype, void* stream);

/* Data path of the FeedForward backward in ONE launch (autograd of lid/conformer.py:153-171 + PreNorm :81-89), d == 256, bf16:
 *     da [M][ff] (T, stored)  = (dyT . W2) * swish'(a)        dyT [M][256] (T) = alpha * d(loss)/d(xo), a = saved pre-activation
 *     dh [M][256]             = da . W1
 *     dx (f32, optional) = dres (f32, optional) + LN'(dh; x, mean, rstd, gamma);  dxT (T, optional) = dxT_scale * dx
 * W2T [ff][256] / W1T [256][ff]: the transposed operand copies (row n of W2T = W2[:, n]; row j of W1T = W1[:, j]); ldw2t == 256,
 * ldw1t == ff.  partial: lidk_ffn_bwd_partial_rows(M) rows of (dgamma | dbeta) [2*256] f32, one per workgroup - finish them with
 * lidk_layernorm_param_grads_rows.  With dh != NULL the kernel stops after the second product and writes dh (T) instead of running
 * the LayerNorm backward (x .. partial are ignored): the site whose PreNorm backward is fused with the neighbouring block's
 * post_norm (lidk_layernorm2_bwd).  da feeds the weight gradients (lidk_gemm_tn: dW1 = da^T h, db1 = colsum da; dW2 = dyT^T u).
 * Replaces 2 x lidk_gemm_nt + lidk_layernorm_bwd; returns LIDK_ERR_UNSUPPORTED as lidk_ffn_fwd does. */
int lidk_ffn_bwd_partial_rows(int M);
/* torch.optim.Adam / torch.optim.SGD over a list of f32 tensors in one launch (the reference's optimizers for the wav2vec2 / WavLM
 * confs: lid/LidModule_ASR.py:143-150, lid/conf/xf_asr_wav2vec.yaml:25, xf_asr_extra_finetune.yaml:22).  chunks = device array of
 * n_chunks records {float* p; const float* g; float* m; float* v; int n; int pad;} (lidk_mt_chunk_bytes() bytes each), n <=
 * lidk_mt_chunk_elems() consecutive elements of one tensor: parameter, gradient, exp_avg / momentum buffer, exp_avg_sq (unused by
 * SGD; m == NULL: SGD without momentum).  Update rules of torch's single-tensor implementations; bias_correction{1,2} = 1 - beta^t. */
#define LIDK_MT_CHUNK 16384
int lidk_mt_chunk_bytes(void);
int lidk_mt_chunk_elems(void);
int lidk_adam_multi(const void* chunks, int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                    double bias_correction1, double bias_correction2, int maximize, void* stream);
int lidk_sgd_multi(const void* chunks, int n_chunks, float lr, float momentum, float dampening, float weight_decay, int nesterov,
                   int first_step, int maximize, void* stream);
/* Operand refresh of the transformer backbones after an optimizer step (speech-lid_amd/lidk/wavlm.py _refresh_inplace; the
 * reference keeps one f32 tensor per Linear and lets autograd / AMP cast it: lid/wavlm/modules.py, s3prl wav2vec2.py).  descs =
 * device array of n records {const float* src; bf16* dst; bf16* dstT; float* dst32; int R, C, lds, ldd, ldt, ld32, tiles_c, tile0;}
 * (lidk_cast_transpose_desc_bytes() bytes each): src [R][C] f32 is read once and written as bf16 (dst, may be NULL), as the
 * transposed bf16 matrix dstT [C][R] (may be NULL) and / or as f32 (dst32, may be NULL).  C % 4 == 0; R % 4 == 0 where dstT is
 * given; leading dimensions multiples of 4.  tiles_c = ceil(C / 64), tile0 = first of the record's ceil(R / 64) * tiles_c
 * workgroups; total_tiles = their sum over the n records (n small: a workgroup finds its record by scanning). */
int lidk_cast_transpose_desc_bytes(void);
int lidk_cast_transpose_grouped(const void* descs, int n, int total_tiles, void* stream);
/* Workgroup height of lidk_ffn_fwd* / lidk_ffn_bwd* / lidk_dgrad_ln_bwd: "LIDK_FFN_RG" = 3 (48 rows, 6 waves) or 4 (64 rows, 8 waves)
 * forces one form, 0 chooses by M (48 rows while ceil(M / 48) <= 256, one workgroup per CU in one round), negative re-reads the
 * environment variable of that name.  lidk_ffn_bwd_partial_rows follows the setting: query it after changing this. */
int lidk_ffn_option(const char* name, long value);
/* lidk_ffn_bwd with a SECOND LayerNorm backward behind the PreNorm's: the pair post_norm (input rows x1, mean1, rstd1, gamma1)
 * -> this module's PreNorm (lid/conformer.py:252-259 followed by :153-171 of the next block): dv = LN'(dh; x, mean, rstd, gamma) +
 * dres, dx = LN1'(dv; x1, mean1, rstd1, gamma1), dxT = dxT_scale * dx; partial / partial1: the two LayerNorms' (dgamma | dbeta) rows.
 * What lidk_ffn_bwd(dh form) + lidk_layernorm2_bwd compute in two launches. */
int lidk_ffn_bwd_ln2(const void* dyT, const void* a, const void* W2T, int ldw2t, const void* W1T, int ldw1t, void* da, const float* x,
                     const float* mean, const float* rstd, const float* gamma, const float* dres, const float* x1, const float* mean1,
                     const float* rstd1, const float* gamma1, float* dx, void* dxT, float dxT_scale, float* partial, float* partial1,
                     int M, int d, int ff, int dtype, void* stream);

int lidk_ffn_bwd(const void* dyT, const void* a, const void* W2T, int ldw2t, const void* W1T, int ldw1t, void* da, const float* x,
                 const float* mean, const float* rstd, const float* gamma, const float* dres, float* dx, void* dxT, float dxT_scale,
                 float* partial, void* dh, int M, int d, int ff, int dtype, void* stream);

/* Data gradient of a Linear / 1x1 Conv1d whose INPUT is a PreNorm output (N == 256 = the model width), fused with that LayerNorm's
 * backward (lid/conformer.py:81-89 with the q/kv projections :98-100 and the conv module's first pointwise conv :192), bf16:
 *     dh [M][256] = dy [M][K] . W [K][256]   (WT = W^T [256][ldwt == K], the transposed operand copy; K % 64 == 0)
 *     dx (f32, optional) = dres (f32, optional) + LN'(dh; x, mean, rstd, gamma);  dxT (T, optional) = dxT_scale * dx
 * partial: lidk_ffn_bwd_partial_rows(M) rows of (dgamma | dbeta) - finish with lidk_layernorm_param_grads_rows.  dh never exists in
 * HBM.  The kernel is the second half of lidk_ffn_bwd (same workgroup / LDS-DMA structure).  Replaces lidk_gemm_nt + lidk_layernorm_bwd. */
int lidk_dgrad_ln_bwd_supported(int M, int N, int K, int dtype);
int lidk_dgrad_ln_bwd(const void* dy, const void* WT, int ldwt, const float* x, const float* mean, const float* rstd,
                      const float* gamma, const float* dres, float* dx, void* dxT, float dxT_scale, float* partial, int M, int N,
                      int K, int dtype, void* stream);

/* Weight-gradient GEMM ("TN"): C[N1][N2] (f32) += alpha * sum_{m<M} X[m][n1] * Y[m][n2]; colsum[n1] (f32, optional) += alpha *
 * sum_m X[m][n1] (the bias gradient).  X [M][ldx], Y [M][ldy] are the activations as stored (T, row-major); rows are readable
 * (zero padded) up to the next multiple of 8 columns; ldx, ldy % 8 == 0.  The contraction over M is split `splitk` ways with
 * float atomics.  This is the autograd of every nn.Linear / 1x1 Conv1d weight on the path (dW = dY^T . X). */
int lidk_gemm_tn(const void* X, int ldx, const void* Y, int ldy, float* C, int ldc, float* colsum, int M, int N1, int N2,
                 float alpha, int splitk, int dtype, void* stream);
/* The same for SEVERAL weight gradients in ONE launch (bf16 operands): descs = n_desc device records of
 * lidk_gemm_tn_desc_bytes() bytes each: { const void* X; const void* Y; float* C; float* colsum; int ldx, ldy, ldc, M, N1, N2,
 * mchunk (rows per item, a multiple of 64), item0 (first item of the record), nsplit (= ceil(M / mchunk)), pad; float alpha;
 * int pad }.  Item i of a record = (row chunk, tile row, tile column) of its 64x64-tile decomposition; total_items = the sum
 * of tiles * nsplit; full != 0: every record has M, N1, N2 multiples of 64.  A ConformerBlock's eight Linear / 1x1-conv weight
 * gradients are one such launch (ccml/trainer.py:531 loss.backward() is where the reference forms them). */
int lidk_gemm_tn_desc_bytes(void);
int lidk_gemm_tn_grouped(const void* descs, int n_desc, int total_items, int full, void* stream);
/* The grouped launch on 128x128 output tiles: item0 / total_items count 128-tiles; N1 % 128 == N2 % 128 == M % 64 == 0 in every record. */
int lidk_gemm_tn_grouped128(const void* descs, int n_desc, int total_items, void* stream);
/* The same on 256x256 output tiles (8 waves, both operand tiles by LDS-DMA, two 64 KB stages): item0 / total_items count 256-tiles;
 * N1 % 256 == N2 % 256 == M % 64 == 0 in every record.  Half the operand bytes per FLOP of the 128-tile through L2. */
int lidk_gemm_tn_grouped256(const void* descs, int n_desc, int total_items, void* stream);

/* ------------------------------------------------------------------ Attention core with Shaw relative positions (lid/conformer.py:117-148)
 * qkv [B*T][3*heads*dh] (T): q | k | v column blocks, head h at columns h*dh.. within each.  rel_emb [2*max_pos+1][dh] f32.
 * scores = (q.k^T + q.rel_emb[clamp(i-j)+max_pos]) * dh^-0.5 ; probs = softmax_j ; out = probs.v -> [B*T][heads*dh] (T).
 * probs [B][heads][T][ldp] (T) is saved for backward. */
/* Sequence-length limits: the resident MFMA kernels take T <= 256 (K, V and the relative-embedding slice of a (batch, head) in LDS);
 * longer sequences (12 s training utterances give T = 600, 16.7 s validation utterances T = 835) run through key-tiled kernels of
 * the same arithmetic: for bf16 with rel_emb_T and dh in {32, 64} the MFMA kernels of csrc/attn_shaw.hip (probs and dscores are then
 * REQUIRED: the four kernels hand P and dS to each other through them), otherwise the VALU kernels of csrc/attn_long.hip.  lidk_attn_max_frames = the largest T those accept for a head dimension (16 score rows of T floats must fit in
 * LDS: 2,069 for bf16 / dh = 64 = 41 s of audio, 1,655 in f32 mode); beyond it lidk_attn_fwd / lidk_attn_bwd return LIDK_ERR_UNSUPPORTED. */
int lidk_attn_max_frames(int dh, int dtype);
int lidk_attn_fwd(const void* qkv, const float* rel_emb, const void* rel_emb_T, void* out, void* probs, int ldp, int B, int T,
                  int heads, int dh, int max_pos, int dtype, void* stream);
/* dqkv [B*T][3*heads*dh] (T) written; drel_emb += (atomic f32).  dscores: scratch of B*heads*T*ldp floats. */
int lidk_attn_bwd(const void* qkv, const float* rel_emb, const void* rel_emb_T, const void* probs, int ldp, const void* dout,
                  void* dqkv, float* drel_emb, float* dscores, int B, int T, int heads, int dh, int max_pos, int dtype,
                  void* stream);
/* Split form: lidk_attn_bwd with drel_emb == NULL leaves the relative-position embedding gradient out (allowed when
 * lidk_attn_bwd_relpos_supported: bf16, dh 32/64, T within the MFMA kernels' LDS budget) and lidk_attn_bwd_relpos adds it
 * later from the dS rows that call stored in `dscores`:  drel_emb[clamp(r)+max_pos][:] += scale * sum_i dS[i][i-r] q[i][:]. */
/* Shapes for which `probs` may be NULL in lidk_attn_fwd and lidk_attn_bwd (bf16, dh 32 / 64, T <= 256): the forward then stores
 * nothing T x T and the backward RECOMPUTES the probabilities from q, k and the relative embeddings (row kernel: softmax rows, dS,
 * dq, each row's log-sum-exp and delta; key-block kernel: dk, dv from recomputed tiles).  dscores keeps its size: the bf16 dS rows
 * fill its first half, the row statistics sit behind them. */
int lidk_attn_recompute_supported(int T, int dh, int dtype);
int lidk_attn_bwd_relpos_supported(int T, int dh, int dtype);
int lidk_attn_bwd_relpos(const void* qkv, const float* dscores, int ldp, float* drel_emb, int B, int T, int heads, int dh,
                         int max_pos, int dtype, void* stream);
/* Row stride (elements) the probs buffer must have for (T, dh, dtype): T rounded up to 32 when the MFMA kernels apply
 * (bf16, dh in {32,64}, T <= 256), else T.  probs is [B][heads][T][ldp]; rel_emb_T is the T-typed copy of rel_emb
 * (required by the MFMA kernels, may be NULL otherwise). */
int lidk_attn_ldp(int T, int dh, int dtype);
/* Device self-test of the ds_read_b64_tr_b16 lane mapping the transposed-operand kernels assume:
 * in [8][64] int16 -> out [64 lanes][2 halves][4] int16 as read by the documented addressing. */
int lidk_selftest_tr16(const void* in_8x64_i16, void* out_64x8_i16, void* stream);

/* ------------------------------------------------------------------ Conv module pieces (lid/conformer.py:47-65,174-205) */
/* GLU over channels: y [M][2C] -> g [M][C] = y[:, :C] * sigmoid(y[:, C:]) */
int lidk_glu_fwd(const void* y, void* g, int M, int C, int dtype, void* stream);
int lidk_glu_bwd(const void* y, const void* dg, void* dy, int M, int C, int dtype, void* stream);
/* DepthWiseConv1d, channel-last: c[b][t][ch] = bias[ch] + sum_k w[ch][k] * g[b][t+k-pad_left][ch], zero outside [0,T).
 * stat_partial (optional) [B*ceil(T/32)][2][C]: per-block (sum, sum of squares) of c for the BatchNorm batch statistics. */
int lidk_dwconv_fwd(const void* g, const float* w, const float* bias, void* c, float* stat_partial, int B, int T, int C,
                    int K, int pad_left, int dtype, void* stream);
/* The same forward with the GLU of lid/conformer.py:47-54 fused in front: y [B*T][2C] -> c; g (may be NULL) receives
 * a*sigmoid(gate) [B*T][C] for the weight gradient.  C % 4 == 0. */
int lidk_glu_dwconv_fwd(const void* y, const float* w, const float* bias, void* g, void* c, float* stat_partial, int B, int T,
                        int C, int K, int pad_left, int dtype, void* stream);
int lidk_dwconv_stat_parts(int B, int T, int C, int dtype);
/* Depthwise-conv weight gradient with the BatchNorm+Swish backward apply step fused into its operand load (bf16, C % 8 == 0:
 * lidk_dwconv_bwd_weight_bn_supported): equivalent to lidk_bn_swish_bwd_apply (-> dc, dgamma, dbeta) followed by
 * lidk_dwconv_bwd_weight(dc, g, ...), without dc in HBM.  Replaces the reference's autograd of
 * BatchNorm1d -> Swish after the depthwise Conv1d (lid/conformer.py:189-201) on the weight-gradient side. */
int lidk_dwconv_bwd_weight_bn_supported(int C, int dtype);
int lidk_dwconv_bwd_weight_bn(const void* ds, const void* c, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, const double* sums, const double* sums_local, double count, const void* g,
                              float* dw, float* db, float* dgamma, float* dbeta, float* partial, int B, int T, int C, int K,
                              int pad_left, int dtype, void* stream);
int lidk_dwconv_bwd_input(const void* dc, const float* w, void* dg, int B, int T, int C, int K, int pad_left, int dtype,
                          void* stream);
/* dw [C][K] += , db [C] += ;  partial: >= B*C*(K+1) floats */
/* Input gradient with the GLU backward fused behind it: dy [B*T][2C] = (dg*sigmoid(gate) | dg*a*sigmoid'(gate)), a/gate from y. */
int lidk_dwconv_bwd_input_glu(const void* dc, const float* w, const void* y, void* dy, int B, int T, int C, int K, int pad_left,
                              int dtype, void* stream);
/* The same with the BatchNorm+Swish backward "apply" step (lidk_bn_swish_bwd_apply) in front: the gradient dc at the conv
 * output is formed from ds, c and the (all-rank) sums while the tile is loaded and never stored.  C % 4 == 0. */
int lidk_dwconv_bwd_input_bn_glu(const void* ds, const void* c, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, const double* sums, double count, const float* w, const void* y, void* dy,
                                 int B, int T, int C, int K, int pad_left, int dtype, void* stream);
int lidk_dwconv_bwd_weight(const void* dc, const void* g, float* dw, float* db, float* partial, int B, int T, int C,
                           int K, int pad_left, int dtype, void* stream);
/* `count` convention of the three BatchNorm entry points below: count > 0 is the row count as a host scalar; count <= 0 means
 * "read it from device memory at sums[2*C]" (see lidk_reduce_partials_f64's tail) - no per-shape scalar in a captured launch.
 * BatchNorm1d training statistics (lid/conformer.py:197): sums [2][C] f64 = (sum x, sum x^2) over `count` rows (already
 * all-reduced over ranks for SyncBatchNorm, ccml/trainer.py:428) -> mean, rstd (biased var); running stats momentum update
 * with unbiased var; num_batches_tracked += 1. */
int lidk_bn_train_stats(const double* sums, double count, float* mean, float* rstd, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int C, void* stream);
/* The same statistics straight from the depthwise conv's partial rows (partial [nparts][2][C] f32, as lidk_glu_dwconv_fwd
 * writes them): column reduction in f64 + lidk_bn_train_stats in ONE launch.  Single-process training only - with data
 * parallelism the SyncBatchNorm all-reduce needs the reduced sums in between. */
int lidk_bn_train_stats_from_partials(const float* partial, int nparts, double count, float* mean, float* rstd,
                                      float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps, int C,
                                      void* stream);
int lidk_bn_eval_stats(const float* running_mean, const float* running_var, float* mean, float* rstd, float eps, int C,
                       void* stream);
/* s = swish(gamma*(c-mean)*rstd + beta) */
int lidk_bn_swish_fwd(const void* c, const float* mean, const float* rstd, const float* gamma, const float* beta,
                      void* s, int M, int C, int dtype, void* stream);
/* backward pass 1: partial [LIDK_BN_PARTIAL_BLOCKS][2][C] of (sum dz, sum dz*xhat), dz = ds*swish'(z) */
int lidk_bn_swish_bwd_reduce(const void* ds, const void* c, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, float* partial, int M, int C, int dtype, void* stream);
/* backward pass 2: dc = gamma*rstd*(dz - sums[0]/count - xhat*sums[1]/count); dgamma += sums_local[1]; dbeta += sums_local[0].
 * sums: all-rank totals [2][C] f64; sums_local: this rank's totals (== sums on one GPU). */
int lidk_bn_swish_bwd_apply(const void* ds, const void* c, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, const double* sums, const double* sums_local, double count, void* dc,
                            float* dgamma, float* dbeta, int M, int C, int dtype, void* stream);

/* ------------------------------------------------------------------ Conv1dSubSampling2 im2col (lid/conformer.py:328-348)
 * mel [B][F][C] f32 -> out [B*T][3*C] (T): row (b,t) = frames 2t-1, 2t, 2t+1 (zero outside), T = (F+2-3)/2+1. */
int lidk_im2col_k3s2(const float* mel, void* out, int B, int F, int C, int T, int dtype, void* stream);

/* ------------------------------------------------------------------ CTC loss + LID score
 * lid/LidModule_ASR_Supervised.py:162-168: CTCLoss(blank, reduction='none', zero_infinity)(log_softmax(logits).T, ...).
 * logits [B][T][V1] f32; targets [B][Lmax] int64; in_len/tg_len [B] int64.  loss [B] f32 (per utterance).
 * dlogits (optional) [B][T][V1] = grad_scale * d loss_b / d logits.  workspace: lidk_ctc_workspace_bytes(). */
long lidk_ctc_workspace_bytes(int B, int T, int V1, int Lmax);
int lidk_ctc_loss(const float* logits, const int64_t* targets, const int64_t* in_len, const int64_t* tg_len,
                  float* loss, float* dlogits, void* workspace, int B, int T, int V1, int Lmax, int blank,
                  float grad_scale, int zero_infinity, void* stream);
/* LangDiscriminator.forward ASR half (lid/ConformerLangModel.py:386-393) for ONE language head:
 * scores[b*score_stride] = sum_t [argmax!=blank] max_c log_softmax / (count*ln(blank) + 1e-5). */
int lidk_lid_score(const float* logits, float* scores, int score_stride, int B, int T, int V1, int blank, void* stream);
/* Greedy CTC decode on the device (CTCTokenizer.ctc_decode, lid/tokenizer.py:55-70): per frame argmax (ties -> lowest index), a
 * frame's symbol is kept iff it is not `blank` and differs from the previous frame's.  logits [B][T][V1] f32; in_len [B]
 * (int64, may be NULL = T frames); ids [B][T] int32 receives the kept symbols of utterance b at ids[b][0..out_len[b]). */
/* The training step's form of the same loss (lid/LidModule_ASR_Supervised.py:162-168 then ccml/trainer.py:521,531):
 * lidk_ctc_forward : in_len = (long)(T * wav_pct), tg_len = (long)(Lmax * txt_pct) (f32 products truncated, as the module's
 *                    `(out.shape[1] * wav_percents).long()`), per-utterance losses loss [B] and their mean loss_mean [1].
 * lidk_ctc_backward: dlogits [B*T][ld] of `dtype` (columns V1..ld-1 zero) = grad_scale * (*grad_scale_dev, if given) * d loss_b /
 *                    d logits from the lattices the forward left in `workspace` - written straight in the layout of the
 *                    vocabulary projection's gradient GEMMs, with the mean's 1/B and autograd's upstream scalar folded in.
 * LIDK_ERR_UNSUPPORTED when the lattice does not fit LDS (callers use lidk_ctc_loss). */
int lidk_ctc_forward(const float* logits, const int64_t* targets, const float* wav_pct, const float* txt_pct, int64_t* in_len,
                     int64_t* tg_len, float* loss, float* loss_mean, void* workspace, int B, int T, int V1, int Lmax, int blank,
                     int zero_infinity, void* stream);
int lidk_ctc_backward(const float* logits, const int64_t* targets, const int64_t* in_len, const int64_t* tg_len, void* dlogits,
                      int ld, int dtype, void* workspace, int B, int T, int V1, int Lmax, int blank, float grad_scale,
                      const float* grad_scale_dev, void* stream);
int lidk_ctc_greedy(const float* logits, const int64_t* in_len, int* ids, int* out_len, int B, int T, int V1, int blank,
                    void* stream);
/* LangDiscriminator.linear on the detached scores (lid/ConformerLangModel.py:374-378,394): out [B,C] =
 * W2 [C,H] relu(W0 [H,C] scores [B,C] + b0) + b2, H <= 64.  One wave per utterance: a row's result does not depend on the batch. */
int lidk_lid_mlp(const float* scores, const float* w0, const float* b0, const float* w2, const float* b2, float* out,
                 int B, int C, int H, void* stream);

/* ------------------------------------------------------------------ WavLM backbone, forward (lid/wavlm/WavLM.py,
 * lid/wavlm/modules.py).  Everything else of the backbone is lidk_gemm_nt (strided-view convolutions with lda < K, GELU epilogue),
 * lidk_layernorm_fwd and lidk_scale_cast_2d launches; see speech-lid_amd/lidk/wavlm.py for the sequence.
 * conv0: Conv1d(1, C, k10, s5, bias=False) + GroupNorm(C, C) + GELU (WavLM.py:430-463, layer 0 of the feature extractor),
 * wav [B][L] f32 -> out [B*P0][C] T=bf16, channel-last, P0 >= T0 = (L-10)/5+1 rows per utterance (rows T0.. are zero).
 * workspace: lidk_wavlm_conv0_workspace bytes. */
long lidk_wavlm_conv0_workspace(int B, int T0, int C);
int lidk_wavlm_conv0(const float* wav, int B, int L, const float* w, const float* gamma, const float* beta, float eps, void* out,
                     int T0, int P0, int C, float* workspace, void* stream);
/* Operand layout of the grouped positional convolution (WavLM.py:541-556: Conv1d(C, C, k, padding k/2, groups G) + SamePad + GELU):
 * x [B*T][C] f32 -> xg [G][rows_total][C/G] bf16, row (b, u) of a group = x[b][u - pad_left][group's channels], zero outside;
 * with it group g's convolution is lidk_gemm_nt(A = xg[g], lda = C/G, K = k*C/G).  rows_total >= B*Pp + k (read slack). */
int lidk_wavlm_posconv_prep(const float* x, void* xg, int B, int T, int C, int G, int Pp, int pad_left, long rows_total,
                            void* stream);
/* out[b][t] = x[b][t] + y[b*Pp + t]  (y has Pp rows per utterance, the first T valid): the pos-conv residual (WavLM.py:607). */
int lidk_wavlm_add_rows(const float* x, const float* y, float* out, int B, int T, int Pp, int C, void* stream);
/* Span masking of the projected features in training (WavLM.apply_mask, WavLM.py:300-337), in place on x [B*T][C] f32:
 * rows with time_mask [B][T] != 0 become mask_emb [C]; then columns with chan_mask [B][C] != 0 become 0.  Either mask may be NULL.
 * The spans themselves are drawn on the host in the reference's numpy draw order (lidk/wavlm.py span_mask). */
int lidk_wavlm_apply_mask(float* x, const unsigned char* time_mask, const unsigned char* chan_mask, const float* mask_emb, int B,
                          int T, int C, void* stream);
/* Gate of the relative position bias (modules.py:519-528): gate [B][H][T] from the layer input x [B*T][H*dh] f32,
 * grep_linear wg [8][dh], bg [8], grep_a [H]. */
int lidk_wavlm_gate(const float* x, const float* wg, const float* bg, const float* grep_a, float* gate, int B, int T, int H, int dh,
                    void* stream);
/* Self-attention core with the gated bucketed relative position bias (modules.py:505-560 -> F.multi_head_attention_forward):
 * out = softmax_j(q_i.k_j / sqrt(dh) + gate[b][h][i] * rb[h][j - i + RB - 1]) . v ; qkv [B*T][3*H*dh] bf16 (q | k | v blocks),
 * rb [H][2*RB-1] f32 (the head's bias as a function of the offset j - i, RB >= T), out [B*T][H*dh] bf16.  dh = 64, T <= 256. */
int lidk_wavlm_attn_max_frames(int dh);
/* Training path of the same attention.  lidk_wavlm_attn_probs stores the probabilities probs [B][H][T][ldp] bf16 (ldp =
 * lidk_wavlm_attn_ldp(T), pad columns zero) for the backward pass; lidk_wavlm_attn_bwd: dqkv [B*T][3*H*dh] bf16 (dq | dk | dv),
 * dgate [B][H][T] f32 (written), drb [H][2*RB-1] f32 (ACCUMULATED: the bias table's gradient, to be scattered into the bucket
 * embedding by the caller), dscores [B][H][T][T] f32 scratch.  Two-pass VALU form (row pass, column pass). */
int lidk_wavlm_attn_ldp(int T);
int lidk_wavlm_attn_probs(const void* qkv, const float* gate, const float* rb, void* probs, int B, int T, int H, int dh, int RB,
                          void* stream);
int lidk_wavlm_attn_bwd(const void* qkv, const void* probs, const void* dout, const float* gate, const float* rb, void* dqkv,
                        float* dgate, float* drb, float* dscores, int B, int T, int H, int dh, int RB, void* stream);
/* Backward of lidk_wavlm_gate: dx [B*T][H*dh] f32 += the gate's contribution to the layer-input gradient; dwg [8][dh], dbg [8],
 * dgrep_a [H] are ACCUMULATED. */
int lidk_wavlm_gate_bwd(const float* x, const float* wg, const float* bg, const float* grep_a, const float* dgate, float* dx,
                        float* dwg, float* dbg, float* dgrep_a, int B, int T, int H, int dh, void* stream);
/* Operand of the positional convolution's backward: dpc [rows_total][C] bf16, row (b, t < T) = dy[b][t] * gelu'(pre[b*Pp + t]),
 * zero in the pitch padding and the slack rows (so they add nothing to the weight gradient).  dpg (optional) [G][rows_total][C/G]
 * bf16 receives the same values group-major at row b*Pp + goff + t: the operand of the data gradient; rows it never writes must
 * be zero from allocation. */
int lidk_wavlm_posconv_dprep(const float* dy, const void* pre, void* dpc, void* dpg, int B, int T, int Pp, int C, int G, int goff,
                             long rows_total, void* stream);
int lidk_wavlm_attn_fwd(const void* qkv, const float* gate, const float* rb, void* out, int B, int T, int H, int dh, int RB,
                        void* stream);
/* Training form of lidk_wavlm_attn_fwd: also writes probs [B][H][T][ldp] bf16 (ldp = lidk_wavlm_attn_ldp(T)) for the backward pass. */
int lidk_wavlm_attn_fwd_probs(const void* qkv, const float* gate, const float* rb, void* out, void* probs, int B, int T, int H, int dh,
                              int RB, void* stream);
/* Gradients of the gated relative bias from dS [B][H][T][ldp] bf16 as lidk_attn_bwd (drel_emb = NULL, zero relative-position
 * table) leaves it: dgate [B][H][T] written, drb [H][2*RB-1] accumulated.  With lidk_attn_bwd for dQ/dK/dV this is the autograd of
 * lid/wavlm/modules.py:505-560 on the MFMA path. */
int lidk_wavlm_attn_bias_grads(const void* ds, const float* gate, const float* rb, float* dgate, float* drb, int B, int T, int H,
                               int RB, int ldp, void* stream);

/* ------------------------------------------------------------------ backward of the conv feature extractor
 * (lid/wavlm/WavLM.py:409-531, un-frozen by lid/WavLMMutiLangModel.py:86-94).  Layers 1-6 are the forward's strided-view GEMMs
 * run backwards (lidk_gemm_tn / lidk_gemm_nt) plus these layout kernels; layer 0 is recomputed from the waveform.
 *   conv_dlast : dpre [B*P][C] bf16 = dsrc [B*T][C] f32 * gelu'(pre) for t < T (pre NULL: no factor), zero in the pitch padding rows
 *   conv_col2im: dprev [B*2P][C] bf16, row (b, t < Tprev) = (window terms of dcol [B*P][kW*C] bf16 that touch input row t:
 *                k2 s2 one term, k3 s2 two on even rows) * gelu'(pre [B*2P][C]) (pre NULL: no factor), zero for t >= Tprev
 *   conv0_bwd  : dy0 [B*P0][C] bf16 at layer 0's output, stats [B][C][2] (mean, rstd) from lidk_wavlm_conv0's workspace
 *                (offset B*ceil(T0/128)*C*2 floats); dw [C][10], dgamma [C], dbeta [C] ACCUMULATED; sums: scratch B*C*2 floats. */
int lidk_wavlm_conv_dlast(const float* dsrc, const void* pre, void* dpre, int B, int T, int P, int C, void* stream);
int lidk_wavlm_conv_col2im(const void* dcol, const void* pre, void* dprev, int B, int P, int T, int Tprev, int kW, int C,
                           void* stream);
int lidk_wavlm_conv0_bwd(const float* wav, int B, int L, const float* w, const float* gamma, const float* beta, const float* stats,
                         const void* dy0, float* sums, float* dw, float* dgamma, float* dbeta, int T0, int P0, int C, void* stream);

/* ------------------------------------------------------------------ wav2vec2 Large / XLS-R and WavLM Large: the "layer_norm" feature
 * extractor and waveform normalisation (the checkpoints the reference's wav2vec confs load: lid/conf/xf_asr_wav2vec.yaml:12,
 * lid/conf/xf_asr_extra_finetune.yaml:12).
 * lidk_wav_layernorm: task.normalize, F.layer_norm(wav, wav.shape) per utterance (lid/s3prl_updream/wav2vec/wav2vec2_expert.py:71-72):
 *   (x - mean) / sqrt(var_biased + eps) over the utterance's own n_samples[b] samples (NULL: all L), zeros behind them.
 * ConvFeatureExtractionModel(mode="layer_norm", conv_bias) (lid/wavlm/WavLM.py:415-477 = wav2vec2.py:769-848): every layer is
 * Conv1d(+bias) -> LayerNorm over the C channels of a time step -> GELU.  Channel-last activations make that a row LayerNorm:
 *   lidk_conv0_ln_fwd : layer 0 (k10 s5 on the raw waveform) with its LayerNorm + GELU, wav [B][L] f32 -> out [B*P0][C] bf16
 *                       (rows T0..P0-1 of an utterance zero); bias may be NULL (conv_bias=False).  C = 512.
 *   lidk_ln_gelu_fwd  : layers 1..: out = gelu(LayerNorm(pre)) on rows of C = 512 (pre = lidk_gemm_nt over the strided view, bias in
 *                       its epilogue); pre / out [rows][C] of `dtype`.
 *   lidk_ln_gelu_bwd  : dy at the GELU output -> dpre (rows t >= Tv of each utterance's P rows: zero); dgamma / dbeta ACCUMULATED.
 *   lidk_conv0_ln_bwd : layer 0 backward, recomputed from the waveform; dw [C][10], dbias [C] (NULL without conv bias), dgamma,
 *                       dbeta ACCUMULATED.
 * LIDK_ERR_UNSUPPORTED for C != 512. */
int lidk_wav_layernorm(const float* wav, float* out, int B, int L, const int32_t* n_samples, float eps, void* stream);
int lidk_conv0_ln_fwd(const float* wav, int B, int L, const float* w, const float* bias, const float* gamma, const float* beta,
                      float eps, void* out, int T0, int P0, int C, void* stream);
int lidk_ln_gelu_fwd(const void* pre, const float* gamma, const float* beta, void* out, long rows, int C, float eps, int dtype,
                     void* stream);
int lidk_ln_gelu_bwd(const void* dy, const void* pre, const float* gamma, const float* beta, void* dpre, float* dgamma, float* dbeta,
                     int B, int P, int Tv, int C, float eps, int dtype, void* stream);
int lidk_conv0_ln_bwd(const float* wav, int B, int L, const float* w, const float* bias, const float* gamma, const float* beta,
                      float eps, const void* dy0, float* dw, float* dbias, float* dgamma, float* dbeta, int T0, int P0, int C,
                      void* stream);

/* ------------------------------------------------------------------ wav2vec2 pieces
 * lidk_zero_padded_rows: x [B][T][C] f32, rows t >= klen[b] become zero - TransformerEncoder.extract_features zeroes padded frames
 * before the positional convolution (lid/s3prl_updream/wav2vec/wav2vec2.py:906-907); the backward zeroes the same gradient rows.
 * lidk_hidden_mix_*: s3prl Featurizer._weighted_sum (lid/s3prl_updream/interfaces.py:227-252), feature = sum_l softmax(w)[l] * h_l
 * over the n_states = L + 1 hidden states (w [n_states] f32, the trainable `model.featurizer.weights`):
 *   axpy : out[n] = (overwrite ? 0 : out[n]) + softmax(w)[l] * h[n]      (forward accumulation; backward dh_l += sm[l] * dfeat)
 *   dot  : *dot += <a, b>                                                  (dots[l] = <dfeat, h_l>)
 *   wgrad: dw[l] += sm[l] * (dots[l] - sum_k sm[k] * dots[k])              (softmax backward, one launch) */
int lidk_zero_padded_rows(float* x, const int* klen, int B, int T, int C, void* stream);
int lidk_hidden_mix_axpy(const float* h, const float* w, int n_states, int l, float* out, long n, int overwrite, void* stream);
int lidk_hidden_mix_dot(const float* a, const float* b, float* dot, long n, void* stream);
int lidk_hidden_mix_wgrad(const float* w, const float* dots, float* dw, int n_states, void* stream);

/* ------------------------------------------------------------------ key-tiled attention of the transformer backbones
 * Replaces F.multi_head_attention_forward as lid/wavlm/modules.py:505-700 calls it (WavLM: additive gated relative-position bias,
 * attention dropout) and fairseq's MultiheadAttention as lid/s3prl_updream/wav2vec/wav2vec2.py:1009-1078 calls it (wav2vec2:
 * key_padding_mask, attention dropout).  qkv [B*T][3*H*dh] bf16 (q | k | v blocks), out [B*T][H*dh] bf16, dh = 64, any
 * T <= lidk_xattn_max_frames(dh):
 *   S[i][j] = q_i.k_j / sqrt(dh) + gate[b][h][i] * rb[h][j - i + RB - 1]   (gate [B][H][T], rb [H][2*RB-1] f32, RB >= T; both NULL:
 *             no bias);  keys j >= klen[b] are masked out (klen [B] int32, NULL: none; padding is a suffix);
 *   A = softmax_j S;  with drop_p > 0 or a forced mask `keep` [B][H][T][T] u8: Ad = A * keep / (1 - drop_p), the decision for
 *   element (b, h, i, j) being keep[...] != 0 or uniform(seed, ((b*H + h)*T + i)*T + j) >= drop_p;  out = Ad . v.
 * No T x T tensor reaches HBM: lse [B][H][T] f32 (log-sum-exp of each score row) is all the backward needs besides out.
 * lidk_xattn_bwd: dqkv [B*T][3*H*dh] bf16 (dq | dk | dv) is written; delta [B][H][T] f32 is scratch; dgate [B][H][T] f32 is
 * written and drb [H][2*RB-1] f32 ACCUMULATED (either may be NULL; both ignored without a bias). */
int lidk_xattn_max_frames(int dh);
int lidk_xattn_fwd(const void* qkv, const float* gate, const float* rb, const int* klen, void* out, float* lse,
                   const unsigned char* keep, float drop_p, unsigned long long seed, int B, int T, int H, int dh, int RB,
                   void* stream);
int lidk_xattn_bwd(const void* qkv, const float* gate, const float* rb, const int* klen, const void* out, const void* dout,
                   const float* lse, const unsigned char* keep, float drop_p, unsigned long long seed, void* dqkv, float* delta,
                   float* dgate, float* drb, int B, int T, int H, int dh, int RB, void* stream);

/* ------------------------------------------------------------------ fused clip + Novograd over the flat arenas
 * ccml/trainer.py:541-543 clip_grad_norm_(max_norm) + ccml/optim/novograd.py:75-145 (amsgrad=False, luc=False).
 * work [n_work][3] int64 = (tensor id, element offset into the flat arenas, length <= LIDK_OPT_CHUNK); items of one tensor
 * are contiguous and tensors appear in ascending id order.  Only tensors present in `work` are touched ("grad is None" for
 * the rest).  exp_avg_sq [n_tensors]; scratch >= n_work + n_tensors + 8 floats.  total_norm_out [1] (pre-clip norm).
 * The gradients of the touched tensors are consumed: they are ZERO on return (what optimizer.zero_grad() would do next). */
#define LIDK_OPT_CHUNK 8192
int lidk_novograd_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const int64_t* work, int n_work,
                       int n_tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int grad_averaging, float max_norm, float* scratch, float* total_norm_out, void* stream);
/* Refresh the T-typed GEMM operands from the f32 master parameters in ONE launch.  mats [n_mats][8] int64 (DEVICE) =
 * (src offset in `params`, rows R, cols C, dst offset of W [R][C] in wT or -1, dst offset of W^T [C][ldt] in wT or -1,
 * ldt >= R (pad columns are not written), index of the matrix's first 32x32 tile, tiles per tile-row = ceil(C/32));
 * rows sorted by first-tile index; total_tiles = sum over matrices of ceil(R/32)*ceil(C/32). */
int lidk_cast_weights(const float* params, void* wT, const int64_t* mats, int n_mats, long total_tiles, int dtype,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif
