"""Tensor-level wrappers over the C-ABI (one Python function per include/lidk.h entry point).

torch is used only for device memory and the current HIP stream.  Every wrapper refuses CPU tensors: there is no
fallback path.  Outputs are passed in (the engine preallocates them); a few allocate when ``out`` is None for tests.
"""
import ctypes as C
import math

import torch

from . import _lib as L
from ._lib import LidkError, check, dtype_code, lib


IS_HIP_BACKEND = True      # marks the real kernel backend (the engine refuses CPU tensors when this is set)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise LidkError(f"lidk op got a tensor on {t.device}: the HIP path has no CPU fallback")
    if not t.is_contiguous():
        raise LidkError("lidk op needs contiguous tensors")
    return t.data_ptr()


def _pv(t):
    """Like _p but accepts row-strided 2-D views (unit inner stride)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LidkError(f"lidk op got a tensor on {t.device}: the HIP path has no CPU fallback")
    if t.stride(-1) != 1:
        raise LidkError("lidk op needs unit inner stride")
    return t.data_ptr()


def _code(t):
    return dtype_code(t.dtype)


# ----------------------------------------------------------------------------------------------- small host -> device uploads
_STAGE = {}


def upload_async(values, dtype, device, key):
    """A small host list / array (per-utterance lengths, table indices) -> a new device tensor WITHOUT stalling the host:
    torch.tensor(list, device=cuda) copies from pageable memory, which blocks until everything queued on the stream has drained
    (one full host-device sync per call: 20 ms per step on ragged wav2vec2 batches).  Here the values go through one of two pinned
    staging buffers per ``key``; a buffer is reused two calls later, after its copy's event has passed."""
    src = torch.as_tensor(values, dtype=dtype).reshape(-1)
    n = src.numel()
    k = (key, str(device), dtype)
    st = _STAGE.get(k)
    if st is None or st["cap"] < n:
        cap = max(64, 1 << max(n - 1, 1).bit_length())
        st = _STAGE[k] = dict(cap=cap, bufs=[torch.empty(cap, dtype=dtype).pin_memory() for _ in range(2)], events=[None, None], turn=0)
    i = st["turn"]
    st["turn"] = 1 - i
    if st["events"][i] is not None:
        st["events"][i].synchronize()
    st["bufs"][i][:n].copy_(src)
    out = torch.empty(n, dtype=dtype, device=device)
    out.copy_(st["bufs"][i][:n], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st["events"][i] = ev
    return out


# ----------------------------------------------------------------------------------------------- features
_TABLES = {}


def fft_tables(device, win_length=400, n_mels=80):
    """window [512] (hann(win) periodic, centred), twiddle [256][2], melfb [257][n_mels] — host-computed once."""
    key = (str(device), win_length, n_mels)
    if key not in _TABLES:
        win = torch.zeros(512)
        left = (512 - win_length) // 2
        win[left:left + win_length] = torch.hann_window(win_length)
        k = torch.arange(256, dtype=torch.float64) * (2.0 * math.pi / 512.0)
        tw = torch.stack([torch.cos(k), torch.sin(k)], dim=1).float()
        fb = melscale_fbanks(n_mels).contiguous()
        # compact form of the (triangular, hence banded) filterbank: first / last non-zero bin of every filter and its taps
        rng, coef = torch.zeros(2, 128, dtype=torch.int32), torch.zeros(128, 32)
        rng[0].fill_(257)
        rng[1].fill_(-1)
        compact = n_mels <= 128
        for m in range(min(n_mels, 128)):
            nz = torch.nonzero(fb[:, m]).flatten()
            if nz.numel():
                lo, hi = int(nz[0]), int(nz[-1])
                rng[0, m], rng[1, m] = lo, hi
                if hi - lo < 32:
                    coef[m, :hi - lo + 1] = fb[lo:hi + 1, m]
                else:
                    compact = False
        _TABLES[key] = (win.to(device), tw.contiguous().to(device), fb.to(device),
                        rng.contiguous().to(device) if compact else None, coef.contiguous().to(device) if compact else None)
    return _TABLES[key]


def melscale_fbanks(n_mels=80, n_freqs=257, f_min=0.0, f_max=8000.0, sample_rate=16000):
    """HTK mel filterbank, norm=None — what torchaudio.transforms.MelSpectrogram builds by default for the call at
    lid/audio_processor.py:91-101 (sample_rate is torchaudio's default 16000 whatever `sr` is: SURVEY Q1)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def normalize_wav(wav, out=None, n_samples=None):
    """n_samples: optional int32 (B,) true lengths of a ragged, zero-padded batch."""
    out = torch.empty_like(wav) if out is None else out
    B, Lw = wav.shape
    check(lib().lidk_normalize_wav(_p(wav), _p(out), B, Lw, _p(n_samples), _stream()), "normalize_wav")
    return out


def dither_preemph(wav, coef=0.97, dither=1e-5, seed=0, noise=None, out=None):
    out = torch.empty_like(wav) if out is None else out
    B, Lw = wav.shape
    check(lib().lidk_dither_preemph(_p(wav), _p(out), _p(noise), B, Lw, coef, dither, seed, _stream()), "dither_preemph")
    return out


_RESAMPLE = {}


def resample_taps(p: int, q: int, zeros: int = 16, beta: float = 8.6):
    """FIR rows of the polyphase resampler y[n] = x(n*p/q): (taps float64 [q][ntaps], left).  Row r holds the weights of
    x[base - left + j], base = (n*p)//q, for the outputs whose phase (n*p) % q is r: a Kaiser-windowed sinc with cut-off
    min(1, q/p) of the input Nyquist and `zeros` zero crossings on each side, each row normalised to unit DC gain."""
    import numpy as np
    if p == q:
        return np.ones((1, 1)), 0
    fc = min(1.0, q / p)
    half = zeros / fc
    left = int(math.ceil(half))
    ntaps = 2 * left + 1
    j = np.arange(ntaps, dtype=np.float64)[None, :]
    r = np.arange(q, dtype=np.float64)[:, None]
    t = (j - left) - r / q                                   # tap position minus the (fractional) output position
    w = np.where(np.abs(t) <= half, np.i0(beta * np.sqrt(np.clip(1.0 - (t / half) ** 2, 0.0, 1.0))) / np.i0(beta), 0.0)
    h = fc * np.sinc(fc * t) * w
    return h / h.sum(1, keepdims=True), left


def speed_out_len(n: int, p: int, q: int) -> int:
    return int(n * q / p + 0.5)


def speed_perturb(wav, factors, n_samples=None):
    """wav (B, L) f32 on the GPU, factors: per-utterance (p, q) with speed v = p/q (e.g. (11, 10), (1, 1), (9, 10)) ->
    (out (B, Lout) f32, n_out (B,) int32 on the GPU, list of output lengths).  n_samples: optional int32 (B,) true lengths."""
    import numpy as np
    B, Lin = wav.shape
    dev = wav.device
    host_n = n_samples.tolist() if n_samples is not None else [Lin] * B
    out_lens = [speed_out_len(n, p, q) for n, (p, q) in zip(host_n, factors)]
    Lout = max(out_lens)
    uniq = sorted(set(factors))
    recs = []
    for (p, q) in uniq:
        key = (str(dev), p, q)
        if key not in _RESAMPLE:
            taps, left = resample_taps(p, q)
            _RESAMPLE[key] = (torch.from_numpy(taps.astype(np.float32)).contiguous().to(dev), left)
        t, left = _RESAMPLE[key]
        recs.append([t.data_ptr(), (p & 0xFFFFFFFF) | (q << 32), (t.shape[1] & 0xFFFFFFFF) | (left << 32)])
    # {ptr; int p, q; int ntaps, left} = 24 bytes per record; all three through pinned staging (no host stall per augmented batch)
    tables = upload_async(recs, torch.int64, dev, "speed.tables").view(len(recs), 3)
    table_of = upload_async([uniq.index(f) for f in factors], torch.int32, dev, "speed.table_of")
    n_out = upload_async(out_lens, torch.int32, dev, "speed.n_out")
    out = torch.empty(B, Lout, device=dev, dtype=torch.float32)
    check(lib().lidk_speed_perturb(_p(wav), B, Lin, _p(n_samples), _p(out), Lout, _p(n_out), _p(tables), len(uniq), _p(table_of),
                                   _stream()), "speed_perturb")
    return out, n_out, out_lens


def logmel(wav, pad=0, hop=160, win_length=400, n_mels=80, spans=None, top_db=80.0, out=None, n_samples=None):
    """wav (B, L) f32 -> (B, F, n_mels) f32 dB with per-utterance top_db floor and optional SpecAugment spans
    (int32 (B, mask_times, 4)).  n_samples: optional int32 (B,) true lengths; rows behind an utterance's own frames are 0."""
    B, Lw = wav.shape
    F_ = 1 + (Lw + 2 * pad) // hop
    win, tw, fb, mrng, mcoef = fft_tables(wav.device, win_length, n_mels)
    out = torch.empty(B, F_, n_mels, device=wav.device, dtype=torch.float32) if out is None else out
    umax = torch.empty(2 * B, device=wav.device, dtype=torch.float32)      # dB maxima + one int ticket per utterance
    mt = 0 if spans is None else spans.shape[1]
    check(lib().lidk_logmel(_p(wav), _p(win), _p(tw), _p(fb), _p(out), _p(umax), B, Lw, pad, hop, n_mels, _p(spans), mt,
                            top_db, _p(n_samples), _p(mrng), _p(mcoef), _stream()), "logmel")
    return out


def wav2mel(wav, pad=0, hop=160, win_length=400, n_mels=80, spans=None, top_db=80.0, out=None, n_samples=None, coef=0.97,
            dither=1e-5, seed=0, noise=None):
    """RAW wav (B, L) f32 -> (B, F, n_mels) f32 dB: normalize_wav + dither / pre-emphasis + logmel in three launches (the
    waveform preparation happens in the STFT's frame load)."""
    B, Lw = wav.shape
    F_ = 1 + (Lw + 2 * pad) // hop
    win, tw, fb, mrng, mcoef = fft_tables(wav.device, win_length, n_mels)
    out = torch.empty(B, F_, n_mels, device=wav.device, dtype=torch.float32) if out is None else out
    umax = torch.empty(2 * B, device=wav.device, dtype=torch.float32)      # dB maxima + one int ticket per utterance
    stats = torch.empty(B * 16, device=wav.device, dtype=torch.float64)
    mt = 0 if spans is None else spans.shape[1]
    check(lib().lidk_wav2mel(_p(wav), _p(win), _p(tw), _p(fb), _p(out), _p(umax), _p(stats), B, Lw, pad, hop, n_mels, _p(spans), mt,
                             top_db, _p(n_samples), coef, dither, seed, _p(noise), _p(mrng), _p(mcoef), _stream()), "wav2mel")
    return out


# ----------------------------------------------------------------------------------------------- element-wise
def scale_cast(x, out, scale=1.0):
    check(lib().lidk_scale_cast(_p(x), _code(x), _p(out), _code(out), x.numel(), scale, _stream()), "scale_cast")
    return out


def scale_cast_2d(x, out, M, N, scale=1.0):
    """out[:M, :N] = scale * x[:M, :N] for row-strided 2-D tensors (pad columns of out untouched)."""
    check(lib().lidk_scale_cast_2d(_pv(x), x.stride(0), _code(x), _pv(out), out.stride(0), _code(out), M, N, scale,
                                   _stream()), "scale_cast_2d")
    return out


def dropout(x, out, p, seed=0, keep_in=None, keep_out=None):
    check(lib().lidk_dropout(_p(x), _code(x), _p(out), _code(out), _p(keep_in), _p(keep_out), x.numel(), p, seed,
                             _stream()), "dropout")
    return out


def dropout_add(x, res, out, p, seed=0, keep_in=None):
    """out = res + dropout(x) (f32), decisions as ``dropout`` for the same seed / forced mask."""
    check(lib().lidk_dropout_add(_p(x), _p(res), _p(out), _p(keep_in), x.numel(), float(p), int(seed), _stream()), "dropout_add")
    return out


def relu_bwd(dy, y, dx):
    check(lib().lidk_relu_bwd(_p(dy), _p(y), _p(dx), dy.numel(), _code(dy), _stream()), "relu_bwd")
    return dx


def colsum(x, out, partial, scale=1.0):
    """out (N,) f32 += scale * x.sum(0); x (M, N)."""
    M, N = x.shape
    check(lib().lidk_colsum(_pv(x), x.stride(0), _code(x), _p(out), _p(partial), M, N, scale, _stream()), "colsum")
    return out


def transpose(x, out):
    R, Cc = x.shape
    check(lib().lidk_transpose(_pv(x), x.stride(0), _pv(out), out.stride(0), R, Cc, _code(x), _stream()), "transpose")
    return out


def reduce_partials_f64(partial, nparts, ncols, out, out2=None, tail=0.0):
    """tail > 0 is stored at out[ncols] (and out2[ncols]): the row count behind the BatchNorm sums (one all-reduce for both)."""
    check(lib().lidk_reduce_partials_f64(_p(partial), nparts, ncols, _p(out), _p(out2), float(tail), _stream()),
          "reduce_partials_f64")
    return out


# ----------------------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, gamma, beta, yT=None, y32=None, mean=None, rstd=None, eps=1e-5, dtype=None):
    M, Cc = x.shape
    code = _code(yT) if yT is not None else (dtype_code(dtype) if dtype is not None else L.F32)
    check(lib().lidk_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(yT), _p(y32), _p(mean), _p(rstd), M, Cc, eps, code,
                                   _stream()), "layernorm_fwd")


def layernorm_bwd(dy, x, mean, rstd, gamma, partial, dres=None, dx=None, dxT=None, dxT_scale=1.0, dgamma=None,
                  dbeta=None, dtype=None):
    M, Cc = x.shape
    code = _code(dxT) if dxT is not None else (dtype_code(dtype) if dtype is not None else _code(dy))
    check(lib().lidk_layernorm_bwd(_p(dy), _code(dy), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dxT),
                                   dxT_scale, _p(dgamma), _p(dbeta), _p(partial), M, Cc, code, _stream()), "layernorm_bwd")


def layernorm2_fwd(x, g1, b1, y1, mean1, rstd1, g2, b2, y2, mean2, rstd2, eps=1e-5):
    """y1 = LN1(x) (f32), y2 = LN2(y1) (bf16 / f32 by y2's dtype) in one pass; C <= 256."""
    M, Cc = x.shape
    check(lib().lidk_layernorm2_fwd(_p(x), _p(g1), _p(b1), _p(y1), _p(mean1), _p(rstd1), _p(g2), _p(b2), _p(y2), _p(mean2),
                                    _p(rstd2), M, Cc, eps, _code(y2), _stream()), "layernorm2_fwd")


def layernorm2_bwd(dy, dres, y1, mean2, rstd2, g2, x, mean1, rstd1, g1, dx, dxT, dxT_scale, partial1, partial2):
    M, Cc = x.shape
    check(lib().lidk_layernorm2_bwd(_p(dy), _p(dres), _p(y1), _p(mean2), _p(rstd2), _p(g2), _p(x), _p(mean1), _p(rstd1), _p(g1),
                                    _p(dx), _p(dxT), dxT_scale, _p(partial1), _p(partial2), M, Cc, _code(dy), _stream()),
          "layernorm2_bwd")


def layernorm_param_grads(partial, M, C, dgamma, dbeta):
    """Finish dgamma/dbeta from the partial rows a ``layernorm_bwd(..., dgamma=None, dbeta=None)`` call left behind."""
    check(lib().lidk_layernorm_param_grads(_p(partial), M, C, _p(dgamma), _p(dbeta), _stream()), "layernorm_param_grads")


# ----------------------------------------------------------------------------------------------- GEMM
def gemm_nt(A, B, out, bias=None, act=L.ACT_NONE, alpha=1.0, res=None, out2=None, aux=None, splitk=1, M=None, N=None,
            K=None):
    """out[M,N] = epilogue(A[M,K] @ B[N,K]^T).  A, B share the activation dtype; out is that dtype or f32."""
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = B.shape[0] if N is None else N
    g = L.GemmArgs()
    g.A, g.B = _pv(A), _pv(B)
    g.M, g.N, g.K, g.lda, g.ldb = M, N, K, A.stride(0), B.stride(0)
    g.bias, g.act, g.alpha = _p(bias), act, alpha
    g.res, g.ldres = _pv(res), (res.stride(0) if res is not None else 0)
    g.out, g.ldo, g.out_f32 = _pv(out), out.stride(0), int(out.dtype == torch.float32)
    g.out2, g.ldo2 = _pv(out2), (out2.stride(0) if out2 is not None else 0)
    g.aux, g.ldaux = _pv(aux), (aux.stride(0) if aux is not None else 0)
    g.splitk = splitk
    if A.dtype != B.dtype:
        raise LidkError("gemm_nt: A and B must share a dtype")
    check(lib().lidk_gemm_nt(C.byref(g), _code(A), _stream()), "gemm_nt")
    return out


def gemm_option(name: str, value: int):
    """Change one of lidk_gemm_nt's kernel-family knobs (LIDK_GEMM_PIPEK / LIDK_GEMM_DMA / LIDK_GEMM_DMA_TILES) after start-up;
    a negative value re-reads the environment."""
    check(lib().lidk_gemm_option(name.encode(), int(value)), "gemm_option")


def gemm_nt_bn_sums(A, B, out, c, mean, rstd, gamma, beta, partial, M=None, N=None, K=None):
    """out = A @ B^T (bf16) with the BatchNorm + Swish backward sums of the output in the epilogue (c: the BatchNorm input).
    -> number of partial rows written (finish with reduce_partials_f64), or 0 when the shape is outside the fused kernel."""
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = B.shape[0] if N is None else N
    g = L.GemmArgs()
    g.A, g.B = _pv(A), _pv(B)
    g.M, g.N, g.K, g.lda, g.ldb = M, N, K, A.stride(0), B.stride(0)
    g.bias, g.act, g.alpha = None, L.ACT_NONE, 1.0
    g.res, g.ldres = None, 0
    g.out, g.ldo, g.out_f32 = _pv(out), out.stride(0), 0
    g.out2, g.ldo2 = None, 0
    g.aux, g.ldaux = _pv(c), c.stride(0)
    g.splitk = 1
    if A.dtype != torch.bfloat16 or B.dtype != A.dtype or out.dtype != A.dtype or c.dtype != A.dtype:
        return 0
    n = C.c_int(0)
    rc = lib().lidk_gemm_nt_bn_sums(C.byref(g), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(partial), C.byref(n), _code(A), _stream())
    if rc == L.ERR_UNSUPPORTED:
        return 0
    check(rc, "gemm_nt_bn_sums")
    return n.value


def ln_gemm_supported(M, N, K, dtype):
    return bool(lib().lidk_ln_gemm_supported(M, N, K, dtype_code(dtype)))


def ln_gemm_nt(x, gamma, beta, B, out, h=None, mean=None, rstd=None, bias=None, act=L.ACT_NONE, out2=None, aux=None, A=None,
               eps=1e-5):
    """out[M,N] = epilogue(LN(x)[M,256] @ B[N,256]^T) with the LayerNorm fused into the operand load (x f32 residual stream;
    h / mean / rstd get what layernorm_fwd would have written).  x=None, A=bf16 operand: the same row-panel kernel without LN."""
    src = x if x is not None else A
    M, K = src.shape
    N = B.shape[0]
    g = L.GemmArgs()
    g.A, g.B = (_pv(A) if A is not None else None), _pv(B)
    g.M, g.N, g.K, g.lda, g.ldb = M, N, K, (A.stride(0) if A is not None else 0), B.stride(0)
    g.bias, g.act, g.alpha = _p(bias), act, 1.0
    g.res, g.ldres = None, 0
    g.out, g.ldo, g.out_f32 = _pv(out), out.stride(0), 0
    g.out2, g.ldo2 = _pv(out2), (out2.stride(0) if out2 is not None else 0)
    g.aux, g.ldaux = _pv(aux), (aux.stride(0) if aux is not None else 0)
    g.splitk = 1
    check(lib().lidk_ln_gemm_nt(C.byref(g), _pv(x), (x.stride(0) if x is not None else 0), _p(gamma), _p(beta), eps, _p(h),
                                _p(mean), _p(rstd), _code(B), _stream()), "ln_gemm_nt")
    return out


def ffn_fwd_supported(M, d, ff, dtype):
    return bool(lib().lidk_ffn_fwd_supported(M, d, ff, dtype_code(dtype)))


def ffn_fwd(x, W1, b1, W2, b2, xo, gamma=None, beta=None, h_in=None, h=None, mean=None, rstd=None, a=None, u=None, alpha=0.5,
            eps=1e-5, next_ln=None):
    """xo = x + alpha * (swish(LN(x) W1^T + b1) W2^T + b2) in one launch (d = 256, bf16 operands); see lidk_ffn_fwd.
    next_ln: dict(gA, bA, meanA, rstdA, yA32 and / or yAT[, gB, bB, yBT, meanB, rstdB]) - the LayerNorm(s) consuming xo, applied in
    the epilogue (lidk_ffn_fwd_ln)."""
    M, d = x.shape
    ff = W1.shape[0]
    if next_ln is None:
        check(lib().lidk_ffn_fwd(_p(x), _pv(h_in), _p(gamma), _p(beta), eps, _pv(W1), _p(b1), _pv(W2), _p(b2), _pv(h), _p(mean),
                                 _p(rstd), _pv(a), _pv(u), _p(xo), alpha, M, d, ff, _code(W1), _stream()), "ffn_fwd")
        return xo
    n = next_ln
    check(lib().lidk_ffn_fwd_ln(_p(x), _pv(h_in), _p(gamma), _p(beta), eps, _pv(W1), _p(b1), _pv(W2), _p(b2), _pv(h), _p(mean),
                                _p(rstd), _pv(a), _pv(u), _p(xo), alpha, _p(n["gA"]), _p(n["bA"]), _p(n.get("yA32")), _p(n.get("yAT")),
                                _p(n["meanA"]), _p(n["rstdA"]), _p(n.get("gB")), _p(n.get("bB")), _p(n.get("yBT")), _p(n.get("meanB")),
                                _p(n.get("rstdB")), M, d, ff, _code(W1), _stream()), "ffn_fwd_ln")
    return xo


def ffn_bwd_partial_rows(M):
    return lib().lidk_ffn_bwd_partial_rows(M)


def ffn_option(name: str, value: int):
    """LIDK_FFN_RG: 3 / 4 = 48- / 64-row workgroups in the fused FeedForward kernels, 0 = by M, negative = re-read the environment."""
    check(lib().lidk_ffn_option(name.encode(), int(value)), "ffn_option")


def ffn_bwd(dyT, a, W1T, W2T, da, x=None, mean=None, rstd=None, gamma=None, dres=None, dx=None, dxT=None, dxT_scale=1.0,
            partial=None, dh=None, pair=None):
    """da = (dyT W2) * swish'(a); dh = da W1; then the PreNorm backward (dx / dxT / partial dgamma-dbeta rows), or dh itself when
    ``dh`` is given; see lidk_ffn_bwd.  W1T [256, ff], W2T [ff, 256]: the transposed operand copies."""
    M, d = dyT.shape
    ff = a.shape[1]
    if pair is not None:            # dict(x1, mean1, rstd1, gamma1, partial1): the LayerNorm in FRONT of this PreNorm (post_norm)
        check(lib().lidk_ffn_bwd_ln2(_p(dyT), _p(a), _pv(W2T), W2T.stride(0), _pv(W1T), W1T.stride(0), _p(da), _p(x), _p(mean), _p(rstd),
                                     _p(gamma), _p(dres), _p(pair["x1"]), _p(pair["mean1"]), _p(pair["rstd1"]), _p(pair["gamma1"]),
                                     _p(dx), _p(dxT), dxT_scale, _p(partial), _p(pair["partial1"]), M, d, ff, _code(dyT), _stream()),
              "ffn_bwd_ln2")
        return
    check(lib().lidk_ffn_bwd(_p(dyT), _p(a), _pv(W2T), W2T.stride(0), _pv(W1T), W1T.stride(0), _p(da), _p(x), _p(mean), _p(rstd),
                             _p(gamma), _p(dres), _p(dx), _p(dxT), dxT_scale, _p(partial), _p(dh), M, d, ff, _code(dyT), _stream()),
          "ffn_bwd")


def dgrad_ln_bwd_supported(M, N, K, dtype):
    return bool(lib().lidk_dgrad_ln_bwd_supported(M, N, K, dtype_code(dtype)))


def dgrad_ln_bwd(dy, WT, x, mean, rstd, gamma, partial, dres=None, dx=None, dxT=None, dxT_scale=1.0):
    """dh = dy @ W (WT = W^T [256, K]); dx = dres + LN'(dh), dxT = dxT_scale * dx; partial (dgamma | dbeta) rows; see lidk_dgrad_ln_bwd."""
    M, K = dy.shape
    check(lib().lidk_dgrad_ln_bwd(_p(dy), _pv(WT), WT.stride(0), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dxT),
                                  dxT_scale, _p(partial), M, WT.shape[0], K, _code(dy), _stream()), "dgrad_ln_bwd")


def layernorm_param_grads_rows(partial, rows, C, dgamma, dbeta):
    check(lib().lidk_layernorm_param_grads_rows(_p(partial), rows, C, _p(dgamma), _p(dbeta), _stream()), "layernorm_param_grads_rows")


class _LnPgDesc(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("rows", C.c_int), ("C", C.c_int)]


def build_ln_param_group(entries):
    """Descriptor table for layernorm_param_grads_grouped.  entries: list of (partial, rows, C, dgamma, dbeta); raw pointers inside."""
    if lib().lidk_ln_param_grads_desc_bytes() != C.sizeof(_LnPgDesc):
        raise LidkError("LnPgDesc layout mismatch between ops.py and liblidk.so")
    arr = (_LnPgDesc * len(entries))()
    for i, (partial, rows, Cn, dg, db) in enumerate(entries):
        arr[i].partial, arr[i].dgamma, arr[i].dbeta, arr[i].rows, arr[i].C = _p(partial), _p(dg), _p(db), rows, Cn
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(entries[0][0].device), len(entries), entries[0][2]


def layernorm_param_grads_grouped(group):
    table, n, Cn = group
    check(lib().lidk_layernorm_param_grads_grouped(_p(table), n, Cn, _stream()), "layernorm_param_grads_grouped")


class _CtDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("dstT", C.c_void_p), ("dst32", C.c_void_p), ("R", C.c_int), ("C", C.c_int),
                ("lds", C.c_int), ("ldd", C.c_int), ("ldt", C.c_int), ("ld32", C.c_int), ("tiles_c", C.c_int), ("tile0", C.c_int)]


def build_cast_transpose_group(entries):
    """Descriptor table for cast_transpose_grouped.  entries: list of (src f32 [R, C], dst bf16 [R, C] | None, dstT bf16 [C, R] |
    None, dst32 f32 [R, C] | None); views with a unit inner stride; raw pointers inside - rebuild when a tensor moves."""
    if lib().lidk_cast_transpose_desc_bytes() != C.sizeof(_CtDesc):
        raise LidkError("CtDesc layout mismatch between ops.py and liblidk.so")
    arr = (_CtDesc * len(entries))()
    tile0 = 0
    for i, (src, dst, dstT, dst32) in enumerate(entries):
        src2 = src if src.dim() == 2 else src.view(1, -1)
        R, Cn = src2.shape
        bad = Cn % 4 or src2.stride(1) != 1 or src2.dtype != torch.float32 or (R > 1 and src2.stride(0) % 4)
        for t, shape, dt in ((dst, (R, Cn), torch.bfloat16), (dstT, (Cn, R), torch.bfloat16), (dst32, (R, Cn), torch.float32)):
            if t is not None:
                t2 = t if t.dim() == 2 else t.view(1, -1)
                bad = bad or tuple(t2.shape) != shape or t2.dtype != dt or t2.stride(1) != 1 or (t2.shape[0] > 1 and t2.stride(0) % 4)
        if bad or (dstT is not None and R % 4):
            raise LidkError(f"cast_transpose_grouped: record {i} has an unsupported shape / stride / dtype")
        ld = lambda t: 0 if t is None else (t.stride(0) if t.dim() == 2 else t.numel())
        a = arr[i]
        a.src, a.dst, a.dstT, a.dst32 = _pv(src2), _pv(dst), _pv(dstT), _pv(dst32)
        a.R, a.C, a.lds, a.ldd, a.ldt, a.ld32 = R, Cn, ld(src2), ld(dst), ld(dstT), ld(dst32)
        a.tiles_c, a.tile0 = -(-Cn // 64), tile0
        tile0 += -(-R // 64) * a.tiles_c
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(entries[0][0].device), len(entries), tile0


def cast_transpose_grouped(group):
    table, n, tiles = group
    check(lib().lidk_cast_transpose_grouped(_p(table), n, tiles, _stream()), "cast_transpose_grouped")


def layernorm_bwd_partial_rows(M):
    """Partial rows lidk_layernorm_bwd / lidk_layernorm2_bwd write for M rows (what lidk_layernorm_param_grads assumes)."""
    return min(-(-M // 4), L.LN_BWD_BLOCKS)


def gemm_tn(X, Y, C, colsum=None, alpha=1.0, splitk=1, M=None, N1=None, N2=None):
    """C[N1,N2] (f32) += alpha * X[:M,:N1]^T @ Y[:M,:N2]; colsum[N1] (f32, optional) += alpha * X.sum(0)."""
    M = X.shape[0] if M is None else M
    N1 = X.shape[1] if N1 is None else N1
    N2 = Y.shape[1] if N2 is None else N2
    if X.dtype != Y.dtype or C.dtype != torch.float32:
        raise LidkError("gemm_tn: X and Y share the activation dtype, C is float32")
    check(lib().lidk_gemm_tn(_pv(X), X.stride(0), _pv(Y), Y.stride(0), _pv(C), C.stride(0), _p(colsum), M, N1, N2, alpha,
                             splitk, _code(X), _stream()), "gemm_tn")
    return C


# ----------------------------------------------------------------------------------------------- attention
class _TnDesc(C.Structure):
    _fields_ = [("X", C.c_void_p), ("Y", C.c_void_p), ("C", C.c_void_p), ("colsum", C.c_void_p),
                ("ldx", C.c_int), ("ldy", C.c_int), ("ldc", C.c_int), ("M", C.c_int), ("N1", C.c_int), ("N2", C.c_int),
                ("mchunk", C.c_int), ("item0", C.c_int), ("nsplit", C.c_int), ("pad0", C.c_int), ("alpha", C.c_float), ("pad1", C.c_int)]


def build_tn_group(entries, split=4, tile=64):
    """Descriptor table of a grouped weight-gradient launch.  entries: list of (X (M, N1) bf16, Y (M, N2) bf16, C (N1, N2) f32,
    colsum (N1,) f32 or None, M, N1, N2): C += X[:M, :N1]^T @ Y[:M, :N2].  split: row chunks per output tile (one int, or one per
    entry); tile: 64, 128 or 256 (output tile edge; 128 / 256 need full shapes).  -> (device uint8 tensor, n, total_items, full or
    tile); the table holds raw pointers: rebuild it when any buffer is reallocated."""
    if lib().lidk_gemm_tn_desc_bytes() != C.sizeof(_TnDesc):
        raise LidkError("TnDesc layout mismatch between ops.py and liblidk.so")
    arr, item, full, dev = (_TnDesc * len(entries))(), 0, True, None
    splits = list(split) if isinstance(split, (list, tuple)) else [split] * len(entries)       # row chunks per entry
    for i, (X, Y, Cm, cs, M, N1, N2) in enumerate(entries):
        split = splits[i]
        if X.dtype != torch.bfloat16 or Y.dtype != torch.bfloat16 or Cm.dtype != torch.float32:
            raise LidkError("build_tn_group: bf16 operands, f32 output")
        if X.stride(-1) != 1 or Y.stride(-1) != 1 or Cm.stride(-1) != 1 or (X.stride(0) & 7) or (Y.stride(0) & 7):
            raise LidkError("build_tn_group: unit inner strides and row pitches that are multiples of 8")
        mchunk = -(-(-(-M // split)) // 64) * 64
        nsplit = -(-M // mchunk)
        d = arr[i]
        d.X, d.Y, d.C, d.colsum = X.data_ptr(), Y.data_ptr(), Cm.data_ptr(), (cs.data_ptr() if cs is not None else None)
        d.ldx, d.ldy, d.ldc, d.M, d.N1, d.N2 = X.stride(0), Y.stride(0), Cm.stride(0), M, N1, N2
        d.mchunk, d.item0, d.nsplit, d.alpha = mchunk, item, nsplit, 1.0
        item += (-(-N1 // tile)) * (-(-N2 // tile)) * nsplit
        full = full and not (M % 64 or N1 % tile or N2 % tile)
        dev = X.device
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    if tile != 64 and not full:
        raise LidkError(f"build_tn_group: {tile}-tiles need N1 % {tile} == N2 % {tile} == M % 64 == 0")
    return host.to(dev), len(entries), item, (full if tile == 64 else tile)


def gemm_tn_grouped(group):
    table, n, items, full = group
    if full == 128:
        check(lib().lidk_gemm_tn_grouped128(_p(table), n, items, _stream()), "gemm_tn_grouped128")
        return
    if full == 256:
        check(lib().lidk_gemm_tn_grouped256(_p(table), n, items, _stream()), "gemm_tn_grouped256")
        return
    check(lib().lidk_gemm_tn_grouped(_p(table), n, items, int(full), _stream()), "gemm_tn_grouped")


def attn_ldp(T, dh, dtype):
    """Row stride of the probs buffer for this (T, dh, dtype): padded to 32 when the MFMA kernels apply."""
    return lib().lidk_attn_ldp(T, dh, dtype_code(dtype))


def attn_max_frames(dh, dtype):
    """Largest sequence length T (after subsampling) the attention kernels accept for this head dimension."""
    return lib().lidk_attn_max_frames(dh, dtype_code(dtype))


def attn_recompute_supported(T, dh, dtype):
    """True when attn_fwd / attn_bwd accept probs=None for this shape: the backward recomputes the probabilities."""
    return bool(lib().lidk_attn_recompute_supported(T, dh, dtype_code(dtype)))


def attn_fwd(qkv, rel_emb, out, probs, B, T, heads, dh, rel_emb_T=None):
    """probs (B, heads, T, ldp) receives the softmax rows; None (attn_recompute_supported shapes): nothing T x T is stored."""
    max_pos = (rel_emb.shape[0] - 1) // 2
    ldp = probs.shape[-1] if probs is not None else attn_ldp(T, dh, qkv.dtype)
    check(lib().lidk_attn_fwd(_p(qkv), _p(rel_emb), _p(rel_emb_T), _p(out), _p(probs), ldp, B, T, heads, dh,
                              max_pos, _code(qkv), _stream()), "attn_fwd")


def attn_bwd(qkv, rel_emb, probs, dout, dqkv, drel_emb, dscores, B, T, heads, dh, rel_emb_T=None):
    """probs None: recompute path (the forward was run with probs=None); dscores then also carries the row statistics."""
    max_pos = (rel_emb.shape[0] - 1) // 2
    ldp = probs.shape[-1] if probs is not None else attn_ldp(T, dh, qkv.dtype)
    check(lib().lidk_attn_bwd(_p(qkv), _p(rel_emb), _p(rel_emb_T), _p(probs), ldp, _p(dout), _p(dqkv),
                              _p(drel_emb), _p(dscores), B, T, heads, dh, max_pos, _code(qkv), _stream()), "attn_bwd")


def attn_bwd_relpos_supported(T, dh, dtype):
    return bool(lib().lidk_attn_bwd_relpos_supported(T, dh, dtype_code(dtype)))


def attn_bwd_relpos(qkv, dscores, ldp, drel_emb, B, T, heads, dh):
    """Relative-position embedding gradient from the dS rows an ``attn_bwd(..., drel_emb=None, ...)`` call left in dscores."""
    max_pos = (drel_emb.shape[0] - 1) // 2
    check(lib().lidk_attn_bwd_relpos(_p(qkv), _p(dscores), ldp, _p(drel_emb), B, T, heads, dh, max_pos, _code(qkv), _stream()),
          "attn_bwd_relpos")


def selftest_tr16(inp, out):
    check(lib().lidk_selftest_tr16(_p(inp), _p(out), _stream()), "selftest_tr16")


# ----------------------------------------------------------------------------------------------- conv module
def glu_fwd(y, g):
    M, C2 = y.shape
    check(lib().lidk_glu_fwd(_p(y), _p(g), M, C2 // 2, _code(y), _stream()), "glu_fwd")


def glu_bwd(y, dg, dy):
    M, C2 = y.shape
    check(lib().lidk_glu_bwd(_p(y), _p(dg), _p(dy), M, C2 // 2, _code(y), _stream()), "glu_bwd")


def dwconv_stat_parts(B, T, C, dtype):
    """Rows of the (parts, 2, C) BatchNorm partial-sum buffer glu_dwconv_fwd / dwconv_fwd fill for this shape and dtype."""
    return lib().lidk_dwconv_stat_parts(B, T, C, dtype_code(dtype))


def dwconv_fwd(g, w, bias, c, stat_partial, B, T, pad_left):
    Cc, K = w.shape
    check(lib().lidk_dwconv_fwd(_p(g), _p(w), _p(bias), _p(c), _p(stat_partial), B, T, Cc, K, pad_left, _code(g),
                                _stream()), "dwconv_fwd")


def dwconv_bwd_input(dc, w, dg, B, T, pad_left):
    Cc, K = w.shape
    check(lib().lidk_dwconv_bwd_input(_p(dc), _p(w), _p(dg), B, T, Cc, K, pad_left, _code(dc), _stream()), "dwconv_bwd_input")


def glu_dwconv_fwd(y, w, bias, g, c, stat_partial, B, T, pad_left):
    """GLU + depthwise conv in one launch: y [M, 2C] -> c [M, C]; g [M, C] (optional) keeps the GLU output for the wgrad."""
    Cc, K = w.shape
    check(lib().lidk_glu_dwconv_fwd(_p(y), _p(w), _p(bias), _p(g), _p(c), _p(stat_partial), B, T, Cc, K, pad_left, _code(y),
                                    _stream()), "glu_dwconv_fwd")


def dwconv_bwd_input_glu(dc, w, y, dy, B, T, pad_left):
    """Depthwise-conv input gradient + GLU backward in one launch: dc [M, C], y [M, 2C] -> dy [M, 2C]."""
    Cc, K = w.shape
    check(lib().lidk_dwconv_bwd_input_glu(_p(dc), _p(w), _p(y), _p(dy), B, T, Cc, K, pad_left, _code(dc), _stream()),
          "dwconv_bwd_input_glu")


def dwconv_bwd_input_bn_glu(ds, c, mean, rstd, gamma, beta, sums, count, w, y, dy, B, T, pad_left):
    """BatchNorm+Swish backward (apply step) + depthwise-conv input gradient + GLU backward in one launch."""
    Cc, K = w.shape
    check(lib().lidk_dwconv_bwd_input_bn_glu(_p(ds), _p(c), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(sums), float(count),
                                             _p(w), _p(y), _p(dy), B, T, Cc, K, pad_left, _code(ds), _stream()),
          "dwconv_bwd_input_bn_glu")


def dwconv_bwd_weight(dc, g, dw, db, partial, B, T, pad_left):
    Cc, K = dw.shape
    check(lib().lidk_dwconv_bwd_weight(_p(dc), _p(g), _p(dw), _p(db), _p(partial), B, T, Cc, K, pad_left, _code(dc),
                                       _stream()), "dwconv_bwd_weight")


def dwconv_bwd_weight_bn_supported(C, dtype):
    return bool(lib().lidk_dwconv_bwd_weight_bn_supported(C, dtype_code(dtype)))


def dwconv_bwd_weight_bn(ds, c, mean, rstd, gamma, beta, sums, sums_local, count, g, dw, db, dgamma, dbeta, partial, B, T, pad_left):
    """bn_swish_bwd_apply + dwconv_bwd_weight in one pass over ds / c / g (dc is never written)."""
    Cc, K = dw.shape
    check(lib().lidk_dwconv_bwd_weight_bn(_p(ds), _p(c), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(sums), _p(sums_local),
                                          float(count), _p(g), _p(dw), _p(db), _p(dgamma), _p(dbeta), _p(partial), B, T, Cc, K,
                                          pad_left, _code(ds), _stream()), "dwconv_bwd_weight_bn")


def bn_train_stats(sums, count, mean, rstd, running_mean, running_var, nbt, momentum=0.1, eps=1e-5):
    Cc = mean.shape[0]
    check(lib().lidk_bn_train_stats(_p(sums), float(count), _p(mean), _p(rstd), _p(running_mean), _p(running_var), _p(nbt),
                                    momentum, eps, Cc, _stream()), "bn_train_stats")


def bn_train_stats_from_partials(partial, nparts, count, mean, rstd, running_mean, running_var, nbt, momentum=0.1, eps=1e-5):
    """Partial rows (nparts, 2, C) f32 -> batch statistics + running statistics in one launch (no data parallelism)."""
    Cc = mean.shape[0]
    check(lib().lidk_bn_train_stats_from_partials(_p(partial), nparts, float(count), _p(mean), _p(rstd), _p(running_mean),
                                                  _p(running_var), _p(nbt), momentum, eps, Cc, _stream()),
          "bn_train_stats_from_partials")


def bn_eval_stats(running_mean, running_var, mean, rstd, eps=1e-5):
    check(lib().lidk_bn_eval_stats(_p(running_mean), _p(running_var), _p(mean), _p(rstd), eps, mean.shape[0], _stream()),
          "bn_eval_stats")


def bn_swish_fwd(c, mean, rstd, gamma, beta, s):
    M, Cc = c.shape
    check(lib().lidk_bn_swish_fwd(_p(c), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(s), M, Cc, _code(c), _stream()),
          "bn_swish_fwd")


def bn_swish_bwd_reduce(ds, c, mean, rstd, gamma, beta, partial):
    M, Cc = c.shape
    check(lib().lidk_bn_swish_bwd_reduce(_p(ds), _p(c), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(partial), M, Cc,
                                         _code(c), _stream()), "bn_swish_bwd_reduce")


def bn_swish_bwd_apply(ds, c, mean, rstd, gamma, beta, sums, sums_local, count, dc, dgamma, dbeta):
    M, Cc = c.shape
    check(lib().lidk_bn_swish_bwd_apply(_p(ds), _p(c), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(sums), _p(sums_local),
                                        float(count), _p(dc), _p(dgamma), _p(dbeta), M, Cc, _code(c), _stream()),
          "bn_swish_bwd_apply")


def im2col_k3s2(mel, out, T):
    B, F_, Cc = mel.shape
    check(lib().lidk_im2col_k3s2(_p(mel), _p(out), B, F_, Cc, T, _code(out), _stream()), "im2col_k3s2")


# ----------------------------------------------------------------------------------------------- loss
def ctc_workspace_bytes(B, T, V1, Lmax):
    return lib().lidk_ctc_workspace_bytes(B, T, V1, Lmax)


def ctc_loss(logits, targets, in_len, tg_len, loss, dlogits, workspace, blank, grad_scale=1.0, zero_infinity=True):
    B, T, V1 = logits.shape
    Lmax = targets.shape[1]
    check(lib().lidk_ctc_loss(_p(logits), _p(targets), _p(in_len), _p(tg_len), _p(loss), _p(dlogits), _p(workspace), B, T,
                              V1, Lmax, blank, grad_scale, int(zero_infinity), _stream()), "ctc_loss")


def ctc_forward(logits, targets, wav_pct, txt_pct, in_len, tg_len, loss, loss_mean, workspace, blank, zero_infinity=True):
    """Lengths from the batch's percents, per-utterance losses and their mean; the lattices stay in ``workspace`` for
    ``ctc_backward``.  Returns False when the fast path does not apply (the caller then uses ``ctc_loss``)."""
    B, T, V1 = logits.shape
    rc = lib().lidk_ctc_forward(_p(logits), _p(targets), _p(wav_pct), _p(txt_pct), _p(in_len), _p(tg_len), _p(loss), _p(loss_mean),
                                _p(workspace), B, T, V1, targets.shape[1], blank, int(zero_infinity), _stream())
    if rc == L.ERR_UNSUPPORTED:
        return False
    check(rc, "ctc_forward")
    return True


def ctc_backward(logits, targets, in_len, tg_len, dlogits, workspace, blank, grad_scale=1.0, grad_scale_dev=None):
    """dlogits (B*T, ld) bf16 / f32, pad columns zeroed: grad_scale * grad_scale_dev[0] * d loss_b / d logits."""
    B, T, V1 = logits.shape
    check(lib().lidk_ctc_backward(_p(logits), _p(targets), _p(in_len), _p(tg_len), _p(dlogits), dlogits.stride(0), _code(dlogits),
                                  _p(workspace), B, T, V1, targets.shape[1], blank, grad_scale, _p(grad_scale_dev), _stream()),
          "ctc_backward")


def lid_score(logits, scores_col, stride, blank):
    """scores_col: 1-element-offset view into a (B, n_lang) f32 matrix column; stride = n_lang."""
    B, T, V1 = logits.shape
    if not scores_col.is_cuda:
        raise LidkError("lid_score: CPU tensor")
    check(lib().lidk_lid_score(_p(logits), scores_col.data_ptr(), stride, B, T, V1, blank, _stream()), "lid_score")


def ctc_greedy(logits, in_len, blank):
    """Greedy CTC collapse on the device: logits (B, T, V1) f32, in_len (B,) int64 or None ->
    (ids (B, T) int32, lengths (B,) int32): the kept symbols of utterance b are ids[b, :lengths[b]]."""
    B, T, V1 = logits.shape
    ids = torch.empty(B, T, device=logits.device, dtype=torch.int32)
    lens = torch.empty(B, device=logits.device, dtype=torch.int32)
    check(lib().lidk_ctc_greedy(_p(logits), _p(in_len), _p(ids), _p(lens), B, T, V1, blank, _stream()), "ctc_greedy")
    return ids, lens


def lid_mlp(scores, w0, b0, w2, b2, out):
    """LangDiscriminator MLP: out (B, C) = W2 relu(W0 scores + b0) + b2, all f32."""
    B, Cn = scores.shape
    check(lib().lidk_lid_mlp(_p(scores), _p(w0), _p(b0), _p(w2), _p(b2), _p(out), B, Cn, w0.shape[0], _stream()), "lid_mlp")
    return out


# ----------------------------------------------------------------------------------------------- WavLM backbone (forward)
def wavlm_conv0_workspace(B, T0, C):
    return max(lib().lidk_wavlm_conv0_workspace(B, T0, C) // 4, 1)


def wavlm_conv0(wav, w, gamma, beta, out, T0, P0, eps=1e-5, workspace=None):
    """wav (B, L) f32 -> out (B*P0 [+slack], C) bf16: Conv1d(1, C, k10, s5) + GroupNorm(C, C) + GELU, channel-last."""
    B, Lw = wav.shape
    Cc = w.shape[0]
    ws = workspace if workspace is not None else torch.empty(wavlm_conv0_workspace(B, T0, Cc), device=wav.device, dtype=torch.float32)
    check(lib().lidk_wavlm_conv0(_p(wav), B, Lw, _p(w), _p(gamma), _p(beta), eps, _p(out), T0, P0, Cc, _p(ws), _stream()),
          "wavlm_conv0")
    return out


# ----------------------------------------------------------------------------------------------- layer_norm feature extractor
def wav_layernorm(wav, out=None, n_samples=None, eps=1e-5):
    """task.normalize of wav2vec2 Large / XLS-R: F.layer_norm(wav, wav.shape) per utterance (biased variance, eps inside the
    root) over the utterance's own samples; n_samples: optional int32 (B,) true lengths of a zero-padded batch."""
    out = torch.empty_like(wav) if out is None else out
    B, Lw = wav.shape
    check(lib().lidk_wav_layernorm(_p(wav), _p(out), B, Lw, _p(n_samples), eps, _stream()), "wav_layernorm")
    return out


def conv0_ln_fwd(wav, w, bias, gamma, beta, out, T0, P0, eps=1e-5):
    """wav (B, L) f32 -> out (B*P0 [+slack], 512) bf16: Conv1d(1, C, k10, s5, bias) + LayerNorm(C) + GELU, channel-last."""
    B, Lw = wav.shape
    check(lib().lidk_conv0_ln_fwd(_p(wav), B, Lw, _p(w), _p(bias), _p(gamma), _p(beta), eps, _p(out), T0, P0, w.shape[0],
                                  _stream()), "conv0_ln_fwd")
    return out


def ln_gelu_fwd(pre, gamma, beta, out, eps=1e-5):
    rows, Cc = pre.shape
    check(lib().lidk_ln_gelu_fwd(_p(pre), _p(gamma), _p(beta), _p(out), rows, Cc, eps, _code(pre), _stream()), "ln_gelu_fwd")
    return out


def ln_gelu_bwd(dy, pre, gamma, beta, dpre, dgamma, dbeta, B, P, Tv, eps=1e-5):
    check(lib().lidk_ln_gelu_bwd(_p(dy), _p(pre), _p(gamma), _p(beta), _p(dpre), _p(dgamma), _p(dbeta), B, P, Tv, pre.shape[-1], eps,
                                 _code(pre), _stream()), "ln_gelu_bwd")
    return dpre


def conv0_ln_bwd(wav, w, bias, gamma, beta, dy0, dw, dbias, dgamma, dbeta, T0, P0, eps=1e-5):
    B, Lw = wav.shape
    check(lib().lidk_conv0_ln_bwd(_p(wav), B, Lw, _p(w), _p(bias), _p(gamma), _p(beta), eps, _p(dy0), _p(dw), _p(dbias), _p(dgamma),
                                  _p(dbeta), T0, P0, w.shape[0], _stream()), "conv0_ln_bwd")


def wavlm_posconv_prep(x, xg, B, T, G, Pp, pad_left):
    """x (B*T, C) f32 -> xg (G, rows_total, C/G) bf16 zero-padded group-major copy."""
    check(lib().lidk_wavlm_posconv_prep(_p(x), _p(xg), B, T, x.shape[1], G, Pp, pad_left, xg.shape[1], _stream()),
          "wavlm_posconv_prep")
    return xg


def wavlm_add_rows(x, y, out, B, T, Pp):
    check(lib().lidk_wavlm_add_rows(_p(x), _p(y), _p(out), B, T, Pp, x.shape[1], _stream()), "wavlm_add_rows")
    return out


def wavlm_apply_mask(x, time_mask, chan_mask, mask_emb, B, T):
    """x (B*T, C) f32 in place; time_mask (B, T) uint8 or None; chan_mask (B, C) uint8 or None."""
    check(lib().lidk_wavlm_apply_mask(_p(x), _p(time_mask), _p(chan_mask), _p(mask_emb), B, T, x.shape[1], _stream()),
          "wavlm_apply_mask")
    return x


def wavlm_gate(x, wg, bg, grep_a, gate, B, T, H, dh):
    check(lib().lidk_wavlm_gate(_p(x), _p(wg), _p(bg), _p(grep_a), _p(gate), B, T, H, dh, _stream()), "wavlm_gate")
    return gate


def wavlm_attn_max_frames(dh):
    return lib().lidk_wavlm_attn_max_frames(dh)


def wavlm_attn_fwd(qkv, gate, rb, out, B, T, H, dh, probs=None):
    """rb (H, 2*RB-1) f32: the head's bias as a function of the offset j - i (entry r + RB - 1).  probs (B, H, T, ldp) bf16
    (training): the softmax probabilities are written too, for the backward pass."""
    RB = (rb.shape[1] + 1) // 2
    if probs is None:
        check(lib().lidk_wavlm_attn_fwd(_p(qkv), _p(gate), _p(rb), _p(out), B, T, H, dh, RB, _stream()), "wavlm_attn_fwd")
    else:
        if probs.shape[-1] != wavlm_attn_ldp(T) or probs.dtype != torch.bfloat16:
            raise LidkError("wavlm_attn_fwd: probs must be bf16 with last dimension wavlm_attn_ldp(T)")
        check(lib().lidk_wavlm_attn_fwd_probs(_p(qkv), _p(gate), _p(rb), _p(out), _p(probs), B, T, H, dh, RB, _stream()),
              "wavlm_attn_fwd_probs")
    return out


def wavlm_attn_bias_grads(ds, gate, rb, dgate, drb, B, T, H, ldp):
    """dgate / drb from the bf16 dS rows an ``attn_bwd(..., drel_emb=None)`` call left in its scratch buffer."""
    RB = (rb.shape[1] + 1) // 2
    check(lib().lidk_wavlm_attn_bias_grads(_p(ds), _p(gate), _p(rb), _p(dgate), _p(drb), B, T, H, RB, ldp, _stream()),
          "wavlm_attn_bias_grads")


def wavlm_attn_ldp(T):
    return lib().lidk_wavlm_attn_ldp(T)


def wavlm_attn_probs(qkv, gate, rb, probs, B, T, H, dh):
    RB = (rb.shape[1] + 1) // 2
    check(lib().lidk_wavlm_attn_probs(_p(qkv), _p(gate), _p(rb), _p(probs), B, T, H, dh, RB, _stream()), "wavlm_attn_probs")
    return probs


def wavlm_attn_bwd(qkv, probs, dout, gate, rb, dqkv, dgate, drb, dscores, B, T, H, dh):
    RB = (rb.shape[1] + 1) // 2
    check(lib().lidk_wavlm_attn_bwd(_p(qkv), _p(probs), _p(dout), _p(gate), _p(rb), _p(dqkv), _p(dgate), _p(drb), _p(dscores),
                                    B, T, H, dh, RB, _stream()), "wavlm_attn_bwd")


def wavlm_conv_dlast(dsrc, pre, dpre, B, T, P):
    check(lib().lidk_wavlm_conv_dlast(_p(dsrc), _p(pre), _p(dpre), B, T, P, dsrc.shape[-1], _stream()), "wavlm_conv_dlast")
    return dpre


def wavlm_conv_col2im(dcol, pre, dprev, B, P, T, Tprev, kW, C):
    check(lib().lidk_wavlm_conv_col2im(_p(dcol), _p(pre), _p(dprev), B, P, T, Tprev, kW, C, _stream()), "wavlm_conv_col2im")
    return dprev


def wavlm_conv0_stats_offset(B, T0, C):
    """Float offset of the (mean, rstd) block inside lidk_wavlm_conv0's workspace."""
    return B * (-(-T0 // 128)) * C * 2


def wavlm_conv0_bwd(wav, w, gamma, beta, stats, dy0, sums, dw, dgamma, dbeta, T0, P0):
    B, Lw = wav.shape
    check(lib().lidk_wavlm_conv0_bwd(_p(wav), B, Lw, _p(w), _p(gamma), _p(beta), _p(stats), _p(dy0), _p(sums), _p(dw), _p(dgamma),
                                     _p(dbeta), T0, P0, w.shape[0], _stream()), "wavlm_conv0_bwd")


def zero_padded_rows(x, klen, B, T):
    """x (B*T, C) f32 in place: rows t >= klen[b] become zero (wav2vec2 zeroes padded frames in front of the positional conv)."""
    check(lib().lidk_zero_padded_rows(_p(x), _p(klen), B, T, x.shape[-1], _stream()), "zero_padded_rows")
    return x


def hidden_mix_axpy(h, w, l, out, overwrite=False):
    """out = (0 if overwrite else out) + softmax(w)[l] * h  (s3prl Featurizer weighted sum, one hidden state per launch)."""
    check(lib().lidk_hidden_mix_axpy(_p(h), _p(w), w.numel(), l, _p(out), h.numel(), int(overwrite), _stream()), "hidden_mix_axpy")
    return out


def hidden_mix_dot(a, b, dot):
    check(lib().lidk_hidden_mix_dot(_p(a), _p(b), _p(dot), a.numel(), _stream()), "hidden_mix_dot")


def hidden_mix_wgrad(w, dots, dw):
    check(lib().lidk_hidden_mix_wgrad(_p(w), _p(dots), _p(dw), w.numel(), _stream()), "hidden_mix_wgrad")


def xattn_max_frames(dh):
    return lib().lidk_xattn_max_frames(dh)


def xattn_fwd(qkv, out, lse, B, T, H, dh, gate=None, rb=None, klen=None, keep=None, drop_p=0.0, seed=0):
    """Key-tiled attention forward (any T): out (B*T, H*dh) bf16, lse (B, H, T) f32.  gate (B, H, T) + rb (H, 2*RB-1): WavLM's
    gated relative-position bias; klen (B,) int32: keys at or beyond it are masked (wav2vec2's key padding mask); keep
    (B, H, T, T) uint8 forces the attention-dropout mask (tests), drop_p > 0 draws it from (seed, element index)."""
    if (gate is None) != (rb is None):
        raise LidkError("xattn_fwd: gate and rb come together")
    RB = (rb.shape[1] + 1) // 2 if rb is not None else T
    if klen is not None and klen.dtype != torch.int32:
        raise LidkError("xattn_fwd: klen must be int32")
    check(lib().lidk_xattn_fwd(_p(qkv), _p(gate), _p(rb), _p(klen), _p(out), _p(lse), _p(keep), float(drop_p), int(seed), B, T, H,
                               dh, RB, _stream()), "xattn_fwd")
    return out


def xattn_bwd(qkv, out, dout, lse, dqkv, delta, B, T, H, dh, gate=None, rb=None, klen=None, keep=None, drop_p=0.0, seed=0,
              dgate=None, drb=None):
    """Backward of xattn_fwd from out, lse and dout: dqkv (dq | dk | dv) written, dgate written, drb accumulated."""
    RB = (rb.shape[1] + 1) // 2 if rb is not None else T
    check(lib().lidk_xattn_bwd(_p(qkv), _p(gate), _p(rb), _p(klen), _p(out), _p(dout), _p(lse), _p(keep), float(drop_p), int(seed),
                               _p(dqkv), _p(delta), _p(dgate), _p(drb), B, T, H, dh, RB, _stream()), "xattn_bwd")
    return dqkv


def wavlm_gate_bwd(x, wg, bg, grep_a, dgate, dx, dwg, dbg, dgrep_a, B, T, H, dh):
    check(lib().lidk_wavlm_gate_bwd(_p(x), _p(wg), _p(bg), _p(grep_a), _p(dgate), _p(dx), _p(dwg), _p(dbg), _p(dgrep_a), B, T, H,
                                    dh, _stream()), "wavlm_gate_bwd")


def wavlm_posconv_dprep(dy, pre, dpc, B, T, Pp, dpg=None, goff=0):
    G = dpg.shape[0] if dpg is not None else 0
    if dpg is not None and dpg.shape[1] != dpc.shape[0]:
        raise LidkError("wavlm_posconv_dprep: dpg must hold as many rows per group as dpc")
    check(lib().lidk_wavlm_posconv_dprep(_p(dy), _p(pre), _p(dpc), _p(dpg), B, T, Pp, dy.shape[1], G, goff, dpc.shape[0],
                                         _stream()), "wavlm_posconv_dprep")
    return dpc


# ----------------------------------------------------------------------------------------------- optimizer
def novograd_step(params, grads, exp_avg, exp_avg_sq, work, n_tensors, lr, betas, eps, weight_decay, grad_averaging,
                  max_norm, scratch, total_norm):
    check(lib().lidk_novograd_step(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), _p(work), work.shape[0], n_tensors,
                                   lr, betas[0], betas[1], eps, weight_decay, int(grad_averaging), max_norm, _p(scratch),
                                   _p(total_norm), _stream()), "novograd_step")


def cast_weights(params, wT, mats, total_tiles):
    """mats: device int64 (n, 8) tile table built by ``build_cast_table``."""
    check(lib().lidk_cast_weights(_p(params), _p(wT), _p(mats), mats.shape[0], int(total_tiles), _code(wT), _stream()),
          "cast_weights")


def build_cast_table(entries, device):
    """entries: iterable of (src_off, R, C, w_off, t_off, ldt) -> (device table (n, 8) int64, total 32x32 tiles)."""
    rows, tiles = [], 0
    for src, R, C, w_off, t_off, ldt in entries:
        tx = (C + 31) // 32
        rows.append([src, R, C, w_off, t_off, ldt or R, tiles, tx])
        tiles += ((R + 31) // 32) * tx
    return torch.tensor(rows, dtype=torch.int64, device=device), tiles
