"""Oracle: waveform -> log-mel feature path (test infrastructure, see oracle/__init__.py).

Follows lid/audio_processor.py (reference) and the torchaudio 0.12.1 transforms it
calls.  torchaudio is NOT in this image, so the torchaudio half is restated from its
published algorithm: parity against torchaudio itself is UNPINNED (see package doc).
"""
import math

import numpy as np
import torch

N_FFT = 512
N_FREQ = N_FFT // 2 + 1
MEL_SAMPLE_RATE = 16000  # MelSpectrogram() is built without sample_rate -> 16000


def normalize_wav(wav: torch.Tensor) -> torch.Tensor:
    """lid/audio_processor.py:108-115 — (x - mean) / (std_unbiased + 1e-6) over the last dim."""
    std, mean = torch.std_mean(wav, dim=-1, keepdim=True)
    return (wav - mean) / (std + 1e-6)


def dither_preemphasis(wav: torch.Tensor, noise: torch.Tensor = None, coef: float = 0.97) -> torch.Tensor:
    """lid/audio_processor.py:128-134 — x += 1e-5*U[0,1); y[0]=x[0], y[t]=x[t]-0.97*x[t-1].

    ``noise`` is the U[0,1) draw (same shape as wav); None means no dither.
    """
    if noise is not None:
        wav = wav + 1e-5 * noise
    return torch.cat((wav[..., :1], wav[..., 1:] - coef * wav[..., :-1]), dim=-1)


def melscale_fbanks(n_freqs: int = N_FREQ, f_min: float = 0.0, f_max: float = MEL_SAMPLE_RATE // 2,
                    n_mels: int = 80, sample_rate: int = MEL_SAMPLE_RATE) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> (n_freqs, n_mels) f32.

    Called through MelSpectrogram -> MelScale at lid/audio_processor.py:91-101.
    """
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def frame_geometry(sr: int = 16000, win_length: float = 0.025, hop_length: float = 0.01):
    """lid/audio_processor.py:89-90 — win = int(sr*0.025), hop = int(sr*0.01)."""
    return int(sr * win_length), int(sr * hop_length)


def num_frames(n_samples: int, pad: int = 0, hop: int = 160) -> int:
    """torch.stft(center=True): 1 + (L + 2*pad) // hop."""
    return 1 + (n_samples + 2 * pad) // hop


def power_spectrogram(wav: torch.Tensor, win: int = 400, hop: int = 160, pad: int = 0) -> torch.Tensor:
    """torchaudio Spectrogram(power=2): zero-pad ``pad`` both sides, torch.stft(n_fft=512,
    hann(win) periodic centred in the 512 frame, center=True, reflect), |X|^2.
    wav (..., L) -> (..., 257, F)."""
    if pad > 0:
        wav = torch.nn.functional.pad(wav, (pad, pad))
    window = torch.hann_window(win)
    spec = torch.stft(wav, N_FFT, hop_length=hop, win_length=win, window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return spec.abs().pow(2.0)


def amplitude_to_db(x: torch.Tensor, top_db: float = 80.0) -> torch.Tensor:
    """torchaudio AmplitudeToDB('power', top_db=80) on ONE utterance (1, n_mels, F)
    (lid/audio_processor.py:104): 10*log10(max(x,1e-10)), then floor at
    (max over the whole utterance) - top_db.  A leading batch dim is treated per item."""
    x_db = 10.0 * torch.log10(torch.clamp(x, min=1e-10))
    if top_db is not None:
        floor = x_db.amax(dim=(-2, -1), keepdim=True) - top_db
        x_db = torch.max(x_db, floor)
    return x_db


def wav2mel(wav: torch.Tensor, n_mels: int = 80, pad: int = 0, sr: int = 16000,
            win_length: float = 0.025, hop_length: float = 0.01) -> torch.Tensor:
    """lid/audio_processor.py:72-105 (_internal_wav2mel).  wav (B, L) or (1, L) -> (B, n_mels, F) dB."""
    win, hop = frame_geometry(sr, win_length, hop_length)
    spec = power_spectrogram(wav, win, hop, pad)                      # (B, 257, F)
    fb = melscale_fbanks(n_mels=n_mels)                               # (257, n_mels)
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)  # (B, n_mels, F)
    return amplitude_to_db(mel, 80.0)


def draw_mask_span(size: int, mask_param: float, gen: torch.Generator = None):
    """torchaudio.functional.mask_along_axis span: value=U*param; min=U*(size-value);
    start=floor(min); end=start+floor(value).  Consumes two torch.rand(1) draws, in that order."""
    value = torch.rand(1, generator=gen) * mask_param
    min_value = torch.rand(1, generator=gen) * (size - value)
    start = int(min_value.long())
    return start, start + int(value.long())


def draw_specaug_spans(n_frames: int, n_mels: int = 80, t_mask: float = 0.05, f_mask: float = 27,
                       mask_times: int = 0, gen: torch.Generator = None):
    """lid/audio_processor.py:225-227 — per repetition: TimeMasking(int(F*t_mask)) then
    FrequencyMasking(f_mask).  Returns [(t0, t1, f0, f1)] * mask_times."""
    spans = []
    for _ in range(mask_times):
        t0, t1 = draw_mask_span(n_frames, int(n_frames * t_mask), gen)
        f0, f1 = draw_mask_span(n_mels, f_mask, gen)
        spans.append((t0, t1, f0, f1))
    return spans


def apply_specaug(spec: torch.Tensor, spans) -> torch.Tensor:
    """Fill [t0,t1) on time and [f0,f1) on mel with literal 0.0 (dB) — SURVEY Q2.  spec (n_mels, F)."""
    spec = spec.clone()
    for (t0, t1, f0, f1) in spans:
        spec[..., :, t0:t1] = 0.0
        spec[..., f0:f1, :] = 0.0
    return spec


def collate_mel(specs):
    """lid/raw_datasets.py:345-365 (mel branch): list of (n_mels, F_i) -> (B, Fmax, n_mels) zero padded,
    wav_percents = F_i / Fmax."""
    fmax = max(s.shape[-1] for s in specs)
    out = torch.zeros(len(specs), fmax, specs[0].shape[0])
    for i, s in enumerate(specs):
        out[i, : s.shape[-1]] = s.transpose(0, 1)
    pct = torch.tensor([s.shape[-1] / fmax for s in specs], dtype=torch.float32)
    return out, pct


# ---------------------------------------------------------------------------------------
# Independent float64 numpy restatement of the STFT definition (used to pin torch.stft use)
# ---------------------------------------------------------------------------------------
def power_spectrogram_np(wav: np.ndarray, win: int = 400, hop: int = 160, pad: int = 0) -> np.ndarray:
    """Direct definition: frame t covers padded[t*hop : t*hop+512] of the reflect-padded signal,
    multiplied by hann_periodic(win) centred in 512, rFFT, |.|^2.  wav (L,) -> (257, F) float64."""
    x = np.asarray(wav, dtype=np.float64)
    if pad:
        x = np.concatenate([np.zeros(pad), x, np.zeros(pad)])
    x = np.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect")
    n = np.arange(win)
    w = np.zeros(N_FFT)
    left = (N_FFT - win) // 2
    w[left:left + win] = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win)
    nfr = 1 + (len(x) - N_FFT) // hop
    out = np.empty((N_FREQ, nfr))
    for t in range(nfr):
        out[:, t] = np.abs(np.fft.rfft(x[t * hop:t * hop + N_FFT] * w)) ** 2
    return out


def speed_perturb_np(x, p: int, q: int, zeros: int = 16, beta: float = 8.6):
    """Speed perturbation y[n] = x(n*p/q), n < round(len/v) (lid/audio_processor.py:136-156: sox "speed" v + "rate" sr with
    v = p/q in {9/10, 1, 11/10}), restated in float64 numpy as direct band-limited interpolation: Kaiser-windowed sinc,
    cut-off min(1, q/p) of Nyquist, `zeros` zero crossings per side, unit DC gain per output sample.  sox's own resampling
    filter is not part of the reference tree (parity unpinned); tests also compare with scipy.signal.resample_poly."""
    import numpy as np
    x = np.asarray(x, dtype=np.float64)
    n_out = int(len(x) * q / p + 0.5)
    if p == q:
        return x[:n_out].copy()
    fc = min(1.0, q / p)
    half = zeros / fc
    y = np.zeros(n_out)
    for n in range(n_out):
        c = n * p / q
        k = np.arange(int(np.floor(c - half)), int(np.ceil(c + half)) + 1)
        t = k - c
        w = np.where(np.abs(t) <= half, np.i0(beta * np.sqrt(np.clip(1.0 - (t / half) ** 2, 0.0, 1.0))) / np.i0(beta), 0.0)
        h = fc * np.sinc(fc * t) * w
        h = h / h.sum()
        ok = (k >= 0) & (k < len(x))
        y[n] = float((h[ok] * x[k[ok]]).sum())
    return y
