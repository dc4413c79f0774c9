"""Feature path of the LID pipeline on the GPU — the reference's lid/audio_processor.py surface
(``normalize_wav``, ``wav_augment`` dither + pre-emphasis, ``wav2mel``, ``spectrogram_augment``) backed by lidk kernels.

In the reference these run per utterance on CPU DataLoader workers (torchaudio).  Here a batch travels to the GPU as raw
waveforms (``WaveBatch``) and one fused launch sequence produces the ``(B, F, n_mels)`` dB features the model consumes:
reflect-padded STFT (LDS FFT-512) -> |X|^2 -> mel -> dB -> per-utterance top_db floor -> SpecAugment masks.  SpecAugment
spans are drawn on the host with torch's generator in torchaudio's draw order, so runs are reproducible from the seed.

Speed perturbation (sox "speed" {0.9, 1.0, 1.1} + "rate", lid/audio_processor.py:136-156) is polyphase resampling on the GPU
(``speed_perturb``); the draw happens in the collate like the reference's per-item draw.
Out of scope (CPU-only libsox effects): pitch perturbation, reverb, awgn, the kaldi fbank variant.
"""
from typing import List, Optional

import torch

from lidk import ops as _ops
from lidk._lib import LidkError


SPEED_FACTORS = ((9, 10), (1, 1), (11, 10))       # speed v = p/q: the reference's random.choice([0.9, 1.0, 1.1])


def speed_out_len(n_samples: int, p: int, q: int) -> int:
    """Samples left after speeding an utterance up by v = p/q."""
    return _ops.speed_out_len(n_samples, p, q)


def speed_perturb(wav: torch.Tensor, factors, n_samples: Optional[torch.Tensor] = None):
    """wav (B, L) on the GPU, factors [(p, q)] per utterance -> (wav' (B, L'), n_samples' int32 (B,), lengths list)."""
    return _ops.speed_perturb(wav.contiguous(), list(factors), n_samples)


def frame_geometry(sr: int = 16000, win_length: float = 0.025, hop_length: float = 0.01):
    return int(sr * win_length), int(sr * hop_length)


def num_frames(n_samples: int, pad: int = 0, hop: int = 160) -> int:
    return 1 + (n_samples + 2 * pad) // hop


def normalize_wav(wav: torch.Tensor) -> torch.Tensor:
    """(x - mean) / (std_unbiased + 1e-6) per utterance; wav (B, L) or (1, L) on the GPU."""
    return _ops.normalize_wav(wav.contiguous())


def wav_augment(wav: torch.Tensor, sr: int, speed_shift: bool = False, pitch_shift: bool = False, reverb: bool = False,
                seed: int = 0):
    """Dither (1e-5 * U[0,1)) + pre-emphasis 0.97, then (speed_shift) one speed factor drawn from {0.9, 1.0, 1.1} for the
    whole (1, L) or (B, L) input, as the reference's per-utterance call does.  Pitch shift and reverb (libsox) are not built."""
    import random as _random
    if pitch_shift or reverb:
        raise NotImplementedError("sox pitch/reverb are CPU-only libsox effects outside the lidk hot path")
    out = _ops.dither_preemph(wav.contiguous(), coef=0.97, dither=1e-5, seed=seed)
    if speed_shift:
        f = _random.choice(SPEED_FACTORS)
        out, _, _ = _ops.speed_perturb(out, [f] * out.shape[0])
    return out, sr


def wav2mel(x: torch.Tensor, use_kaildi: bool = False, win_length: float = 0.025, hop_length: float = 0.01,
            n_mels: int = 80, n_fft: int = 512, pad: int = 0, sr: int = 16000) -> torch.Tensor:
    """x (1, L) [or (B, L)] f32 on the GPU -> (1, n_mels, F) [or (B, n_mels, F)] dB, as the reference's wav2mel."""
    if use_kaildi:
        raise NotImplementedError("kaldi fbank is never enabled by the lid confs (SURVEY 2 #1)")
    if n_fft != 512:
        raise NotImplementedError("the lidk STFT kernel is a 512-point FFT")
    win, hop = frame_geometry(sr, win_length, hop_length)
    return _ops.logmel(x.contiguous(), pad=pad, hop=hop, win_length=win, n_mels=n_mels).transpose(1, 2)


def draw_mask_span(size: int, mask_param: float, gen: Optional[torch.Generator] = None):
    """torchaudio.functional.mask_along_axis: value = U*param, min = U*(size-value); [floor(min), floor(min)+floor(value))."""
    value = torch.rand(1, generator=gen) * mask_param
    min_value = torch.rand(1, generator=gen) * (size - value)
    start = int(min_value.long())
    return start, start + int(value.long())


def draw_specaug_spans(n_frames: int, n_mels: int = 80, t_mask: float = 0.05, f_mask: float = 27, mask_times: int = 0,
                       gen: Optional[torch.Generator] = None) -> List[tuple]:
    """Per repetition: TimeMasking(int(F * t_mask)) then FrequencyMasking(f_mask) (lid/audio_processor.py:225-227)."""
    spans = []
    for _ in range(mask_times):
        t0, t1 = draw_mask_span(n_frames, int(n_frames * t_mask), gen)
        f0, f1 = draw_mask_span(n_mels, f_mask, gen)
        spans.append((t0, t1, f0, f1))
    return spans


def spectrogram_augment(spec: torch.Tensor, sr: int = 16000, n_mels: int = 80, hop_length: float = 0.01,
                        t_mask: float = 0.05, f_mask: float = 27, mask_times: int = 0, t_stretch: bool = False):
    """spec (1, n_mels, F) -> masked copy (mask value 0.0 dB).  Stand-alone form for callers that hold a finished
    spectrogram; in training the masks are fused into the log-mel launch through ``WaveBatch.spans``."""
    if t_stretch:
        raise NotImplementedError("TimeStretch is never enabled by the lid confs (SURVEY 2 #2)")
    out = spec.clone()
    for t0, t1, f0, f1 in draw_specaug_spans(spec.size(-1), spec.size(-2), t_mask, f_mask, mask_times):
        out[..., :, t0:t1] = 0.0
        out[..., f0:f1, :] = 0.0
    return out


class WaveBatch:
    """A batch of equal-length raw utterances plus everything the GPU feature path needs.

    It stands where the reference's collate puts the mel tensor (``batch[0]``): ``Trainer.batch_to_device`` moves it with
    ``.to(device)`` and the model calls ``to_mel()``.  Utterances shorter than ``wav.shape[1]`` are zero padded and
    ``n_samples`` holds their true lengths: every utterance is then normalised and transformed as if it were alone, and the
    rows behind its own frames are exactly 0.0 - the zero-padded mel the reference's collate builds
    (lid/raw_datasets.py:345-365)."""

    def __init__(self, wav: torch.Tensor, spans: Optional[torch.Tensor] = None, pad: int = 0, n_mels: int = 80, sr: int = 16000,
                 normalize: bool = True, preemph: bool = False, dither_seed: int = 0, n_samples: Optional[torch.Tensor] = None,
                 speed=None):
        self.wav, self.spans = wav, spans
        self.n_samples = n_samples          # int32 (B,) true lengths of a ragged batch, or None when all rows are full
        self.speed = speed                  # per-utterance (p, q) speed factors (training augmentation) or None
        self.pad, self.n_mels, self.sr = pad, n_mels, sr
        self.normalize, self.preemph, self.dither_seed = normalize, preemph, dither_seed

    @property
    def shape(self):
        B, L = self.wav.shape
        if self.speed is not None:          # frames after the speed perturbation
            lens = self.n_samples.tolist() if self.n_samples is not None else [L] * B
            L = max(speed_out_len(n, p, q) for n, (p, q) in zip(lens, self.speed))
        return (B, num_frames(L, self.pad), self.n_mels)

    @property
    def device(self):
        return self.wav.device

    def to(self, device, non_blocking: bool = False):
        self.wav = self.wav.to(device, non_blocking=non_blocking)
        if self.spans is not None:
            self.spans = self.spans.to(device, non_blocking=non_blocking)
        if self.n_samples is not None:
            self.n_samples = self.n_samples.to(device, non_blocking=non_blocking)
        return self

    def pin_memory(self):
        self.wav = self.wav.pin_memory()
        if self.spans is not None:
            self.spans = self.spans.pin_memory()
        if self.n_samples is not None:
            self.n_samples = self.n_samples.pin_memory()
        return self

    _mel = None
    _mel_ready = None

    def prefetch_mel(self, stream, ordered: bool = False) -> None:
        """Start the feature kernels for this batch on ``stream`` (the Trainer's feature stream) so that they run beside the
        training step of the batch before it; ``to_mel`` then only waits for their completion event.  The reference gets the
        same overlap from its DataLoader workers, which compute mel on the CPU while the GPU trains."""
        if self._mel is not None or not self.wav.is_cuda:
            return
        if not ordered:                                          # (ordered: the caller has placed `stream` behind its inputs)
            stream.wait_stream(torch.cuda.current_stream())      # the H2D copy of wav and everything issued before
        with torch.cuda.stream(stream):
            self._mel = self._compute_mel()
            self._mel_ready = torch.cuda.Event()
            self._mel_ready.record(stream)

    def to_mel(self) -> torch.Tensor:
        """-> (B, F, n_mels) f32 dB on the GPU (the model input)."""
        if self._mel is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self._mel_ready)
            mel, self._mel, self._mel_ready = self._mel, None, None
            mel.record_stream(cur)
            return mel
        return self._compute_mel()

    def _compute_mel(self) -> torch.Tensor:
        if getattr(_ops, "IS_HIP_BACKEND", False) and not self.wav.is_cuda:
            raise LidkError("WaveBatch.to_mel: the feature path runs on the GPU only (no CPU fallback)")
        x = self.wav.contiguous()
        ns = self.n_samples.contiguous() if self.n_samples is not None else None
        spans = self.spans.contiguous() if self.spans is not None else None
        if self.normalize and self.preemph and self.speed is None and hasattr(_ops, "wav2mel"):
            # normalise + dither + pre-emphasis inside the STFT's frame load: three launches, no intermediate waveforms
            return _ops.wav2mel(x, pad=self.pad, n_mels=self.n_mels, spans=spans, n_samples=ns, coef=0.97, dither=1e-5,
                                seed=self.dither_seed)
        if self.normalize:
            x = _ops.normalize_wav(x, n_samples=ns)
        if self.preemph:
            x = _ops.dither_preemph(x, coef=0.97, dither=1e-5, seed=self.dither_seed)
        if self.speed is not None:          # reference order: dither, pre-emphasis, then the sox speed effect
            x, ns, _ = _ops.speed_perturb(x, self.speed, ns)
        return _ops.logmel(x, pad=self.pad, n_mels=self.n_mels, spans=spans, n_samples=ns)
