"""Datasets, collate and samplers for the LID pipeline — the reference's lid/raw_datasets.py surface
(``RawDataset``, ``MergedDataset``, ``RandomSamplerWithBase``, ``MutiBatchSampler``) plus a synthetic source.

Difference that matters: ``collate_fn`` does NOT compute features on CPU workers.  With ``type: mel`` it returns a
``WaveBatch`` (raw waveforms + host-drawn SpecAugment spans) in the slot where the reference puts the mel tensor; the
log-mel / SpecAugment kernels run on the GPU when the model consumes it.  The six-tuple layout is unchanged:
``(wavs, texts (B,Lmax) int64, wav_percents (B,), text_percents (B,), audio_paths list[str], langs (B,) int64)``.
"""
import csv
import logging
import math
import os
import random
import wave
from typing import Any, Dict, Iterator, List, Optional

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset, Sampler

from lid.audio_processor import SPEED_FACTORS, WaveBatch, draw_specaug_spans, num_frames, speed_out_len


# ------------------------------------------------------------------------------------------------ audio I/O (host side)
def read_audio(audio_path: str, normalize: bool = True):
    """-> (wav (1, L) f32 on the host, sample_rate).  Reads .npy arrays and PCM .wav files (stdlib ``wave``); compressed
    formats need torchaudio/soundfile, which this image lacks.  Normalisation happens on the GPU (WaveBatch.normalize)."""
    if audio_path.endswith(".npy"):
        wav, sr = torch.from_numpy(np.load(audio_path).astype(np.float32)).reshape(1, -1), 16000
    else:
        with wave.open(audio_path, "rb") as f:
            sr, n, width, ch = f.getframerate(), f.getnframes(), f.getsampwidth(), f.getnchannels()
            raw = f.readframes(n)
        if width != 2:
            raise ValueError(f"{audio_path}: only 16-bit PCM wav is supported")
        wav = torch.from_numpy(np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0).reshape(-1, ch).mean(1)[None]
    return wav, sr


def audio_duration(path: str) -> float:
    if path.endswith(".npy"):
        return np.load(path, mmap_mode="r").shape[-1] / 16000.0
    with wave.open(path, "rb") as f:
        return f.getnframes() / f.getframerate()


class RawDataset(Dataset):
    """One language's manifest: common-voice TSV (``clips/`` beside it) or xf ``.label`` (``wav/train/`` beside it)."""

    def __init__(self, manifest_path: str, max_duration=16.7, train=False, source: str = "common_voice") -> None:
        self.train = train
        rows = self._read_cv(manifest_path) if source == "common_voice" else self._read_xf(manifest_path)
        self.datasets = [r for r in rows if max_duration <= 0 or r["duration"] <= max_duration]
        logging.info("%s: %d utterances kept of %d", manifest_path, len(self.datasets), len(rows))

    @staticmethod
    def _read_cv(manifest_path):
        base = os.path.join(os.path.dirname(manifest_path), "clips")
        out = []
        with open(manifest_path, encoding="utf-8") as f:
            for row in csv.DictReader(f, delimiter="\t"):
                path = os.path.join(base, row["path"])
                out.append({"path": path, "locale": row["locale"], "sentence": row["sentence"], "duration": audio_duration(path)})
        return out

    @staticmethod
    def _read_xf(manifest_path):
        lang = os.path.basename(os.path.dirname(manifest_path))
        base = os.path.join(os.path.dirname(manifest_path), "wav", "train")
        out = []
        with open(manifest_path) as f:
            for line in f:
                name, text = line.rstrip("\n").split("\t")[:2]
                path = os.path.join(base, name)
                out.append({"path": path, "locale": lang, "sentence": text.strip(), "duration": audio_duration(path)})
        return out

    def __getitem__(self, index):
        return self.datasets[index]

    def __len__(self):
        return len(self.datasets)

    def lang(self):
        return self.datasets[0]["locale"]

    def export_vocab(self):
        return sorted({c for item in self.datasets for c in item["sentence"]})


# ------------------------------------------------------------------------------------------------ samplers
class RandomSamplerWithBase(Sampler[int]):
    """Random permutation of one language's items, offset into the merged index space."""

    def __init__(self, data_source, generator=None) -> None:
        self.data_source, self.generator, self.base_value = data_source, generator, 0

    @property
    def num_samples(self) -> int:
        return len(self.data_source)

    def set_base_value(self, value: int):
        self.base_value = value

    def __iter__(self) -> Iterator[int]:
        g = self.generator
        if g is None:
            g = torch.Generator()
            g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        yield from (i + self.base_value for i in torch.randperm(len(self.data_source), generator=g).tolist())

    def __len__(self) -> int:
        return self.num_samples


class MutiBatchSampler(Sampler[List[int]]):
    """Single-language batches, languages picked in proportion to their remaining size (reference:
    lid/raw_datasets.py:374-440; a batch never mixes languages because one head is trained per step).

    Data parallel: with ``world_size > 1`` every rank runs the SAME seeded draw of a global batch of
    ``batch_size * world_size`` indices from one language and keeps its contiguous ``batch_size`` slice, so all ranks train
    the same head on disjoint utterances.  ``set_epoch`` reseeds the draw."""

    def __init__(self, samplers: List[Sampler[int]], batch_size: int, drop_last: bool, rank: int = 0, world_size: int = 1,
                 seed: Optional[int] = None, lengths=None, bucket_window: int = 0) -> None:
        """lengths (index -> samples) + bucket_window > 0: bucketed padding (BASELINE config 5).  Each language's permutation is
        cut into windows of ``bucket_window`` global batches, a window is sorted by length and cut into batches (so a batch
        pads to a similar length), and the window's batches are then drawn in random order.  0 = the reference's behaviour."""
        self.samplers, self.batch_size, self.drop_last = samplers, batch_size, drop_last
        self.rank, self.world_size, self.seed, self.epoch = rank, world_size, seed, 0
        self.weight = [len(s) for s in samplers]
        self.lengths, self.bucket_window = lengths, int(bucket_window)

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __iter__(self) -> Iterator[List[int]]:
        gb = self.batch_size * self.world_size
        rng = random.Random((self.seed, self.epoch).__hash__()) if (self.seed is not None or self.world_size > 1) else random
        if self.world_size > 1 or self.seed is not None:
            for k, s in enumerate(self.samplers):                           # identical permutations on every rank
                s.generator = torch.Generator().manual_seed(((self.seed or 0) * 1000003 + self.epoch * 1009 + k) % (2 ** 31))
        iters = [iter(s) for s in self.samplers]
        if self.lengths is not None and self.bucket_window > 0:
            iters = [iter(self._bucketed(list(it), gb, rng)) for it in iters]
        remain = [len(s) for s in self.samplers]
        while sum(remain) > 0:
            area = rng.randint(0, sum(remain) - 1)
            idx = 0
            while area >= remain[idx]:
                area -= remain[idx]
                idx += 1
            take = min(gb, remain[idx])
            batch = [next(iters[idx]) for _ in range(take)]
            remain[idx] -= take
            if take == gb:
                yield batch[self.rank * self.batch_size:(self.rank + 1) * self.batch_size]
            elif not self.drop_last:
                per = math.ceil(take / self.world_size)
                mine = batch[self.rank * per:(self.rank + 1) * per]
                yield mine if mine else batch[:1]

    def _bucketed(self, order: List[int], gb: int, rng) -> List[int]:
        """One language's permutation re-ordered so that consecutive runs of ``gb`` indices have similar lengths."""
        out, win = [], self.bucket_window * gb
        for w0 in range(0, len(order), win):
            chunk = sorted(order[w0:w0 + win], key=lambda i: self.lengths(i))
            batches = [chunk[b0:b0 + gb] for b0 in range(0, len(chunk), gb)]
            full = [b for b in batches if len(b) == gb]
            rng.shuffle(full)
            out += [i for b in full for i in b] + [i for b in batches if len(b) != gb for i in b]
        return out

    def __len__(self) -> int:
        gb = self.batch_size * self.world_size
        if self.drop_last:
            return sum(len(s) // gb for s in self.samplers)
        return sum((len(s) + gb - 1) // gb for s in self.samplers)


# ------------------------------------------------------------------------------------------------ datasets
class _FeatureCfg:
    def __init__(self, type="wav", speed_shift=True, pitch_shift=True, reverb=True, use_kaildi=False, win_length=0.025,
                 hop_length=0.01, n_mels=80, n_fft=512, pad=0, sr=16000, t_mask=0.05, f_mask=27, mask_times=2, t_stretch=False,
                 **_ignored):
        self.type, self.pad, self.sr, self.n_mels = type, pad, sr, n_mels
        self.t_mask, self.f_mask, self.mask_times = t_mask, f_mask, mask_times
        self.speed_shift = bool(speed_shift)          # speed perturbation {0.9, 1.0, 1.1}: polyphase resampling on the GPU
        if pitch_shift or reverb:
            logging.warning("pitch shift and reverb are libsox CPU effects outside the lidk path: disabled")


class _CollateMixin:
    def collate_fn(self, batch):
        """batch: list of (wav (L,) or (1,L), encoded text, path, lang)."""
        fc = self.feat
        wavs = [b[0].reshape(-1) for b in batch]
        lens = [w.shape[0] for w in wavs]
        texts = pad_sequence([b[1] for b in batch]).transpose(1, 0)
        langs = torch.LongTensor([self.lang2index_dict[b[3]] for b in batch])
        text_pct = torch.FloatTensor([b[1].shape[-1] / (texts.shape[1] + 1e-9) for b in batch])
        if fc.type != "mel":
            wav_pct = torch.FloatTensor([n / max(lens) for n in lens])
            return wavs, texts, wav_pct, text_pct, [b[2] for b in batch], langs
        wav = pad_sequence(wavs, batch_first=True)
        speed = None
        if self.train and fc.speed_shift:
            # one draw per utterance, as the reference's wav_augment (lid/audio_processor.py:136-139); the resampling itself
            # runs on the GPU behind normalisation and pre-emphasis, and the lengths below are the perturbed ones
            speed = [random.choice(SPEED_FACTORS) for _ in lens]
            in_lens, lens = lens, [speed_out_len(n, p, q) for n, (p, q) in zip(lens, speed)]
        frames = [num_frames(n, fc.pad) for n in lens]
        wav_pct = torch.FloatTensor([f / max(frames) for f in frames])
        spans = None
        if self.train and fc.mask_times > 0:
            spans = torch.tensor([draw_specaug_spans(f, fc.n_mels, fc.t_mask, fc.f_mask, fc.mask_times) for f in frames],
                                 dtype=torch.int32)
        if speed is not None:
            ragged = torch.tensor(in_lens, dtype=torch.int32)
        else:
            ragged = torch.tensor(lens, dtype=torch.int32) if min(lens) != max(lens) else None
        wb = WaveBatch(wav, spans, pad=fc.pad, n_mels=fc.n_mels, sr=fc.sr, normalize=True, preemph=self.train,
                       dither_seed=random.getrandbits(31) if self.train else 0, n_samples=ragged, speed=speed)
        wb.lang_id = int(langs[0])
        return wb, texts, wav_pct, text_pct, [b[2] for b in batch], langs


class MergedDataset(_CollateMixin, Dataset):
    """File-based multi-language dataset (reference: lid/raw_datasets.py:187-365)."""

    def __init__(self, train: bool = False, manifest_files: List[str] = None, lang2index_dict: dict = None,
                 lang2tokenizer: Dict = None, max_duration: float = 16.7, source: str = "common_voice", **feature) -> None:
        self.train, self.lang2index_dict, self.lang2tokenizer = train, lang2index_dict, lang2tokenizer
        self.feat = _FeatureCfg(**feature)
        self.type = self.feat.type
        self.datasets, self.samplers = [], []
        for manifest in manifest_files:
            ds = RawDataset(manifest_path=manifest, train=train, max_duration=max_duration, source=source)
            sampler = RandomSamplerWithBase(data_source=ds)
            sampler.set_base_value(len(self.datasets))
            self.samplers.append(sampler)
            self.datasets.extend(ds.datasets)

    def __len__(self):
        return len(self.datasets)

    def __getitem__(self, index):
        item = self.datasets[index]
        wav, _ = read_audio(item["path"], normalize=False)
        text = self.lang2tokenizer[item["locale"]].encoder(item["sentence"]) if self.lang2tokenizer else torch.LongTensor([0])
        return wav, text, item["path"], item["locale"]

    def export_dict(self):
        return {s.data_source.lang(): s.data_source.export_vocab() for s in self.samplers}


class _SynthLang:
    def __init__(self, lang, n):
        self._lang, self._n = lang, n

    def __len__(self):
        return self._n

    def lang(self):
        return self._lang


class SyntheticMergedDataset(_CollateMixin, Dataset):
    """Synthetic LID data.  Language k is white noise through a 2-pole resonator at f_k = 300 + 500 k Hz (radius
    0.97): a distinct spectral signature per language.  Item i is a pure function of (seed, i): every rank and every epoch sees
    the same corpus.

    ``transcript="random"``: ``text_len`` uniform tokens unrelated to the audio (throughput runs only: CTC can learn nothing).

    ``transcript="tones"`` (learnable, used for the val-Cavg half of the metric): the utterance is cut into ``text_len`` equal
    segments; segment j carries token t_j as a PAIR of simultaneous tones (f_a, f_b), f_a = 300 + 100 a Hz (a < 16),
    f_b = 2200 + 200 b Hz (b < 16), on for the first 80 % of the segment with 5 ms raised-cosine edges (the gap gives CTC a
    boundary between repeated tokens).  Which of the 256 (a, b) pairs spell token t is a LANGUAGE-SPECIFIC choice
    (``torch.randperm(256, seed 99 + k)[t]``): like phoneme inventories, languages share the same sounds but not the same
    symbols, so a head trained on language k reads its own language confidently and foreign audio - mostly pairs it never saw -
    hesitantly, which is what the CTC-confidence LID score (lid/ConformerLangModel.py:383-395) measures."""

    N_A = N_B = 16

    def __init__(self, train: bool, langs: Dict[str, int], lang2vocab: Dict[str, int], items_per_lang: int = 64,
                 seconds: float = 3.0, text_len: int = 20, seed: int = 1234, lang2tokenizer: Dict = None,
                 min_seconds: Optional[float] = None, transcript: str = "random", cache: bool = True,
                 bucket_seconds: Optional[float] = None, **feature):
        """min_seconds: if given, item i lasts U[min_seconds, seconds] (a pure function of (seed, i)): ragged batches;
        bucket_seconds: those durations are rounded UP to a multiple of it (SURVEY 8d's cfg5 recipe: U[1, 10] s in 1 s bins)."""
        if transcript not in ("random", "tones"):
            raise ValueError(f"transcript must be 'random' or 'tones', got {transcript!r}")
        self.train, self.lang2index_dict, self.lang2vocab = train, dict(langs), dict(lang2vocab)
        self.min_samples = None if min_seconds is None else int(min_seconds * feature.get("sr", 16000))
        self.bucket_samples = None if not bucket_seconds else int(bucket_seconds * feature.get("sr", 16000))
        self.feat = _FeatureCfg(**{"speed_shift": False, "pitch_shift": False, "reverb": False, **feature})
        self.type = self.feat.type
        self.n_samples, self.text_len, self.seed = int(seconds * self.feat.sr), text_len, seed
        self.transcript = transcript
        self.lang2tokenizer = lang2tokenizer
        self.datasets, self.samplers = [], []
        for lang in langs:
            s = RandomSamplerWithBase(_SynthLang(lang, items_per_lang))
            s.set_base_value(len(self.datasets))
            self.samplers.append(s)
            self.datasets += [{"locale": lang, "path": f"synthetic://{lang}/{i}"} for i in range(items_per_lang)]
        # Generating an utterance (resonator filter + tone synthesis) costs ~5 ms of CPU: at 64 utterances per batch the
        # DataLoader workers, not the GPU, would set the pace of every epoch.  Items are pure functions of (seed, index), so each
        # worker keeps what it has generated (up to ~1 GB of waveforms per worker); from the second epoch on it only collates.
        self._cache = {} if cache and len(self.datasets) * self.n_samples * 4 <= (1 << 30) else None
        self._pairs = {}
        if transcript == "tones":
            for lang, k in self.lang2index_dict.items():
                if self.lang2vocab[lang] > self.N_A * self.N_B:
                    raise ValueError(f"'tones' transcripts support up to {self.N_A * self.N_B} symbols per language")
                self._pairs[lang] = torch.randperm(self.N_A * self.N_B, generator=torch.Generator().manual_seed(99 + k)).tolist()

    def __len__(self):
        return len(self.datasets)

    def _length(self, index: int, g: torch.Generator) -> int:
        if self.min_samples is None:
            return self.n_samples
        n = self.min_samples + int(torch.randint(0, max(self.n_samples - self.min_samples, 0) + 1, (1,), generator=g))
        if self.bucket_samples:
            n = min(-(-n // self.bucket_samples) * self.bucket_samples, max(self.n_samples, self.bucket_samples))
        return n

    def n_samples_of(self, index: int) -> int:
        """Length of item ``index`` without synthesising it (what a manifest's duration column gives a real corpus)."""
        g = torch.Generator().manual_seed((self.seed * 7919 + index) % (2 ** 31))
        return self._length(index, g)

    def text(self, index: int) -> torch.Tensor:
        item = self.datasets[index]
        g = torch.Generator().manual_seed((self.seed * 104729 + index + 4321) % (2 ** 31))
        return torch.randint(0, self.lang2vocab[item["locale"]], (self.text_len,), generator=g)

    def waveform(self, index: int) -> torch.Tensor:
        from scipy.signal import lfilter
        lang = self.datasets[index]["locale"]
        k = self.lang2index_dict[lang]
        g = torch.Generator().manual_seed((self.seed * 7919 + index) % (2 ** 31))
        n = self._length(index, g)
        x = 0.1 * torch.randn(n, generator=g)
        f0, r = 300.0 + 500.0 * k, 0.97
        w = 2 * math.pi * f0 / self.feat.sr
        y = lfilter([1.0], [1.0, -2 * r * math.cos(w), r * r], x.numpy().astype(np.float64))
        if self.transcript == "tones":
            sr, seg = self.feat.sr, n // self.text_len
            on, edge = int(0.8 * seg), int(0.005 * self.feat.sr)
            tau = np.arange(on) / sr
            env = np.ones(on)
            ramp = 0.5 - 0.5 * np.cos(np.pi * np.arange(edge) / edge)
            env[:edge], env[on - edge:] = ramp, ramp[::-1]
            for j, t in enumerate(self.text(index).tolist()):
                q = self._pairs[lang][t]
                fa, fb = 300.0 + 100.0 * (q // self.N_B), 2200.0 + 200.0 * (q % self.N_B)
                y[j * seg:j * seg + on] += env * (np.sin(2 * math.pi * fa * tau) + np.sin(2 * math.pi * fb * tau))
        return torch.from_numpy(y.astype(np.float32))

    def __getitem__(self, index):
        item = self.datasets[index]
        if self._cache is None:
            return self.waveform(index), self.text(index), item["path"], item["locale"]
        hit = self._cache.get(index)
        if hit is None:
            hit = self._cache[index] = (self.waveform(index), self.text(index))
        return hit[0], hit[1], item["path"], item["locale"]
