"""Variable-length (1-10 s, bucketed to 1 s: SURVEY 8d's cfg5 recipe) batches as pure functions of seeds.

``conformer_batch``: ten utterances of language l03 lasting 1 .. 10 s, collated the way the reference's
``MergedDataset.collate_fn`` does (lid/raw_datasets.py:345-365): per-utterance log-mel, ZERO-padded in the dB domain to the
longest, ``wav_percents = F_i / F_max``, texts zero-padded with ``text_percents = L_i / (L_max + 1e-9)``.  The reference never
masks the padding: padded frames attend, are attended and enter the BatchNorm statistics; only the CTC lengths
know the true sizes.  ``oracle/gen_golden_r3.py`` feeds the REFERENCE model with this batch; the GPU test feeds the HIP engine.

``wavlm_batch``: four raw waveforms of 1 / 2 / 3 / 5 s for the WavLM wrapper, which pads the list itself
(lid/WavLMMutiLangModel.py:268-270)."""
import math

import torch

import cfg2_case as c2

SECONDS = [7, 1, 10, 3, 5, 2, 8, 4, 9, 6]             # one utterance per 1 s bucket, shuffled
PAD = 16
LANG = c2.LANG


def _utterance(ds_cache, seconds: int, i: int):
    from lid.raw_datasets import SyntheticMergedDataset
    key = seconds
    if key not in ds_cache:
        ds_cache[key] = SyntheticMergedDataset(False, c2.L2I, c2.L2V, items_per_lang=len(SECONDS), seconds=float(seconds),
                                               text_len=2 * seconds + 1, seed=977 + seconds, transcript="tones", type="mel",
                                               pad=PAD)
    ds = ds_cache[key]
    base = c2.L2I[LANG] * len(SECONDS)
    return ds.waveform(base + i), ds.text(base + i)


def conformer_batch():
    """-> dict(mel (10, 1001, 80) f32 dB zero-padded, texts (10, 21) int64 zero-padded, wav_percents (10,), text_percents (10,),
    frames list)."""
    from oracle import features as of
    cache, mels, texts = {}, [], []
    for i, s in enumerate(SECONDS):
        wav, txt = _utterance(cache, s, i)
        mel = of.wav2mel(of.normalize_wav(wav[None]), pad=PAD)[0]          # (80, F_i)
        mels.append(mel.transpose(0, 1).contiguous())
        texts.append(txt)
    frames = [m.shape[0] for m in mels]
    mel = torch.nn.utils.rnn.pad_sequence(mels, batch_first=True)
    tx = torch.nn.utils.rnn.pad_sequence(texts).transpose(1, 0).contiguous()
    wav_pct = torch.FloatTensor([f / max(frames) for f in frames])
    txt_pct = torch.FloatTensor([t.shape[-1] / (tx.shape[1] + 1e-9) for t in texts])
    return dict(mel=mel.contiguous(), texts=tx, wav_percents=wav_pct, text_percents=txt_pct, frames=frames)


WAVLM_SECONDS = [2.0, 5.0, 1.0, 3.0]


def wavlm_batch():
    """-> (list of 4 raw waveforms (L_i,), texts (4, 12) int64 zero-padded, wav_percents, text_percents)."""
    g = torch.Generator().manual_seed(31337)
    wavs, texts = [], []
    for b, s in enumerate(WAVLM_SECONDS):
        n = int(s * 16000)
        t = torch.arange(n) / 16000.0
        x = 0.3 * torch.randn(n, generator=g) + 0.5 * torch.sin(2 * math.pi * (180.0 * (b + 1)) * t)
        wavs.append(x)
        texts.append(torch.randint(0, 40, (int(2 * s) + 2,), generator=g))
    tx = torch.nn.utils.rnn.pad_sequence(texts).transpose(1, 0).contiguous()
    longest = max(w.shape[0] for w in wavs)
    wav_pct = torch.FloatTensor([w.shape[0] / longest for w in wavs])
    txt_pct = torch.FloatTensor([t.shape[-1] / (tx.shape[1] + 1e-9) for t in texts])
    return wavs, tx, wav_pct, txt_pct


def wavlm_long_batch():
    """(2, 208000) f32: two 13 s waveforms (T = 649 frames after the conv stack: the reference WavLM confs' max_duration)."""
    g = torch.Generator().manual_seed(4711)
    n = 13 * 16000
    t = torch.arange(n) / 16000.0
    x = 0.3 * torch.randn(2, n, generator=g)
    for b in range(2):
        x[b] += 0.5 * torch.sin(2 * math.pi * (150.0 * (b + 1)) * t) * (0.5 + 0.5 * torch.sin(2 * math.pi * 0.7 * t))
    return x
