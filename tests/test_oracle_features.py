"""Known-answer tests for oracle/features.py.  torchaudio (where the reference's mel arithmetic lives) is not in
this image, so these pin the restatement analytically and against an independent float64 numpy DFT."""
import math

import numpy as np
import torch

from oracle import features as of


def test_stft_matches_direct_definition():
    g = torch.Generator().manual_seed(0)
    wav = torch.randn(1, 4000, generator=g)
    for pad in (0, 16):
        a = of.power_spectrogram(wav, pad=pad)[0].numpy()
        b = of.power_spectrogram_np(wav[0].numpy(), pad=pad)
        assert a.shape == b.shape == (257, of.num_frames(4000, pad))
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=1e-3 * b.max())
        assert np.median(np.abs(a - b) / (b + 1e-6)) < 1e-5


def test_pure_tone_lands_in_expected_bin_and_mel():
    sr, f0 = 16000, 1000.0
    t = torch.arange(16000) / sr
    wav = torch.sin(2 * math.pi * f0 * t)[None]
    spec = of.power_spectrogram(wav)[0]
    assert int(spec[:, 50].argmax()) == round(f0 / (sr / 512))            # bin 32
    # periodic hann of 400: coherent gain = sum(w) = 200 -> |X|^2 at the peak = (A/2 * 200)^2 = 1e4
    np.testing.assert_allclose(float(spec[32, 50]), 1.0e4, rtol=1e-3)
    mel = of.wav2mel(wav)[0]
    m = 2595 * math.log10(1 + f0 / 700)
    edges = np.linspace(0, 2595 * math.log10(1 + 8000 / 700), 82)
    assert abs(int(mel[:, 50].argmax()) - (int(np.searchsorted(edges, m)) - 1)) <= 1
    assert mel.shape == (80, 101)
    assert float(mel.max() - mel.min()) <= 80.0 + 1e-4                        # top_db floor


def test_mel_filterbank_shape_and_triangles():
    fb = of.melscale_fbanks()
    assert fb.shape == (257, 80) and float(fb.min()) >= 0 and float(fb.max()) <= 1.0
    assert float(fb[0].sum()) == 0.0 and float(fb[256].sum()) == 0.0          # DC and Nyquist rows are zero
    centres = (fb * torch.arange(257)[:, None]).sum(0) / fb.sum(0).clamp_min(1e-9)
    nz = fb.sum(0) > 0
    assert bool((centres[nz][1:] > centres[nz][:-1]).all())


def test_amplitude_to_db_edges():
    x = torch.zeros(1, 80, 7)
    np.testing.assert_allclose(of.amplitude_to_db(x).numpy(), -100.0)
    x[0, 3, 2] = 1.0                                                          # 0 dB peak -> floor at -80
    y = of.amplitude_to_db(x)
    assert float(y.max()) == 0.0 and float(y.min()) == -80.0
    xb = torch.stack([x[0], 1e-3 * torch.ones(80, 7)])                        # floor is per utterance
    yb = of.amplitude_to_db(xb)
    assert float(yb[0].min()) == -80.0 and abs(float(yb[1].min()) + 30.0) < 1e-5


def test_normalize_and_preemphasis():
    g = torch.Generator().manual_seed(1)
    w = 3.0 + 2.0 * torch.randn(2, 1000, generator=g)
    n = of.normalize_wav(w)
    assert n.mean(-1).abs().max() < 1e-5 and (n.std(-1) - 1).abs().max() < 1e-4
    y = of.dither_preemphasis(w)
    assert torch.equal(y[:, 0], w[:, 0])
    np.testing.assert_allclose(y[:, 1:].numpy(), (w[:, 1:] - 0.97 * w[:, :-1]).numpy(), rtol=1e-6)


def test_specaug_spans_follow_torchaudio_draw_order():
    g = torch.Generator().manual_seed(3)
    spans = of.draw_specaug_spans(301, 80, t_mask=0.05, f_mask=12, mask_times=2, gen=g)
    g2 = torch.Generator().manual_seed(3)
    u = [float(torch.rand(1, generator=g2)) for _ in range(8)]
    exp = []
    for r in range(2):
        v = np.float32(u[4 * r]) * np.float32(15)
        s = np.float32(u[4 * r + 1]) * (np.float32(301) - v)
        fv = np.float32(u[4 * r + 2]) * np.float32(12)
        fs = np.float32(u[4 * r + 3]) * (np.float32(80) - fv)
        exp.append((int(s), int(s) + int(v), int(fs), int(fs) + int(fv)))
    assert spans == exp
    spec = torch.full((80, 301), -5.0)
    out = of.apply_specaug(spec, spans)
    t0, t1, f0, f1 = spans[0]
    assert (out[:, t0:t1] == 0).all() and (out[f0:f1] == 0).all()
    assert int((out == 0).sum()) <= sum((a[1] - a[0]) * 80 + (a[3] - a[2]) * 301 for a in spans)


def test_collate_layout():
    a, b = torch.randn(80, 30), torch.randn(80, 21)
    x, pct = of.collate_mel([a, b])
    assert x.shape == (2, 30, 80) and torch.equal(x[1, 21:], torch.zeros(9, 80))
    np.testing.assert_allclose(pct.numpy(), [1.0, 21 / 30], rtol=1e-6)


def test_speed_perturbation_restatement_against_scipy_and_the_polyphase_table():
    """oracle.features.speed_perturb_np (direct band-limited interpolation) against scipy.signal.resample_poly (independent
    polyphase implementation, different window) and against the product's host-built polyphase table (lidk.ops.resample_taps)."""
    from scipy.signal import resample_poly
    from lidk.ops import resample_taps, speed_out_len
    rng = np.random.RandomState(0)
    t = np.arange(4000) / 16000.0
    x = np.sin(2 * math.pi * 440 * t) + 0.5 * np.sin(2 * math.pi * 3000 * t) + 0.05 * rng.randn(4000)
    assert np.array_equal(of.speed_perturb_np(x, 1, 1), x)
    for p, q in ((11, 10), (9, 10)):
        y = of.speed_perturb_np(x, p, q)
        assert len(y) == speed_out_len(len(x), p, q) == int(len(x) * q / p + 0.5)
        r = resample_poly(x, q, p)
        m = min(len(r), len(y))
        assert np.abs(r[:m] - y[:m])[100:m - 100].max() < 1e-2 and np.corrcoef(r[100:m - 100], y[100:m - 100])[0, 1] > 0.99999
        taps, left = resample_taps(p, q)
        assert taps.shape[0] == q and np.allclose(taps.sum(1), 1.0)
        n = np.arange(len(y))
        base, ph = (n * p) // q - left, (n * p) % q
        yp = np.zeros(len(y))
        for i in range(len(y)):
            k = base[i] + np.arange(taps.shape[1])
            ok = (k >= 0) & (k < len(x))
            yp[i] = (taps[ph[i]][ok] * x[k[ok]]).sum()
        assert np.abs(yp - y).max() < 1e-10
    # a pure tone comes out at v times its frequency
    tone = np.sin(2 * math.pi * 1000 * np.arange(16000) / 16000.0)
    y = of.speed_perturb_np(tone, 11, 10)
    spec = np.abs(np.fft.rfft(y[:8192] * np.hanning(8192)))
    assert abs(spec.argmax() * 16000 / 8192 - 1100.0) < 4.0
