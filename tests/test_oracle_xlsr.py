"""Pins the oracle's Large-model flags (oracle/wavlm.py: layer_norm conv extractor with bias, pre-LN layers + closing LayerNorm,
waveform layer-norm, key padding mask, hidden-state mix; WavLM-Large: the relative-bias gate on LN1's output) against runs of the
REFERENCE's own lid/wavlm/WavLM.py classes with those flags (tests/golden/xlsr_step.npz, wavlm_large_step.npz, written by
oracle/gen_golden_xlsr.py)."""
import numpy as np
import torch

from conftest import load_npz
import ragged_case as rc
import wavlm_case as wc
from oracle import conformer as oc
from oracle import wavlm as ow


def _hcfg():
    return oc.ModelCfg(lang2vocab=wc.L2V, lang2index=wc.L2I, n_blocks=0, encoder_dim=1024, last_dim_head=wc.HEAD_LARGE["dim_head"],
                       last_heads=wc.HEAD_LARGE["num_head"], dropout=0.0, hidden_dim=wc.HEAD_LARGE["hidden_dim"])


def _check_grads(g, tensors, prefix=""):
    for name, norm in zip(g[prefix + "grad_names"], g[prefix + "grad_norms"]):
        name = str(name)
        got = tensors[name].grad.reshape(-1)
        assert abs(float(got.double().norm()) - norm) <= 2e-3 * norm + 1e-5, name
        idx = torch.randperm(got.numel(), generator=torch.Generator().manual_seed(wc._seed(name)))[:2048].sort().values \
            if got.numel() > 2048 else torch.arange(got.numel())
        np.testing.assert_allclose(got[idx].numpy(), g[prefix + "gs::" + name], atol=2e-3 * norm / got.numel() ** 0.5 + 1e-5, rtol=5e-3,
                                   err_msg=name)


def test_oracle_xlsr_eval_and_full_training_step_match_the_reference_classes():
    g = load_npz("xlsr_step.npz")
    wavs, texts, wp, tp = rc.wavlm_batch()
    sd = wc.backbone_weights_cfg(wc.XLSR_CFG)
    heads = wc.head_weights(encoder_dim=1024, hidden_dim=wc.HEAD_LARGE["hidden_dim"])
    mix = torch.from_numpy(g["mix"]).clone()
    with torch.no_grad():
        normed = torch.nn.utils.rnn.pad_sequence([torch.nn.functional.layer_norm(w, w.shape) for w in wavs], batch_first=True)
        np.testing.assert_allclose(normed[:, ::97].numpy(), g["wav_norm"], atol=1e-6)
        conv = ow.feature_extractor(normed, sd, mode="layer_norm").transpose(1, 2)
        np.testing.assert_allclose(conv[:, ::4, ::2].numpy(), g["conv"], atol=5e-5)
        assert ow.conv_out_lengths([w.shape[0] for w in wavs]).tolist() == g["klen"].tolist()
        last = ow.wav2vec2_features(wavs, sd, wc.XLSR_CFG)
        np.testing.assert_allclose(last[:, ::4].numpy(), g["eval_last"], atol=1e-4)
        feat = ow.wav2vec2_features(wavs, sd, wc.XLSR_CFG, mix)
        np.testing.assert_allclose(feat[:, ::4].numpy(), g["eval_mix"], atol=1e-4)
        np.testing.assert_allclose(oc.head(feat, heads, _hcfg(), "b", oc.RunOpts()).numpy(), g["eval_logits_b_mix"], atol=5e-4)
    # the training-mode step with everything un-frozen (autograd through the oracle against the reference's)
    for v in sd.values():
        v.requires_grad_(True)
    heads = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in heads.items()}
    mix.requires_grad_(True)
    z = oc.head(ow.wav2vec2_features(wavs, sd, wc.XLSR_CFG, mix), heads, _hcfg(), "b", oc.RunOpts(training=True))
    np.testing.assert_allclose(z.detach().numpy(), g["full::train_logits_b"], atol=5e-4)
    per = torch.nn.functional.ctc_loss(torch.log_softmax(z, -1).transpose(0, 1), texts, (z.shape[1] * wp).long(),
                                       (texts.shape[-1] * tp).long(), blank=40, reduction="none", zero_infinity=True)
    per.mean().backward()
    assert abs(float(per.mean().detach()) - float(g["full::train_loss"])) < 2e-3
    tensors = {"model.featurizer.upstream.model." + k: v for k, v in sd.items()}
    tensors.update(heads)
    tensors["model.featurizer.weights"] = mix
    _check_grads(g, tensors, "full::")
    assert len(g["full::grad_names"]) >= 95


def test_oracle_wavlm_large_matches_the_reference():
    g = load_npz("wavlm_large_step.npz")
    wavs, texts, wp, tp = rc.wavlm_batch()
    sd = wc.backbone_weights_cfg(wc.WAVLM_LARGE_CFG)
    heads = wc.head_weights(encoder_dim=1024, hidden_dim=wc.HEAD_LARGE["hidden_dim"])
    wav = torch.nn.utils.rnn.pad_sequence(list(wavs), batch_first=True)
    with torch.no_grad():
        last = ow.backbone(wav, sd, wc.WAVLM_LARGE_CFG)
    np.testing.assert_allclose(last[:, ::4].numpy(), g["eval_last"], atol=1e-4)
