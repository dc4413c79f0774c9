"""Pins oracle/wavlm.py (torch-CPU restatement of the WavLM backbone + WavLMMutiLangModel heads) against outputs of the REFERENCE
itself (tests/golden/wavlm_fwd.npz, wavlm_model.npz, written by oracle/gen_golden_wavlm.py from the imported lid/wavlm/WavLM.py and
lid/WavLMMutiLangModel.py)."""
import numpy as np
import torch

from conftest import load_npz
import wavlm_case as wc
from oracle import conformer as oc
from oracle import wavlm as ow


def test_oracle_backbone_stages_match_the_reference():
    g = load_npz("wavlm_fwd.npz")
    taps = {}
    with torch.no_grad():
        out = ow.backbone(wc.waveforms(), wc.backbone_weights(), wc.CFG, taps)
    for key in ("conv", "proj", "enc_in", "layer0", "layer1"):
        np.testing.assert_allclose(taps[key].numpy(), g[key], atol=5e-5, err_msg=key)
    np.testing.assert_allclose(out.numpy(), g["features"], atol=5e-5)
    assert float((ow.position_bias(wc.backbone_weights(), 49) - torch.from_numpy(g["pos_bias0"])).abs().max()) == 0.0


def test_oracle_full_model_matches_the_reference():
    g = load_npz("wavlm_model.npz")
    hcfg = oc.ModelCfg(lang2vocab=wc.L2V, lang2index=wc.L2I, n_blocks=0, encoder_dim=768, last_dim_head=wc.HEAD["dim_head"],
                       last_heads=wc.HEAD["num_head"], dropout=0.0, hidden_dim=wc.HEAD["hidden_dim"])
    wav = wc.waveforms()
    with torch.no_grad():
        logits, (lid_asr, lid_linear) = ow.model_forward([wav[i] for i in range(wav.shape[0])], wc.backbone_weights(),
                                                          wc.head_weights(), wc.CFG, hcfg)
    for lang in wc.L2V:
        np.testing.assert_allclose(logits[lang].numpy(), g["logits_" + lang], atol=2e-4, err_msg=lang)
    np.testing.assert_allclose(lid_asr.numpy(), g["lid_asr"], atol=2e-5)
    np.testing.assert_allclose(lid_linear.numpy(), g["lid_linear"], atol=2e-5)


def _oracle_step(fixture, trainable, masks=None):
    g = load_npz(fixture)
    hcfg = oc.ModelCfg(lang2vocab=wc.L2V, lang2index=wc.L2I, n_blocks=0, encoder_dim=768, last_dim_head=wc.HEAD["dim_head"],
                       last_heads=wc.HEAD["num_head"], dropout=0.0, hidden_dim=wc.HEAD["hidden_dim"])
    sd = wc.backbone_weights()
    for k, v in sd.items():
        v.requires_grad_(k.startswith(trainable))
    heads = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in wc.head_weights().items()}
    wav, texts = wc.waveforms(), wc.texts()
    logits, _ = ow.model_forward([wav[i] for i in range(wav.shape[0])], sd, heads, wc.CFG, hcfg, lang="b",
                                 opts=oc.RunOpts(training=True), masks=masks)
    z = logits["b"]
    np.testing.assert_allclose(z.detach().numpy(), g["train_logits_b"], atol=3e-4)
    per = torch.nn.functional.ctc_loss(torch.log_softmax(z, -1).transpose(0, 1), texts, torch.full((3,), z.shape[1]),
                                       torch.full((3,), texts.shape[1]), blank=40, reduction="none", zero_infinity=True)
    per.mean().backward()
    assert abs(float(per.mean().detach()) - float(g["train_loss"])) < 1e-3
    for name, norm in zip(g["grad_names"], g["grad_norms"]):
        name = str(name)
        t = sd[name[len("model.featurizer.model."):]] if name.startswith("model.featurizer.model.") else heads[name]
        got = t.grad.reshape(-1)
        assert abs(float(got.double().norm()) - norm) <= 2e-3 * norm + 1e-5, name
        idx = torch.randperm(got.numel(), generator=torch.Generator().manual_seed(wc._seed(name)))[:2048].sort().values \
            if got.numel() > 2048 else torch.arange(got.numel())
        # + 1e-5: gradients that are zero in exact arithmetic (a bias in front of BatchNorm)
        np.testing.assert_allclose(got[idx].numpy(), g["gs::" + name], atol=2e-3 * norm / got.numel() ** 0.5 + 1e-5, rtol=5e-3,
                                   err_msg=name)
    return len(g["grad_names"])


def test_oracle_finetune_gradients_match_the_reference():
    """autograd through the oracle (encoder parameters + the backbone's layer_norm trainable, extractor frozen) against the
    reference's own training-mode step with the transformer encoder un-frozen (wavlm_finetune.npz)."""
    assert _oracle_step("wavlm_finetune.npz", ("encoder.", "layer_norm.")) == 79


def test_oracle_frozen_masked_step_matches_the_reference():
    """The reference's default first-epoch regime (extractor + encoder frozen, span masking on): the spans this repo's host
    code draws under the fixture's numpy seed (time spans first, then channel spans - WavLM.apply_mask's order) put through the
    oracle reproduce the reference's logits, loss and the gradients of the heads, layer_norm and mask_emb."""
    from lidk.wavlm import span_mask
    np.random.seed(wc.MASK_SEED)
    pad = torch.zeros(wc.B, 49, dtype=torch.bool)                # the model always passes a padding mask (here: nothing padded)
    tm = span_mask((wc.B, 49), pad, wc.MASK_PROB, 10, min_masks=2)
    cm = span_mask((wc.B, 768), None, wc.MASK_CHANNEL_PROB, 10)
    assert _oracle_step("wavlm_frozen_masked.npz", ("layer_norm.", "mask_emb"), masks=(tm, cm)) == 36
