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
