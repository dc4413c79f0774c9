"""Pins the CPU oracle against the round-3 reference-run vectors (tests/golden/frontend_ref.npz, ragged_step.npz, written by
oracle/gen_golden_r3.py from the imported reference): waveform normalisation and dither + pre-emphasis, and the forward + CTC
loss of the variable-length (1-10 s, zero-padded, unmasked) batch."""
import numpy as np
import torch

from conftest import load_npz
import cfg2_case as c2
import ragged_case as rc
from oracle import conformer as oc
from oracle import features as of


def test_oracle_normalize_and_dither_preemphasis_match_the_reference_run():
    g = load_npz("frontend_ref.npz")
    for case in range(3):
        wav = torch.from_numpy(g[f"wav{case}"])
        np.testing.assert_allclose(of.normalize_wav(wav).numpy(), g[f"norm{case}"], rtol=0, atol=1e-6)
        aug = of.dither_preemphasis(torch.from_numpy(g[f"norm{case}"]), torch.from_numpy(g[f"noise{case}"]))
        np.testing.assert_allclose(aug.numpy(), g[f"aug{case}"], rtol=0, atol=1e-7)


def test_oracle_forward_and_ctc_on_the_ragged_batch_match_the_reference():
    g = load_npz("ragged_step.npz")
    b = rc.conformer_batch()
    assert b["frames"] == list(g["frames"])
    weights = c2.weights()
    lang = str(g["lang"])
    cfg = oc.ModelCfg(lang2vocab=c2.L2V, lang2index=c2.L2I, dropout=0.0, pos_dropout=0.0, **c2.DIMS)
    torch.set_num_threads(8)
    with torch.no_grad():
        sd = {k: v.clone() for k, v in weights.items()}
        logits, _ = oc.forward(b["mel"], sd, cfg, lang, oc.RunOpts(training=True, keep_layers=[True] * 12))
        out = logits[lang]
        pick = [int(i) for i in g["pick"]]
        np.testing.assert_allclose(out[pick].numpy(), g["logits_pick"], atol=2e-4 * float(g["logit_absmax"]))
        loss = oc.ctc_loss(out, b["texts"], b["wav_percents"], b["text_percents"], blank=40)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * float(g["loss"])
