"""Pins the CPU oracle against the round-2 reference vectors: the full BASELINE config 2 shape (B = 64, T = 151, 14 languages)
and the reference-trained config 1 checkpoint scored on 72 held-out utterances (tests/golden/cfg2_step.npz, cfg1_trained.npz,
written by oracle/gen_golden_r2.py from the imported reference)."""
import numpy as np
import torch

from conftest import load_npz
import cfg1_trained_case as c1
import cfg2_case as c2
from oracle import conformer as oc
from oracle import features as of
from oracle import metrics as om


def test_oracle_matches_reference_at_cfg2_own_shape():
    g = load_npz("cfg2_step.npz")
    assert bool(g["init_matches_reference"])          # lidk.layout.init_values == the reference's own init under manual_seed(0)
    weights = c2.weights()
    mel, texts = c2.batch()
    lang = str(g["lang"])
    cfg = oc.ModelCfg(lang2vocab=c2.L2V, lang2index=c2.L2I, dropout=0.0, pos_dropout=0.0, **c2.DIMS)
    names = [k for k, v in weights.items() if v.is_floating_point() and "running_" not in k]
    sd = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in weights.items()}
    torch.set_num_threads(8)
    logits, _ = oc.forward(mel, sd, cfg, lang, oc.RunOpts(training=True, keep_layers=[True] * 12))
    out = logits[lang]
    np.testing.assert_allclose(out[:2].detach().numpy(), g["logits_first2"], atol=2e-4 * float(g["logit_absmax"]))
    loss = oc.ctc_loss(out, texts, torch.ones(64), torch.ones(64), blank=40)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * float(g["loss"])
    loss.backward()
    got = {k: sd[k].grad for k in names if sd[k].grad is not None}
    assert sorted(got) == sorted(str(n) for n in g["grad_names"])
    big = float(g["grad_norms"].max())
    for name, ref_norm in zip(g["grad_names"], g["grad_norms"]):
        name = str(name)
        gr = got[name].reshape(-1)
        if ref_norm < 1e-6 * big:
            continue
        assert abs(float(gr.double().norm()) - ref_norm) <= 2e-3 * ref_norm, name
        s = gr[c2.sample_index(name, gr.numel())].double()
        r = torch.from_numpy(g["gs::" + name]).double()
        assert float((s @ r) / (s.norm() * r.norm())) >= 0.99999, name


def test_oracle_scores_the_reference_trained_checkpoint_like_the_reference():
    g = load_npz("cfg1_trained.npz")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w::")}
    cfg = oc.ModelCfg(lang2vocab=c1.L2V, lang2index=c1.L2I, dropout=0.1, hidden_dim=32, **c1.DIMS)
    wav, _, tgt = c1.heldout()
    mel = of.wav2mel(of.normalize_wav(wav), pad=16).transpose(1, 2).contiguous()
    with torch.no_grad():
        _, (asr, lin) = oc.forward(mel, sd, cfg, None, oc.RunOpts(training=False))
    np.testing.assert_allclose(asr.numpy(), g["lid_asr"], atol=2e-5)
    np.testing.assert_allclose(lin.numpy(), g["lid_linear"], atol=2e-5)
    assert np.array_equal(asr.argmax(-1).numpy(), g["argmax"])
    pairs = om.make_pairs([om.score_to_prob(r) for r in asr.tolist()], tgt.tolist())
    assert om.cavg(pairs, 3) == float(g["cavg"]) and float(g["cavg"]) < 0.2
