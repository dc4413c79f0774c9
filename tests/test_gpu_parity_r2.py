"""Parity at the benchmarked configuration and on a trained checkpoint.

* ``cfg2_step.npz``: ONE training step of the REFERENCE model at BASELINE config 2's own shape (B = 64, 3 s, F = 301, T = 151,
  14 languages, 12 x d256) - loss, logits, every gradient tensor (norm + seeded 2048-element sample) - against the HIP engine in
  f32 and in bf16 (the benchmarked dtype), with hipGraph replay on (the compared step is the third: eager, capture, replay).
* ``cfg1_trained.npz``: the REFERENCE model trained on the learnable synthetic corpus, scored on 72 held-out utterances -
  lid_asr, lid_linear, argmax language labels and Cavg against the HIP path (features by the HIP log-mel kernels) from the same
  checkpoint.

Tolerances.  f32: logits 5e-4 * max|logit|, loss 1e-4 rel, gradient norms 2e-3 rel, sampled cosine >= 0.9999.
bf16: logits 6e-2 * max(1, max|logit|), loss 2 % (VERDICT), gradient norms 5 % rel, sampled cosine >= 0.999 on tensors whose
norm is above 1e-3 of the largest (smaller ones are reported and held to 0.99).
"""
import numpy as np
import pytest
import torch

from conftest import load_npz
import cfg1_trained_case as c1
import cfg2_case as c2
from lidk import ops
from lidk.engine import Engine

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def cfg2_inputs():
    return c2.weights(), c2.batch()


@pytest.mark.parametrize("dt,poison", [(torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)])
def test_cfg2_own_shape_training_step_against_the_reference(cfg2_inputs, dt, poison, monkeypatch):
    """poison: the same comparison with LIDK_POISON_SCRATCH=1 - every backward scratch set is overwritten with NaN at the moment
    the data-gradient chain is first allowed to write it (the round-3 race of the late-forked weight-gradient stream
    on the previous scratch set showed up as a 3e-4 drift in one comparison; with the poison a premature write - or a weight
    gradient that still reads a handed-over set - turns into NaN gradients)."""
    if poison:
        monkeypatch.setenv("LIDK_POISON_SCRATCH", "1")
    g = load_npz("cfg2_step.npz")
    weights, (mel, texts) = cfg2_inputs
    lang = str(g["lang"])
    assert mel.shape == (64, 301, 80) and texts.shape == (64, 20) and len(c2.L2V) == 14
    eng = Engine(c2.product_cfg(), act_dtype=dt)
    eng.to(DEV)
    assert eng.graphs.enabled
    wdev = {k: v.to(DEV) for k, v in weights.items()}
    mel_d, texts_d = mel.to(DEV), texts.to(DEV)
    B, T = 64, 151
    in_len = torch.full((B,), T, device=DEV, dtype=torch.long)
    tg_len = torch.full((B,), texts.shape[1], device=DEV, dtype=torch.long)
    per = torch.empty(B, device=DEV)
    dl = torch.empty(B, T, 41, device=DEV)
    ws = torch.empty(ops.ctc_workspace_bytes(B, T, 41, texts.shape[1]) // 4 + 1, device=DEV)
    for rep in range(3):                                         # eager, capture, replay: compare the replayed step
        eng.load_state(wdev)
        eng.zero_grad()
        out = eng.forward(mel_d, lang, training=True, keep_layers=[True] * 12)[lang]
        ops.ctc_loss(out.contiguous(), texts_d, in_len, tg_len, per, dl, ws, 40, grad_scale=1.0 / B)
        eng.backward(dl)
    torch.cuda.synchronize()
    assert sum(1 for st in eng.graphs.state.values() if st[0] is not None) == 3       # encoder forward, head forward, whole backward chain
    f32 = dt == torch.float32
    assert out.shape == (B, T, 41) and bool(torch.isfinite(out).all()) and bool(torch.isfinite(per).all())
    # ---- logits and loss
    scale = max(1.0, float(g["logit_absmax"]))
    lerr = float((out[:2].cpu() - torch.from_numpy(g["logits_first2"])).abs().max())
    loss, ref_loss = float(per.mean()), float(g["loss"])
    perr = float((per.cpu() - torch.from_numpy(g["loss_per_utt"])).abs().max() / ref_loss)
    print(f"[cfg2 B=64 {dt}] logits max_abs_err={lerr:.3e} (max|logit| {scale:.2f}); loss {loss:.4f} vs {ref_loss:.4f} "
          f"(rel {abs(loss - ref_loss) / ref_loss:.2e}); worst per-utterance loss err / mean loss {perr:.2e}")
    assert lerr <= (5e-4 if f32 else 6e-2) * scale
    assert abs(loss - ref_loss) <= (1e-4 if f32 else 2e-2) * ref_loss
    assert perr <= (1e-3 if f32 else 5e-2)
    # ---- every gradient tensor the reference produced: norm and direction (seeded sample)
    names, norms = [str(n) for n in g["grad_names"]], g["grad_norms"]
    big = float(norms.max())
    worst = dict(cos=1.0, cos_small=1.0, nrm=0.0)
    bad = []
    for name, ref_norm in zip(names, norms):
        got = eng.gview(name).reshape(-1)
        ref_s = torch.from_numpy(g["gs::" + name]).double()
        got_s = got[c2.sample_index(name, got.numel()).to(DEV)].cpu().double()
        got_norm = float(got.double().norm())
        if ref_norm < 1e-6 * big:                                # exact-zero gradients (bias in front of BatchNorm): rounding noise
            assert got_norm <= (1e-4 if f32 else 2e-2) * big, (name, got_norm)
            continue
        nrel = abs(got_norm - ref_norm) / ref_norm
        cos = float((got_s @ ref_s) / (got_s.norm() * ref_s.norm() + 1e-300))
        small = ref_norm < 1e-3 * big
        key = "cos_small" if small else "cos"
        worst[key] = min(worst[key], cos)
        worst["nrm"] = max(worst["nrm"], nrel)
        lim = (0.9999 if f32 else (0.99 if small else 0.999))
        if cos < lim or nrel > (2e-3 if f32 else 5e-2):
            bad.append((name, round(cos, 6), round(nrel, 5), float(ref_norm)))
    print(f"[cfg2 B=64 {dt}] {len(names)} gradient tensors: worst sampled cosine {worst['cos']:.6f} "
          f"(tiny tensors {worst['cos_small']:.6f}), worst norm rel err {worst['nrm']:.3e}")
    assert not bad, bad[:10]
    # tensors the reference left without a gradient (other heads, unused featurizer.linear, discriminator) stay exactly zero
    have = set(names)
    for s in eng.specs:
        if s.name not in have:
            assert float(eng.gview(s.name).abs().max()) == 0.0, s.name
    # ---- BatchNorm running statistics after the step
    for k in g:
        if k.startswith("bn::"):
            np.testing.assert_allclose(eng.buffers[k[4:]].cpu().numpy(), g[k], rtol=(2e-4 if f32 else 2e-2),
                                       atol=(2e-5 if f32 else 2e-3), err_msg=k)


def _trained_module(dt):
    from lid.LidModule_ASR_Supervised import LidSuperviseModule
    from lid.tokenizer import CTCTokenizer
    toks = {k: CTCTokenizer([chr(0x4E00 + i) for i in range(v)]) for k, v in c1.L2V.items()}
    mod = LidSuperviseModule(optimizer_name="novograd", optimizer_param={"lr": 0.01}, scheduler="none", lang2index_dict=c1.L2I,
                             tokenizer_dict=toks, lang2vocab=c1.L2V, dropout=0.1, linear_dim=64, **c1.DIMS)
    mod.model.set_compute_dtype(dt)
    g = load_npz("cfg1_trained.npz")
    mod.model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w::")})   # checkpoint path
    mod.model.to(DEV).eval()
    return mod, g


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_trained_checkpoint_lid_scores_argmax_and_cavg_against_the_reference(dt):
    """72 held-out utterances, scored one at a time (as the reference's val_loop does) AND as one batch, from raw waveforms
    through the HIP feature kernels: lid_asr, lid_linear (the LangDiscriminator MLP, row a15), argmax labels, Cavg."""
    from lid.audio_processor import WaveBatch
    from lid.eer import CAvg
    from oracle import metrics as om
    mod, g = _trained_module(dt)
    wav, _, tgt = c1.heldout()
    feats = WaveBatch(wav.clone(), None, pad=16, preemph=False).to(DEV).to_mel()
    assert feats.shape == (72, 101, 80)
    with torch.no_grad():
        _, (asr_b, lin_b) = mod.model(feats, 16000, None)
        rows = [mod.model(feats[i:i + 1].contiguous(), 16000, None)[1] for i in range(feats.shape[0])]
    asr = torch.cat([r[0] for r in rows]).cpu()
    lin = torch.cat([r[1] for r in rows]).cpu()
    assert torch.equal(asr, asr_b.cpu()) and torch.equal(lin, lin_b.cpu())       # batching does not change an utterance's score
    ref_asr, ref_lin = torch.from_numpy(g["lid_asr"]), torch.from_numpy(g["lid_linear"])
    f32 = dt == torch.float32
    e_asr, e_lin = float((asr - ref_asr).abs().max()), float((lin - ref_lin).abs().max())
    srt = ref_asr.sort(-1).values
    margin = srt[:, -1] - srt[:, -2]
    same = asr.argmax(-1) == ref_asr.argmax(-1)
    print(f"[trained cfg1 {dt}] lid_asr max_abs_err={e_asr:.3e} lid_linear max_abs_err={e_lin:.3e}; argmax equal on "
          f"{int(same.sum())}/72 (reference top-2 margins: min {float(margin.min()):.4f}, median {float(margin.median()):.4f})")
    # bf16: the score is a mean over NON-BLANK frames, so a frame whose blank / non-blank decision flips under bf16 rounding
    # moves it by a discrete step (measured 1.5e-2 on scores of magnitude 0.01-0.2); the labels below are still exact
    assert e_asr <= (2e-4 if f32 else 2.5e-2) and e_lin <= (2e-4 if f32 else 1e-2)
    assert bool(same.all()), [(i, float(margin[i])) for i in (~same).nonzero().flatten().tolist()]   # labels bit-exact, all 72
    assert torch.equal(asr.argmax(-1), torch.from_numpy(g["argmax"]))
    metric = CAvg(num_class=3)
    for row, t in zip(asr.tolist(), tgt.tolist()):
        metric.update([mod.score_to_prob(row)], [t])
    cavg = metric.compute()
    ref_pairs = om.make_pairs([om.score_to_prob(r) for r in ref_asr.tolist()], tgt.tolist())
    print(f"[trained cfg1 {dt}] Cavg HIP {cavg} | reference fixture {float(g['cavg'])} | oracle metric on reference scores {om.cavg(ref_pairs, 3)}")
    assert cavg == float(g["cavg"]) == om.cavg(ref_pairs, 3)                       # equal to 4 decimals (both are rounded to 4)
    assert cavg < 0.2                                                            # and far below chance (0.5)
    # greedy transcripts of the first utterance agree with the reference's logits
    with torch.no_grad():
        out, _ = mod.model(feats[:1].contiguous(), 16000, None)
    for lang in c1.L2V:
        ref = torch.from_numpy(g["logits0_" + lang])
        if f32:
            assert float((out[lang][0].cpu() - ref).abs().max()) <= 5e-3
        agree = float((out[lang][0].cpu().argmax(-1) == ref.argmax(-1)).float().mean())
        assert agree >= (1.0 if f32 else 0.97), (lang, agree)


def test_test_stage_result_files_from_the_trained_checkpoint(tmp_path):
    """stage: test plumbing: every held-out utterance through LidSuperviseModule.infer_tensor (HIP features, all
    heads, device-side greedy CTC decode), LID accuracy and Cavg far from chance on the reference-trained checkpoint, result
    files in the reference's TSV layouts."""
    import csv
    from lid.test_supervised import score_dataset
    mod, g = _trained_module(torch.float32)
    ds = c1._ds(False, c1.HELD_ITEMS, 1235)
    ds.lang2tokenizer = mod.tokenizer_dict
    res = score_dataset(mod, ds, str(tmp_path / "res" / "result.txt"))
    print(f"[test stage] acc {res['acc']:.4f} cavg {res['cavg']}  (fixture, pad-16 features: acc {float(g['accuracy']):.4f} cavg {float(g['cavg'])})")
    # infer_tensor computes pad = 0 features like the reference's (lid/LidModule_ASR_Supervised.py:229-239) while the fixture was
    # scored on the training-time pad = 16 features: close, not equal
    assert res["cavg"] < 0.2 and res["acc"] > 0.7
    rows = list(csv.DictReader(open(tmp_path / "res" / "result.txt"), delimiter="\t"))
    assert len(rows) == 72 and set(rows[0]) == {"wav_name", "text"}
    for lang in c1.L2V:
        rows = list(csv.DictReader(open(tmp_path / "res" / f"{lang}.csv"), delimiter="\t"))
        assert len(rows) == c1.HELD_ITEMS and list(rows[0]) == ["true", "pred", "a", "b", "c"]
        cer = sum(r["true"] != r["pred"] for r in rows) / len(rows)
        assert cer < 0.8                                    # the trained heads do transcribe their own language
