"""wav2vec2 backbone (BASELINE config 5) on the GPU against tests/golden/w2v2_step.npz - vectors produced by running
the reference's own encoder code (lid/wavlm/WavLM.py with the relative position embedding off and the padding mask on; see
oracle/gen_golden_w2v2.py for what is run and what is restated) on a ragged batch of 2 / 5 / 1 / 3 s waveforms: hidden-state
mix, last hidden state, logits, and one training-mode step through ``LidModule``'s model surface with every gradient, the
Featurizer's mixing weights included.  bf16 operands on this side, f32 on the reference's."""
import numpy as np
import pytest
import torch

from conftest import load_npz
import ragged_case as rc
import wavlm_case as wc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _module(feature_selection="hidden_states"):
    from lid.LidModule_ASR import LidModule
    from lid.tokenizer import CTCTokenizer
    toks = {k: CTCTokenizer([chr(0x4E00 + i) for i in range(v)]) for k, v in wc.L2V.items()}
    mod = LidModule(optimizer_name="adam", optimizer_param={"lr": 1e-4}, scheduler="none", use_wav2vec=True, lang2vocab=wc.L2V,
                    lang2index_dict=wc.L2I, tokenizer_dict=toks, conformer_linear=True, dropout=0.0, linear_dim=768,
                    hidden_dim=wc.HEAD["hidden_dim"], dim_head=wc.HEAD["dim_head"], num_head=wc.HEAD["num_head"], mask=False,
                    sr=16000, feature_selection=feature_selection, wav2vec_cfg=wc.W2V_CFG)
    m = mod.model
    sd = {"model.featurizer.upstream.model." + k: v for k, v in wc.backbone_weights(2, rel_pos=False).items()}
    sd.update(wc.head_weights())
    if feature_selection == "hidden_states":
        sd["model.featurizer.weights"] = torch.from_numpy(load_npz("w2v2_step.npz")["mix"])
    m.load_state_dict(sd)
    return mod, m.to(DEV)


def test_wav2vec2_eval_features_and_logits_against_the_reference_encoder():
    g = load_npz("w2v2_step.npz")
    mod, m = _module()
    m.eval()
    wavs, _, _, _ = rc.wavlm_batch()
    wavs = [w.to(DEV) for w in wavs]
    n = [int(w.shape[0]) for w in wavs]
    wav = torch.nn.utils.rnn.pad_sequence(wavs, batch_first=True).contiguous()
    bb = m.backbone
    with torch.no_grad():
        mix = bb.forward(wav, n_samples=n, mix_w=m._mix_w()).clone()
        assert bb._ws[tuple(wav.shape)]["klen_host"] == g["klen"].tolist()
        last = bb.forward(wav, n_samples=n).clone()
        logits, _ = m(wavs, 16000, "b")
    for key, got in (("eval_mix", mix[:, ::4]), ("eval_last", last[:, ::4])):
        ref = torch.from_numpy(g[key])
        err, scale = float((got.cpu() - ref).abs().max()), float(ref.abs().max())
        rel = float((got.cpu() - ref).norm() / ref.norm())
        print(f"[w2v2 {key}] max_abs_err={err:.3e} (max |ref| {scale:.2f}) rel_l2={rel:.3e}")
        assert err <= 2e-2 * scale and rel <= 1.5e-2
    ref = torch.from_numpy(g["eval_logits_b"])
    e = float((logits["b"].cpu() - ref).abs().max())
    print(f"[w2v2 eval logits] max_abs_err={e:.3e} (max |ref| {float(ref.abs().max()):.2f})")
    assert e <= 4e-2 * max(1.0, float(ref.abs().max()))
    # no padding -> no mask at all (the reference passes None then), and "last_hidden_state" has no mixing weights
    same = [w[:16000] for w in wavs]
    with torch.no_grad():
        m(same, 16000, "b")
    assert bb._ws[(4, 16000)]["klen_host"] is None
    _, m2 = _module("last_hidden_state")
    m2.eval()
    with torch.no_grad():
        l2, _ = m2(wavs, 16000, "b")
    assert l2["b"].shape == logits["b"].shape and bool(torch.isfinite(l2["b"]).all())


def test_wav2vec2_training_step_against_the_reference_encoder():
    from lid.ConformerLangModel import CtcLossFn
    from test_gpu_wavlm import _cmp_grads
    g = load_npz("w2v2_step.npz")
    mod, m = _module()
    m.train()
    m.freeze_feature_extractor()
    m.unfreeze_tranformer_encoder()
    wavs, texts, wp, tp = rc.wavlm_batch()
    wavs, texts = [w.to(DEV) for w in wavs], texts.to(DEV)
    m.zero_grad()
    logits, _ = m(wavs, 16000, "b")
    z = logits["b"]
    in_len, tg_len = (z.shape[1] * wp).long(), (texts.shape[-1] * tp).long()
    assert in_len.tolist() == g["in_len"].tolist() and tg_len.tolist() == g["tg_len"].tolist()
    per = CtcLossFn.apply(z, texts, in_len.to(DEV), tg_len.to(DEV), 40, m.lidk_engine.k)
    per.mean().backward()
    torch.cuda.synchronize()
    ref = torch.from_numpy(g["train_logits_b"])
    lerr, loss, ref_loss = float((z.detach().cpu() - ref).abs().max()), float(per.mean().detach()), float(g["train_loss"])
    print(f"[w2v2 train step] logits err {lerr:.3e} (max |ref| {float(ref.abs().max()):.2f}); loss {loss:.4f} vs {ref_loss:.4f}")
    assert lerr <= 4e-2 * max(1.0, float(ref.abs().max())) and abs(loss - ref_loss) <= 2e-2 * ref_loss
    params = dict(m.named_parameters())
    n = _cmp_grads(g, lambda name: params[name].grad, "w2v2 train step")
    assert n >= 65 and params["model.featurizer.weights"].grad is not None
    # frozen encoder, gradient stopped at the features: the mixing weights still train (their gradient needs no chain)
    mod2, m2 = _module()
    m2.train_input_norm = False
    m2.train()
    m2.freeze_feature_extractor()
    m2.freeze_tranformer_encoder()
    m2.zero_grad()
    logits, _ = m2(wavs, 16000, "b")
    per = CtcLossFn.apply(logits["b"], texts, in_len.to(DEV), tg_len.to(DEV), 40, m2.lidk_engine.k)
    per.mean().backward()
    p2 = dict(m2.named_parameters())
    gw, rw = p2["model.featurizer.weights"].grad.cpu().double(), torch.from_numpy(g["gs::model.featurizer.weights"]).double()
    cos = float((gw @ rw) / (gw.norm() * rw.norm()))
    print(f"[w2v2 frozen] mixing-weight gradient cosine {cos:.5f}")
    assert cos >= 0.999 and all(p2[k].grad is None for k in p2 if ".encoder." in k)
