"""The LARGE backbones on the GPU - wav2vec2 Large / XLS-R 300M (what every wav2vec conf of the reference loads:
lid/conf/xf_asr_wav2vec.yaml:6,12) and WavLM Large (lid/conf/xf_asr_extra_finetune.yaml:12) - against tests/golden/xlsr_step.npz
and wavlm_large_step.npz: vectors produced by running the reference's own lid/wavlm/WavLM.py classes with
``extractor_mode="layer_norm"``, ``conv_bias``, ``layer_norm_first`` (oracle/gen_golden_xlsr.py says what is run and what is
restated) on the ragged 2 / 5 / 1 / 3 s batch.  bf16 operands on this side, f32 on the reference's."""
import numpy as np
import pytest
import torch

from conftest import load_npz
import ragged_case as rc
import wavlm_case as wc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _module(feature_selection="hidden_states", **kw):
    from lid.LidModule_ASR import LidModule
    from lid.tokenizer import CTCTokenizer
    toks = {k: CTCTokenizer([chr(0x4E00 + i) for i in range(v)]) for k, v in wc.L2V.items()}
    mod = LidModule(optimizer_name="adam", optimizer_param={"lr": 1e-4}, scheduler="none", use_wav2vec=True, lang2vocab=wc.L2V,
                    lang2index_dict=wc.L2I, tokenizer_dict=toks, conformer_linear=True, dropout=0.0, linear_dim=1024,
                    hidden_dim=wc.HEAD_LARGE["hidden_dim"], dim_head=wc.HEAD_LARGE["dim_head"], num_head=wc.HEAD_LARGE["num_head"],
                    mask=False, sr=16000, feature_selection=feature_selection, wav2vec_cfg=wc.XLSR_CFG, **kw)
    m = mod.model
    sd = {"model.featurizer.upstream.model." + k: v for k, v in wc.backbone_weights_cfg(wc.XLSR_CFG).items()}
    sd.update(wc.head_weights(encoder_dim=1024, hidden_dim=wc.HEAD_LARGE["hidden_dim"]))
    if feature_selection == "hidden_states":
        sd["model.featurizer.weights"] = torch.from_numpy(load_npz("xlsr_step.npz")["mix"])
    m.load_state_dict(sd)
    return mod, m.to(DEV)


def _close(tag, got, ref, abs_rel=2e-2, l2=1.5e-2):
    ref = torch.from_numpy(ref)
    err, scale = float((got.cpu().float() - ref).abs().max()), float(ref.abs().max())
    rel = float((got.cpu().float() - ref).norm() / ref.norm())
    print(f"[xlsr {tag}] max_abs_err={err:.3e} (max |ref| {scale:.2f}) rel_l2={rel:.3e}")
    assert err <= abs_rel * max(1.0, scale) and rel <= l2, tag


def test_waveform_layer_norm_matches_the_reference_call():
    """task.normalize (wav2vec2_expert.py:71-72): F.layer_norm(wav, wav.shape) per utterance over its own samples."""
    from lidk import ops
    g = load_npz("xlsr_step.npz")
    wavs, _, _, _ = rc.wavlm_batch()
    n = torch.tensor([w.shape[0] for w in wavs], dtype=torch.int32, device=DEV)
    wav = torch.nn.utils.rnn.pad_sequence(wavs, batch_first=True).to(DEV).contiguous()
    out = ops.wav_layernorm(wav, n_samples=n)
    err = float((out[:, ::97].cpu() - torch.from_numpy(g["wav_norm"])).abs().max())
    print(f"[xlsr wav layer-norm] max_abs_err={err:.3e}")
    assert err <= 2e-5
    same = ops.wav_layernorm(wav[:, :16000].contiguous())                       # no lengths: statistics over the whole row
    ref = torch.nn.functional.layer_norm(wav[:, :16000].cpu(), (16000,))
    assert float((same.cpu() - ref).abs().max()) <= 2e-5


def test_xlsr_eval_features_and_logits_against_the_reference_classes():
    g = load_npz("xlsr_step.npz")
    mod, m = _module()
    m.eval()
    wavs, _, _, _ = rc.wavlm_batch()
    wavs = [w.to(DEV) for w in wavs]
    n = [int(w.shape[0]) for w in wavs]
    wav = torch.nn.utils.rnn.pad_sequence(wavs, batch_first=True).contiguous()
    bb = m.backbone
    assert bb.ln_extractor and bb.conv_bias and bb.pre_ln and bb.normalize and bb.pad_mask
    with torch.no_grad():
        taps = {}
        bb.forward(wav, n_samples=n, mix_w=m._mix_w(), taps=taps)
        _close("conv stack", taps["conv"][:, ::4, ::2], g["conv"])
        mix = bb.forward(wav, n_samples=n, mix_w=m._mix_w()).clone()
        assert bb._ws[tuple(wav.shape)]["klen_host"] == g["klen"].tolist()
        last = bb.forward(wav, n_samples=n).clone()
        again = bb.forward(wav, n_samples=n).clone()                             # second use of the shape: the captured graphs
        logits, _ = m(wavs, 16000, "b")
    assert float((again - last).abs().max()) == 0.0
    _close("eval_mix", mix[:, ::4], g["eval_mix"])
    _close("eval_last", last[:, ::4], g["eval_last"])
    _close("eval logits (mix)", logits["b"], g["eval_logits_b_mix"], abs_rel=4e-2, l2=2e-2)
    _, m2 = _module("last_hidden_state")                                         # the confs' feature_selection
    m2.eval()
    with torch.no_grad():
        l2, _ = m2(wavs, 16000, "b")
    _close("eval logits (last_hidden_state)", l2["b"], g["eval_logits_b_last"], abs_rel=4e-2, l2=2e-2)


def _train_step(m, wavs, texts, wp, tp):
    from lid.ConformerLangModel import CtcLossFn
    m.zero_grad()
    logits, _ = m(wavs, 16000, "b")
    z = logits["b"]
    in_len, tg_len = (z.shape[1] * wp).long(), (texts.shape[-1] * tp).long()
    per = CtcLossFn.apply(z, texts, in_len.to(DEV), tg_len.to(DEV), 40, m.lidk_engine.k)
    per.mean().backward()
    torch.cuda.synchronize()
    return z, per, in_len, tg_len


@pytest.mark.parametrize("tag", ["", "full::"])
def test_xlsr_training_step_against_the_reference_classes(tag):
    """'' : transformer un-frozen, conv extractor frozen (the regime after ``freeze_tranformer_epoch``); 'full::' : the extractor
    un-frozen too (after ``freeze_encoder_epoch``): every conv weight / bias and extractor LayerNorm takes a gradient."""
    from test_gpu_wavlm import _cmp_grads
    g = load_npz("xlsr_step.npz")
    mod, m = _module()
    m.train()
    m.unfreeze_tranformer_encoder()
    if tag:
        m.unfreeze_feature_extractor()
    else:
        m.freeze_feature_extractor()
    wavs, texts, wp, tp = rc.wavlm_batch()
    wavs, texts = [w.to(DEV) for w in wavs], texts.to(DEV)
    z, per, in_len, tg_len = _train_step(m, wavs, texts, wp, tp)
    assert in_len.tolist() == g["in_len"].tolist() and tg_len.tolist() == g["tg_len"].tolist()
    ref = torch.from_numpy(g[tag + "train_logits_b"])
    lerr, loss, ref_loss = float((z.detach().cpu() - ref).abs().max()), float(per.mean().detach()), float(g[tag + "train_loss"])
    print(f"[xlsr train step {tag}] logits err {lerr:.3e} (max |ref| {float(ref.abs().max()):.2f}); loss {loss:.4f} vs {ref_loss:.4f}")
    assert lerr <= 4e-2 * max(1.0, float(ref.abs().max())) and abs(loss - ref_loss) <= 2e-2 * ref_loss
    params = dict(m.named_parameters())
    view = {"grad_names": g[tag + "grad_names"], "grad_norms": g[tag + "grad_norms"]}
    view.update({"gs::" + str(k): g[tag + "gs::" + str(k)] for k in g[tag + "grad_names"]})
    n = _cmp_grads(view, lambda name: params[name].grad, "xlsr train step " + tag, skip=("model.featurizer.weights",))
    assert n >= (94 if tag else 64)
    # The three mixing logits' gradient is sm * (dots - sum sm dots), dots[l] = <d feat, h_l>: a sum that cancels to ~1e-3 of
    # |d feat| |h_l| (the fixture stores that uncancelled scale: 1.0e3 .. 3.8e3 against gradients of 0.1 .. 2.8, so bf16 roundings
    # of d feat worth 2e-4 of the scale - measured: 0.28 absolute - are 6 % of the result).  Judged against the scale.
    got = params["model.featurizer.weights"].grad.cpu().double()
    ref, scale = torch.from_numpy(g[tag + "gs::model.featurizer.weights"]).double(), float(g[tag + "mix_scale"].max())
    err = float((got - ref).abs().max())
    print(f"[xlsr train step {tag}] mixing-weight gradient {got.tolist()} vs {ref.tolist()}: err {err:.3e} = {err / scale:.2e} of the "
          f"uncancelled scale {scale:.0f}")
    assert err <= 1e-3 * scale and float((got @ ref) / (got.norm() * ref.norm())) >= 0.99
    if not tag:
        assert all(params[k].grad is None for k in params if ".feature_extractor." in k)


def test_wavlm_large_against_the_reference():
    """WavLM Large form: layer_norm extractor without conv bias, pre-LN layers whose relative-bias gate reads LN1's output, no
    waveform normalisation (the reference's WavLM wrapper ignores cfg.normalize), no padding mask inside the encoder."""
    from lid.WavLMMutiLangModel import WavLMMutiLangModel
    from test_gpu_wavlm import _cmp_grads
    g = load_npz("wavlm_large_step.npz")
    m = WavLMMutiLangModel(wavlm_cfg=dict(wc.WAVLM_LARGE_CFG), lang2vocab=wc.L2V, lang2index=wc.L2I, conformer_linear=True,
                           dropout=0.0, linear_dim=1024, hidden_dim=wc.HEAD_LARGE["hidden_dim"], dim_head=wc.HEAD_LARGE["dim_head"],
                           num_head=wc.HEAD_LARGE["num_head"], mask=False)
    sd = {"model.featurizer.model." + k: v for k, v in wc.backbone_weights_cfg(wc.WAVLM_LARGE_CFG).items()}
    sd.update(wc.head_weights(encoder_dim=1024, hidden_dim=wc.HEAD_LARGE["hidden_dim"]))
    m.load_state_dict(sd)
    m.to(DEV)
    assert m.backbone.pre_ln and m.backbone.ln_extractor and not m.backbone.normalize and m.backbone.rel_pos
    wavs, texts, wp, tp = rc.wavlm_batch()
    wavs, texts = [w.to(DEV) for w in wavs], texts.to(DEV)
    m.eval()
    with torch.no_grad():
        logits, _ = m(wavs, 16000, "b")
        wav = torch.nn.utils.rnn.pad_sequence(wavs, batch_first=True).contiguous()
        last = m.backbone.forward(wav, n_samples=[int(w.shape[0]) for w in wavs]).clone()
    _close("wavlm-large eval_last", last[:, ::4], g["eval_last"])
    _close("wavlm-large eval logits", logits["b"], g["eval_logits_b"], abs_rel=4e-2, l2=2e-2)
    m.train()
    m.unfreeze_tranformer_encoder()
    m.unfreeze_feature_extractor()
    z, per, in_len, tg_len = _train_step(m, wavs, texts, wp, tp)
    ref = torch.from_numpy(g["train_logits_b"])
    lerr, loss, ref_loss = float((z.detach().cpu() - ref).abs().max()), float(per.mean().detach()), float(g["train_loss"])
    print(f"[wavlm-large train step] logits err {lerr:.3e} (max |ref| {float(ref.abs().max()):.2f}); loss {loss:.4f} vs {ref_loss:.4f}")
    assert lerr <= 4e-2 * max(1.0, float(ref.abs().max())) and abs(loss - ref_loss) <= 2e-2 * ref_loss
    params = dict(m.named_parameters())
    n = _cmp_grads(g, lambda name: params[name].grad, "wavlm-large train step")
    assert n >= 95
