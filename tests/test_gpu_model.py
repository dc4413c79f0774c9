"""End-to-end parity on the GPU: HIP engine vs vectors captured from the reference (tests/golden/) and vs the oracle.

f32 mode is the algorithm check (tolerance 2e-4 on logits of magnitude ~1-3, gradients to 1e-3 relative); bf16 mode is the
production mode: logits within 6e-2 absolute (bf16 has 8 significant bits and the path is ~40 GEMMs deep), loss within
2 %, per-tensor gradient direction cosine >= 0.99.
"""
import numpy as np
import pytest
import torch

from conftest import load_npz
from lidk import ops
from lidk.engine import Engine
from lidk.layout import ConformerCfg
from oracle import conformer as oc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_cfg(**kw):
    base = dict(lang2vocab={"a": 30, "b": 40, "c": 50}, lang2index={"a": 0, "b": 1, "c": 2}, n_blocks=2, encoder_dim=64,
                dim_head=16, heads=4, last_dim_head=8, dropout=0.1, hidden_dim=32)
    base.update(kw)
    return ConformerCfg(**base)


def make_engine(cfg, weights, dt):
    eng = Engine(cfg, act_dtype=dt)
    eng.to(DEV)
    eng.load_state({k: v.to(DEV) for k, v in weights.items()})
    return eng


def gpu_ctc(out, g, blank=40):
    B, T, V1 = out.shape
    texts = torch.from_numpy(g["texts"]).to(DEV)
    in_len = (T * torch.from_numpy(g["wav_pct"]).to(DEV)).long()
    tg_len = (texts.shape[-1] * torch.from_numpy(g["text_pct"]).to(DEV)).long()
    loss = torch.empty(B, device=DEV)
    dl = torch.empty(B, T, V1, device=DEV)
    ws = torch.empty(ops.ctc_workspace_bytes(B, T, V1, texts.shape[1]) // 4 + 1, device=DEV)
    ops.ctc_loss(out.contiguous(), texts, in_len, tg_len, loss, dl, ws, blank, grad_scale=1.0 / B)
    return float(loss.mean()), dl


# f32 logits: BASELINE.md section 2's own figure, "fp32-mode tolerance 1e-4 abs" (the achieved value is printed by each test)
TOL = {torch.float32: dict(logit=1e-4, loss=1e-4, grel=2e-3, cos=0.99999), torch.bfloat16: dict(logit=6e-2, loss=2e-2, grel=None, cos=0.99)}


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_eval_forward_and_lid_scores(cfg1_weights, cfg1_cfg, dt):
    g = load_npz("cfg1_eval.npz")
    eng = make_engine(make_cfg(), cfg1_weights, dt)
    mel = torch.from_numpy(g["mel"]).to(DEV)
    out = eng.forward(mel, None, training=False)
    worst = 0.0
    for lang in "abc":
        err = float((out[lang].cpu() - torch.from_numpy(g[f"logits_{lang}"])).abs().max())
        worst = max(worst, err)
    print(f"[eval logits {dt}] max_abs_err={worst:.3e}")
    assert worst <= TOL[dt]["logit"]
    scores = torch.zeros(4, 3, device=DEV)
    for lang, v in cfg1_cfg.lang2vocab.items():
        ops.lid_score(out[lang].contiguous(), scores[:, cfg1_cfg.lang2index[lang]:], 3, v)
    ref = torch.from_numpy(g["lid_asr"])
    print(f"[lid_asr {dt}] max_abs_err={float((scores.cpu() - ref).abs().max()):.3e} margin={float((ref.sort(-1).values[:, -1] - ref.sort(-1).values[:, -2]).min()):.3e}")
    assert float((scores.cpu() - ref).abs().max()) <= (1e-5 if dt == torch.float32 else 5e-3)
    if dt == torch.float32:
        assert torch.equal(scores.argmax(-1).cpu(), ref.argmax(-1))          # argmax language labels exact
    # LangDiscriminator MLP (lid_linear, lid/ConformerLangModel.py:394) on the HIP scores against the reference's output
    lin = torch.empty_like(scores)
    pv = eng.pview
    ops.lid_mlp(scores, pv("lang_discriminator.linear.0.weight"), pv("lang_discriminator.linear.0.bias"),
                pv("lang_discriminator.linear.2.weight"), pv("lang_discriminator.linear.2.bias"), lin)
    lerr = float((lin.cpu() - torch.from_numpy(g["lid_linear"])).abs().max())
    print(f"[lid_linear {dt}] max_abs_err={lerr:.3e}")
    assert lerr <= (1e-5 if dt == torch.float32 else 5e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_eval_rows_are_independent_of_the_batch(cfg1_weights, dt):
    """Validation scores equal-length utterances together instead of one by one (LidSuperviseModule.val_loop): in eval mode the
    logits of an utterance must not depend on what else is in the batch, bit for bit."""
    g = load_npz("cfg1_eval.npz")
    eng = make_engine(make_cfg(), cfg1_weights, dt)
    mel = torch.from_numpy(g["mel"]).to(DEV)
    full = {k: v.clone() for k, v in eng.forward(mel, None, training=False).items()}
    for i in range(mel.shape[0]):
        one = eng.forward(mel[i:i + 1].contiguous(), None, training=False)
        for lang in "abc":
            assert torch.equal(one[lang][0], full[lang][i]), (lang, i)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_train_step_A_loss_and_all_gradients(cfg1_weights, dt):
    g = load_npz("cfg1_trainA.npz")
    mel = torch.from_numpy(load_npz("cfg1_eval.npz")["mel"]).to(DEV)
    eng = make_engine(make_cfg(dropout=0.0, pos_dropout=0.0), cfg1_weights, dt)
    eng.zero_grad()
    out = eng.forward(mel, "b", training=True, keep_layers=[True, True])
    err = float((out["b"].cpu() - torch.from_numpy(g["logits_b"])).abs().max())
    print(f"[trainA logits {dt}] max_abs_err={err:.3e}")
    assert err <= TOL[dt]["logit"]
    loss, dl = gpu_ctc(out["b"], g)
    print(f"[trainA loss {dt}] got={loss:.6f} ref={float(g['loss']):.6f}")
    assert abs(loss - float(g["loss"])) <= TOL[dt]["loss"] * abs(float(g["loss"]))
    eng.backward(dl)
    torch.cuda.synchronize()
    worst_cos, worst_rel, bad = 1.0, 0.0, []
    for k in g:
        if not k.startswith("grad::"):
            continue
        name, ref = k[6:], torch.from_numpy(g[k]).reshape(-1).double()
        got = eng.gview(name).cpu().reshape(-1).double()
        nr = float(ref.norm())
        if nr < 1e-4:                                  # exact-zero gradients in exact arithmetic (bias before BatchNorm):
            print(f"[zero-grad {name} {dt}] |got|={float(got.norm()):.3e}")      # what is left is rounding noise of dc
            assert float(got.norm()) < (1e-3 if dt == torch.float32 else 0.25), name
            continue
        cos = float((got @ ref) / (got.norm() * ref.norm() + 1e-30))
        rel = float((got - ref).norm() / nr)
        worst_cos, worst_rel = min(worst_cos, cos), max(worst_rel, rel)
        if cos < TOL[dt]["cos"] or (TOL[dt]["grel"] and rel > TOL[dt]["grel"]):
            bad.append((name, cos, rel))
    print(f"[trainA grads {dt}] worst_cos={worst_cos:.6f} worst_rel_l2={worst_rel:.3e}")
    assert not bad, bad[:8]
    ref_names = {k[6:] for k in g if k.startswith("grad::")}
    for s in eng.specs:
        if s.name not in ref_names:
            assert float(eng.gview(s.name).abs().max()) == 0.0, s.name
    if dt == torch.float32:
        for k in g:
            if k.startswith("bn::"):
                np.testing.assert_allclose(eng.buffers[k[4:]].cpu().numpy(), g[k], atol=2e-5, rtol=2e-5, err_msg=k)


def test_train_step_B_masks_and_skipped_layer_f32(cfg1_weights):
    g = load_npz("cfg1_trainB.npz")
    gA = load_npz("cfg1_trainA.npz")
    mel = torch.from_numpy(load_npz("cfg1_eval.npz")["mel"]).to(DEV)
    eng = make_engine(make_cfg(), cfg1_weights, torch.float32)
    eng.zero_grad()
    masks = {"pos": torch.from_numpy(g["pos_mask"]).reshape(-1).to(torch.uint8).to(DEV),
             "head": torch.from_numpy(g["head_mask"]).reshape(-1).to(torch.uint8).to(DEV)}
    out = eng.forward(mel, "b", training=True, keep_layers=list(g["keep"]), masks=masks)
    assert float((out["b"].cpu() - torch.from_numpy(g["logits_b"])).abs().max()) <= 2e-4
    loss, dl = gpu_ctc(out["b"], gA)
    assert abs(loss - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    eng.backward(dl)
    for n, ref in zip(g["grad_names"], g["grad_norms"]):
        np.testing.assert_allclose(float(eng.gview(str(n)).norm()), ref, rtol=2e-3, atol=2e-5, err_msg=str(n))
    assert float(eng.grad[slice(*eng.stage_range("enc.1"))].abs().max()) == 0.0


def test_own_dropout_masks_are_consistent_between_fwd_and_bwd():
    """With the kernel's own counter-based masks: export them, feed them to the oracle, compare logits and one gradient."""
    torch.manual_seed(3)
    cfg = make_cfg()
    eng = Engine(cfg, act_dtype=torch.float32)
    eng.to(DEV)
    sd = {k: v.cpu().clone() for k, v in eng.state().items()}
    mel = (20 * torch.randn(3, 61, 80) - 30)
    eng.zero_grad()
    out = eng.forward(mel.to(DEV), "a", training=True, keep_layers=[True, True])
    w = eng.work(3, 61)
    pos = w.pos_keep.cpu().view(3, w.T, 64).bool()
    head = w.head_keep.cpu().view(3, w.T, 64).bool()
    assert 0.85 < pos.float().mean() < 0.95 and 0.85 < head.float().mean() < 0.95
    ocfg = oc.ModelCfg(lang2vocab=cfg.lang2vocab, lang2index=cfg.lang2index, n_blocks=2, encoder_dim=64, dim_head=16,
                       heads=4, last_dim_head=8, dropout=0.1)
    sdr = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
    ref, _ = oc.forward(mel, sdr, ocfg, "a", oc.RunOpts(training=True, keep_layers=[True, True], pos_keep_mask=pos,
                                                        head_keep_mask=head))
    assert float((out["a"].cpu() - ref["a"].detach()).abs().max()) <= 3e-4
    dl = torch.randn(3, w.T, 31) * 0.01
    ref["a"].backward(dl)
    eng.backward(dl.to(DEV))
    n = "model.featurizer.sub_sampling.linear.weight"
    got, want = eng.gview(n).cpu(), sdr[n].grad
    assert float((got - want).norm() / want.norm()) < 2e-3


@pytest.mark.parametrize("whole", ["1", "0"])
def test_graph_replay_matches_eager(cfg1_weights, whole, monkeypatch):
    """Sequences replayed from captured hipGraphs must give what the eager launch path gives (third call = replay): with the
    whole encoder forward / whole backward chain as one graph each (single-process default) and with one graph per block."""
    monkeypatch.setenv("LIDK_WHOLE_GRAPHS", whole)
    g = load_npz("cfg1_trainA.npz")
    mel = torch.from_numpy(load_npz("cfg1_eval.npz")["mel"]).to(DEV)
    results = []
    for graphs in (False, True):
        eng = make_engine(make_cfg(dropout=0.0, pos_dropout=0.0), cfg1_weights, torch.bfloat16)
        eng.graphs.enabled = graphs
        for _ in range(3):                      # eager, capture(+replay), replay
            eng.load_state({k: v.to(DEV) for k, v in cfg1_weights.items()})      # same BN buffers every round
            eng.zero_grad()
            out = eng.forward(mel, "b", training=True, keep_layers=[True, True])
            logits = out["b"].clone()
            loss, dl = gpu_ctc(out["b"], g)
            eng.backward(dl)
        torch.cuda.synchronize()
        if graphs:
            captured = sum(1 for st in eng.graphs.state.values() if st[0] is not None)
            assert captured == 3 if whole == "1" else captured >= 6          # encoder forward, head forward, backward chain
        results.append((logits.cpu(), eng.grad.cpu().clone(), {k: v.cpu().clone() for k, v in eng.buffers.items()}))
    (l0, g0, b0), (l1, g1, b1) = results
    assert torch.equal(l0, l1)
    rel = float((g0 - g1).norm() / g0.norm())
    print(f"[graph vs eager] grad rel diff {rel:.3e}")
    assert rel < 1e-5                            # split-K float atomics make the sum order run-dependent
    for k in b0:
        assert torch.allclose(b0[k].float(), b1[k].float(), atol=1e-6), k


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_baseline_cfg2_depth_against_the_oracle(dt):
    """BASELINE config 2 architecture at full depth and width (12 blocks, d = 256, 4 x 64 heads, ff x4, conv expansion 2,
    kernel 31, head block 8 x 32; 3 of the 14 languages and 1 s utterances to keep the CPU oracle quick): one training
    forward/backward on the HIP engine against the torch-CPU oracle with the same random weights.  f32 mode checks the
    algorithm end to end through 13 blocks; bf16 mode is the production setting (tolerances as in the module docstring)."""
    torch.manual_seed(5)
    l2v, l2i = {"a": 30, "b": 40, "c": 50}, {"a": 0, "b": 1, "c": 2}
    dims = dict(n_blocks=12, encoder_dim=256, dim_head=64, heads=4, last_dim_head=32)
    cfg = ConformerCfg(lang2vocab=l2v, lang2index=l2i, dropout=0.0, pos_dropout=0.0, hidden_dim=32, **dims)
    eng = Engine(cfg, act_dtype=dt)
    eng.to(DEV)
    weights = {k: v.detach().cpu().clone() for k, v in eng.state().items()}
    B, F_ = 4, 101
    mel = 20.0 * torch.randn(B, F_, 80) - 30.0
    texts = torch.randint(0, 40, (B, 8))
    eng.zero_grad()
    out = eng.forward(mel.to(DEV), "b", training=True, keep_layers=[True] * 12)["b"]
    ocfg = oc.ModelCfg(lang2vocab=l2v, lang2index=l2i, dropout=0.0, pos_dropout=0.0, **dims)
    names = [k for k, v in weights.items() if v.is_floating_point() and "running_" not in k]
    ref = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in weights.items()}
    logits, _ = oc.forward(mel, ref, ocfg, "b", oc.RunOpts(training=True, keep_layers=[True] * 12))
    err = float((out.cpu() - logits["b"].detach()).abs().max())
    scale = float(logits["b"].detach().abs().max())
    print(f"[cfg2-depth logits {dt}] max_abs_err={err:.3e} ref_max={scale:.3e}")
    assert err <= (1e-4 if dt == torch.float32 else 0.1) * max(1.0, scale)
    loss_ref = oc.ctc_loss(logits["b"], texts, torch.ones(B), torch.ones(B), blank=40)
    loss_ref.backward()
    g = {"texts": texts.numpy(), "wav_pct": np.ones(B, np.float32), "text_pct": np.ones(B, np.float32)}
    loss, dl = gpu_ctc(out, g)
    print(f"[cfg2-depth loss {dt}] got={loss:.5f} ref={float(loss_ref):.5f}")
    assert abs(loss - float(loss_ref)) <= (1e-5 if dt == torch.float32 else 5e-3) * abs(float(loss_ref))
    eng.backward(dl)
    torch.cuda.synchronize()
    worst_cos, worst_rel = 1.0, 0.0
    for k in names:
        gr = ref[k].grad
        if gr is None or float(gr.norm()) < 1e-5:
            continue
        got, want = eng.gview(k).cpu().reshape(-1).double(), gr.reshape(-1).double()
        cos = float((got @ want) / (got.norm() * want.norm() + 1e-30))
        rel = float((got - want).norm() / want.norm())
        worst_cos, worst_rel = min(worst_cos, cos), max(worst_rel, rel)
    print(f"[cfg2-depth grads {dt}] worst_cos={worst_cos:.6f} worst_rel_l2={worst_rel:.3e}")
    assert worst_cos >= (0.999999 if dt == torch.float32 else 0.99)
    assert worst_rel <= (5e-4 if dt == torch.float32 else 0.15)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fused_ctc_node_equals_the_two_step_route(dt, monkeypatch):
    """``forward_ctc`` (engine forward + lengths + CTC + mean as one autograd node, the CTC gradient written at backward time
    straight into the vocabulary projection's operand with the upstream scalar folded in) against the two-step route the module
    used before (forward, ``CtcLossFn``, ``.mean()``, torch's autograd in between): ragged lengths, an upstream gradient != 1
    (accumulate_grad = 2), same loss, same lengths, same gradients."""
    from lid.LidModule_ASR_Supervised import LidSuperviseModule
    from lid.tokenizer import CTCTokenizer
    l2v, l2i = {"a": 30, "b": 40, "c": 50}, {"a": 0, "b": 1, "c": 2}
    toks = {k: CTCTokenizer([chr(0x4E00 + i) for i in range(v)]) for k, v in l2v.items()}
    g = torch.Generator().manual_seed(11)
    mel = torch.randn(6, 101, 80, generator=g).to(DEV)
    texts = torch.randint(0, 40, (6, 9), generator=g).to(DEV)
    wp = torch.tensor([1.0, 0.52, 0.8, 0.33, 1.0, 0.9], device=DEV)
    tp = torch.tensor([1.0, 0.45, 0.7, 0.25, 0.9, 1.0], device=DEV)
    langs = torch.full((6,), 1, device=DEV)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("LIDK_CTC_FUSED", fused)
        torch.manual_seed(0)
        mod = LidSuperviseModule(optimizer_name="novograd", optimizer_param={"lr": 0.01}, scheduler="none", lang2index_dict=l2i,
                                 tokenizer_dict=toks, lang2vocab=l2v, dropout=0.0, linear_dim=64, n_blocks=2, encoder_dim=64,
                                 dim_head=16, last_dim_head=8, heads=4)
        m = mod.model
        m.set_compute_dtype(dt)
        m.lidk_engine.cfg.pos_dropout = 0.0
        m.use_stochastic_depth = False
        m.to(DEV).train()
        assert m.lidk_engine.ctc_supported() == (fused == "1")
        m.zero_grad()
        out = mod.common_loop([mel, texts, wp, tp, ["x"] * 6, langs], with_text=True)
        (out["loss"] / 2).backward()
        torch.cuda.synchronize()
        res[fused] = (float(out["loss"]), out["predict_texts"], {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    (l1, t1, g1), (l0, t0, g0) = res["1"], res["0"]
    print(f"[fused ctc {dt}] loss {l1:.6f} vs {l0:.6f}; {len(g1)} gradient tensors")
    assert abs(l1 - l0) <= 2e-6 * abs(l0) and t1 == t0 and g1.keys() == g0.keys() and len(g1) > 40
    tol = 2e-5 if dt == torch.float32 else 2e-2
    for k in g1:
        scale = float(g0[k].abs().max()) + 1e-12
        assert float((g1[k] - g0[k]).abs().max()) <= tol * scale, (k, float((g1[k] - g0[k]).abs().max()), scale)
