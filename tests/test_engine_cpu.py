"""Host-logic test (no GPU): the engine's forward / hand-written backward ORCHESTRATION, driven through the torch-CPU
fake backend (tests/fake_ops.py), must reproduce the reference's logits, CTC-loss gradients and BatchNorm buffers stored
in tests/golden/.  The same checks run against the real HIP kernels in tests/test_gpu_model.py."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fake_ops
from conftest import load_npz
from lidk.engine import Engine
from lidk.layout import ConformerCfg, init_values


def make_cfg(**kw):
    base = dict(lang2vocab={"a": 30, "b": 40, "c": 50}, lang2index={"a": 0, "b": 1, "c": 2}, n_blocks=2, encoder_dim=64,
                dim_head=16, heads=4, last_dim_head=8, dropout=0.1, hidden_dim=32)
    base.update(kw)
    return ConformerCfg(**base)


def make_engine(cfg, weights):
    eng = Engine(cfg, act_dtype=torch.float32, backend=fake_ops)
    eng.to("cpu")
    eng.load_state(weights)
    return eng


def ctc_dlogits(logits, g):
    lg = logits.clone().requires_grad_()
    out = lg
    loss = F.ctc_loss(torch.log_softmax(out, -1).transpose(0, 1), torch.from_numpy(g["texts"]),
                      (out.shape[1] * torch.from_numpy(g["wav_pct"])).long(),
                      (g["texts"].shape[-1] * torch.from_numpy(g["text_pct"])).long(), blank=40, reduction="none",
                      zero_infinity=True).mean()
    loss.backward()
    return float(loss), lg.grad


def test_init_matches_reference_construction_order(cfg1_weights):
    torch.manual_seed(0)
    vals = init_values(make_cfg())
    for n in ("model.featurizer.sub_sampling.sub_sampling.0.weight", "model.featurizer.encoders.1.ff2.fn.fn.net.3.weight",
              "model.featurizer.encoders.0.attn.fn.rel_pos_emb.weight", "model.last_projects.c.linear.weight",
              "model.last_projects.a.block.conv.net.4.conv.weight", "lang_discriminator.linear.2.bias"):
        assert torch.equal(vals[n], cfg1_weights[n]), n


def test_eval_forward_all_heads(cfg1_weights):
    g = load_npz("cfg1_eval.npz")
    eng = make_engine(make_cfg(), cfg1_weights)
    out = eng.forward(torch.from_numpy(g["mel"]), None, training=False)
    for lang in "abc":
        np.testing.assert_allclose(out[lang].numpy(), g[f"logits_{lang}"], atol=3e-5, rtol=0)
    one = eng.forward(torch.from_numpy(g["mel"]), "b", training=False)
    np.testing.assert_allclose(one["b"].numpy(), g["logits_b_only"], atol=3e-5, rtol=0)


def _compare_grads(eng, g, names=None):
    names = names or [k[6:] for k in g if k.startswith("grad::")]
    for k in names:
        ref = g["grad::" + k]
        got = eng.gview(k).numpy()
        np.testing.assert_allclose(got, ref, atol=3e-5 * max(1.0, float(np.abs(ref).max())), rtol=2e-4, err_msg=k)


def test_train_step_A_backward_chain(cfg1_weights):
    g = load_npz("cfg1_trainA.npz")
    mel = torch.from_numpy(load_npz("cfg1_eval.npz")["mel"])
    eng = make_engine(make_cfg(dropout=0.0, pos_dropout=0.0), cfg1_weights)
    eng.zero_grad()
    out = eng.forward(mel, "b", training=True, keep_layers=[True, True])
    np.testing.assert_allclose(out["b"].numpy(), g["logits_b"], atol=3e-5, rtol=0)
    loss, dl = ctc_dlogits(out["b"], g)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=1e-5)
    eng.backward(dl)
    _compare_grads(eng, g)
    # tensors without a gradient in the reference stay exactly zero here (Q5-Q7)
    ref_names = {k[6:] for k in g if k.startswith("grad::")}
    for s in eng.specs:
        if s.name not in ref_names:
            assert float(eng.gview(s.name).abs().max()) == 0.0, s.name
    active = {eng.specs[t].name for t in eng.active_tensor_ids("b", [True, True])}
    assert active == ref_names
    for k in g:
        if k.startswith("bn::"):
            np.testing.assert_allclose(eng.buffers[k[4:]].numpy(), g[k], atol=1e-5, rtol=1e-5, err_msg=k)


def test_train_step_B_masks_and_skipped_layer(cfg1_weights):
    g = load_npz("cfg1_trainB.npz")
    gA = load_npz("cfg1_trainA.npz")
    mel = torch.from_numpy(load_npz("cfg1_eval.npz")["mel"])
    eng = make_engine(make_cfg(), cfg1_weights)
    eng.zero_grad()
    masks = {"pos": torch.from_numpy(g["pos_mask"]).reshape(-1).to(torch.uint8),
             "head": torch.from_numpy(g["head_mask"]).reshape(-1).to(torch.uint8)}
    out = eng.forward(mel, "b", training=True, keep_layers=list(g["keep"]), masks=masks)
    np.testing.assert_allclose(out["b"].numpy(), g["logits_b"], atol=3e-5, rtol=0)
    loss, dl = ctc_dlogits(out["b"], gA)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=1e-5)
    eng.backward(dl)
    for n, ref in zip(g["grad_names"], g["grad_norms"]):
        # atol: a bias in front of BatchNorm has an exactly-zero gradient; both sides hold ~1e-6 of rounding noise there
        np.testing.assert_allclose(float(eng.gview(str(n)).norm()), ref, rtol=3e-4, atol=2e-5, err_msg=str(n))
    _compare_grads(eng, g)
    assert float(eng.grad[slice(*eng.stage_range("enc.1"))].abs().max()) == 0.0      # skipped layer: no gradient


def test_real_backend_refuses_cpu():
    import lidk
    eng = Engine(make_cfg())
    with pytest.raises(lidk.LidkError):
        eng.to("cpu") and eng.forward(torch.zeros(1, 101, 80), "a", training=False)


def test_workspace_pools_serve_many_ragged_shapes(cfg1_weights, monkeypatch):
    """Ragged corpora: shapes of one capacity class (frames rounded up to 64) share ONE pool of HBM and get exact-size views;
    classes are evicted least-recently-used; results do not depend on which shapes ran before."""
    monkeypatch.setenv("LIDK_MAX_WORKSPACES", "2")
    eng = make_engine(make_cfg(dropout=0.0, pos_dropout=0.0), cfg1_weights)
    g = torch.Generator().manual_seed(0)
    mels = {F_: 20.0 * torch.randn(2, F_, 80, generator=g) - 30.0 for F_ in (61, 45, 64, 101, 150, 33)}
    first = {}
    for F_, mel in mels.items():
        eng.zero_grad()
        out = eng.forward(mel, "a", training=True, keep_layers=[True, True])["a"]
        first[F_] = out.clone()
        eng.backward(0.01 * torch.ones_like(out))
        first[(F_, "g")] = eng.grad.clone()
        assert len(eng._pools) <= 2
    assert eng.work(2, 61).pool is eng.work(2, 45).pool is eng.work(2, 64).pool          # one class: frames <= 64
    assert eng.work(2, 61).pool is not eng.work(2, 101).pool
    w61, w45 = eng.work(2, 61), eng.work(2, 45)
    assert w61.col.data_ptr() == w45.col.data_ptr() and w61.x0.shape[0] == 2 * 31 and w45.x0.shape[0] == 2 * 23   # same HBM, exact views
    for F_ in (150, 61, 33, 101, 45):                       # again, in another order, across evictions
        eng.zero_grad()
        eng.load_state(cfg1_weights)                        # BatchNorm running statistics back to the start
        out = eng.forward(mels[F_], "a", training=True, keep_layers=[True, True])["a"]
        assert torch.equal(out, first[F_]), F_
        eng.backward(0.01 * torch.ones_like(out))
        assert torch.allclose(eng.grad, first[(F_, "g")], rtol=1e-5, atol=1e-7), F_
