"""Host-logic tests of the drop-in boundary on CPU (BASELINE config 1: 2-layer d64, 3 languages, 1 s utterances, batch 4,
ccml Trainer): launcher wiring, Trainer.fit/test, callbacks, checkpoint keys + resume, and world_size-2 gloo data parallelism.
The kernels are replaced by tests/fake_ops.py (torch CPU); the product itself has no CPU path (see test_abi.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import fake_ops
from conftest import PKG, ROOT


@pytest.fixture
def fake_backend(monkeypatch):
    import lidk.engine
    import lid.audio_processor
    monkeypatch.setattr(lidk.engine, "ops", fake_ops)
    monkeypatch.setattr(lid.audio_processor, "_ops", fake_ops)
    return fake_ops


def load_cfg(*overrides):
    from lid import hydra_lite
    base = ["trainer.gpu_id=null", "trainer.use_amp=false", "trainer.total_epoch=1", "trainer.log_interval=2",
            "data.synthetic.items_per_lang=8", "data.synthetic.val_items_per_lang=4", "data.synthetic.seconds=0.5",
            "module.interval=2"]
    return hydra_lite.load_config(os.path.join(PKG, "lid", "conf"), "synthetic_cfg1", base + list(overrides))


def test_fit_cfg1_end_to_end_checkpoint_and_resume(fake_backend, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    import lid.main as launcher
    from ccml import seed_everything
    from ccml.callbacks.ckpt_callback import CkptCallback
    from ccml.callbacks.lr_callback import LrCallback
    from ccml.loggers.jsonl_logger import JsonlLogger
    from ccml.trainer import Trainer
    seed_everything(0)
    cfg = load_cfg()
    module, sets, params = launcher.build(cfg)
    trainer = Trainer(callbacks=[CkptCallback(file_name_metric=["epoch", "val_loss"], save_topk=2), LrCallback()],
                      loggers=[JsonlLogger("metrics.jsonl")], **dict(cfg["trainer"]))
    w0 = module.model.state_dict()["model.featurizer.encoders.0.ff1.fn.fn.net.0.weight"].clone()
    trainer.fit(module, train_dataset=sets["train"], val_dataset=sets["val"], test_dataset=sets["test"], dataloader_params=params)
    assert trainer.current_step == len(trainer.train_dataloader) == 6
    sd = module.model.state_dict()
    assert not torch.equal(sd["model.featurizer.encoders.0.ff1.fn.fn.net.0.weight"], w0)          # it trained
    assert torch.isfinite(torch.cat([v.reshape(-1).float() for v in sd.values()])).all()
    assert 0.0 <= module.last_val["cavg"] <= 1.0 and np.isfinite(module.last_val["val_loss"])
    state = torch.load("ckpt/last.pt", weights_only=False)
    assert {"model", "hyper_parameters", "epoch", "optimizer", "scalar", "logger", "lr_scheduler"} <= set(state)
    assert sorted(state["model"]) == sorted(sd)
    assert any(f.startswith("epoch_0_val_loss_") for f in os.listdir("ckpt"))
    # resume: a new trainer restores model + optimizer + scheduler and continues at epoch 0's end
    cfg2 = load_cfg("trainer.total_epoch=2", "trainer.checkpoint_path=ckpt/last.pt")
    module2, sets2, params2 = launcher.build(cfg2)
    t2 = Trainer(callbacks=[], loggers=[], **dict(cfg2["trainer"]))
    t2.fit(module2, train_dataset=sets2["train"], val_dataset=sets2["val"], test_dataset=sets2["test"], dataloader_params=params2)
    assert t2.current_epoch == 1 and t2.lr_scheduler.last_epoch >= 6
    # inference-time restore from hyper-parameters (CCMLModule.resume_from_checkpoint)
    from lid.LidModule_ASR_Supervised import LidSuperviseModule
    m3 = LidSuperviseModule.resume_from_checkpoint("ckpt/last.pt", "cpu")
    assert torch.equal(m3.model.state_dict()["model.last_projects.c.linear.bias"], state["model"]["model.last_projects.c.linear.bias"])


def test_ragged_collate_matches_reference_layout(fake_backend):
    """Host side of a variable-length batch: the collate carries the true lengths, the feature call is per utterance with
    zero-padded mel rows (the reference's collate, lid/raw_datasets.py:345-365) and wav_percents are frame fractions."""
    import lid.main as launcher
    from lid.audio_processor import num_frames
    from oracle import features as of
    cfg = load_cfg("+data.synthetic.min_seconds=0.2")
    module, sets, params = launcher.build(cfg)
    ds = sets["val"]
    items = [ds[i] for i in range(4)]
    lens = [int(it[0].shape[-1]) for it in items]
    assert len(set(lens)) > 1
    wb, texts, wav_pct, text_pct, paths, langs = ds.collate_fn(items)
    assert wb.n_samples.tolist() == lens and wb.wav.shape == (4, max(lens))
    pad = wb.pad
    frames = [num_frames(n, pad) for n in lens]
    np.testing.assert_allclose(wav_pct.numpy(), np.array(frames, np.float32) / max(frames), rtol=1e-6)
    mel = wb.to_mel()
    ref, pct = of.collate_mel([of.wav2mel(of.normalize_wav(items[i][0].reshape(1, -1)), pad=pad)[0] for i in range(4)])
    assert mel.shape == ref.shape
    np.testing.assert_allclose(mel.numpy(), ref.numpy(), atol=1e-4)
    np.testing.assert_allclose(pct.numpy(), wav_pct.numpy(), rtol=1e-6)


def test_state_dict_keys_match_reference(cfg1_weights):
    from lid.ConformerLangModel import ConformerMutiLangModel
    m = ConformerMutiLangModel(lang2vocab={"a": 30, "b": 40, "c": 50}, lang2index={"a": 0, "b": 1, "c": 2}, hidden_dim=32,
                               conformer_linear=True, dropout=0.1, linear_dim=64, n_blocks=2, encoder_dim=64, dim_head=16,
                               last_dim_head=8, heads=4)
    sd = m.state_dict()
    assert sorted(sd) == sorted(cfg1_weights)
    assert all(tuple(sd[k].shape) == tuple(cfg1_weights[k].shape) for k in sd)
    m.load_state_dict(cfg1_weights)                                    # a reference checkpoint loads as is ...
    flat = m.lidk_engine.flat
    p = m.model.featurizer.encoders[1].attn.fn.to_kv.weight if hasattr(m.model.featurizer.encoders, "__getitem__") else None
    name = "model.featurizer.encoders.1.attn.fn.to_kv.weight"
    assert torch.equal(m.lidk_engine.pview(name), cfg1_weights[name])  # ... straight into the flat arena
    assert m.lidk_engine.pview(name).data_ptr() == dict(m.named_parameters())[name].data_ptr()


def _dp_worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    import fake_ops as fo
    import lidk.engine
    import lid.audio_processor
    lidk.engine.ops = fo
    lid.audio_processor._ops = fo
    import lid.main as launcher
    from ccml import seed_everything
    from ccml.trainer import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    seed_everything(0)
    ov = ["model.dropout=0.0", "data.feature.mask_times=0", f"data.sampler_common.train_batch_size={4 // world}",
          "trainer.total_epoch=1", f"trainer.ddp={'true' if world > 1 else 'false'}", f"trainer.world_size={world}",
          f"trainer.local_rank={rank}", "trainer.backend=gloo", f"trainer.master_port={port}"]
    cfg = load_cfg(*ov)
    module, sets, params = launcher.build(cfg, rank, world)
    module.model.lidk_engine.cfg.pos_dropout = 0.0
    module.model.use_stochastic_depth = False
    for ds in sets.values():
        ds.train = False                       # no dither: both layouts must see identical waveforms
    params["train_batch_sampler"].seed = 0     # same global batches whatever the world size
    trainer = Trainer(callbacks=[], loggers=[], **dict(cfg["trainer"]))
    trainer.fit(module, train_dataset=sets["train"], val_dataset=sets["val"], test_dataset=sets["test"], dataloader_params=params)
    torch.save({k: v.clone() for k, v in module.model.state_dict().items()}, os.path.join(out_dir, f"w{world}_r{rank}.pt"))


@pytest.mark.timeout(600)
def test_data_parallel_gloo_matches_single_process(tmp_path):
    """2 ranks x batch 2 (SyncBN sums + averaged gradients) == 1 rank x batch 4, and the ranks stay bit-identical."""
    ctx = mp.get_context("spawn")
    p1 = ctx.Process(target=_dp_worker, args=(0, 1, 29611, str(tmp_path)))
    p1.start(); p1.join(300)
    assert p1.exitcode == 0
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, 29613, str(tmp_path))) for r in range(2)]
    [p.start() for p in procs]
    [p.join(400) for p in procs]
    assert [p.exitcode for p in procs] == [0, 0]
    one = torch.load(tmp_path / "w1_r0.pt")
    r0, r1 = torch.load(tmp_path / "w2_r0.pt"), torch.load(tmp_path / "w2_r1.pt")
    for k in one:
        assert torch.equal(r0[k], r1[k]), f"ranks diverged on {k}"
        # a bias in front of BatchNorm has an exactly-zero gradient; Novograd divides by the tensor's own gradient norm, so
        # what is left (rounding noise, different in the two layouts) becomes a unit-norm update: not comparable
        if one[k].is_floating_point() and not k.endswith(("conv.net.4.conv.bias", "conv.net.5.running_mean")):   # (the running mean tracks that bias)
            np.testing.assert_allclose(r0[k].numpy(), one[k].numpy(), rtol=2e-3, atol=2e-5, err_msg=k)


def test_collate_with_speed_perturbation(fake_backend):
    """`speed_shift: true` (the reference's xf confs): one factor of {0.9, 1.0, 1.1} per training utterance, lengths, frame
    fractions and SpecAugment spans follow the PERTURBED lengths, the mel rows behind an utterance's own frames are zero."""
    import random
    import lid.main as launcher
    from lid.audio_processor import SPEED_FACTORS, num_frames, speed_out_len
    cfg = load_cfg("data.feature.speed_shift=true", "+data.synthetic.min_seconds=0.3")
    module, sets, params = launcher.build(cfg)
    ds = sets["train"]
    assert ds.feat.speed_shift and not sets["val"].train
    random.seed(5)
    items = [ds[i] for i in range(4)]
    wb, texts, wav_pct, text_pct, paths, langs = ds.collate_fn(items)
    assert wb.speed is not None and all(f in SPEED_FACTORS for f in wb.speed) and len(set(wb.speed)) > 1
    in_lens = [int(it[0].shape[-1]) for it in items]
    assert wb.n_samples.tolist() == in_lens
    out_lens = [speed_out_len(n, p, q) for n, (p, q) in zip(in_lens, wb.speed)]
    frames = [num_frames(n, wb.pad) for n in out_lens]
    np.testing.assert_allclose(wav_pct.numpy(), np.array(frames, np.float32) / max(frames), rtol=1e-6)
    assert wb.shape == (4, max(frames), 80)
    mel = wb.to_mel()
    assert mel.shape == (4, max(frames), 80)
    for b, f in enumerate(frames):
        assert float(mel[b, f:].abs().max()) == 0.0 if f < max(frames) else True
    wbv = sets["val"].collate_fn([sets["val"][i] for i in range(4)])[0]
    assert wbv.speed is None                                  # no augmentation outside training


def test_background_batch_handover_order_errors_and_early_exit():
    """ccml.trainer._threaded / _lookahead(background=True): same items in the same order, a producer exception reaches the
    consumer, and a consumer that stops early releases the helper thread (it must not stay blocked on a full queue)."""
    import threading
    import time
    from ccml.trainer import _lookahead, _threaded
    assert list(_threaded(range(7))) == list(range(7))
    assert list(_lookahead(range(4), background=True)) == [(0, 1), (1, 2), (2, 3), (3, None)]

    def failing():
        yield 1
        raise ValueError("collate failed")

    with pytest.raises(ValueError, match="collate failed"):
        list(_threaded(failing()))
    before = {t.name for t in threading.enumerate()}
    g = _threaded(iter(range(10 ** 6)))
    assert next(g) == 0 and next(g) == 1
    g.close()
    deadline = time.time() + 5
    while time.time() < deadline and any(t.name == "ccml-batch-prefetch" and t.is_alive() for t in threading.enumerate()):
        time.sleep(0.05)
    assert not any(t.name == "ccml-batch-prefetch" and t.is_alive() for t in threading.enumerate()), before
