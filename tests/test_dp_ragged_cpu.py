"""Data parallelism with ragged per-rank shapes: world size 4 over gloo, every rank holds a
different (B, F) batch, so the SyncBatchNorm row counts differ per rank.  The engine's hand-written DP path (f64 sums + row
count in ONE all-reduce per conv module and direction, per-stage averaged gradients) must equal an independent restatement:
the CPU oracle under torch autograd with ``oracle.conformer.SyncBatchNorm`` (torch.nn.SyncBatchNorm semantics, which the
reference's Trainer installs under DDP, ccml/trainer.py:428) and gradients averaged over the ranks as DDP does.
Kernels are the torch-CPU fake backend; the real kernels run the same wiring in tests/test_gpu_dp.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT

SHAPES = [(2, 61), (1, 45), (3, 53), (2, 61)]          # (utterances, frames) per rank: M = B*T differs across ranks


def _worker(rank, world, port, out_dir, compress):
    sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import fake_ops
    from ccml.trainer import Trainer
    from conftest import load_npz
    from lidk.engine import Engine
    from lidk.layout import ConformerCfg
    from oracle import conformer as oc
    torch.set_num_threads(2)
    trainer = Trainer(ddp=True, backend="gloo", world_size=world, local_rank=rank, master_port=str(port), gpu_id=None,
                      grad_compress=compress)
    l2v, l2i = {"a": 30, "b": 40, "c": 50}, {"a": 0, "b": 1, "c": 2}
    dims = dict(n_blocks=2, encoder_dim=64, dim_head=16, heads=4, last_dim_head=8)
    cfg = ConformerCfg(lang2vocab=l2v, lang2index=l2i, dropout=0.0, pos_dropout=0.0, hidden_dim=32, **dims)
    eng = Engine(cfg, act_dtype=torch.float32, backend=fake_ops)
    eng.to("cpu")
    weights = {k: torch.from_numpy(v) for k, v in load_npz("cfg1_weights.npz").items()}
    if rank == 0:
        eng.load_state(weights)                      # the other ranks start from their own init: the attach broadcast must fix that
    trainer._attach_native_dp(eng)
    assert trainer.grad_compress == compress
    for k, v in eng.state().items():
        assert torch.equal(v, weights[k].to(v.dtype)), k
    B, F_ = SHAPES[rank]
    g = torch.Generator().manual_seed(100 + rank)
    mel = 20.0 * torch.randn(B, F_, 80, generator=g) - 30.0
    T = (F_ - 1) // 2 + 1
    dl = 0.01 * torch.randn(B, T, 41, generator=g)
    eng.zero_grad()
    out = eng.forward(mel, "b", training=True, keep_layers=[True, True])["b"].clone()
    eng.backward(dl)

    # ---- independent restatement: autograd oracle + SyncBatchNorm + DDP's gradient mean
    ocfg = oc.ModelCfg(lang2vocab=l2v, lang2index=l2i, dropout=0.0, pos_dropout=0.0, **dims)
    names = [k for k, v in weights.items() if v.is_floating_point() and "running_" not in k]
    sd = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in weights.items()}
    opts = oc.RunOpts(training=True, keep_layers=[True, True], all_reduce=lambda t: dist.all_reduce(t))
    ref, _ = oc.forward(mel, sd, ocfg, "b", opts)
    ref["b"].backward(dl)
    err = float((out - ref["b"].detach()).abs().max())
    assert err <= 2e-4, f"rank {rank}: logits differ by {err}"
    worst = 0.0
    for k in names:
        gr = sd[k].grad
        if gr is None:
            assert float(eng.gview(k).abs().max()) == 0.0, k
            continue
        gr = gr.clone()
        dist.all_reduce(gr)
        gr /= world
        got = eng.gview(k)
        scale = float(gr.norm())
        if scale < 1e-7:
            continue
        rel = float((got - gr).norm()) / scale
        worst = max(worst, rel)
        assert rel <= (2e-3 if compress == "none" else 1.5e-2), (k, rel)
    for k, v in opts.bn_buffers.items():             # running statistics: global mean / unbiased global variance
        np.testing.assert_allclose(eng.buffers[k].numpy(), v.numpy(), rtol=2e-4, atol=2e-5, err_msg=k)
    torch.save({"grad": eng.grad.clone(), "worst": worst}, os.path.join(out_dir, f"r{rank}_{compress}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("compress", ["none", "bf16"])
def test_world4_ragged_shapes_match_syncbn_oracle(tmp_path, compress):
    world = len(SHAPES)
    ctx = mp.get_context("spawn")
    port = 29641 if compress == "none" else 29647
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), compress)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(500) for p in procs]
    assert [p.exitcode for p in procs] == [0] * world
    res = [torch.load(tmp_path / f"r{r}_{compress}.pt") for r in range(world)]
    for r in range(1, world):
        assert torch.equal(res[0]["grad"], res[r]["grad"]), f"rank {r}'s averaged gradients differ from rank 0's"
    print(f"[dp world 4 ragged, payload {compress}] worst gradient rel err vs SyncBN oracle: {max(x['worst'] for x in res):.2e}")
