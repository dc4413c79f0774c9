#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Run once, here:  python oracle/gen_golden.py      (needs /root/reference; never runs on the GPU box)

What it does: puts /root/reference on sys.path, imports the reference's own
``lid.ConformerLangModel``, ``ccml.optim.*``, ``lid.cavg`` and ``lid.tokenizer``, feeds them seeded
inputs and stores inputs + outputs as raw arrays.  Only arrays are written — no reference source,
bytecode or pickled classes.

Three third-party imports of the reference are absent from this image and are never executed on the
16 kHz Conformer path: ``torchaudio`` (module-level import in lid/conformer.py:7 used
only by the never-instantiated FBank, and ``transforms.Resample`` objects that DataProcessor bypasses
at 16 kHz, lid/ConformerLangModel.py:156-157), ``torchmetrics`` (metric objects constructed but not
called by forward) and ``torch.utils.tensorboard``.  They are registered as EMPTY placeholder modules
so the import statements succeed; no arithmetic goes through them.
"""
import os
import random
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def _placeholders():
    class _Unused(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    ta = mod("torchaudio")
    ta.transforms = mod("torchaudio.transforms", Resample=_Unused)
    mod("torchmetrics", WER=_Unused, CharErrorRate=_Unused, Accuracy=_Unused)
    mod("torch.utils.tensorboard", SummaryWriter=object)


CFG1 = dict(lang2vocab={"a": 30, "b": 40, "c": 50}, lang2index={"a": 0, "b": 1, "c": 2}, hidden_dim=32,
            conformer_linear=True, dropout=0.1, linear_dim=64, n_blocks=2, n_mels=80, encoder_dim=64,
            dim_head=16, last_dim_head=8, heads=4)


def np_sd(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def main():
    os.makedirs(OUT, exist_ok=True)
    _placeholders()
    sys.path.insert(0, REF)
    from lid.ConformerLangModel import ConformerMutiLangModel  # noqa: E402

    # ------------------------------------------------------------------ model: eval forward (cfg1)
    torch.manual_seed(0)
    model = ConformerMutiLangModel(**CFG1)
    # make BN running stats and LN/BN affine params non-trivial so eval-mode BN is actually exercised
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for name, buf in model.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(0.2 * torch.randn(buf.shape, generator=g))
            if name.endswith("running_var"):
                buf.copy_(0.5 + torch.rand(buf.shape, generator=g))
        for name, p in model.named_parameters():
            if ".norm.weight" in name or "post_norm.weight" in name or "net.0.weight" in name and p.dim() == 1 \
                    or "net.5.weight" in name:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            if ".norm.bias" in name or "post_norm.bias" in name or "net.5.bias" in name:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    np.savez(os.path.join(OUT, "cfg1_weights.npz"), **np_sd(model))

    mel = (20.0 * torch.randn(4, 101, 80, generator=g) - 30.0)
    model.eval()
    with torch.no_grad():
        logits, (lid_asr, lid_linear) = model(mel, 16000, None)
        one, none_pair = model(mel, 16000, "b")
    assert none_pair == (None, None)
    np.savez(os.path.join(OUT, "cfg1_eval.npz"), mel=mel.numpy(), lid_asr=lid_asr.numpy(),
             lid_linear=lid_linear.numpy(), logits_b_only=one["b"].numpy(),
             **{f"logits_{k}": v.numpy() for k, v in logits.items()})

    # ------------------------------------------------------------------ model: train step A (deterministic)
    texts = torch.randint(0, 40, (4, 12), generator=g)
    text_len = torch.tensor([12, 9, 12, 6])
    for i in range(4):
        texts[i, text_len[i]:] = 0
    text_pct = (text_len / 12.0).float()
    wav_pct = torch.tensor([1.0, 0.9, 1.0, 0.8])

    def run_train(pos_p, head_p, stochastic, seed):
        model.load_state_dict({k: torch.from_numpy(v) for k, v in np.load(os.path.join(OUT, "cfg1_weights.npz")).items()})
        model.train()
        model.zero_grad(set_to_none=True)
        feat = model.model.featurizer
        feat.use_stochastic_depth = stochastic
        feat.pos.dropout.p = pos_p
        model.model.last_projects["b"].dr.p = head_p
        masks = {}

        def pos_hook(mod, inp, out):
            if "pos" not in masks and inp[0].shape[0] == 4:
                masks["pos"] = (out != 0) | (inp[0] == 0)

        def head_hook(mod, inp, out):
            masks["head"] = (out != 0) | (inp[0] == 0)

        kept = []
        hooks = [feat.pos.dropout.register_forward_hook(pos_hook),
                 model.model.last_projects["b"].dr.register_forward_hook(head_hook)]
        for i, blk in enumerate(feat.encoders):
            hooks.append(blk.register_forward_hook(lambda m, a, b, i=i: kept.append(i)))
        random.seed(seed)
        torch.manual_seed(seed)
        out, _ = model(mel, 16000, "b")
        out = out["b"]
        loss = model.model.loss_fns["b"](torch.log_softmax(out, dim=-1).transpose(1, 0), texts,
                                         (out.shape[1] * wav_pct).long().cpu(),
                                         (texts.shape[-1] * text_pct).long().cpu())
        per_utt = loss.detach().clone()
        loss = torch.mean(loss)
        loss.backward()
        for h in hooks:
            h.remove()
        grads = {k: p.grad.numpy() for k, p in model.named_parameters() if p.grad is not None}
        keep = [i in kept for i in range(len(feat.encoders))]
        return out.detach().numpy(), per_utt.numpy(), float(loss), grads, masks, keep

    out, per_utt, loss, grads, masks, keep = run_train(0.0, 0.0, False, 3)
    bn = {k: v for k, v in np_sd(model).items() if "running_" in k or "num_batches" in k}
    np.savez(os.path.join(OUT, "cfg1_trainA.npz"), texts=texts.numpy(), text_pct=text_pct.numpy(),
             wav_pct=wav_pct.numpy(), logits_b=out, loss_per_utt=per_utt, loss=np.float32(loss),
             keep=np.array(keep), **{"grad::" + k: v for k, v in grads.items()},
             **{"bn::" + k: v for k, v in bn.items()})

    # ------------------------------------------------------------------ model: train step B (dropout + stochastic depth)
    for seed in range(1, 200):      # find a seed where exactly one encoder layer is dropped
        random.seed(seed)
        draws = [random.random() for _ in range(2)]
        ps = [1 - ((i + 1) / 2) * (1 - 0.7) for i in range(2)]
        if draws[0] <= ps[0] and not draws[1] <= ps[1]:
            break
    out, per_utt, loss, grads, masks, keep = run_train(0.1, 0.1, True, seed)
    assert keep == [True, False], keep
    np.savez(os.path.join(OUT, "cfg1_trainB.npz"), seed=np.int64(seed), keep=np.array(keep),
             pos_mask=masks["pos"].numpy(), head_mask=masks["head"].numpy(), logits_b=out,
             loss_per_utt=per_utt, loss=np.float32(loss),
             grad_names=np.array(sorted(grads)), grad_norms=np.array([np.linalg.norm(grads[k]) for k in sorted(grads)]),
             **{"grad::" + k: grads[k] for k in sorted(grads) if "encoders.0.attn" in k or "sub_sampling" in k
                or "last_projects.b.linear" in k})

    # ------------------------------------------------------------------ Novograd + TriStage traces
    from ccml.optim.novograd import Novograd  # noqa: E402
    from ccml.optim.tri_state import TriStageLRSchedule  # noqa: E402
    gp = torch.Generator().manual_seed(11)
    shapes = [(7, 5), (13,), (3, 4, 2)]
    p0 = [torch.randn(s, generator=gp) for s in shapes]
    gsteps = [[torch.randn(s, generator=gp) * (0.1 + k) for s in shapes] for k in range(4)]
    params = [nn.Parameter(p.clone()) for p in p0]
    opt = Novograd(params, lr=0.01, weight_decay=1e-5)
    sched = TriStageLRSchedule(optimizer=opt, phase_ratio=[0.1, 0.4, 0.5], init_lr_scale=0.05,
                               final_lr_scale=0.02, max_update=50, lr=0.01)
    trace = {}
    lrs = [opt.param_groups[0]["lr"]]
    for k in range(4):
        for j, p in enumerate(params):
            p.grad = None if (k == 1 and j == 2) else gsteps[k][j].clone()   # step 1: tensor 2 has no grad
        opt.step()
        sched.step()
        lrs.append(opt.param_groups[0]["lr"])
        for j, p in enumerate(params):
            trace[f"p{j}_after{k}"] = p.detach().numpy().copy()
    for _ in range(60):
        sched.step()
        lrs.append(opt.param_groups[0]["lr"])
    np.savez(os.path.join(OUT, "optim_trace.npz"), lrs=np.array(lrs, dtype=np.float64),
             **{f"p{j}_init": p.numpy() for j, p in enumerate(p0)},
             **{f"g{k}_{j}": gsteps[k][j].numpy() for k in range(4) for j in range(3)}, **trace)

    # ------------------------------------------------------------------ Cavg + greedy CTC collapse KATs
    import lid.cavg as ref_cavg  # noqa: E402
    from lid.tokenizer import CTCTokenizer  # noqa: E402
    rng = np.random.RandomState(5)
    kat = {}
    for case, (n_utt, n_lang) in enumerate([(40, 3), (90, 14), (12, 2)]):
        scores = rng.rand(n_utt, n_lang)
        tgt = rng.randint(0, n_lang, n_utt)
        scores[np.arange(n_utt), tgt] += 0.35 * rng.rand(n_utt)
        scores /= scores.sum(1, keepdims=True)
        pairs = [(j, int(tgt[i]), float(scores[i, j])) for i in range(n_utt) for j in range(n_lang)]
        lo, hi = min(p[2] for p in pairs), max(p[2] for p in pairs)
        _, mn = ref_cavg.get_cavg(pairs, n_lang, lo, hi, 20, 0.5)
        kat[f"cavg{case}_scores"], kat[f"cavg{case}_tgt"], kat[f"cavg{case}_value"] = scores, tgt, np.float64(round(mn, 4))
    tok = CTCTokenizer([chr(ord("a") + i) for i in range(6)])
    seqs = rng.randint(0, 7, (8, 30))
    lens = rng.randint(5, 31, 8)
    dec = tok.ctc_decode(torch.from_numpy(seqs), torch.from_numpy(lens))
    kat["ctc_seqs"], kat["ctc_lens"], kat["ctc_decoded"] = seqs, lens, np.array(dec)
    np.savez(os.path.join(OUT, "metrics_kat.npz"), **kat)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
