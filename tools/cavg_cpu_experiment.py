#!/usr/bin/env python3
"""CPU experiment (oracle only): is the 'tones' synthetic corpus learnable, and does the CTC-confidence LID score separate
languages?  Trains BASELINE config 1 with the torch-CPU oracle for N steps and prints the validation Cavg.
    python tools/cavg_cpu_experiment.py --steps 300
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch  # noqa: E402

from lid.raw_datasets import SyntheticMergedDataset  # noqa: E402
from oracle import conformer as oc, features as of, metrics as om, optim as oo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--text-len", type=int, default=8)
    ap.add_argument("--items", type=int, default=64)
    ap.add_argument("--val-items", type=int, default=16)
    ap.add_argument("--eval-every", type=int, default=100)
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    l2v, l2i = {"a": 30, "b": 40, "c": 50}, {"a": 0, "b": 1, "c": 2}
    cfg = oc.ModelCfg(lang2vocab=l2v, lang2index=l2i, n_blocks=2, encoder_dim=64, dim_head=16, heads=4, last_dim_head=8,
                      dropout=0.1, hidden_dim=32)
    from lidk.layout import ConformerCfg, init_values
    sd = init_values(ConformerCfg(lang2vocab=l2v, lang2index=l2i, n_blocks=2, encoder_dim=64, dim_head=16, heads=4,
                                  last_dim_head=8, hidden_dim=32))
    for p in [k[:-len(".weight")] for k in sd if k.endswith("conv.net.5.weight")]:
        sd[p + ".running_mean"], sd[p + ".running_var"] = torch.zeros_like(sd[p + ".weight"]), torch.ones_like(sd[p + ".weight"])
    names = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    states = {k: oo.NovogradState() for k in names}
    kw = dict(seconds=a.seconds, text_len=a.text_len, transcript="tones", type="mel", pad=16)
    tr = SyntheticMergedDataset(True, l2i, l2v, items_per_lang=a.items, seed=1234, **kw)
    va = SyntheticMergedDataset(False, l2i, l2v, items_per_lang=a.val_items, seed=1235, **kw)
    sched = oo.TriStage(lr=a.lr, max_update=a.steps)

    def feats(ds, idx, train):
        wav = of.normalize_wav(torch.stack([ds[i][0] for i in idx]))
        if train:
            wav = of.dither_preemphasis(wav, torch.rand_like(wav))
        mel = of.wav2mel(wav, pad=16)
        if train:
            mel = torch.stack([of.apply_specaug(m, of.draw_specaug_spans(m.shape[-1], 80, 0.05, 12, 1)) for m in mel])
        return mel.transpose(1, 2).contiguous()

    def evaluate():
        full = dict(sd)
        pairs = []
        for k, lang in enumerate(l2v):
            idx = [k * a.val_items + j for j in range(a.val_items)]
            with torch.no_grad():
                _, (s, _) = oc.forward(feats(va, idx, False), full, cfg, None, oc.RunOpts(training=False))
            probs = [om.score_to_prob(r) for r in s.tolist()]
            pairs += om.make_pairs(probs, [k] * len(idx))
            print(f"   lang {lang}: mean scores {s.mean(0).tolist()}")
        return om.cavg(pairs, 3)

    t0 = time.time()
    for step in range(a.steps):
        k = step % 3
        lang = list(l2v)[k]
        g = torch.Generator().manual_seed(step)
        idx = (k * a.items + torch.randperm(a.items, generator=g)[:a.batch]).tolist()
        x = feats(tr, idx, True)
        p = {n: sd[n].requires_grad_(True) for n in names}
        opts = oc.RunOpts(training=True, keep_layers=[True, True])
        out, _ = oc.forward(x, {**sd, **p}, cfg, lang, opts)
        texts = torch.stack([tr[i][1] for i in idx])
        loss = oc.ctc_loss(out[lang], texts, torch.ones(len(idx)), torch.ones(len(idx)), blank=l2v[lang])
        loss.backward()
        with torch.no_grad():
            act = [n for n in names if sd[n].grad is not None]
            oo.clip_grad_norm([sd[n].grad for n in act], 20.0)
            oo.novograd_step([sd[n] for n in act], [sd[n].grad for n in act], [states[n] for n in act], lr=sched.lr_at(step),
                             weight_decay=1e-5)
            for n in names:
                sd[n].grad = None
                sd[n].requires_grad_(False)
            sd.update(opts.bn_buffers)
        if step % 20 == 0:
            print(f"step {step} lang {lang} loss {float(loss):.3f}  ({time.time() - t0:.0f}s)", flush=True)
        if (step + 1) % a.eval_every == 0:
            print(f"== step {step + 1}: val cavg {evaluate()}", flush=True)


if __name__ == "__main__":
    main()
