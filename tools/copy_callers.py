#!/usr/bin/env python3
"""Which Python lines issue device-to-device Tensor.copy_ calls in a training step of a backbone workload?  (rocprofv3 shows them as
__amd_rocclr_copyBuffer: ~140 per WavLM fine-tune step, ~280 per wav2vec2 ragged step.)   python tools/copy_callers.py --model w2v2 --ragged"""
import argparse, collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="wavlm")
ap.add_argument("--wavlm-regime", default="finetune")
ap.add_argument("--ragged", action="store_true")
ap.add_argument("--steps", type=int, default=4)
a = ap.parse_args()
args = argparse.Namespace(model=a.model, batch=64, resident=1, val_items=2, blocks=12, lr=0.01, stochastic_depth=False, warmup=0,
                          steps=a.steps, cavg_steps=0, wavlm_regime=a.wavlm_regime, ragged=a.ragged, gpus=1)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg, module, trainer, sets = bench.build(args, 0, 1, dev)
batches = bench.resident_batches(sets["train"], 0, 1, dev, args.batch, 1, ragged=a.ragged)
for b in batches: b.pop()
for i in range(6):
    trainer.train_step(i, batches[i % len(batches)], 10 ** 9)
torch.cuda.synchronize()
counts = collections.Counter()
orig = torch.Tensor.copy_
def copy_(self, src, *k, **kw):
    if self.is_cuda and getattr(src, "is_cuda", False):
        fr = [f for f in traceback.extract_stack(limit=8)[:-1] if "speech-lid_amd" in f.filename or "bench.py" in f.filename]
        key = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "other"
        counts[(key, "same" if self.dtype == src.dtype else "cast")] += 1
    return orig(self, src, *k, **kw)
torch.Tensor.copy_ = copy_
for i in range(a.steps):
    trainer.train_step(6 + i, batches[(6 + i) % len(batches)], 10 ** 9)
torch.cuda.synchronize()
torch.Tensor.copy_ = orig
for k, v in counts.most_common(25):
    print(f"{v / a.steps:8.1f} per step  {k}")
