#!/usr/bin/env python3
"""Where does the HOST spend a training step?  cProfile over N steps of the bench's step function (resident batches).
    python tools/host_profile.py --model wavlm --steps 10
"""
import argparse
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="conformer")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--wavlm-regime", default="frozen")
    ap.add_argument("--ragged", action="store_true")
    a = ap.parse_args()
    args = argparse.Namespace(model=a.model, batch=a.batch, resident=1, val_items=2, blocks=12, lr=0.01, stochastic_depth=False,
                              warmup=0, steps=a.steps, cavg_steps=0, wavlm_regime=a.wavlm_regime, ragged=a.ragged, gpus=1)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    cfg, module, trainer, sets = bench.build(args, 0, 1, dev)
    batches = bench.resident_batches(sets["train"], 0, 1, dev, args.batch, 1, ragged=a.ragged)
    for b in batches: b.pop()
    nb = len(batches)
    for i in range(3 * nb):
        trainer.train_step(i, batches[i % nb], 10 ** 9)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(a.steps):
        trainer.train_step(i, batches[i % nb], 10 ** 9)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(35)


if __name__ == "__main__":
    main()
