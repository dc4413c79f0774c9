#!/usr/bin/env python3
"""How many block sequences of a training step replay a captured hipGraph and how many still run eagerly?  Counts per key family over
the last N steps of a short resident-batch run (the bench's own batches).   python tools/graph_stats.py [--ragged] [--steps 28]"""
import argparse, collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="conformer")
ap.add_argument("--steps", type=int, default=28)
ap.add_argument("--setup", type=int, default=56)
ap.add_argument("--ragged", action="store_true")
a = ap.parse_args()
args = argparse.Namespace(model=a.model, batch=64, resident=1, val_items=2, blocks=12, lr=0.01, stochastic_depth=False, warmup=0,
                          steps=a.steps, cavg_steps=0, wavlm_regime="frozen", ragged=a.ragged, gpus=1)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfg, module, trainer, sets = bench.build(args, 0, 1, dev)
batches = bench.resident_batches(sets["train"], 0, 1, dev, args.batch, 1, ragged=a.ragged)
for b in batches: b.pop()
nb = len(batches)
gc = trainer.engine.graphs
print("eager_uses", gc.eager_uses, "batches", nb, "frames per batch", [int(b[0].shape[-1]) if hasattr(b[0], "shape") else len(b[0]) for b in batches])
for i in range(a.setup):
    trainer.train_step(i, batches[i % nb], 10 ** 9)
torch.cuda.synchronize()
stats = collections.Counter()
orig = gc.run
def run(key, fn):
    st = gc.state.get(key)
    kind = "replay" if (st is not None and st[0] is not None) else ("capture" if (st is not None and st[1] >= gc.eager_uses) else "eager")
    stats[(key[0], kind)] += 1
    t0 = time.perf_counter(); r = orig(key, fn); stats[(key[0], kind, "host_ms")] += (time.perf_counter() - t0) * 1e3
    return r
gc.run = run
t0 = time.perf_counter()
for i in range(a.steps):
    trainer.train_step(a.setup + i, batches[(a.setup + i) % nb], 10 ** 9)
host = (time.perf_counter() - t0) / a.steps * 1e3
torch.cuda.synchronize()
print(f"host {host:.2f} ms/step (issue only), wall {(time.perf_counter() - t0) / a.steps * 1e3:.2f} ms/step; graph keys alive {len(gc.state)}; enabled {gc.enabled}")
for k in sorted(stats, key=str): print(k, round(stats[k], 2))
print("workspaces", len(getattr(trainer.engine, "_works", getattr(trainer.engine, "works", {}))))
