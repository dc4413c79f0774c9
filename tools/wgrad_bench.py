#!/usr/bin/env python3
"""Graph-timed micro-benchmark of the grouped weight-gradient launch of one ConformerBlock at the cfg2 shape (M = 9664 rows, d = 256,
ff = 1024, conv inner 512): register-staged 128-tiles (LIDK_TN_DMA=0) against the LDS-DMA ring (LIDK_TN_DMA = 2 / 3 / 4 stages),
for several row splits.  Two buffer sets alternate so that a launch does not find its operands in a warm L2 / MALL.  Every
variant's result is checked against the first one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops

dev = "cuda:0"
M = int(os.environ.get("M", 9664))
SITES = [(1024, 256), (256, 1024), (1024, 256), (256, 1024), (768, 256), (256, 256), (1024, 256), (256, 512)]      # (n, k): dW [n, k]
NSET = 2
torch.manual_seed(0)
S = []
for _ in range(NSET):
    S.append([dict(dy=(torch.randn(M, n, device=dev) / 8).bfloat16(), x=torch.randn(M, k, device=dev).bfloat16(),
                   dW=torch.zeros(n, k, device=dev), db=torch.zeros(n, device=dev)) for n, k in SITES])
flop = sum(2 * M * n * k for n, k in SITES)
byts = sum(M * (n + k) * 2 for n, k in SITES)


def group(i, split, tile=128):
    return ops.build_tn_group([(s["dy"], s["x"], s["dW"], s["db"], M, s["dy"].shape[1], s["x"].shape[1]) for s in S[i]], split=split, tile=tile)


def run(split, n=8, tile=128):
    gs = [group(i, split, tile) for i in range(NSET)]
    for s in S[0]: s["dW"].zero_(); s["db"].zero_()
    ops.gemm_tn_grouped(gs[0]); torch.cuda.synchronize()
    res = [(s["dW"].clone(), s["db"].clone()) for s in S[0]]
    for g in gs: ops.gemm_tn_grouped(g)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(n): ops.gemm_tn_grouped(gs[i % NSET])
    gr.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): gr.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3, res, gs[0][2]


ref = None
for dma in os.environ.get("VARIANTS", "0,2,3,4").split(","):
    for split in (int(v) for v in os.environ.get("SPLITS", "2,3,4").split(",")):
        os.environ["LIDK_TN_DMA"] = dma
        us, res, items = run(split)
        if ref is None: ref = res
        err = max(float((a - c).abs().max() / c.abs().max()) for (a, _), (c, _) in zip(res, ref))
        errb = max(float((a - c).abs().max() / c.abs().max()) for (_, a), (_, c) in zip(res, ref))
        print(f"LIDK_TN_DMA={dma} split={split} items={items:4d}  {us:7.1f} us  {flop / us * 1e-6:6.0f} TFLOP/s  {byts / us * 1e-6:5.2f} TB/s"
              f"   max rel diff dW {err:.1e} db {errb:.1e}", flush=True)

# 256x256 tiles: tiles per site 4 4 4 4 3 1 4 2 = 26; "bal" gives every site the split that fills 256 CUs with one item each
for split in os.environ.get("SPLITS256", "4,6,8,9,10,bal").split(","):
    sp = [10, 10, 10, 10, 10, 9, 10, 9] if split == "bal" else int(split)
    us, res, items = run(sp, tile=256)
    err = max(float((a - c).abs().max() / c.abs().max()) for (a, _), (c, _) in zip(res, ref))
    errb = max(float((a - c).abs().max() / c.abs().max()) for (_, a), (_, c) in zip(res, ref))
    print(f"256-tiles split={split} items={items:4d}  {us:7.1f} us  {flop / us * 1e-6:6.0f} TFLOP/s  {byts / us * 1e-6:5.2f} TB/s"
          f"   max rel diff dW {err:.1e} db {errb:.1e}", flush=True)

# ablation builds only (csrc built with -DLIDK_TN_ABLATION): LIDK_TN_ABL bits 1 no DMA in the loop, 2 no reads / MFMAs, 4 no stores,
# 8 reads without MFMAs, 16 MFMAs without reads (results are wrong by construction)
for abl in [a for a in os.environ.get("ABLS", "").split(",") if a]:
    os.environ["LIDK_TN_DMA"], os.environ["LIDK_TN_ABL"] = "3", abl
    us, _, items = run(2)
    print(f"ablation {abl:>2s} (ring of 3, split 2): {us:7.1f} us", flush=True)
os.environ.pop("LIDK_TN_ABL", None)
