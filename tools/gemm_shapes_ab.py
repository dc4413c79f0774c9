#!/usr/bin/env python3
"""Backbone GEMM shapes under the default dispatch vs the persistent 256x256 kernel forced (LIDK_GEMM_DMA256 = 1): where should the
fill rule sit?  Graph-timed, rotating buffers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops, _lib as L
dev = "cuda:0"


def timed(fn, n=8):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i % 3)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (4 * n) * 1e3


def case(m, n, k, mode):
    A = [torch.randn(m, k, device=dev).bfloat16() for _ in range(3)]
    B = [(torch.randn(n, k, device=dev) / k ** 0.5).bfloat16() for _ in range(3)]
    bias = torch.randn(n, device=dev)
    f32 = mode == "res"
    out = [torch.empty(m, n, device=dev, dtype=torch.float32 if f32 else torch.bfloat16) for _ in range(3)]
    kw = {}
    if mode == "bias": kw = dict(bias=bias)
    elif mode == "gelu": kw = dict(bias=bias, act=L.ACT_GELU, out2=torch.empty(m, n, device=dev, dtype=torch.bfloat16))
    elif mode == "res": kw = dict(bias=bias, res=torch.randn(m, n, device=dev))
    elif mode == "gelu_grad": kw = dict(act=L.ACT_GELU_GRAD, aux=torch.randn(m, n, device=dev).bfloat16())
    res = []
    for v, bn in ((0, 0), (-1, -1), (1, 256), (1, 128)):       # persistent kernel off / default dispatch / forced 256 x 256 / forced 256 x 128
        ops.gemm_option("LIDK_GEMM_DMA256", v)
        ops.gemm_option("LIDK_GEMM_DMA256_BN", bn)
        res.append(timed(lambda i: ops.gemm_nt(A[i], B[i], out[i], **kw)))
    ops.gemm_option("LIDK_GEMM_DMA256", -1)
    ops.gemm_option("LIDK_GEMM_DMA256_BN", -1)
    t256 = -(-m // 256) * (n // 256)
    print(f"M={m:6d} N={n:5d} K={k:5d} {mode:9s} tiles256={t256:5d} ({t256 / 256:5.2f} rounds): off {res[0]:7.1f}  default {res[1]:7.1f}  256x256 {res[2]:7.1f}  "
          f"256x128 {res[3]:7.1f} us   {2.0 * m * n * k / min(res) / 1e6:6.0f} TFLOP/s best")


for m in (4800, 9536, 16000, 24000):
    for (n, k, mode) in ((2304, 768, "bias"), (768, 768, "res"), (3072, 768, "gelu"), (768, 3072, "res"), (768, 3072, "gelu_grad"),
                         (3072, 768, "plain"), (768, 2304, "plain")):
        case(m, n, k, mode)
