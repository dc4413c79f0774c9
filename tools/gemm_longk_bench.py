#!/usr/bin/env python3
"""Graph-timed gemm_nt at the cfg2 long-K data-gradient shapes (M = 9664, N = 256, K = 512 / 768 / 1024, bf16 out): the direct
64x64 kernel (LIDK_GEMM_PIPEK=0) against the K-generic pipelined kernel with 1 / 2 / 3 tiles per workgroup.  Four operand sets
alternate so that a launch does not find its inputs in a warm L2."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops

dev, NSET = "cuda:0", 4


def t(fn, n=16):
    for i in range(NSET): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i % NSET)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3


for M, N, K in ((9664, 256, 768), (9664, 256, 1024), (9664, 256, 512), (9664, 512, 768), (9536, 768, 768)):
    S = [dict(A=torch.randn(M, K, device=dev).bfloat16(), B=(torch.randn(N, K, device=dev) / 16).bfloat16(),
              o=torch.empty(M, N, device=dev, dtype=torch.bfloat16)) for _ in range(NSET)]
    row = []
    for pk in ("0", "1", "2", "3"):
        ops.gemm_option("LIDK_GEMM_PIPEK", int(pk))
        us = t(lambda i: ops.gemm_nt(S[i]["A"], S[i]["B"], S[i]["o"]))
        row.append(f"PIPEK={pk} {us:6.1f} us ({2 * M * N * K / us * 1e-6:4.0f} TF)")
    print(f"M={M} N={N} K={K}: " + "   ".join(row), flush=True)
