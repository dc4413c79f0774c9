#!/usr/bin/env python3
"""What do the vendor GEMMs (hipBLASLt / rocBLAS through torch.matmul) reach on this model's shapes?  A yardstick for the
hand-written kernels (tools/gemm_bench.py), not part of the product path."""
import torch
dev = "cuda:0"
M = 9664


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3


for name, m, n, k in [("NT ff-up", M, 1024, 256), ("NT qkv", M, 768, 256), ("NT ff-down", M, 256, 1024), ("NT out", M, 256, 256),
                      ("NT pw1-dgrad", M, 256, 1024), ("NT K=512", M, 256, 512)]:
    A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16()
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.matmul(A, B.t(), out=out))
    print(f"{name:16s} M={m} N={n} K={k}: {us:7.1f} us  {2.0*m*n*k/us/1e6:7.1f} TF/s")
for name, n1, n2 in [("TN dW1", 1024, 256), ("TN dW2", 256, 1024), ("TN dWqkv", 768, 256), ("TN dWo", 256, 256)]:
    X = torch.randn(M, n1, device=dev).bfloat16(); Y = torch.randn(M, n2, device=dev).bfloat16()
    out = torch.empty(n1, n2, device=dev, dtype=torch.bfloat16)
    us = t(lambda: torch.matmul(X.t(), Y, out=out))
    print(f"{name:16s} [{n1},{n2}] over M={M}: {us:7.1f} us  {2.0*M*n1*n2/us/1e6:7.1f} TF/s")
