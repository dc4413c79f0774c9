#!/usr/bin/env python3
"""Graph-timed lidk_gemm_nt on one shape: gemm_time.py M N K [plain|bias|gelu|res] -> us per launch, TFLOP/s (env knobs apply)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops, _lib as L
m, n, k = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "plain"
dev = "cuda:0"
NS = 3
A = [torch.randn(m, k, device=dev).bfloat16() for _ in range(NS)]
B = [(torch.randn(n, k, device=dev) / k ** 0.5).bfloat16() for _ in range(NS)]
bias = torch.randn(n, device=dev)
out = [torch.empty(m, n, device=dev, dtype=torch.float32 if mode == "res" else torch.bfloat16) for _ in range(NS)]
kw = {}
if mode == "bias": kw = dict(bias=bias)
elif mode == "gelu": kw = dict(bias=bias, act=L.ACT_GELU, out2=torch.empty(m, n, device=dev, dtype=torch.bfloat16))
elif mode == "res": kw = dict(bias=bias, res=torch.randn(m, n, device=dev))
def fn(i): ops.gemm_nt(A[i], B[i], out[i], **kw)
for i in range(NS): fn(i)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(12): fn(i % NS)
g.replay(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): g.replay()
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 60 * 1e3
print(f"M={m} N={n} K={k} {mode}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s  (LIDK_GEMM_DBG={os.environ.get('LIDK_GEMM_DBG', '0')})")
