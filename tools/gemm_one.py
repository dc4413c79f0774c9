#!/usr/bin/env python3
"""One GEMM shape launched N times eagerly (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K [act] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops, _lib as L
m, n, k = (int(v) for v in sys.argv[1:4])
act = sys.argv[4] if len(sys.argv) > 4 else "swish"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = "cuda:0"
A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16()
bias = torch.randn(n, device=dev); out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
kw = {}
if act == "swish": kw = dict(act=L.ACT_SWISH, out2=torch.empty_like(out), bias=bias)
elif act == "plain": kw = {}
elif act == "res":
    out = torch.empty(m, n, device=dev); kw = dict(res=torch.randn(m, n, device=dev), bias=bias)
for _ in range(reps): ops.gemm_nt(A, B, out, **kw)
torch.cuda.synchronize()
