#!/usr/bin/env python3
"""Micro-benchmark of lidk_gemm_nt on the shapes of the cfg2 training step (M = 64*151 = 9664), HIP-event timed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops, _lib as L

dev = "cuda:0"
M = 9664
def t(fn, n=20):
    """GPU time per call: n launches captured in one hipGraph and replayed (no host launch gaps)."""
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3   # us

def run(name, m, n, k, **kw):
    A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16()
    bias = torch.randn(n, device=dev)
    out_f32 = kw.pop("out_f32", False)
    out = torch.empty(m, n, device=dev, dtype=torch.float32 if out_f32 else torch.bfloat16)
    extra = {}
    if kw.get("act") == L.ACT_SWISH: extra["out2"] = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    if kw.get("act") == L.ACT_SWISH_GRAD: extra["aux"] = torch.randn(m, n, device=dev).bfloat16()
    if kw.pop("res", False): extra["res"] = torch.randn(m, n, device=dev)
    if kw.get("splitk", 1) > 1: out.zero_(); bias = None
    us = t(lambda: ops.gemm_nt(A, B, out, bias=bias if kw.get("splitk", 1) == 1 else None, **kw, **extra))
    fl = 2.0 * m * n * k
    byt = (m * k + n * k) * 2 + m * n * (4 if out_f32 else 2) * (2 if "out2" in extra else 1) + (m * n * 2 if "aux" in extra else 0) + (m * n * 4 if "res" in extra else 0)
    print(f"{name:34s} M={m:5d} N={n:5d} K={k:5d}  {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {byt/us/1e3:7.1f} GB/s(alg)")

x = torch.empty(M * 1024, device=dev, dtype=torch.bfloat16); y = torch.empty_like(x)
us = t(lambda: y.copy_(x)); print(f"copy 19.8MB bf16: {us:.1f} us -> {2*x.numel()*2/us/1e3:.0f} GB/s")
z = torch.empty(2 * M * 1024, device=dev, dtype=torch.bfloat16)
us = t(lambda: z.fill_(1.0)); print(f"fill 39.6MB bf16: {us:.1f} us -> {z.numel()*2/us/1e3:.0f} GB/s")
us = t(lambda: y.fill_(1.0)); print(f"fill 19.8MB bf16: {us:.1f} us -> {y.numel()*2/us/1e3:.0f} GB/s")
run("ff up: bias+swish+pre", M, 1024, 256, act=L.ACT_SWISH)
run("pw1: bias (N=512)", M, 512, 256)
run("qkv: plain (N=768)", M, 768, 256)
run("ff down: bias,0.5,res f32", M, 256, 1024, alpha=0.5, res=True, out_f32=True)
run("attn out / pw2 (K=256)", M, 256, 256, res=True, out_f32=True)
run("dgrad da: swish_grad", M, 1024, 256, act=L.ACT_SWISH_GRAD)
run("dgrad dh (N=256,K=1024)", M, 256, 1024)
run("dgrad dh (N=256,K=768)", M, 256, 768)
run("dgrad dh (N=256,K=512)", M, 256, 512)
run("dgrad (N=256,K=256)", M, 256, 256)
def run_panel(name, m, n, ln, **kw):
    """row-panel kernel (K = 256): LN fused (reads the f32 residual stream) or bf16 operand"""
    k = 256
    x = torch.randn(m, k, device=dev); A = x.bfloat16(); B = torch.randn(n, k, device=dev).bfloat16()
    gamma, beta = torch.ones(k, device=dev), torch.zeros(k, device=dev)
    h = torch.empty(m, k, device=dev, dtype=torch.bfloat16); mean = torch.empty(m, device=dev); rstd = torch.empty(m, device=dev)
    bias = torch.randn(n, device=dev) if kw.pop("bias", True) else None
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    extra = {}
    if kw.get("act") == L.ACT_SWISH: extra["out2"] = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    if kw.get("act") == L.ACT_SWISH_GRAD: extra["aux"] = torch.randn(m, n, device=dev).bfloat16(); bias = None
    if ln:
        us = t(lambda: ops.ln_gemm_nt(x, gamma, beta, B, out, h=h, mean=mean, rstd=rstd, bias=bias, **kw, **extra))
        us0 = t(lambda: (ops.layernorm_fwd(x, gamma, beta, yT=h, mean=mean, rstd=rstd), ops.gemm_nt(h, B, out, bias=bias, **kw, **extra)))
    else:
        us = t(lambda: ops.ln_gemm_nt(None, None, None, B, out, A=A, bias=bias, **kw, **extra))
        us0 = t(lambda: ops.gemm_nt(A, B, out, bias=bias, **kw, **extra))
    print(f"{name:34s} M={m:5d} N={n:5d} K={k:5d}  panel {us:8.1f} us   current path {us0:8.1f} us")

if os.environ.get("PANEL"):
    run_panel("LN + ff up (bias+swish+pre)", M, 1024, True, act=L.ACT_SWISH)
    run_panel("LN + qkv (plain)", M, 768, True, bias=False)
    run_panel("LN + pw1 (bias)", M, 1024, True)
    run_panel("ff up, bf16 A", M, 1024, False, act=L.ACT_SWISH)
    run_panel("dgrad da (swish_grad), bf16 A", M, 1024, False, act=L.ACT_SWISH_GRAD)
    run_panel("dgrad N=256 K=256 plain", M, 256, False, bias=False)
    sys.exit(0)
if os.environ.get("BLAS_REF"):      # what the vendor library (hipBLASLt via torch) reaches on the same shapes, plain bf16 output
    for (m, n, k) in ((M, 1024, 256), (M, 768, 256), (M, 512, 256), (M, 256, 256), (M, 256, 1024), (1024, 256, M), (256, 1024, M), (256, 256, M)):
        A = torch.randn(m, k, device=dev).bfloat16(); B = torch.randn(n, k, device=dev).bfloat16(); out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        us = t(lambda: torch.mm(A, B.t(), out=out))
        print(f"torch.mm (hipBLASLt) NT             M={m:5d} N={n:5d} K={k:5d}  {us:8.1f} us  {2.0*m*n*k/us/1e6:7.1f} TF/s")
        run("lidk plain", m, n, k) if k <= 1024 else None
    for (m, n1, n2) in ((M, 1024, 256), (M, 256, 1024), (M, 256, 256)):
        X = torch.randn(m, n1, device=dev).bfloat16(); Y = torch.randn(m, n2, device=dev).bfloat16(); out = torch.empty(n1, n2, device=dev, dtype=torch.bfloat16)
        us = t(lambda: torch.mm(X.t(), Y, out=out))
        print(f"torch.mm (hipBLASLt) TN             M={m:5d} N1={n1:4d} N2={n2:4d}  {us:8.1f} us  {2.0*m*n1*n2/us/1e6:7.1f} TF/s")
if os.environ.get("NT_ONLY"): sys.exit(0)

def run_tn(name, m, n1, n2, splitk):
    X = torch.randn(m, n1, device=dev).bfloat16(); Y = torch.randn(m, n2, device=dev).bfloat16()
    C = torch.zeros(n1, n2, device=dev); cs = torch.zeros(n1, device=dev)
    us = t(lambda: ops.gemm_tn(X, Y, C, colsum=cs, splitk=splitk))
    fl = 2.0 * m * n1 * n2
    print(f"{name:34s} M={m:5d} N1={n1:4d} N2={n2:4d} sk={splitk:2d} {us:8.1f} us  {fl/us/1e6:7.1f} TF/s")
for sk in (4, 8, 16, 32):
    run_tn("TN dW1 [1024,256]", M, 1024, 256, sk)
    run_tn("TN dW2 [256,1024]", M, 256, 1024, sk)
for sk in (8, 16):
    run_tn("TN dWqkv [768,256]", M, 768, 256, sk)
    run_tn("TN dWpw1 [512,256]", M, 512, 256, sk)
for sk in (8, 16, 32):
    run_tn("TN dWo [256,256]", M, 256, 256, sk)
