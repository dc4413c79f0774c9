#!/usr/bin/env python3
"""lidk_gemm_nt on the WavLM conv feature extractor's shapes (strided-view convolutions, B = 64 x 3 s): per-layer time and TF/s.
Env LIDK_GEMM_DIRECT=0 LIDK_GEMM_TILE=128 selects the 128x128-tile kernel for comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops, _lib as L

dev = "cuda:0"
B, C = 64, 512


def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


P = [9600, 4800, 2400, 1200, 600, 300, 150]
tot = 0.0
for l in range(1, 7):
    kw = 3 if l < 5 else 2
    rows_in = B * P[l - 1] + 8
    x = torch.randn(rows_in, C, device=dev).bfloat16()
    W = (torch.randn(C, kw * C, device=dev) / (kw * C) ** 0.5).bfloat16()
    M = B * P[l]
    A = x.as_strided((M, kw * C), (2 * C, 1))
    out = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    us = t(lambda: ops.gemm_nt(A, W, out, act=L.ACT_GELU))
    fl = 2.0 * M * C * kw * C
    tot += us
    print(f"conv layer {l}: M={M:7d} N={C} K={kw * C}: {us:9.1f} us  {fl / us / 1e6:7.1f} TF/s")
print(f"conv layers 1-6 total {tot / 1e3:.2f} ms")

print("transformer layer GEMMs (M = 64 x 149 = 9536, d = 768, ffn 3072)")
M = 64 * 149
def run(name, m, n, k, **kw):
    A = torch.randn(m, k, device=dev).bfloat16(); Bm = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    f32 = kw.pop("out_f32", False)
    out = torch.empty(m, n, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    extra = {}
    if kw.pop("res", False): extra["res"] = torch.randn(m, n, device=dev)
    if kw.get("act") == L.ACT_GELU: extra["out2"] = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    us = t(lambda: ops.gemm_nt(A, Bm, out, bias=bias, **kw, **extra), n=10)
    ven = t(lambda: torch.nn.functional.linear(A, Bm), n=10)          # yardstick only: the vendor GEMM torch dispatches to, no epilogue
    print(f"{name:22s} M={m} N={n:5d} K={k:5d}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TF/s   (vendor GEMM, plain: {ven:7.1f} us)")
run("qkv", M, 2304, 768)
run("out (+res f32)", M, 768, 768, res=True, out_f32=True)
run("fc1 (gelu, +pre)", M, 3072, 768, act=L.ACT_GELU)
run("fc2 (+res f32)", M, 768, 3072, res=True, out_f32=True)
run("dgrad fc1 (gelu')", M, 768, 3072)
run("dgrad fc2", M, 3072, 768)
if os.environ.get("XLSR", "1") == "1":
    print("XLS-R width (M = 16 000, d = 1024, ffn 4096)")
    M2 = 16000
    run("qkv", M2, 3072, 1024)
    run("out (+res f32)", M2, 1024, 1024, res=True, out_f32=True)
    run("fc1 (gelu, +pre)", M2, 4096, 1024, act=L.ACT_GELU)
    run("fc2 (+res f32)", M2, 1024, 4096, res=True, out_f32=True)

print("weight-gradient (TN) GEMMs, M = 9536; LIDK_TN_TILE=128 forces the 128x128 tile")
def run_tn(name, n1, n2, sk):
    X = torch.randn(M, n1, device=dev).bfloat16(); Y = torch.randn(M, n2, device=dev).bfloat16()
    Cm = torch.zeros(n1, n2, device=dev); cs = torch.zeros(n1, device=dev)
    us = t(lambda: ops.gemm_tn(X, Y, Cm, colsum=cs, splitk=sk), n=10)
    print(f"{name:12s} [{n1:5d},{n2:5d}] sk={sk:2d}: {us:8.1f} us  {2.0 * M * n1 * n2 / us / 1e6:7.1f} TF/s")
for sk in (1, 2, 4):
    run_tn("dW fc1", 3072, 768, sk); run_tn("dW fc2", 768, 3072, sk); run_tn("dW qkv", 2304, 768, sk); run_tn("dW out", 768, 768, sk)
