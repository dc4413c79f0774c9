#!/usr/bin/env python3
"""Micro-benchmark of the convolution-module kernels at the cfg2 shape (B=32, T=302, C=512, K=31; env B, T), hipGraph-replayed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops, _lib as L

dev = "cuda:0"
B, T, C, K = int(os.environ.get("B", 32)), int(os.environ.get("T", 302)), 512, 31
M = B * T
bf = torch.bfloat16


def t(fn, n=20):
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
    return a.elapsed_time(b) / (5 * n) * 1e3


def line(name, us, mb):
    print(f"{name:44s} {us:7.1f} us   {mb:6.1f} MB algorithmic -> {mb / us * 1e3 / 1e3:6.2f} TB/s")


torch.manual_seed(0)
y = torch.randn(M, 2 * C, device=dev).to(bf)
w = torch.randn(C, K, device=dev) * 0.2
bias = torch.randn(C, device=dev) * 0.1
g = torch.empty(M, C, device=dev, dtype=bf)
c = torch.empty(M, C, device=dev, dtype=bf)
parts = ops.dwconv_stat_parts(B, T, C, torch.bfloat16)
sp = torch.empty(parts * 2 * C, device=dev)
line("glu_dwconv_fwd (+g, +stats)", t(lambda: ops.glu_dwconv_fwd(y, w, bias, g, c, sp, B, T, K // 2)), (M * 2 * C + 2 * M * C) * 2 / 1e6)
line("glu_dwconv_fwd (eval: no g, no stats)", t(lambda: ops.glu_dwconv_fwd(y, w, bias, None, c, None, B, T, K // 2)), (M * 2 * C + M * C) * 2 / 1e6)
mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(1, device=dev, dtype=torch.long)
line("bn_train_stats_from_partials", t(lambda: ops.bn_train_stats_from_partials(sp, parts, M, mean, rstd, rm, rv, nbt)), parts * 2 * C * 4 / 1e6)
gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
s = torch.empty(M, C, device=dev, dtype=bf)
line("bn_swish_fwd", t(lambda: ops.bn_swish_fwd(c, mean, rstd, gamma, beta, s)), 2 * M * C * 2 / 1e6)
ds = torch.randn(M, C, device=dev).to(bf)
partial = torch.empty(L.BN_PARTIAL_BLOCKS * 2 * C + 4096, device=dev)
line("bn_swish_bwd_reduce", t(lambda: ops.bn_swish_bwd_reduce(ds, c, mean, rstd, gamma, beta, partial)), 2 * M * C * 2 / 1e6)
sums = torch.zeros(2 * C + 1, device=dev, dtype=torch.float64)
sums_l = torch.zeros(2 * C + 1, device=dev, dtype=torch.float64)
line("reduce_partials_f64", t(lambda: ops.reduce_partials_f64(partial, L.BN_PARTIAL_BLOCKS, 2 * C, sums, sums_l, tail=M)), L.BN_PARTIAL_BLOCKS * 2 * C * 4 / 1e6)
dy = torch.empty(M, 2 * C, device=dev, dtype=bf)
line("dwconv_bwd_input_bn_glu", t(lambda: ops.dwconv_bwd_input_bn_glu(ds, c, mean, rstd, gamma, beta, sums, 0, w, y, dy, B, T, K // 2)),
     (2 * M * C + 2 * M * 2 * C) * 2 / 1e6)
dc = torch.empty(M, C, device=dev, dtype=bf)
dg_, db_ = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
line("bn_swish_bwd_apply (-> dc, dgamma, dbeta)", t(lambda: ops.bn_swish_bwd_apply(ds, c, mean, rstd, gamma, beta, sums, sums_l, 0, dc, dg_, db_)), 3 * M * C * 2 / 1e6)
dw, dwb = torch.zeros(C, K, device=dev), torch.zeros(C, device=dev)
wp = torch.empty(B * C * (K + 1) + 4096, device=dev)
line("dwconv_bwd_weight", t(lambda: ops.dwconv_bwd_weight(dc, g, dw, dwb, wp, B, T, K // 2)), 2 * M * C * 2 / 1e6)
x = torch.empty(M * C, device=dev, dtype=bf); x2 = torch.empty_like(x)
line("copy M*C bf16 (reference)", t(lambda: x2.copy_(x)), 2 * M * C * 2 / 1e6)
dgm, dbt = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
line("dwconv_bwd_weight_bn (fused apply + wgrad)", t(lambda: ops.dwconv_bwd_weight_bn(ds, c, mean, rstd, gamma, beta, sums, sums_l, 0, g, dw, dwb, dgm, dbt, wp, B, T, K // 2)), 3 * M * C * 2 / 1e6)
