#!/usr/bin/env python3
"""Graph-timed micro-benchmark of the non-GEMM kernels at the cfg2 shapes (B=64, T=151, d=256)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops, _lib as L

dev = "cuda:0"
B, T, d, ci, K = 64, 151, 256, 256, 31
M = B * T
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
f32 = lambda *s: torch.randn(*s, device=dev)

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

def show(name, us, mbytes):
    print(f"{name:34s} {us:7.1f} us   {mbytes:6.1f} MB algorithmic -> {mbytes / us * 1e-3 * 1e3:7.0f} GB/s")

g_, c_, dc_, dg_ = bf(M, ci), bf(M, ci), bf(M, ci), bf(M, ci)
w, bias = f32(ci, K), f32(ci)
parts = ops.dwconv_stat_parts(B, T, ci, torch.bfloat16)
stat = torch.empty(parts * 2 * ci, device=dev)
show("dwconv fwd (+BN partial sums)", t(lambda: ops.dwconv_fwd(g_, w, bias, c_, stat, B, T, K // 2)), 2 * M * ci * 2 / 1e6)
show("dwconv fwd (no stats)", t(lambda: ops.dwconv_fwd(g_, w, bias, c_, None, B, T, K // 2)), 2 * M * ci * 2 / 1e6)
show("dwconv dgrad", t(lambda: ops.dwconv_bwd_input(dc_, w, dg_, B, T, K // 2)), 2 * M * ci * 2 / 1e6)
dw, db = torch.zeros(ci, K, device=dev), torch.zeros(ci, device=dev)
part = torch.empty(B * ci * (K + 1), device=dev)
show("dwconv wgrad (+finalize)", t(lambda: ops.dwconv_bwd_weight(dc_, g_, dw, db, part, B, T, K // 2)), 2 * M * ci * 2 / 1e6)
y, gg = bf(M, 2 * ci), bf(M, ci)
show("glu fwd", t(lambda: ops.glu_fwd(y, gg)), 3 * M * ci * 2 / 1e6)
dy = bf(M, 2 * ci)
show("glu bwd", t(lambda: ops.glu_bwd(y, gg, dy)), 5 * M * ci * 2 / 1e6)
x, gam, bet = f32(M, d), f32(d), f32(d)
h, mean, rstd = bf(M, d), f32(M), f32(M)
show("layernorm fwd", t(lambda: ops.layernorm_fwd(x, gam, bet, yT=h, mean=mean, rstd=rstd)), (M * d * 6) / 1e6)
partial = torch.empty(max(L.LN_PARTIAL_BLOCKS * 2 * 1024, L.LN_BWD_BLOCKS * 2 * d), device=dev)
dyh, dres, dx, dxT, dga, dbe = bf(M, d), f32(M, d), f32(M, d), bf(M, d), torch.zeros(d, device=dev), torch.zeros(d, device=dev)
show("layernorm bwd (+colreduce)", t(lambda: ops.layernorm_bwd(dyh, x, mean, rstd, gam, partial, dres=dres, dx=dx, dxT=dxT, dgamma=dga, dbeta=dbe)), (M * d * (2 + 4 + 4 + 4 + 2)) / 1e6)
bm, br = f32(ci), f32(ci).abs() + 0.5
s_ = bf(M, ci)
show("bn+swish fwd", t(lambda: ops.bn_swish_fwd(c_, bm, br, gam, bet, s_)), 2 * M * ci * 2 / 1e6)
show("bn+swish bwd reduce", t(lambda: ops.bn_swish_bwd_reduce(dc_, c_, bm, br, gam, bet, partial)), 2 * M * ci * 2 / 1e6)
sums, sl = torch.zeros(2 * ci, device=dev, dtype=torch.float64), torch.zeros(2 * ci, device=dev, dtype=torch.float64)
show("bn+swish bwd apply", t(lambda: ops.bn_swish_bwd_apply(dc_, c_, bm, br, gam, bet, sums, sl, M, dg_, dga, dbe)), 3 * M * ci * 2 / 1e6)
wav = f32(B, 48000) * 0.1
show("normalize_wav", t(lambda: ops.normalize_wav(wav)), 2 * B * 48000 * 4 / 1e6)
nw = ops.normalize_wav(wav)
show("dither + pre-emphasis", t(lambda: ops.dither_preemph(nw, coef=0.97, dither=1e-5, seed=7)), 2 * B * 48000 * 4 / 1e6)
show("logmel (stft+mel+dB+floor)", t(lambda: ops.logmel(nw, pad=16, n_mels=80, spans=None)), (B * 48000 * 4 + B * 301 * 80 * 4) / 1e6)
