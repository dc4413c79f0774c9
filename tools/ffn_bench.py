#!/usr/bin/env python3
"""Graph-timed micro-benchmark of the FeedForward module at the cfg2 shape (M = 9664, d = 256, ff = 1024): the fused kernels of
csrc/ffn.hip against the launch sequences they replace.  Each sample rotates over 4 independent buffer sets so that the timed
launches do not hit a warm L2 / MALL that a training step would not have."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops, _lib as L

dev = "cuda:0"
M, d, ff = int(os.environ.get("M", 9664)), 256, int(os.environ.get("FF", 1024))
NSET = 4
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
f32 = lambda *s: torch.randn(*s, device=dev)


def t(fn, n=12):
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


S = []
for _ in range(NSET):
    s = dict(x=f32(M, d), gam=f32(d), bet=f32(d), W1=bf(ff, d) / 16, b1=f32(ff), W2=bf(d, ff) / 32, b2=f32(d), h=bf(M, d), mean=f32(M),
             rstd=f32(M), a=bf(M, ff), u=bf(M, ff), xo=f32(M, d), dyT=bf(M, d), da=bf(M, ff), dh=bf(M, d), dres=f32(M, d), dx=f32(M, d),
             dxT=bf(M, d))
    s["W1T"], s["W2T"] = s["W1"].t().contiguous(), s["W2"].t().contiguous()
    S.append(s)
partial = torch.empty(max(L.LN_PARTIAL_BLOCKS * 2 * 1024, L.LN_BWD_BLOCKS * 2 * d), device=dev)
dga, dbe = torch.zeros(d, device=dev), torch.zeros(d, device=dev)


def fwd3(i):
    s = S[i]
    ops.layernorm_fwd(s["x"], s["gam"], s["bet"], yT=s["h"], mean=s["mean"], rstd=s["rstd"])
    ops.gemm_nt(s["h"], s["W1"], s["u"], bias=s["b1"], act=L.ACT_SWISH, out2=s["a"])
    ops.gemm_nt(s["u"], s["W2"], s["xo"], bias=s["b2"], alpha=0.5, res=s["x"])


def fwd1(i):
    s = S[i]
    ops.ffn_fwd(s["x"], s["W1"], s["b1"], s["W2"], s["b2"], s["xo"], gamma=s["gam"], beta=s["bet"], h=s["h"], mean=s["mean"],
                rstd=s["rstd"], a=s["a"], u=s["u"])


def bwd3(i):
    s = S[i]
    ops.gemm_nt(s["dyT"], s["W2T"], s["da"], act=L.ACT_SWISH_GRAD, aux=s["a"], N=ff, K=d)
    ops.gemm_nt(s["da"], s["W1T"], s["dh"], N=d, K=ff)
    ops.layernorm_bwd(s["dh"], s["x"], s["mean"], s["rstd"], s["gam"], partial, dres=s["dres"], dx=s["dx"], dxT=s["dxT"], dgamma=dga,
                      dbeta=dbe)


flop = 2 * 2 * M * d * ff
RG = os.environ.get("LIDK_FFN_RG", "by M")
print(f"M = {M}, ff = {ff}, workgroup height LIDK_FFN_RG = {RG}: {ops.ffn_bwd_partial_rows(M)} workgroups")
us3, us1 = t(fwd3), t(fwd1)
print(f"forward  LN + up + down (3 launches) {us3:7.1f} us   {flop / us3 * 1e-6:6.0f} TFLOP/s")
print(f"forward  fused lidk_ffn_fwd          {us1:7.1f} us   {flop / us1 * 1e-6:6.0f} TFLOP/s")
usb = t(bwd3)
print(f"backward dgrad x 2 + LN bwd (4 launches incl. colreduce) {usb:7.1f} us")
if hasattr(ops, "ffn_bwd"):
    def bwd1(i):
        s = S[i]
        ops.ffn_bwd(s["dyT"], s["a"], s["W1T"], s["W2T"], s["da"], x=s["x"], mean=s["mean"], rstd=s["rstd"], gamma=s["gam"],
                    dres=s["dres"], dx=s["dx"], dxT=s["dxT"], partial=partial)
    print(f"backward fused lidk_ffn_bwd          {t(bwd1):7.1f} us")


def fwd1h(i):
    s = S[i]
    ops.ffn_fwd(s["x"], s["W1"], s["b1"], s["W2"], s["b2"], s["xo"], h_in=s["h"], a=s["a"], u=s["u"])


def bwd1dh(i):
    s = S[i]
    ops.ffn_bwd(s["dyT"], s["a"], s["W1T"], s["W2T"], s["da"], dh=s["dh"])


def dgln(i):
    s = S[i]
    ops.dgrad_ln_bwd(s["da"], s["W1T"], s["x"], s["mean"], s["rstd"], s["gam"], partial, dres=s["dres"], dx=s["dx"], dxT=s["dxT"])


print(f"forward  fused, h from the previous epilogue {t(fwd1h):7.1f} us")
print(f"backward fused, stops at dh                  {t(bwd1dh):7.1f} us")
print(f"dgrad (K = {ff}) + LayerNorm backward          {t(dgln):7.1f} us")
