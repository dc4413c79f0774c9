#!/usr/bin/env python3
"""Graph-timed key-tiled attention (csrc/xattn.hip) at backbone shapes: forward / backward us per launch and TFLOP/s.
    python tools/xattn_bench.py [B T H]      (default 64 250 12; dh = 64; env DROP=0.1 adds attention dropout, KLEN=1 a ragged key mask)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "speech-lid_amd")]
import torch
from lidk import ops
dev = "cuda:0"
B, T, H = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 250, 12)
dh, p = 64, float(os.environ.get("DROP", "0"))
qkv = (0.5 * torch.randn(B * T, 3 * H * dh, device=dev)).bfloat16()
out = torch.empty(B * T, H * dh, device=dev, dtype=torch.bfloat16)
dout = (0.5 * torch.randn(B * T, H * dh, device=dev)).bfloat16()
dqkv = torch.empty_like(qkv)
lse, delta = torch.empty(B, H, T, device=dev), torch.empty(B, H, T, device=dev)
klen = None
if os.environ.get("KLEN") == "1":
    klen = torch.randint(T // 2, T + 1, (B,), device=dev, dtype=torch.int32)


def timed(fn, n=6):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (4 * n) * 1e3


f = timed(lambda: ops.xattn_fwd(qkv, out, lse, B, T, H, dh, klen=klen, drop_p=p, seed=7))
bw = timed(lambda: ops.xattn_bwd(qkv, out, dout, lse, dqkv, delta, B, T, H, dh, klen=klen, drop_p=p, seed=7))
fl = 4.0 * B * H * T * T * dh
print(f"B={B} T={T} H={H} drop={p} klen={'ragged' if klen is not None else 'none'}: fwd {f:7.1f} us ({fl / f / 1e6:5.0f} TFLOP/s)   "
      f"bwd (q + kv kernels) {bw:7.1f} us ({2.5 * fl / bw / 1e6:5.0f} TFLOP/s)")
