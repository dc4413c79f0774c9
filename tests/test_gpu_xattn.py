"""Key-tiled attention of the transformer backbones (lidk_xattn_fwd / lidk_xattn_bwd, csrc/xattn.hip) against a plain torch
fp32 restatement with autograd (lid/wavlm/modules.py:505-700: q.k^T / sqrt(dh) + gated relative-position bias, softmax, attention
dropout, .v; wav2vec2: the same without a bias and with a key padding mask).  bf16 operands, f32 accumulation: outputs within
2e-2 of O(1) values, gradients by relative L2 error."""
import math

import pytest
import torch

from lidk import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(qkv, B, T, H, gate, rb, klen, keep, drop_p, dout):
    dh, inner = 64, H * 64
    x = qkv.float().clone().requires_grad_(True)
    q, k, v = (t.reshape(B, T, H, dh).transpose(1, 2) for t in x.split(inner, dim=-1))
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    gg = rr = None
    if gate is not None:
        gg, rr = gate.clone().requires_grad_(True), rb.clone().requires_grad_(True)
        RB = (rb.shape[1] + 1) // 2
        i, j = torch.arange(T)[:, None], torch.arange(T)[None, :]
        s = s + gg[..., None] * rr[:, (j - i) + RB - 1][None]
    if klen is not None:
        pad = torch.arange(T)[None, :] >= klen[:, None].long()
        s = s.masked_fill(pad[:, None, None, :], float("-inf"))
    a = s.softmax(-1)
    lse = torch.logsumexp(s, -1)
    if keep is not None:
        a = a * keep.float() / (1.0 - drop_p)
    o = (a @ v).transpose(1, 2).reshape(B * T, inner)
    o.backward(dout.float())
    return o.detach(), lse.detach(), x.grad, (gg.grad if gg is not None else None), (rr.grad if rr is not None else None)


CASES = [  # B, T, H, bias, ragged keys, dropout
    (2, 49, 12, True, False, 0.0),
    (1, 149, 12, True, False, 0.0),
    (2, 64, 2, False, False, 0.0),
    (3, 300, 3, True, True, 0.0),
    (2, 700, 2, False, True, 0.0),
    (2, 149, 4, True, False, 0.1),
    (2, 130, 3, False, True, 0.1),
]


@pytest.mark.parametrize("B,T,H,bias,ragged,drop_p", CASES)
def test_xattn_forward_and_backward_against_torch(B, T, H, bias, ragged, drop_p):
    torch.manual_seed(B * 1000 + T)
    dh, inner = 64, H * 64
    qkv = (0.8 * torch.randn(B * T, 3 * inner)).bfloat16()
    dout = (0.5 * torch.randn(B * T, inner)).bfloat16()
    gate = rb = klen = keep = None
    RB = 1024 if T > 256 else 256
    if bias:
        gate = 1.0 + torch.rand(B, H, T)
        rb = 0.7 * torch.randn(H, 2 * RB - 1)
    if ragged:
        klen = torch.tensor([T] + [max(1, int(T * f)) for f in (0.37, 0.81)][:B - 1], dtype=torch.int32)
    if drop_p > 0:
        keep = (torch.rand(B, H, T, T) >= drop_p).to(torch.uint8)
    ref_o, ref_lse, ref_dqkv, ref_dgate, ref_drb = _reference(qkv, B, T, H, gate, rb, klen, keep, drop_p, dout)
    d = lambda t: None if t is None else t.to(DEV).contiguous()
    out = torch.empty(B * T, inner, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device=DEV)
    kw = dict(gate=d(gate), rb=d(rb), klen=d(klen), keep=d(keep), drop_p=drop_p, seed=0)
    ops.xattn_fwd(d(qkv), out, lse, B, T, H, dh, **kw)
    e_o = float((out.float().cpu() - ref_o).abs().max())
    e_l = float((lse.cpu() - ref_lse).abs().max())
    dqkv = torch.full((B * T, 3 * inner), 7.0, device=DEV, dtype=torch.bfloat16)
    delta = torch.empty(B, H, T, device=DEV)
    dgate = torch.zeros(B, H, T, device=DEV) if bias else None
    drb = torch.zeros(H, 2 * RB - 1, device=DEV) if bias else None
    # the backward consumes the bf16 `out` the forward wrote (delta = dO . O), as the product does
    ops.xattn_bwd(d(qkv), out, d(dout), lse, dqkv, delta, B, T, H, dh, dgate=dgate, drb=drb, **kw)
    torch.cuda.synchronize()
    rel = lambda got, ref: float((got.float().cpu() - ref).norm() / (ref.norm() + 1e-30))
    names = ("dq", "dk", "dv")
    errs = {n: rel(dqkv[:, c * inner:(c + 1) * inner], ref_dqkv[:, c * inner:(c + 1) * inner]) for c, n in enumerate(names)}
    msg = f"[xattn B={B} T={T} H={H} bias={bias} ragged={ragged} p={drop_p}] out {e_o:.3e} lse {e_l:.3e} " + \
          " ".join(f"{n} {v:.3e}" for n, v in errs.items())
    if bias:
        errs["dgate"], errs["drb"] = rel(dgate, ref_dgate), rel(drb, ref_drb)
        msg += f" dgate {errs['dgate']:.3e} drb {errs['drb']:.3e}"
    print(msg)
    assert e_o <= 2e-2 and e_l <= 2e-2
    assert all(v <= 2e-2 for v in errs.values()), errs
    if ragged:                                    # padded keys receive exactly zero gradient
        for b in range(B):
            n = int(klen[b])
            assert float(dqkv[b * T + n:(b + 1) * T, inner:].abs().max() if n < T else 0.0) == 0.0


def test_xattn_counter_based_dropout_is_the_same_mask_forward_and_backward():
    """drop_p > 0 without a forced mask: the decisions come from (seed, element index).  Recover the mask the forward used from
    its output on an identity-like V, feed it back as the forced mask, and require identical results from both passes."""
    torch.manual_seed(3)
    B, T, H, dh, p = 1, 64, 1, 64, 0.25
    qkv = (0.5 * torch.randn(B * T, 3 * 64)).bfloat16().to(DEV)
    dout = (0.5 * torch.randn(B * T, 64)).bfloat16().to(DEV)
    out1 = torch.empty(B * T, 64, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device=DEV)
    ops.xattn_fwd(qkv, out1, lse, B, T, H, dh, drop_p=p, seed=1234)
    # the same decisions, reproduced on the host from the documented generator
    idx = torch.arange(T * T, dtype=torch.int64)
    M = (1 << 32) - 1
    keep = []
    for e in idx.tolist():                                             # common.h uniform32_from(seed = 1234, index e)
        h = (e * 0x9E3779B1 + 1234) & M
        h ^= h >> 16; h = (h * 0x21F0AAAD) & M
        h ^= 0                                                         # seed >> 32 and index >> 32 are both zero here
        h ^= h >> 15; h = (h * 0x735A2D97) & M
        h ^= h >> 15
        keep.append(((h >> 8) / 16777216.0) >= p)
    keep = torch.tensor(keep, dtype=torch.uint8).view(1, 1, T, T).to(DEV)
    out2 = torch.empty_like(out1)
    ops.xattn_fwd(qkv, out2, lse, B, T, H, dh, keep=keep, drop_p=p)
    assert torch.equal(out1, out2)
    d1, d2 = torch.empty(B * T, 192, device=DEV, dtype=torch.bfloat16), torch.empty(B * T, 192, device=DEV, dtype=torch.bfloat16)
    delta = torch.empty(B, H, T, device=DEV)
    ops.xattn_bwd(qkv, out1, dout, lse, d1, delta, B, T, H, dh, drop_p=p, seed=1234)
    ops.xattn_bwd(qkv, out1, dout, lse, d2, delta, B, T, H, dh, keep=keep, drop_p=p)
    torch.cuda.synchronize()
    assert torch.equal(d1, d2)
    assert 0.15 < 1.0 - float(keep.float().mean()) < 0.35
