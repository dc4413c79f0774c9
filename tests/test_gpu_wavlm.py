"""WavLM backbone forward on the GPU against stage outputs of the REFERENCE lid/wavlm/WavLM.py
(tests/golden/wavlm_fwd.npz, written by oracle/gen_golden_wavlm.py from the imported reference) at WavLM-Base+ width, and
op-level checks of the new kernels against torch.  bf16 GEMM operands / activations, f32 residual stream and statistics:
stage tolerances are absolute on O(1) activations."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_npz
import wavlm_case as wc
from lidk import _lib as L
from lidk import ops
from lidk.wavlm import WavLMBackbone, relative_buckets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_conv0_groupnorm_gelu_against_torch():
    torch.manual_seed(0)
    B, Lw, C = 3, 4000, 512
    wav = torch.randn(B, Lw)
    w = torch.randn(C, 1, 10) * 0.4
    gamma, beta = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    ref = F.gelu(F.group_norm(F.conv1d(wav[:, None], w, stride=5), C, gamma, beta, 1e-5)).transpose(1, 2)      # (B, T0, C)
    T0 = ref.shape[1]
    P0 = T0 + 5
    out = torch.full((B * P0 + 8, C), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.wavlm_conv0(wav.to(DEV), w.reshape(C, 10).to(DEV), gamma.to(DEV), beta.to(DEV), out, T0, P0)
    got = out[:B * P0].view(B, P0, C).float().cpu()
    assert float((got[:, :T0] - ref).abs().max()) <= 2e-2
    assert float(got[:, T0:].abs().max()) == 0.0                          # pitch padding rows are zero


def test_strided_view_gemm_is_the_convolution():
    """Conv1d(C, C, k3, s2) / (k2, s2) over a channel-last signal = lidk_gemm_nt on a strided view (lda = 2C < K = kW*C)."""
    torch.manual_seed(1)
    C, T_in = 512, 201
    x = torch.randn(1, C, T_in)
    for kw in (3, 2):
        w = torch.randn(C, C, kw) / math.sqrt(C * kw)
        ref = F.gelu(F.conv1d(x.bfloat16().float(), w.bfloat16().float(), stride=2)).transpose(1, 2)[0]       # (T_out, C)
        T_out = ref.shape[0]
        buf = torch.zeros(T_in + 8, C, device=DEV, dtype=torch.bfloat16)
        buf[:T_in] = x[0].t().to(DEV)
        A = buf.as_strided((T_out, kw * C), (2 * C, 1))
        Wk = w.permute(0, 2, 1).reshape(C, kw * C).to(DEV).bfloat16().contiguous()
        out = torch.empty(T_out, C, device=DEV)
        ops.gemm_nt(A, Wk, out, act=L.ACT_GELU)
        assert float((out.cpu() - ref).abs().max()) <= 1e-2


def test_gate_and_relative_bias_table_against_the_reference():
    g = load_npz("wavlm_fwd.npz")
    w = wc.backbone_weights()
    x0 = torch.from_numpy(g["enc_in"]).to(DEV)                              # layer-0 input (B, T, 768)
    B, T, d = x0.shape
    a = "encoder.layers.0.self_attn."
    gate = torch.empty(B, 12, T, device=DEV)
    ops.wavlm_gate(x0.reshape(B * T, d).contiguous(), w[a + "grep_linear.weight"].to(DEV), w[a + "grep_linear.bias"].to(DEV),
                   w[a + "grep_a"].reshape(-1).to(DEV), gate, B, T, 12, 64)
    assert float((gate.cpu() - torch.from_numpy(g["gate0"])).abs().max()) <= 1e-5
    # pos_bias0[h][i][j] = emb[bucket(j - i)][h]: the 1-D table the attention kernel indexes by the offset
    emb = w[a + "relative_attention_bias.weight"]
    RB = 1024
    r = torch.arange(-(RB - 1), RB)
    rb = emb[relative_buckets(r, 320, 800)].t()                              # (H, 2*RB-1)
    i, j = torch.arange(T)[:, None], torch.arange(T)[None, :]
    table = rb[:, (j - i) + RB - 1]                                          # (H, T, T)
    assert float((table - torch.from_numpy(g["pos_bias0"])).abs().max()) == 0.0


def test_attention_with_gated_bias_against_torch():
    torch.manual_seed(2)
    for B, T, H in ((2, 49, 12), (1, 149, 12), (2, 200, 3)):
        dh, inner = 64, H * 64
        qkv = (0.8 * torch.randn(B * T, 3 * inner)).bfloat16()
        gate = 1.0 + torch.rand(B, H, T)
        RB = 256
        rb = 0.7 * torch.randn(H, 2 * RB - 1)
        q, k, v = (t.float().reshape(B, T, H, dh).transpose(1, 2) for t in qkv.split(inner, dim=-1))
        i, j = torch.arange(T)[:, None], torch.arange(T)[None, :]
        bias = gate[..., None] * rb[:, (j - i) + RB - 1][None]
        ref = ((q @ k.transpose(-1, -2) / 8.0 + bias).softmax(-1) @ v).transpose(1, 2).reshape(B * T, inner)
        out = torch.empty(B * T, inner, device=DEV, dtype=torch.bfloat16)
        ops.wavlm_attn_fwd(qkv.to(DEV), gate.to(DEV), rb.to(DEV), out, B, T, H, dh)
        err = float((out.float().cpu() - ref).abs().max())
        print(f"[wavlm attn B={B} T={T} H={H}] max_abs_err={err:.3e}")
        assert err <= 2e-2


def test_backbone_forward_stages_against_the_reference():
    g = load_npz("wavlm_fwd.npz")
    bb = WavLMBackbone(wc.CFG)
    bb.load_state_dict(wc.backbone_weights())
    bb.to(DEV)
    taps = {}
    out = bb.forward(wc.waveforms().to(DEV), taps)
    torch.cuda.synchronize()
    assert out.shape == (wc.B, 49, 768) and bb.frame_counts(wc.SAMPLES) == [3199, 1599, 799, 399, 199, 99, 49]
    tol = {"conv": 4e-2, "proj": 6e-2, "enc_in": 5e-2, "gate0": 2e-2, "layer0": 6e-2, "layer1": 8e-2}
    for key, t in tol.items():
        ref = torch.from_numpy(g[key])
        got = taps[key].float().cpu()
        err, scale = float((got - ref).abs().max()), float(ref.abs().max())
        rel = float((got - ref).norm() / ref.norm())
        print(f"[wavlm stage {key}] max_abs_err={err:.3e} (max |ref| {scale:.2f}) rel_l2={rel:.3e}")
        assert err <= t and rel <= 1.5e-2, key
    ref = torch.from_numpy(g["features"])
    assert float((out.cpu() - ref).abs().max()) <= 8e-2
    # a second call with another batch shape reuses nothing stale
    out2 = bb.forward(wc.waveforms()[:1, :12000].contiguous().to(DEV))
    assert out2.shape == (1, 37, 768) and bool(torch.isfinite(out2).all())
