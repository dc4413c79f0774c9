"""Per-kernel parity tests (GPU): every C-ABI op against a torch-CPU / oracle computation of the same math.

f32 mode must agree to ~1e-5 (it is the algorithm check); bf16 mode is compared against the same reference
evaluated on bf16-rounded inputs, with tolerances sized by bf16's 2^-8 relative rounding.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import lidk
from lidk import ops
from lidk import _lib as L
from lidk._lib import LidkError
from oracle import conformer as oc
from oracle import features as of
from oracle import optim as oo

pytestmark = pytest.mark.gpu


@pytest.fixture
def gemm_options():
    """Set lidk_gemm_nt's kernel-family knobs for one test (lidk_gemm_option); afterwards they are re-read from the environment."""
    touched = []

    def set_(name, value):
        ops.gemm_option(name, value)
        touched.append(name)

    yield set_
    for name in touched:
        ops.gemm_option(name, -1)
DEV = "cuda:0"
DT = [torch.float32, torch.bfloat16]


def g(seed=0):
    return torch.Generator().manual_seed(seed)


def dev(t, dtype=None):
    t = t.to(DEV)
    return t.to(dtype).contiguous() if dtype is not None else t.contiguous()


def rt(t, dtype):
    """round-trip through the activation dtype on CPU (reference sees exactly what the kernel sees)"""
    return t.to(dtype).float()


def tol(dtype, f32, bf16):
    return f32 if dtype == torch.float32 else bf16


def check(name, got, ref, atol, rtol=0.0):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs()
    lim = atol + rtol * ref.abs()
    worst = float((err - lim).max())
    print(f"[{name}] max_abs_err={float(err.max()):.3e} ref_max={float(ref.abs().max()):.3e}")
    assert worst <= 0, f"{name}: max abs err {float(err.max()):.3e} exceeds atol={atol} rtol={rtol}"


# ------------------------------------------------------------------------------------------------ element-wise
@pytest.mark.parametrize("dt", DT)
def test_scale_cast_dropout_relu(dt):
    x = torch.randn(1003, generator=g(1))
    out = torch.empty(1003, device=DEV, dtype=dt)
    ops.scale_cast(dev(x), out, 0.5)
    check("scale_cast", out, rt(0.5 * x, dt), 1e-6 if dt == torch.float32 else 0, 0)
    xd = dev(x, dt)
    y = torch.empty_like(xd)
    keep = torch.empty(1003, device=DEV, dtype=torch.uint8)
    ops.dropout(xd, y, 0.25, seed=123, keep_out=keep)
    k = keep.cpu().bool()
    assert 0.65 < k.float().mean() < 0.85
    check("dropout", y, rt(rt(x, dt) * k / 0.75, dt), 1e-6, 1e-2 if dt == torch.bfloat16 else 1e-6)
    y2 = torch.empty_like(xd)
    ops.dropout(xd, y2, 0.25, keep_in=keep)
    assert torch.equal(y, y2)
    y3 = torch.empty_like(xd)
    ops.dropout(xd, y3, 0.25, seed=123)
    assert torch.equal(y, y3)                      # stateless generator: same (seed, index) -> same mask
    dy = torch.randn(1003, generator=g(2))
    dx = torch.empty_like(xd)
    ops.relu_bwd(dev(dy, dt), xd, dx)
    check("relu_bwd", dx, rt(dy, dt) * (rt(x, dt) > 0), 0)


@pytest.mark.parametrize("dt", DT)
def test_colsum_transpose(dt):
    x = torch.randn(777, 130, generator=g(3))
    out = torch.full((130,), 2.0, device=DEV)
    partial = torch.empty(L.LN_PARTIAL_BLOCKS * 130, device=DEV)
    ops.colsum(dev(x, dt), out, partial, 0.5)
    check("colsum", out, 2.0 + 0.5 * rt(x, dt).double().sum(0).float(), 2e-4)
    xt = torch.empty(130, 777, device=DEV, dtype=dt)
    ops.transpose(dev(x, dt), xt)
    assert torch.equal(xt.cpu(), x.to(dt).t().contiguous())
    p = torch.randn(37, 10, generator=g(4))
    o64 = torch.empty(10, device=DEV, dtype=torch.float64)
    ops.reduce_partials_f64(dev(p), 37, 10, o64)
    check("reduce_partials_f64", o64, p.double().sum(0), 1e-6)


# ------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,C", [(204, 64), (1000, 256), (37, 144)])
def test_layernorm_fwd_bwd(dt, M, C):
    x = (2.0 * torch.randn(M, C, generator=g(5)) + 0.5).requires_grad_()
    gamma = (1 + 0.2 * torch.randn(C, generator=g(6))).requires_grad_()
    beta = (0.1 * torch.randn(C, generator=g(7))).requires_grad_()
    y = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    dy = torch.randn(M, C, generator=g(8))
    dres = torch.randn(M, C, generator=g(9))
    y.backward(rt(dy, dt))
    xd, gd, bd = dev(x.detach()), dev(gamma.detach()), dev(beta.detach())
    yT = torch.empty(M, C, device=DEV, dtype=dt)
    y32 = torch.empty(M, C, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(xd, gd, bd, yT=yT, y32=y32, mean=mean, rstd=rstd)
    check("ln_fwd_f32", y32, y, 2e-5)
    check("ln_fwd_T", yT, y, tol(dt, 2e-5, 0), tol(dt, 0, 8e-3))
    check("ln_mean", mean, x.detach().mean(1), 1e-5)
    dx = torch.empty(M, C, device=DEV)
    dxT = torch.empty(M, C, device=DEV, dtype=dt)
    dgam = torch.full((C,), 1.0, device=DEV)
    dbet = torch.full((C,), -1.0, device=DEV)
    partial = torch.empty(L.LN_BWD_BLOCKS * 2 * C, device=DEV)
    ops.layernorm_bwd(dev(dy, dt), xd, mean, rstd, gd, partial, dres=dev(dres), dx=dx, dxT=dxT, dxT_scale=0.5,
                      dgamma=dgam, dbeta=dbet)
    check("ln_bwd_dx", dx, x.grad + dres, 5e-5, 1e-5)
    check("ln_bwd_dxT", dxT, 0.5 * (x.grad + dres), tol(dt, 5e-5, 1e-3), tol(dt, 1e-5, 8e-3))
    check("ln_bwd_dgamma", dgam, 1.0 + gamma.grad, 2e-3, 1e-4)
    check("ln_bwd_dbeta", dbet, -1.0 + beta.grad, 2e-3, 1e-4)
    # f32 dy path (encoder output gradient enters the last post_norm as f32)
    dx2 = torch.empty(M, C, device=DEV)
    ops.layernorm_bwd(dev(rt(dy, dt)), xd, mean, rstd, gd, partial, dx=dx2, dtype=dt)
    check("ln_bwd_dx_f32dy", dx2, x.grad, 5e-5, 1e-5)
    # split form: partial rows now, parameter gradients later (the engine finishes them on the weight-gradient stream)
    dgam2 = torch.full((C,), 1.0, device=DEV)
    dbet2 = torch.full((C,), -1.0, device=DEV)
    ops.layernorm_bwd(dev(dy, dt), xd, mean, rstd, gd, partial, dres=dev(dres), dx=dx, dxT=dxT, dxT_scale=0.5)
    ops.layernorm_param_grads(partial, M, C, dgam2, dbet2)
    assert torch.equal(dgam2, dgam) and torch.equal(dbet2, dbet)


# ------------------------------------------------------------------------------------------------ GEMM
def _gemm_ref(A, B, dt):
    return rt(A, dt).double() @ rt(B, dt).double().t()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(300, 41, 256), (130, 70, 240), (64, 64, 64), (1000, 256, 1024), (4100, 1536, 256),
                                   (204, 768, 64), (33, 80, 80)])
def test_gemm_plain_and_tails(dt, M, N, K):
    A = torch.randn(M, K, generator=g(10)) * (1 + torch.arange(K) / K)            # asymmetric data
    B = torch.randn(N, K, generator=g(11)) + 0.1 * torch.arange(N)[:, None] / N
    ref = _gemm_ref(A, B, dt).float()
    out = torch.full((M, N), 7.0, device=DEV, dtype=dt)
    ops.gemm_nt(dev(A, dt), dev(B, dt), out)
    scale = float(ref.abs().max())
    check(f"gemm_{M}x{N}x{K}", out, ref, tol(dt, 2e-4 * scale, 6e-3 * scale))
    o32 = torch.empty(M, N, device=DEV)
    ops.gemm_nt(dev(A, dt), dev(B, dt), o32)
    check(f"gemm_f32out_{M}x{N}x{K}", o32, ref, 2e-4 * scale)


@pytest.mark.parametrize("dt", DT)
def test_gemm_epilogues(dt):
    M, N, K = 333, 200, 128
    A = torch.randn(M, K, generator=g(12)) * 0.3
    B = torch.randn(N, K, generator=g(13)) * 0.3
    bias = torch.randn(N, generator=g(14))
    res = torch.randn(M, N, generator=g(15))
    aux = torch.randn(M, N, generator=g(16))
    acc = _gemm_ref(A, B, dt).float()
    Ad, Bd = dev(A, dt), dev(B, dt)
    # bias + swish, pre-activation saved
    out = torch.empty(M, N, device=DEV, dtype=dt)
    pre = torch.empty(M, N, device=DEV, dtype=dt)
    ops.gemm_nt(Ad, Bd, out, bias=dev(bias), act=L.ACT_SWISH, out2=pre)
    a = acc + bias
    check("gemm_swish_pre", pre, a, tol(dt, 2e-4, 4e-2))
    check("gemm_swish", out, a * torch.sigmoid(a), tol(dt, 2e-4, 4e-2))
    # relu
    ops.gemm_nt(Ad, Bd, out, bias=dev(bias), act=L.ACT_RELU)
    check("gemm_relu", out, torch.relu(a), tol(dt, 2e-4, 4e-2))
    # alpha + f32 residual, f32 out
    o32 = torch.empty(M, N, device=DEV)
    ops.gemm_nt(Ad, Bd, o32, bias=dev(bias), alpha=0.5, res=dev(res))
    check("gemm_alpha_res", o32, res + 0.5 * a, 2e-4 if dt == torch.float32 else 2e-4)
    # swish-grad multiplier
    s = torch.sigmoid(rt(aux, dt))
    ops.gemm_nt(Ad, Bd, out, act=L.ACT_SWISH_GRAD, aux=dev(aux, dt))
    check("gemm_swish_grad", out, acc * s * (1 + rt(aux, dt) * (1 - s)), tol(dt, 2e-4, 4e-2))
    # split-K atomic accumulation onto an existing f32 gradient
    M2, N2, K2 = 256, 96, 4104
    A2 = torch.randn(M2, K2, generator=g(17)) * 0.1
    B2 = torch.randn(N2, K2, generator=g(18)) * 0.1
    base = torch.randn(M2, N2, generator=g(19))
    o = dev(base.clone())
    ops.gemm_nt(dev(A2, dt), dev(B2, dt), o, splitk=8)
    check("gemm_splitk", o, base + _gemm_ref(A2, B2, dt).float(), 5e-4)


@pytest.mark.parametrize("M,N", [(9664, 1024), (6400, 512), (4160, 768), (3136, 1024)])
def test_gemm_pipelined_tiles(M, N):
    """Wide K = 256 shapes take the tile-pipelined kernel (several tiles per workgroup, epilogue of tile j-1 issued inside
    tile j's K loop, clamped duplicate tiles at the end of the range): all four compiled epilogues against the reference."""
    dt, K = torch.bfloat16, 256
    A = torch.randn(M, K, generator=g(30)) * 0.3
    B = torch.randn(N, K, generator=g(31)) * 0.3
    bias = torch.randn(N, generator=g(32))
    aux = torch.randn(M, N, generator=g(33))
    acc = _gemm_ref(A, B, dt).float()
    Ad, Bd = dev(A, dt), dev(B, dt)
    out = torch.full((M, N), 5.0, device=DEV, dtype=dt)
    pre = torch.full((M, N), 5.0, device=DEV, dtype=dt)
    ops.gemm_nt(Ad, Bd, out)
    check("pipe_plain", out, acc, 4e-2)
    ops.gemm_nt(Ad, Bd, out, bias=dev(bias))
    check("pipe_bias", out, acc + bias, 4e-2)
    ops.gemm_nt(Ad, Bd, out, bias=dev(bias), act=L.ACT_SWISH, out2=pre)
    a = acc + bias
    check("pipe_swish_pre", pre, a, 4e-2)
    check("pipe_swish", out, a * torch.sigmoid(a), 4e-2)
    s_ = torch.sigmoid(rt(aux, dt))
    ops.gemm_nt(Ad, Bd, out, act=L.ACT_SWISH_GRAD, aux=dev(aux, dt))
    check("pipe_swish_grad", out, acc * s_ * (1 + rt(aux, dt) * (1 - s_)), 4e-2)


@pytest.mark.parametrize("tpb", ["1", "2", "3"])
@pytest.mark.parametrize("M,N,K", [(9664, 256, 768), (9664, 256, 1024), (9664, 256, 512), (128, 256, 768), (2112, 192, 512)])
def test_gemm_pipelined_long_k(M, N, K, tpb, gemm_options):
    """K = 512 / 768 / 1024 through the K-generic pipelined kernel (LIDK_GEMM_PIPEK = tiles per workgroup): plain and bias
    epilogues, one tile per workgroup when the launch is small, clamped duplicate tiles at the end of the range."""
    gemm_options("LIDK_GEMM_PIPEK", int(tpb))
    dt = torch.bfloat16
    A = torch.randn(M, K, generator=g(34)) * 0.3
    B = torch.randn(N, K, generator=g(35)) * 0.3
    bias = torch.randn(N, generator=g(36))
    acc = _gemm_ref(A, B, dt).float()
    Ad, Bd = dev(A, dt), dev(B, dt)
    out = torch.full((M, N), 5.0, device=DEV, dtype=dt)
    ops.gemm_nt(Ad, Bd, out)
    check("pipek_plain", out, acc, 2e-2, rtol=1e-2)
    out.fill_(5.0)
    ops.gemm_nt(Ad, Bd, out, bias=dev(bias))
    check("pipek_bias", out, acc + bias, 2e-2, rtol=1e-2)
    gemm_options("LIDK_GEMM_PIPEK", 0)
    ref = torch.empty_like(out)
    ops.gemm_nt(Ad, Bd, ref, bias=dev(bias))
    assert float((out.float() - ref.float()).abs().max()) <= 2e-2 * float(ref.float().abs().max())


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N1,N2,splitk", [(9664, 1024, 256, 8), (9664, 256, 1024, 8), (204, 256, 64, 4), (1000, 41, 256, 16),
                                            (333, 80, 240, 3), (4100, 768, 256, 5), (130, 64, 64, 1), (2048, 256, 256, 16),
                                            (9536, 768, 3072, 4), (9536, 2304, 768, 4), (9536, 768, 768, 4), (1088, 512, 1536, 2)])
def test_gemm_tn_wgrad(dt, M, N1, N2, splitk):
    """dW = dY^T . X from the row-major activations (no transposes), bias gradient fused as a column sum.  bf16 shapes with
    N1 % 128 == N2 % 128 == M % 64 == 0 and >= 16 128-tiles run the LDS-DMA ring (one, several and > 256 items per launch:
    the transformer backbones' d = 768 / ffn 3072 gradients and the conv extractor's)."""
    N1p = (N1 + 7) // 8 * 8
    X = torch.zeros(M, N1p)
    X[:, :N1] = torch.randn(M, N1, generator=g(90)) * (0.5 + torch.arange(N1) / N1)
    Y = torch.randn(M, N2, generator=g(91)) + 0.2 * torch.arange(N2) / N2
    base, cbase = torch.randn(N1, N2, generator=g(92)), torch.randn(N1, generator=g(93))
    ref = base.double() + 0.5 * rt(X[:, :N1], dt).double().t() @ rt(Y, dt).double()
    cref = cbase.double() + 0.5 * rt(X[:, :N1], dt).double().sum(0)
    C, cs = dev(base.clone()), dev(cbase.clone())
    ops.gemm_tn(dev(X, dt), dev(Y, dt), C, colsum=cs, alpha=0.5, splitk=splitk, N1=N1)
    check(f"gemm_tn_{M}x{N1}x{N2}", C, ref.float(), 3e-4 * float(ref.abs().max()))
    check(f"gemm_tn_colsum_{M}x{N1}", cs, cref.float(), 3e-4 * float(cref.abs().max()))


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, emb, B, T, H, dh):
    inner = H * dh
    q, k, v = qkv.split(inner, dim=-1)
    q, k, v = (t.reshape(B, T, H, dh).transpose(1, 2) for t in (q, k, v))
    scale = dh ** -0.5
    dots = q @ k.transpose(-1, -2) * scale + oc.rel_pos_scores(q, emb, scale)
    p = dots.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(B * T, inner), p


@pytest.mark.parametrize("path", ["v1", "mfma"])
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,T,H,dh,maxpos", [(2, 51, 4, 16, 512), (2, 151, 4, 64, 512), (1, 70, 8, 32, 512),
                                             (2, 51, 8, 8, 512), (2, 60, 2, 16, 20), (2, 100, 2, 64, 30), (3, 33, 2, 32, 512),
                                             (1, 200, 2, 64, 512)])     # T = 200: backward column kernel stages its two phases in turn
def test_attention_fwd_bwd(dt, B, T, H, dh, maxpos, path):
    if path == "mfma" and (dt != torch.bfloat16 or dh not in (32, 64)):
        pytest.skip("MFMA attention kernels are bf16, dh in {32, 64}")
    if path == "v1" and dt == torch.float32 and T * dh > 10000:
        pytest.skip("the f32 (parity-mode) kernels keep K, V and the rel-pos slice in LDS as f32: T = 200 x dh = 64 does not fit")
    inner = H * dh
    qkv = (0.7 * torch.randn(B * T, 3 * inner, generator=g(20)))
    emb = (0.5 * torch.randn(2 * maxpos + 1, dh, generator=g(21)))
    qkv_r = rt(qkv, dt).requires_grad_()
    emb_r = (rt(emb, dt) if dt == torch.bfloat16 else emb.clone()).requires_grad_()   # kernel rounds E to T in LDS
    out_ref, p_ref = _attn_ref(qkv_r, emb_r, B, T, H, dh)
    dout = torch.randn(B * T, inner, generator=g(22))
    out_ref.backward(rt(dout, dt))
    qd, ed = dev(qkv, dt), dev(emb)
    embT = dev(emb, dt) if path == "mfma" else None
    ldp = ops.attn_ldp(T, dh, dt) if path == "mfma" else T
    if path == "mfma":
        assert ldp == (T + 31) // 32 * 32
    out = torch.empty(B * T, inner, device=DEV, dtype=dt)
    probs = torch.zeros(B, H, T, ldp, device=DEV, dtype=dt)
    ops.attn_fwd(qd, ed, out, probs, B, T, H, dh, rel_emb_T=embT)
    check(f"attn_probs[{path}]", probs[..., :T], p_ref, tol(dt, 2e-6, 4e-3))
    if ldp > T:
        assert float(probs[..., T:].float().abs().max()) == 0.0          # padded keys get exactly zero probability
    check(f"attn_out[{path}]", out, out_ref, tol(dt, 1e-5, 1.5e-2))
    dqkv = torch.zeros(B * T, 3 * inner, device=DEV, dtype=dt)
    demb = torch.zeros_like(ed)
    dsc = torch.empty(B, H, T, (T + 31) // 32 * 32, device=DEV)
    ops.attn_bwd(qd, ed, probs, dev(dout, dt), dqkv, demb, dsc, B, T, H, dh, rel_emb_T=embT)
    gs = float(qkv_r.grad.abs().max())
    check(f"attn_dqkv[{path}]", dqkv, qkv_r.grad, tol(dt, 2e-5 * max(1, gs), 3e-2 * max(1, gs)))
    check(f"attn_demb[{path}]", demb, emb_r.grad, tol(dt, 1e-4, 6e-2) * max(1.0, float(emb_r.grad.abs().max())))
    if path == "mfma" and ops.attn_bwd_relpos_supported(T, dh, dt):
        # split form: dqkv now, the embedding gradient later from the stored dS rows (the engine defers it)
        dqkv2 = torch.zeros_like(dqkv)
        demb2 = torch.zeros_like(ed)
        ops.attn_bwd(qd, ed, probs, dev(dout, dt), dqkv2, None, dsc, B, T, H, dh, rel_emb_T=embT)
        assert float(demb2.abs().max()) == 0.0 and torch.equal(dqkv2, dqkv)
        ops.attn_bwd_relpos(qd, dsc, probs.shape[-1], demb2, B, T, H, dh)
        check("attn_demb_split", demb2, demb, 1e-4 * max(1.0, float(demb.abs().max())))      # float atomics: order only
    if path == "mfma" and ops.attn_recompute_supported(T, dh, dt):
        # recompute path: the forward stores no probabilities, the backward rebuilds them from q, k and the embeddings
        out3 = torch.empty_like(out)
        ops.attn_fwd(qd, ed, out3, None, B, T, H, dh, rel_emb_T=embT)
        assert torch.equal(out3, out)
        for split in (False, True):
            dqkv3 = torch.full_like(dqkv, 9.0)
            demb3 = torch.zeros_like(ed)
            ops.attn_bwd(qd, ed, None, dev(dout, dt), dqkv3, None if split else demb3, dsc, B, T, H, dh, rel_emb_T=embT)
            if split:
                ops.attn_bwd_relpos(qd, dsc, ldp, demb3, B, T, H, dh)
            check(f"attn_dqkv[recompute split={split}]", dqkv3, qkv_r.grad, 3e-2 * max(1, gs))
            check(f"attn_demb[recompute split={split}]", demb3, emb_r.grad, 6e-2 * max(1.0, float(emb_r.grad.abs().max())))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,T,H,dh,maxpos", [(1, 600, 4, 64, 512), (1, 835, 2, 64, 512), (2, 420, 8, 32, 512), (1, 1100, 1, 64, 512),
                                             (2, 301, 4, 64, 512), (1, 257, 8, 32, 512)])
def test_attention_long_sequences(dt, B, T, H, dh, maxpos):
    """Sequences beyond the LDS-resident kernels (conf max_duration 12 s -> T = 600, validation 16.7 s -> T = 835;
    T = 1100 also exceeds the relative-position table, so offsets clamp at +-512): the key-tiled kernels against torch."""
    inner = H * dh
    qkv = (0.7 * torch.randn(B * T, 3 * inner, generator=g(30)))
    emb = (0.5 * torch.randn(2 * maxpos + 1, dh, generator=g(31)))
    qkv_r = rt(qkv, dt).requires_grad_()
    emb_r = (rt(emb, dt) if dt == torch.bfloat16 else emb.clone()).requires_grad_()
    out_ref, p_ref = _attn_ref(qkv_r, emb_r, B, T, H, dh)
    dout = torch.randn(B * T, inner, generator=g(32))
    out_ref.backward(rt(dout, dt))
    assert T <= ops.attn_max_frames(dh, dt)
    qd, ed = dev(qkv, dt), dev(emb)
    ldp = ops.attn_ldp(T, dh, dt)
    assert ldp == T                                             # not the MFMA layout
    out = torch.empty(B * T, inner, device=DEV, dtype=dt)
    probs = torch.zeros(B, H, T, ldp, device=DEV, dtype=dt)
    ops.attn_fwd(qd, ed, out, probs, B, T, H, dh, rel_emb_T=dev(emb, dt))
    check("attn_long_probs", probs, p_ref, tol(dt, 2e-6, 4e-3))
    check("attn_long_out", out, out_ref, tol(dt, 1e-5, 1.5e-2))
    dqkv = torch.zeros(B * T, 3 * inner, device=DEV, dtype=dt)
    demb = torch.zeros_like(ed)
    dsc = torch.empty(B, H, T, (T + 31) // 32 * 32, device=DEV)
    ops.attn_bwd(qd, ed, probs, dev(dout, dt), dqkv, demb, dsc, B, T, H, dh, rel_emb_T=dev(emb, dt))
    gs = float(qkv_r.grad.abs().max())
    check("attn_long_dqkv", dqkv, qkv_r.grad, tol(dt, 2e-5 * max(1, gs), 3e-2 * max(1, gs)))
    check("attn_long_demb", demb, emb_r.grad, tol(dt, 2e-4, 6e-2) * max(1.0, float(emb_r.grad.abs().max())))


def test_attention_refuses_sequences_beyond_the_documented_limit():
    T = ops.attn_max_frames(64, torch.bfloat16) + 1
    qkv = torch.zeros(T, 3 * 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(LidkError):
        ops.attn_fwd(qkv, torch.zeros(1025, 64, device=DEV), torch.empty(T, 64, device=DEV, dtype=torch.bfloat16),
                     torch.empty(1, 1, T, T, device=DEV, dtype=torch.bfloat16), 1, T, 1, 64)


def test_tr16_hardware_mapping():
    """Pins the ds_read_b64_tr_b16 lane mapping (MI355X guide, T10) that the transposed-operand kernels are written against."""
    tile = (torch.arange(8)[:, None] * 100 + torch.arange(64)[None, :]).to(torch.int16)
    out = torch.zeros(64, 2, 4, dtype=torch.int16, device=DEV)
    ops.selftest_tr16(dev(tile), out)
    got = out.cpu()
    for lane in range(64):
        gidx, idx = lane >> 4, lane & 15
        for half in range(2):
            want = [int(tile[q + 4 * half, 16 * gidx + idx]) for q in range(4)]
            assert got[lane, half].tolist() == want, (lane, half, got[lane, half].tolist(), want)


# ------------------------------------------------------------------------------------------------ conv module
@pytest.mark.parametrize("dt", DT)
def test_glu(dt):
    M, C = 203, 128
    y = torch.randn(M, 2 * C, generator=g(30))
    yr = rt(y, dt).requires_grad_()
    gl = yr[:, :C] * torch.sigmoid(yr[:, C:])
    dg = torch.randn(M, C, generator=g(31))
    gl.backward(rt(dg, dt))
    yd = dev(y, dt)
    go = torch.empty(M, C, device=DEV, dtype=dt)
    ops.glu_fwd(yd, go)
    check("glu_fwd", go, gl, tol(dt, 1e-6, 2e-2))
    dy = torch.empty(M, 2 * C, device=DEV, dtype=dt)
    ops.glu_bwd(yd, dev(dg, dt), dy)
    check("glu_bwd", dy, yr.grad, tol(dt, 1e-6, 3e-2))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,T,C,K", [(3, 51, 128, 31), (2, 151, 512, 31), (2, 40, 96, 7), (1, 33, 64, 4)])
def test_dwconv_fwd_bwd_and_stats(dt, B, T, C, K):
    pad = (K // 2, K // 2 - (K + 1) % 2)
    x = torch.randn(B, T, C, generator=g(32))
    w = (0.3 * torch.randn(C, K, generator=g(33))).requires_grad_()
    b = (0.1 * torch.randn(C, generator=g(34))).requires_grad_()
    xr = rt(x, dt).requires_grad_()
    y = F.conv1d(F.pad(xr.transpose(1, 2), pad), w[:, None, :], b, groups=C).transpose(1, 2)   # (B,T,C)
    dy = torch.randn(B, T, C, generator=g(35))
    y.backward(rt(dy, dt))
    xd, wd, bd = dev(x, dt), dev(w.detach()), dev(b.detach())
    c = torch.empty(B, T, C, device=DEV, dtype=dt)
    nparts = ops.dwconv_stat_parts(B, T, C, dt)
    stat = torch.empty(nparts, 2, C, device=DEV)
    ops.dwconv_fwd(xd, wd, bd, c, stat, B, T, pad[0])
    check("dwconv_fwd", c, y, tol(dt, 2e-5, 3e-2))
    sums = torch.empty(2 * C, device=DEV, dtype=torch.float64)
    ops.reduce_partials_f64(stat, nparts, 2 * C, sums)
    yf = y.detach().reshape(-1, C).double()
    check("dwconv_stat_sum", sums[:C], yf.sum(0), 2e-3)
    check("dwconv_stat_sumsq", sums[C:], (yf * yf).sum(0), 2e-3, 1e-5)
    dg = torch.empty(B, T, C, device=DEV, dtype=dt)
    ops.dwconv_bwd_input(dev(dy, dt), wd, dg, B, T, pad[0])
    check("dwconv_dgrad", dg, xr.grad, tol(dt, 2e-5, 3e-2))
    dw = torch.zeros(C, K, device=DEV)
    db = torch.zeros(C, device=DEV)
    partial = torch.empty(B * C * (K + 1), device=DEV)
    ops.dwconv_bwd_weight(dev(dy, dt), xd, dw, db, partial, B, T, pad[0])
    check("dwconv_wgrad", dw, w.grad, 2e-4, 1e-4)
    check("dwconv_bgrad", db, b.grad, 2e-4, 1e-4)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,T,C,K", [(3, 70, 128, 31), (2, 151, 256, 31), (2, 40, 68, 7)])
def test_glu_dwconv_fused_matches_separate_kernels(dt, B, T, C, K):
    """GLU fused in front of the depthwise conv (forward) and behind its input gradient (backward) against the two-kernel
    sequences: the forward tile holds the same T-rounded GLU output, so c, g and the BatchNorm sums are bit-identical; the
    backward skips the T rounding of dg, so it is compared with the tolerance of one bf16 rounding."""
    M, pad = B * T, K // 2
    y = torch.randn(M, 2 * C, generator=g(130))
    w = 0.3 * torch.randn(C, K, generator=g(131))
    b = 0.1 * torch.randn(C, generator=g(132))
    dc = torch.randn(M, C, generator=g(133))
    yd, wd, bd, dcd = dev(y, dt), dev(w), dev(b), dev(dc, dt)
    nparts = ops.dwconv_stat_parts(B, T, C, dt)
    g_sep, c_sep = torch.empty(M, C, device=DEV, dtype=dt), torch.empty(M, C, device=DEV, dtype=dt)
    st_sep = torch.empty(nparts, 2, C, device=DEV)
    ops.glu_fwd(yd, g_sep)
    ops.dwconv_fwd(g_sep, wd, bd, c_sep, st_sep, B, T, pad)
    g_fus, c_fus = torch.full((M, C), 3.0, device=DEV, dtype=dt), torch.full((M, C), 3.0, device=DEV, dtype=dt)
    st_fus = torch.empty(nparts, 2, C, device=DEV)
    ops.glu_dwconv_fwd(yd, wd, bd, g_fus, c_fus, st_fus, B, T, pad)
    assert torch.equal(g_fus, g_sep) and torch.equal(c_fus, c_sep) and torch.equal(st_fus, st_sep)
    c_nog = torch.full((M, C), 3.0, device=DEV, dtype=dt)
    ops.glu_dwconv_fwd(yd, wd, bd, None, c_nog, None, B, T, pad)           # evaluation form: no g, no statistics
    assert torch.equal(c_nog, c_sep)
    dg = torch.empty(M, C, device=DEV, dtype=torch.float32)
    ops.dwconv_bwd_input(dev(dc, torch.float32) if dt == torch.float32 else dcd.float(), wd, dg, B, T, pad)
    dy_ref = torch.empty(M, 2 * C, device=DEV, dtype=torch.float32)
    ops.glu_bwd(yd.float(), dg, dy_ref)
    dy = torch.full((M, 2 * C), 3.0, device=DEV, dtype=dt)
    ops.dwconv_bwd_input_glu(dcd, wd, yd, dy, B, T, pad)
    check("glu_dwconv_bwd", dy, dy_ref, tol(dt, 2e-5, 3e-2))
    # ... and with the BatchNorm+Swish backward in front (dc formed on the fly from ds, c and the batch sums)
    cbuf = 1.5 * torch.randn(M, C, generator=g(134)) + 0.3
    ds = torch.randn(M, C, generator=g(135))
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g(136)), 0.1 * torch.randn(C, generator=g(137))
    cd, dsd, gd, btd = dev(cbuf, dt), dev(ds, dt), dev(gamma), dev(beta)
    cr = rt(cbuf, dt).double()
    mean, var = cr.mean(0), cr.var(0, unbiased=False)
    md, rd = dev(mean.float()), dev((1.0 / torch.sqrt(var + 1e-5)).float())
    partial = torch.empty(L.BN_PARTIAL_BLOCKS * 2 * C, device=DEV)
    ops.bn_swish_bwd_reduce(dsd, cd, md, rd, gd, btd, partial)
    sums = torch.empty(2 * C, device=DEV, dtype=torch.float64)
    ops.reduce_partials_f64(partial, L.BN_PARTIAL_BLOCKS, 2 * C, sums)
    dc32 = torch.empty(M, C, device=DEV, dtype=torch.float32)
    dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.bn_swish_bwd_apply(dsd.float(), cd.float(), md, rd, gd, btd, sums, sums, M, dc32, dgam, dbet)
    ops.dwconv_bwd_input(dc32, wd, dg, B, T, pad)
    ops.glu_bwd(yd.float(), dg, dy_ref)
    dy.fill_(3.0)
    ops.dwconv_bwd_input_bn_glu(dsd, cd, md, rd, gd, btd, sums, M, wd, yd, dy, B, T, pad)
    check("bn_glu_dwconv_bwd", dy, dy_ref, tol(dt, 5e-5, 3e-2), tol(dt, 1e-5, 1e-2))


@pytest.mark.parametrize("dt", DT)
def test_batchnorm_swish_train_and_eval(dt):
    M, C = 408, 128
    c = 1.5 * torch.randn(M, C, generator=g(36)) + 0.3
    gamma = (1 + 0.2 * torch.randn(C, generator=g(37))).requires_grad_()
    beta = (0.1 * torch.randn(C, generator=g(38))).requires_grad_()
    rm, rv = 0.1 * torch.randn(C, generator=g(39)), 0.5 + torch.rand(C, generator=g(40))
    cr = rt(c, dt).requires_grad_()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.batch_norm(cr, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    s = z * torch.sigmoid(z)
    ds = torch.randn(M, C, generator=g(41))
    s.backward(rt(ds, dt))
    cd = dev(c, dt)
    sums = torch.stack([cr.detach().double().sum(0), (cr.detach().double() ** 2).sum(0)]).reshape(-1).to(DEV)
    mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    rmd, rvd = dev(rm), dev(rv)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    ops.bn_train_stats(sums, M, mean, rstd, rmd, rvd, nbt)
    check("bn_running_mean", rmd, rm_ref, 1e-6)
    check("bn_running_var", rvd, rv_ref, 1e-5)
    assert int(nbt) == 1
    so = torch.empty(M, C, device=DEV, dtype=dt)
    gd, bd = dev(gamma.detach()), dev(beta.detach())
    ops.bn_swish_fwd(cd, mean, rstd, gd, bd, so)
    check("bn_swish_fwd", so, s, tol(dt, 2e-5, 3e-2))
    partial = torch.empty(L.BN_PARTIAL_BLOCKS * 2 * C, device=DEV)
    ops.bn_swish_bwd_reduce(dev(ds, dt), cd, mean, rstd, gd, bd, partial)
    bs = torch.empty(2 * C, device=DEV, dtype=torch.float64)
    ops.reduce_partials_f64(partial, L.BN_PARTIAL_BLOCKS, 2 * C, bs)
    dc = torch.empty(M, C, device=DEV, dtype=dt)
    dgam, dbet = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.bn_swish_bwd_apply(dev(ds, dt), cd, mean, rstd, gd, bd, bs, bs, M, dc, dgam, dbet)
    check("bn_bwd_dc", dc, cr.grad, tol(dt, 2e-5, 2e-2))
    check("bn_bwd_dgamma", dgam, gamma.grad, 1e-3, 1e-4)
    check("bn_bwd_dbeta", dbet, beta.grad, 1e-3, 1e-4)
    # eval-mode statistics
    ops.bn_eval_stats(dev(rm), dev(rv), mean, rstd)
    ops.bn_swish_fwd(cd, mean, rstd, gd, bd, so)
    ze = F.batch_norm(rt(c, dt), rm, rv, gamma.detach(), beta.detach(), False, 0.1, 1e-5)
    check("bn_swish_eval", so, ze * torch.sigmoid(ze), tol(dt, 2e-5, 3e-2))


@pytest.mark.parametrize("B,T,C,K", [(3, 151, 128, 31), (2, 37, 72, 15), (5, 200, 64, 31)])
def test_dwconv_wgrad_with_fused_bn_apply_equals_the_two_launch_sequence(B, T, C, K):
    """lidk_dwconv_bwd_weight_bn (dc formed in the operand load, never written) against lidk_bn_swish_bwd_apply followed by
    lidk_dwconv_bwd_weight: the same bf16-rounded dc values enter the same f32 sums, only grouped differently (80-row tiles of
    20 rows per wave instead of 32 / 8), so dw / db agree to f32 summation noise and dgamma / dbeta to the last bit; and
    against torch autograd within bf16 tolerance."""
    dt = torch.bfloat16
    M = B * T
    assert ops.dwconv_bwd_weight_bn_supported(C, dt)
    gin = torch.randn(M, C, generator=g(140))
    c = 1.2 * torch.randn(M, C, generator=g(141)) + 0.2
    ds = torch.randn(M, C, generator=g(142))
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g(143)), 0.1 * torch.randn(C, generator=g(144))
    gd, cd, dsd = dev(gin, dt), dev(c, dt), dev(ds, dt)
    cr = rt(c, dt)
    sums = torch.stack([cr.double().sum(0), (cr.double() ** 2).sum(0)]).reshape(-1).to(DEV)
    mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_train_stats(sums, M, mean, rstd, None, None, None)
    gmd, btd = dev(gamma), dev(beta)
    partial = torch.empty(L.BN_PARTIAL_BLOCKS * 2 * C, device=DEV)
    ops.bn_swish_bwd_reduce(dsd, cd, mean, rstd, gmd, btd, partial)
    bs = torch.zeros(2 * C + 1, device=DEV, dtype=torch.float64)
    ops.reduce_partials_f64(partial, L.BN_PARTIAL_BLOCKS, 2 * C, bs, tail=M)
    # two launches
    dc = torch.empty(M, C, device=DEV, dtype=dt)
    dgam0, dbet0 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    ops.bn_swish_bwd_apply(dsd, cd, mean, rstd, gmd, btd, bs, bs, 0, dc, dgam0, dbet0)
    dw0, db0 = torch.zeros(C, K, device=DEV), torch.zeros(C, device=DEV)
    wp = torch.empty(B * C * (K + 1), device=DEV)
    ops.dwconv_bwd_weight(dc, gd, dw0, db0, wp, B, T, K // 2)
    # fused
    dgam1, dbet1 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dw1, db1 = torch.zeros(C, K, device=DEV), torch.zeros(C, device=DEV)
    ops.dwconv_bwd_weight_bn(dsd, cd, mean, rstd, gmd, btd, bs, bs, 0, gd, dw1, db1, dgam1, dbet1, wp, B, T, K // 2)
    assert torch.equal(dgam0, dgam1) and torch.equal(dbet0, dbet1)
    scale = float(dw0.abs().max())
    assert float((dw0 - dw1).abs().max()) <= 2e-6 * scale and float((db0 - db1).abs().max()) <= 2e-6 * float(db0.abs().max() + 1)
    # torch autograd of conv -> BatchNorm(train) -> Swish with respect to the conv weight, on the bf16-rounded operands
    w = (0.2 * torch.randn(C, K, generator=g(145))).requires_grad_()
    xg = rt(gin, dt).view(B, T, C)
    conv = F.conv1d(F.pad(xg.transpose(1, 2), (K // 2, K - 1 - K // 2)), w[:, None, :], None, groups=C).transpose(1, 2).reshape(M, C)
    # the kernel differentiates at the GIVEN c (the forward's bf16 conv output), so substitute it without cutting the graph
    cc = conv + (cr - conv).detach()
    z = F.batch_norm(cc, None, None, gamma, beta, True, 0.1, 1e-5)
    (z * torch.sigmoid(z)).backward(rt(ds, dt))
    check("dwconv_wgrad_bn_dw", dw1, w.grad, 0.15, 2e-2)        # dc is rounded to bf16 (rel 4e-3) before ~T*B products are summed


@pytest.mark.parametrize("mode", ["bf16_plain", "f32_bias_res", "gelu_pre"])
def test_gemm_nt_long_k_large_shape_uses_the_128_tile(mode):
    """K >= 2048 with >= 256 tiles of 128x128 dispatches gemm_nt_bf16_big_kernel (WavLM's ffn-down shape class); M is not a
    multiple of 128 so the clamped last row tile is covered.  Reference: f32 matmul of the bf16-rounded operands."""
    M, N, K = 8192 + 77, 512, 2048
    A = (torch.randn(M, K, generator=g(160)) * 0.5).bfloat16()
    B = (torch.randn(N, K, generator=g(161)) / K ** 0.5).bfloat16()
    bias = torch.randn(N, generator=g(162))
    ref = A.float() @ B.float().t()
    Ad, Bd = A.to(DEV), B.to(DEV)
    if mode == "bf16_plain":
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(Ad, Bd, out)
        check("gemm_big_bf16", out, ref, 3e-2, 1e-2)
    elif mode == "f32_bias_res":
        res = torch.randn(M, N, generator=g(163))
        out = torch.empty(M, N, device=DEV)
        ops.gemm_nt(Ad, Bd, out, bias=bias.to(DEV), alpha=0.5, res=res.to(DEV))
        check("gemm_big_f32", out, 0.5 * (ref + bias) + res, 2e-3, 1e-4)
    else:
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        pre = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(Ad, Bd, out, bias=bias.to(DEV), act=L.ACT_GELU, out2=pre)
        check("gemm_big_pre", pre, ref + bias, 3e-2, 1e-2)
        check("gemm_big_gelu", out, F.gelu(ref + bias), 3e-2, 1e-2)


@pytest.mark.parametrize("mode", ["bf16_plain", "f32_bias_res", "gelu_pre", "strided_view"])
def test_gemm_nt_lds_dma_128_tile(mode, gemm_options):
    """With LIDK_GEMM_DMA=512 (opt-in through lidk_gemm_option), K >= 512, N % 128 == 0 and >= 384 tiles of 128x128 dispatch
    gemm_nt_bf16_dma_kernel (operands by global_load_lds, the transformer
    backbones' d = 768 shapes); M is not a multiple of 128 (clamped last row tile); 'strided_view': overlapping rows
    (lda < K), the conv feature extractor's operand form.  Reference: f32 matmul of the bf16-rounded operands."""
    gemm_options("LIDK_GEMM_DMA", 512)
    M, N, K = 9536 + 77, 768, 768
    A = (torch.randn(M, K, generator=g(170)) * 0.5).bfloat16()
    B = (torch.randn(N, K, generator=g(171)) / K ** 0.5).bfloat16()
    bias = torch.randn(N, generator=g(172))
    Ad, Bd = A.to(DEV), B.to(DEV)
    if mode == "strided_view":
        x = (torch.randn(2 * M + 8, 256, generator=g(173)) * 0.5).bfloat16()
        Ad = x.to(DEV).as_strided((M, K), (512, 1))                 # row t = 768 contiguous values from input row 2 t
        A = x.as_strided((M, K), (512, 1))
    ref = A.float() @ B.float().t()
    if mode in ("bf16_plain", "strided_view"):
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(Ad, Bd, out)
        check("gemm_dma_bf16", out, ref, 3e-2, 1e-2)
    elif mode == "f32_bias_res":
        res = torch.randn(M, N, generator=g(174))
        out = torch.empty(M, N, device=DEV)
        ops.gemm_nt(Ad, Bd, out, bias=bias.to(DEV), alpha=0.5, res=res.to(DEV))
        check("gemm_dma_f32", out, 0.5 * (ref + bias) + res, 2e-3, 1e-4)
    else:
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        pre = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(Ad, Bd, out, bias=bias.to(DEV), act=L.ACT_GELU, out2=pre)
        check("gemm_dma_pre", pre, ref + bias, 3e-2, 1e-2)
        check("gemm_dma_gelu", out, F.gelu(ref + bias), 3e-2, 1e-2)


@pytest.mark.parametrize("mode", ["bf16_plain", "f32_bias_res", "gelu_pre", "gelu_grad", "strided_view"])
@pytest.mark.parametrize("N,K,bn", [(2304, 768, 256), (1024, 4096, 256), (512, 1536, 256), (2304, 768, 128), (640, 1024, 128)])
def test_gemm_nt_lds_dma_256_tile(mode, N, K, bn, gemm_options):
    """N % 256 == 0, K % 64 == 0, K >= 512 and >= LIDK_GEMM_DMA256 (here forced to 1; default 200) tiles of 256 x 256 dispatch
    gemm_nt_bf16_dma256_kernel: the transformer backbones' QKV / FFN / conv-stack shapes.  M = 4 x 256 + 77: a clamped last row tile.
    Every epilogue the backbones use; 'strided_view': overlapping rows (lda < K), the conv feature extractor's operand form.
    Reference: f32 matmul of the bf16-rounded operands."""
    gemm_options("LIDK_GEMM_DMA256", 1)
    gemm_options("LIDK_GEMM_DMA256_BN", bn)                      # 256 x 256 tiles (waves 2 x 4) or 256 x 128 (waves 4 x 2)
    M = 4 * 256 + 77
    A = (torch.randn(M, K, generator=g(180)) * 0.5).bfloat16()
    B = (torch.randn(N, K, generator=g(181)) / K ** 0.5).bfloat16()
    bias = torch.randn(N, generator=g(182))
    Ad, Bd = A.to(DEV), B.to(DEV)
    if mode == "strided_view":
        x = (torch.randn(2 * M + 8 * K // 256, 256, generator=g(183)) * 0.5).bfloat16()
        Ad = x.to(DEV).as_strided((M, K), (512, 1))                 # row t = K contiguous values from input row 2 t
        A = x.as_strided((M, K), (512, 1))
    ref = A.float() @ B.float().t()
    if mode in ("bf16_plain", "strided_view"):
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(Ad, Bd, out)
        check("gemm_dma256_bf16", out, ref, 3e-2, 1e-2)
    elif mode == "f32_bias_res":
        res = torch.randn(M, N, generator=g(184))
        out = torch.empty(M, N, device=DEV)
        ops.gemm_nt(Ad, Bd, out, bias=bias.to(DEV), alpha=0.5, res=res.to(DEV))
        check("gemm_dma256_f32", out, 0.5 * (ref + bias) + res, 2e-3, 1e-4)
    elif mode == "gelu_grad":
        aux = torch.randn(M, N, generator=g(185)).bfloat16()
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(Ad, Bd, out, act=L.ACT_GELU_GRAD, aux=aux.to(DEV))
        a = aux.float().requires_grad_()
        F.gelu(a).sum().backward()
        check("gemm_dma256_gelu_grad", out, ref * a.grad, 3e-2, 1e-2)
    else:
        out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        pre = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(Ad, Bd, out, bias=bias.to(DEV), act=L.ACT_GELU, out2=pre)
        check("gemm_dma256_pre", pre, ref + bias, 3e-2, 1e-2)
        check("gemm_dma256_gelu", out, F.gelu(ref + bias), 3e-2, 1e-2)


@pytest.mark.parametrize("M,C", [(9664, 256), (453, 144), (7, 64)])
def test_double_layernorm_equals_two_single_launches(M, C):
    """lidk_layernorm2_fwd / _bwd (post_norm of block i + the first PreNorm of block i + 1 in one pass) against two
    lidk_layernorm_fwd / _bwd launches: same formulas on the same f32 intermediates -> outputs agree to f32 rounding, the bf16
    operand to one ulp on a vanishing fraction of elements."""
    dt = torch.bfloat16
    x = dev(1.5 * torch.randn(M, C, generator=g(150)) + 0.2)
    g1, b1 = dev(1 + 0.1 * torch.randn(C, generator=g(151))), dev(0.1 * torch.randn(C, generator=g(152)))
    g2, b2 = dev(1 + 0.1 * torch.randn(C, generator=g(153))), dev(0.1 * torch.randn(C, generator=g(154)))
    f = lambda *sh: torch.empty(*sh, device=DEV)
    y1a, m1a, r1a, y2a, m2a, r2a = f(M, C), f(M), f(M), torch.empty(M, C, device=DEV, dtype=dt), f(M), f(M)
    ops.layernorm_fwd(x, g1, b1, y32=y1a, mean=m1a, rstd=r1a, dtype=dt)
    ops.layernorm_fwd(y1a, g2, b2, yT=y2a, mean=m2a, rstd=r2a)
    y1b, m1b, r1b, y2b, m2b, r2b = f(M, C), f(M), f(M), torch.empty(M, C, device=DEV, dtype=dt), f(M), f(M)
    ops.layernorm2_fwd(x, g1, b1, y1b, m1b, r1b, g2, b2, y2b, m2b, r2b)
    assert float((y1a - y1b).abs().max()) <= 2e-6 * float(y1a.abs().max()) and float((m1a - m1b).abs().max()) <= 1e-6
    assert float(((r1a - r1b) / r1a).abs().max()) <= 2e-6 and float(((r2a - r2b) / r2a).abs().max()) <= 1e-5
    dh = (y2a.float() - y2b.float()).abs()
    assert float(dh.max()) <= 2.0 ** -7 * float(y2a.float().abs().max()) and float((dh > 0).float().mean()) < 2e-3
    # backward
    dy = dev(torch.randn(M, C, generator=g(155)), dt)
    dres = dev(torch.randn(M, C, generator=g(156)))
    part = f(L.LN_BWD_BLOCKS * 2 * C)
    dv, dxa, dxTa = f(M, C), f(M, C), torch.empty(M, C, device=DEV, dtype=dt)
    dg2a, db2a, dg1a, db1a = [torch.zeros(C, device=DEV) for _ in range(4)]
    ops.layernorm_bwd(dy, y1a, m2a, r2a, g2, part, dres=dres, dx=dv, dgamma=dg2a, dbeta=db2a, dtype=dt)
    ops.layernorm_bwd(dv, x, m1a, r1a, g1, part, dx=dxa, dxT=dxTa, dxT_scale=0.5, dgamma=dg1a, dbeta=db1a, dtype=dt)
    p1, p2 = f(L.LN_BWD_BLOCKS * 2 * C), f(L.LN_BWD_BLOCKS * 2 * C)
    dxb, dxTb = f(M, C), torch.empty(M, C, device=DEV, dtype=dt)
    ops.layernorm2_bwd(dy, dres, y1a, m2a, r2a, g2, x, m1a, r1a, g1, dxb, dxTb, 0.5, p1, p2)
    dg2b, db2b, dg1b, db1b = [torch.zeros(C, device=DEV) for _ in range(4)]
    ops.layernorm_param_grads(p1, M, C, dg1b, db1b)
    ops.layernorm_param_grads(p2, M, C, dg2b, db2b)
    scale = float(dxa.abs().max())
    assert float((dxa - dxb).abs().max()) <= 1e-5 * scale
    assert float((dxTa.float() - dxTb.float()).abs().max()) <= 2.0 ** -7 * 0.5 * scale
    for a, b, n in ((dg1a, dg1b, "dg1"), (db1a, db1b, "db1"), (dg2a, dg2b, "dg2"), (db2a, db2b, "db2")):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max())), n


# ------------------------------------------------------------------------------------------------ front-end
def test_normalize_dither_preemph():
    w = 3.0 + 2.0 * torch.randn(3, 16000, generator=g(50))
    out = ops.normalize_wav(dev(w))
    check("normalize_wav", out, of.normalize_wav(w), 2e-5)
    noise = torch.rand(3, 16000, generator=g(51))
    out = ops.dither_preemph(dev(w), noise=dev(noise))
    check("dither_preemph", out, of.dither_preemphasis(w, noise), 2e-6)
    out2 = ops.dither_preemph(dev(w), seed=5)
    check("dither_preemph_gen", out2, of.dither_preemphasis(w), 2.1e-5)        # differs only by the 1e-5*U dither


@pytest.mark.parametrize("ragged", [False, True])
def test_wav2mel_fused_equals_the_three_kernel_path_and_the_oracle(ragged):
    """lidk_wav2mel (normalise + dither + pre-emphasis inside the STFT's frame load) against normalize_wav -> dither_preemph ->
    logmel with the same explicit dither values, and against the oracle's composition; SpecAugment spans included."""
    B, L_, pad = 5, 48000, 16
    w = 2.0 + 1.5 * torch.randn(B, L_, generator=g(180)) * torch.linspace(0.1, 1.0, L_)
    noise = torch.rand(B, L_, generator=g(181))
    ns = torch.tensor([48000, 31000, 47999, 16000, 40123], dtype=torch.int32) if ragged else None
    if ragged:
        for i, n in enumerate(ns.tolist()):
            w[i, n:] = 0.0
    spans = torch.tensor([[[10, 25, 5, 17]]] * B, dtype=torch.int32)
    nsd = ns.to(DEV) if ragged else None
    x1 = ops.normalize_wav(dev(w), n_samples=nsd)
    x2 = ops.dither_preemph(x1, noise=dev(noise))
    ref3 = ops.logmel(x2, pad=pad, spans=spans.to(DEV), n_samples=nsd)
    got = ops.wav2mel(dev(w), pad=pad, spans=spans.to(DEV), n_samples=nsd, noise=dev(noise))
    assert got.shape == ref3.shape
    err = (got - ref3).abs()
    print(f"[wav2mel fused vs 3 kernels, ragged={ragged}] max {float(err.max()):.2e} dB, median {float(err.median()):.2e}")
    assert float(err.max()) <= 2e-3 and float(err.median()) <= 2e-5
    # oracle composition (per utterance when ragged)
    for i in range(B):
        n = int(ns[i]) if ragged else L_
        xo = of.dither_preemphasis(of.normalize_wav(w[i:i + 1, :n]), noise[i:i + 1, :n])
        mo = of.apply_specaug(of.wav2mel(xo, pad=pad)[0], [tuple(spans[i, 0].tolist())]).transpose(0, 1)
        e = (got[i, :mo.shape[0]].cpu() - mo).abs()
        assert float(e.max()) <= 5e-3, (i, float(e.max()))
        assert float(got[i, mo.shape[0]:].abs().max() if mo.shape[0] < got.shape[1] else 0.0) == 0.0


@pytest.mark.parametrize("L_,pad,B", [(16000, 0, 4), (48000, 16, 3), (4000, 0, 2)])
def test_logmel_matches_oracle(L_, pad, B):
    w = of.normalize_wav(torch.randn(B, L_, generator=g(52)) * torch.linspace(0.2, 1.0, L_))
    w[0] = torch.sin(2 * math.pi * 1000.0 * torch.arange(L_) / 16000.0)         # tonal utterance: deep floor
    ref = of.wav2mel(w, pad=pad)                                                 # (B, 80, F)
    out = ops.logmel(dev(w), pad=pad)
    assert out.shape == (B, ref.shape[-1], 80)
    # tolerance: 5e-3 dB absolute (0.1 % in power).  Both sides are f32 FFTs (LDS radix-2 vs torch's pocketfft/MKL);
    # bins 80 dB under a pure tone carry ~1e-7*|X|max of rounding noise on either side.
    check("logmel_db", out, ref.transpose(1, 2), 5e-3)
    err = (out.cpu() - ref.transpose(1, 2)).abs()
    assert float(err.median()) < 5e-5, float(err.median())
    spans = torch.tensor([[[3, 9, 10, 22], [30, 31, 0, 5]]] * B, dtype=torch.int32)
    out = ops.logmel(dev(w), pad=pad, spans=dev(spans))
    ref_aug = torch.stack([of.apply_specaug(ref[i], [tuple(s) for s in spans[i].tolist()]) for i in range(B)])
    check("logmel_specaug", out, ref_aug.transpose(1, 2), 5e-3)


@pytest.mark.parametrize("pad", [0, 16])
def test_ragged_batch_matches_the_reference_collate(pad):
    """A ragged, zero-padded waveform batch with its true lengths: every utterance must come out exactly as if it had been
    processed alone (own normalisation statistics, reflection at its own end, own top_db maximum, SpecAugment spans inside
    its own frames) and the rows behind its frames must be 0.0 - the zero-padded mel of lid/raw_datasets.py:345-365."""
    lens = [16000, 9000, 12345, 400]
    L_ = max(lens)
    raw = torch.zeros(len(lens), L_)
    for i, n in enumerate(lens):
        raw[i, :n] = 0.5 + (1.0 + i) * torch.randn(n, generator=g(56 + i))
    ns = torch.tensor(lens, dtype=torch.int32)
    norm = ops.normalize_wav(dev(raw), n_samples=dev(ns))
    spans = torch.tensor([[[3, 9, 10, 22]], [[5, 6, 0, 5]], [[0, 2, 70, 80]], [[1, 2, 3, 4]]], dtype=torch.int32)
    out = ops.logmel(norm, pad=pad, spans=dev(spans), n_samples=dev(ns))
    full = 1 + (L_ + 2 * pad) // 160
    assert out.shape == (len(lens), full, 80)
    for i, n in enumerate(lens):
        wi = of.normalize_wav(raw[i:i + 1, :n])
        check(f"ragged_normalize[{i}]", norm[i, :n], wi[0], 3e-5)
        assert float(norm[i, n:].abs().max()) == 0.0 if n < L_ else True
        ref = of.apply_specaug(of.wav2mel(wi, pad=pad)[0], [tuple(s) for s in spans[i].tolist()])       # (80, F_i)
        fi = ref.shape[-1]
        assert fi == 1 + (n + 2 * pad) // 160
        check(f"ragged_logmel[{i}]", out[i, :fi], ref.transpose(0, 1), 5e-3)
        if fi < full:
            assert float(out[i, fi:].abs().max()) == 0.0
    ref_b, pct = of.collate_mel([of.apply_specaug(of.wav2mel(of.normalize_wav(raw[i:i + 1, :n]), pad=pad)[0],
                                                  [tuple(s) for s in spans[i].tolist()]) for i, n in enumerate(lens)])
    check("ragged_collate", out, ref_b, 5e-3)


@pytest.mark.parametrize("dt", DT)
def test_im2col(dt):
    B, F_, C = 3, 101, 80
    T = (F_ + 2 - 3) // 2 + 1
    mel = torch.randn(B, F_, C, generator=g(53))
    out = torch.empty(B * T, 3 * C, device=DEV, dtype=dt)
    ops.im2col_k3s2(dev(mel), out, T)
    padded = F.pad(mel, (0, 0, 1, 1))
    ref = torch.stack([padded[:, 2 * t:2 * t + 3].reshape(B, 3 * C) for t in range(T)], 1).reshape(B * T, 3 * C)
    check("im2col", out, rt(ref, dt), 0)
    # and the GEMM against the permuted conv weight reproduces Conv1d(k3, s2, p1)
    w = 0.1 * torch.randn(C, C, 3, generator=g(54))
    bias = torch.randn(C, generator=g(55))
    conv = F.conv1d(rt(mel, dt).transpose(1, 2), rt(w, dt), bias, stride=2, padding=1).transpose(1, 2).reshape(B * T, C)
    w2 = w.permute(0, 2, 1).reshape(C, 3 * C)
    y = torch.empty(B * T, C, device=DEV, dtype=dt)
    ops.gemm_nt(out, dev(w2, dt), y, bias=dev(bias), act=L.ACT_RELU)
    check("subsample_conv", y, torch.relu(conv), tol(dt, 1e-4, 5e-2))


# ------------------------------------------------------------------------------------------------ loss
def _ctc_case(B, T, V, Lmax, seed, in_len, tg_len, force_repeat=True):
    logits = 2.0 * torch.randn(B, T, V + 1, generator=g(seed))
    targets = torch.randint(0, V, (B, Lmax), generator=g(seed + 1))
    if force_repeat:
        targets[0, 1] = targets[0, 0]
        targets[0, 3] = targets[0, 2]
    return logits, targets, torch.tensor(in_len), torch.tensor(tg_len)


@pytest.mark.parametrize("case", [
    dict(B=4, T=51, V=40, Lmax=12, in_len=[51, 45, 51, 40], tg_len=[12, 9, 12, 6]),
    dict(B=3, T=151, V=40, Lmax=20, in_len=[151, 151, 100], tg_len=[20, 20, 20]),
    dict(B=3, T=20, V=5, Lmax=12, in_len=[20, 8, 20], tg_len=[12, 12, 0]),        # utt 1 infeasible -> zero_infinity
    dict(B=2, T=300, V=4441, Lmax=60, in_len=[300, 280], tg_len=[60, 33]),        # large vocabulary (the reference's "cn")
    dict(B=2, T=400, V=40, Lmax=60, in_len=[400, 333], tg_len=[60, 41]),          # lattice > LDS: one-workgroup fallback kernel
    dict(B=3, T=20, V=5, Lmax=4, in_len=[20, 0, 0], tg_len=[4, 0, 3]),            # empty inputs
    dict(B=2, T=100, V=30, Lmax=40, in_len=[100, 90], tg_len=[40, 35]),           # S = 81 > one wave
])
def test_ctc_loss_and_grad(case):
    B, T, V, Lmax = case["B"], case["T"], case["V"], case["Lmax"]
    logits, targets, in_len, tg_len = _ctc_case(B, T, V, Lmax, 60, case["in_len"], case["tg_len"])
    lr = logits.clone().requires_grad_()
    per = F.ctc_loss(torch.log_softmax(lr, -1).transpose(0, 1), targets, in_len, tg_len, blank=V, reduction="none",
                     zero_infinity=True)
    per.mean().backward()
    loss = torch.empty(B, device=DEV)
    dl = torch.full((B, T, V + 1), 3.0, device=DEV)
    ws = torch.empty(ops.ctc_workspace_bytes(B, T, V + 1, Lmax) // 4, device=DEV)
    ops.ctc_loss(dev(logits), dev(targets), dev(in_len), dev(tg_len), loss, dl, ws, V, grad_scale=1.0 / B)
    check("ctc_loss", loss, per, 2e-4, 2e-5)
    # gradient = softmax - exp(log(alpha*beta) + nll - lp): the two O(|nll|) terms cancel, so f32 leaves an absolute
    # error of a few ulp(|nll|) on both sides (torch computes the same expression in f32)
    check("ctc_grad", dl, lr.grad, 2e-6 + 4e-7 * float(per.abs().max()), 1e-4)


def test_lid_score():
    cfg = oc.ModelCfg(lang2vocab={"a": 30, "b": 40, "c": 50}, lang2index={"a": 0, "b": 1, "c": 2})
    logits = {k: 3.0 * torch.randn(5, 51, v + 1, generator=g(70 + v)) for k, v in cfg.lang2vocab.items()}
    ref = oc.lang_scores(logits, cfg)
    scores = torch.zeros(5, 3, device=DEV)
    for k, v in cfg.lang2vocab.items():
        ops.lid_score(dev(logits[k]), scores[:, cfg.lang2index[k]:], 3, v)
    check("lid_score", scores, ref, 2e-5)
    assert torch.equal(scores.argmax(-1).cpu(), ref.argmax(-1))


# ------------------------------------------------------------------------------------------------ optimizer
def test_novograd_clip_and_cast():
    shapes = [(64, 48), (300,), (10, 7, 3), (9000,), (16, 16)]
    sizes = [int(np.prod(s)) for s in shapes]
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += (n + 7) // 8 * 8
    total = o
    gen = g(80)
    params = torch.zeros(total)
    for off, n in zip(offs, sizes):
        params[off:off + n] = torch.randn(n, generator=gen)
    pd, gd = dev(params), torch.zeros(total, device=DEV)
    md = torch.zeros(total, device=DEV)
    vd = torch.zeros(len(shapes), device=DEV)
    scratch = torch.empty(4096, device=DEV)
    tn = torch.empty(1, device=DEV)
    ref_p = [params[off:off + n].clone() for off, n in zip(offs, sizes)]
    states = [oo.NovogradState() for _ in shapes]
    for step, active in enumerate([[0, 1, 2, 3, 4], [0, 2, 3], [0, 1, 2, 3, 4]]):
        grads = [None] * len(shapes)
        gflat = torch.zeros(total)
        for t in active:
            grads[t] = (30.0 if step == 0 else 0.5) * torch.randn(sizes[t], generator=gen)   # step 0 exceeds max_norm=20
            gflat[offs[t]:offs[t] + sizes[t]] = grads[t]
        gd.copy_(gflat)
        work = []
        for t in active:
            for s in range(0, sizes[t], L.OPT_CHUNK):
                work.append([t, offs[t] + s, min(L.OPT_CHUNK, sizes[t] - s)])
        ops.novograd_step(pd, gd, md, vd, dev(torch.tensor(work, dtype=torch.int64)), len(shapes), 0.01, (0.95, 0.98),
                          1e-8, 1e-3, False, 20.0, scratch, tn)
        act_g = [grads[t] for t in active]
        total_norm = oo.clip_grad_norm(act_g, 20.0)
        oo.novograd_step(ref_p, grads, states, 0.01, weight_decay=1e-3)
        check(f"total_norm_{step}", tn, total_norm.reshape(1), 1e-3, 1e-5)
        got = pd.cpu()
        for t, (off, n) in enumerate(zip(offs, sizes)):
            check(f"novograd_p{t}_step{step}", got[off:off + n], ref_p[t], 2e-6, 1e-5)
    # cast_weights: W and W^T in the T arena
    for dt in DT:
        wT = torch.zeros(2 * 64 * 48 + 16, device=DEV, dtype=dt)
        mats, tiles = ops.build_cast_table([(offs[0], 64, 48, 0, 64 * 48 + 8, 0), (offs[4], 16, 16, -1, 2 * 64 * 48 + 16, 24)], DEV)
        wT = torch.zeros(2 * 64 * 48 + 16 + 16 * 24, device=DEV, dtype=dt)
        ops.cast_weights(pd, wT, mats, tiles)
        W4 = pd[offs[4]:offs[4] + 256].reshape(16, 16).cpu()
        t4 = wT[2 * 64 * 48 + 16:].reshape(16, 24).cpu()
        assert torch.equal(t4[:, :16], W4.t().contiguous().to(dt)) and float(t4[:, 16:].abs().max()) == 0.0   # padded ld
        W = pd[offs[0]:offs[0] + 64 * 48].reshape(64, 48).cpu()
        assert torch.equal(wT[:64 * 48].reshape(64, 48).cpu(), W.to(dt))
        assert torch.equal(wT[64 * 48 + 8:64 * 48 + 8 + 64 * 48].reshape(48, 64).cpu(), W.t().contiguous().to(dt))


# ----------------------------------------------------------------------------------------------- LayerNorm fused into the consuming GEMM
@pytest.mark.parametrize("mode", ["plain", "bias", "bias_swish_pre", "swish_grad"])
@pytest.mark.parametrize("M,N", [(9664, 1024), (9664, 768), (200, 256), (77, 512)])
def test_ln_gemm_matches_layernorm_then_gemm(mode, M, N):
    """lidk_ln_gemm_nt (row-panel kernel, LN in the operand load) against lidk_layernorm_fwd + lidk_gemm_nt on the same
    inputs: h / mean / rstd bit-identical (same arithmetic), outputs equal up to MFMA accumulation order; and against torch."""
    torch.manual_seed(M + N)
    K = 256
    x = (3.0 * torch.randn(M, K, device=DEV) + 0.5)
    gamma, beta = 1 + 0.1 * torch.randn(K, device=DEV), 0.1 * torch.randn(K, device=DEV)
    W = (0.06 * torch.randn(N, K, device=DEV)).bfloat16()
    bias = 0.1 * torch.randn(N, device=DEV) if mode in ("bias", "bias_swish_pre") else None
    aux = torch.randn(M, N, device=DEV).bfloat16() if mode == "swish_grad" else None
    act = {"plain": L.ACT_NONE, "bias": L.ACT_NONE, "bias_swish_pre": L.ACT_SWISH, "swish_grad": L.ACT_SWISH_GRAD}[mode]
    assert ops.ln_gemm_supported(M, N, K, torch.bfloat16)
    # reference path: the two existing launches
    h0 = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
    mean0, rstd0 = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.layernorm_fwd(x, gamma, beta, yT=h0, mean=mean0, rstd=rstd0)
    out0 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    pre0 = torch.empty_like(out0) if mode == "bias_swish_pre" else None
    ops.gemm_nt(h0, W, out0, bias=bias, act=act, out2=pre0, aux=aux)
    # fused
    h1 = torch.zeros_like(h0)
    mean1, rstd1 = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    out1 = torch.zeros_like(out0)
    pre1 = torch.zeros_like(out0) if mode == "bias_swish_pre" else None
    ops.ln_gemm_nt(x, gamma, beta, W, out1, h=h1, mean=mean1, rstd=rstd1, bias=bias, act=act, out2=pre1, aux=aux)
    torch.cuda.synchronize()
    # the same formulas in two kernels (the standalone LayerNorm now walks two rows per wave): statistics agree to f32 rounding,
    # the bf16 h to one ulp on a vanishing fraction of elements
    assert float((mean0 - mean1).abs().max()) <= 1e-6 and float(((rstd0 - rstd1) / rstd0).abs().max()) <= 1e-6
    dh = (h0.float() - h1.float()).abs()
    assert float(dh.max()) <= 2.0 ** -7 * float(h0.float().abs().max()) and float((dh > 0).float().mean()) < 1e-3
    scale = float(out0.float().abs().max())
    err = float((out0.float() - out1.float()).abs().max())
    print(f"[ln_gemm {mode} M={M} N={N}] max |fused - unfused| = {err:.3e} (max |out| {scale:.2f})")
    assert err <= 2e-2 * scale                       # bf16 outputs: one ulp at the top of the range is 2^-8 relative
    if pre0 is not None:
        assert float((pre0.float() - pre1.float()).abs().max()) <= 2e-2 * float(pre0.float().abs().max())
    ref = F.layer_norm(x, (K,), gamma, beta, 1e-5).bfloat16().float() @ W.float().t()
    if bias is not None:
        ref = ref + bias
    if mode == "bias_swish_pre":
        ref = ref * torch.sigmoid(ref)
    if mode == "swish_grad":
        s = torch.sigmoid(aux.float())
        ref = ref * (s * (1 + aux.float() * (1 - s)))
    assert float((out1.float() - ref).abs().max()) <= 3e-2 * max(1.0, float(ref.abs().max()))
    # the same kernel without LayerNorm, on the bf16 operand
    out2 = torch.zeros_like(out0)
    ops.ln_gemm_nt(None, None, None, W, out2, A=h1, bias=bias, act=act, out2=pre1, aux=aux)       # h1: the panel kernel's own h
    assert torch.equal(out1, out2)


def test_ln_gemm_refuses_unsupported_shapes():
    x = torch.randn(64, 128, device=DEV)
    W = torch.randn(256, 128, device=DEV).bfloat16()
    assert not ops.ln_gemm_supported(64, 256, 128, torch.bfloat16) and not ops.ln_gemm_supported(64, 320, 256, torch.bfloat16)
    assert not ops.ln_gemm_supported(64, 256, 256, torch.float32)
    with pytest.raises(LidkError):
        ops.ln_gemm_nt(x, torch.ones(128, device=DEV), torch.zeros(128, device=DEV), W, torch.empty(64, 256, device=DEV, dtype=torch.bfloat16))


# ----------------------------------------------------------------------------------------------- greedy CTC decode on the device
def test_ctc_greedy_matches_the_reference_collapse():
    """lidk_ctc_greedy against the reference's own collapse KAT (tests/golden/metrics_kat.npz, captured from lid/tokenizer.py)
    and against the host implementation on random logits with ragged lengths (V+1 = 41 and 4442, T up to 835)."""
    from conftest import load_npz
    from lid.tokenizer import CTCTokenizer
    g = load_npz("metrics_kat.npz")
    seqs, lens = torch.from_numpy(g["ctc_seqs"]), torch.from_numpy(g["ctc_lens"])
    logits = torch.full((seqs.shape[0], seqs.shape[1], 7), -5.0)
    logits.scatter_(2, seqs.unsqueeze(-1), 3.0)                              # argmax path = the KAT's symbol sequence
    tok = CTCTokenizer([chr(ord("a") + i) for i in range(6)])
    ids, n = ops.ctc_greedy(logits.to(DEV), lens.to(DEV), 6)
    assert tok.ids_to_text(ids, n) == [str(s) for s in g["ctc_decoded"]]
    torch.manual_seed(0)
    for B, T, V1 in ((64, 151, 41), (3, 835, 41), (2, 60, 4442), (1, 1, 5)):
        lg = torch.randn(B, T, V1)
        lg[..., V1 - 1] += 1.5                                                 # plenty of blanks
        lg[:, 1::3] = lg[:, 0:-1:3][:, :lg[:, 1::3].shape[1]]                  # repeated frames -> merged symbols
        in_len = torch.randint(1, T + 1, (B,))
        tok = CTCTokenizer([chr(0x4E00 + i) for i in range(V1 - 1)])
        want = tok.ctc_decode(lg.argmax(-1), in_len)
        ids, n = ops.ctc_greedy(lg.to(DEV), in_len.to(DEV), V1 - 1)
        assert tok.ids_to_text(ids, n) == want, (B, T, V1)
        ids, n = ops.ctc_greedy(lg.to(DEV), None, V1 - 1)
        assert tok.ids_to_text(ids, n) == tok.ctc_decode(lg.argmax(-1)), (B, T, V1)


# ----------------------------------------------------------------------------------------------- speed perturbation on the device
def test_speed_perturb_matches_the_float64_restatement():
    """lidk_speed_perturb (polyphase FIR, one launch for a batch with mixed factors and ragged lengths) against
    oracle.features.speed_perturb_np per utterance; factor 1 is an exact copy; rows are zero behind the perturbed length."""
    torch.manual_seed(0)
    B, L = 6, 9000
    wav = torch.randn(B, L)
    lens = [9000, 7000, 8123, 9000, 5000, 6001]
    for b, n in enumerate(lens):
        wav[b, n:] = 0
    factors = [(11, 10), (1, 1), (9, 10), (9, 10), (11, 10), (1, 1)]
    out, n_out, out_lens = ops.speed_perturb(wav.to(DEV), factors, torch.tensor(lens, dtype=torch.int32, device=DEV))
    assert n_out.tolist() == out_lens == [ops.speed_out_len(n, p, q) for n, (p, q) in zip(lens, factors)]
    assert out.shape == (B, max(out_lens))
    out = out.cpu()
    for b, (p, q) in enumerate(factors):
        ref = of.speed_perturb_np(wav[b, :lens[b]].numpy(), p, q)
        got = out[b, :out_lens[b]].numpy()
        err = np.abs(got - ref).max()
        assert err <= (0.0 if p == q else 2e-5), (b, err)
        assert float(out[b, out_lens[b]:].abs().max() if out_lens[b] < out.shape[1] else 0.0) == 0.0
    # full rows (no length vector), every utterance the same factor: the wav_augment() form
    out2, _, lens2 = ops.speed_perturb(wav[:2].to(DEV), [(11, 10)] * 2)
    assert lens2 == [ops.speed_out_len(L, 11, 10)] * 2
    np.testing.assert_allclose(out2[0].cpu().numpy(), of.speed_perturb_np(wav[0].numpy(), 11, 10), atol=2e-5)


@pytest.mark.parametrize("M", [9664, 1000])
def test_grouped_weight_gradients_equal_the_separate_launches(M, monkeypatch):
    """lidk_gemm_tn_grouped (one launch for a block's weight gradients) against lidk_gemm_tn per site and against torch:
    full-tile shapes (M = 9664 = 151 * 64) and a ragged M with a narrow, non-multiple-of-64 site."""
    torch.manual_seed(M)
    shapes = [(256, 1024), (1024, 256), (256, 512), (768, 256), (256, 256)] + ([(48, 256)] if M % 64 else [])
    ents, refs, seps = [], [], []
    for n1, n2 in shapes:
        X = (0.5 * torch.randn(M, n1)).bfloat16().to(DEV)
        Y = (0.5 * torch.randn(M, n2)).bfloat16().to(DEV)
        C0 = torch.randn(n1, n2).to(DEV)
        cs0 = torch.randn(n1).to(DEV)
        Cg, csg, Cs, css = C0.clone(), cs0.clone(), C0.clone(), cs0.clone()
        ents.append((X, Y, Cg, csg, M, n1, n2))
        ops.gemm_tn(X, Y, Cs, colsum=css, splitk=8)
        seps.append((Cs, css))
        refs.append((C0 + X.float().t() @ Y.float(), cs0 + X.float().sum(0)))
    grp = ops.build_tn_group(ents, split=4)
    assert grp[3] == (M % 64 == 0)
    ops.gemm_tn_grouped(grp)
    torch.cuda.synchronize()
    for (X, Y, Cg, csg, *_), (Cs, css), (Cr, csr) in zip(ents, seps, refs):
        scale = float(Cr.abs().max())
        assert float((Cg - Cr).abs().max()) <= 2e-3 * scale and float((Cg - Cs).abs().max()) <= 2e-3 * scale
        assert float((csg - csr).abs().max()) <= 2e-3 * float(csr.abs().max())
    if M % 64:                                   # 128x128 output tiles need full shapes; a ragged group is refused
        with pytest.raises(LidkError):
            ops.build_tn_group(ents, split=2, tile=128)
        return
    # the training default is 128-tiles x 2 row chunks (lidk_gemm_tn_grouped128); LIDK_TN_DMA picks the register-staged tile (0)
    # or the LDS-DMA ring of 2 / 3 / 4 stages (read per call); split 7 leaves chunks of 22 / 19 row tiles, 151 a single one
    for dma, split in ((None, 2), (None, 4), ("0", 1), ("0", 2), ("0", 3), ("2", 2), ("3", 1), ("3", 2), ("4", 2), ("4", 7), ("3", 151)):
        if dma is None:
            monkeypatch.delenv("LIDK_TN_DMA", raising=False)          # the default: ring of 3 (<= 256 items) or 2
        else:
            monkeypatch.setenv("LIDK_TN_DMA", dma)
        ents2 = [(X, Y, torch.zeros_like(Cg), torch.zeros_like(csg), M, n1, n2) for X, Y, Cg, csg, _, n1, n2 in ents]
        grp = ops.build_tn_group(ents2, split=split, tile=128)
        assert grp[3] == 128
        ops.gemm_tn_grouped(grp)
        torch.cuda.synchronize()
        for (X, Y, Cg, csg, *_), (Cr, csr) in zip(ents2, refs):
            ref, refc = X.float().t() @ Y.float(), X.float().sum(0)
            assert float((Cg - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
            assert float((csg - refc).abs().max()) <= 2e-3 * float(refc.abs().max())
    # 256x256 output tiles (lidk_gemm_tn_grouped256, 8 waves, two 64 KB LDS-DMA stages), uniform and per-entry row splits
    for split in (1, 8, 151, [10, 9, 3, 1, 2]):
        ents3 = [(X, Y, torch.zeros_like(Cg), torch.zeros_like(csg), M, n1, n2) for X, Y, Cg, csg, _, n1, n2 in ents]
        grp = ops.build_tn_group(ents3, split=split, tile=256)
        assert grp[3] == 256
        ops.gemm_tn_grouped(grp)
        torch.cuda.synchronize()
        for (X, Y, Cg, csg, *_), (Cr, csr) in zip(ents3, refs):
            ref, refc = X.float().t() @ Y.float(), X.float().sum(0)
            assert float((Cg - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
            assert float((csg - refc).abs().max()) <= 2e-3 * float(refc.abs().max())


def test_gemm_with_batchnorm_swish_backward_sums_in_the_epilogue():
    """lidk_gemm_nt_bn_sums (the conv module's second pointwise convolution run backwards, lid/conformer.py:197-199, with the
    BatchNorm + Swish backward statistics of its output formed in the GEMM's epilogue) against the two launches it replaces:
    the same ds bit for bit, and the same per-channel sums (sum dz | sum dz * xhat) up to the summation order."""
    from lidk import _lib as L
    g = torch.Generator().manual_seed(3)
    M, N, K = 9664, 512, 256
    A = (0.5 * torch.randn(M, K, generator=g)).to(DEV).bfloat16()
    W = (torch.randn(N, K, generator=g) / 16).to(DEV).bfloat16()
    c = torch.randn(M, N, generator=g).to(DEV).bfloat16()
    mean, rstd = (0.1 * torch.randn(N, generator=g)).to(DEV), (0.5 + torch.rand(N, generator=g)).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
    ds0, ds1 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16), torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    part = torch.zeros(L.BN_PARTIAL_BLOCKS * 2 * N, device=DEV)
    ops.gemm_nt(A, W, ds0)
    ops.bn_swish_bwd_reduce(ds0, c, mean, rstd, gamma, beta, part)
    ref = torch.zeros(2 * N + 1, device=DEV, dtype=torch.float64)
    ops.reduce_partials_f64(part, L.BN_PARTIAL_BLOCKS, 2 * N, ref, tail=M)
    part2 = torch.full((L.BN_PARTIAL_BLOCKS * 2 * N,), float("nan"), device=DEV)
    n = ops.gemm_nt_bn_sums(A, W, ds1, c, mean, rstd, gamma, beta, part2)
    assert n == (M // 64) * 2
    got = torch.zeros(2 * N + 1, device=DEV, dtype=torch.float64)
    ops.reduce_partials_f64(part2, n, 2 * N, got, tail=M)
    torch.cuda.synchronize()
    assert torch.equal(ds0, ds1)
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"[gemm + bn sums] max rel err of the sums {err:.2e} (|sums| up to {float(ref[:2 * N].abs().max()):.1f})")
    assert err <= 2e-6 and float(got[2 * N]) == M
    # shapes outside the pipelined kernel are refused (the engine then issues the two launches)
    assert ops.gemm_nt_bn_sums(A[:640], W, ds1[:640], c[:640], mean, rstd, gamma, beta, part2) == 0


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_multi_tensor_adam_and_sgd_follow_torch(wd):
    """ccml.optim.multi_tensor.Adam / SGD (one lidk_adam_multi / lidk_sgd_multi launch per step) against torch.optim.Adam / SGD on the
    same parameters and gradients for several steps: tensors of odd sizes (scalar tail, unaligned views of a flat buffer), one
    parameter that only gets a gradient from the third step on (its own bias-correction step count), state in torch's layout."""
    from ccml.optim.multi_tensor import SGD, Adam
    gen = g(300)
    shapes = [(1024, 257), (13,), (4096,), (3, 5, 7), (16384 * 2 + 5,)]
    flat = torch.randn(sum(int(np.prod(s)) for s in shapes) + 3, generator=gen)
    for cls, ref_cls, kw in ((Adam, torch.optim.Adam, dict(lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd)),
                             (SGD, torch.optim.SGD, dict(lr=0.05, momentum=0.9, weight_decay=wd, nesterov=True)),
                             (SGD, torch.optim.SGD, dict(lr=0.05, weight_decay=wd))):
        ours, theirs, off = [], [], 3                                    # offset 3: views that are not 16-byte aligned
        buf = flat.clone().to(DEV)
        for s in shapes:
            n = int(np.prod(s))
            ours.append(torch.nn.Parameter(buf[off:off + n].view(s)))
            theirs.append(torch.nn.Parameter(flat[off:off + n].clone().view(s).to(DEV)))
            off += n
        oa, ob = cls(ours, **kw), ref_cls(theirs, foreach=False, **kw)
        for step in range(5):
            for i, (a, b) in enumerate(zip(ours, theirs)):
                if i == 1 and step < 2:
                    a.grad = b.grad = None
                    continue
                gr = torch.randn(a.shape, generator=gen).to(DEV)
                a.grad, b.grad = gr.clone(), gr.clone()
            oa.step(); ob.step()
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(ours, theirs)):
            err = float((a - b).abs().max())
            assert err <= 2e-6 * max(1.0, float(b.abs().max())), (cls.__name__, kw, i, err)
        if cls is Adam:
            sa, sb = oa.state[ours[1]], ob.state[theirs[1]]
            assert float(sa["step"]) == float(sb["step"]) == 3.0
            assert float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 1e-6
