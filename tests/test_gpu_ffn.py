"""Fused FeedForward kernels (csrc/ffn.hip) against a torch fp32 computation of lid/conformer.py:153-171 (+ PreNorm :81-89, Scale 0.5 and
the residual add :252-259) on the bf16-rounded operands the kernel sees, and against the three-launch sequence they replace."""
import pytest
import torch
import torch.nn.functional as F

from lidk import ops
from lidk import _lib as L

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


@pytest.fixture(autouse=True, params=[3, 4], ids=["rows48", "rows64"])
def _workgroup_height(request):
    """Every test runs with both workgroup heights of csrc/ffn.hip forced (48 rows / 6 waves, 64 rows / 8 waves); the default
    policy (by M) picks between exactly these two."""
    ops.ffn_option("LIDK_FFN_RG", request.param)
    yield request.param
    ops.ffn_option("LIDK_FFN_RG", -1)


def _case(M, ff, seed, d=256):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, d, generator=g) * 1.5 + 0.2
    gamma, beta = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    W1, b1 = torch.randn(ff, d, generator=g) / 16, 0.1 * torch.randn(ff, generator=g)
    W2, b2 = torch.randn(d, ff, generator=g) / 32, 0.1 * torch.randn(d, generator=g)
    return x, gamma, beta, W1, b1, W2, b2


def _ref_fwd(x, gamma, beta, W1, b1, W2, b2, alpha=0.5):
    """fp32 reference on bf16-rounded operands; h, a, u rounded where the kernel rounds them (operands of the next MFMA)."""
    h = F.layer_norm(x, (x.shape[1],), gamma, beta, 1e-5).to(BF).float()
    a = h @ W1.to(BF).float().t() + b1
    u = (a * torch.sigmoid(a)).to(BF).float()
    xo = x + alpha * (u @ W2.to(BF).float().t() + b2)
    mean = x.mean(1)
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt()
    return h, a, u, xo, mean, rstd


@pytest.mark.parametrize("M,ff", [(9664, 1024), (64, 64), (200, 256), (151, 1024), (1000, 3072)])
@pytest.mark.parametrize("ln_in", [True, False])
def test_ffn_fwd_matches_torch_and_saves_what_the_backward_needs(M, ff, ln_in):
    x, gamma, beta, W1, b1, W2, b2 = _case(M, ff, seed=M + ff)
    assert ops.ffn_fwd_supported(M, 256, ff, BF)
    xd = x.to(DEV)
    W1d, W2d = W1.to(DEV, BF), W2.to(DEV, BF)
    e = lambda *s, dt=BF: torch.full(s, float("nan"), device=DEV, dtype=dt)
    h, a, u, xo, mean, rstd = e(M, 256), e(M, ff), e(M, ff), e(M, 256, dt=torch.float32), e(M, dt=torch.float32), e(M, dt=torch.float32)
    rh, ra, ru, rxo, rmean, rrstd = _ref_fwd(x, gamma, beta, W1, b1, W2, b2)
    if ln_in:
        ops.ffn_fwd(xd, W1d, b1.to(DEV), W2d, b2.to(DEV), xo, gamma=gamma.to(DEV), beta=beta.to(DEV), h=h, mean=mean, rstd=rstd, a=a, u=u)
        assert float((mean.cpu() - rmean).abs().max()) <= 2e-6 and float(((rstd.cpu() - rrstd) / rrstd).abs().max()) <= 2e-6
        dh = (h.float().cpu() - rh).abs()
        assert float(dh.max()) <= 2 ** -6 * float(rh.abs().max()) and float(dh.mean()) <= 1e-4, (float(dh.max()), float(dh.mean()))
    else:           # h from the fused LayerNorm pair of the previous block: LayerNorm is skipped, mean / rstd / h untouched
        ops.ffn_fwd(xd, W1d, b1.to(DEV), W2d, b2.to(DEV), xo, h_in=rh.to(DEV, BF), a=a, u=u)
        assert bool(torch.isnan(mean).all()) and bool(torch.isnan(h.float()).all())
    torch.cuda.synchronize()
    # a, u: bf16 roundings of fp32 values that agree to ~1e-5 -> at most one bf16 ulp apart on a handful of elements
    for name, got, ref in (("a", a, ra.to(BF).float()), ("u", u, ru)):
        d = (got.float().cpu() - ref).abs()
        lim = 2 ** -7 * ref.abs() + 1e-6 + (0.03 if ln_in else 0.0)        # ln_in: h itself may differ by a bf16 ulp
        assert bool((d <= lim).all()), (name, float(d.max()))
        assert float(d.mean()) <= (2e-3 if ln_in else 1e-4), (name, float(d.mean()))
    d = (xo.cpu() - rxo).abs()
    assert float(d.max()) <= (2e-2 if ln_in else 2e-3) and float(d.mean()) <= (1e-3 if ln_in else 1e-4), (float(d.max()), float(d.mean()))


def test_ffn_fwd_equals_the_three_launch_sequence():
    """Same inputs through lidk_layernorm_fwd + 2 x lidk_gemm_nt: the saved tensors and the output agree to bf16 rounding."""
    M, ff = 9664, 1024
    x, gamma, beta, W1, b1, W2, b2 = _case(M, ff, seed=7)
    xd, gd, bd = x.to(DEV), gamma.to(DEV), beta.to(DEV)
    W1d, W2d, b1d, b2d = W1.to(DEV, BF), W2.to(DEV, BF), b1.to(DEV), b2.to(DEV)
    e = lambda *s, dt=BF: torch.empty(s, device=DEV, dtype=dt)
    h0, a0, u0, xo0, mean0, rstd0 = e(M, 256), e(M, ff), e(M, ff), e(M, 256, dt=torch.float32), e(M, dt=torch.float32), e(M, dt=torch.float32)
    ops.layernorm_fwd(xd, gd, bd, yT=h0, mean=mean0, rstd=rstd0)
    ops.gemm_nt(h0, W1d, u0, bias=b1d, act=L.ACT_SWISH, out2=a0)
    ops.gemm_nt(u0, W2d, xo0, bias=b2d, alpha=0.5, res=xd)
    h1, a1, u1, xo1, mean1, rstd1 = e(M, 256), e(M, ff), e(M, ff), e(M, 256, dt=torch.float32), e(M, dt=torch.float32), e(M, dt=torch.float32)
    ops.ffn_fwd(xd, W1d, b1d, W2d, b2d, xo1, gamma=gd, beta=bd, h=h1, mean=mean1, rstd=rstd1, a=a1, u=u1)
    torch.cuda.synchronize()
    assert float((mean0 - mean1).abs().max()) <= 2e-6
    assert float((h0.float() - h1.float()).abs().max()) <= 2 ** -6 * float(h0.float().abs().max())
    assert float((xo0 - xo1).abs().mean()) <= 1e-3 and float((xo0 - xo1).abs().max()) <= 3e-2
    # with the SAME h the two paths must agree to accumulation order
    ops.ffn_fwd(xd, W1d, b1d, W2d, b2d, xo1, h_in=h0, a=a1, u=u1)
    torch.cuda.synchronize()
    assert float((a0.float() - a1.float()).abs().max()) <= 2 ** -7 * float(a0.float().abs().max())
    assert float((xo0 - xo1).abs().max()) <= 2e-3


def test_ffn_fwd_refuses_other_widths():
    assert not ops.ffn_fwd_supported(128, 768, 3072, BF)
    assert not ops.ffn_fwd_supported(128, 256, 1000, BF)
    assert not ops.ffn_fwd_supported(128, 256, 1024, torch.float32)


def _ref_bwd(dyT, a, W1, W2, x, gamma, dres):
    """fp32 autograd-free restatement on the bf16-rounded operands: da (rounded to bf16 where the kernel rounds it), dh, LN'."""
    sg = torch.sigmoid(a)
    da = ((dyT @ W2.to(BF).float()) * (sg * (1 + a * (1 - sg)))).to(BF).float()
    dh = da @ W1.to(BF).float()
    mean = x.mean(1, keepdim=True)
    rstd = (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    xh = (x - mean) * rstd
    g = dh * gamma
    dx = dres + rstd * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
    return da, dh, dx, (dh * xh).sum(0), dh.sum(0), mean[:, 0], rstd[:, 0]


@pytest.mark.parametrize("M,ff", [(9664, 1024), (64, 64), (200, 256), (151, 1024)])
def test_ffn_bwd_matches_torch(M, ff):
    x, gamma, beta, W1, b1, W2, b2 = _case(M, ff, seed=3 * M + ff)
    g = torch.Generator().manual_seed(M)
    dyT = (0.5 * torch.randn(M, 256, generator=g)).to(BF).float()
    a = (torch.randn(M, ff, generator=g) * 1.5).to(BF).float()
    dres = torch.randn(M, 256, generator=g)
    rda, rdh, rdx, rdg, rdb, mean, rstd = _ref_bwd(dyT, a, W1, W2, x, gamma, dres)
    W1T, W2T = W1.t().contiguous().to(DEV, BF), W2.t().contiguous().to(DEV, BF)
    e = lambda *s, dt=BF: torch.full(s, float("nan"), device=DEV, dtype=dt)
    da, dx, dxT = e(M, ff), e(M, 256, dt=torch.float32), e(M, 256)
    rows = ops.ffn_bwd_partial_rows(M)
    partial = e(rows * 512, dt=torch.float32)
    ops.ffn_bwd(dyT.to(DEV, BF), a.to(DEV, BF), W1T, W2T, da, x=x.to(DEV), mean=mean.to(DEV), rstd=rstd.to(DEV), gamma=gamma.to(DEV),
                dres=dres.to(DEV), dx=dx, dxT=dxT, dxT_scale=0.5, partial=partial)
    dg, db = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    ops.layernorm_param_grads_rows(partial, rows, 256, dg, db)
    torch.cuda.synchronize()
    d = (da.float().cpu() - rda).abs()
    assert bool((d <= 2 ** -7 * rda.abs() + 1e-6).all()) and float(d.mean()) <= 1e-4, float(d.max())
    scale = float(rdx.abs().max())
    d = (dx.cpu() - rdx).abs()
    assert float(d.max()) <= 2e-3 * scale and float(d.mean()) <= 1e-4 * scale, (float(d.max()), float(d.mean()), scale)
    d = (dxT.float().cpu() - 0.5 * rdx).abs()
    assert float(d.max()) <= 2 ** -7 * 0.5 * scale
    for name, got, ref in (("dgamma", dg, rdg), ("dbeta", db, rdb)):
        assert float((got.cpu() - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-3, name
    # dh-output form (the LayerNorm backward is left to the caller)
    dh = e(M, 256)
    ops.ffn_bwd(dyT.to(DEV, BF), a.to(DEV, BF), W1T, W2T, da, dh=dh)
    torch.cuda.synchronize()
    d = (dh.float().cpu() - rdh).abs()
    assert float(d.max()) <= 2 ** -7 * float(rdh.abs().max()) + 1e-5


def test_ffn_bwd_equals_the_launch_sequence_it_replaces():
    M, ff = 9664, 1024
    x, gamma, beta, W1, b1, W2, b2 = _case(M, ff, seed=11)
    g = torch.Generator().manual_seed(5)
    dyT = (0.5 * torch.randn(M, 256, generator=g)).to(DEV, BF)
    a = (torch.randn(M, ff, generator=g) * 1.5).to(DEV, BF)
    dres, xd, gd = torch.randn(M, 256, generator=g).to(DEV), x.to(DEV), gamma.to(DEV)
    mean, rstd = xd.mean(1), (xd.var(1, unbiased=False) + 1e-5).rsqrt()
    W1T, W2T = W1.t().contiguous().to(DEV, BF), W2.t().contiguous().to(DEV, BF)
    e = lambda *s, dt=BF: torch.empty(s, device=DEV, dtype=dt)
    da0, dh0, dx0, dxT0 = e(M, ff), e(M, 256), e(M, 256, dt=torch.float32), e(M, 256)
    part0 = e(max(L.LN_PARTIAL_BLOCKS * 2 * 1024, L.LN_BWD_BLOCKS * 512), dt=torch.float32)
    dg0, db0 = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    ops.gemm_nt(dyT, W2T, da0, act=L.ACT_SWISH_GRAD, aux=a, N=ff, K=256)
    ops.gemm_nt(da0, W1T, dh0, N=256, K=ff)
    ops.layernorm_bwd(dh0, xd, mean, rstd, gd, part0, dres=dres, dx=dx0, dxT=dxT0, dxT_scale=1.0, dgamma=dg0, dbeta=db0)
    da1, dx1, dxT1 = e(M, ff), e(M, 256, dt=torch.float32), e(M, 256)
    rows = ops.ffn_bwd_partial_rows(M)
    part1 = e(rows * 512, dt=torch.float32)
    dg1, db1 = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    ops.ffn_bwd(dyT, a, W1T, W2T, da1, x=xd, mean=mean, rstd=rstd, gamma=gd, dres=dres, dx=dx1, dxT=dxT1, dxT_scale=1.0, partial=part1)
    ops.layernorm_param_grads_rows(part1, rows, 256, dg1, db1)
    torch.cuda.synchronize()
    assert float((da0.float() - da1.float()).abs().max()) <= 2 ** -7 * float(da0.float().abs().max())
    # the sequence rounds dh to bf16 between the GEMM and the LayerNorm backward; the fused kernel keeps it in f32
    scale = float(dx0.abs().max())
    assert float((dx0 - dx1).abs().max()) <= 1e-2 * scale and float((dx0 - dx1).abs().mean()) <= 5e-4 * scale
    assert float((dg0 - dg1).abs().max()) <= 5e-3 * float(dg0.abs().max()) and float((db0 - db1).abs().max()) <= 5e-3 * float(db0.abs().max())


def test_grouped_layernorm_param_grads_equal_the_single_launches():
    """lidk_layernorm_param_grads_grouped: five finalisers (mixed row counts) in one launch, bit-identical to the separate ones."""
    g = torch.Generator().manual_seed(3)
    M, Cn = 9664, 256
    rows = [ops.layernorm_bwd_partial_rows(M), ops.ffn_bwd_partial_rows(M), 1024, 151, 7]
    parts = [torch.randn(r * 2 * Cn, generator=g).to(DEV) for r in rows]
    dg0 = [torch.randn(Cn, generator=g).to(DEV) for _ in rows]
    db0 = [torch.randn(Cn, generator=g).to(DEV) for _ in rows]
    dg1, db1 = [t.clone() for t in dg0], [t.clone() for t in db0]
    for p, r, a, b in zip(parts, rows, dg0, db0):
        ops.layernorm_param_grads_rows(p, r, Cn, a, b)
    grp = ops.build_ln_param_group([(p, r, Cn, a, b) for p, r, a, b in zip(parts, rows, dg1, db1)])
    ops.layernorm_param_grads_grouped(grp)
    torch.cuda.synchronize()
    for a, b in zip(dg0 + db0, dg1 + db1):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,K", [(9664, 1024), (9664, 768), (200, 64), (151, 512)])
def test_dgrad_ln_bwd_matches_torch_and_the_two_launches(M, K):
    """lidk_dgrad_ln_bwd: data gradient of a projection behind a PreNorm + that LayerNorm's backward (lid/conformer.py:81-89 with
    :98-100 / :192), against fp32 torch on the bf16-rounded operands and against lidk_gemm_nt + lidk_layernorm_bwd."""
    g = torch.Generator().manual_seed(M + K)
    dy = (0.5 * torch.randn(M, K, generator=g)).to(BF).float()
    W = (torch.randn(K, 256, generator=g) / 16).to(BF).float()           # y = h @ W^T-style weight [K_out = K][256]
    x = torch.randn(M, 256, generator=g) * 1.5 + 0.2
    gamma = 1 + 0.1 * torch.randn(256, generator=g)
    dres = torch.randn(M, 256, generator=g)
    dh = dy @ W
    mean = x.mean(1, keepdim=True)
    rstd = (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    xh = (x - mean) * rstd
    gg = dh * gamma
    rdx = dres + rstd * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True))
    rdg, rdb = (dh * xh).sum(0), dh.sum(0)
    assert ops.dgrad_ln_bwd_supported(M, 256, K, BF)
    WT = W.t().contiguous().to(DEV, BF)                                  # [256][K]
    e = lambda *s, dt=BF: torch.full(s, float("nan"), device=DEV, dtype=dt)
    dx, dxT = e(M, 256, dt=torch.float32), e(M, 256)
    rows = ops.ffn_bwd_partial_rows(M)
    partial = e(rows * 512, dt=torch.float32)
    args = dict(x=x.to(DEV), mean=mean[:, 0].contiguous().to(DEV), rstd=rstd[:, 0].contiguous().to(DEV), gamma=gamma.to(DEV))
    ops.dgrad_ln_bwd(dy.to(DEV, BF), WT, args["x"], args["mean"], args["rstd"], args["gamma"], partial, dres=dres.to(DEV), dx=dx, dxT=dxT,
                     dxT_scale=0.5)
    dg, db = torch.zeros(256, device=DEV), torch.zeros(256, device=DEV)
    ops.layernorm_param_grads_rows(partial, rows, 256, dg, db)
    torch.cuda.synchronize()
    scale = float(rdx.abs().max())
    d = (dx.cpu() - rdx).abs()
    assert float(d.max()) <= 2e-3 * scale and float(d.mean()) <= 1e-4 * scale, (float(d.max()), float(d.mean()), scale)
    assert float((dxT.float().cpu() - 0.5 * rdx).abs().max()) <= 2 ** -7 * 0.5 * scale
    for name, got, ref in (("dgamma", dg, rdg), ("dbeta", db, rdb)):
        assert float((got.cpu() - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-3, name
    # the two launches it replaces (dh rounded to bf16 in between)
    dh0, dx0 = torch.empty(M, 256, device=DEV, dtype=BF), torch.empty(M, 256, device=DEV)
    part0 = torch.empty(max(L.LN_PARTIAL_BLOCKS * 2 * 1024, L.LN_BWD_BLOCKS * 512), device=DEV)
    ops.gemm_nt(dy.to(DEV, BF), WT, dh0, N=256, K=K)
    ops.layernorm_bwd(dh0, args["x"], args["mean"], args["rstd"], args["gamma"], part0, dres=dres.to(DEV), dx=dx0)
    torch.cuda.synchronize()
    assert float((dx0 - dx).abs().max()) <= 1e-2 * scale and float((dx0 - dx).abs().mean()) <= 5e-4 * scale


@pytest.mark.parametrize("M,ff", [(9664, 1024), (151, 256)])
@pytest.mark.parametrize("pair", [False, True])
def test_ffn_fwd_applies_the_consuming_layernorms_in_its_epilogue(M, ff, pair):
    """lidk_ffn_fwd_ln: A = LN(xo) (the next module's PreNorm, or post_norm) and optionally B = LN(A) (the next block's first
    PreNorm) against lidk_layernorm_fwd / torch on the kernel's own xo."""
    x, gamma, beta, W1, b1, W2, b2 = _case(M, ff, seed=5 * M + ff)
    g = torch.Generator().manual_seed(M + 1)
    gA, bA = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    gB, bB = 1 + 0.1 * torch.randn(256, generator=g), 0.1 * torch.randn(256, generator=g)
    e = lambda *s, dt=BF: torch.full(s, float("nan"), device=DEV, dtype=dt)
    f = lambda *s: e(*s, dt=torch.float32)
    h, a, u, xo, mean, rstd = e(M, 256), e(M, ff), e(M, ff), f(M, 256), f(M), f(M)
    yA32, yAT, meanA, rstdA, yBT, meanB, rstdB = f(M, 256), e(M, 256), f(M), f(M), e(M, 256), f(M), f(M)
    nl = dict(gA=gA.to(DEV), bA=bA.to(DEV), yA32=yA32, yAT=yAT, meanA=meanA, rstdA=rstdA)
    if pair:
        nl.update(gB=gB.to(DEV), bB=bB.to(DEV), yBT=yBT, meanB=meanB, rstdB=rstdB)
    ops.ffn_fwd(x.to(DEV), W1.to(DEV, BF), b1.to(DEV), W2.to(DEV, BF), b2.to(DEV), xo, gamma=gamma.to(DEV), beta=beta.to(DEV), h=h,
                mean=mean, rstd=rstd, a=a, u=u, next_ln=nl)
    torch.cuda.synchronize()
    xo_c = xo.cpu()
    assert bool(torch.isfinite(xo_c).all())
    rA = F.layer_norm(xo_c, (256,), gA, bA, 1e-5)
    assert float((yA32.cpu() - rA).abs().max()) <= 2e-5 * max(1.0, float(rA.abs().max()))
    assert float((yAT.float().cpu() - rA).abs().max()) <= 2 ** -7 * float(rA.abs().max())
    assert float((meanA.cpu() - xo_c.mean(1)).abs().max()) <= 2e-6 * max(1.0, float(xo_c.abs().max()))
    assert float(((rstdA.cpu() - (xo_c.var(1, unbiased=False) + 1e-5).rsqrt()) * xo_c.std(1)).abs().max()) <= 1e-5
    if pair:
        yA = yA32.cpu()
        rB = F.layer_norm(yA, (256,), gB, bB, 1e-5)
        assert float((yBT.float().cpu() - rB).abs().max()) <= 2 ** -7 * float(rB.abs().max())
        assert float((meanB.cpu() - yA.mean(1)).abs().max()) <= 2e-6
    else:
        assert bool(torch.isnan(yBT.float()).all())


def test_ffn_bwd_with_the_layernorm_pair_equals_dh_plus_layernorm2_bwd():
    """lidk_ffn_bwd_ln2 (this PreNorm's backward + the preceding post_norm's backward in the epilogue) against the two launches it
    replaces: lidk_ffn_bwd (dh form) + lidk_layernorm2_bwd; the two paths differ only by dh's bf16 rounding."""
    M, ff = 9664, 1024
    x, gamma, beta, W1, b1, W2, b2 = _case(M, ff, seed=21)
    g = torch.Generator().manual_seed(9)
    dyT = (0.5 * torch.randn(M, 256, generator=g)).to(DEV, BF)
    a = (torch.randn(M, ff, generator=g) * 1.5).to(DEV, BF)
    dres = torch.randn(M, 256, generator=g).to(DEV)
    x1 = (torch.randn(M, 256, generator=g) * 1.3 - 0.1).to(DEV)               # post_norm's input rows (the previous block's x4)
    g1 = (1 + 0.1 * torch.randn(256, generator=g)).to(DEV)
    b1n = (0.1 * torch.randn(256, generator=g)).to(DEV)
    mean1, rstd1 = x1.mean(1), (x1.var(1, unbiased=False) + 1e-5).rsqrt()
    y1 = F.layer_norm(x1, (256,), g1, b1n, 1e-5)                              # = this module's input x
    mean2, rstd2 = y1.mean(1), (y1.var(1, unbiased=False) + 1e-5).rsqrt()
    gd = gamma.to(DEV)
    W1T, W2T = W1.t().contiguous().to(DEV, BF), W2.t().contiguous().to(DEV, BF)
    e = lambda *s, dt=BF: torch.empty(s, device=DEV, dtype=dt)
    # reference sequence
    da0, dh0, dx0, dxT0 = e(M, ff), e(M, 256), e(M, 256, dt=torch.float32), e(M, 256)
    pa0, pb0 = e(L.LN_BWD_BLOCKS * 512, dt=torch.float32), e(L.LN_BWD_BLOCKS * 512, dt=torch.float32)
    ops.ffn_bwd(dyT, a, W1T, W2T, da0, dh=dh0)
    ops.layernorm2_bwd(dh0, dres, y1, mean2, rstd2, gd, x1, mean1, rstd1, g1, dx0, dxT0, 0.5, pa0, pb0)
    dg1_0, db1_0, dg2_0, db2_0 = (torch.zeros(256, device=DEV) for _ in range(4))
    ops.layernorm_param_grads(pa0, M, 256, dg1_0, db1_0)
    ops.layernorm_param_grads(pb0, M, 256, dg2_0, db2_0)
    # fused
    rows = ops.ffn_bwd_partial_rows(M)
    da1, dx1, dxT1 = e(M, ff), e(M, 256, dt=torch.float32), e(M, 256)
    p2, p1 = e(rows * 512, dt=torch.float32), e(rows * 512, dt=torch.float32)
    ops.ffn_bwd(dyT, a, W1T, W2T, da1, x=y1, mean=mean2, rstd=rstd2, gamma=gd, dres=dres, dx=dx1, dxT=dxT1, dxT_scale=0.5, partial=p2,
                pair=dict(x1=x1, mean1=mean1, rstd1=rstd1, gamma1=g1, partial1=p1))
    dg1_1, db1_1, dg2_1, db2_1 = (torch.zeros(256, device=DEV) for _ in range(4))
    ops.layernorm_param_grads_rows(p1, rows, 256, dg1_1, db1_1)
    ops.layernorm_param_grads_rows(p2, rows, 256, dg2_1, db2_1)
    torch.cuda.synchronize()
    assert torch.equal(da0, da1)
    scale = float(dx0.abs().max())
    assert float((dx0 - dx1).abs().max()) <= 1e-2 * scale and float((dx0 - dx1).abs().mean()) <= 5e-4 * scale
    assert float((dxT0.float() - dxT1.float()).abs().max()) <= 1e-2 * scale
    for name, r0, r1 in (("dg1", dg1_0, dg1_1), ("db1", db1_0, db1_1), ("dg2", dg2_0, dg2_1), ("db2", db2_0, db2_1)):
        assert float((r0 - r1).abs().max()) <= 5e-3 * float(r0.abs().max()) + 1e-3, name
