"""TEST INFRASTRUCTURE: a torch-CPU stand-in for ``lidk.ops`` with the same call signatures and in-place output
semantics, so the host-side orchestration (engine forward/backward chain, trainer step, data-parallel bookkeeping) can be
exercised on a machine without a GPU.  Never imported by the product; the real kernels are checked against torch per op
in tests/test_gpu_ops.py and end to end in tests/test_gpu_model.py."""
import math

import torch
import torch.nn.functional as F

ACT_NONE, ACT_SWISH, ACT_RELU, ACT_SWISH_GRAD = 0, 1, 2, 3
LN_PARTIAL_BLOCKS = 256
BN_PARTIAL_BLOCKS = 1024


def _f(t):
    return t.float()


def scale_cast(x, out, scale=1.0):
    out.copy_((_f(x) * scale).to(out.dtype))
    return out


def scale_cast_2d(x, out, M, N, scale=1.0):
    out[:M, :N] = (_f(x[:M, :N]) * scale).to(out.dtype)
    return out


def _keep(n, p, seed):
    g = torch.Generator().manual_seed(int(seed) % (2 ** 31))
    return torch.rand(n, generator=g) >= p


def dropout(x, out, p, seed=0, keep_in=None, keep_out=None):
    keep = keep_in.reshape(-1).bool() if keep_in is not None else _keep(x.numel(), p, seed)
    if keep_out is not None:
        keep_out.copy_(keep.to(torch.uint8))
    out.copy_((_f(x).reshape(-1) * keep / (1 - p)).reshape(x.shape).to(out.dtype))
    return out


def relu_bwd(dy, y, dx):
    dx.copy_(dy * (_f(y) > 0))
    return dx


def colsum(x, out, partial, scale=1.0):
    out += scale * _f(x).sum(0)
    return out


def transpose(x, out):
    out[:x.shape[1], :x.shape[0]] = x.t()
    return out


def reduce_partials_f64(partial, nparts, ncols, out, out2=None, tail=0.0):
    out[:ncols].copy_(partial[:nparts * ncols].view(nparts, ncols).double().sum(0))
    if tail > 0:
        out[ncols] = tail
    if out2 is not None:
        out2[:out.shape[0]].copy_(out)
    return out


def layernorm_fwd(x, gamma, beta, yT=None, y32=None, mean=None, rstd=None, eps=1e-5, dtype=None):
    mu = x.mean(-1)
    var = x.var(-1, unbiased=False)
    rs = torch.rsqrt(var + eps)
    y = (x - mu[:, None]) * rs[:, None] * gamma + beta
    if yT is not None:
        yT.copy_(y.to(yT.dtype))
    if y32 is not None:
        y32.copy_(y)
    if mean is not None:
        mean.copy_(mu)
    if rstd is not None:
        rstd.copy_(rs)


def layernorm_bwd(dy, x, mean, rstd, gamma, partial, dres=None, dx=None, dxT=None, dxT_scale=1.0, dgamma=None, dbeta=None,
                  dtype=None):
    dyf = _f(dy)
    xh = (x - mean[:, None]) * rstd[:, None]
    g = dyf * gamma
    d = rstd[:, None] * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    if dres is not None:
        d = d + dres
    C = x.shape[1]
    partial[:2 * C] = torch.cat([(dyf * xh).sum(0), dyf.sum(0)])          # "row 0" of the partial sums; the rest unused
    if dgamma is not None:
        dgamma += partial[:C]
    if dbeta is not None:
        dbeta += partial[C:2 * C]
    if dx is not None:
        dx.copy_(d)
    if dxT is not None:
        dxT.copy_((d * dxT_scale).to(dxT.dtype))


def layernorm_param_grads(partial, M, C, dgamma, dbeta):
    if dgamma is not None:
        dgamma += partial[:C]
    if dbeta is not None:
        dbeta += partial[C:2 * C]


def gemm_nt(A, B, out, bias=None, act=ACT_NONE, alpha=1.0, res=None, out2=None, aux=None, splitk=1, M=None, N=None, K=None):
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = B.shape[0] if N is None else N
    v = _f(A[:M, :K]) @ _f(B[:N, :K]).t()
    if bias is not None:
        v = v + bias
    if act == ACT_SWISH:
        if out2 is not None:
            out2[:M, :N] = v.to(out2.dtype)
        v = v * torch.sigmoid(v)
    elif act == ACT_RELU:
        v = torch.relu(v)
    elif act == ACT_SWISH_GRAD:
        a = _f(aux[:M, :N])
        s = torch.sigmoid(a)
        v = v * s * (1 + a * (1 - s))
    v = v * alpha
    if res is not None:
        v = v + res[:M, :N]
    if splitk > 1:
        out[:M, :N] += v
    else:
        out[:M, :N] = v.to(out.dtype)
    return out


def gemm_tn(X, Y, C, colsum=None, alpha=1.0, splitk=1, M=None, N1=None, N2=None):
    M = X.shape[0] if M is None else M
    N1 = X.shape[1] if N1 is None else N1
    N2 = Y.shape[1] if N2 is None else N2
    x, y = _f(X[:M, :N1]), _f(Y[:M, :N2])
    C[:N1, :N2] += alpha * (x.t() @ y)
    if colsum is not None:
        colsum[:N1] += alpha * x.sum(0)
    return C


def _attn(qkv, emb, B, T, H, dh):
    inner = H * dh
    q, k, v = _f(qkv).split(inner, dim=-1)
    q, k, v = (t.reshape(B, T, H, dh).transpose(1, 2) for t in (q, k, v))
    scale = dh ** -0.5
    mp = (emb.shape[0] - 1) // 2
    seq = torch.arange(T)
    dist = (seq[:, None] - seq[None, :]).clamp(-mp, mp) + mp
    dots = (q @ k.transpose(-1, -2) + torch.einsum("bhnd,nrd->bhnr", q, emb[dist])) * scale
    p = dots.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(B * T, inner), p


def attn_ldp(T, dh, dtype):
    return T


def attn_fwd(qkv, rel_emb, out, probs, B, T, heads, dh, rel_emb_T=None):
    o, p = _attn(qkv, rel_emb, B, T, heads, dh)
    out.copy_(o.to(out.dtype))
    probs[..., :T].copy_(p.to(probs.dtype))


@torch.enable_grad()
def attn_bwd(qkv, rel_emb, probs, dout, dqkv, drel_emb, dscores, B, T, heads, dh, rel_emb_T=None):
    q = _f(qkv).detach().clone().requires_grad_()
    e = rel_emb.detach().clone().requires_grad_()
    o, _ = _attn(q, e, B, T, heads, dh)
    o.backward(_f(dout))
    dqkv.copy_(q.grad.to(dqkv.dtype))
    drel_emb += e.grad


def glu_fwd(y, g):
    C = y.shape[1] // 2
    g.copy_((_f(y[:, :C]) * torch.sigmoid(_f(y[:, C:]))).to(g.dtype))


def glu_bwd(y, dg, dy):
    C = y.shape[1] // 2
    a, b, d = _f(y[:, :C]), _f(y[:, C:]), _f(dg)
    s = torch.sigmoid(b)
    dy[:, :C] = (d * s).to(dy.dtype)
    dy[:, C:] = (d * a * s * (1 - s)).to(dy.dtype)


def dwconv_stat_parts(B, T, C=0, dtype=None):
    return B * ((T + 31) // 32)


def _dw(x, w, bias, B, T, pad_left):
    C, K = w.shape
    xx = _f(x).view(B, T, C).transpose(1, 2)
    y = F.conv1d(F.pad(xx, (pad_left, K - 1 - pad_left)), w[:, None, :], bias, groups=C)
    return y.transpose(1, 2).reshape(B * T, C)


def dwconv_fwd(g, w, bias, c, stat_partial, B, T, pad_left):
    y = _dw(g, w, bias, B, T, pad_left)
    c.view(B * T, -1).copy_(y.to(c.dtype))
    if stat_partial is not None:
        C = w.shape[0]
        stat_partial.zero_()
        stat_partial[:C] = y.sum(0)
        stat_partial[C:2 * C] = (y * y).sum(0)


@torch.enable_grad()
def dwconv_bwd_input(dc, w, dg, B, T, pad_left):
    x = torch.zeros(B * T, w.shape[0], requires_grad=True)
    _dw(x, w, None, B, T, pad_left).backward(_f(dc).view(B * T, -1))
    dg.view(B * T, -1).copy_(x.grad.to(dg.dtype))


def glu_dwconv_fwd(y, w, bias, g, c, stat_partial, B, T, pad_left):
    gg = g if g is not None else torch.empty(y.shape[0], y.shape[1] // 2, dtype=y.dtype)
    glu_fwd(y, gg)
    dwconv_fwd(gg, w, bias, c, stat_partial, B, T, pad_left)


def dwconv_bwd_input_glu(dc, w, y, dy, B, T, pad_left):
    dg = torch.empty(y.shape[0], y.shape[1] // 2, dtype=torch.float32)
    dwconv_bwd_input(dc, w, dg, B, T, pad_left)
    glu_bwd(y, dg, dy)


def dwconv_bwd_input_bn_glu(ds, c, mean, rstd, gamma, beta, sums, count, w, y, dy, B, T, pad_left):
    dz, xh = _dz(ds, c, mean, rstd, gamma, beta)
    C = c.shape[1]
    count = count if count > 0 else float(sums[2 * C])
    m0, m1 = (sums[:C] / count).float(), (sums[C:2 * C] / count).float()
    dc = gamma * rstd * (dz - m0 - xh * m1)
    dwconv_bwd_input_glu(dc, w, y, dy, B, T, pad_left)


@torch.enable_grad()
def dwconv_bwd_weight(dc, g, dw, db, partial, B, T, pad_left):
    w = torch.zeros_like(dw).requires_grad_()
    b = torch.zeros(dw.shape[0], requires_grad=True)
    _dw(g, w, b, B, T, pad_left).backward(_f(dc).view(B * T, -1))
    dw += w.grad
    if db is not None:
        db += b.grad


def bn_train_stats(sums, count, mean, rstd, running_mean, running_var, nbt, momentum=0.1, eps=1e-5):
    C = mean.shape[0]
    count = count if count > 0 else float(sums[2 * C])
    mu = sums[:C] / count
    var = (sums[C:2 * C] / count - mu * mu).clamp_min(0)
    mean.copy_(mu.float())
    rstd.copy_((1.0 / torch.sqrt(var + eps)).float())
    if running_mean is not None:
        running_mean.mul_(1 - momentum).add_(momentum * mu.float())
    if running_var is not None:
        running_var.mul_(1 - momentum).add_(momentum * (var * count / max(count - 1, 1)).float())
    if nbt is not None:
        nbt += 1


def bn_eval_stats(running_mean, running_var, mean, rstd, eps=1e-5):
    mean.copy_(running_mean)
    rstd.copy_(1.0 / torch.sqrt(running_var + eps))


def bn_swish_fwd(c, mean, rstd, gamma, beta, s):
    z = (_f(c) - mean) * rstd * gamma + beta
    s.copy_((z * torch.sigmoid(z)).to(s.dtype))


def _dz(ds, c, mean, rstd, gamma, beta):
    xh = (_f(c) - mean) * rstd
    z = xh * gamma + beta
    sg = torch.sigmoid(z)
    return _f(ds) * sg * (1 + z * (1 - sg)), xh


def bn_swish_bwd_reduce(ds, c, mean, rstd, gamma, beta, partial):
    dz, xh = _dz(ds, c, mean, rstd, gamma, beta)
    C = c.shape[1]
    partial[:BN_PARTIAL_BLOCKS * 2 * C].zero_()
    partial[:C] = dz.sum(0)
    partial[C:2 * C] = (dz * xh).sum(0)


def bn_swish_bwd_apply(ds, c, mean, rstd, gamma, beta, sums, sums_local, count, dc, dgamma, dbeta):
    dz, xh = _dz(ds, c, mean, rstd, gamma, beta)
    C = c.shape[1]
    count = count if count > 0 else float(sums[2 * C])
    m0, m1 = (sums[:C] / count).float(), (sums[C:2 * C] / count).float()
    dc.copy_((gamma * rstd * (dz - m0 - xh * m1)).to(dc.dtype))
    if dbeta is not None:
        dbeta += sums_local[:C].float()
    if dgamma is not None:
        dgamma += sums_local[C:2 * C].float()


def im2col_k3s2(mel, out, T):
    B, F_, C = mel.shape
    p = F.pad(mel, (0, 0, 1, 1))
    cols = torch.stack([p[:, 2 * t:2 * t + 3].reshape(B, 3 * C) for t in range(T)], 1).reshape(B * T, 3 * C)
    out.copy_(cols.to(out.dtype))


def build_cast_table(entries, device):
    return torch.tensor([list(e) + [0, 0] for e in entries], dtype=torch.int64), 1


def cast_weights(params, wT, mats, total_tiles):
    for src, R, C, w_off, t_off, ldt, _, _ in mats.tolist():
        W = params[src:src + R * C].view(R, C)
        if w_off >= 0:
            wT[w_off:w_off + R * C] = W.reshape(-1).to(wT.dtype)
        if t_off >= 0:
            ldt = ldt or R
            wT[t_off:t_off + C * ldt].view(C, ldt)[:, :R] = W.t().to(wT.dtype)


# ------------------------------------------------------------------------------------------------ loss / optimizer / features
def ctc_workspace_bytes(B, T, V1, Lmax):
    return 4


@torch.enable_grad()
def ctc_loss(logits, targets, in_len, tg_len, loss, dlogits, workspace, blank, grad_scale=1.0, zero_infinity=True):
    lg = logits.detach().clone().requires_grad_()
    per = F.ctc_loss(torch.log_softmax(lg, -1).transpose(0, 1), targets, in_len, tg_len, blank=blank, reduction="none",
                     zero_infinity=zero_infinity)
    loss.copy_(per.detach())
    if dlogits is not None:
        per.sum().backward()
        dlogits.copy_(lg.grad * grad_scale)


def lid_score(logits, scores_col, stride, blank):
    vmax, arg = torch.max(torch.log_softmax(logits, -1), -1)
    mask = arg != blank
    scores_col[:, 0] = (vmax * mask).sum(-1) / (mask.sum(-1) * math.log(blank) + 1e-5)


def speed_out_len(n, p, q):
    return int(n * q / p + 0.5)


def speed_perturb(wav, factors, n_samples=None):
    from oracle.features import speed_perturb_np
    B, Lin = wav.shape
    lens = n_samples.tolist() if n_samples is not None else [Lin] * B
    outs = [torch.from_numpy(speed_perturb_np(wav[b, :lens[b]].numpy(), p, q)).float() for b, (p, q) in enumerate(factors)]
    out_lens = [o.shape[0] for o in outs]
    out = torch.zeros(B, max(out_lens))
    for b, o in enumerate(outs):
        out[b, :o.shape[0]] = o
    return out, torch.tensor(out_lens, dtype=torch.int32), out_lens


def ctc_greedy(logits, in_len, blank):
    B, T, V1 = logits.shape
    arg = logits.argmax(-1)
    ids = torch.zeros(B, T, dtype=torch.int32)
    lens = torch.zeros(B, dtype=torch.int32)
    for b in range(B):
        n = T if in_len is None else int(min(T, max(0, int(in_len[b]))))
        prev, k = blank, 0
        for t in range(n):
            p = int(arg[b, t])
            if p != blank and p != prev:
                ids[b, k] = p
                k += 1
            prev = p
        lens[b] = k
    return ids, lens


def lid_mlp(scores, w0, b0, w2, b2, out):
    out.copy_(F.linear(F.relu(F.linear(scores, w0, b0)), w2, b2))
    return out


def novograd_step(params, grads, exp_avg, exp_avg_sq, work, n_tensors, lr, betas, eps, weight_decay, grad_averaging,
                  max_norm, scratch, total_norm):
    items = {}
    for t, off, ln in work.tolist():
        lo, hi = items.get(t, (off, off))
        items[t] = (min(lo, off), max(hi, off + ln))
    tot = math.sqrt(sum(float((grads[a:b] ** 2).sum()) for a, b in items.values()))
    total_norm.fill_(tot)
    clip = min(1.0, max_norm / (tot + 1e-6)) if max_norm > 0 else 1.0
    for t, (a, b) in items.items():
        g = grads[a:b] * clip
        n = (g ** 2).sum()
        v = n if float(exp_avg_sq[t]) == 0 else betas[1] * exp_avg_sq[t] + (1 - betas[1]) * n
        exp_avg_sq[t] = v
        g = g / (v.sqrt() + eps) + weight_decay * params[a:b]
        if grad_averaging:
            g = g * (1 - betas[0])
        exp_avg[a:b] = betas[0] * exp_avg[a:b] + g
        params[a:b] -= lr * exp_avg[a:b]
        grads[a:b] = 0                                # consumed, like the HIP launch


# ------------------------------------------------------------------------------------------------ feature path (via the oracle)
def normalize_wav(wav, out=None, n_samples=None):
    from oracle import features as of
    if n_samples is None:
        return of.normalize_wav(wav)
    y = torch.zeros_like(wav)
    for i, n in enumerate(n_samples.tolist()):
        y[i, :n] = of.normalize_wav(wav[i:i + 1, :n])[0]
    return y


def dither_preemph(wav, coef=0.97, dither=1e-5, seed=0, noise=None, out=None):
    from oracle import features as of
    if noise is None and dither:
        noise = torch.rand(wav.shape, generator=torch.Generator().manual_seed(int(seed) % (2 ** 31)))
    return of.dither_preemphasis(wav, noise if dither else None, coef)


def logmel(wav, pad=0, hop=160, win_length=400, n_mels=80, spans=None, top_db=80.0, out=None, n_samples=None):
    from oracle import features as of
    if n_samples is not None:                                           # ragged: each utterance alone, mel zero padded
        specs = []
        for i, n in enumerate(n_samples.tolist()):
            m = of.wav2mel(wav[i:i + 1, :n], n_mels=n_mels, pad=pad)[0]
            if spans is not None:
                m = of.apply_specaug(m, [tuple(s) for s in spans[i].tolist()])
            specs.append(m)
        full = 1 + (wav.shape[1] + 2 * pad) // hop
        out_ = torch.zeros(len(specs), full, n_mels)
        for i, m in enumerate(specs):
            out_[i, :m.shape[-1]] = m.transpose(0, 1)
        return out_
    mel = of.wav2mel(wav, n_mels=n_mels, pad=pad)                       # (B, n_mels, F)
    if spans is not None:
        mel = torch.stack([of.apply_specaug(mel[i], [tuple(s) for s in spans[i].tolist()]) for i in range(mel.shape[0])])
    return mel.transpose(1, 2).contiguous()
