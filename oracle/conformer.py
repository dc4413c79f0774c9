"""Oracle: Conformer-LID model, functional torch-CPU fp32 restatement (test infrastructure).

Weights are a flat ``{name: tensor}`` dict using the reference's ``state_dict`` key names
(SURVEY.md section 8b), so a reference checkpoint drives this code directly.  Shapes:
mel ``(B, F, n_mels)`` -> encoder ``(B, T, d)`` -> per-language logits ``(B, T, V+1)``.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

LN_EPS = 1e-5
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


@dataclass
class ModelCfg:
    """Subset of ConformerMutiLangModel.__init__ (lid/ConformerLangModel.py:22-47)."""
    lang2vocab: Dict[str, int]
    lang2index: Dict[str, int]
    n_blocks: int = 14
    n_mels: int = 80
    encoder_dim: int = 144
    dim_head: int = 64
    heads: int = 4
    last_dim_head: int = 32
    last_heads: int = 8            # ConformerLinear hard-codes heads=8 (lid/ConformerLangModel.py:337-348)
    dropout: float = 0.0           # head Dropout (lid/ConformerLangModel.py:349)
    pos_dropout: float = 0.1       # RelPositionalEncoding(dropout_rate=0.1) (lid/conformer.py:426)
    stochastic_depth_p: float = 0.7
    hidden_dim: int = 32


@dataclass
class RunOpts:
    """Per-call randomness, supplied explicitly so both sides of a parity test share it."""
    training: bool = False
    keep_layers: Optional[List[bool]] = None       # stochastic depth decisions, lid/conformer.py:460-466
    pos_keep_mask: Optional[torch.Tensor] = None   # (B,T,d) bool/0-1, Dropout after x*sqrt(d)
    head_keep_mask: Optional[torch.Tensor] = None  # (B,T,d) Dropout in ConformerLinear
    bn_buffers: Dict[str, torch.Tensor] = field(default_factory=dict)  # updated running stats (training)
    # data-parallel oracle: all-reduce(SUM) of a float64 tensor over the ranks (None = single process).  With it the
    # BatchNorm layers follow torch.nn.SyncBatchNorm, which the reference's Trainer installs under DDP (ccml/trainer.py:428)
    all_reduce: Optional[object] = None


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def swish(x):
    """lid/conformer.py:34-36."""
    return x * torch.sigmoid(x)


def feed_forward(x, sd, p):
    """PreNorm(FeedForward) — lid/conformer.py:81-89,153-171; p = '<block>.ff1.fn'."""
    h = _ln(x, sd, p + ".norm")
    a = F.linear(h, sd[p + ".fn.net.0.weight"], sd[p + ".fn.net.0.bias"])
    return F.linear(swish(a), sd[p + ".fn.net.3.weight"], sd[p + ".fn.net.3.bias"])


def rel_pos_scores(q, emb, scale):
    """Shaw relative-position term — lid/conformer.py:123-127.
    q (B,h,T,dh), emb (2*max_pos+1, dh) -> (B,h,T,T) with [.., i, j] = q_i . emb[clamp(i-j)+max_pos] * scale."""
    n = q.shape[-2]
    max_pos = (emb.shape[0] - 1) // 2
    seq = torch.arange(n)
    dist = (seq[:, None] - seq[None, :]).clamp(-max_pos, max_pos) + max_pos
    e = emb[dist]                                        # (T, T, dh)
    return torch.einsum("bhnd,nrd->bhnr", q, e) * scale


def attention(x, sd, p, heads):
    """PreNorm(Attention) — lid/conformer.py:92-150; p = '<block>.attn'.  Mask branch is dead (Q3)."""
    h = _ln(x, sd, p + ".norm")
    q = F.linear(h, sd[p + ".fn.to_q.weight"])
    k, v = F.linear(h, sd[p + ".fn.to_kv.weight"]).chunk(2, dim=-1)
    B, T, inner = q.shape
    dh = inner // heads
    q, k, v = (t.reshape(B, T, heads, dh).transpose(1, 2) for t in (q, k, v))
    scale = dh ** -0.5
    dots = torch.matmul(q, k.transpose(-1, -2)) * scale
    dots = dots + rel_pos_scores(q, sd[p + ".fn.rel_pos_emb.weight"], scale)
    attn = dots.softmax(dim=-1)
    out = torch.matmul(attn, v).transpose(1, 2).reshape(B, T, inner)
    return F.linear(out, sd[p + ".fn.to_out.weight"], sd[p + ".fn.to_out.bias"])


def conv_module(x, sd, p, opts: RunOpts):
    """ConformerConvModule — lid/conformer.py:174-205; p = '<block>.conv.net'."""
    h = _ln(x, sd, p + ".0").transpose(1, 2)                              # (B, d, T)
    y = F.conv1d(h, sd[p + ".2.weight"], sd[p + ".2.bias"])               # (B, 4d, T)
    out, gate = y.chunk(2, dim=1)
    g = out * torch.sigmoid(gate)                                         # GLU, lid/conformer.py:47-54
    w = sd[p + ".4.conv.weight"]
    k = w.shape[-1]
    pad = (k // 2, k // 2 - (k + 1) % 2)                                  # calc_same_padding :26-28
    c = F.conv1d(F.pad(g, pad), w, sd[p + ".4.conv.bias"], groups=w.shape[0])
    rm, rv = sd[p + ".5.running_mean"], sd[p + ".5.running_var"]
    if opts.training and opts.all_reduce is not None:
        z, rm, rv = SyncBatchNorm.apply(c, sd[p + ".5.weight"], sd[p + ".5.bias"], rm, rv, opts.all_reduce)
        opts.bn_buffers[p + ".5.running_mean"] = rm
        opts.bn_buffers[p + ".5.running_var"] = rv
    elif opts.training:
        rm, rv = rm.clone(), rv.clone()
        z = F.batch_norm(c, rm, rv, sd[p + ".5.weight"], sd[p + ".5.bias"], True, BN_MOMENTUM, BN_EPS)
        opts.bn_buffers[p + ".5.running_mean"] = rm
        opts.bn_buffers[p + ".5.running_var"] = rv
    else:
        z = F.batch_norm(c, rm, rv, sd[p + ".5.weight"], sd[p + ".5.bias"], False, BN_MOMENTUM, BN_EPS)
    s = swish(z)
    return F.conv1d(s, sd[p + ".7.weight"], sd[p + ".7.bias"]).transpose(1, 2)


class SyncBatchNorm(torch.autograd.Function):
    """torch.nn.SyncBatchNorm restated for the CPU oracle (torch's own module is GPU-only): statistics over the rows of ALL
    ranks.  Forward: every rank contributes (sum x, sum x^2, row count) - torch all-gathers per-rank mean/var/count, which
    is the same global mean and biased variance; running_var gets the unbiased N/(N-1) correction with the GLOBAL count N.
    Backward: dx = w*rstd*(dy - mean_N(dy) - xhat*mean_N(dy*xhat)) with both means over all ranks; dweight/dbias stay
    rank-local (DDP averages them like any other gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, all_reduce):
        C = x.shape[1]
        xd = x.double()
        pack = torch.cat([xd.sum((0, 2)), (xd * xd).sum((0, 2)), torch.tensor([x.shape[0] * x.shape[2]], dtype=torch.float64)])
        all_reduce(pack)
        n = float(pack[-1])
        mean = pack[:C] / n
        var = (pack[C:2 * C] / n - mean * mean).clamp_min(0)
        rstd = 1.0 / torch.sqrt(var + BN_EPS)
        xhat = ((xd - mean[None, :, None]) * rstd[None, :, None]).float()
        ctx.save_for_backward(xhat, weight, rstd.float())
        ctx.all_reduce, ctx.n = all_reduce, n
        rm = (1 - BN_MOMENTUM) * running_mean + BN_MOMENTUM * mean.float()
        rv = (1 - BN_MOMENTUM) * running_var + BN_MOMENTUM * (var * (n / max(n - 1, 1))).float()
        ctx.mark_non_differentiable(rm, rv)
        return xhat * weight[None, :, None] + bias[None, :, None], rm, rv

    @staticmethod
    def backward(ctx, dy, _drm, _drv):
        xhat, weight, rstd = ctx.saved_tensors
        C = dy.shape[1]
        pack = torch.cat([dy.double().sum((0, 2)), (dy * xhat).double().sum((0, 2))])
        dbias, dweight = pack[:C].float(), pack[C:].float()
        pack = pack * weight.double().repeat(2)               # sums of dxhat = dy*w and dxhat*xhat
        ctx.all_reduce(pack)
        a, b = (pack[:C] / ctx.n).float(), (pack[C:] / ctx.n).float()
        dx = rstd[None, :, None] * (dy * weight[None, :, None] - a[None, :, None] - xhat * b[None, :, None])
        return dx, dweight, dbias, None, None, None


def conformer_block(x, sd, p, heads, opts: RunOpts):
    """ConformerBlock.forward — lid/conformer.py:252-259."""
    x = 0.5 * feed_forward(x, sd, p + ".ff1.fn") + x
    x = attention(x, sd, p + ".attn", heads) + x
    x = conv_module(x, sd, p + ".conv.net", opts) + x
    x = 0.5 * feed_forward(x, sd, p + ".ff2.fn") + x
    return _ln(x, sd, p + ".post_norm")


def _dropout(x, p, keep_mask, training):
    if not training or p == 0.0:
        return x
    if keep_mask is None:
        keep_mask = torch.rand_like(x) >= p
    return x * keep_mask.to(x.dtype) / (1.0 - p)


def encoder(mel, sd, cfg: ModelCfg, opts: RunOpts, p="model.featurizer"):
    """ConformerModel.forward — lid/conformer.py:445-466 (sub_sampling=2 path)."""
    x = mel.transpose(1, 2)
    x = F.relu(F.conv1d(x, sd[p + ".sub_sampling.sub_sampling.0.weight"],
                        sd[p + ".sub_sampling.sub_sampling.0.bias"], stride=2, padding=1))
    x = F.linear(x.transpose(1, 2), sd[p + ".sub_sampling.linear.weight"], sd[p + ".sub_sampling.linear.bias"])
    x = x * math.sqrt(cfg.encoder_dim)                                    # RelPositionalEncoding :587-590 (Q4)
    x = _dropout(x, cfg.pos_dropout, opts.pos_keep_mask, opts.training)
    for i in range(cfg.n_blocks):
        if opts.training and opts.keep_layers is not None and not opts.keep_layers[i]:
            continue                                                      # stochastic depth (Q5)
        x = conformer_block(x, sd, f"{p}.encoders.{i}", cfg.heads, opts)
    return x


def head(feat, sd, cfg: ModelCfg, lang, opts: RunOpts):
    """ConformerLinear.forward — lid/ConformerLangModel.py:352-356."""
    p = f"model.last_projects.{lang}"
    x = conformer_block(feat, sd, p + ".block", cfg.last_heads, opts)
    x = _dropout(x, cfg.dropout, opts.head_keep_mask, opts.training)
    return F.linear(x, sd[p + ".linear.weight"], sd[p + ".linear.bias"])


def lang_scores(logits: Dict[str, torch.Tensor], cfg: ModelCfg):
    """LangDiscriminator.forward, ASR half — lid/ConformerLangModel.py:383-393."""
    first = next(iter(logits.values()))
    res = torch.zeros(first.shape[0], len(cfg.lang2vocab))
    for lang, x in logits.items():
        vmax, arg = torch.max(torch.log_softmax(x, dim=-1), dim=-1)
        mask = arg != cfg.lang2vocab[lang]
        n = mask.sum(dim=-1)
        res[:, cfg.lang2index[lang]] = (vmax * mask).sum(dim=-1) / (n * math.log(cfg.lang2vocab[lang]) + 1e-5)
    return res


def lang_linear(scores, sd):
    """LangDiscriminator.linear on detached scores — lid/ConformerLangModel.py:374-378,394."""
    h = F.relu(F.linear(scores.detach(), sd["lang_discriminator.linear.0.weight"], sd["lang_discriminator.linear.0.bias"]))
    return F.linear(h, sd["lang_discriminator.linear.2.weight"], sd["lang_discriminator.linear.2.bias"])


def forward(mel, sd, cfg: ModelCfg, lang: Optional[str] = None, opts: Optional[RunOpts] = None):
    """ConformerMutiLangModel.forward at 16 kHz — lid/ConformerLangModel.py:77-83,272-294."""
    opts = opts or RunOpts()
    feat = encoder(mel, sd, cfg, opts)
    if lang is not None:
        return {lang: head(feat, sd, cfg, lang, opts)}, (None, None)
    res = {l: head(feat, sd, cfg, l, opts) for l in cfg.lang2vocab}
    s = lang_scores(res, cfg)
    return res, (s, lang_linear(s, sd))


def ctc_loss(logits, texts, wav_percents, text_percents, blank):
    """LidSuperviseModule.common_loop — lid/LidModule_ASR_Supervised.py:162-168."""
    lp = torch.log_softmax(logits, dim=-1).transpose(1, 0)
    in_len = (logits.shape[1] * wav_percents).long()
    tg_len = (texts.shape[-1] * text_percents).long()
    per = F.ctc_loss(lp, texts, in_len, tg_len, blank=blank, reduction="none", zero_infinity=True)
    return per.mean()


def n_subsampled(n_frames: int) -> int:
    """Conv1d(k3, s2, p1) output length."""
    return (n_frames + 2 - 3) // 2 + 1
