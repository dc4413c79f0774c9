"""Parameter layout of the Conformer-LID model in ONE flat f32 arena.

Names and shapes are the reference's ``state_dict`` keys (lid/conformer.py, lid/ConformerLangModel.py), so
checkpoints interchange.  Laying every parameter, gradient and optimizer moment out in parallel flat arenas (in forward
order) is the MI355X-first part: the optimizer is three launches over the arena, gradient exchange is a handful of large
contiguous collectives issued in reverse layer order, and weights are re-cast to bf16 GEMM operands in one sweep.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

ALIGN = 8   # elements; keeps every tensor 32-byte aligned and K % 8 == 0 for the GEMM operands


@dataclass
class ConformerCfg:
    lang2vocab: Dict[str, int]
    lang2index: Dict[str, int]
    n_blocks: int = 14
    n_mels: int = 80
    encoder_dim: int = 144
    dim_head: int = 64
    heads: int = 4
    ff_mult: int = 4
    conv_expansion_factor: int = 2
    conv_kernel_size: int = 31
    last_heads: int = 8                # ConformerLinear hard-codes heads=8 (lid/ConformerLangModel.py:340)
    last_dim_head: int = 32
    hidden_dim: int = 32
    dropout: float = 0.0               # ConformerLinear.dr
    pos_dropout: float = 0.1           # RelPositionalEncoding(dropout_rate=0.1), lid/conformer.py:426
    max_pos_emb: int = 512
    stochastic_depth_p: float = 0.7
    use_stochastic_depth: bool = True
    # "subsample": log-mel input through Conv1dSubSampling2 (the Conformer LID model).  "features": the input already is the
    # (B, T, d) feature sequence of a frozen backbone (WavLM), no front-end parameters and no encoder blocks of its own
    # (lid/WavLMMutiLangModel.py:185-284: featurizer -> per-language ConformerLinear heads).
    front: str = "subsample"

    @property
    def d(self):
        return self.encoder_dim


@dataclass
class Spec:
    name: str
    shape: Tuple[int, ...]
    kind: str          # 'w' GEMM weight [N,K(,1)] | 'vec' | 'emb' | 'dw' depthwise [C,1,K] | 'conv3' [Co,Ci,3]
    offset: int = 0
    tid: int = 0

    @property
    def numel(self):
        return int(math.prod(self.shape))


def block_specs(p: str, d: int, heads: int, dh: int, ff_mult: int, conv_exp: int, k: int, max_pos: int) -> List[Spec]:
    inner, ff, ci = heads * dh, d * ff_mult, d * conv_exp
    S = Spec
    out = []
    for ffn in ("ff1",):
        out += [S(f"{p}.{ffn}.fn.fn.net.0.weight", (ff, d), "w"), S(f"{p}.{ffn}.fn.fn.net.0.bias", (ff,), "vec"),
                S(f"{p}.{ffn}.fn.fn.net.3.weight", (d, ff), "w"), S(f"{p}.{ffn}.fn.fn.net.3.bias", (d,), "vec"),
                S(f"{p}.{ffn}.fn.norm.weight", (d,), "vec"), S(f"{p}.{ffn}.fn.norm.bias", (d,), "vec")]
    out += [S(f"{p}.attn.fn.to_q.weight", (inner, d), "w"), S(f"{p}.attn.fn.to_kv.weight", (2 * inner, d), "w"),
            S(f"{p}.attn.fn.to_out.weight", (d, inner), "w"), S(f"{p}.attn.fn.to_out.bias", (d,), "vec"),
            S(f"{p}.attn.fn.rel_pos_emb.weight", (2 * max_pos + 1, dh), "emb"),
            S(f"{p}.attn.norm.weight", (d,), "vec"), S(f"{p}.attn.norm.bias", (d,), "vec")]
    out += [S(f"{p}.conv.net.0.weight", (d,), "vec"), S(f"{p}.conv.net.0.bias", (d,), "vec"),
            S(f"{p}.conv.net.2.weight", (2 * ci, d, 1), "w"), S(f"{p}.conv.net.2.bias", (2 * ci,), "vec"),
            S(f"{p}.conv.net.4.conv.weight", (ci, 1, k), "dw"), S(f"{p}.conv.net.4.conv.bias", (ci,), "vec"),
            S(f"{p}.conv.net.5.weight", (ci,), "vec"), S(f"{p}.conv.net.5.bias", (ci,), "vec"),
            S(f"{p}.conv.net.7.weight", (d, ci, 1), "w"), S(f"{p}.conv.net.7.bias", (d,), "vec")]
    ffn = "ff2"
    out += [S(f"{p}.{ffn}.fn.fn.net.0.weight", (ff, d), "w"), S(f"{p}.{ffn}.fn.fn.net.0.bias", (ff,), "vec"),
            S(f"{p}.{ffn}.fn.fn.net.3.weight", (d, ff), "w"), S(f"{p}.{ffn}.fn.fn.net.3.bias", (d,), "vec"),
            S(f"{p}.{ffn}.fn.norm.weight", (d,), "vec"), S(f"{p}.{ffn}.fn.norm.bias", (d,), "vec")]
    out += [S(f"{p}.post_norm.weight", (d,), "vec"), S(f"{p}.post_norm.bias", (d,), "vec")]
    return out


def block_buffers(p: str, d: int, conv_exp: int):
    ci = d * conv_exp
    return [(f"{p}.conv.net.5.running_mean", (ci,), torch.float32), (f"{p}.conv.net.5.running_var", (ci,), torch.float32),
            (f"{p}.conv.net.5.num_batches_tracked", (), torch.int64)]


def model_specs(cfg: ConformerCfg):
    """(specs in arena order, buffers, stage ranges).  Stages: 'front', 'enc.i', 'head.<lang>', 'disc'."""
    d, fz = cfg.d, "model.featurizer"
    specs: List[Spec] = []
    stages: Dict[str, Tuple[int, int]] = {}

    def add(stage, items):
        a = len(specs)
        specs.extend(items)
        stages[stage] = (a, len(specs))

    if cfg.front == "subsample":
        add("front", [Spec(f"{fz}.sub_sampling.sub_sampling.0.weight", (cfg.n_mels, cfg.n_mels, 3), "conv3"),
                      Spec(f"{fz}.sub_sampling.sub_sampling.0.bias", (cfg.n_mels,), "vec"),
                      Spec(f"{fz}.sub_sampling.linear.weight", (d, cfg.n_mels), "w"),
                      Spec(f"{fz}.sub_sampling.linear.bias", (d,), "vec"),
                      Spec(f"{fz}.linear.weight", (d, cfg.n_mels), "w"),          # never used by forward
                      Spec(f"{fz}.linear.bias", (d,), "vec")])
    else:
        stages["front"] = (0, 0)
    buffers = []
    for i in range(cfg.n_blocks):
        p = f"{fz}.encoders.{i}"
        add(f"enc.{i}", block_specs(p, d, cfg.heads, cfg.dim_head, cfg.ff_mult, cfg.conv_expansion_factor,
                                    cfg.conv_kernel_size, cfg.max_pos_emb))
        buffers += block_buffers(p, d, cfg.conv_expansion_factor)
    for lang, v in cfg.lang2vocab.items():
        p = f"model.last_projects.{lang}"
        add(f"head.{lang}", block_specs(p + ".block", d, cfg.last_heads, cfg.last_dim_head, 4, 2, 31, cfg.max_pos_emb)
            + [Spec(p + ".linear.weight", (v + 1, d), "w"), Spec(p + ".linear.bias", (v + 1,), "vec")])
        buffers += block_buffers(p + ".block", d, 2)
    c = len(cfg.lang2vocab)
    add("disc", [Spec("lang_discriminator.linear.0.weight", (cfg.hidden_dim, c), "vec"),
                 Spec("lang_discriminator.linear.0.bias", (cfg.hidden_dim,), "vec"),
                 Spec("lang_discriminator.linear.2.weight", (c, cfg.hidden_dim), "vec"),
                 Spec("lang_discriminator.linear.2.bias", (c,), "vec")])
    off = 0
    for tid, s in enumerate(specs):
        s.offset, s.tid = off, tid
        off += (s.numel + ALIGN - 1) // ALIGN * ALIGN
    return specs, buffers, stages, off


# ------------------------------------------------------------------------------------------------ initialisation
def _kaiming_linear(out_f, in_f, bias=True):
    m = nn.Linear(in_f, out_f, bias=bias)
    return m.weight.data, (m.bias.data if bias else None)


def init_values(cfg: ConformerCfg) -> Dict[str, torch.Tensor]:
    """Fresh parameters drawn by torch.nn's own layer constructors IN THE REFERENCE'S CONSTRUCTION ORDER
    (lid/ConformerLangModel.py:48-75, lid/conformer.py:416-443,226-250), so that under the same torch.manual_seed the
    initial weights equal the reference's.  Returns name -> CPU tensor."""
    d, fz = cfg.d, "model.featurizer"
    out: Dict[str, torch.Tensor] = {}

    def lin(name, o, i, bias=True):
        w, b = _kaiming_linear(o, i, bias)
        out[name + ".weight"] = w
        if bias:
            out[name + ".bias"] = b

    def conv(name, co, ci, k, groups=1):
        m = nn.Conv1d(ci, co, k, groups=groups)
        out[name + ".weight"], out[name + ".bias"] = m.weight.data, m.bias.data

    def ones_zeros(name, n):
        out[name + ".weight"], out[name + ".bias"] = torch.ones(n), torch.zeros(n)

    def block(p, heads, dh, ff_mult, conv_exp, k):
        inner, ff, ci = heads * dh, d * ff_mult, d * conv_exp
        lin(p + ".ff1.fn.fn.net.0", ff, d); lin(p + ".ff1.fn.fn.net.3", d, ff)
        lin(p + ".attn.fn.to_q", inner, d, False); lin(p + ".attn.fn.to_kv", 2 * inner, d, False)
        lin(p + ".attn.fn.to_out", d, inner)
        out[p + ".attn.fn.rel_pos_emb.weight"] = nn.Embedding(2 * cfg.max_pos_emb + 1, dh).weight.data
        ones_zeros(p + ".conv.net.0", d)
        conv(p + ".conv.net.2", 2 * ci, d, 1)
        conv(p + ".conv.net.4.conv", ci, ci, k, groups=ci)
        ones_zeros(p + ".conv.net.5", ci)
        conv(p + ".conv.net.7", d, ci, 1)
        lin(p + ".ff2.fn.fn.net.0", ff, d); lin(p + ".ff2.fn.fn.net.3", d, ff)
        for n in (".attn.norm", ".ff1.fn.norm", ".ff2.fn.norm", ".post_norm"):
            ones_zeros(p + n, d)

    if cfg.front == "subsample":
        conv(fz + ".sub_sampling.sub_sampling.0", cfg.n_mels, cfg.n_mels, 3)
        lin(fz + ".sub_sampling.linear", d, cfg.n_mels)
        lin(fz + ".linear", d, cfg.n_mels)
    for i in range(cfg.n_blocks):
        block(f"{fz}.encoders.{i}", cfg.heads, cfg.dim_head, cfg.ff_mult, cfg.conv_expansion_factor, cfg.conv_kernel_size)
    for lang, v in cfg.lang2vocab.items():
        p = f"model.last_projects.{lang}"
        block(p + ".block", cfg.last_heads, cfg.last_dim_head, 4, 2, 31)
        lin(p + ".linear", v + 1, d)
    c = len(cfg.lang2vocab)
    lin("lang_discriminator.linear.0", cfg.hidden_dim, c)
    lin("lang_discriminator.linear.2", c, cfg.hidden_dim)
    return out
