"""Oracle: WavLM backbone forward + WavLMMutiLangModel heads, functional torch-CPU fp32 restatement (test infrastructure).

Weights are a flat ``{name: tensor}`` dict with the reference's ``WavLM.state_dict()`` names (lid/wavlm/WavLM.py) so a public
WavLM checkpoint drives this code directly.  Pinned by tests/golden/wavlm_fwd.npz / wavlm_model.npz, which
oracle/gen_golden_wavlm.py wrote from the imported reference (tests/test_oracle_wavlm.py); autograd through it is pinned by
wavlm_finetune.npz / wavlm_frozen_masked.npz (the reference's own training-mode gradients).  No dropout; span masks are an input;
like the reference the encoder never sees a padding mask (WavLM.py:390-394).
"""
import math
from typing import Dict, List

import torch
import torch.nn.functional as F

from oracle import conformer as oc


def feature_extractor(wav, sd, spec=((512, 10, 5),) + ((512, 3, 2),) * 4 + ((512, 2, 2),) * 2):
    """ConvFeatureExtractionModel, mode "default", conv_bias False (WavLM.py:409-531): (B, L) -> (B, C, T)."""
    x = wav.unsqueeze(1)
    for i, (_, _, stride) in enumerate(spec):
        x = F.conv1d(x, sd[f"feature_extractor.conv_layers.{i}.0.weight"], stride=stride)
        if i == 0:
            C = x.shape[1]
            x = F.group_norm(x, C, sd["feature_extractor.conv_layers.0.2.weight"], sd["feature_extractor.conv_layers.0.2.bias"], 1e-5)
        x = F.gelu(x)
    return x


def relative_buckets(rel, num_buckets=320, max_distance=800):
    """MultiheadAttention._relative_positions_bucket, bidirectional (modules.py:409-433)."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float().clamp_min(1) / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(n < max_exact, n, large)


def position_bias(sd, T, num_buckets=320, max_distance=800):
    """compute_bias (modules.py:435-446): (H, T, T), owned by layer 0 and shared by every layer."""
    ctx, mem = torch.arange(T)[:, None], torch.arange(T)[None, :]
    emb = sd["encoder.layers.0.self_attn.relative_attention_bias.weight"]
    return emb[relative_buckets(mem - ctx, num_buckets, max_distance)].permute(2, 0, 1)


def attention(x, sd, p, H, pos_bias):
    """MultiheadAttention.forward with gru_rel_pos (modules.py:505-560): x (B, T, d)."""
    B, T, d = x.shape
    dh = d // H
    ql = x.view(B, T, H, dh).permute(0, 2, 1, 3)                                   # the gate reads the LAYER INPUT
    u = F.linear(ql, sd[p + "grep_linear.weight"], sd[p + "grep_linear.bias"]).view(B, H, T, 2, 4).sum(-1)
    ga, gb = torch.sigmoid(u).chunk(2, dim=-1)
    gate = ga * (gb * sd[p + "grep_a"] - 1.0) + 2.0                                 # (B, H, T, 1)
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).view(B, T, H, dh).transpose(1, 2)
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(B, T, H, dh).transpose(1, 2)
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(B, T, H, dh).transpose(1, 2)
    scores = q @ k.transpose(-1, -2) * dh ** -0.5 + gate * pos_bias[None]
    o = (scores.softmax(-1) @ v).transpose(1, 2).reshape(B, T, d)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def backbone(wav, sd, cfg: Dict, taps: Dict = None, masks=None):
    """WavLM.extract_features(source, padding_mask, mask)[0] (WavLM.py:339-406): (B, L) -> (B, T, d).  masks = (time (B, T) bool
    or None, channel (B, d) bool or None): the spans WavLM.apply_mask (WavLM.py:300-337) drew, applied to the projected features -
    masked frames become ``mask_emb``, then masked channels are zeroed over all frames."""
    H, n_layers = cfg.get("encoder_attention_heads", 12), cfg.get("encoder_layers", 12)
    feats = feature_extractor(wav, sd).transpose(1, 2)
    if taps is not None:
        taps["conv"] = feats
    C = feats.shape[-1]
    x = F.layer_norm(feats, (C,), sd["layer_norm.weight"], sd["layer_norm.bias"])
    x = F.linear(x, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    if taps is not None:
        taps["proj"] = x
    if masks is not None:
        tm, cm = masks
        if tm is not None:
            x = torch.where(torch.as_tensor(tm)[:, :, None], sd["mask_emb"].to(x.dtype)[None, None, :], x)
        if cm is not None:
            x = x.masked_fill(torch.as_tensor(cm)[:, None, :], 0.0)
    d = x.shape[-1]
    k, groups = cfg.get("conv_pos", 128), cfg.get("conv_pos_groups", 16)
    w = torch._weight_norm(sd["encoder.pos_conv.0.weight_v"], sd["encoder.pos_conv.0.weight_g"], 2)
    pc = F.conv1d(x.transpose(1, 2), w, sd["encoder.pos_conv.0.bias"], padding=k // 2, groups=groups)
    if k % 2 == 0:
        pc = pc[:, :, :-1]                                                            # SamePad
    x = x + F.gelu(pc).transpose(1, 2)
    x = F.layer_norm(x, (d,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"])
    if taps is not None:
        taps["enc_in"] = x
    pb = position_bias(sd, x.shape[1], cfg.get("num_buckets", 320), cfg.get("max_distance", 800))
    for i in range(n_layers):
        p = f"encoder.layers.{i}."
        x = F.layer_norm(x + attention(x, sd, p + "self_attn.", H, pb), (d,), sd[p + "self_attn_layer_norm.weight"],
                         sd[p + "self_attn_layer_norm.bias"])
        h = F.linear(F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"])), sd[p + "fc2.weight"], sd[p + "fc2.bias"])
        x = F.layer_norm(x + h, (d,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"])
        if taps is not None:
            taps[f"layer{i}"] = x
    return x


def model_forward(wavs: List[torch.Tensor], sd_backbone, sd_heads, wcfg: Dict, hcfg: "oc.ModelCfg", lang=None, opts=None, masks=None):
    """WavLMMutiLangModel.forward at 16 kHz (lid/WavLMMutiLangModel.py:71-77,262-284): backbone -> ConformerLinear heads ->
    LangDiscriminator.  sd_heads uses the reference names model.last_projects.<lang>.* / lang_discriminator.*."""
    opts = opts or oc.RunOpts()
    wav = torch.nn.utils.rnn.pad_sequence(list(wavs), batch_first=True)
    feat = backbone(wav, sd_backbone, wcfg, masks=masks)
    if lang is not None:
        return {lang: oc.head(feat, sd_heads, hcfg, lang, opts)}, (None, None)
    res = {l: oc.head(feat, sd_heads, hcfg, l, opts) for l in hcfg.lang2vocab}
    s = oc.lang_scores(res, hcfg)
    return res, (s, oc.lang_linear(s, sd_heads))
