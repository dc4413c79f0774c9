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


def feature_extractor(wav, sd, spec=((512, 10, 5),) + ((512, 3, 2),) * 4 + ((512, 2, 2),) * 2, mode="default"):
    """ConvFeatureExtractionModel (WavLM.py:409-531): (B, L) -> (B, C, T).  mode "default": GroupNorm(C, C) behind conv layer 0
    only; "layer_norm" (the Large models, WavLM.py:438-450): a LayerNorm over the channels of every time step behind EVERY conv.
    The conv bias is applied when the state dict holds one (conv_bias=True)."""
    x = wav.unsqueeze(1)
    for i, (_, _, stride) in enumerate(spec):
        fe = f"feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, sd[fe + "0.weight"], sd.get(fe + "0.bias"), stride=stride)
        C = x.shape[1]
        if mode == "layer_norm":
            x = F.layer_norm(x.transpose(1, 2), (C,), sd[fe + "2.1.weight"], sd[fe + "2.1.bias"], 1e-5).transpose(1, 2)
        elif i == 0:
            x = F.group_norm(x, C, sd[fe + "2.weight"], sd[fe + "2.bias"], 1e-5)
        x = F.gelu(x)
    return x


def conv_out_lengths(n_samples, spec=((512, 10, 5),) + ((512, 3, 2),) * 4 + ((512, 2, 2),) * 2):
    """Wav2Vec2Model._get_feat_extract_output_lengths (wav2vec2.py:521-539): frames of an utterance of n samples."""
    n = torch.as_tensor(n_samples).float()
    for _, k, s in spec:
        n = torch.floor((n - k) / s + 1)
    return n.long()


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


def attention(x, sd, p, H, pos_bias, key_pad=None):
    """MultiheadAttention.forward (modules.py:505-560): x (B, T, d) = the attention's query / key / value input (the layer input
    of a post-LN layer, LN1's output of a pre-LN one).  pos_bias (H, T, T) with the GRU-style gate (gru_rel_pos) or None
    (wav2vec2); key_pad (B, T) bool: padded keys (wav2vec2 hands its encoder the padding mask) or None."""
    B, T, d = x.shape
    dh = d // H
    q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"]).view(B, T, H, dh).transpose(1, 2)
    k = F.linear(x, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"]).view(B, T, H, dh).transpose(1, 2)
    v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"]).view(B, T, H, dh).transpose(1, 2)
    scores = q @ k.transpose(-1, -2) * dh ** -0.5
    if pos_bias is not None:
        ql = x.view(B, T, H, dh).permute(0, 2, 1, 3)                               # the gate reads the attention's QUERY input
        u = F.linear(ql, sd[p + "grep_linear.weight"], sd[p + "grep_linear.bias"]).view(B, H, T, 2, 4).sum(-1)
        ga, gb = torch.sigmoid(u).chunk(2, dim=-1)
        gate = ga * (gb * sd[p + "grep_a"] - 1.0) + 2.0                             # (B, H, T, 1)
        scores = scores + gate * pos_bias[None]
    if key_pad is not None:
        scores = scores.masked_fill(key_pad[:, None, None, :], float("-inf"))
    o = (scores.softmax(-1) @ v).transpose(1, 2).reshape(B, T, d)
    return F.linear(o, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def backbone(wav, sd, cfg: Dict, taps: Dict = None, masks=None, n_samples=None, states: List = None):
    """WavLM.extract_features(source, padding_mask, mask)[0] (WavLM.py:339-406): (B, L) -> (B, T, d).  masks = (time (B, T) bool
    or None, channel (B, d) bool or None): the spans WavLM.apply_mask (WavLM.py:300-337) drew, applied to the projected features -
    masked frames become ``mask_emb``, then masked channels are zeroed over all frames.

    The Large models' flags (cfg): ``extractor_mode="layer_norm"`` / conv bias (see feature_extractor), ``layer_norm_first``
    (WavLM.py:596-612,727-755: no LayerNorm behind the positional convolution, x += attn(LN1(x)); x += ffn(LN2(x)) per layer, the
    encoder's LayerNorm behind the last layer).  wav2vec2 (cfg ``encoder_padding_mask``, wav2vec2.py:564-583,898-957): with
    n_samples the frames beyond each utterance's conv output length are zeroed in front of the positional convolution and masked
    as attention keys; no relative position bias.  states (a list): receives the hidden states s3prl's hooks collect
    (wav2vec2_expert.py:58-66: every layer's input and the encoder output)."""
    H, n_layers = cfg.get("encoder_attention_heads", 12), cfg.get("encoder_layers", 12)
    pre_ln = bool(cfg.get("layer_norm_first", False))
    feats = feature_extractor(wav, sd, eval(cfg["conv_feature_layers"]) if "conv_feature_layers" in cfg else
                              ((512, 10, 5),) + ((512, 3, 2),) * 4 + ((512, 2, 2),) * 2, cfg.get("extractor_mode", "default")).transpose(1, 2)
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
    key_pad = None
    if cfg.get("encoder_padding_mask", False) and n_samples is not None and min(int(n) for n in n_samples) < wav.shape[1]:
        key_pad = torch.arange(x.shape[1])[None, :] >= conv_out_lengths(list(n_samples))[:, None]
        x = x.masked_fill(key_pad[:, :, None], 0.0)
    k, groups = cfg.get("conv_pos", 128), cfg.get("conv_pos_groups", 16)
    w = torch._weight_norm(sd["encoder.pos_conv.0.weight_v"], sd["encoder.pos_conv.0.weight_g"], 2)
    pc = F.conv1d(x.transpose(1, 2), w, sd["encoder.pos_conv.0.bias"], padding=k // 2, groups=groups)
    if k % 2 == 0:
        pc = pc[:, :, :-1]                                                            # SamePad
    x = x + F.gelu(pc).transpose(1, 2)
    if not pre_ln:
        x = F.layer_norm(x, (d,), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"])
    if taps is not None:
        taps["enc_in"] = x
    pb = position_bias(sd, x.shape[1], cfg.get("num_buckets", 320), cfg.get("max_distance", 800)) \
        if cfg.get("relative_position_embedding", False) else None
    ln = lambda t, n: F.layer_norm(t, (d,), sd[n + ".weight"], sd[n + ".bias"])
    for i in range(n_layers):
        p = f"encoder.layers.{i}."
        if states is not None:
            states.append(x)
        if pre_ln:
            x = x + attention(ln(x, p + "self_attn_layer_norm"), sd, p + "self_attn.", H, pb, key_pad)
            h = ln(x, p + "final_layer_norm")
            x = x + F.linear(F.gelu(F.linear(h, sd[p + "fc1.weight"], sd[p + "fc1.bias"])), sd[p + "fc2.weight"], sd[p + "fc2.bias"])
        else:
            x = ln(x + attention(x, sd, p + "self_attn.", H, pb, key_pad), p + "self_attn_layer_norm")
            h = F.linear(F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"])), sd[p + "fc2.weight"], sd[p + "fc2.bias"])
            x = ln(x + h, p + "final_layer_norm")
        if taps is not None:
            taps[f"layer{i}"] = x
    if pre_ln:
        x = ln(x, "encoder.layer_norm")
    if states is not None:
        states.append(x)
    return x


def wav2vec2_features(wavs: List[torch.Tensor], sd, cfg: Dict, mix=None):
    """The s3prl wav2vec2 upstream + Featurizer as lid/Wav2vecMutiLangModel.py:247-250 drives them: per-utterance layer-norm of the
    waveform when the checkpoint's task says ``normalize`` (wav2vec2_expert.py:71-72), zero padding, the backbone with the frame
    padding mask, then the last hidden state (mix None) or the softmax(mix)-weighted sum of the hidden states
    (interfaces.py:227-252).  -> (B, T, d)."""
    n = [int(w.shape[0]) for w in wavs]
    if cfg.get("normalize", False):
        wavs = [F.layer_norm(w, w.shape) for w in wavs]
    wav = torch.nn.utils.rnn.pad_sequence(list(wavs), batch_first=True)
    states: List = []
    last = backbone(wav, sd, dict(cfg, encoder_padding_mask=True), n_samples=n, states=states)
    if mix is None:
        return last
    sm = torch.softmax(mix, -1)
    return sum(sm[l] * states[l] for l in range(len(states)))


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
