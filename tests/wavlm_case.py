"""WavLM-Base+ width (conv extractor 512 ch, d = 768, 12 heads x 64, FFN 3072, bucketed relative position bias with GRU gating,
pos-conv k128 g16 - the values of the public WavLM-Base+ config) at reduced depth, as a pure function of seeds.
``oracle/gen_golden_wavlm.py`` feeds the REFERENCE's lid/wavlm/WavLM.py + lid/WavLMMutiLangModel.py with it in the build
container and stores stage outputs in tests/golden/wavlm_fwd.npz; the GPU tests run the HIP backbone on the same weights.
The weights (25 M at 2 layers) are never stored: both sides rebuild them from the seed."""
import math

import torch

CFG = dict(encoder_layers=2, encoder_embed_dim=768, encoder_ffn_embed_dim=3072, encoder_attention_heads=12,
           conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2", conv_pos=128, conv_pos_groups=16,
           relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True, layer_norm_first=False,
           extractor_mode="default", conv_bias=False, mask_prob=0.0, mask_channel_prob=0.0)
# fine-tuning fixtures: the same model with every dropout and LayerDrop switched off, so a training-mode step is deterministic
CFG_TRAIN = dict(CFG, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0, dropout_features=0.0,
                 encoder_layerdrop=0.0)
# wav2vec2 Base architecture (fairseq Wav2Vec2Config names; lid/s3prl_updream/wav2vec/wav2vec2.py) at reduced depth: no relative
# position embedding, the encoder receives the padding mask; dropouts off for deterministic training-mode fixtures
W2V_CFG = dict(encoder_layers=2, encoder_embed_dim=768, encoder_ffn_embed_dim=3072, encoder_attention_heads=12,
               conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2", conv_pos=128, conv_pos_groups=16,
               layer_norm_first=False, extractor_mode="default", conv_bias=False, dropout=0.0, attention_dropout=0.0,
               activation_dropout=0.0, dropout_input=0.0, dropout_features=0.0, encoder_layerdrop=0.0)
# wav2vec2 Large / XLS-R 300M architecture (what xlsr2_300m.pt of lid/conf/xf_asr_wav2vec*.yaml holds) at reduced depth:
# layer-norm conv extractor with conv bias, pre-LN layers + the encoder's final LayerNorm, d = 1024, 16 heads, ffn 4096,
# task.normalize (per-utterance layer-norm of the waveform)
XLSR_CFG = dict(encoder_layers=2, encoder_embed_dim=1024, encoder_ffn_embed_dim=4096, encoder_attention_heads=16,
                conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2", conv_pos=128, conv_pos_groups=16,
                layer_norm_first=True, extractor_mode="layer_norm", conv_bias=True, normalize=True, dropout=0.0,
                attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0, dropout_features=0.0, encoder_layerdrop=0.0,
                feature_grad_mult=1.0)
# WavLM Large architecture (WavLM-Large.pt of lid/conf/xf_asr_extra_finetune.yaml:12) at reduced depth: the same extractor without
# conv bias, pre-LN layers, gated bucketed relative position bias; the reference's WavLM wrapper neither normalises the waveform
# nor hands the encoder a padding mask
WAVLM_LARGE_CFG = dict(encoder_layers=2, encoder_embed_dim=1024, encoder_ffn_embed_dim=4096, encoder_attention_heads=16,
                       conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2", conv_pos=128, conv_pos_groups=16,
                       relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True, layer_norm_first=True,
                       extractor_mode="layer_norm", conv_bias=False, normalize=True, dropout=0.0, attention_dropout=0.0,
                       activation_dropout=0.0, dropout_input=0.0, dropout_features=0.0, encoder_layerdrop=0.0, mask_prob=0.0,
                       mask_channel_prob=0.0, feature_grad_mult=1.0)
HEAD_LARGE = dict(dim_head=32, num_head=8, linear_dim=1024, hidden_dim=64)      # lid/conf/xf_asr_wav2vec.yaml:6,9,19-20
MASK_PROB, MASK_CHANNEL_PROB, MASK_SEED = 0.3, 0.2, 5           # the span-masked frozen-regime fixture (numpy seed before the step)
L2V = {"a": 30, "b": 40, "c": 50}
L2I = {"a": 0, "b": 1, "c": 2}
HEAD = dict(dim_head=32, num_head=8, linear_dim=768, hidden_dim=32)
B, SAMPLES = 3, 16000                      # 1 s utterances -> 49 frames


def _seed(name: str) -> int:
    return (sum((i + 1) * ord(c) for i, c in enumerate(name)) * 2654435761) % (2 ** 31)


def backbone_shapes(layers: int, rel_pos: bool = True):
    s = {"mask_emb": (768,), "feature_extractor.conv_layers.0.0.weight": (512, 1, 10),
         "feature_extractor.conv_layers.0.2.weight": (512,), "feature_extractor.conv_layers.0.2.bias": (512,)}
    for i in range(1, 7):
        s[f"feature_extractor.conv_layers.{i}.0.weight"] = (512, 512, 3 if i < 5 else 2)
    s.update({"post_extract_proj.weight": (768, 512), "post_extract_proj.bias": (768,), "encoder.pos_conv.0.bias": (768,),
              "encoder.pos_conv.0.weight_g": (1, 1, 128), "encoder.pos_conv.0.weight_v": (768, 48, 128)})
    for i in range(layers):
        p = f"encoder.layers.{i}."
        if rel_pos:
            s[p + "self_attn.grep_a"] = (1, 12, 1, 1)
        if i == 0 and rel_pos:
            s[p + "self_attn.relative_attention_bias.weight"] = (320, 12)
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"], s[p + f"self_attn.{n}.bias"] = (768, 768), (768,)
        if rel_pos:
            s[p + "self_attn.grep_linear.weight"], s[p + "self_attn.grep_linear.bias"] = (8, 64), (8,)
        s[p + "self_attn_layer_norm.weight"], s[p + "self_attn_layer_norm.bias"] = (768,), (768,)
        s[p + "fc1.weight"], s[p + "fc1.bias"], s[p + "fc2.weight"], s[p + "fc2.bias"] = (3072, 768), (3072,), (768, 3072), (768,)
        s[p + "final_layer_norm.weight"], s[p + "final_layer_norm.bias"] = (768,), (768,)
    s.update({"encoder.layer_norm.weight": (768,), "encoder.layer_norm.bias": (768,), "layer_norm.weight": (512,),
              "layer_norm.bias": (512,)})
    return s


def backbone_weights(layers: int = CFG["encoder_layers"], rel_pos: bool = True):
    """name -> CPU f32 tensor for lid/wavlm/WavLM.py's state_dict: every tensor drawn from its own seeded generator, scaled so
    that activations stay O(1) through the stack (fan-in scaling; norm scales near 1, biases small but non-zero)."""
    out = {}
    for name, shape in backbone_shapes(layers, rel_pos).items():
        g = torch.Generator().manual_seed(_seed(name))
        if name.endswith(("norm.weight", "conv_layers.0.2.weight")):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("grep_a"):
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif name.endswith("weight_g"):
            t = 0.5 + 0.5 * torch.rand(shape, generator=g)
        elif name.endswith("relative_attention_bias.weight"):
            t = 0.5 * torch.randn(shape, generator=g)
        elif len(shape) >= 2:
            fan_in = math.prod(shape[1:])
            t = torch.randn(shape, generator=g) * (1.6 / math.sqrt(fan_in))
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        out[name] = t
    return out


def backbone_weights_cfg(cfg):
    """The same recipe for any backbone config: names and shapes from ``WavLMBackbone.param_shapes`` (the reference's
    ``load_state_dict(strict=True)`` in the fixture generators checks them against lid/wavlm/WavLM.py's own)."""
    from lidk.wavlm import WavLMBackbone
    out = {}
    for name, shape in WavLMBackbone.param_shapes(cfg).items():
        g = torch.Generator().manual_seed(_seed(name))
        if name.endswith(("norm.weight", "conv_layers.0.2.weight", ".2.1.weight")):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("grep_a"):
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif name.endswith("weight_g"):
            t = 0.5 + 0.5 * torch.rand(shape, generator=g)
        elif name.endswith("relative_attention_bias.weight"):
            t = 0.5 * torch.randn(shape, generator=g)
        elif len(shape) >= 2:
            t = torch.randn(shape, generator=g) * (1.6 / math.sqrt(math.prod(shape[1:])))
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        out[name] = t
    return out


def waveforms():
    g = torch.Generator().manual_seed(2024)
    t = torch.arange(SAMPLES) / 16000.0
    x = 0.3 * torch.randn(B, SAMPLES, generator=g)
    for b in range(B):
        x[b] += 0.5 * torch.sin(2 * math.pi * (220.0 * (b + 1)) * t)
    return x


def head_cfg(**kw):
    from lidk.layout import ConformerCfg
    base = dict(lang2vocab=dict(L2V), lang2index=dict(L2I), n_blocks=0, encoder_dim=768, last_heads=HEAD["num_head"],
                last_dim_head=HEAD["dim_head"], hidden_dim=HEAD["hidden_dim"], dropout=0.0, pos_dropout=0.0,
                use_stochastic_depth=False, front="features")
    base.update(kw)
    return ConformerCfg(**base)


def head_weights(**kw):
    """name -> tensor for the per-language ConformerLinear heads + LangDiscriminator of WavLMMutiLangModel (reference state_dict
    names: model.last_projects.<lang>.block.*, .linear.*, lang_discriminator.linear.*), seeded per tensor name; BatchNorm
    running statistics non-trivial."""
    from lidk.layout import model_specs
    specs, buffers, _, _ = model_specs(head_cfg(**kw))
    out = {}
    for s in specs:
        g = torch.Generator().manual_seed(_seed(s.name))
        shape = tuple(s.shape)
        if s.name.endswith(("norm.weight", ".conv.net.0.weight", ".conv.net.5.weight")):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif s.name.endswith("rel_pos_emb.weight"):
            t = 0.3 * torch.randn(shape, generator=g)
        elif len(shape) >= 2:
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(math.prod(shape[1:])))
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        out[s.name] = t
    for name, shape, dt in buffers:
        g = torch.Generator().manual_seed(_seed(name))
        if name.endswith("running_mean"):
            out[name] = 0.2 * torch.randn(shape, generator=g)
        elif name.endswith("running_var"):
            out[name] = 0.5 + torch.rand(shape, generator=g)
        else:
            out[name] = torch.zeros(shape, dtype=dt)
    return out


def texts():
    g = torch.Generator().manual_seed(77)
    return torch.randint(0, 40, (B, 10), generator=g)
