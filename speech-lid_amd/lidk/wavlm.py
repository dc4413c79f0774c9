"""WavLM backbone on the lidk kernels - forward pass (reference lid/wavlm/WavLM.py, lid/wavlm/modules.py).

``WavLMBackbone`` holds the backbone's parameters (names and shapes of the reference ``WavLM.state_dict()``, so public
WavLM-Base(+) checkpoints load as they are), keeps bf16 GEMM operands prepared once, and maps a batch of raw 16 kHz waveforms
``(B, L)`` to the transformer output ``(B, T, d)`` exactly as ``WavLM.extract_features(source, padding_mask, mask=False)``
does (the reference never passes the padding mask into the encoder, WavLM.py:390-394).  No torch layers, no autograd: the
reference trains its LID heads on top of a FROZEN backbone for the first epochs (``freeze_encoder_epoch`` /
``freeze_tranformer_epoch``, lid/LidModule_ASR.py:24-26,243-258), which is the regime this forward-only path serves;
fine-tuning the backbone itself (its backward pass) is not built yet.

MI355X-first design of the pieces:
  * activations are channel-last ``[rows][C]``; the strided Conv1d layers 1-6 of the feature extractor are GEMMs over strided
    VIEWS of the previous layer's output (row t = the kW*C contiguous values from input row stride*t: ``lda = stride*C < K``),
    no im2col, GELU in the epilogue.  Per-utterance row pitches are chosen P_l = 2*P_{l+1} so one launch covers the batch;
  * layer 0 (k10 s5 on the raw waveform + per-channel GroupNorm over time + GELU) is recomputed in two passes instead of
    storing its f32 pre-norm output (lidk_wavlm_conv0);
  * the grouped positional convolution (k128, 16 groups) runs as 16 GEMMs with lda = 48 over a group-major zero-padded copy;
  * q/k/v projections are one fused [3d, d] GEMM; the bucketed relative position bias is a per-head 1-D table over the offset
    j - i (built once from the layer-0 embedding), gated per (batch, head, query) by a GRU-style gate of the layer input and
    added inside the key-tiled attention kernel (lidk_xattn_fwd: any T, nothing T x T in HBM).
"""
import math
from typing import Dict, List

import numpy as np
import torch

from . import _lib as L
from . import ops
from ._lib import LidkError


def conv_out_len(n: int, k: int, s: int) -> int:
    return (n - k) // s + 1


def relative_buckets(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Bucket of a relative position (memory - context), bidirectional (lid/wavlm/modules.py:409-433)."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float().clamp_min(1) / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(is_small, n, large)


def span_mask(shape, padding_mask, mask_prob: float, mask_length: int, min_masks: int = 0, rng=None):
    """Random spans to mask, the "static" overlapping form of the reference's compute_mask_indices (lid/wavlm/WavLM.py:35-158)
    with the same numpy draw order, so a seeded run masks the same positions: one rounding draw for the span count (per row
    when a padding mask is given), ``choice`` of span starts without replacement, spans of ``mask_length``, and every row
    thinned to the shortest row's count.  shape = (rows, size); padding_mask (rows, size) bool tensor or None -> bool ndarray."""
    import numpy as np
    rng = np.random if rng is None else rng
    bsz, all_sz = shape
    mask = np.full((bsz, all_sz), False)
    all_num = max(min_masks, int(mask_prob * all_sz / float(mask_length) + rng.rand()))
    rows = []
    for i in range(bsz):
        if padding_mask is not None:
            sz = all_sz - int(np.asarray(padding_mask[i]).sum())
            num = max(min_masks, int(mask_prob * sz / float(mask_length) + rng.rand()))
        else:
            sz, num = all_sz, all_num
        lengths = np.full(num, mask_length)
        if sum(lengths) == 0:
            lengths[0] = min(mask_length, sz - 1)
        min_len = min(lengths)
        if sz - min_len <= num:
            min_len = sz - num - 1
        starts = rng.choice(sz - min_len, num, replace=False)
        idc = np.asarray([starts[j] + off for j in range(len(starts)) for off in range(lengths[j])])
        rows.append(np.unique(idc[idc < sz]))
    keep = min(len(r) for r in rows)
    for i, r in enumerate(rows):
        if len(r) > keep:
            r = rng.choice(r, keep, replace=False)
        mask[i, r] = True
    return mask


def _same_tree(a, b):
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, dict):
        return isinstance(b, dict) and a.keys() == b.keys() and all(_same_tree(a[k], b[k]) for k in a)
    if isinstance(a, list):
        return isinstance(b, list) and len(a) == len(b) and all(_same_tree(x, y) for x, y in zip(a, b))
    return torch.is_tensor(a) and torch.is_tensor(b) and a.shape == b.shape and a.dtype == b.dtype and a.device == b.device


def _copy_tree(dst, src):
    if isinstance(dst, dict):
        for k in dst:
            _copy_tree(dst[k], src[k])
    elif isinstance(dst, list):
        for x, y in zip(dst, src):
            _copy_tree(x, y)
    elif dst is not None:
        dst.copy_(src)


class WavLMBackbone:
    RB = 2048                       # the bias table covers offsets |j - i| < RB (buckets saturate at max_distance = 800)

    # WavLMConfig's own defaults for the keys a checkpoint cfg may leave out (lid/wavlm/WavLM.py:181-196)
    CFG_DEFAULTS = dict(dropout=0.1, attention_dropout=0.1, activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0,
                        dropout_features=0.0)

    def __init__(self, cfg: Dict):
        self.cfg = {**self.CFG_DEFAULTS, **dict(cfg)}
        self.layers_spec = eval(cfg.get("conv_feature_layers", "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2"))
        self.C = self.layers_spec[0][0]
        if any(c != self.C for c, _, _ in self.layers_spec) or self.layers_spec[0][1:] != (10, 5):
            raise NotImplementedError("lidk WavLM: conv feature extractor of the (C,10,5) + (C,k,s)* form with one width")
        # Base models: extractor_mode "default" (GroupNorm on conv layer 0 only, no conv bias), post-LN transformer layers.
        # Large models (wav2vec2 Large / XLS-R 300M - xlsr2_300m.pt of lid/conf/xf_asr_wav2vec*.yaml - and WavLM Large of
        # lid/conf/xf_asr_extra_finetune.yaml): extractor_mode "layer_norm" (Conv1d(+bias) -> LayerNorm(C) -> GELU in EVERY conv
        # layer, WavLM.py:415-477), layer_norm_first (pre-LN layers + the encoder's final LayerNorm, WavLM.py:596-612,727-755),
        # ``normalize``: the task-level per-utterance layer-norm of the waveform (wav2vec2_expert.py:49,71-72).
        mode = cfg.get("extractor_mode", "default")
        if mode not in ("default", "layer_norm"):
            raise ValueError(f"extractor_mode {mode!r}: 'default' or 'layer_norm' (WavLM.py:420)")
        self.ln_extractor = mode == "layer_norm"
        self.conv_bias = bool(cfg.get("conv_bias", False))
        if self.conv_bias and not self.ln_extractor:
            raise NotImplementedError("lidk backbone: conv_bias with extractor_mode=default (no released checkpoint has it)")
        self.pre_ln = bool(cfg.get("layer_norm_first", False))
        self.normalize = bool(cfg.get("normalize", False))
        if self.ln_extractor and self.C != 512:
            raise NotImplementedError("lidk backbone: the layer_norm feature extractor is built for 512 channels")
        # WavLM: bucketed relative-position bias with GRU gating (every released WavLM checkpoint).  wav2vec2 (SURVEY 8f N2:
        # lid/s3prl_updream/wav2vec/wav2vec2.py): neither, but the encoder receives the key padding mask
        # (``encoder_padding_mask``; the reference's WavLM call leaves it out, WavLM.py:390-394).
        if bool(cfg.get("relative_position_embedding", False)) != bool(cfg.get("gru_rel_pos", False)):
            raise NotImplementedError("lidk backbone: relative_position_embedding and gru_rel_pos come together (WavLM) or not at all "
                                      "(wav2vec2)")
        self.rel_pos = bool(cfg.get("relative_position_embedding", False))
        self.pad_mask = bool(cfg.get("encoder_padding_mask", False))
        self.train_extractor = False              # un-frozen conv feature extractor: the forward keeps its pre-activations
        # dropout decisions are functions of (seed, rank salt, step, site, element index); the Trainer sets seed (the run's seed,
        # equal on all ranks) and rank_salt (_seed_native_generators) and moves ``step`` forward on a resume
        self.seed, self.rank_salt, self.step = 0, 0, 0
        self.forced_keep: Dict = {}               # tests: site -> uint8 keep mask (reference-captured dropout masks)
        self.d = cfg.get("encoder_embed_dim", 768)
        self.ffn = cfg.get("encoder_ffn_embed_dim", 3072)
        self.H = cfg.get("encoder_attention_heads", 12)
        self.dh = self.d // self.H
        self.n_layers = cfg.get("encoder_layers", 12)
        self.kpos, self.gpos = cfg.get("conv_pos", 128), cfg.get("conv_pos_groups", 16)
        self.params: Dict[str, torch.Tensor] = {}
        self.device = torch.device("cpu")
        self._prepared = False
        self._ws: Dict[tuple, dict] = {}
        import os
        from .engine import _GraphCache
        # the forward is a fixed launch sequence per (batch, samples) shape over static buffers: replayed from hipGraphs
        # (first use of a shape runs eagerly, the second is captured), like the engine's block sequences
        self.graphs = _GraphCache(os.environ.get("LIDK_GRAPHS", "1") != "0")

    # ------------------------------------------------------------------ parameters
    @staticmethod
    def param_shapes(cfg: Dict) -> Dict[str, tuple]:
        """Names and shapes of lid/wavlm/WavLM.py's ``state_dict()`` for a config (either extractor mode, with or without conv
        bias; the relative position embedding is owned by layer 0)."""
        spec = eval(cfg.get("conv_feature_layers", "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2"))
        C, d = spec[0][0], cfg.get("encoder_embed_dim", 768)
        ffn, H = cfg.get("encoder_ffn_embed_dim", 3072), cfg.get("encoder_attention_heads", 12)
        kpos, gpos = cfg.get("conv_pos", 128), cfg.get("conv_pos_groups", 16)
        ln_mode, bias = cfg.get("extractor_mode", "default") == "layer_norm", bool(cfg.get("conv_bias", False))
        s = {"mask_emb": (d,)}
        for i in range(len(spec)):
            fe = f"feature_extractor.conv_layers.{i}."
            s[fe + "0.weight"] = (C, 1 if i == 0 else C, spec[i][1])
            if bias:
                s[fe + "0.bias"] = (C,)
            if ln_mode:                              # Sequential(conv, Dropout, Sequential(TransposeLast, LayerNorm, TransposeLast), GELU)
                s[fe + "2.1.weight"], s[fe + "2.1.bias"] = (C,), (C,)
            elif i == 0:                             # Sequential(conv, Dropout, GroupNorm, GELU)
                s[fe + "2.weight"], s[fe + "2.bias"] = (C,), (C,)
        s.update({"post_extract_proj.weight": (d, C), "post_extract_proj.bias": (d,), "encoder.pos_conv.0.bias": (d,),
                  "encoder.pos_conv.0.weight_g": (1, 1, kpos), "encoder.pos_conv.0.weight_v": (d, d // gpos, kpos)})
        for i in range(cfg.get("encoder_layers", 12)):
            p = f"encoder.layers.{i}."
            if cfg.get("relative_position_embedding", False):
                s[p + "self_attn.grep_a"] = (1, H, 1, 1)
                if i == 0:
                    s[p + "self_attn.relative_attention_bias.weight"] = (cfg.get("num_buckets", 320), H)
            for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
                s[p + f"self_attn.{n}.weight"], s[p + f"self_attn.{n}.bias"] = (d, d), (d,)
            if cfg.get("relative_position_embedding", False):
                s[p + "self_attn.grep_linear.weight"], s[p + "self_attn.grep_linear.bias"] = (8, d // H), (8,)
            s[p + "self_attn_layer_norm.weight"], s[p + "self_attn_layer_norm.bias"] = (d,), (d,)
            s[p + "fc1.weight"], s[p + "fc1.bias"], s[p + "fc2.weight"], s[p + "fc2.bias"] = (ffn, d), (ffn,), (d, ffn), (d,)
            s[p + "final_layer_norm.weight"], s[p + "final_layer_norm.bias"] = (d,), (d,)
        s.update({"encoder.layer_norm.weight": (d,), "encoder.layer_norm.bias": (d,), "layer_norm.weight": (C,),
                  "layer_norm.bias": (C,)})
        return s

    def load_state_dict(self, sd: Dict[str, torch.Tensor], share: bool = False):
        """``share``: keep references to the caller's f32 tensors (the model's nn.Parameters) instead of copies, so that an
        optimizer's in-place updates reach the next ``refresh()``."""
        self.params = {k: (v.detach() if share and v.dtype == torch.float32 else v.detach().clone().float()) for k, v in sd.items()}
        self._prepared = False

    def state_dict(self):
        return dict(self.params)

    def to(self, device):
        self.device = torch.device(device)
        self._prepared = False
        self._ws.clear()
        self.graphs.clear()
        return self

    def _prepare(self):
        if self.device.type != "cuda":
            raise LidkError("WavLMBackbone: the HIP path runs on the GPU only (no CPU fallback)")
        p, dev, bf = self.params, self.device, torch.bfloat16
        g = lambda n: p[n].to(dev).contiguous()
        fe = "feature_extractor.conv_layers."
        W = {"conv0_w": g(fe + "0.0.weight").reshape(self.C, 10), "conv": []}
        n_conv = len(self.layers_spec)
        if self.ln_extractor:
            W["xln_w"] = [g(f"{fe}{i}.2.1.weight") for i in range(n_conv)]
            W["xln_b"] = [g(f"{fe}{i}.2.1.bias") for i in range(n_conv)]
            W["conv_b"] = [g(f"{fe}{i}.0.bias") if self.conv_bias else None for i in range(n_conv)]
        else:
            W["gn_w"], W["gn_b"] = g(fe + "0.2.weight"), g(fe + "0.2.bias")
        for i in range(1, len(self.layers_spec)):
            w = g(f"{fe}{i}.0.weight")                                        # [Co][Ci][kW] -> [Co][kW*Ci], k-major
            W["conv"].append(w.permute(0, 2, 1).reshape(w.shape[0], -1).to(bf).contiguous())
        W["convT"] = [w_.t().contiguous() for w_ in W["conv"]]                   # [kW*Ci][Co]: data gradient of the conv stack
        W["ln0_w"], W["ln0_b"] = g("layer_norm.weight"), g("layer_norm.bias")
        W["proj_w"], W["proj_b"] = g("post_extract_proj.weight").to(bf), g("post_extract_proj.bias")
        wv, wg = g("encoder.pos_conv.0.weight_v"), g("encoder.pos_conv.0.weight_g")
        w = torch._weight_norm(wv, wg, 2)                                       # weight_norm(dim=2): per kernel position
        cg = self.d // self.gpos
        W["pos_w"] = [w[i * cg:(i + 1) * cg].permute(0, 2, 1).reshape(cg, -1).to(bf).contiguous() for i in range(self.gpos)]
        W["pos_b"] = g("encoder.pos_conv.0.bias")
        W["enc_ln_w"], W["enc_ln_b"] = g("encoder.layer_norm.weight"), g("encoder.layer_norm.bias")
        if self.rel_pos:
            emb = g("encoder.layers.0.self_attn.relative_attention_bias.weight")      # [buckets][H], shared by every layer
            r = torch.arange(-(self.RB - 1), self.RB, device=dev)
            bucket = relative_buckets(r, self.cfg.get("num_buckets", 320), self.cfg.get("max_distance", 800))
            W["rb"] = emb[bucket].t().contiguous()                                   # [H][2*RB-1], entry r + RB - 1
            W["rb_bucket"] = bucket
        W["mask_emb"] = g("mask_emb")
        W["proj_wT"] = W["proj_w"].t().contiguous()                              # data gradient of post_extract_proj
        # pos-conv data gradient = the same strided-view GEMM with the kernel flipped: [c][j'*cg + o] = w[o][c][k-1-j']
        W["pos_wd"] = [w[i * cg:(i + 1) * cg].flip(2).permute(1, 2, 0).reshape(cg, -1).to(bf).contiguous() for i in range(self.gpos)]
        W["layers"] = []
        for i in range(self.n_layers):
            q = f"encoder.layers.{i}."
            a = q + "self_attn."
            W["layers"].append(dict(
                wqkv=torch.cat([g(a + "q_proj.weight"), g(a + "k_proj.weight"), g(a + "v_proj.weight")]).to(bf).contiguous(),
                bqkv=torch.cat([g(a + "q_proj.bias"), g(a + "k_proj.bias"), g(a + "v_proj.bias")]).contiguous(),
                wo=g(a + "out_proj.weight").to(bf), bo=g(a + "out_proj.bias"),
                **(dict(wg=g(a + "grep_linear.weight"), bg=g(a + "grep_linear.bias"),
                        grep_a=g(a + "grep_a").reshape(-1).contiguous()) if self.rel_pos else {}),
                ln1_w=g(q + "self_attn_layer_norm.weight"), ln1_b=g(q + "self_attn_layer_norm.bias"),
                w1=g(q + "fc1.weight").to(bf), b1=g(q + "fc1.bias"), w2=g(q + "fc2.weight").to(bf), b2=g(q + "fc2.bias"),
                ln2_w=g(q + "final_layer_norm.weight"), ln2_b=g(q + "final_layer_norm.bias")))
        for Lw_ in W["layers"]:                       # W^T copies: the data gradients are NT GEMMs on them (un-frozen encoder)
            for n in ("wqkv", "wo", "w1", "w2"):
                Lw_[n + "T"] = Lw_[n].t().contiguous()
        old = getattr(self, "W", None)
        if old is not None and _same_tree(old, W):    # refresh: same addresses, so captured graphs and views stay valid
            _copy_tree(old, W)
        else:
            self.W = W
            self.graphs.clear()
        self._prepared = True

    def _refresh_inplace(self):
        """The operands of _prepare written straight into the existing buffers (same addresses: captured graphs and views stay
        valid): one converting / transposing copy per operand instead of build-new + copy-over.  1-D parameters (biases, LayerNorm
        weights, mask_emb) ARE the parameter tensors when those already live on the device, so they need nothing."""
        p, W, bf = self.params, self.W, torch.bfloat16
        fe = "feature_extractor.conv_layers."

        def alias(dst, name):                                  # f32 operand kept as a copy only if _prepare had to make one
            if dst.data_ptr() != p[name].data_ptr():
                dst.copy_(p[name].reshape(dst.shape), non_blocking=True)

        alias(W["conv0_w"], fe + "0.0.weight")
        if self.ln_extractor:
            for i in range(len(self.layers_spec)):
                alias(W["xln_w"][i], f"{fe}{i}.2.1.weight"); alias(W["xln_b"][i], f"{fe}{i}.2.1.bias")
                if self.conv_bias:
                    alias(W["conv_b"][i], f"{fe}{i}.0.bias")
        else:
            alias(W["gn_w"], fe + "0.2.weight"); alias(W["gn_b"], fe + "0.2.bias")
        for i in range(1, len(self.layers_spec)):
            w = p[f"{fe}{i}.0.weight"]                                           # [Co][Ci][kW] -> [Co][kW][Ci]
            W["conv"][i - 1].view(w.shape[0], w.shape[2], w.shape[1]).copy_(w.permute(0, 2, 1))
            W["convT"][i - 1].copy_(W["conv"][i - 1].t())
        alias(W["ln0_w"], "layer_norm.weight"); alias(W["ln0_b"], "layer_norm.bias")
        W["proj_w"].copy_(p["post_extract_proj.weight"]); alias(W["proj_b"], "post_extract_proj.bias")
        W["proj_wT"].copy_(W["proj_w"].t())
        w = torch._weight_norm(p["encoder.pos_conv.0.weight_v"], p["encoder.pos_conv.0.weight_g"], 2)
        cg = self.d // self.gpos
        for i in range(self.gpos):
            blk = w[i * cg:(i + 1) * cg]                                           # [cg][cg][k]
            W["pos_w"][i].view(cg, blk.shape[2], cg).copy_(blk.permute(0, 2, 1))
            W["pos_wd"][i].view(cg, blk.shape[2], cg).copy_(blk.flip(2).permute(1, 2, 0))
        alias(W["pos_b"], "encoder.pos_conv.0.bias")
        alias(W["enc_ln_w"], "encoder.layer_norm.weight"); alias(W["enc_ln_b"], "encoder.layer_norm.bias")
        if self.rel_pos:
            emb = p["encoder.layers.0.self_attn.relative_attention_bias.weight"]
            W["rb"].copy_(emb[W["rb_bucket"]].t())
        alias(W["mask_emb"], "mask_emb")
        d = self.d
        for i, Lw_ in enumerate(W["layers"]):
            q = f"encoder.layers.{i}."
            a = q + "self_attn."
            # the four Linears of the layer: f32 parameter -> bf16 operand + its transpose (+ the packed q | k | v bias) in ONE
            # launch (lidk_cast_transpose_grouped) instead of 14 torch copies, 4 of them transposing (XLS-R: ~310 -> ~45 us per layer)
            lin = [p[a + n + s_] for n in ("q_proj", "k_proj", "v_proj", "out_proj") for s_ in (".weight", ".bias")]
            lin += [p[q + "fc1.weight"], p[q + "fc2.weight"]]
            if all(t.dtype == torch.float32 and t.is_contiguous() for t in lin):
                ops.cast_transpose_grouped(self._layer_refresh_group(i, Lw_))
            else:                                                  # parameters held in another dtype: plain converting copies
                for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                    Lw_["wqkv"][j * d:(j + 1) * d].copy_(p[a + n + ".weight"])
                    Lw_["bqkv"][j * d:(j + 1) * d].copy_(p[a + n + ".bias"])
                Lw_["wo"].copy_(p[a + "out_proj.weight"])
                Lw_["w1"].copy_(p[q + "fc1.weight"])
                Lw_["w2"].copy_(p[q + "fc2.weight"])
                for n in ("wqkv", "wo", "w1", "w2"):
                    Lw_[n + "T"].copy_(Lw_[n].t())
            alias(Lw_["bo"], a + "out_proj.bias")
            if self.rel_pos:
                alias(Lw_["wg"], a + "grep_linear.weight"); alias(Lw_["bg"], a + "grep_linear.bias"); alias(Lw_["grep_a"], a + "grep_a")
            alias(Lw_["ln1_w"], q + "self_attn_layer_norm.weight"); alias(Lw_["ln1_b"], q + "self_attn_layer_norm.bias")
            alias(Lw_["b1"], q + "fc1.bias")
            alias(Lw_["b2"], q + "fc2.bias")
            alias(Lw_["ln2_w"], q + "final_layer_norm.weight"); alias(Lw_["ln2_b"], q + "final_layer_norm.bias")
        self._prepared = True

    def _layer_refresh_group(self, i, Lw_):
        """Descriptor table (device) of layer i's operand refresh; rebuilt when a parameter or operand tensor has moved."""
        p, d = self.params, self.d
        q = f"encoder.layers.{i}."
        a = q + "self_attn."
        ent = []
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            ent.append((p[a + n + ".weight"], Lw_["wqkv"][j * d:(j + 1) * d], Lw_["wqkvT"][:, j * d:(j + 1) * d], None))
            ent.append((p[a + n + ".bias"], None, None, Lw_["bqkv"][j * d:(j + 1) * d]))
        ent.append((p[a + "out_proj.weight"], Lw_["wo"], Lw_["woT"], None))
        ent.append((p[q + "fc1.weight"], Lw_["w1"], Lw_["w1T"], None))
        ent.append((p[q + "fc2.weight"], Lw_["w2"], Lw_["w2T"], None))
        key = tuple(t.data_ptr() for e in ent for t in e if t is not None)
        cache = self.__dict__.setdefault("_refresh_groups", {})
        hit = cache.get(i)
        if hit is None or hit[0] != key:
            hit = (key, ops.build_cast_transpose_group(ent))
            cache[i] = hit
        return hit[1]

    @property
    def _inplace_ok(self) -> bool:
        """All parameters already live on the kernels' device (the in-place refresh reads them where they are)."""
        return all(t.device == self.device for t in self.params.values())

    # ------------------------------------------------------------------ geometry
    def frame_counts(self, n_samples: int) -> List[int]:
        t, out = n_samples, []
        for _, k, s in self.layers_spec:
            t = conv_out_len(t, k, s)
            out.append(t)
        return out

    def _workspace(self, B: int, Lw: int):
        key = (B, Lw)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        T = self.frame_counts(Lw)
        if T[-1] < 1:
            raise LidkError(f"WavLM: {Lw} samples are too few for the feature extractor")
        lim = min(ops.xattn_max_frames(self.dh), self.RB - 1)
        if T[-1] > lim:
            raise LidkError(f"WavLM: {T[-1]} frames ({Lw / 16000:.1f} s) exceed the attention kernel's limit of "
                            f"{lim} frames (dh = {self.dh}): lower data.max_duration")
        n = len(T)
        # row pitches with P_l = stride_{l+1} * P_{l+1}, large enough for every layer's valid rows
        mult = [1] * n
        for l in range(n - 2, -1, -1):
            mult[l] = mult[l + 1] * self.layers_spec[l + 1][2]
        P_last = max(-(-T[l] // mult[l]) for l in range(n))
        P = [P_last * m for m in mult]
        dev, bf = self.device, torch.bfloat16
        # layer_norm extractor: every layer's output is bf16 (the row LayerNorm + GELU kernel keeps the operand type); default
        # mode: the last layer's GEMM writes f32 straight away
        bufs = [torch.zeros(B * P[l] + 8, self.C, device=dev, dtype=(bf if l < n - 1 or self.ln_extractor else torch.float32))
                for l in range(n)]
        Tn, M, d = T[-1], B * T[-1], self.d
        Pp = Tn + self.kpos
        ws = dict(T=T, P=P, bufs=bufs, M=M, Tn=Tn, Pp=Pp, wav=torch.empty(B, Lw, device=dev),
                  c0ws=None if self.ln_extractor else torch.empty(ops.wavlm_conv0_workspace(B, T[0], self.C), device=dev),
                  # layer_norm extractor: a conv layer's pre-norm GEMM output (one scratch for all layers while the extractor is
                  # frozen; per-layer buffers when it trains, _extractor_buffers)
                  xpre=torch.zeros(B * P[1] + 8, self.C, device=dev, dtype=bf) if self.ln_extractor and n > 1 else None,
                  nsamp=torch.full((B,), Lw, device=dev, dtype=torch.int32),
                  xf=torch.empty(M, d, device=dev) if self.pre_ln else None,
                  tmask=torch.zeros(B, Tn, device=dev, dtype=torch.uint8), cmask=torch.zeros(B, d, device=dev, dtype=torch.uint8),
                  xc=torch.empty(M, self.C, device=dev), mean_in=torch.empty(M, device=dev), rstd_in=torch.empty(M, device=dev), h0=torch.empty(M, self.C, device=dev, dtype=bf),
                  x=torch.empty(M, d, device=dev), xb=torch.empty(M, d, device=dev, dtype=bf),
                  x1=torch.empty(M, d, device=dev), x1b=torch.empty(M, d, device=dev, dtype=bf), y=torch.empty(M, d, device=dev),
                  xg=torch.empty(self.gpos, B * Pp + self.kpos, d // self.gpos, device=dev, dtype=bf),
                  pc=torch.empty(B * Pp, d, device=dev), qkv=torch.empty(M, 3 * d, device=dev, dtype=bf),
                  o=torch.empty(M, d, device=dev, dtype=bf), hm=torch.empty(M, self.ffn, device=dev, dtype=bf),
                  gate=torch.empty(B, self.H, Tn, device=dev), lse=torch.empty(B, self.H, Tn, device=dev),
                  klen=torch.full((B,), Tn, device=dev, dtype=torch.int32), feat=torch.empty(M, d, device=dev),
                  tmp=torch.empty(M, d, device=dev))
        if len(self._ws) > 8:
            self._ws.clear()
            self.graphs.clear()
        self._ws[key] = ws
        return ws

    def _train_buffers(self, ws, B):
        """Per-layer saved activations + backward scratch of one input shape (allocated on the first training forward)."""
        if "save" in ws:
            return ws["save"]
        dev, bf, d, M, Tn, H = self.device, torch.bfloat16, self.d, ws["M"], ws["Tn"], self.H
        e = lambda *sh: torch.empty(*sh, device=dev, dtype=bf)
        f = lambda *sh: torch.empty(*sh, device=dev)
        layers = []
        for _ in range(self.n_layers):
            # post-LN: xin / xinb = the layer input (a LayerNorm output), y1 / y2 = the sums in front of the two LayerNorms, x1 /
            # x1b = LN1's output.  pre-LN (layer_norm_first): xin = the residual stream, xinb = LN1(xin), y1 = xin + attention,
            # x1b = LN2(y1); x1 (LN1's f32 output) is only the relative-bias gate's input, y2 is the next layer's xin.
            layers.append(dict(xin=f(M, d), xinb=e(M, d), qkv=e(M, 3 * d), gate=f(B, H, Tn), lse=f(B, H, Tn), o=e(M, d),
                               y1=f(M, d), mean1=f(M), rstd1=f(M), x1=f(M, d) if (not self.pre_ln or self.rel_pos) else None,
                               x1b=e(M, d), pre=e(M, self.ffn), hm=e(M, self.ffn),
                               y2=None if self.pre_ln else f(M, d), mean2=f(M), rstd2=f(M)))
        Pp = ws["Pp"]
        sv = dict(layers=layers, y0=f(M, d), mean0=f(M), rstd0=f(M), pcp=e(B * Pp, d), meanF=f(M), rstdF=f(M),
                  dxa=f(M, d), dxb=f(M, d), da=f(M, d), dab=e(M, d), dpre=e(M, self.ffn), dx1=f(M, d), db=f(M, d), dbb=e(M, d),
                  do=e(M, d), dqkv=e(M, 3 * d), dgate=f(B, H, Tn), delta=f(B, H, Tn), partial=f(L.LN_BWD_BLOCKS * 2 * d),
                  dpc=torch.zeros(B * Pp + self.kpos, d, device=dev, dtype=bf), dy0=f(M, d),
                  dpg=torch.zeros(self.gpos, B * Pp + self.kpos, d // self.gpos, device=dev, dtype=bf), dxp=f(B * Pp, d),
                  dxc=f(M, self.C), dots=f(self.n_layers + 1), dm=e(M, d))
        ws["save"] = sv
        return sv

    def _key_lengths(self, ws, B, Tn, Lw, n_samples):
        """wav2vec2 (encoder_padding_mask): frames of utterance b beyond the conv stack's output length of its TRUE sample count are
        padding (Wav2Vec2Model._get_feat_extract_output_lengths, wav2vec2.py:521-539,564-583) -> ws["klen"] on the device and the
        host list; None when nothing is padded (the reference then passes no mask at all)."""
        if not self.pad_mask or n_samples is None or all(int(n) >= Lw for n in n_samples):
            ws["klen_host"] = None
            return None
        lens = [min(max(self.frame_counts(int(n))[-1], 1), Tn) for n in n_samples]
        ws["klen_host"] = lens
        return self._upload_mask(ws, "klen", np.asarray(lens, dtype=np.int32))

    def _apply_mask(self, ws, B, Tn, Lw, n_samples):
        cfg, dev = self.cfg, self.device
        pad = None
        if self.pad_mask:                 # wav2vec2: the frame mask of the conv-length formula, None without padding
            lens = ws.get("klen_host")
            if lens is not None:
                pad = np.arange(Tn)[None, :] >= np.asarray(lens)[:, None]
        elif n_samples is not None:     # the model surface ALWAYS hands WavLM a padding mask (lid/WavLMMutiLangModel.py:271-274),
            # even when nothing is padded - which changes compute_mask_indices' draw order (one rounding draw per row).
            # WavLM.forward_padding_mask: a frame is padding if all of its Lw // T samples are (remainder samples dropped), i.e.
            # frame t of an utterance of n samples is padding iff t * per >= n.  (Closed form: no (B, Lw) host tensor per step.)
            per = Lw // Tn
            first_pad = -(-np.asarray(n_samples, dtype=np.int64) // per)
            pad = np.arange(Tn)[None, :] >= first_pad[:, None]
        tmask = cmask = None
        if cfg.get("mask_prob", 0.0) > 0:
            m = span_mask((B, Tn), pad, cfg["mask_prob"], cfg.get("mask_length", 10), min_masks=2)
            tmask = self._upload_mask(ws, "tmask", m)
        if cfg.get("mask_channel_prob", 0.0) > 0:
            m = span_mask((B, self.d), None, cfg["mask_channel_prob"], cfg.get("mask_channel_length", 10))
            cmask = self._upload_mask(ws, "cmask", m)
        ops.wavlm_apply_mask(ws["x"], tmask, cmask, self.W["mask_emb"], B, Tn)
        ws["masked"] = (tmask, cmask)

    @staticmethod
    def _upload_mask(ws, name, m):
        """Host-drawn spans -> the device mask through one of two PINNED staging buffers (a copy from pageable memory would
        block the host until the stream drains); a buffer is reused two steps later, after its copy's event has passed."""
        dst = ws[name]
        st = ws.setdefault(name + "_stage", dict(bufs=[torch.empty(dst.shape, dtype=dst.dtype).pin_memory() for _ in range(2)],
                                                 events=[None, None], turn=0))
        i = st["turn"]
        st["turn"] = 1 - i
        if st["events"][i] is not None:
            st["events"][i].synchronize()
        st["bufs"][i].numpy()[...] = m.view("uint8") if dst.dtype == torch.uint8 else m
        dst.copy_(st["bufs"][i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["events"][i] = ev
        return dst

    def _extractor_buffers(self, ws, B):
        """Pre-activations of conv layers 1.. (kept by the forward while the extractor trains) and the backward's scratch."""
        ex = ws.get("ex")
        if ex is None:
            dev, bf, C, P, n = self.device, torch.bfloat16, self.C, ws["P"], len(ws["P"])
            kmax = max(kw for _, kw, _ in self.layers_spec[1:])
            ex = ws["ex"] = dict(
                pre=[None] + [torch.zeros(B * P[l] + 8, C, device=dev, dtype=bf) for l in range(1, n)],
                d=[torch.zeros(B * P[l] + 8, C, device=dev, dtype=bf) for l in range(n)],
                dcol=torch.empty(B * P[1], kmax * C, device=dev, dtype=bf),
                sums=None if self.ln_extractor else torch.empty(B * C * 2, device=dev),
                gconv=[None] + [torch.zeros(C, self.layers_spec[l][1] * C, device=dev) for l in range(1, n)],
                gconv0=torch.zeros(C, self.layers_spec[0][1], device=dev))
        return ex

    def _fwd_pre(self, ws, B, taps, keep_pre=False):
        """Feature extractor, LayerNorm, post_extract_proj -> ws["x"] (B*T, d) f32.  keep_pre: the GELU epilogues also write
        the pre-activations (the extractor's backward needs gelu')."""
        W, k, wav = self.W, ops, ws["wav"]
        T, P, bufs, C, d = ws["T"], ws["P"], ws["bufs"], self.C, self.d
        ex = self._extractor_buffers(ws, B) if keep_pre else None
        if self.ln_extractor:
            # every layer: Conv1d(+bias) -> LayerNorm over channels -> GELU (WavLM.py:438-450).  Layer 0 in one kernel; layers 1..:
            # the strided-view GEMM (bias in its epilogue) writes the pre-norm rows, lidk_ln_gelu_fwd the layer output
            k.conv0_ln_fwd(wav, W["conv0_w"], W["conv_b"][0], W["xln_w"][0], W["xln_b"][0], bufs[0], T[0], P[0])
            for l in range(1, len(T)):
                _, kw, st = self.layers_spec[l]
                rows = B * P[l]
                A = bufs[l - 1].as_strided((rows, kw * C), (st * C, 1))
                pre = (ex["pre"][l] if ex else ws["xpre"])[:rows]
                k.gemm_nt(A, W["conv"][l - 1], pre, bias=W["conv_b"][l])
                k.ln_gelu_fwd(pre, W["xln_w"][l], W["xln_b"][l], bufs[l][:rows])
        else:
            k.wavlm_conv0(wav, W["conv0_w"], W["gn_w"], W["gn_b"], bufs[0], T[0], P[0], workspace=ws["c0ws"])
            for l in range(1, len(T)):
                _, kw, st = self.layers_spec[l]
                A = bufs[l - 1].as_strided((B * P[l], kw * C), (st * C, 1))          # strided view: the convolution is this GEMM
                k.gemm_nt(A, W["conv"][l - 1], bufs[l][:B * P[l]], act=L.ACT_GELU, out2=ex["pre"][l][:B * P[l]] if ex else None)
        Tn, M, Pp = ws["Tn"], ws["M"], ws["Pp"]
        last = bufs[-1][:B * P[-1]]
        k.scale_cast_2d(last.view(B, P[-1] * C), ws["xc"].view(B, Tn * C), B, Tn * C)       # drop the pitch padding rows
        if taps is not None:
            taps["conv"] = ws["xc"].view(B, Tn, C).clone()
        k.layernorm_fwd(ws["xc"], W["ln0_w"], W["ln0_b"], yT=ws["h0"], mean=ws["mean_in"], rstd=ws["rstd_in"])
        k.gemm_nt(ws["h0"], W["proj_w"], ws["x"], bias=W["proj_b"])
        if taps is not None:
            taps["proj"] = ws["x"].view(B, Tn, d).clone()

    def _attn_bias(self, gate, active=None):
        """(gate, rb) of the gated relative-position bias, or (None, None): wav2vec2 has none, and in WavLM only layer 0 owns
        the bucket embedding (modules.py:500), so when LayerDrop skips layer 0 the layers that do run get ``position_bias``
        None - no bias and no gate for that step (WavLM.py:620-631, modules.py:516-531)."""
        if not self.rel_pos or (active is not None and 0 not in active):
            return None, None
        return gate, self.W["rb"]

    def _fwd_post(self, ws, B, taps, save=None, klen=None, mix_w=None):
        """Positional convolution + LayerNorm, transformer layers -> ws["x"] (and, with ``mix_w``, the softmax-weighted sum of
        the hidden states -> ws["feat"]).  ``save``: the training buffers (activations kept for ``backward``); None on the
        inference / frozen path.  klen: device int32 (B,) key lengths (wav2vec2 padding mask) or None."""
        if save is not None:
            return self._fwd_post_train(ws, B, save, klen, mix_w)
        W, k = self.W, ops
        Tn, M, Pp, d = ws["Tn"], ws["M"], ws["Pp"], self.d
        if klen is not None:
            k.zero_padded_rows(ws["x"], klen, B, Tn)
        # positional convolution + residual + LayerNorm
        cg = d // self.gpos
        k.wavlm_posconv_prep(ws["x"], ws["xg"], B, Tn, self.gpos, Pp, self.kpos // 2)
        for g in range(self.gpos):
            A = ws["xg"][g].as_strided((B * Pp, self.kpos * cg), (cg, 1))
            k.gemm_nt(A, W["pos_w"][g], ws["pc"][:, g * cg:(g + 1) * cg], bias=W["pos_b"][g * cg:(g + 1) * cg].contiguous(),
                      act=L.ACT_GELU)
        if self.pre_ln:
            return self._fwd_post_preln(ws, B, taps, klen, mix_w)
        k.wavlm_add_rows(ws["x"], ws["pc"], ws["y"], B, Tn, Pp)
        k.layernorm_fwd(ws["y"], W["enc_ln_w"], W["enc_ln_b"], yT=ws["xb"], y32=ws["x"])
        if taps is not None:
            taps["enc_in"] = ws["x"].view(B, Tn, d).clone()
        nL = len(W["layers"])
        for i, Lw_ in enumerate(W["layers"]):
            if mix_w is not None:
                k.hidden_mix_axpy(ws["x"], mix_w, i, ws["feat"], overwrite=(i == 0))
            k.gemm_nt(ws["xb"], Lw_["wqkv"], ws["qkv"], bias=Lw_["bqkv"])
            gate = None
            if self.rel_pos:
                gate = k.wavlm_gate(ws["x"], Lw_["wg"], Lw_["bg"], Lw_["grep_a"], ws["gate"], B, Tn, self.H, self.dh)
                if taps is not None and i == 0:
                    taps["gate0"] = ws["gate"].clone()
            gate, rb = self._attn_bias(gate)
            k.xattn_fwd(ws["qkv"], ws["o"], ws["lse"], B, Tn, self.H, self.dh, gate=gate, rb=rb, klen=klen)
            k.gemm_nt(ws["o"], Lw_["wo"], ws["y"], bias=Lw_["bo"], res=ws["x"])
            k.layernorm_fwd(ws["y"], Lw_["ln1_w"], Lw_["ln1_b"], yT=ws["x1b"], y32=ws["x1"])
            k.gemm_nt(ws["x1b"], Lw_["w1"], ws["hm"], bias=Lw_["b1"], act=L.ACT_GELU)
            k.gemm_nt(ws["hm"], Lw_["w2"], ws["y"], bias=Lw_["b2"], res=ws["x1"])
            k.layernorm_fwd(ws["y"], Lw_["ln2_w"], Lw_["ln2_b"], yT=ws["xb"], y32=ws["x"])
            if taps is not None:
                taps[f"layer{i}"] = ws["x"].view(B, Tn, d).clone()
        if mix_w is not None:
            k.hidden_mix_axpy(ws["x"], mix_w, nL, ws["feat"], overwrite=(nL == 0))

    def _fwd_post_preln(self, ws, B, taps, klen, mix_w):
        """The inference encoder with layer_norm_first (WavLM.py:596-612,727-755 = wav2vec2.py:898-957,1037-1056): no LayerNorm
        behind the positional convolution; a layer is x += attn(LN1(x)); x += ffn(LN2(x)); the encoder's LayerNorm closes the
        stack.  Hidden states (s3prl hooks, wav2vec2_expert.py:58-66): the residual stream entering every layer + the final
        LayerNorm's output.  Called after the positional convolution has filled ws["pc"]."""
        W, k = self.W, ops
        Tn, Pp, d = ws["Tn"], ws["Pp"], self.d
        k.wavlm_add_rows(ws["x"], ws["pc"], ws["y"], B, Tn, Pp)
        x, y = ws["y"], ws["x"]                                        # the residual stream ping-pongs between the two buffers
        if taps is not None:
            taps["enc_in"] = x.view(B, Tn, d).clone()
        nL = len(W["layers"])
        for i, Lw_ in enumerate(W["layers"]):
            if mix_w is not None:
                k.hidden_mix_axpy(x, mix_w, i, ws["feat"], overwrite=(i == 0))
            k.layernorm_fwd(x, Lw_["ln1_w"], Lw_["ln1_b"], yT=ws["xb"], y32=ws["x1"] if self.rel_pos else None)
            k.gemm_nt(ws["xb"], Lw_["wqkv"], ws["qkv"], bias=Lw_["bqkv"])
            gate = None
            if self.rel_pos:                                           # the gate reads the attention's query = LN1(x)
                gate = k.wavlm_gate(ws["x1"], Lw_["wg"], Lw_["bg"], Lw_["grep_a"], ws["gate"], B, Tn, self.H, self.dh)
            gate, rb = self._attn_bias(gate)
            k.xattn_fwd(ws["qkv"], ws["o"], ws["lse"], B, Tn, self.H, self.dh, gate=gate, rb=rb, klen=klen)
            k.gemm_nt(ws["o"], Lw_["wo"], y, bias=Lw_["bo"], res=x)
            k.layernorm_fwd(y, Lw_["ln2_w"], Lw_["ln2_b"], yT=ws["x1b"])
            k.gemm_nt(ws["x1b"], Lw_["w1"], ws["hm"], bias=Lw_["b1"], act=L.ACT_GELU)
            k.gemm_nt(ws["hm"], Lw_["w2"], x, bias=Lw_["b2"], res=y)
            if taps is not None:
                taps[f"layer{i}"] = x.view(B, Tn, d).clone()
        k.layernorm_fwd(x, W["enc_ln_w"], W["enc_ln_b"], y32=ws["xf"])
        if mix_w is not None:
            k.hidden_mix_axpy(ws["xf"], mix_w, nL, ws["feat"], overwrite=(nL == 0))

    # ------------------------------------------------------------------ training: forward with saved activations, backward
    # dropout sites of one step: seed = base + 16 * (layer + 1) + site  (site 0 = encoder input / input features)
    _D1, _ATT, _D2, _D3 = 1, 2, 3, 4

    def _drop_base(self):
        return ((((self.seed + 7919 * self.rank_salt) * 1000003 + self.step) & 0x7FFFFFFF) << 12)

    def _site(self, sv, layer, site):
        return sv["drop"]["base"] + 16 * (layer + 1) + site

    def _fwd_post_train(self, ws, B, sv, klen=None, mix_w=None):
        """Training-mode forward of the encoder (the reference runs the backbone in train() mode in every regime, so the
        dropouts of lid/wavlm/WavLM.py:615 (encoder input), :745-771 (dropout1 / dropout2 / dropout3 of every layer) and the
        attention dropout inside multi_head_attention_forward apply even while the encoder's parameters are frozen).  Masks
        are functions of (seed, step, site, element index): the backward regenerates them, nothing is stored."""
        W, k, cfg = self.W, ops, self.cfg
        Tn, M, Pp, d = ws["Tn"], ws["M"], ws["Pp"], self.d
        p_drop, p_att, p_act = (float(cfg.get(n, 0.0)) for n in ("dropout", "attention_dropout", "activation_dropout"))
        sv["drop"] = dict(p=p_drop, att=p_att, act=p_act, base=self._drop_base())
        fk = self.forced_keep
        sv["klen"], sv["mix"] = klen, mix_w is not None
        if klen is not None:
            k.zero_padded_rows(ws["x"], klen, B, Tn)
        cg = d // self.gpos
        k.wavlm_posconv_prep(ws["x"], ws["xg"], B, Tn, self.gpos, Pp, self.kpos // 2)
        for g in range(self.gpos):
            A = ws["xg"][g].as_strided((B * Pp, self.kpos * cg), (cg, 1))
            k.gemm_nt(A, W["pos_w"][g], ws["pc"][:, g * cg:(g + 1) * cg], bias=W["pos_b"][g * cg:(g + 1) * cg].contiguous(),
                      act=L.ACT_GELU, out2=sv["pcp"][:, g * cg:(g + 1) * cg])
        k.wavlm_add_rows(ws["x"], ws["pc"], sv["y0"], B, Tn, Pp)
        # LayerDrop (WavLM.py:620-628: one np.random.random() per layer and step; a layer runs when the draw exceeds
        # encoder_layerdrop): the kept layers are chained through each other's saved-input buffers
        drop = float(self.cfg.get("encoder_layerdrop", 0.0))
        active = [i for i in range(self.n_layers) if np.random.random() > drop or drop <= 0.0]
        sv["active"] = active
        if mix_w is not None and len(active) != self.n_layers:
            raise LidkError("hidden_states weighted sum with LayerDrop: the number of hidden states changes from step to step "
                            "(the reference's Featurizer asserts on this too, interfaces.py:228): set encoder_layerdrop to 0")
        if self.pre_ln:
            return self._layers_train_preln(ws, B, sv, klen, mix_w, active, (p_drop, p_att, p_act))
        first = sv["layers"][active[0]] if active else None
        o32, ob = (first["xin"], first["xinb"]) if first is not None else (ws["x"], ws["xb"])
        k.layernorm_fwd(sv["y0"], W["enc_ln_w"], W["enc_ln_b"], yT=ob, y32=o32, mean=sv["mean0"], rstd=sv["rstd0"])
        if p_drop > 0:                                                    # F.dropout(x, p=self.dropout) in front of the layers
            k.dropout(o32, o32, p_drop, seed=self._site(sv, -1, 0), keep_in=fk.get("enc"))
            k.scale_cast(o32, ob, 1.0)
        for n, i in enumerate(active):
            Lw_, S = W["layers"][i], sv["layers"][i]
            nxt = sv["layers"][active[n + 1]] if n + 1 < len(active) else None
            xo32, xob = (nxt["xin"], nxt["xinb"]) if nxt is not None else (ws["x"], ws["xb"])
            k.gemm_nt(S["xinb"], Lw_["wqkv"], S["qkv"], bias=Lw_["bqkv"])
            gate = None
            if self.rel_pos and 0 in active:
                gate = k.wavlm_gate(S["xin"], Lw_["wg"], Lw_["bg"], Lw_["grep_a"], S["gate"], B, Tn, self.H, self.dh)
            gate, rb = self._attn_bias(gate, active)
            k.xattn_fwd(S["qkv"], S["o"], S["lse"], B, Tn, self.H, self.dh, gate=gate, rb=rb, klen=klen, drop_p=p_att,
                        seed=self._site(sv, i, self._ATT), keep=fk.get(("att", i)))
            if p_drop > 0:
                k.gemm_nt(S["o"], Lw_["wo"], ws["tmp"], bias=Lw_["bo"])
                k.dropout_add(ws["tmp"], S["xin"], S["y1"], p_drop, seed=self._site(sv, i, self._D1), keep_in=fk.get(("d1", i)))
            else:
                k.gemm_nt(S["o"], Lw_["wo"], S["y1"], bias=Lw_["bo"], res=S["xin"])
            k.layernorm_fwd(S["y1"], Lw_["ln1_w"], Lw_["ln1_b"], yT=S["x1b"], y32=S["x1"], mean=S["mean1"], rstd=S["rstd1"])
            k.gemm_nt(S["x1b"], Lw_["w1"], S["hm"], bias=Lw_["b1"], act=L.ACT_GELU, out2=S["pre"])
            if p_act > 0:
                k.dropout(S["hm"], S["hm"], p_act, seed=self._site(sv, i, self._D2), keep_in=fk.get(("d2", i)))
            if p_drop > 0:
                k.gemm_nt(S["hm"], Lw_["w2"], ws["tmp"], bias=Lw_["b2"])
                k.dropout_add(ws["tmp"], S["x1"], S["y2"], p_drop, seed=self._site(sv, i, self._D3), keep_in=fk.get(("d3", i)))
            else:
                k.gemm_nt(S["hm"], Lw_["w2"], S["y2"], bias=Lw_["b2"], res=S["x1"])
            k.layernorm_fwd(S["y2"], Lw_["ln2_w"], Lw_["ln2_b"], yT=xob, y32=xo32, mean=S["mean2"], rstd=S["rstd2"])
        if mix_w is not None:
            for n, i in enumerate(active):
                k.hidden_mix_axpy(sv["layers"][i]["xin"], mix_w, n, ws["feat"], overwrite=(n == 0))
            k.hidden_mix_axpy(ws["x"], mix_w, len(active), ws["feat"], overwrite=(len(active) == 0))

    def _layers_train_preln(self, ws, B, sv, klen, mix_w, active, drops):
        """Training-mode layers with layer_norm_first (WavLM.py:727-755): residual = x; x = residual + dropout1(attn(LN1(x)));
        x = x + dropout3(fc2(dropout2(gelu(fc1(LN2(x)))))), the encoder's LayerNorm behind the last layer (WavLM.py:596-600).  The
        residual stream of layer i lives in its ``xin`` buffer (f32) and is hidden state i of the s3prl mix; sv["y0"] (positional
        convolution + residual, no LayerNorm in front of the layers here) is copied / dropped out into the first one."""
        W, k = self.W, ops
        Tn, d = ws["Tn"], self.d
        p_drop, p_att, p_act = drops
        fk = self.forced_keep
        first = sv["layers"][active[0]]["xin"] if active else ws["x"]
        if p_drop > 0:                                                    # F.dropout(x, p=self.dropout) in front of the layers
            k.dropout(sv["y0"], first, p_drop, seed=self._site(sv, -1, 0), keep_in=fk.get("enc"))
        else:
            k.scale_cast(sv["y0"], first, 1.0)
        for n, i in enumerate(active):
            Lw_, S = W["layers"][i], sv["layers"][i]
            xo = sv["layers"][active[n + 1]]["xin"] if n + 1 < len(active) else ws["x"]
            k.layernorm_fwd(S["xin"], Lw_["ln1_w"], Lw_["ln1_b"], yT=S["xinb"], y32=S["x1"] if self.rel_pos else None,
                            mean=S["mean1"], rstd=S["rstd1"])
            k.gemm_nt(S["xinb"], Lw_["wqkv"], S["qkv"], bias=Lw_["bqkv"])
            gate = None
            if self.rel_pos and 0 in active:
                gate = k.wavlm_gate(S["x1"], Lw_["wg"], Lw_["bg"], Lw_["grep_a"], S["gate"], B, Tn, self.H, self.dh)
            gate, rb = self._attn_bias(gate, active)
            k.xattn_fwd(S["qkv"], S["o"], S["lse"], B, Tn, self.H, self.dh, gate=gate, rb=rb, klen=klen, drop_p=p_att,
                        seed=self._site(sv, i, self._ATT), keep=fk.get(("att", i)))
            if p_drop > 0:
                k.gemm_nt(S["o"], Lw_["wo"], ws["tmp"], bias=Lw_["bo"])
                k.dropout_add(ws["tmp"], S["xin"], S["y1"], p_drop, seed=self._site(sv, i, self._D1), keep_in=fk.get(("d1", i)))
            else:
                k.gemm_nt(S["o"], Lw_["wo"], S["y1"], bias=Lw_["bo"], res=S["xin"])
            k.layernorm_fwd(S["y1"], Lw_["ln2_w"], Lw_["ln2_b"], yT=S["x1b"], mean=S["mean2"], rstd=S["rstd2"])
            k.gemm_nt(S["x1b"], Lw_["w1"], S["hm"], bias=Lw_["b1"], act=L.ACT_GELU, out2=S["pre"])
            if p_act > 0:
                k.dropout(S["hm"], S["hm"], p_act, seed=self._site(sv, i, self._D2), keep_in=fk.get(("d2", i)))
            if p_drop > 0:
                k.gemm_nt(S["hm"], Lw_["w2"], ws["tmp"], bias=Lw_["b2"])
                k.dropout_add(ws["tmp"], S["y1"], xo, p_drop, seed=self._site(sv, i, self._D3), keep_in=fk.get(("d3", i)))
            else:
                k.gemm_nt(S["hm"], Lw_["w2"], xo, bias=Lw_["b2"], res=S["y1"])
        k.layernorm_fwd(ws["x"], W["enc_ln_w"], W["enc_ln_b"], y32=ws["xf"], mean=sv["meanF"], rstd=sv["rstdF"])
        if mix_w is not None:
            for n, i in enumerate(active):
                k.hidden_mix_axpy(sv["layers"][i]["xin"], mix_w, n, ws["feat"], overwrite=(n == 0))
            k.hidden_mix_axpy(ws["xf"], mix_w, len(active), ws["feat"], overwrite=(len(active) == 0))

    TRAINABLE_PREFIX = "encoder."
    INPUT_SIDE = ("layer_norm.weight", "layer_norm.bias", "mask_emb")      # never frozen by the reference's freeze_* helpers
    EXTRACTOR_PREFIXES = ("feature_extractor.", "post_extract_proj.")      # what (un)freeze_feature_extractor toggles

    def zero_grads(self):
        if getattr(self, "grads", None):
            self.grad_flat.zero_()

    def _alloc_grads(self):
        """f32 gradients of the trainable (encoder) parameters as views into ONE flat arena (``grad_flat``: a single buffer
        to zero and, under data parallelism, to all-reduce); q/k/v projections share a fused [3d, d] block whose row blocks
        are the three parameters' gradients."""
        if getattr(self, "grads", None):
            return
        dev, d = self.device, self.d
        plan, off = [], 0

        def take(name, shape):
            nonlocal off
            n = 1
            for v in shape:
                n *= v
            plan.append((name, off, tuple(shape)))
            off += -(-n // 64) * 64

        # Arena order = what trains in the reference's three regimes, so that under data parallelism only the live prefix is
        # exchanged: [hidden-state mixing logits of the s3prl Featurizer (owned by the model above, averaged with the rest) |
        # layer_norm, mask_emb: never frozen] [encoder.*: after freeze_tranformer_epoch] [extractor: after freeze_encoder_epoch]
        fused = (".q_proj.", ".k_proj.", ".v_proj.")
        shapes = self.param_shapes(self.cfg)
        self.grad_regions = {}
        take("featurizer.weights", (self.n_layers + 1,))
        for name in self.INPUT_SIDE:
            take(name, self.params[name].shape)
        self.grad_regions["input"] = off
        for name, t in self.params.items():
            if name.startswith(self.TRAINABLE_PREFIX) and not any(f in name for f in fused) and name in shapes:
                take(name, t.shape)
        for i in range(self.n_layers):
            take(f"qkv_w.{i}", (3 * d, d))
            take(f"qkv_b.{i}", (3 * d,))
        self.grad_regions["encoder"] = off
        for name, t in self.params.items():
            if name.startswith(self.EXTRACTOR_PREFIXES) and name in shapes:
                take(name, t.shape)
        self.grad_regions["extractor"] = off
        self.grad_flat = torch.zeros(off, device=dev)
        view = {name: self.grad_flat[o:o + math.prod(shape)].view(shape) for name, o, shape in plan}
        g = {n: v for n, v in view.items() if not n.startswith("qkv_") and n != "featurizer.weights"}
        self.mix_grad = view["featurizer.weights"]
        self._gqkv = []
        for i in range(self.n_layers):
            a = f"encoder.layers.{i}.self_attn."
            gw, gb = view[f"qkv_w.{i}"], view[f"qkv_b.{i}"]
            self._gqkv.append((gw, gb))
            for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                g[a + n + ".weight"], g[a + n + ".bias"] = gw[j * d:(j + 1) * d], gb[j * d:(j + 1) * d]
        self._gposw = torch.zeros(self.gpos, d // self.gpos, self.kpos * (d // self.gpos), device=dev)
        self._drb = torch.zeros(self.H, 2 * self.RB - 1, device=dev)
        self.grads = g

    @staticmethod
    def _splitk(n, kk):
        """Split of the M contraction for a [n, kk] weight gradient.  Measured at M = 9536 (tools/gemm_bench_wavlm.py): 4 splits
        beat 1 even for the 576-tile FFN gradients (115 vs 142 us) and more than 4 only adds float-atomic traffic."""
        tiles = -(-n // 64) * -(-kk // 64)
        return max(1, min(4, round(2048 / tiles)))

    def _wgrad(self, dy, x, gw, gb=None):
        ops.gemm_tn(dy, x, gw, colsum=gb, splitk=self._splitk(gw.shape[0], gw.shape[1]))

    def backward(self, dfeat: torch.Tensor, B: int, Lw: int, wgrads: bool = True, mix_w=None, mix_dw=None,
                 data_grads: bool = True, extractor: bool = False):
        """Backward of the last ``forward(..., train=True)`` of this (B, L) shape from dfeat = d(loss)/d(features) (B, T, d) f32.
        Accumulates into ``self.grads`` (reference names): always the parameters upstream of the transformer that the reference
        never freezes (``layer_norm.*`` in front of post_extract_proj and, under span masking, ``mask_emb``); with ``wgrads``
        (un-frozen encoder) also every ``encoder.*`` parameter.  With wgrads False the chain carries data gradients only.
        mix_w / mix_dw: the hidden-state mixing logits of the forward and where their gradient is accumulated; with
        data_grads False (nothing upstream of the features trains) only that gradient is produced."""
        ws = self._ws[(B, Lw)]
        sv, W, k = ws["save"], self.W, ops
        Tn, M, Pp, d, H = ws["Tn"], ws["M"], ws["Pp"], self.d, self.H
        self._alloc_grads()
        g, bf = self.grads, torch.bfloat16
        G = (lambda n: g[n]) if wgrads else (lambda n: None)
        dr, fk, klen, active = sv["drop"], self.forced_keep, sv["klen"], sv["active"]
        p_drop, p_att, p_act = dr["p"], dr["att"], dr["act"]
        if self.rel_pos:
            self._drb.zero_()
        dx = sv["dxa"]
        dfe = dfeat.reshape(M, d).contiguous()
        last = ws["xf"] if self.pre_ln else ws["x"]                          # the last hidden state = the encoder output
        if not data_grads:
            if sv["mix"] and mix_dw is not None:
                sv["dots"].zero_()
                for n, i in enumerate(active):
                    k.hidden_mix_dot(dfe, sv["layers"][i]["xin"], sv["dots"][n:n + 1])
                k.hidden_mix_dot(dfe, last, sv["dots"][len(active):len(active) + 1])
                k.hidden_mix_wgrad(mix_w, sv["dots"], mix_dw)
            return
        if sv["mix"]:
            if mix_w is None:
                raise LidkError("backward: the forward mixed hidden states; pass the same mix_w")
            sv["dots"].zero_()
            nL = len(active)
            k.hidden_mix_dot(dfe, last, sv["dots"][nL:nL + 1])
            k.hidden_mix_axpy(dfe, mix_w, nL, dx, overwrite=True)
        else:
            k.scale_cast(dfe, dx, 1.0)
        if self.pre_ln:                                                      # the encoder's closing LayerNorm (WavLM.py:598-599)
            k.layernorm_bwd(dx, ws["x"], sv["meanF"], sv["rstdF"], W["enc_ln_w"], sv["partial"], dx=sv["dxb"],
                            dgamma=G("encoder.layer_norm.weight"), dbeta=G("encoder.layer_norm.bias"), dtype=bf)
            dx = sv["dxb"]
        for n, i in reversed(list(enumerate(active))):
            if not self.pre_ln:
                break
            # x_out = y1 + dropout3(fc2(dropout2(gelu(fc1(LN2(y1)))))),  y1 = xin + dropout1(out_proj(attn(LN1(xin))))
            S, Lw_ = sv["layers"][i], W["layers"][i]
            q = f"encoder.layers.{i}."
            a = q + "self_attn."
            other = sv["dxb"] if dx is sv["dxa"] else sv["dxa"]
            if p_drop > 0:                                                 # dropout3 masks the fc2 branch only
                k.dropout(dx, sv["dm"], p_drop, seed=self._site(sv, i, self._D3), keep_in=fk.get(("d3", i)))
            else:
                k.scale_cast(dx, sv["dm"], 1.0)
            dy2 = sv["dm"]
            if wgrads:
                self._wgrad(dy2, S["hm"], g[q + "fc2.weight"], g[q + "fc2.bias"])
            k.gemm_nt(dy2, Lw_["w2T"], sv["dpre"], act=L.ACT_GELU_GRAD, aux=S["pre"])
            if p_act > 0:
                k.dropout(sv["dpre"], sv["dpre"], p_act, seed=self._site(sv, i, self._D2), keep_in=fk.get(("d2", i)))
            if wgrads:
                self._wgrad(sv["dpre"], S["x1b"], g[q + "fc1.weight"], g[q + "fc1.bias"])
            k.gemm_nt(sv["dpre"], Lw_["w1T"], sv["dx1"])                   # d LN2(y1)
            k.layernorm_bwd(sv["dx1"], S["y1"], S["mean2"], S["rstd2"], Lw_["ln2_w"], sv["partial"], dres=dx, dx=sv["da"],
                            dgamma=G(q + "final_layer_norm.weight"), dbeta=G(q + "final_layer_norm.bias"), dtype=bf)
            if p_drop > 0:                                                 # dropout1 masks the attention branch only
                k.dropout(sv["da"], sv["dab"], p_drop, seed=self._site(sv, i, self._D1), keep_in=fk.get(("d1", i)))
            else:
                k.scale_cast(sv["da"], sv["dab"], 1.0)
            dy1 = sv["dab"]
            if wgrads:
                self._wgrad(dy1, S["o"], g[a + "out_proj.weight"], g[a + "out_proj.bias"])
            k.gemm_nt(dy1, Lw_["woT"], sv["do"])
            gate, rb = self._attn_bias(S["gate"], active)
            k.xattn_bwd(S["qkv"], S["o"], sv["do"], S["lse"], sv["dqkv"], sv["delta"], B, Tn, H, self.dh, gate=gate, rb=rb,
                        klen=klen, drop_p=p_att, seed=self._site(sv, i, self._ATT), keep=fk.get(("att", i)),
                        dgate=sv["dgate"] if gate is not None else None, drb=self._drb if gate is not None else None)
            if wgrads:
                gw, gb = self._gqkv[i]
                self._wgrad(sv["dqkv"], S["xinb"], gw, gb)
            k.gemm_nt(sv["dqkv"], Lw_["wqkvT"], sv["db"])                  # d LN1(xin)
            if gate is not None:                                           # the gate read LN1's output too
                gp = g if wgrads else self._gate_scratch()
                k.wavlm_gate_bwd(S["x1"], Lw_["wg"], Lw_["bg"], Lw_["grep_a"], sv["dgate"], sv["db"], gp[a + "grep_linear.weight"],
                                 gp[a + "grep_linear.bias"], gp[a + "grep_a"].view(-1), B, Tn, H, self.dh)
            k.layernorm_bwd(sv["db"], S["xin"], S["mean1"], S["rstd1"], Lw_["ln1_w"], sv["partial"], dres=sv["da"], dx=other,
                            dgamma=G(q + "self_attn_layer_norm.weight"), dbeta=G(q + "self_attn_layer_norm.bias"), dtype=bf)
            if sv["mix"]:                                                  # this layer's residual input is hidden state n
                k.hidden_mix_dot(dfe, S["xin"], sv["dots"][n:n + 1])
                k.hidden_mix_axpy(dfe, mix_w, n, other)
            dx = other
        for n, i in reversed(list(enumerate(active))):
            if self.pre_ln:
                break
            S, Lw_ = sv["layers"][i], W["layers"][i]
            q = f"encoder.layers.{i}."
            a = q + "self_attn."
            other = sv["dxb"] if dx is sv["dxa"] else sv["dxa"]
            k.layernorm_bwd(dx, S["y2"], S["mean2"], S["rstd2"], Lw_["ln2_w"], sv["partial"], dx=sv["da"], dxT=sv["dab"],
                            dgamma=G(q + "final_layer_norm.weight"), dbeta=G(q + "final_layer_norm.bias"), dtype=bf)
            dy2 = sv["dab"]
            if p_drop > 0:                                                 # dropout3: only the fc2 branch is masked
                dy2 = k.dropout(sv["dab"], sv["dm"], p_drop, seed=self._site(sv, i, self._D3), keep_in=fk.get(("d3", i)))
            if wgrads:
                self._wgrad(dy2, S["hm"], g[q + "fc2.weight"], g[q + "fc2.bias"])
            k.gemm_nt(dy2, Lw_["w2T"], sv["dpre"], act=L.ACT_GELU_GRAD, aux=S["pre"])
            if p_act > 0:
                k.dropout(sv["dpre"], sv["dpre"], p_act, seed=self._site(sv, i, self._D2), keep_in=fk.get(("d2", i)))
            if wgrads:
                self._wgrad(sv["dpre"], S["x1b"], g[q + "fc1.weight"], g[q + "fc1.bias"])
            k.gemm_nt(sv["dpre"], Lw_["w1T"], sv["dx1"], res=sv["da"])
            k.layernorm_bwd(sv["dx1"], S["y1"], S["mean1"], S["rstd1"], Lw_["ln1_w"], sv["partial"], dx=sv["db"], dxT=sv["dbb"],
                            dgamma=G(q + "self_attn_layer_norm.weight"), dbeta=G(q + "self_attn_layer_norm.bias"), dtype=bf)
            dy1 = sv["dbb"]
            if p_drop > 0:                                                 # dropout1: only the attention branch is masked
                dy1 = k.dropout(sv["dbb"], sv["dm"], p_drop, seed=self._site(sv, i, self._D1), keep_in=fk.get(("d1", i)))
            if wgrads:
                self._wgrad(dy1, S["o"], g[a + "out_proj.weight"], g[a + "out_proj.bias"])
            k.gemm_nt(dy1, Lw_["woT"], sv["do"])
            # key-tiled backward: S tiles recomputed from Q, K and the saved log-sum-exp; dgate written, drb accumulated
            gate, rb = self._attn_bias(S["gate"], active)
            k.xattn_bwd(S["qkv"], S["o"], sv["do"], S["lse"], sv["dqkv"], sv["delta"], B, Tn, H, self.dh, gate=gate, rb=rb,
                        klen=klen, drop_p=p_att, seed=self._site(sv, i, self._ATT), keep=fk.get(("att", i)),
                        dgate=sv["dgate"] if gate is not None else None, drb=self._drb if gate is not None else None)
            if wgrads:
                gw, gb = self._gqkv[i]
                self._wgrad(sv["dqkv"], S["xinb"], gw, gb)
            k.gemm_nt(sv["dqkv"], Lw_["wqkvT"], other, res=sv["db"])
            if gate is not None:
                gp = g if wgrads else self._gate_scratch()
                k.wavlm_gate_bwd(S["xin"], Lw_["wg"], Lw_["bg"], Lw_["grep_a"], sv["dgate"], other, gp[a + "grep_linear.weight"],
                                 gp[a + "grep_linear.bias"], gp[a + "grep_a"].view(-1), B, Tn, H, self.dh)
            if sv["mix"]:                                                  # this layer's input is hidden state n
                k.hidden_mix_dot(dfe, S["xin"], sv["dots"][n:n + 1])
                k.hidden_mix_axpy(dfe, mix_w, n, other)
            dx = other
        if sv["mix"] and mix_dw is not None:
            k.hidden_mix_wgrad(mix_w, sv["dots"], mix_dw)
        if p_drop > 0:                                                     # the encoder-input dropout
            k.dropout(dx, dx, p_drop, seed=self._site(sv, -1, 0), keep_in=fk.get("enc"))
        # encoder.layer_norm (post-LN models: it sits in front of the layers), then y0 = x + gelu(pos_conv(x))
        if self.pre_ln:
            k.scale_cast(dx, sv["dy0"], 1.0)
        else:
            k.layernorm_bwd(dx, sv["y0"], sv["mean0"], sv["rstd0"], W["enc_ln_w"], sv["partial"], dx=sv["dy0"],
                            dgamma=G("encoder.layer_norm.weight"), dbeta=G("encoder.layer_norm.bias"), dtype=bf)
        k.wavlm_posconv_dprep(sv["dy0"], sv["pcp"], sv["dpc"], B, Tn, Pp, dpg=sv["dpg"], goff=self.kpos - 1 - self.kpos // 2)
        cg = d // self.gpos
        if wgrads:
            self._gposw.zero_()
            for gi in range(self.gpos):
                A = ws["xg"][gi].as_strided((B * Pp, self.kpos * cg), (cg, 1))
                X = sv["dpc"][:B * Pp, gi * cg:(gi + 1) * cg]
                ops.gemm_tn(X, A, self._gposw[gi], colsum=g["encoder.pos_conv.0.bias"][gi * cg:(gi + 1) * cg], splitk=16)
            self._posconv_weight_grads()
            if self.rel_pos:
                # the bias table's gradient back into the bucket embedding (owned by layer 0): emb[bucket(r)][h] += drb[h][r]
                g["encoder.layers.0.self_attn.relative_attention_bias.weight"].index_add_(0, W["rb_bucket"], self._drb.t().contiguous())
        # data gradient of the positional convolution (the same strided-view GEMM on the flipped kernel) + the residual path
        for gi in range(self.gpos):
            A = sv["dpg"][gi].as_strided((B * Pp, self.kpos * cg), (cg, 1))
            k.gemm_nt(A, W["pos_wd"][gi], sv["dxp"][:, gi * cg:(gi + 1) * cg])
        k.wavlm_add_rows(sv["dy0"], sv["dxp"], sv["da"], B, Tn, Pp)            # da = d(loss)/d(x after masking)
        dxm = sv["da"]
        if klen is not None:                                                   # x[padding] = 0 in front of the pos-conv
            k.zero_padded_rows(dxm, klen, B, Tn)
        if ws.get("masked"):                                                   # WavLM.apply_mask backward (WavLM.py:300-337)
            tm, cm = ws["masked"]
            v3 = dxm.view(B, Tn, d)
            if cm is not None:
                v3.mul_((cm == 0).to(v3.dtype)[:, None, :])
            if tm is not None:
                sel = tm.bool()
                g["mask_emb"] += (v3 * sel[:, :, None].to(v3.dtype)).sum((0, 1))
                v3.mul_((~sel)[:, :, None].to(v3.dtype))
        p_in = float(self.cfg.get("dropout_input", 0.0))
        if p_in > 0:
            k.dropout(dxm, dxm, p_in, seed=self._site(sv, -1, 1), keep_in=fk.get("in"))
        # post_extract_proj (frozen: data gradient only), then the LayerNorm on the conv features
        k.scale_cast(dxm, sv["dab"], 1.0)
        k.gemm_nt(sv["dab"], W["proj_wT"], sv["dxc"])
        dconv = sv["db"].view(-1)[:M * self.C].view(M, self.C)                  # gradient at the conv stack's output
        k.layernorm_bwd(sv["dxc"], ws["xc"], ws["mean_in"], ws["rstd_in"], W["ln0_w"], sv["partial"], dx=dconv,
                        dgamma=g["layer_norm.weight"], dbeta=g["layer_norm.bias"], dtype=bf)
        if extractor:
            self._wgrad(sv["dab"], ws["h0"], g["post_extract_proj.weight"], g["post_extract_proj.bias"])
            self._backward_extractor(ws, B, dconv)

    def _backward_extractor(self, ws, B, dconv):
        """Conv feature extractor backward (lid/wavlm/WavLM.py:409-531): layers 6..1 are the forward's strided-view GEMMs run
        backwards (weight gradient = TN GEMM on the same view, data gradient = NT GEMM to window space + col2im), layer 0 is
        recomputed from the waveform.  Needs a forward that kept the pre-activations (``train_extractor``)."""
        ex, W, k, g = ws.get("ex"), self.W, ops, self.grads
        if ex is None or not ws.get("ex_valid"):
            raise LidkError("extractor backward without a forward that kept the pre-activations (set train_extractor first)")
        T, P, bufs, C = ws["T"], ws["P"], ws["bufs"], self.C
        n = len(T)
        fe = "feature_extractor.conv_layers."
        fgm = float(self.cfg.get("feature_grad_mult", 1.0))       # WavLM.py:362-368: GradMultiply on the extractor's output
        if fgm <= 0:
            return                                                # the reference runs the extractor under no_grad then
        if fgm != 1.0:
            k.scale_cast(dconv, dconv, fgm)
        if self.ln_extractor:
            return self._backward_extractor_ln(ws, B, dconv)
        k.wavlm_conv_dlast(dconv, ex["pre"][n - 1][:B * P[n - 1]], ex["d"][n - 1][:B * P[n - 1]], B, T[n - 1], P[n - 1])
        for l in range(n - 1, 0, -1):
            _, kw, st = self.layers_spec[l]
            rows = B * P[l]
            dpre = ex["d"][l][:rows]
            A = bufs[l - 1].as_strided((rows, kw * C), (st * C, 1))
            gw = ex["gconv"][l]
            gw.zero_()
            k.gemm_tn(dpre, A, gw, splitk=16)
            g[f"{fe}{l}.0.weight"] += gw.view(C, kw, C).permute(0, 2, 1)                  # [Co][kW*Ci] -> [Co][Ci][kW]: layout glue
            dcol = ex["dcol"].view(-1)[:rows * kw * C].view(rows, kw * C)
            k.gemm_nt(dpre, W["convT"][l - 1], dcol)
            k.wavlm_conv_col2im(dcol, ex["pre"][l - 1][:B * P[l - 1]] if l > 1 else None, ex["d"][l - 1][:B * P[l - 1]], B, P[l],
                                T[l], T[l - 1], kw, C)
        off = k.wavlm_conv0_stats_offset(B, T[0], C)
        ex["gconv0"].zero_()
        k.wavlm_conv0_bwd(ws["wav"], W["conv0_w"], W["gn_w"], W["gn_b"], ws["c0ws"][off:off + B * C * 2], ex["d"][0][:B * P[0]],
                          ex["sums"], ex["gconv0"], g[fe + "0.2.weight"], g[fe + "0.2.bias"], T[0], P[0])
        g[fe + "0.0.weight"] += ex["gconv0"].view(C, 1, -1)

    def _backward_extractor_ln(self, ws, B, dconv):
        """The layer_norm extractor's backward (WavLM.py:438-450 per layer: conv(+bias) -> LayerNorm(C) -> GELU): the gradient at a
        layer's output goes through lidk_ln_gelu_bwd (in place: -> gradient at the conv output, + the LayerNorm's parameter
        gradients), the convolution's weight / bias gradient is the TN GEMM on the forward's strided view (bias = column sums),
        its data gradient the NT GEMM to window space + col2im; layer 0 is recomputed from the waveform."""
        ex, W, k, g = ws["ex"], self.W, ops, self.grads
        T, P, bufs, C = ws["T"], ws["P"], ws["bufs"], self.C
        n = len(T)
        fe = "feature_extractor.conv_layers."
        k.wavlm_conv_dlast(dconv, None, ex["d"][n - 1][:B * P[n - 1]], B, T[n - 1], P[n - 1])
        for l in range(n - 1, 0, -1):
            _, kw, st = self.layers_spec[l]
            rows = B * P[l]
            d = ex["d"][l][:rows]
            k.ln_gelu_bwd(d, ex["pre"][l][:rows], W["xln_w"][l], W["xln_b"][l], d, g[f"{fe}{l}.2.1.weight"], g[f"{fe}{l}.2.1.bias"],
                          B, P[l], T[l])
            A = bufs[l - 1].as_strided((rows, kw * C), (st * C, 1))
            gw = ex["gconv"][l]
            gw.zero_()
            k.gemm_tn(d, A, gw, colsum=g[f"{fe}{l}.0.bias"] if self.conv_bias else None, splitk=16)
            g[f"{fe}{l}.0.weight"] += gw.view(C, kw, C).permute(0, 2, 1)
            dcol = ex["dcol"].view(-1)[:rows * kw * C].view(rows, kw * C)
            k.gemm_nt(d, W["convT"][l - 1], dcol)
            k.wavlm_conv_col2im(dcol, None, ex["d"][l - 1][:B * P[l - 1]], B, P[l], T[l], T[l - 1], kw, C)
        ex["gconv0"].zero_()
        k.conv0_ln_bwd(ws["wav"], W["conv0_w"], W["conv_b"][0], W["xln_w"][0], W["xln_b"][0], ex["d"][0][:B * P[0]], ex["gconv0"],
                       g[fe + "0.0.bias"] if self.conv_bias else None, g[fe + "0.2.1.weight"], g[fe + "0.2.1.bias"], T[0], P[0])
        g[fe + "0.0.weight"] += ex["gconv0"].view(C, 1, -1)

    def _gate_scratch(self):
        """Throw-away targets for the gate's parameter gradients while the encoder is frozen (data gradients only)."""
        if getattr(self, "_gscratch", None) is None:
            dev = self.device
            t = dict(w=torch.zeros(8, self.dh, device=dev), b=torch.zeros(8, device=dev), a=torch.zeros(self.H, device=dev))

            class _D(dict):
                def __missing__(self_, key):
                    return t["w"] if key.endswith("grep_linear.weight") else t["b"] if key.endswith("grep_linear.bias") else t["a"]
            self._gscratch = _D()
        return self._gscratch

    def _posconv_weight_grads(self):
        """Gradient of the effective weight w = g * v / ||v|| (weight_norm, dim = 2) back to weight_g / weight_v: a few
        element-wise passes over 4.7 M values (torch ops on the device: parameter bookkeeping, not the hot path)."""
        cg = self.d // self.gpos
        dw = self._gposw.view(self.gpos, cg, self.kpos, cg).permute(0, 1, 3, 2).reshape(self.d, cg, self.kpos)   # [o][c][j]
        v, gg = self.params["encoder.pos_conv.0.weight_v"].to(self.device), self.params["encoder.pos_conv.0.weight_g"].to(self.device)
        norm = v.pow(2).sum((0, 1), keepdim=True).sqrt()
        dot = (dw * v).sum((0, 1), keepdim=True)
        self.grads["encoder.pos_conv.0.weight_g"] += dot / norm
        self.grads["encoder.pos_conv.0.weight_v"] += gg / norm * (dw - v * dot / norm.pow(2))

    def refresh(self, changed=None):
        """Re-derive the kernels' operands after the optimizer changed parameters (in place).  ``changed``: names of the backbone
        parameters that take gradients; when they are only the input LayerNorm and ``mask_emb`` (the frozen regime: everything else
        is unchanged) just those three operands are refreshed - a full re-derivation re-casts / transposes / concatenates ~95 M
        parameters and costs ~290 device copies, ~3 ms of GPU time and as much host time per step for nothing."""
        W = getattr(self, "W", None)
        if changed is not None and self._prepared and W is not None and set(changed) <= set(self.INPUT_SIDE):
            for key, name in (("ln0_w", "layer_norm.weight"), ("ln0_b", "layer_norm.bias"), ("mask_emb", "mask_emb")):
                if name in changed:
                    W[key].copy_(self.params[name], non_blocking=True)
            return
        self._prepared = False

    # ------------------------------------------------------------------ forward
    def forward(self, wav: torch.Tensor, taps: Dict[str, torch.Tensor] = None, mask: bool = False,
                n_samples: List[int] = None, train: bool = False, mix_w: torch.Tensor = None) -> torch.Tensor:
        """wav (B, L) f32 on the GPU -> (B, T, d) f32.  ``taps`` (tests): receives copies of the stage outputs.
        mask=True (training, WavLM.apply_mask): spans of the projected features are replaced by ``mask_emb`` / zeroed channels
        with cfg mask_prob / mask_channel_prob.  n_samples (true lengths of a zero-padded batch): in WavLM it only shapes the
        padding mask those spans avoid - like the reference, the encoder itself never sees a padding mask; with
        cfg ``encoder_padding_mask`` (wav2vec2) padded frames are zeroed in front of the positional convolution and masked out
        as attention keys.  mix_w (L + 1,) f32 on the GPU: return the softmax(mix_w)-weighted sum of the hidden states (the input
        of every layer and the encoder output: s3prl's ``hidden_states`` feature) instead of the last one."""
        if not wav.is_cuda or wav.dtype != torch.float32:
            raise LidkError("WavLMBackbone.forward needs a float32 GPU tensor (B, L)")
        if not self._prepared:
            if getattr(self, "W", None) is not None and self._inplace_ok:
                self._refresh_inplace()
            else:
                self._prepare()
        wav = wav.contiguous()
        B, Lw = wav.shape
        ws = self._workspace(B, Lw)
        Tn, d = ws["Tn"], self.d
        if self.normalize:                                    # task.normalize: F.layer_norm(wav, wav.shape) per utterance
            ragged = n_samples is not None and any(int(n) < Lw for n in n_samples)
            ns = self._upload_mask(ws, "nsamp", np.asarray([min(int(n), Lw) for n in n_samples], dtype=np.int32)) if ragged else None
            ops.wav_layernorm(wav, out=ws["wav"], n_samples=ns)
        else:
            ws["wav"].copy_(wav)                              # static input buffer: captured launches see one address
        masking = mask and (self.cfg.get("mask_prob", 0.0) > 0 or self.cfg.get("mask_channel_prob", 0.0) > 0)
        ws["masked"] = None
        klen = self._key_lengths(ws, B, Tn, Lw, n_samples)
        if train:                                             # keep what backward needs (eager launches, dropouts on)
            self.step += 1
            keep_pre = bool(self.train_extractor)
            self.graphs.run(("pre", B, Lw, keep_pre), lambda: self._fwd_pre(ws, B, None, keep_pre))
            ws["ex_valid"] = keep_pre
            sv = self._train_buffers(ws, B)
            p_in = float(self.cfg.get("dropout_input", 0.0))
            if p_in > 0:
                sv["drop"] = dict(base=self._drop_base())
                ops.dropout(ws["x"], ws["x"], p_in, seed=self._site(sv, -1, 1), keep_in=self.forced_keep.get("in"))
            if masking:
                self._apply_mask(ws, B, Tn, Lw, n_samples)
            self._fwd_post_train(ws, B, sv, klen, mix_w)
        elif taps is not None:                                # tests: eager, with copies of the stage outputs
            self._fwd_pre(ws, B, taps)
            if masking:
                self._apply_mask(ws, B, Tn, Lw, n_samples)
            self._fwd_post(ws, B, taps, klen=klen, mix_w=mix_w)
        else:
            self.graphs.run(("pre", B, Lw, False), lambda: self._fwd_pre(ws, B, None))
            ws["ex_valid"] = False
            if masking:
                self._apply_mask(ws, B, Tn, Lw, n_samples)     # host-drawn spans -> two small H2D copies + one launch
            self.graphs.run(("post", B, Lw, klen is not None, None if mix_w is None else mix_w.data_ptr()),
                            lambda: self._fwd_post(ws, B, None, klen=klen, mix_w=mix_w))
        return (ws["feat"] if mix_w is not None else ws["xf"] if self.pre_ln else ws["x"]).view(B, Tn, d)
