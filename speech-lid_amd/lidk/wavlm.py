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
    added inside the MFMA attention kernel (lidk_wavlm_attn_fwd).
"""
import math
from typing import Dict, List

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
            sz = all_sz - int(padding_mask[i].long().sum().item())
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


class WavLMBackbone:
    RB = 1024                       # the bias table covers offsets |j - i| < RB

    def __init__(self, cfg: Dict):
        self.cfg = dict(cfg)
        self.layers_spec = eval(cfg.get("conv_feature_layers", "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2"))
        self.C = self.layers_spec[0][0]
        if any(c != self.C for c, _, _ in self.layers_spec) or self.layers_spec[0][1:] != (10, 5):
            raise NotImplementedError("lidk WavLM: conv feature extractor of the (C,10,5) + (C,k,s)* form with one width")
        if cfg.get("extractor_mode", "default") != "default" or cfg.get("conv_bias", False) or cfg.get("layer_norm_first", False):
            raise NotImplementedError("lidk WavLM: extractor_mode=default, conv_bias=False, layer_norm_first=False (WavLM-Base/Base+)")
        if not (cfg.get("relative_position_embedding", False) and cfg.get("gru_rel_pos", False)):
            raise NotImplementedError("lidk WavLM: relative_position_embedding + gru_rel_pos (every released WavLM checkpoint)")
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
        """Names and shapes of lid/wavlm/WavLM.py's ``state_dict()`` for a config (conv_bias=False, extractor_mode=default,
        relative position embedding owned by layer 0)."""
        spec = eval(cfg.get("conv_feature_layers", "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2"))
        C, d = spec[0][0], cfg.get("encoder_embed_dim", 768)
        ffn, H = cfg.get("encoder_ffn_embed_dim", 3072), cfg.get("encoder_attention_heads", 12)
        kpos, gpos = cfg.get("conv_pos", 128), cfg.get("conv_pos_groups", 16)
        s = {"mask_emb": (d,), "feature_extractor.conv_layers.0.0.weight": (C, 1, spec[0][1]),
             "feature_extractor.conv_layers.0.2.weight": (C,), "feature_extractor.conv_layers.0.2.bias": (C,)}
        for i in range(1, len(spec)):
            s[f"feature_extractor.conv_layers.{i}.0.weight"] = (C, C, spec[i][1])
        s.update({"post_extract_proj.weight": (d, C), "post_extract_proj.bias": (d,), "encoder.pos_conv.0.bias": (d,),
                  "encoder.pos_conv.0.weight_g": (1, 1, kpos), "encoder.pos_conv.0.weight_v": (d, d // gpos, kpos)})
        for i in range(cfg.get("encoder_layers", 12)):
            p = f"encoder.layers.{i}."
            s[p + "self_attn.grep_a"] = (1, H, 1, 1)
            if i == 0:
                s[p + "self_attn.relative_attention_bias.weight"] = (cfg.get("num_buckets", 320), H)
            for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
                s[p + f"self_attn.{n}.weight"], s[p + f"self_attn.{n}.bias"] = (d, d), (d,)
            s[p + "self_attn.grep_linear.weight"], s[p + "self_attn.grep_linear.bias"] = (8, d // H), (8,)
            s[p + "self_attn_layer_norm.weight"], s[p + "self_attn_layer_norm.bias"] = (d,), (d,)
            s[p + "fc1.weight"], s[p + "fc1.bias"], s[p + "fc2.weight"], s[p + "fc2.bias"] = (ffn, d), (ffn,), (d, ffn), (d,)
            s[p + "final_layer_norm.weight"], s[p + "final_layer_norm.bias"] = (d,), (d,)
        s.update({"encoder.layer_norm.weight": (d,), "encoder.layer_norm.bias": (d,), "layer_norm.weight": (C,),
                  "layer_norm.bias": (C,)})
        return s

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        self.params = {k: v.detach().clone().float() for k, v in sd.items()}
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
        W = {"conv0_w": g(fe + "0.0.weight").reshape(self.C, 10), "gn_w": g(fe + "0.2.weight"), "gn_b": g(fe + "0.2.bias"), "conv": []}
        for i in range(1, len(self.layers_spec)):
            w = g(f"{fe}{i}.0.weight")                                        # [Co][Ci][kW] -> [Co][kW*Ci], k-major
            W["conv"].append(w.permute(0, 2, 1).reshape(w.shape[0], -1).to(bf).contiguous())
        W["ln0_w"], W["ln0_b"] = g("layer_norm.weight"), g("layer_norm.bias")
        W["proj_w"], W["proj_b"] = g("post_extract_proj.weight").to(bf), g("post_extract_proj.bias")
        wv, wg = g("encoder.pos_conv.0.weight_v"), g("encoder.pos_conv.0.weight_g")
        w = torch._weight_norm(wv, wg, 2)                                       # weight_norm(dim=2): per kernel position
        cg = self.d // self.gpos
        W["pos_w"] = [w[i * cg:(i + 1) * cg].permute(0, 2, 1).reshape(cg, -1).to(bf).contiguous() for i in range(self.gpos)]
        W["pos_b"] = g("encoder.pos_conv.0.bias")
        W["enc_ln_w"], W["enc_ln_b"] = g("encoder.layer_norm.weight"), g("encoder.layer_norm.bias")
        emb = g("encoder.layers.0.self_attn.relative_attention_bias.weight")      # [buckets][H], shared by every layer
        r = torch.arange(-(self.RB - 1), self.RB, device=dev)
        bucket = relative_buckets(r, self.cfg.get("num_buckets", 320), self.cfg.get("max_distance", 800))
        W["rb"] = emb[bucket].t().contiguous()                                   # [H][2*RB-1], entry r + RB - 1
        W["layers"] = []
        for i in range(self.n_layers):
            q = f"encoder.layers.{i}."
            a = q + "self_attn."
            W["layers"].append(dict(
                wqkv=torch.cat([g(a + "q_proj.weight"), g(a + "k_proj.weight"), g(a + "v_proj.weight")]).to(bf).contiguous(),
                bqkv=torch.cat([g(a + "q_proj.bias"), g(a + "k_proj.bias"), g(a + "v_proj.bias")]).contiguous(),
                wo=g(a + "out_proj.weight").to(bf), bo=g(a + "out_proj.bias"),
                wg=g(a + "grep_linear.weight"), bg=g(a + "grep_linear.bias"), grep_a=g(a + "grep_a").reshape(-1).contiguous(),
                ln1_w=g(q + "self_attn_layer_norm.weight"), ln1_b=g(q + "self_attn_layer_norm.bias"),
                w1=g(q + "fc1.weight").to(bf), b1=g(q + "fc1.bias"), w2=g(q + "fc2.weight").to(bf), b2=g(q + "fc2.bias"),
                ln2_w=g(q + "final_layer_norm.weight"), ln2_b=g(q + "final_layer_norm.bias")))
        self.W = W
        self._prepared = True

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
        if T[-1] > ops.wavlm_attn_max_frames(self.dh):
            raise LidkError(f"WavLM: {T[-1]} frames ({Lw / 16000:.1f} s) exceed the attention kernel's limit of "
                            f"{ops.wavlm_attn_max_frames(self.dh)} frames (dh = {self.dh})")
        n = len(T)
        # row pitches with P_l = stride_{l+1} * P_{l+1}, large enough for every layer's valid rows
        mult = [1] * n
        for l in range(n - 2, -1, -1):
            mult[l] = mult[l + 1] * self.layers_spec[l + 1][2]
        P_last = max(-(-T[l] // mult[l]) for l in range(n))
        P = [P_last * m for m in mult]
        dev, bf = self.device, torch.bfloat16
        bufs = [torch.zeros(B * P[l] + 8, self.C, device=dev, dtype=(bf if l < n - 1 else torch.float32)) for l in range(n)]
        Tn, M, d = T[-1], B * T[-1], self.d
        Pp = Tn + self.kpos
        ws = dict(T=T, P=P, bufs=bufs, M=M, Tn=Tn, Pp=Pp, wav=torch.empty(B, Lw, device=dev),
                  c0ws=torch.empty(ops.wavlm_conv0_workspace(B, T[0], self.C), device=dev),
                  tmask=torch.zeros(B, Tn, device=dev, dtype=torch.uint8), cmask=torch.zeros(B, d, device=dev, dtype=torch.uint8),
                  xc=torch.empty(M, self.C, device=dev), h0=torch.empty(M, self.C, device=dev, dtype=bf),
                  x=torch.empty(M, d, device=dev), xb=torch.empty(M, d, device=dev, dtype=bf),
                  x1=torch.empty(M, d, device=dev), x1b=torch.empty(M, d, device=dev, dtype=bf), y=torch.empty(M, d, device=dev),
                  xg=torch.empty(self.gpos, B * Pp + self.kpos, d // self.gpos, device=dev, dtype=bf),
                  pc=torch.empty(B * Pp, d, device=dev), qkv=torch.empty(M, 3 * d, device=dev, dtype=bf),
                  o=torch.empty(M, d, device=dev, dtype=bf), hm=torch.empty(M, self.ffn, device=dev, dtype=bf),
                  gate=torch.empty(B, self.H, Tn, device=dev))
        if len(self._ws) > 8:
            self._ws.clear()
            self.graphs.clear()
        self._ws[key] = ws
        return ws

    def _apply_mask(self, ws, B, Tn, Lw, n_samples):
        cfg, dev = self.cfg, self.device
        pad = None
        if n_samples is not None and min(n_samples) < Lw:          # WavLM.forward_padding_mask: a frame is padding if all of
            per = Lw // Tn                                          # its Lw // T samples are (the remainder samples are dropped)
            pm = torch.ones(B, Lw, dtype=torch.bool)
            for i, n in enumerate(n_samples):
                pm[i, :n] = False
            pad = pm[:, :per * Tn].view(B, Tn, per).all(-1)
        tmask = cmask = None
        if cfg.get("mask_prob", 0.0) > 0:
            m = span_mask((B, Tn), pad, cfg["mask_prob"], cfg.get("mask_length", 10), min_masks=2)
            tmask = ws["tmask"]
            tmask.copy_(torch.from_numpy(m.view("uint8")), non_blocking=True)
        if cfg.get("mask_channel_prob", 0.0) > 0:
            m = span_mask((B, self.d), None, cfg["mask_channel_prob"], cfg.get("mask_channel_length", 10))
            cmask = ws["cmask"]
            cmask.copy_(torch.from_numpy(m.view("uint8")), non_blocking=True)
        if "mask_emb" not in self.W:
            self.W["mask_emb"] = self.params["mask_emb"].to(dev).contiguous()
        ops.wavlm_apply_mask(ws["x"], tmask, cmask, self.W["mask_emb"], B, Tn)


    def _fwd_pre(self, ws, B, taps):
        """Feature extractor, LayerNorm, post_extract_proj -> ws["x"] (B*T, d) f32."""
        W, k, wav = self.W, ops, ws["wav"]
        T, P, bufs, C, d = ws["T"], ws["P"], ws["bufs"], self.C, self.d
        k.wavlm_conv0(wav, W["conv0_w"], W["gn_w"], W["gn_b"], bufs[0], T[0], P[0], workspace=ws["c0ws"])
        for l in range(1, len(T)):
            _, kw, st = self.layers_spec[l]
            A = bufs[l - 1].as_strided((B * P[l], kw * C), (st * C, 1))          # strided view: the convolution is this GEMM
            k.gemm_nt(A, W["conv"][l - 1], bufs[l][:B * P[l]], act=L.ACT_GELU)
        Tn, M, Pp = ws["Tn"], ws["M"], ws["Pp"]
        last = bufs[-1][:B * P[-1]]
        k.scale_cast_2d(last.view(B, P[-1] * C), ws["xc"].view(B, Tn * C), B, Tn * C)       # drop the pitch padding rows
        if taps is not None:
            taps["conv"] = ws["xc"].view(B, Tn, C).clone()
        k.layernorm_fwd(ws["xc"], W["ln0_w"], W["ln0_b"], yT=ws["h0"])
        k.gemm_nt(ws["h0"], W["proj_w"], ws["x"], bias=W["proj_b"])
        if taps is not None:
            taps["proj"] = ws["x"].view(B, Tn, d).clone()

    def _fwd_post(self, ws, B, taps):
        """Positional convolution + LayerNorm, transformer layers -> ws["x"]."""
        W, k = self.W, ops
        Tn, M, Pp, d = ws["Tn"], ws["M"], ws["Pp"], self.d
        # positional convolution + residual + LayerNorm
        cg = d // self.gpos
        k.wavlm_posconv_prep(ws["x"], ws["xg"], B, Tn, self.gpos, Pp, self.kpos // 2)
        for g in range(self.gpos):
            A = ws["xg"][g].as_strided((B * Pp, self.kpos * cg), (cg, 1))
            k.gemm_nt(A, W["pos_w"][g], ws["pc"][:, g * cg:(g + 1) * cg], bias=W["pos_b"][g * cg:(g + 1) * cg].contiguous(),
                      act=L.ACT_GELU)
        k.wavlm_add_rows(ws["x"], ws["pc"], ws["y"], B, Tn, Pp)
        k.layernorm_fwd(ws["y"], W["enc_ln_w"], W["enc_ln_b"], yT=ws["xb"], y32=ws["x"])
        if taps is not None:
            taps["enc_in"] = ws["x"].view(B, Tn, d).clone()
        for i, Lw_ in enumerate(W["layers"]):
            k.gemm_nt(ws["xb"], Lw_["wqkv"], ws["qkv"], bias=Lw_["bqkv"])
            k.wavlm_gate(ws["x"], Lw_["wg"], Lw_["bg"], Lw_["grep_a"], ws["gate"], B, Tn, self.H, self.dh)
            if taps is not None and i == 0:
                taps["gate0"] = ws["gate"].clone()
            k.wavlm_attn_fwd(ws["qkv"], ws["gate"], W["rb"], ws["o"], B, Tn, self.H, self.dh)
            k.gemm_nt(ws["o"], Lw_["wo"], ws["y"], bias=Lw_["bo"], res=ws["x"])
            k.layernorm_fwd(ws["y"], Lw_["ln1_w"], Lw_["ln1_b"], yT=ws["x1b"], y32=ws["x1"])
            k.gemm_nt(ws["x1b"], Lw_["w1"], ws["hm"], bias=Lw_["b1"], act=L.ACT_GELU)
            k.gemm_nt(ws["hm"], Lw_["w2"], ws["y"], bias=Lw_["b2"], res=ws["x1"])
            k.layernorm_fwd(ws["y"], Lw_["ln2_w"], Lw_["ln2_b"], yT=ws["xb"], y32=ws["x"])
            if taps is not None:
                taps[f"layer{i}"] = ws["x"].view(B, Tn, d).clone()

    # ------------------------------------------------------------------ forward
    def forward(self, wav: torch.Tensor, taps: Dict[str, torch.Tensor] = None, mask: bool = False,
                n_samples: List[int] = None) -> torch.Tensor:
        """wav (B, L) f32 on the GPU -> (B, T, d) f32.  ``taps`` (tests): receives copies of the stage outputs.
        mask=True (training, WavLM.apply_mask): spans of the projected features are replaced by ``mask_emb`` / zeroed channels
        with cfg mask_prob / mask_channel_prob; n_samples (true lengths of a zero-padded batch) only shapes the padding mask
        those spans avoid - like the reference, the encoder itself never sees a padding mask."""
        if not wav.is_cuda or wav.dtype != torch.float32:
            raise LidkError("WavLMBackbone.forward needs a float32 GPU tensor (B, L)")
        if not self._prepared:
            self._prepare()
        wav = wav.contiguous()
        B, Lw = wav.shape
        ws = self._workspace(B, Lw)
        Tn, d = ws["Tn"], self.d
        ws["wav"].copy_(wav)                                  # static input buffer: captured launches see one address
        masking = mask and (self.cfg.get("mask_prob", 0.0) > 0 or self.cfg.get("mask_channel_prob", 0.0) > 0)
        if taps is not None:                                  # tests: eager, with copies of the stage outputs
            self._fwd_pre(ws, B, taps)
            if masking:
                self._apply_mask(ws, B, Tn, Lw, n_samples)
            self._fwd_post(ws, B, taps)
        else:
            self.graphs.run(("pre", B, Lw), lambda: self._fwd_pre(ws, B, None))
            if masking:
                self._apply_mask(ws, B, Tn, Lw, n_samples)     # host-drawn spans -> two small H2D copies + one launch
            self.graphs.run(("post", B, Lw), lambda: self._fwd_post(ws, B, None))
        return ws["x"].view(B, Tn, d)
