"""WavLMMutiLangModel - the reference's WavLM-backbone LID model surface (lid/WavLMMutiLangModel.py:19-284) on the lidk kernels.

  DataProcessor (16 kHz pass-through) -> WavLMMutiModel: WavLM backbone (lid/wavlm/WavLM.py) -> per-language ConformerLinear
  heads (one ConformerBlock dim = linear_dim, heads num_head x dim_head, + Linear to V+1) -> LangDiscriminator.

Kept: constructor keywords, ``forward(x: list of 1-D waveforms, sample_rate, lang) -> ({lang: logits (B, T, V+1)},
(lid_asr, lid_linear))``, attribute paths (``model.model.featurizer``, ``model.model.last_projects[lang]``, ``loss_fns``,
``wer_fn``, ``lang_discriminator``), ``freeze_* / unfreeze_*`` and the ``state_dict`` names: ``model.featurizer.model.<WavLM
key>``, ``model.last_projects.<lang>.block.*`` / ``.linear.*``, ``lang_discriminator.linear.{0,2}.*`` (the reference's
torchaudio ``data_processor.resampler*.kernel`` buffers are ignored on load and not written).

What runs: the backbone on hand-written HIP (lidk/wavlm.py) without autograd, the heads forward + backward on the lidk Engine
(``front="features"``).  While the backbone is frozen (``freeze_encoder_epoch`` / ``freeze_tranformer_epoch``,
lid/LidModule_ASR.py:243-258) and in inference the backbone forward is two captured graphs.  After
``unfreeze_tranformer_encoder()`` the transformer encoder (pos-conv, encoder LayerNorm, every layer, the relative-position
bucket embedding) trains too: the forward keeps each layer's activations, the Engine hands back d(loss)/d(features) and
``WavLMBackbone.backward`` produces the encoder gradients, published as the ``.grad`` of the reference-named Parameters.
``unfreeze_feature_extractor()`` (the reference reaches it after ``freeze_encoder_epoch`` = 100, lid/LidModule_ASR.py:26) adds
the convolutional feature extractor and post_extract_proj: the conv stack's backward mirrors its strided-view GEMMs.
"""
import logging
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from lid.ConformerLangModel import CtcLossFn, _child, _EngineBoundModel, _EngineFn, char_error_rate  # noqa: F401
from lidk.engine import Engine
from lidk.layout import ConformerCfg
from lidk.wavlm import WavLMBackbone
from lidk._lib import LidkError


class WavLMMutiLangModel(_EngineBoundModel):
    BB_PREFIX = "model.featurizer.model."         # where the backbone's parameters sit in the reference state_dict
    MIX_NAME = None                               # wav2vec2: the s3prl Featurizer's layer-mixing logits
    APPLY_NORMALIZE = False                       # whether the checkpoint cfg's ``normalize`` is applied to the waveforms

    def __init__(self, pt_path: str = None, feature_selection: str = "hidden_states", dropout: float = 0.0, linear_dim: int = 768,
                 mask: bool = True, num_layers: int = 1, lang2vocab: Dict = None, lang2index: Dict = None, hidden_dim: int = 128,
                 conformer_linear: bool = False, double_swish: bool = False, use_pre_train: bool = True,
                 mask_channel_prob: float = 0, mask_prob: float = 0.0, conformer_pure: bool = False, use_mask: bool = False,
                 dim_head: int = 32, num_head: int = 8, compute_dtype=torch.bfloat16, wavlm_cfg: Optional[Dict] = None,
                 train_input_norm: bool = True, _weights: Optional[Dict] = None, **_ignored):
        super().__init__()
        if not conformer_linear:
            raise NotImplementedError("LSTM heads (conformer_linear=False) are outside the lidk path (SURVEY 2 #3)")
        if num_layers != 1 or double_swish or use_mask or conformer_pure:
            raise NotImplementedError("lidk WavLM LID model: one ConformerBlock per head, Swish, use_mask=False, conformer_pure=False")
        if pt_path is not None:
            ckpt = torch.load(pt_path, map_location="cpu", weights_only=False)       # {"cfg": dict, "model": state_dict}
            cfg, weights = dict(ckpt["cfg"]), (ckpt["model"] if use_pre_train else None)
        elif wavlm_cfg is not None:                                                   # tests / synthetic runs: no checkpoint file
            cfg, weights = dict(wavlm_cfg), _weights
        else:
            raise ValueError("WavLMMutiLangModel needs pt_path (a WavLM checkpoint with 'cfg' and 'model') or wavlm_cfg")
        # WavLMModel.forward (lid/WavLMMutiLangModel.py:160-182) hands the raw padded batch to WavLM.extract_features: the
        # checkpoint's ``normalize`` flag (WavLM Large: True) is never acted on by the reference, so it is not here either;
        # the wav2vec2 wrapper (s3prl UpstreamExpert, wav2vec2_expert.py:71-72) does normalise.
        cfg["normalize"] = bool(cfg.get("normalize", False)) and self.APPLY_NORMALIZE
        cfg["mask_prob"] = mask_prob if mask else 0.0
        cfg["mask_channel_prob"] = mask_channel_prob if mask else 0.0
        self.backbone = WavLMBackbone(cfg)
        if self.backbone.d != linear_dim:
            raise ValueError(f"linear_dim {linear_dim} must equal the backbone width {self.backbone.d}")
        self.cfg = ConformerCfg(lang2vocab=dict(lang2vocab), lang2index=dict(lang2index), n_blocks=0, encoder_dim=linear_dim,
                                last_heads=num_head, last_dim_head=dim_head, hidden_dim=hidden_dim, dropout=dropout,
                                pos_dropout=0.0, use_stochastic_depth=False, front="features")
        self._bind_engine(Engine(self.cfg, act_dtype=compute_dtype))
        # backbone parameters: reference names under model.featurizer.model.*, frozen (no gradient path exists for them)
        self._bb_names: List[str] = []
        sd = weights if weights is not None else self._random_backbone(cfg)
        for name, t in sd.items():
            *path, leaf = (self.BB_PREFIX + name).split(".")
            _child(self, path).register_parameter(leaf, nn.Parameter(t.detach().clone().float(), requires_grad=False))
            self._bb_names.append(name)
        self._sync_backbone()
        inner = self.model
        inner.lang2vocab, inner.conformer_linear = self.cfg.lang2vocab, True
        inner.loss_fns = {k: nn.CTCLoss(blank=v, reduction="none", zero_infinity=True) for k, v in lang2vocab.items()}
        inner.wer_fn = char_error_rate
        self.lang_discriminator.lang2vocab, self.lang_discriminator.lang2index = self.cfg.lang2vocab, self.cfg.lang2index
        self.lang_discriminator.classes = len(lang2vocab)
        self.forced_masks = None
        self._backbone_frozen = {"extractor": True, "encoder": True}
        self._bb_stale = False                    # encoder parameters changed since the bf16 operands were derived
        self._bb_changed = set()                  # names of the backbone parameters that took gradients since the last refresh
        # The reference's freeze_feature_extractor / freeze_tranformer_encoder (lid/WavLMMutiLangModel.py:78-104) leave WavLM's
        # own ``layer_norm`` (in front of post_extract_proj) and ``mask_emb`` trainable, so even the "frozen" regime back-
        # propagates through the whole transformer to reach them.  train_input_norm=True (default) keeps that; False stops
        # the gradient at the features (backbone forward = two captured graphs, nothing saved: the fast frozen regime).
        self.train_input_norm = bool(train_input_norm)
        for n in WavLMBackbone.INPUT_SIDE:
            dict(self.named_parameters())[self.BB_PREFIX + n].requires_grad = self.train_input_norm
        self.on_backbone_grads_ready = None       # data parallelism: called with the backbone's flat gradient arena

    @staticmethod
    def _random_backbone(cfg):
        """Seeded stand-in weights of the right names and shapes when no checkpoint is given (synthetic runs, tests)."""
        from lidk.wavlm import WavLMBackbone as _B
        shapes = _B.param_shapes(cfg)
        g = torch.Generator().manual_seed(0)
        out = {}
        for name, shape in shapes.items():
            if name.endswith(("norm.weight", "conv_layers.0.2.weight", ".2.1.weight", "grep_a", "weight_g")):
                out[name] = torch.ones(shape)
            elif len(shape) >= 2:
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
                out[name] = torch.randn(shape, generator=g) * (1.4 / fan_in ** 0.5)
            else:
                out[name] = torch.zeros(shape)
        return out

    def _backbone_params(self) -> Dict[str, torch.Tensor]:
        params = dict(self.named_parameters())
        return {n: params[self.BB_PREFIX + n].data for n in self._bb_names}

    def _sync_backbone(self):
        self.backbone.load_state_dict(self._backbone_params(), share=True)      # optimizer updates reach backbone.refresh()
        self._bb_stale = False
        self._bb_changed = set()

    def _moved(self, device):
        for n in self._bb_names:                                   # the frozen backbone's parameters follow the model
            *path, leaf = (self.BB_PREFIX + n).split(".")
            mod = _child(self, path)
            mod._parameters[leaf].data = mod._parameters[leaf].data.to(device)
        if self.MIX_NAME is not None:                              # the Featurizer's mixing logits live outside the engine arena
            mp = dict(self.named_parameters()).get(self.MIX_NAME)
            if mp is not None:
                mp.data = mp.data.to(device)
                self._mix_grad = None
        self.backbone.to(device)
        self._sync_backbone()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {k: v for k, v in state_dict.items() if not k.startswith("data_processor.")}      # torchaudio resampler buffers
        res = super().load_state_dict(sd, strict=strict, assign=False)
        self._sync_backbone()
        return res

    # ------------------------------------------------------------------ forward
    def forward(self, x, sample_rate: int = 16000, lang: str = None):
        if sample_rate != 16000:
            raise NotImplementedError("resampling (22.05/44.1 kHz) is outside the lidk path (SURVEY 2 #3)")
        wavs = list(x) if isinstance(x, (list, tuple)) else [w for w in x]
        n_samples = [int(w.shape[-1]) for w in wavs]
        wav = torch.nn.utils.rnn.pad_sequence([w.reshape(-1) for w in wavs], batch_first=True).float()
        if not wav.is_cuda:
            raise LidkError("WavLMMutiLangModel.forward: waveforms must be on the GPU (no CPU fallback)")
        grad_path = self.training and torch.is_grad_enabled() and lang is not None
        if self._bb_stale:
            # the parameters that TOOK gradients in the steps since the last refresh (recorded by _backbone_backward) - not what
            # requires_grad says now: a freeze_* call between the optimizer step and this forward must not hide an update
            self.backbone.refresh(changed=sorted(self._bb_changed))
            self._bb_stale = False
            self._bb_changed = set()
        self._bb_shape = tuple(wav.shape)
        mix_w = self._mix_w()
        train_path = grad_path and (self.train_input_norm or not self._backbone_frozen["encoder"]
                                    or not self._backbone_frozen["extractor"] or mix_w is not None)
        if grad_path and not train_path and not getattr(self, "_warned_fast_frozen", False):
            # train_input_norm=False with everything frozen is NOT a regime of the reference: its backbone stays in train() mode,
            # so hidden / attention dropouts apply even while nothing in it trains; this path replays the inference graphs
            # (span masking on, dropouts off)
            logging.warning("lidk backbone: train_input_norm=False with a frozen backbone runs the captured inference graphs - "
                            "span masking applies, the backbone's dropouts do not (the reference applies them in this case)")
            self._warned_fast_frozen = True
        with torch.no_grad():
            feats = self.backbone.forward(wav.contiguous(), mask=self.training, n_samples=n_samples, mix_w=mix_w, train=train_path)
        eng = self.lidk_engine
        if grad_path:
            if self._anchor is None or self._anchor.device != feats.device:
                self._anchor = torch.zeros(1, device=feats.device, requires_grad=True)
            logits = self._engine_apply(feats, lang, [], self.forced_masks)
            return {lang: logits}, (None, None)
        out = eng.forward(feats, lang, self.training, None, self.forced_masks)
        out = {k: v.clone() for k, v in out.items()}
        if lang is not None:
            return out, (None, None)
        return out, self.lang_discriminator_forward(out)

    def _mix_w(self):
        """Device tensor of the hidden-state mixing logits (``model.featurizer.weights``), or None (last hidden state)."""
        if self.MIX_NAME is None:
            return None
        p = dict(self.named_parameters()).get(self.MIX_NAME)
        return None if p is None else p.data

    # ------------------------------------------------------------------ reference helper surface
    def freeze_feature_extractor(self):
        self._backbone_frozen["extractor"] = True
        self.backbone.train_extractor = False
        for p in self._extractor_parameters().values():
            p.requires_grad = False

    def _encoder_parameters(self):
        params = dict(self.named_parameters())
        return {n: params[self.BB_PREFIX + n] for n in self._bb_names if n.startswith(WavLMBackbone.TRAINABLE_PREFIX)}

    def freeze_tranformer_encoder(self):                                      # noqa: F811 (reference spelling)
        self._backbone_frozen["encoder"] = True
        for p in self._encoder_parameters().values():
            p.requires_grad = False

    def _extractor_parameters(self):
        params = dict(self.named_parameters())
        return {n: params[self.BB_PREFIX + n] for n in self._bb_names if n.startswith(WavLMBackbone.EXTRACTOR_PREFIXES)
                and n in WavLMBackbone.param_shapes(self.backbone.cfg)}

    def unfreeze_feature_extractor(self):
        """lid/WavLMMutiLangModel.py:86-94: the conv feature extractor and post_extract_proj take gradients from now on (the
        reference reaches this after ``freeze_encoder_epoch``); the forward then keeps the conv stack's pre-activations."""
        self._backbone_frozen["extractor"] = False
        self.backbone.train_extractor = True
        for p in self._extractor_parameters().values():
            p.requires_grad = True

    def unfreeze_tranformer_encoder(self):
        """lid/WavLMMutiLangModel.py:106-112: the transformer encoder's parameters take gradients from now on."""
        self._backbone_frozen["encoder"] = False
        for p in self._encoder_parameters().values():
            p.requires_grad = True

    def _backbone_backward(self, dfeat: torch.Tensor):
        """Called by the autograd node after the heads' backward: encoder gradients from d(loss)/d(features)."""
        frozen = self._backbone_frozen["encoder"]
        mix_w = self._mix_w()
        if frozen and not self.train_input_norm and mix_w is None and self._backbone_frozen["extractor"]:
            return
        bb = self.backbone
        params = dict(self.named_parameters())
        live = {n: params[self.BB_PREFIX + n] for n in self._bb_names
                if params[self.BB_PREFIX + n].requires_grad}
        bb._alloc_grads()
        masked = bb.cfg.get("mask_prob", 0.0) > 0
        # zero_grad(set_to_none) since the last backward = a new optimizer step: start the arena from zero.  mask_emb never gets
        # a .grad while nothing is masked, so it must not take part in this test: with it, every micro-batch of an
        # accumulate_grad > 1 step would wipe the arena and only the last one's gradients would survive.
        if any(p.grad is None for n, p in live.items() if masked or n != "mask_emb"):
            bb.zero_grads()
        mix_dw = None
        if mix_w is not None:                      # the Featurizer's mixing logits: their gradient lives at the head of the
            mp = params[self.MIX_NAME]             # backbone's flat arena, so data parallelism averages it with the rest
            self._mix_grad = bb.mix_grad
            if mp.grad is None:
                self._mix_grad.zero_()
            mix_dw = self._mix_grad if mp.requires_grad else None
        bb.backward(dfeat, *self._bb_shape, wgrads=not frozen, mix_w=mix_w, mix_dw=mix_dw,
                    data_grads=not (frozen and not self.train_input_norm) or not self._backbone_frozen["extractor"],
                    extractor=not self._backbone_frozen["extractor"])
        if mix_dw is not None:
            params[self.MIX_NAME].grad = self._mix_grad
        for n, p in live.items():
            if n == "mask_emb" and not masked:                     # no span was replaced: the reference leaves .grad None
                continue
            p.grad = bb.grads[n]
        if self.on_backbone_grads_ready is not None:          # only the prefix of the arena this regime writes
            region = "extractor" if not self._backbone_frozen["extractor"] else "input" if frozen else "encoder"
            self.on_backbone_grads_ready(bb.grad_flat[:bb.grad_regions[region]])
        self._bb_stale = True                                      # an optimizer step follows
        self._bb_changed |= set(live)

    def keep_last_lang_model_train(self, lang):
        """lid/WavLMMutiLangModel.py:114-123 (``module.keep_train_lang``, lid/conf/xf_asr_extra_finetune.yaml:43): every head
        but ``lang``'s stops training.  A batch of a frozen language still runs through its head and back-propagates into
        the backbone; the head's own parameters get no ``.grad`` (``_publish_grads`` leaves them out), so the optimizer skips them
        and allocates no state for them - what ``requires_grad = False`` does in the reference."""
        for item_lang in self.cfg.lang2vocab:
            if item_lang == lang:
                continue
            logging.info(f"freeze lang model: {item_lang}")
            for p in self.model.last_projects[item_lang].parameters():
                p.requires_grad = False
                p.grad = None

    def reset_param(self):
        logging.info("reset parameters...")
        self.lidk_engine.reset_parameters()
