"""ConformerMutiLangModel — the reference's model surface (lid/ConformerLangModel.py:16-129) over the lidk HIP engine.

What is kept: constructor keywords, ``forward(x, sample_rate=16000, lang=None) -> ({lang: logits}, (lid_asr, lid_linear))``,
attribute paths used by callers (``model.model.featurizer``, ``model.model.last_projects[lang]``, ``model.model.loss_fns``,
``model.model.wer_fn``, ``model.lang_discriminator``, ``freeze_*`` / ``reset_param``) and every ``state_dict`` key and shape,
so checkpoints interchange with the reference.

What is different: there are no torch layers.  The module tree below is a naming skeleton whose Parameters are VIEWS into the
engine's flat f32 arena; ``forward`` hands the batch to ``lidk.Engine`` (hand-written HIP kernels) and autograd sees one
node for the whole network, whose backward runs the engine's explicit backward pass and publishes ``p.grad`` as views of the
flat gradient arena (tensors that took no part in the step keep ``grad None``, like the reference: SURVEY Q5-Q7).
"""
import logging
import math
import random
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from lidk import ops as _ops
from lidk.engine import Engine
from lidk.layout import ConformerCfg
from lidk._lib import LidkError


class _Node(nn.Module):
    """Naming skeleton: holds parameters / buffers / children, computes nothing."""

    def forward(self, *a, **k):
        raise LidkError("sub-modules of the lidk Conformer are a naming skeleton; call the top-level model")


def _child(root: nn.Module, parts: List[str]) -> nn.Module:
    node = root
    for part in parts:
        nxt = node._modules.get(part)
        if nxt is None:
            nxt = nn.ModuleDict() if (part == "last_projects") else _Node()
            node.add_module(part, nxt)
        node = nxt
    return node


class _EngineFn(torch.autograd.Function):
    """Autograd node for the whole network: forward = Engine.forward, backward = Engine.backward."""

    @staticmethod
    def forward(ctx, anchor, feats, model, lang, keep, masks, ctc=None):
        """ctc = (texts, wav_percents, text_percents) (``forward_ctc``): the mean CTC loss of the module's training loop rides
        in the same node - outputs (logits, loss, in_len, tg_len), only the loss differentiable."""
        eng = model.lidk_engine
        out = eng.forward(feats, lang, True, keep, masks)[lang]
        ctx.model, ctx.lang, ctx.keep, ctx.ctc = model, lang, keep, False
        if ctc is None:
            return out
        res = eng.ctc_forward(out, ctc[0], ctc[1], ctc[2], model.cfg.lang2vocab[lang])
        if res is None:                                  # lattice too large for the fused kernels: the caller takes CtcLossFn
            return out
        loss, in_len, tg_len = res
        ctx.ctc = True
        ctx.mark_non_differentiable(out, in_len, tg_len)
        ctx.set_materialize_grads(False)                 # no zero-filled stand-ins for the three outputs that carry no gradient
        return out, loss, in_len, tg_len

    @staticmethod
    def backward(ctx, dlogits, gloss=None, _g_in=None, _g_tg=None):
        model = ctx.model
        dfeat = model.lidk_engine.backward(None, ctc_gscale=gloss) if ctx.ctc else model.lidk_engine.backward(dlogits)
        model._publish_grads(ctx.lang, ctx.keep)
        if dfeat is not None and hasattr(model, "_backbone_backward"):      # features front: the backbone's own backward
            model._backbone_backward(dfeat)
        return None, None, None, None, None, None, None


class CtcLossFn(torch.autograd.Function):
    """Fused log_softmax + CTC (blank = V, reduction='none', zero_infinity) on logits (B,T,V+1):
    the reference's ``loss_fns[lang](log_softmax(out).transpose(1,0), ...)`` (lid/LidModule_ASR_Supervised.py:162-167)."""

    @staticmethod
    def forward(ctx, logits, targets, in_len, tg_len, blank, kernels):
        B, T, V1 = logits.shape
        loss = torch.empty(B, device=logits.device, dtype=torch.float32)
        dl = torch.empty_like(logits)
        ws = torch.empty(max(kernels.ctc_workspace_bytes(B, T, V1, targets.shape[1]) // 4, 1), device=logits.device)
        kernels.ctc_loss(logits.contiguous(), targets.contiguous(), in_len.contiguous(), tg_len.contiguous(), loss, dl, ws,
                         blank, 1.0, True)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (dl,) = ctx.saved_tensors
        return dl * gout.view(-1, 1, 1), None, None, None, None, None


def char_error_rate(preds: List[str], refs: List[str]) -> torch.Tensor:
    """Character error rate = sum of edit distances / sum of reference lengths (what torchmetrics.CharErrorRate computes
    for the reference's ``wer_fn``, lid/ConformerLangModel.py:268-270)."""
    errs = total = 0
    for p, r in zip(preds, refs):
        prev = list(range(len(r) + 1))
        for i, cp in enumerate(p, 1):
            cur = [i]
            for j, cr in enumerate(r, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (cp != cr)))
            prev = cur
        errs += prev[len(r)]
        total += len(r)
    return torch.tensor(errs / max(total, 1), dtype=torch.float32)


class _EngineBoundModel(nn.Module):
    """nn.Module whose Parameters are views into a lidk Engine's flat arena (shared by the Conformer and WavLM LID models):
    naming skeleton, device moves, state-dict loading, gradient publication and the LangDiscriminator."""

    def _bind_engine(self, engine: Engine):
        self.lidk_engine = engine
        self._owner: Dict[str, tuple] = {}
        for s in engine.specs:
            *path, leaf = s.name.split(".")
            mod = _child(self, path)
            mod.register_parameter(leaf, nn.Parameter(engine.pview(s.name)))
            self._owner[s.name] = (mod, leaf)
        for name, t in engine.buffers.items():
            *path, leaf = name.split(".")
            mod = _child(self, path)
            mod.register_buffer(leaf, t)
            self._owner[name] = (mod, leaf)
        self._tag_params()
        self._anchor = None

    # ------------------------------------------------------------------ arena <-> Parameter binding
    def _tag_params(self):
        eng = self.lidk_engine
        for s in eng.specs:
            mod, leaf = self._owner[s.name]
            p = mod._parameters[leaf]
            p.data = eng.pview(s.name)
            p._lidk_engine, p._lidk_tid = eng, s.tid
        for name, t in eng.buffers.items():
            mod, leaf = self._owner[name]
            mod._buffers[leaf] = t

    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, dtype=torch.float32, device=self.lidk_engine.flat.device))
        if probe.dtype != torch.float32:
            raise LidkError("master parameters are always float32; choose the compute dtype with set_compute_dtype()")
        if probe.device != self.lidk_engine.flat.device or not self.lidk_engine._built:
            self.lidk_engine.to(probe.device)
            self._tag_params()
            self._anchor = None
            self._moved(probe.device)
        return self

    def _moved(self, device):
        pass

    def _engine_apply(self, feats, lang, keep, masks):
        """The autograd node of the whole network; with a pending ``forward_ctc`` request the mean CTC loss rides in it."""
        req = getattr(self, "_ctc_request", None)
        res = _EngineFn.apply(self._anchor, feats, self, lang, keep, masks, req)
        if isinstance(res, tuple):
            self._ctc_result = res[1:]
            return res[0]
        return res

    def forward_ctc(self, x, sample_rate: int, lang: str, texts, wav_percents, text_percents):
        """``model(x, sample_rate, lang)`` + the training loop's loss (lid/LidModule_ASR_Supervised.py:160-168: CTCLoss(reduction=
        'none', zero_infinity)(log_softmax(out).T, texts, (T * wav_percents).long(), (L * text_percents).long()).mean()) as ONE
        autograd node: -> (logits (B, T, V+1), loss 0-dim, in_len, tg_len).  Same numbers as the two-step route (forward, then
        ``CtcLossFn`` + ``.mean()``), which is what runs in eval mode, without gradients, on the CPU test backend and for lattices
        beyond the fused kernels' LDS budget."""
        blank = self.cfg.lang2vocab[lang]
        self._ctc_request, self._ctc_result = None, None
        if self.training and torch.is_grad_enabled() and self.lidk_engine.ctc_supported():
            self._ctc_request = (texts, wav_percents, text_percents)
        try:
            out, _ = self.forward(x, sample_rate, lang)
        finally:
            self._ctc_request = None
        out = out[lang]
        if self._ctc_result is not None:
            (loss, in_len, tg_len), self._ctc_result = self._ctc_result, None
            return out, loss, in_len, tg_len
        in_len = (out.shape[1] * wav_percents).long()
        tg_len = (texts.shape[-1] * text_percents).long()
        per_utt = CtcLossFn.apply(out, texts, in_len, tg_len, blank, self.lidk_engine.k)
        return out, per_utt.mean(), in_len, tg_len

    def set_compute_dtype(self, dtype):
        """bf16 (production) or f32 (parity mode); must be chosen before the model is moved to the GPU."""
        eng = self.lidk_engine
        if eng.act_dtype != dtype:
            if eng._built:
                raise LidkError("set_compute_dtype after the engine was built")
            eng.act_dtype = dtype

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        res = super().load_state_dict(state_dict, strict=strict, assign=False)
        if self.lidk_engine._built:
            self.lidk_engine.refresh_weights()
        return res

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=True)
        if self.lidk_engine.grad is not None:
            self.lidk_engine.zero_grad()

    def _publish_grads(self, lang: str, keep: List[bool]):
        eng = self.lidk_engine
        for tid in eng.active_tensor_ids(lang, keep):
            s = eng.specs[tid]
            mod, leaf = self._owner[s.name]
            if mod._parameters[leaf].requires_grad:             # a frozen head (keep_last_lang_model_train): .grad stays None
                mod._parameters[leaf].grad = eng.gview(s.name)

    def lang_discriminator_forward(self, logits: Dict[str, torch.Tensor]):
        """LangDiscriminator.forward (lid/ConformerLangModel.py:383-395): per-language CTC-path confidence, then the MLP."""
        cfg, eng = self.cfg, self.lidk_engine
        first = next(iter(logits.values()))
        scores = torch.zeros(first.shape[0], len(cfg.lang2vocab), device=first.device, dtype=torch.float32)
        for lang, lg in logits.items():
            eng.k.lid_score(lg.contiguous(), scores[:, cfg.lang2index[lang]:], len(cfg.lang2vocab), cfg.lang2vocab[lang])
        pv = eng.pview
        lin = torch.empty_like(scores)
        eng.k.lid_mlp(scores, pv("lang_discriminator.linear.0.weight"), pv("lang_discriminator.linear.0.bias"),
                      pv("lang_discriminator.linear.2.weight"), pv("lang_discriminator.linear.2.bias"), lin)
        return scores, lin


class ConformerMutiLangModel(_EngineBoundModel):
    def __init__(self, num_layers: int = 1, lang2vocab: Dict = None, lang2index: Dict = None, hidden_dim: int = 32,
                 use_cer: bool = True, conformer_linear: bool = False, dropout: float = 0.0, linear_dim: int = 144,
                 n_blocks: int = 14, n_mels: int = 80, encoder_dim: int = 144, dim_head=64, last_dim_head: int = 32, heads=4,
                 ff_mult=4, conv_expansion_factor=2, conv_kernel_size=31, attn_dropout=0.0, ff_dropout=0.0, conv_dropout=0.0,
                 double_swish=False, sub_sampling: int = 2, compute_dtype=torch.bfloat16, **_ignored):
        super().__init__()
        if not conformer_linear:
            raise NotImplementedError("LSTM heads (conformer_linear=False) are outside the lidk hot path (SURVEY 2 #3)")
        if sub_sampling != 2 or double_swish or attn_dropout or ff_dropout or conv_dropout:
            raise NotImplementedError("lidk builds the configuration the lid confs select: sub_sampling=2, Swish, and zero "
                                      "attention/ff/conv dropout (SURVEY 2 #2)")
        if linear_dim != encoder_dim:
            raise ValueError("linear_dim must equal encoder_dim")
        self.cfg = ConformerCfg(lang2vocab=dict(lang2vocab), lang2index=dict(lang2index), n_blocks=n_blocks, n_mels=n_mels,
                                encoder_dim=encoder_dim, dim_head=dim_head, heads=heads, ff_mult=ff_mult,
                                conv_expansion_factor=conv_expansion_factor, conv_kernel_size=conv_kernel_size,
                                last_dim_head=last_dim_head, hidden_dim=hidden_dim, dropout=dropout)
        self._bind_engine(Engine(self.cfg, act_dtype=compute_dtype))
        self.use_stochastic_depth = self.cfg.use_stochastic_depth
        self.stochastic_depth_p = self.cfg.stochastic_depth_p
        self.forced_masks: Optional[Dict[str, torch.Tensor]] = None       # parity tests only
        inner = self.model
        inner.lang2vocab = self.cfg.lang2vocab
        inner.conformer_linear = True
        inner.loss_fns = {k: nn.CTCLoss(blank=v, reduction="none", zero_infinity=True) for k, v in lang2vocab.items()}
        inner.wer_fn = char_error_rate
        self.lang_discriminator.lang2vocab, self.lang_discriminator.lang2index = self.cfg.lang2vocab, self.cfg.lang2index
        self.lang_discriminator.classes = len(lang2vocab)

    # ------------------------------------------------------------------ forward
    def _layer_keep(self) -> List[bool]:
        """Stochastic depth decisions (lid/conformer.py:460-466).  Single process: the reference's own stream (the global
        python ``random``).  Data parallel: every rank must skip the SAME blocks (a skipped block's stage never issues its
        gradient all-reduce), so the draws come from a dedicated generator keyed by (seed, training-forward count) that
        nothing else - collate, callbacks, an uneven loader - can advance on one rank only."""
        n = self.cfg.n_blocks
        if not (self.training and self.use_stochastic_depth):
            return [True] * n
        eng = self.lidk_engine
        rng = random
        if eng.on_stage_grads_ready is not None or eng.stat_allreduce is not None:
            self._train_forwards = getattr(self, "_train_forwards", 0) + 1
            rng = random.Random(eng.seed * 1000003 + self._train_forwards)
        return [rng.random() <= 1 - ((i + 1) / n) * (1 - self.stochastic_depth_p) for i in range(n)]

    def features(self, x):
        if hasattr(x, "to_mel"):                      # lid.audio_processor.WaveBatch: features computed on the GPU
            return x.to_mel()
        if isinstance(x, (list, tuple)):
            x = torch.nn.utils.rnn.pad_sequence(list(x), batch_first=True)
        return x

    def forward(self, x, sample_rate: int = 16000, lang: str = None):
        if sample_rate != 16000:
            raise NotImplementedError("resampling (22.05/44.1 kHz) is outside the lidk hot path (SURVEY 2 #3)")
        feats = self.features(x)
        eng = self.lidk_engine
        if self.training and torch.is_grad_enabled() and lang is not None:
            if self._anchor is None or self._anchor.device != feats.device:
                self._anchor = torch.zeros(1, device=feats.device, requires_grad=True)
            keep = self._layer_keep()
            logits = self._engine_apply(feats, lang, keep, self.forced_masks)
            return {lang: logits}, (None, None)
        out = eng.forward(feats, lang, self.training, self._layer_keep() if self.training else None, self.forced_masks)
        out = {k: v.clone() for k, v in out.items()}          # the engine reuses its logits buffers on the next call
        if lang is not None:
            return out, (None, None)
        return out, self.lang_discriminator_forward(out)

    # ------------------------------------------------------------------ reference helper surface
    def freeze_feature_extractor(self):
        pass        # the reference only freezes wav2vec-style featurizers (hasattr(featurizer, "model")): no-op for Conformer

    unfreeze_feature_extractor = freeze_tranformer_encoder = unfreeze_tranformer_encoder = freeze_feature_extractor

    def reset_param(self):
        logging.info("reset parameters...")
        self.lidk_engine.reset_parameters()
