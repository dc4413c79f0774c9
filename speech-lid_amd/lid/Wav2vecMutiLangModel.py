"""Wav2vecMutiLangModel - the reference's wav2vec2-backbone LID model surface (lid/Wav2vecMutiLangModel.py:18-260) on the lidk
kernels (BASELINE config 5).

  DataProcessor (16 kHz pass-through) -> Wav2vecMutiModel: s3prl ``Featurizer`` over an ``UpstreamExpert`` wrapping fairseq's
  ``Wav2Vec2Model`` (lid/s3prl_updream/wav2vec/wav2vec2_expert.py:40-92, wav2vec2.py:541-640,850-1078) -> per-language
  ConformerLinear heads -> LangDiscriminator.

What the backbone computes (the WavLM kernels, lidk/wavlm.py, configured for wav2vec2): conv feature extractor, LayerNorm,
post_extract_proj, span masking in training, padded frames zeroed, positional convolution, post-LN transformer layers with plain
scaled-dot-product attention whose KEYS beyond each utterance's conv-output length are masked (the padding mask wav2vec2 - unlike
the reference's WavLM call - hands its encoder), attention / hidden dropouts and LayerDrop.  ``feature_selection``:
``"last_hidden_state"`` = the encoder output; anything else (the reference's default ``"hidden_states"``) = s3prl's weighted sum:
softmax(``model.featurizer.weights``) over the L + 1 hidden states (the input of every layer and the encoder output,
interfaces.py:227-252), a trainable parameter of this model.

State-dict names: ``model.featurizer.upstream.model.<fairseq Wav2Vec2Model key>``, ``model.featurizer.weights``,
``model.last_projects.<lang>.*``, ``lang_discriminator.linear.{0,2}.*``.  Checkpoint: fairseq checkpoints pickle an omegaconf
config and need fairseq to open; neither is in this image, so ``pt_path`` takes the plain form ``{"cfg": dict of the model config,
"model": state_dict}`` (one ``torch.save`` away from a fairseq checkpoint on a machine that has fairseq).  Pre-training-only tensors
of such a state dict (quantizer, project_q, final_proj) are carried but unused.

Parity: the transformer arithmetic is pinned through the reference's own lid/wavlm/WavLM.py encoder with the relative-position
bias off and a padding mask on (tests/golden/w2v2_*.npz); what lives only in the un-vendored fairseq (its compute_mask_indices
draw order, MultiheadAttention internals) is restated and UNPINNED.  Built: wav2vec2 Base (extractor_mode=default,
conv_bias=False, layer_norm_first=False, normalize=False) and the Large / XLS-R variant every wav2vec conf of the reference loads
(lid/conf/xf_asr_wav2vec.yaml:12 ``xlsr2_300m.pt``: extractor_mode=layer_norm with conv bias, layer_norm_first, ``task.normalize``
= per-utterance layer-norm of the waveform, d = 1024 / 16 heads / ffn 4096 / 24 layers), pinned by tests/golden/xlsr_step.npz - a
run of the reference's own lid/wavlm/WavLM.py classes with those flags (oracle/gen_golden_xlsr.py).  ``normalize`` is read from
the checkpoint cfg, flat (``cfg["normalize"]``) or in fairseq's nesting (``cfg["task"]["normalize"]`` beside ``cfg["model"]``)."""
from typing import Dict, Optional

import torch
import torch.nn as nn

from lid.ConformerLangModel import _child
from lid.WavLMMutiLangModel import WavLMMutiLangModel


class Wav2vecMutiLangModel(WavLMMutiLangModel):
    BB_PREFIX = "model.featurizer.upstream.model."
    MIX_NAME = "model.featurizer.weights"
    APPLY_NORMALIZE = True

    def __init__(self, pt_path: str = None, feature_selection: str = "hidden_states", dropout: float = 0.0, linear_dim: int = 768,
                 mask: bool = True, num_layers: int = 1, lang2vocab: Dict = None, lang2index: Dict = None, hidden_dim: int = 128,
                 conformer_linear: bool = False, use_mask: bool = False, dim_head: int = 32, num_head: int = 8,
                 compute_dtype=torch.bfloat16, wav2vec_cfg: Optional[Dict] = None, train_input_norm: bool = True, **_ignored):
        if pt_path is not None:
            ckpt = torch.load(pt_path, map_location="cpu", weights_only=False)
            if not (isinstance(ckpt, dict) and isinstance(ckpt.get("cfg"), dict) and "model" in ckpt):
                raise ValueError("Wav2vecMutiLangModel: pt_path must hold {'cfg': dict, 'model': state_dict} (a fairseq checkpoint "
                                 "re-saved without its omegaconf object; fairseq is not available to this build)")
            cfg, weights = dict(ckpt["cfg"]), ckpt["model"]
        elif wav2vec_cfg is not None:
            cfg, weights = dict(wav2vec_cfg), None
        else:
            raise ValueError("Wav2vecMutiLangModel needs pt_path or wav2vec_cfg")
        if isinstance(cfg.get("model"), dict):             # the nested fairseq form {"model": {...}, "task": {"normalize": ...}}
            task = cfg.get("task") or {}
            cfg = dict(cfg["model"], normalize=bool(task.get("normalize", cfg["model"].get("normalize", False))))
        if cfg.get("relative_position_embedding", False) or cfg.get("gru_rel_pos", False):
            raise ValueError("a wav2vec2 config has no relative position embedding")
        cfg["encoder_padding_mask"] = True
        # load_wav2vec2_for_finetune (wav2vec2_expert.py:196-215) overrides the checkpoint's masking with fixed values -
        # mask_prob 0.2, mask_channel_prob 0.2, mask_channel_length 64 (the YAML's model.mask_prob / mask_channel_prob never reach
        # the wav2vec2 model: lid/LidModule_ASR.py:95-110 does not pass them) - and switches LayerDrop off unless
        # feature_selection == "last_hidden_state" (drop_layer, lid/Wav2vecMutiLangModel.py:186).
        mask_prob, mask_channel_prob = 0.2, 0.2
        cfg["mask_channel_length"] = 64
        if feature_selection != "last_hidden_state":
            cfg["encoder_layerdrop"] = -1.0
        super().__init__(pt_path=None, feature_selection=feature_selection, dropout=dropout, linear_dim=linear_dim, mask=mask,
                         num_layers=num_layers, lang2vocab=lang2vocab, lang2index=lang2index, hidden_dim=hidden_dim,
                         conformer_linear=conformer_linear, use_mask=use_mask, dim_head=dim_head, num_head=num_head,
                         compute_dtype=compute_dtype, wavlm_cfg=cfg, train_input_norm=train_input_norm, mask_prob=mask_prob,
                         mask_channel_prob=mask_channel_prob, _weights=weights)
        self.feature_selection = feature_selection
        if feature_selection != "last_hidden_state":                    # Featurizer: learned mix of the L + 1 hidden states
            _child(self, ["model", "featurizer"]).register_parameter("weights", nn.Parameter(torch.zeros(self.backbone.n_layers + 1)))

    def _mix_w(self):
        return None if self.feature_selection == "last_hidden_state" else super()._mix_w()

    # lid/Wav2vecMutiLangModel.py:108-114: every parameter under self.model (backbone, mixing weights AND heads)
    def froze_wav2vec_model(self):
        for p in self.model.parameters():
            p.requires_grad = False

    def unfroze_wav2vec_model(self):
        for p in self.model.parameters():
            p.requires_grad = True
