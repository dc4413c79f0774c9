"""LidModule - the reference's CCMLModule for the pretrained-backbone LID models (lid/LidModule_ASR.py:17-409): joint per-language
CTC + ASR-confidence LID on a WavLM backbone (``use_wav2vec=False``) or a wav2vec2 backbone (``use_wav2vec=True``, BASELINE config 5).  Same constructor keywords (YAML ``module:`` + ``model:``
of lid/conf/xf_asr_wavlm*.yaml), hooks, logged names and quirks; the arithmetic runs on the lidk kernels: waveform
normalisation / dither / pre-emphasis on the GPU, the WavLM backbone forward (lidk/wavlm.py), Conformer heads forward + backward
(lidk Engine), fused log-softmax + CTC, LID scoring, device-side greedy decode.

The reference launcher imports this name unconditionally (lid/main.py:13) and builds it when the YAML says ``supervised: false``.
Scope of this build: WavLM / wav2vec2 (Base architecture) with the conv feature extractor frozen (the reference's regime
until ``freeze_encoder_epoch`` = 100 passes) and the transformer frozen or fine-tuned (``freeze_tranformer_epoch``).
"""
import logging
from typing import Any, Dict, List

import numpy as np
import torch

from ccml.ccml_module import CCMLModule
from ccml.optim.multi_tensor import SGD, Adam
from ccml.optim.novograd import Novograd
from ccml.optim.tri_state import TriStageLRSchedule
from lid.ConformerLangModel import CtcLossFn
from lid.eer import CAvg, EER2
from lidk import ops as _ops


class LidModule(CCMLModule):
    def __init__(self, optimizer_name: str = "adam", optimizer_param: Dict = None, scheduler: str = "reduce",
                 scheduler_param: Dict = None, interval: int = 10, freeze_tranformer_epoch: int = 1, freeze_encoder_epoch: int = 100,
                 froze_wav2vec_model_epoch: int = 100, pt_path: str = None, feature_selection: str = "hidden_states",
                 dropout: float = 0.0, linear_dim: int = 768, mask: bool = True, num_layers: int = 1, hidden_dim: int = 128,
                 lang2vocab: Dict = None, lang2index_dict: Dict = None, tokenizer_dict: Dict = None, use_wav2vec: bool = False,
                 conformer_linear: bool = False, double_swish: bool = False, use_pre_train: bool = True,
                 mask_channel_prob: float = 0.0, mask_prob: float = 0.0, sr: int = 22050, conformer_pure: bool = False,
                 extrme_mode: bool = False, keep_train_lang: str = None, use_mask: bool = False, dim_head: int = 32,
                 num_head: int = 8, wavlm_cfg: Dict = None, train_input_norm: bool = True, *args, **kwargs):
        super().__init__(pt_path=pt_path, feature_selection=feature_selection, linear_dim=linear_dim, mask=mask,
                         num_layers=num_layers, hidden_dim=hidden_dim, lang2vocab=lang2vocab, lang2index_dict=lang2index_dict,
                         tokenizer_dict=tokenizer_dict, use_wav2vec=use_wav2vec, conformer_linear=conformer_linear,
                         double_swish=double_swish, mask_channel_prob=mask_channel_prob, mask_prob=mask_prob,
                         keep_train_lang=keep_train_lang, use_mask=use_mask, dim_head=dim_head, num_head=num_head,
                         dropout=dropout, use_pre_train=use_pre_train, sr=sr, wavlm_cfg=wavlm_cfg)
        from lid.Wav2vecMutiLangModel import Wav2vecMutiLangModel
        from lid.WavLMMutiLangModel import WavLMMutiLangModel
        self.optimizer_name, self.optimizer_param = optimizer_name, dict(optimizer_param or {})
        self.scheduler, self.scheduler_param = scheduler, dict(scheduler_param or {})
        self.lang2index_dict, self.tokenizer_dict = lang2index_dict, tokenizer_dict
        self.interval = max(1, interval)
        self.freeze_tranformer_epoch, self.freeze_encoder_epoch = freeze_tranformer_epoch, freeze_encoder_epoch
        self.froze_wav2vec_model_epoch = froze_wav2vec_model_epoch
        self.sr, self.extrme_mode, self.keep_train_lang = sr, extrme_mode, keep_train_lang
        self.index2lang_dict = {v: k for k, v in lang2index_dict.items()}
        logging.info("sample rate: %s, double swish: %s, mask channel prob %s", sr, double_swish, mask_channel_prob)
        if use_wav2vec:                    # reference :95-110 (no double_swish / use_pre_train / mask probabilities on this branch)
            self.model = Wav2vecMutiLangModel(
                pt_path=pt_path, feature_selection=feature_selection, dropout=dropout, linear_dim=linear_dim, mask=mask,
                num_layers=num_layers, lang2vocab=lang2vocab, lang2index=lang2index_dict, hidden_dim=hidden_dim,
                conformer_linear=conformer_linear, use_mask=use_mask, dim_head=dim_head, num_head=num_head,
                wav2vec_cfg=kwargs.get("wav2vec_cfg"), train_input_norm=train_input_norm)
        else:
            self.model = WavLMMutiLangModel(
                pt_path=pt_path, feature_selection=feature_selection, dropout=dropout, linear_dim=linear_dim, mask=mask,
                num_layers=num_layers, lang2vocab=lang2vocab, lang2index=lang2index_dict, hidden_dim=hidden_dim,
                conformer_linear=conformer_linear, double_swish=double_swish, use_pre_train=use_pre_train,
                mask_channel_prob=mask_channel_prob, mask_prob=mask_prob, conformer_pure=conformer_pure, use_mask=use_mask,
                dim_head=dim_head, num_head=num_head, wavlm_cfg=wavlm_cfg, train_input_norm=train_input_norm)
        self.count, self.avg_loss, self.avg_wer = 1, 0.0, 0.0
        self.predict_texts, self.label_texts, self.wer = None, None, 0.0
        self.countdown_20 = 0
        self.eer = EER2()
        self.cavg = CAvg(num_class=len(lang2index_dict))

    # ------------------------------------------------------------------ optimizer / schedule (reference :140-176)
    def config_optim(self, *args, **kwargs):
        # all parameters, like the reference (:140-150): frozen ones carry no gradient, so the optimizer skips them (and
        # allocates no state) until freeze_tranformer_epoch / freeze_encoder_epoch have passed
        params = list(self.model.parameters())
        name = self.optimizer_name
        if name == "sgd":
            optimizer = SGD(params, **self.optimizer_param)
        elif name == "adam":
            optimizer = Adam(params, **self.optimizer_param)
        elif name == "novograd":
            optimizer = Novograd(params, **self.optimizer_param)
        else:
            logging.warning("optimizer %s unknown, using SGD", name)
            optimizer = SGD(params, **self.optimizer_param)
        if self.scheduler == "reduce":
            sched = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer=optimizer, **self.scheduler_param)
            return optimizer, sched, {"monitor": "val_loss", "interval": "epoch"}
        sched = None
        if self.scheduler == "tristage":
            sched = TriStageLRSchedule(optimizer=optimizer, phase_ratio=[0.1, 0.4, 0.5], init_lr_scale=0.05, final_lr_scale=0.2,
                                       max_update=self.trainer.total_steps, lr=self.optimizer_param["lr"])
        return optimizer, sched, {"monitor": None, "interval": "step"}

    # ------------------------------------------------------------------ waveform preparation on the GPU
    def _prepare_wavs(self, wavs: List[torch.Tensor], train: bool) -> List[torch.Tensor]:
        """What the reference's DataLoader workers do per utterance on the CPU (lid/raw_datasets.py:269-280: read_audio's
        normalisation, and in training wav_augment's dither + pre-emphasis), for the whole batch on the GPU."""
        lens = [int(w.shape[-1]) for w in wavs]
        x = torch.nn.utils.rnn.pad_sequence([w.reshape(-1).float() for w in wavs], batch_first=True).contiguous()
        # (through pinned staging: torch.tensor(lens, device=cuda) blocks the host until the stream has drained - one sync per step)
        ns = _ops.upload_async(lens, torch.int32, x.device, "prepare_wavs.ns") if min(lens) != max(lens) else None
        x = _ops.normalize_wav(x, n_samples=ns)
        if train:
            y = _ops.dither_preemph(x, coef=0.97, dither=1e-5, seed=int(torch.randint(0, 2 ** 31 - 1, (1,))))
            if ns is not None:                                     # keep the zero padding behind each utterance exact
                keep = torch.arange(x.shape[1], device=x.device)[None, :] < ns[:, None]
                y = y * keep
            x = y
        return [x[i, :n] for i, n in enumerate(lens)]

    # ------------------------------------------------------------------ shared forward + loss (reference :178-223)
    def common_loop(self, batch, train_stat: bool = True) -> Dict:
        wavs, texts, wav_percents, text_percents, langs = batch[0], batch[1], batch[2], batch[3], batch[5]
        # single-language batches (Q7): the id comes from the host copy Trainer.batch_to_device keeps beside the device tensor -
        # int(langs[0]) on the device tensor is a blocking read behind everything queued, i.e. one full host-device sync per step
        lang = self.index2lang_dict[int(getattr(langs, "_host", langs)[0])]
        wavs = self._prepare_wavs(wavs, train_stat and self.model.training)
        # forward + CTCLoss(reduction='none', zero_infinity)(...).mean() (reference :185-192); in training one autograd node
        out, loss, in_len, tg_len = self.model.forward_ctc(wavs, self.sr, lang, texts, wav_percents, text_percents)
        # greedy transcripts: every step, or in extrme_mode only every 20th training step (reference :200-216)
        if self.countdown_20 == 0 or (self.predict_texts is None or not self.extrme_mode) or not train_stat:
            self.countdown_20 = 20
            tok, k = self.tokenizer_dict[lang], self.model.lidk_engine.k
            self.predict_texts = tok.ids_to_text(*k.ctc_greedy(out.detach().contiguous(), in_len.contiguous(), tok.blank_id))
            self.label_texts = tok.decoder(texts, target_lengths=tg_len)
            self.wer = float(self.model.model.wer_fn(self.predict_texts, self.label_texts))
        self.countdown_20 -= 1
        return {"loss": loss, "wer": self.wer, "lang": lang, "predict_texts": self.predict_texts, "label_texts": self.label_texts,
                "wavs": wavs}

    def infer(self, x: torch.Tensor, sample_rate: int = 16000, language: str = None):
        """x (1, L) prepared waveform on the GPU -> ({lang: [text]}, lid_asr (1, C), logits dict)  (reference :241-251)."""
        with torch.no_grad():
            out, (lid_asr, _) = self.model([x[0, :]], sample_rate, language)
        k = self.model.lidk_engine.k
        texts = {l: self.tokenizer_dict[l].ids_to_text(*k.ctc_greedy(o.contiguous(), None, self.tokenizer_dict[l].blank_id))
                 for l, o in out.items()}
        return texts, lid_asr, out

    def _log_running(self, out, key, stage, only_tbar=False):
        loss = float(out["loss"].detach())
        if not np.isnan(loss):
            self.avg_loss = 0.98 * self.avg_loss + 0.02 * loss
            self.avg_wer = 0.98 * self.avg_wer + 0.02 * out["wer"]
            self.count += 1
            corr = 1 - np.power(0.98, self.count)
            self.trainer.logger.log(data={"loss": self.avg_loss / corr, key: self.avg_wer / corr}, progress=True, stage=stage,
                                    **({"only_tbar": True} if only_tbar else {}))

    def train_loop(self, batch):
        out = self.common_loop(batch)
        # host-sync policy (as in LidSuperviseModule): the reference reads loss.item() every step; here the running averages are
        # refreshed every `interval` steps, so the host keeps issuing work ahead of the GPU in between
        if self.trainer.current_step % self.interval == self.interval - 1:
            logging.info("wer %.4f | predict: %s | label: %s", out["wer"], out["predict_texts"][0], out["label_texts"][0])
            self._log_running(out, "tr_wer", "train")
        return {"loss": out["loss"], "wer": out["wer"]}

    def before_train_loop(self, value):
        self.count, self.avg_loss, self.avg_wer = 1, 0.0, 0.0
        epoch = self.trainer.current_epoch
        if epoch <= self.freeze_encoder_epoch:
            self.model.freeze_feature_extractor()
        else:
            self.model.unfreeze_feature_extractor()
        if epoch <= self.freeze_tranformer_epoch:
            self.model.freeze_tranformer_encoder()
        else:
            self.model.unfreeze_tranformer_encoder()
        if self.keep_train_lang is not None:
            self.model.keep_last_lang_model_train(self.keep_train_lang)

    def train_loop_end(self, outputs: List[Any] = None):
        self.count, self.avg_loss, self.avg_wer = 1, 0.0, 0.0
        if not outputs:
            return
        data = {"tr_avg_loss": float(torch.stack([o["loss"].float() for o in outputs]).mean()),
                "tr_avg_wer": float(np.mean([o["wer"] for o in outputs]))}
        logging.info("train epoch: %s", data)
        self.trainer.logger.log(data=data, progress=False, stage="val", commit=False, only_tbar=False)

    # ------------------------------------------------------------------ validation (reference :300-353)
    def val_loop(self, batch):
        out = self.common_loop(batch, False)
        if self.count % self.interval == self.interval - 1:
            logging.info("wer %.4f | predict: %s | label: %s", out["wer"], out["predict_texts"][0], out["label_texts"][0])
        self._log_running(out, "val_wer", "val", only_tbar=True)
        # As the reference does: the LID score of the FIRST utterance of the batch, all heads, and - its quirk - the metrics
        # are updated with the PREDICTED index as the target; "lang_corr" compares the prediction with the batch's language.
        _, lid_asr, _ = self.infer(out["wavs"][0].reshape(1, -1), 16000)
        index = int(torch.argmax(lid_asr, dim=-1)[0])
        prob = [(-1 / (s - 1e-9)) for s in lid_asr[0].tolist()]
        tot = sum(prob)
        prob = [p / tot for p in prob]
        self.eer.update([prob], [index])
        self.cavg.update([prob], [index])
        return {"val_loss": out["loss"].detach(), "val_wer": out["wer"], "predict_texts": out["predict_texts"],
                "label_texts": out["label_texts"], "lang_corr": self.index2lang_dict[index] == self.index2lang_dict[int(batch[5][0])]}

    def val_loop_end(self, outputs: List[Any] = None):
        preds, labels, total, corr = [], [], 0.0, 0
        for item in outputs or []:
            preds.extend(item["predict_texts"])
            labels.extend(item["label_texts"])
            if torch.isnan(item["val_loss"]).item():
                logging.warning("val loss is nan, ignored")
                continue
            total += float(item["val_loss"])
            corr += int(item["lang_corr"])
        n = max(len(outputs or []), 1)
        data = {"val_loss": total / n, "val_acc": corr / n, "val_wer": float(self.model.model.wer_fn(preds, labels)) if preds else 0.0,
                "epoch": self.trainer.current_epoch, "eer": self.eer.compute(), "cavg": self.cavg.compute()}
        self.eer.reset()
        self.cavg.reset()
        self.last_val = data
        self.trainer.logger.log(data=data, progress=True, stage="val", commit=False, only_tbar=False)
        logging.info("epoch %d: %s", self.trainer.current_epoch, data)
        self.trainer.logger.remove_key(["loss", "wer"])

    def test_loop(self, batch):
        return self.val_loop(batch)

    def test_loop_end(self, outputs: List[Any] = None):
        return self.val_loop_end(outputs)
