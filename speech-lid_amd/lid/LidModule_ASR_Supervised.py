"""LidSuperviseModule — the Conformer ASR-based LID task behind the reference's CCMLModule surface
(lid/LidModule_ASR_Supervised.py:14-383): same constructor keywords (YAML ``module:`` + ``model:``), same hooks and logged
metric names.  Arithmetic goes through the lidk engine: model forward/backward, fused log-softmax + CTC, LID scoring.

Host-sync policy (the reference syncs three times per step): the language id is read once per batch (it selects
which head runs, so the host needs it); greedy decode + CER are evaluated every ``interval`` steps instead of every step
(``wer`` between evaluations repeats the last value); the running-loss EMA uses device tensors.
"""
import logging
from typing import Any, Dict, List, Tuple

import numpy as np
import torch

from ccml.ccml_module import CCMLModule
from ccml.optim.multi_tensor import SGD, Adam
from ccml.optim.novograd import Novograd
from ccml.optim.tri_state import TriStageLRSchedule
from lid.ConformerLangModel import ConformerMutiLangModel, CtcLossFn
from lid.eer import CAvg, EER2


class LidSuperviseModule(CCMLModule):
    def __init__(self, optimizer_name: str = "adam", optimizer_param: Dict = None, scheduler: str = "reduce",
                 scheduler_param: Dict = None, interval: int = 10, lang2index_dict: Dict = None, tokenizer_dict: Dict = None,
                 lang2vocab: Dict = None, num_layers: int = 1, hidden_dim: int = 32, use_cer: bool = True,
                 conformer_linear: bool = True, dropout: float = 0.0, linear_dim: int = 144, n_blocks: int = 14, sr=16000,
                 n_mels: int = 80, encoder_dim: int = 144, dim_head=64, last_dim_head: int = 32, heads=4, ff_mult=4,
                 conv_expansion_factor=2, conv_kernel_size=31, attn_dropout=0.0, ff_dropout=0.0, conv_dropout=0.0,
                 double_swish=False, sub_sampling: int = 2, *args, **kwargs):
        super().__init__(lang2vocab=lang2vocab, lang2index_dict=lang2index_dict, tokenizer_dict=tokenizer_dict,
                         num_layers=num_layers, hidden_dim=hidden_dim, conformer_linear=conformer_linear, linear_dim=linear_dim,
                         n_blocks=n_blocks, n_mels=n_mels, encoder_dim=encoder_dim, dim_head=dim_head,
                         last_dim_head=last_dim_head, heads=heads, ff_mult=ff_mult, conv_expansion_factor=conv_expansion_factor,
                         conv_kernel_size=conv_kernel_size, double_swish=double_swish, lang2index=lang2index_dict,
                         sub_sampling=sub_sampling, dropout=dropout, optimizer_name=optimizer_name,
                         optimizer_param=optimizer_param, scheduler=scheduler, scheduler_param=scheduler_param,
                         interval=interval, sr=sr)
        self.optimizer_name, self.optimizer_param = optimizer_name, dict(optimizer_param or {})
        self.scheduler, self.scheduler_param = scheduler, dict(scheduler_param or {})
        self.lang2index_dict, self.tokenizer_dict = lang2index_dict, tokenizer_dict
        self.interval, self.sr = max(1, interval), sr
        self.index2lang_dict = {v: k for k, v in lang2index_dict.items()}
        self.model = ConformerMutiLangModel(
            num_layers=num_layers, lang2vocab=lang2vocab, lang2index=lang2index_dict, hidden_dim=hidden_dim, use_cer=use_cer,
            conformer_linear=conformer_linear, dropout=dropout, linear_dim=linear_dim, n_blocks=n_blocks, n_mels=n_mels,
            encoder_dim=encoder_dim, dim_head=dim_head, last_dim_head=last_dim_head, heads=heads, ff_mult=ff_mult,
            conv_expansion_factor=conv_expansion_factor, conv_kernel_size=conv_kernel_size, attn_dropout=attn_dropout,
            ff_dropout=ff_dropout, conv_dropout=conv_dropout, double_swish=double_swish, sub_sampling=sub_sampling)
        self.count, self.avg_loss, self.avg_wer = 1, 0.0, 0.0
        self._last_wer = 0.0
        self._calls = 0
        self.eer = EER2()
        self.cavg = CAvg(num_class=len(lang2index_dict))

    # ------------------------------------------------------------------ optimizer / schedule
    def config_optim(self, *args, **kwargs):
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
        if self.scheduler == "tristage":
            sched = TriStageLRSchedule(optimizer=optimizer, phase_ratio=[0.1, 0.4, 0.5], init_lr_scale=0.05,
                                       final_lr_scale=0.02, max_update=self.trainer.total_steps, lr=self.optimizer_param["lr"])
            return optimizer, sched, {"monitor": None, "interval": "step"}
        return optimizer, None, None

    # ------------------------------------------------------------------ shared forward + loss
    def common_loop(self, batch, with_text: bool = True) -> Dict:
        wavs, texts, wav_percents, text_percents, langs = batch[0], batch[1], batch[2], batch[3], batch[5]
        lang_id = getattr(wavs, "lang_id", None)                          # host-side metadata: no device sync
        lang = self.index2lang_dict[int(getattr(langs, "_host", langs)[0]) if lang_id is None else lang_id]   # single-language batches (Q7)
        # forward + CTCLoss(reduction='none', zero_infinity)(...).mean() (reference :160-168); in training one autograd node
        out, loss, in_len, tg_len = self.model.forward_ctc(wavs, self.sr, lang, texts, wav_percents, text_percents)
        res = {"loss": loss, "wer": self._last_wer, "lang": lang, "predict_texts": [], "label_texts": []}
        if with_text:
            tok = self.tokenizer_dict[lang]
            k = self.model.lidk_engine.k                     # greedy collapse on the device; the host only maps ids to symbols
            pred = tok.ids_to_text(*k.ctc_greedy(out.detach().contiguous(), in_len.contiguous(), tok.blank_id))
            label = tok.decoder(texts, target_lengths=tg_len)
            self._last_wer = float(self.model.model.wer_fn(pred, label))
            res.update(wer=self._last_wer, predict_texts=pred, label_texts=label)
        return res

    def train_loop(self, batch):
        self._calls += 1
        verbose = self._calls % self.interval == 0
        out = self.common_loop(batch, with_text=verbose)
        if verbose:
            loss = float(out["loss"].detach())
            if not np.isnan(loss):
                self.avg_loss = 0.98 * self.avg_loss + 0.02 * loss
                self.avg_wer = 0.98 * self.avg_wer + 0.02 * out["wer"]
                self.count += 1
                corr = 1 - np.power(0.98, self.count)
                self.trainer.logger.log(data={"loss": self.avg_loss / corr, "tr_wer": self.avg_wer / corr}, progress=True,
                                        stage="train")
            if out["predict_texts"]:
                logging.info("wer %.4f | predict: %s | label: %s", out["wer"], out["predict_texts"][0], out["label_texts"][0])
        return {"loss": out["loss"], "wer": out["wer"]}

    def before_train_loop(self, value):
        self.count, self.avg_loss, self.avg_wer = 1, 0.0, 0.0

    def train_loop_end(self, outputs: List[Any] = None):
        self.count, self.avg_loss, self.avg_wer = 1, 0.0, 0.0
        if not outputs:
            return
        losses = torch.stack([o["loss"].float() for o in outputs])
        data = {"tr_avg_loss": float(losses[~torch.isnan(losses)].mean()), "tr_avg_wer": float(np.mean([o["wer"] for o in outputs]))}
        logging.info("train epoch: %s", data)
        self.trainer.logger.log(data=data, progress=False, stage="val", commit=False, only_tbar=False)

    # ------------------------------------------------------------------ validation: CTC loss + per-utterance LID scores
    def lid_scores(self, feats: torch.Tensor) -> torch.Tensor:
        """(B, F, n_mels) -> (B, C) ASR-confidence LID scores, all language heads (eval mode)."""
        was = self.model.training
        self.model.eval()
        with torch.no_grad():
            _, (lid_asr, _) = self.model(feats, self.sr, None)
        self.model.train(was)
        return lid_asr

    def infer_tensor(self, x, sr: int, language: str = None, device=None):
        """x: (1, L) waveform on the GPU.  -> ({lang: text}, lid_asr (1, C), logits dict) like the reference's infer."""
        from lid.audio_processor import wav2mel
        feats = wav2mel(x, win_length=0.025, hop_length=0.01, n_mels=80, n_fft=512, pad=0, sr=sr).transpose(1, 2).contiguous()
        with torch.no_grad():
            out, (lid_asr, _) = self.model(feats, sr, language)
        k = self.model.lidk_engine.k
        texts = {l: self.tokenizer_dict[l].ids_to_text(*k.ctc_greedy(o.contiguous(), None, self.tokenizer_dict[l].blank_id))
                 for l, o in out.items()}
        return texts, lid_asr, out

    def infer(self, x, language: str = None, device=None):
        from lid.raw_datasets import read_audio
        wav, sr = read_audio(x, normalize=True)
        return self.infer_tensor(wav.to(self.trainer.device if device is None else device), sr, language)

    @staticmethod
    def score_to_prob(scores: List[float]) -> List[float]:
        """p_i = -1/(s_i - 1e-9), normalised (lid/LidModule_ASR_Supervised.py:316-318)."""
        prob = [(-1 / (s - 1e-9)) for s in scores]
        tot = sum(prob)
        return [p / tot for p in prob]

    def val_loop(self, batch):
        out = self.common_loop(batch, with_text=True)
        feats = self.model.features(batch[0])
        n_frames = (feats.shape[1] * batch[2]).long().tolist()
        targets = batch[5].tolist()
        # Per utterance, unpadded, as the reference scores them.  Utterances with the same frame count go through
        # the model together: in eval mode nothing couples the rows of a batch (BatchNorm uses running statistics, attention
        # is per utterance), so the scores equal the B=1 scores while equal-length sets cost one forward instead of B.
        groups: Dict[int, List[int]] = {}
        for i, nf in enumerate(n_frames):
            groups.setdefault(max(int(nf), 3), []).append(i)
        for nf, idx in groups.items():
            scores = self.lid_scores(feats[idx, :nf].contiguous()).tolist()
            for row, i in zip(scores, idx):
                prob = self.score_to_prob(row)
                self.eer.update([prob], [targets[i]])
                self.cavg.update([prob], [targets[i]])
        loss = float(out["loss"].detach())
        if not np.isnan(loss):
            self.avg_loss = 0.98 * self.avg_loss + 0.02 * loss
            self.avg_wer = 0.98 * self.avg_wer + 0.02 * out["wer"]
            self.count += 1
            corr = 1 - np.power(0.98, self.count)
            self.trainer.logger.log(data={"loss": self.avg_loss / corr, "val_wer": self.avg_wer / corr}, progress=True,
                                    only_tbar=True, stage="val")
        return {"val_loss": out["loss"].detach(), "val_wer": out["wer"], "predict_texts": out["predict_texts"],
                "label_texts": out["label_texts"]}

    def val_loop_end(self, outputs: List[Any] = None):
        preds, labels, losses = [], [], []
        for item in outputs or []:
            preds.extend(item["predict_texts"])
            labels.extend(item["label_texts"])
            if torch.isnan(item["val_loss"]).item():
                logging.warning("val loss is nan, ignored")
                continue
            losses.append(float(item["val_loss"]))
        total_wer = float(self.model.model.wer_fn(preds, labels)) if preds else 0.0
        total_eer, total_cavg = self.eer.compute(), self.cavg.compute()
        self.eer.reset()
        self.cavg.reset()
        data = {"val_loss": sum(losses) / max(len(outputs or []), 1), "val_wer": total_wer, "epoch": self.trainer.current_epoch,
                "eer": total_eer, "cavg": total_cavg}
        self.last_val = data
        self.trainer.logger.log(data=data, progress=True, stage="val", commit=False, only_tbar=False)
        logging.info("epoch %d: %s", self.trainer.current_epoch, data)
        self.trainer.logger.remove_key(["loss", "wer", "_runtime", "_timestamp"])

    def test_loop(self, batch):
        return self.val_loop(batch)

    def test_loop_end(self, outputs: List[Any] = None):
        return self.val_loop_end(outputs)
