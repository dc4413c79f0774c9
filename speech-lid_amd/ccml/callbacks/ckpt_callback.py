"""Checkpoint callback: ``ckpt/last.pt`` every evaluated epoch plus the best-k by a metric
(reference: ccml/callbacks/ckpt_callback.py:17-169; same file names and state-dict keys)."""
import heapq
import logging
import os
import uuid
from typing import List

import torch

from ccml.train_callback import Callback


class CkptCallback(Callback):
    def __init__(self, interval: int = 1, ckpt_path: str = "ckpt", save_topk: int = 1,
                 file_name_metric: List = ("epoch", "avg_val_loss"), metric: str = "avg_val_loss", manager: str = "min",
                 *args, **kwargs):
        super().__init__(interval=interval)
        os.makedirs(ckpt_path, exist_ok=True)
        self.ckpt_path = os.path.abspath(ckpt_path)
        self.parttern = list(file_name_metric)
        self.save_topk = save_topk
        self.metric = metric
        self.manager = manager
        self._best: list = []          # heap of (-badness, path): the worst kept checkpoint on top

    # -- helpers --------------------------------------------------------------------------------
    @staticmethod
    def _mean_of(results, key):
        vals = []
        for item in results:
            if key not in item:
                return None
            v = item[key]
            if isinstance(v, torch.Tensor):
                if torch.isnan(v).any():
                    continue
                v = float(v)
            vals.append(v)
        return sum(vals) / len(vals) if vals else None

    def result_has_key(self, target: dict, key: str):
        if not isinstance(target, dict):
            return None
        if key in target:
            return target[key]
        res = target.get("all_val_results") or []
        return self._mean_of(res, key) if res else None

    def parse2abspath(self, value: dict, file_name_metric: List):
        parts = []
        for name in file_name_metric:
            v = self.result_has_key(value, name)
            if v is None:
                continue
            parts.append(f"{name}_{v}" if isinstance(v, int) else f"{name}_{float(v):.2f}")
        stem = "_".join(parts) if parts else "default_" + str(uuid.uuid4())[:4]
        return os.path.join(self.ckpt_path, stem + ".pt")

    def get_state(self) -> dict:
        t = self.trainer
        state = {"model": t.model.state_dict(), "hyper_parameters": t.ccml_module.get_hyper_parameters(),
                 "epoch": t.current_epoch, "optimizer": t.optimizer.state_dict(), "scalar": t.scalar.state_dict(),
                 "logger": t.logger.state_dict()}
        if t.lr_scheduler is not None:
            state["lr_scheduler"] = t.lr_scheduler.state_dict()
        return state

    # -- hook -----------------------------------------------------------------------------------
    def after_eval_epoch(self, value: dict):
        self.after_eval_epoch_count += 1
        if self.after_eval_epoch_count % self.interval != 0 or self.trainer.local_rank > 0:
            return
        state = self.get_state()
        if "swa" in value:
            torch.save(state, os.path.join(self.ckpt_path, "swa_final.pt"))
            return
        torch.save(state, os.path.join(self.ckpt_path, "last.pt"))
        metric = self.result_has_key(value, self.metric)
        if metric is None:
            return
        metric = float(metric)
        badness = metric if self.manager == "min" else -metric
        path = self.parse2abspath(value, self.parttern)
        if len(self._best) < self.save_topk:
            torch.save(state, path)
            heapq.heappush(self._best, (-badness, path))
        elif badness < -self._best[0][0]:
            _, old = heapq.heapreplace(self._best, (-badness, path))
            torch.save(state, path)
            if old != path and os.path.exists(old):
                os.remove(old)
        logging.info("checkpoint saved under %s", self.ckpt_path)
