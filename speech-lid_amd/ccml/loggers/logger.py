"""Fan-out logger + tqdm postfix (reference: ccml/loggers/logger.py:9-116)."""
import logging
from typing import Any, Dict, List

import torch

from ccml.loggers.base_logger import BaseLogger


class Logger:
    def __init__(self, rank: int = -1, interval: int = 1):
        self.rank = rank
        self.loggers: List[BaseLogger] = []
        self.global_tqdm_elements: Dict[str, float] = {}
        self.interval = max(1, interval)
        self.global_step = 0
        self.trainer = None

    def add_logger(self, logger: BaseLogger):
        if isinstance(logger, BaseLogger):
            self.loggers.append(logger)
        else:
            logging.warning("ignored a logger that is not a BaseLogger: %r", logger)

    def attach_trainer(self, trainer):
        self.trainer = trainer

    def log(self, data: Dict[str, Any] = None, progress: bool = False, stage: str = "train", only_tbar: bool = False,
            *args, **kwargs):
        if data is None:
            return
        self.global_step += 1
        if stage == "train" and self.global_step % self.interval != 0:
            return
        if not only_tbar:
            for lg in self.loggers:
                lg.log(data, *args, **kwargs)
        if progress and self.rank <= 0:
            for k, v in data.items():
                if isinstance(v, torch.Tensor):
                    v = float(v.detach()) if v.numel() == 1 else 0.0
                self.global_tqdm_elements[k] = v if isinstance(v, (int, float)) else 0.0
            tbar = getattr(self.trainer, "tbar", None)
            if tbar is not None:
                tbar.set_postfix(self.global_tqdm_elements)

    def watch_model(self, model, *args, **kwargs):
        if self.rank == 0:
            for lg in self.loggers:
                lg.watch_model(model, *args, **kwargs)

    def get_checkpoint_by_name(self, name, path):
        for lg in self.loggers:
            p = lg.get_checkpoint_by_name(name, path)
            if p is not None:
                return p
        return None

    def state_dict(self):
        out = {}
        for lg in self.loggers:
            k, v = lg.get_resume_state()
            if k is not None:
                out[k] = v
        return out

    def load_state_dict(self, state_dict=None):
        for lg in self.loggers:
            lg.resume_from(state_dict)

    def remove_key(self, keys: List[str]):
        for k in keys:
            self.global_tqdm_elements.pop(k, None)
