"""Local JSON-lines metrics logger.  The reference's backends (Comet, W&B, TensorBoard) are network services or absent
packages; this one needs nothing and keeps the ``BaseLogger`` contract, including resume state."""
import json
import os
import time

import torch

from ccml.loggers.base_logger import BaseLogger


class JsonlLogger(BaseLogger):
    def __init__(self, path: str = "metrics.jsonl", name: str = "run", *args, **kwargs):
        super().__init__()
        self.path, self.name, self.step = path, name, 0
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)

    def log(self, data=None, *args, **kwargs):
        if not data:
            return
        row = {"t": time.time(), "step": self.step}
        for k, v in data.items():
            row[k] = float(v) if isinstance(v, (int, float)) or (isinstance(v, torch.Tensor) and v.numel() == 1) else str(v)
        self.step += 1
        with open(self.path, "a") as f:
            f.write(json.dumps(row) + "\n")

    def get_resume_state(self):
        return "jsonl", {"step": self.step, "path": self.path}

    def resume_from(self, checkpoint: dict):
        st = (checkpoint or {}).get("jsonl")
        if st:
            self.step = st.get("step", 0)
