"""EER2 / CAvg metric accumulators with the reference's update/compute/reset surface (lid/eer.py:39-123), without
torchmetrics.  Under data parallelism ``compute`` first gathers every rank's records (the reference's
``dist_reduce_fx="cat"``)."""
from typing import List

import numpy as np
import torch.distributed as dist

from lid.cavg import get_cavg


def _gather(items: list) -> list:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, items)
        return [x for part in out for x in part]
    return items


class EER2:
    def __init__(self, dist_sync_on_step=False, num_class=3):
        self.num_class = num_class
        self.reset()

    def reset(self):
        self.pos_list, self.score_list = [], []

    def update(self, predict: List[List[float]], target: List[int]) -> None:
        for p, t in zip(predict, target):
            for j, s in enumerate(p):
                self.score_list.append(float(s))
                self.pos_list.append(int(j == t))

    def compute(self) -> float:
        """Equal error rate: the point of the ROC curve where FPR = 1 - TPR (linear interpolation)."""
        y = np.asarray(_gather(self.pos_list), dtype=bool)
        s = np.asarray(_gather(self.score_list), dtype=np.float64)
        if y.size == 0 or y.all() or (~y).all():
            return 0.0
        order = np.argsort(-s, kind="mergesort")
        y, s = y[order], s[order]
        distinct = np.r_[np.nonzero(np.diff(s))[0], y.size - 1]
        tps = np.cumsum(y)[distinct]
        fps = 1 + distinct - tps
        tpr = np.r_[0.0, tps / tps[-1]]
        fpr = np.r_[0.0, fps / fps[-1]]
        f = 1.0 - tpr - fpr                      # root of 1 - x - tpr(x)
        k = int(np.argmax(f <= 0))
        if k == 0:
            return float(fpr[0])
        x0, x1, f0, f1 = fpr[k - 1], fpr[k], f[k - 1], f[k]
        return float(x0 + (x1 - x0) * f0 / (f0 - f1)) if f0 != f1 else float(x0)


class CAvg:
    def __init__(self, dist_sync_on_step=False, num_class=3):
        self.num_class = num_class
        self.reset()

    def reset(self):
        self.pairs = []

    def update(self, predict: List[List[float]], target: List[int]) -> None:
        for p, t in zip(predict, target):
            for j, s in enumerate(p):
                self.pairs.append((j, int(t), float(s)))

    def get_cavg(self, pairs, lang_num, min_score, max_score, bins=20, p_target=0.5):
        return get_cavg(pairs, lang_num, min_score, max_score, bins, p_target)

    def compute(self) -> float:
        pairs = _gather(self.pairs)
        if not pairs:
            return 0.0
        lo, hi = min(p[2] for p in pairs), max(p[2] for p in pairs)
        _, mn = self.get_cavg(pairs, self.num_class, lo, hi, 20, 0.5)
        return round(mn, 4)
