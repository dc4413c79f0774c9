"""Tri-stage LR schedule (warm-up / hold / exponential decay), same constructor and LR sequence as the reference's
ccml/optim/tri_state.py:6-116, written statelessly on top of ``last_epoch`` so that save/resume needs no extra counter."""
import math
from typing import Optional, Tuple

from torch.optim.lr_scheduler import LRScheduler


class TriStageLRSchedule(LRScheduler):
    def __init__(self, optimizer, warmup_steps: int = 0, hold_steps: int = 0, decay_steps: int = 0,
                 phase_ratio: Optional[Tuple[float, float, float]] = None, init_lr_scale: float = 0.01,
                 final_lr_scale: float = 0.01, max_update: float = 1000, lr: float = 1e-4):
        self.peak_lr = lr
        self.init_lr = init_lr_scale * lr
        self.final_lr = final_lr_scale * lr
        if phase_ratio is not None:
            if not (max_update > 0 and abs(sum(phase_ratio) - 1.0) < 1e-9):
                raise ValueError("phase ratios must add up to 1 and max_update must be positive")
            warmup_steps, hold_steps, decay_steps = (int(max_update * r) for r in phase_ratio)
        if warmup_steps + hold_steps + decay_steps <= 0:
            raise ValueError("please specify steps or phase_ratio")
        self.warmup_steps, self.hold_steps, self.decay_steps = warmup_steps, hold_steps, decay_steps
        self.warmup_rate = (self.peak_lr - self.init_lr) / warmup_steps if warmup_steps else 0.0
        self.decay_factor = -math.log(final_lr_scale) / decay_steps if decay_steps else 0.0
        super().__init__(optimizer)

    def lr_at(self, k: int) -> float:
        if k < self.warmup_steps:
            return self.init_lr + self.warmup_rate * k
        k -= self.warmup_steps
        if k < self.hold_steps:
            return self.peak_lr
        k -= self.hold_steps
        if k <= self.decay_steps:
            return self.peak_lr * math.exp(-self.decay_factor * k)
        return self.final_lr

    def get_lr(self):
        return [self.lr_at(self.last_epoch) for _ in self.optimizer.param_groups]
