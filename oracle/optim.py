"""Oracle: optimizer / LR schedule / gradient clipping restatement (test infrastructure)."""
import math
from typing import Dict, List, Optional

import torch


def clip_grad_norm(grads: List[torch.Tensor], max_norm: float = 20.0):
    """torch.nn.utils.clip_grad_norm_(params, max_norm=20) as called at ccml/trainer.py:541-543:
    total = ||[||g_i||_2]||_2 ; coef = min(1, max_norm / (total + 1e-6)); g *= coef.  Returns total."""
    total = torch.sqrt(sum((g.detach().float() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


class NovogradState:
    def __init__(self):
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None   # 0-dim
        self.step = 0


def novograd_step(params: List[torch.Tensor], grads: List[Optional[torch.Tensor]], states: List[NovogradState],
                  lr: float, betas=(0.95, 0.98), eps: float = 1e-8, weight_decay: float = 0.0,
                  grad_averaging: bool = False):
    """ccml/optim/novograd.py:75-145 (amsgrad=False, luc=False), per tensor:
    n = ||g||^2 ; v = n on first use (v == 0) else b2*v + (1-b2)*n ; g /= sqrt(v)+eps ;
    g += wd*p ; [g *= 1-b1] ; m = b1*m + g ; p -= lr*m.  Tensors with grad None are skipped."""
    b1, b2 = betas
    for p, g, st in zip(params, grads, states):
        if g is None:
            continue
        g = g.clone()
        if st.exp_avg is None:
            st.exp_avg = torch.zeros_like(p)
            st.exp_avg_sq = torch.zeros([])
        st.step += 1
        norm = g.norm().pow(2)
        if st.exp_avg_sq == 0:
            st.exp_avg_sq.copy_(norm)
        else:
            st.exp_avg_sq.mul_(b2).add_(norm, alpha=1.0 - b2)
        denom = st.exp_avg_sq.sqrt().add_(eps)
        g.div_(denom)
        if weight_decay != 0:
            g.add_(p, alpha=weight_decay)
        if grad_averaging:
            g.mul_(1 - b1)
        st.exp_avg.mul_(b1).add_(g)
        p.add_(st.exp_avg, alpha=-lr)


class TriStage:
    """ccml/optim/tri_state.py:6-116 with phase_ratio, as built at lid/LidModule_ASR_Supervised.py:142-149.

    ``lr_at(k)`` is the LR produced by the k-th call of ``get_lr`` (k = 0 is the call made inside
    ``_LRScheduler.__init__``; after n ``scheduler.step()`` calls the optimizer holds ``lr_at(n)``)."""

    def __init__(self, lr: float, max_update: float, phase_ratio=(0.1, 0.4, 0.5),
                 init_lr_scale: float = 0.05, final_lr_scale: float = 0.02):
        self.peak_lr = lr
        self.init_lr = init_lr_scale * lr
        self.final_lr = final_lr_scale * lr
        self.warmup_steps = int(max_update * phase_ratio[0])
        self.hold_steps = int(max_update * phase_ratio[1])
        self.decay_steps = int(max_update * phase_ratio[2])
        self.warmup_rate = (self.peak_lr - self.init_lr) / self.warmup_steps if self.warmup_steps != 0 else 0
        self.decay_factor = -math.log(final_lr_scale) / self.decay_steps

    def lr_at(self, k: int) -> float:
        if k < self.warmup_steps:
            return self.init_lr + self.warmup_rate * k
        off = self.warmup_steps
        if k < off + self.hold_steps:
            return self.peak_lr
        off += self.hold_steps
        if k <= off + self.decay_steps:
            return self.peak_lr * math.exp(-self.decay_factor * (k - off))
        return self.final_lr
