"""Novograd (layer-wise second moments; arXiv:1905.11286) with the reference's constructor
(ccml/optim/novograd.py:30-68) and update rule (:75-145).

Two execution paths, chosen by what the parameters are:
  * parameters that are views of a lidk ``Engine`` arena (the Conformer-LID model): ONE fused HIP call per step over the
    flat parameter / gradient / moment arenas, with gradient-norm clipping folded in (``step(max_norm=...)``).  Tensors
    whose ``.grad`` is None are left out of the launch, exactly like the reference's ``if p.grad is None: continue``.
    This path raises if liblidk.so is unavailable; it never degrades to the generic loop.
  * any other parameters (arbitrary user models): a plain per-tensor torch implementation of the same rule.
"""
from typing import Dict, List, Optional

import torch
from torch.optim.optimizer import Optimizer

__all__ = ["Novograd"]


class Novograd(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.95, 0.98), eps=1e-8, weight_decay=0, grad_averaging=False,
                 amsgrad=False, luc=False, luc_trust=1e-3, luc_eps=1e-8):
        if lr < 0 or eps < 0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid Novograd hyper-parameters lr={lr} eps={eps} betas={betas}")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_averaging=grad_averaging, amsgrad=amsgrad)
        self.luc, self.luc_trust, self.luc_eps = luc, luc_trust, luc_eps
        super().__init__(params, defaults)
        self._engine = None
        engines = {getattr(p, "_lidk_engine", None) for g in self.param_groups for p in g["params"]}
        if len(engines) == 1 and None not in engines:
            if amsgrad or luc:
                raise NotImplementedError("fused Novograd: amsgrad / luc variants are not built (unused by the lid configs)")
            if len(self.param_groups) != 1:
                raise NotImplementedError("fused Novograd expects the model's parameters in a single group")
            self._engine = engines.pop()
            self._exp_avg = None
            self._exp_avg_sq = None
            self._work_cache: Dict[tuple, torch.Tensor] = {}
            self._head_cache: Dict[tuple, Optional[str]] = {}
            self._scratch = None
            self.total_norm = None          # device scalar: gradient norm before clipping, from the last step
        elif None not in engines:
            raise ValueError("parameters of several lidk engines in one optimizer")
        self.fused_clip = self._engine is not None
        self.zeroes_grads = self._engine is not None       # the fused launch leaves the gradients it consumed at zero

    # ------------------------------------------------------------------ fused path
    def _fused_step(self, max_norm: float):
        from lidk._lib import OPT_CHUNK
        eng, grp = self._engine, self.param_groups[0]
        dev = eng.flat.device
        if self._exp_avg is None:
            self._exp_avg = torch.zeros_like(eng.flat)
            self._exp_avg_sq = torch.zeros(len(eng.specs), device=dev)
            self.total_norm = torch.zeros(1, device=dev)
        tids = tuple(p._lidk_tid for p in grp["params"] if p.grad is not None)
        if not tids:
            return
        work = self._work_cache.get(tids)
        if work is None:
            rows = []
            for t in sorted(tids):
                s = eng.specs[t]
                for a in range(0, s.numel, OPT_CHUNK):
                    rows.append((t, s.offset + a, min(OPT_CHUNK, s.numel - a)))
            work = torch.tensor(rows, dtype=torch.int64, device=dev)
            if len(self._work_cache) > 64:
                self._work_cache.clear()
            self._work_cache[tids] = work
        need = work.shape[0] + len(eng.specs) + 8
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, device=dev)
        eng.k.novograd_step(eng.flat, eng.grad, self._exp_avg, self._exp_avg_sq, work, len(eng.specs), float(grp["lr"]),
                            grp["betas"], grp["eps"], grp["weight_decay"], grp["grad_averaging"],
                            float(max_norm) if max_norm else 0.0, self._scratch, self.total_norm)
        lang = self._head_cache.get(tids)
        if tids not in self._head_cache:
            lang = self._head_cache[tids] = eng.single_active_head(tids)
        eng.refresh_weights(lang)

    # ------------------------------------------------------------------ generic path
    @torch.no_grad()
    def _generic_step(self):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros([], device=p.device)
                    if group["amsgrad"]:
                        st["max_exp_avg_sq"] = torch.zeros([], device=p.device)
                st["step"] += 1
                n = g.norm().pow(2)
                v = st["exp_avg_sq"]
                if v == 0:
                    v.copy_(n)
                else:
                    v.mul_(b2).add_(n, alpha=1.0 - b2)
                if group["amsgrad"]:
                    torch.max(st["max_exp_avg_sq"], v, out=st["max_exp_avg_sq"])
                    denom = st["max_exp_avg_sq"].sqrt() + group["eps"]
                else:
                    denom = v.sqrt() + group["eps"]
                g = g / denom
                if group["weight_decay"] != 0:
                    g = g.add(p, alpha=group["weight_decay"])
                if group["grad_averaging"]:
                    g = g * (1 - b1)
                st["exp_avg"].mul_(b1).add_(g)
                if self.luc:
                    factor = min(float(self.luc_trust * p.norm() / (st["exp_avg"].norm() + self.luc_eps)), group["lr"])
                    p.add_(st["exp_avg"], alpha=-factor)
                else:
                    p.add_(st["exp_avg"], alpha=-group["lr"])

    def step(self, closure=None, max_norm: Optional[float] = None):
        loss = closure() if closure is not None else None
        if self._engine is not None:
            self._fused_step(max_norm or 0.0)
        else:
            if max_norm:
                torch.nn.utils.clip_grad_norm_([p for g in self.param_groups for p in g["params"]], max_norm)
            self._generic_step()
        return loss

    # ------------------------------------------------------------------ checkpoint format (torch layout on both paths)
    def state_dict(self):
        if self._engine is None or self._exp_avg is None:
            return super().state_dict()
        eng, state = self._engine, {}
        for i, p in enumerate(self.param_groups[0]["params"]):
            s = eng.specs[p._lidk_tid]
            if float(self._exp_avg_sq[s.tid]) != 0.0:
                state[i] = {"step": 0, "exp_avg": self._exp_avg[s.offset:s.offset + s.numel].view(s.shape).clone(),
                            "exp_avg_sq": self._exp_avg_sq[s.tid].clone()}
        groups = [{k: v for k, v in g.items() if k != "params"} | {"params": list(range(len(g["params"])))}
                  for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        if self._engine is None:
            return super().load_state_dict(state_dict)
        eng = self._engine
        for g, sg in zip(self.param_groups, state_dict["param_groups"]):
            g.update({k: v for k, v in sg.items() if k != "params"})
        self._exp_avg = torch.zeros_like(eng.flat)
        self._exp_avg_sq = torch.zeros(len(eng.specs), device=eng.flat.device)
        self.total_norm = torch.zeros(1, device=eng.flat.device)
        params = self.param_groups[0]["params"]
        for i, st in state_dict["state"].items():
            s = eng.specs[params[int(i)]._lidk_tid]
            self._exp_avg[s.offset:s.offset + s.numel] = st["exp_avg"].reshape(-1).to(eng.flat.device)
            self._exp_avg_sq[s.tid] = st["exp_avg_sq"].to(eng.flat.device)
