"""torch.optim.Adam / torch.optim.SGD (the reference's optimizers for the wav2vec2 / WavLM confs: lid/LidModule_ASR.py:143-150 with
lid/conf/xf_asr_wav2vec.yaml:25 ``optimizer_name: adam`` and xf_asr_extra_finetune.yaml:22 ``sgd``) with the update of every
parameter that holds a gradient done by ONE HIP launch (``lidk_adam_multi`` / ``lidk_sgd_multi``: a device table of 16 K-element
chunks, one read-modify-write pass) instead of torch's ~10 multi-tensor passes of ~25 launches each.

Subclasses of the torch optimizers: constructor, ``param_groups``, ``state`` (``step`` / ``exp_avg`` / ``exp_avg_sq`` /
``momentum_buffer``), ``state_dict`` and checkpoints are torch's own, and the update rule is that of torch's single-tensor
implementation.  Parameters that are not f32 CUDA tensors (CPU runs of the unit tests, exotic dtypes), ``amsgrad``, ``capturable``,
``differentiable`` and sparse gradients take torch's path unchanged."""
from typing import Dict, List

import numpy as np
import torch

__all__ = ["Adam", "SGD"]

_CHUNK_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i4"), ("pad", "<i4")])


def _lib():
    from lidk import _lib as L
    lib = L.lib()
    if lib.lidk_mt_chunk_bytes() != _CHUNK_DT.itemsize:
        raise RuntimeError("MtChunk layout mismatch between ccml/optim/multi_tensor.py and liblidk.so")
    return lib, lib.lidk_mt_chunk_elems()


def _eligible(p: torch.Tensor) -> bool:
    g = p.grad
    return (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g is not None and not g.is_sparse and
            g.dtype == torch.float32 and g.is_contiguous() and g.device == p.device)


class _Tables:
    """Chunk tables on the device, keyed by the addresses they hold (parameters, gradients and state tensors keep their addresses
    from step to step: the engine's and the backbone's gradients are views of flat arenas)."""

    def __init__(self):
        self._cache: Dict[tuple, tuple] = {}

    def get(self, triples: List[tuple]):
        """triples: (param, grad, m | None, v | None) per tensor -> (device table, number of chunks)."""
        key = tuple((p.data_ptr(), g.data_ptr(), 0 if m is None else m.data_ptr(), 0 if v is None else v.data_ptr(), p.numel())
                    for p, g, m, v in triples)
        hit = self._cache.get(key)
        if hit is None:
            _, chunk = _lib()
            parts = []
            for pp, gp, mp, vp, n in key:
                starts = np.arange(0, n, chunk, dtype=np.uint64)
                rec = np.zeros(len(starts), dtype=_CHUNK_DT)
                rec["p"], rec["g"] = pp + 4 * starts, gp + 4 * starts
                rec["m"] = (mp + 4 * starts) if mp else 0
                rec["v"] = (vp + 4 * starts) if vp else 0
                rec["n"] = np.minimum(chunk, n - starts.astype(np.int64)).astype(np.int32)
                parts.append(rec)
            table = np.concatenate(parts)
            dev = torch.from_numpy(table.view(np.uint8).copy()).to(triples[0][0].device)
            if len(self._cache) > 16:                       # regime changes (un-freezing) are rare; do not grow without bound
                self._cache.clear()
            hit = (dev, len(table))
            self._cache[key] = hit
        return hit


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Adam(torch.optim.Adam):
    def __init__(self, params, *args, **kwargs):
        super().__init__(params, *args, **kwargs)
        self._tables = _Tables()

    @torch.no_grad()
    def step(self, closure=None):
        plain = all(not g.get("amsgrad") and not g.get("capturable") and not g.get("differentiable") and not g.get("fused")
                    for g in self.param_groups)
        with_grad = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not plain or not with_grad or not all(_eligible(p) for p in with_grad):
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib, _ = _lib()
        for group in self.param_groups:
            by_step: Dict[int, list] = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                by_step.setdefault(int(st["step"]), []).append((p, p.grad, st["exp_avg"], st["exp_avg_sq"]))
            b1, b2 = group["betas"]
            lr = float(group["lr"])
            for t, triples in by_step.items():
                table, n = self._tables.get(triples)
                rc = lib.lidk_adam_multi(table.data_ptr(), n, lr, b1, b2, group["eps"], group["weight_decay"], 1.0 - b1 ** t,
                                         1.0 - b2 ** t, int(bool(group.get("maximize"))), _stream())
                if rc != 0:
                    raise RuntimeError(f"lidk_adam_multi failed with code {rc}")
        return loss


class SGD(torch.optim.SGD):
    def __init__(self, params, *args, **kwargs):
        super().__init__(params, *args, **kwargs)
        self._tables = _Tables()

    @torch.no_grad()
    def step(self, closure=None):
        plain = all(not g.get("differentiable") and not g.get("fused") for g in self.param_groups)
        with_grad = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not plain or not with_grad or not all(_eligible(p) for p in with_grad):
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib, _ = _lib()
        for group in self.param_groups:
            mom = float(group["momentum"])
            first, later = [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if mom == 0.0:
                    later.append((p, p.grad, None, None))
                    continue
                st = self.state[p]
                if st.get("momentum_buffer") is None:          # torch: buf = clone(grad) on a parameter's first step
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    first.append((p, p.grad, st["momentum_buffer"], None))
                else:
                    later.append((p, p.grad, st["momentum_buffer"], None))
            for is_first, triples in ((1, first), (0, later)):
                if triples:
                    table, n = self._tables.get(triples)
                    rc = lib.lidk_sgd_multi(table.data_ptr(), n, float(group["lr"]), mom, float(group["dampening"]),
                                            float(group["weight_decay"]), int(bool(group["nesterov"])), is_first,
                                            int(bool(group.get("maximize"))), _stream())
                    if rc != 0:
                        raise RuntimeError(f"lidk_sgd_multi failed with code {rc}")
        return loss
