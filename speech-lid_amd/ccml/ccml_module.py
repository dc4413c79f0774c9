"""CCMLModule — the plug-in interface between a task and ``ccml.Trainer`` (reference: ccml/ccml_module.py:12-171)."""
from typing import Any, Dict, List, Optional, Tuple

import torch


class CCMLModule:
    """Subclass, build ``self.model`` in ``__init__`` (pass the constructor kwargs to ``super().__init__`` so they are
    stored as ``hyper_parameters`` and saved in checkpoints), and implement ``config_optim``, ``train_loop`` (must return a
    dict with a ``"loss"`` tensor) and ``val_loop`` (dict with ``"val_loss"``)."""

    def __init__(self, *args, **kwargs):
        self.hyper_parameters: Dict[str, Any] = dict(kwargs)
        self.model: Optional[torch.nn.Module] = None
        self.trainer = None
        self.dataloader_param: Dict[str, Any] = {}
        self._train_dataset = self._val_dataset = self._test_dataset = None

    # -- wiring ---------------------------------------------------------------------------------
    def get_model(self) -> torch.nn.Module:
        return self.model

    def point_trainer(self, trainer=None):
        self.trainer = trainer

    def save_hyper_parameters(self, hyper_parameters: dict):
        self.hyper_parameters = hyper_parameters

    def get_hyper_parameters(self) -> dict:
        return self.hyper_parameters

    @property
    def train_dataset(self):
        return self._train_dataset

    @property
    def val_dataset(self):
        return self._val_dataset

    @property
    def test_dataset(self):
        return self._test_dataset

    def config_datasource(self):
        pass

    # -- hooks ----------------------------------------------------------------------------------
    def config_optim(self, *args, **kwargs) -> Tuple[torch.optim.Optimizer, Any, Optional[dict]]:
        """-> (optimizer, scheduler | None, {"monitor": key | None, "interval": "step" | "epoch"} | None)"""
        opt = torch.optim.SGD(self.model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.1, patience=10, cooldown=3, min_lr=1e-5)
        return opt, sched, {"monitor": "val_loss", "interval": "epoch"}

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def train_loop(self, batch) -> dict:
        raise NotImplementedError

    def val_loop(self, batch) -> dict:
        raise NotImplementedError

    def test_loop(self, batch) -> dict:
        raise NotImplementedError

    def before_train_loop(self, value):
        pass

    def train_loop_end(self, outputs: List[Any] = None):
        pass

    def val_loop_end(self, outputs: List[Any] = None):
        pass

    def test_loop_end(self, outputs: List[Any] = None):
        pass

    # -- inference-time restore ------------------------------------------------------------------
    @classmethod
    def resume_from_checkpoint(cls, checkpont: str, map_location: str = "cpu", **overrides):
        """Rebuild the module from the hyper-parameters stored in a checkpoint and load its weights."""
        device = torch.device(map_location)
        state = torch.load(checkpont, map_location="cpu", weights_only=False)
        hp = dict(state.get("hyper_parameters", {}))
        hp.update({k: v for k, v in overrides.items() if k in hp})
        module = cls(**hp)
        model = module.get_model()
        sd = {k[7:] if k.startswith("module.") else k: v for k, v in state["model"].items()}
        model.load_state_dict(sd)
        model.to(device)
        return module
