import random

import numpy as np
import torch


def seed_everything(seed: int = 0):
    """Seeds python, numpy and torch (CPU + every GPU) — reference: ccml/train_helper.py:6-12.  The python stream
    also drives stochastic depth (lid/conformer.py:460-466), so identical seeds give identical layer drops on all ranks."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
