import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "speech-lid_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def cfg1_weights():
    return {k: torch.from_numpy(v) for k, v in load_npz("cfg1_weights.npz").items()}


@pytest.fixture(scope="session")
def cfg1_cfg():
    from oracle.conformer import ModelCfg
    return ModelCfg(lang2vocab={"a": 30, "b": 40, "c": 50}, lang2index={"a": 0, "b": 1, "c": 2}, n_blocks=2,
                    encoder_dim=64, dim_head=16, heads=4, last_dim_head=8, dropout=0.1, hidden_dim=32)
