"""No-GPU checks of the drop-in boundary: liblidk.so loads, exports every symbol include/lidk.h declares, the ctypes
table covers exactly that set, and the product path refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
import lidk
from lidk import _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "lidk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(lidk_\w+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert len(names) >= 30
    handle = ctypes.CDLL(lidk.lib_path())
    for n in names:
        assert hasattr(handle, n), f"{n} declared in lidk.h but not exported by liblidk.so"
    assert sorted(_lib.SIGNATURES) == names


def test_version_call_without_gpu():
    assert lidk.lib().lidk_version() == 1


def test_product_ops_refuse_cpu_tensors():
    x = torch.randn(4, 64)
    with pytest.raises(lidk.LidkError):
        lidk.ops.layernorm_fwd(x, torch.ones(64), torch.zeros(64), y32=torch.empty_like(x))
    with pytest.raises(lidk.LidkError):
        lidk.ops.normalize_wav(torch.randn(2, 100))
