"""lidk — ctypes binding of liblidk.so, the HIP/gfx950 kernels behind the LID training hot path.

There is NO CPU fallback: importing works anywhere (so host-side logic can be tested), but every op
raises ``LidkError`` unless liblidk.so is built (``make -C speech-lid_amd/csrc``) and the tensors live on a GPU.
"""
from ._lib import LidkError, lib, lib_path, BF16, F32, dtype_code  # noqa: F401
from . import ops  # noqa: F401
