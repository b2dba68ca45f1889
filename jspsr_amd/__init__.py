"""jspsr_amd -- MI355X (gfx950) native implementation of the JSPSR forward/backward hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed).  Compute:
hand-written HIP kernels in ``csrc/`` behind the C ABI of ``include/jspsr_hip.h``
(``lib/libjspsr_hip.so``).  There is no CPU or eager fallback: every op raises if the
library is missing or a tensor is not on the GPU.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
