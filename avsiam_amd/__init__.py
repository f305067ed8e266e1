"""avsiam_amd - MI355X-native implementation of the AVSiam (GenjiB/AVSiam) pre-training hot path.

Python host code on PyTorch-ROCm (device memory, streams, torch.distributed) calling hand-written gfx950
HIP kernels through the C ABI of libavsiam_hip.so (include/avsiam_hip.h).  See DESIGN.md.
"""
from .config import AVSiamConfig  # noqa: F401

__all__ = ["AVSiamConfig", "models"]
