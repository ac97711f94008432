"""desire_amd -- MI355X-native DESIRE sample-generation + ranking/refinement hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); all
arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of include/desire_hip.h
(libdesire_hip.so, loaded with ctypes).  There is no CPU fallback: without the library every
compute entry point raises.
"""
from .spec import Dims, init_weights, weight_shapes, flops_per_sample  # noqa: F401

__all__ = ["Dims", "init_weights", "weight_shapes", "flops_per_sample"]
