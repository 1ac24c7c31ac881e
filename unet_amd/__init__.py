"""unet_amd -- MI355X-native drop-in for the U-Net hot path of abin102/unet.

Mirrors the reference's Python boundary (models.REGISTRY, losses.get_loss, utils.train_step.step_batch,
trainer.Trainer) on top of libunet_hip.so (hand-written gfx950 HIP kernels, C ABI in include/unet_hip.h).
There is no CPU or eager fallback: ops raise if the library is missing or a tensor is not on the GPU.
"""
from ._lib import UNetHipError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
