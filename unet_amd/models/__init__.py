"""Model registry, same keys and call convention as /root/reference/models/__init__.py:6-9."""
from .unet import build_unet, UNet, DoubleConv  # noqa: F401

REGISTRY = {
    "unet": build_unet,
}

try:  # RepUNet lands with the GroupNorm / fold kernels
    from .reparm_unet import build_reparam_unet, ReparamUNet  # noqa: F401
    REGISTRY["reparam_unet"] = build_reparam_unet
except ImportError:  # pragma: no cover
    pass
