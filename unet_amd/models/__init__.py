"""Model registry, same keys and call convention as /root/reference/models/__init__.py:6-9."""
from .unet import build_unet, UNet, DoubleConv  # noqa: F401
from .reparm_unet import build_reparam_unet, ReparamUNet, MultiConvBlock, AsymmConvBlock  # noqa: F401

REGISTRY = {
    "unet": build_unet,
    "reparam_unet": build_reparam_unet,
}
