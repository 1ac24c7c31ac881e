"""Loss registry, same keys as /root/reference/losses/__init__.py:13-24."""
from .segmentation import CEDiceLoss, CrossEntropyLoss, dice_loss, check_labels  # noqa: F401

_LOSSES = {}
REGISTRY = _LOSSES


def register(name):
    def decorator(fn):
        _LOSSES[name] = fn
        return fn
    return decorator


def get_loss(name, **kwargs):
    if name not in _LOSSES:
        raise ValueError(f"Unknown loss: {name}. Available: {list(_LOSSES.keys())}")
    return _LOSSES[name](**kwargs)


@register("ce")
def build_ce(**kwargs):
    return CrossEntropyLoss(**kwargs)


@register("ce_dice")
def build_ce_dice(**kwargs):
    return CEDiceLoss(**kwargs)
