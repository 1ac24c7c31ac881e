"""Builders with the reference's names and behaviour (/root/reference/utils/builders.py:15-260),
loguru-free: build_model_from_cfg, build_loss, get_optim, get_scheduler."""
import logging

import torch

logger = logging.getLogger("unet_amd")


def build_model_from_cfg(cfg, device=None):
    from ..models import REGISTRY
    name = cfg["model"]
    if name not in REGISTRY:
        raise ValueError(f"Unknown model '{name}'. Available: {list(REGISTRY.keys())}")
    model = REGISTRY[name](**(cfg.get("model_args") or {}))
    return model.to(device) if device is not None else model


def build_loss(loss_name, loss_args, device):
    from ..losses import CEDiceLoss, CrossEntropyLoss
    if loss_name is None:
        raise ValueError("loss_name must be specified in config")
    lname = loss_name.lower()
    if lname == "ce":
        loss = CrossEntropyLoss(**loss_args)
    elif lname == "ce_dice":
        loss = CEDiceLoss(**loss_args)
    else:
        raise ValueError(f"Unknown loss: {loss_name}")
    return loss.to(device)


def get_optim(name, params, **kw):
    name = name.lower()
    if name == "adamw":
        from ..optim import FusedAdamW          # a torch.optim.AdamW whose step() is the fused HIP kernel
        return FusedAdamW(params, **kw)
    if name == "adam":
        return torch.optim.Adam(params, **kw)
    if name == "sgd":
        momentum = kw.pop("momentum", 0.9)
        nesterov = kw.pop("nesterov", True)
        return torch.optim.SGD(params, momentum=momentum, nesterov=nesterov, **kw)
    raise ValueError(f"Unknown optim: {name}")


def get_scheduler(name, optim, **kw):
    if name is None:
        return None
    lname = name.lower()
    if lname in ("none", ""):
        return None
    if lname == "cosine":
        return torch.optim.lr_scheduler.CosineAnnealingLR(optim, **kw)
    if lname in ("multistep", "step"):
        warmup_epochs = kw.pop("warmup_epochs", None)
        main = torch.optim.lr_scheduler.MultiStepLR(optim, **kw)
        if warmup_epochs is None or warmup_epochs <= 0:
            return main

        def warm(epoch_idx, w=warmup_epochs):
            factor = float(epoch_idx + 1) / float(w)
            return (1.0 / 3.0) * (1.0 - factor) + factor
        warmup = torch.optim.lr_scheduler.LambdaLR(optim, lr_lambda=warm)
        return torch.optim.lr_scheduler.SequentialLR(optim, schedulers=[warmup, main], milestones=[warmup_epochs])
    raise ValueError(f"Unknown scheduler: {name}")
