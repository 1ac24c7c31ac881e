"""Checkpoint save / load with the reference's file format and signatures
(/root/reference/utils/checkpoint_utils.py:15-89), loguru-free.

A checkpoint written here loads in the reference and vice versa: keys `epoch`, `model_state` (+ compat
`model`), `optim_state` (`opt`), `scheduler_state` (`sched`), `scaler_state`, `trainer_backbone_frozen`,
`latest_val_stats`, `cfg`; files `last.ckpt` / `best.ckpt`.  The module state_dict is the reference's
(identical keys and shapes), unwrapped from unet_amd.parallel.DataParallel / torch DDP / nn.DataParallel;
the `module.` prefix fix-up on load is the reference's (:66-79).  FusedAdamW's state_dict is torch AdamW's.

`torch.load` runs with weights_only=True by default (tensors, containers and plain scalars: everything
save_checkpoint writes); pass weights_only=False only for a checkpoint you trust that pickled other objects.
"""
import logging
import os

import torch
import torch.nn as nn

log = logging.getLogger("unet_amd.checkpoint")


def _pick(*dicts, keys=None):
    for d in dicts:
        if not isinstance(d, dict):
            continue
        for k in keys or []:
            if k in d:
                return d[k], k
    return None, None


def _unwrap(model):
    from ..parallel import DataParallel
    if isinstance(model, (DataParallel, nn.DataParallel, nn.parallel.DistributedDataParallel)):
        return model.module
    return model


def save_checkpoint(path: str, trainer, epoch: int, is_best: bool = False) -> None:
    out_dir = os.path.dirname(path) if os.path.splitext(path)[1] else path
    os.makedirs(out_dir, exist_ok=True)
    model_ref = getattr(trainer, "model", None)
    model_state = _unwrap(model_ref).state_dict() if model_ref is not None else None

    def sd(name):
        obj = getattr(trainer, name, None)
        return obj.state_dict() if obj else None

    ckpt = {
        "epoch": int(epoch),
        "model_state": model_state,
        "optim_state": sd("optimizer"),
        "scheduler_state": sd("scheduler"),
        "scaler_state": sd("scaler"),
        "trainer_backbone_frozen": getattr(trainer, "_backbone_frozen", False),
        "latest_val_stats": getattr(trainer, "latest_val_stats", {}),
        "cfg": getattr(trainer, "cfg", None),
    }
    ckpt["model"] = ckpt["model_state"]
    ckpt["opt"] = ckpt["optim_state"]
    ckpt["sched"] = ckpt["scheduler_state"]
    torch.save(ckpt, os.path.join(out_dir, "last.ckpt"))
    if is_best:
        torch.save(ckpt, os.path.join(out_dir, "best.ckpt"))


def load_checkpoint(path: str, model: torch.nn.Module, optimizer=None, scheduler=None, trainer=None,
                    map_location=None, strict=False, weights_only=True):
    log.info("Attempting to load checkpoint from: %s", path)
    if not os.path.exists(path):
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    if map_location is None:
        map_location = "cpu" if not torch.cuda.is_available() else f"cuda:{torch.cuda.current_device()}"
    ckpt = torch.load(path, map_location=map_location, weights_only=weights_only)

    model_state, _ = _pick(ckpt, keys=["model_state", "model", "state_dict"])
    optim_state, _ = _pick(ckpt, keys=["optim_state", "opt", "optimizer"])
    sched_state, _ = _pick(ckpt, keys=["scheduler_state", "sched", "scheduler"])
    scaler_state, _ = _pick(ckpt, keys=["scaler_state", "scaler"])

    if model_state is not None:
        curr_keys = set(model.state_dict().keys())
        loaded_keys = list(model_state.keys())
        new_state = model_state
        if any(k.startswith("module.") for k in loaded_keys) and not any(k.startswith("module.") for k in curr_keys):
            new_state = {k.replace("module.", ""): v for k, v in model_state.items()}
        elif any(k.startswith("module.") for k in curr_keys) and not any(k.startswith("module.") for k in loaded_keys):
            new_state = {"module." + k: v for k, v in model_state.items()}
        model.load_state_dict(new_state, strict=strict)

    if optimizer and optim_state:
        optimizer.load_state_dict(optim_state)
    if scheduler and sched_state:
        scheduler.load_state_dict(sched_state)
    if trainer and scaler_state and hasattr(trainer, "scaler"):
        trainer.scaler.load_state_dict(scaler_state)

    epoch = ckpt.get("epoch") or ckpt.get("epo")
    start_epoch = int(epoch) + 1 if epoch is not None else 1
    return {"start_epoch": start_epoch, "raw": ckpt}
