"""Checkpoint files in the reference's on-disk format (written by /root/reference/utils/checkpoint_utils.py:15-48,
read by :50-89 and by /root/reference/utils/io.py:19-45), so that a `last.ckpt` / `best.ckpt` produced by either side
loads on the other.  Only the FORMAT is the contract; it is stated here as data:

  * FIELDS: what one checkpoint dict holds, each entry under a canonical key plus the older aliases readers accept;
  * a state_dict may carry the `module.` prefix of a DDP / DataParallel wrapper on either side: `match_prefix`
    rewrites the keys to whatever the receiving model uses;
  * files: `<dir>/last.ckpt` always, `<dir>/best.ckpt` when asked.

`torch.load` runs with weights_only=True by default (tensors, containers and plain scalars: everything written here);
pass weights_only=False only for a trusted checkpoint that pickled other objects.
"""
import logging
import os
from typing import Callable, Dict, Iterable, Mapping, NamedTuple, Optional, Tuple

import torch
import torch.nn as nn

log = logging.getLogger("unet_amd.checkpoint")

PREFIX = "module."


class Field(NamedTuple):
    key: str                         # canonical key in the file
    aliases: Tuple[str, ...]         # further keys under which readers look for it, in order
    dup: Optional[str]               # second key the writer stores the same object under (older readers)
    source: Callable                 # trainer -> object to store


def _state_of(attr):
    def get(trainer):
        obj = getattr(trainer, attr, None)
        return obj.state_dict() if obj else None
    return get


def unwrap(model):
    """The module inside a data-parallel wrapper (ours, DDP, nn.DataParallel), else the model itself."""
    from ..parallel import DataParallel
    wrappers = (DataParallel, nn.DataParallel, nn.parallel.DistributedDataParallel)
    return model.module if isinstance(model, wrappers) else model


def _model_state(trainer):
    model = getattr(trainer, "model", None)
    return None if model is None else unwrap(model).state_dict()


FIELDS: Dict[str, Field] = {
    "model": Field("model_state", ("model", "state_dict", "model_state_dict"), "model", _model_state),
    "optimizer": Field("optim_state", ("opt", "optimizer"), "opt", _state_of("optimizer")),
    "scheduler": Field("scheduler_state", ("sched", "scheduler"), "sched", _state_of("scheduler")),
    "scaler": Field("scaler_state", ("scaler",), None, _state_of("scaler")),
    "frozen": Field("trainer_backbone_frozen", (), None, lambda t: getattr(t, "_backbone_frozen", False)),
    "val_stats": Field("latest_val_stats", (), None, lambda t: getattr(t, "latest_val_stats", {})),
    "cfg": Field("cfg", (), None, lambda t: getattr(t, "cfg", None)),
}
EPOCH_KEYS = ("epoch", "epo")


def section(ckpt, name: str):
    """The object stored for FIELDS[name] under its canonical key or the first alias present, else None."""
    if not isinstance(ckpt, Mapping):
        return None
    f = FIELDS[name]
    for k in (f.key,) + f.aliases:
        if k in ckpt:
            return ckpt[k]
    return None


def match_prefix(state: Mapping[str, torch.Tensor], target_keys: Iterable[str]) -> Dict[str, torch.Tensor]:
    """Add or strip the leading `module.` so the keys line up with the receiving model's."""
    want = any(k.startswith(PREFIX) for k in target_keys)
    have = any(k.startswith(PREFIX) for k in state)
    if want == have:
        return dict(state)
    if have:
        return {(k[len(PREFIX):] if k.startswith(PREFIX) else k): v for k, v in state.items()}
    return {PREFIX + k: v for k, v in state.items()}


def _dir_of(path: str) -> str:
    return os.path.dirname(path) if os.path.splitext(path)[1] else path


def save_checkpoint(path: str, trainer, epoch: int, is_best: bool = False) -> None:
    """`path` names the run directory (or a file inside it).  Same signature as the reference's."""
    out_dir = _dir_of(path)
    os.makedirs(out_dir, exist_ok=True)
    ckpt = {EPOCH_KEYS[0]: int(epoch)}
    for f in FIELDS.values():
        ckpt[f.key] = f.source(trainer)
    for f in FIELDS.values():
        if f.dup:
            ckpt[f.dup] = ckpt[f.key]
    for name in ("last.ckpt",) + (("best.ckpt",) if is_best else ()):
        torch.save(ckpt, os.path.join(out_dir, name))


def read(path: str, map_location=None, weights_only: bool = True):
    if not os.path.exists(path):
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    if map_location is None:
        map_location = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cpu"
    return torch.load(path, map_location=map_location, weights_only=weights_only)


def load_checkpoint(path: str, model: torch.nn.Module, optimizer=None, scheduler=None, trainer=None,
                    map_location=None, strict=False, weights_only=True):
    """Restore model (+ optimizer / scheduler / trainer.scaler when given).  Returns {"start_epoch", "raw"} like the
    reference: training resumes at the epoch after the stored one, or at 1 when none is stored."""
    log.info("loading checkpoint %s", path)
    ckpt = read(path, map_location, weights_only)
    state = section(ckpt, "model")
    if state is not None:
        model.load_state_dict(match_prefix(state, model.state_dict().keys()), strict=strict)
    receivers = (("optimizer", optimizer), ("scheduler", scheduler),
                 ("scaler", getattr(trainer, "scaler", None) if trainer else None))
    for name, obj in receivers:
        stored = section(ckpt, name)
        if obj and stored:
            obj.load_state_dict(stored)
    epoch = next((ckpt[k] for k in EPOCH_KEYS if isinstance(ckpt, Mapping) and ckpt.get(k)), None)
    return {"start_epoch": 1 if epoch is None else int(epoch) + 1, "raw": ckpt}
