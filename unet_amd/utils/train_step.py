"""step_batch with the reference's signature, return value and side effects
(/root/reference/utils/train_step.py:7-79):

    zero_grad -> (autocast) forward -> loss -> scaled backward -> unscale + clip -> optimizer step
    -> trainer._global_step += 1 -> info{'main_loss','lr'} -> on_batch_end callbacks

Differences: autocast selects the bf16 HIP kernels (the reference's fp16 autocast has no bf16 path);
with `unet_amd.optim.FusedAdamW` (what `get_optim("adamw")` builds) and a disabled GradScaler the
clip + optimizer tail is the fused multi-tensor path;
the one host sync that fetches `main_loss` also carries the loss kernel's illegal-label counter, so a
label the reference would have raised on still raises here -- at the end of the step instead of inside loss_fn; with the
fused optimizer tail the same counter vetoes the update on the device, so the bad batch changes neither weights nor
optimizer moments nor Adam's step count (as in the reference, where the raise comes before backward); under
unet_amd.parallel.DataParallel the counter is first MAX-reduced over the ranks, so all of them veto and raise together.
"""
import os
from contextlib import nullcontext
from typing import Any, Dict, Tuple

import torch


_PINNED = {}
_LATE = os.environ.get("UNET_AMD_LATE_STATUS") == "1"      # A/B: read the status at the end of the step (drains the stream)


def _queue_status_read(status):
    """-> (pinned host tensor, event) for an fp32 CUDA status vector, or None.  Two pinned buffers per device, alternating, so
    a caller that has not read step k's values yet does not see them overwritten by step k+1."""
    if status is None or not status.is_cuda or status.dtype != torch.float32 or status.numel() > 16 or _LATE:
        return None
    slot = _PINNED.setdefault(status.device, [torch.empty(16, dtype=torch.float32).pin_memory(),
                                              torch.empty(16, dtype=torch.float32).pin_memory(), 0])
    slot[2] ^= 1
    host = slot[slot[2]][:status.numel()]
    host.copy_(status.detach(), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(status.device))
    return host, ev


def step_batch(trainer, x: torch.Tensor, y: torch.Tensor, scaler, batch_idx: int = 0) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
    trainer.optimizer.zero_grad(set_to_none=True)

    amp_on = getattr(trainer, "amp", False)
    amp_ctx = torch.amp.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=True) if amp_on else nullcontext()
    info: Dict[str, Any] = {}

    with amp_ctx:
        try:
            outputs = trainer.model(x)
        except TypeError:
            outputs = trainer.model(x, current_epoch=getattr(trainer, "current_epoch", -1))
        if isinstance(outputs, dict):
            logits = outputs.get("logits", outputs.get("out", None))
        elif isinstance(outputs, tuple):
            logits = outputs[0]
        else:
            logits = outputs
        if batch_idx == 0:
            trainer.logger.debug(f"Step | Logits Shape: {logits.shape}")
        if y.ndim == 4:
            y = y.squeeze(1)
            if batch_idx == 0:
                trainer.logger.debug(f"Step | Y Squeezed: {y.shape}")
        main_loss = trainer.loss_fn(logits, y)

    # data parallel: the illegal-label counter becomes the MAX over ranks, so that the veto below and the error at the end of
    # the step are the same on every rank
    status = getattr(main_loss, "_unet_status", None)
    share = getattr(trainer.model, "share_status", None)
    if share is not None and status is not None and status.numel() >= 4:
        share(status[3:4])
    # The step's one host read (loss value + illegal-label counter) is queued NOW, right behind the loss kernels, as an
    # asynchronous copy into pinned memory with an event behind it.  At the end of the step the host waits for that event only
    # -- reached long before backward and the optimizer have run -- instead of draining the stream, so it is already queueing
    # the next step while this one finishes and the GPU never sits idle between steps (0.2-0.35 ms per step of Python start-up
    # otherwise: zero_grad, module call, first launches).
    early = _queue_status_read(status)

    scaler.scale(main_loss).backward()
    fused = getattr(trainer.optimizer, "clip_and_step", None)
    use_fused = fused is not None and not scaler.is_enabled() and trainer.grad_clip > 0
    sync = getattr(trainer.model, "finish_grad_sync", None)
    grad_scale = 1.0
    if sync is not None:          # data-parallel wrapper: wait for the bucketed all-reduces
        # with the fused tail the division by the world size rides on the clip coefficient (no pass of its own)
        grad_scale = sync(average=not use_fused) or 1.0

    if use_fused:
        # unet_amd.optim.FusedAdamW: norm -> clip -> AdamW in two multi-tensor passes, no host sync.  The loss kernel's
        # illegal-label counter rides along as a veto: a batch the reference would have raised on inside loss_fn (before
        # backward) leaves weights and optimizer state untouched here as well; the error itself is raised below.
        status = getattr(main_loss, "_unet_status", None)
        kw = {}
        if grad_scale != 1.0:
            kw["grad_scale"] = grad_scale
        if status is not None and status.dtype == torch.float32 and status.numel() >= 4:
            kw["veto"] = status[3:4]
        fused(trainer.model.parameters(), trainer.grad_clip, **kw)
    else:
        if trainer.grad_clip > 0:
            scaler.unscale_(trainer.optimizer)
            torch.nn.utils.clip_grad_norm_(trainer.model.parameters(), trainer.grad_clip)
        scaler.step(trainer.optimizer)
    scaler.update()
    inner = getattr(trainer.model, "module", trainer.model)
    prepack = getattr(inner, "prepack_weights", None)
    if prepack is not None:        # re-pack the updated weights now, behind the optimizer kernels, not before the next forward
        prepack()

    step_idx = int(getattr(trainer, "_global_step", 0)) + 1
    setattr(trainer, "_global_step", step_idx)

    if status is not None:
        if early is not None:
            early[1].synchronize()                  # the step's host sync: the copy queued right behind the loss kernels
            vals = early[0].tolist()
        else:
            vals = status.tolist()
        if vals[3] > 0:
            # the update kernel saw the same flag and did nothing: take back the host-side Adam step count as well, so a
            # caller that catches this and skips the batch continues with the bias correction of the updates that ran
            undo = getattr(trainer.optimizer, "undo_vetoed_step", None)
            if use_fused and undo is not None:
                undo()
            raise RuntimeError("Class values must be smaller than num_classes.")
        info["main_loss"] = vals[0]
    else:
        info["main_loss"] = main_loss.item()
    info["lr"] = float(trainer.optimizer.param_groups[0]["lr"])

    for cb in trainer.cbs:
        if hasattr(cb, "on_batch_end"):
            try:
                cb.on_batch_end(trainer, step_idx, info)
            except Exception:
                pass
    return main_loss, logits, info
