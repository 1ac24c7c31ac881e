"""Trainer with the reference's constructor and fit/evaluate API (/root/reference/trainer.py:12-173).

Host loop only; the device work is step_batch -> HIP kernels.  Kept quirks: epoch train loss divides by
len(train_loader.dataset) (trainer.py:133), callbacks' exceptions are swallowed where the reference
swallows them.  GradScaler follows `amp` as in the reference (trainer.py:60); with bf16 compute it has
nothing to rescue, so it is created disabled.
"""
import logging
from typing import Iterable, Optional

import torch
from torch import nn

from . import ops
from .utils.eval_utils import evaluate
from .utils.train_step import step_batch


class Trainer:
    def __init__(self, model: nn.Module, loss_fn, optimizer, scheduler=None, amp=False, grad_clip=0.0,
                 device="cuda", callbacks: Iterable = (), progress_bar=True, debug=False, debug_dir="debug",
                 tb_logdir: Optional[str] = None, use_wandb: bool = False, cfg: dict = None, logger=None,
                 tb_writer=None, **kwargs):
        self.model = model
        self.loss_fn = loss_fn
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.amp = bool(amp)
        self.grad_clip = float(grad_clip)
        self.device = device if isinstance(device, torch.device) else torch.device(device)
        self.cbs = list(callbacks)
        self.progress_bar = progress_bar
        self.debug = bool(debug)
        self.debug_dir = debug_dir if self.debug else None
        self.tb_logdir = tb_logdir
        self.use_wandb = bool(use_wandb)
        self.cfg = cfg or {}
        self.logger = logger if logger is not None else logging.getLogger("unet_amd")
        self.tb_writer = tb_writer
        self.scaler = torch.amp.GradScaler(enabled=False)      # bf16 needs no loss scaling
        self.latest_val_stats = {}

    def fit(self, train_loader, val_loader, epochs: int, start_epoch: int = 1, train_sampler=None):
        for cb in self.cbs:
            try:
                cb.on_train_begin(self)
            except Exception:
                self.logger.error("Callback on_train_begin failed")
        when = str(self.cfg.get("scheduler_step", "epoch")).lower()
        self._global_step = 0
        for epoch in range(start_epoch, epochs + 1):
            if train_sampler is not None:
                train_sampler.set_epoch(epoch)
            self.model.train()
            self.current_epoch = epoch
            total_pixels = correct_pixels = 0
            loss_sum = 0.0
            iterator = train_loader
            if self.progress_bar:
                iterator = __import__("tqdm").tqdm(train_loader, desc=f"Train {epoch}/{epochs}", leave=False)
            for batch_idx, batch in enumerate(iterator):
                x = batch[0].to(self.device, non_blocking=True)
                y = batch[1].to(self.device, non_blocking=True)
                try:
                    loss, logits, info = step_batch(self, x, y, scaler=self.scaler, batch_idx=batch_idx)
                except Exception:
                    self.logger.exception("Exception during training step")
                    raise
                bs = x.size(0)
                loss_sum += float(info["main_loss"]) * bs          # already on the host, no second sync
                try:
                    preds = ops.argmax(logits.detach().float().contiguous())
                    y_t = y.squeeze(1) if y.ndim == 4 else y
                    correct_pixels += int((preds == y_t).sum().item())
                    total_pixels += y_t.numel()
                except Exception:
                    pass
                if self.progress_bar:
                    iterator.set_postfix(loss=f"{info['main_loss']:.4f}", pix_acc=f"{correct_pixels / max(total_pixels, 1):.4f}")
            train_loss = loss_sum / len(train_loader.dataset)
            train_acc = correct_pixels / max(total_pixels, 1)
            val_loss, main_score, stats = evaluate(self, val_loader, epoch=epoch)
            self.latest_val_stats = stats
            if getattr(self, "scheduler", None) is not None and when == "epoch":
                try:
                    if isinstance(self.scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
                        self.scheduler.step(val_loss)
                    else:
                        self.scheduler.step()
                except Exception:
                    pass
            msg = f"Epoch {epoch:03d}/{epochs} | Tr.Loss={train_loss:.4f} Tr.PixAcc={train_acc:.4f} | Val.Loss={val_loss:.4f}"
            if "dice_infection" in stats:
                msg += f" Dice(Inf)={stats['dice_infection']:.4f}"
            if "mae" in stats:
                msg += f" MAE={stats['mae']:.4f}"
            self.logger.info(msg)
            for cb in self.cbs:
                try:
                    cb.on_epoch_end(self, epoch, train_loss, train_acc, val_loss, main_score)
                except Exception:
                    self.logger.exception("Callback on_epoch_end failed")
        for cb in self.cbs:
            try:
                getattr(cb, "on_train_end", lambda *_: None)(self)
            except Exception:
                pass

    def evaluate(self, loader, epoch=None):
        return evaluate(self, loader, epoch)
