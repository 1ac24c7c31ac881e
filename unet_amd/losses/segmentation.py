"""Dice + cross-entropy on the fused HIP kernels.  Mirrors /root/reference/losses/segmentation.py:5-27.

The forward kernel makes one pass over (logits, target) producing the per-(n,c) Dice totals and the
CE sum; backward makes one more pass and writes dlogits.  Labels the reference would raise on
(F.one_hot on a value >= n_classes, e.g. ignore_index 255 inside dice_loss) are counted on the device
and reported by `check_labels(loss)` / at step_batch's existing host sync, not silently accepted.
"""
import torch
import torch.nn as nn

from .. import ops


class _DiceCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ce_w, dice_w, ignore_index, ce_only, status_sink):
        lg = logits.detach()
        if lg.dtype != torch.float32:
            lg = lg.float()
        lg = lg.contiguous()
        tg = target.detach().contiguous()
        out4, acc = ops.dice_ce_fwd(lg, tg, ce_w, dice_w, ignore_index, ce_only)
        ctx.save_for_backward(lg, tg, acc)
        ctx.cfg = (ce_w, dice_w, ignore_index, ce_only, logits.dtype)
        status_sink.append(out4)
        return out4[0].clone()

    @staticmethod
    def backward(ctx, g):
        lg, tg, acc = ctx.saved_tensors
        ce_w, dice_w, ignore_index, ce_only, in_dtype = ctx.cfg
        dl = ops.dice_ce_bwd(lg, tg, acc, g, ce_w, dice_w, ignore_index, ce_only)
        if dl.dtype != in_dtype:
            dl = dl.to(in_dtype)
        return dl, None, None, None, None, None, None


def _run(logits, target, ce_w, dice_w, ignore_index, ce_only):
    sink = []
    loss = _DiceCEFn.apply(logits, target, float(ce_w), float(dice_w), int(ignore_index), bool(ce_only), sink)
    loss._unet_status = sink[0]          # {loss, ce, dice, illegal-label count} on the device
    return loss


def check_labels(loss):
    """Raise what the reference raises for out-of-range labels (one host sync)."""
    st = getattr(loss, "_unet_status", None)
    if st is not None and float(st[3]) > 0:
        raise RuntimeError("Class values must be smaller than num_classes.")


def dice_loss(logits, target, eps=1e-6):
    if eps != 1e-6:
        raise ops.UNetHipError("dice_loss: the fused kernel is built for the reference's eps=1e-6")
    return _run(logits, target, 0.0, 1.0, -100, False)


class CEDiceLoss(nn.Module):
    def __init__(self, ce_weight=1.0, dice_weight=1.0, ignore_index=-100):
        super().__init__()
        self.ce_weight = ce_weight
        self.dice_weight = dice_weight
        self.ce = nn.CrossEntropyLoss(ignore_index=ignore_index)   # attribute kept for API parity; not executed

    def forward(self, logits, target):
        return _run(logits, target, self.ce_weight, self.dice_weight, self.ce.ignore_index, False)


class CrossEntropyLoss(nn.Module):
    """The `ce` registry entry (reference: nn.CrossEntropyLoss(**kwargs)); mean reduction + ignore_index."""

    def __init__(self, ignore_index=-100, **kwargs):
        super().__init__()
        if kwargs:
            raise ops.UNetHipError(f"CrossEntropyLoss: unsupported arguments {sorted(kwargs)} (only ignore_index)")
        self.ignore_index = ignore_index

    def forward(self, logits, target):
        return _run(logits, target, 1.0, 0.0, self.ignore_index, True)
