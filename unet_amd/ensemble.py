"""Two-model entropy-weighted ensemble inference (BASELINE configs[4]; reference /root/reference/eval_ensemble.py:61-99).

The reference builds two models from two configs, runs both on every batch, and fuses their softmax outputs:
confidence = inverse prediction entropy; where the entropies differ by more than `entropy_thresh` the more confident
model decides alone, otherwise the entropy-weighted average does (eval_ensemble.py:77-99).  That code is written for
(B, C) classification logits; on segmentation logits (B, C, H, W) the same rule is applied per pixel.

Here: model 1 = UNet in eval mode (BatchNorm folded into the convs), model 2 = a deployed ReparamUNet (one k_max x k_max
conv per block), both bf16 on the HIP engine under torch.no_grad(); softmax, entropies, weights and the three-way decision
are one HIP kernel per pixel (`unet_ensemble_fuse`), so the only full-resolution tensors that exist are the two fp32
logits and the int64 prediction.  Replicas-only across GPUs (shard the slices; no collective).
"""
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import ops
from ._lib import UNetHipError


def get_logits(model, x):
    """eval_ensemble.py:14-19: models that return three heads are summed; everything else is the logits."""
    outputs = model(x)
    if isinstance(outputs, (tuple, list)) and len(outputs) == 3:
        return outputs[0] + outputs[1] + outputs[2]
    return outputs


class EnsemblePredictor:
    """predict(x) -> (pred_ens, pred1, pred2), int64 [N,H,W] each -- the three prediction sets eval_ensemble.evaluate
    collects per batch (:101-105)."""

    def __init__(self, model1, model2, entropy_thresh: float = 0.6, amp: bool = True):
        self.model1, self.model2 = model1, model2
        self.entropy_thresh = float(entropy_thresh)
        self.amp = bool(amp)
        model1.eval()
        model2.eval()

    @torch.no_grad()
    def logits(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if not x.is_cuda:
            raise UNetHipError("EnsemblePredictor runs on MI355X only (input is a CPU tensor)")
        with torch.amp.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=self.amp):
            l1 = get_logits(self.model1, x)
            l2 = get_logits(self.model2, x)
        if l1.shape != l2.shape:
            raise UNetHipError(f"the two models disagree on the output shape: {tuple(l1.shape)} vs {tuple(l2.shape)}")
        return l1.float().contiguous(), l2.float().contiguous()

    @torch.no_grad()
    def predict(self, x: torch.Tensor, want_members: bool = True):
        l1, l2 = self.logits(x)
        ens = ops.ensemble_fuse(l1, l2, self.entropy_thresh)
        if not want_members:
            return ens
        return ens, ops.argmax(l1), ops.argmax(l2)

    @torch.no_grad()
    def evaluate(self, loader: Iterable, device, num_classes: Optional[int] = None) -> Dict[str, object]:
        """Accuracies of model 1, model 2 and the ensemble over a loader of (x, y) batches, from per-batch confusion counts
        (`unet_seg_counts`); y may be [N,H,W] or [N,1,H,W].  Returns {"Model1" | "Model2" | "Ensemble": {"overall",
        "per_class"}} with the reference's definitions (eval_ensemble.py:22-30,123-130: overall = mean(pred == target),
        per_class[c] = correct_c / max(total_c, 1) over the pixels whose target is c), plus the flat `acc_pixel_*` / `pixels`
        keys.  The reference's head / medium / tail groups are its CIFAR-LT class split (:113-118) and have no meaning for
        segmentation classes: not produced.  num_classes defaults to the logits' channel count."""
        names = ("Model1", "Model2", "Ensemble")
        conf = None
        total = 0
        K = num_classes
        for batch in loader:
            x = batch[0].to(device, non_blocking=True)
            y = batch[1].to(device, non_blocking=True)
            if y.ndim == 4:
                y = y.squeeze(1)
            l1, l2 = self.logits(x)
            if K is None:
                K = int(l1.shape[1])
            if conf is None:
                conf = {n: torch.zeros(K, K, dtype=torch.int64) for n in names}
            preds = (ops.argmax(l1), ops.argmax(l2), ops.ensemble_fuse(l1, l2, self.entropy_thresh))
            for name, p in zip(names, preds):
                c = ops.seg_counts(p, y.long(), K).cpu()
                conf[name] += c[:K * K].view(K, K)              # [pred][target]
            total += y.numel()
        out: Dict[str, object] = {"pixels": total}
        for name, flat in zip(names, ("model1", "model2", "ensemble")):
            m = conf[name] if conf is not None else torch.zeros(0, 0, dtype=torch.int64)
            per_target = m.sum(0)
            overall = float(m.diag().sum()) / total if total else float("nan")
            out[name] = {"overall": overall,
                         "per_class": (m.diag().float() / per_target.clamp(min=1).float()).tolist()}
            out[f"acc_pixel_{flat}"] = overall
        return out


def build_ensemble(cfg1: dict, state1: Optional[dict], cfg2: dict, state2: Optional[dict], device,
                   entropy_thresh: float = 0.6, amp: bool = True) -> EnsemblePredictor:
    """Models from two reference-schema configs (`model`, `model_args`) and optional state_dicts, as
    eval_ensemble.evaluate does (:61-65).  A ReparamUNet with branches left (norm='batch') is deployed here."""
    from .models import REGISTRY
    models = []
    for cfg, state in ((cfg1, state1), (cfg2, state2)):
        m = REGISTRY[cfg["model"]](**cfg.get("model_args", {}))
        if state is not None:
            m.load_state_dict(state, strict=False)
        m = m.to(device).eval()
        if hasattr(m, "switch_to_deploy") and any(".conv_list." in k for k in m.state_dict()):
            m.switch_to_deploy()
        models.append(m)
    return EnsemblePredictor(models[0], models[1], entropy_thresh, amp)
