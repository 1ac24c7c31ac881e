"""Evaluation with the reference's metric definitions (/root/reference/utils/eval_utils.py:8-145).

calculate_segmentation_metrics: one fused confusion-count kernel + one host copy per batch instead
of ~4+7C `.item()` syncs; the metrics are derived on the host with the reference's eps placement and
stay PER BATCH (the reference averages per-batch metrics with batch-size weights, :104-107,116).
"""
import logging
from typing import Any, Dict, Tuple

import torch

from .. import ops
from ..losses import check_labels

logger = logging.getLogger("unet_amd")


def metrics_from_counts(counts, total, num_classes, epsilon=1e-6) -> Dict[str, float]:
    """counts: 66 ints (K*K confusion, pred-major; [64] = sum|p-t|).  fp32 arithmetic like the reference."""
    f32 = torch.float32
    K = num_classes
    conf = torch.tensor(counts[:K * K], dtype=torch.float64).view(K, K)
    res = {"acc_pixel": float(conf.diag().sum().item()) / max(total, 1),
           "mae": float((torch.tensor(float(counts[64]), dtype=f32) / max(total, 1)).item()) if total else float("nan")}
    tot = torch.tensor(float(total), dtype=f32)
    ds = ss = sp = 0.0
    eps = torch.tensor(epsilon, dtype=f32)
    for c in range(K):
        tp = conf[c, c].to(f32)
        fp = (conf[c, :].sum() - conf[c, c]).to(f32)
        fn = (conf[:, c].sum() - conf[c, c]).to(f32)
        tn = tot - tp - fp - fn
        dice = ((2. * tp + eps) / (2. * tp + fp + fn + eps)).item()
        sens = ((tp + eps) / (tp + fn + eps)).item()
        spec = ((tn + eps) / (tn + fp + eps)).item()
        res[f"dice_class_{c}"], res[f"sen_class_{c}"], res[f"spec_class_{c}"] = dice, sens, spec
        ds, ss, sp = ds + dice, ss + sens, sp + spec
    res["dice_macro"], res["sen_macro"], res["spec_macro"] = ds / K, ss / K, sp / K
    return res


def calculate_segmentation_metrics(preds, targets, num_classes, epsilon=1e-6):
    counts = ops.seg_counts(preds, targets, num_classes).tolist()
    return metrics_from_counts(counts, targets.numel(), num_classes, epsilon)


@torch.no_grad()
def evaluate(trainer, loader, epoch=-1) -> Tuple[float, float, Dict[str, Any]]:
    model = trainer.model
    was_training = model.training
    model.eval()
    device = trainer.device
    requested = set(trainer.cfg.get("metrics", []))
    monitor_key = trainer.cfg.get("early_stop", {}).get("monitor", "val_loss").replace("val/", "")
    n_classes = trainer.cfg.get("model_args", {}).get("n_classes", None)
    amp_on = getattr(trainer, "amp", False)

    total_loss, total_batches, sums = 0.0, 0, {}
    for batch in loader:
        try:
            x = batch[0].to(device, non_blocking=True)
            y = batch[1].to(device, non_blocking=True)
            if y.ndim == 4:
                y = y.squeeze(1)
            with torch.amp.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=bool(amp_on)):
                try:
                    outputs = model(x)
                except TypeError:
                    outputs = model(x, current_epoch=epoch)
            if isinstance(outputs, dict):
                logits = outputs.get("logits", outputs.get("out"))
            elif isinstance(outputs, tuple):
                logits = outputs[0]
            else:
                logits = outputs
            loss = trainer.loss_fn(logits, y)
            check_labels(loss)
            bs = x.size(0)
            total_loss += loss.item() * bs
            total_batches += bs
            if n_classes is None:
                n_classes = logits.size(1)
            preds = ops.argmax(logits.float().contiguous())
            for k, v in calculate_segmentation_metrics(preds, y, num_classes=n_classes).items():
                sums[k] = sums.get(k, 0.0) + v * bs
        except Exception as e:
            logger.exception(f"Evaluation failed at batch {total_batches}: {e}")
            raise
    if total_batches == 0:
        return 0.0, 0.0, {}
    avg_loss = total_loss / total_batches
    full = {k: v / total_batches for k, v in sums.items()}
    full["val_loss"] = avg_loss
    final = {"val_loss": avg_loss}
    for k in requested:
        if k in full:
            final[k] = full[k]
    if monitor_key in full and monitor_key not in final:
        final[monitor_key] = full[monitor_key]
    main_score = final.get(monitor_key, avg_loss)
    parts = [f"Loss={avg_loss:.4f}"] + [f"{k}={final[k]:.4f}" for k in sorted(requested) if k in final]
    logger.info(f"Eval Epoch {epoch}: " + " | ".join(parts))
    if getattr(trainer, "tb_writer", None):
        for k, v in final.items():
            trainer.tb_writer.add_scalar(f"val/{k}", v, epoch)
    if was_training:
        model.train()
    return avg_loss, main_score, final
