"""Oracle: Dice + CE loss, ensemble fusion, segmentation metrics, optimizer tail
(torch CPU fp32; test infrastructure only).

Reference followed:
  dice_loss                       /root/reference/losses/segmentation.py:5-13
  CEDiceLoss.forward              /root/reference/losses/segmentation.py:23-27
  prediction_entropy              /root/reference/eval_ensemble.py:9-11
  entropy-weighted fusion         /root/reference/eval_ensemble.py:77-99 (applied per pixel)
  calculate_segmentation_metrics  /root/reference/utils/eval_utils.py:8-57
  clip + AdamW step               /root/reference/utils/train_step.py:58-63
"""
import torch
import torch.nn.functional as F


def dice(logits, target, eps=1e-6):
    """segmentation.py:5-13. No ignore_index handling (one_hot raises on 255)."""
    c = logits.shape[1]
    p = torch.softmax(logits, 1)
    oh = F.one_hot(target, c).permute(0, 3, 1, 2).float()
    inter = (p * oh).sum((2, 3))
    union = p.sum((2, 3)) + oh.sum((2, 3))
    return 1.0 - ((2 * inter + eps) / (union + eps)).mean()


def cross_entropy(logits, target, ignore_index=-100):
    """nn.CrossEntropyLoss(ignore_index): mean of -log_softmax[target] over non-ignored pixels."""
    lsm = torch.log_softmax(logits, 1)
    valid = target != ignore_index
    t = torch.where(valid, target, torch.zeros_like(target))
    nll = -lsm.gather(1, t[:, None])[:, 0]
    return (nll * valid).sum() / valid.sum()


def ce_dice(logits, target, ce_weight=1.0, dice_weight=1.0, ignore_index=-100):
    """segmentation.py:23-27."""
    return ce_weight * cross_entropy(logits, target, ignore_index) + dice_weight * dice(logits, target)


def entropy(p):
    """eval_ensemble.py:9-11."""
    return -(p * p.clamp(min=1e-8).log()).sum(1)


def ensemble_fuse(logits1, logits2, thresh=0.6):
    """eval_ensemble.py:77-99 with every pixel treated as one sample: (B,C,H,W) -> (B,H,W) int64."""
    p1, p2 = torch.softmax(logits1, 1), torch.softmax(logits2, 1)
    e1, e2 = entropy(p1), entropy(p2)
    diff = (e1 - e2).abs()
    w1, w2 = 1 / (e1 + 1e-8), 1 / (e2 + 1e-8)
    p_ens = (w1[:, None] * p1 + w2[:, None] * p2) / (w1 + w2)[:, None]
    hc1 = (diff > thresh) & (e1 < e2)
    hc2 = (diff > thresh) & (e2 < e1)
    pred = p_ens.argmax(1)
    pred = torch.where(hc1, p1.argmax(1), pred)
    pred = torch.where(hc2, p2.argmax(1), pred)
    return pred


def seg_metrics(pred, target, num_classes, eps=1e-6):
    """eval_utils.py:8-57, same eps placement; per batch."""
    res = {"acc_pixel": (pred == target).sum().item() / max(target.numel(), 1),
           "mae": (pred.float() - target.float()).abs().mean().item()}
    ds = ss = sp = 0.0
    for c in range(num_classes):
        p = (pred == c).float().view(-1)
        t = (target == c).float().view(-1)
        tp, fp = (p * t).sum(), (p * (1 - t)).sum()
        fn, tn = ((1 - p) * t).sum(), ((1 - p) * (1 - t)).sum()
        d = ((2. * tp + eps) / (2. * tp + fp + fn + eps)).item()
        s = ((tp + eps) / (tp + fn + eps)).item()
        q = ((tn + eps) / (tn + fp + eps)).item()
        res[f"dice_class_{c}"], res[f"sen_class_{c}"], res[f"spec_class_{c}"] = d, s, q
        ds, ss, sp = ds + d, ss + s, sp + q
    res["dice_macro"], res["sen_macro"], res["spec_macro"] = ds / num_classes, ss / num_classes, sp / num_classes
    return res


def clip_adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr, weight_decay, max_norm,
                    betas=(0.9, 0.999), eps=1e-8):
    """train_step.py:58-63 with torch defaults: clip_grad_norm_(max_norm) (coef = max_norm/(norm+1e-6),
    clamped to 1) then decoupled-decay AdamW.  Lists of tensors, updated in place; returns the norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0) if max_norm > 0 else torch.tensor(1.0)
    b1, b2 = betas
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        g = g * coef
        p.mul_(1 - lr * weight_decay)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)
    return total
