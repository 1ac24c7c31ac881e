#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by importing the REAL reference.

Run in the build container only (``/root/reference`` is not present on the GPU box):

    python tests/golden/make_goldens.py

Every fixture is data: seeded inputs plus the outputs the reference's own code produced
for them (torch CPU fp32).  No reference source is copied; the modules are imported from
``/root/reference`` and called.  The torch version that produced the numbers is recorded
in ``meta.json``.

What is captured:
  unet_f8x16.npz         reference UNet(features=[8,16]) 2x1x32x32: logits, CE, Dice, loss,
                         every parameter grad, BN buffers after one train forward, parameters
                         after one reference ``step_batch`` (AdamW 1e-4/1e-4, clip 1.0)
  unet_f8x16x32x64.npz   the same for a 4-level model on 2x1x64x64
  repunet_gn_f8x16_k35.npz  as-shipped (GroupNorm) ReparamUNet, kernel_sizes [3,5], fwd+bwd
  fold_k3.npz / fold_k35.npz  BatchNorm-variant ReparamUNet fold (switch_to_deploy): pre-fold
                         state, post-fold mergedConv weights/biases, eval logits before/after
  loss_cases.npz         Dice / CE / CE+Dice edge cases (all background, ignore_index for CE)
  ensemble.npz           eval_ensemble.evaluate() driven per pixel, captured predictions
  metrics.json           calculate_segmentation_metrics for a fixed pred/target pair
  cfg_unet.json / cfg_repunet.json  load_cfg() of the two shipped YAMLs
"""
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _np(t):
    return t.detach().cpu().numpy()


def synth_batch(n, h, seed=42, fg=0.95):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 1, h, h, generator=g)
    y = (torch.rand(n, h, h, generator=g) > fg).long()
    return x, y


def state_digest(sd):
    m = hashlib.sha256()
    for k, v in sd.items():
        m.update(k.encode())
        m.update(_np(v).tobytes())
    return m.hexdigest()


class _Log:
    def debug(self, *a, **k):
        pass

    info = warning = error = exception = debug


def whole_model(name, ctor, ctor_kwargs, n, h, do_step=True):
    """forward / loss / backward of the reference model, then one reference step_batch."""
    from losses import get_loss
    from losses.segmentation import dice_loss
    from utils.train_step import step_batch

    torch.manual_seed(42)
    model = ctor(**ctor_kwargs)
    model.train()
    init_sd = {k: v.clone() for k, v in model.state_dict().items()}
    x, y = synth_batch(n, h)
    loss_fn = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)

    logits = model(x)
    loss = loss_fn(logits, y)
    ce = nn.CrossEntropyLoss(ignore_index=255)(logits, y)
    dice = dice_loss(logits, y)
    logits.retain_grad()
    loss.backward()

    out = {
        "x": _np(x), "y": _np(y), "logits": _np(logits), "dlogits": _np(logits.grad),
        "loss": _np(loss), "ce": _np(ce), "dice": _np(dice),
        "argmax": _np(logits.argmax(1)),
    }
    for k, p in model.named_parameters():
        out["grad/" + k] = _np(p.grad)
    for k, b in model.named_buffers():
        out["buf1/" + k] = _np(b)          # buffers after ONE train-mode forward
    for k, v in init_sd.items():
        out["init/" + k] = _np(v)

    if do_step:
        # fresh, identically initialised model through the reference's own step_batch
        torch.manual_seed(42)
        model2 = ctor(**ctor_kwargs)
        model2.train()
        assert state_digest(model2.state_dict()) == state_digest(init_sd)
        opt = torch.optim.AdamW([p for p in model2.parameters() if p.requires_grad],
                                lr=1e-4, weight_decay=1e-4)
        tr = types.SimpleNamespace(optimizer=opt, amp=False, model=model2, loss_fn=loss_fn,
                                   grad_clip=1.0, cbs=[], logger=_Log(), current_epoch=1)
        scaler = torch.amp.GradScaler(enabled=False)
        l2, lg2, info = step_batch(tr, x, y, scaler, batch_idx=0)
        out["step/main_loss"] = np.float64(info["main_loss"])
        out["step/lr"] = np.float64(info["lr"])
        out["step/global_step"] = np.int64(tr._global_step)
        for k, p in model2.named_parameters():
            out["step/" + k] = _np(p)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    return {"init_sha256": state_digest(init_sd), "loss": float(loss)}


def fold_case(name, kernel_sizes):
    """BatchNorm-variant fold oracle (SURVEY 8c 'Fold oracle')."""
    import models.reparm_unet as R

    orig = R.getBN
    R.getBN = lambda c, eps=1e-5, momentum=0.01, affine=True: nn.BatchNorm2d(
        c, eps=eps, momentum=momentum, affine=affine)
    try:
        torch.manual_seed(7)
        m = R.ReparamUNet(in_channels=1, n_classes=2, features=[8, 16], kernel_sizes=list(kernel_sizes))
        g = torch.Generator().manual_seed(123)
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.5)
                mod.running_var.copy_(torch.rand(mod.num_features, generator=g) * 2 + 0.25)
                mod.weight.data.copy_(torch.randn(mod.num_features, generator=g))
                mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.3)
        m.eval()
        pre = {k: v.clone() for k, v in m.state_dict().items()}
        x, _ = synth_batch(2, 32, seed=5)
        with torch.no_grad():
            y_multi = m(x)
        m.switch_to_deploy()
        with torch.no_grad():
            y_deploy = m(x)
        post = m.state_dict()
        out = {"x": _np(x), "logits_multibranch": _np(y_multi), "logits_deploy": _np(y_deploy)}
        for k, v in pre.items():
            out["pre/" + k] = _np(v)
        for k, v in post.items():
            out["post/" + k] = _np(v)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        return {"max_abs_deploy_vs_multi": float((y_multi - y_deploy).abs().max())}
    finally:
        R.getBN = orig


def loss_cases():
    from losses import get_loss
    from losses.segmentation import dice_loss

    g = torch.Generator().manual_seed(11)
    out = {}
    # case A: random logits, ~5% foreground, 2 classes
    la = torch.randn(2, 2, 16, 24, generator=g) * 2
    ya = (torch.rand(2, 16, 24, generator=g) > 0.95).long()
    # case B: all background
    lb = torch.randn(2, 2, 16, 24, generator=g)
    yb = torch.zeros(2, 16, 24, dtype=torch.long)
    # case C: 3 classes
    lc = torch.randn(3, 3, 8, 8, generator=g) * 3
    yc = torch.randint(0, 3, (3, 8, 8), generator=g)
    for tag, lg, y in (("A", la, ya), ("B", lb, yb), ("C", lc, yc)):
        lg = lg.clone().requires_grad_(True)
        f = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)
        loss = f(lg, y)
        loss.backward()
        out[f"{tag}/logits"] = _np(lg)
        out[f"{tag}/target"] = _np(y)
        out[f"{tag}/loss"] = _np(loss)
        out[f"{tag}/dlogits"] = _np(lg.grad)
        out[f"{tag}/dice"] = _np(dice_loss(lg.detach(), y))
        out[f"{tag}/ce"] = _np(nn.CrossEntropyLoss(ignore_index=255)(lg.detach(), y))
    # case D: weighted combination 0.3 / 0.7
    lg = la.clone().requires_grad_(True)
    f = get_loss("ce_dice", ce_weight=0.3, dice_weight=0.7, ignore_index=255)
    loss = f(lg, ya)
    loss.backward()
    out["D/loss"] = _np(loss)
    out["D/dlogits"] = _np(lg.grad)
    # case E: CE only with ignore_index pixels (dice_loss cannot take 255: one_hot raises)
    ye = ya.clone()
    ye[0, :4, :] = 255
    lg = la.clone().requires_grad_(True)
    f = get_loss("ce", ignore_index=255)
    loss = f(lg, ye)
    loss.backward()
    out["E/target"] = _np(ye)
    out["E/loss"] = _np(loss)
    out["E/dlogits"] = _np(lg.grad)
    try:
        dice_loss(la, ye)
        out["E/dice_raises"] = np.int64(0)
    except RuntimeError:
        out["E/dice_raises"] = np.int64(1)
    np.savez_compressed(os.path.join(OUT, "loss_cases.npz"), **out)


def ensemble_case():
    """Drive the real eval_ensemble.evaluate() with every pixel as one sample."""
    from torch.utils.data import TensorDataset

    g = torch.Generator().manual_seed(21)
    B, C, H, W = 2, 2, 24, 24
    l1 = torch.randn(B, C, H, W, generator=g) * 2.0
    l2 = torch.randn(B, C, H, W, generator=g) * 0.7
    # make a band of pixels where model 2 is very confident and model 1 is not, and vice versa
    l2[:, 0, :6, :] += 6.0
    l1[:, 1, 6:12, :] += 6.0
    l1[:, :, 12:14, :] = 0.0      # exact ties for model 1
    l2[:, :, 12:13, :] = 0.0      # both exactly tied on one row
    tgt = torch.randint(0, C, (B, H, W), generator=g)
    flat1 = l1.permute(0, 2, 3, 1).reshape(-1, C)
    flat2 = l2.permute(0, 2, 3, 1).reshape(-1, C)
    xs = torch.cat([flat1, flat2], 1)

    fake_data = types.ModuleType("data")
    fake_data.REGISTRY = {"px": lambda data_dir, **kw: (None, TensorDataset(xs, tgt.reshape(-1)))}
    saved = sys.modules.get("data")
    sys.modules["data"] = fake_data
    try:
        import eval_ensemble as E
    finally:
        if saved is not None:
            sys.modules["data"] = saved
        else:
            del sys.modules["data"]

    class Pick(nn.Module):
        def __init__(self, lo):
            super().__init__()
            self.lo = lo

        def forward(self, x):
            return x[:, self.lo:self.lo + C]

    E.MODEL_REG = {"m1": lambda **kw: Pick(0), "m2": lambda **kw: Pick(C)}
    E.load_checkpoint = lambda path, model: model
    captured = []
    orig_acc = E.compute_classwise_acc

    def spy(preds, targets, num_classes=10):
        captured.append(preds.clone())
        return orig_acc(preds, targets, num_classes=num_classes)

    E.compute_classwise_acc = spy
    with tempfile.TemporaryDirectory() as td:
        import yaml
        base = dict(dataset="px", data_dir="", batch_size=100, num_workers=0, pin_memory=False,
                    persistent_workers=False, model_args={})
        p1, p2 = os.path.join(td, "a.yaml"), os.path.join(td, "b.yaml")
        yaml.safe_dump(dict(base, model="m1"), open(p1, "w"))
        yaml.safe_dump(dict(base, model="m2"), open(p2, "w"))
        E.evaluate(p1, "x", p2, "x", out_csv=os.path.join(td, "o.csv"), device="cpu", entropy_thresh=0.6)
    preds1, preds2, preds_ens = captured
    np.savez_compressed(
        os.path.join(OUT, "ensemble.npz"),
        logits1=_np(l1), logits2=_np(l2), entropy_thresh=np.float64(0.6),
        pred1=_np(preds1.reshape(B, H, W)), pred2=_np(preds2.reshape(B, H, W)),
        pred_ens=_np(preds_ens.reshape(B, H, W)))
    return {"n_ens_ne_1": int((preds_ens != preds1).sum()), "n_ens_ne_2": int((preds_ens != preds2).sum())}


def metrics_case():
    # utils.eval_utils imports loguru (absent): give it an inert stand-in for the import only
    lg = types.ModuleType("loguru")
    lg.logger = _Log()
    sys.modules.setdefault("loguru", lg)
    from utils.eval_utils import calculate_segmentation_metrics

    g = torch.Generator().manual_seed(31)
    pred = (torch.rand(2, 20, 20, generator=g) > 0.8).long()
    tgt = (torch.rand(2, 20, 20, generator=g) > 0.85).long()
    res = calculate_segmentation_metrics(pred, tgt, num_classes=2)
    # degenerate: class 1 absent from both
    z = torch.zeros(1, 6, 6, dtype=torch.long)
    res0 = calculate_segmentation_metrics(z, z, num_classes=2)
    json.dump({"pred": pred.tolist(), "target": tgt.tolist(), "metrics": res,
               "empty_metrics": res0}, open(os.path.join(OUT, "metrics.json"), "w"))


def cfg_case():
    from utils.config_utils import load_cfg
    for src, dst in (("configs/unet_ct_segmentation.yaml", "cfg_unet.json"),
                     ("configs/repar_unet_ct_segmentation.yaml", "cfg_repunet.json")):
        cfg = load_cfg(os.path.join(REF, src))
        json.dump(cfg, open(os.path.join(OUT, dst), "w"), indent=1, sort_keys=True)
    cfg = load_cfg(os.path.join(REF, "configs/unet_ct_segmentation.yaml"),
                   ["batch_size=32", "optim_args.lr=3e-4", "model_args.features=[8,16]", "exp_name=abc"])
    json.dump(cfg, open(os.path.join(OUT, "cfg_unet_over.json"), "w"), indent=1, sort_keys=True)


def main():
    if not os.path.isdir(REF):
        print("reference not present; goldens are committed, nothing to do")
        return
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    torch.use_deterministic_algorithms(True)
    from models import REGISTRY

    meta = {"torch": torch.__version__, "numpy": np.__version__}
    meta["unet_f8x16"] = whole_model(
        "unet_f8x16", REGISTRY["unet"], dict(in_channels=1, n_classes=2, features=[8, 16]), 2, 32)
    meta["unet_f8x16x32x64"] = whole_model(
        "unet_f8x16x32x64", REGISTRY["unet"],
        dict(in_channels=1, n_classes=2, features=[8, 16, 32, 64]), 2, 64)
    meta["unet_f16_c3"] = whole_model(
        "unet_f16_c3", REGISTRY["unet"], dict(in_channels=1, features=[16]), 1, 16, do_step=False)
    meta["repunet_gn_f8x16_k35"] = whole_model(
        "repunet_gn_f8x16_k35", REGISTRY["reparam_unet"],
        dict(in_channels=1, n_classes=2, features=[8, 16], kernel_sizes=[3, 5]), 2, 32)
    meta["repunet_gn_f8_k3"] = whole_model(
        "repunet_gn_f8_k3", REGISTRY["reparam_unet"],
        dict(in_channels=1, n_classes=2, features=[8]), 2, 16, do_step=False)
    meta["fold_k3"] = fold_case("fold_k3", [3])
    meta["fold_k35"] = fold_case("fold_k35", [3, 5])
    loss_cases()
    meta["ensemble"] = ensemble_case()
    metrics_case()
    cfg_case()
    json.dump(meta, open(os.path.join(OUT, "meta.json"), "w"), indent=1, sort_keys=True)
    for f in sorted(os.listdir(OUT)):
        print(f"{os.path.getsize(os.path.join(OUT, f)):>10}  {f}")


if __name__ == "__main__":
    main()
