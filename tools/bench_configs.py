#!/usr/bin/env python3
"""Secondary benchmark lines for the BASELINE configs that bench.py (configs[1], the headline) does not time:

  --workload repunet_train   configs[2]: ReparamUNet per repar_unet_ct_segmentation.yaml (features [64,128,256,512],
                             kernel_sizes [3,5], GroupNorm as shipped), full step_batch on synthetic 512x512 slices, bf16
  --workload ensemble        configs[4]: UNet (eval, BN folded) + deployed ReparamUNet ([3,5] -> 5x5 convs) + per-pixel
                             entropy fusion on 1024x1024 slices, bf16, no grad (unet_amd.ensemble.EnsemblePredictor);
                             also the time for a 301-slice volume (nifti_analysis_results.csv:4-5)

One JSON line each, same shape as bench.py's.  Inputs are resident in HBM before the timed region.
"""
import argparse
import json
import logging
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FEATURES = [64, 128, 256, 512]
# forward GFLOP per 1024^2 slice, UNet 1539.07 and deployed RepUNet (5x5) 4152.6; RepUNet [3,5] training =
# 50 taps instead of 9 on every 3x3 layer of the UNet's 1154 GFLOP/image train step (first layer has no dgrad)
GFLOP_FWD_1024 = 1539.07 + 4152.6


def timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def repunet_train(a, dev):
    from unet_amd.losses import get_loss
    from unet_amd.models import REGISTRY
    from unet_amd.utils.builders import get_optim
    from unet_amd.utils.train_step import step_batch
    torch.manual_seed(42)
    m = REGISTRY["reparam_unet"](in_channels=1, n_classes=2, features=FEATURES, kernel_sizes=[3, 5]).to(dev).train()
    opt = get_optim("adamw", [p for p in m.parameters()], lr=1e-4, weight_decay=1e-4)
    tr = types.SimpleNamespace(optimizer=opt, amp=True, model=m, grad_clip=1.0, cbs=[], current_epoch=1,
                               logger=logging.getLogger("bench"), loss_fn=get_loss("ce_dice", ignore_index=255))
    g = torch.Generator().manual_seed(42)
    x = torch.rand(a.batch, 1, a.size, a.size, generator=g).to(dev)
    y = (torch.rand(a.batch, a.size, a.size, generator=g) > 0.95).long().to(dev)
    scaler = torch.amp.GradScaler(enabled=False)
    dt, last = timed(lambda: step_batch(tr, x, y, scaler, 1), a.warmup, a.steps)
    return {"metric": "RepUNet train images/sec (512x512 CT, fwd+bwd)", "value": a.batch / dt, "unit": "images/s", "n_gpus": 1,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt, "higher_is_better": True, "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"ReparamUNet features {FEATURES} kernel_sizes [3,5] GroupNorm, full step_batch, "
                                   f"{a.batch}x1x{a.size}x{a.size}", "final_loss": last[2]["main_loss"],
                       "params": sum(p.numel() for p in m.parameters())}}


def ensemble(a, dev):
    from unet_amd.ensemble import EnsemblePredictor
    from unet_amd.models import REGISTRY
    torch.manual_seed(42)
    u = REGISTRY["unet"](in_channels=1, n_classes=2, features=FEATURES).to(dev)
    r = REGISTRY["reparam_unet"](in_channels=1, n_classes=2, features=FEATURES, kernel_sizes=[3, 5], norm="batch").to(dev).eval()
    r.switch_to_deploy()
    ens = EnsemblePredictor(u, r, entropy_thresh=0.6, amp=True)
    x = torch.rand(a.batch, 1, a.size, a.size, generator=torch.Generator().manual_seed(42)).to(dev)
    dt, pred = timed(lambda: ens.predict(x, want_members=False), a.warmup, a.steps)
    sps = a.batch / dt
    return {"metric": "ensemble inference slices/sec (UNet + deployed RepUNet + entropy fusion)", "value": sps, "unit": "slices/s",
            "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt, "higher_is_better": True,
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"UNet(eval, BN folded) + deployed ReparamUNet [3,5]->5x5 + per-pixel entropy fusion, "
                                   f"{a.batch}x1x{a.size}x{a.size}, no grad", "seconds_per_301_slice_volume": 301.0 / sps,
                       "fwd_tflops": GFLOP_FWD_1024 * (a.size / 1024.0) ** 2 * sps / 1e3,
                       "foreground_fraction": float((pred > 0).float().mean()),
                       "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True, choices=["repunet_train", "ensemble"])
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.workload == "repunet_train":
        a.batch, a.size = a.batch or 8, a.size or 512
        out = repunet_train(a, dev)
    else:
        a.batch, a.size = a.batch or 64, a.size or 1024
        out = ensemble(a, dev)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
