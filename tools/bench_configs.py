#!/usr/bin/env python3
"""Secondary benchmark lines for the BASELINE configs that bench.py (configs[1], the headline) does not time:

  --workload repunet_train   configs[2]: ReparamUNet per repar_unet_ct_segmentation.yaml (features [64,128,256,512],
                             kernel_sizes [3,5], GroupNorm as shipped), full step_batch on synthetic 512x512 slices, bf16
  --workload ensemble        configs[4]: UNet (eval, BN folded) + deployed ReparamUNet ([3,5] -> 5x5 convs) + per-pixel
                             entropy fusion on 1024x1024 slices, bf16, no grad (unet_amd.ensemble.EnsemblePredictor);
                             also the time for a 301-slice volume (nifti_analysis_results.csv:4-5)

One JSON line each, same shape as bench.py's, including `roofline` (dominant kernel of the workload: the 5x5 instantiation of
the LDS-DMA conv kernel; algorithmic FLOPs of its launches / HIP-event durations of the LAST timed step on the launch stream;
`traffic` = PMC HBM bytes per launch from tools/pmc_cfg.sh, committed under profiles/) and `cpu_baseline` (the oracle on a
bounded sample on this host's cores).  Inputs are resident in HBM before the timed region.
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


MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class LaunchTimer:
    """HIP events around every ops.conv_fwd launch of the current stream (all bf16 LDS-DMA conv instantiations), keyed by
    (kh, kw); installed for ONE step (the event pairs cost launch overlap)."""

    def __init__(self):
        self.rec = []

    def install(self):
        from unet_amd import ops
        self._ops, self._orig = ops, ops.conv_fwd
        orig, rec = ops.conv_fwd, self.rec

        def timed_conv(x, wpacked, bias, y, kh, kw, Cin, Cout, *a, **k):
            if x.dt != ops.BF16 or k.get("epi", 0):
                return orig(x, wpacked, bias, y, kh, kw, Cin, Cout, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(x, wpacked, bias, y, kh, kw, Cin, Cout, *a, **k)
            e.record()
            rec.append(((kh, kw), 2.0 * x.M * kh * kw * Cin * Cout, s, e))
            return r
        ops.conv_fwd = timed_conv

    def remove(self):
        self._ops.conv_fwd = self._orig

    def block(self, taps, kernel, traffic_file):
        sel = [r for r in self.rec if r[0] == taps]
        if not sel:
            return None
        fl, ms = sum(r[1] for r in sel), sum(r[2].elapsed_time(r[3]) for r in sel)
        allfl, allms = sum(r[1] for r in self.rec), sum(r[2].elapsed_time(r[3]) for r in self.rec)
        traffic = tb = None
        try:
            with open(os.path.join(ROOT, traffic_file)) as f:
                j = json.load(f)
            traffic, tb = float(j["hbm_bytes_per_launch"]), j.get("build")
        except (OSError, KeyError, ValueError):
            pass
        ach = fl / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_file, "traffic_build": tb, "kernel": kernel, "launches": len(sel),
                "avg_launch_ms": ms / len(sel), "avg_launch_gflop": fl / len(sel) / 1e9,
                "share_of_conv_time": ms / allms, "all_conv_launches": len(self.rec),
                "all_conv_tflops": allfl / (allms * 1e-3) / 1e12}


def timed(fn, warmup, steps, timer=None):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if timer is not None and i == steps - 1:
            timer.install()
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if timer is not None:
        timer.remove()
    return dt, out


def lib_build_id():
    import hashlib
    from unet_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:12]


def cpu_repunet_train(size=512, iters=2):
    """oracle.repunet_forward + Dice/CE + backward (GroupNorm, [3,5], real widths) on one slice: ~6.4 TFLOP per iteration."""
    from oracle import loss_ref, unet_ref
    from unet_amd.models import REGISTRY
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    m = REGISTRY["reparam_unet"](in_channels=1, n_classes=2, features=FEATURES, kernel_sizes=[3, 5])
    st = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
    del m
    g = torch.Generator().manual_seed(42)
    x = torch.rand(1, 1, size, size, generator=g)
    y = (torch.rand(1, size, size, generator=g) > 0.95).long()
    times = []
    for _ in range(iters + 1):
        for v in st.values():
            v.grad = None
        t0 = time.perf_counter()
        loss_ref.ce_dice(unet_ref.repunet_forward(x, st, [3, 5], norm="group", train=True), y, 1.0, 1.0, 255).backward()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {"value": 1.0 / best, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch CPU fp32 restatement) RepUNet [3,5] GroupNorm fwd+loss+bwd on 1x1x{size}x{size}, best of {iters} "
                      f"after 1 warm-up, {best:.2f} s/iter"}


def cpu_ensemble(size=1024, iters=2):
    """oracle: UNet eval forward + folded RepUNet (5x5) forward + per-pixel fusion on one slice: ~5.7 TFLOP per iteration."""
    from oracle import loss_ref, unet_ref
    from unet_amd.models import REGISTRY
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    ust = {k: v.clone() for k, v in REGISTRY["unet"](in_channels=1, n_classes=2, features=FEATURES).state_dict().items()}
    rst = {k: v.clone() for k, v in REGISTRY["reparam_unet"](in_channels=1, n_classes=2, features=FEATURES, kernel_sizes=[3, 5],
                                                             norm="batch").state_dict().items()}
    folded = unet_ref.fold_state(rst, [3, 5])
    del rst
    x = torch.rand(1, 1, size, size, generator=torch.Generator().manual_seed(42))
    times = []
    with torch.no_grad():
        for _ in range(iters + 1):
            t0 = time.perf_counter()
            l1 = unet_ref.unet_forward(x, ust, train=False)
            l2 = unet_ref.repunet_forward(x, folded, [3, 5], norm="batch", train=False)
            loss_ref.ensemble_fuse(l1, l2, 0.6)
            times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {"value": 1.0 / best, "unit": "slices/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch CPU fp32 restatement) UNet eval + folded RepUNet 5x5 + entropy fusion on 1x1x{size}x{size}, "
                      f"best of {iters} after 1 warm-up, {best:.2f} s/iter"}


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
    timer = None if a.no_timer else LaunchTimer()
    dt, last = timed(lambda: step_batch(tr, x, y, scaler, 1), a.warmup, a.steps, timer)
    extra = {}
    if timer is not None:
        extra["roofline"] = timer.block((5, 5), "conv3x3_v2_kernel<1,...,5,5> (bf16 5x5 branch convs: forward + accumulating dgrad launches)",
                                        "profiles/r03_pmc_traffic_repunet_train.json")
        extra["build"] = lib_build_id()
    if not a.no_cpu_baseline:
        extra["cpu_baseline"] = cpu_repunet_train()
    return {**extra, "metric": "RepUNet train images/sec (512x512 CT, fwd+bwd)", "value": a.batch / dt, "unit": "images/s", "n_gpus": 1,
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
    timer = None if a.no_timer else LaunchTimer()
    dt, pred = timed(lambda: ens.predict(x, want_members=False), a.warmup, a.steps, timer)
    sps = a.batch / dt
    extra = {}
    if timer is not None:
        extra["roofline"] = timer.block((5, 5), "conv3x3_v2_kernel<1,...,5,5> (bf16 5x5 convs of the deployed RepUNet)",
                                        "profiles/r03_pmc_traffic_ensemble.json")
        extra["build"] = lib_build_id()
    if not a.no_cpu_baseline:
        extra["cpu_baseline"] = cpu_ensemble()
    return {**extra, "metric": "ensemble inference slices/sec (UNet + deployed RepUNet + entropy fusion)", "value": sps, "unit": "slices/s",
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-timer", action="store_true")
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
