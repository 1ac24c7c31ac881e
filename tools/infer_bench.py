#!/usr/bin/env python3
"""Inference throughput of the plain UNet (eval mode, bf16, no grad): BN folded into the conv vs conv -> BN -> ReLU.
SURVEY configuration C5 shape (1024x1024 slices); batch kept at 16 so one run stays short."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd.models import REGISTRY

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16); ap.add_argument("--size", type=int, default=1024); ap.add_argument("--iters", type=int, default=5); ap.add_argument("--ensemble", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[64, 128, 256, 512]).to(dev).eval()
m.compute_dtype = "bf16"
x = torch.rand(a.batch, 1, a.size, a.size, device=dev)
out = {}
with torch.no_grad():
    for fold in (True, False):
        m._engine.fold_bn_eval = fold
        for _ in range(2): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.iters): y = m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
        out["folded" if fold else "unfolded"] = {"ms_per_batch": 1e3 * dt, "slices_per_s": a.batch / dt}
# BASELINE configs[4]: UNet + deployed RepUNet ([3,5] -> one 5x5 conv per block, norm="batch" so the fold exists)
# + per-pixel entropy fusion (eval_ensemble.py:77-99)
from unet_amd import ops
if "--ensemble" in sys.argv:
    r = REGISTRY["reparam_unet"](in_channels=1, n_classes=2, features=[64, 128, 256, 512], kernel_sizes=[3, 5], norm="batch").to(dev).eval()
    r.compute_dtype = "bf16"
    r.switch_to_deploy()
    m._engine.fold_bn_eval = True
    with torch.no_grad():
        for _ in range(2): ops.ensemble_fuse(m(x).float(), r(x).float(), 0.6)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.iters): pred = ops.ensemble_fuse(m(x).float(), r(x).float(), 0.6)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
    out["ensemble_unet_plus_deployed_repunet"] = {"ms_per_batch": 1e3 * dt, "slices_per_s": a.batch / dt,
                                                  "s_per_301_slice_volume": 301 / (a.batch / dt)}
print(json.dumps({"workload": f"UNet eval bf16 {a.batch}x1x{a.size}x{a.size}", **out}))
