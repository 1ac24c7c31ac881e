#!/usr/bin/env python3
"""Times the ConvTranspose2d(k=2,s=2) forward (into a channel slice of the concat buffer), its dgrad and its wgrad
at the four decoder shapes of the C2 workload (UNet 32x1x512x512, bf16)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops  # noqa: E402
from unet_amd.ops import Act  # noqa: E402
from conv_microbench import timeit  # noqa: E402

SHAPES = [("ups.0", 1024, 512, 32), ("ups.2", 512, 256, 64), ("ups.4", 256, 128, 128), ("ups.6", 128, 64, 256)]   # Ci, Co, low-res H


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = ops.BF16
    ws = ops.Workspace(dev)
    tot = 0.0
    for name, ci, co, h in SHAPES:
        if a.only and a.only not in name:
            continue
        N = a.batch
        x = Act((torch.randn((N, h, h, ci), device=dev) * 0.5).to(torch.bfloat16), 0, ci)
        cat = torch.empty((N, 2 * h, 2 * h, 2 * co), device=dev, dtype=torch.bfloat16)
        dup = Act((torch.randn((N, 2 * h, 2 * h, co), device=dev) * 0.5).to(torch.bfloat16), 0, co)
        dx = ops.dense(N, h, h, ci, dt, dev)
        w = torch.randn((ci, co, 2, 2), device=dev) * (ci * 4) ** -0.5
        bias = torch.zeros(co, device=dev)
        w2, w3 = ops.pack_weights(w, 2, dt), ops.pack_weights(w, 3, dt)
        for _ in range(2):     # clocks up before the first timed launch
            ops.conv_fwd(x, w2, bias, Act(cat, co, co), 1, 1, ci, 4 * co, epi=1)
        t_f = timeit(lambda: ops.conv_fwd(x, w2, bias, Act(cat, co, co), 1, 1, ci, 4 * co, epi=1), a.iters)
        t_d = timeit(lambda: ops.convT_dgrad(dup, w3, ci, dx), a.iters)
        t_w = timeit(lambda: ops.convT_wgrad(x, dup, ws), a.iters)
        gb = (x.buf.numel() + N * 4 * h * h * co) * 2 / 1e9
        tot += t_f + t_d + t_w
        print(f"{name} {ci:5d}->{co:4d} @{h:3d}^2  fwd {t_f:6.3f} ms {gb / t_f:6.2f} TB/s | dgrad {t_d:6.3f} ms {gb / t_d:6.2f} TB/s | "
              f"wgrad {t_w:6.3f} ms  -> {t_f + t_d + t_w:6.3f}", flush=True)
    print(f"total {tot:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
