#!/usr/bin/env python3
"""Timing of the Cin = 1 first convolution with a 5x5 / 3x3-generic kernel (deployed RepUNet's first layer) at 16 x 1024^2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
from conv_microbench import timeit
dev = torch.device("cuda:0")
N, H, C = 16, 1024, 64
x = torch.rand(N, H, H, 1, device=dev)
z = ops.dense(N, H, H, C, ops.BF16, dev)
for k in (5, 3):
    w = torch.randn(C, 1, k, k, device=dev) * 0.2
    b = torch.zeros(C, device=dev)
    t = timeit(lambda: ops.conv_first_fwd(x, w, b, z, k, k, stats=False), 10)
    print(f"first conv {k}x{k} 1->{C} @ {N}x{H}^2: {t:.3f} ms  ({z.buf.numel() * 2 / t / 1e9:.2f} TB/s written)", flush=True)
