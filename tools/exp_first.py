#!/usr/bin/env python3
"""Timing of the 1-channel first conv (forward + weight gradient) at the benchmark shape (A/B via UNET_HIP_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
N, H, C = 32, 512, 64
x = torch.rand(N, H, H, 1, device=dev)
w = torch.randn(C, 1, 3, 3, device=dev) * 0.3
b = torch.zeros(C, device=dev)
z = ops.dense(N, H, H, C, ops.BF16, dev)
dz = ops.Act((torch.randn((N, H, H, C), device=dev) * 0.5).bfloat16(), 0, C)
ws = ops.Workspace(dev)
for rep in range(3):
    tf = timeit(lambda: ops.conv_first_fwd(x, w, b, z, 3, 3, stats=True))
    tw = timeit(lambda: ops.conv_first_wgrad(dz, x, 3, 3, ws))
    print(f"first conv fwd {tf*1e3:6.1f} us   wgrad {tw*1e3:6.1f} us", flush=True)
