#!/usr/bin/env python3
"""Timing of the fused first / last layer passes at the benchmark shape (32 x 512^2 x 64, bf16); A/B via UNET_HIP_LIB."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
from exp_first import timeit
dev = torch.device("cuda:0")
N, H, C, K = 32, 512, 64, 2
x = torch.rand(N, H, H, 1, device=dev)
w1 = torch.randn(C, 1, 3, 3, device=dev) * 0.3
b1 = torch.zeros(C, device=dev)
st = ops.BNState(torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5)
y = ops.dense(N, H, H, C, ops.BF16, dev)
z = ops.Act((torch.randn((N, H, H, C), device=dev) * 0.5).bfloat16(), 0, C)
dz = ops.dense(N, H, H, C, ops.BF16, dev)
wk = torch.randn(K, C, device=dev) * 0.1
bk = torch.zeros(K, device=dev)
dl = torch.randn(N, K, H, H, device=dev) * 1e-3
ws = ops.Workspace(dev)
for rep in range(2):
    t = [timeit(lambda: ops.first_stats_moments(x, w1, b1)),
         timeit(lambda: ops.first_bn_act(x, w1, b1, st, y)),
         timeit(lambda: ops.first_bn_relu_bwd(z, x, w1, b1, st, ws)),
         timeit(lambda: ops.affine_act_out(z, st, wk, bk)),
         timeit(lambda: ops.bn_relu_bwd_out(dl, wk, z, dz, st, ws))]
    print("first: moments-stats %6.1f  bn_act %6.1f  bwd %6.1f us | last: act_out %6.1f  bwd_out %6.1f us" % tuple(v * 1e3 for v in t), flush=True)
