#!/usr/bin/env python3
"""Times the bf16 5x5 convolution (deployed RepUNet, kernel_sizes [3,5]) with bias + ReLU at the ensemble's shapes
(64 x 1 x 1024^2 input): A/B via UNET_HIP_LIB."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
from conv_microbench import timeit
dev = torch.device("cuda:0")
tot = 0.0
for name, cin, cout, h, n in (("l0", 64, 64, 1024, 16), ("l1", 128, 128, 512, 16), ("l2", 256, 256, 256, 16), ("l3", 512, 512, 128, 16),
                              ("bott", 1024, 1024, 64, 16), ("up1", 512, 256, 256, 16)):
    x = ops.Act((torch.randn((n, h, h, cin), device=dev) * 0.5).bfloat16(), 0, cin)
    y = ops.dense(n, h, h, cout, ops.BF16, dev)
    w = ops.pack_weights(torch.randn((cout, cin, 5, 5), device=dev) * (cin * 25) ** -0.5, 0, ops.BF16)
    b = torch.zeros(cout, device=dev)
    for _ in range(2):
        ops.conv_fwd(x, w, b, y, 5, 5, cin, cout, relu=True)
    t = timeit(lambda: ops.conv_fwd(x, w, b, y, 5, 5, cin, cout, relu=True), 10)
    fl = 2.0 * n * h * h * cin * cout * 25
    tot += t
    print(f"{name:5s} {cin:5d}->{cout:5d} @{h:5d}^2 x{n}: {t:7.3f} ms {fl / t / 1e9:7.1f} TFLOP/s", flush=True)
print(f"total {tot:.3f} ms")
