#!/usr/bin/env python3
"""Which part of the forward launch costs more than the dgrad launch of the same shape: times conv_fwd with the bias and the
statistics switched on and off (same input, same weights)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
from conv_microbench import timeit

dev = torch.device("cuda:0")
for cin, cout, h in ((256, 256, 128), (1024, 1024, 32), (128, 128, 256)):
    N, dt = 32, ops.BF16
    x = ops.Act((torch.randn((N, h, h, cin), device=dev) * 0.5).to(torch.bfloat16), 0, cin)
    y = ops.dense(N, h, h, cout, dt, dev)
    w = torch.randn((cout, cin, 3, 3), device=dev) * (cin * 9) ** -0.5
    bias = torch.zeros(cout, device=dev)
    for mode in (0, 1):
        wp = ops.pack_weights(w, mode, dt)
        for b in (None, bias):
            for st in (False, True):
                t = timeit(lambda: ops.conv_fwd(x, wp, b, y, 3, 3, cin, cout, stats=st), 20)
                print(f"{cin}->{cout}@{h} pack={mode} bias={'y' if b is not None else 'n'} stats={'y' if st else 'n'}: {t:.3f} ms", flush=True)
