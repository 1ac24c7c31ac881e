#!/usr/bin/env python3
"""Timing of the BN(+ReLU) backward kernels at the benchmark's shapes (A/B of kernel variants via UNET_HIP_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
N = 32
for C, H in ((64, 512), (128, 256), (256, 128), (512, 64)):
    g = torch.Generator(device=dev).manual_seed(1)
    z = ops.Act((torch.randn((N, H, H, C), generator=g, device=dev) * 0.5).bfloat16(), 0, C)
    dy = ops.Act((torch.randn((N, H, H, C), generator=g, device=dev) * 0.5).bfloat16(), 0, C)
    dz = ops.dense(N, H, H, C, ops.BF16, dev)
    ws = ops.Workspace(dev)
    zz = z.buf.double()
    rows = torch.stack([zz.sum((1, 2)), (zz * zz).sum((1, 2))], -1).float().contiguous()
    st = ops.bn_finalize(rows, N * H * H, torch.ones(C, device=dev), torch.zeros(C, device=dev), None, None, None, 0.1, 1e-5, ws)
    t = timeit(lambda: ops.bn_relu_bwd(dy, z, dz, st, ws, relu=True))
    gb = 5 * N * H * H * C * 2 / 1e9
    print(f"C={C:4d} H={H:4d}  bn_relu_bwd (reduce+finalize+apply) {t*1e3:7.1f} us  {gb / t:7.1f} GB/s-equivalent x1e3", flush=True)
