"""Times the fused first-layer kernels (csrc/edge_fused.hip, conv_edge.hip) at the bench shape 32 x 512^2 x 1 -> 64, bf16:
statistics pass, forward pass, one-pass backward.  python tools/first_layer_microbench.py [--iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, H, C, dt = a.batch, a.size, 64, ops.BF16
    g = torch.Generator().manual_seed(1)
    x = torch.rand(N, H, H, 1, generator=g).to(dev)
    w = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).to(dev)
    b = (torch.randn(C, generator=g) * 0.1).to(dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ws = ops.Workspace(dev)
    rows = ops.conv_first_stats(x, w, b, dt)
    st = ops.bn_finalize(rows, N * H * H, gamma, beta, None, None, None, 0.1, 1e-5, ws)
    ya = ops.Act(torch.zeros((N, H, H, 2 * C), dtype=torch.bfloat16, device=dev), 0, C)      # the skip half of the decoder concat buffer, as in the engine
    dy = ops.Act(torch.randn((N, H, H, C), device=dev).to(torch.bfloat16), 0, C)
    t_s = timeit(lambda: ops.conv_first_stats(x, w, b, dt), a.iters)
    t_f = timeit(lambda: ops.first_bn_act(x, w, b, st, ya), a.iters)
    t_b = timeit(lambda: ops.first_bn_relu_bwd(dy, x, w, b, st, ws), a.iters)
    print(f"first layer {N}x{H}x{H}x1->{C} bf16: statistics {t_s:.1f} us, forward {t_f:.1f} us, backward {t_b:.1f} us")


if __name__ == "__main__":
    main()
