#!/usr/bin/env python3
"""Can an HBM-bound streaming kernel and an MFMA-bound persistent kernel share the CUs productively?

The persistent conv / weight-gradient kernels of the training step own every CU (all 512 registers per SIMD lane, ~150 KB of
LDS), so the BatchNorm passes (HBM-bound, 21 % of the step) never overlap with matrix work.  This probe takes an MFMA-bound
kernel that leaves room on the CU -- the 1x3 instantiation of the LDS-DMA conv kernel: 512 threads, 161 registers = 2 waves
per SIMD, 190 registers per lane free -- and a streaming kernel that needs no LDS (affine_act, 42 registers), and runs them
alone and together on two streams:

    python tools/exp_corun.py [--prio]

If "together" is close to max(alone) the pairing is worth building into the backward schedule (weight gradients under the
BatchNorm passes); if it is close to the sum, the hardware serialises them or they starve each other.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prio", action="store_true", help="matrix kernel on a high-priority stream")
    ap.add_argument("--reps", type=int, default=12)
    a = ap.parse_args()
    from unet_amd import ops
    dev = torch.device("cuda:0")
    dt = ops.BF16
    # MFMA-bound: 1x3 conv 512 -> 512 on 32 x 64 x 64 (206 GFLOP, 34 MB in + out)
    N, H, W, C = 32, 64, 64, 512
    x = ops.Act(torch.randn(N, H, W, C, device=dev).to(torch.bfloat16), 0, C)
    y = ops.dense(N, H, W, C, dt, dev)
    wp = ops.pack_weights(torch.randn(C, C, 1, 3, device=dev) * 0.02, 0, dt)
    # HBM-bound: BatchNorm apply + ReLU on 32 x 512 x 512 x 64 (1.07 GB in, 1.07 GB out)
    z = ops.dense(32, 512, 512, 64, dt, dev)
    z.buf.normal_()
    o = ops.dense(32, 512, 512, 64, dt, dev)
    st = ops.BNState(torch.ones(64, device=dev), torch.zeros(64, device=dev))

    def matrix(n):
        for _ in range(n):
            ops.conv_fwd(x, wp, None, y, 1, 3, C, C)

    def stream(n):
        for _ in range(n):
            ops.affine_act(z, o, st, relu=True)

    sa = torch.cuda.Stream(dev, priority=-1 if a.prio else 0)
    sb = torch.cuda.Stream(dev)

    def timed(fa, fb):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    matrix(2); stream(2)
    nm, ns = a.reps * 3, a.reps
    for it in range(3):
        ta = timed(lambda: matrix(nm), None)
        tb = timed(None, lambda: stream(ns))
        tab = timed(lambda: matrix(nm), lambda: stream(ns))
        print(f"run {it}: matrix alone {ta:.2f} ms ({nm} launches), streaming alone {tb:.2f} ms ({ns} launches), together {tab:.2f} ms "
              f"(sum {ta + tb:.2f}, max {max(ta, tb):.2f})", flush=True)


if __name__ == "__main__":
    main()
