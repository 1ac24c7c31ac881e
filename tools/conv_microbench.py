#!/usr/bin/env python3
"""Times every MFMA conv launch of the C2 workload (UNet 32x1x512x512, bf16) layer by layer:
forward, dgrad and wgrad, reporting TFLOP/s against the 2.5 PFLOP/s dense bf16 peak."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd import ops  # noqa: E402

LAYERS = [  # name, Cin, Cout, H (square), from SURVEY 8d
    ("down0.b", 64, 64, 512), ("down1.a", 64, 128, 256), ("down1.b", 128, 128, 256),
    ("down2.a", 128, 256, 128), ("down2.b", 256, 256, 128), ("down3.a", 256, 512, 64),
    ("down3.b", 512, 512, 64), ("bott.a", 512, 1024, 32), ("bott.b", 1024, 1024, 32),
    ("up0.a", 1024, 512, 64), ("up1.a", 512, 256, 128), ("up2.a", 256, 128, 256), ("up3.a", 128, 64, 512),
]


def timeit(fn, iters):
    for _ in range(4):          # the first launches of a process run at a lower clock: warm up before timing
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    ap.add_argument("--per-tile-stats", action="store_true", help="forward with one statistics row per tile (flags bit 2): the request-wave kernel on 64-channel tiles")
    a = ap.parse_args()
    dt = ops.BF16 if a.dtype == "bf16" else ops.F32
    dev = torch.device("cuda:0")
    ws = ops.Workspace(dev)
    out = []
    for name, cin, cout, h in LAYERS:
        if a.only and a.only not in name:
            continue
        N = a.batch
        x = ops.Act((torch.randn((N, h, h, cin), device=dev) * 0.5).to(ops.torch_dtype(dt)), 0, cin)
        dz = ops.Act((torch.randn((N, h, h, cout), device=dev) * 0.5).to(ops.torch_dtype(dt)), 0, cout)
        y = ops.dense(N, h, h, cout, dt, dev)
        dx = ops.dense(N, h, h, cin, dt, dev)
        w = torch.randn((cout, cin, 3, 3), device=dev) * (cin * 9) ** -0.5
        wf, wd = ops.pack_weights(w, 0, dt), ops.pack_weights(w, 1, dt)
        bias = torch.zeros(cout, device=dev)
        flops = 2.0 * N * h * h * cin * cout * 9
        t_f = timeit(lambda: ops.conv_fwd(x, wf, bias, y, 3, 3, cin, cout, stats=True, per_image_stats=a.per_tile_stats), a.iters)
        t_d = timeit(lambda: ops.conv_fwd(dz, wd, None, dx, 3, 3, cout, cin), a.iters)
        t_w = timeit(lambda: ops.conv_wgrad(dz, x, 3, 3, ws), a.iters)
        rec = dict(layer=name, cin=cin, cout=cout, hw=h, gflop=flops / 1e9,
                   fwd_ms=t_f, dgrad_ms=t_d, wgrad_ms=t_w,
                   fwd_tf=flops / t_f / 1e9, dgrad_tf=flops / t_d / 1e9, wgrad_tf=flops / t_w / 1e9)
        out.append(rec)
        print(f"{name:8s} {cin:5d}->{cout:5d} @{h:4d}^2  fwd {t_f:7.3f} ms {rec['fwd_tf']:7.1f} TF | "
              f"dgrad {t_d:7.3f} ms {rec['dgrad_tf']:7.1f} TF | wgrad {t_w:7.3f} ms {rec['wgrad_tf']:7.1f} TF", flush=True)
        del x, dz, y, dx
    tot = sum(r["fwd_ms"] + r["dgrad_ms"] + r["wgrad_ms"] for r in out)
    fl = sum(3 * r["gflop"] for r in out)
    print(f"total {tot:.2f} ms for {fl/1e3:.2f} TFLOP -> {fl / tot:.1f} TFLOP/s average", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
