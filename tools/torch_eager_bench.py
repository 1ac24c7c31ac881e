#!/usr/bin/env python3
"""What stock PyTorch-ROCm (MIOpen / rocBLAS / ATen kernels) does on the SAME box with the SAME workload as bench.py:
the number the reference itself would get on an MI355X, since the reference is plain torch.nn
(/root/reference/models/unet.py:41-62, utils/train_step.py:7-79).  Comparison only: nothing in unet_amd calls this.

The network is evaluated with torch.nn.functional on the parameters of an unet_amd UNet (same shapes and init), under
torch.autocast (fp16 + GradScaler as the reference's `amp: true`, or bf16) or in fp32, NCHW or channels_last.

  python tools/torch_eager_bench.py --dtype bf16 --channels-last --steps 10 --warmup 3
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def double_conv(x, blk, training):
    seq = blk.double_conv
    for conv, bn in ((seq[0], seq[1]), (seq[3], seq[4])):
        x = F.conv2d(x, conv.weight, conv.bias, padding=1)
        x = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
        x = F.relu(x)
    return x


def eager_forward(m, x):
    skips = []
    for blk in m.downs:
        x = double_conv(x, blk, m.training)
        skips.append(x)
        x = F.max_pool2d(x, 2, 2)
    x = double_conv(x, m.bottleneck, m.training)
    for i in range(0, len(m.ups), 2):
        up = m.ups[i]
        x = F.conv_transpose2d(x, up.weight, up.bias, stride=2)
        x = torch.cat((skips[-1 - i // 2], x), 1)
        x = double_conv(x, m.ups[i + 1], m.training)
    return F.conv2d(x, m.final_conv.weight, m.final_conv.bias)


def ce_dice(logits, y, eps=1e-6):
    ce = F.cross_entropy(logits, y, ignore_index=255)
    p = torch.softmax(logits, 1)
    oh = F.one_hot(y, p.shape[1]).permute(0, 3, 1, 2).float()
    inter = (p * oh).sum((2, 3))
    union = p.sum((2, 3)) + oh.sum((2, 3))
    return ce + 1.0 - ((2 * inter + eps) / (union + eps)).mean()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--benchmark", action="store_true", help="torch.backends.cudnn.benchmark (MIOpen exhaustive find)")
    a = ap.parse_args()

    from unet_amd.models import REGISTRY
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = a.benchmark
    torch.manual_seed(0)
    m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[64, 128, 256, 512]).to(dev)
    m.train()
    if a.channels_last:
        for p in m.parameters():
            if p.dim() == 4:
                p.data = p.data.contiguous(memory_format=torch.channels_last)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
    scaler = torch.amp.GradScaler("cuda", enabled=a.dtype == "fp16")
    g = torch.Generator().manual_seed(1)
    x = torch.rand(a.batch, 1, a.size, a.size, generator=g).to(dev)
    y = (torch.rand(a.batch, a.size, a.size, generator=g) > 0.95).long().to(dev)
    if a.channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    amp_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[a.dtype]

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=amp_dtype, enabled=amp_dtype is not None):
            logits = eager_forward(m, x)
            loss = ce_dice(logits.float(), y)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        scaler.step(opt)
        scaler.update()
        return loss

    import threading
    t0 = time.time()
    done = threading.Event()

    def heartbeat():                      # MIOpen's first find/compile of ~60 conv problems can take minutes
        while not done.wait(30):
            print(f"... {time.time() - t0:.0f}s", flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()
    for i in range(a.warmup):
        loss = step()
        torch.cuda.synchronize()
        print(f"warmup {i}: {time.time() - t0:.1f}s loss {float(loss):.4f}", flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    done.set()
    print(json.dumps({"what": "stock torch eager step (MIOpen)", "dtype": a.dtype, "channels_last": a.channels_last,
                      "miopen_benchmark": a.benchmark, "batch": a.batch, "size": a.size, "ms_per_step": round(dt * 1e3, 2),
                      "images_per_s": round(a.batch / dt, 1), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1),
                      "loss": float(loss)}), flush=True)


if __name__ == "__main__":
    main()
