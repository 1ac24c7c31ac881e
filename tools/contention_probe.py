#!/usr/bin/env python3
"""What does a collective that HOLDS compute units cost the overlapped backward?  

One GPU cannot run a multi-rank RCCL all-reduce, so this probe emulates its footprint: the data-parallel wrapper runs as in
`bench.py --force-dist` (one-rank process group, buckets launched from inside the backward schedule), and every bucket's
all-reduce is replaced by `cu_thief_kernel` (tools/cu_thief.hip) on a side stream: W workgroups that each exclude a conv /
wgrad workgroup from their CU for T microseconds.  The step time is reported for a sweep of (W, T); W = 0 is the baseline.

    hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/cu_thief.hip -o tools/libcu_thief.so     (on the CPU box)
    python3 tools/contention_probe.py [--steps 8]
"""
import argparse
import ctypes
import logging
import os
import sys
import time
import types

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


class _Work:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--sweep", default="0:0,16:250,32:250,32:1000,64:250,64:1000,32:4000,0:0")
    ap.add_argument("--side-index", type=int, default=0, help="use the (k+1)-th stream created (HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues round robin; two streams on one queue serialise)")
    ap.add_argument("--side-priority", type=int, default=-1, help="-1: high-priority side stream, which gets a hardware queue of its own; 0: a pool stream, which on this image lands on the compute stream's hardware queue, so the two serialise")
    args = ap.parse_args()

    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcu_thief.so"))
    lib.cu_thief_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

    os.environ["UNET_AMD_DP_FORCE_COLLECTIVES"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)

    import unet_amd.parallel as par
    from unet_amd.losses import get_loss
    from unet_amd.models import REGISTRY
    from unet_amd.utils.builders import get_optim
    from unet_amd.utils.train_step import step_batch

    torch.manual_seed(42)
    model = par.DataParallel(REGISTRY["unet"](in_channels=1, n_classes=2, features=[64, 128, 256, 512]).to(dev).train(),
                             device_ids=[0])
    opt = get_optim("adamw", [p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
    trainer = types.SimpleNamespace(optimizer=opt, amp=True, model=model, grad_clip=1.0, cbs=[], current_epoch=1,
                                    logger=logging.getLogger("probe"),
                                    loss_fn=get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255))
    scaler = torch.amp.GradScaler(enabled=False)
    g = torch.Generator().manual_seed(42)
    x = torch.rand(args.batch, 1, args.size, args.size, generator=g).to(dev)
    y = (torch.rand(args.batch, args.size, args.size, generator=g) > 0.95).long().to(dev)

    _spare = [torch.cuda.Stream(priority=args.side_priority) for _ in range(args.side_index)]
    side = torch.cuda.Stream(priority=args.side_priority)
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    cfg = {"wgs": 0, "us": 0}
    real_all_reduce = dist.all_reduce

    def fake_all_reduce(flat, op=None, group=None, async_op=False):
        # the bucket is already the one-rank sum; stand in for the collective's footprint on the machine
        if cfg["wgs"] <= 0:
            return real_all_reduce(flat, op=op, group=group, async_op=async_op)
        side.wait_stream(torch.cuda.current_stream())
        rc = lib.cu_thief_launch(cfg["wgs"], cfg["us"], sink.data_ptr(), side.cuda_stream)
        assert rc == 0, rc
        ev = torch.cuda.Event()
        ev.record(side)
        return _Work(ev)

    par.dist.all_reduce = fake_all_reduce          # parallel.py calls dist.all_reduce through its own module reference

    print(f"{'thief WGs':>10} {'hold us':>8} {'buckets':>8} {'ms/step':>9} {'delta ms':>9}", flush=True)
    base = None
    for item in args.sweep.split(","):
        w, t = (int(v) for v in item.split(":"))
        cfg["wgs"], cfg["us"] = w, t
        for i in range(args.warmup):
            step_batch(trainer, x, y, scaler, batch_idx=i + 1)
        torch.cuda.synchronize()
        n0 = model.reducer.buckets_launched
        t0 = time.perf_counter()
        for i in range(args.steps):
            step_batch(trainer, x, y, scaler, batch_idx=i + 1)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        nb = (model.reducer.buckets_launched - n0) / args.steps
        if base is None:
            base = ms
        print(f"{w:>10d} {t:>8d} {nb:>8.1f} {ms:>9.2f} {ms - base:>+9.2f}", flush=True)
    par.dist.all_reduce = real_all_reduce
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
