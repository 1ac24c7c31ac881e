#!/usr/bin/env python3
"""Headline benchmark: UNet train step (step_batch: zero_grad -> fwd -> Dice+CE -> bwd -> clip -> AdamW)
on synthetic 512x512 CT slices, bf16 compute, per-GPU batch 32 (BASELINE.json configs[1]); with
--gpus N > 1 the same per-rank batch under bucketed RCCL gradient all-reduce (weak scaling).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the bf16 3x3 implicit-GEMM conv,
forward + dgrad launches): algorithmic FLOPs of its launches / their durations measured with HIP
events on the launch stream inside the timed region (the launches of the last timed step).  `cpu_baseline` is the CPU oracle (torch fp32
restatement of the reference, pinned to golden vectors) timed on a bounded sample on this host.
"""
import argparse
import json
import logging
import os
import sys
import time
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
TRAIN_GFLOP_PER_IMAGE_512 = 1154.0


class ConvTimer:
    """HIP-event timing of every bf16 3x3 implicit-GEMM launch on the launch stream: forward and dgrad
    (conv3x3_v2_kernel, `records`) and the weight gradient (conv_wgrad2_kernel + its split-K reduce, `wgrad_records`)."""

    def __init__(self):
        self.records = []          # (flops, start, end)
        self.wgrad_records = []

    def install(self):
        from unet_amd import ops
        orig = ops.conv_fwd
        orig_w = ops.conv_wgrad
        recs = self.records
        wrecs = self.wgrad_records

        def timed_w(dz, x, kh, kw, *a, **k):
            if kh == 3 and kw == 3 and x.dt == ops.BF16 and x.C >= 64:
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                r = orig_w(dz, x, kh, kw, *a, **k)
                e.record()
                wrecs.append((2.0 * x.M * 9 * x.C * dz.C, s, e))
                return r
            return orig_w(dz, x, kh, kw, *a, **k)

        def timed(x, wpacked, bias, y, kh, kw, Cin, Cout, *a, **k):
            if kh == 3 and kw == 3 and x.dt == ops.BF16:
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                r = orig(x, wpacked, bias, y, kh, kw, Cin, Cout, *a, **k)
                e.record()
                recs.append((2.0 * x.M * 9 * Cin * Cout, s, e))
                return r
            return orig(x, wpacked, bias, y, kh, kw, Cin, Cout, *a, **k)

        ops.conv_fwd = timed
        ops.conv_wgrad = timed_w
        self._ops, self._orig, self._orig_w = ops, orig, orig_w

    def remove(self):
        self._ops.conv_fwd = self._orig
        self._ops.conv_wgrad = self._orig_w

    @staticmethod
    def _sum(records):
        return sum(r[0] for r in records), sum(r[1].elapsed_time(r[2]) for r in records), len(records)

    def summary(self):
        return self._sum(self.records)

    def wgrad_summary(self):
        return self._sum(self.wgrad_records)


PMC_TRAFFIC = os.path.join("profiles", "r02_pmc_traffic.json")


def lib_build_id():
    """First 12 hex digits of the SHA-256 of the HIP library this process loads (the build the numbers belong to)."""
    import hashlib
    from unet_amd import _lib
    try:
        with open(_lib.LIB_PATH, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:12]
    except OSError:
        return None


def pmc_traffic():
    """(HBM-side bytes per launch of the dominant kernel, build id of the library they were counted on) from the
    committed PMC passes (rocprofv3 --pmc cannot run inside this process): tools/pmc_bench.sh -> PMC_TRAFFIC;
    (None, None) if absent."""
    try:
        with open(os.path.join(ROOT, PMC_TRAFFIC)) as f:
            j = json.load(f)
        return float(j["hbm_bytes_per_launch"]), j.get("build")
    except (OSError, KeyError, ValueError):
        return None, None


def cpu_baseline(sample_images=4, size=512, iters=3):
    """Oracle fwd + loss + bwd on the host cores (fp32), bounded sample of the same workload."""
    from oracle import loss_ref, unet_ref
    from unet_amd.models import UNet
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    m = UNet(in_channels=1, n_classes=2, features=[64, 128, 256, 512])
    st = {k: v.clone() for k, v in m.state_dict().items()}
    names = [k for k in st if "running_" not in k and "num_batches" not in k]
    for k in names:
        st[k].requires_grad_(True)
    g = torch.Generator().manual_seed(42)
    x = torch.rand(sample_images, 1, size, size, generator=g)
    y = (torch.rand(sample_images, size, size, generator=g) > 0.95).long()
    times = []
    for it in range(iters + 1):
        for k in names:
            st[k].grad = None
        t0 = time.perf_counter()
        loss = loss_ref.ce_dice(unet_ref.unet_forward(x, st, True, {}), y, 1.0, 1.0, 255)
        loss.backward()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {"value": sample_images / best, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch CPU fp32 restatement) fwd+loss+bwd on {sample_images}x1x{size}x{size}, "
                      f"best of {iters} after 1 warm-up, {best:.2f} s/iter"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-conv-timer", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the data-parallel path (process group, DataParallel wrap, bucketed all-reduce) "
                         "even with one rank")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    use_dist = world > 1 or args.force_dist
    if args.force_dist:
        os.environ["UNET_AMD_DP_FORCE_COLLECTIVES"] = "1"      # the one-rank rehearsal still launches the RCCL collectives
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from unet_amd.losses import get_loss
    from unet_amd.models import REGISTRY
    from unet_amd.parallel import DataParallel
    from unet_amd.utils.train_step import step_batch

    torch.manual_seed(42)
    model = REGISTRY["unet"](in_channels=1, n_classes=2, features=[64, 128, 256, 512]).to(dev).train()
    if use_dist:
        model = DataParallel(model, device_ids=[local_rank])
    from unet_amd.utils.builders import get_optim          # the reference's builder API (utils/builders.py:128-146)
    opt = get_optim("adamw", [p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
    trainer = types.SimpleNamespace(
        optimizer=opt, amp=(args.dtype == "bf16"), model=model, grad_clip=1.0, cbs=[], current_epoch=1,
        logger=logging.getLogger("bench"),
        loss_fn=get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255))
    scaler = torch.amp.GradScaler(enabled=False)

    g = torch.Generator().manual_seed(42 + rank)
    x = torch.rand(args.batch, 1, args.size, args.size, generator=g).to(dev)
    y = (torch.rand(args.batch, args.size, args.size, generator=g) > 0.95).long().to(dev)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step_batch(trainer, x, y, scaler, batch_idx=i + 1)
    # The per-launch HIP events of the roofline measurement cost ~0.1 ms each in lost launch overlap (3-4 ms over a
    # step's 35 conv launches), so they bracket the conv launches of the LAST timed step only: still inside the timed
    # region, but `value` is perturbed by 1/K of that instead of all of it.
    timer = ConvTimer()
    sync()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        if i == args.steps - 1 and not args.no_conv_timer:
            timer.install()
        last = step_batch(trainer, x, y, scaler, batch_idx=i + 1)
    sync()
    dt = time.perf_counter() - t0
    if not args.no_conv_timer:
        timer.remove()
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)

    if rank == 0:
        images = args.batch * world * args.steps
        value = images / dt
        out = {
            "metric": "train images/sec (512x512 CT, fwd+bwd)", "value": value, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"UNet features [64,128,256,512] full step_batch (fwd+Dice/CE+bwd+clip+AdamW), "
                                   f"{args.batch}x1x{args.size}x{args.size} per GPU",
                       "global_batch": args.batch * world, "image_size": args.size,
                       "parallelism": f"dp{world}", "final_loss": last[2]["main_loss"]},
        }
        if not args.no_conv_timer and timer.records:
            fl, ms, n = timer.summary()
            ach = fl / (ms * 1e-3) / 1e12
            traffic, traffic_build = pmc_traffic()
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
                               "traffic_source": PMC_TRAFFIC, "traffic_build": traffic_build,
                               "kernel": "conv3x3_v2_kernel<WC,GEN> (bf16 3x3 fwd + dgrad launches)",
                               "launches": n, "avg_launch_ms": ms / n, "avg_launch_gflop": fl / n / 1e9}
            wfl, wms, wn = timer.wgrad_summary()
            if wn:
                wach = wfl / (wms * 1e-3) / 1e12
                out["roofline_wgrad"] = {"bound": "mfma", "achieved": wach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": wach / MFMA_BF16_PEAK_TFLOPS,
                                         "kernel": "conv_wgrad2_kernel<3,3,...> + split-K reduce (bf16 3x3 weight gradient launches)",
                                         "launches": wn, "avg_launch_ms": wms / wn}
            out["build"] = lib_build_id()
            out["train_step_tflops"] = TRAIN_GFLOP_PER_IMAGE_512 * (args.size / 512.0) ** 2 * value / 1e3
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
