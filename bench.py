#!/usr/bin/env python3
"""Headline benchmark: UNet train step (step_batch: zero_grad -> fwd -> Dice+CE -> bwd -> clip -> AdamW)
on synthetic 512x512 CT slices, bf16 compute, per-GPU batch 32 (BASELINE.json configs[1]); with
--gpus N > 1 BASELINE.json configs[3]: global batch 256 under bucketed RCCL gradient all-reduce, i.e. 128 / 64 / 32
images per rank at N = 2 / 4 / 8 (--batch overrides the per-rank batch).

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


PMC_TRAFFIC = os.path.join("profiles", "r03_pmc_traffic.json")


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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=0,
                    help="per-GPU batch; default 32 on one GPU (configs[1]), 256 // N on N > 1 GPUs (configs[3]: global batch 256)")
    ap.add_argument("--global-batch", type=int, default=256, help="global batch the N > 1 default per-rank batch is derived from")
    ap.add_argument("--no-single-gpu-ref", action="store_true",
                    help="N > 1: skip the un-wrapped reference steps each rank times before the data-parallel run")
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
    if args.batch <= 0:
        args.batch = 32 if world == 1 else max(1, args.global_batch // world)
    if use_dist:
        # HIP's default of 4 hardware queues per process lets torch's RCCL stream land on the compute stream's queue, where
        # the all-reduces run in order with the kernels instead of under them; must be set before the
        # first HIP call of the process.  The pool's boxes EXPORT the default (4), so a setdefault would do nothing: raise
        # anything below 8 to 8, leave a larger setting of the operator's alone
        try:
            hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
        except ValueError:
            hwq = 4
        if hwq < 8:
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
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

    from unet_amd.utils.builders import get_optim          # the reference's builder API (utils/builders.py:128-146)
    torch.manual_seed(42)
    model = REGISTRY["unet"](in_channels=1, n_classes=2, features=[64, 128, 256, 512]).to(dev).train()
    engine = model._engine
    scaler = torch.amp.GradScaler(enabled=False)

    def make_trainer(m):
        opt = get_optim("adamw", [p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
        return types.SimpleNamespace(
            optimizer=opt, amp=(args.dtype == "bf16"), model=m, grad_clip=1.0, cbs=[], current_epoch=1,
            logger=logging.getLogger("bench"),
            loss_fn=get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255))

    def batch_of(n):
        g = torch.Generator().manual_seed(42 + rank)
        return (torch.rand(n, 1, args.size, args.size, generator=g).to(dev),
                (torch.rand(n, args.size, args.size, generator=g) > 0.95).long().to(dev))

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: what ONE GPU does without the wrapper, measured here on every rank at once (no collective in flight), at this
    # run's per-rank batch and at the 1-GPU configuration's batch of 32: the two denominators of SURVEY 8(d) C4's efficiency
    # definitions.  Reported next to `value`; the driver computes its own from the per-N lines.
    single_ref = {}
    if world > 1 and not args.no_single_gpu_ref:
        for nb in sorted({args.batch, 32}):
            tr = make_trainer(model)
            xb, yb = batch_of(nb)
            for i in range(2):
                step_batch(tr, xb, yb, scaler, batch_idx=i + 1)
            sync()
            t0 = time.perf_counter()
            for i in range(3):
                step_batch(tr, xb, yb, scaler, batch_idx=i + 1)
            torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            single_ref[nb] = 3 * nb / float(t)              # images/s of the slowest rank running alone
            del tr, xb, yb
        torch.cuda.empty_cache()

    if use_dist:
        model = DataParallel(model, device_ids=[local_rank])
    trainer = make_trainer(model)
    x, y = batch_of(args.batch)

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
            # the instrumented step runs the weight gradients on the compute stream again (product default: a side stream),
            # so that a launch's event pair brackets that kernel alone and not the time it queued behind the other stream
            engine.overlap_wgrad = False
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
                       "global_batch": args.batch * world, "per_rank_batch": args.batch, "image_size": args.size,
                       "parallelism": f"dp{world}", "final_loss": last[2]["main_loss"]},
        }
        if use_dist:
            out["config"]["rccl_world_size"] = dist.get_world_size()
            out["config"]["GPU_MAX_HW_QUEUES"] = os.environ.get("GPU_MAX_HW_QUEUES")
        if world > 1:
            # per-rank work is 256/N images: fixed total work from N = 2 on ("strong"); against the N = 1 line (32 images)
            # it is neither, hence the two measured denominators below
            out["scaling"] = "strong"
        if single_ref:
            out["single_gpu_ref"] = {
                "images_per_s_same_per_rank_batch": single_ref[args.batch], "images_per_s_batch32": single_ref[32],
                "efficiency_same_per_rank_batch": value / (world * single_ref[args.batch]),
                "efficiency_vs_1gpu_batch32": value / (world * single_ref[32]),
                "note": "un-wrapped step_batch timed on every rank at once before the wrap (3 steps after 2 warm-up, slowest rank)"}
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
