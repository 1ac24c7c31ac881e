#!/usr/bin/env python3
"""One-rank RCCL self-check of the data-parallel wrapper with the real HIP engine (run by tests/test_gpu_parallel.py in
a fresh process: the "nccl" process group is created before anything touches the GPU).

UNET_AMD_DP_FORCE_COLLECTIVES=1 makes the wrapper issue its collectives even in a one-rank group, so the parameter /
buffer broadcasts and every bucket's asynchronous all-reduce really run on RCCL's stream.  With one rank the sum over
ranks is the identity, so everything must equal the un-wrapped model BIT FOR BIT:
  * gradients after forward + Dice/CE + backward + finish_grad_sync, fp32 and bf16 kernels, >= 4 arrival-order buckets;
  * parameters and buffers after one full step_batch call (fused clip + AdamW tail, the 1/world factor folded into
    the clip coefficient, buffers re-broadcast before the training forward) equal the un-wrapped model's; two more
    steps keep tracking it;
  * a forward BEFORE wrapping must not leave stale packed weights behind (the wrap drops the engine's caches).
Reference: /root/reference/train.py:139-144 (DDP wrap), utils/train_step.py:7-79.
"""
import copy
import logging
import os
import sys
import types

os.environ["UNET_AMD_DP_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29731")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist


def main():
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from unet_amd.losses import get_loss
    from unet_amd.models import REGISTRY
    from unet_amd.parallel import DataParallel
    from unet_amd.utils.builders import get_optim
    from unet_amd.utils.train_step import step_batch
    loss_fn = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(4, 1, 64, 64, generator=g).to(dev)
    y = (torch.rand(4, 64, 64, generator=g) > 0.9).long().to(dev)
    ok = True
    for dtype in ("fp32", "bf16"):
        torch.manual_seed(7)
        net = REGISTRY["unet"](in_channels=1, n_classes=2, features=[16, 32, 64, 128]).to(dev).train()
        net.compute_dtype = dtype
        with torch.no_grad():
            net(x)                                     # a forward before the wrap fills the packed-weight cache
        plain = copy.deepcopy(net)
        plain._grad_hook = plain._grad_flush = None
        dp = DataParallel(net, bucket_cap_mb=0.5)
        assert dp.collective, "the forced collectives are off"
        assert not net._engine._packed, "the wrap must drop packed images of the pre-broadcast weights"
        loss_fn(plain(x), y).backward()
        before = dp.reducer.buckets_launched
        loss_fn(dp(x), y).backward()
        dp.finish_grad_sync()
        nb = dp.reducer.buckets_launched - before
        torch.cuda.synchronize()
        same_g = all(torch.equal(p.grad, q.grad) for p, q in zip(net.parameters(), plain.parameters()))

        def trainer(model):
            opt = get_optim("adamw", [p for p in model.parameters()], lr=1e-3, weight_decay=1e-4)
            return types.SimpleNamespace(optimizer=opt, amp=False, model=model, grad_clip=1.0, cbs=[], current_epoch=1,
                                         logger=logging.getLogger("dp"), loss_fn=loss_fn)

        def gap(ma, mb):
            """largest |a-b| / max|b| over parameters and float buffers (inf when an integer buffer differs)"""
            worst = 0.0
            for a, b in zip(list(ma.parameters()) + list(ma.buffers()), list(mb.parameters()) + list(mb.buffers())):
                a, b = a.detach(), b.detach()
                if a.is_floating_point():
                    worst = max(worst, float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)))
                elif not torch.equal(a, b):
                    return float("inf")
            return worst

        ta, tb = trainer(dp), trainer(plain)
        scaler = torch.amp.GradScaler(enabled=False)
        gaps, losses = [], []
        for i in range(3):
            la = step_batch(ta, x, y, scaler, i)[2]["main_loss"]
            lb = step_batch(tb, x, y, scaler, i)[2]["main_loss"]
            torch.cuda.synchronize()
            gaps.append(gap(net, plain))
            losses.append((la, lb))
        # One full step (clip + AdamW, buffers re-broadcast) must leave the same parameters and buffers: the gradients
        # are bit-equal and so are the bucket views' alignments, so the fused tail sees the same numbers.  Later steps
        # are only required to track: any last-bit difference in a BatchNorm batch sum (fp64 atomics, order varies) is
        # amplified by ReLU decisions flipping (tests/test_gpu_realwidth.py), with or without the wrapper.
        tol3 = 1e-3 if dtype == "fp32" else 2e-2
        good = (same_g and nb >= 4 and gaps[0] <= 1e-6 and losses[0][0] == losses[0][1]
                and all(abs(a - b) <= tol3 * abs(b) for a, b in losses))
        ok = ok and good
        print(f"{dtype}: buckets={nb} grads bit-equal={same_g}; wrapped vs plain after steps 1/2/3: "
              + " / ".join(f"{v:.1e}" for v in gaps) + f"; losses {losses} -> {'OK' if good else 'FAIL'}", flush=True)
    dist.destroy_process_group()
    print("DP SELFCHECK", "OK" if ok else "FAILED", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
