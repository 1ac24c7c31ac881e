#!/usr/bin/env python3
"""Two (or more) data-parallel ranks on ONE GPU over gloo: checks the whole multi-rank path with the real kernels.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \\
        tools/dist_rehearsal.py

Launched the way the driver launches bench.py (the launcher never touches the GPU).  Checks, per compute dtype:
  * parameters and buffers are identical on every rank after the wrap;
  * after forward + loss + backward + finish_grad_sync every p.grad equals the average of the per-rank gradients
    computed WITHOUT the wrapper from the same weights and the same per-rank inputs;
  * the arrival-order buckets were actually used (more than one all-reduce);
  * after three full step_batch calls the parameters are still identical on every rank and the loss is finite;
  * a batch with an illegal label on ONE rank (the reference raises inside loss_fn on that rank, before backward,
    /root/reference/losses/segmentation.py:8): every rank raises, no rank's parameters, moments or Adam step count move,
    and the next good step leaves the ranks identical again.
Collected by pytest as tests/test_gpu_parallel.py::test_two_ranks_one_gpu_rehearsal.
"""
import copy
import logging
import os
import sys
import types

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def batch(rank, n, size, dev):
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.rand(n, 1, size, size, generator=g).to(dev)
    y = (torch.rand(n, size, size, generator=g) > 0.9).long().to(dev)
    return x, y


def same_everywhere(tensors, world):
    flat = torch.cat([t.detach().reshape(-1).double() for t in tensors]).cpu()
    parts = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat)
    return all(torch.equal(parts[0], p) for p in parts)


class _Done:
    def wait(self):
        return True


def stage_collectives_through_host():
    """gloo's device-tensor path hangs on this ROCm build: run the collectives on host copies (synchronously).  The
    wrapper's logic (bucketing, arrival order, averaging, copy-back) is what this rehearsal checks, not the transport."""
    real_bcast, real_ar = dist.broadcast, dist.all_reduce

    def bcast(t, src=0, group=None, async_op=False):
        if not t.is_cuda:
            return real_bcast(t, src=src, group=group, async_op=async_op)
        h = t.cpu()
        real_bcast(h, src=src, group=group)
        t.copy_(h)
        return _Done() if async_op else None

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not t.is_cuda:
            return real_ar(t, op=op, group=group, async_op=async_op)
        h = t.cpu()
        real_ar(h, op=op, group=group)
        t.copy_(h)
        return _Done() if async_op else None

    dist.broadcast, dist.all_reduce = bcast, all_reduce


def say(rank, msg):
    print(f"[rank {rank}] {msg}", flush=True)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if "--device-collectives" not in sys.argv:
        stage_collectives_through_host()
    say(rank, "process group up")
    from unet_amd.losses import get_loss
    from unet_amd.models import REGISTRY
    from unet_amd.parallel import DataParallel
    from unet_amd.utils.builders import get_optim
    from unet_amd.utils.train_step import step_batch
    loss_fn = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)
    n, size = 4, 64
    ok = True
    for dtype, tol in (("fp32", 2e-5), ("bf16", 1e-3)):
        torch.manual_seed(7 + rank)                      # different init per rank: the wrap must fix that
        net = REGISTRY["unet"](in_channels=1, n_classes=2, features=[16, 32, 64, 128]).to(dev).train()
        net.compute_dtype = dtype
        say(rank, f"{dtype}: model built")
        dp = DataParallel(net, bucket_cap_mb=0.5)
        say(rank, f"{dtype}: wrapped")
        same0 = same_everywhere(list(net.parameters()) + [b.float() for b in net.buffers()], world)
        plain = copy.deepcopy(net)
        plain._grad_hook = None
        # reference: per-rank gradients without the wrapper, averaged
        want = None
        for r in range(world):
            ref = copy.deepcopy(plain)
            xr, yr = batch(r, n, size, dev)
            loss_fn(ref(xr), yr).backward()
            gs = {k: p.grad.clone() for k, p in ref.named_parameters()}
            want = gs if want is None else {k: want[k] + gs[k] for k in gs}
        want = {k: v / world for k, v in want.items()}
        say(rank, f"{dtype}: per-rank reference gradients done")
        x, y = batch(rank, n, size, dev)
        before = dp.reducer.buckets_launched
        loss_fn(dp(x), y).backward()
        dp.finish_grad_sync()
        nb = dp.reducer.buckets_launched - before
        worst = 0.0
        for k, p in net.named_parameters():
            err = float((p.grad - want[k]).abs().max()) / (float(want[k].abs().max()) + 1e-12)
            if float(want[k].abs().max()) > 1e-6:        # conv biases in front of a train-mode BN have noise-only grads
                worst = max(worst, err)
        grads_same = same_everywhere([p.grad for p in net.parameters()], world)
        say(rank, f"{dtype}: data-parallel backward done")
        # three full steps
        opt = get_optim("adamw", list(dp.parameters()), lr=1e-3, weight_decay=1e-4)
        trainer = types.SimpleNamespace(optimizer=opt, amp=(dtype == "bf16"), model=dp, grad_clip=1.0, cbs=[], current_epoch=1,
                                        logger=logging.getLogger("rehearsal"), loss_fn=loss_fn)
        net.compute_dtype = None if dtype == "bf16" else "fp32"
        scaler = torch.amp.GradScaler(enabled=False)
        last = None
        for i in range(3):
            last = step_batch(trainer, x, y, scaler, batch_idx=i + 1)
        same3 = same_everywhere(list(net.parameters()), world)
        loss = float(last[2]["main_loss"])
        # illegal label on the last rank only
        snap = [p.detach().clone() for p in net.parameters()]
        steps_before = [float(opt.state[p]["step"]) for p in opt.param_groups[0]["params"]]
        ybad = y.clone()
        if rank == world - 1:
            ybad[0, 3, 5] = 2
        raised = False
        try:
            step_batch(trainer, x, ybad, scaler, batch_idx=9)
        except RuntimeError as e:
            raised = "Class values must be smaller than num_classes" in str(e)
        torch.cuda.synchronize()
        untouched = all(torch.equal(a, b.detach()) for a, b in zip(snap, net.parameters()))
        steps_same = steps_before == [float(opt.state[p]["step"]) for p in opt.param_groups[0]["params"]]
        step_batch(trainer, x, y, scaler, batch_idx=10)
        same_after = same_everywhere(list(net.parameters()), world)
        moved = not all(torch.equal(a, b.detach()) for a, b in zip(snap, net.parameters()))
        veto_ok = raised and untouched and steps_same and same_after and moved
        say(rank, f"{dtype}: illegal label on rank {world - 1}: raised here={raised} parameters untouched={untouched} "
                  f"Adam step counts unchanged={steps_same}; next step: ranks identical={same_after}")
        good = same0 and grads_same and same3 and nb > 1 and worst < tol and loss == loss and veto_ok
        ok = ok and good
        if rank == 0:
            print(f"{dtype}: params identical after wrap={same0} buckets={nb} worst rel grad error vs per-rank average={worst:.2e} "
                  f"grads identical on all ranks={grads_same} params identical after 3 steps={same3} loss={loss:.4f} -> "
                  f"{'OK' if good else 'FAIL'}", flush=True)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST REHEARSAL", "OK" if float(flag) == 1.0 else "FAILED", f"(world {world})", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if float(flag) == 1.0 else 1)


if __name__ == "__main__":
    main()
