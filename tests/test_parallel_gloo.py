"""world_size-2 gloo tests (CPU) of the data-parallel host logic: arrival-order bucketing, averaging,
rank-0 broadcast, and batch sharding that matches the reference's DistributedSampler use."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from unet_amd.parallel import BucketReducer, DataParallel
        # --- bucketed averaging in arrival order
        red = BucketReducer(bucket_bytes=200)
        g = torch.Generator().manual_seed(100 + rank)
        names = [f"p{i}" for i in range(7)]
        shapes = [(3, 5), (40,), (2, 2, 2), (64,), (1,), (30, 2), (7,)]
        ts = [torch.randn(*s, generator=g) for s in shapes]
        for i in range(0, 7, 2):
            red.add(names[i:i + 2], ts[i:i + 2])
        avg = red.finish()
        want = []
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            want.append([torch.randn(*s, generator=gr) for s in shapes])
        ok = red.buckets_launched >= 3
        for i, n in enumerate(names):
            ref = sum(w[i] for w in want) / world
            ok = ok and torch.allclose(avg[n], ref, atol=1e-6) and avg[n].shape == ref.shape
        # --- wrapper: rank-0 broadcast of parameters and buffers, gradient averaging into p.grad
        torch.manual_seed(rank)              # different init per rank on purpose
        net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
        dp = DataParallel(net, bucket_cap_mb=1e-5)
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()] + [b.reshape(-1).float() for b in net.buffers()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        ok = ok and all(torch.equal(gathered[0], x) for x in gathered)
        x = torch.randn(6, 4, generator=torch.Generator().manual_seed(7 + rank))
        dp(x).square().mean().backward()
        local = {n: p.grad.clone() for n, p in net.named_parameters()}
        for n in reversed(list(local)):       # the engine would call the hook in backward order
            net._grad_hook([n], [local[n]])
        dp.finish_grad_sync()
        for n, p in net.named_parameters():
            parts = [torch.zeros_like(local[n]) for _ in range(world)]
            dist.all_gather(parts, local[n])
            ok = ok and torch.allclose(p.grad, sum(parts) / world, atol=1e-6)
        # --- in-place buckets: a second backward while p.grad still holds the first one's bucket views must raise
        #     (it would overwrite the first micro-step's gradients, not accumulate); zero_grad(set_to_none=True) clears it
        try:
            net._grad_hook(["0.weight"], [local["0.weight"]])
            ok = False
        except RuntimeError as e:
            ok = ok and "second backward" in str(e)
        net.zero_grad(set_to_none=True)
        # --- a backward that was never finished (exception between backward and finish_grad_sync) is dropped by the next
        #     one instead of piling up: same buckets, same cached flat buffers
        nflats = len(dp.reducer._flat_cache)
        for attempt in range(2):
            for n in reversed(list(local)):
                net._grad_hook([n], [local[n] * (attempt + 1)])
            net._grad_flush()
        for p in net.parameters():
            p.grad = torch.zeros_like(p)
        dp.finish_grad_sync()
        ok = ok and len(dp.reducer._flat_cache) == nflats
        for n, p in net.named_parameters():
            parts = [torch.zeros_like(local[n]) for _ in range(world)]
            dist.all_gather(parts, local[n])
            ok = ok and torch.allclose(p.grad, 2 * sum(parts) / world, atol=1e-6)
        net.zero_grad(set_to_none=True)
        # --- the status word (step_batch: the loss kernel's illegal-label counter) becomes the MAX over ranks, so every
        #     rank vetoes the same update
        st = torch.tensor([0.5 + rank, 0.0, 0.0, 3.0 if rank == world - 1 else 0.0])
        dp.share_status(st[3:4])
        dp.finish_grad_sync()
        ok = ok and float(st[3]) == 3.0 and float(st[0]) == 0.5 + rank
        # --- BN buffers: per-rank statistics after a training forward; rank 0's again at the next forward's sync,
        #     broadcast in place through the flat tensors the buffers are views of
        def flat_bufs():
            return torch.cat([b.reshape(-1).float() for b in net.buffers()])
        parts = [torch.zeros_like(flat_bufs()) for _ in range(world)]
        dist.all_gather(parts, flat_bufs())
        ok = ok and not torch.equal(parts[0], parts[1])          # different inputs -> different running statistics
        ok = ok and dp._buffers_are_views()
        dp._sync_buffers()                                        # what the next training forward does first
        ok = ok and torch.equal(flat_bufs(), parts[0])
        net[1].running_mean.data = net[1].running_mean.data.clone() + (1.0 if rank else 0.0)    # storage replaced (as .to() does)
        ok = ok and not dp._buffers_are_views()
        dp._sync_buffers()                                        # re-flattens, then broadcasts
        ok = ok and dp._buffers_are_views() and torch.equal(flat_bufs(), parts[0])
        # --- the flattened buffers still save and load like ordinary ones (torch.save keeps the shared storage once)
        import io
        f = io.BytesIO()
        torch.save(dp.state_dict(), f)
        f.seek(0)
        back = torch.load(f, weights_only=True)
        ok = ok and all(torch.equal(back[k], v) for k, v in dp.state_dict().items())
        dp.load_state_dict(back)
        ok = ok and dp._buffers_are_views()                       # load_state_dict copies in place
        # --- sharding: rank r takes samples r::W of the epoch permutation, like DistributedSampler
        from torch.utils.data import DistributedSampler
        ds = list(range(10))
        s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=42)
        s.set_epoch(3)
        mine = torch.tensor(list(s))
        allidx = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allidx, mine)
        ok = ok and sorted(torch.cat(allidx).tolist()) == sorted(ds)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
