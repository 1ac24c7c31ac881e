"""Batch-sharded data parallelism over RCCL (one process per GPU), replacing the reference's
DistributedDataParallel wrap (/root/reference/train.py:139-144).

  * parameters and buffers are broadcast from rank 0 at wrap time; BN buffers again before every
    training forward (DDP's broadcast_buffers=True default); BatchNorm statistics stay per rank
    (the reference leaves SyncBatchNorm commented out, train.py:142);
  * gradients are averaged with bucketed all-reduces launched from INSIDE the explicit backward
    schedule: the engine hands each layer's gradients over as soon as their kernels are enqueued
    (backward-arrival order: final_conv, ups.7 .. ups.0, bottleneck, downs.3 .. downs.0), a bucket is
    all-reduced asynchronously on the process group's stream as soon as it is full, and
    finish_grad_sync() (called by step_batch before clipping) waits and scatters the averages into
    p.grad.  xGMI is point-to-point, so buckets are large (default 32 MiB: 4 all-reduces per step).
"""
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


ALIGN = 64          # elements: 256 bytes for fp32 gradients


class BucketReducer:
    """Average named gradient tensors across the process group in arrival-order buckets."""

    def __init__(self, process_group=None, bucket_bytes: int = 32 << 20, comm_dtype: Optional[torch.dtype] = None):
        self.pg = process_group
        self.bucket_bytes = int(bucket_bytes)
        self.comm_dtype = comm_dtype
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # rehearsal on one GPU: issue the collectives even in a one-rank group (UNET_AMD_DP_FORCE_COLLECTIVES=1)
        self.collective = dist.is_initialized() and (self.world > 1 or os.environ.get("UNET_AMD_DP_FORCE_COLLECTIVES") == "1")
        self._pending: List = []          # (names, shapes, flat, work)
        self._cur_names: List[str] = []
        self._cur: List[torch.Tensor] = []
        self._cur_bytes = 0
        self.buckets_launched = 0
        self._flat_cache: Dict[tuple, torch.Tensor] = {}      # (bucket index in this backward, numel, dtype, device) -> flat buffer
        self._bucket_idx = 0
        self._open = False                # a backward is handing gradients over: first add() ... flush() (end of backward) / finish()
        self.views_installed = None       # set by DataParallel: are last step's bucket views still some p.grad?

    def add(self, names: List[str], tensors: List[torch.Tensor]):
        if not self._open:
            # First gradient of a new backward.  The flat buffers are rewritten in place, so (a) nothing of an earlier
            # backward may still be waiting in them (a backward that was never finished, e.g. an exception between
            # backward and finish_grad_sync: drop it, its keys would otherwise miss and allocate new flats every step),
            # and (b) no p.grad may still be a view of them: a second backward before the optimizer step would
            # overwrite the first micro-step's gradients instead of accumulating into them.
            if self._pending or self._cur:
                for *_, work in self._pending:
                    if work is not None:
                        work.wait()
                self._pending, self._cur_names, self._cur, self._cur_bytes = [], [], [], 0
            self._bucket_idx = 0
            if self.views_installed is not None and self.views_installed():
                raise RuntimeError(
                    "unet_amd.parallel: a second backward arrived while p.grad still holds the previous backward's bucket "
                    "views; call optimizer.zero_grad(set_to_none=True) between backward passes (gradient accumulation "
                    "across backward calls is not supported by the in-place buckets)")
            self._open = True
        for n, t in zip(names, tensors):
            if t is None:
                continue
            self._cur_names.append(n)
            self._cur.append(t)
            self._cur_bytes += t.numel() * t.element_size()
        if self._cur_bytes >= self.bucket_bytes:
            self._launch()

    def _launch(self):
        if not self._cur:
            return
        dtype = self.comm_dtype or self._cur[0].dtype
        # the bucket layout repeats every step (fixed arrival order), so the flat buffers are allocated once and the
        # gather writes into them; p.grad of the previous step (views of these buffers, as with DDP's
        # gradient_as_bucket_view) is dropped by zero_grad(set_to_none=True) before they are written again
        # every tensor starts on a 256-byte boundary of the bucket, as a fresh allocation would: the multi-tensor
        # norm / AdamW kernels pick their vector path by alignment, and the clip norm must not depend (in its last
        # bit) on whether p.grad is a bucket view or its own allocation
        offs, numel = [], 0
        for t in self._cur:
            offs.append(numel)
            numel += (t.numel() + ALIGN - 1) // ALIGN * ALIGN
        key = (self._bucket_idx, tuple(offs), numel, dtype, self._cur[0].device)
        flat = self._flat_cache.get(key)
        if flat is None:
            flat = self._flat_cache[key] = torch.zeros(numel, dtype=dtype, device=self._cur[0].device)
        views = [flat[o:o + t.numel()] for o, t in zip(offs, self._cur)]
        torch._foreach_copy_(views, [t.reshape(-1) for t in self._cur])
        self._bucket_idx += 1
        work = None
        if self.collective:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._pending.append((self._cur_names, [t.shape for t in self._cur], offs, flat, work))
        self._cur_names, self._cur, self._cur_bytes = [], [], 0
        self.buckets_launched += 1

    def flush(self):
        """All gradients of this backward pass have been handed over: launch the partial last bucket.  An add() after this
        belongs to a NEW backward."""
        self._launch()
        self._open = False

    def finish(self, average: bool = True) -> Dict[str, torch.Tensor]:
        """Wait for every bucket; returns {name: gradient view}: the average over ranks, or with average=False the SUM
        (the caller folds 1/world into its clip coefficient and saves the pass)."""
        self._launch()
        self._bucket_idx = 0
        self._open = False
        out: Dict[str, torch.Tensor] = {}
        for names, shapes, offs, flat, work in self._pending:
            if work is not None:
                work.wait()
            if self.world > 1 and average:
                flat.div_(self.world)
            for n, s, off in zip(names, shapes, offs):
                k = 1
                for d in s:
                    k *= d
                out[n] = flat[off:off + k].view(s)
        self._pending = []
        return out


class DataParallel(nn.Module):
    """DDP-shaped wrapper (`.module`, same forward) for unet_amd models."""

    def __init__(self, module: nn.Module, device_ids=None, output_device=None, process_group=None,
                 bucket_cap_mb: float = 32, broadcast_buffers: bool = True, comm_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.broadcast_buffers = broadcast_buffers
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        bucket_cap_mb = float(os.environ.get("UNET_AMD_DP_BUCKET_MB", bucket_cap_mb))       # experiments: 1e9 = one all-reduce after backward
        self.reducer = BucketReducer(process_group, int(bucket_cap_mb * (1 << 20)), comm_dtype)
        self._params = dict(module.named_parameters())
        module._grad_hook = self.reducer.add
        module._grad_flush = self.reducer.flush
        self.collective = self.reducer.collective
        self.reducer.views_installed = self._grad_views_installed
        self._installed: List = []          # [(parameter, bucket view)] of the last finish_grad_sync
        self._buf_flats: List = []          # [(flat tensor, [buffers that are views of it])] per dtype
        if self.collective:
            self._broadcast([p.data for p in module.parameters()])
            # the copies above went through .data (no version bump): drop every image the engine derived from the old
            # weights (a forward before wrapping would otherwise leave ranks > 0 computing with their pre-broadcast weights)
            torch.autograd.graph.increment_version(list(module.parameters()))
            eng = getattr(module, "_engine", None)
            if eng is not None:
                eng._packed.clear()
                eng._folded.clear()
            self._sync_buffers()

    def _src(self):
        return dist.get_global_rank(self.pg, 0) if self.pg is not None else 0

    def _broadcast(self, tensors: List[torch.Tensor]):
        """rank 0 -> all, coalesced per dtype into one flat message (wrap time only)."""
        by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
        for t in tensors:
            by_dtype.setdefault(t.dtype, []).append(t)
        for ts in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=self._src(), group=self.pg)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()

    # The BN buffers are re-broadcast before EVERY training forward (DDP's broadcast_buffers=True), so they are kept as
    # views of one flat tensor per dtype: the broadcast then runs in place on that tensor, with no gather copy before
    # and no scatter copies after it (54 buffers for the 4-level U-Net).
    def _flatten_buffers(self):
        self._buf_flats = []
        by_dtype: Dict[tuple, List[torch.Tensor]] = {}
        for b in self.module.buffers():
            if b is not None and b.is_contiguous():
                by_dtype.setdefault((b.dtype, b.device), []).append(b)
        for ts in by_dtype.values():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            off = 0
            for t in ts:
                t.data = flat[off:off + t.numel()].view(t.shape)
                off += t.numel()
            self._buf_flats.append((flat, ts))

    def _buffers_are_views(self) -> bool:
        """False after .to() / load_state_dict(assign=True) / new buffers: the module's buffers left the flat tensors."""
        n = 0
        for flat, ts in self._buf_flats:
            base, esz, off = flat.data_ptr(), flat.element_size(), 0
            for t in ts:
                if t.data_ptr() != base + off * esz or t.device != flat.device:
                    return False
                off += t.numel()
            n += len(ts)
        return n == sum(1 for b in self.module.buffers() if b is not None and b.is_contiguous())

    def _sync_buffers(self):
        if not self._buf_flats or not self._buffers_are_views():
            self._flatten_buffers()
        for flat, _ in self._buf_flats:
            dist.broadcast(flat, src=self._src(), group=self.pg)

    def forward(self, *args, **kwargs):
        if self.collective and self.broadcast_buffers and self.module.training:
            self._sync_buffers()
        return self.module(*args, **kwargs)

    def _grad_views_installed(self) -> bool:
        return any(p.grad is g for p, g in self._installed)

    def share_status(self, flag: torch.Tensor):
        """MAX over ranks, in place and asynchronously, of a small status tensor (step_batch: the loss kernel's illegal-label
        counter).  The fused optimizer tail reads it as its veto, so every rank must see the same value or one rank would skip
        an update the others apply to the same all-reduced gradients."""
        if self.collective:
            # four bytes: no point in carrying the handle through backward -- the compute stream waits for it here (a stream
            # dependency, not a host wait), so that everything queued after this call sees the reduced value
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.pg, async_op=True).wait()

    def finish_grad_sync(self, average: bool = True):
        """Called after loss.backward(): p.grad <- average over ranks.  average=False leaves the SUM in p.grad and
        returns the factor still owed (1/world); step_batch hands it to FusedAdamW.clip_and_step(grad_scale=...).
        p.grad becomes a view of a bucket that the NEXT backward rewrites in place: callers must drop it with
        zero_grad(set_to_none=True) (step_batch does) before that backward; a second backward without it raises."""
        avg = self.reducer.finish(average)
        self._installed = []
        for n, g in avg.items():
            p = self._params[n]
            if p.grad is None:
                continue
            if g.dtype == p.grad.dtype:
                p.grad = g                  # a contiguous view of the reduced bucket (DDP's gradient_as_bucket_view)
                self._installed.append((p, g))
            else:
                p.grad.copy_(g)             # communication dtype differs from the gradient's
        return 1.0 if (average or self.world <= 1) else 1.0 / self.world
