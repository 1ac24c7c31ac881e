"""Fused optimizer tail of step_batch (/root/reference/utils/train_step.py:56-63):

    scaler.unscale_(opt); clip_grad_norm_(model.parameters(), grad_clip); scaler.step(opt)

with the AdamW that /root/reference/utils/builders.py:143-146 builds.  `FusedAdamW` IS a
`torch.optim.AdamW` (same constructor, param_groups, state and state_dict layout, LR schedulers and step
hooks work unchanged); only `step()` differs: one multi-tensor HIP kernel per 40 tensors updates
parameter, exp_avg and exp_avg_sq in a single pass (28 B/parameter instead of torch's ~10 foreach
passes), and `clip_and_step()` folds clip_grad_norm_ into it: one pass for the global L2 norm, the
clip coefficient stays on the device (no host sync) and is applied while the update reads the gradient.

Anything the kernels do not cover (amsgrad, maximize, capturable, tensor lr, non-fp32 / CPU / sparse /
non-contiguous tensors) takes torch's own path, so behaviour never silently differs.
"""
from typing import Iterable, Optional

import torch

from . import ops


class FusedAdamW(torch.optim.AdamW):
    def __init__(self, params, *args, **kwargs):
        super().__init__(params, *args, **kwargs)
        self._mt_ws = {}            # device -> (partials fp64, norm_out fp32[2])
        self._pending_clip = None
        self._clip_norm = None
        self._vetoable = []         # parameters whose step count the last fused step() advanced under a device-side veto

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _tensor_ok(t: torch.Tensor) -> bool:
        return t.is_cuda and t.dtype == torch.float32 and not t.is_sparse and t.is_contiguous()

    @staticmethod
    def _group_ok(g) -> bool:
        return not (g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or g.get("fused")
                    or isinstance(g["lr"], torch.Tensor))

    def _ws(self, device, nblocks):
        ws = self._mt_ws.get(device)
        if ws is None or ws[0].numel() < nblocks:
            ws = (torch.empty(max(nblocks, 1024), dtype=torch.float64, device=device),
                  torch.empty(2, dtype=torch.float32, device=device))
            self._mt_ws[device] = ws
        return ws

    def _init_states(self, params):
        """Adam's lazy state (torch.optim.Adam._init_group) for every parameter of the group that has a gradient and no
        state yet: the moments are views of ONE zeroed buffer per moment (each on a 256-byte boundary) instead of two
        zeros_like launches per parameter (164 fills in the first step of the 4-level U-Net)."""
        todo = [p for p in params if p.grad is not None and len(self.state[p]) == 0 and self._tensor_ok(p)]
        offs, total = [], 0
        for p in todo:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        flats = {}
        for p, off in zip(todo, offs):
            if p.device not in flats:
                flats[p.device] = (torch.zeros(total, dtype=torch.float32, device=p.device),
                                   torch.zeros(total, dtype=torch.float32, device=p.device))
            m, v = flats[p.device]
            st = self.state[p]
            st["step"] = torch.tensor(0.0, dtype=torch.get_default_dtype())
            st["exp_avg"] = m[off:off + p.numel()].view_as(p)
            st["exp_avg_sq"] = v[off:off + p.numel()].view_as(p)
        for p in params:                                    # anything the fused path will refuse anyway: torch's own layout
            if p.grad is not None and len(self.state[p]) == 0:
                st = self.state[p]
                st["step"] = torch.tensor(0.0, dtype=torch.get_default_dtype())
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _collect(self):
        """-> [(group, step, params, grads, exp_avg, exp_avg_sq)] with the state initialised as
        torch.optim.Adam._init_group does, or None when some tensor needs torch's own path."""
        work = []
        device = None
        for group in self.param_groups:
            if not self._group_ok(group):
                return None
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (self._tensor_ok(p) and self._tensor_ok(p.grad)):
                    return None
                if device is None:
                    device = p.device
                elif p.device != device:
                    return None
                st = self.state[p]
                if len(st) == 0:
                    self._init_states(group["params"])
                if not (self._tensor_ok(st["exp_avg"]) and self._tensor_ok(st["exp_avg_sq"])):
                    return None
                step_t = st["step"]
                if not isinstance(step_t, torch.Tensor) or step_t.is_cuda:
                    return None
                by_step.setdefault(int(step_t.item()), []).append(p)
            for step, plist in by_step.items():
                work.append((group, step, plist))
        return work

    # ------------------------------------------------------------------ API
    def clip_and_step(self, parameters: Iterable[torch.Tensor], max_norm: float, grad_scale: float = 1.0,
                      veto: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """clip_grad_norm_(parameters, max_norm) followed by step(), without a host sync.
        Returns the total gradient norm (0-dim device tensor), as clip_grad_norm_ does.
        grad_scale: the .grad tensors hold 1/grad_scale times the gradients (data parallel hands over the all-reduced
        SUM with grad_scale = 1/world); norm, clipping and update are those of grad_scale*grad, and the clipped
        gradients written back are the averaged ones.
        veto: one fp32 device element; if it is > 0 when the update kernel runs, parameters, moments and gradients are left
        untouched (step_batch passes the loss kernel's illegal-label counter; only the fused path honours it)."""
        if isinstance(parameters, torch.Tensor):
            parameters = [parameters]
        self._pending_clip = ([p for p in parameters if p.grad is not None], float(max_norm), float(grad_scale))
        self._clip_norm = None
        self._veto = veto
        try:
            self.step()
        finally:
            self._pending_clip = None
            self._veto = None
        return self._clip_norm

    def undo_vetoed_step(self):
        """The veto element was > 0 when the last clip_and_step() ran, i.e. its update kernels did nothing: take back the
        host-side step counts they were launched with.  The caller (step_batch) learns the flag at its host sync."""
        for p in self._vetoable:
            self.state[p]["step"] -= 1
        self._vetoable = []

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        clip = self._pending_clip
        self._pending_clip = None
        work = self._collect()
        clip_ok = clip is None or all(self._tensor_ok(p.grad) for p in clip[0])
        if clip is not None and clip[2] != 1.0 and (work is None or not clip_ok or len({p.grad.device for p in clip[0]}) > 1
                                                    or {id(p) for p in clip[0]} != {id(p) for _, _, pl in (work or []) for p in pl}):
            for p in clip[0]:                           # a path below that cannot fold the scale: apply it first
                p.grad.mul_(clip[2])
            clip = (clip[0], clip[1], 1.0)
        if work is None or not clip_ok:
            if clip is not None and clip[0]:
                self._clip_norm = torch.nn.utils.clip_grad_norm_(clip[0], clip[1])
            super().step()
            return loss
        norm_out = None
        if clip is not None and clip[0]:
            grads = [p.grad for p in clip[0]]
            dev = grads[0].device
            if any(g.device != dev for g in grads):
                self._clip_norm = torch.nn.utils.clip_grad_norm_(clip[0], clip[1])
            else:
                partials, norm_out = self._ws(dev, ops.mt_blocks([g.numel() for g in grads]))
                ops.grad_norm(grads, clip[1], partials, norm_out, clip[2])
                self._clip_norm = norm_out[0].clone()
                stepped = {id(p) for _, _, plist in work for p in plist}
                for p in clip[0]:                       # clipped but not stepped by this optimizer
                    if id(p) not in stepped:
                        p.grad.mul_(norm_out[1])
                if work and work[0][2][0].device != dev:
                    for _, _, plist in work:            # norm lives on another device than the parameters
                        for p in plist:
                            p.grad.mul_(norm_out[1].to(p.device))
                    norm_out = None
        veto = getattr(self, "_veto", None)
        self._vetoable = []
        for group, step, plist in work:
            beta1, beta2 = group["betas"]
            for p in plist:
                self.state[p]["step"] += 1
            if veto is not None:
                self._vetoable += plist
            ops.adamw_step(plist, [p.grad for p in plist], [self.state[p]["exp_avg"] for p in plist],
                           [self.state[p]["exp_avg_sq"] for p in plist], norm_out, norm_out is not None,
                           group["lr"], beta1, beta2, group["eps"], group["weight_decay"], step + 1,
                           veto=veto)
            # the kernels write through raw pointers: tell autograd (and the engines' packed-weight caches)
            touched = plist + [self.state[p]["exp_avg"] for p in plist] + [self.state[p]["exp_avg_sq"] for p in plist]
            if norm_out is not None:
                touched += [p.grad for p in plist]
            torch.autograd.graph.increment_version(touched)
        return loss
