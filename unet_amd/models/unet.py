"""UNet with the reference's constructor, registration order and state_dict, computed by HIP kernels.

Mirrors /root/reference/models/unet.py:5-67.  The nn.Conv2d / nn.BatchNorm2d / nn.ConvTranspose2d
children are parameter containers only (same names, shapes, init order => same state_dict and, under
the same torch.manual_seed, the same initial weights as the reference); forward runs
unet_amd.engine.UNetEngine on the GPU.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from ..engine import UNetEngine


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2 -- parameter layout of models/unet.py:5-19."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
        )

    def forward(self, x):
        raise ops.UNetHipError("DoubleConv is executed by UNet.forward on the HIP engine; it has no standalone path")


def resolve_dtype(module) -> int:
    """bf16 under torch.autocast (the reference's `amp: true`), else fp32 (`amp: false`);
    `module.compute_dtype` or UNET_AMD_DTYPE ('bf16' / 'fp32') override."""
    forced = getattr(module, "compute_dtype", None) or os.environ.get("UNET_AMD_DTYPE")
    if forced:
        forced = str(forced).lower()
        if forced in ("bf16", "bfloat16"):
            return ops.BF16
        if forced in ("fp32", "float32", "f32"):
            return ops.F32
        raise ops.UNetHipError(f"unknown compute dtype {forced!r}")
    return ops.BF16 if torch.is_autocast_enabled() else ops.F32


class SlotCache:
    """named_parameters() / named_buffers() without walking the module tree every forward: the (owning dict, key,
    qualified name) slots are collected once, the tensors are fetched from the dicts at each call, so replaced
    tensors (.to(), load_state_dict(assign=True)) are seen; only a change of STRUCTURE needs invalidate_slots()
    (switch_to_deploy / make_deploy_skeleton call it)."""

    def _slots(self):
        c = self.__dict__.get("_slot_cache")
        if c is None:
            ps, bs = [], []
            for mname, mod in self.named_modules():
                pre = mname + "." if mname else ""
                for k in mod._parameters:
                    ps.append((mod._parameters, k, pre + k))
                for k in mod._buffers:
                    bs.append((mod._buffers, k, pre + k))
            c = self.__dict__["_slot_cache"] = (ps, bs)
        return c

    def invalidate_slots(self):
        self.__dict__.pop("_slot_cache", None)

    def named_params_fast(self):
        return [(n, d[k]) for d, k, n in self._slots()[0] if d.get(k) is not None]

    def named_buffers_fast(self):
        return {n: d[k] for d, k, n in self._slots()[1] if d.get(k) is not None}


def check_input_needs_no_grad(x):
    """The engine's backward stops at the first conv (the reference's dataset tensors never require grad).  A caller that
    asks for dL/dx (saliency, adversarial examples, a learnable stem in front) must not get a silent None."""
    if torch.is_grad_enabled() and isinstance(x, torch.Tensor) and x.requires_grad:
        raise ops.UNetHipError("unet_amd does not compute the gradient with respect to the input image "
                               "(x.requires_grad is True): detach x, or run the model under torch.no_grad()")


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, dt, need_grad, x, *params):
        names = module._cur_names
        P = dict(zip(names, params))
        B = module.named_buffers_fast()
        logits, sv = module._engine.forward(x, P, B, dt, module.training, need_grad)
        ctx.module, ctx.sv, ctx.names = module, sv, names
        ctx.P = P if need_grad else None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        module, sv = ctx.module, ctx.sv
        if sv is None:
            raise ops.UNetHipError("backward through a forward that did not save activations")
        hook = getattr(module, "_grad_hook", None)
        grads = module._engine.backward(dlogits, ctx.P, sv, hook)
        ctx.sv = None
        flush = getattr(module, "_grad_flush", None)
        if flush is not None:
            flush()          # data parallel: the last partial bucket leaves now, and drops its references to the gradients,
                             # so autograd takes them over below instead of cloning each one
        out = []
        for n in ctx.names:
            p = ctx.P[n]
            out.append(grads.get(n) if p.requires_grad else None)
        return (None, None, None, None, *out)


class UNet(SlotCache, nn.Module):
    def __init__(self, in_channels=1, n_classes=3, features=[64, 128, 256, 512]):
        super().__init__()
        features = list(features)
        self.ups = nn.ModuleList()
        self.downs = nn.ModuleList()
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        cin = in_channels
        for feature in features:
            self.downs.append(DoubleConv(cin, feature))
            cin = feature
        for feature in reversed(features):
            self.ups.append(nn.ConvTranspose2d(feature * 2, feature, kernel_size=2, stride=2))
            self.ups.append(DoubleConv(feature * 2, feature))
        self.bottleneck = DoubleConv(features[-1], features[-1] * 2)
        self.final_conv = nn.Conv2d(features[0], n_classes, kernel_size=1)

        for f in features:
            if f % 8:
                raise ValueError(f"unet_amd needs every feature width to be a multiple of 8 (got {features})")
        if in_channels > 4:
            raise ValueError("unet_amd's first-layer kernel serves in_channels <= 4")
        if not 1 <= n_classes <= 8:
            raise ValueError("unet_amd serves 1..8 classes")
        self._engine = UNetEngine(features, in_channels, n_classes)
        self.compute_dtype = None
        self._grad_hook = None

    def forward(self, x, current_epoch=None):
        check_input_needs_no_grad(x)
        named = self.named_params_fast()
        params = [p for _, p in named]
        self._cur_names = [n for n, _ in named]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _UNetFn.apply(self, resolve_dtype(self), need_grad, x, *params)

    def prepack_weights(self):
        """Refresh the packed compute-dtype weight images now (step_batch calls this right after the optimizer step,
        so the work is queued behind it instead of sitting in front of the next forward)."""
        dt = getattr(self._engine, "_last_dt", None)
        if dt is not None:
            self._engine.prepack(dict(self.named_params_fast()), dt)


def build_unet(**kwargs):
    return UNet(**kwargs)
