"""ReparamUNet with the reference's constructors, registration order and state_dict, on HIP kernels.

Mirrors /root/reference/models/reparm_unet.py:10-234.  Children are parameter containers; forward runs
unet_amd.rep_engine.RepUNetEngine.

The reference's switch_to_deploy() raises AttributeError as shipped, because getBN returns GroupNorm
(:17-25) while mergeBN reads running_var (:28-31).  Decision: the constructor
keeps its signature and gains an optional `norm` kwarg: "group" (default, as shipped) or "batch" (the
author's commented-out BatchNorm2d(momentum=0.01) variant, :14-15).  The fold is defined, HIP-executed
and bit-exact-tested for "batch"; for "group" it raises a clear error (GroupNorm statistics depend on
the input, there is nothing to fold).
"""
import torch
import torch.nn as nn

from .. import ops
from ..rep_engine import RepUNetEngine
from .unet import SlotCache, _UNetFn, check_input_needs_no_grad, resolve_dtype


def getConv2D(in_channels, out_channels, kernel_size=(3, 3), stride=1, padding=0, groups=1, bias=True):
    return nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, stride=stride,
                     padding=padding, groups=groups, bias=bias)


def getBN(channels, eps=1e-5, momentum=None, affine=True, norm="group"):
    if norm == "batch":
        return nn.BatchNorm2d(num_features=channels, eps=eps, momentum=0.01 if momentum is None else momentum, affine=affine)
    return nn.GroupNorm(num_groups=min(8, channels), num_channels=channels, eps=eps, affine=affine)


def _bn_tuple(bn):
    return (bn.weight.detach().contiguous(), bn.bias.detach().contiguous(), bn.running_mean.contiguous(), bn.running_var.contiguous())


class AsymmConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, depthwise=False, norm="group"):
        super().__init__()
        if depthwise or stride != 1:
            raise NotImplementedError("unet_amd AsymmConvBlock: stride 1, groups 1 only (what ReparamUNet uses)")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.padding, self.stride, self.depthwise = kernel_size, padding, stride, depthwise
        self.norm = norm
        self.mergedFlag = False
        self.mergedConv = None
        k = kernel_size
        self.convVe1 = getConv2D(in_channels, out_channels, (k, 1), stride, [padding, 0], bias=False)
        self.bnVe1 = getBN(out_channels, norm=norm)
        self.convHo1 = getConv2D(in_channels, out_channels, (1, k), stride, [0, padding], bias=False)
        self.bnHo1 = getBN(out_channels, norm=norm)
        self.convSq = getConv2D(in_channels, out_channels, (k, k), stride, [padding, padding], bias=False)
        self.bnSq = getBN(out_channels, norm=norm)

    def forward(self, x):
        raise ops.UNetHipError("AsymmConvBlock is executed by ReparamUNet.forward on the HIP engine")

    def mergeAsyKernels(self, sqrt="ieee"):
        if self.mergedFlag:
            return
        if self.norm != "batch":
            raise ops.UNetHipError("the fold needs running statistics: build ReparamUNet(..., norm='batch'); "
                                   "GroupNorm (the shipped default) is input-dependent and cannot be folded")
        w, b = ops.fold_asymm(self.kernel_size, self.convSq.weight.detach().contiguous(), self.convVe1.weight.detach().contiguous(),
                              self.convHo1.weight.detach().contiguous(), _bn_tuple(self.bnSq), _bn_tuple(self.bnVe1),
                              _bn_tuple(self.bnHo1), self.bnSq.eps, sqrt=sqrt)
        k = self.kernel_size
        self.mergedConv = getConv2D(self.in_channels, self.out_channels, (k, k), self.stride, [self.padding, self.padding], bias=True)
        self.mergedConv.weight.data = w
        self.mergedConv.bias.data = b
        self.mergedFlag = True
        for attr in ["convVe1", "bnVe1", "convHo1", "bnHo1", "convSq", "bnSq"]:
            if hasattr(self, attr):
                delattr(self, attr)


class MultiConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_sizes, stride, depthwise=False, norm="group"):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_sizes = sorted(kernel_sizes)
        self.stride, self.depthwise = stride, depthwise
        self.mergedFlag = False
        self.mergedConv = None
        self.conv_list = nn.ModuleList()
        for kernel in self.kernel_sizes:
            self.conv_list.append(AsymmConvBlock(in_channels, out_channels, kernel, stride, kernel // 2, depthwise, norm=norm))

    def forward(self, x):
        raise ops.UNetHipError("MultiConvBlock is executed by ReparamUNet.forward on the HIP engine")

    def mergeMulKernels(self, sqrt="ieee"):
        if self.mergedFlag:
            return
        for block in self.conv_list:
            block.mergeAsyKernels(sqrt)
        target_w = self.conv_list[-1].mergedConv.weight.data.clone()
        target_b = self.conv_list[-1].mergedConv.bias.data.clone()
        for i in range(len(self.conv_list) - 1):
            ops.fold_embed_add(target_w, self.conv_list[i].mergedConv.weight.data, target_b, self.conv_list[i].mergedConv.bias.data)
        max_k = self.kernel_sizes[-1]
        self.mergedConv = getConv2D(self.in_channels, self.out_channels, (max_k, max_k), self.stride, [max_k // 2, max_k // 2], bias=True)
        self.mergedConv.weight.data = target_w
        self.mergedConv.bias.data = target_b
        self.mergedFlag = True
        if hasattr(self, "conv_list"):
            delattr(self, "conv_list")


class ReparamDoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_sizes=[3], norm="group"):
        super().__init__()
        self.reparam_conv = nn.Sequential(
            MultiConvBlock(in_channels, out_channels, kernel_sizes, stride=1, depthwise=False, norm=norm),
            nn.ReLU(inplace=True),
            MultiConvBlock(out_channels, out_channels, kernel_sizes, stride=1, depthwise=False, norm=norm),
            nn.ReLU(inplace=True),
        )

    def forward(self, x):
        raise ops.UNetHipError("ReparamDoubleConv is executed by ReparamUNet.forward on the HIP engine")


class ReparamUNet(SlotCache, nn.Module):
    def __init__(self, in_channels=1, n_classes=3, features=[64, 128, 256, 512], kernel_sizes=[3], norm="group"):
        super().__init__()
        if norm not in ("group", "batch"):
            raise ValueError("norm must be 'group' (as shipped) or 'batch' (foldable variant)")
        features = list(features)
        self.ups = nn.ModuleList()
        self.downs = nn.ModuleList()
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.kernel_sizes = kernel_sizes
        self.norm = norm
        cin = in_channels
        for feature in features:
            self.downs.append(ReparamDoubleConv(cin, feature, kernel_sizes, norm))
            cin = feature
        for feature in reversed(features):
            self.ups.append(nn.ConvTranspose2d(feature * 2, feature, kernel_size=2, stride=2))
            self.ups.append(ReparamDoubleConv(feature * 2, feature, kernel_sizes, norm))
        self.bottleneck = ReparamDoubleConv(features[-1], features[-1] * 2, kernel_sizes, norm)
        self.final_conv = nn.Conv2d(features[0], n_classes, kernel_size=1)
        for f in features:
            if f % 8:
                raise ValueError(f"unet_amd needs every feature width to be a multiple of 8 (got {features})")
        if in_channels > 4 or not 1 <= n_classes <= 8:
            raise ValueError("unet_amd serves in_channels <= 4 and 1..8 classes")
        self._engine = RepUNetEngine(features, in_channels, n_classes, kernel_sizes, norm)
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

    def make_deploy_skeleton(self):
        """Turn an un-deployed model into the deploy-mode STRUCTURE (one zero-initialised `mergedConv` per
        MultiConvBlock, branches removed) without folding, so that a deployed checkpoint -- keys
        `...reparam_conv.{0,2}.mergedConv.{weight,bias}`, the reference's layout after switch_to_deploy
        (models/reparm_unet.py:139-144) -- can be loaded.  load_state_dict() calls it when it sees such keys."""
        for module in list(self.modules()):
            if isinstance(module, MultiConvBlock) and not module.mergedFlag:
                max_k = module.kernel_sizes[-1]
                ref = next(module.parameters())
                module.mergedConv = getConv2D(module.in_channels, module.out_channels, (max_k, max_k), module.stride,
                                              [max_k // 2, max_k // 2], bias=True).to(device=ref.device, dtype=ref.dtype)
                module.mergedFlag = True
                delattr(module, "conv_list")
        self._engine._packed.clear()
        self.invalidate_slots()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        deployed_keys = any(".mergedConv." in k for k in state_dict)
        has_branches = any(isinstance(m, MultiConvBlock) and not m.mergedFlag for m in self.modules())
        if deployed_keys and has_branches:
            self.make_deploy_skeleton()
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def switch_to_deploy(self, sqrt="ieee"):
        """Fuse every MultiConvBlock into one Conv2d(k_max) on the GPU (HIP fold kernels); returns self.
        The reference's mergeBN (models/reparm_unet.py:29) takes std = sqrt(running_var + eps) with torch's sqrt on whatever
        device the BN buffers live on; every other step is one exactly rounded fp32 operation in the reference's order.
        sqrt="ieee" (default): correctly rounded sqrt on the device -- what a reference model folded ON THE GPU gets
        (torch's device sqrt is correctly rounded), identical on every host, no device-to-host sync; parity with a
        GPU-side reference fold is not pinned by a fixture (no GPU where the goldens are made), the IEEE oracle
        (oracle/fold_ref.py) is.
        sqrt="reference": std from torch's CPU sqrt (MKL VML, one ulp off for ~0.7 % of the channels, host dependent) --
        what a reference model folded on the CPU gets; torch.equal to the committed CPU-made golden when given that
        host's std table (tests/test_gpu_repunet.py).  A callable supplies any other sqrt."""
        print("[ReparamUNet] Switching to deploy mode (fusing kernels)...")
        for module in list(self.modules()):
            if isinstance(module, MultiConvBlock):
                module.mergeMulKernels(sqrt)
        self._engine._packed.clear()
        self.invalidate_slots()
        print("[ReparamUNet] Fusion complete.")
        return self


def build_reparam_unet(**kwargs):
    if "kernel_sizes" not in kwargs:
        kwargs["kernel_sizes"] = [3]
    return ReparamUNet(**kwargs)
