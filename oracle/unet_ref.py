"""Oracle: functional U-Net / RepUNet forward on torch CPU fp32 (test infrastructure only).

State is a flat ``dict[str, Tensor]`` with the reference's state_dict keys, so golden
``init/*`` arrays feed it directly.  Gradients come from torch autograd on these ops.

Reference followed:
  DoubleConv            /root/reference/models/unet.py:5-19
  UNet.__init__/forward /root/reference/models/unet.py:21-62
  getBN (GroupNorm)     /root/reference/models/reparm_unet.py:17-25
  AsymmConvBlock        /root/reference/models/reparm_unet.py:39-72
  MultiConvBlock        /root/reference/models/reparm_unet.py:96-122
  ReparamDoubleConv     /root/reference/models/reparm_unet.py:150-170
  ReparamUNet.forward   /root/reference/models/reparm_unet.py:194-215
  mergeBN/kernelFuse    /root/reference/models/reparm_unet.py:28-37
  mergeAsyKernels       /root/reference/models/reparm_unet.py:74-93
  mergeMulKernels       /root/reference/models/reparm_unet.py:124-144
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]

BN_EPS = 1e-5
BN_MOMENTUM = 0.1          # nn.BatchNorm2d default, unet.py:11,14
REP_BN_MOMENTUM = 0.01     # commented-out BatchNorm variant, reparm_unet.py:14-15


class _RoundBoth(torch.autograd.Function):
    """bf16 storage emulation: the value is rounded going forward and its gradient going backward,
    the two points where the HIP path stores a bf16 activation / activation gradient."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _RoundFwd(torch.autograd.Function):
    """bf16 weight image of an fp32 master weight (gradient passes straight through)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _ra(x, emu):
    return _RoundBoth.apply(x) if emu else x


def _rw(w, emu):
    return _RoundFwd.apply(w) if emu else w


def batchnorm(x, st: State, prefix: str, train: bool, momentum: float, new_buffers: Optional[State]):
    """nn.BatchNorm2d semantics: biased var normalises, unbiased var feeds running_var,
    running = (1-m)*running + m*stat, num_batches_tracked += 1 (train only)."""
    w, b = st[prefix + ".weight"], st[prefix + ".bias"]
    rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    if not train:
        return F.batch_norm(x, rm, rv, w, b, False, 0.0, BN_EPS)
    mean = x.mean((0, 2, 3))
    var_b = x.var((0, 2, 3), unbiased=False)
    y = (x - mean[None, :, None, None]) / torch.sqrt(var_b[None, :, None, None] + BN_EPS)
    y = y * w[None, :, None, None] + b[None, :, None, None]
    if new_buffers is not None:
        n = x.numel() // x.shape[1]
        var_u = var_b.detach() * (n / max(n - 1, 1))
        new_buffers[prefix + ".running_mean"] = (1 - momentum) * rm + momentum * mean.detach()
        new_buffers[prefix + ".running_var"] = (1 - momentum) * rv + momentum * var_u
        new_buffers[prefix + ".num_batches_tracked"] = st[prefix + ".num_batches_tracked"] + 1
    return y


def _relu(x, key: str, decisions: Optional[dict], taps: Optional[dict]):
    """ReLU.  `decisions[key]` (a bool mask) pins WHICH side of the kink every element is on: ReLU and max-pool make
    the network piecewise linear, and two correct fp32 evaluations whose pre-activations differ by rounding can put an
    element with |pre-activation| ~ 1e-6 on different sides; its whole upstream gradient then moves by ~1e-2 (the BN
    batch sums that follow lose one term of a heavily cancelling sum).  A parity test that compares gradients to 1e-4
    therefore evaluates the oracle on the SAME linear piece as the path under test and checks separately that every
    pinned decision was a genuine near-tie (tests/test_gpu_realwidth.py)."""
    y = F.relu(x) if decisions is None else x * decisions[key].to(x.dtype)
    if taps is not None:
        taps["relu/" + key] = y
    return y


def double_conv(x, st: State, p: str, train: bool, nb: Optional[State], emu: bool = False, first: bool = False,
                decisions: Optional[dict] = None, taps: Optional[dict] = None):
    """unet.py:9-16: conv3x3(pad 1, bias) -> BN -> ReLU, twice; params at indices 0,1,3,4.
    emu=True rounds to bf16 wherever the HIP bf16 path stores a tensor (conv output, BN+ReLU output,
    packed weights; the Cin<=4 first conv keeps fp32 weights) so the bf16 kernels can be checked tightly."""
    x = F.conv2d(x, _rw(st[p + ".double_conv.0.weight"], emu and not first), st[p + ".double_conv.0.bias"], padding=1)
    x = _ra(_relu(batchnorm(_ra(x, emu), st, p + ".double_conv.1", train, BN_MOMENTUM, nb), p + ".a", decisions, taps), emu)
    x = F.conv2d(x, _rw(st[p + ".double_conv.3.weight"], emu), st[p + ".double_conv.3.bias"], padding=1)
    x = _ra(_relu(batchnorm(_ra(x, emu), st, p + ".double_conv.4", train, BN_MOMENTUM, nb), p + ".b", decisions, taps), emu)
    return x


def _levels(st: State) -> int:
    n = 0
    while f"downs.{n}." + "double_conv.0.weight" in st or f"downs.{n}.reparam_conv.0.conv_list.0.convSq.weight" in st \
            or f"downs.{n}.reparam_conv.0.mergedConv.weight" in st:
        n += 1
    return n


def _wire(x, st: State, block, n_levels: int, taps: Optional[dict] = None, emu: bool = False,
          decisions: Optional[dict] = None):
    """Encoder/decoder wiring shared by UNet.forward (unet.py:41-62) and
    ReparamUNet.forward (reparm_unet.py:194-215): skip FIRST in the concat."""
    skips = []
    for i in range(n_levels):
        x = block(x, f"downs.{i}")
        skips.append(x)
        if taps is not None:
            taps[f"skip{i}"] = x
            taps[f"pool{i}"] = F.max_pool2d(x.detach(), 2, 2, return_indices=True)[1]
        if decisions is None:
            x = F.max_pool2d(x, 2, 2)                                       # unet.py:26,47
        else:                                                               # pinned argmax (see _relu)
            idx = decisions[f"pool{i}"]
            x = x.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
    x = block(x, "bottleneck")                                              # unet.py:49
    skips = skips[::-1]
    for j in range(n_levels):
        x = _ra(F.conv_transpose2d(x, _rw(st[f"ups.{2*j}.weight"], emu), st[f"ups.{2*j}.bias"], stride=2), emu)  # unet.py:53
        if taps is not None and x.requires_grad:
            # gradient w.r.t. the ConvTranspose2d output: its per-channel sum is that layer's bias gradient, and tests use
            # it to state how heavily that sum cancels (conditioning of the comparison)
            x.register_hook(lambda g, j=j: taps.__setitem__(f"dup{j}", g.detach()))
        s = skips[j]
        if x.shape != s.shape:                                              # unet.py:56-57
            x = F.interpolate(x, size=s.shape[2:], mode="bilinear", align_corners=True)
        x = torch.cat((s, x), 1)                                            # unet.py:59
        x = block(x, f"ups.{2*j+1}")                                        # unet.py:60
    return F.conv2d(x, st["final_conv.weight"], st["final_conv.bias"])      # unet.py:62


def unet_forward(x, st: State, train: bool = True, new_buffers: Optional[State] = None, taps=None,
                 emulate_bf16: bool = False, decisions: Optional[dict] = None):
    """decisions: {"<block>.a" / "<block>.b": bool ReLU masks (NCHW), "pool<i>": int64 max-pool indices} pins the linear
    piece the network is evaluated on (see _relu); taps receives the oracle's own post-ReLU tensors and pool indices."""
    n = _levels(st)
    return _wire(x, st, lambda t, p: double_conv(t, st, p, train, new_buffers, emulate_bf16, first=(p == "downs.0" and t is x),
                                                 decisions=decisions, taps=taps),
                 n, taps, emulate_bf16, decisions)


# ---------------------------------------------------------------- RepUNet --------------

def _norm(x, st: State, p: str, norm: str, train: bool, nb):
    if norm == "group":                                # reparm_unet.py:17-25 (as shipped)
        c = x.shape[1]
        return F.group_norm(x, min(8, c), st[p + ".weight"], st[p + ".bias"], BN_EPS)
    return batchnorm(x, st, p, train, REP_BN_MOMENTUM, nb)   # reparm_unet.py:14-15 variant


def asymm_block(x, st: State, p: str, k: int, norm: str, train: bool, nb):
    """reparm_unet.py:67-72: GN(conv_kx1) + GN(conv_1xk) + GN(conv_kxk), bias-free convs."""
    if p + ".mergedConv.weight" in st:
        return F.conv2d(x, st[p + ".mergedConv.weight"], st[p + ".mergedConv.bias"], padding=k // 2)
    ve = _norm(F.conv2d(x, st[p + ".convVe1.weight"], None, padding=(k // 2, 0)), st, p + ".bnVe1", norm, train, nb)
    ho = _norm(F.conv2d(x, st[p + ".convHo1.weight"], None, padding=(0, k // 2)), st, p + ".bnHo1", norm, train, nb)
    sq = _norm(F.conv2d(x, st[p + ".convSq.weight"], None, padding=k // 2), st, p + ".bnSq", norm, train, nb)
    return ve + ho + sq


def multi_block(x, st: State, p: str, ks: List[int], norm: str, train: bool, nb):
    """reparm_unet.py:114-122: sum over sorted kernel sizes; deployed form is one conv."""
    if p + ".mergedConv.weight" in st:
        w = st[p + ".mergedConv.weight"]
        return F.conv2d(x, w, st[p + ".mergedConv.bias"], padding=w.shape[-1] // 2)
    y = None
    for i, k in enumerate(sorted(ks)):
        o = asymm_block(x, st, f"{p}.conv_list.{i}", k, norm, train, nb)
        y = o if y is None else y + o
    return y


def rep_double_conv(x, st, p, ks, norm, train, nb, decisions=None, taps=None):
    """reparm_unet.py:162-167."""
    x = _relu(multi_block(x, st, p + ".reparam_conv.0", ks, norm, train, nb), p + ".a", decisions, taps)
    x = _relu(multi_block(x, st, p + ".reparam_conv.2", ks, norm, train, nb), p + ".b", decisions, taps)
    return x


def repunet_forward(x, st: State, kernel_sizes, norm="group", train=True, new_buffers=None, taps=None, decisions=None):
    """decisions / taps: as unet_forward (keys "<block>.a", "<block>.b" for the two ReLUs of a ReparamDoubleConv, "pool<i>")."""
    n = _levels(st)
    return _wire(x, st, lambda t, p: rep_double_conv(t, st, p, kernel_sizes, norm, train, new_buffers, decisions, taps),
                 n, taps, False, decisions)


# ---------------------------------------------------------------- fold -----------------

def merge_bn(conv_w, bn_w, bn_b, rm, rv, eps=BN_EPS):
    """reparm_unet.py:28-31, same operation order (bit-exact target)."""
    std = (rv + eps).sqrt()
    t = (bn_w / std).reshape(-1, 1, 1, 1)
    return conv_w * t, bn_b - rm * bn_w / std


def kernel_fuse(target, sec):
    """reparm_unet.py:33-37: centre-embed add, in place."""
    sh, sw = sec.size(2), sec.size(3)
    th, tw = target.size(2), target.size(3)
    target[:, :, th // 2 - sh // 2: th // 2 - sh // 2 + sh, tw // 2 - sw // 2: tw // 2 - sw // 2 + sw] += sec


def fold_asymm(st: State, p: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """reparm_unet.py:74-82."""
    def mb(conv, bn):
        return merge_bn(st[f"{p}.{conv}.weight"], st[f"{p}.{bn}.weight"], st[f"{p}.{bn}.bias"],
                        st[f"{p}.{bn}.running_mean"], st[f"{p}.{bn}.running_var"])
    w_ve, b_ve = mb("convVe1", "bnVe1")
    w_ho, b_ho = mb("convHo1", "bnHo1")
    w_sq, b_sq = mb("convSq", "bnSq")
    kernel_fuse(w_sq, w_ve)
    kernel_fuse(w_sq, w_ho)
    return w_sq, b_sq + b_ve + b_ho


def fold_multi(st: State, p: str, ks: List[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """reparm_unet.py:124-135: clone of the largest, += each smaller in ascending order."""
    ks = sorted(ks)
    parts = [fold_asymm(st, f"{p}.conv_list.{i}") for i in range(len(ks))]
    tw, tb = parts[-1][0].clone(), parts[-1][1].clone()
    for i in range(len(ks) - 1):
        kernel_fuse(tw, parts[i][0])
        tb += parts[i][1]
    return tw, tb


def fold_state(st: State, ks: List[int]) -> State:
    """switch_to_deploy (reparm_unet.py:217-225) on a state dict: every
    '<blk>.reparam_conv.{0,2}.conv_list.*' group becomes '<blk>.reparam_conv.{0,2}.mergedConv.*'."""
    out: State = {}
    prefixes = []
    for k in st:
        if ".conv_list." in k:
            pre = k.split(".conv_list.")[0]
            if pre not in prefixes:
                prefixes.append(pre)
        else:
            out[k] = st[k]
    for pre in prefixes:
        w, b = fold_multi(st, pre, ks)
        out[pre + ".mergedConv.weight"] = w
        out[pre + ".mergedConv.bias"] = b
    return out


def max_pool_indices(x):
    """Oracle for pool indices: ATen's own kernel."""
    return F.max_pool2d(x, 2, 2, return_indices=True)
