"""Tensor-level wrappers over the C ABI (include/unet_hip.h).

Every function takes torch CUDA(HIP) tensors, checks shapes on the host (a kernel fault can reset
the whole GPU host) and enqueues the HIP kernel on torch's current stream.  torch is used for
device memory and streams only.
"""
from dataclasses import dataclass
from typing import Optional

import torch

from ._lib import UNetHipError, check, lib

F32, BF16 = 0, 1
_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def torch_dtype(dt: int):
    return _TORCH_DT[dt]


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise UNetHipError(f"unsupported activation dtype {t.dtype}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def set_dynamic_tiles(level: int) -> None:
    """Tile scheduling of the persistent bf16 conv / weight-gradient kernels (include/unet_hip.h: unet_set_dynamic_tiles):
    0 static partition (default), 1 per-XCD atomic tile queues for launches with several rounds of tiles (runs that share
    the GPU with collectives), 2 queues wherever the kernels allow (tests)."""
    check(lib().unet_set_dynamic_tiles(int(level)))


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise UNetHipError("unet_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback)")


@dataclass
class Act:
    """A channel slice [off, off+C) of a contiguous NHWC buffer [N,H,W,ld]."""
    buf: torch.Tensor
    off: int
    C: int

    @property
    def N(self): return self.buf.shape[0]
    @property
    def H(self): return self.buf.shape[1]
    @property
    def W(self): return self.buf.shape[2]
    @property
    def ld(self): return self.buf.shape[3]
    @property
    def M(self): return self.buf.shape[0] * self.buf.shape[1] * self.buf.shape[2]
    @property
    def dt(self): return dt_of(self.buf)

    def check(self):
        b = self.buf
        if b.dim() != 4 or not b.is_contiguous() or not b.is_cuda:
            raise UNetHipError("activation buffers must be contiguous 4-D NHWC device tensors")
        if self.off < 0 or self.off + self.C > b.shape[3]:
            raise UNetHipError("channel slice outside the buffer")
        return self


def dense(N, H, W, C, dt, device) -> Act:
    return Act(torch.empty((N, H, W, C), dtype=_TORCH_DT[dt], device=device), 0, C)


class Workspace:
    """Small fp64/fp32 scratch reused by stream-ordered kernels, plus the wgrad slab."""

    def __init__(self, device):
        self.device = device
        self.acc = torch.empty(16384, dtype=torch.float64, device=device)
        self.c1c2 = torch.empty(8192, dtype=torch.float32, device=device)
        self.slab = None
        # zeroed fp64 arena: the engines clear it ONCE per backward pass (clear_arena) and the BatchNorm backward launches
        # take their accumulators out of it (take_zeroed) instead of each clearing its own (18 memsets per U-Net step)
        self.arena = torch.empty(32768, dtype=torch.float64, device=device)
        self._arena_off = self.arena.numel()          # nothing to hand out until clear_arena()

    def clear_arena(self):
        self.arena.zero_()
        self._arena_off = 0

    def take_zeroed(self, n: int):
        """n zeroed doubles (on a 256-byte boundary) that nobody else will be handed before the next clear_arena(), or None."""
        n = (n + 31) // 32 * 32
        if self._arena_off + n > self.arena.numel():
            return None
        t = self.arena[self._arena_off:self._arena_off + n]
        self._arena_off += n
        return t

    def get_slab(self, nfloats):
        if self.slab is None or self.slab.numel() < nfloats:
            self.slab = None
            self.slab = torch.empty(nfloats, dtype=torch.float32, device=self.device)
        return self.slab


# ----------------------------------------------------------------------------- packing

def pack_weights(w: torch.Tensor, mode: int, dt: int) -> torch.Tensor:
    """fp32 OIHW (modes 0,1) or (Cin,Cout,2,2) (modes 2,3) -> packed compute-dtype image."""
    _need_cuda(w)
    w = w.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    O, I = w.shape[0], w.shape[1]
    nt = w.shape[2] * w.shape[3]
    out = torch.empty(w.numel(), dtype=_TORCH_DT[dt], device=w.device)
    check(lib().unet_pack_weights(dt, mode, _p(w), _p(out), O, I, nt, _stream()), "pack_weights")
    return out


def fold_bn_conv(w, bias, gamma, beta, running_mean, running_var, eps):
    """(w', b') of conv followed by eval-mode BatchNorm2d, fp32 on the device."""
    _need_cuda(w, gamma, beta, running_mean, running_var)
    w = w.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    O = w.shape[0]
    for t in (gamma, beta, running_mean, running_var) + ((bias,) if bias is not None else ()):
        if t.numel() != O or t.dtype != torch.float32 or not t.is_contiguous():
            raise UNetHipError("fold_bn_conv: per-channel tensors must be contiguous fp32 of length Cout")
    wo = torch.empty_like(w)
    bo = torch.empty(O, dtype=torch.float32, device=w.device)
    check(lib().unet_fold_bn_conv(_p(w), _p(bias), _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps),
                                  O, w.numel() // O, _p(wo), _p(bo), _stream()), "fold_bn_conv")
    return wo, bo


def pack_weights_multi(items, dt: int):
    """items: [(fp32 weight, mode, destination tensor)] -> every packed image refreshed in one launch per 40."""
    import ctypes
    n = len(items)
    if n == 0:
        return
    srcs, dsts, modes, Os, Is, NTs = [], [], [], [], [], []
    for w, mode, out in items:
        _need_cuda(w, out)
        if w.dtype != torch.float32 or not w.is_contiguous() or w.dim() != 4:
            raise UNetHipError("pack_weights_multi: weights must be contiguous 4-D fp32 tensors")
        if out.numel() != w.numel() or out.dtype != _TORCH_DT[dt] or not out.is_contiguous():
            raise UNetHipError("pack_weights_multi: destination does not match the weight")
        srcs.append(w.data_ptr()); dsts.append(out.data_ptr()); modes.append(mode)
        Os.append(w.shape[0]); Is.append(w.shape[1]); NTs.append(w.shape[2] * w.shape[3])
    IA, PA = ctypes.c_int * n, ctypes.c_void_p * n
    check(lib().unet_pack_weights_multi(dt, n, IA(*modes), PA(*srcs), PA(*dsts), IA(*Os), IA(*Is), IA(*NTs), _stream()),
          "pack_weights_multi")


# ----------------------------------------------------------------------------- optimizer tail

def _f32_list(ts, what):
    for t in ts:
        if not t.is_cuda:
            raise UNetHipError("unet_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback)")
        if t.dtype != torch.float32 or not t.is_contiguous() or t.is_sparse:
            raise UNetHipError(f"{what}: tensors must be dense contiguous fp32")


def mt_blocks(numels) -> int:
    import ctypes
    n = len(numels)
    r = lib().unet_mt_blocks(n, (ctypes.c_longlong * n)(*numels))
    if r < 0:
        raise UNetHipError("mt_blocks: too many elements")
    return r


def grad_norm(grads, max_norm: float, partials: torch.Tensor, norm_out: torch.Tensor, grad_scale: float = 1.0):
    """Global L2 norm of grad_scale*`grads` and clip_grad_norm_'s coefficient times grad_scale, on the device:
    norm_out[0] = norm, norm_out[1] = grad_scale * min(1, max_norm / (norm + 1e-6))."""
    import ctypes
    _f32_list(grads, "grad_norm")
    n = len(grads)
    numels = [g.numel() for g in grads]
    nb = mt_blocks(numels)
    if partials.dtype != torch.float64 or partials.numel() < nb or norm_out.dtype != torch.float32 or norm_out.numel() < 2:
        raise UNetHipError("grad_norm: workspace too small")
    NA = (ctypes.c_longlong * n)(*numels)
    check(lib().unet_grad_sumsq(n, (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads]), NA, _p(partials), _stream()), "grad_sumsq")
    check(lib().unet_grad_norm_finalize(_p(partials), nb, float(max_norm), float(grad_scale), _p(norm_out), _stream()), "grad_norm_finalize")


def adamw_step(params, grads, exp_avg, exp_avg_sq, norm_out: Optional[torch.Tensor], write_grads: bool,
               lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, step: int,
               veto: Optional[torch.Tensor] = None):
    """veto: one fp32 device element; > 0 at kernel time turns the whole update into a no-op."""
    import ctypes
    if veto is not None:
        _need_cuda(veto)
        if veto.dtype != torch.float32 or veto.numel() < 1:
            raise UNetHipError("adamw_step: veto must be an fp32 device tensor")
    n = len(params)
    if not (len(grads) == len(exp_avg) == len(exp_avg_sq) == n):
        raise UNetHipError("adamw_step: list lengths differ")
    for lst in (params, grads, exp_avg, exp_avg_sq):
        _f32_list(lst, "adamw_step")
    numels = [p.numel() for p in params]
    for g, m, v, k in zip(grads, exp_avg, exp_avg_sq, numels):
        if g.numel() != k or m.numel() != k or v.numel() != k:
            raise UNetHipError("adamw_step: parameter / gradient / state sizes differ")
    PA = ctypes.c_void_p * n
    check(lib().unet_adamw_step(n, PA(*[t.data_ptr() for t in params]), PA(*[t.data_ptr() for t in grads]),
                                PA(*[t.data_ptr() for t in exp_avg]), PA(*[t.data_ptr() for t in exp_avg_sq]),
                                (ctypes.c_longlong * n)(*numels), _p(norm_out), int(bool(write_grads)),
                                float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), _p(veto), _stream()),
          "adamw_step")


# ----------------------------------------------------------------------------- convolutions

def conv_fwd(x: Act, wpacked: torch.Tensor, bias: Optional[torch.Tensor], y: Act, kh: int, kw: int,
             Cin: int, Cout: int, stats: bool = False, stride_h: int = 1, epi: int = 0,
             pad_h: Optional[int] = None, pad_w: Optional[int] = None,
             y2: Optional[Act] = None, csplit: int = 0, accumulate: bool = False, relu: bool = False,
             per_image_stats: bool = False):
    """y = conv(x) (+bias).  Returns the BN partial-stat rows tensor when stats=True (per_image_stats: one row per
    pixel tile, grouped per image, as gn_finalize needs; otherwise only the sum over the rows is defined).
    epi=1: 1x1 conv with Cout = 4*C and 2x2 pixel shuffle into y (ConvTranspose2d forward)."""
    x.check(); y.check()
    dt = x.dt
    if y.dt != dt or wpacked.dtype != _TORCH_DT[dt]:
        raise UNetHipError("conv: dtype mismatch between input, weights and output")
    if x.C != Cin:
        raise UNetHipError(f"conv: input slice has {x.C} channels, weights expect {Cin}")
    if wpacked.numel() != kh * kw * Cin * Cout:
        raise UNetHipError("conv: packed weight size mismatch")
    ph = kh // 2 if pad_h is None else pad_h
    pw = kw // 2 if pad_w is None else pad_w
    Hout = (x.H + 2 * ph - kh) // stride_h + 1
    Wout = x.W + 2 * pw - kw + 1
    if epi == 1:
        if (y.N, y.H, y.W) != (x.N, 2 * Hout, 2 * Wout) or y.C * 4 != Cout:
            raise UNetHipError("convT: output buffer shape mismatch")
    else:
        ytot = y.C + (y2.C if y2 is not None else 0)
        if (y.N, y.H, y.W) != (x.N, Hout, Wout) or ytot != Cout:
            raise UNetHipError(f"conv: output buffer shape mismatch {(y.N, y.H, y.W, ytot)} vs {(x.N, Hout, Wout, Cout)}")
        if y2 is not None:
            y2.check()
            if (y2.N, y2.H, y2.W) != (x.N, Hout, Wout) or csplit != y.C:
                raise UNetHipError("conv: split output mismatch")
    if bias is not None:
        _need_cuda(bias)
        nb = Cout // 4 if epi == 1 else Cout
        if bias.numel() != nb or bias.dtype != torch.float32:
            raise UNetHipError("conv: bias must be fp32 of length Cout")
    rows = None
    flags = (1 if accumulate else 0) | (2 if relu else 0) | (4 if per_image_stats else 0)
    cs = csplit if y2 is not None else 0
    if stats:
        nrows = lib().unet_conv_tiles(dt, kh, kw, stride_h, epi, x.N, Hout, Wout, Cin, Cout, flags, cs)
        rows = torch.empty((nrows, Cout, 2), dtype=torch.float32, device=x.buf.device)
    check(lib().unet_conv_fwd(dt, kh, kw, stride_h, epi, _p(x.buf), _p(wpacked), _p(bias), _p(y.buf), _p(rows),
                              x.N, x.H, x.W, Cin, x.ld, x.off, Hout, Wout, Cout, y.ld, y.off, ph, pw,
                              _p(y2.buf) if y2 is not None else None, y2.ld if y2 is not None else 0,
                              y2.off if y2 is not None else 0, cs, flags, _stream()), "conv_fwd")
    return rows


def conv_wgrad(dz: Act, x: Act, kh: int, kw: int, ws: Workspace, dw: Optional[torch.Tensor] = None,
               stride_h: int = 1, pad_h: Optional[int] = None, pad_w: Optional[int] = None,
               dw_taps: Optional[int] = None, dw_tap0: int = 0, accumulate: bool = False, reduce: bool = True):
    """dW[o][i][tap] = sum_pixels dz[.,o] * x[. + tap, i].  Returns (dw, nsplit, slab)."""
    dz.check(); x.check()
    dt = x.dt
    if dz.dt != dt:
        raise UNetHipError("wgrad: dtype mismatch")
    ph = kh // 2 if pad_h is None else pad_h
    pw = kw // 2 if pad_w is None else pad_w
    Cout, Cin = dz.C, x.C
    if dz.N != x.N:
        raise UNetHipError("wgrad: batch mismatch")
    nt = kh * kw
    dw_taps = nt if dw_taps is None else dw_taps
    nsplit = lib().unet_conv_wgrad_splits(dt, kh, kw, stride_h, dz.N, dz.H, dz.W, Cin, Cout)
    slab = ws.get_slab(nsplit * nt * Cout * Cin)
    if reduce and dw is None:
        dw = torch.empty((Cout, Cin, dw_taps), dtype=torch.float32, device=x.buf.device)
    if reduce and (dw.numel() != Cout * Cin * dw_taps or dw.dtype != torch.float32 or not dw.is_contiguous()):
        raise UNetHipError("wgrad: gradient tensor shape mismatch")
    check(lib().unet_conv_wgrad(dt, kh, kw, stride_h, _p(dz.buf), _p(x.buf), _p(slab), _p(dw) if reduce else None,
                                1 if accumulate else 0, dw_taps, dw_tap0,
                                x.N, x.H, x.W, Cin, x.ld, x.off, dz.H, dz.W, Cout, dz.ld, dz.off, ph, pw, _stream()),
          "conv_wgrad")
    return dw, nsplit, slab


def convT_wgrad(xin: Act, dup: Act, ws: Workspace) -> torch.Tensor:
    """Weight grad (Ci,Co,2,2) of ConvTranspose2d(k=2,s=2): xin [N,h,w,Ci], dup DENSE [N,2h,2w,Co]."""
    if dup.off != 0 or dup.ld != dup.C:
        raise UNetHipError("convT wgrad needs a dense output-gradient buffer")
    Ci, Co = xin.C, dup.C
    view = Act(dup.buf.view(dup.N, dup.H, dup.W // 2, 2 * Co), 0, 2 * Co)
    _, nsplit, slab = conv_wgrad(xin, view, 2, 1, ws, stride_h=2, pad_h=0, pad_w=0, reduce=False)
    dw = torch.empty((Ci, Co, 2, 2), dtype=torch.float32, device=xin.buf.device)
    check(lib().unet_convT_wgrad_reduce(_p(slab), _p(dw), nsplit, Ci, Co, _stream()), "convT_wgrad_reduce")
    return dw


def convT_dgrad(dup: Act, wpacked3: torch.Tensor, Ci: int, out: Act):
    """dx [N,h,w,Ci] of ConvTranspose2d(k=2,s=2) from the dense output gradient [N,2h,2w,Co]."""
    if dup.off != 0 or dup.ld != dup.C:
        raise UNetHipError("convT dgrad needs a dense output-gradient buffer")
    Co = dup.C
    view = Act(dup.buf.view(dup.N, dup.H, dup.W // 2, 2 * Co), 0, 2 * Co)
    conv_fwd(view, wpacked3, None, out, 2, 1, 2 * Co, Ci, stride_h=2, pad_h=0, pad_w=0)


def conv_first_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], z: Act, kh: int, kw: int, stats=True):
    """x fp32 [N,H,W,Cin] (Cin<=4), w fp32 OIHW -> z (dense) and BN partial rows."""
    _need_cuda(x, w)
    z.check()
    N, H, W, Cin = x.shape
    Cout = z.C
    if x.dtype != torch.float32 or not x.is_contiguous() or z.off != 0 or z.ld != Cout or (z.N, z.H, z.W) != (N, H, W):
        raise UNetHipError("first conv: bad input/output layout")
    if tuple(w.shape) != (Cout, Cin, kh, kw) or w.dtype != torch.float32 or not w.is_contiguous():
        raise UNetHipError("first conv: bad weight tensor")
    rows = None
    if stats:
        nrows = lib().unet_conv_first_rows(z.dt, N, H, W, Cout)
        if nrows <= 0:
            raise UNetHipError("first conv: unsupported Cout")
        rows = torch.empty((nrows, Cout, 2), dtype=torch.float32, device=x.device)
    check(lib().unet_conv_first_fwd(z.dt, kh, kw, _p(x), _p(w), _p(bias), _p(z.buf), _p(rows), N, H, W, Cin, Cout, _stream()),
          "conv_first_fwd")
    return rows


def conv_first_stats(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], dt: int) -> torch.Tensor:
    """BN partial rows of the Cin = 1 3x3 first conv WITHOUT storing its output (the fused first layer recomputes it)."""
    _need_cuda(x, w)
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    if Cin != 1 or tuple(w.shape) != (Cout, 1, 3, 3) or x.dtype != torch.float32 or not x.is_contiguous() \
            or w.dtype != torch.float32 or not w.is_contiguous():
        raise UNetHipError("first conv statistics: fp32 [N,H,W,1] input and (Cout,1,3,3) weights expected")
    nrows = lib().unet_conv_first_rows(dt, N, H, W, Cout)
    if nrows <= 0:
        raise UNetHipError("first conv: unsupported Cout")
    rows = torch.empty((nrows, Cout, 2), dtype=torch.float32, device=x.device)
    check(lib().unet_conv_first_fwd(dt, 3, 3, _p(x), _p(w), _p(bias), None, _p(rows), N, H, W, 1, Cout, _stream()), "conv_first_stats")
    return rows


def first_stats_moments(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]):
    """(BN partial rows [2, Cout, 2], patch moments [64] fp64) of the Cin = 1 3x3 first conv from the 3x3 patch moments of the
    image: neither the conv output nor its per-channel values are computed (csrc/edge_fused.hip).  The moments go to
    first_bn_relu_bwd(moments=...), which needs the same 55 numbers."""
    _need_cuda(x, w)
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    if Cin != 1 or tuple(w.shape) != (Cout, 1, 3, 3) or x.dtype != torch.float32 or not x.is_contiguous() \
            or w.dtype != torch.float32 or not w.is_contiguous():
        raise UNetHipError("first conv statistics: fp32 [N,H,W,1] input and (Cout,1,3,3) weights expected")
    rows = torch.empty((2, Cout, 2), dtype=torch.float32, device=x.device)
    mom = torch.empty(64, dtype=torch.float64, device=x.device)
    check(lib().unet_first_stats_moments(_p(x), _p(w), _p(bias), N, H, W, Cout, _p(mom), _p(rows), _stream()), "first_stats_moments")
    return rows, mom


def first_bn_act(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], st: "BNState", y: Act, relu=True):
    """y = act(scale * conv3x3(x) + shift) for the Cin = 1 first layer in one pass (the conv output is not stored)."""
    _need_cuda(x, w)
    y.check()
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    if Cin != 1 or tuple(w.shape) != (Cout, 1, 3, 3) or (y.N, y.H, y.W, y.C) != (N, H, W, Cout) or st.scale.numel() != Cout:
        raise UNetHipError("first_bn_act: shape mismatch")
    if x.dtype != torch.float32 or not x.is_contiguous() or w.dtype != torch.float32 or not w.is_contiguous():
        raise UNetHipError("first_bn_act: fp32 contiguous input and weights expected")
    check(lib().unet_first_bn_act(y.dt, 1 if relu else 0, _p(x), _p(w), _p(bias), _p(st.scale), _p(st.shift), _p(y.buf), y.ld, y.off,
                                  N, H, W, Cout, _stream()), "first_bn_act")


def first_bn_relu_bwd(dy: Act, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], st: "BNState", ws: "Workspace",
                      relu=True, moments: Optional[torch.Tensor] = None):
    """BN(+ReLU) backward of the fused first layer -> (dgamma, dbeta, dbias, dw (Cout,1,3,3)); nothing else is written."""
    dy.check()
    _need_cuda(x, w)
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    if Cin != 1 or tuple(w.shape) != (Cout, 1, 3, 3) or (dy.N, dy.H, dy.W, dy.C) != (N, H, W, Cout):
        raise UNetHipError("first_bn_relu_bwd: shape mismatch")
    if x.dtype != torch.float32 or not x.is_contiguous() or w.dtype != torch.float32 or not w.is_contiguous():
        raise UNetHipError("first_bn_relu_bwd: fp32 contiguous input and weights expected")
    if 12 * Cout + 64 > ws.acc.numel() or 2 * Cout > ws.c1c2.numel():
        raise UNetHipError("first_bn_relu_bwd: workspace too small")
    g = torch.empty((3, Cout), dtype=torch.float32, device=x.device)
    dw = torch.empty((Cout, 1, 3, 3), dtype=torch.float32, device=x.device)
    check(lib().unet_first_bn_relu_bwd(dy.dt, _bn_bwd_flags(st, relu), _p(dy.buf), dy.ld, dy.off, _p(x), _p(w), _p(bias),
                                       _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd), _p(g[0]), _p(g[1]), _p(g[2]), _p(dw),
                                       _p(ws.c1c2), _p(ws.acc), _p(moments), N, H, W, Cout, _stream()), "first_bn_relu_bwd")
    return g[0], g[1], g[2], dw


def affine_act_out(z: Act, st: "BNState", w: torch.Tensor, bias: Optional[torch.Tensor], relu=True,
                   y: Optional[Act] = None) -> torch.Tensor:
    """logits (fp32 NCHW) = W act(scale*z + shift) + b in one pass; the activation is written only if `y` is given."""
    z.check()
    K, C = w.shape[0], w.shape[1]
    if C != z.C or w.dtype != torch.float32 or not w.is_contiguous() or st.scale.numel() != C:
        raise UNetHipError("affine_act_out: shape mismatch")
    if y is not None:
        y.check()
        if y.C != C or y.M != z.M or y.dt != z.dt:
            raise UNetHipError("affine_act_out: activation buffer shape mismatch")
    logits = torch.empty((z.N, K, z.H, z.W), dtype=torch.float32, device=z.buf.device)
    check(lib().unet_affine_act_out(z.dt, 1 if relu else 0, _p(z.buf), z.ld, z.off, _p(st.scale), _p(st.shift), _p(w), _p(bias),
                                    _p(logits), _p(y.buf) if y is not None else None, y.ld if y is not None else 0,
                                    y.off if y is not None else 0, z.M, z.H * z.W, C, K, _stream()), "affine_act_out")
    return logits


def fused_out_classes_ok(K: int, C: int, dt: int) -> bool:
    return 1 <= K <= 4 and C % 8 == 0 and C // (8 if dt == BF16 else 4) <= 64


def bn_relu_bwd_out(dlogits: torch.Tensor, w: torch.Tensor, z: Act, dz: Act, st: "BNState", ws: "Workspace", relu=True):
    """BN(+ReLU) backward of the last layer with dy = W^T dlogits recomputed -> (dgamma, dbeta, dbias, dW (K,C), db (K))."""
    z.check(); dz.check()
    K, C = w.shape[0], w.shape[1]
    if C != z.C or dz.C != C or dz.M != z.M or dz.dt != z.dt or w.dtype != torch.float32 or not w.is_contiguous():
        raise UNetHipError("bn_relu_bwd_out: shape mismatch")
    if dlogits.dtype != torch.float32 or not dlogits.is_contiguous() or tuple(dlogits.shape) != (z.N, K, z.H, z.W):
        raise UNetHipError("bn_relu_bwd_out: dlogits must be contiguous fp32 NCHW")
    if 3 * C + K * C + K > ws.acc.numel() or 2 * C > ws.c1c2.numel():
        raise UNetHipError("bn_relu_bwd_out: workspace too small")
    g = torch.empty((3, C), dtype=torch.float32, device=z.buf.device)
    dw = torch.empty((K, C), dtype=torch.float32, device=z.buf.device)
    db = torch.empty((K,), dtype=torch.float32, device=z.buf.device)
    check(lib().unet_bn_relu_bwd_out(z.dt, _bn_bwd_flags(st, relu), _p(dlogits), _p(w), K, z.H * z.W, _p(z.buf), z.ld, z.off,
                                     _p(dz.buf), dz.ld, dz.off, _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd),
                                     _p(g[0]), _p(g[1]), _p(g[2]), _p(dw), _p(db), _p(ws.c1c2), _p(ws.acc), z.M, C, _stream()),
          "bn_relu_bwd_out")
    return g[0], g[1], g[2], dw, db


def conv_first_wgrad(dz: Act, x: torch.Tensor, kh: int, kw: int, ws: Workspace) -> torch.Tensor:
    dz.check()
    N, H, W, Cin = x.shape
    Cout = dz.C
    if dz.off != 0 or dz.ld != Cout or (dz.N, dz.H, dz.W) != (N, H, W):
        raise UNetHipError("first conv wgrad: bad gradient layout")
    if Cout * kh * kw > ws.acc.numel():
        raise UNetHipError("first conv wgrad: workspace too small for Cout * taps accumulators")
    dw = torch.empty((Cout, Cin, kh, kw), dtype=torch.float32, device=x.device)
    check(lib().unet_conv_first_wgrad(dz.dt, kh, kw, _p(dz.buf), _p(x), _p(dw), N, H, W, Cin, Cout, _p(ws.acc), _stream()),
          "conv_first_wgrad")
    return dw


def conv_out_fwd(x: Act, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """final 1x1 conv -> fp32 NCHW logits."""
    x.check()
    K, C = w.shape[0], w.shape[1]
    if C != x.C or w.dtype != torch.float32 or not w.is_contiguous():
        raise UNetHipError("final conv: bad weight tensor")
    logits = torch.empty((x.N, K, x.H, x.W), dtype=torch.float32, device=x.buf.device)
    check(lib().unet_conv_out_fwd(x.dt, _p(x.buf), x.ld, x.off, _p(w), _p(bias), _p(logits), x.M, x.H * x.W, C, K, _stream()),
          "conv_out_fwd")
    return logits


def conv_out_bwd(dlogits: torch.Tensor, x: Act, w: torch.Tensor, ws: Workspace):
    x.check()
    K, C = w.shape[0], w.shape[1]
    if dlogits.dtype != torch.float32 or not dlogits.is_contiguous() or tuple(dlogits.shape) != (x.N, K, x.H, x.W):
        raise UNetHipError("final conv bwd: dlogits must be contiguous fp32 NCHW")
    if K * C + K > ws.acc.numel():
        raise UNetHipError("final conv bwd: workspace too small")
    dx = dense(x.N, x.H, x.W, C, x.dt, x.buf.device)
    dw = torch.empty((K, C, 1, 1), dtype=torch.float32, device=x.buf.device)
    db = torch.empty((K,), dtype=torch.float32, device=x.buf.device)
    check(lib().unet_conv_out_bwd(x.dt, _p(dlogits), _p(x.buf), x.ld, x.off, _p(w), _p(dx.buf), _p(dw), _p(db),
                                  x.M, x.H * x.W, C, K, _p(ws.acc), _stream()), "conv_out_bwd")
    return dx, dw, db


# ----------------------------------------------------------------------------- BN / ReLU / pool

@dataclass
class BNState:
    scale: torch.Tensor
    shift: torch.Tensor
    mean: Optional[torch.Tensor] = None
    invstd: Optional[torch.Tensor] = None
    frozen: bool = False        # eval-mode BatchNorm: mean / invstd are the running statistics (no gradient through them)


def bn_finalize(rows: torch.Tensor, count: int, gamma, beta, running_mean, running_var, nbt, momentum, eps,
                ws: Workspace) -> BNState:
    C = rows.shape[1]
    dev = rows.device
    if 2 * C > ws.acc.numel():
        raise UNetHipError("bn_finalize: workspace too small")
    buf = torch.empty((4, C), dtype=torch.float32, device=dev)
    st = BNState(buf[0], buf[1], buf[2], buf[3])
    check(lib().unet_bn_finalize(_p(rows), rows.shape[0], C, count, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                 _p(nbt), float(momentum), float(eps), _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd),
                                 _p(ws.acc), _stream()), "bn_finalize")
    return st


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps, for_backward: bool = False) -> BNState:
    """Eval-mode coefficients; for_backward also saves (running_mean, invstd) so bn_relu_bwd can run with frozen statistics."""
    _need_cuda(gamma, beta, running_mean, running_var)
    C = gamma.numel()
    for t in (gamma, beta, running_mean, running_var):
        if t.numel() != C or t.dtype != torch.float32 or not t.is_contiguous():
            raise UNetHipError("bn_eval_coeffs: per-channel tensors must be contiguous fp32 of equal length")
    buf = torch.empty((4 if for_backward else 2, C), dtype=torch.float32, device=gamma.device)
    check(lib().unet_bn_eval_coeffs(C, _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), _p(buf[0]), _p(buf[1]),
                                    _p(buf[2]) if for_backward else None, _p(buf[3]) if for_backward else None,
                                    _stream()), "bn_eval_coeffs")
    if for_backward:
        return BNState(buf[0], buf[1], buf[2], buf[3], frozen=True)
    return BNState(buf[0], buf[1])


def _bn_bwd_flags(st: BNState, relu: bool) -> int:
    if st.mean is None or st.invstd is None:
        raise UNetHipError("bn_relu_bwd: this BatchNorm state was built without the statistics backward needs "
                           "(bn_eval_coeffs(..., for_backward=True) for an eval-mode forward)")
    return (1 if relu else 0) | (2 if st.frozen else 0)


def affine_act(z: Act, y: Act, st: BNState, relu=True):
    z.check(); y.check()
    if z.C != y.C or z.M != y.M or z.dt != y.dt or st.scale.numel() != z.C:
        raise UNetHipError("affine_act: shape mismatch")
    check(lib().unet_affine_act(z.dt, 1 if relu else 0, _p(z.buf), z.ld, z.off, _p(y.buf), y.ld, y.off,
                                _p(st.scale), _p(st.shift), z.M, z.C, _stream()), "affine_act")


def affine_act_pool(z: Act, y: Act, st: BNState, relu=True):
    """y = relu(z*scale+shift) and its 2x2 max-pool in one pass -> (pooled dense Act, uint8 code)."""
    z.check(); y.check()
    if z.C != y.C or z.M != y.M or z.dt != y.dt or st.scale.numel() != z.C:
        raise UNetHipError("affine_act_pool: shape mismatch")
    if z.H % 2 or z.W % 2:
        raise UNetHipError(f"affine_act_pool: H and W must be even, got {z.H}x{z.W}")
    pooled = dense(z.N, z.H // 2, z.W // 2, z.C, z.dt, z.buf.device)
    code = torch.empty((z.N, z.H // 2, z.W // 2, z.C), dtype=torch.uint8, device=z.buf.device)
    check(lib().unet_affine_act_pool(z.dt, 1 if relu else 0, _p(z.buf), z.ld, z.off, _p(y.buf), y.ld, y.off,
                                     _p(st.scale), _p(st.shift), _p(pooled.buf), _p(code), z.N, z.H, z.W, z.C, _stream()),
          "affine_act_pool")
    return pooled, code


def conv_dgrad_bnred(dz: Act, w_dgrad: torch.Tensor, dy: Act, z: Act, st: "BNState") -> Optional[torch.Tensor]:
    """dy = 3x3 dgrad of dz (w_dgrad = pack_weights(w, 1, dt)) AND, from the same kernel's epilogue, the per-tile rows
    (sum dy', sum dy'*xhat) of the BatchNorm + ReLU whose pre-norm tensor is z and whose output gradient dy is: the first pass of
    bn_relu_bwd without a pass of its own (hand the rows to bn_relu_bwd(sum_rows=...)).  Returns None -- nothing launched --
    when the shape is not one the fused kernel takes (bf16, whole 16x32 tiles, 64-channel blocks): use conv_fwd then."""
    dz.check(); dy.check(); z.check()
    N, H, W, Cin, Cout = dz.N, dz.H, dz.W, dz.C, dy.C
    if (dy.N, dy.H, dy.W) != (N, H, W) or (z.N, z.H, z.W, z.C) != (N, H, W, Cout) or not (dz.dt == dy.dt == z.dt):
        raise UNetHipError("conv_dgrad_bnred: shape mismatch")
    if st.mean is None or st.invstd is None or not lib().unet_conv_dgrad_bnred_ok(dz.dt, H, W, Cin, Cout):
        return None
    rows = torch.empty((N * (H // 16) * (W // 32), Cout, 2), dtype=torch.float32, device=dz.buf.device)
    check(lib().unet_conv_dgrad_bnred(dz.dt, _p(dz.buf), dz.ld, dz.off, _p(w_dgrad), _p(dy.buf), dy.ld, dy.off, _p(rows),
                                      _p(z.buf), z.ld, z.off, _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd),
                                      N, H, W, Cin, Cout, _stream()), "conv_dgrad_bnred")
    return rows


def bn_relu_bwd(dy: Act, z: Act, dz: Act, st: BNState, ws: Workspace, relu=True, want_dbias=True,
                sum_rows: Optional[torch.Tensor] = None):
    """-> (dgamma, dbeta, dbias|None); dz is written.  sum_rows: the first pass's sums as per-tile rows from conv_dgrad_bnred."""
    dy.check(); z.check(); dz.check()
    C = z.C
    if dy.C != C or dz.C != C or dy.M != z.M or dz.M != z.M or dy.dt != z.dt or dz.dt != z.dt:
        raise UNetHipError("bn_relu_bwd: shape mismatch")
    if 3 * C + 2 > ws.acc.numel() or 2 * C > ws.c1c2.numel():
        raise UNetHipError("bn_relu_bwd: workspace too small")
    g = torch.empty((3, C), dtype=torch.float32, device=z.buf.device)
    acc = ws.take_zeroed(3 * C + 2)
    if sum_rows is not None:
        if not relu or sum_rows.dtype != torch.float32 or not sum_rows.is_contiguous() or tuple(sum_rows.shape[1:]) != (C, 2):
            raise UNetHipError("bn_relu_bwd: sum_rows must be contiguous fp32 [nrows, C, 2] of a BatchNorm + ReLU")
        check(lib().unet_bn_relu_bwd_rows(z.dt, _bn_bwd_flags(st, relu) | (4 if acc is not None else 0), _p(dy.buf), dy.ld, dy.off,
                                          _p(z.buf), z.ld, z.off, _p(dz.buf), dz.ld, dz.off,
                                          _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd),
                                          _p(g[0]), _p(g[1]), _p(g[2]) if want_dbias else None, _p(sum_rows), sum_rows.shape[0],
                                          _p(acc if acc is not None else ws.acc), z.M, C, _stream()), "bn_relu_bwd_rows")
        return g[0], g[1], (g[2] if want_dbias else None)
    check(lib().unet_bn_relu_bwd(z.dt, _bn_bwd_flags(st, relu) | (4 if acc is not None else 0), _p(dy.buf), dy.ld, dy.off,
                                 _p(z.buf), z.ld, z.off,
                                 _p(dz.buf), dz.ld, dz.off, _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd),
                                 _p(g[0]), _p(g[1]), _p(g[2]) if want_dbias else None, _p(ws.c1c2), _p(acc if acc is not None else ws.acc),
                                 z.M, C, _stream()), "bn_relu_bwd")
    return g[0], g[1], (g[2] if want_dbias else None)


@dataclass
class PoolGrad:
    """dy of an encoder level's output, not materialised: dskip + unpool(dpool, code)."""
    dpool: "Act"
    code: torch.Tensor
    dskip: "Act"


def bn_relu_bwd_pool(src: PoolGrad, z: Act, dz: Act, st: BNState, ws: Workspace, relu=True, want_dbias=True):
    """bn_relu_bwd with dy = src.dskip + unpool(src.dpool, src.code) evaluated inside the two passes."""
    dp, code, dskip = src.dpool, src.code, src.dskip
    dp.check(); dskip.check(); z.check(); dz.check()
    C = z.C
    N, H, W = z.N, z.H, z.W
    if dp.off != 0 or dp.ld != dp.C or (dp.N, dp.H, dp.W, dp.C) != (N, H // 2, W // 2, C) or H % 2 or W % 2:
        raise UNetHipError("bn_relu_bwd_pool: pooled gradient must be dense [N,H/2,W/2,C]")
    if tuple(code.shape) != (N, H // 2, W // 2, C) or code.dtype != torch.uint8 or not code.is_contiguous():
        raise UNetHipError("bn_relu_bwd_pool: code shape mismatch")
    if (dskip.N, dskip.H, dskip.W, dskip.C) != (N, H, W, C) or (dz.N, dz.H, dz.W, dz.C) != (N, H, W, C):
        raise UNetHipError("bn_relu_bwd_pool: shape mismatch")
    if not (dp.dt == dskip.dt == z.dt == dz.dt):
        raise UNetHipError("bn_relu_bwd_pool: dtype mismatch")
    if 3 * C + 2 > ws.acc.numel() or 2 * C > ws.c1c2.numel():
        raise UNetHipError("bn_relu_bwd_pool: workspace too small")
    g = torch.empty((3, C), dtype=torch.float32, device=z.buf.device)
    acc = ws.take_zeroed(3 * C + 2)
    check(lib().unet_bn_relu_bwd_pool(z.dt, _bn_bwd_flags(st, relu) | (4 if acc is not None else 0), _p(dp.buf), _p(code), _p(dskip.buf), dskip.ld, dskip.off,
                                      _p(z.buf), z.ld, z.off, _p(dz.buf), dz.ld, dz.off,
                                      _p(st.scale), _p(st.shift), _p(st.mean), _p(st.invstd),
                                      _p(g[0]), _p(g[1]), _p(g[2]) if want_dbias else None, _p(ws.c1c2),
                                      _p(acc if acc is not None else ws.acc), N, H, W, C, _stream()), "bn_relu_bwd_pool")
    return g[0], g[1], (g[2] if want_dbias else None)


def relu_bwd(dy: Act, y: Act, dz: Act):
    dy.check(); y.check(); dz.check()
    check(lib().unet_relu_bwd(y.dt, _p(dy.buf), dy.ld, dy.off, _p(y.buf), y.ld, y.off, _p(dz.buf), dz.ld, dz.off,
                              y.M, y.C, _stream()), "relu_bwd")


def channel_sum(x: Act, ws: Workspace, out: Optional[torch.Tensor] = None, accumulate=False) -> torch.Tensor:
    x.check()
    if out is None:
        out = torch.empty((x.C,), dtype=torch.float32, device=x.buf.device)
    if x.C > ws.acc.numel() or out.numel() != x.C:
        raise UNetHipError("channel_sum: workspace or output too small")
    check(lib().unet_channel_sum(x.dt, _p(x.buf), x.ld, x.off, x.M, x.C, _p(out), 1 if accumulate else 0, _p(ws.acc), _stream()),
          "channel_sum")
    return out


def rows_colsum(rows: torch.Tensor, off: int, C: int, stride: int) -> torch.Tensor:
    """out[c] = sum_r rows2d[r, off + c*stride] for fp32 partial rows [nrows, ...] (deterministic, no workspace)."""
    _need_cuda(rows)
    if rows.dtype != torch.float32 or not rows.is_contiguous() or rows.dim() < 2:
        raise UNetHipError("rows_colsum: rows must be a contiguous fp32 tensor [nrows, ...]")
    nrows, ld = rows.shape[0], rows.numel() // rows.shape[0]
    out = torch.empty((C,), dtype=torch.float32, device=rows.device)
    check(lib().unet_rows_colsum(_p(rows), nrows, ld, off, C, stride, _p(out), _stream()), "rows_colsum")
    return out


def maxpool_fwd(x: Act):
    """-> (pooled dense Act, uint8 code tensor [N,H//2,W//2,C]); odd sizes floor, as nn.MaxPool2d(2, 2)."""
    x.check()
    if x.H < 2 or x.W < 2:
        raise UNetHipError(f"maxpool: H and W must be at least 2, got {x.H}x{x.W}")
    y = dense(x.N, x.H // 2, x.W // 2, x.C, x.dt, x.buf.device)
    code = torch.empty((x.N, x.H // 2, x.W // 2, x.C), dtype=torch.uint8, device=x.buf.device)
    check(lib().unet_maxpool2x2_fwd(x.dt, _p(x.buf), x.ld, x.off, _p(y.buf), _p(code), x.N, x.H, x.W, x.C, _stream()), "maxpool_fwd")
    return y, code


def maxpool_bwd(dy: Act, code: torch.Tensor, dskip: Optional[Act], out_hw=None) -> Act:
    """dx[N,H,W,C] = unpool(dy) (+ dskip).  (H, W) = dskip's, else out_hw, else (2*Ho, 2*Wo); an odd H or W leaves its
    last row / column outside every window (gradient = dskip's alone)."""
    dy.check()
    if dy.off != 0 or dy.ld != dy.C:
        raise UNetHipError("maxpool_bwd: pooled gradient must be dense")
    N, Ho, Wo, C = dy.N, dy.H, dy.W, dy.C
    if tuple(code.shape) != (N, Ho, Wo, C):
        raise UNetHipError("maxpool_bwd: code shape mismatch")
    H, W = (dskip.H, dskip.W) if dskip is not None else (out_hw if out_hw is not None else (2 * Ho, 2 * Wo))
    if H // 2 != Ho or W // 2 != Wo:
        raise UNetHipError("maxpool_bwd: output size does not pool to the gradient's size")
    if dskip is not None:
        dskip.check()
        if (dskip.N, dskip.C) != (N, C):
            raise UNetHipError("maxpool_bwd: skip gradient shape mismatch")
    dx = dense(N, H, W, C, dy.dt, dy.buf.device)
    check(lib().unet_maxpool2x2_bwd(dy.dt, _p(dy.buf), _p(code), _p(dskip.buf) if dskip is not None else None,
                                    dskip.ld if dskip is not None else 0, dskip.off if dskip is not None else 0,
                                    _p(dx.buf), N, H, W, C, _stream()), "maxpool_bwd")
    return dx


def resize_bilinear(x: Act, y: Act):
    """y = F.interpolate(x, size=(y.H, y.W), mode="bilinear", align_corners=True) (models/unet.py:56-57), NHWC slices."""
    x.check(); y.check()
    if (x.N, x.C, x.dt) != (y.N, y.C, y.dt):
        raise UNetHipError("resize_bilinear: batch / channel / dtype mismatch")
    check(lib().unet_resize_bilinear_fwd(x.dt, _p(x.buf), x.ld, x.off, x.H, x.W, _p(y.buf), y.ld, y.off, y.H, y.W, x.N, x.C, _stream()),
          "resize_bilinear_fwd")


def resize_bilinear_bwd(dy: Act, dx: Act):
    """dx = adjoint of resize_bilinear applied to dy (dx has the small shape)."""
    dy.check(); dx.check()
    if (dx.N, dx.C, dx.dt) != (dy.N, dy.C, dy.dt):
        raise UNetHipError("resize_bilinear_bwd: batch / channel / dtype mismatch")
    check(lib().unet_resize_bilinear_bwd(dy.dt, _p(dy.buf), dy.ld, dy.off, dy.H, dy.W, _p(dx.buf), dx.ld, dx.off, dx.H, dx.W, dx.N, dx.C,
                                         _stream()), "resize_bilinear_bwd")


def pool_indices_int64(code: torch.Tensor, W: int) -> torch.Tensor:
    """2-bit code -> torch's int64 flat index (h*W + w) into the input plane, NCHW order."""
    N, Ho, Wo, C = code.shape
    c = code.to(torch.int64)
    ho = torch.arange(Ho, device=code.device).view(1, Ho, 1, 1)
    wo = torch.arange(Wo, device=code.device).view(1, 1, Wo, 1)
    idx = (2 * ho + (c >> 1)) * W + 2 * wo + (c & 1)
    return idx.permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------- loss & friends

def dice_ce_fwd(logits: torch.Tensor, target: torch.Tensor, ce_w, dice_w, ignore_index, ce_only=False, eps=1e-6):
    _need_cuda(logits, target)
    if logits.dim() != 4 or logits.dtype != torch.float32 or not logits.is_contiguous():
        raise UNetHipError("dice_ce: logits must be contiguous fp32 NCHW")
    N, C, H, W = logits.shape
    if target.dtype != torch.int64 or tuple(target.shape) != (N, H, W) or not target.is_contiguous():
        raise UNetHipError("dice_ce: target must be contiguous int64 [N,H,W]")
    acc = torch.empty(N * C * 3 + 3, dtype=torch.float64, device=logits.device)
    out4 = torch.empty(4, dtype=torch.float32, device=logits.device)
    check(lib().unet_dice_ce_fwd(_p(logits), _p(target), N, C, H * W, float(ce_w), float(dice_w), int(ignore_index),
                                 1 if ce_only else 0, float(eps), _p(acc), _p(out4), _stream()), "dice_ce_fwd")
    return out4, acc


def dice_ce_bwd(logits, target, acc, grad_out, ce_w, dice_w, ignore_index, ce_only=False, eps=1e-6):
    N, C, H, W = logits.shape
    dl = torch.empty_like(logits)
    if grad_out is not None:
        grad_out = grad_out.reshape(1).float().contiguous()
    check(lib().unet_dice_ce_bwd(_p(logits), _p(target), N, C, H * W, float(ce_w), float(dice_w), int(ignore_index),
                                 1 if ce_only else 0, float(eps), _p(acc), _p(grad_out), _p(dl), _stream()), "dice_ce_bwd")
    return dl


def argmax(logits: torch.Tensor) -> torch.Tensor:
    _need_cuda(logits)
    if logits.dim() != 4 or logits.dtype != torch.float32 or not logits.is_contiguous():
        raise UNetHipError("argmax: logits must be contiguous fp32 NCHW")
    N, C, H, W = logits.shape
    out = torch.empty((N, H, W), dtype=torch.int64, device=logits.device)
    check(lib().unet_argmax(_p(logits), _p(out), N, C, H * W, _stream()), "argmax")
    return out


def ensemble_fuse(l1: torch.Tensor, l2: torch.Tensor, thresh: float = 0.6) -> torch.Tensor:
    _need_cuda(l1, l2)
    if l1.shape != l2.shape or l1.dtype != torch.float32 or l2.dtype != torch.float32:
        raise UNetHipError("ensemble_fuse: two fp32 NCHW logits tensors of equal shape expected")
    l1, l2 = l1.contiguous(), l2.contiguous()
    N, C, H, W = l1.shape
    out = torch.empty((N, H, W), dtype=torch.int64, device=l1.device)
    check(lib().unet_ensemble_fuse(_p(l1), _p(l2), _p(out), N, C, H * W, float(thresh), _stream()), "ensemble_fuse")
    return out


def seg_counts(pred: torch.Tensor, target: torch.Tensor, K: int) -> torch.Tensor:
    _need_cuda(pred, target)
    if pred.dtype != torch.int64 or target.dtype != torch.int64 or pred.shape != target.shape:
        raise UNetHipError("seg_counts: int64 tensors of equal shape expected")
    pred, target = pred.contiguous(), target.contiguous()
    counts = torch.empty(66, dtype=torch.int64, device=pred.device)
    check(lib().unet_seg_counts(_p(pred), _p(target), pred.numel(), K, _p(counts), _stream()), "seg_counts")
    return counts


# ----------------------------------------------------------------------------- RepUNet

@dataclass
class GNState:
    scale: torch.Tensor      # [N][C]
    shift: torch.Tensor      # [N][C]
    mean: torch.Tensor       # [N][groups]
    invstd: torch.Tensor
    groups: int


def gn_finalize(rows: torch.Tensor, N: int, HW: int, gamma, beta, groups: int, eps: float = 1e-5) -> GNState:
    """per-image conv statistics rows [N*rpi][C][2] -> GroupNorm coefficients."""
    nrows, C, _ = rows.shape
    if nrows % N:
        raise UNetHipError("gn_finalize: statistics rows are not grouped per image")
    dev = rows.device
    coef = torch.empty((2, N, C), dtype=torch.float32, device=dev)
    ms = torch.empty((2, N, groups), dtype=torch.float32, device=dev)
    check(lib().unet_gn_finalize(_p(rows), N, nrows // N, C, groups, HW, _p(gamma), _p(beta), float(eps),
                                 _p(coef[0]), _p(coef[1]), _p(ms[0]), _p(ms[1]), _stream()), "gn_finalize")
    return GNState(coef[0], coef[1], ms[0], ms[1], groups)


def branch_sum_act(zs, scales, shifts, per_sample: bool, y: Act, relu=True):
    """y = act(sum_b z_b*scale_b + shift_b); z_b dense Acts of y's shape."""
    import ctypes
    y.check()
    nb = len(zs)
    for z in zs:
        z.check()
        if z.off != 0 or z.ld != z.C or z.C != y.C or z.M != y.M or z.dt != y.dt:
            raise UNetHipError("branch_sum_act: branch tensors must be dense and shaped like the output")
    C = y.C
    for s in list(scales) + list(shifts):
        if s.dtype != torch.float32 or s.numel() != (y.N * C if per_sample else C):
            raise UNetHipError("branch_sum_act: coefficient shape mismatch")
    PA = ctypes.c_void_p * nb
    check(lib().unet_branch_sum_act(y.dt, 1 if relu else 0, nb, PA(*[z.buf.data_ptr() for z in zs]),
                                    PA(*[s.data_ptr() for s in scales]), PA(*[s.data_ptr() for s in shifts]),
                                    C if per_sample else 0, _p(y.buf), y.ld, y.off, y.N, y.H * y.W, C, _stream()),
          "branch_sum_act")


def gn_bwd(u: Act, z: Act, st: GNState, gamma: torch.Tensor):
    """-> (dz dense Act, dgamma, dbeta)."""
    u.check(); z.check()
    if u.off or z.off or u.ld != u.C or z.ld != z.C or u.C != z.C or u.M != z.M:
        raise UNetHipError("gn_bwd: dense tensors of equal shape expected")
    N, C, dev = z.N, z.C, z.buf.device
    dz = dense(z.N, z.H, z.W, C, z.dt, dev)
    g = torch.empty((2, C), dtype=torch.float32, device=dev)
    acc = torch.empty(N * C * 2 + C * 2, dtype=torch.float64, device=dev)
    coef = torch.empty(3 * N * C, dtype=torch.float32, device=dev)
    check(lib().unet_gn_bwd(z.dt, _p(u.buf), _p(z.buf), _p(dz.buf), N, z.H * z.W, C, st.groups, _p(gamma), _p(st.mean),
                            _p(st.invstd), _p(g[0]), _p(g[1]), _p(acc), _p(coef), _stream()), "gn_bwd")
    return dz, g[0], g[1]


GN_MULTI_MAX = 8


def gn_bwd_multi(u: Act, zs, sts, gammas):
    """GroupNorm backward of the branches of one block (they share u) -> [(dz dense Act, dgamma, dbeta), ...]."""
    import ctypes
    nb = len(zs)
    if not 1 <= nb <= GN_MULTI_MAX:
        raise UNetHipError(f"gn_bwd_multi: 1..{GN_MULTI_MAX} branches")
    u.check()
    z0 = zs[0]
    for z in zs:
        z.check()
        if u.off or z.off or u.ld != u.C or z.ld != z.C or u.C != z.C or u.M != z.M or z.dt != u.dt:
            raise UNetHipError("gn_bwd_multi: dense tensors of equal shape expected")
    groups = sts[0].groups
    if any(st.groups != groups for st in sts):
        raise UNetHipError("gn_bwd_multi: branches must share the group count")
    N, C, dev = z0.N, z0.C, z0.buf.device
    dzs = [dense(z0.N, z0.H, z0.W, C, z0.dt, dev) for _ in zs]
    g = torch.empty((nb, 2, C), dtype=torch.float32, device=dev)
    acc = torch.empty(nb * (N * C * 2 + C * 2), dtype=torch.float64, device=dev)
    coef = torch.empty(nb * 3 * N * C, dtype=torch.float32, device=dev)
    PA = ctypes.c_void_p * nb
    check(lib().unet_gn_bwd_multi(z0.dt, nb, _p(u.buf), PA(*[z.buf.data_ptr() for z in zs]), PA(*[d.buf.data_ptr() for d in dzs]),
                                  N, z0.H * z0.W, C, groups, PA(*[t.data_ptr() for t in gammas]),
                                  PA(*[st.mean.data_ptr() for st in sts]), PA(*[st.invstd.data_ptr() for st in sts]),
                                  PA(*[g[b, 0].data_ptr() for b in range(nb)]), PA(*[g[b, 1].data_ptr() for b in range(nb)]),
                                  _p(acc), _p(coef), _stream()), "gn_bwd_multi")
    return [(dzs[b], g[b, 0], g[b, 1]) for b in range(nb)]


def fold_asymm(k, w_sq, w_ve, w_ho, bn_sq, bn_ve, bn_ho, eps=1e-5, sqrt="ieee"):
    """One AsymmConvBlock -> (weight (O,I,k,k), bias (O)); bn_* = (weight, bias, running_mean, running_var).
    sqrt="ieee": std = correctly rounded sqrt(running_var + eps) on the device (reproducible everywhere);
    sqrt="reference": std is computed the way the reference computes it -- torch's CPU sqrt on this host (three vectors
    of <= 1024 floats cross PCIe) -- so the folded tensors are torch.equal to what the reference's own
    switch_to_deploy() produces on this host.
    sqrt=callable: std = sqrt(cpu fp32 tensor running_var + eps) from the caller (e.g. another host's torch sqrt)."""
    ts = [w_sq, w_ve, w_ho, *bn_sq, *bn_ve, *bn_ho]
    _need_cuda(*ts)
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise UNetHipError("fold: contiguous fp32 tensors expected")
    O, I = w_sq.shape[0], w_sq.shape[1]
    if tuple(w_sq.shape) != (O, I, k, k) or tuple(w_ve.shape) != (O, I, k, 1) or tuple(w_ho.shape) != (O, I, 1, k):
        raise UNetHipError("fold: branch kernel shapes do not match")
    out_w = torch.empty((O, I, k, k), dtype=torch.float32, device=w_sq.device)
    out_b = torch.empty((O,), dtype=torch.float32, device=w_sq.device)
    if sqrt == "reference" or callable(sqrt):
        fn = torch.sqrt if sqrt == "reference" else sqrt
        stds = [fn(bn[3].cpu() + eps).to(device=w_sq.device, dtype=torch.float32).contiguous()
                for bn in (bn_sq, bn_ve, bn_ho)]                                                   # mergeBN: (running_var + eps).sqrt()
        for sd in stds:
            if sd.numel() != O:
                raise UNetHipError("fold: the sqrt callable must return one value per channel")
        args = []
        for bn, sd in zip((bn_sq, bn_ve, bn_ho), stds):
            args += [_p(t) for t in bn] + [_p(sd)]
        check(lib().unet_fold_asymm_std(k, O, I, _p(w_sq), _p(w_ve), _p(w_ho), *args, _p(out_w), _p(out_b), _stream()), "fold_asymm_std")
        return out_w, out_b
    if sqrt != "ieee":
        raise UNetHipError(f"fold: sqrt must be 'ieee' or 'reference', got {sqrt!r}")
    check(lib().unet_fold_asymm(k, O, I, _p(w_sq), _p(w_ve), _p(w_ho), *[_p(t) for t in bn_sq], *[_p(t) for t in bn_ve],
                                *[_p(t) for t in bn_ho], float(eps), _p(out_w), _p(out_b), _stream()), "fold_asymm")
    return out_w, out_b


def fold_embed_add(target_w, src_w, target_b, src_b):
    K, k = target_w.shape[-1], src_w.shape[-1]
    O, I = target_w.shape[0], target_w.shape[1]
    if tuple(src_w.shape) != (O, I, k, k) or tuple(target_w.shape) != (O, I, K, K):
        raise UNetHipError("fold: cannot embed, shape mismatch")
    check(lib().unet_fold_embed_add(K, k, O, I, _p(target_w), _p(src_w), _p(target_b), _p(src_b), _stream()), "fold_embed_add")
