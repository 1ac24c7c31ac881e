"""RepUNet schedule on the HIP kernels: multi-branch training path and deployed single-conv path.

Reference: /root/reference/models/reparm_unet.py -- AsymmConvBlock (:39-72), MultiConvBlock (:96-122),
ReparamDoubleConv (:150-170), ReparamUNet.forward (:194-215; same wiring as UNet).

One "stage" = MultiConvBlock + ReLU:
  train  : for k in sorted(kernel_sizes): z_ve = conv_kx1(x), z_ho = conv_1xk(x), z_sq = conv_kxk(x)
           (bias-free implicit-GEMM convs, statistics from the epilogue) -> per-branch GroupNorm (as shipped)
           or BatchNorm(momentum 0.01) coefficients -> ONE kernel y = relu(sum_b z_b*scale_b + shift_b).
  deploy : y = relu(conv_KxK(x) + bias) in the conv epilogue (K = max kernel size).
"""
from typing import Dict, List

import torch

from . import ops
from .engine import UNetEngine, _Saved
from .ops import Act

EPS = 1e-5
REP_BN_MOMENTUM = 0.01


def _branches(k):
    return [("convVe1", "bnVe1", k, 1), ("convHo1", "bnHo1", 1, k), ("convSq", "bnSq", k, k)]


class RepUNetEngine(UNetEngine):
    def __init__(self, features, in_channels, n_classes, kernel_sizes, norm):
        super().__init__(features, in_channels, n_classes)
        self.fused_pool_bwd = False     # the multi-branch blocks take a materialised dy
        self.fused_pool_fwd = False
        self.ks = sorted(kernel_sizes)
        self.norm = norm
        for k in self.ks:
            if k not in (1, 3, 5):
                raise ValueError(f"unet_amd RepUNet kernels are built for kernel sizes 1, 3, 5 (got {kernel_sizes})")

    # ------------------------------------------------------------------ forward
    def _stage_fwd(self, P, B, sp, xin, yout, dt, train, save, x_f32=None):
        dev = (x_f32 if x_f32 is not None else xin.buf).device
        ws = self.ws(dev)
        if x_f32 is not None:
            N, H, W, Cin = x_f32.shape
        else:
            N, H, W, Cin = xin.N, xin.H, xin.W, xin.C
        mw = sp + ".mergedConv.weight"
        if mw in P:                                              # deployed: one conv + bias + ReLU
            w, b = P[mw], P[sp + ".mergedConv.bias"]
            Cout, K = w.shape[0], w.shape[-1]
            y = yout if yout is not None else ops.dense(N, H, W, Cout, dt, dev)
            if x_f32 is not None:
                z = ops.dense(N, H, W, Cout, dt, dev)
                ops.conv_first_fwd(x_f32, w.detach(), b.detach(), z, K, K, stats=False)
                one = ops.BNState(torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev))
                ops.affine_act(z, y, one, relu=True)
            else:
                ops.conv_fwd(xin, self.packed(mw, w, 0, dt), b.detach(), y, K, K, Cin, Cout, relu=True)
            if save:
                raise ops.UNetHipError("deployed RepUNet is inference-only here: call it under torch.no_grad()")
            return y, None
        zs, sts, names = [], [], []
        Cout = None
        for i, k in enumerate(self.ks):
            for conv, bn, kh, kw in _branches(k):
                wname = f"{sp}.conv_list.{i}.{conv}.weight"
                bnp = f"{sp}.conv_list.{i}.{bn}"
                w = P[wname]
                Cout = w.shape[0]
                z = ops.dense(N, H, W, Cout, dt, dev)
                if x_f32 is not None:
                    rows = ops.conv_first_fwd(x_f32, w.detach(), None, z, kh, kw, stats=True)
                else:
                    rows = ops.conv_fwd(xin, self.packed(wname, w, 0, dt), None, z, kh, kw, Cin, Cout, stats=True,
                                        per_image_stats=self.norm == "group")
                gamma, beta = P[bnp + ".weight"].detach(), P[bnp + ".bias"].detach()
                if self.norm == "group":
                    st = ops.gn_finalize(rows, N, H * W, gamma, beta, min(8, Cout), EPS)
                elif train:
                    st = ops.bn_finalize(rows, N * H * W, gamma, beta, B[bnp + ".running_mean"], B[bnp + ".running_var"],
                                         B[bnp + ".num_batches_tracked"], REP_BN_MOMENTUM, EPS, ws)
                else:
                    st = ops.bn_eval_coeffs(gamma, beta, B[bnp + ".running_mean"], B[bnp + ".running_var"], EPS, for_backward=save)
                zs.append(z); sts.append(st); names.append((wname, bnp, kh, kw))
        y = yout if yout is not None else ops.dense(N, H, W, Cout, dt, dev)
        ops.branch_sum_act(zs, [s.scale for s in sts], [s.shift for s in sts], self.norm == "group", y, relu=True)
        rec = None
        if save:
            rec = _Saved()
            rec.xin, rec.x_f32, rec.zs, rec.sts, rec.names, rec.y = xin, x_f32, zs, sts, names, y
        return y, rec

    def block_fwd(self, P, B, prefix, xin, yout, dt, train, save, x_f32=None):
        y1, r1 = self._stage_fwd(P, B, prefix + ".reparam_conv.0", xin, None, dt, train, save, x_f32=x_f32)
        y2, r2 = self._stage_fwd(P, B, prefix + ".reparam_conv.2", y1, yout, dt, train, save)
        return y2, ((r1, r2) if save else None)

    def block_param_names(self, prefix):
        out = []
        for s in (0, 2):
            for i, k in enumerate(self.ks):
                for conv, bn, _, _ in _branches(k):
                    base = f"{prefix}.reparam_conv.{s}.conv_list.{i}"
                    out += [f"{base}.{conv}.weight", f"{base}.{bn}.weight", f"{base}.{bn}.bias"]
        return out

    # ------------------------------------------------------------------ backward
    def _stage_bwd(self, P, rec, dy: Act, grads, dt, need_dx, split_f):
        dev = dy.buf.device
        ws = self.ws(dev)
        y = rec.y
        u = ops.dense(y.N, y.H, y.W, y.C, dt, dev)
        ops.relu_bwd(dy, y, u)
        first = rec.x_f32 is not None
        outs = None
        if need_dx and not first:
            Cin = rec.xin.C
            if split_f:
                outs = (ops.dense(y.N, y.H, y.W, split_f, dt, dev), ops.dense(y.N, y.H, y.W, Cin - split_f, dt, dev))
            else:
                outs = (ops.dense(y.N, y.H, y.W, Cin, dt, dev),)
        gn = None
        if self.norm == "group":                # every branch's GroupNorm backward in one reduction + one apply pass (they share u)
            gammas = [P[bnp + ".weight"].detach() for _, bnp, _, _ in rec.names]
            if len(rec.zs) <= ops.GN_MULTI_MAX:
                gn = ops.gn_bwd_multi(u, rec.zs, rec.sts, gammas)
            else:
                gn = [ops.gn_bwd(u, z, st, g_) for z, st, g_ in zip(rec.zs, rec.sts, gammas)]
        for bi, (z, st, (wname, bnp, kh, kw)) in enumerate(zip(rec.zs, rec.sts, rec.names)):
            gamma = P[bnp + ".weight"].detach()
            if gn is not None:
                dz, dg, db = gn[bi]
            else:
                dz = ops.dense(z.N, z.H, z.W, z.C, dt, dev)
                dg, db, _ = ops.bn_relu_bwd(u, z, dz, st, ws, relu=False, want_dbias=False)
            grads[bnp + ".weight"], grads[bnp + ".bias"] = dg, db
            w = P[wname]
            if first:
                grads[wname] = ops.conv_first_wgrad(dz, rec.x_f32, kh, kw, ws)
                continue
            if kh * kw <= 9:
                grads[wname] = ops.conv_wgrad(dz, rec.xin, kh, kw, ws)[0].view_as(w)
            else:
                dw = torch.empty_like(w, dtype=torch.float32)
                r = 0
                while r < kh:               # bf16: two kernel rows per launch (2x5 instantiation), the odd row alone
                    nr = 2 if (dt == ops.BF16 and kw == 5 and r + 1 < kh) else 1
                    ops.conv_wgrad(dz, rec.xin, nr, kw, ws, dw=dw, pad_h=kh // 2 - r, pad_w=kw // 2, dw_taps=kh * kw, dw_tap0=r * kw)
                    r += nr
                grads[wname] = dw
            if outs is not None:
                wd = self.packed(wname, w, 1, dt)
                if split_f:
                    ops.conv_fwd(dz, wd, None, outs[0], kh, kw, w.shape[0], w.shape[1], y2=outs[1], csplit=split_f, accumulate=bi > 0)
                else:
                    ops.conv_fwd(dz, wd, None, outs[0], kh, kw, w.shape[0], w.shape[1], accumulate=bi > 0)
        if outs is None:
            return None
        return outs if split_f else outs[0]

    def block_bwd(self, P, sv, prefix, dy: Act, grads: Dict[str, torch.Tensor], dt, need_dx=True, split_f: int = 0):
        r1, r2 = sv.blocks[prefix]
        dy1 = self._stage_bwd(P, r2, dy, grads, dt, True, 0)
        return self._stage_bwd(P, r1, dy1, grads, dt, need_dx, split_f)
