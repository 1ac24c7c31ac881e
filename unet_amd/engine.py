"""Forward / backward schedule of the U-Net on the HIP kernels (one autograd node per model call).

Follows the reference wiring (/root/reference/models/unet.py:41-62): 4x [DoubleConv, MaxPool] ->
bottleneck -> 4x [ConvTranspose2d, cat(skip, up), DoubleConv] -> 1x1 conv.  Differences that are
invisible to the caller:
  * activations are NHWC in the compute dtype (bf16 under autocast / amp, else fp32);
  * torch.cat is designed away: the encoder's second BN+ReLU and the transposed conv both write into
    one pre-allocated 2f-channel buffer (skip first, as models/unet.py:59);
  * BN batch statistics come from the conv epilogue; pool saves a 2-bit argmax code;
  * backward is an explicit schedule; parameter gradients are produced in backward-arrival order and
    handed to `grad_hook(names, tensors)` as soon as they are enqueued (data-parallel bucketing).
"""
from typing import Callable, Dict, List, Optional

import os

import torch

from . import ops
from .ops import Act, Workspace

BN_EPS = 1e-5


class _Saved:
    pass


def _double_conv_names(prefix):
    return [f"{prefix}.double_conv.{i}.{k}" for i, ks in ((0, ("weight", "bias")), (1, ("weight", "bias")),
                                                          (3, ("weight", "bias")), (4, ("weight", "bias"))) for k in ks]


class UNetEngine:
    """Executes UNet forward/backward for a module that owns reference-layout parameters."""

    def __init__(self, features, in_channels, n_classes):
        self.features = list(features)
        self.in_channels = in_channels
        self.n_classes = n_classes
        self._ws: Dict[torch.device, Workspace] = {}
        self._packed = {}       # (name, mode, dt) -> (version, tensor)
        self._folded = {}       # (conv weight name, dt) -> (versions, packed folded weights, folded bias)
        self._stats_gen = 0     # bumped by every training forward (running statistics change under raw pointers)
        self.fold_bn_eval = True
        self.fused_pool_bwd = True   # UNet: dy of an encoder level is never stored (ops.PoolGrad)
        self.fused_pool_fwd = True   # UNet: max-pool fused into the BN+ReLU pass of the level's second conv
        self._pooled = None
        # The two ends of the network fused into their BatchNorm passes (csrc/edge_fused.hip): the first conv's output and the
        # last activation / its gradient are recomputed inside the bandwidth-bound passes instead of stored (plain UNet,
        # in_channels = 1, <= 4 classes; anything else takes the separate kernels).  False: the separate kernels (tests).
        self.fuse_edges = os.environ.get("UNET_AMD_FUSE_EDGES", "1") != "0"
        # Batch statistics of the fused first layer from the image's 3x3 patch moments (z is linear in the patch): 73 us instead
        # of 257 us at 32x512x512, and the backward reuses the moments.  UNET_AMD_FIRST_STATS_CONV=1: the statistics-only conv pass.
        self.first_stats_moments = os.environ.get("UNET_AMD_FIRST_STATS_CONV", "0") != "1"
        # First pass of the BatchNorm backward of a DoubleConv's FIRST norm inside the dgrad of its second conv (ops.conv_dgrad_bnred).
        # Correct and tested (tests/test_gpu_realwidth.py), but MEASURED NOT TO PAY (round 3, same box, per step): the eight
        # reduction passes it removes cost 1.05 ms, the z reads + sums + per-tile cross-wave reduction it adds to the epilogues of
        # eight dgrad launches cost 1.2 ms (the epilogue is the one part of those kernels no MFMA overlaps).  Off by default.
        self.bn_reduce_in_dgrad = os.environ.get("UNET_AMD_BN_REDUCE_IN_DGRAD", "0") == "1"
        self.keep_saved = False      # tests: leave the last forward's saved activations in self.last_saved
        self.last_saved = None
        # Weight gradients on a side stream (wgrad_async).  Measured: -0.4...-0.7 ms per step (1-1.5 %), all of it tail
        # filling between the big persistent kernels (a wgrad block starts on a CU the moment the dgrad conv's block
        # there retires; the two cannot share a CU, 134 + 151 KB of LDS).  Results are bit-identical (tested).  ON by
        # default; per-kernel durations then include time queued behind the other stream's kernel, so measurements that
        # want clean per-kernel numbers (bench.py's one instrumented step, the rocprof profiles) switch it off:
        # UNET_AMD_OVERLAP_WGRAD=0 or engine.overlap_wgrad = False.
        self.overlap_wgrad = os.environ.get("UNET_AMD_OVERLAP_WGRAD", "1") != "0"
        self._side_streams = {}
        self._side_busy = False

    # ------------------------------------------------------------------ helpers
    def ws(self, device) -> Workspace:
        w = self._ws.get(device)
        if w is None:
            w = self._ws[device] = Workspace(device)
        return w

    def _side(self, device):
        st = self._side_streams.get(device)
        if st is None:
            st = self._side_streams[device] = (torch.cuda.Stream(device), Workspace(device))
        return st

    def wgrad_async(self, device, fn, tensors):
        """Run fn(workspace) -- a weight-gradient kernel sequence -- on the side stream, ordered after everything
        queued so far on the current stream.  Weight gradients are off the backward critical path
        (dz -> dgrad -> BN backward -> dz ...), MFMA-bound, and leave registers and LDS for a second resident kernel,
        so they overlap with the HBM-bound BN / pooling kernels of the next layer.  join_side() orders the current
        stream after them."""
        if not self.overlap_wgrad:
            return fn(self.ws(device))
        main = torch.cuda.current_stream(device)
        side, ws_side = self._side(device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out = fn(ws_side)
        for t in tensors:
            t.record_stream(side)            # the caching allocator must not hand these out again before `side` is done
        self._side_busy = True
        return out

    def join_side(self, device):
        if self._side_busy:
            torch.cuda.current_stream(device).wait_stream(self._side(device)[0])
            self._side_busy = False

    def packed(self, name, p: torch.Tensor, mode, dt):
        key = (name, mode, dt)
        ver = (p._version, p.data_ptr())
        hit = self._packed.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        t = ops.pack_weights(p, mode, dt)
        self._packed[key] = (ver, t)
        return t

    def folded(self, wname, bname, bnp, P, B, dt):
        """Packed conv weights and bias with the eval-mode BatchNorm folded in (inference path)."""
        ts = [P[wname], P[bname], P[bnp + ".weight"], P[bnp + ".bias"], B[bnp + ".running_mean"], B[bnp + ".running_var"]]
        ver = tuple((t._version, t.data_ptr()) for t in ts) + (self._stats_gen,)
        hit = self._folded.get((wname, dt))
        if hit is not None and hit[0] == ver:
            return hit[1], hit[2]
        wf, bf = ops.fold_bn_conv(ts[0], ts[1].detach(), ts[2].detach(), ts[3].detach(), ts[4], ts[5], BN_EPS)
        wp = ops.pack_weights(wf, 0, dt)
        self._folded[(wname, dt)] = (ver, wp, bf)
        return wp, bf

    def prepack(self, P, dt):
        """Refresh, in ONE launch, every packed image this engine has used before whose parameter has changed
        since (i.e. after an optimizer step: all of them).  First use of an image still goes through packed()."""
        stale = []
        for key, (ver, t) in self._packed.items():
            name, mode, kdt = key
            p = P.get(name)
            if kdt != dt or p is None or (p._version, p.data_ptr()) == ver:
                continue
            if (p.numel() != t.numel() or p.device != t.device or p.dtype != torch.float32
                    or not p.is_contiguous() or p.dim() != 4):
                continue                      # packed() re-creates it
            stale.append((key, p, mode, t))
        if stale:
            ops.pack_weights_multi([(p.detach(), mode, t) for _, p, mode, t in stale], dt)
            for key, p, mode, t in stale:
                self._packed[key] = ((p._version, p.data_ptr()), t)

    # ------------------------------------------------------------------ forward
    def _conv_bn_relu(self, P, B, prefix, idx, xin, yout: Optional[Act], dt, train, save, first=False, x_f32=None,
                      pool=False, out_head=None):
        """conv3x3(+bias) -> BN -> ReLU.  Returns (y Act, saved record).
        out_head = (w (K,C), b): the final 1x1 conv rides on this layer's BN+ReLU pass; the logits are left in
        self._logits and y is not materialised (returns None for it)."""
        wname, bname = f"{prefix}.double_conv.{idx}.weight", f"{prefix}.double_conv.{idx}.bias"
        bnp = f"{prefix}.double_conv.{idx + 1}"
        w, b = P[wname], P[bname]
        Cout = w.shape[0]
        dev = w.device
        ws = self.ws(dev)
        if not train and not save and not first and self.fold_bn_eval:
            # inference: BatchNorm folded into the conv, ReLU in its epilogue -- one pass instead of two
            wp, bf = self.folded(wname, bname, bnp, P, B, dt)
            y = yout if yout is not None else ops.dense(xin.N, xin.H, xin.W, Cout, dt, dev)
            ops.conv_fwd(xin, wp, bf, y, 3, 3, w.shape[1], Cout, relu=True)
            self._pooled = ops.maxpool_fwd(y) if pool else None
            return y, None
        fuse_first = (first and self.fuse_edges and x_f32.shape[3] == 1 and tuple(w.shape[2:]) == (3, 3) and not pool
                      and type(self) is UNetEngine)
        if fuse_first:
            # conv output never stored: statistics-only pass, then conv + BN + ReLU in one pass; backward recomputes it
            N, H, W_, _ = x_f32.shape
            gamma, beta = P[bnp + ".weight"].detach(), P[bnp + ".bias"].detach()
            mom = None
            if train:
                self._stats_gen += 1
                if self.first_stats_moments:
                    rows, mom = ops.first_stats_moments(x_f32, w.detach(), b.detach())
                else:
                    rows = ops.conv_first_stats(x_f32, w.detach(), b.detach(), dt)
                st = ops.bn_finalize(rows, N * H * W_, gamma, beta, B[bnp + ".running_mean"], B[bnp + ".running_var"],
                                     B[bnp + ".num_batches_tracked"], 0.1, BN_EPS, ws)
            else:
                st = ops.bn_eval_coeffs(gamma, beta, B[bnp + ".running_mean"], B[bnp + ".running_var"], BN_EPS, for_backward=save)
            y = yout if yout is not None else ops.dense(N, H, W_, Cout, dt, dev)
            ops.first_bn_act(x_f32, w.detach(), b.detach(), st, y, relu=True)
            rec = None
            if save:
                rec = _Saved()
                rec.xin, rec.x_f32, rec.z, rec.st, rec.first = xin, x_f32, None, st, True
                rec.wname, rec.bname, rec.bnp = wname, bname, bnp
                rec.fused_first = True
                rec.moments = mom
            return y, rec
        if first:
            N, H, W_, _ = x_f32.shape
            z = ops.dense(N, H, W_, Cout, dt, dev)
            rows = ops.conv_first_fwd(x_f32, w.detach(), b.detach(), z, 3, 3, stats=train)
        else:
            Cin = w.shape[1]
            z = ops.dense(xin.N, xin.H, xin.W, Cout, dt, dev)
            rows = ops.conv_fwd(xin, self.packed(wname, w, 0, dt), b.detach(), z, 3, 3, Cin, Cout, stats=train)
        gamma, beta = P[bnp + ".weight"].detach(), P[bnp + ".bias"].detach()
        if train:
            self._stats_gen += 1
            st = ops.bn_finalize(rows, z.M, gamma, beta, B[bnp + ".running_mean"], B[bnp + ".running_var"],
                                 B[bnp + ".num_batches_tracked"], 0.1, BN_EPS, ws)
        else:
            st = ops.bn_eval_coeffs(gamma, beta, B[bnp + ".running_mean"], B[bnp + ".running_var"], BN_EPS, for_backward=save)
        if out_head is not None:
            # last layer: logits = W relu(bn(z)) + b in one pass; the activation is recomputed by the backward passes
            self._logits = ops.affine_act_out(z, st, out_head[0], out_head[1], relu=True)
            y = None
        else:
            y = yout if yout is not None else ops.dense(z.N, z.H, z.W, Cout, dt, dev)
            if pool:        # encoder level: the 2x2 max-pool rides on the BN+ReLU pass
                self._pooled = ops.affine_act_pool(z, y, st, relu=True)
            else:
                ops.affine_act(z, y, st, relu=True)
        rec = None
        if save:
            rec = _Saved()
            rec.xin, rec.x_f32, rec.z, rec.st, rec.first = xin, x_f32, z, st, first
            rec.wname, rec.bname, rec.bnp = wname, bname, bnp
            rec.fused_first = False
        return y, rec

    def block_fwd(self, P, B, prefix, xin, yout, dt, train, save, x_f32=None, pool=False, out_head=None):
        """One DoubleConv (models/unet.py:9-16).  Returns (output Act, saved record); with pool=True the 2x2
        max-pool of the output (pooled Act, code) is left in self._pooled."""
        y1, r1 = self._conv_bn_relu(P, B, prefix, 0, xin, None, dt, train, save, first=x_f32 is not None, x_f32=x_f32)
        y2, r2 = self._conv_bn_relu(P, B, prefix, 3, y1, yout, dt, train, save, pool=pool, out_head=out_head)
        return y2, ((r1, r2) if save else None)

    def block_param_names(self, prefix):
        return _double_conv_names(prefix)

    def forward(self, x: torch.Tensor, P: Dict[str, torch.Tensor], B: Dict[str, torch.Tensor], dt: int,
                train: bool, save: bool):
        """x: fp32 NCHW [N,Cin,H,W] on the GPU.  Returns (logits fp32 NCHW, saved state or None)."""
        if not x.is_cuda:
            raise ops.UNetHipError("unet_amd.UNet runs on MI355X only (input is a CPU tensor; there is no CPU fallback)")
        N, Cin, H, W = x.shape
        L = len(self.features)
        if Cin != self.in_channels:
            raise ops.UNetHipError(f"expected {self.in_channels} input channels, got {Cin}")
        if (H >> L) < 1 or (W >> L) < 1:
            raise ops.UNetHipError(f"input {H}x{W} is too small for {L} pooling levels")
        # H or W not divisible by 2^levels: a level with an odd size pools with the floor rule (stand-alone kernels instead
        # of the fused ones) and the decoder resizes the upsampled tensor to its skip's size (models/unet.py:56-57)
        dev = x.device
        self._last_dt = dt
        self.prepack(P, dt)
        x = x.detach()
        if x.dtype != torch.float32:
            x = x.float()
        x_nhwc = x.contiguous() if Cin == 1 else x.permute(0, 2, 3, 1).contiguous()
        x_nhwc = x_nhwc.view(N, H, W, Cin)
        sv = _Saved() if save else None
        if save:
            sv.blocks, sv.codes, sv.cats, sv.up_in = {}, [], [], []
            sv.dt, sv.shape = dt, (N, H, W)

        # the final 1x1 conv rides on the last BN+ReLU pass whenever that pass exists as a separate kernel (training, or
        # eval with gradients; the no-grad inference path folds BN into the conv and keeps its own epilogue)
        fuse_out = (self.fuse_edges and type(self) is UNetEngine and (train or save or not self.fold_bn_eval)
                    and ops.fused_out_classes_ok(self.n_classes, self.features[0], dt))
        self._logits = None
        cats: List[torch.Tensor] = []
        cur: Optional[Act] = None
        h, w = H, W
        for i, f in enumerate(self.features):
            cat = torch.empty((N, h, w, 2 * f), dtype=ops.torch_dtype(dt), device=dev)
            cats.append(cat)
            p = f"downs.{i}"
            skip = Act(cat, 0, f)
            if self.fused_pool_fwd and h % 2 == 0 and w % 2 == 0:
                _, rec = self.block_fwd(P, B, p, cur, skip, dt, train, save, x_f32=x_nhwc if i == 0 else None, pool=True)
                cur, code = self._pooled
                self._pooled = None
            else:
                _, rec = self.block_fwd(P, B, p, cur, skip, dt, train, save, x_f32=x_nhwc if i == 0 else None)
                cur, code = ops.maxpool_fwd(skip)
            if save:
                sv.blocks[p] = rec
            if save:
                sv.codes.append(code)
            h, w = h // 2, w // 2
        cur, rec = self.block_fwd(P, B, "bottleneck", cur, None, dt, train, save)
        if save:
            sv.blocks["bottleneck"] = rec
        for j, f in enumerate(reversed(self.features)):
            cat = cats[L - 1 - j]
            wt, bt = P[f"ups.{2 * j}.weight"], P[f"ups.{2 * j}.bias"]
            if save:
                sv.up_in.append(cur)
            if (2 * cur.H, 2 * cur.W) == (cat.shape[1], cat.shape[2]):
                # ConvTranspose2d(2f -> f, k2, s2) as GEMM + pixel shuffle into cat[..., f:2f]
                ops.conv_fwd(cur, self.packed(f"ups.{2 * j}.weight", wt, 2, dt), bt.detach(), Act(cat, f, f), 1, 1,
                             2 * f, 4 * f, epi=1)
            else:
                # the skip is a pixel larger (odd size at this level): upsample into a buffer of its own, then
                # F.interpolate(..., mode="bilinear", align_corners=True) into the concat slice (models/unet.py:56-57)
                up = ops.dense(cur.N, 2 * cur.H, 2 * cur.W, f, dt, dev)
                ops.conv_fwd(cur, self.packed(f"ups.{2 * j}.weight", wt, 2, dt), bt.detach(), up, 1, 1, 2 * f, 4 * f, epi=1)
                ops.resize_bilinear(up, Act(cat, f, f))
            p = f"ups.{2 * j + 1}"
            head = None
            if j == L - 1 and fuse_out:
                head = (P["final_conv.weight"].detach().view(self.n_classes, -1).contiguous(), P["final_conv.bias"].detach())
            cur, rec = self.block_fwd(P, B, p, Act(cat, 0, 2 * f), None, dt, train, save, out_head=head) if head is not None \
                else self.block_fwd(P, B, p, Act(cat, 0, 2 * f), None, dt, train, save)
            if save:
                sv.blocks[p] = rec
        if fuse_out:
            logits, self._logits = self._logits, None
        else:
            logits = ops.conv_out_fwd(cur, P["final_conv.weight"].detach().view(self.n_classes, -1).contiguous(),
                                      P["final_conv.bias"].detach())
        if save:
            sv.fused_out = fuse_out
            sv.last, sv.cats = cur, cats
            if self.keep_saved:
                self.last_saved = sv
        return logits, sv

    # ------------------------------------------------------------------ backward
    def block_bwd(self, P, sv, prefix, dy, grads: Dict[str, torch.Tensor], dt, need_dx=True,
                  split_f: int = 0):
        """Backward of one DoubleConv.  Returns dx (Act), or (dskip, dup) when split_f > 0.
        dy: an Act, an ops.PoolGrad, or -- last block of the fused output head -- the fp32 NCHW logit gradient."""
        r1, r2 = sv.blocks[prefix]
        dev = r2.z.buf.device
        ws = self.ws(dev)
        # second conv
        z2 = r2.z
        dz2 = ops.dense(z2.N, z2.H, z2.W, z2.C, dt, dev)
        if isinstance(dy, torch.Tensor):      # fused output head: dy = W^T dlogits is recomputed inside the BN backward passes
            wf = P["final_conv.weight"]
            dg, db_, dbias, dwf, dbf = ops.bn_relu_bwd_out(dy, wf.detach().view(self.n_classes, -1).contiguous(), z2, dz2, r2.st, ws)
            grads["final_conv.weight"], grads["final_conv.bias"] = dwf.view_as(wf), dbf
        elif isinstance(dy, ops.PoolGrad):    # encoder level: un-pool + skip add happen inside the BN backward passes
            dg, db_, dbias = ops.bn_relu_bwd_pool(dy, z2, dz2, r2.st, ws)
        else:
            dg, db_, dbias = ops.bn_relu_bwd(dy, z2, dz2, r2.st, ws)
        grads[r2.bnp + ".weight"], grads[r2.bnp + ".bias"], grads[r2.bname] = dg, db_, dbias
        w2 = P[r2.wname]
        grads[r2.wname] = self.wgrad_async(dev, lambda w_: ops.conv_wgrad(dz2, r2.xin, 3, 3, w_)[0].view_as(w2), [dz2.buf, r2.xin.buf])
        dy1 = ops.dense(z2.N, z2.H, z2.W, w2.shape[1], dt, dev)
        # the first pass of the NEXT BatchNorm backward (sum dy', sum dy'*xhat over dy1 and z1) rides on this dgrad's epilogue
        # where the fused kernel takes the shape (bf16, whole tiles): dy1 is then read once, by the apply pass, not twice
        rows1 = None
        if self.bn_reduce_in_dgrad and not getattr(r1, "fused_first", False) and r1.z is not None and type(self) is UNetEngine:
            rows1 = ops.conv_dgrad_bnred(dz2, self.packed(r2.wname, w2, 1, dt), dy1, r1.z, r1.st)
        if rows1 is None:
            ops.conv_fwd(dz2, self.packed(r2.wname, w2, 1, dt), None, dy1, 3, 3, w2.shape[0], w2.shape[1])
        del dz2
        # first conv
        w1 = P[r1.wname]
        if getattr(r1, "fused_first", False):
            # first layer of the network: its conv output is recomputed from the image inside both BN backward passes and
            # the weight gradient comes out of the second one (nothing but the per-channel gradients is written)
            dg, db_, dbias, dw1 = ops.first_bn_relu_bwd(dy1, r1.x_f32, w1.detach(), P[r1.bname].detach(), r1.st, ws,
                                                        moments=getattr(r1, "moments", None))
            grads[r1.bnp + ".weight"], grads[r1.bnp + ".bias"], grads[r1.bname], grads[r1.wname] = dg, db_, dbias, dw1
            return None
        z1 = r1.z
        dz1 = ops.dense(z1.N, z1.H, z1.W, z1.C, dt, dev)
        dg, db_, dbias = ops.bn_relu_bwd(dy1, z1, dz1, r1.st, ws, sum_rows=rows1)
        grads[r1.bnp + ".weight"], grads[r1.bnp + ".bias"], grads[r1.bname] = dg, db_, dbias
        del dy1
        if r1.first:
            grads[r1.wname] = self.wgrad_async(dev, lambda w_: ops.conv_first_wgrad(dz1, r1.x_f32, 3, 3, w_), [dz1.buf, r1.x_f32])
            return None
        grads[r1.wname] = self.wgrad_async(dev, lambda w_: ops.conv_wgrad(dz1, r1.xin, 3, 3, w_)[0].view_as(w1), [dz1.buf, r1.xin.buf])
        if not need_dx:
            return None
        Cin = w1.shape[1]
        wd = self.packed(r1.wname, w1, 1, dt)
        if split_f:
            dskip = ops.dense(z1.N, z1.H, z1.W, split_f, dt, dev)
            dup = ops.dense(z1.N, z1.H, z1.W, Cin - split_f, dt, dev)
            # per-tile channel sums of the epilogue (the fp32 values before rounding): their upper half is the
            # bias gradient of the transposed conv that produced `up`, so no extra pass over dup is needed
            rows = ops.conv_fwd(dz1, wd, None, dskip, 3, 3, w1.shape[0], Cin, y2=dup, csplit=split_f, stats=True)
            return dskip, dup, rows
        dx = ops.dense(z1.N, z1.H, z1.W, Cin, dt, dev)
        ops.conv_fwd(dz1, wd, None, dx, 3, 3, w1.shape[0], Cin)
        return dx

    def backward(self, dlogits: torch.Tensor, P: Dict[str, torch.Tensor], sv,
                 grad_hook: Optional[Callable[[List[str]], None]] = None) -> Dict[str, torch.Tensor]:
        dt = sv.dt
        L = len(self.features)
        dev = dlogits.device
        ws = self.ws(dev)
        ws.clear_arena()             # one fill for every BatchNorm backward's accumulators of this pass (ops.Workspace)
        grads: Dict[str, torch.Tensor] = {}

        def emit(names):
            if grad_hook is not None:
                self.join_side(dev)          # the reducer reads these gradients in current-stream order
                grad_hook(names, [grads[n] for n in names])

        dlogits = dlogits.contiguous().float()
        wf = P["final_conv.weight"]
        if getattr(sv, "fused_out", False):
            dcur = dlogits              # block_bwd of the last decoder block consumes the logit gradient directly
        else:
            dcur, dwf, dbf = ops.conv_out_bwd(dlogits, sv.last, wf.detach().view(self.n_classes, -1).contiguous(), ws)
            grads["final_conv.weight"], grads["final_conv.bias"] = dwf.view_as(wf), dbf
            emit(["final_conv.weight", "final_conv.bias"])
        dskips = [None] * L
        for j in reversed(range(L)):
            f = list(reversed(self.features))[j]
            p = f"ups.{2 * j + 1}"
            res = self.block_bwd(P, sv, p, dcur, grads, dt, split_f=f)
            if j == L - 1 and getattr(sv, "fused_out", False):
                emit(["final_conv.weight", "final_conv.bias"])
            dskip, dup = res[0], res[1]
            emit(self.block_param_names(p))
            dskips[L - 1 - j] = dskip
            xin = sv.up_in[j]
            tname = f"ups.{2 * j}"
            if (2 * xin.H, 2 * xin.W) != (dup.H, dup.W):
                # the upsampled tensor went through the bilinear resize: its adjoint first, then the bias gradient from
                # the small tensor (the dgrad epilogue's rows belong to the resized one)
                dsmall = ops.dense(xin.N, 2 * xin.H, 2 * xin.W, f, dt, dev)
                ops.resize_bilinear_bwd(dup, dsmall)
                dup = dsmall
                res = (dskip, dup)
            if len(res) > 2:        # [tiles][2f][sum, sum of squares] from the dgrad epilogue
                rows = res[2]
                ctot = rows.shape[1]
                grads[tname + ".bias"] = ops.rows_colsum(rows, 2 * f, ctot - f, 2)      # the sums of the upsampled half
            else:
                grads[tname + ".bias"] = ops.channel_sum(dup, ws)
            grads[tname + ".weight"] = self.wgrad_async(dev, lambda w_: ops.convT_wgrad(xin, dup, w_), [xin.buf, dup.buf])
            dcur = ops.dense(xin.N, xin.H, xin.W, xin.C, dt, dev)
            ops.convT_dgrad(dup, self.packed(tname + ".weight", P[tname + ".weight"], 3, dt), xin.C, dcur)
            del dup
            emit([tname + ".weight", tname + ".bias"])
        dcur = self.block_bwd(P, sv, "bottleneck", dcur, grads, dt)
        emit(self.block_param_names("bottleneck"))
        for i in reversed(range(L)):
            if self.fused_pool_bwd and dskips[i].H % 2 == 0 and dskips[i].W % 2 == 0:
                dy2 = ops.PoolGrad(dcur, sv.codes[i], dskips[i])
            else:
                dy2 = ops.maxpool_bwd(dcur, sv.codes[i], dskips[i])
            dskips[i] = None
            dcur = self.block_bwd(P, sv, f"downs.{i}", dy2, grads, dt, need_dx=(i > 0))
            emit(self.block_param_names(f"downs.{i}"))
        self.join_side(dev)
        return grads
