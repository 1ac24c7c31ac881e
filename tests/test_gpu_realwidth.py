"""Oracle parity at the BENCHMARK's channel widths (BASELINE configs[1]: features [64,128,256,512], bottleneck 1024).

tests/test_gpu_kernels.py checks every kernel on small channel counts and tests/test_gpu_fullsize.py checks the full
32x1x512x512 shapes through size-independent properties; this file closes the gap between them: the exact kernel
instantiations bench.py times (conv3x3_v2<2,false>, the deferred-statistics 64-channel variant <1,false,true>,
bn_stats_finalize, rows_colsum bias gradient, split dgrad, conv_wgrad2, convT2 at 512 / 1024 channels, fused pool
passes) compared with torch-CPU fp32 (F.conv2d / F.conv_transpose2d; accumulation over K = 9*1024 = 9216) and, for
the whole engine, with the pinned oracle (oracle/unet_ref.py, oracle/loss_ref.py) computed live on the same inputs.

Spatial sizes are multiples of the 16x32 output tile so the specialised (GEN=false) paths run.

Tolerances: fp32 mode <= 1e-4 relative (max-abs error / max-abs reference per tensor), as north_star states;
bf16 per kernel: both sides see bf16-rounded inputs, fp32 accumulation, only the output rounding differs (see
tests/test_gpu_kernels.py::close); bf16 whole model: stated in test_engine_real_widths_bf16 below.
"""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import close, from_act, q, rnd, to_act

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def relmax(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rell2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


# N, H, W, Cin, Cout: the (Cin, Cout) pairs of the benchmark model's 3x3 convs (SURVEY 8(d) per-layer table)
REAL_CONV = [
    (2, 32, 64, 64, 64),        # down0.b / up3.b  (64-channel tiles, deferred BN statistics)
    (2, 32, 64, 128, 64),       # up3.a
    (1, 32, 32, 64, 128),       # down1.a
    (1, 32, 32, 128, 128),      # down1.b / up2.b
    (1, 16, 64, 256, 128),      # up2.a
    (1, 16, 32, 128, 256),      # down2.a
    (1, 16, 32, 256, 256),      # down2.b / up1.b
    (1, 16, 32, 512, 256),      # up1.a
    (1, 16, 32, 256, 512),      # down3.a
    (1, 16, 32, 512, 512),      # down3.b / up0.b
    (1, 16, 32, 1024, 512),     # up0.a
    (1, 16, 32, 512, 1024),     # bott.a
    (1, 16, 32, 1024, 1024),    # bott.b
]


REAL_CONV5 = [          # deployed RepUNet [3,5] -> 5x5 (models/reparm_unet.py:139) at its real widths, whole 16x32 tiles
    (2, 12, 20, 24, 40),        # ragged: the generic instantiations (bounds checked per lane) of the same entry points
    (1, 9, 33, 64, 72),
    (2, 32, 64, 64, 64),
    (1, 32, 32, 128, 128),
    (1, 16, 64, 256, 128),
    (1, 16, 32, 512, 256),
    (1, 16, 32, 1024, 1024),
]


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("k", [(5, 5), (3, 1), (5, 1), (1, 3), (1, 5)])
@pytest.mark.parametrize("shape", REAL_CONV5)
def test_conv5x5_and_asymmetric_real_widths(dev, dt, k, shape):
    """bf16: the 5x5 / k x 1 / 1 x k instantiations of the LDS-DMA kernel (models/reparm_unet.py:55-61,139); fp32: conv_igemm.
    Forward with statistics, forward with the deployed path's bias + ReLU epilogue, dgrad; against F.conv2d on the CPU."""
    from unet_amd import ops
    N, H, W, Cin, Cout = shape
    kh, kw = k
    x = q(rnd((N, Cin, H, W), 111), dt)
    w = q(rnd((Cout, Cin, kh, kw), 112, (Cin * kh * kw) ** -0.5), dt)
    b = rnd((Cout,), 113)
    x.requires_grad_(True); w.requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=(kh // 2, kw // 2))
    dyv = q(rnd(tuple(ref.shape), 114), dt)
    ref.backward(dyv)

    xa = to_act(x.detach(), dt, dev)
    ya = ops.dense(N, H, W, Cout, dt, dev)
    wp = ops.pack_weights(w.detach().to(dev), 0, dt)
    rows = ops.conv_fwd(xa, wp, b.to(dev), ya, kh, kw, Cin, Cout, stats=True)
    close(from_act(ya), ref.detach(), dt, "conv fwd")
    s = rows.double().sum(0).cpu()
    r = ref.detach().double()
    close(s[:, 0].float(), r.sum((0, 2, 3)).float(), dt, "stats sum", f32_out=True)
    close(s[:, 1].float(), (r * r).sum((0, 2, 3)).float(), dt, "stats sumsq", f32_out=True)

    ops.conv_fwd(xa, wp, b.to(dev), ya, kh, kw, Cin, Cout, relu=True)
    close(from_act(ya), torch.relu(ref.detach()), dt, "conv + ReLU")

    dza = to_act(dyv, dt, dev)
    dxa = ops.dense(N, H, W, Cin, dt, dev)
    wd = ops.pack_weights(w.detach().to(dev), 1, dt)
    ops.conv_fwd(dza, wd, None, dxa, kh, kw, Cout, Cin)
    close(from_act(dxa), x.grad, dt, "conv dgrad")
    # the branch dgrads of a block are summed into one dx (rep_engine: accumulate=True on all but the first)
    base = q(rnd((N, Cin, H, W), 115), dt)
    dxb = to_act(base, dt, dev)
    ops.conv_fwd(dza, wd, None, dxb, kh, kw, Cout, Cin, accumulate=True)
    close(from_act(dxb), base + x.grad, dt, "conv dgrad, accumulated")

    # weight gradient the way rep_engine issues it: up to 9 taps in one launch, a 5x5 row-wise (bf16: 2x5 + 2x5 + 1x5)
    ws = ops.Workspace(dev)
    if kh * kw <= 9:
        dw = ops.conv_wgrad(dza, xa, kh, kw, ws)[0]
    else:
        dw = torch.empty((Cout, Cin, kh * kw), dtype=torch.float32, device=dev)
        r = 0
        while r < kh:
            nr = 2 if (dt == 1 and r + 1 < kh) else 1
            ops.conv_wgrad(dza, xa, nr, kw, ws, dw=dw, pad_h=kh // 2 - r, pad_w=kw // 2, dw_taps=kh * kw, dw_tap0=r * kw)
            r += nr
    close(dw.view(Cout, Cin, kh, kw).cpu(), w.grad, dt, "conv wgrad", f32_out=True)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", REAL_CONV)
def test_conv3x3_real_widths(dev, dt, shape):
    from unet_amd import ops
    N, H, W, Cin, Cout = shape
    x = q(rnd((N, Cin, H, W), 101), dt)
    w = q(rnd((Cout, Cin, 3, 3), 102, (Cin * 9) ** -0.5), dt)
    b = rnd((Cout,), 103)
    x.requires_grad_(True); w.requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=1)
    dyv = q(rnd(tuple(ref.shape), 104), dt)
    ref.backward(dyv)

    xa = to_act(x.detach(), dt, dev)
    ya = ops.dense(N, H, W, Cout, dt, dev)
    rows = ops.conv_fwd(xa, ops.pack_weights(w.detach().to(dev), 0, dt), b.to(dev), ya, 3, 3, Cin, Cout, stats=True)
    close(from_act(ya), ref.detach(), dt, "conv fwd")
    s = rows.double().sum(0).cpu()
    r = ref.detach().double()
    close(s[:, 0].float(), r.sum((0, 2, 3)).float(), dt, "stats sum", f32_out=True)
    close(s[:, 1].float(), (r * r).sum((0, 2, 3)).float(), dt, "stats sumsq", f32_out=True)

    dza = to_act(dyv, dt, dev)
    dxa = ops.dense(N, H, W, Cin, dt, dev)
    wd = ops.pack_weights(w.detach().to(dev), 1, dt)
    ops.conv_fwd(dza, wd, None, dxa, 3, 3, Cout, Cin)
    close(from_act(dxa), x.grad, dt, "conv dgrad")
    base = q(rnd((N, Cin, H, W), 105), dt)
    dxb = to_act(base, dt, dev)
    ops.conv_fwd(dza, wd, None, dxb, 3, 3, Cout, Cin, accumulate=True)
    close(from_act(dxb), base + x.grad, dt, "conv dgrad, accumulated")

    # the decoder's first conv hands the two halves of the concat gradient to two buffers and takes the ConvTranspose2d
    # bias gradient from the epilogue's statistics rows (engine.block_bwd, split_f)
    if Cin == 2 * Cout:
        f = Cout
        d1, d2 = ops.dense(N, H, W, f, dt, dev), ops.dense(N, H, W, f, dt, dev)
        rows2 = ops.conv_fwd(dza, wd, None, d1, 3, 3, Cout, Cin, y2=d2, csplit=f, stats=True)
        close(from_act(d1), x.grad[:, :f], dt, "split dgrad skip half")
        close(from_act(d2), x.grad[:, f:], dt, "split dgrad up half")
        dbias = ops.rows_colsum(rows2, 2 * f, Cin - f, 2)
        close(dbias.cpu(), x.grad[:, f:].double().sum((0, 2, 3)).float(), dt, "convT bias grad from rows", f32_out=True)

    dw, _, _ = ops.conv_wgrad(dza, xa, 3, 3, ops.Workspace(dev))
    close(dw.view(Cout, Cin, 3, 3).cpu(), w.grad, dt, "conv wgrad", f32_out=True)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", [(2, 8, 8, 1024, 512), (1, 16, 32, 512, 256), (1, 16, 32, 256, 128), (2, 16, 32, 128, 64)])
def test_conv_transpose_real_widths(dev, dt, shape):
    from unet_amd import ops
    N, h, w_, Ci, Co = shape
    x = q(rnd((N, Ci, h, w_), 107), dt)
    wt = q(rnd((Ci, Co, 2, 2), 108, Ci ** -0.5), dt)
    b = rnd((Co,), 109)
    x.requires_grad_(True); wt.requires_grad_(True)
    ref = F.conv_transpose2d(x, wt, b, stride=2)
    dyv = q(rnd(tuple(ref.shape), 110), dt)
    ref.backward(dyv)
    xa = to_act(x.detach(), dt, dev)
    cat = torch.zeros((N, 2 * h, 2 * w_, 2 * Co), dtype=ops.torch_dtype(dt), device=dev)
    ops.conv_fwd(xa, ops.pack_weights(wt.detach().to(dev), 2, dt), b.to(dev), ops.Act(cat, Co, Co), 1, 1, Ci, 4 * Co, epi=1)
    close(from_act(ops.Act(cat, Co, Co)), ref.detach(), dt, "convT fwd")
    assert float(cat[..., :Co].float().abs().max()) == 0.0
    dup = to_act(dyv, dt, dev)
    dx = ops.dense(N, h, w_, Ci, dt, dev)
    ops.convT_dgrad(dup, ops.pack_weights(wt.detach().to(dev), 3, dt), Ci, dx)
    close(from_act(dx), x.grad, dt, "convT dgrad")
    dw = ops.convT_wgrad(xa, dup, ops.Workspace(dev))
    close(dw.cpu(), wt.grad, dt, "convT wgrad", f32_out=True)


@pytest.mark.parametrize("shape", [(2, 32, 64, 64, 64), (1, 32, 32, 128, 128), (1, 16, 64, 256, 256), (1, 16, 32, 512, 512),
                                   (1, 16, 32, 1024, 1024), (2, 16, 32, 128, 64)])
def test_dgrad_with_bn_backward_sums_real_widths(dev, shape):
    """unet_conv_dgrad_bnred: the 3x3 dgrad of a DoubleConv's second conv with the first pass of the BatchNorm backward in front
    of it in its epilogue (models/unet.py:11-13 under autograd).  dy must be bit-equal to the plain dgrad launch; the sums must
    equal an fp64 evaluation of sum dy' and sum dy'*xhat on the CPU (from the unrounded dgrad result), to summation noise; and
    the BatchNorm backward fed with the rows must agree with the two-pass form, which sums the bf16-ROUNDED dy: the difference
    is that rounding's random walk, bounded here by 6 standard deviations of it."""
    from unet_amd import ops
    N, H, W, Cmid, C = shape                # dz has Cmid channels (the conv's outputs), dy / z have C (its inputs)
    dt = 1
    dzv = q(rnd((N, Cmid, H, W), 121), dt)
    w = q(rnd((Cmid, C, 3, 3), 122, (C * 9) ** -0.5), dt)
    zv = q(rnd((N, C, H, W), 123) * 1.5 + 0.3, dt)
    gamma, beta = rnd((C,), 124).abs() + 0.5, rnd((C,), 125) * 0.3
    dy_ref = F.conv_transpose2d(dzv.double(), w.double(), padding=1)         # dgrad of conv2d(pad 1) in fp64
    n = N * H * W
    mean, var = zv.double().mean((0, 2, 3)), zv.double().var((0, 2, 3), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    scale, shift = gamma.double() * invstd, beta.double() - mean * gamma.double() * invstd
    st = ops.BNState(scale.float().to(dev), shift.float().to(dev), mean.float().to(dev), invstd.float().to(dev))
    sc, sh, mu, is_ = (t.float().double()[None, :, None, None] for t in (scale, shift, mean, invstd))
    mask = (zv.double() * sc + sh) > 0
    xhat = (zv.double() - mu) * is_
    d = dy_ref * mask
    S1, S2 = d.sum((0, 2, 3)), (d * xhat).sum((0, 2, 3))

    dza, za = to_act(dzv, dt, dev), to_act(zv, dt, dev)
    wd = ops.pack_weights(w.to(dev), 1, dt)
    dy_a, dy_b = ops.dense(N, H, W, C, dt, dev), ops.dense(N, H, W, C, dt, dev)
    ops.conv_fwd(dza, wd, None, dy_a, 3, 3, Cmid, C)
    rows = ops.conv_dgrad_bnred(dza, wd, dy_b, za, st)
    assert rows is not None and tuple(rows.shape) == (N * (H // 16) * (W // 32), C, 2)
    assert torch.equal(dy_a.buf, dy_b.buf)
    got = rows.double().sum(0).cpu()
    # fp32 accumulators over K = 9*Cmid products of bf16 operands, then a sum of n terms: relative to the root sum of squares
    rss1, rss2 = d.pow(2).sum((0, 2, 3)).sqrt(), (d * xhat).pow(2).sum((0, 2, 3)).sqrt()
    assert float(((got[:, 0] - S1).abs() / rss1).max()) <= 2e-4, "sum dy'"
    assert float(((got[:, 1] - S2).abs() / rss2).max()) <= 2e-4, "sum dy' xhat"
    ws = ops.Workspace(dev)
    dz_a, dz_b = ops.dense(N, H, W, C, dt, dev), ops.dense(N, H, W, C, dt, dev)
    dg_a, db_a, _ = ops.bn_relu_bwd(dy_a, za, dz_a, st, ws)
    dg_b, db_b, dbias_b = ops.bn_relu_bwd(dy_b, za, dz_b, st, ws, sum_rows=rows)
    assert float(((dg_b.cpu().double() - S2).abs() / rss2).max()) <= 2e-4 and float(((db_b.cpu().double() - S1).abs() / rss1).max()) <= 2e-4
    walk1, walk2 = 2.0 ** -9 * rss1, 2.0 ** -9 * rss2                       # bf16 storage rounding of dy in the two-pass form
    assert float(((db_a.cpu().double() - db_b.cpu().double()).abs() / walk1).max()) <= 6.0
    assert float(((dg_a.cpu().double() - dg_b.cpu().double()).abs() / walk2).max()) <= 6.0
    assert float((dz_a.buf.float() - dz_b.buf.float()).abs().max()) <= 2.0 ** -6 * float(dz_a.buf.float().abs().max())
    assert float(dbias_b.abs().max()) <= 2e-2 * float(dy_ref.abs().max()) * n ** 0.5          # noise-only in train mode


def test_engine_with_bn_sums_in_dgrad(dev):
    """The opt-in schedule (UNET_AMD_BN_REDUCE_IN_DGRAD=1 / engine.bn_reduce_in_dgrad): whole bf16 backward with the first pass
    of every eligible BatchNorm backward inside the producing dgrad.  Same forward; gradients agree with the default schedule to
    the rounding of dy it skips (the sums see dy before it is stored as bf16): per tensor rel-L2 <= 1e-2."""
    from unet_amd.losses import get_loss
    from unet_amd.models import REGISTRY
    torch.manual_seed(11)
    m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[64, 128]).to(dev).train()
    m.compute_dtype = "bf16"
    g = torch.Generator().manual_seed(12)
    x = torch.rand(2, 1, 64, 128, generator=g).to(dev)
    y = (torch.rand(2, 64, 128, generator=g) > 0.9).long().to(dev)
    loss_fn = get_loss("ce_dice", ignore_index=255)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    outs = []
    for fused in (False, True):
        m.load_state_dict(state)
        m._engine.bn_reduce_in_dgrad = fused
        m.zero_grad(set_to_none=True)
        lg = m(x)
        loss_fn(lg, y).backward()
        outs.append((lg.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    m._engine.bn_reduce_in_dgrad = False
    assert torch.equal(outs[0][0], outs[1][0])
    tbias = lambda k: k.startswith("ups.") and k.endswith(".bias") and ".double_conv." not in k      # cancelling sums (see test_gpu_benchsize)
    err = {k: rell2(outs[1][1][k].cpu(), outs[0][1][k].cpu()) for k in outs[0][1] if not _noise_bias(k)}
    worst = max((e, k) for k, e in err.items() if not tbias(k))
    worst_t = max((e, k) for k, e in err.items() if tbias(k))
    print(f"BN sums in the dgrad epilogue vs the two-pass form: worst per-tensor gradient rel-L2 {worst[0]:.2e} ({worst[1]}), "
          f"ConvTranspose2d biases {worst_t[0]:.2e} ({worst_t[1]})")
    assert worst[0] <= 1e-2 and worst_t[0] <= 5e-2, (worst, worst_t)


# ---------------------------------------------------------------------------------------------------------------
# whole engine at the benchmark's widths against the pinned oracle, computed live
#
# ReLU and max-pool make the network piecewise linear.  Two correct fp32 evaluations differ by summation-order
# rounding (~1e-6 after a K = 1152 ... 9216 reduction), so out of the ~3.5 M ReLU decisions of this model one or two
# elements with |pre-activation| ~ 1e-6 land on different sides of zero (measured: tools/diag_masks.py, 2 of 3.5 M).
# Each such flip drops one term from a BatchNorm backward batch sum that cancels to ~1 % of its terms, and every
# gradient upstream moves by ~1e-2 -- in the reference on another machine just as here.  The comparison is therefore
# made on the SAME linear piece: the oracle is evaluated with the HIP path's ReLU masks and pool winners
# (oracle/unet_ref.py `decisions`), and the test separately requires that every decision on which the free-running
# oracle disagrees is a genuine near-tie and that there are almost none of them.

FEATURES = [64, 128, 256, 512]


def _noise_bias(k):
    return k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias")


def _nchw(act):
    return act.buf[..., act.off:act.off + act.C].float().cpu().permute(0, 3, 1, 2).contiguous()


def _hip_decisions(engine):
    """Post-ReLU tensors and pool winners of the engine's last training forward (engine.keep_saved)."""
    from unet_amd import ops
    sv = engine.last_saved
    L = len(engine.features)
    post, pool = {}, {}
    for p, (r1, r2) in sv.blocks.items():
        post[p + ".a"] = _nchw(r2.xin)                      # the first conv's BN+ReLU output is the second conv's input
    for i, f in enumerate(engine.features):
        post[f"downs.{i}.b"] = sv.cats[i][..., :f].float().cpu().permute(0, 3, 1, 2).contiguous()
        pool[f"pool{i}"] = ops.pool_indices_int64(sv.codes[i], sv.cats[i].shape[2]).cpu()
    post["bottleneck.b"] = _nchw(sv.up_in[0])
    for j in range(L):
        if j + 1 < L:
            post[f"ups.{2 * j + 1}.b"] = _nchw(sv.up_in[j + 1])
        elif sv.last is not None:
            post[f"ups.{2 * j + 1}.b"] = _nchw(sv.last)
        else:
            # fused output head: the last activation is never stored; the backward passes decide the ReLU by
            # fmaf(z, scale, shift) > 0, which is exactly what the stand-alone BN+ReLU kernel computes
            r2 = sv.blocks[f"ups.{2 * j + 1}"][1]
            y = ops.dense(r2.z.N, r2.z.H, r2.z.W, r2.z.C, r2.z.dt, r2.z.buf.device)
            ops.affine_act(r2.z, y, r2.st, relu=True)
            post[f"ups.{2 * j + 1}.b"] = _nchw(y)
    return post, pool


def _oracle(x, y, state, emulate_bf16=False, decisions=None, taps=None):
    from oracle import loss_ref, unet_ref
    st = {k: v.clone() for k, v in state.items()}
    names = [k for k in st if "running_" not in k and "num_batches" not in k]
    for k in names:
        st[k].requires_grad_(True)
    nb = {}
    logits = unet_ref.unet_forward(x, st, True, nb, emulate_bf16=emulate_bf16, decisions=decisions, taps=taps)
    loss = loss_ref.ce_dice(logits, y, 1.0, 1.0, 255)
    loss.backward()
    return logits.detach(), float(loss.detach()), {k: st[k].grad for k in names}, nb


@pytest.fixture(scope="module")
def real_case():
    """UNet[64,128,256,512], 2x1x64x64: the 3x3 layers cover whole 16x32 tiles down to 16x32 at level 2 (the deeper
    levels, 8x8 and 4x4, run the general path exactly as they would for any input that small)."""
    from unet_amd.models import REGISTRY
    torch.manual_seed(42)
    m = REGISTRY["unet"](in_channels=1, n_classes=2, features=FEATURES)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(42)
    x = torch.rand(2, 1, 64, 64, generator=g)
    y = (torch.rand(2, 64, 64, generator=g) > 0.95).long()
    taps = {}
    free = _oracle(x, y, state, taps=taps)
    return m, state, x, y, free, taps


def _run_hip(m, state, x, y, dev, mode):
    from unet_amd.losses import get_loss
    m.load_state_dict(state)
    m = m.to(dev).train()
    m.compute_dtype = mode
    m._engine.keep_saved = True
    m.zero_grad(set_to_none=True)
    logits = m(x.to(dev))
    loss = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)(logits, y.to(dev))
    loss.backward()
    post, pool = _hip_decisions(m._engine)
    m._engine.keep_saved = False
    m._engine.last_saved = None
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    bufs = {k: b.detach().cpu().clone() for k, b in m.named_buffers()}
    return logits.detach().float().cpu(), float(loss.detach()), grads, bufs, post, pool


def _decision_report(post, pool, taps, tie_tol):
    """(#ReLU decisions, #flips, #pool flips); asserts every disagreement with the free-running oracle is a near-tie."""
    total = flips = pflips = 0
    for key, yh in post.items():
        yr = taps["relu/" + key].detach()
        mism = (yh > 0) != (yr > 0)
        total += mism.numel()
        flips += int(mism.sum())
        if mism.any():                  # one side is exactly 0, the other must be within rounding of it
            assert float(torch.maximum(yh.abs(), yr.abs())[mism].max()) <= tie_tol, key
    for key, idx in pool.items():
        ri = taps[key]
        mism = idx != ri
        pflips += int(mism.sum())
        if mism.any():                  # both winners must hold (nearly) the same value
            i = int(key[4:])
            src = taps[f"skip{i}"].detach().flatten(2)
            a, b = src.gather(2, idx.flatten(2)), src.gather(2, ri.flatten(2))
            assert float((a - b).abs()[mism.flatten(2)].max()) <= tie_tol, key
    return total, flips, pflips


def _bn_operand_scale(state, nb, count_of):
    """Per BN layer key ("<block>.a" / "<block>.b") the per-channel size of the operands of the pre-activation
    y = scale*z + shift at the kink: 1 + |shift_c| (at y = 0, |scale*z| = |shift|; the 1 is the unit variance of the normalised
    activation, i.e. the upstream noise scale).  One bf16 rounding step of the stored conv output z moves y by 2^-8 of that.
    mean and variance of the batch are recovered from the oracle's updated running statistics (momentum 0.1)."""
    out = {}
    for k in nb:
        if not k.endswith(".running_mean"):
            continue
        bnp = k[:-len(".running_mean")]
        blk, idx = bnp.rsplit(".double_conv.", 1)
        n = count_of(blk)
        mean = (nb[k] - 0.9 * state[k]) / 0.1
        var_b = (nb[bnp + ".running_var"] - 0.9 * state[bnp + ".running_var"]) / 0.1 * ((n - 1) / n)
        scale = state[bnp + ".weight"] / torch.sqrt(var_b + 1e-5)
        shift = state[bnp + ".bias"] - mean * scale
        out[blk + (".a" if idx == "1" else ".b")] = 1.0 + shift.abs()
    return out


def _decision_stats(post, pool, taps, operand_scale=None):
    """-> (#ReLU decisions, magnitudes of the flipped ones, #pool decisions, value gaps of the flipped pool winners).
    A flipped ReLU decision has one side exactly 0; its magnitude is the other side's value.  A flipped pool decision's gap is
    the difference between the two winners' values in the oracle's tensor.  operand_scale ({key: per-channel tensor}): the
    ReLU magnitudes are divided by it (see _bn_operand_scale)."""
    total, mags, ptotal, pgaps = 0, [], 0, []
    for key, yh in post.items():
        yr = taps["relu/" + key].detach()
        mism = (yh > 0) != (yr > 0)
        total += mism.numel()
        if mism.any():
            mag = torch.maximum(yh.abs(), yr.abs())
            if operand_scale is not None:
                mag = mag / operand_scale[key][None, :, None, None]
            mags.append(mag[mism])
    for key, idx in pool.items():
        ri = taps[key]
        mism = idx != ri
        ptotal += mism.numel()
        if mism.any():
            src = taps[f"skip{int(key[4:])}"].detach().flatten(2)
            a, b = src.gather(2, idx.flatten(2)), src.gather(2, ri.flatten(2))
            pgaps.append((a - b).abs()[mism.flatten(2)])
    cat = lambda ts: torch.cat(ts) if ts else torch.zeros(0)
    return total, cat(mags), ptotal, cat(pgaps)


def _q(t, q):
    if t.numel() == 0:
        return 0.0
    k = min(t.numel(), max(1, int(round(q * t.numel()))))
    return float(t.flatten().kthvalue(k).values)


def _cosine(ga, gb, keys):
    a = torch.cat([ga[k].double().flatten() for k in keys])
    b = torch.cat([gb[k].double().flatten() for k in keys])
    return float((a @ b) / (a.norm() * b.norm()))


def test_engine_real_widths_fp32(dev, real_case):
    m, state, x, y, free, taps = real_case
    logits, loss, grads, bufs, post, pool = _run_hip(m, state, x, y, dev, "fp32")
    total, flips, pflips = _decision_report(post, pool, taps, tie_tol=2e-5)
    assert flips <= 2e-5 * total and pflips <= 8, (flips, pflips, total)
    # forward: no pinning needed (a flipped element is ~1e-6 either way)
    ref_logits, ref_loss = free[0], free[1]
    assert relmax(logits, ref_logits) < 1e-4
    assert abs(loss - ref_loss) < 1e-4 * abs(ref_loss)
    assert torch.equal(logits.argmax(1), ref_logits.argmax(1))
    for k, b in bufs.items():
        r = free[3][k]
        if r.dtype == torch.int64:
            assert torch.equal(b, r), k
        else:
            assert relmax(b, r) < 1e-5, k
    # backward: on the linear piece the HIP path evaluated
    dec = {k: v > 0 for k, v in post.items()}
    dec.update(pool)
    pin_logits, _, ref_grads, _ = _oracle(x, y, state, decisions=dec)
    assert relmax(pin_logits, ref_logits) < 1e-5          # pinning near-ties does not move the forward
    gmax = max(float(v.abs().max()) for v in ref_grads.values())
    worst = ("", 0.0)
    for k, gk in grads.items():
        r = ref_grads[k]
        if _noise_bias(k):
            assert float((gk - r).abs().max()) < 1e-5 * gmax, k
        else:
            e = relmax(gk, r)
            if e > worst[1]:
                worst = (k, e)
            assert e < 1e-4, (k, e)
    print(f"fp32 real widths: {flips} ReLU flips of {total}, {pflips} pool flips; logits relmax {relmax(logits, ref_logits):.2e}, "
          f"worst grad relmax {worst[1]:.2e} ({worst[0]})")
    m.cpu()


# Stated bf16 tolerances for the whole model at the benchmark's widths, shared with tests/test_gpu_benchsize.py (512 x 512);
# measured values: DESIGN.md 3.2
BF16_LOGITS_RELL2 = 2e-2
BF16_LOSS_REL = 1e-2
BF16_ARGMAX_AGREE = 0.995
BF16_GRAD_RELL2_PINNED = 8e-2       # per parameter tensor, fp32 oracle on the HIP path's linear piece (measured worst 7.4e-2)
BF16_GRAD_RELL2_EMU = 6e-2          # per parameter tensor, bf16-storage-emulating oracle on the same piece (measured worst 4.1e-2)
BF16_FLIP_FRACTION = 1e-2           # ReLU decisions that differ from the fp32 oracle
BF16_POOL_FLIP_FRACTION = 2e-2      # pool winners that differ (post-ReLU ties between two zeros do not count: same index rule)
# A decision may differ only where the pre-activation y = scale*z + shift is within bf16 rounding noise of the kink.  One
# rounding step of a stored bf16 operand is 2^-8 of its size; the operands of y at y = 0 have size |shift_c| (the stored conv
# output times its scale) and 1 (unit variance of the normalised activation, which carries the noise of the up to 18 layers
# above it).  Measured at 512x512 in units of 2^-8 * (1 + |shift_c|): median 1.8, 99th percentile 17, maximum 62 steps of
# 1.42 M differing decisions (the tail: the noise of a pre-activation scales with the size of the activations under ITS
# 3x3 window, and single pixels carry activations many times the channel's rms).
# Asserted: 99 % of the differing decisions within 32 steps, every one within 128; and in absolute terms (RepUNet's GroupNorm
# branches have no single shift) 99 % below 2^-3 and every one below 2^-1 of the O(1) activation scale.
BF16_TIE_STEPS_Q99 = 32.0
BF16_TIE_STEPS_MAX = 128.0
BF16_TIE_Q99 = 2.0 ** -3
BF16_TIE_MAX = 2.0 ** -1
BF16_GRAD_COSINE_FREE_SMALL = 0.95  # the same at 2 x 64 x 64 (tests/test_gpu_realwidth.py)
BF16_GRAD_COSINE_FREE = 0.999       # concatenated gradient (all tensors but the noise-only biases) vs the FREE-RUNNING fp32 oracle


def test_engine_real_widths_bf16(dev, real_case):
    m, state, x, y, free, taps = real_case
    logits, loss, grads, _, post, pool = _run_hip(m, state, x, y, dev, "bf16")
    ref_logits, ref_loss, free_grads = free[0], free[1], free[2]
    e_log = rell2(logits, ref_logits)
    e_loss = abs(loss - ref_loss) / abs(ref_loss)
    agree = float((logits.argmax(1) == ref_logits.argmax(1)).float().mean())
    # the decisions that differ from the free-running fp32 oracle: how many, and how close to the kink each one is
    total, mags, ptotal, pgaps = _decision_stats(post, pool, taps)
    shapes = {k: v.shape for k, v in post.items()}
    osc = _bn_operand_scale(state, free[3], lambda blk: shapes[blk + ".a"][0] * shapes[blk + ".a"][2] * shapes[blk + ".a"][3])
    steps = _decision_stats(post, {}, taps, operand_scale=osc)[1] * 256.0
    keys = [k for k in grads if not _noise_bias(k)]
    cos_free = _cosine(grads, free_grads, keys)
    dec = {k: v > 0 for k, v in post.items()}
    dec.update(pool)
    _, _, pin_grads, _ = _oracle(x, y, state, decisions=dec)
    worst = max(((rell2(grads[k], pin_grads[k]), k) for k in keys))
    worst_free = max(((rell2(grads[k], free_grads[k]), k) for k in keys))
    print(f"bf16 real widths 64x64: logits rel-L2 {e_log:.3e}, loss rel {e_loss:.3e}, argmax agreement {agree:.5f}; "
          f"{mags.numel()} of {total} ReLU decisions ({mags.numel() / total:.3%}) and {pgaps.numel()} of {ptotal} pool winners differ from "
          f"the fp32 oracle, magnitude q99 {_q(mags, 0.99):.2e} / max {_q(mags, 1.0):.2e} = {_q(steps, 0.99):.1f} / {_q(steps, 1.0):.1f} bf16 "
          f"rounding steps; free-running gradient cosine {cos_free:.5f}; worst grad rel-L2 on the same linear piece "
          f"{worst[0]:.3e} ({worst[1]}), against the free-running fp32 oracle {worst_free[0]:.3e} ({worst_free[1]})")
    assert e_log <= BF16_LOGITS_RELL2 and e_loss <= BF16_LOSS_REL and agree >= BF16_ARGMAX_AGREE
    assert mags.numel() <= BF16_FLIP_FRACTION * total and pgaps.numel() <= BF16_POOL_FLIP_FRACTION * ptotal
    assert _q(steps, 0.99) <= BF16_TIE_STEPS_Q99 and _q(steps, 1.0) <= BF16_TIE_STEPS_MAX
    assert _q(mags, 0.99) <= BF16_TIE_Q99 and _q(mags, 1.0) <= BF16_TIE_MAX
    assert _q(pgaps, 0.99) <= BF16_TIE_Q99 and _q(pgaps, 1.0) <= BF16_TIE_MAX
    # free-running: 8 192 pixels average far less of the decision noise away than the 524 288 of the 512 x 512 test (0.9999 there)
    assert cos_free >= BF16_GRAD_COSINE_FREE_SMALL
    assert worst[0] <= BF16_GRAD_RELL2_PINNED, worst
    m.cpu()


def test_bf16_vs_fp32_loss_trajectory(dev):
    """30 full step_batch calls (fused clip + AdamW) from the same init, fp32 kernels against bf16 kernels: the two
    loss curves must fall together.  Stated bound: |loss_bf16 - loss_fp32| <= 3e-2 * loss_fp32 at every step, and
    both end below 75 % of where they started."""
    import logging
    import types
    from unet_amd.losses import get_loss
    from unet_amd.models import REGISTRY
    from unet_amd.utils.builders import get_optim
    from unet_amd.utils.train_step import step_batch
    g = torch.Generator().manual_seed(7)
    x = torch.rand(4, 1, 64, 64, generator=g).to(dev)
    # a learnable target: the mask is a function of the image
    y = (F.avg_pool2d(x, 5, 1, 2)[:, 0] > 0.55).long()

    def run(amp):
        torch.manual_seed(42)
        m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[16, 32, 64, 128]).to(dev).train()
        opt = get_optim("adamw", m.parameters(), lr=1e-3, weight_decay=1e-4)
        tr = types.SimpleNamespace(optimizer=opt, amp=amp, model=m, grad_clip=1.0, cbs=[], current_epoch=1,
                                   logger=logging.getLogger("t"), loss_fn=get_loss("ce_dice", ignore_index=255))
        out = []
        for i in range(30):
            _, _, info = step_batch(tr, x, y, torch.amp.GradScaler(enabled=False), i)
            out.append(info["main_loss"])
        return out

    a, b = run(False), run(True)
    dev_max = max(abs(p - r) / r for p, r in zip(b, a))
    print(f"loss trajectory fp32 {a[0]:.4f} -> {a[-1]:.4f}, bf16 {b[0]:.4f} -> {b[-1]:.4f}, max relative gap {dev_max:.3e}")
    assert a[-1] < 0.75 * a[0] and b[-1] < 0.75 * b[0]
    assert dev_max <= 3e-2
