"""Per-kernel parity: every HIP kernel (through the C ABI) against the CPU oracle (torch fp32 ops)
on the same seeded inputs.  Needs an MI355X: run with `pytest -m gpu`.

Tolerances (stated per north_star):
  f32 kernels : max |err| <= 2e-5 * max|ref|           (exact-f32 MFMA, different summation order)
  bf16 kernels: inputs are rounded to bf16 for BOTH sides, accumulation is fp32 on both sides, so the
                only difference is the final bf16 rounding of the output: <= 2^-8 relative per element
                (+ 1e-3*max|ref| absolute for cancellation); fp32 outputs (weight grads, stats) 2e-3.
  integer outputs (pool codes/indices, argmax, ensemble, counts): bit-exact.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def ops_mod():
    from unet_amd import ops
    return ops


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dt):
    """round to the compute dtype (and back to fp32) so both sides see identical inputs"""
    return t.bfloat16().float() if dt == 1 else t.clone()


def to_act(x_nchw, dt, dev, ld=None, off=0):
    ops = ops_mod()
    N, C, H, W = x_nchw.shape
    ld = C if ld is None else ld
    buf = torch.zeros((N, H, W, ld), dtype=ops.torch_dtype(dt), device=dev)
    buf[..., off:off + C] = x_nchw.permute(0, 2, 3, 1).to(dev).to(ops.torch_dtype(dt))
    return ops.Act(buf, off, C)


def from_act(a):
    return a.buf[..., a.off:a.off + a.C].float().cpu().permute(0, 3, 1, 2).contiguous()


def close(got, ref, dt, what="", f32_out=False):
    ref_max = float(ref.abs().max().clamp_min(1e-30))
    err = (got - ref).abs()
    if dt == 0:
        tol = 2e-5 * ref_max
        assert float(err.max()) <= tol, f"{what}: max err {float(err.max()):.3e} > {tol:.3e} (ref max {ref_max:.3e})"
    elif f32_out:
        tol = 2e-3 * ref_max
        assert float(err.max()) <= tol, f"{what}: max err {float(err.max()):.3e} > {tol:.3e} (ref max {ref_max:.3e})"
    else:
        bound = ref.abs() * 2 ** -8 + 1e-3 * ref_max
        bad = err > bound
        assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements beyond bf16 rounding, max err {float(err.max()):.3e} (ref max {ref_max:.3e})"


CONV_SHAPES = [
    # N, H, W, Cin, Cout
    (2, 16, 40, 16, 24),
    (1, 8, 32, 64, 64),
    (2, 20, 20, 40, 72),
    (1, 4, 4, 128, 136),
    (3, 9, 33, 8, 8),
    # whole 16x32 tiles and channel multiples of the block: the specialised bf16 3x3 kernels
    (1, 16, 32, 64, 64),
    (2, 32, 64, 32, 128),
    (1, 48, 32, 128, 256),
]
KERNELS = [(3, 3), (1, 1), (5, 5), (3, 1), (1, 3), (5, 1), (1, 5)]


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("k", KERNELS)
@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv_fwd_dgrad_wgrad(dev, dt, k, shape):
    ops = ops_mod()
    N, H, W, Cin, Cout = shape
    kh, kw = k
    x = q(rnd((N, Cin, H, W), 1), dt)
    w = q(rnd((Cout, Cin, kh, kw), 2, (Cin * kh * kw) ** -0.5), dt)
    b = rnd((Cout,), 3)
    x.requires_grad_(True); w.requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=(kh // 2, kw // 2))
    dyv = q(rnd(tuple(ref.shape), 4), dt)
    ref.backward(dyv)

    # forward (+ BN partial statistics), reading a channel slice of a wider buffer
    xa = to_act(x.detach(), dt, dev, ld=Cin + 8, off=8)
    ya = to_act(torch.zeros(N, Cout, H, W), dt, dev, ld=Cout + 16, off=16)
    wp = ops.pack_weights(w.detach().to(dev), 0, dt)
    rows = ops.conv_fwd(xa, wp, b.to(dev), ya, kh, kw, Cin, Cout, stats=True)
    torch.cuda.synchronize()
    close(from_act(ya), ref.detach(), dt, "conv fwd")
    assert float(ya.buf[..., :16].float().abs().max()) == 0.0, "wrote outside the output channel slice"
    s = rows.double().sum(0).cpu()
    r = ref.detach().double()
    close(s[:, 0].float(), r.sum((0, 2, 3)).float(), dt, "stats sum", f32_out=True)
    close(s[:, 1].float(), (r * r).sum((0, 2, 3)).float(), dt, "stats sumsq", f32_out=True)

    # dgrad = conv with flipped / transposed weights
    dza = to_act(dyv, dt, dev)
    dxa = ops.dense(N, H, W, Cin, dt, dev)
    wd = ops.pack_weights(w.detach().to(dev), 1, dt)
    ops.conv_fwd(dza, wd, None, dxa, kh, kw, Cout, Cin)
    close(from_act(dxa), x.grad, dt, "conv dgrad")

    # wgrad
    ws = ops.Workspace(dev)
    if kh * kw <= 9:
        dw, _, _ = ops.conv_wgrad(dza, xa, kh, kw, ws)
        close(dw.view(Cout, Cin, kh, kw).cpu(), w.grad, dt, "conv wgrad", f32_out=True)
    else:
        dw = torch.empty((Cout, Cin, kh, kw), dtype=torch.float32, device=dev)
        for r_ in range(kh):
            ops.conv_wgrad(dza, xa, 1, kw, ws, dw=dw, pad_h=kh // 2 - r_, pad_w=kw // 2, dw_taps=kh * kw, dw_tap0=r_ * kw)
        close(dw.cpu(), w.grad, dt, "conv wgrad 5x5 by rows", f32_out=True)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("geom", [(2, 12, 36, 32, 48, 16), (1, 32, 32, 64, 192, 64), (1, 16, 64, 32, 128, 64)])
def test_conv_split_output(dev, dt, geom):
    ops = ops_mod()
    N, H, W, Cin, Cout, cs = geom
    x = q(rnd((N, Cin, H, W), 5), dt)
    w = q(rnd((Cout, Cin, 3, 3), 6, 0.1), dt)
    ref = F.conv2d(x, w, None, padding=1)
    xa = to_act(x, dt, dev)
    y1, y2 = ops.dense(N, H, W, cs, dt, dev), ops.dense(N, H, W, Cout - cs, dt, dev)
    ops.conv_fwd(xa, ops.pack_weights(w.to(dev), 0, dt), None, y1, 3, 3, Cin, Cout, y2=y2, csplit=cs)
    close(from_act(y1), ref[:, :cs], dt, "split lo")
    close(from_act(y2), ref[:, cs:], dt, "split hi")


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", [(2, 6, 10, 32, 16), (1, 16, 32, 128, 64), (2, 3, 5, 16, 8), (2, 32, 64, 256, 128), (1, 32, 32, 128, 64)])
def test_conv_transpose(dev, dt, shape):
    ops = ops_mod()
    N, h, w_, Ci, Co = shape
    x = q(rnd((N, Ci, h, w_), 7), dt)
    wt = q(rnd((Ci, Co, 2, 2), 8, Ci ** -0.5), dt)
    b = rnd((Co,), 9)
    x.requires_grad_(True); wt.requires_grad_(True); b.requires_grad_(True)
    ref = F.conv_transpose2d(x, wt, b, stride=2)
    dyv = q(rnd(tuple(ref.shape), 10), dt)
    ref.backward(dyv)
    xa = to_act(x.detach(), dt, dev)
    cat = torch.zeros((N, 2 * h, 2 * w_, 2 * Co), dtype=ops.torch_dtype(dt), device=dev)
    ops.conv_fwd(xa, ops.pack_weights(wt.detach().to(dev), 2, dt), b.detach().to(dev), ops.Act(cat, Co, Co), 1, 1, Ci, 4 * Co, epi=1)
    close(from_act(ops.Act(cat, Co, Co)), ref.detach(), dt, "convT fwd")
    assert float(cat[..., :Co].float().abs().max()) == 0.0
    ws = ops.Workspace(dev)
    dup = to_act(dyv, dt, dev)
    dx = ops.dense(N, h, w_, Ci, dt, dev)
    ops.convT_dgrad(dup, ops.pack_weights(wt.detach().to(dev), 3, dt), Ci, dx)
    close(from_act(dx), x.grad, dt, "convT dgrad")
    dw = ops.convT_wgrad(xa, dup, ws)
    close(dw.cpu(), wt.grad, dt, "convT wgrad", f32_out=True)
    db = ops.channel_sum(dup, ws)
    close(db.cpu(), b.grad, dt, "convT dbias", f32_out=True)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("k", [(3, 3), (5, 5), (3, 1), (1, 5)])
@pytest.mark.parametrize("cin", [1, 3])
def test_first_conv(dev, dt, k, cin):
    ops = ops_mod()
    kh, kw = k
    N, H, W, Cout = 2, 12, 20, 24
    x = torch.rand(N, cin, H, W, generator=torch.Generator().manual_seed(11))
    w = rnd((Cout, cin, kh, kw), 12, 0.3)
    b = rnd((Cout,), 13)
    w.requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=(kh // 2, kw // 2))
    dyv = q(rnd(tuple(ref.shape), 14), dt)
    ref.backward(dyv)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    z = ops.dense(N, H, W, Cout, dt, dev)
    rows = ops.conv_first_fwd(xh, w.detach().to(dev), b.to(dev), z, kh, kw)
    close(from_act(z), ref.detach(), dt, "first conv fwd")
    s = rows.double().sum(0).cpu()
    close(s[:, 0].float(), ref.detach().double().sum((0, 2, 3)).float(), dt, "first conv stats", f32_out=True)
    close(s[:, 1].float(), (ref.detach().double() ** 2).sum((0, 2, 3)).float(), dt, "first conv sumsq", f32_out=True)
    dw = ops.conv_first_wgrad(to_act(dyv, dt, dev), xh, kh, kw, ops.Workspace(dev))
    close(dw.cpu(), w.grad, dt, "first conv wgrad", f32_out=True)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", [(3, 40, 70, 64), (1, 16, 32, 8), (2, 33, 31, 128), (1, 64, 64, 32)])
def test_first_conv_cin1_strips(dev, dt, shape):
    """The Cin=1 3x3 sliding-window kernels: several 16-row strips, ragged right / bottom edges, idle lanes."""
    ops = ops_mod()
    N, H, W, Cout = shape
    x = torch.rand(N, 1, H, W, generator=torch.Generator().manual_seed(21))
    w = rnd((Cout, 1, 3, 3), 22, 0.3)
    b = rnd((Cout,), 23)
    w.requires_grad_(True)
    ref = F.conv2d(x, w, b, padding=1)
    dyv = q(rnd(tuple(ref.shape), 24), dt)
    ref.backward(dyv)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    z = ops.dense(N, H, W, Cout, dt, dev)
    rows = ops.conv_first_fwd(xh, w.detach().to(dev), b.to(dev), z, 3, 3)
    close(from_act(z), ref.detach(), dt, "first conv fwd")
    s = rows.double().sum(0).cpu()
    close(s[:, 0].float(), ref.detach().double().sum((0, 2, 3)).float(), dt, "first conv stats", f32_out=True)
    close(s[:, 1].float(), (ref.detach().double() ** 2).sum((0, 2, 3)).float(), dt, "first conv sumsq", f32_out=True)
    dw = ops.conv_first_wgrad(to_act(dyv, dt, dev), xh, 3, 3, ops.Workspace(dev))
    close(dw.cpu(), w.grad, dt, "first conv wgrad", f32_out=True)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("C,K", [(64, 2), (8, 3), (24, 2), (16, 8)])
def test_final_conv(dev, dt, C, K):
    ops = ops_mod()
    N, H, W = 2, 10, 14
    x = q(rnd((N, C, H, W), 15), dt)
    w = rnd((K, C, 1, 1), 16, C ** -0.5)
    b = rnd((K,), 17)
    x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
    ref = F.conv2d(x, w, b)
    dl = rnd(tuple(ref.shape), 18)
    ref.backward(dl)
    xa = to_act(x.detach(), dt, dev, ld=C + 8, off=0)
    logits = ops.conv_out_fwd(xa, w.detach().view(K, C).contiguous().to(dev), b.detach().to(dev))
    close(logits.cpu(), ref.detach(), 0, "final conv fwd")
    dx, dw, db = ops.conv_out_bwd(dl.to(dev), xa, w.detach().view(K, C).contiguous().to(dev), ops.Workspace(dev))
    close(from_act(dx), x.grad, dt, "final conv dx")
    close(dw.cpu(), w.grad, 0, "final conv dw")
    close(db.cpu(), b.grad, 0, "final conv db")


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("C", [8, 24, 64, 1024])
def test_batchnorm_relu(dev, dt, C):
    ops = ops_mod()
    N, H, W = 2, 6, 10
    z = q(rnd((N, C, H, W), 19) * 1.5 + 0.3, dt)
    gamma, beta = rnd((C,), 20).abs() + 0.5, rnd((C,), 21)
    rm, rv = rnd((C,), 22), rnd((C,), 23).abs() + 0.5
    z.requires_grad_(True); gamma.requires_grad_(True); beta.requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.relu(F.batch_norm(z, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    dyv = q(rnd(tuple(y.shape), 24), dt)
    y.backward(dyv)
    ws = ops.Workspace(dev)
    za = to_act(z.detach(), dt, dev)
    # the stat rows a conv epilogue would have produced: one row per image here
    zz = z.detach().double()
    rows = torch.stack([zz.sum((2, 3)), (zz * zz).sum((2, 3))], -1).float().to(dev).contiguous()
    rmd, rvd = rm.to(dev), rv.to(dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    st = ops.bn_finalize(rows, N * H * W, gamma.detach().to(dev), beta.detach().to(dev), rmd, rvd, nbt, 0.1, 1e-5, ws)
    ya = ops.dense(N, H, W, C, dt, dev)
    ops.affine_act(za, ya, st)
    close(from_act(ya), y.detach(), dt, "bn+relu fwd")
    assert int(nbt) == 1
    close(rmd.cpu(), rm_ref, 0, "running_mean")
    close(rvd.cpu(), rv_ref, 0, "running_var")
    dz = ops.dense(N, H, W, C, dt, dev)
    dg, db, dbias = ops.bn_relu_bwd(to_act(dyv, dt, dev), za, dz, st, ws)
    close(dg.cpu(), gamma.grad, dt, "dgamma", f32_out=True)
    close(db.cpu(), beta.grad, dt, "dbeta", f32_out=True)
    close(from_act(dz), z.grad, dt, "bn dz")
    ref_dbias = from_act(dz).sum((0, 2, 3))
    assert float((dbias.cpu() - ref_dbias).abs().max()) <= 1e-3 * float(z.grad.abs().max()) * (N * H * W) ** 0.5
    # eval mode
    ste = ops.bn_eval_coeffs(gamma.detach().to(dev), beta.detach().to(dev), rm.to(dev), rv.to(dev), 1e-5)
    ops.affine_act(za, ya, ste)
    close(from_act(ya), F.relu(F.batch_norm(z.detach(), rm, rv, gamma.detach(), beta.detach(), False, 0.0, 1e-5)), dt, "bn eval")


@pytest.mark.parametrize("dt", [0, 1])
def test_maxpool_indices_bit_exact(dev, dt):
    ops = ops_mod()
    N, C, H, W = 2, 16, 12, 20
    x = F.relu(q(rnd((N, C, H, W), 25), dt))        # post-ReLU: ties (zeros) are the common case
    x[0, 0, 0:2, 0:2] = 0.0
    x[0, 1, 0:2, 0:2] = 1.0
    x[0, 2, 0, 1] = float("nan")
    x[0, 3, 1, 0] = float("nan"); x[0, 3, 1, 1] = float("nan")
    ref, idx = F.max_pool2d(x, 2, 2, return_indices=True)
    xa = to_act(x, dt, dev, ld=2 * C, off=0)
    y, code = ops.maxpool_fwd(xa)
    got = from_act(y)
    assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(ref, nan=-7.0))
    assert torch.equal(ops.pool_indices_int64(code, W).cpu(), idx)          # bit-exact indices
    # backward: unpool + skip gradient
    dyv = q(rnd(tuple(ref.shape), 26), dt)
    dsk = q(rnd((N, C, H, W), 27), dt)
    xr = torch.nan_to_num(x.clone(), nan=0.0).requires_grad_(True)
    # reference scatter with the oracle's own indices
    dref = torch.zeros(N, C, H * W)
    dref.scatter_add_(2, idx.view(N, C, -1), dyv.view(N, C, -1))
    dref = dref.view(N, C, H, W) + dsk
    dx = ops.maxpool_bwd(to_act(dyv, dt, dev), code, to_act(dsk, dt, dev, ld=C + 8, off=8))
    close(from_act(dx), dref, dt, "maxpool bwd")


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("geom", [(2, 16, 12, 20), (1, 64, 8, 8), (3, 24, 6, 10), (1, 1024, 2, 4)])
def test_affine_act_pool_bit_identical_to_two_kernels(dev, dt, geom):
    ops = ops_mod()
    N, C, H, W = geom
    z = q(rnd((N, C, H, W), 41) * 1.5, dt)
    z[0, 0, 0:2, 0:2] = -3.0                             # an all-zero window after ReLU: ties
    z[0, 1, 0, 1] = float("nan")
    st = ops.BNState(rnd((C,), 42).abs().to(dev) + 0.5, rnd((C,), 43).to(dev))
    za = to_act(z, dt, dev, ld=C + 8, off=8)
    y1 = ops.Act(torch.zeros((N, H, W, 2 * C), dtype=ops.torch_dtype(dt), device=dev), 0, C)
    y2 = ops.Act(torch.zeros((N, H, W, 2 * C), dtype=ops.torch_dtype(dt), device=dev), 0, C)
    ops.affine_act(za, y1, st)
    p1, c1 = ops.maxpool_fwd(y1)
    p2, c2 = ops.affine_act_pool(za, y2, st)
    same = lambda a, b: torch.equal(torch.nan_to_num(a.float(), nan=-7.0), torch.nan_to_num(b.float(), nan=-7.0))
    assert same(y1.buf, y2.buf) and same(p1.buf, p2.buf) and torch.equal(c1, c2)


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("geom", [(2, 16, 12, 20), (1, 64, 8, 8), (3, 24, 6, 10), (1, 512, 4, 6)])
def test_bn_relu_bwd_with_fused_unpool(dev, dt, geom):
    """BN+ReLU backward fed by (pooled gradient, argmax code, skip gradient) against torch autograd through
    max_pool2d + skip add + batch_norm + relu, and against the two-kernel path (unpool, then BN backward)."""
    ops = ops_mod()
    N, C, H, W = geom
    z = q(rnd((N, C, H, W), 31) * 1.5 + 0.3, dt)
    gamma, beta = rnd((C,), 32).abs() + 0.5, rnd((C,), 33)
    z.requires_grad_(True); gamma.requires_grad_(True); beta.requires_grad_(True)
    y = F.relu(F.batch_norm(z, None, None, gamma, beta, True, 0.1, 1e-5))
    yq = q(y.detach(), dt)                            # what the pool kernel reads
    dpool = q(rnd((N, C, H // 2, W // 2), 34), dt)
    dskip = q(rnd((N, C, H, W), 35), dt)
    pooled, idx = F.max_pool2d(yq, 2, 2, return_indices=True)
    dy_ref = torch.zeros(N, C, H * W).scatter_add_(2, idx.view(N, C, -1), dpool.view(N, C, -1)).view(N, C, H, W) + dskip
    y.backward(dy_ref)
    ws = ops.Workspace(dev)
    za = to_act(z.detach(), dt, dev)
    zz = z.detach().double()
    rows = torch.stack([zz.sum((2, 3)), (zz * zz).sum((2, 3))], -1).float().to(dev).contiguous()
    st = ops.bn_finalize(rows, N * H * W, gamma.detach().to(dev), beta.detach().to(dev), None, None, None, 0.1, 1e-5, ws)
    _, code = ops.maxpool_fwd(to_act(yq, dt, dev))
    assert torch.equal(ops.pool_indices_int64(code, W).cpu(), idx)
    src = ops.PoolGrad(to_act(dpool, dt, dev), code, to_act(dskip, dt, dev, ld=C + 8, off=8))
    dz = ops.dense(N, H, W, C, dt, dev)
    dg, db, dbias = ops.bn_relu_bwd_pool(src, za, dz, st, ws)
    close(dg.cpu(), gamma.grad, dt, "dgamma", f32_out=True)
    close(db.cpu(), beta.grad, dt, "dbeta", f32_out=True)
    close(from_act(dz), z.grad, dt, "bn dz")
    # two-kernel path on the same inputs: identical in fp32 (in bf16 it rounds the stored dy once more)
    dy2 = ops.maxpool_bwd(src.dpool, code, src.dskip)
    dz2 = ops.dense(N, H, W, C, dt, dev)
    dg2, db2, dbias2 = ops.bn_relu_bwd(dy2, za, dz2, st, ws)
    if dt == 0:
        torch.testing.assert_close(dz.buf, dz2.buf, rtol=1e-6, atol=1e-7)      # fp64 atomics: summation order varies
        torch.testing.assert_close(dg, dg2, rtol=1e-6, atol=1e-7)


def load(name):
    z = np.load(os.path.join(G, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def test_loss_cases_vs_reference_golden(dev):
    ops = ops_mod()
    from unet_amd.losses import get_loss, check_labels
    g = load("loss_cases")
    for tag, cw, dw in (("A", 1.0, 1.0), ("B", 1.0, 1.0), ("C", 1.0, 1.0)):
        lg = g[f"{tag}/logits"].to(dev).requires_grad_(True)
        loss = get_loss("ce_dice", ce_weight=cw, dice_weight=dw, ignore_index=255)(lg, g[f"{tag}/target"].to(dev))
        loss.backward()
        st = loss._unet_status.cpu()
        assert abs(float(loss) - float(g[f"{tag}/loss"])) < 2e-6
        assert abs(float(st[1]) - float(g[f"{tag}/ce"])) < 2e-6 and abs(float(st[2]) - float(g[f"{tag}/dice"])) < 2e-6
        close(lg.grad.cpu(), g[f"{tag}/dlogits"], 0, f"dlogits {tag}")
        check_labels(loss)
    lg = g["A/logits"].to(dev).requires_grad_(True)
    loss = get_loss("ce_dice", ce_weight=0.3, dice_weight=0.7, ignore_index=255)(lg, g["A/target"].to(dev))
    (loss * 2.0).backward()
    assert abs(float(loss) - float(g["D/loss"])) < 2e-6
    close(lg.grad.cpu(), 2.0 * g["D/dlogits"], 0, "dlogits D (upstream grad 2)")
    lg = g["A/logits"].to(dev).requires_grad_(True)
    loss = get_loss("ce", ignore_index=255)(lg, g["E/target"].to(dev))
    loss.backward()
    assert abs(float(loss) - float(g["E/loss"])) < 2e-6
    close(lg.grad.cpu(), g["E/dlogits"], 0, "dlogits E")
    check_labels(loss)
    # the reference's dice_loss raises on label 255; ours reports it at the host sync
    loss = get_loss("ce_dice", ignore_index=255)(g["A/logits"].to(dev), g["E/target"].to(dev))
    with pytest.raises(RuntimeError):
        check_labels(loss)


def test_argmax_ensemble_metrics_vs_reference_golden(dev):
    ops = ops_mod()
    from oracle import loss_ref
    g = load("ensemble")
    l1, l2 = g["logits1"].to(dev), g["logits2"].to(dev)
    assert torch.equal(ops.argmax(l1).cpu(), g["pred1"])          # ties -> lowest class index
    assert torch.equal(ops.argmax(l2).cpu(), g["pred2"])
    pred = ops.ensemble_fuse(l1, l2, float(g["entropy_thresh"])).cpu()
    # The rule (eval_ensemble.py:77-99 per pixel) is discontinuous in two places only: |e1-e2| crossing the threshold
    # and a tie of the fused probabilities.  A pixel within float rounding of either may legitimately fall on the
    # other side of it; everywhere else the integer output must equal the reference's bit for bit.
    p1, p2 = torch.softmax(g["logits1"], 1), torch.softmax(g["logits2"], 1)
    e1, e2 = loss_ref.entropy(p1), loss_ref.entropy(p2)
    w1, w2 = 1 / (e1 + 1e-8), 1 / (e2 + 1e-8)
    pe = (w1[:, None] * p1 + w2[:, None] * p2) / (w1 + w2)[:, None]
    srt = pe.sort(1, descending=True).values
    near_thresh = ((e1 - e2).abs() - 0.6).abs() < 1e-5
    near_tie = (srt[:, 0] - srt[:, 1]) < 1e-6
    # the fixture's deliberate exact ties have an ANALYTIC answer and are checked unmasked:
    #   row 12: both models output logits (0, 0): p1 = p2 = (.5, .5), e1 = e2, fused (.5, .5) -> argmax tie -> class 0
    #   row 13: model 1 outputs (0, 0): it adds the same weight to both classes, so whichever branch is taken the
    #           prediction is model 2's argmax
    assert torch.equal(pred[:, 12, :], torch.zeros_like(pred[:, 12, :]))
    assert torch.equal(pred[:, 13, :], g["logits2"][:, :, 13, :].argmax(1))
    assert torch.equal(pred[:, 12:14, :], g["pred_ens"][:, 12:14, :])
    near_tie[:, 12:14, :] = False
    fragile = near_thresh | near_tie
    n_frag, n_bad = int(fragile.sum()), int((pred[fragile] != g["pred_ens"][fragile]).sum())
    print(f"ensemble: {n_frag} of {fragile.numel()} pixels within rounding of a decision boundary, {n_bad} of them differ from the reference")
    assert n_frag <= 0.01 * fragile.numel()
    assert torch.equal(pred[~fragile], g["pred_ens"][~fragile])
    nan_l = l1.clone(); nan_l[0, 1, 0, 0] = float("nan")
    assert int(ops.argmax(nan_l)[0, 0, 0]) == int(nan_l.cpu().argmax(1)[0, 0, 0])

    from unet_amd.utils.eval_utils import calculate_segmentation_metrics
    j = json.load(open(os.path.join(G, "metrics.json")))
    got = calculate_segmentation_metrics(torch.tensor(j["pred"]).to(dev), torch.tensor(j["target"]).to(dev), 2)
    for k, v in j["metrics"].items():
        assert abs(got[k] - v) < 1e-6, k
    z = torch.zeros(1, 6, 6, dtype=torch.long, device=dev)
    got0 = calculate_segmentation_metrics(z, z, 2)
    for k, v in j["empty_metrics"].items():
        assert abs(got0[k] - v) < 1e-6, k


@pytest.mark.gpu
@pytest.mark.parametrize("nrows", [1, 63, 64, 300, 16384])
def test_rows_colsum_matches_torch(dev, nrows):
    """Column sums of fp32 partial rows (the ConvTranspose2d bias gradient is read off the split dgrad's statistics rows)."""
    from unet_amd import ops
    g = torch.Generator(device=dev).manual_seed(nrows)
    ctot, f = 24, 8
    rows = torch.randn((nrows, ctot, 2), generator=g, device=dev)
    got = ops.rows_colsum(rows, 2 * f, ctot - f, 2)
    ref = rows[:, f:, 0].double().sum(0)
    torch.testing.assert_close(got.double(), ref, rtol=1e-6, atol=1e-4)
    again = ops.rows_colsum(rows, 2 * f, ctot - f, 2)
    assert torch.equal(got, again)                       # fixed summation order
    with pytest.raises(ops.UNetHipError):
        ops.rows_colsum(rows, 2 * f, ctot, 2)            # columns past the end of the row


@pytest.mark.parametrize("C", [2, 3])
def test_ensemble_fuse_large_vs_oracle(dev, C):
    """65 k pixels of mixed-confidence logits (plus exact-tie rows) against the pinned oracle (oracle/loss_ref.py
    ensemble_fuse = eval_ensemble.py:77-99 per pixel).  Integer output: bit-exact except within float rounding of the
    two discontinuities; the size of that set and the mismatches inside it are reported."""
    ops = ops_mod()
    from oracle import loss_ref
    g = torch.Generator().manual_seed(77 + C)
    N, H, W = 4, 128, 128
    l1 = torch.randn(N, C, H, W, generator=g) * torch.rand(N, 1, H, W, generator=g) * 4.0
    l2 = torch.randn(N, C, H, W, generator=g) * torch.rand(N, 1, H, W, generator=g) * 4.0
    l1[:, :, 0:4, :] = 0.0                       # model 1 exactly undecided on four rows
    l2[:, :, 2:6, :] = 0.0                       # model 2 on four rows, two of them shared
    ref = loss_ref.ensemble_fuse(l1, l2, 0.6)
    pred = ops.ensemble_fuse(l1.to(dev), l2.to(dev), 0.6).cpu()
    p1, p2 = torch.softmax(l1, 1), torch.softmax(l2, 1)
    e1, e2 = loss_ref.entropy(p1), loss_ref.entropy(p2)
    w1, w2 = 1 / (e1 + 1e-8), 1 / (e2 + 1e-8)
    pe = (w1[:, None] * p1 + w2[:, None] * p2) / (w1 + w2)[:, None]
    srt = pe.sort(1, descending=True).values
    fragile = (((e1 - e2).abs() - 0.6).abs() < 1e-5) | ((srt[:, 0] - srt[:, 1]) < 1e-6)
    fragile[:, 0:6, :] = False                   # the exact ties are analytic and stay in the checked set
    assert torch.equal(pred[:, 2:4, :], torch.zeros_like(pred[:, 2:4, :]))              # both undecided -> class 0
    assert torch.equal(pred[:, 0:2, :], l2[:, :, 0:2, :].argmax(1))                    # only model 1 undecided
    assert torch.equal(pred[:, 4:6, :], l1[:, :, 4:6, :].argmax(1))                    # only model 2 undecided
    n_frag, n_bad = int(fragile.sum()), int((pred[fragile] != ref[fragile]).sum())
    print(f"ensemble C={C}: {n_frag} of {fragile.numel()} pixels within rounding of a decision boundary, {n_bad} of them differ")
    assert n_frag <= 1e-3 * fragile.numel()
    assert torch.equal(pred[~fragile], ref[~fragile])
